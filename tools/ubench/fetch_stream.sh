cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/fetch_stream; rm -rf $out; mkdir -p $out
tools/ubench/stream46k 44 | tail -3
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -- tools/ubench/stream46k 44 > /dev/null 2> $out/err
python - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/f/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE":
            print(row["Kernel_Name"][:40], "FETCH_SIZE KiB", row["Counter_Value"], " GB", float(row["Counter_Value"]) * 1024 / 1e9)
PY
