cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof50
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof50 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --reads 20000 --read-len 50000 > /dev/null 2> gpurun_out/prof50.err
f=$(find gpurun_out/prof50 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if int(r["Calls"]) in (3, 6, 9, 12, 15, 18) or "hit_filter" in n or "probe" in n:
        print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e6:9.3f} ms avg  {n[:70]}")
PY
