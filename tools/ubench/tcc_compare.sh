cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/tcc; rm -rf $out; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -oE "TCC_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > $out/avail.txt; echo >> $out/avail.txt
for set in "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_READ_SECTORS_sum TCC_READ_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/u_$n -- tools/ubench/stream46k 44 > /dev/null 2> $out/u_$n.err
  MM_L2_STOP=2 timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/k_$n -- python bench.py --steps 1 --warmup 0 --workers 1 --no-cpu-baseline --no-other-shape > /dev/null 2> $out/k_$n.err
done
python - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag, pat in (("ubench st<8> (18.68 GB requested per launch)", "st<8>"), ("l2_kernel 4-wave MM_L2_STOP=2", "l2_kernel<true, unsigned char, 4")):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{'u' if 'ubench' in tag else 'k'}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(tag)
    for k, v in sorted(acc.items()): print(f"   {k:28s} first launch {v[0]:.4g}  (n={len(v)})")
PY
