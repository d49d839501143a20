cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r02
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err
f=$(find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1)
echo $f; head -30 $f | cut -c1-200
