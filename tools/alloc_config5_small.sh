#!/bin/bash
# configs[4]-at-size (bench.py --config 5) at a tenth of its size on a "device" a tenth of the real one (MM_DEVICE_BYTES_CAP), with the
# allocator's trace: which index-scale blocks of a chunk build come from the pool and which from the driver (the driver clears what it
# hands out: at full size that is where a 7 s chunk build spends 5.6 s).  Summary per build on stdout, the raw trace under gpurun_out/.
cd $GRAFT_REPO_ROOT
out=gpurun_out/alloc5; mkdir -p $out
export MM_ALLOC_TRACE=1 MM_INDEX_SCALE_MB=${MM_INDEX_SCALE_MB:-820} MM_DEVICE_BYTES_CAP=${MM_DEVICE_BYTES_CAP:-30923764531} MM_BENCH_RANGE_GBP=0.8
timeout 600 python bench.py --config 5 --scale 1.12 --chunk-gib 15 --reads 10000 --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/bench.json 2> $out/trace.txt
python - $out <<'PY'
import json, re, sys
out = sys.argv[1]
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
c = d["config"]
print("chunks", c["chunks"], "build_s", c["chunk_index_build_s"], "ms_per_step", round(d["ms_per_step"], 1), "ms_index_builds", c["per_timed_region"]["ms_index_builds"])
n_direct = n_reuse = 0; b_direct = b_reuse = 0; ms = 0.0
for l in open(out + "/trace.txt"):
    m = re.match(r"MM_ALLOC_TRACE direct hipMalloc (\d+) bytes ([\d.]+) ms", l)
    if m: n_direct += 1; b_direct += int(m.group(1)); ms += float(m.group(2)); continue
    m = re.match(r"MM_ALLOC_TRACE big block of (\d+) bytes reused for (\d+)", l)
    if m: n_reuse += 1; b_reuse += int(m.group(2))
print(f"index-scale blocks: {n_direct} from the driver ({b_direct / 2**30:.1f} GiB, {ms:.0f} ms in hipMalloc), {n_reuse} from the pool ({b_reuse / 2**30:.1f} GiB)")
PY
