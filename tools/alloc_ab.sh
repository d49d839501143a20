#!/bin/bash
# the device pool with and without the rescue path (MM_NO_POOL_RESCUE=1: pool handed back on every miss under memory pressure, as until round 4's
# last session) on the bench shapes that build chunk indexes: configs 3, 4, the strong-scaling proxy, and (FULL=1) configs[4] at its size
cd $GRAFT_REPO_ROOT
out=gpurun_out/alloc_ab; mkdir -p $out
show() { python -c "
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=j['config']
print(sys.argv[2], 'value', round(j['value'],3), 'ms_per_step', round(j['ms_per_step'],2), 'steps', j['steps'], 'builds', c.get('chunk_index_build_s', c.get('index_build_s')), 'ms_index_builds', (c.get('per_timed_region') or {}).get('ms_index_builds'))" $1 "$2"; }
for mode in new old; do
  if [ $mode = old ]; then export MM_NO_POOL_RESCUE=1; else unset MM_NO_POOL_RESCUE; fi
  timeout 600 python bench.py --config 3 --steps 12 --warmup 2 --no-cpu-baseline > $out/c3_$mode.json 2> $out/c3_$mode.err; show $out/c3_$mode.json "config3 $mode"
  timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $out/c4_$mode.json 2> $out/c4_$mode.err; show $out/c4_$mode.json "config4 $mode"
  timeout 600 python bench.py --scaling strong --reads 12500 --steps 20 --warmup 5 --no-cpu-baseline --no-other-shape --no-e2e-full > $out/strong_$mode.json 2> $out/strong_$mode.err; show $out/strong_$mode.json "strong12500 $mode"
done
unset MM_NO_POOL_RESCUE
if [ -n "$FULL" ]; then timeout 900 python bench.py --config 5 --steps 1 --warmup 0 > $out/c5_new.json 2> $out/c5_new.err; show $out/c5_new.json "config5 new"; fi
