#!/bin/bash
# allocator regimes (mm_common.hpp): step times of the default bench (three runs: no step may stall), configs 3 and 4
cd $GRAFT_REPO_ROOT
for i in ${RUNS:-1 2 3}; do timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e-full --no-other-shape 2> /dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=j['config']; print('default', round(j['value'],2), round(j['ms_per_step'],2), j['step_ms']['all'], 'index_build_s', c['index_build_s'], 'one batch', round(c['one_batch_repeated']['ms_per_step'],2))"; done
MM_ALLOC_TRACE=1 timeout 600 python bench.py --config 3 --steps 5 --warmup 2 2> gpurun_out/c3.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config3', j['value'], j['ms_per_step'], j['step_ms'])"
grep -c "MM_ALLOC_TRACE" gpurun_out/c3.err
timeout 600 python bench.py --config 4 --steps 3 --warmup 1 2> /dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config4', j['value'], j['ms_per_step'], j['step_ms'])"
