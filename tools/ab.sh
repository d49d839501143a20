#!/bin/bash
# tools/ab.sh ROUNDS NAME1 NAME2 ...  — bench steps with each library variant in turns on this box ("main" = the in-tree build, "env:KEY=VAL" = the in-tree build under that
# environment switch, others: _ab/NAME from tools/ab_build.sh).
# One line per run: variant, Gbp/s, ms per step, K1 / K3 / K5 alone and inside the timed region.
cd "$(dirname "$0")/.."
rounds=$1; shift
F=${AB_FLAGS:-"--steps 12 --warmup 3 --no-cpu-baseline --no-other-shape --no-e2e-full"}
for i in $(seq $rounds); do
for v in "$@"; do
  unset MM_LIB_PATH $AB_LAST_ENV; AB_LAST_ENV=""
  case $v in
    main) ;;
    env:*) kv=${v#env:}; export "$kv"; AB_LAST_ENV=${kv%%=*} ;;          # the in-tree build under an environment switch, e.g. env:MM_L2_ONE_STREAM=1
    *) export MM_LIB_PATH=$PWD/_ab/$v/libmetamaps_hip.so ;;
  esac
  python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['kernels']
print('$v', round(d['value'],2), 'Gbp/s', round(d['ms_per_step'],2), 'ms/step | alone / timed:', ' '.join(f\"{n} {k[n]['ms_alone']:.2f}/{k[n]['ms_timed_region']:.2f}\" for n in k), '| index build', d['config']['index_build_s'], 's | stage', d['config']['stage_ms'])"
done; done
