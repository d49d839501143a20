# K5 with its workgroups in the order of their first candidate (the default), unsorted (MM_L2_NO_GROUP_SORT=1) and dealt out per XCD (MM_L2_XCD_ORDER=1): bench step, K5 alone
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "position_order or mapping_stages or skip_ahead" 2>&1 | tail -2
F="--steps 20 --warmup 5 --no-cpu-baseline --no-other-shape --no-e2e-full"
for i in 1 2; do
for m in sorted plain xcd; do
  unset MM_L2_NO_GROUP_SORT MM_L2_XCD_ORDER; if [ $m = plain ]; then export MM_L2_NO_GROUP_SORT=1; fi; if [ $m = xcd ]; then export MM_L2_XCD_ORDER=1; fi
  python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$m', round(d['value'],2), round(d['ms_per_step'],2), 'l2 alone', round(d['roofline']['ms_per_launch'],2), 'frac', round(d['roofline']['frac'],4), 'l2 timed', round(d['roofline']['ms_per_launch_timed_region'],2), d['config']['stage_ms'])"
done; done
