# 4 000 reads of 75-140 kb (beyond the LDS classes of K2 / K4 / K5): stage times and counts of one batch
python bench.py --read-len 140000 --read-len-min 75000 --reads 4000 --steps 3 --warmup 1 --no-cpu-baseline --no-other-shape --workers 1 2>gpurun_out/long.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms'], d['config']['per_step'], d['config']['host_wall_ms'])"
grep -E "^host" gpurun_out/long.err | tail -24
