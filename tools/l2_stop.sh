for st in 1 2 3 4 5 0; do
  echo "== MM_L2_STOP=$st"
  MM_L2_STOP=$st timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['stage_ms']['ms_l2'])"
done
