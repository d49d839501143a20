# K5/K6 kernel time when it leaves after phase n (MM_L2_STOP=n; see tools/pmc_l2_traffic_stops.sh), bench workload
for st in ${STOPS:-1 2 4 8 9 5 0}; do
  echo -n "MM_L2_STOP=$st  ms_l2 = "
  MM_L2_STOP=$st timeout 600 python bench.py --no-cpu-baseline --no-other-shape --workers 1 --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['stage_ms']['ms_l2'])"
done
