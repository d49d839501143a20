#!/bin/bash
# What to run on the first box with more than one MI355X (SCALE_r0N has been a `skipped` record in every round so far: RCCL has never run with
# more than one rank, the CLI's one-thread-per-GPU mode never on two physical devices).  One informative pass:
#   1. bench.py at N = 1, 2, 4, 8 (as far as the node goes), weak AND strong scaling (configs[1] per GPU / configs[2]: one 100k batch sharded),
#      one JSON line each under gpurun_out/scale/, plus a table: Gbp/s, ms per step, EM iterations, scaling efficiency against N = 1;
#   2. per-rank facts of the N-rank runs (MM_BENCH_RANK_LOG=1): ranks of the communicator (ncclCommCount), this rank's step time, its EM iteration latency;
#   3. the CLI: `metamaps mapDirectly --gpus N` (replicated, and --shard-index with the RCCL record gather) and `classify --gpus N`, every output
#      file compared with the --gpus 1 run.
# Usage: tools/scale_check.sh [max_gpus]      (from the repository root)
set -u
cd "$(dirname "$0")/.."
NMAX=${1:-$(python -c 'import torch; print(torch.cuda.device_count())')}
OUT=gpurun_out/scale; mkdir -p $OUT
export MM_BENCH_RANK_LOG=1
FLAGS="--steps 12 --warmup 4 --no-cpu-baseline --no-other-shape --no-e2e-full"
run() {   # n scaling reads tag
  local n=$1 sc=$2 reads=$3 tag=$4 port=$((29500 + RANDOM % 2000))
  if [ "$n" = 1 ]; then python bench.py --gpus 1 --scaling $sc --reads $reads $FLAGS > $OUT/$tag.out 2> $OUT/$tag.err
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --scaling $sc --reads $reads $FLAGS > $OUT/$tag.out 2> $OUT/$tag.err; fi
  tail -1 $OUT/$tag.out > $OUT/$tag.json
}
for n in 1 2 4 8; do
  [ "$n" -le "$NMAX" ] || continue
  run $n weak 100000 weak_$n
  run $n strong 100000 strong_$n
done
python - "$OUT" "$NMAX" <<'PY'
import json, sys, glob, os
out, nmax = sys.argv[1], int(sys.argv[2])
print(f"{'mode':<8}{'N':>3}{'Gbp/s':>10}{'ms/step':>10}{'EM it':>7}{'efficiency':>12}")
for mode in ("weak", "strong"):
    base = None
    for n in (1, 2, 4, 8):
        f = os.path.join(out, f"{mode}_{n}.json")
        if not os.path.exists(f): continue
        try: d = json.load(open(f))
        except Exception as e: print(mode, n, "no JSON line:", e); continue
        assert d["n_gpus"] == n and d["scaling"] == mode, (d["n_gpus"], d["scaling"])
        base = base or d["value"]
        print(f"{mode:<8}{n:>3}{d['value']:>10.2f}{d['ms_per_step']:>10.2f}{d['config']['em_iterations']:>7}{d['value'] / (base * n):>12.3f}")
    for f in sorted(glob.glob(os.path.join(out, f"{mode}_*.err"))):
        for ln in open(f):
            if ln.startswith("RANKLOG"): print("   ", os.path.basename(f), ln.strip())
PY
# ---- the CLI on N devices against one
if [ "$NMAX" -ge 2 ]; then
  D=$(mktemp -d /tmp/scale_cli.XXXX)
  python - "$D" <<'PY'
import sys
sys.path.insert(0, ".")
from metamaps_amd import synth
d = sys.argv[1]
db = synth.make_db(d + "/db", n_genomes=40, genome_len=1_000_000, seed=5)
synth.make_reads(db, d + "/r.fq", n_reads=20000, read_len=8000, seed=3)
PY
  CLI=metamaps_amd/csrc/metamaps
  $CLI mapDirectly --all -r $D/db/DB.fa -q $D/r.fq -o $D/g1 > $D/g1.log 2>&1 && $CLI classify --DB $D/db --mappings $D/g1 --minreads 3 >> $D/g1.log 2>&1
  for mode in "" "--maxmemory-bytes 600000000 --shard-index"; do
    tag=gN$(echo $mode | tr -cd 'a-z' | cut -c1-5)
    $CLI mapDirectly --all -r $D/db/DB.fa -q $D/r.fq -o $D/$tag --gpus $NMAX $mode > $D/$tag.log 2>&1 || { echo "mapDirectly --gpus $NMAX $mode FAILED"; tail -5 $D/$tag.log; continue; }
    $CLI classify --DB $D/db --mappings $D/$tag --minreads 3 --gpus $NMAX >> $D/$tag.log 2>&1 || { echo "classify --gpus $NMAX FAILED"; tail -5 $D/$tag.log; continue; }
    if [ -z "$mode" ]; then
      for suf in "" .meta .meta.unmappedReadsLengths .EM.reads2Taxon; do cmp -s $D/g1$suf $D/$tag$suf && echo "same: $tag$suf" || echo "DIFFERENT: $tag$suf"; done
      python - $D/g1.EM.WIMP $D/$tag.EM.WIMP <<'PY'
import sys
a, b = ([l.split("\t") for l in open(f)] for f in sys.argv[1:3])
ok = len(a) == len(b) and all(x[:4] == y[:4] and all(abs(float(p) - float(q)) <= 1e-5 for p, q in zip(x[4:], y[4:])) for x, y in zip(a[1:], b[1:]))
print("WIMP equal within 1e-5:" , ok)
PY
    else echo "sharded run done (chunked: compare with a --gpus 1 run of the same --maxmemory): $(wc -l < $D/$tag) mapping lines"; fi
    grep -c "ncclCommInitRank\|NCCL INFO" $D/$tag.log > /dev/null
  done
  rm -rf $D
fi
