# end-to-end timing of `metamaps classify` (host text parsing + device EM + output files) on a synthetic DB with taxonomy
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, time
sys.path.insert(0, ".")
from metamaps_amd import synth
d = "/tmp/clst"; os.makedirs(d, exist_ok=True)
t = time.time()
db = synth.make_db(d + "/db", n_genomes=24, genome_len=1_000_000, seed=11, contigs_per_genome=3)
rd = synth.make_reads(db, d + "/reads.fq", n_reads=int(os.environ.get("NR", "50000")), read_len=10_000, seed=5)
print("data", round(time.time() - t, 1), "s", db.fasta)
PY
B=metamaps_amd/csrc/metamaps
t0=$(date +%s%N); $B mapDirectly --all -r /tmp/clst/db/DB.fa -q /tmp/clst/reads.fq -o /tmp/clst/out > /dev/null; t1=$(date +%s%N); echo "mapDirectly $(( (t1 - t0) / 1000000 )) ms"
wc -l /tmp/clst/out
t0=$(date +%s%N); MM_CLI_TIMING=1 $B classify --DB /tmp/clst/db --mappings /tmp/clst/out --minreads 20 > /dev/null; t1=$(date +%s%N); echo "classify $(( (t1 - t0) / 1000000 )) ms"
ls -la /tmp/clst/out.EM* | awk '{print $5, $9}'
