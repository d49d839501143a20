# oracle thread scaling on the GPU box's host cores (informs bench.py's cpu_baseline thread count)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, subprocess, json, tempfile
sys.path.insert(0, ".")
from metamaps_amd import synth
d = tempfile.mkdtemp()
db = synth.make_db(d + "/db", n_genomes=8, genome_len=400_000, seed=7)
rd = synth.make_reads(db, d + "/reads.fq", n_reads=4000, read_len=8000, seed=3)
subprocess.run(["make", "-s", "-C", "oracle", "all"], check=True)
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for t in (1, 8, 16, 32, 64, 128, 256):
    p = subprocess.run(["oracle/_build/metamaps_oracle", "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", d + "/o", "-t", str(t)], capture_output=True)
    js = json.loads(p.stderr.decode().strip().splitlines()[-1])
    print(t, "threads: map_seconds", round(js["map_seconds"], 3), "Gbp/s", round(js["bases"] / js["map_seconds"] / 1e9, 4), flush=True)
PY
