# end-to-end timing of the CLI on a mid-sized synthetic data set (host parsing / packing / formatting vs device time)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, numpy as np, time
sys.path.insert(0, ".")
d = "/tmp/clit"; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
G, L = 60, 1_000_000
genomes = [rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L) for _ in range(G)]
with open(d + "/DB.fa", "wb") as f:
    for g, s in enumerate(genomes):
        f.write(f">C{g}|kraken:taxid|{g+1}|x\n".encode())
        for i in range(0, L, 80): f.write(s[i:i+80].tobytes() + b"\n")
with open(d + "/reads.fq", "wb") as f:
    for r in range(int(os.environ.get("NR", "30000"))):
        g = int(rng.integers(G)); p = int(rng.integers(0, L - 10000))
        s = genomes[g][p:p+10000].copy()
        m = rng.random(10000) < 0.06
        s[m] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(m.sum()))
        f.write(f"@r{r}\n".encode() + s.tobytes() + b"\n+\n" + b"I" * 10000 + b"\n")
print("files written")
PY
MM_CLI_TIMING=1 metamaps_amd/csrc/metamaps mapDirectly --all -r /tmp/clit/DB.fa -q /tmp/clit/reads.fq -o /tmp/clit/out 2>&1 | grep -E "INFO|rror"
wc -l /tmp/clit/out
if [ -n "$SMALL_BATCH" ]; then
MM_CLI_BATCH_READS=$SMALL_BATCH MM_CLI_TIMING=1 metamaps_amd/csrc/metamaps mapDirectly --all -r /tmp/clit/DB.fa -q /tmp/clit/reads.fq -o /tmp/clit/out2 2>&1 | grep -E "worker" | head -12
fi
if [ -n "$WPG" ]; then for w in $WPG; do echo "workers-per-gpu $w"; MM_CLI_TIMING=1 metamaps_amd/csrc/metamaps mapDirectly --all -r /tmp/clit/DB.fa -q /tmp/clit/reads.fq -o /tmp/clit/out3 --workers-per-gpu $w 2>&1 | grep -E "lap 3 index|lap 8" | tail -2; done; fi
