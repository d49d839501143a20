"""bench.py plumbing that runs without a GPU: the database directory written around device-generated contigs for the CPU baseline /
CLI sample must be one the reference's `classify` accepts (here: the oracle CLI, end to end, with -t N)."""
import json
import os
import subprocess

import numpy as np


def test_sample_db_dir_runs_through_oracle_map_and_classify(oracle_lib, tmp_path):
    import orc
    from metamaps_amd import synth
    rng = np.random.default_rng(4)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = [rng.choice(acgt, size=40_000) for _ in range(5)]
    genomes[3][2000:2600] = ord("N")
    db = synth.write_db_dir(str(tmp_path / "db"), [(7 + 3000 * i, g.tobytes()) for i, g in enumerate(genomes)] + [(7, rng.choice(acgt, size=9_000).tobytes())])
    fq = str(tmp_path / "r.fq")
    with open(fq, "wb") as f:
        for r in range(60):
            g = genomes[r % 5]; p = int(rng.integers(0, 40_000 - 3000)); s = g[p:p + 3000].copy()
            m = rng.random(3000) < 0.05; s[m] = rng.choice(acgt, size=int(m.sum()))
            f.write(f"@r{r}\n".encode() + s.tobytes() + b"\n+\n" + b"I" * 3000 + b"\n")
    out = {}
    for t in ("1", "4"):
        pre = str(tmp_path / f"o{t}")
        p = subprocess.run([orc.CLI, "mapDirectly", "--all", "-r", db["fasta"], "-q", fq, "-o", pre, "-w", "8", "-t", t], capture_output=True, check=True, timeout=600)
        js = json.loads(p.stderr.decode().strip().splitlines()[-1])
        assert js["reads"] == 60 and js["mappings"] >= 50
        subprocess.run([orc.CLI, "classify", "--DB", db["dir"], "--mappings", pre, "-t", t], capture_output=True, check=True, timeout=600)
        out[t] = (open(pre).read(), open(pre + ".EM.WIMP").read(), open(pre + ".EM.reads2Taxon").read())
    assert out["1"] == out["4"]                                   # the threaded index build / mapping of the oracle change nothing
    assert "1000007" in out["1"][2] and os.path.exists(str(tmp_path / "o1.EM.evidenceUnknownSpecies"))


def test_bench_arguments_parse():
    import sys
    import bench
    argv = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1"]
        a = bench.parse_args()
    finally:
        sys.argv = argv
    assert a.shape == "community" and a.reads == 100_000 and a.read_len == 10_000 and a.window == 8
