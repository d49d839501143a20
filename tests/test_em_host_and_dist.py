"""Host logic of `classify` (metamaps_amd/emhost.py) and the multi-rank EM exchange (metamaps_amd/dist.py,
gloo, world_size 2) against the oracle's doEM on BASELINE config 0-shaped data.  CPU only: the per-rank
E-step here is a numpy restatement standing in for the HIP kernel."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def numpy_estep(prob, lo, hi):
    """E+M partial sums over reads [lo,hi) — what mm_em_iterate computes on the device."""
    def step(f):
        part = np.zeros(len(prob.taxa)); ll = 0.0
        for r in range(lo, hi):
            a, b = prob.read_off[r], prob.read_off[r + 1]
            l = f[prob.taxon[a:b]] * prob.inv_nloc[a:b] * prob.mapq[a:b]
            s = 0.0
            for x in l:
                s += x
            np.add.at(part, prob.taxon[a:b], l / s)
            ll += np.log(s)
        return part, ll
    return step


@pytest.fixture(scope="module")
def classified(tmp_path_factory, oracle_lib):
    import orc
    from metamaps_amd import synth
    d = tmp_path_factory.mktemp("cls")
    db = synth.make_db(str(d / "db"), n_genomes=8, genome_len=40_000, seed=21)
    rd = synth.make_reads(db, str(d / "r.fq"), n_reads=120, read_len=2500, seed=4)
    prefix = str(d / "out")
    subprocess.run([orc.CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", prefix], check=True, capture_output=True, timeout=600)
    p = subprocess.run([orc.CLI, "classify", "--DB", db.dir, "--mappings", prefix], check=True, capture_output=True, timeout=600)
    info = json.loads(p.stderr.decode().strip().splitlines()[-1])
    return {"prefix": prefix, "db": db.dir, "info": info}


def test_emhost_reproduces_oracle_trajectory(classified):
    from metamaps_amd import emhost
    prob = emhost.load_problem(classified["prefix"], classified["db"])
    step = numpy_estep(prob, 0, len(prob.read_ids))

    def full(f):
        part, ll = step(f)
        return part / part.sum(), ll
    f, lls = emhost.run_em(full, len(prob.taxa))
    assert len(lls) == classified["info"]["iterations"]
    assert np.allclose(lls, classified["info"]["ll"], rtol=1e-12)
    wimp = [l.rstrip("\n").split("\t") for l in open(classified["prefix"] + ".EM.WIMP")][1:]
    emf = {r[1]: float(r[4]) for r in wimp if r[0] == "definedGenomes" and r[1] not in ("0", "-3")}
    for t, v in emf.items():
        assert abs(f[prob.taxa.index(t)] - v) < 1e-5 * max(v, 1e-3) + 1e-7     # WIMP prints 6 significant digits


WORKER = r'''
import os, sys, json, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from metamaps_amd import emhost, dist as mmdist
from test_em_host_and_dist import numpy_estep
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[4]}", rank=int(sys.argv[5]), world_size=2)
prob = emhost.load_problem(sys.argv[2], sys.argv[3])
lo, hi = mmdist.shard_range(len(prob.read_ids), dist.get_rank(), 2)
def allreduce(v):
    t = torch.from_numpy(np.ascontiguousarray(v)); dist.all_reduce(t); return t.numpy()
seen = np.zeros(len(prob.taxa)); seen[prob.taxon[prob.read_off[lo]:prob.read_off[hi]]] = 1
f, lls = mmdist.em_distributed(numpy_estep(prob, lo, hi), allreduce, seen)
if dist.get_rank() == 0:
    print(json.dumps({"f": f.tolist(), "ll": lls}))
dist.destroy_process_group()
'''


def test_two_rank_gloo_em_equals_single_rank(classified, tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, classified["prefix"], classified["db"], str(port), str(r)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-2000:] for o in outs]
    res = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert len(res["ll"]) == classified["info"]["iterations"]
    assert np.allclose(res["ll"], classified["info"]["ll"], rtol=1e-12)
    from metamaps_amd import emhost
    prob = emhost.load_problem(classified["prefix"], classified["db"])
    step = numpy_estep(prob, 0, len(prob.read_ids))
    f1, _ = emhost.run_em(lambda f: (lambda p, l: (p / p.sum(), l))(*step(f)), len(prob.taxa))
    assert np.allclose(res["f"], f1, rtol=1e-12, atol=1e-15)


def test_shard_ranges_cover_in_order():
    from metamaps_amd.dist import shard_range
    for n in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
