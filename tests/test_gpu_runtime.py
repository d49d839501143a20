"""The library must give the same answers whether it runs on the system HIP runtime (plain process, e.g. the
C++ CLI) or on the runtime PyTorch bundles and loads first (pytest / bench.py).  A stream-ordered-allocator
problem once made the two differ; this test keeps them tied."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, json
sys.path.insert(0, sys.argv[1])
if sys.argv[2] == "torch":
    import torch; torch.cuda.set_device(0)
from metamaps_amd import capi
ctx = capi.Context(0)
ref = ctx.synth_reference(seed=3, n_species=48, strains_per_species=4, genome_len=500000, strain_divergence=0.02, genus_divergence=0.2)
idx = ctx.index(ref, 16, 8)
reads, _ = ctx.synth_reads(ref, seed=4, n_reads=1500, read_len=9000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=50)
out = []
for it in range(2):
    M = ctx.map_batch(idx, reads, 16, 8); M.add_qualities(16)
    off, rec = M.fetch(); s = M.stats(); M.close()
    out.append([int(off[-1]), int(rec["ref_start"].astype("int64").sum()), int(rec["shared"].sum()), s["sum_hits"], s["n_candidates"], idx.info()["n_unique_hashes"]])
print(json.dumps(out))
'''


def test_same_results_with_and_without_pytorch_runtime(tmp_path):
    p = tmp_path / "w.py"
    p.write_text(SCRIPT)
    res = {}
    for mode in ("plain", "torch"):
        r = subprocess.run([sys.executable, str(p), ROOT, mode], capture_output=True, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        res[mode] = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert res["plain"] == res["torch"]
    assert res["plain"][0] == res["plain"][1] and res["plain"][0][0] > 3000


def test_cached_blocks_of_one_context_serve_another():
    """A context keeps the device blocks it frees (its allocator's cache) and gives them up when one of ITS allocations fails for lack of
    memory.  Worker contexts beside the context that built the indexes need that memory too: an allocation that fails now also trims the
    caches of the other contexts of the device (mm_common.hpp: alloc_trim_others).  Context A fills three quarters of the free memory with
    ~3 GiB sequence sets and closes them (all cached, nothing back at the driver); context B then allocates half of what was free."""
    from metamaps_amd import capi
    a, b = capi.Context(0), capi.Context(0)
    free0 = a.device_info()["hbm_free"]
    if free0 < (48 << 30):
        pytest.skip("needs 48 GiB of free device memory")
    shape = dict(n_species=60, strains_per_species=4, genome_len=50_000_000, strain_divergence=0.02, genus_divergence=0.2)   # 12 Gbases = 3 GiB packed
    sets = []
    while a.device_info()["hbm_free"] > free0 // 4:
        sets.append(a.synth_reference(seed=len(sets) + 1, **shape))
    n_a = len(sets)
    for s_ in sets:
        s_.close()
    a.synchronize()
    assert a.device_info()["hbm_free"] < free0 // 3              # closed, but cached by A
    want = int(free0 // 2 // (3 << 30))
    assert want > n_a // 4 + 1                                    # more than what is free without A's cache
    sets = [b.synth_reference(seed=100 + i, **shape) for i in range(want)]   # (fails with MM_ERR_NOMEM without the cross-context trim)
    assert all(s_.total_bases == 12_000_000_000 for s_ in sets)
    for s_ in sets:
        s_.close()
    ref = a.synth_reference(seed=3, n_species=8, strains_per_species=2, genome_len=200_000, strain_divergence=0.02, genus_divergence=0.2)
    idx = a.index(ref, 16, 8)                                     # A works as before
    assert idx.info()["n_entries"] > 100_000
    idx.close(); ref.close(); b.close(); a.close()


def test_strict_env_refuses_unknown_switches():
    """MM_STRICT_ENV=1: a MM_* variable that metamaps_amd/csrc/mm_env.hpp does not list (a misspelt switch) fails mm_ctx_create and the CLI instead of being ignored"""
    code = "import sys; sys.path.insert(0, sys.argv[1])\nfrom metamaps_amd import capi\ntry:\n    capi.Context(0); print('created')\nexcept capi.MMError as e:\n    print('refused', e.status)\n"
    cli = os.path.join(ROOT, "metamaps_amd", "csrc", "metamaps")
    for extra, expect in (({}, "created"), ({"MM_L2_FUL": "1"}, "refused -1"), ({"MM_L2_FULL": "1", "MM_BENCH_READS": "7"}, "created")):
        env = dict(os.environ, MM_STRICT_ENV="1", **extra)
        r = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, timeout=600, env=env)
        assert r.stdout.decode().strip().splitlines()[-1] == expect, (extra, r.stdout.decode(), r.stderr.decode()[-500:])
        if "MM_L2_FUL" in extra:
            assert "MM_L2_FUL" in r.stderr.decode()
            c = subprocess.run([cli, "classify", "--DB", "/nonexistent", "--mappings", "/nonexistent"], capture_output=True, timeout=600, env=env)
            assert c.returncode == 1 and "MM_L2_FUL" in c.stderr.decode()
