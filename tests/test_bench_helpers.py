"""bench.py helper: the 6-significant-digit text round trip used for mapping qualities (mapWrap.h:318 → fEM.h:265)."""
import numpy as np


def test_parse6_equals_print_and_parse():
    import bench
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.random(20000), 10.0 ** rng.uniform(-300, 0, 20000), [1.0, 0.5, 0.999999499, 0.9999995, 1e-5, 123456.5e-10, 0.0]])
    got = bench.parse6(v)
    exp = np.array([float(f"{x:g}") for x in v])
    rel = np.abs(got - exp) / np.maximum(exp, 1e-320)
    assert np.all((got == exp) | (rel < 4e-16))          # identical up to one ulp of the power-of-ten scaling
    big = v >= 1e-16                                       # power-of-ten scaling exact: the round trip is reproduced bit for bit
    assert np.mean(got[big] == exp[big]) > 0.999
