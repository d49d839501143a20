"""Golden table for the Boost.Math binomial calls of the reference (map_stats.hpp:88,204; mapWrap.h:340).
Boost is not vendored in the reference and not installed here; scipy 1.15.3 embeds Boost.Math for
binom.pmf / sf / isf, so its answers are the third-party library's own.  q = (float)0.05 as in
md_lower_bound (map_stats.hpp:82).  Output: tests/golden/binom_golden.json"""
import json
import math
import os

import numpy as np
import scipy
import scipy.stats as st

q = float(np.float32((1.0 - np.float32(0.9)) / 2))


def md2j(d, k):
    return float(np.float32(1.0 / (2.0 * math.exp(k * float(d)) - 1.0)))


def j2md(j, k):
    j = np.float32(j)
    if j == 0:
        return np.float32(1.0)
    if j == 1:
        return np.float32(0.0)
    return np.float32((-1.0 / k) * math.log(2.0 * float(j) / float(np.float32(1) + j)))


rng = np.random.default_rng(42)
quant, pmf, sf = [], [], []
for n in [1, 2, 3, 7, 20, 100, 250, 453, 500, 785, 1176, 2353, 5000, 12000]:
    shs = sorted(set([0, 1, 2, n // 100, n // 20, n // 12, n // 10, n // 5, n // 2, n - 1, n] + [int(x) for x in rng.integers(0, n + 1, 12)]))
    for sh in shs:
        if not 0 <= sh <= n:
            continue
        p = md2j(j2md(np.float32(1.0 * sh / n), 16), 16)
        quant.append([n, p, q, int(st.binom.isf(q, n, p))])
for n in [10, 453, 785, 2353, 4985, 12000]:
    for p in [1e-9, 1e-3, 0.0421, 0.0791, 0.1388, 0.3, 0.5, 0.9, 0.999]:
        for kk in sorted(set([0, 1, n // 50, n // 12, n // 3, n // 2, n - 1, n])):
            pmf.append([n, p, kk, float(st.binom.pmf(kk, n, p))])
            sf.append([n, p, kk, float(st.binom.sf(kk, n, p))])
out = {"generator": f"scipy {scipy.__version__} (Boost.Math)", "q": q, "quantile_upper": quant, "pmf": pmf, "sf": sf}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "binom_golden.json"), "w") as f:
    json.dump(out, f)
print(len(quant), len(pmf), len(sf))
