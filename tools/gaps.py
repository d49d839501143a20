"""GPU idle gaps inside one bench step, from a rocprofv3 --kernel-trace CSV: which host section leaves the device waiting.
usage: python tools/gaps.py <dir with *_kernel_trace.csv> [min_gap_us]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
starts = [i for i, r in enumerate(rows) if "minimizer_kernel<2>" in r[2]]
if len(starts) < 2: sys.exit("need two steps")
a, b = starts[-2], starts[-1]                                    # the last complete step
step = rows[a:b]
wall = (step[-1][1] - step[0][0]) / 1e3
busy = 0.0; end = step[0][0]; gaps = []
for s, e, n in step:
    if s > end: gaps.append(((s - end) / 1e3, prev, n))
    busy += (max(e, end) - max(s, end)) / 1e3 if e > end else 0
    if e > end: end = e; prev = n
    elif s <= end: pass
print(f"step: {len(step)} kernels, wall {wall/1e3:.2f} ms, busy {busy/1e3:.2f} ms, idle {(wall-busy)/1e3:.2f} ms")
short = lambda n: n.split("(")[0][:60]
tot = 0
for g, p, n in sorted(gaps, reverse=True):
    if g < min_gap: break
    tot += g
    print(f"  {g:8.1f} us  after {short(p):60s} before {short(n)}")
print(f"gaps >= {min_gap} us: {tot/1e3:.2f} ms")
if len(sys.argv) > 3:                                            # context of the largest gaps
    big = sorted(((step[i + 1][0] - max(x[1] for x in step[:i + 1])) / 1e3, i) for i in range(len(step) - 1))[-int(sys.argv[3]):]
    for g, i in sorted(big, key=lambda t: t[1]):
        print(f"--- gap {g:.0f} us at +{(step[i][1] - step[0][0]) / 1e6:.2f} ms")
        for j in range(max(0, i - 3), min(len(step), i + 4)):
            print(f"     {'>' if j == i + 1 else ' '} {(step[j][1] - step[j][0]) / 1e3:8.1f} us  {short(step[j][2])}")
