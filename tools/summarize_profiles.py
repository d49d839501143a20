"""Turns the rocprofv3 CSV output of tools/collect_profiles.sh into the text/json summaries kept under profiles/."""
import collections, csv, glob, json, os, sys

out = sys.argv[1]

def short(name):
    n = name.split("(")[0].replace("void ", "")
    if "rocprim" in n:
        n = "rocprim::" + ("radix_sort_onesweep" if "onesweep" in name else "radix_sort_histogram" if "histogram" in name else "other")
    return n[:60]

for sub, name, what in (("stats", "kernel_stats_default_cmd.txt", "python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-shape --no-e2e-full   (the default scheduling: three worker contexts, the kernels of their steps share the GPU, so a kernel's duration includes what it waits for and runs beside"),
                        ("stats_serialised", "kernel_stats.txt", "MM_L2_ONE_STREAM=1 python bench.py --steps 6 --warmup 2 --workers 1 --no-cpu-baseline --no-other-shape --no-e2e-full   (ONE worker context and K5's two launches one behind the other: every duration is that of a kernel that owns the GPU")):
    fs = glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not fs: continue
    rows = list(csv.DictReader(open(fs[0])))
    with open(os.path.join(out, name), "w") as w:
        w.write(f"# rocprofv3 --kernel-trace --stats -- {what}; setup kernels: index build, synthetic data; per-step kernels: warm-up + timed steps (a different read batch each) + the steps behind the timed region (lock held to the end, one batch repeated)\n")
        w.write(f"{'calls':>7} {'total_ms':>12} {'avg_ms':>12} {'%':>7}  kernel\n")
        for r in rows:
            w.write(f"{int(r['Calls']):>7} {float(r['TotalDurationNs'])/1e6:>12.3f} {float(r['AverageNs'])/1e6:>12.3f} {float(r['Percentage']):>7.3f}  {short(r['Name'])}\n")

def pmc(kind):
    f = glob.glob(os.path.join(out, kind, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    return acc

fe, wr = pmc("fetch"), pmc("write")

def fetch_corr(kernel):
    return 1.0 if ("hit_filter_kernel" in kernel or "probe_kernel" in kernel or "seed_filter" in kernel) else 2.0

with open(os.path.join(out, "pmc_hbm_traffic.txt"), "w") as w:
    w.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE   and   --pmc WRITE_SIZE   (separate passes), bench.py --steps 1 --warmup 0 --workers 1 --distinct-batches 1 --no-cpu-baseline --no-other-shape --no-e2e-full\n")
    w.write("# counter unit: KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> x2;\n")
    w.write("# kernels whose reads are random pieces of <= 64 bytes (seed filter, hit filter, probe: 4 lanes x 16 B) are counted exactly -> x1\n")
    w.write("# (calibration on a known byte count in that access pattern: profiles/r01_fetch_size_calibration.txt).\n")
    w.write(f"{'kernel':<62} {'launches':>8} {'FETCH_KiB':>14} {'corr':>5} {'fetch_GB':>10} {'WRITE_KiB':>14} {'write_GB':>9}\n")
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        w.write(f"{k:<62} {fe[k][0]:>8} {fe[k][1]:>14.0f} {fetch_corr(k):>5.0f} {fe[k][1]*1024*fetch_corr(k)/1e9:>10.2f} {wr.get(k,[0,0])[1]:>14.0f} {wr.get(k,[0,0])[1]*1024/1e9:>9.2f}\n")
traffic = {}
for k in fe:
    if "l2_kernel" in k or "l2z_kernel" in k or "hit_filter_kernel<false>" in k or "seed_filter" in k or "minimizer_kernel<2>" in k:
        traffic[k] = (fe[k][1] * 1024 * fetch_corr(k) + wr.get(k, [0, 0])[1] * 1024) / max(fe[k][0], 1)
json.dump({"shape": "community", "reads": 100000, "read_len": 10000,
           "source": "profiles/r06_pmc_hbm_traffic.txt (FETCH_SIZE*1024*corr + WRITE_SIZE*1024 per launch, separate rocprofv3 --pmc passes; corr = 1 for the kernels that read "
                     "random pieces of <= 64 bytes, 2 for streaming kernels: profiles/r01_fetch_size_calibration.txt; tools/collect_profiles.sh)",
           "by_kernel": traffic}, open(os.path.join(out, "traffic_by_kernel.json"), "w"), indent=1)
print(open(os.path.join(out, "kernel_stats.txt")).read()[:3000])
print(open(os.path.join(out, "pmc_hbm_traffic.txt")).read()[:3000])
print(traffic)
