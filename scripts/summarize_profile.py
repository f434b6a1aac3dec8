"""Turn a rocprofv3 --kernel-trace --stats output directory into a committed summary under profiles/.
usage: python scripts/summarize_profile.py gpurun_out/prof3 profiles/r01_bench_kernel_stats.csv [steps_incl_warmup]"""
import csv, glob, sys
src, dst = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
f = glob.glob(src + "/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
with open(dst, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "calls", "calls_per_step", "total_ms", "ms_per_step", "avg_us", "min_us", "max_us", "pct"])
    for r in rows:
        w.writerow([r["Name"][:160], r["Calls"], f"{int(r['Calls']) / steps:.1f}", f"{float(r['TotalDurationNs']) / 1e6:.3f}",
                    f"{float(r['TotalDurationNs']) / 1e6 / steps:.3f}", f"{float(r['AverageNs']) / 1e3:.1f}",
                    f"{float(r['MinNs']) / 1e3:.1f}", f"{float(r['MaxNs']) / 1e3:.1f}", r["Percentage"]])
print("wrote", dst, len(rows), "kernels")
