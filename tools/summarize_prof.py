"""Summarise rocprofv3 output dirs produced by tools/profile.sh into a small text file for profiles/."""
import csv, glob, os, sys
from collections import defaultdict

def short(name):
    name = name.replace("void d2fe::", "").replace("d2fe::", "")
    return name[:110]

def main(root):
    out = []
    stats = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
    for f in stats:
        out.append("== kernel stats (%s)" % os.path.relpath(f, root))
        rows = list(csv.DictReader(open(f)))
        for r in rows[:25]:
            out.append("  %-112s calls=%-5s total_ms=%9.3f avg_us=%10.2f pct=%s" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    # per-kernel median / min from the raw trace (the stats CSV averages the cold warm-up launches in)
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        out.append("== kernel trace: median / min / max duration per launch (us), d2fe kernels")
        for k in sorted(dur, key=lambda k: -sum(dur[k])):
            if "at::native" in k or "rocclr" in k or "anonymous" in k:
                continue
            v = sorted(dur[k])
            out.append("  %-112s n=%-4d median=%10.2f min=%10.2f max=%10.2f" % (k, len(v), v[len(v) // 2], v[0], v[-1]))
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_lds"):
        fs = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
        for f in fs:
            agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
            out.append("== %s per-dispatch averages" % sub)
            watch = ("match_kernel", "softmax_cand", "select_b", "sample_b", "desc_head_sparse", "conv1x1_256_65")      # the HBM / latency-bound kernels north_star names
            top = sorted(agg, key=lambda k: -sum(agg[k].values()))
            for k in top[:14] + [k for k in top[14:] if any(w in k for w in watch)]:
                out.append("  " + k)
                out.append("      " + "  ".join("%s=%.4g" % (c, v / cnt[(k, c)]) for c, v in sorted(agg[k].items())))
    print("\n".join(out))

if __name__ == "__main__":
    main(sys.argv[1])
