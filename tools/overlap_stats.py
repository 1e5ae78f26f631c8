#!/usr/bin/env python3
"""Concurrency statistics of a rocprofv3 kernel trace (csv): over the last `frac` of the trace, the union of kernel intervals (GPU busy),
the sum of kernel durations (average concurrency = sum / union), per-kernel average duration and launch count, per-queue kernel counts.
Usage: overlap_stats.py <trace dir> [frac=0.5]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t_lo = int(rows[0]["Start_Timestamp"]); t_hi = max(int(r["End_Timestamp"]) for r in rows)
cut = t_hi - (t_hi - t_lo) * frac
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
union = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        union += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
tot = sum(e - s for s, e in iv)
wall = iv[-1][1] - iv[0][0]
print("window %.3f ms, %d kernels; GPU busy (union) %.3f ms = %.1f %%; sum of kernel time %.3f ms; average concurrency %.2f"
      % (wall / 1e6, len(iv), union / 1e6, 100.0 * union / wall, tot / 1e6, tot / max(union, 1)))
by = collections.defaultdict(list)
q = collections.Counter()
for r in rows:
    n = r["Kernel_Name"].replace("d2fe::", "").replace("void ", "")
    n = n.split("(")[0][:70]
    by[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    q[r.get("Queue_Id", "?")] += 1
print("queues:", dict(q))
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%9.1f us total  %5d x  avg %7.1f  p50 %7.1f  min %7.1f  max %7.1f  %s" % (sum(v) / 1e3, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3, n))
