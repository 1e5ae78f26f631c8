#!/usr/bin/env python3
"""Per-dispatch timeline of ONE pass of the frames-in-flight pipe from a rocprofv3 kernel trace (tools/gpu_run.sh py ... or:
   rocprofv3 --output-format csv --kernel-trace -d <dir> -o t -- python tools/pipe_probe.py --sweep 1x1 --seconds 0.05;  python tools/pipe_timeline.py <dir>)
Prints every kernel between the last two fused-conv1b launches: start offset, duration, gap to the previous kernel's end, stream (queue) id."""
import csv, glob, sys

d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "conv_wino_kernel<64, true, true, 0, 1, true" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += (e - s) / 1e3
    print("%9.1f us  dur %7.1f  gap %7.1f  q %-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("d2fe::", "")[:90]))
    prev_end = max(prev_end, e)
print("kernels: %d, sum of durations %.1f us, pass period %.1f us" % (b - a, tot, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
