#!/usr/bin/env python3
"""Timeline of the LAST extract call in a rocprofv3 kernel trace (csv): per kernel start offset, duration, gap to the previous kernel."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a call starts at the first conv kernel after a match / sample kernel
starts = [i for i, r in enumerate(rows) if "conv_wino_kernel<64" in r["Kernel_Name"] and "true, true, true" in r["Kernel_Name"].replace("(bool)1", "true")]
if not starts:
    starts = [i for i, r in enumerate(rows) if "conv_wino" in r["Kernel_Name"]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 25      # the profile-mode calls at the end carry event records: step back
i0 = starts[-skip] if len(starts) >= skip else starts[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
tot = 0
for r in rows[i0:i0 + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - t0) > 3e6:
        break
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"].replace("d2fe::", "").replace("void ", "")[:90]))
    prev = e; tot += e - s
print("kernel time in window: %.1f us" % (tot / 1e3))
