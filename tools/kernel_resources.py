"""VGPRs / spills / occupancy of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: python tools/kernel_resources.py netvlad_pair.hip [name filter]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d2slam_amd import build as hb
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = [hb._hipcc()] + hb.FLAGS + hb.EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(hb.CSRC, src), "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"]
r = subprocess.run(cmd, capture_output=True, text=True)
cur = None
for l in r.stderr.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", l)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("d2fe::", "").replace("(d2fe::NvBlockArgs)", "")
        vals = {}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", l)
    if m and cur:
        vals[m.group(1).strip()] = m.group(2)
        if m.group(1).startswith("LDS Size"):
            if flt in cur:
                print("%-60s vgpr %4s agpr %4s scratch %4s spill %3s occ %s" % (cur[:60], vals.get("VGPRs"), vals.get("AGPRs"), vals.get("ScratchSize"), vals.get("VGPRs Spill"), vals.get("Occupancy")))
            cur = None
