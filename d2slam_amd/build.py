"""Builds d2slam_amd/lib/libd2fe_hip.so (gfx950) with hipcc.  No torch involvement: the library is a plain
C-ABI shared object (include/d2fe.h)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libd2fe_hip.so")
SOURCES = ["api.hip", "conv.hip", "conv_f16.hip", "conv_pc.hip", "conv_wino.hip", "conv1x1.hip", "postproc.hip", "match.hip", "netvlad.hip", "netvlad_fused.hip", "netvlad_pair.hip", "next.hip", "swarm.hip", "lk.hip"]
HEADERS = ["kernels.h", "conv_common.h", os.path.join("..", "..", "include", "d2fe.h")]
# -ffp-contract=off: the post-processing arithmetic must follow the oracle operation by operation
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"]


# per-file extras.  conv_wino.hip: the persistent kernels claim their next work item with one returning atomicAdd per item whose result is consumed an
# item later; the atomic optimizer's wave-wide rewrite would wait for it immediately (see the kernel)
EXTRA_FLAGS = {"conv_wino.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
