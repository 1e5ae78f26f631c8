"""Builds d2slam_amd/lib/libd2fe_hip.so (gfx950) with hipcc.  No torch involvement: the library is a plain
C-ABI shared object (include/d2fe.h)."""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libd2fe_hip.so")
SOURCES = ["api.hip", "conv.hip", "conv_f16.hip", "conv_pc.hip", "conv_wino.hip", "conv1x1.hip", "postproc.hip", "match.hip", "netvlad.hip", "netvlad_fused.hip", "netvlad_pair.hip", "next.hip", "swarm.hip", "lk.hip", "pipe.hip", "exchange.hip"]
HEADERS = ["kernels.h", "conv_common.h", "context.h", "stream_deal.h", os.path.join("..", "..", "include", "d2fe.h"), os.path.join("..", "..", "include", "d2fe_debug.h")]
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


def needs_build(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


DEV_LIB = os.path.join(LIBDIR, "libd2fe_hip_dev.so")     # -DD2FE_DEVTOOLS: phase stamps, ablation switches, d2fe_debug_* exports


def build(force=False, verbose=False, dev=False):
    lib = DEV_LIB if dev else LIB
    if not force and not needs_build(lib):
        return lib
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj_dev" if dev else "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    hdr_t = max(os.path.getmtime(os.path.join(CSRC, x)) for x in HEADERS)
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))

    fallbacks, recompiled = [], []      # sources whose EXTRA_FLAGS were dropped in this run / compiled in this run

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        # incremental: an object newer than its source, every header and this script is kept
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(os.path.join(CSRC, src)), hdr_t):
            return obj
        cmd = [hipcc] + FLAGS + (["-DD2FE_DEVTOOLS"] if dev else []) + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        recompiled.append(src)
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 and EXTRA_FLAGS.get(src):
            # the per-file extras are performance flags of newer toolchains (-mllvm options): a hipcc that does not know them must not fail the build --
            # but the library it produces is NOT the tuned one (other scheduling, measurably slower), so say so loudly and record it (build_info)
            first_err = (r.stderr or r.stdout).strip().splitlines()[-1:] or [""]
            cmd = [c for c in cmd if c not in EXTRA_FLAGS[src]]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode == 0:
                print("d2slam_amd.build: WARNING: %s was compiled WITHOUT its tuned flags %s (this hipcc rejected them: %s); benchmark numbers of this build "
                      "are not those of the tuned configuration" % (src, " ".join(EXTRA_FLAGS[src]), first_err[0][:200]), file=sys.stderr, flush=True)
                fallbacks.append(src)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    # sidecar next to the library (travels with it): which per-file flags the objects were built with.  An incremental build keeps what earlier runs recorded
    # for the objects it did not recompile
    info_path = lib + ".build_info.json"
    prev = {}
    if not force and os.path.exists(info_path):
        try:
            prev = json.load(open(info_path))
        except Exception:      # noqa: BLE001
            prev = {}
    kept = [s for s in prev.get("compiled_without_tuned_flags", []) if s in SOURCES and s not in recompiled]
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().splitlines()[:1]
    json.dump({"library": os.path.basename(lib), "dev": bool(dev), "hipcc": ver[0] if ver else None, "flags": FLAGS,
               "tuned_flags": {k: v for k, v in EXTRA_FLAGS.items()}, "compiled_without_tuned_flags": sorted(set(kept + fallbacks))}, open(info_path, "w"), indent=1)
    return lib


def build_info(lib=None):
    """what build() recorded for this library, or None (bench.py puts it into its JSON line)"""
    try:
        return json.load(open((lib or LIB) + ".build_info.json"))
    except Exception:      # noqa: BLE001
        return None


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True, dev="--dev" in sys.argv))
