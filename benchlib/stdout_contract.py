"""The stdout contract of bench.py: ONE JSON line, the LAST thing on stdout, on every rank, small enough to survive a tail."""
import json
import os
import sys

from .common import ROOT

# ---- the stdout contract: ONE JSON line, the LAST thing on stdout, on every rank, small enough to survive a tail ------------------------------------------------------
# RCCL prints its version banner through C stdio, which on a pipe or file is flushed at process exit -- behind anything Python printed (round 5's line was lost to that).
# So (i) fd 1 is pointed at stderr for the whole run (the banner, torch, rocprofv3 children, stray prints land there), the JSON line goes to the saved descriptor;
# (ii) before the line is written every C stream is flushed (a driver that merges stderr into stdout still sees the banner BEFORE the line); (iii) after the line fds 1 and 2
# of this process go to /dev/null: nothing can follow it.  Ranks other than 0 never own stdout at all.
_REAL_STDOUT = None
LINE_BUDGET = 6000                  # bytes; the driver keeps an 8 KB tail
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                 "roofline", "cpu_baseline", "step_roofline", "parity", "roofline_netvlad", "rccl", "exchange", "netvlad_gate", "cross_agent", "extras")
# dropped from the line (kept in the extras file) in this order while the line is above LINE_BUDGET
SHED_ORDER = ("cross_agent", "netvlad_gate", "exchange", "roofline_netvlad", "rccl")


def claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def silence_this_process():
    """fds 1 and 2 -> /dev/null (after flushing every C and Python stream): whatever this process still prints (exit-time banners, teardown warnings) goes nowhere"""
    import ctypes
    try:
        sys.stdout.flush(); sys.stderr.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    nul = os.open(os.devnull, os.O_WRONLY)
    os.dup2(nul, 1); os.dup2(nul, 2)


def emit_line(obj):
    """the JSON line: everything buffered so far is flushed first, the line is written to the process's ORIGINAL stdout in one piece, then the process goes silent"""
    global _REAL_STDOUT
    import ctypes
    data = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush(); sys.stderr.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    fd = _REAL_STDOUT if _REAL_STDOUT is not None else 1
    while data:
        n = os.write(fd, data)
        data = data[n:]
    silence_this_process()
    if _REAL_STDOUT is not None:
        os.close(_REAL_STDOUT)
        _REAL_STDOUT = None


def slim(o, maxlen=150, keep=("workload", "sample", "kernel", "api")):
    """the headline form of a (nested) record: prose longer than `maxlen` characters lives in the extras file; the strings a reader needs to identify the
    workload stay, cut to 320 characters"""
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if isinstance(v, str) and len(v) > maxlen:
                if k in keep:
                    out[k] = v if len(v) <= 320 else v[:317] + "..."
                continue
            out[k] = slim(v, maxlen, keep)
        return out
    if isinstance(o, list):
        return [slim(v, maxlen, keep) for v in o]
    return o


def extras_path():
    p = os.environ.get("D2FE_BENCH_EXTRAS")
    if p:
        return p
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, "bench_extras.json")
    except OSError:
        import tempfile
        return os.path.join(tempfile.gettempdir(), "d2fe_bench_extras.json")


def headline(full):
    """(line, path): the full record goes to the extras file (named in the line), the line keeps HEADLINE_KEYS in slim form and fits LINE_BUDGET"""
    path = extras_path()
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        rel = os.path.relpath(path, ROOT)
        named = rel if not rel.startswith("..") else path
    except OSError as e:
        named = "not written: %s" % str(e)[:80]
    always = ("vs_baseline", "cpu_baseline", "roofline")          # present (null when this run has none) in every line
    line = {k: slim(full.get(k)) for k in HEADLINE_KEYS if k in always or full.get(k) is not None}
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):          # the per-stage tables and the fmaf-chain extra stay in the file
        cb = {k: ({a: b for a, b in v.items() if a != "ms_per_stereo_frame"} if isinstance(v, dict) else v) for k, v in cb.items() if k != "fmaf_oracle"}
        line["cpu_baseline"] = dict(cb, per_stage="extras file: cpu_baseline.{all_cores,single_thread}.ms_per_stereo_frame")
    line["extras"] = {"file": named, "keys": sorted(k for k in full if k not in line)}
    if isinstance(line.get("rccl"), dict) and len(line["rccl"].get("ranks") or []) > 2:
        line["rccl"] = dict(line["rccl"], ranks="%d entries in the extras file" % len(line["rccl"]["ranks"]))
    for k in SHED_ORDER:
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        if k in line:
            del line[k]
            line["extras"]["keys"] = sorted(line["extras"]["keys"] + [k])
    if len(json.dumps(line)) > LINE_BUDGET:
        line["extras"]["keys"] = "see the file"
    return line, path


def real_stdout():
    """the process's ORIGINAL stdout descriptor once claim_stdout() has run (None before)"""
    return _REAL_STDOUT
