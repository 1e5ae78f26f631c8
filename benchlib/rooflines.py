"""Roofline objects of the bench line: analytic FLOP counts, HIP-event durations, PMC traffic."""
import json
import os
import sys
import time

from .common import *  # noqa: F401,F403


# executed matrix-pipe FLOPs of ONE 640x480 image through SuperPoint (sparse descriptor head at <= 4 corner cells per keypoint, 200 keypoints): the layer table of
# SURVEY.md section 8(a) with the arithmetic each mode runs.  GFLOP: conv1a 0.354 (as staged inside conv1b's kernel: 60 MFMAs per 8x16 item = 0.590), conv1b 22.65,
# conv2a 5.66, conv2b 5.66, conv3a 2.83, conv3b 5.66, conv4a 1.416, conv4b 1.416, convPa 2.831, convPb 0.160, descriptor head at the selected cells 0.28 (18 GFLOP per
# 64 images; the dense convDa + convDb would be 3.46)
_SP_WINO_LAYERS_GF = 22.65 + 5.66 + 5.66 + 2.83 + 5.66 + 1.416 + 1.416 + 2.831       # the eight 3x3 layers the Winograd mode runs as F(2x2,3x3): 16/36 of these
_SP_OTHER_GF = 0.160 + 0.28


def sp_executed_gflop_per_image(precision):
    if precision == "wino":
        return 0.590 + _SP_WINO_LAYERS_GF * 16.0 / 36.0 + _SP_OTHER_GF
    direct = 0.354 + _SP_WINO_LAYERS_GF + _SP_OTHER_GF
    return direct * (3.0 if precision == "f16x2" else 1.0)


def step_roofline(precision, F, ms_per_step, netvlad, npairs):
    """The WHOLE step against the matrix pipe: executed MFMA FLOPs of everything a step launches / ms_per_step / peak.  Analytic counts (the layer
    table above; NetVLAD 2.478 GFLOP per left image (MobileNetV2-0.75 trunk) on the fp32 pipe; matchKNN 2 strips x 2 na nb 256 per pair); the per-kernel SQ_INSTS_MFMA sums of the committed
    profile of the same command agree (profiles/: 3.46e8 x 4096 = 1.42 TFLOP per 64-image step in Winograd mode)."""
    peak = PEAK_TFLOPS[precision]
    sp = sp_executed_gflop_per_image(precision) * 1e9 * 2 * F
    nv = NV_FLOP_PER_IMG * F if netvlad else 0.0
    mt = 2 * 2.0 * CAP * CAP * 256 * npairs
    # NetVLAD and the matcher run on the fp32 pipe in every mode; in f16x2 mode their FLOPs are priced at the fp32 peak separately
    if precision == "f16x2":
        frac = (sp / (peak * 1e12) + (nv + mt) / (PEAK_TFLOPS["f32"] * 1e12)) / (ms_per_step * 1e-3)
    else:
        frac = (sp + nv + mt) / (peak * 1e12) / (ms_per_step * 1e-3)
    return {"bound": "mfma", "executed_mfma_tflop_per_step": round((sp + nv + mt) / 1e12, 4), "superpoint": round(sp / 1e12, 4), "netvlad": round(nv / 1e12, 4), "matcher": round(mt / 1e12, 4),
            "ms_per_step": round(ms_per_step, 3), "achieved": round((sp + nv + mt) / (ms_per_step * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(frac, 4),
            "algorithmic_tflop_per_step": round((SP_FLOP_PER_IMG * 2 * F + nv + mt / 2) / 1e12, 4),
            "note": "executed matrix-pipe FLOPs of every launch of a step / wall time of the step (H2D, D2H, post-processing and launch gaps included) / peak"
                    + ("; f16x2: SuperPoint's 3 MFMA FLOPs per algorithmic FLOP against the f16 peak, NetVLAD + matcher against the fp32 peak" if precision == "f16x2" else "")}


def flag_above_peak(r):
    """a roofline object whose ALGORITHMIC fraction exceeds 1 says why, next to the number"""
    if r and (r.get("frac_algorithmic") or 0) > 1.0:
        r["algorithmic_above_peak"] = "Winograd F(2x2,3x3): 16/36 of the direct convolution's multiplies are executed; `frac` (= frac_executed) is the matrix pipe's fraction"
    return r


def build_record():
    """what d2slam_amd.build recorded for the library this run timed (a build that fell back to untuned flags must be visible in the line)"""
    try:
        from d2slam_amd import build as hb
        bi = hb.build_info() or {}
        return {"hipcc": bi.get("hipcc"), "tuned_flags": bi.get("tuned_flags"), "compiled_without_tuned_flags": bi.get("compiled_without_tuned_flags"),
                "recorded": bool(bi)}
    except Exception:      # noqa: BLE001
        return {"recorded": False}


def netvlad_roofline(t_ms, F, how, flop_per_img=NV_FLOP_PER_IMG):
    ach = flop_per_img * F / (t_ms * 1e-3) / 1e12
    return {"kernel": "NetVLAD launch sequence (one launch per MobileNetV2 block: nv_fpair_kernel, nv_pblock_kernel (stride 1), nv_xblock_kernel (stride 2), nv_slab_sum_kernel, nv_tail_kernel, "
                      "nv_vlad_* x2; d2slam_amd/csrc/netvlad*.hip): MobileNetV2-%.2f trunk + NetVLAD head" % (NV_MULT if abs(flop_per_img - NV_FLOP_PER_IMG) < 1 else -1),
            "bound": "mfma", "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 4), "traffic": None,
            "ms_per_call": round(t_ms, 4), "images_per_call": F, "algorithmic_flop_per_call": flop_per_img * F,
            "note": "fp32 MFMA (v_mfma_f32_16x16x4_f32) + VALU depthwise; instruction/latency-bound small layers (DESIGN.md section 4); " + how}


def index_parity_evidence():
    """The committed run of tools/mode_disagreement.py (profiles/r04_mode_disagreement.json): keypoint / match index differences of the Winograd and
    fp16 hi/lo modes against the exact fp32 mode over 1056 images (992 synthetic + 64 derived from the real crops of the reference's sample image) at
    N = 100 / 150 / 200 and thresholds 0.015 / 0.15.  Summarised here as the worst rate over the six configurations; the bench frames of THIS run are
    compared live in `wino_vs_exact_on_bench_frames` / `f16x2_vs_exact_on_bench_frames`."""
    name = next((n for n in ("r05_mode_disagreement.json", "r04_mode_disagreement.json") if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
    if name is None:
        return None
    j = json.load(open(os.path.join(ROOT, "profiles", name)))
    out = {"source": "profiles/" + name + " (python tools/mode_disagreement.py on MI355X, same kernels; the FULL study is not collected inside this run -- its 128-image subset is: "
                     "`index_parity_in_run`)",
           "images": j["images"], "real_derived_images": j["real_derived_images"], "pairs": j["pairs"], "configs": "N in {100, 150, 200} x threshold in {0.015, 0.15}"}
    for m in ("wino", "f16x2"):
        rows = [c["%s_vs_f32_all" % m] for c in j["configs"]]
        out[m + "_vs_f32"] = {"keypoints_compared": sum(r["keypoints"] for r in rows), "keypoints_in_one_mode_only": sum(r["keypoints_in_one_mode_only"] for r in rows),
                              "worst_per_1e4_keypoints": max(r["per_1e4_keypoints"] for r in rows),
                              "matches_compared": sum(r["matches"] for r in rows), "matches_in_one_mode_only": sum(r["matches_in_one_mode_only"] for r in rows),
                              "worst_per_1e4_matches": max(r["per_1e4_matches"] for r in rows)}
    return out


def profiled_traffic(kernel_tag):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC passes of this command (profiles/r05_wino_rocprofv3_summary.txt, else round 4's:
    separate --pmc FETCH_SIZE and WRITE_SIZE passes, tools/profile.sh; KiB per dispatch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950's wide reads).  bench.py itself does not collect counters: null when the file is absent."""
    name = next((n for n in ("r06_wino_rocprofv3_summary.txt", "r05_wino_rocprofv3_summary.txt", "r04_wino_rocprofv3_summary.txt") if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
    if name is None:
        return None, None
    path = os.path.join(ROOT, "profiles", name)
    fetch = write = None
    lines = open(path).read().split("\n")
    sect = ""
    for i, l in enumerate(lines):
        if l.startswith("== "):
            sect = l
        if kernel_tag in l and i + 1 < len(lines):
            nxt = lines[i + 1]
            if "pmc_fetch" in sect and "FETCH_SIZE=" in nxt:
                fetch = float(nxt.split("FETCH_SIZE=")[1].split()[0])
            if "pmc_write" in sect and "WRITE_SIZE=" in nxt:
                write = float(nxt.split("WRITE_SIZE=")[1].split()[0])
    if fetch is None or write is None:
        return None, None
    return int((2.0 * fetch + write) * 1024), "profiles/" + name + ": 2 x FETCH_SIZE %.4g KiB + WRITE_SIZE %.4g KiB per dispatch (rocprofv3 --pmc passes of `python bench.py`, same build; not collected inside this run)" % (fetch, write)


def live_traffic(kernel_tag):
    """roofline.traffic measured by THIS run: two child passes of this script under `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE, then WRITE_SIZE --
    separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; KiB per dispatch; FETCH_SIZE doubled for gfx950's wide reads), the headline step with ONE
    submit in flight and nothing else (--single-mode), averaged over the dispatches of the dominant kernel.  None when rocprofv3 is not there or a pass fails
    (the committed profile is then quoted, see profiled_traffic)."""
    import csv, glob, shutil, subprocess, tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp) or os.environ.get("D2FE_BENCH_CHILD"):
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None              # this run is itself being profiled: no nested profiler
    vals, t0 = {}, time.time()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="d2fe_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, D2FE_BENCH_CHILD="1", TMPDIR="/tmp")
            cmd = [rp, "--output-format", "csv", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "3", "--warmup", "1", "--precision", "wino", "--single-mode", "--no-cpu-baseline", "--lanes", "1"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not fs:
                return None
            acc = n = 0
            for row in csv.DictReader(open(fs[0])):
                if kernel_tag in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    acc += float(row["Counter_Value"]); n += 1
            if not n:
                return None
            vals[ctr] = (acc / n, n)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    return {"traffic": int((2.0 * fetch + write) * 1024),
            "counters": {"FETCH_SIZE_KiB_per_dispatch": round(fetch, 1), "WRITE_SIZE_KiB_per_dispatch": round(write, 1), "dispatches_averaged": [vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]],
                         "seconds": round(time.time() - t0, 1)},
            "note": "measured by this run: two child passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --single-mode --lanes 1 --steps 3` (same build, same "
                    "step, one submit in flight); traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch of the dominant kernel (the doubling: gfx950's 128-byte reads, MI355X_MICROARCH.md)"}


def conv1b_roofline(precision, avg_ms, launches, NI, fused):
    peak = PEAK_TFLOPS[precision]
    alg = CONV1B_FLOP_PER_IMG * NI
    if precision == "wino":
        # Winograd F(2x2,3x3): 16 multiply-adds per output and channel pair where the direct convolution has 36.  The launch is the
        # conv1a-fused kernel (D2FE_FUSE1A default): per 8x16-pixel work item 1024 conv1b MFMAs + 60 conv1a MFMAs (v_mfma_f32_32x32x2_f32).
        items = NI * (H // 8) * (W // 16)
        executed = items * (1024 + (60 if fused is not False else 0)) * 4096.0
        ach_e = executed / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        ach_a = alg / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic, tnote = profiled_traffic("conv_wino_kernel<64, true, true, 0, 1, true") if (fused is not False and NI == 64) else (None, None)
        return {"kernel": "conv_wino_kernel<64,POOL,RELU,FUSE> (conv1a from the u8 frame fused into conv1b as Winograd F(2x2,3x3), + ReLU + 2x2 max-pool)",
                "bound": "mfma", "achieved": round(ach_e, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach_e / peak, 4),
                "frac_executed": round(ach_e / peak, 4), "frac_algorithmic": round(ach_a / peak, 4),
                "achieved_algorithmic": round(ach_a, 2),
                "traffic": traffic, "traffic_note": tnote or "HBM bytes per launch are in profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes); not collected inside this run",
                "compulsory_bytes_per_launch": int(NI * (H * W + (H // 2) * (W // 2) * 64 * 4)),
                "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                "algorithmic_flop_per_launch": alg, "executed_mfma_flop_per_launch": executed,
                "note": "frac = frac_executed = MFMA FLOPs the kernel executes / HIP-event time / 157.3 TF (the matrix pipe's roofline fraction); "
                        "frac_algorithmic = SURVEY section 8(d)'s direct-convolution FLOPs / time / peak, above 1 because Winograd executes 16/36 of them"}
    ach = alg / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    ex = 3.0 if precision == "f16x2" else 1.0
    return {"kernel": "conv_%s_kernel<64,3,4,32,2,2,2,1,POOL,RELU,FUSE1A> (conv1a fused into conv1b)" % ("f32" if precision == "f32" else "f16x2"),
            "bound": "mfma", "achieved": round(ach * ex, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach * ex / peak, 4),
            "frac_executed": round(ach * ex / peak, 4), "frac_algorithmic": round(ach / peak, 4), "achieved_algorithmic": round(ach, 2),
            "traffic": None, "traffic_note": "profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): 22.2 MB per image, 19.7 MB of it the pooled output",
            "avg_launch_ms": round(avg_ms, 4), "launches": launches, "algorithmic_flop_per_launch": alg,
            "executed_mfma_flop_per_launch": alg * ex,
            "note": "f16x2 executes 3 MFMA FLOPs (hi*hi + hi*lo + lo*hi) per algorithmic FLOP" if precision == "f16x2" else "one MFMA FLOP per algorithmic FLOP"}
