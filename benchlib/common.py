"""Constants of the benchmark (BASELINE.json's metric configuration) shared by bench.py and the benchlib modules."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
# lanes in flight per frames-per-submit for the batch curve (measured: tools/pipe_probe.py; more lanes than this do not pay)
LANES_FOR = {1: 4, 2: 4, 4: 4, 8: 4, 16: 4, 32: 4}
# ... and with the cross-agent exchange beside the lanes (--gpus N > 1, and the one-GPU RCCL leg): the device runs four busy streams side by side, so two lanes (SuperPoint
# and NetVLAD stream each) leave the exchange stream a hardware pipe it shares with a NetVLAD stream only: 2445-2453 stereo frames/s per rank with the exchange against
# 2416-2425 with three or four lanes, where it takes turns with a lane's SuperPoint stream (measured over one-rank RCCL)
LANES_WITH_EXCHANGE = {16: 2, 32: 2}          # round 6 A/B over RCCL loopback (profiles/r06_exchange_placement_ab.txt): d2fe_exchange_* on ONE stream of its own beside TWO lanes
                                              # 2230-2235; beside four lanes 2200-2206; on the producing lanes' streams 2128-2151 (the lane's next pass waits for the sequence)
REFUSED_ENV = ("D2FE_ABLATE", "D2FE_MATCH_NOFALLBACK")     # switches that make results wrong or parity unproven: never inside a benchmark

H, W, CAP = 480, 640, 200
CONV1B_FLOP_PER_IMG = 2.0 * H * W * 64 * 64 * 9          # 22.65 GFLOP (SURVEY.md section 8a layer table)
SP_FLOP_PER_IMG = 52.1e9
NV_MULT = 0.75                                            # MobileNetV2 alpha = 0.75 trunk (HF-Net's width) -> NetVLAD K = 32 -> 4096
NV_FLOP_PER_IMG = 2.4780096e9                             # the stand-in MobileNetVLAD trunk at that width, 640x480: 1.239 GMAC (d2slam_amd.netvlad.arch_flops(0.75))
PEAK_TFLOPS = {"f32": 157.3, "f16x2": 2500.0, "wino": 157.3}             # MI355X_MICROARCH.md: fp32 MFMA / dense f16 MFMA
NETVLAD_GATE = 0.8                                        # track_remote_netvlad_thres stand-in (the YAMLs carry 0.5..0.8)
CPU_WARMUP = 5          # warm-up 5, >= 50 timed iterations, median + p95
