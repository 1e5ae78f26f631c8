"""Generates the committed fixtures in tests/golden/.  Network outputs come from PyTorch fp64 (independent of
the oracle and of the HIP path); selection / sampling / matching outputs come from the oracle and serve as
regression pins.  Run from the repo root:  python tests/golden/make_golden.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from d2slam_amd.synth import synth_descriptor_pair, synth_image
from d2slam_amd.weights import synthetic_superpoint_weights
from oracle import oracle as orc
from test_oracle_cpu import _torch_forward

w = synthetic_superpoint_weights(dustbin_bias=7.5)
img = synth_image(64, 96, 3)
logits, draw, descn, semi = _torch_forward(img, w)
f = orc.superpoint_forward(img, w)
k, s, i = orc.select_b(f["semi"], 0.015, 1, 50)
d = orc.sample_b(f["desc"], k)
np.savez_compressed(os.path.join(HERE, "superpoint_64x96.npz"), image=img, torch_logits=logits.astype(np.float32),
                    torch_semi=semi.astype(np.float32), torch_desc=descn.astype(np.float32), sel_idx=i, sel_scores=s, sel_desc=d)
a, b, pa, pb = synth_descriptor_pair(120, 90, 256, seed=3)
q, t, dd = orc.match_knn(a, b, 0.8, pa, pb, 40.0)
np.savez_compressed(os.path.join(HERE, "match_120x90.npz"), a=a, b=b, pts_a=pa, pts_b=pb, q=q, t=t, d=dd)
print("wrote fixtures:", len(i), "keypoints,", len(q), "matches")
