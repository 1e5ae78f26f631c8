#!/bin/bash
# experimental F(4,3) x F(2,3) kernel: layer tests, then per-layer timing against F(2x2,3x3) and with parts of the kernel switched off (D2FE_ABLATE: 1 no
# epilogue, 2 no window reads / column transform, 4 no patch copies, 8 no output stores; results are wrong with a bit set)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_wino43.py -q -m gpu 2>&1 | tail -2
L=${W43_LAYERS:-conv2a,conv3b,conv4a}
echo "== F(2x2,3x3)"; timeout 300 python tools/bench_wino.py --imgs 64 --iters 10 --layers $L 2>&1 | grep -v amdgpu
for ab in ${W43_ABL:-0 1 2 4 7}; do echo "== F(4,3)xF(2,3) ablate $ab"; D2FE_WINO43=1 D2FE_ABLATE=$ab timeout 300 python tools/bench_wino.py --imgs 64 --iters 10 --layers $L 2>&1 | grep -v amdgpu; done
