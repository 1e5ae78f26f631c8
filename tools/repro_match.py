import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d2slam_amd import api
from oracle import oracle as orc
orc.build()
fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=100, input_width=64, input_height=64, max_batch=1))
rng = np.random.RandomState(3)
def unit(x): return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
for name, n, dup in (("self", 100, 0), ("self_dups", 100, 30), ("self62", 62, 0), ("self62_dups", 62, 20)):
    a = unit(rng.randn(n, 256))
    if dup:
        a[rng.choice(n, dup, replace=False)] = a[0]
    b = a.copy()
    fe.match_fallback_rows(reset=True)
    q, t, d = fe.match_knn(a, b, 0.8)
    rq, rt, rd = orc.match_knn(a, b, 0.8)
    print(name, "host:", len(q), "oracle:", len(rq), "equal:", np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d, rd), "stats", fe.match_fallback_rows(full=True))
    # batched device form, two pairs
    import torch
    dev = torch.device("cuda", 0)
    pool = torch.from_numpy(np.concatenate([a, b, a, b])).to(dev)
    a_off = torch.tensor([0, 2 * n], dtype=torch.int32, device=dev); b_off = torch.tensor([n, 3 * n], dtype=torch.int32, device=dev)
    cnt = torch.full((2,), n, dtype=torch.int32, device=dev)
    mq = torch.zeros((2, 100), dtype=torch.int32, device=dev); mt = torch.zeros_like(mq); md = torch.zeros((2, 100), dtype=torch.float32, device=dev); mn = torch.zeros(2, dtype=torch.int32, device=dev)
    fe.match_batch_device(pool.data_ptr(), pool.data_ptr(), a_off.data_ptr(), b_off.data_ptr(), cnt.data_ptr(), cnt.data_ptr(), 2, 256, 100, mq.data_ptr(), mt.data_ptr(), md.data_ptr(), mn.data_ptr())
    torch.cuda.synchronize()
    print("   batch:", mn.cpu().numpy(), "first pair equal:", np.array_equal(mq[0, :int(mn[0])].cpu().numpy(), rq))
