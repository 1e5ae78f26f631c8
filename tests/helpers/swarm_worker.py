"""One rank of the world-2 cross-agent test (tests/test_swarm_gpu.py launches two of these on ONE GPU with the gloo backend).
Each rank extracts its own frames with the HIP path, packs exchange blocks, all-gathers them, evaluates the NetVLAD gate and matches
its frames against the remote ones with the gate applied -- and checks every step against the CPU oracle run on the SAME seeded frames
of both ranks.  Exit code 0 = all assertions held."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from d2slam_amd import api, netvlad as nvm, swarm
    from d2slam_amd.synth import synth_stereo
    from d2slam_amd.weights import synthetic_superpoint_weights
    from oracle import oracle as orc

    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    H, W, CAP, F = 120, 160, 60, 2
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    nv = nvm.synthetic_netvlad_weights()

    def frames(r):
        # the ranks look at the same scenes (seed f) with rank-specific noise, so that cross-agent matches exist
        out = []
        for f in range(F):
            l, rr = synth_stereo(H, W, seed=40 + f)
            rng = np.random.RandomState(1000 * r + f)
            l = np.clip(l.astype(np.int32) + rng.randint(-2, 3, l.shape), 0, 255).astype(np.uint8)
            out.append((l, rr))
        return out

    mine = frames(rank)
    host = np.stack([p[0] for p in mine] + [p[1] for p in mine])           # [L0..L(F-1), R0..R(F-1)]
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2 * F, precision=api.PREC_F32))
    fe.load_superpoint(w); fe.load_netvlad(nv)
    G = fe.netvlad_dim
    BLK = api.block_words(CAP, G)
    imgs = torch.from_numpy(host).to(dev)
    n_local_rows = 3 * F * CAP
    pool = torch.zeros((n_local_rows * 256 + world * F * BLK,), dtype=torch.float32, device=dev)
    desc = pool[:n_local_rows * 256].view(3 * F, CAP, 256)
    gath = pool[n_local_rows * 256:].view(world, F, BLK)
    kps = torch.zeros((3 * F, CAP, 2), device=dev); cnt = torch.zeros(3 * F, dtype=torch.int32, device=dev)
    scores = torch.zeros((2 * F, CAP), device=dev); kidx = torch.zeros((2 * F, CAP), dtype=torch.int32, device=dev)
    gdesc = torch.zeros((F, G), device=dev); blocks = torch.zeros((F, BLK), device=dev)
    st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
    fe.netvlad_device(imgs.data_ptr(), F, W, H, gdesc.data_ptr(), stream=s)
    fe.extract_device(imgs.data_ptr(), 2 * F, W, H, kps.data_ptr(), scores.data_ptr(), desc.data_ptr(), kidx.data_ptr(), CAP, cnt.data_ptr(), stream=s)
    fe.pack_blocks_device(desc.data_ptr(), kps.data_ptr(), scores.data_ptr(), cnt.data_ptr(), gdesc.data_ptr(), 0, 1, F, CAP, G, blocks.data_ptr(), stream=s)
    swarm.all_gather_blocks(gath, blocks)

    # ---- the block of every frame of every rank against the oracle's extraction of the same frame --------------------------------
    off = {f: api.block_field_offset(CAP, G, f) for f in ("desc", "kps", "scores", "netvlad", "n")}
    gh = gath.cpu().numpy()
    ref = {}
    for r in range(world):
        for f, (l, _) in enumerate(frames(r)):
            rk, rs, rd, _, _ = orc.extract_b(l, w, 0.015, 1, CAP)
            rg = orc.netvlad_forward(l, nv)
            ref[(r, f)] = (rk, rs, rd, rg)
            b = gh[r, f]
            n = int(b.view(np.int32)[off["n"]])
            assert n == len(rk), (r, f, n, len(rk))
            assert np.array_equal(b[off["kps"]:off["kps"] + 2 * n].reshape(n, 2), rk)
            assert np.array_equal(b[off["scores"]:off["scores"] + n], rs)
            assert np.abs(b[off["desc"]:off["desc"] + 256 * n].reshape(n, 256) - rd).max() <= 1e-6
            assert not b[off["desc"] + 256 * n:off["kps"]].any(), "rows beyond n must be zero"
            assert np.abs(b[off["netvlad"]:off["netvlad"] + G] - rg).max() <= 1e-4

    # ---- NetVLAD gate + gated cross-agent matching ---------------------------------------------------------------------------------
    pl = swarm.PairList(F, CAP, world, rank, BLK)
    t = lambda x, dt: torch.tensor(x, dtype=dt, device=dev)
    a_off, b_off = t(pl.a_off, torch.int32), t(pl.b_off, torch.int32)
    a_cnt = cnt[t(pl.a_cnt_row, torch.int64)].contiguous()
    b_cnt = torch.zeros(pl.npairs, dtype=torch.int32, device=dev)
    b_cnt[:pl.n_local] = cnt[t(pl.b_cnt_row, torch.int64)]
    rem = t(pl.remote_block, torch.int64)
    b_cnt[pl.n_local:] = gath.view(torch.int32).view(world * F, BLK)[rem, off["n"]]
    # threshold between the similarities of same-scene and different-scene pairs so that both outcomes occur
    sims_ref = np.array([float(np.dot(ref[(rank, f)][3], ref[divmod(b, F)][3])) for f, b in zip(pl.remote_q_frame, pl.remote_block)])
    thres = 0.5 * (sims_ref.max() + np.sort(sims_ref)[0]) if sims_ref.max() - sims_ref.min() > 1e-3 else float(sims_ref.min()) - 1.0
    gp = torch.zeros(pl.n_remote, dtype=torch.int32, device=dev); gs = torch.zeros(pl.n_remote, device=dev); gn = torch.zeros(1, dtype=torch.int32, device=dev)
    cnt_gate = a_cnt[pl.n_local:].clone()
    gq_t, gdb_t = t(pl.remote_q_frame, torch.int32), rem.to(torch.int32)
    fe.gate_pairs_device(gdesc.data_ptr(), G, gath.data_ptr() + 4 * off["netvlad"], BLK, G, gq_t.data_ptr(),
                         gdb_t.data_ptr(), pl.n_remote, float(thres), d_cnt_inout=cnt_gate.data_ptr(), d_pass=gp.data_ptr(),
                         d_sims=gs.data_ptr(), d_n_pass=gn.data_ptr(), stream=s)
    a_cnt[pl.n_local:] = cnt_gate
    mq = torch.zeros((pl.npairs, CAP), dtype=torch.int32, device=dev); mt = torch.zeros_like(mq)
    md = torch.zeros((pl.npairs, CAP), device=dev); mn = torch.zeros(pl.npairs, dtype=torch.int32, device=dev)
    fe.match_batch_device(pool.data_ptr(), pool.data_ptr(), a_off.data_ptr(), b_off.data_ptr(), a_cnt.data_ptr(), b_cnt.data_ptr(), pl.npairs, 256, CAP,
                          mq.data_ptr(), mt.data_ptr(), md.data_ptr(), mn.data_ptr(), mode=0, ratio=0.8, radius=-1.0, stream=s)
    torch.cuda.synchronize()
    sims = gs.cpu().numpy(); passed = gp.cpu().numpy()
    assert np.abs(sims - sims_ref).max() <= 2e-4
    n_cross = 0
    for i in range(pl.n_remote):
        f, blk = pl.remote_q_frame[i], pl.remote_block[i]
        if abs(sims_ref[i] - thres) > 1e-3:
            assert bool(passed[i]) == (sims_ref[i] >= thres), (i, sims_ref[i], thres)
        p = pl.n_local + i
        n = int(mn[p].item())
        if not passed[i]:
            assert n == 0, "a pair the NetVLAD gate rejects must not be matched"
            continue
        # matches of the GPU descriptors (local frame f vs the remote block's descriptors) == the oracle's matchKNN of the same sets
        da = desc[f, :int(cnt[f].item())].cpu().numpy()
        nb = int(gh[blk // F, blk % F].view(np.int32)[off["n"]])
        db = gh[blk // F, blk % F][off["desc"]:off["desc"] + 256 * nb].reshape(nb, 256)
        rq, rt, rd = orc.match_knn(da, db, 0.8)
        assert n == len(rq), (i, n, len(rq))
        assert np.array_equal(mq[p, :n].cpu().numpy(), rq) and np.array_equal(mt[p, :n].cpu().numpy(), rt) and np.array_equal(md[p, :n].cpu().numpy(), rd)
        n_cross += n
    assert int(gn.item()) == int(passed.sum())
    assert passed.sum() >= 1 and n_cross >= 3, "the test scene must produce cross-agent matches (%d pairs pass, %d matches)" % (passed.sum(), n_cross)
    # local pairs as well
    for p in range(pl.n_local):
        ra, rb = pl.a_cnt_row[p], pl.b_cnt_row[p]
        da = desc[ra, :int(cnt[ra].item())].cpu().numpy(); db = desc[rb, :int(cnt[rb].item())].cpu().numpy()
        rq, rt, rd = orc.match_knn(da, db, 0.8)
        n = int(mn[p].item())
        assert n == len(rq) and np.array_equal(mq[p, :n].cpu().numpy(), rq) and np.array_equal(mt[p, :n].cpu().numpy(), rt)
    fe.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d OK: %d remote pairs, %d pass the NetVLAD gate, %d cross-agent matches" % (rank, pl.n_remote, int(passed.sum()), n_cross))


if __name__ == "__main__":
    main()
