"""Host-side weight packing of the NetVLAD block kernels (d2slam_amd/csrc/netvlad_pair.hip, netvlad_fused.hip) -- no GPU needed.

Each kernel reads its weights as MFMA fragments of v_mfma_f32_16x16x4_f32 (B[k = lane >> 4][n = lane & 15]) from per-chunk records the host packs;
which input / hidden channel a (k-step, lane group) pair means differs per kernel (that is what lets the kernels load 2 / 4 consecutive channels per lane, or
chain two GEMMs without a transpose).  The tests re-derive every record from the kernels' READ patterns and check that the GEMMs they feed are the plain
matrix products (reference boundary: MobileNetVLADONNX::inference, d2frontend/include/d2frontend/CNN/mobilenetvlad_onnx.h:49-74)."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from d2slam_amd import build
    l = C.CDLL(build.build(dev=True))      # the packing hooks are test hooks of the development library (include/d2fe_debug.h)
    l.d2fe_debug_pack_netvlad.restype = C.c_long
    return l


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _pack(lib, kind, cin, chid, cout, we=None, be=None, wd=None, bd=None, wp=None):
    out = np.full(1 << 20, np.nan, np.float32)
    n = lib.d2fe_debug_pack_netvlad(kind, _p(we), _p(be), _p(wd), _p(bd), _p(wp), cin, chid, cout, _p(out), C.c_long(out.size))
    assert n > 0, n
    assert not np.isnan(out[:n]).any()
    return out[:n]


def _weights(cin, chid, cout, seed):
    r = np.random.default_rng(seed)
    return (r.standard_normal((chid, cin)).astype(np.float32), r.standard_normal(chid).astype(np.float32),
            r.standard_normal((chid, 9)).astype(np.float32), r.standard_normal(chid).astype(np.float32),
            r.standard_normal((cout, chid)).astype(np.float32))


def _ntiles(cout):
    nt = (cout + 15) // 16
    return nt if nt <= 2 else 4 if nt <= 4 else 8


LANES = np.arange(64)
N, LQ = LANES & 15, LANES >> 4


@pytest.mark.parametrize("cin,chid,cout", [(8, 48, 8), (16, 96, 16), (24, 144, 32), (32, 192, 32), (56, 336, 56), (56, 336, 112),
                                           (24, 144, 24), (48, 288, 48), (48, 288, 72), (72, 432, 72), (120, 720, 120), (120, 720, 96)])
def test_pixel_pair_kernel_records(lib, cin, chid, cout):
    """nv_pblock_kernel: lane group lq owns input channels lq*nk .. (k-step s -> channel lq*nk + s); hidden channel of depthwise / project k-step ks for
    lane group lq: ks + 4 (lq >> 1) + 8 (lq & 1).  Project fragments of a k-step (nvp_project): nt // 4 groups of four n-tiles as [lane][4], then the remaining
    1 / 2 / 3 n-tiles as [lane][1], [lane][2] or [lane][4] with the last float unused -- the MobileNetV2-0.75 widths (48, 72, 120) have 3 / 5 / 8 n-tiles."""
    we, be, wd, bd, wp = _weights(cin, chid, cout, 1)
    nk, nt = cin // 4, (cout + 15) // 16
    ng4, rem = nt // 4, nt % 4
    remw = {0: 0, 1: 1, 2: 2, 3: 4}[rem]
    ksf = 64 * (4 * ng4 + remw)
    rec_e = (nk * 64 + 16 + 255) // 256 * 256
    pe = _pack(lib, 0, cin, chid, cout, we=we, be=be).reshape(chid // 16, rec_e)
    rec_d = (256 + 4 * ksf + 255) // 256 * 256
    pd = _pack(lib, 1, cin, chid, cout, wd=wd, bd=bd, wp=wp).reshape(chid // 16, rec_d)
    x = np.random.default_rng(2).standard_normal(cin).astype(np.float64)
    d = np.random.default_rng(3).standard_normal(chid).astype(np.float64)
    for ch in range(chid // 16):
        # expand fragments as the kernel reads them: float4 groups [s / 4][lane][4], a float2 remainder [lane][2], then bias[16]
        frag = np.zeros((nk, 64))
        for s in range(nk):
            frag[s] = pe[ch, ((s >> 2) * 64 + LANES) * 4 + (s & 3)] if s < 4 * (nk // 4) else pe[ch, (nk // 4) * 256 + LANES * 2 + (s - 4 * (nk // 4))]
            assert np.array_equal(frag[s], we[ch * 16 + N, LQ * nk + s])
        assert np.array_equal(pe[ch, nk * 64:nk * 64 + 16], be[ch * 16:ch * 16 + 16])
        assert not pe[ch, nk * 64 + 16:].any()
        # the GEMM those fragments feed: out[n] = sum over k-steps and lane groups of B[k = lq][n] * A[k = lq]
        out = np.zeros(16)
        for s in range(nk):
            np.add.at(out, N, frag[s].astype(np.float64) * x[LQ * nk + s])
        assert np.allclose(out, we[ch * 16:ch * 16 + 16].astype(np.float64) @ x, rtol=1e-12, atol=1e-12)
        # depthwise weights [lq][ks][12] = 9 taps, bias, 0, 0
        for lq in range(4):
            for ks in range(4):
                c = ch * 16 + ks + 4 * (lq >> 1) + 8 * (lq & 1)
                blk = pd[ch, (lq * 4 + ks) * 12:(lq * 4 + ks) * 12 + 12]
                assert np.array_equal(blk[:9], wd[c]) and blk[9] == bd[c] and not blk[10:].any()
        assert not pd[ch, 192:256].any()
        # project fragments as nvp_project reads them; every hidden channel of the chunk is used exactly once
        proj = np.zeros(nt * 16)
        seen = set()
        for ks in range(4):
            cmap = ch * 16 + ks + 4 * (LQ >> 1) + 8 * (LQ & 1)
            seen.update(cmap.tolist())
            base = 256 + ks * ksf
            for t in range(nt):
                hf, e = t // 4, t % 4
                frag_p = pd[ch, base + (hf * 64 + LANES) * 4 + e] if hf < ng4 else pd[ch, base + ng4 * 256 + LANES * remw + e]
                co = t * 16 + N
                ref = np.where(co < cout, wp[np.minimum(co, cout - 1), cmap], 0.0)
                assert np.array_equal(frag_p, ref.astype(np.float32))
                np.add.at(proj, co, frag_p.astype(np.float64) * d[cmap])
            if rem == 3:
                assert not pd[ch, base + ng4 * 256 + LANES * 4 + 3].any()
        assert seen == set(range(ch * 16, ch * 16 + 16))
        assert np.allclose(proj[:cout], wp[:, ch * 16:ch * 16 + 16].astype(np.float64) @ d[ch * 16:ch * 16 + 16], rtol=1e-12, atol=1e-12)
        assert not proj[cout:].any()


@pytest.mark.parametrize("cin,chid,cout", [(8, 48, 16), (16, 96, 24), (32, 192, 56), (56, 336, 56), (112, 672, 112)])
def test_per_pixel_kernel_records(lib, cin, chid, cout):
    """nv_xblock_kernel (stride-2 blocks): a float4 per lane hands it 4 consecutive input channels, so k-step (j, e) means channel (lq + 4 j) * 4 + e (Cin = 8:
    a float2, channel 2 lq + e); hidden channel of k-step ks: 4 ks + lq; depthwise weights as [lq][kp][tap][2] + bias[2] for the channel pairs of v_pk_fma_f32."""
    we, be, wd, bd, wp = _weights(cin, chid, cout, 4)
    nt = _ntiles(cout)
    nj = 0 if cin == 8 else (cin + 15) // 16
    nj = nj if nj <= 2 else 4 if nj <= 4 else 7
    ks_n = nj * 4 if nj else 2
    rec_e = ((ks_n + 1) * 64 + 255) // 256 * 256
    pe = _pack(lib, 2, cin, chid, cout, we=we, be=be).reshape(chid // 16, rec_e)
    pd = _pack(lib, 3, cin, chid, cout, wd=wd, bd=bd, wp=wp).reshape(chid // 16, 256 + nt * 256)
    x = np.random.default_rng(5).standard_normal(cin).astype(np.float64)
    for ch in range(chid // 16):
        out = np.zeros(16)
        for k in range(ks_n):
            chan = LQ * 2 + k if nj == 0 else (LQ + 4 * (k >> 2)) * 4 + (k & 3)
            ref = np.where(chan < cin, we[ch * 16 + N, np.minimum(chan, cin - 1)], 0.0).astype(np.float32)
            assert np.array_equal(pe[ch, k * 64:k * 64 + 64], ref)
            np.add.at(out, N, ref.astype(np.float64) * np.where(chan < cin, x[np.minimum(chan, cin - 1)], 0.0))
        assert np.allclose(out, we[ch * 16:ch * 16 + 16].astype(np.float64) @ x, rtol=1e-12, atol=1e-12)
        assert np.array_equal(pe[ch, ks_n * 64:ks_n * 64 + 16], be[ch * 16:ch * 16 + 16])      # bias k-step: row k = 0
        assert not pe[ch, ks_n * 64 + 16:].any()
        for lq in range(4):
            for kp in range(2):
                blk = pd[ch, (lq * 2 + kp) * 20:(lq * 2 + kp) * 20 + 20]
                for h in range(2):
                    c = ch * 16 + kp * 8 + lq + 4 * h
                    assert np.array_equal(blk[h:18:2], wd[c]) and blk[18 + h] == bd[c]
        assert not pd[ch, 160:256].any()
        for ks in range(4):
            for t in range(nt):
                co = t * 16 + N
                ref = np.where(co < cout, wp[np.minimum(co, cout - 1), ch * 16 + ks * 4 + LQ], 0.0).astype(np.float32)
                assert np.array_equal(pd[ch, 256 + (ks * nt + t) * 64:256 + (ks * nt + t) * 64 + 64], ref)


def test_tail_kernel_records(lib):
    """nv_tail_kernel (last 1x1 chained with the NetVLAD pre-projection): the expand GEMM is evaluated transposed, so lane group lq ends up holding
    hidden channels 4 lq + r of the chunk and the project GEMM's k-step ks takes the ks-th of them."""
    cin, chid, cout = 112, 1280, 128
    we, be, wd, bd, wp = _weights(cin, chid, cout, 6)
    rec_e = ((cin // 4 + 1) * 64 + 255) // 256 * 256
    pe = _pack(lib, 4, cin, chid, cout, we=we, be=be).reshape(chid // 16, rec_e)
    pp = _pack(lib, 5, cin, chid, cout, wp=wp).reshape(chid // 16, 256 + 8 * 256)
    d = np.random.default_rng(7).standard_normal(chid).astype(np.float64)
    for ch in (0, 1, 37, 79):
        for k in range(cin // 4):
            assert np.array_equal(pe[ch, k * 64:k * 64 + 64], we[ch * 16 + N, (LQ + 4 * (k >> 2)) * 4 + (k & 3)])
        assert np.array_equal(pe[ch, (cin // 4) * 64:(cin // 4) * 64 + 16], be[ch * 16:ch * 16 + 16])
        assert not pp[ch, :256].any()
        proj = np.zeros(128)
        for ks in range(4):
            hid = ch * 16 + LQ * 4 + ks
            for t in range(8):
                frag = pp[ch, 256 + (ks * 8 + t) * 64:256 + (ks * 8 + t) * 64 + 64]
                assert np.array_equal(frag, wp[t * 16 + N, hid])
                np.add.at(proj, t * 16 + N, frag.astype(np.float64) * d[hid])
        assert np.allclose(proj, wp[:, ch * 16:ch * 16 + 16].astype(np.float64) @ d[ch * 16:ch * 16 + 16], rtol=1e-12, atol=1e-12)


def test_unsupported_shapes_are_refused(lib):
    out = np.zeros(16, np.float32)
    w = np.zeros((48, 8), np.float32)
    assert lib.d2fe_debug_pack_netvlad(0, _p(w), _p(w), None, None, None, 8, 48, 8, _p(out), C.c_long(out.size)) < 0       # destination too small
    assert lib.d2fe_debug_pack_netvlad(0, _p(w), _p(w), None, None, None, 12, 48, 8, _p(out), C.c_long(out.size)) < 0      # cin not a multiple of 8
    assert lib.d2fe_debug_pack_netvlad(9, _p(w), _p(w), None, None, None, 8, 48, 8, _p(out), C.c_long(out.size)) < 0
    assert lib.d2fe_debug_pack_netvlad(4, _p(w), _p(w), None, None, None, 8, 48, 8, _p(out), C.c_long(out.size)) < 0       # not a tail shape


def test_tile_shapes_respect_the_kernels_limits(lib):
    """The launchers' tile choice for every output map from 1 x 1 to 130 x 170 (and the shipped 240 x 320 / 200 x 400 ones): the pixel-pair kernels need
    an even width, <= 64 pairs and a patch of <= 192 pixels (first block: the u8 receptive field of the patch inside its 24 x 40 byte LDS copy);
    nv_xblock_kernel <= 128 outputs and a patch of <= 192 (stride 1) / 576 (stride 2) pixels.  A launcher refusing its own tile is an error at run time."""
    th, tw = C.c_int(), C.c_int()
    sizes = [(h, w) for h in list(range(1, 41)) + [60, 75, 100, 120, 130] for w in list(range(1, 41)) + [50, 80, 100, 125, 160, 170]] + [(240, 320), (200, 400)]
    for ho, wo in sizes:
        assert lib.d2fe_debug_netvlad_tile(0, ho, wo, 1, C.byref(th), C.byref(tw)) == 0
        assert tw.value % 2 == 0 and tw.value >= 2 and th.value >= 1
        assert th.value * tw.value <= 128 and (th.value + 2) * (tw.value + 2) <= 192
        for cs in (1, 2):
            assert lib.d2fe_debug_netvlad_tile(1, ho, wo, cs, C.byref(th), C.byref(tw)) == 0
            assert tw.value % 2 == 0 and th.value * tw.value <= 128 and (th.value + 2) * (tw.value + 2) <= 192
            assert (th.value + 1) * cs + 3 <= 24 and (tw.value + 1) * cs + 3 <= 40
        for s_ in (1, 2):
            assert lib.d2fe_debug_netvlad_tile(2, ho, wo, s_, C.byref(th), C.byref(tw)) == 0
            assert th.value * tw.value <= 128
            assert ((th.value - 1) * s_ + 3) * ((tw.value - 1) * s_ + 3) <= (192 if s_ == 1 else 576)
    assert lib.d2fe_debug_netvlad_tile(7, 10, 10, 1, C.byref(th), C.byref(tw)) < 0
