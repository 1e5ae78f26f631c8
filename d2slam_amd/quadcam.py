"""BASELINE configs[2] on the device: the quadcam (FOURCORNER_FISHEYE) per-frame chain of the reference's front end.

  4 raw fisheye frames -> FisheyeUndist (remap + photometric gain, d2common/include/d2common/fisheye_undistort.h:152-176)
  -> SuperPoint + NetVLAD on every undistorted view (loop_cam.cpp:589-648)
  -> neighbour matching (0,1) (1,2) (2,3) LEFT_RIGHT and (0,3) RIGHT_LEFT (d2featuretracker.cpp:121-133) the way
     D2FeatureTracker::matchLocalFeatures does it for those types (:1144-1182): getFeatureHalfImg on both views, the a-side x
     shifted by +-move_cols, matchKNN with the search radius search_local_max_dist_lr * W_u (:669), indices mapped back
  -> temporal matchKNN of every view against the same view of the previous frame (track(), :403-456).

Everything between the raw frames and the match lists stays on the device; this module only sequences C-ABI calls
(torch is used for buffers).  The same chain on the host, with the oracle / the reference's own C++, is tests/test_quadcam_chain.py.
"""
import numpy as np

NEIGHBOURS = [(0, 1, 1), (1, 2, 1), (2, 3, 1), (0, 3, 2)]     # (view a, view b, type): 1 LEFT_RIGHT_IMG_MATCH, 2 RIGHT_LEFT_IMG_MATCH


def synthetic_maps(c, RH, RW, UH, UW):
    """Seeded cylinder-like undistortion maps + vignetting gain of virtual camera c (bench / tests; real maps: d2fe_gen_cylinder_map)."""
    yy, xx = np.mgrid[0:UH, 0:UW].astype(np.float32)
    mx = (xx / UW * (RW - 80) + 40 + 12 * np.sin(yy / 60.0 + c)).astype(np.float32)
    my = (yy / UH * (RH - 60) + 30 + 10 * np.cos(xx / 90.0 + c)).astype(np.float32)
    g = (1.0 + 0.4 * ((xx - UW / 2) ** 2 + (yy - UH / 2) ** 2) / (UW * UW / 4)).astype(np.float32)
    return mx, my, g


class QuadcamChain:
    """Device buffers + the launch sequence of one step over Q quad frames.  View order inside a step is camera-major:
    image row c*Q + q is camera c of quad frame q (one undistort launch per camera writes a contiguous slab)."""

    def __init__(self, fe, torch, dev, Q, UH, UW, cap, undistort_fov=200.0, knn_ratio=0.8, search_local_max_dist=0.2):
        self.fe, self.torch, self.Q, self.UH, self.UW, self.cap = fe, torch, Q, UH, UW, cap
        self.fov, self.ratio = undistort_fov, knn_ratio
        self.radius = search_local_max_dist * UW
        NI = 4 * Q
        self.NI = NI
        f32, i32 = torch.float32, torch.int32
        self.und = torch.zeros((NI, UH, UW), dtype=torch.uint8, device=dev)
        # descriptor / point pools in rows of `cap`: [0, NI) current views, [NI, 2NI) previous views, [2NI, 2NI + 8Q) half-image jobs
        NJ = 8 * Q
        self.NJ = NJ
        self.desc = torch.zeros((2 * NI + NJ, cap, 256), dtype=f32, device=dev)
        self.pts = torch.zeros((2 * NI + NJ, cap, 2), dtype=f32, device=dev)
        self.cnt = torch.zeros((2 * NI + NJ,), dtype=i32, device=dev)
        self.scores = torch.zeros((NI, cap), dtype=f32, device=dev)
        self.kidx = torch.zeros((NI, cap), dtype=i32, device=dev)
        self.gdesc = torch.zeros((NI, fe.netvlad_dim), dtype=f32, device=dev)
        self.maps = torch.zeros((NJ, cap), dtype=i32, device=dev)
        mc = float(fe.half_move_cols(UW, undistort_fov))
        job_row, job_left, job_shift, a_off, b_off, a_row, b_row, map_a, map_b = [], [], [], [], [], [], [], [], []
        for q in range(Q):
            for (ca, cb, typ) in NEIGHBOURS:
                ja = len(job_row); job_row.append(ca * Q + q); job_left.append(1 if typ == 1 else 0); job_shift.append(mc if typ == 1 else -mc)
                jb = len(job_row); job_row.append(cb * Q + q); job_left.append(1 if typ == 2 else 0); job_shift.append(0.0)
                a_off.append((2 * NI + ja) * cap); b_off.append((2 * NI + jb) * cap)
                a_row.append(2 * NI + ja); b_row.append(2 * NI + jb); map_a.append(ja); map_b.append(jb)
        self.n_nb = len(a_off)
        for v in range(NI):                      # temporal: view v against the same view of the previous step (whole image, no radius)
            a_off.append(v * cap); b_off.append((NI + v) * cap); a_row.append(v); b_row.append(NI + v)
        self.NP = len(a_off)
        t = lambda x, dt: torch.tensor(x, dtype=dt, device=dev)
        self.job_row, self.job_left, self.job_shift = t(job_row, i32), t(job_left, i32), t(job_shift, f32)
        self.a_off, self.b_off = t(a_off, i32), t(b_off, i32)
        self.a_row, self.b_row = t(a_row, torch.int64), t(b_row, torch.int64)
        self.map_a, self.map_b = t(map_a, i32), t(map_b, i32)
        self.a_cnt = torch.zeros(self.NP, dtype=i32, device=dev); self.b_cnt = torch.zeros(self.NP, dtype=i32, device=dev)
        self.mq = torch.zeros((self.NP, cap), dtype=i32, device=dev); self.mt = torch.zeros((self.NP, cap), dtype=i32, device=dev)
        self.md = torch.zeros((self.NP, cap), dtype=f32, device=dev); self.mn = torch.zeros(self.NP, dtype=i32, device=dev)

    def step(self, raw, RH, RW, maps, st):
        """raw: u8 [4Q][RH][RW] camera-major; maps: per camera (mapx, mapy, gain) device tensors [UH][UW]."""
        fe, torch, Q, NI, cap, UH, UW = self.fe, self.torch, self.Q, self.NI, self.cap, self.UH, self.UW
        for c in range(4):
            mx, my, g = maps[c]
            fe.undistort_device(raw.data_ptr() + c * Q * RH * RW, Q, RW, RH, mx.data_ptr(), my.data_ptr(), g.data_ptr(), UW, UH,
                                self.und.data_ptr() + c * Q * UH * UW, stream=st)
        fe.netvlad_device(self.und.data_ptr(), NI, UW, UH, self.gdesc.data_ptr(), stream=st)
        fe.extract_device(self.und.data_ptr(), NI, UW, UH, self.pts.data_ptr(), self.scores.data_ptr(), self.desc.data_ptr(), self.kidx.data_ptr(),
                          cap, self.cnt.data_ptr(), stream=st)
        self.run_matching(st)
        self.desc[NI:2 * NI].copy_(self.desc[:NI]); self.pts[NI:2 * NI].copy_(self.pts[:NI]); self.cnt[NI:2 * NI].copy_(self.cnt[:NI])

    def run_matching(self, st):
        """Neighbour (half-image, shifted, radius-gated, remapped) + temporal matchKNN over the pools; also used by the chain parity test
        with the pools filled from the host."""
        fe, torch, NI, cap = self.fe, self.torch, self.NI, self.cap
        job0 = 2 * NI
        fe.half_image_compact_device(self.desc.data_ptr(), self.pts.data_ptr(), self.cnt.data_ptr(), self.job_row.data_ptr(), self.job_left.data_ptr(),
                                     self.job_shift.data_ptr(), self.NJ, cap, 256, self.UW, self.fov,
                                     self.desc[job0:].data_ptr(), self.pts[job0:].data_ptr(), self.maps.data_ptr(), self.cnt[job0:].data_ptr(), stream=st)
        torch.index_select(self.cnt, 0, self.a_row, out=self.a_cnt); torch.index_select(self.cnt, 0, self.b_row, out=self.b_cnt)
        nb = self.n_nb
        # neighbour pairs: radius gate on the shifted points; temporal pairs: whole image, no gate (two launches: the radius is per call)
        fe.match_batch_device(self.desc.data_ptr(), self.desc.data_ptr(), self.a_off.data_ptr(), self.b_off.data_ptr(), self.a_cnt.data_ptr(),
                              self.b_cnt.data_ptr(), nb, 256, cap, self.mq.data_ptr(), self.mt.data_ptr(), self.md.data_ptr(), self.mn.data_ptr(),
                              mode=0, ratio=self.ratio, radius=self.radius, d_pts_a=self.pts.data_ptr(), d_pts_b=self.pts.data_ptr(), stream=st)
        fe.remap_matches_device(self.mq.data_ptr(), self.mt.data_ptr(), self.mn.data_ptr(), self.map_a.data_ptr(), self.map_b.data_ptr(),
                                self.maps.data_ptr(), nb, cap, cap, stream=st)
        nt = self.NP - nb
        fe.match_batch_device(self.desc.data_ptr(), self.desc.data_ptr(), self.a_off[nb:].data_ptr(), self.b_off[nb:].data_ptr(),
                              self.a_cnt[nb:].data_ptr(), self.b_cnt[nb:].data_ptr(), nt, 256, cap, self.mq[nb:].data_ptr(), self.mt[nb:].data_ptr(),
                              self.md[nb:].data_ptr(), self.mn[nb:].data_ptr(), mode=0, ratio=self.ratio, radius=-1.0, stream=st)
