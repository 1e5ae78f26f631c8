"""Seeded synthetic inputs shaped like the reference's workloads:
u8 gray frames with corner-rich structure, stereo pairs, and descriptor sets for the matcher."""
import numpy as np


def _blur3(img):
    k = np.array([1.0, 2.0, 1.0], np.float32) / 4.0
    p = np.pad(img, 1, mode="edge")
    t = p[:, :-2] * k[0] + p[:, 1:-1] * k[1] + p[:, 2:] * k[2]
    return t[:-2] * k[0] + t[1:-1] * k[1] + t[2:] * k[2]


def synth_image(h, w, seed, n_shapes=200):
    rng = np.random.RandomState(seed)
    img = np.full((h, w), rng.uniform(60, 180), np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_shapes):
        g = rng.uniform(0, 255)
        if rng.rand() < 0.6:
            x0 = rng.randint(0, w); y0 = rng.randint(0, h)
            ww = rng.randint(4, max(5, w // 6)); hh = rng.randint(4, max(5, h // 6))
            img[y0:y0 + hh, x0:x0 + ww] = g
        else:
            cx = rng.randint(0, w); cy = rng.randint(0, h); r = rng.randint(3, max(4, min(h, w) // 10))
            y1, y2 = max(0, cy - r), min(h, cy + r + 1); x1, x2 = max(0, cx - r), min(w, cx + r + 1)
            m = (yy[y1:y2, x1:x2] - cy) ** 2 + (xx[y1:y2, x1:x2] - cx) ** 2 <= r * r
            img[y1:y2, x1:x2][m] = g
    img = _blur3(img)
    img += rng.normal(0, 3.0, size=img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_stereo(h, w, seed):
    """Left frame + right frame = left shifted by a per-seed disparity (8..40 px) with independent noise."""
    rng = np.random.RandomState(seed + 100003)
    disp = int(rng.randint(8, 41))
    wide = synth_image(h, w + disp, seed).astype(np.float32)
    left = wide[:, disp:]
    right = wide[:, :w] + rng.normal(0, 3.0, size=(h, w)).astype(np.float32)
    return (np.clip(np.rint(left), 0, 255).astype(np.uint8),
            np.clip(np.rint(right), 0, 255).astype(np.uint8))


def synth_descriptor_pair(na, nb, dim, seed, sigma=0.05, outlier_frac=0.2):
    """A = unit rows ~ N(0,I); B = permuted A + N(0,sigma^2) with a fraction of outlier rows."""
    rng = np.random.RandomState(seed)
    a = rng.normal(size=(na, dim)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    src = rng.permutation(max(na, nb))[:nb] % max(na, 1)
    b = a[src] + rng.normal(0, sigma, size=(nb, dim)).astype(np.float32)
    out = rng.rand(nb) < outlier_frac
    b[out] = rng.normal(size=(int(out.sum()), dim)).astype(np.float32)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    pts_a = rng.uniform(0, 640, size=(na, 2)).astype(np.float32)
    pts_b = (pts_a[src] + rng.normal(0, 6.0, size=(nb, 2))).astype(np.float32)
    return a.astype(np.float32), b.astype(np.float32), pts_a, pts_b
