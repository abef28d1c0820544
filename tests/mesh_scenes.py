"""Synthetic triangle soups for the mesh z-buffer tests (face_verts[F,3,3]: pytorch3d NDC x, y and view-space depth z)."""
import numpy as np


def soup(F, seed, extent=1.4, size=0.08, zlo=0.5, zhi=6.0, dtype=np.float32):
    """small random triangles scattered over (a bit more than) the image"""
    rng = np.random.default_rng(seed)
    c = np.empty((F, 1, 3))
    c[..., 0] = rng.uniform(-extent, extent, (F, 1))
    c[..., 1] = rng.uniform(-extent, extent, (F, 1))
    c[..., 2] = rng.uniform(zlo, zhi, (F, 1))
    v = c + rng.normal(0, 1, (F, 3, 3)) * np.array([size, size, 0.15 * size * 10])
    v[..., 2] = np.maximum(v[..., 2], 0.05)
    return v.astype(dtype)


def mixed(F, seed):
    """small faces + a few screen-filling ones + slivers + degenerate / behind-the-camera / non-finite faces"""
    rng = np.random.default_rng(seed)
    v = soup(F, seed).astype(np.float64)
    n_big = max(2, F // 200)
    big = rng.choice(F, n_big, replace=False)
    v[big, :, :2] = rng.uniform(-3, 3, (n_big, 3, 2))
    v[big, :, 2] = rng.uniform(2.0, 9.0, (n_big, 3))
    sl = rng.choice(F, max(2, F // 50), replace=False)                      # slivers: third vertex almost on the first edge
    t = rng.uniform(0.2, 0.8, (len(sl), 1))
    v[sl, 2, :2] = v[sl, 0, :2] * (1 - t) + v[sl, 1, :2] * t + rng.normal(0, 1e-6, (len(sl), 2))
    dg = rng.choice(F, max(2, F // 100), replace=False)
    v[dg[0::4], 1] = v[dg[0::4], 0]                                         # two equal vertices: zero area
    v[dg[1::4], 0, 2] = -0.3                                                # one vertex behind the camera: z_invalid
    v[dg[2::4], 1, 0] = np.nan
    v[dg[3::4], 2, 2] = 5e-4                                                # very close: the "irregular" sort key
    return v.astype(np.float32)


def splat_like(P, seed, W, H):
    """two coplanar triangles per "Gaussian" (a camera-facing diamond, like SuGaR's splat mesh): faces 2g, 2g+1"""
    rng = np.random.default_rng(seed)
    ax, ay = W / min(W, H), H / min(W, H)
    cx = rng.uniform(-1.05 * ax, 1.05 * ax, P); cy = rng.uniform(-1.05 * ay, 1.05 * ay, P)
    z = rng.uniform(0.8, 8.0, P)
    r = rng.lognormal(np.log(0.012), 0.6, P) / np.sqrt(z / 2.0)
    ang = rng.uniform(0, np.pi, P)
    e1 = np.stack([np.cos(ang), np.sin(ang)], -1) * r[:, None]
    e2 = np.stack([-np.sin(ang), np.cos(ang)], -1) * (r * rng.uniform(0.3, 1.0, P))[:, None]
    c = np.stack([cx, cy], -1)
    q = np.stack([c + e1, c + e2, c - e1, c - e2], 1)                        # diamond corners 0..3
    zz = z[:, None] * (1 + rng.normal(0, 0.01, (P, 4)))
    vq = np.concatenate([q, zz[..., None]], -1)
    tri = np.array([[0, 2, 1], [0, 3, 2]])                                   # sugar_model.py:255
    return vq[:, tri].reshape(2 * P, 3, 3).astype(np.float32)
