"""Synthetic DTU-scan1-like inputs (no dataset exists offline): BASELINE.md §4 / SURVEY.md §8(d).

Pure numpy; shared by bench.py, the tests and the golden-vector generator so that every party
sees bit-identical inputs for a given seed.
"""
import numpy as np

# scan1 bbox grown by world_bound_scale=1.5 (configs/dtu_e2e/scan1.py:37-38, lib/recon_scene.py:132-136)
XYZ_MIN = np.array([-0.6, -0.6, -0.7], dtype=np.float32)
XYZ_MAX = np.array([0.6, 0.6, 0.5], dtype=np.float32)
NEAR, FAR = 0.24, 4.8                       # 0.3*0.8, 4.0*1.2 (scan1.py:35-36, load_data.py:91)


def range_shape():
    """range_shape / rect_size as built at lib/recon_scene.py:138-150: bbox / (1.5*1.05)."""
    return ((XYZ_MAX - XYZ_MIN) / (1.5 * 1.05)).astype(np.float32)


def look_at_w2c(cam_pos, target):
    """OpenCV-convention (x right, y down, z forward; inverse_y=True) world->camera [3,4]."""
    z = target - cam_pos
    z = z / np.linalg.norm(z)
    up = np.array([0., -1., 0.])
    x = np.cross(up, z)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], 0)               # rows = camera axes in world coords
    t = -R @ cam_pos
    return np.concatenate([R, t[:, None]], 1).astype(np.float32)


def cameras(n_views=3, radius=2.2, yaw_span=0.3):
    """n_views cameras on an arc of +-yaw_span rad around the bbox centre, looking at it."""
    centre = ((XYZ_MIN + XYZ_MAX) / 2).astype(np.float64)
    yaws = np.linspace(-yaw_span, yaw_span, n_views) if n_views > 1 else np.array([0.])
    poses = []
    for a in yaws:
        pos = centre + radius * np.array([np.sin(a), 0.0, -np.cos(a)])
        poses.append(look_at_w2c(pos, centre))
    return np.stack(poses, 0)


def intrinsics(n_views, H, W, focal=None):
    f = float(focal if focal is not None else 500. * W / 400.)
    K = np.array([[f, 0, W / 2.], [0, f, H / 2.], [0, 0, 1]], dtype=np.float32)
    return np.repeat(K[None], n_views, 0)


def make_views(n_views=3, H=400, W=400, seed=777):
    """-> dict(images [V,H,W,3] U[0,1), masks [V,H,W,1] Bernoulli(.5), Ks [V,3,3], w2c [V,3,4])."""
    rng = np.random.RandomState(seed)
    images = rng.rand(n_views, H, W, 3).astype(np.float32)
    masks = (rng.rand(n_views, H, W, 1) < 0.5).astype(np.float32)
    return dict(images=images, masks=masks, Ks=intrinsics(n_views, H, W), w2c=cameras(n_views))


def step_randomness(n_total_rays, n_rand, seed):
    """Ray indices (a prefix of a permutation, as recon_scene.py:598) and per-ray jitter U[0,1)."""
    rng = np.random.RandomState(seed)
    idx = rng.permutation(n_total_rays)[:n_rand].astype(np.int64)
    jitter = rng.rand(n_rand).astype(np.float32)
    return idx, jitter


def se3_perturbation(n_views, std=1e-2, seed=778):
    rng = np.random.RandomState(seed)
    return (rng.randn(n_views, 6) * std).astype(np.float32)


def voxel_id_grid(shape):
    """Deterministic [1,1,X,Y,Z] fp32 grid whose value identifies the voxel (flat index mod 4099, fp32-exact): a lookup
    that lands on a neighbouring voxel is visible.  Shared by oracle/make_golden.py (flatindex fixture) and the tests."""
    n = int(shape[0]) * int(shape[1]) * int(shape[2])
    return (np.arange(n, dtype=np.int64) % 4099).astype(np.float32).reshape(1, 1, *[int(v) for v in shape])
