"""CPU ORACLE (test infrastructure, NOT product code) — the per-cloud steps around ``upsample``.

numpy restatement of /root/reference/generate.py:43-74 (``normalize_pointcloud``, ``farthest_point_sample``).
Only ``tests/`` may import it.  Pinned by tests/golden/fps.npz, which tests/golden/make_fixtures.py produced by
running the reference's own functions (torch, CPU).
"""
import numpy as np


def normalize_pointcloud(cloud):
    """generate.py:43-54."""
    lo, hi = np.min(cloud, axis=0), np.max(cloud, axis=0)
    loc = (lo + hi) / 2
    scale = (hi - lo).max()
    inv = 1.0 / scale if scale > 0 else 1.0
    return (cloud - loc) * inv, loc, scale


def farthest_point_sample(xyz, npoint):
    """generate.py:56-74 in float32: start at N//2; each step takes the point with the largest running minimum of
    the squared distance ((dx*dx + dy*dy) + dz*dz, every operation rounded to f32) to the chosen set; the first of
    equal maxima wins (torch.max's documented rule)."""
    x = np.asarray(xyz).astype(np.float32)
    n = x.shape[0]
    out = np.zeros(npoint, dtype=np.int64)
    distance = np.full(n, 1e32, dtype=np.float32)
    far = n // 2
    for i in range(npoint):
        out[i] = far
        d = x - x[far]
        dist = d[:, 0] * d[:, 0]
        dist = dist + d[:, 1] * d[:, 1]
        dist = dist + d[:, 2] * d[:, 2]
        np.minimum(distance, dist, out=distance)
        far = int(np.argmax(distance))
    return out
