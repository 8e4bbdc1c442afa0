"""CPU ORACLE (test infrastructure, NOT product code) — the float64 geometry around the nets.

numpy restatement of the per-batch steps of ``Generator3D6.generateiopoint``
(/root/reference/generation.py:122-172) plus ``rotation_matrix_from_vectors`` (:30-47).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.

The reference's outer kNN is ``sklearn.neighbors.KDTree(data).query(q, k)`` (generation.py
:110,127,153; sklearn 1.7.2, euclidean, leaf 40, sorted).  sklearn is a third-party
dependency that is not vendored in /root/reference; its published algorithm for this metric is
"reduced distance" ``sum_c (q_c - p_c)^2`` accumulated left to right in float64, k smallest,
ascending.  ``knn_bruteforce`` restates exactly that arithmetic (no FMA: numpy float64 mul and
add are separate IEEE operations) and orders equal distances by ascending point index.  It is
pinned against a KDTree golden vector (tests/golden/outer_knn.npz).
"""
import numpy as np


def knn_bruteforce(cloud, queries, k):
    """cloud [N,3] f64, queries [B,3] f64 -> idx [B,k] int64 ascending (dist, index)."""
    cloud = np.asarray(cloud, dtype=np.float64)
    queries = np.asarray(queries, dtype=np.float64)
    out = np.empty((queries.shape[0], k), dtype=np.int64)
    step = max(1, (1 << 22) // max(1, cloud.shape[0]))
    for s in range(0, queries.shape[0], step):
        q = queries[s:s + step]
        dx = q[:, None, 0] - cloud[None, :, 0]
        dy = q[:, None, 1] - cloud[None, :, 1]
        dz = q[:, None, 2] - cloud[None, :, 2]
        d = dx * dx
        d = d + dy * dy
        d = d + dz * dz
        out[s:s + step] = np.argsort(d, axis=1, kind="stable")[:, :k]
    return out


def gather_centre(cloud, queries, idx):
    """patch = cloud[idx] - q in float64 (generation.py:128-129); the caller rounds to f32."""
    return cloud[idx] - queries[:, None, :]


def rotation_to_x(n):
    """Rodrigues matrix taking unit(n) onto +x (generation.py:30-47, vec2 = [1,0,0]).

    Identity when the cross product is exactly zero — also for n = -x (reference quirk)."""
    n = np.asarray(n)                        # stays float32 when the model output is (as in the reference)
    a = (n / np.linalg.norm(n)).reshape(3)   # normalised in n's own precision (generation.py:39)
    b = np.array([1.0, 0.0, 0.0])
    v = np.cross(a, b)
    if not v.any():
        return np.eye(3)
    c = np.dot(a, b)
    s = np.linalg.norm(v)
    K = np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    return np.eye(3) + K + K.dot(K) * ((1 - c) / (s ** 2))


def rotate_patches(patches, normals):
    """patches [b,M,3] f64, normals [b,3] -> rotated copy, one matrix per patch (:158-160)."""
    out = np.empty_like(patches)
    for j in range(patches.shape[0]):
        R = rotation_to_x(normals[j])
        out[j] = np.matmul(R, patches[j].T).T
    return out


def displace(queries, normals, dist):
    """p + n*d with the f32 product promoted to f64 (generation.py:171-172)."""
    length = np.tile(np.expand_dims(dist, 1), (1, 3))          # f32
    return queries + normals * length                           # (f32*f32 -> f32) + f64


def split_batches(n, batch_size):
    """Chunk boundaries of np.array_split(seeds, max(1, n // batch_size)) (generation.py:122-123)."""
    pp = max(1, n // batch_size)
    sizes = [n // pp + (1 if i < n % pp else 0) for i in range(pp)]
    edges = np.concatenate([[0], np.cumsum(sizes)])
    return [(int(edges[i]), int(edges[i + 1])) for i in range(pp)]


def upsample_core(cloud, seeds, fn_fwd, fd_fwd, k_neighbors, batch_size, knn_cache_mode="reference"):
    """The two hot loops of generateiopoint (generation.py:126-172) without seed generation
    (:112-118) and the outlier filter (:176-183).

    fn_fwd(patch_f32 [b,M,3], knn_idx or None) -> (normals f32 [b,3] already unit, knn_idx used)
    fd_fwd(patch_f32 [b,M,3]) -> distances f32 [b]
    ``knn_cache_mode='reference'`` reproduces fn's shape-keyed KNNCache (fn/snn_coder.py:47-59):
    the first batch of each distinct batch size fixes the in-patch neighbour indices that all later
    batches of that size reuse.  Returns (refined [n,3] f64, normals [n,3] f32, dist [n] f32, idx)."""
    import torch
    cloud = np.asarray(cloud, dtype=np.float64)
    seeds = np.asarray(seeds, dtype=np.float64)
    chunks = split_batches(seeds.shape[0], batch_size)
    cache = {}
    normals = []
    all_idx = []
    for (s, e) in chunks:
        q = seeds[s:e]
        idx = knn_bruteforce(cloud, q, k_neighbors)
        all_idx.append(idx)
        patch = torch.from_numpy(gather_centre(cloud, q, idx)).float()
        key = e - s
        pre = cache.get(key) if knn_cache_mode == "reference" else None
        n, used = fn_fwd(patch, pre)
        if knn_cache_mode == "reference" and key not in cache:
            cache[key] = used
        n = torch.nn.functional.normalize(n, dim=-1)            # generation.py:139
        normals.append(n.numpy())
    normals = np.concatenate(normals, axis=0)
    out = []
    dists = []
    for ci, (s, e) in enumerate(chunks):
        q = seeds[s:e]
        patch = gather_centre(cloud, q, all_idx[ci])
        patch = rotate_patches(patch, normals[s:e])
        d = fd_fwd(torch.from_numpy(patch).float()).numpy()
        dists.append(d)
        out.append(displace(q, normals[s:e], d))
    return (np.concatenate(out, axis=0), normals, np.concatenate(dists, axis=0),
            np.concatenate(all_idx, axis=0))


def outlier_filter(xyz, threshold=1.5, k=30):
    """Boolean keep-mask of generation.py:176-183: mean distance to the k nearest points of the
    output itself (self included, distance 0) below ``threshold`` x the global mean."""
    xyz = np.asarray(xyz, dtype=np.float64)
    k = min(k, xyz.shape[0])
    idx = knn_bruteforce(xyz, xyz, k)
    diff = xyz[idx] - xyz[:, None, :]
    d2 = diff[..., 0] * diff[..., 0]
    d2 = d2 + diff[..., 1] * diff[..., 1]
    d2 = d2 + diff[..., 2] * diff[..., 2]
    dist = np.sqrt(d2)
    return np.mean(dist, axis=1) < np.mean(dist) * threshold
