"""Helpers for the GPU parity tests (and __graft_entry__.smoke / bench cpu-baseline checks)."""
import numpy as np
import torch

from oracle import geom_path as G
from oracle import snn_path as O

FN_HP = {"k_values": [24, 18, 12], "emb_dims": 640, "time_steps_enc": 4, "num_heads": 8}
FD_HP = {"k": 32, "k_scales": [8, 16, 32, 48], "emb_dims": 768, "time_steps_enc": 4, "num_heads": 8}


def dev():
    return torch.device("cuda:0")


def build_gpu_models(weights, fn_over=None, fd_over=None):
    import sapcu_amd
    from conftest import FD_KW, FN_KW
    fn = sapcu_amd.ImprovedSNNNormalEstimation(**dict(FN_KW, **(fn_over or {})))
    fd = sapcu_amd.EnhancedSNNDistanceEstimation(**dict(FD_KW, **(fd_over or {})))
    sdn, sdd = weights("fn", **(fn_over or {})), weights("fd", **(fd_over or {}))
    fn.load_state_dict(sdn, strict=True)
    fd.load_state_dict(sdd, strict=True)
    return fn.to(dev()), fd.to(dev()), sdn, sdd


def build_gpu_models_under(weights, monkeypatch, env, fn_over=None, fd_over=None):
    """Models whose HANDLES were created under the environment switches `env` (sapcu_model_create reads SAPCU_* once; a forward
    never reads the environment): the switches are set, both engines are built, the switches are removed again."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    out = build_gpu_models(weights, fn_over, fd_over)
    out[0]._engine(), out[1]._engine()
    for k in env:
        monkeypatch.delenv(k, raising=False)
    out[0].knn_cache_mode = "fresh"
    return out


def sphere_patches(nq, k, n=5000, qseed=0, skip=0):
    from sapcu_amd import testing as T
    cloud = T.sphere_cloud(n, 0)
    q = T.grid_queries(nq + skip, qseed)[skip:]
    idx = G.knn_bruteforce(cloud, q, k)
    return torch.from_numpy(G.gather_centre(cloud, q, idx)).float()


def fd_forward_forced(fd_gpu, sdd, patch_cpu, hp=FD_HP):
    """Run the device fd, read back the feature-space neighbour tables it chose, evaluate the oracle on
    exactly those.  Returns (gpu_dist, oracle_dist_forced, oracle_dist_free, flip_rows[3], taps)."""
    b, m = patch_cpu.shape[0], patch_cpu.shape[1]
    kk = min(hp["k"], m)
    knn = torch.empty((3, b, m, kk), dtype=torch.int32, device=dev())
    d_gpu = fd_gpu(patch_cpu.to(dev()), taps={"knn": knn})
    torch.cuda.synchronize()
    knn_cpu = knn.cpu().long()
    taps_free = {}
    with torch.no_grad():
        d_forced = O.fd_forward(fd_state(sdd), patch_cpu, hp, force_idx=[knn_cpu[0], knn_cpu[1], knn_cpu[2]])
        d_free = O.fd_forward(fd_state(sdd), patch_cpu, hp, taps=taps_free)
    flips = []
    for i in (1, 2, 3):
        a = taps_free["encoder.knn%d" % i].sort(-1)[0]
        g = knn_cpu[i - 1].sort(-1)[0]
        flips.append((a != g).any(-1))           # [b, m] rows whose neighbour SET differs
    return d_gpu.cpu(), d_forced, d_free, flips, taps_free


def fd_state(sd):
    return sd
