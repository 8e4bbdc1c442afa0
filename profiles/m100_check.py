#!/usr/bin/env python3
"""Sanity run at the reference's DEFAULT patch size (k_neighbors = 100, batch_size = 400): 40 000 seeds through Generator3D6.refine."""
import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench, sapcu_amd
from sapcu_amd import testing as T, generation as gen
dev = torch.device('cuda:0')
fn, fd, _, _ = bench.build_models(dev)
g = sapcu_amd.Generator3D6(fn, fd, dev, k_neighbors=100, dense_spacing=0.004, batch_size=400)
cloud = T.sphere_cloud(5000, 0)
seeds = gen.dense_seeds(cloud, 0.004)[:40000]
c, s = torch.as_tensor(cloud, device=dev), torch.as_tensor(seeds, device=dev)
with torch.no_grad():
    g.refine(c, s[:4400]); torch.cuda.synchronize()
    t = time.perf_counter(); out, _, _ = g.refine(c, s); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("M=100 (reference default k_neighbors), batch_size=400: %d seeds in %.2f s = %.0f q/s; finite=%s; peak mem %.1f GB" % (
    s.shape[0], dt, s.shape[0] / dt, bool(torch.isfinite(out).all()), torch.cuda.max_memory_allocated() / 1e9))
g.check_numeric_guards()
