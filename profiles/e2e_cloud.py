#!/usr/bin/env python3
"""Whole-cloud run of generate.py's per-cloud body (seeds in process -> hot path -> outlier filter -> FPS to 4N), stage-timed."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import sapcu_amd  # noqa: E402
from sapcu_amd import testing as T, generation as gen  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda:0")
    fn, fd, _, _ = bench.build_models(dev)
    g = sapcu_amd.Generator3D6(fn, fd, dev, k_neighbors=48, dense_spacing=0.004, batch_size=bs)
    cloud = T.sphere_cloud(n, 0)
    t0 = time.perf_counter()
    seeds = gen.dense_seeds(cloud, 0.004)
    t1 = time.perf_counter()
    c_dev, s_dev = torch.as_tensor(cloud, device=dev), torch.as_tensor(seeds, device=dev)
    with torch.no_grad():
        g.refine(c_dev, s_dev[: 4 * bs])                  # warm-up (workspace, module handles)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        refined, _, _ = g.refine(c_dev, s_dev)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        keep = g.outlier_filter(refined)
        t4 = time.perf_counter()
    out = refined.cpu().numpy()[keep]
    t5 = time.perf_counter()
    from sapcu_amd import pipeline
    target = min(4 * n, out.shape[0])                      # generate.py: FPS to 4x the input size
    pipeline.farthest_point_sample(out[:1000], 8)         # (library warm-up)
    t6 = time.perf_counter()
    picked = pipeline.farthest_point_sample(out, target)
    t7 = time.perf_counter()
    print("cloud N=%d: %d seeds | seeds %.2f s | hot path %.2f s (%.0f query-points/s) | outlier filter %.2f s | copy-out %.2f s | "
          "FPS to %d: %.3f s | kept %d | radius mean %.4f std %.4f" % (
              n, seeds.shape[0], t1 - t0, t3 - t2, seeds.shape[0] / (t3 - t2), t4 - t3, t5 - t4, target, t7 - t6, out.shape[0],
              np.linalg.norm(out, axis=1).mean(), np.linalg.norm(out, axis=1).std()))
    assert np.unique(picked).shape[0] == target


if __name__ == "__main__":
    main()
