#!/usr/bin/env python3
"""FPS timing on the GPU box: the persistent HIP kernel (csrc/fps.hip) against the step-per-launch torch loop
(what generate.py:56-74 executes on a GPU).  Usage: python profiles/fps_microbench.py [n] [npoint]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sapcu_amd  # noqa: E402,F401
from sapcu_amd import pipeline  # noqa: E402


def torch_loop(x, npoint):
    n = x.shape[0]
    picks = torch.zeros(npoint, dtype=torch.long, device=x.device)
    running = torch.full((n,), 1e32, device=x.device)
    far = torch.tensor([n // 2], dtype=torch.long, device=x.device)
    for i in range(npoint):
        picks[i] = far
        d = ((x - x[far, :]) ** 2).sum(-1)
        m = d < running
        running[m] = d[m]
        far = running.max(-1)[1]
    return picks


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 385582
    npoint = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((n, 3))).float().cuda()
    pipeline.farthest_point_sample_device(x, 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = pipeline.farthest_point_sample_device(x, npoint)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sub = min(npoint, 1024)
    torch_loop(x, 8)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    b = torch_loop(x, sub)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print({"n": n, "npoint": npoint, "hip_ms": round((t1 - t0) * 1e3, 2), "hip_us_per_step": round((t1 - t0) * 1e6 / npoint, 2),
           "torch_loop_us_per_step": round((t3 - t2) * 1e6 / sub, 2), "same_indices": bool((a[:sub] == b).all())})


if __name__ == "__main__":
    main()
