#!/usr/bin/env python3
"""Outer kNN kernel alone (generation.py:110,127 / :176-183): time and the north-star's "streamed bytes" roofline model —
every query reads the whole cloud once from HBM (B * N * 24 bytes of f64 xyz).  The kernel tiles the cloud through LDS and
reuses each tile for all queries of a workgroup, so the model overstates its real HBM traffic; it is the accounting the
target (>= 40 % of the 8 TB/s HBM peak on the kNN kernel) is stated in.  Usage: knn_microbench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sapcu_amd  # noqa: E402,F401
from sapcu_amd import generation as gen  # noqa: E402


def run(n, b, k, reps):
    rng = np.random.default_rng(0)
    cloud = torch.as_tensor(rng.standard_normal((n, 3)), device="cuda")
    q = cloud[:b].contiguous() if b <= n else torch.as_tensor(rng.standard_normal((b, 3)), device="cuda")
    gen.knn_gather(cloud, q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gen.knn_gather(cloud, q, k)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    streamed = b * n * 24.0
    print("N=%d B=%d k=%d: %.1f us | streamed-bytes model %.2f TB/s = %.0f %% of 8 TB/s | %.1f G pair distances/s (f64)" %
          (n, b, k, t * 1e6, streamed / t / 1e12, 100 * streamed / t / 8e12, b * n / t / 1e9))


if __name__ == "__main__":
    run(5000, 4096, 48, 20)          # the bench workload's kNN
    run(5000, 65536, 48, 5)          # one cloud's seeds in large batches
    run(385582, 385582, 30, 1)       # the outlier filter's self-kNN (generation.py:176-183)
