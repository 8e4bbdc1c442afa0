#!/usr/bin/env python3
"""Reader of the fused fd encoder's diagnostic stamps (library built with EXTRA_CXXFLAGS=-DFE_STAMPS): s_memtime at the phase
boundaries of fd_encoder_kernel (wave 0, lane 0), median over the patches of one 4096-patch forward."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    from sapcu_amd import testing as T, generation as gen
    B, M = 4096, 48
    os.environ["SAPCU_FD_FUSED"] = "1"
    _, fd, _, _ = bench.build_models(dev)
    cloud = torch.as_tensor(T.sphere_cloud(5000, 0), device=dev)
    seeds = torch.as_tensor(T.grid_queries(B, 0), device=dev)
    _, _, patch = gen.knn_gather(cloud, seeds, M)
    stamps = torch.zeros((B, 32), dtype=torch.int64, device=dev)
    buf = stamps.view(torch.float32)
    with torch.no_grad():
        fd(patch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fd(patch, taps={"spikes": buf})
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    s = stamps.cpu().numpy().astype(np.float64)
    total = s[:, 19] - s[:, 0]
    names = {1: "xyz kNN", 2: "block-0 EdgeConv", 3: "scale fusion GEMM",
             4: "L1 kNN", 5: "L1 panel", 6: "L1 GEMM", 7: "L1 max",
             8: "L2 kNN", 9: "L2 panel", 10: "L2 GEMM", 11: "L2 max",
             12: "L3 kNN", 13: "L3 panel", 14: "L3 GEMM", 15: "L3 max + transposition", 16: "(to MSC start)"}
    print("forward wall %.2f ms; per patch (median over %d patches), s_memtime ticks:" % (wall * 1e3, B))
    print("  total %.0f ticks" % np.median(total))
    prev = 0
    for i in range(1, 17):
        d = s[:, i] - s[:, prev]
        print("  %-26s %8.0f  (%.1f %%)" % (names[i], np.median(d), 100 * np.median(d) / np.median(total)))
        prev = i
    msc = s[:, 19] - s[:, 16]
    print("  %-26s %8.0f  (%.1f %%)  = emission %.0f + MFMA rounds %.0f + epilogues etc. %.0f" % (
        "multi_scale_conv", np.median(msc), 100 * np.median(msc) / np.median(total), np.median(s[:, 17]), np.median(s[:, 18]),
        np.median(msc - s[:, 17] - s[:, 18])))
    if s[:, 28].any() or s[:, 29].any():
        print("  multi_scale_conv, per patch: starts of the thirds (x0 dump, accumulators, first fragments) %.0f, epilogues of the thirds %.0f"
              % (np.median(s[:, 28]), np.median(s[:, 29])))
    if s[:, 20].any():      # finer stamps inside fe_knn (ad-hoc diagnostic builds): xyz search at 20.., block-3 search at 24..
        for base, name in ((20, "xyz search"), (24, "L3 search")):
            print("  %s: to end of chunk loop %.0f, keys %.0f" % (name, np.median(s[:, base + 1] - s[:, base]), np.median(s[:, base + 2] - s[:, base + 1])))
        print("  xyz rank + tail %.0f ; L3 rank + tail %.0f" % (np.median(s[:, 1] - s[:, 22]), np.median(s[:, 12] - s[:, 26])))
    # ticks -> time: patches per CU = B / 256 run back to back
    print("  => %.3f us per tick if the forward is %d patches per CU back to back" % (wall * 1e6 / (B / 256.0) / np.median(total), B // 256))


if __name__ == "__main__":
    main()
