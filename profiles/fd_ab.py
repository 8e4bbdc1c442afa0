#!/usr/bin/env python3
"""fd forward alone at the bench shape (B = 4096, M = 48, T = 4): fused encoder (csrc/fd_encoder.hip) against the per-stage
kernels (SAPCU_FD_FUSED=0 at handle creation), ms per forward; checks that the two agree bit for bit first."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def build(env):
    for k, v in env.items():
        os.environ[k] = v
    fn, fd, _, _ = bench.build_models(torch.device("cuda", 0))
    fd._engine()
    for k in env:
        del os.environ[k]
    return fd


def main():
    dev = torch.device("cuda", 0)
    from sapcu_amd import testing as T, generation as gen
    B, M = int(os.environ.get("FD_AB_B", "4096")), int(os.environ.get("FD_AB_M", "48"))
    cloud = torch.as_tensor(T.sphere_cloud(5000, 0), device=dev)
    seeds = torch.as_tensor(T.grid_queries(B, 0), device=dev)
    _, _, patch = gen.knn_gather(cloud, seeds, M)
    fused, stage = build({"SAPCU_FD_FUSED": "1"}), build({"SAPCU_FD_FUSED": "0"})
    print("fused_blocks:", fused.fused_blocks(M), stage.fused_blocks(M))
    with torch.no_grad():
        a, b = fused(patch), stage(patch)
        torch.cuda.synchronize()
        print("bit-identical distances:", bool(torch.equal(a, b)), "max diff %.3g" % float((a - b).abs().max()))
        for name, model in (("fused", fused), ("per-stage", stage)):
            for _ in range(2):
                model(patch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                model(patch)
            torch.cuda.synchronize()
            print("%-10s %.3f ms per forward (B=%d, M=%d)" % (name, (time.perf_counter() - t0) / n * 1e3, B, M))
    print("gate violations:", fused.gate_violations(), stage.gate_violations())


if __name__ == "__main__":
    main()
