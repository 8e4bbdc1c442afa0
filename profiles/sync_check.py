"""Where does Generator3D6.refine synchronise with the device?  Runs a few refine() passes with torch's sync debug mode on
(warnings at every implicit device synchronisation: .item(), .cpu(), bool(tensor) ...) and times M = 48 and M = 100 passes.
usage: python3 profiles/sync_check.py"""
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sapcu_amd  # noqa: E402
from sapcu_amd import testing as T  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    fn, fd, _, _ = bench.build_models(dev)
    cloud = torch.as_tensor(T.sphere_cloud(bench.N_CLOUD, 0), device=dev)
    seeds = torch.as_tensor(T.grid_queries(bench.B_PER_GPU, 0), device=dev)
    for m in (48, 100):
        gen = sapcu_amd.Generator3D6(fn, fd, dev, k_neighbors=m, batch_size=bench.B_PER_GPU)
        with torch.no_grad():
            gen.refine(cloud, seeds)
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode("warn")
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                gen.refine(cloud, seeds)
            torch.cuda.set_sync_debug_mode("default")
            print("M=%d: %d synchronisation warnings in one refine()" % (m, len(w)))
            for x in w[:8]:
                print("   ", str(x.message)[:100], "@", x.filename.split("/")[-1], x.lineno)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                gen.refine(cloud, seeds)
            torch.cuda.synchronize()
            print("M=%d: %.2f ms per refine of %d seeds" % (m, (time.perf_counter() - t0) / 5 * 1e3, bench.B_PER_GPU))


if __name__ == "__main__":
    main()
