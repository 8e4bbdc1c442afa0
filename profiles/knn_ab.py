"""Timing of the in-patch kNN kernel (csrc/patch_ops.hip patch_knn_kernel) through its C-ABI entry at the bench shapes:
4096 patches x 48 points, c = 3 (xyz, k = 24) and c = 64 / 128 / 256 (fd's feature space, k = 32); plus M = 100.
SAPCU_AB_LIB=<path> times a diagnostic build of the library instead."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sapcu_amd import _lib  # noqa: E402

if os.environ.get("SAPCU_AB_LIB"):
    _lib.LIB_PATH = os.environ["SAPCU_AB_LIB"]


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    out = {"lib": os.environ.get("SAPCU_AB_LIB", "default")}
    for b, m, c, k in ((4096, 48, 3, 24), (4096, 48, 64, 32), (4096, 48, 256, 32), (4096, 100, 3, 24), (4096, 100, 256, 32)):
        g = torch.Generator(device="cpu").manual_seed(1)
        feat = torch.rand((b, m, c), generator=g).to(dev)
        idx = torch.empty((b, m, k), dtype=torch.int32, device=dev)

        def launch():
            _lib.check(lib.sapcu_patch_knn(_lib.ptr(feat), b, m, c, c, k, _lib.ptr(idx), _lib.current_stream()))

        launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            launch()
        e1.record()
        torch.cuda.synchronize()
        out["m%d_c%d_us" % (m, c)] = round(e0.elapsed_time(e1) * 100, 1)
        out["m%d_c%d_sum" % (m, c)] = int(idx.long().sum())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
