#!/usr/bin/env python3
"""Where a tile of the big-tile GEMM spends its time (diagnostic build libsapcu_BT_STAMPS.so, profiles/ablate.sh BT_STAMPS):
s_memtime stamps in wave 0 (activation stream) and wave 4 (weight stream), median over workgroups.
Usage: python profiles/bt_stamps.py R K N [lif]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sapcu_amd  # noqa: E402,F401
from sapcu_amd import _lib  # noqa: E402

NAMES = ["k-step 0 of a tile (whole)", "k-step 1 (whole)", "k-steps 2.. (whole, sum)", "  wait own DMA, k-step 0", "  wait own DMA, k-step 1",
         "  wait own DMA, k-steps 2.. (sum)", "  barrier, k-step 0", "  barrier, k-step 1", "  barrier, k-steps 2.. (sum)",
         "epilogue: parameter loads + wait", "epilogue: arithmetic + stores issued"]


def main():
    r, k, n = (int(x) for x in sys.argv[1:4])
    lif_on = "lif" in sys.argv[4:]
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "abl", "libsapcu_BT_STAMPS.so")
    lib, raw = _lib.load(so), ctypes.CDLL(so)
    dev = torch.device("cuda:0")
    a = torch.rand((r, k), device=dev)
    a2 = torch.empty_like(a)
    _lib.check(lib.sapcu_to_split_rows(_lib.ptr(a), r, k, k, _lib.ptr(a2), k, _lib.current_stream()))
    w = (torch.rand((n, k), device=dev) - 0.5) * (2.0 / k ** 0.5)
    b = torch.rand((n,), device=dev)
    c = torch.empty((r, n), device=dev)
    lif = torch.stack([torch.full((n,), 0.9), torch.full((n,), 0.01), torch.full((n,), 0.5), torch.ones(n)]).to(dev)
    ws = torch.zeros(4 * n * k + 16, dtype=torch.uint8, device=dev)
    for _ in range(2):
        _lib.check(lib.sapcu_gemm_f32(_lib.ptr(a2), r, k, k, _lib.ptr(w), n, _lib.ptr(b), _lib.ptr(lif) if lif_on else None, 4,
                                      _lib.ptr(c), n, _lib.ptr(ws), 1, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    out = np.zeros((256, 2, 44), dtype=np.uint64)
    assert raw.sapcu_debug_bt_stamps(ctypes.c_void_p(out.ctypes.data)) == 0
    out = out[out[:, 0, 11] > 0].astype(np.float64)
    print("r=%d k=%d n=%d %s: %d workgroups, %.0f tiles each; s_memtime ticks PER TILE (median over workgroups)" %
          (r, k, n, "lif" if lif_on else "bias", out.shape[0], np.median(out[:, 0, 11])))
    for wv, nm in ((0, "wave 0 (activation stream)"), (1, "wave 4 (weight stream)")):
        tiles = out[:, wv, 11]
        tot = (out[:, wv, 0] + out[:, wv, 1] + out[:, wv, 2] + out[:, wv, 9] + out[:, wv, 10]) / tiles
        print(" %s: %.0f ticks per tile" % (nm, np.median(tot)))
        for i, name in enumerate(NAMES):
            print("   %-40s %8.0f   %5.1f %%" % (name, np.median(out[:, wv, i] / tiles), 100 * np.median(out[:, wv, i] / tiles / tot)))
        print("   whole k-step by index in the tile: " + " ".join("%.0f" % np.median(out[:, wv, 12 + j] / tiles) for j in range(min(k // 32, 32))))


if __name__ == "__main__":
    main()
