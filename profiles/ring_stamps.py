#!/usr/bin/env python3
"""Where a producer wave of the ring GEMM spends a k-step (diagnostic build libsapcu_STAMPS.so, profiles/ablate.sh).
Usage: python profiles/ring_stamps.py R K N [lif]   -- shares per segment, median over workgroups."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sapcu_amd  # noqa: E402,F401
from sapcu_amd import _lib  # noqa: E402

NAMES = ["reads(f1)+mfma6(f0)", "wait own DMA landed", "barrier", "DMA issue (4 pieces)", "reads(f0')+mfma6(f1)", "hand-off (per tile)"]


def main():
    r, k, n = (int(x) for x in sys.argv[1:4])
    lif_on = "lif" in sys.argv[4:]
    here = os.path.dirname(os.path.abspath(__file__))
    lib = _lib.load(os.path.join(here, "abl", "libsapcu_STAMPS.so"))
    raw = ctypes.CDLL(os.path.join(here, "abl", "libsapcu_STAMPS.so"))
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
                                      _lib.ptr(c), n, _lib.ptr(ws), 1, 1, _lib.current_stream()))
    torch.cuda.synchronize()
    out = np.zeros((256, 8), dtype=np.uint64)
    rc = raw.sapcu_debug_ring_stamps(ctypes.c_void_p(out.ctypes.data))
    assert rc == 0, rc
    out = out[out[:, 7] > 0].astype(np.float64)
    steps, tiles = out[:, 6], out[:, 7]
    print("r=%d k=%d n=%d %s: %d workgroups, %.0f tiles x %.0f k-steps each (s_memtime ticks; stamps drain LDS reads: read SHARES)" %
          (r, k, n, "lif" if lif_on else "bias", out.shape[0], np.median(tiles), np.median(steps / tiles)))
    tot = out[:, :6].sum(1)
    for i, name in enumerate(NAMES):
        per = out[:, i] / (tiles if i == 5 else steps)
        print("  %-24s %7.0f ticks per %s   %5.1f %%" % (name, np.median(per), "tile" if i == 5 else "k-step", 100 * np.median(out[:, i] / tot)))
    print("  total per tile %.0f ticks" % np.median(tot / tiles))


if __name__ == "__main__":
    main()
