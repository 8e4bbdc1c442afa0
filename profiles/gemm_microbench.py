#!/usr/bin/env python3
"""Micro-benchmark of the library GEMMs on one shape: python profiles/gemm_microbench.py R K N [lif] [f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sapcu_amd  # noqa: E402
from sapcu_amd import _lib  # noqa: E402


def main():
    r, k, n = (int(x) for x in sys.argv[1:4])
    lif_on = "lif" in sys.argv[4:]
    f32 = "f32" in sys.argv[4:]
    dev = torch.device("cuda:0")
    lib = _lib.load(os.environ.get("SAPCU_LIB"))
    a = torch.rand((r, k), device=dev)
    w = (torch.rand((n, k), device=dev) - 0.5) * (2.0 / k ** 0.5)
    b = torch.rand((n,), device=dev)
    c = torch.empty((r, n), device=dev)
    lif = torch.stack([torch.full((n,), 0.9), torch.full((n,), 0.01), torch.full((n,), 0.5), torch.ones(n)]).to(dev)
    ws = None if f32 else torch.zeros(4 * n * k + 16, dtype=torch.uint8, device=dev)
    ring = "ring" in sys.argv[4:]
    if ring:
        a2 = torch.empty_like(a)
        _lib.check(lib.sapcu_to_split_rows(_lib.ptr(a), r, k, k, _lib.ptr(a2), k, _lib.current_stream()))
        a = a2

    def run():
        _lib.check(lib.sapcu_gemm_f32(_lib.ptr(a), r, k, k, _lib.ptr(w), n, _lib.ptr(b), _lib.ptr(lif) if lif_on else None, 4,
                                      _lib.ptr(c), n, _lib.ptr(ws), 1 if ring else 0, 0, _lib.current_stream()))

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    print("r=%d k=%d n=%d %s %s: %.1f us  %.1f TFLOP/s algorithmic  %.2f TB/s (A+C)" %
          (r, k, n, "lif" if lif_on else "bias", "f32" if f32 else ("ring" if ring else "sf16"), t * 1e6, 2.0 * r * k * n / t / 1e12,
           (r * k + r * n) * 4 / t / 1e12))


if __name__ == "__main__":
    main()
