#!/usr/bin/env python3
"""Per-shape timing of the three edge GEMM flavours of one fn block (chunk of 2048 patches, M=48):
pos-enc GEMM with the attention epilogue (gemm_ring_kernel<EPI_LIF_ATTN>), LIF epilogue, bias epilogue.
SAPCU_LIB=<path> selects another build of the library (ablations).  Usage: attn_gemm_microbench.py [chunk]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sapcu_amd  # noqa: E402,F401
from sapcu_amd import _lib  # noqa: E402

M_PTS = 48


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = torch.device("cuda:0")
    lib = _lib.load(os.environ.get("SAPCU_LIB"))
    torch.manual_seed(0)
    out = []
    for l, kk in enumerate((24, 18, 12)):
        d = 128 << l
        pts = chunk * M_PTS
        r = pts * kk
        pe = torch.rand((r, d), device=dev)
        pes = torch.empty_like(pe)
        _lib.check(lib.sapcu_to_split_rows(_lib.ptr(pe), r, d, d, _lib.ptr(pes), d, _lib.current_stream()))
        qkv = torch.rand((pts, 3 * d), device=dev)
        idx = torch.randint(0, M_PTS, (r,), dtype=torch.int32, device=dev)
        w = (torch.rand((d, d), device=dev) - 0.5) * (2.0 / d ** 0.5)
        bias = torch.rand((d,), device=dev) + 0.3
        lif = torch.stack([torch.full((d,), 0.9), torch.full((d,), 0.01), torch.full((d,), 0.5), torch.ones(d)]).to(dev)
        o1 = torch.empty((r, d), device=dev)
        o2 = torch.empty((r, d), device=dev)
        tab = torch.empty((r, 2), dtype=torch.int32, device=dev)
        w16 = torch.zeros(4 * d * d + 16, dtype=torch.uint8, device=dev)
        st = _lib.current_stream()

        def attn():
            _lib.check(lib.sapcu_posenc_gemm_f32(_lib.ptr(pes), r, d, _lib.ptr(w), _lib.ptr(bias), _lib.ptr(lif), 4, _lib.ptr(qkv),
                                                 _lib.ptr(idx), kk, M_PTS, _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(tab), _lib.ptr(w16), 1, st))

        def plain(lif_on):
            _lib.check(lib.sapcu_gemm_f32(_lib.ptr(pes), r, d, d, _lib.ptr(w), d, _lib.ptr(bias), _lib.ptr(lif) if lif_on else None, 4,
                                          _lib.ptr(o1), d, _lib.ptr(w16), 1, 1, st))

        ta, tl, tb = timeit(attn), timeit(lambda: plain(True)), timeit(lambda: plain(False))
        gb = r * d * 4 / 1e9
        out.append("d=%d r=%d: attn %.0f us (%.2f TB/s of 3 passes) | lif %.0f us (%.2f TB/s of 2) | bias %.0f us (%.2f TB/s of 2)" %
                   (d, r, ta * 1e6, 3 * gb / ta / 1e3, tl * 1e6, 2 * gb / tl / 1e3, tb * 1e6, 2 * gb / tb / 1e3))
        del pe, pes, qkv, idx, o1, o2, tab
    print("\n".join(out))


if __name__ == "__main__":
    main()
