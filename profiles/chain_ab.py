"""A/B timing of fn's fused edge-chain kernels (csrc/fn_edge_chain.hip) through their C-ABI entry, one launch shape per fn block:
(d, kk) = (128, 24), (256, 18), (512, 12) at 4096 patches x 48 points.  Environment switches of the library (e.g. SAPCU_CHAIN_PIPE)
are read per launch, so run it once per setting:   SAPCU_CHAIN_PIPE=2 python3 profiles/chain_ab.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sapcu_amd import _lib  # noqa: E402

if os.environ.get("SAPCU_AB_LIB"):        # a diagnostic build of the library (ablations)
    _lib.LIB_PATH = os.environ["SAPCU_AB_LIB"]


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    B, M, heads, T = 4096, 48, 8, 4
    P = B * M
    out = {"lib": os.environ.get("SAPCU_AB_LIB", "default")}
    for d, kk in ((128, 24), (256, 18), (512, 12)):
        g = torch.Generator(device="cpu").manual_seed(0)
        patch = (torch.randn((B, M, 3), generator=g) * 0.05).to(dev)
        idx = torch.stack([torch.randperm(M, generator=g)[:kk] for _ in range(P)]).to(torch.int32).to(dev)
        if os.environ.get("SAPCU_AB_ZERO_IDX"):        # every neighbour = point 0 of the patch: the k / v gathers always hit
            idx.zero_()
        qkv = torch.rand((P, 3 * d), generator=g).to(dev)

        def lin(n, k, gain):
            return ((torch.rand((n, k), generator=g) * 2 - 1) * gain / k ** 0.5).to(dev), (torch.randn(n, generator=g) * 0.4 + 0.6).to(dev)

        def lif():
            return torch.stack([torch.full((d,), 0.9), torch.full((d,), 0.01), torch.full((d,), 0.5), torch.ones(d)]).to(dev)

        wd, bd = lin(d, 3, 20.0)
        (w1, b1), (w2, b2), (w3, b3) = lin(d, d, 2.0), lin(d, d, 2.0), lin(d, d, 4.0)
        ops = [patch.view(P, 3), idx.view(-1), qkv, wd, bd, lif(), w1, b1, lif(), w2, b2, lif(), w3, b3]
        need = lib.sapcu_fn_edge_chain_workspace_bytes(P, d, kk)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        res = torch.empty((P, d), device=dev)

        def launch():
            _lib.check(lib.sapcu_fn_edge_chain_f32(_lib.ptr(ops[0]), _lib.ptr(ops[1]), P, M, d, kk, *[_lib.ptr(t) for t in ops[2:]],
                                                   heads, T, _lib.ptr(res), _lib.ptr(ws), need, _lib.current_stream()))

        launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            launch()
        e1.record()
        torch.cuda.synchronize()
        out["d%d_ms" % d] = round(e0.elapsed_time(e1) / 5, 3)
        out["d%d_sum" % d] = float(res.double().sum())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
