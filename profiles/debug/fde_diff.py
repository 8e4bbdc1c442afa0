#!/usr/bin/env python3
"""Debug aid: where do the fused fd encoder's spikes differ from the per-stage path's?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import gpu_utils as U  # noqa: E402


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
    nq, m, T = 16, 48, 4
    patch = U.sphere_patches(nq, m, skip=1700).to(dev)
    fused, stage = build({"SAPCU_FD_FUSED": "1"}), build({"SAPCU_FD_FUSED": "0"})
    z = lambda *s: torch.full(s, float("nan"), device=dev)
    taps = [{"fused0": z(nq, m, 64), "spikes": z(T, nq, m, 960), "knn": torch.full((3, nq, m, 32), -1, dtype=torch.int32, device=dev),
             "pooled": z(T, nq, 768), "enc": z(nq, 768), "x0": z(nq, m, 960)} for _ in range(2)]
    fused(patch, taps=taps[0])
    stage(patch, taps=taps[1])
    torch.cuda.synchronize()
    coff = (0, 64, 192, 448, 960)
    if os.environ.get("FE_DEBUG_PANEL"):
        for l in range(3):
            pa = taps[0]["x0"][:, :, coff[l]:coff[l + 1]].cpu().numpy()
            sb = taps[1]["spikes"][0, :, :, coff[l]:coff[l + 1]].cpu().numpy()
            sa = taps[0]["spikes"][0, :, :, coff[l]:coff[l + 1]].cpu().numpy()
            print("panel source of block %d vs per-stage spikes t=0: differing %d of %d; vs fused emit: %d" % (l, int((pa != sb).sum()), pa.size, int((pa != sa).sum())))
    if os.environ.get("FE_DEBUG_AB"):
        from sapcu_amd import _lib, packing
        lib = _lib.load()
        blob, directory = packing.pack_fd({k: v.cpu() for k, v in fused.state_dict().items()}, 4)
        W = torch.from_numpy(blob[directory[5]:directory[5] + 256 * 64].reshape(256, 64).copy()).to(dev)
        F = taps[1]["spikes"][0, :, :, 0:64].reshape(nq * m, 64).contiguous()
        outs = {}
        for name, asplit in (("f32A", 0), ("ring", 2), ("auto", 1)):
            C = torch.full((nq * m, 256), float("nan"), device=dev)
            ws = torch.zeros(4 * 256 * 64 + 16, dtype=torch.uint8, device=dev)
            A = F
            if asplit:
                A = torch.empty_like(F)
                _lib.check(lib.sapcu_to_split_rows(_lib.ptr(F), nq * m, 64, 64, _lib.ptr(A), 64, _lib.current_stream()))
            _lib.check(lib.sapcu_gemm_f32(_lib.ptr(A), nq * m, 64, 64, _lib.ptr(W), 256, None, None, 0, _lib.ptr(C), 256, _lib.ptr(ws), asplit, 0, _lib.current_stream()))
            torch.cuda.synchronize()
            outs[name] = C.cpu().numpy().reshape(nq, m, 256)
        mine = np.concatenate([taps[0]["x0"][:, :, 192:320].cpu().numpy(), taps[0]["x0"][:, :, 320:448].cpu().numpy()], -1)
        ref64 = (F.cpu().numpy().astype(np.float64) @ W.cpu().numpy().astype(np.float64).T).reshape(nq, m, 256)
        for name, o in outs.items():
            d = mine != o
            print("GEMM block 1: fused vs %s: differing %d of %d (A' %d, B %d); |fused-ref64| %.3g |%s-ref64| %.3g" % (
                name, int(d.sum()), d.size, int(d[..., :128].sum()), int(d[..., 128:].sum()), np.abs(mine - ref64).mean(), name, np.abs(o - ref64).mean()))
            if d.any():
                cols = np.argwhere(d)[:, 2]
                print("   columns:", dict(zip(*np.unique(cols, return_counts=True))))
        print("ring vs f32A differing:", int((outs["ring"] != outs["f32A"]).sum()))
    for l in range(4):
        if l:
            print("knn%d equal:" % l, bool(torch.equal(taps[0]["knn"][l - 1], taps[1]["knn"][l - 1])))
        xa, xb = taps[0]["x0"][:, :, coff[l]:coff[l + 1]].cpu().numpy(), taps[1]["x0"][:, :, coff[l]:coff[l + 1]].cpu().numpy()
        dx = xa != xb
        print("block %d x0: differing %.5f of elements, max ulp %d, nan %d/%d" % (l, dx.mean(), np.abs(xa.view(np.int32).astype(np.int64) - xb.view(np.int32).astype(np.int64)).max(), int(np.isnan(xa).sum()), int(np.isnan(xb).sum())))
        if l and dx.any():
            sa, sb = taps[0]["spikes"][0, :, :, coff[l]:coff[l + 1]].cpu().numpy(), taps[1]["spikes"][0, :, :, coff[l]:coff[l + 1]].cpu().numpy()
            print("   spikes(t=0) differ where x0 equal: %d; x0 differ: %d; both: %d" % (int(((sa != sb) & ~dx).sum()), int(dx.sum()), int(((sa != sb) & dx).sum())))
            ii = np.argwhere(dx)[:5]
            for (q, i, c) in ii:
                print("   patch %d row %d ch %d: x0 fused %.9g stage %.9g" % (q, i, c, xa[q, i, c], xb[q, i, c]))
        for t in range(T):
            a = taps[0]["spikes"][t, :, :, coff[l]:coff[l + 1]].cpu().numpy()
            b = taps[1]["spikes"][t, :, :, coff[l]:coff[l + 1]].cpu().numpy()
            d = a != b
            ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
            print("block %d t %d: differing %.4f of elements, max ulp %d, per-channel fraction min %.3f max %.3f, per-row min %.3f max %.3f, nan %d"
                  % (l, t, d.mean(), ulp.max(), d.mean((0, 1)).min(), d.mean((0, 1)).max(), d.mean((0, 2)).min(), d.mean((0, 2)).max(), int(np.isnan(a).sum())))
    # f64 reference of block 1's x0 from the (identical) block-0 spikes: which of the two is closer?
    from sapcu_amd import packing
    sd = {k: v for k, v in fused.state_dict().items()}
    blob, directory = packing.pack_fd({k: v.cpu() for k, v in sd.items()}, 4)
    W = blob[directory[5]:directory[5] + 256 * 64].reshape(256, 64).astype(np.float64)
    sh = blob[directory[6]:directory[6] + 128].astype(np.float64)
    S0 = taps[1]["spikes"][0, :, :, 0:64].cpu().numpy().astype(np.float64)          # [nq, m, 64]
    AB = S0 @ W.T                                                                   # [nq, m, 256]
    knn = taps[1]["knn"][0].cpu().numpy()                                           # [nq, m, 32]
    A = AB[:, :, :128]
    mx = np.stack([A[q][knn[q]].max(1) for q in range(nq)])                         # [nq, m, 128]
    pre = mx - AB[:, :, 128:] + sh
    ref = np.where(pre >= 0, pre, 0.2 * pre)
    xa, xb = taps[0]["x0"][:, :, 64:192].cpu().numpy().astype(np.float64), taps[1]["x0"][:, :, 64:192].cpu().numpy().astype(np.float64)
    dx = xa != xb
    print("block 1 x0 vs f64: |fused - ref| mean %.3g max %.3g; |stage - ref| mean %.3g max %.3g (all elements)" % (
        np.abs(xa - ref).mean(), np.abs(xa - ref).max(), np.abs(xb - ref).mean(), np.abs(xb - ref).max()))
    print("   on the %d differing elements: |fused - ref| mean %.3g, |stage - ref| mean %.3g; fused closer in %d" % (
        int(dx.sum()), np.abs(xa - ref)[dx].mean(), np.abs(xb - ref)[dx].mean(), int((np.abs(xa - ref)[dx] < np.abs(xb - ref)[dx]).sum())))
    ch = np.argwhere(dx)[:, 2]
    print("   differing channels (count):", dict(zip(*np.unique(ch, return_counts=True))))
    wmax = np.abs(W).max(1)
    for c in np.unique(ch)[:8]:
        print("   ch %d: max|w| A' row %.4f  B row %.4f; min nonzero |w| A' %.3g B %.3g" % (c, wmax[c], wmax[128 + c], np.abs(W[c])[np.abs(W[c]) > 0].min(), np.abs(W[128 + c])[np.abs(W[128 + c]) > 0].min()))
    print("   all channels: median max|w| %.4f, max %.4f" % (np.median(wmax), wmax.max()))
    a, b = taps[0]["pooled"].cpu().numpy(), taps[1]["pooled"].cpu().numpy()
    print("pooled: differing %.4f, max abs %.3g" % ((a != b).mean(), np.nanmax(np.abs(a - b))))


if __name__ == "__main__":
    main()
