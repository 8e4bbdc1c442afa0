"""Worker of tests/test_gpu_parity.py::test_exact_operation_order_build — run with SAPCU_LIB_PATH pointing at
csrc/libsapcu_hip_exact.so (-DSAPCU_LIF_EXACT_ORDER: every neuron update in the reference's operation order, op for op):
the neuron unit against the reference vectors, fn and fd forwards against the oracle, and the fused fd encoder against the per-stage
kernels of the same build, bit for bit."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    from sapcu_amd import _lib
    import gpu_utils as U
    from conftest import golden
    from flip_sources import measure          # noqa: F401  (same model builders)
    import sapcu_amd
    from sapcu_amd import testing as T
    from oracle import snn_path as O
    assert os.path.basename(_lib.LIB_PATH) == "libsapcu_hip_exact.so", _lib.LIB_PATH
    lib = _lib.load()
    dev = U.dev()
    # neuron unit: the exact-order arithmetic is the reference's sequence of separately rounded operations
    g = golden("neuron_unit.npz")
    x = torch.from_numpy(g["x"]).to(dev)
    raw = [torch.from_numpy(r).to(dev) for r in g["raw_params"]]
    rows, ch = g["x"].shape
    worst = 0.0
    for kind in ("lif", "eif"):
        for Tn in (1, 4, 7):
            outs = [torch.empty_like(x) for _ in range(4)]
            dT, rh = (raw[4], raw[5]) if kind == "eif" else (None, None)
            _lib.check(lib.sapcu_neuron_selfloop(_lib.ptr(x), rows, ch, Tn, _lib.ptr(raw[0]), _lib.ptr(raw[1]), _lib.ptr(raw[2]),
                                                 _lib.ptr(raw[3]), _lib.ptr(dT), _lib.ptr(rh), *[_lib.ptr(o) for o in outs],
                                                 _lib.current_stream()))
            ref = g["%s_T%d_spikes" % (kind, Tn)]
            worst = max(worst, float(np.abs(outs[0].cpu().numpy() - ref).max()))
    assert worst <= 1e-6, worst
    # the stepping form fd runs, far outside the +-10 clamp (neuron_wide.npz): the exact-order build keeps the clamp, the gate stays closed
    gw = golden("neuron_wide.npz")
    xw = torch.from_numpy(gw["x"]).to(dev)
    raww = [torch.from_numpy(r).to(dev) for r in gw["raw_params"]]
    rw, cw = gw["x"].shape
    for kind in ("lif", "eif"):
        Tn = gw[kind + "_spikes"].shape[0]
        for pairv in (0, 1):
            spk = torch.empty((Tn, rw, cw), device=dev)
            gate = torch.zeros(1, dtype=torch.int32, device=dev)
            dT, rh = (raww[4], raww[5]) if kind == "eif" else (None, None)
            _lib.check(lib.sapcu_neuron_drive(_lib.ptr(xw), rw, cw, Tn, _lib.ptr(raww[0]), _lib.ptr(raww[1]), _lib.ptr(raww[2]), _lib.ptr(raww[3]),
                                              _lib.ptr(dT), _lib.ptr(rh), pairv, _lib.ptr(spk), None, None, None, _lib.ptr(gate), _lib.current_stream()))
            np.testing.assert_allclose(spk.cpu().numpy(), gw[kind + "_spikes"], rtol=2e-5, atol=1e-6)
            assert int(gate.item()) == 0

    class W:
        def __call__(self, kind, **over):
            from conftest import FD_KW, FN_KW
            if kind == "fn":
                m = sapcu_amd.ImprovedSNNNormalEstimation(**dict(FN_KW, **over))
                bn = dict(golden("bn_calib_fn.npz"))
            else:
                m = sapcu_amd.EnhancedSNNDistanceEstimation(**dict(FD_KW, **over))
                bn = dict(golden("bn_calib_fd.npz"))
            return T.conditioned_state_dict(m.state_dict(), 0, bn_stats=bn)

    fn, fd, sdn, sdd = U.build_gpu_models(W())
    fn.knn_cache_mode = "fresh"
    patch = U.sphere_patches(32, 48, skip=100)
    with torch.no_grad():
        n_ref = O.fn_forward(sdn, patch, U.FN_HP)
    e_n = float((fn(patch.to(dev)).cpu() - n_ref).abs().max())
    d_gpu, d_forced, _, flips, _ = U.fd_forward_forced(fd, sdd, patch)
    e_d = float((d_gpu - d_forced).abs().max())
    assert e_n <= 1e-4 and e_d <= 1e-4, (e_n, e_d)
    assert fd.fused_blocks(48) == 1
    os.environ["SAPCU_FD_FUSED"] = "0"
    _, fd_stage, _, _ = U.build_gpu_models(W())
    fd_stage._engine()
    del os.environ["SAPCU_FD_FUSED"]
    assert fd_stage.fused_blocks(48) == 0
    same = bool(torch.equal(fd(patch.to(dev)), fd_stage(patch.to(dev))))
    assert same, "fused fd encoder differs from the per-stage kernels in the exact-order build"
    assert fd.gate_violations() == 0 and fd_stage.gate_violations() == 0
    print("EXACT_ORDER_OK neuron %.2e normals %.2e distances %.2e flipped_rows %d" % (worst, e_n, e_d, int(sum(f.sum() for f in flips))), flush=True)


if __name__ == "__main__":
    main()
