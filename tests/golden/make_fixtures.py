#!/usr/bin/env python3
"""Generate the golden vectors in this directory by RUNNING THE REAL REFERENCE (CPU).

Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py [--skip-e2e]
    ... --only-suite | --suite-shape NAME   shape_suite.npz: BASELINE config 3 stand-in, six 2048-point shapes through the
                                            reference's Generator3D6.upsample (~15 min of CPU)
    ... --only-scale16                      scale16.npz: config 4 stand-in, the 16x generate.py body on a 256-point cloud (~11 min)
    ... --only-patch-knn                    patch_knn.npz incl. the score matrices the reference ranked

What is committed is data: inputs, expected outputs and per-stage intermediates produced by
``fn.snn_coder`` / ``fd.snn_coder`` / ``generation`` imported from /root/reference, plus
BatchNorm statistics calibrated through the reference (see sapcu_amd/testing.py for why).
No reference source text is stored.  Weights are NOT stored: every box rebuilds them from
``testing.conditioned_state_dict(shapes, seed=0, bn_stats=<bn_calib_*.npz>)``.
"""
import argparse
import zlib
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.modules.setdefault("trimesh", types.ModuleType("trimesh"))   # imported, never used (generation.py:21)

import sapcu_amd  # noqa: E402
from sapcu_amd import testing as T  # noqa: E402
from oracle import geom_path as G  # noqa: E402  (only for building input patches)

from fn import snn_coder as ref_fn  # noqa: E402
from fd import snn_coder as ref_fd  # noqa: E402
import generation as ref_gen  # noqa: E402

FN_KW = dict(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=4, time_steps_dec=12, num_heads=8,
             use_snn_decoder=False, decoder_dropout=0.1)
FD_KW = dict(k=32, emb_dims=768, time_steps_enc=4, time_steps_dec=8, num_heads=8, dropout=0.1,
             use_snn_decoder=False, k_scales=[8, 16, 32, 48])


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote %-24s %8.1f KiB" % (name, os.path.getsize(path) / 1024))


def npy(t):
    return t.detach().cpu().numpy()


def calibrate_bn(model, x):
    """One eval forward with pre-hooks that set each BatchNorm's running stats to the statistics of
    the activations it sees on first use (execution order => upstream layers already calibrated)."""
    seen, hooks = set(), []

    def mk(name):
        def hook(mod, inp):
            if name in seen:
                return
            seen.add(name)
            a = inp[0]
            dims = [d for d in range(a.dim()) if d != 1]
            mod.running_mean.copy_(a.mean(dims))
            mod.running_var.copy_(a.var(dims, unbiased=False).clamp_min(1e-8))
        return hook

    for name, mod in model.named_modules():
        if isinstance(mod, (nn.BatchNorm1d, nn.BatchNorm2d)):
            hooks.append(mod.register_forward_pre_hook(mk(name)))
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    return {k: npy(v) for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}


def build_models(bn_fn=None, bn_fd=None, fn_kw=None, fd_kw=None):
    fn = ref_fn.ImprovedSNNNormalEstimation(**(fn_kw or FN_KW)).eval()
    fd = ref_fd.EnhancedSNNDistanceEstimation(**(fd_kw or FD_KW)).eval()
    fn.load_state_dict(T.conditioned_state_dict(fn.state_dict(), 0, bn_stats=bn_fn), strict=True)
    fd.load_state_dict(T.conditioned_state_dict(fd.state_dict(), 0, bn_stats=bn_fd), strict=True)
    return fn, fd


def sphere_patches(nq, k, n=5000, qseed=0):
    cloud = T.sphere_cloud(n, 0)
    q = T.grid_queries(nq, qseed)
    idx = G.knn_bruteforce(cloud, q, k)
    return torch.from_numpy(G.gather_centre(cloud, q, idx)).float()


def hook_outputs(model, names):
    store, hooks = {}, []
    mods = dict(model.named_modules())
    for n in names:
        def mk(n):
            def hook(mod, inp, out):
                store.setdefault(n, []).append(out[0] if isinstance(out, tuple) else out)
            return hook
        hooks.append(mods[n].register_forward_hook(mk(n)))
    return store, hooks


def run_reference_dense(cloud, cell):
    """Run the reference's own seed generator (dense.cpp compiled by oracle/Makefile) the way generation.py:114-118
    does — in a scratch directory, through its text files."""
    dense = os.path.join(ROOT, "oracle", "_ref", "dense")
    if not os.path.exists(dense):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    work = tempfile.mkdtemp(prefix="sapcu_dense_")
    try:
        np.savetxt(os.path.join(work, "test.xyz"), cloud, fmt="%.6f")
        subprocess.check_call([dense, str(cell), str(cloud.shape[0])], cwd=work)
        tgt = os.path.join(work, "target.xyz")
        if os.path.getsize(tgt) == 0:
            return np.zeros((0, 3))
        return np.loadtxt(tgt, ndmin=2)[:, 0:3]
    finally:
        shutil.rmtree(work, ignore_errors=True)


def seed_fixture():
    """Golden outputs of the reference seed generator: small cases in full, the full-size case (sphere N=5000, cell
    0.004: ~385 k seeds) as count + SHA-256 of the float64 bytes + head/tail rows."""
    import hashlib
    out = {}
    for name, cloud, cell in (("sphere2048_c030", T.sphere_cloud(2048, 0), 0.03), ("torus2048_c020", T.analytic_cloud("torus", 2048, 1), 0.02),
                              ("cube300_c050", T.analytic_cloud("cube", 300, 2), 0.05), ("tiny7_c050", T.sphere_cloud(7, 3), 0.05)):
        seeds = run_reference_dense(cloud, cell)
        out[name] = seeds.reshape(-1, 3)
        out[name + "_cell"] = np.float64(cell)
    big = run_reference_dense(T.sphere_cloud(5000, 0), 0.004)
    out["sphere5000_c004_count"] = np.int64(big.shape[0])
    out["sphere5000_c004_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(big, dtype=np.float64).tobytes()).hexdigest())
    out["sphere5000_c004_head"] = big[:64]
    out["sphere5000_c004_tail"] = big[-64:]
    save("dense_seeds.npz", **out)


def fps_fixture():
    """Golden outputs of the reference's own ``normalize_pointcloud`` / ``farthest_point_sample`` (generate.py:43-74).
    generate.py imports h5py (absent here, never used on this path) and hard-codes device 'cuda' (no GPU in this
    container): the module is imported with an empty h5py stand-in and, for the duration of the calls, tensors
    asked to move to 'cuda' stay on the CPU — the arithmetic is torch's own either way."""
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    import generate as ref_generate
    real_to = torch.Tensor.to

    def to_cpu(self, *a, **kw):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        return real_to(self, *a, **kw)

    cases = {name: T.fps_case(name) for name in T.FPS_CASES}
    out = {"names": np.array(list(cases))}
    torch.Tensor.to = to_cpu
    try:
        for name, (cloud, npoint) in cases.items():
            idx = ref_generate.farthest_point_sample(cloud, npoint)
            out[name + "_idx"] = idx.astype(np.int64)
            out[name + "_npoint"] = np.int64(npoint)
            print("fps", name, cloud.shape, npoint, idx[:6])
        c, loc, scale = ref_generate.normalize_pointcloud(cases["sphere2048"][0])
        out["norm_cloud"], out["norm_loc"], out["norm_scale"] = c, loc, np.float64(scale)
        out["norm_in"] = cases["sphere2048"][0]
        flat = np.tile(np.array([[1.0, 2.0, 3.0]]), (4, 1))          # zero extent: scale_inv falls back to 1
        c, loc, scale = ref_generate.normalize_pointcloud(flat)
        out["flat_in"], out["flat_cloud"], out["flat_loc"], out["flat_scale"] = flat, c, loc, np.float64(scale)
    finally:
        torch.Tensor.to = real_to
    save("fps.npz", **out)


def neuron_train_fixture():
    """The reference's own LIF neuron IN TRAINING MODE (hard spikes + straight-through surrogate gradient), self-feeding for
    T steps as fn drives it (fn/snn_coder.py:318-320): forward spikes and the gradients of sum(spikes * g) w.r.t. the input
    and the four raw per-channel parameters, for 2-D, 3-D and 4-D inputs; parameters partly outside their clamp ranges."""
    rng = np.random.default_rng(11)
    out = {}
    for tag, shape, T in (("2d", (96, 40), 4), ("3d", (5, 24, 17), 4), ("4d", (3, 16, 7, 5), 6), ("t1", (33, 8), 1)):
        C = shape[1]
        x = torch.tensor(rng.normal(0.6, 1.0, shape).astype(np.float32), requires_grad=True)
        g = torch.tensor(rng.normal(0.0, 1.0, shape).astype(np.float32))
        raw = np.stack([rng.uniform(0.05, 1.1, C), rng.uniform(-0.02, 0.15, C), rng.uniform(0.05, 1.0, C),
                        rng.normal(0.7, 0.4, C)]).astype(np.float32)
        neuron = ref_fn.MultiTimeConstantLIFNeuron(C)
        with torch.no_grad():
            neuron.membrane_decay.copy_(torch.from_numpy(raw[0]))
            neuron.threshold_adapt.copy_(torch.from_numpy(raw[1]))
            neuron.refractory_decay.copy_(torch.from_numpy(raw[2]))
            neuron.threshold_base.copy_(torch.from_numpy(raw[3]))
        neuron.train()
        v, st = x, [None, None, None]
        for _ in range(T):
            v, *st = neuron(v, *st)
        (v * g).sum().backward()
        def gnp(p):                      # a parameter the loss does not depend on (T = 1) has no gradient: zeros
            return npy(p.grad) if p.grad is not None else np.zeros(C, np.float32)
        out.update({tag + "_x": npy(x), tag + "_g": npy(g), tag + "_raw": raw, tag + "_T": np.int64(T), tag + "_spikes": npy(v),
                    tag + "_gx": npy(x.grad), tag + "_gmd": gnp(neuron.membrane_decay),
                    tag + "_gta": gnp(neuron.threshold_adapt), tag + "_grd": gnp(neuron.refractory_decay),
                    tag + "_gtb": gnp(neuron.threshold_base)})
        print("neuron_train", tag, shape, "spike rate %.3f" % float(v.mean()), "|gx| %.3g" % float(x.grad.abs().mean()))
    out["tags"] = np.array(["2d", "3d", "4d", "t1"])
    # the reference's layer in training mode: Conv1d(64, 128, 1) + BatchNorm1d(128) + LIF x 4 (fn/snn_coder.py:225-229, 317-320)
    B, N, cin, cout = 6, 40, 64, 128
    conv, bn, neuron = nn.Conv1d(cin, cout, 1), nn.BatchNorm1d(cout), ref_fn.MultiTimeConstantLIFNeuron(cout)
    with torch.no_grad():
        bn.weight.copy_(torch.tensor(rng.uniform(0.5, 1.5, cout).astype(np.float32)))
        bn.bias.copy_(torch.tensor(rng.normal(0.4, 0.5, cout).astype(np.float32)))
        neuron.membrane_decay.copy_(torch.tensor(rng.uniform(0.05, 1.1, cout).astype(np.float32)))
        neuron.threshold_adapt.copy_(torch.tensor(rng.uniform(-0.02, 0.15, cout).astype(np.float32)))
        neuron.refractory_decay.copy_(torch.tensor(rng.uniform(0.05, 1.0, cout).astype(np.float32)))
        neuron.threshold_base.copy_(torch.tensor(rng.normal(0.7, 0.4, cout).astype(np.float32)))
    for mod in (conv, bn, neuron):
        mod.train()
    x = torch.tensor(rng.normal(0.0, 1.0, (B, cin, N)).astype(np.float32), requires_grad=True)
    g = torch.tensor(rng.normal(0.0, 1.0, (B, cout, N)).astype(np.float32))
    v, st = bn(conv(x)), [None, None, None]
    for _ in range(4):
        v, *st = neuron(v, *st)
    (v * g).sum().backward()
    out.update({"layer_x": npy(x), "layer_g": npy(g), "layer_spikes": npy(v), "layer_gx": npy(x.grad),
                "layer_w": npy(conv.weight), "layer_b": npy(conv.bias), "layer_gamma": npy(bn.weight), "layer_beta": npy(bn.bias),
                "layer_raw": np.stack([npy(neuron.membrane_decay), npy(neuron.threshold_adapt), npy(neuron.refractory_decay),
                                       npy(neuron.threshold_base)]),
                "layer_gw": npy(conv.weight.grad), "layer_gb": npy(conv.bias.grad), "layer_ggamma": npy(bn.weight.grad),
                "layer_gbeta": npy(bn.bias.grad),
                "layer_graw": np.stack([npy(neuron.membrane_decay.grad), npy(neuron.threshold_adapt.grad),
                                        npy(neuron.refractory_decay.grad), npy(neuron.threshold_base.grad)]),
                "layer_running_mean": npy(bn.running_mean), "layer_running_var": npy(bn.running_var)})
    print("layer", tuple(x.shape), "->", tuple(v.shape), "spike rate %.3f" % float(v.mean()))
    save("neuron_train.npz", **out)
    block_train_fixture()


def block_train_fixture():
    """One MultiHeadSNNTransformerBlock of the reference IN TRAINING MODE (fn/snn_coder.py:294-396; dropout 0 — the
    attention dropout is random): forward and the autograd gradients of sum(out * g) w.r.t. the input features and every
    parameter.  BatchNorm affine and neuron parameters are randomised so that the hard spikes are not degenerate."""
    rng = np.random.default_rng(23)
    torch.manual_seed(23)
    B, N, k, dp, dm, H = 2, 24, 8, 64, 128, 8
    blk = ref_fn.MultiHeadSNNTransformerBlock(dp, dm, k, 4, num_heads=H, dropout=0.0)
    with torch.no_grad():
        for name, prm in blk.named_parameters():
            if name.endswith(".1.weight"):                                  # BatchNorm gamma
                prm.copy_(torch.tensor(rng.uniform(0.6, 1.4, prm.shape).astype(np.float32)))
            elif name.endswith(".1.bias"):
                prm.copy_(torch.tensor(rng.normal(0.5, 0.4, prm.shape).astype(np.float32)))
            elif name.endswith("membrane_decay"):
                prm.copy_(torch.tensor(rng.uniform(0.3, 0.95, prm.shape).astype(np.float32)))
            elif name.endswith("threshold_adapt"):
                prm.copy_(torch.tensor(rng.uniform(0.005, 0.08, prm.shape).astype(np.float32)))
            elif name.endswith("refractory_decay"):
                prm.copy_(torch.tensor(rng.uniform(0.2, 0.9, prm.shape).astype(np.float32)))
            elif name.endswith("threshold_base"):
                prm.copy_(torch.tensor(rng.normal(0.6, 0.3, prm.shape).astype(np.float32)))
    blk.train()
    xyz = torch.tensor((rng.normal(0, 0.05, (B, N, 3))).astype(np.float32))
    feats = torch.tensor(rng.normal(0, 1, (B, N, dp)).astype(np.float32), requires_grad=True)
    g = torch.tensor(rng.normal(0, 1, (B, N, dp)).astype(np.float32))
    res, _ = blk(xyz, feats)
    (res * g).sum().backward()
    knn_idx = blk.knn_cache.get_knn(xyz, k, "block_%d" % id(blk))
    out = {"xyz": npy(xyz), "features": npy(feats), "g": npy(g), "out": npy(res), "g_features": npy(feats.grad),
           "knn_idx": npy(knn_idx).astype(np.int32), "names": np.array([n for n, _ in blk.named_parameters()])}
    for n, prm in blk.named_parameters():
        out["p:" + n] = npy(prm)
        out["g:" + n] = npy(prm.grad) if prm.grad is not None else np.zeros(tuple(prm.shape), np.float32)
    print("block", tuple(res.shape), "|out| %.3f" % float(res.abs().mean()), "|g_feat| %.3g" % float(feats.grad.abs().mean()))
    save("block_train.npz", **out)


def fn_train_fixture():
    """One TRAINING step of the reference's whole fn model (fn/snn_coder.py:627-699 in train() mode, every nn.Dropout set to
    p=0 because dropout is random): unit normals, enhanced_angular_loss_with_consistency (fn:587-625, xyz passed as the
    reference's trainer does) and the autograd gradients of the loss.  Parameters are testing.training_state_dict(seed 5), so the
    fixture carries only inputs, outputs and gradients; gradients of tensors above 8192 elements are stored as a seeded
    sample of 2048 entries plus their L2 norm."""
    rng = np.random.default_rng(31)
    torch.manual_seed(31)
    B, N = 6, 32
    model = ref_fn.ImprovedSNNNormalEstimation(**FN_KW)
    model.load_state_dict(T.training_state_dict(model.state_dict(), 5), strict=True)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model.train()
    d = rng.normal(size=(B, N, 3))
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    pts = torch.tensor((d * rng.uniform(0.2, 1.0, (B, N, 1)) * np.array([1.0, 1.0, 0.15])).astype(np.float32))
    gt = rng.normal(size=(B, 3))
    gt = torch.tensor((gt / np.linalg.norm(gt, axis=1, keepdims=True)).astype(np.float32))
    normals = model(pts)
    loss, conf = ref_fn.enhanced_angular_loss_with_consistency(normals, gt, xyz=pts)
    loss.backward()
    enc = model.encoder
    out = {"points": npy(pts), "gt": npy(gt), "normals": npy(normals), "loss": np.float64(loss.item()), "seed": np.int64(5)}
    for i, blk in enumerate((enc.trans1, enc.trans2, enc.trans3)):
        out["knn%d" % i] = npy(blk.knn_cache.get_knn(pts, min(blk.k, N), "block_%d" % id(blk))).astype(np.int32)
    names = []
    for n, prm in model.named_parameters():
        g = npy(prm.grad).ravel() if prm.grad is not None else np.zeros(prm.numel(), np.float32)
        names.append(n)
        if g.size <= 8192:
            out["g:" + n] = g.reshape(tuple(prm.shape))
        else:
            sel = np.sort(np.random.default_rng(zlib.crc32(n.encode())).choice(g.size, 2048, replace=False))
            out["gi:" + n] = sel.astype(np.int64)
            out["gs:" + n] = g[sel]
            out["gn:" + n] = np.float64(np.linalg.norm(g.astype(np.float64)))
    out["names"] = np.array(names)
    print("fn train step: loss %.6f  normals[0] %s" % (loss.item(), npy(normals)[0]))
    save("fn_train.npz", **out)


def fn_trainer_fixture():
    """One ``Trainer.train_step`` of the reference (fn/trainer.py:41-148) on a 4-D batch [B, patches, points, 3] — the shape
    its data loader yields — with plain SGD (lr 1e-3) and grad_clip 0.15 ('norm'), dropout off: the returned loss and
    confidence, every BatchNorm running statistic after the step, and the parameter UPDATE (new - old; tensors above 1024
    elements as a seeded sample of 256 entries plus the L2 norm of the whole update).  Exercises the 4-D path, the
    consistency term of the loss (fn:557-583), global-norm clipping and the BatchNorm bookkeeping."""
    from fn import trainer as ref_trainer
    from oracle import train_path as TP
    B, NP, M = 2, 8, 12
    # Hard spikes make the forward discontinuous: an input whose pre-activation lies within f32 rounding of a threshold gives
    # different spikes under ANY reordering of the sums (torch CPU vs GPU included).  Take the first input seed for which
    # the oracle's differently-ordered f32 arithmetic reproduces the reference's normals, i.e. one away from every threshold.
    for data_seed in range(41, 80):
        rng = np.random.default_rng(data_seed)
        torch.manual_seed(data_seed)
        model = ref_fn.ImprovedSNNNormalEstimation(**FN_KW)
        model.load_state_dict(T.training_state_dict(model.state_dict(), 7), strict=True)
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        centres = rng.normal(size=(B, NP, 1, 3)) * 0.4
        pts = torch.tensor((centres + rng.normal(size=(B, NP, M, 3)) * np.array([0.08, 0.08, 0.01])).astype(np.float32))
        gt = torch.tensor(rng.normal(size=(B, NP, 3)).astype(np.float32))
        model.train()
        with torch.no_grad():
            flat = pts.reshape(B * NP, M, 3)
            d2 = ((flat[:, :, None, :] - flat[:, None, :, :]) ** 2).sum(-1)
            knn = [d2.topk(min(k, M), dim=-1, largest=False)[1] for k in FN_KW["k_values"]]
            mine = TP.fn_train_forward({n: v.detach() for n, v in model.named_parameters()}, flat, knn)
            # (the probe forward below must not leave its BatchNorm statistics behind: rebuild the model afterwards)
            theirs = model(pts).reshape(B * NP, 3)
        gap = float((mine - theirs).abs().max())
        print("data seed %d: oracle vs reference normals %.2e" % (data_seed, gap))
        if gap < 1e-5:
            break
    else:
        raise RuntimeError("no well-conditioned input found")
    model = ref_fn.ImprovedSNNNormalEstimation(**FN_KW)
    model.load_state_dict(T.training_state_dict(model.state_dict(), 7), strict=True)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    old = {n: prm.detach().clone() for n, prm in model.named_parameters()}
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    tr = ref_trainer.Trainer(model, opt, device=torch.device("cpu"), grad_clip=0.15, grad_clip_type="norm")
    loss_value, loss_dict = tr.train_step({"input": pts, "normal": gt})
    assert loss_value is not None
    out = {"points": npy(pts), "gt": npy(gt), "loss": np.float64(loss_value), "confidence": np.float64(loss_dict["confidence"]),
           "seed": np.int64(7), "lr": np.float64(1e-3), "grad_clip": np.float64(0.15)}
    names = []
    for n, prm in model.named_parameters():
        d = (prm.detach() - old[n]).numpy().ravel()
        names.append(n)
        if d.size <= 1024:
            out["d:" + n] = d.reshape(tuple(prm.shape))
        else:
            sel = np.sort(np.random.default_rng(zlib.crc32(n.encode())).choice(d.size, 256, replace=False))
            out["di:" + n] = sel.astype(np.int64)
            out["ds:" + n] = d[sel]
            out["dn:" + n] = np.float64(np.linalg.norm(d.astype(np.float64)))
    out["names"] = np.array(names)
    bufs = []
    for n, b in model.named_buffers():
        if n.endswith("running_mean") or n.endswith("running_var") or n.endswith("num_batches_tracked"):
            out["b:" + n] = npy(b)
            bufs.append(n)
    out["buffers"] = np.array(bufs)
    print("fn trainer step: loss %.6f confidence %.4f" % (loss_value, loss_dict["confidence"]))
    save("fn_trainer.npz", **out)


def patch_knn_fixture(test):
    """Reference knn() (fn/snn_coder.py:31-39 == fd/snn_coder.py:25-32) on xyz patches and on soft-spike-like features:
    the neighbour indices AND the score matrix the reference ranked (captured from inside its own call: the tensor it hands to
    topk), for C in {3, 64, 128, 256} (SURVEY.md 8c fixture 2)."""
    out = {}
    feats = {3: test[:4].permute(0, 2, 1).contiguous()}
    for c in (64, 128, 256):
        feats[c] = torch.from_numpy(np.random.default_rng(c).random((4, c, 48)).astype(np.float32))
    real_topk = torch.Tensor.topk
    seen = []

    def rec_topk(self, *a, **kw):
        seen.append(self.detach().clone())
        return real_topk(self, *a, **kw)

    for c, f in feats.items():
        out["feat_c%d" % c] = npy(f)
        with torch.no_grad():
            for k in (8, 12, 16, 18, 24, 32, 48):
                torch.Tensor.topk = rec_topk
                try:
                    idx = ref_fn.knn(f, k)
                finally:
                    torch.Tensor.topk = real_topk
                out["idx_c%d_k%d" % (c, k)] = npy(idx).astype(np.int8)
                assert torch.equal(idx, ref_fd.knn(f, k))
            assert all(torch.equal(seen[-1], t) for t in seen[-7:])          # one score matrix per feature set, whatever k
            out["score_c%d" % c] = npy(seen[-1])
    save("patch_knn.npz", **out)


def _reference_upsample(fn, fd, cloud, spacing, batch_size=64, k=48):
    """The reference's own Generator3D6.upsample on `cloud` [N,3] (run in a scratch directory beside its own compiled
    dense, which reads test.xyz — generate.py never writes that file): (seeds, unfiltered, filtered)."""
    dense = os.path.join(ROOT, "oracle", "_ref", "dense")
    if not os.path.exists(dense):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    work = tempfile.mkdtemp(prefix="sapcu_suite_")
    shutil.copy(dense, os.path.join(work, "dense"))
    cwd = os.getcwd()
    os.chdir(work)
    RealTree = ref_gen.KDTree
    try:
        np.savetxt("test.xyz", cloud, fmt="%.6f")
        captured = []

        class RecTree(RealTree):                          # records the array each KDTree is built on
            def __init__(self, data, *a, **kw):
                captured.append(np.array(data, copy=True))
                super().__init__(data, *a, **kw)

        ref_gen.KDTree = RecTree
        for b in (fn.encoder.trans1, fn.encoder.trans2, fn.encoder.trans3):
            b.knn_cache.cache.clear()
        gen = ref_gen.Generator3D6(fn, fd, torch.device("cpu"), k_neighbors=k, dense_spacing=spacing, batch_size=batch_size)
        result = np.asarray(gen.upsample(cloud[None]))
        seeds = np.loadtxt("target.xyz", ndmin=2)[:, 0:3]
        return seeds, captured[1], result
    finally:
        ref_gen.KDTree = RealTree
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


def _calibrated_models():
    bn_fn = dict(np.load(os.path.join(HERE, "bn_calib_fn.npz")))
    bn_fd = dict(np.load(os.path.join(HERE, "bn_calib_fd.npz")))
    return build_models(bn_fn, bn_fd)


def shape_suite_fixture(only=None):
    """BASELINE config 3 stand-in: the reference's Generator3D6.upsample on six shapes of 2048 points
    (sphere, torus, cube with sharp edges, cylinder, union of two spheres, the tree's own Icosahedron.xyz subsampled),
    k_neighbors 48, T 4, batch 64, seeds from its own dense.cpp: seeds / unfiltered / filtered clouds per shape."""
    import time
    torch.set_num_threads(8)
    fn, fd = _calibrated_models()
    out = {"names": np.array([s[0] for s in T.SHAPE_SUITE])}
    if only:                                              # redo one shape, keep the others as committed
        out.update(dict(np.load(os.path.join(HERE, "shape_suite.npz"))))
    ico = np.loadtxt(os.path.join(REF, "external", "SPU-PMD", "evaluation_code", "Icosahedron.xyz"))[:, :3]
    for name, kind, seed, spacing in T.SHAPE_SUITE:
        if only and name != only:
            continue
        if kind is None:
            pick = np.sort(np.random.default_rng(seed).permutation(ico.shape[0])[:T.SHAPE_SUITE_N])
            sub = ico[pick]
            lo, hi = sub.min(0), sub.max(0)
            cloud = np.round((sub - (lo + hi) / 2) / (hi - lo).max(), 6)          # bbox-normalised like generate.py:43-54
            out["%s_cloud" % name] = cloud
        else:
            cloud = T.suite_cloud(name)
        t0 = time.time()
        seeds, unf, filt = _reference_upsample(fn, fd, cloud, spacing)
        print("suite %-12s spacing %.3f: %d seeds -> %d kept  (%.0f s)" % (name, spacing, seeds.shape[0], filt.shape[0], time.time() - t0),
              flush=True)
        out["%s_seeds" % name], out["%s_unfiltered" % name], out["%s_filtered" % name] = seeds, unf, filt
        out["%s_spacing" % name] = np.float64(spacing)
    save("shape_suite.npz", **out)


def scale16_fixture():
    """BASELINE config 4 stand-in: arbitrary scale 16x = the body of generate.py:81-99 on a 256-point cloud with a
    non-trivial bounding box: normalize_pointcloud -> Generator3D6.upsample -> denormalise -> farthest_point_sample(16 N)."""
    import time
    torch.set_num_threads(8)
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    import generate as ref_generate
    fn, fd = _calibrated_models()
    c = T.SCALE16_CASE
    raw = T.scale16_cloud()
    cloud, loc, scale = ref_generate.normalize_pointcloud(raw)
    t0 = time.time()
    seeds, unf, filt = _reference_upsample(fn, fd, cloud, c["spacing"])
    print("16x: %d seeds -> %d kept (%.0f s)" % (seeds.shape[0], filt.shape[0], time.time() - t0), flush=True)
    target = c["ratio"] * c["n"]
    assert filt.shape[0] >= target, "the 16x case needs >= %d refined points, got %d: lower SCALE16_CASE spacing" % (target, filt.shape[0])
    up = filt * scale + loc
    real_to = torch.Tensor.to

    def to_cpu(self, *a, **kw):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        return real_to(self, *a, **kw)

    torch.Tensor.to = to_cpu
    try:
        idx = ref_generate.farthest_point_sample(up, target)
    finally:
        torch.Tensor.to = real_to
    save("scale16.npz", seeds=seeds, unfiltered=unf, filtered=filt, norm_cloud=cloud, loc=loc, scale=np.float64(scale),
         fps_idx=idx.astype(np.int64), output=up[idx])


def ref_vs_ref_fixture():
    """The reference against ITSELF: the same inputs with torch.set_num_threads(1) and with 8 threads (oneDNN / OpenBLAS sum in a
    thread-count-dependent order, so fd's feature-space neighbour searches break near-ties differently).  These figures are the
    floor under any parity statement about free-running fd: (a) fd on the 256 patches of test_fd_forward_256_patches...: patches
    whose neighbour sets differ, distances within 2e-4; (b) the end-to-end sphere-2048 run of e2e_upsample.npz (made with the
    container's default 8 threads) repeated with ONE thread: fraction of refined points within 2e-4."""
    fn, fd = _calibrated_models()
    cloud = T.sphere_cloud(5000, 0)
    q = T.grid_queries(256 + 200, 0)[200:]
    idx = G.knn_bruteforce(cloud, q, 48)
    patch = torch.from_numpy(G.gather_centre(cloud, q, idx)).float()
    real_knn = ref_fd.knn
    runs = {}
    for nt in (1, 8):
        torch.set_num_threads(nt)
        tables = []

        def rec(x, k):
            out = real_knn(x, k)
            tables.append(out.detach().clone())
            return out

        ref_fd.knn = rec
        try:
            with torch.no_grad():
                fd.reset_states()
                d = fd(patch)
        finally:
            ref_fd.knn = real_knn
        # per forward: T x (4 xyz scales + 3 feature-space) searches; the feature-space ones of t = 0 are calls 4, 5, 6
        runs[nt] = (npy(d), [npy(t.sort(-1)[0]) for t in tables[4:7]])
    torch.set_num_threads(8)
    (d1, k1), (d8, k8) = runs[1], runs[8]
    flip_rows = [(a != b).any(-1) for a, b in zip(k1, k8)]               # [256, 48] per block
    flip_patch = np.zeros(256, bool)
    for f in flip_rows:
        flip_patch |= f.any(-1)
    fd_stats = dict(fd256_flip_patches=int(flip_patch.sum()), fd256_flip_rows=int(sum(f.sum() for f in flip_rows)),
                    fd256_max_abs_diff=float(np.abs(d1 - d8).max()), fd256_within_2e4=float((np.abs(d1 - d8) <= 2e-4).mean()),
                    fd256_max_abs_diff_flipfree=float(np.abs(d1 - d8)[~flip_patch].max()))
    print("reference 1 thread vs 8 threads, fd on 256 patches:", fd_stats)
    # (b) end to end with one thread against the committed 8-thread run
    g = np.load(os.path.join(HERE, "e2e_upsample.npz"))
    torch.set_num_threads(1)
    try:
        seeds, unfiltered, _ = _reference_upsample(fn, fd, T.sphere_cloud(2048, 0), 0.03)
    finally:
        torch.set_num_threads(8)
    assert np.array_equal(seeds, g["seeds"])
    err = np.abs(unfiltered - g["unfiltered"]).max(axis=1)
    e2e = dict(e2e_within_2e4=float((err <= 2e-4).mean()), e2e_median=float(np.median(err)), e2e_p90=float(np.quantile(err, 0.9)),
               e2e_points=int(err.size))
    print("reference 1 thread vs 8 threads, sphere-2048 end to end:", e2e)
    save("ref_vs_ref.npz", threads=np.array([1, 8]), **{k: np.array(v) for k, v in dict(fd_stats, **e2e).items()})


def neuron_wide_fixture():
    """neuron_wide.npz (round 4): the reference's LIF / EIF neurons (fd classes, fd/snn_coder.py:94-155,198-275) far outside the
    spike function's +-10 clamp — pre-activations out to |x| = 40 and a few at +-100 / +-1e4 — driven the way fd's encoder drives
    them: the SAME input at every step (the refractory gate decides what enters), every step's spikes and the state after the last
    step.  This is the input class where an un-clamped spike surrogate underflows to exactly 0 (x - theta below about -13.2) while
    the reference's clamped one stays at 3.85e-23, i.e. where its gate is closed with r > 0."""
    C, T = 8, 7
    rng = np.random.default_rng(11)
    x = np.concatenate([np.linspace(-40, 40, 321), [-10.0, 10.0, -12.0, 12.0, -13.2, -13.3, -14.3, -14.4, -15.0, 15.0, -20.0, 20.0,
                                                    -100.0, 100.0, -1e4, 1e4, 0.0, 1.0]])
    x = np.tile(x[:, None], (1, C)).astype(np.float32)
    x[:321] += rng.normal(0, 0.05, (321, C)).astype(np.float32)
    raw = np.stack([rng.uniform(0.05, 1.05, C), rng.uniform(0.0, 0.12, C), rng.uniform(0.05, 1.0, C),
                    rng.uniform(0.6, 1.4, C), rng.uniform(0.05, 5.5, C), rng.uniform(0.05, 2.2, C)]).astype(np.float32)
    out = {"x": x, "raw_params": raw}
    for kind, cls in (("lif", ref_fd.MultiTimeConstantLIFNeuron), ("eif", ref_fd.MultiTimeConstantEIFNeuron)):
        neu = cls(C).eval()
        with torch.no_grad():
            neu.membrane_decay.copy_(torch.from_numpy(raw[0])); neu.threshold_adapt.copy_(torch.from_numpy(raw[1]))
            neu.refractory_decay.copy_(torch.from_numpy(raw[2])); neu.threshold_base.copy_(torch.from_numpy(raw[3]))
            if kind == "eif":
                neu.delta_T.copy_(torch.from_numpy(raw[4])); neu.theta_rh.copy_(torch.from_numpy(raw[5]))
            st, spikes, gate_open = [None, None, None], [], 0
            for step in range(T):
                if step > 0:
                    gate_open += int((st[2] <= 0).sum())            # what the reference's own gate sees (fd:133,249)
                v, *st = neu(torch.from_numpy(x), *st)
                spikes.append(npy(v))
        out["%s_spikes" % kind] = np.stack(spikes)                  # [T, rows, C]
        out["%s_membrane" % kind], out["%s_threshold" % kind], out["%s_refractory" % kind] = npy(st[0]), npy(st[1]), npy(st[2])
        out["%s_gate_open" % kind] = np.int64(gate_open)
        print(kind, "reference gate-open events at t >= 1:", gate_open, " min spike", float(np.stack(spikes).min()))
    save("neuron_wide.npz", **out)


def e2e_default_fixture():
    """e2e_default.npz (round 4): the reference's Generator3D6.upsample at the configuration `generate.py` + `config/*.yaml` really
    run — k_neighbors = 100 (generation.py:68; generate.py:135 keeps it), batch_size = 256 (generate.py:135), fn time_steps_enc = 6
    (config/fn.yaml:41), fd time_steps_enc = 7 (config/fd.yaml:47) — on the 2048-point sphere with dense_spacing 0.03 (901 seeds from
    its own dense.cpp; the default spacing 0.004 gives ~60 000 seeds: hours of CPU).  Conditioned weights seed 0 with the T = 4
    BatchNorm calibration (the variants fixture does the same).  ~10 min of CPU."""
    bn_fn, bn_fd = dict(np.load(os.path.join(HERE, "bn_calib_fn.npz"))), dict(np.load(os.path.join(HERE, "bn_calib_fd.npz")))
    fn, fd = build_models(bn_fn, bn_fd, dict(FN_KW, time_steps_enc=6), dict(FD_KW, time_steps_enc=7))
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cloud = T.sphere_cloud(2048, 0)
    with torch.no_grad():
        seeds, unfiltered, filtered = _reference_upsample(fn, fd, cloud, 0.03, batch_size=256, k=100)
    print("e2e at the reference defaults: %d seeds -> %d refined, %d after the outlier filter" % (seeds.shape[0], unfiltered.shape[0], filtered.shape[0]))
    save("e2e_default.npz", seeds=seeds, unfiltered=unfiltered, filtered=filtered)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-ref-vs-ref", action="store_true", help="only (re)generate ref_vs_ref.npz (reference 1 thread vs 8 threads)")
    ap.add_argument("--skip-e2e", action="store_true")
    ap.add_argument("--only-fps", action="store_true", help="only (re)generate fps.npz")
    ap.add_argument("--only-train", action="store_true", help="only (re)generate neuron_train.npz")
    ap.add_argument("--only-fn-train", action="store_true", help="only (re)generate fn_train.npz")
    ap.add_argument("--only-seeds", action="store_true", help="only (re)generate dense_seeds.npz")
    ap.add_argument("--only-patch-knn", action="store_true", help="only (re)generate patch_knn.npz")
    ap.add_argument("--suite-shape", default=None, help="(re)generate ONE shape of shape_suite.npz")
    ap.add_argument("--only-suite", action="store_true", help="only (re)generate shape_suite.npz (BASELINE config 3 stand-in)")
    ap.add_argument("--only-scale16", action="store_true", help="only (re)generate scale16.npz (BASELINE config 4 stand-in)")
    ap.add_argument("--only-neuron-wide", action="store_true", help="only (re)generate neuron_wide.npz (neurons out to |x| = 1e4)")
    ap.add_argument("--only-e2e-default", action="store_true", help="only (re)generate e2e_default.npz (upsample at k = 100, fn T = 6, fd T = 7)")
    args = ap.parse_args()
    if args.only_neuron_wide:
        neuron_wide_fixture()
        return
    if args.only_e2e_default:
        e2e_default_fixture()
        return
    if args.only_ref_vs_ref:
        ref_vs_ref_fixture()
        return
    if args.only_suite or args.suite_shape:
        shape_suite_fixture(args.suite_shape)
        return
    if args.only_patch_knn:
        patch_knn_fixture(sphere_patches(64 + 16, 48)[64:])
        return
    if args.only_scale16:
        scale16_fixture()
        return
    if args.only_seeds:
        seed_fixture()
        return
    if args.only_fps:
        fps_fixture()
        return
    if args.only_train:
        neuron_train_fixture()
        return
    if args.only_fn_train:
        fn_train_fixture()
        fn_trainer_fixture()
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---- 0. key/shape parity of the drop-in shells with the reference modules
    fn, fd = build_models()
    my_fn = sapcu_amd.ImprovedSNNNormalEstimation(**FN_KW)
    my_fd = sapcu_amd.EnhancedSNNDistanceEstimation(**FD_KW)
    for ref_m, my_m in ((fn, my_fn), (fd, my_fd)):
        a = {k: tuple(v.shape) for k, v in ref_m.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in my_m.state_dict().items()}
        assert a == b, "state_dict layout differs from the reference"
        assert list(a) == list(b), "state_dict key ORDER differs"
    save("state_dict_layout.npz",
         fn_keys=np.array(list(fn.state_dict())), fn_shapes=np.array([str(tuple(v.shape)) for v in fn.state_dict().values()]),
         fd_keys=np.array(list(fd.state_dict())), fd_shapes=np.array([str(tuple(v.shape)) for v in fd.state_dict().values()]))

    # ---- 1. BatchNorm calibration (64 sphere patches, queries 0..63 of grid_queries(…, seed 0))
    calib = sphere_patches(64 + 16, 48)
    bn_fn = calibrate_bn(fn, calib[:64])
    bn_fd = calibrate_bn(fd, calib[:64])
    save("bn_calib_fn.npz", **bn_fn)
    save("bn_calib_fd.npz", **bn_fd)
    fn, fd = build_models(bn_fn, bn_fd)
    test = calib[64:]           # 16 patches the calibration never saw

    # ---- 2. neuron unit vectors: LIF (fn class) and EIF (fd class), self-feeding loop
    C = 8
    rng = np.random.default_rng(7)
    x = np.concatenate([np.linspace(-12, 12, 193), [-10.0, 10.0, -10.000001, 10.000001, 0.0, 1.0, 0.999999, 1.000001]])
    x = np.tile(x[:, None], (1, C)).astype(np.float32)
    x += rng.normal(0, 0.05, x.shape).astype(np.float32)
    raw = np.stack([rng.uniform(0.05, 1.05, C), rng.uniform(0.0, 0.12, C), rng.uniform(0.05, 1.0, C),
                    rng.uniform(0.6, 1.4, C), rng.uniform(0.05, 5.5, C), rng.uniform(0.05, 2.2, C)]).astype(np.float32)
    out = {"x": x, "raw_params": raw}
    for kind, cls in (("lif", ref_fn.MultiTimeConstantLIFNeuron), ("eif", ref_fd.MultiTimeConstantEIFNeuron)):
        neu = cls(C).eval()
        with torch.no_grad():
            neu.membrane_decay.copy_(torch.from_numpy(raw[0])); neu.threshold_adapt.copy_(torch.from_numpy(raw[1]))
            neu.refractory_decay.copy_(torch.from_numpy(raw[2])); neu.threshold_base.copy_(torch.from_numpy(raw[3]))
            if kind == "eif":
                neu.delta_T.copy_(torch.from_numpy(raw[4])); neu.theta_rh.copy_(torch.from_numpy(raw[5]))
            for steps in (1, 4, 7):
                v, st = torch.from_numpy(x), [None, None, None]
                for _ in range(steps):
                    v, *st = neu(v, *st)
                out["%s_T%d_spikes" % (kind, steps)] = npy(v)
                out["%s_T%d_membrane" % (kind, steps)] = npy(st[0])
                out["%s_T%d_threshold" % (kind, steps)] = npy(st[1])
                out["%s_T%d_refractory" % (kind, steps)] = npy(st[2])
    save("neuron_unit.npz", **out)

    # ---- 3. in-patch kNN: reference knn() on xyz patches and on soft-spike-like features
    patch_knn_fixture(test)

    # ---- 4. fn stage taps (b = 4)
    xin = test[:4]
    names = ["encoder.snn_init", "encoder.trans1", "encoder.trans2", "encoder.trans3", "encoder", "decoder.fc_out"]
    store, hooks = hook_outputs(fn, names)
    with torch.no_grad():
        for b in (fn.encoder.trans1, fn.encoder.trans2, fn.encoder.trans3):
            b.knn_cache.cache.clear()
        normals = fn(xin)
    for h in hooks:
        h.remove()
    knn_tabs = [list(b.knn_cache.cache.values())[0] for b in (fn.encoder.trans1, fn.encoder.trans2, fn.encoder.trans3)]
    save("fn_taps.npz", patch=npy(xin),
         stem=npy(store["encoder.snn_init"][-1]).transpose(0, 2, 1),      # last of the T self-loop calls -> [b,M,64]
         block1=npy(store["encoder.trans1"][0]), block2=npy(store["encoder.trans2"][0]), block3=npy(store["encoder.trans3"][0]),
         enc=npy(store["encoder"][0]), logits=npy(store["decoder.fc_out"][0]), normals=npy(normals),
         knn0=npy(knn_tabs[0]).astype(np.int8), knn1=npy(knn_tabs[1]).astype(np.int8), knn2=npy(knn_tabs[2]).astype(np.int8))

    # ---- 5. the stale-cache pair: second call of the same shape reuses the first call's neighbours
    with torch.no_grad():
        for b in (fn.encoder.trans1, fn.encoder.trans2, fn.encoder.trans3):
            b.knn_cache.cache.clear()
        nA = fn(test[4:8]); fn.reset_states()
        nB = fn(test[8:12])                      # stale: uses A's tables
        for b in (fn.encoder.trans1, fn.encoder.trans2, fn.encoder.trans3):
            b.knn_cache.cache.clear()
        nB_fresh = fn(test[8:12])
    save("fn_cache_pair.npz", patch_a=npy(test[4:8]), patch_b=npy(test[8:12]), normals_a=npy(nA), normals_b_stale=npy(nB),
         normals_b_fresh=npy(nB_fresh))

    # ---- 6. fd stage taps (b = 4): monkey-free — hooks on modules + a recording get_graph_feature
    xin = test[:4]
    rec = {"knn": []}
    orig_knn = ref_fd.knn

    def rec_knn(x, k):
        idx = orig_knn(x, k)
        rec["knn"].append((x.shape[1], k, idx))
        return idx

    ref_fd.knn = rec_knn
    names = ["encoder.scale_fusion", "encoder.snn_blocks.0", "encoder.snn_blocks.1", "encoder.snn_blocks.2",
             "encoder.snn_blocks.3", "encoder.multi_scale_conv", "encoder"]
    store, hooks = hook_outputs(fd, names)
    with torch.no_grad():
        fd.reset_states()
        dist = fd(xin)
    for h in hooks:
        h.remove()
    ref_fd.knn = orig_knn
    Tn = FD_KW["time_steps_enc"]
    spikes = np.stack([np.concatenate([npy(store["encoder.snn_blocks.%d" % i][t]) for i in range(4)], axis=1).transpose(0, 2, 1)
                       for t in range(Tn)], 0)                                    # [T,b,M,960]
    pooled = np.stack([npy(store["encoder.multi_scale_conv"][t].max(dim=2)[0]) for t in range(Tn)], 0)
    per_t = len(rec["knn"]) // Tn
    knn_feat = [npy(rec["knn"][i][2]).astype(np.int8) for i in range(per_t) if rec["knn"][i][0] != 3]   # t = 0, blocks 1..3
    save("fd_taps.npz", patch=npy(xin), fused0=npy(store["encoder.scale_fusion"][0]).transpose(0, 2, 1),
         spikes_t0=spikes[0], spikes_tlast=spikes[-1], pooled=pooled, enc=npy(store["encoder"][0]), dist=npy(dist),
         knn1=knn_feat[0], knn2=knn_feat[1], knn3=knn_feat[2])

    # ---- 7. shape/T variants (final outputs only)
    out = {}
    for M in (12, 100):
        p = sphere_patches(3, M, qseed=1)
        with torch.no_grad():
            for b in (fn.encoder.trans1, fn.encoder.trans2, fn.encoder.trans3):
                b.knn_cache.cache.clear()
            out["patch_M%d" % M] = npy(p)
            out["normals_M%d" % M] = npy(fn(p))
            fd.reset_states()
            out["dist_M%d" % M] = npy(fd(p))
    p = sphere_patches(3, 48, qseed=2)
    out["patch_T"] = npy(p)
    for Tv in (6, 7):
        fkw = dict(FN_KW, time_steps_enc=Tv)
        dkw = dict(FD_KW, time_steps_enc=Tv)
        fnv, fdv = build_models(bn_fn, bn_fd, fkw, dkw)
        with torch.no_grad():
            out["normals_T%d" % Tv] = npy(fnv(p))
            out["dist_T%d" % Tv] = npy(fdv(p))
    save("variants.npz", **out)

    # ---- 8. outer kNN through sklearn's KDTree exactly as generation.py:110,127 calls it
    from sklearn.neighbors import KDTree
    cloud = T.sphere_cloud(5000, 0)
    q = T.grid_queries(4096, 0)
    d, idx = KDTree(cloud).query(q, 48)
    cloud2 = T.sphere_cloud(2048, 0)
    q2 = T.grid_queries(256, 3)
    d2, idx2 = KDTree(cloud2).query(q2, 100)
    save("outer_knn.npz", idx_n5000_k48=idx.astype(np.int16), dist_head=d[:64], idx_n2048_k100=idx2.astype(np.int16),
         dist2_head=d2[:16])

    # ---- 9. rotation matrices from the reference function
    rng = np.random.default_rng(11)
    nr = rng.normal(size=(58, 3)).astype(np.float32)
    nr /= np.linalg.norm(nr, axis=1, keepdims=True)
    special = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, 0, -1], [0.99999994, 1e-4, 0], [-0.99999994, 0, 1e-4]], np.float32)
    nr = np.concatenate([special, nr], 0)
    mats = np.stack([ref_gen.rotation_matrix_from_vectors(n, [1, 0, 0]) for n in nr], 0)
    pat = G.gather_centre(cloud, q[:64], idx[:64])
    rot = np.stack([np.matmul(mats[j], pat[j].T).T for j in range(64)], 0)
    save("rotation.npz", normals=nr, matrices=mats, rotated_f32=rot.astype(np.float32))

    # ---- 9b. seed generator outputs, FPS / normalisation (generate.py), training-mode neuron (row f-4)
    seed_fixture()
    fps_fixture()
    neuron_train_fixture()
    fn_train_fixture()
    fn_trainer_fixture()
    if not args.skip_e2e:
        shape_suite_fixture()
        scale16_fixture()

    # ---- 10. end-to-end Generator3D6.upsample on sphere N=2048, dense_spacing 0.03 (~900 seeds)
    if not args.skip_e2e:
        dense = os.path.join(ROOT, "oracle", "_ref", "dense")
        if not os.path.exists(dense):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        work = tempfile.mkdtemp(prefix="sapcu_e2e_")
        shutil.copy(dense, os.path.join(work, "dense"))
        cwd = os.getcwd()
        os.chdir(work)
        try:
            cloud = T.sphere_cloud(2048, 0)
            np.savetxt("test.xyz", cloud, fmt="%.6f")        # dense reads test.xyz (generate.py never writes it, §8f-1)
            captured = []
            RealTree = ref_gen.KDTree

            class RecTree(RealTree):                          # records the array each KDTree is built on
                def __init__(self, data, *a, **k):
                    captured.append(np.array(data, copy=True))
                    super().__init__(data, *a, **k)

            ref_gen.KDTree = RecTree
            for b in (fn.encoder.trans1, fn.encoder.trans2, fn.encoder.trans3):
                b.knn_cache.cache.clear()
            gen = ref_gen.Generator3D6(fn, fd, torch.device("cpu"), k_neighbors=48, dense_spacing=0.03, batch_size=64)
            result = gen.upsample(cloud[None])
            ref_gen.KDTree = RealTree
            seeds = np.loadtxt("target.xyz")[:, 0:3]
            save("e2e_upsample.npz", seeds=seeds, unfiltered=captured[1], filtered=result)
        finally:
            os.chdir(cwd)
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
