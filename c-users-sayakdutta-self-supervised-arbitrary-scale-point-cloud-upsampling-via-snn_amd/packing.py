"""Pack a reference-layout ``state_dict`` into the flat f32 blob + slot directory the C ABI takes.

* eval-mode BatchNorm is folded into the preceding 1x1 conv / Linear in float64
  (``W' = s W``, ``b' = s (b - mean) + beta`` with ``s = gamma / sqrt(var + 1e-5)``);
* q/k/v projections (and their neurons) are concatenated so one GEMM produces all three;
* fd EdgeConv weights are stored in factored form: ``W.cat(xj - xi, xj) = (W1 + W2) xj - W1 xi``
  (fd/snn_coder.py:67), rows ``[s (W1 + W2) ; s W1]``, BN shift kept separately;
* neuron parameters are stored RAW as ``[n_params][channels]`` (the kernels clamp, as the
  reference does at every call — fn/snn_coder.py:116-118);
* slot order == the enums in ``csrc/model.hip``.
"""
import numpy as np

ALIGN = 64  # floats (256 B)
BN_EPS = 1e-5


def _np(t):
    return t.detach().cpu().double().numpy() if hasattr(t, "detach") else np.asarray(t, dtype=np.float64)


def fold_bn(sd, lin, bn):
    """(W2d, b) of ``bn(lin(x))`` in float64.  lin bias may be absent."""
    w = _np(sd[lin + ".weight"])
    w = w.reshape(w.shape[0], -1)
    b = _np(sd[lin + ".bias"]) if (lin + ".bias") in sd else np.zeros(w.shape[0])
    s = _np(sd[bn + ".weight"]) / np.sqrt(_np(sd[bn + ".running_var"]) + BN_EPS)
    return w * s[:, None], s * (b - _np(sd[bn + ".running_mean"])) + _np(sd[bn + ".bias"])


def plain(sd, lin):
    w = _np(sd[lin + ".weight"])
    return w.reshape(w.shape[0], -1), _np(sd[lin + ".bias"])


def neuron(sd, pfx, eif=False):
    names = ["membrane_decay", "threshold_adapt", "refractory_decay", "threshold_base"]
    if eif:
        names += ["delta_T", "theta_rh"]
    return np.stack([_np(sd[pfx + "." + n]) for n in names], 0)


class _Blob:
    def __init__(self):
        self.parts, self.dir, self.n = [], [], 0

    def add(self, arr):
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float64).astype(np.float32)).ravel()
        self.dir.append(self.n)
        pad = (-a.size) % ALIGN
        self.parts.append(a)
        if pad:
            self.parts.append(np.zeros(pad, np.float32))
        self.n += a.size + pad

    def finish(self):
        return np.concatenate(self.parts), np.asarray(self.dir, dtype=np.int64)


FN_SLOTS = 81
FD_SLOTS = 42


def pack_fn(sd, decoder_linear_idx=(0, 4, 8)):
    """state_dict of ImprovedSNNNormalEstimation -> (blob f32, dir int64[81])."""
    B = _Blob()
    w, b = fold_bn(sd, "encoder.conv1.0", "encoder.conv1.1")
    B.add(w); B.add(b); B.add(neuron(sd, "encoder.snn_init"))
    for name in ("trans1", "trans2", "trans3"):
        p = "encoder." + name
        w, b = fold_bn(sd, p + ".fc1.0", p + ".fc1.1")
        B.add(w); B.add(b); B.add(neuron(sd, p + ".snn1"))
        ws, bs, ns = [], [], []
        for proj, snn in (("w_qs", "snn_q"), ("w_ks", "snn_k"), ("w_vs", "snn_v")):
            w, b = fold_bn(sd, p + "." + proj + ".0", p + "." + proj + ".1")
            ws.append(w); bs.append(b); ns.append(neuron(sd, p + "." + snn))
        B.add(np.concatenate(ws, 0)); B.add(np.concatenate(bs, 0)); B.add(np.concatenate(ns, 1))
        for conv, snn in (("fc_delta", "snn_delta"), ("fc_delta2", "snn_delta2"), ("fc_gamma", "snn_gamma")):
            w, b = fold_bn(sd, p + "." + conv + ".0", p + "." + conv + ".1")
            B.add(w); B.add(b); B.add(neuron(sd, p + "." + snn))
        for conv in ("fc_gamma2", "out_proj", "fc2"):
            w, b = fold_bn(sd, p + "." + conv + ".0", p + "." + conv + ".1")
            B.add(w); B.add(b)
    w, b = fold_bn(sd, "encoder.conv_final.0", "encoder.conv_final.1")
    B.add(w); B.add(b); B.add(neuron(sd, "encoder.snn_final"))
    w, b = plain(sd, "encoder.fc_out")
    B.add(w); B.add(b)
    for li in decoder_linear_idx:
        w, b = fold_bn(sd, "decoder.mlp.%d" % li, "decoder.mlp.%d" % (li + 1))
        B.add(w); B.add(b)
    w, b = plain(sd, "decoder.fc_out")
    B.add(w); B.add(b)
    B.add(_np(sd["decoder.norm_out.weight"])); B.add(_np(sd["decoder.norm_out.bias"]))
    blob, d = B.finish()
    assert d.size == FN_SLOTS, d.size
    return blob, d


def pack_fd(sd, n_scales):
    """state_dict of EnhancedSNNDistanceEstimation -> (blob f32, dir int64[42])."""
    B = _Blob()
    e = "encoder."
    ws, bs = [], []
    for s in range(n_scales):
        w, b = fold_bn(sd, e + "multi_scale_first_conv.%d.0" % s, e + "multi_scale_first_conv.%d.1" % s)
        ws.append(w); bs.append(b)
    B.add(np.stack(ws, 0)); B.add(np.stack(bs, 0))
    w, b = fold_bn(sd, e + "scale_fusion.0", e + "scale_fusion.1")
    B.add(w); B.add(b); B.add(neuron(sd, e + "snn_blocks.0", eif=True))
    for l in (1, 2, 3):
        w, shift = fold_bn(sd, e + "conv_blocks.%d.0" % (l - 1), e + "conv_blocks.%d.1" % (l - 1))
        cin = w.shape[1] // 2
        w1, w2 = w[:, :cin], w[:, cin:]
        B.add(np.concatenate([w1 + w2, w1], 0)); B.add(shift)
        B.add(neuron(sd, e + "snn_blocks.%d" % l, eif=(l == 1)))
    w, b = fold_bn(sd, e + "multi_scale_conv.0", e + "multi_scale_conv.1")
    B.add(w); B.add(b)
    B.add(_np(sd[e + "temporal_integration.weights"])); B.add(neuron(sd, e + "snn_fc"))
    d = "distance_decoder."
    w, b = fold_bn(sd, d + "fc_in.0", d + "fc_in.1")
    B.add(w); B.add(b)
    for r in (0, 1):
        p = d + "residual_blocks.%d." % r
        w, b = fold_bn(sd, p + "fc.0", p + "fc.1"); B.add(w); B.add(b)
        w, b = fold_bn(sd, p + "fc.4", p + "fc.5"); B.add(w); B.add(b)
        w, b = plain(sd, p + "res_proj"); B.add(w); B.add(b)
    w, b = plain(sd, d + "attention.to_qkv"); B.add(w); B.add(b)
    w, b = plain(sd, d + "attention.to_out.0"); B.add(w.T); B.add(b)
    B.add(_np(sd[d + "attention.norm.weight"])); B.add(_np(sd[d + "attention.norm.bias"]))
    w, b = fold_bn(sd, d + "fc_hidden.0", d + "fc_hidden.1"); B.add(w.T); B.add(b)
    w, b = plain(sd, d + "fc_distance"); B.add(w[0]); B.add(b)
    blob, dr = B.finish()
    assert dr.size == FD_SLOTS, dr.size
    return blob, dr
