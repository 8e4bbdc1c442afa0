"""CPU ORACLE (test infrastructure, NOT product code) — training-mode neuron loop.

torch restatement of /root/reference/fn/snn_coder.py:87-151 with ``self.training`` (hard spikes forward, soft-surrogate
derivative backward through autograd's straight-through construction, constant gate mask), driven as at :318-320.
Gradients come from torch autograd on this restatement.  Only ``tests/`` may import it.  Pinned by
tests/golden/neuron_train.npz (the reference's own module in train mode, forward and backward).
"""
import math

import torch


def _expand(p, x):
    return p.view([1, -1] + [1] * (x.dim() - 2)).expand_as(x)


def spike_train(u, grad_width=10.0):
    uc = torch.clamp(u, -10.0, 10.0)
    soft = 0.5 * torch.exp(-(uc ** 2) / 2) / math.sqrt(2 * math.pi) + 0.5 * torch.sigmoid(grad_width * uc)
    hard = (u > 0).float()
    return soft + (hard - soft).detach()


def lif_selfloop_train(x, membrane_decay, threshold_adapt, refractory_decay, threshold_base, steps=4):
    decay = _expand(torch.clamp(membrane_decay, 0.1, 0.99), x)
    adapt = _expand(torch.clamp(threshold_adapt, 0.001, 0.1), x)
    rdecay = _expand(torch.clamp(refractory_decay, 0.1, 0.95), x)
    theta0 = _expand(threshold_base, x)
    m, th, r = torch.zeros_like(x), theta0, torch.zeros_like(x)
    for _ in range(steps):
        x = x * (r <= 0).float()
        m = m * decay * (1 - r) + x
        sp = spike_train(m - th)
        m = m * (1 - sp)
        r = r * rdecay + sp
        th = th + adapt * sp
        th = theta0 + (th - theta0) * 0.95
        x = sp
    return x


def conv_bn_lif_train(x, weight, bias, gamma, beta, membrane_decay, threshold_adapt, refractory_decay, threshold_base,
                      steps=4, eps=1e-5):
    """fn/snn_coder.py:225-229 + 317-320 in training mode on channels-last rows: x [rows, c_in] -> spikes [rows, c_out]."""
    y = x @ weight.reshape(weight.shape[0], -1).t() + bias
    mean = y.mean(0)
    var = y.var(0, unbiased=False)
    z = (y - mean) / torch.sqrt(var + eps) * gamma + beta
    return lif_selfloop_train(z, membrane_decay, threshold_adapt, refractory_decay, threshold_base, steps)


def softmax_agg(a, pe, v, idx, m, sqrt_hd):
    """fn/snn_coder.py:379-389 on edge rows (the stage oracle/snn_path.py:158-159 evaluates channels-first): a, pe
    [P*k, d], v [P, d], idx [P*k] in-patch neighbour indices, m points per patch -> [P, d]."""
    pts, d = v.shape
    kk = a.shape[0] // pts
    nbr = (torch.arange(pts).div(m, rounding_mode="floor") * m).repeat_interleave(kk) + idx.long()
    w = torch.softmax(a.view(pts, kk, d) / sqrt_hd, dim=1)
    u = v[nbr].view(pts, kk, d) + pe.view(pts, kk, d)
    return (w * u).sum(1)


def conv_bn_train(x, weight, bias, gamma, beta, eps=1e-5):
    y = x @ weight.reshape(weight.shape[0], -1).t() + bias
    return (y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + eps) * gamma + beta


def transformer_block_train(p, xyz, features, knn_idx, time_steps=4, num_heads=8, eps=1e-5):
    """MultiHeadSNNTransformerBlock.forward in training mode (fn/snn_coder.py:294-396, dropout 0), restated on
    channels-last rows with plain torch ops; p = the block's parameters under the reference's names."""
    B, N, _ = xyz.shape
    k = knn_idx.shape[-1]
    P = B * N
    nbr = (knn_idx.long() + (torch.arange(B) * N).view(B, 1, 1)).reshape(P * k)
    ptr = torch.arange(P).repeat_interleave(k)
    feat, xyzr = features.reshape(P, -1), xyz.reshape(P, 3)

    def layer(x, conv, bn, snn=None):
        z = conv_bn_train(x, p[conv + ".weight"], p[conv + ".bias"], p[bn + ".weight"], p[bn + ".bias"], eps)
        if snn is None:
            return z
        return lif_selfloop_train(z, p[snn + ".membrane_decay"], p[snn + ".threshold_adapt"], p[snn + ".refractory_decay"],
                                  p[snn + ".threshold_base"], time_steps)

    x = layer(feat, "fc1.0", "fc1.1", "snn1")
    q, kf, v = (layer(x, "w_%ss.0" % c, "w_%ss.1" % c, "snn_" + c) for c in "qkv")
    pe = layer(xyzr[ptr] - xyzr[nbr], "fc_delta.0", "fc_delta.1", "snn_delta")
    pe = layer(pe, "fc_delta2.0", "fc_delta2.1", "snn_delta2")
    a = layer(q[ptr] - kf[nbr] + pe, "fc_gamma.0", "fc_gamma.1", "snn_gamma")
    a = layer(a, "fc_gamma2.0", "fc_gamma2.1")
    d_model = a.shape[1]
    res = softmax_agg(a, pe, v, knn_idx.reshape(P * k), N, float((d_model // num_heads) ** 0.5))
    res = layer(res, "out_proj.0", "out_proj.1")
    return (layer(res, "fc2.0", "fc2.1") + feat).view(B, N, -1)


def fn_train_forward(p, points, knn, time_steps_enc=4, num_heads=8, eps=1e-5):
    """ImprovedSNNNormalEstimation.forward in training mode, dropout off (fn/snn_coder.py:430-476, 542-549), restated with
    plain torch ops on channels-last rows; knn = the three blocks' in-patch neighbour tables [B, N, k]."""
    import torch.nn.functional as F
    B, N, _ = points.shape
    P = B * N

    def sub(prefix):
        return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}

    def lif(z, e, name):
        return lif_selfloop_train(z, e[name + ".membrane_decay"], e[name + ".threshold_adapt"], e[name + ".refractory_decay"],
                                  e[name + ".threshold_base"], time_steps_enc)

    enc = sub("encoder.")
    cur = lif(conv_bn_train(points.reshape(P, 3), enc["conv1.0.weight"], enc["conv1.0.bias"], enc["conv1.1.weight"], enc["conv1.1.bias"], eps),
              enc, "snn_init").view(B, N, 64)
    feats = []
    for i in range(3):
        cur = transformer_block_train(sub("encoder.trans%d." % (i + 1)), points, cur, knn[i], 4, num_heads, eps)
        feats.append(cur)
    g = lif(conv_bn_train(torch.cat(feats, 2).reshape(P, 192), enc["conv_final.0.weight"], enc["conv_final.0.bias"],
                          enc["conv_final.1.weight"], enc["conv_final.1.bias"], eps), enc, "snn_final")
    x = g.view(B, N, -1).max(dim=1)[0] @ enc["fc_out.weight"].t() + enc["fc_out.bias"]
    dec = sub("decoder.")
    for li in sorted({int(k.split(".")[1]) for k in dec if k.startswith("mlp.") and k.endswith(".weight") and dec[k].dim() == 2}):
        x = F.gelu(conv_bn_train(x, dec["mlp.%d.weight" % li], dec["mlp.%d.bias" % li], dec["mlp.%d.weight" % (li + 1)],
                                 dec["mlp.%d.bias" % (li + 1)], eps))
    x = x @ dec["fc_out.weight"].t() + dec["fc_out.bias"]
    return F.normalize(F.layer_norm(x, (3,), dec["norm_out.weight"], dec["norm_out.bias"], 1e-5), dim=1)


def angular_loss(pred, gt, temperature=0.1, alpha=0.1):
    """enhanced_angular_loss_with_consistency without the consistency term (fn/snn_coder.py:603-612): for [B, 3] predictions
    that term compares each normal with copies of itself — its value is 0.15 * mean(1 - 1) and its gradient is zero."""
    cos = torch.nn.functional.cosine_similarity(pred, gt, dim=1)
    err = torch.acos(torch.clamp(cos, -1 + 1e-6, 1 - 1e-6))
    conf = torch.sigmoid(err.detach() / temperature)
    return (err * conf + alpha * (conf - 0.5) ** 2).mean()


def angular_loss_with_consistency(pred, gt, xyz, temperature=0.1, alpha=0.1, consistency_weight=0.15, k_neighbors=8):
    """enhanced_angular_loss_with_consistency with its consistency term (fn/snn_coder.py:557-625): pred/gt [B, N, 3] or [B, 3],
    xyz [B, N, 3] -> (loss, mean confidence)."""
    F = torch.nn.functional
    base = angular_loss(pred.reshape(-1, 3), gt.reshape(-1, 3), temperature, alpha)
    cos = F.cosine_similarity(pred.reshape(-1, 3), gt.reshape(-1, 3), dim=1)
    conf = torch.sigmoid(torch.acos(torch.clamp(cos, -1 + 1e-6, 1 - 1e-6)).detach() / temperature).mean()
    B, N, _ = xyz.shape
    d = ((xyz[:, :, None, :] - xyz[:, None, :, :]) ** 2).sum(-1)
    nbr = d.argsort()[:, :, 1:k_neighbors + 1]
    full = pred.unsqueeze(1).expand(B, N, 3) if pred.dim() == 2 else pred.view(B, N, 3)
    nb = torch.gather(full.unsqueeze(1).expand(B, N, N, 3), 2, nbr.unsqueeze(-1).expand(B, N, nbr.shape[2], 3))
    return base + consistency_weight * (1 - F.cosine_similarity(full.unsqueeze(2), nb, dim=-1)).mean(), conf
