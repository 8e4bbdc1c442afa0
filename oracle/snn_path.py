"""CPU ORACLE (test infrastructure, NOT product code) — restatement of the SNN forwards.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``sapcu_amd``) never does and fails loudly when
its HIP library is missing.

This is an independent, functional (state_dict-in, tensor-out) restatement of the
reference's per-query-point networks *as the reference executes them* (every time
step, every gate, every dead stage is evaluated — no algebraic shortcut), written
from the maths in SURVEY.md §8a and pinned against golden vectors generated from the
real reference (``tests/golden/make_fixtures.py``).  It runs on torch-CPU fp32 ops so
that its rounding behaviour (oneDNN conv, ATen elementwise) is as close to the
reference's CPU path as a restatement can be.

Reference lines followed (paths relative to /root/reference):
  neuron step ............ fn/snn_coder.py:87-153, fd/snn_coder.py:94-155, 198-275
  in-patch kNN ........... fn/snn_coder.py:31-39 (cache semantics :47-59), fd/snn_coder.py:25-32
  fn transformer block ... fn/snn_coder.py:294-396
  fn encoder / decoder ... fn/snn_coder.py:430-476, 542-549
  fd graph feature ....... fd/snn_coder.py:52-68
  fd encoder ............. fd/snn_coder.py:392-492
  fd decoder ............. fd/snn_coder.py:711-725, 751-758, 777-798
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

_INV_SQRT_2PI_DIV = float(np.sqrt(2 * np.pi))  # fn/snn_coder.py:140 divides by this


# --------------------------------------------------------------------------- neurons
def soft_spike(x):
    """Eval-mode spike surrogate 0.5*N(x)+0.5*sigmoid(10x) on clamp(x,+-10) (fn:135-146)."""
    xc = torch.clamp(x, -10.0, 10.0)
    gauss = torch.exp(-(xc ** 2) / 2) / _INV_SQRT_2PI_DIV
    sig = torch.sigmoid(10.0 * xc)
    return 0.5 * gauss + 0.5 * sig


def _bcast(p, x):
    shape = [1, -1] + [1] * (x.dim() - 2)
    return p.view(shape)


def neuron_params(sd, pfx):
    """Fetch (and clamp, as the reference does at every call) one neuron's parameters."""
    p = {
        "decay": torch.clamp(sd[pfx + ".membrane_decay"], 0.1, 0.99),
        "adapt": torch.clamp(sd[pfx + ".threshold_adapt"], 0.001, 0.1),
        "rdecay": torch.clamp(sd[pfx + ".refractory_decay"], 0.1, 0.95),
        "theta0": sd[pfx + ".threshold_base"],
    }
    if pfx + ".delta_T" in sd:
        p["delta_T"] = torch.clamp(sd[pfx + ".delta_T"], 0.1, 5.0)
        p["theta_rh"] = torch.clamp(sd[pfx + ".theta_rh"], 0.1, 2.0)
    return p


def neuron_step(x, state, p):
    """One LIF (fn:125-131) or EIF (fd:245-259) step. state = (m, theta, r) or None."""
    decay, adapt, rdecay, theta0 = (_bcast(p[k], x) for k in ("decay", "adapt", "rdecay", "theta0"))
    if state is None:
        m = torch.zeros_like(x)
        theta = theta0.expand_as(x)
        r = torch.zeros_like(x)
    else:
        m, theta, r = state
    extra = None
    if "delta_T" in p:  # EIF: exponential term from the PRE-update membrane, not gated
        dT = _bcast(p["delta_T"], x)
        rh = _bcast(p["theta_rh"], x)
        extra = dT * torch.exp(torch.clamp((m - rh) / (dT + 1e-6), -5.0, 5.0))
    x = x * (r <= 0).float()
    m = m * decay * (1 - r) + x
    if extra is not None:
        m = m + extra
    s = soft_spike(m - theta)
    m = m * (1 - s)
    r = r * rdecay + s
    theta = theta + adapt * s
    theta = theta0 + (theta - theta0) * 0.95
    return s, (m, theta, r)


def neuron_selfloop(x, p, T):
    """``for t: x, *st = snn(x, *st)`` — the spikes are fed back as the next input (fn:319-320)."""
    st = None
    for _ in range(T):
        x, st = neuron_step(x, st, p)
    return x


# --------------------------------------------------------------------------- layers
def _bn_eval(x, sd, pfx):
    return F.batch_norm(x, sd[pfx + ".running_mean"], sd[pfx + ".running_var"],
                        sd[pfx + ".weight"], sd[pfx + ".bias"], False, 0.0, 1e-5)


def conv_bn(x, sd, pfx):
    """1x1 Conv1d/Conv2d (pfx.0) followed by eval-mode BatchNorm (pfx.1)."""
    w = sd[pfx + ".0.weight"]
    b = sd.get(pfx + ".0.bias")
    y = F.conv2d(x, w, b) if w.dim() == 4 else F.conv1d(x, w, b)
    return _bn_eval(y, sd, pfx + ".1")


def linear(x, sd, pfx):
    return F.linear(x, sd[pfx + ".weight"], sd.get(pfx + ".bias"))


def inpatch_knn_scores(x):
    """-|xi|^2 + 2 xi.xj - |xj|^2 for x [b,C,N], in the reference's op order (fn:35-37)."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    return -xx - inner - xx.transpose(2, 1)


def inpatch_knn(x, k):
    """Indices [b,N,k] of the k largest scores per row (self included), descending (fn:38)."""
    k = min(k, x.shape[2])
    return inpatch_knn_scores(x).topk(k=k, dim=-1)[1]


def gather_cols(x, idx):
    """x [b,C,N], idx [b,N,k] -> [b,C,N,k] with out[b,c,i,j] = x[b,c,idx[b,i,j]]."""
    b, C, N = x.shape
    k = idx.shape[2]
    flat = idx.reshape(b, 1, N * k).expand(b, C, N * k)
    return torch.gather(x, 2, flat).view(b, C, N, k)


# --------------------------------------------------------------------------- fn
FN_BLOCKS = (("trans1", 128), ("trans2", 256), ("trans3", 512))
FN_BLOCK_T = 4  # hard-coded in the reference (fn/snn_coder.py:417-419)


def fn_block(sd, pfx, xyz, feat, knn_idx, heads, taps=None):
    """One SNN point-transformer block. xyz [b,N,3], feat [b,N,64] -> [b,N,64]."""
    T = FN_BLOCK_T
    d = sd[pfx + ".fc1.0.weight"].shape[0]
    pos = xyz.permute(0, 2, 1)                                  # [b,3,N]
    pos_diff = pos.unsqueeze(-1) - gather_cols(pos, knn_idx)    # xi - xj  [b,3,N,k]
    pre = feat.permute(0, 2, 1).contiguous()                    # [b,64,N]
    x = neuron_selfloop(conv_bn(pre, sd, pfx + ".fc1"), neuron_params(sd, pfx + ".snn1"), T)
    q = neuron_selfloop(conv_bn(x, sd, pfx + ".w_qs"), neuron_params(sd, pfx + ".snn_q"), T)
    kf = neuron_selfloop(conv_bn(x, sd, pfx + ".w_ks"), neuron_params(sd, pfx + ".snn_k"), T)
    v = neuron_selfloop(conv_bn(x, sd, pfx + ".w_vs"), neuron_params(sd, pfx + ".snn_v"), T)
    kg = gather_cols(kf, knn_idx)
    vg = gather_cols(v, knn_idx)
    pe = neuron_selfloop(conv_bn(pos_diff.contiguous(), sd, pfx + ".fc_delta"),
                         neuron_params(sd, pfx + ".snn_delta"), T)
    pe = neuron_selfloop(conv_bn(pe, sd, pfx + ".fc_delta2"), neuron_params(sd, pfx + ".snn_delta2"), T)
    a = q.unsqueeze(-1) - kg + pe
    a = neuron_selfloop(conv_bn(a, sd, pfx + ".fc_gamma"), neuron_params(sd, pfx + ".snn_gamma"), T)
    a = conv_bn(a, sd, pfx + ".fc_gamma2")
    a = F.softmax(a / np.sqrt(d // heads), dim=-1)
    res = torch.einsum("bcnk,bcnk->bcn", a, vg + pe)          # [b,d,N]
    res = conv_bn(res, sd, pfx + ".out_proj")
    res = conv_bn(res, sd, pfx + ".fc2") + pre
    if taps is not None:
        taps[pfx + ".x"] = x
        taps[pfx + ".q"] = q
        taps[pfx + ".pe"] = pe
        taps[pfx + ".attn"] = a
    return res.permute(0, 2, 1).contiguous()


def fn_encoder(sd, patch, hp, knn_idx=None, taps=None):
    """patch [b,M,3] (or [b,3,M], fn:441) -> [b,2048]. ``knn_idx``: optional list of three
    [b,M,k] index tensors (stale-cache emulation, fn:47-59); computed fresh when None."""
    x = patch if patch.shape[1] == 3 else patch.permute(0, 2, 1).contiguous()   # [b,3,N]
    xyz = x.permute(0, 2, 1).contiguous()
    T = hp["time_steps_enc"]
    feat = neuron_selfloop(conv_bn(x, sd, "encoder.conv1"), neuron_params(sd, "encoder.snn_init"), T)
    feat = feat.permute(0, 2, 1).contiguous()
    if taps is not None:
        taps["encoder.snn_init"] = feat
    outs = []
    used_idx = []
    for bi, (name, _d) in enumerate(FN_BLOCKS):
        kk = min(hp["k_values"][bi], xyz.shape[1])
        idx = knn_idx[bi] if knn_idx is not None else inpatch_knn(xyz.permute(0, 2, 1).contiguous(), kk)
        used_idx.append(idx)
        feat = fn_block(sd, "encoder." + name, xyz, feat, idx, hp["num_heads"], taps)
        outs.append(feat)
        if taps is not None:
            taps["encoder." + name] = feat
    ms = torch.cat(outs, dim=2).permute(0, 2, 1)
    g = neuron_selfloop(conv_bn(ms, sd, "encoder.conv_final"), neuron_params(sd, "encoder.snn_final"), T)
    g = g.max(dim=2)[0]
    if taps is not None:
        taps["encoder.pooled"] = g
        taps["knn_idx"] = used_idx
    return linear(g, sd, "encoder.fc_out")


def fn_decoder(sd, f, taps=None):
    for li, bi in ((0, 1), (4, 5), (8, 9)):
        f = linear(f, sd, "decoder.mlp.%d" % li)
        f = F.gelu(_bn_eval(f, sd, "decoder.mlp.%d" % bi))
    f = linear(f, sd, "decoder.fc_out")
    if taps is not None:
        taps["decoder.logits"] = f
    f = F.layer_norm(f, (3,), sd["decoder.norm_out.weight"], sd["decoder.norm_out.bias"], 1e-5)
    return F.normalize(f, dim=1)


def fn_forward(sd, patch, hp, knn_idx=None, taps=None):
    """ImprovedSNNNormalEstimation.forward for 3-D input (fn:670-699) -> unit normals [b,3]."""
    feats = fn_encoder(sd, patch, hp, knn_idx, taps)
    if taps is not None:
        taps["encoder.out"] = feats
    return fn_decoder(sd, feats, taps)


# --------------------------------------------------------------------------- fd
def graph_feature(x, k, idx=None):
    """cat(x_j - x_i, x_j) over the k in-patch neighbours j of i (fd:52-68) -> [b,2C,N,k]."""
    if idx is None:
        idx = inpatch_knn(x, k)
    nb = gather_cols(x, idx)
    return torch.cat((nb - x.unsqueeze(-1), nb), dim=1), idx


def _conv_bn_lrelu(x, sd, pfx):
    return F.leaky_relu(conv_bn(x, sd, pfx), 0.2)


def fd_encoder(sd, patch, hp, taps=None, force_idx=None):
    """patch [b,M,3] -> [b,emb]. Every stage is evaluated at every t (fd:408-480).

    ``force_idx``: optional list of three [b,M,k] tensors replacing the feature-space neighbours of
    blocks 1..3 at t = 0 (checker protocol for near-tie flips: run the device path, read back the
    neighbours it chose, and evaluate the reference arithmetic on exactly those)."""
    x = patch if patch.shape[1] == 3 else patch.transpose(1, 2).contiguous()    # [b,3,M]
    M = x.shape[2]
    T = hp["time_steps_enc"]
    prm = [neuron_params(sd, "encoder.snn_blocks.%d" % i) for i in range(4)]
    states = [None] * 4
    pooled = []
    for t in range(T):
        feats = []
        sc = []
        for si, ks in enumerate(hp["k_scales"]):
            g, _ = graph_feature(x, min(ks, M))
            sc.append(_conv_bn_lrelu(g, sd, "encoder.multi_scale_first_conv.%d" % si).max(dim=-1)[0])
        cur = _conv_bn_lrelu(torch.cat(sc, dim=1), sd, "encoder.scale_fusion")
        if taps is not None and t == 0:
            taps["encoder.fused0"] = cur
        cur, states[0] = neuron_step(cur, states[0] if t > 0 else None, prm[0])
        feats.append(cur)
        for bi in range(1, 4):
            g, idx = graph_feature(cur, min(hp["k"], M),
                                   force_idx[bi - 1] if (force_idx is not None and t == 0) else None)
            if taps is not None and t == 0:
                taps["encoder.knn%d" % bi] = idx
            cur = _conv_bn_lrelu(g, sd, "encoder.conv_blocks.%d" % (bi - 1)).max(dim=-1)[0]
            cur, states[bi] = neuron_step(cur, states[bi] if t > 0 else None, prm[bi])
            feats.append(cur)
        if taps is not None and t in (0, T - 1):
            for bi in range(4):
                taps["encoder.spk%d.t%d" % (bi, t)] = feats[bi]
        agg = _conv_bn_lrelu(torch.cat(feats, dim=1), sd, "encoder.multi_scale_conv")
        pooled.append(agg.max(dim=2)[0])
    pooled = torch.stack(pooled, dim=0)                                    # [T,b,emb]
    if taps is not None:
        taps["encoder.pooled_t"] = pooled
    w = F.softmax(sd["encoder.temporal_integration.weights"], dim=0)
    y = torch.einsum("t,tbf->bf", w, pooled)
    y, _ = neuron_step(y, None, neuron_params(sd, "encoder.snn_fc"))       # one step, zero state (fd:485-490)
    return y


def _res_block(sd, pfx, x):
    h = linear(x, sd, pfx + ".fc.0")
    h = F.gelu(_bn_eval(h, sd, pfx + ".fc.1"))
    h = _bn_eval(linear(h, sd, pfx + ".fc.4"), sd, pfx + ".fc.5")
    r = linear(x, sd, pfx + ".res_proj") if (pfx + ".res_proj.weight") in sd else x
    return F.gelu(h + r)


def fd_decoder(sd, f, hp, taps=None):
    d = "distance_decoder"
    x = F.gelu(_bn_eval(linear(f, sd, d + ".fc_in.0"), sd, d + ".fc_in.1"))
    x = _res_block(sd, d + ".residual_blocks.0", x)
    x = _res_block(sd, d + ".residual_blocks.1", x)
    H = hp["num_heads"]
    b, dim = x.shape
    qkv = linear(x, sd, d + ".attention.to_qkv")
    q, k, v = (t.view(b, H, dim // H) for t in qkv.chunk(3, dim=-1))
    a = F.softmax((q * k).sum(-1) * ((dim // H) ** -0.5), dim=-1)          # softmax over HEADS (fd:791-792)
    o = (a.unsqueeze(-1) * v).reshape(b, dim)
    o = linear(o, sd, d + ".attention.to_out.0")
    x = F.layer_norm(o + x, (dim,), sd[d + ".attention.norm.weight"], sd[d + ".attention.norm.bias"], 1e-5)
    if taps is not None:
        taps["decoder.attn_out"] = x
    x = F.gelu(_bn_eval(linear(x, sd, d + ".fc_hidden.0"), sd, d + ".fc_hidden.1"))
    x = linear(x, sd, d + ".fc_distance")
    return F.softplus(x, beta=5.0).squeeze(-1)


def fd_forward(sd, patch, hp, taps=None, force_idx=None):
    """EnhancedSNNDistanceEstimation.forward for 3-D input (fd:853-871) -> distances [b]."""
    f = fd_encoder(sd, patch, hp, taps, force_idx)
    if taps is not None:
        taps["encoder.out"] = f
    return fd_decoder(sd, f, hp, taps)


FN_HP = {"k_values": [24, 18, 12], "emb_dims": 640, "time_steps_enc": 4, "num_heads": 8}
FD_HP = {"k": 32, "k_scales": [8, 16, 32, 48], "emb_dims": 768, "time_steps_enc": 4, "num_heads": 8}
