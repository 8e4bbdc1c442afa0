"""Training-side ops: the surrogate-gradient neuron loop and one trainable layer.

``lif_selfloop_train`` is the T-step self-feeding loop of ``MultiTimeConstantLIFNeuron`` in training mode
(/root/reference/fn/snn_coder.py:87-151, driven as at :318-320) as one differentiable op: hard spikes forward, the soft
surrogate's derivative backward.  ``conv_bn_lif_train`` is fn's basic building block in training mode — 1x1 convolution +
BatchNorm (batch statistics) + that neuron loop (fn/snn_coder.py:225-229,317-320) — forward and backward on HIP kernels
(csrc/train_ops.hip; the convolution's forward and data gradient are the library's exact-f32 GEMM).  The rest of the
training step (attention/softmax/gather backward, pooling, decoder, losses, optimiser, bf16) is not built yet.
"""
import torch

from . import _lib


class _LifSelfLoopTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, membrane_decay, threshold_adapt, refractory_decay, threshold_base, steps):
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError("lif_selfloop_train: expected a CUDA float32 tensor (there is no CPU path)")
        lib = _lib.load()
        shape = x.shape
        ch = shape[1]
        x2 = x.movedim(1, -1).contiguous()                      # channels last: [rows, C]
        rows = x2.numel() // ch
        params = [p.detach().contiguous() for p in (membrane_decay, threshold_adapt, refractory_decay, threshold_base)]
        out = torch.empty_like(x2)
        with torch.cuda.device(x.device):
            _lib.check(lib.sapcu_lif_train_forward(_lib.ptr(x2), rows, ch, int(steps), *[_lib.ptr(p) for p in params],
                                                   _lib.ptr(out), _lib.current_stream()))
        ctx.save_for_backward(x2, *params)
        ctx.steps, ctx.shape = int(steps), shape
        return out.movedim(-1, 1)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x2, md, ta, rd, tb = ctx.saved_tensors
        ch = x2.shape[-1]
        rows = x2.numel() // ch
        g2 = grad_out.movedim(1, -1).contiguous()
        gx = torch.empty_like(x2)
        gp = [torch.empty_like(md) for _ in range(4)]
        nbytes = int(lib.sapcu_lif_train_workspace_bytes(rows, ch))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x2.device)
        with torch.cuda.device(x2.device):
            _lib.check(lib.sapcu_lif_train_backward(_lib.ptr(x2), _lib.ptr(g2), rows, ch, ctx.steps, _lib.ptr(md), _lib.ptr(ta),
                                                    _lib.ptr(rd), _lib.ptr(tb), _lib.ptr(gx), *[_lib.ptr(g) for g in gp],
                                                    _lib.ptr(ws), nbytes, _lib.current_stream()))
        return (gx.movedim(-1, 1), gp[0], gp[1], gp[2], gp[3], None)


def lif_selfloop_train(x, membrane_decay, threshold_adapt, refractory_decay, threshold_base, steps=4):
    """x [B, C] | [B, C, N] | [B, C, N, k] (channel axis 1, as the reference's neuron takes it) -> hard spikes of the last of
    `steps` self-feeding neuron steps; differentiable w.r.t. x and the four raw per-channel parameters."""
    return _LifSelfLoopTrain.apply(x, membrane_decay, threshold_adapt, refractory_decay, threshold_base, steps)


def _ws(lib, rows, ch, k, dev):
    nbytes = int(lib.sapcu_train_workspace_bytes(rows, ch, k))
    return torch.empty((nbytes,), dtype=torch.uint8, device=dev), nbytes


# Arithmetic of the training GEMMs (forward, data gradient, weight gradient): "f32" = the exact-f32 MFMA kernels (the parity
# reference), "bf16" = operands rounded to bf16, f32 accumulation (csrc/train_bf16.hip) — what Trainer(use_amp=True) selects,
# the counterpart of the reference's torch.amp.autocast region (fn/trainer.py:67-83).  Everything else (BatchNorm statistics,
# neuron loops, softmax, losses, the optimiser) stays f32 in both modes, as under autocast.
_GEMM_PRECISION = ["f32"]


class gemm_precision(object):
    """``with gemm_precision("bf16"): loss = ...; loss.backward()`` — covers the forward AND the backward launched inside."""

    def __init__(self, mode):
        if mode not in ("f32", "bf16"):
            raise ValueError("gemm_precision: 'f32' or 'bf16'")
        self.mode = mode

    def __enter__(self):
        self.prev = _GEMM_PRECISION[0]
        _GEMM_PRECISION[0] = self.mode
        return self

    def __exit__(self, *exc):
        _GEMM_PRECISION[0] = self.prev
        return False


def _gemm(lib, a, w, bias, out):
    """out[r, n] = a[r, k] . w[n, k]^T (+ bias): the exact-f32 MFMA kernel (k % 32 == 0), or bf16 operands."""
    r, k = a.shape
    n = w.shape[0]
    if _GEMM_PRECISION[0] == "bf16":
        _lib.check(lib.sapcu_gemm_bf16(_lib.ptr(a), r, k, k, _lib.ptr(w), n, _lib.ptr(bias), _lib.ptr(out), n, _lib.current_stream()))
        return out
    _lib.check(lib.sapcu_gemm_f32(_lib.ptr(a), r, k, k, _lib.ptr(w), n, _lib.ptr(bias), None, 0, _lib.ptr(out), n, None, 0, 0,
                                  _lib.current_stream()))
    return out


def _wgrad(lib, dy, x, rows, cout, cin, dw, db, ws, nbytes, st):
    """dw[cout, cin] = dy^T . x, db = column sums of dy — f32 or bf16 operands like _gemm."""
    if _GEMM_PRECISION[0] == "bf16":
        need = int(lib.sapcu_wgrad_bf16_workspace_bytes(rows, cout, cin))
        if need > nbytes:
            ws = torch.empty((need,), dtype=torch.uint8, device=dy.device)
            nbytes = need
        _lib.check(lib.sapcu_conv1x1_wgrad_bf16(_lib.ptr(dy), cout, _lib.ptr(x), cin, rows, cout, cin, _lib.ptr(dw), _lib.ptr(db),
                                                _lib.ptr(ws), nbytes, st))
        return
    _lib.check(lib.sapcu_conv1x1_wgrad_f32(_lib.ptr(dy), cout, _lib.ptr(x), cin, rows, cout, cin, _lib.ptr(dw), _lib.ptr(db),
                                           _lib.ptr(ws), nbytes, st))


class _ConvBnLifTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, membrane_decay, threshold_adapt, refractory_decay, threshold_base, steps, eps,
                running=None):
        lib = _lib.load()
        rows, cin = x.shape
        cout = weight.shape[0]
        if cin % 32 or cout % 32:
            raise ValueError("conv_bn_lif_train: channel counts must be multiples of 32 (got %d -> %d)" % (cin, cout))
        dev = x.device
        x = x.contiguous()
        w = weight.detach().contiguous()
        prm = [p.detach().contiguous() for p in (membrane_decay, threshold_adapt, refractory_decay, threshold_base)] if int(steps) > 0 else []
        y = torch.empty((rows, cout), dtype=torch.float32, device=dev)
        z = torch.empty_like(y)
        mean, var, invstd = (torch.empty((cout,), dtype=torch.float32, device=dev) for _ in range(3))
        ws, nbytes = _ws(lib, rows, cout, 0, dev)
        with torch.cuda.device(dev):
            _gemm(lib, x, w, bias.detach().contiguous(), y)
            _lib.check(lib.sapcu_bn_train_forward(_lib.ptr(y), rows, cout, _lib.ptr(gamma.detach().contiguous()),
                                                  _lib.ptr(beta.detach().contiguous()), float(eps), _lib.ptr(z), _lib.ptr(mean),
                                                  _lib.ptr(var), _lib.ptr(invstd), _lib.ptr(ws), nbytes, _lib.current_stream()))
            out = z
            if int(steps) > 0:
                out = torch.empty_like(y)
                _lib.check(lib.sapcu_lif_train_forward(_lib.ptr(z), rows, cout, int(steps), *[_lib.ptr(p) for p in prm],
                                                       _lib.ptr(out), _lib.current_stream()))
        if running is not None:                     # nn.BatchNorm's train()-mode bookkeeping: momentum update, unbiased variance
            r_mean, r_var, n_tracked, momentum = running
            if rows < 2:
                raise ValueError("BatchNorm in training mode needs more than 1 value per channel (got %d rows)" % rows)
            with torch.no_grad():
                r_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
                r_var.mul_(1.0 - momentum).add_(var, alpha=momentum * rows / (rows - 1.0))
                if n_tracked is not None:
                    n_tracked.add_(1)
        ctx.save_for_backward(x, w, gamma.detach().contiguous(), y, z, mean, invstd, *prm)
        ctx.steps = int(steps)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x, w, gamma, y, z, mean, invstd, *nrn = ctx.saved_tensors
        rows, cin = x.shape
        cout = w.shape[0]
        dev = x.device
        g = grad_out.contiguous()
        dz, dy, dx = torch.empty_like(y), torch.empty_like(y), torch.empty_like(x)
        gp = [None] * 4                                   # no neuron (steps == 0): the four parameter slots are placeholders
        dgamma, dbeta, dbias = (torch.empty((cout,), dtype=torch.float32, device=dev) for _ in range(3))
        dw = torch.empty_like(w)
        ws, nbytes = _ws(lib, rows, cout, cin, dev)
        with torch.cuda.device(dev):
            st = _lib.current_stream()
            if ctx.steps > 0:
                md, ta, rd, tb = nrn
                gp = [torch.empty_like(md) for _ in range(4)]
                lws_bytes = int(lib.sapcu_lif_train_workspace_bytes(rows, cout))
                lws = torch.empty((lws_bytes,), dtype=torch.uint8, device=dev)
                _lib.check(lib.sapcu_lif_train_backward(_lib.ptr(z), _lib.ptr(g), rows, cout, ctx.steps, _lib.ptr(md), _lib.ptr(ta),
                                                        _lib.ptr(rd), _lib.ptr(tb), _lib.ptr(dz), *[_lib.ptr(t) for t in gp],
                                                        _lib.ptr(lws), lws_bytes, st))
            else:
                dz = g
            _lib.check(lib.sapcu_bn_train_backward(_lib.ptr(y), _lib.ptr(dz), rows, cout, _lib.ptr(gamma), _lib.ptr(mean),
                                                   _lib.ptr(invstd), _lib.ptr(dy), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws),
                                                   nbytes, st))
            _wgrad(lib, dy, x, rows, cout, cin, dw, dbias, ws, nbytes, st)
            if ctx.needs_input_grad[0]:
                _gemm(lib, dy, w.t().contiguous(), None, dx)        # dx[r, cin] = dy[r, cout] . (W^T)[cin, cout]^T
            else:
                dx = None
        return (dx, dw, dbias, dgamma, dbeta, gp[0], gp[1], gp[2], gp[3], None, None, None)


def conv_bn_lif_train(x, weight, bias, gamma, beta, membrane_decay, threshold_adapt, refractory_decay, threshold_base,
                      steps=4, eps=1e-5, running=None):
    """fn's basic layer in TRAINING mode (fn/snn_coder.py:225-229 + 317-320): x [rows, c_in] (channels last) ->
    hard spikes [rows, c_out] of `steps` neuron steps on BatchNorm_train(x . W^T + b).  Differentiable w.r.t. x and all nine
    parameter tensors.  running: None, or (running_mean, running_var, num_batches_tracked | None, momentum) updated in place
    the way nn.BatchNorm does in train() mode."""
    w2 = weight.reshape(weight.shape[0], -1)                      # Conv1d/Conv2d 1x1 weights [c_out, c_in, 1(,1)]
    return _ConvBnLifTrain.apply(x, w2, bias, gamma, beta, membrane_decay, threshold_adapt, refractory_decay, threshold_base,
                                 steps, eps, running)


class _SoftmaxAgg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, pe, v, idx, m, sqrt_hd, keep):
        lib = _lib.load()
        pts, d = v.shape
        kk = a.shape[0] // pts
        a, pe, v = a.contiguous(), pe.contiguous(), v.contiguous()
        idx = idx.contiguous().to(torch.int32)
        keep = keep.contiguous() if keep is not None else None
        res = torch.empty((pts, d), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            _lib.check(lib.sapcu_softmax_agg_forward(_lib.ptr(a), _lib.ptr(pe), _lib.ptr(v), d, _lib.ptr(idx),
                                                     _lib.ptr(keep) if keep is not None else None, pts, int(m), kk, d,
                                                     float(sqrt_hd), _lib.ptr(res), _lib.current_stream()))
        ctx.save_for_backward(a, pe, v, idx)
        ctx.keep = keep
        ctx.m, ctx.sqrt_hd = int(m), float(sqrt_hd)
        return res

    @staticmethod
    def backward(ctx, grad_res):
        lib = _lib.load()
        a, pe, v, idx = ctx.saved_tensors
        pts, d = v.shape
        kk = a.shape[0] // pts
        ga, gpe, gv = torch.empty_like(a), torch.empty_like(pe), torch.empty_like(v)
        with torch.cuda.device(a.device):
            _lib.check(lib.sapcu_softmax_agg_backward(_lib.ptr(a), _lib.ptr(pe), _lib.ptr(v), d, _lib.ptr(idx),
                                                      _lib.ptr(ctx.keep) if ctx.keep is not None else None,
                                                      _lib.ptr(grad_res.contiguous()), pts, ctx.m, kk, d, ctx.sqrt_hd, _lib.ptr(ga),
                                                      _lib.ptr(gpe), _lib.ptr(gv), d, _lib.current_stream()))
        return ga, gpe, gv, None, None, None, None


def softmax_agg(a, pe, v, idx, m, sqrt_hd, keep=None):
    """fn/snn_coder.py:379-389 on edge rows: a, pe [P*k, d]; v [P, d]; idx [P*k] in-patch neighbour indices (m points per
    patch) -> res [P, d] = sum_j softmax_j(a / sqrt_hd) * keep_j * (v[nbr_j] + pe_j).  keep: None, or [P*k, d] holding 0 or
    1/(1-p) (the attention dropout of train() mode, fn:383).  Differentiable w.r.t. a, pe and v."""
    return _SoftmaxAgg.apply(a, pe, v, idx, m, sqrt_hd, keep)


def dropout_keep(shape, p, device, generator=None):
    """The scale mask nn.Dropout(p) multiplies with in train() mode: 0 with probability p, else 1/(1-p)."""
    return torch.empty(shape, dtype=torch.float32, device=device).bernoulli_(1.0 - p, generator=generator).div_(1.0 - p)


def conv_bn_train(x, weight, bias, gamma, beta, eps=1e-5, running=None):
    """1x1 convolution + BatchNorm in training mode, no neuron (fn's fc_gamma2 / out_proj / fc2): x [rows, c_in] -> [rows, c_out]."""
    w2 = weight.reshape(weight.shape[0], -1)
    return _ConvBnLifTrain.apply(x, w2, bias, gamma, beta, None, None, None, None, 0, eps, running)


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, group=None):
        lib = _lib.load()
        src = src.contiguous()
        index = index.contiguous().to(torch.int64)
        rows, d = index.shape[0], src.shape[1]
        out = torch.empty((rows, d), dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            _lib.check(lib.sapcu_gather_rows(_lib.ptr(src), d, _lib.ptr(index), rows, d, _lib.ptr(out), _lib.current_stream()))
        ctx.save_for_backward(index)
        ctx.src_rows = src.shape[0]
        ctx.group = group
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        (index,) = ctx.saved_tensors
        g = grad_out.contiguous()
        rows, d = g.shape
        gsrc = torch.empty((ctx.src_rows, d), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            if ctx.group is not None:           # patch-structured index: the deterministic segmented sum
                gs, gr = ctx.group
                _lib.check(lib.sapcu_scatter_add_rows_grouped(_lib.ptr(g), _lib.ptr(index), rows, d, _lib.ptr(gsrc), d, ctx.src_rows,
                                                              int(gs), int(gr), None, _lib.current_stream()))
            else:                               # arbitrary index: float atomics (summation order not fixed)
                _lib.check(lib.sapcu_scatter_add_rows(_lib.ptr(g), _lib.ptr(index), rows, d, _lib.ptr(gsrc), d, ctx.src_rows,
                                                      _lib.current_stream()))
        return gsrc, None, None


def gather_rows(src, index, group=None):
    """out[r] = src[index[r]] on [rows, d] tensors (index_points, fn/snn_coder.py:19-29); backward = scatter-add.
    group = (destination rows per group, index rows per group) when the index is patch-structured — rows r of group g (a
    patch's m*k edge rows) only point into source rows [g*gs, (g+1)*gs) (the patch's m points): the backward is then the
    deterministic segmented sum (sapcu_scatter_add_rows_grouped) instead of float atomics."""
    return _GatherRows.apply(src, index, group)


def _pad_channels(t, mult=32):
    """zero-pad the channel (last) axis to a multiple of 32: the GEMM kernels step the reduction axis by 32"""
    c = t.shape[-1]
    pad = (-c) % mult
    return torch.nn.functional.pad(t, (0, pad)) if pad else t


def _running(p, bn, momentum):
    """(running_mean, running_var, num_batches_tracked, momentum) of BatchNorm `bn` when p carries its buffers, else None."""
    if momentum is None or (bn + ".running_mean") not in p:
        return None
    return (p[bn + ".running_mean"], p[bn + ".running_var"], p.get(bn + ".num_batches_tracked"), float(momentum))


def transformer_block_train(p, xyz, features, knn_idx, time_steps=4, num_heads=8, eps=1e-5, momentum=None, attn_dropout=0.0,
                            generator=None):
    """One ``MultiHeadSNNTransformerBlock`` in TRAINING mode (fn/snn_coder.py:294-396, dropout 0) on channels-last rows.

    p: dict of the block's parameters under the reference's names (``fc1.0.weight``, ``fc1.1.weight`` (BN gamma),
    ``snn1.membrane_decay`` ...), xyz [B, N, 3], features [B, N, d_points], knn_idx [B, N, k] in-patch neighbours ->
    [B, N, d_points].  Every 1x1 convolution + BatchNorm (+ neuron loop), the neighbour gathers and the softmax-aggregate
    are the HIP ops of this module; the adds/subtractions between them are torch glue."""
    B, N, _ = xyz.shape
    k = knn_idx.shape[-1]
    dev = xyz.device
    P = B * N
    base = (torch.arange(B, device=dev) * N).view(B, 1, 1)
    nbr = (knn_idx.to(torch.int64) + base).reshape(P * k)                 # global row of each edge's neighbour
    ptr = torch.arange(P, device=dev).repeat_interleave(k)                # ... and of its centre point
    feat = features.reshape(P, -1)
    xyzr = xyz.reshape(P, 3)

    def layer(x, conv, bn, snn=None):
        w = _pad_channels(p[conv + ".weight"].reshape(p[conv + ".weight"].shape[0], -1))
        args = (_pad_channels(x), w, p[conv + ".bias"], p[bn + ".weight"], p[bn + ".bias"])
        run = _running(p, bn, momentum)
        if snn is None:
            return conv_bn_train(*args, eps=eps, running=run)
        return conv_bn_lif_train(*args, p[snn + ".membrane_decay"], p[snn + ".threshold_adapt"], p[snn + ".refractory_decay"],
                                 p[snn + ".threshold_base"], steps=time_steps, eps=eps, running=run)

    x = layer(feat, "fc1.0", "fc1.1", "snn1")
    q = layer(x, "w_qs.0", "w_qs.1", "snn_q")
    kf = layer(x, "w_ks.0", "w_ks.1", "snn_k")
    v = layer(x, "w_vs.0", "w_vs.1", "snn_v")
    grp = (N, N * k)                                                      # every index below stays inside its patch
    pos_diff = gather_rows(xyzr, ptr, grp) - gather_rows(xyzr, nbr, grp)  # [P*k, 3]
    pe = layer(pos_diff, "fc_delta.0", "fc_delta.1", "snn_delta")
    pe = layer(pe, "fc_delta2.0", "fc_delta2.1", "snn_delta2")
    attn_in = gather_rows(q, ptr, grp) - gather_rows(kf, nbr, grp) + pe
    a = layer(attn_in, "fc_gamma.0", "fc_gamma.1", "snn_gamma")
    a = layer(a, "fc_gamma2.0", "fc_gamma2.1")
    d_model = a.shape[1]
    keep = dropout_keep((P * k, d_model), attn_dropout, dev, generator) if attn_dropout > 0 else None
    res = softmax_agg(a, pe, v, knn_idx.reshape(P * k), N, float((d_model // num_heads) ** 0.5), keep)
    res = layer(res, "out_proj.0", "out_proj.1")
    res = layer(res, "fc2.0", "fc2.1") + feat
    return res.view(B, N, -1)


class _LinearTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x, w = x.contiguous(), weight.detach().contiguous()
        out = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _gemm(lib, x, w, bias.detach().contiguous(), out)
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        rows, cin = x.shape
        cout = w.shape[0]
        g = grad_out.contiguous()
        dw, db, dx = torch.empty_like(w), torch.empty((cout,), dtype=torch.float32, device=x.device), torch.empty_like(x)
        ws, nbytes = _ws(lib, rows, cout, cin, x.device)
        with torch.cuda.device(x.device):
            _wgrad(lib, g, x, rows, cout, cin, dw, db, ws, nbytes, _lib.current_stream())
            _gemm(lib, g, w.t().contiguous(), None, dx)
        return dx, dw, db


def linear_train(x, weight, bias):
    """y = x . W^T + b with HIP forward / weight-gradient / data-gradient kernels.  Both channel counts are zero-padded to
    multiples of 32 here (the exact-f32 GEMM steps its reduction axis by 32: c_in forward, c_out in the data gradient)."""
    cout = weight.shape[0]
    w = _pad_channels(weight.reshape(cout, -1))
    pad_o = (-cout) % 32
    if pad_o:
        w = torch.nn.functional.pad(w, (0, 0, 0, pad_o))
        bias = torch.nn.functional.pad(bias, (0, pad_o))
    return _LinearTrain.apply(_pad_channels(x), w, bias)[:, :cout]


class _GroupMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m):
        lib = _lib.load()
        x = x.contiguous()
        c = x.shape[1]
        groups = x.shape[0] // m
        out = torch.empty((groups, c), dtype=torch.float32, device=x.device)
        arg = torch.empty((groups, c), dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.sapcu_group_max_forward(_lib.ptr(x), groups, int(m), c, _lib.ptr(out), _lib.ptr(arg), _lib.current_stream()))
        ctx.save_for_backward(arg)
        ctx.m = int(m)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        groups, c = arg.shape
        gx = torch.empty((groups * ctx.m, c), dtype=torch.float32, device=arg.device)
        with torch.cuda.device(arg.device):
            _lib.check(lib.sapcu_group_max_backward(_lib.ptr(grad_out.contiguous()), _lib.ptr(arg), groups, ctx.m, c, _lib.ptr(gx),
                                                    _lib.current_stream()))
        return gx, None


def group_max(x, m):
    """max over the m points of each patch: x [groups*m, c] -> [groups, c]; gradient to the first arg-max (fn:472)."""
    return _GroupMax.apply(x, m)


def inpatch_knn(xyz, k):
    """In-patch kNN of fn's blocks (fn/snn_coder.py:31-39) on the device: xyz [B, N, 3] -> int32 [B, N, k]."""
    lib = _lib.load()
    B, N, _ = xyz.shape
    x = xyz.contiguous()
    idx = torch.empty((B, N, k), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(lib.sapcu_patch_knn(_lib.ptr(x), B, N, 3, 3, k, _lib.ptr(idx), _lib.current_stream()))
    return idx


def fn_train_forward(p, points, k_values=(24, 18, 12), time_steps_enc=4, num_heads=8, eps=1e-5, knn=None, momentum=None,
                     attn_dropout=0.0, decoder_dropout=0.0, generator=None):
    """``ImprovedSNNNormalEstimation.forward`` in TRAINING mode (fn/snn_coder.py:430-476, 542-549, 670-699; every dropout off:
    dropout is random) on the HIP training ops: points [B, M, 3] -> unit normals [B, 3], differentiable w.r.t. every tensor
    of p (the model's parameters under the reference's state_dict names).  GELU, LayerNorm(3), the concatenation, the
    residual adds and the final normalisation are torch glue on small tensors.  momentum: when set and p also carries the
    BatchNorm buffers, the running statistics are updated as train() mode does (0.1 in the reference).  attn_dropout /
    decoder_dropout: the blocks' attention dropout (fn:286, 0.1 in the reference) and the decoder's (fn:532-533), drawn from
    torch's generator on the device."""
    B, N, _ = points.shape
    P = B * N
    F = torch.nn.functional

    def sub(prefix):
        return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}

    enc = sub("encoder.")
    feat = conv_bn_lif_train(_pad_channels(points.reshape(P, 3)), _pad_channels(enc["conv1.0.weight"].reshape(64, -1)), enc["conv1.0.bias"],
                             enc["conv1.1.weight"], enc["conv1.1.bias"], enc["snn_init.membrane_decay"], enc["snn_init.threshold_adapt"],
                             enc["snn_init.refractory_decay"], enc["snn_init.threshold_base"], steps=time_steps_enc, eps=eps,
                             running=_running(enc, "conv1.1", momentum))
    feats, cur = [], feat.view(B, N, 64)
    for i, kk in enumerate(k_values):
        k = min(kk, N)
        idx = knn[i] if knn is not None else inpatch_knn(points, k)
        cur = transformer_block_train(sub("encoder.trans%d." % (i + 1)), points, cur, idx, time_steps=4, num_heads=num_heads, eps=eps,
                                      momentum=momentum, attn_dropout=attn_dropout, generator=generator)
        feats.append(cur)
    multi = torch.cat(feats, dim=2).reshape(P, 192)
    g = conv_bn_lif_train(multi, enc["conv_final.0.weight"].reshape(enc["conv_final.0.weight"].shape[0], -1), enc["conv_final.0.bias"],
                          enc["conv_final.1.weight"], enc["conv_final.1.bias"], enc["snn_final.membrane_decay"],
                          enc["snn_final.threshold_adapt"], enc["snn_final.refractory_decay"], enc["snn_final.threshold_base"],
                          steps=time_steps_enc, eps=eps, running=_running(enc, "conv_final.1", momentum))
    x = linear_train(group_max(g, N), enc["fc_out.weight"], enc["fc_out.bias"])
    dec = sub("decoder.")
    lin = sorted({int(k.split(".")[1]) for k in dec if k.startswith("mlp.") and k.endswith(".weight") and dec[k].dim() == 2})
    for li in lin:                                               # Linear, BatchNorm1d, GELU (, Dropout off)
        x = F.gelu(conv_bn_train(_pad_channels(x), _pad_channels(dec["mlp.%d.weight" % li]), dec["mlp.%d.bias" % li],
                                 dec["mlp.%d.weight" % (li + 1)], dec["mlp.%d.bias" % (li + 1)], eps=eps,
                                 running=_running(dec, "mlp.%d" % (li + 1), momentum)))
        if decoder_dropout > 0:
            x = x * dropout_keep(x.shape, decoder_dropout, x.device, generator)
    x = linear_train(x, dec["fc_out.weight"], dec["fc_out.bias"])
    x = F.layer_norm(x, (3,), dec["norm_out.weight"], dec["norm_out.bias"], 1e-5)
    return F.normalize(x, dim=1)


def angular_loss_with_consistency(pred_normals, gt_normals, xyz=None, temperature=0.1, alpha=0.1, consistency_weight=0.15,
                                  k_neighbors=8):
    """enhanced_angular_loss_with_consistency (fn/snn_coder.py:587-625) + normal_consistency_loss (fn:557-583):
    -> (loss, mean confidence).  pred/gt [B, 3] or [B, N, 3]; xyz [B, N, 3] (patch centres) or None."""
    F = torch.nn.functional
    pred = pred_normals.reshape(-1, 3)
    gt = gt_normals.reshape(-1, 3)
    cos = F.cosine_similarity(pred, gt, dim=1)
    err = torch.acos(torch.clamp(cos, -1 + 1e-6, 1 - 1e-6))
    conf = torch.sigmoid(err.detach() / temperature)
    loss = (err * conf + alpha * (conf - 0.5) ** 2).mean()
    if xyz is not None and consistency_weight > 0:
        B, N, _ = xyz.shape
        k = min(k_neighbors + 1, N)
        # dists.argsort()[:, :, 1:k+1]: the k nearest after the first entry (the point itself, distance 0)
        nbr = inpatch_knn(xyz.detach().float(), k)[:, :, 1:].to(torch.int64)
        full = pred_normals.unsqueeze(1).expand(B, N, 3) if pred_normals.dim() == 2 else pred_normals.view(B, N, 3)
        nb = torch.gather(full.unsqueeze(1).expand(B, N, N, 3), 2, nbr.unsqueeze(-1).expand(B, N, nbr.shape[2], 3))
        loss = loss + consistency_weight * (1 - F.cosine_similarity(full.unsqueeze(2), nb, dim=-1)).mean()
    return loss, conf.mean()
