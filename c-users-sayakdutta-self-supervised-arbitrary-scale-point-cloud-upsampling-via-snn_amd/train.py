"""Training-side ops: so far the surrogate-gradient neuron loop.

``lif_selfloop_train`` is the T-step self-feeding loop of ``MultiTimeConstantLIFNeuron`` in training mode
(/root/reference/fn/snn_coder.py:87-151, driven as at :318-320) as one differentiable op: hard spikes forward, the soft
surrogate's derivative backward, both as HIP kernels (csrc/train_ops.hip).  The rest of the training step (the blocks'
backward, BatchNorm in training mode, losses, optimiser) is not built yet.
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
