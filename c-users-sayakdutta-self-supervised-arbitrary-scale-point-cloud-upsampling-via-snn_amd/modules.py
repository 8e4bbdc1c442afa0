"""Drop-in ``nn.Module`` shells for the reference's two networks, running on libsapcu_hip.so.

They keep the reference's constructor signatures, ``forward`` / ``reset_states`` behaviour and —
name for name, shape for shape — its ``state_dict`` layout, so ``CheckpointIO.load`` /
``load_state_dict(strict=True)`` of a reference checkpoint works unchanged:

* ``ImprovedSNNNormalEstimation``   <- /root/reference/fn/snn_coder.py:627-738
* ``EnhancedSNNDistanceEstimation`` <- /root/reference/fd/snn_coder.py:805-893

The torch modules inside are parameter containers only; ``forward`` packs the parameters once
(re-packing when any tensor's version counter changes), then calls the C ABI on the current
stream.  Inference (eval) only: there is no autograd path and no CPU path — a module whose
parameters are not on a ROCm device raises.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib, packing


class NeuronParams(nn.Module):
    """Learnable per-channel neuron parameters (LIF: 4, EIF: 6); inits as fn/snn_coder.py:64-80."""

    def __init__(self, channels, eif=False):
        super().__init__()
        mk = lambda v: nn.Parameter(torch.full((channels,), float(v)))
        self.membrane_decay = mk(0.9)
        self.threshold_adapt = mk(0.01)
        self.refractory_decay = mk(0.5)
        self.threshold_base = mk(1.0)
        if eif:
            self.delta_T = mk(1.0)
            self.theta_rh = mk(0.8)


def _conv_bn(cin, cout, nd, bias=True, act=None):
    conv = (nn.Conv1d if nd == 1 else nn.Conv2d)(cin, cout, 1, bias=bias)
    bn = (nn.BatchNorm1d if nd == 1 else nn.BatchNorm2d)(cout)
    return nn.Sequential(*([conv, bn] + ([act] if act is not None else [])))


class _Bag(nn.Module):
    """Plain named container."""


class _HipModel(nn.Module):
    KIND = None

    def __init__(self):
        super().__init__()
        self._handle = None
        self._packed_key = None
        self._ws = None

    # -- engine -------------------------------------------------------------------------------
    def _hparams(self):
        raise NotImplementedError

    def _pack(self, sd):
        raise NotImplementedError

    def _state_key(self):
        ts = list(self.parameters()) + list(self.buffers())
        return tuple((t.data_ptr(), t._version) for t in ts)

    def _device(self):
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("sapcu_amd models run on a ROCm GPU only (parameters are on %s); "
                               "there is no CPU path — call .to('cuda')" % p.device)
        return p.device

    def _engine(self):
        dev = self._device()
        key = self._state_key()
        if self._handle is not None and key == self._packed_key:
            return self._handle
        lib = _lib.load()
        self._release()
        blob, directory = self._pack({k: v for k, v in self.state_dict().items()})
        blob_dev = torch.from_numpy(blob).to(dev)
        hp = np.asarray(self._hparams(), dtype=np.int32)
        handle = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.sapcu_model_create(
                self.KIND, hp.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), hp.size, _lib.ptr(blob_dev),
                blob.size, directory.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), directory.size,
                ctypes.byref(handle)))
        self._handle, self._packed_key = handle, key
        return handle

    def _release(self):
        if self._handle is not None:
            try:
                _lib.load().sapcu_model_destroy(self._handle)
            except Exception:
                pass
            object.__setattr__(self, "_handle", None)     # (plain attribute: nn.Module.__setattr__ is not usable at interpreter exit)

    def __del__(self):
        try:
            self._release()
        except Exception:       # interpreter shutdown: modules this needs may already be gone; the process is ending anyway
            pass

    def _workspace(self, handle, b, m, dev):
        need = _lib.load().sapcu_workspace_bytes(handle, b, m)
        if need < 0:
            _lib.check(int(need))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
        return self._ws

    def gemm_mode(self):
        """(split_f16, range_overflows): which GEMM kernels the handle uses and how many activation tiles left
        the f16 range so far (must be 0; sapcu.h sapcu_model_gemm_mode)."""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.load().sapcu_model_gemm_mode(self._engine(), ctypes.byref(a), ctypes.byref(b)))
        return bool(a.value), b.value

    def fused_blocks(self, m_pts):
        """Bit mask of the stages that run as fused LDS-resident kernels for patches of `m_pts` points (sapcu.h
        sapcu_model_fused_blocks): fn bit l = transformer block l+1 on csrc/fn_edge_chain.hip — only for the reference's
        (d, k) pairs (128, 24), (256, 18), (512, 12), other k_values or patches smaller than k run the five-kernel chain;
        fd bit 0 = the whole encoder on csrc/fd_encoder.hip.  Speed and workspace differ, results do not."""
        mask = ctypes.c_int(0)
        _lib.check(_lib.load().sapcu_model_fused_blocks(self._engine(), int(m_pts), ctypes.byref(mask)))
        return mask.value

    @staticmethod
    def _taps_array(names, taps):
        if not taps:
            return None
        arr = (ctypes.c_void_p * len(names))()
        for i, n in enumerate(names):
            t = taps.get(n)
            arr[i] = t.data_ptr() if t is not None else None
        return arr

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("sapcu_amd models are inference-only (eval mode); training backward "
                                      "is a later row of SURVEY.md §8f")
        return super().train(False)

    def _as_patches(self, x):
        """[B,M,3] | [B,3,M] -> contiguous f32 [B,M,3]; a dim-1 of size 3 means channels-first
        (fn/snn_coder.py:441, fd/snn_coder.py:393)."""
        if x.dim() != 3:
            raise ValueError("expected a 3-D or 4-D point tensor, got shape %s" % (tuple(x.shape),))
        if x.shape[1] == 3:
            x = x.permute(0, 2, 1)
        if x.shape[2] != 3:
            raise ValueError("last dimension must be 3 (xyz), got shape %s" % (tuple(x.shape),))
        if not (1 <= x.shape[1] <= 128):
            raise ValueError("patch size must be in 1..128 points, got %d" % x.shape[1])
        return x.to(dtype=torch.float32).contiguous()


class ImprovedSNNNormalEstimation(_HipModel):
    """Normal estimator: SNN point-transformer encoder + MLP decoder -> unit normal per patch."""
    KIND = _lib.KIND_FN

    def __init__(self, k_values=[20, 20, 16], emb_dims=1024, time_steps_enc=8, time_steps_dec=12, num_heads=4,
                 use_snn_decoder=False, decoder_dropout=0.1):
        super().__init__()
        if use_snn_decoder:
            raise NotImplementedError("use_snn_decoder=True is the reference's legacy decoder (fn/snn_coder.py:481-514); "
                                      "not built")
        self.use_snn_decoder = False
        self.k_values, self.emb_dims, self.time_steps_enc, self.num_heads = list(k_values), emb_dims, time_steps_enc, num_heads
        enc = _Bag()
        enc.conv1 = _conv_bn(3, 64, 1)
        enc.snn_init = NeuronParams(64)
        for i, d in enumerate((128, 256, 512)):
            blk = _Bag()
            blk.fc1 = _conv_bn(64, d, 1)
            blk.snn1 = NeuronParams(d)
            blk.fc2 = _conv_bn(d, 64, 1)
            blk.fc_delta = _conv_bn(3, d, 2)
            blk.snn_delta = NeuronParams(d)
            blk.fc_delta2 = _conv_bn(d, d, 2)
            blk.snn_delta2 = NeuronParams(d)
            blk.fc_gamma = _conv_bn(d, d, 2)
            blk.snn_gamma = NeuronParams(d)
            blk.fc_gamma2 = _conv_bn(d, d, 2)
            for proj, snn in (("w_qs", "snn_q"), ("w_ks", "snn_k"), ("w_vs", "snn_v")):
                setattr(blk, proj, _conv_bn(d, d, 1))
                setattr(blk, snn, NeuronParams(d))
            blk.out_proj = _conv_bn(d, d, 1)
            setattr(enc, "trans%d" % (i + 1), blk)
        enc.conv_final = _conv_bn(192, emb_dims, 1)
        enc.snn_final = NeuronParams(emb_dims)
        enc.fc_out = nn.Linear(emb_dims, 2048)
        self.encoder = enc
        dec = _Bag()
        layers, cin, self._dec_linear_idx = [], 2048, []
        for h in (1024, 512, 256):
            self._dec_linear_idx.append(len(layers))
            layers += [nn.Linear(cin, h), nn.BatchNorm1d(h), nn.GELU()]
            if decoder_dropout > 0:
                layers.append(nn.Dropout(decoder_dropout))
            cin = h
        dec.mlp = nn.Sequential(*layers)
        dec.fc_out = nn.Linear(256, 3)
        dec.norm_out = nn.LayerNorm(3)
        self.decoder = dec
        # reference quirk (fn/snn_coder.py:47-59): in-patch neighbour tables are cached by tensor
        # SHAPE, never cleared by reset_states().  'reference' replays it, 'fresh' recomputes.
        self.knn_cache_mode = "reference"
        self._knn_cache = {}
        self.decoder_dropout = float(decoder_dropout)
        self.attn_dropout = 0.1                   # MultiHeadSNNTransformerBlock's default, which the encoder never overrides (fn:213, 421-423)
        super().train(False)

    def train(self, mode=True):
        """fn has a training path (row f-4, sapcu_amd/train.py): train() switches forward() to hard spikes, batch statistics,
        dropout and autograd through the HIP training ops."""
        return nn.Module.train(self, mode)

    def _train_forward(self, point_cloud):
        from . import train as T
        if point_cloud.ndim == 4:
            B, N, M, C = point_cloud.shape
            return self._train_forward(point_cloud.reshape(B * N, M, C)).view(B, N, 3)
        x = self._as_patches(point_cloud)
        self._device()
        p = dict(self.named_parameters())
        p.update(dict(self.named_buffers()))
        return T.fn_train_forward(p, x, tuple(self.k_values), self.time_steps_enc, self.num_heads, momentum=0.1,
                                  attn_dropout=self.attn_dropout, decoder_dropout=self.decoder_dropout)

    def compute_loss(self, pred_normals, gt_normals, xyz=None, consistency_weight=0.15, k_neighbors=8):
        """fn/snn_coder.py:701-724: (loss, {'total_loss', 'confidence'}) — angular loss with confidence weighting plus the
        neighbour-consistency term (fn:557-625).  A few dozen torch ops on [B(,N),3] tensors; the neighbour search of the
        consistency term is sapcu_patch_knn."""
        from . import train as T
        if xyz is not None and xyz.ndim == 4:
            xyz = xyz.mean(dim=2)
        loss, confidence = T.angular_loss_with_consistency(pred_normals, gt_normals, xyz, consistency_weight=consistency_weight,
                                                           k_neighbors=k_neighbors)
        return loss, {"total_loss": loss.item(), "confidence": confidence.item()}

    def _hparams(self):
        return list(self.k_values) + [self.emb_dims, self.time_steps_enc, self.num_heads]

    def _pack(self, sd):
        return packing.pack_fn(sd, tuple(self._dec_linear_idx))

    def _knn_table_len(self, b, m):
        return sum(b * m * min(k, m) for k in self.k_values)

    def forward(self, point_cloud, taps=None, knn_in=None):
        """[B,M,3] | [B,3,M] -> [B,3];  [B,N,M,3] -> [B,N,3]  (fn/snn_coder.py:670-699).
        knn_in (ours, optional): explicit in-patch neighbour tables (flat int32, three [B,M,k] blocks) — used by
        Generator3D6 to run several reference batches in one device pass; bypasses the shape-keyed cache."""
        if self.training:
            return self._train_forward(point_cloud)
        if point_cloud.ndim == 4:
            B, N, M, C = point_cloud.shape
            return self.forward(point_cloud.reshape(B * N, M, C), taps).view(B, N, 3)
        x = self._as_patches(point_cloud)
        handle = self._engine()
        lib = _lib.load()
        dev = x.device
        b, m = x.shape[0], x.shape[1]
        out = torch.empty((b, 3), dtype=torch.float32, device=dev)
        if b == 0:
            return out
        knn_out = None
        if knn_in is not None:
            if knn_in.numel() != self._knn_table_len(b, m) or knn_in.dtype != torch.int32 or knn_in.device != dev:
                raise ValueError("knn_in: expected %d int32 entries on %s" % (self._knn_table_len(b, m), dev))
        elif self.knn_cache_mode == "reference":
            key = (b, m)
            if key in self._knn_cache:
                knn_in = self._knn_cache[key]
            else:
                knn_out = torch.empty(self._knn_table_len(b, m), dtype=torch.int32, device=dev)
        elif self.knn_cache_mode != "fresh":
            raise ValueError("knn_cache_mode must be 'reference' or 'fresh'")
        ws = self._workspace(handle, b, m, dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sapcu_fn_forward(handle, _lib.ptr(x), b, m, _lib.ptr(knn_in), _lib.ptr(knn_out), _lib.ptr(out),
                                            _lib.ptr(ws), ws.numel(), self._taps_array(_lib.FN_TAPS, taps),
                                            _lib.current_stream()))
        if knn_out is not None:
            self._knn_cache[(b, m)] = knn_out
            if len(self._knn_cache) > 32:          # KNNCache.max_size eviction, fn/snn_coder.py:56-57
                del self._knn_cache[next(iter(self._knn_cache))]
        return out

    def tiled_knn_tables(self, b, m, times):
        """The cached tables of batch shape (b, m) repeated `times` along the batch axis, in the flat layout forward()
        takes as knn_in for a batch of times*b patches (every sub-batch replays the cached batch's neighbours, which is
        what the reference's shape-keyed cache does to consecutive batches of one shape)."""
        flat = self._knn_cache[(b, m)]
        out, off = [], 0
        for k in self.k_values:
            cnt = b * m * min(k, m)
            out.append(flat[off:off + cnt].repeat(times))
            off += cnt
        return torch.cat(out)

    def knn_tables(self, b, m):
        """The cached in-patch neighbour tables for batch shape (b, m): three int32 [b,m,k] tensors."""
        flat = self._knn_cache.get((b, m))
        if flat is None:
            return None
        out, off = [], 0
        for k in self.k_values:
            kk = min(k, m)
            out.append(flat[off:off + b * m * kk].view(b, m, kk))
            off += b * m * kk
        return out

    def reset_states(self):
        """Clears SNN state managers in the reference (fn/snn_coder.py:726-738) — which hold nothing on
        the inference path — and notably NOT the kNN cache.  Same here."""


class EnhancedSNNDistanceEstimation(_HipModel):
    """Distance estimator: multi-scale EdgeConv + EIF/LIF encoder over T steps + MLP decoder."""
    KIND = _lib.KIND_FD

    def __init__(self, k=20, emb_dims=512, time_steps_enc=5, time_steps_dec=8, num_heads=4, dropout=0.1,
                 use_snn_decoder=False, k_scales=[10, 20, 40]):
        super().__init__()
        if use_snn_decoder:
            raise NotImplementedError("use_snn_decoder=True is the reference's legacy decoder (fd/snn_coder.py:497-664); "
                                      "not built")
        self.use_snn_decoder = False
        self.k, self.emb_dims, self.time_steps_enc, self.num_heads, self.k_scales = k, emb_dims, time_steps_enc, num_heads, list(k_scales)
        act = lambda: nn.LeakyReLU(0.2)
        enc = _Bag()
        enc.conv_blocks = nn.ModuleList([_conv_bn(128, 128, 2, False, act()), _conv_bn(256, 256, 2, False, act()),
                                         _conv_bn(512, 512, 2, False, act())])
        enc.snn_blocks = nn.ModuleList([NeuronParams(64, True), NeuronParams(128, True), NeuronParams(256),
                                        NeuronParams(512)])
        enc.multi_scale_first_conv = nn.ModuleList([_conv_bn(6, 64, 2, False, act()) for _ in k_scales])
        enc.scale_fusion = _conv_bn(64 * len(k_scales), 64, 1, False, act())
        enc.multi_scale_conv = _conv_bn(960, emb_dims, 1, False, act())
        enc.snn_fc = NeuronParams(emb_dims)
        enc.temporal_integration = _Bag()
        enc.temporal_integration.weights = nn.Parameter(torch.ones(time_steps_enc))
        self.encoder = enc
        dec = _Bag()
        dec.fc_in = nn.Sequential(nn.Linear(emb_dims, 256), nn.BatchNorm1d(256), nn.GELU())
        blocks = []
        for cin, cout in ((256, 128), (128, 64)):
            rb = _Bag()
            rb.fc = nn.Sequential(nn.Linear(cin, cout), nn.BatchNorm1d(cout), nn.GELU(), nn.Dropout(dropout),
                                  nn.Linear(cout, cout), nn.BatchNorm1d(cout))
            rb.res_proj = nn.Linear(cin, cout)
            blocks.append(rb)
        dec.residual_blocks = nn.ModuleList(blocks)
        att = _Bag()
        att.to_qkv = nn.Linear(64, 192)
        att.to_out = nn.Sequential(nn.Linear(64, 64), nn.Dropout(dropout))
        att.norm = nn.LayerNorm(64)
        dec.attention = att
        dec.fc_hidden = nn.Sequential(nn.Linear(64, 32), nn.BatchNorm1d(32), nn.GELU(), nn.Dropout(dropout))
        dec.fc_distance = nn.Linear(32, 1)
        self.distance_decoder = dec
        super().train(False)

    def _hparams(self):
        return [self.k, self.emb_dims, self.time_steps_enc, self.num_heads, len(self.k_scales)] + list(self.k_scales)

    def _pack(self, sd):
        return packing.pack_fd(sd, len(self.k_scales))

    def forward(self, Xc_rotated, taps=None, knn_force=None):
        """[B,M,3] -> [B];  [B,N,M,3] -> [B,N]  (fd/snn_coder.py:853-871).
        ``knn_force``: int32 [3,B,M,min(k,M)] feature-space neighbour tables (parity-test protocol)."""
        if Xc_rotated.dim() == 4:
            B, N, M, _ = Xc_rotated.shape
            return self.forward(Xc_rotated.reshape(B * N, M, 3), taps, knn_force).view(B, N)
        x = self._as_patches(Xc_rotated)
        handle = self._engine()
        lib = _lib.load()
        dev = x.device
        b, m = x.shape[0], x.shape[1]
        out = torch.empty((b,), dtype=torch.float32, device=dev)
        if b == 0:
            return out
        if knn_force is not None:
            kk = min(self.k, m)
            if tuple(knn_force.shape) != (3, b, m, kk) or knn_force.dtype != torch.int32:
                raise ValueError("knn_force must be int32 [3,%d,%d,%d]" % (b, m, kk))
            knn_force = knn_force.contiguous()
        ws = self._workspace(handle, b, m, dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sapcu_fd_forward(handle, _lib.ptr(x), b, m, _lib.ptr(knn_force), _lib.ptr(out), _lib.ptr(ws),
                                            ws.numel(), self._taps_array(_lib.FD_TAPS, taps), _lib.current_stream()))
        return out

    def gate_violations(self):
        """Times the kernels found the refractory gate open at t >= 1 (must stay 0; sapcu.h)."""
        n = ctypes.c_int(0)
        _lib.check(_lib.load().sapcu_model_gate_violations(self._engine(), ctypes.byref(n)))
        return n.value

    def reset_states(self):
        """Reference: clears the encoder's state manager (fd/snn_coder.py:889-893), whose 'final' entry is
        created zero and never updated — a functional no-op kept for signature compatibility."""
