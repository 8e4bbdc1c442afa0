"""Training driver for fn — the reference's ``fn/trainer.py`` ``Trainer`` (:9-148 train_step, :150-227 evaluate, :229-247
eval_step) over the HIP training ops of sapcu_amd/train.py.  Same constructor arguments, same return values, same
skip-the-batch behaviour on NaN/Inf.  The optimiser is whatever torch optimiser the caller built on model.parameters()
(the reference's trainfn.py:107-109 builds AdamW/Adam).  ``use_amp=True`` runs the Conv / Linear GEMMs of the step — forward,
data gradient, weight gradient — on bf16 operands with f32 accumulation (sapcu_amd.train.gemm_precision, csrc/train_bf16.hip):
the counterpart of the reference's ``torch.amp.autocast`` region (fn/trainer.py:67-83; BASELINE config 5 names bf16, whose f32
exponent range needs no loss scaling — a GradScaler, if given, is still driven through scale / unscale_ / step / update so a
reference training script runs unchanged).  ``run_epoch`` is the batch loop of trainfn.py:253-330; ``SyntheticPU1K`` stands in
for the PU1K mesh sampler (fn/datacore.py needs trimesh and the meshes, absent here)."""
import time

import numpy as np
import torch
from torch.nn import functional as F


class Trainer:
    def __init__(self, model, optimizer, device=None, input_type='pointcloud', vis_dir=None, threshold=0.5, eval_sample=False,
                 gradient_accumulation=1, use_amp=False, scaler=None, grad_clip=None, grad_clip_type='norm'):
        self.model, self.optimizer, self.device = model, optimizer, device
        self.input_type, self.vis_dir, self.threshold, self.eval_sample = input_type, vis_dir, threshold, eval_sample
        self.gradient_accumulation = gradient_accumulation
        self.use_amp, self.scaler = use_amp, scaler
        self.grad_clip, self.grad_clip_type = grad_clip, grad_clip_type
        self.accumulation_step = 0
        if device is not None:
            self.model.to(device)

    def _batch(self, data):
        dd = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in data.items()}
        return dd['input'].float(), dd['normal'].float()

    @staticmethod
    def _finite(t):
        return bool(torch.isfinite(t).all())

    def _abort(self):
        self.optimizer.zero_grad()
        self.accumulation_step = 0
        return None, None

    def train_step(self, data):
        """One optimisation step (fn/trainer.py:41-148): -> (loss value, loss dict), or (None, None) for a skipped batch."""
        self.model.train()
        self.accumulation_step += 1
        points, gt = self._batch(data)
        if not self._finite(points) or not self._finite(gt):
            print("WARNING: NaN/Inf detected in the batch at step %d" % self.accumulation_step)
            return None, None
        from . import train as T
        gt = F.normalize(gt, dim=-1)
        amp = self.use_amp and self.scaler is not None
        with T.gemm_precision("bf16" if self.use_amp else "f32"):      # forward and backward, like the autocast region + backward
            pred = self.model(points)
            if not self._finite(pred):
                print("WARNING: NaN/Inf in model predictions")
                return None, None
            pred = F.normalize(pred, dim=-1)
            loss, loss_dict = self.model.compute_loss(pred, gt, points)
            if not self._finite(loss):
                print("WARNING: NaN/Inf in loss value")
                return None, None
            scaled = loss / self.gradient_accumulation
            (self.scaler.scale(scaled) if amp else scaled).backward()
        if self.accumulation_step % self.gradient_accumulation == 0:
            total = None
            if self.grad_clip is not None:
                if amp:
                    self.scaler.unscale_(self.optimizer)
                if self.grad_clip_type == 'norm':
                    total = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip)
                else:
                    torch.nn.utils.clip_grad_value_(self.model.parameters(), self.grad_clip)
            if total is None:
                grads = [prm.grad for prm in self.model.parameters() if prm.grad is not None]
                total = torch.stack(torch._foreach_norm(grads)).sum() if grads else torch.zeros(())
            # the reference tests every gradient tensor for NaN/Inf (one host sync each); the global norm is finite exactly
            # when all of them are, so one sync decides
            if not self._finite(total):
                print("WARNING: NaN/Inf in gradients")
                return self._abort()
            if amp:
                self.scaler.step(self.optimizer)
                self.scaler.update()
            else:
                self.optimizer.step()
            self.optimizer.zero_grad()
            self.accumulation_step = 0
        return loss.item(), loss_dict

    @staticmethod
    def compute_angular_error(pred, gt):
        cos = torch.clamp(F.cosine_similarity(pred, gt, dim=-1), -1 + 1e-6, 1 - 1e-6)
        return torch.rad2deg(torch.acos(cos)).mean()

    cos_sim = compute_angular_error

    def eval_step(self, data):
        self.model.eval()
        points, gt = self._batch(data)
        with torch.no_grad():
            pred = self.model(points)
            loss, loss_dict = self.model.compute_loss(pred, gt, points)
        return loss.item(), (loss_dict or {}).get('confidence', 0.0)

    def evaluate(self, val_loader):
        """fn/trainer.py:150-227: -> (mean loss, mean confidence, {'confidence', 'angular_error_deg'})."""
        self.model.eval()
        total_loss = total_conf = total_err = 0.0
        batches, confs, errs = 0, [], []
        with torch.no_grad():
            for data in val_loader:
                points, gt = self._batch(data)
                if not self._finite(points):
                    print("WARNING: NaN/Inf in validation input, skipping batch")
                    continue
                gt = F.normalize(gt, dim=-1)
                pred = F.normalize(self.model(points), dim=-1)
                loss, loss_dict = self.model.compute_loss(pred, gt, points)
                if not self._finite(loss):
                    print("WARNING: NaN/Inf in validation loss, skipping batch")
                    continue
                err = self.compute_angular_error(pred.reshape(-1, 3), gt.reshape(-1, 3)).item()
                total_loss += loss.item()
                total_err += err
                total_conf += loss_dict['confidence']
                confs.append(loss_dict['confidence'])
                errs.append(err)
                batches += 1
        n = max(batches, 1)
        metrics = {}
        if confs:
            metrics['confidence'] = float(np.mean(confs))
        metrics['angular_error_deg'] = total_err / n
        return total_loss / n, total_conf / n, metrics


class SyntheticPU1K(object):
    """Stand-in for the reference's PU1K training loader (fn/datacore.py; config/fn.yaml: batch_size 4, 64 patches of 12 points
    per cloud): batches {'input': [B, N, M, 3] f32 centred patches, 'normal': [B, N, 3] unit normals} sampled from analytic
    surfaces whose normals are known (sphere and torus shells, random pose), deterministic in (seed, batch index)."""

    def __init__(self, batches, batch_size=4, patches=64, points=12, seed=0, radius=0.06):
        self.batches, self.batch_size, self.patches, self.points, self.seed, self.radius = batches, batch_size, patches, points, seed, radius

    def __len__(self):
        return self.batches

    def _cloud(self, rng):
        n, m = self.patches, self.points
        c = rng.normal(size=(n, 3))
        c /= np.linalg.norm(c, axis=1, keepdims=True)                    # patch centres on a unit sphere ...
        nrm = c.copy()
        if rng.random() < 0.5:                                           # ... or on a torus (R = 0.7, r = 0.3): normals differ from positions
            u, v = rng.uniform(0, 2 * np.pi, (2, n))
            c = np.stack([(0.7 + 0.3 * np.cos(v)) * np.cos(u), (0.7 + 0.3 * np.cos(v)) * np.sin(u), 0.3 * np.sin(v)], 1)
            nrm = np.stack([np.cos(v) * np.cos(u), np.cos(v) * np.sin(u), np.sin(v)], 1)
        # tangent-plane samples around each centre, pushed back along the normal by the local curvature (1 / 2 x^2 term)
        t1 = np.cross(nrm, rng.normal(size=(n, 3)))
        t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
        t2 = np.cross(nrm, t1)
        ab = rng.uniform(-self.radius, self.radius, (n, m, 2))
        pts = ab[..., :1] * t1[:, None, :] + ab[..., 1:] * t2[:, None, :] - 0.5 * (ab ** 2).sum(-1, keepdims=True) * nrm[:, None, :]
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))                     # random pose of the whole cloud
        return (pts @ q.T).astype(np.float32), (nrm @ q.T).astype(np.float32)

    def __iter__(self):
        for b in range(self.batches):
            rng = np.random.default_rng([self.seed, b])
            clouds = [self._cloud(rng) for _ in range(self.batch_size)]
            yield {"input": torch.from_numpy(np.stack([c[0] for c in clouds])), "normal": torch.from_numpy(np.stack([c[1] for c in clouds]))}


def clamp_neuron_parameters(model):
    """The post-step clamp of the neuron parameters the reference's fd training loop applies (trainfd.py:305-313); trainfn.py has
    none — the forward clamps the values it uses either way (fn/snn_coder.py:87-110), this keeps the stored parameters in range."""
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if 'membrane_decay' in name:
                prm.data.clamp_(0.1, 0.99)
            elif 'threshold_adapt' in name:
                prm.data.clamp_(0.001, 0.1)
            elif 'refractory_decay' in name:
                prm.data.clamp_(0.1, 0.95)


def run_epoch(trainer, train_loader, it=0, epoch_it=0, lr=None, warmup_steps=0, warmup_factor=0.01, state_reset_freq=0, print_every=0,
              log=print, clamp_parameters=False):
    """One pass over ``train_loader`` — the batch loop of the reference's trainfn.py:253-330 without its logging / checkpoint
    side effects: iteration counter, SNN state reset every ``state_reset_freq`` iterations, linear learning-rate warm-up, skipped
    invalid or non-finite batches, per-iteration losses, samples per second; optionally the neuron-parameter clamp of the fd loop.
    -> (it, losses, stats)."""
    model, optimizer = trainer.model, trainer.optimizer
    lr = optimizer.param_groups[0]["lr"] if lr is None else lr
    losses, skipped, start, seen = [], 0, time.time(), 0
    for batch in train_loader:
        it += 1
        if "input" not in batch:                                         # trainfn.py:259-260
            continue
        if state_reset_freq > 0 and it % state_reset_freq == 0 and hasattr(model, "reset_states"):
            model.reset_states()
        if warmup_steps > 0 and it < warmup_steps:                       # trainfn.py:266-269
            f = warmup_factor + (1 - warmup_factor) * (it / warmup_steps)
            for g in optimizer.param_groups:
                g["lr"] = lr * f
        loss, _ = trainer.train_step(batch)
        if clamp_parameters:                                             # (trainfd.py:305-313; off = trainfn.py's behaviour)
            clamp_neuron_parameters(model)
        if loss is None:
            skipped += 1
            continue
        losses.append(float(loss))
        seen += int(batch["input"].shape[0])
        if print_every > 0 and it % print_every == 0:
            log("[Epoch %03d] it=%06d, loss=%.6f, avg_loss=%.6f, lr=%.6f, samples/s=%.1f" % (
                epoch_it, it, losses[-1], float(np.mean(losses[-print_every:])), optimizer.param_groups[0]["lr"],
                seen / max(time.time() - start, 1e-9)))
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.time() - start
    return it, losses, {"clouds": seen, "seconds": dt, "clouds_per_s": seen / max(dt, 1e-9), "skipped": skipped}


class GraphedTrainStep:
    """One optimisation step of a FIXED batch shape captured as a HIP graph and replayed (ours; the reference has no
    counterpart): forward, loss, backward, global-norm clipping and optimizer.step() are ~1300 kernel launches that leave
    the GPU idle between them at the reference's batch size; replaying them as one graph removes the host from the loop
    (17.9 ms against 25.2 ms per step at the reference's batch shape).

    Differences from ``Trainer.train_step``: the optimiser must be built with ``capturable=True``; a batch with non-finite
    gradients cannot be skipped from the host (the update is part of the graph): its gradients are zeroed on the device and
    ``__call__`` returns (None, None) like a skipped batch; gradient accumulation is not supported; the `warmup` steps before the capture are real optimisation steps on `example`.  Dropout masks still
    change from replay to replay (torch's graph-safe generator)."""

    def __init__(self, trainer, example, warmup=3):
        from . import train as T
        self.trainer = tr = trainer
        if tr.gradient_accumulation != 1 or (tr.use_amp and tr.scaler is not None):
            raise ValueError("GraphedTrainStep: gradient accumulation / GradScaler are not supported")
        if not all(g.get("capturable", False) for g in tr.optimizer.param_groups):
            raise ValueError("GraphedTrainStep: build the optimiser with capturable=True")
        pts, gt = tr._batch(example)
        self.pts, self.gt = pts.clone(), gt.clone()
        model, opt = tr.model, tr.optimizer
        model.train()

        precision = "bf16" if tr.use_amp else "f32"          # the same GEMM arithmetic as Trainer.train_step (warm-up AND capture)
        self.gemm_precision = precision

        def core():
            opt.zero_grad(set_to_none=True)
            gtn = F.normalize(self.gt, dim=-1)
            with T.gemm_precision(precision):
                pred = F.normalize(model(self.pts), dim=-1)
                xyz = self.pts.mean(dim=2) if self.pts.ndim == 4 else self.pts
                loss, conf = T.angular_loss_with_consistency(pred, gtn, xyz)
                loss.backward()
            if tr.grad_clip is not None and tr.grad_clip_type == 'norm':
                total = torch.nn.utils.clip_grad_norm_(model.parameters(), tr.grad_clip, foreach=True)
            else:
                if tr.grad_clip is not None:
                    torch.nn.utils.clip_grad_value_(model.parameters(), tr.grad_clip, foreach=True)
                total = torch.stack(torch._foreach_norm([q.grad for q in model.parameters() if q.grad is not None])).sum()
            # a batch with non-finite gradients cannot be skipped from the host inside a graph: its gradients are zeroed instead
            # (the parameters stay finite; the optimiser still decays its moments and applies weight decay for that step)
            bad = ~torch.isfinite(total)
            for q in model.parameters():
                if q.grad is not None:
                    q.grad.masked_fill_(bad, 0.0)
            opt.step()
            return loss.detach(), conf.detach(), total.detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                core()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.conf, self.total = core()

    def __call__(self, data):
        pts, gt = self.trainer._batch(data)
        self.pts.copy_(pts)
        self.gt.copy_(gt)
        self.graph.replay()
        # a replay changes parameters and BatchNorm statistics without running any Python in-place op, so the tensors' version
        # counters (what _HipModel._state_key watches) do not move: drop the packed inference blob explicitly
        self.trainer.model._packed_key = None
        vals = torch.stack([self.loss, self.conf, self.total.to(self.loss.dtype)]).tolist()      # one sync
        if not all(np.isfinite(vals)):
            print("WARNING: NaN/Inf in loss or gradients (batch applied with zero gradients): loss %r, gradient norm %r" % (vals[0], vals[2]))
            return None, None
        return vals[0], {"total_loss": vals[0], "confidence": vals[1]}
