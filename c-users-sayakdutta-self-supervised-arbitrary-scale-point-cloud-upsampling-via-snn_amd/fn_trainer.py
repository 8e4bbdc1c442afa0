"""Training driver for fn — the reference's ``fn/trainer.py`` ``Trainer`` (:9-148 train_step, :150-227 evaluate, :229-247
eval_step) over the HIP training ops of sapcu_amd/train.py.  Same constructor arguments, same return values, same
skip-the-batch behaviour on NaN/Inf.  The optimiser is whatever torch optimiser the caller built on model.parameters()
(the reference's trainfn.py:107-109 builds AdamW/Adam); the HIP ops compute in f32 whatever ``use_amp`` says — the flag
and the scaler are accepted and the scaler's scale/unscale/step protocol is honoured so a reference training script runs
unchanged."""
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
        gt = F.normalize(gt, dim=-1)
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
        amp = self.use_amp and self.scaler is not None
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
