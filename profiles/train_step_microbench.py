"""Row f-4 throughput: fn training steps per second on one MI355X at the reference's training shape (config/fn.yaml:
batch 4 clouds x 64 patches x 12 points, k=[24,18,12] clamped to 12, emb 640, time_steps_enc 6, AdamW lr 1.8e-4 wd 1e-4,
grad_clip 0.15 'norm', dropout 0.1) and at a larger batch.  Synthetic patches, parameters = testing.training_state_dict.
usage: python3 profiles/train_step_microbench.py [--steps 20] [--batches 4,32] [--amp] [--epoch 200]
  --amp     Trainer(use_amp=True): the GEMMs of the step on bf16 operands (BASELINE config 5)
  --epoch N one epoch of N batches through fn_trainer.run_epoch (the trainfn.py:253-330 loop) over the synthetic PU1K-shaped
            loader instead of a repeated batch"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sapcu_amd  # noqa: E402
from sapcu_amd import fn_trainer, testing as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batches", default="4,32")
    ap.add_argument("--graph", action="store_true", help="replay the step as a HIP graph (fn_trainer.GraphedTrainStep)")
    ap.add_argument("--amp", action="store_true", help="bf16 GEMM operands (Trainer(use_amp=True))")
    ap.add_argument("--epoch", type=int, default=0, help="run one epoch of this many synthetic batches through run_epoch")
    args = ap.parse_args()
    out = []
    for B in [int(b) for b in args.batches.split(",")]:
        model = sapcu_amd.ImprovedSNNNormalEstimation(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=6, time_steps_dec=9, num_heads=8,
                                                      use_snn_decoder=False, decoder_dropout=0.1)
        model.load_state_dict(T.training_state_dict(model.state_dict(), 3), strict=True)
        model.cuda()
        opt = torch.optim.AdamW(model.parameters(), lr=1.8e-4, weight_decay=1e-4, betas=(0.9, 0.999), capturable=args.graph)
        tr = fn_trainer.Trainer(model, opt, device=torch.device("cuda"), use_amp=args.amp, grad_clip=0.15, grad_clip_type="norm")
        if args.epoch:
            warm = fn_trainer.SyntheticPU1K(args.warmup, batch_size=B, seed=1)
            fn_trainer.run_epoch(tr, warm, lr=1.8e-4, warmup_steps=2000, state_reset_freq=25)
            it, losses, st = fn_trainer.run_epoch(tr, fn_trainer.SyntheticPU1K(args.epoch, batch_size=B, seed=2), it=args.warmup, lr=1.8e-4,
                                                  warmup_steps=2000, state_reset_freq=25)
            out.append({"mode": "epoch", "gemm_operands": "bf16" if args.amp else "f32", "batches": args.epoch, "batch_clouds": B, "patches_per_cloud": 64,
                        "points_per_patch": 12, "seconds": round(st["seconds"], 3), "clouds_per_s": round(st["clouds_per_s"], 1),
                        "ms_per_step": round(1e3 * st["seconds"] / max(len(losses), 1), 3), "skipped": st["skipped"],
                        "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4), "loss_mean": round(float(np.mean(losses)), 4)})
            print(json.dumps(out[-1]), flush=True)
            continue
        rng = np.random.default_rng(0)
        NP, M = 64, 12
        centres = rng.normal(size=(B, NP, 1, 3)) * 0.4
        pts = torch.tensor((centres + rng.normal(size=(B, NP, M, 3)) * np.array([0.08, 0.08, 0.01])).astype(np.float32)).cuda()
        gt = torch.tensor(rng.normal(size=(B, NP, 3)).astype(np.float32)).cuda()
        data = {"input": pts, "normal": gt}
        step = fn_trainer.GraphedTrainStep(tr, data, warmup=args.warmup) if args.graph else tr.train_step
        for _ in range(args.warmup):
            step(data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        for _ in range(args.steps):
            loss, _ = step(data)
            done += loss is not None
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out.append({"batch_clouds": B, "patches": B * NP, "points_per_patch": M, "ms_per_step": round(dt * 1e3, 3),
                    "graph": bool(args.graph), "gemm_operands": "bf16" if args.amp else "f32", "clouds_per_s": round(B / dt, 1), "patches_per_s": round(B * NP / dt, 1), "steps_ok": done, "last_loss": loss})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
