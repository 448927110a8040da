#!/usr/bin/env python3
"""BASELINE configs[2]: a GiantSteps-Key-shaped synthetic set (604 clips of 15 s), train_model.py's defaults (batch 8,
accumulate_grad_batches 8, Adam lr 3e-4, ExponentialLR 0.96, --genre), 10 epochs on one MI355X through the drop-in classes.
Reports seconds per epoch, the loss per epoch and the validation MIREX score (sanity only: synthetic labels).

    python3 tools/config3_train.py [epochs] [clips]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
import ake_amd

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 604
opt = Namespace(conv_layers=3, n_filters=4, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5, octaves=8, lr=3e-4,
                gamma=0.96, acc_grad=8, reg=0, key_weight=1.0, tonic_weight=1.0, genre_weight=0.1, use_cos=False, no_ckpt=True, local=False,
                only_semitones=False, multi_scale=False)
t0 = time.perf_counter()
train = ake_amd.KeyDataset(True, opt)
train.import_data(ake_amd.SyntheticSineMixLoader(n_clips), shuffle=True)
val = ake_amd.KeyDataset(True, opt)
val.import_data(ake_amd.SyntheticSineMixLoader(96, first=10_000), shuffle=False)
print(f"dataset: {n_clips} + 96 clips, CQT on the GPU, {time.perf_counter() - t0:.1f} s (incl. synthesising the audio on the host)")
torch.manual_seed(0)
net = ake_amd.PitchClassNet(288, 12, 2, 7, opt, batch_size=8, train_set=train, val_set=val).cuda()
trainer = ake_amd.Trainer(max_epochs=1, accumulate_grad_batches=opt.acc_grad)
optim_state = None
for ep in range(epochs):
    torch.cuda.synchronize(); t = time.perf_counter()
    n0 = len(trainer.train_losses)
    # one epoch at a time so that each can be timed; the optimizer / scheduler state carries over through the module
    if ep == 0:
        optimizers, schedulers = net.configure_optimizers()
        net.configure_optimizers = lambda: (optimizers, schedulers)
    trainer.fit(net)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    losses = trainer.train_losses[n0:]
    res = trainer.val_results[-1]
    print(f"epoch {ep}: {dt:6.2f} s  ({len(losses)} batches, {len(losses) * 8 / dt:7.1f} clips/s)  train loss {sum(losses) / len(losses):.4f}  "
          f"val loss {res['val_loss']:.4f}  val mirex {res['val_mirex_score']:.3f}  val acc {res['val_accuracy']:.3f}", flush=True)
