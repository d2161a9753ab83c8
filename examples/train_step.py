#!/usr/bin/env python3
"""End-to-end training step on synthetic VOC-sized input: a PyTorch-ROCm
convolutional backbone (built here from the layer table of the YOLOv1 paper,
Fig. 3: 24 convolutions + 2 fully connected layers, LeakyReLU 0.1, dropout 0.5)
feeding the HIP head.  The call shape is the reference's training loop
(train.py:84-94): ``loss = model.get_loss(image, ...)``; ``loss.backward()``;
``optim.step()``.

    python examples/train_step.py [--batch 8] [--steps 5] [--width 1.0]

``--width`` scales the channel counts (0.25 gives a backbone that fits a quick test).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import YOLOv1Head, synth  # noqa: E402

# (out_channels, kernel, stride) or "M" for a 2x2 max-pool -- YOLOv1 paper, Fig. 3
LAYERS = [(64, 7, 2), "M", (192, 3, 1), "M", (128, 1, 1), (256, 3, 1), (256, 1, 1), (512, 3, 1), "M",
          (256, 1, 1), (512, 3, 1), (256, 1, 1), (512, 3, 1), (256, 1, 1), (512, 3, 1), (256, 1, 1),
          (512, 3, 1), (512, 1, 1), (1024, 3, 1), "M", (512, 1, 1), (1024, 3, 1), (512, 1, 1),
          (1024, 3, 1), (1024, 3, 1), (1024, 3, 2), (1024, 3, 1), (1024, 3, 1)]


def make_backbone(width: float = 1.0, n_cells: int = 7, depth: int = 30, hidden: int = 4096):
    mods, cin = [], 3
    for spec in LAYERS:
        if spec == "M":
            mods.append(nn.MaxPool2d(2, 2))
            continue
        cout, k, s = spec
        cout = max(8, int(cout * width))
        mods += [nn.Conv2d(cin, cout, k, s, k // 2), nn.LeakyReLU(0.1)]
        cin = cout
    hidden = max(64, int(hidden * width))
    mods += [nn.Flatten(), nn.Linear(cin * n_cells * n_cells, hidden), nn.LeakyReLU(0.1),
             nn.Dropout(0.5), nn.Linear(hidden, n_cells * n_cells * depth)]
    return nn.Sequential(*mods)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--width", type=float, default=1.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = make_backbone(args.width).to(dev)
    head = YOLOv1Head(backbone=net)
    optim = torch.optim.AdamW(net.parameters(), lr=1e-4)
    image = torch.randn(args.batch, 3, 448, 448, device=dev)
    _, gx, gc, om = (t.to(dev) for t in synth.make_inputs(args.batch, seed=0))
    for step in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = head.get_loss(image, gx, gc, om)           # train.py:84
        optim.zero_grad()
        loss.backward()                                   # train.py:93
        optim.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c, f, k = head.last_loss_terms[:3].tolist()
        print(f"step {step}: loss {loss.item():.4f}  (coord {c:.1f}, conf {f:.1f}, cls {k:.1f})  {1e3 * dt:.1f} ms")
    dets, count = head.postprocess(head(image).detach(), 0.5, 0.5)
    print("detections kept per image:", count.tolist())


if __name__ == "__main__":
    main()
