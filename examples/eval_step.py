#!/usr/bin/env python3
"""Evaluation pass on the device: decode + confidence threshold + per-class NMS of a batch
of head tensors, then the 11-point mAP of the kept rows against ground truth -- the call
shape of the reference's evaluation code (eval.py:145 ``get_map(pred, gt, n_classes,
iou_thresh)`` with ``pred`` rows ``l, t, r, b, conf, class`` and ``gt`` rows ``l, t, r, b,
class``, eval.py:89-94), with nothing leaving the GPU in between.  The validation loss of
the same batch (train.py:100-110 calls ``get_loss`` per batch) comes out of the same launch:
``head.head_step`` reads the head tensor once for the loss and for the detections.

    python examples/eval_step.py [--batch 256] [--conf 0.3] [--iou 0.5]

The head tensors are synthetic (a trained backbone is out of scope here); the ground
truth is the synthetic box list the targets were encoded from.  The reference pools the
detections of all images before the metric, and so does this script.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import MODE_REFERENCE, YOLOv1Head, box_ops, synth  # noqa: E402


def evaluate(head: YOLOv1Head, out: torch.Tensor, gt_ltrb, gt_label, gt_count, conf_thresh, iou_thresh,
             map_iou_thresh=0.5):
    """-> (mAP as a 0-d float64 tensor, pooled detections (P, 6), pooled ground truth (G, 5),
    validation loss as a 0-d tensor)."""
    targets = head.encode_targets(gt_ltrb, gt_label, gt_count)       # data.py:173-194 on the device
    terms, _, dets, count = head.head_step(out, *targets, conf_thresh, iou_thresh, MODE_REFERENCE,
                                           rows=head.n_grid)         # one box per cell at most
    rows = torch.arange(dets.shape[1], device=dets.device)[None, :] < count[:, None]
    pred = dets[rows]                                                  # (P, 6), image after image
    m = torch.arange(gt_ltrb.shape[1], device=gt_ltrb.device)[None, :] < gt_count[:, None]
    gt = torch.cat([gt_ltrb[m], gt_label[m][:, None].float()], dim=1)  # (G, 5)
    return box_ops.get_map(pred.contiguous(), gt.contiguous(), head.n_classes, map_iou_thresh), pred, gt, terms[4]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--conf", type=float, default=0.3)
    ap.add_argument("--iou", type=float, default=0.5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    head = YOLOv1Head()
    out = synth.make_pred(args.batch, seed=0).to(dev)
    ltrb, label, cnt = (t.to(dev) for t in synth.make_gt_boxes(args.batch, seed=1))
    evaluate(head, out, ltrb, label, cnt, args.conf, args.iou)          # first use: code loads
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m, pred, gt, loss = evaluate(head, out, ltrb, label, cnt, args.conf, args.iou)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.batch} images: {pred.shape[0]} detections kept, {gt.shape[0]} ground truths, "
          f"mAP@{0.5} = {m.item():.6f}, validation loss {loss.item():.4f}  "
          f"({1e3 * dt:.2f} ms for target encoding + loss + decode + NMS + pooling + mAP)")


if __name__ == "__main__":
    main()
