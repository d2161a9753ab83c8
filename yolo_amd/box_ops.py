"""Pairwise box geometry on the GPU with the names and shapes of the reference's
eval.py: get_area (eval.py:22-32), get_intersection_area (:35-50), get_iou
(:53-65), get_giou (:68-83).  fp32 results are bit-identical to the reference's
PyTorch expressions.  get_ap / get_map (eval.py:129-149) return float64 device
tensors that equal the reference's numpy results."""
from __future__ import annotations

import torch

from . import _lib
from .head import _need, _stream


def _boxes(t: torch.Tensor, name: str) -> torch.Tensor:
    _need(t, name, torch.float32)
    if t.dim() != 2 or t.shape[1] != 4:
        raise ValueError(f"{name}: expected shape (N, 4), got {tuple(t.shape)}")
    return t


def _pairwise(ltrb1, ltrb2, kind):
    _boxes(ltrb1, "ltrb1")
    _boxes(ltrb2, "ltrb2")
    if ltrb1.device != ltrb2.device:
        raise ValueError("ltrb1 and ltrb2 live on different devices")
    n, m = ltrb1.shape[0], ltrb2.shape[0]
    out = torch.empty(n, m, dtype=torch.float32, device=ltrb1.device)
    with torch.cuda.device(ltrb1.device):
        _lib.check(_lib.load().yh_pairwise_boxes(ltrb1.data_ptr(), ltrb2.data_ptr(), out.data_ptr(),
                                                 n, m, kind, _stream(ltrb1.device)))
    return out


def get_area(ltrb):
    """eval.py:22-32: (N, 4) -> (N)."""
    _boxes(ltrb, "ltrb")
    out = torch.empty(ltrb.shape[0], dtype=torch.float32, device=ltrb.device)
    with torch.cuda.device(ltrb.device):
        _lib.check(_lib.load().yh_box_area(ltrb.data_ptr(), out.data_ptr(), ltrb.shape[0],
                                           _stream(ltrb.device)))
    return out


def get_intersection_area(ltrb1, ltrb2):
    """eval.py:35-50: (N, 4), (M, 4) -> (N, M)."""
    return _pairwise(ltrb1, ltrb2, _lib.YH_BOX_INTERSECTION)


def get_iou(ltrb1, ltrb2):
    """eval.py:53-65: (N, 4), (M, 4) -> (N, M); 0/0 -> 0."""
    return _pairwise(ltrb1, ltrb2, _lib.YH_BOX_IOU)


def get_smallest_enclosing_area(bbox1, bbox2):
    """eval.py:68-73: (N, 4), (M, 4) -> (N, M), the area of the smallest box around each pair."""
    return _pairwise(bbox1, bbox2, _lib.YH_BOX_ENCLOSING)


def get_giou(bbox1, bbox2):
    """eval.py:76-83: (N, 4), (M, 4) -> (N, M); -1 where the enclosing box is empty."""
    return _pairwise(bbox1, bbox2, _lib.YH_BOX_GIOU)


def _map(pred, gt, n_classes, iou_thresh):
    _need(pred, "pred", torch.float32)
    _need(gt, "gt", torch.float32)
    if pred.dim() != 2 or pred.shape[1] != 6:
        raise ValueError(f"pred: expected shape (P, 6), got {tuple(pred.shape)}")      # eval.py:94
    if gt.dim() != 2 or gt.shape[1] != 5:
        raise ValueError(f"gt: expected shape (G, 5), got {tuple(gt.shape)}")          # eval.py:93
    dev = pred.device
    lib = _lib.load()
    p, g = pred.shape[0], gt.shape[0]
    ws = torch.empty(max(256, lib.yh_map_workspace_bytes(p, g)), dtype=torch.uint8, device=dev)
    ap = torch.empty(int(n_classes), dtype=torch.float64, device=dev)
    m = torch.empty((), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.yh_map_11point(pred.data_ptr(), gt.data_ptr(), p, g, int(n_classes),
                                      float(iou_thresh), ap.data_ptr(), m.data_ptr(),
                                      ws.data_ptr(), ws.numel(), _stream(dev)))
    return ap, m


def get_ap(pred, gt, cls_idx, iou_thresh, n_classes=None):
    """eval.py:129-142 for one class (0-d float64 tensor)."""
    ap, _ = _map(pred, gt, (cls_idx + 1) if n_classes is None else n_classes, iou_thresh)
    return ap[cls_idx]


def get_map(pred, gt, n_classes, iou_thresh=0.5):
    """eval.py:145-149: pred (P, 6) rows l,t,r,b,conf,class; gt (G, 5) rows l,t,r,b,class."""
    return _map(pred, gt, n_classes, iou_thresh)[1]
