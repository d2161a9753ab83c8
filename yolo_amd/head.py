"""YOLOv1 detection head on MI355X: the reference's head methods
(model.py:150-317, 379-394) backed by the HIP kernels of libyolohead.so.

``YOLOv1Head`` keeps the reference's names, argument meaning and constructor
keywords (model.py:96-104), so a caller written against ``YOLOv1`` keeps
working:

    head = YOLOv1Head()                       # or YOLOv1Head(backbone=net)
    loss = head.get_loss(out, gt_norm_xywh, gt_cls_prob, obj_mask)
    loss.backward()                           # or scaler.scale(loss).backward()
    dets, count = head.postprocess(out, conf_thresh=0.5, iou_thresh=0.5)

Tensors stay on the GPU; PyTorch supplies memory and the current stream only.
Every op raises if the HIP library is missing -- there is no fallback path.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _on_device:
    """``torch.cuda.device(dev)`` only when ``dev`` is not already current (the
    context manager costs several microseconds per call otherwise)."""

    __slots__ = ("ctx",)

    def __init__(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.ctx = None if idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _need(t: torch.Tensor, name: str, dtype, shape=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU (got device {t.device}); "
                         "yolo_amd has no CPU path")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


class LossLog:
    """Device-side ring of the per-step loss sums (coord, conf, class, total).

    ``yh_loss_finalize`` appends one row per step and advances a cursor that lives
    on the device, so a captured HIP graph logs every replay without host work.
    Nothing on the training critical path reads the ring; a data-parallel job
    reduces it across ranks every ``capacity`` steps at most (yolo_amd.dist)."""

    def __init__(self, device, capacity: int = 64):
        self.capacity = int(capacity)
        self.ring = torch.zeros(self.capacity, 4, dtype=torch.float32, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=device)

    def reset(self):
        self.ring.zero_()
        self.cursor.zero_()


class _LossFunction(torch.autograd.Function):
    """Forward launches the fused forward+backward kernel and keeps the
    gradient; backward hands it out, multiplied on the device by the incoming
    grad_output (1 for ``loss.backward()``, the GradScaler scale under
    ``scaler.scale(loss).backward()``, train.py:88-93)."""

    @staticmethod
    def forward(ctx, out, gt_norm_xywh, gt_cls_prob, obj_mask, head, inv_global_batch):
        terms, grad = head._launch_loss(out, gt_norm_xywh, gt_cls_prob, obj_mask, inv_global_batch)
        ctx.grad_pred = grad
        head.last_loss_terms = terms
        return terms[4]

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        grad = ctx.grad_pred
        ctx.grad_pred = None
        if grad is None:
            raise RuntimeError("YOLOv1Head loss: backward called twice; the fused kernel keeps "
                               "one gradient buffer per forward")
        go = grad_output
        if go.dtype != torch.float32 or go.numel() != 1 or go.device != grad.device:
            go = go.to(device=grad.device, dtype=torch.float32).reshape(())
        with _on_device(grad.device):
            _lib.check(_lib.load().yh_scale_grad(grad.data_ptr(), go.data_ptr(), grad.numel(),
                                                 _stream(grad.device)))
        return grad, None, None, None, None, None


class _LinearLossFunction(torch.autograd.Function):
    """The loss on (z, bias) of the last nn.Linear: the forward launch produces the
    gradient for z, the bias gradient is its fixed-order sum over the batch."""

    @staticmethod
    def forward(ctx, z, bias, gt_norm_xywh, gt_cls_prob, obj_mask, head, global_batch):
        terms, grad, grad_bias = head.loss_fwd_bwd_linear(
            z, bias, gt_norm_xywh, gt_cls_prob, obj_mask, global_batch,
            want_bias_grad=bias.requires_grad)
        ctx.grads = (grad, grad_bias)
        ctx.bias_shape = bias.shape
        head.last_loss_terms = terms
        return terms[4]

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        grad, grad_bias = ctx.grads
        ctx.grads = None
        if grad is None:
            raise RuntimeError("YOLOv1Head loss: backward called twice")
        go = grad_output
        if go.dtype != torch.float32 or go.numel() != 1 or go.device != grad.device:
            go = go.to(device=grad.device, dtype=torch.float32).reshape(())
        lib = _lib.load()
        with _on_device(grad.device):
            _lib.check(lib.yh_scale_grad(grad.data_ptr(), go.data_ptr(), grad.numel(),
                                         _stream(grad.device)))
            if grad_bias is not None:
                _lib.check(lib.yh_scale_grad(grad_bias.data_ptr(), go.data_ptr(), grad_bias.numel(),
                                             _stream(grad.device)))
                grad_bias = grad_bias.view(ctx.bias_shape)
        return grad, grad_bias, None, None, None, None, None


class YOLOv1Head(nn.Module):
    def __init__(
        self,
        img_size: int = 448,
        n_cells: int = 7,
        n_bboxes: int = 2,
        n_classes: int = 20,
        coord_coeff: float = 5,
        noobj_coeff: float = 0.5,
        backbone: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
    ):
        super().__init__()
        self.img_size = int(img_size)
        self.n_cells = int(n_cells)
        self.n_bboxes = int(n_bboxes)
        self.n_classes = int(n_classes)
        self.coord_coeff = float(coord_coeff)
        self.noobj_coeff = float(noobj_coeff)
        self.cell_size = self.img_size // self.n_cells           # model.py:114
        self.backbone = backbone
        self.last_loss_terms: Optional[torch.Tensor] = None
        self._workspace = {}
        self._fin_event = {}     # device -> event after the last side-stream finalize on its workspace
        self._pipe = {}          # device -> [workspace, phase of the last pipelined call]
        _lib.load()  # fail at construction time if the HIP library is missing

    # ------------------------------------------------------------------ helpers
    @property
    def depth(self) -> int:
        return 5 * self.n_bboxes + self.n_classes

    @property
    def n_grid(self) -> int:
        return self.n_cells * self.n_cells

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """The head tensor (B, 5*Bx+C, S, S) of model.py:139-148: ``backbone(x)``
        reshaped, or ``x`` itself when no backbone is attached."""
        out = self.backbone(x) if self.backbone is not None else x
        return out.view(-1, self.depth, self.n_cells, self.n_cells)

    def _check_out(self, out: torch.Tensor, allow_channels_last: bool = False) -> torch.Tensor:
        if not isinstance(out, torch.Tensor):
            raise TypeError(f"out: expected a torch.Tensor, got {type(out).__name__}")
        if out.dim() != 4 or tuple(out.shape[1:]) != (self.depth, self.n_cells, self.n_cells):
            raise ValueError(f"out: expected shape (B, {self.depth}, {self.n_cells}, "
                             f"{self.n_cells}), got {tuple(out.shape)}")
        if allow_channels_last and self._cell_major(out):
            if not out.is_cuda:
                raise ValueError(f"out: must live on the GPU (got device {out.device}); yolo_amd has no CPU path")
            if out.dtype != torch.float32:
                raise ValueError(f"out: expected dtype torch.float32, got {out.dtype}")
            return out
        return _need(out, "out", torch.float32)

    @staticmethod
    def _cell_major(out: torch.Tensor) -> bool:
        """A (B, D, S, S) tensor in channels-last memory format: its storage is
        (B, S, S, D), a cell's D values consecutive -- what a backbone that ends in a
        channels-last convolution emits.  The loss and postprocess read it in place
        (yh_loss_fwd_bwd_nhwc, yh_decode_nms_nhwc); the gradient comes back in the same format."""
        return (not out.is_contiguous()) and out.is_contiguous(memory_format=torch.channels_last)

    def _check_bias(self, bias: torch.Tensor, device) -> torch.Tensor:
        bias = _need(bias.detach(), "bias", torch.float32)
        if bias.numel() != self.depth * self.n_grid:
            raise ValueError(f"bias: expected {self.depth * self.n_grid} values, got {bias.numel()}")
        if bias.device != device:
            raise ValueError(f"bias: on {bias.device}, the head tensor is on {device}")
        return bias

    def _get_workspace(self, device, nbytes: int) -> torch.Tensor:
        # One scratch buffer per device, reused by every loss call of this head: calls
        # are ordered by the stream they are enqueued on, so a head serves one stream
        # per device at a time (use one head per stream otherwise).
        ws = self._workspace.get(device)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(nbytes, 4096), dtype=torch.uint8, device=device)
            self._workspace[device] = ws
        return ws

    # --------------------------------------------------------------------- loss
    def _launch_loss(self, out, gt_norm_xywh, gt_cls_prob, obj_mask, inv_global_batch, log=None,
                     finalize_stream=None):
        out = self._check_out(out, allow_channels_last=True)
        nhwc = self._cell_major(out)
        if nhwc and (log is not None or finalize_stream is not None):
            raise ValueError("out: a channels-last head tensor goes through the plain loss call only "
                             "(no device-side log, no side-stream finalize)")
        bsz, n = out.shape[0], self.n_grid
        _need(gt_norm_xywh, "gt_norm_xywh", torch.float32, (bsz, n, 1, 4))
        _need(gt_cls_prob, "gt_cls_prob", torch.float32, (bsz, n, 1, self.n_classes))
        if obj_mask.dtype == torch.bool:
            obj_mask = obj_mask.view(torch.uint8)
        _need(obj_mask, "obj_mask", torch.uint8, (bsz, n, 1, 1))
        for name, t in (("gt_norm_xywh", gt_norm_xywh), ("gt_cls_prob", gt_cls_prob),
                        ("obj_mask", obj_mask)):
            if t.device != out.device:
                raise ValueError(f"{name}: on {t.device}, out is on {out.device}")
        lib = _lib.load()
        nbytes = lib.yh_workspace_bytes(_lib.YH_OP_LOSS, bsz, self.n_cells, self.n_bboxes,
                                        self.n_classes)
        ws = self._get_workspace(out.device, nbytes)
        grad = torch.empty_like(out)
        terms = torch.empty(5, dtype=torch.float32, device=out.device)
        if log is not None and log.ring.device != out.device:
            raise ValueError(f"loss log on {log.ring.device}, out is on {out.device}")
        with _on_device(out.device):
            st = _stream(out.device)
            pending = self._fin_event.pop(out.device, None)
            if pending is not None:
                # a finalize on a side stream may still be reading the shared workspace
                torch.cuda.current_stream(out.device).wait_event(pending)
            if log is None and finalize_stream is None:
                fn = lib.yh_loss_fwd_bwd_nhwc if nhwc else lib.yh_loss_fwd_bwd
                _lib.check(fn(
                    out.data_ptr(), gt_norm_xywh.data_ptr(), gt_cls_prob.data_ptr(),
                    obj_mask.data_ptr(), grad.data_ptr(), terms.data_ptr(), ws.data_ptr(),
                    ws.numel(), bsz, self.n_cells, self.n_bboxes, self.n_classes, self.img_size,
                    self.coord_coeff, self.noobj_coeff, float(inv_global_batch), st))
                return terms, grad
            _lib.check(lib.yh_loss_partials(
                out.data_ptr(), gt_norm_xywh.data_ptr(), gt_cls_prob.data_ptr(),
                obj_mask.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(), bsz,
                self.n_cells, self.n_bboxes, self.n_classes, self.img_size, self.coord_coeff,
                self.noobj_coeff, float(inv_global_batch), st))
            fst = st
            if finalize_stream is not None:
                # the scalar reduction leaves the critical path: it runs on the side
                # stream as soon as the partial rows exist
                finalize_stream.wait_stream(torch.cuda.current_stream(out.device))
                fst = finalize_stream.cuda_stream
            _lib.check(lib.yh_loss_finalize(
                ws.data_ptr(), ws.numel(), bsz, self.n_cells, float(inv_global_batch),
                terms.data_ptr(), None if log is None else log.ring.data_ptr(),
                0 if log is None else log.capacity,
                None if log is None else log.cursor.data_ptr(), fst))
            if finalize_stream is not None:
                # `terms` and the workspace were allocated on the current stream but are
                # written / read on the side stream: tell the allocator, and make the next
                # loss call on this device wait before it overwrites the workspace
                terms.record_stream(finalize_stream)
                ws.record_stream(finalize_stream)
                ev = torch.cuda.Event()
                ev.record(finalize_stream)
                self._fin_event[out.device] = ev
        return terms, grad

    def loss_step_pipelined(self, out, gt_norm_xywh, gt_cls_prob, obj_mask, log: "LossLog",
                            global_batch=None, terms_prev: Optional[torch.Tensor] = None):
        """One launch per step for loops that only log the loss: writes and returns
        grad = d(total/global_batch)/d(out); the step's four sums reach ``log`` during
        the NEXT call (its first workgroup adds them), or at ``flush_loss_log``.
        ``terms_prev`` (5 floats) receives the previous step's terms.

        The calls alternate between the two halves of a workspace the head keeps
        per device; capture an even number of them when building a HIP graph.  The
        head counts the phases of the calls it sees: replays of a captured graph are
        not among them, so call ``flush_loss_log`` between replays and eager calls."""
        out = self._check_out(out.detach())
        bsz, n = out.shape[0], self.n_grid
        _need(gt_norm_xywh, "gt_norm_xywh", torch.float32, (bsz, n, 1, 4))
        _need(gt_cls_prob, "gt_cls_prob", torch.float32, (bsz, n, 1, self.n_classes))
        if obj_mask.dtype == torch.bool:
            obj_mask = obj_mask.view(torch.uint8)
        _need(obj_mask, "obj_mask", torch.uint8, (bsz, n, 1, 1))
        if terms_prev is not None:
            _need(terms_prev, "terms_prev", torch.float32, (5,))
        for name, t in (("gt_norm_xywh", gt_norm_xywh), ("gt_cls_prob", gt_cls_prob),
                        ("obj_mask", obj_mask), ("log", log.ring)):
            if t.device != out.device:
                raise ValueError(f"{name}: on {t.device}, out is on {out.device}")
        lib = _lib.load()
        gb = bsz if global_batch is None else int(global_batch)
        shape = (bsz, self.n_cells, self.n_bboxes, self.n_classes)
        nbytes = lib.yh_workspace_bytes(_lib.YH_OP_LOSS_PIPELINED, *shape)
        state = self._pipe.get(out.device)
        if state is None or state[0].numel() < nbytes:
            if state is not None:
                self.flush_loss_log(log)         # the halves move: log what is pending first
            # zero-filled: the headers of the two halves say "nothing pending"
            state = [torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=out.device), 1]
            self._pipe[out.device] = state
        ws, phase = state[0], 1 - state[1]
        grad = torch.empty_like(out)
        with _on_device(out.device):
            _lib.check(lib.yh_loss_fwd_bwd_pipelined(
                out.data_ptr(), gt_norm_xywh.data_ptr(), gt_cls_prob.data_ptr(),
                obj_mask.data_ptr(), grad.data_ptr(), _ptr(terms_prev), log.ring.data_ptr(),
                log.capacity, log.cursor.data_ptr(), ws.data_ptr(), ws.numel(), phase, *shape,
                self.img_size, self.coord_coeff, self.noobj_coeff, 1.0 / max(gb, 1),
                _stream(out.device)))
        state[1] = phase
        return grad

    def _check_step_args(self, out, gt_norm_xywh, gt_cls_prob, obj_mask, mode, rows, limit=True):
        out = self._check_out(out.detach())
        bsz, n = out.shape[0], self.n_grid
        _need(gt_norm_xywh, "gt_norm_xywh", torch.float32, (bsz, n, 1, 4))
        _need(gt_cls_prob, "gt_cls_prob", torch.float32, (bsz, n, 1, self.n_classes))
        if obj_mask.dtype == torch.bool:
            obj_mask = obj_mask.view(torch.uint8)
        _need(obj_mask, "obj_mask", torch.uint8, (bsz, n, 1, 1))
        for name, t in (("gt_norm_xywh", gt_norm_xywh), ("gt_cls_prob", gt_cls_prob),
                        ("obj_mask", obj_mask)):
            if t.device != out.device:
                raise ValueError(f"{name}: on {t.device}, out is on {out.device}")
        nn_ = n * self.n_bboxes
        if limit and nn_ > _lib.YH_MAX_CANDIDATES:
            raise ValueError(f"head_step: up to {_lib.YH_MAX_CANDIDATES} candidates per image")
        nrows = nn_ if rows is None else int(rows)
        return out, obj_mask, bsz, nrows

    def head_step(self, out, gt_norm_xywh, gt_cls_prob, obj_mask, conf_thresh=0.5, iou_thresh=0.5,
                  mode=_lib.YH_MODE_REFERENCE, global_batch=None, rows: Optional[int] = None,
                  return_idx=False):
        """Loss forward + backward AND decode + NMS of the same head tensor in one call (an
        evaluation wants the validation loss of train.py:100-110 and the detections for
        eval.get_map, eval.py:145, from one forward pass): returns
        ``(terms, grad, dets, count[, keep_idx])``, each the bits ``loss_fwd_bwd`` and
        ``postprocess(..., rows=rows)`` return.  At S=7, Bx=2, C=20 in reference mode one
        kernel reads the head tensor once for both."""
        out, obj_mask, bsz, nrows = self._check_step_args(out, gt_norm_xywh, gt_cls_prob, obj_mask, mode, rows)
        dev = out.device
        lib = _lib.load()
        gb = bsz if global_batch is None else int(global_batch)
        shape = (bsz, self.n_cells, self.n_bboxes, self.n_classes)
        ws = self._get_workspace(dev, lib.yh_workspace_bytes(_lib.YH_OP_LOSS, *shape))
        grad = torch.empty_like(out)
        terms = torch.empty(5, dtype=torch.float32, device=dev)
        dets = torch.empty(bsz, nrows, 6, dtype=torch.float32, device=dev)
        count = torch.empty(bsz, dtype=torch.int32, device=dev)
        kidx = torch.empty(bsz, nrows, dtype=torch.int32, device=dev) if return_idx else None
        with _on_device(dev):
            pending = self._fin_event.pop(dev, None)
            if pending is not None:
                torch.cuda.current_stream(dev).wait_event(pending)
            _lib.check(lib.yh_head_step(
                out.data_ptr(), gt_norm_xywh.data_ptr(), gt_cls_prob.data_ptr(), obj_mask.data_ptr(),
                grad.data_ptr(), terms.data_ptr(), dets.data_ptr(), count.data_ptr(), _ptr(kidx),
                ws.data_ptr(), ws.numel(), int(mode), float(conf_thresh), float(iou_thresh),
                0 if rows is None else nrows, *shape, self.img_size, self.coord_coeff,
                self.noobj_coeff, 1.0 / max(gb, 1), _stream(dev)))
        self.last_loss_terms = terms
        return (terms, grad, dets, count, kidx) if return_idx else (terms, grad, dets, count)

    def head_step_pipelined(self, out, gt_norm_xywh, gt_cls_prob, obj_mask, log: "LossLog",
                            conf_thresh=0.5, iou_thresh=0.5, mode=_lib.YH_MODE_REFERENCE,
                            global_batch=None, rows: Optional[int] = None,
                            terms_prev: Optional[torch.Tensor] = None):
        """``head_step`` as ONE launch per step: the loss sums reach ``log`` during the next
        call (or at ``flush_loss_log``), as with ``loss_step_pipelined``, whose workspace
        and phase counter it shares.  Returns ``(grad, dets, count)``."""
        out, obj_mask, bsz, nrows = self._check_step_args(out, gt_norm_xywh, gt_cls_prob, obj_mask, mode, rows)
        dev = out.device
        if log.ring.device != dev:
            raise ValueError(f"loss log on {log.ring.device}, out is on {dev}")
        if terms_prev is not None:
            _need(terms_prev, "terms_prev", torch.float32, (5,))
        lib = _lib.load()
        gb = bsz if global_batch is None else int(global_batch)
        shape = (bsz, self.n_cells, self.n_bboxes, self.n_classes)
        nbytes = lib.yh_workspace_bytes(_lib.YH_OP_LOSS_PIPELINED, *shape)
        state = self._pipe.get(dev)
        if state is None or state[0].numel() < nbytes:
            if state is not None:
                self.flush_loss_log(log)         # the halves move: log what is pending first
            state = [torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev), 1]
            self._pipe[dev] = state
        ws, phase = state[0], 1 - state[1]
        grad = torch.empty_like(out)
        dets = torch.empty(bsz, nrows, 6, dtype=torch.float32, device=dev)
        count = torch.empty(bsz, dtype=torch.int32, device=dev)
        with _on_device(dev):
            _lib.check(lib.yh_head_step_pipelined(
                out.data_ptr(), gt_norm_xywh.data_ptr(), gt_cls_prob.data_ptr(), obj_mask.data_ptr(),
                grad.data_ptr(), dets.data_ptr(), count.data_ptr(), None, _ptr(terms_prev),
                log.ring.data_ptr(), log.capacity, log.cursor.data_ptr(), ws.data_ptr(), ws.numel(),
                phase, int(mode), float(conf_thresh), float(iou_thresh), 0 if rows is None else nrows,
                *shape, self.img_size, self.coord_coeff, self.noobj_coeff, 1.0 / max(gb, 1),
                _stream(dev)))
        state[1] = phase
        return grad, dets, count

    def flush_loss_log(self, log: "LossLog", terms: Optional[torch.Tensor] = None):
        """Logs the step ``loss_step_pipelined`` still holds back (no-op if none)."""
        state = self._pipe.get(log.ring.device)
        if state is None:
            return
        with _on_device(log.ring.device):
            _lib.check(_lib.load().yh_loss_pipelined_flush(
                state[0].data_ptr(), state[0].numel(), state[1], _ptr(terms), log.ring.data_ptr(),
                log.capacity, log.cursor.data_ptr(), _stream(log.ring.device)))

    def loss_fwd_bwd(self, out, gt_norm_xywh, gt_cls_prob, obj_mask, global_batch=None,
                     log: Optional[LossLog] = None,
                     finalize_stream: Optional[torch.cuda.Stream] = None):
        """Functional form: one kernel pass returning ``(terms, grad)`` with
        terms = [coord_sum, conf_sum, cls_sum, total_sum, total/global_batch] and
        grad = d(total/global_batch)/d(out).  No autograd graph is recorded.
        ``log`` receives the four sums as one more row (see LossLog).

        ``finalize_stream``: stream for the small kernel that adds the per-workgroup
        loss rows.  Nothing downstream of the gradient needs the scalar, so it can
        overlap whatever the current stream runs next.  The next loss call on this
        device waits for it by itself (it shares the workspace); the caller makes
        the current stream wait (``wait_stream``) before reading ``terms``/the log."""
        gb = out.shape[0] if global_batch is None else int(global_batch)
        return self._launch_loss(out.detach(), gt_norm_xywh, gt_cls_prob, obj_mask,
                                 1.0 / max(gb, 1), log, finalize_stream)

    def loss_fwd_bwd_linear(self, z, bias, gt_norm_xywh, gt_cls_prob, obj_mask, global_batch=None,
                            want_bias_grad=True):
        """The last layer's bias and view folded into the kernel's read: ``z`` (B, S*S*D)
        is nn.Linear's product without its bias (model.py:133-135), ``bias`` that
        bias.  Returns (terms, grad_z, grad_bias); grad_z has z's shape, grad_bias is
        the fixed-order sum of grad_z over the batch (None unless wanted)."""
        if z.dim() != 2 or z.shape[1] != self.depth * self.n_grid:
            raise ValueError(f"z: expected shape (B, {self.depth * self.n_grid}), got {tuple(z.shape)}")
        out = self._check_out(z.detach().view(-1, self.depth, self.n_cells, self.n_cells))
        bias = self._check_bias(bias, out.device)
        bsz, n = out.shape[0], self.n_grid
        _need(gt_norm_xywh, "gt_norm_xywh", torch.float32, (bsz, n, 1, 4))
        _need(gt_cls_prob, "gt_cls_prob", torch.float32, (bsz, n, 1, self.n_classes))
        if obj_mask.dtype == torch.bool:
            obj_mask = obj_mask.view(torch.uint8)
        _need(obj_mask, "obj_mask", torch.uint8, (bsz, n, 1, 1))
        for name, t in (("gt_norm_xywh", gt_norm_xywh), ("gt_cls_prob", gt_cls_prob),
                        ("obj_mask", obj_mask)):
            if t.device != out.device:
                raise ValueError(f"{name}: on {t.device}, z is on {out.device}")
        lib = _lib.load()
        gb = bsz if global_batch is None else int(global_batch)
        shape = (bsz, self.n_cells, self.n_bboxes, self.n_classes)
        ncols = self.depth * n
        nbytes = max(lib.yh_workspace_bytes(_lib.YH_OP_LOSS, *shape),
                     lib.yh_bias_grad_workspace_bytes(bsz, ncols))
        ws = self._get_workspace(out.device, nbytes)
        grad = torch.empty_like(z)
        terms = torch.empty(5, dtype=torch.float32, device=out.device)
        grad_bias = None
        with _on_device(out.device):
            st = _stream(out.device)
            _lib.check(lib.yh_loss_fwd_bwd_linear(
                out.data_ptr(), bias.data_ptr(), gt_norm_xywh.data_ptr(), gt_cls_prob.data_ptr(),
                obj_mask.data_ptr(), grad.data_ptr(), terms.data_ptr(), ws.data_ptr(), ws.numel(),
                *shape, self.img_size, self.coord_coeff, self.noobj_coeff, 1.0 / max(gb, 1), st))
            if want_bias_grad:
                grad_bias = torch.empty(ncols, dtype=torch.float32, device=out.device)
                _lib.check(lib.yh_bias_grad(grad.data_ptr(), grad_bias.data_ptr(), ws.data_ptr(),
                                            ws.numel(), bsz, ncols, st))
        return terms, grad, grad_bias

    def get_loss_linear(self, z, bias, gt_norm_xywh, gt_cls_prob, obj_mask, global_batch=None):
        """``get_loss`` for a model whose last nn.Linear hands over its product and its
        bias separately (model.py:133-135, 146-148): a 0-d loss whose ``.backward()``
        reaches both ``z`` and ``bias``."""
        if torch.is_grad_enabled() and (z.requires_grad or bias.requires_grad):
            return _LinearLossFunction.apply(z, bias, gt_norm_xywh, gt_cls_prob, obj_mask, self,
                                             global_batch)
        terms, _, _ = self.loss_fwd_bwd_linear(z, bias, gt_norm_xywh, gt_cls_prob, obj_mask,
                                               global_batch, want_bias_grad=False)
        self.last_loss_terms = terms
        return terms[4]

    def get_loss(self, image, gt_norm_xywh, gt_cls_prob, obj_mask, global_batch=None):
        """model.py:296-317: (coord + conf + cls) / B as a 0-d tensor that supports
        ``.backward()``.  ``image`` goes through the attached backbone; without
        one it is taken as the head tensor itself.  ``global_batch`` replaces B
        when this call sees one shard of a data-parallel batch (the shard losses
        then add up to the global loss)."""
        out = self(image)
        if out.dtype in (torch.float16, torch.bfloat16):
            out = out.float()      # a backbone run under autocast: the head computes in fp32
        gb = out.shape[0] if global_batch is None else int(global_batch)
        inv = 1.0 / max(gb, 1)
        if torch.is_grad_enabled() and out.requires_grad:
            return _LossFunction.apply(out, gt_norm_xywh, gt_cls_prob, obj_mask, self, inv)
        terms, _ = self._launch_loss(out.detach(), gt_norm_xywh, gt_cls_prob, obj_mask, inv)
        self.last_loss_terms = terms
        return terms[4]

    # ------------------------------------------------------------------ targets
    def encode_targets(self, gt_ltrb, gt_label, gt_count):
        """Ground truth -> loss inputs on the GPU (data.py:173-194, model.py:213-222).

        gt_ltrb (B, M, 4) f32 pixel boxes, gt_label (B, M) integer labels, gt_count (B,)
        number of real boxes per image (padding after them).  Returns gt_norm_xywh
        (B, S*S, 1, 4), gt_cls_prob (B, S*S, 1, C), obj_mask (B, S*S, 1, 1) bool;
        the last box that falls into a cell wins."""
        _need(gt_ltrb, "gt_ltrb", torch.float32)
        if gt_ltrb.dim() != 3 or gt_ltrb.shape[2] != 4:
            raise ValueError(f"gt_ltrb: expected shape (B, M, 4), got {tuple(gt_ltrb.shape)}")
        bsz, m, dev = gt_ltrb.shape[0], gt_ltrb.shape[1], gt_ltrb.device
        gt_label = gt_label.to(device=dev, dtype=torch.int32).contiguous()
        gt_count = gt_count.to(device=dev, dtype=torch.int32).contiguous()
        _need(gt_label, "gt_label", torch.int32, (bsz, m))
        _need(gt_count, "gt_count", torch.int32, (bsz,))
        n = self.n_grid
        gx = torch.empty(bsz, n, 1, 4, dtype=torch.float32, device=dev)
        gc = torch.empty(bsz, n, 1, self.n_classes, dtype=torch.float32, device=dev)
        om = torch.empty(bsz, n, 1, 1, dtype=torch.uint8, device=dev)
        with _on_device(dev):
            _lib.check(_lib.load().yh_encode_targets(
                gt_ltrb.data_ptr(), gt_label.data_ptr(), gt_count.data_ptr(), gx.data_ptr(),
                gc.data_ptr(), om.data_ptr(), bsz, m, self.n_cells, self.n_classes,
                self.img_size, _stream(dev)))
        return gx, gc, om.view(torch.bool)

    # ------------------------------------------------------------------- decode
    def _decode(self, out, want, conf_thresh=0.0):
        out = self._check_out(out.detach())
        bsz, n, dev = out.shape[0], self.n_grid, out.device
        res = {}
        if "ltrb" in want:
            res["ltrb"] = torch.empty(bsz, n, self.n_bboxes, 4, dtype=torch.float32, device=dev)
        if "conf" in want:
            res["conf"] = torch.empty(bsz, n, self.n_bboxes, 1, dtype=torch.float32, device=dev)
        if "cls_prob" in want:
            res["cls_prob"] = torch.empty(bsz, n, 1, self.n_classes, dtype=torch.float32, device=dev)
        if "cls_idx" in want:
            res["cls_idx"] = torch.empty(bsz, n, 1, 1, dtype=torch.int64, device=dev)
        if "best_k" in want:
            res["best_k"] = torch.empty(bsz, n, 1, 1, dtype=torch.int64, device=dev)
        if "best_ltrb" in want:
            res["best_ltrb"] = torch.empty(bsz, n, 1, 4, dtype=torch.float32, device=dev)
        if "conf_mask" in want:
            res["conf_mask"] = torch.empty(bsz, n, 1, 1, dtype=torch.uint8, device=dev)
        with _on_device(dev):
            _lib.check(_lib.load().yh_decode_grid(
                out.data_ptr(), _ptr(res.get("ltrb")), _ptr(res.get("conf")),
                _ptr(res.get("cls_prob")), _ptr(res.get("cls_idx")), _ptr(res.get("best_k")),
                _ptr(res.get("best_ltrb")), _ptr(res.get("conf_mask")), float(conf_thresh),
                bsz, self.n_cells, self.n_bboxes, self.n_classes, self.img_size, _stream(dev)))
        return res

    def get_ltrb(self, out):
        """model.py:185-188, batched: (B, S*S, Bx, 4) pixel boxes."""
        return self._decode(out, ("ltrb",))["ltrb"]

    def get_norm_xywh(self, out):
        """model.py:157-165, batched: (B, S*S, Bx, 4) sigmoid of the box channels,
        cell index i*S + j."""
        out = self._check_out(out.detach())
        res = torch.empty(out.shape[0], self.n_grid, self.n_bboxes, 4, dtype=torch.float32,
                          device=out.device)
        with _on_device(out.device):
            _lib.check(_lib.load().yh_norm_xywh(out.data_ptr(), res.data_ptr(), out.shape[0],
                                                self.n_cells, self.n_bboxes, self.n_classes,
                                                _stream(out.device)))
        return res

    def denormalize_xywh(self, norm_xywh):
        """model.py:167-183 on (B, S*S, K, 4) (or (S*S, K, 4) for one image): pixels,
        the first grid index carries x."""
        _need(norm_xywh, "norm_xywh", torch.float32)
        shape = norm_xywh.shape
        if norm_xywh.dim() not in (3, 4) or shape[-1] != 4 or shape[-3] != self.n_grid:
            raise ValueError(f"norm_xywh: expected (B, {self.n_grid}, K, 4), got {tuple(shape)}")
        bsz = shape[0] if norm_xywh.dim() == 4 else 1
        res = torch.empty_like(norm_xywh)
        with _on_device(norm_xywh.device):
            _lib.check(_lib.load().yh_denormalize_xywh(
                norm_xywh.data_ptr(), res.data_ptr(), bsz, self.n_cells, shape[-2], self.img_size,
                _stream(norm_xywh.device)))
        return res

    def xywh_to_ltrb(self, xywh):
        """model.py:150-155 on (..., 4): one-sided clips at 0 and img_size."""
        _need(xywh, "xywh", torch.float32)
        if xywh.shape[-1] != 4:
            raise ValueError(f"xywh: expected (..., 4), got {tuple(xywh.shape)}")
        res = torch.empty_like(xywh)
        with _on_device(xywh.device):
            _lib.check(_lib.load().yh_xywh_to_ltrb(xywh.data_ptr(), res.data_ptr(),
                                                   xywh.numel() // 4, self.img_size,
                                                   _stream(xywh.device)))
        return res

    def _get_responsibility_mask(self, out, gt_norm_xywh, obj_mask):
        """model.py:190-211 as get_loss needs it: (B, S*S, Bx, 1) bool, True for the
        predictor with the highest IoU against the ground truth of a cell that
        holds an object (first index on ties)."""
        out = self._check_out(out.detach())
        bsz, n = out.shape[0], self.n_grid
        _need(gt_norm_xywh, "gt_norm_xywh", torch.float32, (bsz, n, 1, 4))
        if obj_mask.dtype == torch.bool:
            obj_mask = obj_mask.view(torch.uint8)
        _need(obj_mask, "obj_mask", torch.uint8, (bsz, n, 1, 1))
        res = torch.empty(bsz, n, self.n_bboxes, 1, dtype=torch.uint8, device=out.device)
        with _on_device(out.device):
            _lib.check(_lib.load().yh_responsibility_mask(
                out.data_ptr(), gt_norm_xywh.data_ptr(), obj_mask.data_ptr(), res.data_ptr(), bsz,
                self.n_cells, self.n_bboxes, self.n_classes, self.img_size, _stream(out.device)))
        return res.view(torch.bool)

    def get_object_mask(self, gt_ltrb):
        """model.py:213-222: (S, S) bool grid of one image's boxes (M, 4), indexed
        [x_cell, y_cell]."""
        _need(gt_ltrb, "gt_ltrb", torch.float32)
        m = gt_ltrb.shape[0]
        if m == 0:
            return torch.zeros(self.n_cells, self.n_cells, dtype=torch.bool, device=gt_ltrb.device)
        label = torch.zeros(1, m, dtype=torch.int32, device=gt_ltrb.device)
        count = torch.full((1,), m, dtype=torch.int32, device=gt_ltrb.device)
        _, _, om = self.encode_targets(gt_ltrb.view(1, m, 4), label, count)
        return om.view(self.n_cells, self.n_cells)

    def get_loss_terms(self, out, gt_norm_xywh, gt_cls_prob, obj_mask):
        """The three sums get_loss adds (model.py:311-313), undivided: coordinate,
        confidence, classification.  Values only; gradients flow through get_loss."""
        terms, _ = self._launch_loss(self._check_out(out.detach()), gt_norm_xywh, gt_cls_prob,
                                     obj_mask, 1.0)
        return terms[0], terms[1], terms[2]

    def get_loss_per_image(self, out, gt_norm_xywh, gt_cls_prob, obj_mask):
        """The loss terms of every image by itself: (B, 4) = coordinate, confidence and class
        sums and their total (not divided by anything); ``.sum(0) / B`` is what ``get_loss``
        returns, to rounding (model.py:296-317 forms only the batch sums)."""
        out, obj_mask, bsz, _ = self._check_step_args(out, gt_norm_xywh, gt_cls_prob, obj_mask,
                                                      _lib.YH_MODE_REFERENCE, 1, limit=False)
        per_image = torch.empty(bsz, 4, dtype=torch.float32, device=out.device)
        with _on_device(out.device):
            _lib.check(_lib.load().yh_loss_per_image(
                out.data_ptr(), gt_norm_xywh.data_ptr(), gt_cls_prob.data_ptr(), obj_mask.data_ptr(),
                per_image.data_ptr(), bsz, self.n_cells, self.n_bboxes, self.n_classes, self.img_size,
                self.coord_coeff, self.noobj_coeff, _stream(out.device)))
        return per_image

    def get_coordinate_loss(self, out, gt_norm_xywh, gt_cls_prob, obj_mask):
        """model.py:224-240 (the arguments are those get_loss has in scope)."""
        return self.get_loss_terms(out, gt_norm_xywh, gt_cls_prob, obj_mask)[0]

    def get_confidence_loss(self, out, gt_norm_xywh, gt_cls_prob, obj_mask):
        """model.py:252-276."""
        return self.get_loss_terms(out, gt_norm_xywh, gt_cls_prob, obj_mask)[1]

    def get_classification_loss(self, out, gt_norm_xywh, gt_cls_prob, obj_mask):
        """model.py:288-294."""
        return self.get_loss_terms(out, gt_norm_xywh, gt_cls_prob, obj_mask)[2]

    def get_confidence(self, out):
        """model.py:242-250: (B, S*S, Bx, 1)."""
        return self._decode(out, ("conf",))["conf"]

    def get_classification_prob(self, out):
        """model.py:278-286: (B, S*S, 1, C) softmax."""
        return self._decode(out, ("cls_prob",))["cls_prob"]

    def _get_classification_index(self, out):
        """model.py:392-394: (B, S*S, 1, 1) int64."""
        return self._decode(out, ("cls_idx",))["cls_idx"]

    def get_confidence_mask(self, out, conf_thresh):
        """model.py:379-382: (B, S*S, 1, 1) bool, max_k conf >= conf_thresh."""
        return self._decode(out, ("conf_mask",), conf_thresh)["conf_mask"].view(torch.bool)

    def get_max_confidence_ltrb(self, out):
        """model.py:384-390: (B, S*S, 1, 4) box of the higher-confidence predictor."""
        return self._decode(out, ("best_ltrb",))["best_ltrb"]

    # ---------------------------------------------------------------------- NMS
    def decode_candidates(self, out, mode=_lib.YH_MODE_REFERENCE):
        """The NMS candidates of ``postprocess`` as tensors: ltrb (B, N, 4),
        score (B, N), cls (B, N) int32, present (B, N) bool; N = S*S*Bx."""
        out = self._check_out(out.detach())
        bsz, nn_, dev = out.shape[0], self.n_grid * self.n_bboxes, out.device
        ltrb = torch.empty(bsz, nn_, 4, dtype=torch.float32, device=dev)
        score = torch.empty(bsz, nn_, dtype=torch.float32, device=dev)
        cls = torch.empty(bsz, nn_, dtype=torch.int32, device=dev)
        present = torch.empty(bsz, nn_, dtype=torch.uint8, device=dev)
        with _on_device(dev):
            _lib.check(_lib.load().yh_decode_candidates(
                out.data_ptr(), ltrb.data_ptr(), score.data_ptr(), cls.data_ptr(),
                present.data_ptr(), int(mode), bsz, self.n_cells, self.n_bboxes, self.n_classes,
                self.img_size, _stream(dev)))
        return ltrb, score, cls, present.view(torch.bool)

    def postprocess(self, out, conf_thresh=0.5, iou_thresh=0.5, mode=_lib.YH_MODE_REFERENCE,
                    return_idx=False, bias: Optional[torch.Tensor] = None,
                    rows: Optional[int] = None):
        """Decode + confidence threshold + per-class NMS in one kernel.

        With ``bias`` (S*S*D floats), ``out`` is the last layer's product before its
        bias (model.py:133-135) and the bias is added as the kernel reads it.
        With ``rows`` the detection tensor has that many rows per image instead of
        S*S*Bx (at least the candidates of the mode: S*S in reference mode, which keeps
        one box per cell, model.py:384-390); channel-major ``out`` without a bias only.

        Returns dets (B, S*S*Bx, 6) with the kept rows ``l, t, r, b, conf, class``
        first (the row layout eval.py:89-91 consumes; class ascending, then
        confidence descending) and zero rows after them, keep_count (B,) int32,
        and with ``return_idx`` keep_idx (B, S*S*Bx) int32 (flat index
        cell*Bx + k of each kept row, -1 padding)."""
        out = self._check_out(out.detach(), allow_channels_last=True)
        nhwc = self._cell_major(out)
        bsz, nn_, dev = out.shape[0], self.n_grid * self.n_bboxes, out.device
        if nhwc and (bias is not None or nn_ > _lib.YH_MAX_CANDIDATES):
            raise ValueError("out: a channels-last head tensor is read in place only without a bias and "
                             f"up to {_lib.YH_MAX_CANDIDATES} candidates per image; pass out.contiguous()")
        if rows is not None:
            if nhwc or bias is not None or nn_ > _lib.YH_MAX_CANDIDATES:
                raise ValueError("rows: only for a channel-major head tensor without a bias and up to "
                                 f"{_lib.YH_MAX_CANDIDATES} candidates per image")
            nrows = int(rows)
        else:
            nrows = nn_
        dets = torch.empty(bsz, nrows, 6, dtype=torch.float32, device=dev)
        count = torch.empty(bsz, dtype=torch.int32, device=dev)
        kidx = torch.empty(bsz, nrows, dtype=torch.int32, device=dev) if return_idx else None
        tail = (dets.data_ptr(), count.data_ptr(), _ptr(kidx), int(mode), float(conf_thresh),
                float(iou_thresh), bsz, self.n_cells, self.n_bboxes, self.n_classes, self.img_size)
        lib = _lib.load()
        if bias is not None:
            bias = self._check_bias(bias, dev)
        with _on_device(dev):
            if nn_ > _lib.YH_MAX_CANDIDATES:
                # more candidates per image than a CU's LDS holds: the global-memory path
                ws = torch.empty(max(256, lib.yh_nms_workspace_bytes(bsz, nn_, 1)), dtype=torch.uint8,
                                 device=dev)
                _lib.check(lib.yh_decode_nms_ws(out.data_ptr(), _ptr(bias), *tail, ws.data_ptr(),
                                                ws.numel(), _stream(dev)))
            elif nhwc:
                _lib.check(lib.yh_decode_nms_nhwc(out.data_ptr(), *tail, _stream(dev)))
            elif rows is not None:
                _lib.check(lib.yh_decode_nms_rows(out.data_ptr(), *tail[:6], nrows, *tail[6:], _stream(dev)))
            elif bias is None:
                _lib.check(lib.yh_decode_nms(out.data_ptr(), *tail, _stream(dev)))
            else:
                _lib.check(lib.yh_decode_nms_linear(out.data_ptr(), bias.data_ptr(), *tail, _stream(dev)))
        return (dets, count, kidx) if return_idx else (dets, count)

    @staticmethod
    def nms(ltrb, score, cls, present=None, conf_thresh=0.0, iou_thresh=0.5, return_idx=False):
        """Per-class NMS over caller-supplied candidates: ltrb (B, N, 4) f32,
        score (B, N) f32, cls (B, N) int32, present (B, N) bool or None."""
        _need(ltrb, "ltrb", torch.float32)
        bsz, n = ltrb.shape[0], ltrb.shape[1]
        _need(ltrb, "ltrb", torch.float32, (bsz, n, 4))
        _need(score, "score", torch.float32, (bsz, n))
        _need(cls, "cls", torch.int32, (bsz, n))
        if present is not None:
            if present.dtype == torch.bool:
                present = present.view(torch.uint8)
            _need(present, "present", torch.uint8, (bsz, n))
        dev = ltrb.device
        dets = torch.empty(bsz, n, 6, dtype=torch.float32, device=dev)
        count = torch.empty(bsz, dtype=torch.int32, device=dev)
        kidx = torch.empty(bsz, n, dtype=torch.int32, device=dev) if return_idx else None
        lib = _lib.load()
        with _on_device(dev):
            ws = None
            if n > _lib.YH_MAX_CANDIDATES:
                ws = torch.empty(max(256, lib.yh_nms_workspace_bytes(bsz, n, 0)), dtype=torch.uint8,
                                 device=dev)
            _lib.check(lib.yh_nms_ws(
                ltrb.data_ptr(), score.data_ptr(), cls.data_ptr(), _ptr(present),
                dets.data_ptr(), count.data_ptr(), _ptr(kidx), float(conf_thresh),
                float(iou_thresh), bsz, n, _ptr(ws), 0 if ws is None else ws.numel(), _stream(dev)))
        return (dets, count, kidx) if return_idx else (dets, count)

    def detections(self, out, conf_thresh=0.5, iou_thresh=0.5, mode=_lib.YH_MODE_REFERENCE):
        """Host-side convenience: a list with one (n_i, 6) tensor per image -- the
        ``pred`` argument of eval.get_map (eval.py:86-94).  Synchronises."""
        dets, count = self.postprocess(out, conf_thresh, iou_thresh, mode)
        counts = count.tolist()
        return [dets[b, :k] for b, k in enumerate(counts)]
