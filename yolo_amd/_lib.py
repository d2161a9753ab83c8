"""ctypes binding of libyolohead.so (the C ABI of include/yolo_head.h).

The library is built in-tree by ``yolo_amd/csrc/Makefile`` (or
``__graft_entry__.build()``) into ``yolo_amd/lib/``.  There is no fallback: if
the shared object is missing, loading raises and every op of the package fails.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  -- must be imported first: its HIP runtime is the one the library binds to

LIB_PATH = os.environ.get(
    "YOLO_AMD_LIB",  # measurement builds only (tools/); the package ships and tests the path below
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libyolohead.so"))

# error codes of include/yolo_head.h
YH_OK = 0
YH_ERR_NULL = -1
YH_ERR_SHAPE = -2
YH_ERR_WORKSPACE = -3
YH_ERR_ALIGN = -4
YH_ERR_ARG = -5
YH_ERR_UNSUPPORTED = -6
YH_ERR_SCRATCH = -7

YH_OP_LOSS = 0
YH_OP_DECODE_NMS = 1
YH_OP_DECODE_GRID = 2
YH_OP_NMS = 3
YH_OP_MAP = 4
YH_OP_LOSS_PIPELINED = 5

YH_MODE_REFERENCE = 0
YH_MODE_DENSE = 1

YH_MAX_CANDIDATES = 2048
YH_MAX_CANDIDATES_WS = 65536
YH_STATIC_SCRATCH_BYTES = 262144

YH_BOX_IOU = 0
YH_BOX_GIOU = 1
YH_BOX_INTERSECTION = 2
YH_BOX_ENCLOSING = 3

# name -> (restype, argtypes); mirrors include/yolo_head.h declaration by declaration
SIGNATURES = {
    "yh_version": (c_int, []),
    "yh_strerror": (c_char_p, [c_int]),
    "yh_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int, c_int]),
    "yh_loss_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_int64, c_int, c_int, c_int, c_int,
                                c_float, c_float, c_float, c_void_p]),
    "yh_loss_fwd_bwd_static": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int64, c_int, c_int, c_int, c_int, c_float, c_float,
                                       c_float, c_void_p]),
    "yh_loss_fwd_bwd_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_int64, c_int, c_int, c_int, c_int,
                                     c_float, c_float, c_float, c_void_p]),
    "yh_decode_nms_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float,
                                   c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "yh_loss_per_image": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                  c_int, c_int, c_float, c_float, c_void_p]),
    "yh_loss_partials": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_size_t, c_int64, c_int, c_int, c_int, c_int, c_float, c_float,
                                 c_float, c_void_p]),
    "yh_loss_finalize": (c_int, [c_void_p, c_size_t, c_int64, c_int, c_float, c_void_p, c_void_p,
                                 c_int, c_void_p, c_void_p]),
    "yh_loss_workspace_init": (c_int, [c_void_p, c_size_t, c_void_p]),
    "yh_loss_fwd_bwd_pipelined": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                          c_int, c_int64, c_int, c_int, c_int, c_int, c_float,
                                          c_float, c_float, c_void_p]),
    "yh_loss_pipelined_flush": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_int,
                                        c_void_p, c_void_p]),
    "yh_loss_fwd_bwd_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int, c_int,
                                       c_int, c_float, c_float, c_float, c_void_p]),
    "yh_decode_nms_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_float, c_float, c_int64, c_int, c_int, c_int, c_int,
                                     c_void_p]),
    "yh_bias_grad_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "yh_bias_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int, c_void_p]),
    "yh_scale_grad": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "yh_pairwise_boxes": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "yh_box_area": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "yh_map_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "yh_map_11point": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_float, c_void_p,
                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "yh_encode_targets": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "yh_decode_grid": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_float, c_int64, c_int, c_int, c_int, c_int,
                               c_void_p]),
    "yh_norm_xywh": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "yh_denormalize_xywh": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "yh_xywh_to_ltrb": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "yh_responsibility_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                       c_int, c_int, c_int, c_void_p]),
    "yh_decode_candidates": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "yh_decode_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float,
                              c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "yh_decode_nms_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float,
                                   c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "yh_head_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_float, c_float, c_int,
                             c_int64, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    "yh_head_step_pipelined": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                       c_size_t, c_int, c_int, c_float, c_float, c_int, c_int64, c_int,
                                       c_int, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    "yh_nms_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "yh_decode_nms_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                 c_float, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                 c_void_p]),
    "yh_nms_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_float, c_float, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "yh_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_float, c_float, c_int64, c_int, c_void_p]),
}

_lib = None


class YoloHeadError(RuntimeError):
    def __init__(self, rc: int, msg: str):
        super().__init__(f"{msg} (rc={rc})")
        self.rc = rc


def load() -> ctypes.CDLL:
    """Load the library once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C yolo_amd/csrc` "
                "(hipcc --offload-arch=gfx950); yolo_amd has no CPU or PyTorch fallback"
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def strerror(rc: int) -> str:
    return load().yh_strerror(rc).decode()


def check(rc: int) -> None:
    if rc != YH_OK:
        raise YoloHeadError(rc, strerror(rc))
