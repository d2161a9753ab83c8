"""What the head kernels' read pattern costs by itself: the (batch, 30, 7, 7) head tensor of
the benchmark read once per launch over rotating buffers (HBM proper) -- a lane per cell
with 30 four-byte loads 196 bytes apart (as nms_kernel and head_step_kernel read it), the
workgroup's contiguous block in 16-byte loads, and a plain grid-stride streaming read."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "yolo_amd", "lib", "libyhbench.so"))
lib.yhb_read_head.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
lib.yhb_stream_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_BATCH", "4096"))
NSETS = int(os.environ.get("KB_SETS", "24"))
sets = [torch.randn(B, 30, 7, 7, device=dev) for _ in range(NSETS)]
sink = torch.zeros(1, device=dev)
st = torch.cuda.current_stream().cuda_stream
nbytes = B * 5880


def timeit(fn, iters=480):
    for i in range(2 * NSETS): fn(i % NSETS)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fn(i % NSETS)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters


for name, fn in (("lane per cell, 30 x 4-byte loads", lambda i: lib.yhb_read_head(sets[i].data_ptr(), sink.data_ptr(), B, 0, st)),
                 ("block of 4 images, 16-byte loads", lambda i: lib.yhb_read_head(sets[i].data_ptr(), sink.data_ptr(), B, 1, st)),
                 ("grid-stride streaming read", lambda i: lib.yhb_stream_read(sets[i].data_ptr(), sink.data_ptr(), nbytes, 4096, st))):
    us = timeit(fn)
    print(f"{name:36s}: {us:6.2f} us  {nbytes / us / 1e6:5.2f} TB/s  ({nbytes / 1e6:.1f} MB per launch, {NSETS} buffers in rotation)", flush=True)
