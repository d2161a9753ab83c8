"""Calibration: what a plain streaming copy reaches on this part.  The hand-written
copy of libyhbench (16 bytes per lane, grid-stride, `unroll` loads in flight, loads and
stores non-temporal or not) over sizes from the step's own bytes to 4 GiB of traffic, and
torch's copy_ beside it.  bench.py uses the configuration that wins at 2 GiB."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "yolo_amd", "lib", "libyhbench.so"))
lib.yhb_stream_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.yhb_stream_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters


sizes = [float(x) for x in sys.argv[1:]] or [67.6, 540.0, 2147.5, 4295.0]
sink = torch.zeros(1, device=dev)
for total_mb in sizes:
    n = int(total_mb * 1e6 / 2 / 16) * 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    nb = n * 4
    st = torch.cuda.current_stream().cuda_stream
    iters = max(5, int(20000 / total_mb))
    us = timeit(lambda: y.copy_(x), iters)
    print(f"{total_mb:7.1f} MB torch copy_                     : {us:8.2f} us {total_mb / us:6.2f} TB/s", flush=True)
    best = None
    for blocks in (2048, 4096, 8192, 16384):
        for unroll in (1, 2, 4, 8):
            for flags in (0, 1, 2, 3):
                us = timeit(lambda: lib.yhb_stream_copy(y.data_ptr(), x.data_ptr(), nb, blocks, unroll, flags, st), iters)
                if best is None or us < best[0]:
                    best = (us, blocks, unroll, flags)
                if os.environ.get("CC_ALL"):
                    print(f"          blocks {blocks:5d} unroll {unroll} flags {flags}: {us:8.2f} us {total_mb / us:6.2f} TB/s")
    us, blocks, unroll, flags = best
    assert torch.equal(x, y)
    print(f"{total_mb:7.1f} MB stream copy (blocks {blocks}, unroll {unroll}, flags {flags}): {us:8.2f} us "
          f"{total_mb / us:6.2f} TB/s", flush=True)
    us = timeit(lambda: lib.yhb_stream_read(x.data_ptr(), sink.data_ptr(), nb, 4096, st), iters)
    print(f"{total_mb / 2:7.1f} MB read only                       : {us:8.2f} us {total_mb / 2 / us:6.2f} TB/s", flush=True)
    del x, y
