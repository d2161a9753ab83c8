#!/usr/bin/env python3
"""Headline benchmark: YOLOv1 head images/sec (loss fwd+bwd + decode/NMS).

One "step" = one pass of the hot path over one batch that is already resident
in HBM: the fused loss forward+backward (gradient w.r.t. the head tensor and
the loss sums) and decode + confidence threshold + per-class NMS of the same
head tensor.  The two halves are independent readers of the head tensor; they
run on two HIP streams inside ONE captured graph per chunk of steps.

Regime.  By default the timed region ROTATES through as many buffer sets as put
twice the 256 MiB memory-side cache (Infinity Cache) behind every kernel's own
bytes, so no step finds its batch in that cache: `value` is an HBM-proper
figure (`config.regime`).  The same step on ONE resident buffer set, which the
cache serves in part, is reported beside it as `cache_resident`
(`--regime resident` makes that the timed region instead).

Workloads (S=7, Bx=2, C=20, fp32, synthetic logits and targets, yolo_amd.synth):
  --gpus 1   batch 4096 on one GPU.
  --gpus N   batch 65,536 sharded by image over the N GPUs (strong scaling: N=2/4/8
             hold 32,768/16,384/8,192 images each); the loss is divided by the
             GLOBAL batch (model.py:317) and the only exchange is the all-reduce of
             the per-step loss sums, batched over `--allreduce-every` steps and
             completed inside the timed region.  `--scaling weak` keeps --batch
             images per GPU instead.  The one-GPU line carries `configs.dp_strong_n1`:
             the same step at 65,536 / 32,768 / 16,384 / 8,192 images on ONE GPU, the
             denominators of the strong-scaling curve.
  --force-dist   with --gpus 1: initialise the process group ("nccl" = RCCL) at world
             size 1 and take every branch the N > 1 run takes (barriers, in-place
             asynchronous all-reduce of the ring rows, MAX all-reduce of the time).

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment the command starts its
own `python -m torch.distributed.run` with N ranks (before anything touches a
GPU) and relays rank 0's line.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the
fused loss kernel) over the rotation, i.e. against HBM proper: algorithmic
bytes per launch / its HIP-event duration, against the 8 TB/s HBM3E peak, and
the same with the bytes the PMC counters saw (`frac_counter_bytes`); the
cache-resident figure, the decode+NMS kernel and measured streaming-copy and
streaming-read ceilings sit beside it.  `cpu_baseline` is the CPU oracle (a port
of the reference's formulas, oracle/ref_cpu.py) timed on this host on bounded
samples: 3 warm-up passes, then the median of 20.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (spec)
MALL_BYTES = 256 * 1024 * 1024  # memory-side cache (Infinity Cache)
S, BX, C, IMG = 7, 2, 20, 448
CONF_T, IOU_T = 0.5, 0.5
STRONG_GLOBAL_BATCH = 65536
KERNEL_SOURCES = ("head_math.hpp", "loss_kernels.hpp", "nms_kernels.hpp", "step_kernels.hpp", "yolo_head.hip")


def loss_bytes_per_image(s=S, bx=BX, c=C):
    """Algorithmic bytes of the fused loss kernel per image: head tensor read +
    targets read (xywh, class probabilities, object mask) + gradient written."""
    d, n = 5 * bx + c, s * s
    return 4 * d * n + (n * (4 + c) * 4 + n) + 4 * d * n


def nms_bytes_per_image(s=S, bx=BX, c=C, rows=None):
    """Algorithmic bytes of decode+NMS per image: head tensor read + detections
    written (`rows` rows of 6 floats, S*S*Bx unless given) + keep count."""
    d, n = 5 * bx + c, s * s
    return 4 * d * n + 24 * (n * bx if rows is None else rows) + 4


def kernel_sources_sha():
    """Fingerprint of the kernel sources: a PMC traffic figure measured on other
    sources is not reported."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "yolo_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--batch", type=int, default=4096, help="images per GPU (one GPU, or --scaling weak)")
    ap.add_argument("--scaling", choices=("auto", "strong", "weak"), default="auto",
                    help="auto: one GPU runs --batch images, N > 1 shard --global-batch (strong)")
    ap.add_argument("--global-batch", type=int, default=STRONG_GLOBAL_BATCH,
                    help="images per step over all GPUs under strong scaling")
    ap.add_argument("--allreduce-every", type=int, default=128,
                    help="steps per captured graph; their loss sums share one all-reduce")
    ap.add_argument("--regime", choices=("hbm", "resident"), default="hbm",
                    help="hbm: the timed region rotates through buffer sets worth twice the memory-side "
                         "cache per kernel; resident: one buffer set, served by that cache in part")
    ap.add_argument("--step", choices=("auto", "two-stream", "fused"), default="auto",
                    help="two-stream: loss kernel and decode+NMS kernel on two streams; fused: one "
                         "kernel that reads the head tensor once for both (yh_head_step_pipelined); "
                         "auto: fused when the library has it")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group even at world size 1 and take the N > 1 branches")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--prewarm-seconds", type=float, default=0.3,
                    help="untimed steps run before the warm-up so the GPU clocks are at steady state")
    ap.add_argument("--loss-path", choices=("pipelined", "two-launch"), default="pipelined",
                    help="pipelined: one launch per step, the loss sums reach the device log one "
                         "step late; two-launch: gradient kernel + finalize kernel per step")
    ap.add_argument("--serial", action="store_true",
                    help="run the loss and decode+NMS halves of the step on one stream instead of two")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--verbose", action="store_true", help="progress lines on stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=14.0, help="budget of the whole CPU leg")
    ap.add_argument("--kernel-iters", type=int, default=320)
    ap.add_argument("--cold-sets", type=int, default=0,
                    help="buffer sets rotated through for the HBM-proper figures (0: as many as put "
                         "twice the memory-side cache behind every kernel's own bytes)")
    ap.add_argument("--measure", choices=("all", "step", "rotated", "resident"), default="all",
                    help="step: timed region only; rotated / resident: only that kernel-timing leg "
                         "(what the rocprofv3 passes under profiles/ run)")
    ap.add_argument("--no-stress", action="store_true", help="skip the S=14,B=5,C=80 sub-record")
    ap.add_argument("--no-dp-table", action="store_true", help="skip configs.dp_strong_n1")
    ap.add_argument("--dump-log", default=None, help="write the timed region's loss rows to this .npy file")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N as a plain command: this process has not touched a GPU; it starts N
    ranks as children and leaves with their exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # RCCL's ranks hand each other device buffers through IPC handles; this host driver
    # only supports the dmabuf kind, and with the legacy kind hipIpcGetMemHandle fails
    # ("invalid argument") as soon as two processes share a buffer
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds):
    """The oracle on this host: loss forward + autograd backward + decode/NMS, at one
    thread and at the box's share of cores, on config 1 (batch 16) and on a bounded
    sample of the batch-4096 workload.  Per leg: 3 warm-up passes, then 20 timed passes,
    median; a leg whose 23 passes would not fit its share of the budget halves its
    sample first (never below 16 images)."""
    import statistics

    import torch
    from oracle import ref_cpu
    from yolo_amd import synth

    all_cores = max(1, min(16, os.cpu_count() or 1))
    pred, gx, gc, om = synth.make_inputs(4096, S, BX, C, IMG, seed=0)
    ref_cpu.loss_and_grad(pred[:16], gx[:16], gc[:16], om[:16])        # first-use costs
    ref_cpu.postprocess(pred[:16], CONF_T, IOU_T, mode=0)
    legs = (("batch16", 16, 1), ("batch16", 16, all_cores), ("batch4096", 512, 1),
            ("batch4096", 1024, all_cores))
    share = seconds / len(legs)
    detail = {}

    def one_pass(p, x, c, m):
        t0 = time.perf_counter()
        ref_cpu.loss_and_grad(p, x, c, m)
        ref_cpu.postprocess(p, CONF_T, IOU_T, mode=0)
        return time.perf_counter() - t0

    for name, n, threads in legs:
        torch.set_num_threads(threads)
        while True:
            p, x, c, m = pred[:n], gx[:n], gc[:n], om[:n]
            warm = [one_pass(p, x, c, m) for _ in range(3)]
            if n <= 16 or 20 * min(warm) <= share:
                break
            n //= 2
        times = [one_pass(p, x, c, m) for _ in range(20)]
        med = statistics.median(times)
        detail.setdefault(name, {})[f"threads_{threads}"] = {
            "images_per_s": n / med, "images_per_pass": n, "warmup_passes": 3, "timed_passes": 20,
            "median_pass_s": med, "min_pass_s": min(times), "max_pass_s": max(times)}
    torch.set_num_threads(all_cores)
    top = detail["batch4096"][f"threads_{all_cores}"]
    one = detail["batch4096"]["threads_1"]
    return {
        "value": top["images_per_s"], "unit": "images/s", "cores": all_cores, "kind": "port",
        "sample": f"the first {top['images_per_pass']} images of the batch-4096 workload, 3 warm-up passes "
                  f"then the median of 20: PyTorch fp32 CPU loss forward + autograd backward "
                  f"({all_cores} threads) and single-threaded numpy decode+NMS; the oracle is a chain of "
                  f"small tensor ops and is bound by op dispatch, not by the cores (one thread on "
                  f"{one['images_per_pass']} images: {one['images_per_s']:.0f} images/s); `detail` also "
                  f"holds config 1 (batch 16) at both thread counts",
        "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "detail": detail,
    }


class Kernels:
    """The step's launches straight through the C ABI on caller-owned buffers."""

    def __init__(self, dev, batch, global_batch, pipelined, shape=(S, BX, C, IMG)):
        import torch
        from yolo_amd import _lib
        self.torch, self._lib, self.lib = torch, _lib, _lib.load()
        self.dev, self.batch, self.inv_gb, self.pipelined = dev, batch, 1.0 / global_batch, pipelined
        self.shape = shape
        s, bx, c, _ = shape
        lib = self.lib
        self.ws = torch.empty(max(16, lib.yh_workspace_bytes(_lib.YH_OP_LOSS, batch, s, bx, c)),
                              dtype=torch.uint8, device=dev)
        self.wsp = torch.zeros(lib.yh_workspace_bytes(_lib.YH_OP_LOSS_PIPELINED, batch, s, bx, c),
                               dtype=torch.uint8, device=dev)
        self.terms = torch.empty(5, device=dev)
        self.phase = 1
        self.has_fused = hasattr(lib, "yh_head_step_pipelined") and shape[:3] == (7, 2, 20)

    def rows(self, mode=0):
        """Rows of `dets` per image: what the candidates of the mode can fill."""
        s, bx, _, _ = self.shape
        return s * s if mode == 0 else s * s * bx

    def make_set(self, inputs, mode=0):
        torch = self.torch
        pred, gx, gc, om = (t.to(self.dev).clone() for t in inputs)
        return dict(pred=pred, gx=gx, gc=gc, om=om.view(torch.uint8), grad=torch.empty_like(pred),
                    dets=torch.empty(self.batch, self.rows(mode), 6, device=self.dev),
                    cnt=torch.empty(self.batch, dtype=torch.int32, device=self.dev))

    def loss(self, b, log, stream):
        lib, _lib = self.lib, self._lib
        s, bx, c, img = self.shape
        st = stream.cuda_stream
        if self.pipelined:
            self.phase ^= 1
            _lib.check(lib.yh_loss_fwd_bwd_pipelined(
                b["pred"].data_ptr(), b["gx"].data_ptr(), b["gc"].data_ptr(), b["om"].data_ptr(),
                b["grad"].data_ptr(), None, log.ring.data_ptr(), log.capacity, log.cursor.data_ptr(),
                self.wsp.data_ptr(), self.wsp.numel(), self.phase, self.batch, s, bx, c, img, 5.0, 0.5,
                self.inv_gb, st))
        else:
            _lib.check(lib.yh_loss_partials(
                b["pred"].data_ptr(), b["gx"].data_ptr(), b["gc"].data_ptr(), b["om"].data_ptr(),
                b["grad"].data_ptr(), self.ws.data_ptr(), self.ws.numel(), self.batch, s, bx, c, img,
                5.0, 0.5, self.inv_gb, st))
            _lib.check(lib.yh_loss_finalize(
                self.ws.data_ptr(), self.ws.numel(), self.batch, s, self.inv_gb, self.terms.data_ptr(),
                log.ring.data_ptr(), log.capacity, log.cursor.data_ptr(), st))

    def loss_flush(self, log, stream):
        if self.pipelined:
            self._lib.check(self.lib.yh_loss_pipelined_flush(
                self.wsp.data_ptr(), self.wsp.numel(), self.phase, None, log.ring.data_ptr(),
                log.capacity, log.cursor.data_ptr(), stream.cuda_stream))

    def nms(self, b, stream, mode=0, conf=CONF_T, iou=IOU_T):
        s, bx, c, img = self.shape
        self._lib.check(self.lib.yh_decode_nms_rows(
            b["pred"].data_ptr(), b["dets"].data_ptr(), b["cnt"].data_ptr(), None, mode, conf, iou,
            b["dets"].shape[1], self.batch, s, bx, c, img, stream.cuda_stream))

    def fused(self, b, log, stream, conf=CONF_T, iou=IOU_T):
        """Loss forward+backward and decode+NMS (reference mode) in ONE launch that reads the
        head tensor once; the loss sums reach the log one call late, as in the pipelined loss."""
        lib, _lib = self.lib, self._lib
        s, bx, c, img = self.shape
        self.phase ^= 1
        _lib.check(lib.yh_head_step_pipelined(
            b["pred"].data_ptr(), b["gx"].data_ptr(), b["gc"].data_ptr(), b["om"].data_ptr(),
            b["grad"].data_ptr(), b["dets"].data_ptr(), b["cnt"].data_ptr(), None, None,
            log.ring.data_ptr(), log.capacity, log.cursor.data_ptr(), self.wsp.data_ptr(),
            self.wsp.numel(), self.phase, 0, conf, iou, b["dets"].shape[1], self.batch, s, bx, c, img,
            5.0, 0.5, self.inv_gb, stream.cuda_stream))


def time_launches(torch, fn, nsets, iters, stream):
    """Average duration of one launch: HIP events recorded on the launch stream around
    `iters` consecutive launches (launch i works on buffer set i % nsets)."""
    with torch.cuda.stream(stream):
        for i in range(max(32, 2 * nsets)):
            fn(i % nsets)
        stream.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for i in range(iters):
            fn(i % nsets)
        b.record(stream)
        stream.synchronize()
    return 1e3 * a.elapsed_time(b) / iters


_bench_lib = None


def bench_lib():
    """libyhbench.so: the hand-written streaming copy / read the kernels are compared with."""
    global _bench_lib
    if _bench_lib is None:
        import ctypes
        lib = ctypes.CDLL(os.path.join(ROOT, "yolo_amd", "lib", "libyhbench.so"))
        lib.yhb_stream_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.yhb_stream_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                        ctypes.c_void_p]
        lib.yhb_stream_rw.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _bench_lib = lib
    return _bench_lib


def copy_ceiling(torch, dev, stream):
    """What streaming reaches on this part, far beyond the memory-side cache: a hand-written
    copy (16 bytes per lane, grid-stride, two loads in flight, non-temporal loads and stores,
    16,384 workgroups) of 1 GiB -> 1 GiB, the read of 1 GiB alone, and torch's copy_ of the
    same buffers for comparison."""
    lib = bench_lib()
    n = (1 << 30) // 4
    x = torch.empty(n, device=dev).normal_()
    y = torch.empty_like(x)
    sink = torch.zeros(1, device=dev)
    st = stream.cuda_stream

    def timed(fn, reps=10):
        with torch.cuda.stream(stream):
            for _ in range(3):
                fn()
            stream.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(reps):
                fn()
            b.record(stream)
            stream.synchronize()
        return 1e3 * a.elapsed_time(b) / reps

    def hand_copy():
        rc = lib.yhb_stream_copy(y.data_ptr(), x.data_ptr(), n * 4, 16384, 2, 3, st)
        if rc:
            raise SystemExit(f"yhb_stream_copy failed: {rc}")

    us = timed(hand_copy)
    if not torch.equal(x[:4096], y[:4096]) or not torch.equal(x[-4096:], y[-4096:]):
        raise SystemExit("the streaming copy did not copy")
    us_read = timed(lambda: lib.yhb_stream_read(x.data_ptr(), sink.data_ptr(), n * 4, 4096, st))
    us_torch = timed(lambda: y.copy_(x))
    del x, y
    return {"gbs": 2 * (1 << 30) / (us * 1e-6) / 1e9, "bytes_per_launch": 2 * (1 << 30), "us": us,
            "kernel": "libyhbench stream_copy_kernel<2, nt loads, nt stores>, 16,384 workgroups",
            "read_only_gbs": (1 << 30) / (us_read * 1e-6) / 1e9,
            "torch_copy_gbs": 2 * (1 << 30) / (us_torch * 1e-6) / 1e9}


def streaming_floor(torch, dev, stream, rbytes, wbytes, iters=200):
    """Duration of a plain streaming kernel with a given traffic -- rbytes read, wbytes written,
    16 bytes per lane, non-temporal -- launched back to back over a rotation of buffers that
    exceeds twice the memory-side cache: what the memory system alone allows a one-launch
    kernel of that size (ramp and drain included).  Microseconds per launch."""
    lib = bench_lib()
    rb, wb = (int(rbytes) + 15) & ~15, (int(wbytes) + 15) & ~15
    wb = min(wb, rb)
    nsets = -(-2 * MALL_BYTES // min(rb, max(wb, 1 << 20))) + 1
    nsets = max(2, min(nsets, 128))
    src = [torch.empty(rb // 4, device=dev).normal_() for _ in range(nsets)]
    dst = [torch.empty(max(wb, 16) // 4, device=dev) for _ in range(nsets)]
    sink = torch.zeros(1, device=dev)
    st = stream.cuda_stream

    def run(i):
        rc = lib.yhb_stream_rw(src[i].data_ptr(), rb, dst[i].data_ptr(), wb, sink.data_ptr(), 4096, st)
        if rc:
            raise SystemExit(f"yhb_stream_rw failed: {rc}")

    us = time_launches(torch, run, nsets, iters, stream)
    del src, dst
    return us


def rotation_sets(args, batch):
    """Buffer sets in the rotation: each kernel's own bytes (the smaller, decode+NMS, decides)
    must add up to twice the memory-side cache."""
    if args.cold_sets > 1:
        return args.cold_sets
    return -(-2 * MALL_BYTES // (nms_bytes_per_image() * batch)) + 1


def time_step_graph(torch, dev, K, sets, klog, fused, serial=False, reps=8):
    """The whole step as the timed region runs it (one graph of `n` steps: a multiple of the
    number of buffer sets, at least 64 steps so that the graph's start-up and join weigh
    little), HIP-event timed over `reps` replays.  Returns microseconds per step."""
    s1 = torch.cuda.Stream(dev)
    s2 = s1 if (serial or fused) else torch.cuda.Stream(dev)
    n = len(sets) * max(2, -(-64 // len(sets)))

    def enqueue():
        if fused:
            for i in range(n):
                K.fused(sets[i % len(sets)], klog, s1)
            K.loss_flush(klog, s1)
            return
        if s2 is not s1:
            s2.wait_stream(s1)
        for i in range(n):
            K.loss(sets[i % len(sets)], klog, s1)
        K.loss_flush(klog, s1)
        for i in range(n):
            K.nms(sets[i % len(sets)], s2)
        if s2 is not s1:
            s1.wait_stream(s2)

    with torch.cuda.stream(s1):
        enqueue()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1, capture_error_mode="thread_local"):
        enqueue()
    with torch.cuda.stream(s1):
        for _ in range(3):
            g.replay()
        s1.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s1)
        for _ in range(reps):
            g.replay()
        b.record(s1)
        s1.synchronize()
    del g
    return 1e3 * a.elapsed_time(b) / (reps * n)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist_on = world > 1 or args.force_dist     # every collective branch below keys on this

    import torch
    import torch.distributed as dist
    from yolo_amd import LossLog, synth
    from yolo_amd.dist import LossAllReducer, shard_range

    # rehearsal aid: BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0 (with --backend gloo),
    # so the multi-process path can be exercised on a one-GPU machine
    dev_index = 0 if os.environ.get("BENCH_SINGLE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:         # --force-dist without a launcher
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # see self_launch
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    def say(msg):
        if args.verbose:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    scaling = args.scaling
    if scaling == "auto":
        scaling = "strong" if world > 1 else "weak"
    if scaling == "strong":
        global_batch = args.global_batch
        lo, hi = shard_range(global_batch, rank, world)
        batch = hi - lo
    else:
        batch = args.batch
        global_batch = batch * world
    say(f"process group ready, world {world}, device {dev}, {batch} of {global_batch} images")

    pipelined = args.loss_path == "pipelined"
    use_graph = not args.no_graph
    K = Kernels(dev, batch, global_batch, pipelined)
    if args.step == "fused" and not K.has_fused:
        raise SystemExit("--step fused: the library has no yh_head_step_pipelined")
    fused = K.has_fused and pipelined and args.step in ("auto", "fused")
    base = synth.make_inputs(batch, S, BX, C, IMG, seed=2 * rank)
    resident = K.make_set(base)
    nsets = rotation_sets(args, batch) if args.regime == "hbm" else 1
    # every set holds the same batch (the loss rows of all steps must agree), in memory of its own
    sets = [resident] + [K.make_set(base) for _ in range(nsets - 1)]
    s_loss = torch.cuda.Stream(dev)
    s_nms = s_loss if (args.serial or fused) else torch.cuda.Stream(dev)
    every = max(1, min(args.allreduce_every, 4096))
    # the ring holds every step of the timed region when that is at most 4096 rows: the
    # all-reduce then runs in place on the rows of a chunk, without a copy
    in_place = args.steps <= 4096
    log = LossLog(dev, capacity=max(args.steps, every, 1) if in_place else every)
    cursor = [0]        # buffer set of the next step: the rotation carries on across chunks

    def enqueue_chunk(n, first):
        """n steps on buffer sets first, first + 1, ...: n loss passes (+ the flush of the
        pipelined path) on s_loss, n decode+NMS passes on s_nms, forked from and joined back
        into s_loss; or n launches of the fused step."""
        pick = lambda i: sets[(first + i) % len(sets)]
        if fused:
            for i in range(n):
                K.fused(pick(i), log, s_loss)
            K.loss_flush(log, s_loss)
            return
        if s_nms is not s_loss:
            s_nms.wait_stream(s_loss)
        for i in range(n):
            K.loss(pick(i), log, s_loss)
        K.loss_flush(log, s_loss)
        for i in range(n):
            K.nms(pick(i), s_nms)
        if s_nms is not s_loss:
            s_loss.wait_stream(s_nms)

    with torch.cuda.stream(s_loss):
        enqueue_chunk(4, 0)
    torch.cuda.synchronize()
    graphs = {}

    def chunk_graph(n, first):
        g = graphs.get((n, first))
        if g is None:
            g = torch.cuda.CUDAGraph()
            # thread_local: the process group's watchdog thread may touch the HIP
            # runtime while this thread captures
            with torch.cuda.graph(g, stream=s_loss, capture_error_mode="thread_local"):
                enqueue_chunk(n, first)
            graphs[(n, first)] = g
        return g

    def run_chunk(n):
        first = cursor[0] % len(sets)
        cursor[0] += n
        with torch.cuda.stream(s_loss):
            if use_graph:
                chunk_graph(n, first).replay()
            else:
                enqueue_chunk(n, first)

    def chunks_of(total):
        out, left = [], total
        while left > 0:
            out.append(min(every, left))
            left -= out[-1]
        return out

    def sync_all():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()

    def timed_steps(total, reducer):
        """Enqueue `total` steps chunk by chunk; every chunk's loss rows are all-reduced
        (stream-ordered after the chunk) while the next chunk runs."""
        done = 0
        for n in chunks_of(total):
            row0 = int(done % log.capacity) if in_place else 0
            run_chunk(n)
            with torch.cuda.stream(s_loss):      # ordered after the kernels that filled the rows
                if in_place:
                    reducer.flush_rows(log.ring[row0:row0 + n])
                else:
                    reducer.flush(log.ring, n)       # chunks are ring-aligned: rows 0 .. n-1
            done += n
        with torch.cuda.stream(s_loss):
            t_wait = time.perf_counter()
            out = reducer.results()              # waits for the collectives (stream-ordered)
            return out, time.perf_counter() - t_wait

    result = None
    if args.measure in ("all", "step"):
        # untimed: bring the clocks up, then the W warm-up steps, then one rehearsal of the
        # timed region's bookkeeping (the first use of each piece loads code on the host and,
        # under a graph, captures the chunk graphs the timed region replays)
        def dry_run(total):
            cursor[0] = 0
            timed_steps(total, LossAllReducer(always=dist_on))
            torch.cuda.synchronize()
            log.reset()

        try:
            dry_run(args.steps)
        except Exception as exc:  # capture unsupported: fall back to eager launches
            if not use_graph:
                raise
            if rank == 0:
                print(f"[bench] graph capture failed ({exc!r}); running eagerly", file=sys.stderr)
            graphs.clear()
            use_graph = False
            torch.cuda.synchronize()
            dry_run(args.steps)
        say(f"step built ({'one graph per chunk' if use_graph else 'eager'}, "
            f"{'fused kernel' if fused else 'two kernels'}, {len(sets)} buffer sets)")
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_seconds:
            cursor[0] = 0                        # the very graph the timed region starts with
            run_chunk(chunks_of(args.steps)[0])
            torch.cuda.synchronize()
            log.reset()
        for n in chunks_of(args.warmup):
            run_chunk(n)
        torch.cuda.synchronize()
        log.reset()
        cursor[0] = 0

        # ---- timed region: exactly K steps --------------------------------------
        say("warm-up done")
        reducer = LossAllReducer(always=dist_on)
        sync_all()
        t0 = time.perf_counter()
        losses, waited = timed_steps(args.steps, reducer)
        torch.cuda.synchronize()
        own_elapsed = time.perf_counter() - t0
        if dist_on:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64,
                          device=dev if args.backend == "nccl" else "cpu")
        per_rank = None
        if dist_on:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            mine = torch.tensor([own_elapsed, waited], dtype=torch.float64, device=el.device)
            gathered = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            per_rank = torch.stack(gathered).cpu()
        elapsed = float(el.item())
        say(f"timed region done: {elapsed:.6f} s")

        assert losses.shape == (args.steps, 4), losses.shape
        loss_value = float(losses[-1, 3].item()) / global_batch
        if not torch.isfinite(losses).all():
            raise SystemExit("non-finite loss in the timed region")
        # Every step saw the same batch (each buffer set holds a copy): a row that differs was
        # lost or logged twice.  One rank: the rows are the same bits.  Several: a ring
        # all-reduce adds the ranks in an order that depends on where in the buffer an element
        # sits, so the rows agree to rounding.
        first = losses[:1].expand_as(losses)
        same = torch.equal(losses, first) if world == 1 else torch.allclose(losses, first, rtol=1e-5, atol=0)
        if not same:
            raise SystemExit("the per-step loss log is not the same row for every step")
        if args.dump_log and rank == 0:
            import numpy as np
            np.save(args.dump_log, losses.cpu().numpy())

        value = args.steps * global_batch / elapsed
        regime = ("buffers rotate past the memory-side cache (HBM proper): %d sets, %.0f MB" % (
                      len(sets), len(sets) * sum(t.numel() * t.element_size() for t in resident.values()) / 1e6)
                  if len(sets) > 1 else "one resident buffer set (memory-side cache, Infinity Cache, serves it in part)")
        result = {
            "metric": "YOLOv1 head images/sec (loss fwd+bwd + NMS)",
            "value": value,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"YOLOv1 head S={S},B={BX},C={C} batch {global_batch} sharded by image over "
                             f"{world} GPUs ({batch} per GPU) fwd+bwd + NMS (fp32), loss divided by the "
                             f"global batch, RCCL all-reduce of the loss sums" if world > 1 else
                             f"YOLOv1 head S={S},B={BX},C={C} batch {batch} fwd+bwd + NMS on 1 GPU (fp32)"),
                "regime": regime,
                "global_batch": global_batch,
                "images_per_gpu": batch,
                "conf_thresh": CONF_T, "iou_thresh": IOU_T, "nms_mode": "reference",
                "detection_rows_per_image": K.rows(0),
                "parallelism": f"dp{world} by image, loss all-reduce every {every} steps",
                "world_size": dist.get_world_size() if dist_on else 1,
                "backend": (dist.get_backend() if dist_on else None),
                "launch": (f"one HIP graph per chunk of <= {every} steps" +
                           ("" if fused else ", both streams inside (fork/join)") if use_graph else "eager"),
                "step": ("one launch per step: loss fwd+bwd and decode+NMS read the head tensor once "
                         "(yh_head_step_pipelined), loss sums logged on the device one step late") if fused
                        else ("loss kernel and decode+NMS kernel, " +
                              ("one launch each per step, loss sums logged on the device one step late "
                               "(yh_loss_fwd_bwd_pipelined)" if pipelined
                               else "gradient kernel + finalize kernel per step")),
                "streams": 1 if (args.serial or fused) else 2,
            },
            "loss": loss_value,
            "step_gbs": value * (loss_bytes_per_image() + nms_bytes_per_image()) / 1e9,
        }
        if len(sets) > 1:
            result["value_hbm_proper"] = value
        if per_rank is not None:
            ms = per_rank[:, 0] * 1e3 / args.steps
            result["ranks"] = {"ms_per_step_min": float(ms.min()), "ms_per_step_max": float(ms.max()),
                               "results_wait_ms_max": float(per_rank[:, 1].max() * 1e3),
                               "note": "per rank, its own clock from the common start to its own "
                                       "synchronize; results_wait = host time inside LossAllReducer.results()"}
    elif rank == 0:
        result = {"metric": "YOLOv1 head kernel timing only", "measure": args.measure, "n_gpus": world}

    # the collective part of the run is over: every rank leaves the group now, and only
    # then does rank 0 go on to time single kernels (nobody waits in a barrier for that)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    for g in list(graphs.values()):
        del g
    graphs.clear()

    if rank == 0 and args.measure != "step":
        measure_kernels(args, torch, dev, K, sets, log, result, batch, rank, fused)
        if world == 1 and not args.no_cpu_baseline and args.measure == "all":
            result["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if rank == 0:
        print(json.dumps(result), flush=True)


def measure_kernels(args, torch, dev, K, sets, log, result, batch, rank, fused):
    """Dominant kernel -> roofline, from HIP events on the launch stream.  `frac` is the
    HBM-proper figure: the kernel over a rotation of buffer sets whose own bytes exceed
    twice the memory-side cache, so no launch finds its data there."""
    from yolo_amd import LossLog, synth
    st = torch.cuda.Stream(dev)
    klog = LossLog(dev, capacity=64)
    iters = args.kernel_iters
    algo = loss_bytes_per_image() * batch
    nms_algo = nms_bytes_per_image() * batch
    nms_rows_algo = nms_bytes_per_image(rows=K.rows(0)) * batch
    resident = sets[0]
    res, rot = None, None
    if args.measure in ("all", "resident"):
        res = {"loss_fwd_bwd_us": time_launches(torch, lambda i: K.loss(resident, klog, st), 1, iters, st),
               "decode_nms_us": time_launches(torch, lambda i: K.nms(resident, st), 1, iters, st)}
        K.loss_flush(klog, st)
        if K.has_fused:
            res["head_step_us"] = time_launches(torch, lambda i: K.fused(resident, klog, st), 1, iters, st)
            K.loss_flush(klog, st)
    if args.measure in ("all", "rotated"):
        nsets = rotation_sets(args, batch)
        if len(sets) < nsets:
            base = synth.make_inputs(batch, S, BX, C, IMG, seed=2 * rank)
            sets = sets + [K.make_set(base) for _ in range(nsets - len(sets))]
        set_bytes = sum(t.numel() * t.element_size() for t in sets[0].values())
        rot = {"buffer_sets": nsets, "total_mb": set_bytes * nsets / 1e6,
               "loss_bytes_in_rotation_mb": algo * nsets / 1e6,
               "decode_nms_bytes_in_rotation_mb": nms_algo * nsets / 1e6,
               "loss_fwd_bwd_us": time_launches(torch, lambda i: K.loss(sets[i], klog, st), nsets, iters, st),
               "decode_nms_us": time_launches(torch, lambda i: K.nms(sets[i], st), nsets, iters, st)}
        K.loss_flush(klog, st)
        if K.has_fused:
            rot["head_step_us"] = time_launches(torch, lambda i: K.fused(sets[i], klog, st), nsets, iters, st)
            K.loss_flush(klog, st)
        if args.measure == "all" and result.get("value") is not None:
            # the whole step over the same rotation (one graph), both ways of running it
            rot["step_us"] = time_step_graph(torch, dev, K, sets, klog, fused, args.serial)
            rot["two_stream_step_us"] = (rot["step_us"] if not fused else
                                         time_step_graph(torch, dev, K, sets, klog, False))
            if K.has_fused:
                rot["fused_step_us"] = (rot["step_us"] if fused else
                                        time_step_graph(torch, dev, K, sets, klog, True))
    torch.cuda.synchronize()
    if args.measure == "all" and result.get("value") is not None:
        # the same step on ONE buffer set: what the memory-side cache makes of it
        us = time_step_graph(torch, dev, K, [resident], klog, fused, args.serial)
        result["cache_resident"] = {
            "step_us": us, "images_per_s": batch / (us * 1e-6),
            "note": "one resident buffer set, HIP-event timed graph replays; the 256 MiB memory-side "
                    "cache serves part of the step's bytes: NOT an HBM figure"}

    traffic, nms_traffic, step_traffic, traffic_note = None, None, None, "no profiles/traffic.json"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("batch") != batch:
                traffic_note = f"profiles/traffic.json was measured at batch {tj.get('batch')}"
            elif tj.get("sources_sha") != kernel_sources_sha():
                traffic_note = "profiles/traffic.json was measured on other kernel sources: not reported"
            else:
                traffic = tj.get("loss_kernel_hbm_bytes_per_launch")
                nms_traffic = tj.get("nms_kernel_hbm_bytes_per_launch")
                step_traffic = tj.get("head_step_kernel_hbm_bytes_per_launch")
                traffic_note = f"PMC FETCH_SIZE/WRITE_SIZE passes of profile {tj.get('profile')} (rotated buffers)"
        except Exception as exc:
            traffic_note = f"profiles/traffic.json unreadable: {exc!r}"

    def gbs(nbytes, us):
        return nbytes / (us * 1e-6) / 1e9

    # The dominant kernel of the step as it ran: the one-kernel step when the step is fused
    # (it is the whole step), else the loss kernel.  Its algorithmic bytes: the per-image
    # figures of loss_bytes_per_image / nms_bytes_per_image, the head tensor counted once
    # for the kernel that reads it once.
    hs_algo = algo + nms_algo - 4 * (5 * BX + C) * S * S * batch
    src = rot or res
    if fused and src.get("head_step_us"):
        main_us, main_algo, main_traffic = src["head_step_us"], hs_algo, step_traffic
        main_name = "head_step_kernel<%d,%d,%d,true>" % (S, BX, C)
        main_note = ("loss fwd+bwd and decode+NMS in one launch; algorithmic bytes = head tensor read ONCE + "
                     "targets read + gradient written + (batch, S*S*Bx, 6) detections + counts; the bytes it "
                     "moves are fewer still: targets of cells without an object are never read and "
                     "yh_decode_nms_rows-style S*S rows are written")
    else:
        main_us, main_algo, main_traffic = src["loss_fwd_bwd_us"], algo, traffic
        main_name = "loss_fwd_bwd_kernel<%d,%d,%d,%s,false>" % (S, BX, C, "true" if K.pipelined else "false")
        main_note = ("the kernel never reads the targets of cells without an object, so the bytes it moves "
                     "(PMC) are below the algorithmic count")
    achieved = gbs(main_algo, main_us)
    roof = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": main_traffic, "traffic_source": traffic_note,
        "achieved_counter_bytes": gbs(main_traffic, main_us) if main_traffic else None,
        "frac_counter_bytes": gbs(main_traffic, main_us) / HBM_PEAK_GBS if main_traffic else None,
        "counter_bytes_note": main_note + "; `frac` prices the algorithmic bytes, `frac_counter_bytes` the bytes moved",
        "kernel": main_name,
        "regime": "buffers rotate past the memory-side cache (HBM proper)" if rot
                  else "one resident buffer set (memory-side cache)",
        "algorithmic_bytes_per_launch": main_algo,
        "kernel_us": main_us,
    }
    if fused and src.get("head_step_us"):
        roof["two_kernel_algorithmic_bytes"] = algo + nms_algo
        roof["frac_on_two_kernel_bytes"] = gbs(algo + nms_algo, main_us) / HBM_PEAK_GBS
    if rot:
        rot["loss_fwd_bwd_frac"] = gbs(algo, rot["loss_fwd_bwd_us"]) / HBM_PEAK_GBS
        rot["decode_nms_frac"] = gbs(nms_algo, rot["decode_nms_us"]) / HBM_PEAK_GBS
        if rot.get("head_step_us"):
            rot["head_step_frac"] = gbs(hs_algo, rot["head_step_us"]) / HBM_PEAK_GBS
        roof["beyond_memory_side_cache"] = rot
    if res:
        res["loss_fwd_bwd_frac"] = gbs(algo, res["loss_fwd_bwd_us"]) / HBM_PEAK_GBS
        res["decode_nms_frac"] = gbs(nms_algo, res["decode_nms_us"]) / HBM_PEAK_GBS
        if res.get("head_step_us"):
            res["head_step_frac"] = gbs(hs_algo, res["head_step_us"]) / HBM_PEAK_GBS
        res["note"] = "cache-assisted: a fraction above what HBM delivers is the memory-side cache, not HBM"
        roof["infinity_cache_resident"] = res
    loss_us = src["loss_fwd_bwd_us"]
    roof["loss_fwd_bwd"] = {
        "kernel": "loss_fwd_bwd_kernel<%d,%d,%d,%s,false>" % (S, BX, C, "true" if K.pipelined else "false"),
        "algorithmic_bytes_per_launch": algo, "kernel_us": loss_us, "frac": gbs(algo, loss_us) / HBM_PEAK_GBS,
        "traffic": traffic, "frac_counter_bytes": gbs(traffic, loss_us) / HBM_PEAK_GBS if traffic else None}
    nms_us = src["decode_nms_us"]
    roof["decode_nms"] = {
        "kernel": "nms_kernel<64,1,0,%d,%d,%d>" % (S, BX, C), "algorithmic_bytes_per_launch": nms_algo,
        "kernel_us": nms_us, "frac": gbs(nms_algo, nms_us) / HBM_PEAK_GBS,
        "rows_written_per_image": K.rows(0),
        "bytes_with_rows_written": nms_rows_algo,
        "frac_rows_written": gbs(nms_rows_algo, nms_us) / HBM_PEAK_GBS,
        "traffic": nms_traffic,
        "frac_counter_bytes": gbs(nms_traffic, nms_us) / HBM_PEAK_GBS if nms_traffic else None,
        "note": "`frac` prices the algorithmic bytes of a (batch, S*S*Bx, 6) detection tensor; reference "
                "mode has at most S*S candidates and yh_decode_nms_rows writes that many rows per "
                "image, which `frac_rows_written` prices"}
    if args.measure == "all":
        cc = copy_ceiling(torch, dev, st)
        roof["copy_ceiling"] = cc
        roof["frac_of_copy_ceiling"] = achieved / cc["gbs"]
        if main_traffic:
            roof["frac_counter_bytes_of_copy_ceiling"] = gbs(main_traffic, main_us) / cc["gbs"]
        # what a plain streaming kernel needs for each kernel's traffic, one launch at a time over
        # rotating buffers: the bytes the PMC counters saw when known, else the algorithmic bytes
        d4, cells = 4 * (5 * BX + C) * S * S * batch, S * S * batch
        shapes = {"loss_fwd_bwd": (d4 + cells * (4 * (4 + C) + 1), d4),
                  "decode_nms": (d4, (24 * K.rows(0) + 4) * batch),
                  "head_step": (d4 + cells * (4 * (4 + C) + 1), d4 + (24 * K.rows(0) + 4) * batch)}
        basis = "algorithmic bytes (read, written)"
        tpath2 = os.path.join(ROOT, "profiles", "traffic.json")
        if main_traffic and os.path.exists(tpath2):
            tj = json.load(open(tpath2))
            for key, name in (("loss_fwd_bwd", "loss_kernel"), ("decode_nms", "nms_kernel"), ("head_step", "head_step_kernel")):
                if tj.get(name + "_fetch_size_kib_raw"):
                    shapes[key] = (int(2048 * tj[name + "_fetch_size_kib_raw"]), int(1024 * tj[name + "_write_size_kib"]))
            basis = "bytes the PMC counters saw (FETCH_SIZE doubled, WRITE_SIZE)"
        floor = {"basis": basis, "kernel": "libyhbench stream_rw_kernel<2>, 4,096 workgroups, non-temporal, "
                                           "back to back over rotating buffers"}
        for key, us_key in (("loss_fwd_bwd", "loss_fwd_bwd_us"), ("decode_nms", "decode_nms_us"), ("head_step", "head_step_us")):
            if not src.get(us_key):
                continue
            rb, wb = shapes[key]
            fus = streaming_floor(torch, dev, st, rb, wb)
            floor[key] = {"read_bytes": rb, "written_bytes": wb, "streaming_kernel_us": fus,
                          "kernel_us": src[us_key], "kernel_over_streaming": src[us_key] / fus}
        roof["streaming_floor"] = floor
        if roof["frac_of_copy_ceiling"] > 1.0:
            roof["frac_of_copy_ceiling_note"] = ("above 1 because the algorithmic bytes include target rows "
                                                 "the kernel never reads; see frac_counter_bytes_of_copy_ceiling")
        if rot and rot.get("step_us"):
            moved = (step_traffic if fused else ((traffic + nms_traffic) if traffic and nms_traffic else None))
            if moved:
                # the whole step over rotating buffers: the HBM bytes it moves by the PMC counters /
                # its duration, against the streaming-copy ceiling
                rot["step_hbm_gbs"] = gbs(moved, rot["step_us"])
                rot["step_frac_of_copy_ceiling"] = rot["step_hbm_gbs"] / cc["gbs"]
                rot["step_frac_counter_bytes"] = rot["step_hbm_gbs"] / HBM_PEAK_GBS
    result["roofline"] = roof
    configs = {}
    if args.measure == "all" and not args.no_dp_table and result.get("value") is not None:
        del sets
        configs["dp_strong_n1"] = dp_strong_table(args, torch, dev, K.pipelined, fused)
    if args.measure == "all" and not args.no_stress:
        configs["nms_stress"] = stress_record(args, torch, dev, st)
    if configs:
        result["configs"] = configs


def dp_strong_table(args, torch, dev, pipelined, fused):
    """The denominators of the strong-scaling curve: the step of BASELINE config 3 (global
    batch 65,536, loss divided by it) with all, a half, a quarter and an eighth of the images
    on ONE GPU -- what each GPU of an N = 1, 2, 4, 8 run executes per step, without any
    exchange.  HIP-event timed graph replays over a rotation of buffer sets (at these sizes
    one set exceeds the memory-side cache except at 8,192, so the rotation matters only there).
    Parallel efficiency at N GPUs = T(65,536) / (N * T_N measured on N GPUs); T(65,536/N)
    here is what a perfect exchange would leave."""
    from yolo_amd import LossLog, synth
    out = {"global_batch": STRONG_GLOBAL_BATCH, "regime": "rotating buffer sets (HBM proper)", "rows": []}
    klog = LossLog(dev, capacity=64)
    for n_gpus in (1, 2, 4, 8):
        b = STRONG_GLOBAL_BATCH // n_gpus
        K = Kernels(dev, b, STRONG_GLOBAL_BATCH, pipelined)
        base = synth.make_inputs(b, S, BX, C, IMG, seed=0)
        nsets = max(2, rotation_sets(args, b))
        sets = [K.make_set(base) for _ in range(nsets)]
        us = time_step_graph(torch, dev, K, sets, klog, fused and K.has_fused, args.serial, reps=4)
        out["rows"].append({"as_on_n_gpus": n_gpus, "images_on_this_gpu": b, "buffer_sets": nsets,
                            "step_us": us, "images_per_s_per_gpu": b / (us * 1e-6)})
        del sets, K
        torch.cuda.empty_cache()
    t1 = out["rows"][0]["step_us"]
    for row in out["rows"]:
        row["ideal_speedup"] = row["as_on_n_gpus"]
        row["speedup_of_kernels_alone"] = t1 / row["step_us"]
    return out


def stress_record(args, torch, dev, st):
    """BASELINE config 4: S=14, B=5, C=80 (980 candidates per image), conf 0.001, dense
    candidates, batch 1024 -- decode+NMS and the run-time-shape loss kernel, resident
    and over rotating buffers."""
    from yolo_amd import LossLog, synth
    shape = (14, 5, 80, 448)
    s, bx, c, img = shape
    b = 1024
    K = Kernels(dev, b, b, False, shape)
    base = synth.make_inputs(b, s, bx, c, img, seed=0)
    nms_algo = nms_bytes_per_image(s, bx, c) * b
    loss_algo = loss_bytes_per_image(s, bx, c) * b
    nsets = -(-2 * MALL_BYTES // nms_algo) + 1
    sets = [K.make_set(base, mode=1) for _ in range(nsets)]
    klog = LossLog(dev, capacity=64)
    iters = max(20, args.kernel_iters // 8)
    out = {"workload": f"S={s},B={bx},C={c} batch {b}, conf_thresh 0.001, iou_thresh 0.5, dense candidates",
           "buffer_sets": nsets}
    for name, nset in (("resident", 1), ("rotated", nsets)):
        t_nms = time_launches(torch, lambda i: K.nms(sets[i], st, mode=1, conf=0.001), nset, iters, st)
        t_loss = time_launches(torch, lambda i: K.loss(sets[i], klog, st), nset, iters, st)
        out[name] = {
            "decode_nms_us": t_nms, "decode_nms_frac": nms_algo / (t_nms * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "loss_us": t_loss, "loss_frac": loss_algo / (t_loss * 1e-6) / 1e9 / HBM_PEAK_GBS}
    torch.cuda.synchronize()
    out["decode_nms_algorithmic_bytes_per_launch"] = nms_algo
    out["loss_algorithmic_bytes_per_launch"] = loss_algo
    out["loss_note"] = ("two launches (gradient kernel + finalize); run-time shape, the 82 KB image block leaves "
                        "through a rolling LDS window in non-temporal stores (loss_fwd_bwd_roll_kernel); the "
                        "kernel never reads the targets of cells without an object, so loss_frac prices more "
                        "bytes than it moves")
    out["kept_per_image"] = float(sets[0]["cnt"].float().mean().item())
    # The synthetic classes are uniform (12 candidates per class on average).  Detections of a
    # real image crowd into few classes: with one class winning 30 % of the cells a class
    # segment has some 300 members, and the kernel takes a round per block of 64 of them.
    crowd = sets[0]
    gen = torch.Generator(device=dev).manual_seed(1)
    mask = torch.rand(b, s, s, device=dev, generator=gen) < 0.3
    crowd["pred"][:, 5 * bx + c // 2][mask] += 30.0
    t_crowd = time_launches(torch, lambda i: K.nms(crowd, st, mode=1, conf=0.001), 1, iters, st)
    torch.cuda.synchronize()
    out["one_class_in_30pct_of_cells"] = {"decode_nms_us": t_crowd,
                                          "kept_per_image": float(crowd["cnt"].float().mean().item())}
    return out


if __name__ == "__main__":
    main()
