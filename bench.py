#!/usr/bin/env python3
"""Headline benchmark: YOLOv1 head images/sec (loss fwd+bwd + decode/NMS).

One "step" = one pass of the hot path over one batch that is already resident
in HBM: the fused loss forward+backward (gradient w.r.t. the head tensor and
the loss sums) and decode + confidence threshold + per-class NMS of the same
head tensor.  The two halves are independent readers of the head tensor; they
run on two HIP streams inside ONE captured graph per chunk of steps.

Workloads (S=7, Bx=2, C=20, fp32, synthetic logits and targets, yolo_amd.synth):
  --gpus 1   batch 4096 on one GPU.
  --gpus N   batch 65,536 sharded by image over the N GPUs (strong scaling: N=2/4/8
             hold 32,768/16,384/8,192 images each); the loss is divided by the
             GLOBAL batch (model.py:317) and the only exchange is the all-reduce of
             the per-step loss sums, batched over `--allreduce-every` steps and
             completed inside the timed region.  `--scaling weak` keeps --batch
             images per GPU instead.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment the command starts its
own `python -m torch.distributed.run` with N ranks (before anything touches a
GPU) and relays rank 0's line.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the
fused loss kernel) over a rotation of buffer sets that exceeds the 256 MiB
memory-side cache, i.e. against HBM proper: algorithmic bytes per launch / its
HIP-event duration, against the 8 TB/s HBM3E peak; the cache-resident figure,
the decode+NMS kernel and a measured device-copy ceiling sit beside it.
`cpu_baseline` is the CPU oracle (a port of the reference's formulas,
oracle/ref_cpu.py) timed on this host on bounded samples.
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
KERNEL_SOURCES = ("head_math.hpp", "loss_kernels.hpp", "nms_kernels.hpp", "yolo_head.hip")


def loss_bytes_per_image(s=S, bx=BX, c=C):
    """Algorithmic bytes of the fused loss kernel per image: head tensor read +
    targets read (xywh, class probabilities, object mask) + gradient written."""
    d, n = 5 * bx + c, s * s
    return 4 * d * n + (n * (4 + c) * 4 + n) + 4 * d * n


def nms_bytes_per_image(s=S, bx=BX, c=C):
    """Algorithmic bytes of decode+NMS per image: head tensor read + detections
    written (S*S*Bx rows of 6 floats) + keep count."""
    d, n = 5 * bx + c, s * s
    return 4 * d * n + 24 * n * bx + 4


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
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the whole CPU leg")
    ap.add_argument("--kernel-iters", type=int, default=320)
    ap.add_argument("--cold-sets", type=int, default=0,
                    help="buffer sets rotated through for the HBM-proper figures (0: as many as put "
                         "twice the memory-side cache behind every kernel's own bytes)")
    ap.add_argument("--measure", choices=("all", "step", "rotated", "resident"), default="all",
                    help="step: timed region only; rotated / resident: only that kernel-timing leg "
                         "(what the rocprofv3 passes under profiles/ run)")
    ap.add_argument("--no-stress", action="store_true", help="skip the S=14,B=5,C=80 sub-record")
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
    sample of the batch-4096 workload."""
    import torch
    from oracle import ref_cpu
    from yolo_amd import synth

    all_cores = max(1, min(16, os.cpu_count() or 1))
    pred, gx, gc, om = synth.make_inputs(4096, S, BX, C, IMG, seed=0)
    ref_cpu.loss_and_grad(pred[:16], gx[:16], gc[:16], om[:16])        # first-use costs
    ref_cpu.postprocess(pred[:16], CONF_T, IOU_T, mode=0)
    legs = (("batch16", 16, 1), ("batch16", 16, all_cores), ("batch4096", 512, 1),
            ("batch4096", 4096, all_cores))
    share = seconds / len(legs)
    detail = {}
    for name, n, threads in legs:
        torch.set_num_threads(threads)
        p, x, c, m = pred[:n], gx[:n], gc[:n], om[:n]
        reps, t0 = 0, time.perf_counter()
        while True:
            ref_cpu.loss_and_grad(p, x, c, m)
            ref_cpu.postprocess(p, CONF_T, IOU_T, mode=0)
            reps += 1
            el = time.perf_counter() - t0
            if (el >= share and reps >= 2) or reps >= 200:
                break
        detail.setdefault(name, {})[f"threads_{threads}"] = {
            "images_per_s": reps * n / el, "images_per_pass": n, "passes": reps, "seconds": el}
    torch.set_num_threads(all_cores)
    top = detail["batch4096"][f"threads_{all_cores}"]
    return {
        "value": top["images_per_s"], "unit": "images/s", "cores": all_cores, "kind": "port",
        "sample": f"{top['passes']} passes over the {top['images_per_pass']} images of the workload: "
                  f"PyTorch fp32 CPU loss forward + autograd backward ({all_cores} threads) and "
                  f"single-threaded numpy decode+NMS, {top['seconds']:.1f} s; `detail` holds one thread "
                  f"and {all_cores} threads on config 1 (batch 16) and on the batch-4096 workload "
                  f"(one thread: its first 512 images)",
        "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "detail": detail,
    }


class Kernels:
    """The step's two launches straight through the C ABI on caller-owned buffers."""

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

    def make_set(self, inputs):
        torch = self.torch
        s, bx, c, _ = self.shape
        pred, gx, gc, om = (t.to(self.dev).clone() for t in inputs)
        return dict(pred=pred, gx=gx, gc=gc, om=om.view(torch.uint8), grad=torch.empty_like(pred),
                    dets=torch.empty(self.batch, s * s * bx, 6, device=self.dev),
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
        self._lib.check(self.lib.yh_decode_nms(
            b["pred"].data_ptr(), b["dets"].data_ptr(), b["cnt"].data_ptr(), None, mode, conf, iou,
            self.batch, s, bx, c, img, stream.cuda_stream))


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


def copy_ceiling(torch, dev, stream):
    """What a plain device copy reaches on this part: 1 GiB read + 1 GiB written per
    launch, far beyond the memory-side cache."""
    n = (1 << 30) // 4
    x = torch.empty(n, device=dev).normal_()
    y = torch.empty_like(x)
    with torch.cuda.stream(stream):
        for _ in range(3):
            y.copy_(x)
        stream.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(10):
            y.copy_(x)
        b.record(stream)
        stream.synchronize()
    us = 1e3 * a.elapsed_time(b) / 10
    del x, y
    return {"gbs": 2 * (1 << 30) / (us * 1e-6) / 1e9, "bytes_per_launch": 2 * (1 << 30), "us": us}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from yolo_amd import LossLog, synth
    from yolo_amd.dist import LossAllReducer, shard_range

    # rehearsal aid: BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0 (with --backend gloo),
    # so the multi-process path can be exercised on a one-GPU machine
    dev_index = 0 if os.environ.get("BENCH_SINGLE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

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
    resident = K.make_set(synth.make_inputs(batch, S, BX, C, IMG, seed=2 * rank))
    s_loss = torch.cuda.Stream(dev)
    s_nms = s_loss if args.serial else torch.cuda.Stream(dev)
    every = max(1, min(args.allreduce_every, 4096))
    # the ring holds every step of the timed region when that is at most 4096 rows: the
    # all-reduce then runs in place on the rows of a chunk, without a copy
    in_place = args.steps <= 4096
    log = LossLog(dev, capacity=max(args.steps, every, 1) if in_place else every)

    def enqueue_chunk(n, sets):
        """n steps: n loss passes (+ the flush of the pipelined path) on s_loss, n
        decode+NMS passes on s_nms, forked from and joined back into s_loss."""
        if s_nms is not s_loss:
            s_nms.wait_stream(s_loss)
        for i in range(n):
            K.loss(sets[i % len(sets)], log, s_loss)
        K.loss_flush(log, s_loss)
        for i in range(n):
            K.nms(sets[i % len(sets)], s_nms)
        if s_nms is not s_loss:
            s_loss.wait_stream(s_nms)

    with torch.cuda.stream(s_loss):
        enqueue_chunk(4, [resident])
    torch.cuda.synchronize()
    graphs = {}

    def chunk_graph(n, sets, key):
        g = graphs.get((key, n))
        if g is None:
            g = torch.cuda.CUDAGraph()
            # thread_local: the process group's watchdog thread may touch the HIP
            # runtime while this thread captures
            with torch.cuda.graph(g, stream=s_loss, capture_error_mode="thread_local"):
                enqueue_chunk(n, sets)
            graphs[(key, n)] = g
        return g

    def run_chunk(n, sets=None, key="resident"):
        sets = sets or [resident]
        with torch.cuda.stream(s_loss):
            if use_graph:
                chunk_graph(n, sets, key).replay()
            else:
                enqueue_chunk(n, sets)

    def chunks_of(total):
        out, left = [], total
        while left > 0:
            out.append(min(every, left))
            left -= out[-1]
        return out

    if use_graph:
        try:
            for n in sorted(set(chunks_of(args.steps) + chunks_of(args.warmup))):
                chunk_graph(n, [resident], "resident")
        except Exception as exc:  # capture unsupported: fall back to eager launches
            if rank == 0:
                print(f"[bench] graph capture failed ({exc!r}); running eagerly", file=sys.stderr)
            graphs.clear()
            use_graph = False
            torch.cuda.synchronize()

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed_steps(total, reducer, sets=None, key="resident"):
        """Enqueue `total` steps chunk by chunk; every chunk's loss rows are all-reduced
        (stream-ordered after the chunk) while the next chunk runs."""
        done = 0
        for n in chunks_of(total):
            row0 = int(done % log.capacity) if in_place else 0
            run_chunk(n, sets, key)
            with torch.cuda.stream(s_loss):      # ordered after the kernels that filled the rows
                if in_place:
                    reducer.flush_rows(log.ring[row0:row0 + n])
                else:
                    reducer.flush(log.ring, n)       # chunks are ring-aligned: rows 0 .. n-1
            done += n
        with torch.cuda.stream(s_loss):
            return reducer.results()             # waits for the collectives (stream-ordered)

    result = None
    if args.measure in ("all", "step"):
        say(f"step built ({'one graph per chunk' if use_graph else 'eager'})")
        # untimed: bring the clocks up, then the W warm-up steps, then one rehearsal of the
        # timed region's bookkeeping (the first use of each piece loads code on the host)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_seconds:
            run_chunk(chunks_of(args.steps)[0])
            torch.cuda.synchronize()
            log.reset()
        for n in chunks_of(args.warmup):
            run_chunk(n)
        torch.cuda.synchronize()
        log.reset()
        timed_steps(min(args.steps, 2 * every), LossAllReducer())
        torch.cuda.synchronize()
        log.reset()

        # ---- timed region: exactly K steps --------------------------------------
        say("warm-up done")
        reducer = LossAllReducer()
        sync_all()
        t0 = time.perf_counter()
        losses = timed_steps(args.steps, reducer)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64,
                          device=dev if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        say(f"timed region done: {elapsed:.6f} s")

        assert losses.shape == (args.steps, 4), losses.shape
        loss_value = float(losses[-1, 3].item()) / global_batch
        if not torch.isfinite(losses).all():
            raise SystemExit("non-finite loss in the timed region")
        # Every step saw the same resident batch: a row that differs was lost or logged twice.
        # One rank: the rows are the same bits.  Several: a ring all-reduce adds the ranks in an
        # order that depends on where in the buffer an element sits, so the rows agree to rounding.
        first = losses[:1].expand_as(losses)
        same = torch.equal(losses, first) if world == 1 else torch.allclose(losses, first, rtol=1e-5, atol=0)
        if not same:
            raise SystemExit("the per-step loss log is not the same row for every step")

        value = args.steps * global_batch / elapsed
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
                "global_batch": global_batch,
                "images_per_gpu": batch,
                "conf_thresh": CONF_T, "iou_thresh": IOU_T, "nms_mode": "reference",
                "parallelism": f"dp{world} by image, loss all-reduce every {every} steps",
                "world_size": dist.get_world_size() if world > 1 else 1,
                "backend": (dist.get_backend() if world > 1 else None),
                "launch": (f"one HIP graph per chunk of <= {every} steps, both streams inside (fork/join)"
                           if use_graph else "eager"),
                "loss_path": "one launch per step, loss sums logged on the device one step late "
                             "(yh_loss_fwd_bwd_pipelined)" if pipelined
                             else "gradient kernel + finalize kernel per step",
                "streams": 1 if args.serial else 2,
            },
            "loss": loss_value,
            "step_gbs": value * (loss_bytes_per_image() + nms_bytes_per_image()) / 1e9,
        }
    elif rank == 0:
        result = {"metric": "YOLOv1 head kernel timing only", "measure": args.measure, "n_gpus": world}

    if rank == 0 and args.measure != "step":
        measure_kernels(args, torch, dev, K, resident, log, result, batch, rank)
        if world == 1 and not args.no_cpu_baseline and args.measure == "all":
            result["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if rank == 0:
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure_kernels(args, torch, dev, K, resident, log, result, batch, rank):
    """Dominant kernel -> roofline, from HIP events on the launch stream.  `frac` is the
    HBM-proper figure: the kernel over a rotation of buffer sets whose own bytes exceed
    twice the memory-side cache, so no launch finds its data there."""
    from yolo_amd import LossLog, synth
    st = torch.cuda.Stream(dev)
    klog = LossLog(dev, capacity=64)
    iters = args.kernel_iters
    algo = loss_bytes_per_image() * batch
    nms_algo = nms_bytes_per_image() * batch
    res, rot = None, None
    if args.measure in ("all", "resident"):
        res = {"loss_fwd_bwd_us": time_launches(torch, lambda i: K.loss(resident, klog, st), 1, iters, st),
               "decode_nms_us": time_launches(torch, lambda i: K.nms(resident, st), 1, iters, st)}
        K.loss_flush(klog, st)
    if args.measure in ("all", "rotated"):
        # each kernel's own bytes (the smaller, decode+NMS, decides) must add up to 2x the cache
        nsets = args.cold_sets if args.cold_sets > 1 else -(-2 * MALL_BYTES // nms_algo) + 1
        base = synth.make_inputs(batch, S, BX, C, IMG, seed=2 * rank)
        sets = [K.make_set(base) for _ in range(nsets)]
        set_bytes = sum(t.numel() * t.element_size() for t in sets[0].values())
        rot = {"buffer_sets": nsets, "total_mb": set_bytes * nsets / 1e6,
               "loss_bytes_in_rotation_mb": algo * nsets / 1e6,
               "decode_nms_bytes_in_rotation_mb": nms_algo * nsets / 1e6,
               "loss_fwd_bwd_us": time_launches(torch, lambda i: K.loss(sets[i], klog, st), nsets, iters, st),
               "decode_nms_us": time_launches(torch, lambda i: K.nms(sets[i], st), nsets, iters, st)}
        K.loss_flush(klog, st)
        if args.measure == "all" and result.get("value") is not None:
            # the whole step over the same rotation (one graph of 2 * nsets steps)
            rot["step_us"] = time_step_rotated(torch, dev, K, sets, klog)
        del sets
    torch.cuda.synchronize()

    traffic, nms_traffic, traffic_note = None, None, "no profiles/traffic.json"
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
                traffic_note = f"PMC FETCH_SIZE/WRITE_SIZE passes of profile {tj.get('profile')} (rotated buffers)"
        except Exception as exc:
            traffic_note = f"profiles/traffic.json unreadable: {exc!r}"

    main_us = (rot or res)["loss_fwd_bwd_us"]
    achieved = algo / (main_us * 1e-6) / 1e9
    roof = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
        "kernel": "loss_fwd_bwd_kernel<%d,%d,%d,%s,false>" % (S, BX, C, "true" if K.pipelined else "false"),
        "regime": "buffers rotate past the memory-side cache (HBM proper)" if rot
                  else "one resident buffer set (memory-side cache)",
        "algorithmic_bytes_per_launch": algo,
        "kernel_us": main_us,
    }
    if rot:
        rot["loss_fwd_bwd_frac"] = algo / (rot["loss_fwd_bwd_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        rot["decode_nms_frac"] = nms_algo / (rot["decode_nms_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        roof["beyond_memory_side_cache"] = rot
    if res:
        res["loss_fwd_bwd_frac"] = algo / (res["loss_fwd_bwd_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        res["decode_nms_frac"] = nms_algo / (res["decode_nms_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        roof["infinity_cache_resident"] = res
    roof["decode_nms"] = {
        "kernel": "nms_kernel<64,1,0,%d,%d,%d>" % (S, BX, C), "algorithmic_bytes_per_launch": nms_algo,
        "kernel_us": (rot or res)["decode_nms_us"],
        "frac": nms_algo / ((rot or res)["decode_nms_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS}
    if args.measure == "all":
        cc = copy_ceiling(torch, dev, st)
        roof["copy_ceiling"] = cc
        roof["frac_of_copy_ceiling"] = achieved / cc["gbs"]
        if rot and rot.get("step_us") and traffic and nms_traffic:
            # the whole step (both kernels, two streams) over rotating buffers: the HBM bytes the
            # two kernels move by the PMC counters / the step's duration, against the copy ceiling
            rot["step_hbm_gbs"] = (traffic + nms_traffic) / (rot["step_us"] * 1e-6) / 1e9
            rot["step_frac_of_copy_ceiling"] = rot["step_hbm_gbs"] / cc["gbs"]
    result["roofline"] = roof
    if args.measure == "all" and not args.no_stress:
        result["configs"] = {"nms_stress": stress_record(args, torch, dev, st)}


def time_step_rotated(torch, dev, K, sets, klog):
    """The whole step (both streams, one graph) over rotating buffer sets, HIP-event timed."""
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    n = 2 * len(sets)

    def enqueue():
        s2.wait_stream(s1)
        for i in range(n):
            K.loss(sets[i % len(sets)], klog, s1)
        K.loss_flush(klog, s1)
        for i in range(n):
            K.nms(sets[i % len(sets)], s2)
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
        for _ in range(8):
            g.replay()
        b.record(s1)
        s1.synchronize()
    return 1e3 * a.elapsed_time(b) / (8 * n)


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
    sets = [K.make_set(base) for _ in range(nsets)]
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
                        "through a rolling LDS window in non-temporal stores (loss_fwd_bwd_roll_kernel)")
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
