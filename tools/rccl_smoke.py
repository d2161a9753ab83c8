"""One-rank check of the pieces bench.py uses at N > 1 with the "nccl" (RCCL)
backend: process group with device_id, barrier, asynchronous all-reduce of the
loss log on a side stream while HIP graphs replay, results()."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
from yolo_amd import YOLOv1Head, LossLog, synth
from yolo_amd.dist import LossAllReducer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
dist.barrier()
head = YOLOv1Head(); log = LossLog(dev, 32); red = LossAllReducer()
pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(256))
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(4): head.loss_step_pipelined(pred, gx, gc, om, log)
    head.flush_loss_log(log)
torch.cuda.synchronize(); log.reset()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    for _ in range(16): head.loss_step_pipelined(pred, gx, gc, om, log)
for chunk in range(3):
    with torch.cuda.stream(s):
        g.replay(); g.replay()
        head.flush_loss_log(log)
        red.flush(log.ring, 32)
with torch.cuda.stream(s):
    out = red.results()
torch.cuda.synchronize(); dist.barrier()
assert out.shape == (96, 4) and torch.equal(out, out[:1].expand_as(out)), out[:3]
print("rccl smoke ok", out[0].tolist())
dist.destroy_process_group()
