"""Kernel and step timings of several builds of the library side by side.

    python tools/variant_bench.py build/libyolohead_a.so build/libyolohead_b.so ...

Each build runs `bench.py --steps 400` in its own process (YOLO_AMD_LIB selects the
library) and one line per build is printed: the loss, decode+NMS and one-kernel-step
kernels, resident and over rotating buffers, and the step in both regimes (us)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:]:
    env = dict(os.environ, YOLO_AMD_LIB=os.path.abspath(lib))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "400", "--warmup", "50",
                        "--no-cpu-baseline", "--no-stress", "--no-dp-table"] + os.environ.get("VB_ARGS", "").split(),
                       env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        print(lib, "FAILED", p.stderr[-400:])
        continue
    r = json.loads(p.stdout.strip().splitlines()[-1])
    roof = r["roofline"]
    res, rot = roof["infinity_cache_resident"], roof["beyond_memory_side_cache"]
    print(f"{os.path.basename(lib):34s} loss {res['loss_fwd_bwd_us']:6.2f} / {rot['loss_fwd_bwd_us']:6.2f}   "
          f"nms {res['decode_nms_us']:6.2f} / {rot['decode_nms_us']:6.2f}   "
          f"fused {res.get('head_step_us', 0):6.2f} / {rot.get('head_step_us', 0):6.2f}   "
          f"step {r['cache_resident']['step_us']:6.2f} / {1e3 * r['ms_per_step']:6.2f}   "
          f"two-stream step - / {rot['two_stream_step_us']:6.2f}   (resident / rotated, us)", flush=True)
