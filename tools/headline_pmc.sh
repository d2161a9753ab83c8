# Run on the GPU box from the repository root (probe build present): vector / scalar / LDS instruction
# counts per launch of the three batch-4096 kernels (loss, decode+NMS, one-kernel step; 4,096 wavefronts of
# work each), and of the decode+NMS kernel up to its early exit behind the candidate decode (stop 1).
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/nms_pmc
rm -rf $out && mkdir -p $out
export YOLO_AMD_LIB=build/libyolohead_probe.so
for stop in 1 0; do
  YH_NMS_STOP=$stop timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/s$stop -- python3 tools/pmc_driver.py > $out/s$stop.log 2>&1
  python3 - $out/s$stop $stop <<'PY'
import csv, glob, os, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
v = {}
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    key = "nms" if "nms_kernel" in n else "step" if "head_step" in n else "loss" if "loss_fwd" in n else None
    if key: v.setdefault((key, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
print("stop", sys.argv[2], {k: sorted(x)[len(x)//2] for k, x in sorted(v.items())}, flush=True)
PY
  rm -rf $out/s$stop
done
