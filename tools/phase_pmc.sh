# Run on the GPU box from the repository root (probe build present: make -C yolo_amd/csrc probe):
# vector / scalar / LDS instruction counts of the stress-shape decode+NMS kernel up to each early exit.
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/phase_pmc
rm -rf $out && mkdir -p $out
export YOLO_AMD_LIB=${PROBE_LIB:-build/libyolohead_probe.so}
for stop in 1 2 3 5 6 4 0; do
  YH_NMS_STOP=$stop timeout -s KILL 100 rocprofv3 --pmc ${PMC_COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES} --output-format csv -d $out/s$stop -- python3 tools/stress_pmc_driver.py > $out/s$stop.log 2>&1
  python3 - $out/s$stop $stop <<'PY'
import csv, glob, os, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
v = {}
for r in csv.DictReader(open(path)):
    if "nms_kernel" in r["Kernel_Name"]:
        v.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("stop", sys.argv[2], {k: sorted(x)[len(x)//2] for k, x in sorted(v.items())}, flush=True)
PY
  rm -rf $out/s$stop
done
