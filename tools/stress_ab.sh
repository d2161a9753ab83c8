#!/bin/bash
# A/B of the stress-shape decode+NMS between two builds of the library, alternating, on one box:
# tools/stress_ab.sh build/libyolohead_base.so yolo_amd/lib/libyolohead.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for lib in $A $B; do
    echo "== $lib"
    YOLO_AMD_LIB=$lib KB_BATCH=${KB_BATCH:-1024} timeout -k 10 120 python tools/stress_bench.py 2>/dev/null | grep -E "dense NMS" || exit 1
  done
done
