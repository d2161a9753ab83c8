# Run on the GPU box from the repository root: the counter passes over the stress shape
# (tools/stress_pmc_driver.py), one rocprofv3 --pmc run per group (at most 8 SQ_ counters
# in a run).  Writes profiles/$1_pmc_stress_shape.csv and a copy under gpurun_out/.
set -e
tag=$1
out=$PWD/gpurun_out/stress_pmc_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq -- python3 tools/stress_pmc_driver.py > $out/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/insts -- python3 tools/stress_pmc_driver.py > $out/insts.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/stress_pmc_driver.py > $out/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/stress_pmc_driver.py > $out/write.log 2>&1
python3 - $tag $out <<'PY'
import csv, glob, os, re, sys
tag, out = sys.argv[1:3]
rows = []
for name in ("sq", "insts", "fetch", "write"):
    path = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = {}
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "nms_kernel" in n or "loss_fwd" in n:
            short = ("decode+NMS " if "nms_kernel" in n else "loss ") + re.sub(r"^void yh::|\(.*$", "", n)
            vals.setdefault((short, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(vals.items()):
        rows.append([name, k, c, len(v), sorted(v)[len(v) // 2]])
os.makedirs("profiles", exist_ok=True)
for path in (f"profiles/{tag}_pmc_stress_shape.csv", os.path.join(out, f"{tag}_pmc_stress_shape.csv")):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# rocprofv3 --pmc <one group per pass> --output-format csv -- python3 tools/stress_pmc_driver.py (MI355X): S=14,B=5,C=80, conf 0.001, dense candidates, batch 1024, one resident buffer set; medians over 12 launches; FETCH_SIZE and WRITE_SIZE in KiB (FETCH_SIZE tallies 128-B requests at 64 B: double it); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles"])
        w.writerow(["pass", "kernel", "counter", "launches", "median"])
        w.writerows(rows)
for r in rows:
    print(*r)
PY
rm -rf $out/sq $out/insts $out/fetch $out/write
