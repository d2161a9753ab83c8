"""Turn the rocprofv3 output of a gpurun call into the summaries kept under profiles/.

    python tools/collect_profiles.py <tag> <trace_dir> <pmc_fetch_dir> <pmc_write_dir> "<bench command>"
"""
import csv, glob, json, os, sys
tag, trace_dir, fetch_dir, write_dir, cmd = sys.argv[1:6]
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(os.path.join(trace_dir, "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow([f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd} (MI355X)"])
    w.writerow(list(rows[0].keys()))
    for r in rows:
        r = dict(r); r["Name"] = r["Name"][:110]
        w.writerow(list(r.values()))
for r in rows[:4]:
    print(r["Name"][:48], r["Calls"], r["AverageNs"])
vals, out = {}, []
for counter, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    path = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if any(k in n for k in ("loss_fwd", "nms_kernel", "rocclr_copyBuffer")):
            out.append([counter, r["Dispatch_Id"], n[:70], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Counter_Value"]])
            key = "loss" if "loss_fwd" in n else "nms" if "nms_kernel" in n else "copy"
            vals.setdefault((counter, key), []).append(float(r["Counter_Value"]))
with open(f"profiles/{tag}_pmc_hbm_traffic.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["# rocprofv3 --pmc FETCH_SIZE -- python tools/pmc_driver.py  and  rocprofv3 --pmc WRITE_SIZE -- python tools/pmc_driver.py (separate passes); Counter_Value in KiB"])
    w.writerow(["# the two rocclr_copyBuffer dispatches of 64 MiB are the calibration copies (known bytes): FETCH_SIZE reads 1/2 of the bytes, WRITE_SIZE exact; then 4 launches of the loss kernel (pipelined step) and 3 of decode+NMS, batch 4096"])
    w.writerow(["counter", "dispatch", "kernel", "grid", "workgroup", "vgpr", "value_KiB"])
    w.writerows(out)
med = lambda x: sorted(x)[len(x) // 2]
fl, wl = med(vals[("FETCH_SIZE", "loss")]), med(vals[("WRITE_SIZE", "loss")])
fn, wn = med(vals[("FETCH_SIZE", "nms")]), med(vals[("WRITE_SIZE", "nms")])
fc = [v for v in vals[("FETCH_SIZE", "copy")] if v > 1000]
tj = {
    "batch": 4096, "profile": tag,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_driver.py; values in KiB; "
              "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B: the 64 MiB calibration copies report "
              f"{fc[0]:.0f} KiB fetched; the decode+NMS kernel must read every byte of the 24.08 MB head tensor once and reports "
              f"{fn * 1024 / 1e6:.2f} MB)",
    "loss_kernel_fetch_size_kib_raw": fl, "loss_kernel_write_size_kib": wl,
    "loss_kernel_hbm_bytes_per_launch": int((2 * fl + wl) * 1024),
    "nms_kernel_fetch_size_kib_raw": fn, "nms_kernel_write_size_kib": wn,
    "nms_kernel_hbm_bytes_per_launch": int((2 * fn + wn) * 1024),
}
json.dump(tj, open("profiles/traffic.json", "w"), indent=1)
print(tj["loss_kernel_hbm_bytes_per_launch"], tj["nms_kernel_hbm_bytes_per_launch"])
