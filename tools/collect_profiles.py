"""Turn the rocprofv3 output of a gpurun call into the summaries kept under profiles/.

    python tools/collect_profiles.py <tag> <pmc_fetch_dir> <pmc_write_dir> <name>=<trace_dir>=<command> ...

Every <name>=<trace_dir>=<command> becomes profiles/<tag>_kernel_stats_<name>.csv (the
rocprofv3 --kernel-trace --stats summary of that command); the two PMC passes over
tools/pmc_driver.py become profiles/<tag>_pmc_hbm_traffic.csv and profiles/traffic.json,
which carries a fingerprint of the kernel sources it was measured on."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tag, fetch_dir, write_dir = sys.argv[1:4]
os.makedirs("profiles", exist_ok=True)
for spec in sys.argv[4:]:
    name, trace_dir, cmd = spec.split("=", 2)
    stats = glob.glob(os.path.join(trace_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    with open(f"profiles/{tag}_kernel_stats_{name}.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow([f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd} (MI355X)"])
        w.writerow(list(rows[0].keys()))
        for r in rows:
            r = dict(r); r["Name"] = r["Name"][:110]
            w.writerow(list(r.values()))
    print(name)
    for r in rows[:4]:
        print("  ", r["Name"][:60], r["Calls"], r["AverageNs"])
vals, out = {}, []
for counter, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if any(k in n for k in ("loss_fwd", "nms_kernel", "head_step", "rocclr_copyBuffer")):
            key = "loss" if "loss_fwd" in n else "nms" if "nms_kernel" in n else "step" if "head_step" in n else "copy"
            vals.setdefault((counter, key), []).append(float(r["Counter_Value"]))
            if (key == "copy" and float(r["Counter_Value"]) > 30000) or (key != "copy" and len(vals[(counter, key)]) <= 6):      # the first launches of each, as a sample
                out.append([counter, r["Dispatch_Id"], n[:70], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Counter_Value"]])
med = lambda x: sorted(x)[len(x) // 2]
with open(f"profiles/{tag}_pmc_hbm_traffic.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["# rocprofv3 --pmc FETCH_SIZE -- python3 tools/pmc_driver.py  and  rocprofv3 --pmc WRITE_SIZE -- python3 tools/pmc_driver.py (separate passes); Counter_Value in KiB"])
    w.writerow(["# the two rocclr_copyBuffer dispatches of 64 MiB are the calibration copies (known bytes): FETCH_SIZE reads 1/2 of the bytes, WRITE_SIZE exact; then 34 launches each of the loss kernel (pipelined step), of decode+NMS (49 rows per image) and of the one-kernel step, batch 4096, over 17 rotating buffer sets (first six of each listed, medians of all below)"])
    w.writerow(["counter", "dispatch", "kernel", "grid", "workgroup", "vgpr", "value_KiB"])
    w.writerows(out)
    for (counter, key), v in sorted(vals.items()):
        if key != "copy":
            w.writerow([counter, "median", key, len(v), "", "", f"{med(v):.4f}"])
fl, wl = med(vals[("FETCH_SIZE", "loss")]), med(vals[("WRITE_SIZE", "loss")])
fn, wn = med(vals[("FETCH_SIZE", "nms")]), med(vals[("WRITE_SIZE", "nms")])
fs, wst = med(vals[("FETCH_SIZE", "step")]), med(vals[("WRITE_SIZE", "step")])
fc = [v for v in vals[("FETCH_SIZE", "copy")] if 30000 < v < 36000]    # the 64 MiB calibration copies
tj = {
    "batch": 4096, "profile": tag, "sources_sha": bench.kernel_sources_sha(),
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_driver.py "
              "(17 rotating buffer sets, medians over 34 launches per kernel); values in KiB; FETCH_SIZE doubled "
              "(gfx950 tallies 128-B read requests at 64 B: the 64 MiB calibration copies report "
              f"{fc[0]:.0f} KiB fetched; the decode+NMS kernel must read every byte of the 24.08 MB head tensor once and reports "
              f"{fn * 1024 / 1e6:.2f} MB)",
    "loss_kernel_fetch_size_kib_raw": fl, "loss_kernel_write_size_kib": wl,
    "loss_kernel_hbm_bytes_per_launch": int((2 * fl + wl) * 1024),
    "nms_kernel_fetch_size_kib_raw": fn, "nms_kernel_write_size_kib": wn,
    "nms_kernel_hbm_bytes_per_launch": int((2 * fn + wn) * 1024),
    "head_step_kernel_fetch_size_kib_raw": fs, "head_step_kernel_write_size_kib": wst,
    "head_step_kernel_hbm_bytes_per_launch": int((2 * fs + wst) * 1024),
}
json.dump(tj, open("profiles/traffic.json", "w"), indent=1)
print(tj["loss_kernel_hbm_bytes_per_launch"], tj["nms_kernel_hbm_bytes_per_launch"], tj["head_step_kernel_hbm_bytes_per_launch"])
