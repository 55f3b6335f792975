"""Reduce the rocprofv3 CSVs of tools/collect_profiles.sh to profiles/<tag>_{kernel_stats,pmc_*}.csv and
profiles/<tag>_traffic.json (HBM bytes per pt_wave_kernel launch, corrected as MI355X_MICROARCH.md says:
FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled)."""
import csv
import glob
import json
import os
import shutil
import sys


def find(out, sub, suffix):
    hits = sorted(glob.glob(os.path.join(out, sub, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit(f"no {suffix} under {out}/{sub}")
    return hits[0]


def per_launch(path, kernel_substr):
    """counter -> mean value per dispatch of the named kernel"""
    sums, calls = {}, {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if kernel_substr not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            sums[c] = sums.get(c, 0.0) + float(row["Counter_Value"])
            calls[c] = calls.get(c, 0) + 1
    return {c: sums[c] / calls[c] for c in sums}, (max(calls.values()) if calls else 0)


def main():
    out, tag = sys.argv[1], sys.argv[2]
    os.makedirs("profiles", exist_ok=True)
    stats = find(out, "stats", "kernel_stats.csv")
    fetch = find(out, "fetch", "counter_collection.csv")
    write = find(out, "write", "counter_collection.csv")
    shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
    shutil.copy(fetch, f"profiles/{tag}_pmc_fetch_size.csv")
    shutil.copy(write, f"profiles/{tag}_pmc_write_size_l2.csv")
    f, nf = per_launch(fetch, "pt_wave_kernel")
    w, nw = per_launch(write, "pt_wave_kernel")
    avg_ns = None
    with open(stats, newline="") as fh:
        for row in csv.DictReader(fh):
            if "pt_wave_kernel" in row["Name"]:
                avg_ns = float(row["AverageNs"])
    fetch_b = f["FETCH_SIZE"] * 1024 * 2
    write_b = w["WRITE_SIZE"] * 1024
    doc = {
        "workload": {"scene": "cbox", "size": 1024, "spp_per_step": 64, "n_gpus": 1},
        "kernel": "pt_wave_kernel",
        "kernel_avg_ns": avg_ns,
        "launches_sampled": {"fetch": nf, "write": nw},
        "FETCH_SIZE_KB_per_launch": f["FETCH_SIZE"],
        "WRITE_SIZE_KB_per_launch": w["WRITE_SIZE"],
        "TCC_HIT_sum": w.get("TCC_HIT_sum"),
        "TCC_MISS_sum": w.get("TCC_MISS_sum"),
        "hbm_read_bytes_per_launch": fetch_b,
        "hbm_write_bytes_per_launch": write_b,
        "hbm_bytes_per_launch": fetch_b + write_b,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes of "
                  "`bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-raster`; FETCH_SIZE doubled per the gfx950 "
                  "note in MI355X_MICROARCH.md; WRITE_SIZE is uncalibrated for this kernel's dword-per-lane stores",
    }
    with open(f"profiles/{tag}_traffic.json", "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
