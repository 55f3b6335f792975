"""Reduce the rocprofv3 passes of `tools/collect_profiles.sh <tag> raster` to profiles/<tag>_raster.json: per-frame means of the
SQ counters of raster_tiles (and the binning passes) on BASELINE configs[1], stamped with the kernel-source digest."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find(out, sub, suffix):
    hits = sorted(glob.glob(os.path.join(out, sub, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit(f"no {suffix} under {out}/{sub}")
    return hits[0]


def per_dispatch(path, kernel_substr):
    sums, calls = {}, {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if kernel_substr not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            sums[c] = sums.get(c, 0.0) + float(row["Counter_Value"])
            calls[c] = calls.get(c, 0) + 1
    return {c: sums[c] / calls[c] for c in sums}


def main():
    out, tag = sys.argv[1], sys.argv[2]
    from bench import kernel_source_sha

    os.makedirs("profiles", exist_ok=True)
    stats = find(out, "stats", "kernel_stats.csv")
    shutil.copy(stats, f"profiles/{tag}_raster_kernel_stats.csv")
    kernels = {}
    with open(stats, newline="") as fh:
        for row in csv.DictReader(fh):
            if "raster_" in row["Name"]:
                kernels[row["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3}
    tiles, bins = {}, {}
    for sub in ("a", "b"):
        path = find(out, sub, "counter_collection.csv")
        shutil.copy(path, f"profiles/{tag}_raster_pmc_{sub}.csv")
        tiles.update(per_dispatch(path, "raster_tiles<false"))
        bins.update({k: v for k, v in per_dispatch(path, "raster_bin_pass<1>").items()})
    doc = {"workload": {"scene": "raster_cfg2", "svg": "basic/test3.svg", "size": 1024, "sample_rate": 4},
           "kernel_source_sha16": kernel_source_sha(), "kernels": kernels, "raster_tiles_per_frame": tiles, "bin_pass1_per_frame": bins,
           "method": "rocprofv3 --kernel-trace --stats and two --pmc passes (SQ counters) of tools/raster_bench.py raster_cfg2_test3_1024_ss4.npz 100; "
                     "per-dispatch means of raster_tiles<false, 16>"}
    with open(f"profiles/{tag}_raster.json", "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
