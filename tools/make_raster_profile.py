"""Reduce the rocprofv3 passes of `tools/collect_profiles.sh <tag> raster` to profiles/<tag>_raster.json: per-frame means of the
SQ counters and of the memory-side traffic (FETCH_SIZE, WRITE_SIZE; units and the gfx950 correction as MI355X_MICROARCH.md says:
both in KB, FETCH_SIZE doubled) of the tile kernel on BASELINE configs[1] ("cfg2") and on SURVEY.md 8(d)'s stress frame ("stress"),
plus the kernel trace's average durations, stamped with the kernel-source digest."""
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


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]


def main():
    out, tag = sys.argv[1], sys.argv[2]
    from bench import kernel_source_sha

    os.makedirs("profiles", exist_ok=True)
    doc = {"kernel_source_sha16": kernel_source_sha(),
           "method": "per workload: rocprofv3 --kernel-trace --stats, two --pmc passes of SQ counters, --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum "
                     "TCC_MISS_sum (separate runs, never combined with a trace) of tools/raster_bench.py <fixture> <frames> - full frames (setup + binning + "
                     "tiles) of the resident stream; per-dispatch means of raster_tiles<false, ..>; FETCH_SIZE / WRITE_SIZE are KB, FETCH_SIZE doubled (gfx950)"}
    for wl, fixture in (("cfg2", "raster_cfg2_test3_1024_ss4.npz"), ("stress", "stress_degenerate2_1024_ss4.npz")):
        stats = find(out, f"{wl}_stats", "kernel_stats.csv")
        shutil.copy(stats, f"profiles/{tag}_raster_{wl}_kernel_stats.csv")
        kernels = {}
        with open(stats, newline="") as fh:
            for row in csv.DictReader(fh):
                if "raster_" in row["Name"]:
                    kernels[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3}
        tiles = {}
        for sub in ("a", "b"):
            path = find(out, f"{wl}_{sub}", "counter_collection.csv")
            shutil.copy(path, f"profiles/{tag}_raster_{wl}_pmc_{sub}.csv")
            tiles.update(per_dispatch(path, "raster_tiles<false"))
        traffic = None
        try:
            f = per_dispatch(find(out, f"{wl}_fetch", "counter_collection.csv"), "raster_tiles<false")
            w = per_dispatch(find(out, f"{wl}_write", "counter_collection.csv"), "raster_tiles<false")
            shutil.copy(find(out, f"{wl}_fetch", "counter_collection.csv"), f"profiles/{tag}_raster_{wl}_pmc_fetch_size.csv")
            shutil.copy(find(out, f"{wl}_write", "counter_collection.csv"), f"profiles/{tag}_raster_{wl}_pmc_write_size_l2.csv")
            rd, wr = f["FETCH_SIZE"] * 1024 * 2, w["WRITE_SIZE"] * 1024
            traffic = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr, "FETCH_SIZE_KB": f["FETCH_SIZE"], "WRITE_SIZE_KB": w["WRITE_SIZE"],
                       "TCC_HIT_sum": w.get("TCC_HIT_sum"), "TCC_MISS_sum": w.get("TCC_MISS_sum")}
        except SystemExit:
            pass
        doc[wl] = {"fixture": fixture, "kernels": kernels, "tiles_per_frame": tiles, "traffic_per_frame": traffic}
    with open(f"profiles/{tag}_raster.json", "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
