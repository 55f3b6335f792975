"""Reduce the rocprofv3 CSVs of tools/collect_profiles.sh to profiles/<tag>_{kernel_stats,pmc_*}.csv and
profiles/<tag>_traffic.json: memory-side bytes and SQ instruction counts of ONE EPOCH of the workload (all dispatches of the
path-tracer kernels of the run / the number of epochs), corrected as MI355X_MICROARCH.md says - FETCH_SIZE and WRITE_SIZE are
in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled.  The file is stamped with the digest of the
kernel sources it was taken on; bench.py reports its numbers only while that digest still matches.

usage: make_traffic.py <prof dir> <tag> <sq dir | -> <scene> <epochs>"""
import csv
import glob
import json
import os
import shutil


def keep_counters(src, dst, max_rows=4000):
    """Copies a rocprofv3 counter_collection.csv into profiles/.  A launch of the streamed forms is ~1000 kernel dispatches
    and a raw file runs to 10 MB: beyond `max_rows` rows the file is reduced to one row per (kernel, counter) - dispatches,
    sum, mean, min, max of Counter_Value - which is all make_traffic / bench.py read from it."""
    import csv
    rows = list(csv.DictReader(open(src)))
    if len(rows) <= max_rows:
        shutil.copy(src, dst)
        return
    acc = {}
    for r in rows:
        k = (r["Kernel_Name"], r["Counter_Name"])
        v = float(r["Counter_Value"])
        a = acc.setdefault(k, [0, 0.0, v, v])
        a[0] += 1; a[1] += v; a[2] = min(a[2], v); a[3] = max(a[3], v)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum", "Mean", "Min", "Max"])
        for (kn, cn), (n, sm, mn, mx) in sorted(acc.items()):
            w.writerow([kn, cn, n, repr(sm), repr(sm / n), repr(mn), repr(mx)])
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNELS = ("pt_wave_kernel", "pt_cast_kernel", "pt_compact_kernel", "pt_unit_kernel")


def find(out, sub, suffix):
    hits = sorted(glob.glob(os.path.join(out, sub, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit(f"no {suffix} under {out}/{sub}")
    return hits[0]


def per_epoch(path, epochs, only_three_ray=True):
    """counter -> sum over the dispatches of the path-tracer kernels / epochs"""
    sums = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if not any(k in name for k in KERNELS):
                continue
            sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return {c: v / epochs for c, v in sums.items()}


def per_kernel(path, epochs, counter, scale):
    """kernel (short name) -> bytes per epoch of `counter` (works on raw and on reduced counter files)"""
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter or not any(k in row["Kernel_Name"] for k in KERNELS + ("pt_reduce_kernel",)):
                continue
            v = float(row["Sum"]) if "Sum" in row else float(row["Counter_Value"])
            name = row["Kernel_Name"].replace("void ", "").replace("srt::", "").split("(")[0]
            out[name] = out.get(name, 0.0) + v * scale / epochs
    return out


def main():
    out, tag = sys.argv[1], sys.argv[2]
    sq_dir = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
    scene = sys.argv[4] if len(sys.argv) > 4 else "cbox"
    epochs = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    from bench import kernel_source_sha

    os.makedirs("profiles", exist_ok=True)
    stats = find(out, "stats", "kernel_stats.csv")
    fetch = find(out, "fetch", "counter_collection.csv")
    write = find(out, "write", "counter_collection.csv")
    shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
    keep_counters(fetch, f"profiles/{tag}_pmc_fetch_size.csv")
    keep_counters(write, f"profiles/{tag}_pmc_write_size_l2.csv")
    f = per_epoch(fetch, epochs)
    w = per_epoch(write, epochs)
    kernels = {}
    with open(stats, newline="") as fh:
        for row in csv.DictReader(fh):
            if any(k in row["Name"] for k in KERNELS + ("pt_reduce_kernel", "raster_")):
                kernels[row["Name"][:96]] = {"calls": int(row["Calls"]), "total_ms": float(row["TotalDurationNs"]) / 1e6,
                                             "avg_us": float(row["AverageNs"]) / 1e3}
    fetch_b = f["FETCH_SIZE"] * 1024 * 2
    write_b = w["WRITE_SIZE"] * 1024
    doc = {
        "workload": {"scene": scene, "size": 1024, "spp_per_step": 64, "n_gpus": 1},
        "kernel_source_sha16": kernel_source_sha(),
        "epochs_sampled": epochs,
        "kernels": kernels,
        "FETCH_SIZE_KB_per_launch": f["FETCH_SIZE"],
        "WRITE_SIZE_KB_per_launch": w["WRITE_SIZE"],
        "TCC_HIT_sum": w.get("TCC_HIT_sum"),
        "TCC_MISS_sum": w.get("TCC_MISS_sum"),
        "hbm_read_bytes_per_launch": fetch_b,
        "hbm_write_bytes_per_launch": write_b,
        "hbm_bytes_per_launch": fetch_b + write_b,
        # the same split by kernel (bytes per epoch): which kernel's reads and writes the total is made of
        "per_kernel_read_bytes": per_kernel(fetch, epochs, "FETCH_SIZE", 2048.0),
        "per_kernel_write_bytes": per_kernel(write, epochs, "WRITE_SIZE", 1024.0),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes of "
                  f"`bench.py --scene {scene} --steps 2 --warmup 0 --no-cpu-baseline --no-raster --no-cfg5 --no-overlap --no-elision` "
                  "(2 timed + 2 measuring epochs), summed over the path tracer's kernels and divided by the epochs; FETCH_SIZE doubled per the "
                  "gfx950 note in MI355X_MICROARCH.md; WRITE_SIZE is uncalibrated for dword-per-lane stores",
    }
    if sq_dir:
        sq = {}
        for sub, name in (("a", "pmc_sq_a"), ("b", "pmc_sq_b")):
            path = find(sq_dir, sub, "counter_collection.csv")
            keep_counters(path, f"profiles/{tag}_{name}.csv")
            sq.update(per_epoch(path, epochs))
        doc["sq_per_launch"] = {k: sq[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                                   "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                                   "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_THREAD_CYCLES_VALU") if k in sq}
        doc["valu_issue_ns"] = 1.09   # measured: tools/ubench/pk_rate.hip, v_mul_f32 / v_fma_f32 with >= 2 waves per SIMD
    with open(f"profiles/{tag}_traffic.json", "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
