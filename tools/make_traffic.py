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
    avg_ns = avg_ns_two = None      # the all-rays build (NR = 3, the bench's `value`) and the two-ray build of the elision pass
    with open(stats, newline="") as fh:
        for row in csv.DictReader(fh):
            if "pt_wave_kernel" in row["Name"]:
                if ", 2>" in row["Name"]:
                    avg_ns_two = float(row["AverageNs"])
                else:
                    avg_ns = float(row["AverageNs"])
    fetch_b = f["FETCH_SIZE"] * 1024 * 2
    write_b = w["WRITE_SIZE"] * 1024
    doc = {
        "workload": {"scene": "cbox", "size": 1024, "spp_per_step": 64, "n_gpus": 1},
        "kernel": "pt_wave_kernel",
        "kernel_avg_ns": avg_ns,
        "two_ray_kernel_avg_ns": avg_ns_two,
        "launches_sampled": {"fetch": nf, "write": nw},
        "FETCH_SIZE_KB_per_launch": f["FETCH_SIZE"],
        "WRITE_SIZE_KB_per_launch": w["WRITE_SIZE"],
        "TCC_HIT_sum": w.get("TCC_HIT_sum"),
        "TCC_MISS_sum": w.get("TCC_MISS_sum"),
        "hbm_read_bytes_per_launch": fetch_b,
        "hbm_write_bytes_per_launch": write_b,
        "hbm_bytes_per_launch": fetch_b + write_b,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes of "
                  "`bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-raster --no-overlap --no-elision`; FETCH_SIZE doubled per the gfx950 "
                  "note in MI355X_MICROARCH.md; WRITE_SIZE is uncalibrated for this kernel's dword-per-lane stores",
    }
    # optional: the SQ passes of tools/pmc_sq.sh <tag minus the round prefix> (gpurun_out/pmc_*/{a,b}) -> wave-instruction counts
    sq_dir = sys.argv[3] if len(sys.argv) > 3 else None
    if sq_dir:
        sq = {}
        for sub, name in (("a", "pmc_sq_a"), ("b", "pmc_sq_b")):
            path = find(sq_dir, sub, "counter_collection.csv")
            shutil.copy(path, f"profiles/{tag}_{name}.csv")
            sq.update(per_launch(path, "pt_wave_kernel")[0])
        doc["sq_per_launch"] = {k: sq[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                                   "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in sq}
        doc["valu_issue_ns"] = 1.09   # measured: tools/ubench/pk_rate.hip, v_mul_f32 / v_fma_f32 with >= 2 waves per SIMD
    with open(f"profiles/{tag}_traffic.json", "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
