#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native hot paths.

Metric (BASELINE.json): Mrays/s on the Cornell box, 1024x1024, 2048 spp, image-tiled over N GPUs;
plus Mfrags/s of the DrawSVG triangle fill (single GPU) as a second object on the same JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One STEP = one epoch (Pathtracer::do_trace) of --spp-per-step (default 64) samples per pixel over the whole
1024x1024 image: 67.1 M camera samples, ~0.5 G rays.  32 steps are the full 2048-spp render of
BASELINE configs[3]; throughput does not depend on K.  Each rank renders its tiles (32x32, round-robin),
rank 0 gathers the tile radiance with one RCCL gather per step, un-tiles and folds the epoch into the
running mean (Pathtracer::accumulate).  A ray is one scene.hit call; rays are counted
on the device by the production kernel.  Inputs (scene, camera) are resident in HBM before the timed
region; the timed region is bracketed by barrier + synchronize and the max over ranks is reported.
"""
import argparse
import ctypes
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth (8 XCDs x 4 MiB), same guide: what bounds a working set that lives in L2 / Infinity Cache
SIMDS = 256 * 4
VALU_PEAK_GUIDE = SIMDS * 2.4e9 / 2.0     # wave-instructions / s: one wave64 VALU op per 2 cycles per SIMD at 2.4 GHz (the guide)
VALU_NS_UBENCH = 1.09                     # measured issue interval of a SIMD with >= 2 waves (tools/ubench/pk_rate.hip)


def kernel_source_sha():
    """Digest of the kernel sources a profile was taken on: stale PMC numbers must not end up in a fresh bench line."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "soft-rendering-toolsets_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")) or f == "Makefile":
            h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def bytes_per_ray(counters):
    """Algorithmic bytes per ray, SURVEY.md §8(d): 32 B per BVH node visited (TLAS + BLAS), 128 B per
    object entered (trans + itrans), 36 B per triangle tested, 4 B per sphere tested, 36 B hit normals."""
    r = float(counters["rays"])
    return (32.0 * (counters["tlas_nodes"] + counters["blas_nodes"]) + 128.0 * counters["objects_entered"]
            + 36.0 * counters["tri_tests"] + 4.0 * counters["sphere_tests"]) / r + 36.0


def cpu_cores():
    """Host threads to use: the cgroup CPU quota when there is one (a 1-GPU box gets a 16-CPU share of a
    much larger host), else the affinity mask."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("SRT_BENCH_CPU_THREADS", "16")))


def pt_cpu_baseline(scene, w, h, depth, seed, pt, budget_s=15.0):
    """The CPU path on the host cores, same scene / size / seed, reduced spp, rows split over `cores` threads
    (ctypes releases the GIL), the way the reference's thread pool spreads its epochs.
    kind "reference": the reference's own PT::Pathtracer::trace_pixel compiled from /root/reference into
    oracle/_ref/libref_pt.so (prebuilt, travels with the tree); its image must equal the HIP image of the same
    samples bit for bit, and its ray count is the HIP path's count for those samples.
    kind "port": oracle/pt_oracle.c (the validated restatement) when oracle/_ref is absent."""
    import _harness as H

    cores = cpu_cores()
    use_ref = H.ref_pt_lib() is not None
    o = H.RefPT(scene, w, h, depth, True) if use_ref else H.OraclePT(scene, w, h, depth, True, math_mode=1)
    img = np.zeros((h, w, 3), np.float32)

    def run(spp, rows):
        cnts = [np.zeros(8, np.uint64) for _ in range(cores)]
        bounds = np.linspace(0, rows, cores + 1).astype(int)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            if use_ref:
                list(ex.map(lambda k: o.epoch_rows(seed, 0, spp, int(bounds[k]), int(bounds[k + 1]), img), range(cores)))
            else:
                list(ex.map(lambda k: o.epoch(seed, 0, spp, int(bounds[k]), int(bounds[k + 1]), img, cnts[k]), range(cores)))
        dt = time.perf_counter() - t0
        return dt, int(sum(int(c[0]) for c in cnts))

    dt, rays = run(1, max(cores, h // 8))           # probe: 1/8 of the rows at 1 spp
    per_spp_full = dt * 8.0
    spp = int(max(1, min(64, budget_s / max(per_spp_full, 1e-3))))
    dt, rays = run(spp, h)
    out = {"unit": "Mrays/s", "cores": cores, "kind": "reference" if use_ref else "port"}
    if use_ref:
        # the HIP path on exactly these samples: same image bits, and its scene.hit count is the reference's
        pt.set_tiling(32, 32, 0, 1)
        pt.ray_count(reset=True)
        gpu = pt.render_epoch(seed, 0, spp)
        rays, _ = pt.ray_count(reset=True)
        out["image_equals_hip_bit_for_bit"] = bool(np.array_equal(gpu.view(np.uint32), img.view(np.uint32)))
        what = "oracle/_ref/libref_pt.so = the reference's PT::Pathtracer::trace_pixel (clang++ -O2) with the seeded SRT-RNG"
    else:
        what = "oracle/pt_oracle.c"
    out["value"] = rays / dt / 1e6
    out["sample"] = (f"{what}, same scene/seed, {w}x{h} at {spp} spp ({rays / 1e6:.1f} M rays in {dt:.1f} s), "
                     f"{w * h * spp / dt / 1e6:.2f} M camera samples/s")
    return out


def _raster_roofline(prof, tag, kernel_prefix, ms_tiles_live, alg_bytes):
    """The tile kernel against vector-instruction issue (what binds it; counters from the committed PMC passes of the same
    kernel sources) with SURVEY.md 8(d)'s HBM figure beside it."""
    hbm = {"bound": "hbm", "achieved": alg_bytes / (ms_tiles_live * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg_bytes / (ms_tiles_live * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "note": "SURVEY.md 8(d): 40 B x (primitive, tile) entries + 4 B x pixels over the tile kernel's live duration; the 256 MiB "
                   "supersample buffer never leaves the CUs, so this is tiny by design"}
    if not prof or tag not in prof:
        return dict(hbm, traffic=None)
    t = prof[tag]["tiles_per_frame"]
    kname, kus = next(((k, v["avg_us"]) for k, v in prof[tag]["kernels"].items() if k.startswith(kernel_prefix)), (None, None))
    rate = t["SQ_INSTS_VALU"] / (kus * 1e-6)
    tr = prof[tag].get("traffic_per_frame")
    return {"bound": "valu", "achieved": rate / 1e9, "peak": VALU_PEAK_GUIDE / 1e9, "unit": "G wave-instructions/s",
            "frac": rate / VALU_PEAK_GUIDE, "frac_ubench_ceiling": rate / (SIMDS / (VALU_NS_UBENCH * 1e-9)),
            "traffic": tr["hbm_bytes"] if tr else None, "traffic_detail": tr,
            "kernel": kname, "kernel_us_profiled": kus, "kernel_us_live": ms_tiles_live * 1e3,
            "valu_lane_utilisation": (t.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * t["SQ_ACTIVE_INST_VALU"])) if t.get("SQ_ACTIVE_INST_VALU") else None,
            "wave_cycles_waiting": t.get("SQ_WAIT_ANY", 0.0) / t["SQ_WAVE_CYCLES"] if t.get("SQ_WAVE_CYCLES") else None,
            "wave_cycles_issuing": t.get("SQ_ACTIVE_INST_ANY", 0.0) / t["SQ_WAVE_CYCLES"] if t.get("SQ_WAVE_CYCLES") else None,
            "lds_instructions_per_frame": t.get("SQ_INSTS_LDS"), "lds_bank_conflict_cycles_per_frame": t.get("SQ_LDS_BANK_CONFLICT"),
            "lds_active_cycles_per_frame": t.get("SQ_LDS_IDX_ACTIVE"),
            "kernels_us_profiled": {k: v["avg_us"] for k, v in prof[tag]["kernels"].items()},
            "source": prof["_file"], "hbm": hbm}


def _timed_frames(torch, ren, stream, frames, full):
    """ms per frame (HIP events on the launch stream) of `frames` device frames; full: setup + binning + tiles (the stream is
    resident, what the library derived from it is discarded before every frame), else the tile kernel alone (a redraw)."""
    for _ in range(3):
        if full:
            ren.invalidate()
        ren.resolve_device(stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(frames):
        if full:
            ren.invalidate()
        ren.resolve_device(stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / frames


def raster_bench(device, frames=30, warmup=3):
    """BASELINE configs[1]: basic/test3.svg (1963 triangles + 1992 lines), 1024x1024, supersample 4.
    `value` follows SURVEY.md 8(d): covered fragments per second of the `draw_svg` WALL - DrawSVG's redraw through the drop-in
    class (stream build on the host, upload, setup, binning, tiles, resolve, read-back into the application's framebuffer),
    measured in integration/_build/libdropin_raster.so = SoftwareRendererHIP inside the reference's own DrawSVG sources (an
    integration harness of the PRODUCT, built in the authoring container; every pixel comes from libsrt_hip.so).  Without that
    library `value` is null - its definition never changes; the C-ABI wall from the prebuilt stream (`c_abi_wall_ms`, `c_abi_value`) and
    the device-only figures - a full frame of the resident stream, the tile kernel alone - are always reported under their own keys."""
    import ctypes
    import glob

    import torch

    import _harness as H
    import srt_amd

    del warmup
    g = np.load(os.path.join(H.GOLDEN, "raster_cfg2_test3_1024_ss4.npz"))
    w, h, sr = (int(x) for x in g["meta"])
    ren = srt_amd.SoftwareRenderer(device)
    fb = np.empty((h, w, 4), np.uint8)
    ren.set_render_target(fb, w, h)
    ren.set_sample_rate(sr)
    out = ren.draw_stream(g["prims"]).copy()     # uploads the stream, checks the result
    ok = bool(np.array_equal(out, g["rgba"]))
    st = ren.stats()
    stream = torch.cuda.current_stream().cuda_stream
    ms_full = _timed_frames(torch, ren, stream, frames, True)
    ms_tiles = _timed_frames(torch, ren, stream, frames, False)
    # the boundary from the prebuilt stream: clear + submit + resolve into the pinned framebuffer (no SVG walk)
    alt = g["prims"].copy()
    alt["rgba"][:, 0] *= 0.5
    t0 = time.perf_counter()
    for k in range(frames):
        ren.draw_stream(alt if k % 2 else g["prims"])
    abi_new_ms = (time.perf_counter() - t0) * 1e3 / frames
    t0 = time.perf_counter()
    for k in range(frames):
        ren.draw_stream(g["prims"])
    abi_same_ms = (time.perf_counter() - t0) * 1e3 / frames
    alg_bytes = 40.0 * st.bin_entries + 4.0 * w * h          # SURVEY.md 8(d) rasterizer formula (fused resolve)
    o_rgba, _, counts = H.oracle_raster_frame(g["prims"], w, h, sr)       # fragment count of the frame (checker)
    use_ref = os.path.exists(os.path.join(H.ORACLE_DIR, "_ref", "libref_raster.so"))
    t = time.perf_counter()
    if use_ref:   # the reference's SoftwareRendererImp (clear + rasterize_* + resolve) on the same primitive stream
        r_rgba, _ = H.ref_raster_prims(g["prims"], w, h, sr)
    else:
        r_rgba, _, _ = H.oracle_raster_frame(g["prims"], w, h, sr)
    cpu_s = time.perf_counter() - t
    ok = ok and bool(np.array_equal(r_rgba, out))
    ren.close()
    # DrawSVG's redraw through the drop-in class
    dropin, wall = os.path.join(ROOT, "integration", "_build", "libdropin_raster.so"), None
    svg = os.path.join(H.GOLDEN, "svg", "test3.svg")
    if os.path.exists(dropin) and os.path.exists(svg):
        lib = ctypes.CDLL(dropin)
        ms3 = (ctypes.c_double * 5)()
        d_out = np.zeros((h, w, 4), np.uint8)
        if lib.dropin_raster_bench(svg.encode(), device, w, h, sr, frames, ms3, H.P(d_out)) == 0:
            wall = {"draw_svg_wall_ms": ms3[3], "draw_svg_unchanged_view_wall_ms": ms3[4],
                    "redraw_wall_ms": ms3[0], "redraw_unchanged_view_wall_ms": ms3[1], "host_stream_build_ms": ms3[2],
                    "framebuffer_equals_reference_golden": bool(np.array_equal(d_out, g["rgba"])),
                    "what": "integration/_build/libdropin_raster.so: CMU462::SoftwareRendererHIP linked into the reference's DrawSVG sources and driven "
                            "like DrawSVG::redraw (clear(), set_svg_2_screen, draw_svg), the view moving every frame.  draw_svg_wall_ms = the "
                            "time inside software_renderer->draw_svg() - SURVEY.md 8(d)'s `draw_svg` wall (clear + fill + resolve): SVG walk "
                            "with cached triangulations on the host, upload, setup, binning, tiles, resolve, 4 MiB read-back into the "
                            "application's framebuffer; redraw_wall_ms = the whole of DrawSVG::redraw, i.e. plus the APPLICATION's clear() - a "
                            "4 MiB memset of the framebuffer in the reference's base class, before the renderer is called; *_unchanged_view_* = "
                            "the same with the view left alone (stream found identical: tile kernel + read-back)"}
            ok = ok and wall["framebuffer_equals_reference_golden"]
            ph = (ctypes.c_double * 5)()
            if hasattr(lib, "dropin_raster_phases") and lib.dropin_raster_phases(svg.encode(), device, w, h, sr, frames, ph) == 0:
                # the same redraw step by step over the C ABI: where the wall time goes (host clock around each step)
                wall["phases_ms"] = {"application_clear_target_memset": ph[0], "host_stream_build": ph[1], "clear_and_submit": ph[2],
                                     "resolve_upload_kernels_readback_wait": ph[3], "redraw": ph[4]}
    e2e_ms = wall["draw_svg_wall_ms"] if wall else None       # (never another quantity under the same key)
    # what binds the tile kernel: vector-instruction issue and LDS, from the committed SQ counter passes of the same kernel sources
    prof, why_not = None, "no committed PMC pass for the rasterizer"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_raster.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if doc.get("kernel_source_sha16") == kernel_source_sha() and "cfg2" in doc:
            prof, why_not = doc, None
            prof["_file"] = os.path.relpath(path, ROOT)
            break
        why_not = f"{os.path.relpath(path, ROOT)} was taken on other kernel sources: counters not reported"
    roof = _raster_roofline(prof, "cfg2", "raster_tiles<false", ms_tiles, alg_bytes)
    if not prof and why_not:
        roof["counters"] = why_not
    stress = None
    sp = os.path.join(H.GOLDEN, "stress_degenerate2_1024_ss4.npz")
    if os.path.exists(sp):     # SURVEY.md 8(d)'s stress variant: 1000 frame-sized translucent triangles, 5.86 G sample tests (22.7 s in the reference)
        gs = np.load(sp)
        sw, sh, ssr = (int(x) for x in gs["meta"])
        r2 = srt_amd.SoftwareRenderer(device)
        fb2 = np.empty((sh, sw, 4), np.uint8)
        r2.set_render_target(fb2, sw, sh)
        r2.set_sample_rate(ssr)
        o2 = r2.draw_stream(gs["prims"]).copy()
        st2 = r2.stats()
        sms = _timed_frames(torch, r2, stream, 5, True)
        sms_tiles = _timed_frames(torch, r2, stream, 5, False)
        t0 = time.perf_counter()
        for _ in range(5):
            r2.invalidate(); r2.draw_stream(gs["prims"])
        s_wall = (time.perf_counter() - t0) * 1e3 / 5
        stress = {"workload": "DrawSVG hardcore/02_degenerate_square2.svg 1024x1024 supersample=4", "ms_per_frame": sms, "tiles_ms": sms_tiles,
                  "wall_ms_per_frame_through_the_c_abi": s_wall,
                  "sample_tests": int(st2.sample_tests), "fragments": int(st2.fragments), "sample_tests_per_s": st2.sample_tests / (sms * 1e-3),
                  "mfrags_per_s": st2.fragments / (s_wall * 1e-3) / 1e6, "bit_exact_vs_reference_golden": bool(np.array_equal(o2, gs["rgba"])),
                  "roofline": _raster_roofline(prof, "stress", "raster_tiles<false", sms_tiles, 40.0 * st2.bin_entries + 4.0 * sw * sh),
                  "reference_cpu_s": 22.7, "reference_cpu_source": "BASELINE.md section 2 (survey-time probe, 1 thread)"}
        r2.close()
    images = None
    ip = os.path.join(H.GOLDEN, "raster_test7_image_256_ss2.npz")
    if os.path.exists(ip):     # basic/test7.svg (<image> elements): DrawSVG's redraw clears and re-adds every mip chain each frame
        gi = np.load(ip)
        iw, ih, isr = (int(x) for x in gi["meta"])
        tex = H.Textures.from_npz(gi)
        r3 = srt_amd.SoftwareRenderer(device)
        fb3 = np.empty((ih, iw, 4), np.uint8)
        r3.set_render_target(fb3, iw, ih)
        r3.set_sample_rate(isr)
        chains = [tex.texture(k) for k in range(len(tex))]

        def redraw():
            r3.clear_textures()
            for c in chains:
                r3.add_texture(c)
            return r3.draw_stream(gi["prims"])

        first = redraw().copy()
        up_first = r3.texture_upload_bytes()
        t0 = time.perf_counter()
        for _ in range(frames):
            redraw()
        img_ms = (time.perf_counter() - t0) * 1e3 / frames
        images = {"workload": f"DrawSVG basic/test7.svg {iw}x{ih} supersample={isr}: {int((gi['prims']['kind'] == 3).sum())} image records, {len(tex)} textures",
                  "redraw_wall_ms_through_the_c_abi": img_ms, "texel_bytes_uploaded_first_frame": up_first,
                  "texel_bytes_uploaded_per_later_redraw": (r3.texture_upload_bytes() - up_first) / frames,
                  "bit_exact_vs_reference_golden": bool(np.array_equal(first, gi["rgba"]))}
        ok = ok and images["bit_exact_vs_reference_golden"]
        r3.close()
    return {
        "images": images,
        "metric": "Mfrags/s triangle fill", "value": (st.fragments / (e2e_ms * 1e-3) / 1e6) if e2e_ms else None, "unit": "Mfrags/s",
        "value_is": "covered fragments / draw_svg wall (SURVEY.md 8(d)): the time inside SoftwareRendererHIP::draw_svg during DrawSVG's redraw, the view "
                    "moving every frame (draw_svg.redraw_wall_ms adds the application's own clear()); null when integration/_build/libdropin_raster.so is absent",
        "wall_ms_per_frame": e2e_ms, "draw_svg": wall,
        "c_abi_value": st.fragments / (abi_new_ms * 1e-3) / 1e6,
        "c_abi_wall_ms": {"new_stream_every_frame": abi_new_ms, "same_stream": abi_same_ms,
                          "what": "srt_raster_clear + srt_raster_submit + srt_raster_resolve into the pinned framebuffer from Python (ctypes), "
                                  "no SVG walk"},
        "device_ms_per_frame": ms_full, "device_tiles_only_ms": ms_tiles,
        "device_mfrags_per_s": st.fragments / (ms_full * 1e-3) / 1e6,
        "binning_share_of_device_frame": max(0.0, 1.0 - ms_tiles / ms_full), "frames": frames,
        "config": {"workload": "DrawSVG basic/test3.svg 1024x1024 supersample=4 (BASELINE configs[1])",
                   "triangles": int((g["prims"]["kind"] == 1).sum()), "lines": int((g["prims"]["kind"] == 4).sum()),
                   "points": int((g["prims"]["kind"] == 2).sum()),
                   "fragments": int(st.fragments), "sample_tests": int(st.sample_tests), "bin_entries": int(st.bin_entries)},
        "sample_tests_per_s": st.sample_tests / ((e2e_ms or abi_new_ms) * 1e-3),
        "bit_exact_vs_reference_golden": ok, "dtype": "f64 edge functions / f32 blend",
        "roofline": roof, "list_bytes": int(st.list_bytes), "stress": stress,
        "cpu_baseline": {"value": int(counts[2]) / cpu_s / 1e6, "unit": "Mfrags/s", "cores": 1,
                         "kind": "reference" if use_ref else "port",
                         "sample": (("oracle/_ref/libref_raster.so = the reference's SoftwareRendererImp (g++ -O2)" if use_ref
                                     else "oracle/raster_oracle.c")
                                    + f", one full frame of the same stream (clear + rasterize_* + resolve, no SVG walk) in {cpu_s * 1e3:.0f} ms, "
                                      "output equals the HIP frame")},
    }


def profile_doc(scene, size, spp, world):
    """(doc, why_not): the committed PMC reduction (profiles/*_traffic.json, tools/collect_profiles.sh) of THIS workload taken on
    THESE kernel sources - matched on scene / size / spp / GPUs and on the digest of csrc/ - or (None, reason)."""
    import glob
    sha = kernel_source_sha()
    seen = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        wl = doc.get("workload", {})
        if (wl.get("scene"), wl.get("size"), wl.get("spp_per_step"), wl.get("n_gpus")) != (scene, size, spp, world):
            continue
        if doc.get("kernel_source_sha16") == sha:
            doc["_file"] = os.path.relpath(path, ROOT)
            return doc, None
        seen = os.path.relpath(path, ROOT)
    if seen:
        return None, f"{seen} was taken on other kernel sources (csrc digest now {sha}): counters not reported"
    return None, "no committed PMC pass for this workload"


def valu_roofline(doc, kernel_ms):
    """Vector-instruction issue, the resource that binds the wave kernel: wave-instructions per launch (SQ_INSTS_VALU of the
    committed PMC pass) / the live launch time, against (i) the guide's rate - one wave64 VALU op per 2 cycles per SIMD at
    2.4 GHz - and (ii) the issue interval a SIMD was measured to sustain (tools/ubench/pk_rate.hip, 1.09 ns)."""
    sq = (doc or {}).get("sq_per_launch") or {}
    if "SQ_INSTS_VALU" not in sq or not kernel_ms:
        return None
    rate = sq["SQ_INSTS_VALU"] / (kernel_ms * 1e-3)
    return {"wave_instructions_per_launch": sq["SQ_INSTS_VALU"], "achieved_per_s": rate, "peak_guide_per_s": VALU_PEAK_GUIDE,
            "frac_guide": rate / VALU_PEAK_GUIDE, "issue_ns_per_simd_ubench": VALU_NS_UBENCH,
            "frac_ubench": rate / (SIMDS / (VALU_NS_UBENCH * 1e-9)),
            "source": f"{doc['_file']} (rocprofv3 --pmc SQ_INSTS_VALU ...), MI355X_MICROARCH.md, tools/ubench/pk_rate.hip"}


def pt_roofline(cnt, rays_per_launch, kernel_ms, doc, why_not, kernel, scratch_bytes_per_launch=None, extra=None):
    """The `roofline` object of a path-tracer workload.  SURVEY.md section 8(d)'s figure: algorithmic bytes per ray (32 B per node visited +
    128 B per object entered + 36 B per triangle tested + 4 B per sphere + 36 B hit normals, the counts being the reference
    traversal's own averages from an instrumented launch) x rays per launch / the kernel's launch time, against the HBM peak.
    For a cache-resident scene that figure is not a bound (it exceeds the peak); what binds the kernel - vector-instruction
    issue - is reported from the PMC pass of the same sources when one is committed, and is then the headline fraction."""
    bpr = bytes_per_ray(cnt)
    alg_bytes = bpr * rays_per_launch + (scratch_bytes_per_launch or 0.0)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "algorithmic_bytes_per_ray": bpr, "ray_state_bytes_per_launch": scratch_bytes_per_launch,
           "frac_of_l2_peak": achieved / L2_PEAK_GBS, "l2_peak": L2_PEAK_GBS,
           "note": "SURVEY.md 8(d): algorithmic bytes (scene reads by the reference's traversal counts + ray state spilled to memory) / "
                   "kernel time; the scene is served by the scalar cache / L2 / Infinity Cache, so against the HBM peak this may exceed 1 by construction"}
    valu = valu_roofline(doc, kernel_ms)
    traffic = (doc or {}).get("hbm_bytes_per_launch")
    out = dict(hbm)
    if valu:
        out = {"bound": "valu", "achieved": valu["achieved_per_s"] / 1e9, "peak": VALU_PEAK_GUIDE / 1e9, "unit": "G wave-instructions/s",
               "frac": valu["frac_guide"], "frac_ubench_ceiling": valu["frac_ubench"], "valu": valu, "hbm": hbm}
    out.update({"traffic": traffic, "kernel": kernel, "kernel_ms": kernel_ms,
                "per_ray": {k: cnt[k] / cnt["rays"] for k in cnt if k != "rays"}})
    if traffic is not None:
        out["traffic_frac_of_hbm_peak"] = traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["traffic_over_algorithmic_scratch"] = (traffic / scratch_bytes_per_launch) if scratch_bytes_per_launch else None
        out["l2_hit_rate"] = (doc["TCC_HIT_sum"] / (doc["TCC_HIT_sum"] + doc["TCC_MISS_sum"])) if doc.get("TCC_HIT_sum") else None
    # the same story as SCALARS on the roofline object itself (a record that keeps only one level of the line still tells it)
    out["algorithmic_bytes_per_ray"] = bpr
    out["hbm_algorithmic_frac"] = hbm["frac"]                                  # SURVEY.md 8(d)'s figure over the HBM peak (may exceed 1, see note)
    out["valu_frac"] = valu["frac_guide"] if valu else None                     # vector-instruction issue over the guide's rate: what binds
    out["counter_hbm_frac"] = out.get("traffic_frac_of_hbm_peak")               # rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE over the HBM peak
    out["traffic_over_algorithmic"] = out.get("traffic_over_algorithmic_scratch")   # counter bytes / the ray state the kernels must move
    if why_not:
        out["counters"] = why_not
    if extra:
        out.update(extra)
    return out


def ray_state_bytes(rays_per_launch, cams_per_launch, stream_counters=None):
    """SURVEY.md 8(d)'s last term - ray state that goes through memory per launch - and how it is made up.  Every kernel form: a 32-byte
    record per shaded bounce, written once and read once (a bounce = three rays issued by the reference), and one 16-byte store per
    sample, read again by the ordered reduction.  The streamed forms add what passes between their kernels, from DEVICE counters of the
    launch (srt_pt_stream_counters): the saved path state per alive slot and generation and the ray / list / hit traffic per queued
    entry, at the byte counts the kernels' layout has."""
    bounces = max(0.0, (rays_per_launch - cams_per_launch) / 3.0)
    terms = {"bounce_records": 64.0 * bounces, "sample_stores_and_reduction": 32.0 * cams_per_launch}
    if stream_counters:
        terms["slot_state"] = float(stream_counters["alive_slot_generations"] * stream_counters["bytes_per_alive_slot_generation"])
        terms["queued_entries"] = float(stream_counters["entries_queued"] * stream_counters["bytes_per_queued_entry"])
        terms["device_counters"] = {k: stream_counters[k] for k in ("entries_queued", "alive_slot_generations")}
    return sum(v for v in terms.values() if isinstance(v, float)), terms


def cfg5_bench(device, args, steps=4):
    """BASELINE configs[4]'s workload on one GPU as a second object of the line: the Cornell box with a 131 072-triangle glass
    mesh (80 127-node BVH<Triangle>, depth 18) and the mirror sphere, 1024 x 1024, 64 spp per step.  The streamed sweeps
    (kernel mode 7): wave-uniform sweeps in the logic kernel, the mesh's walks queued to the persistent ray-cast kernel."""
    import hashlib

    import torch

    import srt_amd
    from soft_rendering_toolsets_amd import scenes

    W = H = args.size
    spp = args.spp_per_step
    scene = scenes.cornell_with_mesh(7, "glass")
    pt = srt_amd.Pathtracer(device)
    pt.set_params(W, H, spp * steps, args.depth, True)
    t0 = time.perf_counter()
    pt.build_scene(scene)
    build_s = time.perf_counter() - t0
    pt.set_camera(scene["camera"])
    pt.set_tiling(32, 32, 0, 1)
    form = pt.kernel_form()
    _, per_rank, fpt = pt.tile_info()
    dev = torch.device("cuda", device)
    # consecutive epochs alternate between two streams, as in the headline run (each with its own buffers; the running mean
    # stays in epoch order through an event)
    nstreams = 1 if args.no_overlap else 2
    streams = [torch.cuda.current_stream()] if nstreams == 1 else [torch.cuda.Stream(device=dev) for _ in range(2)]
    tiles = [torch.zeros(per_rank * fpt, dtype=torch.float32, device=dev) for _ in range(nstreams)]
    image = [torch.zeros(W * H * 3, dtype=torch.float32, device=dev) for _ in range(nstreams)]
    acc = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)
    acc_done = [None]

    def step(i):
        k = i % nstreams
        S = streams[k]
        with torch.cuda.stream(S):
            pt.render_epoch_device(S.cuda_stream, args.seed, i * spp, spp, tiles[k].data_ptr())
            pt.untile_device(S.cuda_stream, tiles[k].data_ptr(), image[k].data_ptr())
            if acc_done[0] is not None:
                S.wait_event(acc_done[0])
            pt.accumulate_device(S.cuda_stream, acc.data_ptr(), image[k].data_ptr(), image[k].numel(), i + 1)
            ev = torch.cuda.Event()
            ev.record(S)
            acc_done[0] = ev

    for i in range(nstreams):                               # warm-up: both streams' scratch exists before the clock starts
        step(i)
    torch.cuda.synchronize()
    acc.zero_()
    pt.ray_count(reset=True)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rays, cams = pt.ray_count(reset=True)
    # the same steps on ONE stream (no overlap of consecutive epochs): what a step costs on its own, wall clock
    no_overlap_ms = None
    if nstreams > 1:
        t1 = time.perf_counter()
        for i in range(steps):
            pt.render_epoch_device(streams[0].cuda_stream, args.seed, i * spp, spp, tiles[0].data_ptr())
        torch.cuda.synchronize()
        no_overlap_ms = (time.perf_counter() - t1) * 1e3 / steps
        pt.ray_count(reset=True)
    # one more epoch on its own for the kernels' own durations (HIP events inside the library; not part of `value`)
    pt.kernel_time(enable=True)
    if form >= 3:
        pt.stream_times(enable=True)
        pt.stream_counters(reset=True)
    pt.render_epoch_device(streams[0].cuda_stream, args.seed, steps * spp, spp, tiles[0].data_ptr())
    torch.cuda.synchronize()
    rays_1, cams_1 = pt.ray_count(reset=True)
    ms_total, launches = pt.kernel_time(enable=False)
    kernel_ms = ms_total / max(1, launches)
    split = sc = None
    if form >= 3:
        ms3, gens = pt.stream_times(enable=False)
        split = dict(ms3)
        split["generations_enqueued_per_step"] = gens
        sc = pt.stream_counters(reset=True)
    rng = np.random.default_rng(1)
    n = 1 << 15
    xs, ys = rng.integers(0, W, n).astype(np.uint32), rng.integers(0, H, n).astype(np.uint32)
    pt.trace_samples(args.seed, xs, ys, rng.integers(0, spp * steps, n).astype(np.uint32))
    cnt = pt.counters()
    doc, why_not = profile_doc("cfg5", W, spp, 1)
    scratch, scratch_terms = ray_state_bytes(rays / steps, cams / steps, sc)
    out = {
        "metric": "Mrays/s", "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": 1, "steps": steps, "ms_per_step": elapsed * 1e3 / steps,
        "ms_per_step_no_overlap": no_overlap_ms,
        "config": {"workload": f"Scotty3D Pathtracer: Cornell box + 131 072-triangle glass mesh + mirror sphere (BASELINE configs[4] on one GPU; seeded "
                               f"procedural stand-in for the Stanford dragon, a missing blob of the reference checkout), {W}x{H}, {spp} spp per step, "
                               f"depth {args.depth}, BVH on", "triangles": 131072, "bvh_nodes": 80127, "kernel_form": form},
        "host_scene_build_s": build_s,
        "camera_samples_per_s": cams / elapsed, "rays": rays, "rays_per_camera_sample": rays / max(1, cams),
        "image_sha256_16": hashlib.sha256(acc.cpu().numpy().tobytes()).hexdigest()[:16],
        "roofline": pt_roofline(cnt, rays / steps, kernel_ms, doc, why_not,
                                "pt_wave_kernel<.., 4, .., 1> (resolve) + <.., 2> (probe) + pt_compact_kernel + pt_cast_kernel (every kernel of one epoch)" if form == 4 else str(form),
                                scratch, {"stream_kernels_ms": split, "ray_state_terms": scratch_terms,
                                          "working_set": "2.8 MB of interior records + 4.7 MB of triangles as the cast kernel reads them (36-byte records; + 6.3 MB padded records and 6.3 MB normals for shading): L2 (4 MiB per XCD) / Infinity Cache resident"}),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = pt_cpu_baseline(scene, W, H, args.depth, args.seed, pt, budget_s=8.0)
    pt.close()
    return out


def dropin_pt_bench(device, args, samples=256):
    """The PT::Pathtracer CLASS end to end: integration/_build/libdropin_pt_full.so - pathtracer_hip.cpp +
    pathtracer_core.cpp compiled inside the reference's scene layer - renders the Cornell box (five walls, area light, mirror and
    glass sphere, as Scene_Object instances) through set_params / begin_render / poll in_progress() / get_output(), 1024 x 1024,
    `samples` spp, the reference's epoch arithmetic for this host's thread count.  Reported next to it: the SAME scene (the
    harness's dump, read the way the reference's build_scene reads it) through the C ABI from Python with the headline's step -
    64-spp epochs alternating on two streams, elision on as in the class - and the class's image against that scene rendered
    epoch by epoch (the reference's epochs, running mean on the host)."""
    import torch

    import _harness as H
    import srt_amd

    path = os.path.join(ROOT, "integration", "_build", "libdropin_pt_full.so")
    if not os.path.exists(path):
        return {"value": None, "why": "integration/_build/libdropin_pt_full.so is built in the authoring container (make -C integration)"}
    del device
    srt_amd.load_library()
    lib = ctypes.CDLL(path)
    W = Hh = args.size
    threads = cpu_cores()
    rgb = np.zeros((Hh, W, 3), np.float32)
    cam = np.zeros(18, np.float32)
    dump = np.zeros(1 << 20, np.uint8)
    n = ctypes.c_uint64(0)
    out = (ctypes.c_double * 8)()
    rc = lib.dropin_pt_full_bench(3, W, Hh, samples, args.depth, 1, threads, ctypes.c_uint64(args.seed), H.P(rgb), H.P(cam), H.P(dump),
                                  ctypes.c_uint64(dump.size), ctypes.byref(n), out)
    if rc != 0:
        return {"value": None, "why": f"dropin_pt_full_bench returned {rc}"}
    render_s, build_s, epochs, rays, elided, logged, wall_s = (out[i] for i in range(7))
    scene = H.parse_scene_dump(dump[: n.value].tobytes())
    scene["camera"] = {"iview": cam[:16].copy(), "vfov": float(cam[16]), "ar": float(cam[17])}
    # the same scene through the C ABI, the headline's way
    pt = srt_amd.Pathtracer(0)
    pt.set_params(W, Hh, samples, args.depth, True)
    pt.build_scene(scene)
    pt.set_camera(scene["camera"])
    pt.set_elision(True)
    pt.set_tiling(32, 32, 0, 1)
    _, per_rank, fpt = pt.tile_info()
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    tiles = [torch.zeros(per_rank * fpt, dtype=torch.float32, device=dev) for _ in range(2)]
    spp = args.spp_per_step
    steps = max(2, samples // spp)
    for i in range(2):
        pt.render_epoch_device(streams[i].cuda_stream, args.seed, i * spp, spp, tiles[i].data_ptr())
    torch.cuda.synchronize()
    pt.ray_count(reset=True)
    t0 = time.perf_counter()
    for i in range(steps):
        pt.render_epoch_device(streams[i % 2].cuda_stream, args.seed, i * spp, spp, tiles[i % 2].data_ptr())
    torch.cuda.synchronize()
    abi_s = time.perf_counter() - t0
    abi_rays, _ = pt.ray_count(reset=True)
    # the class's image against the reference's epoch scheme on that scene, epoch by epoch through the C ABI
    spe = max(1, samples // (threads * 10))
    acc = np.zeros((Hh, W, 3), np.float32)
    k = s0 = 0
    while s0 < samples:
        take = min(spe, samples - s0)
        k += 1
        e = pt.render_epoch(args.seed, s0, take)
        acc += ((e - acc) * (np.float32(1.0) / np.float32(k))).astype(np.float32)
        s0 += take
    pt.close()
    value, abi_value = rays / render_s / 1e6, abi_rays / abi_s / 1e6
    return {"metric": "Mrays/s", "value": value, "unit": "Mrays/s", "render_s": render_s, "build_scene_s": build_s, "wall_s_begin_render_to_done": wall_s,
            "samples": samples, "threads_for_the_epoch_arithmetic": threads, "samples_per_epoch": spe, "epochs": int(epochs),
            "rays": int(rays), "rays_not_traced_dead_ray_elision": int(elided), "log_ray_calls_delivered_to_the_gui": int(logged),
            "c_abi_value_same_scene": abi_value, "fraction_of_c_abi": value / abi_value,
            "get_output_equals_epoch_by_epoch_c_abi_render_bit_for_bit": bool(np.array_equal(rgb.view(np.uint32), acc.view(np.uint32))),
            "config": {"workload": f"PT::Pathtracer (drop-in class inside the reference's scene layer): Cornell box as 8 Scene_Objects, {W}x{Hh}, {samples} spp, "
                                   f"depth {args.depth}, BVH on; rays = the reference's scene.hit calls (dead-ray elision on, as the class sets it)"},
            "what": "value = the reference's scene.hit calls of the render / completion_time().second (begin_render's worker: launches of <= 64 spp on two "
                    "lanes, the reference's epoch means and running mean folded on the device, nothing copied per epoch); c_abi_value_same_scene = "
                    "srt_pt_render_epoch_device from Python, 64-spp epochs alternating on two streams, same scene, elision on"}


def group_bench(args):
    """`--group N[,M,..]`: the in-process multi-GPU path the C++ drop-in uses (srt_pt_create_multi: a context per rank, image tiles
    round-robin, ONE gather per epoch to rank 0) - timed as srt_pt_group_render_epoch_device, per N.  With N devices it is the
    RCCL path of csrc/pt_group.cpp (grouped ncclGather over xGMI); on a box with fewer devices the ranks share device 0 (a
    REHEARSAL: the ranks' kernels run concurrently on one GPU, the gather is device-to-device copies, and the line says so).
    Per N: wall ms per step, Mrays/s, the image hash against N = 1, each rank's own kernel time and the exchange step
    (gather + un-tiling incl. waiting for the slowest rank) from HIP events, and the roofline object of rank 0's kernel."""
    import hashlib

    import torch

    import srt_amd
    from soft_rendering_toolsets_amd import scenes

    W = H = args.size
    spp = args.spp_per_step
    steps = max(2, min(args.steps, 8))
    scene = scenes.cornell_with_mesh(7, "glass") if args.scene == "cfg5" else scenes.cornell_box(args.scene)
    ndev = torch.cuda.device_count()
    rows, base_sha = [], None
    for n in [int(x) for x in args.group.split(",")]:
        devices = list(range(n)) if ndev >= n else [0] * n
        grp = srt_amd.PathtracerGroup(devices)
        grp.set_params(W, H, spp * steps, args.depth, True)
        grp.build_scene(scene)
        grp.set_camera(scene["camera"])
        grp.render_epoch_device(args.seed, 0, spp)                 # warm-up: scratch exists before the clock starts
        torch.cuda.synchronize()
        for d in set(devices):
            torch.cuda.synchronize(d)
        grp.ray_count(reset=True)
        for m in grp.members:
            m.kernel_time(enable=True)
        grp.gather_time(enable=True)
        t0 = time.perf_counter()
        for i in range(steps):
            d_img, stream = grp.render_epoch_device(args.seed, i * spp, spp)
        for d in set(devices):
            torch.cuda.synchronize(d)
        elapsed = time.perf_counter() - t0
        rays, cams = grp.ray_count(reset=True)
        kms = [m.kernel_time(enable=False) for m in grp.members]
        gms, gn = grp.gather_time(enable=False)
        img = grp.render_epoch(args.seed, 0, spp)                   # the epoch image of sample 0.. (host copy) for the hash
        full_sha = hashlib.sha256(img.tobytes()).hexdigest()
        sha = full_sha[:16]
        base_sha = base_sha or sha
        want = None
        if (args.scene, W, args.depth, args.seed, spp) == ("cbox", 1024, 8, 0, 64):
            import _harness as harness                   # (H is the image height here)
            want = harness.load_fullsize().get("cfg4_cbox_1024_64spp")
        # rank 0's kernel against the issue ceiling / SURVEY.md 8(d)'s byte figure, as in the headline (counters only at N = 1)
        rng = np.random.default_rng(1)
        k = 1 << 14
        xs, ys = rng.integers(0, W, k).astype(np.uint32), rng.integers(0, H, k).astype(np.uint32)
        grp.members[0].set_tiling(32, 32, 0, 1)
        grp.members[0].trace_samples(args.seed, xs, ys, rng.integers(0, spp, k).astype(np.uint32))
        cnt = grp.members[0].counters()
        doc, why_not = profile_doc(args.scene, W, spp, 1) if n == 1 else (None, "the committed counter passes are one-GPU runs")
        rank0_ms = kms[0][0] / max(1, kms[0][1])
        scratch, _ = ray_state_bytes(rays / steps / n, cams / steps / n)
        rows.append({
            "n_ranks": n, "devices": devices, "distinct_devices": len(set(devices)), "uses_rccl": grp.uses_rccl(),
            "rehearsal_on_one_gpu": len(set(devices)) < n,
            "ms_per_step": elapsed * 1e3 / steps, "value": rays / elapsed / 1e6, "unit": "Mrays/s", "steps": steps,
            "image_sha256_16": sha, "image_equals_n1": sha == base_sha,
            "image_equals_golden": bool(want and full_sha == want["sha256"]) if want else None,   # the reference build's image of this epoch (tests/golden/pt_fullsize.json)
            "rccl_communicators": n if grp.uses_rccl() else 0,
            "rank_kernel_ms": [t / max(1, c) for t, c in kms], "exchange_ms": gms / max(1, gn),
            "roofline": pt_roofline(cnt, rays / steps / n, rank0_ms, doc, why_not, "rank 0's dominant kernel", scratch),
        })
        grp.close()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--scene", default="cbox", choices=["cbox", "cbox_lambertian", "cfg5"],
                    help="cfg5 = BASELINE configs[4] stand-in: the Cornell box with a 131 072-triangle glass mesh (the dragon is a missing blob)")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the extra object that measures BASELINE configs[4]'s workload")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-raster", action="store_true")
    ap.add_argument("--no-golden-check", action="store_true", help="skip the extra epoch whose image is compared with the reference-built golden (profiling passes)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the object that times the PT::Pathtracer class end to end")
    ap.add_argument("--no-elision", action="store_true", help="skip the extra pass that measures dead-ray elision")
    ap.add_argument("--no-overlap", action="store_true", help="every step on one stream (no overlap of consecutive launches)")
    ap.add_argument("--group", default=None, metavar="N[,M..]",
                    help="time the in-process multi-GPU path (srt_pt_create_multi: what the C++ drop-in uses) for these rank counts and print "
                         "one JSON line {\"group\": [...]}; ranks share device 0 where the box has fewer GPUs (a rehearsal, labelled)")
    args = ap.parse_args()
    if args.group:
        print(json.dumps({"metric": "Mrays/s", "mode": "in-process group (srt_pt_group_render_epoch_device)", "scene": args.scene,
                          "group": group_bench(args)}), flush=True)
        return

    import torch
    import torch.distributed as dist

    import srt_amd
    from soft_rendering_toolsets_amd import scenes
    from soft_rendering_toolsets_amd.dist import TileShard, gather_tiles

    world = int(os.environ.get("WORLD_SIZE", "1")) if args.gpus > 1 else 1
    rank = int(os.environ.get("RANK", "0")) if world > 1 else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # Rehearsal mode for a 1-GPU box: SRT_BENCH_REHEARSE=1 runs the N ranks on cuda:0 over gloo (tile
    # buffers staged through host memory for the gather).  Never used for reported numbers.
    rehearse = os.environ.get("SRT_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W = H = args.size
    spp = args.spp_per_step
    scene = scenes.cornell_with_mesh(7, "glass") if args.scene == "cfg5" else scenes.cornell_box(args.scene)
    pt = srt_amd.Pathtracer(local_rank)
    pt.set_params(W, H, spp * args.steps, args.depth, True)
    pt.build_scene(scene)
    pt.set_camera(scene["camera"])
    pt.set_tiling(32, 32, rank, world)
    form = pt.kernel_form()
    kernel_name = {0: "pt_wave_kernel", 1: "pt_wave_kernel", 2: "pt_wave_kernel", 3: "pt_wave_kernel<.., 3, ..> + pt_cast_kernel",
                   4: "pt_wave_kernel<.., 4, ..> + pt_cast_kernel", -1: "pt_unit_kernel", -2: "pt_epoch_kernel"}[form]
    local_tiles, per_rank, fpt = pt.tile_info()
    per_rank_floats = per_rank * fpt
    shard = TileShard(W, H, 32, 32, rank, world)
    assert (local_tiles, per_rank, fpt) == (len(shard.local), shard.tiles_per_rank, shard.floats_per_tile)

    dev = torch.device("cuda", local_rank)
    # Two launch streams, used alternately: the persistent kernel of step i+1 fills the CUs that the tail of step i
    # leaves idle (a 1/8 image shard: 8.9 -> 8.2 ms per step, tools/overlap_bench.py).  Each stream has its own tile /
    # gather / image buffers (and the library keeps one set of epoch scratch per stream); the running-mean
    # accumulate is order dependent, so it waits for the previous step's accumulate through an event.
    # The streamed forms (cfg5) gain as well: one epoch's generations leave gaps at every kernel boundary that the other
    # epoch's kernels fill (359.8 -> 341.8 ms per step, same image).
    nstreams = 1 if (rehearse or args.no_overlap) else int(os.environ.get("SRT_BENCH_STREAMS", "2"))
    streams = [torch.cuda.current_stream()] if nstreams == 1 else [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    tiles = [torch.zeros(per_rank * fpt, dtype=torch.float32, device=dev) for _ in range(nstreams)]
    gathered = [torch.zeros(world * per_rank * fpt, dtype=torch.float32, device=dev) if (rank == 0 and world > 1) else None
                for _ in range(nstreams)]
    image = [torch.zeros(W * H * 3, dtype=torch.float32, device=dev) if rank == 0 else None for _ in range(nstreams)]
    acc = torch.zeros(W * H * 3, dtype=torch.float32, device=dev) if rank == 0 else None
    acc_done = [None]
    kernel_events = []
    gather_events = []

    def step(i, timed):
        k = i % nstreams
        S = streams[k]
        with torch.cuda.stream(S):
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(S)
            pt.render_epoch_device(S.cuda_stream, args.seed, i * spp, spp, tiles[k].data_ptr())
            if timed:
                e1.record(S)
                kernel_events.append((e0, e1))
            if world > 1 and rehearse:
                host = tiles[k].cpu()
                ghost = torch.zeros(world * host.numel()) if rank == 0 else None
                gather_tiles(host, ghost, world, rank)
                if rank == 0:
                    gathered[k].copy_(ghost)
            elif world > 1:
                gather_tiles(tiles[k], gathered[k], world, rank)  # one RCCL gather over xGMI: tile radiance -> rank 0
            if timed and world > 1 and rank == 0:
                e2 = torch.cuda.Event(enable_timing=True)          # (e1 .. e2 on rank 0's stream: the exchange incl. waiting for the slowest rank)
                e2.record(S)
                gather_events.append((e1, e2))
            if rank == 0:
                src = gathered[k] if world > 1 else tiles[k]
                pt.untile_device(S.cuda_stream, src.data_ptr(), image[k].data_ptr())
                if acc_done[0] is not None:
                    S.wait_event(acc_done[0])                     # running mean: epoch i after epoch i - 1
                pt.accumulate_device(S.cuda_stream, acc.data_ptr(), image[k].data_ptr(), image[k].numel(), i + 1)
                ev = torch.cuda.Event()
                ev.record(S)
                acc_done[0] = ev

    if form >= 3 and args.warmup < nstreams:
        args.warmup = nstreams        # the streamed forms allocate ~1.5 GB of path state per stream on first use: not inside the clock
    for i in range(args.warmup):
        step(i, False)
    torch.cuda.synchronize()
    pt.ray_count(reset=True)
    if rank == 0:
        acc.zero_()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    rays, cams = pt.ray_count()
    # One more step THROUGH THE SAME PATH (tiles, gather, un-tiling), outside the clock: seed 0, samples 0 .. 63 - the epoch whose
    # image the reference build rendered for tests/golden/pt_fullsize.json (cfg4_cbox_1024_64spp).  Every `--gpus N` run says
    # whether its N ranks and its collective reproduce that image bit for bit.
    golden = None
    gkey = "cfg4_cbox_1024_64spp"
    if args.scene == "cbox" and (W, args.depth) == (1024, 8) and not args.no_golden_check:
        import hashlib

        import _harness as harness                     # (H is the image height here)
        want = harness.load_fullsize().get(gkey)
        S0 = streams[0]
        with torch.cuda.stream(S0):
            pt.render_epoch_device(S0.cuda_stream, 0, 0, 64, tiles[0].data_ptr())
            if world > 1 and rehearse:
                host = tiles[0].cpu()
                ghost = torch.zeros(world * host.numel()) if rank == 0 else None
                gather_tiles(host, ghost, world, rank)
                if rank == 0:
                    gathered[0].copy_(ghost)
            elif world > 1:
                gather_tiles(tiles[0], gathered[0], world, rank)
            if rank == 0:
                pt.untile_device(S0.cuda_stream, (gathered[0] if world > 1 else tiles[0]).data_ptr(), image[0].data_ptr())
        torch.cuda.synchronize()
        if rank == 0:
            sha = hashlib.sha256(image[0].cpu().numpy().tobytes()).hexdigest()
            golden = {"image_sha256": sha, "golden": gkey, "golden_sha256": want["sha256"] if want else None,
                      "image_equals_golden": bool(want and sha == want["sha256"]),
                      "golden_from": "the reference's PT::Pathtracer compiled from /root/reference (tests/golden/make_pt_fullsize_golden.py), seed 0, samples 0..63"}
        pt.ray_count(reset=True)
    # the same steps on ONE stream: a step on its own, wall clock (the timed region above overlaps consecutive launches on two streams)
    no_overlap_ms = None
    if nstreams > 1 and world == 1:
        pt.ray_count(reset=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            pt.render_epoch_device(streams[0].cuda_stream, args.seed, i * spp, spp, tiles[0].data_ptr())
        torch.cuda.synchronize()
        no_overlap_ms = (time.perf_counter() - t1) * 1e3 / args.steps
    pt.kernel_time(enable=True)
    sc = None
    if form >= 3:
        pt.stream_times(enable=True)
        pt.stream_counters(reset=True)
    # the dominant kernel on its own (HIP events inside the library, on the launch stream): two launches after the
    # timed region, nothing else in flight - inside the timed region consecutive launches overlap by design
    for j in range(2):
        pt.render_epoch_device(streams[0].cuda_stream, args.seed, j * spp, spp, tiles[0].data_ptr())
    torch.cuda.synchronize()
    wave_ms_total, wave_launches = pt.kernel_time(enable=False)
    kernel_ms = wave_ms_total / max(1, wave_launches)          # agrees with rocprofv3 AverageNs of a non-overlapped run
    stream_split = None
    if form >= 3:                                              # the streamed forms: the epoch's kernels one by one
        ms3, gens = pt.stream_times(enable=False)
        stream_split = {k: v / 2 for k, v in ms3.items()}
        stream_split["generations_enqueued_per_step"] = gens // 2
        sc = pt.stream_counters(reset=True)
        sc["entries_queued"] //= 2; sc["alive_slot_generations"] //= 2        # (two launches were counted)
    pt.ray_count(reset=True)
    epoch_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / max(1, len(kernel_events))  # per step, overlapped
    if world > 1:
        cdev = torch.device("cpu") if rehearse else dev
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_max = float(t[0]), float(t[1])
        c = torch.tensor([rays, cams], dtype=torch.int64, device=cdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_rays, total_cams = int(c[0]), int(c[1])
        rank_stats = [torch.zeros(3, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(rank_stats, torch.tensor([kernel_ms, epoch_ms, float(rays)], dtype=torch.float64, device=cdev))
        rank_kernel_ms = [float(t[0]) for t in rank_stats]
        rank_step_span_ms = [float(t[1]) for t in rank_stats]
        rank_rays = [int(t[2]) for t in rank_stats]
    else:
        kernel_ms_max, total_rays, total_cams = kernel_ms, rays, cams
        rank_kernel_ms, rank_step_span_ms, rank_rays = [kernel_ms], [epoch_ms], [rays]
    gather_ms = (sum(a.elapsed_time(b) for a, b in gather_events) / len(gather_events)) if gather_events else None

    main_sha = main_mean = None
    if rank == 0:
        import hashlib
        main_sha = hashlib.sha256(acc.cpu().numpy().tobytes()).hexdigest()
        main_mean = float(acc.mean().item())
    elision = None
    if world == 1 and not args.no_elision:
        # Second pass with dead-ray elision (include/srt_pt.h: srt_pt_set_elision): the headline above traces EVERY ray the
        # reference issues; here the wave kernel skips the BSDF-sampled direct ray whose term the reference adds and subtracts
        # again (legal, both ray counts reported).  Same steps, fresh accumulator, image compared.
        full_sha = main_sha
        pt.set_elision(True)
        acc_done[0] = None
        step(0, False)
        torch.cuda.synchronize()
        pt.ray_count(reset=True); pt.rays_elided(reset=True)
        acc.zero_()
        acc_done[0] = None
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(i, False)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        rays2, cams2 = pt.ray_count(reset=True)
        elided = pt.rays_elided(reset=True)
        elision = {
            "in_value": False, "ms_per_step": el2 * 1e3 / args.steps, "camera_samples_per_s": cams2 / el2,
            "reference_equivalent_mrays_per_s": rays2 / el2 / 1e6, "traced_mrays_per_s": (rays2 - elided) / el2 / 1e6,
            "rays_elided_fraction": elided / max(1, rays2),
            "image_equals_full_trace_bit_for_bit": hashlib.sha256(acc.cpu().numpy().tobytes()).hexdigest() == full_sha,
            "note": "two-ray batches: the BSDF-sampled direct ray of a Lambertian bounce is provably dead without delta / "
                    "environment lights ((0 + d) - d == +0); its random draws are kept, the ray is not traced",
        }
        pt.set_elision(False)

    if rank == 0:
        # traversal averages of this workload from an instrumented launch (outside the timed region)
        rng = np.random.default_rng(1)
        n = 1 << 16
        xs, ys = rng.integers(0, W, n).astype(np.uint32), rng.integers(0, H, n).astype(np.uint32)
        ss = rng.integers(0, spp * args.steps, n).astype(np.uint32)
        pt.trace_samples(args.seed, xs, ys, ss)
        cnt = pt.counters()
        rays_per_launch_rank0 = rays / args.steps
        cams_per_launch_rank0 = cams / args.steps
        mean_radiance = main_mean
        image_sha = main_sha[:16]
        prof_doc, why_not = profile_doc(args.scene, W, spp, world)
        # ray state that goes through memory per launch (SURVEY.md 8(d)'s last term): a 32-byte record per shaded bounce,
        # written once and read once, and one 16-byte store per sample (read again by the ordered reduction)
        scratch, scratch_terms = ray_state_bytes(rays_per_launch_rank0, cams_per_launch_rank0, sc)
        out = {
            "metric": "Mrays/s", "value": total_rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "ms_per_step_no_overlap": no_overlap_ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": (f"Scotty3D Pathtracer: Cornell box + 131 072-triangle glass mesh + mirror sphere (BASELINE configs[4]; seeded procedural "
                             f"stand-in for the Stanford dragon, a missing blob of the reference checkout), {W}x{H}, {spp} spp per step "
                             f"({args.steps} steps = {spp * args.steps} spp), depth {args.depth}, BVH on") if args.scene == "cfg5" else
                            (f"Scotty3D Pathtracer: Cornell box ({args.scene}: area light + MIS"
                             f"{', mirror + glass spheres' if args.scene == 'cbox' else ''}), {W}x{H}, {spp} spp per step "
                             f"({args.steps} steps = {spp * args.steps} spp; 32 steps = BASELINE configs[3] 2048 spp), depth {args.depth}, BVH on"),
                "tiles": "32x32 round-robin over ranks", "collective": ("one gloo gather of tile radiance per step (rehearsal: N ranks on one GPU, never a reported number)" if rehearse else
                               "one RCCL gather of tile radiance per step") if world > 1 else "none (1 GPU)",
                "seed": args.seed,
            },
            # what a multi-GPU record needs to verify and decompose itself: the epoch image against the reference-built golden, every
            # rank's own kernel time (one launch alone, after the timed region) and step span, the exchange step on rank 0
            "image_equals_golden": golden["image_equals_golden"] if golden else None, "golden_check": golden,
            "rank_kernel_ms": rank_kernel_ms, "rank_step_span_ms": rank_step_span_ms, "rank_rays": rank_rays,
            "gather_ms": gather_ms,
            "collective": {"backend": (dist.get_backend() if world > 1 else None), "ranks": (dist.get_world_size() if world > 1 else 1),
                           "is_rccl": bool(world > 1 and not rehearse), "bytes_per_step_to_rank0": (world - 1) * per_rank_floats * 4 if world > 1 else 0},
            "camera_samples_per_s": total_cams / elapsed, "rays": total_rays, "camera_samples": total_cams,
            "rays_per_camera_sample": total_rays / max(1, total_cams),
            "rays_counted": "every scene.hit the reference performs (no ray is elided)",
            "mean_radiance": mean_radiance, "image_sha256_16": image_sha,
            "roofline": pt_roofline(cnt, rays_per_launch_rank0, kernel_ms, prof_doc, why_not, kernel_name, scratch, {
                "kernel_ms_max_over_ranks": kernel_ms_max, "kernel_launches": wave_launches,
                "kernel_ms_is": "one launch on its own, after the timed region (the streamed forms: every kernel of one epoch); "
                                "`ms_per_step` is wall time of the timed region / steps, where consecutive launches overlap on two streams",
                "step_span_ms_on_its_stream": epoch_ms,   # start-to-end of a step on its own stream; the other stream's step shares the GPU meanwhile
                "stream_kernels_ms": stream_split, "ray_state_terms": scratch_terms,
            }),
        }
        if elision is not None:
            out["dead_ray_elision"] = elision
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = pt_cpu_baseline(scene, W, H, args.depth, args.seed, pt)
        if not args.no_cfg5 and world == 1 and args.scene != "cfg5":
            out["cfg5"] = cfg5_bench(local_rank, args)
            out["cfg5_value"], out["cfg5_ms_per_step"] = out["cfg5"]["value"], out["cfg5"]["ms_per_step"]
        if not args.no_dropin and world == 1 and args.scene == "cbox":
            out["dropin"] = dropin_pt_bench(local_rank, args)
            out["dropin_value"], out["dropin_fraction_of_c_abi"] = out["dropin"].get("value"), out["dropin"].get("fraction_of_c_abi")
        if not args.no_raster and world == 1:
            out["raster"] = raster_bench(local_rank)
            out["raster_value"], out["raster_wall_ms"] = out["raster"]["value"], out["raster"]["wall_ms_per_frame"]
            out["raster_c_abi_wall_ms"] = out["raster"]["c_abi_wall_ms"]["new_stream_every_frame"]
            out["raster_device_ms_per_frame"] = out["raster"]["device_ms_per_frame"]
        print(json.dumps(out), flush=True)
    pt.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
