#!/usr/bin/env python3
"""Instruction mix of the Cornell kernel (pt_wave_kernel<false, 0, false, 3>), for DESIGN.md section 2 / profiles/.

Two views that do not need each other:
  static   llvm-objdump of the built code object, the kernel's instructions classified by what they do.  The stamped build of the same
           kernel (<true, 0, false, 3>: s_memtime at the section boundaries) gives the same classification PER SECTION of the loop body
           (refill, top-down sweep, leaf objects, combine, finish direct light, terminate, shade - in code order).
  dynamic  (needs a GPU) the stamped build's own section cycle sums (srt_pt_section_cycles): the share of the wave cycles each section takes.
Static counts are not execution counts - the leaf section runs once per object, the sweeps once per node - but inside one section the mix is
what executes, and the IEEE sequences (v_div_scale / v_div_fmas / v_div_fixup per divide, v_sqrt + correction per root) can be read off.

usage: instruction_mix.py [--gpu] [out.json]"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

CLASSES = [
    ("divide (IEEE sequence)", r"v_div_scale_f32|v_div_fmas_f32|v_div_fixup_f32|v_rcp_f32"),
    ("square root (IEEE sequence)", r"v_sqrt_f32|v_rsq_f32"),
    ("fp64", r"_f64"),
    ("fp32 fma / mul / add", r"v_fma_f32|v_fmac_f32|v_mul_f32|v_add_f32|v_sub_f32|v_subrev_f32|v_mad_f32|v_pk_"),
    ("fp32 min / max / other", r"v_min|v_max|v_med3|v_floor|v_fract|v_trunc|v_rndne|v_frexp|v_ldexp|v_exp|v_log|v_sin|v_cos|v_ceil"),
    ("compare", r"v_cmp|v_cmpx"),
    ("select / move", r"v_cndmask|v_mov_b|v_readlane|v_readfirstlane|v_writelane|v_swap|v_accvgpr|v_perm|v_bfi"),
    ("convert", r"v_cvt"),
    ("integer / bit", r"v_and|v_or|v_xor|v_not|v_lshl|v_lshr|v_ashr|v_add_u|v_add_co|v_addc|v_sub_u|v_sub_co|v_subb|v_mul_lo|v_mul_hi|v_mul_u|v_mad_u|v_mad_i|v_bfe|v_add3|v_lshl_add|v_add_lshl|v_or3|v_and_or|v_xad|v_mbcnt|v_bcnt|v_ffb|v_alignbit|v_mul_i|v_sub_i|v_add_i|v_cmp_class"),
    ("LDS", r"ds_"),
    ("vector memory", r"global_|buffer_|flat_|scratch_"),
    ("scalar memory", r"s_load|s_buffer_load|s_store|s_memtime|s_dcache"),
    ("scalar ALU / branch", r"s_"),
]


def disassemble():
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(ROOT, "soft-rendering-toolsets_amd", "lib", "pt.hip.o")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={tmp}/fat.bin", obj], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={tmp}/fat.bin",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/k.hsaco"], check=True)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"{tmp}/k.hsaco"], check=True, capture_output=True, text=True).stdout


def kernel_body(text, mangled_part):
    lines, on = [], False
    for line in text.splitlines():
        if re.match(r"^[0-9a-f]+ <", line):
            on = mangled_part in line
            continue
        if on and line.strip():
            lines.append(line.strip().split("//")[0].strip())
    return lines


def classify(lines):
    out = collections.OrderedDict((name, 0) for name, _ in CLASSES)
    out["other"] = 0
    for ins in lines:
        op = ins.split()[0]
        for name, pat in CLASSES:
            if re.match(pat, op) or (name == "fp64" and re.search(pat, op)):
                out[name] += 1
                break
        else:
            out["other"] += 1
    return out


def main():
    gpu = "--gpu" in sys.argv
    outs = [a for a in sys.argv[1:] if not a.startswith("--")]
    text = disassemble()
    plain = kernel_body(text, "pt_wave_kernelILb0ELi0ELb0ELi3ELi0E")
    stamped = kernel_body(text, "pt_wave_kernelILb1ELi0ELb0ELi3ELi0E")
    doc = {"kernel": "pt_wave_kernel<false, 0, false, 3> (the Cornell kernel, BASELINE configs[2-3])",
           "static_instructions": len(plain), "static_by_class": classify(plain),
           "ieee_divides_static": sum(1 for l in plain if l.startswith("v_div_fixup_f32")),
           "ieee_square_roots_static": sum(1 for l in plain if l.startswith("v_sqrt_f32")),
           "method": "llvm-objdump -d of lib/pt.hip.o (gfx950 code object); classes by mnemonic, first match in tools/instruction_mix.py:CLASSES"}
    # the stamped build, cut at its s_memtime instructions: the loop body's sections in code order
    names = ["prologue (before the loop)", "refill", "top-down sweep", "leaf objects", "combine (visit rule, Trace::min)", "finish the bounce's direct light",
             "terminate (fold the records, store the sample)", "shade (next bounce batch)", "epilogue"]
    cuts = [i for i, l in enumerate(stamped) if l.startswith("s_memtime")]
    sections, prev = [], 0
    for k, c in enumerate(cuts + [len(stamped)]):
        sections.append({"section": names[k] if k < len(names) else f"section {k}", "static_instructions": c - prev,
                         "static_by_class": {a: b for a, b in classify(stamped[prev:c]).items() if b}})
        prev = c
    doc["stamped_build_sections_in_code_order"] = sections
    doc["stamped_build_note"] = ("pt_wave_kernel<true, 0, false, 3>, cut at its s_memtime instructions; the compiler may move code across a stamp, and the leaf "
                                 "section's code runs once per object of the leaf / the sweeps' once per node: a guide to what each section is made of, not a count")
    if gpu:
        sys.path.insert(0, ROOT)
        import srt_amd
        from soft_rendering_toolsets_amd import scenes

        scene = scenes.cornell_box("cbox")
        pt = srt_amd.Pathtracer(0)
        pt.set_params(1024, 1024, 64, 8, True)
        pt.build_scene(scene); pt.set_camera(scene["camera"])
        pt.set_kernel(3)
        pt.render_epoch(0, 0, 16)
        pt.section_cycles(reset=True); pt.ray_count(reset=True)
        pt.render_epoch(0, 0, 64)
        sec = pt.section_cycles()
        rays, _ = pt.ray_count()
        tot = float(sum(sec.values()))
        doc["dynamic_section_share_of_wave_cycles"] = {k: v / tot for k, v in sec.items()}
        doc["dynamic_note"] = f"kernel mode 3 (the stamped build), one 64-spp epoch of 1024 x 1024, {rays} rays; s_memtime deltas summed per wave"
        pt.close()
    js = json.dumps(doc, indent=1)
    if outs:
        open(outs[0], "w").write(js + "\n")
    print(js)


if __name__ == "__main__":
    main()
