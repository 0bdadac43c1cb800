#!/usr/bin/env python3
"""Static VALU instruction mix of the two kernels that own the benchmark step, from the gfx950 assembly hipcc emits for
the product's flags -- the weights of the mix-weighted ISSUE BOUND bench.py prints in `roofline_kernels` next to the
157.3 TFLOP/s vector peak.

    python tools/isa_mix.py            # writes profiles/isa_mix.json (runs here: hipcc cross-compiles without a GPU)

Why: the spec peak prices every wave64 VALU instruction at two cycles per SIMD.  tools/microbench/valu_rate.hip (one
inline-assembly block per instruction form; profiles/r03/valu_rate_forms.txt) measures what one SIMD of an MI355X
sustains, and it is a matter of the FORM, not of the number of VGPR operands (the first version of that microbenchmark
let the compiler choose the instructions and mislabelled half of its rows; the classes and costs used here until then
-- "one / several VGPR sources", packed at 6.2 cycles -- were wrong):

  fast            2.65 cycles  v_fma / v_fmac / v_add / v_sub / v_mul _f32, v_add / v_sub _u32, v_and / v_or / v_xor,
                               v_mov_b32 -- with VGPR, inline-constant or literal operands
  slow            4.2          everything else at "full" rate: ANY of the above with an SGPR operand, v_max / v_min
                               (2- and 3-input), compares, v_cndmask, conversions, floor, shifts, v_lshl_add, v_add3,
                               v_mad_*, v_mul_lo, v_bfe, v_perm, v_readlane, every DPP and SDWA form, every packed
                               (v_pk_*) instruction, v_mov_b64
  transcendental  8.2          v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos

A lone wave issues one VALU instruction per ~8 cycles whatever the form, so a SIMD needs two resident waves for the
slow class and three to four for the fast one; with two waves (the fused detection: 224 VGPRs) a slow instruction costs
4.6-5.7 cycles (it varies from run to run between those two values per form: issue arbitration, not the form) and a
fast one 2.8-3.2.  Two numbers per kernel therefore: `cycles_per_instruction_mix_weighted` prices the mix at the BEST
the hardware does for each class (eight waves) -- a bound whatever the occupancy -- and
`cycles_per_instruction_at_occupancy` at what the classes cost at the kernel's own occupancy.  The bound for a kernel
that retires N wave-instructions of a given mix on S SIMDs at clock f is  N * sum_c(frac_c * cycles_c) / (S * f)  (the
microbenchmark's cycles are time at the nominal 2.4 GHz with the whole chip loaded: a time calibration).

Classes are counted over the basic blocks (split at labels AND at the assembler's fall-through block comments) that
sit INSIDE A LOOP (between a label and a later backward branch to it): the prologue, the window fill and other
once-per-wave code do not weigh in.  For the fused detection that is still not the hot path: its loop holds 60 inlined
copies of the 122-instruction candidate refinement (cold: ~1 % of the dynamic instructions).  Its row step (x3
unrolled) is two kinds of block: the BLUR blocks (packed instructions: blur + DoG, then the ten-instruction threshold
pre-test) that every wave-row executes, and the ANALYSIS blocks (v_min3 / v_max3 trees of the 26-neighbour test) that
only the wave-rows run in which the pre-test finds a centre above the threshold.  They are reported separately
("mix" + "analysis") and bench.py weighs the second with the measured pass fraction of the content.  The count is
static -- every counted block weighs one -- so it is an estimate of the dynamic mix, not a trace; the PMC total it is
applied to is exact.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cusift_amd import build as B  # noqa: E402

KERNELS = {  # name -> (source, mangled-name needle, block selector)
    # the instantiations the benchmark's step launches since round 5 (heads to a list per octave, the next octave's image
    # as a by-product): <kIdent0 = true, 64, kDown = true> is octave 0, <false, 64, true> octaves 1 .. 3
    "detect_fused_kernel": ("sift_stencils.hip", "detect_fused_kernelILb1ELi64ELb1E", "detect"),
    "detect_fused_kernel<false>": ("sift_stencils.hip", "detect_fused_kernelILb0ELi64ELb1E", "detect"),
    "detect_fused_kernel<true, 588, false> (round 4's octave 0)": ("sift_stencils.hip", "detect_fused_kernelILb1ELi588ELb0E", "detect"),
    "describe_all_kernel": ("sift_keypoints.hip", "describe_all_kernel", "loop"),
    "laplace_multi_fast_kernel": ("sift_stencils.hip", "laplace_multi_fast_kernelILi2E", "loop"),
}
# cycles per wave-instruction per SIMD by class and resident waves (profiles/r03/valu_rate_forms.txt; "best" = 8 waves)
BEST = {"fast": 2.65, "slow": 4.2, "transcendental": 8.2}
CYCLES = {"best": BEST,
          2: {"fast": 3.0, "slow": 5.1, "transcendental": 8.7},   # slow: 4.6-5.7 by run, fast: 2.8-3.2
          4: {"fast": 2.75, "slow": 4.5, "transcendental": 8.3}}
WAVES = {"detect_fused_kernel": 2, "detect_fused_kernel<false>": 2,
         "detect_fused_kernel<true, 588, false> (round 4's octave 0)": 2, "describe_all_kernel": 4,
         "laplace_multi_fast_kernel": 4}
TRANSCENDENTAL = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
FAST = ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32")
DPP = re.compile(r"\b(row_shr|row_shl|row_ror|wave_shr|wave_shl|wave_ror|wave_rol|quad_perm|row_bcast|row_mirror|"
                 r"row_half_mirror|row_share|row_xmask|dpp8)\b")
SCALAR_OPERAND = re.compile(r"^-?\|?(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|src_\w+|ttmp\d+)")


def assembly(src):
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [B.find_hipcc()] + flags + ["-S", "--cuda-device-only", "-o", "-", os.path.join(B.CSRC, src)]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def kernel_body(asm, needle):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and l.rstrip().split(":")[0].startswith("_Z"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    return lines[start + 1:end]


def classify(line):
    t = line.strip()
    if not t.startswith("v_"):
        return None
    op = t.split()[0]
    if op.startswith(("v_readfirstlane", "v_writelane", "v_nop")):
        return None
    if any(op.startswith(q) for q in TRANSCENDENTAL):
        return "transcendental"
    if DPP.search(t) or "_dpp" in op or "_sdwa" in op:
        return "slow"
    base = re.sub(r"_e(32|64)$", "", op)
    if base not in FAST:
        return "slow"
    ops = t[len(op):].split(";")[0]
    srcs = [p.strip() for p in ops.split(",")][1:]
    return "slow" if any(SCALAR_OPERAND.match(p) for p in srcs) else "fast"


def loop_blocks(body):
    """Per basic block inside a loop: {class: count} plus the raw mnemonic counts."""
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            label_at[m.group(1)] = i
    in_loop = [False] * len(body)
    for i, l in enumerate(body):
        m = re.search(r"\bs_c?branch\w*\s+(?:\S+,\s*)?(\.LBB[0-9_]+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            for j in range(label_at[m.group(1)], i + 1):
                in_loop[j] = True
    starts = set([0] + list(label_at.values()))
    starts |= {i for i, l in enumerate(body) if l.lstrip().startswith("; %bb.")}
    starts = sorted(starts) + [len(body)]
    blocks, total_all = [], 0
    for a, b in zip(starts[:-1], starts[1:]):
        blk, ops = {}, {}
        for i in range(a, b):
            c = classify(body[i])
            if c is not None:
                total_all += 1
                if in_loop[i]:
                    blk[c] = blk.get(c, 0) + 1
                    op = body[i].split()[0]
                    ops[op] = ops.get(op, 0) + 1
        if blk:
            blocks.append((blk, ops))
    return blocks, total_all


def add(into, blk):
    for k, v in blk.items():
        into[k] = into.get(k, 0) + v


def loop_mix(body, selector):
    """Class counts over the loop blocks.  selector "loop": all of them; "detect": (blur blocks, analysis blocks)."""
    blocks, total_all = loop_blocks(body)
    main, ana, n_main, n_ana = {}, {}, 0, 0
    for blk, ops in blocks:
        if selector == "loop":
            add(main, blk)
            n_main += 1
        elif ops.get("v_pk_fma_f32", 0) >= 20:
            add(main, blk)
            n_main += 1
        elif ops.get("v_min3_f32", 0) >= 20 and not blk.get("transcendental"):
            add(ana, blk)
            n_ana += 1
        elif ops.get("v_max3_f32", 0) == 10 and sum(blk.values()) <= 16:  # the threshold pre-test: every row
            add(main, blk)
    return main, total_all, n_main, ana, n_ana


def main():
    out = {"_source": "python tools/isa_mix.py: static VALU class counts over the loop blocks of each kernel (hipcc -S with "
                      "the product's flags); cycles per class from profiles/r03/valu_rate_forms.txt "
                      "(tools/microbench/valu_rate.hip: one inline-assembly block per instruction form)",
           "_cycles_per_wave_instruction_per_simd": {str(k): v for k, v in CYCLES.items()}}
    cache = {}

    def priced(counts, table):
        n = sum(counts.values())
        return sum(counts[k] / n * table[k] for k in counts)

    for name, (src, needle, selector) in KERNELS.items():
        if src not in cache:
            cache[src] = assembly(src)
        counts, total_all, n_blocks, ana, n_ana = loop_mix(kernel_body(cache[src], needle), selector)
        n = sum(counts.values())
        own = CYCLES[WAVES[name]]
        frac = {k: round(v / n, 4) for k, v in sorted(counts.items())}
        out[name] = {"counted_valu_instructions_static": n, "counted_blocks": n_blocks,
                     "blocks": "blur + DoG + pre-test blocks (every wave-row)" if selector == "detect" else "all loop blocks",
                     "kernel_valu_instructions_static": total_all,
                     "waves_per_simd": WAVES[name], "mix": frac,
                     "cycles_per_instruction_mix_weighted": round(priced(counts, BEST), 3),
                     "cycles_per_instruction_at_occupancy": round(priced(counts, own), 3)}
        print("%-28s loop VALU %5d of %5d  %s  -> %.2f cycles/inst best, %.2f at %d waves" % (
            name, n, total_all, frac, priced(counts, BEST), priced(counts, own), WAVES[name]))
        if ana:
            na = sum(ana.values())
            out[name]["analysis"] = {"counted_valu_instructions_static": na, "counted_blocks": n_ana,
                                     "blocks": "26-neighbour analysis blocks (only wave-rows that pass the pre-test)",
                                     "mix": {k: round(v / na, 4) for k, v in sorted(ana.items())},
                                     "cycles_per_instruction_mix_weighted": round(priced(ana, BEST), 3),
                                     "cycles_per_instruction_at_occupancy": round(priced(ana, own), 3),
                                     "instructions_per_row_step": round(na / max(1, n_blocks), 1)}
            out[name]["instructions_per_row_step"] = round(n / max(1, n_blocks), 1)
            print("%-28s analysis  %5d in %d blocks %s -> %.2f best, %.2f at occupancy" % (
                "", na, n_ana, out[name]["analysis"]["mix"], priced(ana, BEST), priced(ana, own)))
    dst = os.path.join(ROOT, "profiles", "isa_mix.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
