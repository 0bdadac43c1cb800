#!/usr/bin/env python3
"""Writes tools/microbench/valu_rate.hip: one kernel per VALU instruction form, each ONE inline-assembly block of 32
instructions over 16 accumulators, so that what is timed is what the label says (see the header it writes).

    python tools/microbench/gen_valu_rate.py

Placeholders of a form: %D accumulator i (destination), %A accumulator i+3, %B accumulator i+7, %S / %T two SGPRs,
%P an SGPR pair.  mode 0: 32-bit accumulators, 1: register pairs (packed forms), 2: every instruction on accumulator 0
(a dependent chain)."""
import os

FORMS = [
    ("v_fma_f32 v,v,v", "v_fma_f32 %D, %D, %A, %B", 0),
    ("v_fma_f32 v,s,v", "v_fma_f32 %D, %D, %S, %A", 0),
    ("v_fma_f32 v,s,s (one SGPR twice)", "v_fma_f32 %D, %D, %S, %S", 0),
    ("v_fmac_f32 s,v", "v_fmac_f32 %D, %S, %A", 0),
    ("v_add_f32 v,v", "v_add_f32 %D, %D, %A", 0),
    ("v_mul_f32 s,v", "v_mul_f32 %D, %S, %D", 0),
    ("v_max_f32 v,v", "v_max_f32 %D, %D, %A", 0),
    ("v_max3_f32 v,v,v", "v_max3_f32 %D, %D, %A, %B", 0),
    ("v_min3_f32 v,s,v", "v_min3_f32 %D, %D, %S, %A", 0),
    ("v_cmp_lt_f32 (vcc)", "v_cmp_lt_f32 vcc, %D, %A", 0),
    ("v_cndmask_b32 (vcc)", "v_cndmask_b32 %D, %D, %A, vcc", 0),
    ("v_cmp + v_cndmask (pair)", "v_cmp_lt_f32 vcc, %D, %A\\n\\tv_cndmask_b32 %D, %D, %B, vcc", 0),
    ("v_mov_b32 dpp row_shr", "v_mov_b32_dpp %D, %A row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", 0),
    ("v_add_f32 dpp row_shr", "v_add_f32_dpp %D, %A, %D row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", 0),
    ("v_mov_b32 dpp quad_perm", "v_mov_b32_dpp %D, %A quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 0),
    ("v_floor_f32", "v_floor_f32 %D, %A", 0),
    ("v_cvt_i32_f32", "v_cvt_i32_f32 %D, %A", 0),
    ("v_cvt_f32_i32", "v_cvt_f32_i32 %D, %A", 0),
    ("v_rcp_f32", "v_rcp_f32 %D, %A", 0),
    ("v_sqrt_f32", "v_sqrt_f32 %D, %A", 0),
    ("v_exp_f32", "v_exp_f32 %D, %A", 0),
    ("v_add_u32 v,v", "v_add_u32 %D, %D, %A", 0),
    ("v_add_u32 s,v", "v_add_u32 %D, %S, %D", 0),
    ("v_lshlrev_b32", "v_lshlrev_b32 %D, 3, %A", 0),
    ("v_and_b32 v,v", "v_and_b32 %D, %D, %A", 0),
    ("v_lshl_add_u32", "v_lshl_add_u32 %D, %A, 2, %D", 0),
    ("v_add3_u32", "v_add3_u32 %D, %D, %A, %B", 0),
    ("v_mad_u32_u24", "v_mad_u32_u24 %D, %A, %B, %D", 0),
    ("v_mul_lo_u32", "v_mul_lo_u32 %D, %D, %A", 0),
    ("v_bfe_u32", "v_bfe_u32 %D, %A, 3, 5", 0),
    ("v_perm_b32", "v_perm_b32 %D, %D, %A, %B", 0),
    ("v_cmp_eq_u32 (vcc)", "v_cmp_eq_u32 vcc, %D, %A", 0),
    ("v_readlane_b32 (to vcc_lo)", "v_readlane_b32 vcc_lo, %D, 3", 0),
    ("v_mov_b32", "v_mov_b32 %D, %A", 0),
    ("v_pk_fma_f32 v,s,v", "v_pk_fma_f32 %D, %D, %P, %A", 1),
    ("v_pk_fma_f32 v,v,v", "v_pk_fma_f32 %D, %D, %A, %B", 1),
    ("v_pk_fma_f32 v,s,s", "v_pk_fma_f32 %D, %D, %P, %P", 1),
    ("v_pk_add_f32 v,v", "v_pk_add_f32 %D, %D, %A", 1),
    ("v_pk_mul_f32 v,s", "v_pk_mul_f32 %D, %D, %P", 1),
    ("v_pk_mov_b32", "v_pk_mov_b32 %D, %A, %B op_sel:[0,1]", 1),
    ("v_mov_b64", "v_mov_b64 %D, %A", 1),
    ("v_fma_f32 dependent chain", "v_fma_f32 %D, %D, %S, %D", 2),
    ("v_mul_f32 v,v", "v_mul_f32 %D, %D, %A", 0),
    ("v_sub_f32 v,v", "v_sub_f32 %D, %D, %A", 0),
    ("v_fmac_f32 v,v", "v_fmac_f32 %D, %A, %B", 0),
    ("v_min_f32 v,v", "v_min_f32 %D, %D, %A", 0),
    ("v_or_b32 v,v", "v_or_b32 %D, %D, %A", 0),
    ("v_xor_b32 v,v", "v_xor_b32 %D, %D, %A", 0),
    ("v_sub_u32 v,v", "v_sub_u32 %D, %D, %A", 0),
    ("v_lshlrev_b32 v,v", "v_lshlrev_b32 %D, %A, %D", 0),
    ("v_add_f32 0.5,v (inline)", "v_add_f32 %D, 0.5, %D", 0),
    ("v_add_f32 literal,v", "v_add_f32 %D, 0x40490fdb, %D", 0),
    ("v_mul_f32 literal,v", "v_mul_f32 %D, 0x40490fdb, %D", 0),
    ("v_fmaak_f32 v,v,literal", "v_fmaak_f32 %D, %D, %A, 0x40490fdb", 0),
    ("v_add_u32 inline,v", "v_add_u32 %D, 17, %D", 0),
    ("v_cndmask_b32 e64 (sgpr mask)", "v_cndmask_b32_e64 %D, %D, %A, s[2:3]", 0),
    ("v_min_i32 s,v", "v_min_i32 %D, %S, %D", 0),
    ("v_mul_u32_u24 v,v", "v_mul_u32_u24 %D, %D, %A", 0),
    ("v_med3_f32 v,v,v", "v_med3_f32 %D, %D, %A, %B", 0),
    ("v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %D, %A", 0),
    ("v_add_u32 sdwa", "v_add_u32_sdwa %D, %D, %A dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0", 0),
    ("v_rsq_f32", "v_rsq_f32 %D, %A", 0),
    ("v_log_f32", "v_log_f32 %D, %A", 0),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %D, %A, %D", 0),
    ("v_pk_fma_f32 v,v,v op_sel (broadcast lo)", "v_pk_fma_f32 %D, %D, %A, %B op_sel_hi:[1,0,1]", 1),
    # the orientation histogram's "acc += (b == mybin) ? w : 0" (one dependent chain on the accumulator), as a select and
    # with the compare writing EXEC; these two rows are cycles per GROUP (= per sample), not per instruction
    ("GROUP v_cmp + s_nop + v_cndmask + v_add chain", "v_cmp_eq_u32 vcc, %A, %B\\n\\ts_nop 1\\n\\tv_cndmask_b32 %A, 0, %B, vcc\\n\\tv_add_f32 %D, %D, %A", 2),
    ("GROUP v_cmpx + v_add chain + s_mov exec", "v_cmpx_eq_u32 vcc, %A, %B\\n\\tv_add_f32 %D, %D, %A\\n\\ts_mov_b64 exec, -1", 2),
    ("GROUP v_cmp + v_add chain (no select)", "v_cmp_eq_u32 vcc, %A, %B\\n\\tv_add_f32 %D, %D, %A", 2),
]

HEAD = r'''// valu_rate.hip -- GENERATED by tools/microbench/gen_valu_rate.py (edit that, not this).
// What one SIMD of gfx950 sustains in VALU wave-instructions per cycle, by instruction form and by the number of waves
// resident on the SIMD.  Every form is ONE inline-assembly block of 32 instructions over 16 accumulators (accumulator i,
// i+3 and i+7 as operands; two SGPRs; an SGPR pair): the first version of this file left the choice of instruction to
// the compiler, which packed its "scalar" fused multiply-adds into v_pk_fma_f32 behind its back -- half of the rows of
// the table it produced (profiles/r03/valu_rate_microbench.txt) described another instruction than their label.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/microbench/valu_rate.hip && /tmp/valu_rate [substring]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef float f2 __attribute__((ext_vector_type(2)));

#define ACC(v) "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), \
               "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])

template <int kKind>
__global__ void __launch_bounds__(64) k(float *out, int iters, float a, float b) {
  float x[16];
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x[i] = threadIdx.x * 0.001f + i;
    p[i] = f2{x[i], x[i] + 1.0f};
  }
  const f2 s2 = f2{a, b};
  for (int it = 0; it < iters; ++it) {
'''

TAIL = r'''  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int kKind>
void run(const char *name, int per_block, int cus, const char *only) {
  if (only && !strstr(name, only)) return;
  float *d;
  (void)hipMalloc(&d, 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 20000;
  int clk_khz = 0;
  (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("%-42s", name);
  for (int waves_per_simd : {1, 2, 3, 4, 8}) {
    const int blocks = cus * 4 * waves_per_simd;  // 64-thread blocks: one wave each, 4 SIMDs per CU
    hipLaunchKernelGGL(k<kKind>, dim3(blocks), dim3(64), 0, 0, d, 100, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<kKind>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)waves_per_simd * iters * per_block;
    const double cycles = ms * 1e-3 * clk_khz * 1e3;
    printf("  %dw %5.2f", waves_per_simd, cycles / insts_per_simd);
  }
  printf("   cycles per wave-instruction per SIMD at %d MHz\n", clk_khz / 1000);
  (void)hipFree(d);
}

int main(int argc, char **argv) {
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const char *only = argc > 1 ? argv[1] : nullptr;
'''


def main():
    out = [HEAD]
    for kid, (name, fmt, mode) in enumerate(FORMS):
        lines = []
        for _ in range(2):
            for i in range(16):
                d = 0 if mode == 2 else i
                lines.append(fmt.replace("%D", "%%%d" % d).replace("%A", "%%%d" % ((i + 3) & 15))
                             .replace("%B", "%%%d" % ((i + 7) & 15)).replace("%S", "%16").replace("%T", "%17")
                             .replace("%P", "%18"))
        body = "\\n\\t".join(lines)
        var = "p" if mode == 1 else "x"
        out.append('    if (kKind == %d)  // %s\n      asm volatile("%s"\n                   : ACC(%s) : "s"(a), "s"(b), '
                   '"s"(s2) : "vcc", "m0");\n' % (kid, name, body, var))
    out.append(TAIL)
    for kid, (name, fmt, _) in enumerate(FORMS):
        out.append('  run<%d>("%s", %d, cus, only);\n' % (kid, name, 32 if name.startswith("GROUP") else 64 if "\\n" in fmt else 32))
    out.append("  return 0;\n}\n")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_rate.hip")
    with open(path, "w") as f:
        f.write("".join(out))
    print(path)


if __name__ == "__main__":
    main()
