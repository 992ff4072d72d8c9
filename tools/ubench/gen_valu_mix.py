#!/usr/bin/env python3
"""Generate valu_mix.hip: does a 2-cycle VALU instruction keep its rate between 4-cycle packed ones?

Round 1's table (profiles/r01_ubench_valu_rates2.txt) shows two classes on gfx950 at >= 2 waves per SIMD:
  ~0.9-1.0 ns per wave64 instruction  (v_add_u32, v_bitop3_b32, v_add_f16, v_fma_f32, logic ops ...)
  ~1.7 ns                             (every VOP3P packed op, v_perm_b32, most 3-operand VOP3 ops)
but its only mixes were DEPENDENT chains (each instruction read the register the previous one wrote).
Here every instruction of a unit writes its own accumulator (16 independent chains per wave), and the
units are the instruction mix of the bound build's cell pair (sw_cb_kernel.inc):
    today6   v_perm_b32, v_pk_add_f16, v_pk_maximum3_f16, v_pk_add_f16, v_pk_max_f16, v_pk_maximum3_f16
candidates that replace the v_perm_b32 (pairing the two sequences' profile scores) by a 2-cycle class op:
    bitop6   v_bitop3_b32 (bit select under an SGPR mask; needs the second sequence's profile stored with
             swapped halves and odd rows kept in swapped halves, which packed ops read through op_sel for free)
    half_*   perm + pk_add replaced by two half-word adds: SDWA / v_fma_f16 op_sel / v_fma_mixlo|hi_f16
plus alternations slow/fast and single-instruction rates of the forms involved.
"""
import sys

NACC = 16

SLOW = {
    "perm": "v_perm_b32 {d}, {d}, {b}, {c}",
    "pk_add": "v_pk_add_f16 {d}, {d}, {b}",
    "pk_max": "v_pk_max_f16 {d}, {d}, {b}",
    "pk_max3": "v_pk_maximum3_f16 {d}, {d}, {b}, {c}",
    "pk_add_sw": "v_pk_add_f16 {d}, {d}, {b} op_sel:[0,1] op_sel_hi:[1,0]",
    "pk_max_sw": "v_pk_max_f16 {d}, {d}, {b} op_sel:[0,1] op_sel_hi:[1,0]",
    "pk_max3_sw": "v_pk_maximum3_f16 {d}, {d}, {b}, {c} op_sel:[0,1,0] op_sel_hi:[1,0,1]",
}
FAST = {
    "bitop3": "v_bitop3_b32 {d}, {d}, {b}, {s} bitop3:0xca",
    "add_u32": "v_add_u32 {d}, {d}, {b}",
    "add_f16": "v_add_f16 {d}, {d}, {b}",
    "max_f16": "v_max_f16 {d}, {d}, {b}",
    "mov": "v_mov_b32 {d}, {b}",
    "and": "v_and_b32 {d}, {d}, {b}",
    "fma_f32": "v_fma_f32 {d}, {d}, {b}, {c}",
}
HALF = {
    "sdwa_lo": "v_add_f16_sdwa {d}, {d}, {b} dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1",
    "sdwa_hi": "v_add_f16_sdwa {d}, {d}, {c} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1",
    "fma16_lo": "v_fma_f16 {d}, {d}, 1.0, {b} op_sel:[0,0,1,0]",
    "fma16_hi": "v_fma_f16 {d}, {d}, 1.0, {c} op_sel:[1,0,1,1]",
    "mix_lo": "v_fma_mixlo_f16 {d}, {d}, 1.0, {b} op_sel:[0,0,1] op_sel_hi:[1,0,1]",
    "mix_hi": "v_fma_mixhi_f16 {d}, {d}, 1.0, {c} op_sel:[1,0,1] op_sel_hi:[1,0,1]",
    "max3_f16": "v_max3_f16 {d}, {d}, {b}, {c} op_sel:[0,1,0,0]",
    "add_i16": "v_add_i16 {d}, {d}, {b} op_sel:[0,1,0]",
    "bfi": "v_bfi_b32 {d}, {s}, {d}, {b}",
    "sdwa_or": "v_or_b32_sdwa {d}, {d}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
}
ALL = {**SLOW, **FAST, **HALF}

UNITS = {
    "today6": ["perm", "pk_add", "pk_max3", "pk_add", "pk_max", "pk_max3"],
    "bitop6": ["bitop3", "pk_add", "pk_max3", "pk_add", "pk_max", "pk_max3"],
    "bitop6_opsel": ["bitop3", "pk_add_sw", "pk_max3_sw", "pk_add", "pk_max_sw", "pk_max3"],
    "pk5": ["pk_add", "pk_max3", "pk_add", "pk_max", "pk_max3"],
    "half_sdwa": ["sdwa_lo", "sdwa_hi", "pk_max3", "pk_add", "pk_max", "pk_max3"],
    "half_fma16": ["fma16_lo", "fma16_hi", "pk_max3", "pk_add", "pk_max", "pk_max3"],
    "half_mix": ["mix_lo", "mix_hi", "pk_max3", "pk_add", "pk_max", "pk_max3"],
    "exact7": ["perm", "pk_add", "pk_max3", "pk_add", "pk_max", "pk_add", "pk_max3", "pk_max"],  # exact kernel, 7.5 -> 8 ops
    "exact7_bitop": ["bitop3", "pk_add", "pk_max3", "pk_add", "pk_max", "pk_add", "pk_max3", "pk_max"],
    "alt_max3_bitop3": ["pk_max3", "bitop3"],
    "alt_max3_addu32": ["pk_max3", "add_u32"],
    "alt_max3_addf16": ["pk_max3", "add_f16"],
    "alt_max3_mov": ["pk_max3", "mov"],
    "alt_max3_fma32": ["pk_max3", "fma_f32"],
    "alt_pkadd_bitop3": ["pk_add", "bitop3"],
    "s2f1": ["pk_max3", "pk_add", "bitop3"],
    "s1f2": ["pk_max3", "bitop3", "add_u32"],
    "s1f3": ["pk_max3", "bitop3", "add_u32", "and"],
    "ff": ["bitop3", "add_u32"],
}

# the real dependency structure of the bound build's rows (F runs down the rows; sc / a / E hang off it):
#   sc = pair(wa, wb); a = hold + sc; h = max3(a, E, F); t = h + negQR; F = max(F, t); E = max3(E, t, floor)
CHAIN = {
    "chain_today": "v_perm_b32 {sc}, {wa}, {wb}, {c}\nv_pk_add_f16 {a}, {H}, {sc}\nv_pk_maximum3_f16 {H}, {a}, {E}, {F}\n"
                   "v_pk_add_f16 {t}, {H}, {b}\nv_pk_max_f16 {F}, {F}, {t}\nv_pk_maximum3_f16 {E}, {E}, {t}, {c}",
    "chain_bitop": "v_bitop3_b32 {sc}, {wa}, {wb}, {s} bitop3:0xca\nv_pk_add_f16 {a}, {H}, {sc} op_sel:[1,0] op_sel_hi:[0,1]\n"
                   "v_pk_maximum3_f16 {H}, {a}, {E}, {F} op_sel:[0,0,1] op_sel_hi:[1,1,0]\n"
                   "v_pk_add_f16 {t}, {H}, {b}\nv_pk_max_f16 {F}, {F}, {t} op_sel:[1,0] op_sel_hi:[0,1]\nv_pk_maximum3_f16 {E}, {E}, {t}, {c}",
}


def unit_body(names, nunits):
    out, k = [], 0
    for _ in range(nunits):
        for n in names:
            out.append(ALL[n].format(d=f"%{k % NACC}", b="%16", c="%17", s="%18"))
            k += 1
    return "\\n".join(out) + "\\n", k


def chain_body(templ, rows):
    # registers: %0 = F (carried down the rows), rows use H = %1..%6, E = %7..%12 (6 rows), temporaries %13 %14 %15
    out = []
    for r in range(rows):
        rr = r % 6
        out.append(templ.format(F="%0", H=f"%{1 + rr}", E=f"%{7 + rr}", sc="%13", a="%14", t="%15", wa="%16", wb="%17",
                                b="%16", c="%17", s="%18").replace("\n", "\\n"))
    return "\\n".join(out) + "\\n", rows * 6


def main():
    src = [r'''// generated by gen_valu_mix.py -- do not edit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int REP = 1500;
#define DEFINE_TEST(NAME, ASMSTR)                                                     \
__global__ void __launch_bounds__(256) k_##NAME(unsigned* out, unsigned seed) {       \
  unsigned b = seed * 3 + threadIdx.x, c = seed + 7;                                  \
  unsigned a[16];                                                                     \
  for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x + i;                         \
  unsigned mask = 0xffffu * seed;                                                     \
  for (int r = 0; r < REP; ++r) {                                                     \
    asm volatile(ASMSTR                                                               \
       : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
       : "v"(b), "v"(c), "s"(mask));                                                  \
  }                                                                                   \
  unsigned acc = 0;                                                                   \
  for (int i = 0; i < 16; ++i) acc ^= a[i];                                           \
  if (acc == 0x12345678u) out[1] = acc;                                               \
}
''']
    tests = []
    for name in ALL:
        body, n = unit_body([name], 48)
        src.append(f'DEFINE_TEST(one_{name}, "{body}")')
        tests.append((f"one_{name}", n, 1))
    for name, names in UNITS.items():
        body, n = unit_body(names, max(1, 48 // len(names)))
        src.append(f'DEFINE_TEST({name}, "{body}")')
        tests.append((name, n, len(names)))
    for name, templ in CHAIN.items():
        body, n = chain_body(templ, 8)
        src.append(f'DEFINE_TEST({name}, "{body}")')
        tests.append((name, n, 6))
    src.append(r'''
typedef void (*kern_t)(unsigned*, unsigned);
struct Test { const char* name; kern_t k; int n; int per; };
int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  unsigned* d; CHECK(hipMalloc(&d, 16));
  std::vector<Test> tests = {''')
    for name, n, per in tests:
        src.append(f'    {{"{name}", k_{name}, {n}, {per}}},')
    src.append(r'''  };
  printf("device %s, %d CUs, %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  printf("%-20s %8s %8s %8s %8s %8s  ns per wave64 instruction per SIMD at 1/2/3/4/8 waves per SIMD; [ns per unit at 2 | 8 waves]\n", "test", "1w", "2w", "3w", "4w", "8w");
  for (auto& t : tests) {
    printf("%-20s", t.name);
    double at2 = 0, at8 = 0;
    for (int wps : {1, 2, 3, 4, 8}) {
      int blocks = p.multiProcessorCount * wps;
      hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
      CHECK(hipDeviceSynchronize());
      double best = 1e30;
      for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
        double instr = double(REP) * t.n * wps;
        best = ms * 1e6 / instr < best ? ms * 1e6 / instr : best;
      }
      if (wps == 2) at2 = best;
      if (wps == 8) at8 = best;
      printf(" %8.3f", best);
    }
    if (t.per > 1) printf("   [%6.2f | %6.2f ns per %d-instruction unit]", at2 * t.per, at8 * t.per, t.per);
    printf("\n");
  }
  return 0;
}''')
    open(sys.argv[1] if len(sys.argv) > 1 else "valu_mix.hip", "w").write("\n".join(src))


if __name__ == "__main__":
    main()
