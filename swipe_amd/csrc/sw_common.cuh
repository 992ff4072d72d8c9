// Device primitives shared by the kernel translation units (sw_kernels.hip, sw_mp_kernels.hip).
#ifndef SW_COMMON_CUH
#define SW_COMMON_CUH
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sw_device.h"

typedef unsigned int u32;
typedef unsigned long long u64;

// ------------------------------------------------------------------ packed f16 primitives
// hipcc lowers __builtin_elementwise_maximum on a 2 x f16 vector to v_pk_maximum3_f16 (fusing
// nested calls and literal zero operands) without canonicalising inputs; a + b is v_pk_add_f16.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 pk_max(h2 a, h2 b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ h2 pk_max3(h2 a, h2 b, h2 c)
{ return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }
__device__ __forceinline__ h2 as_h2(u32 v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ u32 as_u32(h2 v) { return __builtin_bit_cast(u32, v); }

#define DPP_ROW_SHR1 0x111
#define DPP_ROW_SHL1 0x101
#define DPP_ROW_ROR1 0x121       /* rotate right within the 16 lanes of a row: lane 0 takes lane 15's */
#define DPP_ROW_SHR(n) (0x110 + (n))

// value of lane-1 within the 16-lane row; lane 0 of each row receives `fill`
__device__ __forceinline__ u32 row_shr1(u32 v, u32 fill)
{ return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, DPP_ROW_SHR1, 0xF, 0xF, false); }
// The residue register of a chain of G lanes moves on by one lane and the chain's first lane takes `fresh`.  A 16-lane
// chain is a DPP row: shift, with `fresh` preset where no lane sends (a v_mov, since the preset has to sit in the
// destination).  Shorter chains need a select for the first lanes inside the row anyway: rotate instead (every lane
// has a sender, nothing to preset) and select on all first lanes - one instruction less per step.
// LDS offsets of a stream word's two residues (low byte: the pair's first sequence) as offA | offB << 16, cs = bytes per
// residue of the profile.  Written with 24-bit multiplies: left alone the compiler folds the shift into the factor
// ((raw >> 8) * (cs << 16)), which no longer fits 24 bits and costs a quarter-rate v_mul_lo_u32 per block of columns.
__device__ __forceinline__ u32 pair_offsets(u32 raw, u32 cs)
{
  return __umul24(raw & 0xFFu, cs) + __umul24(raw & 0xFF00u, cs << 8);
}
// (round 6) ... and rotate and select are ONE instruction: v_cndmask_b32 is VOP2, so its first source can come through DPP -
// dst = vcc ? fresh : row_ror:1(cur).  The compiler keeps the lane-0 condition in an SGPR pair and emits the VOP3 form, which
// cannot, so the instruction is written out; the lanes that start a chain are a constant of G, loaded into vcc in front of it
// (two SALU moves, which are also the two wait states a DPP read wants behind a VALU write of its source).
template <int G>
__device__ __forceinline__ u32 chain_advance(u32 cur, u32 fresh, bool first)
{
  if constexpr (G < 16) {
    static_assert(G == 2 || G == 4 || G == 8, "chains inside a DPP row");
    constexpr u32 heads = G == 8 ? 0x01010101u : G == 4 ? 0x11111111u : 0x55555555u;       // lanes with lane % G == 0
    (void)first;
    u32 out;
    asm("s_mov_b32 vcc_lo, %3\n\ts_mov_b32 vcc_hi, %3\n\tv_cndmask_b32_dpp %0, %1, %2, vcc row_ror:1 row_mask:0xf bank_mask:0xf"
        : "=v"(out) : "v"(cur), "v"(fresh), "n"(heads) : "vcc");
    return out;
  } else {
    return row_shr1(cur, fresh);
  }
}
__device__ __forceinline__ u32 row_shl1(u32 v)
{ return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHL1, 0xF, 0xF, false); }

__device__ __forceinline__ u32 float_to_half_bits(float f)
{ _Float16 x = (_Float16)f; unsigned short s; __builtin_memcpy(&s, &x, 2); return s; }

__device__ __forceinline__ void seq_span(const swa_seqs& s, int id, int64_t& o, int64_t& len)
{
  if (id < s.nseq) { o = s.offsets[id]; len = s.offsets[id + 1] - o; }
  else { o = s.wstart[id - s.nseq]; len = s.wlen[id - s.nseq]; }
}
__device__ __forceinline__ u32 seq_residue(const swa_seqs& s, int64_t idx)
{
  return s.packed ? ((u32)s.residues[idx >> 1] >> ((int)(idx & 1) * 4)) & 15u : (u32)s.residues[idx];
}

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u4v* lds_u4_ptr;

// scores of the 8 sequences of a batch + overflow re-queue: ballot, one atomic per wave, compacted append
__device__ __forceinline__ void narrow_write_scores(const swa_narrow_params& p, int b, int lane, h2 S)
{
  const bool writer = (lane & 15) == 15;
  const int grp = lane >> 4;
  int sA = -1, sB = -1, idA = -1, idB = -1;
  if (writer) {
    idA = p.slots[(int64_t)b * SWA_SLOTS + grp * 2];
    idB = p.slots[(int64_t)b * SWA_SLOTS + grp * 2 + 1];
    sA = (int)(float)S.x;
    sB = (int)(float)S.y;
    if (idA >= 0) p.scores[idA] = sA;
    if (idB >= 0) p.scores[idB] = sB;
  }
  // overflow re-queue: ballot, one atomic per wave, compacted append
  const bool oA = writer && idA >= 0 && sA >= p.limit;
  const bool oB = writer && idB >= 0 && sB >= p.limit;
  const u64 mA = __ballot(oA), mB = __ballot(oB);
  const int nA = __popcll(mA), nB = __popcll(mB);
  if (nA + nB) {
    int base = 0;
    if (lane == 0) base = atomicAdd(p.ovf_count, nA + nB);
    base = __builtin_amdgcn_readfirstlane(base);
    const u64 below = (1ull << lane) - 1;
    if (oA) p.ovf_list[base + __popcll(mA & below)] = idA;
    if (oB) p.ovf_list[base + nA + __popcll(mB & below)] = idB;
  }
}

#endif
