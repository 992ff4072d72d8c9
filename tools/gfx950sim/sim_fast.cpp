// gfx950sim: AVX-512 (FP16 / BW / VBMI) forms of the handful of wave instructions the DP kernels spend their time in.
// Compiled with clang++ (g++ 11 has no -mavx512fp16) and used only when cpuid reports the features; every function here
// has a scalar twin in sim_isa.cpp that defines the semantics, and tools/gfx950sim/selftest compares the two.
#include <immintrin.h>
#include <cpuid.h>
#include <cstdint>
#include <cstring>

namespace simfast {

bool available() {
    static int ok = -1;
    if (ok < 0) {
        unsigned a, b, c, d;
        ok = 0;
        if (__get_cpuid_count(7, 0, &a, &b, &c, &d)) {
            bool f = b & (1u << 16), bw = b & (1u << 30), vl = b & (1u << 31), vbmi = c & (1u << 1), fp16 = d & (1u << 23);
            ok = f && bw && vl && vbmi && fp16;
        }
    }
    return ok == 1;
}

static inline __m512i ld(const uint32_t* p, int st, int part) { return st ? _mm512_loadu_si512((const void*)(p + 16 * part)) : _mm512_set1_epi32((int)*p); }
static inline void st_masked(uint32_t* d, int part, uint64_t exec, __m512i v) { _mm512_mask_storeu_epi32((void*)(d + 16 * part), (__mmask16)(exec >> (16 * part)), v); }

void pk_add_f16(uint32_t* d, const uint32_t* a, int sa, uint32_t xa, const uint32_t* b, int sb, uint32_t xb, uint64_t exec) {
    const __m512i XA = _mm512_set1_epi32((int)xa), XB = _mm512_set1_epi32((int)xb);
    for (int p = 0; p < 4; p++) {
        __m512h x = _mm512_castsi512_ph(_mm512_xor_si512(ld(a, sa, p), XA)), y = _mm512_castsi512_ph(_mm512_xor_si512(ld(b, sb, p), XB));
        st_masked(d, p, exec, _mm512_castph_si512(_mm512_add_ph(x, y)));
    }
}
void pk_mul_f16(uint32_t* d, const uint32_t* a, int sa, uint32_t xa, const uint32_t* b, int sb, uint32_t xb, uint64_t exec) {
    const __m512i XA = _mm512_set1_epi32((int)xa), XB = _mm512_set1_epi32((int)xb);
    for (int p = 0; p < 4; p++) {
        __m512h x = _mm512_castsi512_ph(_mm512_xor_si512(ld(a, sa, p), XA)), y = _mm512_castsi512_ph(_mm512_xor_si512(ld(b, sb, p), XB));
        st_masked(d, p, exec, _mm512_castph_si512(_mm512_mul_ph(x, y)));
    }
}
void pk_fma_f16(uint32_t* d, const uint32_t* a, int sa, uint32_t xa, const uint32_t* b, int sb, uint32_t xb, const uint32_t* c, int sc, uint32_t xc, uint64_t exec) {
    const __m512i XA = _mm512_set1_epi32((int)xa), XB = _mm512_set1_epi32((int)xb), XC = _mm512_set1_epi32((int)xc);
    for (int p = 0; p < 4; p++) {
        __m512h x = _mm512_castsi512_ph(_mm512_xor_si512(ld(a, sa, p), XA)), y = _mm512_castsi512_ph(_mm512_xor_si512(ld(b, sb, p), XB)),
                z = _mm512_castsi512_ph(_mm512_xor_si512(ld(c, sc, p), XC));
        st_masked(d, p, exec, _mm512_castph_si512(_mm512_fmadd_ph(x, y, z)));
    }
}
// IEEE-754 maximum of three (NaN wins, -0 < +0) through the sign-magnitude -> two's complement key; false (nothing written)
// when an operand of an active lane is a NaN: the caller's scalar form handles that
bool pk_maximum3_f16(uint32_t* d, const uint32_t* a, int sa, uint32_t xa, const uint32_t* b, int sb, uint32_t xb, const uint32_t* c, int sc, uint32_t xc, uint64_t exec) {
    const __m512i XA = _mm512_set1_epi32((int)xa), XB = _mm512_set1_epi32((int)xb), XC = _mm512_set1_epi32((int)xc);
    const __m512i ABS = _mm512_set1_epi16(0x7fff), INF = _mm512_set1_epi16(0x7c00);
    __m512i out[4];
    for (int p = 0; p < 4; p++) {
        __m512i x = _mm512_xor_si512(ld(a, sa, p), XA), y = _mm512_xor_si512(ld(b, sb, p), XB), z = _mm512_xor_si512(ld(c, sc, p), XC);
        __mmask32 nan = _mm512_cmpgt_epi16_mask(_mm512_and_si512(x, ABS), INF) | _mm512_cmpgt_epi16_mask(_mm512_and_si512(y, ABS), INF) |
                        _mm512_cmpgt_epi16_mask(_mm512_and_si512(z, ABS), INF);
        if (nan) {
            uint32_t lanes = (uint32_t)(exec >> (16 * p)) & 0xffff, hit = 0;
            for (int l = 0; l < 16; l++) if ((nan >> (2 * l)) & 3) hit |= 1u << l;
            if (hit & lanes) return false;
        }
        auto key = [&](__m512i h) { return _mm512_xor_si512(h, _mm512_and_si512(_mm512_srai_epi16(h, 15), ABS)); };
        __m512i k = _mm512_max_epi16(_mm512_max_epi16(key(x), key(y)), key(z));
        out[p] = key(k);          // the key map is an involution
    }
    for (int p = 0; p < 4; p++) st_masked(d, p, exec, out[p]);
    return true;
}
// v_perm_b32 with selector bytes 0..7 and 12 only; false otherwise
bool perm_b32(uint32_t* d, const uint32_t* s0, int st0, const uint32_t* s1, int st1, const uint32_t* sel, int sts, uint64_t exec) {
    alignas(64) static uint8_t lane_off[64];
    static bool init = false;
    if (!init) { for (int i = 0; i < 64; i++) lane_off[i] = (uint8_t)(i & ~3); init = true; }
    const __m512i OFF = _mm512_load_si512((const void*)lane_off);
    __m512i out[4];
    for (int p = 0; p < 4; p++) {
        __m512i s = ld(sel, sts, p);
        __mmask64 zero = _mm512_cmpeq_epi8_mask(s, _mm512_set1_epi8(12));
        __mmask64 bad = _mm512_cmpgt_epu8_mask(s, _mm512_set1_epi8(7)) & ~zero;
        if (bad) {
            uint32_t lanes = (uint32_t)(exec >> (16 * p)) & 0xffff;
            for (int l = 0; l < 16; l++) if (((bad >> (4 * l)) & 15) && ((lanes >> l) & 1)) return false;
        }
        __m512i idx = _mm512_add_epi8(OFF, _mm512_add_epi8(_mm512_and_si512(s, _mm512_set1_epi8(3)), _mm512_slli_epi16(_mm512_and_si512(s, _mm512_set1_epi8(4)), 4)));
        out[p] = _mm512_maskz_permutex2var_epi8(~zero, ld(s1, st1, p), idx, ld(s0, st0, p));
    }
    for (int p = 0; p < 4; p++) st_masked(d, p, exec, out[p]);
    return true;
}
void mov_b32(uint32_t* d, const uint32_t* a, int sa, uint64_t exec) {
    for (int p = 0; p < 4; p++) st_masked(d, p, exec, ld(a, sa, p));
}

}  // namespace simfast
