// gfx950sim: the parser of llvm-objdump text and the wave64 interpreter. See sim_core.h. Test infrastructure only.
#include "sim_core.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <random>
#include <sstream>
#include <stdexcept>
#include <atomic>
#include <mutex>
#include <thread>
#include <algorithm>

namespace simfast {
bool available();
void pk_add_f16(uint32_t* d, const uint32_t* a, int sa, uint32_t xa, const uint32_t* b, int sb, uint32_t xb, uint64_t exec);
void pk_mul_f16(uint32_t* d, const uint32_t* a, int sa, uint32_t xa, const uint32_t* b, int sb, uint32_t xb, uint64_t exec);
void pk_fma_f16(uint32_t* d, const uint32_t* a, int sa, uint32_t xa, const uint32_t* b, int sb, uint32_t xb, const uint32_t* c, int sc, uint32_t xc, uint64_t exec);
bool pk_maximum3_f16(uint32_t* d, const uint32_t* a, int sa, uint32_t xa, const uint32_t* b, int sb, uint32_t xb, const uint32_t* c, int sc, uint32_t xc, uint64_t exec);
bool perm_b32(uint32_t* d, const uint32_t* s0, int st0, const uint32_t* s1, int st1, const uint32_t* sel, int sts, uint64_t exec);
void mov_b32(uint32_t* d, const uint32_t* a, int sa, uint64_t exec);
}

namespace sim {

static bool g_fast = false;            // AVX-512 forms of the hot instructions (HIPSIM_FAST=0 keeps the scalar definitions)

enum Op : uint16_t {
    OP_INVALID = 0,
#define X(m, c) OP_##m,
#include "sim_ops.inc"
#undef X
    OP_COUNT
};
static const char* op_name[] = {"?",
#define X(m, c) #m,
#include "sim_ops.inc"
#undef X
};
static const uint8_t op_cls[] = {C_OTHER,
#define X(m, c) c,
#include "sim_ops.inc"
#undef X
};

static std::unordered_map<std::string, uint16_t>& op_table() {
    static std::unordered_map<std::string, uint16_t> t;
    if (t.empty())
        for (int i = 1; i < OP_COUNT; i++) t[op_name[i]] = (uint16_t)i;
    return t;
}

struct Fault : std::runtime_error { using std::runtime_error::runtime_error; };

static constexpr int SG_VCC = 106, SG_M0 = 124, SG_EXEC = 126;
static constexpr uint32_t SHARED_BASE_HI = 0x00010000u, PRIVATE_BASE_HI = 0x00020000u;

// ------------------------------------------------------------------------------------------------ f16
static float g_h2f[65536];
static int16_t g_h2i[65536];          // small integer value of a half, or INT16_MIN
static uint16_t g_i2h[8193];          // half pattern of the integer i - 4096
static inline float bits2f(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

static uint16_t d2h(double d) {       // round to nearest even, one rounding
    uint64_t b; memcpy(&b, &d, 8);
    uint16_t sign = (uint16_t)((b >> 48) & 0x8000);
    int e = (int)((b >> 52) & 0x7ff);
    uint64_t m = b & 0xfffffffffffffull;
    if (e == 0x7ff) return m ? (uint16_t)(sign | 0x7e00) : (uint16_t)(sign | 0x7c00);
    if (e == 0) return sign;                              // double subnormals are far below half's range
    int ue = e - 1023;
    if (ue > 15) return sign | 0x7c00;
    m |= 1ull << 52;                                      // 53-bit significand, value = m * 2^(ue-52)
    int shift;                                            // bits to drop to land on half's grid
    int he;
    if (ue >= -14) { shift = 42; he = ue + 15; } else { shift = 42 + (-14 - ue); he = 0; }
    if (shift > 63) return sign;
    uint64_t q = m >> shift, r = m & ((1ull << shift) - 1), half = 1ull << (shift - 1);
    if (r > half || (r == half && (q & 1))) q++;
    uint32_t out;
    if (he == 0) out = (uint32_t)q;                       // subnormal (q may carry into exponent 1: that is right)
    else {
        out = ((uint32_t)he << 10) + (uint32_t)(q - 1024);   // q in [1024, 2048]; 2048 carries into the exponent
    }
    if (out >= 0x7c00) return sign | 0x7c00;
    return sign | (uint16_t)out;
}

static void init_tables() {
    static bool done = false;
    if (done) return;
    done = true;
    for (uint32_t h = 0; h < 65536; h++) {
        uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
        float f;
        if (e == 31) f = m ? NAN : INFINITY;
        else if (e == 0) f = ldexpf((float)m, -24);
        else f = ldexpf((float)(m | 1024), (int)e - 25);
        g_h2f[h] = s ? -f : f;
        g_h2i[h] = INT16_MIN;
    }
    for (int i = -4096; i <= 4096; i++) {
        uint16_t h = d2h((double)i);
        g_i2h[i + 4096] = h;
        if (g_h2f[h] == (float)i && !(i == 0 && (h & 0x8000))) g_h2i[h] = (int16_t)i;   // integers exact in f16 (|i| <= 2048, evens beyond)
    }
    g_h2i[0x8000] = INT16_MIN;        // -0 keeps the slow path (its sign matters to maximum())
}

static inline uint16_t h_add(uint16_t a, uint16_t b) {
    int ia = g_h2i[a], ib = g_h2i[b];
    if (ia != INT16_MIN && ib != INT16_MIN) {
        int s = ia + ib;
        if (s >= -2048 && s <= 2048) return g_i2h[s + 4096];
    }
    return d2h((double)g_h2f[a] + (double)g_h2f[b]);
}
static inline uint16_t h_maximum(uint16_t a, uint16_t b) {      // IEEE 754-2019 maximum: NaN wins, -0 < +0
    int ia = g_h2i[a], ib = g_h2i[b];
    if (ia != INT16_MIN && ib != INT16_MIN) return ia >= ib ? a : b;
    float fa = g_h2f[a], fb = g_h2f[b];
    if (fa != fa || fb != fb) return 0x7e00;
    if (fa == fb) return (a & 0x8000) ? b : a;
    return fa > fb ? a : b;
}
static inline uint16_t h_minimum(uint16_t a, uint16_t b) {
    float fa = g_h2f[a], fb = g_h2f[b];
    if (fa != fa || fb != fb) return 0x7e00;
    if (fa == fb) return (a & 0x8000) ? a : b;
    return fa < fb ? a : b;
}
static inline uint16_t h_max_legacy(uint16_t a, uint16_t b) {   // v_pk_max_f16: a NaN operand loses
    float fa = g_h2f[a], fb = g_h2f[b];
    if (fa != fa) return b;
    if (fb != fb) return a;
    if (fa == fb) return (a & 0x8000) ? b : a;
    return fa > fb ? a : b;
}
static inline uint16_t h_min_legacy(uint16_t a, uint16_t b) {
    float fa = g_h2f[a], fb = g_h2f[b];
    if (fa != fa) return b;
    if (fb != fb) return a;
    if (fa == fb) return (a & 0x8000) ? a : b;
    return fa < fb ? a : b;
}
static inline uint16_t h_fma(uint16_t a, uint16_t b, uint16_t c) {
    int ia = g_h2i[a], ib = g_h2i[b], ic = g_h2i[c];
    if (ia != INT16_MIN && ib != INT16_MIN && ic != INT16_MIN) {
        int s = ia * ib + ic;
        if (s >= -2048 && s <= 2048) return g_i2h[s + 4096];
    }
    return d2h(std::fma((double)g_h2f[a], (double)g_h2f[b], (double)g_h2f[c]));
}
static inline uint16_t h_mul(uint16_t a, uint16_t b) { return d2h((double)g_h2f[a] * (double)g_h2f[b]); }

}  // namespace sim
// the interpreter's binary16 arithmetic, for tests/test_sim_kernels.py (checked against numpy's float16): op 0 add, 1 mul, 2 fma,
// 3 IEEE-754-2019 maximum of three; fast != 0: the AVX-512 form of the packed instruction (low halves of the operands)
extern "C" int hipsim_f16_op(int op, int fast, const uint16_t* a, const uint16_t* b, const uint16_t* c, uint16_t* out, long n) {
    using namespace sim;
    init_tables();
    if (fast && !simfast::available()) return 1;
    for (long i = 0; i < n; i += 64) {
        uint32_t A[64] = {0}, B[64] = {0}, C[64] = {0}, D[64] = {0};
        const long m = n - i < 64 ? n - i : 64;
        for (long l = 0; l < m; l++) { A[l] = a[i + l]; B[l] = b[i + l]; C[l] = c[i + l]; }
        if (fast) {
            const uint64_t exec = m == 64 ? ~0ull : ((1ull << m) - 1);
            if (op == 0) simfast::pk_add_f16(D, A, 1, 0, B, 1, 0, exec);
            else if (op == 1) simfast::pk_mul_f16(D, A, 1, 0, B, 1, 0, exec);
            else if (op == 2) simfast::pk_fma_f16(D, A, 1, 0, B, 1, 0, C, 1, 0, exec);
            else if (!simfast::pk_maximum3_f16(D, A, 1, 0, B, 1, 0, C, 1, 0, exec))
                for (long l = 0; l < m; l++) D[l] = h_maximum(h_maximum((uint16_t)A[l], (uint16_t)B[l]), (uint16_t)C[l]);      // NaN operands: the scalar form, as exec_vop3p does
        } else {
            for (long l = 0; l < m; l++) {
                const uint16_t x = (uint16_t)A[l], y = (uint16_t)B[l], z = (uint16_t)C[l];
                D[l] = op == 0 ? h_add(x, y) : op == 1 ? h_mul(x, y) : op == 2 ? h_fma(x, y, z) : h_maximum(h_maximum(x, y), z);
            }
        }
        for (long l = 0; l < m; l++) out[i + l] = (uint16_t)D[l];
    }
    return 0;
}
namespace sim {

// ------------------------------------------------------------------------------------------------ parsing
static bool parse_int(const std::string& t, int64_t& v) {
    if (t.empty()) return false;
    const char* s = t.c_str();
    char* e = nullptr;
    if (t.size() > 2 && t[0] == '0' && (t[1] == 'x' || t[1] == 'X')) { v = (int64_t)strtoull(s, &e, 16); return *e == 0; }
    if (t.size() > 3 && t[0] == '-' && t[1] == '0' && (t[2] == 'x' || t[2] == 'X')) { v = -(int64_t)strtoull(s + 1, &e, 16); return *e == 0; }
    v = strtoll(s, &e, 10);
    return *e == 0;
}

static bool parse_reg_range(const std::string& t, size_t p, int& first, int& n) {    // "12" or "[4:5]"
    if (p >= t.size()) return false;
    if (t[p] == '[') {
        int a, b;
        if (sscanf(t.c_str() + p, "[%d:%d]", &a, &b) != 2) return false;
        first = a; n = b - a + 1;
        return true;
    }
    char* e = nullptr;
    first = (int)strtol(t.c_str() + p, &e, 10);
    n = 1;
    return e != t.c_str() + p && *e == 0;
}

static bool parse_operand(std::string t, Opnd& o, std::string& err) {
    o = Opnd();
    if (t.rfind("sext(", 0) == 0 && t.back() == ')') { o.sext = true; t = t.substr(5, t.size() - 6); }
    if (t.rfind("neg(", 0) == 0 && t.back() == ')') { o.neg = true; t = t.substr(4, t.size() - 5); }
    if (t.rfind("abs(", 0) == 0 && t.back() == ')') { o.abs = true; t = t.substr(4, t.size() - 5); }
    bool minus = false;
    if (t.size() > 1 && t[0] == '-' && (t[1] == 'v' || t[1] == 's' || t[1] == '|' || t[1] == 'a')) { minus = true; t = t.substr(1); }
    if (t.size() > 2 && t[0] == '|' && t.back() == '|') { o.abs = true; t = t.substr(1, t.size() - 2); }
    if (minus) o.neg = true;
    int first, n;
    if ((t[0] == 'v' || t[0] == 's' || t[0] == 'a') && t.size() > 1 && (isdigit((unsigned char)t[1]) || t[1] == '[') && parse_reg_range(t, 1, first, n)) {
        o.kind = t[0] == 's' ? K_SGPR : K_VGPR;
        o.reg = (uint16_t)(first + (t[0] == 'a' ? 256 : 0));
        o.n = (uint8_t)n;
        return true;
    }
    struct { const char* name; int reg, n; } named[] = {{"vcc", SG_VCC, 2}, {"vcc_lo", SG_VCC, 1}, {"vcc_hi", SG_VCC + 1, 1}, {"exec", SG_EXEC, 2},
        {"exec_lo", SG_EXEC, 1}, {"exec_hi", SG_EXEC + 1, 1}, {"m0", SG_M0, 1}, {"flat_scratch", 102, 2}, {"flat_scratch_lo", 102, 1},
        {"flat_scratch_hi", 103, 1}, {"xnack_mask", 104, 2}};
    for (auto& r : named)
        if (t == r.name) { o.kind = K_SGPR; o.reg = (uint16_t)r.reg; o.n = (uint8_t)r.n; return true; }
    if (t == "off" || t == "null") { o.kind = K_OFF; return true; }
    if (t == "scc" || t == "src_scc") { o.kind = K_SCC; return true; }
    if (t == "src_shared_base") { o.kind = K_IMM; o.imm = (uint64_t)SHARED_BASE_HI << 32; o.n = 2; return true; }
    if (t == "src_private_base") { o.kind = K_IMM; o.imm = (uint64_t)PRIVATE_BASE_HI << 32; o.n = 2; return true; }
    if (t == "src_shared_limit" || t == "src_private_limit") { o.kind = K_IMM; o.imm = 0xffffffffull << 32; o.n = 2; return true; }
    int64_t v;
    if (parse_int(t, v)) { o.kind = K_IMM; o.imm = (uint64_t)v; return true; }
    char* e = nullptr;
    double d = strtod(t.c_str(), &e);
    if (e != t.c_str() && *e == 0) { o.kind = K_IMM; o.isfloat = true; o.imm = f2bits((float)d); return true; }
    err = "operand '" + t + "'";
    return false;
}

static std::vector<std::string> tokenize(const std::string& s) {      // split on blanks and commas outside [] () ||
    std::vector<std::string> out;
    std::string cur;
    int depth = 0;
    for (char ch : s) {
        if (ch == '[' || ch == '(') depth++;
        if (ch == ']' || ch == ')') depth--;
        if ((ch == ',' || ch == ' ' || ch == '\t') && depth == 0) {
            if (!cur.empty()) out.push_back(cur), cur.clear();
        } else cur += ch;
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

static int sel_code(const std::string& v) {
    if (v == "DWORD") return 6;
    if (v == "WORD_0") return 4;
    if (v == "WORD_1") return 5;
    if (v.rfind("BYTE_", 0) == 0) return v[5] - '0';
    return -1;
}
static uint8_t parse_mask_list(const std::string& v) {      // "[1,0,1]" -> bit per element
    uint8_t m = 0; int i = 0;
    for (char ch : v) { if (ch == '0' || ch == '1') { if (ch == '1') m |= 1u << i; i++; } }
    return m;
}

static uint8_t parse_quad(const std::string& v) {            // "[a,b,c,d]" -> 2 bits each
    uint8_t q = 0; int i = 0;
    for (char ch : v) if (ch >= '0' && ch <= '3' && i < 4) { q |= (uint8_t)((ch - '0') << (2 * i)); i++; }
    return q;
}

static bool parse_line(const std::string& line, Inst& in, std::string& err) {
    size_t cpos = line.find("//");
    std::string body = line.substr(0, cpos);
    if (cpos != std::string::npos) in.addr = strtoull(line.c_str() + cpos + 2, nullptr, 16);
    auto toks = tokenize(body);
    if (toks.empty()) { err = "empty"; return false; }
    std::string m = toks[0];
    for (const char* suf : {"_e32", "_e64", "_dpp", "_sdwa"}) {
        size_t L = strlen(suf);
        if (m.size() > L && m.compare(m.size() - L, L, suf) == 0) {
            if (!strcmp(suf, "_dpp")) in.enc = 1;
            if (!strcmp(suf, "_sdwa")) in.enc = 2;
            m.erase(m.size() - L);
            break;
        }
    }
    if (m.rfind("v_cmp_", 0) == 0) {
        char rel[8] = {0}, ty[8] = {0};
        if (sscanf(m.c_str(), "v_cmp_%7[a-z]_%7[a-z0-9]", rel, ty) != 2) { err = "mnemonic " + m; return false; }
        static const char* rels[] = {"f", "lt", "eq", "le", "gt", "ne", "ge", "t"};
        static const char* tys[] = {"i32", "u32", "i64", "u64", "u16", "i16", "f32", "f16"};
        int r = -1, t = -1;
        std::string R = rel;
        if (R == "lg") R = "ne";
        for (int i = 0; i < 8; i++) { if (R == rels[i]) r = i; if (!strcmp(ty, tys[i])) t = i; }
        if (r < 0 || t < 0) { err = "mnemonic " + m; return false; }
        in.op = OP_v_cmp; in.simm = r; in.bitop3 = (uint8_t)t;
    } else {
        auto it = op_table().find(m);
        if (it == op_table().end()) { err = "unknown mnemonic " + m; return false; }
        in.op = it->second;
    }
    in.cls = op_cls[in.op];
    if (in.op == OP_s_waitcnt || in.op == OP_s_nop || in.op == OP_buffer_wbl2 || in.op == OP_buffer_inv || in.op == OP_s_sleep || in.op == OP_s_setprio)
        return true;
    for (size_t i = 1; i < toks.size(); i++) {
        const std::string& t = toks[i];
        size_t colon = t.find(':');
        bool is_mod = false;
        if (colon != std::string::npos && colon > 0 && t.find('[') > colon) {
            is_mod = true;
            for (size_t k = 0; k < colon; k++) if (!(isalnum((unsigned char)t[k]) || t[k] == '_')) is_mod = false;
        }
        if (is_mod) {
            std::string k = t.substr(0, colon), v = t.substr(colon + 1);
            int64_t iv = 0;
            bool isnum = parse_int(v, iv);
            if (k == "offset") in.offset = (int32_t)iv;
            else if (k == "offset0") in.offset = (int32_t)iv;
            else if (k == "offset1") in.offset1 = (int32_t)iv;
            else if (k == "row_shl") { in.dpp_kind = 1; in.dpp_n = (uint8_t)iv; }
            else if (k == "row_shr") { in.dpp_kind = 2; in.dpp_n = (uint8_t)iv; }
            else if (k == "row_ror") { in.dpp_kind = 3; in.dpp_n = (uint8_t)iv; }
            else if (k == "row_bcast") { in.dpp_kind = iv == 15 ? 4 : 5; in.dpp_n = (uint8_t)iv; }
            else if (k == "wave_shl") { in.dpp_kind = 6; in.dpp_n = 1; }
            else if (k == "wave_shr") { in.dpp_kind = 7; in.dpp_n = 1; }
            else if (k == "wave_rol") { in.dpp_kind = 8; in.dpp_n = 1; }
            else if (k == "wave_ror") { in.dpp_kind = 9; in.dpp_n = 1; }
            else if (k == "quad_perm") { in.dpp_kind = 10; in.dpp_n = parse_quad(v); }
            else if (k == "row_mask") in.row_mask = (uint8_t)iv;
            else if (k == "bank_mask") in.bank_mask = (uint8_t)iv;
            else if (k == "bound_ctrl") in.bound_ctrl = true;      // both spellings (bound_ctrl:0 of old, :1 of new) mean "read 0"
            else if (k == "dst_sel") in.dst_sel = (uint8_t)sel_code(v);
            else if (k == "src0_sel") in.src0_sel = (uint8_t)sel_code(v);
            else if (k == "src1_sel") in.src1_sel = (uint8_t)sel_code(v);
            else if (k == "dst_unused") in.dst_unused = v == "UNUSED_PAD" ? 0 : v == "UNUSED_SEXT" ? 1 : 2;
            else if (k == "neg_lo") in.neg_lo = parse_mask_list(v);
            else if (k == "neg_hi") in.neg_hi = parse_mask_list(v);
            else if (k == "bitop3") in.bitop3 = (uint8_t)iv;
            else { err = "modifier " + t; return false; }
            (void)isnum;
            continue;
        }
        if (t == "sc0" || t == "glc") { in.sc0 = true; continue; }
        if (t == "sc1" || t == "nt" || t == "slc" || t == "dlc") continue;
        if (t == "clamp") { in.clamp = true; continue; }
        if (in.nops >= 6) { err = "too many operands"; return false; }
        if (!parse_operand(t, in.o[in.nops], err)) return false;
        in.nops++;
    }
    switch (in.op) {
        case OP_s_branch: case OP_s_cbranch_scc0: case OP_s_cbranch_scc1: case OP_s_cbranch_vccz: case OP_s_cbranch_vccnz:
        case OP_s_cbranch_execz: case OP_s_cbranch_execnz:
            in.simm = (int16_t)(in.o[0].imm & 0xffff);
            break;
        case OP_s_movk_i32: case OP_s_addk_i32: case OP_s_mulk_i32:
            in.o[1].imm = (uint64_t)(int64_t)(int16_t)(in.o[1].imm & 0xffff);
            break;
        case OP_s_cmpk_gt_i32: case OP_s_cmpk_lt_i32: case OP_s_cmpk_eq_i32: case OP_s_cmpk_lg_i32: case OP_s_cmpk_ge_i32: case OP_s_cmpk_le_i32:
            in.o[1].imm = (uint64_t)(int64_t)(int16_t)(in.o[1].imm & 0xffff);
            break;
        case OP_s_cmpk_gt_u32: case OP_s_cmpk_lt_u32: case OP_s_cmpk_eq_u32: case OP_s_cmpk_lg_u32: case OP_s_cmpk_ge_u32: case OP_s_cmpk_le_u32:
            in.o[1].imm &= 0xffff;
            break;
        default: break;
    }
    return true;
}

// one symbol's instructions appended to k.code; the symbol is named (want_name) or is the one that contains want_addr
static bool parse_symbol(Kernel& k, const std::string& want_name, uint64_t want_addr, std::string& err) {
    init_tables();
    std::ifstream f(k.sfile);
    if (!f) { err = "cannot open " + k.sfile; return false; }
    std::string want = " <" + want_name + ">:";
    std::string line;
    uint32_t ln = 0;
    bool in_k = false;
    const size_t first = k.code.size();
    // by address: remember the last label at or below the address, then parse from there (second pass)
    if (want_name.empty()) {
        std::string best;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '\t' || !isxdigit((unsigned char)line[0]) || line.back() != ':') continue;
            uint64_t a = strtoull(line.c_str(), nullptr, 16);
            size_t lt = line.find('<');
            if (lt == std::string::npos) continue;
            if (a <= want_addr) best = line.substr(lt + 1, line.size() - lt - 3);
        }
        if (best.empty()) { err = "no symbol contains address " + std::to_string(want_addr); return false; }
        return parse_symbol(k, best, 0, err);
    }
    while (std::getline(f, line)) {
        ln++;
        if (!in_k) {
            if (!line.empty() && line[0] != '\t' && line.size() > want.size() && line.compare(line.size() - want.size(), want.size(), want) == 0) in_k = true;
            continue;
        }
        if (line.empty()) continue;
        if (line[0] != '\t') {
            if (isxdigit((unsigned char)line[0]) && line.back() == ':') break;      // the next symbol
            continue;
        }
        if (line.find("...") != std::string::npos && line.find("//") == std::string::npos) continue;
        Inst in;
        in.line = ln;
        std::string e;
        if (!parse_line(line, in, e)) { err = want_name + ": co.s:" + std::to_string(ln) + ": " + e + " in '" + line + "'"; return false; }
        k.at[in.addr] = (int)k.code.size();
        k.code.push_back(in);
    }
    if (k.code.size() == first) { err = "symbol " + want_name + " not found in " + k.sfile; return false; }
    for (size_t i = first; i < k.code.size(); i++) {
        Inst& in = k.code[i];
        if (in.cls != C_BRANCH || in.op == OP_s_swappc_b64 || in.op == OP_s_setpc_b64) continue;
        uint64_t tgt = in.addr + 4 + (int64_t)in.simm * 4;
        auto it = k.at.find(tgt);
        if (it == k.at.end()) { err = want_name + ": branch target outside the symbol"; return false; }
        in.target = it->second;
    }
    return true;
}

bool parse_kernel(Kernel& k, std::string& err) {
    if (k.parsed) return true;
    if (!parse_symbol(k, k.name, 0, err)) return false;
    // a kernel that calls: every other symbol of the code object is parsed NOW (launches are serialised; the workers of a
    // launch only ever read k.code)
    bool calls = false;
    for (const Inst& in : k.code) calls |= in.op == OP_s_swappc_b64;
    if (calls) {
        std::vector<std::string> names;
        std::ifstream f(k.sfile);
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '\t' || !isxdigit((unsigned char)line[0]) || line.back() != ':') continue;
            size_t lt = line.find('<');
            if (lt == std::string::npos) continue;
            uint64_t a = strtoull(line.c_str(), nullptr, 16);
            if (!k.at.count(a)) names.push_back(line.substr(lt + 1, line.size() - lt - 3));
        }
        for (const std::string& n : names) {
            std::string e;
            (void)parse_symbol(k, n, 0, e);       // a symbol the interpreter cannot parse only matters if it is called
        }
    }
    k.parsed = true;
    return true;
}

int parse_function_at(Kernel& k, uint64_t addr, std::string& err) {
    auto it = k.at.find(addr);
    if (it != k.at.end()) return it->second;
    err = "no parsed instruction at that address (parse_kernel reads every symbol of a code object whose kernel calls)";
    return -1;
}

// ------------------------------------------------------------------------------------------------ execution
struct Wave {
    uint32_t s[128];
    uint32_t (*v)[64] = nullptr;
    bool scc = false;
    int pc = 0;
    int state = 0;                     // 0 running, 1 at a barrier, 2 ended
    uint8_t* scratch = nullptr;        // 64 lanes x scratch bytes
};

struct Ctx {
    Kernel* k;
    std::vector<uint8_t> lds;
    std::vector<uint8_t> lds_def;      // HIPSIM_LDS_UNDEF=1: 1 = some lane of this workgroup has written the byte
    uint32_t scratch_bytes = 0;
    uint64_t steps = 0;
    int cur = -1;                      // the wave being stepped (fault attribution)
    uint64_t count[C_N] = {0};
};

static inline uint64_t get64(const uint32_t* p) { return (uint64_t)p[0] | ((uint64_t)p[1] << 32); }
static inline void set64(uint32_t* p, uint64_t v) { p[0] = (uint32_t)v; p[1] = (uint32_t)(v >> 32); }
#define EXEC(w) get64(&(w).s[SG_EXEC])
#define VCC(w) get64(&(w).s[SG_VCC])

static inline uint32_t rs32(const Wave& w, const Opnd& o) {
    switch (o.kind) {
        case K_SGPR: return w.s[o.reg];
        case K_IMM: return (uint32_t)o.imm;
        case K_SCC: return w.scc;
        default: return 0;
    }
}
static inline uint64_t rs64(const Wave& w, const Opnd& o) {
    switch (o.kind) {
        case K_SGPR: return o.n >= 2 ? get64(&w.s[o.reg]) : w.s[o.reg];
        case K_IMM: return o.imm;
        case K_SCC: return w.scc;
        default: return 0;
    }
}
static inline void ws32(Wave& w, const Opnd& o, uint32_t v) { w.s[o.reg] = v; }
static inline void ws64(Wave& w, const Opnd& o, uint64_t v) { set64(&w.s[o.reg], v); }

struct Src { const uint32_t* lo; const uint32_t* hi; int st; uint32_t c[2]; };
static inline void mk_src(const Wave& w, const Opnd& o, Src& s) {
    if (o.kind == K_VGPR) { s.lo = w.v[o.reg]; s.hi = w.v[o.reg + (o.n > 1 ? 1 : 0)]; s.st = 1; return; }
    uint64_t x = rs64(w, o);
    s.c[0] = (uint32_t)x; s.c[1] = (uint32_t)(x >> 32);
    s.lo = &s.c[0]; s.hi = &s.c[1]; s.st = 0;
}
#define L32(S, l) ((S).lo[(l) * (S).st])
#define L64(S, l) ((uint64_t)(S).lo[(l) * (S).st] | ((uint64_t)(S).hi[(l) * (S).st] << 32))

static inline uint32_t fmod32(uint32_t x, const Opnd& o) {
    if (o.abs) x &= 0x7fffffffu;
    if (o.neg) x ^= 0x80000000u;
    return x;
}

static inline uint32_t sdwa_sel(uint32_t x, int sel, bool sext) {
    if (sel == 6) return x;
    if (sel >= 4) { uint32_t h = (x >> ((sel - 4) * 16)) & 0xffff; return sext ? (uint32_t)(int32_t)(int16_t)h : h; }
    uint32_t b = (x >> (sel * 8)) & 0xff;
    return sext ? (uint32_t)(int32_t)(int8_t)b : b;
}
static inline uint32_t sdwa_dst(uint32_t old, uint32_t r, int sel, int unused) {
    if (sel == 6) return r;
    int sh, bits;
    if (sel >= 4) { sh = (sel - 4) * 16; bits = 16; } else { sh = sel * 8; bits = 8; }
    uint32_t mask = ((1u << bits) - 1) << sh, val = (r << sh) & mask;
    if (unused == 2) return (old & ~mask) | val;
    if (unused == 1) {      // sign-extend above, zero below
        uint32_t out = val;
        if ((r >> (bits - 1)) & 1) out |= ~(mask | ((1u << sh) - 1));
        return out;
    }
    return val;
}

static inline int32_t cvt_i32_f32(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
static inline uint32_t cvt_u32_f32(float f) {
    if (f != f || f <= 0.0f) return 0;
    if (f >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)f;
}
static inline uint32_t f2h_bits(float f) { return d2h((double)f); }

static inline uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t selw) {
    uint64_t both = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t sel = (selw >> (8 * i)) & 0xff, b;
        if (sel <= 7) b = (uint32_t)(both >> (8 * sel)) & 0xff;
        else if (sel <= 11) { int byte = (int)(sel - 8) * 2 + 1; b = ((both >> (8 * byte + 7)) & 1) ? 0xff : 0; }
        else if (sel == 12) b = 0;
        else b = 0xff;
        r |= b << (8 * i);
    }
    return r;
}

// one lane of the one-dword ALU
static inline uint32_t alu32(const Inst& in, uint32_t a, uint32_t b, uint32_t c, int lane) {
    switch (in.op) {
        case OP_v_mov_b32: case OP_v_accvgpr_read_b32: case OP_v_accvgpr_write_b32: return a;
        case OP_v_not_b32: return ~a;
        case OP_v_bfrev_b32: { uint32_t r = 0; for (int i = 0; i < 32; i++) if (a & (1u << i)) r |= 1u << (31 - i); return r; }
        case OP_v_cvt_f32_i32: return f2bits((float)(int32_t)a);
        case OP_v_cvt_f32_u32: return f2bits((float)a);
        case OP_v_cvt_i32_f32: return (uint32_t)cvt_i32_f32(bits2f(a));
        case OP_v_cvt_u32_f32: return cvt_u32_f32(bits2f(a));
        case OP_v_cvt_f32_f16: return f2bits(g_h2f[a & 0xffff]);
        case OP_v_cvt_f16_f32: return f2h_bits(bits2f(a));
        case OP_v_add_u32: return a + b;
        case OP_v_sub_u32: return a - b;
        case OP_v_subrev_u32: return b - a;
        case OP_v_and_b32: return a & b;
        case OP_v_or_b32: return a | b;
        case OP_v_xor_b32: return a ^ b;
        case OP_v_lshlrev_b32: return b << (a & 31);
        case OP_v_lshrrev_b32: return b >> (a & 31);
        case OP_v_ashrrev_i32: return (uint32_t)((int32_t)b >> (a & 31));
        case OP_v_max_i32: return (uint32_t)std::max((int32_t)a, (int32_t)b);
        case OP_v_min_i32: return (uint32_t)std::min((int32_t)a, (int32_t)b);
        case OP_v_max_u32: return std::max(a, b);
        case OP_v_min_u32: return std::min(a, b);
        case OP_v_mul_u32_u24: return (a & 0xffffff) * (b & 0xffffff);
        case OP_v_mul_i32_i24: return (uint32_t)(((int32_t)(a << 8) >> 8) * ((int32_t)(b << 8) >> 8));
        case OP_v_mul_lo_u32: return a * b;
        case OP_v_mul_hi_u32: return (uint32_t)(((uint64_t)a * b) >> 32);
        case OP_v_mul_hi_i32: return (uint32_t)(((int64_t)(int32_t)a * (int32_t)b) >> 32);
        case OP_v_add_f32: return f2bits(bits2f(a) + bits2f(b));
        case OP_v_sub_f32: return f2bits(bits2f(a) - bits2f(b));
        case OP_v_mul_f32: return f2bits(bits2f(a) * bits2f(b));
        case OP_v_mul_lo_u16: return (a * b) & 0xffff;
        case OP_v_sub_u16: return (a - b) & 0xffff;
        case OP_v_add_u16: return (a + b) & 0xffff;
        case OP_v_lshlrev_b16: return (b << (a & 15)) & 0xffff;
        case OP_v_lshrrev_b16: return (b & 0xffff) >> (a & 15);
        case OP_v_max_u16: return std::max(a & 0xffff, b & 0xffff);
        case OP_v_min_u16: return std::min(a & 0xffff, b & 0xffff);
        case OP_v_bcnt_u32_b32: return (uint32_t)__builtin_popcount(a) + b;
        case OP_v_mbcnt_lo_u32_b32: { uint32_t m = lane >= 32 ? 0xffffffffu : ((1u << lane) - 1); return (uint32_t)__builtin_popcount(a & m) + b; }
        case OP_v_mbcnt_hi_u32_b32: { uint32_t m = lane <= 32 ? 0 : ((1u << (lane - 32)) - 1); return (uint32_t)__builtin_popcount(a & m) + b; }
        case OP_v_mad_u32_u24: return (a & 0xffffff) * (b & 0xffffff) + c;
        case OP_v_mad_i32_i24: return (uint32_t)(((int32_t)(a << 8) >> 8) * ((int32_t)(b << 8) >> 8)) + c;
        case OP_v_lshl_add_u32: return (a << (b & 31)) + c;
        case OP_v_add_lshl_u32: return (a + b) << (c & 31);
        case OP_v_lshl_or_b32: return (a << (b & 31)) | c;
        case OP_v_and_or_b32: return (a & b) | c;
        case OP_v_or3_b32: return a | b | c;
        case OP_v_add3_u32: return a + b + c;
        case OP_v_xad_u32: return (a ^ b) + c;
        case OP_v_max3_i32: return (uint32_t)std::max(std::max((int32_t)a, (int32_t)b), (int32_t)c);
        case OP_v_min3_i32: return (uint32_t)std::min(std::min((int32_t)a, (int32_t)b), (int32_t)c);
        case OP_v_max3_u32: return std::max(std::max(a, b), c);
        case OP_v_min3_u32: return std::min(std::min(a, b), c);
        case OP_v_med3_i32: { int32_t x = (int32_t)a, y = (int32_t)b, z = (int32_t)c; return (uint32_t)std::max(std::min(x, y), std::min(std::max(x, y), z)); }
        case OP_v_bfe_u32: { uint32_t off = b & 31, w = c & 31; return w ? (a >> off) & ((1u << w) - 1) : 0; }
        case OP_v_bfe_i32: { uint32_t off = b & 31, w = c & 31; if (!w) return 0; uint32_t x = (a >> off) & ((1u << w) - 1); if (x & (1u << (w - 1))) x |= ~((1u << w) - 1); return x; }
        case OP_v_bfi_b32: return (a & b) | (~a & c);
        case OP_v_alignbit_b32: return (uint32_t)((((uint64_t)a << 32) | b) >> (c & 31));
        case OP_v_perm_b32: return perm_b32(a, b, c);
        case OP_v_bitop3_b32: {
            uint32_t r = 0, tt = in.bitop3;
            for (int i = 0; i < 8; i++) if (tt & (1u << i)) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
            return r;
        }
        default: throw Fault(std::string("alu32: no semantics for ") + op_name[in.op]);
    }
}
static inline bool is_f32_op(uint16_t op) {
    return op == OP_v_add_f32 || op == OP_v_sub_f32 || op == OP_v_mul_f32 || op == OP_v_cvt_i32_f32 || op == OP_v_cvt_u32_f32 || op == OP_v_cvt_f16_f32;
}
static inline bool is_b16_op(uint16_t op) {
    return op == OP_v_mul_lo_u16 || op == OP_v_sub_u16 || op == OP_v_add_u16 || op == OP_v_lshlrev_b16 || op == OP_v_lshrrev_b16 || op == OP_v_max_u16 || op == OP_v_min_u16;
}

static inline bool dpp_source(const Inst& in, int lane, uint64_t exec, int& srcl) {     // false: the source lane is invalid
    int row = lane & ~15, i = lane & 15, j;
    switch (in.dpp_kind) {
        case 1: j = i + in.dpp_n; if (j > 15) return false; break;          // row_shl: lane i reads lane i + n
        case 2: j = i - in.dpp_n; if (j < 0) return false; break;           // row_shr: lane i reads lane i - n
        case 3: j = (i - in.dpp_n) & 15; break;                             // row_ror
        case 4: if (row == 0) return false; srcl = row - 1; return (exec >> srcl) & 1;          // row_bcast:15: lane 15 of the row before
        case 5: if (row < 32) return false; srcl = 31; return (exec >> srcl) & 1;               // row_bcast:31: lane 31 to rows 2 and 3
        case 6: if (lane == 63) return false; srcl = lane + 1; return (exec >> srcl) & 1;       // wave_shl:1
        case 7: if (lane == 0) return false; srcl = lane - 1; return (exec >> srcl) & 1;        // wave_shr:1
        case 8: srcl = (lane + 1) & 63; return (exec >> srcl) & 1;                              // wave_rol:1
        case 9: srcl = (lane - 1) & 63; return (exec >> srcl) & 1;                              // wave_ror:1
        case 10: srcl = (lane & ~3) + ((in.dpp_n >> (2 * (lane & 3))) & 3); return (exec >> srcl) & 1;   // quad_perm
        default: throw Fault("dpp control not modelled");
    }
    srcl = row + j;
    return (exec >> srcl) & 1;         // gfx9 has no fetch-inactive bit: a disabled source lane is invalid
}

static bool g_lds_undef = false;
static uint8_t* lds_at(Ctx& c, uint64_t a, uint32_t n, const Inst& in, int lane) {
    if (g_lds_undef && a + n <= c.lds.size()) {
        const char* nm = op_name[in.op];
        const bool wr = strstr(nm, "write") != nullptr || !strncmp(nm, "flat_store", 10);
        const bool rmw = !strncmp(nm, "ds_add", 6) || !strncmp(nm, "ds_max", 6) || !strncmp(nm, "ds_or", 5);
        if (wr) memset(&c.lds_def[a], 1, n);
        else {
            for (uint32_t i = 0; i < n; i++) if (!c.lds_def[a + i]) {
                char buf[256];
                snprintf(buf, sizeof buf, "read of LDS byte %llu that no lane of the workgroup has written (%s, %u bytes at %llu, lane %d)", (unsigned long long)(a + i), nm, n, (unsigned long long)a, lane);
                throw Fault(buf);
            }
            if (rmw) memset(&c.lds_def[a], 1, n);
        }
    }
    if (a + n > c.lds.size()) {
        char buf[256];
        snprintf(buf, sizeof buf, "LDS access out of range: address %llu + %u > %zu bytes (%s, lane %d)", (unsigned long long)a, n, c.lds.size(), op_name[in.op], lane);
        throw Fault(buf);
    }
    return c.lds.data() + a;
}
static uint8_t* glob_at(uint64_t a, uint32_t n, const Inst& in, int lane, bool write) {
    if (!mem_ok(a, n)) {
        char buf[512];
        snprintf(buf, sizeof buf, "global %s out of bounds: %u bytes at 0x%llx (%s, lane %d): %s", write ? "store" : "load", n, (unsigned long long)a, op_name[in.op], lane,
                 mem_describe(a).c_str());
        throw Fault(buf);
    }
    if (uint8_t* sh = mem_shadow(a)) {
        if (write) memset(sh, 1, n);
        else for (uint32_t i = 0; i < n; i++) if (!sh[i]) {
            char buf[512];
            snprintf(buf, sizeof buf, "load of device memory nothing has written: byte %u of %u at 0x%llx (%s, lane %d): %s", i, n, (unsigned long long)a, op_name[in.op], lane,
                     mem_describe(a).c_str());
            throw Fault(buf);
        }
    }
    return (uint8_t*)(uintptr_t)a;
}
static uint8_t* scratch_at(Ctx& c, Wave& w, uint64_t a, uint32_t n, const Inst& in, int lane) {
    if (a + n > c.scratch_bytes) {
        char buf[256];
        snprintf(buf, sizeof buf, "scratch access out of range: offset %llu + %u > %u bytes (%s, lane %d)", (unsigned long long)a, n, c.scratch_bytes, op_name[in.op], lane);
        throw Fault(buf);
    }
    return w.scratch + (size_t)lane * c.scratch_bytes + a;
}
// a flat / global / scratch address of one lane resolved to host memory
static uint8_t* vmem_ptr(Ctx& c, Wave& w, const Inst& in, int space, int lane, const Opnd& vaddr, const Opnd& saddr, uint32_t n, bool write) {
    if (space == 2) {          // scratch: saddr (sgpr | off) + vaddr (vgpr | off) + offset, private to the lane
        uint64_t a = (uint64_t)(int64_t)in.offset;
        if (saddr.kind == K_SGPR) a += w.s[saddr.reg];
        if (vaddr.kind == K_VGPR) a += w.v[vaddr.reg][lane];
        return scratch_at(c, w, a, n, in, lane);
    }
    uint64_t a;
    if (saddr.kind == K_SGPR) a = get64(&w.s[saddr.reg]) + w.v[vaddr.reg][lane];
    else a = (uint64_t)w.v[vaddr.reg][lane] | ((uint64_t)w.v[vaddr.reg + 1][lane] << 32);
    a += (uint64_t)(int64_t)in.offset;
    if (space == 1) {          // flat: apertures
        uint32_t hi = (uint32_t)(a >> 32);
        if (hi == SHARED_BASE_HI) return lds_at(c, a & 0xffffffffu, n, in, lane);
        if (hi == PRIVATE_BASE_HI) return scratch_at(c, w, a & 0xffffffffu, n, in, lane);
    }
    return glob_at(a, n, in, lane, write);
}

static void exec_vmem(Ctx& c, Wave& w, const Inst& in) {
    const uint64_t exec = EXEC(w);
    const char* nm = op_name[in.op];
    int space = nm[0] == 'g' ? 0 : nm[0] == 'f' ? 1 : 2;
    const char* kind = strchr(nm, '_') + 1;                 // "load_dword", "store_byte", "atomic_add" ...
    if (!strncmp(kind, "load_", 5)) {
        const Opnd& dst = in.o[0];
        const Opnd& va = in.o[1];
        static const Opnd none;
        const Opnd& sa = space == 1 ? none : in.o[2];
        const char* t = kind + 5;
        for (int l = 0; l < 64; l++) {
            if (!((exec >> l) & 1)) continue;
            if (!strncmp(t, "dword", 5)) {
                int nd = t[5] == 'x' ? t[6] - '0' : 1;
                uint8_t* p = vmem_ptr(c, w, in, space, l, va, sa, 4 * nd, false);
                uint32_t tmp[4];
                memcpy(tmp, p, 4 * nd);
                for (int i = 0; i < nd; i++) w.v[dst.reg + i][l] = tmp[i];
            } else if (!strcmp(t, "ubyte")) w.v[dst.reg][l] = *vmem_ptr(c, w, in, space, l, va, sa, 1, false);
            else if (!strcmp(t, "sbyte")) w.v[dst.reg][l] = (uint32_t)(int32_t)(int8_t)*vmem_ptr(c, w, in, space, l, va, sa, 1, false);
            else if (!strcmp(t, "ushort")) { uint16_t x; memcpy(&x, vmem_ptr(c, w, in, space, l, va, sa, 2, false), 2); w.v[dst.reg][l] = x; }
            else if (!strcmp(t, "sshort")) { int16_t x; memcpy(&x, vmem_ptr(c, w, in, space, l, va, sa, 2, false), 2); w.v[dst.reg][l] = (uint32_t)(int32_t)x; }
            else if (!strcmp(t, "short_d16")) { uint16_t x; memcpy(&x, vmem_ptr(c, w, in, space, l, va, sa, 2, false), 2); w.v[dst.reg][l] = (w.v[dst.reg][l] & 0xffff0000u) | x; }
            else if (!strcmp(t, "short_d16_hi")) { uint16_t x; memcpy(&x, vmem_ptr(c, w, in, space, l, va, sa, 2, false), 2); w.v[dst.reg][l] = (w.v[dst.reg][l] & 0xffffu) | ((uint32_t)x << 16); }
            else if (!strcmp(t, "ubyte_d16")) { uint8_t x = *vmem_ptr(c, w, in, space, l, va, sa, 1, false); w.v[dst.reg][l] = (w.v[dst.reg][l] & 0xffff0000u) | x; }
            else if (!strcmp(t, "ubyte_d16_hi")) { uint8_t x = *vmem_ptr(c, w, in, space, l, va, sa, 1, false); w.v[dst.reg][l] = (w.v[dst.reg][l] & 0xffffu) | ((uint32_t)x << 16); }
            else throw Fault(std::string("load form not modelled: ") + nm);
        }
        return;
    }
    if (!strncmp(kind, "store_", 6)) {
        const Opnd& va = in.o[0];
        const Opnd& data = in.o[1];
        static const Opnd none;
        const Opnd& sa = space == 1 ? none : in.o[2];
        const char* t = kind + 6;
        for (int l = 0; l < 64; l++) {
            if (!((exec >> l) & 1)) continue;
            if (!strncmp(t, "dword", 5)) {
                int nd = t[5] == 'x' ? t[6] - '0' : 1;
                uint32_t tmp[4];
                for (int i = 0; i < nd; i++) tmp[i] = w.v[data.reg + i][l];
                memcpy(vmem_ptr(c, w, in, space, l, va, sa, 4 * nd, true), tmp, 4 * nd);
            } else if (!strcmp(t, "byte")) *vmem_ptr(c, w, in, space, l, va, sa, 1, true) = (uint8_t)w.v[data.reg][l];
            else if (!strcmp(t, "byte_d16_hi")) *vmem_ptr(c, w, in, space, l, va, sa, 1, true) = (uint8_t)(w.v[data.reg][l] >> 16);
            else if (!strcmp(t, "short")) { uint16_t x = (uint16_t)w.v[data.reg][l]; memcpy(vmem_ptr(c, w, in, space, l, va, sa, 2, true), &x, 2); }
            else if (!strcmp(t, "short_d16_hi")) { uint16_t x = (uint16_t)(w.v[data.reg][l] >> 16); memcpy(vmem_ptr(c, w, in, space, l, va, sa, 2, true), &x, 2); }
            else throw Fault(std::string("store form not modelled: ") + nm);
        }
        return;
    }
    if (!strncmp(kind, "atomic_", 7)) {
        // returning form: vdst, vaddr, vdata, saddr ; non-returning: vaddr, vdata, saddr (flat: no saddr)
        int nop_noret = space == 1 ? 2 : 3;
        bool ret = in.nops == nop_noret + 1;
        const Opnd& dst = in.o[0];
        const Opnd& va = in.o[ret ? 1 : 0];
        const Opnd& data = in.o[ret ? 2 : 1];
        static const Opnd none;
        const Opnd& sa = space == 1 ? none : in.o[ret ? 3 : 2];
        const char* t = kind + 7;
        bool x2 = strstr(t, "_x2") != nullptr;
        for (int l = 0; l < 64; l++) {
            if (!((exec >> l) & 1)) continue;
            uint8_t* p = vmem_ptr(c, w, in, space, l, va, sa, x2 ? 8 : 4, true);
            if (x2) {
                if (((uintptr_t)p & 7) != 0) throw Fault("misaligned 64-bit atomic");
                uint64_t* ap = (uint64_t*)p;
                uint64_t d = (uint64_t)w.v[data.reg][l] | ((uint64_t)w.v[data.reg + 1][l] << 32), old;
                if (!strcmp(t, "add_x2")) old = __atomic_fetch_add(ap, d, __ATOMIC_SEQ_CST);
                else {
                    old = __atomic_load_n(ap, __ATOMIC_SEQ_CST);
                    for (;;) {
                        uint64_t nv;
                        if (!strcmp(t, "smax_x2")) nv = (uint64_t)std::max((int64_t)old, (int64_t)d);
                        else if (!strcmp(t, "umax_x2")) nv = std::max(old, d);
                        else throw Fault(std::string("atomic not modelled: ") + nm);
                        if (__atomic_compare_exchange_n(ap, &old, nv, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) break;
                    }
                }
                if (ret) { w.v[dst.reg][l] = (uint32_t)old; w.v[dst.reg + 1][l] = (uint32_t)(old >> 32); }
            } else {
                if (((uintptr_t)p & 3) != 0) throw Fault("misaligned atomic");
                uint32_t* ap = (uint32_t*)p;
                uint32_t d = w.v[data.reg][l], old;
                if (!strcmp(t, "add")) old = __atomic_fetch_add(ap, d, __ATOMIC_SEQ_CST);
                else if (!strcmp(t, "sub")) old = __atomic_fetch_sub(ap, d, __ATOMIC_SEQ_CST);
                else if (!strcmp(t, "or")) old = __atomic_fetch_or(ap, d, __ATOMIC_SEQ_CST);
                else if (!strcmp(t, "and")) old = __atomic_fetch_and(ap, d, __ATOMIC_SEQ_CST);
                else if (!strcmp(t, "swap")) old = __atomic_exchange_n(ap, d, __ATOMIC_SEQ_CST);
                else {
                    old = __atomic_load_n(ap, __ATOMIC_SEQ_CST);
                    for (;;) {
                        uint32_t nv;
                        if (!strcmp(t, "smax")) nv = (uint32_t)std::max((int32_t)old, (int32_t)d);
                        else if (!strcmp(t, "umax")) nv = std::max(old, d);
                        else if (!strcmp(t, "smin")) nv = (uint32_t)std::min((int32_t)old, (int32_t)d);
                        else if (!strcmp(t, "umin")) nv = std::min(old, d);
                        else if (!strcmp(t, "inc")) nv = old >= d ? 0 : old + 1;
                        else if (!strcmp(t, "cmpswap")) nv = old == w.v[data.reg + 1][l] ? d : old;
                        else throw Fault(std::string("atomic not modelled: ") + nm);
                        if (__atomic_compare_exchange_n(ap, &old, nv, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) break;
                    }
                }
                if (ret) w.v[dst.reg][l] = old;
            }
        }
        return;
    }
    throw Fault(std::string("memory form not modelled: ") + nm);
}

static void exec_ds(Ctx& c, Wave& w, const Inst& in) {
    const uint64_t exec = EXEC(w);
    auto rd = [&](int l, uint32_t a, uint32_t n) { return lds_at(c, a, n, in, l); };
    switch (in.op) {
        case OP_ds_read_b32: case OP_ds_read_b64: case OP_ds_read_b128: {
            int nd = in.op == OP_ds_read_b32 ? 1 : in.op == OP_ds_read_b64 ? 2 : 4;
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint32_t tmp[4];
                memcpy(tmp, rd(l, w.v[in.o[1].reg][l] + (uint32_t)in.offset, 4 * nd), 4 * nd);
                for (int i = 0; i < nd; i++) w.v[in.o[0].reg + i][l] = tmp[i];
            }
            return;
        }
        case OP_ds_read_u8: case OP_ds_read_i8: case OP_ds_read_u16: case OP_ds_read_i16:
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint32_t a = w.v[in.o[1].reg][l] + (uint32_t)in.offset;
                if (in.op == OP_ds_read_u8) w.v[in.o[0].reg][l] = *rd(l, a, 1);
                else if (in.op == OP_ds_read_i8) w.v[in.o[0].reg][l] = (uint32_t)(int32_t)(int8_t)*rd(l, a, 1);
                else { uint16_t x; memcpy(&x, rd(l, a, 2), 2); w.v[in.o[0].reg][l] = in.op == OP_ds_read_u16 ? x : (uint32_t)(int32_t)(int16_t)x; }
            }
            return;
        case OP_ds_read2_b32: case OP_ds_read2st64_b32: case OP_ds_read2_b64: {
            uint32_t unit = in.op == OP_ds_read2_b64 ? 8 : 4, mul = in.op == OP_ds_read2st64_b32 ? 64 : 1;
            int nd = unit / 4;
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint32_t base = w.v[in.o[1].reg][l], tmp[4];
                memcpy(tmp, rd(l, base + (uint32_t)in.offset * unit * mul, unit), unit);
                memcpy(tmp + nd, rd(l, base + (uint32_t)in.offset1 * unit * mul, unit), unit);
                for (int i = 0; i < 2 * nd; i++) w.v[in.o[0].reg + i][l] = tmp[i];
            }
            return;
        }
        case OP_ds_write_b8: case OP_ds_write_b16: case OP_ds_write_b32: case OP_ds_write_b64: case OP_ds_write_b128: {
            uint32_t n = in.op == OP_ds_write_b8 ? 1 : in.op == OP_ds_write_b16 ? 2 : in.op == OP_ds_write_b32 ? 4 : in.op == OP_ds_write_b64 ? 8 : 16;
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint32_t tmp[4];
                for (uint32_t i = 0; i < (n + 3) / 4; i++) tmp[i] = w.v[in.o[1].reg + i][l];
                memcpy(rd(l, w.v[in.o[0].reg][l] + (uint32_t)in.offset, n), tmp, n);
            }
            return;
        }
        case OP_ds_write2_b32: case OP_ds_write2st64_b32: case OP_ds_write2_b64: {
            uint32_t unit = in.op == OP_ds_write2_b64 ? 8 : 4, mul = in.op == OP_ds_write2st64_b32 ? 64 : 1;
            int nd = unit / 4;
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint32_t base = w.v[in.o[0].reg][l], t0[2], t1[2];
                for (int i = 0; i < nd; i++) { t0[i] = w.v[in.o[1].reg + i][l]; t1[i] = w.v[in.o[2].reg + i][l]; }
                memcpy(rd(l, base + (uint32_t)in.offset * unit * mul, unit), t0, unit);
                memcpy(rd(l, base + (uint32_t)in.offset1 * unit * mul, unit), t1, unit);
            }
            return;
        }
        case OP_ds_bpermute_b32: {       // dst, addr, data: lane l reads data of lane (addr/4) mod 64; a disabled source lane gives 0
            uint32_t snap[64];
            for (int l = 0; l < 64; l++) snap[l] = ((exec >> l) & 1) ? w.v[in.o[2].reg][l] : 0;
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint32_t a = w.v[in.o[1].reg][l] + (uint32_t)in.offset;
                w.v[in.o[0].reg][l] = snap[(a >> 2) & 63];
            }
            return;
        }
        case OP_ds_permute_b32: {        // forward: lane l sends its data to lane (addr/4) mod 64; highest writer wins
            uint32_t out[64] = {0};
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint32_t a = w.v[in.o[1].reg][l] + (uint32_t)in.offset;
                out[(a >> 2) & 63] = w.v[in.o[2].reg][l];
            }
            for (int l = 0; l < 64; l++) if ((exec >> l) & 1) w.v[in.o[0].reg][l] = out[l];
            return;
        }
        case OP_ds_add_u32: case OP_ds_max_i32: case OP_ds_max_u32: case OP_ds_or_b32:
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint8_t* p = rd(l, w.v[in.o[0].reg][l] + (uint32_t)in.offset, 4);
                uint32_t old, d = w.v[in.o[1].reg][l]; memcpy(&old, p, 4);
                uint32_t nv = in.op == OP_ds_add_u32 ? old + d : in.op == OP_ds_max_i32 ? (uint32_t)std::max((int32_t)old, (int32_t)d) : in.op == OP_ds_max_u32 ? std::max(old, d) : (old | d);
                memcpy(p, &nv, 4);
            }
            return;
        case OP_ds_add_rtn_u32: case OP_ds_max_rtn_i32:
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint8_t* p = rd(l, w.v[in.o[1].reg][l] + (uint32_t)in.offset, 4);
                uint32_t old, d = w.v[in.o[2].reg][l]; memcpy(&old, p, 4);
                uint32_t nv = in.op == OP_ds_add_rtn_u32 ? old + d : (uint32_t)std::max((int32_t)old, (int32_t)d);
                memcpy(p, &nv, 4);
                w.v[in.o[0].reg][l] = old;
            }
            return;
        default: throw Fault(std::string("LDS form not modelled: ") + op_name[in.op]);
    }
}

static void exec_smem(Wave& w, const Inst& in) {
    int nd = in.op == OP_s_load_dword ? 1 : in.op == OP_s_load_dwordx2 ? 2 : in.op == OP_s_load_dwordx4 ? 4 : in.op == OP_s_load_dwordx8 ? 8 : 16;
    uint64_t a = rs64(w, in.o[1]) + (uint32_t)rs32(w, in.o[2]) + (uint64_t)(int64_t)in.offset;
    a &= ~3ull;
    if (!mem_ok(a, 4u * nd)) {
        char buf[384];
        snprintf(buf, sizeof buf, "scalar load out of bounds: %d bytes at 0x%llx: %s", 4 * nd, (unsigned long long)a, mem_describe(a).c_str());
        throw Fault(buf);
    }
    if (const uint8_t* sh = mem_shadow(a)) {
        for (int i = 0; i < 4 * nd; i++) if (!sh[i]) {
            char buf[384];
            snprintf(buf, sizeof buf, "scalar load of device memory nothing has written: byte %d of %d at 0x%llx: %s", i, 4 * nd, (unsigned long long)a, mem_describe(a).c_str());
            throw Fault(buf);
        }
    }
    uint32_t tmp[16];
    memcpy(tmp, (const void*)(uintptr_t)a, 4u * nd);
    for (int i = 0; i < nd; i++) w.s[in.o[0].reg + i] = tmp[i];
}

static inline uint32_t pk_src(uint32_t x, int idx, const Inst& in) {        // neg_lo / neg_hi of operand idx
    if (in.neg_lo & (1 << idx)) x ^= 0x8000u;
    if (in.neg_hi & (1 << idx)) x ^= 0x80000000u;
    return x;
}

static void exec_vop3p(Wave& w, const Inst& in) {
    const uint64_t exec = EXEC(w);
    Src A, B, C;
    mk_src(w, in.o[1], A);
    mk_src(w, in.o[2], B);
    bool three = in.nops > 3;
    if (three) mk_src(w, in.o[3], C);
    else { C.c[0] = 0; C.lo = C.hi = C.c; C.st = 0; }
    // a float literal in a packed-f16 operand is the half in the low word (gfx9: the high half reads 0)
    auto fix = [&](const Opnd& o, Src& S) { if (o.kind == K_IMM && o.isfloat) S.c[0] = d2h((double)bits2f((uint32_t)o.imm)); };
    fix(in.o[1], A); fix(in.o[2], B); if (three) fix(in.o[3], C);
    uint32_t* dst = w.v[in.o[0].reg];
    if (g_fast) {
        auto xm = [&](int i) { return ((in.neg_lo >> i) & 1 ? 0x8000u : 0u) | ((in.neg_hi >> i) & 1 ? 0x80000000u : 0u); };
        switch (in.op) {
            case OP_v_pk_add_f16: simfast::pk_add_f16(dst, A.lo, A.st, xm(0), B.lo, B.st, xm(1), exec); return;
            case OP_v_pk_mul_f16: simfast::pk_mul_f16(dst, A.lo, A.st, xm(0), B.lo, B.st, xm(1), exec); return;
            case OP_v_pk_fma_f16: simfast::pk_fma_f16(dst, A.lo, A.st, xm(0), B.lo, B.st, xm(1), C.lo, C.st, xm(2), exec); return;
            case OP_v_pk_maximum3_f16: if (simfast::pk_maximum3_f16(dst, A.lo, A.st, xm(0), B.lo, B.st, xm(1), C.lo, C.st, xm(2), exec)) return; break;
            default: break;
        }
    }
    for (int l = 0; l < 64; l++) {
        if (!((exec >> l) & 1)) continue;
        uint32_t a = pk_src(L32(A, l), 0, in), b = pk_src(L32(B, l), 1, in), c = three ? pk_src(L32(C, l), 2, in) : 0;
        uint16_t al = (uint16_t)a, ah = (uint16_t)(a >> 16), bl = (uint16_t)b, bh = (uint16_t)(b >> 16), cl = (uint16_t)c, ch = (uint16_t)(c >> 16), rl, rh;
        switch (in.op) {
            case OP_v_pk_add_f16: rl = h_add(al, bl); rh = h_add(ah, bh); break;
            case OP_v_pk_mul_f16: rl = h_mul(al, bl); rh = h_mul(ah, bh); break;
            case OP_v_pk_fma_f16: rl = h_fma(al, bl, cl); rh = h_fma(ah, bh, ch); break;
            case OP_v_pk_maximum3_f16: rl = h_maximum(h_maximum(al, bl), cl); rh = h_maximum(h_maximum(ah, bh), ch); break;
            case OP_v_pk_minimum3_f16: rl = h_minimum(h_minimum(al, bl), cl); rh = h_minimum(h_minimum(ah, bh), ch); break;
            case OP_v_pk_max_f16: rl = h_max_legacy(al, bl); rh = h_max_legacy(ah, bh); break;
            case OP_v_pk_min_f16: rl = h_min_legacy(al, bl); rh = h_min_legacy(ah, bh); break;
            case OP_v_pk_add_u16: case OP_v_pk_add_i16: rl = al + bl; rh = ah + bh; break;
            case OP_v_pk_sub_u16: case OP_v_pk_sub_i16: rl = al - bl; rh = ah - bh; break;
            case OP_v_pk_max_i16: rl = (uint16_t)std::max((int16_t)al, (int16_t)bl); rh = (uint16_t)std::max((int16_t)ah, (int16_t)bh); break;
            case OP_v_pk_min_i16: rl = (uint16_t)std::min((int16_t)al, (int16_t)bl); rh = (uint16_t)std::min((int16_t)ah, (int16_t)bh); break;
            case OP_v_pk_max_u16: rl = std::max(al, bl); rh = std::max(ah, bh); break;
            case OP_v_pk_min_u16: rl = std::min(al, bl); rh = std::min(ah, bh); break;
            case OP_v_pk_lshlrev_b16: rl = (uint16_t)(bl << (al & 15)); rh = (uint16_t)(bh << (ah & 15)); break;
            case OP_v_pk_lshrrev_b16: rl = (uint16_t)(bl >> (al & 15)); rh = (uint16_t)(bh >> (ah & 15)); break;
            case OP_v_pk_mul_lo_u16: rl = (uint16_t)(al * bl); rh = (uint16_t)(ah * bh); break;
            default: throw Fault(std::string("packed op not modelled: ") + op_name[in.op]);
        }
        dst[l] = (uint32_t)rl | ((uint32_t)rh << 16);
    }
}

static void exec_cmp(Wave& w, const Inst& in) {
    const uint64_t exec = EXEC(w);
    Src A, B;
    mk_src(w, in.o[1], A);
    mk_src(w, in.o[2], B);
    uint64_t r = 0;
    int ty = in.bitop3, rel = in.simm;
    for (int l = 0; l < 64; l++) {
        if (!((exec >> l) & 1)) continue;
        int cmp;      // -1, 0, 1 ; 2 = unordered
        switch (ty) {
            case 0: { int32_t a = (int32_t)L32(A, l), b = (int32_t)L32(B, l); cmp = a < b ? -1 : a > b; break; }
            case 1: { uint32_t a = L32(A, l), b = L32(B, l); cmp = a < b ? -1 : a > b; break; }
            case 2: { int64_t a = (int64_t)L64(A, l), b = (int64_t)L64(B, l); cmp = a < b ? -1 : a > b; break; }
            case 3: { uint64_t a = L64(A, l), b = L64(B, l); cmp = a < b ? -1 : a > b; break; }
            case 4: { uint16_t a = (uint16_t)L32(A, l), b = (uint16_t)L32(B, l); cmp = a < b ? -1 : a > b; break; }
            case 5: { int16_t a = (int16_t)L32(A, l), b = (int16_t)L32(B, l); cmp = a < b ? -1 : a > b; break; }
            case 6: { float a = bits2f(fmod32(L32(A, l), in.o[1])), b = bits2f(fmod32(L32(B, l), in.o[2])); cmp = (a != a || b != b) ? 2 : a < b ? -1 : a > b; break; }
            default: { float a = g_h2f[L32(A, l) & 0xffff], b = g_h2f[L32(B, l) & 0xffff]; cmp = (a != a || b != b) ? 2 : a < b ? -1 : a > b; break; }
        }
        bool t;
        switch (rel) {
            case 0: t = false; break;
            case 1: t = cmp == -1; break;
            case 2: t = cmp == 0; break;
            case 3: t = cmp == -1 || cmp == 0; break;
            case 4: t = cmp == 1; break;
            case 5: t = ty >= 6 ? (cmp == -1 || cmp == 1) : cmp != 0; break;        // float "lg"; integer "ne"
            case 6: t = cmp == 1 || cmp == 0; break;
            default: t = true; break;
        }
        if (t) r |= 1ull << l;
    }
    ws64(w, in.o[0], r);
}

// the generic one-dword VALU path with DPP / SDWA / float modifiers
static void exec_alu32(Wave& w, const Inst& in) {
    const uint64_t exec = EXEC(w);
    const int nsrc = in.nops - 1;
    Src S[3];
    for (int i = 0; i < 3; i++) {
        if (i < nsrc) mk_src(w, in.o[1 + i], S[i]);
        else { S[i].c[0] = S[i].c[1] = 0; S[i].lo = S[i].hi = S[i].c; S[i].st = 0; }
    }
    uint32_t* dst = w.v[in.o[0].reg];
    const bool fop = is_f32_op(in.op) || in.op == OP_v_cvt_f32_f16;
    if (in.enc == 0 && !fop) {
        if (g_fast) {
            if (in.op == OP_v_mov_b32) { simfast::mov_b32(dst, S[0].lo, S[0].st, exec); return; }
            if (in.op == OP_v_perm_b32 && simfast::perm_b32(dst, S[0].lo, S[0].st, S[1].lo, S[1].st, S[2].lo, S[2].st, exec)) return;
        }
        // scalars are spread over 64 entries so that every loop below is unit-stride (the compiler vectorises them)
        alignas(64) uint32_t bc[3][64], tmp[64];
        const uint32_t* P[3];
        for (int i = 0; i < 3; i++) {
            if (S[i].st) P[i] = S[i].lo;
            else { for (int l = 0; l < 64; l++) bc[i][l] = S[i].c[0]; P[i] = bc[i]; }
        }
        const uint32_t *A = P[0], *B = P[1], *C = P[2];
        bool done = true;
#define LOOP(EXPR) for (int l = 0; l < 64; l++) { const uint32_t a = A[l], b = B[l], c = C[l]; (void)a; (void)b; (void)c; tmp[l] = (EXPR); } break;
        switch (in.op) {
            case OP_v_mov_b32: LOOP(a)
            case OP_v_not_b32: LOOP(~a)
            case OP_v_add_u32: LOOP(a + b)
            case OP_v_sub_u32: LOOP(a - b)
            case OP_v_subrev_u32: LOOP(b - a)
            case OP_v_and_b32: LOOP(a & b)
            case OP_v_or_b32: LOOP(a | b)
            case OP_v_xor_b32: LOOP(a ^ b)
            case OP_v_lshlrev_b32: LOOP(b << (a & 31))
            case OP_v_lshrrev_b32: LOOP(b >> (a & 31))
            case OP_v_ashrrev_i32: LOOP((uint32_t)((int32_t)b >> (a & 31)))
            case OP_v_max_i32: LOOP((uint32_t)std::max((int32_t)a, (int32_t)b))
            case OP_v_min_i32: LOOP((uint32_t)std::min((int32_t)a, (int32_t)b))
            case OP_v_max_u32: LOOP(std::max(a, b))
            case OP_v_min_u32: LOOP(std::min(a, b))
            case OP_v_mul_u32_u24: LOOP((a & 0xffffff) * (b & 0xffffff))
            case OP_v_mul_lo_u32: LOOP(a * b)
            case OP_v_mad_u32_u24: LOOP((a & 0xffffff) * (b & 0xffffff) + c)
            case OP_v_lshl_add_u32: LOOP((a << (b & 31)) + c)
            case OP_v_add_lshl_u32: LOOP((a + b) << (c & 31))
            case OP_v_lshl_or_b32: LOOP((a << (b & 31)) | c)
            case OP_v_and_or_b32: LOOP((a & b) | c)
            case OP_v_or3_b32: LOOP(a | b | c)
            case OP_v_add3_u32: LOOP(a + b + c)
            case OP_v_max3_i32: LOOP((uint32_t)std::max(std::max((int32_t)a, (int32_t)b), (int32_t)c))
            case OP_v_bfe_u32: LOOP((c & 31) ? (a >> (b & 31)) & ((1u << (c & 31)) - 1) : 0)
            case OP_v_bcnt_u32_b32: LOOP((uint32_t)__builtin_popcount(a) + b)
            default: done = false; break;
        }
#undef LOOP
        if (done) {
            if (exec == ~0ull) memcpy(dst, tmp, sizeof tmp);
            else for (int l = 0; l < 64; l++) if ((exec >> l) & 1) dst[l] = tmp[l];
            return;
        }
    }
    uint32_t snap[64];
    if (in.enc == 1) memcpy(snap, w.v[in.o[1].reg], sizeof snap);
    for (int l = 0; l < 64; l++) {
        if (!((exec >> l) & 1)) continue;
        uint32_t a = L32(S[0], l), b = L32(S[1], l), c = L32(S[2], l);
        if (in.enc == 1) {
            if (!((in.row_mask >> (l >> 4)) & 1) || !((in.bank_mask >> ((l >> 2) & 3)) & 1)) continue;
            int sl;
            if (dpp_source(in, l, exec, sl)) a = snap[sl];
            else if (in.bound_ctrl) a = 0;
            else continue;
        }
        if (in.enc == 2) {
            a = sdwa_sel(a, in.src0_sel, in.o[1].sext);
            if (nsrc > 1) b = sdwa_sel(b, in.src1_sel, in.o[2].sext);
        }
        if (fop) {
            if (in.op == OP_v_cvt_f32_f16) { if (in.o[1].abs) a &= 0x7fff; if (in.o[1].neg) a ^= 0x8000; }
            else { a = fmod32(a, in.o[1]); if (nsrc > 1) b = fmod32(b, in.o[2]); }
        }
        uint32_t r = alu32(in, a, b, c, l);
        if (in.enc == 2) r = sdwa_dst(dst[l], r, in.dst_sel, in.dst_unused);
        dst[l] = r;
    }
}

static void fault_unknown(const Inst& in) { throw Fault(std::string("no semantics for ") + op_name[in.op]); }

// returns false when the wave stops (barrier or end)
static bool step(Ctx& c, Wave& w) {
    const Inst& in = c.k->code[w.pc];
    c.count[in.cls]++;
    int next = w.pc + 1;
    switch (in.op) {
        case OP_s_nop: case OP_s_waitcnt: case OP_buffer_wbl2: case OP_buffer_inv: case OP_s_sleep: case OP_s_setprio: break;
        case OP_s_endpgm: w.state = 2; return false;
        case OP_s_barrier: w.state = 1; w.pc = next; return false;
        case OP_s_trap: throw Fault("s_trap " + std::to_string((long long)in.o[0].imm) + " (abort / failed assert in device code)");
        case OP_s_getpc_b64: ws64(w, in.o[0], in.addr + 4); break;
        case OP_s_swappc_b64: case OP_s_setpc_b64: {
            const uint64_t tgt = rs64(w, in.o[in.op == OP_s_swappc_b64 ? 1 : 0]);
            if (in.op == OP_s_swappc_b64) ws64(w, in.o[0], in.addr + 4);
            std::string e;
            next = parse_function_at(*c.k, tgt, e);
            if (next < 0) throw Fault("call / return to 0x" + std::to_string(tgt) + ": " + e);
            break;
        }
        case OP_s_branch: next = in.target; break;
        case OP_s_cbranch_scc0: if (!w.scc) next = in.target; break;
        case OP_s_cbranch_scc1: if (w.scc) next = in.target; break;
        case OP_s_cbranch_vccz: if (VCC(w) == 0) next = in.target; break;
        case OP_s_cbranch_vccnz: if (VCC(w) != 0) next = in.target; break;
        case OP_s_cbranch_execz: if (EXEC(w) == 0) next = in.target; break;
        case OP_s_cbranch_execnz: if (EXEC(w) != 0) next = in.target; break;

        case OP_s_mov_b32: ws32(w, in.o[0], rs32(w, in.o[1])); break;
        case OP_s_mov_b64: ws64(w, in.o[0], rs64(w, in.o[1])); break;
        case OP_s_movk_i32: ws32(w, in.o[0], (uint32_t)in.o[1].imm); break;
        case OP_s_addk_i32: { int64_t r = (int64_t)(int32_t)w.s[in.o[0].reg] + (int32_t)in.o[1].imm; w.scc = r != (int32_t)r; ws32(w, in.o[0], (uint32_t)r); break; }
        case OP_s_mulk_i32: ws32(w, in.o[0], w.s[in.o[0].reg] * (uint32_t)in.o[1].imm); break;
        case OP_s_not_b32: { uint32_t r = ~rs32(w, in.o[1]); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_not_b64: { uint64_t r = ~rs64(w, in.o[1]); ws64(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_bcnt1_i32_b64: { uint32_t r = (uint32_t)__builtin_popcountll(rs64(w, in.o[1])); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_bcnt1_i32_b32: { uint32_t r = (uint32_t)__builtin_popcount(rs32(w, in.o[1])); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_ff1_i32_b64: { uint64_t x = rs64(w, in.o[1]); ws32(w, in.o[0], x ? (uint32_t)__builtin_ctzll(x) : 0xffffffffu); break; }
        case OP_s_ff1_i32_b32: { uint32_t x = rs32(w, in.o[1]); ws32(w, in.o[0], x ? (uint32_t)__builtin_ctz(x) : 0xffffffffu); break; }
        case OP_s_brev_b32: { uint32_t a = rs32(w, in.o[1]), r = 0; for (int i = 0; i < 32; i++) if (a & (1u << i)) r |= 1u << (31 - i); ws32(w, in.o[0], r); break; }
        case OP_s_brev_b64: { uint64_t a = rs64(w, in.o[1]), r = 0; for (int i = 0; i < 64; i++) if (a & (1ull << i)) r |= 1ull << (63 - i); ws64(w, in.o[0], r); break; }
        case OP_s_flbit_i32_b32: { uint32_t a = rs32(w, in.o[1]); ws32(w, in.o[0], a ? (uint32_t)__builtin_clz(a) : 0xffffffffu); break; }
        case OP_s_flbit_i32_b64: { uint64_t a = rs64(w, in.o[1]); ws32(w, in.o[0], a ? (uint32_t)__builtin_clzll(a) : 0xffffffffu); break; }
        case OP_s_ff0_i32_b32: { uint32_t a = ~rs32(w, in.o[1]); ws32(w, in.o[0], a ? (uint32_t)__builtin_ctz(a) : 0xffffffffu); break; }
        case OP_s_bitset1_b32: ws32(w, in.o[0], w.s[in.o[0].reg] | (1u << (rs32(w, in.o[1]) & 31))); break;
        case OP_s_bitset0_b32: ws32(w, in.o[0], w.s[in.o[0].reg] & ~(1u << (rs32(w, in.o[1]) & 31))); break;
        case OP_s_sext_i32_i16: ws32(w, in.o[0], (uint32_t)(int32_t)(int16_t)rs32(w, in.o[1])); break;
        case OP_s_sext_i32_i8: ws32(w, in.o[0], (uint32_t)(int32_t)(int8_t)rs32(w, in.o[1])); break;
        case OP_s_abs_i32: { int32_t x = (int32_t)rs32(w, in.o[1]); uint32_t r = x < 0 ? (uint32_t)-(int64_t)x : (uint32_t)x; ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_and_saveexec_b64: { uint64_t e = EXEC(w), s = rs64(w, in.o[1]); ws64(w, in.o[0], e); set64(&w.s[SG_EXEC], s & e); w.scc = (s & e) != 0; break; }
        case OP_s_or_saveexec_b64: { uint64_t e = EXEC(w), s = rs64(w, in.o[1]); ws64(w, in.o[0], e); set64(&w.s[SG_EXEC], s | e); w.scc = (s | e) != 0; break; }
        case OP_s_andn2_saveexec_b64: { uint64_t e = EXEC(w), s = rs64(w, in.o[1]); ws64(w, in.o[0], e); set64(&w.s[SG_EXEC], s & ~e); w.scc = (s & ~e) != 0; break; }
        case OP_s_add_i32: { int64_t r = (int64_t)(int32_t)rs32(w, in.o[1]) + (int32_t)rs32(w, in.o[2]); ws32(w, in.o[0], (uint32_t)r); w.scc = r != (int32_t)r; break; }
        case OP_s_sub_i32: { int64_t r = (int64_t)(int32_t)rs32(w, in.o[1]) - (int32_t)rs32(w, in.o[2]); ws32(w, in.o[0], (uint32_t)r); w.scc = r != (int32_t)r; break; }
        case OP_s_add_u32: { uint64_t r = (uint64_t)rs32(w, in.o[1]) + rs32(w, in.o[2]); ws32(w, in.o[0], (uint32_t)r); w.scc = (r >> 32) != 0; break; }
        case OP_s_addc_u32: { uint64_t r = (uint64_t)rs32(w, in.o[1]) + rs32(w, in.o[2]) + (w.scc ? 1 : 0); ws32(w, in.o[0], (uint32_t)r); w.scc = (r >> 32) != 0; break; }
        case OP_s_sub_u32: { uint32_t a = rs32(w, in.o[1]), b = rs32(w, in.o[2]); ws32(w, in.o[0], a - b); w.scc = b > a; break; }
        case OP_s_subb_u32: { uint64_t a = rs32(w, in.o[1]), b = (uint64_t)rs32(w, in.o[2]) + (w.scc ? 1 : 0); ws32(w, in.o[0], (uint32_t)(a - b)); w.scc = b > a; break; }
        case OP_s_mul_i32: ws32(w, in.o[0], rs32(w, in.o[1]) * rs32(w, in.o[2])); break;
        case OP_s_mul_hi_u32: ws32(w, in.o[0], (uint32_t)(((uint64_t)rs32(w, in.o[1]) * rs32(w, in.o[2])) >> 32)); break;
        case OP_s_min_i32: { int32_t a = (int32_t)rs32(w, in.o[1]), b = (int32_t)rs32(w, in.o[2]); w.scc = a < b; ws32(w, in.o[0], (uint32_t)(a < b ? a : b)); break; }
        case OP_s_max_i32: { int32_t a = (int32_t)rs32(w, in.o[1]), b = (int32_t)rs32(w, in.o[2]); w.scc = a > b; ws32(w, in.o[0], (uint32_t)(a > b ? a : b)); break; }
        case OP_s_min_u32: { uint32_t a = rs32(w, in.o[1]), b = rs32(w, in.o[2]); w.scc = a < b; ws32(w, in.o[0], a < b ? a : b); break; }
        case OP_s_max_u32: { uint32_t a = rs32(w, in.o[1]), b = rs32(w, in.o[2]); w.scc = a > b; ws32(w, in.o[0], a > b ? a : b); break; }
        case OP_s_lshl_b32: { uint32_t r = rs32(w, in.o[1]) << (rs32(w, in.o[2]) & 31); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_lshr_b32: { uint32_t r = rs32(w, in.o[1]) >> (rs32(w, in.o[2]) & 31); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_ashr_i32: { uint32_t r = (uint32_t)((int32_t)rs32(w, in.o[1]) >> (rs32(w, in.o[2]) & 31)); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_lshl_b64: { uint64_t r = rs64(w, in.o[1]) << (rs32(w, in.o[2]) & 63); ws64(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_lshr_b64: { uint64_t r = rs64(w, in.o[1]) >> (rs32(w, in.o[2]) & 63); ws64(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_lshl1_add_u32: case OP_s_lshl2_add_u32: case OP_s_lshl3_add_u32: case OP_s_lshl4_add_u32: {
            int sh = in.op == OP_s_lshl1_add_u32 ? 1 : in.op == OP_s_lshl2_add_u32 ? 2 : in.op == OP_s_lshl3_add_u32 ? 3 : 4;
            uint64_t r = ((uint64_t)rs32(w, in.o[1]) << sh) + rs32(w, in.o[2]); ws32(w, in.o[0], (uint32_t)r); w.scc = (r >> 32) != 0; break;
        }
        case OP_s_and_b32: { uint32_t r = rs32(w, in.o[1]) & rs32(w, in.o[2]); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_or_b32: { uint32_t r = rs32(w, in.o[1]) | rs32(w, in.o[2]); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_xor_b32: { uint32_t r = rs32(w, in.o[1]) ^ rs32(w, in.o[2]); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_andn2_b32: { uint32_t r = rs32(w, in.o[1]) & ~rs32(w, in.o[2]); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_and_b64: { uint64_t r = rs64(w, in.o[1]) & rs64(w, in.o[2]); ws64(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_or_b64: { uint64_t r = rs64(w, in.o[1]) | rs64(w, in.o[2]); ws64(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_xor_b64: { uint64_t r = rs64(w, in.o[1]) ^ rs64(w, in.o[2]); ws64(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_andn2_b64: { uint64_t r = rs64(w, in.o[1]) & ~rs64(w, in.o[2]); ws64(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_orn2_b64: { uint64_t r = rs64(w, in.o[1]) | ~rs64(w, in.o[2]); ws64(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_cselect_b32: ws32(w, in.o[0], w.scc ? rs32(w, in.o[1]) : rs32(w, in.o[2])); break;
        case OP_s_cselect_b64: ws64(w, in.o[0], w.scc ? rs64(w, in.o[1]) : rs64(w, in.o[2])); break;
        case OP_s_bfe_u32: { uint32_t a = rs32(w, in.o[1]), b = rs32(w, in.o[2]), off = b & 31, wd = (b >> 16) & 0x7f; uint32_t r = wd == 0 ? 0 : wd >= 32 ? a >> off : (a >> off) & ((1u << wd) - 1); ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_bfe_i32: { uint32_t a = rs32(w, in.o[1]), b = rs32(w, in.o[2]), off = b & 31, wd = (b >> 16) & 0x7f; uint32_t r = 0; if (wd) { if (wd > 32 - off) wd = 32 - off; r = (a >> off) & (wd >= 32 ? 0xffffffffu : ((1u << wd) - 1)); if (wd < 32 && (r >> (wd - 1)) & 1) r |= ~((1u << wd) - 1); } ws32(w, in.o[0], r); w.scc = r != 0; break; }
        case OP_s_cmp_eq_u32: case OP_s_cmp_eq_i32: w.scc = rs32(w, in.o[0]) == rs32(w, in.o[1]); break;
        case OP_s_cmp_lg_u32: case OP_s_cmp_lg_i32: w.scc = rs32(w, in.o[0]) != rs32(w, in.o[1]); break;
        case OP_s_cmp_eq_u64: w.scc = rs64(w, in.o[0]) == rs64(w, in.o[1]); break;
        case OP_s_cmp_lg_u64: w.scc = rs64(w, in.o[0]) != rs64(w, in.o[1]); break;
        case OP_s_cmp_ge_i32: case OP_s_cmpk_ge_i32: w.scc = (int32_t)rs32(w, in.o[0]) >= (int32_t)rs32(w, in.o[1]); break;
        case OP_s_cmp_lt_i32: case OP_s_cmpk_lt_i32: w.scc = (int32_t)rs32(w, in.o[0]) < (int32_t)rs32(w, in.o[1]); break;
        case OP_s_cmp_le_i32: case OP_s_cmpk_le_i32: w.scc = (int32_t)rs32(w, in.o[0]) <= (int32_t)rs32(w, in.o[1]); break;
        case OP_s_cmp_gt_i32: case OP_s_cmpk_gt_i32: w.scc = (int32_t)rs32(w, in.o[0]) > (int32_t)rs32(w, in.o[1]); break;
        case OP_s_cmpk_eq_i32: case OP_s_cmpk_eq_u32: w.scc = rs32(w, in.o[0]) == rs32(w, in.o[1]); break;
        case OP_s_cmpk_lg_i32: case OP_s_cmpk_lg_u32: w.scc = rs32(w, in.o[0]) != rs32(w, in.o[1]); break;
        case OP_s_cmp_lt_u32: case OP_s_cmpk_lt_u32: w.scc = rs32(w, in.o[0]) < rs32(w, in.o[1]); break;
        case OP_s_cmp_ge_u32: case OP_s_cmpk_ge_u32: w.scc = rs32(w, in.o[0]) >= rs32(w, in.o[1]); break;
        case OP_s_cmp_le_u32: case OP_s_cmpk_le_u32: w.scc = rs32(w, in.o[0]) <= rs32(w, in.o[1]); break;
        case OP_s_cmp_gt_u32: case OP_s_cmpk_gt_u32: w.scc = rs32(w, in.o[0]) > rs32(w, in.o[1]); break;
        case OP_s_load_dword: case OP_s_load_dwordx2: case OP_s_load_dwordx4: case OP_s_load_dwordx8: case OP_s_load_dwordx16: exec_smem(w, in); break;

        case OP_v_cmp: exec_cmp(w, in); break;
        case OP_v_cndmask_b32: {
            const uint64_t exec = EXEC(w), m = rs64(w, in.o[3]);
            Src A, B; mk_src(w, in.o[1], A); mk_src(w, in.o[2], B);
            uint32_t* dst = w.v[in.o[0].reg];
            if (in.enc == 2) throw Fault("v_cndmask with sdwa not modelled");
            if (in.enc == 1) {        // VOP2: the first source comes through DPP (a lane without a valid source keeps its destination unless bound_ctrl)
                if (in.o[1].kind != K_VGPR) throw Fault("v_cndmask_b32_dpp: src0 must be a VGPR");
                uint32_t snap[64], res[64];
                memcpy(snap, w.v[in.o[1].reg], sizeof snap);
                for (int l = 0; l < 64; l++) res[l] = L32(B, l);
                for (int l = 0; l < 64; l++) {
                    if (!((exec >> l) & 1)) continue;
                    if (!((in.row_mask >> (l >> 4)) & 1) || !((in.bank_mask >> ((l >> 2) & 3)) & 1)) continue;
                    uint32_t a;
                    int sl;
                    if (dpp_source(in, l, exec, sl)) a = snap[sl];
                    else if (in.bound_ctrl) a = 0;
                    else continue;
                    dst[l] = ((m >> l) & 1) ? res[l] : a;
                }
                break;
            }
            for (int l = 0; l < 64; l++) if ((exec >> l) & 1) dst[l] = ((m >> l) & 1) ? fmod32(L32(B, l), in.o[2]) : fmod32(L32(A, l), in.o[1]);
            break;
        }
        case OP_v_mov_b64: {
            const uint64_t exec = EXEC(w); Src A; mk_src(w, in.o[1], A);
            for (int l = 0; l < 64; l++) if ((exec >> l) & 1) { uint64_t x = L64(A, l); w.v[in.o[0].reg][l] = (uint32_t)x; w.v[in.o[0].reg + 1][l] = (uint32_t)(x >> 32); }
            break;
        }
        case OP_v_lshl_add_u64: case OP_v_lshlrev_b64: case OP_v_lshrrev_b64: case OP_v_ashrrev_i64: {
            const uint64_t exec = EXEC(w); Src A, B, C; mk_src(w, in.o[1], A); mk_src(w, in.o[2], B);
            if (in.nops > 3) mk_src(w, in.o[3], C);
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint64_t r;
                if (in.op == OP_v_lshl_add_u64) r = (L64(A, l) << (L32(B, l) & 7)) + L64(C, l);
                else if (in.op == OP_v_lshlrev_b64) r = L64(B, l) << (L32(A, l) & 63);
                else if (in.op == OP_v_lshrrev_b64) r = L64(B, l) >> (L32(A, l) & 63);
                else r = (uint64_t)((int64_t)L64(B, l) >> (L32(A, l) & 63));
                w.v[in.o[0].reg][l] = (uint32_t)r; w.v[in.o[0].reg + 1][l] = (uint32_t)(r >> 32);
            }
            break;
        }
        case OP_v_mad_u64_u32: case OP_v_mad_i64_i32: {       // dst, carry-out, s0, s1, s2(64)
            const uint64_t exec = EXEC(w); Src A, B, C; mk_src(w, in.o[2], A); mk_src(w, in.o[3], B); mk_src(w, in.o[4], C);
            uint64_t carry = 0;
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint64_t r;
                if (in.op == OP_v_mad_u64_u32) { unsigned __int128 t = (unsigned __int128)L32(A, l) * L32(B, l) + L64(C, l); r = (uint64_t)t; if (t >> 64) carry |= 1ull << l; }
                else r = (uint64_t)((int64_t)(int32_t)L32(A, l) * (int32_t)L32(B, l) + (int64_t)L64(C, l));
                w.v[in.o[0].reg][l] = (uint32_t)r; w.v[in.o[0].reg + 1][l] = (uint32_t)(r >> 32);
            }
            if (in.o[1].kind == K_SGPR) ws64(w, in.o[1], carry);
            break;
        }
        case OP_v_add_co_u32: case OP_v_addc_co_u32: case OP_v_sub_co_u32: case OP_v_subb_co_u32: case OP_v_subrev_co_u32: case OP_v_subbrev_co_u32: {
            const uint64_t exec = EXEC(w); Src A, B; mk_src(w, in.o[2], A); mk_src(w, in.o[3], B);
            bool has_cin = in.op == OP_v_addc_co_u32 || in.op == OP_v_subb_co_u32 || in.op == OP_v_subbrev_co_u32;
            uint64_t cin = has_cin ? rs64(w, in.o[4]) : 0, cout = 0;
            for (int l = 0; l < 64; l++) {
                if (!((exec >> l) & 1)) continue;
                uint64_t a = L32(A, l), b = L32(B, l), ci = (cin >> l) & 1, r;
                if (in.op == OP_v_subrev_co_u32 || in.op == OP_v_subbrev_co_u32) std::swap(a, b);
                if (in.op == OP_v_add_co_u32 || in.op == OP_v_addc_co_u32) { r = a + b + ci; if (r >> 32) cout |= 1ull << l; }
                else { r = a - b - ci; if (b + ci > a) cout |= 1ull << l; }
                w.v[in.o[0].reg][l] = (uint32_t)r;
            }
            if (in.o[1].kind == K_SGPR) ws64(w, in.o[1], cout);
            break;
        }
        case OP_v_readfirstlane_b32: { uint64_t e = EXEC(w); int l = e ? __builtin_ctzll(e) : 0; ws32(w, in.o[0], in.o[1].kind == K_VGPR ? w.v[in.o[1].reg][l] : rs32(w, in.o[1])); break; }
        case OP_v_readlane_b32: ws32(w, in.o[0], w.v[in.o[1].reg][rs32(w, in.o[2]) & 63]); break;
        case OP_v_writelane_b32: w.v[in.o[0].reg][rs32(w, in.o[2]) & 63] = rs32(w, in.o[1]); break;
        case OP_v_pk_maximum3_f16: case OP_v_pk_minimum3_f16: case OP_v_pk_add_f16: case OP_v_pk_fma_f16: case OP_v_pk_mul_f16: case OP_v_pk_max_f16: case OP_v_pk_min_f16:
        case OP_v_pk_add_u16: case OP_v_pk_sub_u16: case OP_v_pk_add_i16: case OP_v_pk_sub_i16: case OP_v_pk_max_i16: case OP_v_pk_min_i16: case OP_v_pk_max_u16: case OP_v_pk_min_u16:
        case OP_v_pk_lshlrev_b16: case OP_v_pk_lshrrev_b16: case OP_v_pk_mul_lo_u16:
            exec_vop3p(w, in); break;
        default:
            if (in.cls == C_VALU) exec_alu32(w, in);
            else if (in.cls == C_LDS) exec_ds(c, w, in);
            else if (in.cls == C_VMEM) exec_vmem(c, w, in);
            else fault_unknown(in);
    }
    w.pc = next;
    return true;
}

static uint32_t g_poison = 0xBAD0BAD1u;
static uint64_t g_trace = 0;           // HIPSIM_TRACE_INSN=n: the first n instructions of wave 0 of the first workgroup, to stderr
static uint64_t g_switch = 0;          // > 0: waves of a workgroup take turns after this many instructions, in a seeded random order
static uint64_t g_max_steps = 0;
static int g_threads = 0;

// one worker: its own registers, LDS and scratch; workgroups come off a shared counter
struct Worker {
    Ctx c;
    std::vector<std::unique_ptr<uint32_t[]>> vpool;
    std::vector<Wave> waves;
    std::vector<uint8_t> scratch;
    std::mt19937_64 rng{12345};
    uint64_t wgs = 0;
};

static void run_workgroup(Worker& W, Kernel& k, Dim3 block, const uint8_t* kernarg, uint32_t dyn_lds, uint32_t gx, uint32_t gy, uint32_t gz) {
    Ctx& c = W.c;
    const uint32_t threads = block.x * block.y * block.z, nw = (threads + 63) / 64;
    const uint32_t user_sgprs = (k.rsrc2 >> 1) & 31;
    auto& waves = W.waves;
    W.wgs++;
    c.lds.resize((size_t)k.lds + dyn_lds);
    if (g_lds_undef) c.lds_def.assign(c.lds.size(), 0);
    for (size_t i = 0; i + 4 <= c.lds.size(); i += 4) memcpy(&c.lds[i], &g_poison, 4);
    const uint32_t nv = std::min<uint32_t>(512, (((k.rsrc1 & 0x3f) + 1) * 8) + 8);        // arch VGPRs of the kernel (+ margin); AGPRs start at 256
    for (uint32_t wi = 0; wi < nw; wi++) {
        Wave& w = waves[wi];
        w.v = (uint32_t(*)[64])W.vpool[wi].get();
        for (uint32_t r = 0; r < nv; r++) for (int l = 0; l < 64; l++) w.v[r][l] = g_poison;
        if (k.rsrc3 || true) for (uint32_t r = 256; r < 256 + 16; r++) for (int l = 0; l < 64; l++) w.v[r][l] = g_poison;
        for (int i = 0; i < 128; i++) w.s[i] = g_poison;
        w.scc = false; w.pc = 0; w.state = 0;
        w.scratch = k.scratch ? W.scratch.data() + (size_t)wi * 64 * k.scratch : nullptr;
        int sg = 0;
        if (k.props & 1) sg += 4;                                           // private segment buffer
        if (k.props & 2) { set64(&w.s[sg], 0); sg += 2; }                   // dispatch ptr (not provided)
        if (k.props & 4) { set64(&w.s[sg], 0); sg += 2; }                   // queue ptr
        if (k.props & 8) { set64(&w.s[sg], (uint64_t)(uintptr_t)kernarg); sg += 2; }
        if (k.props & 16) { set64(&w.s[sg], 0); sg += 2; }                  // dispatch id
        if (k.props & 32) { set64(&w.s[sg], 0); sg += 2; }                  // flat scratch init
        if (k.props & 64) { w.s[sg] = k.scratch; sg += 1; }
        sg = (int)user_sgprs;
        if (k.rsrc2 & (1u << 7)) w.s[sg++] = gx;
        if (k.rsrc2 & (1u << 8)) w.s[sg++] = gy;
        if (k.rsrc2 & (1u << 9)) w.s[sg++] = gz;
        if (k.rsrc2 & (1u << 10)) w.s[sg++] = 0;
        uint64_t exec = 0;
        for (int l = 0; l < 64; l++) {
            uint32_t t = wi * 64 + l;
            if (t >= threads) { w.v[0][l] = 0; continue; }
            exec |= 1ull << l;
            uint32_t tx = t % block.x, ty = (t / block.x) % block.y, tz = t / (block.x * block.y);
            w.v[0][l] = tx | (ty << 10) | (tz << 20);
        }
        set64(&w.s[SG_EXEC], exec);
        set64(&w.s[SG_VCC], 0);
        w.s[SG_M0] = 0;
    }
    // waves take turns at barriers (and, with HIPSIM_SWITCH, in between)
    uint32_t done = 0;
    while (done < nw) {
        bool progressed = false;
        uint32_t order[16];
        for (uint32_t i = 0; i < nw; i++) order[i] = i;
        if (g_switch) for (uint32_t i = nw; i > 1; i--) std::swap(order[i - 1], order[W.rng() % i]);
        for (uint32_t oi = 0; oi < nw; oi++) {
            Wave& w = waves[order[oi]];
            if (w.state != 0) continue;
            c.cur = (int)order[oi];
            uint64_t quantum = g_switch ? 1 + W.rng() % g_switch : UINT64_MAX;
            progressed = true;
            while (quantum--) {
                if (g_trace && order[oi] == 0 && W.wgs == 1) {
                    g_trace--;
                    const Inst& ti = c.k->code[w.pc];
                    fprintf(stderr, "T %u exec=%016llx vcc=%016llx scc=%d", ti.line, (unsigned long long)EXEC(w), (unsigned long long)VCC(w), (int)w.scc);
                    for (int oi2 = 0; oi2 < ti.nops; oi2++) {
                        const Opnd& o = ti.o[oi2];
                        if (o.kind == K_SGPR) fprintf(stderr, " s%u=%x", o.reg, w.s[o.reg]);
                        else if (o.kind == K_VGPR) fprintf(stderr, " v%u=[%x %x %x .. %x]", o.reg, w.v[o.reg][0], w.v[o.reg][1], w.v[o.reg][2], w.v[o.reg][63]);
                    }
                    fprintf(stderr, "\n");
                }
                if (!step(c, w)) break;
                if (g_max_steps && ++c.steps > g_max_steps) throw Fault("HIPSIM_MAX_STEPS exceeded (a wave that never ends?)");
            }
            if (w.state == 2) done++;
        }
        if (!progressed) {           // every live wave waits at the barrier: release
            bool any = false;
            for (uint32_t wi = 0; wi < nw; wi++) if (waves[wi].state == 1) { waves[wi].state = 0; any = true; }
            if (!any) break;
        }
    }
}

static std::string describe_fault(Worker& W, Kernel& k, const char* what) {
    Ctx& c = W.c;
    int pc = c.cur >= 0 ? W.waves[c.cur].pc : -1;
    uint32_t ln = pc >= 0 && pc < (int)k.code.size() ? k.code[pc].line : 0;
    std::string text;
    if (ln) { std::ifstream sf(k.sfile); for (uint32_t i = 0; i < ln && std::getline(sf, text); i++) {} }
    size_t cut = text.find("//");
    if (cut != std::string::npos) text.erase(cut);
    while (!text.empty() && (text.back() == ' ' || text.back() == '\t')) text.pop_back();
    size_t b0 = text.find_first_not_of(" \t");
    if (b0 != std::string::npos) text.erase(0, b0);
    char buf[320];
    snprintf(buf, sizeof buf, " [kernel %.120s, wave %d, %s:%u: ", k.name.c_str(), c.cur, k.sfile.c_str(), ln);
    return std::string(what) + buf + text + "]";
}

std::string run_kernel(Kernel& k, Dim3 grid, Dim3 block, const uint8_t* kernarg, uint32_t dyn_lds) {
    static bool env = false;
    if (!env) {
        env = true;
        if (const char* e = getenv("HIPSIM_POISON")) g_poison = (uint32_t)strtoul(e, nullptr, 0);
        if (const char* e = getenv("HIPSIM_SWITCH")) g_switch = strtoull(e, nullptr, 0);
        if (const char* e = getenv("HIPSIM_TRACE_INSN")) g_trace = strtoull(e, nullptr, 0);
        if (const char* e = getenv("HIPSIM_LDS_UNDEF")) g_lds_undef = atoi(e) != 0;
        if (const char* e = getenv("HIPSIM_MAX_STEPS")) g_max_steps = strtoull(e, nullptr, 0);
        const char* f = getenv("HIPSIM_FAST");
        g_fast = (!f || atoi(f) != 0) && simfast::available();
        const char* t = getenv("HIPSIM_THREADS");
        g_threads = t ? atoi(t) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        if (g_threads < 1) g_threads = 1;
    }
    std::string err;
    if (!parse_kernel(k, err)) return err;
    if (k.preload) return "kernarg preload is not modelled";
    const uint32_t threads = block.x * block.y * block.z, nw = (threads + 63) / 64;
    if (threads == 0 || threads > 1024) return "block of " + std::to_string(threads) + " threads";
    // what the hardware would refuse: more LDS than a workgroup can have (160 KiB on gfx950), or more registers than the waves
    // of one workgroup get on a SIMD (512 VGPRs per lane, the block's waves spread over 4 SIMDs; AGPRs share the file)
    if ((uint64_t)k.lds + dyn_lds > 160u * 1024u)
        return "launch needs " + std::to_string((uint64_t)k.lds + dyn_lds) + " bytes of LDS per workgroup (gfx950: 163840)";
    {
        const uint32_t vg = ((k.rsrc1 & 0x3f) + 1) * 8, waves_per_simd = (nw + 3) / 4;
        if (vg * waves_per_simd > 512)
            return "launch needs " + std::to_string(vg) + " VGPRs x " + std::to_string(waves_per_simd) + " waves of the workgroup per SIMD (512 per lane)";
    }
    const uint64_t total = (uint64_t)grid.x * grid.y * grid.z;
    const int T = (int)std::min<uint64_t>((uint64_t)g_threads, total);
    static std::vector<std::unique_ptr<Worker>> pool;          // (launches are serialised by the runtime's lock)
    while ((int)pool.size() < std::max(T, 1)) pool.emplace_back(new Worker());
    std::atomic<uint64_t> next{0};
    std::atomic<bool> stop{false};
    std::mutex fmu;
    std::string fault;
    k.launches++;
    auto body = [&](int t) {
        Worker& W = *pool[t];
        W.c.k = &k;
        W.c.scratch_bytes = k.scratch;
        W.c.steps = 0;
        memset(W.c.count, 0, sizeof W.c.count);
        W.wgs = 0;
        while (W.vpool.size() < nw) W.vpool.emplace_back(new uint32_t[512 * 64]);
        W.waves.resize(std::max<size_t>(W.waves.size(), nw));
        W.scratch.resize((size_t)nw * 64 * k.scratch);
        try {
            for (;;) {
                if (stop.load(std::memory_order_relaxed)) break;
                uint64_t id = next.fetch_add(1);
                if (id >= total) break;
                uint32_t gx = (uint32_t)(id % grid.x), gy = (uint32_t)((id / grid.x) % grid.y), gz = (uint32_t)(id / ((uint64_t)grid.x * grid.y));
                run_workgroup(W, k, block, kernarg, dyn_lds, gx, gy, gz);
            }
        } catch (const Fault& f) {
            stop.store(true);
            std::lock_guard<std::mutex> g(fmu);
            if (fault.empty()) fault = describe_fault(W, k, f.what());
        }
    };
    if (T <= 1) body(0);
    else {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(body, t);
        body(0);
        for (auto& x : th) x.join();
    }
    for (int t = 0; t < std::max(T, 1); t++) {
        k.wgs += pool[t]->wgs;
        for (int i = 0; i < C_N; i++) k.count[i] += pool[t]->c.count[i];
    }
    return fault;
}

}  // namespace sim
