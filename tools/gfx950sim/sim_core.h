// gfx950sim - a CPU interpreter for the gfx950 (CDNA4) code objects hipcc emits for this repository.
// TEST INFRASTRUCTURE ONLY: nothing under swipe_amd/ includes, links or loads it. It exists so that the compiled kernels
// of libswipe_amd.so can be run against the oracle on a machine without a GPU (tests/test_sim_*.py), with every device
// access bounds-checked and every wave instruction counted. It interprets llvm-objdump's text, one wave64 at a time.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <unordered_map>

namespace sim {

enum OKind : uint8_t { K_NONE = 0, K_SGPR, K_VGPR, K_IMM, K_OFF, K_SCC };

struct Opnd {
    uint8_t kind = K_NONE;
    uint8_t n = 1;            // dwords
    uint16_t reg = 0;         // SGPR file index (vcc 106, m0 124, exec 126) or VGPR index (AGPR a<k> = 256 + k)
    uint64_t imm = 0;         // K_IMM: value already widened (sign-extended integer / float bit pattern in the low dword)
    bool isfloat = false;     // the text was a float literal (1.0): f16 users take the half pattern
    bool neg = false, abs = false, sext = false;
};

struct Inst {
    uint16_t op = 0;
    uint8_t enc = 0;          // 0 plain, 1 dpp, 2 sdwa
    uint8_t nops = 0;
    Opnd o[6];
    int32_t offset = 0, offset1 = 0;
    uint8_t dpp_kind = 0, dpp_n = 0, row_mask = 15, bank_mask = 15;   // dpp_kind: 1 row_shl, 2 row_shr, 3 row_ror
    bool bound_ctrl = false, sc0 = false, clamp = false;
    uint8_t dst_sel = 6, dst_unused = 0, src0_sel = 6, src1_sel = 6;  // 0-3 BYTE_n, 4-5 WORD_n, 6 DWORD; unused: 0 PAD 1 SEXT 2 PRESERVE
    uint8_t neg_lo = 0, neg_hi = 0, bitop3 = 0, cls = 0;
    uint64_t addr = 0;
    int32_t target = -1;      // instruction index of a branch target
    int32_t simm = 0;
    uint32_t line = 0;        // line in co.s (diagnostics)
};

enum Cls : uint8_t { C_SALU = 0, C_VALU, C_VOP3P, C_LDS, C_VMEM, C_SMEM, C_BRANCH, C_OTHER, C_N };

struct KArg { uint32_t offset, size; std::string kind; };

struct Kernel {
    std::string name;
    uint64_t code_addr = 0;
    uint32_t lds = 0, scratch = 0, kernarg = 0, rsrc1 = 0, rsrc2 = 0, rsrc3 = 0, props = 0, preload = 0;
    std::vector<KArg> args;
    std::vector<Inst> code;   // parsed lazily; device functions the kernel calls are appended when first called
    std::unordered_map<uint64_t, int> at;       // instruction address -> index into code
    bool parsed = false;
    std::string sfile;
    // statistics (wave instructions)
    uint64_t launches = 0, wgs = 0, count[C_N] = {0};
};

struct Dim3 { uint32_t x, y, z; };

// run one dispatch to completion; returns "" or the description of the fault that ended it
std::string run_kernel(Kernel& k, Dim3 grid, Dim3 block, const uint8_t* kernarg, uint32_t dyn_lds);
bool parse_kernel(Kernel& k, std::string& err);
int parse_function_at(Kernel& k, uint64_t addr, std::string& err);   // index of the instruction at addr (parsing its function if needed), -1 on error

// memory map of the stand-in runtime (every device access is checked against it)
bool mem_ok(uint64_t addr, uint64_t n);
std::string mem_describe(uint64_t addr);
// HIPSIM_MEM_UNDEF=1: one shadow byte per byte of device memory (1 = written by a host copy, a memset or a kernel); nullptr when
// the mode is off or the range is page-locked host memory.  Call after mem_ok(addr, n) said yes.
uint8_t* mem_shadow(uint64_t addr);

}  // namespace sim
