// TEST DRIVER, not product code: the BLAST v4 reader (swipe_amd/csrc/blastdb.cpp) against damaged database files, under
// AddressSanitizer + UndefinedBehaviorSanitizer on a machine without a GPU.  A real NCBI database arrives from a
// download that may have been cut short; the reference answers damaged files with fatal() or a crash
// (database.cc:566-601 trusts the index).  The library must answer with a status: every call below may fail, none may
// read outside a mapped file.  Exit 0 = no sanitizer report and no uncaught exception over all rounds.
//   usage: blastdb_fuzz <scratch directory> <rounds> <seed> [<database base name>:<symtype> ...]
#include "../../include/swipe_amd.h"
#include "../../swipe_amd/csrc/host_util.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

namespace swa {
static thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace swa
extern "C" const char* swa_last_error(void) { return swa::g_err.c_str(); }

static std::vector<uint8_t> slurp(const std::string& path)
{
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& path, const std::vector<uint8_t>& v)
{
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  f.write(reinterpret_cast<const char*>(v.data()), std::streamsize(v.size()));
}

// every reader entry point that needs no device, over whatever the files hold now; returns how many calls succeeded
static int walk(const std::string& base, int symtype, std::mt19937_64& rng)
{
  int ok = 0;
  std::vector<int64_t> off;
  if (swa::read_blast_lengths(base.c_str(), symtype, off) == SWA_OK) ++ok;
  swa::HostDb db;
  if (swa::read_blast_db(base.c_str(), symtype, 0, -1, db) == SWA_OK) {
    ++ok;
    uint64_t sum = 0;                                   // touch every residue the reader claims to have produced
    for (size_t i = 0; i < db.residues.size(); ++i) sum += db.residues.data()[i];
    if (sum == 0xdeadbeefdeadbeefULL) std::puts("");
  }
  swa::HostDb part;
  if (swa::read_blast_db(base.c_str(), symtype, int64_t(rng() % 8), int64_t(rng() % 40), part) == SWA_OK) ++ok;
  swa_headers* h = nullptr;
  if (swa_headers_open(base.c_str(), symtype, nullptr, &h) == SWA_OK && h) {
    ++ok;
    int64_t nseq = 0, nsym = 0, mseq = 0, msym = 0, longest = 0;
    char title[64], stamp[64];
    swa_headers_info(h, &nseq, &nsym, &mseq, &msym, &longest, title, sizeof title);
    swa_headers_time(h, stamp, sizeof stamp);
    const int64_t n = nseq < 64 ? nseq : 64;
    std::vector<uint8_t> inc(size_t(n) + 1);
    swa_headers_inclusion(h, 0, n, inc.data());
    std::vector<char> buf(256);
    for (int64_t s = -1; s <= n; ++s)
      for (int flags = 0; flags < 2; ++flags) {
        int64_t need = 0;
        int rc = swa_headers_get(h, s, flags, buf.data(), int64_t(buf.size()), &need);
        if (rc == SWA_ERANGE && need > 0 && need < (1 << 24)) {
          std::vector<char> big(static_cast<size_t>(need), '\0');
          rc = swa_headers_get(h, s, flags, big.data(), need, &need);
        }
        if (rc == SWA_OK) ++ok;
      }
    swa_headers_close(h);
  }
  std::vector<int64_t> seqnos = {0, 1, int64_t(rng() % 50), -3, 1 << 20}, lens;
  std::vector<std::string> lines;
  if (swa::read_blast_deflines(base.c_str(), symtype, seqnos, lines, lens) == SWA_OK) ++ok;
  return ok;
}

// one database (its three files) damaged `rounds` times, one file at a time, restored after each walk
static long damage(const std::string& base, int symtype, int rounds, std::mt19937_64& rng, long* intact)
{
  const char* ext_aa[] = {".pin", ".psq", ".phr"};
  const char* ext_nt[] = {".nin", ".nsq", ".nhr"};
  const char** ext = symtype ? ext_aa : ext_nt;
  std::vector<uint8_t> orig[3];
  for (int k = 0; k < 3; ++k) orig[k] = slurp(base + ext[k]);
  *intact += walk(base, symtype, rng);
  long survived = 0;
  for (int r = 0; r < rounds; ++r) {
    const int k = int(rng() % 3);
    std::vector<uint8_t> v = orig[k];
    if (v.empty()) continue;
    const int how = int(rng() % 5);
    if (how == 0) v.resize(size_t(rng() % v.size()));                                                                        // cut short
    else if (how == 1) for (int i = 0; i < 1 + int(rng() % 4); ++i) v[size_t(rng() % v.size())] = uint8_t(rng());          // a few bytes
    else if (how == 2) for (int i = 0; i < 4; ++i) v[size_t(rng() % v.size())] = uint8_t(rng() % 2 ? 0xff : 0x00);          // extremes
    else if (how == 3) { const size_t at = size_t(rng() % v.size()); for (size_t i = at; i < v.size() && i < at + 4; ++i) v[i] = 0xff; }   // a huge big-endian word
    else { const size_t at = size_t(rng() % v.size()); v.insert(v.begin() + long(at), size_t(1 + rng() % 9), uint8_t(rng())); }        // shifted tail
    spit(base + ext[k], v);
    survived += walk(base, symtype, rng);
    spit(base + ext[k], orig[k]);
  }
  return survived;
}

// usage: blastdb_fuzz <scratch directory> <rounds> <seed> [<database base name>:<symtype> ...]
// The extra databases (written by the caller: every Seq-id flavour in the headers, ambiguity runs, a masked alias) are
// damaged like the two generated here.
int main(int argc, char** argv)
{
  if (argc < 4) { std::fprintf(stderr, "usage: blastdb_fuzz <dir> <rounds> <seed> [base:symtype ...]\n"); return 2; }
  const std::string dir = argv[1];
  const int rounds = std::atoi(argv[2]);
  std::mt19937_64 rng(uint64_t(std::atoll(argv[3])));
  long intact = 0, survived = 0;
  for (int symtype = 0; symtype <= 1; ++symtype) {
    // a small valid database: 37 sequences of 0..90 residues, written by the library's own writer
    const int nseq = 37;
    std::vector<int64_t> off(size_t(nseq) + 1, 0);
    for (int s = 0; s < nseq; ++s) off[size_t(s) + 1] = off[size_t(s)] + int64_t(s == 5 ? 0 : rng() % 91);
    std::vector<uint8_t> res(size_t(off[size_t(nseq)]) + 1);
    for (auto& c : res) c = symtype ? uint8_t(1 + rng() % 24) : uint8_t(1u << (rng() % 4));
    const std::string base = dir + (symtype ? "/fz_aa" : "/fz_nt");
    if (swa_blastdb_write(base.c_str(), symtype, res.data(), off.data(), nseq, 1, "fuzz") != SWA_OK) {
      std::fprintf(stderr, "writer failed: %s\n", swa_last_error());
      return 1;
    }
    survived += damage(base, symtype, rounds, rng, &intact);
    // alias files: no volumes, a volume that does not exist, a mask without a file, nonsense numbers
    const char* alias[] = {"TITLE x\nDBLIST\n", "TITLE x\nDBLIST nowhere\n", "TITLE x\nDBLIST %s\nOIDLIST nomask\nMEMB_BIT 1\nMAXOID 99999999999\n",
                           "DBLIST %s %s\nNSEQ -5\nLENGTH 99999999999999999999\n", "GILIST x\nDBLIST %s\n"};
    const std::string leaf = symtype ? "fz_aa" : "fz_nt";
    for (const char* a : alias) {
      char text[512];
      std::snprintf(text, sizeof text, a, leaf.c_str(), leaf.c_str());
      const std::string ab = dir + (symtype ? "/fz_alias_aa" : "/fz_alias_nt");
      spit(ab + (symtype ? ".pal" : ".nal"), std::vector<uint8_t>(text, text + std::strlen(text)));
      survived += walk(ab, symtype, rng);
    }
  }
  for (int a = 4; a < argc; ++a) {
    const std::string arg = argv[a];
    const size_t colon = arg.rfind(':');
    if (colon == std::string::npos) { std::fprintf(stderr, "expected base:symtype, got %s\n", argv[a]); return 2; }
    const long before = intact;
    survived += damage(arg.substr(0, colon), std::atoi(arg.c_str() + colon + 1), rounds, rng, &intact);
    if (intact == before) { std::fprintf(stderr, "no call succeeded on the intact database %s: %s\n", argv[a], swa_last_error()); return 1; }
  }
  std::printf("intact calls ok %ld, calls ok on damaged files %ld, rounds %d: no sanitizer report\n", intact, survived, rounds);
  return intact > 0 ? 0 : 1;
}
