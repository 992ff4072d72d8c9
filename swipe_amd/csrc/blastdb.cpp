// BLAST database version 4 reader: the wire format the loader ingests.
//
// Follows what the reference accepts: alias files (.pal/.nal: TITLE / DBLIST, one level of
// nesting, database.cc:406-489 and 775-925), index files (.pin/.nin, database.cc:566-601:
// everything big-endian except the little-endian residue total at 595) and sequence files
// (.psq: NCBIstdaa bytes, NUL-terminated entries; .nsq: 2 bits per base, remainder count in
// the last byte, optional ambiguity runs, database.cc:1237-1323), OID masks behind a MEMB_BIT alias
// (database.cc:670-716), taxid lists (718-772) and the binary ASN.1 definition lines (asnparse.cc).
#include "../../include/swipe_amd.h"
#include "host_util.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <thread>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
struct Mapped {
  const uint8_t* p = nullptr;
  size_t n = 0;
  ~Mapped() { if (p) munmap(const_cast<uint8_t*>(p), n); }
  bool open(const std::string& path)
  {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); return false; }
    n = size_t(st.st_size);
    if (n) {
      void* m = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
      if (m == MAP_FAILED) { ::close(fd); n = 0; return false; }
      p = static_cast<const uint8_t*>(m);
    }
    ::close(fd);
    return n > 0;
  }
};

inline uint32_t be32(const uint8_t* p) { return uint32_t(p[0]) << 24 | uint32_t(p[1]) << 16 | uint32_t(p[2]) << 8 | p[3]; }
inline uint64_t be64(const uint8_t* p) { return uint64_t(be32(p)) << 32 | be32(p + 4); }

struct Volume {
  std::string base;
  Mapped index, seq, hdr;
  const uint8_t* hdr_off = nullptr;    // big-endian u32 [nseq + 1]
  int64_t nseq = 0, nsym = 0, longest = 0;
  const uint8_t* seq_off = nullptr;    // big-endian u32 [nseq + 1]
  const uint8_t* amb_off = nullptr;    // nucleotide only
  std::string title, time;
};

bool open_volume(const std::string& base, bool protein, Volume& v, std::string& err)
{
  v.base = base;
  if (!v.index.open(base + (protein ? ".pin" : ".nin"))) { err = "Unable to open file " + base + (protein ? ".pin." : ".nin."); return false; }
  if (!v.seq.open(base + (protein ? ".psq" : ".nsq"))) { err = "Unable to open file " + base + (protein ? ".psq." : ".nsq."); return false; }
  const uint8_t* p = v.index.p;
  const uint8_t* end = p + v.index.n;
  if (v.index.n < 32 || be32(p) != 4) { err = "Illegal database version (must be 4)."; return false; }
  size_t o = 8;
  const uint32_t tl = be32(p + o); o += 4;
  if (o + tl + 4 > v.index.n) { err = "Truncated database index."; return false; }
  v.title.assign(reinterpret_cast<const char*>(p + o), tl); o += tl;
  const uint32_t dl = be32(p + o); o += 4;
  if (o + dl + 4 > v.index.n) { err = "Truncated database index."; return false; }
  v.time.assign(reinterpret_cast<const char*>(p + o), dl); o += dl;
  o = (o + 3) & ~size_t(3);                                          // database.cc:587-592
  if (o + 16 > v.index.n) { err = "Truncated database index."; return false; }
  v.nseq = be32(p + o); o += 4;
  uint64_t total = 0;
  std::memcpy(&total, p + o, 8); o += 8;                             // little-endian, database.cc:595
  v.nsym = int64_t(total);
  v.longest = be32(p + o); o += 4;
  const size_t tab = size_t(v.nseq + 1) * 4;
  if (p + o + tab * (protein ? 2 : 3) > end) { err = "Truncated database index."; return false; }
  v.hdr_off = p + o;
  v.seq_off = p + o + tab;
  v.amb_off = protein ? nullptr : p + o + 2 * tab;
  return true;
}

// Length of sequence s of a volume out of its index, every offset checked against the mapped files: a database cut
// short by a download must end in a status, not in a read beyond the mapping (the reference trusts the index).
int sequence_length(const Volume& v, bool protein, int64_t s, int64_t* len)
{
  const uint64_t o1 = be32(v.seq_off + 4 * s), o2 = be32(v.seq_off + 4 * (s + 1));
  if (o2 < o1 || o2 > v.seq.n) return swa::fail(SWA_EIO, "corrupt sequence offsets in " + v.base);
  if (protein) {
    *len = o2 > o1 ? int64_t(o2 - o1 - 1) : 0;                         // entry includes its NUL terminator
    return SWA_OK;
  }
  const uint64_t o3 = be32(v.amb_off + 4 * s);
  if (o3 <= o1 || o3 > o2) return swa::fail(SWA_EIO, "corrupt ambiguity offsets in " + v.base);
  const size_t packed = size_t(o3 - o1);
  *len = int64_t(4 * (packed - 1) + (v.seq.p[o1 + packed - 1] & 3));   // database.cc:1260-1261
  return SWA_OK;
}

// One alias file (database.cc:406-489): TITLE, DBLIST, OIDLIST, LENGTH, NSEQ, MAXOID, MEMB_BIT
struct Alias {
  bool present = false;
  std::string title;
  std::vector<std::string> dblist, oidlist;
  int64_t length = 0, nseq = 0, maxoid = 0, memb_bit = 0;
};

std::vector<std::string> words(const char* text)
{
  std::vector<std::string> out;
  std::string cur;
  for (const char* p = text; *p; ++p) {
    if (std::strchr(" \t\r\n", *p)) { if (!cur.empty()) out.push_back(cur), cur.clear(); }
    else cur += *p;
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

Alias read_alias(const std::string& base, bool protein, std::string* error)
{
  Alias a;
  FILE* f = std::fopen((base + (protein ? ".pal" : ".nal")).c_str(), "r");
  if (!f) return a;
  a.present = true;
  char line[10000];
  while (std::fgets(line, sizeof line, f)) {
    if (std::strncmp(line, "TITLE ", 6) == 0) {
      std::string t(line + 6);
      const size_t b = t.find_first_not_of(" \t");
      t = b == std::string::npos ? std::string() : t.substr(b);
      a.title = t.substr(0, t.find_first_of("\r\n"));
    } else if (std::strncmp(line, "DBLIST", 6) == 0) a.dblist = words(line + 6);
    else if (std::strncmp(line, "OIDLIST", 7) == 0) a.oidlist = words(line + 7);
    else if (std::strncmp(line, "GILIST", 6) == 0) { if (error) *error = "GILIST in database alias files not implemented."; }
    else if (std::strncmp(line, "LENGTH ", 7) == 0) a.length = std::atol(line + 7);
    else if (std::strncmp(line, "NSEQ ", 5) == 0) a.nseq = std::atol(line + 5);
    else if (std::strncmp(line, "MAXOID ", 7) == 0) a.maxoid = std::atol(line + 7);
    else if (std::strncmp(line, "MEMB_BIT ", 9) == 0) a.memb_bit = std::atol(line + 9);
  }
  std::fclose(f);
  if (a.title.empty()) a.title = base;
  return a;
}

std::string dir_of(const std::string& base)
{
  const size_t s = base.rfind('/');
  return s == std::string::npos ? std::string() : base.substr(0, s + 1);
}

// taxid list file -> bitmap (db_read_taxid_file, database.cc:733-772)
bool read_taxid_bitmap(const char* path, std::vector<uint8_t>& bitmap, std::string& err)
{
  FILE* f = std::fopen(path, "r");
  if (!f) { err = std::string("Unable to open taxid file ") + path + "."; return false; }
  bitmap.assign(64 * 1024, 0);
  unsigned long taxid;
  while (std::fscanf(f, "%lu\n", &taxid) > 0) {
    const size_t byte = taxid / 8;
    if (byte >= bitmap.size()) bitmap.resize(byte + 1, 0);
    bitmap[byte] = uint8_t(bitmap[byte] | (1u << (taxid & 7)));
  }
  std::fclose(f);
  return true;
}

// A database as db_open leaves it (database.cc:775-911): volumes behind at most two alias levels, an OID
// mask per volume when the top alias names a membership bit, optional taxid list.
struct BlastDb {
  bool protein = true;
  std::vector<Volume> vols;
  std::vector<Mapped> masks;            // one per volume when memb_bit != 0
  std::vector<int64_t> maxoid;
  std::string title;
  int64_t memb_bit = 0;
  int64_t nseq = 0, nsym = 0, longest = 0, masked_nseq = 0, masked_nsym = 0;
  bool have_taxids = false;
  std::vector<uint8_t> taxids;

  int open(const char* basename, int symtype, const char* taxidfile, bool want_headers)
  {
    if (!basename) return swa::fail(SWA_EINVAL, "null database name");
    if (symtype != SWA_SYMTYPE_PROTEIN && symtype != SWA_SYMTYPE_NUCLEOTIDE)
      return swa::fail(SWA_EINVAL, "database files are nucleotide (0) or protein (1)");
    protein = symtype == SWA_SYMTYPE_PROTEIN;
    const std::string base(basename), dir = dir_of(base);
    std::string err;
    struct Entry { std::string base, msk; int64_t mnseq, mlen, maxoid; };
    std::vector<Entry> entries;
    Alias top = read_alias(base, protein, &err);
    if (!err.empty()) return swa::fail(SWA_EIO, err);
    if (!top.present) {
      entries.push_back({base, "", 0, 0, 0});
    } else {
      title = top.title;
      memb_bit = top.memb_bit;
      for (size_t i = 0; i < top.dblist.size(); ++i) {
        Alias sub = read_alias(dir + top.dblist[i], protein, &err);
        if (!err.empty()) return swa::fail(SWA_EIO, err);
        if (sub.present) {
          if (top.memb_bit && (sub.oidlist.size() != 1 || sub.dblist.size() != 1)) return swa::fail(SWA_EIO, "Illegal alias file (2).");
          for (size_t j = 0; j < sub.dblist.size(); ++j)
            entries.push_back({dir + sub.dblist[j], top.memb_bit ? dir + sub.oidlist[j] : "", sub.nseq, sub.length, sub.maxoid});
        } else {
          if (top.oidlist.empty()) { top.memb_bit = 0; memb_bit = 0; }                 // database.cc:838-842
          if (top.memb_bit && (top.oidlist.size() != 1 || top.dblist.size() != 1)) return swa::fail(SWA_EIO, "Illegal alias file (1).");
          entries.push_back({dir + top.dblist[i], top.memb_bit ? dir + top.oidlist[i] : "", top.nseq, top.length, top.maxoid});
        }
      }
    }
    if (entries.empty()) return swa::fail(SWA_EIO, "alias file lists no volumes");
    if (entries.size() > 256) return swa::fail(SWA_EIO, "too many database volumes");   // database.cc:216
    vols.resize(entries.size());
    masks.resize(memb_bit ? entries.size() : 0);
    maxoid.assign(entries.size(), 0);
    for (size_t i = 0; i < entries.size(); ++i) {
      if (!open_volume(entries[i].base, protein, vols[i], err)) return swa::fail(SWA_EIO, err);
      if (want_headers && !vols[i].hdr.open(entries[i].base + (protein ? ".phr" : ".nhr")))
        return swa::fail(SWA_EIO, "Unable to open file " + entries[i].base + (protein ? ".phr." : ".nhr."));
      nseq += vols[i].nseq;
      nsym += vols[i].nsym;
      longest = std::max(longest, vols[i].longest);
      if (memb_bit) {
        if (!masks[i].open(entries[i].msk)) return swa::fail(SWA_EIO, "Unable to open msk file " + entries[i].msk + ".");
        maxoid[i] = entries[i].maxoid;
        masked_nseq += entries[i].mnseq;
        masked_nsym += entries[i].mlen;
      }
    }
    if (!memb_bit) { masked_nseq = nseq; masked_nsym = nsym; }
    if (title.empty()) title = vols[0].title;
    if (taxidfile && *taxidfile) {
      if (!read_taxid_bitmap(taxidfile, taxids, err)) return swa::fail(SWA_EIO, err);
      have_taxids = true;
    }
    return SWA_OK;
  }

  bool locate(int64_t seqno, size_t* vol, int64_t* local) const
  {
    if (seqno < 0) return false;
    for (size_t i = 0; i < vols.size(); ++i) {
      if (seqno < vols[i].nseq) { *vol = i; *local = seqno; return true; }
      seqno -= vols[i].nseq;
    }
    return false;
  }
  bool taxid_ok(unsigned long taxid) const                       // db_check_taxid, database.cc:718-731
  {
    if (!have_taxids) return true;
    const size_t byte = taxid / 8;
    return byte < taxids.size() && ((taxids[byte] >> (taxid & 7)) & 1);
  }
  bool in_mask(size_t vol, int64_t local) const                  // db_check_msk, database.cc:687-706
  {
    if (!memb_bit) return true;
    if (local > maxoid[vol]) return false;
    const size_t at = 4 + size_t(local >> 3);
    return at < masks[vol].n && ((masks[vol].p[at] >> (7 - (local & 7))) & 1);
  }
};
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// A range of a database opened for reading piece by piece.  read_blast_db takes the whole range in one go; a shard over its
// HBM budget (swa_db_open_streamed) takes it a part at a time, straight into the part's page-locked block - the reference
// maps what it is about to search and nothing else (db_mapsequences, database.cc:1082-1131).
struct swa::RangeReader::Impl {
  BlastDb bd;
  struct Span { const Volume* v; int64_t lo, first; };        // volume, its first local sequence in the range, range index of that
  std::vector<Span> spans;
};
swa::RangeReader::RangeReader() : impl(new Impl) {}
swa::RangeReader::~RangeReader() { delete impl; }

int swa::RangeReader::open(const char* basename, int symtype, int64_t first, int64_t last)
{
  BlastDb& bd = impl->bd;
  const int rc_open = bd.open(basename, symtype, nullptr, false);
  if (rc_open != SWA_OK) return rc_open;
  protein = bd.protein;
  const std::vector<Volume>& V = bd.vols;
  masked = bd.memb_bit != 0;
  masked_seqcount = bd.masked_nseq;
  masked_symcount = bd.masked_nsym;
  total_seqcount = bd.nseq;
  total_symcount = bd.nsym;
  longest = bd.longest;
  title = bd.title.empty() ? V[0].title : bd.title;
  if (first < 0) first = 0;
  if (last < 0 || last >= bd.nseq) last = bd.nseq - 1;
  first_seqno = first;
  offsets.assign(1, 0);
  included.clear();
  impl->spans.clear();
  if (last < first) return SWA_OK;
  // lengths -> offsets (and the OID mask), a few threads per volume, each sequence's own slot, then one running sum
  const int64_t count = last - first + 1;
  offsets.assign(size_t(count) + 1, 0);
  if (masked) included.assign(size_t(count), 0);
  int64_t vbase = 0, done = 0;
  for (const Volume& v : V) {
    const int64_t lo = first > vbase ? first - vbase : 0;
    const int64_t hi = last - vbase < v.nseq - 1 ? last - vbase : v.nseq - 1;
    vbase += v.nseq;
    if (hi < lo) continue;
    const int64_t cnt = hi - lo + 1;
    impl->spans.push_back({&v, lo, done});
    const int64_t nthreads = std::max<int64_t>(1, std::min<int64_t>({int64_t(std::thread::hardware_concurrency()), 16, cnt >> 16}));
    std::vector<int> rcs(size_t(nthreads), SWA_OK);
    std::vector<std::string> errs(static_cast<size_t>(nthreads));
    const size_t vol = size_t(&v - V.data());
    auto walk = [&](int64_t t) {
      for (int64_t k = cnt * t / nthreads; k < cnt * (t + 1) / nthreads; ++k) {
        const int rc_len = sequence_length(v, protein, lo + k, &offsets[size_t(done + k) + 1]);
        if (rc_len != SWA_OK) { rcs[size_t(t)] = rc_len; errs[size_t(t)] = swa_last_error(); return; }
        if (masked) included[size_t(done + k)] = bd.in_mask(vol, lo + k) ? 1 : 0;
      }
    };
    if (nthreads == 1) walk(0);
    else {
      std::vector<std::thread> pool;
      for (int64_t t = 0; t < nthreads; ++t) pool.emplace_back(walk, t);
      for (std::thread& th : pool) th.join();
    }
    for (int64_t t = 0; t < nthreads; ++t) if (rcs[size_t(t)] != SWA_OK) return swa::fail(rcs[size_t(t)], errs[size_t(t)]);
    done += cnt;
  }
  for (int64_t i = 0; i < count; ++i) offsets[size_t(i) + 1] += offsets[size_t(i)];
  return SWA_OK;
}

// Residues of the sequences [from, to) of the range, back to back, residue k of the span at dst[k] - or, nibbles, in the low
// (k even) / high (k odd) half of dst[k >> 1], which is how nucleotide shards are held.  One thread per call; calls on disjoint
// spans may run side by side if, with nibbles, each span starts on an even residue of the buffer they share.  *or_codes
// collects the OR of every residue code (the caller validates: codes index the LDS profile).
int swa::RangeReader::fill(int64_t from, int64_t to, uint8_t* dst, bool nibbles, unsigned* or_codes) const
{
  const std::vector<Impl::Span>& spans = impl->spans;
  if (from >= to) return SWA_OK;
  size_t sp = 0;
  while (sp + 1 < spans.size() && spans[sp + 1].first <= from) ++sp;
  const int64_t base = offsets[size_t(from)];
  unsigned acc = 0;
  std::vector<uint8_t> tmp;
  for (int64_t i = from; i < to; ++i) {
    while (sp + 1 < spans.size() && spans[sp + 1].first <= i) ++sp;
    const Volume& v = *spans[sp].v;
    const int64_t s = spans[sp].lo + (i - spans[sp].first);
    const uint64_t o1 = be32(v.seq_off + 4 * s), o2 = be32(v.seq_off + 4 * (s + 1));
    const int64_t at = offsets[size_t(i)] - base;
    const size_t n = size_t(offsets[size_t(i) + 1] - offsets[size_t(i)]);
    if (protein) {
      if (!n) continue;
      const uint8_t* src = v.seq.p + o1;
      if (!nibbles) {
        std::memcpy(dst + at, src, n);
        if (or_codes) for (size_t k = 0; k < n; ++k) acc |= src[k];
        continue;
      }
      tmp.assign(src, src + n);
    } else {
      const uint64_t o3 = be32(v.amb_off + 4 * s);
      const uint8_t* body = v.seq.p + o1;
      uint8_t* out = dst + at;
      if (nibbles) { tmp.resize(n); out = tmp.data(); }
      for (size_t k = 0; k < n; ++k)
        out[k] = uint8_t(1u << ((body[k >> 2] >> ((3 - (k & 3)) << 1)) & 3));  // A=1 C=2 G=4 T=8
      if (o2 > o3) {                                                 // ambiguity runs in file order, database.cc:1284-1323
        const uint8_t* a = v.seq.p + o3;
        const size_t bytes = size_t(o2 - o3);
        if (bytes >= 4) {
          const uint32_t hdr = be32(a);
          if (hdr >> 31) {
            for (size_t k = 0; k + 8 <= bytes - 4; k += 8) {
              const uint64_t e = be64(a + 4 + k);
              const uint64_t code = e >> 60, run = ((e >> 48) & 0xfff) + 1, off = e & 0x0000fffffffffffULL;
              for (uint64_t r = 0; r < run && off + r < n; ++r) out[off + r] = uint8_t(code);
            }
          } else {
            for (size_t k = 0; k + 4 <= bytes - 4; k += 4) {
              const uint32_t e = be32(a + 4 + k);
              const uint32_t code = e >> 28, run = ((e >> 24) & 0xf) + 1, off = e & 0x00ffffff;
              for (uint32_t r = 0; r < run && size_t(off) + r < n; ++r) out[off + r] = uint8_t(code);
            }
          }
        }
      }
      if (!nibbles) continue;                                        // (one-hot or 4-bit codes: always in range)
    }
    // two residues per byte: a byte's low half is written first (and clears the high half), its high half is OR-ed in
    for (size_t k = 0; k < n; ++k) {
      const int64_t g = at + int64_t(k);
      const uint8_t c = uint8_t(tmp[k] & 15);
      acc |= tmp[k];
      if (g & 1) dst[g >> 1] = uint8_t(dst[g >> 1] | (c << 4));
      else dst[g >> 1] = c;
    }
  }
  if (or_codes) *or_codes |= acc;
  return SWA_OK;
}

// The mapped pages of the sequence files behind [from, to) leave this process's resident set (they stay in the page cache):
// a shard that is read once into page-locked memory should not be counted twice.
void swa::RangeReader::forget(int64_t from, int64_t to) const
{
  const std::vector<Impl::Span>& spans = impl->spans;
  const int64_t page = 4096;
  for (size_t sp = 0; sp < spans.size(); ++sp) {
    const int64_t first = spans[sp].first, next = sp + 1 < spans.size() ? spans[sp + 1].first : int64_t(offsets.size()) - 1;
    const int64_t a = std::max(from, first), b = std::min(to, next);
    if (a >= b) continue;
    const Volume& v = *spans[sp].v;
    const int64_t o1 = int64_t(be32(v.seq_off + 4 * (spans[sp].lo + a - first))), o2 = int64_t(be32(v.seq_off + 4 * (spans[sp].lo + b - first)));
    const int64_t lo = (o1 + page - 1) / page * page, hi = o2 / page * page;      // whole pages inside the byte range only
    if (hi > lo) (void)madvise(const_cast<uint8_t*>(v.seq.p) + lo, size_t(hi - lo), MADV_DONTNEED);
  }
}

int swa::read_blast_db(const char* basename, int symtype, int64_t first_seqno, int64_t last_seqno, HostDb& out)
{
  RangeReader rd;
  const int rc_open = rd.open(basename, symtype, first_seqno, last_seqno);
  if (rc_open != SWA_OK) return rc_open;
  out.masked = rd.masked;
  out.masked_seqcount = rd.masked_seqcount;
  out.masked_symcount = rd.masked_symcount;
  out.total_seqcount = rd.total_seqcount;
  out.total_symcount = rd.total_symcount;
  out.longest = rd.longest;
  out.title = rd.title;
  out.first_seqno = rd.first_seqno;
  out.offsets = rd.offsets;
  out.included = rd.included;
  out.residues.clear();
  const size_t total = out.offsets.size() - 1;
  if (!total) return SWA_OK;
  out.residues.resize(size_t(out.offsets.back()));
  const size_t nthreads = std::max<size_t>(1, std::min<size_t>({size_t(std::thread::hardware_concurrency()), size_t(32),
                                                                 size_t(out.residues.size() >> 24) + 1}));
  if (nthreads == 1) return rd.fill(0, int64_t(total), out.residues.data(), false, nullptr);
  std::vector<std::thread> pool;                                     // split by residues, not by sequence count
  size_t from = 0;
  for (size_t t = 0; t < nthreads; ++t) {
    const int64_t target = out.offsets.back() * int64_t(t + 1) / int64_t(nthreads);
    const size_t to = t + 1 == nthreads ? total
                                        : size_t(std::lower_bound(out.offsets.begin(), out.offsets.end(), target) - out.offsets.begin());
    const size_t end = std::min(std::max(to, from), total);
    pool.emplace_back([&rd, &out, from, end]() { (void)rd.fill(int64_t(from), int64_t(end), out.residues.data() + out.offsets[from], false, nullptr); });
    from = end;
  }
  for (std::thread& th : pool) th.join();
  return SWA_OK;
}

// The index half of an open, for the pipelined loader (swipe_amd.cpp "loading"): lengths of the wanted range as prefix sums
// and the file ranges that hold its entries.  Everything is checked against the mapped files here, so that the loader's
// reader threads never leave them.
int swa::plan_blast_load(const char* basename, int symtype, int64_t first_seqno, int64_t last_seqno, LoadPlan& out)
{
  BlastDb bd;
  const int rc_open = bd.open(basename, symtype, nullptr, false);
  if (rc_open != SWA_OK) return rc_open;
  out = LoadPlan{};
  out.total_seqcount = bd.nseq;
  out.total_symcount = bd.nsym;
  out.nucleotide = !bd.protein;
  out.masked = bd.memb_bit != 0;
  out.masked_seqcount = bd.masked_nseq;
  out.masked_symcount = bd.masked_nsym;
  if (first_seqno < 0) first_seqno = 0;
  if (last_seqno < 0 || last_seqno >= bd.nseq) last_seqno = bd.nseq - 1;
  out.first_seqno = first_seqno;
  if (last_seqno < first_seqno) return SWA_OK;                                              // regular stays false
  const int64_t n = last_seqno - first_seqno + 1;
  out.offsets.assign(size_t(n) + 1, 0);
  if (out.nucleotide) out.raw_seq.assign(size_t(n) + 1, 0);
  if (out.masked) out.included.assign(size_t(n), 0);
  int64_t vbase = 0, done = 0, residues = 0, raw = 0;
  for (const Volume& v : bd.vols) {
    const int64_t lo = first_seqno > vbase ? first_seqno - vbase : 0;
    const int64_t hi = last_seqno - vbase < v.nseq - 1 ? last_seqno - vbase : v.nseq - 1;    // inclusive
    vbase += v.nseq;
    if (hi < lo) continue;
    const int64_t cnt = hi - lo + 1;
    const size_t vol = size_t(&v - bd.vols.data());
    const uint64_t o_lo = be32(v.seq_off + 4 * lo), o_hi = be32(v.seq_off + 4 * (hi + 1));
    if (o_hi > v.seq.n || o_hi < o_lo) return SWA_OK;                                      // read_blast_db reports it
    // protein: entry s = [o_s, o_{s+1}), at least its terminator; lengths o_{s+1} - o_s - 1, from the index alone.
    // nucleotide: entry s = [o_s, o_{s+1}) = packed bases [o_s, a_s) + ambiguity data [a_s, o_{s+1}); the length wants the
    // remainder count in the last packed byte (database.cc:1260-1261): one byte of the sequence file per sequence.
    // A few threads over the index; lengths land in dst[i + 1], the running sum follows.
    const int64_t nthreads = std::max<int64_t>(1, std::min<int64_t>({int64_t(std::thread::hardware_concurrency()), 16, cnt >> 16}));
    std::vector<int64_t> longest(size_t(nthreads), 0), longest_entry(size_t(nthreads), 0);
    std::vector<uint8_t> bad(size_t(nthreads), 0);
    int64_t* dst = out.offsets.data() + done;
    int64_t* rdst = out.nucleotide ? out.raw_seq.data() + done : nullptr;
    uint8_t* inc = out.masked ? out.included.data() + done : nullptr;
    auto walk = [&](int64_t t) {
      const int64_t a = cnt * t / nthreads, b = cnt * (t + 1) / nthreads;
      uint64_t prev = be32(v.seq_off + 4 * (lo + a));
      int64_t lg = 0, le = 0;
      bool wrong = false;
      for (int64_t i = a; i < b; ++i) {
        const uint64_t next = be32(v.seq_off + 4 * (lo + i + 1));
        wrong |= next <= prev;
        int64_t len;
        if (out.nucleotide) {
          const uint64_t amb = be32(v.amb_off + 4 * (lo + i));
          if (amb <= prev || amb > next || next > v.seq.n) { wrong = true; len = 0; }
          else len = int64_t(4 * (amb - prev - 1) + (v.seq.p[amb - 1] & 3));
          rdst[i] = raw + int64_t(prev - o_lo);
          le = std::max<int64_t>(le, int64_t(next - prev));
        } else {
          len = int64_t(next - prev) - 1;
        }
        lg = std::max(lg, len);
        dst[i + 1] = len;
        if (inc) inc[i] = bd.in_mask(vol, lo + i) ? 1 : 0;
        prev = next;
      }
      longest[size_t(t)] = lg;
      longest_entry[size_t(t)] = le;
      bad[size_t(t)] = wrong;
    };
    if (nthreads == 1) walk(0);
    else {
      std::vector<std::thread> pool;
      for (int64_t t = 0; t < nthreads; ++t) pool.emplace_back(walk, t);
      for (std::thread& th : pool) th.join();
    }
    for (int64_t t = 0; t < nthreads; ++t) {
      if (bad[size_t(t)]) return SWA_OK;                                                   // not back to back: the old reader decides
      out.longest = std::max(out.longest, longest[size_t(t)]);
      out.longest_entry = std::max(out.longest_entry, longest_entry[size_t(t)]);
    }
    for (int64_t i = 0; i < cnt; ++i) { residues += dst[i + 1]; dst[i + 1] = residues; }  // lengths -> prefix sums
    LoadPiece piece;
    piece.path = v.base + (bd.protein ? ".psq" : ".nsq");
    piece.file_begin = int64_t(o_lo);
    piece.file_end = int64_t(o_hi);
    piece.first = done;
    out.pieces.push_back(piece);
    raw += int64_t(o_hi - o_lo);
    done += cnt;
  }
  if (done != n) return SWA_OK;
  if (out.nucleotide) out.raw_seq[size_t(n)] = raw;
  out.regular = true;
  return SWA_OK;
}

// sequence and residue totals of a database out of its index headers (no walk over the sequences)
int swa::read_blast_totals(const char* basename, int symtype, int64_t* nseq, int64_t* nsym)
{
  BlastDb bd;
  const int rc_open = bd.open(basename, symtype, nullptr, false);
  if (rc_open != SWA_OK) return rc_open;
  if (nseq) *nseq = bd.nseq;
  if (nsym) *nsym = bd.nsym;
  return SWA_OK;
}

// Sequence lengths alone, out of the index files (pass 1 of read_blast_db): offsets[s] = residues before sequence s.
// What the multi-device group needs to cut a database into residue-balanced shards before any shard is read.
int swa::read_blast_lengths(const char* basename, int symtype, std::vector<int64_t>& offsets)
{
  BlastDb bd;
  const int rc_open = bd.open(basename, symtype, nullptr, false);
  if (rc_open != SWA_OK) return rc_open;
  offsets.assign(size_t(bd.nseq) + 1, 0);
  int64_t done = 0;
  for (const Volume& v : bd.vols) {
    // lengths on a few threads (each sequence's own slot), then one running sum
    const int64_t nthreads = std::max<int64_t>(1, std::min<int64_t>({int64_t(std::thread::hardware_concurrency()), 16, v.nseq >> 16}));
    std::vector<int> rcs(size_t(nthreads), SWA_OK);
    std::vector<std::string> errs(static_cast<size_t>(nthreads));
    int64_t* dst = offsets.data() + done + 1;
    auto walk = [&](int64_t t) {
      for (int64_t s = v.nseq * t / nthreads; s < v.nseq * (t + 1) / nthreads; ++s) {
        const int rc_len = sequence_length(v, bd.protein, s, dst + s);
        if (rc_len != SWA_OK) { rcs[size_t(t)] = rc_len; errs[size_t(t)] = swa_last_error(); return; }
      }
    };
    if (nthreads == 1) walk(0);
    else {
      std::vector<std::thread> pool;
      for (int64_t t = 0; t < nthreads; ++t) pool.emplace_back(walk, t);
      for (std::thread& th : pool) th.join();
    }
    for (int64_t t = 0; t < nthreads; ++t) if (rcs[size_t(t)] != SWA_OK) return swa::fail(rcs[size_t(t)], errs[size_t(t)]);
    done += v.nseq;
  }
  for (int64_t s = 0; s < bd.nseq; ++s) offsets[size_t(s) + 1] += offsets[size_t(s)];
  return SWA_OK;
}

// ---- definition lines -----------------------------------------------------------------------
// BER walker for Blast-def-line-set (reference asnparse.cc:94-1095).  Every element of these headers is
// context-tagged with indefinite length, primitives are short definite; the walker accepts both forms.
namespace {
struct Ber {
  // Damaged headers must not take the walker outside [p, end): every advance is clamped to `end`, a length that does
  // not fit what is left of the buffer means "to the end of the buffer", nesting deeper than any header of the
  // formatter (they reach 8 levels) ends the walk.
  const uint8_t* p;
  const uint8_t* end;
  size_t left() const { return size_t(end - p); }
  void advance(size_t n) { p = n < left() ? p + n : end; }
  bool eoc() const { return left() >= 2 && p[0] == 0 && p[1] == 0; }
  bool more(const uint8_t* stop) const { return p < end && (stop ? p < stop : !eoc()); }
  int peek() const { return p < end ? *p : -1; }
  // reads tag + length; content length (at most what is left) or -1 for the indefinite form
  bool head(int& tag, long& len)
  {
    if (p >= end) return false;
    tag = *p++;
    if (p >= end) return false;
    const int l = *p++;
    if (l == 0x80) { len = -1; return true; }
    size_t v = size_t(l);
    if (l & 0x80) {
      int n = l & 0x7f;
      v = 0;
      bool huge = false;
      while (n-- && p < end) { huge = huge || (v >> 48) != 0; v = (v << 8) | *p++; }
      if (huge) v = left();
    }
    len = long(std::min(v, left()));
    return true;
  }
  const uint8_t* stop_of(long len) const { return len >= 0 ? p + std::min(size_t(len), left()) : nullptr; }
  void skip(int depth = 0)
  {
    int tag; long len;
    if (depth > 64 || !head(tag, len)) { p = end; return; }
    if (len >= 0) { advance(size_t(len)); return; }
    while (p < end && !eoc()) skip(depth + 1);
    advance(2);
  }
  // leaves a constructed element entered with head(): to its end, past the end-of-contents octets if indefinite
  void leave(const uint8_t* stop)
  {
    if (stop) { p = stop; return; }
    while (p < end && !eoc()) skip();
    advance(2);
  }
  std::string str()
  {
    int t; long l;
    if (!head(t, l) || l < 0) return std::string();
    std::string s(reinterpret_cast<const char*>(p), size_t(l));       // head() keeps l within the buffer
    advance(size_t(l));
    return s;
  }
  unsigned long integer() { int t; long l; unsigned long v = 0; if (!head(t, l) || l < 0) return 0; for (long i = 0; i < l && p < end; ++i) v = (v << 8) | *p++; return v; }
  // [tag] EXPLICIT string / integer if it is the next element
  bool opt_str(int tag, std::string& out)
  {
    if (peek() != tag) return false;
    int t; long l;
    head(t, l);
    const uint8_t* stop = stop_of(l);
    out = str();
    leave(stop);
    return true;
  }
  bool opt_int(int tag, unsigned long& out)
  {
    if (peek() != tag) return false;
    int t; long l;
    head(t, l);
    const uint8_t* stop = stop_of(l);
    out = integer();
    leave(stop);
    return true;
  }
};

// Object-id ::= CHOICE { id [0] INTEGER, str [1] VisibleString }  (parse_object_id, asnparse.cc:236-256)
std::string object_id(Ber& b)
{
  std::string s;
  unsigned long v = 0;
  if (b.opt_str(0xA1, s)) return s;
  if (b.opt_int(0xA0, v)) return std::to_string(v);
  return "0";
}

// One Seq-id rendered as the reference prints it (parse_seq_id + show_*, asnparse.cc:615-751); empty for a gi
// when gi's are not shown
std::string seq_id(Ber& b, bool show_gis)
{
  static const char* const names[] = {"lcl", "bbs", "bbm", "gim", "gb", "emb", "pir", "sp", "pat", "ref",
                                      "gnl", "gi", "dbj", "prf", "pdb", "tpg", "tpe", "tpd", "gpp", "nat"};
  int tag; long len;
  if (!b.head(tag, len)) return std::string();
  const uint8_t* stop = b.stop_of(len);
  const int choice = tag - 0xA0;
  const std::string db = (choice >= 0 && choice < 20) ? names[choice] : "";
  std::string out;
  int t; long l;
  switch (choice) {
    case 0: out = db + "|" + object_id(b); break;                                   // local
    case 1: case 2: out = db + "|" + std::to_string(b.integer()); break;            // gibbsq, gibbmt
    case 3: {                                                                        // giim: SEQUENCE { id, db?, release? }
      unsigned long id = 0;
      if (b.head(t, l)) { const uint8_t* s2 = b.stop_of(l); b.opt_int(0xA0, id); b.leave(s2); }
      out = db + "|" + std::to_string(id);
      break;
    }
    case 8: {                                                                        // patent
      unsigned long seqno = 0;
      std::string country, number, doctype;
      bool granted = true;
      if (b.head(t, l)) {
        const uint8_t* s2 = b.stop_of(l);
        b.opt_int(0xA0, seqno);
        if (b.peek() == 0xA1 && b.head(t, l)) {                                      // cit Id-pat
          const uint8_t* s3 = b.stop_of(l);
          if (b.head(t, l)) {
            const uint8_t* s4 = b.stop_of(l);
            b.opt_str(0xA0, country);
            if (b.peek() == 0xA1 && b.head(t, l)) {                                  // id CHOICE { number [0], app-number [1] }
              const uint8_t* s5 = b.stop_of(l);
              if (b.opt_str(0xA0, number)) granted = true;
              else if (b.opt_str(0xA1, number)) granted = false;
              b.leave(s5);
            }
            b.opt_str(0xA2, doctype);
            b.leave(s4);
          }
          b.leave(s3);
        }
        b.leave(s2);
      }
      out = std::string(granted ? "pat" : "pgp") + "|" + country + "|" + number + "|" + std::to_string(seqno);
      break;
    }
    case 10: {                                                                       // general: Dbtag { db, tag Object-id }
      std::string dbname, tagv = "0";
      if (b.head(t, l)) {
        const uint8_t* s2 = b.stop_of(l);
        b.opt_str(0xA0, dbname);
        if (b.peek() == 0xA1 && b.head(t, l)) { const uint8_t* s3 = b.stop_of(l); tagv = object_id(b); b.leave(s3); }
        b.leave(s2);
      }
      out = db + "|" + dbname + "|" + tagv;
      break;
    }
    case 11: {                                                                       // gi
      const unsigned long gi = b.integer();
      if (show_gis) out = db + "|" + std::to_string(gi);
      break;
    }
    case 14: {                                                                       // pdb { mol, chain DEFAULT 32, rel? }
      std::string mol;
      unsigned long chain = 32;
      if (b.head(t, l)) { const uint8_t* s2 = b.stop_of(l); b.opt_str(0xA0, mol); b.opt_int(0xA1, chain); b.leave(s2); }
      std::string ch;
      if (chain > 95) { ch += char(chain - 32); ch += char(chain - 32); } else ch += char(chain);   // asnparse.cc:741-744
      out = db + "|" + mol + "|" + ch;
      break;
    }
    default:                                                                         // Textseq-id
      if (!db.empty()) {
        std::string name, acc, release;
        unsigned long version = 0;
        if (b.head(t, l)) {
          const uint8_t* s2 = b.stop_of(l);
          b.opt_str(0xA0, name); b.opt_str(0xA1, acc); b.opt_str(0xA2, release); b.opt_int(0xA3, version);
          b.leave(s2);
        }
        const std::string shown = (db == "sp" && release == "unreviewed") ? "tr" : db;   // show_seq_id, asnparse.cc:596-604
        out = shown + "|" + acc + (version ? "." + std::to_string(version) : std::string()) + "|" + name;
      }
  }
  b.leave(stop);
  return out;
}

struct DeflineFilter {
  bool show_gis = false, show_taxid = false;
  unsigned long memb = 0;                      // the alias's MEMB_BIT, used as a mask (asnparse.cc:990)
  const BlastDb* db = nullptr;                 // taxid list
};

// every Blast-def-line of the set that passes the membership / taxid filter (parse_blast_def_line_set_new,
// asnparse.cc:970-1013), each rendered "ids[|taxid|..] title" (parse_blast_def_line, 753-887), joined by '\n'.
// *count = number of passing lines.
std::string render_deflines(const uint8_t* p, size_t n, const DeflineFilter& f, long* count)
{
  Ber b{p, p + n};
  int tag; long len;
  std::string out;
  long passed = 0;
  if (count) *count = 0;
  if (!b.head(tag, len) || tag != 0x30) return out;                   // Blast-def-line-set
  const uint8_t* set_stop = b.stop_of(len);
  while (b.more(set_stop)) {
    int t2; long l2;
    if (!b.head(t2, l2) || t2 != 0x30) break;                          // one Blast-def-line
    const uint8_t* stop = b.stop_of(l2);
    std::string title = "unnamed protein product", ids;               // asnparse.cc:768
    unsigned long taxid = 0, memb = 0, links = 0;
    b.opt_str(0xA0, title);
    int t; long l;
    if (b.peek() == 0xA1 && b.head(t, l)) {                            // seqid SEQUENCE OF Seq-id
      const uint8_t* s1 = b.stop_of(l);
      if (b.head(t, l)) {
        const uint8_t* s2 = b.stop_of(l);
        while (b.more(s2)) {
          const std::string id = seq_id(b, f.show_gis);
          if (!ids.empty()) ids += "|";                                // asnparse.cc:793-796: even before an empty id
          ids += id;
        }
        b.leave(s2);
      }
      b.leave(s1);
    }
    b.opt_int(0xA2, taxid);
    for (int which = 0; which < 2; ++which) {                          // memberships [3], links [4]: SEQUENCE OF INTEGER, last wins
      if (b.peek() != 0xA3 + which || !b.head(t, l)) continue;
      const uint8_t* s1 = b.stop_of(l);
      if (b.head(t, l)) {
        const uint8_t* s2 = b.stop_of(l);
        while (b.more(s2)) (which ? links : memb) = b.integer();
        b.leave(s2);
      }
      b.leave(s1);
    }
    b.leave(stop);
    if (f.db && !f.db->taxid_ok(taxid)) continue;
    if ((memb & f.memb) != f.memb) continue;
    std::string line = ids;
    if (f.show_taxid) {                                                // asnparse.cc:859-878
      if (taxid) line += "|taxid|" + std::to_string(taxid);
      if (links) line += "|link|" + std::to_string(links);
      if (memb) line += "|memb|" + std::to_string(memb);
    }
    if (!line.empty() && !title.empty()) line += " ";
    line += title;
    if (passed++) out += '\n';
    out += line;
  }
  if (count) *count = passed;
  return out;
}
}  // namespace

struct swa_headers {
  BlastDb db;
};

extern "C" int swa_headers_open(const char* basename, int symtype, const char* taxidfile, swa_headers** out)
try {
  if (!out) return swa::fail(SWA_EINVAL, "null output handle");
  *out = nullptr;
  swa_headers* h = new (std::nothrow) swa_headers;
  if (!h) return swa::fail(SWA_ENOMEM, "out of host memory");
  const int rc = h->db.open(basename, symtype, taxidfile, true);
  if (rc != SWA_OK) { delete h; return rc; }
  *out = h;
  return SWA_OK;
} SWA_CATCH

extern "C" void swa_headers_close(swa_headers* h) { delete h; }

extern "C" int swa_headers_time(const swa_headers* h, char* buf, int64_t cap)      // db_gettime: the first volume's stamp
{
  if (!h || !buf || cap < 1) return swa::fail(SWA_EINVAL, "bad argument");
  std::snprintf(buf, size_t(cap), "%s", h->db.vols[0].time.c_str());
  return SWA_OK;
}

extern "C" int swa_headers_info(const swa_headers* h, int64_t* seqcount, int64_t* symcount, int64_t* masked_seqcount,
                                int64_t* masked_symcount, int64_t* longest, char* title, int64_t title_cap)
try {
  if (!h) return swa::fail(SWA_EINVAL, "null handle");
  if (seqcount) *seqcount = h->db.nseq;
  if (symcount) *symcount = h->db.nsym;
  if (masked_seqcount) *masked_seqcount = h->db.masked_nseq;
  if (masked_symcount) *masked_symcount = h->db.masked_nsym;
  if (longest) *longest = h->db.longest;
  if (title && title_cap > 0) std::snprintf(title, size_t(title_cap), "%s", h->db.title.c_str());
  return SWA_OK;
} SWA_CATCH

namespace {
int header_bytes(const swa_headers* h, int64_t seqno, const uint8_t** p, size_t* n, size_t* vol, int64_t* local)
{
  if (!h->db.locate(seqno, vol, local)) return swa::fail(SWA_EINVAL, "Cant find database volume.");
  const Volume& v = h->db.vols[*vol];
  const uint64_t h1 = be32(v.hdr_off + 4 * *local), h2 = be32(v.hdr_off + 4 * (*local + 1));
  if (h2 < h1 || h2 > v.hdr.n) return swa::fail(SWA_EIO, "corrupt header offsets in " + v.base);
  *p = v.hdr.p + h1;
  *n = size_t(h2 - h1);
  return SWA_OK;
}
}  // namespace

extern "C" int swa_headers_get(const swa_headers* h, int64_t seqno, int flags, char* buf, int64_t buflen, int64_t* needed)
try {
  if (!h || buflen < 0 || (buflen > 0 && !buf) || !needed) return swa::fail(SWA_EINVAL, "bad argument");
  const uint8_t* p = nullptr; size_t n = 0, vol = 0; int64_t local = 0;
  const int rc = header_bytes(h, seqno, &p, &n, &vol, &local);
  if (rc != SWA_OK) return rc;
  DeflineFilter f;
  f.show_gis = flags & SWA_HEADERS_SHOW_GIS;
  f.show_taxid = flags & SWA_HEADERS_SHOW_TAXID;
  f.memb = (unsigned long)h->db.memb_bit;
  f.db = &h->db;
  const std::string d = render_deflines(p, n, f, nullptr);
  *needed = int64_t(d.size()) + 1;
  if (*needed > buflen) return swa::fail(SWA_ERANGE, "defline buffer too small");
  std::memcpy(buf, d.c_str(), d.size() + 1);
  return SWA_OK;
} SWA_CATCH

// db_check_inclusion (database.cc:1465-1481) for the sequences [first_seqno, first_seqno + n)
extern "C" int swa_headers_inclusion(const swa_headers* h, int64_t first_seqno, int64_t n, uint8_t* include)
try {
  if (!h || n < 0 || (n > 0 && !include)) return swa::fail(SWA_EINVAL, "bad argument");
  DeflineFilter f;
  f.memb = (unsigned long)h->db.memb_bit;
  f.db = &h->db;
  for (int64_t i = 0; i < n; ++i) {
    const uint8_t* p = nullptr; size_t len = 0, vol = 0; int64_t local = 0;
    const int rc = header_bytes(h, first_seqno + i, &p, &len, &vol, &local);
    if (rc != SWA_OK) return rc;
    bool ok = h->db.in_mask(vol, local);
    if (ok && h->db.have_taxids) {
      long count = 0;
      render_deflines(p, len, f, &count);
      ok = count > 0;
    }
    include[i] = ok ? 1 : 0;
  }
  return SWA_OK;
} SWA_CATCH

int swa::read_blast_deflines(const char* basename, int symtype, const std::vector<int64_t>& seqnos,
                             std::vector<std::string>& deflines, std::vector<int64_t>& lengths)
{
  swa_headers* h = nullptr;
  int rc = swa_headers_open(basename, symtype, nullptr, &h);
  if (rc != SWA_OK || !h) return rc;
  deflines.clear();
  lengths.clear();
  DeflineFilter f;
  f.memb = (unsigned long)h->db.memb_bit;
  f.db = &h->db;
  const bool protein = h->db.protein;
  for (int64_t s : seqnos) {
    const uint8_t* p = nullptr; size_t n = 0, vol = 0; int64_t local = 0;
    rc = header_bytes(h, s, &p, &n, &vol, &local);
    if (rc != SWA_OK) break;
    deflines.push_back(render_deflines(p, n, f, nullptr));
    int64_t len;
    rc = sequence_length(h->db.vols[vol], protein, local, &len);
    if (rc != SWA_OK) break;
    lengths.push_back(len);
  }
  swa_headers_close(h);
  return rc;
}

// ---- BLAST v4 volume writer ------------------------------------------------------------------------------------
// The image has no makeblastdb / formatdb; the benchmark's CPU baseline and cold-open figures need
// the synthetic database as files the REFERENCE opens (database.cc:566-601, 1082-1131).  One streaming pass, O(1)
// extra memory, so a 10 M-sequence volume costs seconds - the Python writer (swipe_amd/blastdb.py) builds index
// arrays of 8 bytes per residue and is kept for the small fixtures.  Headers are the smallest Blast-def-line-set
// parse_blast_def_line accepts (asnparse.cc:753): they render as "lcl|s<N> seq<N>".
// Nucleotide volumes: 2 bits per base, last byte = remainder count (database.cc:1260-1261); ambiguity codes are not
// written by this fast path (SWA_EINVAL) - synthetic databases hold A, C, G, T only.
namespace {
inline void put_be32(std::vector<uint8_t>& v, uint32_t x)
{
  v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}
inline void ber_string(std::vector<uint8_t>& v, const std::string& s)
{
  v.push_back(0x1a);
  if (s.size() < 128) v.push_back(uint8_t(s.size()));
  else { v.push_back(0x82); v.push_back(uint8_t(s.size() >> 8)); v.push_back(uint8_t(s.size())); }
  v.insert(v.end(), s.begin(), s.end());
}
}  // namespace

extern "C" int swa_blastdb_write(const char* basename, int symtype, const uint8_t* residues, const int64_t* offsets,
                                 int64_t nseq, int64_t first_id, const char* title)
try {
  if (!basename || !offsets || nseq < 0 || (nseq > 0 && !residues)) return swa::fail(SWA_EINVAL, "bad argument");
  if (symtype != SWA_SYMTYPE_PROTEIN && symtype != SWA_SYMTYPE_NUCLEOTIDE) return swa::fail(SWA_EINVAL, "symtype must be 0 or 1");
  const bool protein = symtype == SWA_SYMTYPE_PROTEIN;
  const std::string base(basename);
  const char* ext = protein ? "p" : "n";
  FILE* fsq = std::fopen((base + "." + ext + "sq").c_str(), "wb");
  FILE* fhr = std::fopen((base + "." + ext + "hr").c_str(), "wb");
  if (!fsq || !fhr) { if (fsq) std::fclose(fsq); if (fhr) std::fclose(fhr); return swa::fail(SWA_EIO, "cannot create " + base); }
  std::vector<uint32_t> hdr_off(size_t(nseq) + 1), seq_off(size_t(nseq) + 1), amb_off;
  if (!protein) amb_off.resize(size_t(nseq) + 1);
  std::vector<uint8_t> buf, hb;
  buf.reserve(1 << 22);
  uint64_t sq_pos = 1, hr_pos = 0, total = 0, longest = 0;
  std::fputc(0, fsq);
  int rc = SWA_OK;
  for (int64_t s = 0; s < nseq && rc == SWA_OK; ++s) {
    const int64_t len = offsets[s + 1] - offsets[s];
    const uint8_t* p = residues + offsets[s];
    if (len < 0) { rc = swa::fail(SWA_EINVAL, "sequence offsets must be non-decreasing"); break; }
    total += uint64_t(len);
    longest = std::max<uint64_t>(longest, uint64_t(len));
    seq_off[size_t(s)] = uint32_t(sq_pos);
    buf.clear();
    if (protein) {
      buf.insert(buf.end(), p, p + len);
      buf.push_back(0);
    } else {
      static const int8_t two[16] = {-1, 0, 1, -1, 2, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1, -1};
      const int64_t full = len / 4;
      for (int64_t k = 0; k < full; ++k) {
        const int a = two[p[4 * k] & 15], b = two[p[4 * k + 1] & 15], c = two[p[4 * k + 2] & 15], d = two[p[4 * k + 3] & 15];
        if ((a | b | c | d) < 0) { rc = swa::fail(SWA_EINVAL, "swa_blastdb_write: ambiguity codes are not supported"); break; }
        buf.push_back(uint8_t(a << 6 | b << 4 | c << 2 | d));
      }
      uint8_t last = uint8_t(len - 4 * full);
      for (int64_t k = 4 * full; k < len && rc == SWA_OK; ++k) {
        const int a = two[p[k] & 15];
        if (a < 0) { rc = swa::fail(SWA_EINVAL, "swa_blastdb_write: ambiguity codes are not supported"); break; }
        last |= uint8_t(a << (6 - 2 * (k - 4 * full)));
      }
      buf.push_back(last);
      amb_off[size_t(s)] = uint32_t(sq_pos + buf.size());
    }
    if (sq_pos + buf.size() >= (uint64_t(1) << 32)) { rc = swa::fail(SWA_EINVAL, "volume exceeds the 4 GiB offset limit of the v4 format"); break; }
    if (std::fwrite(buf.data(), 1, buf.size(), fsq) != buf.size()) { rc = swa::fail(SWA_EIO, "write failed"); break; }
    sq_pos += buf.size();
    // 30 80 | 30 80 | A0 80 1A n title 00 00 | A1 80 30 80 A0 80 A1 80 1A n id 00 00 x4 | 00 00 | 00 00
    hdr_off[size_t(s)] = uint32_t(hr_pos);
    hb.clear();
    const std::string id = "s" + std::to_string(first_id + s), ti = "seq" + std::to_string(first_id + s);
    const uint8_t open1[] = {0x30, 0x80, 0x30, 0x80, 0xa0, 0x80};
    hb.insert(hb.end(), open1, open1 + 6);
    ber_string(hb, ti);
    const uint8_t mid[] = {0x00, 0x00, 0xa1, 0x80, 0x30, 0x80, 0xa0, 0x80, 0xa1, 0x80};
    hb.insert(hb.end(), mid, mid + 10);
    ber_string(hb, id);
    for (int k = 0; k < 12; ++k) hb.push_back(0);
    if (std::fwrite(hb.data(), 1, hb.size(), fhr) != hb.size()) { rc = swa::fail(SWA_EIO, "write failed"); break; }
    hr_pos += hb.size();
    if (hr_pos >= (uint64_t(1) << 32)) { rc = swa::fail(SWA_EINVAL, "header file exceeds 4 GiB"); break; }
  }
  std::fclose(fsq);
  std::fclose(fhr);
  if (rc != SWA_OK) return rc;
  seq_off[size_t(nseq)] = uint32_t(sq_pos);
  hdr_off[size_t(nseq)] = uint32_t(hr_pos);
  if (!protein) amb_off[size_t(nseq)] = uint32_t(sq_pos);
  std::vector<uint8_t> pin;
  const std::string t = title ? title : "swipe_amd synthetic", d = "Jan 1, 2026  0:00 AM";
  put_be32(pin, 4);
  put_be32(pin, protein ? 1 : 0);
  put_be32(pin, uint32_t(t.size())); pin.insert(pin.end(), t.begin(), t.end());
  put_be32(pin, uint32_t(d.size())); pin.insert(pin.end(), d.begin(), d.end());
  while (pin.size() % 4) pin.push_back(0);
  put_be32(pin, uint32_t(nseq));
  for (int k = 0; k < 8; ++k) pin.push_back(uint8_t(total >> (8 * k)));        // little-endian (database.cc:595)
  put_be32(pin, uint32_t(longest));
  for (uint32_t v : hdr_off) put_be32(pin, v);
  for (uint32_t v : seq_off) put_be32(pin, v);
  if (!protein) for (uint32_t v : amb_off) put_be32(pin, v);
  FILE* fin = std::fopen((base + "." + ext + "in").c_str(), "wb");
  if (!fin) return swa::fail(SWA_EIO, "cannot create " + base);
  const bool ok = std::fwrite(pin.data(), 1, pin.size(), fin) == pin.size();
  std::fclose(fin);
  return ok ? SWA_OK : swa::fail(SWA_EIO, "write failed");
} SWA_CATCH
