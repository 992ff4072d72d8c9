// BLAST database version 4 reader: the wire format the loader ingests.
//
// Follows what the reference accepts: alias files (.pal/.nal: TITLE / DBLIST, one level of
// nesting, database.cc:406-489 and 775-925), index files (.pin/.nin, database.cc:566-601:
// everything big-endian except the little-endian residue total at 595) and sequence files
// (.psq: NCBIstdaa bytes, NUL-terminated entries; .nsq: 2 bits per base, remainder count in
// the last byte, optional ambiguity runs, database.cc:1237-1323).  OID masks, taxid filters
// and translated (6-frame) access are outside this path.
#include "../../include/swipe_amd.h"
#include "host_util.h"

#include <cstdio>
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
  std::string title;
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
  const uint32_t dl = be32(p + o); o += 4 + dl;
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

// DBLIST entries of an alias file, or empty if there is no alias
std::vector<std::string> read_alias(const std::string& base, bool protein, std::string* title)
{
  std::vector<std::string> names;
  FILE* f = std::fopen((base + (protein ? ".pal" : ".nal")).c_str(), "r");
  if (!f) return names;
  char line[4096];
  while (std::fgets(line, sizeof line, f)) {
    if (std::strncmp(line, "TITLE", 5) == 0 && title) {
      std::string t(line + 5);
      while (!t.empty() && (t.back() == '\n' || t.back() == '\r')) t.pop_back();
      size_t b = t.find_first_not_of(" \t");
      *title = b == std::string::npos ? "" : t.substr(b);
    } else if (std::strncmp(line, "DBLIST", 6) == 0) {
      char* tok = std::strtok(line + 6, " \t\r\n");
      while (tok) { names.emplace_back(tok); tok = std::strtok(nullptr, " \t\r\n"); }
    }
  }
  std::fclose(f);
  return names;
}

std::string dir_of(const std::string& base)
{
  const size_t s = base.rfind('/');
  return s == std::string::npos ? std::string() : base.substr(0, s + 1);
}
}  // namespace

int swa::read_blast_db(const char* basename, int symtype, int64_t first_seqno, int64_t last_seqno, HostDb& out)
{
  if (!basename) return fail(SWA_EINVAL, "null database name");
  if (symtype != SWA_SYMTYPE_PROTEIN && symtype != SWA_SYMTYPE_NUCLEOTIDE)
    return fail(SWA_EINVAL, "only symtype 0 (nucleotide) and 1 (protein) databases are supported");
  const bool protein = symtype == SWA_SYMTYPE_PROTEIN;
  const std::string base(basename);
  std::vector<std::string> vols;
  std::string title;
  const std::vector<std::string> top = read_alias(base, protein, &title);
  if (top.empty()) {
    vols.push_back(base);
  } else {
    const std::string dir = dir_of(base);
    for (const std::string& n : top) {
      const std::vector<std::string> nested = read_alias(dir + n, protein, nullptr);
      if (nested.empty()) vols.push_back(dir + n);
      else for (const std::string& m : nested) vols.push_back(dir + m);
    }
  }
  if (vols.size() > 256) return fail(SWA_EIO, "too many database volumes");    // database.cc:216
  std::vector<Volume> V(vols.size());
  std::string err;
  int64_t nseq = 0, nsym = 0, longest = 0;
  for (size_t i = 0; i < vols.size(); ++i) {
    if (!open_volume(vols[i], protein, V[i], err)) return fail(SWA_EIO, err);
    nseq += V[i].nseq;
    nsym += V[i].nsym;
    if (V[i].longest > longest) longest = V[i].longest;
  }
  out.total_seqcount = nseq;
  out.total_symcount = nsym;
  out.longest = longest;
  out.title = title.empty() ? V[0].title : title;
  if (first_seqno < 0) first_seqno = 0;
  if (last_seqno < 0 || last_seqno >= nseq) last_seqno = nseq - 1;
  out.first_seqno = first_seqno;
  out.offsets.assign(1, 0);
  out.residues.clear();
  if (last_seqno < first_seqno) return SWA_OK;
  out.offsets.reserve(size_t(last_seqno - first_seqno + 2));

  int64_t vbase = 0;
  for (const Volume& v : V) {
    const int64_t lo = first_seqno > vbase ? first_seqno - vbase : 0;
    const int64_t hi = last_seqno - vbase < v.nseq - 1 ? last_seqno - vbase : v.nseq - 1;
    for (int64_t s = lo; s <= hi; ++s) {
      const uint64_t o1 = be32(v.seq_off + 4 * s), o2 = be32(v.seq_off + 4 * (s + 1));
      if (o2 < o1 || o2 > v.seq.n) return fail(SWA_EIO, "corrupt sequence offsets in " + v.base);
      if (protein) {
        const size_t len = o2 > o1 ? size_t(o2 - o1 - 1) : 0;        // entry includes its NUL terminator
        out.residues.insert(out.residues.end(), v.seq.p + o1, v.seq.p + o1 + len);
      } else {
        const uint64_t o3 = be32(v.amb_off + 4 * s);
        if (o3 <= o1 || o3 > o2) return fail(SWA_EIO, "corrupt ambiguity offsets in " + v.base);
        const uint8_t* body = v.seq.p + o1;
        const size_t packed = size_t(o3 - o1);
        const size_t ntlen = 4 * (packed - 1) + (body[packed - 1] & 3);   // database.cc:1260-1261
        const size_t at = out.residues.size();
        out.residues.resize(at + ntlen);
        uint8_t* dst = out.residues.data() + at;
        for (size_t i = 0; i < ntlen; ++i)
          dst[i] = uint8_t(1u << ((body[i >> 2] >> ((3 - (i & 3)) << 1)) & 3));  // A=1 C=2 G=4 T=8
        if (o2 > o3) {                                               // ambiguity runs, database.cc:1284-1323
          const uint8_t* a = v.seq.p + o3;
          const size_t bytes = size_t(o2 - o3);
          if (bytes >= 4) {
            const uint32_t hdr = be32(a);
            if (hdr >> 31) {
              for (size_t k = 0; k + 8 <= bytes - 4; k += 8) {
                const uint64_t e = be64(a + 4 + k);
                const uint64_t code = e >> 60, run = ((e >> 48) & 0xfff) + 1, off = e & 0x0000fffffffffffULL;
                for (uint64_t r = 0; r < run && off + r < ntlen; ++r) dst[off + r] = uint8_t(code);
              }
            } else {
              for (size_t k = 0; k + 4 <= bytes - 4; k += 4) {
                const uint32_t e = be32(a + 4 + k);
                const uint32_t code = e >> 28, run = ((e >> 24) & 0xf) + 1, off = e & 0x00ffffff;
                for (uint32_t r = 0; r < run && size_t(off) + r < ntlen; ++r) dst[off + r] = uint8_t(code);
              }
            }
          }
        }
      }
      out.offsets.push_back(int64_t(out.residues.size()));
    }
    vbase += v.nseq;
  }
  return SWA_OK;
}

// ---- definition lines -----------------------------------------------------------------------
// Minimal BER walker for Blast-def-line-set (reference asnparse.cc:652-887): renders the first
// def-line of an entry as "<seqids joined by |> <title>".  Seq-id choices follow the tag table at
// asnparse.cc:657-658.
namespace {
struct Ber {
  const uint8_t* p;
  const uint8_t* end;
  bool eoc() const { return p + 1 < end && p[0] == 0 && p[1] == 0; }
  // reads tag + length; returns content length or -1 for the indefinite form
  bool head(int& tag, long& len)
  {
    if (p >= end) return false;
    tag = *p++;
    if (p >= end) return false;
    int l = *p++;
    if (l == 0x80) { len = -1; return true; }
    if (l & 0x80) {
      int n = l & 0x7f;
      len = 0;
      while (n-- && p < end) len = (len << 8) | *p++;
    } else {
      len = l;
    }
    return true;
  }
  // skips one complete element
  void skip()
  {
    int tag; long len;
    if (!head(tag, len)) { p = end; return; }
    if (len >= 0) { p += len; return; }
    while (p < end && !eoc()) skip();
    p += 2;
  }
  std::string str(long len) { std::string s(reinterpret_cast<const char*>(p), size_t(len)); p += len; return s; }
  long integer(long len) { long v = 0; for (long i = 0; i < len && p < end; ++i) v = (v << 8) | *p++; return v; }
  void close(long len) { if (len < 0 && eoc()) p += 2; }
};

std::string object_id(Ber& b)                       // CHOICE { id [0] INTEGER, str [1] VisibleString }
{
  int tag; long len;
  std::string out;
  if (!b.head(tag, len)) return out;
  int t2; long l2;
  if (b.head(t2, l2)) out = (tag == 0xA0) ? std::to_string(b.integer(l2)) : b.str(l2);
  b.close(len);
  return out;
}

std::string seq_id(Ber& b)
{
  static const char* const names[] = {"lcl", "bbs", "bbm", "gim", "gb", "emb", "pir", "sp", "pat", "ref",
                                      "gnl", "gi", "dbj", "prf", "pdb", "tpg", "tpe", "tpd", "gpp", "nat"};
  int tag; long len;
  if (!b.head(tag, len)) return std::string();
  const int choice = tag - 0xA0;
  const std::string db = (choice >= 0 && choice < 20) ? names[choice] : "unk";
  std::string out;
  const uint8_t* stop = len >= 0 ? b.p + len : nullptr;
  if (choice == 0) {
    out = db + "|" + object_id(b);
  } else if (choice == 11 || choice == 1 || choice == 2 || choice == 3) {
    int t; long l;
    if (b.head(t, l)) out = db + "|" + std::to_string(b.integer(l));
  } else if (choice == 10) {                        // Dbtag { db VisibleString, tag Object-id }
    int t; long l;
    if (b.head(t, l)) {                             // SEQUENCE
      int t1; long l1, l1b; int t1b;
      std::string dbname, tagv;
      if (b.head(t1, l1)) { if (b.head(t1b, l1b)) dbname = b.str(l1b); b.close(l1); }
      if (b.head(t1, l1)) { tagv = object_id(b); b.close(l1); }
      b.close(l);
      out = db + "|" + dbname + "|" + tagv;
    }
  } else {                                          // Textseq-id { name, accession, release, version }
    int t; long l;
    std::string name, acc;
    long version = 0;
    if (b.head(t, l)) {
      const uint8_t* sstop = l >= 0 ? b.p + l : nullptr;
      while (b.p < b.end && (sstop ? b.p < sstop : !b.eoc())) {
        int ft; long fl;
        if (!b.head(ft, fl)) break;
        int it; long il;
        if (!b.head(it, il)) break;
        if (ft == 0xA0) name = b.str(il);
        else if (ft == 0xA1) acc = b.str(il);
        else if (ft == 0xA3) version = b.integer(il);
        else b.p += il;
        b.close(fl);
      }
      b.close(l);
    }
    out = db + "|" + acc + (version ? "." + std::to_string(version) : std::string()) + "|" + name;
  }
  if (stop) b.p = stop; else { while (b.p < b.end && !b.eoc()) b.skip(); b.p += 2; }
  return out;
}

// every Blast-def-line of the set (one per identical sequence merged into this entry), each rendered
// "ids title" as parse_blast_def_line does (asnparse.cc:753-887), joined by '\n'
std::string all_deflines(const uint8_t* p, size_t n)
{
  Ber b{p, p + n};
  int tag; long len;
  std::string out;
  if (!b.head(tag, len) || tag != 0x30) return out;                   // Blast-def-line-set
  const uint8_t* set_stop = len >= 0 ? b.p + len : nullptr;
  bool first = true;
  while (b.p < b.end && (set_stop ? b.p < set_stop : !b.eoc())) {
    int t2; long l2;
    if (!b.head(t2, l2) || t2 != 0x30) break;                          // one Blast-def-line
    std::string title, ids;
    const uint8_t* stop = l2 >= 0 ? b.p + l2 : nullptr;
    while (b.p < b.end && (stop ? b.p < stop : !b.eoc())) {
      int ft; long fl;
      if (!b.head(ft, fl)) break;
      if (ft == 0xA0) {                                                // title
        int it; long il;
        if (b.head(it, il)) title = b.str(il);
        b.close(fl);
      } else if (ft == 0xA1) {                                         // seqid SEQUENCE OF Seq-id
        int st; long sl;
        if (b.head(st, sl)) {
          const uint8_t* sstop = sl >= 0 ? b.p + sl : nullptr;
          while (b.p < b.end && (sstop ? b.p < sstop : !b.eoc())) {
            const std::string id = seq_id(b);
            if (!ids.empty()) ids += "|";
            ids += id;
          }
          b.close(sl);
        }
        b.close(fl);
      } else {
        if (fl >= 0) b.p += fl; else { while (b.p < b.end && !b.eoc()) b.skip(); b.p += 2; }
      }
    }
    b.close(l2);
    if (!first) out += '\n';
    first = false;
    out += ids + ((ids.empty() || title.empty()) ? "" : " ") + title;
  }
  return out;
}
}  // namespace

int swa::read_blast_deflines(const char* basename, int symtype, const std::vector<int64_t>& seqnos,
                             std::vector<std::string>& deflines, std::vector<int64_t>& lengths)
{
  const bool protein = symtype == SWA_SYMTYPE_PROTEIN;
  const std::string base(basename);
  std::vector<std::string> vols;
  const std::vector<std::string> top = read_alias(base, protein, nullptr);
  if (top.empty()) vols.push_back(base);
  else {
    const std::string dir = dir_of(base);
    for (const std::string& n : top) {
      const std::vector<std::string> nested = read_alias(dir + n, protein, nullptr);
      if (nested.empty()) vols.push_back(dir + n);
      else for (const std::string& m : nested) vols.push_back(dir + m);
    }
  }
  std::vector<Volume> V(vols.size());
  std::string err;
  for (size_t i = 0; i < vols.size(); ++i) {
    if (!open_volume(vols[i], protein, V[i], err)) return fail(SWA_EIO, err);
    if (!V[i].hdr.open(vols[i] + (protein ? ".phr" : ".nhr"))) return fail(SWA_EIO, "Unable to open file " + vols[i] + (protein ? ".phr." : ".nhr."));
  }
  deflines.clear();
  lengths.clear();
  for (int64_t s : seqnos) {
    int64_t local = s;
    const Volume* v = nullptr;
    for (const Volume& x : V) { if (local < x.nseq) { v = &x; break; } local -= x.nseq; }
    if (!v || local < 0) return fail(SWA_EINVAL, "Cant find database volume.");
    const uint64_t h1 = be32(v->hdr_off + 4 * local), h2 = be32(v->hdr_off + 4 * (local + 1));
    if (h2 < h1 || h2 > v->hdr.n) return fail(SWA_EIO, "corrupt header offsets in " + v->base);
    deflines.push_back(all_deflines(v->hdr.p + h1, size_t(h2 - h1)));
    const uint64_t o1 = be32(v->seq_off + 4 * local), o2 = be32(v->seq_off + 4 * (local + 1));
    if (protein) lengths.push_back(o2 > o1 ? int64_t(o2 - o1 - 1) : 0);
    else {
      const uint64_t o3 = be32(v->amb_off + 4 * local);
      const size_t packed = size_t(o3 - o1);
      lengths.push_back(int64_t(4 * (packed - 1) + (v->seq.p[o1 + packed - 1] & 3)));
    }
  }
  return SWA_OK;
}
