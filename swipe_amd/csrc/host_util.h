// Small host-side helpers shared by the library's translation units.
#ifndef SWA_HOST_UTIL_H
#define SWA_HOST_UTIL_H
#include <cstdint>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

namespace swa {
// records the message returned by swa_last_error() on this thread and returns `code`
int fail(int code, const std::string& msg);

// No exception leaves the C ABI: every entry point that returns a status is a function-try-block closed by this.  The
// reference ends the process when an allocation fails (xmalloc -> fatal, swipe.cc:158-182); a library hands the
// decision to its caller.
#define SWA_CATCH                                                                                               \
  catch (const std::bad_alloc&) { return swa::fail(SWA_ENOMEM, "out of host memory"); }                         \
  catch (const std::length_error& e) { return swa::fail(SWA_ENOMEM, std::string("size out of range: ") + e.what()); } \
  catch (const std::exception& e) { return swa::fail(SWA_EINVAL, std::string("unexpected: ") + e.what()); }

// A database (or a range of one) read from BLAST v4 files into host memory:
// sequence s = residues[offsets[s] .. offsets[s+1]) in reference symbol codes.
// byte buffer that is NOT zero-filled on allocation (a 3 GB std::vector would spend 0.3 s on that)
struct RawBytes {
  std::unique_ptr<uint8_t[]> p;
  size_t n = 0;
  void resize(size_t bytes) { p.reset(new uint8_t[bytes ? bytes : 1]); n = bytes; }
  void clear() { p.reset(); n = 0; }
  uint8_t* data() { return p.get(); }
  const uint8_t* data() const { return p.get(); }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
};

struct HostDb {
  RawBytes residues;
  std::vector<int64_t> offsets;       // nseq + 1
  int64_t first_seqno = 0;
  int64_t total_seqcount = 0, total_symcount = 0, longest = 0;
  std::string title;
  // OID mask of a masked alias (database.cc:687-706): included[s] per loaded sequence, and the alias's own
  // NSEQ / LENGTH, which the reference uses for statistics instead of the volume totals (hits.cc:333-342)
  bool masked = false;
  std::vector<uint8_t> included;
  int64_t masked_seqcount = 0, masked_symcount = 0;
};
// Mirrors db_open (alias + volumes, database.cc:775-925) and db_getsequence
// (database.cc:1237-1401) for symtype 0 and 1.  Returns SWA_OK or records an error.
int read_blast_db(const char* basename, int symtype, int64_t first_seqno, int64_t last_seqno, HostDb& out);
// The same range, opened for reading piece by piece (blastdb.cpp): open() reads the index (lengths as prefix sums, the OID
// mask), fill() delivers the residues of any run of sequences into the caller's buffer.
struct RangeReader {
  RangeReader();
  ~RangeReader();
  RangeReader(const RangeReader&) = delete;
  RangeReader& operator=(const RangeReader&) = delete;
  int open(const char* basename, int symtype, int64_t first_seqno, int64_t last_seqno);
  int fill(int64_t from, int64_t to, uint8_t* dst, bool nibbles, unsigned* or_codes) const;
  void forget(int64_t from, int64_t to) const;
  std::vector<int64_t> offsets;       // nseq + 1 prefix sums of the lengths
  int64_t first_seqno = 0, total_seqcount = 0, total_symcount = 0, longest = 0;
  std::string title;
  bool protein = true, masked = false;
  std::vector<uint8_t> included;
  int64_t masked_seqcount = 0, masked_symcount = 0;
 private:
  struct Impl;
  Impl* impl;
};
// What a pipelined open needs before the first residue is read (db_open + the index half of db_mapsequences,
// database.cc:775-925, 1082-1131): the lengths of the sequences [first_seqno, last_seqno] out of the index files, and the
// byte ranges of the sequence files that hold them.  `regular` says the loader may take the files as they lie: protein
// volumes whose entries are [residues NUL] back to back, nucleotide volumes whose entries are [packed bases | ambiguity data]
// back to back (round 5), with or without an OID mask.  An index whose entries overlap or run backwards is left to
// read_blast_db, which reports it.
struct LoadPiece {
  std::string path;                   // a .psq file
  int64_t file_begin = 0, file_end = 0;   // its bytes [file_begin, file_end): whole entries, each [residues NUL]
  int64_t first = 0;                  // range-relative index of the first sequence in the piece
};
struct LoadPlan {
  bool regular = false;
  std::vector<int64_t> offsets;       // nseq + 1 prefix sums of the lengths (offsets[0] = 0)
  std::vector<LoadPiece> pieces;      // in sequence order: concatenated they are the range as [residues NUL]* (protein)
  int64_t first_seqno = 0, total_seqcount = 0, total_symcount = 0, longest = 0;
  // nucleotide volumes: the pieces are .nsq ranges, whole entries [packed bases | ambiguity data]; raw_seq[s] = where entry s
  // starts in the concatenation of the pieces (nseq + 1 values), longest_entry = bytes of the largest entry
  bool nucleotide = false;
  std::vector<int64_t> raw_seq;
  int64_t longest_entry = 0;
  // OID mask of a masked alias (database.cc:687-706): included[s] per sequence of the range, and the alias's own totals
  bool masked = false;
  std::vector<uint8_t> included;
  int64_t masked_seqcount = 0, masked_symcount = 0;
};
int plan_blast_load(const char* basename, int symtype, int64_t first_seqno, int64_t last_seqno, LoadPlan& out);
// prefix sums of the sequence lengths of the whole database, from the index files alone (nseq + 1 entries)
int read_blast_lengths(const char* basename, int symtype, std::vector<int64_t>& offsets);
int read_blast_totals(const char* basename, int symtype, int64_t* nseq, int64_t* nsym);
// Definition lines ("lcl|id title" style, first defline of each entry) of the given sequences
int read_blast_deflines(const char* basename, int symtype, const std::vector<int64_t>& seqnos,
                        std::vector<std::string>& deflines, std::vector<int64_t>& lengths);
}  // namespace swa
#endif
