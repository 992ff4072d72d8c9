// Multi-threaded C++ twin of swipe_amd/synth.py (same integer recipe, bit-identical output).
#include "../../include/swipe_amd_synth.h"

#include <algorithm>
#include <thread>
#include <vector>

namespace {
constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;
constexpr uint64_t MIX_SEQ = 0xD1B54A32D192ED03ull;
constexpr uint64_t PLANT_PERIOD = 8192;

inline uint64_t splitmix64(uint64_t x)
{
  uint64_t z = x + GOLDEN;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t seq_key(uint64_t seed, int64_t seqno) { return splitmix64(splitmix64(seed) ^ (uint64_t(seqno) * MIX_SEQ)); }

inline void random_residues(uint64_t key, uint64_t salt, int64_t n, const uint8_t* rtab, uint8_t* out)
{
  const uint64_t base = key + salt;
  for (int64_t i = 0; i < n; i += 5) {
    const uint64_t h = splitmix64(base + uint64_t(i / 5));
    const int64_t m = std::min<int64_t>(5, n - i);
    for (int64_t k = 0; k < m; ++k) out[i + k] = rtab[(h >> (12 * k)) & 4095];
  }
}

struct Plant { bool planted; int copies; int rate; int64_t left, right; };
inline Plant plant_of(uint64_t key, int64_t qlen)
{
  Plant p{false, 0, 0, 0, 0};
  if (qlen > 0 && key % PLANT_PERIOD == 0) {
    p.planted = true;
    const int kind = int((key >> 13) & 7);
    p.copies = kind == 0 ? 3 : 1;
    p.rate = kind == 0 ? 5 : kind * 20;
    p.left = int64_t((key >> 20) & 63);
    p.right = int64_t((key >> 28) & 63);
  }
  return p;
}
// one mutated copy of the query; returns its length, writes it if out != nullptr
inline int64_t mutated_copy(uint64_t key, int c, int rate, const uint8_t* q, int64_t qlen, const uint8_t* rtab, uint8_t* out)
{
  const uint64_t base = key + uint64_t(2 + c) * (1ull << 24);
  int64_t n = 0;
  for (int64_t i = 0; i < qlen; ++i) {
    const uint64_t h = splitmix64(base + uint64_t(i));
    const int64_t ev = int64_t((h >> 40) & 0x3FF);
    if (ev < 6) continue;                                           // deletion
    if (out) out[n] = (int64_t(h & 0xFF) < rate) ? rtab[(h >> 20) & 4095] : q[i];
    ++n;
    if (ev < 12) { if (out) out[n] = rtab[(h >> 50) & 4095]; ++n; } // insertion after the residue
  }
  return n;
}

int64_t length_of(uint64_t seed, int64_t seqno, const int32_t* ltab, const uint8_t* q, int64_t qlen)
{
  const uint64_t key = seq_key(seed, seqno);
  const Plant p = plant_of(key, q ? qlen : 0);
  if (!p.planted) return ltab[(key >> 40) & 4095];
  int64_t n = p.left + p.right;
  for (int c = 0; c < p.copies; ++c) n += mutated_copy(key, c, p.rate, q, qlen, nullptr, nullptr);
  return n;
}

template <typename F> void parallel_for(int64_t n, int threads, F f)
{
  if (threads < 1) threads = 1;
  if (threads == 1 || n < 4096) { f(0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t) th.emplace_back([=] { f(n * t / threads, n * (t + 1) / threads); });
  for (auto& x : th) x.join();
}
}  // namespace

extern "C" int64_t swa_synth_length(uint64_t seed, int64_t seqno, const int32_t* len_table, const uint8_t* query, int64_t qlen)
{
  return length_of(seed, seqno, len_table, query, qlen);
}

extern "C" int64_t swa_synth_offsets(uint64_t seed, int64_t first, int64_t nseq, const int32_t* len_table,
                                     const uint8_t* query, int64_t qlen, int64_t* offsets, int threads)
{
  parallel_for(nseq, threads, [=](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) offsets[i + 1] = length_of(seed, first + i, len_table, query, qlen);
  });
  offsets[0] = 0;
  for (int64_t i = 0; i < nseq; ++i) offsets[i + 1] += offsets[i];
  return offsets[nseq];
}

extern "C" int swa_synth_fill(uint64_t seed, int64_t first, int64_t nseq, const int32_t* len_table,
                              const uint8_t* res_table, const uint8_t* query, int64_t qlen,
                              const int64_t* offsets, uint8_t* residues, int threads)
{
  parallel_for(nseq, threads, [=](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const uint64_t key = seq_key(seed, first + i);
      uint8_t* out = residues + offsets[i];
      const Plant p = plant_of(key, query ? qlen : 0);
      if (!p.planted) {
        random_residues(key, 1, len_table[(key >> 40) & 4095], res_table, out);
        continue;
      }
      random_residues(key, 1, p.left, res_table, out);
      out += p.left;
      for (int c = 0; c < p.copies; ++c) out += mutated_copy(key, c, p.rate, query, qlen, res_table, out);
      random_residues(key, 1ull << 30, p.right, res_table, out);
    }
  });
  return 0;
}
