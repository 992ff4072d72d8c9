// Which build of the one-query first pass a search runs (round 3: one table instead of nested conditions).
#ifndef SWA_KERNEL_CHOICE_H
#define SWA_KERNEL_CHOICE_H
#include <cstdint>

namespace swa {
// what the choice may depend on: the query, the scoring system's reach, the shard, the caller's knobs
struct ChoiceEnv {
  int64_t qlen = 0;
  bool want_bound = false;       // a top-K search whose threshold clears the bound build's slack (bound_wanted)
  int64_t hi = 11, goe = 12, ge = 1;   // highest matrix entry, gap open + extend, gap extend
  int64_t longest = 35000;       // longest sequence of the shard (reach of a score)
  double mean_len = 325;         // mean sequence length of the shard: the skew of a chain is drained once per batch
  int lanes = 0;                 // option "lanes": chain length to use if the query fits it (tests, A/B); 0 = free
  bool long_lanes = true;        // option "long_lanes": bound builds of more than 48 rows per lane
  int bound_period = 16;
};
struct KernelPick {
  int G = 0, K = 0;              // lanes per sequence pair, rows per lane; G = 0: no single-pass build takes this query
  bool bound = false;            // the bound build of that shape
  int predicted_gcups = 0;       // the table's figure for it, padding rows and skew counted
};
// limits of the builds that exist (sw_kernels.hip, sw_one_*.hip, sw_cb_*.hip)
constexpr int kExactRows = 48, kExactRows16 = 58, kBoundLongRows = 62, kBoundOneRows = 60, kBoundRows16 = 58;
int64_t f16_exact_limit(int64_t hi, int64_t ge, int K);                 // scores below it are exact in packed f16 with K rows per lane
bool chains_isolated(int64_t qlen, int64_t longest, int64_t hi, int64_t goe, int64_t ge);   // see short_chains_safe
bool build_exists(bool bound, int G, int K);
KernelPick pick_first_pass(const ChoiceEnv& e);
// the same rule for the two-query kernels (both strands of a nucleotide query, two protein queries or frames): nres = 16 for
// the nucleotide alphabet, 32 otherwise; kmax > 0 caps the rows per lane (option "dual_kmax")
bool dual_build_exists(bool bound, int nres, int G, int K);
KernelPick pick_dual(const ChoiceEnv& e, int nres, int kmax);
}  // namespace swa
#endif
