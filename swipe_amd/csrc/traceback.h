// Host half of the alignment phase (see traceback.cpp).
#ifndef SWA_TRACEBACK_H
#define SWA_TRACEBACK_H
#include <cstdint>
#include <vector>

namespace swa {
struct EditOp { char kind; int64_t count; };   // 'M' column pair, 'D' query symbol vs gap, 'I' database symbol vs gap

// forward sweep of region() (align.cc:70-106): score and the first cell (query-row major) that reaches it
int64_t forward_end(const uint8_t* query, int64_t qlen, const uint8_t* dseq, int64_t dlen, const int32_t* matrix,
                    int64_t gapopen, int64_t gapextend, int64_t* q_end, int64_t* d_end);
// backward sweep of region() (align.cc:111-154); false = the reference's "Internal error in align function."
bool backward_start(const uint8_t* query, const uint8_t* dseq, const int32_t* matrix, int64_t gapopen,
                    int64_t gapextend, int64_t score, int64_t q_end, int64_t d_end, int64_t* q_start, int64_t* d_start);
// diff() (align.cc:236-467) between the two cells, run-length encoded
void edit_script(const uint8_t* query, const uint8_t* dseq, const int32_t* matrix, int64_t gapopen, int64_t gapextend,
                 int64_t q_start, int64_t d_start, int64_t q_end, int64_t d_end, std::vector<EditOp>& ops);
// count_align (hits.cc:1021-1109)
void count_columns(const uint8_t* query, const uint8_t* dseq, const int32_t* matrix, int64_t q_start, int64_t d_start,
                   const std::vector<EditOp>& ops, int64_t* identities, int64_t* positives, int64_t* indels,
                   int64_t* aligned, int64_t* gaps);
}  // namespace swa
#endif
