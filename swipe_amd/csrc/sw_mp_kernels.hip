// Second kernel translation unit: the multi-pass kernel (long queries, 32/64-bit re-queue) and the single-pass
// two-query kernel.  Kept apart from sw_kernels.hip only so that the two compile in parallel.
#include "sw_common.cuh"

#include "sw_mp_kernel.inc"
