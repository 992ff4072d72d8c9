// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the search kernels use.
// MI355X_MICROARCH.md vouches for one case only: FETCH_SIZE reports half the bytes of a wide (16 B per lane) coalesced
// streaming read.  The first-pass kernels read their residue stream 2 B per lane (one 128-byte line per wave and load);
// scores leave as scattered 4-byte stores.  Every kernel here moves a KNOWN number of bytes (1 GiB, far beyond the
// 256 MiB Infinity Cache) so that counter x factor = bytes can be solved for the factor:
//   read16   16 B per lane, coalesced          (the guide's case)
//   read4     4 B per lane, coalesced
//   read2     2 B per lane, coalesced: 128 B per wave-load, the stream access of swa_narrow_*_kernel
//   write4c   4 B per lane, coalesced
//   write4s   4 B per lane, one store per 64-byte sector (scattered scores)
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- ./fetch_calib.bin   (and once more with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

constexpr size_t BYTES = size_t(1) << 30;
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <typename T>
__global__ void __launch_bounds__(256) calib_read(const T* __restrict__ p, size_t n, unsigned* out)
{
  unsigned acc = 0;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const T v = p[i];
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
    for (unsigned k = 0; k < sizeof(T); k += 2) acc += b[k];
  }
  if (acc == 0x12345u) out[0] = acc;
}
__global__ void __launch_bounds__(256) calib_write4c(unsigned* p, size_t n)
{
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = unsigned(i);
}
__global__ void __launch_bounds__(256) calib_write4s(unsigned* p, size_t nstores)       // one 4-byte store per 64-byte sector
{
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nstores; i += size_t(gridDim.x) * blockDim.x) p[i * 16] = unsigned(i);
}

int main()
{
  CHECK(hipSetDevice(0));
  void* buf;
  unsigned* out;
  CHECK(hipMalloc(&buf, BYTES));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 1, BYTES));
  CHECK(hipDeviceSynchronize());
  const int blocks = 256 * 16;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(calib_read<u4v>, dim3(blocks), dim3(256), 0, 0, (const u4v*)buf, BYTES / 16, out);
    hipLaunchKernelGGL(calib_read<unsigned>, dim3(blocks), dim3(256), 0, 0, (const unsigned*)buf, BYTES / 4, out);
    hipLaunchKernelGGL(calib_read<unsigned short>, dim3(blocks), dim3(256), 0, 0, (const unsigned short*)buf, BYTES / 2, out);
    hipLaunchKernelGGL(calib_write4c, dim3(blocks), dim3(256), 0, 0, (unsigned*)buf, BYTES / 4);
    hipLaunchKernelGGL(calib_write4s, dim3(blocks), dim3(256), 0, 0, (unsigned*)buf, BYTES / 64);
    CHECK(hipDeviceSynchronize());
  }
  printf("bytes per kernel: read16 / read4 / read2 / write4c = %zu; write4s = %zu stores of 4 B (%zu B), one per 64-byte sector of %zu B\n",
         BYTES, BYTES / 64, BYTES / 16, BYTES);
  return 0;
}
