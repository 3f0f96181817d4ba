// Microbenchmark: what a plain streaming pass achieves on this box (calibrates the "HBM-bound" claims for the NTT passes):
// out-of-place copy, in-place read-modify-write, read-only and write-only, 16 B per lane, 268 MB working set per direction
// (the size of one pass over 512 limbs at N = 2^16).
//   hipcc --offload-arch=gfx950 -O3 -o copy copy.hip && ./copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int MODE>  // 0 copy, 1 in place, 2 read only, 3 write only
__global__ __launch_bounds__(256) void k(ulonglong2 *dst, const ulonglong2 *src, size_t n, int per) {
  size_t i = (size_t)blockIdx.x * 256 * per + threadIdx.x;
  ulonglong2 acc = {0, 0};
  for (int r = 0; r < per; ++r, i += 256) {
    if (i >= n) break;
    if (MODE == 3) { dst[i] = ulonglong2{i, i}; continue; }
    ulonglong2 v = src[i];
    if (MODE == 2) { acc.x ^= v.x; acc.y += v.y; continue; }
    v.x += 1; v.y ^= 5;
    dst[i] = v;
  }
  if (MODE == 2 && acc.x == 0x1234567 && acc.y == 7) dst[0] = acc;  // keeps the loads alive
}

// the same with 8 B per lane (the access width of the NTT passes' strided loads)
__global__ __launch_bounds__(256) void k8(unsigned long long *dst, const unsigned long long *src, size_t n, int per) {
  size_t i = (size_t)blockIdx.x * 256 * per + threadIdx.x;
  for (int r = 0; r < per; ++r, i += 256)
    if (i < n) dst[i] = src[i] + 1;
}
// 8 loads of 8 B in flight per lane, then 4 stores of 16 B (the row pass's memory shape without its arithmetic)
__global__ __launch_bounds__(256) void krow(unsigned long long *dst, const unsigned long long *src) {
  const size_t row = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5), t = threadIdx.x & 31;
  const unsigned long long *s = src + row * 256;
  unsigned long long e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = s[t + 32 * k];
#pragma unroll
  for (int k = 0; k < 8; k += 2) *reinterpret_cast<ulonglong2 *>(dst + row * 256 + 64 * (k / 2) + 2 * t) = ulonglong2{e[k] + 1, e[k + 1] ^ 3};
}

int main() {
  const size_t bytes = (size_t)512 << 19;  // 512 limbs x 512 KiB
  const size_t n = bytes / 16;
  ulonglong2 *a, *b;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char *names[] = {"copy", "in-place rmw", "read only", "write only"};
  for (int mode = 0; mode < 4; ++mode)
    for (int per : {1, 4, 16}) {
      const int grid = (int)((n + 256 * (size_t)per - 1) / (256 * (size_t)per));
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        if (mode == 0) k<0><<<grid, 256>>>(b, a, n, per);
        else if (mode == 1) k<1><<<grid, 256>>>(a, a, n, per);
        else if (mode == 2) k<2><<<grid, 256>>>(b, a, n, per);
        else k<3><<<grid, 256>>>(b, a, n, per);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
      }
      const double moved = (mode < 2 ? 2.0 : 1.0) * bytes;
      printf("%-14s per=%-2d grid=%-7d %.4f ms  %.2f TB/s\n", names[mode], per, grid, best, moved / (best * 1e-3) / 1e12);
    }
  for (int per : {1, 4}) {
    const size_t n8 = bytes / 8;
    const int grid = (int)((n8 + 256 * (size_t)per - 1) / (256 * (size_t)per));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CHECK(hipEventRecord(e0));
      k8<<<grid, 256>>>((unsigned long long *)a, (const unsigned long long *)a, n8, per);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("%-14s per=%-2d grid=%-7d %.4f ms  %.2f TB/s\n", "in-place 8 B", per, grid, best, 2.0 * bytes / (best * 1e-3) / 1e12);
  }
  {
    const int grid = (int)(bytes / 8 / 256 / 8);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CHECK(hipEventRecord(e0));
      krow<<<grid, 256>>>((unsigned long long *)a, (const unsigned long long *)a);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("%-14s grid=%-7d %.4f ms  %.2f TB/s\n", "row-pass shape", grid, best, 2.0 * bytes / (best * 1e-3) / 1e12);
  }
  return 0;
}
