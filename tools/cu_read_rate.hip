// How fast can ONE compute unit pull, and does that depend on how many others pull at the same time?
// (Round 3, DESIGN.md section 8: the two-step kernel's pull phase accepts one 256-byte pull per ~22 cycles and CU.)
//
// One 704-thread block per CU (150 KB of dynamic LDS keep a second one away, as in k_step2).  Every wave streams rows of 64
// fp32 cells from Q "population" arrays the way phase A does: per trip Q independent 4-byte loads per lane (one aligned
// 256-byte row per wave and population), DEPTH trips in flight, nothing else — no stores, no LDS traffic, no arithmetic beyond
// one add per value.  Blocks read disjoint slices, sized far beyond L2 + MALL.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_read_rate tools/cu_read_rate.hip && /tmp/cu_read_rate
//
// Prints, per number of active blocks, the aggregate GB/s and the bytes per shader clock and CU (s_memtime deltas).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                      \
  do {                                                                                \
    hipError_t r_ = (e);                                                              \
    if (r_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));                     \
      std::exit(1);                                                                   \
    }                                                                                 \
  } while (0)

constexpr int Q = 19, THREADS = 704;

template <int DEPTH>
__global__ void __launch_bounds__(THREADS) k_pull(const float* __restrict__ src, size_t pop_stride, size_t rows_per_block, int trips, float* out,
                                                   unsigned long long* cycles) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = THREADS / 64;
  // block b owns rows [b * rows_per_block, (b + 1) * rows_per_block) of every population; wave w takes every waves-th row
  const float* base = src + ((size_t)blockIdx.x * rows_per_block + wave) * 64 + lane;
  float acc = 0.f;
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float v[DEPTH][Q];
  size_t row = 0;
  for (int d = 0; d < DEPTH - 1; ++d, row += waves)
#pragma unroll
    for (int l = 0; l < Q; ++l) v[d][l] = base[(size_t)l * pop_stride + row * 64];
  for (int t = 0; t < trips; ++t) {
    // DEPTH - 1 trips are in flight; issue one more, then retire the oldest
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int newest = (d + DEPTH - 1) % DEPTH;
#pragma unroll
      for (int l = 0; l < Q; ++l) v[newest][l] = base[(size_t)l * pop_stride + row * 64];
      row += waves;
#pragma unroll
      for (int l = 0; l < Q; ++l) acc += v[d][l];
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (acc == 12345.678f) lds[threadIdx.x] = acc;  // never true: keeps the LDS allocation and the loads alive
  out[(size_t)blockIdx.x * THREADS + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int DEPTH>
static void run(const float* src, size_t pop_stride, float* out, unsigned long long* cyc, int blocks, size_t rows_per_block, int trips) {
  const size_t lds_bytes = 150 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pull<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {  // the second one counts
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_pull<DEPTH>, dim3(blocks), dim3(THREADS), lds_bytes, 0, src, pop_stride, rows_per_block, trips, out, cyc);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : h) mean += (double)c;
  mean /= blocks;
  const double bytes_per_block = (double)trips * DEPTH * (THREADS / 64) * Q * 256.0;
  std::printf("depth %d (%3d pulls in flight per wave)  blocks %3d  %8.1f GB/s  %6.2f B/clk/CU  (%.0f cycles per block, %.3f ms)\n", DEPTH,
              (DEPTH - 1) * Q + Q, blocks, bytes_per_block * blocks / (ms * 1e-3) / 1e9, bytes_per_block / mean, mean, ms);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

int main() {
  // every block reads rows_per_block rows of 64 cells from each of Q populations: 256 blocks x 8 MiB x 19 = 38 GiB would be too
  // much — 2 MiB per block and population (8192 rows): 256 x 19 x 2 MiB = 9.5 GiB, beyond every cache
  const size_t rows_per_block = 8192, max_blocks = 256;
  const size_t pop_stride = max_blocks * rows_per_block * 64 + 4096;  // elements
  float *src, *out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&src, Q * pop_stride * sizeof(float)));
  CHECK(hipMemset(src, 0, Q * pop_stride * sizeof(float)));
  CHECK(hipMalloc(&out, max_blocks * THREADS * sizeof(float)));
  CHECK(hipMalloc(&cyc, max_blocks * sizeof(unsigned long long)));
  const int waves = THREADS / 64;
  for (int blocks : {256, 192, 128, 64, 32, 8}) {
    run<1>(src, pop_stride, out, cyc, blocks, rows_per_block, (int)(rows_per_block / waves / 1));
    run<2>(src, pop_stride, out, cyc, blocks, rows_per_block, (int)(rows_per_block / waves / 2) - 1);
    run<3>(src, pop_stride, out, cyc, blocks, rows_per_block, (int)(rows_per_block / waves / 3) - 1);
  }
  return 0;
}
