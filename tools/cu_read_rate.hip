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


// The same question for the TILE pattern of k_step2 ((8 x 64) tiles of a 512-cell-wide plane, marching along x): wave j < 8 of block b
// pulls row j of its tile — 256 bytes, 2 KB from the next row — from Q populations per plane, and (STORE) writes the row of the
// previous plane to a second set of arrays, as phase B does.  No arithmetic, no LDS traffic: what the memory system does with
// this access pattern and this occupancy.  Variants: plain instead of non-temporal stores; stores before / after the pulls in
// program order; a workgroup barrier between the plane's stores and its pulls (the phased order of k_step2); 1, 2 or 4 blocks per CU.
template <int STORE /*0 none, 1 non-temporal, 2 plain*/, int ORDER /*0 pulls then stores, 1 stores then pulls, 2 stores | barrier | pulls*/, int NT = THREADS>
__global__ void __launch_bounds__(NT) k_tile(const float* __restrict__ src, float* __restrict__ dst, size_t pop_stride, size_t plane, int planes,
                                                   float* out, int tz /*tile width in cells: 64 (8 x 64 tiles), 128 (4 x 128), 256, 512*/,
                                                   int interleaved /*1: populations interleaved row by row, [x][y][l][z], instead of [l][x][y][z]*/) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // 512 cells per tile whatever its shape: ty = 512 / tz rows of tz cells, 512 / tz tiles across the plane's 512-cell rows
  const int per_row = tz / 64, ty = 512 / tz, tiles_z = 512 / tz;
  const int ty0 = (blockIdx.x / tiles_z) * ty, tz0 = (blockIdx.x % tiles_z) * tz;
  size_t cell = (size_t)(ty0 + (wave & 7) / per_row) * 512 + tz0 + ((wave & 7) % per_row) * 64 + lane;
  if (interleaved) {  // same bytes per population, plane and row; the Q rows (x, y, .) of the populations follow one another
    cell = (size_t)(ty0 + (wave & 7) / per_row) * 512 * Q + tz0 + ((wave & 7) % per_row) * 64 + lane;
    pop_stride = 512;
    plane *= Q;
  }
  float acc = 0.f;
  const bool active = wave < 8;
  float v[Q], w[Q];
  if (active) {
#pragma unroll
    for (int l = 0; l < Q; ++l) v[l] = src[(size_t)l * pop_stride + cell];
  }
  for (int x = 1; x < planes; ++x) {
    if (active) {
#pragma unroll
      for (int l = 0; l < Q; ++l) w[l] = v[l];
      auto pulls = [&]() {
#pragma unroll
        for (int l = 0; l < Q; ++l) v[l] = src[(size_t)l * pop_stride + (size_t)x * plane + cell];
      };
      auto stores = [&]() {
#pragma unroll
        for (int l = 0; l < Q; ++l) {
          float* d = dst + (size_t)l * pop_stride + (size_t)(x - 1) * plane + cell;
          if (STORE == 1)
            __builtin_nontemporal_store(w[l] + 1.0f, d);
          else if (STORE == 2)
            *d = w[l] + 1.0f;
          else
            acc += w[l];
        }
      };
      if (ORDER == 0) {
        pulls();
        __builtin_amdgcn_sched_barrier(0);
        stores();
      } else {
        stores();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ORDER == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (active && ORDER != 0) {
#pragma unroll
      for (int l = 0; l < Q; ++l) v[l] = src[(size_t)l * pop_stride + (size_t)x * plane + cell];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (active) {
#pragma unroll
    for (int l = 0; l < Q; ++l) acc += v[l];
  }
  if (acc == 12345.678f) lds[threadIdx.x] = acc;
  out[(size_t)blockIdx.x * THREADS + threadIdx.x] = acc;
}

// BATCH planes per trip: the stores of BATCH planes, a barrier, the pulls of the next BATCH planes (longer runs of one direction
// per CU); NTLOAD: non-temporal pulls.
template <int BATCH, bool NTLOAD>
__global__ void __launch_bounds__(THREADS) k_tile_batch(const float* __restrict__ src, float* __restrict__ dst, size_t pop_stride, size_t plane, int planes,
                                                         float* out) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ty0 = (blockIdx.x / 8) * 8, tz0 = (blockIdx.x % 8) * 64;
  const size_t cell = (size_t)(ty0 + (wave & 7)) * 512 + tz0 + lane;
  const bool active = wave < 8;
  float v[BATCH][Q];
  float acc = 0.f;
  auto pull = [&](int b, int x) {
#pragma unroll
    for (int l = 0; l < Q; ++l) {
      const float* p = src + (size_t)l * pop_stride + (size_t)x * plane + cell;
      v[b][l] = NTLOAD ? __builtin_nontemporal_load(p) : *p;
    }
  };
  if (active)
    for (int b = 0; b < BATCH; ++b) pull(b, b);
  for (int x = BATCH; x + BATCH <= planes; x += BATCH) {
    if (active) {
#pragma unroll
      for (int b = 0; b < BATCH; ++b)
#pragma unroll
        for (int l = 0; l < Q; ++l) __builtin_nontemporal_store(v[b][l] + 1.0f, dst + (size_t)l * pop_stride + (size_t)(x - BATCH + b) * plane + cell);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (active) {
#pragma unroll
      for (int b = 0; b < BATCH; ++b) pull(b, x + b);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (active)
    for (int b = 0; b < BATCH; ++b)
#pragma unroll
      for (int l = 0; l < Q; ++l) acc += v[b][l];
  if (acc == 12345.678f) lds[threadIdx.x] = acc;
  out[(size_t)blockIdx.x * THREADS + threadIdx.x] = acc;
}

// the phased tile copy with 16-BYTE lanes: lane i of wave w moves cells 4 (i % 16) .. + 3 of tile row 4 w + i / 16 — a wave covers four
// rows (1 KB per instruction), WAVES16 = 2 waves the tile; with 8 active waves each takes a quarter of the populations' planes instead
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PLANES_PER_WAVE /*1: two active waves, every plane; 4: eight active waves, wave w takes planes x with x % 4 == w / 2*/>
__global__ void __launch_bounds__(THREADS) k_tile16(const float* __restrict__ src, float* __restrict__ dst, size_t pop_stride, size_t plane, int planes) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ty0 = (blockIdx.x / 8) * 8, tz0 = (blockIdx.x % 8) * 64;
  if (threadIdx.x == 100000) lds[0] = 1.f;
  const bool active = PLANES_PER_WAVE == 1 ? wave < 2 : wave < 8;
  const int half = wave & 1, phase = (wave >> 1) & 3;  // (idle waves run the same number of trips: the barrier counts arrivals)
  const size_t cell = (size_t)(ty0 + 4 * half + lane / 16) * 512 + tz0 + 4 * (lane % 16);
  f32x4 v[Q];
  int x0 = PLANES_PER_WAVE == 1 ? 0 : phase;
  if (active)
#pragma unroll
    for (int l = 0; l < Q; ++l) v[l] = *reinterpret_cast<const f32x4*>(src + (size_t)l * pop_stride + (size_t)x0 * plane + cell);
  for (int x = x0 + PLANES_PER_WAVE; x < planes + PLANES_PER_WAVE; x += PLANES_PER_WAVE) {
    if (active) {
#pragma unroll
      for (int l = 0; l < Q; ++l)
        __builtin_nontemporal_store(v[l], reinterpret_cast<f32x4*>(dst + (size_t)l * pop_stride + (size_t)(x - PLANES_PER_WAVE) * plane + cell));
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (active && x < planes) {
#pragma unroll
      for (int l = 0; l < Q; ++l) v[l] = *reinterpret_cast<const f32x4*>(src + (size_t)l * pop_stride + (size_t)x * plane + cell);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int PPW>
static void run_tile16(const float* src, float* dst, size_t pop_stride) {
  const size_t lds_bytes = 150 * 1024;
  const int blocks = 256;
  const size_t plane = (size_t)512 * 256;
  const int planes = (int)((pop_stride - 4096) / plane) / 4 * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile16<PPW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_tile16<PPW>, dim3(blocks), dim3(THREADS), lds_bytes, 0, src, dst, pop_stride, plane, planes);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 2.0 * blocks * (double)planes * 8 * Q * 256.0;
  std::printf("(8 x  64) tiles, 1 block per CU, 16-byte lanes, %d active waves, stores | barrier | pulls: %7.1f GB/s read + write (%.3f ms)\n",
              PPW == 1 ? 2 : 8, bytes / (ms * 1e-3) / 1e9, ms);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

// stores only, same rows
__global__ void __launch_bounds__(THREADS) k_tile_store(float* __restrict__ dst, size_t pop_stride, size_t plane, int planes, int active_waves) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ty0 = (blockIdx.x / 8) * 8, tz0 = (blockIdx.x % 8) * 64;
  if (threadIdx.x == 100000) lds[0] = 1.f;
  // active_waves = 8: one row per wave; 4: two rows per wave (rows w and w + 4)
  const int rows_per_wave = 8 / active_waves;
  if (wave >= active_waves) return;
  for (int x = 0; x < planes; ++x)
    for (int r = 0; r < rows_per_wave; ++r) {
      const size_t cell = (size_t)(ty0 + wave + r * active_waves) * 512 + tz0 + lane;
#pragma unroll
      for (int l = 0; l < Q; ++l) __builtin_nontemporal_store((float)(x + l), dst + (size_t)l * pop_stride + (size_t)x * plane + cell);
    }
}

static void run_store(float* dst, size_t pop_stride, int blocks_per_cu, int active_waves) {
  const size_t lds_bytes = (150 * 1024) / blocks_per_cu;
  const int blocks = 256 * blocks_per_cu;
  const size_t plane = (size_t)512 * (blocks / 8) * 8;
  const int planes = (int)((pop_stride - 4096) / plane);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_store), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_tile_store, dim3(blocks), dim3(THREADS), lds_bytes, 0, dst, pop_stride, plane, planes, active_waves);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)blocks * planes * 8 * Q * 256.0;
  std::printf("(8 x  64) tiles, %d block(s) per CU, %d storing waves per block, stores only: %7.1f GB/s (%.3f ms)\n", blocks_per_cu, active_waves,
              bytes / (ms * 1e-3) / 1e9, ms);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

template <int BATCH, bool NTLOAD>
static void run_batch(const float* src, float* dst, size_t pop_stride, float* out) {
  const size_t lds_bytes = 150 * 1024;
  const int blocks = 256;
  const size_t plane = (size_t)512 * 256;
  const int planes = (int)((pop_stride - 4096) / plane);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_batch<BATCH, NTLOAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_tile_batch<BATCH, NTLOAD>), dim3(blocks), dim3(THREADS), lds_bytes, 0, src, dst, pop_stride, plane, planes, out);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 2.0 * blocks * (double)(planes / BATCH * BATCH) * 8 * Q * 256.0;
  std::printf("(8 x  64) tiles, stores of %d plane(s) | barrier | %spulls of %d plane(s): %7.1f GB/s read + write (%.3f ms)\n", BATCH,
              NTLOAD ? "non-temporal " : "", BATCH, bytes / (ms * 1e-3) / 1e9, ms);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

template <int STORE, int ORDER, int NT = THREADS>
static void run_tile(const float* src, float* dst, size_t pop_stride, float* out, int blocks_per_cu, const char* what, int tz = 64, int interleaved = 0) {
  const size_t lds_bytes = (150 * 1024) / blocks_per_cu;
  const int blocks = 256 * blocks_per_cu;
  const size_t plane = (size_t)512 * (blocks / (512 / tz)) * (512 / tz);  // rows = tiles_y * ty
  const int planes = (int)((pop_stride - 4096) / plane);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<STORE, ORDER, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_tile<STORE, ORDER, NT>), dim3(blocks), dim3(NT), lds_bytes, 0, src, dst, pop_stride, plane, planes, out, tz, interleaved);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double rd = (double)blocks * planes * 8 * Q * 256.0, wr = STORE ? (double)blocks * (planes - 1) * 8 * Q * 256.0 : 0.0;
  std::printf("(%d x %3d) tiles%s, %d block(s) per CU, %-44s read %7.1f + write %7.1f = %7.1f GB/s (%.3f ms)\n", 512 / tz, tz, interleaved ? ", populations interleaved by row" : "", blocks_per_cu, what, rd / (ms * 1e-3) / 1e9,
              wr / (ms * 1e-3) / 1e9, (rd + wr) / (ms * 1e-3) / 1e9, ms);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
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
  CHECK(hipMalloc(&out, 2048 * THREADS * sizeof(float)));  // (up to 1024 blocks in the many-waves runs)
  CHECK(hipMalloc(&cyc, max_blocks * sizeof(unsigned long long)));
  const int waves = THREADS / 64;
  for (int blocks : {256, 192, 128, 64, 32, 8}) {
    run<1>(src, pop_stride, out, cyc, blocks, rows_per_block, (int)(rows_per_block / waves / 1));
    run<2>(src, pop_stride, out, cyc, blocks, rows_per_block, (int)(rows_per_block / waves / 2) - 1);
    run<3>(src, pop_stride, out, cyc, blocks, rows_per_block, (int)(rows_per_block / waves / 3) - 1);
  }
  // tile pattern: pop_stride elements hold 512 x 256 x planes cells -> planes = 1024 (537 MB per population, as a 512^3 field)
  float* dst;
  CHECK(hipMalloc(&dst, Q * pop_stride * sizeof(float)));
  CHECK(hipMemset(dst, 0, Q * pop_stride * sizeof(float)));
  for (int bpc : {1, 2}) {
    run_tile<0, 0>(src, dst, pop_stride, out, bpc, "pulls only");
    run_tile<1, 0>(src, dst, pop_stride, out, bpc, "pulls, then non-temporal stores");
    run_tile<2, 0>(src, dst, pop_stride, out, bpc, "pulls, then plain stores");
    run_tile<1, 1>(src, dst, pop_stride, out, bpc, "non-temporal stores, then pulls");
    run_tile<1, 2>(src, dst, pop_stride, out, bpc, "non-temporal stores | barrier | pulls");
  }
  run_store(dst, pop_stride, 1, 8);
  run_store(dst, pop_stride, 1, 4);
  run_store(dst, pop_stride, 2, 8);
  run_store(dst, pop_stride, 1, 8);
  run_batch<1, false>(src, dst, pop_stride, out);
  run_batch<1, true>(src, dst, pop_stride, out);
  run_batch<2, false>(src, dst, pop_stride, out);
  run_batch<3, false>(src, dst, pop_stride, out);
  run_batch<1, false>(src, dst, pop_stride, out);
  run_batch<2, false>(src, dst, pop_stride, out);
  run_tile16<1>(src, dst, pop_stride);
  run_tile16<4>(src, dst, pop_stride);
  run_tile16<1>(src, dst, pop_stride);
  // many waves per CU: 512-thread blocks (8 waves, all storing), 3 and 4 of them per CU (24 / 32 waves)
  for (int bpc : {3, 4}) {
    run_tile<1, 0, 512>(src, dst, pop_stride, out, bpc, "512-thread blocks: pulls, then stores");
    run_tile<1, 2, 512>(src, dst, pop_stride, out, bpc, "512-thread blocks: stores | barrier | pulls");
  }
  for (int tz : {64, 512}) {
    run_tile<0, 0>(src, dst, pop_stride, out, 1, "pulls only", tz, 1);
    run_tile<1, 0>(src, dst, pop_stride, out, 1, "pulls, then non-temporal stores", tz, 1);
    run_tile<1, 2>(src, dst, pop_stride, out, 1, "non-temporal stores | barrier | pulls", tz, 1);
  }
  for (int tz : {128, 256, 512}) {
    run_tile<0, 0>(src, dst, pop_stride, out, 1, "pulls only", tz);
    run_tile<1, 0>(src, dst, pop_stride, out, 1, "pulls, then non-temporal stores", tz);
    run_tile<1, 2>(src, dst, pop_stride, out, 1, "non-temporal stores | barrier | pulls", tz);
  }
  return 0;
}
