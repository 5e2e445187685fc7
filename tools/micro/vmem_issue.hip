// Micro-benchmark (runs on the GPU box): cycles a CU needs per vector-memory wave-instruction when the data is L2-resident, for the
// access shapes the two-step kernel uses.   hipcc --offload-arch=gfx950 -O3 tools/micro/vmem_issue.hip -o /tmp/vmem_issue && /tmp/vmem_issue
// Every block (704 threads = 11 waves, one block per CU through 150 KB of LDS) loops over its own 64 KB window of a buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(704) k(const float* __restrict__ src, float* __restrict__ dst, int iters, unsigned long long* cyc) {
  extern __shared__ float pad[];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const float* base = src + (size_t)blockIdx.x * 16384;  // 64 KB window per block
  float* out = dst + (size_t)blockIdx.x * 16384;
  float acc = 0.f;
  unsigned long long t0 = 0;
  if (t == 0) t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int row = (it * 11 + w) & 63;  // 256-B rows
    if constexpr (MODE == 0) {  // aligned dword loads: 64 lanes x 4 B = one 256-B row
#pragma unroll
      for (int k2 = 0; k2 < 19; ++k2) acc += base[((row + k2) & 63) * 256 / 4 * 1 + lane + 0];
    } else if constexpr (MODE == 1) {  // dword loads shifted by one float (c_z = +-1 pulls)
#pragma unroll
      for (int k2 = 0; k2 < 19; ++k2) acc += base[((row + k2) & 62) * 64 + lane + 1];
    } else if constexpr (MODE == 2) {  // the grown-tile shape: 66-float rows, a wave spans the tail of one row and the head of the next
#pragma unroll
      for (int k2 = 0; k2 < 19; ++k2) {
        const int cell = w * 64 + lane, r = cell / 66, c = cell % 66;
        acc += base[((r + k2) & 31) * 512 + 127 + c];
      }
    } else if constexpr (MODE == 3) {  // aligned dwordx4 loads: 64 lanes x 16 B = 1 KB
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) {
        const f4 v = reinterpret_cast<const f4*>(base)[((row + k2 * 4) & 63) / 4 * 64 + lane];
        acc += v.x + v.y + v.z + v.w;
      }
    } else if constexpr (MODE == 4) {  // aligned dword stores
#pragma unroll
      for (int k2 = 0; k2 < 19; ++k2) __builtin_nontemporal_store(acc + k2, &out[((row + k2) & 63) * 64 + lane]);
    } else if constexpr (MODE == 5) {  // aligned dwordx4 stores
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) {
        f4 v = {acc, acc + 1, acc + 2, acc + 3};
        __builtin_nontemporal_store(v, &reinterpret_cast<f4*>(out)[((row + k2 * 4) & 63) / 4 * 64 + lane]);
      }
    } else if constexpr (MODE == 6) {  // plain (cached) aligned dword stores
#pragma unroll
      for (int k2 = 0; k2 < 19; ++k2) out[((row + k2) & 63) * 64 + lane] = acc + k2;
    }
  }
  if (acc == 12345.678f) dst[t] = acc;
  __syncthreads();
  if (t == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

template <int MODE>
void run(const char* name, int per_iter, int bytes_per_instr, float* a, float* b, unsigned long long* cyc) {
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(704), 150 * 1024, 0, a, b, 10, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(704), 150 * 1024, 0, a, b, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v / blocks;
  const double instrs = (double)iters * per_iter * 11;  // wave-instructions per CU
  printf("%-44s %8.1f cycles / wave-instruction / CU   %6.1f B/clk/CU   (%.3f ms, %.2f GHz by the event clock)\n", name, mean / instrs,
         bytes_per_instr * instrs / mean, ms, mean / (ms * 1e6));
}

int main() {
  float *a, *b;
  unsigned long long* cyc;
  hipMalloc(&a, 256 * 65536 + 4096);
  hipMalloc(&b, 256 * 65536 + 4096);
  hipMalloc(&cyc, 256 * 8);
  hipMemset(a, 0, 256 * 65536 + 4096);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  run<0>("dword loads, aligned 256-B rows", 19, 256, a, b, cyc);
  run<1>("dword loads, rows shifted by 4 B", 19, 256, a, b, cyc);
  run<2>("dword loads, 66-float grown-tile rows", 19, 256, a, b, cyc);
  run<3>("dwordx4 loads, aligned 1 KB", 5, 1024, a, b, cyc);
  run<4>("dword nt stores, aligned", 19, 256, a, b, cyc);
  run<5>("dwordx4 nt stores, aligned", 5, 1024, a, b, cyc);
  run<6>("dword plain stores, aligned", 19, 256, a, b, cyc);
  return 0;
}
