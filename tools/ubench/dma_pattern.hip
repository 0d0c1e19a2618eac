// Micro-benchmark: LDS-DMA (global_load_lds_dwordx4) and plain global_load_dwordx4 throughput per access pattern.
//   pattern 0: wave instruction reads 1 KiB contiguous (lane stride 16 B)
//   pattern 1: lane stride 128 B  (64 different lines, 16 B used per line)   <- what the v1-v3 kernels did
//   pattern 2: lane stride 128 B, 4 consecutive instructions cover 64 B of each line
//   pattern 3: lane stride 1 KiB
// buffer 32 MiB (L2/MALL resident after the first pass), 256 CUs x 8 waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ void dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int PAT, bool DMA>
__global__ __launch_bounds__(512) void k(const unsigned char* x, size_t bytes, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t gw = (size_t)blockIdx.x * 8 + wave;  // global wave id
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      size_t off;
      const size_t blk = (gw * iters + it) * 4;
      if (PAT == 0) off = ((blk + j) * 1024 + lane * 16);
      else if (PAT == 1) off = ((blk + j) * 8192 + lane * 128);
      else if (PAT == 2) off = (blk * 2048 + lane * 128 + j * 16);
      else if (PAT == 3) off = ((blk + j) * 65536 + lane * 1024);
      else if (PAT == 4) off = ((blk + j) * 1024 + (lane & 7) * 128 + (lane >> 3) * 16);        // 1 KiB, granule-major permutation
      else if (PAT == 5) off = ((blk + j) * 16384 + (lane >> 4) * 2304 + (lane & 15) * 16);     // 4 runs of 256 B
      else if (PAT == 6) off = ((blk + j) * 2048 + (lane >> 2) * 128 + (lane & 3) * 16);        // 16 half lines
      else if (PAT == 7) off = ((blk + j) * 2048 + (lane >> 3) * 256 + (lane & 7) * 16);        // 8 full lines, 256 B apart
      else if (PAT == 8) off = ((blk + j) * 2048 + (lane & 7) * 256 + (lane >> 3) * 16);        // same, granule-major
      else off = ((blk + j) * 4096 + (lane >> 3) * 512 + (lane & 7) * 16);                      // 8 full lines, 512 B apart (stride-2 conv, C=128)
      off %= bytes;
      if (DMA) dma16(x + off, smem + wave * 4096 + j * 1024);
      else { u32x4 v = *(const u32x4*)(x + off); acc += v; }
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  if (DMA) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = *(const u32x4*)(smem + threadIdx.x * 16);
  }
  if (acc[0] == 0x12345678u) sink[0] = acc[1] + acc[2] + acc[3];
}
template <int PAT, bool DMA>
void run(const unsigned char* x, size_t bytes, unsigned* sink) {
  const int iters = 256, grid = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<PAT, DMA><<<grid, 512, 32768>>>(x, bytes, 8, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<PAT, DMA><<<grid, 512, 32768>>>(x, bytes, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double req = (double)grid * 8 * iters * 4 * 64;  // 16-byte lane requests
  printf("pattern %d %s: %.3f ms  %.1f G lane-req/s  useful %.2f TB/s  (%.2f cyc/req/CU @2.1GHz)\n", PAT, DMA ? "dma " : "vgpr",
         ms, req / ms / 1e6, req * 16 / ms / 1e9, ms * 1e-3 * 2.1e9 * 256 / req);
}
int main() {
  const size_t bytes = 32u << 20;
  unsigned char* x; unsigned* sink;
  hipMalloc(&x, bytes); hipMalloc(&sink, 64); hipMemset(x, 1, bytes);
  run<0, true>(x, bytes, sink); run<1, true>(x, bytes, sink); run<2, true>(x, bytes, sink); run<3, true>(x, bytes, sink);
  run<4, true>(x, bytes, sink); run<5, true>(x, bytes, sink); run<6, true>(x, bytes, sink); run<7, true>(x, bytes, sink); run<8, true>(x, bytes, sink); run<9, true>(x, bytes, sink);
  run<0, false>(x, bytes, sink); run<1, false>(x, bytes, sink); run<2, false>(x, bytes, sink); run<3, false>(x, bytes, sink);
  return 0;
}
