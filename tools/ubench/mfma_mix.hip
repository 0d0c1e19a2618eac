// Micro-benchmark: the instruction mix of one K-stage of the Flipout tap kernel (fragment reads from LDS, s_in masks on
// the activation fragments, mean + delta MFMAs, one barrier) at different register-tile shapes, run long enough for the
// power management to settle.  What it answers: at the clock the chip sustains under each mix, how many TFLOP/s does
//   A  2x2 MFMA tiles per wave (128 accumulators in VGPRs), 4-wave blocks, 2 blocks per CU      — contract_taps_kernel
//   B  4x2 tiles (256 accumulators in AGPRs), 4-wave blocks, 1 block per CU (1 wave per SIMD, 512 registers)
//   C  2x4 tiles (256 accumulators in AGPRs), same occupancy                                   — halves the masks per MFMA
//   D  A without the masks, E  A without masks and with half the fragment reads (upper bounds of what removing them buys)
//   G  A + the kernel's weight-tile DMA (8 KiB per block-stage out of an L2-resident buffer, 4-slot LDS ring)
//   H  G + 4 KiB per block-stage of activations streamed from a 1-GiB buffer (HBM),  I  B + the weight-tile DMA
// deliver?  A-F: no global memory traffic in the loop (weights and activations stay in LDS).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_mix tools/ubench/mfma_mix.hip && tools/ubench/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../bayesian_torch_amd/csrc/btx_rng.h"  // BTX-RNG v1: Philox4x32-10 + Box-Muller (the kernels' own generator)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <bool AG>
__device__ __forceinline__ void mma(f32x16& acc, const u32x4& w, const u32x4& a) {
  if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
  else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

// MI x NI tiles of 32 pixels x 32 channels per wave; MASK: s_in masks; RD: fragment reads per stage (1 = all, 2 = every 2nd stage)
// SAMPLE (round 6, row g3 of the verdict: "in-kernel sampling on the convolution fast path"): every stage, each of the block's 256
// threads draws the 8 normals of its 16 bytes of the stage's 4-KiB delta tile (2 x Philox4x32-10 + 4 Box-Muller pairs, hardware
// log / sqrt / sin / cos as in btx_presample.h), multiplies by a sigma it has in a register, rounds to bf16 and stores the granule —
// what a workgroup of the tap kernel would have to do per stage to make its own sigma * eps tile instead of DMA-ing a pre-sampled one.
template <int MI, int NI, bool AG, bool MASK, int RD, int BPC, int DMA = 0, bool SAMPLE = false>
__global__ __launch_bounds__(256, BPC) void k(int stages, float* sink, unsigned* clk, const unsigned char* wbuf = nullptr,
                                              const unsigned char* xbuf = nullptr, unsigned xbytes = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < 65536 / 4; i += 256) ((unsigned*)lds)[i] = 0x3c003c00u + ((i * 2654435761u) & 0x007f007fu);
  __syncthreads();
  f32x16 accm[MI][NI], accd[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }
  if constexpr (AG) {
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b) { asm volatile("" : "+a"(accm[a][b])); asm volatile("" : "+a"(accd[a][b])); }
  }
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime(), r0 = (unsigned)__builtin_amdgcn_s_memrealtime();
  struct Frag { u32x4 A[2][MI], WM[2][NI], WD[2][NI]; unsigned sw[MI]; };
  auto load = [&](Frag& f, int s) __attribute__((always_inline)) {
    const unsigned char* ab = lds + ((s * 4096) & 8191);
    const unsigned char* wb = lds + 12288;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) f.A[kk][mi] = *(const u32x4*)(ab + ((mi * 32 + l31) & 63) * 64 + (((2 * kk + h) ^ ((l31 >> 2) & 3)) * 16));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        f.WM[kk][ni] = *(const u32x4*)(wb + ((2 * kk + h) * 64 + ((ni * 32 + l31) & 63)) * 16);
        f.WD[kk][ni] = *(const u32x4*)(wb + 4096 + ((2 * kk + h) * 64 + ((ni * 32 + l31) & 63)) * 16);
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) f.sw[mi] = *(const unsigned*)(lds + 24576 + ((s * 256 + mi * 32 + l31) & 1023) * 4);
  };
  // DMA stream of the kernel: every wave fetches 2 x 1 KiB of the stage's weight tiles (mean + delta row of an 8-KiB stage,
  // 147 KiB per lane of tiles: L2-resident) into a 4-slot ring three stages ahead; DMA == 2: plus 1 KiB per wave of
  // activations streamed from a buffer far larger than the caches
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wbuf, 0, 147456u * 8u, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xbuf, 0, xbytes, 0x00020000);
  unsigned xoff = (unsigned)((blockIdx.x * 4u + wave) * 1048576u) % (xbytes ? xbytes : 1u);
  auto dma = [&](int s) __attribute__((always_inline)) {
    if constexpr (DMA >= 1) {
      const unsigned so = (unsigned)((s % 18) * 8192 + (blockIdx.x & 7) * 147456) + wave * 1024u;
      unsigned char* ld = lds + 28672 + (s & 3) * 8192 + wave * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)ld, 16, lane * 16u, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(ld + 4096), 16, lane * 16u, so + 4096u, 0, 0);
    }
    if constexpr (DMA >= 2) {
      unsigned char* ld = lds + 61440 + (s & 1) * 4096 + wave * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)ld, 16, lane * 16u, xoff, 0, 0);
      xoff += 1024u; if (xoff + 1024u > xbytes) xoff = 0;
    }
  };
  // stage s multiplies `cur` while the fragments of stage s+1 are read into `nxt` (as the kernel does)
  auto stage = [&](Frag& cur, Frag& nxt, int s) __attribute__((always_inline)) {
    dma(s + 3);
    if constexpr (SAMPLE) {
      float z[8];
      btx_normal4_hw((uint32_t)(s * 512 + 2 * tid), 7u, 3u, 0u, 0x1234u, 0x5678u, z);
      btx_normal4_hw((uint32_t)(s * 512 + 2 * tid + 1), 7u, 3u, 0u, 0x1234u, 0x5678u, z + 4);
      const float sg = 0.0486f + 1e-6f * (float)lane;
      typedef __attribute__((ext_vector_type(8))) float f32x8_;
      const f32x8_ v = {z[0] * sg, z[1] * sg, z[2] * sg, z[3] * sg, z[4] * sg, z[5] * sg, z[6] * sg, z[7] * sg};
      *(bf16x8*)(lds + 12288 + 4096 + ((s & 1) ? 8192 : 0) + tid * 16) = __builtin_convertvector(v, bf16x8);
    }
    if (RD == 1 || (s & 2) == 0) load(nxt, s + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma<AG>(accm[mi][ni], cur.WM[kk][ni], cur.A[kk][mi]);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if constexpr (MASK) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const unsigned swr = cur.sw[mi] << (4 * (2 * kk + h));
#pragma unroll
          for (int d = 0; d < 4; ++d) cur.A[kk][mi][d] ^= ((swr << d) & 0x80008000u);
        }
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma<AG>(accd[mi][ni], cur.WD[kk][ni], cur.A[kk][mi]);
    }
    if constexpr (DMA == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // what was issued two stages ago has landed
    if constexpr (DMA == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  Frag fa, fb;
  load(fa, 0); load(fb, 1);
  for (int s = 0; s < stages; s += 2) {
    stage(fa, fb, s);
    stage(fb, fa, s + 1);
  }
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime(), r1 = (unsigned)__builtin_amdgcn_s_memrealtime();
  float v = 0.f;
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NI; ++b) {
      if constexpr (AG) { asm volatile("" : "+a"(accm[a][b])); asm volatile("" : "+a"(accd[a][b])); }
      v += accm[a][b][0] + accd[a][b][5];
    }
  if (v == 12345.678f) sink[0] = v;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

static unsigned char* g_wbuf = nullptr;
static unsigned char* g_xbuf = nullptr;
static const unsigned g_xbytes = 1u << 30;
template <int MI, int NI, bool AG, bool MASK, int RD, int BPC, int DMA = 0, bool SAMPLE = false>
void run(const char* name, float* sink, unsigned* clk) {
  auto fn = k<MI, NI, AG, MASK, RD, BPC, DMA, SAMPLE>;
  const int lds_bytes = BPC == 2 ? 81920 : 163840;  // pins the blocks per CU (DMA rings: 64 KiB .. 108 KiB > the 80 KiB of a block when BPC == 2: folded below)
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int grid = 256 * BPC;
  const int stages = 100000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 6; ++w) fn<<<grid, 256, lds_bytes>>>(stages, sink, clk, g_wbuf, g_xbuf, g_xbytes);  // ~0.1 s each: power management settles
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 6;
  for (int w = 0; w < reps; ++w) fn<<<grid, 256, lds_bytes>>>(stages, sink, clk, g_wbuf, g_xbuf, g_xbytes);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned c[2];
  hipMemcpy(c, clk, 8, hipMemcpyDeviceToHost);
  const double flop = (double)reps * grid * 4 * stages * (2.0 * MI * NI * 2) * 32768.0;
  const double ghz = (double)c[0] / (double)c[1] * 0.1;
  const double cyc_stage = (double)c[0] / stages;
  printf("%-58s %7.1f TFLOP/s  clock %.3f GHz  %6.0f cycles/stage (MFMA pipe needs %d: %.0f %% busy)\n", name, flop / (ms * 1e-3) / 1e12, ghz,
         cyc_stage, 2 * MI * NI * 2 * 32 * BPC, 100.0 * (2 * MI * NI * 2 * 32 * BPC) / cyc_stage);
}

int main() {
  float* sink; unsigned* clk;
  hipMalloc(&sink, 16); hipMalloc(&clk, 16);
  hipMalloc(&g_wbuf, 147456 * 8); hipMemset(g_wbuf, 0x3c, 147456 * 8);
  hipMalloc(&g_xbuf, g_xbytes); hipMemset(g_xbuf, 0x3c, g_xbytes);
  for (int round = 0; round < 2; ++round) {
    run<2, 2, false, true, 1, 2>("A  2x2 tiles, VGPR acc, 2 blocks/CU, masks", sink, clk);
    run<4, 2, true, true, 1, 1>("B  4x2 tiles, AGPR acc, 1 block/CU, masks", sink, clk);
    run<2, 4, true, true, 1, 1>("C  2x4 tiles, AGPR acc, 1 block/CU, masks", sink, clk);
    run<2, 2, false, false, 1, 2>("D  A without masks", sink, clk);
    run<2, 2, false, false, 2, 2>("E  A without masks, half the fragment reads", sink, clk);
    run<4, 2, true, false, 1, 1>("F  B without masks", sink, clk);
    run<2, 2, false, true, 1, 2, 1>("G  A + weight-tile DMA (8 KiB per block-stage from L2)", sink, clk);
    run<2, 2, false, true, 1, 2, 2>("H  G + 4 KiB per block-stage of activations from HBM", sink, clk);
    run<4, 2, true, true, 1, 1, 1>("I  B + weight-tile DMA (8 KiB per 512-pixel block-stage)", sink, clk);
    run<2, 2, false, true, 1, 2, 0, true>("J  A + the stage's delta tile sampled in the block (8 normals/thread)", sink, clk);
  }
  return 0;
}
