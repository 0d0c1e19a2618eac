// Micro-benchmark for the Winograd go/no-go: the K-loop instruction mix of the only Winograd form of
// the Flipout 3x3 kernel that fits the register file, run long enough for the power management to settle, next to the mix of
// contract_taps_kernel (tools/ubench/mfma_mix.hip, line A) — ALGORITHMIC TFLOP/s (direct-convolution FLOP per output).
//
// Why only this form.  Winograd keeps one accumulator per TRANSFORM-DOMAIN position until the output transform: F(2x2,3x3) needs
// 16 positions per 4 outputs, Flipout doubles it (mean and delta products are combined only after s_out): a wave tile of 32
// Winograd tiles x 32 channels — the smallest MFMA tile, fragment reuse 1 — already needs 16 x 2 x 16 = 512 accumulator registers,
// the whole register file of a SIMD.  The 1-D form F(2,3) along the image row (3 kernel rows x 4 positions per 2 outputs: 6 instead
// of 9 MACs per output, 1.5x fewer MFMAs) needs 4 x 2 sets: 64 outputs x 64 channels = 256 accumulators (AGPRs), one wave per SIMD.
//
// One stage = one (kernel row, 32-channel block) of a wave tile of 32 output PAIRS x 64 channels:
//   LDS   8 reads of the input pixels d0..d3 of the lane's pair (2 k-halves), 4 sign words, 32 weight fragments (4 positions x
//         mean/delta x 2 n-tiles x 2 k-halves)                                             = 40 b128 + 4 b32   (direct, 3 taps: 36 + 6)
//   VALU  s_in masks on d (the sign does not commute with the transform), bf16 -> f32, V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1,
//         V3 = d1 - d3 for x and for x * s_in (v_pk_add_f32), f32 -> bf16 (v_cvt_pk_bf16_f32)   ~ 320            (direct, 3 taps: 108)
//   MFMA  4 positions x 2 n-tiles x 2 k-halves x (mean, delta)                              = 32               (direct, 3 taps: 48)
// No global traffic (upper bound, like lines A-F of mfma_mix): the real kernel adds the weight DMA (4/3 of today's: 4 positions per
// kernel row instead of 3 taps), the patch DMA and the output transform y0 = m0 + m1 + m2, y1 = m1 - m2 - m3 on 256 accumulators.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/wino_mix tools/ubench/wino_mix.hip && tools/ubench/wino_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ void mma_a(f32x16& acc, const u32x4& w, const u32x4& a) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
// packed pair of bf16 -> two f32
__device__ __forceinline__ f32x2 unpack(unsigned p) { return (f32x2){u2f(p << 16), u2f(p & 0xffff0000u)}; }
__device__ __forceinline__ unsigned pack(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }

// MASKS: the x * s_in variant gets its sign masks (false: upper bound without them); XF: the transforms are computed (false: the
// d fragments are fed to the MFMAs as they are — what the MFMA + LDS mix alone sustains)
template <bool MASKS, bool XF>
__global__ __launch_bounds__(256, 1) void kw(int stages, float* sink, unsigned* clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < 65536 / 4; i += 256) ((unsigned*)lds)[i] = 0x3c003c00u + ((i * 2654435761u) & 0x007f007fu);
  __syncthreads();
  f32x16 accm[4][2], accd[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }
      asm volatile("" : "+a"(accm[a][b]));
      asm volatile("" : "+a"(accd[a][b]));
    }
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime(), r0 = (unsigned)__builtin_amdgcn_s_memrealtime();
  struct In { u32x4 D[2][4]; unsigned sw[4]; };
  auto load_in = [&](In& f, int s) __attribute__((always_inline)) {
    const unsigned char* ab = lds + ((s * 4096) & 8191);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int q = (2 * l31 + p) & 127;  // the pair's four input pixels (stride 2 between neighbouring pairs)
        f.D[kk][p] = *(const u32x4*)(ab + q * 64 + (((2 * kk + h) ^ ((q >> 2) & 3)) * 16));
      }
#pragma unroll
    for (int p = 0; p < 4; ++p) f.sw[p] = *(const unsigned*)(lds + 24576 + ((s * 256 + 2 * l31 + p) & 1023) * 4);
  };
  struct Wt { u32x4 WM[2][2], WD[2][2]; };
  auto load_w = [&](Wt& w, int p) __attribute__((always_inline)) {
    const unsigned char* wb = lds + 12288 + (p & 1) * 8192;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        w.WM[kk][ni] = *(const u32x4*)(wb + ((2 * kk + h) * 64 + ((ni * 32 + l31) & 63)) * 16);
        w.WD[kk][ni] = *(const u32x4*)(wb + 4096 + ((2 * kk + h) * 64 + ((ni * 32 + l31) & 63)) * 16);
      }
  };
  // V[variant][kk][position]: the transform-domain fragments of a stage
  auto transform = [&](const In& f, u32x4 (&V)[2][2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int var = 0; var < 2; ++var) {
        u32x4 d[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          d[p] = f.D[kk][p];
          if (var == 1 && MASKS) {
            const unsigned swr = f.sw[p] << (4 * (2 * kk + h));
#pragma unroll
            for (int r = 0; r < 4; ++r) d[p][r] ^= ((swr << r) & 0x80008000u);
          }
        }
        if constexpr (XF) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x2 x0 = unpack(d[0][r]), x1 = unpack(d[1][r]), x2 = unpack(d[2][r]), x3 = unpack(d[3][r]);
            V[var][kk][0][r] = pack(x0 - x2);
            V[var][kk][1][r] = pack(x1 + x2);
            V[var][kk][2][r] = pack(x2 - x1);
            V[var][kk][3][r] = pack(x1 - x3);
          }
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p) V[var][kk][p] = d[p];
        }
      }
  };
  In fa, fb;
  u32x4 V[2][2][4];
  load_in(fa, 0);
  auto stage = [&](In& cur, In& nxt, int s) __attribute__((always_inline)) {
    load_in(nxt, s + 1);
    transform(cur, V);
    Wt w0, w1;
    load_w(w0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      Wt& wc = (p & 1) ? w1 : w0;
      Wt& wn = (p & 1) ? w0 : w1;
      if (p < 3) load_w(wn, p + 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          mma_a(accm[p][ni], wc.WM[kk][ni], V[0][kk][p]);
          mma_a(accd[p][ni], wc.WD[kk][ni], V[1][kk][p]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int s = 0; s < stages; s += 2) {
    stage(fa, fb, s);
    stage(fb, fa, s + 1);
  }
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime(), r1 = (unsigned)__builtin_amdgcn_s_memrealtime();
  float v = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      asm volatile("" : "+a"(accm[a][b]));
      asm volatile("" : "+a"(accd[a][b]));
      v += accm[a][b][0] + accd[a][b][5];
    }
  if (v == 12345.678f) sink[0] = v;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <bool MASKS, bool XF>
void run(const char* name, float* sink, unsigned* clk) {
  auto fn = kw<MASKS, XF>;
  const int lds_bytes = 163840;  // one 4-wave block per CU: one wave per SIMD, 512 registers
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int grid = 256, stages = 60000, reps = 6;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 6; ++w) fn<<<grid, 256, lds_bytes>>>(stages, sink, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) fn<<<grid, 256, lds_bytes>>>(stages, sink, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned c[2];
  hipMemcpy(c, clk, 8, hipMemcpyDeviceToHost);
  // a stage = one kernel row x 32 channels of 64 outputs x 64 channels, mean + delta: direct FLOP = 2 * 2 * 3 taps * 32 * 64 * 64
  const double algo = (double)reps * grid * 4 * stages * (2.0 * 2 * 3 * 32 * 64 * 64);
  const double issued = (double)reps * grid * 4 * stages * 32 * 32768.0;
  const double ghz = (double)c[0] / (double)c[1] * 0.1;
  printf("%-64s algorithmic %7.1f TFLOP/s (MFMA-issued %7.1f)  clock %.3f GHz  %6.0f cycles/stage (32 MFMAs = 1024)\n", name,
         algo / (ms * 1e-3) / 1e12, issued / (ms * 1e-3) / 1e12, ghz, (double)c[0] / stages);
}

int main() {
  float* sink; unsigned* clk;
  hipMalloc(&sink, 16); hipMalloc(&clk, 16);
  for (int round = 0; round < 2; ++round) {
    run<true, true>("W1 F(2,3) rows, 64 px x 64 ch, AGPR acc, 1 block/CU, masks + transforms", sink, clk);
    run<false, true>("W2 W1 without the s_in masks", sink, clk);
    run<false, false>("W3 W1 without masks and transforms (its MFMA + LDS mix alone)", sink, clk);
  }
  return 0;
}
