// Micro-benchmark: does the VALU stream of ONE wave slow down the MFMA stream of ANOTHER wave on the same SIMD?  (round 6, the
// stem + pool kernel: a K-role wave and a store-role wave share every SIMD.)  One 8-wave block per CU: waves 0-3 issue independent
// v_mfma_f32_32x32x16_bf16 back to back (4 accumulators), waves 4-7 — one per SIMD beside them — do nothing | issue independent
// v_fma_f32 back to back | issue v_pk_fma_f32 | issue ds_read_b128.  Reported: MFMA wave's cycles per MFMA (s_memtime ticks and
// wall-clock via HIP events).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/coissue tools/ubench/coissue.hip && tools/ubench/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(int iters, float* sink, unsigned* clk) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 4096; i += 512) ((unsigned*)lds)[i] = i;
  __syncthreads();
  if (wave < 4) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 w = {0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, x = {0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u};
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime(), r0 = (unsigned)__builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc[a], 0, 0, 0);
    }
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime(), r1 = (unsigned)__builtin_amdgcn_s_memrealtime();
    float v = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (v == 12345.f) sink[0] = v;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    // tell the companions to stop
    __atomic_store_n((volatile unsigned*)(lds + 16380), 1u, __ATOMIC_RELAXED);
  } else {
    if (MODE == 0) return;
    float f[8];
    for (int a = 0; a < 8; ++a) f[a] = (float)(lane + a);
    f32x2 g[8];
    for (int a = 0; a < 8; ++a) g[a] = (f32x2){(float)lane, (float)a};
    u32x4 q = {0, 0, 0, 0};
    volatile unsigned* stop = (volatile unsigned*)(lds + 16380);
    while (*stop == 0u) {
#pragma unroll
      for (int rep = 0; rep < 16; ++rep) {
        if (MODE == 1) {
#pragma unroll
          for (int a = 0; a < 8; ++a) f[a] = __builtin_fmaf(f[a], 1.0001f, 0.5f);
        } else if (MODE == 2) {
#pragma unroll
          for (int a = 0; a < 8; ++a) g[a] = g[a] * (f32x2){1.0001f, 1.0001f} + (f32x2){0.5f, 0.5f};
        } else {
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const u32x4 t = *(const u32x4*)(lds + ((lane * 16 + a * 1024 + rep * 16) & 8191));
            q[0] ^= t[0]; q[1] ^= t[1];
          }
        }
      }
    }
    float v = 0.f;
    for (int a = 0; a < 8; ++a) v += f[a] + g[a][0] + g[a][1];
    if (v == 12345.f || q[0] == 0x12345u) sink[1] = v;
  }
}

template <int MODE>
void run(const char* name, float* sink, unsigned* clk) {
  const int iters = 200000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<MODE><<<256, 512>>>(iters, sink, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256, 512>>>(iters, sink, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned c[2];
  hipMemcpy(c, clk, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 8;
  const double ghz = (double)c[0] / (double)c[1] * 0.1;
  printf("%-44s %6.2f s_memtime ticks / MFMA   %6.2f ns / MFMA (events)   tick ratio %.3f 'GHz'   -> %.1f cycles / MFMA at 2.4 GHz\n", name, c[0] / n,
         ms * 1e6 / n, ghz, ms * 1e6 / n * 2.4);
}

int main() {
  float* sink; unsigned* clk;
  hipMalloc(&sink, 16); hipMalloc(&clk, 16);
  run<0>("MFMA wave alone on its SIMD", sink, clk);
  run<1>("+ a wave of back-to-back v_fma_f32", sink, clk);
  run<2>("+ a wave of back-to-back v_pk_fma_f32", sink, clk);
  run<3>("+ a wave of back-to-back ds_read_b128", sink, clk);
  run<0>("MFMA wave alone on its SIMD", sink, clk);
  return 0;
}
