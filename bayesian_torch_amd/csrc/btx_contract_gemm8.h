// btx_contract_gemm8.h — pointwise Flipout contractions with a long K (1x1x1 convolutions without padding, any stride, and
// Linear layers, K >= 128: the "reduce" and 3x3-neighbour 1x1 convolutions of a ResNet50 bottleneck, reference
// models/deterministic/resnet_large.py:85-105, layers/flipout_layers/conv_flipout.py:376-417) as ONE 8-wave workgroup per CU
// with the K loop of the tap-unrolled kernel (gfx950; bf16, and the f32 / split-bf16 precisions on f32 activations).
//
// contract_dma_kernel runs these GEMMs at 0.23-0.29 of the MFMA peak: its 256-pixel tile needs 16 KB of activations per
// K-stage, rings of three is all that fits twice into a CU's LDS, so a stage reads its own fragments (LDS latency exposed every
// stage), waits on a run-time vmcnt and has its DMAs two stages ahead of an HBM latency of three.  A 1x1 convolution has no
// patch to reuse, so the room has to come from the occupancy: one workgroup per CU, 160 KB of LDS —
//
//   * tile 256 pixels x 128 channels: waves 0-3 own n-tile 2j, waves 4-7 n-tile 2j+1, on the SAME pixels (wave = 64 pixels x
//     64 channels, two accumulator sets: 128 registers).  One 16-KB activation stage serves both halves: half the L2 -> LDS
//     bytes per FLOP of the 64-channel tile;
//   * rings of FOUR for activations (64 KB), weight tiles (2 x 8 KB per stage: 64 KB) and s_in words: stage s+3 is requested
//     while stage s multiplies and stage s+1's fragments are read — the tap-unrolled kernel's loop: fragments double-buffered in
//     registers, ONE `s_waitcnt vmcnt(4)` immediate per stage (each wave issues exactly 2 + 2 DMA instructions per stage);
//   * staged store side of btx_epilogue.h (8 x 17 KB: the rings are dead by then), per-half constants.
//
// Same K order per output element as every other variant: bit-identical to contract_dma_kernel on the same launch.
// Host-checked (btx_api.hip): Flipout, bf16 activations and MFMA, pointwise geometry, K % 32 == 0, K >= 4 stages, an even
// number of whole n-tiles per group, one K split.  ContractParams: pt_rtiles = n-tile pairs per group.
#pragma once
#include <type_traits>
#include "btx_contract.h"
#include "btx_contract_dma.h"
#include "btx_contract_taps.h"
#include "btx_epilogue.h"
#include "btx_mma.h"
#include "btx_presample.h"

namespace btx {

struct G8Lds {
  static constexpr int TP = 256, RD = 4;
  static constexpr int A_STAGE = NG * TP * 16;   // 16384
  static constexpr int W_STAGE = 2 * DW_STAGE;   // 16384: [half][mu 4 KiB | delta 4 KiB]
  static constexpr int S_STAGE = TP * 4;
  static constexpr int A_OFF = 0;
  static constexpr int W_OFF = A_OFF + RD * A_STAGE;
  static constexpr int S_OFF = W_OFF + RD * W_STAGE;
  static constexpr int MAIN = S_OFF + RD * S_STAGE;         // 135168
  static constexpr int C_OFF = 8 * PT_EP_WAVE;              // 139264: constants of the two n-tiles behind the staging areas
  static constexpr int BYTES = C_OFF + 2 * 1024;            // 141312
};
static_assert(G8Lds::MAIN <= G8Lds::C_OFF && G8Lds::BYTES <= 163840, "LDS budget");

template <int PREC, int KIND>
__global__ __launch_bounds__(512, 2) void contract_gemm8_kernel(const ContractParams) {
  static_assert(KIND == 1, "Flipout only: every wave issues the same number of weight DMAs per stage");
  BTX_SECTION_PARAMS(p, logical);
  using LD = G8Lds;
  using ACT = typename std::conditional<PREC == 1, __bf16, float>::type;
  constexpr int G = (PREC == 1) ? 8 : 4, BK = NG * G, TP = LD::TP;
  constexpr uint32_t ESZ = (uint32_t)sizeof(ACT);
  // bf16: the NEXT stage's fragments are read while this one multiplies (two fragment sets).  f32 / split-bf16: a stage reads
  // its own fragments in its first part (the split's hi / lo halves take the registers of the second set) — their latency hides
  // behind the other wave group's MFMAs either way
  constexpr bool PF = (PREC == 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t smp = p.sample;
  if (p.sample_ptr) smp = *p.sample_ptr;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, w4 = wave & 3;

  // workgroup -> (pixel tile, group, n-tile pair): the pairs of a pixel tile are neighbours (its activations stay in one L2)
  uint32_t u_mtile, u_rem, u_group, u_pair;
  fdivmod((uint32_t)logical, p.fd_inner, (uint32_t)(p.pt_rtiles * p.groups), u_mtile, u_rem);
  fdivmod(u_rem, p.fd_rtiles, (uint32_t)p.pt_rtiles, u_group, u_pair);
  const int mtile = (int)u_mtile, group = (int)u_group, ntile0 = 2 * (int)u_pair;
  const int nstages = p.K / BK;  // host: >= 4

  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

  // ---- weight loader: the stage's 16 rows of 1 KiB — (half 0|1) x (mu | delta) x (granule row 0..3) — rows 2w, 2w+1 to wave w
  uint32_t w_src[2];
  int w_dst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 2 * wave + j, hf = r >> 3, set = (r >> 2) & 1, row = r & 3;
    w_src[j] = (uint32_t)(group * p.ntiles + ntile0 + hf) * (uint32_t)(p.K / G) * 1024u + (uint32_t)row * 1024u + (uint32_t)lane * 16u +
               (set ? p.wt_delta_off : 0u);
    w_dst[j] = LD::W_OFF + hf * DW_STAGE + set * 4096 + row * 1024;
  }
  // output pixel m -> input pixel: itself for stride 1 (a plain GEMM), else the strided position (1x1 down-sampling
  // convolutions: no padding, so every output pixel has its input pixel)
  auto pix_in = [&](int m) __attribute__((always_inline)) -> uint32_t {
    if (p.pointwise) return (uint32_t)m;
    uint32_t t, t2, ow, oh, od, nb;
    fdivmod((uint32_t)m, p.fd_Wo, (uint32_t)p.Wo, t, ow);
    fdivmod(t, p.fd_Ho, (uint32_t)p.Ho, t2, oh);
    fdivmod(t2, p.fd_Do, (uint32_t)p.Do, nb, od);
    return ((nb * (uint32_t)p.D + od * (uint32_t)p.sd) * (uint32_t)p.H + oh * (uint32_t)p.sh) * (uint32_t)p.W + ow * (uint32_t)p.sw;
  };
  // ---- activation loader: the stage's 16 pieces of 1 KiB (16 pixels x 64 B) — pieces 2w, 2w+1 to wave w; source-side XOR
  //      swizzle as in btx_contract_dma.h; out-of-range pixels read zeros
  const int g_lane = (lane & 3) ^ ((lane >> 4) & 3);
  uint32_t a_src[2];
  int a_dst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int piece = 2 * wave + j;
    const int mq = mtile * TP + piece * 16 + (lane >> 2);
    a_src[j] = mq < p.M ? (pix_in(mq) * (uint32_t)p.C + (uint32_t)(group * p.Cg + G * g_lane)) * ESZ : DMA_OOB;
    a_dst[j] = LD::A_OFF + piece * 1024;
  }
  auto issue = [&](int s) __attribute__((always_inline)) {  // all L2 -> LDS traffic of stage s: 4 DMA instructions per wave
    const int slot = s & (LD::RD - 1);
    const uint32_t wo = (uint32_t)s * (uint32_t)NG * 1024u, ao = (uint32_t)(s * BK) * ESZ;
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(wt_rsrc, w_src[j] + wo, smem + w_dst[j] + slot * LD::W_STAGE);
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(x_rsrc, a_src[j] == DMA_OOB ? DMA_OOB : a_src[j] + ao, smem + a_dst[j] + slot * LD::A_STAGE);
  };
  issue(0);
  issue(1);
  issue(2);

  // ---- the sign keys of this (sample, layer), derived while the first DMAs travel
  RngLive rl = {smp, p.kin_a, p.kin_b, p.kout_a, p.kout_b};
  if (p.sample_ptr || p.lanes > 1) {
    rl.sample = __builtin_amdgcn_readfirstlane(smp);
    const uint32_t si = p.swap_signs ? 3u : 2u, so = p.swap_signs ? 2u : 3u;
    const BtxPhilox4 ki = btx_philox4x32_10(0u, rl.sample, p.layer, si, p.seed_lo, p.seed_hi);
    const BtxPhilox4 ko = btx_philox4x32_10(0u, rl.sample, p.layer, so, p.seed_lo, p.seed_hi);
    rl.kin_a = __builtin_amdgcn_readfirstlane(ki.x[0]); rl.kin_b = __builtin_amdgcn_readfirstlane(ki.x[1]);
    rl.kout_a = __builtin_amdgcn_readfirstlane(ko.x[0]); rl.kout_b = __builtin_amdgcn_readfirstlane(ko.x[1]);
  }
  // ---- s_in words: thread t < 256 hashes the word of pixel t for a stage (32 channels = one word)
  const int m_own = mtile * TP + (tid & 255);
  const uint32_t sg_off = pix_in(m_own < p.M ? m_own : 0) * (uint32_t)p.C + (uint32_t)(group * p.Cg);
  auto write_signs = [&](int s) __attribute__((always_inline)) {
    if (half == 0) {  // wave-uniform
      const uint32_t off = sg_off + (uint32_t)(s * BK);
      uint32_t w = p.sign_in ? sign_word_explicit(p.sign_in, off, p.x_bytes / ESZ) : btx_sign_word(off >> 5, rl.kin_a, rl.kin_b);
      if constexpr (G == 4) w <<= 8 * ((off >> 4) & 1);
      *(uint32_t*)(smem + LD::S_OFF + (s & (LD::RD - 1)) * LD::S_STAGE + (tid & 255) * 4) = w;
    }
  };
  write_signs(0);
  write_signs(1);
  write_signs(2);

  f32x16 accm[2][2], accd[2][2];
  using Frag = StageFragT<2>;
  auto load_frag = [&](Frag& f, int s) __attribute__((always_inline)) {
    const int slot = s & (LD::RD - 1);
    const unsigned char* as = smem + LD::A_OFF + slot * LD::A_STAGE;
    const unsigned char* ss = smem + LD::S_OFF + slot * LD::S_STAGE;
    const unsigned char* ws = smem + LD::W_OFF + slot * LD::W_STAGE + half * DW_STAGE;
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      const int row = 2 * kk + h;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        f.a[kk][mi] = *(const u32x4*)(as + (w4 * 64 + mi * 32 + l31) * 64 + ((row ^ ((l31 >> 2) & 3)) * 16));
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) f.wm[kk][ni] = *(const u32x4*)(ws + (row * BN + ni * 32 + l31) * 16);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) f.sw[mi] = *(const uint32_t*)(ss + (w4 * 64 + mi * 32 + l31) * 4);
  };

  // stages 0 and 1 landed (stage 2's four DMAs may be in flight); meet
  asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  Frag fa, fb;
  if constexpr (PF) load_frag(fa, 0);
  else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }
  }
  // Eight waves marching in lock-step request, read and multiply together: the two waves of a SIMD queue for the matrix pipe
  // and then leave it idle together.  A stage is therefore two parts with a workgroup barrier after each — A: requests of
  // stage s+3, this stage's delta fragments, the next (bf16) or this (f32 / split-bf16) stage's fragments; B: the MFMAs — and
  // waves 4-7 run one part behind waves 0-3 (one extra barrier in front of their first part, one behind the last part of waves
  // 0-3): on every SIMD one wave multiplies while the other loads (the K-group scheme of btx_contract_taps.h; measured against
  // all eight in lock-step: +0.4 .. 2 % on cfg5).  A wave that requested in part A leaves those four DMAs in flight at both of
  // the stage's barriers; stage s+2 (requested a stage ago by BOTH groups, the later one three parts before its first reader)
  // has landed.
  DeltaFrag df;
  auto part_a = [&](int s, Frag& f) __attribute__((always_inline)) {  // f: bf16 the set of stage s+1, else of stage s
    const bool more = s + 3 < nstages;  // wave-uniform
    if (more) { issue(s + 3); write_signs(s + 3); }
    load_delta<KIND>(df, smem + LD::W_OFF + (s & (LD::RD - 1)) * LD::W_STAGE + half * DW_STAGE, l31, h);
    if constexpr (PF) { if (s + 1 < nstages) load_frag(f, s + 1); }
    else load_frag(f, s);
    if (more) end_stage<4>(); else end_stage<0>();
  };
  auto part_b = [&](int s, Frag& cur, auto first_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    if constexpr (FIRST && PREC == 1) stage_mma<PREC, KIND, 2, 2, true>(cur, df, accm, accd, l31, h);
    else stage_mma<PREC, KIND>(cur, df, accm, accd, l31, h);
    if (s + 3 < nstages) end_stage<4>(); else end_stage<0>();
  };
  if (half == 1) asm volatile("s_barrier" ::: "memory");
  if constexpr (PF) {
    part_a(0, fb);
    part_b(0, fa, std::true_type{});
    int s = 1;
    for (; s + 1 < nstages; s += 2) {
      part_a(s, fa);
      part_b(s, fb, std::false_type{});
      part_a(s + 1, fb);
      part_b(s + 1, fa, std::false_type{});
    }
    if (s < nstages) { part_a(s, fa); part_b(s, fb, std::false_type{}); }
  } else {
    for (int s = 0; s < nstages; ++s) {
      part_a(s, fa);
      part_b(s, fa, std::false_type{});
    }
  }
  if (half == 0) asm volatile("s_barrier" ::: "memory");

  // =================== store side (btx_epilogue.h): every wave stages its own 64 x 64 tile ============================
  {
    BTX_SECTION_PARAMS(pe, logical2);
    const bool has_bias = pe.mu_b != nullptr;
    const bool has_aff = (pe.ep_scale != nullptr) || (pe.ep_shift != nullptr);
    float* ba = (float*)(smem + LD::C_OFF + half * 1024);
    if (has_bias || has_aff) {
      if (w4 == 0) ep_fill_constants<KIND>(pe, rl, ba, lane, ntile0 + half, group, has_bias, has_aff);  // waves 0 and 4
      __syncthreads();
    }
    const uint32_t m0 = (uint32_t)mtile * (uint32_t)TP;
    const PixContig pm = {m0, min(TP, pe.M - (int)m0)};
    staged_epilogue_pm<KIND, 8, PixContig>(pe, rl, accm, accd, smem, tid, wave, lane, ntile0 + half, group, 0, pm, nullptr, w4, true, ba);
  }
}

template <int PREC>
static int launch_contract_gemm8_impl(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  if (kind != 1) return -3;
  auto kfn = contract_gemm8_kernel<PREC, 1>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G8Lds::BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  int rc = launch_presample_impl<PREC>(kind, p, st);
  if (rc) return rc;
  hipLaunchKernelGGL(kfn, dim3(nwg), dim3(512), G8Lds::BYTES, st, p);
  return (int)hipGetLastError();
}

}  // namespace btx
