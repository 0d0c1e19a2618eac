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

#ifndef BTX_G8_ROT
#define BTX_G8_ROT 0
#endif
#ifndef BTX_G8_CPRE
#define BTX_G8_CPRE 1  // per-channel constants of the store side filled in the prologue (beside the first DMAs' latency)
#endif
#ifndef BTX_G8_RPRE
#define BTX_G8_RPRE 1  // residual rows requested in front of the store side's first stage
#endif
#ifndef BTX_G8_HALF
#define BTX_G8_HALF 0  // bf16: all eight waves in lock-step, operands pipelined half a stage (one 32x32x16 K-step) ahead: slower
#endif
#ifndef BTX_G8_ALGKM
#define BTX_G8_ALGKM 1  // no LDS wait at the barrier behind part A
#endif
#ifndef BTX_G8_FLAT
#define BTX_G8_FLAT 1  // launch parameters read in one batch of scalar loads per section, the MC sample word by s_load
#endif
#ifndef BTX_G8_L2PF
#define BTX_G8_L2PF 1  // the store side touches the first activation stages of the workgroup that follows on this CU
#endif
#if BTX_G8_FLAT
#define BTX_G8_PARAMS BTX_SECTION_PARAMS_FLAT
#else
#define BTX_G8_PARAMS BTX_SECTION_PARAMS
#endif

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

// DIRECT (tuning builds, BTX_G8_DIRECT=1; ContractParams.ep_direct): the store side from the fragment registers
// (direct_epilogue, btx_epilogue.h: no LDS staging — 128 KB written and read back per workgroup — and about two thirds of
// the staged side's VALU work, but 32-byte pieces instead of whole lines).  Its own instantiation: with both store sides behind
// a run-time branch hipcc spills (btx_contract_taps.h).  Measured here, where nothing covers the store side, as in the
// tap-unrolled kernel: bit-identical and slower — 9.4k / 11.9k cycles + 1k of drain against 3.8k + 4.2k for the two stages of
// the staged side on 256 -> 1024 at 14x14, 5-25 % on the layer call (profiles/r04_gemm8_phase_timers.txt).  Not shipped.
template <int PREC, int KIND, bool DIRECT = false>
__global__ __launch_bounds__(512, 2) void contract_gemm8_kernel(const ContractParams) {
  static_assert(KIND == 1, "Flipout only: every wave issues the same number of weight DMAs per stage");
  BTX_G8_PARAMS(p, logical);
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
#if BTX_G8_FLAT
  if (p.sample_ptr) smp = sample_word_scalar(p.sample_ptr);
#else
  if (p.sample_ptr) smp = *p.sample_ptr;
#endif
#ifdef BTX_PT_TRACE
  const uint32_t tr_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
  const uint32_t tr_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
  uint32_t tr_t1 = 0, tr_t2 = 0;
  uint32_t tr_ep[4] = {0, 0, 0, 0};  // store side: [0] stage 1 done, [1] end, [2] / [3] around the body of stage 1 (BTX_EP_TRACE2)
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, w4 = wave & 3;
#if BTX_G8_CPRE
  const bool c_bias = p.mu_b != nullptr, c_aff = (p.ep_scale != nullptr) || (p.ep_shift != nullptr);
#endif

  // workgroup -> (pixel tile, group, n-tile pair): the pairs of a pixel tile are neighbours (its activations stay in one L2)
  uint32_t u_mtile, u_rem, u_group, u_pair;
  fdivmod((uint32_t)logical, p.fd_inner, (uint32_t)(p.pt_rtiles * p.groups), u_mtile, u_rem);
  fdivmod(u_rem, p.fd_rtiles, (uint32_t)p.pt_rtiles, u_group, u_pair);
  const int mtile = (int)u_mtile, group = (int)u_group, ntile0 = 2 * (int)u_pair;
  const int nstages = p.K / BK;  // host: >= 4
#if BTX_G8_CPRE
  // One workgroup per CU: nothing runs beside this workgroup's store side, so what the store needs and the K loop does not is
  // fetched here — the per-channel constants (bias parameters, BN scale / shift) of the two n-tiles, by waves 0 and 4: requested
  // in front of the first DMAs (they return first), turned into the store's [bias | bias delta | scale | shift] rows behind
  // the sign keys; their area lies behind the rings and the K loop's barriers publish it.
  EpRaw c_raw = {0.f, 0.f, 0.f, 1.f, 0.f};
  if ((c_bias || c_aff) && w4 == 0) c_raw = ep_load_constants(p, lane, ntile0 + half, group, c_bias, c_aff);
#endif

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
#if BTX_G8_CPRE
  {
    // an opaque copy of the wave test: with the same condition on both blocks the compiler moves the first instruction of the
    // softplus up to the loads — and their s_waitcnt vmcnt(0) in front of the first DMAs of waves 0 and 4
    int w4b = w4;
    asm volatile("" : "+s"(w4b));
    if ((c_bias || c_aff) && w4b == 0) {
      asm volatile("" : "+v"(c_raw.mu), "+v"(c_raw.rho), "+v"(c_raw.eps), "+v"(c_raw.sc), "+v"(c_raw.sh));  // (nor speculated)
      ep_store_constants<KIND>(p, rl, c_raw, (float*)(smem + LD::C_OFF + half * 1024), lane, ntile0 + half, group, c_bias);
    }
  }
#endif

  // More work for the idle issue slots in front of the first barrier (the first stages are still travelling): what the store
  // side would otherwise compute with the matrix pipe idle — the four hashed s_out words of the lane's fragment blocks and the
  // address of the line this thread touches for the workgroup that follows on the CU (BTX_G8_L2PF below).  Five registers
  // across the K loop.
  uint32_t wsh_pre[4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const uint32_t orow = (uint32_t)(mtile * TP + w4 * 64 + mi * 32 + l31) * (uint32_t)p.N + (uint32_t)(group * p.Ng + (ntile0 + half) * BN);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) wsh_pre[mi * 2 + ni] = btx_sign_word((orow + 32u * ni) >> 5, rl.kout_a, rl.kout_b) << (2 * h);
  }
  uint32_t l2pf_off = 0xffffffffu;
#if BTX_G8_L2PF
  {
    const int nxt = logical + (p.reverse ? -32 : 32);
    if (nxt >= 0 && nxt < p.lane_nwg) {  // wave-uniform
      uint32_t n_mtile, n_rem, n_group, n_pair;
      fdivmod((uint32_t)nxt, p.fd_inner, (uint32_t)(p.pt_rtiles * p.groups), n_mtile, n_rem);
      fdivmod(n_rem, p.fd_rtiles, (uint32_t)p.pt_rtiles, n_group, n_pair);
      const int mq = (int)n_mtile * TP + (tid & 255);
      if (mq < p.M) {
        const size_t off = ((size_t)pix_in(mq) * (size_t)p.C + (size_t)((int)n_group * p.Cg)) * ESZ + (size_t)(tid >> 8) * 128u;
        if (off + 4 <= (size_t)p.x_bytes) l2pf_off = (uint32_t)off;
      }
    }
  }
#endif

  // (pinned here: left alone the compiler sinks both computations back to their use behind the K loop)
  asm volatile("" : "+v"(wsh_pre[0]), "+v"(wsh_pre[1]), "+v"(wsh_pre[2]), "+v"(wsh_pre[3]), "+v"(l2pf_off));

  f32x16 accm[2][2], accd[2][2];
  using Frag = StageFragT<2>;
  auto load_frag = [&](Frag& f, int s) __attribute__((always_inline)) {
    const int slot = s & (LD::RD - 1);
    const unsigned char* as = smem + LD::A_OFF + slot * LD::A_STAGE;
    const unsigned char* ss = smem + LD::S_OFF + slot * LD::S_STAGE;
    const unsigned char* ws = smem + LD::W_OFF + slot * LD::W_STAGE + half * DW_STAGE;
    if constexpr (BTX_PT_ABL & 2) {  // measurement builds: no fragment reads
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) f.a[kk][mi] = (u32x4){(uint32_t)slot, 5u, 1u, 4u};
        f.wm[kk][0] = f.wm[kk][1] = (u32x4){7u, 7u, 1u, (uint32_t)slot};
      }
      f.sw[0] = f.sw[1] = (uint32_t)slot;
      return;
    }
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
#ifdef BTX_PT_TRACE
  tr_t1 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
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
  // and then leave it idle together.  A stage is therefore two parts — A: this stage's delta fragments, the requests of stage
  // s+3, the next (bf16) or this (f32 / split-bf16) stage's fragments; B: the MFMAs — and waves 4-7 run one part behind waves
  // 0-3: on every SIMD one wave multiplies while the other loads (the K-group scheme of btx_contract_taps.h).
  //   BTX_G8_ROT=1: ONE workgroup barrier per stage — one instruction stream A(s) B(s) for both groups, the barrier behind B
  //   for waves 0-3 and behind A for waves 4-7, so that between two barriers the first group runs A(s) B(s) and the second
  //   B(s-1) A(s): both touch the LDS rings exactly as in lock-step, the order inside an interval is free.
  //   BTX_G8_ROT=0: a barrier behind each part.
  // What a barrier has to guarantee (bf16, BTX_G8_ALGKM): stage s+2 has landed (requested an interval ago: at most this
  // interval's four DMAs stay in flight) and nobody still reads the slot that is requested next — slot s, last read by
  // delta(s).  Part A therefore reads delta(s) FIRST and waits for it in front of its fragment reads (the requests and the sign
  // words are issued in between); the prefetched fragments stay in flight across the barrier — LDS returns in order, so they
  // are back when the next part A has its delta, a whole stage before their slot is requested again.  That wait is the
  // compiler's builtin: its own wait insertion sees it (an s_waitcnt inside inline asm it does not) and does not make the next
  // stage's MFMAs wait for reads that completed there.  f32 / split-bf16 (a stage multiplies the fragments it read in its own
  // part A): every barrier waits for all LDS reads.
  // Phase timers + ablation builds (tools/gpu_diag.py g8trace, -DBTX_PT_ABL): MFMAs + barriers alone 1280 cycles per stage
  // (1030 = the matrix pipe), s_in masks +200, DMA +70, sign words +55: 1480.
  constexpr bool HALFP = (BTX_G8_HALF != 0) && (PREC == 1);
  if constexpr (HALFP) {
    // BTX_G8_HALF (measured, not shipped: 1 950 - 2 040 cycles per stage against 1 500, bit-identical, cfg5 -1 %) — the other
    // way to keep the matrix pipe fed: ALL operands of a K-step (activation, mean AND delta fragments:
    // 26 registers) are read half a stage before they are multiplied, into two alternating sets (52 registers against the 84 of
    // two full fragment sets + the delta set), so that every wave has MFMAs to issue the moment a barrier releases it and its
    // reads ride under them; all eight waves in lock-step, ONE barrier per stage, both waves of a SIMD multiply at the same time
    // (a single wave issues a 32x32x16 MFMA every 40 cycles, two fill each other's gaps).  Same K order per accumulator.
    struct HalfFrag { u32x4 a[2], wm[2], wd[2]; uint32_t sw[2]; };
    auto load_half = [&](HalfFrag& f, int s, int kk) __attribute__((always_inline)) {
      const int slot = s & (LD::RD - 1);
      const unsigned char* as = smem + LD::A_OFF + slot * LD::A_STAGE;
      const unsigned char* ss = smem + LD::S_OFF + slot * LD::S_STAGE;
      const unsigned char* ws = smem + LD::W_OFF + slot * LD::W_STAGE + half * DW_STAGE;
      const int row = 2 * kk + h;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        f.a[mi] = *(const u32x4*)(as + (w4 * 64 + mi * 32 + l31) * 64 + ((row ^ ((l31 >> 2) & 3)) * 16));
        f.sw[mi] = *(const uint32_t*)(ss + (w4 * 64 + mi * 32 + l31) * 4);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        f.wm[ni] = *(const u32x4*)(ws + (row * BN + ni * 32 + l31) * 16);
        f.wd[ni] = *(const u32x4*)(ws + NG * BN * 16 + (row * BN + ni * 32 + l31) * 16);
      }
    };
    auto mma_half = [&](HalfFrag& f, int kk, auto zero_tag) __attribute__((always_inline)) {
      constexpr bool ZERO = decltype(zero_tag)::value;
      const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          accm[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.wm[ni]), __builtin_bit_cast(bf16x8, f.a[mi]),
                                                                 ZERO ? zc : accm[mi][ni], 0, 0, 0);
      const int row = 2 * kk + h;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const uint32_t swr = f.sw[mi] << (4 * row);
#pragma unroll
        for (int d = 0; d < 4; ++d) f.a[mi][d] ^= ((swr << d) & 0x80008000u);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          accd[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.wd[ni]), __builtin_bit_cast(bf16x8, f.a[mi]),
                                                                 ZERO ? zc : accd[mi][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    // a barrier guarantees: stage s+2 landed (at most this stage's four DMAs in flight) and every read of the slot that is
    // requested next (stage s: read in this stage's first half, multiplied in its second) is back; the K-step read ahead for
    // the next stage belongs to a slot that is requested two stages later
    auto meet_h = [&](int s) __attribute__((always_inline)) {
      if (s + 3 < nstages) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    };
    HalfFrag h0, h1;
    // (the prologue's load_frag(fa, 0) is dead code here: fa is not used)
    load_half(h0, 0, 0);
    using TTh = std::true_type;
    using FFh = std::false_type;
    for (int s = 0; s < nstages; ++s) {
      if (s + 3 < nstages) { issue(s + 3); write_signs(s + 3); }  // wave-uniform
      load_half(h1, s, 1);
      if (s == 0) mma_half(h0, 0, TTh{}); else mma_half(h0, 0, FFh{});
      // (unconditional — behind the last stage it reads a slot nobody writes any more and nobody multiplies: under a
      // condition the compiler's wait insertion merges the two paths and makes the MFMAs below wait for THESE reads)
      load_half(h0, s + 1, 0);
      mma_half(h1, 1, FFh{});
      meet_h(s);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the store side's staging area lies over the rings
  } else {
  constexpr bool RELAX = PF && (BTX_G8_ALGKM != 0);
  DeltaFrag df;
  auto a_body = [&](int s, Frag& f) __attribute__((always_inline)) {  // f: bf16 the set of stage s+1, else of stage s
    load_delta<KIND>(df, smem + LD::W_OFF + (s & (LD::RD - 1)) * LD::W_STAGE + half * DW_STAGE, l31, h);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 3 < nstages) {  // wave-uniform
      if constexpr (!(BTX_PT_ABL & 4)) issue(s + 3);
      if constexpr (!(BTX_PT_ABL & 512)) write_signs(s + 3);
    }
    if constexpr (RELAX) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PF) { if (s + 1 < nstages) load_frag(f, s + 1); }
    else load_frag(f, s);
  };
  auto b_body = [&](Frag& cur, auto first_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    // the parts stay apart in the instruction stream: loads hoisted over the MFMAs in front of them would need a third
    // fragment set (the compiler renames, then spills)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FIRST && PREC == 1) stage_mma<PREC, KIND, 2, 2, true>(cur, df, accm, accd, l31, h);
    else stage_mma<PREC, KIND>(cur, df, accm, accd, l31, h);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto meet = [&](int s) __attribute__((always_inline)) {  // end of the interval in which stage s+3 was requested (or not)
    if constexpr (BTX_PT_ABL & 1024) return;  // measurement builds: no barriers in the K loop (timing only, results garbage)
    if constexpr (RELAX) {
      if (s + 3 < nstages) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      if (s + 3 < nstages) end_stage<4>(); else end_stage<0>();
    }
  };
  using TT = std::true_type;
  using FF = std::false_type;
#if BTX_G8_ROT
  auto meet0 = [&](int s) __attribute__((always_inline)) { if (half == 0) meet(s); };
  auto meet1 = [&](int s) __attribute__((always_inline)) { if (half == 1) meet(s); };
  if constexpr (PF) {
    a_body(0, fb); meet1(0); b_body(fa, TT{}); meet0(0);
    int s = 1;
    for (; s + 1 < nstages; s += 2) {
      a_body(s, fa); meet1(s); b_body(fb, FF{}); meet0(s);
      a_body(s + 1, fb); meet1(s + 1); b_body(fa, FF{}); meet0(s + 1);
    }
    if (s < nstages) { a_body(s, fa); meet1(s); b_body(fb, FF{}); meet0(s); }
  } else {
    for (int s = 0; s < nstages; ++s) { a_body(s, fa); meet1(s); b_body(fa, FF{}); meet0(s); }
  }
  end_stage<0>();  // waves 4-7 behind their last MFMAs, waves 0-3 one interval later
#else
  if (half == 1) asm volatile("s_barrier" ::: "memory");
  if constexpr (PF) {
    a_body(0, fb); meet(0); b_body(fa, TT{}); meet(0);
    int s = 1;
    for (; s + 1 < nstages; s += 2) {
      a_body(s, fa); meet(s); b_body(fb, FF{}); meet(s);
      a_body(s + 1, fb); meet(s + 1); b_body(fa, FF{}); meet(s + 1);
    }
    if (s < nstages) { a_body(s, fa); meet(s); b_body(fb, FF{}); meet(s); }
  } else {
    for (int s = 0; s < nstages; ++s) { a_body(s, fa); meet(s); b_body(fa, FF{}); meet(s); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (half == 0) asm volatile("s_barrier" ::: "memory");
#endif

  }
#ifdef BTX_PT_TRACE
  tr_t2 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
  // =================== store side (btx_epilogue.h): every wave stages its own 64 x 64 tile ============================
  {
    BTX_G8_PARAMS(pe, logical2);
#if BTX_G8_FLAT
    // what the store side reads, fetched in the same batch of scalar loads (else: one round trip per branch that tests one)
    asm volatile("" ::"s"(pe.N), "s"(pe.Ng), "s"(pe.M), "s"(pe.ksplits), "s"(pe.mu_b), "s"(pe.ep_scale), "s"(pe.ep_shift),
                 "s"(pe.sign_out), "s"(pe.out_bf16), "s"(pe.ep_relu));
#endif
    // The workgroup that follows this one on the CU is the one 32 slots further on in the XCD's range of logical ids (32 CUs
    // per XCD, one workgroup each; xcd_logical()).  Its first wait is the HBM round trip of its first activation stages —
    // 3.9k .. 7.3k cycles of prologue in the phase timers, longest on the layers that keep HBM busy.  Thread t touches one
    // 128-byte line of that tile (pixel t % 256, stages 0-1 | 2-3; address from the prologue), so the rows are in this XCD's L2
    // when they are asked for: prologue 5.0k -> 3.1k cycles on 256 -> 1024 at 14x14.  Issued from the store side's hook, BEHIND
    // the residual requests: in front of them stage 1 grew by what the prologue saved.
    uint32_t l2pf = 0;
    auto touch_next = [&]() __attribute__((always_inline)) {
      if (l2pf_off != 0xffffffffu) l2pf = *(const uint32_t*)((const unsigned char*)pe.x + l2pf_off);
    };
    const bool has_bias = pe.mu_b != nullptr;
    const bool has_aff = (pe.ep_scale != nullptr) || (pe.ep_shift != nullptr);
    float* ba = (float*)(smem + LD::C_OFF + half * 1024);
    if constexpr (DIRECT) {
      static_assert(BTX_G8_CPRE != 0, "the direct store side reads the constants the prologue wrote");
      using OUT = typename std::conditional<PREC == 1, __bf16, float>::type;  // host: the output has the activation dtype
      touch_next();
      if (!(has_bias || has_aff)) {  // identities: the direct store side always applies its constants (wave-uniform)
        if (w4 == 0) { ba[lane] = 0.f; ba[BN + lane] = 0.f; ba[2 * BN + lane] = 1.f; ba[3 * BN + lane] = 0.f; }
        __syncthreads();
      }
      int lane_o = lane, tid_o = tid;  // opaque copies: the lane-dependent addresses are not computed in front of the K loop
      asm volatile("" : "+v"(lane_o), "+v"(tid_o));
      uint32_t gp[2];
      bool gok[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int pl = w4 * 64 + mi * 32 + (lane_o & 31);
        gp[mi] = (uint32_t)(mtile * TP + pl);
        gok[mi] = (int)gp[mi] < pe.M;
      }
#ifdef BTX_PT_TRACE
      tr_ep[0] = tr_t2;  // (no stage split: the whole store side is reported as stage 2)
#endif
      direct_epilogue<KIND, OUT, true>(pe, rl, accm, accd, ba, tid_o, lane_o, ntile0 + half, group, gp, gok);
      asm volatile("" ::"v"(l2pf));
    } else {
#if !BTX_G8_CPRE
    if (has_bias || has_aff) {
      if (w4 == 0) ep_fill_constants<KIND>(pe, rl, ba, lane, ntile0 + half, group, has_bias, has_aff);  // waves 0 and 4
      __syncthreads();
    }
#endif
    const uint32_t m0 = (uint32_t)mtile * (uint32_t)TP;
    const PixContig pm = {m0, min(TP, pe.M - (int)m0)};
#ifdef BTX_PT_TRACE
    uint32_t* const ep_tr = tr_ep;
#else
    uint32_t* const ep_tr = nullptr;
#endif
    staged_epilogue_pm<KIND, 8, PixContig, BTX_G8_RPRE != 0, decltype(touch_next)>(pe, rl, accm, accd, smem, tid, wave, lane, ntile0 + half,
                                                                                   group, 0, pm, ep_tr, w4, true, ba, touch_next, wsh_pre);
    asm volatile("" ::"v"(l2pf));  // the touch is "used"
    }
  }
#ifdef BTX_PT_TRACE
  if (p.trace) {  // per wave: prologue | K loop | 100-MHz ticks | store stage 1 | store stage 2 + drain | total | start | HW_ID
    const uint32_t tr_t3s = (uint32_t)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t tr_t3 = (uint32_t)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
      uint32_t* tr = (uint32_t*)p.trace + (size_t)(blockIdx.x * 8 + wave) * 8;
      tr[0] = tr_t1 - tr_t0; tr[1] = tr_t2 - tr_t1; tr[2] = (uint32_t)__builtin_amdgcn_s_memrealtime() - tr_r0;
      tr[3] = tr_ep[0] - tr_t2; tr[4] = tr_t3s - tr_ep[0]; tr[5] = tr_t3 - tr_t0;
#ifdef BTX_EP_TRACE2  // stage 1 split instead: head (parameters, residual requests, touch) | body | LDS drain
      tr[0] = tr_ep[2] - tr_t2; tr[1] = tr_ep[3] - tr_ep[2]; tr[3] = tr_ep[0] - tr_ep[3];
#endif
      tr[6] = tr_t0;
      tr[7] = (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xffffu) |
              (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 16);  // HW_ID[15:0] | XCC_ID << 16
    }
  }
#endif
}

template <int PREC>
static int launch_contract_gemm8_impl(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  if (kind != 1) return -3;
  auto kfn = contract_gemm8_kernel<PREC, 1, false>;
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
  auto kfd = contract_gemm8_kernel<PREC, 1, true>;  // the direct store side: measured, not shipped
#endif
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G8Lds::BYTES);
    if (e != hipSuccess) return (int)e;
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
    e = hipFuncSetAttribute((const void*)kfd, hipFuncAttributeMaxDynamicSharedMemorySize, G8Lds::BYTES);
    if (e != hipSuccess) return (int)e;
#endif
    attr_done = true;
  }
  int rc = launch_presample_impl<PREC>(kind, p, st);
  if (rc) return rc;
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
  if (p.ep_direct) {
    hipLaunchKernelGGL(kfd, dim3(nwg), dim3(512), G8Lds::BYTES, st, p);
    return (int)hipGetLastError();
  }
#endif
  hipLaunchKernelGGL(kfn, dim3(nwg), dim3(512), G8Lds::BYTES, st, p);
  return (int)hipGetLastError();
}

}  // namespace btx
