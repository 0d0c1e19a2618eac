// btx_contract_stempool.h — the row-fused stem contraction with the 3x3 / stride-2 / pad-1 max-pool that follows it in a
// ResNet folded into its store (bf16, gfx950).
//
// What the un-fused chain costs on ResNet18, batch 64: the stem kernel (btx_contract_stem.h) writes 112x112x64 per
// image (103 MB), the pool kernel reads them back and writes 56x56x64 (26 MB); a stem tile is 7 K-stages between a
// patch fetch and a 28-KB store, so its workgroups spend more time waiting than multiplying.  Here one 8-wave
// workgroup per CU walks down a BAND of an image, two conv rows (a "half tile") at a time:
//
//   * the layer's pre-sampled weight tiles (all K-stages: 56 KiB for a 7x7 Flipout stem) are fetched ONCE and stay in
//     LDS: the K loop of a half tile has no weight ring and no DMA wait;
//   * the two groups of four waves alternate roles.  In phase f one group multiplies half tile f (its K loop) while
//     the other runs the store side of half tile f-1 from its accumulator registers — per SIMD one wave feeds the matrix
//     pipe while the other does the VALU / LDS work of the epilogue, and the next phase they swap.  (A first version
//     that had all eight waves multiply, then all eight store, spent 30 % of its time in the K loops: 105 us per launch.)
//     A phase is three steps with a workgroup barrier after each.  (Measured alternative, git history: barriers among the
//     four waves of a group through LDS arrival counters and a single s_barrier per phase, 99 instead of 103 us per stem
//     call — dropped: spin-waits give up the forward-progress guarantee of s_barrier for 4 %.)
//   * Flipout: x * s_in of the half tile is written ONCE per input element into a second copy of the patch (step 1 of
//     the K role) and the K loop runs as two passes — mean pass: raw patch x mu tiles, delta pass: signed copy x delta
//     tiles — instead of masking every activation fragment (each input element sits in ~12 fragments of a 7x7 / stride-2
//     stem; the second version was bound by the VALU port: ~100 VALU per stage and wave for 16 MFMAs);
//     K role: sign copy + mean pass | delta stages 0-3 | delta stages 4-6; store role: stage channels 0-31 | stage
//     channels 32-63 | pool all 64 and prepare the next carry row;
//   * the input rows of half tile f+1 are fetched (one contiguous byte range, LDS-DMA) by the K group and their s_in
//     words hashed by the store group during phase f;
//   * the epilogue (bias, Flipout combine, BN affine, bf16 rounding — the same arithmetic, in the same order, as
//     btx_epilogue.h) writes the half tile's conv rows r0, r1 into LDS (128 bytes per pixel, XOR-swizzled 16-byte
//     chunks); pooled row P0+u-1 = max3x3 over (carry, r0) of half tile u, carry = max(r0, r1) of half tile u-1 in a
//     third LDS row (c0 = 2*P0 - 1 is the band's first conv row, r0 = c0 + 2u).  A band of PB pooled rows needs
//     2*PB + 1 conv rows: PB half tiles and a closing one-row half tile.  ReLU is applied after the max (both are
//     monotonic, bf16 rounding too): the pool then compares the bf16 bit patterns as signed 16-bit integers — equal to
//     the float order whenever the maximum is non-negative, and a negative maximum becomes 0 either way.  Values are
//     bit-identical to the two-launch chain.
//
// HBM traffic of the stem + pool on the ResNet18 shape: 26 (input) + 26 (pooled output) MB instead of 26 + 103 + 103 + 26.
#pragma once
#include <type_traits>
#include "btx_contract.h"
#include "btx_contract_dma.h"
#include "btx_epilogue.h"
#include "btx_mma.h"
#include "btx_presample.h"

namespace btx {

#ifndef BTX_STEM_STEPS
#define BTX_STEM_STEPS 1  // step layout of a Flipout K phase (run_k): 0 = round 2's (sign copy + mean) | delta 0-3 | delta 4-6,
#endif                    // 1 = sign copy | mean | delta
#ifndef BTX_SP_ABL
#define BTX_SP_ABL 0  // measurement-only ablation bits (wrong results; tools/r06/e6.sh, with -DBTX_STEM_PASS3=0 -DBTX_STEM_NB=2): 1 no MFMA in
#endif                // the two-register-set passes (run_pass), 2 no fragment reads after their first stage, 4 the store role does nothing
#ifndef BTX_STEM_NB
#define BTX_STEM_NB 1  // workgroup barriers INSIDE a phase (besides its closing one).  2 (rounds 2-5): sign copy | mean | delta against
#endif                 // stage 0-31 | stage 32-63 | pool.  1 (round 6): [copy +] first part | rest against staging | pool — the barrier between
                       // the two staging halves guarded nothing (the pool is what reads the rows), and with the store role's steps now 2.1k |
                       // 1.6k cycles (Reparameterization) the three-way split left the K role's first third (patch DMA issue + stages) alone
                       // on the critical path
#ifndef BTX_STEM_PRIO
#define BTX_STEM_PRIO 2
#endif
#ifndef BTX_STEM_PASS3
#define BTX_STEM_PASS3 1  // fragments two stages ahead in the Reparameterization pass and in Flipout's mean pass (run_pass3)
#endif
constexpr int SP_HROWS = 2;  // conv rows per half tile
constexpr int SP_LROWS = 3;  // LDS rows (64 channels each) of the store side: r0, r1 and the carry row
constexpr int SP_MAXST = 7;  // K-stages whose weight tiles stay resident

typedef __attribute__((ext_vector_type(8))) short i16x8;

// elementwise maximum of eight packed bf16.  RELU: the caller clamps at 0 afterwards (signed 16-bit compare, see above)
template <bool RELU>
__device__ __forceinline__ u32x4 sp_max8(const u32x4 a, const u32x4 b) {
  if constexpr (RELU) {
    return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(i16x8, a), __builtin_bit_cast(i16x8, b)));
  } else {
    u32x4 r;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float lo = fmaxf(u2f(a[d] << 16), u2f(b[d] << 16));
      const float hi = fmaxf(u2f(a[d] & 0xffff0000u), u2f(b[d] & 0xffff0000u));
      r[d] = (f2u(lo) >> 16) | (f2u(hi) & 0xffff0000u);
    }
    return r;
  }
}

// fragments of one K-stage of one pass: activations a[kk][mi] and weights w[kk][ni]
struct SpFrag {
  u32x4 a[NG / 2][2], w[NG / 2][2];
};
template <int MIA, bool ZERO>
__device__ __forceinline__ void sp_mma(const SpFrag& f, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int kk = 0; kk < NG / 2; ++kk)
#pragma unroll
    for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if constexpr (BTX_SP_ABL & 1) { asm volatile("" ::"v"(f.w[kk][ni]), "v"(f.a[kk][mi])); acc[mi][ni][0] += 1.f; continue; }
        if constexpr (ZERO) {
          const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.w[kk][ni]),
                                                               __builtin_bit_cast(bf16x8, f.a[kk][mi]),
                                                               kk == 0 ? zc : acc[mi][ni], 0, 0, 0);
        } else {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.w[kk][ni]),
                                                               __builtin_bit_cast(bf16x8, f.a[kk][mi]), acc[mi][ni], 0, 0, 0);
        }
      }
}

// ContractParams fields used: pt_R (pooled rows per band), pt_rtiles (bands per image), pt_PP (patch bytes), pt_astage
// (patch slot bytes, 1-KiB multiple), st_sbytes (bytes of one sign-word slot, 128-byte multiple), sp_Hq / sp_Wq (pooled
// extent).  Geometry as for contract_stem_kernel (BTX_FLAG_ROWFUSE).
template <int KIND>
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const ContractParams pk) {
  int logical = xcd_logical();
  const ContractParams p = lane_view(pk, logical);
  constexpr int G = 8, BK = NG * G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef BTX_PT_TRACE
  const uint32_t tr_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
  uint32_t tr_pro = 0, tr_k = 0, tr_st = 0, tr_pool = 0, tr_bar = 0, tr_x = 0, tr_bld = 0;
#define SP_T(var) { __builtin_amdgcn_sched_barrier(0); const uint32_t n_ = (uint32_t)__builtin_amdgcn_s_memtime(); var += n_ - tr_x; tr_x = n_; __builtin_amdgcn_sched_barrier(0); }
#else
#define SP_T(var)
#endif
#if defined(BTX_PT_TRACE) && defined(BTX_SP_TR2)  // the passes of a Flipout K role without their barrier waits (tr[0] mean, tr[1] delta, tr[2] waits)
  uint32_t tr_m = 0, tr_d = 0, tr_w = 0, tr_y = 0;
#define SP_T2(var) { __builtin_amdgcn_sched_barrier(0); const uint32_t n_ = (uint32_t)__builtin_amdgcn_s_memtime(); var += n_ - tr_y; tr_y = n_; __builtin_amdgcn_sched_barrier(0); }
#else
#define SP_T2(var)
#endif
#define SP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  const RngLive rl = rng_live<KIND>(p);

  const int tid = threadIdx.x;
  int lane = tid & 63;  // (not const: made opaque once per phase, see the band loop)
  int l31 = lane & 31;
  int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  int gtid = tid & 255;  // thread index inside the group

  uint32_t u_rest, u_ntile, u_img, u_band;
  fdivmod((uint32_t)logical, p.fd_ntiles, (uint32_t)p.ntiles, u_rest, u_ntile);
  fdivmod(u_rest, p.fd_rtiles, (uint32_t)p.pt_rtiles, u_img, u_band);
  const int ntile = (int)u_ntile, img = (int)u_img, band = (int)u_band;

  const int PB = p.pt_R;
  const int P0 = band * PB;   // first pooled row of the band
  const int c0 = 2 * P0 - 1;  // first conv row of the band (-1 for the first band: the pool's padding row)
  const int Hq = p.sp_Hq, Wq = p.sp_Wq, Wo = p.Wo, Ho = p.Ho;
  const int NH = min(PB, Hq - P0) + 1;  // half tiles of the band (the last one only contributes its first row)
  const int RowE = p.W * p.C;    // elements per input row
  const int nstages = p.K / BK;  // K = KH * Cg, Cg % BK == 0
  const int spr = p.Cg / BK;     // stages per kernel row
  const int W_OFF = 0, A_OFF = nstages * DW_STAGE, X_OFF = A_OFF + 2 * p.pt_astage;  // weights | raw patches | signed patch
  const int S_OFF = X_OFF + (KIND == 1 ? p.pt_astage : 0), R_OFF = S_OFF + 2 * p.st_sbytes;
  const int row_b = Wo * 128;    // bytes of one LDS row of the store side: 64 channels of Wo pixels
  const int C_OFF = R_OFF + SP_HROWS * row_b;  // carry row: max(r0, r1) of the previous half tile
  float* const ba_lds = (float*)(smem + R_OFF + SP_LROWS * row_b);
  const int NINF_OFF = R_OFF + SP_LROWS * row_b + 4 * BN * 4;  // two 16-byte chunks behind the constants: "below everything" for the pool
  if (threadIdx.x < 8) *(uint32_t*)(smem + NINF_OFF + 4 * threadIdx.x) = threadIdx.x < 4 ? 0x80008000u : 0xff80ff80u;

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);

  // ---- the weight tiles of every stage: 4 rows of mu (+ 4 of delta) of 1 KiB per stage, resident for the whole band
  {
    constexpr int PER = (KIND == 1) ? 8 : 4;
    const uint32_t w_tile = (uint32_t)ntile * (uint32_t)(p.K / G) * 1024u + (uint32_t)lane * 16u;
    const int npc = nstages * PER;
    for (int i = wave; i < npc; i += 8) {
      const int st = i / PER, r = i % PER;
      const uint32_t src = w_tile + (uint32_t)st * (uint32_t)NG * 1024u + (uint32_t)(r & 3) * 1024u +
                           (r >= 4 ? p.wt_delta_off : 0u);
      dma16(wt_rsrc, src, smem + W_OFF + st * DW_STAGE + (r >= 4 ? 4096 : 0) + (r & 3) * 1024);
    }
  }
  // ---- input rows of half tile u: bytes [base, +pt_PP) of x, 1 KiB per DMA instruction, issued by `nw` waves of which
  //      this one is number `wi`.  32-bit wrap-around arithmetic: the rows above the image of the first band's first half
  //      tile (conv row -1) come out as offsets beyond the descriptor, i.e. zeros, and everything at or below row 0 lands
  //      where it belongs.
  auto tile_base_e = [&](int u) __attribute__((always_inline)) {
    return (img * p.H + (c0 + SP_HROWS * u) * p.sh) * RowE;  // element offset of the patch in x (may be negative)
  };
  auto issue_patch = [&](int u, int wi, int nw) __attribute__((always_inline)) {
    const uint32_t base_b = (uint32_t)tile_base_e(u) * 2u;
    unsigned char* dst = smem + A_OFF + (u & 1) * p.pt_astage;
    const int npieces = p.pt_astage >> 10;
    for (int i = wi; i < npieces; i += nw) {
      const uint32_t o = (uint32_t)i * 1024u + (uint32_t)lane * 16u;
      dma16(x_rsrc, o < (uint32_t)p.pt_PP ? base_b + o : DMA_OOB, dst + i * 1024);
    }
  };
  // ---- s_in words of the patch's element range (index space = the row-fused x; floor semantics for negative offsets),
  //      by `nt` threads of which this one is number `ti`
  auto write_signs = [&](int u, int ti, int nt) __attribute__((always_inline)) {
    if constexpr (KIND == 1) {
      const int word0 = tile_base_e(u) >> 5;
      const int nwords = p.st_sbytes >> 2;
      unsigned char* ss = smem + S_OFF + (u & 1) * p.st_sbytes;
      for (int w = ti; w < nwords; w += nt)
        *(uint32_t*)(ss + w * 4) = btx_sign_word((uint32_t)(word0 + w), rl.kin_a, rl.kin_b);
    }
  };
  // ---- x * s_in of half tile u, once per element: every 16-byte granule of the patch (8 elements, inside one 32-sign
  //      word: rows are whole granules) gets its signs.  The delta pass of the K loop then reads this copy as it is — the
  //      per-fragment masks of the other kernels would be applied ~12 times per input element here (7x7 windows, stride 2).
  auto build_signed = [&](int u) __attribute__((always_inline)) {
    if constexpr (KIND == 1) {
      const int base_e = tile_base_e(u);
      const int word0 = base_e >> 5;
      const unsigned char* raw = smem + A_OFF + (u & 1) * p.pt_astage;
      const unsigned char* ss = smem + S_OFF + (u & 1) * p.st_sbytes;
      unsigned char* dst = smem + X_OFF;
      const int nchunks = p.pt_PP >> 4;
      // Up to 5 chunks per thread (a 7x7 / stride-2 stem at 224^2: 1 035 chunks): branch-free — a thread whose index runs past the
      // end repeats the last chunk (same bytes to the same address) — so that all reads go out before the first store.  As a loop
      // with one chunk per iteration each chunk waited for its own LDS round trip (2.0k cycles per half tile for ~50 VALU).
      constexpr int SP_CPY = 5;
      if (nchunks <= SP_CPY * 256) {
        u32x4 v[SP_CPY];
        uint32_t w[SP_CPY];
        int ci[SP_CPY];
#pragma unroll
        for (int k = 0; k < SP_CPY; ++k) {
          ci[k] = min(gtid + 256 * k, nchunks - 1);
          v[k] = *(const u32x4*)(raw + ci[k] * 16);
          w[k] = *(const uint32_t*)(ss + (((base_e + 8 * ci[k]) >> 5) - word0) * 4);
        }
#pragma unroll
        for (int k = 0; k < SP_CPY; ++k) {
          const int e = base_e + 8 * ci[k];
          const uint32_t ws = w[k] << (((uint32_t)e >> 3 & 3u) * 4u);
#pragma unroll
          for (int d = 0; d < 4; ++d) v[k][d] ^= (ws << d) & 0x80008000u;
        }
#pragma unroll
        for (int k = 0; k < SP_CPY; ++k) *(u32x4*)(dst + ci[k] * 16) = v[k];
        return;
      }
      for (int i = gtid; i < nchunks; i += 256) {
        u32x4 v = *(const u32x4*)(raw + i * 16);
        const int e = base_e + 8 * i;
        const uint32_t w = *(const uint32_t*)(ss + ((e >> 5) - word0) * 4);
        // element pair j of the word (elements 2j, 2j+1) has its signs at bits 15-j and 31-j: dword d of granule c is pair 4c+d
        const uint32_t ws = w << (((uint32_t)e >> 3 & 3u) * 4u);
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] ^= (ws << d) & 0x80008000u;
        *(u32x4*)(dst + i * 16) = v;
      }
    }
  };
  issue_patch(0, wave, 8);
  write_signs(0, tid, 512);
  {
    const bool has_bias = p.mu_b != nullptr;
    const bool has_aff = (p.ep_scale != nullptr) || (p.ep_shift != nullptr);
    ep_fill_constants<KIND>(p, rl, ba_lds, tid, ntile, 0, has_bias, has_aff);
  }
  const bool has_bias = p.mu_b != nullptr;
  const bool has_aff = (p.ep_scale != nullptr) || (p.ep_shift != nullptr);
  const bool relu = p.ep_relu != 0;

  // ---- MFMA role: the wave owns pixels [64*w4, +64) of its group's half tile (2 conv rows, flattened (row, col))
  int eo[2];      // byte offset of the pixel's window inside the patch
  int st_off[2];  // byte offset of the pixel's 8-byte piece (lane half h) in the store-side rows, chunk swizzle in bits
                  // 4-6 (the address of chunk c is st_off ^ (c << 4)); -1: the pixel does not exist
  uint32_t st_orow[2];  // s_out index of the pixel's first channel in half tile 0
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int pl = w4 * 64 + mi * 32 + l31;
    const bool ok = pl < 2 * Wo;
    const int plc = ok ? pl : 0;
    const int r = plc >= Wo ? 1 : 0;
    const int col = plc - r * Wo;
    eo[mi] = (r * p.sh * RowE + col * p.sw * p.C) * 2;
    st_off[mi] = ok ? R_OFF + r * row_b + col * 128 + (((col >> 1) & 7) << 4) + h * 8 : -1;
    st_orow[mi] = (uint32_t)(((img * Ho + c0 + r) * Wo + col) * p.N + ntile * BN);
  }
  // ---- pool role: thread (of the group) handles pooled pixels (gtid + 256 j) >> 3, j = 0, 1, 8-channel chunk gtid & 7;
  //      byte offsets of that chunk in a store-side row for conv columns 2pc-1, 2pc, 2pc+1 (-1: none)
  int pool_coff[2][3];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int pc = (gtid + 256 * j) >> 3, col = 2 * pc - 1 + k;
      pool_coff[j][k] = (pc < Wq && col >= 0 && col < Wo) ? col * 128 + (((gtid & 7) ^ ((col >> 1) & 7)) << 4) : -1;
    }
  // carry role: the thread owns chunk gtid & 7 of columns (gtid + 256 j) >> 3, j = 0..3 (-1: none)
  int carry_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = (gtid + 256 * j) >> 3;
    carry_off[j] = col < Wo ? col * 128 + (((gtid & 7) ^ ((col >> 1) & 7)) << 4) : -1;
  }
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (uint32_t)((size_t)p.NB * Hq * Wq * p.N * 2), 0x00020000);

  f32x16 accm[2][2], accd[2][2];

  // =================== K role ===================================================================================
  // One pass over the K-stages of half tile u: activations from `abase` (the raw patch: mean pass; the signed copy: delta
  // pass), weights from the resident tiles at +woff (0: mu, 4096: delta).  A workgroup barrier is passed before stages
  // bs0 and bs1 (if inside the pass); returns how many.
  auto run_pass = [&](const unsigned char* abase, int woff, f32x16 (&acc)[2][2], int bs0, int bs1, auto mia_tag)
                      __attribute__((always_inline)) {
    constexpr int MIA = decltype(mia_tag)::value;
    int l_j = 0, l_row = 0, l_b = 0;  // stage being loaded: byte offset of (kernel row, stage within the row)
    auto advance_load = [&]() __attribute__((always_inline)) {
      l_b += BK * 2;
      if (++l_j == spr) { l_j = 0; l_row += RowE * 2; l_b = l_row; }
    };
    auto load = [&](SpFrag& f, int s) __attribute__((always_inline)) {
      const unsigned char* as = abase + l_b;
      const unsigned char* ws = smem + W_OFF + s * DW_STAGE + woff;
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk) {
        const int row = 2 * kk + h;
#pragma unroll
        for (int mi = 0; mi < MIA; ++mi) f.a[kk][mi] = *(const u32x4*)(as + eo[mi] + row * 16);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) f.w[kk][ni] = *(const u32x4*)(ws + (row * BN + ni * 32 + l31) * 16);
      }
      advance_load();
    };
    int nb = 0;
    SpFrag fa, fb;
    load(fa, 0);
    auto iter = [&](int s, SpFrag& cur, SpFrag& nxt, auto zero_tag) __attribute__((always_inline)) {
      if (s == bs0 || s == bs1) { SP_BARRIER(); ++nb; }
      if constexpr (BTX_SP_ABL & 2) { nxt = cur; asm volatile("" : "+v"(nxt.a[0][0]), "+v"(nxt.w[0][0])); }
      else if (s + 1 < nstages) load(nxt, s + 1);
      sp_mma<MIA, decltype(zero_tag)::value>(cur, acc);
    };
    iter(0, fa, fb, std::true_type{});
    int s = 1;
    for (; s + 1 < nstages; s += 2) {
      iter(s, fb, fa, std::false_type{});
      iter(s + 1, fa, fb, std::false_type{});
    }
    if (s < nstages) iter(s, fb, fa, std::false_type{});
    return nb;
  };
  // The same pass with the fragments requested TWO stages ahead (round 6).  Measured (profiles/r06_experiments.txt E6): a pass
  // takes as long without its MFMAs as with them — a stage's eight fragment reads come back after ~500 cycles (the store group's
  // LDS traffic and the patch DMA queue in front of them), its eight MFMAs cover 256: with one stage of lead every stage exposes
  // half a round trip.  Three register sets give two stages (512 cycles) of lead.  What makes it work as C++:
  //   * never more than 15 LDS requests in flight — the counter has 4 bits, and a compiler whose wait insertion cannot count
  //     what is outstanding waits for everything in front of every MFMA: the 8th request of a stage goes out behind an
  //     lgkmcnt(14), i.e. after the oldest request of the stage in front has returned;
  //   * every stage requests "the stage after next" — the last two as well (reads of LDS bytes nobody uses): with `if` around
  //     requests and waits the wait insertion also walks the paths that take one branch and skip the other;
  //   * the waits are s_waitcnt BUILTINS (instructions the wait insertion sees), pinned by sched_barrier.
  // Needs (nstages - 1) % 3 == 0 (7 for the ResNet stem: one body starts the accumulators, one body per register set rotates);
  // one accumulator set may be live beside the three fragment sets (Reparameterization; Flipout's mean pass, whose delta
  // accumulators are not yet live), not two.
  auto run_pass3 = [&](const unsigned char* abase, int woff, f32x16 (&acc)[2][2], int bs0, int bs1) __attribute__((always_inline)) {
    int l_j = 0, l_row = 0, l_b = 0, l_s = 0;  // the stage whose fragments are requested next
    // requests 0..7 of a stage: (k-half, a0, a1, w0, w1) — the first k-half's fragments first
    auto req = [&](SpFrag& f, auto i_tag) __attribute__((always_inline)) {
      constexpr int i = decltype(i_tag)::value, kk = i >> 2, j = i & 3;
      const int row = 2 * kk + h;
      if constexpr (j < 2) f.a[kk][j] = *(const u32x4*)(abase + l_b + eo[j] + row * 16);
      else f.w[kk][j - 2] = *(const u32x4*)(smem + W_OFF + l_s * DW_STAGE + woff + (row * BN + (j - 2) * 32 + l31) * 16);
    };
    auto advance = [&]() __attribute__((always_inline)) {
      ++l_s;
      l_b += BK * 2;
      if (++l_j == spr) { l_j = 0; l_row += RowE * 2; l_b = l_row; }
    };
    auto mma_half = [&](const SpFrag& f, auto kk_tag, auto zero_tag) __attribute__((always_inline)) {
      constexpr int kk = decltype(kk_tag)::value;
      constexpr bool ZERO = decltype(zero_tag)::value;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.w[kk][ni]),
                                                               __builtin_bit_cast(bf16x8, f.a[kk][mi]),
                                                               (ZERO && kk == 0) ? zc : acc[mi][ni], 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    int nb = 0;
    // stage s_ from `cur`; `pre` takes the fragments of stage s_ + 2.  In flight at the top: the 8 requests of stage s_ + 1.
    auto stage = [&](SpFrag& cur, SpFrag& pre, int s_, auto zero_tag) __attribute__((always_inline)) {
      if (s_ == bs0 || s_ == bs1) { SP_BARRIER(); ++nb; }
      __builtin_amdgcn_s_waitcnt(0xc07f | (8 << 8));   // lgkmcnt(8): stage s_ is here
      __builtin_amdgcn_sched_barrier(0);
      static_for_ep<0, 7>([&](auto i_tag) __attribute__((always_inline)) { req(pre, i_tag); });  // 8 + 7 = 15 in flight
      __builtin_amdgcn_sched_barrier(0);
      mma_half(cur, I0{}, zero_tag);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xc07f | (14 << 8));  // lgkmcnt(14): the oldest of them has returned
      __builtin_amdgcn_sched_barrier(0);
      req(pre, std::integral_constant<int, 7>{});
      advance();
      __builtin_amdgcn_sched_barrier(0);
      mma_half(cur, I1{}, zero_tag);
      __builtin_amdgcn_sched_barrier(0);
    };
    SpFrag f0, f1, f2;
    static_for_ep<0, 8>([&](auto i_tag) __attribute__((always_inline)) { req(f0, i_tag); });
    advance();
    __builtin_amdgcn_sched_barrier(0);
    static_for_ep<0, 7>([&](auto i_tag) __attribute__((always_inline)) { req(f1, i_tag); });
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f | (14 << 8));
    __builtin_amdgcn_sched_barrier(0);
    req(f1, std::integral_constant<int, 7>{});
    advance();
    __builtin_amdgcn_sched_barrier(0);
    stage(f0, f2, 0, std::true_type{});
    for (int s_ = 1; s_ < nstages; s_ += 3) {
      stage(f1, f0, s_, std::false_type{});
      stage(f2, f1, s_ + 1, std::false_type{});
      stage(f0, f2, s_ + 2, std::false_type{});
    }
    return nb;
  };
  // the K role of one phase: three workgroup barriers (the store group runs its three steps beside it)
  auto run_k = [&](int u, auto mia_tag) __attribute__((always_inline)) {
    const unsigned char* raw = smem + A_OFF + (u & 1) * p.pt_astage;
    int nb;
    // the K role's wave ahead of the store role's in the SIMD's issue arbitration: Flipout band 139.2k -> 136.8k cycles (E11)
    __builtin_amdgcn_s_setprio(BTX_STEM_PRIO);
    if constexpr (KIND == 1) {
#if BTX_STEM_STEPS == 0
      run_pass(raw, 0, accm, -1, -1, mia_tag);
      SP_BARRIER();  // the signed copy is complete (build_signed ran before this pass)
      nb = 1 + run_pass(smem + X_OFF, 4096, accd, (nstages + 1) >> 1, -1, mia_tag);
#else
      // steps of the phase: sign copy | mean pass | delta pass, against the store group's stage 0-31 | stage 32-63 | pool.
      // Phase timers (profiles/r02_stem_pool.txt, per wave and step): sign copy 2.35k cycles, each pass 2.6k; store steps 1.9k,
      // 1.9k, 2.5k.  With the mean pass in the first step (round 2) the phase cost max(4.95, 1.9) + max(1.5, 1.9) + max(1.1,
      // 2.5) = 9.35k; one K step per store step: 2.35 + 2.6 + 2.6 = 7.55k.  Measured at 20 lanes: 1300 -> 1200 us per launch.
      // (Also measured, not kept: the sign copy sliced into the shadow of the mean pass's MFMAs with the delta pass in two
      // steps: 1350 us; the copy's LDS reads four chunks deep instead of one at a time: no change.)
#if BTX_STEM_NB == 2
      SP_BARRIER();  // the signed copy is complete (build_signed ran before this call)
#endif
      SP_T2(tr_w)
      if (decltype(mia_tag)::value == 2 && BTX_STEM_PASS3 && (nstages - 1) % 3 == 0) run_pass3(raw, 0, accm, -1, -1);  // (accd is not live yet)
      else run_pass(raw, 0, accm, -1, -1, mia_tag);
      SP_T2(tr_m)
      SP_BARRIER();  // (BTX_STEM_NB == 1: this is the barrier behind which every wave's part of the signed copy is complete)
      SP_T2(tr_w)
      run_pass(smem + X_OFF, 4096, accd, -1, -1, mia_tag);
      SP_T2(tr_d)
      nb = BTX_STEM_NB;
#endif
    } else {
#if BTX_STEM_NB == 2
      const int b0_ = (nstages + 2) / 3, b1_ = 2 * b0_;
#else
      const int b0_ = (nstages * 4 + 3) / 7 > 0 ? (nstages * 4 + 3) / 7 : -1, b1_ = -1;  // one inner barrier, before stage 4 of 7
#endif
      if (decltype(mia_tag)::value == 2 && BTX_STEM_PASS3 && (nstages - 1) % 3 == 0) nb = run_pass3(raw, 0, accm, b0_, b1_);
      else nb = run_pass(raw, 0, accm, b0_, b1_, mia_tag);
    }
    for (; nb < BTX_STEM_NB; ++nb) SP_BARRIER();
    __builtin_amdgcn_s_setprio(0);
  };

  // =================== store role ================================================================================
  // fragments -> LDS rows r0 / r1, channel half NI (btx_epilogue.h stage 1 + the bf16 rounding of its stage 2; the ReLU
  // follows the pool)
  // m0_tag / m1_tag: what the wave knows about its two 32-pixel tiles — 1: every lane's pixel exists (the store is unconditional),
  // 0: none does (the tile is skipped), 2: some do (the store sits under the lane's predicate).  With a predicate per store the
  // arithmetic of each (channel run, tile) is a basic block of its own between exec-mask juggling: 16 blocks of ~20 instructions
  // per call that the scheduler cannot interleave (a one-wave-per-SIMD role: 7.7 cycles per VALU instruction measured).  On the
  // ResNet stem every tile is all-or-nothing (224 pixels of a half tile = 7 tiles of 32), so the whole call is straight-line code.
  auto stage_half = [&](int u, int mia, auto ni_tag, auto bias_tag, auto aff_tag, auto m0_tag, auto m1_tag) __attribute__((always_inline)) {
    constexpr int ni = decltype(ni_tag)::value;
    constexpr bool BIAS = decltype(bias_tag)::value, AFF = decltype(aff_tag)::value;
    constexpr int TM0 = decltype(m0_tag)::value, TM1 = decltype(m1_tag)::value;
    uint32_t SB = 0x80000000u;
    asm volatile("" : "+s"(SB));  // (the mask as an SGPR operand of v_bitop3: a 32-bit literal does not fit the VOP3 encoding)
    uint32_t wsh[2] = {0u, 0u};
    if constexpr (KIND == 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const uint32_t orow = st_orow[mi] + (uint32_t)(SP_HROWS * u * Wo * p.N);
        wsh[mi] = btx_sign_word((orow + 32u * ni) >> 5, rl.kout_a, rl.kout_b) << (2 * h);
      }
    }
    bool wr[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) wr[mi] = mi < mia && st_off[mi] >= 0;
    // the per-channel constants of all four 8-channel runs first: read where they are used, every run starts with an LDS round
    // trip the wave has nothing to put beside (one wave per SIMD in this role): Reparameterization staging 23.0k -> 19.4k cycles
    // per band (profiles/r06_experiments.txt E5)
    f32x4 bm_[4], bd_[4], sc_[4], sh_[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cl = ni * 32 + 8 * q + 4 * h;
      if constexpr (BIAS) {
        bm_[q] = *(const f32x4*)(ba_lds + cl);
        bd_[q] = *(const f32x4*)(ba_lds + BN + cl);
      }
      if constexpr (AFF) {
        sc_[q] = *(const f32x4*)(ba_lds + 2 * BN + cl);
        sh_[q] = *(const f32x4*)(ba_lds + 3 * BN + cl);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bm = bm_[q], bd = bd_[q], sc = sc_[q], sh = sh_[q];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int tm = (mi == 0) ? TM0 : TM1;  // (a constant once the loop is unrolled)
        if (tm == 0) continue;
        f32x4 v;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float val = accm[mi][ni][4 * q + rr];
          if constexpr (BIAS) val += bm[rr];
          if constexpr (KIND == 1) {
            float dl = accd[mi][ni][4 * q + rr];
            if constexpr (BIAS) dl += bd[rr];
            // element e = 8q + 4h + rr of the word sits at bit ((e&1) ? 31 : 15) - (e>>1); wsh is pre-shifted by 2h
            const int sft = 31 - (((rr & 1) ? 31 : 15) - 4 * q - (rr >> 1));
            val += u2f(__builtin_amdgcn_bitop3_b32(f2u(dl), wsh[mi] << sft, SB, 0x78));  // dl ^ (w & SB): one v_bitop3
          }
          if constexpr (AFF) val = __builtin_fmaf(val, sc[rr], sh[rr]);
          v[rr] = val;
        }
        if (tm == 1 || wr[mi])
          *(u32x2*)(smem + (st_off[mi] ^ ((ni * 4 + q) << 4))) = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
      }
    }
  };
  auto stage_dispatch = [&](int u, int mia, auto ni_tag) __attribute__((always_inline)) {
    using T = std::true_type;
    using F = std::false_type;
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    if (has_bias) stage_half(u, mia, ni_tag, T{}, T{}, M2{}, M2{});  // (a bias without an affine: scale 1, shift 0 from the constants)
    else if (has_aff) {
      // wave-uniform: which of the wave's two tiles hold pixels for every lane / for none
      const unsigned long long e0 = __builtin_amdgcn_ballot_w64(0 < mia && st_off[0] >= 0);
      const unsigned long long e1 = __builtin_amdgcn_ballot_w64(1 < mia && st_off[1] >= 0);
      if (e0 == ~0ull && e1 == ~0ull) stage_half(u, mia, ni_tag, F{}, T{}, M1{}, M1{});
      else if (e0 == ~0ull && e1 == 0ull) stage_half(u, mia, ni_tag, F{}, T{}, M1{}, M0{});
      else stage_half(u, mia, ni_tag, F{}, T{}, M2{}, M2{});
    } else stage_half(u, mia, ni_tag, F{}, F{}, M2{}, M2{});
  };
  // LDS rows (+ the carry row of the previous half tile) -> pooled row P0+u-1; returns this thread's pieces of the next
  // carry row, max(r0, r1) (written behind the phase's last barrier, write_carry)
  auto pool_all = [&](int u, u32x4 (&cnew)[4], auto relu_tag) __attribute__((always_inline)) {
    constexpr bool RL = decltype(relu_tag)::value;
    const uint32_t ninf = RL ? 0x80008000u : 0xff80ff80u;  // below everything: most negative int16 pair | -inf pair
    const int prow = P0 + u - 1;
    const int cr0 = c0 + SP_HROWS * u;
    const bool r0_ok = cr0 >= 0 && cr0 < Ho, r1_ok = (cr0 + 1) < Ho && u + 1 < NH;  // the closing half tile has no r1
    const bool row_ok = u >= 1 && prow < Hq;
    // Branch-free: a chunk that does not exist (a column outside the row, a conv row outside the image) is read from a 16-byte
    // chunk of LDS that holds `ninf`, so every load of the thread goes out at once instead of one per exec-mask block (the
    // predicated form measured 2.3k cycles per role for ~160 VALU and 20 LDS reads)
    const int NI_OFF = NINF_OFF + (RL ? 0 : 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = row_ok && pool_coff[j][1] >= 0;
      u32x4 la[3], lb[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const bool in = pool_coff[j][k] >= 0;
        la[k] = *(const u32x4*)(smem + (in ? C_OFF + pool_coff[j][k] : NI_OFF));  // rows that do not exist were folded in as `ninf`
        lb[k] = *(const u32x4*)(smem + ((in && r0_ok) ? R_OFF + pool_coff[j][k] : NI_OFF));
      }
      u32x4 m = sp_max8<RL>(la[0], lb[0]);
#pragma unroll
      for (int k = 1; k < 3; ++k) m = sp_max8<RL>(m, sp_max8<RL>(la[k], lb[k]));
      if constexpr (RL) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        m = sp_max8<true>(m, z);
      }
      const int pc = (gtid + 256 * j) >> 3;
      const uint32_t off = ok ? (uint32_t)((((img * Hq + prow) * Wq + pc) * p.N + ntile * BN + (gtid & 7) * 8) * 2) : DMA_OOB;
      __builtin_amdgcn_raw_buffer_store_b128(m, out_rsrc, off, 0, 0);
    }
    u32x4 l0[4], l1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = carry_off[j] >= 0;
      l0[j] = *(const u32x4*)(smem + ((in && r0_ok) ? R_OFF + carry_off[j] : NI_OFF));
      l1[j] = *(const u32x4*)(smem + ((in && r1_ok) ? R_OFF + row_b + carry_off[j] : NI_OFF));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) cnew[j] = sp_max8<RL>(l0[j], l1[j]);
  };
  auto write_carry = [&](const u32x4 (&cnew)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (carry_off[j] >= 0) *(u32x4*)(smem + C_OFF + carry_off[j]) = cnew[j];
  };
  // Every path through the K role redefines ALL accumulators (those a wave does not compute are cleared): were some left
  // as they are, their previous values would be live through the K loops on every path and spill (116 VGPRs).
  auto clear_acc = [&](int mi0) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
      if (a >= mi0) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }
      }
  };
  // wave's 32-pixel tiles that feed a pooled row: the closing half tile (u == NH-1) only needs its first conv row
  auto mia_of = [&](int u) __attribute__((always_inline)) {
    const int limit = (u == NH - 1) ? Wo : 2 * Wo;
    const int m = (limit - w4 * 64 + 31) >> 5;
    return m < 0 ? 0 : (m > 2 ? 2 : m);
  };

  // =================== the band ==================================================================================
  u32x4 cnew[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  bool carry_pending = false;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // weights, first patch, constants
  SP_BARRIER();
#ifdef BTX_PT_TRACE
  tr_x = tr_t0;
  SP_T(tr_pro)
#endif
  using T = std::true_type;
  using F = std::false_type;
  for (int ph = 0; ph <= NH; ++ph) {
    // keep everything derived from the per-lane indices inside the phase: hoisted out of this loop, the address vectors of
    // both roles stay live across the K loops and the accumulators spill
    asm volatile("" : "+v"(lane), "+v"(l31), "+v"(h), "+v"(gtid), "+v"(eo[0]), "+v"(eo[1]), "+v"(st_off[0]), "+v"(st_off[1]));
    asm volatile("" : "+v"(st_orow[0]), "+v"(st_orow[1]), "+v"(carry_off[0]), "+v"(carry_off[1]), "+v"(carry_off[2]),
                      "+v"(carry_off[3]));
    asm volatile("" : "+v"(pool_coff[0][0]), "+v"(pool_coff[0][1]), "+v"(pool_coff[0][2]), "+v"(pool_coff[1][0]),
                      "+v"(pool_coff[1][1]), "+v"(pool_coff[1][2]));
    if (grp == (ph & 1)) {
      // ---------------- K role: half tile ph.  First the carry row this group produced as last phase's store group:
      // every read of the old one ended before the barrier that closed that phase, the next read is two barriers away.
      if (carry_pending) { write_carry(cnew); carry_pending = false; }
      if (ph < NH) {
        if (ph + 1 < NH) issue_patch(ph + 1, w4, 4);
        build_signed(ph);
        SP_T(tr_bld)
        const int mia = mia_of(ph);
        if (mia == 2) run_k(ph, std::integral_constant<int, 2>{});
        else if (mia == 1) { clear_acc(1); run_k(ph, std::integral_constant<int, 1>{}); }
        else { clear_acc(0); for (int b_ = 0; b_ < BTX_STEM_NB; ++b_) SP_BARRIER(); }
        SP_T(tr_k)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next patch: issued a whole K loop ago
      } else {
        clear_acc(0);
        for (int b_ = 0; b_ < BTX_STEM_NB; ++b_) SP_BARRIER();
      }
    } else {
      // ---------------- store role: half tile ph-1 (this group's accumulators of the previous phase)
      const int u = ph - 1;
      // (measured, not kept: the next patch requested HERE, by the group that multiplies it next phase — the K role's step gets 1k
      // cycles shorter, the band not one cycle: E10)
      if (ph + 1 < NH) write_signs(ph + 1, gtid, 256);
      if ((BTX_SP_ABL & 4) && u >= 0) {
        for (int b_ = 0; b_ < BTX_STEM_NB; ++b_) SP_BARRIER();
      } else if (u >= 0) {
        const int mia = mia_of(u);
        stage_dispatch(u, mia, std::integral_constant<int, 0>{});
        SP_T(tr_st)
#if BTX_STEM_NB == 2
        SP_BARRIER();
        SP_T(tr_bar)
#endif
        stage_dispatch(u, mia, std::integral_constant<int, 1>{});
        SP_T(tr_st)
        SP_BARRIER();
        SP_T(tr_bar)
        if (relu) pool_all(u, cnew, T{}); else pool_all(u, cnew, F{});
        carry_pending = true;
        SP_T(tr_pool)
      } else {
        for (int b_ = 0; b_ < BTX_STEM_NB; ++b_) SP_BARRIER();
      }
    }
    SP_BARRIER();
    SP_T(tr_bar)
  }
#ifdef BTX_PT_TRACE
  if (p.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t tr_t3 = (uint32_t)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
      uint32_t* tr = (uint32_t*)p.trace + (size_t)(blockIdx.x * 8 + wave) * 8;
      tr[0] = tr_pro; tr[1] = tr_bld; tr[2] = tr_k; tr[3] = tr_st; tr[4] = tr_pool; tr[5] = tr_t3 - tr_t0; tr[6] = tr_bar;
#ifdef BTX_SP_TR2
      tr[0] = tr_m; tr[1] = tr_d; tr[6] = tr_w;
#endif
      tr[7] = tr_t0;
    }
  }
#endif
}
#undef SP_T
#undef SP_T2
#undef SP_BARRIER

static int launch_stem_pool_impl(int kind, const ContractParams& p, int nwg, hipStream_t st) {
#define BTX_LAUNCH_SP(KIND)                                                                                       \
  do {                                                                                                            \
    auto kfn = stem_pool_kernel<KIND>;                                                                            \
    static bool attr_done = false;                                                                                \
    if (!attr_done) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);   \
      if (e != hipSuccess) return (int)e;                                                                         \
      attr_done = true;                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(512), p.pt_lds, st, p);                                               \
  } while (0)
  int rc = launch_presample_impl<1>(kind, p, st);
  if (rc) return rc;
  if (kind == 0) BTX_LAUNCH_SP(0); else BTX_LAUNCH_SP(1);
#undef BTX_LAUNCH_SP
  return (int)hipGetLastError();
}

}  // namespace btx
