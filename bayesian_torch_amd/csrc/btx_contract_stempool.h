// btx_contract_stempool.h — the row-fused stem contraction with the 3x3 / stride-2 / pad-1 max-pool that follows it in a
// ResNet folded into its store (bf16, gfx950).
//
// What the un-fused chain costs on ResNet18, batch 64: the stem kernel (btx_contract_stem.h) writes 112x112x64 per
// image (103 MB), the pool kernel reads them back and writes 56x56x64 (26 MB); a stem tile is 7 K-stages between a
// patch fetch and a 28-KB store, so its workgroups spend more time waiting than multiplying.  Here one 8-wave
// workgroup per CU walks down a BAND of an image:
//
//   * the layer's pre-sampled weight tiles (all K-stages: 56 KiB for a 7x7 Flipout stem) are fetched ONCE and stay in
//     LDS; the K loop of a tile therefore has no ring, no DMA and NO barrier: the eight waves free-run through their
//     stages and interleave on the matrix pipe by themselves;
//   * a tile = 4 conv rows x the full width (waves 0-3: rows 0-1, waves 4-7: rows 2-3); the input rows of tile t+1 are
//     fetched (one contiguous byte range, LDS-DMA) and their s_in words hashed while tile t multiplies;
//   * the epilogue (bias, Flipout combine, BN affine, ReLU, bf16 rounding — the same arithmetic, in the same order, as
//     btx_epilogue.h: results are bit-identical to the un-fused chain) writes the tile's four conv rows r0..r3 into LDS,
//     half the channels at a time (32 channels x Wo pixels x 4 rows, XOR-swizzled 16-byte chunks);
//   * pooled row 2t-1 = max3x3 over (carry, r0) and pooled row 2t = max3x3 over (r0, r1, r2), where carry = max(r2, r3) of
//     tile t-1, kept in one LDS row per channel half (rows counted from the band's first conv row 2*P0 - 1); 16 B per
//     lane, 64 contiguous bytes per pooled pixel and channel half.  A band of PB pooled rows needs 2*PB + 1 conv rows:
//     PB/2 tiles and a closing one-row tile.
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

constexpr int SP_RING = 6;   // LDS rows of the store side: the tile's 4 conv rows + one carry row per channel half
constexpr int SP_TROWS = 4;  // conv rows per tile
constexpr int SP_MAXST = 7;  // K-stages whose weight tiles stay resident

// ContractParams fields used: pt_R (tiles per band), pt_rtiles (bands per image), pt_PP (patch bytes), pt_astage (patch
// slot bytes: a multiple of 8 KiB so every wave issues the same number of DMA instructions), st_sbytes (bytes of one
// sign-word slot), sp_Hq / sp_Wq (pooled extent).  Geometry as for contract_stem_kernel (BTX_FLAG_ROWFUSE).
template <int KIND>
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const ContractParams p) {
  constexpr int G = 8, BK = NG * G, NT = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef BTX_PT_TRACE
  const uint32_t tr_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
  uint32_t tr_pro = 0, tr_pre = 0, tr_k = 0, tr_st = 0, tr_pool = 0, tr_bar = 0, tr_x = 0;
#define SP_T(var) { __builtin_amdgcn_sched_barrier(0); const uint32_t n_ = (uint32_t)__builtin_amdgcn_s_memtime(); var += n_ - tr_x; tr_x = n_; __builtin_amdgcn_sched_barrier(0); }
#else
#define SP_T(var)
#endif
  const RngLive rl = rng_live<KIND>(p);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, w4 = wave & 3;

  int logical;
  {
    const int nwg = gridDim.x, L = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = L & 7, slot = L >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  uint32_t u_rest, u_ntile, u_img, u_band;
  fdivmod((uint32_t)logical, p.fd_ntiles, (uint32_t)p.ntiles, u_rest, u_ntile);
  fdivmod(u_rest, p.fd_rtiles, (uint32_t)p.pt_rtiles, u_img, u_band);
  const int ntile = (int)u_ntile, img = (int)u_img, band = (int)u_band;

  const int TB = p.pt_R;
  const int P0 = band * 2 * TB;  // first pooled row of the band
  const int c0 = 2 * P0 - 1;     // first conv row of the band (-1 for the first band: the pool's padding row)
  const int Hq = p.sp_Hq, Wq = p.sp_Wq, Wo = p.Wo, Ho = p.Ho;
  const int RowE = p.W * p.C;    // elements per input row
  const int nstages = p.K / BK;  // K = KH * Cg, Cg % BK == 0
  const int spr = p.Cg / BK;     // stages per kernel row
  const int W_OFF = 0, A_OFF = nstages * DW_STAGE, S_OFF = A_OFF + 2 * p.pt_astage, R_OFF = S_OFF + 2 * p.st_sbytes;
  const int row_b = Wo * 64;     // bytes of one ring row: 32 channels of Wo pixels
  const int C_OFF = R_OFF + SP_TROWS * row_b;  // carry rows (one per channel half): max of the previous tile's last two rows
  float* const ba_lds = (float*)(smem + R_OFF + SP_RING * row_b);

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
  // ---- input rows of tile t: bytes [base, +pt_PP) of x, 1 KiB per DMA instruction.  32-bit wrap-around arithmetic: the
  //      rows above the image of the first band's first tile (conv row -1) come out as offsets beyond the descriptor,
  //      i.e. zeros, and everything at or below row 0 lands where it belongs.
  auto tile_base_e = [&](int t) __attribute__((always_inline)) {
    return (img * p.H + (c0 + SP_TROWS * t) * p.sh) * RowE;  // element offset of the patch in x (may be negative)
  };
  auto issue_patch = [&](int t) __attribute__((always_inline)) {
    const uint32_t base_b = (uint32_t)tile_base_e(t) * 2u;
    unsigned char* dst = smem + A_OFF + (t & 1) * p.pt_astage;
    const int npieces = p.pt_astage >> 10;
    for (int i = wave; i < npieces; i += 8) {
      const uint32_t o = (uint32_t)i * 1024u + (uint32_t)lane * 16u;
      dma16(x_rsrc, o < (uint32_t)p.pt_PP ? base_b + o : DMA_OOB, dst + i * 1024);
    }
  };
  // ---- s_in words of the patch's element range (index space = the row-fused x; floor semantics for negative offsets)
  auto write_signs = [&](int t) __attribute__((always_inline)) {
    if constexpr (KIND == 1) {
      const int word0 = tile_base_e(t) >> 5;
      const int nwords = p.st_sbytes >> 2;
      unsigned char* ss = smem + S_OFF + (t & 1) * p.st_sbytes;
      for (int w = tid; w < nwords; w += NT)
        *(uint32_t*)(ss + w * 4) = btx_sign_word((uint32_t)(word0 + w), rl.kin_a, rl.kin_b);
    }
  };
  issue_patch(0);
  write_signs(0);
  {
    const bool has_bias = p.mu_b != nullptr;
    const bool has_aff = (p.ep_scale != nullptr) || (p.ep_shift != nullptr);
    ep_fill_constants<KIND>(p, rl, ba_lds, tid, ntile, 0, has_bias, has_aff);
  }
  const bool has_ba = (p.mu_b != nullptr) || (p.ep_scale != nullptr) || (p.ep_shift != nullptr);

  // ---- MFMA role: the wave owns pixels [64*w4, +64) of its half tile (2 conv rows, flattened (row, col))
  int eo[2];        // element offset of the pixel's window inside the patch
  int st_col[2];    // column of the pixel
  int st_j[2];      // row of the pixel inside the tile (0..3)
  bool px_ok[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int pl = w4 * 64 + mi * 32 + l31;
    px_ok[mi] = pl < 2 * Wo;
    const int plc = px_ok[mi] ? pl : 0;
    const int r = plc >= Wo ? 1 : 0;
    st_col[mi] = plc - r * Wo;
    st_j[mi] = 2 * half + r;
    eo[mi] = st_j[mi] * p.sh * RowE + st_col[mi] * p.sw * p.C;
  }
  // ---- pool role: thread = (pooled row of the pair, pooled column, 8-channel chunk of the 32-channel half)
  const bool pool_thread = tid < 2 * Wq * 4;
  int pool_pr, pool_pc, pool_c16;
  {
    const int tt = pool_thread ? tid : 0;
    pool_c16 = tt & 3;
    const int u = tt >> 2;
    pool_pr = u >= Wq ? 1 : 0;
    pool_pc = u - pool_pr * Wq;
  }
  int pool_coff[3];  // byte offset of the chunk in ring-row coordinates for conv columns 2pc-1, 2pc, 2pc+1 (-1: none)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int col = 2 * pool_pc - 1 + k;
    pool_coff[k] = (col >= 0 && col < Wo) ? col * 64 + ((pool_c16 ^ ((col >> 2) & 3)) * 16) : -1;
  }
  const bool carry_thread = tid < Wo * 4;
  const int carry_off = carry_thread ? (tid >> 2) * 64 + (((tid & 3) ^ ((tid >> 4) & 3)) * 16) : 0;
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (uint32_t)((size_t)p.NB * Hq * Wq * p.N * 2), 0x00020000);

  f32x16 accm[2][2], accd[2][2];

  // fragments of stage (kh, j) of tile t: element offset st_e = kh*RowE + j*BK inside the lane's window
  auto load_frag = [&](StageFrag& f, int t, int st_e, int s, int base_e, auto mia_tag) __attribute__((always_inline)) {
    constexpr int MIA = decltype(mia_tag)::value;
    const unsigned char* as = smem + A_OFF + (t & 1) * p.pt_astage + st_e * 2;
    const unsigned char* ss = smem + S_OFF + (t & 1) * p.st_sbytes;
    const unsigned char* ws = smem + W_OFF + s * DW_STAGE;
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      const int row = 2 * kk + h;
#pragma unroll
      for (int mi = 0; mi < MIA; ++mi) f.a[kk][mi] = *(const u32x4*)(as + eo[mi] * 2 + row * 16);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) f.wm[kk][ni] = *(const u32x4*)(ws + (row * BN + ni * 32 + l31) * 16);
    }
    if constexpr (KIND == 1) {
      const int word0 = base_e >> 5;
#pragma unroll
      for (int mi = 0; mi < MIA; ++mi) {
        // the stage's 32 signs start at element e0: composed from the two hashed words they straddle (btx_contract_stem.h)
        const int e0 = base_e + eo[mi] + st_e;
        const int wi = (e0 >> 5) - word0;
        const uint32_t w = *(const uint32_t*)(ss + wi * 4);
        const uint32_t w1 = *(const uint32_t*)(ss + wi * 4 + 4);
        const uint32_t k = ((uint32_t)e0 & 31u) >> 1;
        const uint32_t lo = ((w & 0xffffu) << 16) | (w1 & 0xffffu);
        const uint32_t hi = (w & 0xffff0000u) | (w1 >> 16);
        f.sw[mi] = ((lo << k) >> 16) | ((hi << k) & 0xffff0000u);
      }
    }
  };

  // =================== K loop of one tile: weights and patch are resident, no barrier ===========================
  auto run_tile = [&](int t, auto mia_tag) __attribute__((always_inline)) {
    constexpr int MIA = decltype(mia_tag)::value;
    const int base_e = tile_base_e(t);
    int l_j = 0, l_rowE = 0, l_e = 0;  // stage being loaded: element offset of (kernel row, stage within the row)
    auto advance_load = [&]() __attribute__((always_inline)) {
      l_e += BK;
      if (++l_j == spr) { l_j = 0; l_rowE += RowE; l_e = l_rowE; }
    };
    StageFrag fa, fb;
    load_frag(fa, t, 0, 0, base_e, mia_tag);
    advance_load();
    auto iter = [&](int s, StageFrag& cur, StageFrag& nxt) __attribute__((always_inline)) {
      DeltaFrag dfrag;
      load_delta<KIND>(dfrag, smem + W_OFF + s * DW_STAGE, l31, h);
      if (s + 1 < nstages) { load_frag(nxt, t, l_e, s + 1, base_e, mia_tag); advance_load(); }
      stage_mma<1, KIND, 2, MIA>(cur, dfrag, accm, accd, l31, h);
    };
    int s = 0;
    for (; s + 1 < nstages; s += 2) {
      iter(s, fa, fb);
      iter(s + 1, fb, fa);
    }
    if (s < nstages) iter(s, fa, fb);
  };

  // =================== store side ================================================================================
  // fragments -> ring rows, channel half NI (btx_epilogue.h stage 1 + the ReLU / bf16 rounding of its stage 2)
  auto stage_half = [&](int t, int mia, auto ni_tag, auto ba_tag) __attribute__((always_inline)) {
    constexpr int ni = decltype(ni_tag)::value;
    constexpr bool BA = decltype(ba_tag)::value;
    uint32_t wsh[2] = {0u, 0u};
    if constexpr (KIND == 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int cr = c0 + SP_TROWS * t + st_j[mi];
        const uint32_t orow = (uint32_t)(((img * Ho + cr) * Wo + st_col[mi]) * p.N + ntile * BN);
        wsh[mi] = btx_sign_word((orow + 32u * ni) >> 5, rl.kout_a, rl.kout_b) << (2 * h);
      }
    }
    unsigned char* dst[2];
    int swz[2];
    bool wr[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      dst[mi] = smem + R_OFF + st_j[mi] * row_b + st_col[mi] * 64 + h * 8;
      swz[mi] = (st_col[mi] >> 2) & 3;
      wr[mi] = mi < mia && px_ok[mi];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cl = ni * 32 + 8 * q + 4 * h;
      f32x4 bm, bd, sc, sh;
      if constexpr (BA) {
        bm = *(const f32x4*)(ba_lds + cl);
        bd = *(const f32x4*)(ba_lds + BN + cl);
        sc = *(const f32x4*)(ba_lds + 2 * BN + cl);
        sh = *(const f32x4*)(ba_lds + 3 * BN + cl);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        f32x4 v;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float val = accm[mi][ni][4 * q + rr];
          if constexpr (BA) val += bm[rr];
          if constexpr (KIND == 1) {
            float dl = accd[mi][ni][4 * q + rr];
            if constexpr (BA) dl += bd[rr];
            // element e = 8q + 4h + rr of the word sits at bit ((e&1) ? 31 : 15) - (e>>1); wsh is pre-shifted by 2h
            const int sft = 31 - (((rr & 1) ? 31 : 15) - 4 * q - (rr >> 1));
            val += u2f(f2u(dl) ^ ((wsh[mi] << sft) & 0x80000000u));
          }
          if constexpr (BA) val = __builtin_fmaf(val, sc[rr], sh[rr]);
          if (p.ep_relu) val = fmaxf(val, 0.f);
          v[rr] = val;
        }
        if (wr[mi]) *(u32x2*)(dst[mi] + ((q ^ swz[mi]) * 16)) = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
      }
    }
  };
  // ring (+ the carry row of the previous tile) -> pooled rows of tile t, channel half ni.  With r0..r3 the tile's conv rows:
  //   pool_pr == 0: pooled row P0+2t-1 = max3x3 over (carry = max(r2, r3) of tile t-1, r0)
  //   pool_pr == 1: pooled row P0+2t   = max3x3 over (r0, r1, r2)
  // Also returns this thread's piece of the next carry row, max(r2, r3) of column tid>>2, chunk tid&3.
  auto pool_half = [&](int t, int ni, u32x4& cnew) __attribute__((always_inline)) {
    const int prow = P0 + 2 * t - 1 + pool_pr;
    const bool ok = pool_thread && prow < Hq && (pool_pr == 0 ? (t >= 1) : (t < TB));
    const int cr0 = c0 + SP_TROWS * t;
    const unsigned char* ring = smem + R_OFF;
    const unsigned char* carry = smem + C_OFF + ni * row_b;
    const unsigned char* sp[3];
    bool sv[3];
    if (pool_pr == 0) {
      sp[0] = carry; sv[0] = true;  // rows that do not exist were folded in as -inf
      sp[1] = ring; sv[1] = cr0 >= 0 && cr0 < Ho;
      sp[2] = ring; sv[2] = false;
    } else {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) { sp[rr] = ring + rr * row_b; sv[rr] = (cr0 + rr) >= 0 && (cr0 + rr) < Ho; }
    }
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -__builtin_inff();
    if (ok) {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        if (sv[rr]) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if (pool_coff[k] >= 0) {
              const u32x4 v = *(const u32x4*)(sp[rr] + pool_coff[k]);
#pragma unroll
              for (int d = 0; d < 4; ++d) {
                m[2 * d] = fmaxf(m[2 * d], u2f(v[d] << 16));
                m[2 * d + 1] = fmaxf(m[2 * d + 1], u2f(v[d] & 0xffff0000u));
              }
            }
          }
        }
      }
    }
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f2u(m[2 * d]) >> 16) | (f2u(m[2 * d + 1]) & 0xffff0000u);
    const uint32_t off = ok ? (uint32_t)((((img * Hq + prow) * Wq + pool_pc) * p.N + ntile * BN + ni * 32 + pool_c16 * 8) * 2)
                            : DMA_OOB;
    __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, off, 0, 0);
    // next carry
    float c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = -__builtin_inff();
    if (carry_thread) {
#pragma unroll
      for (int rr = 2; rr < 4; ++rr) {
        if ((cr0 + rr) >= 0 && (cr0 + rr) < Ho) {
          const u32x4 v = *(const u32x4*)(ring + rr * row_b + carry_off);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            c[2 * d] = fmaxf(c[2 * d], u2f(v[d] << 16));
            c[2 * d + 1] = fmaxf(c[2 * d + 1], u2f(v[d] & 0xffff0000u));
          }
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) cnew[d] = (f2u(c[2 * d]) >> 16) | (f2u(c[2 * d + 1]) & 0xffff0000u);
  };
  auto write_carry = [&](int ni, const u32x4& cnew) __attribute__((always_inline)) {
    if (carry_thread) *(u32x4*)(smem + C_OFF + ni * row_b + carry_off) = cnew;
  };

  // =================== the band ==================================================================================
  u32x4 cnew1 = {0u, 0u, 0u, 0u};
#ifdef BTX_PT_TRACE
  tr_x = tr_t0;
  SP_T(tr_pro)
#endif
  for (int t = 0; t <= TB; ++t) {
    if (t == 0 ? (P0 >= Hq) : (P0 + 2 * t - 1 >= Hq)) break;  // nothing below contributes to a pooled row that exists
    // patch / sign words of tile t (and, t == 0, the weights and constants) are in LDS once every wave has waited for its
    // own DMA instructions — at the bottom of the previous tile, where they are long done — and met the others here
    if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SP_T(tr_bar)
    const bool more = (t + 1 <= TB) && (P0 + 2 * (t + 1) - 1 < Hq);
    if (more) {
      issue_patch(t + 1);
      write_signs(t + 1);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }
    SP_T(tr_pre)
    // pixels of this wave that feed a pooled row: the closing tile (t == TB) only needs its first conv row
    const int limit = (t == TB) ? (half == 0 ? Wo : 0) : 2 * Wo;
    int mia = (limit - w4 * 64 + 31) >> 5;
    mia = mia < 0 ? 0 : (mia > 2 ? 2 : mia);
    if (mia == 2) run_tile(t, std::integral_constant<int, 2>{});
    else if (mia == 1) run_tile(t, std::integral_constant<int, 1>{});
    SP_T(tr_k)

    using T = std::true_type;
    using F = std::false_type;
    if (t > 0) write_carry(1, cnew1);  // from the previous tile's second pool phase (behind this tile's first barrier)
    if (has_ba) stage_half(t, mia, std::integral_constant<int, 0>{}, T{});
    else stage_half(t, mia, std::integral_constant<int, 0>{}, F{});
    SP_T(tr_st)
    // the DMA instructions issued at the top of this tile have had the whole K loop to land: waiting for them here
    // costs nothing and keeps the pooled stores below out of the count
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SP_T(tr_bar)
    u32x4 cnew0;
    pool_half(t, 0, cnew0);
    SP_T(tr_pool)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SP_T(tr_bar)
    write_carry(0, cnew0);
    if (has_ba) stage_half(t, mia, std::integral_constant<int, 1>{}, T{});
    else stage_half(t, mia, std::integral_constant<int, 1>{}, F{});
    SP_T(tr_st)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SP_T(tr_bar)
    pool_half(t, 1, cnew1);
    SP_T(tr_pool)
  }
#ifdef BTX_PT_TRACE
  if (p.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t tr_t3 = (uint32_t)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
      uint32_t* tr = (uint32_t*)p.trace + (size_t)(blockIdx.x * 8 + wave) * 8;
      tr[0] = tr_pro; tr[1] = tr_pre; tr[2] = tr_k; tr[3] = tr_st; tr[4] = tr_pool; tr[5] = tr_t3 - tr_t0; tr[6] = tr_bar;
      tr[7] = tr_t0;
    }
  }
#endif
}
#undef SP_T

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
