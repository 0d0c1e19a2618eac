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
//     that had all eight waves multiply, then all eight store, spent 30 % of its time in the K loops: 105 us.)  A phase
//     is four steps with a workgroup barrier after each: K role: a quarter of the stages per step; store role: stage
//     channels 0-31 | pool them | stage channels 32-63 | pool them;
//   * the input rows of half tile f+1 are fetched (one contiguous byte range, LDS-DMA) by the K group and their s_in
//     words hashed by the store group during phase f;
//   * the epilogue (bias, Flipout combine, BN affine, bf16 rounding — the same arithmetic, in the same order, as
//     btx_epilogue.h) writes the half tile's conv rows r0, r1 into LDS, 32 channels at a time (XOR-swizzled 16-byte
//     chunks); pooled row P0+u-1 = max3x3 over (carry, r0) of half tile u, carry = max(r0, r1) of half tile u-1, one LDS
//     row per channel half (c0 = 2*P0 - 1 is the band's first conv row, r0 = c0 + 2u).  A band of PB pooled rows needs
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

constexpr int SP_HROWS = 2;  // conv rows per half tile
constexpr int SP_LROWS = 6;  // LDS rows of the store side: r0, r1 and, per channel half, the carry rows of even / odd half tiles
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

// ContractParams fields used: pt_R (pooled rows per band), pt_rtiles (bands per image), pt_PP (patch bytes), pt_astage
// (patch slot bytes, 1-KiB multiple), st_sbytes (bytes of one sign-word slot), sp_Hq / sp_Wq (pooled extent).  Geometry
// as for contract_stem_kernel (BTX_FLAG_ROWFUSE).
template <int KIND>
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const ContractParams p) {
  constexpr int G = 8, BK = NG * G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef BTX_PT_TRACE
  const uint32_t tr_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
  uint32_t tr_pro = 0, tr_k = 0, tr_st = 0, tr_pool = 0, tr_bar = 0, tr_x = 0;
#define SP_T(var) { __builtin_amdgcn_sched_barrier(0); const uint32_t n_ = (uint32_t)__builtin_amdgcn_s_memtime(); var += n_ - tr_x; tr_x = n_; __builtin_amdgcn_sched_barrier(0); }
#else
#define SP_T(var)
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

  const int PB = p.pt_R;
  const int P0 = band * PB;   // first pooled row of the band
  const int c0 = 2 * P0 - 1;  // first conv row of the band (-1 for the first band: the pool's padding row)
  const int Hq = p.sp_Hq, Wq = p.sp_Wq, Wo = p.Wo, Ho = p.Ho;
  const int NH = min(PB, Hq - P0) + 1;  // half tiles of the band (the last one only contributes its first row)
  const int RowE = p.W * p.C;    // elements per input row
  const int nstages = p.K / BK;  // K = KH * Cg, Cg % BK == 0
  const int spr = p.Cg / BK;     // stages per kernel row
  const int cs = (nstages + 3) >> 2;  // stages per step of a phase
  const int W_OFF = 0, A_OFF = nstages * DW_STAGE, S_OFF = A_OFF + 2 * p.pt_astage, R_OFF = S_OFF + 2 * p.st_sbytes;
  const int row_b = Wo * 64;     // bytes of one LDS row of the store side: 32 channels of Wo pixels
  const int C_OFF = R_OFF + SP_HROWS * row_b;  // carry rows [channel half][half-tile parity]: written by half tile u, read by u+1
  float* const ba_lds = (float*)(smem + R_OFF + SP_LROWS * row_b);

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
  issue_patch(0, wave, 8);
  write_signs(0, tid, 512);
  {
    const bool has_bias = p.mu_b != nullptr;
    const bool has_aff = (p.ep_scale != nullptr) || (p.ep_shift != nullptr);
    ep_fill_constants<KIND>(p, rl, ba_lds, tid, ntile, 0, has_bias, has_aff);
  }
  const bool has_ba = (p.mu_b != nullptr) || (p.ep_scale != nullptr) || (p.ep_shift != nullptr);
  const bool relu = p.ep_relu != 0;

  // ---- MFMA role: the wave owns pixels [64*w4, +64) of its group's half tile (2 conv rows, flattened (row, col))
  int eo[2];      // element offset of the pixel's window inside the patch
  int st_off[2];  // byte offset of the pixel's 8-byte piece (lane half h) in the store-side rows, chunk swizzle in bits
                  // 4-5 (the address of chunk q is st_off ^ (q << 4)); -1: the pixel does not exist
  uint32_t st_orow[2];  // s_out index of the pixel's first channel in half tile 0
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int pl = w4 * 64 + mi * 32 + l31;
    const bool ok = pl < 2 * Wo;
    const int plc = ok ? pl : 0;
    const int r = plc >= Wo ? 1 : 0;
    const int col = plc - r * Wo;
    eo[mi] = r * p.sh * RowE + col * p.sw * p.C;
    st_off[mi] = ok ? R_OFF + r * row_b + col * 64 + (((col >> 2) & 3) << 4) + h * 8 : -1;
    st_orow[mi] = (uint32_t)(((img * Ho + c0 + r) * Wo + col) * p.N + ntile * BN);
  }
  // ---- pool role: thread (of the group) = (pooled column gtid>>2, 8-channel chunk gtid&3 of the 32-channel half)
  bool pool_thread = gtid < Wq * 4;
  int pool_coff[3];  // byte offset of the chunk in a store-side row for conv columns 2pc-1, 2pc, 2pc+1 (-1: none)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int col = 2 * (gtid >> 2) - 1 + k;
    pool_coff[k] = (pool_thread && col >= 0 && col < Wo) ? col * 64 + (((gtid & 3) ^ ((col >> 2) & 3)) * 16) : -1;
  }
  // carry role: the thread owns chunk gtid&3 of columns gtid>>2 and (gtid>>2) + 64 of the carry rows (-1: none)
  int carry_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = (gtid >> 2) + 64 * j;
    carry_off[j] = col < Wo ? col * 64 + (((gtid & 3) ^ ((col >> 2) & 3)) * 16) : -1;
  }
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (uint32_t)((size_t)p.NB * Hq * Wq * p.N * 2), 0x00020000);

  f32x16 accm[2][2], accd[2][2];

  // fragments of stage (kh, j) of half tile u: element offset st_e = kh*RowE + j*BK inside the lane's window
  auto load_frag = [&](StageFrag& f, int u, int st_e, int s, int base_e, auto mia_tag) __attribute__((always_inline)) {
    constexpr int MIA = decltype(mia_tag)::value;
    const unsigned char* as = smem + A_OFF + (u & 1) * p.pt_astage + st_e * 2;
    const unsigned char* ss = smem + S_OFF + (u & 1) * p.st_sbytes;
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

  // =================== K role: the stages of one half tile, a workgroup barrier after each of the phase's first three
  // steps (the store group runs its four steps beside it) =========================================================
  auto run_k = [&](int u, auto mia_tag) __attribute__((always_inline)) {
    constexpr int MIA = decltype(mia_tag)::value;
    const int base_e = tile_base_e(u);
    int l_j = 0, l_rowE = 0, l_e = 0;  // stage being loaded: element offset of (kernel row, stage within the row)
    auto advance_load = [&]() __attribute__((always_inline)) {
      l_e += BK;
      if (++l_j == spr) { l_j = 0; l_rowE += RowE; l_e = l_rowE; }
    };
    StageFrag fa, fb;
    load_frag(fa, u, 0, 0, base_e, mia_tag);
    advance_load();
    int nb = 0, nextb = cs;
    auto iter = [&](int s, StageFrag& cur, StageFrag& nxt, auto zero_tag) __attribute__((always_inline)) {
      constexpr bool ZERO = decltype(zero_tag)::value;
      if (s == nextb && nb < 3) { SP_BARRIER(); ++nb; nextb += cs; }
      DeltaFrag dfrag;
      load_delta<KIND>(dfrag, smem + W_OFF + s * DW_STAGE, l31, h);
      if (s + 1 < nstages) { load_frag(nxt, u, l_e, s + 1, base_e, mia_tag); advance_load(); }
      stage_mma<1, KIND, 2, MIA, ZERO>(cur, dfrag, accm, accd, l31, h);
    };
    iter(0, fa, fb, std::true_type{});
    int s = 1;
    for (; s + 1 < nstages; s += 2) {
      iter(s, fb, fa, std::false_type{});
      iter(s + 1, fa, fb, std::false_type{});
    }
    if (s < nstages) iter(s, fb, fa, std::false_type{});
    for (; nb < 3; ++nb) SP_BARRIER();
  };

  // =================== store role ================================================================================
  // fragments -> LDS rows r0 / r1, channel half NI (btx_epilogue.h stage 1 + the bf16 rounding of its stage 2; the ReLU
  // follows the pool)
  auto stage_half = [&](int u, int mia, auto ni_tag, auto ba_tag) __attribute__((always_inline)) {
    constexpr int ni = decltype(ni_tag)::value;
    constexpr bool BA = decltype(ba_tag)::value;
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
          v[rr] = val;
        }
        if (wr[mi])
          *(u32x2*)(smem + (st_off[mi] ^ (q << 4))) = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
      }
    }
  };
  // LDS rows (+ the carry row of the previous half tile) -> pooled row P0+u-1, channel half ni; writes the carry row of
  // this half tile, max(r0, r1), for the next one (the other parity's row: nobody reads it during this phase)
  auto pool_half = [&](int u, int ni, auto relu_tag) __attribute__((always_inline)) {
    constexpr bool RL = decltype(relu_tag)::value;
    const uint32_t ninf = RL ? 0x80008000u : 0xff80ff80u;  // below everything: most negative int16 pair | -inf pair
    const int prow = P0 + u - 1;
    const int cr0 = c0 + SP_HROWS * u;
    const bool r0_ok = cr0 >= 0 && cr0 < Ho, r1_ok = (cr0 + 1) < Ho && u + 1 < NH;  // the closing half tile has no r1
    const bool ok = pool_thread && u >= 1 && prow < Hq;
    const unsigned char* rows = smem + R_OFF;
    const unsigned char* carry = smem + C_OFF + (2 * ni + ((u + 1) & 1)) * row_b;  // written by half tile u-1
    u32x4 m = {ninf, ninf, ninf, ninf};
    if (ok) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (pool_coff[k] >= 0) {
          m = sp_max8<RL>(m, *(const u32x4*)(carry + pool_coff[k]));  // rows that do not exist were folded in as `ninf`
          if (r0_ok) m = sp_max8<RL>(m, *(const u32x4*)(rows + pool_coff[k]));
        }
      }
      if constexpr (RL) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        m = sp_max8<true>(m, z);
      }
    }
    const uint32_t off = ok ? (uint32_t)((((img * Hq + prow) * Wq + (gtid >> 2)) * p.N + ntile * BN + ni * 32 + (gtid & 3) * 8) * 2)
                            : DMA_OOB;
    __builtin_amdgcn_raw_buffer_store_b128(m, out_rsrc, off, 0, 0);
    unsigned char* cdst = smem + C_OFF + (2 * ni + (u & 1)) * row_b;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (carry_off[j] >= 0) {
        u32x4 c = {ninf, ninf, ninf, ninf};
        if (r0_ok) c = sp_max8<RL>(c, *(const u32x4*)(rows + carry_off[j]));
        if (r1_ok) c = sp_max8<RL>(c, *(const u32x4*)(rows + row_b + carry_off[j]));
        *(u32x4*)(cdst + carry_off[j]) = c;
      }
    }
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
    // both roles (~100 VGPRs) stay live across the K loops and the accumulators spill
    asm volatile("" : "+v"(lane), "+v"(l31), "+v"(h), "+v"(gtid), "+v"(eo[0]), "+v"(eo[1]), "+v"(st_off[0]), "+v"(st_off[1]));
    asm volatile("" : "+v"(st_orow[0]), "+v"(st_orow[1]), "+v"(pool_coff[0]), "+v"(pool_coff[1]), "+v"(pool_coff[2]),
                      "+v"(carry_off[0]), "+v"(carry_off[1]));
    if (grp == (ph & 1)) {
      // ---------------- K role: half tile ph
      if (ph < NH) {
        if (ph + 1 < NH) issue_patch(ph + 1, w4, 4);
        const int mia = mia_of(ph);
        if (mia == 2) run_k(ph, std::integral_constant<int, 2>{});
        else if (mia == 1) { clear_acc(1); run_k(ph, std::integral_constant<int, 1>{}); }
        else { clear_acc(0); SP_BARRIER(); SP_BARRIER(); SP_BARRIER(); }
        SP_T(tr_k)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next patch: issued a whole K loop ago
      } else {
        clear_acc(0);
        SP_BARRIER(); SP_BARRIER(); SP_BARRIER();
      }
    } else {
      // ---------------- store role: half tile ph-1 (this group's accumulators of the previous phase)
      const int u = ph - 1;
      if (ph + 1 < NH) write_signs(ph + 1, gtid, 256);
      if (u >= 0) {
        const int mia = mia_of(u);
        if (has_ba) stage_half(u, mia, std::integral_constant<int, 0>{}, T{});
        else stage_half(u, mia, std::integral_constant<int, 0>{}, F{});
        SP_T(tr_st)
        SP_BARRIER();
        SP_T(tr_bar)
        if (relu) pool_half(u, 0, T{}); else pool_half(u, 0, F{});
        SP_T(tr_pool)
        SP_BARRIER();
        SP_T(tr_bar)
        if (has_ba) stage_half(u, mia, std::integral_constant<int, 1>{}, T{});
        else stage_half(u, mia, std::integral_constant<int, 1>{}, F{});
        SP_T(tr_st)
        SP_BARRIER();
        SP_T(tr_bar)
        if (relu) pool_half(u, 1, T{}); else pool_half(u, 1, F{});
        SP_T(tr_pool)
      } else {
        SP_BARRIER(); SP_BARRIER(); SP_BARRIER();
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
      tr[0] = tr_pro; tr[1] = 0; tr[2] = tr_k; tr[3] = tr_st; tr[4] = tr_pool; tr[5] = tr_t3 - tr_t0; tr[6] = tr_bar;
      tr[7] = tr_t0;
    }
  }
#endif
}
#undef SP_T
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
