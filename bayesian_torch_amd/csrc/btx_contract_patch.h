// btx_contract_patch.h — "patch" variant of the fused sample-and-contract kernel for stride-1 2-D convolutions (gfx950).
//
// The LDS-DMA kernel (btx_contract_dma.h) re-reads every input pixel once per filter tap (9x for 3x3) and fetches
// each 128-byte line twice (a K-stage uses a 64-byte half): its L2->LDS pipeline alone costs ~80 us on the ResNet18
// layer1 shape (ablation).  Here the workgroup owns a 2-D tile of output pixels — G images x R rows x
// the full output width, <= 512 pixels — and, per 32-channel (bf16) / 16-channel (f32) block of the input:
//
//   * the halo'd input PATCH of the tile, (R + (KH-1)dh) x (Wo + (KW-1)dw) pixels x 64 bytes, is DMA'd into LDS
//     ONCE (two-slot ring; the next block's patch is fetched piecewise, one 1-KiB DMA instruction per wave per
//     stage, while the current block's KH*KW stages run), zero-filled outside the image by the buffer descriptor;
//   * all KH*KW taps are multiplied out of that patch: the MFMA fragment address is patch_pixel(lane) + a
//     wave-uniform tap offset.  The pixel-major image with the XOR swizzle slot = granule ^ ((q>>2)&3) stays
//     conflict-free for ANY window of 16 consecutive patch pixels, so shifted windows cost nothing;
//   * s_in is hashed once per patch pixel per channel block (not once per tap), bounds are checked once per patch
//     pixel per workgroup;
//   * weights: unchanged (raw f32 (mu,rho) quads by DMA, sampled by the fetching wave, bf16/f32 tile in LDS).
//
// K order is (channel block, tap) instead of (tap, channel block); the element index k = tap*Cg + c used for the
// weights and for BTX-RNG is the same, so the noise — and up to summation order the result — is identical to the
// other variants.  Split-K splits the channel blocks.
#pragma once
#include <type_traits>
#include "btx_contract.h"
#include "btx_contract_dma.h"

namespace btx {

constexpr int PT_A_STAGE = PT_PPMAX * 64;          // 53248
constexpr int PT_S_STAGE = PT_PPMAX * 4;           // 3328
constexpr int PT_NW = NTHREADS / 64;
constexpr int PT_WD = 4;                           // depth of the weight-tile ring
constexpr int PT_A_OFF = 0;                        // 2 patch slots
constexpr int PT_S_OFF = PT_A_OFF + 2 * PT_A_STAGE;    // 106496
constexpr int PT_W_OFF = PT_S_OFF + 2 * PT_S_STAGE;    // 113152 : weight tiles, 4 x 8 KiB (mu at +0, delta at +4096)
constexpr int PT_LDS_BYTES = PT_W_OFF + PT_WD * DW_STAGE;  // 145920 <= 163840

// ---- sampling pre-pass --------------------------------------------------------------------------------------
// Every workgroup of a convolution needs the same sampled weight tile: with 392 pixel tiles (ResNet18 layer1) the
// in-kernel sampler of the other variants repeats each Philox/Box-Muller/softplus 392 times, and that — not the
// memory pipeline — is what bounds them (A/B in DESIGN.md §5).  Here the weights are sampled ONCE per launch into
// MFMA-ready tiles in the workspace:   wt[tile = group*ntiles + ntile][kg = k/G][ch 0..63][G elements]  (16-byte
// granules; `mu` array, then the `delta` = sigma*eps array for Flipout; Reparameterization stores mu + sigma*eps in the
// first array).  One K-stage of a workgroup is then 4 (+4) contiguous 1-KiB rows: one LDS-DMA instruction each.
// Same element indices, same _hw sampling functions and the same rounding as the in-kernel sampler: the values are
// bit-identical to what the other variants compute.
template <int PREC, int KIND>
__global__ __launch_bounds__(256) void presample_kernel(const float* __restrict__ mu, const float* __restrict__ rho,
                                                        unsigned char* __restrict__ wt, uint32_t delta_off, int Ng,
                                                        int K, int ntiles, uint32_t nquads_total, uint32_t seed_lo,
                                                        uint32_t seed_hi, uint32_t sample, uint32_t layer) {
  constexpr int G = (PREC == 1) ? 8 : 4;
  const uint32_t kq = (uint32_t)K >> 2;
  for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < nquads_total; t += gridDim.x * 256u) {
    const uint32_t quad = t % kq, np = t / kq;
    const int ch = np & 63, tile = np >> 6;
    const int group = tile / ntiles, ntile = tile - group * ntiles;
    const int col = ntile * BN + ch;
    float wm[4] = {0.f, 0.f, 0.f, 0.f}, wd[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < Ng) {
      const uint32_t e0 = (uint32_t)(group * Ng + col) * (uint32_t)K + 4u * quad;
      const f32x4 mu4 = *(const f32x4*)(mu + e0);
      const f32x4 rho4 = *(const f32x4*)(rho + e0);
      float eps[4];
      btx_normal4_hw(e0 >> 2, sample, layer, 0u, seed_lo, seed_hi, eps);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sg = btx_softplus_hw(rho4[e]);
        if constexpr (KIND == 0) wm[e] = __builtin_fmaf(sg, eps[e], mu4[e]);
        else { wm[e] = mu4[e]; wd[e] = sg * eps[e]; }
      }
    }
    const uint32_t kg = (4u * quad) / G;
    const uint32_t o = (((uint32_t)tile * ((uint32_t)K / G) + kg) * 64u + (uint32_t)ch) * 16u;
    if constexpr (PREC == 1) {
      const uint32_t oo = o + (quad & 1u) * 8u;
      *(u32x2*)(wt + oo) = pack_quad_bf16(wm);
      if constexpr (KIND == 1) *(u32x2*)(wt + delta_off + oo) = pack_quad_bf16(wd);
    } else {
      *(u32x4*)(wt + o) = (u32x4){f2u(wm[0]), f2u(wm[1]), f2u(wm[2]), f2u(wm[3])};
      if constexpr (KIND == 1) *(u32x4*)(wt + delta_off + o) = (u32x4){f2u(wd[0]), f2u(wd[1]), f2u(wd[2]), f2u(wd[3])};
    }
  }
}

#ifndef BTX_PT_ABL
#define BTX_PT_ABL 0  // measurement-only ablation bits: 1 no MFMA, 2 no LDS fragment reads, 4 no DMA in the loop,
#endif                // 8 no per-stage barrier, 16 no sign masks, 32 no epilogue
#define BTX_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt(int n) {  // n is wave-uniform
  switch (n) {
    BTX_VMCNT_CASE(0) BTX_VMCNT_CASE(1) BTX_VMCNT_CASE(2) BTX_VMCNT_CASE(3) BTX_VMCNT_CASE(4) BTX_VMCNT_CASE(5)
    BTX_VMCNT_CASE(6) BTX_VMCNT_CASE(7) BTX_VMCNT_CASE(8) BTX_VMCNT_CASE(9) BTX_VMCNT_CASE(10) BTX_VMCNT_CASE(11)
    BTX_VMCNT_CASE(12) BTX_VMCNT_CASE(13) BTX_VMCNT_CASE(14) BTX_VMCNT_CASE(15) BTX_VMCNT_CASE(16)
    default: if (n > 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
#undef BTX_VMCNT_CASE

// ContractParams fields used in addition: pt_G, pt_R, pt_Rp, pt_Wp, pt_PP, pt_NI, pt_rtiles, wt (pre-sampled weight
// tiles), wt_bytes, wt_delta_off; kper = channel blocks per split * BK.
template <int PREC, int KIND>
__global__ __launch_bounds__(NTHREADS, 2) void contract_patch_kernel(const ContractParams p) {
  using ACT = typename std::conditional<PREC == 1, __bf16, float>::type;
  constexpr int G = (PREC == 1) ? 8 : 4;
  constexpr int BK = NG * G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int logical;
  {
    const int nwg = gridDim.x, L = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = L & 7, slot = L >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int inner = p.ntiles * p.groups * p.ksplits;
  const int mtile = logical / inner;
  int rem = logical - mtile * inner;
  const int split = rem % p.ksplits;
  rem /= p.ksplits;
  const int ntile = rem % p.ntiles;
  const int group = rem / p.ntiles;

  // tile geometry
  const int ig = mtile / p.pt_rtiles;                 // image group
  const int rt = mtile - ig * p.pt_rtiles;            // row tile
  const int img0 = ig * p.pt_G, row0 = rt * p.pt_R;   // first image / first output row of the tile
  const int T = p.KH * p.KW;
  const int ncb_total = p.Cg / BK;
  const int cb_per = p.kper / BK;
  const int cb_begin = split * cb_per;
  const int cb_end = min(ncb_total, cb_begin + cb_per);
  const int ncb = cb_end - cb_begin;
  const int nstages = ncb * T;
  const int esz = (int)sizeof(ACT);

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);

  // ---- patch loader role: DMA instruction i (i = wave + 8*j, j < pt_NI) moves patch pixels 16i + (lane>>2),
  //      granule slot lane&3 (source-side swizzle as in the DMA variant).  Byte offset of channel block 0, or OOB.
  const int g_lane = (lane & 3) ^ ((lane >> 4) & 3);
  uint32_t pp_boff[PT_MAXNI];
#pragma unroll
  for (int j = 0; j < PT_MAXNI; ++j) {
    const int q = 16 * (wave + 8 * j) + (lane >> 2);
    uint32_t bo = DMA_OOB;
    if (j < p.pt_NI && q < p.pt_PP) {
      const int pc = q % p.pt_Wp;
      const int t = q / p.pt_Wp;
      const int pr = t % p.pt_Rp;
      const int gi = t / p.pt_Rp;
      const int img = img0 + gi, ih = row0 + pr - p.ph, iw = pc - p.pw;
      if (img < p.NB && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
        bo = ((uint32_t)((img * p.H + ih) * p.W + iw) * (uint32_t)p.C + (uint32_t)(group * p.Cg + G * g_lane)) *
             (uint32_t)esz;
    }
    pp_boff[j] = bo;
  }
  // ---- sign role: thread t hashes the words of patch pixels t and t+512 (element offset of channel block 0)
  uint32_t sg_off[2];
  bool sg_ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = tid + 512 * j;
    sg_ok[j] = q < p.pt_PP;
    const int qq = sg_ok[j] ? q : 0;
    const int pc = qq % p.pt_Wp;
    const int t = qq / p.pt_Wp;
    const int pr = t % p.pt_Rp;
    const int gi = t / p.pt_Rp;
    sg_off[j] = (uint32_t)(((img0 + gi) * p.H + (row0 + pr - p.ph)) * p.W + (pc - p.pw)) * (uint32_t)p.C +
                (uint32_t)(group * p.Cg);
  }
  // ---- MFMA role: wave owns output pixels [64*wave, +64) of the tile, flattened (image, row, col)
  int q0[2];       // patch pixel of tap (0,0) for this lane's two output pixels
  int out_m[2];    // global output pixel index or -1
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int pl = wave * 64 + mi * 32 + l31;
    const int c = pl % p.Wo;
    const int t = pl / p.Wo;
    const int r = t % p.pt_R;
    const int gi = t / p.pt_R;
    const bool ok = (gi < p.pt_G) && (img0 + gi < p.NB) && (row0 + r < p.Ho);
    q0[mi] = ok ? (gi * p.pt_Rp + r) * p.pt_Wp + c : 0;
    out_m[mi] = ok ? ((img0 + gi) * p.Ho + row0 + r) * p.Wo + c : -1;
  }
  // ---- weight loader role: wave w fetches row (w & 3) of the mu tile (w < 4) or of the delta tile (w >= 4, Flipout)
  const bool w_wave = (KIND == 1) || (wave < 4);
  const uint32_t w_base = (uint32_t)(group * p.ntiles + ntile) * (uint32_t)(p.K / G) * 1024u + (uint32_t)lane * 16u +
                          (wave >= 4 ? p.wt_delta_off : 0u) + (uint32_t)(wave & 3) * 1024u;
  const int w_lds = PT_W_OFF + (wave >= 4 ? 4096 : 0) + (wave & 3) * 1024;

  // Stage s <-> (channel block cb_begin + s / T, tap s % T); first k of the stage = tap*Cg + cb*BK.  Both walks over
  // the stages (the weight fetch, three stages ahead, and the multiply) keep their position incrementally.
  int wi_s = 0, wi_t = 0, wi_cbk = cb_begin * BK, wi_k0 = cb_begin * BK;  // next stage whose weights get fetched
  auto issue_w_next = [&]() __attribute__((always_inline)) {
    if (w_wave)
      dma16(wt_rsrc, w_base + (uint32_t)(wi_k0 / G) * 1024u, smem + w_lds + (wi_s & (PT_WD - 1)) * DW_STAGE);
    ++wi_s;
    wi_k0 += p.Cg;
    if (++wi_t == T) { wi_t = 0; wi_cbk += BK; wi_k0 = wi_cbk; }
  };
  // one 1-KiB piece (16 patch pixels x 64 B) of the patch of channel block `cbi` (index relative to cb_begin)
  auto issue_patch_piece = [&](int cbi, int j) __attribute__((always_inline)) {
    const uint32_t cboff = (uint32_t)((cb_begin + cbi) * BK * esz);
    unsigned char* as = smem + PT_A_OFF + (cbi & 1) * PT_A_STAGE + (wave + 8 * j) * 1024;
    uint32_t bo = DMA_OOB;
#pragma unroll
    for (int jj = 0; jj < PT_MAXNI; ++jj)
      if (jj == j) bo = pp_boff[jj];
    dma16(x_rsrc, bo == DMA_OOB ? DMA_OOB : bo + cboff, as);
  };
  auto write_signs = [&](int cbi) __attribute__((always_inline)) {
    if constexpr (KIND == 1) {
      unsigned char* ss = smem + PT_S_OFF + (cbi & 1) * PT_S_STAGE;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (sg_ok[j]) {
          const uint32_t off = sg_off[j] + (uint32_t)((cb_begin + cbi) * BK);
          uint32_t w = btx_sign_word(off >> 5, p.kin_a, p.kin_b);
          if constexpr (G == 4) w <<= 8 * ((off >> 4) & 1);
          *(uint32_t*)(ss + (tid + 512 * j) * 4) = w;
        }
      }
    }
  };
  f32x16 accm[2][2], accd[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }

  // one K-stage: every LDS fragment of the stage is requested up front (12 x ds_read_b128 + the two sign words), the
  // MFMAs follow in the order the data arrives
  auto mma_stage = [&](int s, int cbi, int toff) __attribute__((always_inline)) {
    const unsigned char* as = smem + PT_A_OFF + (cbi & 1) * PT_A_STAGE;
    const unsigned char* ss = smem + PT_S_OFF + (cbi & 1) * PT_S_STAGE;
    const unsigned char* ws = smem + PT_W_OFF + (s & (PT_WD - 1)) * DW_STAGE;
    u32x4 a[NG / 2][2], wm[NG / 2][2], wd[NG / 2][2];
    uint32_t sw[2];
    int q[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) q[mi] = q0[mi] + toff;
    if constexpr (BTX_PT_ABL & 2) {
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[kk][i] = (u32x4){(uint32_t)q[0], (uint32_t)kk, 3u, 4u};
          wm[kk][i] = wd[kk][i] = (u32x4){(uint32_t)q[1], 7u, (uint32_t)s, 4u};
        }
      sw[0] = sw[1] = (uint32_t)s;
    } else {
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk) {
        const int row = 2 * kk + h;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[kk][mi] = *(const u32x4*)(as + q[mi] * 64 + ((row ^ ((q[mi] >> 2) & 3)) * 16));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) wm[kk][ni] = *(const u32x4*)(ws + (row * BN + ni * 32 + l31) * 16);
      }
      if constexpr (KIND == 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) sw[mi] = *(const uint32_t*)(ss + q[mi] * 4);
#pragma unroll
        for (int kk = 0; kk < NG / 2; ++kk)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            wd[kk][ni] = *(const u32x4*)(ws + NG * BN * 16 + ((2 * kk + h) * BN + ni * 32 + l31) * 16);
      }
    }
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      const int row = 2 * kk + h;
      if constexpr (PREC == 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if constexpr (BTX_PT_ABL & 1) { asm volatile("" ::"v"(wm[kk][ni]), "v"(a[kk][mi])); accm[mi][ni][0] += 1.f; }
            else accm[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, wm[kk][ni]), __builtin_bit_cast(bf16x8, a[kk][mi]), accm[mi][ni], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              accm[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(wm[kk][ni][e]), u2f(a[kk][mi][e]), accm[mi][ni], 0, 0, 0);
      }
    }
    if constexpr (KIND == 1) {
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk) {
        const int row = 2 * kk + h;
        if constexpr (PREC == 1) {
          if constexpr (!(BTX_PT_ABL & 16)) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
              const uint32_t swr = sw[mi] << (4 * row);
#pragma unroll
              for (int d = 0; d < 4; ++d) a[kk][mi][d] ^= ((swr << d) & 0x80008000u);
            }
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              if constexpr (BTX_PT_ABL & 1) { asm volatile("" ::"v"(wd[kk][ni]), "v"(a[kk][mi])); accd[mi][ni][0] += 1.f; }
              else accd[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, wd[kk][ni]), __builtin_bit_cast(bf16x8, a[kk][mi]), accd[mi][ni], 0, 0, 0);
            }
        } else {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const uint32_t swr = sw[mi] << (2 * row);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[kk][mi][e] ^= ((swr << ((e >> 1) + ((e & 1) ? 0 : 16))) & 0x80000000u);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                accd[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(wd[kk][ni][e]), u2f(a[kk][mi][e]), accd[mi][ni], 0, 0, 0);
        }
      }
    }
  };

  // =================== main loop ==========================================================================
  // Per stage s (tap t of channel block cbi) every wave: issues W(s+3) and its share of the pieces of the NEXT block's
  // patch, multiplies stage s, then waits until (a) W(s+1) and (b) — on the last stage of a block — the next patch
  // have landed, and meets the others at one barrier.  vmcnt retires in order, so "landed" = at most as many
  // operations outstanding as this wave has issued after the one it needs; those counts are wave-uniform scalars.
  const int ppst = (p.pt_NI + (T > 2 ? T - 3 : 0)) / (T > 2 ? T - 2 : 1);  // pieces per stage: done 2 stages early
  if (nstages > 0) {
    for (int j = 0; j < p.pt_NI; ++j)
      if (16 * (wave + 8 * j) < p.pt_PP) issue_patch_piece(0, j);
    write_signs(0);
    for (int s = 0; s < PT_WD - 1 && s < nstages; ++s) issue_w_next();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int nissued = 0, m1 = 0, m2 = 0, mpiece = 0;  // marks: nissued right after W(s+1), W(s+2), the last patch piece
    int cbi = 0, t = 0, kw = 0, toff = 0, rowoff = 0;
    const int row_step = p.dh * p.pt_Wp;
    for (int s = 0; s < nstages; ++s) {
      const bool next_cb = cbi + 1 < ncb;
      int m3 = nissued;
      if (!(BTX_PT_ABL & 4) && wi_s < nstages) { issue_w_next(); if (w_wave) m3 = ++nissued; }
      if (!(BTX_PT_ABL & 4) && next_cb) {
        for (int j = t * ppst; j < (t + 1) * ppst && j < p.pt_NI; ++j)
          if (16 * (wave + 8 * j) < p.pt_PP) { issue_patch_piece(cbi + 1, j); mpiece = ++nissued; }
        if (t == 0) write_signs(cbi + 1);  // the sign slot of block cbi+1 was last read during block cbi-1
      }
      mma_stage(s, cbi, toff);
      int allowed = nissued - m1;
      if (t == T - 1 && next_cb) allowed = min(allowed, nissued - mpiece);
      wait_vmcnt(allowed);
      if constexpr (BTX_PT_ABL & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      m1 = m2; m2 = m3;
      toff += p.dw;
      if (++kw == p.KW) { kw = 0; rowoff += row_step; toff = rowoff; }
      if (++t == T) { t = 0; ++cbi; toff = 0; rowoff = 0; }
    }
  }
  if constexpr (BTX_PT_ABL & 32) return;

  // =================== epilogue ===========================================================================
  // Stage 1: bias, Flipout combine (s_out), BN affine on the MFMA fragments; the f32 tile of the wave (64 pixels x 64
  // channels) goes to LDS (272-byte pixel rows: conflict-free both ways).  Stage 2: every lane takes 8 consecutive
  // channels of one pixel, adds the residual, applies ReLU, converts and stores 16 (bf16) / 32 (f32) contiguous bytes:
  // 8 lanes cover a pixel's 64 channels, so each store instruction writes whole 128-byte lines.  (Storing straight
  // from the fragments scatters 8-byte pieces over 64 different lines per instruction.)
  constexpr int EP_ROW = 272;
  constexpr int EP_WAVE = 64 * EP_ROW;  // 17408
  static_assert(PT_NW * EP_WAVE + 1024 <= PT_LDS_BYTES, "epilogue staging does not fit");
  const bool to_partial = p.ksplits > 1;
  const bool has_bias = (split == 0) && (p.mu_b != nullptr);
  float* bias_lds = (float*)(smem + PT_NW * EP_WAVE);
  if (has_bias) {
    if (tid < BN) {
      const int col = ntile * BN + tid;
      float bm = 0.f, bdl = 0.f;
      if (col < p.Ng) {
        const int gcol = group * p.Ng + col;
        const float eb = p.eps_b ? p.eps_b[gcol]
                                 : btx_normal1((unsigned long long)gcol, p.sample, p.layer, 1u, p.seed_lo, p.seed_hi);
        const float sb_ = btx_softplus_fast(p.rho_b[gcol]);
        if constexpr (KIND == 0) { bm = __builtin_fmaf(sb_, eb, p.mu_b[gcol]); }
        else { bm = p.mu_b[gcol]; bdl = sb_ * eb; }
      }
      bias_lds[tid] = bm;
      bias_lds[BN + tid] = bdl;
    }
  }
  const bool has_aff = !to_partial && ((p.ep_scale != nullptr) || (p.ep_shift != nullptr));
  float* aff_lds = bias_lds + 2 * BN;
  if (has_aff) {
    if (tid < BN) {
      const int col = ntile * BN + tid;
      const int gcol = group * p.Ng + (col < p.Ng ? col : 0);
      aff_lds[tid] = p.ep_scale ? p.ep_scale[gcol] : 1.f;
      aff_lds[BN + tid] = p.ep_shift ? p.ep_shift[gcol] : 0.f;
    }
  }
  if (has_bias || has_aff) __syncthreads();
  unsigned char* ep = smem + wave * EP_WAVE;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const uint32_t orow = (uint32_t)(out_m[mi] < 0 ? 0 : out_m[mi]) * (uint32_t)p.N + (uint32_t)(group * p.Ng);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int colbase = ntile * BN + ni * 32;
      const uint32_t o0 = orow + colbase;
      const bool word_fast = (KIND == 1) && !p.sign_out && ((o0 & 31u) == 0) && (colbase + 32 <= p.Ng);
      uint32_t wout = 0;
      if (word_fast) wout = btx_sign_word(o0 >> 5, p.kout_a, p.kout_b);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cl = ni * 32 + 8 * q + 4 * h;
        const int c0 = ntile * BN + cl;
        float v[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int col = c0 + rr;
          float val = accm[mi][ni][4 * q + rr];
          if (has_bias) val += bias_lds[cl + rr];
          if constexpr (KIND == 1) {
            float dl = accd[mi][ni][4 * q + rr];
            if (has_bias) dl += bias_lds[BN + cl + rr];
            uint32_t flip = 0;
            if (word_fast) {
              const int bp = ((rr & 1) ? 31 : 15) - 4 * q - 2 * h - (rr >> 1);
              flip = (wout << (31 - bp)) & 0x80000000u;
            } else if (col < p.Ng && out_m[mi] >= 0) {
              if (p.sign_out) {
                flip = (p.sign_out[orow + col] < 0) ? 0x80000000u : 0u;
              } else {
                const uint32_t io = orow + col;
                const uint32_t w1 = btx_sign_word(io >> 5, p.kout_a, p.kout_b);
                flip = (w1 << (31 - btx_sign_bitpos(io & 31u))) & 0x80000000u;
              }
            }
            val += u2f(f2u(dl) ^ flip);
          }
          if (has_aff) val = __builtin_fmaf(val, aff_lds[cl + rr], aff_lds[BN + cl + rr]);
          v[rr] = val;
        }
        *(f32x4*)(ep + (mi * 32 + l31) * EP_ROW + cl * 4) = (f32x4){v[0], v[1], v[2], v[3]};
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area is private to the wave
  {
    // valid pixels of the tile are a prefix of its flattened (image, row, col) order, contiguous in the output
    const int nimg = min(p.pt_G, p.NB - img0), nrow = min(p.pt_R, p.Ho - row0);
    const int nvalid = nimg * nrow * p.Wo;
    const uint32_t m0 = (uint32_t)(img0 * p.Ho + row0) * (uint32_t)p.Wo;
    const int cg = lane & 7;
    const int col0 = ntile * BN + cg * 8;
    const int nv = min(8, p.Ng - col0);
    const bool relu = !to_partial && p.ep_relu;
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      const int pix = r8 * 8 + (lane >> 3);
      const int pl = wave * 64 + pix;
      if (pl >= nvalid || nv <= 0) continue;
      const f32x4 lo = *(const f32x4*)(ep + pix * EP_ROW + cg * 32);
      const f32x4 hi = *(const f32x4*)(ep + pix * EP_ROW + cg * 32 + 16);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const size_t idx = (size_t)(m0 + (uint32_t)pl) * (size_t)p.N + (size_t)(group * p.Ng + col0);
      if (to_partial) {
        float* dst = p.partial + (size_t)split * p.M * p.N + idx;
        if (nv == 8 && (idx & 3) == 0) {
          *(f32x4*)dst = lo;
          *(f32x4*)(dst + 4) = hi;
        } else {
          for (int j = 0; j < nv; ++j) dst[j] = v[j];
        }
      } else if (p.out_bf16) {
        __bf16* dst = (__bf16*)p.out + idx;
        const __bf16* res = p.ep_res ? (const __bf16*)p.ep_res + idx : nullptr;
        if (nv == 8 && (idx & 7) == 0) {
          if (res) {
            const bf16x8 rv = *(const bf16x8*)res;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
          }
          if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          const f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
          const bf16x4 b0 = __builtin_convertvector(x0, bf16x4), b1 = __builtin_convertvector(x1, bf16x4);
          u32x4 pk;
          const u32x2 p0 = __builtin_bit_cast(u32x2, b0), p1 = __builtin_bit_cast(u32x2, b1);
          pk[0] = p0[0]; pk[1] = p0[1]; pk[2] = p1[0]; pk[3] = p1[1];
          *(u32x4*)dst = pk;
        } else {
          for (int j = 0; j < nv; ++j) {
            float y = v[j] + (res ? (float)res[j] : 0.f);
            if (relu) y = fmaxf(y, 0.f);
            dst[j] = (__bf16)y;
          }
        }
      } else {
        float* dst = (float*)p.out + idx;
        const float* res = p.ep_res ? (const float*)p.ep_res + idx : nullptr;
        if (nv == 8 && (idx & 3) == 0) {
          if (res) {
            const f32x4 r0 = *(const f32x4*)res, r1 = *(const f32x4*)(res + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += r0[j]; v[4 + j] += r1[j]; }
          }
          if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
          *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
          for (int j = 0; j < nv; ++j) {
            float y = v[j] + (res ? res[j] : 0.f);
            if (relu) y = fmaxf(y, 0.f);
            dst[j] = y;
          }
        }
      }
    }
  }
}

template <int PREC>
static int launch_presample_impl(int kind, const ContractParams& p, hipStream_t st) {
  const uint32_t nq = (uint32_t)(p.groups * p.ntiles * 64) * ((uint32_t)p.K >> 2);
  uint32_t blocks = (nq + 255u) / 256u;
  if (blocks > 4096u) blocks = 4096u;
  if (kind == 0)
    hipLaunchKernelGGL((presample_kernel<PREC, 0>), dim3(blocks), dim3(256), 0, st, p.mu, p.rho, (unsigned char*)p.wt,
                       p.wt_delta_off, p.Ng, p.K, p.ntiles, nq, p.seed_lo, p.seed_hi, p.sample, p.layer);
  else
    hipLaunchKernelGGL((presample_kernel<PREC, 1>), dim3(blocks), dim3(256), 0, st, p.mu, p.rho, (unsigned char*)p.wt,
                       p.wt_delta_off, p.Ng, p.K, p.ntiles, nq, p.seed_lo, p.seed_hi, p.sample, p.layer);
  return (int)hipGetLastError();
}

template <int PREC>
static int launch_contract_patch_impl(int kind, const ContractParams& p, int nwg, hipStream_t st) {
#define BTX_LAUNCH_PT(KIND)                                                                                         \
  do {                                                                                                              \
    auto kfn = contract_patch_kernel<PREC, KIND>;                                                                   \
    static bool attr_done = false;                                                                                  \
    if (!attr_done) {                                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, PT_LDS_BYTES); \
      if (e != hipSuccess) return (int)e;                                                                           \
      attr_done = true;                                                                                             \
    }                                                                                                               \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(NTHREADS), PT_LDS_BYTES, st, p);                                        \
  } while (0)
  int rc = launch_presample_impl<PREC>(kind, p, st);
  if (rc) return rc;
  if (kind == 0) BTX_LAUNCH_PT(0); else BTX_LAUNCH_PT(1);
#undef BTX_LAUNCH_PT
  return (int)hipGetLastError();
}

}  // namespace btx
