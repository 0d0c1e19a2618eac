// btx_mma.h — one K-stage of the sample-and-contract MFMA block, shared by the patch and the stem kernels (gfx950).
#pragma once
#include "btx_contract.h"

#ifndef BTX_PT_ABL
#define BTX_PT_ABL 0  // measurement-only ablation bits: 1 no MFMA, 2 no LDS fragment reads, 4 no DMA in the loop,
#endif                // 8 no per-stage barrier, 16 no sign masks, 32 no epilogue

namespace btx {

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// ---- split-bf16 ("bf16x3", PREC == 2) -------------------------------------------------------------------------------
// The reference contracts in f32 (F.conv2d on f32 tensors, layers/flipout_layers/conv_flipout.py:376-382, 408-417).  A
// product of two f32 values x = xh + xl, w = wh + wl (h = the value rounded to bf16, l = the remainder rounded to bf16:
// 16 of the 24 mantissa bits) is wh*xh + wh*xl + wl*xh up to 2^-16 relative: three v_mfma_f32_32x32x16_bf16 per tile and
// K-step instead of the eight v_mfma_f32_32x32x2_f32 of the exact mode — 1/3 of the bf16 matrix rate (833 TFLOP/s peak
// equivalent against 157) at a per-layer rel-L2 of ~1e-6 (bar 1e-4, north_star's tolerance).  LDS images and K order are
// those of the f32 mode (16-byte granules of 4 k, 16 k per stage): activations are f32 in HBM and in LDS and are split
// when a fragment is read; the weight tiles are pre-split by the sampling pre-pass (btx_presample.h: a granule holds
// [4 bf16 hi | 4 bf16 lo] of its 4 k), so their fragments need no VALU at all.
// One K-step of the 32x32x16 MFMA per stage: the lane's 8 k are granule rows h and 2+h.
__device__ __forceinline__ void split_bf16_pair(uint32_t x0, uint32_t x1, uint32_t& hi, uint32_t& lo) {
#pragma clang fp contract(off)
  const f32x2 v = {u2f(x0), u2f(x1)};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32 (round to nearest even)
  const f32x2 r = {v[0] - u2f(hi << 16), v[1] - u2f(hi & 0xffff0000u)};   // exact in f32
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
}
// 8 f32 (two granules) -> 8 bf16 hi + 8 bf16 lo in MFMA operand order
__device__ __forceinline__ void split_bf16_granules(const u32x4& g0, const u32x4& g1, u32x4& hi, u32x4& lo) {
  uint32_t h_, l_;
  split_bf16_pair(g0[0], g0[1], h_, l_); hi[0] = h_; lo[0] = l_;
  split_bf16_pair(g0[2], g0[3], h_, l_); hi[1] = h_; lo[1] = l_;
  split_bf16_pair(g1[0], g1[1], h_, l_); hi[2] = h_; lo[2] = l_;
  split_bf16_pair(g1[2], g1[3], h_, l_); hi[3] = h_; lo[3] = l_;
}
__device__ __forceinline__ f32x16 mfma3(const u32x4& wh, const u32x4& wl, const u32x4& ah, const u32x4& al, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl), __builtin_bit_cast(bf16x8, ah), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, al), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, ah), c, 0, 0, 0);
}

// Fragments of one K-stage held in registers: activations a[kk][mi], mean weights wm[kk][ni], the sign word of each of
// the lane's two pixels (32 bf16 / 16 f32 elements of the stage, bit of element e at ((e&1) ? 31 : 15) - (e>>1)).
template <int MI>
struct StageFragT {
  u32x4 a[NG / 2][MI], wm[NG / 2][2];
  uint32_t sw[MI];
};
using StageFrag = StageFragT<2>;

// delta-weight fragments of a stage (Flipout), read from the LDS tile `ws` (mu at +0, delta at +NG*BN*16)
struct DeltaFrag {
  u32x4 w[NG / 2][2];
};
template <int KIND>
__device__ __forceinline__ void load_delta(DeltaFrag& d, const unsigned char* ws, int l31, int h) {
  if constexpr (KIND == 1) {
    if constexpr (BTX_PT_ABL & 2) {
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk) d.w[kk][0] = d.w[kk][1] = (u32x4){7u, 7u, 1u, 4u};
    } else {
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          d.w[kk][ni] = *(const u32x4*)(ws + NG * BN * 16 + ((2 * kk + h) * BN + ni * 32 + l31) * 16);
    }
  }
}

// mean MFMAs from the fragment registers, then the activations get their s_in signs (XOR mask) and the delta MFMAs
// follow.  The caller issues load_delta() BEFORE any LDS read it wants to stay in flight across this call (the fragment
// prefetch of the next stage): LDS returns in order, so delta fragments queued behind a prefetch would make the delta
// MFMAs wait for data they do not need.
// MIA = the wave's 32-pixel tiles that hold real pixels (<= MI): the tiles behind them are tile padding and are skipped.
// ZERO (bf16): first stage of a tile — the accumulators start from the MFMA's inline zero operand instead of 128 v_mov.
// MASK == false (the wide Reparameterization tile, btx_contract_taps.h): the second weight tile is the next n-tile of the same
// sampled weights — same activations, no signs.
template <int PREC, int KIND, int MI = 2, int MIA = MI, bool ZERO = false, bool MASK = true>
__device__ __forceinline__ void stage_mma(StageFragT<MI>& f, const DeltaFrag& dfrag, f32x16 (&accm)[MI][2],
                                          f32x16 (&accd)[MI][2], int l31, int h) {
    const u32x4 (&wd)[NG / 2][2] = dfrag.w;
#ifdef BTX_MMA_PRIO
    __builtin_amdgcn_s_setprio(BTX_MMA_PRIO);
#endif
    if constexpr (PREC == 2) {
      static_assert(NG == 4, "split-bf16: one 32x32x16 K-step per 16-k stage");
      // weight fragments: granule = [hi k0..3 | lo k0..3]; rows h and 2+h make the lane's 8 k
      u32x4 ah[MI], al[MI];
#pragma unroll
      for (int mi = 0; mi < MIA; ++mi) split_bf16_granules(f.a[0][mi], f.a[1][mi], ah[mi], al[mi]);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const u32x4 wh = {f.wm[0][ni][0], f.wm[0][ni][1], f.wm[1][ni][0], f.wm[1][ni][1]};
        const u32x4 wl = {f.wm[0][ni][2], f.wm[0][ni][3], f.wm[1][ni][2], f.wm[1][ni][3]};
#pragma unroll
        for (int mi = 0; mi < MIA; ++mi) accm[mi][ni] = mfma3(wh, wl, ah[mi], al[mi], accm[mi][ni]);
      }
      if constexpr (KIND == 1) {
        // s_in: bit of element e of granule row r sits 2r + (e>>1) + (e odd ? 0 : 16) below the word's top bit, i.e. the
        // packed pair j of row r takes (sw << (2r + j)) & 0x80008000 — the same mask on the hi and the lo half (rounding
        // to nearest is symmetric, so the split of -x is the negated split of x)
#pragma unroll
        for (int mi = 0; mi < MIA; ++mi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t m = (f.sw[mi] << (2 * ((r >> 1) * 2 + h) + (r & 1))) & 0x80008000u;
            ah[mi][r] ^= m;
            al[mi][r] ^= m;
          }
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const u32x4 dh = {wd[0][ni][0], wd[0][ni][1], wd[1][ni][0], wd[1][ni][1]};
          const u32x4 dl = {wd[0][ni][2], wd[0][ni][3], wd[1][ni][2], wd[1][ni][3]};
#pragma unroll
          for (int mi = 0; mi < MIA; ++mi) accd[mi][ni] = mfma3(dh, dl, ah[mi], al[mi], accd[mi][ni]);
        }
      }
#ifdef BTX_MMA_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      return;
    }
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      if constexpr (PREC == 1) {
#pragma unroll
        for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if constexpr (BTX_PT_ABL & 1) { asm volatile("" ::"v"(f.wm[kk][ni]), "v"(f.a[kk][mi])); accm[mi][ni][0] += 1.f; }
            else if constexpr (ZERO) {
              const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              accm[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, f.wm[kk][ni]), __builtin_bit_cast(bf16x8, f.a[kk][mi]), kk == 0 ? zc : accm[mi][ni], 0, 0, 0);
            } else accm[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, f.wm[kk][ni]), __builtin_bit_cast(bf16x8, f.a[kk][mi]), accm[mi][ni], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              accm[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(f.wm[kk][ni][e]), u2f(f.a[kk][mi][e]), accm[mi][ni], 0, 0, 0);
      }
    }
    if constexpr (KIND == 1) {
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk) {
        const int row = 2 * kk + h;
        if constexpr (PREC == 1) {
          if constexpr (MASK && !(BTX_PT_ABL & 16)) {
#pragma unroll
            for (int mi = 0; mi < MIA; ++mi) {
              const uint32_t swr = f.sw[mi] << (4 * row);
#pragma unroll
              for (int d = 0; d < 4; ++d) f.a[kk][mi][d] ^= ((swr << d) & 0x80008000u);
            }
          }
#pragma unroll
          for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              if constexpr (BTX_PT_ABL & 1) { asm volatile("" ::"v"(wd[kk][ni]), "v"(f.a[kk][mi])); accd[mi][ni][0] += 1.f; }
              else if constexpr (ZERO) {
                const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                accd[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, wd[kk][ni]), __builtin_bit_cast(bf16x8, f.a[kk][mi]), kk == 0 ? zc : accd[mi][ni], 0, 0, 0);
              } else accd[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, wd[kk][ni]), __builtin_bit_cast(bf16x8, f.a[kk][mi]), accd[mi][ni], 0, 0, 0);
            }
        } else {
#pragma unroll
          for (int mi = 0; mi < MIA; ++mi) {
            const uint32_t swr = f.sw[mi] << (2 * row);
#pragma unroll
            for (int e = 0; e < 4; ++e) f.a[kk][mi][e] ^= ((swr << ((e >> 1) + ((e & 1) ? 0 : 16))) & 0x80000000u);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                accd[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(wd[kk][ni][e]), u2f(f.a[kk][mi][e]), accd[mi][ni], 0, 0, 0);
        }
      }
    }
#ifdef BTX_MMA_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

}  // namespace btx
