// btx_mma.h — one K-stage of the sample-and-contract MFMA block, shared by the patch and the stem kernels (gfx950).
#pragma once
#include "btx_contract.h"

#ifndef BTX_PT_ABL
#define BTX_PT_ABL 0  // measurement-only ablation bits: 1 no MFMA, 2 no LDS fragment reads, 4 no DMA in the loop,
#endif                // 8 no per-stage barrier, 16 no sign masks, 32 no epilogue

namespace btx {

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
template <int PREC, int KIND, int MI = 2, int MIA = MI, bool ZERO = false>
__device__ __forceinline__ void stage_mma(StageFragT<MI>& f, const DeltaFrag& dfrag, f32x16 (&accm)[MI][2],
                                          f32x16 (&accd)[MI][2], int l31, int h) {
    const u32x4 (&wd)[NG / 2][2] = dfrag.w;
#ifdef BTX_MMA_PRIO
    __builtin_amdgcn_s_setprio(BTX_MMA_PRIO);
#endif
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
          if constexpr (!(BTX_PT_ABL & 16)) {
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
