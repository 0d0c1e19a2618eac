// btx_epilogue.h — the store side shared by the LDS-DMA and the patch variants of the fused contraction (gfx950).
//
// Stage 1: bias, Flipout combine (s_out), BN affine on the MFMA fragments; the f32 tile of the wave (64 pixels x 64
// channels) goes to LDS (272-byte pixel rows: conflict-free both ways).  Stage 2: every lane takes 8 consecutive
// channels of one pixel, adds the residual, applies ReLU, converts and stores 16 (bf16) / 32 (f32) contiguous bytes:
// 8 lanes cover a pixel's 64 channels, so each store instruction writes whole 128-byte lines.  (Storing straight from
// the fragments scatters 8-byte pieces over 64 different lines per instruction: measured ~2x the time of the whole
// K loop on the ResNet18 layer1 shape.)
//
// Contract: the wave's fragment pixel (mi, lane&31) is output pixel m0 + 64*wave + 32*mi + (lane&31); the first
// `nvalid` pixels of the block's tile exist.  Every wave of the block has passed a barrier after its last LDS read of
// the K loop; the block's dynamic LDS holds at least NW * PT_EP_WAVE + 1024 bytes.
#pragma once
#include <type_traits>
#include "btx_contract.h"

namespace btx {

// store 8 consecutive channels of one output pixel (stage 2); OUTK: 0 = f32 split-K partial, 1 = bf16 out, 2 = f32 out
template <int OUTK, bool RES, bool RELU>
__device__ __forceinline__ void ep_store8(const ContractParams& p, const f32x4 lo, const f32x4 hi, uint32_t idx,
                                          int split, int nv, bool vec) {
  float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  if constexpr (OUTK == 0) {
    float* dst = p.partial + (size_t)split * p.M * p.N + idx;
    if (vec) {
      *(f32x4*)dst = lo;
      *(f32x4*)(dst + 4) = hi;
    } else {
      for (int j = 0; j < nv; ++j) dst[j] = v[j];
    }
  } else if constexpr (OUTK == 1) {
    __bf16* dst = (__bf16*)p.out + idx;
    const __bf16* res = RES ? (const __bf16*)p.ep_res + idx : nullptr;
    if (vec) {
      if constexpr (RES) {
        const u32x4 rv = *(const u32x4*)res;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] += u2f(rv[j] << 16); v[2 * j + 1] += u2f(rv[j] & 0xffff0000u); }
      }
      if constexpr (RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      const f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
      const u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(x0, bf16x4));
      const u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(x1, bf16x4));
      *(u32x4*)dst = (u32x4){p0[0], p0[1], p1[0], p1[1]};
    } else {
      for (int j = 0; j < nv; ++j) {
        float y = v[j] + (RES ? (float)res[j] : 0.f);
        if (RELU) y = fmaxf(y, 0.f);
        dst[j] = (__bf16)y;
      }
    }
  } else {
    float* dst = (float*)p.out + idx;
    const float* res = RES ? (const float*)p.ep_res + idx : nullptr;
    if (vec) {
      if constexpr (RES) {
        const f32x4 r0 = *(const f32x4*)res, r1 = *(const f32x4*)(res + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += r0[j]; v[4 + j] += r1[j]; }
      }
      if constexpr (RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
      *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
      for (int j = 0; j < nv; ++j) {
        float y = v[j] + (RES ? res[j] : 0.f);
        if (RELU) y = fmaxf(y, 0.f);
        dst[j] = y;
      }
    }
  }
}

// per-channel constants of the tile in LDS: [bias_mean | bias_delta | scale | shift] x 64 (identity where absent);
// written by threads t = 0..63 of the caller, who also provides the barrier before they are read
template <int KIND>
__device__ __forceinline__ void ep_fill_constants(const ContractParams& p, const RngLive& rl, float* ba_lds, int t,
                                                  int ntile, int group, bool has_bias, bool has_aff) {
  if (t < BN) {
    const int col = ntile * BN + t;
    const int gcol = group * p.Ng + (col < p.Ng ? col : 0);
    float bm = 0.f, bdl = 0.f;
    if (has_bias && col < p.Ng) {
      const float eb = p.eps_b ? p.eps_b[gcol]
                               : btx_normal1((unsigned long long)gcol, rl.sample, p.layer, 1u, p.seed_lo, p.seed_hi);
      const float sb_ = btx_softplus_fast(p.rho_b[gcol]);
      if constexpr (KIND == 0) { bm = __builtin_fmaf(sb_, eb, p.mu_b[gcol]); }
      else { bm = p.mu_b[gcol]; bdl = sb_ * eb; }
    }
    ba_lds[t] = bm;
    ba_lds[BN + t] = bdl;
    ba_lds[2 * BN + t] = (has_aff && p.ep_scale) ? p.ep_scale[gcol] : 1.f;
    ba_lds[3 * BN + t] = (has_aff && p.ep_shift) ? p.ep_shift[gcol] : 0.f;
  }
}

template <int KIND, int NW>
__device__ __forceinline__ void staged_epilogue(const ContractParams& p, const RngLive& rl, const f32x16 (&accm)[2][2],
                                                const f32x16 (&accd)[2][2], unsigned char* smem, int tid, int wave,
                                                int lane, int ntile, int group, int split, uint32_t m0, int nvalid,
                                                uint32_t* ep_t = nullptr, int pwave = -1, bool first = true,
                                                float* ba_ext = nullptr) {
  // pwave: index of the 64-pixel group of the tile these fragments hold (default: the wave index); `wave` selects the
  // wave-private staging area.  first == false: the per-channel constants are already in LDS (second half of a wave
  // that owns 128 pixels).  ba_ext: the constants were written (and a barrier passed) by the caller, at this address.
  if (pwave < 0) pwave = wave;
  constexpr int EP_ROW = PT_EP_ROW;
  constexpr int EP_WAVE = PT_EP_WAVE;
  const int l31 = lane & 31, h = lane >> 5;
  const bool to_partial = p.ksplits > 1;
  const bool has_bias = (split == 0) && (p.mu_b != nullptr);
  const bool has_aff = !to_partial && ((p.ep_scale != nullptr) || (p.ep_shift != nullptr));
  const bool has_ba = has_bias || has_aff;
  float* ba_lds = ba_ext ? ba_ext : (float*)(smem + NW * EP_WAVE);
  if (has_ba && first && !ba_ext) {
    ep_fill_constants<KIND>(p, rl, ba_lds, tid, ntile, group, has_bias, has_aff);
    __syncthreads();
  }
  unsigned char* ep = smem + wave * EP_WAVE;

  // ---- stage 1: fragments -> f32 tile in LDS.  The common case — the whole 64-channel tile exists and the s_out words
  // are aligned with the fragment columns — runs without a branch per value: one hashed word per (32 pixels x 32
  // channels), two shifts and a mask per element.
  const bool fast = (ntile * BN + BN <= p.Ng) &&
                    (KIND == 0 || (!p.sign_out && (p.N & 31) == 0 && ((group * p.Ng) & 31) == 0));
  if (fast) {
    uint32_t wsh[2][2];
    if constexpr (KIND == 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int pl_ = min(pwave * 64 + mi * 32 + l31, max(nvalid - 1, 0));
        const uint32_t orow = (m0 + (uint32_t)pl_) * (uint32_t)p.N + (uint32_t)(group * p.Ng + ntile * BN);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          wsh[mi][ni] = btx_sign_word((orow + 32u * ni) >> 5, rl.kout_a, rl.kout_b) << (2 * h);
      }
    }
    auto body = [&](auto ba_tag) __attribute__((always_inline)) {
      constexpr bool BA = decltype(ba_tag)::value;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
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
                constexpr int dummy = 0;
                const int sft = 31 - (((rr & 1) ? 31 : 15) - 4 * q - (rr >> 1)) + dummy;
                val += u2f(f2u(dl) ^ ((wsh[mi][ni] << sft) & 0x80000000u));
              }
              if constexpr (BA) val = __builtin_fmaf(val, sc[rr], sh[rr]);
              v[rr] = val;
            }
            *(f32x4*)(ep + (mi * 32 + l31) * EP_ROW + cl * 4) = v;
          }
        }
    };
    if (has_ba) body(std::true_type{}); else body(std::false_type{});
  } else {
    // generic path: ragged channel tiles, unaligned s_out words, explicit sign arrays (parity mode)
#pragma unroll 1
    for (int mi = 0; mi < 2; ++mi) {
      const int pl_ = pwave * 64 + mi * 32 + l31;
      const bool pix_ok = pl_ < nvalid;
      const uint32_t orow = (m0 + (uint32_t)(pix_ok ? pl_ : 0)) * (uint32_t)p.N + (uint32_t)(group * p.Ng);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cl = ni * 32 + 8 * q + 4 * h;
          const int c0 = ntile * BN + cl;
          float v[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int col = c0 + rr;
            float val = (mi == 0 ? accm[0][ni][4 * q + rr] : accm[1][ni][4 * q + rr]);
            if (has_ba) val += ba_lds[cl + rr];
            if constexpr (KIND == 1) {
              float dl = (mi == 0 ? accd[0][ni][4 * q + rr] : accd[1][ni][4 * q + rr]);
              if (has_ba) dl += ba_lds[BN + cl + rr];
              uint32_t flip = 0;
              if (col < p.Ng && pix_ok) {
                if (p.sign_out) {
                  flip = (p.sign_out[orow + col] < 0) ? 0x80000000u : 0u;
                } else {
                  const uint32_t io = orow + col;
                  const uint32_t w1 = btx_sign_word(io >> 5, rl.kout_a, rl.kout_b);
                  flip = (w1 << (31 - btx_sign_bitpos(io & 31u))) & 0x80000000u;
                }
              }
              val += u2f(f2u(dl) ^ flip);
            }
            if (has_ba) val = __builtin_fmaf(val, ba_lds[2 * BN + cl + rr], ba_lds[3 * BN + cl + rr]);
            v[rr] = val;
          }
          *(f32x4*)(ep + (mi * 32 + l31) * EP_ROW + cl * 4) = (f32x4){v[0], v[1], v[2], v[3]};
        }
      }
    }
  }
#ifdef BTX_EP_TRACE
  if (ep_t) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ep_t[0] = (uint32_t)__builtin_amdgcn_s_memtime(); }
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area is private to the wave

  // ---- stage 2: LDS -> global, 8 lanes per pixel, whole 128-byte lines per instruction.  The output mode is uniform:
  // one specialised loop per mode.  Element offsets fit 32 bits (host: M*N < 2^31).
  {
    const int cg = lane & 7;
    const int col0 = ntile * BN + cg * 8;
    const int nv = min(8, p.Ng - col0);
    const bool vec_ok = ((p.N & 7) == 0) && (((group * p.Ng) & 7) == 0) && (nv == 8);
    const uint32_t cbase = (uint32_t)(group * p.Ng + col0);
    auto run = [&](auto outk_tag, auto res_tag, auto relu_tag) __attribute__((always_inline)) {
      constexpr int OUTK = decltype(outk_tag)::value;
      constexpr bool RES = decltype(res_tag)::value, RELU = decltype(relu_tag)::value;
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int pix = r8 * 8 + (lane >> 3);
        const int pl = pwave * 64 + pix;
        if (pl < nvalid && nv > 0) {
          const f32x4 lo = *(const f32x4*)(ep + pix * EP_ROW + cg * 32);
          const f32x4 hi = *(const f32x4*)(ep + pix * EP_ROW + cg * 32 + 16);
          const uint32_t idx = (m0 + (uint32_t)pl) * (uint32_t)p.N + cbase;
          ep_store8<OUTK, RES, RELU>(p, lo, hi, idx, split, nv, vec_ok);
        }
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    const bool res = !to_partial && p.ep_res != nullptr, relu = !to_partial && p.ep_relu;
    if (to_partial) run(std::integral_constant<int, 0>{}, F{}, F{});
    else if (p.out_bf16) {
      if (res) { if (relu) run(std::integral_constant<int, 1>{}, T{}, T{}); else run(std::integral_constant<int, 1>{}, T{}, F{}); }
      else { if (relu) run(std::integral_constant<int, 1>{}, F{}, T{}); else run(std::integral_constant<int, 1>{}, F{}, F{}); }
    } else {
      if (res) { if (relu) run(std::integral_constant<int, 2>{}, T{}, T{}); else run(std::integral_constant<int, 2>{}, T{}, F{}); }
      else { if (relu) run(std::integral_constant<int, 2>{}, F{}, T{}); else run(std::integral_constant<int, 2>{}, F{}, F{}); }
    }
  }
#ifdef BTX_EP_TRACE
  if (ep_t) { __builtin_amdgcn_sched_barrier(0); ep_t[1] = (uint32_t)__builtin_amdgcn_s_memtime(); }
#endif
}

}  // namespace btx
