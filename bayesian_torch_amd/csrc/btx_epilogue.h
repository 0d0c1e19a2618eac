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
#include "btx_contract.h"

namespace btx {

template <int KIND, int NW>
__device__ __forceinline__ void staged_epilogue(const ContractParams& p, const RngLive& rl, const f32x16 (&accm)[2][2],
                                                const f32x16 (&accd)[2][2], unsigned char* smem, int tid, int wave,
                                                int lane, int ntile, int group, int split, uint32_t m0, int nvalid) {
  // Stage 1: bias, Flipout combine (s_out), BN affine on the MFMA fragments; the f32 tile of the wave (64 pixels x 64
  // channels) goes to LDS (272-byte pixel rows: conflict-free both ways).  Stage 2: every lane takes 8 consecutive
  // channels of one pixel, adds the residual, applies ReLU, converts and stores 16 (bf16) / 32 (f32) contiguous bytes:
  // 8 lanes cover a pixel's 64 channels, so each store instruction writes whole 128-byte lines.  (Storing straight
  // from the fragments scatters 8-byte pieces over 64 different lines per instruction.)
  constexpr int EP_ROW = PT_EP_ROW;
  constexpr int EP_WAVE = PT_EP_WAVE;
  const int l31 = lane & 31, h = lane >> 5;
  const bool to_partial = p.ksplits > 1;
  const bool has_bias = (split == 0) && (p.mu_b != nullptr);
  float* bias_lds = (float*)(smem + NW * EP_WAVE);
  if (has_bias) {
    if (tid < BN) {
      const int col = ntile * BN + tid;
      float bm = 0.f, bdl = 0.f;
      if (col < p.Ng) {
        const int gcol = group * p.Ng + col;
        const float eb = p.eps_b ? p.eps_b[gcol]
                                 : btx_normal1((unsigned long long)gcol, rl.sample, p.layer, 1u, p.seed_lo, p.seed_hi);
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
    const int pl_ = wave * 64 + mi * 32 + l31;
    const bool pix_ok = pl_ < nvalid;
    const uint32_t orow = (m0 + (uint32_t)(pix_ok ? pl_ : 0)) * (uint32_t)p.N + (uint32_t)(group * p.Ng);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int colbase = ntile * BN + ni * 32;
      const uint32_t o0 = orow + colbase;
      const bool word_fast = (KIND == 1) && !p.sign_out && ((o0 & 31u) == 0) && (colbase + 32 <= p.Ng);
      uint32_t wout = 0;
      if (word_fast) wout = btx_sign_word(o0 >> 5, rl.kout_a, rl.kout_b);
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
            } else if (col < p.Ng && pix_ok) {
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
          if (has_aff) val = __builtin_fmaf(val, aff_lds[cl + rr], aff_lds[BN + cl + rr]);
          v[rr] = val;
        }
        *(f32x4*)(ep + (mi * 32 + l31) * EP_ROW + cl * 4) = (f32x4){v[0], v[1], v[2], v[3]};
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area is private to the wave
  {
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

}  // namespace btx
