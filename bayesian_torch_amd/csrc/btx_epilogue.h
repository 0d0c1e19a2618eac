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
#ifndef BTX_PT_ABL
#define BTX_PT_ABL 0
#endif
#ifndef BTX_EP_PRED_STORES
#define BTX_EP_PRED_STORES 0  // A/B: 1 = the bf16 stores of stage 2 under `if (pixel exists)` again (rounds 1-5)
#endif

namespace btx {

template <int I, int N, class F>
__device__ __forceinline__ void static_for_ep(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_ep<I + 1, N>(f);
  }
}

// per-channel constants of the tile in LDS: [bias_mean | bias_delta | scale | shift] x 64 (identity where absent);
// written by threads t = 0..63 of the caller, who also provides the barrier before they are read
struct EpRaw {  // what thread t of the filling wave read from memory for channel t of the tile
  float mu, rho, eps, sc, sh;
};
__device__ __forceinline__ EpRaw ep_load_constants(const ContractParams& p, int t, int ntile, int group, bool has_bias,
                                                   bool has_aff) {
  EpRaw r = {0.f, 0.f, 0.f, 1.f, 0.f};
  if (t < BN) {
    const int col = ntile * BN + t;
    const int gcol = group * p.Ng + (col < p.Ng ? col : 0);
    if (has_bias && col < p.Ng) {
      r.mu = p.mu_b[gcol];
      r.rho = p.rho_b[gcol];
      if (p.eps_b) r.eps = p.eps_b[gcol];
    }
    if (has_aff && p.ep_scale) r.sc = p.ep_scale[gcol];
    if (has_aff && p.ep_shift) r.sh = p.ep_shift[gcol];
  }
  return r;
}
template <int KIND>
__device__ __forceinline__ void ep_store_constants(const ContractParams& p, const RngLive& rl, const EpRaw& r, float* ba_lds,
                                                   int t, int ntile, int group, bool has_bias) {
  if (t < BN) {
    const int col = ntile * BN + t;
    const int gcol = group * p.Ng + (col < p.Ng ? col : 0);
    float bm = 0.f, bdl = 0.f;
    if (has_bias && col < p.Ng) {
      const float eb = p.eps_b ? r.eps : btx_normal1((unsigned long long)gcol, rl.sample, p.layer, 1u, p.seed_lo, p.seed_hi);
      const float sb_ = btx_softplus_fast(r.rho);
      if constexpr (KIND == 0) { bm = __builtin_fmaf(sb_, eb, r.mu); }
      else { bm = r.mu; bdl = sb_ * eb; }
    }
    ba_lds[t] = bm;
    ba_lds[BN + t] = bdl;
    ba_lds[2 * BN + t] = r.sc;
    ba_lds[3 * BN + t] = r.sh;
  }
}
template <int KIND>
__device__ __forceinline__ void ep_fill_constants(const ContractParams& p, const RngLive& rl, float* ba_lds, int t,
                                                  int ntile, int group, bool has_bias, bool has_aff) {
  const EpRaw r = ep_load_constants(p, t, ntile, group, has_bias, has_aff);
  ep_store_constants<KIND>(p, rl, r, ba_lds, t, ntile, group, has_bias);
}

// Tile pixel -> output pixel.  The default: the tile's pixels are consecutive output pixels from m0 on, the first `nvalid`
// exist.  The tall-strip tiles of btx_contract_taps.h bring their own map (rows of a column strip, dummy rows between
// images).
struct PixContig {
  uint32_t m0;
  int nvalid;
  __device__ __forceinline__ uint32_t operator()(int pl, bool& ok) const { ok = pl < nvalid; return m0 + (uint32_t)pl; }
  __device__ __forceinline__ uint32_t first() const { return m0; }
  // walk over tile pixels pl, pl + 8, pl + 16, ... (stage 2 of the store)
  struct Walk {
    uint32_t g; int pl, nvalid;
    __device__ __forceinline__ uint32_t get(bool& ok) const { ok = pl < nvalid; return g; }
    __device__ __forceinline__ void step8() { g += 8u; pl += 8; }
  };
  __device__ __forceinline__ Walk walk(int pl) const { return Walk{m0 + (uint32_t)pl, pl, nvalid}; }
};
// parity-major tiles (ContractParams.par_major): logical pixel L = m0 + pl -> the output pixel it names
__device__ __forceinline__ uint32_t par_major_pixel(const ContractParams& p, uint32_t L) {
  uint32_t cls, q, t, b, nb, a;
  fdivmod(L, p.fd_par_Mqp, (uint32_t)p.par_Mqp, cls, q);
  fdivmod(q, p.fd_par_Wh, (uint32_t)p.par_Wh, t, b);
  fdivmod(t, p.fd_par_Hh, (uint32_t)p.par_Hh, nb, a);
  return (nb * (uint32_t)p.Ho + 2u * a + (cls >> 1)) * (uint32_t)p.Wo + 2u * b + (cls & 1u);
}
struct PixParity {
  const ContractParams& p;
  uint32_t m0;
  int nvalid;
  __device__ __forceinline__ uint32_t operator()(int pl, bool& ok) const {
    ok = pl < nvalid;
    return ok ? par_major_pixel(p, m0 + (uint32_t)pl) : 0u;
  }
  __device__ __forceinline__ uint32_t first() const { return 0u; }
  struct Walk {
    const ContractParams& p;
    uint32_t L; int pl, nvalid;
    __device__ __forceinline__ uint32_t get(bool& ok) const { ok = pl < nvalid; return ok ? par_major_pixel(p, L) : 0u; }
    __device__ __forceinline__ void step8() { L += 8u; pl += 8; }
  };
  __device__ __forceinline__ Walk walk(int pl) const { return Walk{p, m0 + (uint32_t)pl, pl, nvalid}; }
};
struct PixTall {  // tile = pt_R virtual rows from row0 x pt_Wt columns from col0; virtual row k = image k / pt_P, row k % pt_P
  const ContractParams& p;
  int row0, col0;
  __device__ __forceinline__ uint32_t operator()(int pl, bool& ok) const {
    uint32_t r, c, img, oh;
    fdivmod((uint32_t)pl, p.fd_Wt, (uint32_t)p.pt_Wt, r, c);
    fdivmod((uint32_t)row0 + r, p.fd_P, (uint32_t)p.pt_P, img, oh);
    const int ow = col0 + (int)c;
    ok = (int)r < p.pt_R && (int)oh < p.Ho && (int)img < p.NB && ow < p.Wo;
    return ok ? ((img * (uint32_t)p.Ho + oh) * (uint32_t)p.Wo + (uint32_t)ow) : 0u;
  }
  __device__ __forceinline__ uint32_t first() const { return 0u; }
  // the same map, stepped 8 pixels at a time without divisions (pt_Wt >= 4: at most two row wraps per step)
  struct Walk {
    const ContractParams& p;
    int r, c, oh, img, col0;
    __device__ __forceinline__ uint32_t get(bool& ok) const {
      const int ow = col0 + c;
      ok = r < p.pt_R && oh < p.Ho && img < p.NB && ow < p.Wo;
      return ok ? (uint32_t)((img * p.Ho + oh) * p.Wo + ow) : 0u;
    }
    __device__ __forceinline__ void step8() {
      c += 8;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (c >= p.pt_Wt) { c -= p.pt_Wt; ++r; if (++oh >= p.pt_P) { oh -= p.pt_P; ++img; } }
    }
  };
  __device__ __forceinline__ Walk walk(int pl) const {
    uint32_t r, c, img, oh;
    fdivmod((uint32_t)pl, p.fd_Wt, (uint32_t)p.pt_Wt, r, c);
    fdivmod((uint32_t)row0 + r, p.fd_P, (uint32_t)p.pt_P, img, oh);
    return Walk{p, (int)r, (int)c, (int)oh, (int)img, col0};
  }
};

// RES_PRE (btx_contract_gemm8.h: one workgroup per CU, nobody covers this one's waits): a bf16 residual is requested in front
// of stage 1 — all 8 rows of the lane, 32 registers — and arrives while the fragments are staged, instead of two rounds of four
// loads each waited for in stage 2.
struct EpNoHook {
  __device__ __forceinline__ void operator()() const {}
};
// HOOK: called once behind the residual requests and in front of stage 1 (btx_contract_gemm8.h issues its L2 touches there:
// requests the wave does not wait for, queued BEHIND the residual rows it does wait for).
template <int KIND, int NW, class PM = PixContig, bool RES_PRE = false, class HOOK = EpNoHook>
__device__ __forceinline__ void staged_epilogue_pm(const ContractParams& p, const RngLive& rl, const f32x16 (&accm)[2][2],
                                                   const f32x16 (&accd)[2][2], unsigned char* smem, int tid, int wave,
                                                   int lane, int ntile, int group, int split, const PM& pm,
                                                   uint32_t* ep_t = nullptr, int pwave = -1, bool first = true,
                                                   float* ba_ext = nullptr, const HOOK& hook = HOOK{},
                                                   const uint32_t* wsh_ext = nullptr) {
  // wsh_ext: the four hashed s_out words of the lane's fragment blocks ([mi][ni], already shifted by 2h), computed by the
  // caller where it had idle issue slots (btx_contract_gemm8.h: while its first stages travel); only read on the fast path
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

  // BUFIO (RES_PRE on consecutive-pixel tiles, bf16 outputs): the lane's eight pixels are 8 * N elements apart, so the residual
  // loads and the stores of stage 2 are buffer instructions with ONE vector offset (the first pixel) and the pixel step in the
  // scalar offset — no per-pixel address arithmetic, no exec masks: pixels behind the tensor's end (the last tile) fall outside
  // the descriptor's range (loads return zeros, stores are dropped).  Byte offsets fit 32 bits (host: M * N < 2^31).
  constexpr bool BUFIO = RES_PRE && std::is_same<PM, PixContig>::value;
  u32x4 rpre[8];
  auto io_off = [&](int lane_) __attribute__((always_inline)) -> uint32_t {  // (evaluated where it is used: not kept across stage 1)
    return ((pm.first() + (uint32_t)(pwave * 64 + (lane_ >> 3))) * (uint32_t)p.N + (uint32_t)(group * p.Ng + ntile * BN + (lane_ & 7) * 8)) * 2u;
  };
  const uint32_t io_step = 8u * (uint32_t)p.N * 2u, io_bytes = (uint32_t)p.M * (uint32_t)p.N * 2u;
  if constexpr (RES_PRE) {
    const bool all_vec = ((p.N & 7) == 0) && (((group * p.Ng) & 7) == 0) && (ntile * BN + BN <= p.Ng);
    if (all_vec && !to_partial && p.ep_res != nullptr && p.out_bf16) {  // exactly the case fast<1, RES> of stage 2 handles
      if constexpr (BUFIO) {
        const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.ep_res, 0, io_bytes, 0x00020000);
        const uint32_t io_voff = io_off(lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) rpre[i] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, io_voff, io_step * (uint32_t)i, 0);
      } else {
        const uint32_t cbase = (uint32_t)(group * p.Ng + ntile * BN + (lane & 7) * 8);
        auto wk = pm.walk(pwave * 64 + (lane >> 3));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          bool ok;
          const uint32_t gp = wk.get(ok);
          wk.step8();
          rpre[i] = *(const u32x4*)((const __bf16*)p.ep_res + (ok ? gp * (uint32_t)p.N + cbase : cbase + pm.first() * (uint32_t)p.N));
        }
      }
    }
  }

  hook();

  // ---- stage 1: fragments -> f32 tile in LDS.  The common case — the whole 64-channel tile exists and the s_out words
  // are aligned with the fragment columns — runs without a branch per value: one hashed word per (32 pixels x 32
  // channels), two shifts and a mask per element.
  const bool fast = (ntile * BN + BN <= p.Ng) &&
                    (KIND == 0 || (!p.sign_out && (p.N & 31) == 0 && ((group * p.Ng) & 31) == 0));
  if (fast) {
    uint32_t wsh[2][2];
    // dl ^ (w & sign bit) as ONE v_bitop3 per value (the compiler's own selection is v_and + v_xor: 256 of the ~580
    // instructions of this stage); the constant lives in an SGPR — a 32-bit literal does not fit the VOP3 encoding
    uint32_t SB = 0x80000000u;
    asm volatile("" : "+s"(SB));
    if constexpr (KIND == 1) {
      if (wsh_ext) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wsh[i >> 1][i & 1] = wsh_ext[i];
      } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          bool pok;  // a pixel that does not exist hashes some word: its values are never stored
          const uint32_t orow = pm(pwave * 64 + mi * 32 + l31, pok) * (uint32_t)p.N + (uint32_t)(group * p.Ng + ntile * BN);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            wsh[mi][ni] = btx_sign_word((orow + 32u * ni) >> 5, rl.kout_a, rl.kout_b) << (2 * h);
        }
      }
    }
    // three forms of the body: nothing per channel | BN affine only (every convolution of a converted ResNet: no bias) |
    // bias and affine (identity scale / shift where absent) — the bias adds are a quarter of the body's VALU work
    auto body = [&](auto bias_tag, auto aff_tag) __attribute__((always_inline)) {
      constexpr bool BIAS = decltype(bias_tag)::value, BA = decltype(aff_tag)::value;
      // eight groups of (32-channel half ni, 8-channel run q); the constants of group g+1 are requested in front of group g's
      // arithmetic (two register sets): read where they are used, every group starts with an LDS round trip — eight of them,
      // 1.2k+ of the ~3.4k cycles a wave spends in this stage when nobody covers it (btx_contract_gemm8.h)
      struct Cst { f32x4 bm, bd, sc, sh; };
      auto ldc = [&](int g, Cst& c) __attribute__((always_inline)) {
        const int cl = (g >> 2) * 32 + 8 * (g & 3) + 4 * h;
        if constexpr (BIAS) {
          c.bm = *(const f32x4*)(ba_lds + cl);
          c.bd = *(const f32x4*)(ba_lds + BN + cl);
        }
        if constexpr (BA) {
          c.sc = *(const f32x4*)(ba_lds + 2 * BN + cl);
          c.sh = *(const f32x4*)(ba_lds + 3 * BN + cl);
        }
      };
      Cst c0, c1;
      ldc(0, c0);
      static_for_ep<0, 8>([&](auto g_tag) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value, ni = g >> 2, q = g & 3;
        Cst& cur = (g & 1) ? c1 : c0;
        Cst& nxt = (g & 1) ? c0 : c1;
        if constexpr (g < 7) ldc(g + 1, nxt);
        const int cl = ni * 32 + 8 * q + 4 * h;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          f32x4 v;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float val = accm[mi][ni][4 * q + rr];
            if constexpr (BIAS) val += cur.bm[rr];
            if constexpr (KIND == 1) {
              float dl = accd[mi][ni][4 * q + rr];
              if constexpr (BIAS) dl += cur.bd[rr];
              // element e = 8q + 4h + rr of the word sits at bit ((e&1) ? 31 : 15) - (e>>1); wsh is pre-shifted by 2h
              const int sft = 31 - (((rr & 1) ? 31 : 15) - 4 * q - (rr >> 1));
              val += u2f(__builtin_amdgcn_bitop3_b32(f2u(dl), wsh[mi][ni] << sft, SB, 0x78));  // dl ^ (w & SB)
            }
            if constexpr (BA) val = __builtin_fmaf(val, cur.sc[rr], cur.sh[rr]);
            v[rr] = val;
          }
          *(f32x4*)(ep + (mi * 32 + l31) * EP_ROW + cl * 4) = v;
        }
      });
    };
#ifdef BTX_EP_TRACE2
    if (ep_t) { __builtin_amdgcn_sched_barrier(0); ep_t[2] = (uint32_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
    if (has_bias) body(std::true_type{}, std::true_type{});
    else if (has_ba) body(std::false_type{}, std::true_type{});
    else body(std::false_type{}, std::false_type{});
#ifdef BTX_EP_TRACE2
    if (ep_t) { __builtin_amdgcn_sched_barrier(0); ep_t[3] = (uint32_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
  } else {
    // generic path: ragged channel tiles, unaligned s_out words, explicit sign arrays (parity mode)
#pragma unroll 1
    for (int mi = 0; mi < 2; ++mi) {
      bool pix_ok;
      const uint32_t gp_ = pm(pwave * 64 + mi * 32 + l31, pix_ok);
      const uint32_t orow = (pix_ok ? gp_ : pm.first()) * (uint32_t)p.N + (uint32_t)(group * p.Ng);
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
  __builtin_amdgcn_sched_barrier(0);  // nothing of stage 2 (index arithmetic) above this line: the accumulators are dead only now

  // ---- stage 2: LDS -> global, 8 lanes per pixel, whole 128-byte lines per instruction.  Element offsets fit 32 bits
  // (host: M*N < 2^31).
  {
    const int cg = lane & 7;
    const int col0 = ntile * BN + cg * 8;
    const int nv = min(8, p.Ng - col0);
    const uint32_t cbase = (uint32_t)(group * p.Ng + col0);
    // the whole 64-channel tile exists and 8-channel runs are 16-byte aligned (wave-uniform): no per-lane tails
    const bool all_vec = ((p.N & 7) == 0) && (((group * p.Ng) & 7) == 0) && (ntile * BN + BN <= p.Ng);
    const bool res = !to_partial && p.ep_res != nullptr, relu = !to_partial && p.ep_relu;
    // Straight-line form, one per output mode (uniform): the 8 residual loads of the lane go out first (a pixel that does
    // not exist reads the tile's first pixel instead: no branch in front of a load), then the 16 LDS reads, then the
    // arithmetic and the 8 predicated stores.  (A loop that handles one pixel per iteration waits for each residual load
    // before the next LDS read: eight dependent L2 round trips per wave — measured 4.8k -> cycles of the store side.)
    auto fast = [&](auto outk_tag, auto res_tag, auto relu_tag) __attribute__((always_inline)) {
      constexpr int OUTK = decltype(outk_tag)::value;
      constexpr bool RES = decltype(res_tag)::value, RELU = decltype(relu_tag)::value;
      auto wk = pm.walk(pwave * 64 + (lane >> 3));
      int lane_s2 = lane;
      asm volatile("" : "+v"(lane_s2));  // (opaque: the store offset is computed here, not kept from the residual requests)
      const uint32_t io_voff = BUFIO ? io_off(lane_s2) : 0u;
      // two rounds of four pixels: half the registers, still four loads in flight.  RES_PRE (the accumulators are dead and the
      // caller has the registers: one workgroup per CU): one round of eight — all 16 LDS reads in front of the arithmetic
      constexpr int NP = RES_PRE ? 8 : 4;
#pragma unroll
      for (int hf = 0; hf < 8 / NP; ++hf) {
        uint32_t idx[NP];
        bool pk[NP];
        u32x4 r0[NP], r1[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          if constexpr (BUFIO && OUTK == 1) {  // addressed by (io_voff, scalar step): no per-pixel index
            pk[i] = true; idx[i] = 0;
            if constexpr (RES) r0[i] = rpre[hf * NP + i];
            continue;
          }
          const uint32_t gp = wk.get(pk[i]);
          wk.step8();
          idx[i] = gp * (uint32_t)p.N + cbase;
          if constexpr (RES) {
            const uint32_t li = pk[i] ? idx[i] : cbase + pm.first() * (uint32_t)p.N;
            if constexpr (OUTK == 1 && RES_PRE) r0[i] = rpre[hf * NP + i];
            else if constexpr (OUTK == 1) r0[i] = *(const u32x4*)((const __bf16*)p.ep_res + li);
            else { r0[i] = *(const u32x4*)((const float*)p.ep_res + li); r1[i] = *(const u32x4*)((const float*)p.ep_res + li + 4); }
          }
        }
        f32x4 lo[NP], hi[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int pix = (hf * NP + i) * 8 + (lane >> 3);
          lo[i] = *(const f32x4*)(ep + pix * EP_ROW + cg * 32);
          hi[i] = *(const f32x4*)(ep + pix * EP_ROW + cg * 32 + 16);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          float v[8] = {lo[i][0], lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1], hi[i][2], hi[i][3]};
          if constexpr (RES) {
            if constexpr (OUTK == 1) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { v[2 * j] += u2f(r0[i][j] << 16); v[2 * j + 1] += u2f(r0[i][j] & 0xffff0000u); }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) { v[j] += u2f(r0[i][j]); v[4 + j] += u2f(r1[i][j]); }
            }
          }
          if constexpr (RELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          if constexpr (OUTK == 1 && !BUFIO && !(BTX_PT_ABL & 64) && !BTX_EP_PRED_STORES) {
            // bf16 outputs: the store is a buffer store whose offset is out of the descriptor's range for a pixel that does not
            // exist (dropped by the hardware) — no exec-mask block per store.  Under `if (pk[i])` each of the lane's eight stores
            // sat in a basic block of its own together with its ReLU / rounding arithmetic (s_and_saveexec .. s_or exec around
            // every global_store in the ISA): nothing of one pixel could be scheduled beside another's.  (Round 6, E15; byte
            // offsets fit 32 bits: the host admits M * N < 2^31 elements.)
            const f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
            const u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(x0, bf16x4));
            const u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(x1, bf16x4));
            const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, io_bytes, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){p0[0], p0[1], p1[0], p1[1]}, out_rsrc, pk[i] ? idx[i] * 2u : 0xfffffff0u, 0, 0);
          } else
          if (pk[i]) {
            if constexpr (OUTK == 0) {
              float* dst = p.partial + (size_t)split * p.M * p.N + idx[i];
              *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
              *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            } else if constexpr (OUTK == 1) {
              const f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
              const u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(x0, bf16x4));
              const u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(x1, bf16x4));
              if constexpr (BUFIO) {
                const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, io_bytes, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){p0[0], p0[1], p1[0], p1[1]}, out_rsrc, io_voff,
                                                       io_step * (uint32_t)(hf * NP + i), 0);
              } else if constexpr (BTX_PT_ABL & 64) {  // measurement builds: everything but the store itself
                asm volatile("" ::"v"(p0), "v"(p1), "v"(idx[i]));
              } else {
                *(u32x4*)((__bf16*)p.out + idx[i]) = (u32x4){p0[0], p0[1], p1[0], p1[1]};
              }
            } else {
              float* dst = (float*)p.out + idx[i];
              *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
              *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            }
          }
        }
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (all_vec) {
      if (to_partial) fast(std::integral_constant<int, 0>{}, F{}, F{});
      else if (p.out_bf16) {
        if (res) { if (relu) fast(std::integral_constant<int, 1>{}, T{}, T{}); else fast(std::integral_constant<int, 1>{}, T{}, F{}); }
        else { if (relu) fast(std::integral_constant<int, 1>{}, F{}, T{}); else fast(std::integral_constant<int, 1>{}, F{}, F{}); }
      } else {
        if (res) { if (relu) fast(std::integral_constant<int, 2>{}, T{}, T{}); else fast(std::integral_constant<int, 2>{}, T{}, F{}); }
        else { if (relu) fast(std::integral_constant<int, 2>{}, F{}, T{}); else fast(std::integral_constant<int, 2>{}, F{}, F{}); }
      }
    } else {
      // ragged channel tiles / unaligned runs: one loop with run-time modes, element by element (rare shapes)
      auto wk = pm.walk(pwave * 64 + (lane >> 3));
#pragma unroll 1
      for (int r8 = 0; r8 < 8; ++r8) {
        const int pix = r8 * 8 + (lane >> 3);
        bool pok;
        const uint32_t gp = wk.get(pok);
        wk.step8();
        if (pok && nv > 0) {
          const uint32_t idx = gp * (uint32_t)p.N + cbase;
          const float* row = (const float*)(ep + pix * EP_ROW + cg * 32);
#pragma unroll 1
          for (int j = 0; j < nv; ++j) {
            float y = row[j];
            if (to_partial) { p.partial[(size_t)split * p.M * p.N + idx + j] = y; continue; }
            if (res) y += p.out_bf16 ? (float)((const __bf16*)p.ep_res)[idx + j] : ((const float*)p.ep_res)[idx + j];
            if (relu) y = fmaxf(y, 0.f);
            if (p.out_bf16) ((__bf16*)p.out)[idx + j] = (__bf16)y;
            else ((float*)p.out)[idx + j] = y;
          }
        }
      }
    }
  }
#ifdef BTX_EP_TRACE
  if (ep_t) { __builtin_amdgcn_sched_barrier(0); ep_t[1] = (uint32_t)__builtin_amdgcn_s_memtime(); }
#endif
}

// ---- the store side straight from the fragment registers (bf16 outputs, whole aligned 64-channel tiles) -----------------
// No staging through LDS: bias + Flipout combine fold the delta accumulators into the mean ones in place, the affine runs on
// the lane's 4-channel runs, v_permlane32_swap pairs the two half-waves' runs into one 8-channel run per lane, then residual,
// ReLU, rounding and ONE 16-byte buffer store per (pixel, 16 channels) — the 2 lanes of a pixel write 32 contiguous bytes,
// four store instructions complete its 128-byte line in L2.  A lane addresses its two pixels directly (gp = pm(pixel)), so
// tiles whose pixels are not consecutive output pixels (the tall strips of btx_contract_taps.h) cost two divisions per lane
// instead of a (row, column) walk per stored pixel.  First written for the persistent kernel (btx_contract_taps3.h: ~2k
// cycles against the 4.8k of the staged store of a 56x56 tile, 8.1k for a tall strip; profiles/r03_phase_timers.txt).
// Contract: as staged_epilogue_pm, plus: ksplits == 1, outputs bf16, N % 32 == 0, (group * Ng) % 32 == 0, the tile's 64
// channels exist, hashed s_out, M * N * 2 < 0x7ff00000 (32-bit offsets with an out-of-range value to spare): the host sets
// ContractParams.ep_direct only then.
// gp[mi] / gok[mi]: output pixel index of the lane's pixel (wave * 64 + mi * 32 + lane % 32) of the tile, and whether it exists
// (the caller evaluates its pixel map — PixContig or PixTall — so that this body exists ONCE per kernel: with one copy per
// map behind a run-time branch the compiler hoists the copies' common sign-word shifts above the branch and spills them).
// OUT: the output element type (bf16: one 16-byte store per (pixel, 16 channels); float: two).  FILLED: the caller has
// written the tile's constants to ba_lds (ep_fill_constants) and passed a barrier since — else this call does both.
template <int KIND, typename OUT = __bf16, bool FILLED = false>
__device__ __forceinline__ void direct_epilogue(const ContractParams& p, const RngLive& rl, const f32x16 (&accm)[2][2],
                                                const f32x16 (&accd)[2][2], float* ba_lds, int tid, int lane,
                                                int ntile, int group, const uint32_t (&gp)[2], const bool (&gok)[2]) {
  constexpr bool BF = sizeof(OUT) == 2;
  constexpr uint32_t ESZ = (uint32_t)sizeof(OUT);
  const int h = lane >> 5;
  const bool has_bias = p.mu_b != nullptr;
  const bool has_aff = (p.ep_scale != nullptr) || (p.ep_shift != nullptr);
  // ba_lds: [bias mean | bias delta | scale | shift] x 64, identities where absent
  if constexpr (!FILLED) ep_fill_constants<KIND>(p, rl, ba_lds, tid, ntile, group, has_bias, has_aff);
  const uint32_t cbase = (uint32_t)(group * p.Ng + ntile * BN);
  const uint32_t out_bytes = (uint32_t)p.M * (uint32_t)p.N * ESZ;
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, out_bytes, 0x00020000);
  uint32_t eo[2];  // byte offset of the lane's 8-channel run of (pixel mi, half 0, pair 0); a pixel that does not exist
#pragma unroll     // gets an out-of-range offset: its loads return zeros, its stores are dropped
  for (int mi = 0; mi < 2; ++mi) eo[mi] = gok[mi] ? (gp[mi] * (uint32_t)p.N + cbase + 8u * (uint32_t)h) * ESZ : 0x80000000u;
  const bool res = p.ep_res != nullptr;
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.ep_res, 0, res ? out_bytes : 0u, 0x00020000);
  // residual rows of pixel mi: (32-channel half ni) x (16-channel pair k) -> 8 channels of this lane (16 | 32 bytes)
  struct Res { u32x4 a[2][2], b[2][2]; };
  auto load_res = [&](Res& r, int mi) __attribute__((always_inline)) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int k = 0; k < 2; ++k) {  // (no residual: a zero-length descriptor, the loads return zeros without a memory access)
        const uint32_t o = eo[mi] + (uint32_t)(ni * 32 + 16 * k) * ESZ;
        r.a[ni][k] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, o, 0, 0);
        if constexpr (!BF) r.b[ni][k] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, o + 16u, 0, 0);
      }
  };
  uint32_t wsh[2][2];
  if constexpr (KIND == 1) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const uint32_t orow = gp[mi] * (uint32_t)p.N + cbase;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) wsh[mi][ni] = btx_sign_word((orow + 32u * ni) >> 5, rl.kout_a, rl.kout_b) << (2 * h);
    }
  }
  uint32_t SB = 0x80000000u;
  asm volatile("" : "+s"(SB));
  const float lowb = p.ep_relu ? 0.f : -__builtin_inff();  // ReLU as a lower bound: one instruction stream for both
  if constexpr (!FILLED) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the constants are in LDS
  // Phase A — (mean + bias) + s_out * (delta + bias delta) -> o (scalars, not the MFMA tuples: written back in place, the
  // partially updated 16-register tuples made the allocator copy and spill).  ONE instantiation: without a bias the
  // constants are zeros (16 LDS reads, 128 additions).
  float o[2][2][16];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cl = ni * 32 + 8 * q + 4 * h;
      const f32x4 bm = *(const f32x4*)(ba_lds + cl);
      const f32x4 bd = *(const f32x4*)(ba_lds + BN + cl);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float val = accm[mi][ni][4 * q + rr] + bm[rr];
          if constexpr (KIND == 1) {
            const float dl = accd[mi][ni][4 * q + rr] + bd[rr];
            const int sft = 31 - (((rr & 1) ? 31 : 15) - 4 * q - (rr >> 1));
            val += u2f(__builtin_amdgcn_bitop3_b32(f2u(dl), wsh[mi][ni] << sft, SB, 0x78));  // dl ^ (w & SB)
          }
          o[mi][ni][4 * q + rr] = val;
        }
#ifndef BTX_DIRECT_NO_FOLD_FENCE
      __builtin_amdgcn_sched_barrier(0);  // one (ni, q) group at a time: left alone the scheduler requests the constants of
#endif                                    // all eight groups up front (64 registers) and the fold's results go to scratch
    }
  __builtin_amdgcn_sched_barrier(0);
  // Phase B — one (pixel mi, 32-channel half ni, 16-channel pair k) group at a time: affine, pairing, residual, ReLU, store
  Res rv, rnx;
  load_res(rv, 0);  // (behind the fold: in front of it these registers push the fold's results into scratch)
  struct Cst { f32x4 sc[2], sh[2]; };
  auto load_cst = [&](Cst& c, int ni, int k) __attribute__((always_inline)) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int cl = ni * 32 + 8 * (2 * k + hf) + 4 * h;
      c.sc[hf] = *(const f32x4*)(ba_lds + 2 * BN + cl);
      c.sh[hf] = *(const f32x4*)(ba_lds + 3 * BN + cl);
    }
  };
  Cst cst[2];
  load_cst(cst[0], 0, 0);
  static_for_ep<0, 8>([&](auto g_tag) __attribute__((always_inline)) {
    constexpr int g = decltype(g_tag)::value;
    constexpr int mi = g >> 2, ni = (g >> 1) & 1, k = g & 1;
    if constexpr (g + 1 < 8) load_cst(cst[(g + 1) & 1], ((g + 1) >> 1) & 1, (g + 1) & 1);
    if constexpr (g == 0) load_res(rnx, 1);  // pixel 1's residual rows land while pixel 0 is stored
    const Cst& c = cst[g & 1];
    float t[2][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) t[hf][rr] = __builtin_fmaf(o[mi][ni][4 * (2 * k + hf) + rr], c.sc[hf][rr], c.sh[hf][rr]);
    float v[8];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const auto r = __builtin_amdgcn_permlane32_swap(f2u(t[0][rr]), f2u(t[1][rr]), false, false);
      v[rr] = u2f(r[0]);
      v[4 + rr] = u2f(r[1]);
    }
    const Res& rr_ = (mi == 0) ? rv : rnx;
    if constexpr (BF) {
      const u32x4 rw = rr_.a[ni][k];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] += u2f(rw[j] << 16); v[2 * j + 1] += u2f(rw[j] & 0xffff0000u); }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += u2f(rr_.a[ni][k][j]); v[4 + j] += u2f(rr_.b[ni][k][j]); }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], lowb);
    const uint32_t so = eo[mi] + (uint32_t)(ni * 32 + 16 * k) * ESZ;
    if constexpr (BF) {
      const f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
      const u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(x0, bf16x4));
      const u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(x1, bf16x4));
      __builtin_amdgcn_raw_buffer_store_b128((u32x4){p0[0], p0[1], p1[0], p1[1]}, out_rsrc, so, 0, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b128((u32x4){f2u(v[0]), f2u(v[1]), f2u(v[2]), f2u(v[3])}, out_rsrc, so, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128((u32x4){f2u(v[4]), f2u(v[5]), f2u(v[6]), f2u(v[7])}, out_rsrc, so + 16u, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// the contiguous-tile form every other kernel uses
template <int KIND, int NW, bool RES_PRE = false>
__device__ __forceinline__ void staged_epilogue(const ContractParams& p, const RngLive& rl, const f32x16 (&accm)[2][2],
                                                const f32x16 (&accd)[2][2], unsigned char* smem, int tid, int wave,
                                                int lane, int ntile, int group, int split, uint32_t m0, int nvalid,
                                                uint32_t* ep_t = nullptr, int pwave = -1, bool first = true,
                                                float* ba_ext = nullptr) {
  const PixContig pm = {m0, nvalid};
  staged_epilogue_pm<KIND, NW, PixContig, RES_PRE>(p, rl, accm, accd, smem, tid, wave, lane, ntile, group, split, pm, ep_t, pwave,
                                                   first, ba_ext);
}

}  // namespace btx
