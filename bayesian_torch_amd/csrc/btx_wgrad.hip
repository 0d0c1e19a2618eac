// btx_wgrad.hip — weight gradient of a variational contraction (gfx950): the one contraction of the training path that is
// not Flipout-forward-shaped (the data gradient is: bayesian_torch_amd/autograd.py).
//
//   dW_mu   [n][tap][c] = sum over output pixels p of   dy[p][n]            * x[p @ tap][c]
//   dW_delta[n][tap][c] = sum over output pixels p of  (dy[p][n] s_out[p][n]) * (x[p @ tap][c] s_in[p @ tap][c])   (Flipout)
//
// (reference: what torch autograd derives for F.conv*d / F.linear inside conv_flipout.py:376-417, conv_variational.py:379-380;
// dmu = dW_mu, drho = dW_delta * eps * sigmoid(rho) follow elementwise on the host side.)
//
// A GEMM whose reduction axis is the PIXEL axis: both operands are stored channels-last, i.e. with the reduction index
// slowest.  Two forms of the product:
//   * general (f32 activations, odd channel counts, sign tensors): the exact-f32 MFMA v_mfma_f32_32x32x2_f32 takes ONE
//     element per lane per operand, so its fragments are plain 4-byte LDS reads of a pixel-major f32 tile;
//   * FAST (bf16 activations in whole 16-channel runs, hashed signs — a ResNet body and its row-fused stem): 16-byte
//     loads, the next step's loads in flight during the MFMAs, tiles written to LDS transposed and still bf16 so that a
//     fragment (8 consecutive pixels of a channel) is one 16-byte read for v_mfma_f32_32x32x16_bf16.  bf16 x bf16 products
//     are exact in f32 and the accumulation is f32 in both forms: the same gradient.
//
// Workgroup = 4 waves <-> (64 output channels, 64 input channels of ONE tap, a chunk of output pixels).  Per 64-pixel
// step the tile dy[64 px][64 n] and the tap-shifted tile x[64 px][64 c] (zeros outside the input) are staged as f32,
// for Flipout also their sign-flipped copies; wave w multiplies pixels {16w .. 16w+15}: 8 MFMA k-steps x 4 (8) output
// tiles.  The four partial sums are added through LDS and leave with f32 atomics (dW must be zero on entry: the entry
// point clears it).  Bias gradients (column sums of dy, of dy*s_out) ride along in the workgroups of tap 0 / channel
// block 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/btx.h"
#include "btx_contract.h"
#include "btx_rng.h"

using namespace btx;

namespace {

struct WgradParams {
  const void* x;
  const void* dy;
  float* dwm;
  float* dwd;
  float* dbm;
  float* dbd;
  const int8_t* sign_in;
  const int8_t* sign_out;
  int NB, D, H, W, C, Cg, Do, Ho, Wo, N, Ng, KD, KH, KW;
  int sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int M, K, T, groups, ntiles, ctiles, chunks, chunk_px;
  uint32_t kin_a, kin_b, kout_a, kout_b;
  const uint32_t* sample_ptr;  // BtxRng.sample_idx_dev: the sign keys are then derived on the device (captured training steps)
  uint32_t seed_lo, seed_hi, layer;
  int swap;
  int tune;     // measurement builds (BTX_WGRAD_T3_ABL): 1 = no slab stores, 2 = no MFMA section (results wrong: time only)
  int pair, Tw; // pair: a tap has 32 channels (row-fused stems) — a workgroup takes TWO taps, one per 32-column half of its x tile, and
                // shares the dy tile between them (Tw = workgroups along the tap axis: ceil(T / 2), else T)
  int direct;   // slab mode with ONE chunk: its sums are the result — plain stores straight into dW, no reduction launch
  float* slab;  // btx_contract_wgrad_ws: [chunk][mean | delta][N*K] partial sums, plain stores (nullptr: f32 atomics into dW)
  FastDiv fd_Wo, fd_Ho, fd_Do, fd_T, fd_ctiles, fd_ntiles, fd_groups;
};

#include "btx_wgrad_taps.h"

constexpr int WG_PX = 64;          // pixels per step
constexpr int WG_ROW = 64 * 4 + 16;  // bytes per pixel row of a staged tile: 64 f32 + pad (conflict-free 16-B writes)
constexpr int WG_TILE = WG_PX * WG_ROW;

__device__ __forceinline__ float sgn_flip(float v, bool neg) { return neg ? -v : v; }

// FAST: the bf16 staging / bf16-MFMA path below (chosen by the host: wgrad_fast_ok); otherwise exact-f32 MFMA on f32 tiles
template <typename ACT, int KIND, bool FAST>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // tiles: dy, x (+ signed copies for Flipout)
  unsigned char* t_dy = smem;
  unsigned char* t_x = smem + WG_TILE;
  unsigned char* t_dys = smem + 2 * WG_TILE;
  unsigned char* t_xs = smem + 3 * WG_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hk = lane >> 5;
  // sign keys of this (sample, layer): the host's, or — with a device-resident sample index — derived here, as the forward does
  uint32_t kin_a = p.kin_a, kin_b = p.kin_b, kout_a = p.kout_a, kout_b = p.kout_b;
  if (KIND == 1 && p.sample_ptr) {
    const uint32_t smp = __builtin_amdgcn_readfirstlane(*p.sample_ptr);
    const uint32_t si = p.swap ? BTX_STREAM_SIGN_OUT : BTX_STREAM_SIGN_IN, so = p.swap ? BTX_STREAM_SIGN_IN : BTX_STREAM_SIGN_OUT;
    const BtxPhilox4 ki = btx_philox4x32_10(0u, smp, p.layer, si, p.seed_lo, p.seed_hi);
    const BtxPhilox4 ko = btx_philox4x32_10(0u, smp, p.layer, so, p.seed_lo, p.seed_hi);
    kin_a = __builtin_amdgcn_readfirstlane(ki.x[0]); kin_b = __builtin_amdgcn_readfirstlane(ki.x[1]);
    kout_a = __builtin_amdgcn_readfirstlane(ko.x[0]); kout_b = __builtin_amdgcn_readfirstlane(ko.x[1]);
  }

  uint32_t u, u_chunk, u_tap, u_ct, u_nt, u_g;
  fdivmod(blockIdx.x, p.fd_T, (uint32_t)p.Tw, u, u_tap);
  fdivmod(u, p.fd_ctiles, (uint32_t)p.ctiles, u, u_ct);
  fdivmod(u, p.fd_ntiles, (uint32_t)p.ntiles, u, u_nt);
  fdivmod(u, p.fd_groups, (uint32_t)p.groups, u_chunk, u_g);
  const bool pair = FAST && p.pair != 0;
  const int tap = pair ? 2 * (int)u_tap : (int)u_tap, ct = (int)u_ct, nt = (int)u_nt, grp = (int)u_g, chunk = (int)u_chunk;
  const int m_begin = chunk * p.chunk_px, m_end = min(p.M, m_begin + p.chunk_px);
  const bool do_bias = (tap == 0) && (ct == 0) && (p.dbm != nullptr);
  const int ni_live = (p.Ng - nt * 64 > 32) ? 2 : 1, nj_live = pair ? 2 : ((p.Cg - ct * 64 > 32) ? 2 : 1);

  // staging role: thread t loads 16 consecutive channels of pixel (t >> 2) of each tile: quarter q = t & 3
  const int s_px = tid >> 2, s_q = tid & 3;
  // the tap and the channel run this thread stages of x (pair: quarters 0-1 = tap, quarters 2-3 = tap + 1, 32 channels each)
  const int tapx = pair ? tap + (s_q >> 1) : tap;
  const int xc0 = pair ? 16 * (s_q & 1) : ct * 64 + 16 * s_q;
  const bool tapx_ok = tapx < p.T;
  const int kw = tapx % p.KW, kh = (tapx / p.KW) % p.KH, kd = tapx / (p.KW * p.KH);
  f32x16 acc_m[2][2], acc_d[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_m[a][b][r] = 0.f; acc_d[a][b][r] = 0.f; }
  float bsum_m = 0.f, bsum_d = 0.f;  // threads 0..63: column n = tid of the dy tile

  // bf16 fast path of the staging (the shapes of a ResNet body): whole 16-channel runs, 32-byte aligned, hashed signs — two
  // 16-byte loads per operand, the Flipout signs as XOR masks on the packed pairs (one hashed word covers the run), and the
  // loads of step m0 + 64 are requested before the MFMAs of step m0 so that their latency runs beside the matrix pipe.
  constexpr bool fast = FAST;
  static_assert(!FAST || sizeof(ACT) == 2, "the fast path stages bf16");
  const bool x_al16 = (p.C % 16) == 0;
  struct Raw {
    u32x4 x[2], y[2];
    uint32_t xi, yi;  // element offsets of the runs (sign words)
  };
  auto fetch = [&](int m0, Raw& r) __attribute__((always_inline)) {
    const int m = m0 + s_px;
    const bool pix_ok = m < m_end;
    uint32_t t1, uow, uoh, uod, unb;
    fdivmod((uint32_t)(pix_ok ? m : 0), p.fd_Wo, (uint32_t)p.Wo, t1, uow);
    fdivmod(t1, p.fd_Ho, (uint32_t)p.Ho, t1, uoh);
    fdivmod(t1, p.fd_Do, (uint32_t)p.Do, unb, uod);
    const int id = (int)uod * p.sd - p.pd + kd * p.dd, ih = (int)uoh * p.sh - p.ph + kh * p.dh,
              iw = (int)uow * p.sw - p.pw + kw * p.dw;
    const bool in_ok = pix_ok && tapx_ok && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W &&
                       (xc0 < p.Cg);
    const bool out_ok = pix_ok && (nt * 64 + 16 * s_q < p.Ng);
    const long long xo = ((((long long)unb * p.D + id) * p.H + ih) * p.W + iw) * p.C + grp * p.Cg + xc0;
    const long long yo = (long long)m * p.N + grp * p.Ng + nt * 64 + 16 * s_q;
    const u32x4 z = {0u, 0u, 0u, 0u};
    const u32x4* yp = (const u32x4*)((const uint16_t*)p.dy + (out_ok ? yo : 0));
    const u32x4 y0 = yp[0], y1 = yp[1];
    u32x4 x0, x1;
    if (x_al16) {
      const u32x4* xp = (const u32x4*)((const uint16_t*)p.x + (in_ok ? xo : 0));
      x0 = xp[0]; x1 = xp[1];
    } else {
      const u32x2* xp = (const u32x2*)((const uint16_t*)p.x + (in_ok ? xo : 0));
      const u32x2 a0 = xp[0], a1 = xp[1], a2 = xp[2], a3 = xp[3];
      x0 = (u32x4){a0[0], a0[1], a1[0], a1[1]}; x1 = (u32x4){a2[0], a2[1], a3[0], a3[1]};
    }
    r.x[0] = in_ok ? x0 : z; r.x[1] = in_ok ? x1 : z;
    r.y[0] = out_ok ? y0 : z; r.y[1] = out_ok ? y1 : z;
    r.xi = (uint32_t)xo; r.yi = (uint32_t)yo;
  };
  // Fast path (round 5): the tiles stay bf16 and PIXEL-MAJOR in LDS — four blocks of 16 channels, each [64 pixels][16 channels] with
  // 32-byte rows — written with 16-byte stores (two per operand per thread; round 2-4 wrote the transpose with sixteen 2-byte
  // stores per operand: the kernel was bound by that staging).  The MFMA fragments — 8 consecutive pixels of one channel — come out
  // of gfx950's transpose read instead: ds_read_b64_tr_b16 hands lane c of a 16-lane group column c of the 4 x 16 block the group's
  // lanes address (lane i: row i/4, columns 4(i%4)..; probed on hardware: tools/ubench/tr_probe.hip), i.e. 4 pixels of channel c.
  // Block bases are skewed by {0, 16, 64, 80} bytes so that the eight lanes of a 16-byte store group hit eight distinct slots.
  // bf16 x bf16 products are exact in f32 and the accumulation stays f32: the same gradient as before.
  constexpr int TB_BLK = 64 * 32 + 128;  // bytes per 16-channel block (+ room for the skew)
  auto blk_base = [](int q) __attribute__((always_inline)) { return q * TB_BLK + ((q & 1) + 4 * (q >> 1)) * 16; };
  auto stash = [&](const Raw& r) __attribute__((always_inline)) {
    uint32_t wx = 0, wy = 0;
    if constexpr (KIND == 1) {
      // one hashed word covers an aligned 16-element run; pair j of the word has its signs at bits 15 - j and 31 - j
      wy = btx_sign_word(r.yi >> 5, kout_a, kout_b) << ((r.yi & 31u) >> 1);
      if (x_al16) {
        wx = btx_sign_word(r.xi >> 5, kin_a, kin_b) << ((r.xi & 31u) >> 1);
      } else {  // the run may straddle two words: the 32 signs starting at element xi (btx_contract_stem.h)
        const uint32_t w = btx_sign_word(r.xi >> 5, kin_a, kin_b), w1 = btx_sign_word((r.xi >> 5) + 1u, kin_a, kin_b);
        const uint32_t k = (r.xi & 31u) >> 1;
        const uint32_t lo = ((w & 0xffffu) << 16) | (w1 & 0xffffu), hi = (w & 0xffff0000u) | (w1 >> 16);
        wx = ((lo << k) >> 16) | ((hi << k) & 0xffff0000u);
      }
    }
    const int off = blk_base(s_q) + s_px * 32;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      *(u32x4*)(t_x + off + 16 * hf) = r.x[hf];
      *(u32x4*)(t_dy + off + 16 * hf) = r.y[hf];
      if constexpr (KIND == 1) {
        u32x4 xs_, ys_;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int j = 4 * hf + d;  // pair index of the dword inside the run
          xs_[d] = r.x[hf][d] ^ ((wx << j) & 0x80008000u);
          ys_[d] = r.y[hf][d] ^ ((wy << j) & 0x80008000u);
        }
        *(u32x4*)(t_xs + off + 16 * hf) = xs_;
        *(u32x4*)(t_dys + off + 16 * hf) = ys_;
      }
    }
  };
  // two register sets, fetched TWO steps ahead: a step of this kernel is short (8 MFMAs per wave; 4 on a row-fused stem) and one
  // step of run-ahead left the L2 / HBM round trip exposed (profiles/r05_experiments.txt E13)
  Raw raw_a, raw_b;
  if constexpr (fast) {
    if (m_begin < m_end) fetch(m_begin, raw_a);
    if (m_begin + WG_PX < m_end) fetch(m_begin + WG_PX, raw_b);
  }

  auto body = [&](int m0, Raw& raw) __attribute__((always_inline)) {
    if constexpr (fast) {
      stash(raw);
    } else {
    // ---- stage: dy[m0 + s_px][nt*64 + 16 s_q ..] and x[(m0 + s_px) @ tap][ct*64 + 16 s_q ..]
    {
      const int m = m0 + s_px;
      const bool pix_ok = m < m_end;
      uint32_t t1, uow, uoh, uod, unb;
      fdivmod((uint32_t)(pix_ok ? m : 0), p.fd_Wo, (uint32_t)p.Wo, t1, uow);
      fdivmod(t1, p.fd_Ho, (uint32_t)p.Ho, t1, uoh);
      fdivmod(t1, p.fd_Do, (uint32_t)p.Do, unb, uod);
      const int id = (int)uod * p.sd - p.pd + kd * p.dd, ih = (int)uoh * p.sh - p.ph + kh * p.dh,
                iw = (int)uow * p.sw - p.pw + kw * p.dw;
      const bool in_ok = pix_ok && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const long long xo = ((((long long)unb * p.D + id) * p.H + ih) * p.W + iw) * p.C + grp * p.Cg + ct * 64 + 16 * s_q;
      const long long yo = (long long)m * p.N + grp * p.Ng + nt * 64 + 16 * s_q;
      float xv[16], yv[16], xsv[16], ysv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const bool cok = in_ok && (ct * 64 + 16 * s_q + e < p.Cg);
        const bool nok = pix_ok && (nt * 64 + 16 * s_q + e < p.Ng);
        xv[e] = cok ? (float)((const ACT*)p.x)[xo + e] : 0.f;
        yv[e] = nok ? (float)((const ACT*)p.dy)[yo + e] : 0.f;
      }
      if constexpr (KIND == 1) {
        uint32_t cwi = 0xffffffffu, cw = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned long long i = (unsigned long long)(xo + e);
          bool neg = false;
          if (in_ok && (ct * 64 + 16 * s_q + e < p.Cg)) {
            if (p.sign_in) neg = p.sign_in[i] < 0;
            else {
              const uint32_t wi = (uint32_t)(i >> 5);
              if (wi != cwi) { cwi = wi; cw = btx_sign_word(wi, kin_a, kin_b); }
              neg = (cw >> btx_sign_bitpos((uint32_t)i & 31u)) & 1u;
            }
          }
          xsv[e] = sgn_flip(xv[e], neg);
        }
        cwi = 0xffffffffu;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned long long i = (unsigned long long)(yo + e);
          bool neg = false;
          if (pix_ok && (nt * 64 + 16 * s_q + e < p.Ng)) {
            if (p.sign_out) neg = p.sign_out[i] < 0;
            else {
              const uint32_t wi = (uint32_t)(i >> 5);
              if (wi != cwi) { cwi = wi; cw = btx_sign_word(wi, kout_a, kout_b); }
              neg = (cw >> btx_sign_bitpos((uint32_t)i & 31u)) & 1u;
            }
          }
          ysv[e] = sgn_flip(yv[e], neg);
        }
      }
      unsigned char* rx = t_x + s_px * WG_ROW + s_q * 64;
      unsigned char* ry = t_dy + s_px * WG_ROW + s_q * 64;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        *(f32x4*)(rx + 16 * v) = (f32x4){xv[4 * v], xv[4 * v + 1], xv[4 * v + 2], xv[4 * v + 3]};
        *(f32x4*)(ry + 16 * v) = (f32x4){yv[4 * v], yv[4 * v + 1], yv[4 * v + 2], yv[4 * v + 3]};
      }
      if constexpr (KIND == 1) {
        unsigned char* rxs = t_xs + s_px * WG_ROW + s_q * 64;
        unsigned char* rys = t_dys + s_px * WG_ROW + s_q * 64;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          *(f32x4*)(rxs + 16 * v) = (f32x4){xsv[4 * v], xsv[4 * v + 1], xsv[4 * v + 2], xsv[4 * v + 3]};
          *(f32x4*)(rys + 16 * v) = (f32x4){ysv[4 * v], ysv[4 * v + 1], ysv[4 * v + 2], ysv[4 * v + 3]};
        }
      }
    }
    }
    __syncthreads();
    if constexpr (fast) { if (m0 + 2 * WG_PX < m_end) fetch(m0 + 2 * WG_PX, raw); }  // in flight during this step's and the next step's MFMAs
    if constexpr (fast) {
      // ---- multiply (bf16 MFMA): wave w takes pixels 16w..16w+15 of the step = one k-step of 16.  Fragment of lane (l31, hk) for
      // channel half i: channel 32 i + l31, pixels 16w + 8hk + 0..7 = two transpose reads of 4 pixels each; within the lane's
      // 16-lane group (channel block 2i + (l31 >> 4)) lane c addresses row (c >> 2), columns 4 (c & 3).. of the 4-pixel x 16-channel block
      bf16x8 a[2], b[2], as_[2], bs_[2];
      {
        const int c16 = lane & 15, g16 = (lane >> 4) & 1;
        const int lane_off = (16 * wave + 8 * hk + (c16 >> 2)) * 32 + (c16 & 3) * 8;
        auto lds_addr = [&](const unsigned char* tile, int blk) __attribute__((always_inline)) -> uint32_t {
          return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(tile + blk_base(blk) + lane_off);
        };
        // all transpose reads of the step in flight together, one wait (inline asm: the compiler does not track these reads)
        const uint32_t ad0 = lds_addr(t_dy, g16), ad1 = lds_addr(t_dy, 2 + g16), bd0 = lds_addr(t_x, g16), bd1 = lds_addr(t_x, 2 + g16);
        u32x2 a0l, a0h, a1l, a1h, b0l, b0h, b1l, b1h;
        if constexpr (KIND == 1) {
          const uint32_t as0 = lds_addr(t_dys, g16), as1 = lds_addr(t_dys, 2 + g16), bs0 = lds_addr(t_xs, g16), bs1 = lds_addr(t_xs, 2 + g16);
          u32x2 c0l, c0h, c1l, c1h, d0l, d0h, d1l, d1h;
          asm volatile(
              "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:128\n\t"
              "ds_read_b64_tr_b16 %2, %17\n\tds_read_b64_tr_b16 %3, %17 offset:128\n\t"
              "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %18 offset:128\n\t"
              "ds_read_b64_tr_b16 %6, %19\n\tds_read_b64_tr_b16 %7, %19 offset:128\n\t"
              "ds_read_b64_tr_b16 %8, %20\n\tds_read_b64_tr_b16 %9, %20 offset:128\n\t"
              "ds_read_b64_tr_b16 %10, %21\n\tds_read_b64_tr_b16 %11, %21 offset:128\n\t"
              "ds_read_b64_tr_b16 %12, %22\n\tds_read_b64_tr_b16 %13, %22 offset:128\n\t"
              "ds_read_b64_tr_b16 %14, %23\n\tds_read_b64_tr_b16 %15, %23 offset:128\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(a0l), "=&v"(a0h), "=&v"(a1l), "=&v"(a1h), "=&v"(b0l), "=&v"(b0h), "=&v"(b1l), "=&v"(b1h), "=&v"(c0l), "=&v"(c0h),
                "=&v"(c1l), "=&v"(c1h), "=&v"(d0l), "=&v"(d0h), "=&v"(d1l), "=&v"(d1h)
              : "v"(ad0), "v"(ad1), "v"(bd0), "v"(bd1), "v"(as0), "v"(as1), "v"(bs0), "v"(bs1)
              : "memory");
          as_[0] = __builtin_bit_cast(bf16x8, (u32x4){c0l[0], c0l[1], c0h[0], c0h[1]});
          as_[1] = __builtin_bit_cast(bf16x8, (u32x4){c1l[0], c1l[1], c1h[0], c1h[1]});
          bs_[0] = __builtin_bit_cast(bf16x8, (u32x4){d0l[0], d0l[1], d0h[0], d0h[1]});
          bs_[1] = __builtin_bit_cast(bf16x8, (u32x4){d1l[0], d1l[1], d1h[0], d1h[1]});
        } else {
          asm volatile(
              "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:128\n\t"
              "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:128\n\t"
              "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:128\n\t"
              "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:128\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(a0l), "=&v"(a0h), "=&v"(a1l), "=&v"(a1h), "=&v"(b0l), "=&v"(b0h), "=&v"(b1l), "=&v"(b1h)
              : "v"(ad0), "v"(ad1), "v"(bd0), "v"(bd1)
              : "memory");
        }
        a[0] = __builtin_bit_cast(bf16x8, (u32x4){a0l[0], a0l[1], a0h[0], a0h[1]});
        a[1] = __builtin_bit_cast(bf16x8, (u32x4){a1l[0], a1l[1], a1h[0], a1h[1]});
        b[0] = __builtin_bit_cast(bf16x8, (u32x4){b0l[0], b0l[1], b0h[0], b0h[1]});
        b[1] = __builtin_bit_cast(bf16x8, (u32x4){b1l[0], b1l[1], b1h[0], b1h[1]});
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (i < ni_live && j < nj_live) {
            acc_m[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc_m[i][j], 0, 0, 0);
            if constexpr (KIND == 1) acc_d[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_[i], bs_[j], acc_d[i][j], 0, 0, 0);
          }
        }
      if (do_bias && tid < 64) {  // column n = tid of dy: one 2-byte element per pixel row of its 16-channel block
        const unsigned char* cb = t_dy + blk_base(tid >> 4) + (tid & 15) * 2;
        const unsigned char* cbs = t_dys + blk_base(tid >> 4) + (tid & 15) * 2;
        for (int q = 0; q < WG_PX; ++q) {
          bsum_m += u2f((uint32_t)(*(const uint16_t*)(cb + q * 32)) << 16);
          if constexpr (KIND == 1) bsum_d += u2f((uint32_t)(*(const uint16_t*)(cbs + q * 32)) << 16);
        }
      }
    } else {
    // ---- multiply: wave w takes pixels 16w..16w+15 of the step, two per MFMA
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int px = 16 * wave + 2 * kk + hk;
      float a[2], b[2], as_[2], bs_[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *(const float*)(t_dy + px * WG_ROW + (32 * i + l31) * 4);
        b[i] = *(const float*)(t_x + px * WG_ROW + (32 * i + l31) * 4);
        if constexpr (KIND == 1) {
          as_[i] = *(const float*)(t_dys + px * WG_ROW + (32 * i + l31) * 4);
          bs_[i] = *(const float*)(t_xs + px * WG_ROW + (32 * i + l31) * 4);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (i < ni_live && j < nj_live) {  // uniform: halves of the tile beyond Ng / Cg (row-fused stems: 32 of 64 columns)
            acc_m[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc_m[i][j], 0, 0, 0);
            if constexpr (KIND == 1) acc_d[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(as_[i], bs_[j], acc_d[i][j], 0, 0, 0);
          }
        }
    }
    if (do_bias && tid < 64) {
      for (int q = 0; q < WG_PX; ++q) {
        bsum_m += *(const float*)(t_dy + q * WG_ROW + tid * 4);
        if constexpr (KIND == 1) bsum_d += *(const float*)(t_dys + q * WG_ROW + tid * 4);
      }
    }
    }
    __syncthreads();
  };
  for (int m0 = m_begin; m0 < m_end; m0 += 2 * WG_PX) {
    body(m0, raw_a);
    if (m0 + WG_PX < m_end) body(m0 + WG_PX, raw_b);
  }

  // ---- reduce the four waves' partial tiles through LDS and add them to dW.
  // C/D layout of 32x32 MFMAs: reg r of lane (l31, hk) = D[row = (r&3) + 8*(r>>2) + 4*hk][col = l31]; row = n, col = c.
  float* red = (float*)smem;  // [wave][64 n][64 c]
  const size_t slab_e = (size_t)p.N * p.T * p.Cg;  // elements of one dW tensor (N = groups * Ng)
  auto reduce_store = [&](const f32x16 (&acc)[2][2], float* dst, int which) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hk, c = 32 * j + l31;
          red[(wave * 64 + n) * 64 + c] = acc[i][j][r];
        }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int n = e >> 6, c = e & 63;
      const float v = red[e] + red[4096 + e] + red[8192 + e] + red[12288 + e];
      const int ng = nt * 64 + n;
      const int tapc = pair ? tap + (c >> 5) : tap, cg = pair ? (c & 31) : ct * 64 + c;  // pair: column half -> tap
      if (ng < p.Ng && cg < p.Cg && tapc < p.T) {
        const size_t e = ((size_t)(grp * p.Ng + ng) * p.T + tapc) * p.Cg + cg;
        if (p.direct) dst[e] = v;
        else if (p.slab) p.slab[((size_t)chunk * (KIND == 1 ? 2 : 1) + which) * slab_e + e] = v;  // this chunk's slab: a plain store
        else atomicAdd(dst + e, v);
      }
    }
  };
  reduce_store(acc_m, p.dwm, 0);
  if constexpr (KIND == 1) reduce_store(acc_d, p.dwd, 1);
  if (do_bias && tid < 64 && nt * 64 + tid < p.Ng) {
    atomicAdd(p.dbm + grp * p.Ng + nt * 64 + tid, bsum_m);
    if constexpr (KIND == 1) atomicAdd(p.dbd + grp * p.Ng + nt * 64 + tid, bsum_d);
  }
}

void sign_keys_host(const BtxRng* rng, uint32_t stream, uint32_t* ka, uint32_t* kb) {
  const BtxPhilox4 k = btx_philox4x32_10(0u, rng->sample_idx, rng->layer_id, stream, (uint32_t)rng->seed,
                                         (uint32_t)(rng->seed >> 32));
  *ka = k.x[0];
  *kb = k.x[1];
}

// ---- the operands of the DATA gradient in one pass (btx_dgrad_weights) ---------------------------------------------------
// dx of  y = conv(x, W)  at stride 1 is a convolution of dy with the spatially flipped, channel-transposed kernel; of a Linear
// layer a product with W^T.  Its three weight operands — mu, rho and the eps the FORWARD drew — in the GEMM-major order of that
// transposed geometry:  out[c][tp][n] = src[n][flip ? T-1-tp : tp][c],  eps regenerated at the SOURCE index (BTX-RNG v1 is a pure
// function of it).  32 x 32 (n, c) tiles of one tap through LDS: reads run along c (one Philox call per four consecutive source
// elements), writes along n.  Replaces, per layer and training step, fill_eps + unpack + three flips + four pack copies (ten ATen /
// libbtx launches of 3-6 us, ~150 per ResNet18 step).
__global__ __launch_bounds__(256) void dgrad_weights_kernel(const float* __restrict__ mu, const float* __restrict__ rho,
                                                            float* __restrict__ omu, float* __restrict__ orho,
                                                            float* __restrict__ oeps, int N, int T, int C, int flip, int ctiles,
                                                            int ntiles, uint32_t k0, uint32_t k1, uint32_t sample, uint32_t layer,
                                                            const uint32_t* __restrict__ sample_ptr) {
  __shared__ float tm[32][33], tr[32][33], te[32][33];
  if (sample_ptr) sample = __builtin_amdgcn_readfirstlane(*sample_ptr);
  const int b = blockIdx.x;
  const int ct = b % ctiles, nt = (b / ctiles) % ntiles, t = b / (ctiles * ntiles);
  const int tp = flip ? T - 1 - t : t;
  const int tid = threadIdx.x;
  {  // read: thread = (row r of 32 n, quad q of 8) -> 4 consecutive c
    const int r = tid >> 3, q = tid & 7;
    const int n = nt * 32 + r, c0 = ct * 32 + 4 * q;
    float m4[4] = {0.f, 0.f, 0.f, 0.f}, r4[4] = {0.f, 0.f, 0.f, 0.f}, e4[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N && c0 < C) {
      const size_t e0 = ((size_t)n * T + t) * C + c0;
      if ((C & 3) == 0) {  // e0 is a multiple of 4: one Philox block, 16-byte loads
        const f32x4 a = *(const f32x4*)(mu + e0), bq = *(const f32x4*)(rho + e0);
        m4[0] = a[0]; m4[1] = a[1]; m4[2] = a[2]; m4[3] = a[3];
        r4[0] = bq[0]; r4[1] = bq[1]; r4[2] = bq[2]; r4[3] = bq[3];
        btx_normal4((uint32_t)(e0 >> 2), sample, layer, 0u, k0, k1, e4);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c0 + e < C) {
            m4[e] = mu[e0 + e]; r4[e] = rho[e0 + e];
            float z[4];
            btx_normal4((uint32_t)((e0 + e) >> 2), sample, layer, 0u, k0, k1, z);
            e4[e] = z[(e0 + e) & 3];
          }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { tm[r][4 * q + e] = m4[e]; tr[r][4 * q + e] = r4[e]; te[r][4 * q + e] = e4[e]; }
  }
  __syncthreads();
  {  // write: thread = (row cc of 32 c, quad q of 8) -> 4 consecutive n
    const int cc = tid >> 3, q = tid & 7;
    const int c = ct * 32 + cc, n0 = nt * 32 + 4 * q;
    if (c < C && n0 < N) {
      const size_t o0 = ((size_t)c * T + tp) * N + n0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + e < N) { omu[o0 + e] = tm[4 * q + e][cc]; orho[o0 + e] = tr[4 * q + e][cc]; oeps[o0 + e] = te[4 * q + e][cc]; }
    }
  }
}

}  // namespace

extern "C" int btx_dgrad_weights(const float* mu_w, const float* rho_w, float* out_mu, float* out_rho, float* out_eps, int N, int T,
                                 int C, int flip, const BtxRng* rng, void* stream) {
  if (!mu_w || !rho_w || !out_mu || !out_rho || !out_eps || !rng) return BTX_E_NULL;
  if (N <= 0 || T <= 0 || C <= 0) return BTX_E_SHAPE;
  if ((unsigned long long)N * T * C > 0xfffffffcULL) return BTX_E_UNSUPPORTED;  // BTX-RNG v1 block index is 32 bits
  if (((((uintptr_t)mu_w) | ((uintptr_t)rho_w)) & 15) && (C & 3) == 0) return BTX_E_ALIGN;
  const int ctiles = (C + 31) / 32, ntiles = (N + 31) / 32;
  const long long nwg = (long long)ctiles * ntiles * T;
  if (nwg > 0x7fffffffLL) return BTX_E_UNSUPPORTED;
  hipLaunchKernelGGL(dgrad_weights_kernel, dim3((int)nwg), dim3(256), 0, (hipStream_t)stream, mu_w, rho_w, out_mu, out_rho, out_eps, N,
                     T, C, flip ? 1 : 0, ctiles, ntiles, (uint32_t)rng->seed, (uint32_t)(rng->seed >> 32), rng->sample_idx,
                     rng->layer_id, rng->sample_idx_dev);
  return (int)hipGetLastError();
}

namespace {

// out[e] = the chunks' partial sums added in chunk order (fixed: the result does not depend on the launch's timing).  256 threads =
// QL element groups x CL chunk lanes; lane cl adds chunks cl, cl + CL, ..., the CL sums are added in order through LDS.
// With `rho`: the tensor that feeds the rho gradient (dW_delta of a Flipout layer, dW_mu of a Reparameterization layer) also leaves
// as drho = dW * eps * sigmoid(rho), eps regenerated (what btx_rho_grad computes from the finished dW; drho may alias that dW).
struct RhoFuse {
  const float* rho;
  float* drho;
  const uint32_t* sample_ptr;
  uint32_t k0, k1, sample, layer;
  int which;  // the slab half drho is formed from
};

__device__ __forceinline__ float rho_factor(float z, float rho) { return z * (1.0f / (1.0f + expf(-rho))); }

template <int VEC>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ slab, float* __restrict__ out_m,
                                                           float* __restrict__ out_d, size_t E, int chunks, int nk, int cl_log2,
                                                           const RhoFuse rf) {
  __shared__ float red[256 * VEC];
  const int CL = 1 << cl_log2, QL = 256 >> cl_log2;
  const int tid = threadIdx.x, ql = tid & (QL - 1), cl = tid >> (8 - cl_log2);
  const int which = blockIdx.y;
  const size_t e0 = ((size_t)blockIdx.x * QL + ql) * VEC;
  const size_t cstride = (size_t)nk * E;
  float s[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = 0.f;
  if (e0 < E) {
    const float* src = slab + (size_t)which * E + e0;
#pragma unroll 8
    for (int c = cl; c < chunks; c += CL) {
      if constexpr (VEC == 4) {
        const f32x4 t = *(const f32x4*)(src + (size_t)c * cstride);
        s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
      } else {
        s[0] += src[(size_t)c * cstride];
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) red[tid * VEC + v] = s[v];
  __syncthreads();
  if (cl == 0 && e0 < E) {
    float* out = which ? out_d : out_m;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float t = red[ql * VEC + v];
      for (int k = 1; k < CL; ++k) t += red[(k * QL + ql) * VEC + v];
      s[v] = t;
    }
    const bool fuse = rf.rho != nullptr && which == rf.which;
    if (!fuse || rf.drho != out) {
      if constexpr (VEC == 4) *(f32x4*)(out + e0) = (f32x4){s[0], s[1], s[2], s[3]};
      else out[e0] = s[0];
    }
    if (fuse) {
      const uint32_t smp = rf.sample_ptr ? *rf.sample_ptr : rf.sample;
      float z[4];
      btx_normal4((uint32_t)(e0 >> 2), smp, rf.layer, BTX_STREAM_EPS_W, rf.k0, rf.k1, z);
      if constexpr (VEC == 4) {
        const f32x4 r = *(const f32x4*)(rf.rho + e0);
        *(f32x4*)(rf.drho + e0) = (f32x4){s[0] * rho_factor(z[0], r[0]), s[1] * rho_factor(z[1], r[1]), s[2] * rho_factor(z[2], r[2]),
                                          s[3] * rho_factor(z[3], r[3])};
      } else {
        const uint32_t l = (uint32_t)e0 & 3u;
        const float zz = l == 0 ? z[0] : (l == 1 ? z[1] : (l == 2 ? z[2] : z[3]));
        rf.drho[e0] = s[0] * rho_factor(zz, rf.rho[e0]);
      }
    }
  }
}

// drho = dW * eps * sigmoid(rho) on a finished dW (one-chunk launches store straight into dW: no reduction launch to fuse into)
__global__ __launch_bounds__(256) void wgrad_rho_kernel(const float* dw, size_t n, const RhoFuse rf) {
  const uint32_t smp = rf.sample_ptr ? __builtin_amdgcn_readfirstlane(*rf.sample_ptr) : rf.sample;
  const size_t nblk = (n + 3) >> 2;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < nblk; b += (size_t)gridDim.x * 256) {
    float z[4];
    btx_normal4((uint32_t)b, smp, rf.layer, BTX_STREAM_EPS_W, rf.k0, rf.k1, z);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t i = (b << 2) + e;
      if (i < n) rf.drho[i] = dw[i] * rho_factor(z[e], rf.rho[i]);
    }
  }
}

static inline const char* wg_tune_env(const char* name) {
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// the pixel chunking of the two kernels (shared by btx_wgrad_workspace_bytes and the launch).  Atomics path (rounds 2-5): about
// `target_wgs` workgroups.  Slab paths: at most `target_wgs` = the workgroups the chip holds at once (one round: a second, partly
// filled round costs a whole round's prologue, epilogue and slab; profiles/r05_experiments.txt E13), a single chunk when one
// chunk's tiles already exceed that.
void wgrad_chunks(long long base, long long M, long long target_wgs, bool round_down, long long* chunks_out, long long* cpx_out) {
  long long chunks = round_down ? target_wgs / base : (target_wgs + base - 1) / base;
  const long long max_chunks = (M + WG_PX - 1) / WG_PX;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  const long long cpx = ((M + chunks - 1) / chunks + WG_PX - 1) / WG_PX * WG_PX;
  *chunks_out = (M + cpx - 1) / cpx;
  *cpx_out = cpx;
}
long long wgrad_target_wgs(bool slab) {
  const char* e = wg_tune_env(slab ? "BTX_WGRAD_SLAB_WGS" : "BTX_WGRAD_WGS");
  return e ? atoll(e) : (slab ? 512 : 2048);  // slabs: two 4-wave workgroups per CU
}
bool wgrad_pair_geom() { return wg_tune_env("BTX_WGRAD_NO_PAIR") == nullptr; }
long long wgrad_taps3_target_wgs() {
  const char* e = wg_tune_env("BTX_WGRAD_T3_WGS");
  return e ? atoll(e) : 256;  // one 12-wave workgroup per CU
}

int wgrad_fill_params(int kind, const BtxGeom* g, const void* x, const void* dy, const BtxNoise* noise, uint32_t flags, WgradParams& p) {
  if (flags & BTX_FLAG_TRANSPOSED) return BTX_E_UNSUPPORTED;  // transposed layers: swap x and dy (host)
  // BTX_FLAG_ROWFUSE (small-C stems on the row-fused geometry of the forward, include/btx.h): a kernel row = KW*C contiguous
  // elements of x plays the part of the channel axis — KH "taps" of KW*C "channels" instead of KH*KW taps of C channels (a
  // 7x7x3 stem: 7 x 32 of a 64-wide tile instead of 49 x 3 of 64) — and the signs are the forward's hashed ones
  const bool rowfuse = (flags & BTX_FLAG_ROWFUSE) != 0;
  if (rowfuse && (g->groups != 1 || g->D != 1 || g->KD != 1 || g->pw != 0 || g->dw != 1)) return BTX_E_UNSUPPORTED;
  int32_t Do, Ho, Wo;
  int rc = btx_out_shape(g, 0, &Do, &Ho, &Wo);
  if (rc) return rc;
  memset(&p, 0, sizeof(p));
  p.x = x; p.dy = dy;
  p.sign_in = noise ? noise->sign_in : nullptr;
  p.sign_out = noise ? noise->sign_out : nullptr;
  p.NB = g->NB; p.D = g->D; p.H = g->H; p.W = g->W; p.C = g->C; p.Cg = g->C / g->groups;
  p.Do = Do; p.Ho = Ho; p.Wo = Wo; p.N = g->N; p.Ng = g->N / g->groups;
  p.KD = g->KD; p.KH = g->KH; p.KW = g->KW;
  p.sd = g->sd; p.sh = g->sh; p.sw = g->sw; p.pd = g->pd; p.ph = g->ph; p.pw = g->pw; p.dd = g->dd; p.dh = g->dh; p.dw = g->dw;
  const long long M = (long long)g->NB * Do * Ho * Wo;
  if (M > 0x7fffffffLL) return BTX_E_UNSUPPORTED;
  p.M = (int)M; p.T = g->KD * g->KH * g->KW; p.K = p.T * p.Cg; p.groups = g->groups;
  if (rowfuse) { p.Cg = g->KW * g->C; p.KW = 1; p.T = g->KD * g->KH; }  // K = T * Cg unchanged; the pixel stride p.C stays C
  p.ntiles = (p.Ng + 63) / 64; p.ctiles = (p.Cg + 63) / 64;
  return 0;
}

// geometry test of the all-taps kernel that needs no pointers (workspace sizing)
bool wgrad_taps3_geom(const WgradParams& p, int act_dtype) {
  return act_dtype == BTX_ACT_BF16 && p.groups == 1 && p.D == 1 && p.KD == 1 && p.KH == 3 && p.KW == 3 && p.sh == 1 && p.sw == 1 &&
         p.ph == 1 && p.pw == 1 && p.dh == 1 && p.dw == 1 && p.Ho == p.H && p.Wo == p.W && p.W >= 2 && p.W <= 63 && p.H >= 2 &&
         (p.C % 64) == 0 && (p.N % 64) == 0 && !wg_tune_env("BTX_WGRAD_NO_TAPS3");
}

int wgrad_impl(int kind, const BtxGeom* g, const void* x, const void* dy, float* dw_mu, float* dw_delta, float* db_mu, float* db_delta,
               const BtxRng* rng, const BtxNoise* noise, int act_dtype, uint32_t flags, void* ws, size_t ws_bytes, const float* rho_w,
               float* drho, void* stream) {
  if (!g || !x || !dy || !dw_mu || !rng) return BTX_E_NULL;
  if ((rho_w != nullptr) != (drho != nullptr)) return BTX_E_NULL;
  if (rho_w && kind == BTX_KIND_REPARAM && drho == dw_mu) return BTX_E_UNSUPPORTED;  // dmu IS dw_mu: drho needs its own buffer
  if (kind != BTX_KIND_REPARAM && kind != BTX_KIND_FLIPOUT) return BTX_E_UNSUPPORTED;
  if (kind == BTX_KIND_FLIPOUT && !dw_delta) return BTX_E_NULL;
  if ((db_mu != nullptr) && kind == BTX_KIND_FLIPOUT && !db_delta) return BTX_E_NULL;
  if (act_dtype != BTX_ACT_F32 && act_dtype != BTX_ACT_BF16) return BTX_E_DTYPE;
  WgradParams p;
  int rc = wgrad_fill_params(kind, g, x, dy, noise, flags, p);
  if (rc) return rc;
  p.dwm = dw_mu; p.dwd = dw_delta; p.dbm = db_mu; p.dbd = db_delta;
  const long long M = p.M;
  const int nk = kind == BTX_KIND_FLIPOUT ? 2 : 1;
  const size_t E = (size_t)g->N * p.K;
  const bool slab = ws != nullptr && ws_bytes > 0;
  if (slab && (((uintptr_t)ws) & 15)) return BTX_E_ALIGN;
  p.slab = slab ? (float*)ws : nullptr;
  const bool taps3 = slab && wgrad_taps3_geom(p, act_dtype) && wgrad_taps3_ok(p, act_dtype, db_mu != nullptr);
  // bf16 fast path (the shapes of a ResNet body and its row-fused stem): whole 16-channel runs, x runs at multiples of 4
  // elements (8-byte loads) or 16 (16-byte loads), hashed signs, 16-byte aligned tensors
  const bool fast_ok = act_dtype == BTX_ACT_BF16 && (p.C % 4 == 0) && (p.Cg % 16 == 0) && (p.N % 16 == 0) && (p.Ng % 16 == 0) &&
                       !p.sign_in && !p.sign_out && ((((uintptr_t)x) | ((uintptr_t)dy)) % 16 == 0);
  // a tap of 32 channels (row-fused 7x7x3 stems: 8 columns x 4 channels) fills half of the 64-column x tile: two taps per workgroup,
  // one dy tile for both (the kernel is bound by its loads per MFMA; profiles/r05_experiments.txt E16)
  p.pair = (!taps3 && fast_ok && p.Cg == 32 && p.T >= 2 && wgrad_pair_geom()) ? 1 : 0;
  p.Tw = p.pair ? (p.T + 1) / 2 : p.T;
  const long long base = taps3 ? (long long)p.ntiles * p.ctiles : (long long)p.groups * p.ntiles * p.ctiles * p.Tw;
  long long chunks, cpx;
  wgrad_chunks(base, M, taps3 ? wgrad_taps3_target_wgs() : wgrad_target_wgs(slab), slab, &chunks, &cpx);
  p.chunks = (int)chunks; p.chunk_px = (int)cpx;
  if (base * chunks > 0x7fffffffLL) return BTX_E_UNSUPPORTED;
  if (slab && (size_t)chunks * nk * E * sizeof(float) > ws_bytes) return BTX_E_WORKSPACE;
  p.direct = (slab && chunks == 1) ? 1 : 0;
  const bool swap = (flags & BTX_FLAG_SWAP_SIGNS) != 0;  // transposed layers: the roles of x and dy are exchanged
  sign_keys_host(rng, swap ? BTX_STREAM_SIGN_OUT : BTX_STREAM_SIGN_IN, &p.kin_a, &p.kin_b);
  sign_keys_host(rng, swap ? BTX_STREAM_SIGN_IN : BTX_STREAM_SIGN_OUT, &p.kout_a, &p.kout_b);
  p.sample_ptr = rng->sample_idx_dev; p.seed_lo = (uint32_t)rng->seed; p.seed_hi = (uint32_t)(rng->seed >> 32); p.layer = rng->layer_id;
  p.swap = swap ? 1 : 0;
  { const char* tn = wg_tune_env("BTX_WGRAD_T3_ABL"); p.tune = tn ? atoi(tn) : 0; }
  p.fd_Wo = make_fastdiv((uint32_t)p.Wo); p.fd_Ho = make_fastdiv((uint32_t)p.Ho); p.fd_Do = make_fastdiv((uint32_t)p.Do);
  p.fd_T = make_fastdiv((uint32_t)p.Tw); p.fd_ctiles = make_fastdiv((uint32_t)p.ctiles);
  p.fd_ntiles = make_fastdiv((uint32_t)p.ntiles); p.fd_groups = make_fastdiv((uint32_t)p.groups);
  hipStream_t st = (hipStream_t)stream;
  const size_t wbytes = E * sizeof(float);
  hipError_t e;
  if (!slab) {  // the atomics accumulate into dW: it must be zero on entry
    e = hipMemsetAsync(dw_mu, 0, wbytes, st);
    if (e != hipSuccess) return (int)e;
    if (kind == BTX_KIND_FLIPOUT) { e = hipMemsetAsync(dw_delta, 0, wbytes, st); if (e != hipSuccess) return (int)e; }
  }
  if (db_mu) {
    e = hipMemsetAsync(db_mu, 0, (size_t)g->N * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (kind == BTX_KIND_FLIPOUT) { e = hipMemsetAsync(db_delta, 0, (size_t)g->N * sizeof(float), st); if (e != hipSuccess) return (int)e; }
  }
  const int nwg = (int)(base * chunks);
  if (taps3) {
#define BTX_LAUNCH_T3(KIND)                                                                                       \
  do {                                                                                                            \
    auto kfn = wgrad_taps3_kernel<KIND>;                                                                           \
    static bool attr_done = false;                                                                                \
    if (!attr_done) {                                                                                             \
      hipError_t e2 = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);  \
      if (e2 != hipSuccess) return (int)e2;                                                                       \
      attr_done = true;                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(KIND == 1 ? 768 : 384), T3Lds<KIND>::total, st, p);                    \
  } while (0)
    if (kind == 0) BTX_LAUNCH_T3(0); else BTX_LAUNCH_T3(1);
#undef BTX_LAUNCH_T3
  } else {
  const int lds = (kind == BTX_KIND_FLIPOUT ? 4 : 2) * WG_TILE;
  const int lds_need = lds > 65536 ? lds : 65536;  // the cross-wave reduction uses 64 KiB
#define BTX_LAUNCH_WG(ACT, KIND, FAST)                                                                              \
  do {                                                                                                            \
    auto kfn = wgrad_kernel<ACT, KIND, FAST>;                                                                      \
    static bool attr_done = false;                                                                                \
    if (!attr_done) {                                                                                             \
      hipError_t e2 = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);  \
      if (e2 != hipSuccess) return (int)e2;                                                                       \
      attr_done = true;                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(256), lds_need, st, p);                                               \
  } while (0)
  if (act_dtype == BTX_ACT_F32) { if (kind == 0) BTX_LAUNCH_WG(float, 0, false); else BTX_LAUNCH_WG(float, 1, false); }
  else if (fast_ok) { if (kind == 0) BTX_LAUNCH_WG(__bf16, 0, true); else BTX_LAUNCH_WG(__bf16, 1, true); }
  else { if (kind == 0) BTX_LAUNCH_WG(__bf16, 0, false); else BTX_LAUNCH_WG(__bf16, 1, false); }
#undef BTX_LAUNCH_WG
  }
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  RhoFuse rf;
  memset(&rf, 0, sizeof(rf));
  if (rho_w) {
    if (E > 0xfffffffcULL) return BTX_E_UNSUPPORTED;  // BTX-RNG v1 block index is 32 bits
    rf.rho = rho_w; rf.drho = drho; rf.sample_ptr = rng->sample_idx_dev; rf.k0 = (uint32_t)rng->seed; rf.k1 = (uint32_t)(rng->seed >> 32);
    rf.sample = rng->sample_idx; rf.layer = rng->layer_id; rf.which = nk - 1;
  }
  if (slab && !p.direct) {
    const bool vec = (E % 4 == 0) && ((((uintptr_t)dw_mu) | ((uintptr_t)dw_delta) | ((uintptr_t)rho_w) | ((uintptr_t)drho)) % 16 == 0);
    int cl_log2 = 0;
    while (cl_log2 < 4 && (2 << cl_log2) <= chunks) ++cl_log2;  // up to 16 chunk lanes
    const int QL = 256 >> cl_log2;
    const size_t groups_e = vec ? (E + 3) / 4 : E;
    const size_t gx = (groups_e + QL - 1) / QL;
    if (gx > 0x7fffffffULL) return BTX_E_UNSUPPORTED;
    if (vec) hipLaunchKernelGGL(wgrad_finish_kernel<4>, dim3((unsigned)gx, nk), dim3(256), 0, st, (const float*)ws, dw_mu, dw_delta, E, (int)chunks, nk, cl_log2, rf);
    else hipLaunchKernelGGL(wgrad_finish_kernel<1>, dim3((unsigned)gx, nk), dim3(256), 0, st, (const float*)ws, dw_mu, dw_delta, E, (int)chunks, nk, cl_log2, rf);
    e = hipGetLastError();
  } else if (rho_w) {
    size_t blocks = ((E + 3) / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_rho_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)(nk == 2 ? dw_delta : dw_mu), E, rf);
    e = hipGetLastError();
  }
  return (int)e;
}

}  // namespace

extern "C" int btx_contract_wgrad(int kind, const BtxGeom* g, const void* x, const void* dy, float* dw_mu, float* dw_delta,
                                  float* db_mu, float* db_delta, const BtxRng* rng, const BtxNoise* noise, int act_dtype,
                                  uint32_t flags, void* stream) {
  return wgrad_impl(kind, g, x, dy, dw_mu, dw_delta, db_mu, db_delta, rng, noise, act_dtype, flags, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int btx_contract_wgrad_ws(int kind, const BtxGeom* g, const void* x, const void* dy, float* dw_mu, float* dw_delta,
                                     float* db_mu, float* db_delta, const BtxRng* rng, const BtxNoise* noise, int act_dtype,
                                     uint32_t flags, void* ws, size_t ws_bytes, const float* rho_w, float* drho, void* stream) {
  if (!ws || ws_bytes == 0) return BTX_E_WORKSPACE;
  return wgrad_impl(kind, g, x, dy, dw_mu, dw_delta, db_mu, db_delta, rng, noise, act_dtype, flags, ws, ws_bytes, rho_w, drho, stream);
}

// the slabs of either kernel (the choice between them also depends on pointers and on the bias: the size covers both)
extern "C" size_t btx_wgrad_workspace_bytes(int kind, const BtxGeom* g, int act_dtype, uint32_t flags) {
  if (!g || (kind != BTX_KIND_REPARAM && kind != BTX_KIND_FLIPOUT)) return 0;
  WgradParams p;
  if (wgrad_fill_params(kind, g, nullptr, nullptr, nullptr, flags, p)) return 0;
  const size_t nk = kind == BTX_KIND_FLIPOUT ? 2 : 1, E = (size_t)g->N * p.K;
  long long chunks, cpx;
  wgrad_chunks((long long)p.groups * p.ntiles * p.ctiles * p.T, p.M, wgrad_target_wgs(true), true, &chunks, &cpx);
  long long most = chunks;
  if (p.Cg == 32 && p.T >= 2) {  // the two-taps-per-workgroup form (alignment decides at launch: the size covers both)
    wgrad_chunks((long long)p.groups * p.ntiles * p.ctiles * ((p.T + 1) / 2), p.M, wgrad_target_wgs(true), true, &chunks, &cpx);
    if (chunks > most) most = chunks;
  }
  if (wgrad_taps3_geom(p, act_dtype)) {
    wgrad_chunks((long long)p.ntiles * p.ctiles, p.M, wgrad_taps3_target_wgs(), true, &chunks, &cpx);
    if (chunks > most) most = chunks;
  }
  return (size_t)most * nk * E * sizeof(float);
}
