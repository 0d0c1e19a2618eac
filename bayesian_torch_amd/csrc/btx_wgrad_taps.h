// btx_wgrad_taps.h — weight gradient of a stride-1 3x3 "same" convolution (the body of a ResNet), all nine taps in ONE workgroup.
// Included by btx_wgrad.hip (inside its anonymous namespace, after WgradParams).
//
// wgrad_kernel gives a workgroup ONE tap: the same dy tile is fetched and staged nine times, the same x tile once per n-tile,
// 16 KB of L2 traffic and 32 KB of LDS stores per 32 MFMAs — it runs at a tenth of the matrix rate (profiles/r05_experiments.txt
// E10).  Here a workgroup owns a 64 n x 64 c tile of dW for ALL nine taps over a chunk of output pixels:
//   * x is staged ONCE per pixel, as a ring of 256 input pixels in raster order (at stride 1 / padding 1 the pixel under tap
//     (kh, kw) of output pixel m is raster m + (kh-1) W + (kw-1)); the nine tap-shifted operands are nine ADDRESSES into the
//     ring, not nine tiles.  The MFMA fragment (8 consecutive pixels of one channel) comes from ds_read_b64_tr_b16, whose lanes
//     each bring their own row address: a pixel whose tap falls outside the image points at a row of zeros.
//   * per 64-pixel step a workgroup stages 16 KB (+ the signed copies) for 288 MFMAs instead of 16 KB per 32;
//   * wave = (kind: mean | delta, c half j, kernel row kh): 64 n x 32 c x 3 taps = 96 accumulator registers, the dy fragments
//     shared by its three taps (10 transpose reads per 6 MFMAs).  No cross-wave reduction: every wave owns its output tiles.
//   * ONE barrier per step: dy tiles are double-buffered, the ring rows written during a step (the 64 pixels the NEXT step adds)
//     are the ones the previous step read last.
// The partial sums of a chunk leave as plain stores into the chunk's slab (btx_contract_wgrad_ws); wgrad_finish_kernel adds the
// slabs in a fixed order — f32 atomics are fabric transactions on this part, a slab pass costs less and is deterministic.
#pragma once

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int T3_R = 256;                  // ring rows (input pixels, raster index modulo 256)
constexpr int T3_XBLK = T3_R * 32 + 128;   // bytes per 16-channel block of a ring: [256 rows][16 ch] bf16, 32-byte rows.
constexpr int T3_YBLK = 64 * 32 + 128;     // bytes per 16-channel block of a dy tile: [64 px][16 n]
// both strides are 128 mod 256: the two 16-lane groups of a transpose read (channel blocks 2i, 2i+1) use disjoint bank halves
constexpr int T3_XRING = 4 * T3_XBLK, T3_YTILE = 4 * T3_YBLK;

template <int KIND>
struct T3Lds {
  static constexpr int NK = KIND == 1 ? 2 : 1;
  static constexpr int x(int k) { return k * T3_XRING; }
  static constexpr int y(int k, int b) { return NK * T3_XRING + (k * 2 + b) * T3_YTILE; }
  static constexpr int zero = NK * T3_XRING + NK * 2 * T3_YTILE;
  static constexpr int total = zero + 64;
};

// W + 1 <= 64: the rows a step reads, [m0 - W - 1, m0 + 63 + W + 1], lie inside the 192 live rows [m0 - 64, m0 + 128)
__host__ inline bool wgrad_taps3_ok(const WgradParams& p, int act_dtype, bool bias) {
  return act_dtype == BTX_ACT_BF16 && !bias && !p.sign_in && !p.sign_out && p.groups == 1 && p.D == 1 && p.KD == 1 && p.KH == 3 &&
         p.KW == 3 && p.sh == 1 && p.sw == 1 && p.ph == 1 && p.pw == 1 && p.dh == 1 && p.dw == 1 && p.Ho == p.H && p.Wo == p.W &&
         p.W >= 2 && p.W <= 63 && p.H >= 2 && (p.C % 64) == 0 && (p.N % 64) == 0 && p.slab != nullptr &&
         ((((uintptr_t)p.x) | ((uintptr_t)p.dy)) % 16 == 0);
}

template <int KIND>
__global__ __launch_bounds__(KIND == 1 ? 768 : 384) void wgrad_taps3_kernel(const WgradParams p) {
  constexpr int NK = KIND == 1 ? 2 : 1, NT = 384 * NK, PIECES = (512 + NT - 1) / NT;
  using L = T3Lds<KIND>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
  typedef __attribute__((address_space(3))) s16x4* lds_frag;
  const lds_bytes lds = (lds_bytes)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hk = lane >> 5;

  uint32_t kin_a = p.kin_a, kin_b = p.kin_b, kout_a = p.kout_a, kout_b = p.kout_b;
  if (KIND == 1 && p.sample_ptr) {  // captured training steps: the sign keys of the sample the device-resident index names
    const uint32_t smp = __builtin_amdgcn_readfirstlane(*p.sample_ptr);
    const uint32_t si = p.swap ? BTX_STREAM_SIGN_OUT : BTX_STREAM_SIGN_IN, so = p.swap ? BTX_STREAM_SIGN_IN : BTX_STREAM_SIGN_OUT;
    const BtxPhilox4 ki = btx_philox4x32_10(0u, smp, p.layer, si, p.seed_lo, p.seed_hi);
    const BtxPhilox4 ko = btx_philox4x32_10(0u, smp, p.layer, so, p.seed_lo, p.seed_hi);
    kin_a = __builtin_amdgcn_readfirstlane(ki.x[0]); kin_b = __builtin_amdgcn_readfirstlane(ki.x[1]);
    kout_a = __builtin_amdgcn_readfirstlane(ko.x[0]); kout_b = __builtin_amdgcn_readfirstlane(ko.x[1]);
  }

  uint32_t u, u_ct, u_nt, u_chunk;
  fdivmod(blockIdx.x, p.fd_ctiles, (uint32_t)p.ctiles, u, u_ct);
  fdivmod(u, p.fd_ntiles, (uint32_t)p.ntiles, u_chunk, u_nt);
  const int ct = (int)u_ct, nt = (int)u_nt, chunk = (int)u_chunk;
  const int m_begin = chunk * p.chunk_px, m_end = min(p.M, m_begin + p.chunk_px);
  const int W = p.W, H = p.H;

  // role of the wave: (kind, c half, kernel row)
  const int kind = KIND == 1 ? (wave & 1) : 0, wr = KIND == 1 ? (wave >> 1) : wave;
  const int j = wr & 1, kh = wr >> 1;

  // ---- staging: a 64-row block of dy (64 n) or x (64 c) = 512 pieces of 16 bytes.  Piece pc: rows 8 (pc >> 6) .. +7 by wave
  // instruction; inside it eight lanes = (four rows) x (the two halves of ONE 16-channel block) = 128 contiguous LDS bytes.
  struct Piece { u32x4 v; uint32_t flat; };
  auto piece_geom = [&](int pc, int& row, int& q, int& half) __attribute__((always_inline)) {
    const int l = pc & 63, g8 = l >> 3;
    half = l & 1; q = g8 & 3;
    row = 8 * (pc >> 6) + 4 * (g8 >> 2) + ((l >> 1) & 3);
  };
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  auto fetch_dy = [&](int m0, Piece (&pc_)[PIECES]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
      const int pc = tid + k * NT;
      if (pc < 512) {
        int row, q, half;
        piece_geom(pc, row, q, half);
        const int m = m0 + row;
        const bool ok = m < m_end;
        const long long off = (long long)(ok ? m : 0) * p.N + nt * 64 + 16 * q + 8 * half;
        const u32x4 v = *(const u32x4*)((const uint16_t*)p.dy + off);
        pc_[k].v = ok ? v : z4;
        pc_[k].flat = (uint32_t)off;
      }
    }
  };
  auto fetch_x = [&](int r0, Piece (&pc_)[PIECES]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
      const int pc = tid + k * NT;
      if (pc < 512) {
        int row, q, half;
        piece_geom(pc, row, q, half);
        const int r = r0 + row;
        const bool ok = r >= 0 && r < p.M;
        const long long off = (long long)(ok ? r : 0) * p.C + ct * 64 + 16 * q + 8 * half;
        const u32x4 v = *(const u32x4*)((const uint16_t*)p.x + off);
        pc_[k].v = ok ? v : z4;
        pc_[k].flat = (uint32_t)off;
      }
    }
  };
  // one hashed word covers an aligned 32-element run; pair j of the word has its signs at bits 15 - j and 31 - j
  auto signed_copy = [&](const Piece& pc_, uint32_t ka, uint32_t kb) __attribute__((always_inline)) -> u32x4 {
    const uint32_t w = btx_sign_word(pc_.flat >> 5, ka, kb) << ((pc_.flat & 31u) >> 1);
    u32x4 s;
#pragma unroll
    for (int d = 0; d < 4; ++d) s[d] = pc_.v[d] ^ ((w << d) & 0x80008000u);
    return s;
  };
  auto stash_dy = [&](const Piece (&pc_)[PIECES], int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
      const int pc = tid + k * NT;
      if (pc < 512) {
        int row, q, half;
        piece_geom(pc, row, q, half);
        const int off = q * T3_YBLK + row * 32 + half * 16;
        *(u32x4*)(smem + L::y(0, buf) + off) = pc_[k].v;
        if constexpr (KIND == 1) *(u32x4*)(smem + L::y(1, buf) + off) = signed_copy(pc_[k], kout_a, kout_b);
      }
    }
  };
  auto stash_x = [&](const Piece (&pc_)[PIECES], int r0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
      const int pc = tid + k * NT;
      if (pc < 512) {
        int row, q, half;
        piece_geom(pc, row, q, half);
        const int off = q * T3_XBLK + ((r0 + row) & (T3_R - 1)) * 32 + half * 16;
        *(u32x4*)(smem + L::x(0) + off) = pc_[k].v;
        if constexpr (KIND == 1) *(u32x4*)(smem + L::x(1) + off) = signed_copy(pc_[k], kin_a, kin_b);
      }
    }
  };

  // ---- prologue: the row of zeros, ring rows [m_begin - 64, m_begin + 128), the dy tile of the first step
  if (tid < 16) *(uint32_t*)(smem + L::zero + 4 * tid) = 0u;
  {
    Piece a0[PIECES], a1[PIECES], a2[PIECES], d0[PIECES];
    fetch_x(m_begin - 64, a0);
    fetch_x(m_begin, a1);
    fetch_x(m_begin + 64, a2);
    fetch_dy(m_begin, d0);
    stash_x(a0, m_begin - 64);
    stash_x(a1, m_begin);
    stash_x(a2, m_begin + 64);
    stash_dy(d0, 0);
  }
  __syncthreads();

  f32x16 acc[2][3];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment geometry of the lane: 16-lane group g16 = channel block (2 half + g16); lane c16 addresses row c16 >> 2 of the 4-pixel
  // block, columns 4 (c16 & 3).. and receives channel c16 of the four pixels
  const int c16 = lane & 15, g16 = (lane >> 4) & 1;
  const int rsel = c16 >> 2, csel = (c16 & 3) * 8;
  const int xring = L::x(kind) + (2 * j + g16) * T3_XBLK + csel;
  const int zrow = L::zero + csel;
  const int tap_shift = (kh - 1) * W - 1;  // raster offset of tap (kh, kw = 0)
  const int oh_lo = kh == 0 ? 1 : 0, oh_hi = kh == 2 ? H - 2 : H - 1;  // output rows whose tap row kh lies inside the image

  int buf = 0;
  for (int m0 = m_begin; m0 < m_end; m0 += 64, buf ^= 1) {
    const bool more = m0 + 64 < m_end;
    Piece ndy[PIECES], nx[PIECES];
    if (more) {  // the next step's dy tile and the 64 ring rows it adds: in flight during the MFMAs below
      fetch_dy(m0 + 64, ndy);
      fetch_x(m0 + 128, nx);
    }
    const int ytile = L::y(kind, buf) + g16 * T3_YBLK + (8 * hk + rsel) * 32 + csel;
#if defined(BTX_TUNING)
    if (!(p.tune & 2))
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 a[2], b[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ao = ytile + 2 * i * T3_YBLK + ks * 512;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag)(lds + ao));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag)(lds + ao + 128));
        a[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
      int bo[2][3];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pp = m0 + 16 * ks + 8 * hk + 4 * h + rsel;  // output pixel of this lane's row in the 4-pixel block
        const uint32_t t = fdiv((uint32_t)pp, p.fd_Wo);
        const int ow = pp - (int)t * W;
        const int oh = (int)t - (int)fdiv(t, p.fd_Ho) * H;
        // (no short-circuit: the conditions are lane data, a branch per pixel costs more than the compares)
        const bool rv = (pp < p.M) & (oh >= oh_lo) & (oh <= oh_hi);
        const int r5 = (pp + tap_shift) << 5;
        const int a0 = xring + (r5 & ((T3_R - 1) << 5)), a1 = xring + ((r5 + 32) & ((T3_R - 1) << 5)),
                  a2 = xring + ((r5 + 64) & ((T3_R - 1) << 5));
        bo[h][0] = (rv & (ow >= 1)) ? a0 : zrow;
        bo[h][1] = rv ? a1 : zrow;
        bo[h][2] = (rv & (ow <= W - 2)) ? a2 : zrow;
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag)(lds + bo[0][kw]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag)(lds + bo[1][kw]));
        b[kw] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[i][kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[kw], acc[i][kw], 0, 0, 0);
    }
    if (more) {
      stash_dy(ndy, buf ^ 1);
      stash_x(nx, m0 + 128);
    }
    __syncthreads();
  }

  // ---- the chunk's partial sums: C/D layout of 32x32 MFMAs, reg r of lane (l31, hk) = D[n = (r&3) + 8 (r>>2) + 4 hk][c = l31]
  const size_t E = (size_t)p.N * 9 * p.C;
  float* slab = p.direct ? (kind ? p.dwd : p.dwm) : p.slab + ((size_t)chunk * NK + kind) * E;
#if defined(BTX_TUNING)
  if (p.tune & 1) return;
#endif
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hk, c = ct * 64 + 32 * j + l31;
        slab[((size_t)n * 9 + (kh * 3 + kw)) * p.C + c] = acc[i][kw][r];
      }
}
