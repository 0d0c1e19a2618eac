// btx_contract.h — fused sample-and-implicit-GEMM forward for gfx950 (MI355X).
//
// One kernel family covers Linear / Conv{1,2,3}d / ConvTranspose{1,2,3}d x {Reparameterization, Flipout}
// (reference: layers/variational_layers/{linear,conv}_variational.py, layers/flipout_layers/{linear,conv}_flipout.py;
// exact line ranges in include/btx.h).  DESIGN.md §5 has the derivation; the short version:
//
//   out[pix][n] = sum_k Wmu[n][k] * A[pix][k]  +  s_out[pix][n] * sum_k Wdl[n][k] * (A[pix][k] * s_in[pix][k])
//
//   * implicit GEMM, pixels on the MFMA column axis, output channels on the row axis (operands swapped on
//     purpose: a lane then owns ONE pixel and 4-channel runs, so s_out is one hashed word per (pixel, 32
//     channels) and stores are 16-byte channel-contiguous pieces of a channels-last tensor);
//   * workgroup tile 256 pixels x 64 channels, 8 waves laid out 4 (pixels) x 2 (channels); a wave owns
//     64 pixels x 32 channels = 2 tiles of v_mfma_f32_32x32x{16_bf16 | 2_f32}; Flipout keeps TWO accumulator
//     sets (64 accumulator registers) — the largest tile that stays inside the 256-VGPR budget of two waves
//     per SIMD, which this VALU-heavy kernel needs (one wave per SIMD issues VALU at half rate);
//   * a K-stage is 4 "granule rows"; a granule = 16 bytes = 8 bf16 / 4 f32 consecutive k of one pixel / one
//     output channel.  LDS image [row][pixel|channel][16 B] -> every fragment read is one conflict-free
//     ds_read_b128 and every stage write one conflict-free ds_write_b128 / _b64;
//   * (mu, rho) are read f32 from HBM, softplus + Philox/Box–Muller run in registers (one Philox call per
//     thread per stage = 4 weights) and only the sampled bf16/f32 values are written to LDS: the sampled
//     weight never exists in HBM;
//   * s_in never exists as a tensor: the staging thread hashes ONE word per (pixel, 32 channels), leaves it
//     beside the activations in LDS, and the consuming lane turns it into a sign-bit XOR mask
//     ((w << d) & 0x80008000) on the fragment registers between the mu-MFMAs and the delta-MFMAs;
//   * waves 0-3 run [MFMA ; stage next] and waves 4-7 run [stage next ; MFMA] so the two waves that share a
//     SIMD keep its matrix pipe and its VALU busy at the same time (MI355X_MICROARCH "two waves per SIMD");
//   * small-M layers get their parallelism from split-K (partial sums in a workspace, deterministic second
//     pass) instead of smaller, even more sampling-bound tiles.
//
// FAST variants (GEN=false) need whole 16-byte granules (C/groups % 8 == 0 for bf16, % 4 for f32, aligned
// pointers) and generate all noise in-kernel.  GEN=true is the element-wise gather path: any shape, and the
// explicit-noise parity mode (eps / sign tensors supplied by the caller).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "btx_rng.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

namespace btx {

constexpr int NTHREADS = 512;
constexpr int BM = 256;        // pixels per workgroup tile
constexpr int BN = 64;         // output channels per workgroup tile
constexpr int NG = 4;          // granule rows per K-stage
constexpr int ACTS_OFF = 0;                      // [NG][BM] x 16 B
constexpr int SIGN_OFF = NG * BM * 16;           // [NG][BM] x 4 B
constexpr int WMU_OFF = SIGN_OFF + NG * BM * 4;  // [NG][BN] x 16 B
constexpr int WDL_OFF = WMU_OFF + NG * BN * 16;  // [NG][BN] x 16 B
constexpr int STAGE_BYTES = WDL_OFF + NG * BN * 16;  // 28672
constexpr int LDS_BYTES = 2 * STAGE_BYTES;           // 57344

// Division by a launch-invariant divisor (Granlund-Montgomery / Hacker's Delight 10-9): the host precomputes
// (m, sh1, sh2), the device spends one mul_hi and four integer ops instead of the ~45-instruction generic sequence.
// The tile prologues decode ~20 pixel indices per lane; with generic divisions that alone cost ~8000 cycles per block.
struct FastDiv {
  uint32_t m, sh;  // sh = sh1 | sh2 << 8
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2 d)
  f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  f.sh = (l < 1 ? l : 1) | ((l > 0 ? l - 1 : 0) << 8);
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t x, const FastDiv& f) {
  const uint32_t t = __umulhi(x, f.m);
  return (t + ((x - t) >> (f.sh & 31u))) >> (f.sh >> 8);
}
// quotient and remainder; d is the divisor the FastDiv was made for
__device__ __forceinline__ void fdivmod(uint32_t x, const FastDiv& f, uint32_t d, uint32_t& q, uint32_t& r) {
  q = fdiv(x, f);
  r = x - q * d;
}

// patch variant (btx_contract_patch.h)
constexpr int PT_WD = 4;  // depth of the weight-tile ring: W(s+3) is fetched while stage s multiplies and s+1 is read
constexpr int PT_EP_ROW = 272;
constexpr int PT_EP_WAVE = 64 * PT_EP_ROW;  // 17408: epilogue staging per wave
constexpr int PT_MAXNI = 8;  // 1-KiB DMA instructions per wave per patch slot (host plan keeps pieces <= NW * 8)

struct ContractParams {
  const void* x;
  const float* mu;
  const float* rho;
  const float* mu_b;
  const float* rho_b;
  void* out;
  float* partial;  // split-K workspace [ksplits][M][N] f32 (ksplits > 1)
  const float* eps_w;      // explicit noise: GEN kernels only
  const float* eps_b;
  const int8_t* sign_in;
  const int8_t* sign_out;
  int NB, D, H, W, C, Cg;
  int Do, Ho, Wo, N, Ng;
  int KD, KH, KW;
  int sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int M, K;
  int mtiles, ntiles, groups, ksplits, kper;
  int transposed;
  int pointwise;   // 1x1x1 filter, stride 1, no padding, not transposed: output pixel m reads input pixel m (a plain GEMM)
  uint32_t seed_lo, seed_hi, sample, layer;
  const uint32_t* sample_ptr;  // non-NULL: the MC sample index is read from device memory when the kernel runs
  uint32_t kin_a, kin_b, kout_a, kout_b;
  const float* ep_scale;  // fused epilogue (BtxEpilogue), all nullable
  const float* ep_shift;
  const void* ep_res;
  int ep_relu;
  int out_bf16;        // output element type: 1 bf16, 0 f32 (DMA variant; the others store the activation dtype)
  int sign_unaligned;  // DMA variant: a stage's elements may straddle two 32-sign words (row-fused stems)
  uint32_t x_bytes, w_bytes;  // sizes of x and of mu/rho in bytes (buffer descriptors of the DMA variant)
  // patch variant (btx_contract_patch.h): tile = pt_G images x pt_R output rows x Wo; patch = pt_G x pt_Rp x pt_Wp pixels
  int pt_G, pt_R, pt_Rp, pt_Wp, pt_PP, pt_NI, pt_rtiles;
  FastDiv fd_Wo, fd_Ho, fd_Do, fd_ptWp, fd_ptRp, fd_ptR;
  FastDiv fd_inner, fd_ksplits, fd_ntiles, fd_Cg, fd_KW, fd_KH, fd_rtiles;  // wave-uniform index splits
  FastDiv fd_mtiles;
  int swap_signs;  // BTX_FLAG_SWAP_SIGNS: input signs from stream SIGN_OUT, output signs from SIGN_IN
  // Parity-major pixel order of a stride-2 transposed 2-D launch (btx_api.hip par_major_ok: the data gradient of a stride-2
  // convolution): logical pixel L = class * par_Mqp + q, class = 2 * (oh & 1) + (ow & 1), q < par_Mq the raster index of
  // (image, oh / 2, ow / 2) on the par_Hh x par_Wh half-resolution grid, par_Mqp = par_Mq rounded up to whole pixel tiles.  A tile
  // then lies inside ONE class, and of the KH x KW taps only those whose stride residue matches the class (1, 2, 2 or 4 of a 3x3
  // filter's 9) reach any of its pixels: the others are not walked.  Workgroup t takes class t & 3, tile t >> 2 of that class.
  int par_major, par_Mq, par_Mqp, par_Hh, par_Wh;
  FastDiv fd_par_Mqp, fd_par_Hh, fd_par_Wh;
  FastDiv fd_sd, fd_sh, fd_sw;  // transposed launches: the gather rule divides by the strides once per pixel, tap and K stage
  int wg_order;  // 0: workgroups that share a pixel tile are neighbours (same XCD L2 holds the activations);
                 // 1: workgroups that share a weight tile are (layers whose sampled weights outweigh their activations)
  void* trace;  // BTX_PT_TRACE builds: per-wave phase timings (measurement only)
  int pt_nw, pt_astage, pt_lds;
  int pt_mi;      // patch variant: 32-pixel MFMA tiles per wave (2 | 4)
  int pt_tune;    // BTX_PT_TRACE builds: bit 6 = report the per-stage split instead of the phase timers
  int pt_nopw;    // tuning builds (BTX_NO_DMA_PW=1): the generic LDS-DMA kernel for pointwise shapes too (A/B)
  int pt_kg;      // tap-unrolled kernel: K-groups per workgroup (1 | 2)
  int pt_lds_g;   //                      LDS bytes of one K-group
  int pt_taps;    // 10*KH + KW when the tap-unrolled kernel (btx_contract_taps.h) takes the launch, else 0
  int pt_wide;    // tap-unrolled kernel, Reparameterization: 64-pixel x 128-channel wave tiles; the grid has ntiles / 2 n-tiles
                  // (fd_ntiles and fd_inner are made for that count), workgroup n-tile t contracts weight tiles 2t and 2t + 1
  int sp_Hq, sp_Wq;  // stem + max-pool variant (btx_contract_stempool.h): pooled extent
  int st_sbytes;  // stem variant: bytes of the s_in word array in LDS  // waves per block (4 | 8), bytes per patch slot, dynamic LDS bytes of the block
  void* wt;  // pre-sampled weight tiles (workspace): [group*ntiles + ntile][K/G][64][16 B]; delta array at +wt_delta_off
  uint32_t wt_bytes, wt_delta_off;
  int wt_ready;  // wt was filled by btx_sample_weights: skip the per-launch sampling pre-pass
  // "tall strip" tiles of the tap-unrolled kernel (btx_contract_taps.h): the batch is one tall image of pt_P virtual rows
  // per image (the image's Ho output rows + dummy rows; on the input side pt_P - H zero rows separate consecutive
  // images and serve as the bottom padding of one and the top padding of the next), cut into tiles of pt_R virtual rows
  // x pt_Wt columns (pt_ncs column strips per row tile).  Tile sizes are then free of the divisors of Ho and Wo.
  int pt_tall, pt_P, pt_Wt, pt_ncs;
  FastDiv fd_P, fd_Wt, fd_ncs;
  // MC sample lanes (BtxLanes): the grid holds `lanes` copies of the single-sample grid of `lane_nwg` workgroups; lane l
  // reads x + l*lane_x, the weight tiles wt + l*lane_wt, the sample word sample_ptr[l] (or sample + l), and writes
  // out + l*lane_out (residual + l*lane_res, split-K partials + l*lane_partial): byte strides
  int lanes, lane_nwg;
  FastDiv fd_lane_nwg;
  long long lane_x, lane_out, lane_res, lane_wt, lane_partial;
  int lane_wt_delta;  // 1 (Flipout): the lanes share the mu tiles at wt, lane l's delta tiles sit at wt_delta_off + l*lane_wt
  int reverse;        // BTX_FLAG_REVERSE: the grid's tiles in descending order
  int ep_direct;      // the store side runs from the fragment registers (direct_epilogue_pm, btx_epilogue.h): host-checked
  int pt_persist;     // > 0: the persistent form of the tap-unrolled kernel (btx_contract_taps3.h) with this many workgroups
  uint32_t* pt_queue; // persistent form: one zeroed counter per tile position (the image-group queue of its workgroups)
};

// the MC sample word of a launch whose index lives in device memory, as a SCALAR load: the word does not change while the
// kernel runs, so it may be read through the constant address space (s_load_dword, waited for where it is used) — a plain
// dereference compiles to global_load_dword + s_waitcnt vmcnt(0) + v_readfirstlane at the head of the workgroup
__device__ __forceinline__ uint32_t sample_word_scalar(const uint32_t* ptr) {
  return *(const __attribute__((address_space(4))) uint32_t*)(uintptr_t)ptr;
}

// ---- workgroup id -> logical id, XCD-aware: block b runs on XCD b % 8; every XCD gets a contiguous range of logical ids,
// so the workgroups that share an operand (and, with MC sample lanes, the workgroups of one lane) share an L2
__device__ __forceinline__ int xcd_logical() {
  const int nwg = gridDim.x, L = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = L & 7, slot = L >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// ---- MC sample lanes: the launch parameters as the lane of this workgroup sees them; `logical` becomes the id inside
// the lane's own grid of lane_nwg workgroups.  Everything behind this call is the single-sample kernel, unchanged: the
// noise indices are relative to the lane's own tensors, so a lane computes bit for bit what a launch of its own would.
__device__ __forceinline__ ContractParams lane_view(const ContractParams& q, int& logical) {
  ContractParams p = q;
  if (q.reverse) logical = (int)gridDim.x - 1 - logical;
  if (q.lanes > 1) {
    uint32_t lane, loc;
    fdivmod((uint32_t)logical, q.fd_lane_nwg, (uint32_t)q.lane_nwg, lane, loc);
    logical = (int)loc;
    const long long l = (long long)lane;
    p.x = (const unsigned char*)q.x + l * q.lane_x;
    p.out = (unsigned char*)q.out + l * q.lane_out;
    if (q.ep_res) p.ep_res = (const unsigned char*)q.ep_res + l * q.lane_res;
    if (q.wt) {
      if (q.lane_wt_delta) p.wt_delta_off = q.wt_delta_off + (uint32_t)(l * q.lane_wt);
      else p.wt = (unsigned char*)q.wt + l * q.lane_wt;
    }
    if (q.partial) p.partial = (float*)((unsigned char*)q.partial + l * q.lane_partial);
    if (q.sample_ptr) p.sample_ptr = q.sample_ptr + lane;
    p.sample = q.sample + lane;
  }
  return p;
}

// Launch parameters of a kernel SECTION: a lane view (above) of the ~180-dword argument struct, read through a pointer to
// the kernel-argument segment that the compiler cannot connect to the pointer of another section.  A kernel that takes
// the struct by value loads every field any part of it uses at entry and keeps it to its last use: the tap-unrolled
// Flipout kernel spilt 140 SGPRs into VGPR lanes, ~40 % of the VALU instructions of its prologue and store side were
// v_readlane / v_writelane.  With one view for prologue + K loop and one for the store side: 94, prologue 10.9k -> 8.2k cycles.
#define BTX_SECTION_PARAMS(name, logical_var)                                                                         \
  const __attribute__((address_space(4))) ContractParams* name##_karg =                                               \
      (const __attribute__((address_space(4))) ContractParams*)__builtin_amdgcn_kernarg_segment_ptr();                \
  asm volatile("" : "+s"(name##_karg));                                                                               \
  int logical_var = xcd_logical();                                                                                    \
  const ContractParams name = lane_view(*(const ContractParams*)name##_karg, logical_var)

// The same view without control flow (btx_contract_gemm8.h: one workgroup per CU, so nobody covers this workgroup's scalar-load
// round trips).  lane_view() reads the lane strides inside `if (lanes > 1)`, `if (ep_res)`, ...: five dependent batches of
// s_load + s_waitcnt in front of the first DMA and five more in front of the store side.  Here every field is read
// unconditionally — one batch — and the lane arithmetic runs for lanes == 1 as well (lane_nwg is the whole grid then: lane 0,
// zero strides).
__device__ __forceinline__ ContractParams lane_view_flat(const ContractParams& q, int& logical) {
  ContractParams p = q;
  // every field the lane arithmetic reads, as a value that exists HERE (the empty asm takes them as operands): without it the
  // compiler turns the selects below back into branches and sinks each stride's load into its branch
  const int q_rev = q.reverse, q_lnwg = q.lane_nwg, q_lwd = q.lane_wt_delta;
  const FastDiv q_fd = {q.fd_lane_nwg.m, q.fd_lane_nwg.sh};
  const long long q_lx = q.lane_x, q_lo = q.lane_out, q_lr = q.lane_res, q_lw = q.lane_wt, q_lp = q.lane_partial;
  const unsigned char* q_x = (const unsigned char*)q.x;
  unsigned char* q_out = (unsigned char*)q.out;
  const unsigned char* q_res = (const unsigned char*)q.ep_res;
  unsigned char* q_wt = (unsigned char*)q.wt;
  unsigned char* q_part = (unsigned char*)q.partial;
  const uint32_t* q_sp = q.sample_ptr;
  const uint32_t q_smp = q.sample, q_wdo = q.wt_delta_off;
  asm volatile("" ::"s"(q_rev), "s"(q_lnwg), "s"(q_lwd), "s"(q_fd.m), "s"(q_fd.sh), "s"(q_lx), "s"(q_lo), "s"(q_lr), "s"(q_lw), "s"(q_lp), "s"(q_x),
               "s"(q_out), "s"(q_res), "s"(q_wt), "s"(q_part), "s"(q_sp), "s"(q_smp), "s"(q_wdo));
  logical = q_rev ? (int)gridDim.x - 1 - logical : logical;
  uint32_t lane, loc;
  fdivmod((uint32_t)logical, q_fd, (uint32_t)q_lnwg, lane, loc);
  logical = (int)loc;
  const long long l = (long long)lane;
  const long long wo = l * q_lw;
  p.x = q_x + l * q_lx;
  p.out = q_out + l * q_lo;
  p.ep_res = q_res ? (const void*)(q_res + l * q_lr) : nullptr;
  p.wt_delta_off = q_wdo + (q_lwd ? (uint32_t)wo : 0u);
  p.wt = q_wt ? (void*)(q_wt + (q_lwd ? 0ll : wo)) : nullptr;
  p.partial = q_part ? (float*)(q_part + l * q_lp) : nullptr;
  p.sample_ptr = q_sp ? q_sp + lane : nullptr;
  p.sample = q_smp + lane;
  return p;
}
#define BTX_SECTION_PARAMS_FLAT(name, logical_var)                                                                    \
  const __attribute__((address_space(4))) ContractParams* name##_karg =                                               \
      (const __attribute__((address_space(4))) ContractParams*)__builtin_amdgcn_kernarg_segment_ptr();                \
  asm volatile("" : "+s"(name##_karg));                                                                               \
  int logical_var = xcd_logical();                                                                                    \
  const ContractParams name = lane_view_flat(*(const ContractParams*)name##_karg, logical_var)


// ---- the (sample index, sign keys) a launch actually uses --------------------------------------------------------
// BtxRng.sample_idx_dev lets a captured hipGraph be replayed for successive MC samples: the index — and the Flipout
// sign keys derived from it — are then resolved on the device at run time instead of being baked into the arguments.
struct RngLive {
  uint32_t sample, kin_a, kin_b, kout_a, kout_b;
};
template <int KIND>
__device__ __forceinline__ RngLive rng_live(const ContractParams& p) {
  RngLive r = {p.sample, p.kin_a, p.kin_b, p.kout_a, p.kout_b};
  if (p.sample_ptr || p.lanes > 1) {  // lanes: the host's keys are those of lane 0
    if (p.sample_ptr) r.sample = __builtin_amdgcn_readfirstlane(sample_word_scalar(p.sample_ptr));
    if constexpr (KIND == 1) {
      const uint32_t si = p.swap_signs ? 3u : 2u, so = p.swap_signs ? 2u : 3u;  // BTX_STREAM_SIGN_IN = 2, _OUT = 3
      const BtxPhilox4 ki = btx_philox4x32_10(0u, r.sample, p.layer, si, p.seed_lo, p.seed_hi);
      const BtxPhilox4 ko = btx_philox4x32_10(0u, r.sample, p.layer, so, p.seed_lo, p.seed_hi);
      r.kin_a = __builtin_amdgcn_readfirstlane(ki.x[0]); r.kin_b = __builtin_amdgcn_readfirstlane(ki.x[1]);
      r.kout_a = __builtin_amdgcn_readfirstlane(ko.x[0]); r.kout_b = __builtin_amdgcn_readfirstlane(ko.x[1]);
    }
  }
  return r;
}

// ---- small helpers -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// pack G floats into one 16-byte LDS granule of the contraction precision
template <int PREC>
__device__ __forceinline__ u32x4 pack_granule(const float* f) {
  if constexpr (PREC == 1) {
    f32x8 v = {f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]};
    bf16x8 b = __builtin_convertvector(v, bf16x8);
    return __builtin_bit_cast(u32x4, b);
  } else {
    u32x4 r = {f2u(f[0]), f2u(f[1]), f2u(f[2]), f2u(f[3])};
    return r;
  }
}

__device__ __forceinline__ u32x2 pack_quad_bf16(const float* f) {
  f32x4 v = {f[0], f[1], f[2], f[3]};
  bf16x4 b = __builtin_convertvector(v, bf16x4);
  return __builtin_bit_cast(u32x2, b);
}

// raw activation granule as it comes from HBM: G elements of ACT
template <typename ACT, int G>
struct RawAct;
template <>
struct RawAct<float, 8> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
  __device__ __forceinline__ void get(float* f) const {
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
  }
};
template <>
struct RawAct<float, 4> {
  f32x4 a;
  __device__ __forceinline__ void load(const float* p) { a = *(const f32x4*)p; }
  __device__ __forceinline__ void get(float* f) const { f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; }
};
template <>
struct RawAct<__bf16, 8> {
  u32x4 a;
  __device__ __forceinline__ void load(const __bf16* p) { a = *(const u32x4*)p; }
  __device__ __forceinline__ void get(float* f) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = u2f(a[i] << 16); f[2 * i + 1] = u2f(a[i] & 0xffff0000u); }
  }
};
template <>
struct RawAct<__bf16, 4> {
  u32x2 a;
  __device__ __forceinline__ void load(const __bf16* p) { a = *(const u32x2*)p; }
  __device__ __forceinline__ void get(float* f) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) { f[2 * i] = u2f(a[i] << 16); f[2 * i + 1] = u2f(a[i] & 0xffff0000u); }
  }
};

// fused epilogue on 4 consecutive channels starting at (row offset `o`, channel-in-tile `cl`); `aff` = LDS copy of
// scale (at [0..BN)) and shift (at [BN..2BN)) for this n-tile, or nullptr
template <typename OUT>
__device__ __forceinline__ void apply_epilogue4(float* v, const ContractParams& p, const float* aff, int cl, long long o,
                                                int nvalid, bool vec) {
  if (aff) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], aff[cl + r], aff[64 + cl + r]);
  }
  if (p.ep_res) {
    const OUT* rp = (const OUT*)p.ep_res + o;
    if (vec) {  // same 4-channel run, same alignment as the store: one 8-byte (bf16) / 16-byte (f32) load
      if constexpr (sizeof(OUT) == 4) {
        const f32x4 rv = *(const f32x4*)rp;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rv[r];
      } else {
        const u32x2 rv = *(const u32x2*)rp;
        v[0] += u2f(rv[0] << 16); v[1] += u2f(rv[0] & 0xffff0000u);
        v[2] += u2f(rv[1] << 16); v[3] += u2f(rv[1] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (r < nvalid) v[r] += (float)rp[r];
    }
  }
  if (p.ep_relu) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
  }
}

// The 32-sign word of elements [32*(off>>5), +32) built from an explicit +1/-1 tensor (BtxNoise.sign_in, parity mode) in
// the bit order of btx_sign_word(); elements at or beyond `n` count as +1.  Slow on purpose (32 byte loads): tests only.
__device__ __forceinline__ uint32_t sign_word_explicit(const int8_t* __restrict__ s, uint32_t off, uint32_t n) {
  const uint32_t base = off & ~31u;
  uint32_t w = 0;
#pragma unroll 1
  for (uint32_t e = 0; e < 32; ++e) {
    const uint32_t i = base + e;
    if (i < n && s[i] < 0) w |= 1u << btx_sign_bitpos(e);
  }
  return w;
}

// position of granule-local element e in the pre-shifted sign word (see btx_rng.h)
__device__ __forceinline__ int ws_bit(int e) { return ((e & 1) ? 31 : 15) - (e >> 1); }

// =========================================================================================================
template <int PREC, typename ACT, int KIND, bool GEN>
__global__ __launch_bounds__(NTHREADS, 2) void contract_kernel(const ContractParams pk) {
  int logical = xcd_logical();
  const ContractParams p = lane_view(pk, logical);
  const RngLive rl = rng_live<KIND>(p);
  constexpr int G = (PREC == 1) ? 8 : 4;   // elements per granule
  constexpr int BK = NG * G;               // k per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool upper = wave >= 4;
  const int wv_m = wave & 3;   // MFMA role: which 64-pixel block
  const int wv_n = wave >> 2;  //            which 32-channel half

  // ---- workgroup -> (m-tile, n-tile, group, k-split), XCD-aware (block b runs on XCD b%8: give every XCD a
  //      contiguous chunk of logical ids so the n-tiles / k-splits that share an activation tile share an L2)
  const int inner = p.ntiles * p.groups * p.ksplits;
  const int mtile = logical / inner;
  int rem = logical - mtile * inner;
  const int split = rem % p.ksplits;
  rem /= p.ksplits;
  const int ntile = rem % p.ntiles;
  const int group = rem / p.ntiles;

  const int k_begin = split * p.kper;
  const int k_end = min(p.K, k_begin + p.kper);
  const int nstages = (k_end - k_begin + BK - 1) / BK;

  // ---- staging role: thread stages granule rows {2*rp, 2*rp+1} of pixel (tid & 255); rp = tid>>8 is
  //      wave-uniform, lanes are consecutive pixels -> contiguous 16-byte LDS writes
  const int spix = tid & (BM - 1);
  const int rp = wave >> 2;
  const int m = mtile * BM + spix;
  const bool pix_valid = m < p.M;
  int bd, bh, bw, nbase;
  {
    const int mm = pix_valid ? m : 0;
    const int ow = mm % p.Wo;
    int t = mm / p.Wo;
    const int oh = t % p.Ho;
    t /= p.Ho;
    const int od = t % p.Do;
    const int nb = t / p.Do;
    nbase = nb * p.D;
    if (!p.transposed) {
      bd = od * p.sd - p.pd; bh = oh * p.sh - p.ph; bw = ow * p.sw - p.pw;
    } else {
      bd = od + p.pd; bh = oh + p.ph; bw = ow + p.pw;
    }
  }
  const int cbase = group * p.Cg;
  const ACT* __restrict__ xptr = (const ACT*)p.x;

  // ---- uniform K-walk state of the loader: next k to load and its (c, kd, kh, kw)
  int kL = k_begin;
  int s_c, s_kd, s_kh, s_kw;
  {
    const int tap = k_begin / p.Cg;
    s_c = k_begin - tap * p.Cg;
    s_kw = tap % p.KW;
    const int t2 = tap / p.KW;
    s_kh = t2 % p.KH;
    s_kd = t2 / p.KH;
  }
  auto advance = [&](int step) {  // uniform
    kL += step;
    s_c += step;
    if (s_c >= p.Cg) {
      s_c -= p.Cg;
      if (++s_kw == p.KW) { s_kw = 0; if (++s_kh == p.KH) { s_kh = 0; ++s_kd; } }
    }
  };
  // input element offset of (this pixel, current tap, channel 0 of the group); ok=false in the padding
  auto in_offset = [&](bool& ok) -> long long {
    int id, ih, iw;
    bool v = pix_valid && (kL < k_end);
    if (!p.transposed) {
      id = bd + s_kd * p.dd; ih = bh + s_kh * p.dh; iw = bw + s_kw * p.dw;
    } else {
      const int td = bd - s_kd * p.dd, th = bh - s_kh * p.dh, tw = bw - s_kw * p.dw;
      v = v && td >= 0 && th >= 0 && tw >= 0;
      id = td / p.sd; ih = th / p.sh; iw = tw / p.sw;
      v = v && (id * p.sd == td) && (ih * p.sh == th) && (iw * p.sw == tw);
    }
    v = v && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    ok = v;
    const long long pixlin = ((long long)(nbase + id) * p.H + ih) * p.W + iw;
    return v ? pixlin * p.C + cbase : 0ll;  // clamped: always a loadable address
  };

  // ---- staging registers (kept small: they are live across the MFMA block of the other LDS stage)
  RawAct<ACT, G> rawa[2];       // fast path: raw HBM granules of this thread's two rows
  bool rawok[2] = {false, false};
  float rawg[GEN ? 2 * G : 1];  // generic path: gathered elements
  uint32_t wsr[2] = {0u, 0u};   // pre-shifted sign words
  // weight sampling is done in QUADS (4 consecutive k of one output channel = one Philox call):
  //   bf16 (G=8): all 512 threads, quad = (row = tid>>7, channel = (tid>>1)&63, half = tid&1)
  //   f32  (G=4): threads 0..255,  quad = (row = tid>>6, channel = tid&63)
  constexpr bool ALLW = (G == 8);
  const bool w_thread = ALLW || (wave < 4);
  const int w_row = ALLW ? (wave >> 1) : wave;  // uniform per wave
  const int w_chan = ALLW ? ((tid >> 1) & 63) : lane;
  const int w_half = ALLW ? (tid & 1) : 0;
  float wmu_r[4], wrho_r[4];
  bool w_ok = false;
  int w_k = 0;          // first k of this thread's quad
  long long w_idx = 0;  // eps index of its first element

  // =================== L: issue the global loads of the stage starting at kL ===========================
  auto load_stage = [&]() {
    const int kstage = kL;
    // -- weights first
    if (w_thread) {
      const int col = ntile * BN + w_chan;
      w_k = kstage + w_row * G + 4 * w_half;
      const int nrow = group * p.Ng + col;
      w_idx = (long long)nrow * p.K + w_k;
      if constexpr (!GEN) {
        w_ok = (col < p.Ng) && (w_k < k_end);
        const long long li = w_ok ? w_idx : 0ll;  // clamped, branch-free
        const f32x4 a = *(const f32x4*)(p.mu + li);
        const f32x4 b = *(const f32x4*)(p.rho + li);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wmu_r[e] = a[e]; wrho_r[e] = b[e]; }
      } else {
        w_ok = (col < p.Ng);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = w_ok && (w_k + e < k_end);
          const long long li = ok ? w_idx + e : 0ll;
          wmu_r[e] = p.mu[li];
          wrho_r[e] = p.rho[li];
        }
      }
    }
    // -- activations: this thread's two granule rows of the stage
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const bool mine = (j >> 1) == rp;  // wave-uniform
      if constexpr (!GEN) {
        if (mine) {
          bool ok;
          const long long off = in_offset(ok) + s_c;
          rawa[j & 1].load(xptr + off);
          rawok[j & 1] = ok;
          if constexpr (KIND == 1) {
            const unsigned long long i0 = (unsigned long long)off;
            const uint32_t w = btx_sign_word((uint32_t)(i0 >> 5), rl.kin_a, rl.kin_b);
            const int sh = (int)((i0 >> 3) & 3) * 4 + (G == 4 ? (int)((i0 >> 2) & 1) * 2 : 0);
            wsr[j & 1] = w << sh;
          }
        }
        advance(G);
      } else {
        uint32_t ws = 0;
        uint32_t cw_i = 0xffffffffu, cw = 0;
#pragma unroll
        for (int e = 0; e < G; ++e) {
          if (mine) {
            bool ok;
            const long long off = in_offset(ok) + s_c;
            float v = 0.f;
            if (ok) {
              const unsigned long long i = (unsigned long long)off;
              v = (float)xptr[i];
              if constexpr (KIND == 1) {
                uint32_t bit;
                if (p.sign_in) {
                  bit = p.sign_in[i] < 0 ? 1u : 0u;
                } else {
                  const uint32_t wi = (uint32_t)(i >> 5);
                  if (wi != cw_i) { cw_i = wi; cw = btx_sign_word(wi, rl.kin_a, rl.kin_b); }
                  bit = (cw >> btx_sign_bitpos((uint32_t)i & 31u)) & 1u;
                }
                ws |= bit << ws_bit(e);
              }
            }
            rawg[(j & 1) * G + e] = v;
          }
          advance(1);
        }
        if (mine) wsr[j & 1] = ws;
      }
    }
  };

  // =================== P: sample / convert what L loaded and write the LDS stage ========================
  auto process_stage = [&](int buf) {
    unsigned char* sb = smem + buf * STAGE_BYTES;
    if (w_thread) {
      float eps[4];
      if constexpr (!GEN) {
        btx_normal4((uint32_t)(w_idx >> 2), rl.sample, p.layer, 0u, p.seed_lo, p.seed_hi, eps);
      } else {
        if (p.eps_w) {  // parity mode: explicit eps
#pragma unroll
          for (int e = 0; e < 4; ++e) eps[e] = (w_ok && (w_k + e < k_end)) ? p.eps_w[w_idx + e] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            eps[e] = btx_normal1((unsigned long long)(w_idx + e), rl.sample, p.layer, 0u, p.seed_lo, p.seed_hi);
        }
      }
      float wm[4], wd[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sg = btx_softplus_fast(wrho_r[e]);
        bool ok = w_ok;
        if constexpr (GEN) ok = ok && (w_k + e < k_end);
        if constexpr (KIND == 0) {
          wm[e] = ok ? __builtin_fmaf(sg, eps[e], wmu_r[e]) : 0.f;
          wd[e] = 0.f;
        } else {
          wm[e] = ok ? wmu_r[e] : 0.f;
          wd[e] = ok ? sg * eps[e] : 0.f;
        }
      }
      const int wo = (w_row * BN + w_chan) * 16 + w_half * 8;
      if constexpr (PREC == 1) {
        *(u32x2*)(sb + WMU_OFF + wo) = pack_quad_bf16(wm);
        if constexpr (KIND == 1) *(u32x2*)(sb + WDL_OFF + wo) = pack_quad_bf16(wd);
      } else {
        *(u32x4*)(sb + WMU_OFF + wo) = (u32x4){f2u(wm[0]), f2u(wm[1]), f2u(wm[2]), f2u(wm[3])};
        if constexpr (KIND == 1)
          *(u32x4*)(sb + WDL_OFF + wo) = (u32x4){f2u(wd[0]), f2u(wd[1]), f2u(wd[2]), f2u(wd[3])};
      }
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int row = 2 * rp + jj;
      u32x4 g;
      if constexpr (!GEN) {
        if constexpr (PREC == 1 && G == 8 && sizeof(ACT) == 2) {
          g = rawa[jj].a;  // bf16 activations feed the bf16 MFMA unchanged
        } else {
          float f[G];
          rawa[jj].get(f);
          g = pack_granule<PREC>(f);
        }
        if (!rawok[jj]) g = (u32x4){0u, 0u, 0u, 0u};
      } else {
        g = pack_granule<PREC>(&rawg[jj * G]);
      }
      *(u32x4*)(sb + ACTS_OFF + (row * BM + spix) * 16) = g;
      if constexpr (KIND == 1) *(uint32_t*)(sb + SIGN_OFF + (row * BM + spix) * 4) = wsr[jj];
    }
  };

  // =================== M: the MFMAs of one LDS stage =====================================================
  f32x16 accm[2], accd[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accm[a][r] = 0.f; accd[a][r] = 0.f; }

  auto mma_stage = [&](int buf) {
    const unsigned char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      const int row = 2 * kk + h;
      u32x4 a[2], wm, wd;
      uint32_t sw[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int pl = wv_m * 64 + mi * 32 + l31;
        a[mi] = *(const u32x4*)(sb + ACTS_OFF + (row * BM + pl) * 16);
        if constexpr (KIND == 1) sw[mi] = *(const uint32_t*)(sb + SIGN_OFF + (row * BM + pl) * 4);
      }
      wm = *(const u32x4*)(sb + WMU_OFF + (row * BN + wv_n * 32 + l31) * 16);
      if constexpr (KIND == 1) wd = *(const u32x4*)(sb + WDL_OFF + (row * BN + wv_n * 32 + l31) * 16);
      if constexpr (PREC == 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          accm[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wm),
                                                             __builtin_bit_cast(bf16x8, a[mi]), accm[mi], 0, 0, 0);
        if constexpr (KIND == 1) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            u32x4 a2;
#pragma unroll
            for (int d = 0; d < 4; ++d) a2[d] = a[mi][d] ^ ((sw[mi] << d) & 0x80008000u);
            accd[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wd),
                                                               __builtin_bit_cast(bf16x8, a2), accd[mi], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            accm[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(wm[e]), u2f(a[mi][e]), accm[mi], 0, 0, 0);
          if constexpr (KIND == 1) {
            const int shf = (e >> 1) + ((e & 1) ? 0 : 16);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
              const float a2 = u2f(a[mi][e] ^ ((sw[mi] << shf) & 0x80000000u));
              accd[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2f(wd[e]), a2, accd[mi], 0, 0, 0);
            }
          }
        }
      }
    }
  };

  // =================== main loop ==========================================================================
  if (nstages > 0) {
    load_stage();
    process_stage(0);
    if (nstages > 1) load_stage();
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const int cur = s & 1;
      const bool more = s + 1 < nstages;
      if (!upper) {
        mma_stage(cur);
        if (more) { process_stage(cur ^ 1); if (s + 2 < nstages) load_stage(); }
      } else {
        if (more) { process_stage(cur ^ 1); if (s + 2 < nstages) load_stage(); }
        mma_stage(cur);
      }
      __syncthreads();
    }
  }

  // =================== epilogue ===========================================================================
  // lane owns pixel (lane&31) of each 32-pixel tile; register r = 4q+rr holds channel 32*wv_n + 8q + 4h + rr.
  const bool to_partial = p.ksplits > 1;
  const bool has_bias = (split == 0) && (p.mu_b != nullptr);  // uniform
  float* bias_lds = (float*)smem;  // [0..63] mean part, [64..127] perturbation part (stage buffers are dead now)
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
    __syncthreads();
  }
  const bool has_aff = (p.ep_scale != nullptr) || (p.ep_shift != nullptr);
  float* aff_lds = bias_lds + 2 * BN;
  if (has_aff) {
    if (tid < BN) {
      const int col = ntile * BN + tid;
      const int gcol = group * p.Ng + (col < p.Ng ? col : 0);
      aff_lds[tid] = p.ep_scale ? p.ep_scale[gcol] : 1.f;
      aff_lds[BN + tid] = p.ep_shift ? p.ep_shift[gcol] : 0.f;
    }
    __syncthreads();
  }
  const int colbase = ntile * BN + wv_n * 32;  // within the group
  if (colbase < p.Ng) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int mo = mtile * BM + wv_m * 64 + mi * 32 + l31;
      if (mo >= p.M) continue;
      const long long orow = (long long)mo * p.N + group * p.Ng;
      // hashed s_out word when the 32-channel run is one word of the output tensor
      const unsigned long long o0 = (unsigned long long)(orow + colbase);
      const bool word_fast = (KIND == 1) && !p.sign_out && ((o0 & 31ull) == 0) && (colbase + 32 <= p.Ng);
      uint32_t wout = 0;
      if (word_fast) wout = btx_sign_word((uint32_t)(o0 >> 5), rl.kout_a, rl.kout_b);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cl = wv_n * 32 + 8 * q + 4 * h;  // channel within the n-tile
        const int c0 = ntile * BN + cl;
        if (c0 >= p.Ng) continue;
        float v[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int col = c0 + rr;
          float val = accm[mi][4 * q + rr];
          if (has_bias) val += bias_lds[cl + rr];
          if constexpr (KIND == 1) {
            float dl = accd[mi][4 * q + rr];
            if (has_bias) dl += bias_lds[BN + cl + rr];
            uint32_t flip = 0;
            if (col < p.Ng) {
              if (p.sign_out) {
                flip = (p.sign_out[orow + col] < 0) ? 0x80000000u : 0u;
              } else if (word_fast) {
                const int bp = ((rr & 1) ? 31 : 15) - 4 * q - 2 * h - (rr >> 1);
                flip = (wout << (31 - bp)) & 0x80000000u;
              } else {
                const unsigned long long io = (unsigned long long)(orow + col);
                const uint32_t w1 = btx_sign_word((uint32_t)(io >> 5), rl.kout_a, rl.kout_b);
                flip = (w1 << (31 - btx_sign_bitpos((uint32_t)io & 31u))) & 0x80000000u;
              }
            }
            val += u2f(f2u(dl) ^ flip);
          }
          v[rr] = val;
        }
        const bool vec = (c0 + 3 < p.Ng) && (((orow + c0) & 3) == 0);
        if (to_partial) {
          float* dst = p.partial + (long long)split * p.M * p.N + orow + c0;
          if (vec) {
            *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
          } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) if (c0 + rr < p.Ng) dst[rr] = v[rr];
          }
        } else {
          apply_epilogue4<ACT>(v, p, has_aff ? aff_lds : nullptr, cl, orow + c0, p.Ng - c0, vec);
          ACT* dst = (ACT*)p.out + orow + c0;
          if (vec) {
            if constexpr (sizeof(ACT) == 4) {
              *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
            } else {
              f32x4 fv = {v[0], v[1], v[2], v[3]};
              *(bf16x4*)dst = __builtin_convertvector(fv, bf16x4);
            }
          } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) if (c0 + rr < p.Ng) dst[rr] = (ACT)v[rr];
          }
        }
      }
    }
  }
}

// Second pass of split-K: out[i] = epilogue(sum_s partial[s][i]), converted to the output dtype.
template <typename ACT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, ACT* __restrict__ out,
                                                           long long total, int ksplits, int N,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const ACT* __restrict__ res, int relu,
                                                           long long lane_partial, long long lane_out, long long lane_res) {
  // MC sample lanes: blockIdx.y = lane; byte strides between the lanes' partial sums / outputs / residuals
  partial = (const float*)((const unsigned char*)partial + (long long)blockIdx.y * lane_partial);
  out = (ACT*)((unsigned char*)out + (long long)blockIdx.y * lane_out);
  if (res) res = (const ACT*)((const unsigned char*)res + (long long)blockIdx.y * lane_res);
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += stride) {
    const int nv = (int)((total - i) < 4 ? (total - i) : 4);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (nv == 4) {
      f32x4 acc = *(const f32x4*)(partial + i);
      for (int s = 1; s < ksplits; ++s) acc += *(const f32x4*)(partial + (long long)s * total + i);
      a[0] = acc[0]; a[1] = acc[1]; a[2] = acc[2]; a[3] = acc[3];
    } else {
      for (int r = 0; r < nv; ++r)
        for (int s = 0; s < ksplits; ++s) a[r] += partial[(long long)s * total + i + r];
    }
    for (int r = 0; r < nv; ++r) {
      const int col = (int)((i + r) % N);
      float v = a[r];
      v = __builtin_fmaf(v, scale ? scale[col] : 1.f, shift ? shift[col] : 0.f);
      if (res) v += (float)res[i + r];
      if (relu) v = v > 0.f ? v : 0.f;
      a[r] = v;
    }
    if (nv == 4) {
      if constexpr (sizeof(ACT) == 4) *(f32x4*)(out + i) = (f32x4){a[0], a[1], a[2], a[3]};
      else *(bf16x4*)(out + i) = __builtin_convertvector((f32x4){a[0], a[1], a[2], a[3]}, bf16x4);
    } else {
      for (int r = 0; r < nv; ++r) out[i + r] = (ACT)a[r];
    }
  }
}

// launcher used by btx_api.hip; defined per precision in btx_contract_{f32,bf16}.hip
int launch_contract_f32(int kind, int act_bf16, bool gen, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_bf16(int kind, int act_bf16, bool gen, const ContractParams& p, int nwg, hipStream_t st);
// LDS-DMA pipeline variants (btx_contract_dma.h): activation dtype == contraction dtype, granule-aligned shapes
int launch_contract_dma_f32(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_dma_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_patch_f32(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_stem_f32(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_stem_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_stem_pool_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st);
struct PresampleBatch;
int launch_presample_batch_f32(const PresampleBatch& b, hipStream_t st);
int launch_presample_batch_bf16(const PresampleBatch& b, hipStream_t st);
int launch_contract_patch_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st);
// split-bf16 (BTX_PREC_BF16X3, btx_x3.hip)
int launch_contract_patch_x3(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_stem_x3(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_dma_x3(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_gemm8_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st);  // btx_contract_gemm8.h
int launch_contract_gemm8_f32(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_gemm8_x3(int kind, const ContractParams& p, int nwg, hipStream_t st);
// pointwise Flipout-GEMM with the n-tile loop inside the workgroup (btx_contract_pw.h)
int launch_contract_pw_f32(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_pw_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_contract_pw_x3(int kind, const ContractParams& p, int nwg, hipStream_t st);
int launch_presample_batch_x3(const PresampleBatch& b, hipStream_t st);

template <int PREC>
static int launch_contract_impl(int kind, int act_bf16, bool gen, const ContractParams& p, int nwg, hipStream_t st) {
#define BTX_LAUNCH(ACT, KIND, GENF)                                                                              \
  do {                                                                                                            \
    auto kfn = contract_kernel<PREC, ACT, KIND, GENF>;                                                            \
    static bool attr_done = false;                                                                                \
    if (!attr_done) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
      if (e != hipSuccess) return (int)e;                                                                         \
      attr_done = true;                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(NTHREADS), LDS_BYTES, st, p);                                         \
  } while (0)
  if (!act_bf16) {
    if (kind == 0) { if (gen) BTX_LAUNCH(float, 0, true); else BTX_LAUNCH(float, 0, false); }
    else           { if (gen) BTX_LAUNCH(float, 1, true); else BTX_LAUNCH(float, 1, false); }
  } else {
    if (kind == 0) { if (gen) BTX_LAUNCH(__bf16, 0, true); else BTX_LAUNCH(__bf16, 0, false); }
    else           { if (gen) BTX_LAUNCH(__bf16, 1, true); else BTX_LAUNCH(__bf16, 1, false); }
  }
#undef BTX_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace btx
