// btx_contract_dma.h — the per-tap LDS-DMA pipeline of the sample-and-contract kernel (gfx950): every aligned shape the
// patch / stem kernels do not take (strided and dilated convolutions, 1x1, 1-D / 3-D, transposed, Linear).
//
// Same math, LDS fragment image and MFMA conventions as btx_contract.h::contract_kernel, but nothing is staged through
// registers:
//
//   * workgroup tile 64*NW pixels x 64 channels (NW = 4: two workgroups per CU; 8: one), a wave owns 64 pixels x 64
//     channels (2x2 MFMA 32x32 tiles; Flipout: 128 accumulator registers);
//   * a K-stage (32 bf16 / 16 f32 consecutive k = 64 bytes of one pixel) lies inside ONE filter tap (needs
//     C/groups % stage == 0).  One buffer_load ... lds moves 16 pixels x 64 B: the four 16-byte granules of a
//     pixel sit in ADJACENT LANES of one instruction (measured on MI355X, tools/ubench/dma_pattern.hip: 0.90
//     cycles per 16-B request per CU, vs 1.84 when the same bytes are split over four instructions and 4.9 for one
//     granule per cache line).  The LDS image is pixel-major with an XOR swizzle, slot = granule ^ ((pixel>>2)&3),
//     applied on the SOURCE side (the DMA destination is lane-linear), which makes every 16-lane group of the MFMA
//     fragment reads (16 pixels, one granule) hit 16 distinct 16-byte bank slots; out-of-image taps and tile tails are
//     zero-filled by the buffer descriptor (per-axis tap-validity bitmasks computed once per workgroup);
//   * the weights arrive as MFMA-ready tiles, sampled once per launch (btx_presample.h), by the same DMA path into a
//     ring of WD slots, WD-1 stages ahead;
//   * every VMEM op in the loop is an LDS-DMA, so hipcc inserts no vmcnt of its own: one counted
//     `s_waitcnt vmcnt(N)` + lgkmcnt(0) + raw s_barrier per stage, never vmcnt(0) in steady state
//     (cdna_hip_programming.md §5 "Pipelining across barriers").
//
// Ring bookkeeping, iteration s (stage s is being multiplied):
//   issue   W(s+WD-1) -> weight slot (s+WD-1)%WD, acts/sign(s+2) -> slot (s+2)%3 — both slots were last read during
//           iteration s-1, behind the barrier that ended it
//   M(s)    acts/sign slot s%3, weight tile s%WD
//   end     vmcnt(n), n = the operations this wave issued during THIS iteration: everything older has landed, i.e.
//           acts(s+1) and W(s+1)
#pragma once
#include <type_traits>
#include "btx_contract.h"
#include "btx_epilogue.h"
#include "btx_presample.h"
#include "btx_mma.h"

#ifndef BTX_DMA_RPRE
#define BTX_DMA_RPRE 1  // residual rows requested in front of the store side's first stage (btx_epilogue.h, RES_PRE): +2 % on cfg5
#endif

namespace btx {

constexpr int DBM = 512;                    // pixels per workgroup tile, 8-wave blocks (one per CU)
constexpr int DMA_D = 3;                    // activation + sign ring depth
constexpr int DW_STAGE = 2 * NG * BN * 16;  // 8192 : mu tile at +0, delta tile at +4096
// LDS map of a block of NW waves (tile = 64*NW pixels): DMA_D activation stages, DMA_D sign stages, WD weight tiles
// (pre-sampled, btx_presample.h); the epilogue staging (btx_epilogue.h) reuses the same bytes.  4-wave blocks
// (256 pixels) fit 80 KiB, so two share a CU: one block's prologue/epilogue overlaps the other's K loop — what layers
// with few K stages (stems, 1x1 convs) need.
template <int NW>
struct DmaLds {
  static constexpr int TP = 64 * NW;               // pixels per tile
  static constexpr int WD = (NW == 4) ? 3 : 4;     // weight-tile ring depth; W(s + WD - 1) is fetched during stage s
  static constexpr int A_STAGE = NG * TP * 16;     // 32768 | 16384
  static constexpr int S_STAGE = TP * 4;           // one sign word per (pixel, stage)
  static constexpr int A_OFF = 0;
  static constexpr int S_OFF = A_OFF + DMA_D * A_STAGE;
  static constexpr int W_OFF = S_OFF + DMA_D * S_STAGE;
  static constexpr int MAIN = W_OFF + WD * DW_STAGE;          // 137216 | 76800
  static constexpr int EP = NW * PT_EP_WAVE + 1024;           // 140288 | 70656
  static constexpr int BYTES = MAIN > EP ? MAIN : EP;
};
static_assert(DmaLds<4>::BYTES <= 81920 && DmaLds<8>::BYTES <= 163840, "LDS budget");

constexpr uint32_t DMA_OOB = 0xfffffff0u;  // byte offset beyond every descriptor: the hardware returns zeros

// buffer_load_dwordx4 ... offen lds: descriptor base in SGPRs, one 32-bit byte offset per lane (out-of-range lanes are
// zero-filled by the hardware: that is the conv padding and every tile tail), LDS destination = wave-uniform base +
// lane*16.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off,
                                           0, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a constant).  A small decision tree over the
// counts the pipelines actually produce; a value in between waits for the next smaller one, which is only more
// conservative (fewer operations left in flight).
#define BTX_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void wait_vmcnt(int n) {  // n is wave-uniform
  if (n >= 4) {
    if (n >= 8) {
      if (n >= 16) BTX_VM(16); else if (n >= 12) BTX_VM(12); else BTX_VM(8);
    } else {
      if (n >= 6) BTX_VM(6); else if (n == 5) BTX_VM(5); else BTX_VM(4);
    }
  } else {
    if (n >= 2) {
      if (n == 3) BTX_VM(3); else BTX_VM(2);
    } else {
      if (n == 1) BTX_VM(1); else BTX_VM(0);
    }
  }
}
#undef BTX_VM

// PW: the pointwise form (Linear, 1x1 convolutions at stride 1 without padding, hashed or explicit signs on whole words: the
// K = 64 / N = 64 convolutions of every ResNet50 bottleneck and the classifier heads).  Output pixel m IS input pixel m and the
// one tap is always valid, so the geometry decode, the tap masks, the (kd, kh, kw) walk and the transposed gather rule are not
// even compiled: the generic instantiation keeps ~40 launch parameters alive across its K loop, which hipcc spills into VGPR
// lanes (733 v_readlane + 180 v_writelane in the bf16 Flipout kernel's ISA); this one reads seven.  Same arithmetic, same
// order: bit-identical to the generic form (tests/test_gpu_contract.py).
// TR: transposed launches (the gather rule, its per-tap cache and the parity-major order) are an instantiation of their own — the
// plain one keeps none of that state alive across its K loop (SGPR spills 280 -> see DESIGN.md section 5)
template <int PREC, int KIND, int NW, bool PW = false, bool TR = false>
__global__ __launch_bounds__(64 * NW, 2) void contract_dma_kernel(const ContractParams) {
  BTX_SECTION_PARAMS(p, logical);  // prologue + K loop; the store side has its own view (btx_contract.h)
  using LD = DmaLds<NW>;
  constexpr int TP = LD::TP, WD = LD::WD, DA_STAGE = LD::A_STAGE, DS_STAGE = LD::S_STAGE, DA_OFF = LD::A_OFF,
                DS_OFF = LD::S_OFF, DW_OFF = LD::W_OFF;
#ifdef BTX_PT_TRACE
  const uint32_t tr_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
  uint32_t tr_ab = 0, tr_bc = 0, tr_cd = 0, tr_t1 = 0, tr_t2 = 0, tr_tg = 0;
#endif
  const RngLive rl = rng_live<KIND>(p);
  using ACT = typename std::conditional<PREC == 1, __bf16, float>::type;
  constexpr int G = (PREC == 1) ? 8 : 4;
  constexpr int BK = NG * G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool upper = (NW == 8) ? (wave >= 4) : ((wave & 1) != 0);  // these waves issue their DMAs after the MFMA block

  uint32_t u_mtile, u_rem, u_split, u_ntile, u_group, u_t;
  if (p.wg_order) fdivmod((uint32_t)logical, p.fd_mtiles, (uint32_t)p.mtiles, u_rem, u_mtile);  // weight-major (btx_api.hip)
  else fdivmod((uint32_t)logical, p.fd_inner, (uint32_t)(p.ntiles * p.groups * p.ksplits), u_mtile, u_rem);
  fdivmod(u_rem, p.fd_ksplits, (uint32_t)p.ksplits, u_t, u_split);
  fdivmod(u_t, p.fd_ntiles, (uint32_t)p.ntiles, u_group, u_ntile);
  const int mtile = (int)u_mtile, split = (int)u_split, ntile = (int)u_ntile, group = (int)u_group;

  const int k_begin = split * p.kper;
  const int k_end = min(p.K, k_begin + p.kper);
  // parity-major transposed launches (ContractParams.par_major; ksplits == 1, KD == 1, Cg % BK == 0): every pixel of the tile has
  // the parity class (oh & 1, ow & 1) of the tile, and tap (kh, kw) reaches such a pixel only if oh + ph - kh*dh and ow + pw - kw*dw
  // are multiples of the stride 2 — the stages of the other taps are neither fetched nor multiplied.
  uint32_t tapmask = 0xffffffffu;  // bit t: tap t = kh * KW + kw is walked
  int spt = 1;                     // K stages per tap
  bool par = false;
  int tile_m0 = mtile * TP;        // first (logical) pixel of the tile
  int par_nvalid = 0;              // parity-major: pixels of the tile that exist (the class's last tile is padded)
  if constexpr (!PW && TR) {
    if (p.par_major) {
      // tile t: class t & 3, tile t >> 2 of that class (classes alternate: every XCD's range of tiles holds all four, the classes'
      // unequal stage counts even out); logical pixel = class * par_Mqp + q, q < par_Mq the class's raster index
      par = true;
      spt = p.Cg / BK;
      const int cls = mtile & 3, q0 = (mtile >> 2) * TP;
      tile_m0 = cls * p.par_Mqp + q0;
      par_nvalid = min(TP, p.par_Mq - q0);
      const int pi = cls >> 1, pj = cls & 1;
      tapmask = 0u;
      for (int kh = 0; kh < p.KH; ++kh)
        for (int kw = 0; kw < p.KW; ++kw)
          if ((((pi + p.ph + kh * p.dh) | (pj + p.pw + kw * p.dw)) & 1) == 0) tapmask |= 1u << (kh * p.KW + kw);
    }
  }
  const int nstages = par ? __builtin_popcount(tapmask) * spt : (k_end - k_begin + BK - 1) / BK;

  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);
  // ---- weight loader role: the stage's tile is 4 rows of mu (+ 4 rows of delta, Flipout) of 1 KiB in the pre-sampled
  //      buffer.  8 waves: wave w fetches mu row w (w < 4) or delta row w-4;  4 waves: mu row w and delta row w.
  const bool w_mu = (NW == 4) || (wave < 4);
  const bool w_dl = (KIND == 1) && ((NW == 4) || (wave >= 4));
  const int w_nops = (w_mu ? 1 : 0) + (w_dl ? 1 : 0);
  const uint32_t w_base = (uint32_t)(group * p.ntiles + ntile) * (uint32_t)(p.K / G) * 1024u + (uint32_t)lane * 16u +
                          (uint32_t)(wave & 3) * 1024u + (uint32_t)(k_begin / G) * 1024u;
  const int w_lds = DW_OFF + (wave & 3) * 1024;

  int w_tap = par ? __builtin_ctz(tapmask | 0x80000000u) : 0, w_cs = 0;  // parity-major: the live tap / stage inside it of the next tile
  auto issue_w = [&](int st) __attribute__((always_inline)) {  // called for st = 0, 1, 2, ... in order
    uint32_t go = w_base + (uint32_t)st * (uint32_t)(BK / G) * 1024u;
    if constexpr (!PW) {
      if (par) {
        go = w_base + (uint32_t)(w_tap * spt + w_cs) * (uint32_t)(BK / G) * 1024u;
        if (++w_cs == spt) {
          w_cs = 0;
          do { ++w_tap; } while (w_tap < 32 && !((tapmask >> w_tap) & 1u));
        }
      }
    }
    unsigned char* ld = smem + w_lds + (st % WD) * DW_STAGE;
    if (w_mu) dma16(wt_rsrc, go, ld);
    if (w_dl) dma16(wt_rsrc, go + p.wt_delta_off, ld + 4096);
  };
  // the first weight tiles need nothing but the tile indices: their latency hides behind the pixel geometry below
  for (int s = 0; s < WD - 1 && s < nstages; ++s) issue_w(s);

  // ---- loader role.  DMA instruction q (0..3) of wave w moves pixels 64w + 16q + (lane>>2), granule slot lane&3;
  //      each lane therefore keeps the geometry of FOUR pixels.  All offsets are 32-bit (host guarantees < 2^31
  //      elements); base_off = element offset of (pixel, tap 0, channel 0 of the group), wrap-around arithmetic.
  const int g_lane = (lane & 3) ^ ((lane >> 4) & 3);  // granule this lane fetches (source-side swizzle)
  int pb_d[4], pb_h[4], pb_w[4], pb_n[4];
  uint32_t pb_off[4];
  bool pb_ok[4];
  auto decode = [&](int mm_, int& bd_, int& bh_, int& bw_, int& nbase_, uint32_t& off_) __attribute__((always_inline)) {
    uint32_t t, t2, uow, uoh, uod, unb;
    if (par) {  // logical pixel -> (class, image, oh / 2, ow / 2)   (Do == 1)
      uint32_t cls, q, a, b;
      fdivmod((uint32_t)mm_, p.fd_par_Mqp, (uint32_t)p.par_Mqp, cls, q);
      fdivmod(q, p.fd_par_Wh, (uint32_t)p.par_Wh, t, b);
      fdivmod(t, p.fd_par_Hh, (uint32_t)p.par_Hh, unb, a);
      uoh = 2u * a + (cls >> 1); uow = 2u * b + (cls & 1u); uod = 0u;
    } else {
    fdivmod((uint32_t)mm_, p.fd_Wo, (uint32_t)p.Wo, t, uow);
    fdivmod(t, p.fd_Ho, (uint32_t)p.Ho, t2, uoh);
    fdivmod(t2, p.fd_Do, (uint32_t)p.Do, unb, uod);
    }
    const int ow = (int)uow, oh = (int)uoh, od = (int)uod, nb = (int)unb;
    nbase_ = nb * p.D;
    if (!TR) {
      bd_ = od * p.sd - p.pd; bh_ = oh * p.sh - p.ph; bw_ = ow * p.sw - p.pw;
    } else {
      bd_ = od + p.pd; bh_ = oh + p.ph; bw_ = ow + p.pw;
    }
    off_ = (uint32_t)(((nbase_ + bd_) * p.H + bh_) * p.W + bw_) * (uint32_t)p.C + (uint32_t)(group * p.Cg);
  };
  // Pointwise contractions (Linear, 1x1 convolutions at stride 1: 37 of the 53 convolutions of a ResNet50): output pixel m
  // IS input pixel m — no (n, d, h, w) decode, every tap (there is one) valid.  The decode below was 3.4k of the 7.3k
  // prologue cycles of a block whose K loop (K = 64) takes 2.3k (phase timers, round 3).
  const bool pointwise = PW || (p.pointwise != 0);  // wave-uniform
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pl_ = wave * 64 + q * 16 + (lane >> 2);
    const int mq = tile_m0 + pl_;
    pb_ok[q] = par ? (pl_ < par_nvalid) : (mq < p.M);
    if (PW || pointwise) {
      pb_d[q] = pb_h[q] = pb_w[q] = pb_n[q] = 0;
      pb_off[q] = (uint32_t)(pb_ok[q] ? mq : 0) * (uint32_t)p.C + (uint32_t)(group * p.Cg);
    } else {
      if constexpr (!PW) decode(pb_ok[q] ? mq : 0, pb_d[q], pb_h[q], pb_w[q], pb_n[q], pb_off[q]);
    }
  }
  // tap validity, one bitmask per axis and pixel (bit k: tap k of that axis reads inside the input), computed once per
  // workgroup with KD + KH + KW iterations; a tap is valid iff its three bits are set
  const bool use_mask = !PW && (p.KD <= 32) && (p.KH <= 32) && (p.KW <= 32) && !TR;  // uniform
  uint32_t md[4] = {0u, 0u, 0u, 0u}, mh[4] = {0u, 0u, 0u, 0u}, mw[4] = {0u, 0u, 0u, 0u};
  if constexpr (PW) {
  } else if (pointwise) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { md[q] = pb_ok[q] ? 1u : 0u; mh[q] = 1u; mw[q] = 1u; }
  } else if (use_mask) {
    for (int k = 0; k < p.KD; ++k) {
#pragma unroll
      for (int q = 0; q < 4; ++q) md[q] |= (pb_ok[q] && (unsigned)(pb_d[q] + k * p.dd) < (unsigned)p.D) ? (1u << k) : 0u;
    }
    for (int k = 0; k < p.KH; ++k) {
#pragma unroll
      for (int q = 0; q < 4; ++q) mh[q] |= ((unsigned)(pb_h[q] + k * p.dh) < (unsigned)p.H) ? (1u << k) : 0u;
    }
    for (int k = 0; k < p.KW; ++k) {
#pragma unroll
      for (int q = 0; q < 4; ++q) mw[q] |= ((unsigned)(pb_w[q] + k * p.dw) < (unsigned)p.W) ? (1u << k) : 0u;
    }
  }
  // byte offset of this lane's granule inside its pixel, folded into the per-pixel base
  uint32_t pb_boff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) pb_boff[q] = (pb_off[q] + (uint32_t)(G * g_lane)) * (uint32_t)sizeof(ACT);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  uint32_t sg_off;  // this thread's own pixel (tid): only the sign word needs it
  int sg_bd = 0, sg_bh = 0, sg_bw = 0, sg_nb = 0;
  {
    const int m = tile_m0 + tid;
    const bool m_ok = par ? (tid < par_nvalid) : (m < p.M);
    if (PW || pointwise) sg_off = (uint32_t)(m_ok ? m : 0) * (uint32_t)p.C + (uint32_t)(group * p.Cg);
    else if constexpr (!PW) decode(m_ok ? m : 0, sg_bd, sg_bh, sg_bw, sg_nb, sg_off);
  }
  // transposed launches: the gather rule (three divisions, eight compares per pixel) depends on the TAP, not on the channel block —
  // evaluated once per tap and kept for the tap's Cg / BK stages
  int tp_tap = -1;
  bool tp_ok[4] = {false, false, false, false};
  uint32_t tp_bo[4] = {0u, 0u, 0u, 0u}, tp_sg = 0u;

  // wave-uniform K walk: channel offset inside the tap and the tap itself
  int s_c, s_kd, s_kh, s_kw, s_tap;
  if constexpr (PW) {  // one tap: the walk is the channel offset alone
    s_c = k_begin; s_kd = s_kh = s_kw = s_tap = 0;
  } else {
    uint32_t tap, c0, t2, kw0, kd0, kh0;
    fdivmod((uint32_t)k_begin, p.fd_Cg, (uint32_t)p.Cg, tap, c0);
    fdivmod(tap, p.fd_KW, (uint32_t)p.KW, t2, kw0);
    fdivmod(t2, p.fd_KH, (uint32_t)p.KH, kd0, kh0);
    s_tap = (int)tap; s_c = (int)c0; s_kw = (int)kw0; s_kh = (int)kh0; s_kd = (int)kd0;
    if (par) {  // the walk starts at the first live tap
      s_tap = __builtin_ctz(tapmask | 0x80000000u);
      fdivmod((uint32_t)s_tap, p.fd_KW, (uint32_t)p.KW, kh0, kw0);
      s_kh = (int)kh0; s_kw = (int)kw0; s_kd = 0; s_c = 0;
    }
  }

  int a_slot_issue = 0;  // ring slot the next issue_acts() fills

  // =================== issue: all L2 -> LDS traffic of one stage (called for stages 0,1,2,... in order) ======
  auto issue_acts = [&]() __attribute__((always_inline)) {  // stages in order: the K walk advances by one stage per call
    // activations: one tap for the whole stage; tap_off is wave-uniform
    uint32_t tap_off = 0;
    if constexpr (PW) tap_off = (uint32_t)s_c;
    else if (!TR)
      tap_off = (uint32_t)(((s_kd * p.dd) * p.H + s_kh * p.dh) * p.W + s_kw * p.dw) * (uint32_t)p.C + (uint32_t)s_c;
    const uint32_t tap_boff = tap_off * (uint32_t)sizeof(ACT);
    unsigned char* as = smem + DA_OFF + a_slot_issue * DA_STAGE + wave * 4096;
    if constexpr (!PW) {
      if (TR && tp_tap != s_tap) {  // wave-uniform: a new tap
        tp_tap = s_tap;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int td = pb_d[q] - s_kd * p.dd, th = pb_h[q] - s_kh * p.dh, tw = pb_w[q] - s_kw * p.dw;
          // (multiply-shift division: negative numerators come out wrong and are rejected by the sign tests below)
          const int id = (int)fdiv((uint32_t)td, p.fd_sd), ih = (int)fdiv((uint32_t)th, p.fd_sh), iw = (int)fdiv((uint32_t)tw, p.fd_sw);
          tp_ok[q] = pb_ok[q] && td >= 0 && th >= 0 && tw >= 0 && (id * p.sd == td) && (ih * p.sh == th) &&
                     (iw * p.sw == tw) && id < p.D && ih < p.H && iw < p.W;
          tp_bo[q] = ((uint32_t)(((pb_n[q] + id) * p.H + ih) * p.W + iw) * (uint32_t)p.C +
                      (uint32_t)(group * p.Cg + G * g_lane)) * (uint32_t)sizeof(ACT);
        }
        if constexpr (KIND == 1) {
          // (a pixel whose tap falls outside — negative numerator — carries zero activations: its word is irrelevant)
          const int id = (int)fdiv((uint32_t)(sg_bd - s_kd * p.dd), p.fd_sd), ih = (int)fdiv((uint32_t)(sg_bh - s_kh * p.dh), p.fd_sh),
                    iw = (int)fdiv((uint32_t)(sg_bw - s_kw * p.dw), p.fd_sw);
          tp_sg = (uint32_t)(((sg_nb + id) * p.H + ih) * p.W + iw) * (uint32_t)p.C + (uint32_t)(group * p.Cg);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bool ok;
      uint32_t bo;
      if constexpr (PW) {
        ok = pb_ok[q];
        bo = pb_boff[q] + tap_boff;
      } else if (use_mask) {
        ok = ((md[q] >> s_kd) & (mh[q] >> s_kh) & (mw[q] >> s_kw) & 1u) != 0u;
        bo = pb_boff[q] + tap_boff;
      } else if (!TR) {
        const int id = pb_d[q] + s_kd * p.dd, ih = pb_h[q] + s_kh * p.dh, iw = pb_w[q] + s_kw * p.dw;
        ok = pb_ok[q] && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        bo = pb_boff[q] + tap_boff;
      } else {
        ok = tp_ok[q];
        bo = tp_bo[q] + (uint32_t)s_c * (uint32_t)sizeof(ACT);
      }
      dma16(x_rsrc, ok ? bo : DMA_OOB, as + q * 1024);
    }
    if constexpr (KIND == 1) {
      // one hashed word covers the 32 (bf16) / 16 (f32) channels of pixel `tid`'s stage.  (In the padding the
      // activations are zero, so the word is irrelevant there.  Transposed: recompute the input offset.)
      uint32_t off = sg_off + tap_off;
      if (!PW && TR) off = tp_sg + (uint32_t)s_c;
      // sign layout: element pair e>>1 sits at bit 15-(e>>1) (even e) / 31-(e>>1) (odd e) of its word, so a stage that
      // starts at element offset e0 inside the word needs the word shifted left by e0>>1 within each 16-bit half
      const uint32_t n_in = p.x_bytes / (uint32_t)sizeof(ACT);
      uint32_t w = p.sign_in ? sign_word_explicit(p.sign_in, off, n_in) : btx_sign_word(off >> 5, rl.kin_a, rl.kin_b);
      if (!PW && p.sign_unaligned) {  // uniform: the stage may run into the next word (row-fused stems)
        const uint32_t w1 = p.sign_in ? sign_word_explicit(p.sign_in, off + 32u, n_in)
                                      : btx_sign_word((off >> 5) + 1u, rl.kin_a, rl.kin_b);
        const uint32_t k = (off & 31u) >> 1;
        const uint32_t lo = ((w & 0xffffu) << 16) | (w1 & 0xffffu);
        const uint32_t hi = (w & 0xffff0000u) | (w1 >> 16);
        w = ((lo << k) >> 16) | ((hi << k) & 0xffff0000u);
      } else if constexpr (G == 4) {
        w <<= 8 * ((off >> 4) & 1);
      }
      *(uint32_t*)(smem + DS_OFF + a_slot_issue * DS_STAGE + tid * 4) = w;
    }
    // advance the K walk by one stage
    s_c += BK;
    if (!PW && s_c >= p.Cg) {
      s_c = 0;
      do {  // (parity-major: on to the next LIVE tap)
        ++s_tap;
        if (++s_kw == p.KW) { s_kw = 0; if (++s_kh == p.KH) { s_kh = 0; ++s_kd; } }
      } while (par && s_tap < 32 && !((tapmask >> s_tap) & 1u));
    }
    a_slot_issue = (a_slot_issue == DMA_D - 1) ? 0 : a_slot_issue + 1;
  };

  // =================== M: wave = pixels [64*wave, +64) x all 64 channels ===================================
  f32x16 accm[2][2], accd[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }

  auto mma_stage = [&](int st, int a_slot) __attribute__((always_inline)) {
    const unsigned char* as = smem + DA_OFF + a_slot * DA_STAGE;
    const unsigned char* ss = smem + DS_OFF + a_slot * DS_STAGE;
    const unsigned char* ws = smem + DW_OFF + (st % WD) * DW_STAGE;
    StageFrag f;  // every fragment read of the stage is issued up front; the MFMAs follow as the data arrives (btx_mma.h)
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      const int row = 2 * kk + h;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        f.a[kk][mi] = *(const u32x4*)(as + (wave * 64 + mi * 32 + l31) * 64 + ((row ^ ((l31 >> 2) & 3)) * 16));
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) f.wm[kk][ni] = *(const u32x4*)(ws + (row * BN + ni * 32 + l31) * 16);
    }
    if constexpr (KIND == 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) f.sw[mi] = *(const uint32_t*)(ss + (wave * 64 + mi * 32 + l31) * 4);
    }
    DeltaFrag dfrag;
    load_delta<KIND>(dfrag, ws, l31, h);
    stage_mma<PREC, KIND>(f, dfrag, accm, accd, l31, h);
  };

  // =================== main loop ==========================================================================
  // Iteration s: every wave issues W(s+WD-1) and acts(s+2), multiplies stage s, waits until everything it issued BEFORE
  // this iteration has landed (vmcnt retires in order: at most this iteration's operations stay in flight) and meets
  // the others at one barrier.  acts(s+1) (issued one iteration ago) and W(s+1) (two iterations ago) are then visible.
  // The DMA instructions block at issue while the memory pipeline is full: waves 0-3 issue at the top of the
  // iteration, waves 4-7 after their MFMA block, so that of the two waves sharing a SIMD one is free to compute while
  // the other is stuck issuing.
#ifdef BTX_PT_TRACE
  tr_tg = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
  if (nstages > 0) {
    issue_acts();
    if (nstages > 1) issue_acts();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int a_slot = 0;
#ifdef BTX_PT_TRACE
    tr_t1 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    for (int s = 0; s < nstages; ++s) {
#ifdef BTX_PT_TRACE
      const uint32_t tA = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
      const bool acts_issued = s + 2 < nstages;
      const bool w_issued = s + WD - 1 < nstages;
      if (!upper) {
        if (w_issued) issue_w(s + WD - 1);
        if (acts_issued) issue_acts();
      }
      mma_stage(s, a_slot);
      if (upper) {
        if (w_issued) issue_w(s + WD - 1);
        if (acts_issued) issue_acts();
      }
#ifdef BTX_PT_TRACE
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t tB = (uint32_t)__builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
      wait_vmcnt((acts_issued ? 4 : 0) + (w_issued ? w_nops : 0));
#ifdef BTX_PT_TRACE
      const uint32_t tC = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef BTX_PT_TRACE
      const uint32_t tD = (uint32_t)__builtin_amdgcn_s_memtime();
      tr_ab += tB - tA; tr_bc += tC - tB; tr_cd += tD - tC;
#endif
      a_slot = (a_slot == DMA_D - 1) ? 0 : a_slot + 1;
    }
  }

  // =================== epilogue (btx_epilogue.h) ============================================================
  {
    BTX_SECTION_PARAMS(pe, logical2);
    const uint32_t m0 = (uint32_t)tile_m0;
    const int nvalid = par ? par_nvalid : min(TP, pe.M - (int)m0);
#ifdef BTX_PT_TRACE
    tr_t2 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
#ifdef BTX_EP_TRACE
    uint32_t ep_t[2] = {0, 0};
    staged_epilogue<KIND, NW, BTX_DMA_RPRE != 0>(pe, rl, accm, accd, smem, tid, wave, lane, ntile, group, split, m0, nvalid, ep_t);
    tr_ab = ep_t[0] - tr_t2; tr_bc = ep_t[1] - ep_t[0];
#else
    bool par_ep = false;
    if constexpr (!PW && TR) par_ep = pe.par_major != 0;
    if (par_ep) {
      if constexpr (!PW && TR) {
        const PixParity pm = {pe, m0, nvalid};
        staged_epilogue_pm<KIND, NW, PixParity, false>(pe, rl, accm, accd, smem, tid, wave, lane, ntile, group, split, pm);
      }
    } else {
      staged_epilogue<KIND, NW, BTX_DMA_RPRE != 0>(pe, rl, accm, accd, smem, tid, wave, lane, ntile, group, split, m0, nvalid);
    }
#endif
  }
#ifdef BTX_PT_TRACE
  if (p.trace) {
#ifdef BTX_EP_TRACE
    const uint32_t tr_tx = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t tr_t3 = (uint32_t)__builtin_amdgcn_s_memtime();
#ifdef BTX_EP_TRACE
    tr_cd = tr_t3 - tr_tx;
#endif
    if (lane == 0) {
      uint32_t* tr = (uint32_t*)p.trace + (size_t)(blockIdx.x * NW + wave) * 8;
      tr[0] = tr_t1 - tr_t0; tr[1] = tr_ab; tr[2] = tr_bc; tr[3] = tr_cd; tr[4] = tr_t3 - tr_t2; tr[5] = tr_t3 - tr_t0;
      tr[6] = tr_t0; tr[7] = tr_tg - tr_t0;  // geometry part of the prologue
    }
  }
#endif
}

template <int PREC>
static int launch_contract_dma_impl(int kind, const ContractParams& p, int nwg, hipStream_t st) {
#define BTX_LAUNCH_DMA(KIND, NW, PW, TR)                                                                                \
  do {                                                                                                                  \
    auto kfn = contract_dma_kernel<PREC, KIND, NW, PW, TR>;                                                             \
    static bool attr_done = false;                                                                                      \
    if (!attr_done) {                                                                                                   \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, DmaLds<NW>::BYTES); \
      if (e != hipSuccess) return (int)e;                                                                               \
      attr_done = true;                                                                                                 \
    }                                                                                                                   \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(64 * NW), DmaLds<NW>::BYTES, st, p);                                        \
  } while (0)
  int rc = launch_presample_impl<PREC>(kind, p, st);
  if (rc) return rc;
  // the pointwise form where its contract holds (two workgroups per CU: the shapes that are all prologue and store side)
  const bool pw = p.pointwise && !p.transposed && !p.sign_unaligned && p.pt_nw == 4 && !p.pt_nopw;
  if (pw) { if (kind == 0) BTX_LAUNCH_DMA(0, 4, true, false); else BTX_LAUNCH_DMA(1, 4, true, false); }
  else if (p.pt_nw == 4 && p.transposed) { if (kind == 0) BTX_LAUNCH_DMA(0, 4, false, true); else BTX_LAUNCH_DMA(1, 4, false, true); }
  else if (p.pt_nw == 4) { if (kind == 0) BTX_LAUNCH_DMA(0, 4, false, false); else BTX_LAUNCH_DMA(1, 4, false, false); }
  else if (p.transposed) { if (kind == 0) BTX_LAUNCH_DMA(0, 8, false, true); else BTX_LAUNCH_DMA(1, 8, false, true); }
  else { if (kind == 0) BTX_LAUNCH_DMA(0, 8, false, false); else BTX_LAUNCH_DMA(1, 8, false, false); }
#undef BTX_LAUNCH_DMA
  return (int)hipGetLastError();
}

}  // namespace btx
