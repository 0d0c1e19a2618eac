// btx_contract_taps.h — tap-unrolled form of the patch kernel (btx_contract_patch.h) for stride-1 2-D convolutions whose
// filter size is a compile-time constant (3x3: 13 of the 20 ResNet18 convolutions, 16 of the 53 ResNet50 ones).
//
// Same data structures as contract_patch_kernel (halo'd input patch of the tile in a two-slot LDS ring, pre-sampled
// weight tiles in a four-slot ring, sign words per patch pixel, staged epilogue), same K order, same noise indices —
// bit-identical results (KG == 1).  What changes is the instruction stream of the K loop.  Round-1 counters on the
// ResNet18 layer1 shape: 235 instructions per wave per K-stage for 16 MFMAs (~100 SALU: run-time tap/block bookkeeping,
// the vmcnt decision tree, DMA schedule loops; ~100 VALU of which 48 are the s_in masks); the stage was ISSUE-bound
// (the kernel without its MFMAs ran at 78 % of the time of the full kernel).  Here the taps of a channel block are
// unrolled (~103 instructions per stage: 16 MFMA, 14 LDS, 3 DMA, ~49 VALU, ~6 SALU, the counted waits):
//
//   * every per-stage decision is a compile-time constant: which tap, which patch/sign slot, which pieces of the next
//     block's patch this stage fetches, how many VMEM operations may stay in flight at its end (`s_waitcnt vmcnt(n)`
//     with an immediate) — the stage body has no branch except the wave-uniform "last block of the tile" test;
//   * the DMA schedule is the same in every block: a patch piece that does not exist for this tile is fetched (as
//     zeros, the descriptor's out-of-range rule) into a 1-KiB scratch area instead of being skipped, so the counts stay
//     static;
//   * weight-tile DMAs take their stage offset in the scalar offset operand: no VALU, no per-stage 64-bit address math;
//   * the MC sample word (BtxRng.sample_idx_dev) is requested first and the sign keys are derived AFTER the first DMAs
//     are in flight (contract_patch_kernel: a dependent scalar load + two Philox calls in front of everything).
//
// Measured (tools/gpu_diag.py trace, ResNet18 layer1/2 shapes): 1115-1170 cycles per stage for two co-resident 4-wave
// blocks (1024 = the matrix pipe's own time) against 1830 before.
#pragma once
#include <type_traits>
#include "btx_contract.h"
#include "btx_contract_dma.h"
#include "btx_epilogue.h"
#include "btx_mma.h"

namespace btx {

__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff,
                                       unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff,
                                           0, 0);
}

// end of a K-stage: at most N of this wave's VMEM operations stay in flight, its LDS reads are back, all waves meet
template <int N>
__device__ __forceinline__ void end_stage() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// the lane view of the prologue: BTX_TAPS_FLAT=1 reads every field of the lane arithmetic in ONE batch of scalar loads
// (lane_view_flat, btx_contract.h) instead of one dependent batch per `if` of lane_view
#ifndef BTX_TAPS_FLAT
#define BTX_TAPS_FLAT 0
#endif
#if BTX_TAPS_FLAT
#define BTX_TAPS_PARAMS(name, logical_var) BTX_SECTION_PARAMS_FLAT(name, logical_var)
#else
#define BTX_TAPS_PARAMS(name, logical_var) BTX_SECTION_PARAMS(name, logical_var)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int TP_MAXNI = 6;  // 1-KiB patch pieces per wave: the 4-wave plan caps the patch at 22 pieces (btx_api.hip)

// pieces of the next block's patch fetched in stage t of a block: stages 0..T-4 share the TP_MAXNI piece slots
template <int T>
constexpr int tp_pieces(int t) {
  constexpr int PST = T - 3;
  constexpr int PPS = (TP_MAXNI + PST - 1) / PST;
  if (t < 0) t += T;
  if (t >= PST) return 0;
  int n = 0;
  for (int i = 0; i < PPS; ++i)
    if (t + PST * i < TP_MAXNI) ++n;
  return n;
}

// KG = K-groups per workgroup.  KG == 1: 4 waves, two workgroups per CU (free-running against each other: one block's
// load/issue phases fall beside the other's MFMAs — 93 % of the matrix pipe while both are in their K loops).
// KG == 2 (layers with few pixel tiles, 14x14 and 7x7 maps: at most one 4-wave block per CU, whose single wave per
// SIMD keeps the matrix pipe 57 % busy): 8 waves, one workgroup per CU; the two groups of 4 waves contract the two
// halves of the workgroup's channel blocks on the SAME output tile, each with its own patch / sign / weight rings in its
// own half of the LDS, and meet at the end: group 1 hands its accumulator registers over through LDS, group 0 stores
// the tile.  Split-K without partial sums in HBM and without a reduce launch.  The two groups share every s_barrier,
// and eight waves marching in lock-step issue their loads together and then queue for the matrix pipe together
// (measured: 1800 cycles per stage), so a K-group's stage is two halves — H1: DMA issue + the LDS reads of THIS stage's
// fragments, H2: the MFMAs — with a barrier after each, and group 1 runs half a stage behind group 0 (one extra barrier
// before its first stage, group 0 one after its last): one group's MFMA half sits beside the other's load half.
// DIRECT (bf16, KG == 1; the host launches it when ContractParams.ep_direct): the store side runs from the fragment registers
// (direct_epilogue, btx_epilogue.h) — its own instantiation of the kernel, because with both store sides behind a run-time
// branch of one kernel hipcc spills (the staged side's address arithmetic is interleaved with the fold).
// WIDE (Reparameterization, bf16, KG == 1; the host launches it when ContractParams.pt_wide): the wave's tile is 64 pixels x 128
// channels — 2 x 4 MFMA tiles, the register budget Flipout spends on its second accumulator set.  A Reparameterization stage on
// the 2 x 2 tile reads 8 fragments for 8 MFMAs and re-fetches the patch once per 64 output channels; here the stage's second
// weight tile (n-tile 2*ntile + 1, in the LDS slot Flipout's delta tile takes) shares the activation fragments of the first:
// 12 reads for 16 MFMAs, half the patch DMA per MFMA, the K loop of the Flipout kernel without its sign masks.  The workgroup
// stores two 64-channel tiles, one after the other, through the same staging area.
template <int PREC, int KIND, int KH, int KW, int KG, bool DIRECT = false, bool WIDE = false>
__global__ __launch_bounds__(256 * KG, 2) void contract_taps_kernel(const ContractParams) {
  static_assert(!DIRECT || (PREC == 1 && KG == 1), "direct store side: bf16, one K-group");
  static_assert(!WIDE || (PREC == 1 && KG == 1 && KIND == 0 && !DIRECT), "wide tile: bf16 Reparameterization, one K-group");
  constexpr int K2 = (KIND == 1 || WIDE) ? 1 : 0;  // two weight tiles per stage and two accumulator sets
  BTX_TAPS_PARAMS(p, logical);  // prologue + K loop; the store side has its own view (btx_contract.h)
  constexpr int NW = 4, NT = 256, MI = 2, T = KH * KW;
  static_assert(T >= 5 && T <= 32, "tap-unrolled kernel: 5..32 taps");
  constexpr int MAXNI = TP_MAXNI;
  constexpr int PST = T - 3;                      // stages 0..T-4 of a block carry the next block's patch pieces
  constexpr int WOPS = K2 ? 2 : 1;                // weight DMA instructions per wave per stage
  using ACT = typename std::conditional<PREC == 1, __bf16, float>::type;
  constexpr int G = (PREC == 1) ? 8 : 4;
  constexpr int BK = NG * G;
  constexpr int ESZ = (int)sizeof(ACT);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];

  // the MC sample word first: its latency hides behind the index arithmetic and the first DMA issues
  uint32_t smp = p.sample;
  if (p.sample_ptr) smp = sample_word_scalar(p.sample_ptr);  // s_load: a plain dereference is a global load waited for on the spot

  const int tid = threadIdx.x & 255;  // thread / wave index inside the K-group
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = wave_all & 3;
  const int kg = (KG == 1) ? 0 : (wave_all >> 2);
  unsigned char* const smem = smem_all + kg * p.pt_lds_g;

#ifdef BTX_PT_TRACE
  const uint32_t tr_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
  const uint32_t tr_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime();  // constant 100 MHz reference clock
  uint32_t tr_t1 = 0, tr_t2 = 0, tr_s[4] = {0, 0, 0, 0};
  uint32_t tr_p[6] = {0, 0, 0, 0, 0, 0};  // prologue sub-stamps (pt_tune bit 7)
#define BTX_TR_P(i) do { __builtin_amdgcn_sched_barrier(0); tr_p[i] = (uint32_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define BTX_TR_P(i) do { } while (0)
#endif
  uint32_t u_mtile, u_rem, u_split, u_ntile, u_group, u_t;
  const uint32_t ntg = (uint32_t)(WIDE ? p.ntiles >> 1 : p.ntiles);  // n-tiles of the grid (wide: fd_ntiles / fd_inner are made for it)
  if (p.wg_order) fdivmod((uint32_t)logical, p.fd_mtiles, (uint32_t)p.mtiles, u_rem, u_mtile);  // weight-major (btx_api.hip)
  else fdivmod((uint32_t)logical, p.fd_inner, ntg * (uint32_t)(p.groups * p.ksplits), u_mtile, u_rem);
  fdivmod(u_rem, p.fd_ksplits, (uint32_t)p.ksplits, u_t, u_split);
  fdivmod(u_t, p.fd_ntiles, ntg, u_group, u_ntile);
  const int mtile = (int)u_mtile, split = (int)u_split, ntile = (int)u_ntile, group = (int)u_group;

  // tile origin.  Plain tiles: pt_G whole images or pt_R rows of one image, full width.  Tall strips (pt_tall): pt_R rows
  // of the batch seen as one tall image of pt_P virtual rows per image, pt_Wt columns wide (ContractParams).
  const bool tall = p.pt_tall != 0;
  uint32_t u_ig, u_rt;
  if (tall) fdivmod((uint32_t)mtile, p.fd_ncs, (uint32_t)p.pt_ncs, u_rt, u_ig);  // u_ig = column strip
  else fdivmod((uint32_t)mtile, p.fd_rtiles, (uint32_t)p.pt_rtiles, u_ig, u_rt);
  const int img0 = tall ? 0 : (int)u_ig * p.pt_G, row0 = (int)u_rt * p.pt_R, col0 = tall ? (int)u_ig * p.pt_Wt : 0;
  // patch pixel q -> input pixel index (image, row, column flattened), or "outside" (conv padding, tile tail)
  auto patch_src = [&](int q, bool& ok) __attribute__((always_inline)) -> uint32_t {
    uint32_t ut, upc, ugi, upr;
    fdivmod((uint32_t)q, p.fd_ptWp, (uint32_t)p.pt_Wp, ut, upc);
    int img, ih, iw;
    if (tall) {  // patch row ut = virtual input row row0 + ut: rows [0, ph) of every period are the zero rows
      fdivmod((uint32_t)row0 + ut, p.fd_P, (uint32_t)p.pt_P, ugi, upr);
      img = (int)ugi; ih = (int)upr - p.ph; iw = col0 + (int)upc - p.pw;
      ok = (int)ut < p.pt_Rp;
    } else {
      fdivmod(ut, p.fd_ptRp, (uint32_t)p.pt_Rp, ugi, upr);
      img = img0 + (int)ugi; ih = row0 + (int)upr - p.ph; iw = (int)upc - p.pw;
      ok = true;
    }
    ok = ok && img < p.NB && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    return (uint32_t)((img * p.H + ih) * p.W + iw);
  };
  const int ncb_total = p.Cg / BK;
  const int cb_per = p.kper / BK;  // channel blocks of this workgroup (host: a multiple of KG and every split full when KG > 1)
  const int cb0 = split * cb_per + kg * (cb_per / KG);
  const int ncb = (KG == 1) ? min(ncb_total, cb0 + cb_per) - cb0 : cb_per / KG;
  const int a_stage = p.pt_astage, s_stage = p.pt_astage >> 4;
  const int PT_A_OFF = 0, PT_S_OFF = 2 * a_stage, PT_W_OFF = 2 * a_stage + 2 * s_stage;
  const int PT_X_OFF = PT_W_OFF + PT_WD * DW_STAGE;  // 1-KiB scratch: destination of the pieces a tile does not have

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);

  // ---- weight loader: wave w fetches row w of the stage's mu tile (+ row w of its delta tile): 1 KiB each.
  //      byte offset = [tile base + k-granule row of the stage] (scalar) + [row w, lane] (vector, constant)
  const uint32_t w_voff = (uint32_t)lane * 16u + (uint32_t)wave * 1024u;
  const uint32_t w_sbase = (uint32_t)(group * p.ntiles + (WIDE ? 2 * ntile : ntile)) * (uint32_t)(p.K / G) * 1024u;
  const uint32_t w_second = WIDE ? (uint32_t)(p.K / G) * 1024u : p.wt_delta_off;  // the stage's second tile: next n-tile | delta
  const uint32_t CgG = (uint32_t)(p.Cg / G);
  const int w_lds = PT_W_OFF + wave * 1024;
  int wslot = 0;  // ring slot of the stage being multiplied
  auto issue_w = [&](uint32_t tap, uint32_t cb, int slot) __attribute__((always_inline)) {
    const uint32_t soff = w_sbase + (tap * CgG + cb * (uint32_t)NG) * 1024u;
    unsigned char* ld = smem + w_lds + slot * DW_STAGE;
    dma16s(wt_rsrc, w_voff, soff, ld);
    if constexpr (K2) dma16s(wt_rsrc, w_voff, soff + w_second, ld + 4096);
  };
  if (ncb > 0) issue_w(0u, (uint32_t)cb0, 0);

  // ---- sign role: thread t owns the words of patch pixels t and t+256 (element offset of channel 0 of the group).  The
  // same map — patch pixel -> input pixel, or "outside" — is what the patch loader below needs for ITS pixels: it is
  // evaluated once per patch pixel here and handed over through LDS (the second sign slot, free until the first block's
  // stage 0) instead of six more times per thread (6 x ~100 instructions in front of the last patch DMA of the prologue).
  uint32_t sg_off[2];
  bool sg_ok[2];
  uint32_t* const pix_tab = (uint32_t*)(smem + PT_S_OFF + s_stage);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = tid + NT * j;
    sg_ok[j] = q < p.pt_PP;
    bool inside;
    const uint32_t ipix = patch_src(sg_ok[j] ? q : 0, inside);
    sg_off[j] = ipix * (uint32_t)p.C + (uint32_t)(group * p.Cg);  // outside pixels hold zeros: any word will do
    if (q * 4 < s_stage) pix_tab[q] = (sg_ok[j] && inside) ? sg_off[j] : 0xffffffffu;
    if (p.sign_in)  // explicit signs (parity mode) are read from memory: only pixels inside the input exist
      sg_ok[j] = sg_ok[j] && inside;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- patch loader: DMA instruction j of wave w moves patch pixels 16*(w + 4j) + (lane>>2), granule slot lane&3
  const int g_lane = (lane & 3) ^ ((lane >> 4) & 3);
  uint32_t pp_boff[MAXNI];
  uint32_t pmask = 0;  // bit j: piece j of this wave exists
  // all table reads first (one LDS round trip instead of one per piece; entries behind pt_PP hold whatever the slot held)
  uint32_t pp_e[MAXNI];
#pragma unroll
  for (int j = 0; j < MAXNI; ++j) {
    const int q = 16 * (wave + NW * j) + (lane >> 2);
    pp_e[j] = pix_tab[q < p.pt_PP ? q : 0];
  }
#pragma unroll
  for (int j = 0; j < MAXNI; ++j) {
    const int q = 16 * (wave + NW * j) + (lane >> 2);
    uint32_t bo = DMA_OOB;
    if (j < p.pt_NI && q < p.pt_PP) {
      const uint32_t e = pp_e[j];
      if (e != 0xffffffffu) bo = (e + (uint32_t)(G * g_lane)) * (uint32_t)ESZ;
    }
    pp_boff[j] = bo;
    if (j < p.pt_NI && 16 * (wave + NW * j) < p.pt_PP) {
      pmask |= 1u << j;
      if (ncb > 0)
        dma16(x_rsrc, bo == DMA_OOB ? DMA_OOB : bo + (uint32_t)(cb0 * BK * ESZ), smem + PT_A_OFF + (wave + NW * j) * 1024);
    }
  }
  pmask = __builtin_amdgcn_readfirstlane(pmask);
  BTX_TR_P(0);  // patch DMAs issued
  if (ncb > 0) {  // stages 1 and 2 (taps 1, 2 of the first block)
    issue_w(1u, (uint32_t)cb0, 1);
    issue_w(2u, (uint32_t)cb0, 2);
  }

  // ---- the sign keys of this (sample, layer): needed by the sign words and by the epilogue
  RngLive rl = {smp, p.kin_a, p.kin_b, p.kout_a, p.kout_b};
  if (p.sample_ptr || p.lanes > 1) {  // lanes: the host's keys are those of lane 0
    rl.sample = __builtin_amdgcn_readfirstlane(smp);
    if constexpr (KIND == 1) {
      const uint32_t si = p.swap_signs ? 3u : 2u, so = p.swap_signs ? 2u : 3u;  // BTX_STREAM_SIGN_IN = 2, _OUT = 3
      const BtxPhilox4 ki = btx_philox4x32_10(0u, rl.sample, p.layer, si, p.seed_lo, p.seed_hi);
      const BtxPhilox4 ko = btx_philox4x32_10(0u, rl.sample, p.layer, so, p.seed_lo, p.seed_hi);
      rl.kin_a = __builtin_amdgcn_readfirstlane(ki.x[0]); rl.kin_b = __builtin_amdgcn_readfirstlane(ki.x[1]);
      rl.kout_a = __builtin_amdgcn_readfirstlane(ko.x[0]); rl.kout_b = __builtin_amdgcn_readfirstlane(ko.x[1]);
    }
  }

  BTX_TR_P(1);  // sign keys derived (sample word arrived)
  auto write_signs = [&](int slot, int cb) __attribute__((always_inline)) {
    if constexpr (KIND == 1) {
      unsigned char* ss = smem + PT_S_OFF + slot * s_stage;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (sg_ok[j]) {
          const uint32_t off = sg_off[j] + (uint32_t)(cb * BK);
          uint32_t w;
          if (p.sign_in) w = sign_word_explicit(p.sign_in, off, p.x_bytes / (uint32_t)ESZ);
          else w = btx_sign_word(off >> 5, rl.kin_a, rl.kin_b);
          if constexpr (G == 4) w <<= 8 * ((off >> 4) & 1);
          *(uint32_t*)(ss + (tid + NT * j) * 4) = w;
        }
      }
    }
  };

  // ---- MFMA role: wave owns output pixels [64*wave, +64) of the tile, flattened (image, row, col)
  int q0[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int pl = wave * 64 + mi * 32 + l31;
    uint32_t ut, uc, ugi, ur;
    if (tall) {  // pixels that do not exist (dummy rows, tile tail) multiply whatever the patch holds there: never stored
      fdivmod((uint32_t)pl, p.fd_Wt, (uint32_t)p.pt_Wt, ur, uc);
      q0[mi] = ((int)ur < p.pt_R) ? (int)ur * p.pt_Wp + (int)uc : 0;
    } else {
      fdivmod((uint32_t)pl, p.fd_Wo, (uint32_t)p.Wo, ut, uc);
      fdivmod(ut, p.fd_ptR, (uint32_t)p.pt_R, ugi, ur);
      const int c = (int)uc, r = (int)ur, gi = (int)ugi;
      const bool ok = (gi < p.pt_G) && (img0 + gi < p.NB) && (row0 + r < p.Ho);
      q0[mi] = ok ? (gi * p.pt_Rp + r) * p.pt_Wp + c : 0;
    }
  }
#ifdef BTX_PT_NOJUMP
  // MEASUREMENT ONLY (wrong results): the lane's patch pixel = its raster index — 32 consecutive pixels per MFMA tile, no +2 step
  // at a row end, i.e. fragment reads free of LDS bank conflicts whatever the tap: the time the conflicts cost, as an upper bound
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) q0[mi] = (wave * 64 + mi * 32 + l31) & 127;
#endif
  const int row_step = p.dh * p.pt_Wp;  // patch-pixel offset of tap (kh, kw) = kh*row_step + kw*dw (wave-uniform)

  // bf16, one K-group: the accumulators are started by the first stage's MFMAs (zero C operand, stage_mma<..., ZERO>) instead
  // of 128 v_mov in the prologue; what that stage does not touch (a dead second pixel tile, a block without K work) is
  // cleared explicitly
  constexpr bool ZI = (PREC == 1 && KG == 1);
  f32x16 accm[MI][2], accd[MI][2];
  auto clear_acc = [&](int a0) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (a >= a0) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }
  };
  if (!ZI || ncb <= 0) clear_acc(0);

  using Frag = StageFragT<MI>;
  auto load_frag = [&](Frag& f, int aslot, int toffv, int wsl, auto mia_tag) __attribute__((always_inline)) {
    constexpr int MIA = decltype(mia_tag)::value;
    const unsigned char* as = smem + PT_A_OFF + aslot * a_stage;
    const unsigned char* ss = smem + PT_S_OFF + aslot * s_stage;
    const unsigned char* ws = smem + PT_W_OFF + wsl * DW_STAGE;
    int q[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) q[mi] = q0[mi] + toffv;
    if constexpr (BTX_PT_ABL & 2) {  // measurement builds: no fragment reads
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk) {
#pragma unroll
        for (int mi = 0; mi < MIA; ++mi) f.a[kk][mi] = (u32x4){(uint32_t)q[mi], 5u, 1u, 4u};
        f.wm[kk][0] = f.wm[kk][1] = (u32x4){7u, 7u, 1u, (uint32_t)wsl};
      }
#pragma unroll
      for (int mi = 0; mi < MIA; ++mi) f.sw[mi] = (uint32_t)q[mi];
      return;
    }
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      const int row = 2 * kk + h;
#pragma unroll
      for (int mi = 0; mi < MIA; ++mi) f.a[kk][mi] = *(const u32x4*)(as + q[mi] * 64 + ((row ^ ((q[mi] >> 2) & 3)) * 16));
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) f.wm[kk][ni] = *(const u32x4*)(ws + (row * BN + ni * 32 + l31) * 16);
    }
    if constexpr (KIND == 1) {
#pragma unroll
      for (int mi = 0; mi < MIA; ++mi) f.sw[mi] = *(const uint32_t*)(ss + q[mi] * 4);
    }
  };

  BTX_TR_P(2);  // index arithmetic of the sign and MFMA roles
  if (ncb > 0) {
    write_signs(0, cb0);
    BTX_TR_P(3);
    // KG == 1: patch of the first block, W(0) and W(1) landed — iteration 0 prefetches the fragments of stage 1; W(2),
    // issued last, may still be in flight.  KG == 2: a stage reads its own fragments: W(1) may be in flight as well.
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(KG == 1 ? WOPS : 2 * WOPS) : "memory");
    BTX_TR_P(4);  // first patch + W(0), W(1) landed, all waves met
    Frag fa, fb;
    // The wave's second 32-pixel tile may be pure tile padding (224-pixel tiles: 4 rows of 56, wave 3; 196-pixel tiles:
    // wave 3 as well): its MFMAs, fragment reads and sign masks are skipped — an eighth of the block's matrix work.  The
    // choice is wave-uniform, so the K loop exists in two instantiations and every wave still meets every barrier.
    const int nvalid_px = tall ? p.pt_R * p.pt_Wt : min(p.pt_G, p.NB - img0) * min(p.pt_R, p.Ho - row0) * p.Wo;
    const bool mi1_dead = wave * 64 + 32 >= nvalid_px;
    auto kloop = [&](auto mia_tag) __attribute__((always_inline)) {
    constexpr int MIA = decltype(mia_tag)::value;
    if constexpr (KG == 1) load_frag(fa, 0, 0, 0, mia_tag);
#ifdef BTX_PT_TRACE
    tr_t1 = (uint32_t)__builtin_amdgcn_s_memtime();
    uint32_t tr_tA = tr_t1, tr_ab = 0, tr_lg = 0, tr_bc = 0, tr_cd = 0;
#endif
    // One channel block = T unrolled stages.  PAR = parity of the block = its patch / sign slot; the fragment register
    // sets alternate per stage (KG == 1), T may be odd, hence two instantiations.  `last` (wave-uniform): no next block.
    auto block = [&](auto par_tag, auto first_tag, int cbi, bool last) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_tag)::value;
      constexpr bool FIRST = decltype(first_tag)::value;  // first block of the tile: its first stage starts the accumulators
      static_for<0, T>([&](auto t_tag) __attribute__((always_inline)) {
        constexpr int t = decltype(t_tag)::value;
        constexpr int sp = (PAR * T + t) & 1;
        Frag& cur = (KG == 1 && sp) ? fb : fa;
        Frag& nxt = (KG == 1 && sp) ? fa : fb;
        // keep the per-tap LDS addresses out of long-lived registers: without this the compiler precomputes the address
        // vectors of all T taps outside the block loop (~40 VGPRs) and spills; a reload is a VMEM load, and its
        // compiler-inserted vmcnt(0) drains the DMA pipeline
        asm volatile("" : "+v"(q0[0]), "+v"(q0[1]));
        // 1. W(s+3)
        constexpr int t3 = (t + 3) % T, c3 = (t + 3) / T;
        if constexpr (!(BTX_PT_ABL & 4)) {
          if constexpr (c3 == 0) {
            issue_w((uint32_t)t3, (uint32_t)(cb0 + cbi), (wslot + 3) & 3);
          } else {
            if (!last) issue_w((uint32_t)t3, (uint32_t)(cb0 + cbi + 1), (wslot + 3) & 3);
          }
        }
        // 2. this stage's share of the next block's patch (+ its sign words at the first stage)
        constexpr int KP = tp_pieces<T>(t);
        if constexpr (t < PST && !(BTX_PT_ABL & 4)) {
          if (!last) {
            const uint32_t cboff = (uint32_t)((cb0 + cbi + 1) * BK * ESZ);
#pragma unroll
            for (int i = 0; i < KP; ++i) {
              const int j = t + PST * i;
              const uint32_t bo = pp_boff[j];
              unsigned char* dst = ((pmask >> j) & 1u) ? smem + PT_A_OFF + (PAR ^ 1) * a_stage + (wave + NW * j) * 1024
                                                        : smem + PT_X_OFF;
              dma16(x_rsrc, bo == DMA_OOB ? DMA_OOB : bo + cboff, dst);
            }
            if constexpr (t == 0) write_signs(PAR ^ 1, cb0 + cbi + 1);
          }
        }
        DeltaFrag df;
        if constexpr (KG == 1) {
          // 3. delta weights of this stage, then the fragments of the next one
          load_delta<K2>(df, smem + PT_W_OFF + wslot * DW_STAGE, l31, h);
          constexpr int t1 = (t + 1) % T, c1 = (t + 1) / T;
          load_frag(nxt, PAR ^ c1, (t1 / KW) * row_step + (t1 % KW) * p.dw, (wslot + 1) & 3, mia_tag);
          // 4. multiply
          if constexpr (FIRST && t == 0) stage_mma<PREC, K2, MI, MIA, true, KIND == 1>(cur, df, accm, accd, l31, h);
          else stage_mma<PREC, K2, MI, MIA, false, KIND == 1>(cur, df, accm, accd, l31, h);
          // 5. W(s+2) — and, from stage T-3 on, every piece of the next patch — landed; meet the other waves
#ifdef BTX_PT_TRACE
          {  // split the stage end: issue+MFMA | LDS reads back | VMEM wait | barrier
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t tB = (uint32_t)__builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t tB2 = (uint32_t)__builtin_amdgcn_s_memtime();
            if (!last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WOPS + KP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((c3 == 0) ? WOPS : 0) : "memory");
            const uint32_t tC = (uint32_t)__builtin_amdgcn_s_memtime();
            asm volatile("s_barrier" ::: "memory");
            const uint32_t tD = (uint32_t)__builtin_amdgcn_s_memtime();
            tr_ab += tB - tr_tA; tr_lg += tB2 - tB; tr_bc += tC - tB2; tr_cd += tD - tC; tr_tA = tD;
          }
#else
          if (!last) end_stage<WOPS + KP>();
          else end_stage<(c3 == 0) ? WOPS : 0>();
#endif
        } else {
          // H1: this stage's own fragments (their latency hides behind the other group's MFMA half)
          load_delta<KIND>(df, smem + PT_W_OFF + wslot * DW_STAGE, l31, h);
          load_frag(cur, PAR, (t / KW) * row_step + (t % KW) * p.dw, wslot, mia_tag);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_barrier" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          // H2: multiply
          stage_mma<PREC, KIND, MI, MIA>(cur, df, accm, accd, l31, h);
          __builtin_amdgcn_sched_barrier(0);
          // W(s+1) — read by the next stage's H1 — and from stage T-1 on the next patch landed: everything issued before
          // the last two iterations (the pieces of iteration s-2 were issued after its W)
          if (!last) {
            end_stage<2 * WOPS + KP + tp_pieces<T>(t - 1) + tp_pieces<T>(t - 2)>();
          } else {
            constexpr int a_ = (t + 3 < T) ? 1 : 0, b_ = (t == 0) ? 1 : ((t + 2 < T) ? 1 : 0);
            end_stage<WOPS * (a_ + b_)>();
          }
        }
        wslot = (wslot + 1) & 3;
      });
    };
    if (KG == 2 && kg == 1) asm volatile("s_barrier" ::: "memory");
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if constexpr (ZI) {
      if constexpr (MIA < MI) clear_acc(MIA);
      block(P0{}, std::true_type{}, 0, ncb == 1);
      int cbi = 1;
      for (; cbi + 2 <= ncb; cbi += 2) {
        block(P1{}, std::false_type{}, cbi, false);
        block(P0{}, std::false_type{}, cbi + 1, cbi + 2 == ncb);
      }
      if (cbi < ncb) block(P1{}, std::false_type{}, cbi, true);
    } else {
      int cbi = 0;
      for (; cbi + 2 <= ncb; cbi += 2) {
        block(P0{}, std::false_type{}, cbi, false);
        block(P1{}, std::false_type{}, cbi + 1, cbi + 2 == ncb);
      }
      if (cbi < ncb) block(P0{}, std::false_type{}, cbi, true);
    }
    if (KG == 2 && kg == 0) asm volatile("s_barrier" ::: "memory");
#ifdef BTX_PT_TRACE
    tr_s[0] = tr_ab; tr_s[1] = tr_lg; tr_s[2] = tr_bc; tr_s[3] = tr_cd;
#endif
    };  // kloop
    if (mi1_dead) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 2>{});
  }
#ifdef BTX_PT_TRACE
  tr_t2 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif

  // =================== epilogue (btx_epilogue.h) ============================================================
  if constexpr (BTX_PT_ABL & 32) {  // measurement builds: no store side at all (the accumulators stay live up to here)
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) asm volatile("" ::"v"(accm[a][b]), "v"(accd[a][b]));
    return;
  }
  {
    BTX_SECTION_PARAMS(pe, logical2);  // the store side's own reads
    const int nimg = min(pe.pt_G, pe.NB - img0), nrow = min(pe.pt_R, pe.Ho - row0);
    const int nvalid = nimg * nrow * pe.Wo;
    const uint32_t m0 = (uint32_t)(img0 * pe.Ho + row0) * (uint32_t)pe.Wo;
    const PixTall pmt = {pe, row0, col0};
    if constexpr (KG == 1) {
      if constexpr (DIRECT) {
        // opaque copies of the thread's ids: without them the compiler computes the store side's lane-dependent addresses
        // in front of the K loop and keeps them alive across it (the loop then spills)
        int lane_o = lane, tid_o = tid;
        asm volatile("" : "+v"(lane_o), "+v"(tid_o));
        uint32_t gp[2];
        bool gok[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int pl = wave * 64 + mi * 32 + (lane_o & 31);
          if (tall) gp[mi] = pmt(pl, gok[mi]);
          else { gok[mi] = pl < nvalid; gp[mi] = m0 + (uint32_t)pl; }
        }
        direct_epilogue<KIND>(pe, rl, accm, accd, (float*)smem, tid_o, lane_o, ntile, group, gp, gok);
      } else {
        if constexpr (WIDE) {
          // the two 64-channel tiles one after the other through the wave's private staging area; the per-channel constants of
          // BOTH are written first (waves 0 and 1, side by side behind the staging areas: the host reserves 2 KiB there for a
          // wide launch), so the store side has one workgroup barrier, not three
          float* const ba0 = (float*)(smem + NW * PT_EP_WAVE);
          float* const ba1 = ba0 + 4 * BN;
          {
            const bool has_bias = (split == 0) && (pe.mu_b != nullptr);
            const bool has_aff = (pe.ksplits == 1) && ((pe.ep_scale != nullptr) || (pe.ep_shift != nullptr));
            if (has_bias || has_aff) {
              if (tid < 64) ep_fill_constants<0>(pe, rl, ba0, tid, 2 * ntile, group, has_bias, has_aff);
              else if (tid < 128) ep_fill_constants<0>(pe, rl, ba1, tid - 64, 2 * ntile + 1, group, has_bias, has_aff);
            }
            __syncthreads();
          }
          if (tall) {
            staged_epilogue_pm<0, NW, PixTall>(pe, rl, accm, accm, smem, tid, wave, lane, 2 * ntile, group, split, pmt, nullptr, -1, true, ba0);
            staged_epilogue_pm<0, NW, PixTall>(pe, rl, accd, accd, smem, tid, wave, lane, 2 * ntile + 1, group, split, pmt, nullptr, -1, true, ba1);
          } else {
            staged_epilogue<0, NW>(pe, rl, accm, accm, smem, tid, wave, lane, 2 * ntile, group, split, m0, nvalid, nullptr, -1, true, ba0);
            staged_epilogue<0, NW>(pe, rl, accd, accd, smem, tid, wave, lane, 2 * ntile + 1, group, split, m0, nvalid, nullptr, -1, true, ba1);
          }
        } else
        if (tall) staged_epilogue_pm<KIND, NW, PixTall>(pe, rl, accm, accd, smem, tid, wave, lane, ntile, group, split, pmt);
        else staged_epilogue<KIND, NW>(pe, rl, accm, accd, smem, tid, wave, lane, ntile, group, split, m0, nvalid);
      }
    } else {
      // Every wave is behind the barrier of its last stage: the whole LDS is free.  Group 1 -> exchange area
      // [chunk i][thread] x 16 B (a wave writes 1 KiB per instruction; 128 KiB Flipout, 64 KiB Reparameterization),
      // group 0 adds and runs the store.  The per-channel constants sit behind both the exchange area and the 68-KiB
      // staging area of the store.
      float* ba_lds = (float*)(smem_all + 131072);
      const bool to_partial = pe.ksplits > 1;
      const bool has_bias = (split == 0) && (pe.mu_b != nullptr);
      const bool has_aff = !to_partial && ((pe.ep_scale != nullptr) || (pe.ep_shift != nullptr));
      if (kg == 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int i = (mi * 2 + ni) * 4 + c;
              *(f32x4*)(smem_all + (i * 256 + tid) * 16) =
                  (f32x4){accm[mi][ni][4 * c], accm[mi][ni][4 * c + 1], accm[mi][ni][4 * c + 2], accm[mi][ni][4 * c + 3]};
              if constexpr (KIND == 1)
                *(f32x4*)(smem_all + ((16 + i) * 256 + tid) * 16) =
                    (f32x4){accd[mi][ni][4 * c], accd[mi][ni][4 * c + 1], accd[mi][ni][4 * c + 2], accd[mi][ni][4 * c + 3]};
            }
      } else if (has_bias || has_aff) {
        ep_fill_constants<KIND>(pe, rl, ba_lds, tid, ntile, group, has_bias, has_aff);
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int i = (mi * 2 + ni) * 4 + c;
              const f32x4 v = *(const f32x4*)(smem_all + (i * 256 + tid) * 16);
#pragma unroll
              for (int r = 0; r < 4; ++r) accm[mi][ni][4 * c + r] += v[r];
              if constexpr (KIND == 1) {
                const f32x4 w = *(const f32x4*)(smem_all + ((16 + i) * 256 + tid) * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) accd[mi][ni][4 * c + r] += w[r];
              }
            }
      }
      __syncthreads();  // the staging area of the store overlaps the exchange area
      if (kg == 0) {
        if (tall) staged_epilogue_pm<KIND, NW, PixTall>(pe, rl, accm, accd, smem_all, tid, wave, lane, ntile, group, split, pmt,
                                                        nullptr, -1, true, ba_lds);
        else staged_epilogue<KIND, NW>(pe, rl, accm, accd, smem_all, tid, wave, lane, ntile, group, split, m0, nvalid, nullptr,
                                       -1, true, ba_lds);
      }
    }
  }
#ifdef BTX_PT_TRACE
  if (p.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t tr_t3 = (uint32_t)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
      uint32_t* tr = (uint32_t*)p.trace + (size_t)(blockIdx.x * NW * KG + wave_all) * 8;
      tr[0] = tr_t1 - tr_t0; tr[1] = tr_t2 - tr_t1; tr[2] = (uint32_t)__builtin_amdgcn_s_memrealtime() - tr_r0; tr[3] = 0;
      tr[4] = tr_t3 - tr_t2; tr[5] = tr_t3 - tr_t0;
      if (p.pt_tune & 64) { tr[0] = tr_s[0]; tr[1] = tr_s[1]; tr[3] = tr_s[2]; tr[4] = tr_s[3]; }  // stage split instead
      if (p.pt_tune & 128) { tr[0] = tr_p[0] - tr_t0; tr[1] = tr_p[1] - tr_t0; tr[2] = tr_p[2] - tr_t0; tr[3] = tr_p[3] - tr_t0; tr[4] = tr_p[4] - tr_t0; tr[5] = tr_t1 - tr_t0; }
      tr[6] = tr_t0; tr[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
    }
  }
#endif
}

}  // namespace btx
