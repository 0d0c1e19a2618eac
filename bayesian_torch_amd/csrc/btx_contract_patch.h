// btx_contract_patch.h — "patch" variant of the fused sample-and-contract kernel for stride-1 2-D convolutions (gfx950).
//
// The LDS-DMA kernel (btx_contract_dma.h) re-reads every input pixel once per filter tap (9x for 3x3) and fetches
// each 128-byte line twice (a K-stage uses a 64-byte half): its L2->LDS pipeline alone costs ~80 us on the ResNet18
// layer1 shape (ablation).  Here the workgroup owns a 2-D tile of output pixels — G images x R rows x
// the full output width, <= 512 pixels — and, per 32-channel (bf16) / 16-channel (f32) block of the input:
//
//   * the halo'd input PATCH of the tile, (R + (KH-1)dh) x (Wo + (KW-1)dw) pixels x 64 bytes, is DMA'd into LDS
//     ONCE (two-slot ring; the next block's patch is fetched piecewise, one 1-KiB DMA instruction per wave per
//     stage, while the current block's KH*KW stages run), zero-filled outside the image by the buffer descriptor;
//   * all KH*KW taps are multiplied out of that patch: the MFMA fragment address is patch_pixel(lane) + a
//     wave-uniform tap offset.  The pixel-major image with the XOR swizzle slot = granule ^ ((q>>2)&3) stays
//     conflict-free for ANY window of 16 consecutive patch pixels, so shifted windows cost nothing;
//   * s_in is hashed once per patch pixel per channel block (not once per tap), bounds are checked once per patch
//     pixel per workgroup;
//   * weights: MFMA-ready tiles sampled once per launch / per MC sample (btx_presample.h), fetched by LDS-DMA into a
//     four-slot ring three stages ahead.
//
// K order is (channel block, tap) instead of (tap, channel block); the element index k = tap*Cg + c used for the
// weights and for BTX-RNG is the same, so the noise — and up to summation order the result — is identical to the
// other variants.  Split-K splits the channel blocks.
#pragma once
#include <type_traits>
#include "btx_contract.h"
#include "btx_contract_dma.h"
#include "btx_epilogue.h"
#include "btx_presample.h"
#include "btx_mma.h"
#include "btx_contract_taps.h"
#include "btx_contract_taps2.h"
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
#include "../../tools/experimental/btx_contract_taps3.h"  // persistent form: measured and parked, measurement builds only
#endif

namespace btx {

// LDS map (byte offsets computed at run time from the tile plan): 2 patch slots of pt_astage bytes (1 KiB per 16
// patch pixels), 2 sign slots of pt_astage/16 bytes, PT_WD weight tiles of 8 KiB (mu at +0, delta at +4096).
// Blocks of 4 waves (256-pixel tiles) are planned to fit 80 KiB so that TWO of them share a CU: one block's prologue,
// epilogue and barrier waits then overlap the other's MFMAs.  Blocks of 8 waves (512-pixel tiles, one per CU) take the
// shapes whose patch does not fit.


// ContractParams fields used in addition: pt_G, pt_R, pt_Rp, pt_Wp, pt_PP, pt_NI, pt_rtiles, wt (pre-sampled weight
// tiles), wt_bytes, wt_delta_off; kper = channel blocks per split * BK.
// MI = 32-pixel MFMA tiles per wave: 2 (64 pixels, 128 accumulator registers, two waves per SIMD) or 4 (128 pixels, 256
// accumulators — the unified 512-entry register file of a single wave per SIMD; every weight fragment read from LDS
// then feeds twice as many MFMAs).
template <int PREC, int KIND, int NW, int MI>
__global__ __launch_bounds__(64 * NW, (MI == 4) ? 1 : 2) void contract_patch_kernel(const ContractParams) {
  BTX_SECTION_PARAMS(p, logical);  // prologue + K loop; the store side has its own view (btx_contract.h)
  const RngLive rl = rng_live<KIND>(p);
  constexpr int NT = 64 * NW;
  constexpr int WPX = 32 * MI;                      // pixels per wave
  constexpr int MAXNI = (MI == 4) ? 16 : PT_MAXNI;  // DMA pieces per wave per patch slot
  constexpr int SJ = (MI == 4) ? 4 : 2;             // sign words per thread per patch slot
  using ACT = typename std::conditional<PREC == 1, __bf16, float>::type;
  constexpr int G = (PREC == 1) ? 8 : 4;
  constexpr int BK = NG * G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

#ifdef BTX_PT_TRACE
  const uint32_t tr_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
  uint32_t tr_ab = 0, tr_bc = 0, tr_cd = 0, tr_t1 = 0, tr_t2 = 0;
#endif
  uint32_t u_mtile, u_rem, u_split, u_ntile, u_group, u_t;
  if (p.wg_order) fdivmod((uint32_t)logical, p.fd_mtiles, (uint32_t)p.mtiles, u_rem, u_mtile);  // weight-major (btx_api.hip)
  else fdivmod((uint32_t)logical, p.fd_inner, (uint32_t)(p.ntiles * p.groups * p.ksplits), u_mtile, u_rem);
  fdivmod(u_rem, p.fd_ksplits, (uint32_t)p.ksplits, u_t, u_split);
  fdivmod(u_t, p.fd_ntiles, (uint32_t)p.ntiles, u_group, u_ntile);
  const int mtile = (int)u_mtile, split = (int)u_split, ntile = (int)u_ntile, group = (int)u_group;

  // tile geometry
  uint32_t u_ig, u_rt;
  fdivmod((uint32_t)mtile, p.fd_rtiles, (uint32_t)p.pt_rtiles, u_ig, u_rt);
  const int ig = (int)u_ig, rt = (int)u_rt;           // image group, row tile
  const int img0 = ig * p.pt_G, row0 = rt * p.pt_R;   // first image / first output row of the tile
  const int T = p.KH * p.KW;
  const int ncb_total = p.Cg / BK;
  const int cb_per = p.kper / BK;
  const int cb_begin = split * cb_per;
  const int cb_end = min(ncb_total, cb_begin + cb_per);
  const int ncb = cb_end - cb_begin;
  const int nstages = ncb * T;
  const int esz = (int)sizeof(ACT);
  const int a_stage = p.pt_astage, s_stage = p.pt_astage >> 4;
  const int PT_A_OFF = 0, PT_S_OFF = 2 * a_stage, PT_W_OFF = 2 * a_stage + 2 * s_stage;

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);

  // ---- weight loader role: the stage's tile is 4 rows of mu (+ 4 rows of delta, Flipout) of 1 KiB.  8 waves: wave w
  //      fetches mu row w (w < 4) or delta row w-4;  4 waves: wave w fetches mu row w and delta row w.
  const bool w_mu = (NW == 4) || (wave < 4);
  const bool w_dl = (KIND == 1) && ((NW == 4) || (wave >= 4));
  const int w_nops = (w_mu ? 1 : 0) + (w_dl ? 1 : 0);
  const uint32_t w_base = (uint32_t)(group * p.ntiles + ntile) * (uint32_t)(p.K / G) * 1024u + (uint32_t)lane * 16u +
                          (uint32_t)(wave & 3) * 1024u;
  const int w_lds = PT_W_OFF + (wave & 3) * 1024;

  // Stage s <-> (channel block cb_begin + s / T, tap s % T); first k of the stage = tap*Cg + cb*BK.  Both walks over
  // the stages (the weight fetch, three stages ahead, and the fragment loads) keep their position incrementally.
  int wi_s = 0, wi_t = 0, wi_cbk = cb_begin * BK, wi_k0 = cb_begin * BK;  // next stage to fetch
  auto issue_w_next = [&]() __attribute__((always_inline)) {
    const uint32_t go = w_base + (uint32_t)(wi_k0 / G) * 1024u;
    unsigned char* ld = smem + w_lds + (wi_s & (PT_WD - 1)) * DW_STAGE;
    if (w_mu) dma16(wt_rsrc, go, ld);
    if (w_dl) dma16(wt_rsrc, go + p.wt_delta_off, ld + 4096);
    ++wi_s;
    wi_k0 += p.Cg;
    if (++wi_t == T) { wi_t = 0; wi_cbk += BK; wi_k0 = wi_cbk; }
  };
  // the first weight tiles need nothing but the tile indices: fetch them before the pixel geometry is worked out
  for (int s_ = 0; s_ < PT_WD - 1 && s_ < nstages; ++s_) issue_w_next();

  // ---- patch loader role: DMA instruction i (i = wave + 8*j, j < pt_NI) moves patch pixels 16i + (lane>>2),
  //      granule slot lane&3 (source-side swizzle as in the DMA variant).  Byte offset of channel block 0, or OOB.
  const int g_lane = (lane & 3) ^ ((lane >> 4) & 3);
  uint32_t pp_boff[MAXNI];
#pragma unroll
  for (int j = 0; j < MAXNI; ++j) {
    const int q = 16 * (wave + NW * j) + (lane >> 2);
    uint32_t bo = DMA_OOB;
    if (j < p.pt_NI && q < p.pt_PP) {
      uint32_t ut, upc, ugi, upr;
      fdivmod((uint32_t)q, p.fd_ptWp, (uint32_t)p.pt_Wp, ut, upc);
      fdivmod(ut, p.fd_ptRp, (uint32_t)p.pt_Rp, ugi, upr);
      const int pc = (int)upc, pr = (int)upr, gi = (int)ugi;
      const int img = img0 + gi, ih = row0 + pr - p.ph, iw = pc - p.pw;
      if (img < p.NB && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
        bo = ((uint32_t)((img * p.H + ih) * p.W + iw) * (uint32_t)p.C + (uint32_t)(group * p.Cg + G * g_lane)) *
             (uint32_t)esz;
    }
    pp_boff[j] = bo;
    if (nstages > 0 && j < p.pt_NI && 16 * (wave + NW * j) < p.pt_PP)  // patch of the first channel block: go
      dma16(x_rsrc, bo == DMA_OOB ? DMA_OOB : bo + (uint32_t)(cb_begin * BK * esz),
            smem + PT_A_OFF + (wave + NW * j) * 1024);
  }
  // ---- sign role: thread t hashes the words of patch pixels t and t+512 (element offset of channel block 0)
  uint32_t sg_off[SJ];
  bool sg_ok[SJ];
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int q = tid + NT * j;
    sg_ok[j] = q < p.pt_PP;
    const int qq = sg_ok[j] ? q : 0;
    uint32_t ut, upc, ugi, upr;
    fdivmod((uint32_t)qq, p.fd_ptWp, (uint32_t)p.pt_Wp, ut, upc);
    fdivmod(ut, p.fd_ptRp, (uint32_t)p.pt_Rp, ugi, upr);
    const int pc = (int)upc, pr = (int)upr, gi = (int)ugi;
    sg_off[j] = (uint32_t)(((img0 + gi) * p.H + (row0 + pr - p.ph)) * p.W + (pc - p.pw)) * (uint32_t)p.C +
                (uint32_t)(group * p.Cg);
  }
  // ---- MFMA role: wave owns output pixels [64*wave, +64) of the tile, flattened (image, row, col)
  int q0[MI];      // patch pixel of tap (0,0) for each of this lane's output pixels
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int pl = wave * WPX + mi * 32 + l31;
    uint32_t ut, uc, ugi, ur;
    fdivmod((uint32_t)pl, p.fd_Wo, (uint32_t)p.Wo, ut, uc);
    fdivmod(ut, p.fd_ptR, (uint32_t)p.pt_R, ugi, ur);
    const int c = (int)uc, r = (int)ur, gi = (int)ugi;
    const bool ok = (gi < p.pt_G) && (img0 + gi < p.NB) && (row0 + r < p.Ho);
    q0[mi] = ok ? (gi * p.pt_Rp + r) * p.pt_Wp + c : 0;
  }
  // one 1-KiB piece (16 patch pixels x 64 B) of the patch of channel block `cbi` (index relative to cb_begin)
  auto issue_patch_piece = [&](int cbi, int j) __attribute__((always_inline)) {
    const uint32_t cboff = (uint32_t)((cb_begin + cbi) * BK * esz);
    unsigned char* as = smem + PT_A_OFF + (cbi & 1) * a_stage + (wave + NW * j) * 1024;
    uint32_t bo = DMA_OOB;
#pragma unroll
    for (int jj = 0; jj < MAXNI; ++jj)
      if (jj == j) bo = pp_boff[jj];
    dma16(x_rsrc, bo == DMA_OOB ? DMA_OOB : bo + cboff, as);
  };
  auto write_signs = [&](int cbi) __attribute__((always_inline)) {
    if constexpr (KIND == 1) {
      unsigned char* ss = smem + PT_S_OFF + (cbi & 1) * s_stage;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        if (sg_ok[j]) {
          const uint32_t off = sg_off[j] + (uint32_t)((cb_begin + cbi) * BK);
          uint32_t w = p.sign_in ? sign_word_explicit(p.sign_in, off, p.x_bytes / (uint32_t)esz)
                                 : btx_sign_word(off >> 5, rl.kin_a, rl.kin_b);
          if constexpr (G == 4) w <<= 8 * ((off >> 4) & 1);
          *(uint32_t*)(ss + (tid + NT * j) * 4) = w;
        }
      }
    }
  };
  f32x16 accm[MI][2], accd[MI][2];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }

  // Fragments of one K-stage held in registers: activations a[kk][mi] and mean weights wm[kk][ni] (+ the two sign
  // words).  They are double-buffered: the reads of stage s+1 are issued BEFORE the MFMAs of stage s, so that after the
  // per-stage barrier the LDS pipe and the matrix pipe work at the same time instead of one after the other.  The
  // delta weights of a stage are read at the start of its own multiply (the 8 mean MFMAs cover their latency).
  using Frag = StageFragT<MI>;
  auto load_frag = [&](Frag& f, int cbi, int toff, int wslot) __attribute__((always_inline)) {
    const unsigned char* as = smem + PT_A_OFF + (cbi & 1) * a_stage;
    const unsigned char* ss = smem + PT_S_OFF + (cbi & 1) * s_stage;
    const unsigned char* ws = smem + PT_W_OFF + wslot * DW_STAGE;
    int q[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) q[mi] = q0[mi] + toff;
    if constexpr (BTX_PT_ABL & 2) {
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          f.a[kk][i] = (u32x4){(uint32_t)q[0], (uint32_t)kk, 3u, 4u};
          f.wm[kk][i & 1] = (u32x4){(uint32_t)q[1], 7u, (uint32_t)wslot, 4u};
          f.sw[i] = (uint32_t)toff;
        }
    } else {
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk) {
        const int row = 2 * kk + h;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) f.a[kk][mi] = *(const u32x4*)(as + q[mi] * 64 + ((row ^ ((q[mi] >> 2) & 3)) * 16));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) f.wm[kk][ni] = *(const u32x4*)(ws + (row * BN + ni * 32 + l31) * 16);
      }
      if constexpr (KIND == 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) f.sw[mi] = *(const uint32_t*)(ss + q[mi] * 4);
      }
    }
  };
  // =================== main loop ==========================================================================
  // Iteration s (tap t of channel block cbi), every wave:
  //   1. issues the DMA of W(s+3) and its share of the pieces of the NEXT block's patch (+ that block's sign words);
  //   2. issues the LDS reads of the fragments of stage s+1 (its weights and patch became visible at the last barrier);
  //   3. multiplies stage s out of the registers loaded one iteration ago;
  //   4. waits until W(s+2) — and, on the second-to-last stage of a block, the next patch — have landed, and meets the
  //      other waves at the barrier.
  // vmcnt retires in order, so "landed" = at most as many operations outstanding as this wave has issued after the one
  // it needs; those counts are wave-uniform scalars.
#ifdef BTX_PT_PIECE_STAGES
  const int ppst = (p.pt_NI + BTX_PT_PIECE_STAGES - 1) / BTX_PT_PIECE_STAGES;  // measurement: pieces within N stages
#else
  const int ppst = (p.pt_NI + (T > 3 ? T - 4 : 0)) / (T > 3 ? T - 3 : 1);  // pieces per stage: done 3 stages early
#endif
  if (nstages > 0) {
    write_signs(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int nissued = 0, m1 = 0, mpiece = 0;  // marks: nissued right after W(s+2) / after the last patch piece
    int cbi = 0, t = 0;                                           // position of the stage being multiplied
    int l_cbi = 0, l_t = 0, l_kw = 0, l_toff = 0, l_rowoff = 0;   // position of the stage being loaded
    const int row_step = p.dh * p.pt_Wp;
    auto advance_load = [&]() __attribute__((always_inline)) {
      l_toff += p.dw;
      if (++l_kw == p.KW) { l_kw = 0; l_rowoff += row_step; l_toff = l_rowoff; }
      if (++l_t == T) { l_t = 0; ++l_cbi; l_toff = 0; l_rowoff = 0; }
    };
    Frag fa, fb;
    load_frag(fa, 0, 0, 0);
    advance_load();
#ifdef BTX_PT_TRACE
    tr_t1 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    auto iter = [&](int s, Frag& cur, Frag& nxt) __attribute__((always_inline)) {
      const bool next_cb = cbi + 1 < ncb;
      int m2 = nissued;
#ifdef BTX_PT_TRACE
      const uint32_t tA = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
      if (!(BTX_PT_ABL & (4 | 128)) && wi_s < nstages) { issue_w_next(); nissued += w_nops; m2 = nissued; }
      if (!(BTX_PT_ABL & (4 | 256)) && next_cb) {
        for (int j = t * ppst; j < (t + 1) * ppst && j < p.pt_NI; ++j)
          if (16 * (wave + NW * j) < p.pt_PP) { issue_patch_piece(cbi + 1, j); mpiece = ++nissued; }
        if (t == 0) write_signs(cbi + 1);  // the sign slot of block cbi+1 was last read during block cbi-1
      }
      DeltaFrag dfrag;  // this stage's delta weights first, then the prefetch of the next stage's fragments
      load_delta<KIND>(dfrag, smem + PT_W_OFF + (s & (PT_WD - 1)) * DW_STAGE, l31, h);
      if (s + 1 < nstages) { load_frag(nxt, l_cbi, l_toff, (s + 1) & (PT_WD - 1)); advance_load(); }
      stage_mma<PREC, KIND, MI>(cur, dfrag, accm, accd, l31, h);
#ifdef BTX_PT_TRACE
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t tB = (uint32_t)__builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
      int allowed = nissued - m1;
      if (next_cb && t == (T >= 2 ? T - 2 : 0)) allowed = min(allowed, nissued - mpiece);
      if constexpr (!(BTX_PT_ABL & 64)) wait_vmcnt(allowed);
#ifdef BTX_PT_TRACE
      const uint32_t tC = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
      if constexpr (BTX_PT_ABL & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef BTX_PT_TRACE
      const uint32_t tD = (uint32_t)__builtin_amdgcn_s_memtime();
      tr_ab += tB - tA; tr_bc += tC - tB; tr_cd += tD - tC;
#endif
      m1 = m2;
      if (++t == T) { t = 0; ++cbi; }
    };
    int s = 0;
    for (; s + 1 < nstages; s += 2) {
      iter(s, fa, fb);
      iter(s + 1, fb, fa);
    }
    if (s < nstages) iter(s, fa, fb);
  }
  if constexpr (BTX_PT_ABL & 32) return;
#ifdef BTX_PT_TRACE
  tr_t2 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif

  // =================== epilogue (btx_epilogue.h) ============================================================
  {
    // valid pixels of the tile are a prefix of its flattened (image, row, col) order, contiguous in the output
    BTX_SECTION_PARAMS(pe, logical2);
    const int nimg = min(pe.pt_G, pe.NB - img0), nrow = min(pe.pt_R, pe.Ho - row0);
    const int nvalid = nimg * nrow * pe.Wo;
    const uint32_t m0 = (uint32_t)(img0 * pe.Ho + row0) * (uint32_t)pe.Wo;
    if constexpr (MI == 2) {
      staged_epilogue<KIND, NW>(pe, rl, accm, accd, smem, tid, wave, lane, ntile, group, split, m0, nvalid);
    } else {
#pragma unroll
      for (int hf = 0; hf < MI / 2; ++hf) {
        __builtin_amdgcn_sched_barrier(0);  // one half of the 256 accumulators at a time
        staged_epilogue<KIND, NW>(pe, rl, reinterpret_cast<const f32x16(&)[2][2]>(accm[2 * hf]),
                                  reinterpret_cast<const f32x16(&)[2][2]>(accd[2 * hf]), smem, tid, wave, lane, ntile,
                                  group, split, m0, nvalid, nullptr, wave * (MI / 2) + hf, hf == 0);
      }
    }
  }
#ifdef BTX_PT_TRACE
  if (p.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t tr_t3 = (uint32_t)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
      uint32_t* tr = (uint32_t*)p.trace + (size_t)(blockIdx.x * NW + wave) * 8;
      tr[0] = tr_t1 - tr_t0; tr[1] = tr_ab; tr[2] = tr_bc; tr[3] = tr_cd; tr[4] = tr_t3 - tr_t2; tr[5] = tr_t3 - tr_t0;
      tr[6] = tr_t0; tr[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
    }
  }
#endif
}

template <int PREC>
static int launch_contract_patch_impl(int kind, const ContractParams& p, int nwg, hipStream_t st) {
#define BTX_LAUNCH_PT(KIND, NW, MI)                                                                                 \
  do {                                                                                                            \
    auto kfn = contract_patch_kernel<PREC, KIND, NW, MI>;                                                           \
    static bool attr_done = false;                                                                                \
    if (!attr_done) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);   \
      if (e != hipSuccess) return (int)e;                                                                         \
      attr_done = true;                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(64 * NW), p.pt_lds, st, p);                                           \
  } while (0)
  int rc = launch_presample_impl<PREC>(kind, p, st);
  if (rc) return rc;
  if (p.pt_taps == 332) {  // 3x3 / stride 2 / pad 1: the phase-plane form of the tap-unrolled kernel (btx_contract_taps2.h)
#define BTX_LAUNCH_T2(KIND, ...)                                                                                  \
  do {                                                                                                            \
    auto kfn = contract_taps2_kernel<PREC, KIND, ##__VA_ARGS__>;                                                  \
    static bool attr_done = false;                                                                                \
    if (!attr_done) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);   \
      if (e != hipSuccess) return (int)e;                                                                         \
      attr_done = true;                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(256), p.pt_lds, st, p);                                               \
  } while (0)
    if constexpr (PREC == 1) {
      if (p.pt_wide && kind == 0) {  // Reparameterization on 64 x 128 wave tiles
        BTX_LAUNCH_T2(0, true);
        return (int)hipGetLastError();
      }
    }
    if (kind == 0) BTX_LAUNCH_T2(0); else BTX_LAUNCH_T2(1);
#undef BTX_LAUNCH_T2
    return (int)hipGetLastError();
  }
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
  if constexpr (PREC == 1) {
    if (p.pt_taps == 33 && p.pt_persist > 0) {  // persistent form (btx_contract_taps3.h): the grid is pt_persist workgroups
#define BTX_LAUNCH_T3(KIND)                                                                                       \
  do {                                                                                                            \
    auto kfn = contract_taps3_kernel<KIND>;                                                                       \
    static bool attr_done = false;                                                                                \
    if (!attr_done) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);   \
      if (e != hipSuccess) return (int)e;                                                                         \
      attr_done = true;                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, dim3(p.pt_persist), dim3(256), p.pt_lds + 16, st, p);                                      \
  } while (0)
      if (kind == 0) BTX_LAUNCH_T3(0); else BTX_LAUNCH_T3(1);
#undef BTX_LAUNCH_T3
      return (int)hipGetLastError();
    }
  }
#endif
  if (p.pt_taps == 33) {  // 3x3, 4-wave K-groups: the tap-unrolled kernel (btx_contract_taps.h)
#define BTX_LAUNCH_TP(KIND, KG, ...)                                                                               \
  do {                                                                                                            \
    auto kfn = contract_taps_kernel<PREC, KIND, 3, 3, KG, ##__VA_ARGS__>;                                           \
    static bool attr_done = false;                                                                                \
    if (!attr_done) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);   \
      if (e != hipSuccess) return (int)e;                                                                         \
      attr_done = true;                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(256 * KG), p.pt_lds, st, p);                                          \
  } while (0)
    if (p.pt_kg == 2) { if (kind == 0) BTX_LAUNCH_TP(0, 2); else BTX_LAUNCH_TP(1, 2); }
    else {
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
      if constexpr (PREC == 1) {
        if (p.ep_direct) {  // MEASUREMENT ONLY (BTX_DIRECT=1): the store side from the fragment registers, btx_epilogue.h
          if (kind == 0) BTX_LAUNCH_TP(0, 1, true); else BTX_LAUNCH_TP(1, 1, true);
          return (int)hipGetLastError();
        }
      }
#endif
      if constexpr (PREC == 1) {
        if (p.pt_wide && kind == 0) {  // Reparameterization on 64 x 128 wave tiles (btx_contract_taps.h, WIDE)
          BTX_LAUNCH_TP(0, 1, false, true);
          return (int)hipGetLastError();
        }
      }
      if (kind == 0) BTX_LAUNCH_TP(0, 1); else BTX_LAUNCH_TP(1, 1);
    }
#undef BTX_LAUNCH_TP
    return (int)hipGetLastError();
  }
  if (p.pt_mi == 4) { if (kind == 0) BTX_LAUNCH_PT(0, 4, 4); else BTX_LAUNCH_PT(1, 4, 4); }
  else if (p.pt_nw == 4) { if (kind == 0) BTX_LAUNCH_PT(0, 4, 2); else BTX_LAUNCH_PT(1, 4, 2); }
  else { if (kind == 0) BTX_LAUNCH_PT(0, 8, 2); else BTX_LAUNCH_PT(1, 8, 2); }
#undef BTX_LAUNCH_PT
  return (int)hipGetLastError();
}

}  // namespace btx
