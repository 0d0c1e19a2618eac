// btx_contract_taps2.h — the tap-unrolled patch kernel (btx_contract_taps.h) for 3x3 / stride-2 / pad-1 convolutions: the
// three down-sampling convolutions of a ResNet18, which the per-tap LDS-DMA kernel ran at 0.12-0.18 of the MFMA peak
// (every input pixel fetched once per tap it feeds, 64 bytes per request: bound by the address pipeline).
//
// A stride-2 window touches 4 input pixels per output pixel, so the de-interleaved patch of a 224-pixel tile is 1044
// pixels x 64 B = 67 KB per ring slot — two slots do not fit the 80 KB a block may use with two blocks per CU.  What fits
// is ONE PHASE PLANE at a time.  Plane (a, b) holds the input pixels (2i - a, 2j - b): for every tap (kh, kw) with
// kh in {0, 2} if a else {1}, kw in {0, 2} if b else {1}, output pixel (oh, ow) reads plane pixel (oh + (kh == 2),
// ow + (kw == 2)) — a stride-1 access, the same XOR-swizzled pixel-major LDS image as the stride-1 kernel, a wave-uniform tap
// offset in {0, 1, Wp, Wp + 1}.  Per 32-channel block the 9 stages run plane by plane:
//
//   stage  0 1 2 3 | 4     | 5 6   | 7 8          plane (1,1): taps (0,0) (0,2) (2,0) (2,2)   plane (0,0): tap (1,1)
//   plane  (1,1)   | (0,0) | (1,0) | (0,1)        plane (1,0): taps (0,1) (2,1)               plane (0,1): taps (1,0) (1,2)
//
// Three patch slots ((R+1) x (Wo+1) pixels each, <= 17 KiB) rotate: the plane after next is fetched while a plane
// multiplies (stages 0-1 fetch (1,0), stage 4 fetches (0,1), stages 5-6 the next block's (1,1), stages 7-8 its (0,0)) — every
// piece is issued at least two stages before its first use.  Everything else is the stride-1 kernel: static DMA schedule
// and s_waitcnt immediates, weight tiles by scalar-offset DMA (ring of three here, two stages ahead: the LDS budget — which
// is also why a stage reads its own fragments instead of prefetching the next stage's),
// one hashed s_in word per patch pixel per plane, staged epilogue.  K order (channel block, plane, tap) — the noise indices
// k = tap*Cg + c are those of every other variant.
#pragma once
#include <type_traits>
#include "btx_contract.h"
#include "btx_contract_dma.h"
#include "btx_contract_taps.h"
#include "btx_epilogue.h"
#include "btx_mma.h"

namespace btx {

constexpr int T2_MAXNI = 5;  // 1-KiB pieces per wave per plane: planes of at most 272 pixels (17 pieces)
constexpr int T2_WD = 3;     // weight-tile ring depth
constexpr int T2_T = 9;

// static schedule of a channel block (s = stage 0..8)
constexpr int t2_seq(int s) { return s < 4 ? 0 : (s == 4 ? 1 : (s < 7 ? 2 : 3)); }  // plane sequence index of the stage
constexpr int t2_tap(int s) {                                                        // its 3x3 tap, kh*3 + kw
  return s == 0 ? 0 : s == 1 ? 2 : s == 2 ? 6 : s == 3 ? 8 : s == 4 ? 4 : s == 5 ? 1 : s == 6 ? 7 : s == 7 ? 3 : 5;
}
constexpr int t2_a(int k) { return (k == 0 || k == 2) ? 1 : 0; }  // plane (a, b) of sequence index k (mod 4)
constexpr int t2_b(int k) { return (k == 0 || k == 3) ? 1 : 0; }
constexpr int t2_np(int s) { return (s == 0 || s == 5 || s == 7) ? 3 : ((s == 1 || s == 6 || s == 8) ? 2 : (s == 4 ? 5 : 0)); }
constexpr int t2_p0(int s) { return (s == 1 || s == 6 || s == 8) ? 3 : 0; }          // first piece fetched in the stage
constexpr int t2_fseq(int s) { return s < 2 ? 2 : (s == 4 ? 3 : (s < 7 ? 4 : 5)); }  // plane it belongs to (4, 5: next block)

// WIDE (Reparameterization, bf16; ContractParams.pt_wide): the wave's tile is 64 pixels x 128 channels as in
// contract_taps_kernel<..., WIDE> — the stage's second weight tile (n-tile 2 * ntile + 1) in the LDS slot of Flipout's delta tile,
// the same activation fragments, no sign masks, two 64-channel tiles stored one after the other.  A stride-2 stage pays 3.3x the
// patch DMA per MFMA of a stride-1 one (four phase planes per output pixel): sharing the planes between two n-tiles halves it.
template <int PREC, int KIND, bool WIDE = false>
__global__ __launch_bounds__(256, 2) void contract_taps2_kernel(const ContractParams) {
  static_assert(!WIDE || (PREC == 1 && KIND == 0), "wide tile: bf16 Reparameterization");
  BTX_SECTION_PARAMS(p, logical);  // prologue + K loop; the store side has its own view (btx_contract.h)
  constexpr int NW = 4, NT = 256, MI = 2, T = T2_T, MAXNI = T2_MAXNI, WD = T2_WD;
  constexpr int K2 = (KIND == 1 || WIDE) ? 1 : 0;  // two weight tiles per stage and two accumulator sets
  constexpr int WOPS = K2 ? 2 : 1;  // weight DMA instructions per wave per stage
  using ACT = typename std::conditional<PREC == 1, __bf16, float>::type;
  constexpr int G = (PREC == 1) ? 8 : 4;
  constexpr int BK = NG * G;
  constexpr int ESZ = (int)sizeof(ACT);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  uint32_t smp = p.sample;
  if (p.sample_ptr) smp = sample_word_scalar(p.sample_ptr);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  uint32_t u_mtile, u_rem, u_split, u_ntile, u_group, u_t;
  const uint32_t ntg = (uint32_t)(WIDE ? p.ntiles >> 1 : p.ntiles);  // n-tiles of the grid (wide: fd_ntiles / fd_inner are made for it)
  if (p.wg_order) fdivmod((uint32_t)logical, p.fd_mtiles, (uint32_t)p.mtiles, u_rem, u_mtile);
  else fdivmod((uint32_t)logical, p.fd_inner, ntg * (uint32_t)(p.groups * p.ksplits), u_mtile, u_rem);
  fdivmod(u_rem, p.fd_ksplits, (uint32_t)p.ksplits, u_t, u_split);
  fdivmod(u_t, p.fd_ntiles, ntg, u_group, u_ntile);
  const int mtile = (int)u_mtile, split = (int)u_split, ntile = (int)u_ntile, group = (int)u_group;

  uint32_t u_ig, u_rt;
  fdivmod((uint32_t)mtile, p.fd_rtiles, (uint32_t)p.pt_rtiles, u_ig, u_rt);
  const int img0 = (int)u_ig * p.pt_G, row0 = (int)u_rt * p.pt_R;
  const int ncb_total = p.Cg / BK;
  const int cb_per = p.kper / BK;
  const int cb0 = split * cb_per;
  const int ncb = min(ncb_total, cb0 + cb_per) - cb0;
  const int a_stage = p.pt_astage, s_stage = p.pt_astage >> 4;
  const int PT_A_OFF = 0, PT_S_OFF = 3 * a_stage, PT_W_OFF = 3 * a_stage + 3 * s_stage;
  const int PT_X_OFF = PT_W_OFF + WD * DW_STAGE;  // 1-KiB scratch: destination of the pieces a plane does not have

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);

  // ---- weight loader (as in btx_contract_taps.h): wave w fetches row w of the stage's mu tile (+ row w of its delta tile)
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
  if (ncb > 0) issue_w((uint32_t)t2_tap(0), (uint32_t)cb0, 0);

  // ---- plane loader: DMA instruction j of wave w moves plane pixels 16*(w + 4j) + (lane>>2), granule slot lane&3.
  //      pp_base = byte offset of input pixel (2(row0+i), 2jj) of the plane pixel (gi, i, jj); plane (a, b) reads
  //      (a*W + b) pixels before it; pvalid bit 4j + 2a + b: that pixel lies inside the input
  const int g_lane = (lane & 3) ^ ((lane >> 4) & 3);
  uint32_t pp_base[MAXNI];
  uint32_t pvalid = 0, pmask = 0;
#pragma unroll
  for (int j = 0; j < MAXNI; ++j) {
    const int q = 16 * (wave + NW * j) + (lane >> 2);
    uint32_t bo = 0;
    if (j < p.pt_NI && q < p.pt_PP) {
      uint32_t ut, ujj, ugi, ui;
      fdivmod((uint32_t)q, p.fd_ptWp, (uint32_t)p.pt_Wp, ut, ujj);
      fdivmod(ut, p.fd_ptRp, (uint32_t)p.pt_Rp, ugi, ui);
      const int img = img0 + (int)ugi, ih0 = 2 * (row0 + (int)ui), iw0 = 2 * (int)ujj;
      bo = ((uint32_t)((img * p.H + ih0) * p.W + iw0) * (uint32_t)p.C + (uint32_t)(group * p.Cg + G * g_lane)) * (uint32_t)ESZ;
      if (img < p.NB) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            if ((unsigned)(ih0 - a) < (unsigned)p.H && (unsigned)(iw0 - b) < (unsigned)p.W) pvalid |= 1u << (4 * j + 2 * a + b);
      }
    }
    pp_base[j] = bo;
    if (j < p.pt_NI && 16 * (wave + NW * j) < p.pt_PP) pmask |= 1u << j;
  }
  pmask = __builtin_amdgcn_readfirstlane(pmask);
  const uint32_t pd_row = (uint32_t)(p.W * p.C * ESZ), pd_col = (uint32_t)(p.C * ESZ);  // byte step of a = 1 / b = 1
  // pieces [j0, j0 + n) of plane (a, b) of channel block cb into the slot at byte offset `slot_off`
  auto issue_plane = [&](int j0, int n, int a, int b, int cb, int slot_off) __attribute__((always_inline)) {
    const uint32_t delta = (uint32_t)(cb * BK * ESZ) - (uint32_t)a * pd_row - (uint32_t)b * pd_col;
#pragma unroll
    for (int i = 0; i < MAXNI; ++i) {
      if (i < n) {
        const int j = j0 + i;
        const bool ok = (pvalid >> (4 * j + 2 * a + b)) & 1u;
        unsigned char* dst = ((pmask >> j) & 1u) ? smem + PT_A_OFF + slot_off + (wave + NW * j) * 1024 : smem + PT_X_OFF;
        dma16(x_rsrc, ok ? pp_base[j] + delta : DMA_OOB, dst);
      }
    }
  };

  // ---- sign keys
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
  // ---- sign role: thread t owns the words of plane pixels t and t+256; sg_base = element offset of channel 0 of the group
  //      at input pixel (2(row0+i), 2jj); sg_v bit 4j + 2a + b: plane (a, b)'s pixel lies inside the input
  uint32_t sg_base[2];
  uint32_t sg_v = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = tid + NT * j;
    const int qq = q < p.pt_PP ? q : 0;
    uint32_t ut, ujj, ugi, ui;
    fdivmod((uint32_t)qq, p.fd_ptWp, (uint32_t)p.pt_Wp, ut, ujj);
    fdivmod(ut, p.fd_ptRp, (uint32_t)p.pt_Rp, ugi, ui);
    const int img = img0 + (int)ugi, ih0 = 2 * (row0 + (int)ui), iw0 = 2 * (int)ujj;
    sg_base[j] = (uint32_t)((img * p.H + ih0) * p.W + iw0) * (uint32_t)p.C + (uint32_t)(group * p.Cg);
    if (q < p.pt_PP && img < p.NB) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if ((unsigned)(ih0 - a) < (unsigned)p.H && (unsigned)(iw0 - b) < (unsigned)p.W) sg_v |= 1u << (4 * j + 2 * a + b);
    }
  }
  auto write_signs = [&](int a, int b, int cb, int sslot_off) __attribute__((always_inline)) {
    if constexpr (KIND == 1) {
      unsigned char* ss = smem + PT_S_OFF + sslot_off;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if ((sg_v >> (4 * j + 2 * a + b)) & 1u) {  // pixels outside the input are zeros: their word is never needed
          const uint32_t off = sg_base[j] - (uint32_t)((a * p.W + b) * p.C) + (uint32_t)(cb * BK);
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
    fdivmod((uint32_t)pl, p.fd_Wo, (uint32_t)p.Wo, ut, uc);
    fdivmod(ut, p.fd_ptR, (uint32_t)p.pt_R, ugi, ur);
    const int c = (int)uc, r = (int)ur, gi = (int)ugi;
    const bool ok = (gi < p.pt_G) && (img0 + gi < p.NB) && (row0 + r < p.Ho);
    q0[mi] = ok ? (gi * p.pt_Rp + r) * p.pt_Wp + c : 0;
  }
  const int Wp = p.pt_Wp;

  f32x16 accm[MI][2], accd[MI][2];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }

  using Frag = StageFragT<MI>;
  auto load_frag = [&](Frag& f, int aoff, int soff, int toffv, int wsl, auto mia_tag) __attribute__((always_inline)) {
    constexpr int MIA = decltype(mia_tag)::value;
    const unsigned char* as = smem + PT_A_OFF + aoff;
    const unsigned char* ss = smem + PT_S_OFF + soff;
    const unsigned char* ws = smem + PT_W_OFF + wsl * DW_STAGE;
    int q[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) q[mi] = q0[mi] + toffv;
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

  if (ncb > 0) {
    // slots of the planes of the current block: sequence index k sits in slot sl[k % 3] (k = 4, 5: the next block's first two)
    int sl0 = 0, sl1 = 1, sl2 = 2;
    // prologue: planes (1,1) and (0,0) of the first block, W(0) [above], W(1)
    issue_plane(0, MAXNI, 1, 1, cb0, sl0 * a_stage);
    issue_w((uint32_t)t2_tap(1), (uint32_t)cb0, 1);
    issue_plane(0, MAXNI, 0, 0, cb0, sl1 * a_stage);
    write_signs(1, 1, cb0, sl0 * s_stage);
    write_signs(0, 0, cb0, sl1 * s_stage);
    // W(0) and plane (1,1) landed; W(1) and plane (0,0) may be in flight
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(MAXNI + WOPS) : "memory");
    Frag fa;
    const int nvalid_px = min(p.pt_G, p.NB - img0) * min(p.pt_R, p.Ho - row0) * p.Wo;
    const bool mi1_dead = wave * 64 + 32 >= nvalid_px;
    auto kloop = [&](auto mia_tag) __attribute__((always_inline)) {
      constexpr int MIA = decltype(mia_tag)::value;
      auto block = [&](int cbi, bool last) __attribute__((always_inline)) {
        static_for<0, T>([&](auto t_tag) __attribute__((always_inline)) {
          constexpr int s = decltype(t_tag)::value;
          // keep per-tap addresses out of long-lived registers (btx_contract_taps.h) — and the validity bits: hoisted, every
          // (piece, plane) test becomes a wave mask held in an SGPR pair across the loop (28 of them)
          asm volatile("" : "+v"(q0[0]), "+v"(q0[1]), "+v"(pvalid), "+v"(sg_v), "+v"(sg_base[0]), "+v"(sg_base[1]));
          asm volatile("" : "+v"(pp_base[0]), "+v"(pp_base[1]), "+v"(pp_base[2]), "+v"(pp_base[3]), "+v"(pp_base[4]));
          auto slot_of = [&](int k) __attribute__((always_inline)) { return (k % 3) == 0 ? sl0 : ((k % 3) == 1 ? sl1 : sl2); };
          // 1. W(s+2)
          constexpr int s2 = (s + 2) % T, c2 = (s + 2) / T;
          if constexpr (c2 == 0) issue_w((uint32_t)t2_tap(s2), (uint32_t)(cb0 + cbi), (wslot + 2) % WD);
          else { if (!last) issue_w((uint32_t)t2_tap(s2), (uint32_t)(cb0 + cbi + 1), (wslot + 2) % WD); }
          // 2. this stage's share of the plane after next (+ its sign words with its first piece)
          constexpr int NP = t2_np(s);
          if constexpr (NP > 0) {
            constexpr int FS = t2_fseq(s), FK = FS % 4, FC = FS / 4;
            if (FC == 0 || !last) {
              const int so = slot_of(FS);
              issue_plane(t2_p0(s), NP, t2_a(FK), t2_b(FK), cb0 + cbi + FC, so * a_stage);
              if constexpr (t2_p0(s) == 0) write_signs(t2_a(FK), t2_b(FK), cb0 + cbi + FC, so * s_stage);
            }
          }
          // 3. this stage's fragments (no prefetch across the stage boundary: the weight ring is three deep — W(s+1) is only
          //    guaranteed at the end of this stage — and the other block of the CU covers the LDS latency)
          DeltaFrag df;
          load_delta<K2>(df, smem + PT_W_OFF + wslot * DW_STAGE, l31, h);
          {
            constexpr int TP = t2_tap(s);
            const int so = slot_of(t2_seq(s));
            load_frag(fa, so * a_stage, so * s_stage, ((TP / 3) == 2 ? Wp : 0) + ((TP % 3) == 2 ? 1 : 0), wslot, mia_tag);
          }
          // 4. multiply
          stage_mma<PREC, K2, MI, MIA, false, KIND == 1>(fa, df, accm, accd, l31, h);
          // 5. everything issued before this stage has landed (W(s+1), and any plane that starts at s+1); meet the others
          if (!last) end_stage<WOPS + NP>();
          else end_stage<((s + 2 < T) ? WOPS : 0) + ((t2_fseq(s) < 4) ? NP : 0)>();
          wslot = (wslot + 1) % WD;
        });
        // the next block's planes: sequence 4 -> slot of (k % 3 == 1), i.e. rotate by one
        const int t0 = sl0; sl0 = sl1; sl1 = sl2; sl2 = t0;
      };
      for (int cbi = 0; cbi + 1 < ncb; ++cbi) block(cbi, false);
      block(ncb - 1, true);
    };
    if (mi1_dead) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 2>{});
  }

  // =================== epilogue (btx_epilogue.h) ============================================================
  {
    BTX_SECTION_PARAMS(pe, logical2);
    const int nimg = min(pe.pt_G, pe.NB - img0), nrow = min(pe.pt_R, pe.Ho - row0);
    const int nvalid = nimg * nrow * pe.Wo;
    const uint32_t m0 = (uint32_t)(img0 * pe.Ho + row0) * (uint32_t)pe.Wo;
    if constexpr (WIDE) {  // as in contract_taps_kernel: both tiles' constants first, one workgroup barrier
      float* const ba0 = (float*)(smem + NW * PT_EP_WAVE);
      float* const ba1 = ba0 + 4 * BN;
      const bool has_bias = (split == 0) && (pe.mu_b != nullptr);
      const bool has_aff = (pe.ksplits == 1) && ((pe.ep_scale != nullptr) || (pe.ep_shift != nullptr));
      if (has_bias || has_aff) {
        if (tid < 64) ep_fill_constants<0>(pe, rl, ba0, tid, 2 * ntile, group, has_bias, has_aff);
        else if (tid < 128) ep_fill_constants<0>(pe, rl, ba1, tid - 64, 2 * ntile + 1, group, has_bias, has_aff);
      }
      __syncthreads();
      staged_epilogue<0, NW>(pe, rl, accm, accm, smem, tid, wave, lane, 2 * ntile, group, split, m0, nvalid, nullptr, -1, true, ba0);
      staged_epilogue<0, NW>(pe, rl, accd, accd, smem, tid, wave, lane, 2 * ntile + 1, group, split, m0, nvalid, nullptr, -1, true, ba1);
    } else
    staged_epilogue<KIND, NW>(pe, rl, accm, accd, smem, tid, wave, lane, ntile, group, split, m0, nvalid);
  }
}

}  // namespace btx
