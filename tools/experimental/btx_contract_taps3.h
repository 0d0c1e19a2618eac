// btx_contract_taps3.h — persistent form of the tap-unrolled 3x3 kernel (btx_contract_taps.h), bf16, gfx950.
//
// Same tiles, same LDS rings, same K loop, same noise indices and the same f32 operation order per output element as
// contract_taps_kernel<bf16, KIND, 3, 3, 1> — bit-identical results.  What changes is everything AROUND the K loop, which
// on the ResNet 56x56 layers (18 K-stages per tile) was 40 % of a workgroup's life (phase timers, round 3: prologue 10 k,
// K loop 24 k, store side 8 k cycles):
//
//   * a workgroup is persistent: the grid is about two workgroups per CU; a workgroup owns ONE (row tile, n-tile, group)
//     position and walks it through image groups (of all MC sample lanes) drawn from the position's queue, one atomic
//     per tile, a tile ahead.  Its next tile is therefore the same patch some images further on: every per-thread offset
//     of the tile advances by one scalar, and nothing has to be decoded between tiles.  The K loop does not stop at a tile boundary: the last channel block of a tile fetches —
//     stage by stage, with the static DMA schedule of any other block — the first patch, sign words and weight tiles of
//     the NEXT tile.  No prologue after the first tile.
//   * the store side needs no LDS (the rings hold the next tile's data by then): bias, Flipout combine, eval-BN affine in
//     the MFMA fragment registers, then v_permlane32_swap pairs the half-waves' 4-channel runs into 8-channel runs
//     (cdna_hip_programming.md T21), the residual is read and the result written as 16-byte pieces, two per pixel and
//     instruction (32 contiguous bytes).  The per-channel constants go through the sign-word slot the finished tile no
//     longer needs.
//
// Eligibility (btx_api.hip): 3x3 stride 1, plain (whole-row) tiles whose image count divides the batch, bf16 activations
// and output, one K split, an even number of channel blocks (a tile then always starts on patch slot 0), whole
// 64-channel n-tiles with 32-aligned s_out words, generated noise, no bias.
// Compiled in measurement builds only (-DBTX_TUNING / -DBTX_PT_TRACE, BTX_PERSIST=1 selects it): it needs 29k instead of
// 35k cycles per 56x56 tile and takes the same time per launch — the chip runs it at a lower clock (DESIGN.md section 5,
// round 3; profiles/r03_persistent_ab.txt, r03_phase_timers_sustained.txt, r03_power_probe.txt).  What it took to make hipcc
// allocate it (241 VGPRs, no scratch) is in the comments below: parameters through an address_space(4) kernarg pointer,
// tile scalars pinned to SGPRs, ONE instantiation of the store side.
#pragma once
#include "btx_contract_taps.h"

namespace btx {

struct Taps3Tile {  // wave-uniform description of one tile of the launch
  int lane, ntile, group, img0, row0, col0;
};

template <int KIND>
__global__ __launch_bounds__(256, 2) void contract_taps3_kernel(const ContractParams p) {
  constexpr int PREC = 1, NW = 4, NT = 256, MI = 2, T = 9, KW = 3;
  constexpr int MAXNI = TP_MAXNI;
  constexpr int PST = T - 3;
  constexpr int WOPS = (KIND == 1) ? 2 : 1;
  constexpr int G = 8, BK = NG * G, ESZ = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nlanes = p.lanes > 1 ? p.lanes : 1;
#ifdef BTX_PT_TRACE
  // phase timers (tools/gpu_diag.py trace): prologue | sum of the K loops | sum of the store sides, split in five
  const uint32_t tr_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
  const uint32_t tr_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
  uint32_t tr_pro = 0, tr_k = 0, tr_st[5] = {0, 0, 0, 0, 0}, tr_n = 0, tr_mark = 0;
#define BTX_T3_MARK(acc) do { __builtin_amdgcn_sched_barrier(0); const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime(); acc += t_ - tr_mark; tr_mark = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define BTX_T3_MARK(acc) do { } while (0)
#endif
  // The per-tile code (tile decode, patch / sign offsets of the next tile, the store side) reads the launch parameters
  // through `kp`, a pointer to the kernel-argument segment that is made opaque at the head of each of those sections:
  // the fields are then s_load-ed where they are used instead of living in SGPRs across the K loop (the by-value struct
  // is ~180 dwords: kept in registers it spilt ~500 SGPRs into VGPR lanes, and those VGPRs into scratch).
  // wave-uniform values that change from tile to tile are pinned to SGPRs: left to itself the compiler carries them in
  // VGPRs (and wraps every weight DMA, whose scalar offset they feed, in a v_readfirstlane waterfall loop)
  auto U = [](uint32_t v) __attribute__((always_inline)) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  using KP = const __attribute__((address_space(4))) ContractParams*;  // constant address space: the reads are s_loads
  KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  auto refresh_q = [&]() __attribute__((always_inline)) {
    kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
  };
  refresh_q();
  // the same pointer as a generic one, for callees that take references (the address space is inferred back after inlining)
  auto KPG = [&]() __attribute__((always_inline)) -> const ContractParams* { return (const ContractParams*)kp; };

  // ---- this workgroup's position and its range of image groups (an image group = the pt_G images of one tile)
  const uint32_t IG = (uint32_t)(p.NB / p.pt_G);             // image groups per lane (host: NB % pt_G == 0)
  const uint32_t combos = (uint32_t)(p.pt_rtiles * p.ntiles * p.groups);
  // The workgroups of a position share a queue of image groups: group `seg` is a workgroup's first tile, the following
  // ones are drawn from the position's counter (pt_queue, zeroed by the host before the launch), one per tile, a tile
  // ahead.  (Fixed ranges per workgroup: the slowest of the 512 workgroups took 1.24x the mean — kernel time = its time.)
  uint32_t ig_cur, ig_end, q_first, q_combo;
  Taps3Tile cur;
  {
    const uint32_t b = (uint32_t)xcd_logical(), nseg = gridDim.x / combos, seg = b / combos, combo = b - seg * combos;
    const uint32_t rt = combo % (uint32_t)p.pt_rtiles, rest = combo / (uint32_t)p.pt_rtiles;
    const uint32_t igt = (uint32_t)nlanes * IG;
    ig_cur = U(seg);      // host: nseg <= igt
    ig_end = U(igt);
    q_first = U(nseg); q_combo = U(combo);
    cur.ntile = (int)(rest % (uint32_t)p.ntiles); cur.group = (int)(rest / (uint32_t)p.ntiles);
    cur.row0 = (int)rt * p.pt_R; cur.col0 = 0;
    cur.lane = (int)U(ig_cur / IG); cur.img0 = (int)U((ig_cur % IG) * (uint32_t)p.pt_G);
    cur.ntile = (int)U((uint32_t)cur.ntile); cur.group = (int)U((uint32_t)cur.group); cur.row0 = (int)U((uint32_t)cur.row0);
  }
  if (ig_cur >= ig_end) return;
  const uint32_t img_elems = U((uint32_t)(p.H * p.W * p.C));   // elements of one input image
  // byte offset in x (all lanes behind one descriptor) / element offset in the lane's own tensor of an image group
  auto xoff_of = [&](int ln, int img0) __attribute__((always_inline)) -> uint32_t {
    return (uint32_t)ln * (uint32_t)p.lane_x + (uint32_t)img0 * img_elems * (uint32_t)ESZ;
  };

  // patch pixel q of tile `tl` -> byte offset of its 16-byte granule `g_lane` in x (all lanes behind one descriptor)
  const int g_lane = (lane & 3) ^ ((lane >> 4) & 3);
  auto patch_pix = [&](const Taps3Tile& tl, int q, bool& ok) __attribute__((always_inline)) -> uint32_t {
    uint32_t ut, upc, ugi, upr;
    fdivmod((uint32_t)q, KPG()->fd_ptWp, (uint32_t)kp->pt_Wp, ut, upc);
    fdivmod(ut, KPG()->fd_ptRp, (uint32_t)kp->pt_Rp, ugi, upr);
    const int img = tl.img0 + (int)ugi, ih = tl.row0 + (int)upr - kp->ph, iw = (int)upc - kp->pw;
    ok = img < kp->NB && (unsigned)ih < (unsigned)kp->H && (unsigned)iw < (unsigned)kp->W;
    return (uint32_t)((img * kp->H + ih) * kp->W + iw);
  };
  auto piece_off = [&](const Taps3Tile& tl, int j) __attribute__((always_inline)) -> uint32_t {
    const int q = 16 * (wave + NW * j) + (lane >> 2);
    uint32_t bo = DMA_OOB;
    if (j < kp->pt_NI && q < kp->pt_PP) {
      bool ok;
      const uint32_t ipix = patch_pix(tl, q, ok);
      if (ok) bo = (ipix * (uint32_t)kp->C + (uint32_t)(tl.group * kp->Cg + G * g_lane)) * (uint32_t)ESZ + (uint32_t)tl.lane * (uint32_t)kp->lane_x;  // ipix counts the images of the lane
    }
    return bo;
  };
  // sign keys of a lane's MC sample
  auto lane_keys = [&](int ln) __attribute__((always_inline)) -> RngLive {
    RngLive r = {kp->sample + (uint32_t)ln, kp->kin_a, kp->kin_b, kp->kout_a, kp->kout_b};
    if (kp->sample_ptr || nlanes > 1) {
      if (kp->sample_ptr) r.sample = __builtin_amdgcn_readfirstlane(kp->sample_ptr[ln]);
      if constexpr (KIND == 1) {
        const uint32_t si = kp->swap_signs ? 3u : 2u, so = kp->swap_signs ? 2u : 3u;
        const BtxPhilox4 ki = btx_philox4x32_10(0u, r.sample, kp->layer, si, kp->seed_lo, kp->seed_hi);
        const BtxPhilox4 ko = btx_philox4x32_10(0u, r.sample, kp->layer, so, kp->seed_lo, kp->seed_hi);
        r.kin_a = __builtin_amdgcn_readfirstlane(ki.x[0]); r.kin_b = __builtin_amdgcn_readfirstlane(ki.x[1]);
        r.kout_a = __builtin_amdgcn_readfirstlane(ko.x[0]); r.kout_b = __builtin_amdgcn_readfirstlane(ko.x[1]);
      }
    }
    return r;
  };

  const int ncb = p.Cg / BK;  // channel blocks of a tile (host: even, one K split)
  const int a_stage = p.pt_astage, s_stage = p.pt_astage >> 4;
  const int PT_A_OFF = 0, PT_S_OFF = 2 * a_stage, PT_W_OFF = 2 * a_stage + 2 * s_stage;
  const int PT_X_OFF = PT_W_OFF + PT_WD * DW_STAGE;

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.x, 0, p.x_bytes + (uint32_t)(nlanes - 1) * (uint32_t)p.lane_x, 0x00020000);
  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);

  // ---- weight loader (as contract_taps_kernel): scalar tile base + stage offset, constant vector part
  const uint32_t w_voff = (uint32_t)lane * 16u + (uint32_t)wave * 1024u;
  const uint32_t CgG = (uint32_t)(p.Cg / G);
  const int w_lds = PT_W_OFF + wave * 1024;
  auto w_base_of = [&](const Taps3Tile& tl, uint32_t& doff) __attribute__((always_inline)) -> uint32_t {
    uint32_t sb = (uint32_t)(tl.group * kp->ntiles + tl.ntile) * (uint32_t)(kp->K / G) * 1024u;
    doff = kp->wt_delta_off;
    if (kp->lane_wt_delta) doff += (uint32_t)tl.lane * (uint32_t)kp->lane_wt;
    else sb += (uint32_t)tl.lane * (uint32_t)kp->lane_wt;
    return sb;
  };
  int wslot = 0;
  auto issue_w = [&](uint32_t sbase, uint32_t doff, uint32_t tap, uint32_t cb, int slot) __attribute__((always_inline)) {
    const uint32_t soff = U(sbase + (tap * CgG + cb * (uint32_t)NG) * 1024u);
    unsigned char* ld = smem + w_lds + slot * DW_STAGE;
    dma16s(wt_rsrc, w_voff, soff, ld);
    if constexpr (KIND == 1) dma16s(wt_rsrc, w_voff, U(soff + doff), ld + 4096);
  };

  // ---- first tile: prologue as in contract_taps_kernel
  uint32_t w_sbase, w_doff;
  w_sbase = w_base_of(cur, w_doff);
  w_sbase = U(w_sbase); w_doff = U(w_doff);
  issue_w(w_sbase, w_doff, 0u, 0u, 0);
  uint32_t pp_boff[MAXNI];
  uint32_t pmask = 0;
#pragma unroll
  for (int j = 0; j < MAXNI; ++j) {
    pp_boff[j] = piece_off(cur, j);
    if (j < p.pt_NI && 16 * (wave + NW * j) < p.pt_PP) {
      pmask |= 1u << j;
      dma16(x_rsrc, pp_boff[j], smem + PT_A_OFF + (wave + NW * j) * 1024);
    }
  }
  pmask = __builtin_amdgcn_readfirstlane(pmask);
  issue_w(w_sbase, w_doff, 1u, 0u, 1);
  issue_w(w_sbase, w_doff, 2u, 0u, 2);
  RngLive rl = lane_keys(cur.lane);

  // ---- sign role: thread t owns the words of patch pixels t and t+256
  auto sign_off = [&](const Taps3Tile& tl, int j) __attribute__((always_inline)) -> uint32_t {
    const int q = tid + NT * j;
    bool inside;
    const uint32_t ipix = patch_pix(tl, q < kp->pt_PP ? q : 0, inside);
    return ipix * (uint32_t)kp->C + (uint32_t)(tl.group * kp->Cg);  // outside pixels hold zeros: any word will do
  };
  uint32_t sg_off[2] = {sign_off(cur, 0), sign_off(cur, 1)};
  const bool sg_ok0 = tid < p.pt_PP, sg_ok1 = tid + NT < p.pt_PP;
  auto write_signs = [&](int slot, int cb, uint32_t o0, uint32_t o1, uint32_t ka, uint32_t kb) __attribute__((always_inline)) {
    if constexpr (KIND == 1) {
      unsigned char* ss = smem + PT_S_OFF + slot * s_stage;
      if (sg_ok0) *(uint32_t*)(ss + tid * 4) = btx_sign_word((o0 + (uint32_t)(cb * BK)) >> 5, ka, kb);
      if (sg_ok1) *(uint32_t*)(ss + (tid + NT) * 4) = btx_sign_word((o1 + (uint32_t)(cb * BK)) >> 5, ka, kb);
    }
  };

  // ---- MFMA role: the lane's two pixels of the tile -> patch pixel index.  The same for every tile (pixels that do not
  // exist in an edge tile multiply whatever the patch holds there and are never stored)
  int q0[MI];
  const int tile_px = p.pt_G * p.pt_R * p.Wo;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int pl = wave * 64 + mi * 32 + l31;
    uint32_t ut, uc, ugi, ur;
    fdivmod((uint32_t)pl, p.fd_Wo, (uint32_t)p.Wo, ut, uc);
    fdivmod(ut, p.fd_ptR, (uint32_t)p.pt_R, ugi, ur);
    q0[mi] = (pl < tile_px) ? ((int)ugi * p.pt_Rp + (int)ur) * p.pt_Wp + (int)uc : 0;
  }
  const int row_step = p.dh * p.pt_Wp;

  f32x16 accm[MI][2], accd[MI][2];  // started by the ZERO form of a tile's first stage (btx_mma.h)

  // Fragment registers.  Activations are double-buffered (the next stage's are read while this stage multiplies; this
  // stage's get their s_in signs in place between the two passes).  The weights are NOT: one set for the mean pass, one
  // for the delta pass — the mean set is refilled with the NEXT stage's tile as soon as this stage's mean MFMAs have
  // consumed it (the reads land while the delta MFMAs run), the delta set at the head of its stage while the mean
  // MFMAs run.  48 registers of weight fragments become 32.
  struct AFrag { u32x4 a[NG / 2][MI]; uint32_t sw[MI]; };
  using WFrag = u32x4[NG / 2][2];
  auto load_a = [&](AFrag& f, int aslot, int toffv) __attribute__((always_inline)) {
    const unsigned char* as = smem + PT_A_OFF + aslot * a_stage;
    const unsigned char* ss = smem + PT_S_OFF + aslot * s_stage;
    int q[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) q[mi] = q0[mi] + toffv;
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      const int row = 2 * kk + h;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) f.a[kk][mi] = *(const u32x4*)(as + q[mi] * 64 + ((row ^ ((q[mi] >> 2) & 3)) * 16));
    }
    if constexpr (KIND == 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) f.sw[mi] = *(const uint32_t*)(ss + q[mi] * 4);
    }
  };
  auto load_w = [&](WFrag& w, int wsl, int part) __attribute__((always_inline)) {  // part 0: mean tile, 1: delta tile
    const unsigned char* ws = smem + PT_W_OFF + wsl * DW_STAGE + part * (NG * BN * 16);
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) w[kk][ni] = *(const u32x4*)(ws + ((2 * kk + h) * BN + ni * 32 + l31) * 16);
  };

  write_signs(0, 0, sg_off[0], sg_off[1], rl.kin_a, rl.kin_b);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(WOPS) : "memory");
  AFrag fa, fb;
  WFrag wm, wd;
  load_a(fa, 0, 0);
  load_w(wm, 0, 0);
  bool after_store = false;
#ifdef BTX_PT_TRACE
  tr_mark = tr_t0;
  BTX_T3_MARK(tr_pro);
#endif

  // One channel block = T unrolled stages, exactly contract_taps_kernel's: stage t multiplies tap t, fetches W three
  // stages ahead and its share of the NEXT block's patch (+ the next block's sign words at t = 0), reads the next
  // stage's fragments.  The next block is described by scalars — weight-tile base, delta offset, channel block, sign keys —
  // and by pp_boff / sg_off: for the last block of a tile those describe block 0 of the next TILE (rewritten just before
  // it starts), so the K loop runs across tile boundaries with one instruction stream.  last: nothing follows.
  auto block = [&](auto par_tag, auto first_tag, int cbi, bool last, bool tile_end, uint32_t nb_sbase, uint32_t nb_doff,
                   uint32_t nb_cb, uint32_t nb_ka, uint32_t nb_kb) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;  // first block of a tile: its first stage starts the accumulators
    static_for<0, T>([&](auto t_tag) __attribute__((always_inline)) {
      constexpr int t = decltype(t_tag)::value;
      constexpr int sp = (PAR * T + t) & 1;
      AFrag& curf = sp ? fb : fa;
      AFrag& nxtf = sp ? fa : fb;
      asm volatile("" : "+v"(q0[0]), "+v"(q0[1]));
      constexpr int t3 = (t + 3) % T, c3 = (t + 3) / T;
      if constexpr (c3 == 0) {
        issue_w(w_sbase, w_doff, (uint32_t)t3, (uint32_t)cbi, (wslot + 3) & 3);
      } else {
        if (!last) issue_w(nb_sbase, nb_doff, (uint32_t)t3, nb_cb, (wslot + 3) & 3);
      }
      constexpr int KP = tp_pieces<T>(t);
      if constexpr (t < PST) {
        if (!last) {
          const uint32_t cboff = nb_cb * (uint32_t)(BK * ESZ);
#pragma unroll
          for (int i = 0; i < KP; ++i) {
            const int j = t + PST * i;
            const uint32_t bo = pp_boff[j];
            unsigned char* dst = ((pmask >> j) & 1u) ? smem + PT_A_OFF + (PAR ^ 1) * a_stage + (wave + NW * j) * 1024
                                                      : smem + PT_X_OFF;
            dma16(x_rsrc, bo == DMA_OOB ? DMA_OOB : bo + cboff, dst);
          }
          if constexpr (t == 0) write_signs(PAR ^ 1, (int)nb_cb, sg_off[0], sg_off[1], nb_ka, nb_kb);
        }
      }
      // 3. this stage's delta weights and the next stage's activations (their latency hides behind the mean MFMAs)
      if constexpr (KIND == 1) load_w(wd, wslot, 1);
      constexpr int t1 = (t + 1) % T, c1 = (t + 1) / T;
      // (the last stage of a tile leaves the next tile's first fragments to the code behind the store side)
      if constexpr (c1 == 1) { if (!tile_end) load_a(nxtf, PAR ^ c1, 0); }
      else load_a(nxtf, PAR ^ c1, (t1 / KW) * row_step + (t1 % KW) * p.dw);
      // 4. mean pass
      constexpr bool Z = FIRST && t == 0;
      const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NG / 2; ++kk)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            accm[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wm[kk][ni]),
                                                                  __builtin_bit_cast(bf16x8, curf.a[kk][mi]),
                                                                  (Z && kk == 0) ? zc : accm[mi][ni], 0, 0, 0);
      // 5. the mean set is free: the next stage's mean tile goes into it (nothing above may sink below, nothing below
      //    rise above: the reads must not be hoisted over the MFMAs that still read the registers)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (c1 == 1) { if (!tile_end) load_w(wm, (wslot + 1) & 3, 0); }
      else load_w(wm, (wslot + 1) & 3, 0);
      // 6. delta pass on the sign-flipped activations
      if constexpr (KIND == 1) {
#pragma unroll
        for (int kk = 0; kk < NG / 2; ++kk) {
          const int row = 2 * kk + h;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const uint32_t swr = curf.sw[mi] << (4 * row);
#pragma unroll
            for (int d = 0; d < 4; ++d) curf.a[kk][mi][d] ^= ((swr << d) & 0x80008000u);
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              accd[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wd[kk][ni]),
                                                                    __builtin_bit_cast(bf16x8, curf.a[kk][mi]),
                                                                    (Z && kk == 0) ? zc : accd[mi][ni], 0, 0, 0);
        }
      }
      if constexpr (FIRST && t == 0) {
        // First stage behind a store side: everything this stage and the next one read landed before the stores were
        // issued (vmcnt(0) in front of the store side), so the stage's wait may leave the tile's 8 stores in flight
        // instead of sitting out their write acknowledgements; the next stage's wait covers them, a stage later.
        if (after_store) { end_stage<WOPS + KP + 8>(); after_store = false; }
        else end_stage<WOPS + KP>();
      } else {
        if (!last) end_stage<WOPS + KP>();
        else end_stage<(c3 == 0) ? WOPS : 0>();
      }
      wslot = (wslot + 1) & 3;
    });
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  // The next tile = the same position, the next image group: its patch / sign-word offsets are the current ones plus one
  // scalar, its weight tiles and sign keys change only when the image group belongs to another MC sample lane.
  Taps3Tile nxt = cur;
  uint32_t n_sbase = w_sbase, n_doff = w_doff;
  RngLive rl_n = rl;
  uint32_t dx_next = 0, ds_next = 0;  // byte / element advance of the patch offsets to the next tile
  uint32_t ig_next = ig_end;
  auto plan_next = [&]() __attribute__((always_inline)) {  // ig_next < ig_end
    refresh_q();
    const uint32_t ig = ig_next;
    nxt = cur;
    nxt.lane = (int)U(ig / IG); nxt.img0 = (int)U((ig % IG) * (uint32_t)kp->pt_G);
    dx_next = U(xoff_of(nxt.lane, nxt.img0) - xoff_of(cur.lane, cur.img0));
    ds_next = U((uint32_t)(nxt.img0 - cur.img0) * img_elems);
    n_sbase = w_base_of(nxt, n_doff);
    n_sbase = U(n_sbase); n_doff = U(n_doff);
    rl_n = rl;
    if (nxt.lane != cur.lane) rl_n = lane_keys(nxt.lane);
  };
  uint32_t* const q_word = (uint32_t*)(smem + PT_X_OFF + 1024);  // 16 bytes behind the layout of contract_taps_kernel (host)
  for (;;) {
    // draw the next image group (wave 0, one lane), a tile ahead: the reply is back long before block 0 ends
    uint32_t drawn = 0;
    if (wave == 0 && lane == 0) drawn = __hip_atomic_fetch_add(kp->pt_queue + q_combo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    block(P0{}, std::true_type{}, 0, false, false, w_sbase, w_doff, 1u, rl.kin_a, rl.kin_b);
    if (wave == 0 && lane == 0) *q_word = drawn + q_first;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ig_next = U(*(volatile uint32_t*)q_word);
    const bool has_next = ig_next < ig_end;
    if (has_next) plan_next();
    for (int cbi = 1; cbi + 1 < ncb; cbi += 2) {
      block(P1{}, std::false_type{}, cbi, false, false, w_sbase, w_doff, (uint32_t)(cbi + 1), rl.kin_a, rl.kin_b);
      block(P0{}, std::false_type{}, cbi + 1, false, false, w_sbase, w_doff, (uint32_t)(cbi + 2), rl.kin_a, rl.kin_b);
    }
    // ---- the tile's last block: what follows it is block 0 of the next tile (or nothing)
    if (has_next) {  // the current tile fetches no further patch: its offsets become the next tile's
#pragma unroll
      for (int j = 0; j < MAXNI; ++j) pp_boff[j] = (pp_boff[j] == DMA_OOB) ? DMA_OOB : pp_boff[j] + dx_next;
      sg_off[0] += ds_next; sg_off[1] += ds_next;
    }
    block(P1{}, std::false_type{}, ncb - 1, !has_next, true, n_sbase, n_doff, 0u, rl_n.kin_a, rl_n.kin_b);

    // ============ store side of tile `cur`, straight from the fragment registers =======================================
    // (everything in flight — the next tile's W(2) — lands first: the relaxed wait of the next stage relies on it)
    BTX_T3_MARK(tr_k);
    if (has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BTX_T3_MARK(tr_st[0]);  // everything in flight landed
    refresh_q();
    {
      // Opaque copies of the thread's ids: without them the compiler hoists the store side's address arithmetic (loop-
      // invariant over tiles) out of the tile loop and keeps ~20 VGPRs alive across the K loop, which then spills.
      int lane_o = lane, wave_o = wave, tid_o = tid;
      asm volatile("" : "+v"(lane_o), "+s"(wave_o), "+v"(tid_o));
      const int l31 = lane_o & 31, h = lane_o >> 5, wave = wave_o, tid = tid_o;
      const bool has_bias = kp->mu_b != nullptr;
      const bool has_aff = (kp->ep_scale != nullptr) || (kp->ep_shift != nullptr);
      float* ba_lds = (float*)(smem + PT_S_OFF + s_stage);  // the sign slot of the tile's last block: dead by now
      // per-channel constants [bias mean | bias delta | scale | shift] x 64, identities where absent
      ep_fill_constants<KIND>(*KPG(), rl, ba_lds, tid, cur.ntile, cur.group, has_bias, has_aff);
      uint32_t gp[MI];
      bool gok[MI];
      {
        const uint32_t m0 = (uint32_t)(cur.img0 * kp->Ho + cur.row0) * (uint32_t)kp->Wo;
        const int nvalid = min(kp->pt_G, kp->NB - cur.img0) * min(kp->pt_R, kp->Ho - cur.row0) * kp->Wo;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int pl = wave * 64 + mi * 32 + l31;
          gok[mi] = pl < nvalid;
          gp[mi] = m0 + (uint32_t)(gok[mi] ? pl : 0);
        }
      }
      const uint32_t cbase = (uint32_t)(cur.group * kp->Ng + cur.ntile * BN);
      // output / residual rows behind buffer descriptors (scalar base of the MC sample lane's tensor, 32-bit byte offsets,
      // the (ni, k) part in the instruction's immediate): 64-bit per-store addresses were two VGPRs each and got spilt.
      // A pixel outside the tile gets an out-of-range offset: its loads return zeros, its stores are dropped.
      const uint32_t out_bytes = (uint32_t)kp->NB * (uint32_t)(kp->Ho * kp->Wo) * (uint32_t)kp->N * 2u;  // host: < 2 GiB
      const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (unsigned char*)kp->out + (size_t)cur.lane * (size_t)kp->lane_out, 0, out_bytes, 0x00020000);
      uint32_t eo[MI];  // byte offset of the lane's 8-channel run of (pixel mi, half 0, pair 0)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)  // (0x80000000, not DMA_OOB: the (ni, k) immediate is added to it and must not wrap)
        eo[mi] = gok[mi] ? (gp[mi] * (uint32_t)kp->N + cbase + 8u * (uint32_t)h) * 2u : 0x80000000u;
      // the residual loads of the lane go out first: (pixel mi) x (32-channel half ni) x (16-channel pair k); pixel 1's
      // follow while pixel 0 is being stored (16 registers at a time)
      const bool res = kp->ep_res != nullptr;
      const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (unsigned char*)kp->ep_res + (size_t)cur.lane * (size_t)kp->lane_res, 0, res ? out_bytes : 0u, 0x00020000);
      auto load_res = [&](u32x4 (&r)[2][2], int mi) __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int k = 0; k < 2; ++k)  // (no residual: a zero-length descriptor, the loads return zeros)
            r[ni][k] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, eo[mi] + (uint32_t)(ni * 64 + 32 * k), 0, 0);
      };
      u32x4 rv[2][2];
      load_res(rv, 0);
      uint32_t wsh[MI][2];
      if constexpr (KIND == 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const uint32_t orow = gp[mi] * (uint32_t)kp->N + cbase;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) wsh[mi][ni] = btx_sign_word((orow + 32u * ni) >> 5, rl.kout_a, rl.kout_b) << (2 * h);
        }
      }
      uint32_t SB = 0x80000000u;
      asm volatile("" : "+s"(SB));
      const float lowb = kp->ep_relu ? 0.f : -__builtin_inff();  // ReLU as a lower bound: one instruction stream for both
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the constants are in LDS
      BTX_T3_MARK(tr_st[1]);  // addresses, residual loads issued, constants in LDS, waves met
      // Phase A — fold the delta accumulators into the mean ones, in place: (mean + bias) + s_out * (delta + bias delta).
      // Needs nothing but the sign words; afterwards 64 of the 128 accumulator registers are free, which is what lets
      // phase B run without spills.
      auto fold = [&](auto bias_tag) __attribute__((always_inline)) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        if constexpr (KIND == 1 || BIAS) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 bm, bd;
              if constexpr (BIAS) {
                const int cl = ni * 32 + 8 * q + 4 * h;
                bm = *(const f32x4*)(ba_lds + cl); bd = *(const f32x4*)(ba_lds + BN + cl);
              }
#pragma unroll
              for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  float val = accm[mi][ni][4 * q + rr];
                  if constexpr (BIAS) val += bm[rr];
                  if constexpr (KIND == 1) {
                    float dl = accd[mi][ni][4 * q + rr];
                    if constexpr (BIAS) dl += bd[rr];
                    const int sft = 31 - (((rr & 1) ? 31 : 15) - 4 * q - (rr >> 1));
                    val += u2f(__builtin_amdgcn_bitop3_b32(f2u(dl), wsh[mi][ni] << sft, SB, 0x78));
                  }
                  accm[mi][ni][4 * q + rr] = val;
                }
              if constexpr (BIAS) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      // (one instantiation only — the host does not take this kernel for layers with a bias: with both forms behind a
      // run-time branch the compiler hoists their 64 common `sign word << shift` values above it and spills them)
      fold(std::false_type{});
      BTX_T3_MARK(tr_st[2]);
      // Phase B — one (pixel mi, 32-channel half ni, 16-channel pair k) group at a time, registers -> store: scale/shift
      // on the lane's two 4-channel runs, the half-waves' runs paired into one 8-channel run per lane
      // (v_permlane32_swap), residual, ReLU, round, one 16-byte store.  The constants of the next group are requested
      // before this group's arithmetic.
      {
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
        u32x4 rnx[2][2];
        static_for<0, 8>([&](auto g_tag) __attribute__((always_inline)) {
          constexpr int g = decltype(g_tag)::value;
          constexpr int mi = g >> 2, ni = (g >> 1) & 1, k = g & 1;
          if constexpr (g + 1 < 8) load_cst(cst[(g + 1) & 1], ((g + 1) >> 1) & 1, (g + 1) & 1);
          if constexpr (g == 0 && MI > 1) load_res(rnx, 1);  // pixel 1's residual rows land while pixel 0 is stored
          const Cst& c = cst[g & 1];
          float t[2][4];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
              t[hf][rr] = __builtin_fmaf(accm[mi][ni][4 * (2 * k + hf) + rr], c.sc[hf][rr], c.sh[hf][rr]);
          float v[8];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const auto r = __builtin_amdgcn_permlane32_swap(f2u(t[0][rr]), f2u(t[1][rr]), false, false);
            v[rr] = u2f(r[0]);
            v[4 + rr] = u2f(r[1]);
          }
          const u32x4 rw = (mi == 0) ? rv[ni][k] : rnx[ni][k];
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[2 * j] += u2f(rw[j] << 16); v[2 * j + 1] += u2f(rw[j] & 0xffff0000u); }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], lowb);
          const f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
          const u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(x0, bf16x4));
          const u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(x1, bf16x4));
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){p0[0], p0[1], p1[0], p1[1]}, out_rsrc,
                                                 eo[mi] + (uint32_t)(ni * 64 + 32 * k), 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    }
    BTX_T3_MARK(tr_st[3]);  // pairing, residual, stores issued
#ifdef BTX_PT_TRACE
    ++tr_n;
#endif
    if (!has_next) break;
    // ============ next tile: its first block's data is in the rings, its first fragments in the registers ==============
    ig_cur = ig_next;
    cur = nxt;
    w_sbase = n_sbase; w_doff = n_doff;
    rl = rl_n;
    after_store = !(p.pt_tune & 2);  // pt_tune bit 1 (tuning builds): the plain wait immediate behind a store side
    // every wave is through with the constants before the next tile's first stage rewrites that sign slot
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    load_a(fa, 0, 0);  // the new tile's first fragments (its block 0 sits on patch slot 0: ncb is even)
    load_w(wm, wslot, 0);
    BTX_T3_MARK(tr_st[4]);  // next tile planned, waves met, first fragments requested
  }
#ifdef BTX_PT_TRACE
  if (p.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t tr_t3 = (uint32_t)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
      uint32_t* tr = (uint32_t*)p.trace + (size_t)(blockIdx.x * NW + wave) * 8;
      if (p.pt_tune & 128) { tr[0] = tr_st[0]; tr[1] = tr_st[1]; tr[2] = tr_st[2]; tr[3] = tr_st[3]; tr[4] = tr_st[4]; tr[5] = tr_n; }
      else { tr[0] = tr_pro; tr[1] = tr_k; tr[2] = (uint32_t)__builtin_amdgcn_s_memrealtime() - tr_r0; tr[3] = 0;
             tr[4] = tr_st[0] + tr_st[1] + tr_st[2] + tr_st[3] + tr_st[4]; tr[5] = tr_t3 - tr_t0; }
      tr[6] = tr_t0; tr[7] = tr_n;
    }
  }
#endif
}

}  // namespace btx
