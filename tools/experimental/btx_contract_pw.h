// btx_contract_pw.h — pointwise contractions (Linear, 1x1x1 convolutions at stride 1 without padding: 33 of the 53
// convolutions of a ResNet50 — reference models/deterministic/resnet_large.py:85-105 — and every Linear layer,
// layers/flipout_layers/linear_flipout.py:168-174) as a Flipout-GEMM with the n-tile loop INSIDE the workgroup (gfx950).
//
// contract_dma_kernel (btx_contract_dma.h) gives every (pixel tile, n-tile) pair its own workgroup: for the "expand"
// convolutions of a bottleneck (K = 64 .. 256, N = 4 K) a workgroup then is a ~7k-cycle prologue, 2 .. 8 K-stages and a
// ~6k-cycle store side staged through LDS, and the activation tile is fetched N / 64 times.  Here a workgroup owns a pixel
// tile (256 pixels, 4 waves x 64) and walks `pt_R` n-tiles with it:
//
//   * the (stage, n-tile) pairs form ONE linear sequence: weight tiles arrive by LDS-DMA two stages ahead across n-tile
//     boundaries, so the matrix pipe never sees a prologue after the first one;
//   * K <= 3 stages (96 bf16 / 48 f32 channels: the 64 -> 256 and 64 -> 64 convolutions): the activation stages and their
//     s_in words are fetched / hashed ONCE and stay in the three ring slots for all n-tiles — x is read once from HBM, not
//     N / 64 times; larger K streams them again per n-tile (out of L2: the same workgroup read them a few microseconds ago);
//   * no geometry decode (output pixel m reads input pixel m), no tap masks;
//   * the store side of n-tile t runs from the accumulator registers (direct_epilogue, btx_epilogue.h: no staging area, so
//     the rings stay live) at the top of the first stage of n-tile t + 1, after which that stage's MFMAs restart the
//     accumulators; its stores drain under the following stages.  The tile's per-channel constants are written by wave 0
//     during the tile's last stage, behind that stage's barrier (two buffers, by n-tile parity).
//
// Bit-identical to contract_dma_kernel on the same launch (same K order per output element, same noise indices).
// Host-checked (btx_api.hip): pointwise geometry, one K split, output dtype == activation dtype, N/groups % 64 == 0,
// N % 32 == 0, hashed s_out, M * N * sizeof(out) < 0x7ff00000.  ContractParams: pt_R = n-tiles per workgroup, pt_rtiles =
// n-tile chunks per (pixel tile, group).
#pragma once
#include <type_traits>
#include "btx_contract.h"
#include "btx_contract_dma.h"
#include "btx_epilogue.h"
#include "btx_mma.h"
#include "btx_presample.h"

namespace btx {

constexpr int PW_DA = 3;  // activation / sign ring slots = the stages that can stay resident
constexpr int PW_DW = 3;  // weight ring: W(j + 2) is fetched while stage j multiplies
struct PwLds {
  static constexpr int TP = 256;
  static constexpr int A_STAGE = NG * TP * 16;  // 16384
  static constexpr int S_STAGE = TP * 4;
  static constexpr int A_OFF = 0;
  static constexpr int S_OFF = A_OFF + PW_DA * A_STAGE;
  static constexpr int W_OFF = S_OFF + PW_DA * S_STAGE;
  static constexpr int C_OFF = W_OFF + PW_DW * DW_STAGE;  // 2 x [bias mean | bias delta | scale | shift] x 64 floats
  static constexpr int BYTES = C_OFF + 2 * 1024;
};
static_assert(PwLds::BYTES <= 81920, "two workgroups per CU");

// the lane id from the hardware (two VALU instructions), for code that wants it without keeping a register alive: inside the
// stage loop every VGPR that is not an accumulator or a fragment is a candidate for scratch — and a scratch reload is a VMEM
// load whose wait (vmcnt(0)) drains the LDS-DMA ring.  volatile: not hoisted, not merged.
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <int PREC, int KIND>
__global__ __launch_bounds__(256, 2) void contract_pw_kernel(const ContractParams) {
  BTX_SECTION_PARAMS(p, logical);
  using LD = PwLds;
  using ACT = typename std::conditional<PREC == 1, __bf16, float>::type;
  constexpr int G = (PREC == 1) ? 8 : 4;
  constexpr int BK = NG * G;
  constexpr int TP = LD::TP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const RngLive rl = rng_live<KIND>(p);

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool upper = (wave & 1) != 0;

  // workgroup -> (pixel tile, group, n-tile chunk): the chunks of a pixel tile are neighbours (one XCD's L2 holds the tile)
  uint32_t u_mtile, u_rem, u_group, u_chunk;
  fdivmod((uint32_t)logical, p.fd_inner, (uint32_t)(p.pt_rtiles * p.groups), u_mtile, u_rem);
  fdivmod(u_rem, p.fd_rtiles, (uint32_t)p.pt_rtiles, u_group, u_chunk);
  const int mtile = (int)u_mtile, group = (int)u_group;
  const int nt0 = (int)u_chunk * p.pt_R;
  const int nnt = min(p.pt_R, p.ntiles - nt0);  // n-tiles of this workgroup
  const int nstages = p.K / BK;                 // host: K % BK == 0
  const bool resident = nstages <= PW_DA;       // the activation stages are fetched once
  const int J = nnt * nstages;                  // (n-tile, stage) pairs, n-tile major

  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

  // ---- weight loader: wave w fetches row w of the stage's mu tile (+ row w of its delta tile), 1 KiB each
  const uint32_t w_tile_bytes = (uint32_t)(p.K / G) * 1024u;
  const int w_lds = LD::W_OFF + wave * 1024;
  constexpr int WOPS = (KIND == 1) ? 2 : 1;
  // issue state of the pair being requested: its n-tile / stage and the ring slot it lands in
  int wi_nt = 0, wi_s = 0, wi_slot = 0;
  auto issue_w = [&]() __attribute__((always_inline)) {
    const int lane_w = lane_now();
    const uint32_t go = (uint32_t)(group * p.ntiles + nt0 + wi_nt) * w_tile_bytes + (uint32_t)wi_s * (uint32_t)NG * 1024u +
                        (uint32_t)lane_w * 16u + (uint32_t)wave * 1024u;
    unsigned char* ld = smem + w_lds + wi_slot * DW_STAGE;
    dma16(wt_rsrc, go, ld);
    if constexpr (KIND == 1) dma16(wt_rsrc, go + p.wt_delta_off, ld + 4096);
    if (++wi_s == nstages) { wi_s = 0; ++wi_nt; }
    wi_slot = (wi_slot == PW_DW - 1) ? 0 : wi_slot + 1;
  };
  if (J > 0) issue_w();
  if (J > 1) issue_w();

  // ---- activation loader: DMA instruction q of wave w moves pixels 64w + 16q + (lane >> 2), granule slot lane & 3
  //      (source-side swizzle as in btx_contract_dma.h); out-of-range pixels read zeros.  The thread also hashes the s_in word
  //      of pixel `tid` for the stage.  Nothing per-lane is kept between calls (the offsets are a multiply and two adds
  //      away from the lane id; as loop-carried registers they sat beside 128 accumulators and went to scratch).
  int ai_s = 0, ai_slot = 0;  // next activation stage to request and its ring slot
  auto issue_acts = [&]() __attribute__((always_inline)) {
    const int lane_a = lane_now();
    const int g_lane = (lane_a & 3) ^ ((lane_a >> 4) & 3);
    const uint32_t so = (uint32_t)(group * p.Cg + ai_s * BK);
    unsigned char* as = smem + LD::A_OFF + ai_slot * LD::A_STAGE + wave * 4096;
    const int m0 = mtile * TP + wave * 64 + (lane_a >> 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mq = m0 + q * 16;
      const uint32_t bo = ((uint32_t)mq * (uint32_t)p.C + so + (uint32_t)(G * g_lane)) * (uint32_t)sizeof(ACT);
      dma16(x_rsrc, mq < p.M ? bo : DMA_OOB, as + q * 1024);
    }
    if constexpr (KIND == 1) {
      const int t_ = wave * 64 + lane_a;
      const int m_own = mtile * TP + t_;
      const uint32_t off = (uint32_t)(m_own < p.M ? m_own : 0) * (uint32_t)p.C + so;
      uint32_t w = p.sign_in ? sign_word_explicit(p.sign_in, off, p.x_bytes / (uint32_t)sizeof(ACT))
                             : btx_sign_word(off >> 5, rl.kin_a, rl.kin_b);
      if constexpr (G == 4) w <<= 8 * ((off >> 4) & 1);
      *(uint32_t*)(smem + LD::S_OFF + ai_slot * LD::S_STAGE + t_ * 4) = w;
    }
    if (++ai_s == nstages) ai_s = 0;
    ai_slot = (ai_slot == PW_DA - 1) ? 0 : ai_slot + 1;
  };
  // how many activation stages are requested in all: once each when they stay resident, else once per (n-tile, stage)
  const int a_total = resident ? nstages : J;
  int a_issued = 0;
  if (a_total > 0) { issue_acts(); ++a_issued; }
  if (a_total > 1) { issue_acts(); ++a_issued; }

  f32x16 accm[2][2], accd[2][2];
  auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }
  };
  if (PREC != 1 || J == 0) clear_acc();

  const bool has_bias = p.mu_b != nullptr;
  const bool has_aff = (p.ep_scale != nullptr) || (p.ep_shift != nullptr);

  auto mma_stage = [&](int a_slot, int w_slot, auto zero_tag) __attribute__((always_inline)) {
    constexpr bool ZERO = decltype(zero_tag)::value;
    const int lane_m = lane_now();  // (fragment addresses recomputed per stage: see the store side)
    const int l31 = lane_m & 31, h = lane_m >> 5;
    const unsigned char* as = smem + LD::A_OFF + a_slot * LD::A_STAGE;
    const unsigned char* ss = smem + LD::S_OFF + a_slot * LD::S_STAGE;
    const unsigned char* ws = smem + LD::W_OFF + w_slot * DW_STAGE;
    StageFrag f;
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
    if constexpr (ZERO && PREC == 1) stage_mma<PREC, KIND, 2, 2, true>(f, dfrag, accm, accd, l31, h);
    else stage_mma<PREC, KIND>(f, dfrag, accm, accd, l31, h);
  };

  if (J > 0) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int s = 0, nt = 0;               // the pair being multiplied
    int a_slot = 0, w_slot = 0;      // its ring slots
    for (int j = 0;; ++j) {  // J stages, plus one pass for the store side of the last n-tile
      // 1. the store side of the previous n-tile, from the accumulators this stage is about to restart (ONE call site: with
      //    a second copy behind the loop the allocator spills)
      if (s == 0 && nt > 0) {
        // opaque copies of the thread's ids: the store side's lane-dependent values (output offsets, sign words, constant
        // addresses) are recomputed here instead of being hoisted out of the loop and kept alive beside the accumulators
        const int lane_o = lane_now();
        const int tid_o = wave * 64 + lane_o;
        uint32_t gp[2];
        bool gok[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int m = mtile * TP + wave * 64 + mi * 32 + (lane_o & 31);
          gok[mi] = m < p.M;
          gp[mi] = (uint32_t)(gok[mi] ? m : 0);
        }
        float* ba = (float*)(smem + LD::C_OFF + ((nt - 1) & 1) * 1024);
        direct_epilogue<KIND, ACT, true>(p, rl, accm, accd, ba, tid_o, lane_o, nt0 + nt - 1, group, gp, gok);
        if constexpr (PREC != 1) clear_acc();
      }
      if (j == J) break;
      // 2. requests: W(j + 2), the activation stage two ahead (unless it is resident already) — waves 0 and 2 in front of
      //    their MFMAs, waves 1 and 3 behind them: DMA instructions block at issue while the memory pipeline is full, and of
      //    the waves that share a SIMD (one of each workgroup of the CU) one should be free to multiply meanwhile
      int nops = 0;
      const bool w_now = j + 2 < J, a_now = a_issued < a_total;
      if (!upper) {
        if (w_now) issue_w();
        if (a_now) issue_acts();
      }
      // 3. multiply
      if (s == 0) mma_stage(a_slot, w_slot, std::true_type{});
      else mma_stage(a_slot, w_slot, std::false_type{});
      if (upper) {
        if (w_now) issue_w();
        if (a_now) issue_acts();
      }
      if (w_now) nops += WOPS;
      if (a_now) { ++a_issued; nops += 4; }
      // 4. last stage of an n-tile: wave 0 leaves the tile's constants for the store side (visible behind the barrier)
      if (s == nstages - 1) {
        const int lane_c = lane_now();
        ep_fill_constants<KIND>(p, rl, (float*)(smem + LD::C_OFF + (nt & 1) * 1024), wave * 64 + lane_c, nt0 + nt, group, has_bias,
                                has_aff);
      }
      // 5. everything requested before this iteration has landed; meet
      wait_vmcnt(nops);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (++s == nstages) { s = 0; ++nt; }
      a_slot = resident ? s : ((a_slot == PW_DA - 1) ? 0 : a_slot + 1);
      w_slot = (w_slot == PW_DW - 1) ? 0 : w_slot + 1;
    }
  }
}

template <int PREC>
static int launch_contract_pw_impl(int kind, const ContractParams& p, int nwg, hipStream_t st) {
#define BTX_LAUNCH_PW(KIND)                                                                                             \
  do {                                                                                                                  \
    auto kfn = contract_pw_kernel<PREC, KIND>;                                                                          \
    static bool attr_done = false;                                                                                      \
    if (!attr_done) {                                                                                                   \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, PwLds::BYTES);   \
      if (e != hipSuccess) return (int)e;                                                                               \
      attr_done = true;                                                                                                 \
    }                                                                                                                   \
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(256), PwLds::BYTES, st, p);                                                 \
  } while (0)
  int rc = launch_presample_impl<PREC>(kind, p, st);
  if (rc) return rc;
  if (kind == 0) BTX_LAUNCH_PW(0); else BTX_LAUNCH_PW(1);
#undef BTX_LAUNCH_PW
  return (int)hipGetLastError();
}

}  // namespace btx
