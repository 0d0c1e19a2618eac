// btx_contract_stem.h — small-C ("stem") variant of the fused sample-and-contract kernel (gfx950).
//
// Row-fused geometry (BTX_FLAG_ROWFUSE, include/btx.h): the input is [NB][H][W][C] with C <= 8 channels and the conv
// padding materialised, one kernel row = KW*C contiguous elements = a whole number of K-stages.  The LDS-DMA kernel
// re-fetches 64 bytes per output pixel and stage although neighbouring windows overlap almost completely (a 7x7
// stride-2 stem reads every input element ~12 times through the DMA path).  Here the workgroup owns R output rows x
// the full output width of one image and
//   * copies the input rows those outputs need — ONE contiguous byte range of x — into LDS once (linear LDS-DMA);
//   * reads every MFMA activation fragment straight out of that image: the 16-byte granule of output pixel (r, c),
//     kernel row kh, granule g sits at ((r*sh + kh)*W + c*sw)*C + g*G elements — consecutive pixels are 16 (or 32)
//     bytes apart, so the ds_read_b128 of a wave are bank-conflict free without any swizzle;
//   * hashes the s_in words of the byte range once; a pixel-stage's 32 signs are composed from two of them.
// Weights: pre-sampled tiles by LDS-DMA into a 4-slot ring (btx_presample.h); epilogue: btx_epilogue.h.
#pragma once
#include <type_traits>
#include "btx_contract.h"
#include "btx_contract_dma.h"
#include "btx_epilogue.h"
#include "btx_presample.h"
#include "btx_mma.h"

namespace btx {

// ContractParams fields used: pt_R (output rows per tile), pt_Rp (input rows of the patch), pt_rtiles, pt_PP (patch
// bytes), pt_astage (patch bytes rounded to 1 KiB), st_sbytes (bytes of the sign-word array), pt_nw, pt_lds.
// Geometry as the C-ABI hands it over with BTX_FLAG_ROWFUSE: p.Cg = KW*C elements per kernel row, pad 0, groups 1.
template <int PREC, int KIND, int NW>
__global__ __launch_bounds__(64 * NW, 2) void contract_stem_kernel(const ContractParams) {
  BTX_SECTION_PARAMS(p, logical);  // prologue + K loop; the store side has its own view (btx_contract.h)
  constexpr int NT = 64 * NW;
  using ACT = typename std::conditional<PREC == 1, __bf16, float>::type;
  constexpr int G = (PREC == 1) ? 8 : 4;
  constexpr int BK = NG * G;
  constexpr int WD = PT_WD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const RngLive rl = rng_live<KIND>(p);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  uint32_t u_mtile, u_ntile, u_img, u_rt;
  fdivmod((uint32_t)logical, p.fd_ntiles, (uint32_t)p.ntiles, u_mtile, u_ntile);
  fdivmod(u_mtile, p.fd_rtiles, (uint32_t)p.pt_rtiles, u_img, u_rt);
  const int ntile = (int)u_ntile, img = (int)u_img, row0 = (int)u_rt * p.pt_R;

  const int esz = (int)sizeof(ACT);
  const int RowE = p.W * p.C;                           // elements per input row
  const int nstages = p.K / BK;                         // K = KH * Cg, Cg % BK == 0
  const int spr = p.Cg / BK;                            // stages per kernel row
  const uint32_t base0 = (uint32_t)((img * p.H + row0 * p.sh) * RowE);  // first element of the patch in x
  const int A_OFF = 0, S_OFF = p.pt_astage, W_OFF = p.pt_astage + p.st_sbytes;

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.wt_bytes, 0x00020000);

  // ---- weight loader role (as in the LDS-DMA kernel): stage st = 4 rows of mu (+ 4 of delta) of 1 KiB, k0 = st*BK
  const bool w_mu = (NW == 4) || (wave < 4);
  const bool w_dl = (KIND == 1) && ((NW == 4) || (wave >= 4));
  const int w_nops = (w_mu ? 1 : 0) + (w_dl ? 1 : 0);
  const uint32_t w_base = (uint32_t)ntile * (uint32_t)(p.K / G) * 1024u + (uint32_t)lane * 16u +
                          (uint32_t)(wave & 3) * 1024u;
  const int w_lds = W_OFF + (wave & 3) * 1024;
  auto issue_w = [&](int st) __attribute__((always_inline)) {
    const uint32_t go = w_base + (uint32_t)st * (uint32_t)(BK / G) * 1024u;
    unsigned char* ld = smem + w_lds + (st & (WD - 1)) * DW_STAGE;
    if (w_mu) dma16(wt_rsrc, go, ld);
    if (w_dl) dma16(wt_rsrc, go + p.wt_delta_off, ld + 4096);
  };
  for (int s = 0; s < WD - 1 && s < nstages; ++s) issue_w(s);

  // ---- the patch: bytes [base0*esz, +pt_PP) of x, 1 KiB per DMA instruction, straight into LDS
  {
    const int npieces = p.pt_astage >> 10;
    for (int i = wave; i < npieces; i += NW) {
      const uint32_t o = (uint32_t)i * 1024u + (uint32_t)lane * 16u;
      dma16(x_rsrc, o < (uint32_t)p.pt_PP ? base0 * (uint32_t)esz + o : DMA_OOB, smem + A_OFF + i * 1024);
    }
  }
  // ---- s_in words of the patch's element range (element index space = the row-fused x, as for every other variant)
  const uint32_t word0 = base0 >> 5;
  if constexpr (KIND == 1) {
    const int nwords = p.st_sbytes >> 2;
    for (int w = tid; w < nwords; w += NT)
      *(uint32_t*)(smem + S_OFF + w * 4) = btx_sign_word(word0 + (uint32_t)w, rl.kin_a, rl.kin_b);
  }
  // ---- MFMA role: wave owns output pixels [64*wave, +64) of the tile, flattened (row, col)
  const int nrow = min(p.pt_R, p.Ho - row0);
  const int nvalid = nrow * p.Wo;
  uint32_t a_off[2], e_off[2];  // byte offset of the pixel's window in the patch; absolute element offset (signs)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int pl = wave * 64 + mi * 32 + l31;
    uint32_t ur, uc;
    fdivmod((uint32_t)(pl < nvalid ? pl : 0), p.fd_Wo, (uint32_t)p.Wo, ur, uc);
    const uint32_t eo = ur * (uint32_t)(p.sh * RowE) + uc * (uint32_t)(p.sw * p.C);
    a_off[mi] = eo * (uint32_t)esz;
    e_off[mi] = base0 + eo;
  }

  f32x16 accm[2][2], accd[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accd[a][b][r] = 0.f; }

  // fragments of stage (kh, j): element offset kh*RowE + j*BK inside the lane's window
  auto load_frag = [&](StageFrag& f, int st_e, int wslot) __attribute__((always_inline)) {
    const unsigned char* as = smem + A_OFF + st_e * esz;
    const unsigned char* ws = smem + W_OFF + wslot * DW_STAGE;
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      const int row = 2 * kk + h;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) f.a[kk][mi] = *(const u32x4*)(as + a_off[mi] + row * 16);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) f.wm[kk][ni] = *(const u32x4*)(ws + (row * BN + ni * 32 + l31) * 16);
    }
    if constexpr (KIND == 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        // the stage's 32 signs start at element e0: compose them from the two hashed words they straddle.  Word layout:
        // element pair e>>1 sits at bit 15-(e>>1) (even e) / 31-(e>>1) (odd e), so the run starting at e0 is the word
        // pair shifted left by (e0 & 31) >> 1 within each 16-bit half.
        const uint32_t e0 = e_off[mi] + (uint32_t)st_e;
        const uint32_t wi = (e0 >> 5) - word0;
        const uint32_t w = *(const uint32_t*)(smem + S_OFF + wi * 4);
        const uint32_t w1 = *(const uint32_t*)(smem + S_OFF + wi * 4 + 4);
        const uint32_t k = (e0 & 31u) >> 1;
        const uint32_t lo = ((w & 0xffffu) << 16) | (w1 & 0xffffu);
        const uint32_t hi = (w & 0xffff0000u) | (w1 >> 16);
        f.sw[mi] = ((lo << k) >> 16) | ((hi << k) & 0xffff0000u);
      }
    }
  };

  // =================== main loop: the activations are resident; only the weight ring moves ==================
  if (nstages > 0) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int nissued = 0, m1 = 0;
    int l_j = 0, l_rowE = 0, l_e = 0;  // stage being loaded: (kernel row offset, stage within the row) -> element offset
    auto advance_load = [&]() __attribute__((always_inline)) {
      l_e += BK;
      if (++l_j == spr) { l_j = 0; l_rowE += RowE; l_e = l_rowE; }
    };
    // split-bf16 (PREC == 2): a stage reads its own fragments — with the next stage's prefetched beside the hi / lo halves of
    // this one's the Flipout kernel spilt 5 VGPRs inside the loop, and a scratch reload's vmcnt(0) drains the weight ring
    constexpr bool PF = (PREC != 2);
    StageFrag fa, fb;
    if constexpr (PF) {
      load_frag(fa, 0, 0);
      advance_load();
    }
    auto iter = [&](int s, StageFrag& cur, StageFrag& nxt) __attribute__((always_inline)) {
      int m2 = nissued;
      if (s + WD - 1 < nstages) { issue_w(s + WD - 1); nissued += w_nops; m2 = nissued; }
      DeltaFrag dfrag;  // this stage's delta weights first, then the prefetch of the next stage's fragments
      load_delta<KIND>(dfrag, smem + W_OFF + (s & (WD - 1)) * DW_STAGE, l31, h);
      if constexpr (PF) {
        if (s + 1 < nstages) { load_frag(nxt, l_e, (s + 1) & (WD - 1)); advance_load(); }
      } else {
        load_frag(cur, l_e, s & (WD - 1));
        advance_load();
      }
      stage_mma<PREC, KIND>(cur, dfrag, accm, accd, l31, h);
      wait_vmcnt(nissued - m1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      m1 = m2;
    };
    int s = 0;
    for (; s + 1 < nstages; s += 2) {
      iter(s, fa, fb);
      iter(s + 1, fb, fa);
    }
    if (s < nstages) iter(s, fa, fb);
  }

  // =================== epilogue (btx_epilogue.h) ============================================================
  {
    BTX_SECTION_PARAMS(pe, logical2);
    const uint32_t m0 = (uint32_t)(img * pe.Ho + row0) * (uint32_t)pe.Wo;
    staged_epilogue<KIND, NW>(pe, rl, accm, accd, smem, tid, wave, lane, ntile, 0, 0, m0, nvalid);
  }
}

template <int PREC>
static int launch_contract_stem_impl(int kind, const ContractParams& p, int nwg, hipStream_t st) {
#define BTX_LAUNCH_ST(KIND, NW)                                                                                   \
  do {                                                                                                            \
    auto kfn = contract_stem_kernel<PREC, KIND, NW>;                                                              \
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
  if (p.pt_nw == 4) { if (kind == 0) BTX_LAUNCH_ST(0, 4); else BTX_LAUNCH_ST(1, 4); }
  else { if (kind == 0) BTX_LAUNCH_ST(0, 8); else BTX_LAUNCH_ST(1, 8); }
#undef BTX_LAUNCH_ST
  return (int)hipGetLastError();
}

}  // namespace btx
