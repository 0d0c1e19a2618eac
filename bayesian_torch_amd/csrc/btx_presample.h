// btx_presample.h — sampling pre-pass of the fused contraction kernels (gfx950).
#pragma once
#include "btx_contract.h"
#include "btx_rng.h"

namespace btx {

// ---- sampling pre-pass --------------------------------------------------------------------------------------
// Every workgroup of a convolution needs the same sampled weight tile: with 392 pixel tiles (ResNet18 layer1) the
// in-kernel sampler of the other variants repeats each Philox/Box-Muller/softplus 392 times, and that — not the
// memory pipeline — is what bounds them (A/B in DESIGN.md §5).  Here the weights are sampled ONCE per launch into
// MFMA-ready tiles in the workspace:   wt[tile = group*ntiles + ntile][kg = k/G][ch 0..63][G elements]  (16-byte
// granules; `mu` array, then the `delta` = sigma*eps array for Flipout; Reparameterization stores mu + sigma*eps in the
// first array).  One K-stage of a workgroup is then 4 (+4) contiguous 1-KiB rows: one LDS-DMA instruction each.
// Same element indices, same _hw sampling functions and the same rounding as the in-kernel sampler: the values are
// bit-identical to what the other variants compute.
// one quad (4 consecutive k of one output channel) of the tile image; `t` enumerates (padded channel, quad)
// MC sample lanes (lanes > 1): the quad's mu / sigma are fetched ONCE and every lane's noise is drawn on top — lane l
// samples for index sample + l (sample_ptr[l]) into its own tiles `lane_stride` bytes behind the previous lane's (Flipout:
// the delta tiles, the mean tiles exist once; Reparameterization: the W tiles).  Values per lane are those of a one-lane
// call with that index.
// split-bf16 tiles (PREC == 2, btx_mma.h): the quad's granule holds [4 bf16 hi | 4 bf16 lo], hi = rn(w), lo = rn(w - hi)
__device__ __forceinline__ u32x4 pack_quad_split(const float* w) {
  // no fused multiply-add across the split: `w` is the ROUNDED f32 product sigma*eps in every kernel that samples (hipcc
  // contracts `sigma * eps - hi` into one fma in some call sites and not in others: tiles would differ by an ulp of lo)
#pragma clang fp contract(off)
  const f32x4 v = {w[0], w[1], w[2], w[3]};
  const u32x2 hb = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
  const f32x4 r = {w[0] - u2f(hb[0] << 16), w[1] - u2f(hb[0] & 0xffff0000u), w[2] - u2f(hb[1] << 16),
                   w[3] - u2f(hb[1] & 0xffff0000u)};
  const u32x2 lb = __builtin_bit_cast(u32x2, __builtin_convertvector(r, bf16x4));
  return (u32x4){hb[0], hb[1], lb[0], lb[1]};
}

template <int PREC>
__device__ __forceinline__ void presample_quad(int kind, const float* __restrict__ mu, const float* __restrict__ rho,
                                               unsigned char* __restrict__ wt, uint32_t delta_off, int Ng, int K,
                                               int ntiles, uint32_t t, uint32_t seed_lo, uint32_t seed_hi,
                                               uint32_t sample, uint32_t layer, int Cp = 0, int KWp = 0, int src_KW = 0,
                                               int src_C = 0, const float* __restrict__ eps_w = nullptr,
                                               bool write_mu = true, float* __restrict__ sig = nullptr, int lanes = 1,
                                               uint32_t lane_stride = 0, const uint32_t* __restrict__ sample_ptr = nullptr) {
  // sig (btx_sample_weights_lanes): sigma = softplus(rho) of every weight in tile order, f32.  write_mu: this call
  // computes sigma from rho and stores it beside the mean tiles; !write_mu (BTX_SAMPLE_SKIP_MU): mu is not needed and
  // sigma is READ from there — per MC sample the pre-pass is Philox + Box-Muller, one multiply, one store.
  constexpr int G = (PREC == 1) ? 8 : 4;
  // t enumerates the OUTPUT image linearly — (tile, k-granule, channel, quad of the granule), quad fastest — so a wave
  // writes 512 (bf16) / 1024 (f32) contiguous bytes; its reads are 32-byte (bf16) / 16-byte runs, one per channel
  constexpr uint32_t QPG = G / 4;  // quads per granule
  const uint32_t qh = t % QPG, ch = (t / QPG) & 63u, rest = t / (QPG * 64u);
  const uint32_t KG = (uint32_t)K / G;
  const uint32_t tile = rest / KG, kg0 = rest - tile * KG;
  const uint32_t quad = kg0 * QPG + qh;
  const int group = (int)tile / ntiles, ntile = (int)tile - group * ntiles;
  const int col = ntile * BN + (int)ch;
  const uint32_t kg_ = (4u * quad) / G;
  const uint32_t o = (((uint32_t)tile * ((uint32_t)K / G) + kg_) * 64u + (uint32_t)ch) * 16u;
  const uint32_t so_ = (PREC == 1) ? (o + (quad & 1u) * 8u) * 2u : o;  // byte offset of this quad's f32 sigmas
  const uint32_t e0 = (uint32_t)(group * Ng + col) * (uint32_t)K + 4u * quad;
  const bool live = col < Ng;  // (channels that pad the last n-tile hold zeros)
  const bool cached = (sig != nullptr) && !write_mu && (kind == 1);
  float mu4[4] = {0.f, 0.f, 0.f, 0.f}, sg4[4] = {0.f, 0.f, 0.f, 0.f};
  if (live && cached) {
    const f32x4 s4 = *(const f32x4*)((const unsigned char*)sig + so_);
    sg4[0] = s4[0]; sg4[1] = s4[1]; sg4[2] = s4[2]; sg4[3] = s4[3];
  } else if (live) {
    f32x4 m4, rho4;
    if (Cp == 0) {
      m4 = *(const f32x4*)(mu + e0);
      rho4 = *(const f32x4*)(rho + e0);
    } else {
      // padded layout [n][rows][KWp][Cp] sampled straight from the caller's unpadded [n][rows][src_KW][src_C] weights:
      // padded taps / channels get mu = 0 and sigma = 0 (rho = -1e30), i.e. they contribute exactly nothing
      const uint32_t k = 4u * quad, tap_p = k / (uint32_t)Cp, c0 = k - tap_p * (uint32_t)Cp;
      const uint32_t rowtap = tap_p / (uint32_t)KWp, kw = tap_p - rowtap * (uint32_t)KWp;
      const uint32_t rows = (uint32_t)K / (uint32_t)(Cp * KWp);
      const uint32_t sbase = (((uint32_t)(group * Ng + col) * rows + rowtap) * (uint32_t)src_KW + kw) * (uint32_t)src_C;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = kw < (uint32_t)src_KW && c0 + e < (uint32_t)src_C;
        m4[e] = ok ? mu[sbase + c0 + e] : 0.f;
        rho4[e] = ok ? rho[sbase + c0 + e] : -1e30f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { mu4[e] = m4[e]; sg4[e] = btx_softplus_hw(rho4[e]); }
  }
  if (sig != nullptr && write_mu && kind == 1) *(f32x4*)((unsigned char*)sig + so_) = (f32x4){sg4[0], sg4[1], sg4[2], sg4[3]};
  const uint32_t oo = (PREC == 1) ? o + (quad & 1u) * 8u : o;
  if (kind == 1 && write_mu) {
    if constexpr (PREC == 1) *(u32x2*)(wt + oo) = pack_quad_bf16(mu4);
    else if constexpr (PREC == 2) *(u32x4*)(wt + oo) = pack_quad_split(mu4);
    else *(u32x4*)(wt + oo) = (u32x4){f2u(mu4[0]), f2u(mu4[1]), f2u(mu4[2]), f2u(mu4[3])};
  }
  for (int l = 0; l < lanes; ++l) {
    float eps[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      if (eps_w) {  // BtxNoise.eps_w (parity mode, one lane): same [N][K] layout as mu
        const f32x4 e4 = *(const f32x4*)(eps_w + e0);
        eps[0] = e4[0]; eps[1] = e4[1]; eps[2] = e4[2]; eps[3] = e4[3];
      } else {
        const uint32_t smp = sample_ptr ? __builtin_amdgcn_readfirstlane(sample_ptr[l]) : sample + (uint32_t)l;
        btx_normal4_hw(e0 >> 2, smp, layer, 0u, seed_lo, seed_hi, eps);
      }
    }
    float w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (kind == 0) ? __builtin_fmaf(sg4[e], eps[e], mu4[e]) : sg4[e] * eps[e];
    unsigned char* dst = wt + (kind == 1 ? delta_off : 0u) + (uint32_t)l * lane_stride + oo;
    if constexpr (PREC == 1) *(u32x2*)dst = pack_quad_bf16(w);
    else if constexpr (PREC == 2) *(u32x4*)dst = pack_quad_split(w);
    else *(u32x4*)dst = (u32x4){f2u(w[0]), f2u(w[1]), f2u(w[2]), f2u(w[3])};
  }
}

template <int PREC, int KIND>
__global__ __launch_bounds__(256) void presample_kernel(const float* __restrict__ mu, const float* __restrict__ rho,
                                                        unsigned char* __restrict__ wt, uint32_t delta_off, int Ng,
                                                        int K, int ntiles, uint32_t nquads_total, uint32_t seed_lo,
                                                        uint32_t seed_hi, uint32_t sample, uint32_t layer,
                                                        const uint32_t* __restrict__ sample_ptr,
                                                        const float* __restrict__ eps_w, int write_mu) {
  if (sample_ptr) sample = __builtin_amdgcn_readfirstlane(*sample_ptr);
  for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < nquads_total; t += gridDim.x * 256u)
    presample_quad<PREC>(KIND, mu, rho, wt, delta_off, Ng, K, ntiles, t, seed_lo, seed_hi, sample, layer, 0, 0, 0, 0,
                         eps_w, write_mu != 0);
}

// Batched form (btx_sample_weights): the weights of up to PRESAMPLE_MAX_ITEMS layers in ONE launch — a model's
// 21 per-layer pre-passes cost ~6.5 us each, mostly launch latency.  The item table travels in the kernel arguments.
constexpr int PRESAMPLE_MAX_ITEMS = 32;
struct PresampleItem {
  const float* mu;
  const float* rho;
  unsigned char* wt;
  uint32_t delta_off, nquads, first_block, layer;
  int Ng, K, ntiles, kind;
  int Cp, KWp, src_KW, src_C;  // Cp != 0: mu/rho are the unpadded weights of a channel-/tap-padded layout
};
struct PresampleBatch {
  PresampleItem it[PRESAMPLE_MAX_ITEMS];
  int n;
  uint32_t seed_lo, seed_hi, sample, total_blocks;
  const uint32_t* sample_ptr;
  // MC sample lanes: every thread draws the noise of all lanes for its quad (mu / sigma fetched once); lane l samples for
  // index sample + l (sample_ptr[l]) into its own tiles — Flipout: delta tiles at delta_off * (1 + l), the mean tiles
  // exist once (not rewritten when skip_mu: they are current); Reparameterization: W tiles at delta_off * l
  int lanes, skip_mu;
};
template <int PREC>
__global__ __launch_bounds__(256) void presample_batch_kernel(const PresampleBatch b) {
  const uint32_t blk = blockIdx.x;
  int i = 0;
  for (int j = 1; j < b.n; ++j)
    if (blk >= b.it[j].first_block) i = j;
  const PresampleItem& it = b.it[i];
  const uint32_t nblk = (i + 1 < b.n ? b.it[i + 1].first_block : b.total_blocks) - it.first_block;
  const int lanes = b.lanes > 1 ? b.lanes : 1;
  // Flipout: [mu tiles | delta tiles of lane 0 | lane 1 | ... | sigma cache (f32, tile order)]; the cache is written
  // together with the mean tiles and read instead of (mu, rho) when the caller vouches the parameters are unchanged
  float* sig = (it.kind == 1) ? (float*)(it.wt + (size_t)it.delta_off * (size_t)(1 + lanes)) : nullptr;
  const bool write_mu = !b.skip_mu;
  for (uint32_t t = (blk - it.first_block) * 256u + threadIdx.x; t < it.nquads; t += nblk * 256u)
    presample_quad<PREC>(it.kind, it.mu, it.rho, it.wt, it.delta_off, it.Ng, it.K, it.ntiles, t, b.seed_lo, b.seed_hi,
                         b.sample, it.layer, it.Cp, it.KWp, it.src_KW, it.src_C, nullptr, write_mu, sig, lanes,
                         it.delta_off, b.sample_ptr);
}

template <int PREC>
static int launch_presample_impl(int kind, const ContractParams& p, hipStream_t st) {
  if (p.wt_ready) return 0;  // the caller sampled the weights already (btx_sample_weights)
  const uint32_t nq = (uint32_t)(p.groups * p.ntiles * 64) * ((uint32_t)p.K >> 2);
  uint32_t blocks = (nq + 255u) / 256u;
  if (blocks > 4096u) blocks = 4096u;
  const int lanes = p.lanes > 1 ? p.lanes : 1;
  for (int l = 0; l < lanes; ++l) {  // one pre-pass per MC sample lane (callers that care sample ahead: btx_sample_weights)
    const uint32_t* sp = p.sample_ptr ? p.sample_ptr + l : nullptr;
    if (kind == 0)
      hipLaunchKernelGGL((presample_kernel<PREC, 0>), dim3(blocks), dim3(256), 0, st, p.mu, p.rho,
                         (unsigned char*)p.wt + (size_t)l * (size_t)p.lane_wt, p.wt_delta_off, p.Ng, p.K, p.ntiles, nq, p.seed_lo,
                         p.seed_hi, p.sample + (uint32_t)l, p.layer, sp, p.eps_w, 1);
    else
      hipLaunchKernelGGL((presample_kernel<PREC, 1>), dim3(blocks), dim3(256), 0, st, p.mu, p.rho, (unsigned char*)p.wt,
                         p.wt_delta_off + (uint32_t)((size_t)l * (size_t)p.lane_wt), p.Ng, p.K, p.ntiles, nq, p.seed_lo,
                         p.seed_hi, p.sample + (uint32_t)l, p.layer, sp, p.eps_w, l == 0 ? 1 : 0);
  }
  return (int)hipGetLastError();
}

}  // namespace btx
