// btx_api.hip — the C-ABI of libbtx.so (include/btx.h) + the small HBM-bound kernels
// (KL reduce, noise materialisation, MC predictive accumulation).  gfx950 only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include "../../include/btx.h"
#include <stdlib.h>
#include "btx_contract.h"
#include "btx_rng.h"
#include "btx_presample.h"
namespace btx { constexpr int DBM = 512; }  // pixels per tile of the LDS-DMA variant (btx_contract_dma.h)

// A/B knobs of the measurement builds (tools/build_variants.sh ... "-DBTX_TUNING", tools/kbench.py): the shipped library
// reads no environment variable — every dispatch decision is a function of the call's arguments.
static inline const char* tune_env(const char* name) {
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

using namespace btx;

// ========================================================================================================
// K1: KL(q||p) mean.  Reference: layers/base_variational_layer.py:65-68 (kl_div), sigma = log1p(exp(rho))
// from e.g. layers/flipout_layers/conv_flipout.py:362-368.  Each term is evaluated in f32 exactly as the
// reference spells it; the sum is carried in f64 and reduced in a fixed order (deterministic, no atomics).
// HBM-bound: 8 B/element read once.
// ========================================================================================================
constexpr int KL_BLOCK = 256;
constexpr int KL_MAX_BLOCKS = 1024;

__device__ __forceinline__ float kl_term(float mu, float rho, float pmu, float psig) {
  const float sig = log1pf(expf(rho));
  const float dm = mu - pmu;
  return logf(psig) - logf(sig) + (sig * sig + dm * dm) / (2.0f * (psig * psig)) - 0.5f;
}

__global__ __launch_bounds__(KL_BLOCK) void kl_partial_kernel(const float* __restrict__ mu, const float* __restrict__ rho,
                                                              size_t n, const float* __restrict__ pmu_t,
                                                              const float* __restrict__ psig_t, float pmu, float psig,
                                                              double* __restrict__ partials) {
  double acc = 0.0;
  const size_t nthreads = (size_t)gridDim.x * KL_BLOCK;
  const size_t t = (size_t)blockIdx.x * KL_BLOCK + threadIdx.x;
  const size_t n4 = n >> 2;
  const bool vec_ok = ((((uintptr_t)mu | (uintptr_t)rho) & 15) == 0) && !pmu_t && !psig_t;
  if (vec_ok) {
    for (size_t i = t; i < n4; i += nthreads) {
      const f32x4 m = ((const f32x4*)mu)[i];
      const f32x4 r = ((const f32x4*)rho)[i];
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) s += kl_term(m[e], r[e], pmu, psig);
      acc += (double)s;
    }
    for (size_t i = (n4 << 2) + t; i < n; i += nthreads) acc += (double)kl_term(mu[i], rho[i], pmu, psig);
  } else {
    for (size_t i = t; i < n; i += nthreads)
      acc += (double)kl_term(mu[i], rho[i], pmu_t ? pmu_t[i] : pmu, psig_t ? psig_t[i] : psig);
  }
  // wave64 shuffle reduce -> LDS -> one value per block
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double wsum[KL_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < KL_BLOCK / 64; ++w) s += wsum[w];
    partials[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(64) void kl_final_kernel(const double* __restrict__ partials, int nblocks, double inv_n,
                                                      float* __restrict__ out, int accumulate) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) acc += partials[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) {
    const float kl = (float)(acc * inv_n);
    out[0] = accumulate ? out[0] + kl : kl;
  }
}

// Batched form: every parameter tensor of a model in ONE launch (+ one final reduce).  get_kl_loss() of ResNet18 is 22
// tensors; launched one by one (2 launches each) the reduction ran at ~3 % of the HBM roofline, launch-bound.
constexpr int KL_MAX_ITEMS = 48;
struct KlItemDev {
  const float* mu; const float* rho; const float* pmu_t; const float* psig_t;
  float* dmu; float* drho;  // backward only
  float pmu, psig;
  uint32_t n, first_block;
};
struct KlBatchDev {
  KlItemDev it[KL_MAX_ITEMS];
  int n;
  uint32_t total_blocks;
};
__global__ __launch_bounds__(KL_BLOCK) void kl_model_partial_kernel(const KlBatchDev b, double* __restrict__ partials) {
  int i = 0;
  for (int j = 1; j < b.n; ++j)
    if (blockIdx.x >= b.it[j].first_block) i = j;
  const KlItemDev& it = b.it[i];
  const uint32_t nblk = (i + 1 < b.n ? b.it[i + 1].first_block : b.total_blocks) - it.first_block;
  const size_t nthreads = (size_t)nblk * KL_BLOCK;
  const size_t t = (size_t)(blockIdx.x - it.first_block) * KL_BLOCK + threadIdx.x;
  double acc = 0.0;
  const bool vec_ok = ((((uintptr_t)it.mu | (uintptr_t)it.rho) & 15) == 0) && !it.pmu_t && !it.psig_t;
  const size_t n = it.n, n4 = n >> 2;
  if (vec_ok) {
    for (size_t k = t; k < n4; k += nthreads) {
      const f32x4 m = ((const f32x4*)it.mu)[k];
      const f32x4 r = ((const f32x4*)it.rho)[k];
      float s_ = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) s_ += kl_term(m[e], r[e], it.pmu, it.psig);
      acc += (double)s_;
    }
    for (size_t k = (n4 << 2) + t; k < n; k += nthreads) acc += (double)kl_term(it.mu[k], it.rho[k], it.pmu, it.psig);
  } else {
    for (size_t k = t; k < n; k += nthreads)
      acc += (double)kl_term(it.mu[k], it.rho[k], it.pmu_t ? it.pmu_t[k] : it.pmu, it.psig_t ? it.psig_t[k] : it.psig);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double wsum[KL_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s_ = 0.0;
#pragma unroll
    for (int w = 0; w < KL_BLOCK / 64; ++w) s_ += wsum[w];
    // the reference takes the MEAN of each tensor, rounds it to f32 and sums the means: keep the mean scaling per tensor
    partials[blockIdx.x] = s_ / (double)n;
  }
}
// d(mean KL)/d(mu, rho) of every tensor, scaled by the upstream gradient (a device scalar: no host sync)
__global__ __launch_bounds__(KL_BLOCK) void kl_model_bwd_kernel(const KlBatchDev b, const float* __restrict__ gout) {
  int i = 0;
  for (int j = 1; j < b.n; ++j)
    if (blockIdx.x >= b.it[j].first_block) i = j;
  const KlItemDev& it = b.it[i];
  const uint32_t nblk = (i + 1 < b.n ? b.it[i + 1].first_block : b.total_blocks) - it.first_block;
  const float g = gout[0] / (float)it.n;
  for (size_t k = (size_t)(blockIdx.x - it.first_block) * KL_BLOCK + threadIdx.x; k < it.n; k += (size_t)nblk * KL_BLOCK) {
    const float mu = it.mu[k], rho = it.rho[k];
    const float pm = it.pmu_t ? it.pmu_t[k] : it.pmu, ps = it.psig_t ? it.psig_t[k] : it.psig;
    const float sig = log1pf(expf(rho));
    const float dsig = 1.0f / (1.0f + expf(-rho));  // d softplus / d rho
    const float ips2 = 1.0f / (ps * ps);
    it.dmu[k] = g * (mu - pm) * ips2;
    it.drho[k] = g * (sig * ips2 - 1.0f / sig) * dsig;
  }
}

// ========================================================================================================
// noise materialisation (BTX-RNG v1)
// ========================================================================================================
__global__ __launch_bounds__(256) void fill_eps_kernel(float* __restrict__ out, size_t n, uint32_t k0, uint32_t k1,
                                                       uint32_t sample, uint32_t layer, uint32_t stream,
                                                       const uint32_t* __restrict__ sample_ptr) {
  if (sample_ptr) sample = __builtin_amdgcn_readfirstlane(*sample_ptr);  // BtxRng.sample_idx_dev (captured steps)
  const size_t nblk = (n + 3) >> 2;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < nblk; b += (size_t)gridDim.x * 256) {
    float z[4];
    btx_normal4((uint32_t)b, sample, layer, stream, k0, k1, z);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((b << 2) + e < n) out[(b << 2) + e] = z[e];
  }
}

// drho = dw * eps * sigmoid(rho): the elementwise follow-up of the weight gradient, eps regenerated (never materialised)
__global__ __launch_bounds__(256) void rho_grad_kernel(const float* __restrict__ dw, const float* __restrict__ rho,
                                                       float* __restrict__ drho, size_t n, uint32_t k0, uint32_t k1,
                                                       uint32_t sample, uint32_t layer, uint32_t stream,
                                                       const uint32_t* __restrict__ sample_ptr) {
  if (sample_ptr) sample = __builtin_amdgcn_readfirstlane(*sample_ptr);
  const size_t nblk = (n + 3) >> 2;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < nblk; b += (size_t)gridDim.x * 256) {
    float z[4];
    btx_normal4((uint32_t)b, sample, layer, stream, k0, k1, z);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t i = (b << 2) + e;
      if (i < n) drho[i] = dw[i] * z[e] * (1.0f / (1.0f + expf(-rho[i])));
    }
  }
}

__global__ __launch_bounds__(256) void fill_sign_kernel(int8_t* __restrict__ out, size_t n, uint32_t ka, uint32_t kb) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint32_t w = btx_sign_word((uint32_t)(i >> 5), ka, kb);
    out[i] = ((w >> btx_sign_bitpos((uint32_t)i & 31u)) & 1u) ? (int8_t)-1 : (int8_t)1;
  }
}

// ========================================================================================================
// K6: MC predictive accumulation.  Reference (host side, numpy/torch): torch.stack(output_mc) -> softmax(dim=2)
// -> mean(dim=0)  examples/main_bayesian_imagenet_dnn2bnn.py:483-499 ; predictive_entropy / mutual_information
// utils/util.py:41-60.  One workgroup per batch row; the row is owned by that workgroup so no atomics.
// ========================================================================================================
// lanes > 1 (btx_mc_accumulate_lanes): the logits of `lanes` MC samples back to back ([lanes][bs][C]).  A workgroup owns a
// batch row; its sixteen waves take the lanes round-robin — one wave computes one lane's softmax row (probabilities into LDS, the
// lane's entropy beside them) with no workgroup barrier — then every thread adds its columns' probabilities lane by lane IN
// ORDER: the same additions, in the same order, as `lanes` single-sample launches (which run this very code with one lane),
// at a sixteenth of the serial depth (20 lanes: 78 -> ~20 us per replay of the bench).  The per-row reductions keep a fixed
// shape — 256 "virtual threads" (4 per thread: column v + 256 k), a shuffle tree per virtual wave, ((r0 + r1) + r2) + r3 — so a
// row's figures do not depend on which wave computed it.  LDS: min(lanes, LC) x C floats; more lanes run in chunks of LC.
template <typename ACT>
__global__ __launch_bounds__(1024) void mc_accumulate_kernel(const ACT* __restrict__ logits, int bs, int C, float kl,
                                                            float* __restrict__ packed, int lanes, int LC) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) unsigned char mc_smem[];
  float* const pr_lds = (float*)mc_smem;            // [LC][C]
  float* const ent_lds = pr_lds + (size_t)LC * C;   // [LC]
  const int row = blockIdx.x;
  const int wave = threadIdx.x >> 6, li = threadIdx.x & 63;
  float* const sp = packed + (size_t)row * C;
  float* const sp2 = packed + (size_t)bs * C + (size_t)row * C;
  for (int l0 = 0; l0 < lanes; l0 += LC) {
    const int nl = min(LC, lanes - l0);
    for (int k = wave; k < nl; k += 16) {  // wave-uniform
      const ACT* lr = logits + ((size_t)(l0 + k) * bs + row) * C;
      float* const pk = pr_lds + (size_t)k * C;
      float mx = -INFINITY;
      for (int c = li; c < C; c += 64) mx = fmaxf(mx, (float)lr[c]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      float se[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        for (int c = j * 64 + li; c < C; c += 256) se[j] += expf((float)lr[c] - mx);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) se[j] += __shfl_down(se[j], off, 64);
      const float tot = __shfl(((se[0] + se[1]) + se[2]) + se[3], 0, 64);
      const float inv = 1.0f / tot;
      float en[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        for (int c = j * 64 + li; c < C; c += 256) {
          const float pv = expf((float)lr[c] - mx) * inv;
          pk[c] = pv;
          const float t = pv * logf(pv + 1e-15f);  // utils/util.py:44 epsilon
          en[j] -= t;
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) en[j] += __shfl_down(en[j], off, 64);
      if (li == 0) ent_lds[k] = ((en[0] + en[1]) + en[2]) + en[3];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 1024) {
      float a = sp[c], a2 = sp2[c];
      for (int k = 0; k < nl; ++k) {
        const float pv = pr_lds[(size_t)k * C + c];
        a += pv;
        const float q = pv * pv;
        a2 += q;
      }
      sp[c] = a;
      sp2[c] = a2;
    }
    if (threadIdx.x == 0) {
      float e = packed[(size_t)2 * bs * C + row];
      for (int k = 0; k < nl; ++k) e += ent_lds[k];
      packed[(size_t)2 * bs * C + row] = e;
      if (row == 0) {
        float a = packed[(size_t)2 * bs * C + bs], n = packed[(size_t)2 * bs * C + bs + 1];
        for (int k = 0; k < nl; ++k) { a += kl; n += 1.0f; }
        packed[(size_t)2 * bs * C + bs] = a;
        packed[(size_t)2 * bs * C + bs + 1] = n;
      }
    }
    __syncthreads();  // the next chunk overwrites the LDS rows
  }
}

// ========================================================================================================
// host side
// ========================================================================================================
static void sign_keys(const BtxRng* rng, uint32_t stream, uint32_t* ka, uint32_t* kb) {
  const BtxPhilox4 k = btx_philox4x32_10(0u, rng->sample_idx, rng->layer_id, stream, (uint32_t)rng->seed,
                                         (uint32_t)(rng->seed >> 32));
  *ka = k.x[0];
  *kb = k.x[1];
}


// ========================================================================================================
// btx_rowfuse_pack: the data-format step in front of the small-C stem path (BTX_FLAG_ROWFUSE) — logical [N,C,H,W]
// activations in any strides/dtype -> zero-padded channels-last [N][Hp][Wp][cp] in the MFMA dtype, one pass
// (replaces a fill + a strided copy + a cast).  One thread per output pixel.
// ========================================================================================================
template <typename IN, typename OUT, int CP>
__global__ __launch_bounds__(256) void rowfuse_pack_kernel(const IN* __restrict__ x, OUT* __restrict__ out, int NB, int C,
                                                           int H, int W, int Hp, int Wp, int ph, int pw, long long sn,
                                                           long long sc, long long sh, long long sw, long long total) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int wp = (int)(t % Wp);
    const long long r = t / Wp;
    const int hp = (int)(r % Hp);
    const int n = (int)(r / Hp);
    const int h = hp - ph, w = wp - pw;
    struct alignas(sizeof(OUT) * CP) Px { OUT v[CP]; };
    Px px;
    OUT* v = px.v;
#pragma unroll
    for (int c = 0; c < CP; ++c) v[c] = (OUT)0.f;
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
      const IN* src = x + n * sn + h * sh + w * sw;
#pragma unroll
      for (int c = 0; c < CP; ++c)
        if (c < C) v[c] = (OUT)(float)src[c * sc];
    }
    *(Px*)(out + t * CP) = px;  // one 8/16/32-byte store per pixel
  }
}
template <typename IN, typename OUT>
static int launch_rowfuse_pack(const void* x, void* out, int NB, int C, int H, int W, int Hp, int Wp, int cp, int ph,
                               int pw, const int64_t* st, hipStream_t stream) {
  const long long total = (long long)NB * Hp * Wp;
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (cp == 4)
    hipLaunchKernelGGL((rowfuse_pack_kernel<IN, OUT, 4>), dim3((int)blocks), dim3(256), 0, stream, (const IN*)x, (OUT*)out,
                       NB, C, H, W, Hp, Wp, ph, pw, (long long)st[0], (long long)st[1], (long long)st[2], (long long)st[3], total);
  else
    hipLaunchKernelGGL((rowfuse_pack_kernel<IN, OUT, 8>), dim3((int)blocks), dim3(256), 0, stream, (const IN*)x, (OUT*)out,
                       NB, C, H, W, Hp, Wp, ph, pw, (long long)st[0], (long long)st[1], (long long)st[2], (long long)st[3], total);
  return (int)hipGetLastError();
}


// ========================================================================================================
// btx_maxpool2d_cl: channels-last max pooling, the op between the stem and layer1 of a ResNet (reference
// models/deterministic/resnet_large.py: self.maxpool).  HBM-bound: every thread owns 8 channels (16 B bf16 / 32 B
// f32) of one output pixel, reads its window with 16-byte loads, writes once.
// ========================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void maxpool2d_cl_kernel(const T* __restrict__ x, T* __restrict__ out, int NB, int H,
                                                           int W, int C, int Ho, int Wo, int k, int s, int pad,
                                                           long long total) {
  const int cgs = C >> 3;  // groups of 8 channels
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int cg = (int)(t % cgs);
    long long r = t / cgs;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int kh = 0; kh < k; ++kh) {
      const int h = ho * s - pad + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int w = wo * s - pad + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        const T* src = x + (((long long)n * H + h) * W + w) * C + cg * 8;
        if constexpr (sizeof(T) == 2) {
          const u32x4 v = *(const u32x4*)src;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            m[2 * j] = fmaxf(m[2 * j], u2f(v[j] << 16));
            m[2 * j + 1] = fmaxf(m[2 * j + 1], u2f(v[j] & 0xffff0000u));
          }
        } else {
          const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { m[j] = fmaxf(m[j], a[j]); m[4 + j] = fmaxf(m[4 + j], b[j]); }
        }
      }
    }
    T* dst = out + (((long long)n * Ho + ho) * Wo + wo) * C + cg * 8;
    if constexpr (sizeof(T) == 2) {
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (f2u(m[2 * j]) >> 16) | (f2u(m[2 * j + 1]) & 0xffff0000u);  // exact: inputs are bf16
      *(u32x4*)dst = o;
    } else {
      *(f32x4*)dst = (f32x4){m[0], m[1], m[2], m[3]};
      *(f32x4*)(dst + 4) = (f32x4){m[4], m[5], m[6], m[7]};
    }
  }
}

// Training form (the reference's training loop runs nn.MaxPool2d under autograd: resnet_large.py self.maxpool): the same pass also
// writes, per output element, the position kh * k + kw of its maximum inside the window — the FIRST maximum in scan order, as
// torch's max_pool2d_with_indices picks it (`val > max || isnan(val)`: post-ReLU maps are full of ties at 0) — one byte instead of
// ATen's int64 index; the backward routes dy with it.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2d_cl_idx_kernel(const T* __restrict__ x, T* __restrict__ out, uint8_t* __restrict__ idx,
                                                               int NB, int H, int W, int C, int Ho, int Wo, int k, int s, int pad,
                                                               long long total) {
  const int cgs = C >> 3;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int cg = (int)(t % cgs);
    long long r = t / cgs;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float m[8];
    uint32_t id[8];
    bool first = true;
    for (int kh = 0; kh < k; ++kh) {
      const int h = ho * s - pad + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int w = wo * s - pad + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        const T* src = x + (((long long)n * H + h) * W + w) * C + cg * 8;
        float v[8];
        if constexpr (sizeof(T) == 2) {
          const u32x4 q = *(const u32x4*)src;
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[2 * j] = u2f(q[j] << 16); v[2 * j + 1] = u2f(q[j] & 0xffff0000u); }
        } else {
          const f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = c[j]; }
        }
        const uint32_t pos = (uint32_t)(kh * k + kw);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool take = first || v[j] > m[j] || v[j] != v[j];
          m[j] = take ? v[j] : m[j];
          id[j] = take ? pos : id[j];
        }
        first = false;
      }
    }
    if (first) {  // a window entirely in the padding (2 * pad <= k rules it out; kept total)
#pragma unroll
      for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; id[j] = 0u; }
    }
    T* dst = out + (((long long)n * Ho + ho) * Wo + wo) * C + cg * 8;
    if constexpr (sizeof(T) == 2) {
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (f2u(m[2 * j]) >> 16) | (f2u(m[2 * j + 1]) & 0xffff0000u);  // exact: inputs are bf16
      *(u32x4*)dst = o;
    } else {
      *(f32x4*)dst = (f32x4){m[0], m[1], m[2], m[3]};
      *(f32x4*)(dst + 4) = (f32x4){m[4], m[5], m[6], m[7]};
    }
    u32x2 ib;
    ib[0] = id[0] | (id[1] << 8) | (id[2] << 16) | (id[3] << 24);
    ib[1] = id[4] | (id[5] << 8) | (id[6] << 16) | (id[7] << 24);
    *(u32x2*)(idx + t * 8) = ib;
  }
}

// dx[n][h][w][c] = sum of dy over the (at most ceil(k/s)^2) windows that cover (h, w) and whose recorded maximum sits there; f32
// accumulation, one rounding (as ATen's max_pool_backward_nhwc).  A thread owns 8 channels of one INPUT pixel: no atomics.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2d_cl_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                               T* __restrict__ dx, int NB, int H, int W, int C, int Ho, int Wo, int k,
                                                               int s, int pad, long long total) {
  const int cgs = C >> 3;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int cg = (int)(t % cgs);
    long long r = t / cgs;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int th = h + pad - k + 1, tw = w + pad - k + 1;
    const int ho_lo = th <= 0 ? 0 : (th + s - 1) / s, ho_hi = min(Ho - 1, (h + pad) / s);
    const int wo_lo = tw <= 0 ? 0 : (tw + s - 1) / s, wo_hi = min(Wo - 1, (w + pad) / s);
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      const int kh = h + pad - ho * s;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const uint32_t pos = (uint32_t)(kh * k + (w + pad - wo * s));
        const long long o = (((long long)n * Ho + ho) * Wo + wo) * cgs + cg;
        const u32x2 ib = *(const u32x2*)(idx + o * 8);
        float g[8];
        if constexpr (sizeof(T) == 2) {
          const u32x4 q = *(const u32x4*)(dy + o * 8);
#pragma unroll
          for (int j = 0; j < 4; ++j) { g[2 * j] = u2f(q[j] << 16); g[2 * j + 1] = u2f(q[j] & 0xffff0000u); }
        } else {
          const f32x4 a = *(const f32x4*)(dy + o * 8), c = *(const f32x4*)(dy + o * 8 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { g[j] = a[j]; g[4 + j] = c[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (((ib[j >> 2] >> (8 * (j & 3))) & 0xffu) == pos) ? g[j] : 0.f;
      }
    }
    T* dst = dx + t * 8;
    if constexpr (sizeof(T) == 2) {
      *(u32x4*)dst = pack_granule<1>(acc);  // round to nearest even, as torch's float -> bfloat16
    } else {
      *(f32x4*)dst = (f32x4){acc[0], acc[1], acc[2], acc[3]};
      *(f32x4*)(dst + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
    }
  }
}


// btx_avgpool_global_cl: global average pooling of channels-last activations ([NB][HW][C] -> [NB][C], f32 accumulate),
// the op in front of the classifier of the reference's ResNets (resnet_large.py: avgpool).  One workgroup per image and
// 64-channel slab: 8 lanes cover the slab with 16-byte loads, 32 pixel groups run in parallel, LDS tree at the end.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_global_cl_kernel(const T* __restrict__ x, T* __restrict__ out, int HW, int C,
                                                                float inv) {
  const int n = blockIdx.y, slab = blockIdx.x;
  const int cg = threadIdx.x & 7, pg = threadIdx.x >> 3;  // 8 channel groups x 32 pixel groups
  const int c0 = slab * 64 + cg * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < C) {
    for (int pix = pg; pix < HW; pix += 32) {
      const T* src = x + ((long long)n * HW + pix) * C + c0;
      if constexpr (sizeof(T) == 2) {
        const u32x4 v = *(const u32x4*)src;
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[2 * j] += u2f(v[j] << 16); acc[2 * j + 1] += u2f(v[j] & 0xffff0000u); }
      } else {
        const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] += a[j]; acc[4 + j] += b[j]; }
      }
    }
  }
  __shared__ float red[32][64 + 1];
#pragma unroll
  for (int j = 0; j < 8; ++j) red[pg][cg * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int g = 0; g < 32; ++g) s += red[g][threadIdx.x];  // fixed order: deterministic
    const int c = slab * 64 + threadIdx.x;
    if (c < C) out[(long long)n * C + c] = (T)(s * inv);
  }
}

extern "C" {

int btx_abi_version(void) { return BTX_ABI_VERSION; }

const char* btx_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case BTX_E_NULL: return "btx: required pointer is NULL";
    case BTX_E_SHAPE: return "btx: inconsistent or non-positive shape";
    case BTX_E_UNSUPPORTED: return "btx: unsupported configuration";
    case BTX_E_WORKSPACE: return "btx: workspace too small";
    case BTX_E_DTYPE: return "btx: unknown dtype / precision code";
    case BTX_E_ALIGN: return "btx: pointer not 16-byte aligned";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "btx: unknown error";
  }
}

size_t btx_kl_workspace_bytes(size_t n) {
  (void)n;
  return (size_t)KL_MAX_BLOCKS * sizeof(double);
}

int btx_kl_gauss(const float* mu, const float* rho, size_t n, const float* prior_mu_t, const float* prior_sigma_t,
                 float prior_mu, float prior_sigma, float* kl_out, uint32_t flags, void* ws, size_t ws_bytes,
                 void* stream) {
  if (!mu || !rho || !kl_out || !ws) return BTX_E_NULL;
  if (n == 0) return BTX_E_SHAPE;
  if (ws_bytes < btx_kl_workspace_bytes(n)) return BTX_E_WORKSPACE;
  if (((uintptr_t)ws & 7) != 0) return BTX_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  size_t want = (n + (size_t)KL_BLOCK * 8 - 1) / ((size_t)KL_BLOCK * 8);
  int nblocks = (int)(want < 1 ? 1 : (want > KL_MAX_BLOCKS ? KL_MAX_BLOCKS : want));
  hipLaunchKernelGGL(kl_partial_kernel, dim3(nblocks), dim3(KL_BLOCK), 0, st, mu, rho, n, prior_mu_t, prior_sigma_t,
                     prior_mu, prior_sigma, (double*)ws);
  hipLaunchKernelGGL(kl_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, nblocks, 1.0 / (double)n, kl_out,
                     (flags & BTX_FLAG_KL_ACCUM) ? 1 : 0);
  return (int)hipGetLastError();
}

static int kl_fill_batch(const BtxKlItem* items, int base, int n_items, bool bwd, KlBatchDev* b) {
  memset(b, 0, sizeof(*b));
  b->n = n_items - base < KL_MAX_ITEMS ? n_items - base : KL_MAX_ITEMS;
  uint32_t blocks = 0;
  for (int i = 0; i < b->n; ++i) {
    const BtxKlItem& s = items[base + i];
    if (!s.mu || !s.rho || (bwd && (!s.dmu || !s.drho))) return BTX_E_NULL;
    if (s.n == 0 || s.n > 0xffffffffull) return BTX_E_SHAPE;
    KlItemDev& it = b->it[i];
    it.mu = s.mu; it.rho = s.rho; it.pmu_t = s.prior_mu_t; it.psig_t = s.prior_sigma_t; it.dmu = s.dmu; it.drho = s.drho;
    it.pmu = s.prior_mu; it.psig = s.prior_sigma; it.n = (uint32_t)s.n; it.first_block = blocks;
    uint32_t nb = (uint32_t)((s.n + (size_t)KL_BLOCK * 8 - 1) / ((size_t)KL_BLOCK * 8));
    if (nb < 1) nb = 1;
    if (nb > 256u) nb = 256u;
    blocks += nb;
  }
  b->total_blocks = blocks;
  return 0;
}

size_t btx_kl_model_workspace_bytes(int n_items) {
  if (n_items <= 0) return 0;
  return (size_t)((n_items + KL_MAX_ITEMS - 1) / KL_MAX_ITEMS) * KL_MAX_ITEMS * 256 * sizeof(double);
}

int btx_kl_gauss_model(const BtxKlItem* items, int n_items, float* kl_out, void* ws, size_t ws_bytes, void* stream) {
  if (!items || !kl_out || !ws) return BTX_E_NULL;
  if (n_items <= 0) return BTX_E_SHAPE;
  if (ws_bytes < btx_kl_model_workspace_bytes(n_items)) return BTX_E_WORKSPACE;
  if (((uintptr_t)ws & 7) != 0) return BTX_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  int total = 0;
  for (int base = 0; base < n_items; base += KL_MAX_ITEMS) {
    KlBatchDev b;
    int rc = kl_fill_batch(items, base, n_items, false, &b);
    if (rc) return rc;
    hipLaunchKernelGGL(kl_model_partial_kernel, dim3(b.total_blocks), dim3(KL_BLOCK), 0, st, b, part + total);
    total += (int)b.total_blocks;
  }
  hipLaunchKernelGGL(kl_final_kernel, dim3(1), dim3(64), 0, st, (const double*)part, total, 1.0, kl_out, 0);
  return (int)hipGetLastError();
}

int btx_kl_gauss_model_bwd(const BtxKlItem* items, int n_items, const float* grad_out, void* stream) {
  if (!items || !grad_out) return BTX_E_NULL;
  if (n_items <= 0) return BTX_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < n_items; base += KL_MAX_ITEMS) {
    KlBatchDev b;
    int rc = kl_fill_batch(items, base, n_items, true, &b);
    if (rc) return rc;
    hipLaunchKernelGGL(kl_model_bwd_kernel, dim3(b.total_blocks), dim3(KL_BLOCK), 0, st, b, grad_out);
  }
  return (int)hipGetLastError();
}

int btx_out_shape(const BtxGeom* g, uint32_t flags, int32_t* Do, int32_t* Ho, int32_t* Wo) {
  if (!g || !Do || !Ho || !Wo) return BTX_E_NULL;
  if (g->NB <= 0 || g->D <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->N <= 0 || g->KD <= 0 || g->KH <= 0 ||
      g->KW <= 0 || g->sd <= 0 || g->sh <= 0 || g->sw <= 0 || g->dd <= 0 || g->dh <= 0 || g->dw <= 0 ||
      g->pd < 0 || g->ph < 0 || g->pw < 0 || g->groups <= 0)
    return BTX_E_SHAPE;
  if (g->C % g->groups || g->N % g->groups) return BTX_E_SHAPE;
  if (flags & BTX_FLAG_TRANSPOSED) {
    *Do = (g->D - 1) * g->sd - 2 * g->pd + g->dd * (g->KD - 1) + g->od + 1;
    *Ho = (g->H - 1) * g->sh - 2 * g->ph + g->dh * (g->KH - 1) + g->oh + 1;
    *Wo = (g->W - 1) * g->sw - 2 * g->pw + g->dw * (g->KW - 1) + g->ow + 1;
  } else {
    *Do = (g->D + 2 * g->pd - g->dd * (g->KD - 1) - 1) / g->sd + 1;
    *Ho = (g->H + 2 * g->ph - g->dh * (g->KH - 1) - 1) / g->sh + 1;
    *Wo = (g->W + 2 * g->pw - g->dw * (g->KW - 1) - 1) / g->sw + 1;
  }
  if (*Do <= 0 || *Ho <= 0 || *Wo <= 0) return BTX_E_SHAPE;
  return 0;
}

// Workgroup slots the split-K cost model fills with 4-wave blocks.  512 (two per CU) minimises the latency of a single
// launch on an otherwise idle GPU (ResNet18 layer3: 63.5 vs 69.8 us); 256 splits K half as often, which wins as soon as
// several MC samples are in flight (mc.GraphedMC lanes, the bench default: 1.16 -> 1.22 k MC-samples/s) because the
// partial sums cost HBM traffic and a reduce launch while the other samples fill the idle CUs anyway.  BTX_SLOTS4
// overrides.
static long long slots4() {
  static const char* e = tune_env("BTX_SLOTS4");
  static const long long v = e ? atoll(e) : 256;
  return v > 0 ? v : 256;
}

// tiling plan shared by btx_contract_workspace_bytes and btx_contract_fwd
struct Plan {
  int Do, Ho, Wo, Cg, Ng, M, K, mtiles, ntiles, ksplits, kper, nwg;
};

// MC sample lanes of the launch being planned (BTX_FLAG_LANES(n) in the flags): the grid is `lanes` copies of the
// single-sample grid, so that many times more workgroups fill the workgroup slots before K has to be split
static inline long long plan_lanes(uint32_t flags) {
  const long long n = (flags >> BTX_FLAG_LANES_SHIFT) & 0xffu;
  return n > 1 ? n : 1;
}

static inline bool throughput_plan(uint32_t flags) { return (flags & BTX_FLAG_CONCURRENT) || plan_lanes(flags) > 1; }

static int make_plan(const BtxGeom* g, int prec, uint32_t flags, int bm, Plan* pl) {
  int rc = btx_out_shape(g, flags, &pl->Do, &pl->Ho, &pl->Wo);
  if (rc) return rc;
  if (prec != BTX_PREC_F32 && prec != BTX_PREC_BF16 && prec != BTX_PREC_BF16X3) return BTX_E_DTYPE;
  pl->Cg = g->C / g->groups;
  pl->Ng = g->N / g->groups;
  const long long M = (long long)g->NB * pl->Do * pl->Ho * pl->Wo;
  const long long K = (long long)g->KD * g->KH * g->KW * pl->Cg;
  if (M > 0x7fffffffLL || K > 0x7fffffffLL) return BTX_E_UNSUPPORTED;
  pl->M = (int)M;
  pl->K = (int)K;
  const int bk = NG * (prec == BTX_PREC_BF16 ? 8 : 4);
  pl->mtiles = (pl->M + bm - 1) / bm;
  pl->ntiles = (pl->Ng + BN - 1) / BN;
  const long long base1 = (long long)pl->mtiles * pl->ntiles * g->groups;
  const long long base = base1 * plan_lanes(flags);  // workgroups of all lanes
  const int stages = (pl->K + bk - 1) / bk;
  // split-K: pick the split that minimises (grid rounds on 256 CUs) x (stages per block + fixed per-block cost);
  // each split keeps >= 4 stages so the DMA ring fills.
  int ks = 1;
  {
    const long long ncu = (bm == 256) ? slots4() : 256;  // 256-pixel tiles run two blocks per CU
    long long best = -1;
    const int max_ks = stages / 4 > 1 ? (stages / 4 < 32 ? stages / 4 : 32) : 1;
    for (int c = 1; c <= max_ks; ++c) {
      const long long rounds = (base1 * c + ncu - 1) / ncu;  // of ONE lane: the split must not depend on the lane count
      const long long cost = rounds * ((stages + c - 1) / c + 4) + (c > 1 ? 1 : 0);  // +1: the reduce pass
      if (best < 0 || cost < best) { best = cost; ks = c; }
      // BTX_FLAG_CONCURRENT: other launches fill the CUs this one leaves idle, so what counts is its CU-time, and that
      // only grows with the split (fixed per-block cost, partial sums through HBM, the reduce launch): split just far
      // enough that the launch is not a long thin tail of its own stream.  Launches with MC sample lanes take the same
      // plan, decided by the grid of ONE lane: the K split — the f32 summation order — of a sample then does not depend
      // on how many samples share its launch, nor on how the samples were grouped over launches and ranks.
      // 16 workgroups of at most 16 stages are enough as well: ResNet18's fc (16 n-tiles of 16 stages per lane) in ONE piece —
      // 4 splits of 4 stages + the reduce launch measured 77 us per 20 lanes against 39 (profiles/r06_experiments.txt E18)
      long long tp_min = 64, tp_short = 16;
      if (const char* e = tune_env("BTX_TP_MINWG")) { tp_min = atoll(e); tp_short = 1 << 30; }  // A/B (E18)
      if (throughput_plan(flags) && (base1 * c >= tp_min || (base1 * c >= tp_short && (stages + c - 1) / c <= 16))) { ks = c; break; }
    }
  }
  int per_stages = (stages + ks - 1) / ks;
  pl->kper = per_stages * bk;
  pl->ksplits = (pl->K + pl->kper - 1) / pl->kper;
  const long long nwg = base1 * pl->ksplits;  // per lane
  if (nwg > 0x7fffffffLL) return BTX_E_UNSUPPORTED;
  pl->nwg = (int)nwg;
  return 0;
}

// Shape-level eligibility of the LDS-DMA pipeline (btx_contract_dma.h); pointer alignment is checked at launch.
static bool dma_shape_ok(const BtxGeom* g, int act_dtype, int prec, const Plan& pl) {
  if ((prec == BTX_PREC_BF16) != (act_dtype == BTX_ACT_BF16)) return false;  // DMA cannot convert
  const int bk = NG * (prec == BTX_PREC_BF16 ? 8 : 4);
  if (pl.Cg % bk) return false;  // a K-stage must lie inside one filter tap
  const long long in_elems = (long long)g->NB * g->D * g->H * g->W * g->C;
  const long long esz = (act_dtype == BTX_ACT_BF16) ? 2 : 4;
  if (in_elems * esz >= 0xfff00000LL || (long long)pl.M * g->N >= 0x7fffffffLL ||
      (long long)g->N * pl.K * 4 >= 0xfff00000LL)
    return false;  // 32-bit byte offsets inside the buffer descriptors
  return true;
}

// Tile plan of the patch variant (btx_contract_patch.h): stride-1 2-D convolutions with more than one tap whose
// activations already have the contraction dtype.  Returns false when the shape is not eligible.
struct PatchPlan {
  int G, R, Rp, Wp, PP, NI, rtiles, nw, mi, astage, lds;
  int taps, kg, lds_g;  // tap-unrolled kernel (btx_contract_taps.h): 10*KH+KW or 0; K-groups per workgroup; LDS per group
  int wide;             // tap-unrolled kernel, Reparameterization: 64-pixel x 128-channel wave tiles, ntiles / 2 grid n-tiles
  int tall, P, Wt, ncs;  // tall-strip tiles (ContractParams.pt_tall): virtual rows per image, strip width, strips per row tile
};
// Tall-strip tile of the tap-unrolled kernel: the batch as ONE tall image with P = max(H + ph, Ho) virtual rows per image
// (P - H zero rows between consecutive images: bottom padding of one, top padding of the next; output rows >= Ho of a
// period are dummies), cut into tiles of R virtual rows x Wt columns.  R and Wt need not divide Ho / Wo, so the tile
// can fill the 256 pixel slots of a workgroup whatever the map size (7 | 14 | 28 | 56: 252 pixels) where whole-row /
// whole-image tiles leave an eighth to a quarter of the MFMA tiles empty.  Returns the fraction of pixel slots that hold
// real output pixels (0: no tall tile fits).
static double tall_tile(const BtxGeom* g, const Plan& pl, int tp, int ppcap, PatchPlan* pt) {
  const int halo_r = (g->KH - 1) * g->dh, halo_c = (g->KW - 1) * g->dw;
  if (g->ph > halo_r) return 0.0;
  const int P = (g->H + g->ph > pl.Ho) ? g->H + g->ph : pl.Ho;
  const long long rows_total = (long long)(g->NB - 1) * P + pl.Ho;
  double best = 0.0;
  for (int ncs = 1; ncs <= 8; ++ncs) {
    const int Wt = (pl.Wo + ncs - 1) / ncs;
    if (Wt > tp || Wt < 4 || (ncs > 1 && Wt < 8)) continue;  // Wt >= 4: PixTall::Walk steps 8 pixels with two row wraps
    int R = tp / Wt;
    if (R > rows_total) R = (int)rows_total;
    while (R >= 1 && (R + halo_r) * (Wt + halo_c) > ppcap) --R;
    if (R < 1) continue;
    const long long rtiles = (rows_total + R - 1) / R;
    const double eff = (double)pl.M / ((double)rtiles * ncs * tp);
    if (eff > best * 1.01) {
      best = eff;
      pt->tall = 1; pt->P = P; pt->Wt = Wt; pt->ncs = ncs;
      pt->G = 1; pt->R = R; pt->Rp = R + halo_r; pt->Wp = Wt + halo_c; pt->PP = pt->Rp * pt->Wp;
      pt->rtiles = (int)rtiles;
    }
  }
  return best;
}
#ifndef BTX_GEMM8_MINK_DEFAULT
#define BTX_GEMM8_MINK_DEFAULT 128
#endif
#ifndef BTX_WG_MB_DEFAULT
#define BTX_WG_MB_DEFAULT 3.0
#endif
#ifndef BTX_TALL_MIN_DEFAULT
#define BTX_TALL_MIN_DEFAULT 1.15
#endif
// tile of `tp` output pixels whose patch holds at most `ppcap` pixels
static bool patch_tile(const BtxGeom* g, const Plan& pl, int tp, int ppcap, PatchPlan* pt) {
  const int Ho = pl.Ho, Wo = pl.Wo;
  const int Wp = Wo + (g->KW - 1) * g->dw, halo_r = (g->KH - 1) * g->dh;
  if (Wo > tp || Wp * (1 + halo_r) > ppcap) return false;
  int G = 1, R;
  if (Ho * Wo <= tp / 2 || (Ho * Wo <= tp && (Ho + halo_r) * Wp <= ppcap)) {
    R = Ho;
    const int Rp = R + halo_r;
    if (Rp * Wp > ppcap) return false;
    G = tp / (Ho * Wo);
    if (G > ppcap / (Rp * Wp)) G = ppcap / (Rp * Wp);
    if (G > g->NB) G = g->NB;
    if (G < 1) return false;
  } else {
    int rmax = tp / Wo;
    const int rfit = ppcap / Wp - halo_r;
    if (rfit < rmax) rmax = rfit;
    if (rmax > Ho) rmax = Ho;
    if (rmax < 1) return false;
    const int nrt = (Ho + rmax - 1) / rmax;
    R = (Ho + nrt - 1) / nrt;
    // equal row tiles, unless the tallest tile that fits issues fewer 32-pixel MFMA tiles over the image (a tile's tail of < 32
    // pixels still costs a whole MFMA tile per stage): 28 rows of 28 pixels as 7 + 7 + 7 + 7 are 4 x 7 = 28 MFMA tiles, as
    // 8 + 8 + 8 + 4 they are 3 x 7 + 4 = 25 (the stride-2 3x3 layer at 56 -> 28: -10.7 % of its MFMAs)
    auto mfma_tiles = [&](int r) {
      long long n = 0;
      for (int row = 0; row < Ho; row += r) n += ((long long)((Ho - row < r) ? Ho - row : r) * Wo + 31) / 32;
      return n;
    };
    if (rmax > R && (Ho % rmax == 0 || 2 * (Ho % rmax) >= rmax) && mfma_tiles(rmax) < mfma_tiles(R)) R = rmax;  // (no sliver of a last tile)
  }
  pt->G = G; pt->R = R; pt->Rp = R + halo_r; pt->Wp = Wp; pt->PP = G * pt->Rp * Wp;
  pt->rtiles = (Ho + R - 1) / R;
  return pt->PP <= ppcap && G * R * Wo <= tp;
}
// kind: BTX_KIND_* of the launch being planned, or -1 (the plan every kind can take)
static bool make_patch_plan(const BtxGeom* g, int act_dtype, int prec, uint32_t flags, Plan* pl, PatchPlan* pt, int kind = -1) {
  if (flags & (BTX_FLAG_TRANSPOSED | BTX_FLAG_ROWFUSE)) return false;
  if (make_plan(g, prec, flags, DBM, pl)) return false;
  pt->wide = 0;
  if (!dma_shape_ok(g, act_dtype, prec, *pl)) return false;
  if (g->D != 1 || g->KD != 1 || pl->Do != 1 || g->sh != 1 || g->sw != 1) return false;
  const int T = g->KH * g->KW;
  if (T < 2 || T > 64) return false;
  // 4-wave blocks, two per CU: 2 patch slots + 2 sign slots + 4 weight tiles within 80 KiB -> 22 pieces = 352 pixels;
  // 8-wave blocks, one per CU: 60 pieces = 960 pixels.  BTX_PATCH_NW=8 forces the latter (A/B measurements).
  static const char* nw_env = tune_env("BTX_PATCH_NW");
  const bool force8 = nw_env && atoi(nw_env) == 8;
  // BTX_PATCH_MI=4: 4 waves x 128 pixels, one block per CU, 256 accumulators per wave (A/B measurements)
  static const char* mi_env = tune_env("BTX_PATCH_MI");
  const bool want_mi4 = mi_env && atoi(mi_env) == 4;
  pt->mi = 2;
  pt->tall = 0; pt->P = 1; pt->Wt = 1; pt->ncs = 1;
  if (want_mi4 && patch_tile(g, *pl, 512, 960, pt)) { pt->nw = 4; pt->mi = 4; }
  else if (!force8 && patch_tile(g, *pl, 256, 352, pt)) pt->nw = 4;
  else if (patch_tile(g, *pl, 512, 960, pt)) pt->nw = 8;
  else return false;
  // 3x3 on 4-wave blocks (the tap-unrolled kernel): tall-strip tiles when they fill the pixel slots better
  if (pt->nw == 4 && pt->mi == 2 && g->KH == 3 && g->KW == 3 && !tune_env("BTX_NO_TALL") && !tune_env("BTX_NO_TAPS")) {
    const long long mt_old = (long long)((g->NB + pt->G - 1) / pt->G) * pt->rtiles;
    const double eff_old = (double)pl->M / ((double)mt_old * 256.0);
    PatchPlan tp = *pt;
    const double eff_tall = tall_tile(g, *pl, 256, 352, &tp);
    // measured (tools/kbench.py --throughput-plan, batch 256 / 512 = the tiles of 4 / 8 MC sample lanes): the heavier
    // tile (4 full waves, a store side 50 % longer) pays off only where it removes >= ~15 % of the workgroups (28x28:
    // 19 %, +3 / +8 %; 14x14: 16 %, -5 / +2 %); on 56x56 (9 % fewer workgroups) and 7x7 (6 %) it loses 4-7 %
    const char* te = tune_env("BTX_TALL_MIN");  // A/B: the gain in pixel-slot efficiency from which tall strips are taken
    const double tall_min = te ? atof(te) : BTX_TALL_MIN_DEFAULT;
    if (eff_tall > eff_old * tall_min) *pt = tp;
  }
  const int pieces = (pt->PP + 15) / 16;
  pt->NI = (pieces + pt->nw - 1) / pt->nw;
  if (pt->NI > (pt->mi == 4 ? 16 : PT_MAXNI)) return false;
  pt->astage = pieces * 1024;
  int lds = 2 * pt->astage + 2 * (pt->astage / 16) + PT_WD * 8192 + 1024;  // + the scratch piece of btx_contract_taps.h
  const int ep = pt->nw * PT_EP_WAVE + 1024;
  if (lds < ep) lds = ep;
  if (lds > ((pt->nw == 4 && pt->mi == 2) ? 81920 : 163840)) return false;
  pt->lds = lds;
  pt->lds_g = (lds + 15) & ~15;
  // grid: m-tiles are (image group, row tile); split-K over the channel blocks
  const int bk = NG * (prec == BTX_PREC_BF16 ? 8 : 4);
  const int ncb = pl->Cg / bk;
  pl->mtiles = pt->tall ? pt->rtiles * pt->ncs : ((g->NB + pt->G - 1) / pt->G) * pt->rtiles;
  const bool no_taps = tune_env("BTX_NO_TAPS") != nullptr;  // A/B: the run-time-tap patch kernel instead (read per call)
  pt->taps = (!no_taps && pt->nw == 4 && pt->mi == 2 && pt->NI <= 6 && g->KH == 3 && g->KW == 3) ? 33 : 0;
  // Reparameterization on the tap-unrolled kernel: one accumulator set per output, so the wave can hold a 64-pixel x 128-channel
  // tile (contract_taps_kernel<..., WIDE>) — taken when whole pairs of n-tiles exist and the halved grid still fills the
  // workgroup slots (few-tile launches keep the narrow tile and its K-groups).  BTX_NO_WIDE=1 disables (A/B).
  if (kind == BTX_KIND_REPARAM && pt->taps == 33 && prec == BTX_PREC_BF16 && act_dtype == BTX_ACT_BF16 && (pl->Ng % 128) == 0 &&
      (long long)pl->mtiles * (pl->ntiles / 2) * g->groups * plan_lanes(flags) >= slots4() && !tune_env("BTX_NO_WIDE"))
    pt->wide = 1;
  if (pt->wide && pt->lds < pt->nw * PT_EP_WAVE + 2048) {  // its store side keeps the constants of two channel tiles
    pt->lds = pt->nw * PT_EP_WAVE + 2048;
    pt->lds_g = (pt->lds + 15) & ~15;
  }
  const long long base1 = (long long)pl->mtiles * (pt->wide ? pl->ntiles / 2 : pl->ntiles) * g->groups;
  const long long base = base1 * plan_lanes(flags);
  // Few pixel tiles (at most one 4-wave block per CU): 8-wave blocks of two K-groups — split-K inside the workgroup
  // through LDS instead of through HBM, and two waves per SIMD.  BTX_NO_KG=1 disables (A/B).
  // BTX_FLAG_CONCURRENT: plain 4-wave blocks — an 8-wave block takes the whole LDS of its CU, so two such launches of
  // different MC samples cannot share a CU; 4-wave blocks of two launches pair up and free-run against each other
  // (measured, ResNet18 bs 64, 3 / 4 / 6 samples in flight: 1340 / 1369 / 1346 -> 1382 / 1402 / 1378 MC-samples/s).
  const bool no_kg = tune_env("BTX_NO_KG") != nullptr || throughput_plan(flags);
  pt->kg = (pt->taps && !no_kg && base <= 256 && ncb >= 2 && (ncb % 2) == 0 && 2 * pt->lds_g <= 163840) ? 2 : 1;
  const int units = ncb / pt->kg;  // channel blocks per K-group over the whole K
  int ks = 1;
  {
    const long long slots = (pt->kg == 2) ? 256 : ((pt->nw == 4 && pt->mi == 2) ? slots4() : 256);
    long long best = -1;
    for (int c = 1; c <= units && c <= 32; ++c) {
      const int per = (units + c - 1) / c;
      if (c > 1 && per * T < 4) break;
      if (pt->kg == 2 && units % c) continue;  // the 8-wave kernel wants every split full
      const long long rounds = (base1 * c + slots - 1) / slots;
      const long long cost = rounds * (per * T + 4) + (c > 1 ? 1 : 0);
      if (best < 0 || cost < best) { best = cost; ks = c; }
      if (throughput_plan(flags) && base1 * c * pt->kg >= 64) { ks = c; break; }  // see make_plan
    }
  }
  const int per = (units + ks - 1) / ks;
  pl->kper = per * pt->kg * bk;
  pl->ksplits = (units + per - 1) / per;
  const long long nwg = base1 * pl->ksplits;
  if (nwg > 0x7fffffffLL) return false;
  pl->nwg = (int)nwg;
  if (pt->kg == 2) pt->lds = 2 * pt->lds_g;
  return true;
}

// Tile plan of the stride-2 form of the tap-unrolled kernel (btx_contract_taps2.h): 3x3 / stride 2 / pad 1, one phase
// plane of (R+1) x (Wo+1) pixels per image of the tile in each of three LDS slots.
static bool make_patch2_plan(const BtxGeom* g, int act_dtype, int prec, uint32_t flags, Plan* pl, PatchPlan* pt, int kind = -1) {
  if (flags & (BTX_FLAG_TRANSPOSED | BTX_FLAG_ROWFUSE)) return false;
  pt->wide = 0;
  if (tune_env("BTX_NO_TAPS2")) return false;  // A/B: the per-tap LDS-DMA kernel instead
  if (make_plan(g, prec, flags, DBM, pl)) return false;
  if (!dma_shape_ok(g, act_dtype, prec, *pl)) return false;
  if (g->D != 1 || g->KD != 1 || pl->Do != 1) return false;
  if (g->KH != 3 || g->KW != 3 || g->sh != 2 || g->sw != 2 || g->ph != 1 || g->pw != 1 || g->dh != 1 || g->dw != 1) return false;
  BtxGeom gp = *g;  // the plane of a tile is the halo'd patch of a 2x2 stride-1 window: R+1 rows, Wo+1 columns
  gp.KH = 2; gp.KW = 2;
  if (!patch_tile(&gp, *pl, 256, 272, pt)) return false;
  pt->nw = 4; pt->mi = 2;
  pt->tall = 0; pt->P = 1; pt->Wt = 1; pt->ncs = 1;
  const int pieces = (pt->PP + 15) / 16;
  pt->NI = (pieces + 3) / 4;
  if (pt->NI > 5) return false;
  pt->astage = pieces * 1024;
  int lds = 3 * pt->astage + 3 * (pt->astage / 16) + 3 * 8192 + 1024;
  const int ep = 4 * PT_EP_WAVE + 1024;
  if (lds < ep) lds = ep;
  if (lds > 81920) return false;
  pt->lds = lds;
  pt->lds_g = (lds + 15) & ~15;
  const int bk = NG * (prec == BTX_PREC_BF16 ? 8 : 4);
  const int ncb = pl->Cg / bk;
  pl->mtiles = ((g->NB + pt->G - 1) / pt->G) * pt->rtiles;
  // Reparameterization: 64-pixel x 128-channel wave tiles (contract_taps2_kernel<..., WIDE>) under the conditions of the stride-1 plan
  if (kind == BTX_KIND_REPARAM && prec == BTX_PREC_BF16 && act_dtype == BTX_ACT_BF16 && (pl->Ng % 128) == 0 &&
      (long long)pl->mtiles * (pl->ntiles / 2) * g->groups * plan_lanes(flags) >= slots4() && !tune_env("BTX_NO_WIDE")) {
    pt->wide = 1;
    if (pt->lds < 4 * PT_EP_WAVE + 2048) { pt->lds = 4 * PT_EP_WAVE + 2048; pt->lds_g = (pt->lds + 15) & ~15; }
  }
  const long long base1 = (long long)pl->mtiles * (pt->wide ? pl->ntiles / 2 : pl->ntiles) * g->groups;
  const long long base = base1 * plan_lanes(flags);
  pt->taps = 332;
  pt->kg = 1;
  int ks = 1;
  {
    const long long slots = slots4();
    long long best = -1;
    for (int c = 1; c <= ncb && c <= 32; ++c) {
      const int per = (ncb + c - 1) / c;
      const long long rounds = (base1 * c + slots - 1) / slots;
      const long long cost = rounds * (per * 9 + 4) + (c > 1 ? 1 : 0);
      if (best < 0 || cost < best) { best = cost; ks = c; }
      if (throughput_plan(flags) && base1 * c >= 64) { ks = c; break; }  // see make_plan
    }
  }
  const int per = (ncb + ks - 1) / ks;
  pl->kper = per * bk;
  pl->ksplits = (ncb + per - 1) / per;
  const long long nwg = base1 * pl->ksplits;
  if (nwg > 0x7fffffffLL) return false;
  pl->nwg = (int)nwg;
  return true;
}

// Tile plan of the stem variant (btx_contract_stem.h): row-fused small-C 2-D convolutions; R output rows x full width
// per workgroup, the input rows they need resident in LDS.
struct StemPlan {
  int R, Rp, rtiles, nw, astage, sbytes, lds, patch_bytes, nwg;
};
static bool make_stem_plan(const BtxGeom* g, int act_dtype, int prec, const Plan& pl, StemPlan* st) {
  if (g->D != 1 || g->KD != 1 || pl.Do != 1 || g->groups != 1) return false;
  const int esz = (act_dtype == BTX_ACT_BF16) ? 2 : 4;
  const int bk = NG * (prec == BTX_PREC_BF16 ? 8 : 4);
  if ((g->KW * g->C) % bk || pl.K % bk) return false;
  const long long rowB = (long long)g->W * g->C * esz;
  static const char* snw_env = tune_env("BTX_STEM_NW");  // 8: skip the 4-wave plan (A/B measurements)
  for (int nw = (snw_env && atoi(snw_env) == 8) ? 8 : 4; nw <= 8; nw += 4) {
    const int tp = 64 * nw;
    if (pl.Wo > tp) continue;
    int R = tp / pl.Wo;
    if (R > pl.Ho) R = pl.Ho;
    const long long cap = (nw == 4 ? 81920 : 163840) - PT_WD * 8192;
    for (; R >= 1; --R) {
      const long long Rp = (long long)(R - 1) * g->sh + g->KH;
      const long long pb = Rp * rowB;
      const long long astage = (pb + 1023) / 1024 * 1024;
      const long long sbytes = ((pb / esz + 31) / 32 + 3) * 4;
      const long long sb16 = (sbytes + 15) / 16 * 16;
      if (astage + sb16 > cap) continue;
      long long lds = astage + sb16 + PT_WD * 8192;
      const long long ep = (long long)nw * PT_EP_WAVE + 1024;
      if (lds < ep) lds = ep;
      st->R = R; st->Rp = (int)Rp; st->rtiles = (pl.Ho + R - 1) / R; st->nw = nw; st->astage = (int)astage;
      st->sbytes = (int)sb16; st->lds = (int)lds; st->patch_bytes = (int)pb;
      const long long nwg = (long long)g->NB * st->rtiles * pl.ntiles;
      if (nwg > 0x7fffffffLL) return false;
      st->nwg = (int)nwg;
      return true;
    }
  }
  return false;
}

// Plan of the stem + max-pool variant (btx_contract_stempool.h): 8-wave workgroups, one per CU, each walking a band of
// `PB` pooled rows of one image with all weight tiles resident in LDS.  bf16 only; the pool is 3x3 / stride 2 / pad 1.
struct StemPoolPlan {
  int PB, bands, Rp, astage, sbytes, lds, patch_bytes, nwg, Hq, Wq;
};
static bool make_stem_pool_plan(const BtxGeom* g, int act_dtype, int prec, const Plan& pl, StemPoolPlan* sp, int lanes = 1) {
  if (prec != BTX_PREC_BF16 || act_dtype != BTX_ACT_BF16) return false;
  if (g->D != 1 || g->KD != 1 || pl.Do != 1 || g->groups != 1) return false;
  const int bk = NG * 8;
  if ((g->KW * g->C) % bk || pl.K % bk || (g->N % 64)) return false;
  const int nstages = pl.K / bk;
  if (nstages < 1 || nstages > 7) return false;
  if (2 * pl.Wo > 256 || pl.Ho < 1) return false;
  const int Hq = (pl.Ho - 1) / 2 + 1, Wq = (pl.Wo - 1) / 2 + 1;
  const long long rowB = (long long)g->W * g->C * 2;
  const long long Rp = (long long)g->sh + g->KH;  // input rows of a half tile (two conv rows)
  const long long pb = Rp * rowB;
  const long long astage = (pb + 1023) / 1024 * 1024;
  const long long sbytes = ((pb / 2 + 31) / 32 + 3) * 4;
  const long long sb16 = (sbytes + 127) / 128 * 128;  // keeps the store-side rows 128-byte aligned (chunk swizzle in address bits)
  // weights | raw patch x2 | signed patch copy | sign words x2 | store-side rows r0, r1, carry (128 B per pixel) | constants
  const long long lds = (long long)nstages * 8192 + 3 * astage + 2 * sb16 + 3LL * pl.Wo * 128 + 1024 + 64;  // (+ the pool's two `ninf` chunks)
  if (lds > 163840) return false;
  // bands: about one workgroup per CU (every band pays two phases of fill / drain, a closing one-row half tile and the
  // fetch of the layer's weight tiles).  With MC sample lanes the launch has `lanes` times the (image, n-tile) units, so the
  // bands get longer — 20 lanes of a ResNet stem at batch 64: one band per image instead of four.  Which workgroup computes a
  // row does not change how it is computed: results are bit-identical whatever the band length.
  const long long units = (long long)g->NB * pl.ntiles;
  long long PB = ((long long)Hq * units * (lanes > 1 ? lanes : 1)) / 256;
  if (PB < 4) PB = 4;
  if (PB > Hq) PB = Hq;
  const int bands = (Hq + (int)PB - 1) / (int)PB;
  const long long nwg = units * bands;
  if (nwg > 0x7fffffffLL) return false;
  sp->PB = (int)PB; sp->bands = bands; sp->Rp = (int)Rp; sp->astage = (int)astage; sp->sbytes = (int)sb16;
  sp->lds = (int)lds; sp->patch_bytes = (int)pb; sp->nwg = (int)nwg; sp->Hq = Hq; sp->Wq = Wq;
  return true;
}

// workspace of the patch variant: split-K partials (256-byte padded), then the pre-sampled weight tiles
// Flipout: [mu tiles | delta tiles of lane 0 | lane 1 | ...] — the mu tiles do not depend on the MC sample, one set serves
// every lane; Reparameterization: [W tiles of lane 0 | lane 1 | ...]
static size_t patch_wt_bytes(const Plan& pl, const BtxGeom* g, int kind, int prec, size_t* one, int lanes = 1) {
  const size_t arr = (size_t)g->groups * pl.ntiles * 64 * (size_t)pl.K * (prec == BTX_PREC_BF16 ? 2 : 4);
  if (one) *one = arr;
  return arr * (size_t)(kind == BTX_KIND_FLIPOUT ? 1 + lanes : lanes);
}
static size_t pad256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr size_t BTX_QUEUE_BYTES = 4096;  // image-group queues of the persistent kernel: one counter per tile position (<= 1024)

static size_t plan_ws(const Plan& pl, const BtxGeom* g, int lanes = 1) {  // split-K partials [lane][split][M][N]
  return pl.ksplits > 1 ? (size_t)lanes * (size_t)pl.ksplits * (size_t)pl.M * (size_t)g->N * sizeof(float) : 0;
}

int btx_contract_pool_shape(const BtxGeom* g, int act_dtype, int prec, uint32_t flags, int32_t* Hq, int32_t* Wq) {
  if (!g || !(flags & BTX_FLAG_ROWFUSE) || (flags & (BTX_FLAG_TRANSPOSED | BTX_FLAG_OUT_F32 | BTX_FLAG_SWAP_SIGNS | BTX_FLAG_GATHER)))
    return 0;
  Plan sp;
  StemPlan stp;
  StemPoolPlan spp;
  if (make_plan(g, prec, flags, DBM, &sp) || !make_stem_plan(g, act_dtype, prec, sp, &stp) ||
      !make_stem_pool_plan(g, act_dtype, prec, sp, &spp))
    return 0;
  if (Hq) *Hq = spp.Hq;
  if (Wq) *Wq = spp.Wq;
  return 1;
}

size_t btx_contract_workspace_bytes(const BtxGeom* g, int kind, int act_dtype, int prec, uint32_t flags) {
  Plan a, b;
  if (!g || make_plan(g, prec, flags, BM, &a) || make_plan(g, prec, flags, DBM, &b)) return 0;
  (void)kind;
  const int lanes = (int)plan_lanes(flags);
  size_t wa = plan_ws(a, g, lanes);  // which kernel runs also depends on pointer alignment
  const size_t wb = pad256(plan_ws(b, g, lanes)) + patch_wt_bytes(b, g, BTX_KIND_FLIPOUT, prec, nullptr, lanes);
  if (wb > wa) wa = wb;
  Plan b4;
  if (!make_plan(g, prec, flags, 256, &b4)) {
    const size_t w4 = pad256(plan_ws(b4, g, lanes)) + patch_wt_bytes(b4, g, BTX_KIND_FLIPOUT, prec, nullptr, lanes);
    if (w4 > wa) wa = w4;
  }
  Plan c;
  PatchPlan pt;
  if (make_patch_plan(g, act_dtype, prec, flags, &c, &pt) || make_patch2_plan(g, act_dtype, prec, flags, &c, &pt)) {
    size_t wc = pad256(plan_ws(c, g, lanes)) + patch_wt_bytes(c, g, BTX_KIND_FLIPOUT, prec, nullptr, lanes);
    Plan cw;
    PatchPlan ptw;  // the wide Reparameterization tile halves the grid and may split K differently
    if ((make_patch_plan(g, act_dtype, prec, flags, &cw, &ptw, BTX_KIND_REPARAM) ||
         make_patch2_plan(g, act_dtype, prec, flags, &cw, &ptw, BTX_KIND_REPARAM)) && ptw.wide) {
      const size_t ww = pad256(plan_ws(cw, g, lanes)) + patch_wt_bytes(cw, g, BTX_KIND_FLIPOUT, prec, nullptr, lanes);
      if (ww > wc) wc = ww;
    }
    if (wc > wa) wa = wc;
    if (pt.taps == 33) wa = pad256(wa) + BTX_QUEUE_BYTES;
  }
  return wa;
}

int btx_contract_fwd(int kind, const BtxGeom* g, const void* x, const float* mu_w, const float* rho_w,
                     const float* mu_b, const float* rho_b, void* out, const BtxRng* rng, const BtxNoise* noise,
                     int act_dtype, int prec, uint32_t flags, void* ws, size_t ws_bytes, void* stream) {
  return btx_contract_fwd_ex(kind, g, x, mu_w, rho_w, mu_b, rho_b, out, rng, noise, act_dtype, prec, flags, ws,
                             ws_bytes, stream, nullptr);
}

static int contract_fwd_impl(int kind, const BtxGeom* g, const void* x, const float* mu_w, const float* rho_w,
                             const float* mu_b, const float* rho_b, void* out, const BtxRng* rng, const BtxNoise* noise,
                             int act_dtype, int prec, uint32_t flags, void* ws, size_t ws_bytes, void* stream,
                             const BtxEpilogue* ep, const BtxLanes* ln);

int btx_contract_fwd_ex(int kind, const BtxGeom* g, const void* x, const float* mu_w, const float* rho_w,
                        const float* mu_b, const float* rho_b, void* out, const BtxRng* rng, const BtxNoise* noise,
                        int act_dtype, int prec, uint32_t flags, void* ws, size_t ws_bytes, void* stream,
                        const BtxEpilogue* ep) {
  return contract_fwd_impl(kind, g, x, mu_w, rho_w, mu_b, rho_b, out, rng, noise, act_dtype, prec,
                           flags & ~BTX_FLAG_LANES_MASK, ws, ws_bytes, stream, ep, nullptr);
}

int btx_contract_fwd_lanes(int kind, const BtxGeom* g, const void* x, const float* mu_w, const float* rho_w,
                           const float* mu_b, const float* rho_b, void* out, const BtxRng* rng, const BtxNoise* noise,
                           int act_dtype, int prec, uint32_t flags, void* ws, size_t ws_bytes, void* stream,
                           const BtxEpilogue* ep, const BtxLanes* lanes) {
  if (!lanes) return BTX_E_NULL;
  if (lanes->n < 1 || lanes->n > 255) return BTX_E_SHAPE;
  if ((lanes->x_stride | lanes->out_stride | lanes->res_stride) & 15) return BTX_E_ALIGN;
  if (noise && lanes->n > 1 && (noise->eps_w || noise->eps_b || noise->sign_in || noise->sign_out)) return BTX_E_UNSUPPORTED;
  return contract_fwd_impl(kind, g, x, mu_w, rho_w, mu_b, rho_b, out, rng, noise, act_dtype, prec,
                           (flags & ~BTX_FLAG_LANES_MASK) | BTX_FLAG_LANES(lanes->n), ws, ws_bytes, stream, ep, lanes);
}

static int contract_fwd_impl(int kind, const BtxGeom* g, const void* x, const float* mu_w, const float* rho_w,
                             const float* mu_b, const float* rho_b, void* out, const BtxRng* rng, const BtxNoise* noise,
                             int act_dtype, int prec, uint32_t flags, void* ws, size_t ws_bytes, void* stream,
                             const BtxEpilogue* ep, const BtxLanes* ln) {
  if (!g || !x || !mu_w || !rho_w || !out || !rng) return BTX_E_NULL;
  const int lanes = ln ? ln->n : 1;
  if ((mu_b == nullptr) != (rho_b == nullptr)) return BTX_E_NULL;
  if (kind != BTX_KIND_REPARAM && kind != BTX_KIND_FLIPOUT) return BTX_E_UNSUPPORTED;
  if (act_dtype != BTX_ACT_F32 && act_dtype != BTX_ACT_BF16) return BTX_E_DTYPE;
  Plan pl;
  int rc = make_plan(g, prec, flags, BM, &pl);
  if (rc) return rc;

  // fast (granule) paths need whole 16-byte granules everywhere; otherwise the element-wise gather path
  const int G = (prec == BTX_PREC_BF16) ? 8 : 4;
  const uintptr_t al = (uintptr_t)x | (uintptr_t)mu_w | (uintptr_t)rho_w | (uintptr_t)out |
                       (uintptr_t)(noise && noise->eps_w ? noise->eps_w : nullptr);
  // Explicit noise (parity mode) runs on the same kernels as generated noise: eps_w enters the sampling pre-pass, the
  // sign words are packed from sign_in / sign_out.  BTX_FLAG_GATHER forces the element-wise gather kernel (tests).
  const bool explicit_kloop = (flags & BTX_FLAG_GATHER) != 0;
  const bool gen = (pl.Cg % G != 0) || (al & 15) || explicit_kloop;
  // LDS-DMA pipeline when the activations already have the contraction dtype (no conversion on the way to LDS);
  // BTX_NO_DMA=1 forces the register-staged kernel (A/B measurements).
  static const bool no_dma = tune_env("BTX_NO_DMA") != nullptr;
  const bool rowfuse = (flags & BTX_FLAG_ROWFUSE) != 0;
  bool dma = !gen && !no_dma && dma_shape_ok(g, act_dtype, prec, pl);
  // Sample where the weights are used when nothing shares the sampled tile.  A pointwise layer (Linear, 1x1x1 at stride 1) with
  // at most 256 rows per MC sample reads every weight once per sample: the register-staged kernel — (mu, rho) straight into the
  // wave's registers, softplus + Philox + Box-Muller there, the sampled tile never exists in HBM (north_star's kernel design) —
  // does strictly less memory work than a sampling pre-pass plus a tile DMA (measured equal or faster: BASELINE cfg2 10 718 vs
  // 10 592 MC-samples/s, profiles/r05_experiments.txt E6).  Layers whose tiles are shared by many pixel tiles — every convolution
  // of a ResNet — keep pre-sampled tiles: there an in-kernel sampler repeats each draw once per pixel tile.
  // A caller that hands over pre-sampled tiles (BtxNoise.sampled_w) or explicit noise keeps the LDS-DMA family.
  {
    const bool pointwise_geom = !(flags & BTX_FLAG_TRANSPOSED) && g->KD == 1 && g->KH == 1 && g->KW == 1 && g->sd == 1 && g->sh == 1 &&
                                g->sw == 1 && g->pd == 0 && g->ph == 0 && g->pw == 0;
    // (single-sample launches only: with MC sample lanes the pre-sampled form of the ResNet18 classifier — 20 lanes x 64 rows — runs
    // in 72 us against 109 us, the tiles of all lanes coming from the one sampling launch of the replay)
    // BTX_FLAG_CONCURRENT single-sample launches are planned like lanes (a lane is bit-identical to them): same kernel as the lanes.
    if (dma && !rowfuse && pointwise_geom && lanes == 1 && !(flags & BTX_FLAG_CONCURRENT) && pl.M <= 256 && prec != BTX_PREC_BF16X3 &&
        !(noise && (noise->sampled_w || noise->eps_w || noise->sign_in || noise->sign_out)) &&
        !(flags & (BTX_FLAG_OUT_F32 | BTX_FLAG_OUT_BF16)) && !tune_env("BTX_NO_FUSED_LINEAR"))
      dma = false;
  }
  if (rowfuse) {
    // one K-stage = one kernel row: the K walk sees KW*C "channels" per tap and a single tap per row
    const int esz = (act_dtype == BTX_ACT_BF16) ? 2 : 4;
    const int bk = NG * G;
    const bool ok = !(al & 15) && !explicit_kloop && !(noise && noise->sign_in) && !no_dma && (prec == BTX_PREC_BF16) == (act_dtype == BTX_ACT_BF16) &&
                    g->groups == 1 && g->dw == 1 && g->pw == 0 && !(flags & BTX_FLAG_TRANSPOSED) &&
                    ((g->KW * g->C) % bk == 0) && ((g->sw * g->C * esz) % 16 == 0) && ((g->W * g->C * esz) % 16 == 0) &&
                    (g->C % G == 0 || G % g->C == 0);
    if (!ok) return BTX_E_UNSUPPORTED;
    dma = true;
  }
  // LDS-DMA variant: 4-wave blocks on 256-pixel tiles (two per CU) unless BTX_DMA_NW=8 (A/B measurements)
  static const char* dnw_env = tune_env("BTX_DMA_NW");
  const int dma_nw = (dnw_env && atoi(dnw_env) == 8) ? 8 : 4;
  if (dma) {
    rc = make_plan(g, prec, flags, 64 * dma_nw, &pl);
    if (rc) return rc;
  }
  // Parity-major pixel order (ContractParams.par_major) for the data gradient of a stride-2 2-D convolution — a transposed launch
  // whose gather rule leaves 1, 2, 2 or 4 of a 3x3 filter's 9 taps per output-pixel parity class: with the pixels enumerated class by
  // class every 256-pixel tile walks only its class's taps (2.25 of 9 on average) and needs no K split.  Single-sample launches of
  // the generic LDS-DMA kernel with at least 8 pixel tiles and more than one tap.  BTX_NO_PAR_MAJOR=1 (tuning builds): raster order.
  bool par_major = false;
  int par_mqp = 0;
  if (dma && !rowfuse && (flags & BTX_FLAG_TRANSPOSED) && lanes == 1 && g->groups == 1 && g->D == 1 && g->KD == 1 && g->sd == 1 &&
      g->sh == 2 && g->sw == 2 && g->KH * g->KW <= 31 && g->KH * g->KW > 1 && (pl.Ho % 2) == 0 && (pl.Wo % 2) == 0 &&
      pl.mtiles >= 8 && !tune_env("BTX_NO_PAR_MAJOR")) {
    const int tp = 64 * dma_nw;
    const long long mq = (long long)g->NB * (pl.Ho / 2) * (pl.Wo / 2);
    const long long tiles_per_class = (mq + tp - 1) / tp;  // the last tile of a class is padded: no tile holds two classes
    if (4 * tiles_per_class * pl.ntiles <= 0x7fffffffLL) {
      par_major = true;
      par_mqp = (int)(tiles_per_class * tp);
      pl.mtiles = (int)(4 * tiles_per_class);
      pl.ksplits = 1; pl.kper = pl.K;
      pl.nwg = pl.mtiles * pl.ntiles * g->groups;
    }
  }
  // stem variant: row-fused small-C convolutions with the input rows of the tile resident in LDS (BTX_NO_STEM=1 disables)
  static const bool no_stem = tune_env("BTX_NO_STEM") != nullptr;
  StemPlan stp;
  bool stem = false;
  if (dma && rowfuse && !no_stem) {
    Plan sp;
    if (!make_plan(g, prec, flags, DBM, &sp) && make_stem_plan(g, act_dtype, prec, sp, &stp)) {
      sp.ksplits = 1; sp.kper = sp.K; sp.nwg = stp.nwg;
      pl = sp;
      stem = true;
    }
  }
  // stem + max-pool (BtxEpilogue.pool): the band kernel of btx_contract_stempool.h or nothing
  StemPoolPlan spp;
  const bool want_pool = ep && ep->pool;
  if (want_pool) {
    Plan sp;
    if (ep->pool != 1 || !stem || ep->residual || (noise && (noise->sign_in || noise->sign_out)) ||
        (flags & (BTX_FLAG_OUT_F32 | BTX_FLAG_SWAP_SIGNS)) || make_plan(g, prec, flags, DBM, &sp) ||
        !make_stem_pool_plan(g, act_dtype, prec, sp, &spp, tune_env("BTX_STEM_SHORT_BANDS") ? 1 : lanes))
      return BTX_E_UNSUPPORTED;
    pl.nwg = spp.nwg;
  }
  // patch variant: stride-1 2-D convolutions keep the halo'd input patch of the tile in LDS (BTX_NO_PATCH=1 disables)
  static const bool no_patch = tune_env("BTX_NO_PATCH") != nullptr;
  PatchPlan pt;
  bool patch = false;
  if (dma && !rowfuse && !no_patch) {
    Plan pp;
    if (make_patch_plan(g, act_dtype, prec, flags, &pp, &pt, kind)) { pl = pp; patch = true; }
    else if (make_patch2_plan(g, act_dtype, prec, flags, &pp, &pt, kind)) { pl = pp; patch = true; }
  }
  int out_bf16 = (act_dtype == BTX_ACT_BF16) ? 1 : 0;
  if (flags & (BTX_FLAG_OUT_F32 | BTX_FLAG_OUT_BF16)) {
    if (!dma) return BTX_E_UNSUPPORTED;
    out_bf16 = (flags & BTX_FLAG_OUT_BF16) ? 1 : 0;
  }
  // Pointwise Flipout contractions with a long K on the 8-wave GEMM of btx_contract_gemm8.h: one workgroup per CU, a
  // 256-pixel x 128-channel tile, rings of four (conditions in that header).
  bool gemm8 = false;
  int g8_pairs = 1;
  {
    const char* mk = tune_env("BTX_GEMM8_MINK");
    const int min_k = mk ? atoi(mk) : BTX_GEMM8_MINK_DEFAULT;
    const int bk8 = NG * (prec == BTX_PREC_BF16 ? 8 : 4);  // (dma: the activation dtype is the contraction's)
    if (dma && !rowfuse && !patch && kind == BTX_KIND_FLIPOUT &&
        !(flags & BTX_FLAG_TRANSPOSED) && g->KD == 1 && g->KH == 1 && g->KW == 1 &&
        g->pd == 0 && g->ph == 0 && g->pw == 0 && (pl.K % bk8) == 0 && pl.K >= 4 * bk8 && pl.K >= min_k && (pl.Ng % 128) == 0 &&
        !tune_env("BTX_NO_GEMM8")) {
      const long long mt = (pl.M + 255) / 256;
      g8_pairs = pl.Ng / 128;
      const long long nwg = mt * g->groups * g8_pairs;
      if (nwg * lanes <= 0x7fffffffLL) {
        gemm8 = true;
        pl.mtiles = (int)mt; pl.ksplits = 1; pl.kper = pl.K; pl.nwg = (int)nwg;
      }
    }
  }
  // MEASUREMENT ONLY (tuning builds, BTX_PW=1): pointwise contractions (Linear, 1x1x1 / stride 1 / no padding) on the
  // Flipout-GEMM of btx_contract_pw.h — one workgroup per pixel tile walks `pw_ntb` n-tiles, store side from the fragment
  // registers.  Bit-identical to the LDS-DMA kernel; measured (profiles/r04_pointwise_ab.txt): +4..7 % where the activation
  // stages stay resident (K = 64), 7..16 % SLOWER where they stream — one workgroup per (pixel tile, n-tile) with the staged
  // store already overlaps prologue and store side across the two workgroups of a CU, and whole-line stores drain faster.
  bool pw = false;
  int pw_ntb = 1, pw_chunks = 1;
  if (tune_env("BTX_PW") && !gemm8 && dma && !rowfuse && !patch && dma_nw == 4 && !(flags & BTX_FLAG_TRANSPOSED) && g->KD == 1 && g->KH == 1 && g->KW == 1 &&
      g->sd == 1 && g->sh == 1 && g->sw == 1 && g->pd == 0 && g->ph == 0 && g->pw == 0 &&
      out_bf16 == (act_dtype == BTX_ACT_BF16 ? 1 : 0) && (pl.Ng % 64) == 0 && (g->N % 32) == 0 && !(noise && noise->sign_out) &&
      (long long)pl.M * g->N * (out_bf16 ? 2 : 4) < 0x7ff00000LL) {
    const long long mt = (pl.M + 255) / 256;
    const long long base = mt * g->groups * lanes;
    long long chunks = (1024 + base - 1) / base;  // at least two rounds of workgroups on the 512 slots when the n-tiles allow
    if (chunks > pl.ntiles) chunks = pl.ntiles;
    if (chunks < 1) chunks = 1;
    pw_ntb = (int)((pl.ntiles + chunks - 1) / chunks);
    pw_chunks = (pl.ntiles + pw_ntb - 1) / pw_ntb;
    const long long nwg = mt * g->groups * pw_chunks;
    if (nwg * lanes <= 0x7fffffffLL) {
      pw = true;
      pl.mtiles = (int)mt; pl.ksplits = 1; pl.kper = pl.K; pl.nwg = (int)nwg;
    }
  }
  size_t need = plan_ws(pl, g, lanes);
  // LDS-DMA and patch variants: the weights are sampled once per launch into the workspace (btx_presample.h),
  // behind the split-K partials
  size_t wt_off = 0, wt_one = 0, wt_all = 0;
  const void* sampled_w = (noise && noise->sampled_w) ? noise->sampled_w : nullptr;
  if (sampled_w && (((uintptr_t)sampled_w) & 15)) return BTX_E_ALIGN;
  if (dma) {
    wt_off = pad256(need);
    wt_all = patch_wt_bytes(pl, g, kind, prec, &wt_one, lanes);
    if (sampled_w && wt_all < 0xfff00000ULL) wt_off = need;  // tiles live in the caller's buffer
    if (wt_off + wt_all >= 0xfff00000ULL) {  // 32-bit offsets inside the descriptor: register-staged kernel instead
      if (rowfuse || (flags & (BTX_FLAG_OUT_F32 | BTX_FLAG_OUT_BF16))) return BTX_E_UNSUPPORTED;
      dma = patch = false;
      rc = make_plan(g, prec, flags, BM, &pl);
      if (rc) return rc;
      need = plan_ws(pl, g, lanes);
    } else if (!sampled_w) {
      need = wt_off + wt_all;
    }
  }
  if (need && (!ws || ws_bytes < need)) return BTX_E_WORKSPACE;
  if (need && (((uintptr_t)ws) & 15)) return BTX_E_ALIGN;
  // the persistent tap-unrolled kernel keeps its image-group queues (BTX_QUEUE_BYTES, zeroed per launch) behind everything
  // else; a caller whose workspace has no room for them gets the plain kernel
  const size_t queue_off = pad256(need);
  const bool queue_fits = ws && !(((uintptr_t)ws) & 15) && ws_bytes >= queue_off + BTX_QUEUE_BYTES;

  ContractParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.mu = mu_w; p.rho = rho_w; p.mu_b = mu_b; p.rho_b = rho_b; p.out = out;
  p.partial = pl.ksplits > 1 ? (float*)ws : nullptr;
  p.lanes = lanes; p.lane_nwg = pl.nwg; p.fd_lane_nwg = make_fastdiv((uint32_t)(pl.nwg > 0 ? pl.nwg : 1));
  if (lanes > 1) {
    p.lane_x = ln->x_stride; p.lane_out = ln->out_stride; p.lane_res = ln->res_stride;
    p.lane_partial = (long long)(plan_ws(pl, g, 1));
    if ((long long)pl.nwg * lanes > 0x7fffffffLL) return BTX_E_UNSUPPORTED;
  }
  if (noise) {
    p.eps_w = noise->eps_w; p.eps_b = noise->eps_b;
    p.sign_in = noise->sign_in; p.sign_out = noise->sign_out;
  }
  p.NB = g->NB; p.D = g->D; p.H = g->H; p.W = g->W; p.C = g->C; p.Cg = pl.Cg;
  p.Do = pl.Do; p.Ho = pl.Ho; p.Wo = pl.Wo; p.N = g->N; p.Ng = pl.Ng;
  p.KD = g->KD; p.KH = g->KH; p.KW = g->KW;
  p.sd = g->sd; p.sh = g->sh; p.sw = g->sw; p.pd = g->pd; p.ph = g->ph; p.pw = g->pw;
  p.dd = g->dd; p.dh = g->dh; p.dw = g->dw;
  p.M = pl.M; p.K = pl.K;
  p.mtiles = pl.mtiles; p.ntiles = pl.ntiles; p.groups = g->groups; p.ksplits = pl.ksplits; p.kper = pl.kper;
  p.transposed = (flags & BTX_FLAG_TRANSPOSED) ? 1 : 0;
  p.pointwise = (!p.transposed && g->KD == 1 && g->KH == 1 && g->KW == 1 && g->sd == 1 && g->sh == 1 && g->sw == 1 &&
                 g->pd == 0 && g->ph == 0 && g->pw == 0) ? 1 : 0;
  p.out_bf16 = out_bf16;
  if (ep) { p.ep_scale = ep->scale; p.ep_shift = ep->shift; p.ep_res = ep->residual; p.ep_relu = ep->relu; }
  if (rowfuse) {  // K = KH*(KW*C) unchanged; the pixel stride p.C stays C
    p.Cg = g->KW * g->C;
    p.KW = 1;
    p.sign_unaligned = 1;
  }
  p.fd_Cg = make_fastdiv((uint32_t)p.Cg); p.fd_KW = make_fastdiv((uint32_t)p.KW); p.fd_KH = make_fastdiv((uint32_t)p.KH);
  p.seed_lo = (uint32_t)rng->seed; p.seed_hi = (uint32_t)(rng->seed >> 32);
  p.sample = rng->sample_idx; p.layer = rng->layer_id;
  p.sample_ptr = rng->sample_idx_dev;
  // BTX_FLAG_SWAP_SIGNS (data gradient of a Flipout layer): the op's input carries the forward's s_out, its output the
  // forward's s_in
  p.fd_sd = make_fastdiv((uint32_t)g->sd); p.fd_sh = make_fastdiv((uint32_t)g->sh); p.fd_sw = make_fastdiv((uint32_t)g->sw);
  if (par_major && !patch && !gemm8 && !stem) {
    p.par_major = 1; p.par_Hh = pl.Ho / 2; p.par_Wh = pl.Wo / 2; p.par_Mq = g->NB * p.par_Hh * p.par_Wh; p.par_Mqp = par_mqp;
    p.fd_par_Mqp = make_fastdiv((uint32_t)p.par_Mqp); p.fd_par_Hh = make_fastdiv((uint32_t)p.par_Hh);
    p.fd_par_Wh = make_fastdiv((uint32_t)p.par_Wh);
  }
  p.swap_signs = (flags & BTX_FLAG_SWAP_SIGNS) ? 1 : 0;
  p.reverse = (flags & BTX_FLAG_REVERSE) ? 1 : 0;
  sign_keys(rng, p.swap_signs ? BTX_STREAM_SIGN_OUT : BTX_STREAM_SIGN_IN, &p.kin_a, &p.kin_b);
  sign_keys(rng, p.swap_signs ? BTX_STREAM_SIGN_IN : BTX_STREAM_SIGN_OUT, &p.kout_a, &p.kout_b);
  {
    const long long in_elems = (long long)g->NB * g->D * g->H * g->W * g->C;
    const long long xb = in_elems * (act_dtype == BTX_ACT_BF16 ? 2 : 4), wb = (long long)g->N * pl.K * 4;
    p.x_bytes = xb < 0xffffffffLL ? (uint32_t)xb : 0xffffffffu;
    p.w_bytes = wb < 0xffffffffLL ? (uint32_t)wb : 0xffffffffu;
  }

  if (const char* tp = tune_env("BTX_TRACE_PTR")) p.trace = (void*)strtoull(tp, nullptr, 0);  // BTX_PT_TRACE builds only
  p.pt_nopw = tune_env("BTX_NO_DMA_PW") ? 1 : 0;
  p.pt_nw = dma_nw;
  p.fd_inner = make_fastdiv((uint32_t)(pl.ntiles * g->groups * pl.ksplits)); p.fd_ksplits = make_fastdiv((uint32_t)pl.ksplits);
  p.fd_ntiles = make_fastdiv((uint32_t)pl.ntiles); p.fd_rtiles = make_fastdiv(1u);
  p.fd_mtiles = make_fastdiv((uint32_t)(pl.mtiles > 0 ? pl.mtiles : 1));
  {
    // XCD affinity of the workgroup order (block b runs on XCD b % 8; the remap gives every XCD a contiguous range of
    // logical ids): keep in one L2 whichever operand the neighbours would otherwise re-fetch more bytes of — the
    // activations of a pixel tile (read by its ntiles*ksplits workgroups) or the sampled weights of an (n-tile, k-split)
    // (read by its mtiles workgroups).  ResNet18 layer4 (7x7 maps, 512 channels): 9.4 MB of weight tiles against 3.2 MB
    // of activations; measured HBM-side traffic of that launch in round 1 order: 92 MB for 25 MB algorithmic.
    // Weight tiles up to ~3 MB stay resident in every XCD's 4-MB L2 whatever the order; beyond that the weight-major
    // order is what keeps them on chip.  (Round 4, 20 MC sample lanes per launch, rocprofv3 FETCH/WRITE_SIZE: ResNet18
    // layer3 — 2.36 MB of tiles per lane, 6.4 MB of activations — moves 2.98x its algorithmic bytes weight-major, where the
    // activations are re-fetched once per n-tile, and 2.00x pixel-major; the launch time is the same either way.)
    const double w_b = (prec == BTX_PREC_BF16 ? 2.0 : 4.0) * (double)g->N * pl.K * (kind == BTX_KIND_FLIPOUT ? 2 : 1);
    const char* wmb = tune_env("BTX_WG_MB");  // A/B: the weight-tile size (MiB) from which the order turns weight-major
    p.wg_order = (dma && pl.mtiles > 1 && w_b >= (wmb ? atof(wmb) : BTX_WG_MB_DEFAULT) * 1048576.0) ? 1 : 0;
    if (tune_env("BTX_WG_ORDER")) p.wg_order = atoi(tune_env("BTX_WG_ORDER"));
  }
  p.fd_Wo = make_fastdiv((uint32_t)pl.Wo); p.fd_Ho = make_fastdiv((uint32_t)pl.Ho); p.fd_Do = make_fastdiv((uint32_t)pl.Do);
  if (dma) {
    p.wt = sampled_w ? (void*)sampled_w : (void*)((unsigned char*)ws + wt_off);
    p.wt_ready = sampled_w ? 1 : 0;
    p.wt_bytes = (uint32_t)wt_all;
    p.wt_delta_off = (uint32_t)wt_one;
    p.lane_wt = (long long)wt_one;  // Flipout: lane l's delta tiles at wt_delta_off + l*lane_wt; else its W tiles at l*lane_wt
    p.lane_wt_delta = (kind == BTX_KIND_FLIPOUT) ? 1 : 0;
  }
  hipStream_t st = (hipStream_t)stream;
  if (want_pool) {
    p.pt_R = spp.PB; p.pt_rtiles = spp.bands; p.pt_Rp = spp.Rp; p.pt_astage = spp.astage; p.st_sbytes = spp.sbytes;
    p.pt_lds = spp.lds; p.pt_PP = spp.patch_bytes; p.sp_Hq = spp.Hq; p.sp_Wq = spp.Wq;
    p.fd_rtiles = make_fastdiv((uint32_t)spp.bands);
    rc = launch_stem_pool_bf16(kind, p, pl.nwg * lanes, st);
  } else if (stem) {
    p.pt_R = stp.R; p.pt_Rp = stp.Rp; p.pt_rtiles = stp.rtiles; p.pt_nw = stp.nw; p.pt_astage = stp.astage;
    p.st_sbytes = stp.sbytes; p.pt_lds = stp.lds; p.pt_PP = stp.patch_bytes;
    p.fd_rtiles = make_fastdiv((uint32_t)stp.rtiles);
    rc = (prec == BTX_PREC_BF16) ? launch_contract_stem_bf16(kind, p, pl.nwg * lanes, st)
         : (prec == BTX_PREC_BF16X3) ? launch_contract_stem_x3(kind, p, pl.nwg * lanes, st)
                                     : launch_contract_stem_f32(kind, p, pl.nwg * lanes, st);
  } else if (patch) {
    p.pt_G = pt.G; p.pt_R = pt.R; p.pt_Rp = pt.Rp; p.pt_Wp = pt.Wp; p.pt_PP = pt.PP; p.pt_NI = pt.NI;
    p.fd_ptWp = make_fastdiv((uint32_t)pt.Wp); p.fd_ptRp = make_fastdiv((uint32_t)pt.Rp); p.fd_ptR = make_fastdiv((uint32_t)pt.R);
    p.fd_rtiles = make_fastdiv((uint32_t)pt.rtiles);
    p.pt_rtiles = pt.rtiles; p.pt_nw = pt.nw; p.pt_mi = pt.mi; p.pt_astage = pt.astage; p.pt_lds = pt.lds;
    { const char* tn = tune_env("BTX_TAPS_TUNE"); p.pt_tune = tn ? atoi(tn) : 0; }
    p.pt_taps = pt.taps; p.pt_kg = pt.kg; p.pt_lds_g = pt.lds_g;
    p.pt_wide = (pt.taps == 33 || pt.taps == 332) ? pt.wide : 0;
    if (p.pt_wide) {  // the grid's n-tiles are pairs of weight tiles (p.ntiles stays the tile count of the weight layout)
      p.fd_ntiles = make_fastdiv((uint32_t)(pl.ntiles / 2));
      p.fd_inner = make_fastdiv((uint32_t)((pl.ntiles / 2) * g->groups * pl.ksplits));
    }
    p.pt_tall = pt.tall; p.pt_P = pt.P; p.pt_Wt = pt.Wt; p.pt_ncs = pt.ncs;
    p.fd_P = make_fastdiv((uint32_t)pt.P); p.fd_Wt = make_fastdiv((uint32_t)pt.Wt); p.fd_ncs = make_fastdiv((uint32_t)pt.ncs);
    // Persistent form of the tap-unrolled kernel (btx_contract_taps3.h): about two workgroups per CU, each walking one
    // tile position through the images of all lanes; the K loop runs across tile boundaries and the store side works
    // from the fragment registers.  bf16 in and out, 3x3, plain tiles, one K split, an even number of channel blocks,
    // whole 64-channel n-tiles, 32-aligned s_out words, hashed signs, no bias.  MEASUREMENT ONLY (BTX_PERSIST=1 in tuning
    // builds; the shipped library does not even contain the kernel): bit-identical results, 241 VGPRs and no scratch, a
    // 56x56 tile in 29k instead of 35k cycles — and the same launch time, because the chip, at its package power limit,
    // answers the higher matrix-pipe duty with a lower clock (profiles/r03_persistent_ab.txt, r03_power_probe.txt).
    {
      const int bk = NG * 8;
      const int ncb = pl.Cg / bk;
      const long long x_all = (long long)p.x_bytes + (long long)(lanes - 1) * (lanes > 1 ? ln->x_stride : 0);
      const bool ok = pt.taps == 33 && pt.kg == 1 && !pt.tall && (g->NB % pt.G) == 0 && pl.ksplits == 1 && prec == BTX_PREC_BF16 && act_dtype == BTX_ACT_BF16 &&
                      out_bf16 && (ncb % 2) == 0 && (pl.Ng % 64) == 0 && (g->N % 32) == 0 && !(noise && (noise->sign_in || noise->sign_out)) &&
                      (pt.astage / 16) >= 1024 && x_all < 0xfff00000LL && !mu_b && queue_fits && pt.lds + 16 <= 81920 &&
                      (long long)g->NB * pl.Do * pl.Ho * pl.Wo * g->N * 2 < 0x7ff00000LL && tune_env("BTX_PERSIST") && !tune_env("BTX_NO_PERSIST");
      if (ok) {
        static int n_cu = 0;
        if (!n_cu) {
          int dev = 0;
          hipDeviceProp_t prop;
          if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
          if (n_cu <= 0) n_cu = 256;
        }
        // a workgroup = one (row tile, n-tile, group) position x a range of image groups: `nseg` ranges per position
        const long long combos = (long long)pt.rtiles * pl.ntiles * g->groups;
        const long long igt = (long long)lanes * (g->NB / pt.G);
        long long nseg = (2LL * n_cu) / combos;
        if (nseg < 1) nseg = 1;
        if (nseg > igt) nseg = igt;
        if (combos * 4 <= (long long)BTX_QUEUE_BYTES) {
          p.pt_persist = (int)(combos * nseg);
          p.pt_queue = (uint32_t*)((unsigned char*)ws + queue_off);
          hipError_t e = hipMemsetAsync(p.pt_queue, 0, (size_t)combos * 4, st);
          if (e != hipSuccess) return (int)e;
        }
      }
    }
    // MEASUREMENT ONLY (tuning builds, BTX_DIRECT=1): the store side straight from the fragment registers (direct_epilogue,
    // btx_epilogue.h) for the tap-unrolled 4-wave kernel — bf16 in and out, one K split, whole aligned 64-channel tiles,
    // hashed s_out, offsets with an out-of-range value to spare.  Bit-identical; measured (profiles/r04_direct_store_ab.txt):
    // its 32-byte pieces drain more slowly than the staged side's whole 128-byte lines — 6.1k against 4.9k cycles on a
    // 56x56 tile, 6.5k against 8.3k on a tall strip — and the launch time does not move either way (+1.7 % .. -0.3 %).
    p.ep_direct = (tune_env("BTX_DIRECT") && pt.taps == 33 && pt.kg == 1 && pl.ksplits == 1 && prec == BTX_PREC_BF16 &&
                   act_dtype == BTX_ACT_BF16 && out_bf16 && (pl.Ng % 64) == 0 && (g->N % 32) == 0 && !(noise && noise->sign_out) &&
                   (long long)pl.M * g->N * 2 < 0x7ff00000LL) ? 1 : 0;
    rc = (prec == BTX_PREC_BF16) ? launch_contract_patch_bf16(kind, p, pl.nwg * lanes, st)
         : (prec == BTX_PREC_BF16X3) ? launch_contract_patch_x3(kind, p, pl.nwg * lanes, st)
                                     : launch_contract_patch_f32(kind, p, pl.nwg * lanes, st);
  } else if (dma && gemm8) {
    p.pt_rtiles = g8_pairs;
    p.fd_rtiles = make_fastdiv((uint32_t)g8_pairs); p.fd_inner = make_fastdiv((uint32_t)(g8_pairs * g->groups));
    // MEASUREMENT ONLY (tuning builds, BTX_G8_DIRECT=1): the store side from the fragment registers (direct_epilogue,
    // btx_epilogue.h) where its contract holds — outputs of the activation dtype, hashed s_out, 32-bit byte offsets with an
    // out-of-range value to spare (per lane).  Bit-identical and 5-25 % slower than the staged side.
    p.ep_direct = (tune_env("BTX_G8_DIRECT") && (out_bf16 != 0) == (prec == BTX_PREC_BF16) && (g->N % 32) == 0 &&
                   !(noise && noise->sign_out) && (long long)pl.M * g->N * (out_bf16 ? 2 : 4) < 0x7ff00000LL) ? 1 : 0;
    rc = (prec == BTX_PREC_BF16) ? launch_contract_gemm8_bf16(kind, p, pl.nwg * lanes, st)
         : (prec == BTX_PREC_BF16X3) ? launch_contract_gemm8_x3(kind, p, pl.nwg * lanes, st)
                                     : launch_contract_gemm8_f32(kind, p, pl.nwg * lanes, st);
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
  } else if (dma && pw) {
    p.pt_R = pw_ntb; p.pt_rtiles = pw_chunks;
    p.fd_rtiles = make_fastdiv((uint32_t)pw_chunks); p.fd_inner = make_fastdiv((uint32_t)(pw_chunks * g->groups));
    rc = (prec == BTX_PREC_BF16) ? launch_contract_pw_bf16(kind, p, pl.nwg * lanes, st)
         : (prec == BTX_PREC_BF16X3) ? launch_contract_pw_x3(kind, p, pl.nwg * lanes, st)
                                     : launch_contract_pw_f32(kind, p, pl.nwg * lanes, st);
#endif
  } else if (dma)
    rc = (prec == BTX_PREC_BF16) ? launch_contract_dma_bf16(kind, p, pl.nwg * lanes, st)
         : (prec == BTX_PREC_BF16X3) ? launch_contract_dma_x3(kind, p, pl.nwg * lanes, st)
                                     : launch_contract_dma_f32(kind, p, pl.nwg * lanes, st);
  else {
    // the register-staged fast kernel samples in registers and hashes its own s_in: explicit eps_w / sign_in need either
    // the LDS-DMA family above (pre-pass sampling, packed sign words) or the gather kernel
    // (BTX_PREC_BF16X3 has no register-staged form: such shapes run on the exact-f32 kernel, which is at least as accurate)
    const bool gen2 = gen || (noise && (noise->eps_w || noise->sign_in));
    rc = (prec == BTX_PREC_BF16) ? launch_contract_bf16(kind, act_dtype == BTX_ACT_BF16, gen2, p, pl.nwg * lanes, st)
                                 : launch_contract_f32(kind, act_dtype == BTX_ACT_BF16, gen2, p, pl.nwg * lanes, st);
  }
  if (rc) return rc;
  if (pl.ksplits > 1) {
    const long long total = (long long)pl.M * g->N;
    long long blocks = (total / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    {  // one launch for all MC sample lanes (blockIdx.y)
      const float* part = (const float*)ws;
      const unsigned char* r = (const unsigned char*)p.ep_res;
      if (out_bf16)
        hipLaunchKernelGGL(splitk_reduce_kernel<__bf16>, dim3((int)blocks, lanes), dim3(256), 0, st, part, (__bf16*)out, total,
                           pl.ksplits, g->N, p.ep_scale, p.ep_shift, (const __bf16*)r, p.ep_relu, p.lane_partial, p.lane_out,
                           p.lane_res);
      else
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3((int)blocks, lanes), dim3(256), 0, st, part, (float*)out, total,
                           pl.ksplits, g->N, p.ep_scale, p.ep_shift, (const float*)r, p.ep_relu, p.lane_partial, p.lane_out,
                           p.lane_res);
      rc = (int)hipGetLastError();
    }
  }
  return rc;
}

size_t btx_sampled_w_bytes(const BtxGeom* g, int kind, int prec) { return btx_sampled_w_bytes_lanes(g, kind, prec, 1); }

size_t btx_sampled_w_bytes_lanes(const BtxGeom* g, int kind, int prec, int lanes) {
  Plan pl;
  if (!g || lanes < 1 || lanes > 255 || make_plan(g, prec, 0, DBM, &pl)) return 0;
  size_t one = 0;
  const size_t tiles = patch_wt_bytes(pl, g, kind, prec, &one, lanes);
  // Flipout: + the sigma cache (f32 per weight, tile order) behind the tiles — what BTX_SAMPLE_SKIP_MU reads instead of rho
  return tiles + (kind == BTX_KIND_FLIPOUT ? one * (prec == BTX_PREC_BF16 ? 2 : 1) : 0);
}

int btx_sample_weights(const BtxSampleItem* items, int n_items, const BtxRng* rng, int prec, void* stream) {
  return btx_sample_weights_lanes(items, n_items, rng, prec, stream, 1, 0);
}

int btx_sample_weights_lanes(const BtxSampleItem* items, int n_items, const BtxRng* rng, int prec, void* stream, int lanes,
                             uint32_t sflags) {
  if (!items || !rng) return BTX_E_NULL;
  if (lanes < 1 || lanes > 255) return BTX_E_SHAPE;
  const uint64_t seed = rng->seed;
  const uint32_t sample_idx = rng->sample_idx;
  if (n_items <= 0) return n_items == 0 ? 0 : BTX_E_SHAPE;
  if (prec != BTX_PREC_F32 && prec != BTX_PREC_BF16 && prec != BTX_PREC_BF16X3) return BTX_E_DTYPE;
  for (int base = 0; base < n_items; base += PRESAMPLE_MAX_ITEMS) {
    PresampleBatch b;
    memset(&b, 0, sizeof(b));
    b.n = n_items - base < PRESAMPLE_MAX_ITEMS ? n_items - base : PRESAMPLE_MAX_ITEMS;
    b.seed_lo = (uint32_t)seed; b.seed_hi = (uint32_t)(seed >> 32); b.sample = sample_idx; b.sample_ptr = rng->sample_idx_dev;
    b.lanes = lanes; b.skip_mu = (sflags & BTX_SAMPLE_SKIP_MU) ? 1 : 0;
    uint32_t blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      const BtxSampleItem& s = items[base + i];
      if (!s.geom || !s.mu_w || !s.rho_w || !s.out) return BTX_E_NULL;
      if (s.kind != BTX_KIND_REPARAM && s.kind != BTX_KIND_FLIPOUT) return BTX_E_UNSUPPORTED;
      const bool remap = s.src_C != 0 || s.src_KW != 0;  // scalar reads: no alignment requirement on mu/rho
      if ((((uintptr_t)s.out) & 15) || (!remap && (((uintptr_t)s.mu_w | (uintptr_t)s.rho_w) & 15))) return BTX_E_ALIGN;
      Plan pl;
      int rc = make_plan(s.geom, prec, 0, DBM, &pl);
      if (rc) return rc;
      if (pl.K % (prec == BTX_PREC_BF16 ? 8 : 4)) return BTX_E_UNSUPPORTED;  // whole 16-byte granules of the tile image
      size_t one = 0;
      if (patch_wt_bytes(pl, s.geom, s.kind, prec, &one, lanes) >= 0xfff00000ULL) return BTX_E_UNSUPPORTED;
      PresampleItem& it = b.it[i];
      it.mu = s.mu_w; it.rho = s.rho_w; it.wt = (unsigned char*)s.out;
      it.delta_off = (uint32_t)one;
      it.nquads = (uint32_t)(s.geom->groups * pl.ntiles * 64) * ((uint32_t)pl.K >> 2);
      it.first_block = blocks;
      it.layer = s.layer_id;
      it.Ng = pl.Ng; it.K = pl.K; it.ntiles = pl.ntiles; it.kind = s.kind;
      if (s.src_C != 0 || s.src_KW != 0) {
        if (s.geom->groups != 1 || s.src_C <= 0 || s.src_KW <= 0 || s.src_C > s.geom->C || s.src_KW > s.geom->KW ||
            s.geom->C % 4)
          return BTX_E_UNSUPPORTED;
        it.Cp = s.geom->C; it.KWp = s.geom->KW; it.src_KW = s.src_KW; it.src_C = s.src_C;
      }
      uint32_t nb = lanes > 1 ? (it.nquads + 255u) / 256u : (it.nquads + 1023u) / 1024u;  // ~4 (quad, lane) pairs per thread
      if (nb < 1) nb = 1;
      if (nb > (lanes > 1 ? 4096u : 1024u)) nb = lanes > 1 ? 4096u : 1024u;
      blocks += nb;
    }
    b.total_blocks = blocks;
    int rc = (prec == BTX_PREC_BF16) ? launch_presample_batch_bf16(b, (hipStream_t)stream)
             : (prec == BTX_PREC_BF16X3) ? launch_presample_batch_x3(b, (hipStream_t)stream)
                                         : launch_presample_batch_f32(b, (hipStream_t)stream);
    if (rc) return rc;
  }
  return 0;
}

int btx_fill_eps(float* out, size_t n, const BtxRng* rng, uint32_t rng_stream, void* stream) {
  if (!out || !rng) return BTX_E_NULL;
  if (n == 0) return 0;
  size_t blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_eps_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, out, n,
                     (uint32_t)rng->seed, (uint32_t)(rng->seed >> 32), rng->sample_idx, rng->layer_id, rng_stream,
                     (const uint32_t*)rng->sample_idx_dev);
  return (int)hipGetLastError();
}

int btx_rho_grad(const float* dw, const float* rho, float* drho, size_t n, const BtxRng* rng, uint32_t rng_stream,
                 void* stream) {
  if (!dw || !rho || !drho || !rng) return BTX_E_NULL;
  if (n == 0) return 0;
  if (n > 0xfffffffcULL) return BTX_E_UNSUPPORTED;  // BTX-RNG v1 block index is 32 bits
  size_t blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rho_grad_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dw, rho, drho, n,
                     (uint32_t)rng->seed, (uint32_t)(rng->seed >> 32), rng->sample_idx, rng->layer_id, rng_stream,
                     (const uint32_t*)rng->sample_idx_dev);
  return (int)hipGetLastError();
}

int btx_fill_sign(int8_t* out, size_t n, const BtxRng* rng, uint32_t rng_stream, void* stream) {
  if (!out || !rng) return BTX_E_NULL;
  if (n == 0) return 0;
  uint32_t ka, kb;
  sign_keys(rng, rng_stream, &ka, &kb);
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fill_sign_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, out, n, ka, kb);
  return (int)hipGetLastError();
}

size_t btx_mc_packed_floats(int bs, int C) {
  if (bs <= 0 || C <= 0) return 0;
  return (size_t)2 * bs * C + (size_t)bs + 2;
}

int btx_rowfuse_pack(const void* x, int in_dtype, const int64_t* strides_ncHW, int NB, int C, int H, int W, void* out,
                     int out_dtype, int Hp, int Wp, int cp, int ph, int pw, void* stream) {
  if (!x || !out || !strides_ncHW) return BTX_E_NULL;
  if (NB <= 0 || C <= 0 || H <= 0 || W <= 0 || ph < 0 || pw < 0 || Hp < H + ph || Wp < W + pw) return BTX_E_SHAPE;
  if ((cp != 4 && cp != 8) || C > cp) return BTX_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const bool ib = in_dtype == BTX_ACT_BF16, ob = out_dtype == BTX_ACT_BF16;
  if ((!ib && in_dtype != BTX_ACT_F32) || (!ob && out_dtype != BTX_ACT_F32)) return BTX_E_DTYPE;
  if (ib && ob) return launch_rowfuse_pack<__bf16, __bf16>(x, out, NB, C, H, W, Hp, Wp, cp, ph, pw, strides_ncHW, st);
  if (ib && !ob) return launch_rowfuse_pack<__bf16, float>(x, out, NB, C, H, W, Hp, Wp, cp, ph, pw, strides_ncHW, st);
  if (!ib && ob) return launch_rowfuse_pack<float, __bf16>(x, out, NB, C, H, W, Hp, Wp, cp, ph, pw, strides_ncHW, st);
  return launch_rowfuse_pack<float, float>(x, out, NB, C, H, W, Hp, Wp, cp, ph, pw, strides_ncHW, st);
}

int btx_maxpool2d_cl(const void* x, void* out, int dtype, int NB, int H, int W, int C, int k, int stride, int pad,
                     void* stream) {
  if (!x || !out) return BTX_E_NULL;
  if (NB <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k) return BTX_E_SHAPE;
  if (C % 8) return BTX_E_UNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)out) & 15)) return BTX_E_ALIGN;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return BTX_E_SHAPE;
  const long long total = (long long)NB * Ho * Wo * (C / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 262144) blocks = 262144;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BTX_ACT_BF16)
    hipLaunchKernelGGL(maxpool2d_cl_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, st, (const __bf16*)x, (__bf16*)out, NB,
                       H, W, C, Ho, Wo, k, stride, pad, total);
  else if (dtype == BTX_ACT_F32)
    hipLaunchKernelGGL(maxpool2d_cl_kernel<float>, dim3((int)blocks), dim3(256), 0, st, (const float*)x, (float*)out, NB, H,
                       W, C, Ho, Wo, k, stride, pad, total);
  else
    return BTX_E_DTYPE;
  return (int)hipGetLastError();
}

int btx_maxpool2d_cl_train(const void* x, void* out, uint8_t* idx, int dtype, int NB, int H, int W, int C, int k, int stride, int pad,
                           void* stream) {
  if (!x || !out || !idx) return BTX_E_NULL;
  if (NB <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k) return BTX_E_SHAPE;
  if (C % 8 || k > 15) return BTX_E_UNSUPPORTED;  // the window position must fit a byte
  if ((((uintptr_t)x | (uintptr_t)out) & 15) || (((uintptr_t)idx) & 7)) return BTX_E_ALIGN;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return BTX_E_SHAPE;
  const long long total = (long long)NB * Ho * Wo * (C / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 262144) blocks = 262144;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BTX_ACT_BF16)
    hipLaunchKernelGGL(maxpool2d_cl_idx_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, st, (const __bf16*)x, (__bf16*)out, idx, NB, H, W,
                       C, Ho, Wo, k, stride, pad, total);
  else if (dtype == BTX_ACT_F32)
    hipLaunchKernelGGL(maxpool2d_cl_idx_kernel<float>, dim3((int)blocks), dim3(256), 0, st, (const float*)x, (float*)out, idx, NB, H, W, C,
                       Ho, Wo, k, stride, pad, total);
  else
    return BTX_E_DTYPE;
  return (int)hipGetLastError();
}

int btx_maxpool2d_cl_bwd(const void* dy, const uint8_t* idx, void* dx, int dtype, int NB, int H, int W, int C, int k, int stride,
                         int pad, void* stream) {
  if (!dy || !idx || !dx) return BTX_E_NULL;
  if (NB <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k) return BTX_E_SHAPE;
  if (C % 8 || k > 15) return BTX_E_UNSUPPORTED;
  if ((((uintptr_t)dy | (uintptr_t)dx) & 15) || (((uintptr_t)idx) & 7)) return BTX_E_ALIGN;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return BTX_E_SHAPE;
  const long long total = (long long)NB * H * W * (C / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 262144) blocks = 262144;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BTX_ACT_BF16)
    hipLaunchKernelGGL(maxpool2d_cl_bwd_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, st, (const __bf16*)dy, idx, (__bf16*)dx, NB, H, W,
                       C, Ho, Wo, k, stride, pad, total);
  else if (dtype == BTX_ACT_F32)
    hipLaunchKernelGGL(maxpool2d_cl_bwd_kernel<float>, dim3((int)blocks), dim3(256), 0, st, (const float*)dy, idx, (float*)dx, NB, H, W, C,
                       Ho, Wo, k, stride, pad, total);
  else
    return BTX_E_DTYPE;
  return (int)hipGetLastError();
}

int btx_avgpool_global_cl(const void* x, void* out, int dtype, int NB, int HW, int C, void* stream) {
  if (!x || !out) return BTX_E_NULL;
  if (NB <= 0 || HW <= 0 || C <= 0) return BTX_E_SHAPE;
  if (C % 8) return BTX_E_UNSUPPORTED;
  if (((uintptr_t)x) & 15) return BTX_E_ALIGN;
  const dim3 grid((C + 63) / 64, NB);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BTX_ACT_BF16)
    hipLaunchKernelGGL(avgpool_global_cl_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)x, (__bf16*)out, HW, C,
                       1.0f / (float)HW);
  else if (dtype == BTX_ACT_F32)
    hipLaunchKernelGGL(avgpool_global_cl_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)out, HW, C,
                       1.0f / (float)HW);
  else
    return BTX_E_DTYPE;
  return (int)hipGetLastError();
}

int btx_mc_accumulate(const void* logits, int bs, int C, int act_dtype, float kl, float* packed, void* stream) {
  return btx_mc_accumulate_lanes(logits, 1, bs, C, act_dtype, kl, packed, stream);
}

int btx_mc_accumulate_lanes(const void* logits, int lanes, int bs, int C, int act_dtype, float kl, float* packed,
                            void* stream) {
  if (!logits || !packed) return BTX_E_NULL;
  if (bs <= 0 || C <= 0 || lanes <= 0) return BTX_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  // lanes per LDS chunk: up to 96 KiB of probabilities.  Above the 64 KiB every kernel may use, the limit is an opt-in per
  // function AND per device: asked for once per device, and a device that refuses keeps 64 KiB chunks.
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const bool f32 = act_dtype == BTX_ACT_F32;
  if (!f32 && act_dtype != BTX_ACT_BF16) return BTX_E_DTYPE;
  static unsigned char big_lds[2][64];  // 0 not asked yet, 1 granted, 2 refused
  unsigned char& st_big = big_lds[f32 ? 0 : 1][dev];
  if (!st_big) {
    const void* fn = f32 ? (const void*)mc_accumulate_kernel<float> : (const void*)mc_accumulate_kernel<__bf16>;
    st_big = (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 98304 + 64) == hipSuccess) ? 1 : 2;
    if (st_big == 2) (void)hipGetLastError();
  }
  const size_t chunk = (st_big == 1) ? (size_t)98304 : (size_t)65536;
  const size_t per_lane = (size_t)C * 4 + 4;
  int LC = (int)(chunk / per_lane);
  if (LC < 1) return BTX_E_UNSUPPORTED;  // a row of > 24 575 classes does not fit a chunk (include/btx.h K6)
  if (LC > lanes) LC = lanes;
  const size_t lds = (size_t)LC * per_lane;
  if (f32)
    hipLaunchKernelGGL(mc_accumulate_kernel<float>, dim3(bs), dim3(1024), lds, st, (const float*)logits, bs, C, kl, packed,
                       lanes, LC);
  else
    hipLaunchKernelGGL(mc_accumulate_kernel<__bf16>, dim3(bs), dim3(1024), lds, st, (const __bf16*)logits, bs, C, kl,
                       packed, lanes, LC);
  return (int)hipGetLastError();
}

}  // extern "C"
