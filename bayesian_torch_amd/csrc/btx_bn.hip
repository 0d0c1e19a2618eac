// btx_bn.hip — training-mode BatchNorm on channels-last activations, forward and backward (gfx950).
//
// The step either side of the variational convolutions in the reference's training loop (README.md:114-125 run on
// models/deterministic/resnet_large.py:46-62: conv -> bn -> relu, `model.train()`): torch.nn.BatchNorm2d in training mode —
// batch statistics over (N, H, W) per channel, biased variance for the normalisation, unbiased for the running estimate,
// `running = (1 - momentum) * running + momentum * batch`.  Round-5 trace of the training step (tools/train_profile.py): the stock
// ATen kernels (batch_norm_collect_statistics / _backward_reduce / _backward_elemt / _transform_input, channels-last bf16) took
// 3.65 ms of a 10.6 ms step for 2.5 GB of traffic (0.7 TB/s).  These are plain HBM-bound passes:
//
//   forward   stats: one pass over x  -> per-block sums of (x - pivot), (x - pivot)^2 per channel in f32 (pivot = row 0) -> fixed-order f64 fold: mean, 1/sqrt(var + eps),
//             the running estimates, and the per-channel (scale, shift) of the apply pass
//             apply: y = x * scale[c] + shift[c]                                  (one read, one write)
//   backward  reduce: one pass over (x, dy) -> sum(dy), sum(dy * xhat) per channel -> dgamma, dbeta and the three coefficients of
//             dx = A[c] * dy + B[c] * x + D[c]                                    (apply: two reads, one write)
//
// A thread owns 8 consecutive channels of a row (16 bytes of bf16 / two 16-byte loads of f32): every access is a full 16-byte
// vector, a wave reads whole 128-byte lines.  C % 8 == 0, C <= 2048.  Deterministic: fixed block -> row mapping, fixed-order folds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/btx.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BN_MAX_BLOCKS = 512;

template <typename ACT>
__device__ __forceinline__ void load8(const ACT* p, float v[8]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <>
__device__ __forceinline__ void load8<__bf16>(const __bf16* p, float v[8]) {
  const u32x4 a = *(const u32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __builtin_bit_cast(float, a[i] << 16);
    v[2 * i + 1] = __builtin_bit_cast(float, a[i] & 0xffff0000u);
  }
}
template <typename ACT>
__device__ __forceinline__ void store8(ACT* p, const float v[8]);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
  *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
  *(f32x4*)(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ uint32_t bf16_rn(float f) {  // round to nearest even, as torch's float -> bfloat16
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN stays NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <>
__device__ __forceinline__ void store8<__bf16>(__bf16* p, const float v[8]) {
  u32x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = bf16_rn(v[2 * i]) | (bf16_rn(v[2 * i + 1]) << 16);
  *(u32x4*)p = a;
}

__device__ __forceinline__ float param_load(const void* p, int dtype, int c) {
  if (dtype == BTX_ACT_BF16) return __builtin_bit_cast(float, (uint32_t)((const uint16_t*)p)[c] << 16);
  return ((const float*)p)[c];
}
__device__ __forceinline__ void param_store(void* p, int dtype, int c, float v) {
  if (dtype == BTX_ACT_BF16) ((uint16_t*)p)[c] = (uint16_t)bf16_rn(v);
  else ((float*)p)[c] = v;
}

// Per-channel partial sums of two quantities over the rows of this block:  MODE 0: (x, x*x)   MODE 1: (dy, dy * (x - mean) * invstd)
// partial[blk][2][C] (f32).  Block = 256 threads = RPB rows x (C/8) channel groups per pass; rows of a block: blk, blk + nblk, ...
// MASK (MODE 1, a fused ReLU behind the normalisation): dy counts only where the forward's output was positive — one bit per element,
// the byte of this thread's 8 channels, written by the forward's apply pass.
template <typename ACT, int MODE, bool MASK = false>
__global__ __launch_bounds__(256) void bn_partial_kernel(const ACT* __restrict__ x, const ACT* __restrict__ dy,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         long long M, int C, float* __restrict__ partial,
                                                         const uint8_t* __restrict__ mask = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [256 threads][16 partial sums] = 16 KiB
  const int cg = C >> 3;              // channel groups per row
  const int rpb = 256 / cg > 0 ? 256 / cg : 1;  // rows per pass (cg <= 256)
  const int tid = threadIdx.x;
  const int g = tid % cg, r = tid / cg;
  const bool act = r < rpb;
  float s0[8], s1[8], mu[8], is[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s0[i] = 0.f; s1[i] = 0.f; mu[i] = 0.f; is[i] = 1.f; }
  if (MODE == 1 && act) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { mu[i] = mean[g * 8 + i]; is[i] = invstd[g * 8 + i]; }
  }
  // MODE 0: sums of (x - pivot) and (x - pivot)^2 with pivot = the channel's value in row 0: E[x^2] - E[x]^2 on raw f32 sums loses
  // the variance of a channel whose mean is large against its spread; shifted by any value of the channel it does not
  if (MODE == 0 && act) load8<ACT>(x + g * 8, mu);
  if (act) {
    // four rows in flight per thread (the pass is bound by how many 16-byte loads a CU keeps outstanding), summed in row order
    const long long stride = (long long)gridDim.x * rpb;
    for (long long row = (long long)blockIdx.x * rpb + r; row < M; row += 4 * stride) {
      float v[4][8], d[4][8];
      bool ok[4];
      uint32_t mk[4] = {0xffu, 0xffu, 0xffu, 0xffu};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long rw = row + u * stride;
        ok[u] = rw < M;
        if (ok[u]) {
          load8<ACT>(x + rw * C + g * 8, v[u]);
          if (MODE == 1) load8<ACT>(dy + rw * C + g * 8, d[u]);
          if (MASK) mk[u] = mask[rw * cg + g];
        }
      }
      if (MASK) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) d[u][i] = ((mk[u] >> i) & 1u) ? d[u][i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        if (MODE == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float dlt = v[u][i] - mu[i]; s0[i] += dlt; s1[i] = __builtin_fmaf(dlt, dlt, s1[i]); }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) { s0[i] += d[u][i]; s1[i] = __builtin_fmaf(d[u][i], (v[u][i] - mu[i]) * is[i], s1[i]); }
        }
      }
    }
  }
  // fold the rpb rows of the block per channel group, fixed order (row 0 + row 1 + ...), through LDS: [r][g][16]
  float* my = red + (size_t)tid * 16;
#pragma unroll
  for (int i = 0; i < 8; ++i) { my[i] = s0[i]; my[8 + i] = s1[i]; }
  __syncthreads();
  if (tid < cg) {
    float a0[8], a1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    for (int rr = 0; rr < rpb; ++rr) {
      const float* q = red + (size_t)(rr * cg + tid) * 16;
#pragma unroll
      for (int i = 0; i < 8; ++i) { a0[i] += q[i]; a1[i] += q[8 + i]; }
    }
    float* out = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int i = 0; i < 8; ++i) { out[tid * 8 + i] = a0[i]; out[C + tid * 8 + i] = a1[i]; }
  }
}

// fold partial[nblk][2][C] per channel in f64: 1024 threads = 64 channels x 16 slices (slice j takes blocks j, j + 16, ...), then the
// slices in order 0..15 — a fixed order, whatever the launch looks like.  Returns true in the thread that holds channel c's totals.
__device__ __forceinline__ bool bn_fold(const float* __restrict__ partial, int nblk, int C, int& c, double& s, double& q) {
  __shared__ double fold[16][64][2];
  const int cl = threadIdx.x & 63, j = threadIdx.x >> 6;
  c = blockIdx.x * 64 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
  {  // ALL of this slice's partials requested in one batch (the partials come from other XCDs' write-backs: every dependent
     // round of loads is an HBM-side round trip), then added in block order
    static_assert(BN_MAX_BLOCKS <= 16 * 32, "one batch of 32 loads per slice");
    float pa[32], pb[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int bb = min(j + 16 * u, nblk - 1);  // unconditional loads (a predicated load is waited for on the spot): a clamped
      pa[u] = partial[(size_t)bb * 2 * C + c];     // index re-reads the last block's row, dropped below
      pb[u] = partial[(size_t)bb * 2 * C + C + c];
    }
#pragma unroll
    for (int u = 0; u < 32; ++u)
      if (j + 16 * u < nblk) { a += (double)pa[u]; b += (double)pb[u]; }
  }
  fold[j][cl][0] = a; fold[j][cl][1] = b;
  __syncthreads();
  if (j != 0 || c >= C) return false;
  s = 0.0; q = 0.0;
  for (int k = 0; k < 16; ++k) { s += fold[k][cl][0]; q += fold[k][cl][1]; }
  return true;
}

// forward: fold the partials, statistics, running estimates, (scale, shift) of the apply pass
__global__ __launch_bounds__(1024) void bn_fwd_final_kernel(const float* __restrict__ partial, int nblk, long long M, int C,
                                                           const void* x_row0, int act_dtype,
                                                           const void* gamma, const void* beta, void* running_mean,
                                                           void* running_var, int pdt, float momentum, float eps,
                                                           float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                           float* __restrict__ scale, float* __restrict__ shift,
                                                           long long* __restrict__ batches_tracked) {
  if (batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1;  // nn.BatchNorm's num_batches_tracked
  int c;
  double s, q;
  if (!bn_fold(partial, nblk, C, c, s, q)) return;
  const double pivot = (double)param_load(x_row0, act_dtype, c);  // the sums are those of x - pivot (bn_partial_kernel)
  const double ms = s / (double)M;
  const double m = pivot + ms;
  double var = q / (double)M - ms * ms;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = (float)m;
  save_invstd[c] = invstd;
  const float g = gamma ? param_load(gamma, pdt, c) : 1.f, bb = beta ? param_load(beta, pdt, c) : 0.f;
  const float sc = g * invstd;
  scale[c] = sc;
  shift[c] = bb - (float)m * sc;
  if (running_mean) {
    const float rm = param_load(running_mean, pdt, c);
    param_store(running_mean, pdt, c, (1.f - momentum) * rm + momentum * (float)m);
  }
  if (running_var) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    const float rv = param_load(running_var, pdt, c);
    param_store(running_var, pdt, c, (1.f - momentum) * rv + momentum * (float)unb);
  }
}

// backward: dgamma, dbeta and the coefficients of dx = A*dy + B*x + D
__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(const float* __restrict__ partial, int nblk, long long M, int C,
                                                           const void* gamma, int pdt, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, void* __restrict__ dgamma,
                                                           void* __restrict__ dbeta, float* __restrict__ cA,
                                                           float* __restrict__ cB, float* __restrict__ cD) {
  int c;
  double s, q;
  if (!bn_fold(partial, nblk, C, c, s, q)) return;
  if (dbeta) param_store(dbeta, pdt, c, (float)s);   // in the parameters' dtype: what autograd hands the optimizer
  if (dgamma) param_store(dgamma, pdt, c, (float)q);
  const float g = gamma ? param_load(gamma, pdt, c) : 1.f;
  const double is = (double)invstd[c], mu = (double)mean[c], invM = 1.0 / (double)M;
  const double A = (double)g * is;
  const double B = -A * is * q * invM;      // coefficient of x:  -gamma * invstd^2 * mean(dy * xhat)
  cA[c] = (float)A;
  cB[c] = (float)B;
  cD[c] = (float)(-A * s * invM - B * mu);
}

// forward apply: y = a[c] * x + d[c]  [+ residual]  [ReLU, and the byte of positive outputs of these 8 channels for the backward]
template <typename ACT, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(const ACT* __restrict__ x, const ACT* __restrict__ res, ACT* __restrict__ out,
                                                           const float* __restrict__ ca, const float* __restrict__ cd,
                                                           uint8_t* __restrict__ mask, long long M, int C) {
  const int cg = C >> 3;
  const long long total = M * cg;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int g = (int)(t % cg);
    float v[8], o[8], a[8], d[8];
    load8<ACT>(x + t * 8, v);
    load8<float>(ca + g * 8, a);
    load8<float>(cd + g * 8, d);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf(a[i], v[i], d[i]);
    if (RES) {
      float r[8];
      load8<ACT>(res + t * 8, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += r[i];
    }
    if (RELU) {
      uint32_t m = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        o[i] = o[i] < 0.f ? 0.f : o[i];       // NaN stays NaN, as torch.relu
        m |= (o[i] > 0.f ? 1u : 0u) << i;     // torch's threshold_backward: the gradient passes where the output is > 0
      }
      mask[t] = (uint8_t)m;
    }
    store8<ACT>(out + t * 8, o);
  }
}

// backward apply: dx = a[c] * g + b[c] * x + d[c],  g = dy [where the forward's output was positive]; GOUT: g leaves too (the
// gradient of the residual branch)
template <typename ACT, bool MASK, bool GOUT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const ACT* __restrict__ x, const ACT* __restrict__ dy, ACT* __restrict__ out,
                                                           const float* __restrict__ ca, const float* __restrict__ cb,
                                                           const float* __restrict__ cd, const uint8_t* __restrict__ mask,
                                                           ACT* __restrict__ gout, long long M, int C) {
  const int cg = C >> 3;
  const long long total = M * cg;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int g = (int)(t % cg);
    float v[8], o[8], a[8], d[8], w[8], b[8];
    load8<ACT>(x + t * 8, v);
    load8<ACT>(dy + t * 8, w);
    load8<float>(ca + g * 8, a);
    load8<float>(cb + g * 8, b);
    load8<float>(cd + g * 8, d);
    if (MASK) {
      const uint32_t m = mask[t];
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = ((m >> i) & 1u) ? w[i] : 0.f;
      if (GOUT) store8<ACT>(gout + t * 8, w);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf(a[i], w[i], __builtin_fmaf(b[i], v[i], d[i]));
    store8<ACT>(out + t * 8, o);
  }
}

int bn_blocks(long long M, int C) {
  const int cg = C / 8, rpb = 256 / cg > 0 ? 256 / cg : 1;
  long long nb = (M + (long long)rpb * 8 - 1) / ((long long)rpb * 8);  // at least 8 passes per block
  if (nb > BN_MAX_BLOCKS) nb = BN_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  return (int)nb;
}

bool bn_ok(long long M, int C) { return M > 0 && C > 0 && (C % 8) == 0 && C <= 2048; }

}  // namespace

extern "C" {

// workspace: partial[nblk][2][C] + five coefficient vectors of C floats
size_t btx_bn_workspace_bytes(long long M, int C) {
  if (!bn_ok(M, C)) return 0;
  return ((size_t)bn_blocks(M, C) * 2 * C + (size_t)5 * C) * sizeof(float);
}

int btx_bn_train_fwd(const void* x, void* y, int act_dtype, long long M, int C, const void* gamma, const void* beta,
                     void* running_mean, void* running_var, int param_dtype, float momentum, float eps, float* save_mean,
                     float* save_invstd, long long* num_batches_tracked, const BtxBnFuse* fuse, void* ws, size_t ws_bytes,
                     void* stream) {
  if (!x || !y || !save_mean || !save_invstd || !ws) return BTX_E_NULL;
  if (!bn_ok(M, C)) return (M > 0 && C > 0) ? BTX_E_UNSUPPORTED : BTX_E_SHAPE;
  if (act_dtype != BTX_ACT_F32 && act_dtype != BTX_ACT_BF16) return BTX_E_DTYPE;
  if (param_dtype != BTX_ACT_F32 && param_dtype != BTX_ACT_BF16) return BTX_E_DTYPE;
  if (ws_bytes < btx_bn_workspace_bytes(M, C)) return BTX_E_WORKSPACE;
  const void* res = fuse ? fuse->residual : nullptr;
  const bool relu = fuse && fuse->relu;
  uint8_t* mask = fuse ? (uint8_t*)fuse->mask : nullptr;
  if (relu && !mask) return BTX_E_NULL;
  if ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)ws) | ((uintptr_t)res)) & 15) return BTX_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bn_blocks(M, C);
  float* partial = (float*)ws;
  float* scale = partial + (size_t)nblk * 2 * C;
  float* shift = scale + C;
  const size_t lds = 256 * 16 * sizeof(float);
  long long nap = (M * (C / 8) + 255) / 256;
  if (nap > 8192) nap = 8192;
  if (act_dtype == BTX_ACT_BF16) {
    hipLaunchKernelGGL((bn_partial_kernel<__bf16, 0>), dim3(nblk), dim3(256), lds, st, (const __bf16*)x, (const __bf16*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, M, C, partial, (const uint8_t*)nullptr);
  } else {
    hipLaunchKernelGGL((bn_partial_kernel<float, 0>), dim3(nblk), dim3(256), lds, st, (const float*)x, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, M, C, partial, (const uint8_t*)nullptr);
  }
  hipLaunchKernelGGL(bn_fwd_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, partial, nblk, M, C, x, act_dtype, gamma, beta, running_mean,
                     running_var, param_dtype, momentum, eps, save_mean, save_invstd, scale, shift, num_batches_tracked);
#define BTX_BN_FWD_APPLY(ACT, RES, RELU)                                                                                       \
  hipLaunchKernelGGL((bn_fwd_apply_kernel<ACT, RES, RELU>), dim3((unsigned)nap), dim3(256), 0, st, (const ACT*)x, (const ACT*)res, \
                     (ACT*)y, scale, shift, mask, M, C)
#define BTX_BN_FWD_PICK(ACT)                                               \
  do {                                                                     \
    if (res && relu) BTX_BN_FWD_APPLY(ACT, true, true);                    \
    else if (res) BTX_BN_FWD_APPLY(ACT, true, false);                      \
    else if (relu) BTX_BN_FWD_APPLY(ACT, false, true);                     \
    else BTX_BN_FWD_APPLY(ACT, false, false);                              \
  } while (0)
  if (act_dtype == BTX_ACT_BF16) BTX_BN_FWD_PICK(__bf16); else BTX_BN_FWD_PICK(float);
#undef BTX_BN_FWD_PICK
#undef BTX_BN_FWD_APPLY
  return (int)hipGetLastError();
}

int btx_bn_train_bwd(const void* x, const void* dy, void* dx, int act_dtype, long long M, int C, const void* gamma,
                     int param_dtype, const float* save_mean, const float* save_invstd, void* dgamma, void* dbeta,
                     const BtxBnFuse* fuse, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !dy || !dx || !save_mean || !save_invstd || !ws) return BTX_E_NULL;
  if (!bn_ok(M, C)) return (M > 0 && C > 0) ? BTX_E_UNSUPPORTED : BTX_E_SHAPE;
  if (act_dtype != BTX_ACT_F32 && act_dtype != BTX_ACT_BF16) return BTX_E_DTYPE;
  if (param_dtype != BTX_ACT_F32 && param_dtype != BTX_ACT_BF16) return BTX_E_DTYPE;
  if (ws_bytes < btx_bn_workspace_bytes(M, C)) return BTX_E_WORKSPACE;
  const bool relu = fuse && fuse->relu;
  const uint8_t* mask = fuse ? (const uint8_t*)fuse->mask : nullptr;
  void* dres = fuse ? fuse->dres : nullptr;
  if (relu && !mask) return BTX_E_NULL;
  if (dres && !relu) return BTX_E_UNSUPPORTED;  // without a ReLU the residual branch's gradient IS dy: nothing to write
  if ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)ws) | ((uintptr_t)dres)) & 15) return BTX_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bn_blocks(M, C);
  float* partial = (float*)ws;
  float* cA = partial + (size_t)nblk * 2 * C;
  float* cB = cA + C;
  float* cD = cB + C;
  const size_t lds = 256 * 16 * sizeof(float);
  long long nap = (M * (C / 8) + 255) / 256;
  if (nap > 8192) nap = 8192;
#define BTX_BN_PARTIAL(ACT, MASK)                                                                                               \
  hipLaunchKernelGGL((bn_partial_kernel<ACT, 1, MASK>), dim3(nblk), dim3(256), lds, st, (const ACT*)x, (const ACT*)dy, save_mean, \
                     save_invstd, M, C, partial, mask)
  if (act_dtype == BTX_ACT_BF16) { if (relu) BTX_BN_PARTIAL(__bf16, true); else BTX_BN_PARTIAL(__bf16, false); }
  else { if (relu) BTX_BN_PARTIAL(float, true); else BTX_BN_PARTIAL(float, false); }
#undef BTX_BN_PARTIAL
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, partial, nblk, M, C, gamma, param_dtype, save_mean,
                     save_invstd, dgamma, dbeta, cA, cB, cD);
#define BTX_BN_BWD_APPLY(ACT, MASK, GOUT)                                                                                      \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<ACT, MASK, GOUT>), dim3((unsigned)nap), dim3(256), 0, st, (const ACT*)x, (const ACT*)dy, \
                     (ACT*)dx, cA, cB, cD, mask, (ACT*)dres, M, C)
#define BTX_BN_BWD_PICK(ACT)                                               \
  do {                                                                     \
    if (relu && dres) BTX_BN_BWD_APPLY(ACT, true, true);                   \
    else if (relu) BTX_BN_BWD_APPLY(ACT, true, false);                     \
    else BTX_BN_BWD_APPLY(ACT, false, false);                              \
  } while (0)
  if (act_dtype == BTX_ACT_BF16) BTX_BN_BWD_PICK(__bf16); else BTX_BN_BWD_PICK(float);
#undef BTX_BN_BWD_PICK
#undef BTX_BN_BWD_APPLY
  return (int)hipGetLastError();
}

}  // extern "C"
