// btx_rng.h — BTX-RNG v1: counter-based noise for the variational hot path (device + host).
//
// The reference draws eps with eps_kernel.data.normal_() (layers/variational_layers/conv_variational.py:362)
// and the Flipout signs with x.clone().uniform_(-1,1).sign() (layers/flipout_layers/conv_flipout.py:385-386)
// from torch's global generator.  A global sequential stream cannot be regenerated tile-by-tile inside a
// GEMM, so the noise here is a PURE FUNCTION of (seed, sample_idx, layer_id, stream, element index):
//
//   eps(idx)  : Philox4x32-10, counter = (idx>>2, sample_idx, layer_id, stream), key = (seed_lo, seed_hi);
//               output word pair (x0,x1) -> Box–Muller -> lanes 0,1 ; (x2,x3) -> lanes 2,3 ; lane = idx&3.
//               u = fma(float(x>>8), 2^-24, 2^-25);  r = sqrt(-2 ln u1);  z = r*cos(2πu2), r*sin(2πu2).
//   sign(i)   : 32 signs per word.  (ka,kb) = Philox(counter=(0,sample_idx,layer_id,stream)).x[0..1];
//               word(wi) = mix32(mix32(wi ^ ka) + kb), wi = i>>5;
//               bit(e)   = ((e&1)?31:15) - (((e>>3)<<2) + ((e&7)>>1)),  e = i&31;   sign = bit set ? -1 : +1.
//               (The odd bit order makes the bf16 MFMA fragment mask a shift+and: see sign_mask_bf16.)
//
// oracle/c/bt_oracle.c restates this independently on the CPU; tests pin the two against each other.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BTX_HD __host__ __device__ __forceinline__
#else
#define BTX_HD static inline
#endif

struct BtxPhilox4 { uint32_t x[4]; };

BTX_HD BtxPhilox4 btx_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  BtxPhilox4 o; o.x[0] = c0; o.x[1] = c1; o.x[2] = c2; o.x[3] = c3;
  return o;
}

BTX_HD uint32_t btx_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// 32 Flipout signs for elements [32*wi, 32*wi+32).
BTX_HD uint32_t btx_sign_word(uint32_t wi, uint32_t ka, uint32_t kb) {
  return btx_mix32(btx_mix32(wi ^ ka) + kb);
}

BTX_HD int btx_sign_bitpos(uint32_t e) {  // e in [0,32)
  return ((e & 1u) ? 31 : 15) - (int)(((e >> 3) << 2) + ((e & 7u) >> 1));
}

#if defined(__HIPCC__)
// u in (0,1]: 24 random bits, centred.
__device__ __forceinline__ float btx_u01(uint32_t x) {
  return __builtin_fmaf((float)(x >> 8), 5.9604644775390625e-08f, 2.98023223876953125e-08f);
}

// Two standard normals from two 32-bit words (Box–Muller on the hardware transcendentals:
// v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 take their argument in revolutions).
__device__ __forceinline__ void btx_box_muller(uint32_t xa, uint32_t xb, float& za, float& zb) {
  const float u1 = btx_u01(xa);
  const float u2 = btx_u01(xb);
  const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln2 * log2(u1)
  za = r * __builtin_amdgcn_cosf(u2);
  zb = r * __builtin_amdgcn_sinf(u2);
}

// Four normals for the aligned index group [4*blk, 4*blk+4).
__device__ __forceinline__ void btx_normal4(uint32_t blk, uint32_t sample, uint32_t layer, uint32_t stream,
                                            uint32_t k0, uint32_t k1, float z[4]) {
  const BtxPhilox4 p = btx_philox4x32_10(blk, sample, layer, stream, k0, k1);
  btx_box_muller(p.x[0], p.x[1], z[0], z[1]);
  btx_box_muller(p.x[2], p.x[3], z[2], z[3]);
}

// One normal for an arbitrary index (slow path).
__device__ __forceinline__ float btx_normal1(uint64_t idx, uint32_t sample, uint32_t layer, uint32_t stream,
                                             uint32_t k0, uint32_t k1) {
  float z[4];
  btx_normal4((uint32_t)(idx >> 2), sample, layer, stream, k0, k1, z);
  const uint32_t l = (uint32_t)idx & 3u;
  return l == 0 ? z[0] : (l == 1 ? z[1] : (l == 2 ? z[2] : z[3]));
}

// Same noise / softplus on the RAW hardware transcendentals (v_exp_f32, v_log_f32, v_rcp_f32, v_sqrt_f32: ~1 ulp,
// no denormal fix-up code, no IEEE division sequence): what the MFMA kernels use in their inner loop.  Results
// differ from the functions above by a few ulp, far inside the parity tolerances (tests/test_gpu_contract.py).
__device__ __forceinline__ void btx_box_muller_hw(uint32_t xa, uint32_t xb, float& za, float& zb) {
  const float u1 = btx_u01(xa);
  const float u2 = btx_u01(xb);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  za = r * __builtin_amdgcn_cosf(u2);
  zb = r * __builtin_amdgcn_sinf(u2);
}
__device__ __forceinline__ void btx_normal4_hw(uint32_t blk, uint32_t sample, uint32_t layer, uint32_t stream,
                                               uint32_t k0, uint32_t k1, float z[4]) {
  const BtxPhilox4 p = btx_philox4x32_10(blk, sample, layer, stream, k0, k1);
  btx_box_muller_hw(p.x[0], p.x[1], z[0], z[1]);
  btx_box_muller_hw(p.x[2], p.x[3], z[2], z[3]);
}
// log1p(exp(rho)) = log(u) * e / (u - 1), u = 1 + e (e == u-1 exactly unless 1+e rounded: the ratio repairs it).
// exp underflow (rho < -87) gives e = 0 -> 0; rho > 88.7 gives +inf like the reference's naive form.
__device__ __forceinline__ float btx_softplus_hw(float rho) {
  const float e = __builtin_amdgcn_exp2f(rho * 1.4426950408889634f);
  const float u = 1.0f + e;
  const float d = u - 1.0f;
  const float lg = __builtin_amdgcn_logf(u) * 0.6931471805599453f;
  float sp = lg * (e * __builtin_amdgcn_rcpf(d));
  sp = (d == 0.0f) ? e : sp;
  sp = (e > 3.0e38f) ? e : sp;
  return sp;
}

// softplus exactly as the reference spells it, log1p(exp(rho)) (conv_variational.py:361), on the fast
// transcendentals.  log1p via u=1+e: log(u)*e/(u-1) (exact-rounding-compensated); rho > 88.72 -> +inf as
// the reference's naive form does.
__device__ __forceinline__ float btx_softplus_fast(float rho) {
  const float e = __expf(rho);
  const float u = 1.0f + e;
  const float d = u - 1.0f;
  float sp = (d == 0.0f) ? e : __logf(u) * __fdividef(e, d);
  sp = (e > 3.0e38f) ? e : sp;   // e == inf -> inf
  return sp;
}
#endif
