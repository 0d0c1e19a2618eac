/*
 * bt_oracle.c — CPU ORACLE for the variational-layer forward hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (bayesian_torch_amd + libbtx.so) never does.  It is a plain-C restatement of what the reference computes
 * (paths relative to /root/reference/bayesian_torch), written independently of the HIP kernels:
 *
 *   bto_softplus / bto_kl_mean   sigma = log1p(exp(rho))                layers/variational_layers/linear_variational.py:145,160
 *                                kl = log sp - log sq + (sq^2 + (mq-mp)^2)/(2 sp^2) - 1/2, .mean()
 *                                                                       layers/base_variational_layer.py:65-68
 *   bto_contract_fwd             Reparameterization: W = mu + sigma*eps ; out = conv(x, W, mu_b + sigma_b*eps_b)
 *                                                                       layers/variational_layers/conv_variational.py:361-380
 *                                                                       layers/variational_layers/linear_variational.py:160-178
 *                                Flipout: out = conv(x, mu, mu_b) + conv(x*s_in, sigma*eps, sigma_b*eps_b) * s_out
 *                                                                       layers/flipout_layers/conv_flipout.py:376-417
 *                                                                       layers/flipout_layers/linear_flipout.py:149-174
 *                                (conv = F.conv{1,2,3}d / F.conv_transpose{1,2,3}d / F.linear semantics, direct loops,
 *                                 f64 accumulation)
 *   bto_eps / bto_sign           the noise definition BTX-RNG v1 — NOT a reference algorithm: the
 *                                reference draws from torch's global generator; see DESIGN.md for why parity is
 *                                established through explicit noise + this restatement.
 *
 * Parity pinning: the reference has no tests / golden vectors.  tests/golden/ holds vectors generated
 * by importing the reference itself (tools/make_golden.py); tests/test_oracle.py checks this file against them.
 *
 * Layouts are the C-ABI's (include/btx.h): activations channels-last [NB][D][H][W][C], weights [N][tap][Cg],
 * output channels-last [NB][Do][Ho][Wo][N]; everything f32 on this side.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int32_t NB, D, H, W, C;
  int32_t N;
  int32_t KD, KH, KW;
  int32_t sd, sh, sw;
  int32_t pd, ph, pw;
  int32_t dd, dh, dw;
  int32_t od, oh, ow;
  int32_t groups;
} BtoGeom;

/* ------------------------------------------------------------------------------------------------------ */
float bto_softplus(float rho) { return log1pf(expf(rho)); }

/* round-to-nearest-even f32 -> bf16 -> f32 (what v_cvt_pk_bf16_f32 does) */
float bto_bf16_round(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return f; /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&f, &u, 4);
  return f;
}

double bto_kl_mean(const float* mu, const float* rho, size_t n, const float* pmu_t, const float* psig_t, float pmu,
                   float psig) {
  double acc = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const float sp = psig_t ? psig_t[i] : psig;
    const float mp = pmu_t ? pmu_t[i] : pmu;
    const float sq = bto_softplus(rho[i]);
    const float dm = mu[i] - mp;
    const float t = logf(sp) - logf(sq) + (sq * sq + dm * dm) / (2.0f * (sp * sp)) - 0.5f;
    acc += (double)t;
  }
  return acc / (double)n;
}

/* ------------------------------------------------------------------------------------------------------ */
/* BTX-RNG v1 */
static void philox4x32_10(const uint32_t c[4], const uint32_t k[2], uint32_t out[4]) {
  uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], k0 = k[0], k1 = k[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void bto_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4) {
  const uint32_t c[4] = {c0, c1, c2, c3}, k[2] = {k0, k1};
  philox4x32_10(c, k, out4);
}

static float u01(uint32_t x) { return fmaf((float)(x >> 8), 5.9604644775390625e-08f, 2.98023223876953125e-08f); }

static void box_muller(uint32_t xa, uint32_t xb, float* za, float* zb) {
  const double u1 = (double)u01(xa), u2 = (double)u01(xb);
  const double r = sqrt(-2.0 * log(u1));
  const double th = 6.283185307179586476925286766559 * u2;
  *za = (float)(r * cos(th));
  *zb = (float)(r * sin(th));
}

void bto_eps(float* out, size_t n, uint64_t seed, uint32_t sample, uint32_t layer, uint32_t stream) {
  const uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (size_t b = 0; 4 * b < n; ++b) {
    const uint32_t c[4] = {(uint32_t)b, sample, layer, stream};
    uint32_t x[4];
    float z[4];
    philox4x32_10(c, k, x);
    box_muller(x[0], x[1], &z[0], &z[1]);
    box_muller(x[2], x[3], &z[2], &z[3]);
    for (int e = 0; e < 4; ++e)
      if (4 * b + e < n) out[4 * b + e] = z[e];
  }
}

static uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

void bto_sign(int8_t* out, size_t n, uint64_t seed, uint32_t sample, uint32_t layer, uint32_t stream) {
  const uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const uint32_t c[4] = {0u, sample, layer, stream};
  uint32_t key[4];
  philox4x32_10(c, k, key);
  for (size_t i = 0; i < n; ++i) {
    const uint32_t w = mix32(mix32((uint32_t)(i >> 5) ^ key[0]) + key[1]);
    const uint32_t e = (uint32_t)i & 31u;
    const int bit = ((e & 1u) ? 31 : 15) - (int)(((e >> 3) << 2) + ((e & 7u) >> 1));
    out[i] = ((w >> bit) & 1u) ? -1 : 1;
  }
}

/* ------------------------------------------------------------------------------------------------------ */
int bto_out_shape(const BtoGeom* g, int transposed, int32_t* Do, int32_t* Ho, int32_t* Wo) {
  if (transposed) {
    *Do = (g->D - 1) * g->sd - 2 * g->pd + g->dd * (g->KD - 1) + g->od + 1;
    *Ho = (g->H - 1) * g->sh - 2 * g->ph + g->dh * (g->KH - 1) + g->oh + 1;
    *Wo = (g->W - 1) * g->sw - 2 * g->pw + g->dw * (g->KW - 1) + g->ow + 1;
  } else {
    *Do = (g->D + 2 * g->pd - g->dd * (g->KD - 1) - 1) / g->sd + 1;
    *Ho = (g->H + 2 * g->ph - g->dh * (g->KH - 1) - 1) / g->sh + 1;
    *Wo = (g->W + 2 * g->pw - g->dw * (g->KW - 1) - 1) / g->sw + 1;
  }
  return (*Do > 0 && *Ho > 0 && *Wo > 0) ? 0 : -1;
}

/* One variational contraction with EXPLICIT noise.
 *   kind 0: out = conv(x, mu + sp(rho)*eps_w) + (mu_b + sp(rho_b)*eps_b)
 *   kind 1: out = conv(x, mu) + mu_b + s_out * ( conv(x*s_in, sp(rho)*eps_w) + sp(rho_b)*eps_b )
 * bf16_inputs != 0: x (after the sign flip) and the sampled weights are rounded to bf16 before the f64-accumulated
 * products — the arithmetic of the bf16 MFMA path; biases stay f32.
 * out is written in f32 (caller rounds to bf16 if it compares a bf16 output). */
int bto_contract_fwd(int kind, const BtoGeom* g, int transposed, const float* x, const float* mu_w, const float* rho_w,
                     const float* mu_b, const float* rho_b, const float* eps_w, const float* eps_b,
                     const int8_t* sign_in, const int8_t* sign_out, int bf16_inputs, float* out) {
  int32_t Do, Ho, Wo;
  if (bto_out_shape(g, transposed, &Do, &Ho, &Wo)) return -1;
  const int Cg = g->C / g->groups, Ng = g->N / g->groups;
  const int T = g->KD * g->KH * g->KW;
  const size_t K = (size_t)T * Cg;
  /* sampled weights once */
  const size_t nw = (size_t)g->N * K;
  float* wm = (float*)malloc(nw * sizeof(float));
  float* wd = (float*)malloc(nw * sizeof(float));
  if (!wm || !wd) return -2;
  for (size_t i = 0; i < nw; ++i) {
    const float sg = bto_softplus(rho_w[i]);
    if (kind == 0) { wm[i] = mu_w[i] + sg * eps_w[i]; wd[i] = 0.f; }
    else { wm[i] = mu_w[i]; wd[i] = sg * eps_w[i]; }
    if (bf16_inputs) { wm[i] = bto_bf16_round(wm[i]); wd[i] = bto_bf16_round(wd[i]); }
  }
  for (int nb = 0; nb < g->NB; ++nb)
    for (int od = 0; od < Do; ++od)
      for (int oh = 0; oh < Ho; ++oh)
        for (int ow = 0; ow < Wo; ++ow) {
          const size_t opix = (((size_t)nb * Do + od) * Ho + oh) * Wo + ow;
          for (int n = 0; n < g->N; ++n) {
            const int grp = n / Ng;
            double am = 0.0, ad = 0.0;
            for (int kd = 0; kd < g->KD; ++kd)
              for (int kh = 0; kh < g->KH; ++kh)
                for (int kw = 0; kw < g->KW; ++kw) {
                  int id, ih, iw;
                  if (!transposed) {
                    id = od * g->sd - g->pd + kd * g->dd;
                    ih = oh * g->sh - g->ph + kh * g->dh;
                    iw = ow * g->sw - g->pw + kw * g->dw;
                  } else {
                    const int td = od + g->pd - kd * g->dd, th = oh + g->ph - kh * g->dh, tw = ow + g->pw - kw * g->dw;
                    if (td < 0 || th < 0 || tw < 0) continue;
                    if (td % g->sd || th % g->sh || tw % g->sw) continue;
                    id = td / g->sd; ih = th / g->sh; iw = tw / g->sw;
                  }
                  if (id < 0 || id >= g->D || ih < 0 || ih >= g->H || iw < 0 || iw >= g->W) continue;
                  const size_t ipix = (((size_t)nb * g->D + id) * g->H + ih) * g->W + iw;
                  const int tap = (kd * g->KH + kh) * g->KW + kw;
                  const float* xr = x + ipix * g->C + (size_t)grp * Cg;
                  const int8_t* sr = sign_in ? sign_in + ipix * g->C + (size_t)grp * Cg : NULL;
                  const float* wmr = wm + (size_t)n * K + (size_t)tap * Cg;
                  const float* wdr = wd + (size_t)n * K + (size_t)tap * Cg;
                  for (int c = 0; c < Cg; ++c) {
                    float xv = xr[c];
                    if (bf16_inputs) xv = bto_bf16_round(xv);
                    am += (double)wmr[c] * (double)xv;
                    if (kind == 1) ad += (double)wdr[c] * (double)(sr ? (sr[c] < 0 ? -xv : xv) : xv);
                  }
                }
            float val;
            if (kind == 0) {
              float b = 0.f;
              if (mu_b) b = mu_b[n] + bto_softplus(rho_b[n]) * eps_b[n];
              val = (float)am + b;
            } else {
              float bm = 0.f, bd = 0.f;
              if (mu_b) { bm = mu_b[n]; bd = bto_softplus(rho_b[n]) * eps_b[n]; }
              const float pert = (float)ad + bd;
              const float so = sign_out ? (float)sign_out[opix * g->N + n] : 1.f;
              val = ((float)am + bm) + pert * so;
            }
            out[opix * g->N + n] = val;
          }
        }
  free(wm);
  free(wd);
  return 0;
}

/* MC predictive accumulation restated (utils/util.py:41-60; examples/main_bayesian_imagenet_dnn2bnn.py:483-499).
 * packed: [bs*C sum p | bs*C sum p^2 | bs sum H | sum kl | count] */
void bto_mc_accumulate(const float* logits, int bs, int C, float kl, double* packed) {
  for (int r = 0; r < bs; ++r) {
    const float* lr = logits + (size_t)r * C;
    double mx = -INFINITY, se = 0.0, ent = 0.0;
    for (int c = 0; c < C; ++c) if (lr[c] > mx) mx = lr[c];
    for (int c = 0; c < C; ++c) se += exp((double)lr[c] - mx);
    for (int c = 0; c < C; ++c) {
      const double pr = exp((double)lr[c] - mx) / se;
      packed[(size_t)r * C + c] += pr;
      packed[(size_t)bs * C + (size_t)r * C + c] += pr * pr;
      ent -= pr * log(pr + 1e-15);
    }
    packed[(size_t)2 * bs * C + r] += ent;
  }
  packed[(size_t)2 * bs * C + bs] += kl;
  packed[(size_t)2 * bs * C + bs + 1] += 1.0;
}
