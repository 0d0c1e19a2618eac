/*
 * btx.h — C-ABI of libbtx.so: the MI355X (gfx950) variational-layer forward hot path.
 *
 * This is the drop-in boundary ("C-ABI face").  The reference
 * (IntelLabs/bayesian-torch) has no FFI of its own: its boundary is the Python nn.Module surface,
 * and every entry point below replaces a *chain of ATen ops* inside one reference method.  Each
 * declaration cites the reference lines it subsumes (paths relative to /root/reference/bayesian_torch).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers / sizes / scalars, no torch types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - returns 0 on success; >0 = hipError_t from the runtime; <0 = BTX_E_* argument error.
 *   - never throws, never allocates device memory, never synchronises: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the legacy default stream).  The caller owns every
 *     buffer and keeps it alive until the stream has passed the call.
 *   - stateless => thread-safe for distinct streams/buffers.
 *
 * Data layout:
 *   activations  channels-last: x[nb][d][h][w][c]   (f32 or bf16, `act_dtype`)
 *   weights      GEMM-major  : mu/rho[n][tap][c] f32, tap = (kd*KH + kh)*KW + kw, c within the group
 *                (== torch channels_last physical layout of the logical [Cout,Cin/g,k...] tensor;
 *                 for Linear it is the native [out][in]).
 *   output       channels-last: out[nb][do][ho][wo][n], same dtype as the activations.
 *
 * Noise definition (BTX-RNG v1): eps and the Flipout signs are pure functions of
 * (seed, sample_idx, layer_id, stream, logical element index) — Philox4x32-10 + Box–Muller for eps,
 * a keyed 32-bit mixer for the 1-bit signs — so a regenerated tile is identical in every workgroup
 * and on every rank.  Passing explicit noise tensors (eps_w, eps_b, sign_in, sign_out) overrides it;
 * that is the parity mode used against torch-drawn noise.
 */
#ifndef BTX_H_
#define BTX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BTX_ABI_VERSION 8

/* argument-error codes (negative) */
#define BTX_E_NULL        (-1)   /* required pointer is NULL */
#define BTX_E_SHAPE       (-2)   /* inconsistent / non-positive shape */
#define BTX_E_UNSUPPORTED (-3)   /* valid request this build does not implement */
#define BTX_E_WORKSPACE   (-4)   /* workspace too small (see btx_*_workspace_bytes) */
#define BTX_E_DTYPE       (-5)   /* unknown dtype / precision code */
#define BTX_E_ALIGN       (-6)   /* pointer not 16-byte aligned */

/* kind */
#define BTX_KIND_REPARAM 0
#define BTX_KIND_FLIPOUT 1
/* activation dtype */
#define BTX_ACT_F32  0
#define BTX_ACT_BF16 1
/* contraction precision */
#define BTX_PREC_F32  0   /* v_mfma_f32_32x32x2_f32: exact f32 fma chain (parity mode) */
#define BTX_PREC_BF16 1   /* v_mfma_f32_32x32x16_bf16, f32 accumulate (throughput mode) */
#define BTX_PREC_BF16X3 2 /* split-bf16: f32 activations, every operand as hi + lo bf16 (hi = rn(v), lo = rn(v - hi)), three
                             v_mfma_f32_32x32x16_bf16 per product (hi*hi + hi*lo + lo*hi), f32 accumulate: the f32 result of
                             the reference's F.conv*d / F.linear to ~1e-6 rel-L2 per layer (the tolerance north_star states is
                             1e-4) at a third of the bf16 matrix rate instead of a sixteenth.  Kernel family and workspace
                             sizes are those of BTX_PREC_F32; sampled-weight tiles (btx_sample_weights) are specific to it.
                             RANGE: operands must be finite and below the bf16 maximum (|v| < 3.39e38): for +-inf, or a
                             value whose hi rounds to inf, lo = rn(v - hi) = inf - inf = NaN and the product is NaN where the
                             reference's f32 convolution gives inf or a finite value (the split is not guarded: it sits in
                             the fragment-read path of the hot loop).  BTX_PREC_F32 has no such limit. */
/* flags */
#define BTX_FLAG_TRANSPOSED   1u  /* ConvTranspose gather rule */
#define BTX_FLAG_KL_ACCUM     2u  /* btx_kl_gauss: add to *kl_out instead of overwriting */
#define BTX_FLAG_ROWFUSE      4u  /* small-C stems: the KW taps of a kernel row are contiguous with the channels in
                                     memory (C*KW elements per row), so one K-stage = one kernel row.  Needs
                                     groups=1, dil_w=1, pad_w=0 (padding materialised by the caller), C*KW a multiple
                                     of the stage (32 bf16 / 16 f32 elements), activation dtype == precision dtype;
                                     returns BTX_E_UNSUPPORTED otherwise. */
#define BTX_FLAG_OUT_F32      8u  /* store the output as f32 / bf16 regardless of act_dtype (LDS-DMA kernels only: lets */
#define BTX_FLAG_OUT_BF16    16u  /* a caller that had to copy the input anyway keep "f32 in/out, bf16 MFMA" semantics) */
#define BTX_FLAG_SWAP_SIGNS  64u  /* Flipout: hash the INPUT signs from stream SIGN_OUT and the OUTPUT signs from stream
                                     SIGN_IN.  The data gradient of a Flipout layer is itself a Flipout-shaped contraction,
                                     dx = convT(dy, mu) + s_in * convT(dy * s_out, sigma*eps): the same entry point computes
                                     it on the transposed (or flipped) geometry with the two sign streams exchanged. */
#define BTX_FLAG_CONCURRENT 128u  /* hint: other launches run beside this one (several MC samples in flight on their own
                                     streams): plan for device throughput — CU-time — rather than for the latency of this
                                     launch, i.e. do not split K through HBM just to fill idle CUs */
#define BTX_FLAG_REVERSE    256u  /* walk the launch's workgroup tiles in DESCENDING order.  Results are identical (no launch
                                     depends on its block order); a caller that chains layers alternates the flag so that each
                                     launch starts on the activations its producer wrote LAST — the ones most likely still in
                                     the 256-MB Infinity Cache when the tensors of a launch (MC sample lanes: hundreds of MB)
                                     exceed it. */
#define BTX_FLAG_GATHER      32u  /* force the element-wise gather kernel (any shape / alignment; samples in registers).
                                     The library picks it by itself whenever a fast kernel does not apply; the flag
                                     exists so tests can exercise it on shapes the fast kernels would take. */

/* MC sample lanes (btx_contract_fwd_lanes): n independent Monte-Carlo samples of the SAME layer in one launch.  Whoever
 * asks for a workspace size or a pool shape for such a launch ORs BTX_FLAG_LANES(n) into the flags (the entry point sets
 * it from BtxLanes.n itself). */
#define BTX_FLAG_LANES_SHIFT 16
#define BTX_FLAG_LANES_MASK  (0xffu << BTX_FLAG_LANES_SHIFT)
#define BTX_FLAG_LANES(n)    (((uint32_t)(n) & 0xffu) << BTX_FLAG_LANES_SHIFT)

/* RNG streams of BTX-RNG v1 */
#define BTX_STREAM_EPS_W    0u
#define BTX_STREAM_EPS_B    1u
#define BTX_STREAM_SIGN_IN  2u
#define BTX_STREAM_SIGN_OUT 3u

/* Geometry of one variational contraction.  Linear is the 1x1x1 case: NB=batch, D=H=W=1, C=in_features,
 * KD=KH=KW=1, N=out_features.  Conv1d uses H only... by convention lower-rank convs put their axes last
 * (Conv1d: D=H=1, W=L;  Conv2d: D=1). */
typedef struct BtxGeom {
  int32_t NB, D, H, W, C;        /* input  [NB][D][H][W][C]  (C = total input channels) */
  int32_t N;                     /* total output channels */
  int32_t KD, KH, KW;            /* kernel taps */
  int32_t sd, sh, sw;            /* stride */
  int32_t pd, ph, pw;            /* padding */
  int32_t dd, dh, dw;            /* dilation */
  int32_t od, oh, ow;            /* output_padding (transposed only; else 0) */
  int32_t groups;
} BtxGeom;

typedef struct BtxRng {
  uint64_t seed;
  uint32_t sample_idx;           /* global Monte-Carlo sample index */
  uint32_t layer_id;
  const uint32_t* sample_idx_dev; /* optional DEVICE pointer: when non-NULL the kernels read the sample index from it
                                     when they run and sample_idx is ignored.  This is what lets one captured hipGraph
                                     of a whole MC forward be replayed for successive samples (update the word, replay):
                                     no per-launch host work in the Monte-Carlo loop.  Honoured by every entry point that
                                     takes a BtxRng except btx_fill_sign (ABI 7: also btx_contract_wgrad, btx_rho_grad,
                                     btx_fill_eps, btx_dgrad_weights — a whole TRAINING step can be captured). */
} BtxRng;

/* Optional explicit noise (parity mode).  Any member may be NULL => generated by BTX-RNG v1.
 *   eps_w   f32, same layout as mu_w ([N][tap][c])
 *   eps_b   f32 [N]
 *   sign_in  int8 (+1/-1), same layout as x      (Flipout only)
 *   sign_out int8 (+1/-1), same layout as out    (Flipout only) */
typedef struct BtxNoise {
  const float*  eps_w;
  const float*  eps_b;
  const int8_t* sign_in;
  const int8_t* sign_out;
  const void*   sampled_w;   /* weight tiles filled by btx_sample_weights for the same (seed, sample_idx, layer_id,
                                kind, prec, geometry): the launch skips its own sampling pre-pass.  Not explicit noise:
                                the values are BTX-RNG v1's.  Ignored by the kernels that sample in-register. */
} BtxNoise;

int         btx_abi_version(void);
const char* btx_strerror(int code);

/* K1. Mean Gaussian KL(q||p) of one parameter tensor, sigma_q = log1p(exp(rho)).
 * Replaces: BaseVariationalLayer_.kl_div  layers/base_variational_layer.py:53-68  and the softplus in
 * *.kl_loss()  (layers/variational_layers/linear_variational.py:144-155, layers/flipout_layers/conv_flipout.py:362-368).
 * prior_mu_t / prior_sigma_t: optional full-shape prior tensors (utils/util.py:102-117 MOPED overwrites
 * them); when NULL the scalars are used.  kl_out: 1 float, overwritten (or += with BTX_FLAG_KL_ACCUM).
 * Deterministic two-pass reduction; ws must hold btx_kl_workspace_bytes(n). */
size_t btx_kl_workspace_bytes(size_t n);
int btx_kl_gauss(const float* mu, const float* rho, size_t n,
                 const float* prior_mu_t, const float* prior_sigma_t,
                 float prior_mu, float prior_sigma,
                 float* kl_out, uint32_t flags, void* ws, size_t ws_bytes, void* stream);

/* K1b. KL of a whole model in one launch (+ one final reduce), and its gradient.
 * Replaces get_kl_loss  models/dnn_to_bnn.py:157-165  (sum over modules of the per-tensor MEANS) and, for training
 * (README.md:114-125: loss = ce + kl / batch_size), the autograd graph torch builds through base_variational_layer.py:53-68:
 *   d kl / d mu  = (mu - mu_p) / (sigma_p^2 n)        d kl / d rho = (sigma / sigma_p^2 - 1 / sigma) * sigmoid(rho) / n
 * items_host is a HOST array read before the call returns; mu/rho (and dmu/drho of the backward) are device pointers of
 * n floats in any element order (the terms are elementwise), prior tensors optional as in btx_kl_gauss.
 * grad_out: DEVICE pointer to the upstream gradient scalar (no host sync). */
typedef struct BtxKlItem {
  const float* mu;
  const float* rho;
  const float* prior_mu_t;     /* nullable */
  const float* prior_sigma_t;  /* nullable */
  float*       dmu;            /* backward only */
  float*       drho;           /* backward only */
  float        prior_mu, prior_sigma;
  size_t       n;
} BtxKlItem;
size_t btx_kl_model_workspace_bytes(int n_items);
int btx_kl_gauss_model(const BtxKlItem* items_host, int n_items, float* kl_out, void* ws, size_t ws_bytes, void* stream);
int btx_kl_gauss_model_bwd(const BtxKlItem* items_host, int n_items, const float* grad_out, void* stream);

/* K2-K5. Fused sample-and-contract forward of one variational layer.
 * Replaces the whole body of
 *   LinearReparameterization.forward   layers/variational_layers/linear_variational.py:157-178
 *   LinearFlipout.forward              layers/flipout_layers/linear_flipout.py:145-174
 *   Conv{1,2,3}dReparameterization.forward  layers/variational_layers/conv_variational.py:183-227, 357-380, 530-574
 *   Conv{1,2,3}dFlipout.forward        layers/flipout_layers/conv_flipout.py:175-244, 370-417, 568-637
 *   ConvTranspose{1,2,3}d{Reparameterization,Flipout}.forward  conv_variational.py:577-1094, conv_flipout.py:640-1228
 * (sigma=log1p(exp(rho)); eps~N(0,1); W=mu+sigma*eps | y0=conv(x,mu)+conv(x*s_in, sigma*eps)*s_out; + bias).
 * mu_b/rho_b NULL => no bias.  The KL term of these methods is btx_kl_gauss (RNG-free).
 * ws: split-K partial sums; must hold btx_contract_workspace_bytes(...) bytes (may be 0). */
size_t btx_contract_workspace_bytes(const BtxGeom* g, int kind, int act_dtype, int prec, uint32_t flags);
int btx_contract_fwd(int kind, const BtxGeom* g,
                     const void* x, const float* mu_w, const float* rho_w,
                     const float* mu_b, const float* rho_b,
                     void* out,
                     const BtxRng* rng, const BtxNoise* noise /* nullable */,
                     int act_dtype, int prec, uint32_t flags,
                     void* ws, size_t ws_bytes, void* stream);

/* §8(f)-3, the step either side of the path: eval-mode BatchNorm (+ residual add, + ReLU) folded into the store of the
 * contraction (reference models/deterministic/resnet_large.py:49-60: out = relu(bn(conv(x)) [+ identity])).
 *   y = conv_out * scale[n] + shift[n]  (+ residual[same index as out])  ;  y = max(y, 0) if relu
 * scale/shift: f32 [N] (NULL => 1 / 0); residual: same layout and dtype as `out` (NULL => none).
 * pool = 1: the ResNet stem's nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (resnet_large.py:118,145) is applied to y
 * inside the same launch and `out` is the POOLED tensor [NB][Hq][Wq][N] (btx_contract_pool_shape); the conv output never
 * reaches HBM.  Row-fused bf16 stems only (BTX_FLAG_ROWFUSE, generated noise, no residual): everything else returns
 * BTX_E_UNSUPPORTED and the caller pools with btx_maxpool2d_cl.  Values are bit-identical to the two-launch chain. */
typedef struct BtxEpilogue {
  const float* scale;
  const float* shift;
  const void*  residual;
  int32_t      relu;
  int32_t      pool;
} BtxEpilogue;
/* 1 (and the pooled extent) when btx_contract_fwd_ex would take epilogue.pool = 1 for this geometry, else 0 */
int btx_contract_pool_shape(const BtxGeom* g, int act_dtype, int prec, uint32_t flags, int32_t* Hq, int32_t* Wq);
int btx_contract_fwd_ex(int kind, const BtxGeom* g,
                        const void* x, const float* mu_w, const float* rho_w,
                        const float* mu_b, const float* rho_b,
                        void* out,
                        const BtxRng* rng, const BtxNoise* noise /* nullable */,
                        int act_dtype, int prec, uint32_t flags,
                        void* ws, size_t ws_bytes, void* stream,
                        const BtxEpilogue* epilogue /* nullable */);

/* MC sample lanes.  The Monte-Carlo loop of the reference (examples/main_bayesian_imagenet_dnn2bnn.py:480-499:
 * `for mc_run in range(num_monte_carlo): output = model(images)`) evaluates the same layer once per sample with fresh
 * noise; on a 256-CU GPU one sample of a 7x7 or 14x14 layer is a few hundred workgroups — too few to fill the chip and
 * to overlap one workgroup's prologue / store with another's MFMA loop.  btx_contract_fwd_lanes runs `n` samples of one
 * layer in ONE launch: lane l reads x + l*x_stride (x_stride 0: the lanes share the input, e.g. the network's first
 * layer), uses the MC sample index rng->sample_idx + l (rng->sample_idx_dev[l] when given: n consecutive words) and
 * writes out + l*out_stride (residual + l*res_stride).  Strides in bytes, multiples of 16.  Every lane computes bit for
 * bit what btx_contract_fwd_ex with BTX_FLAG_CONCURRENT would for its sample (the noise indices are relative to the lane's
 * own tensors; a launch with lanes is always planned for throughput, and the K split — the f32 summation order — of that
 * plan is decided by the grid of ONE lane, so it does not depend on n).
 * noise->sampled_w: the buffer btx_sample_weights_lanes filled for the same n.  Explicit noise tensors: n == 1 only.
 * ws: btx_contract_workspace_bytes(..., flags | BTX_FLAG_LANES(n)). */
typedef struct BtxLanes {
  int32_t n;
  int64_t x_stride, out_stride, res_stride;
} BtxLanes;
int btx_contract_fwd_lanes(int kind, const BtxGeom* g,
                           const void* x, const float* mu_w, const float* rho_w,
                           const float* mu_b, const float* rho_b,
                           void* out,
                           const BtxRng* rng, const BtxNoise* noise /* nullable */,
                           int act_dtype, int prec, uint32_t flags,
                           void* ws, size_t ws_bytes, void* stream,
                           const BtxEpilogue* epilogue /* nullable */, const BtxLanes* lanes);

/* Weight gradient of one variational contraction (training; what autograd derives for the F.conv*d / F.linear calls of
 * conv_flipout.py:376-417, conv_variational.py:379-380, linear_flipout.py:168-174):
 *   dw_mu   [n][tap][c] = sum_p dy[p][n] * x[p @ tap][c]
 *   dw_delta[n][tap][c] = sum_p (dy * s_out)[p][n] * (x * s_in)[p @ tap][c]                      (Flipout only)
 *   db_mu[n] = sum_p dy[p][n],  db_delta[n] = sum_p (dy * s_out)[p][n]                            (optional, NULL = skip)
 * in the GEMM-major layout of mu_w, f32 whatever the activation dtype (exact-f32 MFMA, f32 atomics: the summation order
 * over pixel chunks is not fixed).  The outputs are cleared by the call.  x / dy: channels-last as in btx_contract_fwd,
 * dy = gradient of its output.  The Flipout signs are those of the forward with the same BtxRng (or noise->sign_in /
 * sign_out).  dmu = dw_mu and drho = dw_delta * eps * sigmoid(rho) (Reparameterization: dw_mu * eps * sigmoid(rho)) are
 * elementwise follow-ups on the caller's side.  ConvTranspose layers: call it on the plain-convolution geometry with x and
 * dy exchanged and BTX_FLAG_SWAP_SIGNS.  The data gradient needs no entry point of its own: see BTX_FLAG_SWAP_SIGNS. */
int btx_contract_wgrad(int kind, const BtxGeom* g, const void* x, const void* dy, float* dw_mu, float* dw_delta,
                       float* db_mu, float* db_delta, const BtxRng* rng, const BtxNoise* noise /* nullable */,
                       int act_dtype, uint32_t flags, void* stream);
/* The same with a workspace (ABI 8): the partial sums of the pixel chunks leave as plain stores into per-chunk slabs in `ws`
 * and a second launch adds them in chunk order — the result no longer depends on the order in which workgroups retire, and
 * no f32 atomic crosses the fabric (db_* still uses them: N values).  With a workspace the weight gradient of a stride-1
 * 3x3 "same" convolution on bf16 activations (C % 64 == 0, N % 64 == 0, W <= 63, hashed signs, no bias: the body of a
 * ResNet) takes a kernel that holds all nine taps of a 64 x 64 tile in one workgroup (csrc/btx_wgrad_taps.h): x is staged
 * once per pixel instead of once per tap.  ws: 16-byte aligned, btx_wgrad_workspace_bytes(kind, g, act_dtype, flags) bytes
 * (BTX_E_WORKSPACE if smaller); the outputs need not be cleared and are fully overwritten.  A launch whose pixels fit one
 * chunk stores straight into dw_* (no second launch).
 * rho_w / drho (both or neither): the reduction launch also writes what btx_rho_grad would compute from the finished
 * gradient, drho[i] = dw[i] * eps(i) * sigmoid(rho_w[i]) with dw = dw_delta (Flipout) or dw_mu (Reparameterization), eps of
 * stream BTX_STREAM_EPS_W of rng; rho_w in the order of mu_w.  drho may alias dw_delta (which then holds drho only); for a
 * Reparameterization layer it must be a buffer of its own (dw_mu is dmu). */
size_t btx_wgrad_workspace_bytes(int kind, const BtxGeom* g, int act_dtype, uint32_t flags);
int btx_contract_wgrad_ws(int kind, const BtxGeom* g, const void* x, const void* dy, float* dw_mu, float* dw_delta,
                          float* db_mu, float* db_delta, const BtxRng* rng, const BtxNoise* noise /* nullable */,
                          int act_dtype, uint32_t flags, void* ws, size_t ws_bytes, const float* rho_w /* nullable */,
                          float* drho /* nullable */, void* stream);

/* Sampling pre-pass, hoisted.  The LDS-DMA / patch kernels of btx_contract_fwd* first sample the layer's weights ONCE
 * into MFMA-ready tiles (W = mu + sigma*eps for Reparameterization; mu and sigma*eps for Flipout; reference:
 * conv_flipout.py:380-394, conv_variational.py:357-366 materialise the same tensors with ATen ops) and then contract.
 * btx_sample_weights does that pre-pass for a whole model in one launch (per-layer launches cost more than the
 * sampling itself); the buffers are handed to btx_contract_fwd* through BtxNoise.sampled_w.  `out` of each item must
 * hold btx_sampled_w_bytes(geom, kind, prec) bytes (depends on N, K, groups only), 16-byte aligned; the tile layout is
 * private to the library.  items_host is a HOST array, read before the call returns. */
typedef struct BtxSampleItem {
  const BtxGeom* geom;       /* host pointer */
  const float*   mu_w;       /* device, GEMM-major, as passed to btx_contract_fwd */
  const float*   rho_w;
  void*          out;        /* device */
  int32_t        kind;
  uint32_t       layer_id;
  int32_t        src_KW, src_C;  /* both 0: mu_w/rho_w have geom's layout.  Otherwise geom describes a PADDED layout
                                    [N][KD*KH][KW][C] (row-fused stems: KW and C padded; channel-padded layers: C padded)
                                    and mu_w/rho_w are the caller's unpadded [N][KD*KH][src_KW][src_C] weights: padded
                                    positions are sampled as exact zeros, noise indices are those of the padded layout
                                    (what the contraction launched on the padded geometry expects).  groups == 1. */
} BtxSampleItem;
size_t btx_sampled_w_bytes(const BtxGeom* g, int kind, int prec);
int btx_sample_weights(const BtxSampleItem* items_host, int n_items, const BtxRng* rng /* layer_id unused */,
                       int prec, void* stream);
/* The same for `lanes` MC samples at once (sample indices rng->sample_idx + l, or rng->sample_idx_dev[l]); `out` of each
 * item holds btx_sampled_w_bytes_lanes(...) bytes.  The mean tiles of a Flipout layer do not depend on the sample: the
 * buffer keeps ONE set for all lanes, and with BTX_SAMPLE_SKIP_MU in `sflags` the call leaves them untouched — for
 * callers that know mu AND rho have not changed since the call that last wrote them into this buffer (an MC loop over
 * frozen parameters): the buffer also keeps sigma = softplus(rho) (f32, tile order) from that call, and per MC step the
 * pre-pass reads it once and writes sigma*eps for every lane, nothing else. */
#define BTX_SAMPLE_SKIP_MU 1u
size_t btx_sampled_w_bytes_lanes(const BtxGeom* g, int kind, int prec, int lanes);
int btx_sample_weights_lanes(const BtxSampleItem* items_host, int n_items, const BtxRng* rng /* layer_id unused */,
                             int prec, void* stream, int lanes, uint32_t sflags);

/* Training, the step in front of the data gradient (ABI 7).  dx of a stride-1 convolution is the forward's own contraction on the
 * spatially flipped, channel-transposed kernel, of a Linear layer on W^T (what autograd derives for F.conv*d / F.linear inside
 * conv_flipout.py:376-417, linear_flipout.py:168-174).  One launch writes its three weight operands in the GEMM-major order of
 * that geometry — out[c][tp][n] = src[n][flip ? T-1-tp : tp][c] for mu, rho and for the eps the forward drew (regenerated at the
 * source index: stream BTX_STREAM_EPS_W of rng) — from the layer's own GEMM-major parameters mu_w / rho_w [N][T][C] (groups == 1).
 * The results go to btx_contract_fwd as (mu_w, rho_w, BtxNoise.eps_w) of the transposed geometry. */
int btx_dgrad_weights(const float* mu_w, const float* rho_w, float* out_mu, float* out_rho, float* out_eps, int N, int T, int C,
                      int flip, const BtxRng* rng, void* stream);

/* Training, the step behind btx_contract_wgrad: drho[i] = dw[i] * eps(i) * sigmoid(rho[i]) over the n elements of a weight
 * tensor in the order of mu_w (what autograd derives for `sigma = log1p(exp(rho)); delta = sigma * eps` of
 * conv_flipout.py:372-375 / conv_variational.py:358-366 / linear_flipout.py:150-153).  dw = dw_delta of a Flipout layer,
 * dw_mu of a Reparameterization layer (dmu is dw_mu itself).  eps is regenerated from rng (stream BTX_STREAM_EPS_W, or
 * _EPS_B for the bias vectors): the values btx_fill_eps would write, never materialised.  drho may alias dw. */
int btx_rho_grad(const float* dw, const float* rho, float* drho, size_t n, const BtxRng* rng, uint32_t rng_stream,
                 void* stream);

/* §8(f): the data format in front of the path.  Small-C stems (BTX_FLAG_ROWFUSE) take channels-last [NB][Hp][Wp][cp]
 * activations with the conv padding materialised and the channels zero-padded to cp (4 or 8), in the MFMA dtype.
 * btx_rowfuse_pack writes that tensor in ONE pass from the caller's logical [N,C,H,W] activations of any layout:
 * strides_ncHW = element strides of (n, c, h, w), host array of 4; element (n,c,h,w) lands at
 * out[n][h+ph][w+pw][c]; everything else is zero.  (The reference hands F.conv2d the NCHW tensor and a padding
 * argument: layers/flipout_layers/conv_flipout.py:376-383.) */
int btx_rowfuse_pack(const void* x, int in_dtype, const int64_t* strides_ncHW_host, int NB, int C, int H, int W,
                     void* out, int out_dtype, int Hp, int Wp, int cp, int ph, int pw, void* stream);

/* §8(f): the op right behind the stem of the reference's ResNets (models/deterministic/resnet_large.py: maxpool after
 * conv1-bn1-relu, torch.nn.MaxPool2d(kernel, stride, padding), dilation 1, floor mode).  Channels-last [NB][H][W][C]
 * -> [NB][Ho][Wo][C], C % 8 == 0, 2*pad <= k; identical results to torch (max is exact). */
int btx_maxpool2d_cl(const void* x, void* out, int dtype, int NB, int H, int W, int C, int k, int stride, int pad,
                     void* stream);
/* The same under autograd (ABI 8; the reference's training loop runs self.maxpool with gradients enabled).  _train: also idx[NB][Ho][Wo][C]
 * (uint8, 8-byte aligned) = kh * k + kw of each output element's maximum inside its window, the first maximum in scan order as
 * torch.nn.functional.max_pool2d_with_indices picks it (k <= 15).  _bwd: dx[NB][H][W][C] from dy[NB][Ho][Wo][C] and idx — every
 * input element sums dy over the windows whose recorded maximum it is (f32 accumulation, one rounding; no atomics, every element
 * of dx is written).  Results equal torch's max_pool2d forward / backward bit for bit. */
int btx_maxpool2d_cl_train(const void* x, void* out, uint8_t* idx, int dtype, int NB, int H, int W, int C, int k, int stride,
                           int pad, void* stream);
int btx_maxpool2d_cl_bwd(const void* dy, const uint8_t* idx, void* dx, int dtype, int NB, int H, int W, int C, int k, int stride,
                         int pad, void* stream);

/* §8(f)-4, the step either side of the path in the reference's TRAINING loop (README.md:114-125 on
 * models/deterministic/resnet_large.py:46-62: conv -> bn -> relu under model.train()): torch.nn.BatchNorm2d in training mode on
 * channels-last activations x[M][C] (M = N*H*W), ABI 7.  Forward: batch mean / biased variance per channel (f64 fold of per-block
 * f32 sums, fixed order), y = (x - mean) * invstd * gamma + beta, running_mean / running_var updated as torch does
 * (running = (1 - momentum) * running + momentum * batch, unbiased variance), save_mean / save_invstd (f32 [C]) for the backward.
 * Backward: dgamma = sum(dy * xhat), dbeta = sum(dy) ([C] in param_dtype since ABI 8, f32 before; each nullable), dx = gamma * invstd * (dy - dbeta / M - xhat * dgamma / M).
 * x / y / dy / dx: act_dtype (BTX_ACT_F32 | BTX_ACT_BF16), 16-byte aligned; gamma, beta, running_*: param_dtype (same codes),
 * each nullable (affine=False / track_running_stats=False); C % 8 == 0, C <= 2048 (else BTX_E_UNSUPPORTED: the caller keeps
 * torch's own kernels).  Three launches per call on `stream`, workspace btx_bn_workspace_bytes(M, C). */
size_t btx_bn_workspace_bytes(long long M, int C);
/* ABI 8: what the reference's blocks do right behind the normalisation (models/deterministic/resnet_large.py:46-62: `relu(bn1(..))`,
 * `relu(bn2(..) + identity)`), inside the same launches.  Forward: y = [relu](bn(x) [+ residual]) in ONE rounding; with relu the
 * apply pass also writes one bit per element (y > 0; byte t = the 8 channels of 16-byte group t, M*C/8 bytes) into `mask`.
 * Backward: the incoming gradient is first masked with those bits (torch's threshold_backward), g = dy where y > 0 else 0; all
 * sums and dx are formed from g, and `dres` (nullable) receives g itself — the gradient of the residual branch. */
typedef struct BtxBnFuse {
  const void* residual;  /* forward: act_dtype, shape and layout of x, 16-byte aligned; nullable */
  int32_t     relu;
  void*       mask;      /* forward: written; backward: read.  Required when relu. */
  void*       dres;      /* backward: act_dtype [M][C], 16-byte aligned; nullable; needs relu */
} BtxBnFuse;
int btx_bn_train_fwd(const void* x, void* y, int act_dtype, long long M, int C, const void* gamma, const void* beta,
                     void* running_mean, void* running_var, int param_dtype, float momentum, float eps, float* save_mean,
                     float* save_invstd, long long* num_batches_tracked /* nullable, int64 device word: += 1 (ABI 8) */,
                     const BtxBnFuse* fuse /* nullable */, void* ws, size_t ws_bytes, void* stream);
int btx_bn_train_bwd(const void* x, const void* dy, void* dx, int act_dtype, long long M, int C, const void* gamma,
                     int param_dtype, const float* save_mean, const float* save_invstd, void* dgamma, void* dbeta,
                     const BtxBnFuse* fuse /* nullable */, void* ws, size_t ws_bytes, void* stream);

/* Global average pooling in front of the classifier (resnet_large.py: AdaptiveAvgPool2d((1,1))): channels-last
 * [NB][HW][C] -> [NB][C], f32 accumulation in a fixed order, C % 8 == 0. */
int btx_avgpool_global_cl(const void* x, void* out, int dtype, int NB, int HW, int C, void* stream);

/* Output spatial extent for a geometry (same arithmetic as torch's conv / conv_transpose). */
int btx_out_shape(const BtxGeom* g, uint32_t flags, int32_t* Do, int32_t* Ho, int32_t* Wo);

/* Materialise the noise BTX-RNG v1 defines (tests, and the reference's observable side effect that
 * layer.eps_kernel holds the eps of the last forward: conv_variational.py:362, linear_variational.py:161).
 * btx_fill_eps: out[i] = eps(stream, index i), i in [0,n)   (stream = BTX_STREAM_EPS_W / _EPS_B)
 * btx_fill_sign: out[i] = +1/-1 (int8) for element index i  (stream = BTX_STREAM_SIGN_IN / _SIGN_OUT) */
int btx_fill_eps(float* out, size_t n, const BtxRng* rng, uint32_t rng_stream, void* stream);
int btx_fill_sign(int8_t* out, size_t n, const BtxRng* rng, uint32_t rng_stream, void* stream);

/* K6. Monte-Carlo predictive accumulation (what examples/main_bayesian_imagenet_dnn2bnn.py:483-499 and
 * utils/util.py:41-60 do on the host with torch.stack -> softmax -> mean / entropy).
 * packed layout (f32): [bs*C sum p | bs*C sum p^2 | bs sum H(p) | 1 sum kl | 1 sample count]
 * The buffer is accumulated in place (zero it first); one RCCL all-reduce(sum) of it merges ranks.
 * Limit: a row of probabilities stays in LDS, C <= 24575 classes (16383 on a device that refuses the 96-KiB dynamic-LDS
 * opt-in); wider rows return BTX_E_UNSUPPORTED (the Python face then uses torch ops on the device: mc.MC_MAX_CLASSES). */
size_t btx_mc_packed_floats(int bs, int C);
int btx_mc_accumulate(const void* logits, int bs, int C, int act_dtype, float kl,
                      float* packed, void* stream);
/* the same for the logits of `lanes` MC samples back to back ([lanes][bs][C], what a btx_contract_fwd_lanes forward leaves):
 * ONE launch; a workgroup owns a batch row and folds its lanes in order, so the result is bit for bit that of `lanes`
 * btx_mc_accumulate calls. */
int btx_mc_accumulate_lanes(const void* logits, int lanes, int bs, int C, int act_dtype, float kl,
                            float* packed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BTX_H_ */
