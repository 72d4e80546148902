/* perceptor_hip.h — C ABI of libperceptor_hip.so (gfx950 / MI355X).
 *
 * perceptor itself has no FFI: its hot path is stock PyTorch ops dispatched from
 * Python.  Each entry point below replaces the PyTorch op
 * sequence of the cited reference lines; the Python classes in perceptor_amd/
 * (same names and signatures as the reference's) bind them through ctypes with
 * tensor.data_ptr() and the current HIP stream.
 *
 * Conventions: plain pointers + sizes, caller owns every buffer (including
 * workspaces), no global state, no C++ exceptions; return 0 on success,
 * negative on error (PMI_ERR_ARG = -1 bad argument, PMI_ERR_LAUNCH = -2).
 * All launches are asynchronous on `stream`.  Activations are NHWC
 * ("pixel-major": [N][H*W][C]) 16-bit (dtype 0 = f16, 1 = bf16) unless noted;
 * matrices are row-major with the contraction index contiguous.
 */
#ifndef PERCEPTOR_HIP_H
#define PERCEPTOR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* pmi_stream_t; /* hipStream_t */

int pmi_abi_version(void);

/* ---- implicit-GEMM convolution / GEMM on MFMA ----------------------------------
 * D[m][n] = act(alpha * sum_k A(m,k) * B[n][k] + bias[n] + nbias[m / hw][n]) + R[m][n]
 * conv mode (taps == 9, stride 2, or up): m = (img, y, x) over the OUTPUT grid H x W,
 * k = tap * Cin + c; A(m,k) gathers pixel (y*stride + dy, x*stride + dx) of the
 * (optionally nearest-2x upsampled) input with zero padding.  Channels [0,C0) come
 * from A0, [C0,C0+C1) from A1 (skip-concat without materialising torch.cat).
 * Replaces: nn.Conv2d 3x3 / 1x1 / Conv1d k=1 / nn.Linear / einsum-bmm in
 *   guided_diffusion/unet.py:232-252 (ResBlock), :294-300 (AttentionBlock qkv/proj),
 *   :81-138 (Upsample/Downsample), :462-467 (time_embed), :650-652 (th.cat + conv),
 *   velocity_diffusion/yfcc_2.py:17-28,52-70, cc12m_1.py:19-61, and the CLIP ViT
 *   linears (ruclip/model.py:27-58,72-131) incl. their input-gradient GEMMs.        */
typedef struct {
  const void* A0; const void* A1; const void* B;
  const float* bias;   /* [N] or NULL */
  const float* nbias;  /* [M/hw][N] per-sample bias or NULL */
  const void* R;       /* residual or NULL */
  void* D;
  const float* pro_a;  /* optional fused input prologue x <- pro_act(x * pro_a[img][c] + pro_b[img][c]) (GroupNorm-apply+FiLM+SiLU */
  const float* pro_b;  /* of the consumer's input, zero padding stays zero); only where pmi_conv3x3_halo_config() >= 0 */
  void* ws;            /* split-K workspace: splitk * M * N floats (caller-owned), NULL when splitk <= 1 */
  float* stats;        /* optional: per-channel (sum, sumsq) partials of the OUTPUT, [img][stats_p][N][2] fp32, for the next GroupNorm */
  int32_t M, N, K;     /* K % 8 == 0; N % 4 == 0 unless bias/nbias/R are NULL and ldd >= roundup(N,4) */
  int32_t C0, C1;      /* channel split of the K index (C0 + C1 = Cin) */
  int32_t lda0, lda1;  /* elements between consecutive pixels/rows of A0 / A1 */
  int32_t ldb, ldd, ldr;
  int32_t H, W, Hin, Win; /* conv mode only */
  int32_t taps;        /* 1 or 9 */
  int32_t stride;      /* 1 or 2 */
  int32_t up;          /* nearest x2 upsample of the input (Hin = H/2) */
  int32_t res_up;      /* residual is an [H/2 x W/2] grid read at (y>>1, x>>1): nearest x2 upsample of the skip path */
  int32_t act;         /* PMI_ACT_* */
  int32_t out_f32, res_f32;
  int32_t hw;          /* pixels per sample for nbias (0 if unused) */
  float alpha;
  int32_t batch, batch_inner; /* grid.z batches: z -> (z / batch_inner, z % batch_inner) */
  int64_t sA_o, sA_i, sB_o, sB_i, sD_o, sD_i, sR_o, sR_i; /* element strides */
  int32_t dtype;       /* 0 f16, 1 bf16, 2 precise (f16 MFMA on hi + lo pairs, see split_out) */
  int32_t ldnb;        /* row pitch of nbias (0 = N) */
  int32_t pro_act;     /* activation of the fused prologue */
  int32_t stats_p;     /* partial rows per image = pmi_igemm_stats_rows(); 0 = no statistics */
  int32_t splitk;      /* <= 1: none; else K is split over grid.z and reduced by a second kernel (see pmi_igemm_splitk) */
  int32_t reserved;
  const void* Bf;      /* optional: the same weights in MFMA fragment order for the weights-direct conv3x3 kernel (csrc/conv_wd.hip),
                        * [N/32][Cin/ck][3 dx][ck/16][3 dy][64 lanes][8] 16-bit with ck = 64 (tile config 4) or 32 (config 5); NULL = not packed */
  int32_t split_out;   /* "precise" mode (dtype 2): D (and a 16-bit R) hold hi + lo f16 pairs in groups of split_out (8 or 32) logical channels,
                        * ldd / ldr count 16-bit elements of the 2N-wide rows; 0 = plain.  A split INPUT needs no flag: it is a tensor with 2 Cin
                        * channels whose weights are duplicated along K by the caller.  Generic kernel only (no LDS-halo config). */
  int32_t split_in;    /* 1: the inputs are precise (hi + lo) tensors whose 2 Cin physical channels are the K dimension (C0 / C1 / K count them; weights
                        *    duplicated along K).  A fused prologue is taken by the weights-direct conv3x3 configs 6 / 7 only: they stage (hi, lo) PAIRS,
                        *    apply it to the value hi + lo and split the result again (chunk K order [yh of ck/2 channels | their yl]: Bf packed so).
                        * 2: mixed mode, single operand: the inputs are hi + lo tensors but C0 / C1 / K count LOGICAL channels; the fused prologue
                        *    (required) rounds act((hi + lo) * a + b) once to f16: plain weights, plain K loop.  Configs 6 / 7 only. */
  /* fused epilogue extras of the weights-direct GEMM (pmi_gemm_wd_eligible() == 1, 16-bit output, no split-K), the MLP of the CLIP tower
   * (ruclip/model.py:27-58) and its input gradient: */
  void* D2;            /* optional second output [M][ldd] 16-bit: the PRE-activation value (c_fc output kept for the backward pass) */
  const void* aux;     /* optional [M][ldd] 16-bit: the output is multiplied by act'(aux) with act = aux_act (dh * gelu'(h_pre) of the c_proj input gradient) */
  int32_t aux_act;
  int32_t reserved3;
} pmi_igemm_args;
int pmi_igemm(const pmi_igemm_args* a, pmi_stream_t stream);
/* >= 0 when an LDS-halo conv3x3 kernel takes this shape.  csrc/conv3x3.hip: tile config 0: 8x32 px x 256 ch, 1: 16x32 x 128, 2: 8x32 x 128 with two
 * workgroups per CU, 3: 8x32 px x <= 32 output channels; csrc/conv_wd.hip (weights-direct, needs Bf != NULL): 4: 8x32 px x 256 ch, 5: 8x32 px x 128 ch;
 * -1 when pmi_igemm uses the generic implicit-GEMM kernel (which has no fused prologue). */
int pmi_conv3x3_halo_config(const pmi_igemm_args* a);
/* 1 when the weights-direct GEMM (csrc/gemm_wd.hip) takes this call: plain GEMM (taps 1, one source, no per-sample bias / statistics / prologue),
 * K % 128 == 0, N % 256 == 0 and Bf = the weights in its fragment order [N/32][K/128][4][2][64 lanes][8] */
int pmi_gemm_wd_eligible(const pmi_igemm_args* a);
/* split-K factor recommended for this shape (1 = none); with splitk = S the caller passes ws = S*M*N floats */
int pmi_igemm_splitk(const pmi_igemm_args* a);
/* number of per-image partial rows the fused output statistics of this call would produce (0: not available for this shape) */
int pmi_igemm_stats_rows(const pmi_igemm_args* a);
/* debugging / A-B switches: key 0 = allow the LDS-halo conv3x3 kernel (default 1, returns the previous value);
 * key 1 = force halo tile config 0/1/2 where eligible (-1 = automatic);
 * key 2 = prefer the 8-wave 256-channel halo config over two 4-wave workgroups per CU where the grid allows (default 1);
 * key 6 = allow the weights-direct conv3x3 kernel (default 1); key 7 = its 16x16x32-MFMA form, tile config 6, instead of config 4 (default 1);
 * key 8 = allow its 128-channel form, tile config 7 (4 waves, two workgroups per CU), for Cout % 256 != 0 (default 1);
 * key 13 = allow its first-convolution form, tile config 8 (at most 32 input channels, the whole K in registers) (default 1);
 * key 9 = query-tile rows per wave of pmi_attn_flash (0 = automatic); key 10 = allow split-K in the weights-direct conv3x3 (default 1);
 * key 11 = largest split-K factor of the weights-direct GEMM (default 8); key 12 = workgroup count below which that GEMM uses its
 * 128-column tiles (default 128).  `python bench.py --opt "k=v,..."` sets them for a same-box A/B.
 * The library links no vendor GEMM / BLAS: every kernel it launches is in csrc/. */
int pmi_set_option(int key, int value);

/* ---- "precise" mode helpers (dtype 2: hi + lo f16 pairs, eps max-abs error < 1e-3 vs the fp32 reference path) ----------------
 * exact-fp32 batched GEMM on the f32-input MFMA: D[b][m][n] = act(alpha * sum_k A[b][m][k] * B[b][n][k] + bias[n]); transB: B is [k][n].
 * Replaces the attention einsums of unet.py:332-348 / yfcc_2.py:62-70 and the time MLPs (unet.py:462-467) in that mode. */
typedef struct {
  const float* A; const float* B; const float* bias; float* D;
  int32_t M, N, K, lda, ldb, ldd;
  int32_t transB, act;
  float alpha;
  int32_t batch, batch_inner;          /* grid.z batches: z -> (z / batch_inner, z % batch_inner) */
  int64_t sA_o, sA_i, sB_o, sB_i, sD_o, sD_i;
  const float* R;                      /* optional residual added after the activation, laid out like D */
} pmi_gemm_f32_args;
int pmi_gemm_f32(const pmi_gemm_f32_args* a, pmi_stream_t stream);
int pmi_softmax_f32(float* S, int rows, int T, int ld, float scale, pmi_stream_t s);   /* in place, fp32 (unet.py:346) */
/* fp32 [rows][C] (row pitch ld_in) <-> precise [rows][2C] */
int pmi_split_from_f32(const float* in, int ld_in, void* out, int64_t rows, int C, pmi_stream_t s);
int pmi_split_to_f32(const void* in, float* out, int64_t rows, int C, pmi_stream_t s);
/* plain f16 [rows][C] <-> precise [rows][2C] (to_split 1: low parts zero; 0: hi + lo rounded once): the level boundaries of the mixed mode,
 * where unet.py:626-654's tensors change between the two storage forms */
int pmi_split_convert(const void* in, void* out, int64_t rows, int C, int to_split, pmi_stream_t s);

/* ---- GroupNorm (+FiLM, +activation, +2x2 average pool) ---------------------------
 * unet.py:232-252 / nn.py:17-19 (GroupNorm32 -> SiLU, FiLM h*(1+scale)+shift),
 * yfcc_2.py:56 (GroupNorm(1,C)), cc12m_1.py:33-61 (GroupNorm(1,C,affine=False) + Modulation2d + ReLU).
 * stats: partial sums per (sample, pixel-chunk, channel) -> ws[N][nchunk][C][2] (fp32)
 * finalize: coefficients a[n][c], b[n][c] so that y = act(x * a + b)
 * apply: y (optionally 2x2 average-pooled after the activation)                      */
/* x1/C0: optional second source: channels [0,C0) from x, [C0,C) from x1 (normalising a skip-concat, unet.py:650-652,
 * without materialising it); pass x1 = NULL, C0 = C otherwise. */
int pmi_gn_stats(const void* x, const void* x1, int C0, float* ws, int N, int HW, int C, int G, int nchunk, int dtype, pmi_stream_t s);
/* finalize from per-CHANNEL partials s0[N][P0][C0][2] (and optionally s1[N][P1][C1][2] for the second half of a concat):
 * the layout pmi_gn_stats and the fused statistics epilogue of pmi_igemm both write. */
int pmi_gn_finalize(const float* s0, int P0, int C0, const float* s1, int P1, int C1, const float* gamma, const float* beta,
                    const float* film, int film_ld, float* coef_a, float* coef_b, int N, int HW, int G, float eps, pmi_stream_t s);
/* res: optional 16-bit NHWC tensor added after the activation (cc12m_1.py:46-61: relu(mod(norm(conv))) + skip) */
int pmi_gn_apply(const void* x, const void* x1, int C0, const float* coef_a, const float* coef_b, const void* res, void* y, int N, int H, int W, int C,
                 int act, int pool, int dtype, pmi_stream_t s);
/* the down ResBlock's two pooled tensors from ONE pass over x (unet.py:232-243: h = AvgPool(SiLU(GN(x))), skip = AvgPool(x)) */
int pmi_gn_apply_pool_skip(const void* x, const float* coef_a, const float* coef_b, void* y, void* y_raw, int N, int H, int W, int C, int act,
                           int dtype, pmi_stream_t s);

/* ---- attention, head dim 64 (flash-style, MFMA) -----------------------------------
 * unet.py:332-348 (QKVAttentionLegacy), :364-382 (QKVAttention), yfcc_2.py:62-70.
 * qkv_split re-tiles the qkv 1x1-conv output [N][T][3C] into Q,K [N*heads][Tp][64]
 * and V^T [N*heads][64][Tp] (Tp = T rounded up to 32, zero filled).
 * order: 0 = heads then q,k,v (legacy), 1 = q,k,v then heads (new order and v-diffusion). */
/* Flash-style attention for head dims 8..160 (multiples of 8) with a separate key / value sequence: the StableDiffusion transformer
 * blocks' self- and cross-attention (stable_diffusion/attention.py:268-298, replaces the xformers call at :285).
 * q [N][T][ldq], k and v [N][Tk][ldkv] 16-bit with head h at channel offset h*d of each pointer; out [N][T][heads*d];
 * ws: caller-owned scratch of pmi_attn_flash_workspace(...) KiB (operands re-tiled into MFMA fragment order; < 0: unsupported). */
int pmi_attn_flash_workspace(int N, int T, int Tk, int heads, int d);
int pmi_attn_flash(const void* q, int ldq, const void* k, const void* v, int ldkv, void* out, void* ws, int N, int T, int Tk, int heads,
                   int d, float scale, int dtype, pmi_stream_t s);
int pmi_qkv_split(const void* qkv, void* q, void* k, void* vt, int N, int T, int heads, int order, int dtype, pmi_stream_t s);
int pmi_attn_d64(const void* q, const void* k, const void* vt, void* out, int N, int T, int heads, float scale,
                 int dtype, pmi_stream_t s);

/* ViT attention with input-gradient (head dim 64, any T; nn.MultiheadAttention in ruclip/model.py:40-52), flash style:
 * fwd: qkv [N][T][3C] with channels (q|k|v, head, d) -> out [N][T][C]; saves ws16 = 6 x [N*heads][Tp][64] 16-bit
 *      (Q,K,V and their transposes, Tp = T rounded up to 32) and lse [N*heads][Tp] fp32 for the backward.
 * bwd: dout [N][T][C] -> dqkv [N][T][3C]; needs o (= fwd out), ws16b = 2 x [N*heads][Tp][64] 16-bit and delta [N*heads][Tp] fp32 scratch. */
int pmi_vit_attn_fwd(const void* qkv, void* ws16, float* lse, void* out, int N, int T, int heads, float scale, int dtype, pmi_stream_t s);
int pmi_vit_attn_bwd(const void* ws16, const float* lse, const void* o, const void* dout, void* ws16b, float* delta, void* dqkv,
                     int N, int T, int heads, float scale, int dtype, pmi_stream_t s);

/* ---- layout / elementwise on the UNet path -----------------------------------------
 * prep: images NCHW fp32 in [0,1] -> x = 2*img-1 (diffusion_space.py:1-2) as NHWC 16-bit with
 *       Cpad channels; channels [3, 3+nplanes) are per-sample constants planes[n][j]
 *       (yfcc_2.py:73-74,247-249 expand_to_planes of the Fourier timestep features).
 * finish: NHWC (fp32, ld channels) -> NCHW fp32 first `cout` channels (guided_diffusion.py:125-133 [:, :3].float()) */
int pmi_prep_input(const float* img, const float* planes, int nplanes, void* x, int N, int H, int W, int Cpad, int dtype, pmi_stream_t s);
int pmi_finish_output(const float* y, int ld, float* out, int N, int H, int W, int cout, pmi_stream_t s);
/* StableDiffusion latents / VAE (models/stable_diffusion/stable_diffusion.py:194-198,259-271): generic NCHW fp32 <-> NHWC layout
 * conversion with an affine map (x*mul + add: the 1/0.18215 latent scale, diffusion_space.decode's (x+1)/2), and the GEGLU
 * gate of the transformer blocks' feed-forward (stable_diffusion/attention.py:346-348): h[M][2F] = (value | gate) -> value*gelu(gate);
 * interleaved = 1: h holds 16 value then 16 gate columns per 32 (the column order of weights packed for pmi_igemm's act = 5 = GEGLU
 * epilogue, which writes value*gelu(gate) directly -- N / 2 output columns, weights-direct GEMM only -- and makes this pass unnecessary). */
int pmi_nchw_to_nhwc(const float* in, void* x, int N, int C, int H, int W, int Cpad, float mul, float add, int dtype, pmi_stream_t s);
int pmi_nhwc_to_nchw(const float* y, int ld, float* out, int N, int H, int W, int cout, float mul, float add, pmi_stream_t s);
int pmi_geglu(const void* h, void* out, int64_t M, int F, int interleaved, int dtype, pmi_stream_t s);
int pmi_avgpool2(const void* x, void* y, int N, int H, int W, int C, int dtype, pmi_stream_t s);            /* nn.AvgPool2d(2) */
int pmi_upsample_bilinear2(const void* x, void* y, int N, int H, int W, int C, int dtype, pmi_stream_t s);  /* align_corners=False */
int pmi_upsample_nearest2(const void* x, void* y, int N, int H, int W, int C, pmi_stream_t s);              /* nn.Upsample(2, 'nearest'), 16-bit NHWC */
/* nn.py:101-118 sinusoidal embedding [cos|sin] -> 16-bit [N][dim] */
int pmi_timestep_embedding(const float* t, void* out, int N, int dim, float max_period, int dtype, pmi_stream_t s);
/* yfcc_2.py:41-49 Fourier features: out[n] = [cos(2 pi t w_j) | sin(2 pi t w_j)] fp32 */
int pmi_fourier_features(const float* t, const float* w, float* out, int N, int half, pmi_stream_t s);
int pmi_cast_f32_to_16(const float* in, void* out, int64_t n, int act, int dtype, pmi_stream_t s);

/* ---- sampler updates (fp32, NCHW, per-sample scalars) -------------------------------
 * guided_diffusion/predictions.py:51-59,61-98 ; velocity_diffusion/predictions.py:50-62,68-105 ;
 * guided: predictions.py:147-154 (both forms): p += scale * sigma_from * clamp(g,+-c)/c          */
int pmi_ddim_eps_step(const float* img, const float* eps, const float* a_from, const float* s_from,
                      const float* a_to, const float* s_to, float* next_img, float* denoised_img,
                      int N, int64_t chw, pmi_stream_t s);
int pmi_ddim_v_step(const float* img, const float* v, const float* a_from, const float* s_from,
                    const float* a_to, const float* s_to, float* next_img, float* denoised_img,
                    int N, int64_t chw, pmi_stream_t s);
int pmi_guided_update(const float* pred, const float* grad, const float* s_from, float scale, float clamp_value,
                      float* out, int N, int64_t chw, pmi_stream_t s);

/* remaining Predictions algebra (predictions.py:101-145,174-179 ; velocity_diffusion/predictions.py:107-200):
 * out = ca[n]*a + cb[n]*b + cc[n] with per-sample coefficients (b, cb, cc optional); clamp with per-sample bounds */
int pmi_lincomb2(const float* a, const float* b, const float* ca, const float* cb, const float* cc, float* out, int N,
                 int64_t chw, pmi_stream_t s);
int pmi_clamp(const float* a, const float* lo, const float* hi, float* out, int N, int64_t chw, pmi_stream_t s);

/* ---- Predictions variants and clamp_with_grad (csrc/sampling.hip), fp32, one row per sample ------------------
 * pmi_quantile_abs: out[n] = torch.quantile(|x[n, :]|, q) (linear interpolation), radix select -- replaces the quantile in
 *   Predictions.dynamic_threshold (guided_diffusion/predictions.py:156-172 ; velocity_diffusion/predictions.py:148-164).
 * pmi_randn: standard-normal noise for step(eta>0) / resample_noise / noisy_reverse_step (predictions.py:61-98,126-145), replacing
 *   torch.randn_like.  Philox4x32-10 + Box-Muller; element e of the draw (seed, stream) is a function of (seed, stream,
 *   first_element + e) only, so a rank that holds samples [r0, r1) of a batch passes first_element = r0 * chw and gets the same values
 *   as a single process.  pmi_philox4x32_10 exposes the raw generator for known-answer tests.
 * pmi_sort_rows / pmi_wasserstein: Predictions.wasserstein_distance / wasserstein_square_distance (predictions.py:184-198):
 *   rows sorted ascending into work[rows][pmi_sort_rows_padded(n)] (bitonic network, +inf padding), then
 *   out[0] = mean |sorted - Normal(0,1).icdf(linspace(0.5/n, 1-0.5/n, n))|^power (power 1 or 2); partial = 1024 floats of workspace.
 * pmi_clamp_grad: backward of clamp_with_grad (transforms/clamp_with_grad.py:8-23) with per-sample bounds:
 *   out = grad * (grad * (x - clamp(x, lo, hi)) >= 0).                                                                          */
int pmi_quantile_abs(const float* x, float* out, int N, int64_t n, float q, pmi_stream_t s);
int pmi_randn(float* out, int64_t n, int64_t first_element, int64_t seed_bits, int64_t stream_bits, pmi_stream_t s);
int pmi_philox4x32_10(uint32_t* out4, int64_t counter_lo, int64_t counter_hi, int64_t key, pmi_stream_t s);
int pmi_sort_rows_padded(int64_t n);
int pmi_sort_rows(const float* x, float* work, int rows, int64_t n, pmi_stream_t s);
int pmi_wasserstein(const float* sorted, int rows, int64_t n, int power, float* partial, float* out, pmi_stream_t s);
int pmi_clamp_grad(const float* x, const float* grad, const float* lo, const float* hi, float* out, int N, int64_t chw, pmi_stream_t s);

/* ---- input-gradient of the v-diffusion UNets (csrc/backward.hip) --------------------------------------------
 * What autograd computes upstream when losses/velocity_diffusion.py:33-61 (guided_resample_) backpropagates a loss on the denoised
 * image to the noise.  dX of every convolution is pmi_igemm on flipped / transposed packed weights and the attention backward is
 * pmi_vit_attn_bwd; these are the memory-bound adjoints between them (16-bit NHWC, dtype 0 / 1):
 * pmi_add16: out = a + b (ResConvBlock main + skip, yfcc_2.py:17-28, when the ReLU output must survive for its mask);
 * pmi_avgpool2_bwd: dy [N][H/2][W/2][C] -> dx [N][H][W][C] (nn.AvgPool2d(2)); pmi_upsample_bilinear2_bwd: dy [N][2H][2W][C] -> dx
 * [N][H][W][C] (exact adjoint of pmi_upsample_bilinear2); pmi_gn1_bwd: GroupNorm(1, C) with affine, dx = r (g - mean g - xhat mean(g xhat))
 * (+ res), g = (gamma[n * gamma_ld + c] + gamma_add) dy: gamma_ld = 0 for a shared affine weight (SelfAttention2d.norm, yfcc_2.py:41-52),
 * > 0 with gamma_add = 1 for Modulation2d's per-sample scale after GroupNorm(1, C, affine=False) (cc12m_1.py:33-61); two launches
 * (slice sums into `partial`, then every workgroup adds the slices in index order and streams its part): deterministic.                       */
int pmi_add16(const void* a, const void* b, void* out, int64_t n, int dtype, pmi_stream_t s);
int pmi_avgpool2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, pmi_stream_t s);
int pmi_upsample_bilinear2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, pmi_stream_t s);
int pmi_upsample_nearest2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, pmi_stream_t s);   /* wikiart_256.py:117 */
/* GroupNorm32 (+FiLM) + activation backward of the ADM UNet (unet.py:232-252 out_layers / in_layers, nn.py:17-19; the gradient autograd forms
 * when GuidedDiffusion.predicted_noise is differentiated, guided_diffusion.py:125-133).  The forward is y = act(a x + b) with pmi_gn_finalize's
 * coefficients; stats: per-channel partials of dt = dy act'(a x + b) and dt x, ws [N][nchunk][C][2]; finalize: from the FORWARD partials s0 / s1
 * (as given to pmi_gn_finalize) and those, P / Q per channel; apply: dx = a dt + P x + Q (+ gadd), one output per concat source. */
int pmi_gn_bwd_stats(const void* x, const void* x1, int C0, const void* dy, const float* coef_a, const float* coef_b, int act, float* ws,
                     int N, int HW, int C, int nchunk, int dtype, pmi_stream_t s);
int pmi_gn_bwd_finalize(const float* s0, int P0, int C0, const float* s1, int P1, int C1, const float* ws_bwd, int PB, const float* gamma,
                        const float* film, int film_ld, float* coef_p, float* coef_q, int N, int HW, int G, float eps, pmi_stream_t s);
int pmi_gn_bwd_apply(const void* x, const void* x1, int C0, const void* dy, const float* coef_a, const float* coef_b, const float* coef_p,
                     const float* coef_q, int act, const void* gadd0, const void* gadd1, void* dx0, void* dx1, int N, int HW, int C,
                     int dtype, pmi_stream_t s);
int pmi_gn1_bwd_partials(int64_t hw, int C);   /* slices per sample: the caller passes partial = N * this * 4 doubles of workspace */
int pmi_gn1_bwd(const void* x, const void* dy, const float* gamma, int gamma_ld, float gamma_add, const void* res, void* dx,
                double* partial, int N, int64_t hw, int C, float eps, int dtype, pmi_stream_t s);

/* ---- CLIP guidance path (forward + input-gradient) ---------------------------------------
 * ViT arithmetic: open-clip-torch 2.0.2 visual tower == OpenAI-CLIP VisionTransformer, in-tree copy
 * ruclip/model.py:11-131; wrapper models/open_clip.py:109-123; loss losses/clip/clip.py:89-99.
 * The linear layers (and their dX = dY * W input gradients) run on pmi_igemm; these are the
 * memory-bound pieces between them.                                                           */
/* LayerNorm (ruclip/model.py:11-17): fp32 rows -> 16-bit and/or fp32; mean_rstd = [mean[M] | rstd[M]] saved for bwd */
int pmi_layernorm_fwd(const float* x, int ld_x, const float* gamma, const float* beta, void* y16, float* y32, float* mean_rstd,
                      int M, int D, float eps, int dtype, pmi_stream_t s);
/* input gradient; dy row r (stride dy_ld) belongs to row r*row_stride of x/gres/outputs; out = dx + gres */
int pmi_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean_rstd, const float* gres,
                      float* g32, void* g16, int M, int D, int dy_ld, int row_stride, int dtype, pmi_stream_t s);
/* LayerNorm fused with the split-K reduction of the GEMM in front of it: the caller runs pmi_igemm with splitk > 1 and reserved3 = 1 (the
 * slabs ws[splitk][M][D] are left unreduced) and passes them here.  forward: x = sum_s ws[s] + bias + residual -> x_out (fp32) and
 * LayerNorm(x) as 16-bit operand (+ mean_rstd[2][M]); backward: dy = sum_s ws[s], then as pmi_layernorm_bwd.  D % 256 == 0, D <= 2048. */
int pmi_layernorm_fwd_slabs(const float* ws, int nslab, int64_t slab_stride, const float* bias, const float* residual, float* x_out,
                            const float* gamma, const float* beta, void* y16, float* mean_rstd, int M, int D, float eps, int dtype, pmi_stream_t s);
int pmi_layernorm_bwd_slabs(const float* ws, int nslab, int64_t slab_stride, const float* x, const float* gamma, const float* mean_rstd,
                            const float* gres, float* g32, void* g16, int M, int D, int dtype, pmi_stream_t s);
/* softmax over the first T columns of fp32 scores * scale -> 16-bit probabilities (zero padded to ld_out) */
int pmi_softmax_fwd(const float* S, void* P, int rows, int T, int ld_in, int ld_out, float scale, int dtype, pmi_stream_t s);
/* CLIP text tower (open_clip text transformer reached through models/open_clip.py:99-107; transformers CLIPTextModel through
 * models/stable_diffusion/stable_diffusion.py:295-323): causal variant of pmi_softmax_fwd -- rows = batch*heads*T, query row r
 * sees keys 0..(r mod T) -- plus x[n][t][:] = tok[ids[n][t]][:] + pos[t][:] (pos may be NULL) and a row gather (the EOT token's row). */
int pmi_softmax_causal_fwd(const float* S, void* P, int rows, int T, int ld_in, int ld_out, float scale, int dtype, pmi_stream_t s);
int pmi_embed_tokens(const int64_t* ids, const float* tok, const float* pos, float* x, int N, int T, int D, int vocab, pmi_stream_t s);
int pmi_gather_rows(const float* src, const int64_t* idx, float* dst, int R, int D, int ld, int64_t src_rows, pmi_stream_t s);
int pmi_softmax_bwd(const float* dP, const void* P, void* dS, int rows, int T, int ld_dp, int ld_p, float scale, int dtype, pmi_stream_t s);
/* in[b][R][Cc] (row stride ld_in, batch offset (b/batch_inner)*sI_o + (b%batch_inner)*sI_i) -> out[b][Cc][Rp], Rp = R rounded up to 8 */
int pmi_transpose_16(const void* in, void* out, int R, int Cc, int ld_in, int64_t sI_o, int64_t sI_i, int batch_inner, int batch, pmi_stream_t s);
int pmi_act_bwd(const void* dh, const void* hpre, void* out, int64_t n, int act, int dtype, pmi_stream_t s); /* out = dh * act'(hpre) */
/* transforms/resize/resize_right.py:34-189 as a banded operator along the middle axis of [outer][in_sz][inner]:
 * out[o][j][i] = sum_t w[j][t] * in[o][idx[j][t]][i] (idx < 0 skipped).  The adjoint (image gradient) is the same call
 * with the transposed band.  r0/r1 reserved (0). */
int pmi_resize_apply(const float* in, float* out, const int* idx, const float* w, int outer, int in_sz, int inner, int out_sz,
                     int taps, int r0, int r1, pmi_stream_t s);
/* Normalize (models/open_clip.py:78-81) + patch conv as im2col (ruclip/model.py:84-90,105): col[N*g*g][Kp] 16-bit */
int pmi_patchify(const float* img, const float* mean, const float* stdv, void* col, int N, int R, int P, int Kp, int r0, int dtype, pmi_stream_t s);
int pmi_unpatchify(const float* dcol, const float* stdv, float* dimg, int N, int R, int P, int Kp, float mul, pmi_stream_t s);
int pmi_act_fwd(const void* in, void* out, int64_t n, int act, int dtype, pmi_stream_t s);   /* QuickGELU / GELU (ruclip/model.py:20-23) */
int pmi_l2norm_rows(const float* x, float* y, int M, int D, float scale, pmi_stream_t s);    /* scale * F.normalize(x): models/open_clip.py:120-121, cc12m_1.py:294 */
int pmi_vit_assemble(const float* emb, const float* cls, const float* pos, float* x, int N, int T, int D, int r0, pmi_stream_t s);
/* losses/clip/clip.py:89-99 (+ F.normalize of models/open_clip.py:120-121): loss = mult * sum_{n,k} w_k 2 asin(|e_n-t_k|/2)^2 / (n_total*K);
 * demb = gscale * dloss/demb (emb un-normalised).  n_total = global batch (>= N) so a sharded batch keeps the global mean. */
int pmi_spherical_loss(const float* emb, const float* tgt, const float* wts, float* loss, float* demb, int N, int K, int D,
                       int n_total, float mult, float gscale, pmi_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
