// Input-gradient pieces of the v-diffusion UNets (the dX the reference gets from autograd when
// losses/velocity_diffusion.py:33-61 `guided_resample_` backpropagates a loss on the denoised image to the noise).  The matrix work
// (dX of every convolution = the forward kernels on flipped / transposed packed weights, attention backward = pmi_vit_attn_bwd) is not
// here; these are the memory-bound adjoints between them, 16-bit NHWC like the forward tensors, fp32 arithmetic:
//   pmi_add16                   out = a + b                       (ResConvBlock's main + skip when both must be kept: yfcc_2.py:17-28)
//   pmi_avgpool2_bwd            adjoint of nn.AvgPool2d(2)         (yfcc_2.py:101 ff.)
//   pmi_upsample_bilinear2_bwd  adjoint of F.interpolate(x2, bilinear, align_corners=False)   (yfcc_2.py:113 ff.)
//   pmi_upsample_nearest2_bwd   adjoint of nn.Upsample(2, 'nearest')                          (wikiart_256.py:117)
//   pmi_gn1_bwd                 backward of GroupNorm(1, C) with a shared affine weight (SelfAttention2d.norm, yfcc_2.py:41-52) or a per-sample
//                               FiLM scale (Modulation2d after GroupNorm(1, C, affine=False), cc12m_1.py:33-61), plus an optional residual path
//   pmi_gn_bwd_stats / _finalize / _apply   (round 3) backward of GroupNorm32 (+ FiLM) + activation of the ADM UNet (unet.py:232-252, nn.py:17-19),
//                               G groups, one or two (skip-concat) sources: the forward is y = act(a[n][c] x + b[n][c]) with the coefficients of
//                               pmi_gn_finalize, so with dt = dy act'(a x + b):  dx = a dt + P[n][g] x + Q[n][g],
//                               P = -r^2 m2, Q = -r m1 + r^2 mu m2, m1 = mean_g(gamma' dt), m2 = mean_g(gamma' dt xhat) -- three streaming passes
//                               shaped like the forward's stats / finalize / apply
#include "../../include/perceptor_hip.h"
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void add16_kernel(const u16* __restrict__ a, const u16* __restrict__ b, u16* __restrict__ out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float x[8], y[8];
    unpack8<T>(*(const uint4*)(a + i * 8), x);
    unpack8<T>(*(const uint4*)(b + i * 8), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += y[e];
    *(uint4*)(out + i * 8) = pack8<T>(x);
  }
}

// dx[n][y][x][c] = 0.25 * dy[n][y/2][x/2][c]
template <typename T>
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const u16* __restrict__ dy, u16* __restrict__ dx, int N, int H, int W, int C) {
  const int C8 = C >> 3, Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / ((int64_t)H * W));
    const int rem = (int)(pix - (int64_t)n * H * W);
    const int y = rem / W, x = rem - y * W;
    float f[8];
    unpack8<T>(*(const uint4*)(dy + (((int64_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c8 * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= 0.25f;
    *(uint4*)(dx + pix * C + c8 * 8) = pack8<T>(f);
  }
}

// Adjoint of nn.Upsample(2, 'nearest'): dx[y][x] = sum of the 2x2 block of dy
template <typename T>
__global__ __launch_bounds__(256) void upsample_nearest2_bwd_kernel(const u16* __restrict__ dy, u16* __restrict__ dx, int N, int H, int W, int C) {
  const int C8 = C >> 3, Wo = 2 * W;
  const int64_t total = (int64_t)N * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / ((int64_t)H * W));
    const int rem = (int)(pix - (int64_t)n * H * W);
    const int y = rem / W, x = rem - y * W;
    const u16* b = dy + (((int64_t)n * 2 * H + 2 * y) * Wo + 2 * x) * C + c8 * 8;
    float a0[8], a1[8], a2[8], a3[8];
    unpack8<T>(*(const uint4*)b, a0);
    unpack8<T>(*(const uint4*)(b + C), a1);
    unpack8<T>(*(const uint4*)(b + (int64_t)Wo * C), a2);
    unpack8<T>(*(const uint4*)(b + (int64_t)Wo * C + C), a3);
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = (a0[e] + a1[e]) + (a2[e] + a3[e]);
    *(uint4*)(dx + pix * C + c8 * 8) = pack8<T>(a0);
  }
}

// Adjoint of upsample_bilinear2_kernel (elementwise.hip): input index i receives from outputs 2i-1 .. 2i+2 with weights
// .25 [i >= 1], .75 + .25 [i == 0], .75 + .25 [i == n-1], .25 [i <= n-2] (the clamped border taps fold onto the border pixel).
__device__ __forceinline__ void bilinear_adj_weights(int i, int n, float w[4]) {
  w[0] = i >= 1 ? 0.25f : 0.f;
  w[1] = 0.75f + (i == 0 ? 0.25f : 0.f);
  w[2] = 0.75f + (i == n - 1 ? 0.25f : 0.f);
  w[3] = i <= n - 2 ? 0.25f : 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_bilinear2_bwd_kernel(const u16* __restrict__ dy, u16* __restrict__ dx, int N, int H, int W, int C) {
  const int C8 = C >> 3, Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)N * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / ((int64_t)H * W));
    const int rem = (int)(pix - (int64_t)n * H * W);
    const int y = rem / W, x = rem - y * W;
    float wy[4], wx[4], acc[8];
    bilinear_adj_weights(y, H, wy);
    bilinear_adj_weights(x, W, wx);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int oy = 2 * y - 1 + a;
      if (wy[a] == 0.f) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ox = 2 * x - 1 + b;
        if (wx[b] == 0.f) continue;
        float f[8];
        unpack8<T>(*(const uint4*)(dy + (((int64_t)n * Ho + oy) * Wo + ox) * C + c8 * 8), f);
        const float w = wy[a] * wx[b];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += w * f[e];
      }
    }
    *(uint4*)(dx + pix * C + c8 * 8) = pack8<T>(acc);
  }
}

// GroupNorm(1, C) backward:  y = (x - mu) * r * gamma[c] + beta[c]
//   g = gamma * dy;  dx = r * (g - mean(g) - xhat * mean(g * xhat)) (+ res), means over all C*HW elements of the sample.
// Two launches: (1) P workgroups per sample reduce (sum x, sum x^2, sum g, sum g x) over their slice in a fixed order (per-thread
// serial, wave butterfly, per-wave LDS slots added in order) into partial[n][p][4] doubles; (2) every workgroup of the apply pass adds
// the P partials of its sample in index order (same result in every workgroup: deterministic) and streams its slice.
constexpr int GT = 256;

template <typename T>
__global__ __launch_bounds__(GT) void gn1_bwd_reduce_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, const float* __restrict__ gamma_,
                                                            int gamma_ld, float gamma_add, double* __restrict__ partial, int64_t hw, int C, int P) {
  __shared__ double red[GT / 64][4];
  const int n = blockIdx.y, p = blockIdx.x;
  const float* const gamma = gamma_ + (int64_t)n * gamma_ld;     // gamma_ld = 0: shared affine weight; > 0: per-sample (FiLM scale)
  const int C8 = C >> 3;
  const int64_t n8 = hw * C8, base = (int64_t)n * hw * C;
  const int64_t per = (n8 + P - 1) / P, i0 = p * per, i1 = i0 + per < n8 ? i0 + per : n8;
  double s[4] = {0, 0, 0, 0};
  for (int64_t i = i0 + threadIdx.x; i < i1; i += GT) {
    const int c0 = (int)(i % C8) * 8;
    float xv[8], dv[8];
    unpack8<T>(*(const uint4*)(x + base + i * 8), xv);
    unpack8<T>(*(const uint4*)(dy + base + i * 8), dv);
    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = (gamma[c0 + e] + gamma_add) * dv[e];
      q[0] += xv[e]; q[1] += xv[e] * xv[e]; q[2] += g; q[3] += g * xv[e];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += (double)q[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0;
    for (int w = 0; w < GT / 64; ++w) t += red[w][threadIdx.x];
    partial[((int64_t)n * P + p) * 4 + threadIdx.x] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(GT) void gn1_bwd_apply_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, const float* __restrict__ gamma_,
                                                           int gamma_ld, float gamma_add, const u16* __restrict__ res, u16* __restrict__ dx,
                                                           const double* __restrict__ partial, int64_t hw, int C, int P, float eps) {
  __shared__ float coef[4];
  const int n = blockIdx.y, p = blockIdx.x;
  const float* const gamma = gamma_ + (int64_t)n * gamma_ld;
  if (threadIdx.x == 0) {
    double t[4] = {0, 0, 0, 0};
    for (int q = 0; q < P; ++q)
      for (int k = 0; k < 4; ++k) t[k] += partial[((int64_t)n * P + q) * 4 + k];
    const double cnt = (double)hw * C;
    const double mu = t[0] / cnt, var = t[1] / cnt - mu * mu;
    const double r = 1.0 / sqrt((var > 0 ? var : 0) + (double)eps);
    coef[0] = (float)mu; coef[1] = (float)r;
    coef[2] = (float)(t[2] / cnt);                             // mean(g)
    coef[3] = (float)(r * (t[3] - mu * t[2]) / cnt);           // mean(g * xhat)
  }
  __syncthreads();
  const float mu = coef[0], r = coef[1], m1 = coef[2], m2 = coef[3];
  const int C8 = C >> 3;
  const int64_t n8 = hw * C8, base = (int64_t)n * hw * C;
  const int64_t per = (n8 + P - 1) / P, i0 = p * per, i1 = i0 + per < n8 ? i0 + per : n8;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += GT) {
    const int c0 = (int)(i % C8) * 8;
    float xv[8], dv[8], o[8];
    unpack8<T>(*(const uint4*)(x + base + i * 8), xv);
    unpack8<T>(*(const uint4*)(dy + base + i * 8), dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = r * ((gamma[c0 + e] + gamma_add) * dv[e] - m1 - (xv[e] - mu) * r * m2);
    if (res) {
      float rv[8];
      unpack8<T>(*(const uint4*)(res + base + i * 8), rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += rv[e];
    }
    *(uint4*)(dx + base + i * 8) = pack8<T>(o);
  }
}

inline unsigned grid_for(int64_t items) {
  const int64_t b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}


// ---- GroupNorm(G) (+FiLM) + activation backward (ADM UNet) --------------------------------------------------------------------------
// stats: per (sample, pixel chunk, channel) partials A = sum_p dt, B = sum_p dt * x with dt = dy * act'(a x + b); x = concat of two sources.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const u16* __restrict__ x, const u16* __restrict__ x1, int C0, const u16* __restrict__ dy,
                                                           const float* __restrict__ ca, const float* __restrict__ cb, int act,
                                                           float* __restrict__ ws, int HW, int C, int nchunk) {
  __shared__ float s_a[4096], s_b[4096];          // [PPI][C] slots, one per pixel lane (PPI * C <= 4096): no float atomics, fixed order
  const int tid = threadIdx.x, chunk = blockIdx.x, n = blockIdx.y;
  const int C8 = C >> 3;
  const int TPP = C8 < 256 ? C8 : 256;
  const int PPI = 256 / TPP;
  const int my_p = tid / TPP, my_c = tid - my_p * TPP;
  const int ppc = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  const int C1 = C - C0;
  if (my_p < PPI) {
    for (int c8 = my_c; c8 < C8; c8 += TPP) {
      const bool second = c8 * 8 >= C0;
      const int Cs = second ? C1 : C0, cl = second ? c8 * 8 - C0 : c8 * 8;
      const u16* xb = (second ? x1 : x) + (int64_t)n * HW * Cs;
      const u16* db = dy + (int64_t)n * HW * C;
      float a[8], b[8], sa[8], sb[8];
      *(float4*)a = *(const float4*)(ca + (int64_t)n * C + c8 * 8); *(float4*)(a + 4) = *(const float4*)(ca + (int64_t)n * C + c8 * 8 + 4);
      *(float4*)b = *(const float4*)(cb + (int64_t)n * C + c8 * 8); *(float4*)(b + 4) = *(const float4*)(cb + (int64_t)n * C + c8 * 8 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sb[e] = 0.f; }
      for (int p = p0 + my_p; p < p1; p += PPI) {
        float f[8], d[8];
        unpack8<T>(*(const uint4*)(xb + (int64_t)p * Cs + cl), f);
        unpack8<T>(*(const uint4*)(db + (int64_t)p * C + c8 * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dt = d[e] * act_grad(a[e] * f[e] + b[e], act);
          sa[e] += dt; sb[e] += dt * f[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { s_a[my_p * C + c8 * 8 + e] = sa[e]; s_b[my_p * C + c8 * 8 + e] = sb[e]; }
    }
  }
  __syncthreads();
  float* o = ws + ((int64_t)n * nchunk + chunk) * C * 2;
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int pl = 0; pl < PPI; ++pl) { a += s_a[pl * C + c]; b += s_b[pl * C + c]; }
    o[2 * c] = a; o[2 * c + 1] = b;
  }
}

// finalize: one workgroup per (sample, group).  Forward moments (mu, r) from the forward's per-channel (sum, sumsq) partials -- the same
// double-precision combination as gn_finalize_kernel -- then m1, m2 from the backward partials; writes P, Q per channel of the group.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ s0, int P0, int C0, const float* __restrict__ s1, int P1, int C1,
                                                              const float* __restrict__ wsb, int PB, const float* __restrict__ gamma,
                                                              const float* __restrict__ film, int film_ld, float* __restrict__ cp,
                                                              float* __restrict__ cq, int HW, int G, float eps) {
  __shared__ double red[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, n = blockIdx.x, g = blockIdx.y;
  const int C = C0 + C1, cpg = C / G, g_lo = g * cpg;
  double s = 0.0, q = 0.0, a1 = 0.0, a2 = 0.0;          // forward sum, sumsq; backward sum gamma' A, sum gamma' B
  for (int j = 0; j < cpg; ++j) {
    const int c = g_lo + j;
    const bool second = c >= C0;
    const float* base = second ? s1 + (int64_t)n * P1 * C1 * 2 : s0 + (int64_t)n * P0 * C0 * 2;
    const int P = second ? P1 : P0, Cs = second ? C1 : C0, cl = second ? c - C0 : c;
    float gm = gamma ? gamma[c] : 1.f;
    if (film) gm *= 1.f + film[(int64_t)n * film_ld + c];
    double fs = 0.0, fq = 0.0, ba = 0.0, bb = 0.0;
    for (int pr = tid; pr < P; pr += 256) { const float2 v = *(const float2*)(base + ((int64_t)pr * Cs + cl) * 2); fs += v.x; fq += v.y; }
    for (int pr = tid; pr < PB; pr += 256) { const float2 v = *(const float2*)(wsb + (((int64_t)n * PB + pr) * C + c) * 2); ba += v.x; bb += v.y; }
    s += fs; q += fq; a1 += (double)gm * ba; a2 += (double)gm * bb;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
  if (lane == 0) { red[0][wid] = s; red[1][wid] = q; red[2][wid] = a1; red[3][wid] = a2; }
  __syncthreads();
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  a1 = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  a2 = red[3][0] + red[3][1] + red[3][2] + red[3][3];
  const double cnt = (double)HW * cpg;
  const double mu = s / cnt;
  double var = q / cnt - mu * mu;
  if (var < 0.0) var = 0.0;
  const double r = 1.0 / sqrt(var + (double)eps);
  const double m1 = a1 / cnt;                            // mean_g(gamma' dt)
  const double m2 = r * (a2 - mu * a1) / cnt;            // mean_g(gamma' dt xhat), xhat = (x - mu) r
  const float Pv = (float)(-r * r * m2), Qv = (float)(-r * m1 + r * r * mu * m2);
  for (int j = tid; j < cpg; j += 256) { cp[(int64_t)n * C + g_lo + j] = Pv; cq[(int64_t)n * C + g_lo + j] = Qv; }
}

// apply: dx = a dt + P x + Q (+ gadd: the gradient arriving over the block's skip path), written per source
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const u16* __restrict__ x, const u16* __restrict__ x1, int C0, const u16* __restrict__ dy,
                                                           const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ cp,
                                                           const float* __restrict__ cq, int act, const u16* __restrict__ gadd0,
                                                           const u16* __restrict__ gadd1, u16* __restrict__ dx0, u16* __restrict__ dx1,
                                                           int N, int HW, int C) {
  const int C8 = C >> 3, C1 = C - C0;
  const int64_t total = (int64_t)N * HW * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / HW);
    const bool second = c8 * 8 >= C0;
    const int Cs = second ? C1 : C0, cl = second ? c8 * 8 - C0 : c8 * 8;
    const int64_t off = pix * Cs + cl;
    float f[8], d[8], a[8], b[8], pp[8], qq[8], o[8];
    unpack8<T>(*(const uint4*)((second ? x1 : x) + off), f);
    unpack8<T>(*(const uint4*)(dy + pix * C + c8 * 8), d);
    const int64_t co = (int64_t)n * C + c8 * 8;
    *(float4*)a = *(const float4*)(ca + co); *(float4*)(a + 4) = *(const float4*)(ca + co + 4);
    *(float4*)b = *(const float4*)(cb + co); *(float4*)(b + 4) = *(const float4*)(cb + co + 4);
    *(float4*)pp = *(const float4*)(cp + co); *(float4*)(pp + 4) = *(const float4*)(cp + co + 4);
    *(float4*)qq = *(const float4*)(cq + co); *(float4*)(qq + 4) = *(const float4*)(cq + co + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] * (d[e] * act_grad(a[e] * f[e] + b[e], act)) + pp[e] * f[e] + qq[e];
    const u16* ga = second ? gadd1 : gadd0;
    if (ga) {
      float r[8];
      unpack8<T>(*(const uint4*)(ga + off), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += r[e];
    }
    *(uint4*)((second ? dx1 : dx0) + off) = pack8<T>(o);
  }
}
}  // namespace

#define ST ((hipStream_t)s)
#define BY16(KERN, G, B, ...)                                                                      \
  do {                                                                                             \
    if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(KERN<BF16>, G, B, 0, ST, __VA_ARGS__);             \
    else if (dtype == PMI_DT_F16) hipLaunchKernelGGL(KERN<F16>, G, B, 0, ST, __VA_ARGS__);          \
    else return PMI_ERR_ARG;                                                                       \
  } while (0)

extern "C" int pmi_add16(const void* a, const void* b, void* out, int64_t n, int dtype, pmi_stream_t s) {
  if (!a || !b || !out || n <= 0 || (n & 7)) return PMI_ERR_ARG;
  BY16(add16_kernel, dim3(grid_for(n / 8)), dim3(256), (const u16*)a, (const u16*)b, (u16*)out, n / 8);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

/* dy [N][H/2][W/2][C] -> dx [N][H][W][C] */
extern "C" int pmi_avgpool2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, pmi_stream_t s) {
  if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (C & 7)) return PMI_ERR_ARG;
  BY16(avgpool2_bwd_kernel, dim3(grid_for((int64_t)N * H * W * (C / 8))), dim3(256), (const u16*)dy, (u16*)dx, N, H, W, C);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

/* dy [N][2H][2W][C] -> dx [N][H][W][C] */
extern "C" int pmi_upsample_bilinear2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, pmi_stream_t s) {
  if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || (C & 7)) return PMI_ERR_ARG;
  BY16(upsample_bilinear2_bwd_kernel, dim3(grid_for((int64_t)N * H * W * (C / 8))), dim3(256), (const u16*)dy, (u16*)dx, N, H, W, C);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

/* dy [N][2H][2W][C] -> dx [N][H][W][C] */
extern "C" int pmi_upsample_nearest2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, pmi_stream_t s) {
  if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || (C & 7)) return PMI_ERR_ARG;
  BY16(upsample_nearest2_bwd_kernel, dim3(grid_for((int64_t)N * H * W * (C / 8))), dim3(256), (const u16*)dy, (u16*)dx, N, H, W, C);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

/* x, dy, res (optional), dx: [N][hw][C] 16-bit; the scale of channel c of sample n is gamma[n * gamma_ld + c] + gamma_add;
 * partial: workspace of N * pmi_gn1_bwd_partials(hw, C) * 4 doubles */
extern "C" int pmi_gn1_bwd_partials(int64_t hw, int C) {
  const int64_t n8 = hw * (C / 8);
  int64_t p = n8 / 4096;                                       // >= 16 pieces of 16 bytes per thread and slice
  return (int)(p < 1 ? 1 : (p > 256 ? 256 : p));
}

extern "C" int pmi_gn1_bwd(const void* x, const void* dy, const float* gamma, int gamma_ld, float gamma_add, const void* res, void* dx,
                           double* partial, int N, int64_t hw, int C, float eps, int dtype, pmi_stream_t s) {
  if (!x || !dy || !gamma || !dx || !partial || N <= 0 || hw <= 0 || C <= 0 || (C & 7) || gamma_ld < 0) return PMI_ERR_ARG;
  const int P = pmi_gn1_bwd_partials(hw, C);
  BY16(gn1_bwd_reduce_kernel, dim3(P, N), dim3(GT), (const u16*)x, (const u16*)dy, gamma, gamma_ld, gamma_add, partial, hw, C, P);
  PMI_CHECK_LAUNCH();
  BY16(gn1_bwd_apply_kernel, dim3(P, N), dim3(GT), (const u16*)x, (const u16*)dy, gamma, gamma_ld, gamma_add, (const u16*)res, (u16*)dx,
       partial, hw, C, P, eps);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

extern "C" int pmi_gn_bwd_stats(const void* x, const void* x1, int C0, const void* dy, const float* coef_a, const float* coef_b, int act, float* ws,
                                int N, int HW, int C, int nchunk, int dtype, pmi_stream_t s) {
  if (!x || !dy || !coef_a || !coef_b || !ws || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || (C0 & 7) || C0 > C || (C0 < C && !x1) || nchunk <= 0 ||
      dtype == PMI_DT_F16X2)
    return PMI_ERR_ARG;
  const int C8 = C >> 3, TPP = C8 < 256 ? C8 : 256, PPI = 256 / TPP;
  if (PPI * C > 4096) return PMI_ERR_ARG;
  dim3 grid(nchunk, N), block(256);
  hipStream_t st = (hipStream_t)s;
  if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(gn_bwd_stats_kernel<BF16>, grid, block, 0, st, (const u16*)x, (const u16*)x1, C0, (const u16*)dy, coef_a, coef_b, act, ws, HW, C, nchunk);
  else hipLaunchKernelGGL(gn_bwd_stats_kernel<F16>, grid, block, 0, st, (const u16*)x, (const u16*)x1, C0, (const u16*)dy, coef_a, coef_b, act, ws, HW, C, nchunk);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_gn_bwd_finalize(const float* s0, int P0, int C0, const float* s1, int P1, int C1, const float* ws_bwd, int PB, const float* gamma,
                                   const float* film, int film_ld, float* coef_p, float* coef_q, int N, int HW, int G, float eps, pmi_stream_t s) {
  const int C = C0 + C1;
  if (!s0 || !ws_bwd || !coef_p || !coef_q || N <= 0 || HW <= 0 || G <= 0 || C <= 0 || (C % G) || P0 <= 0 || PB <= 0 || (C1 > 0 && (!s1 || P1 <= 0)))
    return PMI_ERR_ARG;
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N, G), dim3(256), 0, (hipStream_t)s, s0, P0, C0, s1, P1, C1, ws_bwd, PB, gamma, film, film_ld,
                     coef_p, coef_q, HW, G, eps);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_gn_bwd_apply(const void* x, const void* x1, int C0, const void* dy, const float* coef_a, const float* coef_b, const float* coef_p,
                                const float* coef_q, int act, const void* gadd0, const void* gadd1, void* dx0, void* dx1, int N, int HW, int C,
                                int dtype, pmi_stream_t s) {
  if (!x || !dy || !coef_a || !coef_b || !coef_p || !coef_q || !dx0 || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || (C0 & 7) || C0 > C ||
      (C0 < C && (!x1 || !dx1)) || dtype == PMI_DT_F16X2)
    return PMI_ERR_ARG;
  const int64_t total = (int64_t)N * HW * (C / 8);
  const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipStream_t st = (hipStream_t)s;
  if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(gn_bwd_apply_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const u16*)x, (const u16*)x1, C0, (const u16*)dy, coef_a, coef_b, coef_p, coef_q, act, (const u16*)gadd0, (const u16*)gadd1, (u16*)dx0, (u16*)dx1, N, HW, C);
  else hipLaunchKernelGGL(gn_bwd_apply_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const u16*)x, (const u16*)x1, C0, (const u16*)dy, coef_a, coef_b, coef_p, coef_q, act, (const u16*)gadd0, (const u16*)gadd1, (u16*)dx0, (u16*)dx1, N, HW, C);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
