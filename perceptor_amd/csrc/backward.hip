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
