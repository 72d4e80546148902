// Layout conversion, pooling / upsampling, timestep features and the fused sampler updates.
// All HBM-bound streaming kernels: grid-stride, 16-byte accesses where the layout allows.
#include "common.h"
#include "../../include/perceptor_hip.h"

namespace {

inline int grid_for(int64_t work) {
  int64_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// NCHW fp32 image in [0,1] -> NHWC 16-bit x = 2*img-1, channels [3,3+nplanes) constant planes, rest zero.
template <typename T>
__global__ __launch_bounds__(256) void prep_input_kernel(const float* __restrict__ img, const float* __restrict__ planes,
                                                         int nplanes, u16* __restrict__ x, int N, int HW, int Cpad) {
  const int C8 = Cpad >> 3;
  const int64_t total = (int64_t)N * HW * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / HW);
    const int64_t p = pix - (int64_t)n * HW;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 * 8 + e;
      float v = 0.f;
      if (c < 3) v = img[((int64_t)n * 3 + c) * HW + p] * 2.f - 1.f;
      else if (c < 3 + nplanes) v = planes[(int64_t)n * nplanes + (c - 3)];
      f[e] = v;
    }
    store8<T>(x + pix * row_elems<T>(Cpad), c8 * 8, Cpad, f);
  }
}

__global__ __launch_bounds__(256) void finish_output_kernel(const float* __restrict__ y, int ld, float* __restrict__ out,
                                                            int N, int HW, int cout) {
  const int64_t total = (int64_t)N * cout * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % HW;
    const int64_t nc = i / HW;
    const int c = (int)(nc % cout), n = (int)(nc / cout);
    out[i] = y[((int64_t)n * HW + p) * ld + c];
  }
}

// NCHW fp32 (C channels) -> NHWC 16-bit with Cpad channels, x = in*mul + add, zero padding channels (SD latents / VAE input)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, u16* __restrict__ x, int N, int C, int HW, int Cpad,
                                                           float mul, float add) {
  const int C8 = Cpad >> 3;
  const int64_t total = (int64_t)N * HW * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / HW);
    const int64_t p = pix - (int64_t)n * HW;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 * 8 + e;
      f[e] = c < C ? in[((int64_t)n * C + c) * HW + p] * mul + add : 0.f;
    }
    store8<T>(x + pix * row_elems<T>(Cpad), c8 * 8, Cpad, f);
  }
}

// NHWC fp32 (ld channels per pixel) -> NCHW fp32 first cout channels, out = y*mul + add
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ y, int ld, float* __restrict__ out, int N, int HW, int cout,
                                                           float mul, float add) {
  const int64_t total = (int64_t)N * cout * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % HW;
    const int64_t nc = i / HW;
    const int c = (int)(nc % cout), n = (int)(nc / cout);
    out[i] = y[((int64_t)n * HW + p) * ld + c] * mul + add;
  }
}

// GEGLU (stable_diffusion/attention.py:346-348): h[M][2F] 16-bit = (value | gate) -> out[M][F] = value * gelu(gate)
template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const u16* __restrict__ h, u16* __restrict__ out, int64_t M, int F, int interleaved) {
  const int F8 = F >> 3;
  const int64_t total = M * F8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % F8);
    const int64_t r = i / F8;
    float v[8], g[8];
    // plain: (value F | gate F); interleaved: per 32 input columns 16 value then 16 gate (the layout the fused GEMM epilogue's weights use)
    const int vo = interleaved ? (c8 >> 1) * 32 + (c8 & 1) * 8 : c8 * 8, go = interleaved ? vo + 16 : F + c8 * 8;
    unpack8<T>(*(const uint4*)(h + r * 2 * F + vo), v);
    unpack8<T>(*(const uint4*)(h + r * 2 * F + go), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= 0.5f * g[e] * (1.f + fast_erff(g[e] * 0.70710678118654752f));
    *(uint4*)(out + r * F + c8 * 8) = pack8<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool2_kernel(const u16* __restrict__ x, u16* __restrict__ y, int N, int H, int W, int C) {
  const int C8 = C >> 3, Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / ((int64_t)Ho * Wo));
    const int rem = (int)(pix - (int64_t)n * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const int ld = row_elems<T>(C);
    const u16* b = x + (((int64_t)n * H + 2 * oy) * W + 2 * ox) * ld;
    float a0[8], a1[8], a2[8], a3[8], o[8];
    load8<T>(b, c8 * 8, C, a0); load8<T>(b + ld, c8 * 8, C, a1);
    load8<T>(b + (int64_t)W * ld, c8 * 8, C, a2); load8<T>(b + (int64_t)W * ld + ld, c8 * 8, C, a3);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.25f * (a0[e] + a1[e] + a2[e] + a3[e]);
    store8<T>(y + pix * ld, c8 * 8, C, o);
  }
}

// bilinear x2, align_corners=False: even o=2i -> .25 x[i-1] + .75 x[i]; odd o=2i+1 -> .75 x[i] + .25 x[i+1] (clamped)
template <typename T>
__global__ __launch_bounds__(256) void upsample_bilinear2_kernel(const u16* __restrict__ x, u16* __restrict__ y, int N, int H, int W, int C) {
  const int C8 = C >> 3, Ho = H * 2, Wo = W * 2;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / ((int64_t)Ho * Wo));
    const int rem = (int)(pix - (int64_t)n * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const int iy = oy >> 1, ix = ox >> 1;
    const int y1 = (oy & 1) ? min(iy + 1, H - 1) : max(iy - 1, 0);
    const int x1 = (ox & 1) ? min(ix + 1, W - 1) : max(ix - 1, 0);
    const int ld = row_elems<T>(C);
    const u16* b = x + (int64_t)n * H * W * ld;
    float a00[8], a01[8], a10[8], a11[8], o[8];
    load8<T>(b + ((int64_t)iy * W + ix) * ld, c8 * 8, C, a00);
    load8<T>(b + ((int64_t)iy * W + x1) * ld, c8 * 8, C, a01);
    load8<T>(b + ((int64_t)y1 * W + ix) * ld, c8 * 8, C, a10);
    load8<T>(b + ((int64_t)y1 * W + x1) * ld, c8 * 8, C, a11);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.75f * (0.75f * a00[e] + 0.25f * a01[e]) + 0.25f * (0.75f * a10[e] + 0.25f * a11[e]);
    store8<T>(y + pix * ld, c8 * 8, C, o);
  }
}

// nearest x2 (wikiart_256.py:117): 16-byte copies
__global__ __launch_bounds__(256) void upsample_nearest2_kernel(const u16* __restrict__ x, u16* __restrict__ y, int N, int H, int W, int C) {
  const int C8 = C >> 3, Ho = H * 2, Wo = W * 2;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / ((int64_t)Ho * Wo));
    const int rem = (int)(pix - (int64_t)n * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    *(uint4*)(y + pix * C + c8 * 8) = *(const uint4*)(x + (((int64_t)n * H + (oy >> 1)) * W + (ox >> 1)) * C + c8 * 8);
  }
}

template <typename T>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, u16* __restrict__ out, int N, int dim, float max_period) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * half) return;
  const int n = i / half, j = i - n * half;
  const float freq = expf(-logf(max_period) * (float)j / (float)half);
  const float arg = t[n] * freq;
  if constexpr (is_split<T>::v) {            // precise mode: fp32 features for the fp32 time MLP
    ((float*)out)[(int64_t)n * dim + j] = cosf(arg);
    ((float*)out)[(int64_t)n * dim + half + j] = sinf(arg);
  } else {
    out[(int64_t)n * dim + j] = T::from_f(cosf(arg));
    out[(int64_t)n * dim + half + j] = T::from_f(sinf(arg));
  }
}

// fp32 [rows][C] <-> precise (hi + lo f16 pairs) [rows][2C]
__global__ __launch_bounds__(256) void split_from_f32_kernel(const float* __restrict__ in, int ld_in, u16* __restrict__ out, int64_t rows, int C) {
  const int C8 = C >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C8; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    float f[8];
    *(float4*)f = *(const float4*)(in + r * ld_in + c8 * 8);
    *(float4*)(f + 4) = *(const float4*)(in + r * ld_in + c8 * 8 + 4);
    store8<F16X2>(out + r * 2 * C, c8 * 8, C, f);
  }
}
__global__ __launch_bounds__(256) void split_to_f32_kernel(const u16* __restrict__ in, float* __restrict__ out, int64_t rows, int C) {
  const int C8 = C >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C8; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    float f[8];
    load8<F16X2>(in + r * 2 * C, c8 * 8, C, f);
    *(float4*)(out + r * C + c8 * 8) = *(const float4*)f;
    *(float4*)(out + r * C + c8 * 8 + 4) = *(const float4*)(f + 4);
  }
}

// plain f16 [rows][C] <-> split (hi + lo) [rows][2C]: the level boundaries of the mixed mode (engine/adm_mixed.py).  Towards the plain side the
// value hi + lo is rounded once; towards the split side the low parts are zero.
__global__ __launch_bounds__(256) void split_convert_kernel(const u16* __restrict__ in, u16* __restrict__ out, int64_t rows, int C, int to_split) {
  const int C8 = C >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C8; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    float f[8];
    if (to_split) {
      load8<F16>(in + r * C, c8 * 8, C, f);
      store8<F16X2>(out + r * 2 * C, c8 * 8, C, f);
    } else {
      load8<F16X2>(in + r * 2 * C, c8 * 8, C, f);
      store8<F16>(out + r * C, c8 * 8, C, f);
    }
  }
}

__global__ void fourier_features_kernel(const float* __restrict__ t, const float* __restrict__ w, float* __restrict__ out, int N, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * half) return;
  const int n = i / half, j = i - n * half;
  const float f = 6.283185307179586f * t[n] * w[j];
  out[(int64_t)n * 2 * half + j] = cosf(f);
  out[(int64_t)n * 2 * half + half + j] = sinf(f);
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ in, u16* __restrict__ out, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = T::from_f(act_apply(in[i], act));
}

// x0 = (x - s_f*eps)/max(a_f,1e-7); x' = x0*a_t + eps*s_t ; images = (x+1)/2
__global__ __launch_bounds__(256) void ddim_eps_kernel(const float* __restrict__ img, const float* __restrict__ eps,
                                                       const float* af, const float* sf, const float* at, const float* st,
                                                       float* __restrict__ next, float* __restrict__ den, int N, int64_t chw) {
  const int64_t total = (int64_t)N * chw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / chw);
    const float x = img[i] * 2.f - 1.f, e = eps[i];
    const float x0 = (x - sf[n] * e) / fmaxf(af[n], 1e-7f);
    if (next) next[i] = (x0 * at[n] + e * st[n] + 1.f) * 0.5f;
    if (den) den[i] = (x0 + 1.f) * 0.5f;
  }
}

// x0 = x*a_f - v*s_f ; eps = x*s_f + v*a_f ; x' = x0*a_t + eps*s_t
__global__ __launch_bounds__(256) void ddim_v_kernel(const float* __restrict__ img, const float* __restrict__ v,
                                                     const float* af, const float* sf, const float* at, const float* st,
                                                     float* __restrict__ next, float* __restrict__ den, int N, int64_t chw) {
  const int64_t total = (int64_t)N * chw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / chw);
    const float x = img[i] * 2.f - 1.f, vv = v[i];
    const float x0 = x * af[n] - vv * sf[n];
    const float e = x * sf[n] + vv * af[n];
    if (next) next[i] = (x0 * at[n] + e * st[n] + 1.f) * 0.5f;
    if (den) den[i] = (x0 + 1.f) * 0.5f;
  }
}

__global__ __launch_bounds__(256) void guided_kernel(const float* __restrict__ pred, const float* __restrict__ grad,
                                                     const float* sf, float scale, float cv, float* __restrict__ out,
                                                     int N, int64_t chw) {
  const int64_t total = (int64_t)N * chw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / chw);
    const float g = fminf(fmaxf(grad[i], -cv), cv);
    out[i] = pred[i] + scale * sf[n] * g / cv;
  }
}

// out = ca[n]*a + cb[n]*b + cc[n]  (b / cb optional): every remaining Predictions formula is of this form
__global__ __launch_bounds__(256) void lincomb2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* ca, const float* cb, const float* cc,
                                                       float* __restrict__ out, int N, int64_t chw) {
  const int64_t total = (int64_t)N * chw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / chw);
    float v = ca[n] * a[i];
    if (b) v += cb[n] * b[i];
    if (cc) v += cc[n];
    out[i] = v;
  }
}

// clamp with per-sample bounds (static / dynamic thresholding forward)
__global__ __launch_bounds__(256) void clamp_kernel(const float* __restrict__ a, const float* lo, const float* hi,
                                                    float* __restrict__ out, int N, int64_t chw) {
  const int64_t total = (int64_t)N * chw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / chw);
    out[i] = fminf(fmaxf(a[i], lo[n]), hi[n]);
  }
}

}  // namespace

#define ST ((hipStream_t)s)
#define BY_DTYPE(KERN, ...)                                                              \
  do {                                                                                   \
    if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(KERN<BF16>, grid, block, 0, ST, __VA_ARGS__); \
    else if (dtype == PMI_DT_F16X2) hipLaunchKernelGGL(KERN<F16X2>, grid, block, 0, ST, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERN<F16>, grid, block, 0, ST, __VA_ARGS__);                 \
  } while (0)

extern "C" int pmi_prep_input(const float* img, const float* planes, int nplanes, void* x, int N, int H, int W, int Cpad, int dtype, pmi_stream_t s) {
  if (!img || !x || N <= 0 || H <= 0 || W <= 0 || (Cpad & 7) || Cpad < 3 + nplanes || (nplanes > 0 && !planes)) return PMI_ERR_ARG;
  dim3 grid(grid_for((int64_t)N * H * W * (Cpad / 8))), block(256);
  BY_DTYPE(prep_input_kernel, img, planes, nplanes, (u16*)x, N, H * W, Cpad);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_finish_output(const float* y, int ld, float* out, int N, int H, int W, int cout, pmi_stream_t s) {
  if (!y || !out || N <= 0 || cout <= 0 || cout > ld) return PMI_ERR_ARG;
  hipLaunchKernelGGL(finish_output_kernel, dim3(grid_for((int64_t)N * cout * H * W)), dim3(256), 0, ST, y, ld, out, N, H * W, cout);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_nchw_to_nhwc(const float* in, void* x, int N, int C, int H, int W, int Cpad, float mul, float add, int dtype, pmi_stream_t s) {
  if (!in || !x || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (Cpad & 7) || Cpad < C) return PMI_ERR_ARG;
  dim3 grid(grid_for((int64_t)N * H * W * (Cpad / 8))), block(256);
  BY_DTYPE(nchw_to_nhwc_kernel, in, (u16*)x, N, C, H * W, Cpad, mul, add);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_nhwc_to_nchw(const float* y, int ld, float* out, int N, int H, int W, int cout, float mul, float add, pmi_stream_t s) {
  if (!y || !out || N <= 0 || cout <= 0 || cout > ld) return PMI_ERR_ARG;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((int64_t)N * cout * H * W)), dim3(256), 0, ST, y, ld, out, N, H * W, cout, mul, add);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_geglu(const void* h, void* out, int64_t M, int F, int interleaved, int dtype, pmi_stream_t s) {
  if (!h || !out || M <= 0 || F <= 0 || (F & 7) || (interleaved && (F & 15)) || dtype == PMI_DT_F16X2) return PMI_ERR_ARG;
  dim3 grid(grid_for(M * (F / 8))), block(256);
  if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(geglu_kernel<BF16>, grid, block, 0, ST, (const u16*)h, (u16*)out, M, F, interleaved);
  else hipLaunchKernelGGL(geglu_kernel<F16>, grid, block, 0, ST, (const u16*)h, (u16*)out, M, F, interleaved);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_avgpool2(const void* x, void* y, int N, int H, int W, int C, int dtype, pmi_stream_t s) {
  if (!x || !y || N <= 0 || (H & 1) || (W & 1) || (C & 7)) return PMI_ERR_ARG;
  dim3 grid(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), block(256);
  BY_DTYPE(avgpool2_kernel, (const u16*)x, (u16*)y, N, H, W, C);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_upsample_bilinear2(const void* x, void* y, int N, int H, int W, int C, int dtype, pmi_stream_t s) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || (C & 7)) return PMI_ERR_ARG;
  dim3 grid(grid_for((int64_t)N * H * W * 4 * (C / 8))), block(256);
  BY_DTYPE(upsample_bilinear2_kernel, (const u16*)x, (u16*)y, N, H, W, C);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_upsample_nearest2(const void* x, void* y, int N, int H, int W, int C, pmi_stream_t s) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || (C & 7)) return PMI_ERR_ARG;
  hipLaunchKernelGGL(upsample_nearest2_kernel, dim3(grid_for((int64_t)N * H * W * 4 * (C / 8))), dim3(256), 0, ST, (const u16*)x, (u16*)y, N, H, W, C);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_timestep_embedding(const float* t, void* out, int N, int dim, float max_period, int dtype, pmi_stream_t s) {
  if (!t || !out || N <= 0 || dim <= 0 || (dim & 1)) return PMI_ERR_ARG;
  dim3 grid((N * dim / 2 + 255) / 256), block(256);
  BY_DTYPE(timestep_embedding_kernel, t, (u16*)out, N, dim, max_period);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_split_convert(const void* in, void* out, int64_t rows, int C, int to_split, pmi_stream_t s) {
  if (!in || !out || rows <= 0 || C <= 0 || (C & 7) || (C > 32 && (C & 31))) return PMI_ERR_ARG;
  hipLaunchKernelGGL(split_convert_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, ST, (const u16*)in, (u16*)out, rows, C, to_split);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_split_from_f32(const float* in, int ld_in, void* out, int64_t rows, int C, pmi_stream_t s) {
  if (!in || !out || rows <= 0 || C <= 0 || (C & 7) || (C > 32 && (C & 31)) || (ld_in & 3)) return PMI_ERR_ARG;
  hipLaunchKernelGGL(split_from_f32_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, ST, in, ld_in, (u16*)out, rows, C);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_split_to_f32(const void* in, float* out, int64_t rows, int C, pmi_stream_t s) {
  if (!in || !out || rows <= 0 || C <= 0 || (C & 7) || (C > 32 && (C & 31))) return PMI_ERR_ARG;
  hipLaunchKernelGGL(split_to_f32_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, ST, (const u16*)in, out, rows, C);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_fourier_features(const float* t, const float* w, float* out, int N, int half, pmi_stream_t s) {
  if (!t || !w || !out || N <= 0 || half <= 0) return PMI_ERR_ARG;
  hipLaunchKernelGGL(fourier_features_kernel, dim3((N * half + 255) / 256), dim3(256), 0, ST, t, w, out, N, half);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_cast_f32_to_16(const float* in, void* out, int64_t n, int act, int dtype, pmi_stream_t s) {
  if (!in || !out || n <= 0) return PMI_ERR_ARG;
  dim3 grid(grid_for(n)), block(256);
  BY_DTYPE(cast_kernel, in, (u16*)out, n, act);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_ddim_eps_step(const float* img, const float* eps, const float* a_from, const float* s_from, const float* a_to,
                                 const float* s_to, float* next_img, float* denoised_img, int N, int64_t chw, pmi_stream_t s) {
  if (!img || !eps || !a_from || !s_from || N <= 0 || chw <= 0 || (next_img && (!a_to || !s_to))) return PMI_ERR_ARG;
  hipLaunchKernelGGL(ddim_eps_kernel, dim3(grid_for(N * chw)), dim3(256), 0, ST, img, eps, a_from, s_from, a_to, s_to, next_img, denoised_img, N, chw);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_ddim_v_step(const float* img, const float* v, const float* a_from, const float* s_from, const float* a_to,
                               const float* s_to, float* next_img, float* denoised_img, int N, int64_t chw, pmi_stream_t s) {
  if (!img || !v || !a_from || !s_from || N <= 0 || chw <= 0 || (next_img && (!a_to || !s_to))) return PMI_ERR_ARG;
  hipLaunchKernelGGL(ddim_v_kernel, dim3(grid_for(N * chw)), dim3(256), 0, ST, img, v, a_from, s_from, a_to, s_to, next_img, denoised_img, N, chw);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_guided_update(const float* pred, const float* grad, const float* s_from, float scale, float clamp_value,
                                 float* out, int N, int64_t chw, pmi_stream_t s) {
  if (!pred || !grad || !s_from || !out || N <= 0 || chw <= 0 || !(clamp_value > 0.f)) return PMI_ERR_ARG;
  hipLaunchKernelGGL(guided_kernel, dim3(grid_for(N * chw)), dim3(256), 0, ST, pred, grad, s_from, scale, clamp_value, out, N, chw);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_lincomb2(const float* a, const float* b, const float* ca, const float* cb, const float* cc, float* out, int N,
                            int64_t chw, pmi_stream_t s) {
  if (!a || !ca || !out || N <= 0 || chw <= 0 || (b && !cb)) return PMI_ERR_ARG;
  hipLaunchKernelGGL(lincomb2_kernel, dim3(grid_for(N * chw)), dim3(256), 0, ST, a, b, ca, cb, cc, out, N, chw);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_clamp(const float* a, const float* lo, const float* hi, float* out, int N, int64_t chw, pmi_stream_t s) {
  if (!a || !lo || !hi || !out || N <= 0 || chw <= 0) return PMI_ERR_ARG;
  hipLaunchKernelGGL(clamp_kernel, dim3(grid_for(N * chw)), dim3(256), 0, ST, a, lo, hi, out, N, chw);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
