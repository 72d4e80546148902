// GroupNorm (statistics, coefficient finalisation, fused apply) and LayerNorm (fwd / input-grad)
// for gfx950.  All HBM-bound: 16-byte vector accesses, fp32 statistics, wave64 reductions.
#include "common.h"
#include "../../include/perceptor_hip.h"

namespace {

constexpr int MAXC = 4096;

// ---- GroupNorm statistics: partial (sum, sumsq) per (sample, pixel chunk, group) -------------
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const u16* __restrict__ x, const u16* __restrict__ x1, int C0,
                                                       float* __restrict__ ws, int HW, int C, int G, int nchunk) {
  __shared__ float s_sum[MAXC], s_sq[MAXC];      // [PPI][C] slots, one per pixel lane (PPI * C <= MAXC): no float atomics
  const int tid = threadIdx.x, chunk = blockIdx.x, n = blockIdx.y;
  const int C8 = C >> 3;
  const int TPP = C8 < 256 ? C8 : 256;      // threads per pixel
  const int PPI = 256 / TPP;                // pixels per iteration
  const int my_p = tid / TPP, my_c = tid - my_p * TPP;
  const int ppc = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  const int C1 = C - C0;
  if (my_p < PPI) {
    for (int c8 = my_c; c8 < C8; c8 += TPP) {
      const bool second = c8 * 8 >= C0;
      const int Cs = second ? C1 : C0, cl = second ? c8 * 8 - C0 : c8 * 8;    // source tensor's channel count, channel inside it
      const int ld = row_elems<T>(Cs);
      const u16* xb = (second ? x1 : x) + (int64_t)n * HW * ld;
      float s[8], q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
      for (int p = p0 + my_p; p < p1; p += PPI) {
        float f[8];
        load8<T>(xb + (int64_t)p * ld, cl, Cs, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { s_sum[my_p * C + c8 * 8 + e] = s[e]; s_sq[my_p * C + c8 * 8 + e] = q[e]; }
    }
  }
  __syncthreads();
  float* o = ws + ((int64_t)n * nchunk + chunk) * C * 2;
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int pl = 0; pl < PPI; ++pl) { a += s_sum[pl * C + c]; b += s_sq[pl * C + c]; }     // fixed order
    o[2 * c] = a; o[2 * c + 1] = b;
  }
}

// ---- finalize: y = act(x * a[n][c] + b[n][c]) with affine and FiLM folded in -------------------
// One workgroup per (sample, group, slice).  Partials are per channel: s0[n][P0][C0][2] (+ s1 for the 2nd concat source); the
// group's (sum, sumsq) pairs of a partial row are contiguous, so the reduction is a flat coalesced walk.  With gridDim.z == 1
// the workgroup writes the coefficients itself.  With few (sample, group) pairs -- GroupNorm(1, C) of the v-diffusion nets at
// batch 1 is ONE pair -- the rows are split over gridDim.z slices that each leave a (sum, sumsq) pair of doubles in the
// group's own slots of coef_b, and gn_finalize2_kernel adds them in order (a single workgroup took 120 us per call there:
// 16 of the 23 ms of a cc12m_1 step).
__device__ __forceinline__ void gn_coeffs_out(double s, double q, int n, int g, int C, int cpg, int HW, float eps,
                                              const float* gamma, const float* beta, const float* film, int film_ld,
                                              float* ca, float* cb, int tid) {
  const double cnt = (double)HW * cpg;
  const double mean = s / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float fmean = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int j = tid; j < cpg; j += 256) {
    const int c = g * cpg + j;
    float a = rstd * (gamma ? gamma[c] : 1.f);
    float b = (beta ? beta[c] : 0.f) - fmean * a;
    if (film) {
      const float sc = 1.f + film[(int64_t)n * film_ld + c], sh = film[(int64_t)n * film_ld + C + c];
      a *= sc; b = b * sc + sh;
    }
    ca[(int64_t)n * C + c] = a; cb[(int64_t)n * C + c] = b;
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ s0, int P0, int C0,
                                                          const float* __restrict__ s1, int P1, int C1,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ film, int film_ld, float* __restrict__ ca,
                                                          float* __restrict__ cb, int HW, int G, float eps) {
  __shared__ double red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, n = blockIdx.x, g = blockIdx.y;
  const int slice = blockIdx.z, S = gridDim.z;
  const int C = C0 + C1, cpg = C / G;
  double s = 0.0, q = 0.0;
  auto walk = [&](const float* base, int P, int Csrc, int c_lo, int c_hi) {     // channels [c_lo, c_hi) of one source
    const int w = c_hi - c_lo;
    if (w <= 0) return;
    float fs = 0.f, fq = 0.f;
    int cnt = 0;
    if (((w | c_lo | Csrc) & 1) == 0) {             // two channels (16 bytes) per load: half the iterations of a latency-bound walk
      const int w2 = w >> 1;
      const int64_t total = (int64_t)P * w2;
      for (int64_t i = (int64_t)slice * 256 + tid; i < total; i += (int64_t)S * 256) {
        const int pr = (int)(i / w2), j = (int)(i - (int64_t)pr * w2);
        const float4 v = *(const float4*)(base + ((int64_t)pr * Csrc + c_lo + 2 * j) * 2);
        fs += v.x; fq += v.y;
        fs += v.z; fq += v.w;                       // (same left-to-right order as the one-channel walk for a thread's elements)
        if (++cnt == 32) { s += fs; q += fq; fs = fq = 0.f; cnt = 0; }
      }
    } else {
      const int64_t total = (int64_t)P * w;
      for (int64_t i = (int64_t)slice * 256 + tid; i < total; i += (int64_t)S * 256) {
        const int pr = (int)(i / w), j = (int)(i - (int64_t)pr * w);
        const float2 v = *(const float2*)(base + ((int64_t)pr * Csrc + c_lo + j) * 2);
        fs += v.x; fq += v.y;
        if (++cnt == 64) { s += fs; q += fq; fs = fq = 0.f; cnt = 0; }   // short fp32 runs, double across them
      }
    }
    s += fs; q += fq;
  };
  const int g_lo = g * cpg, g_hi = g_lo + cpg;
  walk(s0 + (int64_t)n * P0 * C0 * 2, P0, C0, min(g_lo, C0), min(g_hi, C0));
  if (s1) walk(s1 + (int64_t)n * P1 * C1 * 2, P1, C1, max(g_lo, C0) - C0, max(g_hi, C0) - C0);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
  if (lane == 0) { red[0][wid] = s; red[1][wid] = q; }
  __syncthreads();
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  if (S > 1) {
    if (tid == 0) {
      double* part = (double*)(cb + (int64_t)n * C + g_lo) + 2 * slice;
      part[0] = s; part[1] = q;
    }
    return;
  }
  gn_coeffs_out(s, q, n, g, C, cpg, HW, eps, gamma, beta, film, film_ld, ca, cb, tid);
}

__global__ __launch_bounds__(256) void gn_finalize2_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ film, int film_ld, float* __restrict__ ca,
                                                           float* __restrict__ cb, int C, int HW, int G, int S, float eps) {
  __shared__ double tot[2];
  const int tid = threadIdx.x, n = blockIdx.x, g = blockIdx.y, cpg = C / G;
  if (tid == 0) {
    const double* part = (const double*)(cb + (int64_t)n * C + g * cpg);
    double s = 0.0, q = 0.0;
    for (int i = 0; i < S; ++i) { s += part[2 * i]; q += part[2 * i + 1]; }     // fixed order
    tot[0] = s; tot[1] = q;
  }
  __syncthreads();       // the partials are consumed before the coefficients overwrite their slots
  gn_coeffs_out(tot[0], tot[1], n, g, C, cpg, HW, eps, gamma, beta, film, film_ld, ca, cb, tid);
}

template <typename T, bool RAW = false>
__device__ __forceinline__ void affine_act8(const u16* row, int c, int C, const float* a, const float* b, int act, float* out, float w, float* raw = nullptr) {
  float f[8];
  load8<T>(row, c, C, f);
#pragma unroll
  for (int e = 0; e < 8; ++e) out[e] += w * act_apply(f[e] * a[e] + b[e], act);
  if constexpr (RAW) {
#pragma unroll
    for (int e = 0; e < 8; ++e) raw[e] += w * f[e];
  }
}

template <typename T, bool POOL, bool RAW = false>      // RAW (with POOL): second output = the 2x2 average of the raw input
__global__ __launch_bounds__(256) void gn_apply_kernel(const u16* __restrict__ x, const u16* __restrict__ x1, int C0,
                                                       const float* __restrict__ ca, const float* __restrict__ cb,
                                                       const u16* __restrict__ res, u16* __restrict__ y, int N, int H, int W,
                                                       int C, int act, u16* __restrict__ y_raw = nullptr) {
  const int C8 = C >> 3;
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int n = (int)(pix / ((int64_t)Ho * Wo));
    float a[8], b[8], o[8];
    *(float4*)a = *(const float4*)(ca + (int64_t)n * C + c8 * 8);
    *(float4*)(a + 4) = *(const float4*)(ca + (int64_t)n * C + c8 * 8 + 4);
    *(float4*)b = *(const float4*)(cb + (int64_t)n * C + c8 * 8);
    *(float4*)(b + 4) = *(const float4*)(cb + (int64_t)n * C + c8 * 8 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    const bool second = c8 * 8 >= C0;
    const u16* src = second ? x1 : x;
    const int Cs = second ? C - C0 : C0, cl = second ? c8 * 8 - C0 : c8 * 8;
    const int ld = row_elems<T>(Cs);
    if (POOL) {
      const int rem = (int)(pix - (int64_t)n * Ho * Wo);
      const int oy = rem / Wo, ox = rem - oy * Wo;
      const u16* base = src + (((int64_t)n * H + 2 * oy) * W + 2 * ox) * ld;
      float rw[8];                                     // the down ResBlock's skip path: AvgPool2d(2) of the RAW input, same four reads
#pragma unroll
      for (int e = 0; e < 8; ++e) rw[e] = 0.f;
      affine_act8<T, RAW>(base, cl, Cs, a, b, act, o, 0.25f, rw);
      affine_act8<T, RAW>(base + ld, cl, Cs, a, b, act, o, 0.25f, rw);
      affine_act8<T, RAW>(base + (int64_t)W * ld, cl, Cs, a, b, act, o, 0.25f, rw);
      affine_act8<T, RAW>(base + (int64_t)W * ld + ld, cl, Cs, a, b, act, o, 0.25f, rw);
      if constexpr (RAW) store8<T>(y_raw + pix * row_elems<T>(C), c8 * 8, C, rw);
    } else {
      affine_act8<T>(src + pix * ld, cl, Cs, a, b, act, o, 1.f);
    }
    const int ldy = row_elems<T>(C);
    if (res) {
      float r[8];
      load8<T>(res + pix * ldy, c8 * 8, C, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += r[e];
    }
    store8<T>(y + pix * ldy, c8 * 8, C, o);
  }
}

inline int grid_for(int64_t work) {
  int64_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 * 4 ? 2048 * 4 : b));
}

}  // namespace

extern "C" int pmi_gn_stats(const void* x, const void* x1, int C0, float* ws, int N, int HW, int C, int G, int nchunk, int dtype, pmi_stream_t s) {
  if (!x1) C0 = C;
  if ((C0 & 7) || C0 <= 0 || C0 > C) return PMI_ERR_ARG;
  if (!x || !ws || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || C > MAXC || G <= 0 || C % G || nchunk <= 0) return PMI_ERR_ARG;
  dim3 grid(nchunk, N), block(256);
  if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(gn_stats_kernel<BF16>, grid, block, 0, (hipStream_t)s, (const u16*)x, (const u16*)x1, C0, ws, HW, C, G, nchunk);
  else if (dtype == PMI_DT_F16X2) hipLaunchKernelGGL(gn_stats_kernel<F16X2>, grid, block, 0, (hipStream_t)s, (const u16*)x, (const u16*)x1, C0, ws, HW, C, G, nchunk);
  else hipLaunchKernelGGL(gn_stats_kernel<F16>, grid, block, 0, (hipStream_t)s, (const u16*)x, (const u16*)x1, C0, ws, HW, C, G, nchunk);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

extern "C" int pmi_gn_finalize(const float* s0, int P0, int C0, const float* s1, int P1, int C1, const float* gamma, const float* beta,
                               const float* film, int film_ld, float* coef_a, float* coef_b, int N, int HW, int G, float eps,
                               pmi_stream_t s) {
  if (!s1) { C1 = 0; P1 = 0; }
  const int C = C0 + C1;
  if (!s0 || !coef_a || !coef_b || N <= 0 || C0 <= 0 || P0 <= 0 || G <= 0 || C % G || (s1 && P1 <= 0)) return PMI_ERR_ARG;
  // few (sample, group) pairs and many partial rows: split the rows over S slices (each needs 16 bytes inside the group's cpg floats)
  const int cpg = C / G;
  int S = 1;
  const int64_t work = (int64_t)(P0 > P1 ? P0 : P1) * cpg;
  if (N * G < 128 && work >= 32768 && (cpg & 1) == 0) {
    S = 256 / (N * G);
    if (S > cpg / 4) S = cpg / 4;
    if (S > 64) S = 64;
    if (S < 2) S = 1;
  }
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N, G, S), dim3(256), 0, (hipStream_t)s, s0, P0, C0, s1, P1, C1, gamma, beta, film, film_ld,
                     coef_a, coef_b, HW, G, eps);
  PMI_CHECK_LAUNCH();
  if (S > 1) {
    hipLaunchKernelGGL(gn_finalize2_kernel, dim3(N, G), dim3(256), 0, (hipStream_t)s, gamma, beta, film, film_ld, coef_a, coef_b, C, HW, G, S, eps);
    PMI_CHECK_LAUNCH();
  }
  return PMI_OK;
}

extern "C" int pmi_gn_apply(const void* x, const void* x1, int C0, const float* coef_a, const float* coef_b, const void* res, void* y,
                            int N, int H, int W, int C, int act, int pool, int dtype, pmi_stream_t s) {
  if (!x1) C0 = C;
  if ((C0 & 7) || C0 <= 0 || C0 > C) return PMI_ERR_ARG;
  if (!x || !y || !coef_a || !coef_b || N <= 0 || H <= 0 || W <= 0 || (C & 7)) return PMI_ERR_ARG;
  if (pool && ((H & 1) || (W & 1))) return PMI_ERR_ARG;
  const int64_t work = (int64_t)N * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (C / 8);
  dim3 grid(grid_for(work)), block(256);
  hipStream_t st = (hipStream_t)s;
#define GO(TT, PP) hipLaunchKernelGGL((gn_apply_kernel<TT, PP>), grid, block, 0, st, (const u16*)x, (const u16*)x1, C0, coef_a, coef_b, (const u16*)res, (u16*)y, N, H, W, C, act, (u16*)nullptr)
  if (dtype == PMI_DT_BF16) { if (pool) GO(BF16, true); else GO(BF16, false); }
  else if (dtype == PMI_DT_F16X2) { if (pool) GO(F16X2, true); else GO(F16X2, false); }
  else { if (pool) GO(F16, true); else GO(F16, false); }
#undef GO
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

// the down ResBlock's two pooled tensors from one pass over x (unet.py:232-243): y = AvgPool2d(2)(act(x * a + b)) and y_raw = AvgPool2d(2)(x)
extern "C" int pmi_gn_apply_pool_skip(const void* x, const float* coef_a, const float* coef_b, void* y, void* y_raw, int N, int H, int W, int C,
                                      int act, int dtype, pmi_stream_t s) {
  if (!x || !y || !y_raw || !coef_a || !coef_b || N <= 0 || H <= 0 || W <= 0 || (C & 7) || (H & 1) || (W & 1)) return PMI_ERR_ARG;
  const int64_t work = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
  dim3 grid(grid_for(work)), block(256);
  hipStream_t st = (hipStream_t)s;
#define GO(TT) hipLaunchKernelGGL((gn_apply_kernel<TT, true, true>), grid, block, 0, st, (const u16*)x, (const u16*)nullptr, C, coef_a, coef_b, (const u16*)nullptr, (u16*)y, N, H, W, C, act, (u16*)y_raw)
  if (dtype == PMI_DT_BF16) GO(BF16);
  else if (dtype == PMI_DT_F16X2) GO(F16X2);
  else GO(F16);
#undef GO
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
