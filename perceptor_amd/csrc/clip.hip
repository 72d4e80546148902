// CLIP-guidance path kernels around the MFMA GEMMs: LayerNorm fwd / input-grad, row softmax
// fwd / bwd, 16-bit batched transpose, activation backward, separable resize (banded operator,
// forward and adjoint share one kernel), patchify / unpatchify (Normalize fused), token assembly
// and the spherical-distance loss with its gradient through F.normalize.
#include "common.h"
#include "../../include/perceptor_hip.h"

namespace {

inline int grid_for(int64_t work) {
  int64_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ---- LayerNorm forward: one wave per row, fp32 in, 16-bit and/or fp32 out -------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, u16* __restrict__ y16,
                                                            float* __restrict__ y32, float* __restrict__ mean_o,
                                                            float* __restrict__ rstd_o, int M, int D, float eps, int ld_x) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ld_x;
  float s = 0.f;
  for (int c = lane * 4; c < D; c += 256) { const float4 v = *(const float4*)(xr + c); s += v.x + v.y + v.z + v.w; }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 v = *(const float4*)(xr + c);
    const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
    q += a * a + b * b + cc * cc + d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
  if (lane == 0 && mean_o) { mean_o[row] = mean; rstd_o[row] = rstd; }
  for (int c = lane * 4; c < D; c += 256) {
    const float4 v = *(const float4*)(xr + c), g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    const float o0 = (v.x - mean) * rstd * g.x + b.x, o1 = (v.y - mean) * rstd * g.y + b.y;
    const float o2 = (v.z - mean) * rstd * g.z + b.z, o3 = (v.w - mean) * rstd * g.w + b.w;
    if (y16) *(uint2*)(y16 + (int64_t)row * D + c) = pack4<T>(o0, o1, o2, o3);
    if (y32) *(float4*)(y32 + (int64_t)row * D + c) = make_float4(o0, o1, o2, o3);
  }
}

// dx = rstd * (gy - mean(gy) - xhat * mean(gy*xhat)), gy = dy*gamma ; out = dx + gres (fp32) and 16-bit copy
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                            const float* __restrict__ rstd_i, const float* __restrict__ gres,
                                                            float* __restrict__ g32, u16* __restrict__ g16, int M, int D,
                                                            int dy_ld, int row_stride) {
  // row r of dy (stride dy_ld) corresponds to row r*row_stride of x / gres / outputs
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= M) return;
  const int64_t row = (int64_t)r * row_stride;
  const float* dyr = dy + (int64_t)r * dy_ld;
  const float* xr = x + row * D;
  const float mean = mean_i[r], rstd = rstd_i[r];
  float s1 = 0.f, s2 = 0.f;
  if ((D & 255) == 0 && D <= 2048 && (dy_ld & 3) == 0) {
    // one pass over HBM: 16-byte loads, the row (gy, xhat) stays in registers between the reduction and the write-out
    float4 gy[8], xh[8];
    const int nv = D >> 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < nv) {
        const int c = (lane + 64 * k) * 4;
        const float4 d = *(const float4*)(dyr + c), g = *(const float4*)(gamma + c), xv = *(const float4*)(xr + c);
        gy[k] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
        xh[k] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        s1 += gy[k].x + gy[k].y + gy[k].z + gy[k].w;
        s2 += gy[k].x * xh[k].x + gy[k].y * xh[k].y + gy[k].z * xh[k].z + gy[k].w * xh[k].w;
      }
    }
    s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < nv) {
        const int c = (lane + 64 * k) * 4;
        float4 v = make_float4(rstd * (gy[k].x - s1 - xh[k].x * s2), rstd * (gy[k].y - s1 - xh[k].y * s2),
                               rstd * (gy[k].z - s1 - xh[k].z * s2), rstd * (gy[k].w - s1 - xh[k].w * s2));
        if (gres) { const float4 gr = *(const float4*)(gres + row * D + c); v.x += gr.x; v.y += gr.y; v.z += gr.z; v.w += gr.w; }
        if (g32) *(float4*)(g32 + row * D + c) = v;
        if (g16) *(uint2*)(g16 + row * D + c) = pack4<T>(v.x, v.y, v.z, v.w);
      }
    }
    return;
  }
  for (int c = lane; c < D; c += 64) {
    const float gy = dyr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
    s1 += gy; s2 += gy * xh;
  }
  s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
  for (int c = lane; c < D; c += 64) {
    const float gy = dyr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
    float v = rstd * (gy - s1 - xh * s2);
    if (gres) v += gres[row * D + c];
    if (g32) g32[row * D + c] = v;
    if (g16) g16[row * D + c] = T::from_f(v);
  }
}

// ---- LayerNorm fused with the split-K reduction of the GEMM in front of it (ViT, M = 2056: the reduce pass and the LayerNorm pass were two
// launch-bound kernels over the same 8 MB).  x[row] = sum_s ws[s][row] + bias + residual[row]; one wave per row, the row stays in registers.
// a[k] = sum over the split-K slabs of this lane's k-th float4 of a row, in slab order (fixed: deterministic, and the same order as a
// serial loop).  All loads of up to four slabs are issued before the first add: the serial form (one load, one add, per value and slab)
// paid nslab x nv dependent L2 latencies per row -- 16 of them for a 1024-wide row in four slabs, most of these kernels' 15 us.
__device__ __forceinline__ void sum_slabs(const float* __restrict__ ws, int nslab, int64_t slab_stride, int64_t rowoff, int lane, int nv, float4* a) {
  float4 p[3][8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < nv) a[k] = *(const float4*)(ws + rowoff + (lane + 64 * k) * 4);
#pragma unroll
  for (int z = 1; z < 4; ++z)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < nv && z < nslab) p[z - 1][k] = *(const float4*)(ws + z * slab_stride + rowoff + (lane + 64 * k) * 4);
#pragma unroll
  for (int z = 1; z < 4; ++z)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < nv && z < nslab) { a[k].x += p[z - 1][k].x; a[k].y += p[z - 1][k].y; a[k].z += p[z - 1][k].z; a[k].w += p[z - 1][k].w; }
  for (int z = 4; z < nslab; ++z) {
    float4 q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < nv) q[k] = *(const float4*)(ws + z * slab_stride + rowoff + (lane + 64 * k) * 4);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < nv) { a[k].x += q[k].x; a[k].y += q[k].y; a[k].z += q[k].z; a[k].w += q[k].w; }
  }
}

// forward: writes x (the residual stream) and LayerNorm(x) as 16-bit GEMM operand (+ mean / rstd);  D % 256 == 0, D <= 2048.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_slabs_kernel(const float* __restrict__ ws, int nslab, int64_t slab_stride,
                                                                  const float* __restrict__ bias, const float* __restrict__ res,
                                                                  float* __restrict__ xo, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, u16* __restrict__ y16,
                                                                  float* __restrict__ mean_o, float* __restrict__ rstd_o, int M, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nv = D >> 8;
  float4 v[8];
  float s = 0.f;
  sum_slabs(ws, nslab, slab_stride, (int64_t)row * D, lane, nv, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < nv) {
      const int c = (lane + 64 * k) * 4;
      float4 a = v[k];
      if (bias) { const float4 b = *(const float4*)(bias + c); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
      if (res) { const float4 r = *(const float4*)(res + (int64_t)row * D + c); a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
      *(float4*)(xo + (int64_t)row * D + c) = a;
      v[k] = a;
      s += a.x + a.y + a.z + a.w;
    }
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < nv) {
      const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
  if (lane == 0 && mean_o) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < nv) {
      const int c = (lane + 64 * k) * 4;
      const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
      *(uint2*)(y16 + (int64_t)row * D + c) = pack4<T>((v[k].x - mean) * rstd * g.x + b.x, (v[k].y - mean) * rstd * g.y + b.y,
                                                        (v[k].z - mean) * rstd * g.z + b.z, (v[k].w - mean) * rstd * g.w + b.w);
    }
  }
}

// backward: dy[row] = sum_s ws[s][row] (the input-gradient GEMM's split-K slabs), then layernorm_bwd_kernel's single-pass form
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_slabs_kernel(const float* __restrict__ ws, int nslab, int64_t slab_stride,
                                                                  const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                                  const float* __restrict__ gres, float* __restrict__ g32,
                                                                  u16* __restrict__ g16, int M, int D) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= M) return;
  const float mean = mean_i[r], rstd = rstd_i[r];
  const int nv = D >> 8;
  float4 gy[8], xh[8];
  float s1 = 0.f, s2 = 0.f;
  sum_slabs(ws, nslab, slab_stride, (int64_t)r * D, lane, nv, gy);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < nv) {
      const int c = (lane + 64 * k) * 4;
      const float4 d = gy[k];
      const float4 g = *(const float4*)(gamma + c), xv = *(const float4*)(x + (int64_t)r * D + c);
      gy[k] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
      xh[k] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      s1 += gy[k].x + gy[k].y + gy[k].z + gy[k].w;
      s2 += gy[k].x * xh[k].x + gy[k].y * xh[k].y + gy[k].z * xh[k].z + gy[k].w * xh[k].w;
    }
  }
  s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < nv) {
      const int c = (lane + 64 * k) * 4;
      float4 v = make_float4(rstd * (gy[k].x - s1 - xh[k].x * s2), rstd * (gy[k].y - s1 - xh[k].y * s2),
                             rstd * (gy[k].z - s1 - xh[k].z * s2), rstd * (gy[k].w - s1 - xh[k].w * s2));
      if (gres) { const float4 gr = *(const float4*)(gres + (int64_t)r * D + c); v.x += gr.x; v.y += gr.y; v.z += gr.z; v.w += gr.w; }
      if (g32) *(float4*)(g32 + (int64_t)r * D + c) = v;
      if (g16) *(uint2*)(g16 + (int64_t)r * D + c) = pack4<T>(v.x, v.y, v.z, v.w);
    }
  }
}

// ---- row softmax over the first T columns of fp32 scores; 16-bit probabilities, zero padding -----
template <typename TT>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, u16* __restrict__ P, int rows, int T,
                                                          int ld_in, int ld_out, float scale, int causal_t) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = S + (int64_t)row * ld_in;
  if (causal_t > 0) T = min(T, row % causal_t + 1);   // query i of a causal_t-token sequence sees keys 0..i
  float mx = -1e30f;
  for (int c = lane; c < T; c += 64) mx = fmaxf(mx, sr[c] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < T; c += 64) sum += __expf(sr[c] * scale - mx);
  const float inv = 1.f / wave_sum(sum);
  for (int c = lane; c < ld_out; c += 64)
    P[(int64_t)row * ld_out + c] = c < T ? TT::from_f(__expf(sr[c] * scale - mx) * inv) : (u16)0;
}

// dS = scale * P * (dP - sum_s dP*P)
template <typename TT>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dP, const u16* __restrict__ P,
                                                          u16* __restrict__ dS, int rows, int T, int ld_dp, int ld_p, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* dr = dP + (int64_t)row * ld_dp;
  const u16* pr = P + (int64_t)row * ld_p;
  float dot = 0.f;
  for (int c = lane; c < T; c += 64) dot += dr[c] * TT::to_f(pr[c]);
  dot = wave_sum(dot);
  for (int c = lane; c < ld_p; c += 64)
    dS[(int64_t)row * ld_p + c] = c < T ? TT::from_f(scale * TT::to_f(pr[c]) * (dr[c] - dot)) : (u16)0;
}

// ---- batched transpose of 16-bit matrices: in[b][R][Cc] (strided) -> out[b][Cc][Rp], zero padded ----
__global__ __launch_bounds__(256) void transpose16_kernel(const u16* __restrict__ in, u16* __restrict__ out, int R, int Cc,
                                                          int ld_in, int64_t sI_o, int64_t sI_i, int batch_inner, int Rp) {
  __shared__ u16 tile[32][34];
  const int b = blockIdx.z, zo = b / batch_inner, zi = b - zo * batch_inner;
  const u16* src = in + zo * sI_o + zi * sI_i;
  u16* dst = out + (int64_t)b * Cc * Rp;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? src[(int64_t)r * ld_in + c] : (u16)0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < Rp) dst[(int64_t)c * Rp + r] = tile[tx][i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const u16* __restrict__ dh, const u16* __restrict__ hpre,
                                                      u16* __restrict__ out, int64_t n8, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float a[8], b[8];
    unpack8<T>(*(const uint4*)(dh + i * 8), a);
    unpack8<T>(*(const uint4*)(hpre + i * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= act_grad(b[e], act);
    *(uint4*)(out + i * 8) = pack8<T>(a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void act_fwd_kernel(const u16* __restrict__ in, u16* __restrict__ out, int64_t n8, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float a[8];
    unpack8<T>(*(const uint4*)(in + i * 8), a);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = act_apply(a[e], act);
    *(uint4*)(out + i * 8) = pack8<T>(a);
  }
}

// F.normalize(x, dim=1): y = x / max(|x|, 1e-12), one wave per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int M, int D, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = x[(int64_t)row * D + c]; s += v * v; }
  const float inv = scale / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int c = lane; c < D; c += 64) y[(int64_t)row * D + c] = x[(int64_t)row * D + c] * inv;
}

// ---- banded 1-D linear operator along the middle axis of [outer][in_sz][inner] ------------------
__global__ __launch_bounds__(256) void resize_apply_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           const int* __restrict__ idx, const float* __restrict__ w,
                                                           int outer, int in_sz, int inner, int out_sz, int taps) {
  const int64_t total = (int64_t)outer * out_sz * inner;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ii = (int)(i % inner);
    const int64_t t = i / inner;
    const int j = (int)(t % out_sz);
    const int64_t o = t / out_sz;
    const float* base = in + o * in_sz * inner + ii;
    float acc = 0.f;
    for (int k = 0; k < taps; ++k) {
      const int r = idx[j * taps + k];
      if (r >= 0) acc += w[j * taps + k] * base[(int64_t)r * inner];
    }
    out[i] = acc;
  }
}

// resized fp32 NCHW [N][3][R][R] -> im2col [N*g*g][Kp] 16-bit, k = c*P*P + py*P + px, (x-mean)/std fused
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, const float* __restrict__ mean,
                                                       const float* __restrict__ stdv, u16* __restrict__ col, int N, int R,
                                                       int P, int Kp) {
  const int g = R / P, K = 3 * P * P;
  const int64_t total = (int64_t)N * g * g * Kp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % Kp);
    const int64_t m = i / Kp;
    float v = 0.f;
    if (k < K) {
      const int c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
      const int n = (int)(m / (g * g)), pp = (int)(m - (int64_t)n * g * g), gy = pp / g, gx = pp - gy * g;
      v = (img[(((int64_t)n * 3 + c) * R + gy * P + py) * R + gx * P + px] - mean[c]) / stdv[c];
    }
    col[i] = T::from_f(v);
  }
}
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ dcol, const float* __restrict__ stdv,
                                                         float* __restrict__ dimg, int N, int R, int P, int Kp, float mul) {
  const int g = R / P;
  const int64_t total = (int64_t)N * 3 * R * R;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % R), y = (int)((i / R) % R), c = (int)((i / ((int64_t)R * R)) % 3), n = (int)(i / ((int64_t)3 * R * R));
    const int gy = y / P, py = y - gy * P, gx = x / P, px = x - gx * P;
    const int64_t m = (int64_t)n * g * g + gy * g + gx;
    dimg[i] = dcol[m * Kp + c * P * P + py * P + px] / stdv[c] * mul;
  }
}

// x[n][0] = cls + pos[0]; x[n][1+p] = emb[n][p] + pos[1+p]  (fp32)
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ emb, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, float* __restrict__ x, int N, int T, int D) {
  const int64_t total = (int64_t)N * T * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D), t = (int)((i / D) % T), n = (int)(i / ((int64_t)T * D));
    x[i] = (t == 0 ? cls[d] : emb[((int64_t)n * (T - 1) + t - 1) * D + d]) + pos[(int64_t)t * D + d];
  }
}

// loss = mult * mean_{n,k} w_k * 2 asin(|e_n - t_k| / 2)^2, e = emb/|emb| ; demb = gradient wrt emb (x gscale)
__global__ __launch_bounds__(256) void spherical_loss_kernel(const float* __restrict__ emb, const float* __restrict__ tgt,
                                                             const float* __restrict__ wts, float* __restrict__ loss,
                                                             float* __restrict__ demb, int N, int K, int D, float mult,
                                                             float gscale, float inv_count) {
  __shared__ float red[4];
  __shared__ float e_s[2048], g_s[2048];
  // ONE workgroup walks the samples in order: the scalar loss is a fixed-order sum (a global float atomic per sample would
  // make its last bits depend on timing); N x K x D is a few thousand operations.
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float ltot = 0.f;
  for (int n = 0; n < N; ++n) {
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  float s = 0.f;
  for (int d = tid; d < D; d += 256) { const float v = emb[(int64_t)n * D + d]; s += v * v; }
  const float nrm = fmaxf(sqrtf(block_sum(s)), 1e-12f);
  for (int d = tid; d < D; d += 256) { e_s[d] = emb[(int64_t)n * D + d] / nrm; g_s[d] = 0.f; }
  __syncthreads();
  float lsum = 0.f;
  for (int k = 0; k < K; ++k) {
    float q = 0.f;
    for (int d = tid; d < D; d += 256) { const float df = e_s[d] - tgt[(int64_t)k * D + d]; q += df * df; }
    const float u = sqrtf(block_sum(q));
    const float hs = fminf(u * 0.5f, 1.f);
    const float as = asinf(hs);
    lsum += wts[k] * 2.f * as * as;
    // d/du [2 asin(u/2)^2] = 2 asin(u/2) / sqrt(1 - u^2/4); times (e - t)/u
    const float coef = u > 1e-12f ? wts[k] * 2.f * as / (sqrtf(fmaxf(1.f - hs * hs, 1e-12f)) * u) : 0.f;
    for (int d = tid; d < D; d += 256) g_s[d] += coef * (e_s[d] - tgt[(int64_t)k * D + d]);
    __syncthreads();
  }
  float dot = 0.f;
  for (int d = tid; d < D; d += 256) dot += e_s[d] * g_s[d];
  dot = block_sum(dot);
  const float c = mult * inv_count * gscale / nrm;
  for (int d = tid; d < D; d += 256) demb[(int64_t)n * D + d] = c * (g_s[d] - e_s[d] * dot);
  ltot += lsum;
  __syncthreads();
  }
  if (tid == 0) *loss = ltot * mult * inv_count;
}

// ---- CLIP text tower: x[n][t][:] = token_embedding[ids[n][t]] + positional_embedding[t] ------------
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                           const float* __restrict__ pos, float* __restrict__ x, int64_t total, int T, int D, int vocab) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    const int64_t r = i / D;
    const int t = (int)(r % T);
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);      // the host validates ids; never read out of the table
    x[i] = tok[id * D + d] + (pos ? pos[(int64_t)t * D + d] : 0.f);
  }
}

// dst[r][:] = src[idx[r]][:] (fp32 rows)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ dst,
                                                          int R, int D, int ld, int64_t src_rows) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)R * D; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / D), d = (int)(i % D);
    int64_t j = idx[r];
    j = j < 0 ? 0 : (j >= src_rows ? src_rows - 1 : j);
    dst[i] = src[j * ld + d];
  }
}

}  // namespace

#define ST ((hipStream_t)s)
#define BY_DTYPE(KERN, ...)                                                                    \
  do {                                                                                         \
    if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(KERN<BF16>, grid, block, 0, ST, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERN<F16>, grid, block, 0, ST, __VA_ARGS__);                       \
  } while (0)

extern "C" int pmi_layernorm_fwd(const float* x, int ld_x, const float* gamma, const float* beta, void* y16, float* y32, float* mean_rstd,
                                 int M, int D, float eps, int dtype, pmi_stream_t s) {
  if (!x || !gamma || !beta || (!y16 && !y32) || M <= 0 || D <= 0 || (D & 3) || ld_x < D || (ld_x & 3)) return PMI_ERR_ARG;
  dim3 grid((M + 3) / 4), block(256);
  float* mo = mean_rstd; float* ro = mean_rstd ? mean_rstd + M : nullptr;
  BY_DTYPE(layernorm_fwd_kernel, x, gamma, beta, (u16*)y16, y32, mo, ro, M, D, eps, ld_x);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean_rstd, const float* gres,
                                 float* g32, void* g16, int M, int D, int dy_ld, int row_stride, int dtype, pmi_stream_t s) {
  if (!dy || !x || !gamma || !mean_rstd || (!g32 && !g16) || M <= 0 || D <= 0) return PMI_ERR_ARG;
  dim3 grid((M + 3) / 4), block(256);
  BY_DTYPE(layernorm_bwd_kernel, dy, x, gamma, mean_rstd, mean_rstd + M, gres, g32, (u16*)g16, M, D, dy_ld, row_stride);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_layernorm_fwd_slabs(const float* ws, int nslab, int64_t slab_stride, const float* bias, const float* residual, float* x_out,
                                       const float* gamma, const float* beta, void* y16, float* mean_rstd, int M, int D, float eps, int dtype,
                                       pmi_stream_t s) {
  if (!ws || !x_out || !gamma || !beta || !y16 || nslab < 1 || M <= 0 || D <= 0 || (D & 255) || D > 2048) return PMI_ERR_ARG;
  dim3 grid((M + 3) / 4), block(256);
  float* mo = mean_rstd; float* ro = mean_rstd ? mean_rstd + M : nullptr;
  BY_DTYPE(layernorm_fwd_slabs_kernel, ws, nslab, slab_stride, bias, residual, x_out, gamma, beta, (u16*)y16, mo, ro, M, D, eps);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_layernorm_bwd_slabs(const float* ws, int nslab, int64_t slab_stride, const float* x, const float* gamma, const float* mean_rstd,
                                       const float* gres, float* g32, void* g16, int M, int D, int dtype, pmi_stream_t s) {
  if (!ws || !x || !gamma || !mean_rstd || (!g32 && !g16) || nslab < 1 || M <= 0 || D <= 0 || (D & 255) || D > 2048) return PMI_ERR_ARG;
  dim3 grid((M + 3) / 4), block(256);
  BY_DTYPE(layernorm_bwd_slabs_kernel, ws, nslab, slab_stride, x, gamma, mean_rstd, mean_rstd + M, gres, g32, (u16*)g16, M, D);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_softmax_fwd(const float* S, void* P, int rows, int T, int ld_in, int ld_out, float scale, int dtype, pmi_stream_t s) {
  if (!S || !P || rows <= 0 || T <= 0 || ld_in < T || ld_out < T) return PMI_ERR_ARG;
  dim3 grid((rows + 3) / 4), block(256);
  BY_DTYPE(softmax_fwd_kernel, S, (u16*)P, rows, T, ld_in, ld_out, scale, 0);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_softmax_causal_fwd(const float* S, void* P, int rows, int T, int ld_in, int ld_out, float scale, int dtype, pmi_stream_t s) {
  if (!S || !P || rows <= 0 || T <= 0 || rows % T || ld_in < T || ld_out < T) return PMI_ERR_ARG;
  dim3 grid((rows + 3) / 4), block(256);
  BY_DTYPE(softmax_fwd_kernel, S, (u16*)P, rows, T, ld_in, ld_out, scale, T);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_embed_tokens(const int64_t* ids, const float* tok, const float* pos, float* x, int N, int T, int D, int vocab, pmi_stream_t s) {
  if (!ids || !tok || !x || N <= 0 || T <= 0 || D <= 0 || vocab <= 0) return PMI_ERR_ARG;
  const int64_t total = (int64_t)N * T * D;
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(grid_for(total)), dim3(256), 0, ST, ids, tok, pos, x, total, T, D, vocab);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_gather_rows(const float* src, const int64_t* idx, float* dst, int R, int D, int ld, int64_t src_rows, pmi_stream_t s) {
  if (!src || !idx || !dst || R <= 0 || D <= 0 || ld < D || src_rows <= 0) return PMI_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)R * D)), dim3(256), 0, ST, src, idx, dst, R, D, ld, src_rows);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_softmax_bwd(const float* dP, const void* P, void* dS, int rows, int T, int ld_dp, int ld_p, float scale, int dtype, pmi_stream_t s) {
  if (!dP || !P || !dS || rows <= 0 || T <= 0 || ld_dp < T || ld_p < T) return PMI_ERR_ARG;
  dim3 grid((rows + 3) / 4), block(256);
  BY_DTYPE(softmax_bwd_kernel, dP, (const u16*)P, (u16*)dS, rows, T, ld_dp, ld_p, scale);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_transpose_16(const void* in, void* out, int R, int Cc, int ld_in, int64_t sI_o, int64_t sI_i, int batch_inner,
                                int batch, pmi_stream_t s) {
  if (!in || !out || R <= 0 || Cc <= 0 || batch <= 0 || batch_inner <= 0) return PMI_ERR_ARG;
  const int Rp = (R + 7) / 8 * 8;
  dim3 grid((Cc + 31) / 32, (Rp + 31) / 32, batch), block(256);
  hipLaunchKernelGGL(transpose16_kernel, grid, block, 0, ST, (const u16*)in, (u16*)out, R, Cc, ld_in, sI_o, sI_i, batch_inner, Rp);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_act_bwd(const void* dh, const void* hpre, void* out, int64_t n, int act, int dtype, pmi_stream_t s) {
  if (!dh || !hpre || !out || n <= 0 || (n & 7)) return PMI_ERR_ARG;
  dim3 grid(grid_for(n / 8)), block(256);
  BY_DTYPE(act_bwd_kernel, (const u16*)dh, (const u16*)hpre, (u16*)out, n / 8, act);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_resize_apply(const float* in, float* out, const int* idx, const float* w, int outer, int in_sz, int inner,
                                int out_sz, int taps, int r0, int r1, pmi_stream_t s) {
  (void)r0; (void)r1;
  if (!in || !out || !idx || !w || outer <= 0 || in_sz <= 0 || inner <= 0 || out_sz <= 0 || taps <= 0) return PMI_ERR_ARG;
  hipLaunchKernelGGL(resize_apply_kernel, dim3(grid_for((int64_t)outer * out_sz * inner)), dim3(256), 0, ST, in, out, idx, w,
                     outer, in_sz, inner, out_sz, taps);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_patchify(const float* img, const float* mean, const float* stdv, void* col, int N, int R, int P, int Kp,
                            int r0, int dtype, pmi_stream_t s) {
  (void)r0;
  if (!img || !mean || !stdv || !col || N <= 0 || R <= 0 || P <= 0 || R % P || Kp < 3 * P * P || (Kp & 7)) return PMI_ERR_ARG;
  dim3 grid(grid_for((int64_t)N * (R / P) * (R / P) * Kp)), block(256);
  BY_DTYPE(patchify_kernel, img, mean, stdv, (u16*)col, N, R, P, Kp);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_unpatchify(const float* dcol, const float* stdv, float* dimg, int N, int R, int P, int Kp, float mul, pmi_stream_t s) {
  if (!dcol || !stdv || !dimg || N <= 0 || R % P) return PMI_ERR_ARG;
  hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for((int64_t)N * 3 * R * R)), dim3(256), 0, ST, dcol, stdv, dimg, N, R, P, Kp, mul);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_vit_assemble(const float* emb, const float* cls, const float* pos, float* x, int N, int T, int D, int r0, pmi_stream_t s) {
  (void)r0;
  if (!emb || !cls || !pos || !x || N <= 0 || T <= 1 || D <= 0) return PMI_ERR_ARG;
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for((int64_t)N * T * D)), dim3(256), 0, ST, emb, cls, pos, x, N, T, D);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_spherical_loss(const float* emb, const float* tgt, const float* wts, float* loss, float* demb, int N, int K,
                                  int D, int n_total, float mult, float gscale, pmi_stream_t s) {
  if (!emb || !tgt || !wts || !loss || !demb || N <= 0 || K <= 0 || D <= 0 || D > 2048 || n_total < N) return PMI_ERR_ARG;
  hipLaunchKernelGGL(spherical_loss_kernel, dim3(1), dim3(256), 0, ST, emb, tgt, wts, loss, demb, N, K, D, mult, gscale,
                     1.0f / ((float)n_total * (float)K));
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_act_fwd(const void* in, void* out, int64_t n, int act, int dtype, pmi_stream_t s) {
  if (!in || !out || n <= 0 || (n & 7)) return PMI_ERR_ARG;
  dim3 grid(grid_for(n / 8)), block(256);
  BY_DTYPE(act_fwd_kernel, (const u16*)in, (u16*)out, n / 8, act);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
extern "C" int pmi_l2norm_rows(const float* x, float* y, int M, int D, float scale, pmi_stream_t s) {
  if (!x || !y || M <= 0 || D <= 0) return PMI_ERR_ARG;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, ST, x, y, M, D, scale);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
