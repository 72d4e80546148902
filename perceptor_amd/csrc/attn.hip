// Flash-style self-attention for head dim 64 on gfx950 MFMA (32x32x16), one wave per 32 queries.
//
//   S^T[s][t] = sum_d K[s][d] Q[t][d]   (MFMA A = K rows, B = Q rows) -> lane = query t, regs = keys s
//   online softmax over s is lane-local (+ one exchange with lane^32)
//   O^T[d][t] += V^T[d][s] P^T[s][t]    (MFMA A = V^T rows, B = the P accumulator re-used in place)
//
// The P^T accumulator feeds the second product without any lane movement or LDS: registers
// 8ks..8ks+7 of a 32x32 accumulator are, in order, k-elements 16ks + 8(j>>2) + 4(lane>>5) + (j&3),
// so the V^T fragment is read in that same k order (two 8-byte loads per k-step).
// K, Q and V^T fragments come straight from global memory (L2-resident: T <= 1024 keys x 64 x 2 B).
#include "common.h"
#include "../../include/perceptor_hip.h"

namespace {

// qkv [N][T][3C] -> Q,K [N*heads][Tp][64], Vt [N*heads][64][Tp]; zero fill for t >= T.
template <typename T_>
__global__ __launch_bounds__(256) void qkv_split_kernel(const u16* __restrict__ qkv, u16* __restrict__ q,
                                                        u16* __restrict__ k, u16* __restrict__ vt, int T, int Tp,
                                                        int heads, int order) {
  __shared__ u16 sv[32][72];
  const int tid = threadIdx.x, t0 = blockIdx.x * 32, bh = blockIdx.y;
  const int n = bh / heads, h = bh - n * heads;
  const int C = heads * 64;
  const int qoff = order == 0 ? h * 192 : h * 64;
  const int koff = order == 0 ? h * 192 + 64 : C + h * 64;
  const int voff = order == 0 ? h * 192 + 128 : 2 * C + h * 64;
  const int row = tid >> 3, ch = tid & 7;
  const int t = t0 + row;
  uint4 vq = make_uint4(0, 0, 0, 0), vk = vq, vv = vq;
  if (t < T) {
    const u16* src = qkv + ((int64_t)n * T + t) * 3 * C + ch * 8;
    vq = *(const uint4*)(src + qoff); vk = *(const uint4*)(src + koff); vv = *(const uint4*)(src + voff);
  }
  const int64_t o = ((int64_t)bh * Tp + t) * 64 + ch * 8;
  *(uint4*)(q + o) = vq;
  *(uint4*)(k + o) = vk;
  *(uint4*)(&sv[row][ch * 8]) = vv;
  __syncthreads();
  const int d = tid >> 2, tc = tid & 3;
  u16 e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = sv[tc * 8 + j][d];
  uint4 out = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  *(uint4*)(vt + ((int64_t)bh * 64 + d) * Tp + t0 + tc * 8) = out;
}

template <typename T_>
__global__ __launch_bounds__(64) void attn_d64_kernel(const u16* __restrict__ q, const u16* __restrict__ k,
                                                      const u16* __restrict__ vt, u16* __restrict__ out, int T, int Tp,
                                                      int heads, float scale) {
  const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
  const int t0 = blockIdx.x * 32, bh = blockIdx.y;
  const u16* qb = q + ((int64_t)bh * Tp + t0 + l31) * 64 + 8 * lhi;
  const u16* kb = k + (int64_t)bh * Tp * 64 + 8 * lhi;
  const u16* vb = vt + (int64_t)bh * 64 * Tp;

  uint4 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const uint4*)(qb + kk * 16);

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  for (int s0 = 0; s0 < Tp; s0 += 32) {
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint4 kf = *(const uint4*)(kb + (int64_t)(s0 + l31) * 64 + kk * 16);
      sacc = T_::mfma32(kf, qf[kk], sacc);
    }
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s = s0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      float v = sacc[r] * scale;
      v = s < T ? v : -1e30f;
      sacc[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(sacc[r] - m_new);
      sacc[r] = p;
      rs += p;
    }
    rs += __shfl_xor(rs, 32);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float pf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = sacc[8 * ks + j];
      const uint4 pfrag = pack8<T_>(pf);
      const int sk = s0 + 16 * ks + 4 * lhi;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const u16* vp = vb + (int64_t)(db * 32 + l31) * Tp + sk;
        const uint2 lo = *(const uint2*)vp;
        const uint2 hi = *(const uint2*)(vp + 8);
        const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (db == 0) o0 = T_::mfma32(vf, pfrag, o0);
        else o1 = T_::mfma32(vf, pfrag, o1);
      }
    }
  }
  const int t = t0 + l31;
  if (t < T) {
    const float inv = 1.f / l_run;
    const int n = bh / heads, h = bh - n * heads;
    u16* ob = out + ((int64_t)n * T + t) * (heads * 64) + h * 64 + 4 * lhi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *(uint2*)(ob + 8 * g) = pack4<T_>(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      *(uint2*)(ob + 32 + 8 * g) = pack4<T_>(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
  }
}

}  // namespace

extern "C" int pmi_qkv_split(const void* qkv, void* q, void* k, void* vt, int N, int T, int heads, int order, int dtype,
                             pmi_stream_t s) {
  if (!qkv || !q || !k || !vt || N <= 0 || T <= 0 || heads <= 0 || (order != 0 && order != 1)) return PMI_ERR_ARG;
  const int Tp = (T + 31) / 32 * 32;
  dim3 grid(Tp / 32, N * heads), block(256);
  if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(qkv_split_kernel<BF16>, grid, block, 0, (hipStream_t)s, (const u16*)qkv, (u16*)q, (u16*)k, (u16*)vt, T, Tp, heads, order);
  else hipLaunchKernelGGL(qkv_split_kernel<F16>, grid, block, 0, (hipStream_t)s, (const u16*)qkv, (u16*)q, (u16*)k, (u16*)vt, T, Tp, heads, order);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

extern "C" int pmi_attn_d64(const void* q, const void* k, const void* vt, void* out, int N, int T, int heads, float scale,
                            int dtype, pmi_stream_t s) {
  if (!q || !k || !vt || !out || N <= 0 || T <= 0 || heads <= 0) return PMI_ERR_ARG;
  const int Tp = (T + 31) / 32 * 32;
  dim3 grid(Tp / 32, N * heads), block(64);
  if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(attn_d64_kernel<BF16>, grid, block, 0, (hipStream_t)s, (const u16*)q, (const u16*)k, (const u16*)vt, (u16*)out, T, Tp, heads, scale);
  else hipLaunchKernelGGL(attn_d64_kernel<F16>, grid, block, 0, (hipStream_t)s, (const u16*)q, (const u16*)k, (const u16*)vt, (u16*)out, T, Tp, heads, scale);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
