// Exact-fp32 batched GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, 1/16 of the
// 16-bit MFMA rate) and an in-place fp32 row softmax.  Used only where "precise" mode (dtype 2) needs full fp32 products of
// two ACTIVATION operands -- the UNet's attention scores / values (0.3 % of its FLOPs) -- and for the tiny fp32 time MLPs;
// everything weight-shaped runs on the f16 MFMA with hi + lo activation pairs (common.h: F16X2).
//   D[b][m][n] = act(alpha * sum_k A[b][m][k] * B[b][n][k] (or B[b][k][n] when transB) + bias[n]) + R[b][m][n]
// Workgroup tile 64 x 64, 4 waves x one 32 x 32 block, K staged 32 at a time through LDS (rows padded to 33 floats).
#include "common.h"
#include "../../include/perceptor_hip.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 32, LDP = TK + 1;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const pmi_gemm_f32_args a) {
  __shared__ float As[TM * LDP], Bs[TN * LDP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int zo = blockIdx.z / a.batch_inner, zi = blockIdx.z % a.batch_inner;
  const float* A = a.A + zo * a.sA_o + zi * a.sA_i;
  const float* B = a.B + zo * a.sB_o + zi * a.sB_i;
  float* D = a.D + zo * a.sD_o + zi * a.sD_i;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < a.K; k0 += TK) {
    // stage: 64 rows x 32 k of each operand, k fastest across threads (coalesced for row-major A and non-transposed B)
    for (int e = tid; e < TM * TK; e += 256) {
      const int r = e / TK, k = e - r * TK;
      const int m = m0 + r, kk = k0 + k;
      As[r * LDP + k] = (m < a.M && kk < a.K) ? A[(int64_t)m * a.lda + kk] : 0.f;
    }
    if (a.transB) {       // B[k][n]: n fastest across threads
      for (int e = tid; e < TN * TK; e += 256) {
        const int k = e / TN, c = e - k * TN;
        const int n = n0 + c, kk = k0 + k;
        Bs[c * LDP + k] = (n < a.N && kk < a.K) ? B[(int64_t)kk * a.ldb + n] : 0.f;
      }
    } else {
      for (int e = tid; e < TN * TK; e += 256) {
        const int c = e / TK, k = e - c * TK;
        const int n = n0 + c, kk = k0 + k;
        Bs[c * LDP + k] = (n < a.N && kk < a.K) ? B[(int64_t)n * a.ldb + kk] : 0.f;
      }
    }
    __syncthreads();
    const float* ap = As + (wr * 32 + (lane & 31)) * LDP + (lane >> 5);
    const float* bp = Bs + (wc * 32 + (lane & 31)) * LDP + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < TK; kk += 2)     // swapped product (A operand = B matrix rows): a lane ends up with 4 consecutive n of one m
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bp[kk], ap[kk], acc, 0, 0, 0);
    __syncthreads();
  }
  const int m = m0 + wr * 32 + (lane & 31);
  if (m >= a.M) return;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + wc * 32 + 8 * g + 4 * (lane >> 5) + e;
      if (n >= a.N) continue;
      float v = acc[4 * g + e] * a.alpha;
      if (a.bias) v += a.bias[n];
      if (a.act != PMI_ACT_NONE) v = act_apply(v, a.act);
      if (a.R) v += a.R[zo * a.sD_o + zi * a.sD_i + (int64_t)m * a.ldd + n];      // residual: D's layout
      D[(int64_t)m * a.ldd + n] = v;
    }
}

// in place: S[row][0..T) <- softmax(scale * S[row][0..T)); one wave per row
__global__ __launch_bounds__(256) void softmax_f32_kernel(float* __restrict__ S, int rows, int T, int ld, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* p = S + (int64_t)row * ld;
  float mx = -3.0e38f;
  for (int i = lane; i < T; i += 64) mx = fmaxf(mx, p[i] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < T; i += 64) { const float e = expf(p[i] * scale - mx); p[i] = e; sum += e; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int i = lane; i < T; i += 64) p[i] *= inv;
}

}  // namespace

extern "C" int pmi_gemm_f32(const pmi_gemm_f32_args* a, pmi_stream_t s) {
  if (!a || !a->A || !a->B || !a->D || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0 || a->batch_inner <= 0) return PMI_ERR_ARG;
  dim3 grid((a->N + TN - 1) / TN, (a->M + TM - 1) / TM, a->batch);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)s, *a);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

extern "C" int pmi_softmax_f32(float* S, int rows, int T, int ld, float scale, pmi_stream_t s) {
  if (!S || rows <= 0 || T <= 0 || ld < T) return PMI_ERR_ARG;
  hipLaunchKernelGGL(softmax_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, S, rows, T, ld, scale);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
