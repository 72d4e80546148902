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

// Transposed operands (head dim on the MFMA rows, tokens on k) are stored in FRAGMENT order: the 8 tokens a lane needs for one
// MFMA A operand, {16 ks + 4 lhi + 0..3} and {16 ks + 8 + 4 lhi + 0..3} of a 32-token block (the k order of an accumulator tile
// re-used as B operand), are 16 contiguous bytes and the 64 lanes of a wave are contiguous: one coalesced 1 KB load per fragment.
// With plain [bh][64][Tp] rows every fragment load touched 64 separate 8-byte pieces 2*Tp bytes apart.
__device__ __forceinline__ int64_t tfrag(int bh, int ntb, int tb, int ks, int db, int lhi, int l31) {
  return ((((((int64_t)bh * ntb + tb) * 2 + ks) * 2 + db) * 2 + lhi) * 32 + l31) * 8;
}

// The row operands (tokens on the MFMA rows, head dim on k: Q, K, V, dO) use the same idea: element (t, c) of a 32-token block
// lives at [block][kk = c / 16][lhi = (c / 8) & 1][t & 31][c & 7], so the fragment of step kk is one contiguous 1 KB run.
__device__ __forceinline__ int64_t rfrag(int bh, int ntb, int tb, int kk, int lhi, int l31) {
  return ((((((int64_t)bh * ntb + tb) * 4 + kk) * 2 + lhi) * 32 + l31) * 8);
}

// qkv [N][T][3C] -> Q,K [N*heads][Tp/32][fragment order, rfrag], Vt [N*heads][Tp/32][fragment order, tfrag]; zero fill for t >= T.
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
  const int64_t o = rfrag(bh, Tp >> 5, blockIdx.x, ch >> 1, ch & 1, row);
  *(uint4*)(q + o) = vq;
  *(uint4*)(k + o) = vk;
  *(uint4*)(&sv[row][ch * 8]) = vv;
  __syncthreads();
  const int d = tid >> 2, tc = tid & 3, fks = tc >> 1, flhi = tc & 1;
  u16 e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = sv[16 * fks + 4 * flhi + (j & 3) + 8 * (j >> 2)][d];
  uint4 out = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  *(uint4*)(vt + tfrag(bh, Tp >> 5, blockIdx.x, fks, d >> 5, flhi, d & 31)) = out;
}

// Workgroup -> (row block, head) with all row blocks of a head on ONE XCD (round-robin dispatch puts consecutive linear ids on different
// XCDs; each XCD has its own L2, so without this every XCD fetches every head's K/V: FETCH_SIZE of vit_attn_bwd was 8x its operands).
__device__ __forceinline__ void head_xcd_remap(int& bx, int& bh) {
  const int nx = gridDim.x;
  const int lin = xcd_remap(blockIdx.x + nx * blockIdx.y, nx * gridDim.y);
  bh = lin / nx;
  bx = lin - bh * nx;
}

template <typename T_>
__global__ __launch_bounds__(64) void attn_d64_kernel(const u16* __restrict__ q, const u16* __restrict__ k,
                                                      const u16* __restrict__ vt, u16* __restrict__ out, int T, int Tp,
                                                      int heads, float scale) {
  const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
  int bx, bh;
  head_xcd_remap(bx, bh);
  const int t0 = bx * 32;
  const int ntb = Tp >> 5;
  uint4 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const uint4*)(q + rfrag(bh, ntb, bx, kk, lhi, l31));

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
      const uint4 kf = *(const uint4*)(k + rfrag(bh, ntb, s0 >> 5, kk, lhi, l31));
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
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const uint4 vf = *(const uint4*)(vt + tfrag(bh, Tp >> 5, s0 >> 5, ks, db, lhi, l31));
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

// =====================================================================================================================
// ViT attention with backward (head dim 64, any T): forward saves O and the log-sum-exp; backward is two kernels that
// each recompute the 32x32 probability blocks on MFMA (no T x T matrix in HBM, no transposes, no atomics):
//   dq kernel   : one wave per 32-query tile, loops over key tiles    -> dQ
//   dkdv kernel : one wave per 32-key tile,   loops over query tiles  -> dK, dV
// Layouts (all 16-bit, Tp = T rounded up to 32, zero filled): Q,K,V,dO [B*H][Tp/32][fragment order, see rfrag]; Qt,Kt,Vt,dOt [B*H][Tp/32][fragment order, see tfrag];
// lse, delta fp32 [B*H][Tp]  (delta[t] = sum_d dO[t][d] * O[t][d]).
// Replaces nn.MultiheadAttention forward + autograd in the CLIP tower (ruclip/model.py:40-52).
// =====================================================================================================================
namespace {

// qkv [N][T][3C] with channels (which, head, d) -> Q,K,V [bh][Tp][64] and Qt,Kt,Vt in fragment order (tfrag)
template <typename T_>
__global__ __launch_bounds__(256) void vit_qkv_split_kernel(const u16* __restrict__ qkv, u16* __restrict__ q, u16* __restrict__ k,
                                                            u16* __restrict__ v, u16* __restrict__ qt, u16* __restrict__ kt,
                                                            u16* __restrict__ vt, int T, int Tp, int heads) {
  __shared__ u16 s[3][32][72];
  const int tid = threadIdx.x, t0 = blockIdx.x * 32, bh = blockIdx.y;
  const int n = bh / heads, h = bh - n * heads, C = heads * 64;
  const int row = tid >> 3, ch = tid & 7, t = t0 + row;
  u16* dst[3] = {q, k, v};
  u16* dstt[3] = {qt, kt, vt};
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    uint4 val = make_uint4(0, 0, 0, 0);
    if (t < T) val = *(const uint4*)(qkv + ((int64_t)n * T + t) * 3 * C + w * C + h * 64 + ch * 8);
    *(uint4*)(dst[w] + rfrag(bh, Tp >> 5, blockIdx.x, ch >> 1, ch & 1, row)) = val;
    *(uint4*)(&s[w][row][ch * 8]) = val;
  }
  __syncthreads();
  const int d = tid >> 2, tc = tid & 3, fks = tc >> 1, flhi = tc & 1;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    u16 e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = s[w][16 * fks + 4 * flhi + (j & 3) + 8 * (j >> 2)][d];
    *(uint4*)(dstt[w] + tfrag(bh, Tp >> 5, blockIdx.x, fks, d >> 5, flhi, d & 31)) =
        make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  }
}

// dO [N][T][C] (head, d) -> dO [bh][Tp][64], dOt [bh][64][Tp], delta[bh][Tp] = sum_d dO*O   (O in [N][T][C])
template <typename T_>
__global__ __launch_bounds__(256) void vit_do_prep_kernel(const u16* __restrict__ dout, const u16* __restrict__ o,
                                                          u16* __restrict__ d_o, u16* __restrict__ d_ot, float* __restrict__ delta,
                                                          int T, int Tp, int heads) {
  __shared__ u16 s[32][72];
  const int tid = threadIdx.x, t0 = blockIdx.x * 32, bh = blockIdx.y;
  const int n = bh / heads, h = bh - n * heads, C = heads * 64;
  const int row = tid >> 3, ch = tid & 7, t = t0 + row;
  uint4 val = make_uint4(0, 0, 0, 0), ov = val;
  if (t < T) {
    val = *(const uint4*)(dout + ((int64_t)n * T + t) * C + h * 64 + ch * 8);
    ov = *(const uint4*)(o + ((int64_t)n * T + t) * C + h * 64 + ch * 8);
  }
  *(uint4*)(d_o + rfrag(bh, Tp >> 5, blockIdx.x, ch >> 1, ch & 1, row)) = val;
  *(uint4*)(&s[row][ch * 8]) = val;
  float a[8], b[8], p = 0.f;
  unpack8<T_>(val, a); unpack8<T_>(ov, b);
#pragma unroll
  for (int e = 0; e < 8; ++e) p += a[e] * b[e];
  p += __shfl_xor(p, 1); p += __shfl_xor(p, 2); p += __shfl_xor(p, 4);
  if (ch == 0) delta[(int64_t)bh * Tp + t] = p;
  __syncthreads();
  const int d = tid >> 2, tc = tid & 3, fks = tc >> 1, flhi = tc & 1;
  u16 e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = s[16 * fks + 4 * flhi + (j & 3) + 8 * (j >> 2)][d];
  *(uint4*)(d_ot + tfrag(bh, Tp >> 5, blockIdx.x, fks, d >> 5, flhi, d & 31)) =
      make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}

// forward: as attn_d64_kernel, plus lse[bh][t] = m + log(l)
template <typename T_>
__global__ __launch_bounds__(64) void vit_attn_fwd_kernel(const u16* __restrict__ q, const u16* __restrict__ k, const u16* __restrict__ vt,
                                                          u16* __restrict__ out, float* __restrict__ lse, int T, int Tp, int heads, float scale) {
  const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
  int bx, bh;
  head_xcd_remap(bx, bh);
  const int t0 = bx * 32;
  const int ntb = Tp >> 5;
  uint4 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const uint4*)(q + rfrag(bh, ntb, bx, kk, lhi, l31));
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  struct Tile { uint4 k[4], vt[2][2]; };                 // a key tile's fragments, fetched one tile ahead (see vit_attn_dkdv_body)
  auto load_tile = [&](int sb, Tile& f) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f.k[kk] = *(const uint4*)(k + rfrag(bh, ntb, sb, kk, lhi, l31));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < 2; ++db) f.vt[ks][db] = *(const uint4*)(vt + tfrag(bh, ntb, sb, ks, db, lhi, l31));
  };
  auto compute = [&](const Tile& f, int s0) {
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) sacc = T_::mfma32(f.k[kk], qf[kk], sacc);
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s = s0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float v = s < T ? sacc[r] * scale : -1e30f;
      sacc[r] = v; mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx), alpha = __expf(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float p = __expf(sacc[r] - m_new); sacc[r] = p; rs += p; }
    rs += __shfl_xor(rs, 32);
    l_run = l_run * alpha + rs; m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float pf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = sacc[8 * ks + j];
      const uint4 pfrag = pack8<T_>(pf);
      o0 = T_::mfma32(f.vt[ks][0], pfrag, o0);
      o1 = T_::mfma32(f.vt[ks][1], pfrag, o1);
    }
  };
  Tile ta, tb_;
  load_tile(0, ta);
  for (int sb = 0; sb < ntb; sb += 2) {
    load_tile(sb + 1 < ntb ? sb + 1 : sb, tb_);
    __builtin_amdgcn_sched_barrier(0);
    compute(ta, sb * 32);
    __builtin_amdgcn_sched_barrier(0);
    if (sb + 1 < ntb) {
      load_tile(sb + 2 < ntb ? sb + 2 : sb + 1, ta);
      __builtin_amdgcn_sched_barrier(0);
      compute(tb_, (sb + 1) * 32);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int t = t0 + l31;
  if (lhi == 0) lse[(int64_t)bh * Tp + t] = t < T ? m_run + __logf(l_run) : 0.f;
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

// dQ: wave per 32-query tile.  S^T = K.Q^T and dP^T = V.dO^T put the query on the lane, so lse/delta are lane-local;
// dS^T (accumulator) is the B operand of dQ^T += K^T . dS^T without any movement.
template <typename T_>
__device__ __forceinline__ void vit_attn_dq_body(const u16* __restrict__ q, const u16* __restrict__ k, const u16* __restrict__ v,
                                                 const u16* __restrict__ kt, const u16* __restrict__ d_o,
                                                 const float* __restrict__ lse, const float* __restrict__ delta,
                                                 u16* __restrict__ dqkv, int T, int Tp, int heads, float scale, int bx, int bh) {
  const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
  const int t0 = bx * 32;
  const int64_t rb = (int64_t)bh * Tp;
  uint4 qf[4], dof[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    qf[kk] = *(const uint4*)(q + rfrag(bh, Tp >> 5, bx, kk, lhi, l31));
    dof[kk] = *(const uint4*)(d_o + rfrag(bh, Tp >> 5, bx, kk, lhi, l31));
  }
  const float my_lse = lse[rb + t0 + l31], my_delta = delta[rb + t0 + l31];
  f32x16 g0, g1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
  struct Tile { uint4 k[4], v[4], kt[2][2]; };           // a key tile's fragments, fetched one tile ahead (see the dK/dV body)
  const int ntb = Tp >> 5;
  auto load_tile = [&](int sb, Tile& f) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f.k[kk] = *(const uint4*)(k + rfrag(bh, ntb, sb, kk, lhi, l31));
      f.v[kk] = *(const uint4*)(v + rfrag(bh, ntb, sb, kk, lhi, l31));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < 2; ++db) f.kt[ks][db] = *(const uint4*)(kt + tfrag(bh, ntb, sb, ks, db, lhi, l31));
  };
  auto compute = [&](const Tile& f, int s0) {
    f32x16 sacc, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      sacc = T_::mfma32(f.k[kk], qf[kk], sacc);
      dp = T_::mfma32(f.v[kk], dof[kk], dp);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s = s0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float p = s < T ? __expf(sacc[r] * scale - my_lse) : 0.f;
      sacc[r] = p * (dp[r] - my_delta) * scale;                 // dS^T[s][t]
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = sacc[8 * ks + j];
      const uint4 dsf = pack8<T_>(g);
      g0 = T_::mfma32(f.kt[ks][0], dsf, g0);
      g1 = T_::mfma32(f.kt[ks][1], dsf, g1);
    }
  };
  Tile ta, tb_;
  load_tile(0, ta);
  for (int sb = 0; sb < ntb; sb += 2) {
    load_tile(sb + 1 < ntb ? sb + 1 : sb, tb_);
    __builtin_amdgcn_sched_barrier(0);
    compute(ta, sb * 32);
    __builtin_amdgcn_sched_barrier(0);
    if (sb + 1 < ntb) {
      load_tile(sb + 2 < ntb ? sb + 2 : sb + 1, ta);
      __builtin_amdgcn_sched_barrier(0);
      compute(tb_, (sb + 1) * 32);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int t = t0 + l31;
  if (t < T) {
    const int n = bh / heads, h = bh - n * heads, C = heads * 64;
    u16* ob = dqkv + ((int64_t)n * T + t) * 3 * C + h * 64 + 4 * lhi;        // dQ slot
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *(uint2*)(ob + 8 * g) = pack4<T_>(g0[4 * g], g0[4 * g + 1], g0[4 * g + 2], g0[4 * g + 3]);
      *(uint2*)(ob + 32 + 8 * g) = pack4<T_>(g1[4 * g], g1[4 * g + 1], g1[4 * g + 2], g1[4 * g + 3]);
    }
  }
}

// dK, dV: wave per 32-key tile.  S = Q.K^T and dP = dO.V^T put the key on the lane and the query on the accumulator rows,
// so P and dS are the B operands of dV^T += dO^T . P and dK^T += Q^T . dS.
template <typename T_>
__device__ __forceinline__ void vit_attn_dkdv_body(const u16* __restrict__ q, const u16* __restrict__ k, const u16* __restrict__ v,
                                                   const u16* __restrict__ qt, const u16* __restrict__ d_o, const u16* __restrict__ d_ot,
                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                   u16* __restrict__ dqkv, int T, int Tp, int heads, float scale, int bx, int bh) {
  const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
  const int s0 = bx * 32;
  const int64_t rb = (int64_t)bh * Tp;
  uint4 kf[4], vf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    kf[kk] = *(const uint4*)(k + rfrag(bh, Tp >> 5, bx, kk, lhi, l31));
    vf[kk] = *(const uint4*)(v + rfrag(bh, Tp >> 5, bx, kk, lhi, l31));
  }
  const bool key_ok = s0 + l31 < T;
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  // One wave, no LDS: every operand of a query tile comes straight from global memory in fragment order.  The tile's 16 fragment loads and
  // its 32 lse / delta values are fetched ONE TILE AHEAD into a second register set (a 64-thread workgroup has the registers): the loop
  // used to wait a full memory latency per tile with two or three waves per SIMD to cover it.
  struct Tile { uint4 q[4], dO[4], dot[2][2], qt[2][2]; float ls[16], dl[16]; };
  const int ntb = Tp >> 5;
  auto load_tile = [&](int tb, Tile& f) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f.q[kk] = *(const uint4*)(q + rfrag(bh, ntb, tb, kk, lhi, l31));
      f.dO[kk] = *(const uint4*)(d_o + rfrag(bh, ntb, tb, kk, lhi, l31));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int64_t fo = tfrag(bh, ntb, tb, ks, db, lhi, l31);
        f.dot[ks][db] = *(const uint4*)(d_ot + fo); f.qt[ks][db] = *(const uint4*)(qt + fo);
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      f.ls[r] = lse[rb + t]; f.dl[r] = delta[rb + t];
    }
  };
  auto compute = [&](const Tile& f, int t0) {
    f32x16 sacc, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      sacc = T_::mfma32(f.q[kk], kf[kk], sacc);     // rows = queries, lane = key
      dp = T_::mfma32(f.dO[kk], vf[kk], dp);
    }
    float pf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float p = (key_ok && t < T) ? __expf(sacc[r] * scale - f.ls[r]) : 0.f;
      pf[r] = p;
      sacc[r] = p * (dp[r] - f.dl[r]) * scale;            // dS[t][s]
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 pfrag = pack8<T_>(pf + 8 * ks);
      float g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = sacc[8 * ks + j];
      const uint4 dsf = pack8<T_>(g);
      dv0 = T_::mfma32(f.dot[ks][0], pfrag, dv0); dk0 = T_::mfma32(f.qt[ks][0], dsf, dk0);
      dv1 = T_::mfma32(f.dot[ks][1], pfrag, dv1); dk1 = T_::mfma32(f.qt[ks][1], dsf, dk1);
    }
  };
  Tile ta, tb_;
  load_tile(0, ta);
  for (int tb = 0; tb < ntb; tb += 2) {                  // two tiles per trip: the register sets swap roles without copies
    load_tile(tb + 1 < ntb ? tb + 1 : tb, tb_);
    __builtin_amdgcn_sched_barrier(0);
    compute(ta, tb * 32);
    __builtin_amdgcn_sched_barrier(0);
    if (tb + 1 < ntb) {
      load_tile(tb + 2 < ntb ? tb + 2 : tb + 1, ta);
      __builtin_amdgcn_sched_barrier(0);
      compute(tb_, (tb + 1) * 32);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int s = s0 + l31;
  if (s < T) {
    const int n = bh / heads, h = bh - n * heads, C = heads * 64;
    u16* okp = dqkv + ((int64_t)n * T + s) * 3 * C + C + h * 64 + 4 * lhi;       // dK slot
    u16* ovp = okp + C;                                                           // dV slot
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *(uint2*)(okp + 8 * g) = pack4<T_>(dk0[4 * g], dk0[4 * g + 1], dk0[4 * g + 2], dk0[4 * g + 3]);
      *(uint2*)(okp + 32 + 8 * g) = pack4<T_>(dk1[4 * g], dk1[4 * g + 1], dk1[4 * g + 2], dk1[4 * g + 3]);
      *(uint2*)(ovp + 8 * g) = pack4<T_>(dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]);
      *(uint2*)(ovp + 32 + 8 * g) = pack4<T_>(dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]);
    }
  }
}

// dQ and dK/dV of one attention layer in ONE launch (grid.z selects the role): each role alone is 9 x 128 single-wave workgroups
// -- about one wave per SIMD -- and runs latency-bound; issued together the two roles fill each other's stalls.
template <typename T_>
__global__ __launch_bounds__(64) void vit_attn_bwd_kernel(const u16* __restrict__ q, const u16* __restrict__ k, const u16* __restrict__ v,
                                                          const u16* __restrict__ qt, const u16* __restrict__ kt,
                                                          const u16* __restrict__ d_o, const u16* __restrict__ d_ot,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          u16* __restrict__ dqkv, int T, int Tp, int heads, float scale) {
  int bx, bh;
  head_xcd_remap(bx, bh);
  if (blockIdx.z == 0) vit_attn_dkdv_body<T_>(q, k, v, qt, d_o, d_ot, lse, delta, dqkv, T, Tp, heads, scale, bx, bh);
  else vit_attn_dq_body<T_>(q, k, v, kt, d_o, lse, delta, dqkv, T, Tp, heads, scale, bx, bh);
}

}  // namespace

#define VIT_BY_DTYPE(KERN, G, B, ...)                                                             \
  do {                                                                                            \
    if (dtype == PMI_DT_BF16) hipLaunchKernelGGL(KERN<BF16>, G, B, 0, (hipStream_t)s, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERN<F16>, G, B, 0, (hipStream_t)s, __VA_ARGS__);                     \
  } while (0)

extern "C" int pmi_vit_attn_fwd(const void* qkv, void* ws16, float* lse, void* out, int N, int T, int heads, float scale, int dtype,
                                pmi_stream_t s) {
  if (!qkv || !ws16 || !lse || !out || N <= 0 || T <= 0 || heads <= 0) return PMI_ERR_ARG;
  const int Tp = (T + 31) / 32 * 32;
  const int64_t blk = (int64_t)N * heads * Tp * 64;
  u16* w = (u16*)ws16;   // Q, K, V, Qt, Kt, Vt
  dim3 g(Tp / 32, N * heads);
  VIT_BY_DTYPE(vit_qkv_split_kernel, g, dim3(256), (const u16*)qkv, w, w + blk, w + 2 * blk, w + 3 * blk, w + 4 * blk, w + 5 * blk, T, Tp, heads);
  PMI_CHECK_LAUNCH();
  VIT_BY_DTYPE(vit_attn_fwd_kernel, g, dim3(64), w, w + blk, w + 5 * blk, (u16*)out, lse, T, Tp, heads, scale);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

extern "C" int pmi_vit_attn_bwd(const void* ws16, const float* lse, const void* o, const void* dout, void* ws16b, float* delta,
                                void* dqkv, int N, int T, int heads, float scale, int dtype, pmi_stream_t s) {
  if (!ws16 || !lse || !o || !dout || !ws16b || !delta || !dqkv || N <= 0 || T <= 0 || heads <= 0) return PMI_ERR_ARG;
  const int Tp = (T + 31) / 32 * 32;
  const int64_t blk = (int64_t)N * heads * Tp * 64;
  const u16* w = (const u16*)ws16;
  u16* wb = (u16*)ws16b;   // dO, dOt
  dim3 g(Tp / 32, N * heads);
  VIT_BY_DTYPE(vit_do_prep_kernel, g, dim3(256), (const u16*)dout, (const u16*)o, wb, wb + blk, delta, T, Tp, heads);
  PMI_CHECK_LAUNCH();
  dim3 g2(Tp / 32, N * heads, 2);
  VIT_BY_DTYPE(vit_attn_bwd_kernel, g2, dim3(64), w, w + blk, w + 2 * blk, w + 3 * blk, w + 4 * blk, wb, wb + blk, lse, delta, (u16*)dqkv, T, Tp,
               heads, scale);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
