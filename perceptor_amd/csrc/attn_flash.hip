// Flash-style attention for head dims 8..160 (multiples of 8) with a separate key / value sequence, on gfx950 MFMA (32x32x16).
// The StableDiffusion transformer blocks: self-attention over 4096 / 1024 / 256 / 64 latent pixels with 40 / 80 / 160-channel heads and
// cross-attention onto the 77 prompt tokens (perceptor/models/stable_diffusion/attention.py:268-298; the xformers call at :285).
//
// Same scheme as attn.hip's 64-channel kernel, generalised:
//   S^T[s][t] = sum_c K[s][c] Q[t][c]   (A = K rows, B = Q rows; KQ = ceil(d / 16) k-steps, channels zero-padded)
//   online softmax over s lane-local (+ one exchange with lane^32), in the exp2 domain (scale * log2 e folded into one multiply)
//   O^T[c][t] += V^T[c][s] P^T[s][t]    (A = V^T rows in DB = ceil(d / 32) blocks, B = the P accumulator re-used in place)
// One wave owns QT x 32 queries (QT = 1 by default; with QT = 2 every K / V^T fragment streamed from L2 feeds two MFMAs, but the
// 204 VGPRs leave one wave per SIMD and it measured slower: kept as an A/B option, pmi_set_option(9, 2)).
// No score matrix in HBM (the batched-GEMM path wrote T x T fp32 scores and 16-bit probabilities: 0.8 GB per sample at T = 4096).
#include "common.h"
#include "../../include/perceptor_hip.h"

namespace {

// Q / K fragments: element (t, c) of a 32-token block at [block][kk = c / 16][lhi = (c / 8) & 1][t & 31][c & 7]
__device__ __forceinline__ int64_t rfrag_g(int64_t blk, int KQ, int kk, int lhi, int l31) {
  return (((blk * KQ + kk) * 2 + lhi) * 32 + l31) * 8;
}
// V^T fragments: [block][ks (2)][db (DB)][lhi][c & 31][8 tokens {16 ks + 4 lhi + 0..3, 16 ks + 8 + 4 lhi + 0..3}]
__device__ __forceinline__ int64_t tfrag_g(int64_t blk, int DB, int ks, int db, int lhi, int l31) {
  return ((((blk * 2 + ks) * DB + db) * 2 + lhi) * 32 + l31) * 8;
}

// rows of src ([N][T][ld], head h at channel offset h*d) -> fragment order; zero fill for t >= T and c >= d
template <bool TRANSPOSED>
__device__ __forceinline__ void split_block(const u16* __restrict__ src, int ld, u16* __restrict__ dst, int n, int h, int bh, int tb,
                                            int T, int ntb, int d, int KQ, int DB, u16 (*sv)[168]) {
  const int tid = threadIdx.x, row = tid >> 3, ch = tid & 7, t = tb * 32 + row;
  const int64_t blk = (int64_t)bh * ntb + tb;
  const u16* r = src + ((int64_t)n * T + t) * ld + h * d;
  if constexpr (!TRANSPOSED) {
    for (int c8 = ch; c8 < 2 * KQ; c8 += 8) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (t < T && c8 * 8 < d) v = *(const uint4*)(r + c8 * 8);
      *(uint4*)(dst + rfrag_g(blk, KQ, c8 >> 1, c8 & 1, row)) = v;
    }
  } else {
    for (int c8 = ch; c8 < 4 * DB; c8 += 8) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (t < T && c8 * 8 < d) v = *(const uint4*)(r + c8 * 8);
      *(uint4*)(&sv[row][c8 * 8]) = v;
    }
    __syncthreads();
    for (int i = tid; i < DB * 128; i += 256) {
      const int c = i >> 2, fks = (i >> 1) & 1, flhi = i & 1;
      u16 e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = sv[16 * fks + 4 * flhi + (j & 3) + 8 * (j >> 2)][c];
      *(uint4*)(dst + tfrag_g(blk, DB, fks, c >> 5, flhi, c & 31)) =
          make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void flash_split_kernel(const u16* __restrict__ q, int ldq, const u16* __restrict__ k, const u16* __restrict__ v,
                                                          int ldkv, u16* __restrict__ qf, u16* __restrict__ kf, u16* __restrict__ vtf,
                                                          int T, int Tk, int heads, int d, int KQ, int DB) {
  __shared__ u16 sv[32][168];
  const int tb = blockIdx.x, bh = blockIdx.y, n = bh / heads, h = bh - n * heads;
  const int ntq = (T + 31) >> 5, ntk = (Tk + 31) >> 5;
  if (tb < ntq) split_block<false>(q, ldq, qf, n, h, bh, tb, T, ntq, d, KQ, DB, sv);
  if (tb < ntk) {
    split_block<false>(k, ldkv, kf, n, h, bh, tb, Tk, ntk, d, KQ, DB, sv);
    split_block<true>(v, ldkv, vtf, n, h, bh, tb, Tk, ntk, d, KQ, DB, sv);
  }
}

template <typename T_, int KQ, int DB, int QT>
__global__ __launch_bounds__(64) void attn_flash_kernel(const u16* __restrict__ qf, const u16* __restrict__ kf, const u16* __restrict__ vtf,
                                                        u16* __restrict__ out, int T, int Tk, int heads, int d, float scale_log2e) {
  const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
  const int nx = gridDim.x;
  const int lin = xcd_remap(blockIdx.x + nx * blockIdx.y, nx * gridDim.y);     // all query tiles of a head on one XCD (its K/V stay in that L2)
  const int bh = lin / nx, bx = lin - bh * nx;
  const int ntq = (T + 31) >> 5, ntk = (Tk + 31) >> 5;
  uint4 qr[QT][KQ];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int tb = min(bx * QT + qt, ntq - 1);                                   // a tile past the end repeats the last one (never stored)
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) qr[qt][kk] = *(const uint4*)(qf + rfrag_g((int64_t)bh * ntq + tb, KQ, kk, lhi, l31));
  }
  f32x16 o[QT][DB];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -1e30f; l_run[qt] = 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qt][db][r] = 0.f;
  }
  for (int sb = 0; sb < ntk; ++sb) {
    const int64_t blk = (int64_t)bh * ntk + sb;
    f32x16 sacc[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[qt][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
      const uint4 kfr = *(const uint4*)(kf + rfrag_g(blk, KQ, kk, lhi, l31));
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) sacc[qt] = T_::mfma32(kfr, qr[qt][kk], sacc[qt]);
    }
    const bool tail = (sb + 1) * 32 > Tk;
    uint4 pfrag[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (tail && sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi >= Tk) sacc[qt][r] = -1e30f;
        mx = fmaxf(mx, sacc[qt][r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale_log2e;              // the scale is positive: max commutes with it
      // Lazy rescaling: the running maximum only follows when a block exceeds it by more than 2^8 (the probabilities then stay below
      // 256: exact in fp32 sums, well inside the 16-bit operand range), so after the first blocks the accumulator rescale (one multiply per
      // accumulator register) and its exp2 almost never run.  The branch is wave-uniform.
      if (__any(mx > m_run[qt] + 8.f)) {
        const float m_new = fmaxf(m_run[qt], mx);
        const float alpha = exp2f(m_run[qt] - m_new);
        l_run[qt] *= alpha;
        m_run[qt] = m_new;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qt][db][r] *= alpha;
      }
      const float mneg = -m_run[qt];
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = exp2f(fmaf(sacc[qt][r], scale_log2e, mneg));   // one fused multiply-add + one v_exp per score
        sacc[qt][r] = p;
        rs += p;
      }
      rs += __shfl_xor(rs, 32);
      l_run[qt] += rs;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float pf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = sacc[qt][8 * ks + j];
        pfrag[qt][ks] = pack8<T_>(pf);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const uint4 vf = *(const uint4*)(vtf + tfrag_g(blk, DB, ks, db, lhi, l31));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][db] = T_::mfma32(vf, pfrag[qt][ks], o[qt][db]);
      }
  }
  const int n = bh / heads, h = bh - n * heads;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int t = (bx * QT + qt) * 32 + l31;
    if (t >= T) continue;
    const float inv = 1.f / l_run[qt];
    u16* ob = out + ((int64_t)n * T + t) * (heads * d) + h * d;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * db + 8 * g + 4 * lhi;
        if (c < d)
          *(uint2*)(ob + c) = pack4<T_>(o[qt][db][4 * g] * inv, o[qt][db][4 * g + 1] * inv, o[qt][db][4 * g + 2] * inv, o[qt][db][4 * g + 3] * inv);
      }
  }
}

// The same computation with FOUR waves (128 queries) per workgroup sharing every K / V^T fragment through LDS: one wave per 32 queries
// streams 7 KB (d = 40) per key block from L2, at T = 4096 that is 7 GB per call and 12 TB/s of L2 -> register traffic -- the bound
// of the one-wave kernel.  Here the workgroup loads a key block's fragments once (register-staged, one block ahead, double-buffered in
// LDS: one barrier per key block) and the four waves read them from LDS (fragment order: 64 lanes x 16 B contiguous, conflict-free).
template <typename T_, int KQ, int DB>
__global__ __launch_bounds__(256) void attn_flash_lds_kernel(const u16* __restrict__ qf, const u16* __restrict__ kf, const u16* __restrict__ vtf,
                                                            u16* __restrict__ out, int T, int Tk, int heads, int d, float scale_log2e) {
  constexpr int NP = KQ + 2 * DB;                      // 1 KB fragment pieces per key block
  constexpr int NPI = (NP + 3) / 4;                    // pieces staged per thread
  __shared__ uint4 kv[2][NP * 64];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nx = gridDim.x;
  const int lin = xcd_remap(blockIdx.x + nx * blockIdx.y, nx * gridDim.y);
  const int bh = lin / nx, bx = lin - bh * nx;
  const int ntq = (T + 31) >> 5, ntk = (Tk + 31) >> 5;
  const int tq = bx * 4 + wid;                         // this wave's query tile (past the end: repeats the last one, never stored)
  uint4 qr[KQ];
#pragma unroll
  for (int kk = 0; kk < KQ; ++kk) qr[kk] = *(const uint4*)(qf + rfrag_g((int64_t)bh * ntq + min(tq, ntq - 1), KQ, kk, lhi, l31));
  f32x16 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const u16* const kbase = kf + (int64_t)bh * ntk * KQ * 512;
  const u16* const vbase = vtf + (int64_t)bh * ntk * 2 * DB * 512;
  // piece p of key block sb: K fragment p (p < KQ) or V^T fragment p - KQ; thread (wave w, lane) stages pieces w, w + 4, ...
  // (plain macros: as lambdas capturing the staging array the compiler demoted it to scratch)
  uint4 st[NPI];
#define FLASH_LOAD_BLOCK(SB)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < NPI; ++i) {                                                                       \
    const int p = wid + 4 * i;                                                                                            \
    const u16* src = p < KQ ? kbase + ((int64_t)(SB) * KQ + p) * 512 : vbase + ((int64_t)(SB) * 2 * DB + (p - KQ)) * 512; \
    st[i] = p < NP ? *(const uint4*)(src + lane * 8) : make_uint4(0, 0, 0, 0);                                            \
  }
#define FLASH_STORE_BLOCK(BUF)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < NPI; ++i) {                                                                       \
    const int p = wid + 4 * i;                                                                                            \
    if (p < NP) kv[BUF][p * 64 + lane] = st[i];                                                                           \
  }
  FLASH_LOAD_BLOCK(0)
  FLASH_STORE_BLOCK(0)
  if (ntk > 1) { FLASH_LOAD_BLOCK(1) }
  __syncthreads();
  for (int sb = 0; sb < ntk; ++sb) {
    const uint4* const cur = kv[sb & 1];
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) sacc = T_::mfma32(cur[kk * 64 + lane], qr[kk], sacc);
    const bool tail = (sb + 1) * 32 > Tk;
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (tail && sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi >= Tk) sacc[r] = -1e30f;
      mx = fmaxf(mx, sacc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale_log2e;
    if (__any(mx > m_run + 8.f)) {                     // lazy rescaling, as in attn_flash_kernel
      const float m_new = fmaxf(m_run, mx);
      const float alpha = exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    const float mneg = -m_run;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = exp2f(fmaf(sacc[r], scale_log2e, mneg));
      sacc[r] = p;
      rs += p;
    }
    rs += __shfl_xor(rs, 32);
    l_run += rs;
    // the next block's fragments (in flight since the previous iteration) go to the other buffer: every wave finished reading it before
    // the barrier that ended the previous iteration
    if (sb + 1 < ntk) { FLASH_STORE_BLOCK((sb + 1) & 1) }
    if (sb + 2 < ntk) { FLASH_LOAD_BLOCK(sb + 2) }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float pf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = sacc[8 * ks + j];
      const uint4 pfrag = pack8<T_>(pf);
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] = T_::mfma32(cur[(KQ + ks * DB + db) * 64 + lane], pfrag, o[db]);
    }
    __syncthreads();
  }
  const int t = tq * 32 + l31;
  if (tq >= ntq || t >= T) return;
  const int n = bh / heads, h = bh - n * heads;
  const float inv = 1.f / l_run;
  u16* ob = out + ((int64_t)n * T + t) * (heads * d) + h * d;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * db + 8 * g + 4 * lhi;
      if (c < d) *(uint2*)(ob + c) = pack4<T_>(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
    }
#undef FLASH_LOAD_BLOCK
#undef FLASH_STORE_BLOCK
}

static int g_flash_qt = 0;       // A/B switch (pmi_set_option 9): 0 = automatic (LDS-shared kernel for long sequences), 1 / 2 = one-wave kernel with 1 / 2 query tiles per wave

template <typename T_, int KQ, int DB>
void launch_flash(const u16* qf, const u16* kf, const u16* vtf, u16* out, int N, int T, int Tk, int heads, int d, float sl2, hipStream_t st) {
  const int ntq = (T + 31) / 32;
  // two query tiles per wave where the sequence is long enough to still fill the chip (and the accumulators fit: DB <= 3)
  if (g_flash_qt != 1 && g_flash_qt != 2 && ntq >= 64) {     // long query sequences (T >= 2048): four waves per workgroup share K / V^T through LDS (0.61 vs 0.65 ms at T = 4096)
    hipLaunchKernelGGL((attn_flash_lds_kernel<T_, KQ, DB>), dim3((ntq + 3) / 4, N * heads), dim3(256), 0, st, qf, kf, vtf, out, T, Tk, heads, d, sl2);
    return;
  }
  if constexpr (DB <= 2) {
    if (g_flash_qt == 2) {      // measured at T = 4096, d = 40: 0.85 ms against 0.65 ms with one tile per wave (204 VGPRs: one wave per SIMD)
      hipLaunchKernelGGL((attn_flash_kernel<T_, KQ, DB, 2>), dim3((ntq + 1) / 2, N * heads), dim3(64), 0, st, qf, kf, vtf, out, T, Tk, heads, d, sl2);
      return;
    }
  }
  hipLaunchKernelGGL((attn_flash_kernel<T_, KQ, DB, 1>), dim3(ntq, N * heads), dim3(64), 0, st, qf, kf, vtf, out, T, Tk, heads, d, sl2);
}

template <typename T_>
int dispatch_flash(const u16* qf, const u16* kf, const u16* vtf, u16* out, int N, int T, int Tk, int heads, int d, float sl2, hipStream_t st) {
  const int KQ = (d + 15) / 16, DB = (d + 31) / 32;
#define CASE(kq, db) if (KQ == kq && DB == db) { launch_flash<T_, kq, db>(qf, kf, vtf, out, N, T, Tk, heads, d, sl2, st); return PMI_OK; }
  CASE(1, 1) CASE(2, 1) CASE(3, 2) CASE(4, 2) CASE(5, 3) CASE(6, 3) CASE(7, 4) CASE(8, 4) CASE(9, 5) CASE(10, 5)
#undef CASE
  return PMI_ERR_ARG;
}

}  // namespace

void pmi_attn_flash_qt(int v) { g_flash_qt = v; }

extern "C" int pmi_attn_flash_workspace(int N, int T, int Tk, int heads, int d) {       // in KiB (every term is a multiple of 1 KiB)
  if (N <= 0 || T <= 0 || Tk <= 0 || heads <= 0 || d <= 0 || d > 160 || (d & 7)) return -1;
  const int64_t KQ = (d + 15) / 16, DB = (d + 31) / 32, ntq = (T + 31) / 32, ntk = (Tk + 31) / 32;
  const int64_t kib = (int64_t)N * heads * (ntq * KQ + ntk * KQ + ntk * DB * 2);
  return kib > 0x7fffffff ? -1 : (int)kib;
}

extern "C" int pmi_attn_flash(const void* q, int ldq, const void* k, const void* v, int ldkv, void* out, void* ws, int N, int T, int Tk,
                              int heads, int d, float scale, int dtype, pmi_stream_t s) {
  if (!q || !k || !v || !out || !ws || N <= 0 || T <= 0 || Tk <= 0 || heads <= 0 || d <= 0 || d > 160 || (d & 7) || ldq < heads * d ||
      ldkv < heads * d || (ldq & 7) || (ldkv & 7) || (dtype != PMI_DT_BF16 && dtype != PMI_DT_F16))
    return PMI_ERR_ARG;
  const int64_t KQ = (d + 15) / 16, DB = (d + 31) / 32, ntq = (T + 31) / 32, ntk = (Tk + 31) / 32;
  u16* qf = (u16*)ws;
  u16* kf = qf + (int64_t)N * heads * ntq * KQ * 512;
  u16* vtf = kf + (int64_t)N * heads * ntk * KQ * 512;
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(flash_split_kernel, dim3((unsigned)(ntq > ntk ? ntq : ntk), N * heads), dim3(256), 0, st, (const u16*)q, ldq, (const u16*)k,
                     (const u16*)v, ldkv, qf, kf, vtf, T, Tk, heads, d, (int)KQ, (int)DB);
  PMI_CHECK_LAUNCH();
  const float sl2 = scale * 1.4426950408889634f;
  const int rc = dtype == PMI_DT_BF16 ? dispatch_flash<BF16>(qf, kf, vtf, (u16*)out, N, T, Tk, heads, d, sl2, st)
                                      : dispatch_flash<F16>(qf, kf, vtf, (u16*)out, N, T, Tk, heads, d, sl2, st);
  if (rc != PMI_OK) return rc;
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
