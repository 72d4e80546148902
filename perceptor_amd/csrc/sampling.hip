// Native pieces of the Predictions variants and clamp_with_grad, fp32 NCHW, one row = one sample:
//   pmi_quantile_abs     per-sample quantile of |x| (dynamic_threshold)        reference: guided_diffusion/predictions.py:156-172
//   pmi_randn            counter-based normal noise (step eta>0, resample_noise, noisy_reverse_step)   predictions.py:61-98,126-145
//   pmi_sort_rows        ascending sort of every row (wasserstein_*)           predictions.py:184-198
//   pmi_wasserstein      mean |sorted - icdf(linspace)|^p against the standard normal
//   pmi_clamp_grad       backward of clamp_with_grad                           transforms/clamp_with_grad.py:8-23
// All of it is HBM/L2-bound integer and elementwise work: coalesced 4-byte streams, LDS histograms, no MFMA.
#include "../../include/perceptor_hip.h"
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------ quantile
// Radix select on the bit pattern of |x| (non-negative floats order like their uint32 patterns): 4 passes of 8 bits, MSB first, one
// workgroup per sample.  The digit histogram is kept in 32 lane-private copies ([digit][lane & 31]: the copies of one digit sit in 32
// different banks) because almost every key of a pass shares its digit (exponent byte) and a single counter would serialise the wave.
constexpr int QT = 1024;

__device__ __forceinline__ uint32_t abs_key(float v) { return __float_as_uint(v) & 0x7FFFFFFFu; }

__global__ __launch_bounds__(QT) void quantile_abs_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float q) {
  __shared__ uint32_t hist[256 * 32];
  __shared__ uint32_t tot[256];
  __shared__ uint32_t sel[4];            // [0] chosen digit, [1] rank inside the digit, [2] elements <= chosen key so far
  __shared__ uint32_t red[QT / 64];
  const float* row = x + (int64_t)blockIdx.x * n;
  const int tid = threadIdx.x, copy = tid & 31;
  // torch.quantile, interpolation "linear": rank = q * (n - 1) in the input dtype, lerp between floor and ceil order statistics
  const float rank = q * (float)(n - 1);
  const float below = floorf(rank);
  const uint32_t k_lo = (uint32_t)below, k_hi = (uint32_t)ceilf(rank);
  const float w = rank - below;

  uint32_t prefix = 0, mask = 0, k = k_lo, le_before = 0;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = tid; i < 256 * 32; i += QT) hist[i] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n; i += QT) {
      const uint32_t key = abs_key(row[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[((key >> shift) & 255) * 32 + copy], 1u);
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t s = 0;
      for (int c = 0; c < 32; ++c) s += hist[tid * 32 + ((c + tid) & 31)];
      tot[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = 0;
      int d = 0;
      for (; d < 255; ++d) {
        if (acc + tot[d] > k) break;
        acc += tot[d];
      }
      sel[0] = (uint32_t)d;
      sel[1] = k - acc;
      sel[2] = le_before + acc + (pass == 3 ? tot[d] : 0u);
    }
    __syncthreads();
    prefix |= sel[0] << shift;
    mask |= 255u << shift;
    k = sel[1];
    le_before = sel[2];
    __syncthreads();
  }
  // prefix = key of order statistic k_lo; le_before = number of keys <= prefix.  The next order statistic is the same value unless
  // k_hi reaches past them, in which case it is the smallest key above.
  uint32_t best = 0x7FFFFFFFu;
  if (k_hi > k_lo && k_hi >= le_before) {
    for (int64_t i = tid; i < n; i += QT) {
      const uint32_t key = abs_key(row[i]);
      if (key > prefix && key < best) best = key;
    }
    for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < QT / 64; ++i) best = min(best, red[i]);
    }
  } else {
    best = prefix;
  }
  if (tid == 0) {
    const float a = __uint_as_float(prefix), b = __uint_as_float(best);
    // at::lerp for floats: the two forms keep the result monotone and exact at w = 0 and w = 1
    out[blockIdx.x] = w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);
  }
}

// ---------------------------------------------------------------------------------------------------------------------- randn
// Philox4x32-10 (Salmon et al., SC'11): counter = (block of 4 elements: lo, hi; stream: lo, hi), key = seed.  Element e of a draw is
// lane (e & 3) of block (e >> 2): the value depends on (seed, stream, global element index) only, never on launch geometry, device
// count or batch split.  Box-Muller on 23-bit uniforms (u = (r >> 9 + 0.5) * 2^-23, exact in fp32, never 0 or 1).
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ void box_muller(uint32_t r0, uint32_t r1, float& z0, float& z1) {
  const float u1 = ((float)(r0 >> 9) + 0.5f) * 1.1920928955078125e-07f;
  const float u2 = ((float)(r1 >> 9) + 0.5f) * 1.1920928955078125e-07f;
  const float rad = sqrtf(-2.f * logf(u1));
  const float th = 6.283185307179586f * u2;
  z0 = rad * cosf(th);
  z1 = rad * sinf(th);
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, int64_t n, int64_t first, uint64_t seed, uint64_t stream) {
  // one thread per aligned block of 4 global elements that overlaps [first, first + n)
  const int64_t blk0 = first >> 2;
  const int64_t b = blk0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b * 4 >= first + n) return;
  uint32_t c[4] = {(uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  float z[4];
  box_muller(c[0], c[1], z[0], z[1]);
  box_muller(c[2], c[3], z[2], z[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t e = b * 4 + j - first;
    if (e >= 0 && e < n) out[e] = z[j];
  }
}

__global__ __launch_bounds__(64) void philox_raw_kernel(uint32_t* __restrict__ out, uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
  if (threadIdx.x != 0) return;
  uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
  philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
  for (int j = 0; j < 4; ++j) out[j] = c[j];
}

// ----------------------------------------------------------------------------------------------------------------------- sort
// Bitonic network over rows padded to a power of two (pad = +inf).  Stages whose partner distance fits one 4096-element LDS block run
// inside one launch (the first 12 levels as one full local sort, then the tail of every later level); wider distances are one
// coalesced global pass each.  Element i of a row sorts ascending when (i & k) == 0 at level k.
constexpr int SB = 4096, ST_ = 1024;     // block elements, threads

__device__ __forceinline__ void cmpswap(float& a, float& b, bool up) {
  const bool sw = up ? (a > b) : (a < b);
  const float t = a;
  a = sw ? b : a;
  b = sw ? t : b;
}

__device__ __forceinline__ void lds_stage(float* s, int tid, int64_t base, int64_t k, int j) {
  // SB / 2 compare-exchanges at distance j inside the block
  for (int p = tid; p < SB / 2; p += ST_) {
    const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
    const bool up = ((base + i) & k) == 0;
    float a = s[i], b = s[i + j];
    cmpswap(a, b, up);
    s[i] = a;
    s[i + j] = b;
  }
  __syncthreads();
}

__global__ __launch_bounds__(ST_) void bitonic_local_sort_kernel(float* __restrict__ data, int64_t npad) {
  __shared__ float s[SB];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * SB;                 // blocks of all rows back to back; npad is a multiple of SB
  const int64_t in_row = base & (npad - 1);
  for (int i = tid; i < SB; i += ST_) s[i] = data[base + i];
  __syncthreads();
  for (int k = 2; k <= SB; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) lds_stage(s, tid, in_row, k, j);
  for (int i = tid; i < SB; i += ST_) data[base + i] = s[i];
}

__global__ __launch_bounds__(ST_) void bitonic_local_merge_kernel(float* __restrict__ data, int64_t npad, int64_t k) {
  __shared__ float s[SB];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * SB;
  const int64_t in_row = base & (npad - 1);
  for (int i = tid; i < SB; i += ST_) s[i] = data[base + i];
  __syncthreads();
  for (int j = SB >> 1; j > 0; j >>= 1) lds_stage(s, tid, in_row, k, j);
  for (int i = tid; i < SB; i += ST_) data[base + i] = s[i];
}

__global__ __launch_bounds__(256) void bitonic_global_step_kernel(float* __restrict__ data, int64_t npad, int64_t total_pairs, int64_t k, int64_t j) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= total_pairs) return;
  const int64_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));       // j and npad are powers of two, so pairs never straddle rows
  const bool up = ((i & (npad - 1)) & k) == 0;
  float a = data[i], b = data[i + j];
  const bool sw = up ? (a > b) : (a < b);
  if (sw) {
    data[i] = b;
    data[i + j] = a;
  }
}

__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, int64_t npad, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / npad, c = i - r * npad;
  out[i] = c < n ? x[r * n + c] : __builtin_inff();
}

// ---------------------------------------------------------------------------------------------------------------- wasserstein
// sum_i |s_i - icdf(p_i)|^power with p = torch.linspace(0.5/n, 1 - 0.5/n, n) and icdf(p) = erfinv(2p - 1) * sqrt(2), in fp32 like the
// reference.  Fixed-order two-level reduction (per-wave shuffle tree, per-block slots, then one block over the slots): deterministic.
constexpr int WB = 1024;    // partial sums (blocks) of the first level

__device__ __forceinline__ float linspace_at(float start, float end, float step, int64_t i, int64_t n) {
  // at::linspace: symmetric evaluation from both ends
  return i < n / 2 ? start + step * (float)i : end - step * (float)(n - i - 1);
}

__global__ __launch_bounds__(256) void wasserstein_partial_kernel(const float* __restrict__ sorted, int64_t n, int64_t npad, int rows, int power,
                                                                  float* __restrict__ partial) {
  __shared__ float red[4];
  const float margin = 0.5f / (float)n;
  const float start = margin, end = 1.f - margin;
  const float step = (end - start) / (float)(n - 1);
  const int64_t total = (int64_t)rows * n;
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)WB * 256) {
    const int64_t r = e / n, i = e - r * n;
    const float p = n > 1 ? linspace_at(start, end, step, i, n) : start;
    const float expect = erfinvf(2.f * p - 1.f) * 1.4142135623730951f;
    const float d = fabsf(sorted[r * npad + i] - expect);
    acc += power == 1 ? d : d * d;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(WB) void wasserstein_final_kernel(const float* __restrict__ partial, float* __restrict__ out, float inv_count) {
  __shared__ float red[WB / 64];
  float v = partial[threadIdx.x];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < WB / 64; ++i) s += red[i];
    out[0] = s * inv_count;
  }
}

// ------------------------------------------------------------------------------------------------------------ clamp_with_grad
__global__ __launch_bounds__(256) void clamp_grad_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ lo,
                                                         const float* __restrict__ hi, float* __restrict__ out, int64_t chw, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t nidx = i / chw;
  const float v = x[i], gi = g[i];
  const float c = fminf(fmaxf(v, lo[nidx]), hi[nidx]);
  // pass the gradient where the clamp is inactive, or where following it moves the value back towards the interval
  out[i] = (gi * (v - c) >= 0.f) ? gi : 0.f;
}

}  // namespace

#define ST ((hipStream_t)s)

extern "C" int pmi_quantile_abs(const float* x, float* out, int N, int64_t n, float q, pmi_stream_t s) {
  if (!x || !out || N <= 0 || n <= 0 || n >= ((int64_t)1 << 31) || !(q >= 0.f && q <= 1.f)) return PMI_ERR_ARG;
  hipLaunchKernelGGL(quantile_abs_kernel, dim3(N), dim3(QT), 0, ST, x, out, n, q);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

extern "C" int pmi_randn(float* out, int64_t n, int64_t first_element, int64_t seed_bits, int64_t stream_bits, pmi_stream_t s) {
  if (!out || n <= 0 || first_element < 0) return PMI_ERR_ARG;
  const uint64_t seed = (uint64_t)seed_bits, stream = (uint64_t)stream_bits;
  const int64_t blocks4 = ((first_element + n + 3) >> 2) - (first_element >> 2);
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((blocks4 + 255) / 256)), dim3(256), 0, ST, out, n, first_element, seed, stream);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

extern "C" int pmi_philox4x32_10(uint32_t* out4, int64_t counter_lo, int64_t counter_hi, int64_t key, pmi_stream_t s) {
  if (!out4) return PMI_ERR_ARG;
  hipLaunchKernelGGL(philox_raw_kernel, dim3(1), dim3(64), 0, ST, out4, (uint64_t)counter_lo, (uint64_t)counter_hi, (uint64_t)key);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

static int64_t padded(int64_t n) {
  int64_t p = SB;
  while (p < n) p <<= 1;
  return p;
}

// row pitch (elements) of pmi_sort_rows' output for rows of n elements: the next power of two, at least 4096; -1 if n is out of range
extern "C" int pmi_sort_rows_padded(int64_t n) { return (n <= 0 || n > ((int64_t)1 << 30)) ? -1 : (int)padded(n); }

// x [rows, n] -> work [rows, npad] sorted ascending per row (npad = pmi_sort_rows_padded(n); the +inf padding ends up at the back)
extern "C" int pmi_sort_rows(const float* x, float* work, int rows, int64_t n, pmi_stream_t s) {
  if (!x || !work || rows <= 0 || n <= 0 || n > ((int64_t)1 << 30)) return PMI_ERR_ARG;
  const int64_t npad = padded(n), total = npad * rows;
  if (total >= ((int64_t)1 << 40)) return PMI_ERR_ARG;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST, x, work, n, npad, total);
  PMI_CHECK_LAUNCH();
  const unsigned nblk = (unsigned)(total / SB);
  hipLaunchKernelGGL(bitonic_local_sort_kernel, dim3(nblk), dim3(ST_), 0, ST, work, npad);
  PMI_CHECK_LAUNCH();
  for (int64_t k = (int64_t)SB * 2; k <= npad; k <<= 1) {
    for (int64_t j = k >> 1; j >= SB; j >>= 1) {
      hipLaunchKernelGGL(bitonic_global_step_kernel, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, ST, work, npad, total / 2, k, j);
      PMI_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(bitonic_local_merge_kernel, dim3(nblk), dim3(ST_), 0, ST, work, npad, k);
    PMI_CHECK_LAUNCH();
  }
  return PMI_OK;
}

// sorted: [rows, npad] from pmi_sort_rows; partial: WB (1024) floats of workspace; out: 1 float = mean over rows*n of |.|^power
extern "C" int pmi_wasserstein(const float* sorted, int rows, int64_t n, int power, float* partial, float* out, pmi_stream_t s) {
  if (!sorted || !partial || !out || rows <= 0 || n <= 0 || n > ((int64_t)1 << 30) || (power != 1 && power != 2)) return PMI_ERR_ARG;
  const int64_t npad = padded(n);
  hipLaunchKernelGGL(wasserstein_partial_kernel, dim3(WB), dim3(256), 0, ST, sorted, n, npad, rows, power, partial);
  PMI_CHECK_LAUNCH();
  hipLaunchKernelGGL(wasserstein_final_kernel, dim3(1), dim3(WB), 0, ST, partial, out, 1.f / (float)((double)rows * (double)n));
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

extern "C" int pmi_clamp_grad(const float* x, const float* grad, const float* lo, const float* hi, float* out, int N, int64_t chw, pmi_stream_t s) {
  if (!x || !grad || !lo || !hi || !out || N <= 0 || chw <= 0) return PMI_ERR_ARG;
  const int64_t total = (int64_t)N * chw;
  hipLaunchKernelGGL(clamp_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST, x, grad, lo, hi, out, chw, total);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}
