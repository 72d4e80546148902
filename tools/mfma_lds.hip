// Calibration (not product): 8 waves/CU, each: per k-step 6 ds_read_b128 (2 "x" + 4 "w" fragments, conflict-free) feeding
// 8 MFMA 32x32x16 into 8 accumulators — the halo kernel's inner pattern without barriers / global loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[109056];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  for (int i = tid; i < 109056 / 4; i += 512) ((unsigned*)smem)[i] = 0x3f003f00u ^ ((i * 2654435761u) & 0x807f807fu);
  __syncthreads();
  const int wm = wid / 2, wn = wid % 2;
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const char* patch = smem; const char* wb = smem + 43520;
  for (int it = 0; it < iters; ++it) {
    const int tap = it % 9, dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ch = kk * 2 + lhi;
      uint4 xf[2], wf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) xf[i] = *(const uint4*)(patch + swz((2 * wm + i + dy) * 34 + l31 + dx, ch));
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *(const uint4*)(wb + (it & 1) * 32768 + swz(wn * 128 + j * 32 + l31, ch));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[j]), __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0, 0, 0);
    }
    if (MODE == 1) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 512 + tid] = s;
}
int main(int argc, char** argv) {
  int blocks = 256 * 4, iters = 9 * 40;
  float* out; hipMalloc(&out, (size_t)blocks * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 8 * iters * 32 * 32768.0;
      printf("mode=%d (1 = barrier per tap): %.3f ms  %.1f TFLOP/s  -> %.0f cycles/tap @2.1GHz\n", mode, ms, flops / ms / 1e9, ms * 1e-3 * 2.1e9 / (4.0 * iters));
    }
  return 0;
}
