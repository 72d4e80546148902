// Calibration micro-benchmark (not part of the product): bare MFMA rate with W waves per CU, bf16 vs f16 operands, 32x32x16 vs 16x16x32.
// Round 3 use: the f16 engine's convolutions run ~2 % behind the bf16 engine's on the same kernels and instruction mix -- is it the matrix
// pipe's clock under f16 operands (11-bit mantissa multipliers) or the code?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>   // 0: bf16 32x32x16, 1: f16 32x32x16, 2: bf16 16x16x32, 3: f16 16x16x32
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // operands near 1 with varied mantissas (0x3cxx in f16 = 1.xx, 0x3f8x in bf16 = 1.xx): no denormals / infinities in either type
  const unsigned t = threadIdx.x * 2654435761u ^ seed;
  const unsigned hi = (KIND & 1) ? 0x3c003c00u : 0x3f803f80u, mask = (KIND & 1) ? 0x03ff03ffu : 0x007f007fu;
  uint4 a = make_uint4(hi | (t & mask), hi | ((t >> 3) & mask), hi | ((t * 7) & mask), hi | ((t >> 5) & mask));
  uint4 b = make_uint4(hi | ((t * 13) & mask), hi | ((t >> 7) & mask), hi | ((t * 29) & mask), hi | ((t >> 2) & mask));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
      else if constexpr (KIND == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[j], 0, 0, 0);
      else {
        f32x4* q = (f32x4*)&acc[j];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          if constexpr (KIND == 2) q[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), q[h], 0, 0, 0);
          else q[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), q[h], 0, 0, 0);
        }
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  int threads = argc > 1 ? atoi(argv[1]) : 512, blocks = argc > 2 ? atoi(argv[2]) : 256 * 8, iters = 2000;
  float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"bf16 32x32x16", "f16  32x32x16", "bf16 16x16x32", "f16  16x16x32"};
  for (int rep = 0; rep < 3; ++rep)
    for (int kind = 0; kind < 4; ++kind) {
      hipEventRecord(e0);
      for (int l = 0; l < 4; ++l) {       // ~4 x 13 ms: long enough for the clock to settle under the load
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 123u + rep);
        else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 123u + rep);
        else if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 123u + rep);
        else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, out, iters, 123u + rep);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per wave and iteration: 8 x one 32x32x16 (32768 flop) or 8 x four 16x16x32 (4 x 16384 flop)
      double flops = 4.0 * (double)blocks * (threads / 64) * iters * 8 * (kind < 2 ? 32768.0 : 65536.0);
      printf("%s threads=%d blocks=%d: %.3f ms  %.1f TFLOP/s\n", names[kind], threads, blocks, ms, flops / ms / 1e9);
    }
  return 0;
}
