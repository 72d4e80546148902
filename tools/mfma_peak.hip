// Calibration micro-benchmark (not part of the product): bare v_mfma_f32_32x32x16_bf16 rate with W waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  uint4 a = make_uint4(threadIdx.x * 2654435761u ^ seed, threadIdx.x * 40503u + seed, 0x3f803f80u ^ threadIdx.x, 0x3f003f80u + threadIdx.x);
  uint4 b = make_uint4(0x3e803f00u + threadIdx.x, a.x ^ 0x1234567u, a.y * 3, 0x3f803e00u);
  a.x = (a.x & 0x807f807fu) | 0x3f003f00u; a.y = (a.y & 0x807f807fu) | 0x3f003f00u;
  b.y = (b.y & 0x807f807fu) | 0x3f003f00u; b.z = (b.z & 0x807f807fu) | 0x3f003f00u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  int threads = argc > 1 ? atoi(argv[1]) : 512, blocks = argc > 2 ? atoi(argv[2]) : 256 * 8, iters = 2000;
  float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 123u + rep);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (threads / 64) * iters * 8 * 32768.0;
    printf("threads=%d blocks=%d: %.3f ms  %.1f TFLOP/s\n", threads, blocks, ms, flops / ms / 1e9);
  }
  return 0;
}
