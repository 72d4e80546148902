// Shared device helpers for the gfx950 (CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PMI_OK 0
#define PMI_ERR_ARG (-1)
#define PMI_ERR_LAUNCH (-2)

#define PMI_DT_F16 0
#define PMI_DT_BF16 1
#define PMI_DT_F16X2 2   /* "precise": every activation value is a hi + lo pair of f16 (~22 significant bits), see F16X2 below */

#define PMI_ACT_NONE 0
#define PMI_ACT_RELU 1
#define PMI_ACT_SILU 2
#define PMI_ACT_GELU 3
#define PMI_ACT_QUICKGELU 4
#define PMI_ACT_GEGLU 5     /* gemm_wd only: columns are (16 value | 16 gate) per 32; output = value * gelu(gate), N / 2 columns */

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct F16 {
  using elem = _Float16;
  using vec8 = f16x8;
  static __device__ __forceinline__ float to_f(u16 v) { return (float)__builtin_bit_cast(_Float16, v); }
  static __device__ __forceinline__ u16 from_f(float f) { return __builtin_bit_cast(u16, (_Float16)f); }
  // two values -> one packed word with ONE v_cvt_pk_f16_f32 (round to nearest even, as from_f): written as two scalar conversions and an
  // OR the compiler emits two v_cvt_pk (half of each wasted) + v_or_sdwa -- 3 issue slots per pair beside the MFMAs instead of 1
  static __device__ __forceinline__ uint32_t pack2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, f16x2)); }
  static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
struct BF16 {
  using elem = __bf16;
  using vec8 = bf16x8;
  static __device__ __forceinline__ float to_f(u16 v) { return __builtin_bit_cast(float, ((uint32_t)v) << 16); }
  static __device__ __forceinline__ u16 from_f(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2)); }
  static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

// Raw buffer loads: an offset with the top bit set is out of range for every resource made here (num_records < 2^31),
// and out-of-range loads return zeros -- a predicated load without a branch.
typedef unsigned int pmi_u32x4 __attribute__((ext_vector_type(4)));
#define PMI_BUF_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int64_t bytes) {
  const int n = bytes > 0x7ffffff0ll ? 0x7ffffff0 : (int)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voffset, uint32_t soffset) {
  const pmi_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// nontemporal forms (aux = 2) for data this kernel touches exactly once, and the matching store
__device__ __forceinline__ uint4 buf_load16_nt(__amdgpu_buffer_rsrc_t r, uint32_t voffset, uint32_t soffset) {
  const pmi_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 2);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_store16_nt(uint4 v, __amdgpu_buffer_rsrc_t r, uint32_t voffset, uint32_t soffset) {
  __builtin_amdgcn_raw_buffer_store_b128((pmi_u32x4){v.x, v.y, v.z, v.w}, r, voffset, soffset, 2);
}

template <typename T>
__device__ __forceinline__ void unpack8(uint4 v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = T::to_f((u16)(w[i] & 0xffff));
    f[2 * i + 1] = T::to_f((u16)(w[i] >> 16));
  }
}
template <typename T>
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(T::pack2(f[0], f[1]), T::pack2(f[2], f[3]), T::pack2(f[4], f[5]), T::pack2(f[6], f[7]));
}
// ---- "precise" activations (PMI_DT_F16X2) --------------------------------------------------------------------------
// A logical tensor with C channels is stored as 2C f16 per pixel: per group of G = min(32, C) channels first the G high
// parts hi = f16(x), then the G low parts lo = f16(x - hi).  hi + lo carries ~22 bits.  A convolution over such a tensor with
// the weights duplicated along K ([W | W] per group) is W*hi + W*lo with fp32 accumulation on the f16 MFMA, i.e. an fp32-grade
// product for f16-representable weights -- the MFMA main loops are unchanged, only loads of single values (GroupNorm, pooling,
// epilogues) add the two parts and stores split them again.
struct F16X2 : F16 {};
template <typename T> struct is_split { static constexpr bool v = false; };
template <> struct is_split<F16X2> { static constexpr bool v = true; };
template <typename T> __device__ __forceinline__ int row_elems(int C) { return is_split<T>::v ? 2 * C : C; }
__device__ __forceinline__ int split_group(int C) { return C < 32 ? C : 32; }
__device__ __forceinline__ int split_off(int c, int G) { return (c / G) * 2 * G + (c % G); }     // offset of the high part; low part at + G

// 8 consecutive logical channels starting at c (multiple of 8) of the pixel whose channels start at `row`
template <typename T>
__device__ __forceinline__ void load8(const u16* row, int c, int C, float* f) {
  if constexpr (is_split<T>::v) {
    const int G = split_group(C), o = split_off(c, G);
    float lo[8];
    unpack8<F16>(*(const uint4*)(row + o), f);
    unpack8<F16>(*(const uint4*)(row + o + G), lo);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += lo[e];
  } else {
    unpack8<T>(*(const uint4*)(row + c), f);
  }
}
template <typename T>
__device__ __forceinline__ void store8(u16* row, int c, int C, const float* f) {
  if constexpr (is_split<T>::v) {
    const int G = split_group(C), o = split_off(c, G);
    float lo[8];
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u16 h0 = F16::from_f(f[2 * i]), h1 = F16::from_f(f[2 * i + 1]);
      lo[2 * i] = f[2 * i] - F16::to_f(h0); lo[2 * i + 1] = f[2 * i + 1] - F16::to_f(h1);
      w[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
    }
    *(uint4*)(row + o) = make_uint4(w[0], w[1], w[2], w[3]);
    *(uint4*)(row + o + G) = pack8<F16>(lo);
  } else {
    *(uint4*)(row + c) = pack8<T>(f);
  }
}

template <typename T>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  return make_uint2(T::pack2(a, b), T::pack2(c, d));
}

// erf(x) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level): one v_rcp, one v_exp and a 5-term Horner
// chain instead of the ~40-instruction libm erff -- the exact-GELU epilogues are VALU-bound otherwise.
__device__ __forceinline__ float fast_erff(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float y = 1.f - poly * __expf(-ax * ax);
  return copysignf(y, x);
}

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case PMI_ACT_RELU: return x > 0.f ? x : 0.f;
    case PMI_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));       // v_rcp_f32 (1 ulp), no IEEE division sequence
    case PMI_ACT_GELU: return 0.5f * x * (1.f + fast_erff(x * 0.70710678118654752f));
    case PMI_ACT_QUICKGELU: return x * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x));
    default: return x;
  }
}

// Runs f(std::integral_constant<int, ACT>) for the runtime activation code, so an epilogue's element loops are compiled once per
// activation with act_apply's switch folded away (called per element with a runtime code, the switch is a chain of scalar branches per
// value: ~4 us of a 60 us convolution tile).
template <typename F>
__device__ __forceinline__ void act_switch(int act, F&& f) {
  switch (act) {
    case PMI_ACT_RELU: f(std::integral_constant<int, PMI_ACT_RELU>()); break;
    case PMI_ACT_SILU: f(std::integral_constant<int, PMI_ACT_SILU>()); break;
    case PMI_ACT_GELU: f(std::integral_constant<int, PMI_ACT_GELU>()); break;
    case PMI_ACT_QUICKGELU: f(std::integral_constant<int, PMI_ACT_QUICKGELU>()); break;
    default: f(std::integral_constant<int, PMI_ACT_NONE>()); break;
  }
}

// act over a short vector with ONE dispatch on the runtime code
template <int N>
__device__ __forceinline__ void act_apply_n(float* v, int act) {
  act_switch(act, [&](auto act_c) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = act_apply(v[e], decltype(act_c)::value);
  });
}

// d act(x) / dx
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case PMI_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case PMI_ACT_SILU: { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }
    case PMI_ACT_GELU: return 0.5f * (1.f + fast_erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    case PMI_ACT_QUICKGELU: { const float s = 1.f / (1.f + __expf(-1.702f * x)); return s * (1.f + 1.702f * x * (1.f - s)); }
    default: return 1.f;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Bijective XCD-aware remap of a linear workgroup id: workgroups that share an
// XCD (same id % 8 under round-robin dispatch) get a contiguous range of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

#define PMI_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return PMI_ERR_LAUNCH;            \
  } while (0)
