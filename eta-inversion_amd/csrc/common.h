// Shared helpers for the gfx950 kernels of libetainv_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/etainv.h"

namespace etainv {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ---- error plumbing (thread-local message, integer status; no exceptions across the ABI)
void set_error(const std::string& msg);
#define ETAINV_FAIL(msg)                                                        \
  do {                                                                          \
    ::etainv::set_error(std::string(__func__) + ": " + (msg));                  \
    return 1;                                                                   \
  } while (0)
#define ETAINV_CHECK(cond, msg) \
  do {                          \
    if (!(cond)) ETAINV_FAIL(msg); \
  } while (0)
#define ETAINV_HIP(expr)                                                                   \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) ETAINV_FAIL(std::string(#expr) + " -> " + hipGetErrorString(_e)); \
  } while (0)
#define ETAINV_LAUNCH_CHECK()                                                          \
  do {                                                                                 \
    hipError_t _e = hipGetLastError();                                                 \
    if (_e != hipSuccess) ETAINV_FAIL(std::string("launch: ") + hipGetErrorString(_e)); \
  } while (0)

// ---- scalar conversions
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<f16>(f16 v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// ---- wave64 reductions (DPP-free shuffles; 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU; erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, ~14 instructions instead of libm erff's ~40:
// the GEGLU epilogue evaluates it 32 times per lane per output tile)
// x * Phi(x) for two values at once on the packed-fp32 pipe: Phi(x) ~ sigmoid(x (c1 + c3 x^2 + c5 x^4)), |x| clamped to 9 inside the
// polynomial (minimax fit of the coefficients, tools/fit_gelu.py: |error| <= 2.6e-5 absolute on gelu, below half an fp16 ulp
// for |gelu| > 0.05 and far below a bf16 ulp).  7 packed ops + 2 exp2 + 2 rcp per PAIR instead of 14 ops + exp2 + rcp per
// value: the GEGLU epilogue of a K = 320 GEMM spent more VALU cycles in erf than the tile spends in the matrix pipe.
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu_pair(gelu_f32x2 x) {
  gelu_f32x2 xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -9.0f, 9.0f);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -9.0f, 9.0f);
  const gelu_f32x2 x2 = xc * xc;
  // coefficients pre-multiplied by -log2(e): t = -u * log2(e)
  gelu_f32x2 p = x2 * 0.0010142652f + (-0.1067757382f);   // c5 = -0.0007030350676, c3 = 0.07401130191
  p = p * x2 + (-2.3011213228f);                           // c1 = 1.595015757
  const gelu_f32x2 t = p * xc;
  gelu_f32x2 d;
  d[0] = 1.0f + __builtin_amdgcn_exp2f(t[0]);
  d[1] = 1.0f + __builtin_amdgcn_exp2f(t[1]);
  gelu_f32x2 r;
  r[0] = __builtin_amdgcn_rcpf(d[0]);
  r[1] = __builtin_amdgcn_rcpf(d[1]);
  return x * r;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- opt-in per-kernel-class timing with HIP events on the launch stream (bench.py roofline numbers).
// Off by default: the launchers then do nothing extra.  work = algorithmic FLOPs (MFMA kernels) or bytes (HBM-bound).
enum ProfClass { PROF_IGEMM = 0, PROF_SELF_ATTN = 1, PROF_CROSS_ATTN = 2, PROF_GROUPNORM = 3, PROF_LAYERNORM = 4, PROF_OTHER = 5, PROF_NCLASS = 6 };
bool prof_enabled();
void prof_begin(int cls, double work, hipStream_t s);
void prof_end(hipStream_t s);
struct ProfScope {
  hipStream_t s;
  bool on;
  ProfScope(int cls, double work, hipStream_t st) : s(st), on(prof_enabled()) { if (on) prof_begin(cls, work, s); }
  ~ProfScope() { if (on) prof_end(s); }
};

// dispatch a generic lambda over the three element types
#define ETAINV_DISPATCH_DTYPE(dt, T, ...)                          \
  switch (dt) {                                                    \
    case ETAINV_F32: { typedef float T; __VA_ARGS__; } break;      \
    case ETAINV_F16: { typedef ::etainv::f16 T; __VA_ARGS__; } break;  \
    case ETAINV_BF16: { typedef ::etainv::bf16 T; __VA_ARGS__; } break; \
    default: ETAINV_FAIL("bad dtype");                             \
  }
#define ETAINV_DISPATCH_HALF(dt, T, ...)                           \
  switch (dt) {                                                    \
    case ETAINV_F16: { typedef ::etainv::f16 T; __VA_ARGS__; } break;  \
    case ETAINV_BF16: { typedef ::etainv::bf16 T; __VA_ARGS__; } break; \
    default: ETAINV_FAIL("compute dtype must be f16 or bf16");     \
  }

}  // namespace etainv
