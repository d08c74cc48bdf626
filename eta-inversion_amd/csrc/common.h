// Shared helpers for the gfx950 kernels of libetainv_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>

#include "../../include/etainv.h"

namespace etainv {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
// Exact (erf) GELU for two values at once on the packed-fp32 pipe: erf(|x|) = 1 - 2^(-p(|x|)), p = x (c1 + c2 x + ... + c7 x^6) fitted
// to -log2(erfc) on [0, 4.3] (minimax, tools/fit_gelu.py: |erf error| <= 1.8e-7 in fp32, the accuracy of the A&S 7.1.26 form used
// before) -- one transcendental instead of two and 7 packed FMAs per PAIR instead of 14 scalar ops per value: the GEGLU epilogue of a
// K = 320 GEMM is VALU-bound on erf (16384 values per 256 x 128 tile against 5120 cycles of matrix work).
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu_pair(gelu_f32x2 x) {
  gelu_f32x2 ax;
  ax[0] = fminf(fabsf(x[0]) * 0.70710678118654752f, 4.3f);
  ax[1] = fminf(fabsf(x[1]) * 0.70710678118654752f, 4.3f);
  gelu_f32x2 p = ax * 0.000100211372f + (-0.000461519738f);          // coefficients negated: t = -p(ax)
  p = p * ax + (-0.00230234699f);
  p = p * ax + 0.0294526188f;
  p = p * ax + (-0.148963716f);
  p = p * ax + (-0.918328635f);
  p = p * ax + (-1.62791373f);
  const gelu_f32x2 t = p * ax;
  gelu_f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(t[0]);
  e[1] = __builtin_amdgcn_exp2f(t[1]);
  const gelu_f32x2 er = 1.0f - e;
  gelu_f32x2 s;
  s[0] = copysignf(er[0], x[0]);
  s[1] = copysignf(er[1], x[1]);
  const gelu_f32x2 h = x * 0.5f;
  return h * s + h;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- environment switches: ONE rule in both layers (etainv/pipeline.py _env_on has the same): a boolean switch is ON when the variable is set to
// anything but "" or "0"; `env_flag(name, dflt)` reads a switch whose default is `dflt` (unset = default, "0" / "" = off, anything else = on)
static inline bool env_on(const char* name) {
  const char* v = getenv(name);
  return v && v[0] && !(v[0] == '0' && !v[1]);
}
static inline bool env_flag(const char* name, bool dflt) { return getenv(name) ? env_on(name) : dflt; }

// one-time per-device state of the launchers (function attributes, zero pages, workspaces) is indexed by the current device
constexpr int kMaxDevices = 16;
static inline int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return (d >= 0 && d < kMaxDevices) ? d : 0;
}

// ---- opt-in per-kernel-class timing with HIP events on the launch stream (bench.py roofline numbers).
// Off by default: the launchers then do nothing extra.  work = algorithmic FLOPs (MFMA kernels) or bytes (HBM-bound).
enum ProfClass { PROF_IGEMM = 0, PROF_SELF_ATTN = 1, PROF_CROSS_ATTN = 2, PROF_GROUPNORM = 3, PROF_LAYERNORM = 4, PROF_OTHER = 5, PROF_NCLASS = 6 };
bool prof_enabled();
void prof_pause(bool on);   // nested launchers: the outer scope times the whole operation
void prof_begin(int cls, double work, hipStream_t s, double bytes = 0.0);
void prof_end(hipStream_t s);
struct ProfScope {
  hipStream_t s;
  bool on;
  // bytes: algorithmic HBM bytes of the launch (every operand read once, the result written once) -- with `work` = FLOPs it gives the
  // launch's arithmetic intensity, i.e. which roofline bounds it
  ProfScope(int cls, double work, hipStream_t st, double bytes = 0.0) : s(st), on(prof_enabled()) { if (on) prof_begin(cls, work, s, bytes); }
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
