// Shader clock under sustained load on gfx950: s_memtime (shader clock cycles) against s_memrealtime (100 MHz) around loops that run for tens of milliseconds on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe tools/experiments/clock_probe.hip && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP4(x) x x x x
#define MF(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
#define EXP(d) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(b0));
#define CVT(d) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(b0), "v"(b1));
#define FMA(d) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(b0), "v"(b1), "v"(b0));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters, float seed) {
  float b0 = seed * threadIdx.x, b1 = b0 + 1.f;
  float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  f32x16 acc0 = {}, acc1 = {};
  bf16x8 x, y;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(seed * (threadIdx.x + i)); y[i] = (__bf16)(seed * (threadIdx.x * 3 + i)); }   // non-trivial operands (toggling bits cost power)
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { REP4(MF(acc0) MF(acc1)) }
    if (MODE == 1) { REP4(MF(acc0) EXP(c0) EXP(c2) CVT(c1) MF(acc1) EXP(c0) EXP(c2) CVT(c3)) }
    if (MODE == 2) { REP4(EXP(c0) EXP(c2) CVT(c1) FMA(c3) EXP(c0) EXP(c2) CVT(c3) FMA(c1)) }
    if (MODE == 3) { REP4(FMA(c0) FMA(c1) FMA(c2) FMA(c3) FMA(c0) FMA(c1) FMA(c2) FMA(c3)) }
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = c0 + c1 + c2 + c3;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int threads, int iters) {
  const int blocks = 256;
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * blocks * threads);
  (void)hipMalloc(&cyc, sizeof(long long) * blocks * 2);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters / 10, 0.001f);
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.001f);
  (void)hipEventRecord(b, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(blocks * 2);
  (void)hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double st = 0, sr = 0;
  for (int i = 0; i < blocks; ++i) { st += h[2 * i]; sr += h[2 * i + 1]; }
  printf("%-44s waves/SIMD %d: %7.2f ms, s_memtime %6.1f ticks per group, s_memtime / s_memrealtime = %6.2f (x 100 MHz = shader clock if s_memtime counts shader cycles)\n", name, threads / 256, ms,
         st / blocks / iters / 8, st / sr);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<3>("v_fma_f32 only", threads, 400000);
    run<0>("MFMA 32x32x16 bf16 only", threads, 100000);
    run<1>("MFMA + 2 exp + 1 cvt", threads, 100000);
    run<2>("exp / cvt / fma, no MFMA", threads, 100000);
  }
  return 0;
}
