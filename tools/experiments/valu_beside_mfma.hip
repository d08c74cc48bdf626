// What one wave per SIMD pays for VALU instructions issued between its own MFMAs (gfx950; block of 256 threads per CU, 512 registers per wave).
//   hipcc --offload-arch=gfx950 -O3 -o valu_beside_mfma tools/experiments/valu_beside_mfma.hip && ./valu_beside_mfma
// Each mode: 8 x (one v_mfma_f32_32x32x16_bf16 on alternating accumulators + the listed independent VALU instructions); s_memtime (shader cycles) per MFMA.
// Measured: profiles/r06_attention_persistent.log, section 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP4(x) x x x x
#define MF(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
#define CVT(d) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(b0), "v"(b1));
#define CVTH(d) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(b0), "v"(b1));
#define CVTZ(d) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(d) : "v"(b0), "v"(b1));
#define PERM(d) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b0), "v"(b1), "v"(b2));
#define FMA(d) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(b0), "v"(b1), "v"(b2));
#define EXP(d) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(b0));
#define AND(d) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b0), "v"(b1), "v"(b2));

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float* out, long long* cyc, int iters) {
  float b0 = 0.001f * threadIdx.x, b1 = b0 + 1.f, b2 = b0 + 2.f;
  float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  f32x16 acc0 = {}, acc1 = {};
  bf16x8 x = {}, y = {};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { REP4(MF(acc0) MF(acc1)) }
    if (MODE == 1) { REP4(MF(acc0) CVT(c0) MF(acc1) CVT(c1)) }
    if (MODE == 2) { REP4(MF(acc0) CVT(c0) CVT(c2) MF(acc1) CVT(c1) CVT(c3)) }
    if (MODE == 3) { REP4(MF(acc0) FMA(c0) MF(acc1) FMA(c1)) }
    if (MODE == 4) { REP4(MF(acc0) FMA(c0) FMA(c2) MF(acc1) FMA(c1) FMA(c3)) }
    if (MODE == 5) { REP4(MF(acc0) EXP(c0) EXP(c2) MF(acc1) EXP(c1) EXP(c3)) }
    if (MODE == 6) { REP4(MF(acc0) EXP(c0) EXP(c2) CVT(c1) MF(acc1) EXP(c0) EXP(c2) CVT(c3)) }
    if (MODE == 7) { REP4(MF(acc0) PERM(c0) MF(acc1) PERM(c1)) }
    if (MODE == 8) { REP4(MF(acc0) EXP(c0) EXP(c2) PERM(c1) MF(acc1) EXP(c0) EXP(c2) PERM(c3)) }
    if (MODE == 9) { REP4(MF(acc0) CVTH(c0) MF(acc1) CVTH(c1)) }
    if (MODE == 10) { REP4(MF(acc0) CVTZ(c0) MF(acc1) CVTZ(c1)) }
    if (MODE == 11) { REP4(CVT(c0) CVT(c1) CVT(c2) CVT(c3) CVT(c0) CVT(c1) CVT(c2) CVT(c3)) }
    if (MODE == 12) { REP4(PERM(c0) PERM(c1) PERM(c2) PERM(c3) PERM(c0) PERM(c1) PERM(c2) PERM(c3)) }
    if (MODE == 13) { REP4(MF(acc0) EXP(c0) EXP(c2) EXP(c1) MF(acc1) EXP(c0) EXP(c2) EXP(c3)) }
    if (MODE == 14) { REP4(MF(acc0) AND(c0) MF(acc1) AND(c1)) }
    if (MODE == 15) { REP4(MF(acc0) EXP(c0) EXP(c2) FMA(c1) MF(acc1) EXP(c0) EXP(c2) FMA(c3)) }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0 + c1 + c2 + c3;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
  const int blocks = 256, iters = 2000;
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&cyc, sizeof(long long) * blocks * 4);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += v;
  printf("%-58s %7.2f cycles per MFMA (modes without an MFMA: per 8 instructions)\n", name, sum / h.size() / iters / 8);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0>("MFMA (two accumulators alternate)");
  run<1>("MFMA + 1 v_cvt_pk_bf16_f32");
  run<2>("MFMA + 2 v_cvt_pk_bf16_f32");
  run<9>("MFMA + 1 v_cvt_pk_f16_f32");
  run<10>("MFMA + 1 v_cvt_pkrtz_f16_f32");
  run<3>("MFMA + 1 v_fma_f32");
  run<4>("MFMA + 2 v_fma_f32");
  run<7>("MFMA + 1 v_perm_b32");
  run<14>("MFMA + 1 v_and_or_b32");
  run<5>("MFMA + 2 v_exp_f32");
  run<13>("MFMA + 3 v_exp_f32");
  run<6>("MFMA + 2 v_exp_f32 + 1 v_cvt_pk_bf16_f32");
  run<8>("MFMA + 2 v_exp_f32 + 1 v_perm_b32");
  run<15>("MFMA + 2 v_exp_f32 + 1 v_fma_f32");
  run<11>("v_cvt_pk_bf16_f32 alone (per instruction)");
  run<12>("v_perm_b32 alone (per instruction)");
  return 0;
}
