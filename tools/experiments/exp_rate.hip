// Issue / throughput cost of v_exp_f32 on gfx950, alone and beside MFMAs, with 1 or 2 waves per SIMD (one block of 256 / 512 threads per CU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/exp_rate tools/experiments/exp_rate.hip && tools/experiments/exp_rate
// Prints cycles (s_memtime) per instruction for: v_exp_f32, v_fma_f32, v_cvt_pk_bf16_f32, v_max3_f32, v_mfma_f32_32x32x16_bf16, and mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 0.001f * (threadIdx.x + i);
  f32x16 acc0 = {}, acc1 = {};
  bf16x8 x = {}, y = {};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // 32 independent v_exp_f32 (8 registers round-robin)
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));)
    } else if (MODE == 1) {   // 32 v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));)
    } else if (MODE == 2) {   // 32 v_cvt_pk_bf16_f32
      REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %1, %2\n v_cvt_pk_bf16_f32 %4, %1, %2\n v_cvt_pk_bf16_f32 %5, %1, %2" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));)
    } else if (MODE == 3) {   // 32 v_max3_f32
      REP8(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[4]), "v"(a[5]));)
    } else if (MODE == 4) {   // 8 MFMAs, two accumulators
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);)
    } else if (MODE == 5) {   // per MFMA: 4 v_exp + 2 v_cvt_pk (the P V phase of the attention kernel), 8 MFMAs
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);
           asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_cvt_pk_bf16_f32 %4, %0, %1\n v_cvt_pk_bf16_f32 %5, %2, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));)
    } else if (MODE == 6) {   // per MFMA: 2 v_exp + 1 v_cvt_pk
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);
           asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_cvt_pk_bf16_f32 %2, %0, %1" : "+v"(a[0]), "+v"(a[1]), "+v"(a[4]));)
    } else if (MODE == 7) {   // 32 v_exp_f16 (is the 16-bit form faster?)
      REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));)
    } else if (MODE == 8) {   // exp interleaved 1:1 with fma (do transcendentals run beside plain VALU?)
      REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %4, %4\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %5, %5\n v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
    } else if (MODE == 10) {   // waves 0-3: 8 MFMAs; waves 4-7 (the SIMD partners): 32 v_exp + 16 v_cvt_pk
      if (threadIdx.x < 256) {
        REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);)
      } else {
        REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_cvt_pk_bf16_f32 %4, %0, %1\n v_cvt_pk_bf16_f32 %5, %2, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));)
      }
    } else if (MODE == 11) {   // waves 0-3: 8 MFMAs; waves 4-7: 32 v_fma
      if (threadIdx.x < 256) {
        REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);)
      } else {
        REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));)
      }
    } else if (MODE == 12) {   // both waves: 12 MFMAs (two accumulators), then 32 v_exp + 16 v_cvt_pk + 8 MFMAs interleaved; the partner starts half an iteration later
      if (threadIdx.x >= 256 && it == 0) {
        REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_cvt_pk_bf16_f32 %4, %0, %1\n v_cvt_pk_bf16_f32 %5, %2, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));)
      }
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);)
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc1, 0, 0, 0);
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);
           asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_cvt_pk_bf16_f32 %4, %0, %1\n v_cvt_pk_bf16_f32 %5, %2, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));)
    } else if (MODE == 13 || MODE == 14) {   // MODE 12 with the accumulators pinned: 13 = VGPRs ("+v"), 14 = AGPRs ("+a")
#define MF_V(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
#define MF_A(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
#define VALU6 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_cvt_pk_bf16_f32 %4, %0, %1\n v_cvt_pk_bf16_f32 %5, %2, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
      if (threadIdx.x >= 256 && it == 0) { REP8(VALU6) }
      if (MODE == 13) {
        REP8(MF_V(acc0)) MF_V(acc1) MF_V(acc1) MF_V(acc1) MF_V(acc1)
        REP8(MF_V(acc0) VALU6)
      } else {
        REP8(MF_A(acc0)) MF_A(acc1) MF_A(acc1) MF_A(acc1) MF_A(acc1)
        REP8(MF_A(acc0) VALU6)
      }
    } else if (MODE >= 20 && MODE <= 24) {   // INDEPENDENT fillers (sources never written): per MFMA  MODE 20: 2 exp + 1 cvt; 21: 3 exp + 1 cvt; 22: 2 exp + 1 cvt + 1 max3; 23: 9 exp + 4 cvt per 4 MFMAs (the d = 40 ratio); 24: 32 independent exps, no MFMA
#define EXI(d, s_) "v_exp_f32 %" #d ", %" #s_ "\n"
      float b0 = a[0], b1 = a[1], b2 = a[2], b3 = a[3], b4 = a[4], b5 = a[5], b6 = a[6], b7 = a[7];
      float c0, c1, c2, c3, c4, c5, c6, c7, c8;
      if (MODE == 20) {
        REP8(MF_V(acc0) asm volatile("v_exp_f32 %0, %3\n v_exp_f32 %1, %4\n v_cvt_pk_bf16_f32 %2, %5, %6" : "=v"(c0), "=v"(c1), "=v"(c2) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
      } else if (MODE == 21) {
        REP8(MF_V(acc0) asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %5\n v_exp_f32 %2, %6\n v_cvt_pk_bf16_f32 %3, %7, %4" : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
      } else if (MODE == 22) {
        REP8(MF_V(acc0) asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %5\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_max3_f32 %3, %4, %5, %6" : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
      } else if (MODE == 23) {
        REP8(MF_V(acc0) asm volatile("v_exp_f32 %0, %3\n v_exp_f32 %1, %4\n v_cvt_pk_bf16_f32 %2, %5, %6" : "=v"(c0), "=v"(c1), "=v"(c2) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
             MF_V(acc1) asm volatile("v_exp_f32 %0, %3\n v_exp_f32 %1, %4\n v_cvt_pk_bf16_f32 %2, %5, %6" : "=v"(c3), "=v"(c4), "=v"(c5) : "v"(b4), "v"(b5), "v"(b6), "v"(b7));
             MF_V(acc0) asm volatile("v_exp_f32 %0, %3\n v_exp_f32 %1, %4\n v_cvt_pk_bf16_f32 %2, %5, %6" : "=v"(c0), "=v"(c1), "=v"(c2) : "v"(b1), "v"(b2), "v"(b3), "v"(b4));
             MF_V(acc1) asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %5\n v_exp_f32 %2, %6\n v_cvt_pk_bf16_f32 %3, %7, %4" : "=v"(c5), "=v"(c6), "=v"(c7), "=v"(c8) : "v"(b5), "v"(b6), "v"(b7), "v"(b0));)
      } else {
        REP8(asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %5\n v_exp_f32 %2, %6\n v_exp_f32 %3, %7" : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
      }
      a[0] += 0.f;
    } else if (MODE == 25 || MODE == 26) {   // the P V chain of the attention tile: 8 exp -> 4 cvt_pk -> the B operand of 2 MFMAs; 25: in that order, 26: the next group's exps between the converts and the MFMAs
      float b0 = a[0], b1 = a[1], b2 = a[2], b3 = a[3], b4 = a[4], b5 = a[5], b6 = a[6], b7 = a[7];
      float e0, e1, e2, e3, e4, e5, e6, e7;
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      u4 pk;
#define EXP8 asm volatile("v_exp_f32 %0, %8\n v_exp_f32 %1, %9\n v_exp_f32 %2, %10\n v_exp_f32 %3, %11\n v_exp_f32 %4, %12\n v_exp_f32 %5, %13\n v_exp_f32 %6, %14\n v_exp_f32 %7, %15" \
                          : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(e4), "=&v"(e5), "=&v"(e6), "=&v"(e7) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
#define CVT4 asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %6, %7\n v_cvt_pk_bf16_f32 %2, %8, %9\n v_cvt_pk_bf16_f32 %3, %10, %11" \
                          : "=&v"(pk[0]), "=&v"(pk[1]), "=&v"(pk[2]), "=&v"(pk[3]) : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(e4), "v"(e5), "v"(e6), "v"(e7));
#define MF2P asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(acc0), "+v"(acc1) : "v"(y), "v"(pk));
      if (MODE == 25) {
        EXP8 CVT4 MF2P  EXP8 CVT4 MF2P  EXP8 CVT4 MF2P  EXP8 CVT4 MF2P
      } else {
        if (it == 0) { EXP8 }
        CVT4 EXP8 MF2P  CVT4 EXP8 MF2P  CVT4 EXP8 MF2P  CVT4 EXP8 MF2P
      }
      a[0] += e0 * 0.f;
    } else if (MODE == 9) {   // per MFMA: 1 v_exp + 6 plain VALU
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);
           asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, int threads) {
  const int blocks = 256, iters = 2000;
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double sum = 0, g0 = 0, g1 = 0;
  const int wpb = threads / 64;
  for (size_t i = 0; i < h.size(); ++i) {
    sum += h[i];
    if ((int)(i % wpb) < wpb / 2 || wpb == 4) g0 += h[i]; else g1 += h[i];
  }

  const double per_wave = sum / h.size() / iters;
  printf("%-46s waves/SIMD %d: %8.1f cycles per iteration and wave, %6.2f per instruction-slot (%d per iteration); per SIMD %6.2f\n", name, threads / 256, per_wave,
         per_wave / per_iter, per_iter, per_wave / per_iter / (threads / 256));
  if (wpb == 8) printf("    [waves 0-3: %.1f cycles per iteration, waves 4-7: %.1f]\n", g0 / (h.size() / 2) / iters, g1 / (h.size() / 2) / iters);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("32 v_exp_f32", 32, threads);
    run<7>("32 v_exp_f16", 32, threads);
    run<1>("32 v_fma_f32", 32, threads);
    run<2>("32 v_cvt_pk_bf16_f32", 32, threads);
    run<3>("32 v_max3_f32", 32, threads);
    run<8>("32 v_exp_f32 + 32 v_fma_f32 interleaved", 64, threads);
    run<4>("8 v_mfma_f32_32x32x16_bf16", 8, threads);
    run<5>("8 x (MFMA + 4 v_exp + 2 v_cvt_pk)", 8, threads);
    run<6>("8 x (MFMA + 2 v_exp + 1 v_cvt_pk)", 8, threads);
    run<9>("8 x (MFMA + 1 v_exp + 6 v_fma)", 8, threads);
    run<24>("32 INDEPENDENT v_exp_f32", 32, threads);
    run<20>("8 x (MFMA + 2 exp + 1 cvt), independent", 8, threads);
    run<21>("8 x (MFMA + 3 exp + 1 cvt), independent", 8, threads);
    run<22>("8 x (MFMA + 2 exp + 1 cvt + 1 max3), independent", 8, threads);
    run<23>("32 x (MFMA + 2.25 exp + 1 cvt), independent (d = 40 ratio)", 32, threads);
    run<25>("8 MFMAs fed by 32 exp -> 16 cvt (P V chain, program order)", 8, threads);
    run<26>("8 MFMAs fed by 32 exp -> 16 cvt (next group's exps in front of the MFMAs)", 8, threads);
    if (threads == 512) {
      run<10>("waves 0-3: 8 MFMA | waves 4-7: 32 exp + 16 cvt", 1, threads);
      run<11>("waves 0-3: 8 MFMA | waves 4-7: 32 fma", 1, threads);
      run<12>("12 MFMA, then 8 x (MFMA + 4 exp + 2 cvt); partner half a period late", 1, threads);
      run<13>("the same, accumulators in VGPRs", 1, threads);
      run<14>("the same, accumulators in AGPRs", 1, threads);
    }
  }
  return 0;
}
