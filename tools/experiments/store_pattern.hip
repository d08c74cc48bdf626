// Write-throughput of the igemm epilogue's store pattern vs wider contiguous segments (experiment, not product code).
// hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// rows x 320 channels (640 B pitch).  Each wave writes a 64 px x 80 ch sub-tile like the epilogue:
// mode 0: lane (fr=lane&15 -> pixel, fq=lane>>4 -> 4 channels)  8-byte stores, 5 j-blocks x 4 pixel groups
// mode 1: 16-byte stores, 4 lanes cover 64 contiguous bytes of a pixel
// mode 2: 16-byte stores, 8 lanes cover 128 contiguous bytes, rest of the wave = other pixels (needs 160 B = 1.25 lines: use 128 of them)
template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned short* out, int rows, int N, int resident) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = N / 160;
  const int total = (rows / 256) * tiles_n;
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    // resident: every block rewrites its own 256-row stripe (stays in L2): store ISSUE rate, not HBM
    const int tt = resident ? (int)blockIdx.x * tiles_n + (t % tiles_n) : t;
    const int m0 = (tt / tiles_n) * 256 + wm * 64, n0 = (tt % tiles_n) * 160 + wn * 80;
    if (MODE == 0) {
      const int fr = lane & 15, fq = lane >> 4;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 5; ++j) {
          u32x2 v = {(unsigned)t, (unsigned)lane};
          *reinterpret_cast<u32x2*>(out + (size_t)(m0 + i * 16 + fr) * N + n0 + j * 16 + fq * 4) = v;
        }
    } else if (MODE == 1) {
      const int fr = lane & 15, fq = lane >> 4;   // 4 lanes x 16 B = 64 B contiguous (channels 0..31 of a 32-block)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 5; j += 2) {
          u32x4 v = {(unsigned)t, (unsigned)lane, 0u, 1u};
          if (j < 4) *reinterpret_cast<u32x4*>(out + (size_t)(m0 + i * 16 + fr) * N + n0 + j * 16 + fq * 8) = v;
          else if (fq < 2) *reinterpret_cast<u32x4*>(out + (size_t)(m0 + i * 16 + fr) * N + n0 + j * 16 + fq * 8) = v;
        }
    } else {
      // 10 lanes x 16 B = 160 B contiguous per pixel; 6 pixels per instruction (60 lanes), 64 pixels -> 11 instructions
      const int px = lane / 10, part = lane % 10;
      for (int i = 0; i < 11; ++i) {
        const int p = i * 6 + px;
        u32x4 v = {(unsigned)t, (unsigned)lane, 0u, 1u};
        if (lane < 60 && p < 64) *reinterpret_cast<u32x4*>(out + (size_t)(m0 + p) * N + n0 + part * 8) = v;
      }
    }
  }
}

int main() {
  const int rows = 128 * 4096;
  for (int resident = 0; resident < 2; ++resident)
  for (int N : {320, 960}) {
    unsigned short* d;
    hipMalloc(&d, (size_t)rows * N * 2);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9;
      for (int it = 0; it < 5; ++it) {
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d, rows, N, resident);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d, rows, N, resident);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d, rows, N, resident);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      printf("resident=%d N=%d mode %d: %.3f ms  %.0f GB/s\n", resident, N, mode, best, (double)rows * N * 2 / best / 1e6);
    }
    hipFree(d);
  }
  return 0;
}
