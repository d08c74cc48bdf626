// Per-CU store issue rate vs access shape, L2-resident footprint (experiment, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every block rewrites its own 80 KiB stripe `reps` times; 8 waves
template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned char* out, int reps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned char* base = out + (size_t)blockIdx.x * 81920 + wid * 10240;   // 10 KiB per wave
  for (int r = 0; r < reps; ++r) {
    asm volatile("" ::: "memory");
    if (MODE == 0) {          // 16 B / lane, 1 KiB contiguous per instruction
      for (int i = 0; i < 10; ++i) *reinterpret_cast<u32x4*>(base + i * 1024 + lane * 16) = (u32x4){(unsigned)r, 1u, 2u, 3u};
    } else if (MODE == 1) {   // 8 B / lane, 512 B contiguous
      for (int i = 0; i < 20; ++i) *reinterpret_cast<u32x2*>(base + i * 512 + lane * 8) = (u32x2){(unsigned)r, 1u};
    } else if (MODE == 2) {   // 4 B / lane, 256 B contiguous
      for (int i = 0; i < 40; ++i) *reinterpret_cast<unsigned*>(base + i * 256 + lane * 4) = (unsigned)r;
    } else if (MODE == 3) {   // 16 B / lane, 4 lanes = 64 B contiguous, 16 segments 640 B apart (epilogue after a 16-lane swap)
      for (int i = 0; i < 10; ++i) *reinterpret_cast<u32x4*>(base + (lane & 15) * 640 + (lane >> 4) * 16 + i * 64) = (u32x4){(unsigned)r, 1u, 2u, 3u};
    } else if (MODE == 4) {   // 8 B / lane, 4 lanes = 32 B contiguous, 16 segments 640 B apart (current epilogue)
      for (int i = 0; i < 20; ++i) *reinterpret_cast<u32x2*>(base + (lane & 15) * 640 + (lane >> 4) * 8 + i * 32) = (u32x2){(unsigned)r, 1u};
    } else if (MODE == 5) {   // 16 B / lane, 8 lanes = 128 B contiguous (one line), 8 segments 640 B apart
      for (int i = 0; i < 10; ++i) *reinterpret_cast<u32x4*>(base + (lane >> 3) * 640 + (lane & 7) * 16 + (i & 3) * 128 + (i >> 2) * 5120) = (u32x4){(unsigned)r, 1u, 2u, 3u};
    } else if (MODE == 6) {   // 16 B / lane, 16 lanes = 256 B contiguous, 4 segments
      for (int i = 0; i < 10; ++i) *reinterpret_cast<u32x4*>(base + (lane >> 4) * 640 + (lane & 15) * 16 + (i & 1) * 256 + ((i >> 1) * 2560) % 10240) = (u32x4){(unsigned)r, 1u, 2u, 3u};
    }
  }
}

int main() {
  unsigned char* d;
  hipMalloc(&d, (size_t)256 * 81920 + 65536);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 200;
  const char* names[] = {"16B/lane 1KiB contiguous", "8B/lane 512B contiguous", "4B/lane 256B contiguous", "16B/lane 64B segments", "8B/lane 32B segments (now)",
                         "16B/lane 128B segments", "16B/lane 256B segments"};
  for (int mode = 0; mode < 7; ++mode) {
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(a);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d, reps); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d, reps); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d, reps); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, d, reps); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, d, reps); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, d, reps); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, d, reps); break;
      }
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    const double bytes = 256.0 * 81920 * reps;
    printf("%-28s %.3f ms  %.0f GB/s  %.1f B/clk/CU @2.2GHz\n", names[mode], best, bytes / best / 1e6, bytes / best / 1e6 / 256 / 2.2);
  }
  return 0;
}
