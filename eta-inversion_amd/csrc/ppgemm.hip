// Ping-pong implicit GEMM for gfx950: the 256 x 160 x 64 tile of igemm.hip's ring with its two wave groups in ANTI-PHASE.
//
// Why (round 5): the ring kernel runs all eight waves of a block through the same program in lockstep -- both waves of every SIMD reach their matrix
// clusters, their LDS read bursts and the K-step rendezvous together.  In-kernel stamps (profiles/r04_conv_patch_stamps_after.log) put ~600 of the
// ~2370 cycles of a K step in that rendezvous with the matrix pipe idle; SQ counters (profiles/r05_pmc_sq_rows128.json) read 0.50 MFMA-busy for the conv
// instantiation and 0.23-0.37 for the short-K ones.  Here waves 0-3 (group E, one per SIMD) and waves 4-7 (group L) run the same program ONE PHASE apart:
//
//   phase of a wave = one 32-deep k-half of a K step:   MEM: 9 ds_read_b128 (its fragments of that half) + its share of the LDS-DMA issue
//                                                        s_barrier
//                                                        COMPUTE: s_setprio 1, 20 MFMAs 16x16x32 on the fragments, s_setprio 0
//                                                        s_barrier
//   group L executes one extra s_barrier up front, so in every barrier interval one group computes while the other reads / issues: each SIMD's
//   matrix pipe always has exactly one wave feeding it, and the memory instructions of a phase are issued by four waves at once under the other four
//   waves' MFMAs (MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md "The 256^2 8-phase template").
//
// A wave needs ONE fragment set (36 VGPRs instead of the ring's 72: reads never overlap the wave's own MFMAs), which leaves room for a SECOND
// accumulator set: the epilogue of tile t (bias / residual / convert / stores) is cut into slices that run in the MEM phases of tile t+1 while the
// accumulators of t+1 fill -- the short-K GEMMs of the transformer blocks (K = 320: five K steps) no longer stop the matrix pipe for an epilogue as
// long as their main loop.
//
// LDS: 3-slot ring of [256 + 160 rows][64] K tiles exactly as in igemm.hip (row = 8 chunks of 16 B, physical chunk = chunk ^ (row & 7), filled
// lane-linearly by global_load_lds_dwordx4 with the swizzle on the SOURCE chunk).  Hazards, in barrier intervals (group E reads half h of step s in
// interval 4 s + 2 h, group L one interval later; a read issued in interval i has completed before the barrier that ends interval i + 1):
//   * K tile s+2 goes into the slot of step s-1, whose last reads are L's in interval 4 s - 1: its first pieces are issued in a wave's MEM phase of
//     (s, half 1) = interval 4 s + 2 / 4 s + 3, the rest in MEM (s + 1, half 0)                       -> WAR distance >= 3 intervals;
//   * every wave waits (counted vmcnt) for its own pieces of step s+1 at the end of its MEM phase of (s, half 1); the barrier behind that wait
//     precedes the first read of step s+1 (E: interval 4 s + 4)                                      -> RAW: wait, barrier, then read.
// Operand / accumulator conventions are igemm.hip's (weights = MFMA A operand, activations = B: a lane holds 4 consecutive output channels of one
// pixel row), and so is the accumulation order over K: results are bit-identical to the ring kernel's.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace etainv {

namespace {

template <typename T> struct PMfma;
template <> struct PMfma<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct PMfma<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

#define PP_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
#define PP_LGKMCNT0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define PP_GLDS(src, dst, bytes) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), bytes, 0, 0)

constexpr int PBM = 256, PBN = 160, PBK = 64;
constexpr int PMT = 4, PNT = 5;                 // wave tile 64 x 80 (4 x 2 waves)
constexpr int PSLOT_A = PBM * PBK, PSLOT_B = PBN * PBK;

template <typename T>
__global__ void __launch_bounds__(512, 2) pp_gemm_kernel(IGemmParams p) {
  typedef typename PMfma<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sA = reinterpret_cast<T*>(smem);              // [3][256][64]
  T* sB = sA + 3 * PSLOT_A;                        // [3][160][64]
  T* dummy = sB + 3 * PSLOT_B;                     // 1 KiB: target of the pieces of waves that have no row in the last, partial B pass
  float* sBias = reinterpret_cast<float*>(dummy + 512);   // [4][160] bias of the tiles in flight

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const bool late = wid >= 4;                       // group L

  const int K = p.c1;
  const int nk = K / PBK;
  const int tiles_n = p.N / PBN;
  const int total_tiles = (p.M / PBM) * tiles_n;
  const int G = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
  if (my_tiles <= 0) return;
  auto tile_origin = [&](int i, int& m0, int& n0) {
    int v = blockIdx.x + i * G;
    if ((total_tiles & 7) == 0) v = (v & 7) * (total_tiles >> 3) + (v >> 3);   // XCD-contiguous tile runs, n fastest (as in igemm.hip)
    const int tm = v / tiles_n;
    m0 = tm * PBM;
    n0 = (v - tm * tiles_n) * PBN;
  };
  const int total_steps = my_tiles * nk;

  // ---- issue side: the (tile, K tile) position whose pieces go out next; row r = (tid >> 3) + 64 q of a tile, lane chunk swizzled at the source
  const int lchunk = (tid & 7) ^ ((tid >> 3) & 7);
  const int wrow0 = wid * 8;
  const T* a_row[4];
  const T* w_row[3];
  int it_tile = 0, it_kt = 0, it_step = 0;          // it_step: flattened index of the K tile at the issue position
  auto setup_issue = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(tile, m0, n0);
#pragma unroll
    for (int q = 0; q < 4; ++q) a_row[q] = reinterpret_cast<const T*>(p.a1) + (int64_t)(m0 + (tid >> 3) + 64 * q) * K + lchunk * 8;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      int n = n0 + (tid >> 3) + 64 * q;
      n = n < p.N ? n : p.N - 1;
      w_row[q] = reinterpret_cast<const T*>(p.w) + (int64_t)n * K + lchunk * 8;
    }
    if (p.bias && wid < 3) {                        // 160 floats: waves 0, 1 whole, wave 2 its first 32 lanes
      const int c = wid * 64 + lane;
      if (c < PBN) PP_GLDS(p.bias + n0 + c, sBias + (tile & 3) * PBN + wid * 64, 4);
    }
  };
  auto issue_a = [&](int q0, int q1) __attribute__((always_inline)) {
    T* dA = sA + (it_step % 3) * PSLOT_A;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q >= q0 && q < q1) PP_GLDS(a_row[q] + it_kt * PBK, dA + (wrow0 + 64 * q) * PBK, 16);
  };
  auto issue_b = [&]() __attribute__((always_inline)) {
    T* dB = sB + (it_step % 3) * PSLOT_B;
    PP_GLDS(w_row[0] + it_kt * PBK, dB + wrow0 * PBK, 16);
    PP_GLDS(w_row[1] + it_kt * PBK, dB + (wrow0 + 64) * PBK, 16);
    PP_GLDS(w_row[2] + it_kt * PBK, wid < 4 ? dB + (wrow0 + 128) * PBK : dummy, 16);   // rows 128 .. 159: waves 0-3; the others aim at the dummy area (uniform counts)
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++it_step;
    if (++it_kt == nk) {
      it_kt = 0;
      if (++it_tile < my_tiles) setup_issue(it_tile);
    }
  };

  f32x4 acc[PMT][PNT];
#pragma unroll
  for (int i = 0; i < PMT; ++i)
#pragma unroll
    for (int j = 0; j < PNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fa[PMT], fb[PNT];

  auto read_frags = [&](int slot, int kk) __attribute__((always_inline)) {
    const T* tA = sA + slot * PSLOT_A;
    const T* tB = sB + slot * PSLOT_B;
#pragma unroll
    for (int i = 0; i < PMT; ++i) {
      const int row = wm * 64 + i * 16 + fr;
      fa[i] = *reinterpret_cast<const u32x4*>(tA + row * PBK + (((kk * 4 + fq) ^ (row & 7)) << 3));
    }
#pragma unroll
    for (int j = 0; j < PNT; ++j) {
      const int row = wn * 80 + j * 16 + fr;
      fb[j] = *reinterpret_cast<const u32x4*>(tB + row * PBK + (((kk * 4 + fq) ^ (row & 7)) << 3));
    }
  };
  auto compute = [&]() __attribute__((always_inline)) {
    PP_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < PMT; ++i)
#pragma unroll
      for (int j = 0; j < PNT; ++j) acc[i][j] = PMfma<T>::run(__builtin_bit_cast(frag, fb[j]), __builtin_bit_cast(frag, fa[i]), acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  T* out = reinterpret_cast<T*>(p.out);
  const T* res = reinterpret_cast<const T*>(p.residual);
  auto epilogue = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(tile, m0, n0);
    const float* tb = sBias + (tile & 3) * PBN + wn * 80;
    f32x4 bv[PNT];
#pragma unroll
    for (int j = 0; j < PNT; ++j) bv[j] = p.bias ? *reinterpret_cast<const f32x4*>(tb + j * 16 + fq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x2 rv[PMT][PNT];
    if (res) {
#pragma unroll
      for (int i = 0; i < PMT; ++i)
#pragma unroll
        for (int j = 0; j < PNT; ++j)
          rv[i][j] = *reinterpret_cast<const u32x2*>(res + (int64_t)(m0 + wm * 64 + i * 16 + fr) * p.N + n0 + wn * 80 + j * 16 + fq * 4);
    }
#pragma unroll
    for (int i = 0; i < PMT; ++i) {
      T* prow = out + (int64_t)(m0 + wm * 64 + i * 16 + fr) * p.N + n0 + wn * 80;
      u32x2 po[PNT];
#pragma unroll
      for (int j = 0; j < PNT; ++j) {
        f32x4 v = acc[i][j] + bv[j];
        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (res) {
          T r[4];
          *reinterpret_cast<u32x2*>(r) = rv[i][j];
          v[0] += to_f32(r[0]); v[1] += to_f32(r[1]); v[2] += to_f32(r[2]); v[3] += to_f32(r[3]);
        }
        T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
        po[j] = *reinterpret_cast<u32x2*>(o);
      }
      // 16-byte stores after a lane swap between adjacent 16-column blocks (igemm.hip store_row_group): 64-byte segments per pixel row
#pragma unroll
      for (int k = 0; k + 1 < PNT; k += 2) {
        const auto lo = __builtin_amdgcn_permlane16_swap(po[k][0], po[k + 1][0], false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(po[k][1], po[k + 1][1], false, false);
        const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
        *reinterpret_cast<u32x4*>(prow + (k + (fq & 1)) * 16 + (fq >> 1) * 8) = v;
      }
      *reinterpret_cast<u32x2*>(prow + (PNT - 1) * 16 + fq * 4) = po[PNT - 1];
    }
  };

  // ---- prologue: K tiles 0 and 1 whole; wait for tile 0
  setup_issue(0);
  issue_a(0, 4); issue_b(); advance();
  if (total_steps > 1) { issue_a(0, 4); issue_b(); advance(); PP_VMCNT(7); } else { PP_VMCNT(0); }
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();           // group L runs one interval behind

  int slot = 0, ct_kt = 0, ct_tile = 0;
  for (int s = 0; s < total_steps; ++s) {
    // ---- (s, half 0): MEM
    read_frags(slot, 0);
    // rest of the K tile whose first activation pieces went out in the previous step's half 1 (none for s == 0: the prologue issued K tile 1 whole)
    if (s > 0 && it_step < total_steps) { issue_a(2, 4); issue_b(); advance(); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    compute();
    __builtin_amdgcn_s_barrier();
    // ---- (s, half 1): MEM
    read_frags(slot, 1);
    if (it_step < total_steps) {                    // first pieces of K tile s + 2 -> slot of step s - 1
      issue_a(0, 2);
      PP_VMCNT(2);                                  // everything of K tile s + 1 has landed (for this wave); the two pieces just issued stay in flight
    } else {
      PP_VMCNT(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    compute();
    __builtin_amdgcn_s_barrier();
    slot = slot == 2 ? 0 : slot + 1;
    if (++ct_kt == nk) {
      epilogue(ct_tile);
      ct_kt = 0;
      ++ct_tile;
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();
}

}  // namespace

// plain 1x1 / Linear launches on whole tiles (prototype scope of the ping-pong kernel; ETAINV_PP=1)
bool pp_gemm_applicable(const IGemmParams& p, int dtype) {
  if (!env_on("ETAINV_PP") || (dtype != ETAINV_F16 && dtype != ETAINV_BF16)) return false;
  if (p.taps != 1 || p.a2 || p.geglu || p.rowvec || p.out_f32 || p.out_nchw || p.stat_out || p.ln_stat || p.w_batch_stride || p.ksplit > 1 || p.hm_heads) return false;
  if (p.M % PBM != 0 || p.N % PBN != 0 || p.c1 % PBK != 0 || p.c1 < 2 * PBK) return false;
  return (int64_t)(p.M / PBM) * (p.N / PBN) >= 192;
}

int launch_pp_gemm(const IGemmParams& p, int dtype, hipStream_t s) {
  const size_t lds = (size_t)3 * (PSLOT_A + PSLOT_B) * 2 + 1024 + 4 * PBN * sizeof(float);
  const int tiles = (p.M / PBM) * (p.N / PBN);
  const int grid = std::min(tiles, 256);
  static bool attr_set[kMaxDevices][2] = {};
  const int dev = current_device();
  ETAINV_DISPATCH_HALF(dtype, T, {
    const int di = dtype == ETAINV_F16 ? 0 : 1;
    if (!attr_set[dev][di]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_gemm_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set[dev][di] = true;
    }
    hipLaunchKernelGGL(pp_gemm_kernel<T>, dim3(grid), dim3(512), lds, s, p);
  });
  ETAINV_LAUNCH_CHECK();
  return 0;
}

}  // namespace etainv
