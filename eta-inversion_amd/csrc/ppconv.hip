// Ping-pong conv3x3 (stride 1, pad 1) in PATCH form for gfx950: igemm.hip's PATCH-mode ring -- an M tile is a 16 x 16 pixel patch of one image whose halo'd
// 18 x 18 activation patch is brought to LDS once per 64-channel chunk and serves all nine taps -- with the two wave groups of the block in ANTI-PHASE
// (ppgemm.hip) and an issue side without address arithmetic in the loop.
//
// Why (round 5): the ablations of the ping-pong GEMM (profiles/r05_pp_ablation.log) showed that the structure issues MFMAs 96 % of the time when it is fed
// and that kernels on the 256 x 160 tile are bound by the LDS-DMA fill rate once a K step needs more than ~30 KB; a PATCH-mode K step needs 24.6 KB (20 KB
// of weights + 41 / 9 patch pieces), so the conv is the kernel class that can run near the matrix pipe's own limit.  The lockstep ring reaches 1.10-1.28
// PFLOP/s on it (0.50 MFMA-busy in the SQ counters); the dual-N GEMM, same structure as here, reaches 1.30-1.40 on its deep-K shapes.
//
// K loop: chunk major -- (64-channel chunk q, tap) -- one K step = two phases (k-halves) per wave:
//     MEM: the phase's LDS-DMA pieces, 9 ds_read_b128 (4 patch rows at the tap's shift + 5 weight blocks), lgkmcnt(0) | s_barrier | 20 MFMAs | s_barrier
//   waves 4-7 one barrier interval behind waves 0-3.  Issue position = 2 K steps ahead of the compute position:
//     weights of K step s + 2 -> slot (s + 2) % 3 (last read in step s - 1): 64-row passes 0, 1 in MEM (s, half 0), pass 2 in MEM (s, half 1);
//     patch of the NEXT chunk -> patch slot (chunk + 1) & 1: piece k (rows 8 (8 k + wave) .. + 7 of the 18 x 18 patch) in MEM (s, half 1) of the steps whose
//       issue position is tap k + 2, k = 0 .. 5 -- compute taps 0 .. 5 of the current chunk, i.e. after the last read of the slot's previous patch; a
//       dummy piece (zero page -> LDS dummy area) otherwise: every wave issues the same number of pieces per step.
//   MEM (s, half 1) opens with vmcnt(2): everything but the two weight pieces of MEM (s, half 0) has landed -- K step s + 1's weights and every patch
//   piece issued so far -- and the barrier that ends the phase precedes their first read.
// Addresses: weights = scalar base (tile, chunk, tap) + a per-lane byte offset constant for the launch; patch piece k = per-lane 64-bit address built from
// a per-lane pixel offset constant for the launch (pr * W + pc of the lane's patch row) x the channel count + the chunk, or the zero page for halo rows
// outside the image, computed once per chunk.  The nine taps of a chunk are unrolled: tap shifts, patch piece numbers and weight offsets are compile-time
// constants (a tap-indexed table would live in scratch memory -- its loads drain the DMA queue: measured 2x slower).
// Epilogue = igemm.hip's fast path for PATCH tiles: bias + time-embedding row + residual, 16-byte stores after a lane swap, GroupNorm channel statistics
// of the stored values per 64-row wave tile (stat_kind 1).  Accumulation order over K is the ring's PATCH order: results are bit-identical.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace etainv {

namespace {

template <typename T> struct CMfma;
template <> struct CMfma<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct CMfma<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

#define PC_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
#define PC_LGKMCNT0() __builtin_amdgcn_s_waitcnt(0xC07F)
// LDS-DMA by inline asm (no register destination; M0 written in the statement that uses it, nothing else in the kernel uses M0): SGPR-base form and
// per-lane 64-bit address form
#define PC_DMA_S(voff32, sbase, ldsaddr) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff32), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(ldsaddr)) : "memory")
#define PC_DMA_V(vaddr64, ldsaddr) \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(vaddr64), "s"(__builtin_amdgcn_readfirstlane(ldsaddr)) : "memory")
// buffer form: 32-bit per-lane byte offset into the descriptor's range; an offset >= num_records (the halo outside the image) reads zeros
#define PC_DMA_B(voff32, rsrc, ldsaddr) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff32), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(ldsaddr)) : "memory")
#define PC_DMA_S4(voff32, sbase, ldsaddr) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff32), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(ldsaddr)) : "memory")

constexpr int CBN = 160, CBK = 64, CMT = 4, CNT = 5;
constexpr int CPW = 18;                               // patch pitch (16 + halo)
constexpr int CPROWS = 328;                           // 324 patch rows rounded up to whole 8-row pieces
constexpr int CPATCH_BYTES = CPROWS * CBK * 2, CW_BYTES = CBN * CBK * 2;
constexpr int COFF_W = 2 * CPATCH_BYTES, COFF_DUMMY = COFF_W + 3 * CW_BYTES, COFF_BIAS = COFF_DUMMY + 1024, CLDS = COFF_BIAS + 4 * CBN * 4;

// sum over the 16 lanes of a DPP row (igemm.hip row_sum16)
__device__ __forceinline__ float c_row_sum16(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));
  return x;
}

template <typename T, bool GNSTAT>
__global__ void __launch_bounds__(512, 2) pp_conv_kernel(IGemmParams p) {
  typedef typename CMfma<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sBias = reinterpret_cast<float*>(smem + COFF_BIAS);   // [4][160]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const bool late = wid >= 4;

  const int C = p.c1, N = p.N, H = p.H, W = p.W;
  const int kcq = C / CBK;                            // chunks per tile
  const int nk = 9 * kcq;                             // K steps per tile
  const int tpr = W / 16, tpi = (H / 16) * tpr;       // patches per patch row / per image
  const int tiles_n = N / CBN;
  const int total_tiles = (p.M / 256) * tiles_n;
  const int G = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
  if (my_tiles <= 0) return;
  // tile -> (patch index mt, n0); patch -> (image b, origin y0, x0)
  auto tile_origin = [&](int i, int& mt, int& n0) __attribute__((always_inline)) {
    int v = blockIdx.x + i * G;
    if ((total_tiles & 7) == 0) v = (v & 7) * (total_tiles >> 3) + (v >> 3);
    mt = v / tiles_n;
    n0 = (v - mt * tiles_n) * CBN;
  };
  auto patch_origin = [&](int mt, int& b, int& y0, int& x0) __attribute__((always_inline)) {
    b = mt / tpi;
    const int r = mt - b * tpi, ty = r / tpr;
    y0 = ty * 16;
    x0 = (r - ty * tpr) * 16;
  };
  const int total_steps = my_tiles * nk;
  // timing-only ablations (tools/pp_check.py conv), in a variant build only: the run-time flags cost the GroupNorm instantiation its last two registers
  //   VARIANT=ablate VARIANT_FILE=ppconv VARIANT_FLAGS=-DETAINV_PPCONV_ABLATE bash build.sh; ETAINV_LIB=.../libetainv_hip_ablate.so ETAINV_IGEMM_DEBUG=<bits>
#ifdef ETAINV_PPCONV_ABLATE
  const bool no_dma = p.debug & 1, no_epi = p.debug & 2, no_mfma = p.debug & 4, no_reads = p.debug & 8;
#else
  constexpr bool no_dma = false, no_epi = false, no_mfma = false, no_reads = false;
#endif
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- per-lane constants of the issue side
  const unsigned lrow = tid >> 3;
  const unsigned wk = 9u * (unsigned)C;                // weight row length
  unsigned voff_w[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    unsigned r = lrow + 64 * q;
    if (r >= (unsigned)CBN) r = lrow;                  // (pass 2 of waves 4-7: rows past the tile, aimed at the dummy area; any valid row will do)
    voff_w[q] = (r * wk + (((tid & 7) ^ (lrow & 7)) << 3)) * 2u;
  }
  // ---- issue cursor: (tile, chunk) of the K step two steps ahead of the compute position (its tap is a compile-time constant of the unrolled step);
  // `wslot` its weight ring slot, `it_gq` the running chunk count
  int it_tile = 0, it_q = 0, it_cnt = 0, it_gq = 0, wslot = 0;
  const char* w_base = nullptr;                        // weights of the cursor's tile: w + n0 * 9 C
  // the patch being issued = the chunk AFTER the cursor's chunk in the stream: a buffer descriptor over its image (base = the image's first pixel at the
  // chunk's channels, num_records = the image's bytes) and the six pieces' per-lane byte offsets inside it, computed once per chunk; rows outside the image
  // or past the patch carry an offset beyond num_records: the hardware's range check returns zeros for them (no zero page, no 64-bit addresses)
  u32x4 pt_rsrc = {0u, 0u, 0u, 0u};
  unsigned pt_off[6];
  auto set_tile = [&](int tile) __attribute__((always_inline)) {
    int mt, n0;
    tile_origin(tile, mt, n0);
    w_base = reinterpret_cast<const char*>(p.w) + (int64_t)n0 * wk * 2;
    if (p.bias && wid < 3) {
      const char* gb = reinterpret_cast<const char*>(p.bias + n0 + wid * 64);
      const unsigned db = lds0 + COFF_BIAS + ((tile & 3) * CBN + wid * 64) * 4;
      if (wid * 64 + lane < CBN) PC_DMA_S4((unsigned)(lane * 4), gb, db);
    }
  };
  auto set_patch = [&](int tile, int q) __attribute__((always_inline)) {   // (tile, chunk) whose patch is issued next
    const bool tile_ok = tile < my_tiles;
    int mt, n0, b, y0, x0;
    tile_origin(tile_ok ? tile : 0, mt, n0);
    patch_origin(mt, b, y0, x0);
    const uint64_t base = reinterpret_cast<uint64_t>(p.a1) + (uint64_t)(((int64_t)b * H * W * C + q * CBK) * 2);
    pt_rsrc = (u32x4){(unsigned)base, (unsigned)(base >> 32), (unsigned)(H * W * C * 2), 0x00020000u};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      // piece k of this wave: patch rows 8 (8 k + wave) + (lane >> 3) of the 18 x 18 patch
      const int prow = (8 * k + wid) * 8 + (lane >> 3);
      const int pr = (prow * 3641) >> 16, pc = prow - pr * CPW;     // prow / 18 for prow < 1024
      const int y = y0 - 1 + pr, x = x0 - 1 + pc;
      const bool ok = tile_ok & (prow < CPW * CPW) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
      pt_off[k] = ok ? ((unsigned)(y * W + x) * (unsigned)C + (unsigned)(((lane & 7) ^ (pc & 7)) << 3)) * 2u : 0xFFFFFFFFu;
    }
  };
  // weight passes 0, 1 of the K step at the cursor (tap CT)
  auto issue_w01 = [&](auto ct_tag) __attribute__((always_inline)) {
    constexpr int CT = decltype(ct_tag)::value;
    const char* g = w_base + (CT * C + it_q * CBK) * 2;
    const unsigned d = lds0 + COFF_W + wslot * CW_BYTES + wid * 1024;
    const unsigned v0 = voff_w[0], v1 = voff_w[1];     // (copies: an asm operand alone does not make a generic lambda capture the array)
    if (!no_dma) {
      PC_DMA_S(v0, g, d);
      PC_DMA_S(v1, g, d + 64 * 128);
    }
  };
  // pass 2 and this step's patch piece (piece CT - 2 of the next chunk for CT = 2 .. 7, a dummy otherwise); then the cursor moves on
  auto issue_w2_patch_advance = [&](auto ct_tag) __attribute__((always_inline)) {
    constexpr int CT = decltype(ct_tag)::value;
    const char* g = w_base + (CT * C + it_q * CBK) * 2;
    const unsigned d = lds0 + COFF_W + wslot * CW_BYTES + wid * 1024;
    const unsigned v2 = voff_w[2];
    if (!no_dma) PC_DMA_S(v2, g, wid < 4 ? d + 128 * 128 : lds0 + COFF_DUMMY);
    if constexpr (CT >= 2 && CT < 8) {
      constexpr int k = CT - 2;
      const bool real = k < 5 || wid == 0;             // piece 40 (k = 5) belongs to wave 0 only; ids 41 .. 47 do not exist
      const unsigned o = pt_off[k];
      const u32x4 rs = pt_rsrc;
      const unsigned dp = real ? lds0 + ((it_gq + 1) & 1) * CPATCH_BYTES + (8 * k + wid) * 1024 : lds0 + COFF_DUMMY;
      if (!no_dma) PC_DMA_B(o, rs, dp);
    } else {
      const unsigned o = 0xFFFFFFFFu;                  // a dummy piece: zeros into the dummy area (uniform piece counts)
      const u32x4 rs = pt_rsrc;
      if (!no_dma) PC_DMA_B(o, rs, lds0 + COFF_DUMMY);
    }
    ++it_cnt;
    wslot = wslot == 2 ? 0 : wslot + 1;
    if constexpr (CT == 8) {                           // the cursor enters the next chunk; the patch target becomes the chunk after that one
      ++it_gq;
      if (++it_q == kcq) {
        it_q = 0;
        if (++it_tile < my_tiles) set_tile(it_tile);
      }
      const bool last_q = it_q + 1 == kcq;
      set_patch(last_q ? it_tile + 1 : it_tile, last_q ? 0 : it_q + 1);
    }
  };

  // ---- compute side
  f32x4 acc[CMT][CNT];
#pragma unroll
  for (int i = 0; i < CMT; ++i)
#pragma unroll
    for (int j = 0; j < CNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fa[CMT] = {}, fb[CNT] = {};
  // fragment byte offsets: patch rows (wm * 4 + i + ky) at columns fr + kx -> [kx][k-half] per-lane constants (key of the XOR swizzle = patch column & 7),
  // + (i + ky) * 18 * 128 (immediate + scalar); weight rows wn * 80 + j * 16 + fr
  // fragment byte offsets: patch rows (wm * 4 + i + ky) at columns fr + kx -> one per-lane constant per (kx, k-half) (key of the XOR swizzle = patch
  // column & 7; separate named registers: a table indexed at run time would live in scratch memory), + (i + ky) * 18 * 128 as an immediate; weight rows
  // wn * 80 + j * 16 + fr
  unsigned a_rd00, a_rd01, a_rd10, a_rd11, a_rd20, a_rd21, b_rd[2];
  {
    auto ao = [&](int kx, int kk) { return (unsigned)(((wm * 4) * CPW + fr + kx) * 128 + (((kk * 4 + fq) ^ ((fr + kx) & 7)) << 4)); };
    a_rd00 = ao(0, 0); a_rd01 = ao(0, 1); a_rd10 = ao(1, 0); a_rd11 = ao(1, 1); a_rd20 = ao(2, 0); a_rd21 = ao(2, 1);
  }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) b_rd[kk] = (unsigned)(COFF_W + (wn * 80 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4));
  auto read_frags = [&](int pslot, int ws, auto tap_tag, auto kk_tag) __attribute__((always_inline)) {
    constexpr int TAP = decltype(tap_tag)::value, kk = decltype(kk_tag)::value, KY = TAP / 3, KX = TAP % 3;
    const unsigned ao = KX == 0 ? (kk ? a_rd01 : a_rd00) : KX == 1 ? (kk ? a_rd11 : a_rd10) : (kk ? a_rd21 : a_rd20);
    const char* ba = smem + pslot * CPATCH_BYTES + ao;
    const char* bb = smem + ws * CW_BYTES + b_rd[kk];
    if (no_reads) return;
#pragma unroll
    for (int i = 0; i < CMT; ++i) fa[i] = *reinterpret_cast<const u32x4*>(ba + (i + KY) * (CPW * 128));
#pragma unroll
    for (int j = 0; j < CNT; ++j) fb[j] = *reinterpret_cast<const u32x4*>(bb + j * (16 * 128));
  };
  auto cluster = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (!no_mfma) {
#pragma unroll
      for (int i = 0; i < CMT; ++i)
#pragma unroll
        for (int j = 0; j < CNT; ++j) acc[i][j] = CMfma<T>::run(__builtin_bit_cast(frag, fb[j]), __builtin_bit_cast(frag, fa[i]), acc[i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- epilogue (igemm.hip fast path, PATCH rows): row group i = patch row wm * 4 + i, lane fr = its column
  T* const out = reinterpret_cast<T*>(p.out);
  const T* const res = reinterpret_cast<const T*>(p.residual);
  auto epilogue = [&](int tile) __attribute__((always_inline)) {
    int mt, n0, b, y0, x0;
    tile_origin(tile, mt, n0);
    patch_origin(mt, b, y0, x0);
    const int64_t pm0 = ((int64_t)(b * H + y0 + wm * 4) * W + x0 + fr) * N + n0 + wn * 80;   // element offset of (row group 0, channel block 0) of the lane
    const float* tb = sBias + (tile & 3) * CBN + wn * 80 + fq * 4;
    float* gs = p.stat_out;
    const int64_t rt = (int64_t)mt * 4 + wm;          // GroupNorm row tile = 64 VIRTUAL rows (patches enumerated image-major: an image's row tiles are contiguous)
    // By column-block pairs (0, 1), (2, 3), (4) -- the pairs of the 16-byte stores -- over the four row groups each: the GroupNorm sums of a pair
    // (16 registers) instead of all five blocks' (40), the residual of the NEXT pair in flight while this one is converted and stored.
    // Sums: per channel over row groups 0 .. 3, then the 16 pixel lanes by DPP -- igemm.hip's order: identical partials.
    u32x2 rva[CMT][2], rvb[CMT][2];
    auto load_res = [&](auto& rv, auto j0_tag, auto nb_tag) __attribute__((always_inline)) {
      constexpr int j0 = decltype(j0_tag)::value, NB = decltype(nb_tag)::value;
#pragma unroll
      for (int i = 0; i < CMT; ++i)
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) rv[i][jj] = *reinterpret_cast<const u32x2*>(res + pm0 + (int64_t)i * W * N + (j0 + jj) * 16 + fq * 4);
    };
    auto do_pair = [&](auto& rv, auto j0_tag, auto nb_tag) __attribute__((always_inline)) {
      constexpr int j0 = decltype(j0_tag)::value, NB = decltype(nb_tag)::value;
      f32x4 bv[NB], sm[NB], sq[NB];
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        bv[jj] = p.bias ? *reinterpret_cast<const f32x4*>(tb + (j0 + jj) * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.rowvec) bv[jj] += *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)b * p.rowvec_stride + n0 + wn * 80 + (j0 + jj) * 16 + fq * 4);
        sm[jj] = sq[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < CMT; ++i) {
        T* prow = out + pm0 + (int64_t)i * W * N;
        u32x2 po[NB];
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
          f32x4 v = acc[i][j0 + jj] + bv[jj];
          acc[i][j0 + jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (res) {
            T r[4];
            *reinterpret_cast<u32x2*>(r) = rv[i][jj];
            v[0] += to_f32(r[0]); v[1] += to_f32(r[1]); v[2] += to_f32(r[2]); v[3] += to_f32(r[3]);
          }
          T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
          po[jj] = *reinterpret_cast<u32x2*>(o);
          if constexpr (GNSTAT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x = to_f32(o[q]);
              sm[jj][q] += x;
              sq[jj][q] += x * x;
            }
          }
        }
        if constexpr (NB == 2) {
          const auto lo = __builtin_amdgcn_permlane16_swap(po[0][0], po[1][0], false, false);
          const auto hi = __builtin_amdgcn_permlane16_swap(po[0][1], po[1][1], false, false);
          const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
          *reinterpret_cast<u32x4*>(prow + (j0 + (fq & 1)) * 16 + (fq >> 1) * 8) = v;
        } else {
          *reinterpret_cast<u32x2*>(prow + j0 * 16 + fq * 4) = po[0];
        }
      }
      if constexpr (GNSTAT) {
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sm[jj][q] = c_row_sum16(sm[jj][q]);
            sq[jj][q] = c_row_sum16(sq[jj][q]);
          }
          const int n = n0 + wn * 80 + (j0 + jj) * 16 + fq * 4;
          if (fr == 0) {
            *reinterpret_cast<f32x4*>(gs + (rt * 2 + 0) * N + n) = sm[jj];
            *reinterpret_cast<f32x4*>(gs + (rt * 2 + 1) * N + n) = sq[jj];
          }
        }
      }
    };
    typedef std::integral_constant<int, 0> J0;
    typedef std::integral_constant<int, 2> J2;
    typedef std::integral_constant<int, 4> J4;
    typedef std::integral_constant<int, 1> N1;
    typedef std::integral_constant<int, 2> N2;
    if (res) { load_res(rva, J0{}, N2{}); load_res(rvb, J2{}, N2{}); }
    do_pair(rva, J0{}, N2{});
    if (res) load_res(rva, J4{}, N1{});
    do_pair(rvb, J2{}, N2{});
    do_pair(rva, J4{}, N1{});
  };

  // ---- prologue: the whole patch of chunk 0, the weights of K steps 0 and 1 (their issue positions, taps 0 and 1, carry dummy patch pieces)
  set_tile(0);
  set_patch(0, 0);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const bool real = k < 5 || wid == 0;
    const unsigned o = pt_off[k];
    const u32x4 rs = pt_rsrc;
    PC_DMA_B(o, rs, real ? lds0 + (8 * k + wid) * 1024 : lds0 + COFF_DUMMY);
  }
  set_patch(kcq > 1 ? 0 : 1, kcq > 1 ? 1 : 0);
  issue_w01(std::integral_constant<int, 0>{}); issue_w2_patch_advance(std::integral_constant<int, 0>{});
  issue_w01(std::integral_constant<int, 1>{}); issue_w2_patch_advance(std::integral_constant<int, 1>{});
  PC_VMCNT(4);                                        // K step 1's weight pieces (and its dummy patch piece) may stay in flight
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();

  int cslot = 0;
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  // one K step of the compute position (chunk parity pslot, tap TAP): everything that depends on the tap is a compile-time constant
  auto step = [&](int pslot, auto tap_tag) __attribute__((always_inline)) {
    constexpr int TAP = decltype(tap_tag)::value;
    typedef std::integral_constant<int, (TAP + 2) % 9> CT;     // the cursor's tap
    const bool more = it_cnt < total_steps;
    // ---- half 0
    if (more) issue_w01(CT{});
    read_frags(pslot, cslot, tap_tag, I0{});
    PC_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster();
    __builtin_amdgcn_s_barrier();
    // ---- half 1
    if (more) { PC_VMCNT(2); issue_w2_patch_advance(CT{}); } else { PC_VMCNT(0); }
    read_frags(pslot, cslot, tap_tag, I1{});
    PC_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster();
    __builtin_amdgcn_s_barrier();
    cslot = cslot == 2 ? 0 : cslot + 1;
  };
  int gq = 0;
  for (int t = 0; t < my_tiles; ++t) {
    for (int q = 0; q < kcq; ++q, ++gq) {
      if (t > 0 && q == 0 && !no_epi) epilogue(t - 1);   // the previous chunk finished a tile
      const int pslot = gq & 1;
      step(pslot, std::integral_constant<int, 0>{});
      step(pslot, std::integral_constant<int, 1>{});
      step(pslot, std::integral_constant<int, 2>{});
      step(pslot, std::integral_constant<int, 3>{});
      step(pslot, std::integral_constant<int, 4>{});
      step(pslot, std::integral_constant<int, 5>{});
      step(pslot, std::integral_constant<int, 6>{});
      step(pslot, std::integral_constant<int, 7>{});
      step(pslot, std::integral_constant<int, 8>{});
    }
  }
  if (!no_epi) epilogue(my_tiles - 1);
  if (!late) __builtin_amdgcn_s_barrier();
}


// ------------------------------------------------------------------------------------------------ dual-M form
// What bounds the kernel above (timing ablations on MI355X, profiles/r05_ppconv_ablation.log, cycles per K step at 2.1 GHz against 1280 of matrix work):
// MFMAs + barriers alone 1540, LDS-DMA + fragment reads WITHOUT the MFMAs 1600, the two together 2050-2250 -- the memory side of a K step is as long as
// its matrix side.  It is LDS traffic: a 64 x 80 wave tile reads 9 fragments (9 KB) per 20 MFMAs, 8 waves = 112 B / clk of the CU's 128 B / clk, and the
// DMA writes another 25 KB per step into the same banks.  The dual-M form halves what a MAC costs on that side: the M tile is TWO 16 x 16 pixel patches
// (512 pixels) that share every weight fragment -- 13 fragment reads per 40 MFMAs instead of 18, 14.6 KB of DMA per 1280 matrix cycles instead of 24.6 --
// with the K loop in 32-channel chunks so that two double-buffered 2 x 18 x 18 patches (84 KB) and three weight slots (30 KB) fit the LDS.
//   K step = (32-channel chunk, tap) = two phases: (sub-tile 0: weight fragments + 4 patch fragments), (sub-tile 1: 4 patch fragments), 20 MFMAs each
//   (one 16x16x32 k slice), wave groups in anti-phase as above; accumulators 2 x 80 registers.
//   LDS rows are 64 bytes (32 channels); 16-byte chunk c of row r sits at position c ^ ((x >> 1) & 3), x = the patch column (weights: the row) -- eight
//   consecutive rows of a ds_read_b128 then cover the eight 16-byte positions of a 128-byte bank line whatever the first row (18 is even).
//   Issue per step and wave: two weight pieces (16 rows each; rows 128 .. 159 belong to waves 0, 1, the others aim a zero-traffic piece at the dummy
//   area), then vmcnt(2), then one patch piece: piece g = 8 k + wave of the next chunk's patch in the step whose cursor tap is k + 2 (42 pieces of 16 rows:
//   21 per sub-tile), a dummy otherwise.  Same cursor logic as above.
// K order: (32-channel chunk, tap) -- not the ring's (64-channel chunk, tap, half): results differ from the kernels above by fp32 summation order.
constexpr int DBK = 32, DPIECES = 21, DSUB_ROWS = DPIECES * 16;
constexpr int DROW = DBK * 2;                                   // bytes per LDS row
constexpr int DSUB_BYTES = DSUB_ROWS * DROW, DPATCH_BYTES = 2 * DSUB_BYTES, DW_BYTES = CBN * DROW;
constexpr int DOFF_W = 2 * DPATCH_BYTES, DOFF_DUMMY = DOFF_W + 3 * DW_BYTES, DOFF_BIAS = DOFF_DUMMY + 1024, DOFF_PTL = DOFF_BIAS + 4 * CBN * 4,
              DLDS = DOFF_PTL + 7 * 512 * 4;

template <typename T, bool GNSTAT>
__global__ void __launch_bounds__(512, 2) pp_conv2_kernel(IGemmParams p) {
  typedef typename CMfma<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sBias = reinterpret_cast<float*>(smem + DOFF_BIAS);   // [4][160]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const bool late = wid >= 4;

  const int C = p.c1, N = p.N, H = p.H, W = p.W;
  const int kcq = C / DBK;                            // chunks per tile
  const int nk = 9 * kcq;                             // K steps per tile
  const int tpr = W / 16, tpi = (H / 16) * tpr;       // patches per patch row / per image
  const int tiles_n = N / CBN;
  const int total_tiles = (p.M / 512) * tiles_n;
  const int G = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
  if (my_tiles <= 0) return;
  // tile -> (patch PAIR mp, n0): sub-tile u is patch 2 mp + u of the image-major patch enumeration; patch -> (image b, origin y0, x0)
  auto tile_origin = [&](int i, int& mp, int& n0) __attribute__((always_inline)) {
    int v = blockIdx.x + i * G;
    if ((total_tiles & 7) == 0) v = (v & 7) * (total_tiles >> 3) + (v >> 3);
    mp = v / tiles_n;
    n0 = (v - mp * tiles_n) * CBN;
  };
  auto patch_origin = [&](int mt, int& b, int& y0, int& x0) __attribute__((always_inline)) {
    b = mt / tpi;
    const int r = mt - b * tpi, ty = r / tpr;
    y0 = ty * 16;
    x0 = (r - ty * tpr) * 16;
  };
  const int total_steps = my_tiles * nk;
  // timing-only ablations, COMPILE-time here (-DETAINV_PPCONV2_ABL=<bits>: 1 no DMA, 2 no epilogue, 4 no MFMA, 8 no fragment reads; run-time flags change
  // this kernel's code too much to read anything off them): VARIANT=abl9 VARIANT_FILE=ppconv VARIANT_FLAGS=-DETAINV_PPCONV2_ABL=9 bash build.sh
#ifndef ETAINV_PPCONV2_ABL
#define ETAINV_PPCONV2_ABL 0
#endif
  constexpr bool no_dma = ETAINV_PPCONV2_ABL & 1, no_epi = ETAINV_PPCONV2_ABL & 2, no_mfma = ETAINV_PPCONV2_ABL & 4, no_reads = ETAINV_PPCONV2_ABL & 8;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- issue side.  Everything a step needs is either a per-lane launch constant or a scalar that moves by an increment: the MEM phase of a wave has
  // ~320 cycles (the other group's 20 MFMAs), and recomputing addresses and flags from (tile, chunk, tap) cost 104 scalar instructions per step
  // (SQ_INSTS_SALU; the unrolled 256-pixel kernel above: 48).
  // weights: piece = 16 rows x 64 bytes, lane -> (row lane >> 2, position lane & 3); pass 0 = rows 16 wave .. + 15, pass 1 = rows 128 .. 159 (waves 0, 1)
  const unsigned wk = 9u * (unsigned)C;                // weight row length
  unsigned voff_w0, voff_w1;
  {
    const unsigned r0 = 16u * wid + (lane >> 2), r1 = 128u + 16u * (wid & 1) + (lane >> 2);
    voff_w0 = (r0 * wk + ((((unsigned)lane & 3u) ^ ((r0 >> 1) & 3u)) << 3)) * 2u;
    voff_w1 = wid < 2 ? (r1 * wk + ((((unsigned)lane & 3u) ^ ((r1 >> 1) & 3u)) << 3)) * 2u : 0u;
  }
  // patch pieces: piece g = 8 k + wave (0 .. 20 sub-tile 0, 21 .. 41 sub-tile 1, 42 .. 47 do not exist), lane -> patch row 16 (g mod 21) + (lane >> 2).
  // One launch constant per lane and piece row k, kept in LDS ([7][thread]; the kernel has no registers for them, and a register spilled to scratch memory
  // comes back through vmcnt -- every reload would drain the DMA queue): bits 0-24 the byte offset of the lane's 16 bytes from the patch's pixel (-1, -1)
  // ((pr W + pc) C + the swizzled chunk), 25 "a real lane", 26 "never" (pieces and rows that do not exist; row k = 6: the dummy piece of the steps that
  // issue none), 27 the sub-tile, 28-31 "patch row 17 / row 0 / column 0 / column 17".  Per chunk only four scalars change: per sub-tile the offset of its
  // pixel (-1, -1) in the activation tensor (at the chunk's channels; 32-bit wrap-around for the first patch) and the flags "no such tile (25), the patch
  // touches the bottom / top / left / right border (28-31)"; a lane whose flags meet the patch's is sent past the descriptor's range and reads zeros.
  typedef __attribute__((address_space(3))) unsigned lds_u32;     // (an explicit LDS pointer: through a generic one the read-back is a flat_load + vmcnt(0))
  lds_u32* const sPtl = (lds_u32*)((__attribute__((address_space(3))) char*)smem + DOFF_PTL);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int g = 8 * k + wid;
    const int u = g >= DPIECES;
    const int prow = (g - (u ? DPIECES : 0)) * 16 + (lane >> 2);
    const int pr = (prow * 3641) >> 16, pc = prow - pr * CPW;     // prow / 18 for prow < 1024
    const unsigned off = ((unsigned)(pr * W + pc) * (unsigned)C + (unsigned)(((lane & 3) ^ ((pc >> 1) & 3)) << 3)) * 2u;
    const bool exists = (k < 6) & (g < 2 * DPIECES) & (prow < CPW * CPW);
    sPtl[k * 512 + tid] = exists ? off | (1u << 25) | ((unsigned)u << 27) | (pr == CPW - 1 ? 1u << 28 : 0u) | (pr == 0 ? 1u << 29 : 0u) | (pc == 0 ? 1u << 30 : 0u) |
                                       (pc == CPW - 1 ? 1u << 31 : 0u)
                                 : 1u << 26;
  }
  const u32x4 pt_rsrc = {(unsigned)reinterpret_cast<uint64_t>(p.a1), (unsigned)(reinterpret_cast<uint64_t>(p.a1) >> 32), (unsigned)((int64_t)p.M * C * 2), 0x00020000u};
  unsigned pt_base0 = 0u, pt_base1 = 0u, pt_edge0 = 0u, pt_edge1 = 0u;
  int it_tile = 0, it_q = 0, it_cnt = 0;               // issue cursor: (tile, chunk) of the K step two steps ahead of the compute position, steps issued
  const char* w_base = nullptr;                        // weights of the cursor's tile: w + n0 * 9 C
  const char* w_ptr = nullptr;                         // ... at the cursor's (tap, chunk): + 2 C per step
  unsigned wdst = lds0 + DOFF_W + wid * 1024;          // this wave's pass-0 piece in the cursor's weight ring slot
  int wslot = 0;
  unsigned pdst = lds0 + DPATCH_BYTES + wid * 1024;    // this wave's piece row 0 in the patch slot being filled (the chunk after the cursor's)
  auto set_tile = [&](int tile) __attribute__((always_inline)) {
    int mp, n0;
    tile_origin(tile, mp, n0);
    w_base = reinterpret_cast<const char*>(p.w) + (int64_t)n0 * wk * 2;
    if (p.bias && wid < 3) {
      const char* gb = reinterpret_cast<const char*>(p.bias + n0 + wid * 64);
      const unsigned db = lds0 + DOFF_BIAS + ((tile & 3) * CBN + wid * 64) * 4;
      if (wid * 64 + lane < CBN) PC_DMA_S4((unsigned)(lane * 4), gb, db);
    }
  };
  auto set_patch = [&](int tile, int q) __attribute__((always_inline)) {   // (tile, chunk) whose patches are issued next
    const bool tile_ok = tile < my_tiles;
    int mp, n0, b0, ya, xa, b1, yb, xb;
    tile_origin(tile_ok ? tile : 0, mp, n0);
    patch_origin(2 * mp, b0, ya, xa);
    patch_origin(2 * mp + 1, b1, yb, xb);
    pt_base0 = (unsigned)((((b0 * H + ya - 1) * W + xa - 1) * C + q * DBK) * 2);
    pt_base1 = (unsigned)((((b1 * H + yb - 1) * W + xb - 1) * C + q * DBK) * 2);
    auto edge = [&](int y0, int x0) {
      return (1u << 26) | (y0 + 16 == H ? 1u << 28 : 0u) | (y0 == 0 ? 1u << 29 : 0u) | (x0 == 0 ? 1u << 30 : 0u) | (x0 + 16 == W ? 1u << 31 : 0u) | (tile_ok ? 0u : 1u << 25);
    };
    pt_edge0 = edge(ya, xa);
    pt_edge1 = edge(yb, xb);
  };
  auto patch_const = [&](int k) __attribute__((always_inline)) { return (unsigned)sPtl[k * 512 + tid]; };
  // byte offset of a lane's 16 bytes inside the activation tensor from its constant l (per-lane selects: no scalar logic on the piece number)
  auto patch_off = [&](unsigned l) __attribute__((always_inline)) {
    const bool u = (l & (1u << 27)) != 0u;
    const unsigned base = u ? pt_base1 : pt_base0, edge = u ? pt_edge1 : pt_edge0;
    return (l & edge & 0xF6000000u) != 0u ? 0xFFFFFFFFu : (l & 0x01FFFFFFu) + base;
  };
  // the two weight pieces of the K step at the cursor
  // NO dummy pieces: an LDS-DMA instruction costs the issuing wave 60-185 cycles whatever it moves (MI355X_MICROARCH.md, cycle constants), and the MEM
  // phase is what a barrier interval waits for; so waves 0, 1 issue two weight pieces per step and the others one, the 42 patch pieces of a chunk go out
  // in its first five steps (all waves) and the sixth (waves 0, 1), and every wave counts ITS OWN pieces in vmcnt: it waits until all but the weight
  // pieces it has just issued have landed (2 or 1)
  const bool two_w = wid < 2;
  auto issue_w_wait = [&]() __attribute__((always_inline)) {
    const char* g = w_ptr;
    const unsigned d = wdst, d1 = wdst + 8 * 1024;
    const unsigned v0 = voff_w0, v1 = voff_w1;
    if (no_dma) return;
    PC_DMA_S(v0, g, d);
    if (two_w) { PC_DMA_S(v1, g, d1); PC_VMCNT(2); } else { PC_VMCNT(1); }
  };
  // this step's patch piece: row k of the next chunk's pieces (k = 6: the dummy row -- zeros into the dummy area, no traffic; uniform piece counts), lane
  // constant l fetched from the LDS table a step earlier; then the cursor moves on (chunk_end: its tap was 8)
  auto issue_patch_advance = [&](int k, unsigned l, bool chunk_end) __attribute__((always_inline)) {
    const unsigned o = patch_off(l);
    const u32x4 rs = pt_rsrc;
    const unsigned dp = pdst + k * 8192;
    if (!no_dma && 8 * k + wid < 2 * DPIECES) PC_DMA_B(o, rs, dp);
    ++it_cnt;
    const bool wrap = wslot == 2;
    wslot = wrap ? 0 : wslot + 1;
    wdst = wrap ? wdst - 2 * DW_BYTES : wdst + DW_BYTES;
    w_ptr += 2 * C;
    if (chunk_end) {                                   // the cursor enters the next chunk; the patch target becomes the chunk after that one
      pdst ^= (lds0 + wid * 1024) ^ (lds0 + DPATCH_BYTES + wid * 1024);
      if (++it_q == kcq) {
        it_q = 0;
        if (++it_tile < my_tiles) set_tile(it_tile);
      }
      w_ptr = w_base + it_q * (DBK * 2);
      const bool last_q = it_q + 1 == kcq;
      set_patch(last_q ? it_tile + 1 : it_tile, last_q ? 0 : it_q + 1);
    }
  };

  // ---- compute side
  f32x4 acc0[CMT][CNT], acc1[CMT][CNT];
#pragma unroll
  for (int i = 0; i < CMT; ++i)
#pragma unroll
    for (int j = 0; j < CNT; ++j) acc0[i][j] = acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fa0[CMT] = {}, fa1[CMT] = {}, fb[CNT] = {};
  // fragment byte offsets: patch row (wm * 4 + i + ky), column fr + kx -> one per-lane constant per kx (+ (i + ky) * 18 rows and the sub-tile as immediates);
  // weight rows wn * 80 + j * 16 + fr
  // (kx is a run-time value: the row part is linear in it, the three swizzled chunk positions sit in one register, a byte each -- selecting among three
  // per-lane constants with a uniform kx compiles to scalar BRANCHES, and a taken branch costs a wave tens of cycles of instruction fetch in every phase)
  unsigned a_rd, a_sw, b_rd;
  {
    auto sw = [&](int kx) { return (unsigned)((fq ^ (((fr + kx) >> 1) & 3)) << 4); };
    a_rd = (unsigned)(((wm * 4) * CPW + fr) * DROW);
    a_sw = sw(0) | (sw(1) << 8) | (sw(2) << 16);
    b_rd = (unsigned)(DOFF_W + (wn * 80 + fr) * DROW + ((fq ^ ((fr >> 1) & 3)) << 4));
  }
  // a_off = patch slot + (ky * 18 + kx) rows, sh = 8 kx, b_off = weight ring slot: scalars that move by increments (the tap loop below)
  auto read_a = [&](unsigned a_off, unsigned sh, auto u_tag) __attribute__((always_inline)) {
    constexpr int U = decltype(u_tag)::value;
    const char* ba = smem + (a_off + U * DSUB_BYTES) + (a_rd + ((a_sw >> sh) & 0xFFu));
    if (no_reads) return;
#pragma unroll
    for (int i = 0; i < CMT; ++i) {
      if constexpr (U == 0) fa0[i] = *reinterpret_cast<const u32x4*>(ba + i * (CPW * DROW));
      else fa1[i] = *reinterpret_cast<const u32x4*>(ba + i * (CPW * DROW));
    }
  };
  auto read_b = [&](unsigned b_off) __attribute__((always_inline)) {
    const char* bb = smem + b_off + b_rd;
    if (no_reads) return;
#pragma unroll
    for (int j = 0; j < CNT; ++j) fb[j] = *reinterpret_cast<const u32x4*>(bb + j * (16 * DROW));
  };
  // ONE cluster per K step: both sub-tiles, 40 MFMAs = 640 matrix cycles -- longer than the other wave group's MEM phase (13 fragment reads: 52 KB for
  // the group's four waves = 416 LDS cycles + latency + the DMA issue).  With a cluster per sub-tile (20 MFMAs, 320 cycles) the MEM phases were the longer
  // side of every barrier interval and the kernel ran at the speed of the 256-pixel one.
#ifndef ETAINV_PPCONV2_PRIO
#define ETAINV_PPCONV2_PRIO 1
#endif
  auto cluster = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(ETAINV_PPCONV2_PRIO == 1 ? 1 : 0);
    if (!no_mfma) {
#pragma unroll
      for (int i = 0; i < CMT; ++i)
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          acc0[i][j] = CMfma<T>::run(__builtin_bit_cast(frag, fb[j]), __builtin_bit_cast(frag, fa0[i]), acc0[i][j]);
          acc1[i][j] = CMfma<T>::run(__builtin_bit_cast(frag, fb[j]), __builtin_bit_cast(frag, fa1[i]), acc1[i][j]);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- epilogue of one sub-tile (the arithmetic of the kernel above): row group i = patch row wm * 4 + i, lane fr = its column.  By column-block pairs
  // (0, 1), (2, 3), (4); the residual of a pair is loaded right before it (the other sub-tile's 80 accumulator registers are still live)
  const void* const res = p.residual;
  // Global accesses go through BUFFER instructions: a scalar descriptor over the sub-tile's image, a scalar byte offset per (row group, column block) and
  // ONE per-lane 32-bit byte offset that is constant for the launch.  (Plain pointers make hipcc build a 64-bit per-lane address for each of the 4 x 5
  // loads and stores of a sub-tile -- ~40 registers the kernel does not have; spilled, they come back through vmcnt and drain the DMA queue.)
  const int lane_ch = wn * 80 + fq * 4;
  const unsigned lane_px = (unsigned)(((wm * 4) * W + fr) * N + lane_ch) * (unsigned)sizeof(T);   // pixel (wm * 4, fr) of a patch, channel wn * 80 + fq * 4, from (patch origin, n0)
  const unsigned lane_st = (unsigned)(((wm * 4) * W + fr) * N + wn * 80 + (fq & 1) * 16 + (fq >> 1) * 8) * (unsigned)sizeof(T);   // ... at the lane's place in a 16-byte store pair
  const unsigned lane_f32 = (unsigned)lane_ch * 4u;
  auto epilogue = [&](auto u_tag, int tile) __attribute__((always_inline)) {
    constexpr int U = decltype(u_tag)::value;
    int mp, n0, b, y0, x0;
    tile_origin(tile, mp, n0);
    const int mt = 2 * mp + U;
    patch_origin(mt, b, y0, x0);
    const int img_bytes = H * W * N * (int)sizeof(T);
    const int tile_b = ((y0 * W + x0) * N + n0) * (int)sizeof(T);                 // scalar: byte offset of (patch origin, n0) inside the image
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + (int64_t)b * img_bytes, 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.residual ? p.residual : p.out)) + (int64_t)b * img_bytes, 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_vec = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.rowvec ? p.rowvec + (int64_t)b * p.rowvec_stride + n0 : p.bias), 0, CBN * 4, 0x00020000);
    const float* tb = sBias + (tile & 3) * CBN + lane_ch;
    const int64_t rt = (int64_t)mt * 4 + wm;          // GroupNorm row tile = 64 virtual rows (patches enumerated image-major)
    const __amdgpu_buffer_rsrc_t r_gs =
        __builtin_amdgcn_make_buffer_rsrc(GNSTAT ? p.stat_out + (rt * 2) * N + n0 : const_cast<float*>(p.bias), 0, GNSTAT ? (N + CBN) * 4 : 0, 0x00020000);
    auto do_pair = [&](auto j0_tag, auto nb_tag) __attribute__((always_inline)) {
      constexpr int j0 = decltype(j0_tag)::value, NB = decltype(nb_tag)::value;
      u32x2 rv[CMT][NB];
      if (res) {
#pragma unroll
        for (int i = 0; i < CMT; ++i)
#pragma unroll
          for (int jj = 0; jj < NB; ++jj)
            rv[i][jj] = __builtin_amdgcn_raw_buffer_load_b64(r_res, lane_px, tile_b + (i * W * N + (j0 + jj) * 16) * (int)sizeof(T), 0);
      }
      f32x4 bv[NB], sm[NB], sq[NB];
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        bv[jj] = p.bias ? *reinterpret_cast<const f32x4*>(tb + (j0 + jj) * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.rowvec) bv[jj] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_vec, lane_f32, (j0 + jj) * 64, 0));
        sm[jj] = sq[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < CMT; ++i) {
        const int row_b = tile_b + (i * W * N + j0 * 16) * (int)sizeof(T);
        u32x2 po[NB];
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
          f32x4 v = (U == 0 ? acc0[i][j0 + jj] : acc1[i][j0 + jj]) + bv[jj];
          if constexpr (U == 0) acc0[i][j0 + jj] = (f32x4){0.f, 0.f, 0.f, 0.f}; else acc1[i][j0 + jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (res) {
            T r[4];
            *reinterpret_cast<u32x2*>(r) = rv[i][jj];
            v[0] += to_f32(r[0]); v[1] += to_f32(r[1]); v[2] += to_f32(r[2]); v[3] += to_f32(r[3]);
          }
          T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
          po[jj] = *reinterpret_cast<u32x2*>(o);
          if constexpr (GNSTAT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x = to_f32(o[q]);
              sm[jj][q] += x;
              sq[jj][q] += x * x;
            }
          }
        }
        if constexpr (NB == 2) {
          // 16-byte stores after a lane swap between the two blocks: lane (fr, fq) writes block (fq & 1), channels 8 (fq >> 1) .. + 7 of the pixel
          const auto lo = __builtin_amdgcn_permlane16_swap(po[0][0], po[1][0], false, false);
          const auto hi = __builtin_amdgcn_permlane16_swap(po[0][1], po[1][1], false, false);
          const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
          __builtin_amdgcn_raw_buffer_store_b128(v, r_out, lane_st, row_b, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b64(po[0], r_out, lane_px, row_b, 0);
        }
      }
      if constexpr (GNSTAT) {
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sm[jj][q] = c_row_sum16(sm[jj][q]);
            sq[jj][q] = c_row_sum16(sq[jj][q]);
          }
          if (fr == 0) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sm[jj]), r_gs, lane_f32, (j0 + jj) * 64, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sq[jj]), r_gs, lane_f32, (N + (j0 + jj) * 16) * 4, 0);
          }
        }
      }
    };
    // (scheduling fences: one pair's loads, sums and stores at a time)
    __builtin_amdgcn_sched_barrier(0);
    do_pair(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    do_pair(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    do_pair(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: both patches of chunk 0 (slot 0), the weights of K steps 0 and 1 (their issue positions, taps 0 and 1, carry dummy patch pieces)
  set_tile(0);
  w_ptr = w_base;
  set_patch(0, 0);
  pdst = lds0 + wid * 1024;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const unsigned o = patch_off(patch_const(k));
    const u32x4 rs = pt_rsrc;
    const unsigned dp = pdst + k * 8192;
    if (8 * k + wid < 2 * DPIECES) PC_DMA_B(o, rs, dp);
  }
  pdst = lds0 + DPATCH_BYTES + wid * 1024;
  set_patch(kcq > 1 ? 0 : 1, kcq > 1 ? 1 : 0);
  issue_w_wait(); issue_patch_advance(6, 1u << 26, false);   // (piece row 6: none)
  issue_w_wait(); issue_patch_advance(6, 1u << 26, false);   // ... its wait: K step 1's weight pieces may stay in flight
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();

  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  unsigned l_next = patch_const(0);                   // the constant of the piece the next step issues (compute tap 0 issues piece row 0)
  unsigned b_off = 0u;
  int cslot = 0;
  // one K step of the compute position.  The tap is a RUN-TIME position (the nine taps of a chunk are one loop body: unrolled -- nine steps or three --
  // hipcc's allocator goes 30-150 registers over), and nothing in the body selects on it: a select on a uniform value compiles to a scalar BRANCH, and a
  // taken branch costs a wave tens of cycles of instruction fetch in every phase
  auto step = [&](unsigned a_off, unsigned sh, int tap) __attribute__((always_inline)) {
    const bool more = __builtin_expect(it_cnt < total_steps, 1);
    // MEM: the fragment reads first -- the DMA issue and its scalar work run under their latency -- then the step's three DMA pieces.  The cursor's tap is
    // tap + 2: piece row k = tap for taps 0 .. 5, the dummy row after that; tap 6 = cursor tap 8 ends its chunk.
    if (ETAINV_PPCONV2_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    read_b(b_off);
    read_a(a_off, sh, I0{});
    read_a(a_off, sh, I1{});
    if (more) {
      issue_w_wait();            // ... everything older than this step's weight pieces has landed: the weights the NEXT step reads, every patch piece so far
      issue_patch_advance(tap < 6 ? tap : 6, l_next, tap == 6);
    } else {
      PC_VMCNT(0);
    }
    l_next = patch_const(tap < 5 ? tap + 1 : tap == 8 ? 0 : 6);
    PC_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster();
    __builtin_amdgcn_s_barrier();
    const bool wrap = cslot == 2;
    cslot = wrap ? 0 : cslot + 1;
    b_off = wrap ? 0u : b_off + DW_BYTES;
  };
  int gq = 0;
  for (int t = 0; t < my_tiles; ++t) {
    for (int q = 0; q < kcq; ++q, ++gq) {
      if (t > 0 && q == 0 && !no_epi) { epilogue(I0{}, t - 1); epilogue(I1{}, t - 1); }   // the previous chunk finished a tile
      unsigned a_off = (gq & 1) * DPATCH_BYTES, sh = 0u;
#pragma clang loop unroll(disable)
      for (int tap = 0; tap < 9; ++tap) {
        step(a_off, sh, tap);
        const bool wrap = sh == 16u;                   // kx = 2: on to the next tap row
        sh = wrap ? 0u : sh + 8u;
        a_off += wrap ? (CPW - 2) * DROW : DROW;
      }
    }
  }
  if (!no_epi) {
    epilogue(I0{}, my_tiles - 1);
    epilogue(I1{}, my_tiles - 1);
  }
  if (!late) __builtin_amdgcn_s_barrier();
}

}  // namespace

// conv3x3 stride 1 on 16-pixel-aligned images, single source, whole 160-column tiles, plain or GroupNorm-producer epilogue (ETAINV_PPCONV=0 switches it off)
bool pp_conv_applicable(const IGemmParams& p, int dtype) {
  if (!env_flag("ETAINV_PPCONV", true) || (dtype != ETAINV_F16 && dtype != ETAINV_BF16)) return false;
  if (p.taps != 9 || p.stride != 1 || p.ups || p.a2 || p.pad0 || p.geglu || p.ln_stat || p.out_f32 || p.out_nchw || p.w_batch_stride || p.ksplit > 1 || p.hm_heads) return false;
  if (p.H % 16 != 0 || p.W % 16 != 0 || p.H != p.Ho || p.W != p.Wo || p.N % CBN != 0 || p.c1 % CBK != 0 || p.H > 4080 || p.W > 4080) return false;
  if (p.M != (p.M / (p.H * p.W)) * p.H * p.W || p.rows_per_batch != p.H * p.W) return false;
  if (p.stat_out && p.stat_kind != 1) return false;
  if (p.rowvec && p.rowvec_stride < p.N) return false;
  // both ping-pong convs address with 32-bit byte offsets (buffer descriptors per image, int patch bases): larger tensors go to the ring
  const int64_t cmax = std::max(p.c1, p.N);
  if ((int64_t)p.M * cmax * 2 >= (1ll << 31) || (int64_t)p.H * p.W * cmax * 2 >= (1ll << 31)) return false;
  static const int min_tiles = getenv("ETAINV_PPCONV_MIN_TILES") ? atoi(getenv("ETAINV_PPCONV_MIN_TILES")) : 192;
  return (int64_t)(p.M / 256) * (p.N / CBN) >= min_tiles;
}

// the dual-M form: an even number of 16 x 16 patches, and a last round of the persistent grid that is at least 80 % full (512-pixel tiles halve the tile
// count: the 16 x 16 level of a 32-row call has 128 of them).  ETAINV_PPCONV2=0 keeps the 256-pixel kernel everywhere
static bool pp_conv2_ok(const IGemmParams& p) {
  if (!env_flag("ETAINV_PPCONV2", true) || (p.M / 256) % 2 != 0) return false;
  if ((int64_t)p.M * p.c1 * 2 >= (1ll << 32) || (int64_t)CPW * p.W * p.c1 * 2 >= (1ll << 25)) return false;   // 32-bit tensor offsets, 25-bit patch offsets
  const int64_t tiles = (int64_t)(p.M / 512) * (p.N / CBN), rounds = (tiles + 255) / 256;
  return tiles * 5 >= rounds * 256 * 4;
}

int launch_pp_conv(const IGemmParams& p_in, int dtype, hipStream_t s, int* stat_P) {
  IGemmParams p = p_in;
  ETAINV_CHECK(p.zeros, "zero page");
  if (stat_P) *stat_P = p.stat_out ? 64 : 0;          // GroupNorm partials: rows per row block = the 64-row wave tile
  if (p.stat_out) p.stat_P = 64;
  if (pp_conv2_ok(p)) {
    const int tiles2 = (p.M / 512) * (p.N / CBN);
    const int grid2 = std::min(tiles2, 256);
    static bool attr2_set[kMaxDevices] = {};
    const int dev2 = current_device();
    ETAINV_DISPATCH_HALF(dtype, T, {
      if (!attr2_set[dev2]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv2_kernel<f16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv2_kernel<f16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv2_kernel<bf16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv2_kernel<bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS);
        attr2_set[dev2] = true;
      }
      if (p.stat_out) hipLaunchKernelGGL((pp_conv2_kernel<T, true>), dim3(grid2), dim3(512), DLDS, s, p);
      else hipLaunchKernelGGL((pp_conv2_kernel<T, false>), dim3(grid2), dim3(512), DLDS, s, p);
    });
    ETAINV_LAUNCH_CHECK();
    return 0;
  }
  const int tiles = (p.M / 256) * (p.N / CBN);
  const int grid = std::min(tiles, 256);
  static bool attr_set[kMaxDevices] = {};
  const int dev = current_device();
  ETAINV_DISPATCH_HALF(dtype, T, {
    if (!attr_set[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv_kernel<f16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv_kernel<f16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv_kernel<bf16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv_kernel<bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CLDS);
      attr_set[dev] = true;
    }
    if (p.stat_out) hipLaunchKernelGGL((pp_conv_kernel<T, true>), dim3(grid), dim3(512), CLDS, s, p);
    else hipLaunchKernelGGL((pp_conv_kernel<T, false>), dim3(grid), dim3(512), CLDS, s, p);
  });
  ETAINV_LAUNCH_CHECK();
  return 0;
}

}  // namespace etainv
