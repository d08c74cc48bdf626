// THE PRODUCT KERNEL OF THIS FILE IS pp_dualn_kernel (second half: "Dual-N ping-pong GEMM").  The first half documents and -- under EXPERIMENTS=1 only -- builds its
// round-5 predecessor, the double-accumulator ping-pong GEMM (3-10 % on a class it no longer serves; measurements: profiles/HISTORY.md, r05_pp_gemm_check.log):
// Ping-pong GEMM for the short-K 1x1 / Linear layers of the transformer blocks (gfx950): the 256 x 160 x 64 tile of igemm.hip's ring with
//   (1) its two wave groups in ANTI-PHASE, one fragment set per wave, and
//   (2) TWO accumulator sets: the epilogue of tile t runs in slices inside the memory phases of tile t + 1.
//
// Why (round 5, measured: profiles/r05_pp_ablation.log, r05_pmc_sq_rows128.json).  The ring kernel runs its eight waves in lockstep; for the K = 320 .. 1280
// GEMMs a tile is 5 .. 20 K steps of ~2400 cycles each -- bound by the LDS-DMA fill rate of the CU (52 KB per step at ~23 B/clk; the matrix pipe needs 1280
// cycles per step) -- followed by an epilogue of ~7000 cycles (bias / residual / convert / stores, LayerNorm statistics) during which the matrix pipe AND
// the DMA stream of the block stand still: 0.23-0.37 MFMA-busy in the SQ counters.  With the loop DMA-bound there is matrix and VALU time to spare inside it;
// what is missing is a place for the epilogue to run.  Structure here:
//
//   phase of a wave = one 32-deep k-half of a K step:   MEM: 9 ds_read_b128 (its fragments of that half), its share of the LDS-DMA issue (addresses =
//                                                             scalar base + a per-lane byte offset held in a register: no address arithmetic in the loop),
//                                                             one SLICE of the previous tile's epilogue, lgkmcnt(0)
//                                                        s_barrier
//                                                        COMPUTE: s_setprio 1, 20 MFMAs 16x16x32 into the accumulator set of the tile's parity, s_setprio 0
//                                                        s_barrier
//   Waves 4-7 (group L) execute one extra s_barrier up front: in every barrier interval one wave of each SIMD computes while its partner is in a MEM phase
//   (MI355X_MICROARCH.md "Two waves per SIMD"; cdna_hip_programming.md "The 256^2 8-phase template").  Alone (no DMA, no reads) the structure issues MFMAs
//   for 96 % of the time.
//
// Epilogue slices of tile t - 1 (accumulator set 1 - P): row group i of the wave tile in MEM phase ph = i (= 2 k + h, k < 2) of tile t -- residual loads
// (inline asm: hipcc must not count them -- beside LDS-DMA it drains the whole queue at the first use of an ordinary load; the lines were brought to L2
// a tile earlier by three touch pieces per wave aimed at the LDS dummy area), the phase's DMA pieces, a counted wait, then bias + residual, convert, lane
// swap, three stores (+ the LayerNorm row statistics of the stored values).  The residual registers live inside one phase only.
// vmcnt bookkeeping (loads, stores and LDS-DMA retire in issue order): every wait is a compile-time count of the operations issued BEHIND the ones that
// must have landed -- wait_tile() below; an unexpected extra operation (bias or touch piece) only makes a wait cover more.  The counts need DMA issue in the
// first four phases of every tile: launches with K >= 320 only (five K steps: the last tile of a block still issues through its step 2).
//
// LDS: 3-slot ring of [256 + 160 rows][64] K tiles exactly as in igemm.hip (row = 8 chunks of 16 B, physical chunk = chunk ^ (row & 7), filled lane-linearly
// by global_load_lds_dwordx4 with the swizzle on the SOURCE chunk).  Hazards, in barrier intervals (E reads half h of step s in interval 4 s + 2 h, L one later;
// reads are waited for before the barrier that ends their MEM phase):
//   * K tile s + 2 goes into the slot of step s - 1 (last read: L, interval 4 s - 1): weight pieces in MEM (s, half 0) = interval >= 4 s, activation
//     pieces in MEM (s, half 1);
//   * a wave waits for its own pieces of K tile s + 1 (issued >= 4 intervals earlier) in MEM (s, half 1); the barrier behind it precedes the first read.
// Operand / accumulator conventions and the accumulation order over K are igemm.hip's: results are bit-identical to the ring kernel's.
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
// one LDS-DMA piece: scalar base + a per-lane 32-bit byte offset, M0 = the LDS destination written in the statement that uses it (see pp_gemm_kernel's issue side)
#define PP_DMA(BYTES_INSN, voff32, sbase, ldsaddr) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" BYTES_INSN " %0, %1" : : "v"(voff32), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(ldsaddr)) : "memory")

constexpr int PBM = 256, PBN = 160, PBK = 64;
constexpr int PMT = 4, PNT = 5;                 // wave tile 64 x 80 (4 x 2 waves)
constexpr int PA_BYTES = PBM * PBK * 2, PB_BYTES = PBN * PBK * 2;   // one ring slot of each operand
constexpr int POFF_B = 3 * PA_BYTES, POFF_DUMMY = POFF_B + 3 * PB_BYTES, POFF_BIAS = POFF_DUMMY + 1024;
constexpr int PLDS = POFF_BIAS + 4 * PBN * 4;

#ifdef ETAINV_EXPERIMENTS   // the double-accumulator kernel of the header comment: opt-in experiment, built by `EXPERIMENTS=1 build.sh` only
// ---- the slice schedule and the counted waits derived from it.  Slice i (row group i of the previous tile) rides in MEM phase ph = i of the next tile:
//   [residual: five 8-byte loads of the row group]  [the phase's DMA pieces]  [wait: the loads have landed, the pieces stay in flight]  [bias, residual,
//   convert, lane swap, three stores (+ one for the LayerNorm row statistics)]  [fragment reads]
constexpr bool slice_in(int ph) { return ph >= 0 && ph < 4; }
constexpr int n_stores(bool ST, int ph) { return slice_in(ph) ? 3 + (ST ? 1 : 0) : 0; }
constexpr int n_loads(bool RES, int ph) { return RES && slice_in(ph) ? PNT : 0; }
// MEM (k, half 1), behind that phase's residual loads and in front of its activation pieces: operations behind the last piece of K tile s + 1 (issued in
// phase 2 k - 1, in front of that phase's stores)
constexpr int wait_tile(bool RES, bool ST, int k) {
  return n_stores(ST, 2 * k - 1) + n_loads(RES, 2 * k) + 3 + n_stores(ST, 2 * k) + n_loads(RES, 2 * k + 1);
}

template <typename T, bool RES, bool STAT>
__global__ void __launch_bounds__(512, 2) pp_gemm_kernel(IGemmParams p) {
  typedef typename PMfma<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sBias = reinterpret_cast<float*>(smem + POFF_BIAS);   // [4][160] bias of the tiles in flight

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const bool late = wid >= 4;                       // group L

  const int K = p.c1, N = p.N;
  const int nk = K / PBK;
  const int tiles_n = N / PBN;
  const int total_tiles = (p.M / PBM) * tiles_n;
  const int G = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
  if (my_tiles <= 0) return;
  auto tile_origin = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
    int v = blockIdx.x + i * G;
    if ((total_tiles & 7) == 0) v = (v & 7) * (total_tiles >> 3) + (v >> 3);   // XCD-contiguous tile runs, n fastest (as in igemm.hip)
    const int tm = v / tiles_n;
    m0 = tm * PBM;
    n0 = (v - tm * tiles_n) * PBN;
  };
  const int total_steps = my_tiles * nk;
  const bool no_dma = p.debug & 1, no_epi = p.debug & 2, no_mfma = p.debug & 4;   // timing-only ablations (ETAINV_IGEMM_DEBUG)

  // ---- issue side.  Row r = (tid >> 3) + 64 q of a tile, the lane's 16-byte chunk swizzled at the source: byte offset of the lane inside the tile's operand
  // panel, the same for both operands (both are [rows][K] with K contiguous); the tile / K-tile position is a scalar base
  const unsigned lrow = tid >> 3;
  unsigned voff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) voff[q] = ((lrow + 64 * q) * (unsigned)K + (((tid & 7) ^ (lrow & 7)) << 3)) * 2u;
  const unsigned voff_b2 = wid < 4 ? voff[2] : voff[0];   // weight rows 128 .. 159: waves 0-3; the others re-read row group 0 into the dummy area (uniform counts)
  const int wrow_b = wid * 8 * PBK * 2;                   // byte offset of the wave's 8-row piece inside a 64-row pass
  // LDS-DMA in the SGPR-base form by inline asm: `global_load_lds_dwordx4 v_off, s[base:base+1]` with M0 = the piece's LDS address.  (Through the builtin
  // hipcc's loop optimisations widen the lane offsets to 64-bit register pairs and rebuild a 64-bit address per piece -- 10 more registers and, once those
  // spill, scratch reloads with vmcnt(0) inside the loop.)  No register destination: nothing for the compiler to mis-time; M0 is written in the statement
  // that uses it and nothing else in this kernel uses M0.
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  const char* a_base = nullptr;
  const char* w_base = nullptr;
  int it_tile = 0, it_kt = 0, issued = 0, islot = 0;      // issue position: (tile, K tile), K tiles issued so far, ring slot of the position
  auto setup_issue = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(tile, m0, n0);
    a_base = reinterpret_cast<const char*>(p.a1) + (int64_t)m0 * K * 2;
    w_base = reinterpret_cast<const char*>(p.w) + (int64_t)n0 * K * 2;
    if (p.bias && wid < 3) {                        // 160 floats: waves 0, 1 whole, wave 2 its first 32 lanes
      const char* gb = reinterpret_cast<const char*>(p.bias + n0 + wid * 64);
      const unsigned db = lds0 + POFF_BIAS + ((tile & 3) * PBN + wid * 64) * 4;
      if (wid * 64 + lane < PBN) PP_DMA("global_load_lds_dword", (unsigned)(lane * 4), gb, db);
    }
  };
  auto issue_b = [&]() __attribute__((always_inline)) {
    if (no_dma) return;
    const char* g = w_base + it_kt * (PBK * 2);
    const unsigned d = lds0 + POFF_B + islot * PB_BYTES + wrow_b;
    PP_DMA("global_load_lds_dwordx4", voff[0], g, d);
    PP_DMA("global_load_lds_dwordx4", voff[1], g, d + 64 * PBK * 2);
    PP_DMA("global_load_lds_dwordx4", voff_b2, g, wid < 4 ? d + 128 * PBK * 2 : lds0 + POFF_DUMMY);
  };
  auto issue_a = [&]() __attribute__((always_inline)) {
    if (no_dma) return;
    const char* g = a_base + it_kt * (PBK * 2);
    const unsigned d = lds0 + islot * PA_BYTES + wrow_b;
    PP_DMA("global_load_lds_dwordx4", voff[0], g, d);
    PP_DMA("global_load_lds_dwordx4", voff[1], g, d + 1 * (64 * PBK * 2));
    PP_DMA("global_load_lds_dwordx4", voff[2], g, d + 2 * (64 * PBK * 2));
    PP_DMA("global_load_lds_dwordx4", voff[3], g, d + 3 * (64 * PBK * 2));
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++issued;
    islot = islot == 2 ? 0 : islot + 1;
    if (++it_kt == nk) {
      it_kt = 0;
      if (++it_tile < my_tiles) setup_issue(it_tile);
    }
  };

  // ---- compute side
  f32x4 acc0[PMT][PNT], acc1[PMT][PNT];            // (never zeroed: the first cluster of a tile takes C = 0; a row group's registers are free once its slice has stored it)
  u32x4 fa[PMT] = {}, fb[PNT] = {};
  unsigned a_rd[2], b_rd[2];                       // the lane's fragment byte offsets inside a slot, per k-half (row & 7 == fr & 7 for every row group)
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_rd[kk] = ((wm * 64 + fr) * PBK + (((kk * 4 + fq) ^ (fr & 7)) << 3)) * 2;
    b_rd[kk] = POFF_B + ((wn * 80 + fr) * PBK + (((kk * 4 + fq) ^ (fr & 7)) << 3)) * 2;
  }
  auto read_frags = [&](int slot, auto kk_tag) __attribute__((always_inline)) {
    constexpr int kk = decltype(kk_tag)::value;
    const char* ba = smem + slot * PA_BYTES + a_rd[kk];
    const char* bb = smem + slot * PB_BYTES + b_rd[kk];
#pragma unroll
    for (int i = 0; i < PMT; ++i) fa[i] = *reinterpret_cast<const u32x4*>(ba + i * (16 * PBK * 2));
#pragma unroll
    for (int j = 0; j < PNT; ++j) fb[j] = *reinterpret_cast<const u32x4*>(bb + j * (16 * PBK * 2));
  };
  auto cluster = [&](auto p_tag, auto first_tag) __attribute__((always_inline)) {
    constexpr int P = decltype(p_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (!no_mfma) {
#pragma unroll
      for (int i = 0; i < PMT; ++i)
#pragma unroll
        for (int j = 0; j < PNT; ++j) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          if constexpr (P == 0) acc0[i][j] = PMfma<T>::run(__builtin_bit_cast(frag, fb[j]), __builtin_bit_cast(frag, fa[i]), FIRST ? z : acc0[i][j]);
          else acc1[i][j] = PMfma<T>::run(__builtin_bit_cast(frag, fb[j]), __builtin_bit_cast(frag, fa[i]), FIRST ? z : acc1[i][j]);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- epilogue of one tile, by row group (lane: pixel row m = m0 + wm * 64 + i * 16 + fr, channels n0 + wn * 80 + j * 16 + fq * 4 .. + 3)
  T* const out = reinterpret_cast<T*>(p.out);
  const T* const res = reinterpret_cast<const T*>(p.residual);
  const int64_t rg_stride = (int64_t)16 * N;        // elements between two row groups
  // addresses = scalar element offset of the tile's origin (ep_base) + a per-lane byte offset that is the same for every tile (ep_lane: row group 0,
  // channel block 0 of the lane)
  int64_t ep_base = 0;
  const unsigned ep_lane = (unsigned)(((wm * 64 + fr) * N + wn * 80 + fq * 4) * 2);
  int ep_tile = 0, ep_m = 0, ep_pidx = 0;
  auto ep_begin = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(tile, m0, n0);
    ep_tile = tile;
    ep_m = m0 + wm * 64 + fr;
    ep_pidx = n0 / 80 + wn;
    ep_base = (int64_t)m0 * N + n0;
  };
  // (one asm statement per row group: five loads in the SGPR-base form, early-clobber outputs)
#define PP_RES_LOAD5(RV, voff32, sbase)                                                                                                                  \
  asm volatile("global_load_dwordx2 %0, %5, %6\n\tglobal_load_dwordx2 %1, %5, %6 offset:32\n\tglobal_load_dwordx2 %2, %5, %6 offset:64\n\t"              \
               "global_load_dwordx2 %3, %5, %6 offset:96\n\tglobal_load_dwordx2 %4, %5, %6 offset:128"                                                  \
               : "=&v"(RV[0]), "=&v"(RV[1]), "=&v"(RV[2]), "=&v"(RV[3]), "=&v"(RV[4]) : "v"(voff32), "s"(sbase) : "memory")
  // wait until at most n operations are in flight, then the residual registers may be read (names them: no consumer is scheduled above the wait)
#define PP_RES_WAIT(n, RV) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(RV[0]), "+v"(RV[1]), "+v"(RV[2]), "+v"(RV[3]), "+v"(RV[4]) : "i"(n) : "memory")
  // the residual tile of `tile` towards L2: three dword pieces per wave (lane = pixel row of the wave tile, at byte 0 / 128 / 156 of its 160-byte span)
  // aimed at the LDS dummy area -- no register destination
  auto res_touch = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(tile, m0, n0);
    const char* g = reinterpret_cast<const char*>(res) + ((int64_t)(m0 + wm * 64) * N + n0 + wn * 80) * 2;
    const unsigned row = (unsigned)lane * (unsigned)N * 2u;
    PP_DMA("global_load_lds_dword", row, g, lds0 + POFF_DUMMY);
    PP_DMA("global_load_lds_dword", row + 128u, g, lds0 + POFF_DUMMY + 256);
    PP_DMA("global_load_lds_dword", row + 156u, g, lds0 + POFF_DUMMY + 512);
  };
  // one row group: [residual loads] [dma(): the phase's DMA issue, n_young pieces] [wait] [arithmetic, stores]
  auto store_group = [&](auto p_tag, auto i_tag, auto dma, auto young_tag) __attribute__((always_inline)) {
    constexpr int Q = decltype(p_tag)::value, i = decltype(i_tag)::value, YOUNG = decltype(young_tag)::value;
    const float* tb = sBias + (ep_tile & 3) * PBN + wn * 80 + fq * 4;
    unsigned long long rv[PNT];
    if constexpr (RES) {
      const char* g = reinterpret_cast<const char*>(res) + (ep_base + i * rg_stride) * 2;
      const unsigned el = ep_lane;
      PP_RES_LOAD5(rv, el, g);
    }
    dma();
    if constexpr (RES) {
      PP_RES_WAIT(YOUNG, rv);
      __builtin_amdgcn_sched_barrier(0);
    }
    u32x2 po[PNT];
#pragma unroll
    for (int j = 0; j < PNT; ++j) {
      f32x4 v;
      if constexpr (Q == 0) v = acc0[i][j];
      else v = acc1[i][j];
      if (p.bias) v += *reinterpret_cast<const f32x4*>(tb + j * 16);
      if constexpr (RES) {
        const unsigned long long r64 = rv[j];
        T r[4];
        *reinterpret_cast<unsigned long long*>(r) = r64;
        v[0] += to_f32(r[0]); v[1] += to_f32(r[1]); v[2] += to_f32(r[2]); v[3] += to_f32(r[3]);
      }
      T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
      po[j] = *reinterpret_cast<u32x2*>(o);
    }
    // 16-byte stores after a lane swap between adjacent 16-column blocks (igemm.hip store_row_group): 64-byte segments per pixel row
    T* prow = reinterpret_cast<T*>(reinterpret_cast<char*>(out + ep_base + i * rg_stride) + (ep_lane - fq * 8));
#pragma unroll
    for (int k = 0; k + 1 < PNT; k += 2) {
      const auto lo = __builtin_amdgcn_permlane16_swap(po[k][0], po[k + 1][0], false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap(po[k][1], po[k + 1][1], false, false);
      const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
      *reinterpret_cast<u32x4*>(prow + (k + (fq & 1)) * 16 + (fq >> 1) * 8) = v;
    }
    *reinterpret_cast<u32x2*>(prow + (PNT - 1) * 16 + fq * 4) = po[PNT - 1];
    if constexpr (STAT) {
      // LayerNorm producer (igemm.hip emit_row_stat): (mean, M2) of the 20 stored values of this lane, merged over the four fq lanes = the 80 columns of the
      // wave tile -> partial ep_pidx of row m
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < PNT; ++j) {
        T o[4];
        *reinterpret_cast<u32x2*>(o) = po[j];
        sum += (to_f32(o[0]) + to_f32(o[1])) + (to_f32(o[2]) + to_f32(o[3]));
      }
      float mu = sum * (1.0f / (float)(PNT * 4)), m2 = 0.f;
#pragma unroll
      for (int j = 0; j < PNT; ++j) {
        T o[4];
        *reinterpret_cast<u32x2*>(o) = po[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float d = to_f32(o[q]) - mu; m2 += d * d; }
      }
      {
        float ma = mu, mb = mu, qa = m2, qb = m2;     // rows fq 0 | 1 (and 2 | 3), then the halves
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(qa), "+v"(qb));
        float d = mb - ma;
        ma += 0.5f * d;
        qa += qb + d * d * (0.5f * (float)(PNT * 4));
        float m_lo = ma, m_hi = ma, q_lo = qa, q_hi = qa;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(m_lo), "+v"(m_hi));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(q_lo), "+v"(q_hi));
        d = m_hi - m_lo;
        mu = m_lo + 0.5f * d;
        m2 = q_lo;
        m2 += q_hi + d * d * (0.5f * (float)(2 * PNT * 4));
      }
      if (fq == 0) *reinterpret_cast<f32x2*>(p.stat_out + ((int64_t)(ep_m + i * 16) * p.stat_P + ep_pidx) * 2) = (f32x2){mu, m2};   // (one exec-masked store instruction)
    }
  };
  auto full_epilogue = [&](auto p_tag, int tile) __attribute__((always_inline)) {   // not overlapped: the last tile of the block
    if (no_epi) return;
    ep_begin(tile);
    auto nothing = [&]() __attribute__((always_inline)) {};
    typedef std::integral_constant<int, 0> Y0;
    store_group(p_tag, std::integral_constant<int, 0>{}, nothing, Y0{});
    store_group(p_tag, std::integral_constant<int, 1>{}, nothing, Y0{});
    store_group(p_tag, std::integral_constant<int, 2>{}, nothing, Y0{});
    store_group(p_tag, std::integral_constant<int, 3>{}, nothing, Y0{});
  };

  // ---- prologue: K tiles 0 and 1 whole; wait for tile 0
  if constexpr (RES) {
    res_touch(0);
    if (my_tiles > 1) res_touch(1);
  }
  setup_issue(0);
  issue_a(); issue_b(); advance();
  if (total_steps > 1) { issue_a(); issue_b(); advance(); PP_VMCNT(7); } else { PP_VMCNT(0); }
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();           // group L runs one interval behind

  int slot = 0;
  // one tile into accumulator set P; has_prev: the four slices of the previous tile's epilogue (set 1 - P) ride in the MEM phases of its first two steps.
  // Order of a MEM phase: (slice: residual loads,) DMA issue, (slice: wait, arithmetic, stores,) fragment reads -- the slice's temporaries are dead before
  // the fragment registers are written (two accumulator sets + fragments + a slice do not fit 256 registers together); the loop is DMA-bound, the MEM phases
  // have the time
  auto tile_body = [&](auto p_tag, int tile, bool has_prev) __attribute__((always_inline)) {
    typedef std::integral_constant<int, 1 - decltype(p_tag)::value> QT;
    if (has_prev) ep_begin(tile - 1);
    for (int k = 0; k < nk; ++k) {
      const bool more = issued < total_steps;       // a K tile s + 2 exists: its pieces go out in this step
      // ---- (k, half 0): MEM
      auto dma_b = [&]() __attribute__((always_inline)) { if (more) issue_b(); };
      if (has_prev && k < 2 && !no_epi) {
        if (k == 0) store_group(QT{}, std::integral_constant<int, 0>{}, dma_b, std::integral_constant<int, 3>{});
        else store_group(QT{}, std::integral_constant<int, 2>{}, dma_b, std::integral_constant<int, 3>{});
      } else {
        dma_b();
        if (RES && k == 2 && tile + 1 < my_tiles) res_touch(tile + 1);   // (tile + 1's residual is read a whole tile later)
      }
      __builtin_amdgcn_sched_barrier(0);
      read_frags(slot, std::integral_constant<int, 0>{});
      PP_LGKMCNT0();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      if (k == 0) cluster(p_tag, std::true_type{});
      else cluster(p_tag, std::false_type{});
      __builtin_amdgcn_s_barrier();
      // ---- (k, half 1): MEM
      auto dma_a = [&]() __attribute__((always_inline)) {
        if (more) {
          // K tile s + 1 has landed (for this wave); everything issued behind it stays in flight
          if (!has_prev || k >= 3) PP_VMCNT(3);
          else if (k == 0) PP_VMCNT(wait_tile(RES, STAT, 0));
          else if (k == 1) PP_VMCNT(wait_tile(RES, STAT, 1));
          else PP_VMCNT(wait_tile(RES, STAT, 2));
          issue_a();
          advance();
        } else {
          PP_VMCNT(0);
        }
      };
      if (has_prev && k < 2 && !no_epi) {
        if (k == 0) store_group(QT{}, std::integral_constant<int, 1>{}, dma_a, std::integral_constant<int, 4>{});
        else store_group(QT{}, std::integral_constant<int, 3>{}, dma_a, std::integral_constant<int, 4>{});
      } else {
        dma_a();
      }
      __builtin_amdgcn_sched_barrier(0);
      read_frags(slot, std::integral_constant<int, 1>{});
      PP_LGKMCNT0();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(p_tag, std::false_type{});
      __builtin_amdgcn_s_barrier();
      slot = slot == 2 ? 0 : slot + 1;
    }
  };
  for (int t = 0; t < my_tiles; t += 2) {
    tile_body(std::integral_constant<int, 0>{}, t, t > 0);
    if (t + 1 < my_tiles) tile_body(std::integral_constant<int, 1>{}, t + 1, true);
  }
  if ((my_tiles - 1) & 1) full_epilogue(std::integral_constant<int, 1>{}, my_tiles - 1);
  else full_epilogue(std::integral_constant<int, 0>{}, my_tiles - 1);
  if (!late) __builtin_amdgcn_s_barrier();
}


#endif   // ETAINV_EXPERIMENTS

// ============================================================================================================================================================
// Dual-N ping-pong GEMM: 256 x 320 output tile as TWO 160-column halves that share one activation K tile.
//
// Why: the ablations of the kernel above (profiles/r05_pp_ablation.log) say the main loop of every non-PATCH GEMM on the 256 x 160 tile is bound by the
// LDS-DMA fill rate of the CU: 52 KB per K step arrive at ~23 B/clk (2300-2700 cycles) while the matrix pipe needs 1280.  Bytes per FLOP are the lever:
// with both 160-column halves of a 320-column span computed from ONE staged activation tile, a K tile costs 32 KB (activations) + 2 x 20 KB (weights) =
// 72 KB for twice the FLOPs = -31 % (and 14 instead of 18 fragment reads per 40 MFMAs).  Two accumulator sets of 80 registers hold the two halves -- which
// the ping-pong form can afford (one fragment set) and the ring cannot (256 registers with two).  All channel counts of SD1.x are multiples of 320.
//
// A K tile is four phases per wave -- (kk 0, half 0), (kk 0, half 1), (kk 1, half 0), (kk 1, half 1): the activation fragments of a k-half are read in the
// first phase of the pair and kept -- i.e. eight barrier intervals (group E: interval 8 s + 2 ph, group L one later).  LDS: TWO slots of [256 + 320 rows][64]
// (144 KB), recycled by region (every wave finishes its fragment reads before the barrier that ends its MEM phase):
//   activation rows of slot s:  last read in L's phase 2 (interval 8 s + 5)  -> the pieces of K tile s + 2 go out in phase 3 of tile s (>= interval 8 s + 6)
//   weight rows 0 .. 159:       last read in L's phase 2                      -> K tile s + 2 rows 0 .. 127 in phase 0 of tile s + 1 (>= interval 8 s + 8)
//   weight rows 160 .. 319:     last read in L's phase 3 (interval 8 s + 7)   -> rows 128 .. 255 in phase 1, rows 256 .. 319 in phase 2 of tile s + 1
//   every wave waits for ALL its pieces of K tile s + 2 (vmcnt(0): nothing else is in flight) in phase 3 of tile s + 1, in front of the activation pieces of
//   K tile s + 3; the barrier behind that wait precedes the first read of K tile s + 2.  The youngest piece waited for is two intervals old, the oldest eight.
// Epilogue (bias, residual, LayerNorm row statistics -- igemm.hip's arithmetic, bit-identical results): per half, at the end of the output tile, not overlapped.
// Instantiations: HN = 160 (wave tile 64 x 80 per half) with EPI 0 = bias (+ residual) (+ LayerNorm row statistics), 1 = LayerNorm consumer
// (out = rstd (acc - mean s) + c, IGemmParams::ln_stat), 2 = the same writing the head-major QKV planes (IGemmParams::hm_*); HN = 128 (wave tile 64 x 64
// per half, 256 x 256 tile) with EPI 3 = LayerNorm consumer + GEGLU (a * gelu_erf(g), value / gate columns interleaved per 64 by pack mode 2).
// The LayerNorm consumers' s vector and (mean, rstd) rows arrive by DMA with the bias (two-deep rings: the weight cursor enters tile t + 2 only after
// the epilogue of tile t).
// sum over the 16 lanes of a DPP row (igemm.hip row_sum16)
__device__ __forceinline__ float dualn_row_sum16(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));
  return x;
}

template <int HN> struct DualN {
  static constexpr int NT = HN / 32;                 // 16-column blocks per wave tile (two waves across a half)
  static constexpr int WN = HN / 2;
  static constexpr int BN2 = 2 * HN;                 // weight rows per K tile
  static constexpr int PASSES = BN2 / 64;            // 64-row DMA passes of the weight tile
  static constexpr int A_BYTES = PBM * PBK * 2, B_BYTES = BN2 * PBK * 2, SLOT = A_BYTES + B_BYTES;
  static constexpr int OFF_BIAS = 2 * SLOT, OFF_S = OFF_BIAS + 2 * BN2 * 4, OFF_STAT = OFF_S + 2 * BN2 * 4, LDS = OFF_STAT + 2 * PBM * 2 * 4;
};

// A2: two activation sources [M][c1] | [M][c2] (the 1x1 shortcut of an up-block resnet reads the concatenation of the hidden state and the skip tensor without
// materialising it: K tiles 0 .. c1 / 64 - 1 come from a1, the rest from a2; the weight rows are [N][c1 + c2])
template <typename T, int HN, int EPI, bool RES, int STAT, bool A2 = false>   // STAT: 0 none, 1 LayerNorm row statistics, 2 GroupNorm channel statistics of the stored output
__global__ void __launch_bounds__(512, 2) pp_dualn_kernel(IGemmParams p) {
  typedef typename PMfma<T>::frag frag;
  typedef DualN<HN> D;
  constexpr int NT = D::NT, WN = D::WN, BN2 = D::BN2, PASSES = D::PASSES;
  constexpr bool LNC = EPI != 0;                   // LayerNorm consumer
  static_assert((HN == 160 && EPI <= 2) || (HN == 128 && EPI == 3), "instantiations");
  static_assert(!LNC || (!RES && !STAT), "a LayerNorm consumer has no residual and emits no statistics");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sBias = reinterpret_cast<float*>(smem + D::OFF_BIAS);   // [2][BN2] bias (LayerNorm consumer: the folded c vector) of the tiles in flight
  float* sS = reinterpret_cast<float*>(smem + D::OFF_S);         // [2][BN2] s vector
  float* sStat = reinterpret_cast<float*>(smem + D::OFF_STAT);   // [2][256][2] (mean, rstd)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const bool late = wid >= 4;

  const int K1 = p.c1, K = A2 ? p.c1 + p.c2 : p.c1, N = p.N;
  const int nk = K / PBK, nk1 = K1 / PBK;
  const int tiles_n = N / BN2;
  const int total_tiles = (p.M / PBM) * tiles_n;
  const int G = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
  if (my_tiles <= 0) return;
  auto tile_origin = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
    int v = blockIdx.x + i * G;
    if ((total_tiles & 7) == 0) v = (v & 7) * (total_tiles >> 3) + (v >> 3);
    const int tm = v / tiles_n;
    m0 = tm * PBM;
    n0 = (v - tm * tiles_n) * BN2;
  };
  const int total_steps = my_tiles * nk;
  const bool no_dma = p.debug & 1, no_epi = p.debug & 2, no_store = p.debug & 8, no_gelu = p.debug & 16;   // (timing-only ablations, tools/pp_check.py ablate)
  const bool fake_store = p.debug & 64;
  const bool has_bias = p.bias != nullptr;

  // ---- issue side: two cursors (the activation pieces of a K tile go out four phases before its weight pieces)
  const unsigned lrow = tid >> 3;
  unsigned voff[PASSES];
#pragma unroll
  for (int q = 0; q < PASSES; ++q) voff[q] = ((lrow + 64 * q) * (unsigned)K + (((tid & 7) ^ (lrow & 7)) << 3)) * 2u;
  unsigned voa1[A2 ? 4 : 1], voa2[A2 ? 4 : 1];      // (A2) the activation pieces' lane offsets at the row pitches of the two sources
  if constexpr (A2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      voa1[q] = ((lrow + 64 * q) * (unsigned)K1 + (((tid & 7) ^ (lrow & 7)) << 3)) * 2u;
      voa2[q] = ((lrow + 64 * q) * (unsigned)p.c2 + (((tid & 7) ^ (lrow & 7)) << 3)) * 2u;
    }
  }
  const int wrow_b = wid * 8 * PBK * 2;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  const char* a_base = nullptr;
  const char* a_base2 = nullptr;
  const char* w_base = nullptr;
  int a_tile = 0, a_kt = 0, a_cnt = 0;              // activation cursor: (tile, K tile), K tiles issued
  int b_tile = 0, b_kt = 0, b_cnt = 0;              // weight cursor
  auto set_a = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(tile, m0, n0);
    a_base = reinterpret_cast<const char*>(p.a1) + (int64_t)m0 * K1 * 2;
    if constexpr (A2) a_base2 = reinterpret_cast<const char*>(p.a2) + (int64_t)m0 * p.c2 * 2;
  };
  auto set_b = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(tile, m0, n0);
    w_base = reinterpret_cast<const char*>(p.w) + (int64_t)n0 * K * 2;
    if (wid < BN2 / 64) {                           // BN2 floats: one 64-float piece per wave
      const unsigned db = ((tile & 1) * BN2 + wid * 64) * 4;
      if (has_bias) PP_DMA("global_load_lds_dword", (unsigned)(lane * 4), reinterpret_cast<const char*>(p.bias + n0 + wid * 64), lds0 + D::OFF_BIAS + db);
      if constexpr (LNC) PP_DMA("global_load_lds_dword", (unsigned)(lane * 4), reinterpret_cast<const char*>(p.ln_s + n0 + wid * 64), lds0 + D::OFF_S + db);
    }
    if constexpr (LNC)                              // (mean, rstd) of the tile's 256 rows: 512 floats = one dword per lane of the block
      PP_DMA("global_load_lds_dword", (unsigned)(lane * 4), reinterpret_cast<const char*>(p.ln_stat + (int64_t)m0 * 2 + wid * 64),
             lds0 + D::OFF_STAT + ((tile & 1) * 512 + wid * 64) * 4);
  };
  auto issue_a = [&]() __attribute__((always_inline)) {      // the four activation pieces of the K tile at the activation cursor; the cursor moves on
    if (!no_dma) {
      const unsigned d = lds0 + (a_cnt & 1) * D::SLOT + wrow_b;
      if constexpr (A2) {
        const bool second = a_kt >= nk1;
        const char* g = second ? a_base2 + (a_kt - nk1) * (PBK * 2) : a_base + a_kt * (PBK * 2);
        const unsigned v0 = second ? voa2[0] : voa1[0], v1 = second ? voa2[1] : voa1[1], v2 = second ? voa2[2] : voa1[2], v3 = second ? voa2[3] : voa1[3];
        PP_DMA("global_load_lds_dwordx4", v0, g, d);
        PP_DMA("global_load_lds_dwordx4", v1, g, d + 1 * (64 * PBK * 2));
        PP_DMA("global_load_lds_dwordx4", v2, g, d + 2 * (64 * PBK * 2));
        PP_DMA("global_load_lds_dwordx4", v3, g, d + 3 * (64 * PBK * 2));
      } else {
      const char* g = a_base + a_kt * (PBK * 2);
      PP_DMA("global_load_lds_dwordx4", voff[0], g, d);
      PP_DMA("global_load_lds_dwordx4", voff[1], g, d + 1 * (64 * PBK * 2));
      PP_DMA("global_load_lds_dwordx4", voff[2], g, d + 2 * (64 * PBK * 2));
      PP_DMA("global_load_lds_dwordx4", voff[3], g, d + 3 * (64 * PBK * 2));
      }
    }
    ++a_cnt;
    if (++a_kt == nk) {
      a_kt = 0;
      if (++a_tile < my_tiles) set_a(a_tile);
    }
  };
  auto issue_b = [&](auto q0_tag, auto q1_tag) __attribute__((always_inline)) {   // weight row passes [q0, q1) of the K tile at the weight cursor
    constexpr int q0 = decltype(q0_tag)::value, q1 = decltype(q1_tag)::value;
    if (!no_dma) {
      const char* g = w_base + b_kt * (PBK * 2);
      const unsigned d = lds0 + (b_cnt & 1) * D::SLOT + D::A_BYTES + wrow_b;
#pragma unroll
      for (int q = q0; q < q1; ++q) PP_DMA("global_load_lds_dwordx4", voff[q], g, d + q * (64 * PBK * 2));
    }
    if constexpr (q1 == PASSES) {
      ++b_cnt;
      if (++b_kt == nk) {
        b_kt = 0;
        if (++b_tile < my_tiles) set_b(b_tile);
      }
    }
  };

  // ---- compute side
  f32x4 acc0[PMT][NT], acc1[PMT][NT];              // the two halves (zeroed by their epilogue: a "first cluster takes C = 0" variant of the clusters doubles
#pragma unroll                                     // the loop body and sends hipcc's allocator into spilling)
  for (int i = 0; i < PMT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc0[i][j] = acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fa[PMT] = {}, fb[NT] = {};
  unsigned a_rd[2], b_rd[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_rd[kk] = ((wm * 64 + fr) * PBK + (((kk * 4 + fq) ^ (fr & 7)) << 3)) * 2;
    b_rd[kk] = D::A_BYTES + ((wn * WN + fr) * PBK + (((kk * 4 + fq) ^ (fr & 7)) << 3)) * 2;
  }
  auto read_a = [&](int slot, auto kk_tag) __attribute__((always_inline)) {
    const char* ba = smem + slot * D::SLOT + a_rd[decltype(kk_tag)::value];
#pragma unroll
    for (int i = 0; i < PMT; ++i) fa[i] = *reinterpret_cast<const u32x4*>(ba + i * (16 * PBK * 2));
  };
  auto read_b = [&](int slot, auto kk_tag, auto nh_tag) __attribute__((always_inline)) {
    const char* bb = smem + slot * D::SLOT + b_rd[decltype(kk_tag)::value] + decltype(nh_tag)::value * (HN * PBK * 2);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const u32x4*>(bb + j * (16 * PBK * 2));
  };
  auto cluster = [&](auto nh_tag) __attribute__((always_inline)) {
    constexpr int NH = decltype(nh_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < PMT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (NH == 0) acc0[i][j] = PMfma<T>::run(__builtin_bit_cast(frag, fb[j]), __builtin_bit_cast(frag, fa[i]), acc0[i][j]);
        else acc1[i][j] = PMfma<T>::run(__builtin_bit_cast(frag, fb[j]), __builtin_bit_cast(frag, fa[i]), acc1[i][j]);
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- epilogue of one half (igemm.hip's fast-path arithmetic).  Lane: pixel row m = m0 + wm * 64 + i * 16 + fr, channels n0 + wn * WN + j * 16 + fq * 4 .. + 3
  T* const out = reinterpret_cast<T*>(p.out);
  const T* const res = reinterpret_cast<const T*>(p.residual);
  // NB 16-channel blocks of one pixel row: 16-byte stores after a lane swap between adjacent blocks (igemm.hip store_row_group); hm_d / hm_skip: columns
  // >= hm_d of the span belong to the next head, whose plane starts hm_skip elements further on
  auto store_blocks = [&](T* prow, auto& po, auto nb_tag, int hm_d, int hm_skip) __attribute__((always_inline)) {
    constexpr int NB = decltype(nb_tag)::value;
    if (no_store) return;
    auto at = [&](int col) __attribute__((always_inline)) { return prow + col + (col >= hm_d ? hm_skip : 0); };
    if (fake_store) {   // timing only (debug bit 64): the same number of 16-byte stores, each instruction one contiguous KB inside the tile's output rows
#pragma unroll
      for (int k = 0; k + 1 < NB; k += 2) {
        const u32x4 v = {po[k][0], po[k][1], po[k + 1][0], po[k + 1][1]};
        *reinterpret_cast<u32x4*>(prow + (k >> 1) * 512) = v;
      }
      if constexpr (NB & 1) *reinterpret_cast<u32x2*>(prow + (NB >> 1) * 512) = po[NB - 1];
      return;
    }
#pragma unroll
    for (int k = 0; k + 1 < NB; k += 2) {
      const auto lo = __builtin_amdgcn_permlane16_swap(po[k][0], po[k + 1][0], false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap(po[k][1], po[k + 1][1], false, false);
      const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
      *reinterpret_cast<u32x4*>(at((k + (fq & 1)) * 16 + (fq >> 1) * 8)) = v;
    }
    if constexpr (NB & 1) *reinterpret_cast<u32x2*>(at((NB - 1) * 16 + fq * 4)) = po[NB - 1];
  };
  auto epilogue_half = [&](auto nh_tag, int tile) __attribute__((always_inline)) {
    constexpr int NH = decltype(nh_tag)::value;
    int m0, n0;
    tile_origin(tile, m0, n0);
    n0 += NH * HN;
    const int cofs = (tile & 1) * BN2 + NH * HN + wn * WN + fq * 4;   // the lane's first column inside the staged vectors
    const float* tb = sBias + cofs;                 // (bias / c re-read from LDS per row group: 20 registers less across the half)
    const float* ts = sS + cofs;
    const int mrow = m0 + wm * 64 + fr;
    if constexpr (EPI == 3) {
      // GEGLU: wave columns [0, WN/2) hold a, [WN/2, WN) the matching gate; output width N / 2
      const int No = N >> 1;
#pragma unroll
      for (int i = 0; i < PMT; ++i) {
        const f32x2 st = *reinterpret_cast<const f32x2*>(sStat + (tile & 1) * 512 + (wm * 64 + i * 16 + fr) * 2);
        u32x2 po[NT / 2];
#pragma unroll
        for (int j = 0; j < NT / 2; ++j) {
          f32x4 a = NH == 0 ? acc0[i][j] : acc1[i][j], g = NH == 0 ? acc0[i][j + NT / 2] : acc1[i][j + NT / 2];
          if constexpr (NH == 0) { acc0[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc0[i][j + NT / 2] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
          else { acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[i][j + NT / 2] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
          a = (a - st[0] * *reinterpret_cast<const f32x4*>(ts + j * 16)) * st[1];
          g = (g - st[0] * *reinterpret_cast<const f32x4*>(ts + j * 16 + WN / 2)) * st[1];
          a += *reinterpret_cast<const f32x4*>(tb + j * 16);
          g += *reinterpret_cast<const f32x4*>(tb + j * 16 + WN / 2);
          gelu_f32x2 g01 = {g[0], g[1]}, g23 = {g[2], g[3]};
          if (!no_gelu) { g01 = gelu_pair(g01); g23 = gelu_pair(g23); }
          T o[4] = {from_f32<T>(a[0] * g01[0]), from_f32<T>(a[1] * g01[1]), from_f32<T>(a[2] * g23[0]), from_f32<T>(a[3] * g23[1])};
          po[j] = *reinterpret_cast<u32x2*>(o);
        }
        store_blocks(out + (int64_t)(mrow + i * 16) * No + ((n0 + wn * WN) >> 1), po, std::integral_constant<int, NT / 2>{}, 1 << 30, 0);
      }
    } else if constexpr (STAT == 2) {
      // GroupNorm producer (proj_out of the transformer blocks): per channel, sum and sum of squares of the stored values over the 64 rows of the wave tile
      // -> stat_out[row tile][0 / 1][channel] (igemm.hip's gn_emit: row groups 0 .. 3, then the 16 pixel lanes by DPP -- identical partials).  By column-
      // block pairs (0, 1), (2, 3), (4) -- the pairs of the 16-byte stores: 16 registers of sums at a time, the next pair's residual in flight
      const int64_t lane_off = (int64_t)mrow * N + n0 + wn * WN;
      const int64_t rt = (int64_t)(m0 + wm * 64) / 64;
      float* gs = p.stat_out;
      u32x2 rva[PMT][2], rvb[PMT][2];
      auto load_res = [&](auto& rv, auto j0_tag, auto nb_tag) __attribute__((always_inline)) {
        constexpr int j0 = decltype(j0_tag)::value, NB = decltype(nb_tag)::value;
#pragma unroll
        for (int i = 0; i < PMT; ++i)
#pragma unroll
          for (int jj = 0; jj < NB; ++jj) rv[i][jj] = *reinterpret_cast<const u32x2*>(res + lane_off + (int64_t)i * 16 * N + (j0 + jj) * 16 + fq * 4);
      };
      auto do_pair = [&](auto& rv, auto j0_tag, auto nb_tag) __attribute__((always_inline)) {
        constexpr int j0 = decltype(j0_tag)::value, NB = decltype(nb_tag)::value;
        f32x4 sm[NB], sq[NB];
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) sm[jj] = sq[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PMT; ++i) {
          u32x2 po[NB];
#pragma unroll
          for (int jj = 0; jj < NB; ++jj) {
            f32x4 v = NH == 0 ? acc0[i][j0 + jj] : acc1[i][j0 + jj];
            if constexpr (NH == 0) acc0[i][j0 + jj] = (f32x4){0.f, 0.f, 0.f, 0.f}; else acc1[i][j0 + jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (has_bias) v += *reinterpret_cast<const f32x4*>(tb + (j0 + jj) * 16);
            if constexpr (RES) {
              T r[4];
              *reinterpret_cast<u32x2*>(r) = rv[i][jj];
              v[0] += to_f32(r[0]); v[1] += to_f32(r[1]); v[2] += to_f32(r[2]); v[3] += to_f32(r[3]);
            }
            T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
            po[jj] = *reinterpret_cast<u32x2*>(o);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x = to_f32(o[q]);
              sm[jj][q] += x;
              sq[jj][q] += x * x;
            }
          }
          T* prow = out + lane_off + (int64_t)i * 16 * N;
          if constexpr (NB == 2) {
            const auto lo = __builtin_amdgcn_permlane16_swap(po[0][0], po[1][0], false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap(po[0][1], po[1][1], false, false);
            const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
            *reinterpret_cast<u32x4*>(prow + (j0 + (fq & 1)) * 16 + (fq >> 1) * 8) = v;
          } else {
            *reinterpret_cast<u32x2*>(prow + j0 * 16 + fq * 4) = po[0];
          }
        }
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sm[jj][q] = dualn_row_sum16(sm[jj][q]);
            sq[jj][q] = dualn_row_sum16(sq[jj][q]);
          }
          const int n = n0 + wn * WN + (j0 + jj) * 16 + fq * 4;
          if (fr == 0) {
            *reinterpret_cast<f32x4*>(gs + (rt * 2 + 0) * N + n) = sm[jj];
            *reinterpret_cast<f32x4*>(gs + (rt * 2 + 1) * N + n) = sq[jj];
          }
        }
      };
      typedef std::integral_constant<int, 0> J0;
      typedef std::integral_constant<int, 2> J2;
      typedef std::integral_constant<int, 4> J4;
      typedef std::integral_constant<int, 1> N1;
      typedef std::integral_constant<int, 2> N2;
      static_assert(NT == 5, "pairs (0, 1), (2, 3), (4)");
      if constexpr (RES) { load_res(rva, J0{}, N2{}); load_res(rvb, J2{}, N2{}); }
      do_pair(rva, J0{}, N2{});
      if constexpr (RES) load_res(rva, J4{}, N1{});
      do_pair(rvb, J2{}, N2{});
      do_pair(rva, J4{}, N1{});
    } else {
      const int64_t lane_off = (int64_t)mrow * N + n0 + wn * WN;
      u32x2 rv[PMT][NT];
      if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < PMT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) rv[i][j] = *reinterpret_cast<const u32x2*>(res + lane_off + (int64_t)i * 16 * N + j * 16 + fq * 4);
      }
      // head-major QKV planes (igemm.hip, LN == 4): the wave's 80-column span = one head (d = 80) or two (d = 40); the tile lies inside one batch row
      T* plane = nullptr;
      int hm_tok0 = 0;
      if constexpr (EPI == 2) {
        const int g = ((n0 + wn * WN) / 80) * (p.hm_dim == 40 ? 2 : 1);   // first head of the span, counted over q | k | v
        const int part = g >> 3, head0 = g & 7;
        const int b = m0 / p.hm_tokens;
        hm_tok0 = b * p.hm_tokens;
        plane = out + ((int64_t)part * p.M + (int64_t)b * p.hm_tokens) * (8 * p.hm_dim) + (int64_t)head0 * p.hm_tokens * p.hm_dim;
      }
#pragma unroll
      for (int i = 0; i < PMT; ++i) {
        f32x2 st = {0.f, 1.f};
        if constexpr (LNC) st = *reinterpret_cast<const f32x2*>(sStat + (tile & 1) * 512 + (wm * 64 + i * 16 + fr) * 2);
        u32x2 po[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          f32x4 v = NH == 0 ? acc0[i][j] : acc1[i][j];
          if constexpr (NH == 0) acc0[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; else acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if constexpr (LNC) {
            v = (v - st[0] * *reinterpret_cast<const f32x4*>(ts + j * 16)) * st[1] + *reinterpret_cast<const f32x4*>(tb + j * 16);
          } else {
            if (has_bias) v += *reinterpret_cast<const f32x4*>(tb + j * 16);
          }
          if constexpr (RES) {
            T r[4];
            *reinterpret_cast<u32x2*>(r) = rv[i][j];
            v[0] += to_f32(r[0]); v[1] += to_f32(r[1]); v[2] += to_f32(r[2]); v[3] += to_f32(r[3]);
          }
          T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
          po[j] = *reinterpret_cast<u32x2*>(o);
        }
        if (fake_store)
          store_blocks(out + (int64_t)m0 * N + wid * 10240 + (NH * PMT + i) * 1536 + lane * 8, po, std::integral_constant<int, NT>{}, 1 << 30, 0);
        else if constexpr (EPI == 2)
          store_blocks(plane + (int64_t)(mrow + i * 16 - hm_tok0) * p.hm_dim, po, std::integral_constant<int, NT>{}, p.hm_dim, (p.hm_tokens - 1) * p.hm_dim);
        else
          store_blocks(out + lane_off + (int64_t)i * 16 * N, po, std::integral_constant<int, NT>{}, 1 << 30, 0);
        if constexpr (STAT == 1) {
          // (mean, M2) of the 20 stored values of this lane, merged over the four fq lanes by Chan's update in igemm.hip's order (even 16-lane row first, then
          // the lower half first): partial n0 / 80 + wn of row m.  The exchanges go through ds_bpermute (__shfl_xor): inline asm with register outputs makes
          // hipcc spill in this kernel (see pp_gemm_applicable)
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            T o[4];
            *reinterpret_cast<u32x2*>(o) = po[j];
            sum += (to_f32(o[0]) + to_f32(o[1])) + (to_f32(o[2]) + to_f32(o[3]));
          }
          float mu = sum * (1.0f / (float)(NT * 4)), m2 = 0.f;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            T o[4];
            *reinterpret_cast<u32x2*>(o) = po[j];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float d = to_f32(o[q]) - mu; m2 += d * d; }
          }
          {
            const float mo = __shfl_xor(mu, 16, 64), qo = __shfl_xor(m2, 16, 64);
            const bool odd = fq & 1;
            float ma = odd ? mo : mu, mb = odd ? mu : mo, qa = odd ? qo : m2, qb = odd ? m2 : qo;
            float d = mb - ma;
            ma += 0.5f * d;
            qa += qb + d * d * (0.5f * (float)(NT * 4));
            const float mo2 = __shfl_xor(ma, 32, 64), qo2 = __shfl_xor(qa, 32, 64);
            const bool hi = fq >> 1;
            float m_lo = hi ? mo2 : ma, m_hi = hi ? ma : mo2, q_lo = hi ? qo2 : qa, q_hi = hi ? qa : qo2;
            d = m_hi - m_lo;
            mu = m_lo + 0.5f * d;
            m2 = q_lo;
            m2 += q_hi + d * d * (0.5f * (float)(2 * NT * 4));
          }
          if (fq == 0) *reinterpret_cast<f32x2*>(p.stat_out + ((int64_t)(mrow + i * 16) * p.stat_P + n0 / 80 + wn) * 2) = (f32x2){mu, m2};
        }
      }
    }
  };

  // ---- prologue: K tile 0 whole, the activation pieces of K tile 1; wait for K tile 0
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  constexpr int QA = 2, QB = PASSES == 5 ? 4 : 3;   // weight passes [0, QA) in phase 0, [QA, QB) in phase 1, [QB, PASSES) in phase 2
  set_a(0);
  set_b(0);
  issue_a();
  issue_b(I0{}, std::integral_constant<int, PASSES>{});
  if (total_steps > 1) { issue_a(); PP_VMCNT(4); } else { PP_VMCNT(0); }
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();

  int ct_kt = 0, ct_tile = 0;
  for (int s = 0; s < total_steps; ++s) {
    const int slot = s & 1;
    const bool next = s + 1 < total_steps;          // a K tile s + 1 exists: its weight pieces go out in phases 0 .. 2 of this K tile
    // (Tried and removed, round 5: waiting LATER for an epilogue's stores -- the next K tile's weight pieces issued in front of the epilogue and phase 3
    // waiting with vmcnt(<stores>) instead of vmcnt(0) -- and, for GEGLU, holding the converted outputs in registers and storing them two per K tile
    // under the next tile's main loop.  Both bit-identical, both +-0 %: the waves are held at the ISSUE of the stores and the store path is in order
    // with the DMA pieces either way; profiles/r05_dualn_store_ablation.log.)
    // (Tried and removed, round 6: both groups' epilogues in the SAME barrier interval -- E waits one barrier for L's last cluster, L one barrier behind its
    // epilogue -- so that two waves per SIMD issue the ~1000 epilogue instructions side by side instead of one after the other: bit-identical, 1-3 % SLOWER
    // on every GEGLU / Linear shape (profiles/r06_dualn_concurrent_epilogue_ab.log): the epilogue is not bound by the single-wave issue rate.)
    if (s > 0 && ct_kt == 0 && !no_epi) {           // the previous K tile finished an output tile: both halves, in front of this tile's first cluster
      epilogue_half(I0{}, ct_tile - 1);
      epilogue_half(I1{}, ct_tile - 1);
    }
    const bool issue_w = next;
    // (Order inside a MEM phase -- DMA issue first or fragment reads first -- measured both ways on MI355X: +-2-4 % per shape, +-0 on the benchmark.)
    // ---- phase 0: (kk 0, half 0)
    if (issue_w) issue_b(I0{}, std::integral_constant<int, QA>{});
    read_a(slot, I0{});
    read_b(slot, I0{}, I0{});
    PP_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster(I0{});
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: (kk 0, half 1)
    if (issue_w) issue_b(std::integral_constant<int, QA>{}, std::integral_constant<int, QB>{});
    read_b(slot, I0{}, I1{});
    PP_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster(I1{});
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: (kk 1, half 0)
    if (issue_w) issue_b(std::integral_constant<int, QB>{}, std::integral_constant<int, PASSES>{});
    read_a(slot, I1{});
    read_b(slot, I1{}, I0{});
    PP_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster(I0{});
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: (kk 1, half 1): K tile s + 1 has landed (for this wave); the activation pieces of K tile s + 2 go into the rows just read
    PP_VMCNT(0);
    if (s + 2 < total_steps) issue_a();
    read_b(slot, I1{}, I1{});
    PP_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster(I1{});
    __builtin_amdgcn_s_barrier();
    if (++ct_kt == nk) { ct_kt = 0; ++ct_tile; }
  }
  if (!no_epi) {
    epilogue_half(I0{}, ct_tile - 1);
    epilogue_half(I1{}, ct_tile - 1);
  }
  if (!late) __builtin_amdgcn_s_barrier();
}

}  // namespace

#ifdef ETAINV_EXPERIMENTS
// 1x1 / Linear launches on whole tiles with K >= 320 and a bias-only epilogue.  OPT-IN (ETAINV_PP=1), measured on MI355X (profiles/r05_pp_gemm_check.log):
// bit-identical to the ring kernel and 3-10 % faster on the shapes it takes -- but the layers that matter carry a residual (+ LayerNorm statistics), and
// the residual needs register-destination loads that hipcc must not count: every form of inline asm with a VGPR output inside the slice makes the
// allocator spill 130-270 registers (plain loads compile to 245 registers but are waited for with vmcnt(0), which drains the DMA queue four times per
// tile).  The RES / STAT instantiations below are therefore refused by the predicate (ETAINV_PP_FORCE_ALL=1 lets them through: spilling, 4-7x slower -- evidence only).
bool pp_gemm_applicable(const IGemmParams& p, int dtype) {
  static const int mode = env_on("ETAINV_PP_FORCE_ALL") ? 2 : 1;
  if (!env_on("ETAINV_PP") || (dtype != ETAINV_F16 && dtype != ETAINV_BF16)) return false;
  if (mode < 2 && (p.residual || p.stat_out)) return false;
  if (p.taps != 1 || p.a2 || p.geglu || p.rowvec || p.out_f32 || p.out_nchw || p.ln_stat || p.w_batch_stride || p.ksplit > 1 || p.hm_heads) return false;
  if (p.stat_out && (p.stat_kind != 0 || p.rows_per_batch % 64 != 0)) return false;
  if (p.M % PBM != 0 || p.N % PBN != 0 || p.c1 % PBK != 0 || p.c1 < 5 * PBK) return false;
  return (int64_t)(p.M / PBM) * (p.N / PBN) >= 512;   // at least two tiles per block: the overlapped epilogue is the point
}

#else
bool pp_gemm_applicable(const IGemmParams&, int) { return false; }   // (the default library does not carry the double-accumulator experiment)
#endif

// dual-N kernel: 1x1 / Linear on whole tiles -- 256 x 320 with bias (+ residual) (+ LayerNorm row statistics) or as a LayerNorm consumer (row-major or
// head-major QKV planes), 256 x 256 for the LayerNorm-consumer GEGLU projection; ETAINV_DUALN=0 switches it off.  Returns the epilogue kind or -1
static int dualn_kind(const IGemmParams& p, int dtype) {
  if (!env_flag("ETAINV_DUALN", true) || (dtype != ETAINV_F16 && dtype != ETAINV_BF16)) return -1;
  if (p.taps != 1 || p.rowvec || p.out_f32 || p.out_nchw || p.w_batch_stride || p.ksplit > 1) return -1;
  // two sources (the shortcut 1x1 over [hidden | skip]): the plain bias epilogue only
  if (p.a2 && (p.geglu || p.ln_stat || p.residual || p.stat_out || p.hm_heads || p.c2 % PBK != 0 || p.c2 <= 0 || !env_flag("ETAINV_DUALN_A2", true))) return -1;
  if (p.stat_out && p.rows_per_batch % 64 != 0) return -1;     // (LayerNorm rows: partial index per wave tile; GroupNorm: whole wave tiles inside one image)
  if (p.M % PBM != 0 || p.c1 % PBK != 0 || p.c1 < 2 * PBK) return -1;
  int kind;
  if (p.geglu) {
    if (!p.ln_stat || !p.ln_s || !p.bias || p.residual || p.stat_out || p.hm_heads || p.N % 256 != 0) return -1;
    kind = 3;
  } else if (p.ln_stat) {
    if (!p.ln_s || !p.bias || p.residual || p.stat_out || p.N % 320 != 0) return -1;
    kind = p.hm_heads ? 2 : 1;
    if (kind == 2 && ((p.hm_dim != 40 && p.hm_dim != 80) || p.hm_heads != 8 || p.N != 3 * 8 * p.hm_dim || p.hm_tokens % 256 != 0 || p.M % p.hm_tokens != 0)) return -1;
  } else {
    if (p.hm_heads || p.N % 320 != 0) return -1;
    kind = 0;
  }
  // enough tiles, and a last round of the persistent grid that is at least 80 % full (a half-empty last round gives the -31 % DMA bytes back)
  static const int min_tiles = getenv("ETAINV_DUALN_MIN_TILES") ? atoi(getenv("ETAINV_DUALN_MIN_TILES")) : 192;
  const int64_t tiles = (int64_t)(p.M / PBM) * (p.N / (kind == 3 ? 256 : 320));
  const int64_t rounds = (tiles + 255) / 256;
  if (tiles < min_tiles || tiles * 5 < rounds * 256 * 4) return -1;
  return kind;
}
bool pp_dualn_applicable(const IGemmParams& p, int dtype) { return dualn_kind(p, dtype) >= 0; }
bool pp_dualn_hm_ok(const IGemmParams& p, int dtype) { return dualn_kind(p, dtype) == 2; }

template <typename T, int HN, int EPI, bool RES, int STAT, bool A2 = false>
static void launch_dualn_t(const IGemmParams& p, int grid, hipStream_t s) {
  static bool attr_set[kMaxDevices] = {};
  const int dev = current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_dualn_kernel<T, HN, EPI, RES, STAT, A2>), hipFuncAttributeMaxDynamicSharedMemorySize, DualN<HN>::LDS);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((pp_dualn_kernel<T, HN, EPI, RES, STAT, A2>), dim3(grid), dim3(512), DualN<HN>::LDS, s, p);
}

int launch_pp_dualn(const IGemmParams& p_in, int dtype, hipStream_t s, int* stat_P) {
  IGemmParams p = p_in;
  const int kind = dualn_kind(p, dtype);
  ETAINV_CHECK(kind >= 0, "not a dual-N launch (ask pp_dualn_applicable first)");
  if (p.stat_out) p.stat_P = p.stat_kind == 1 ? 64 : p.N / 80;   // GroupNorm: rows per partial block; LayerNorm: partials per row
  if (stat_P) *stat_P = p.stat_out ? p.stat_P : 0;
  const int tiles = (p.M / PBM) * (p.N / (kind == 3 ? 256 : 320));
  static const int grid_cap = getenv("ETAINV_DUALN_GRID") ? std::max(1, atoi(getenv("ETAINV_DUALN_GRID"))) : 256;   // (experiments: fewer persistent blocks than CUs)
  const int grid = std::min(tiles, grid_cap);
  ETAINV_DISPATCH_HALF(dtype, T, {
    if (kind == 3) launch_dualn_t<T, 128, 3, false, 0>(p, grid, s);
    else if (kind == 2) launch_dualn_t<T, 160, 2, false, 0>(p, grid, s);
    else if (kind == 1) launch_dualn_t<T, 160, 1, false, 0>(p, grid, s);
    else if (p.residual) {
      if (p.stat_out && p.stat_kind == 1) launch_dualn_t<T, 160, 0, true, 2>(p, grid, s);
      else if (p.stat_out) launch_dualn_t<T, 160, 0, true, 1>(p, grid, s);
      else launch_dualn_t<T, 160, 0, true, 0>(p, grid, s);
    } else {
      if (p.stat_out && p.stat_kind == 1) launch_dualn_t<T, 160, 0, false, 2>(p, grid, s);
      else if (p.stat_out) launch_dualn_t<T, 160, 0, false, 1>(p, grid, s);
      else if (p.a2) launch_dualn_t<T, 160, 0, false, 0, true>(p, grid, s);
      else launch_dualn_t<T, 160, 0, false, 0>(p, grid, s);
    }
  });
  ETAINV_LAUNCH_CHECK();
  return 0;
}

#ifdef ETAINV_EXPERIMENTS
int launch_pp_gemm(const IGemmParams& p_in, int dtype, hipStream_t s, int* stat_P) {
  IGemmParams p = p_in;
  if (p.stat_out) p.stat_P = p.N / 80;              // one (mean, M2) partial per row and wave-tile column, as the ring kernel
  if (stat_P) *stat_P = p.stat_out ? p.stat_P : 0;
  const int tiles = (p.M / PBM) * (p.N / PBN);
  const int grid = std::min(tiles, 256);
  static bool attr_set[kMaxDevices] = {};
  const int dev = current_device();
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), PLDS, s, p); };
  ETAINV_DISPATCH_HALF(dtype, T, {
    if (!attr_set[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_gemm_kernel<f16, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_gemm_kernel<f16, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_gemm_kernel<f16, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_gemm_kernel<f16, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_gemm_kernel<bf16, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_gemm_kernel<bf16, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_gemm_kernel<bf16, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_gemm_kernel<bf16, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
      attr_set[dev] = true;
    }
    if (p.residual) {
      if (p.stat_out) go(pp_gemm_kernel<T, true, true>); else go(pp_gemm_kernel<T, true, false>);
    } else {
      if (p.stat_out) go(pp_gemm_kernel<T, false, true>); else go(pp_gemm_kernel<T, false, false>);
    }
  });
  ETAINV_LAUNCH_CHECK();
  return 0;
}

#else
int launch_pp_gemm(const IGemmParams&, int, hipStream_t, int*) { ETAINV_FAIL("pp_gemm_kernel is an experiment: build with EXPERIMENTS=1"); }
#endif

}  // namespace etainv
