// Short-K GEMM with a STATIONARY activation tile and two wave groups in anti-phase ("ping-pong") for the LayerNorm-consumer projections of the
// L^2-token transformer blocks (K = C = 320): the GEGLU projection ff.net.0 (N = 2560) and the fused to_q / to_k / to_v (N = 960).
//
// Why: on the generic ring kernel (igemm.hip) these launches sit under BOTH roofs (GEGLU 619 TFLOP/s and 1.2 TB/s, QKV 593 and 2.5: profiles/
// r03): a K = 320 tile is five K steps long, every step re-streams its 256 x 64 activation slice (the same 160 KB panel once per N tile: 20 times
// for the GEGLU projection), and after five steps all eight waves leave the matrix pipe idle for an epilogue that is VALU-bound (exact-erf GELU,
// folded-LayerNorm arithmetic) -- in-kernel stamps: 1164 of 3493 cycles per K step inside the MFMA windows.
//
// Structure (one 512-thread block per CU, persistent over 128-row M tiles):
//   * the activation tile X [128 rows][320] (80 KB) is loaded ONCE per M tile and stays in LDS while the block walks all N tiles of GN columns;
//     the next M tile's X replaces it chunk by chunk during the last N tile (each 64-wide K chunk as soon as its last reader is past it), so no
//     M-tile switch is exposed;
//   * weights stream through a 4-slot ring of [GN][64] chunks (64 KB at GN = 128), ONE global chunk sequence consumed one chunk per phase, issued
//     two phases ahead by LDS-DMA with counted vmcnt waits;
//   * waves 0-3 and waves 4-7 (one wave of each group per SIMD) alternate roles per N tile: while one group runs the five MFMA phases of its tile
//     (one wave per SIMD on the matrix pipe: 32 MFMAs per phase), the other group runs the epilogue of the tile it has just accumulated, sliced
//     into the same five phases -- VALU and stores under the partner's MFMAs instead of behind them.  One s_barrier per phase serves both.
//   Accumulation order over K is the ring kernel's (chunk by chunk, two 32-deep MFMA steps each): results are bit-identical to igemm.hip's.
// Operand / accumulator conventions are igemm.hip's: weight fragment = MFMA A operand, activation fragment = B operand, so a lane holds 4 consecutive
// output channels of one pixel row; LDS rows are 8 chunks of 16 B with physical chunk = chunk ^ (row & 7); GEGLU weights are packed per 64 physical
// columns as 32 value rows followed by their 32 gate rows (launch_pack_weight mode 2).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace etainv {

namespace {

template <typename T> struct XMfma;
template <> struct XMfma<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct XMfma<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

struct XsParams {
  const void* x;         // [M][K] activations (raw rows: the LayerNorm is folded)
  const void* w;         // [N][K] gamma-scaled weights (GEGLU: value / gate rows interleaved per 64)
  const float* c;        // [N] folded bias  (beta W^T + b)
  const float* s;        // [N] row sums of the rounded weights
  const float* stat;     // [M][2] (mean, rstd)
  void* out;             // [M][N]  (GEGLU: [M][N/2])
  int M, N;
  unsigned long long* stamps;   // diagnostic build (-DXS_STAMPS) only: [block][wave][8] cycle sums
};

#define XS_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
#define XS_LGKMCNT0() __builtin_amdgcn_s_waitcnt(0xC07F)

constexpr int XS_BM = 128, XS_BK = 64;

template <typename T, int K, int GN, bool GEGLU>
__global__ void __launch_bounds__(512, 1) xs_gemm_kernel(XsParams p) {
  typedef typename XMfma<T>::frag frag;
  constexpr int NK = K / XS_BK;              // K chunks = phases per tile
  constexpr int WN = GN / 2;                 // wave tile: 64 rows x WN columns (2 x 2 waves per group)
  constexpr int MT = 4, NT = WN / 16;
  constexpr int NPW = GN / 32;               // weight DMA pieces (8 rows x 128 B) per wave and chunk: GN / 8 pieces over the 4 waves of a group
  constexpr int XPW = XS_BM / 32;            // pieces per wave of one X chunk (16 pieces over 4 waves)
  constexpr int SLOT = GN * XS_BK;           // elements per ring slot
  constexpr int XCH = XS_BM * XS_BK;         // elements per X chunk
  static_assert(MT == 4 && NK >= 4 && NK <= 8 && MT < NK, "the epilogue is sliced into MT of the NK phases");
  static_assert(GN % 32 == 0 && (!GEGLU || WN % 32 == 0), "wave tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sX = reinterpret_cast<T*>(smem);        // [NK][128][64]
  T* sW = sX + NK * XCH;                     // [4][GN][64]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int grp = __builtin_amdgcn_readfirstlane(wid >> 2);   // wave-uniform role
  const int w4 = wid & 3, wm = w4 >> 1, wn = w4 & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int ntl = p.N / GN;                                      // N tiles per M tile (even: the launcher checks)
  const int num_mt = p.M / XS_BM;
  const int my_mt = (num_mt - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_mt <= 0) return;
  const int total_tiles = my_mt * ntl;                           // flattened (M tile, N tile) sequence of this block; tile sp belongs to group sp & 1
  const T* X = reinterpret_cast<const T*>(p.x);
  const T* W = reinterpret_cast<const T*>(p.w);
  T* out = reinterpret_cast<T*>(p.out);

  // ---- DMA geometry (lane-linear 1-KiB pieces: lane l -> row l >> 3 of the piece, physical chunk l & 7 = logical chunk (l & 7) ^ (row & 7))
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lrow & 7);
  const int w4u = __builtin_amdgcn_readfirstlane(w4);
  // (source address = wave-uniform base + ONE per-lane 32-bit byte offset shared by every piece: the saddr form of global_load_lds -- 64-bit per-lane
  // addresses cost a v_mad_i64 chain and a register pair per piece)
  const unsigned lane_off = (unsigned)((lrow * K + lchunk * 8) * (int)sizeof(T));
  auto uniform_ptr = [](const void* q) __attribute__((always_inline)) {        // tell the compiler the base is wave-uniform (SGPR pair)
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto issue_w = [&](int sp, int kc) __attribute__((always_inline)) {          // chunk kc of tile sp -> ring slot (sp * NK + kc) & 3
    const int n0 = (sp % ntl) * GN;
    T* dst = sW + ((sp * NK + kc) & 3) * SLOT;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int row = (w4u + 4 * i) * 8;                                       // first row of this wave's piece
      const char* g = uniform_ptr(W + (int64_t)(n0 + row) * K + kc * XS_BK) + lane_off;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(dst + row * XS_BK), 16, 0, 0);
    }
  };
  auto issue_x = [&](int mt_idx, int kc) __attribute__((always_inline)) {      // X chunk kc of this block's mt_idx-th M tile, by the 4 waves of one group
    const int m0 = ((int)blockIdx.x + mt_idx * (int)gridDim.x) * XS_BM;
    T* dst = sX + kc * XCH;
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
      const int row = (w4u + 4 * i) * 8;
      const char* g = uniform_ptr(X + (int64_t)(m0 + row) * K + kc * XS_BK) + lane_off;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(dst + row * XS_BK), 16, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- fragments: F0 = k-half 0 of the chunk being multiplied (read one phase ahead), F1 = its k-half 1
  frag f0a[MT], f0b[NT], f1a[MT], f1b[NT];
  auto read_frags = [&](int c, int kx, int kk, frag (&fa)[MT], frag (&fb)[NT]) __attribute__((always_inline)) {   // global chunk c (ring slot c & 3), X chunk kx
    const T* tX = sX + kx * XCH;
    const T* tW = sW + (c & 3) * SLOT;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * WN + j * 16 + fr;
      fb[j] = *reinterpret_cast<const frag*>(tW + row * XS_BK + (((kk * 4 + fq) ^ (row & 7)) << 3));
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * 64 + i * 16 + fr;
      fa[i] = *reinterpret_cast<const frag*>(tX + row * XS_BK + (((kk * 4 + fq) ^ (row & 7)) << 3));
    }
  };
  auto mfma_range = [&](const frag (&fa)[MT], const frag (&fb)[NT], auto lo_tag, auto hi_tag) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
#pragma unroll
    for (int idx = LO; idx < HI; ++idx) {
      const int i = idx / NT, j = idx - (idx / NT) * NT;
      acc[i][j] = XMfma<T>::run(fb[j], fa[i], acc[i][j]);
    }
  };
  typedef std::integral_constant<int, 0> I0;
  constexpr int NM = MT * NT;                                                  // MFMAs per k-half
  constexpr int NA = (MT + NT + 2 < NM) ? MT + NT + 2 : NM;                    // MFMAs on F0 in front of the rendezvous (they carry the F1 reads)
  typedef std::integral_constant<int, NA> INA;
  typedef std::integral_constant<int, NM> INM;

  // ---- epilogue state of the group's finished tile
  float ln_mean[MT], ln_rstd[MT];
  f32x4 ev_s[NT], ev_c[NT];
  int ep_m0 = 0, ep_n0 = 0;
  auto epi_loads = [&](int sp) __attribute__((always_inline)) {                 // tile sp: geometry, s / c vectors, row statistics (issued a phase and a half early)
    const int mt_idx = sp / ntl, t = sp - mt_idx * ntl;
    ep_m0 = ((int)blockIdx.x + mt_idx * (int)gridDim.x) * XS_BM;
    ep_n0 = t * GN;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = ep_n0 + wn * WN + j * 16 + fq * 4;
      ev_s[j] = *reinterpret_cast<const f32x4*>(p.s + n);
      ev_c[j] = *reinterpret_cast<const f32x4*>(p.c + n);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(p.stat + (int64_t)(ep_m0 + wm * 64 + i * 16 + fr) * 2);
      ln_mean[i] = v[0];
      ln_rstd[i] = v[1];
    }
  };
  // One 16-row group of the wave tile: folded LayerNorm, (GEGLU,) rounding, lane swap to 8 consecutive channels per lane, 16-byte stores.  Two
  // halves, one on each side of the phase's rendezvous; the stores all sit in the second half (S1 per row group: the counted waits rely on it).
  constexpr int NB = GEGLU ? NT / 2 : NT;        // 16-channel output blocks per row group
  constexpr int NB_A = (NB + 1) / 2;             // blocks computed in the first half
  constexpr int S1 = (NB + 1) / 2;               // store instructions per row group
  u32x2 po[NB];
  auto epi_block = [&](int i, int j) __attribute__((always_inline)) {
    if constexpr (GEGLU) {
      f32x4 a = acc[i][j], g = acc[i][j + NB];
      a = (a - ln_mean[i] * ev_s[j]) * ln_rstd[i];
      g = (g - ln_mean[i] * ev_s[j + NB]) * ln_rstd[i];
      a += ev_c[j];
      g += ev_c[j + NB];
      acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[i][j + NB] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const gelu_f32x2 g01 = gelu_pair((gelu_f32x2){g[0], g[1]}), g23 = gelu_pair((gelu_f32x2){g[2], g[3]});
      T o[4] = {from_f32<T>(a[0] * g01[0]), from_f32<T>(a[1] * g01[1]), from_f32<T>(a[2] * g23[0]), from_f32<T>(a[3] * g23[1])};
      po[j] = *reinterpret_cast<u32x2*>(o);
    } else {
      const f32x4 v = (acc[i][j] - ln_mean[i] * ev_s[j]) * ln_rstd[i] + ev_c[j];
      acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
      po[j] = *reinterpret_cast<u32x2*>(o);
    }
  };
  auto epi_part_a = [&](int i) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NB_A; ++j) epi_block(i, j);
  };
  auto epi_part_b = [&](int i) __attribute__((always_inline)) {
#pragma unroll
    for (int j = NB_A; j < NB; ++j) epi_block(i, j);
    const int m = ep_m0 + wm * 64 + i * 16 + fr;
    T* prow = GEGLU ? out + (int64_t)m * (p.N >> 1) + ((ep_n0 + wn * WN) >> 1) : out + (int64_t)m * p.N + ep_n0 + wn * WN;
#pragma unroll
    for (int k = 0; k + 1 < NB; k += 2) {
      const auto lo = __builtin_amdgcn_permlane16_swap(po[k][0], po[k + 1][0], false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap(po[k][1], po[k + 1][1], false, false);
      const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
      *reinterpret_cast<u32x4*>(prow + (k + (fq & 1)) * 16 + (fq >> 1) * 8) = v;
    }
    if (NB & 1) *reinterpret_cast<u32x2*>(prow + (NB - 1) * 16 + fq * 4) = po[NB - 1];
  };

  auto rendezvous = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };
  auto wait_batch = [&](bool w, bool x) __attribute__((always_inline)) {        // leave one batch of that composition in flight
    if (!w) XS_VMCNT(0); else if (x) XS_VMCNT(NPW + XPW); else XS_VMCNT(NPW);
  };

  // ---- prologue: X of the first M tile (all 8 waves: each group loads every other chunk), the first three weight chunks (group 0)
  const int total_chunks = total_tiles * NK;
  auto issue_w_chunk = [&](int c) __attribute__((always_inline)) { const int sp_ = c / NK; issue_w(sp_, c - sp_ * NK); };
  for (int kc = grp; kc < NK; kc += 2) issue_x(0, kc);
  if (grp == 0) {
    issue_w_chunk(0);
    issue_w_chunk(1);
    issue_w_chunk(2);
  }
  XS_VMCNT(0);
  __builtin_amdgcn_sched_barrier(0);
  rendezvous();
  if (grp == 0) read_frags(0, 0, 0, f0a, f0b);

  // ---- the phase machine.  Phase (sp, kc) = global chunk c = sp * NK + kc.
  //   MFMA group (sp & 1): F0 cluster with the F1 reads, rendezvous in its middle (chunk c + 1 has landed for every wave; slot c and X chunk kc are
  //   free), rest of the F0 cluster, F0 reads of chunk c + 1, F1 cluster.  It issues NO memory traffic: a DMA piece costs its wave 60 - 180 issue
  //   cycles, and the one wave per SIMD that feeds the matrix pipe has none to spare (measured: with the DMA in this stream a phase took 1800 cycles
  //   for 512 cycles of MFMAs).
  //   Other group: row group kc of the epilogue of tile sp - 1, half of it on each side of the same rendezvous, then the phase's DMA batch -- weight
  //   chunk c + 3 and, during the last N tile of an M tile, the next M tile's X chunk kc (its last reader is past the rendezvous) -- and in the last
  //   phase the F0 reads of its own next tile.
  // vmcnt bookkeeping (in issue order per wave): a batch issued behind the rendezvous of phase P must have landed at the rendezvous of phase P + 2.
  //   In epilogue mode that wait sits in front of the rendezvous of phases 2 .. NK-1 and leaves [row group kc-1's stores][batch kc-1] in flight; the
  //   batches of phases NK-2 and NK-1 are waited for by the same waves in MFMA mode, phases 0 and 1.
#ifdef XS_STAMPS
  // s_memtime sums per wave: MFMA role [0] window 1a, [1] waits + barrier, [2] rest of the phase; epilogue role [3] waits + first half, [4] barrier,
  // [5] second half + DMA issue, [6] phases counted per role (mfma << 32 | epilogue), [7] epi_loads
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define XS_T() __builtin_amdgcn_s_memtime()
#else
#define XS_T() 0ull
#endif
  bool b_w[NK], b_x[NK];             // composition of this wave's batches of the current epilogue-role super-phase
#pragma unroll
  for (int k = 0; k < NK; ++k) b_w[k] = b_x[k] = false;
  for (int sp = 0; sp <= total_tiles; ++sp) {
    const bool mfma_on = sp < total_tiles && (sp & 1) == grp;
    const bool epi_on = sp >= 1 && ((sp - 1) & 1) == grp;
    const bool pre_on = !mfma_on && sp + 1 < total_tiles;                      // this group multiplies tile sp + 1 next
    const int mt_idx = sp / ntl, t = sp - mt_idx * ntl;
    const bool last_n = sp < total_tiles && t == ntl - 1 && mt_idx + 1 < my_mt;
    if (mfma_on) {
      auto phase = [&](auto kc_tag, frag (&ca)[MT], frag (&cb)[NT], frag (&na)[MT], frag (&nb)[NT]) __attribute__((always_inline)) {
        constexpr int kc = decltype(kc_tag)::value;
        const int c = sp * NK + kc;
        [[maybe_unused]] const unsigned long long t0 = XS_T();
        read_frags(c, kc, 1, f1a, f1b);
        mfma_range(ca, cb, I0{}, INA{});
#pragma unroll
        for (int q = 0; q < (MT + NT + 1) / 2; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NA - (MT + NT + 1) / 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        [[maybe_unused]] const unsigned long long t1 = XS_T();
        XS_LGKMCNT0();                                  // F1 landed: slot c and X chunk kc may be recycled after the barrier
        if (kc == 0) wait_batch(b_w[NK - 1], b_x[NK - 1]);   // this wave's batch of its phase NK-2 has landed (the last one may still fly)
        if (kc == 1) XS_VMCNT(0);                       // ... and the last one
        rendezvous();
        [[maybe_unused]] const unsigned long long t2 = XS_T();
        mfma_range(ca, cb, INA{}, INM{});
        if (kc < NK - 1) read_frags(c + 1, kc + 1, 0, na, nb);
        mfma_range(f1a, f1b, I0{}, INM{});
        __builtin_amdgcn_sched_barrier(0);
#ifdef XS_STAMPS
        const unsigned long long t3 = XS_T();
        st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2; st[6] += 1ull << 32;
#endif
      };
      static_assert(NK == 5, "the unrolled phase sequence below is written for five K chunks");
      // (straight-line code, one F0 set: the reads of chunk c + 1 are issued behind the last MFMA on the old F0 values and overwrite them)
      phase(std::integral_constant<int, 0>{}, f0a, f0b, f0a, f0b);
      phase(std::integral_constant<int, 1>{}, f0a, f0b, f0a, f0b);
      phase(std::integral_constant<int, 2>{}, f0a, f0b, f0a, f0b);
      phase(std::integral_constant<int, 3>{}, f0a, f0b, f0a, f0b);
      phase(std::integral_constant<int, 4>{}, f0a, f0b, f0a, f0b);
    } else {
      // (unconditional, like the F0 reads below: a conditional redefinition would keep the old values alive through the other role's loop)
      [[maybe_unused]] const unsigned long long tl0 = XS_T();
      epi_loads(sp >= 1 ? sp - 1 : 0);
#ifdef XS_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      st[7] += XS_T() - tl0;
#endif
#pragma unroll
      for (int kc = 0; kc < NK; ++kc) {
        const int c = sp * NK + kc;
        [[maybe_unused]] const unsigned long long e0 = XS_T();
        if (kc >= 2) {                                  // batch kc-2 has landed; row group kc-1's stores and batch kc-1 stay in flight
          const int young = (epi_on ? S1 : 0) + (b_w[kc - 1] ? NPW : 0) + (b_x[kc - 1] ? XPW : 0);
          if (young == 0) XS_VMCNT(0);
          else if (young == S1) XS_VMCNT(S1);
          else if (young == NPW) XS_VMCNT(NPW);
          else if (young == S1 + NPW) XS_VMCNT(S1 + NPW);
          else if (young == NPW + XPW) XS_VMCNT(NPW + XPW);
          else XS_VMCNT(S1 + NPW + XPW);
        }
        if (epi_on && kc < MT) epi_part_a(kc);
        __builtin_amdgcn_sched_barrier(0);
        [[maybe_unused]] const unsigned long long e1 = XS_T();
        rendezvous();
        [[maybe_unused]] const unsigned long long e2 = XS_T();
        if (epi_on && kc < MT) epi_part_b(kc);
        b_w[kc] = c + 3 < total_chunks;
        b_x[kc] = last_n;
        if (b_w[kc]) issue_w_chunk(c + 3);
        if (last_n) issue_x(mt_idx + 1, kc);
        if (kc == NK - 1) read_frags(pre_on ? (sp + 1) * NK : 0, 0, 0, f0a, f0b);   // F0 of this group's next tile (landed: issued 3 phases ago)
        __builtin_amdgcn_sched_barrier(0);
#ifdef XS_STAMPS
        const unsigned long long e3 = XS_T();
        st[3] += e1 - e0; st[4] += e2 - e1; st[5] += e3 - e2; st[6] += 1ull;
#endif
      }
    }
  }
#ifdef XS_STAMPS
  if (p.stamps && lane == 0)
    for (int k = 0; k < 8; ++k) p.stamps[((int64_t)blockIdx.x * 8 + wid) * 8 + k] = st[k];
#endif
}

template <typename T, int K, int GN, bool GEGLU>
int launch_xs_t(const XsParams& p, hipStream_t s) {
  constexpr size_t lds = (size_t)(K / XS_BK) * XS_BM * XS_BK * sizeof(T) + (size_t)4 * GN * XS_BK * sizeof(T);
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set[kMaxDevices] = {};
  const int dev = current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xs_gemm_kernel<T, K, GN, GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  const int num_mt = p.M / XS_BM;
  const int grid = num_mt < 256 ? num_mt : 256;
#ifdef XS_STAMPS
  if (env_on("ETAINV_XS_STAMPS")) {
    static unsigned long long* d = nullptr;
    if (!d) (void)hipMalloc(&d, 256 * 8 * 8 * sizeof(unsigned long long));
    XsParams ps = p;
    ps.stamps = d;
    hipLaunchKernelGGL((xs_gemm_kernel<T, K, GN, GEGLU>), dim3(grid), dim3(512), lds, s, ps);
    (void)hipStreamSynchronize(s);
    static unsigned long long h[256 * 8 * 8];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double nm = 0, ne = 0;
    for (int b = 0; b < grid; ++b)
      for (int w = 0; w < 8; ++w) {
        const unsigned long long* o = &h[((size_t)b * 8 + w) * 8];
        for (int k = 0; k < 8; ++k) if (k != 6) sum[k] += (double)o[k];
        nm += (double)(o[6] >> 32);
        ne += (double)(o[6] & 0xffffffffull);
      }
    fprintf(stderr, "[xs stamps N=%d geglu=%d] per phase and wave (s_memtime ticks): MFMA role w1a %.0f wait+barrier %.0f rest %.0f | epilogue role pre %.0f barrier %.0f post+dma %.0f"
                    " | epi_loads per tile %.0f\n", p.N, (int)GEGLU, sum[0] / nm, sum[1] / nm, sum[2] / nm, sum[3] / ne, sum[4] / ne, sum[5] / ne, sum[7] / (ne / 5));
    return 0;
  }
#endif
  hipLaunchKernelGGL((xs_gemm_kernel<T, K, GN, GEGLU>), dim3(grid), dim3(512), lds, s, p);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Can this LayerNorm-consumer GEMM run on the stationary-X kernel?  K = 320, whole 128-row M tiles (at least one per CU), an even number of N tiles
// of 128 (GEGLU) / 96 (plain) columns.  OPT-IN (ETAINV_XSGEMM=1): measured on MI355X the kernel is SLOWER than the ring kernels (GEGLU 1.68 vs 1.31 ms,
// QKV 0.72 vs 0.55 ms at 128 rows: profiles/r04_xsgemm_stamps.log) -- see DESIGN.md section 8 for what the in-kernel stamps say about why.
bool xs_gemm_applicable(const IGemmParams& p, int dtype) {
  if (!env_on("ETAINV_XSGEMM") || (dtype != ETAINV_F16 && dtype != ETAINV_BF16)) return false;
  if (!p.ln_stat || !p.ln_s || !p.bias || p.taps != 1 || p.a2 || p.c1 != 320 || p.residual || p.rowvec || p.out_f32 || p.out_nchw || p.stat_out ||
      p.w_batch_stride || p.ksplit > 1)
    return false;
  if (p.M % XS_BM != 0 || p.M / XS_BM < 256) return false;
  const int gn = p.geglu ? 128 : 96;
  return p.N % (2 * gn) == 0 && p.N / gn >= 4;
}

int launch_xs_gemm(const IGemmParams& p, int dtype, hipStream_t s) {
  XsParams q;
  q.x = p.a1;
  q.w = p.w;
  q.c = p.bias;
  q.s = p.ln_s;
  q.stat = p.ln_stat;
  q.out = p.out;
  q.M = p.M;
  q.N = p.N;
  q.stamps = nullptr;
  if (p.geglu) {
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_xs_t<T, 320, 128, true>(q, s)));
  } else {
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_xs_t<T, 320, 96, false>(q, s)));
  }
  return 0;
}

}  // namespace etainv
