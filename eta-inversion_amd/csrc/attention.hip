// Attention kernels for the SD1.x transformer blocks (heads = 8, head_dim 40 / 80 / 160).
//
// self_attn_kernel : flash-style softmax(Q K^T * scale) V over N = 64 .. 9216 tokens.  The N x N probability
//   matrix the reference materialises for its hooks (ptp_utils.py:238-253: 2.9 GB of traffic per sample
//   forward) never exists; the prompt-to-prompt self-replace (ptp.py:194-199: target probs := source probs,
//   i.e. softmax(Q_s K_s^T) V_t) and MasaCtrl (masactrl.py:56-72: K, V of the source sample) are expressed as
//   per-row batch-index remaps of the Q/K and V operands.
// cross_attn_kernel: 77 text keys (padded to 96), single pass; fuses the prompt-to-prompt cross edit
//   (Refine ptp.py:245-251, Reweight :261-268, time blend :212-214) and the AttentionStore accumulation of
//   the (L/4)^2-token layers (ptp.py:150-167) into the softmax epilogue.
//
// MFMA orientation (both kernels): S^T = K Q^T and O^T = V^T P^T with v_mfma_f32_16x16x32, so the QUERY
// index lives on lane&15 for scores, probabilities and output alike: row max / sum are 2 shuffles, the
// online-softmax rescale is lane-local, and the S^T accumulator registers are already the B operand of the
// PV product (keys of a 32-key step are taken in the order the accumulators hold them; V^T is read in that
// same order), so P never touches LDS.  Output: 4 consecutive head channels per lane -> 8-byte stores.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace etainv {

template <typename T> struct Frag;
template <> struct Frag<f16> {
  typedef f16x8 v8;
  typedef f16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Frag<bf16> {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

constexpr float NEG_BIG = -1.0e30f;
// Measured A/B switches (MI355X, N = 4096, d = 40, 128 rows; same device, same call): packed-FMA / v_max3 softmax 4.66 ms;
// + literal-zero first MFMA 4.66; + lazy running max (wave-uniform branch) 4.93; + bounds-free K/V load variant 5.20.
// The branchy variants lose more to the split basic blocks than the skipped instructions save: all off.
#ifndef ATT_ZERO_LITERAL
#define ATT_ZERO_LITERAL 0
#endif
#ifndef ATT_LOAD_SPLIT
#define ATT_LOAD_SPLIT 0
#endif
#ifndef ATT_LAZY
#define ATT_LAZY 0
#endif
#ifndef ATT_ABL
#define ATT_ABL 0   // timing-only ablations of self_attn40_kernel (bit 0: no running maximum, 1: no exponentials, 2: no barrier, 3: no V fragment reads, 4: no K fragment reads): never in a product build
#endif
#ifndef ATT_SPEC_MAX
#define ATT_SPEC_MAX 1   // A/B: 0 = the running maximum in every tile (rounds 2-5)
#endif
#ifndef ETAINV_QT40
#define ETAINV_QT40 4
#endif
#ifndef ETAINV_QT80
#define ETAINV_QT80 2
#endif
constexpr int SELF_QT(int d) { return d == 40 ? ETAINV_QT40 : d == 80 ? ETAINV_QT80 : 2; }

// batch-row roles for the backward layout [u_s x B, u_t x B, c_s x B, c_t x B]
__device__ __forceinline__ void row_roles(int b, int n_img, int& half, int& role, int& img) {
  half = b / (2 * n_img);
  role = (b / n_img) & 1;
  img = b % n_img;
}

// ------------------------------------------------------------------------------------------------ self
// mode: 0 plain; 1 ptp self-replace (cond target rows take Q,K of their source row); 2 masactrl (target rows
// of both halves take K,V of their source row)
template <typename T, int D, int QT>
__global__ void __launch_bounds__(256) self_attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N, int heads,
                                                        float scale_log2, int mode, int n_img, int stagger, int first_row) {
  typedef typename Frag<T>::v8 v8;
  typedef typename Frag<T>::v4 v4;
  constexpr int DP = (D + 31) / 32 * 32;
  constexpr int KS = DP / 32;
  constexpr int DT = (D + 15) / 16;
  constexpr int NCH = D / 8;
  constexpr int KV = 64;
  constexpr int KSTR = DP + 8;
  constexpr int VSTR = KV + 8;
  constexpr int NLD = (KV * NCH + 255) / 256;
  constexpr int KBUF = KV * KSTR, VBUF = DT * 16 * VSTR;
  // When D is not a multiple of 16 the padded V^T rows are free MFMA work: row D is all ones, so O^T row D = sum_k p
  // (the softmax denominator, rescaled with the same alpha as O) and the per-element VALU add disappears.
  constexpr bool ONES_ROW = (DT * 16 > D);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sK = reinterpret_cast<T*>(smem);      // [2][KV][KSTR]
  T* sVt = sK + 2 * KBUF;                  // [2][DT*16][VSTR]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, q4 = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D, C3 = 3 * C;
  int bq = b, bk = b, bv = b;
  if (mode != 0) {
    int half, role, img;
    if (first_row < 0) {   // rows [u_t, c_t, c_s] x n_img (etainv_attn_ctrl.src_exit_block): the cond target rows take Q, K of the cond source rows BEHIND them
      if (mode == 1 && b / n_img == 1) { bq = b + n_img; bk = b + n_img; }
    } else {
      row_roles(b + first_row, n_img, half, role, img);   // (first_row: the call carries rows [first_row, 4 n_img) of the [u_s,u_t,c_s,c_t] layout)
      if (mode == 1 && half == 1 && role == 1) { bq = b - n_img; bk = b - n_img; }
      if (mode == 2 && role == 1) { bk = b - n_img; bv = b - n_img; }
    }
  }
  const int q_base = blockIdx.x * (64 * QT) + wid * (16 * QT);
  // De-phase the two blocks that share a CU (one wave of each per SIMD): both run [QK^T MFMAs | softmax VALU | PV MFMAs] with the
  // same period, and started together they stay together -- matrix pipe and VALU are then each idle half of the time.
  // Blocks 256..511 of every 512 (the second resident block of each CU under round-robin dispatch) start half a tile late.
  if (stagger > 0) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if ((lin >> 8) & 1)
      for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }

  // one-time LDS init (both buffers): zero K padding columns (0 * garbage could be NaN), V^T padding rows (ones row)
  if (DP > D) {
    constexpr int PCH = (DP - D) / 8;
    for (int idx = tid; idx < 2 * KV * PCH; idx += 256) {
      const int bufi = idx / (KV * PCH), r = idx % (KV * PCH);
      const int key = r / PCH, ch = r % PCH;
      *reinterpret_cast<u32x4*>(sK + bufi * KBUF + key * KSTR + D + ch * 8) = (u32x4){0u, 0u, 0u, 0u};
    }
  }
  if (ONES_ROW) {
    for (int idx = tid; idx < 2 * (DT * 16 - D) * KV; idx += 256) {
      const int bufi = idx / ((DT * 16 - D) * KV), r = idx % ((DT * 16 - D) * KV);
      const int row = D + r / KV, key = r % KV;
      sVt[bufi * VBUF + row * VSTR + key] = (T)(row == D ? 1.0f : 0.0f);
    }
  }

  // Q fragments (B operand of S^T): lane holds Q[query fr][d = ks*32 + q4*8 .. +7]
  v8 qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int query = q_base + qt * 16 + fr;
    query = query < N ? query : N - 1;
    const T* qp = qkv + ((int64_t)bq * N + query) * C3 + h * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 32 + q4 * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (d0 < D) v = *reinterpret_cast<const u32x4*>(qp + d0);
      qf[qt][ks] = *reinterpret_cast<v8*>(&v);
    }
  }

  u32x4 rk[NLD], rv[NLD];
  const T* kbase = qkv + (int64_t)bk * N * C3 + C + h * D;
  const T* vbase = qkv + (int64_t)bv * N * C3 + 2 * C + h * D;
  auto load_kv = [&](int kv0, auto full_tag) {
    constexpr bool FULL = ATT_LOAD_SPLIT && decltype(full_tag)::value;   // every key of the tile exists: no bounds test, no zero fill
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + 256 * i;
      if (FULL) {
        if (idx < KV * NCH) {   // (lanes past the last chunk never store their registers)
          rk[i] = *reinterpret_cast<const u32x4*>(kbase + (int64_t)(kv0 + idx / NCH) * C3 + (idx % NCH) * 8);
          rv[i] = *reinterpret_cast<const u32x4*>(vbase + (int64_t)(kv0 + idx % KV) * C3 + (idx / KV) * 8);
        }
      } else {
        u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
        if (idx < KV * NCH) {
          {
            const int key = idx / NCH, ch = idx % NCH;
            if (kv0 + key < N) a = *reinterpret_cast<const u32x4*>(kbase + (int64_t)(kv0 + key) * C3 + ch * 8);
          }
          {
            const int key = idx % KV, ch = idx / KV;
            if (kv0 + key < N) c = *reinterpret_cast<const u32x4*>(vbase + (int64_t)(kv0 + key) * C3 + ch * 8);
          }
        }
        rk[i] = a;
        rv[i] = c;
      }
    }
  };
  auto store_kv = [&](int bufi) {
    T* dK = sK + bufi * KBUF;
    T* dV = sVt + bufi * VBUF;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + 256 * i;
      if (idx < KV * NCH) {
        {
          const int key = idx / NCH, ch = idx % NCH;
          *reinterpret_cast<u32x4*>(dK + key * KSTR + ch * 8) = rk[i];
        }
        {
          const int key = idx % KV, ch = idx / KV;
          const T* e = reinterpret_cast<const T*>(&rv[i]);
#pragma unroll
          for (int j = 0; j < 8; ++j) dV[(ch * 8 + j) * VSTR + key] = e[j];
        }
      }
    }
  };

  // running max of the RAW scores (scale folded into the exponent: p = exp2(s*c - m*c), one FMA + one v_exp per element)
  float m_run[QT], l_run[QT];
  f32x4 acc[QT][DT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = NEG_BIG;
    l_run[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int ntiles = (N + KV - 1) / KV;
  const int nfull = N / KV;
  if (KV <= N) load_kv(0, std::true_type{});
  else load_kv(0, std::false_type{});
  store_kv(0);
  __syncthreads();

  auto tile_body = [&](int j, auto ragged_tag) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    const int kv0 = j * KV, cur = j & 1;
    if (j + 1 < nfull) load_kv(kv0 + KV, std::true_type{});
    else if (j + 1 < ntiles) load_kv(kv0 + KV, std::false_type{});
    const T* tK = sK + cur * KBUF;
    const T* tV = sVt + cur * VBUF;

    // ---- S^T = K Q^T : s[qt][kt] holds keys kt*16 + q4*4 + r for query fr
    f32x4 s[QT][4];
#if !ATT_ZERO_LITERAL
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) s[qt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        v8 kf = *reinterpret_cast<const v8*>(tK + (kt * 16 + fr) * KSTR + ks * 32 + q4 * 8);
        // first K step: C = literal 0 (an inline-constant MFMA operand; zeroing 64 accumulator registers per tile was 6 % of
        // the loop's issue slots)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][kt] = Frag<T>::mfma(kf, qf[qt][ks], (ATT_ZERO_LITERAL && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : s[qt][kt]);
      }
    if constexpr (RAGGED) {   // only the last tile of a sequence that is not a multiple of 64 keys
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kv0 + kt * 16 + q4 * 4 + r >= N) s[qt][kt][r] = NEG_BIG;
    }

    // ---- online softmax per query (lane-local + 2 shuffles), P packed straight into PV operands.
    // VALU ops and MFMA issue share the SIMD's vector issue port (PMC: 40 % of wave cycles issuing, 79 % of that VALU), so the
    // softmax is written for instruction count: v_max3 chains and packed-fp32 FMAs for the exponent arguments.  (ATT_LAZY: the
    // stored max only moves when a query would exceed it by 2^6 -- fewer instructions, but slower, see the switches above.)
    v8 pf[QT][2];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = fmaxf(fmaxf(s[qt][0][0], s[qt][0][1]), s[qt][0][2]);
      mx = fmaxf(fmaxf(mx, s[qt][0][3]), s[qt][1][0]);
#pragma unroll
      for (int e = 5; e + 1 < 16; e += 2) mx = fmaxf(fmaxf(mx, s[qt][e >> 2][e & 3]), s[qt][(e + 1) >> 2][(e + 1) & 3]);
      mx = fmaxf(mx, s[qt][3][3]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (!ATT_LAZY || __builtin_amdgcn_ballot_w64((mx - m_run[qt]) * scale_log2 > 6.0f) != 0) {
        const float m_new = fmaxf(m_run[qt], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * scale_log2);
        m_run[qt] = m_new;
        if (!ONES_ROW) l_run[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[qt][dt] *= alpha;
      }
      const float nm = -m_run[qt] * scale_log2;
      const f32x2 sc2 = {scale_log2, scale_log2}, nm2 = {nm, nm};
      float rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 t = (f32x2){s[qt][kt][r], s[qt][kt][r + 1]} * sc2 + nm2;
          const float p0 = __builtin_amdgcn_exp2f(t[0]), p1 = __builtin_amdgcn_exp2f(t[1]);
          if (!ONES_ROW) rs += p0 + p1;
          pf[qt][kt >> 1][(kt & 1) * 4 + r] = (T)p0;
          pf[qt][kt >> 1][(kt & 1) * 4 + r + 1] = (T)p1;
        }
      if (!ONES_ROW) l_run[qt] += rs;
    }

    // ---- O^T += V^T P^T
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const T* vp = tV + (dt * 16 + fr) * VSTR + ks * 32 + q4 * 4;
        v4 lo = *reinterpret_cast<const v4*>(vp);
        v4 hi = *reinterpret_cast<const v4*>(vp + 16);
        v8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[qt][dt] = Frag<T>::mfma(vf, pf[qt][ks], acc[qt][dt]);
      }

    if (j + 1 < ntiles) store_kv(cur ^ 1);   // buffer cur^1 was last read in iteration j-1, a barrier ago
    __syncthreads();
  };
  for (int j = 0; j < nfull; ++j) tile_body(j, std::false_type{});
  if (nfull < ntiles) tile_body(nfull, std::true_type{});

  // ---- normalise and store: lane holds channels dt*16 + q4*4 + r of query fr
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l;
    if (ONES_ROW) {
      // denominator = O^T row D: tile DT-1, lane group q4 = (D % 16) / 4, register (D % 4) == 0
      l = __shfl(acc[qt][DT - 1][D % 4], fr + 16 * ((D % 16) / 4), 64);
    } else {
      l = l_run[qt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.f / l;
    const int query = q_base + qt * 16 + fr;
    if (query >= N) continue;
    T* op = out + ((int64_t)b * N + query) * C + h * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int dc = dt * 16 + q4 * 4;
      if (dc < D) {
        T o[4] = {(T)(acc[qt][dt][0] * inv), (T)(acc[qt][dt][1] * inv), (T)(acc[qt][dt][2] * inv), (T)(acc[qt][dt][3] * inv)};
        *reinterpret_cast<u32x2*>(op + dc) = *reinterpret_cast<u32x2*>(o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ self, head_dim 40 (v2)
// The 64^2- and 96^2-token levels (N = 4096 / 9216, d = 40) are 85 % of the self-attention FLOPs of the UNet and ran at 23 % of the MFMA
// peak in the generic kernel above: the loop is bound by the SIMD's vector-issue port (PMC: 40 % of wave cycles issuing, 79 % of that
// VALU), with one v_exp_f32 (8 issue cycles) per 160 FLOPs.  This kernel removes everything else from the per-element path:
//   * v_mfma_f32_32x32x16 for both products: an MFMA holds the issue port 8 of its 32 cycles (16x16x32: 8 of 16), and K = 16 steps pad
//     40 -> 48 (not 64) in S^T = K Q^T.  Query on lane & 31, the 16 accumulator registers of a 32-key block are directly the B operand
//     of two K = 16 steps of O^T = V^T P^T (keys taken in the order the accumulator holds them: step s <- registers 8s .. 8s+7, i.e.
//     key 16s + 8(j>>2) + 4h + (j&3) in element j of lane half h; V^T is read in that same order);
//   * the softmax scale * log2(e) is folded into the to_q weights by the engine (q_scale = 1; the raw-op entry point rescales Q once
//     at load), and the running reference maximum m' enters through the MFMA itself: padding dimension 40 of every K row is 1 and the
//     query operand carries -m' there, so the accumulator IS the exponent argument: p = exp2(acc), no subtract, no multiply;
//   * m' only moves when a tile exceeds it by 2^8 (wave-uniform test; the first tile always sets it): no per-tile rescale of the
//     output accumulators.  P is bounded by 2^8 (fits f16), its relative precision is scale-independent, O and the denominators
//     accumulate in fp32.  m' is kept representable in the operand type, so the value the MFMA subtracts and the value the rescale
//     uses are the same number;
//   * the softmax denominator is row 40 of O^T: the pad chunk of every V row is (1, 0, ...), the MFMA sums p for free;
//   * V stays row-major [key][48] in LDS (16-byte staging stores, no 2-byte transposition: 39 % of the LDS cycles of the generic kernel
//     were bank conflicts of those stores) and is read transposed by ds_read_b64_tr_b16.
// Per 64-key tile and 32-query block: 14 MFMAs (448 matrix cycles), 32 v_exp + 16 v_max3 + 16 v_cvt_pk (~400 issue cycles).
template <typename T> struct Frag32;
template <> struct Frag32<f16> {
  static __device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Frag32<bf16> {
  static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// geometry for head_dim D (40: the L^2-token levels; 80: the (L/2)^2 level).  Every K / V row carries a pad chunk (1, 0 x 7) at dim D.
template <int D> struct A32 {
  static constexpr int KV = 64;                              // keys per tile
  static constexpr int KS = (D + 8 + 15) / 16;               // K = 16 steps of S^T = K Q^T, pad chunk included (40 -> 3, 80 -> 6)
  static constexpr int KROW = (2 * KS + 1) * 8;              // K row in elements: an ODD number of 16-byte chunks (7 / 13): conflict-free ds_read_b128
  // V row in elements: D dims + pad chunk, rounded up to a stride of 64 bytes times an odd number -- the transposed reads of a 32-lane half
  // touch 64 bytes (two 16-dim groups) of 4 consecutive rows, and only such strides put those four segments on different banks (48 elements =
  // 96 bytes wrapped the fourth row onto the first: PMC 37 % of the kernel's LDS cycles were bank conflicts); 96 / 96
#ifndef ETAINV_A40_VROW_PLAIN
  static constexpr int VROW = ((D + 8 + 31) / 32 | 1) * 32;
#else
  static constexpr int VROW = (D + 8 + 15) / 16 * 16;
#endif
  static constexpr int DT = (D + 1 + 31) / 32;               // 32-row tiles of O^T incl. the denominator row D (2 / 3)
  static constexpr bool ZBUF = DT * 32 > VROW;               // V^T rows past VROW are read from an all-zero image (D = 40: rows 48 .. 63)
  static constexpr int KBUF = KV * KROW, VBUF = KV * VROW;
  static constexpr size_t LDS = (size_t)(2 * KBUF + (ZBUF ? 3 : 2) * VBUF) * 2;
  static constexpr int NCH = D / 8;                          // data chunks per row
  static constexpr int NLD = (KV * NCH + 255) / 256;         // staging chunks per thread and tensor
  static constexpr int MS = D / 16, MH = (D % 16) / 8;       // K step and lane half that hold the pad dimension D (where -m' enters)
  static constexpr int LT = D / 32, LR = D % 32;             // O^T tile and row of the denominator
  static constexpr int LI = (LR & 3) + 4 * (LR >> 3), LH = (LR >> 2) & 1;   // its accumulator register and lane half
};
constexpr float A40_THR = 8.0f;

// maximum of the 32 scores a lane holds for one query block: four independent v_max3 chains of 8 elements (two per 32-key score tile, so the first two can run while the
// second tile's MFMAs finish) and a 4-instruction combine -- 16 instructions, depth 6; the straight chain was 17 deep, each link waiting for the previous one's result
__device__ __forceinline__ float max32(const f32x16 (&s)[2]) {
  float c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x16& t = s[k >> 1];
    const int o = (k & 1) * 8;
    float m = fmaxf(fmaxf(t[o], t[o + 1]), t[o + 2]);
    m = fmaxf(fmaxf(m, t[o + 3]), t[o + 4]);
    c[k] = fmaxf(fmaxf(m, t[o + 5]), t[o + 6]);
  }
  const float l = fmaxf(fmaxf(s[0][7], s[0][15]), s[1][7]);
  const float m = fmaxf(fmaxf(c[0], c[1]), c[2]);
  return fmaxf(fmaxf(m, c[3]), fmaxf(l, s[1][15]));
}

template <typename T, int D, bool XCD_REMAP, int QB, int OCC>
__global__ void __launch_bounds__(256, OCC) self_attn40_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N, int heads,
                                                               float q_scale, int mode, int n_img, int nqb, int stagger, int first_row, int hm_rows) {
  // hm_rows > 0: qkv holds three head-major planes [q|k|v][hm_rows batch rows][head][token][D] (IGemmParams::hm_*): a 64-key tile is one contiguous block
  typedef typename Frag<T>::v8 v8;
  typedef A32<D> GEO;
  constexpr int KV = GEO::KV, KS = GEO::KS, KROW = GEO::KROW, VROW = GEO::VROW, KBUF = GEO::KBUF, VBUF = GEO::VBUF, DT = GEO::DT, NCH = GEO::NCH,
                NLD = GEO::NLD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sK = reinterpret_cast<T*>(smem);      // [2][KV][KROW]
  T* sV = sK + 2 * KBUF;                   // [2][KV][VROW]
  T* sZ = sV + 2 * VBUF;                   // [KV][VROW] zeros

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // block -> (query block, head, batch row).  XCD_REMAP: the nqb query blocks of one (row, head) share its K / V (655 KB at N = 4096);
  // blocks b and b + 8 run on the same XCD (round-robin dispatch), so all blocks of a (row, head) are given ids of one residue class
  // and read K / V from that XCD's L2 (speed only: any placement computes the same thing)
  int qblk, hd, b;
  if (XCD_REMAP) {
    const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
    const int set = (slot / nqb) * 8 + xcd;
    qblk = slot - (slot / nqb) * nqb;
    hd = set % heads;
    b = set / heads;
  } else {
    qblk = blockIdx.x;
    hd = blockIdx.y;
    b = blockIdx.z;
  }
  const int C = heads * D, C3 = 3 * C;
  int bq = b, bk = b, bv = b;
  if (mode != 0) {
    int half, role, img;
    if (first_row < 0) {   // rows [u_t, c_t, c_s] x n_img (etainv_attn_ctrl.src_exit_block): the cond target rows take Q, K of the cond source rows BEHIND them
      if (mode == 1 && b / n_img == 1) { bq = b + n_img; bk = b + n_img; }
    } else {
      row_roles(b + first_row, n_img, half, role, img);   // (first_row: the call carries rows [first_row, 4 n_img) of the [u_s,u_t,c_s,c_t] layout)
      if (mode == 1 && half == 1 && role == 1) { bq = b - n_img; bk = b - n_img; }
      if (mode == 2 && role == 1) { bk = b - n_img; bv = b - n_img; }
    }
  }
  const int q_base = qblk * (128 * QB) + wid * (32 * QB);
  // experiment (ETAINV_A40_STAGGER, 64-cycle ticks): delay the second co-resident block of each CU (ids 256 .. 511 of every 512 under
  // round-robin dispatch) so that its matrix phases meet the first block's softmax phases
  if (stagger > 0 && ((blockIdx.x >> 8) & 1))
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);

  // ---- one-time LDS constants: pad chunks (1, 0 x 7) of every K and V row of both buffers, the all-zero V image
  {
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    u32x4 one4 = zero4;
    {
      T one[2] = {(T)1.0f, (T)0.0f};
      one4[0] = *reinterpret_cast<unsigned*>(one);
    }
    for (int idx = tid; idx < 2 * KV; idx += 256) {
      *reinterpret_cast<u32x4*>(sK + idx * KROW + D) = one4;
#pragma unroll
      for (int c = D + 8; c < KROW; c += 8) *reinterpret_cast<u32x4*>(sK + idx * KROW + c) = zero4;
      *reinterpret_cast<u32x4*>(sV + idx * VROW + D) = one4;
#pragma unroll
      for (int c = D + 8; c < VROW; c += 8) *reinterpret_cast<u32x4*>(sV + idx * VROW + c) = zero4;
    }
    if (GEO::ZBUF)
      for (int idx = tid; idx < VBUF / 8; idx += 256) *reinterpret_cast<u32x4*>(sZ + idx * 8) = zero4;
  }

  // ---- Q fragments (B operand of S^T): lane (query r, half h) holds dims 16 s + 8 h .. + 7 of K step s; dims 40 .. 47 (step 2, h = 1)
  // are the pad dimensions: element 0 carries -m' (set below), the rest 0
  v8 qf[QB][KS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int query = q_base + qb * 32 + r;
    query = query < N ? query : N - 1;
    const T* qp = hm_rows ? qkv + (((int64_t)bq * heads + hd) * N + query) * D : qkv + ((int64_t)bq * N + query) * C3 + hd * D;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int d0 = s * 16 + h * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (d0 < D) v = *reinterpret_cast<const u32x4*>(qp + d0);
      v8 q = *reinterpret_cast<v8*>(&v);
      if (q_scale != 1.0f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = (T)((float)q[j] * q_scale);
      }
      qf[qb][s] = q;
    }
  }

  // ---- K / V staging through registers: KV * NCH 16-byte chunks per tensor and tile, chunk c = tid + 256 i (D = 40: 320, two per thread of wave 0, one for the others).
  // Round 6: buffer loads with one descriptor per tensor -- per-lane byte offsets are computed once, the tile enters as the scalar offset, keys past N are out of the
  // descriptor's range and load as zeros (finite, masked in the scores), and which chunk groups a wave carries is wave-uniform: no per-lane bounds test, no 64-bit
  // address arithmetic, no zero fill and no divergent branch in the tile loop (that prologue was ~45 instructions and 8 branches in front of every tile's first MFMA).
  u32x4 rk[NLD], rv[NLD];
  const T* kbase = hm_rows ? qkv + ((int64_t)hm_rows * heads + (int64_t)bk * heads + hd) * N * D : qkv + (int64_t)bk * N * C3 + C + hd * D;
  const T* vbase = hm_rows ? qkv + ((int64_t)2 * hm_rows * heads + (int64_t)bv * heads + hd) * N * D : qkv + (int64_t)bv * N * C3 + 2 * C + hd * D;
  // range of both descriptors: every byte of a key < N lies inside, every byte of a key >= N outside (row-major: the V plane's start is the larger one, (hd + 1) D <= C)
  const unsigned nrec = hm_rows ? (unsigned)N * D * 2u : (unsigned)N * C3 * 2u - (unsigned)(2 * C + hd * D) * 2u;
  const unsigned tile_bytes = hm_rows ? (unsigned)KV * D * 2u : (unsigned)KV * C3 * 2u;
  const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(kbase), 0, nrec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(vbase), 0, nrec, 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wid);
  unsigned goff[NLD];
  int ldk[NLD], ldv[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int c = tid + 256 * i, row = c / NCH, ch = c - row * NCH;
    goff[i] = hm_rows ? (unsigned)c * 16u : (unsigned)row * C3 * 2u + (unsigned)ch * 16u;
    ldk[i] = row * KROW + ch * 8;
    ldv[i] = row * VROW + ch * 8;
  }
  auto load_kv = [&](int tile) __attribute__((always_inline)) {
    const unsigned soff = (unsigned)tile * tile_bytes;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (wv * 64 + 256 * i < KV * NCH) {        // wave-uniform
        rk[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, goff[i], soff, 0));
        rv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, goff[i], soff, 0));
      }
  };
  auto store_kv = [&](int bufi) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (wv * 64 + 256 * i < KV * NCH) {
        *reinterpret_cast<u32x4*>(sK + bufi * KBUF + ldk[i]) = rk[i];
        *reinterpret_cast<u32x4*>(sV + bufi * VBUF + ldv[i]) = rv[i];
      }
  };

  // ---- per-lane LDS read offsets (elements)
  const int kA = r * KROW + h * 8;                                    // K A-operand: key r of the 32-key block, dims 16 s + 8 h
  const int gi = lane & 15, vg = (lane >> 4) & 1, vq = gi >> 2, vp = gi & 3;
  const int vA = (4 * h + vq) * VROW + 16 * vg + 4 * vp;              // V^T A-operand through ds_read_b64_tr_b16: lane 4q+p of a 16-lane group
                                                                      // addresses row (key) q, columns (dims) 4p .. 4p+3 of a 4 x 16 block

  float mref[QB];
  f32x16 o[QB][DT];
  const int ntiles = (N + KV - 1) / KV;
  const int nfull = N / KV;

  // track = false (tiles after the first of the speculative pass, see below): no maximum, no decision -- m' stays what the first tile set.  A wave-uniform
  // run-time flag (one scalar branch per tile): a compile-time variant of the tile body per pass made six copies of it and hipcc spilled 65 registers
  bool track = true;
  auto tile_body = [&](int j, auto ragged_tag) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    const int kv0 = j * KV, cur = j & 1;
    if (j + 1 < ntiles) load_kv(j + 1);
    const T* tK = sK + cur * KBUF + kA;
    // V^T row tiles of 32 dims: lanes with vg = 1 read dims 32 dt + 16 .. + 31, which lie past the row for the last tile of D = 40
    // (rows 48 .. 63 of V^T are zero: read from the zero image)
    const T* tV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      tV[dt] = sV + cur * VBUF + vA + dt * 32;
      if (dt * 32 + 16 >= VROW && vg) tV[dt] = sZ + vA - 16;
    }

    v8 vf[4][DT];
    auto read_v = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const T* vp_ = tV[dt] + ks * 16 * VROW;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp_));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp_ + 8 * VROW));
          vf[ks][dt] = __builtin_bit_cast(v8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#if ATT_ABL & 8   // timing only: no V fragment reads
          vf[ks][dt] = qf[0][(ks + dt) % KS];
#endif
        }
    };

    // ---- S'^T = K Q^T - m' : s[qb][kb] register i = key kb*32 + (i&3) + 8(i>>2) + 4h, query r
    f32x16 s[QB][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        v8 kf = *reinterpret_cast<const v8*>(tK + kb * 32 * KROW + st * 16);
#if ATT_ABL & 16   // timing only: no K fragment reads
        kf = qf[QB - 1][(st + kb) % KS];
#endif
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          if (st == 0) {
            f32x16 z;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.f;
            s[qb][kb] = Frag32<T>::mfma(kf, qf[qb][st], z);
          } else {
            s[qb][kb] = Frag32<T>::mfma(kf, qf[qb][st], s[qb][kb]);
          }
        }
      }
    if constexpr (RAGGED) {   // only the last tile of a sequence that is not a multiple of 64 keys
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= N) s[qb][kb][i] = NEG_BIG;
    }

    // ---- reference maximum: moves only when a query exceeds it by 2^THR (or on the first tile)
    if (track || j == 0) {
    float mx[QB];
#if ATT_ABL & 1
    if (j == 0)
#endif
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float m = max32(s[qb]);
      // the other 32 keys of the query live in lane ^ 32: v_permlane32_swap exchanges lanes 32-63 of the first operand with lanes 0-31 of
      // the second (VALU, no LDS).  Inline asm: with this toolchain the builtin's SECOND result comes back as a copy of the first
      // (hipcc 7.2 folds max(sw[0], sw[1]) to sw[0]), which left a maximum over half of the keys -- still a valid softmax reference, but
      // no overflow guard: fp16 P overflowed on wide score ranges (tests/test_kernels_gpu.py::test_self_attention_d40_maximum_jumps_late)
      float ma = m, mb = m;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
      mx[qb] = fmaxf(ma, mb);   // both key halves of the query
    }
    const bool first = j == 0;
#if ATT_ABL & 1   // timing only: no running maximum after the first tile
    if (first) {
#else
    if (first || __builtin_amdgcn_ballot_w64(fmaxf(mx[0], mx[QB - 1]) > A40_THR) != 0) {
#endif
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float d = first ? mx[qb] : fmaxf(mx[qb], 0.f);
        const T mt = (T)(mref[qb] + d);                                // m' stays representable in the operand type
        const float mnew = (float)mt, de = mnew - mref[qb];
        mref[qb] = mnew;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) s[qb][kb][i] -= de;
        if (!first) {
          const float f = __builtin_amdgcn_exp2f(-de);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qb][dt][i] *= f;
        }
        if (h == GEO::MH) qf[qb][GEO::MS][0] = (T)(-mnew);
      }
    }

    }   // track

    // ---- V^T fragments of the tile (shared by the query blocks), requested BEFORE the exponentials: 16 transposed reads whose LDS latency
    // is then covered by ~300 cycles of v_exp instead of standing in front of every MFMA
    // (all 22 LDS reads of the tile requested in front of the S MFMAs instead: 3.493 vs 3.476 ms, no gain -- the partner wave already covers the LDS latency)
    read_v();
    __builtin_amdgcn_sched_barrier(0);   // keep the reads up here (the scheduler otherwise sinks each pair to just before its MFMA)
    // ---- P = exp2(S') packed straight into the PV operands, O^T += V^T P^T
    v8 pf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
#if ATT_ABL & 2   // timing only: no exponentials
        for (int e = 0; e < 8; ++e) pf[qb][ks][e] = (T)(s[qb][ks >> 1][(ks & 1) * 8 + e]);
#else
        for (int e = 0; e < 8; ++e) pf[qb][ks][e] = (T)__builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + e]);
#endif
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qb][dt] = Frag32<T>::mfma(vf[ks][dt], pf[qb][ks], o[qb][dt]);
    }

    if (j + 1 < ntiles) store_kv(cur ^ 1);   // buffer cur^1 was last read in iteration j-1, a barrier ago
#if ATT_ABL & 4   // timing only: no barrier (races)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    __syncthreads();
#endif
  };
  // Pass 0 is speculative: only tile 0 computes its maximum (m' = the exact maximum of the first 64 keys); the later tiles skip the 34 v_max3 + swap + ballot +
  // branch per wave and tile (the round-6 ablation priced them at 7.5 % of the kernel: every instruction of this loop costs its issue time) and exponentiate
  // against that m' whatever they hold.  That is exact as long as nothing overflows: P keeps its relative precision at any magnitude (bf16: up to 2^127; fp16: up
  // to 2^16 above m'), O and the denominators accumulate in fp32.  If a score exceeds the first tile's maximum by more than that, P (fp16) or the sums overflow,
  // the denominator row comes out non-finite, and the whole block repeats the pass with the running maximum of rounds 2-5 (pass 1): correct for every input,
  // twice the time on the blocks that need it (attention rows of SD1.x: none seen; the jump tests of tests/test_kernels_gpu.py take this path).  Where the
  // tracked pass would never have moved m', both passes are the same instructions on the same data.
  for (int pass = (ATT_SPEC_MAX && ntiles >= 8) ? 0 : 1; pass < 2; ++pass) {   // (a few tiles: the tracked pass at once -- N = 256, d = 160: 0.132 vs 0.140 ms)
    track = pass == 1;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      mref[qb] = 0.f;
      if (h == GEO::MH) qf[qb][GEO::MS][0] = (T)0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[qb][dt][i] = 0.f;
    }
    load_kv(0);
    store_kv(0);
    __syncthreads();
    for (int j = 0; j < nfull; ++j) tile_body(j, std::false_type{});
    if (nfull < ntiles) tile_body(nfull, std::true_type{});
    if (pass == 1) break;
    bool bad = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) bad |= !(fabsf(o[qb][GEO::LT][GEO::LI]) < 1.0e30f);   // (lanes that do not hold row D: another row's sum of the same P)
    if (!__syncthreads_or(bad)) break;
  }

  // ---- normalise and store: lane (query r, half h) holds dims (i&3) + 8(i>>2) + 4h (+32); the denominator is row 40 = tile 1, register 4, h = 0
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l = __shfl(o[qb][GEO::LT][GEO::LI], r + 32 * GEO::LH, 64);
    const float inv = 1.f / l;
    const int query = q_base + qb * 32 + r;
    if (query >= N) continue;
    T* op = out + ((int64_t)b * N + query) * C + hd * D + 4 * h;
#pragma unroll
    for (int g4 = 0; g4 < D / 8; ++g4) {
      const int dt = g4 >> 2, i0 = (g4 & 3) * 4;
      T v[4] = {(T)(o[qb][dt][i0] * inv), (T)(o[qb][dt][i0 + 1] * inv), (T)(o[qb][dt][i0 + 2] * inv), (T)(o[qb][dt][i0 + 3] * inv)};
      *reinterpret_cast<u32x2*>(op + 8 * g4) = *reinterpret_cast<u32x2*>(v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ self, head_dim 40: persistent one-wave-per-SIMD form
// Round 6 (profiles/r06_attention_persistent.log holds every number quoted here).  What the measurements said about self_attn40_kernel:
//   * a block costs ~10 us beside its tiles (a launch cut to ONE tile per block: 0.347 ms for 16384 blocks = 10.8 us per round of the chip): dispatch, the dependent
//     Q / first K-V loads, the first barrier, the epilogue -- 10 % of a 64-tile block;
//   * every 16 bytes per lane that LDS returns cost the SIMD ~20 matrix cycles, and with two waves of two query blocks each SIMD receives the tile's K / V^T
//     fragments twice: a two-wave build of THIS kernel (QB = 2, eight waves) runs 2220 cycles per tile with the exponentials removed, for 1792 cycles of MFMAs;
//   * a lone wave issues in order and pays every instruction's issue time itself (tools/experiments/valu_beside_mfma.hip: v_exp_f32 9.75 cycles, v_cvt_pk ~4, and exactly
//     two v_exp_f32 + one v_cvt_pk fit under one 32-cycle MFMA), but it receives the fragments ONCE for four query blocks and has 512 registers.
// So: 256 blocks (one per CU, 4 waves, QB = 4 query blocks of 32 per wave) walk the (row, head, 512-query block) items.  The tile loop is a software pipeline over units
// (key tile j, query block q) written in the order it must issue: each unit has 14 MFMAs -- the scores of the NEXT unit (6), then P V of the PREVIOUS unit (8) -- and behind
// each MFMA 2-3 v_exp_f32 of the CURRENT unit's scores plus the v_cvt_pk of the pair the previous interval finished; every operand was produced a whole unit earlier.
// K / V tiles: a ring of four LDS buffers = two stages of two tiles, ONE barrier per 128 keys; a tile is requested two tiles before it is stored (two register sets:
// nobody covers a lone wave's wait for memory) and stored a unit before the barrier that publishes it.  The stream runs on into the next item's first four tiles, the next
// item's Q fragments are requested under the last P V MFMAs, and an item change costs the reference-maximum pre-pass (24 MFMAs) + the output stores: ~12 k of ~184 k cycles.
// Reference maximum: m' = the exact maximum over the item's first 64 keys (pre-pass), then the speculative scheme of self_attn40_kernel: no maximum in the tile loop; a block
// whose denominators come out non-finite repeats the item with the running maximum (the tracked tiles at the end of the item loop), correct for every input.
// Measured (same box, N = 4096 x 128 rows / N = 9216 x 32 rows): 3.06 / 3.53 ms against 3.18-3.20 / 3.74 for self_attn40_kernel; the tile loop runs 2670 cycles per tile against an
// in-order issue sum of ~2400 (128 v_exp_f32 = 1250 of it).  QB = 2 with eight waves (NW = 8: same code, 256 registers, 13 spilled): 3.10 ms, not dispatched.
// head_dim 80 (the (L/2)^2 level; 24 MFMAs per unit, matrix-bound) runs the same code with QB = 2, NW = 4 and items of 256 queries: 0.443 against 0.491 ms at N = 1024 x 128 rows.
// Requirements (persistent_self_ok): N a multiple of the item's queries and of 256 (tiles: a multiple of 4, >= 16), the QKV tensor below 4 GB (one buffer descriptor, 32-bit
// scalar offsets), >= 2 items per CU.  ETAINV_A40_PERSIST=0 / ETAINV_A80_PERSIST=0: self_attn40_kernel as before.
template <typename T, int D, int QB, int NW>
__global__ void __launch_bounds__(64 * NW, 1) self_attn40q_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N, int heads, float q_scale, int mode, int n_img,
                                                              int nqb, int n_items, int first_row, int hm_rows, int xcd_remap, unsigned qkv_bytes) {
  typedef typename Frag<T>::v8 v8;
  typedef A32<D> GEO;
  constexpr int NT = 64 * NW;             // threads: four waves (one per SIMD) of QB 32-query blocks; eight waves = two per SIMD
  constexpr int QI = 32 * QB * NW;        // queries per item
  constexpr int KV = GEO::KV, KS = GEO::KS, KROW = GEO::KROW, VROW = GEO::VROW, KBUF = GEO::KBUF, VBUF = GEO::VBUF, DT = GEO::DT, NCH = GEO::NCH;
  constexpr int NLD = (KV * NCH + NT - 1) / NT;   // staging chunks per thread and tensor
  constexpr int NM = 2 * KS + 4 * DT;   // MFMAs per unit
  constexpr int NE = 32;                 // exponentials per unit and lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NBUF = 4;                  // two stages of two key tiles: one barrier per 128 keys
  T* sK = reinterpret_cast<T*>(smem);      // [NBUF][KV][KROW]
  T* sV = sK + NBUF * KBUF;                // [NBUF][KV][VROW]
  T* sZ = sV + NBUF * VBUF;                // [KV][VROW] zeros

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int C = heads * D, C3 = 3 * C;
  const int ntiles = N / KV;

  // ---- one-time LDS constants (they outlive the items): pad chunks (1, 0 x 7) of every K and V row of both buffers, the all-zero V image
  {
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    u32x4 one4 = zero4;
    {
      T one[2] = {(T)1.0f, (T)0.0f};
      one4[0] = *reinterpret_cast<unsigned*>(one);
    }
    for (int idx = tid; idx < NBUF * KV; idx += NT) {
      *reinterpret_cast<u32x4*>(sK + idx * KROW + D) = one4;
#pragma unroll
      for (int c = D + 8; c < KROW; c += 8) *reinterpret_cast<u32x4*>(sK + idx * KROW + c) = zero4;
      *reinterpret_cast<u32x4*>(sV + idx * VROW + D) = one4;
#pragma unroll
      for (int c = D + 8; c < VROW; c += 8) *reinterpret_cast<u32x4*>(sV + idx * VROW + c) = zero4;
    }
    if (GEO::ZBUF)
      for (int idx = tid; idx < VBUF / 8; idx += NT) *reinterpret_cast<u32x4*>(sZ + idx * 8) = zero4;
  }

  // ---- item-independent per-lane offsets.  One descriptor over the whole QKV tensor; (row, head, tile) enter as the scalar offset of each load
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(qkv), 0, qkv_bytes, 0x00020000);
  // K / V staging without a branch in the tile loop (a branch ends the scheduling region, and LLVM sinks a unit's exponentials across it to their first use): every
  // wave carries NLD chunk groups, the groups past the tile's KV * NCH chunks repeat the lane's first chunk (same bytes to the same LDS address)
  unsigned goff[NLD];
  int ldk[NLD], ldv[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int c = tid + NT * i;
    if (c >= KV * NCH) c = tid;
    const int row = c / NCH, ch = c - row * NCH;
    goff[i] = hm_rows ? (unsigned)c * 16u : (unsigned)row * C3 * 2u + (unsigned)ch * 16u;
    ldk[i] = row * KROW + ch * 8;
    ldv[i] = row * VROW + ch * 8;
  }
  const unsigned tile_bytes = hm_rows ? (unsigned)KV * D * 2u : (unsigned)KV * C3 * 2u;
  // Q fragments (B operand of S^T): lane (query r, half h) holds dims 16 s + 8 h .. + 7 of K step s; the chunk at dims D .. D + 7 (step MS, half MH) is the pad chunk
  // (element 0 carries -m', the rest 0): its lanes load whatever follows the head's row (inside the tensor) and are zeroed
  unsigned qoff[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) qoff[qb] = (unsigned)(wid * (32 * QB) + qb * 32 + r) * (hm_rows ? D : C3) * 2u + (unsigned)h * 16u;
  const int kA = r * KROW + h * 8;                                    // K A-operand: key r of the 32-key block, dims 16 s + 8 h
  const int gi = lane & 15, vg = (lane >> 4) & 1, vq = gi >> 2, vp = gi & 3;
  const int vA = (4 * h + vq) * VROW + 16 * vg + 4 * vp;              // V^T A-operand through ds_read_b64_tr_b16 (self_attn40_kernel)

  // item -> (query block, head, batch row) and the rows its Q / K / V come from (prompt-to-prompt / MasaCtrl couplings: self_attn40_kernel); byte offsets of the
  // item's Q block, K plane and V plane.  All wave-uniform
  auto decode = [&](int item, int& qblk, int& hd, int& b, unsigned& qo, unsigned& ko, unsigned& vo) __attribute__((always_inline)) {
    if (xcd_remap) {   // ids of one residue class modulo 8 (one XCD: the grid is a multiple of 8) take the query blocks of one (row, head) in turn
      const int xcd = item & 7, slot = item >> 3;
      const int set = (slot / nqb) * 8 + xcd;
      qblk = slot - (slot / nqb) * nqb;
      hd = set % heads;
      b = set / heads;
    } else {
      qblk = item % nqb;
      const int t = item / nqb;
      hd = t % heads;
      b = t / heads;
    }
    int bq = b, bk = b, bv = b;
    if (mode != 0) {
      int half, role, img;
      if (first_row < 0) {
        if (mode == 1 && b / n_img == 1) { bq = b + n_img; bk = b + n_img; }
      } else {
        row_roles(b + first_row, n_img, half, role, img);
        if (mode == 1 && half == 1 && role == 1) { bq = b - n_img; bk = b - n_img; }
        if (mode == 2 && role == 1) { bk = b - n_img; bv = b - n_img; }
      }
    }
    const int64_t q0 = (int64_t)qblk * QI;
    if (hm_rows) {
      qo = (unsigned)(((((int64_t)bq * heads + hd) * N + q0) * D) * 2);
      ko = (unsigned)(((((int64_t)hm_rows + bk) * heads + hd) * N * D) * 2);
      vo = (unsigned)(((((int64_t)2 * hm_rows + bv) * heads + hd) * N * D) * 2);
    } else {
      qo = (unsigned)((((int64_t)bq * N + q0) * C3 + hd * D) * 2);
      ko = (unsigned)(((int64_t)bk * N * C3 + C + hd * D) * 2);
      vo = (unsigned)(((int64_t)bv * N * C3 + 2 * C + hd * D) * 2);
    }
  };
  auto load_q = [&](unsigned qo, v8 (&dst)[QB][KS]) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, qoff[qb] + (unsigned)s * 32u, qo, 0));
        dst[qb][s] = *reinterpret_cast<v8*>(&v);
      }
  };

  // two register sets in flight: a tile is requested TWO tiles before it is stored to LDS (one wave per SIMD: nobody covers a wait for memory)
  u32x4 rk[2][NLD], rv[2][NLD];
  auto store_p = [&](int set, int bufi) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      *reinterpret_cast<u32x4*>(sK + bufi * KBUF + ldk[i]) = rk[set][i];
      *reinterpret_cast<u32x4*>(sV + bufi * VBUF + ldv[i]) = rv[set][i];
    }
  };
  v8 kf[2][KS], vf[4][DT];
  auto read_k = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < KS; ++st) kf[kb][st] = *reinterpret_cast<const v8*>(sK + buf * KBUF + kA + kb * 32 * KROW + st * 16);
  };
  auto read_vt = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      // lanes with vg = 1 read dims 32 dt + 16 .. + 31, which lie past the row for the last tile of D = 40 (rows 48 .. 63 of V^T are zero: the zero image)
      const T* tv = sV + buf * VBUF + vA + dt * 32;
      if (dt * 32 + 16 >= VROW && vg) tv = sZ + vA - 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const T* vp_ = tv + ks * 16 * VROW;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp_));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp_ + 8 * VROW));
        vf[ks][dt] = __builtin_bit_cast(v8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    }
  };
  f32x16 zero16;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
  v8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = (T)0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf[ks][dt] = zero8;   // (the first unit of an item multiplies them by P = 0: anything finite will do afterwards)

  v8 qf[QB][KS];
  float mref[QB];
  f32x16 o[QB][DT];
  bool warm = false;   // the previous item left this item's tiles 0, 1 in LDS buffers 0, 1, tiles 2, 3 in the two register sets and requested its Q fragments into qf

  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    int qblk, hd, b;
    unsigned qo, ko, vo;
    decode(item, qblk, hd, b, qo, ko, vo);
    const bool has_next = item + (int)gridDim.x < n_items;
    unsigned qo2 = qo, ko2 = ko + (unsigned)(ntiles - 4) * tile_bytes, vo2 = vo + (unsigned)(ntiles - 4) * tile_bytes;   // no next item: the stream re-reads this item's last tiles
    if (has_next) {
      int qblk2, hd2, b2;
      decode(item + gridDim.x, qblk2, hd2, b2, qo2, ko2, vo2);
    }
    auto load_p = [&](int set, int tile) __attribute__((always_inline)) {   // tiles ntiles .. ntiles + 3: the next item's first four
      const unsigned sk = tile < ntiles ? ko + (unsigned)tile * tile_bytes : ko2 + (unsigned)(tile - ntiles) * tile_bytes;
      const unsigned sv = tile < ntiles ? vo + (unsigned)tile * tile_bytes : vo2 + (unsigned)(tile - ntiles) * tile_bytes;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        rk[set][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[i], sk, 0));
        rv[set][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[i], sv, 0));
      }
    };
    if (!warm) {
      load_q(qo, qf);
      load_p(0, 0);
      load_p(1, 1);
      store_p(0, 0);
      store_p(1, 1);
      __syncthreads();
      load_p(0, 2);
      load_p(1, 3);
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (h == GEO::MH) qf[qb][GEO::MS] = zero8;   // the pad chunk
      if (q_scale != 1.0f) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) qf[qb][s][j] = (T)((float)qf[qb][s][j] * q_scale);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[qb][dt] = zero16;
    }

    // ---- m' of every query block = the exact maximum over the first 64 keys (24 MFMAs per item; the pipeline then recomputes tile 0 against it)
    read_k(0);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      f32x16 t[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int st = 0; st < KS; ++st) t[kb] = Frag32<T>::mfma(kf[kb][st], qf[qb][st], st == 0 ? zero16 : t[kb]);
      const float m = max32(t);
      float ma = m, mb = m;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));   // (inline asm: see self_attn40_kernel)
      const T mt = (T)fmaxf(ma, mb);                                                   // m' stays representable in the operand type
      mref[qb] = (float)mt;
      if (h == GEO::MH) qf[qb][GEO::MS][0] = (T)(-mref[qb]);
    }
    f32x16 sc[2], sn[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < KS; ++st) sc[kb] = Frag32<T>::mfma(kf[kb][st], qf[0][st], st == 0 ? zero16 : sc[kb]);
    // P of the previous unit (B operands of its P V MFMAs) and of the current one, as packed pairs: one v_cvt_pk per pair, written where the schedule below puts it (a
    // v8 assembled element by element is converted by four v_cvt_pk in a row at the point where its last element arrives: 24 issue cycles in one MFMA interval)
    u32x4 pfp[4], pfc[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) pfp[ks] = u32x4{0u, 0u, 0u, 0u};

    // one unit: MFMA m of the unit, then 2-3 exponentials of the current scores and the v_cvt_pk of the pairs the PREVIOUS interval finished (a lone wave issues in
    // order: 2 x 9.75 cycles of v_exp_f32 + one conversion + the MFMA's own issue fill the 32 cycles the matrix pipe needs); sched_barrier(0) pins the order written here
    auto unit = [&](auto qtag) __attribute__((always_inline)) {
      constexpr int q = decltype(qtag)::value, qnx = (q + 1) % QB, qpv = (q + QB - 1) % QB;
      constexpr bool PV_FIRST = q == QB - 1;   // the unit that needs the next tile's K fragments: their LDS latency under the eight P V MFMAs
      float ex[NE];
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int mm = PV_FIRST ? (m < 4 * DT ? m + 2 * KS : m - 4 * DT) : m;   // 0 .. 2 KS - 1: scores (the two 32-key blocks alternate), then P V (the O^T tiles alternate)
        if (mm < 2 * KS) {
          const int kb = mm & 1, st = mm >> 1;
          sn[kb] = Frag32<T>::mfma(kf[kb][st], qf[qnx][st], st == 0 ? zero16 : sn[kb]);
        } else {
          const int pv = mm - 2 * KS, ks = pv / DT, dt = pv - ks * DT;
          o[qpv][dt] = Frag32<T>::mfma(vf[ks][dt], __builtin_bit_cast(v8, pfp[ks]), o[qpv][dt]);
        }
        const int e0 = NE * m / NM, e1 = NE * (m + 1) / NM;                     // exponentials of this interval
        const int c0 = m == 0 ? 0 : (NE * (m - 1) / NM) / 2, c1 = m == NM - 1 ? NE / 2 : e0 / 2;   // pairs converted here: complete since the previous interval (last: all)
#pragma unroll
        for (int i = e0; i < e1; ++i) ex[i] = __builtin_amdgcn_exp2f(sc[i >> 4][i & 15]);
#pragma unroll
        for (int pr = c0; pr < c1; ++pr) {
          typedef T T2 __attribute__((ext_vector_type(2)));
          const T2 two = {(T)ex[2 * pr], (T)ex[2 * pr + 1]};
          pfc[pr >> 2][pr & 3] = __builtin_bit_cast(unsigned, two);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) sc[kb] = sn[kb];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) pfp[ks] = pfc[ks];
    };
    // two tiles per trip = one stage of the LDS ring (buffers b0, b0 + 1; the other stage b2, b2 + 1 receives tiles j + 2, j + 3, requested two tiles earlier)
    for (int j = 0; j < ntiles; j += 2) {
      const int b0 = j & 2, b2 = b0 ^ 2;
      unit(std::integral_constant<int, 0>{});          // scores (j, 1) | P V (j - 1, last) with the previous tile's V^T fragments | exp (j, 0)
      read_vt(b0);                                     // V^T of tile j: first used by the P V MFMAs of the next unit, behind its six score MFMAs
      if constexpr (QB == 4) unit(std::integral_constant<int, 1>{});
      store_p(0, b2);                                  // tile j + 2: the other stage's last readers passed the previous barrier; the stores complete under the units that follow
      load_p(0, j + 4);
      if constexpr (QB == 4) unit(std::integral_constant<int, 2>{});
      read_k(b0 + 1);                                  // K of tile j + 1 (same stage: visible since the previous barrier)
      unit(std::integral_constant<int, QB - 1>{});     // P V (j, last - 1) | scores (j + 1, 0) | exp (j, last)
      unit(std::integral_constant<int, 0>{});
      read_vt(b0 + 1);
      if constexpr (QB == 4) unit(std::integral_constant<int, 1>{});
      store_p(1, b2 + 1);                              // tile j + 3
      load_p(1, j + 5);
      if constexpr (QB == 4) unit(std::integral_constant<int, 2>{});
      __syncthreads();                                 // the one barrier per 128 keys: tiles j + 2, j + 3 visible; every wave is done reading this stage's V / the tile j + 1 K
      read_k(b2);                                      // K of tile j + 2 (last trip: the next item's tile 0 -- those scores are never used)
      unit(std::integral_constant<int, QB - 1>{});
    }
    if (has_next) load_q(qo2, qf);   // the item's Q fragments are done (the exact pass below reloads them): the next item's arrive under the last P V MFMAs and the output stores
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[QB - 1][dt] = Frag32<T>::mfma(vf[ks][dt], __builtin_bit_cast(v8, pfp[ks]), o[QB - 1][dt]);
    bool bad = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) bad |= !(fabsf(o[qb][GEO::LT][GEO::LI]) < 1.0e30f);   // (lanes that do not hold row D: another row's sum of the same P)
    warm = has_next;
    if (__syncthreads_or(bad)) {
      // ---- the exact pass: running maximum in every tile (the tile of self_attn40_kernel with four query blocks).  It restages the item from tile 0 and leaves
      // nothing of the next item behind
      warm = false;
      load_q(qo, qf);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        mref[qb] = 0.f;
        if (h == GEO::MH) qf[qb][GEO::MS] = zero8;
        if (q_scale != 1.0f) {
#pragma unroll
          for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qb][s][j] = (T)((float)qf[qb][s][j] * q_scale);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qb][dt] = zero16;
      }
      load_p(0, 0);
      store_p(0, 0);
      __syncthreads();
      for (int j = 0; j < ntiles; ++j) {
        const int cur = j & 1;
        if (j + 1 < ntiles) load_p(0, j + 1);
        read_k(cur);
        f32x16 s[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < KS; ++st) s[qb][kb] = Frag32<T>::mfma(kf[kb][st], qf[qb][st], st == 0 ? zero16 : s[qb][kb]);
        float mx[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float m = max32(s[qb]);
          float ma = m, mb = m;
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
          mx[qb] = fmaxf(ma, mb);
        }
        const bool first = j == 0;
        if (first || __builtin_amdgcn_ballot_w64(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[QB - 2], mx[QB - 1])) > A40_THR) != 0) {
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            const float d = first ? mx[qb] : fmaxf(mx[qb], 0.f);
            const T mt = (T)(mref[qb] + d);
            const float mnew = (float)mt, de = mnew - mref[qb];
            mref[qb] = mnew;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
              for (int i = 0; i < 16; ++i) s[qb][kb][i] -= de;
            if (!first) {
              const float f = __builtin_amdgcn_exp2f(-de);
#pragma unroll
              for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[qb][dt][i] *= f;
            }
            if (h == GEO::MH) qf[qb][GEO::MS][0] = (T)(-mnew);
          }
        }
        read_vt(cur);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          v8 pf[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[ks][e] = (T)__builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + e]);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[qb][dt] = Frag32<T>::mfma(vf[ks][dt], pf[ks], o[qb][dt]);
        }
        if (j + 1 < ntiles) store_p(0, cur ^ 1);
        __syncthreads();
      }
    }

    // ---- normalise and store: lane (query r, half h) holds dims (i&3) + 8(i>>2) + 4h (+32); the denominator is row D of O^T
    const int q_base = qblk * QI + wid * (32 * QB);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float l = __shfl(o[qb][GEO::LT][GEO::LI], r + 32 * GEO::LH, 64);
      const float inv = 1.f / l;
      const int query = q_base + qb * 32 + r;
      T* op = out + ((int64_t)b * N + query) * C + hd * D + 4 * h;
#pragma unroll
      for (int g4 = 0; g4 < D / 8; ++g4) {
        const int dt = g4 >> 2, i0 = (g4 & 3) * 4;
        T v[4] = {(T)(o[qb][dt][i0] * inv), (T)(o[qb][dt][i0 + 1] * inv), (T)(o[qb][dt][i0 + 2] * inv), (T)(o[qb][dt][i0 + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + 8 * g4) = *reinterpret_cast<u32x2*>(v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ cross
// EDIT = false: no source-key tile and no source-probability scratch in LDS (a third of the footprint: 6 instead of 2 resident blocks per
// CU for a kernel that waits on one dependent Q load per 16-query tile).  Only the cond-target rows of a prompt-to-prompt call need
// EDIT = true; the launcher splits such a call into two launches over contiguous row ranges (row0 = first batch row of the launch).
template <typename T, int D, int QT, bool EDIT>
__global__ void __launch_bounds__(256) cross_attn_kernel(const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ out,
                                                         CrossParams p, int row0) {
  typedef typename Frag<T>::v8 v8;
  typedef typename Frag<T>::v4 v4;
  constexpr int DP = (D + 31) / 32 * 32;
  constexpr int KS = DP / 32;
  constexpr int DT = (D + 15) / 16;
  constexpr int NCH = D / 8;
  constexpr int KC = 96;           // padded key count (6 tiles of 16; tile 5 is all padding)
  constexpr int KT = 5;            // key tiles that can hold real keys (80 >= 77)
  constexpr int KSTR = DP + 8;
  constexpr int VSTR = KC + 8;
  constexpr int PSTR = 81;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sK = reinterpret_cast<T*>(smem);            // [KC][KSTR]  keys of this row
  T* sKs = sK + KC * KSTR;                       // [KC][KSTR]  keys of the source row (EDIT only)
  T* sVt = EDIT ? sKs + KC * KSTR : sKs;         // [DT*16][VSTR]
  float* sP = reinterpret_cast<float*>(sVt + DT * 16 * VSTR);  // [4 waves][QT][16][PSTR] source probabilities (EDIT only)
  // (EDIT only) this image's per-token tables -- mapper (as int bits), refine alpha, equalizer, cross-replace alpha of the step -- read once per
  // block instead of four global loads per key and lane in every query tile
  float* sTab = sP + 4 * QT * 16 * PSTR;                       // [4][80]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, q4 = lane >> 4;
  // block -> (query-block lane bx of gx, head, batch row).  p.xcd_gx > 0: a 1-D grid whose ids are dealt so that the eight heads of one (row, query range) have the
  // same id modulo 8, i.e. run on ONE XCD (round-robin dispatch) at about the same time: a head's q / out rows are 80-byte (head_dim 40) pieces of 640-byte token rows,
  // and with the heads on eight XCDs every L2 fetched every line (PMC: 661 MB per launch at the fabric, twice the algorithmic bytes).  Speed only.
  int bx = blockIdx.x, gxn = gridDim.x, h = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_gx > 0) {
    const int lin = blockIdx.x, xcd = lin & 7, t = lin >> 3;
    h = t % p.heads;
    const int c = (t / p.heads) * 8 + xcd;
    gxn = p.xcd_gx;
    bx = c % gxn;
    bz = c / gxn;
  }
  const int b = row0 + bz;
  const int N = p.N, C = p.heads * D, C2 = 2 * C;

  int img = 0, role = -1, is_cond = 0;
  if (p.layout == 2) {
    int half;
    row_roles(b + p.first_row, p.n_img, half, role, img);
    is_cond = half;
  } else if (p.layout == 1) {
    img = b % p.n_img;
    is_cond = (p.rows == p.n_img) ? 1 : (b / p.n_img);
    role = 0;
  } else if (p.layout == 3) {   // rows [u_t, c_t, c_s] x n_img (store only: etainv_attn_ctrl.src_exit_block)
    const int g = b / p.n_img;
    img = b % p.n_img;
    is_cond = g >= 1;
    role = g == 1 ? 1 : 0;
  }
  const bool do_edit = EDIT && p.edit && p.layout == 2 && is_cond && role == 1;
  const bool do_store = p.map_layer >= 0 && is_cond;
  const int bs = b - p.n_img;  // source cond row of a target cond row

  // ---- stage K (and source K), V^T; zero all padding
  for (int idx = tid; idx < KC * (KSTR / 8); idx += 256) {
    const int key = idx / (KSTR / 8), ch = idx % (KSTR / 8);
    u32x4 a = {0u, 0u, 0u, 0u}, s = {0u, 0u, 0u, 0u};
    if (key < p.n_ctx && ch < NCH) {
      a = *reinterpret_cast<const u32x4*>(kv + ((int64_t)b * p.n_ctx + key) * C2 + h * D + ch * 8);
      if (do_edit) s = *reinterpret_cast<const u32x4*>(kv + ((int64_t)bs * p.n_ctx + key) * C2 + h * D + ch * 8);
    }
    *reinterpret_cast<u32x4*>(sK + key * KSTR + ch * 8) = a;
    if (EDIT) *reinterpret_cast<u32x4*>(sKs + key * KSTR + ch * 8) = s;
  }
  for (int idx = tid; idx < KC * NCH; idx += 256) {
    const int key = idx % KC, ch = idx / KC;
    u32x4 c = {0u, 0u, 0u, 0u};
    if (key < p.n_ctx) c = *reinterpret_cast<const u32x4*>(kv + ((int64_t)b * p.n_ctx + key) * C2 + C + h * D + ch * 8);
    const T* e = reinterpret_cast<const T*>(&c);
#pragma unroll
    for (int j = 0; j < 8; ++j) sVt[(ch * 8 + j) * VSTR + key] = e[j];
  }
  if (EDIT && do_edit && tid < p.n_ctx) {
    int mp = p.mapper ? p.mapper[img * 77 + tid] : 0;
    if (mp < 0) mp += p.n_ctx;                      // python negative index: -1 -> last token
    sTab[0 * 80 + tid] = __builtin_bit_cast(float, mp);
    sTab[1 * 80 + tid] = p.alphas ? p.alphas[img * 77 + tid] : 0.f;
    sTab[2 * 80 + tid] = p.equalizer ? p.equalizer[img * 77 + tid] : 1.f;
    sTab[3 * 80 + tid] = p.cross_alpha[img * 77 + tid];
  }
  __syncthreads();

  // a block stages K / V^T once and then walks several 64*QT-query blocks (grid.x < N / (64*QT) at large N)
  // The query fragments of the NEXT 64 QT-query block are requested before the current one is computed (D <= 80: the d = 160 instantiations have no
  // registers for it).  Each wave's chain -- load 16 queries, 10 MFMAs, softmax over 80 keys, 9 MFMAs, store -- otherwise starts with a global-load
  // latency: at 4096 tokens the kernel ran at 2 TB/s of its 0.67 GB with 16 waves per CU waiting on one load each.
  constexpr bool QPF = D <= 80;
  u32x4 qn[QPF ? QT : 1][QPF ? 2 : 1][QPF ? KS : 1];       // [query tile][0 = this row, 1 = the source row (EDIT)][k step]
  auto load_q = [&](int qb, int qt, int brow, u32x4 (&dst)[QPF ? KS : 1]) __attribute__((always_inline)) {
    int query = qb * (64 * QT) + wid * (16 * QT) + qt * 16 + fr;
    query = query < N ? query : N - 1;
    const T* qp = q + ((int64_t)brow * N + query) * C + h * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 32 + q4 * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (d0 < D) v = *reinterpret_cast<const u32x4*>(qp + d0);
      dst[ks] = v;
    }
  };
  auto load_block = [&](int qb) __attribute__((always_inline)) {
    if constexpr (QPF) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        load_q(qb, qt, b, qn[qt][0]);
        if (do_edit) load_q(qb, qt, bs, qn[qt][1]);
      }
    }
  };
  if (bx * (64 * QT) < N) load_block(bx);
  for (int qb = bx; qb * (64 * QT) < N; qb += gxn) {
  u32x4 qc[QPF ? QT : 1][QPF ? 2 : 1][QPF ? KS : 1];
  if constexpr (QPF) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qc[qt][e][ks] = qn[qt][e][ks];
    if ((qb + gxn) * (64 * QT) < N) load_block(qb + gxn);
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q_base = qb * (64 * QT) + wid * (16 * QT);
    int query = q_base + qt * 16 + fr;
    const bool q_ok = query < N;
    query = q_ok ? query : N - 1;
    float* sPw = sP + ((wid * QT + qt) * 16 + fr) * PSTR;

    // probabilities of one (row, K-set): returns p[kt][r] for keys kt*16 + q4*4 + r
    auto probs = [&](int brow, const T* keys, f32x4 (&pr)[KT]) {
      const T* qp = q + ((int64_t)brow * N + query) * C + h * D;
      f32x4 s[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 32 + q4 * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if constexpr (QPF) v = qc[qt][brow == b ? 0 : 1][ks];
        else if (d0 < D) v = *reinterpret_cast<const u32x4*>(qp + d0);
        v8 qf = *reinterpret_cast<v8*>(&v);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          v8 kf = *reinterpret_cast<const v8*>(keys + (kt * 16 + fr) * KSTR + ks * 32 + q4 * 8);
          s[kt] = Frag<T>::mfma(kf, qf, s[kt]);
        }
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = s[kt][r] * p.scale_log2;
          if (kt * 16 + q4 * 4 + r >= p.n_ctx) v = NEG_BIG;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
          s[kt][r] = e;
          rs += e;
        }
      rs += __shfl_xor(rs, 16, 64);
      rs += __shfl_xor(rs, 32, 64);
      const float inv = 1.f / rs;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) pr[kt] = s[kt] * inv;
    };

    f32x4 pr[KT];
    if (do_edit) {
      // source probabilities -> LDS (indexed by token), then the target's own probabilities, then the edit
      f32x4 ps[KT];
      probs(bs, sKs, ps);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sPw[kt * 16 + q4 * 4 + r] = ps[kt][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      probs(b, sK, pr);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + q4 * 4 + r;
          if (key < p.n_ctx) {
            const float tg = pr[kt][r];
            float rep;
            if (p.replace_mat) {                              // AttentionReplace: sum_w base[w] * M[w][key]
              const float* mrow = p.replace_mat + (int64_t)img * 77 * 77 + key;
              rep = 0.f;
              for (int w = 0; w < p.n_ctx; ++w) rep += sPw[w] * mrow[w * 77];
            } else {                                          // AttentionRefine
              const int mp = __builtin_bit_cast(int, sTab[0 * 80 + key]);
              const float a = sTab[1 * 80 + key];
              rep = sPw[mp] * a + tg * (1.f - a);
            }
            rep *= sTab[2 * 80 + key];
            const float ca = sTab[3 * 80 + key];
            pr[kt][r] = rep * ca + (1.f - ca) * tg;
          }
        }
    } else {
      probs(b, sK, pr);
    }

    if (do_store && q_ok) {
      float* mp = p.maps_acc + (((((int64_t)p.map_layer * p.n_img_cap + img) * 2 + role) * p.heads + h) * N + query) * 77;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + q4 * 4 + r;
          if (key < p.n_ctx) mp[key] += pr[kt][r];
        }
    }

    // ---- O^T = V^T P^T over 3 k-steps of 32 keys (keys 80..95 are padding with p = 0)
    v8 pf[3];
#pragma unroll
    for (int kt = 0; kt < 6; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pf[kt >> 1][(kt & 1) * 4 + r] = (kt < KT) ? (T)pr[kt < KT ? kt : 0][r] : (T)0.f;
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const T* vp = sVt + (dt * 16 + fr) * VSTR + ks * 32 + q4 * 4;
        v4 lo = *reinterpret_cast<const v4*>(vp);
        v4 hi = *reinterpret_cast<const v4*>(vp + 16);
        v8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[dt] = Frag<T>::mfma(vf, pf[ks], acc[dt]);
      }
    if (q_ok) {
      T* op = out + ((int64_t)b * N + query) * C + h * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int dc = dt * 16 + q4 * 4;
        if (dc < D) {
          T o[4] = {(T)acc[dt][0], (T)acc[dt][1], (T)acc[dt][2], (T)acc[dt][3]};
          *reinterpret_cast<u32x2*>(op + dc) = *reinterpret_cast<u32x2*>(o);
        }
      }
    }
  }
  }
}

template <typename T, int D>
static int launch_self_t(const void* qkv, void* out, int b, int n, int heads, int mode, int n_img, hipStream_t s, int q_prescaled = 0, int first_row = 0) {
  constexpr int QT = SELF_QT(D);
  constexpr int DP = (D + 31) / 32 * 32, DT = (D + 15) / 16;
  const size_t lds = (size_t)2 * (64 * (DP + 8) + DT * 16 * (64 + 8)) * sizeof(T);
  static bool attr[kMaxDevices] = {};   // per device
  const int dev = current_device();
  if (!attr[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attn_kernel<T, D, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr[dev] = true;
  }
  const float scale_log2 = q_prescaled ? 1.0f : (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  ProfScope prof(PROF_SELF_ATTN, 4.0 * (double)b * heads * (double)n * (double)n * D, s);
  static const int stagger = getenv("ETAINV_ATT_STAGGER") ? atoi(getenv("ETAINV_ATT_STAGGER")) : 0;
  hipLaunchKernelGGL((self_attn_kernel<T, D, QT>), dim3(cdiv(n, 64 * QT), heads, b), dim3(256), lds, s, (const T*)qkv, (T*)out, n,
                     heads, scale_log2, mode, n_img, stagger, first_row);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

bool self_attn40_v2_enabled() {
  static const bool on = !env_on("ETAINV_ATT_OLD");
  return on;
}

template <typename T, int D, int QB, int OCC>
static int launch_self40(const void* qkv, void* out, int b, int n, int heads, int mode, int n_img, int q_prescaled, hipStream_t s, int first_row = 0, int head_major = 0) {
  const int hm_rows = head_major ? b : 0;
  const int nqb = cdiv(n, 128 * QB);
  const bool remap = ((b * heads) % 8) == 0;
  const float q_scale = q_prescaled ? 1.0f : (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  constexpr size_t A40_LDS = A32<D>::LDS;
  ProfScope prof(PROF_SELF_ATTN, 4.0 * (double)b * heads * (double)n * (double)n * D, s);
  static const int stagger = getenv("ETAINV_A40_STAGGER") ? atoi(getenv("ETAINV_A40_STAGGER")) : 0;
  static const size_t lds_pad = getenv("ETAINV_A40_LDSPAD") ? (size_t)atoi(getenv("ETAINV_A40_LDSPAD")) : 0;   // experiment: fewer resident blocks
  const size_t lds = A40_LDS + lds_pad;
  static bool attr_set[kMaxDevices] = {};   // per device and instantiation (T, D, QB, OCC); the LDSPAD experiment changes the size per process only
  const int dev = current_device();
  if ((lds_pad || lds > 64 * 1024) && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attn40_kernel<T, D, true, QB, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attn40_kernel<T, D, false, QB, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  if (remap)
    hipLaunchKernelGGL((self_attn40_kernel<T, D, true, QB, OCC>), dim3(nqb * heads * b), dim3(256), lds, s, (const T*)qkv, (T*)out, n, heads, q_scale, mode, n_img, nqb, stagger, first_row, hm_rows);
  else
    hipLaunchKernelGGL((self_attn40_kernel<T, D, false, QB, OCC>), dim3(nqb, heads, b), dim3(256), lds, s, (const T*)qkv, (T*)out, n, heads, q_scale, mode, n_img, nqb, stagger, first_row, hm_rows);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

// the persistent one-wave-per-SIMD kernel (self_attn40q_kernel): one block per CU walks the (row, head, query block of 32 QB NW queries) items
template <typename T, int D, int QB, int NW>
static int launch_self40q(const void* qkv, void* out, int b, int n, int heads, int mode, int n_img, int q_prescaled, hipStream_t s, int first_row, int head_major, int n_cu) {
  const int hm_rows = head_major ? b : 0;
  const int nqb = n / (32 * QB * NW);
  const int n_items = nqb * heads * b;
  const int remap = ((b * heads) % 8) == 0;
  const float q_scale = q_prescaled ? 1.0f : (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  constexpr size_t lds = (size_t)(4 * (A32<D>::KBUF + A32<D>::VBUF) + A32<D>::VBUF) * 2;   // four tile buffers + the zero image (90 / 112 KB at head_dim 40 / 80; one block per CU)
  static bool attr_set[kMaxDevices] = {};
  const int dev = current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attn40q_kernel<T, D, QB, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  ProfScope prof(PROF_SELF_ATTN, 4.0 * (double)b * heads * (double)n * (double)n * D, s);
  const unsigned qkv_bytes = (unsigned)((int64_t)3 * b * n * heads * D * 2);
  const int grid = n_items < n_cu ? n_items / 8 * 8 : n_cu / 8 * 8;
  hipLaunchKernelGGL((self_attn40q_kernel<T, D, QB, NW>), dim3(grid), dim3(64 * NW), lds, s, (const T*)qkv, (T*)out, n, heads, q_scale, mode, n_img, nqb, n_items, first_row, hm_rows, remap,
                     qkv_bytes);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

static int device_cu_count() {
  static int n_cu[kMaxDevices] = {};
  const int dev = current_device();
  if (n_cu[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 8;
    n_cu[dev] = v;
  }
  return n_cu[dev];
}

// what self_attn40q_kernel asks of a launch: whole items, a tile count the four-buffer ring divides, a tensor one buffer descriptor spans, two items per CU
static bool persistent_self_ok(int b, int n, int heads, int d, int item_queries) {
  return n % item_queries == 0 && n % 256 == 0 && n >= 1024 && (int64_t)3 * b * n * heads * d * 2 < ((int64_t)1 << 32) && (int64_t)(n / item_queries) * heads * b >= 2 * device_cu_count();
}

bool self_attn_head_major_ok(int d, int dtype) {   // which launches read the head-major QKV planes (the 32x32x16 kernel of head_dim 40 / 80)
  static const bool v2_80 = !env_on("ETAINV_ATT80_OLD");
  return dtype != ETAINV_F32 && self_attn40_v2_enabled() && (d == 40 || (d == 80 && v2_80));
}

int launch_self_attention_mode(const void* qkv, void* out, int b, int n, int heads, int d, int mode, int n_img, int dtype,
                               hipStream_t s, int q_prescaled, int first_row, int head_major) {
  ETAINV_CHECK(!head_major || self_attn_head_major_ok(d, dtype), "head-major QKV planes: head_dim 40 / 80 on the 16-bit kernel only");
  ETAINV_CHECK(qkv && out && b > 0 && n > 0, "bad arguments");
  ETAINV_CHECK(mode == 0 || (n_img > 0 && ((first_row >= 0 && b + first_row == 4 * n_img && (first_row == 0 || (first_row == n_img && mode == 1))) ||
                                           (first_row < 0 && b == 3 * n_img && mode == 1))),
               "ptp / masactrl modes need the 4*n_img backward layout (ptp: optionally without its first n_img rows, or rows [u_t, c_t, c_s] with first_row < 0)");
  if (dtype == ETAINV_F32) {
    ETAINV_CHECK(!q_prescaled, "fp32 path: the softmax scale is applied in the kernel");
    return launch_self_attention_f32(qkv, out, b, n, heads, d, mode, n_img, s, first_row);
  }
  ETAINV_CHECK(!q_prescaled || d == 40 || d == 80, "pre-scaled queries: head_dim 40 / 80 only");
  ETAINV_CHECK(!q_prescaled || self_attn40_v2_enabled(), "pre-scaled queries need the d = 40 kernel");
  if (d == 40 && self_attn40_v2_enabled() && (int64_t)cdiv(n, 256) * heads * b <= 256 && n > 128) {
    // few blocks (single-image calls: N = 4096, 8 heads, 1 row = 128 blocks of 256 queries on 256 CUs): one 32-query block per wave, twice the blocks
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_self40<T, 40, 1, 2>(qkv, out, b, n, heads, mode, n_img, q_prescaled, s, first_row, head_major)));
  }
  // enough (row, head, query block) items for two per CU: the persistent one-wave-per-SIMD kernel (items of 512 queries at head_dim 40, 256 at head_dim 80)
  if (d == 40 && self_attn40_v2_enabled() && persistent_self_ok(b, n, heads, d, 512) && env_flag("ETAINV_A40_PERSIST", true)) {
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_self40q<T, 40, 4, 4>(qkv, out, b, n, heads, mode, n_img, q_prescaled, s, first_row, head_major, device_cu_count())));
  }
  if (d == 40 && self_attn40_v2_enabled()) {
    // two 32-query blocks per wave, 2 waves per SIMD (one block per wave with 3 / 4 waves per SIMD: +10 % / +52 % time, re-measured in round 6 on the lean staging:
    // profiles/r06_attention_experiments.log)
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_self40<T, 40, 2, 2>(qkv, out, b, n, heads, mode, n_img, q_prescaled, s, first_row, head_major)));
  }
  if (d == 80 && self_attn40_v2_enabled() && persistent_self_ok(b, n, heads, d, 256) && env_flag("ETAINV_A80_PERSIST", true)) {
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_self40q<T, 80, 2, 4>(qkv, out, b, n, heads, mode, n_img, q_prescaled, s, first_row, head_major, device_cu_count())));
  }
  static const bool v2_80 = !env_on("ETAINV_ATT80_OLD");   // A/B: head_dim 80 on the 32x32x16 kernel (one 32-query block per wave)
  if (d == 80 && self_attn40_v2_enabled() && v2_80) {
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_self40<T, 80, 1, 2>(qkv, out, b, n, heads, mode, n_img, q_prescaled, s, first_row, head_major)));
  }
  static const bool v2_160 = env_flag("ETAINV_ATT160_V2", true);   // A/B: head_dim 160 (the (L/4)^2 level) on the 32x32x16 kernel too: one 32-query block per wave,
  if (d == 160 && self_attn40_v2_enabled() && v2_160 && !head_major) {   // one block per CU (104 KB of K / V tiles, ~300 registers)
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_self40<T, 160, 1, 1>(qkv, out, b, n, heads, mode, n_img, q_prescaled, s, first_row, head_major)));
  }
  // (the generic kernel takes pre-scaled queries with scale 1: ETAINV_ATT80_OLD sends head_dim 80 here while the engine still folds the scale into to_q)
  ETAINV_DISPATCH_HALF(dtype, T, switch (d) {
    case 40: return launch_self_t<T, 40>(qkv, out, b, n, heads, mode, n_img, s, q_prescaled, first_row);
    case 80: return launch_self_t<T, 80>(qkv, out, b, n, heads, mode, n_img, s, q_prescaled, first_row);
    case 160: return launch_self_t<T, 160>(qkv, out, b, n, heads, mode, n_img, s, q_prescaled, first_row);
    default: ETAINV_FAIL("head_dim must be 40, 80 or 160");
  });
  return 0;
}

template <typename T, int D>
static int launch_cross_t(const void* q, const void* kv, void* out, int b, const CrossParams& p, hipStream_t s) {
  constexpr int QT = 2;
  constexpr int DP = (D + 31) / 32 * 32, DT = (D + 15) / 16;
  const size_t lds_edit = (size_t)(2 * 96 * (DP + 8) + DT * 16 * (96 + 8)) * sizeof(T) + (size_t)(4 * QT * 16 * 81 + 4 * 80) * sizeof(float);
  const size_t lds_plain = (size_t)(96 * (DP + 8) + DT * 16 * (96 + 8)) * sizeof(T);
  static bool attr[kMaxDevices] = {};   // per device
  const int dev = current_device();
  if (!attr[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_kernel<T, D, QT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_edit);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_kernel<T, D, QT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_plain);
    attr[dev] = true;
  }
  ProfScope prof(PROF_CROSS_ATTN, 4.0 * (double)b * p.heads * (double)p.N * (double)p.n_ctx * D, s);
  const int nqb = cdiv(p.N, 64 * QT);
  auto launch = [&](bool edit, int row0, int rows) {
    // blocks per (row, head): a block stages its head's 77 keys and values once (49 KB at head_dim 160, with 2-byte transposing LDS stores) and then walks its share of the
    // query blocks.  Few tokens (N <= 1024): as few blocks as fill the chip -- at N = 256 a second block per (row, head) halves the work that amortises a staging and the
    // launch was 0.105 ms, 0.078 with one.  Many tokens (N = 4096): the staging is small beside 32 query blocks, more blocks balance better (0.231 -> 0.224 ms).
    // (round 6, profiles/r06_cross_attention_xcd_heads_ab.log; ETAINV_CROSS_BLOCKS overrides the target)
    static const int target_env = getenv("ETAINV_CROSS_BLOCKS") ? std::max(64, atoi(getenv("ETAINV_CROSS_BLOCKS"))) : 0;
    const int target = target_env ? target_env : p.N >= 2048 ? (edit ? 2048 : 6144) : 1024;   // (the edit launch stages two key sets and its tables: fewer, longer blocks at every N)
    // ... and a cap by token count, from a sweep of 1 .. 32 blocks per (row, head) at 32 / 64 / 96 / 128 rows, 300 launches each (profiles/r06_cross_attention_blocks_sweep.log): N = 256
    // (head_dim 160) wants ONE block at every row count (0.021 / 0.040 / 0.059 ms against 0.028 / 0.053 / 0.079 with two), N = 1024 at most two, N = 4096 at most eight (32 rows: 0.052 against
    // 0.064 ms with 24; the other row counts unchanged)
    const int cap = target_env ? nqb : p.N <= 256 ? 1 : p.N <= 1024 ? 2 : p.N <= 4096 ? 8 : nqb;
    const int gx = std::max(1, std::min(std::min(nqb, cap), cdiv(target, rows * p.heads)));
    const bool xcd_heads = env_flag("ETAINV_CROSS_XCD", true);   // (read per launch: tests/test_kernels_gpu.py compares the two placements in one process)
    CrossParams pl = p;
    dim3 grid(gx, p.heads, rows);
    if (xcd_heads && (gx * rows) % 8 == 0) {   // the heads of a (row, query range) on one XCD (see the kernel)
      pl.xcd_gx = gx;
      grid = dim3(gx * p.heads * rows);
    }
    if (edit)
      hipLaunchKernelGGL((cross_attn_kernel<T, D, QT, true>), grid, dim3(256), lds_edit, s, (const T*)q, (const T*)kv, (T*)out, pl, row0);
    else
      hipLaunchKernelGGL((cross_attn_kernel<T, D, QT, false>), grid, dim3(256), lds_plain, s, (const T*)q, (const T*)kv, (T*)out, pl, row0);
  };
  if (p.edit && p.layout == 2) {          // rows [u_s, u_t, c_s, c_t] x n_img: only the last quarter (cond target) is edited
    const int edit0 = 3 * p.n_img - p.first_row;   // first cond-target row of this call
    launch(false, 0, edit0);
    launch(true, edit0, b - edit0);
  } else {
    launch(false, 0, b);
  }
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_cross_attention_p(const void* q, const void* kv, void* out, int b, int d, const CrossParams& p, int dtype, hipStream_t s) {
  if (dtype == ETAINV_F32) return launch_cross_attention_f32(q, kv, out, b, d, p, s);
  ETAINV_CHECK(q && kv && out && b > 0, "bad arguments");
  ETAINV_CHECK(p.n_ctx >= 1 && p.n_ctx <= 77, "n_ctx must be in [1,77]");
  ETAINV_DISPATCH_HALF(dtype, T, switch (d) {
    case 40: return launch_cross_t<T, 40>(q, kv, out, b, p, s);
    case 80: return launch_cross_t<T, 80>(q, kv, out, b, p, s);
    case 160: return launch_cross_t<T, 160>(q, kv, out, b, p, s);
    default: ETAINV_FAIL("head_dim must be 40, 80 or 160");
  });
  return 0;
}

}  // namespace etainv
