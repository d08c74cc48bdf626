// MFMA implicit-GEMM for gfx950: one kernel family serves every contraction of the SD1.x UNet
//   conv3x3 (stride 1 / stride 2 / fused nearest-2x upsample, zero pad 1)  : K = 9*Cin
//   conv1x1 / Linear (QKV, out-proj, FF, shortcuts, proj_in/out, time MLP) : K = Cin
// Activations are NHWC (= row-major [B*H*W][C]); weights are [Cout][taps][Cin] (K-contiguous), so both MFMA
// operands are 8 consecutive K elements per lane (one ds_read_b128).  The channel concat of the decoder
// (torch.cat([x, skip], 1)) is never materialised: the K loop walks two source tensors.
//
// Tile: BM x BN x 64, 256 threads = 2x2 waves, v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulate.
// The weight tile is fed as the MFMA A operand and the activation tile as B, so an accumulator holds
// 4 consecutive OUTPUT CHANNELS of one pixel per lane -> 8-byte NHWC stores, lane-local GEGLU pairing.
// LDS image: [rows][8 x 16B chunks], physical chunk = chunk ^ (row & 7): conflict-free ds_read_b128
// (16 lanes of a ds_read_b128 group hit 16 distinct (128-B half, 16-B slot) pairs).  Staging is direct-to-LDS
// (global_load_lds_dwordx4, no VGPR round trip, no ds_write): the LDS image of one wave-instruction is lane-linear
// (8 rows x 128 B), so the XOR swizzle is applied to the per-lane SOURCE chunk; zero padding of the 3x3 halo reads
// a 16-byte zero page.  Double-buffered LDS: the DMA of tile k+1 is in flight during the MFMAs of tile k, one
// vmcnt(0) + barrier per K tile.
// Epilogue (fused): + bias[n] + rowvec[batch][n] (time-embedding projection) + residual[m][n], or GEGLU
// a * gelu_erf(g) with (a, g) columns interleaved per 32-column group at weight-pack time.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace etainv {

template <typename T> struct Mfma;
template <> struct Mfma<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

constexpr int BK = 64;  // K elements per LDS tile (8 chunks of 16 B per row)
constexpr int64_t SPLITK_WS_BYTES = 64ll << 20;


// ---- LayerNorm statistics carried between two GEMMs (IGemmParams::stat_out / ln_stat): partial (mean, M2) pairs of equal counts combined by
// Chan's pairwise update -- no E[x^2] - E[x]^2 cancellation, fixed combination order
// both sides hold `n` values each
__device__ __forceinline__ void chan_merge_equal(float n, float& mean, float& m2, float mb, float m2b) {
  const float d = mb - mean;
  mean += 0.5f * d;
  m2 += m2b + d * d * (0.5f * n);
}
// a <- the value of the even 16-lane row of each row pair, b <- the odd one (in both rows of the pair); inline asm: hipcc folds the second
// result of the builtin into a copy of the first when both inputs are the same value
__device__ __forceinline__ void pair_rows16(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// a <- the value of lanes 0-31, b <- the value of lanes 32-63 (in both halves)
__device__ __forceinline__ void pair_halves32(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// combine the partials held by the four lanes fq = 0..3 (lane >> 4) that share a pixel row, `n` values each; every lane ends with the same pair
__device__ __forceinline__ void chan_merge_fq_equal(float n, float& mean, float& m2) {
  float ma, mb, qa, qb;
  pair_rows16(mean, ma, mb); pair_rows16(m2, qa, qb);
  chan_merge_equal(n, ma, qa, mb, qb);
  pair_halves32(ma, mean, mb); pair_halves32(qa, m2, qb);
  chan_merge_equal(2.f * n, mean, m2, mb, qb);
}

// sum over the 16 lanes of a DPP row (lane & 15): butterflies xor 1, xor 2 (quad_perm), i <-> 7 - i (row_half_mirror), i <-> 15 - i (row_mirror);
// every lane ends with the total
__device__ __forceinline__ float row_sum16(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));
  return x;
}

// LN: the folded-LayerNorm role of the launch as a compile-time constant -- the ring kernels sit exactly at 256 VGPRs, and a role read at
// run time makes the register allocator keep all three epilogues' values apart (0 = none, 1 = producer: row statistics from the epilogue,
// 2 = consumer: rstd (acc - mean s) + c, 3 = GroupNorm producer: per-channel (sum, sum of squares) of the stored values over the rows of each wave tile,
// 4 = consumer that stores the head-major QKV planes of IGemmParams::hm_*)
// PATCH (256 x 160 ring, conv3x3 stride 1 on H, W multiples of 16 only): an M tile is a 16 x 16 PIXEL PATCH of one image and the K loop runs channel
// chunk major, the nine taps inside: the activations of a chunk are brought to LDS ONCE as the halo'd 18 x 18 patch (41 DMA pieces of 1 KiB instead of
// 9 x 32) and the nine taps read it at shifted rows; the halo outside the image comes from the zero page.  Measured motivation: profiles/
// r04_conv_traffic_ablation.log (activation pieces for one tap in nine: conv3x3 -9 ... -15 %).
template <typename T, int BM, int BN, int WAVES_M, int STAGES, int UPS = 0, int LN = 0, bool PATCH = false>   // UPS: 1 = nine taps on the upsampled grid, 2 = phase form
__global__ void __launch_bounds__(WAVES_M * 128, 2) igemm_kernel(IGemmParams p) {
  static_assert(!PATCH || (STAGES == 3 && BM == 256 && WAVES_M == 4 && !UPS && (LN == 0 || LN == 3)), "patch mode: the 256-row ring, plain or GroupNorm-producer epilogue");
  constexpr int PW = 18;                                   // patch pitch (16 + halo)
  constexpr int PROWS = 328;                               // 18 x 18 = 324 patch rows, rounded up to whole 8-row DMA pieces
  constexpr int PPIECES = PROWS / 8;                       // 41
  constexpr int A_REGION = PATCH ? 2 * PROWS * 64 : STAGES * BM * 64;   // elements: two patch slots / the A ring
  constexpr bool ln_emit = LN == 1;
  constexpr bool ln_use = LN == 2 || LN == 4;
  constexpr bool hm_out = LN == 4;   // LayerNorm consumer that writes the head-major QKV planes (its own instantiation: the plain consumer sits at 256 VGPRs)
  constexpr bool gn_emit = LN == 3;
  constexpr int NTHR = WAVES_M * 128;      // WAVES_M x 2 waves
  constexpr int RP = NTHR / 8;             // LDS rows staged per pass (8 lanes x 16 B per 128-B row)
  constexpr int WM = BM / WAVES_M, WN = BN / 2;  // wave tile
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int A_LOADS = BM / RP;
  constexpr int B_LOADS = (BN + RP - 1) / RP;
  static_assert(BM % RP == 0, "A tile rows must be a multiple of the rows per pass");
  typedef typename Mfma<T>::frag frag;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sA = reinterpret_cast<T*>(smem);                       // [STAGES][BM*BK]   (PATCH: [2][PROWS*BK])
  T* sB = reinterpret_cast<T*>(smem) + A_REGION;            // [STAGES][BN*BK]
  // bias of the tiles in flight: filled by LDS-DMA together with a tile's first K step, read by its epilogue (a global bias
  // load in the epilogue waits behind every queued DMA: ~1.5 us per tile with the matrix pipe idle)
  float* sBias = reinterpret_cast<float*>(smem + ((size_t)A_REGION + (size_t)STAGES * BN * BK) * sizeof(T));   // [4][BN]
  // LayerNorm consumer on the 256 x 128 ring (the GEGLU projection; 13 KB of LDS to spare): the s vector and the (mean, rstd) rows of the tiles in
  // flight arrive by DMA with the bias -- read from global memory at the start of the epilogue they cost one exposed memory latency per tile
  // (measured +1.5 us on a 7.4 us tile), and the 256 x 160 ring has neither the LDS nor the registers to fetch them early
  constexpr bool LN_STAGED = LN == 2 && STAGES == 3 && BN == 128;
  float* sLnS = sBias + 4 * BN + 256;        // [4][BN]   (after the bias ring and the 1-KiB dummy piece)
  float* sLnStat = sLnS + 4 * BN;            // [4][BM][2]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;

  // PERSISTENT blocks: block b owns output tiles b, b + G, b + 2G, ... and runs ONE flattened (tile, k-tile) pipeline, so
  // the DMA of the next tile's first K tile is in flight during the epilogue of the current one (short-K GEMMs --
  // K = 320 is only 5 K tiles -- otherwise pay a full memory latency + an unoverlapped epilogue per tile).
  // XCD-aware order: virtual ids v and v + 8 share an XCD (round-robin dispatch, G % 8 == 0), so each XCD walks a
  // contiguous run of tiles, n fastest: neighbours reuse the same activation panel from that XCD's L2.
  // SPLIT-K (two-slot kernels only, small M*N with a deep K: the 8x8 / 16x16 levels at batch 1): every output tile becomes
  // `ksplit` virtual tiles that each walk nk/ksplit K tiles and store an fp32 partial; a second kernel adds the partials in
  // a fixed order and applies the epilogue (deterministic, no float atomics).
  const int ksplit = (STAGES != 3 && p.ksplit > 1) ? p.ksplit : 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int total_tiles = ((p.M + BM - 1) / BM) * tiles_n * ksplit;
  const int G = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
  auto tile_origin = [&](int i, int& m0, int& n0) -> int {   // returns the K part of virtual tile i
    int v = blockIdx.x + i * G;
    if (STAGES == 3 && p.xcd_gn > 1) {   // (ring kernels only: the two-slot kernels have no registers to spare for it)
      // 2-D XCD grid (launch_igemm_t checked the divisibilities): XCD x = v & 7 owns M panels [xm * mr, (xm + 1) * mr) x N tiles [xn * nr, (xn + 1) * nr)
      // and walks them n fastest: its slice of the weight matrix (N / gn rows) is what its L2 keeps or re-streams, not the whole matrix
      const int gn = p.xcd_gn, x = v & 7, j = v >> 3;
      const int nr = tiles_n / gn, mr = (total_tiles / tiles_n) / (8 / gn);
      const int ml = j / nr, nl = j - ml * nr;
      m0 = ((x / gn) * mr + ml) * BM;
      n0 = ((x % gn) * nr + nl) * BN;
      return 0;
    }
    if ((total_tiles & 7) == 0) v = (v & 7) * (total_tiles >> 3) + (v >> 3);
    const int part = ksplit > 1 ? v % ksplit : 0;
    if (ksplit > 1) v /= ksplit;
    m0 = (v / tiles_n) * BM;
    n0 = (v - (v / tiles_n) * tiles_n) * BN;
    return part;
  };

  const int cin = p.c1 + p.c2;
  const int kc = cin / BK;            // K tiles per tap
  const int nk = p.taps * kc / ksplit;   // K tiles per (virtual) tile
  const int pad = (p.taps == 9 && !p.pad0) ? 1 : 0;
  const int HWo = p.Ho * p.Wo;
  const int Hin = p.ups ? p.H * 2 : p.H, Win = p.ups ? p.W * 2 : p.W;

  // per-thread staging geometry: row = (tid >> 3) + 32 * i; the lane's LDS slot is chunk (tid & 7) of that row (lane-linear
  // DMA image), which must hold LOGICAL chunk (tid & 7) ^ (row & 7); row & 7 == (tid >> 3) & 7 for every i.
  const int lchunk = (tid & 7) ^ ((tid >> 3) & 7);
  int a_b[A_LOADS], a_y[A_LOADS], a_x[A_LOADS];
  // fast path (no upsample): element offset of tap (0,0) in source 1 / source 2 (lane chunk included) and 9-bit tap validity
  int a_e1[A_LOADS], a_e2[A_LOADS], a_mask[A_LOADS];
  const T* w_row[B_LOADS];
  int it_n0 = 0, it_m0 = 0, it_bias_buf = 0, it_boff = 0;
  // Rarely used fields are read through the kernel-argument segment at their point of use (scalar loads) instead of living in SGPRs across the main
  // loop: the ring kernels are out of scalar registers too, and every SGPR spilled to a VGPR lane costs a vector register.
  auto ln_args = [&]() __attribute__((always_inline)) {
    typedef const __attribute__((address_space(4))) IGemmParams* KArgs;   // constant address space: scalar loads
    KArgs kp = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
  };
  // position of the K tile being issued, advanced incrementally (no integer division in the loop); it_k0 = first K tile of the part
  int it_tap = 0, it_c0 = 0, it_ky = 0, it_kx = 0, it_k0 = 0;
  auto setup_issue = [&](int i) __attribute__((always_inline)) {   // geometry of the tile whose K tiles are being prefetched
    int m0, n0;
    const int part = tile_origin(i, m0, n0);
    it_n0 = n0;
    it_m0 = m0;
    it_bias_buf = i & 3;
    if (ksplit > 1) {
      it_k0 = part * nk;
      it_tap = it_k0 / kc;
      it_c0 = (it_k0 - it_tap * kc) * BK;
      it_ky = it_tap / 3;
      it_kx = it_tap - it_ky * 3;
    }
    if constexpr (PATCH) {
      // (the activation side of a patch tile has no per-row state: issue_a derives the halo'd patch of a chunk from the tile index)
    } else if (p.taps == 1) {
      // 1x1 / Linear: the source row IS the output row -- no (image, y, x) decomposition, no halo mask (a K = 320 tile is
      // only five K steps long, so the ~500 VALU instructions of the general setup were ~20 % of its main loop)
#pragma unroll
      for (int q = 0; q < A_LOADS; ++q) {
        int m = m0 + (tid >> 3) + RP * q;
        m = m < p.M ? m : p.M - 1;
        a_b[q] = a_y[q] = a_x[q] = 0;
        a_e1[q] = m * p.c1 + lchunk * 8;
        a_e2[q] = m * p.c2 + lchunk * 8;
        a_mask[q] = 1;
      }
    } else if (UPS == 2) {
      // conv3x3 behind a nearest-2x upsample as four 2x2 phase convs on the source image (launch_pack_ups4, IGemmParams::ups == 2).  Virtual rows
      // are image-major, then phase (2 py + px), then source pixel: a tile lies inside one phase block (launch_igemm: H * W % BM == 0)
      // (ups_pm: phase-major -- [phase][image][pixel], M / 4 rows per phase: a tile lies inside one phase, its rows may belong to several images)
      const int HWs = p.H * p.W;
      const int PHs = p.M >> 2;
      const int blk = m0 / HWs;
      const int ph = p.ups_pm ? m0 / PHs : blk & 3;
#pragma unroll
      for (int q = 0; q < A_LOADS; ++q) {
        int b = blk >> 2, r = m0 - blk * HWs + (tid >> 3) + RP * q;
        if (p.ups_pm) {
          const int rr = m0 - ph * PHs + (tid >> 3) + RP * q;
          b = rr / HWs;
          r = rr - b * HWs;
        }
        const int ys = r / p.W, xs = r - ys * p.W;
        a_b[q] = b;
        a_y[q] = ys + (ph >> 1) - 1;             // source pixel of tap (0, 0)
        a_x[q] = xs + (ph & 1) - 1;
        const int pix0 = (b * p.H + a_y[q]) * p.W + a_x[q];
        a_e1[q] = pix0 * p.c1 + lchunk * 8;
        a_e2[q] = 0;
        int mask = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int iy = a_y[q] + (t >> 1), ix = a_x[q] + (t & 1);
          mask |= ((iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W)) << t;
        }
        a_mask[q] = mask;
      }
    } else {
#pragma unroll
      for (int q = 0; q < A_LOADS; ++q) {
        int m = m0 + (tid >> 3) + RP * q;
        m = m < p.M ? m : p.M - 1;
        int b = m / HWo, r = m - b * HWo;
        int oy = r / p.Wo, ox = r - oy * p.Wo;
        a_b[q] = b;
        a_y[q] = oy * p.stride - pad;
        a_x[q] = ox * p.stride - pad;
        const int pix0 = (b * p.H + a_y[q]) * p.W + a_x[q];
        a_e1[q] = pix0 * p.c1 + lchunk * 8;
        a_e2[q] = pix0 * p.c2 + lchunk * 8;
        int mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = a_y[q] + t / 3, ix = a_x[q] + t % 3;
          mask |= ((iy >= 0) & (iy < Hin) & (ix >= 0) & (ix < Win)) << t;   // (Hin, Win: the fused-upsample grid when p.ups)
        }
        a_mask[q] = mask;   // (with pad0 the a_y / a_x origin is the output pixel itself)
      }
    }
    // per-image weights / bias (a GroupNorm folded into the 1x1 conv behind it: engine.cpp, transformer()): the tile lies inside one image
    int64_t woff = 0;
    it_boff = 0;
    {
      const auto kp = ln_args();
      const int64_t wbs = kp->w_batch_stride;
      if (wbs) {
        const int img = m0 / p.rows_per_batch;
        woff = img * wbs;
        it_boff = img * kp->bias_batch_stride;
      }
      if constexpr (UPS == 2) woff = (int64_t)(p.ups_pm ? m0 / (p.M >> 2) : (m0 / (p.H * p.W)) & 3) * p.N * (4 * cin);   // the phase's 2x2 kernel
    }
#pragma unroll
    for (int q = 0; q < B_LOADS; ++q) {
      int n = n0 + (tid >> 3) + RP * q;
      n = n < p.N ? n : p.N - 1;
      w_row[q] = reinterpret_cast<const T*>(p.w) + woff + (int64_t)n * (p.taps * cin) + lchunk * 8;
    }
  };
  const T* zero_page = reinterpret_cast<const T*>(p.zeros);
  const int wrow0 = __builtin_amdgcn_readfirstlane(wid) * 8;   // first LDS row of this wave's 1-KiB DMA piece

  auto issue_tile = [&](int kt, int buf) __attribute__((always_inline)) {
    const int tap = it_tap, c0 = it_c0, ky = it_ky, kx = it_kx;
    const bool second = c0 >= p.c1;
    const T* src = reinterpret_cast<const T*>(second ? p.a2 : p.a1);
    const int cs = second ? p.c2 : p.c1;
    const int coff = (second ? c0 - p.c1 : c0) + lchunk * 8;
    T* dA = sA + buf * BM * BK;
    T* dB = sB + buf * BN * BK;
    if (!p.ups) {
      // per K tile and load: one add of a wave-uniform element offset, one shift-add onto the (scalar) base, a validity select
      const int uoff = (ky * p.W + kx) * cs + (second ? c0 - p.c1 : c0);
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const bool ok = (a_mask[i] >> tap) & 1;
        const unsigned elem = (unsigned)((second ? a_e2[i] : a_e1[i]) + uoff);   // garbage for halo lanes, never dereferenced
        const T* g = ok ? src + elem : zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dA + (wrow0 + RP * i) * BK), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        int iy = a_y[i] + ky, ix = a_x[i] + kx;
        const bool ok = (iy >= 0) & (iy < Hin) & (ix >= 0) & (ix < Win);
        iy >>= 1;
        ix >>= 1;
        const T* g = ok ? src + ((int64_t)(a_b[i] * p.H + iy) * p.W + ix) * cs + coff : zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dA + (wrow0 + RP * i) * BK), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      if (BN % RP == 0 || wrow0 + RP * i < BN)   // wave-uniform: a wave stages 8 whole rows
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_row[i] + (int64_t)(it_k0 + kt) * BK),
                                         (__attribute__((address_space(3))) void*)(dB + (wrow0 + RP * i) * BK), 16, 0, 0);
    if (kt == 0 && p.bias && wrow0 * 8 < BN) {   // waves 0 .. BN/64: 64 floats each (wrow0 = 8 * wave)
      const int c = wrow0 * 8 + lane;
      if (c < BN) {
        const int n = it_n0 + c < p.N ? it_n0 + c : p.N - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + it_boff + n),
                                         (__attribute__((address_space(3))) void*)(sBias + it_bias_buf * BN + wrow0 * 8), 4, 0, 0);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  T* out = reinterpret_cast<T*>(p.out);
  const T* res = reinterpret_cast<const T*>(p.residual);
  // ---- epilogue: lane holds out[m][n .. n+3], m = pixel (MFMA column), n = channel (MFMA row)
  // Store NB 16-channel blocks of one pixel row group.  A lane holds 4 channels (8 B) of every block, so a plain store
  // writes sixteen 32-byte segments per instruction, and the CU's store path retires ~one segment per 4 clocks whatever its
  // size (measured, L2-resident: 8.6 B/clk/CU for 32-B segments, 16.8 for 64-B, 24.5 for 128-B) -- the epilogue of a
  // K = 320 GEMM was ~40 % of its time.  v_permlane16_swap exchanges (block 2k, lanes 16-31 / 48-63) with (block 2k+1,
  // lanes 0-15 / 32-47): every lane then owns 8 consecutive channels and an instruction writes 64-byte segments.
  // hm_d / hm_skip (head-major QKV output only): columns >= hm_d of the wave's span belong to the NEXT head, whose plane starts hm_skip elements
  // further on (a store never straddles: heads are multiples of 8 channels)
  auto store_row_group = [&](T* prow, auto& po, auto nb_tag, bool row_ok, bool wide, int hm_d = 1 << 30, int hm_skip = 0) __attribute__((always_inline)) {
    constexpr int NB = decltype(nb_tag)::value;
    if (p.debug & 64) row_ok = row_ok && po[0][0] == 0x12345678u && po[NB - 1][1] == 0x9abcdef0u;   // ablation: compute, (almost) never store
    auto at = [&](int col) __attribute__((always_inline)) { return prow + col + (col >= hm_d ? hm_skip : 0); };
    if (wide) {
#pragma unroll
      for (int k = 0; k + 1 < NB; k += 2) {
        const auto lo = __builtin_amdgcn_permlane16_swap(po[k][0], po[k + 1][0], false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(po[k][1], po[k + 1][1], false, false);
        const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
        if (row_ok) *reinterpret_cast<u32x4*>(at((k + (fq & 1)) * 16 + (fq >> 1) * 8)) = v;
      }
      if (NB & 1)
        if (row_ok) *reinterpret_cast<u32x2*>(at((NB - 1) * 16 + fq * 4)) = po[NB - 1];
    } else {
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if (row_ok) *reinterpret_cast<u32x2*>(at(k * 16 + fq * 4)) = po[k];
    }
  };
  // returns the store class of the tile: 0 = unknown number of store instructions (partial tile / slow path),
  // 1 = exactly MT*ceil(NT/2) per wave, 2 = exactly MT*ceil(NT/4) per wave (GEGLU) -- see the counted vmcnt waits of the ring
  // LayerNorm consumer: mean / rstd of this lane's MT pixel rows from the producer's partials.  The four lanes that share a row (fq) each
  // combine every fourth partial, then merge among themselves.
  auto ln_rows = [&](int mw, float (&mean)[MT], float (&rstd)[MT]) __attribute__((always_inline)) {
    const float* ln_stat = ln_args()->ln_stat;   // finalized (mean, rstd) per row: norm.hip, ln_finalize_kernel / row_stats_kernel
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int m = mw + i * 16 + fr;
      m = m < p.M ? m : p.M - 1;
      const f32x2 v = *reinterpret_cast<const f32x2*>(ln_stat + (int64_t)m * 2);
      mean[i] = v[0];
      rstd[i] = v[1];
    }
  };
  // LayerNorm producer: (mean, M2) of the NT * 4 stored (rounded) values this lane holds of pixel row m, merged over the four fq lanes = the WN
  // columns of the wave tile -> partial `pidx` of row m
  auto emit_row_stat = [&](const u32x2 (&po)[NT], int m, int pidx) __attribute__((always_inline)) {
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
    chan_merge_fq_equal((float)(NT * 4), mu, m2);
    const auto kp = ln_args();
    const int stat_P = kp->stat_P;
    if (fq == 0 && m < p.M && pidx < stat_P) *reinterpret_cast<f32x2*>(kp->stat_out + ((int64_t)m * stat_P + pidx) * 2) = (f32x2){mu, m2};
  };
  int ep_part = 0;   // K part of the tile in the epilogue (split-K)
  auto epilogue = [&](int m0, int n0, int tile) __attribute__((always_inline)) -> int {
    if (p.debug & 2) {   // ablation: no epilogue traffic
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (acc[i][j][0] == 12345.f) out[0] = from_f32<T>(1.f);
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      return 0;
    }
    // Fast path (every hot layer): ALL epilogue inputs (bias + time-embedding row, residual) are requested before the first
    // store.  Loads and stores share vmcnt on gfx9 and may complete out of order with respect to each other, so a load issued
    // after a store makes hipcc wait vmcnt(0) = the L2 ack of that store: interleaved load/store groups serialise 20 store
    // round trips per tile (measured: ~10 us per 256 x 160 tile, 45 % of a K = 320 GEMM).
    if (!p.out_nchw && !p.out_f32 && ksplit == 1 && (p.rows_per_batch % WM) == 0) {
      int mw = m0 + wm * WM;
      const int batch = (mw < p.M ? mw : p.M - 1) / p.rows_per_batch;   // one image per wave tile
      if (p.debug & 8) mw &= 255;   // ablation: all tiles store into the same cache-resident rows
      const bool wide = (n0 + BN <= p.N) && (p.N % 16) == 0 && !(p.debug & 32);
      const float* tile_bias = sBias + (tile & 3) * BN;   // whole tile inside N: 16-byte stores after a lane swap
      // memory row of this lane's 16-row group i.  PATCH: the group is patch row wm * 4 + i of a 16 x 16 pixel patch, lane fr its column (mw stays the
      // VIRTUAL row index, patches enumerated image-major: `batch` and the GroupNorm row-block index derived from it are unchanged)
      int pm0 = 0;
      if constexpr (PATCH) {
        const int mt = m0 / BM, tpr = p.W / 16, tpi = (p.H / 16) * tpr;
        const int b_ = mt / tpi, r_ = mt - b_ * tpi, ty_ = r_ / tpr;
        pm0 = (b_ * p.H + ty_ * 16 + wm * 4) * p.W + (r_ - ty_ * tpr) * 16;
      }
      int um0 = 0, uph = 0;
      if constexpr (UPS == 2) {   // virtual row -> output pixel (2 ys + py, 2 xs + px) of image b
        const int HWs = p.H * p.W, blk = m0 / HWs;
        uph = p.ups_pm ? m0 / (p.M >> 2) : blk & 3;
        um0 = (blk >> 2) * (4 * HWs);          // first output row of the image (image-major order: the tile lies inside one image)
      }
      auto row_m = [&](int i) __attribute__((always_inline)) {
        if constexpr (UPS == 2) {
          const int HWs = p.H * p.W;
          int r = (mw + i * 16 + fr) % HWs, im0 = um0;
          if (p.ups_pm) {
            const int rr = mw + i * 16 + fr - uph * (p.M >> 2), b = rr / HWs;
            r = rr - b * HWs;
            im0 = b * (4 * HWs);
          }
          const int ys = r / p.W, xs = r - ys * p.W;
          return im0 + (2 * ys + (uph >> 1)) * (2 * p.W) + 2 * xs + (uph & 1);
        }
        return PATCH ? pm0 + i * p.W + fr : mw + i * 16 + fr;
      };
      const bool ln = ln_use;   // folded LayerNorm: v = rstd[m] * (acc - mean[m] * s[n]) + c[n]  (c arrives as the bias)
      float ln_mean[MT], ln_rstd[MT];
      if (!p.geglu) {
        f32x4 bv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + wn * WN + j * 16 + fq * 4;
          bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (p.bias && !ln) bv[j] = *reinterpret_cast<const f32x4*>(tile_bias + wn * WN + j * 16 + fq * 4);
          if (p.rowvec && n < p.N) bv[j] += *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)batch * p.rowvec_stride + n);
        }
        if (ln) ln_rows(mw, ln_mean, ln_rstd);
#ifndef ETAINV_RES_PREFETCH
#define ETAINV_RES_PREFETCH 1
#endif
        if (ln) {
          // LayerNorm consumer (never has a residual): its own block, so that the s vector and the row statistics do not extend the register
          // live ranges of the residual path below (256 VGPRs, no spill)
          f32x4 sv[NT];
          const float* ln_s = ln_args()->ln_s;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * WN + j * 16 + fq * 4;
            sv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (n < p.N) sv[j] = *reinterpret_cast<const f32x4*>(ln_s + n);
          }
          ln_rows(mw, ln_mean, ln_rstd);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int m = mw + i * 16 + fr;
            u32x2 po[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              // (c re-read from LDS per row group: 20 registers less across the block than a copy held next to s)
              const f32x4 v = (acc[i][j] - ln_mean[i] * sv[j]) * ln_rstd[i] + *reinterpret_cast<const f32x4*>(tile_bias + wn * WN + j * 16 + fq * 4);
              acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
              T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
              po[j] = *reinterpret_cast<u32x2*>(o);
            }
            if constexpr (hm_out) {   // head-major QKV planes (launch_igemm checked: whole tile inside one batch row, the 80-column span = whole heads)
              // (division-free: the kernel has no registers to spare.  8 heads; the span starts at a multiple of 80 columns; batch row by a magic multiply)
              const int g = ((n0 + wn * WN) / 80) * (p.hm_dim == 40 ? 2 : 1);   // first head of the span, counted over q | k | v
              const int part = g >> 3, head0 = g & 7;
              const int b = (int)(((unsigned long long)(unsigned)m0 * (unsigned)p.hm_magic) >> 38);
              T* plane = out + ((int64_t)part * p.M + (int64_t)b * p.hm_tokens) * (8 * p.hm_dim) + (int64_t)head0 * p.hm_tokens * p.hm_dim;
              store_row_group(plane + (int64_t)(m - b * p.hm_tokens) * p.hm_dim, po, std::integral_constant<int, NT>{}, m < p.M, wide, p.hm_dim,
                              (p.hm_tokens - 1) * p.hm_dim);
            } else {
              store_row_group(out + (int64_t)m * p.N + n0 + wn * WN, po, std::integral_constant<int, NT>{}, m < p.M, wide);
            }
          }
        } else {
#if ETAINV_RES_PREFETCH
        // residual: ALL of the tile's residual loads go out before the first store (MT * NT 8-byte loads per lane; the fragment registers
        // of the finished K step are dead here).  Issued per 16-row group right before that group's stores, every group exposed a full
        // memory latency with nothing else in flight: in-kernel stamps put the epilogue of a 320 -> 320 + residual GEMM at 22.6k cycles
        // per tile for 80 KB in + 80 KB out, twice its main loop.
        u32x2 rv[MT][NT];
        if (res) {
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int m = row_m(i);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const int n = n0 + wn * WN + j * 16 + fq * 4;
              rv[i][j] = (u32x2){0u, 0u};
              if (m < p.M && n < p.N) rv[i][j] = *reinterpret_cast<const u32x2*>(res + (int64_t)m * p.N + n);
            }
          }
        }
#endif
        // (a LayerNorm producer keeps the packed rows until every store is out: they take over the residual's registers row by row, and the
        // statistics run when the bias registers are dead)
        u32x2 po[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int m = row_m(i);
#if !ETAINV_RES_PREFETCH
          u32x2 rvi[NT];
          if (res) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const int n = n0 + wn * WN + j * 16 + fq * 4;
              rvi[j] = (u32x2){0u, 0u};
              if (m < p.M && n < p.N) rvi[j] = *reinterpret_cast<const u32x2*>(res + (int64_t)m * p.N + n);
            }
          }
#endif
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            f32x4 v = acc[i][j] + bv[j];
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (res) {
              T r[4];
#if ETAINV_RES_PREFETCH
              *reinterpret_cast<u32x2*>(r) = rv[i][j];
#else
              *reinterpret_cast<u32x2*>(r) = rvi[j];
#endif
              v[0] += to_f32(r[0]); v[1] += to_f32(r[1]); v[2] += to_f32(r[2]); v[3] += to_f32(r[3]);
            }
            T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
            po[i][j] = *reinterpret_cast<u32x2*>(o);
          }
          store_row_group(out + (int64_t)m * p.N + n0 + wn * WN, po[i], std::integral_constant<int, NT>{}, m < p.M, wide);
        }
        if (ln_emit) {
#pragma unroll
          for (int i = 0; i < MT; ++i) emit_row_stat(po[i], mw + i * 16 + fr, n0 / WN + wn);
        }
        if (gn_emit) {
          // GroupNorm producer: the GroupNorm that reads this output needs sum and sum of squares per (image, group).  Per wave tile (WM rows of one
          // image x WN channels) and channel: this lane's MT rows, then the 16 pixel lanes of the DPP row -> stat_out[row tile][0 / 1][channel]
          // (fixed order: deterministic); norm.hip gn_finalize_kernel adds row tiles and channels of a group.  Replaces the statistics pass over x.
          float* gs = ln_args()->stat_out;
          int64_t rt = mw / WM;
          if constexpr (UPS == 2) {
            if (p.ups_pm) {   // the partials stay image-major -- [image][phase][64-row block] -- for norm.hip's finalize (H * W % 64 == 0: a wave tile lies inside one image)
              const int HWs = p.H * p.W, ph = mw / (p.M >> 2), rr = mw - ph * (p.M >> 2), b = rr / HWs;
              rt = ((int64_t)b * 4 + ph) * (HWs / WM) + (rr - b * HWs) / WM;
            }
          }
          if (m0 + BM > p.M) {   // (wave-uniform) ragged last M tile: rows past M count as zero
#pragma unroll
            for (int i = 0; i < MT; ++i)
              if (mw + i * 16 + fr >= p.M)
#pragma unroll
                for (int j = 0; j < NT; ++j) po[i][j] = (u32x2){0u, 0u};
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            f32x4 sm = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              T o[4];
              *reinterpret_cast<u32x2*>(o) = po[i][j];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float v = to_f32(o[q]);
                sm[q] += v;
                sq[q] += v * v;
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              sm[q] = row_sum16(sm[q]);
              sq[q] = row_sum16(sq[q]);
            }
            const int n = n0 + wn * WN + j * 16 + fq * 4;
            if (fr == 0 && n < p.N && mw < p.M) {   // (mw < M is wave-uniform: a wave tile entirely past M has no row block in stat_out)
              *reinterpret_cast<f32x4*>(gs + (rt * 2 + 0) * p.N + n) = sm;
              *reinterpret_cast<f32x4*>(gs + (rt * 2 + 1) * p.N + n) = sq;
            }
          }
        }
        }
      } else {
        const int No = p.N >> 1;
        f32x4 ba[NT / 2 + 1], bg[NT / 2 + 1], sa[NT / 2 + 1], sg[NT / 2 + 1];
#pragma unroll
        for (int j = 0; j < NT / 2; ++j) {
          const int n = n0 + wn * WN + j * 16 + fq * 4;
          ba[j] = bg[j] = sa[j] = sg[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (ln) {
            if constexpr (LN_STAGED) {
              const float* ts = sLnS + (tile & 3) * BN + wn * WN + j * 16 + fq * 4;
              sa[j] = *reinterpret_cast<const f32x4*>(ts);
              sg[j] = *reinterpret_cast<const f32x4*>(ts + WN / 2);
            } else {
              const float* ln_s = ln_args()->ln_s;
              sa[j] = *reinterpret_cast<const f32x4*>(ln_s + n);
              sg[j] = *reinterpret_cast<const f32x4*>(ln_s + n + WN / 2);
            }
          }
          if (p.bias) {
            ba[j] = *reinterpret_cast<const f32x4*>(tile_bias + wn * WN + j * 16 + fq * 4);
            bg[j] = *reinterpret_cast<const f32x4*>(tile_bias + wn * WN + j * 16 + fq * 4 + WN / 2);
          }
        }
        if (ln) {
          if constexpr (LN_STAGED) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const f32x2 v = *reinterpret_cast<const f32x2*>(sLnStat + (tile & 3) * (2 * BM) + (wm * WM + i * 16 + fr) * 2);
              ln_mean[i] = v[0];
              ln_rstd[i] = v[1];
            }
          } else {
            ln_rows(mw, ln_mean, ln_rstd);
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int m = mw + i * 16 + fr;
          u32x2 po[NT / 2 + 1];
#pragma unroll
          for (int j = 0; j < NT / 2; ++j) {
            f32x4 a = acc[i][j], g = acc[i][j + NT / 2];
            if (ln) {
              a = (a - ln_mean[i] * sa[j]) * ln_rstd[i];
              g = (g - ln_mean[i] * sg[j]) * ln_rstd[i];
            }
            a += ba[j];
            g += bg[j];
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[i][j + NT / 2] = (f32x4){0.f, 0.f, 0.f, 0.f};
            T o[4];
            if (p.debug & 128) {   // A/B: scalar A&S 7.1.26 erf (exp2 + rcp per value)
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = from_f32<T>(a[q] * gelu_erf_f(g[q]));
            } else {
              const gelu_f32x2 g01 = gelu_pair((gelu_f32x2){g[0], g[1]}), g23 = gelu_pair((gelu_f32x2){g[2], g[3]});
              o[0] = from_f32<T>(a[0] * g01[0]); o[1] = from_f32<T>(a[1] * g01[1]);
              o[2] = from_f32<T>(a[2] * g23[0]); o[3] = from_f32<T>(a[3] * g23[1]);
            }
            po[j] = *reinterpret_cast<u32x2*>(o);
          }
          store_row_group(out + (int64_t)m * No + ((n0 + wn * WN) >> 1), po, std::integral_constant<int, NT / 2>{}, m < p.M, wide);
        }
      }
      const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N) && !(p.debug & 16);   // every lane of every wave stored
      return full ? (p.geglu ? 2 : ln_emit ? 3 : gn_emit ? 4 : 1) : 0;   // (full implies wide)
    }
    // general path (ragged tiles, conv_out, fp32 outputs, split-K partials): LayerNorm consumers supported, statistics are never emitted here
    // (the ring kernels take a LayerNorm consumer only when every tile runs the fast path above: launch_igemm)
    const bool ln_slow = ln_use && STAGES != 3 && ksplit == 1;
    float lns_mean[MT], lns_rstd[MT];
    if (ln_slow) ln_rows(m0 + wm * WM, lns_mean, lns_rstd);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * WM + i * 16 + fr;
      const bool m_ok = m < p.M;
      const int batch = m / p.rows_per_batch;
      if (!p.geglu) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + wn * WN + j * 16 + fq * 4;
          f32x4 v = acc[i][j];
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (!m_ok || n >= p.N) continue;
          if (ln_slow) v = (v - lns_mean[i] * *reinterpret_cast<const f32x4*>(ln_args()->ln_s + n)) * lns_rstd[i];
          if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + (int64_t)batch * ln_args()->bias_batch_stride + n);
          if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)batch * p.rowvec_stride + n);
          if (res) {
            const T* r = res + (int64_t)m * p.N + n;
            v[0] += to_f32(r[0]); v[1] += to_f32(r[1]); v[2] += to_f32(r[2]); v[3] += to_f32(r[3]);
          }
          if (ksplit > 1) {   // fp32 partial of K part `ep_part` (bias / row vector / residual are applied by the reduction)
            *reinterpret_cast<f32x4*>(p.ws + ((int64_t)ep_part * p.M + m) * p.N + n) = v;
          } else if (p.out_nchw) {   // conv_out: this lane holds the 4 output channels of pixel m
            const int64_t base = ((int64_t)batch * p.out_nchw) * p.rows_per_batch + (m - batch * p.rows_per_batch);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (c >= p.out_nchw) break;
              const int64_t o = base + (int64_t)c * p.rows_per_batch;
              if (p.out_io_dtype == ETAINV_F32) reinterpret_cast<float*>(p.out)[o] = v[c];
              else if (p.out_io_dtype == ETAINV_F16) reinterpret_cast<f16*>(p.out)[o] = (f16)v[c];
              else reinterpret_cast<bf16*>(p.out)[o] = (bf16)v[c];
            }
          } else if (p.out_f32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.N + n) = v;
          } else {
            T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
            *reinterpret_cast<u32x2*>(out + (int64_t)m * p.N + n) = *reinterpret_cast<u32x2*>(o);
          }
        }
      } else {
        // GEGLU: wave columns [0, WN/2) hold a, [WN/2, WN) hold the matching gate g (weight rows interleaved per
        // WN-column group by pack mode 2); output width N/2.
        const int No = p.N >> 1;
#pragma unroll
        for (int j = 0; j < NT / 2; ++j) {
          const int n = n0 + wn * WN + j * 16 + fq * 4;            // physical column of a
          const int no = ((n0 + wn * WN) >> 1) + j * 16 + fq * 4;  // output column
          f32x4 a = acc[i][j], g = acc[i][j + NT / 2];
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          acc[i][j + NT / 2] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (!m_ok) continue;
          if (ln_slow) {
            const float* ln_s = ln_args()->ln_s;
            a = (a - lns_mean[i] * *reinterpret_cast<const f32x4*>(ln_s + n)) * lns_rstd[i];
            g = (g - lns_mean[i] * *reinterpret_cast<const f32x4*>(ln_s + n + WN / 2)) * lns_rstd[i];
          }
          if (p.bias) {
            a += *reinterpret_cast<const f32x4*>(p.bias + n);
            g += *reinterpret_cast<const f32x4*>(p.bias + n + WN / 2);
          }
          T o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = from_f32<T>(a[q] * gelu_erf_f(g[q]));
          *reinterpret_cast<u32x2*>(out + (int64_t)m * No + no) = *reinterpret_cast<u32x2*>(o);
        }
      }
    }
    return 0;
  };

  if (my_tiles <= 0) return;
  // De-phase the two blocks that share a CU (blocks b and b + 256 of a 512-block persistent grid): they run the same tile
  // period from the same start, so both reach their epilogues (VALU + stores, matrix pipe idle) together.  Delaying the second
  // one by about half a tile lets each block's epilogue run under the other's matrix work.
  if (p.stagger > 0 && ((blockIdx.x >> 8) & 1)) {
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }
  const int total_steps = my_tiles * nk;
  int it_tile = 0, it_kt = 0;          // (tile, k-tile) being issued, ahead of the compute
  int ct_tile = 0, ct_kt = 0;          // (tile, k-tile) being computed
  auto advance_issue = [&]() __attribute__((always_inline)) {
    it_c0 += BK;
    if (it_c0 == cin) {
      it_c0 = 0;
      ++it_tap;
      if (++it_kx == 3) { it_kx = 0; ++it_ky; }
    }
    if (++it_kt == nk) {
      it_kt = 0;
      it_tap = it_ky = it_kx = 0;
      setup_issue(++it_tile);
    }
  };
  auto compute_stage = [&](int st) {
    const T* tA = sA + st * BM * BK;
    const T* tB = sB + st * BN * BK;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      frag fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int row = wm * WM + i * 16 + fr;
        fa[i] = *reinterpret_cast<const frag*>(tA + row * BK + (((kk * 4 + fq) ^ (row & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        int row = wn * WN + j * 16 + fr;
        fb[j] = *reinterpret_cast<const frag*>(tB + row * BK + (((kk * 4 + fq) ^ (row & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = Mfma<T>::run(fb[j], fa[i], acc[i][j]);
    }
    int cls = -1;   // -1: no epilogue in this step (no stores issued)
    if (++ct_kt == nk) {
      int m0, n0;
      ep_part = tile_origin(ct_tile, m0, n0);
      cls = epilogue(m0, n0, ct_tile);
      ct_kt = 0;
      ++ct_tile;
    }
    return cls;
  };

  setup_issue(0);
  if constexpr (STAGES != 3) {
    // Simple S-slot loop (S = STAGES = 2 or 4).  S = 2: the DMA of step s+1 is in flight during the MFMAs of step s, vmcnt(0) + barrier per K tile --
    // every K step costs one memory latency (0.62 us), covered only by the other resident blocks.  S = 4 (the 64 x 64 tiles of small launches,
    // where few blocks are resident: batch 1): three K tiles in flight, COUNTED vmcnt (the wait at the end of step s needs tile s+1 only and
    // leaves tiles s+2, s+3 -- and the epilogue's stores, youngest in the queue -- in flight), raw s_barrier (__syncthreads drains vmcnt).
    constexpr int S = STAGES;
    constexpr int NP = A_LOADS + B_LOADS;   // DMA pieces per wave and K tile (a bias piece at a tile switch only makes a wait cover more)
    static_assert(S == 2 || BN % RP == 0, "counted waits need the same piece count in every wave");
    constexpr int S1 = MT * ((NT + 1) / 2), S2 = MT * ((NT / 2 + 1) / 2), S3 = S1 + MT, S4 = S1 + 2 * NT;   // stores per wave and tile, per store class
#define ETAINV_VMCNT_IMM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
    // wait until all but `tiles` K tiles (and the stores of an epilogue of class cls) have landed
    auto wait_landed = [&](int tiles, int cls) __attribute__((always_inline)) {
      if (cls == 0) { ETAINV_VMCNT_IMM(0); return; }   // epilogue with an unknown store count
      const int st = cls == 1 ? S1 : cls == 2 ? S2 : cls == 3 ? S3 : cls == 4 ? S4 : 0;
      if constexpr (S == 2) {
        if (st == S1) ETAINV_VMCNT_IMM(S1); else if (st == S2) ETAINV_VMCNT_IMM(S2); else if (st == S3) ETAINV_VMCNT_IMM(S3);
        else if (st == S4) ETAINV_VMCNT_IMM(S4); else ETAINV_VMCNT_IMM(0);
      } else {
        static_assert(S == 2 || (S - 2) * NP + S4 < 64, "vmcnt field");
#define ETAINV_WAIT_T(T_)                                                                                         \
        if (st == S1) ETAINV_VMCNT_IMM((T_) * NP + S1); else if (st == S2) ETAINV_VMCNT_IMM((T_) * NP + S2);      \
        else if (st == S3) ETAINV_VMCNT_IMM((T_) * NP + S3); else if (st == S4) ETAINV_VMCNT_IMM((T_) * NP + S4); else ETAINV_VMCNT_IMM((T_) * NP);
        if (tiles >= 2) { ETAINV_WAIT_T(2) } else if (tiles == 1) { ETAINV_WAIT_T(1) } else { ETAINV_WAIT_T(0) }
#undef ETAINV_WAIT_T
      }
    };
    auto block_sync = [&]() __attribute__((always_inline)) {
      if constexpr (S == 2) {
        __syncthreads();
      } else {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS reads of the finished slot are done
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    static_assert(S == 2 || S == 4, "slot counts");
    issue_tile(0, 0);
    if constexpr (S > 2) {
#pragma unroll
      for (int q = 1; q < S - 1; ++q)
        if (q < total_steps) { advance_issue(); issue_tile(it_kt, q); }
    }
    wait_landed(total_steps - 1 < S - 2 ? total_steps - 1 : S - 2, -1);
    block_sync();
    int slot = 0;
    for (int sidx = 0; sidx < total_steps; ++sidx) {
      if (sidx + S - 1 < total_steps) {
        advance_issue();
        issue_tile(it_kt, slot == 0 ? S - 1 : slot - 1);   // the slot read in step sidx - 1 (every wave is past that step's barrier)
      }
      const int cls = compute_stage(slot);
      // vmcnt counts loads, stores and LDS-DMA together in issue order: the epilogue's stores are younger than every DMA piece issued so far,
      // so they may stay in flight across this wait (their L2 acks are not on the critical path)
      int ahead = total_steps - 2 - sidx;   // K tiles issued beyond tile sidx + 1
      ahead = ahead < 0 ? 0 : ahead > S - 2 ? S - 2 : ahead;
      wait_landed(ahead, cls);
      block_sync();
      slot = slot + 1 == S ? 0 : slot + 1;
    }
#undef ETAINV_VMCNT_IMM
  } else {
    // 3-slot LDS ring, software-pipelined fragments, COUNTED vmcnt, raw s_barrier.  Step s (K tile s of the flattened tile
    // stream, slot s % 3) is two straight-line windows, each one basic block so that the scheduler can place the non-matrix
    // instructions INSIDE the MFMA cluster (an MFMA occupies the matrix pipe for 16 cycles but the SIMD's issue port for 8):
    //   window 1:  20 MFMAs on F0 (k-half 0 of slot s, already in registers)  ||  9 ds_read_b128: F1 <- slot s, k-half 1
    //              lgkmcnt(0) (slot s may be recycled), vmcnt(N1) (slot s+1 landed for this wave), s_barrier (for every wave)
    //   window 2:  20 MFMAs on F1  ||  7 LDS-DMA pieces of step s+3 into slot s % 3 with their address arithmetic
    //                              ||  9 ds_read_b128: F0 <- slot s+1, k-half 0
    //   step end:  epilogue if this was the tile's last K tile; advance the issue position (tile switch + bias DMA here, off
    //              the windows).  Before this change the DMA issue (~100 cycles per piece) and the fragment reads ran with
    //              the matrix pipe idle: measured 47 % MFMA-busy on the large convs.
    // Every wave issues the same number of DMA pieces per step: the B tile has BN/8 pieces, the waves of the last, partial
    // pass that have no rows left aim their piece at a 1-KiB dummy area (uniform counted waits, no wave-dependent branch).
    static_assert(STAGES == 3, "ring");
    constexpr int B_PASSES = (BN + RP - 1) / RP;
    constexpr int N1 = A_LOADS + B_PASSES;   // DMA pieces per wave and step
    T* const dummy = reinterpret_cast<T*>(smem + ((size_t)A_REGION + (size_t)STAGES * BN * BK) * sizeof(T) + 4 * BN * sizeof(float));
    int ct_tap = 0, ct_gq = 0;   // PATCH: tap / running chunk count of the K tile being computed
    // (PATCH: pgq / ptap = patch slot parity and tap of the K tile the fragments belong to; wave row group i = patch row wm * 4 + i, lane fr = patch column)
    auto read_frags = [&](int st, int kk, u32x4 (&fa)[MT], u32x4 (&fb)[NT], int pgq = 0, int ptap = 0) __attribute__((always_inline)) {
      const T* tA = sA + st * BM * BK;
      const T* tB = sB + st * BN * BK;
      if constexpr (PATCH) {
        const int ky = ptap / 3, kx = ptap - ky * 3;
        const int col = fr + kx;
        const T* tP = sA + (pgq & 1) * (PROWS * BK) + ((wm * 4 + ky) * PW + col) * BK + (((kk * 4 + fq) ^ (col & 7)) << 3);
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const u32x4*>(tP + i * (PW * BK));
      } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int row = wm * WM + i * 16 + fr;
        fa[i] = *reinterpret_cast<const u32x4*>(tA + row * BK + (((kk * 4 + fq) ^ (row & 7)) << 3));
      }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        int row = wn * WN + j * 16 + fr;
        fb[j] = *reinterpret_cast<const u32x4*>(tB + row * BK + (((kk * 4 + fq) ^ (row & 7)) << 3));
      }
    };
    auto mfma_range = [&](u32x4 (&fa)[MT], u32x4 (&fb)[NT], auto lo_tag, auto hi_tag) __attribute__((always_inline)) {
      constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;   // MFMAs LO .. HI-1 of the cluster (row-major i, j)
#pragma unroll
      for (int idx = LO; idx < HI; ++idx) {
        const int i = idx / NT, j = idx % NT;
        acc[i][j] = Mfma<T>::run(__builtin_bit_cast(frag, fb[j]), __builtin_bit_cast(frag, fa[i]), acc[i][j]);
      }
    };
    auto mfma_all = [&](u32x4 (&fa)[MT], u32x4 (&fb)[NT]) __attribute__((always_inline)) {
      mfma_range(fa, fb, std::integral_constant<int, 0>{}, std::integral_constant<int, MT * NT>{});
    };
    // Branch-free issue of the K tile at the issue position into ring slot `buf`, in two parts: the activation pieces go out in
    // window 2 of step s, the weight pieces in window 1 of step s+1.  The CU's vector-memory path moves 64 B/clk, i.e. 16 clocks
    // per 1-KiB piece: all 56 pieces of a K tile right after the barrier are a 900-clock burst during which window 2 (640
    // clocks of matrix work) waits on its own last piece (in-kernel stamps: window 1 338 clocks, window 2 1154).
    // ---- PATCH mode issue state: the chunk stream (tile, channel chunk) -> two patch slots alternately.  Every step issues exactly ONE patch piece
    // per wave, branch-free (the windows must stay single basic blocks and the counted waits uniform): pieces of the NEXT chunk of the stream from tap 2 on
    // (its slot's previous patch is read until the rendezvous of compute step (chunk - 2, tap 8) = issue position (chunk - 1, tap 2) of the 3-step
    // run-ahead), a piece aimed at the dummy area otherwise (taps 0, 1; ids past the patch; no next tile).
    int it_q = 0, it_gq = 0;                 // chunk index inside the tile / running chunk count (patch slot = count & 1) of the issue position
    int pt_e0 = 0, pt_y0 = 0, pt_x0 = 0, pt_q = 0;   // patch being issued: element offset of its pixel (-1, -1) at its channel chunk q, its image origin
    bool pt_ok = false;
    // (the divisions -- ~125 cycles on the scalar / VALU path, in-kernel stamps of round 4 -- run once per TILE; a chunk change inside a tile only moves
    // the channel offset: this code sits between window 1 and the rendezvous of every ninth step)
    int pt_tile = -1;
    auto patch_target = [&](int tile, int q) __attribute__((always_inline)) {   // scalar work, once per chunk (advance_ring: off the windows)
      if (tile != pt_tile) {
        pt_tile = tile;
        pt_ok = tile < my_tiles;
        int m0, n0;
        tile_origin(pt_ok ? tile : 0, m0, n0);
        const int mt = m0 / BM, tpr = p.W / 16, tpi = (p.H / 16) * tpr;
        const int b = mt / tpi, r = mt - b * tpi, ty = r / tpr;
        pt_y0 = ty * 16;
        pt_x0 = (r - ty * tpr) * 16;
        pt_e0 = ((b * p.H + pt_y0 - 1) * p.W + pt_x0 - 1) * p.c1 + q * BK;
        pt_q = q;
      } else {
        pt_e0 += (q - pt_q) * BK;
        pt_q = q;
      }
    };
    const int lrow8 = lane >> 3;
    // piece `id` (rows id * 8 .. + 7 of the 18 x 18 patch, row-major incl. halo) of the target patch -> slot `ps`; !real: a dummy piece
    // (LDS image of a patch: row = patch pixel (pr, pc) row-major at pitch 18, physical 16-byte chunk = logical chunk ^ (pc & 7).  The key is the patch
    // COLUMN, not the row as in the tile image: rows that a fragment read touches are 16 consecutive columns of one patch row, so (row parity, key)
    // relate exactly as in the tile image -- conflict-free ds_read_b128 -- while the key of a read depends on the tap's kx only, not on ky or the row group)
    // !real: the piece goes to the dummy area from the zero page -- every wave issues exactly one piece per step INSIDE the MFMA cluster of window 2,
    // no branch (a piece issued behind the cluster cost ~170 cycles per step on the critical path: profiles/r04_conv_patch_stamps.log), uniform counts
    auto issue_patch_piece = [&](int ps, int id, bool real) __attribute__((always_inline)) {
      const int prow = id * 8 + lrow8;
      const int pr = (prow * 3641) >> 16, pc = prow - pr * PW;     // prow / 18 for prow < 1024
      const int y = pt_y0 - 1 + pr, x = pt_x0 - 1 + pc;
      const bool ok = real & (prow < PW * PW) & (y >= 0) & (y < p.H) & (x >= 0) & (x < p.W);
      // source = zero page + (ok ? (activations - zero page) + element offset : 0), computed for EVERY lane and masked: a conditional expression here
      // is compiled into a divergent branch, which cuts window 2 in two (9 fragment reads and ~50 address instructions between 6 and 20 MFMAs with the
      // matrix pipe idle: +200 cycles per step in the stamps)
      const uint64_t zp = reinterpret_cast<uint64_t>(zero_page);
      const uint64_t raw = (reinterpret_cast<uint64_t>(p.a1) - zp) +
                           (uint64_t)(unsigned)(pt_e0 + (pr * p.W + pc) * p.c1 + (((lane & 7) ^ (pc & 7)) << 3)) * sizeof(T);
      const uint64_t keep = ok ? ~0ull : 0ull;
      const T* g = reinterpret_cast<const T*>(zp + (raw & keep));
      // (wave-uniform, masked like the source: a select would become a scalar branch in the middle of the window)
      const int dummy_el = (int)(dummy - sA), real_el = ps * (PROWS * BK) + id * 8 * BK;
      T* dst = sA + __builtin_amdgcn_readfirstlane(dummy_el + ((real_el - dummy_el) & -(int)real));
#ifdef ETAINV_ABL_PATCH_ZERO   // timing experiment only (wrong results): every piece reads the zero page
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)zero_page, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      return;
#endif
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const int kcq = cin / BK;                // channel chunks per tile (PATCH: K tile kt = q * 9 + tap)
    auto issue_a = [&](int buf) __attribute__((always_inline)) {
      if constexpr (PATCH) {
        const int id = (it_tap - 2) * 8 + __builtin_amdgcn_readfirstlane(wid);
        issue_patch_piece((it_gq + 1) & 1, id & 63, (it_tap >= 2) & (id < PPIECES) & pt_ok);
        return;
      }
      const bool second = it_c0 >= p.c1;
      const T* src = reinterpret_cast<const T*>(second ? p.a2 : p.a1);
      const int cs = second ? p.c2 : p.c1;
      const int uoff = (it_ky * p.W + it_kx) * cs + (second ? it_c0 - p.c1 : it_c0);
      T* dA = sA + buf * BM * BK;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const bool ok = (a_mask[i] >> it_tap) & 1;
        const T* g;
        if constexpr (UPS == 1) {   // nine taps on the upsampled grid: the tap lands on source pixel ((oy + ky - 1) >> 1, (ox + kx - 1) >> 1)
          const int iy = (a_y[i] + it_ky) >> 1, ix = (a_x[i] + it_kx) >> 1;
          const unsigned elem = (unsigned)(((a_b[i] * p.H + iy) * p.W + ix) * cs + (second ? it_c0 - p.c1 : it_c0) + lchunk * 8);
          g = ok ? src + elem : zero_page;
        } else {   // (UPS == 2: the phase conv is a plain 2 x 2 conv on the source image -- same addressing, tap width 2)
          const unsigned elem = (unsigned)((second ? a_e2[i] : a_e1[i]) + uoff);   // garbage for halo lanes, never dereferenced
          g = ok ? src + elem : zero_page;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dA + (wrow0 + RP * i) * BK), 16, 0, 0);
      }
    };
    auto issue_b = [&](int buf) __attribute__((always_inline)) {
      T* dB = sB + buf * BN * BK;
      const int koff = PATCH ? it_tap * cin + it_q * BK : it_kt * BK;   // (PATCH: weights [N][tap][cin], K tile = (chunk q, tap))
#pragma unroll
      for (int i = 0; i < B_PASSES; ++i) {
        T* dst = (BN % RP == 0 || wrow0 + RP * i < BN) ? dB + (wrow0 + RP * i) * BK : dummy;   // wave-uniform select
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_row[i] + koff),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    };
    auto bias_dma = [&]() __attribute__((always_inline)) {   // bias of the tile at the issue position -> sBias[tile & 3]
      if (p.bias && wrow0 * 8 < BN) {
        const int c = wrow0 * 8 + lane;
        if (c < BN) {
          const int n = it_n0 + c < p.N ? it_n0 + c : p.N - 1;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + it_boff + n),
                                           (__attribute__((address_space(3))) void*)(sBias + it_bias_buf * BN + wrow0 * 8), 4, 0, 0);
          if constexpr (LN_STAGED)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ln_s + n),
                                             (__attribute__((address_space(3))) void*)(sLnS + it_bias_buf * BN + wrow0 * 8), 4, 0, 0);
        }
      }
      if constexpr (LN_STAGED) {   // (mean, rstd) of the tile's BM rows: 2 * BM floats = one dword per lane of the block
        static_assert(2 * BM == NTHR, "one DMA piece per wave");
        int f = it_m0 * 2 + tid;
        f = f < p.M * 2 ? f : p.M * 2 - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ln_stat + f),
                                         (__attribute__((address_space(3))) void*)(sLnStat + it_bias_buf * (2 * BM) + wrow0 * 8), 4, 0, 0);
      }
    };
    auto advance_ring = [&]() __attribute__((always_inline)) {
      bool chunk_changed = false;
      if constexpr (PATCH) {                 // chunk major: nine taps, then the next channel chunk
        if (++it_tap == 9) {
          it_tap = 0;
          ++it_q;
          ++it_gq;
          chunk_changed = true;
        }
      } else {
        it_c0 += BK;
        if (it_c0 == cin) {
          it_c0 = 0;
          ++it_tap;
          if (++it_kx == (UPS == 2 ? 2 : 3)) { it_kx = 0; ++it_ky; }
        }
      }
      if (++it_kt == nk) {
        it_kt = 0;
        it_tap = it_ky = it_kx = 0;
        it_q = 0;
        if (++it_tile < my_tiles) {
          setup_issue(it_tile);
          bias_dma();
        }
      }
      if constexpr (PATCH) {
        if (chunk_changed) {                 // the patch issued while the issue position walks chunk it_q is the one of the chunk AFTER it in the stream
          const bool last_q = it_q + 1 == kcq;
          patch_target(last_q ? it_tile + 1 : it_tile, last_q ? 0 : it_q + 1);
        }
      }
    };
#define ETAINV_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
#define ETAINV_LGKMCNT0() __builtin_amdgcn_s_waitcnt(0xC07F)
    // `young` = store class of an epilogue whose stores are YOUNGER than the DMA that has to have landed (vmcnt counts
    // loads, stores and LDS-DMA together, in issue order): they stay in flight across the wait, so their L2 acks are off the
    // critical path.  Valid for the step right after the epilogue, where the queue reads [A pieces s+2][stores][B pieces s+2]
    // (one step later the B pieces of s+2 must have landed and they are younger than the stores).
    // (A bias DMA issued at a tile switch is younger still: it only makes the wait cover one more piece of the newest slot.)
    auto wait_one_slot_in_flight = [&](int young = 0) {
      constexpr int S1 = MT * ((NT + 1) / 2), S2 = MT * ((NT / 2 + 1) / 2);
      if constexpr (PATCH) {   // pieces per wave of the K tile in flight: its weight pieces + the one patch (or dummy) piece of the last issue_a
        constexpr int NP_ = B_PASSES + 1;
        if (young == 1) ETAINV_VMCNT(NP_ + S1); else if (young == 4) ETAINV_VMCNT(NP_ + S1 + 2 * NT); else ETAINV_VMCNT(NP_);
        return;
      }
      if (young == 1) ETAINV_VMCNT(N1 + S1);
      else if (young == 3) ETAINV_VMCNT(N1 + S1 + MT);   // + the row-statistics stores of a LayerNorm producer
      else if (young == 4) ETAINV_VMCNT(N1 + S1 + 2 * NT);   // + the channel-statistics stores of a GroupNorm producer
      else if (young == 2) ETAINV_VMCNT(N1 + S2);
      else ETAINV_VMCNT(N1);
    };
    // prologue: K tiles 0 and 1 whole, of K tile 2 only the activation pieces (its weight pieces go out in step 0's window 1)
    bias_dma();
    if constexpr (PATCH) {   // the whole patch of the first chunk up front (41 pieces over the 8 waves), then the target moves to the second chunk
      patch_target(0, 0);
      for (int id = __builtin_amdgcn_readfirstlane(wid); id < PPIECES; id += 8) issue_patch_piece(0, id, true);
      patch_target(kcq > 1 ? 0 : 1, kcq > 1 ? 1 : 0);
    }
    issue_a(0);
    issue_b(0);
    if (total_steps > 1) { advance_ring(); issue_a(1); issue_b(1); }
    if (total_steps > 2) { advance_ring(); issue_a(2); }
    // (tile switches inside this prologue put bias pieces between the slots: the counts below then over-wait, never under-wait)
    if constexpr (PATCH) {   // K tile 1's pieces (weights + 1) and the piece of K tile 2's issue may stay in flight (nk >= 9: no tile switch in here)
      if (total_steps > 2) ETAINV_VMCNT(B_PASSES + 2);
      else ETAINV_VMCNT(0);
    } else {
    if (total_steps > 2) ETAINV_VMCNT(N1 + A_LOADS);
    else if (total_steps > 1) ETAINV_VMCNT(N1);
    else ETAINV_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
    u32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    read_frags(0, 0, fa0, fb0);
    int slot = 0;
    int young_cls = 0, young_steps = 0;   // stores of the last epilogue that later waits may leave in flight
#ifdef ETAINV_IGEMM_STAMPS
    uint64_t st_w1 = 0, st_wait = 0, st_bar = 0, st_w2 = 0, st_end = 0;   // diagnostic build only: s_memtime per step segment
    uint64_t st_adv = 0, st_lgkm = 0;                                     // ... and the parts of `wait`: advance_ring, lgkmcnt(0) (the rest is the counted vmcnt)
    const uint64_t clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();   // in-kernel clock = d(memtime)/d(memrealtime) x 100 MHz
#endif
    auto step = [&](int sidx, auto has_next_tag, auto has_issue_tag, auto has_pb_tag) __attribute__((always_inline)) {
      constexpr bool HAS_NEXT = decltype(has_next_tag)::value, HAS_ISSUE = decltype(has_issue_tag)::value;
      constexpr bool HAS_PB = decltype(has_pb_tag)::value;   // the weight pieces of the K tile whose activation pieces went out last step
      const int nslot = slot == 2 ? 0 : slot + 1;
      const int pslot = slot == 0 ? 2 : slot - 1;
#ifdef ETAINV_IGEMM_STAMPS
      const uint64_t t0 = __builtin_amdgcn_s_memtime();
#endif
      // ---- window 1a: the first NA MFMAs on F0 carry the F1 reads and the weight pieces.  The rendezvous comes right after them,
      // NOT at the end of the cluster: it only needs this wave's F1 reads (slot s free) and its DMA of step s+1 (slot s+1 landed),
      // neither depends on the MFMAs -- so the remaining MFMAs on F0 run after the barrier, fused with window 2 into one
      // straight-line block, and a wave waiting at the barrier leaves the matrix pipe to its SIMD partner instead of both draining it
      // (stamps: counted waits + barrier were ~540 of ~2200 cycles per K step with the rendezvous between the clusters).
#ifndef ETAINV_RING_EARLY_SYNC
#define ETAINV_RING_EARLY_SYNC 1
#endif
#ifndef ETAINV_RING_NA_BASE
#define ETAINV_RING_NA_BASE (MT + NT)
#endif
      constexpr int NA = (!HAS_NEXT || !ETAINV_RING_EARLY_SYNC) ? MT * NT
                         : (ETAINV_RING_NA_BASE + (HAS_PB ? B_PASSES : 0) + 2 < MT * NT ? ETAINV_RING_NA_BASE + (HAS_PB ? B_PASSES : 0) + 2 : MT * NT);
      read_frags(slot, 1, fa1, fb1, ct_gq, ct_tap);
      if constexpr (HAS_PB) issue_b(pslot);
      mfma_range(fa0, fb0, std::integral_constant<int, 0>{}, std::integral_constant<int, NA>{});
#ifndef ETAINV_W1_READS1
      // the F1 reads go two per MFMA at the head of the window: the LDS latency of the last one then lies under the rest of the cluster instead of
      // in front of the rendezvous (one per MFMA over the first nine: convs -3.0 ... -3.8 %, short-K GEMMs 0 ... -4 %)
#pragma unroll
      for (int q = 0; q < (MT + NT + 1) / 2; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      constexpr int RSLOTS = (MT + NT + 1) / 2;   // MFMAs that carry the reads
#else
      constexpr int RSLOTS = MT + NT;
#pragma unroll
      for (int q = 0; q < MT + NT; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 ds_read
      }
#endif
      if constexpr (HAS_PB) {
#pragma unroll
        for (int q = 0; q < B_PASSES; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // 1 DMA piece
        }
      }
      static_assert(NA >= RSLOTS + (HAS_PB ? B_PASSES : 0), "window 1a must hold its read and DMA slots");
      if constexpr (NA - RSLOTS - (HAS_PB ? B_PASSES : 0) > 0)
        __builtin_amdgcn_sched_group_barrier(0x008, NA - RSLOTS - (HAS_PB ? B_PASSES : 0), 0);
      __builtin_amdgcn_sched_barrier(0);
#ifdef ETAINV_IGEMM_STAMPS
      const uint64_t t1 = __builtin_amdgcn_s_memtime();
      uint64_t t2 = t1, t3 = t1;
#endif
      if constexpr (HAS_PB) advance_ring();   // that K tile is fully issued: move the issue position (tile switch + bias DMA here)
#ifdef ETAINV_IGEMM_STAMPS
      const uint64_t t1a = __builtin_amdgcn_s_memtime();
      st_adv += t1a - t1;
#endif
      if constexpr (HAS_NEXT) {
        ETAINV_LGKMCNT0();                   // F1 landed; slot may be recycled after the barrier
#ifdef ETAINV_IGEMM_STAMPS
        st_lgkm += __builtin_amdgcn_s_memtime() - t1a;
#endif
        if constexpr (HAS_ISSUE) {
          wait_one_slot_in_flight(young_steps > 0 ? young_cls : 0);
          --young_steps;
        } else {
          if (sidx + 2 < total_steps) wait_one_slot_in_flight();
          else ETAINV_VMCNT(0);
        }
#ifdef ETAINV_IGEMM_STAMPS
        t2 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef ETAINV_IGEMM_STAMPS
        t3 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ---- window 1b + 2: the rest of the F0 cluster, then the F1 cluster with the F0 reads of the next step and the
        // activation pieces of step s+3
        mfma_range(fa0, fb0, std::integral_constant<int, NA>{}, std::integral_constant<int, MT * NT>{});
        read_frags(nslot, 0, fa0, fb0, ct_tap == 8 ? ct_gq + 1 : ct_gq, ct_tap == 8 ? 0 : ct_tap + 1);
        if constexpr (HAS_ISSUE) issue_a(slot);
      }
      mfma_all(fa1, fb1);
      if constexpr (HAS_NEXT) {
        if constexpr (MT * NT - NA > 0) __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - NA, 0);
#pragma unroll
        for (int q = 0; q < MT + NT; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (HAS_ISSUE) {
#pragma unroll
          for (int q = 0; q < (PATCH ? 1 : A_LOADS); ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // 1 DMA piece
          }
        }
        if constexpr (MT * NT - (MT + NT) - (HAS_ISSUE ? (PATCH ? 1 : A_LOADS) : 0) > 0)
          __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - (MT + NT) - (HAS_ISSUE ? (PATCH ? 1 : A_LOADS) : 0), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef ETAINV_IGEMM_STAMPS
      const uint64_t t4 = __builtin_amdgcn_s_memtime();
      st_w1 += t1 - t0; st_wait += t2 - t1; st_bar += t3 - t2; st_w2 += t4 - t3;
#endif
      // ---- step end
      if constexpr (PATCH) {
        if (++ct_tap == 9) { ct_tap = 0; ++ct_gq; }
      }
      if (++ct_kt == nk) {
        int m0, n0;
        tile_origin(ct_tile, m0, n0);
        young_cls = epilogue(m0, n0, ct_tile);
        young_steps = (nk >= 3 && HAS_ISSUE) ? 1 : 0;
        ct_kt = 0;
        ++ct_tile;
      }
      slot = nslot;
#ifdef ETAINV_IGEMM_STAMPS
      st_end += __builtin_amdgcn_s_memtime() - t4;
#endif
    };
    int sidx = 0;
    for (; sidx + 3 < total_steps; ++sidx) step(sidx, std::true_type{}, std::true_type{}, std::true_type{});
    if (total_steps >= 3) { step(sidx, std::true_type{}, std::false_type{}, std::true_type{}); ++sidx; }   // last weight pieces
    for (; sidx + 1 < total_steps; ++sidx) step(sidx, std::true_type{}, std::false_type{}, std::false_type{});
    step(sidx, std::false_type{}, std::false_type{}, std::false_type{});
#ifdef ETAINV_IGEMM_STAMPS
    if (p.stamps && lane == 0) {
      uint64_t* o = p.stamps + ((size_t)blockIdx.x * 8 + wid) * 16;
      o[8] = st_adv; o[9] = st_lgkm;
      o[0] = st_w1; o[1] = st_wait; o[2] = st_bar; o[3] = st_w2; o[4] = st_end; o[5] = (uint64_t)total_steps;
      o[6] = __builtin_amdgcn_s_memtime() - clk0; o[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
#undef ETAINV_VMCNT
#undef ETAINV_LGKMCNT0
  }
}

// algorithmic HBM bytes of one launch: every activation element, weight and residual read once, the result written once (2-byte
// elements; a 3x3 conv reads its input once, not nine times; the fused upsample reads the SMALL source)
static double igemm_algo_bytes(const IGemmParams& p) {
  const double batch = (double)p.M / (double)p.rows_per_batch;
  const double in_px = p.taps != 1 ? batch * (double)p.H * (double)p.W : (double)p.M;
  const double out_el = (double)p.M * (double)(p.geglu ? p.N / 2 : p.N);
  return 2.0 * (in_px * (double)(p.c1 + p.c2) + (double)p.N * (double)p.taps * (double)(p.c1 + p.c2) + out_el * (p.out_f32 ? 2.0 : 1.0) +
                (p.residual ? out_el : 0.0));
}

// Tile order across the 8 XCDs (each has its own 4 MiB L2; workgroup b runs on XCD b & 7) -- an EXPERIMENT OF ROUND 3, OFF BY DEFAULT.  With the 1-D order
// every XCD walks ALL N tiles of its M range, so a weight matrix that does not fit its L2 is re-streamed from the fabric once per round of concurrently
// running tiles: the per-launch PMC pass (profiles/r03_pmc_per_shape_rows128.json) shows 3.6 GB fetched for the 0.18 GB of operands of the C = 640 GEGLU
// projection and 5-10 GB for the deep-K convs (3.5-4.3 TB/s at the fabric while those kernels run; 2.1x the algorithmic bytes over a UNet call).  A
// gm x gn XCD grid gives every XCD an N range (weights / gn: resident in L2, or re-streamed gn times less often) at the price of reading every
// activation panel from gn XCDs.  Measured with the model below (ETAINV_XCD_GN=model, profiles/r03_pmc_per_shape_rows128_xcd.json, same box): the
// GEGLU projections fetch 5x / 1.7x less (3.6 -> 0.7 GB at C = 640) and take the SAME time (1.06 vs 1.04 ms); the 3x3 convs fetch 2-3.6x MORE (their
// activation panel is re-read per tap and no longer shared by the XCD's N tiles) at +1 % time; a 128-row UNet call 92.4 vs 91.4 ms of igemm, the
// benchmark 4.117 vs 4.110 images/s.  Conclusion: the fabric traffic (served by the 256 MiB Infinity Cache) is not what bounds these kernels -- the
// 1-D order stays the default.  ETAINV_XCD_GN=model|2|4|8 switches the 2-D order on for the ring kernels.
//   model, bytes at the fabric per launch: weights W / gn per XCD, resident if <= 2.5 MB (read once), else once per round of that XCD's tiles;
//   activations gn x taps x A (taps: a 3x3 conv streams its input once per tap); gn = the minimiser over {1, 2, 4, 8} that divides the tile grid
static int pick_xcd_gn(const IGemmParams& p, int BM, int BN, int grid, int tiles) {
  if (p.ksplit > 1 || (tiles & 7) != 0 || (grid & 7) != 0) return 1;
  const int tiles_n = cdiv(p.N, BN), tiles_m = tiles / tiles_n;
  static const char* env = getenv("ETAINV_XCD_GN");
  if (!env) return 1;
  static const int forced = atoi(env);   // 0 for "model"
  const double esz = 2.0, K = (double)p.taps * (p.c1 + p.c2);
  const double W = (double)p.N * K * esz;
  const double batch = (double)p.M / (double)p.rows_per_batch;
  const double A = (p.taps == 9 ? batch * (double)p.H * (double)p.W : (double)p.M) * (double)(p.c1 + p.c2) * esz;
  if (forced == 1) return 1;
  int best = 1;
  double best_cost = 0.0;
  const double waves = std::max(1.0, (double)tiles / (double)grid);               // tile rounds of the persistent grid
  for (int gn = 1; gn <= 8; gn *= 2) {
    if (gn > 1 && (tiles_n % gn != 0 || tiles_m % (8 / gn) != 0)) continue;   // (gn = 1 is the 1-D order: no divisibility needed)
    if (forced > 1) {
      if (gn == forced) return gn;
      continue;
    }
    const double w_slice = W / gn;
    const double span = std::min(1.0, (double)(grid / 8) / (double)(tiles_n / gn));   // share of the XCD's N range one round of its CUs covers
    const double w_cost = w_slice <= 2.5e6 ? w_slice * 8.0 : w_slice * span * waves * 8.0;
    const double cost = w_cost + gn * (double)p.taps * A;
    if (best_cost == 0.0 || cost < best_cost * 0.9) {   // (10 % hysteresis towards the smaller gn)
      best = gn;
      best_cost = cost;
    }
  }
  return forced > 1 ? 1 : best;
}

template <typename T, int BM, int BN, int WAVES_M, int STAGES = 2, int UPS = 0, int LN = 0, bool PATCH = false>
static int launch_igemm_t(const IGemmParams& p_in, hipStream_t s, int* stat_P = nullptr) {
  IGemmParams p = p_in;
  // Statistics producers: the fast epilogue (whole wave tiles inside one image and inside N) writes, for a LayerNorm (stat_kind 0), one (mean, M2)
  // partial per row and wave-tile column, for a GroupNorm (stat_kind 1) per-channel sums per wave-tile row block; anything else reports 0 and the
  // caller runs the statistics pass over the output.  *stat_P: partials per row (kind 0) / rows per partial (kind 1).
  constexpr int WM_ = BM / WAVES_M, WN_ = BN / 2;
  if (p.stat_out && !(STAGES == 3 && UPS && p.stat_kind == 0) && !p.geglu && !p.out_nchw && !p.out_f32 && p.ksplit <= 1 && p.rows_per_batch % WM_ == 0 &&
      p.N % WN_ == 0 && p.N % 16 == 0) {
    p.stat_P = p.stat_kind == 0 ? p.N / WN_ : WM_;
  } else {
    p.stat_out = nullptr;
    p.stat_P = 0;
  }
  if (stat_P) *stat_P = p.stat_P;
  ETAINV_CHECK(!p.w_batch_stride || (p.rows_per_batch % BM == 0 && !p.geglu && p.taps == 1), "per-image weights: every M tile inside one image, plain 1x1");
  if constexpr (LN == 0) {   // one instantiation per role (the 256 x 128 ring only runs GEGLU: never a producer; fused upsample: GroupNorm producer only)
    if constexpr (!(STAGES == 3 && BN == 128)) {
      if constexpr (!UPS && !PATCH)
        if (p.stat_out && p.stat_kind == 0) return launch_igemm_t<T, BM, BN, WAVES_M, STAGES, UPS, 1, PATCH>(p, s, nullptr);
      if (p.stat_out && p.stat_kind == 1) return launch_igemm_t<T, BM, BN, WAVES_M, STAGES, UPS, 3, PATCH>(p, s, nullptr);
    }
    if constexpr (!UPS && !PATCH) {
      if constexpr (STAGES == 3 && BM == 256 && BN == 160)
        if (p.ln_stat && p.hm_heads) return launch_igemm_t<T, BM, BN, WAVES_M, STAGES, UPS, 4, PATCH>(p, s, nullptr);
      if (p.ln_stat) return launch_igemm_t<T, BM, BN, WAVES_M, STAGES, UPS, 2, PATCH>(p, s, nullptr);
    }
  }
  if constexpr (PATCH) p.stat_kind = p.stat_out ? 1 : 0;   // (a conv never emits LayerNorm row statistics)
  const int tiles = cdiv(p.M, BM) * cdiv(p.N, BN) * (STAGES != 3 && p.ksplit > 1 ? p.ksplit : 1);   // virtual tiles with split-K
  const size_t lds = (PATCH ? (size_t)(2 * 328 * 64 + STAGES * BN * BK) : (size_t)STAGES * (BM + BN) * BK) * sizeof(T) + 4 * BN * sizeof(float) + (STAGES == 3 ? 1024 : 0) +
                     (LN == 2 && STAGES == 3 && BN == 128 ? (size_t)4 * (BN + 2 * BM) * sizeof(float) : 0);   // staged s vectors and (mean, rstd) rows
  static bool attr_set[kMaxDevices] = {};   // per device: function attributes and the allocations below belong to the current device
  const int dev = current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, WAVES_M, STAGES, UPS, LN, PATCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  // persistent grid: as many blocks as are resident at once (LDS-limited: 160 KiB / lds per CU, 256 CUs), a multiple of 8
  const int per_cu = std::max(1, std::min(8, (int)(160 * 1024 / lds)));
  const int grid = std::min(tiles, 256 * per_cu);
#ifdef ETAINV_IGEMM_STAMPS
  static uint64_t* d_stamps = nullptr;
  IGemmParams ps = p;
  if (STAGES == 3 && env_on("ETAINV_IGEMM_STAMPS")) {
    if (!d_stamps) (void)hipMalloc(&d_stamps, 2048 * 8 * 16 * sizeof(uint64_t));
    (void)hipMemsetAsync(d_stamps, 0, 2048 * 8 * 16 * sizeof(uint64_t), s);
    ps.stamps = d_stamps;
  }
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WAVES_M, STAGES, UPS, LN, PATCH>), dim3(grid), dim3(WAVES_M * 128), lds, s, ps);
  if (ps.stamps) {
    (void)hipStreamSynchronize(s);
    std::vector<uint64_t> h((size_t)grid * 8 * 16);
    (void)hipMemcpy(h.data(), d_stamps, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
    double sum[5] = {0, 0, 0, 0, 0}, steps = 0, ck = 0, rt = 0, adv = 0, lgkm = 0;
    for (int b = 0; b < grid; ++b)
      for (int w = 0; w < 8; ++w) {
        const uint64_t* o = &h[((size_t)b * 8 + w) * 16];
        for (int k = 0; k < 5; ++k) sum[k] += (double)o[k];
        adv += (double)o[8];
        lgkm += (double)o[9];
        steps += (double)o[5];
        ck += (double)o[6];
        rt += (double)o[7];
      }
    fprintf(stderr, "[igemm stamps] in-kernel clock %.3f GHz (s_memtime / s_memrealtime x 100 MHz, mean over waves)\n", rt > 0 ? ck / rt * 0.1 : 0.0);
    fprintf(stderr, "[igemm stamps %dx%d M=%d N=%d K=%d] s_memtime ticks (core clocks) per step and wave: w1 %.2f wait %.2f barrier %.2f w2 %.2f end %.2f\n", BM, BN,
            p.M, p.N, p.taps * (p.c1 + p.c2), sum[0] / steps, sum[1] / steps, sum[2] / steps, sum[3] / steps, sum[4] / steps);
    fprintf(stderr, "[igemm stamps] of `wait`: advance_ring %.2f, lgkmcnt(0) %.2f, counted vmcnt %.2f\n", adv / steps, lgkm / steps, (sum[1] - adv - lgkm) / steps);
  }
  return 0;
#endif
  p.xcd_gn = STAGES == 3 ? pick_xcd_gn(p, BM, BN, grid, tiles) : 1;
  ProfScope prof(PROF_IGEMM, 2.0 * (double)p.M * (double)p.N * (double)(p.taps * (p.c1 + p.c2)), s, igemm_algo_bytes(p));
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WAVES_M, STAGES, UPS, LN, PATCH>), dim3(grid), dim3(WAVES_M * 128), lds, s, p);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

// out[m][n] = sum over the K parts (fixed order) + bias + time-embedding row + residual: the epilogue of a split-K launch
template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(IGemmParams p) {
  const int n4 = p.N >> 2;
  const int64_t total = (int64_t)p.M * n4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int m = (int)(idx / n4), n = (int)(idx - (int64_t)m * n4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.ws + (int64_t)m * p.N + n);
    for (int k = 1; k < p.ksplit; ++k) v += *reinterpret_cast<const f32x4*>(p.ws + ((int64_t)k * p.M + m) * p.N + n);
    if (p.ln_stat) {   // folded LayerNorm (see IGemmParams::ln_stat)
      v = (v - p.ln_stat[(int64_t)m * 2] * *reinterpret_cast<const f32x4*>(p.ln_s + n)) * p.ln_stat[(int64_t)m * 2 + 1];
    }
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + (int64_t)(m / p.rows_per_batch) * p.bias_batch_stride + n);
    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)(m / p.rows_per_batch) * p.rowvec_stride + n);
    if (p.residual) {
      const T* r = reinterpret_cast<const T*>(p.residual) + (int64_t)m * p.N + n;
      v[0] += to_f32(r[0]); v[1] += to_f32(r[1]); v[2] += to_f32(r[2]); v[3] += to_f32(r[3]);
    }
    T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.N + n) = *reinterpret_cast<u32x2*>(o);
  }
}

#ifndef ETAINV_EXPERIMENTS   // xsgemm.hip (stationary-activation GEMM, measured 22 % slower: profiles/HISTORY.md) is built by `EXPERIMENTS=1 build.sh` only
bool xs_gemm_applicable(const IGemmParams&, int) { return false; }
int launch_xs_gemm(const IGemmParams&, int, hipStream_t) { ETAINV_FAIL("xsgemm.hip is an experiment: build with EXPERIMENTS=1"); }
#endif

static int ring_min_tiles() {
  static const int v = getenv("ETAINV_RING_MIN_TILES") ? atoi(getenv("ETAINV_RING_MIN_TILES")) : 144;
  return v;
}

// head-major QKV output: only the LayerNorm-consumer fast path of the 256 x 160 ring implements it (the dispatch below sends exactly these launches there)
// phase form of the fused-upsample conv (IGemmParams::ups == 2, taps == 4, weights from launch_pack_ups4): the 256 x 160 ring only, whole tiles inside
// one phase block of one image, GroupNorm-producer or plain epilogue
bool igemm_ups4_ok(const IGemmParams& p, int dtype) {
  if (dtype == ETAINV_F32 || p.taps != 4 || p.stride != 1 || p.a2 || p.geglu || p.residual || p.rowvec || p.ln_stat || p.out_f32 || p.out_nchw || p.w_batch_stride || p.pad0) return false;
  if (p.N % 160 != 0 || p.Ho != 2 * p.H || p.Wo != 2 * p.W || p.M % (4 * p.H * p.W) != 0 || p.rows_per_batch != 4 * p.H * p.W) return false;
  // whole 256-row tiles per (image, phase) -- or, in phase-major row order, per phase with 64-row wave tiles inside one image (8 x 8, 24 x 24 sources)
  if ((p.H * p.W) % 256 != 0 && ((p.M >> 2) % 256 != 0 || (p.H * p.W) % 64 != 0 || !env_flag("ETAINV_UPS4_PM", true))) return false;
  if (p.stat_out && p.stat_kind != 1) return false;
  if (env_on("ETAINV_NO_RING") || !env_flag("ETAINV_UPS4", true)) return false;
  return (int64_t)cdiv(p.M, 256) * cdiv(p.N, 160) >= ring_min_tiles();
}

bool igemm_hm_ok(const IGemmParams& p, int dtype) {
  if (dtype == ETAINV_F32 || !p.hm_heads || !p.ln_stat || p.geglu || p.taps != 1 || p.a2 || p.residual || p.stat_out || p.out_f32 || p.out_nchw || p.w_batch_stride) return false;
  if ((p.hm_dim != 40 && p.hm_dim != 80) || p.hm_heads != 8 || p.hm_tokens > 16384 || p.M >= (1 << 24)) return false;
  if (p.N != 3 * p.hm_heads * p.hm_dim || p.N % 160 != 0 || p.hm_tokens % 256 != 0 || p.M % p.hm_tokens != 0 || p.rows_per_batch % 64 != 0) return false;
  if (env_on("ETAINV_NO_RING") || xs_gemm_applicable(p, dtype)) return false;
  return (int64_t)cdiv(p.M, 256) * cdiv(p.N, 160) >= ring_min_tiles();
}

int launch_igemm(const IGemmParams& p_in, int dtype, hipStream_t s, int* stat_P) {
  if (stat_P) *stat_P = 0;
  if (dtype == ETAINV_F32) return launch_igemm_f32(p_in, s);   // fp32-operand parity mode (f32path.hip)
  static void* zero_pages[kMaxDevices] = {};   // 256 zero bytes read by the halo lanes of the 3x3 taps (one-time allocation per device)
  const int dev = current_device();
  if (!zero_pages[dev]) {
    ETAINV_HIP(hipMalloc(&zero_pages[dev], 256));
    ETAINV_HIP(hipMemset(zero_pages[dev], 0, 256));
  }
  IGemmParams p = p_in;
  p.zeros = zero_pages[dev];
  if (const char* dbg = getenv("ETAINV_IGEMM_DEBUG")) p.debug = atoi(dbg);
  if (const char* sg = getenv("ETAINV_STAGGER")) p.stagger = atoi(sg);
  ETAINV_CHECK(p.a1 && p.w && p.out, "null pointer");
  ETAINV_CHECK(p.M > 0 && p.N > 0 && (p.N % 4) == 0, "N must be a positive multiple of 4");
  ETAINV_CHECK(p.c1 % BK == 0 && p.c2 % BK == 0 && (p.c1 + p.c2) > 0, "channel counts must be multiples of 64");
  ETAINV_CHECK(p.taps == 1 || p.taps == 9 || (p.taps == 4 && p.ups == 2), "taps must be 1 or 9 (4: the phase form of a fused-upsample conv, ups == 2)");
  ETAINV_CHECK(p.taps != 1 || (p.stride == 1 && !p.ups), "1x1 / Linear: stride 1, no upsample");
  ETAINV_CHECK(p.ups != 2 || igemm_ups4_ok(p, dtype), "phase form of the fused-upsample conv: not available for this launch (ask igemm_ups4_ok first)");
  ETAINV_CHECK(!p.geglu || (p.N % 128) == 0, "GEGLU needs N % 128 == 0");
  ETAINV_CHECK(p.rows_per_batch > 0, "rows_per_batch");
  ETAINV_CHECK(!p.rowvec || p.rowvec_stride >= p.N, "rowvec_stride");
  ETAINV_CHECK(!p.ln_stat || (p.ln_s && p.bias && p.taps == 1 && !p.a2), "folded LayerNorm: s / c vectors, plain GEMM");
  ETAINV_CHECK(!p.ln_stat || (!p.residual && !p.rowvec && !p.stat_out), "folded LayerNorm: no residual / row vector / statistics output on the consumer");
  ETAINV_CHECK(!p.ln_stat || (!p.out_f32 && !p.out_nchw), "folded LayerNorm: the consumer stores the compute dtype, row-major (the fp32 / NCHW epilogues do not apply mean / rstd)");
  ETAINV_CHECK(!p.hm_heads || igemm_hm_ok(p, dtype), "head-major QKV output: not available for this launch (ask igemm_hm_ok first)");
  if (p.hm_heads) p.hm_magic = (int)(((1ull << 38) + (unsigned)p.hm_tokens - 1) / (unsigned)p.hm_tokens);   // m0 / hm_tokens == (m0 * magic) >> 38 for m0 < 2^24, hm_tokens <= 2^14
  static const bool trace = env_on("ETAINV_TRACE_IGEMM");   // one line per launch on stderr, in launch order (tools/unet_call.py --shapes joins it with the event times)
  if (trace) {
    const bool ring = !p.geglu && !p.ups && p.N % 160 == 0 && (int64_t)cdiv(p.M, 256) * cdiv(p.N, 160) >= ring_min_tiles();
    fprintf(stderr, "igemm M=%d N=%d c1=%d c2=%d taps=%d stride=%d ups=%d H=%d W=%d geglu=%d ln=%d stat=%d res=%d rowvec=%d hm=%d route=%s\n", p.M, p.N, p.c1, p.c2, p.taps,
            p.stride, p.ups, p.H, p.W, (int)p.geglu, p.ln_stat ? 1 : 0, p.stat_out ? p.stat_kind : 0, p.residual ? 1 : 0, p.rowvec ? 1 : 0, p.hm_heads,
            pp_conv_applicable(p, dtype) ? "ppconv" : pp_dualn_applicable(p, dtype) ? "dualn" : pp_gemm_applicable(p, dtype) ? "ppgemm" : xs_gemm_applicable(p, dtype) ? "xs" :
            p.ups == 2 ? "ring-ups4" : ring ? "ring" : "other");
  }
  if (pp_conv_applicable(p, dtype)) {    // ping-pong PATCH conv3x3 (ppconv.hip)
    ProfScope prof(PROF_IGEMM, 2.0 * (double)p.M * (double)p.N * (double)(9 * p.c1), s, igemm_algo_bytes(p));
    return launch_pp_conv(p, dtype, s, stat_P);
  }
  if (pp_dualn_applicable(p, dtype)) {   // dual-N ping-pong kernel (ppgemm.hip)
    ProfScope prof(PROF_IGEMM, 2.0 * (double)p.M * (double)p.N * (double)(p.c1 + p.c2), s, igemm_algo_bytes(p));
    return launch_pp_dualn(p, dtype, s, stat_P);
  }
  if (pp_gemm_applicable(p, dtype)) {
    ProfScope prof(PROF_IGEMM, 2.0 * (double)p.M * (double)p.N * (double)p.c1, s, igemm_algo_bytes(p));
    return launch_pp_gemm(p, dtype, s, stat_P);
  }
  if (xs_gemm_applicable(p, dtype)) {   // K = 320 LayerNorm consumers with many rows: stationary activation tile, epilogue under the other wave group's MFMAs
    ProfScope prof(PROF_IGEMM, 2.0 * (double)p.M * (double)p.N * (double)p.c1, s, igemm_algo_bytes(p));
    return launch_xs_gemm(p, dtype, s);
  }
  // tile choice: big tiles when they still fill the 256 CUs, else 64x64 (GEGLU pairing is per wave tile,
  // so the packing of a GEGLU weight fixes its tile: always 128 wide)
  const int64_t big_tiles = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128);
  // ... and for a deep K (3x3 convs of the 8x8 / 16x16 levels at a few dozen rows: 180-360 K tiles) big tiles with split-K even when they
  // alone would leave most CUs empty: 640 resident 64 x 64 blocks each walked all K tiles at one memory latency per tile (182 us per launch,
  // 1.8 % of the benchmark step)
  static const int deepk_min_tiles = getenv("ETAINV_DEEPK_MIN_TILES") ? atoi(getenv("ETAINV_DEEPK_MIN_TILES")) : 64;
  static const int deepk_min_nk = getenv("ETAINV_DEEPK_MIN_NK") ? atoi(getenv("ETAINV_DEEPK_MIN_NK")) : 64;
  const bool deep_k = !p.geglu && p.N % 160 == 0 && p.taps * (p.c1 + p.c2) / BK >= deepk_min_nk && !p.out_nchw && !p.out_f32 &&
                      (int64_t)cdiv(p.M, 128) * cdiv(p.N, 160) >= deepk_min_tiles;
  const bool big = p.geglu || (big_tiles >= 192 && p.N > 64) || deep_k;
  ETAINV_CHECK(!p.out_nchw || p.N == 4, "out_nchw needs N == 4");
  const int64_t huge_tiles = (int64_t)cdiv(p.M, 256) * cdiv(p.N, 160);
  // a LayerNorm consumer on a ring kernel: fast epilogue only (64-row wave tiles inside one image)
  const bool ln_ring_ok = !p.ln_stat || p.rows_per_batch % 64 == 0;
  // (ring from `ring_min` 256 x 160 tiles on.  256 = one per CU was the round-2 threshold; the 96-row backward calls of round 3 bring 192 tiles at the
  // 8 x 8 level, where the ring on 3/4 of the CUs still beats the two-slot 128 x 160 kernel: same-box bench 4.897 (256) / 4.937 (192) / 4.910 (128)
  // images/s.  ETAINV_RING_MIN_TILES tunes it)
  // Round 6: 144.  Config 5's 12 x 12 level at 32 rows is 18 x 8 = 144 tiles (1280 -> 1280 convs with 180 K tiles: 599 TFLOP/s on the two-slot kernel): same-box
  // bench config 5 0.7583 (192) / 0.7697 (144) / 0.7674 (128) / 0.7677 (96) images/s, config 3 5.801 (192) / 5.797 (128): profiles/r06_ring_min_tiles_ab.log)
  static const int ring_min = ring_min_tiles();
  if (!p.geglu && p.ups == 2) {
    p.ups_pm = (p.H * p.W) % 256 != 0;
    // conv3x3 behind a nearest-2x upsample as four 2 x 2 phase convs (4 / 9 of the FLOPs): its own instantiation of the ring (row decode, output scatter)
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_igemm_t<T, 256, 160, 4, 3, 2, 0>(p, s, stat_P)));
  } else if (!p.geglu && p.ups == 1 && p.N % 160 == 0 && huge_tiles >= ring_min && !env_on("ETAINV_NO_RING")) {
    // the nine-tap form (images that are not whole 256-row tiles per phase): the ring's issue is branch-free, so its addressing is its own instantiation
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_igemm_t<T, 256, 160, 4, 3, 1, 0>(p, s, stat_P)));
  } else if (!p.geglu && p.taps == 9 && p.stride == 1 && !p.ups && !p.a2 && !p.pad0 && p.H % 16 == 0 && p.W % 16 == 0 && p.H == p.Ho && p.W == p.Wo &&
             p.N % 160 == 0 && huge_tiles >= ring_min && !p.ln_stat && !p.out_nchw && !p.out_f32 && !p.w_batch_stride &&
             (!p.stat_out || p.stat_kind == 1) && env_flag("ETAINV_PATCHCONV", true) && !env_on("ETAINV_NO_RING")) {
    // conv3x3 stride 1 on 16-pixel-aligned images: 16 x 16 pixel patches, the halo'd activation patch of a channel chunk loaded once for all nine taps
    // (same-box A/B, 128 rows: -1 ... -5 % per launch from the 16 x 16 level up, +1 % on the benchmark step; ETAINV_PATCHCONV=0 keeps the tap-major tiles)
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_igemm_t<T, 256, 160, 4, 3, 0, 0, true>(p, s, stat_P)));
  } else if (!p.geglu && !p.ups && p.N % 160 == 0 && huge_tiles >= ring_min && ln_ring_ok && !env_on("ETAINV_NO_RING")) {   // (a fused upsample that did not fill the ring runs on the two-slot kernels below)
    // experimental (opt-in): 256 x 160 x 64 tile, 8 waves, one resident block per CU (26 % fewer L2 -> LDS bytes per
    // FLOP).  Measured equal to 128 x 160 with two resident blocks (1026 vs 1033 TFLOP/s on conv 1280->1280 @16x16)
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_igemm_t<T, 256, 160, 4, 3, 0, 0>(p, s, stat_P)));
  } else if (p.geglu && ln_ring_ok && p.c1 >= (getenv("ETAINV_GEGLU_RING_MINK") ? atoi(getenv("ETAINV_GEGLU_RING_MINK")) : 320) && (int64_t)cdiv(p.M, 256) * cdiv(p.N, 128) >= 256 && !env_on("ETAINV_NO_RING")) {
    // (since the interleaved windows the ring also wins at K = 320: 1.42 vs 1.55 ms for ff1 320 -> 2560 at 64 x 64 x 128 rows)
    ETAINV_DISPATCH_HALF(dtype, T, return (launch_igemm_t<T, 256, 128, 4, 3, 0, 0>(p, s, stat_P)));
  } else {
    // two-slot kernels: 128 x 160 (every channel count of SD1.x is a multiple of 320: no padded columns, 20 MFMAs per 9 fragment
    // reads), 128 x 128 (GEGLU / other widths), 64 x 64 for small M*N.  A two-slot block is bound by one memory latency per K tile
    // (0.62 us) whatever its tile -- 80 blocks x 180 K tiles of a 1280 -> 1280 conv at 8 x 8 took 112 us for 29.5 MB of weights --
    // so a problem that leaves resident slots empty and has a deep K is SPLIT along K (9 parts: 29 us): fp32 partials, then a
    // fixed-order reduction that applies the epilogue.
    const int cfg = (big && !p.geglu && p.N % 160 == 0) ? 0 : big ? 1 : 2;
    const int bm = cfg == 2 ? 64 : 128, bn = cfg == 0 ? 160 : cfg == 1 ? 128 : 64;
    // resident blocks with two LDS slots: 64 x 64 tiles 4 per CU, the others 2 per CU
    const int slots = cfg == 2 ? 1024 : 512;
    const int64_t tiles = (int64_t)cdiv(p.M, bm) * cdiv(p.N, bn);
    const int nk = p.taps * (p.c1 + p.c2) / BK;
    int ks = 1;
    if (!p.geglu && !p.out_nchw && !p.out_f32 && tiles * 2 <= slots && nk >= 16 && !env_on("ETAINV_NO_SPLITK")) {
      static const int max_split = getenv("ETAINV_SPLITK_MAX") ? atoi(getenv("ETAINV_SPLITK_MAX")) : 32;
      for (int d = 2; d <= max_split; ++d)
        if (nk % d == 0 && nk / d >= 4 && tiles * d <= slots && (int64_t)p.M * p.N * d * 4 <= SPLITK_WS_BYTES) ks = d;
    }
    IGemmParams pk = p;
    if (ks > 1) {
      // one-time 64 MiB workspace per device.  Launches on one device are serialised by the caller's stream (the engine runs one stream);
      // concurrent split-K launches on two streams of the same device would share it.
      static float* ws[kMaxDevices] = {};
      if (!ws[dev]) ETAINV_HIP(hipMalloc(&ws[dev], SPLITK_WS_BYTES));
      pk.ksplit = ks;
      pk.ws = ws[dev];
      pk.bias = nullptr;
      pk.rowvec = nullptr;
      pk.residual = nullptr;
      pk.ln_stat = nullptr;    // applied by the reduction
      pk.stat_out = nullptr;   // (a split-K producer emits no statistics)
    }
    ProfScope prof(PROF_IGEMM, 2.0 * (double)p.M * (double)p.N * (double)(p.taps * (p.c1 + p.c2)), s, igemm_algo_bytes(p));
    prof_pause(true);
    int rc = 0;
    // (four LDS slots with three K tiles in flight -- the S = 4 form of the simple loop -- were measured for these kernels at batch 1: -37 % with
    // four slots everywhere (half the resident blocks), -1.5 % when only launches whose blocks are all resident anyway took it: a K step of a
    // lone block is bound by its own LDS-read -> MFMA chain, not by the memory latency)
    ETAINV_DISPATCH_HALF(dtype, T, rc = cfg == 0   ? launch_igemm_t<T, 128, 160, 2, 2, 0, 0>(pk, s, stat_P)
                                        : cfg == 1 ? launch_igemm_t<T, 128, 128, 2, 2, 0, 0>(pk, s, stat_P)
                                                   : launch_igemm_t<T, 64, 64, 2, 2, 0, 0>(pk, s, stat_P));
    prof_pause(false);
    if (rc) return rc;
    if (ks > 1) {
      IGemmParams pr = p;
      pr.ksplit = ks;
      pr.ws = pk.ws;
      const int64_t total = (int64_t)p.M * (p.N >> 2);
      ETAINV_DISPATCH_HALF(dtype, T, hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 2048)),
                                                        dim3(256), 0, s, pr));
      ETAINV_LAUNCH_CHECK();
    }
  }
  return 0;
}

}  // namespace etainv
