// GroupNorm(32)(+SiLU) and LayerNorm for NHWC activations -- HBM-bound kernels (SURVEY App. A.4:
// 61 GroupNorms + 48 LayerNorms per UNet forward).  16-byte vector loads, wave64 shuffle reductions,
// fixed-order partial sums (bitwise reproducible, no float atomics).
//
// GroupNorm runs as stats -> apply (the apply blocks reduce the per-chunk partial sums themselves).  The stats/apply kernels read up to two source tensors so
// the decoder's torch.cat([x, skip], dim=1) is consumed in place; `apply` writes the concatenated,
// normalised (and SiLU'd) tensor that feeds the following 3x3 convolution.
#include "common.h"
#include "kernels.h"

namespace etainv {

constexpr int GN_MAX_VEC_PER_LANE = 5;  // C <= 2560 -> 320 vectors of 8 channels -> 5 per lane
// pixels in flight per wave in the streaming loops (independent 16-byte loads; C = 320 only fills 40 of a wave's 64 lanes, so one pixel
// per iteration leaves the memory system with too few requests in flight)
#ifndef GN_UNROLL
#define GN_UNROLL 4
#endif

template <typename T> struct Vec8;
template <> struct Vec8<f16> { typedef f16x8 type; };
template <> struct Vec8<bf16> { typedef bf16x8 type; };

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]) {
  typename Vec8<T>::type r = *reinterpret_cast<const typename Vec8<T>::type*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)r[j];
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
  typename Vec8<T>::type r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (T)v[j];
  *reinterpret_cast<typename Vec8<T>::type*>(p) = r;
}

// grid (chunks, B); 256 threads = 4 waves; wave w takes pixels beg+w, beg+w+4, ...; lane l owns channel
// vectors l, l+64, ... so per-channel sums stay in registers across pixels.
// VPL = channel vectors per lane = ceil(C / 8 / 64): a template parameter, because the per-lane sums / scales live in registers (16 per
// vector): sized for C = 2560 (VPL 5, 140 VGPRs, 3 waves per SIMD) the C = 320 / 640 launches -- most of the GroupNorm time -- ran with a
// third of the waves, i.e. a third of the memory requests in flight, that their own register need allows
template <typename T, int VPL>
__global__ void __launch_bounds__(256) gn_stats_kernel(const T* __restrict__ x1, const T* __restrict__ x2, int c1, int c2, int hw,
                                                       int groups, float* __restrict__ partial) {
  const int C = c1 + c2, nvec = C >> 3, nv1 = c1 >> 3, cpg = C / groups;
  const int b = blockIdx.y, chunks = gridDim.x;
  const int per = (hw + chunks - 1) / chunks;
  const int beg = blockIdx.x * per, end = min(hw, beg + per);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float s[VPL][8], q[VPL][8];
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[i][j] = q[i][j] = 0.f;
#pragma unroll(VPL <= 2 ? GN_UNROLL : 1)
  for (int pix = beg + wid; pix < end; pix += 4) {
    const int64_t row = (int64_t)b * hw + pix;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int v = lane + 64 * i;
      if (v < nvec) {
        float t[8];
        if (v < nv1) load8(x1 + row * c1 + v * 8, t);
        else load8(x2 + row * c2 + (v - nv1) * 8, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[i][j] += t[j]; q[i][j] += t[j] * t[j]; }
      }
    }
  }
  extern __shared__ float sm[];  // [4][C][2]
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[((wid * C) + v * 8 + j) * 2 + 0] = s[i][j];
        sm[((wid * C) + v * 8 + j) * 2 + 1] = q[i][j];
      }
    }
  }
  __syncthreads();
  if (groups <= 32) {
    // eight threads per group, each a strided eighth of the group's 4 x cpg values, then a fixed-order butterfly over the eight (consecutive)
    // lanes.  One thread per group walked 4 x cpg dependent LDS reads: 13 us at 1280 channels, 29 us at 2560 -- invisible when thousands of
    // blocks overlap, the whole kernel at batch 1.
    const int g = threadIdx.x >> 3, sub = threadIdx.x & 7;
    float a = 0.f, c = 0.f;
    if (g < groups) {
      for (int w = 0; w < 4; ++w)
        for (int ch = g * cpg + sub; ch < (g + 1) * cpg; ch += 8) {
          a += sm[(w * C + ch) * 2 + 0];
          c += sm[(w * C + ch) * 2 + 1];
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      a += __shfl_xor(a, o);
      c += __shfl_xor(c, o);
    }
    if (g < groups && sub == 0) {
      float* p = partial + (((int64_t)b * chunks + blockIdx.x) * groups + g) * 2;
      p[0] = a;
      p[1] = c;
    }
  } else if (threadIdx.x < groups) {
    const int g = threadIdx.x;
    float a = 0.f, c = 0.f;
    for (int w = 0; w < 4; ++w)
      for (int ch = g * cpg; ch < (g + 1) * cpg; ++ch) {
        a += sm[(w * C + ch) * 2 + 0];
        c += sm[(w * C + ch) * 2 + 1];
      }
    float* p = partial + (((int64_t)b * chunks + blockIdx.x) * groups + g) * 2;
    p[0] = a;
    p[1] = c;
  }
}

// grid (chunks, B), 4 waves; same pixel / vector ownership as gn_stats_kernel, so the per-channel scale and shift
// (rstd*gamma, beta - mean*rstd*gamma) are computed once per lane and reused for every pixel: no integer
// division and 2 FMAs per element in the streaming loop.
template <typename T, int VPL>
__global__ void __launch_bounds__(256) gn_apply_kernel(const T* __restrict__ x1, const T* __restrict__ x2, int c1, int c2, int hw,
                                                       int groups, const float* __restrict__ partial, int chunks_st, float count,
                                                       float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int silu, T* __restrict__ out) {
  const int C = c1 + c2, nvec = C >> 3, nv1 = c1 >> 3, cpg = C / groups;
  const int b = blockIdx.y, chunks = gridDim.x;
  const int per = (hw + chunks - 1) / chunks;
  const int beg = blockIdx.x * per, end = min(hw, beg + per);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // (mean, rstd) of this image's groups from the per-chunk partial sums, in a fixed order (no separate finalize launch: at
  // batch 1 that 64-thread kernel was a 12 us chain of dependent loads, 8.5 % of the run)
  __shared__ double s_red[8][64][2];
  __shared__ float st[64 * 2];
  const bool pre_scsh = chunks_st == 0 && gamma == nullptr;   // `partial` = per-channel scale / shift planes [b][2][C] (gn_finalize_kernel)
  if (pre_scsh) {
  } else if (chunks_st == 0) {   // `partial` already holds (mean, rstd) per (image, group): gn_finalize_kernel
    if (threadIdx.x < groups) {
      st[threadIdx.x * 2 + 0] = partial[((int64_t)b * groups + threadIdx.x) * 2 + 0];
      st[threadIdx.x * 2 + 1] = partial[((int64_t)b * groups + threadIdx.x) * 2 + 1];
    }
    __syncthreads();
  } else {
    // thread = (group, one of NS strided slices of the chunks): the loads of a slice are independent and issued together (one thread per group
    // and wave walked chunks / 4 dependent global loads: 4-7 us in front of every apply block at batch 1)
    const int NS = groups <= 32 ? 8 : 4;
    const int g = groups <= 32 ? (threadIdx.x & 31) : lane, slice = groups <= 32 ? (threadIdx.x >> 5) : wid;
    double a = 0.0, c = 0.0;
    if (g < groups) {
      f32x2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = slice + u * NS;
        v[u] = (f32x2){0.f, 0.f};
        if (k < chunks_st) v[u] = *reinterpret_cast<const f32x2*>(partial + (((int64_t)b * chunks_st + k) * groups + g) * 2);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += v[u][0]; c += v[u][1]; }
      for (int k = slice + 8 * NS; k < chunks_st; k += NS) {   // (GN_MAX_CHUNKS = 64: never with 8 slices)
        const float* p = partial + (((int64_t)b * chunks_st + k) * groups + g) * 2;
        a += p[0];
        c += p[1];
      }
      s_red[slice][g][0] = a;
      s_red[slice][g][1] = c;
    }
    __syncthreads();
    if (threadIdx.x < groups) {
      const int g = threadIdx.x;
      double sa = 0.0, sq = 0.0;
      for (int w = 0; w < NS; ++w) { sa += s_red[w][g][0]; sq += s_red[w][g][1]; }
      const double mean = sa / count;
      double var = sq / count - mean * mean;
      if (var < 0.0) var = 0.0;
      st[g * 2 + 0] = (float)mean;
      st[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
  }
  if constexpr (VPL == 0) {
    // FLAT mode (nvec = 40 / 80 / 160 / 320, i.e. C = 320 / 640 / 1280 / 2560): a wave takes G = 320 / nvec pixels per iteration as 320 flat
    // (pixel, channel vector) items = 5 rounds of 64 lanes, every lane busy.  One pixel per round left 24 of 64 lanes idle at C = 320 (40
    // vectors) and the second round three quarters empty at C = 640 -- the two widths that carry most of the GroupNorm time.
    constexpr int R = 5;
    const int G = 320 / nvec;
    float sc[R][8], sh[R][8];
    int poff[R], voff[R];
    bool second[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int idx = lane + 64 * r;
      const int px = idx / nvec, v = idx - px * nvec;
      poff[r] = px;
      second[r] = v >= nv1;
      voff[r] = (second[r] ? v - nv1 : v) * 8;
      if (pre_scsh) {
        const float* ps = partial + ((int64_t)b * 2) * C + v * 8;
        *reinterpret_cast<f32x4*>(&sc[r][0]) = *reinterpret_cast<const f32x4*>(ps);
        *reinterpret_cast<f32x4*>(&sc[r][4]) = *reinterpret_cast<const f32x4*>(ps + 4);
        *reinterpret_cast<f32x4*>(&sh[r][0]) = *reinterpret_cast<const f32x4*>(ps + C);
        *reinterpret_cast<f32x4*>(&sh[r][4]) = *reinterpret_cast<const f32x4*>(ps + C + 4);
        continue;
      }
      int g = (v * 8) / cpg, rem = v * 8 - g * cpg;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = v * 8 + j;
        const float a = st[g * 2 + 1] * gamma[ch];
        sc[r][j] = a;
        sh[r][j] = beta[ch] - st[g * 2] * a;
        if (++rem == cpg) { rem = 0; ++g; }
      }
    }
    for (int pix0 = beg + wid * G; pix0 < end; pix0 += 4 * G) {
      float t[R][8];
      bool ok[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int pix = pix0 + poff[r];
        ok[r] = pix < end;
        const int64_t row = (int64_t)b * hw + (ok[r] ? pix : beg);
        if (second[r]) load8(x2 + row * c2 + voff[r], t[r]);
        else load8(x1 + row * c1 + voff[r], t[r]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float y = t[r][j] * sc[r][j] + sh[r][j];
          t[r][j] = silu ? silu_f(y) : y;
        }
        const int pix = pix0 + poff[r];
        if (ok[r]) store8(out + ((int64_t)b * hw + pix) * C + (second[r] ? nv1 * 8 : 0) + voff[r], t[r]);
      }
    }
    return;
  }
  constexpr int VP = VPL > 0 ? VPL : 1;
  float sc[VP][8], sh[VP][8];
#pragma unroll
  for (int i = 0; i < VP; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
      if (pre_scsh) {
        const float* ps = partial + ((int64_t)b * 2) * C + v * 8;
        *reinterpret_cast<f32x4*>(&sc[i][0]) = *reinterpret_cast<const f32x4*>(ps);
        *reinterpret_cast<f32x4*>(&sc[i][4]) = *reinterpret_cast<const f32x4*>(ps + 4);
        *reinterpret_cast<f32x4*>(&sh[i][0]) = *reinterpret_cast<const f32x4*>(ps + C);
        *reinterpret_cast<f32x4*>(&sh[i][4]) = *reinterpret_cast<const f32x4*>(ps + C + 4);
        continue;
      }
      int g = (v * 8) / cpg, rem = v * 8 - g * cpg;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = v * 8 + j;
        const float a = st[g * 2 + 1] * gamma[ch];
        sc[i][j] = a;
        sh[i][j] = beta[ch] - st[g * 2] * a;
        if (++rem == cpg) { rem = 0; ++g; }
      }
    }
  }
#pragma unroll(VP <= 2 ? GN_UNROLL : 1)
  for (int pix = beg + wid; pix < end; pix += 4) {
    const int64_t row = (int64_t)b * hw + pix;
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      const int v = lane + 64 * i;
      if (v < nvec) {
        float t[8];
        if (v < nv1) load8(x1 + row * c1 + v * 8, t);
        else load8(x2 + row * c2 + (v - nv1) * 8, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float y = t[j] * sc[i][j] + sh[i][j];
          t[j] = silu ? silu_f(y) : y;
        }
        store8(out + row * C + v * 8, t);
      }
    }
  }
}

// one wave per row, row kept in registers (C <= 1536), exact two-pass mean/variance
template <typename T>
__global__ void __launch_bounds__(256) layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, int rows, int C,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = C >> 3;
  float t[3][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
      load8(x + (int64_t)row * C + v * 8, t[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += t[i][j];
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = t[i][j] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (t[i][j] - mean) * rstd * gamma[v * 8 + j] + beta[v * 8 + j];
      store8(out + (int64_t)row * C + v * 8, o);
    }
  }
}

// Row statistics alone, (mean, rstd) as the LayerNorm-folded GEMMs read them (igemm.hip, IGemmParams::ln_stat): the fallback when the GEMM that
// wrote x could not emit partials from its epilogue (split-K, ragged wave tiles).
template <typename T>
__global__ void __launch_bounds__(256) row_stats_kernel(const T* __restrict__ x, float* __restrict__ stat, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = C >> 3;
  float t[3][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
      load8(x + (int64_t)row * C + v * 8, t[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += t[i][j];
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = t[i][j] - mean; sq += d * d; }
    }
  }
  sq = wave_sum(sq);
  if (lane == 0) {
    stat[(int64_t)row * 2] = mean;
    stat[(int64_t)row * 2 + 1] = rsqrtf(sq / (float)C + eps);
  }
}

// (mean, M2) partials of equal column counts, written per wave tile by the epilogue of the producing GEMM -> (mean, rstd) per row.  Chan's
// pairwise update in index order: no E[x^2] - E[x]^2 cancellation, the same result whatever order the producer's blocks ran in.
__global__ void __launch_bounds__(256) ln_finalize_kernel(const float* __restrict__ part, int P, float cw, float eps, float* __restrict__ stat, int rows) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float2* pr = reinterpret_cast<const float2*>(part) + (int64_t)row * P;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int q = 0; q < P; ++q) {
    const float2 v = pr[q];
    const float nt = n + cw, d = v.x - mean, f = cw / nt;
    mean += d * f;
    m2 += v.y + d * d * n * f;
    n = nt;
  }
  reinterpret_cast<float2*>(stat)[row] = make_float2(mean, rsqrtf(m2 / n + eps));
}

// Per-channel (sum, sum of squares) partials written by the epilogues of the GEMMs that produced x1 (c1 channels) and, for the decoder's concat,
// x2 (c2 channels) -- part[row block][2][c], row blocks of wm pixels, hw / wm of them per image -> final[b][group] = (mean, rstd).
// One wave per (image, group): the 64 lanes take strided (row block, channel) pairs, sums in double, fixed order (a block per image walked
// 320 dependent loads per thread at batch 1: slower than the statistics pass it replaces).
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ part1, int wm1, int c1, const float* __restrict__ part2, int wm2, int c2,
                                                          int hw, int groups, float eps, float* __restrict__ final_stats,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scsh) {
  const int b = blockIdx.y, C = c1 + c2, cpg = C / groups;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= groups) return;
  double a = 0.0, q = 0.0;
  // channels of the group that lie in source 1 / source 2 (a group may straddle the concat boundary)
  const int lo = g * cpg, hi = lo + cpg;
  const int n1 = max(0, min(hi, c1) - lo), nb1 = hw / wm1;
  for (int idx = lane; idx < n1 * nb1; idx += 64) {
    const int r = idx / n1, ch = lo + idx - r * n1;
    const float* base = part1 + ((int64_t)(b * nb1 + r) * 2) * c1 + ch;
    a += base[0];
    q += base[c1];
  }
  if (c2 > 0) {
    const int lo2 = max(lo, c1) - c1, n2 = hi - max(lo, c1), nb2 = hw / wm2;
    for (int idx = lane; idx < n2 * nb2; idx += 64) {
      const int r = idx / n2, ch = lo2 + idx - r * n2;
      const float* base = part2 + ((int64_t)(b * nb2 + r) * 2) * c2 + ch;
      a += base[0];
      q += base[c2];
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    a += __shfl_xor(a, o);
    q += __shfl_xor(q, o);
  }
  const double count = (double)hw * (double)cpg;
  const double mean = a / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float fmean = (float)mean, frstd = (float)(1.0 / sqrt(var + (double)eps));
  if (lane == 0) {
    final_stats[((int64_t)b * groups + g) * 2 + 0] = fmean;
    final_stats[((int64_t)b * groups + g) * 2 + 1] = frstd;
  }
  // per-channel scale / shift of the apply pass, [b][2][C]: computed once here instead of by every one of its blocks
  if (scsh)
    for (int ch = lo + lane; ch < hi; ch += 64) {
      const float sc = frstd * gamma[ch];
      scsh[((int64_t)b * 2 + 0) * C + ch] = sc;
      scsh[((int64_t)b * 2 + 1) * C + ch] = beta[ch] - fmean * sc;
    }
}

// blocks of the apply pass over the whole batch (each block computes its per-lane scale / shift vectors before it streams: fewer, longer
// blocks amortise that better)
static inline int gn_apply_blocks() {
  static const int n = getenv("ETAINV_GN_APPLY_BLOCKS") ? atoi(getenv("ETAINV_GN_APPLY_BLOCKS")) : 2048;
  return n;
}

// apply-pass instantiation: 0 = flat mode (C = 320 / 640: 320 vector items per wave iteration; -18 ... -25 % there, the wider tensors -- whose
// rounds were already full -- lose a few percent to the extra index arithmetic), else vectors per lane
static inline int gn_apply_mode(int C) {
  static const bool no_flat = env_on("ETAINV_GN_NOFLAT");
  const int nvec = C >> 3;
  if (!no_flat && (nvec == 40 || nvec == 80)) return 0;
  return (nvec + 63) / 64;
}

int launch_groupnorm(const void* x1, const void* x2, int c1, int c2, const float* gamma, const float* beta, void* out, int b,
                     int hw, int groups, float eps, int silu, float* scratch, int dtype, hipStream_t s) {
  if (dtype == ETAINV_F32) return launch_groupnorm_f32(x1, x2, c1, c2, gamma, beta, out, b, hw, groups, eps, silu, scratch, s);
  const int C = c1 + c2;
  ETAINV_CHECK(x1 && gamma && beta && out && scratch, "null pointer");
  ETAINV_CHECK(c1 % 8 == 0 && c2 % 8 == 0 && C % groups == 0 && groups <= 64, "channel layout");
  ETAINV_CHECK((C >> 3) <= 64 * GN_MAX_VEC_PER_LANE, "C too large");
  ETAINV_CHECK(c2 == 0 || x2, "second source missing");
  int chunks = std::min(GN_MAX_CHUNKS, std::max(1, std::min(hw / 8, 1024 / std::max(1, b))));
  float* partial = scratch;
  const size_t lds = (size_t)4 * C * 2 * sizeof(float);
  // the apply pass has no cross-block reduction: use more, smaller chunks to fill the chip
  const int chunks_apply = std::max(1, std::min(hw / 4, std::max(chunks, gn_apply_blocks() / std::max(1, b))));
  ProfScope prof(PROF_GROUPNORM, 2.0 * 2.0 * (double)b * hw * C, s);  // algorithmic bytes: read + write once, 2-byte elements
  const int vpl = ((C >> 3) + 63) / 64;
#define ETAINV_GN_LAUNCH(VPL_)                                                                                                             \
  {                                                                                                                                        \
    static bool attr_[kMaxDevices] = {};                                                                                                   \
    if (!attr_[current_device()]) {                                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_stats_kernel<T, VPL_>), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                4 * 64 * VPL_ * 8 * 2 * 4);                                                                                 \
      attr_[current_device()] = true;                                                                                                      \
    }                                                                                                                                      \
    hipLaunchKernelGGL((gn_stats_kernel<T, VPL_>), dim3(chunks, b), dim3(256), lds, s, (const T*)x1, (const T*)x2, c1, c2, hw, groups, partial); \
  }
#define ETAINV_GN_APPLY(VPL_)                                                                                                              \
  hipLaunchKernelGGL((gn_apply_kernel<T, VPL_>), dim3(chunks_apply, b), dim3(256), 0, s, (const T*)x1, (const T*)x2, c1, c2, hw, groups,     \
                     (const float*)partial, chunks, (float)hw * (float)(C / groups), eps, gamma, beta, silu, (T*)out);
  ETAINV_DISPATCH_HALF(dtype, T, switch (vpl) {
    case 1: ETAINV_GN_LAUNCH(1) break;
    case 2: ETAINV_GN_LAUNCH(2) break;
    case 3: ETAINV_GN_LAUNCH(3) break;
    case 4: ETAINV_GN_LAUNCH(4) break;
    default: ETAINV_GN_LAUNCH(5) break;
  } switch (gn_apply_mode(C)) {
    case 0: ETAINV_GN_APPLY(0) break;
    case 1: ETAINV_GN_APPLY(1) break;
    case 2: ETAINV_GN_APPLY(2) break;
    case 3: ETAINV_GN_APPLY(3) break;
    case 4: ETAINV_GN_APPLY(4) break;
    default: ETAINV_GN_APPLY(5) break;
  });
#undef ETAINV_GN_APPLY
#undef ETAINV_GN_LAUNCH
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_gn_finalize(int c1, int c2, const float* part1, int wm1, const float* part2, int wm2, int b, int hw, int groups, float eps, float* final_stats,
                       hipStream_t s, const float* gamma, const float* beta, float* scsh) {
  ETAINV_CHECK(part1 && final_stats && (c1 + c2) % groups == 0 && groups <= 32, "bad arguments");
  ETAINV_CHECK(wm1 > 0 && hw % wm1 == 0 && (c2 == 0 || (part2 && wm2 > 0 && hw % wm2 == 0)), "row blocks must tile an image");
  ProfScope prof(PROF_GROUPNORM, 0.0, s);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((groups + 3) / 4, b), dim3(256), 0, s, part1, wm1, c1, part2, wm2, c2, hw, groups, eps, final_stats, gamma, beta,
                     scsh);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_groupnorm_pre(const void* x1, const void* x2, int c1, int c2, const float* part1, int wm1, const float* part2, int wm2, const float* gamma,
                         const float* beta, void* out, int b, int hw, int groups, float eps, int silu, float* final_stats, int dtype, hipStream_t s) {
  const int C = c1 + c2;
  ETAINV_CHECK(x1 && gamma && beta && out && part1 && final_stats, "null pointer");
  ETAINV_CHECK(c1 % 8 == 0 && c2 % 8 == 0 && C % groups == 0 && groups <= 32, "channel layout");
  ETAINV_CHECK((C >> 3) <= 64 * GN_MAX_VEC_PER_LANE, "C too large");
  ETAINV_CHECK(c2 == 0 || (x2 && part2), "second source missing");
  ETAINV_CHECK(wm1 > 0 && hw % wm1 == 0 && (c2 == 0 || (wm2 > 0 && hw % wm2 == 0)), "row blocks must tile an image");
  ProfScope prof(PROF_GROUPNORM, 2.0 * 2.0 * (double)b * hw * C, s);
  prof_pause(true);
  float* scsh = final_stats + (int64_t)b * groups * 2;   // final_stats: [b][groups][2] (mean, rstd), then the scale / shift planes [b][2][C]
  const int rc = launch_gn_finalize(c1, c2, part1, wm1, part2, wm2, b, hw, groups, eps, final_stats, s, gamma, beta, scsh);
  prof_pause(false);
  if (rc) return 1;
  const int chunks_apply = std::max(1, std::min(hw / 4, std::max(std::min(GN_MAX_CHUNKS, std::max(1, std::min(hw / 8, 1024 / std::max(1, b)))), gn_apply_blocks() / std::max(1, b))));
#define ETAINV_GN_APPLY(VPL_)                                                                                                              \
  hipLaunchKernelGGL((gn_apply_kernel<T, VPL_>), dim3(chunks_apply, b), dim3(256), 0, s, (const T*)x1, (const T*)x2, c1, c2, hw, groups,     \
                     (const float*)scsh, 0, (float)hw * (float)(C / groups), eps, (const float*)nullptr, (const float*)nullptr, silu, (T*)out);
  ETAINV_DISPATCH_HALF(dtype, T, switch (gn_apply_mode(C)) {
    case 0: ETAINV_GN_APPLY(0) break;
    case 1: ETAINV_GN_APPLY(1) break;
    case 2: ETAINV_GN_APPLY(2) break;
    case 3: ETAINV_GN_APPLY(3) break;
    case 4: ETAINV_GN_APPLY(4) break;
    default: ETAINV_GN_APPLY(5) break;
  });
#undef ETAINV_GN_APPLY
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_layernorm(const void* x, const float* gamma, const float* beta, void* out, int rows, int c, float eps, int dtype,
                     hipStream_t s) {
  if (dtype == ETAINV_F32) return launch_layernorm_f32(x, gamma, beta, out, rows, c, eps, s);
  ETAINV_CHECK(x && gamma && beta && out, "null pointer");
  ETAINV_CHECK(c % 8 == 0 && (c >> 3) <= 192, "LayerNorm width must be a multiple of 8 and <= 1536");
  ProfScope prof(PROF_LAYERNORM, 2.0 * 2.0 * (double)rows * c, s);
  ETAINV_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL(layernorm_kernel<T>, dim3(cdiv(rows, 4)), dim3(256), 0, s, (const T*)x, gamma, beta,
                                          (T*)out, rows, c, eps));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_row_stats(const void* x, float* stat, int rows, int c, float eps, int dtype, hipStream_t s) {
  ETAINV_CHECK(x && stat, "null pointer");
  ETAINV_CHECK(c % 8 == 0 && (c >> 3) <= 192, "row width must be a multiple of 8 and <= 1536");
  ProfScope prof(PROF_LAYERNORM, 2.0 * (double)rows * c, s);
  ETAINV_DISPATCH_HALF(dtype, T, hipLaunchKernelGGL(row_stats_kernel<T>, dim3(cdiv(rows, 4)), dim3(256), 0, s, (const T*)x, stat, rows, c, eps));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_ln_finalize(const float* partials, int P, int cw, float eps, float* stat, int rows, hipStream_t s) {
  ETAINV_CHECK(partials && stat && P > 0 && cw > 0 && rows > 0, "bad arguments");
  ProfScope prof(PROF_LAYERNORM, (double)rows * P * 8.0, s);
  hipLaunchKernelGGL(ln_finalize_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, partials, P, (float)cw, eps, stat, rows);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

}  // namespace etainv
