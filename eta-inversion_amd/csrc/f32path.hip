// fp32-operand execution of the UNet (compute_dtype == ETAINV_F32): the reference's DEFAULT precision (edit_image.py:147 `--prec` None ->
// fp32, modules/models/__init__.py:104-138) and the mode in which north_star's rtol 1e-3 / atol 1e-4 on edited latents is checked.
//
// Every contraction runs on the f32-input matrix instruction v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation, bit-for-bit a
// k-ordered fmaf chain; 157 TFLOP/s peak = 1/16 of the bf16 rate), activations and weights stay fp32 in HBM.  This is a parity mode, not
// the throughput mode: the structure is the plain LDS-tiled kernel of the programming guide (128 x 128 x 32 block, 2 x 2 tiles of 32 x 32 per
// wave, register-prefetched global loads), without the ring / DMA / epilogue machinery of igemm.hip, and without the LayerNorm / GroupNorm
// folds (the engine runs the standalone norms in this mode).  What it must share with the 16-bit path is semantics: implicit-GEMM addressing
// (3x3 stride 1 / 2, fused nearest-2x upsample, pad0, dual source), bias / time row / residual / GEGLU epilogues, NCHW output, the attention
// row remaps of prompt-to-prompt and MasaCtrl, the cross-attention edit and the map store.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace etainv {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// C / D layout of the 32x32 shapes: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// ================================================================================================ implicit GEMM
constexpr int FBM = 128, FBN = 128, FBK = 32, FSTR = FBK + 4;

template <typename TIO>
__device__ __forceinline__ void store_io(void* p, int64_t i, float v) { reinterpret_cast<TIO*>(p)[i] = from_f32<TIO>(v); }

// out[m][n] = sum_k A[m][k] W[n][k]: rows m (pixels) are the MFMA's A rows -> accumulator registers, columns n (output channels) lie on the lanes,
// so that a store instruction writes 32 consecutive channels of a pixel (128 contiguous bytes) and a GEGLU value / gate pair (physical
// columns c and c + 32 of a 64-column group, pack mode 2) sits in the same lane and register of two accumulator tiles.
__global__ void __launch_bounds__(256) igemm_f32_kernel(IGemmParams p) {
  __shared__ __attribute__((aligned(16))) float sA[FBM * FSTR];
  __shared__ __attribute__((aligned(16))) float sB[FBN * FSTR];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * FBM, n0 = blockIdx.y * FBN;
  const int cin = p.c1 + p.c2, K = p.taps * cin, nk = K / FBK;
  const int lr = tid >> 3, kc = (tid & 7) * 4;
  const int pad = (p.taps == 9 && !p.pad0) ? 1 : 0;
  const int HWo = p.Ho * p.Wo;
  const int Hin = p.ups ? 2 * p.H : p.H, Win = p.ups ? 2 * p.W : p.W;
  const float* a1 = reinterpret_cast<const float*>(p.a1);
  const float* a2 = reinterpret_cast<const float*>(p.a2);
  const float* w = reinterpret_cast<const float*>(p.w);

  int ab[4], ay[4], ax[4];
  bool aok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + lr + 32 * i;
    aok[i] = m < p.M;
    m = aok[i] ? m : p.M - 1;
    if (p.taps == 9) {
      const int b = m / HWo, r = m - b * HWo, oy = r / p.Wo, ox = r - oy * p.Wo;
      ab[i] = b;
      ay[i] = oy * p.stride - pad;
      ax[i] = ox * p.stride - pad;
    } else {
      ab[i] = m;   // row-major [M][cin]
      ay[i] = ax[i] = 0;
    }
  }
  f32x4 ra[4], rb[4];
  auto load = [&](int kt) {
    const int k0 = kt * FBK, tap = k0 / cin, c0 = k0 - tap * cin, ky = tap / 3, kx = tap - ky * 3;
    const bool second = c0 >= p.c1;
    const float* src = second ? a2 : a1;
    const int cs = second ? p.c2 : p.c1, coff = (second ? c0 - p.c1 : c0) + kc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p.taps == 9) {
        int iy = ay[i] + ky, ix = ax[i] + kx;
        const bool ok = aok[i] && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
        if (p.ups) { iy >>= 1; ix >>= 1; }
        if (ok) v = *reinterpret_cast<const f32x4*>(src + ((int64_t)(ab[i] * p.H + iy) * p.W + ix) * cs + coff);
      } else if (aok[i]) {
        v = *reinterpret_cast<const f32x4*>(src + (int64_t)ab[i] * cs + coff);
      }
      ra[i] = v;
      const int n = n0 + lr + 32 * i;
      rb[i] = n < p.N ? *reinterpret_cast<const f32x4*>(w + (int64_t)n * K + k0 + kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  // Two-level accumulation: the matrix instruction adds its products in k order (one rounding per product, a single chain), so a K = 11520
  // conv would carry a 11520-term sequential sum; every FLUSH K tiles (64 k) the chain is closed into `tot` and restarted from zero -- the
  // error of a sum of n terms grows like sqrt(n), and sqrt(64) + sqrt(K / 64) is 5x smaller than sqrt(K) at K = 11520 (measured against a
  // float64 run of the oracle: tests/test_fp32_gpu.py::test_fp32_noise_floor_against_fp64).  64 v_add per 128 MFMAs.
  constexpr int FLUSH = 2;
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = tot[i][j][r] = 0.f;

  load(0);
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(sA + (lr + 32 * i) * FSTR + kc) = ra[i];
      *reinterpret_cast<f32x4*>(sB + (lr + 32 * i) * FSTR + kc) = rb[i];
    }
    __syncthreads();
    if (kt + 1 < nk) load(kt + 1);   // in flight under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < FBK / 8; ++kk) {
      // lane (row l31, half hh) takes k = 8 kk + 4 hh + c for the c-th MFMA of the group: the same bijection on both operands
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const f32x4*>(sA + (wm * 64 + i * 32 + l31) * FSTR + kk * 8 + hh * 4);
        fb[i] = *reinterpret_cast<const f32x4*>(sB + (wn * 64 + i * 32 + l31) * FSTR + kk * 8 + hh * 4);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i][c], fb[j][c], acc[i][j]);
    }
    if ((kt % FLUSH) == FLUSH - 1 || kt == nk - 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          tot[i][j] += acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = tot[i][j];

  // ---- epilogue
  float* out = reinterpret_cast<float*>(p.out);
  const float* res = reinterpret_cast<const float*>(p.residual);
  if (p.geglu) {
    const int No = p.N >> 1;
    const int nv = n0 + wn * 64 + l31, ng = nv + 32, no = (n0 + wn * 64) / 2 + l31;
    const float bv = p.bias ? p.bias[nv] : 0.f, bg = p.bias ? p.bias[ng] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + crow(r, hh);
        if (m < p.M && ng < p.N) {
          const float a = acc[i][0][r] + bv, g = acc[i][1][r] + bg;
          float v = a * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f)));
          if (res) v += res[(int64_t)m * No + no];
          out[(int64_t)m * No + no] = v;
        }
      }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    if (n >= p.N) continue;
    const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + crow(r, hh);
        if (m >= p.M) continue;
        float v = acc[i][j][r] + bn;
        const int b = m / p.rows_per_batch;
        if (p.rowvec) v += p.rowvec[(int64_t)b * p.rowvec_stride + n];
        if (res) v += res[(int64_t)m * p.N + n];
        if (p.out_nchw) {
          if (n < p.out_nchw) {
            const int64_t o = ((int64_t)b * p.out_nchw + n) * p.rows_per_batch + (m - b * p.rows_per_batch);
            if (p.out_io_dtype == ETAINV_F32) store_io<float>(p.out, o, v);
            else if (p.out_io_dtype == ETAINV_F16) store_io<f16>(p.out, o, v);
            else store_io<bf16>(p.out, o, v);
          }
        } else {
          out[(int64_t)m * p.N + n] = v;
        }
      }
  }
}

// ================================================================================================ norms
// GroupNorm over NHWC cat[x1 (c1), x2 (c2)]: statistics in double per (image, group) -- a group may straddle the two sources
__global__ void __launch_bounds__(256) gn_stats_f32_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int c1, int c2, int hw, int groups,
                                                           float eps, float* __restrict__ stats) {
  const int g = blockIdx.x, b = blockIdx.y, C = c1 + c2, cpg = C / groups;
  const int64_t n = (int64_t)hw * cpg;
  double s = 0.0, ss = 0.0;
  for (int64_t idx = threadIdx.x; idx < n; idx += 256) {
    const int pix = (int)(idx / cpg), c = g * cpg + (int)(idx - (int64_t)pix * cpg);
    const float v = c < c1 ? x1[((int64_t)b * hw + pix) * c1 + c] : x2[((int64_t)b * hw + pix) * c2 + (c - c1)];
    s += v;
    ss += (double)v * v;
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = sh[0][0] / (double)n, var = fmax(sh[1][0] / (double)n - mean * mean, 0.0);
    stats[((int64_t)b * groups + g) * 2] = (float)mean;
    stats[((int64_t)b * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
__global__ void __launch_bounds__(256) gn_apply_f32_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int c1, int c2, int hw, int groups,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ stats, int silu, float* __restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int C = c1 + c2, cpg = C / groups;
  const int c = (int)(i % C);
  const int64_t bp = i / C;
  const int b = (int)(bp / hw);
  const float v = c < c1 ? x1[bp * c1 + c] : x2[bp * c2 + (c - c1)];
  const float* st = stats + ((int64_t)b * groups + c / cpg) * 2;
  float y = (v - st[0]) * st[1] * gamma[c] + beta[c];
  if (silu) y = y / (1.0f + expf(-y));
  out[i] = y;
}
// LayerNorm: one wave per row, two passes over the row held in registers (c <= 64 * 20)
__global__ void __launch_bounds__(256) layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ out, int rows, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * C;
  float v[20];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 20; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < C ? xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 20; ++i) {
    const int c = lane + 64 * i;
    const float d = c < C ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 20; ++i) {
    const int c = lane + 64 * i;
    if (c < C) out[(int64_t)row * C + c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
  }
}

// ================================================================================================ self-attention
// One wave per 32 queries of one (batch row, head); S^T = K Q^T per 32-key block (keys -> accumulator registers, queries -> lanes), online
// softmax per query (the two lanes of a query exchange through lane ^ 32), O^T += V^T P^T with each accumulator register of S^T used directly
// as the B operand of one k = 2 step (lane half hh supplies key crow(r, hh): V^T is read in that order) -- no lane movement, no LDS for P.
// mode 1 / 2: the prompt-to-prompt / MasaCtrl batch-row remaps of attention.hip (row_roles).
template <int D>
__global__ void __launch_bounds__(256) self_attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int heads, float scale_log2,
                                                            int mode, int n_img, int first_row) {
  constexpr int NT = (D + 31) / 32, KSTR = D + 4, VSTR = NT * 32 + 8, KB = 64;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* sK = smem_f;                 // [KB][KSTR]
  float* sV = sK + KB * KSTR;         // [KB][VSTR], columns >= D zero
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, hd = blockIdx.y, C = heads * D, C3 = 3 * C;
  int bq = b, bk = b, bv = b;
  if (mode != 0) {
    if (first_row < 0) {   // rows [u_t, c_t, c_s] x n_img: the cond target rows take Q, K of the cond source rows behind them
      if (mode == 1 && b / n_img == 1) { bq = b + n_img; bk = b + n_img; }
    } else {
      const int bl = b + first_row, half = bl / (2 * n_img), role = (bl / n_img) & 1;
      if (mode == 1 && half == 1 && role == 1) { bq = b - n_img; bk = b - n_img; }
      if (mode == 2 && role == 1) { bk = b - n_img; bv = b - n_img; }
    }
  }
  const int query = blockIdx.x * 128 + wid * 32 + l31;
  const bool q_ok = query < N;
  float qreg[D / 2];
  {
    const float* qp = qkv + ((int64_t)bq * N + (q_ok ? query : N - 1)) * C3 + hd * D;
#pragma unroll
    for (int u = 0; u < D / 8; ++u) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qp + 8 * u + 4 * hh);
#pragma unroll
      for (int c = 0; c < 4; ++c) qreg[4 * u + c] = v[c] * scale_log2;
    }
  }
  for (int i = tid; i < KB * (VSTR - D); i += 256) sV[(i / (VSTR - D)) * VSTR + D + i % (VSTR - D)] = 0.f;
  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -3.0e38f, l_run = 0.f;
  for (int kb0 = 0; kb0 < N; kb0 += KB) {
    __syncthreads();
    for (int i = tid; i < KB * (D / 4); i += 256) {
      const int key = i / (D / 4), ch = i - key * (D / 4);
      f32x4 kv4 = {0.f, 0.f, 0.f, 0.f}, vv4 = {0.f, 0.f, 0.f, 0.f};
      if (kb0 + key < N) {
        kv4 = *reinterpret_cast<const f32x4*>(qkv + ((int64_t)bk * N + kb0 + key) * C3 + C + hd * D + ch * 4);
        vv4 = *reinterpret_cast<const f32x4*>(qkv + ((int64_t)bv * N + kb0 + key) * C3 + 2 * C + hd * D + ch * 4);
      }
      *reinterpret_cast<f32x4*>(sK + key * KSTR + ch * 4) = kv4;
      *reinterpret_cast<f32x4*>(sV + key * VSTR + ch * 4) = vv4;
    }
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < KB / 32; ++sb) {
      if (kb0 + sb * 32 >= N) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int u = 0; u < D / 8; ++u) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(sK + (sb * 32 + l31) * KSTR + 8 * u + 4 * hh);
#pragma unroll
        for (int c = 0; c < 4; ++c) s = mfma32(kf[c], qreg[4 * u + c], s);
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kb0 + sb * 32 + crow(r, hh) >= N) s[r] = -3.0e38f;
        mx = fmaxf(mx, s[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = exp2f(m_run - m_new);
      m_run = m_new;
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = exp2f(s[r] - m_new);
        ps += s[r];
      }
      l_run = l_run * alpha + ps;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vrow = sV + (sb * 32 + crow(r, hh)) * VSTR + l31;
#pragma unroll
        for (int t = 0; t < NT; ++t) o[t] = mfma32(vrow[t * 32], s[r], o[t]);
      }
    }
  }
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l;
  if (q_ok) {
    float* op = out + ((int64_t)b * N + query) * C + hd * D;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = t * 32 + 8 * g + 4 * hh;
        if (d0 < D) *reinterpret_cast<f32x4*>(op + d0) = (f32x4){o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv};
      }
  }
}

// ================================================================================================ cross-attention
// 77 text keys: one wave per query (lane = key for the scores, lane = channel for the output), K / V (and the source row's K for an edited
// row) staged in LDS once per block, fp32 FMAs -- 0.4 % of the UNet's FLOPs.  Semantics = cross_attn_kernel of attention.hip: roles from the
// batch layout, prompt-to-prompt Refine / Replace + Reweight + time blend on cond-target rows (source probabilities of the same query
// recomputed here), AttentionStore accumulation of the post-edit cond-half probabilities.
__global__ void __launch_bounds__(256) cross_attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ kv, float* __restrict__ out,
                                                             CrossParams p, int D) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int KSTR = D + 1, nctx = p.n_ctx;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, N = p.N, C = p.heads * D, C2 = 2 * C;
  int img = 0, role = -1, is_cond = 0;
  if (p.layout == 2) {
    const int bl = b + p.first_row;
    is_cond = bl / (2 * p.n_img);
    role = (bl / p.n_img) & 1;
    img = bl % p.n_img;
  } else if (p.layout == 1) {
    img = b % p.n_img;
    is_cond = (p.rows == p.n_img) ? 1 : (b / p.n_img);
    role = 0;
  } else if (p.layout == 3) {   // rows [u_t, c_t, c_s] x n_img (store only)
    const int g = b / p.n_img;
    img = b % p.n_img;
    is_cond = g >= 1;
    role = g == 1 ? 1 : 0;
  }
  const bool do_edit = p.edit && p.layout == 2 && is_cond && role == 1;
  const bool do_store = p.map_layer >= 0 && is_cond;
  const int bs = b - p.n_img;
  float* sK = smem_f;                        // [80][KSTR]
  float* sV = sK + 80 * KSTR;                // [80][D]
  float* sKs = sV + 80 * D;                  // [80][KSTR] source keys (edit)
  float* sQ = sKs + (do_edit ? 80 * KSTR : 0);   // [4 waves][2][D]  (own query, source query)
  float* sP = sQ + 4 * 2 * D;                // [4 waves][2][80]  (own probabilities, source probabilities)
  for (int i = tid; i < nctx * D; i += 256) {
    const int key = i / D, d = i - key * D;
    sK[key * KSTR + d] = kv[((int64_t)b * nctx + key) * C2 + h * D + d];
    sV[key * D + d] = kv[((int64_t)b * nctx + key) * C2 + C + h * D + d];
    if (do_edit) sKs[key * KSTR + d] = kv[((int64_t)bs * nctx + key) * C2 + h * D + d];
  }
  __syncthreads();
  float* wQ = sQ + wid * 2 * D;
  float* wP = sP + wid * 2 * 80;
  const int k0 = lane, k1 = lane + 64;
  // softmax(q . K^T * scale) of one (row, key set) for the wave's current query: lane holds keys k0 and k1
  auto probs = [&](const float* qv, const float* keys, float& p0, float& p1) {
    float s0 = 0.f, s1 = 0.f;
    const float* kr0 = keys + k0 * KSTR;
    const float* kr1 = keys + (k1 < nctx ? k1 : 0) * KSTR;
    for (int d = 0; d < D; ++d) {
      const float qd = qv[d];
      s0 = fmaf(qd, kr0[d], s0);
      s1 = fmaf(qd, kr1[d], s1);
    }
    s0 = k0 < nctx ? s0 * p.scale_log2 : -3.0e38f;
    s1 = k1 < nctx ? s1 * p.scale_log2 : -3.0e38f;
    const float mx = wave_max(fmaxf(s0, s1));
    const float e0 = exp2f(s0 - mx), e1 = exp2f(s1 - mx);
    const float inv = 1.0f / wave_sum(e0 + e1);
    p0 = e0 * inv;
    p1 = e1 * inv;
  };
  for (int query = blockIdx.x * 4 + wid; query < N; query += gridDim.x * 4) {
    for (int d = lane; d < D; d += 64) {
      wQ[d] = q[((int64_t)b * N + query) * C + h * D + d];
      if (do_edit) wQ[D + d] = q[((int64_t)bs * N + query) * C + h * D + d];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float p0, p1;
    probs(wQ, sK, p0, p1);
    if (do_edit) {
      float ps0, ps1;
      probs(wQ + D, sKs, ps0, ps1);
      wP[80 + k0] = ps0;
      if (k1 < 80) wP[80 + k1] = ps1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      auto edit = [&](int key, float tg) {
        float rep;
        if (p.replace_mat) {                                    // AttentionReplace: sum_w base[w] * M[w][key]
          const float* mrow = p.replace_mat + (int64_t)img * 77 * 77 + key;
          rep = 0.f;
          for (int w = 0; w < nctx; ++w) rep += wP[80 + w] * mrow[w * 77];
        } else {                                                // AttentionRefine
          int mp = p.mapper ? p.mapper[img * 77 + key] : 0;
          if (mp < 0) mp += nctx;                               // python negative index
          const float a = p.alphas ? p.alphas[img * 77 + key] : 0.f;
          rep = wP[80 + mp] * a + tg * (1.f - a);
        }
        rep *= p.equalizer ? p.equalizer[img * 77 + key] : 1.f;
        const float ca = p.cross_alpha[img * 77 + key];
        return rep * ca + (1.f - ca) * tg;
      };
      if (k0 < nctx) p0 = edit(k0, p0);
      if (k1 < nctx) p1 = edit(k1, p1);
    }
    if (do_store) {
      float* mp = p.maps_acc + (((((int64_t)p.map_layer * p.n_img_cap + img) * 2 + role) * p.heads + h) * N + query) * 77;
      if (k0 < nctx) mp[k0] += p0;
      if (k1 < nctx) mp[k1] += p1;
    }
    wP[k0] = p0;
    if (k1 < 80) wP[k1] = k1 < nctx ? p1 : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < D; d += 64) {
      float acc = 0.f;
      for (int key = 0; key < nctx; ++key) acc = fmaf(wP[key], sV[key * D + d], acc);
      out[((int64_t)b * N + query) * C + h * D + d] = acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// ================================================================================================ launchers
int launch_igemm_f32(const IGemmParams& p, hipStream_t s) {
  ETAINV_CHECK(p.a1 && p.w && p.out, "null pointer");
  ETAINV_CHECK(p.M > 0 && p.N > 0, "empty problem");
  ETAINV_CHECK(p.c1 % FBK == 0 && p.c2 % FBK == 0 && (p.c1 + p.c2) > 0, "fp32 path: channel counts must be multiples of 32");
  ETAINV_CHECK(p.taps == 1 || p.taps == 9, "taps must be 1 or 9");
  ETAINV_CHECK(p.taps == 9 || (p.stride == 1 && !p.ups), "1x1 / Linear: stride 1, no upsample");
  ETAINV_CHECK(!p.geglu || (p.N % 128) == 0, "GEGLU needs N % 128 == 0");
  ETAINV_CHECK(p.rows_per_batch > 0, "rows_per_batch");
  ETAINV_CHECK(!p.ln_stat && !p.stat_out && !p.w_batch_stride, "fp32 path: the LayerNorm / GroupNorm folds are 16-bit-path fusions (the engine runs the standalone norms)");
  ETAINV_CHECK(!p.out_nchw || (p.N == 4 && !p.geglu), "out_nchw needs N == 4");
  const double flops = 2.0 * (double)p.M * p.N * (double)(p.taps * (p.c1 + p.c2));
  ProfScope prof(PROF_IGEMM, flops, s, 4.0 * ((double)p.M * (p.c1 + p.c2) + (double)p.N * p.taps * (p.c1 + p.c2) + (double)p.M * p.N));
  hipLaunchKernelGGL(igemm_f32_kernel, dim3(cdiv(p.M, FBM), cdiv(p.N, FBN)), dim3(256), 0, s, p);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_groupnorm_f32(const void* x1, const void* x2, int c1, int c2, const float* gamma, const float* beta, void* out, int b, int hw, int groups,
                         float eps, int silu, float* scratch, hipStream_t s) {
  ETAINV_CHECK(x1 && gamma && beta && out && scratch && (c1 + c2) % groups == 0, "bad arguments");
  ProfScope prof(PROF_GROUPNORM, 2.0 * 4.0 * (double)b * hw * (c1 + c2), s);
  hipLaunchKernelGGL(gn_stats_f32_kernel, dim3(groups, b), dim3(256), 0, s, (const float*)x1, (const float*)x2, c1, c2, hw, groups, eps, scratch);
  const int64_t total = (int64_t)b * hw * (c1 + c2);
  hipLaunchKernelGGL(gn_apply_f32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)x1, (const float*)x2, c1, c2, hw, groups, gamma, beta,
                     scratch, silu, (float*)out, total);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_layernorm_f32(const void* x, const float* gamma, const float* beta, void* out, int rows, int c, float eps, hipStream_t s) {
  ETAINV_CHECK(x && gamma && beta && out && c <= 64 * 20, "bad arguments");
  ProfScope prof(PROF_LAYERNORM, 2.0 * 4.0 * (double)rows * c, s);
  hipLaunchKernelGGL(layernorm_f32_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, (const float*)x, gamma, beta, (float*)out, rows, c, eps);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

template <int D>
static int launch_self_f32_t(const void* qkv, void* out, int b, int n, int heads, int mode, int n_img, hipStream_t s, int first_row) {
  constexpr int NT = (D + 31) / 32;
  const size_t lds = (size_t)64 * ((D + 4) + (NT * 32 + 8)) * sizeof(float);
  static bool attr[kMaxDevices] = {};
  const int dev = current_device();
  if (!attr[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attn_f32_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr[dev] = true;
  }
  ProfScope prof(PROF_SELF_ATTN, 4.0 * (double)b * heads * (double)n * (double)n * D, s);
  hipLaunchKernelGGL(self_attn_f32_kernel<D>, dim3(cdiv(n, 128), heads, b), dim3(256), lds, s, (const float*)qkv, (float*)out, n, heads,
                     (1.0f / sqrtf((float)D)) * 1.4426950408889634f, mode, n_img, first_row);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_self_attention_f32(const void* qkv, void* out, int b, int n, int heads, int d, int mode, int n_img, hipStream_t s, int first_row) {
  switch (d) {
    case 40: return launch_self_f32_t<40>(qkv, out, b, n, heads, mode, n_img, s, first_row);
    case 80: return launch_self_f32_t<80>(qkv, out, b, n, heads, mode, n_img, s, first_row);
    case 160: return launch_self_f32_t<160>(qkv, out, b, n, heads, mode, n_img, s, first_row);
    default: ETAINV_FAIL("head_dim must be 40, 80 or 160");
  }
}

int launch_cross_attention_f32(const void* q, const void* kv, void* out, int b, int d, const CrossParams& p, hipStream_t s) {
  ETAINV_CHECK(p.n_ctx >= 1 && p.n_ctx <= 77 && d >= 8 && d <= 160, "fp32 cross-attention: up to 77 keys, head_dim <= 160");
  const size_t lds = (size_t)(80 * (d + 1) * 2 + 80 * d + 4 * 2 * d + 4 * 2 * 80) * sizeof(float);
  static bool attr[kMaxDevices] = {};
  const int dev = current_device();
  if (!attr[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr[dev] = true;
  }
  ProfScope prof(PROF_CROSS_ATTN, 4.0 * (double)b * p.heads * (double)p.N * (double)p.n_ctx * d, s);
  const int gx = std::max(1, std::min(cdiv(p.N, 4), cdiv(4096, b * p.heads)));
  hipLaunchKernelGGL(cross_attn_f32_kernel, dim3(gx, p.heads, b), dim3(256), lds, s, (const float*)q, (const float*)kv, (float*)out, p, d);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

}  // namespace etainv
