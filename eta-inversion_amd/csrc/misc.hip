// Small kernels around the UNet body: conv_in / conv_out (the only convolutions whose channel count is not
// a multiple of 64; 0.03 % of the FLOPs), sinusoidal timestep embedding, SiLU, weight packing, dtype casts.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace etainv {

template <typename TIO>
__device__ __forceinline__ float ld_io(const void* p, int64_t i) { return to_f32(reinterpret_cast<const TIO*>(p)[i]); }

// conv_in: x NCHW [n_lat][4][L][L] (io type) -> out NHWC [rows][L*L][cout] (T); UNet row r reads latent r % n_lat.
// w: fp32 [36 = (ky*3+kx)*4+ci][cout]; block (cout/8, PIX) threads, each thread 8 output channels of one pixel.
template <typename T, typename TIO>
__global__ void conv_in_kernel(const TIO* __restrict__ x, int n_lat, int L, const float* __restrict__ w, const float* __restrict__ bias,
                               int cout, T* __restrict__ out, int pix_per_block) {
  extern __shared__ float sw[];  // [36][cout]
  const int nthr = blockDim.x * blockDim.y, tid = threadIdx.y * blockDim.x + threadIdx.x;
  for (int i = tid; i < 36 * cout; i += nthr) sw[i] = w[i];
  __syncthreads();
  const int row = blockIdx.y, lat = row % n_lat;
  const int LL = L * L;
  const TIO* xb = x + (int64_t)lat * 4 * LL;
  const int co = threadIdx.x * 8;
  for (int it = threadIdx.y; it < pix_per_block; it += blockDim.y) {
    const int pix = blockIdx.x * pix_per_block + it;
    if (pix >= LL) break;
    const int oy = pix / L, ox = pix - oy * L;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias[co + j];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        if (iy < 0 || iy >= L || ix < 0 || ix >= L) continue;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          const float v = to_f32(xb[(int64_t)ci * LL + iy * L + ix]);
          const float* wr = sw + ((ky * 3 + kx) * 4 + ci) * cout + co;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += v * wr[j];
        }
      }
    T o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(acc[j]);
    *reinterpret_cast<u32x4*>(out + ((int64_t)row * LL + pix) * cout + co) = *reinterpret_cast<u32x4*>(o);
  }
}

// conv_out: x NHWC [rows][L*L][cin] (T) -> out NCHW [rows][4][L][L] (io type).  w: fp32 [9][cin][4].
// one wave per output pixel; lanes stride over (tap, 8-channel vector) pairs, then a wave reduction.
template <typename T, typename TIO>
__global__ void __launch_bounds__(256) conv_out_kernel(const T* __restrict__ x, int L, int cin, const float* __restrict__ w,
                                                       const float* __restrict__ bias, TIO* __restrict__ out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int LL = L * L;
  const int pix = blockIdx.x * 4 + wid;
  const int row = blockIdx.y;
  if (pix >= LL) return;
  const int oy = pix / L, ox = pix - oy * L;
  const int nvec = cin >> 3;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int idx = lane; idx < 9 * nvec; idx += 64) {
    const int tap = idx / nvec, v = idx - tap * nvec;
    const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
    if (iy < 0 || iy >= L || ix < 0 || ix >= L) continue;
    u32x4 raw = *reinterpret_cast<const u32x4*>(x + ((int64_t)row * LL + iy * L + ix) * cin + v * 8);
    const T* e = reinterpret_cast<const T*>(&raw);
    const float* wr = w + ((int64_t)tap * cin + v * 8) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = to_f32(e[j]);
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] += f * wr[j * 4 + o];
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[o] = wave_sum(acc[o]);
  if (lane < 4) out[((int64_t)row * 4 + lane) * LL + pix] = from_f32<TIO>(acc[lane] + bias[lane]);
}

// one thread per (row, pixel): gathers the 3x3x4 patch (zero padded) into 64 contiguous K elements
template <typename T, typename TIO>
__global__ void im2col_in_kernel(const TIO* __restrict__ x, int n_lat, int L, int rows, T* __restrict__ out) {
  const int LL = L * L;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * LL) return;
  const int row = (int)(i / LL), pix = (int)(i - (int64_t)row * LL);
  const int oy = pix / L, ox = pix - oy * L;
  const TIO* xb = x + (int64_t)(row % n_lat) * 4 * LL;
  T v[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) v[k] = (T)0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
    if (iy >= 0 && iy < L && ix >= 0 && ix < L) {
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) v[t * 4 + ci] = from_f32<T>(to_f32(xb[(int64_t)ci * LL + iy * L + ix]));
    }
  }
  u32x4* o = reinterpret_cast<u32x4*>(out + i * 64);
#pragma unroll
  for (int q = 0; q < (int)(64 * sizeof(T) / 16); ++q) o[q] = reinterpret_cast<u32x4*>(v)[q];
}

// The timesteps travel BY VALUE in the kernel arguments (one scalar when every row shares it -- always the case in the loops -- else
// chunks of 64): no host staging buffer that a later call could overwrite before an asynchronous copy has read it.
struct TimeVec { float t[64]; };
template <typename T>
__global__ void time_embedding_kernel(TimeVec tv, int uniform, int row0, int rows, int dim, T* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * dim) return;
  const int r = i / dim, c = i - r * dim, half = dim / 2;
  const int k = c < half ? c : c - half;
  const float freq = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
  const float a = tv.t[uniform ? 0 : r] * freq;
  out[(int64_t)row0 * dim + i] = from_f32<T>(c < half ? cosf(a) : sinf(a));
}

template <typename T>
__global__ void silu_kernel(const T* x, T* out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = from_f32<T>(silu_f(to_f32(x[i])));
}

// weight packing (fp32 diffusers layout -> engine layout), one thread per destination element
//   mode 0: [rows][cols] copy/cast
//   mode 1: conv OIHW [rows=O][cols=I*taps] -> [O][tap][I]
//   mode 2: GEGLU row interleave of a [rows=8c][cols] matrix: physical row p <- logical row
//           (p%64 < 32 ? (p/64)*32 + p%64 : rows/2 + (p/64)*32 + p%64 - 32)
//   mode 3: conv_in  [O][4][3][3] -> fp32-style [tap*4+ci][O]            (rows = O, cols = 36)
//   mode 4: conv_out [4][I][3][3] -> [tap][I][4]                         (rows = 4, cols = I*9)
//   mode 5: conv_in as a K = 64 GEMM: [O][4][3][3] -> [O][64], k = tap*4 + ci, k >= 36 zero   (rows = O, cols = 36)
template <typename TD>
__global__ void pack_weight_kernel(const float* __restrict__ src, TD* __restrict__ dst, int64_t rows, int64_t cols, int mode, int taps, float scale,
                                   const float* __restrict__ colscale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (mode == 5) {
    if (i >= rows * 64) return;
    const int64_t o = i / 64, k = i - o * 64, tap = k / 4, ci = k - tap * 4;
    dst[i] = from_f32<TD>(k < 36 ? src[o * 36 + ci * 9 + tap] : 0.f);
    return;
  }
  if (i >= rows * cols) return;
  int64_t s = i;
  if (mode == 1) {
    const int64_t o = i / cols, r = i - o * cols, cin = cols / taps;
    const int64_t t = r / cin, ci = r - t * cin;
    s = o * cols + ci * taps + t;
  } else if (mode == 2) {
    const int64_t p = i / cols, c = i - p * cols;
    const int64_t blk = p / 64, within = p % 64;
    const int64_t logical = within < 32 ? blk * 32 + within : rows / 2 + blk * 32 + within - 32;
    s = logical * cols + c;
  } else if (mode == 3) {
    const int64_t k = i / rows, o = i - k * rows;  // dst [36][O]
    const int64_t tap = k / 4, ci = k - tap * 4;
    s = o * 36 + ci * 9 + tap;
  } else if (mode == 4) {
    const int64_t cin = cols / 9;                  // dst [9][cin][4]
    const int64_t tap = i / (cin * 4), r = i - tap * cin * 4, ci = r / 4, o = r - ci * 4;
    s = o * cols + ci * 9 + tap;
  }
  // scale: 1, or a constant folded into the weights in fp32 BEFORE the rounding (softmax scale of to_q); colscale (modes 0 / 2): a per-input-
  // column factor (the gamma of a LayerNorm folded into this Linear)
  float v = src[s] * scale;
  if (colscale) v *= colscale[s % cols];
  dst[i] = from_f32<TD>(v);
}

// LayerNorm folded into the Linear that consumes it (reference: BasicTransformerBlock norm1 -> attn1.to_q/k/v, norm2 -> attn2.to_q,
// norm3 -> ff.net.0.proj):   LN(x) W^T + b  =  rstd * (x W'^T - mean * s) + c   with  W'[n][k] = gamma[k] W[n][k]  (packed by
// pack_weight_kernel with colscale = gamma),  s[n] = sum_k W'[n][k] over the ROUNDED operand the MFMAs multiply,  c[n] = sum_k beta[k] W[n][k] + b[n].
// One wave per packed row; `mode` 0 / 2 as in pack_weight_kernel (2: GEGLU row interleave), bias already in packed order.
template <typename TD>
__global__ void __launch_bounds__(256) ln_fold_vectors_kernel(const float* __restrict__ src, const TD* __restrict__ packed, const float* __restrict__ beta,
                                                              const float* __restrict__ bias_packed, int64_t rows, int64_t cols, int mode, float scale,
                                                              float* __restrict__ s_out, float* __restrict__ c_out) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= rows) return;
  int64_t logical = p;
  if (mode == 2) {
    const int64_t blk = p / 64, within = p % 64;
    logical = within < 32 ? blk * 32 + within : rows / 2 + blk * 32 + within - 32;
  }
  float s = 0.f, c = 0.f;
  for (int64_t k = lane; k < cols; k += 64) {
    s += to_f32(packed[p * cols + k]);
    c += beta[k] * (src[logical * cols + k] * scale);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    c += __shfl_xor(c, o);
  }
  if (lane == 0) {
    s_out[p] = s;
    c_out[p] = c + (bias_packed ? bias_packed[p] : 0.f);
  }
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = from_f32<TD>(to_f32(src[i]));
}

int launch_conv_in(const void* latent, int io_dtype, int n_lat, int rows, int L, const void* w, const float* bias, int cout,
                   void* out, int dtype, hipStream_t s) {
  ETAINV_CHECK(latent && w && bias && out && n_lat > 0 && rows > 0 && cout % 8 == 0, "bad arguments");
  const int ppb = 48;
  dim3 block(cout / 8, std::max(1, 256 / (cout / 8)));
  dim3 grid(cdiv(L * L, ppb), rows);
  const size_t lds = (size_t)36 * cout * sizeof(float);
  ETAINV_DISPATCH_HALF(dtype, T, ETAINV_DISPATCH_DTYPE(io_dtype, TIO,
      hipLaunchKernelGGL((conv_in_kernel<T, TIO>), grid, block, lds, s, (const TIO*)latent, n_lat, L, (const float*)w, bias, cout, (T*)out, ppb)));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_im2col_in(const void* latent, int io_dtype, int n_lat, int rows, int L, void* out, int dtype, hipStream_t s) {
  ETAINV_CHECK(latent && out && n_lat > 0 && rows > 0, "bad arguments");
  const int64_t n = (int64_t)rows * L * L;
  ETAINV_DISPATCH_DTYPE(dtype, T, ETAINV_DISPATCH_DTYPE(io_dtype, TIO,
      hipLaunchKernelGGL((im2col_in_kernel<T, TIO>), dim3(cdiv(n, 256)), dim3(256), 0, s, (const TIO*)latent, n_lat, L, rows, (T*)out)));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_conv_out(const void* x, int rows, int L, int cin, const void* w, const float* bias, void* out, int io_dtype, int dtype,
                    hipStream_t s) {
  ETAINV_CHECK(x && w && bias && out && cin % 8 == 0, "bad arguments");
  dim3 grid(cdiv(L * L, 4), rows);
  ETAINV_DISPATCH_HALF(dtype, T, ETAINV_DISPATCH_DTYPE(io_dtype, TIO,
      hipLaunchKernelGGL((conv_out_kernel<T, TIO>), grid, dim3(256), 0, s, (const T*)x, L, cin, (const float*)w, bias, (TIO*)out)));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_time_embedding(const int64_t* t_host, int rows, int dim, void* out, int dtype, hipStream_t s) {
  ETAINV_CHECK(t_host && out && dim % 2 == 0 && rows > 0, "bad arguments");
  bool uniform = true;
  for (int i = 1; i < rows; ++i) uniform = uniform && t_host[i] == t_host[0];
  TimeVec tv;
  if (uniform) {
    tv.t[0] = (float)t_host[0];
    ETAINV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(time_embedding_kernel<T>, dim3(cdiv(rows * dim, 256)), dim3(256), 0, s, tv, 1, 0, rows, dim, (T*)out));
  } else {
    for (int r0 = 0; r0 < rows; r0 += 64) {
      const int n = std::min(64, rows - r0);
      for (int i = 0; i < n; ++i) tv.t[i] = (float)t_host[r0 + i];
      ETAINV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(time_embedding_kernel<T>, dim3(cdiv(n * dim, 256)), dim3(256), 0, s, tv, 0, r0, n, dim, (T*)out));
    }
  }
  ETAINV_LAUNCH_CHECK();
  return 0;
}

// the same embedding with the timesteps read from DEVICE memory (t_dev [rows] floats): the form a captured hipGraph can replay with new timesteps.
// launch_set_timesteps (kernel arguments by value, like above) fills t_dev in front of it.
template <typename T>
__global__ void time_embedding_dev_kernel(const float* __restrict__ t_dev, int rows, int dim, T* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * dim) return;
  const int r = i / dim, c = i - r * dim, half = dim / 2;
  const int k = c < half ? c : c - half;
  const float freq = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
  const float a = t_dev[r] * freq;
  out[i] = from_f32<T>(c < half ? cosf(a) : sinf(a));
}
__global__ void set_timesteps_kernel(TimeVec tv, int row0, int n, float* __restrict__ t_dev) {
  const int i = threadIdx.x;
  if (i < n) t_dev[row0 + i] = tv.t[i];
}
int launch_set_timesteps(const int64_t* t_host, int rows, float* t_dev, hipStream_t s) {
  ETAINV_CHECK(t_host && t_dev && rows > 0, "bad arguments");
  for (int r0 = 0; r0 < rows; r0 += 64) {
    const int n = std::min(64, rows - r0);
    TimeVec tv;
    for (int i = 0; i < n; ++i) tv.t[i] = (float)t_host[r0 + i];
    hipLaunchKernelGGL(set_timesteps_kernel, dim3(1), dim3(64), 0, s, tv, r0, n, t_dev);
  }
  ETAINV_LAUNCH_CHECK();
  return 0;
}
int launch_time_embedding_dev(const float* t_dev, int rows, int dim, void* out, int dtype, hipStream_t s) {
  ETAINV_CHECK(t_dev && out && dim % 2 == 0 && rows > 0, "bad arguments");
  ETAINV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(time_embedding_dev_kernel<T>, dim3(cdiv(rows * dim, 256)), dim3(256), 0, s, t_dev, rows, dim, (T*)out));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_silu(const void* x, void* out, int64_t n, int dtype, hipStream_t s) {
  ETAINV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(silu_kernel<T>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const T*)x, (T*)out, n));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

// conv3x3 behind a nearest-2x upsample == four 2x2 convs on the SOURCE image, one per output phase (py, px) = (y & 1, x & 1): the taps of a 3x3 row / column
// that land on the same source pixel are summed.  Phase py = 0 groups kernel rows {0}, {1, 2}; py = 1 groups {0, 1}, {2} (same for columns).
// src [cout][cin][3][3] fp32 -> dst [phase = 2 py + px][cout][tap = 2 ty + tx][cin], summed in fp32, rounded once
template <typename TD>
__global__ void pack_ups4_kernel(const float* __restrict__ src, TD* __restrict__ dst, int cout, int cin) {
  const int64_t total = (int64_t)16 * cout * cin;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cin);
  const int t = (int)((i / cin) & 3);
  const int n = (int)((i / ((int64_t)4 * cin)) % cout);
  const int ph = (int)(i / ((int64_t)4 * cin * cout));
  const int py = ph >> 1, px = ph & 1, ty = t >> 1, tx = t & 1;
  const int ky0 = py == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = py == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
  const int kx0 = px == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = px == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
  const float* w = src + ((int64_t)n * cin + c) * 9;
  float v = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) v += w[ky * 3 + kx];
  dst[i] = from_f32<TD>(v);
}
int launch_pack_ups4(const float* src, void* dst, int cout, int cin, int dtype, hipStream_t s) {
  ETAINV_CHECK(src && dst && cout > 0 && cin > 0, "bad arguments");
  ETAINV_DISPATCH_DTYPE(dtype, TD, hipLaunchKernelGGL(pack_ups4_kernel<TD>, dim3((unsigned)cdiv((int64_t)16 * cout * cin, (int64_t)256)), dim3(256), 0, s, src, (TD*)dst, cout, cin));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_pack_weight(const float* src, void* dst, int64_t rows, int64_t cols, int mode, int taps, int dtype, hipStream_t s, float scale,
                       const float* colscale) {
  ETAINV_CHECK(src && dst && rows > 0 && cols > 0, "bad arguments");
  ETAINV_CHECK(!colscale || mode == 0 || mode == 2, "column scale: plain / GEGLU packing only");
  const int64_t n = mode == 5 ? rows * 64 : rows * cols;
  ETAINV_DISPATCH_DTYPE(dtype, TD, hipLaunchKernelGGL(pack_weight_kernel<TD>, dim3(cdiv(n, 256)), dim3(256), 0, s, src, (TD*)dst, rows, cols, mode, taps, scale,
                                                      colscale));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

// GroupNorm (no activation) folded into the 1x1 conv behind it (Transformer2DModel.norm -> proj_in): per image b the conv sees
// x * a_b + sh_b with a_b[k] = rstd[b][g(k)] * gamma[k], sh_b[k] = beta[k] - mean[b][g(k)] * a_b[k], so W_b = W . diag(a_b) and c_b = W sh_b + bias.
// One wave per (image, output row n).
template <typename TD>
__global__ void __launch_bounds__(256) gn_fold_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bias, const float* __restrict__ stats, int groups, int n_out, int k_in,
                                                      TD* __restrict__ wb, float* __restrict__ cb) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (n >= n_out) return;
  const int cpg = k_in / groups;
  const float* st = stats + (int64_t)b * groups * 2;
  float c = 0.f;
  for (int k = lane; k < k_in; k += 64) {
    const int g = k / cpg;
    const float a = st[g * 2 + 1] * gamma[k];
    const float wv = w[(int64_t)n * k_in + k];
    const TD wr = from_f32<TD>(wv * a);
    wb[((int64_t)b * n_out + n) * k_in + k] = wr;
    // the mean term against the ROUNDED operand the MFMAs multiply: a group's mean then cancels exactly, whatever its size
    c += beta[k] * wv - st[g * 2] * to_f32(wr);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) cb[(int64_t)b * n_out + n] = c + (bias ? bias[n] : 0.f);
}

int launch_gn_fold(const float* w, const float* gamma, const float* beta, const float* bias, const float* final_stats, int groups, int b, int n, int k,
                   void* wb_out, float* cb_out, int dtype, hipStream_t s) {
  ETAINV_CHECK(w && gamma && beta && final_stats && wb_out && cb_out && b > 0 && n > 0 && k > 0 && k % groups == 0, "bad arguments");
  ProfScope prof(PROF_GROUPNORM, 0.0, s);
  ETAINV_DISPATCH_HALF(dtype, TD, hipLaunchKernelGGL(gn_fold_kernel<TD>, dim3(cdiv(n, 4), b), dim3(256), 0, s, w, gamma, beta, bias, final_stats, groups, n, k,
                                                     (TD*)wb_out, cb_out));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_ln_fold(const float* w_src, const float* gamma, const float* beta, const float* bias_packed, int64_t rows, int64_t cols, int mode, float scale,
                   void* w_dst, float* s_dst, float* c_dst, int dtype, hipStream_t s) {
  ETAINV_CHECK(w_src && gamma && beta && w_dst && s_dst && c_dst && rows > 0 && cols > 0, "bad arguments");
  ETAINV_CHECK(mode == 0 || mode == 2, "plain / GEGLU packing only");
  if (launch_pack_weight(w_src, w_dst, rows, cols, mode, 1, dtype, s, scale, gamma)) return 1;
  ETAINV_DISPATCH_HALF(dtype, TD, hipLaunchKernelGGL(ln_fold_vectors_kernel<TD>, dim3(cdiv(rows, 4)), dim3(256), 0, s, w_src, (const TD*)w_dst, beta, bias_packed,
                                                     rows, cols, mode, scale, s_dst, c_dst));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_cast_f32(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, hipStream_t s) {
  ETAINV_DISPATCH_DTYPE(src_dtype, TS, ETAINV_DISPATCH_DTYPE(dst_dtype, TD,
      hipLaunchKernelGGL((cast_kernel<TS, TD>), dim3(cdiv(n, 256)), dim3(256), 0, s, (const TS*)src, (TD*)dst, n)));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

}  // namespace etainv
