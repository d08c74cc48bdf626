// Kernels used only by the VAE and the CLIP text encoder (outside the DDIM loop, ~1.5 % of the FLOPs per image):
// generic 3x3 im2col for 3/4-channel NCHW inputs (with an optional fused 1x1 pre-mix = post_quant_conv), row softmax for
// the VAE's single-head d=512 attention (scores are materialised once per image: 32 MB at 64x64), quick_gelu, token +
// position embedding, and the 77-token causal attention of the text encoder.
#include "common.h"
#include "kernels.h"

namespace etainv {

// out [rows*H*W][64]: k = tap*cin + ci (cin <= 4), zero for k >= 9*cin and for halo taps.  premix: optional [cin][cin+1]
// (matrix | bias) applied to the valid pixels before the gather (post_quant_conv followed by a zero-padded 3x3 conv).
template <typename T, typename TIO>
__global__ void im2col_small_kernel(const TIO* __restrict__ x, int cin, int H, int W, int rows, const float* __restrict__ premix,
                                    T* __restrict__ out) {
  const int HW = H * W;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * HW) return;
  const int row = (int)(i / HW), pix = (int)(i - (int64_t)row * HW);
  const int oy = pix / W, ox = pix - oy * W;
  const TIO* xb = x + (int64_t)row * cin * HW;
  T v[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) v[k] = (T)0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      float in[4] = {0.f, 0.f, 0.f, 0.f};
      for (int ci = 0; ci < cin; ++ci) in[ci] = to_f32(xb[(int64_t)ci * HW + iy * W + ix]);
      for (int co = 0; co < cin; ++co) {
        float a = in[co];
        if (premix) {
          a = premix[co * (cin + 1) + cin];
          for (int ci = 0; ci < cin; ++ci) a += premix[co * (cin + 1) + ci] * in[ci];
        }
        v[t * cin + co] = from_f32<T>(a);
      }
    }
  }
  u32x4* o = reinterpret_cast<u32x4*>(out + i * 64);
#pragma unroll
  for (int q = 0; q < (int)(64 * sizeof(T) / 16); ++q) o[q] = reinterpret_cast<u32x4*>(v)[q];
}

// in place: x[row][:] = softmax(scale * x[row][:]); one 256-thread block per row, three passes over the (L2-resident) row
template <typename T>
__global__ void __launch_bounds__(256) row_softmax_kernel(T* __restrict__ x, int n, float scale_log2) {
  T* r = x + (int64_t)blockIdx.x * n;
  __shared__ float red[4];
  float mx = -3.0e38f;
  for (int c = threadIdx.x; c < n; c += 256) mx = fmaxf(mx, to_f32(r[c]) * scale_log2);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = threadIdx.x; c < n; c += 256) s += __builtin_amdgcn_exp2f(to_f32(r[c]) * scale_log2 - mx);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = threadIdx.x; c < n; c += 256) r[c] = from_f32<T>(__builtin_amdgcn_exp2f(to_f32(r[c]) * scale_log2 - mx) * inv);
}

template <typename T>
__global__ void quick_gelu_kernel(const T* x, T* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float v = to_f32(x[i]);
    out[i] = from_f32<T>(v / (1.f + __expf(-1.702f * v)));
  }
}

// out[b][p][:] = tok[ids[b][p]][:] + pos[p][:]
template <typename T>
__global__ void embed_kernel(const int64_t* __restrict__ ids, const T* __restrict__ tok, const T* __restrict__ pos, int n_pos, int d,
                             T* __restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t rp = i / d;
  const int c = (int)(i - rp * d), p = (int)(rp % n_pos);
  out[i] = from_f32<T>(to_f32(tok[ids[rp] * d + c]) + to_f32(pos[(int64_t)p * d + c]));
}

// causal multi-head attention for short sequences (n <= 80, d = 64): qkv [b][n][3*heads*64] -> out [b][n][heads*64];
// grid (heads, b), 128 threads: thread i < n owns query i (fp32 throughout).
template <typename T>
__global__ void __launch_bounds__(128) causal_attn_small_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n, int heads) {
  constexpr int D = 64, NMAX = 80;
  __shared__ float sk[NMAX][D + 1], sv[NMAX][D + 1];
  const int h = blockIdx.x, b = blockIdx.y, C = heads * D;
  for (int idx = threadIdx.x; idx < n * D; idx += blockDim.x) {
    const int j = idx / D, c = idx - j * D;
    const T* base = qkv + ((int64_t)b * n + j) * 3 * C + h * D + c;
    sk[j][c] = to_f32(base[C]);
    sv[j][c] = to_f32(base[2 * C]);
  }
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= n) return;
  float q[D], o[D];
  const T* qp = qkv + ((int64_t)b * n + i) * 3 * C + h * D;
#pragma unroll
  for (int c = 0; c < D; ++c) { q[c] = to_f32(qp[c]) * 0.125f; o[c] = 0.f; }
  float m = -3.0e38f, l = 0.f;
  for (int j = 0; j <= i; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) s += q[c] * sk[j][c];
    const float mn = fmaxf(m, s), a = __expf(m - mn), pj = __expf(s - mn);
    l = l * a + pj;
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] = o[c] * a + pj * sv[j][c];
    m = mn;
  }
  T* op = out + ((int64_t)b * n + i) * C + h * D;
  const float inv = 1.f / l;
#pragma unroll
  for (int c = 0; c < D; ++c) op[c] = from_f32<T>(o[c] * inv);
}

}  // namespace etainv

using namespace etainv;

extern "C" int etainv_op_im2col3x3(const void* x_nchw, int io_dtype, int cin, int h, int w, int rows, const float* premix, void* out,
                                   int dtype, void* stream) {
  ETAINV_CHECK(x_nchw && out && cin >= 1 && cin <= 4 && rows >= 1, "bad arguments");
  const int64_t n = (int64_t)rows * h * w;
  ETAINV_DISPATCH_DTYPE(dtype, T, ETAINV_DISPATCH_DTYPE(io_dtype, TIO,
      hipLaunchKernelGGL((im2col_small_kernel<T, TIO>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const TIO*)x_nchw, cin, h, w,
                         rows, premix, (T*)out)));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

extern "C" int etainv_op_row_softmax(void* x, int rows, int n, float scale, int dtype, void* stream) {
  ETAINV_CHECK(x && rows >= 1 && n >= 1, "bad arguments");
  ETAINV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(row_softmax_kernel<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (T*)x, n,
                                                    scale * 1.4426950408889634f));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

extern "C" int etainv_op_quick_gelu(const void* x, void* out, int64_t n, int dtype, void* stream) {
  ETAINV_CHECK(x && out && n >= 0, "bad arguments");
  if (n == 0) return 0;
  ETAINV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(quick_gelu_kernel<T>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                                                    (const T*)x, (T*)out, n));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

extern "C" int etainv_op_embed(const int64_t* ids, const void* tok, const void* pos, int b, int n_pos, int d, void* out, int dtype,
                               void* stream) {
  ETAINV_CHECK(ids && tok && pos && out && b >= 1, "bad arguments");
  const int64_t total = (int64_t)b * n_pos * d;
  ETAINV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(embed_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, ids,
                                                    (const T*)tok, (const T*)pos, n_pos, d, (T*)out, total));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

extern "C" int etainv_op_causal_attention(const void* qkv, void* out, int b, int n, int heads, int d, int dtype, void* stream) {
  ETAINV_CHECK(qkv && out && b >= 1 && n >= 1 && n <= 80 && d == 64, "causal attention supports n <= 80, head_dim 64");
  ETAINV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(causal_attn_small_kernel<T>, dim3(heads, b), dim3(128), 0, (hipStream_t)stream,
                                                    (const T*)qkv, (T*)out, n, heads));
  ETAINV_LAUNCH_CHECK();
  return 0;
}
