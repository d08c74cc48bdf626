// Elementwise / reduction kernels of the DDIM loop (HBM-bound; a few hundred KB per call).
//   cfg_combine        eta_inversion.py:328
//   ddim_step          scheduling_ddim_inverse.py:94-98
//   eta_backward_step  eta_inversion.py:207-273, 296-317, 330-375 fused (2 launches, no host sync)
#include "common.h"

namespace etainv {

template <typename T>
__global__ void cfg_combine_kernel(const T* __restrict__ u, const T* __restrict__ c, float g, T* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float a = to_f32(u[i]), b = to_f32(c[i]);
    out[i] = from_f32<T>(a + g * (b - a));
  }
}

template <typename T>
__global__ void ddim_step_kernel(const T* __restrict__ x, const T* __restrict__ eps, float sqrt_1m_from, float inv_sqrt_from,
                                 float sqrt_to, float sqrt_1m_to, T* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float xv = to_f32(x[i]), e = to_f32(eps[i]);
    float x0 = (xv - sqrt_1m_from * e) * inv_sqrt_from;
    out[i] = from_f32<T>(sqrt_to * x0 + sqrt_1m_to * e);
  }
}

constexpr int ETA_MAX_CAND = 16;
constexpr int ETA_LOSS_BLOCKS = 64;  // partial sums per (image, candidate)

struct EtaCoef {
  float g, eta, sa_t, s1m_t, sa_p, a_p, var, thres;
  float tdir;   // target_dirinv weight (0 = off): the target row takes tdir * dirinv_map * (x_prev_src - x_src_new)
};

// pass 1: guided source noise, DDIM-eta mean (noise = 0), ideal noise z*, squared distance of each candidate
template <typename T>
__global__ void __launch_bounds__(256) eta_loss_kernel(const T* __restrict__ x, const T* __restrict__ eps_all,
                                                       const T* __restrict__ x_prev, const T* __restrict__ noise,
                                                       int n_cand, EtaCoef k, int n_img, int chw, float* __restrict__ partial) {
  const int img = blockIdx.y;
  const int per = (chw + gridDim.x - 1) / gridDim.x;
  const int beg = blockIdx.x * per, end = min(chw, beg + per);
  float acc[ETA_MAX_CAND];
#pragma unroll
  for (int j = 0; j < ETA_MAX_CAND; ++j) acc[j] = 0.f;
  const float std_t = k.eta * sqrtf(k.var);
  const float dir_c = sqrtf(1.f - k.a_p - std_t * std_t);
  const T* xs = x + (int64_t)img * chw;
  const T* eu = eps_all + (int64_t)img * chw;
  const T* ec = eps_all + (int64_t)(2 * n_img + img) * chw;
  const T* xp = x_prev + (int64_t)img * chw;
  for (int e = beg + threadIdx.x; e < end; e += blockDim.x) {
    float u = to_f32(eu[e]), c = to_f32(ec[e]);
    float eps = u + k.g * (c - u);
    float x0 = (to_f32(xs[e]) - k.s1m_t * eps) / k.sa_t;
    float mean = k.sa_p * x0 + dir_c * eps;
    float opt = (to_f32(xp[e]) - mean) / std_t;  // eta == 0 -> inf / NaN exactly like the reference (SURVEY E-7)
#pragma unroll
    for (int j = 0; j < ETA_MAX_CAND; ++j) {
      if (j < n_cand) {
        float d = to_f32(noise[(int64_t)j * chw + e]) - opt;
        acc[j] += d * d;
      }
    }
  }
  __shared__ float red[4][ETA_MAX_CAND];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < ETA_MAX_CAND; ++j) {
    float v = wave_sum(acc[j]);
    if (lane == 0) red[wid][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < n_cand) {
    float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    partial[((int64_t)img * ETA_MAX_CAND + threadIdx.x) * ETA_LOSS_BLOCKS + blockIdx.x] = v;
  }
}

// pass 2: argmin over candidates (torch.argmin semantics: first NaN wins, else first minimum), then the
// per-pixel masked DDIM-eta update of the (src, tgt) rows and the source replay.
template <typename T>
__global__ void __launch_bounds__(256) eta_update_kernel(const T* __restrict__ x, const T* __restrict__ eps_all,
                                                         const T* __restrict__ x_prev, const T* __restrict__ noise,
                                                         int n_cand, EtaCoef k, const T* __restrict__ mask_map, int use_mask,
                                                         const T* __restrict__ dirinv_map, int n_img, int chw, int hw,
                                                         const float* __restrict__ partial,
                                                         T* __restrict__ out_x, T* __restrict__ out_eps,
                                                         int32_t* __restrict__ best_idx, float* __restrict__ losses) {
  const int img = blockIdx.y;
  __shared__ float s_loss[ETA_MAX_CAND];
  __shared__ int s_best;
  if (threadIdx.x < n_cand) {
    const float* p = partial + ((int64_t)img * ETA_MAX_CAND + threadIdx.x) * ETA_LOSS_BLOCKS;
    float s = 0.f;
    for (int b = 0; b < ETA_LOSS_BLOCKS; ++b) s += p[b];
    s_loss[threadIdx.x] = s / (float)chw;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int best = 0;
    bool has_nan = false;
    for (int j = 0; j < n_cand; ++j) {
      float v = s_loss[j];
      if (v != v) { best = j; has_nan = true; break; }
    }
    if (!has_nan) {
      float m = s_loss[0];
      for (int j = 1; j < n_cand; ++j)
        if (s_loss[j] < m) { m = s_loss[j]; best = j; }
    }
    s_best = best;
    if (blockIdx.x == 0) {
      if (best_idx) best_idx[img] = best;
      if (losses)
        for (int j = 0; j < n_cand; ++j) losses[img * n_cand + j] = s_loss[j];
    }
  }
  __syncthreads();
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= chw) return;
  const float z = to_f32(noise[(int64_t)s_best * chw + e]);
  float eta_px = k.eta;
  if (use_mask == 1) eta_px = (to_f32(mask_map[(int64_t)img * hw + (e % hw)]) > k.thres) ? k.eta : 0.f;
  else if (use_mask == 2) eta_px = to_f32(mask_map[(int64_t)img * hw + (e % hw)]) * k.eta;   // soft / precomputed mask (thres None, pow)
  const float std_t = eta_px * sqrtf(k.var);
  const float dir_c = sqrtf(1.f - k.a_p - std_t * std_t);
  const float xp = to_f32(x_prev[(int64_t)img * chw + e]);
  float delta = 0.f;
#pragma unroll
  for (int role = 0; role < 2; ++role) {
    const int64_t lrow = (int64_t)(role * n_img + img) * chw + e;
    float u = to_f32(eps_all[(int64_t)(role * n_img + img) * chw + e]);
    float c = to_f32(eps_all[(int64_t)((2 + role) * n_img + img) * chw + e]);
    float eps = u + k.g * (c - u);
    float x0 = (to_f32(x[lrow]) - k.s1m_t * eps) / k.sa_t;
    float xn = k.sa_p * x0 + dir_c * eps + std_t * z;
    if (role == 0) {
      delta = xp - xn;                          // source correction (eta_inversion.py:247)
      xn = use_mask ? xn + delta : xp;
    } else if (k.tdir != 0.f) {                 // target_dirinv (eta_inversion.py:251-256): part of the correction leaks to the target
      const float md = dirinv_map ? to_f32(dirinv_map[(int64_t)img * hw + (e % hw)]) : 1.f;   // host passes 1 - mask_dirinv
      xn += k.tdir * md * delta;
    }
    out_x[lrow] = from_f32<T>(xn);
    if (out_eps) out_eps[lrow] = from_f32<T>(eps);
  }
}

// generic [3P] DDIMScheduler.step restatement (eps prediction): x' = sqrt(a_p) x0 + sqrt(1-a_p-(eta_px s)^2) eps + eta_px s z
// eta_px = eta * (mask ? mask[row % n_mask][pixel] : 1); z = noise[element] (shared by rows) or 0
template <typename T>
__global__ void ddim_eta_step_kernel(const T* __restrict__ x, const T* __restrict__ eps, float eta, const T* __restrict__ mask, int n_mask,
                                     const T* __restrict__ noise, float sa_t, float s1m_t, float sa_p, float a_p, float var, int rows,
                                     int chw, int hw, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * chw) return;
  const int row = (int)(i / chw), e = (int)(i - (int64_t)row * chw);
  float eta_px = eta;
  if (mask) eta_px *= to_f32(mask[(int64_t)(row % n_mask) * hw + (e % hw)]);
  const float std_t = eta_px * sqrtf(var);
  const float ep = to_f32(eps[i]);
  const float x0 = (to_f32(x[i]) - s1m_t * ep) / sa_t;
  float xn = sa_p * x0 + sqrtf(1.f - a_p - std_t * std_t) * ep;
  if (noise) xn += std_t * to_f32(noise[e]);
  out[i] = from_f32<T>(xn);
}

// out = a x + b y + c z (z optional): the update of the multistep DPM-Solver++ schedulers (x_t = (sigma_t / sigma_s) x - c0 m0 - c1 (m0 - m1))
template <typename T>
__global__ void lincomb3_kernel(const T* __restrict__ x, float a, const T* __restrict__ y, float b, const T* __restrict__ z, float c,
                                T* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float v = a * to_f32(x[i]) + b * to_f32(y[i]);
    if (z) v += c * to_f32(z[i]);
    out[i] = from_f32<T>(v);
  }
}

}  // namespace etainv

using namespace etainv;

extern "C" int etainv_lincomb3(const void* x, float a, const void* y, float b, const void* z, float c, void* out, int64_t n, int io_dtype,
                               void* stream) {
  ETAINV_CHECK(x && y && out && n >= 0, "null pointer or negative size");
  if (n == 0) return 0;
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
  ETAINV_DISPATCH_DTYPE(io_dtype, T,
                        hipLaunchKernelGGL(lincomb3_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, a, (const T*)y, b,
                                           (const T*)z, c, (T*)out, n));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

extern "C" int etainv_ddim_eta_step(const void* x, const void* eps, float eta, const void* eta_mask, int n_mask, const void* noise,
                                    float a_t, float a_p, float var, int rows, int c, int hw, void* out, int io_dtype, void* stream) {
  ETAINV_CHECK(x && eps && out && rows >= 1 && c >= 1 && hw >= 1, "bad arguments");
  ETAINV_CHECK(!eta_mask || n_mask >= 1, "n_mask");
  ETAINV_CHECK(a_t > 0.f && a_t < 1.f && a_p > 0.f && a_p <= 1.f, "alphas_cumprod out of range");
  const int chw = c * hw;
  const int64_t n = (int64_t)rows * chw;
  ETAINV_DISPATCH_DTYPE(io_dtype, T,
                        hipLaunchKernelGGL(ddim_eta_step_kernel<T>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                           (const T*)eps, eta, (const T*)eta_mask, n_mask, (const T*)noise, (float)sqrt((double)a_t),
                                           (float)sqrt(1.0 - (double)a_t), (float)sqrt((double)a_p), a_p, var, rows, chw, hw, (T*)out));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

extern "C" int etainv_cfg_combine(const void* eps_u, const void* eps_c, float g, void* out, int64_t n, int io_dtype, void* stream) {
  ETAINV_CHECK(eps_u && eps_c && out && n >= 0, "null pointer or negative size");
  if (n == 0) return 0;
  int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
  ETAINV_DISPATCH_DTYPE(io_dtype, T,
                        hipLaunchKernelGGL(cfg_combine_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)eps_u,
                                           (const T*)eps_c, g, (T*)out, n));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

extern "C" int etainv_ddim_step(const void* x, const void* eps, float a_from, float a_to, void* out, int64_t n, int io_dtype,
                                void* stream) {
  ETAINV_CHECK(x && eps && out && n >= 0, "null pointer or negative size");
  ETAINV_CHECK(a_from > 0.f && a_from <= 1.f && a_to > 0.f && a_to <= 1.f, "alphas_cumprod must be in (0,1]");
  if (n == 0) return 0;
  int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
  float s1f = (float)sqrt(1.0 - (double)a_from), isf = (float)(1.0 / sqrt((double)a_from));
  float st = (float)sqrt((double)a_to), s1t = (float)sqrt(1.0 - (double)a_to);
  ETAINV_DISPATCH_DTYPE(io_dtype, T,
                        hipLaunchKernelGGL(ddim_step_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                           (const T*)eps, s1f, isf, st, s1t, (T*)out, n));
  ETAINV_LAUNCH_CHECK();
  return 0;
}

extern "C" int etainv_eta_backward_step(const void* x, const void* eps_all, float g, const void* x_prev_src, const void* noise,
                                        int n_cand, float eta, const void* mask_map, float mask_thres, int use_mask, float a_t,
                                        float a_p, float var, int n_img, int c, int hw, void* out_x, void* out_eps,
                                        int32_t* best_idx, float* losses, float* scratch, int io_dtype, void* stream) {
  return etainv_eta_backward_step_ex(x, eps_all, g, x_prev_src, noise, n_cand, eta, mask_map, mask_thres, use_mask, a_t, a_p, var, n_img, c, hw,
                                     out_x, out_eps, best_idx, losses, scratch, io_dtype, 0.f, nullptr, stream);
}

extern "C" int etainv_eta_backward_step_ex(const void* x, const void* eps_all, float g, const void* x_prev_src, const void* noise,
                                           int n_cand, float eta, const void* mask_map, float mask_thres, int use_mask, float a_t,
                                           float a_p, float var, int n_img, int c, int hw, void* out_x, void* out_eps,
                                           int32_t* best_idx, float* losses, float* scratch, int io_dtype, float target_dirinv,
                                           const void* dirinv_map, void* stream) {
  ETAINV_CHECK(target_dirinv == 0.f || use_mask, "target_dirinv is part of the masked update (mask_mode_cfg)");
  ETAINV_CHECK(x && eps_all && x_prev_src && noise && out_x && scratch, "null pointer");
  ETAINV_CHECK(n_cand >= 1 && n_cand <= ETA_MAX_CAND, "noise_sample_count must be in [1,16]");
  ETAINV_CHECK(n_img >= 1 && c >= 1 && hw >= 1, "bad sizes");
  ETAINV_CHECK(!use_mask || mask_map, "use_mask needs mask_map");
  ETAINV_CHECK(a_t > 0.f && a_t < 1.f && a_p > 0.f && a_p <= 1.f, "alphas_cumprod out of range");
  EtaCoef k;
  k.g = g;
  k.eta = eta;
  k.sa_t = (float)sqrt((double)a_t);
  k.s1m_t = (float)sqrt(1.0 - (double)a_t);
  k.sa_p = (float)sqrt((double)a_p);
  k.a_p = a_p;
  k.var = var;
  k.thres = mask_thres;
  k.tdir = target_dirinv;
  const int chw = c * hw;
  hipStream_t s = (hipStream_t)stream;
  ETAINV_DISPATCH_DTYPE(
      io_dtype, T,
      hipLaunchKernelGGL(eta_loss_kernel<T>, dim3(ETA_LOSS_BLOCKS, n_img), dim3(256), 0, s, (const T*)x, (const T*)eps_all,
                         (const T*)x_prev_src, (const T*)noise, n_cand, k, n_img, chw, scratch);
      hipLaunchKernelGGL(eta_update_kernel<T>, dim3(cdiv(chw, 256), n_img), dim3(256), 0, s, (const T*)x, (const T*)eps_all,
                         (const T*)x_prev_src, (const T*)noise, n_cand, k, (const T*)mask_map, use_mask, (const T*)dirinv_map, n_img, chw, hw,
                         (const float*)scratch, (T*)out_x, (T*)out_eps, best_idx, losses));
  ETAINV_LAUNCH_CHECK();
  return 0;
}
