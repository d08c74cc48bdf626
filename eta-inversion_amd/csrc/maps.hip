// Consumers of the attention-map store (fp32 sums over steps of the cond-half cross-attention probabilities of
// the five (L/4)^2-token layers, layout [layer][img][role][head][pixel][77]):
//   word_maps_kernel   -> ControllerAttentionStorePerStep.end_step / get_attention_map / aggregate_attention
//                         (eta_inversion.py:44-49, ptp_editor.py:43-85, ptp.py:288-303)
//   local_blend_kernel -> LocalBlend.__call__ / get_mask (ptp.py:18-47)
#include "common.h"
#include "kernels.h"

namespace etainv {

__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;  // torch upsample_bicubic2d
  float x = t + 1.f;
  w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

// grid (n_tok, n_img), 256 threads.  out [n_img][n_tok][L][L]
__global__ void __launch_bounds__(256) word_maps_kernel(const float* __restrict__ acc, int n_layers, int n_img_cap, int row_sel,
                                                        int heads, int res, int L, const int32_t* __restrict__ tokens, int n_tok,
                                                        float inv_steps, float* __restrict__ out, int accumulate, float scale, unsigned layer_mask) {
  extern __shared__ float sm[];  // [res*res] aggregated map, then sm[res*res .. +4] reduction scratch
  const int img = blockIdx.y, ti = blockIdx.x;
  const int tok = tokens[img * n_tok + ti];
  const int RR = res * res;
  if (tok < 0 || tok >= 77) {   // a word index past the 77-token context (the reference raises IndexError, ptp.py:296): poison the map, never read out of range
    float* o = out + ((int64_t)img * n_tok + ti) * L * L;
    for (int p = threadIdx.x; p < L * L; p += blockDim.x) o[p] = __builtin_nanf("");
    return;
  }
  float lmax = -3.0e38f;
  for (int pix = threadIdx.x; pix < RR; pix += blockDim.x) {
    float s = 0.f;
    for (int l = 0; l < n_layers; ++l)
      if ((layer_mask >> l) & 1u)      // `from_where` of aggregate_attention (ptp.py:296-300): layers 0,1 = down, 2.. = up
        for (int h = 0; h < heads; ++h)
          s += acc[(((((int64_t)l * n_img_cap + img) * 2 + row_sel) * heads + h) * RR + pix) * 77 + tok] * inv_steps;
    s /= (float)(__builtin_popcount(layer_mask & ((1u << n_layers) - 1u)) * heads);
    sm[pix] = s;
    lmax = fmaxf(lmax, s);
  }
  lmax = wave_max(lmax);
  float* red = sm + RR;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lmax;
  __syncthreads();
  const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float inv = 1.f / mx;
  const float sc = (float)res / (float)L;
  float* o = out + ((int64_t)img * n_tok + ti) * L * L;
  for (int p = threadIdx.x; p < L * L; p += blockDim.x) {
    const int oy = p / L, ox = p - oy * L;
    float v;
    if (res == L) {
      v = sm[p] * inv;
    } else {
      const float sy = (oy + 0.5f) * sc - 0.5f, sx = (ox + 0.5f) * sc - 0.5f;
      const float fy = floorf(sy), fx = floorf(sx);
      float wy[4], wx[4];
      cubic_coeffs(sy - fy, wy);
      cubic_coeffs(sx - fx, wx);
      v = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int yy = min(max((int)fy - 1 + a, 0), res - 1);
        float r = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int xx = min(max((int)fx - 1 + c, 0), res - 1);
          r += wx[c] * (sm[yy * res + xx] * inv);
        }
        v += wy[a] * r;
      }
      v = fminf(fmaxf(v, 0.f), 1.f);
    }
    if (accumulate) o[p] += scale * v;
    else o[p] = v;
  }
}

// grid n_img, 256 threads.  x [2*n_img][4][L][L] fp32 in place.
// LocalBlend phase 1 (round 6): t[img][role][pix][l * heads + h] = sum_k acc[l][img][role][h][pix][k] * alpha[img][role][k], k in order -- one block per (layer-head, role,
// image), thread = pixel, each block streams one contiguous [pix][77] plane.  One block per image walking all 40 planes read 202 MB at 0.47 TB/s (B = 32: 432 us per call);
// the partials are summed in the (layer, head) order of that loop by local_blend_kernel: the same bits.
__global__ void __launch_bounds__(256) blend_partials_kernel(const float* __restrict__ acc, int n_img_cap, int heads, int RR, const float* __restrict__ blend_alpha,
                                                             float* __restrict__ tpart, int P) {
  const int lh = blockIdx.x, role = blockIdx.y, img = blockIdx.z;
  const int l = lh / heads, h = lh - l * heads;
  const float* al = blend_alpha + ((int64_t)img * 2 + role) * 77;
  for (int pix = threadIdx.x; pix < RR; pix += blockDim.x) {
    const float* a = acc + (((((int64_t)l * n_img_cap + img) * 2 + role) * heads + h) * RR + pix) * 77;
    float t = 0.f;
    for (int k = 0; k < 77; ++k) t += a[k] * al[k];
    tpart[(((int64_t)img * 2 + role) * RR + pix) * P + lh] = t;
  }
}

// Phase 2: 1024 threads per image; `tpart` != nullptr: the partial dot products come from blend_partials_kernel (summed here in the old order); else computed here.
__global__ void __launch_bounds__(1024) local_blend_kernel(const float* __restrict__ acc, int n_layers, int n_img_cap, int heads, int res,
                                                           int L, float* __restrict__ x, int n_img,
                                                           const float* __restrict__ blend_alpha, float thres, const float* __restrict__ tpart) {
  extern __shared__ float sm[];  // [2][RR] maps, [2][RR] pooled, [32] scratch
  const int img = blockIdx.x;
  const int RR = res * res;
  float* mp = sm;
  float* pl = sm + 2 * RR;
  float* red = sm + 4 * RR;
  {
    // image without blend words in a batch (its alpha rows are all zero): the reference builds no LocalBlend for it
    // (ptp.py:306-320, blend_words None) -- leave its latent untouched
    int any = 0;
    for (int k = threadIdx.x; k < 2 * 77; k += blockDim.x) any |= blend_alpha[(int64_t)img * 2 * 77 + k] != 0.f;
    if (!__syncthreads_or(any)) return;
  }
  const int P = n_layers * heads;
  if (tpart) {
    for (int idx = threadIdx.x; idx < 2 * RR; idx += blockDim.x) {
      const float* tp = tpart + ((int64_t)img * 2 * RR + idx) * P;
      float s = 0.f;
      for (int lh = 0; lh < P; ++lh) s += tp[lh];     // (layer, head) order of the single-thread loop below
      mp[idx] = s / (float)P;
    }
  } else {
  for (int idx = threadIdx.x; idx < 2 * RR; idx += blockDim.x) {
    const int role = idx / RR, pix = idx - role * RR;
    const float* al = blend_alpha + ((int64_t)img * 2 + role) * 77;
    float s = 0.f;
    for (int l = 0; l < n_layers; ++l)
      for (int h = 0; h < heads; ++h) {
        const float* a = acc + (((((int64_t)l * n_img_cap + img) * 2 + role) * heads + h) * RR + pix) * 77;
        float t = 0.f;
        for (int k = 0; k < 77; ++k) t += a[k] * al[k];
        s += t;
      }
    mp[idx] = s / (float)(n_layers * heads);
  }
  }
  __syncthreads();
  float lm[2] = {-3.0e38f, -3.0e38f};
  for (int idx = threadIdx.x; idx < 2 * RR; idx += blockDim.x) {
    const int role = idx / RR, pix = idx - role * RR;
    const int y = pix / res, xx = pix - y * res;
    float m = -3.0e38f;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xc = xx + dx;
        if (yy >= 0 && yy < res && xc >= 0 && xc < res) m = fmaxf(m, mp[role * RR + yy * res + xc]);
      }
    pl[idx] = m;
    lm[role] = fmaxf(lm[role], m);
  }
  lm[0] = wave_max(lm[0]);
  lm[1] = wave_max(lm[1]);
  if ((threadIdx.x & 63) == 0) {
    red[(threadIdx.x >> 6) * 2] = lm[0];
    red[(threadIdx.x >> 6) * 2 + 1] = lm[1];
  }
  __syncthreads();
  float m0 = red[0], m1 = red[1];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { m0 = fmaxf(m0, red[2 * w]); m1 = fmaxf(m1, red[2 * w + 1]); }
  const float sc = (float)res / (float)L;
  const int LL = L * L;
  float* xs = x + (int64_t)img * 4 * LL;
  float* xt = x + (int64_t)(n_img + img) * 4 * LL;
  for (int p = threadIdx.x; p < LL; p += blockDim.x) {
    const int oy = p / L, ox = p - oy * L;
    const int sy = min((int)floorf(oy * sc), res - 1), sx = min((int)floorf(ox * sc), res - 1);
    const bool on = (pl[sy * res + sx] / m0 > thres) || (pl[RR + sy * res + sx] / m1 > thres);
    if (!on) {
#pragma unroll
      for (int c = 0; c < 4; ++c) xt[c * LL + p] = xs[c * LL + p];
    } else {
      // reference: x_src + 1.0 * (x_tgt - x_src) (ptp.py:46); keep the same rounding
#pragma unroll
      for (int c = 0; c < 4; ++c) xt[c * LL + p] = xs[c * LL + p] + (xt[c * LL + p] - xs[c * LL + p]);
    }
  }
}

int launch_word_maps(const float* maps_acc, int n_layers, int n_img_cap, int rows_per_img, int row_sel, int heads, int res, int L,
                     int n_img, const int32_t* tokens, int n_tok, int steps_done, float* out, int accumulate, float scale,
                     hipStream_t s, unsigned layer_mask) {
  (void)rows_per_img;
  ETAINV_CHECK(maps_acc && tokens && out && n_img > 0 && n_tok > 0 && steps_done > 0, "bad arguments");
  ETAINV_CHECK(n_layers >= 1 && n_layers <= 31 && (layer_mask & ((1u << n_layers) - 1u)) != 0, "no layer selected");
  const size_t lds = (size_t)(res * res + 8) * sizeof(float);
  hipLaunchKernelGGL(word_maps_kernel, dim3(n_tok, n_img), dim3(256), lds, s, maps_acc, n_layers, n_img_cap, row_sel, heads, res, L,
                     tokens, n_tok, 1.0f / (float)steps_done, out, accumulate, scale, layer_mask);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

int launch_local_blend(const float* maps_acc, int n_layers, int n_img_cap, int heads, int res, int L, float* x, int n_img,
                       const float* blend_alpha, float thres, hipStream_t s) {
  ETAINV_CHECK(maps_acc && x && blend_alpha && n_img > 0, "bad arguments");
  const int RR = res * res, P = n_layers * heads;
  const size_t lds = (size_t)(4 * RR + 32) * sizeof(float);
  // partials workspace [n_img][2][RR][P] (2.6 MB at B = 32, L = 64): one allocation per device, grown on demand (launches on a device are serialised by the caller's stream)
  static float* ws[kMaxDevices] = {};
  static size_t ws_bytes[kMaxDevices] = {};
  const int dev = current_device();
  float* tpart = nullptr;
  if (!env_on("ETAINV_BLEND_NOSPLIT")) {
    const size_t need = (size_t)n_img * 2 * RR * P * sizeof(float);
    if (need > ws_bytes[dev]) {
      if (ws[dev]) { ETAINV_HIP(hipStreamSynchronize(s)); ETAINV_HIP(hipFree(ws[dev])); }
      ETAINV_HIP(hipMalloc(&ws[dev], need));
      ws_bytes[dev] = need;
    }
    tpart = ws[dev];
    hipLaunchKernelGGL(blend_partials_kernel, dim3(P, 2, n_img), dim3(256), 0, s, maps_acc, n_img_cap, heads, RR, blend_alpha, tpart, P);
  }
  hipLaunchKernelGGL(local_blend_kernel, dim3(n_img), dim3(1024), lds, s, maps_acc, n_layers, n_img_cap, heads, res, L, x, n_img,
                     blend_alpha, thres, (const float*)tpart);
  ETAINV_LAUNCH_CHECK();
  return 0;
}

}  // namespace etainv
