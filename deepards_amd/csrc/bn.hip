// Window-grouped BatchNorm1d (train mode) forward/backward, fused with ReLU / residual add.
//
// Replaces nn.BatchNorm1d + nn.ReLU (+ `out += residual`) of reference models/resnet.py:27-38,
// 141-153 and models/densenet.py:23-29,72-74,146,181-182, as executed by the per-window loop of
// models/torch_cnn_linear_network.py:108-113: statistics are over (rows_per_window x L) per channel
// PER WINDOW (SURVEY.md finding 3), biased variance, eps inside the sqrt.
//
// Pure bandwidth kernels.  A window is one contiguous [Wn = rows_per_window*L][ld] slab; a block
// owns (window, 32-channel group): lanes 0-7 of each 8-lane group read one position's 128 B line as
// float4, 32 position slots stride the slab, LDS folds the slots.  Two-pass mean / centred variance
// (second pass L2-hot) -- no E[x^2]-E[x]^2 cancellation.
#include "common.h"

#define CG 32       // channels per block
#define SLOTS 32    // position slots per block (256 threads / 8 quads)

__device__ __forceinline__ void block_fold(float (&v)[4], float* red, int slot, int q, float (&out)[4]) {
  // red: [SLOTS][CG]; returns the per-channel total to every thread that owns those channels
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) red[slot * CG + q * 4 + e] = v[e];
  __syncthreads();
  if (threadIdx.x < CG) {
    float s = 0.f;
    for (int k = 0; k < SLOTS; ++k) s += red[k * CG + threadIdx.x];
    red[SLOTS * CG + threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) out[e] = red[SLOTS * CG + q * 4 + e];
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int ld, int Wn, int C, float eps,
                                                       float* __restrict__ mean, float* __restrict__ invstd) {
  __shared__ float red[(SLOTS + 1) * CG];
  const int w = blockIdx.x, cg = blockIdx.y;
  const int q = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const float* base = x + (size_t)w * Wn * ld + cg * CG + q * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, m[4];
  for (int p = slot; p < Wn; p += SLOTS) {
    f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * ld);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += v[e];
  }
  block_fold(s, red, slot, q, m);
  const float inv_n = 1.0f / (float)Wn;
#pragma unroll
  for (int e = 0; e < 4; ++e) m[e] *= inv_n;
  float s2[4] = {0.f, 0.f, 0.f, 0.f}, var[4];
  for (int p = slot; p < Wn; p += SLOTS) {
    f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * ld);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d = v[e] - m[e];
      s2[e] += d * d;
    }
  }
  block_fold(s2, red, slot, q, var);
  if (slot == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int c = cg * CG + q * 4 + e;
      mean[(size_t)w * C + c] = m[e];
      invstd[(size_t)w * C + c] = 1.0f / sqrtf(var[e] * inv_n + eps);
    }
  }
}

// running stats: r <- (1-mom) r + mom * stat_w, one update per window, in window order
// (SURVEY.md finding 5; unbiased variance n/(n-1)).  Closed form of the W sequential updates:
//   r_W = (1-mom)^W r_0 + mom * sum_w (1-mom)^(W-1-w) stat_w
// block = 32 channels x 8 window slots, folded through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void bn_running_kernel(const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         int W, int C, int Wn, float eps, float momentum,
                                                         float* __restrict__ rmean, float* __restrict__ rvar,
                                                         long long* __restrict__ num_batches_tracked) {
  __shared__ float red[2][8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), slot = threadIdx.x >> 5;
  const float keep = 1.f - momentum;
  const float unb = (float)Wn / (float)(Wn - 1);
  float am = 0.f, av = 0.f;
  if (c < C) {
    for (int w = slot; w < W; w += 8) {
      float wt = momentum * powf(keep, (float)(W - 1 - w));
      float is = invstd[(size_t)w * C + c];
      am = fmaf(wt, mean[(size_t)w * C + c], am);
      av = fmaf(wt, (1.0f / (is * is) - eps) * unb, av);
    }
  }
  red[0][slot][threadIdx.x & 31] = am;
  red[1][slot][threadIdx.x & 31] = av;
  __syncthreads();
  if (threadIdx.x < 32 && c < C) {
    float sm = 0.f, sv = 0.f;
    for (int k = 0; k < 8; ++k) {
      sm += red[0][k][threadIdx.x];
      sv += red[1][k][threadIdx.x];
    }
    const float decay = powf(keep, (float)W);
    rmean[c] = fmaf(decay, rmean[c], sm);
    rvar[c] = fmaf(decay, rvar[c], sv);
  }
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += W;
}

// out = act( (x-mean)*invstd*gamma + beta (+ res) )
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ res,
                                                       int ldr, float* __restrict__ out, int ldo, int Wn, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int relu, int chunk) {
  const int w = blockIdx.x, cg = blockIdx.y;
  const int q = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int c0 = cg * CG + q * 4;
  f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * C + c0);
  f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * C + c0);
  f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
  const int p_beg = blockIdx.z * chunk;
  const int p_end = min(Wn, p_beg + chunk);
  for (int p = p_beg + slot; p < p_end; p += SLOTS) {
    size_t pos = (size_t)w * Wn + p;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pos * ldx + c0);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[e] - mu[e]) * is[e] * ga[e] + be[e];
    if (res) {
      f32x4 r = *reinterpret_cast<const f32x4*>(res + pos * ldr + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += r[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + pos * ldo + c0) = o;
  }
}

// Backward of out = act(bn(x) (+res)).
//   mask_mode 0: no ReLU            g = dout
//             1: ReLU, no residual  g = dout * [bn(x) > 0]      (mask recomputed, `out` not read)
//             2: ReLU with residual g = dout * [out > 0]
//   dx = gamma*invstd*(g - mean_w(g) - xhat*mean_w(g*xhat));  gout (optional) = g  (residual branch)
//   ds1[w][c] = sum g, ds2[w][c] = sum g*xhat  (folded over windows into dbeta/dgamma afterwards)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dout, int ldd, const float* __restrict__ x,
                                                     int ldx, const float* __restrict__ outp, int ldo,
                                                     float* __restrict__ dx, int lddx, float* __restrict__ gout, int ldg,
                                                     int Wn, int C, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int mask_mode,
                                                     float* __restrict__ ds1, float* __restrict__ ds2) {
  __shared__ float red[(SLOTS + 1) * CG];
  const int w = blockIdx.x, cg = blockIdx.y;
  const int q = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int c0 = cg * CG + q * 4;
  f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * C + c0);
  f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * C + c0);
  f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = slot; p < Wn; p += SLOTS) {
    size_t pos = (size_t)w * Wn + p;
    f32x4 g = *reinterpret_cast<const f32x4*>(dout + pos * ldd + c0);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pos * ldx + c0);
    f32x4 xh;
#pragma unroll
    for (int e = 0; e < 4; ++e) xh[e] = (v[e] - mu[e]) * is[e];
    if (mask_mode == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (xh[e] * ga[e] + be[e] > 0.f) ? g[e] : 0.f;
    } else if (mask_mode == 2) {
      f32x4 o = *reinterpret_cast<const f32x4*>(outp + pos * ldo + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (o[e] > 0.f) ? g[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[e] += g[e];
      s2[e] += g[e] * xh[e];
    }
  }
  float t1[4], t2[4];
  block_fold(s1, red, slot, q, t1);
  block_fold(s2, red, slot, q, t2);
  if (slot == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ds1[(size_t)w * C + c0 + e] = t1[e];
      ds2[(size_t)w * C + c0 + e] = t2[e];
    }
  }
  const float inv_n = 1.0f / (float)Wn;
  for (int p = slot; p < Wn; p += SLOTS) {
    size_t pos = (size_t)w * Wn + p;
    f32x4 g = *reinterpret_cast<const f32x4*>(dout + pos * ldd + c0);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pos * ldx + c0);
    f32x4 xh, d;
#pragma unroll
    for (int e = 0; e < 4; ++e) xh[e] = (v[e] - mu[e]) * is[e];
    if (mask_mode == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (xh[e] * ga[e] + be[e] > 0.f) ? g[e] : 0.f;
    } else if (mask_mode == 2) {
      f32x4 o = *reinterpret_cast<const f32x4*>(outp + pos * ldo + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (o[e] > 0.f) ? g[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = ga[e] * is[e] * (g[e] - t1[e] * inv_n - xh[e] * t2[e] * inv_n);
    *reinterpret_cast<f32x4*>(dx + pos * lddx + c0) = d;
    if (gout) *reinterpret_cast<f32x4*>(gout + pos * ldg + c0) = g;
  }
}

// dbeta[c] (+)= sum_w s1[w][c];  dgamma[c] (+)= sum_w s2[w][c]   (fixed order: deterministic)
// block = 32 channels x 8 window slots.
__global__ __launch_bounds__(256) void bn_param_grad_kernel(const float* __restrict__ s1, const float* __restrict__ s2,
                                                            int W, int C, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int accumulate) {
  __shared__ float red[2][8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), slot = threadIdx.x >> 5;
  float a = 0.f, b = 0.f;
  if (c < C) {
    for (int w = slot; w < W; w += 8) {
      a += s1[(size_t)w * C + c];
      b += s2[(size_t)w * C + c];
    }
  }
  red[0][slot][threadIdx.x & 31] = a;
  red[1][slot][threadIdx.x & 31] = b;
  __syncthreads();
  if (threadIdx.x < 32 && c < C) {
    a = 0.f;
    b = 0.f;
    for (int k = 0; k < 8; ++k) {
      a += red[0][k][threadIdx.x];
      b += red[1][k][threadIdx.x];
    }
    dbeta[c] = accumulate ? dbeta[c] + a : a;
    dgamma[c] = accumulate ? dgamma[c] + b : b;
  }
}

extern "C" {

// mean/invstd: [W][C].  x: [W*Wn][ld].  C % 32 == 0, ld % 4 == 0.
int da_bn_stats(const float* x, int ld, int W, int Wn, int C, float eps, float* mean, float* invstd,
                hipStream_t stream) {
  DA_ENTER();
  if (!x || !mean || !invstd || C % CG || ld % 4 || Wn < 1) return DA_EINVAL;
  if (W == 0) return DA_OK;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(W, C / CG), dim3(256), 0, stream, x, ld, Wn, C, eps, mean, invstd);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_bn_running_update(const float* mean, const float* invstd, int W, int C, int Wn, float eps, float momentum,
                         float* running_mean, float* running_var, long long* num_batches_tracked,
                         hipStream_t stream) {
  DA_ENTER();
  if (!mean || !invstd || !running_mean || !running_var || Wn < 2) return DA_EINVAL;
  hipLaunchKernelGGL(bn_running_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, mean, invstd, W, C, Wn, eps,
                     momentum, running_mean, running_var, num_batches_tracked);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_bn_apply(const float* x, int ldx, const float* res, int ldr, float* out, int ldo, int W, int Wn, int C,
                const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                hipStream_t stream) {
  DA_ENTER();
  if (!x || !out || !mean || !invstd || !gamma || !beta || C % CG || ldx % 4 || ldo % 4 || (res && ldr % 4))
    return DA_EINVAL;
  if (W == 0) return DA_OK;
  int chunk = 256;
  int nz = (Wn + chunk - 1) / chunk;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(W, C / CG, nz), dim3(256), 0, stream, x, ldx, res, ldr, out, ldo, Wn, C,
                     mean, invstd, gamma, beta, relu, chunk);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// scratch: 2*W*C floats.  dgamma/dbeta: [C] (accumulated when accumulate != 0).
int da_bn_bwd(const float* dout, int ldd, const float* x, int ldx, const float* out, int ldo, float* dx, int lddx,
              float* gout, int ldg, int W, int Wn, int C, const float* mean, const float* invstd, const float* gamma,
              const float* beta, int mask_mode, float* scratch, float* dgamma, float* dbeta, int accumulate,
              hipStream_t stream) {
  DA_ENTER();
  if (!dout || !x || !dx || !mean || !invstd || !gamma || !beta || !scratch || !dgamma || !dbeta) return DA_EINVAL;
  if (C % CG || ldd % 4 || ldx % 4 || lddx % 4 || (gout && ldg % 4) || mask_mode < 0 || mask_mode > 2)
    return DA_EINVAL;
  if (mask_mode == 2 && (!out || ldo % 4)) return DA_EINVAL;
  if (W == 0) return DA_OK;
  float* s1 = scratch;
  float* s2 = scratch + (size_t)W * C;
  hipLaunchKernelGGL(bn_bwd_kernel, dim3(W, C / CG), dim3(256), 0, stream, dout, ldd, x, ldx, out, ldo, dx, lddx, gout,
                     ldg, Wn, C, mean, invstd, gamma, beta, mask_mode, s1, s2);
  DA_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, s1, s2, W, C, dgamma, dbeta,
                     accumulate);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

}  // extern "C"
