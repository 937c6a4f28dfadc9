// Classifier head (flatten -> Linear(NB*F, 2)), BCE-with-logits, fused clamp+optimiser steps and the
// small layout/elementwise helpers.
//
// Replaces: reference models/torch_cnn_linear_network.py:102,110-112 (linear_final on view(-1) of the
// (NB, F) feature block), train_ards_detector.py:530,929-930 (BCEWithLogitsLoss, mean over B*2),
// :474-476 (clamp(g, +-clip) hooks), :419-421 (Adam / SGD momentum .9 nesterov weight-decay),
// models/densenet.py:37-40 (F.dropout + torch.cat).
#include "common.h"

// logits[b][o] = bias[o] + sum_i flat[b][i] * W[o][i], o in {0,1}.  flat[b] is the window's (NB, F)
// feature block, contiguous (row r, feature f at r*F + f) -- exactly view(-1).
__global__ __launch_bounds__(256) void linear2_fwd_kernel(const float* __restrict__ flat, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ logits,
                                                          int K) {
  __shared__ float red[2][4];
  const int b = blockIdx.x;
  const float* f = flat + (size_t)b * K;
  float a0 = 0.f, a1 = 0.f;
  for (int i = threadIdx.x * 4; i < K; i += blockDim.x * 4) {
    f32x4 v = *reinterpret_cast<const f32x4*>(f + i);
    f32x4 w0 = *reinterpret_cast<const f32x4*>(W + i);
    f32x4 w1 = *reinterpret_cast<const f32x4*>(W + K + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a0 = fmaf(v[e], w0[e], a0);
      a1 = fmaf(v[e], w1[e], a1);
    }
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = a0;
    red[1][wave] = a1;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    float s = bias[threadIdx.x];
    for (int k = 0; k < 4; ++k) s += red[threadIdx.x][k];
    logits[(size_t)b * 2 + threadIdx.x] = s;
  }
}

// loss = mean(max(x,0) - x t + log1p(exp(-|x|)));  dlogits = (sigmoid(x) - t) * gscale / n
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ x, const float* __restrict__ t, int n,
                                                  float gscale, float* __restrict__ loss, float* __restrict__ dx) {
  __shared__ float red[4];
  float acc = 0.f;
  const float inv = 1.0f / (float)n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float v = x[i], tt = t[i];
    acc += fmaxf(v, 0.f) - v * tt + log1pf(expf(-fabsf(v)));
    if (dx) {
      float sg = 1.0f / (1.0f + expf(-v));
      dx[i] = (sg - tt) * inv * gscale;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) * inv;
}

// dflat[b][i] = dl[b][0] W[0][i] + dl[b][1] W[1][i]
__global__ __launch_bounds__(256) void linear2_bwd_input_kernel(const float* __restrict__ dl, const float* __restrict__ W,
                                                                float* __restrict__ dflat, int B, int K) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)B * (K / 4);
  if (idx >= total) return;
  int b = (int)(idx / (K / 4));
  int i = (int)(idx % (K / 4)) * 4;
  float d0 = dl[b * 2], d1 = dl[b * 2 + 1];
  f32x4 w0 = *reinterpret_cast<const f32x4*>(W + i);
  f32x4 w1 = *reinterpret_cast<const f32x4*>(W + K + i);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = d0 * w0[e] + d1 * w1[e];
  *reinterpret_cast<f32x4*>(dflat + (size_t)b * K + i) = o;
}

// dW[o][i] (+)= sum_b dl[b][o] flat[b][i];  dbias[o] (+)= sum_b dl[b][o].  Block = 64 columns (16 float4 lanes) x 16 row
// slots folded through LDS in a fixed order (deterministic); rows = windows for cnn_linear, window*breath rows for the
// per-breath heads (1280 at B = 64: the one-thread-per-column loop over all rows took hundreds of microseconds there).
// (dl / terms may live in LDS: the fused head's backward kernel runs these blocks itself)
__device__ __forceinline__ void linear2_bwd_weight_block(const float* dl, const float* __restrict__ flat,
                                                         float* __restrict__ dW, float* __restrict__ dbias, int B, int K,
                                                         int accumulate, const float* terms, float inv_n,
                                                         float* __restrict__ loss, int blk, f32x4 (*red)[16][16]) {
  const int kq = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int i = (blk * 16 + kq) * 4;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  if (i < K) {
    for (int b = slot; b < B; b += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(flat + (size_t)b * K + i);
      const float d0 = dl[b * 2], d1 = dl[b * 2 + 1];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] = fmaf(d0, v[e], a0[e]);
        a1[e] = fmaf(d1, v[e], a1[e]);
      }
    }
  }
  red[0][slot][kq] = a0;
  red[1][slot][kq] = a1;
  __syncthreads();
  if (slot == 0 && i < K) {
    for (int k = 1; k < 16; ++k) {
      const f32x4 r0 = red[0][k][kq], r1 = red[1][k][kq];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] += r0[e];
        a1[e] += r1[e];
      }
    }
    if (accumulate) {
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(dW + i);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(dW + K + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] += p0[e];
        a1[e] += p1[e];
      }
    }
    *reinterpret_cast<f32x4*>(dW + i) = a0;
    *reinterpret_cast<f32x4*>(dW + K + i) = a1;
  }
  if (blk == 0 && threadIdx.x < 64) {                 // first wave: the two bias gradients (fixed shuffle tree)
    float s0 = 0.f, s1 = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) {
      s0 += dl[b * 2];
      s1 += dl[b * 2 + 1];
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (threadIdx.x == 0) {
      dbias[0] = accumulate ? dbias[0] + s0 : s0;
      dbias[1] = accumulate ? dbias[1] + s1 : s1;
    }
    if (terms) {                                        // the fused head: the loss is the mean of the windows' BCE terms
      float s = 0.f;
      for (int b = threadIdx.x; b < B; b += 64) s += terms[b];
      s = wave_sum(s);
      if (threadIdx.x == 0) loss[0] = s * inv_n;
    }
  }
}

__global__ __launch_bounds__(256) void linear2_bwd_weight_kernel(const float* __restrict__ dl,
                                                                 const float* __restrict__ flat, float* __restrict__ dW,
                                                                 float* __restrict__ dbias, int B, int K,
                                                                 int accumulate, const float* __restrict__ terms = nullptr,
                                                                 float inv_n = 0.f, float* __restrict__ loss = nullptr) {
  __shared__ f32x4 red[2][16][16];
  linear2_bwd_weight_block(dl, flat, dW, dbias, B, K, accumulate, terms, inv_n, loss, blockIdx.x, red);
}

// ---------------------------------------------------------------------------------------------------------------------
// The head chain of CNNLinearNetwork in TWO launches instead of six (reference models/resnet.py:112,159-160 /
// densenet.py:167,183-184 AvgPool1d(7,1) + view; torch_cnn_linear_network.py:102,110-112 linear_final on view(-1);
// train_ards_detector.py:530 BCEWithLogitsLoss, and their backward):
//   head_pool_dot_kernel  one block per (window, row group): global average pool of its rows -> flat (kept for dW) and its
//                         share of the two dot products (a block per window could not pull its 287 KB fast enough: 12 us)
//   head_bwd_kernel       logits from the partial dot products, the window's BCE terms and dlogits = (sigmoid - t) gscale / n
//                         (n = 2 B elements), dx[row][l][f] = (dl[b][0] W[0][r F + f] + dl[b][1] W[1][r F + f]) / L
//                         ... and, in further blocks of the SAME launch (they recompute every window's dlogits / term from
//                         the partials into LDS), dW, dbias and loss = mean of the terms (linear2_bwd_weight_block, fixed order)
//   (forward only: head_finish_kernel = logits + loss)
// ---------------------------------------------------------------------------------------------------------------------
// block g of window b: 256 of the window's R * F / 4 (row, 4 features) items -- their pooled features -> flat, and the block's
// share of the two dot products -> part[b][g]
template <typename AT>
__global__ __launch_bounds__(256) void head_pool_dot_kernel(const AT* __restrict__ x, int ldx, const float* __restrict__ W,
                                                            float* __restrict__ flat, float* __restrict__ part, int R, int G,
                                                            int L, int F) {
  __shared__ float red[2][4];
  const int b = blockIdx.x, g = blockIdx.y, K = R * F, nq = F >> 2;
  const int i = g * 256 + threadIdx.x;
  float a0 = 0.f, a1 = 0.f;
  if (i < R * nq) {
    const int r = i / nq, q = i - r * nq;
    const AT* xp = x + ((size_t)(b * R + r) * L) * ldx + q * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int l0 = 0; l0 < L; l0 += 8) {                           // 8 loads in flight (L = 7: all of them)
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = Act<AT>::ld4(xp + (size_t)min(l0 + j, L - 1) * ldx);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (l0 + j < L) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += v[j][e];
        }
    }
    const float inv_l = 1.0f / (float)L;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] *= inv_l;                  // (avgpool_fwd_kernel's arithmetic)
    const int k = r * F + q * 4;
    *reinterpret_cast<f32x4*>(flat + (size_t)b * K + k) = acc;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(W + k), w1 = *reinterpret_cast<const f32x4*>(W + K + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a0 = fmaf(acc[e], w0[e], a0);
      a1 = fmaf(acc[e], w1[e], a1);
    }
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a0;
    red[1][threadIdx.x >> 6] = a1;
  }
  __syncthreads();
  if (threadIdx.x < 2) part[((size_t)b * G + g) * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// logits of window b from the row groups' partial dot products (fixed order), its BCE terms and dBCE/dlogits
__device__ __forceinline__ void head_logits(const float* __restrict__ part, const float* __restrict__ bias,
                                            const float* __restrict__ target, int b, int G, float inv_n, float gscale,
                                            float (&lg)[2], float (&dl)[2], float& term) {
  term = 0.f;
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    float v = bias[o];
    for (int g = 0; g < G; ++g) v += part[((size_t)b * G + g) * 2 + o];
    const float tt = target[(size_t)b * 2 + o];
    lg[o] = v;
    term += fmaxf(v, 0.f) - v * tt + log1pf(expf(-fabsf(v)));      // (bce_kernel's)
    dl[o] = (1.0f / (1.0f + expf(-v)) - tt) * inv_n * gscale;
  }
}

// forward only: logits [B][2] and loss = mean of the terms, one block
__global__ __launch_bounds__(256) void head_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                          const float* __restrict__ target, float* __restrict__ logits,
                                                          float* __restrict__ loss, int B, int G, float inv_n) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float lg[2], dl[2], term;
    head_logits(part, bias, target, b, G, inv_n, 1.f, lg, dl, term);
    logits[b * 2] = lg[0];
    logits[b * 2 + 1] = lg[1];
    acc += term;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}

// backward, same (window, 256 items) blocks: the window's dlogits from the partials (two threads, through LDS), then every
// thread writes the L positions of its (row, 4 features); block 0 of a window also publishes logits / dlogits / terms for the
// weight kernel and the caller
#define HEAD_MAXB 512     // windows whose dlogits / terms the weight blocks of head_bwd_kernel keep in LDS (more: two launches)
template <typename AT>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                       const float* __restrict__ target, const float* __restrict__ W,
                                                       AT* __restrict__ dx, int lddx, float* __restrict__ logits,
                                                       float* __restrict__ dlogits, float* __restrict__ terms, int R, int G,
                                                       int L, int F, float inv_n, float gscale, const float* __restrict__ flat,
                                                       float* __restrict__ dW, float* __restrict__ dbias,
                                                       float* __restrict__ loss, int accumulate, int wblocks) {
  if ((int)blockIdx.y >= G) {                           // the weight-gradient blocks (wblocks > 0 only)
    __shared__ float wdl[2 * HEAD_MAXB], wterm[HEAD_MAXB];
    __shared__ f32x4 red[2][16][16];
    const int B = gridDim.x, blk = ((int)blockIdx.y - G) * B + (int)blockIdx.x;
    if (blk >= wblocks) return;
    for (int b = threadIdx.x; b < B; b += 256) {
      float lg[2], dl[2], term;
      head_logits(part, bias, target, b, G, inv_n, gscale, lg, dl, term);
      wdl[2 * b] = dl[0];
      wdl[2 * b + 1] = dl[1];
      wterm[b] = term;
    }
    __syncthreads();
    linear2_bwd_weight_block(wdl, flat, dW, dbias, B, R * F, accumulate, wterm, inv_n, loss, blk, red);
    return;
  }
  __shared__ float sdl[2];
  const int b = blockIdx.x, g = blockIdx.y, K = R * F, nq = F >> 2;
  const int i = g * 256 + threadIdx.x;
  f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
  int r = 0, q = 0;
  if (i < R * nq) {
    r = i / nq;
    q = i - r * nq;
    w0 = *reinterpret_cast<const f32x4*>(W + r * F + q * 4);
    w1 = *reinterpret_cast<const f32x4*>(W + K + r * F + q * 4);
  }
  if (threadIdx.x == 0) {
    float lg[2], dl[2], term;
    head_logits(part, bias, target, b, G, inv_n, gscale, lg, dl, term);
    sdl[0] = dl[0];
    sdl[1] = dl[1];
    if (g == 0) {
      logits[b * 2] = lg[0];
      logits[b * 2 + 1] = lg[1];
      dlogits[b * 2] = dl[0];
      dlogits[b * 2 + 1] = dl[1];
      terms[b] = term;
    }
  }
  __syncthreads();
  if (i >= R * nq) return;
  const float d0 = sdl[0], d1 = sdl[1], inv = 1.0f / (float)L;
  f32x4 gq;
#pragma unroll
  for (int e = 0; e < 4; ++e) gq[e] = (d0 * w0[e] + d1 * w1[e]) * inv;          // (linear2_bwd_input then avgpool_bwd)
  AT* o = dx + ((size_t)(b * R + r) * L) * lddx + q * 4;
  for (int l = 0; l < L; ++l) Act<AT>::st4(o + (size_t)l * lddx, gq);
}

// g <- clamp(g*gscale, +-clip); g += wd*p; buf = first ? g : mom*buf + g; p -= lr*(g + mom*buf)
// (four elements a thread: the flat buffers are 256-B aligned segments, n4 = n / 4 quads + a scalar tail)
__global__ __launch_bounds__(256) void clamp_sgd_nesterov_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                 float* __restrict__ buf, size_t n, float lr, float mom,
                                                                 float wd, float clip, float gscale, int first) {
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(buf)) & 15) == 0;
  const size_t n4 = al ? n >> 2 : 0, stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = t0; i < n4; i += stride) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], bv = {0.f, 0.f, 0.f, 0.f};
    if (!first) bv = reinterpret_cast<const f32x4*>(buf)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gi = gv[e] * gscale;
      if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
      gi = fmaf(wd, pv[e], gi);
      const float bi = first ? gi : fmaf(mom, bv[e], gi);
      bv[e] = bi;
      pv[e] = pv[e] - lr * fmaf(mom, bi, gi);
    }
    reinterpret_cast<f32x4*>(buf)[i] = bv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
  for (size_t i = (n4 << 2) + t0; i < n; i += stride) {
    float gi = g[i] * gscale;
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    float pi = p[i];
    gi = fmaf(wd, pi, gi);
    float bi = first ? gi : fmaf(mom, buf[i], gi);
    buf[i] = bi;
    p[i] = pi - lr * fmaf(mom, bi, gi);
  }
}

// torch.optim.Adam (no weight decay, no amsgrad): bc1 = 1-b1^t, bc2 = 1-b2^t supplied by the host
__global__ __launch_bounds__(256) void clamp_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, size_t n,
                                                         float lr, float b1, float b2, float eps, float bc1, float bc2,
                                                         float clip, float gscale) {
  const float step = lr / bc1;
  const float rs = 1.0f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) * rs + eps;
    p[i] -= step * (mi / denom);
  }
}

// the same with the step count on the device (a captured graph replays it): *step_ptr is incremented by the first thread
// of a preceding launch (counter_inc_kernel), every thread derives the bias corrections from it
__global__ __launch_bounds__(256) void clamp_adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v, size_t n,
                                                             float lr, float b1, float b2, float eps,
                                                             const long long* __restrict__ step_ptr, float clip,
                                                             float gscale) {
  const float t = (float)step_ptr[0];
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float step = lr / bc1;
  const float rs = 1.0f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) * rs + eps;
    p[i] -= step * (mi / denom);
  }
}

__global__ void counter_inc_kernel(long long* c) {
  if (blockIdx.x == 0 && threadIdx.x == 0) c[0] += 1;
}

// W[co][ci][k] (torch) -> Wf[k][co][ci] and Wd[k][ci][co]
__global__ __launch_bounds__(256) void repack_conv_weight_kernel(const float* __restrict__ W, float* __restrict__ Wf,
                                                                 float* __restrict__ Wd, int Co, int Ci, int K) {
  int total = Co * Ci * K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int k = i % K;
    int ci = (i / K) % Ci;
    int co = i / (K * Ci);
    float v = W[i];
    if (Wf) Wf[((size_t)k * Co + co) * Ci + ci] = v;
    if (Wd) Wd[((size_t)k * Ci + ci) * Co + co] = v;
  }
}

// Winograd taps of one (n, c) pair at u[j * tot]: F(2,3) (4 points, the arithmetic of wino_weight_kernel) or
// F(4,3) (6 points, wino4_taps)
__device__ __forceinline__ void emit_wino_taps(float* u, size_t tot, float g0, float g1, float g2, int points) {
  if (points == 6) {
    wino4_taps(g0, g1, g2, u, tot);
    return;
  }
  u[0] = g0;
  u[tot] = (g0 + g1 + g2) * 0.5f;
  u[2 * tot] = (g0 - g1 + g2) * 0.5f;
  u[3 * tot] = g2;
}

// the same repack for up to 32 weights in one launch (blockIdx.y = which weight)
struct RepackDesc {
  const float* W;
  float* Wf;
  float* Wd;
  float* Uf;   // K == 3 only: Winograd taps [points][Co][Ci] (forward) ...
  float* Ud;   // ... and [points][Ci][Co] (data gradient), same arithmetic as the weight kernels of conv_wino.hip
  int Co, Ci, K;
  int points;  // 6: F(4,3) taps, otherwise F(2,3) (4)
};
struct RepackTable {
  RepackDesc d[32];
};
__global__ __launch_bounds__(256) void repack_multi_kernel(RepackTable t) {
  const RepackDesc& d = t.d[blockIdx.y];
  int total = d.Co * d.Ci * d.K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int k = i % d.K;
    int ci = (i / d.K) % d.Ci;
    int co = i / (d.K * d.Ci);
    float v = d.W[i];
    if (d.Wf) d.Wf[((size_t)k * d.Co + co) * d.Ci + ci] = v;
    if (d.Wd) d.Wd[((size_t)k * d.Ci + ci) * d.Co + co] = v;
    if (k == 0 && d.K == 3 && (d.Uf || d.Ud)) {
      const float w0 = v, w1 = d.W[i + 1], w2 = d.W[i + 2];
      const size_t tot = (size_t)d.Co * d.Ci;
      if (d.Uf) {
        emit_wino_taps(d.Uf + (size_t)co * d.Ci + ci, tot, w0, w1, w2, d.points);
      }
      if (d.Ud) {                              // taps reversed: g_t = w[..][2 - t]
        emit_wino_taps(d.Ud + (size_t)ci * d.Co + co, tot, w2, w1, w0, d.points);
      }
    }
  }
}

// one weight into the CHUNKED split-bf16 pack of conv3_x3p_kernel (conv_x3p.hip): the 18 KB a block stages per (64-channel tile, 16-channel
// K step) are contiguous and already in LDS order -- element (tap t, n, c), term s at
//     (((n/64) * (C/16) + c/16) * 18 + (t * 2 + (n%64)/32) * 3 + s) * 512 + ((c%16 / 8) * 32 + n%32) * 8 + c%8
__device__ __forceinline__ void x3p_pack_put(__bf16* pk, int t, int n, int c, int N, int C, float v) {
  const __bf16 h = (__bf16)v;
  const float r1 = v - (float)h;
  const __bf16 m = (__bf16)r1;
  const __bf16 l = (__bf16)(r1 - (float)m);
  __bf16* d = pk + ((((size_t)(n >> 6) * (C >> 4) + (c >> 4)) * 18 + (t * 2 + ((n & 63) >> 5)) * 3) * 64 + ((c & 15) >> 3) * 32 + (n & 31)) * 8 + (c & 7);
  d[0] = h; d[512] = m; d[1024] = l;
}

// The same repack through LDS, 32 co x 32 ci per block: the torch rows (ci, k contiguous) are read coalesced, every
// pack is written with its own fastest index across the lanes (the element-wise form above writes the [ci][co]
// packs with a stride of Co floats between lanes: 55 us per step, this one 15).  Needs Co % 32 == 0, Ci % 32 == 0.
__global__ __launch_bounds__(256) void repack_tiled_kernel(RepackTable t) {
  const RepackDesc& d = t.d[blockIdx.y];
  const int K = d.K, tci_n = d.Ci >> 5;
  const int tiles = (d.Co >> 5) * tci_n;
  if ((int)blockIdx.x >= tiles) return;
  __shared__ float s[32][97];                                   // [co][ci*K + k], odd pitch: conflict-free transposed reads
  const int co0 = ((int)blockIdx.x / tci_n) << 5, ci0 = ((int)blockIdx.x % tci_n) << 5;
  const int rowlen = 32 * K;
  for (int idx = threadIdx.x; idx < 32 * rowlen; idx += 256) {
    const int r = idx / rowlen, c = idx - r * rowlen;
    s[r][c] = d.W[((size_t)(co0 + r) * d.Ci + ci0) * K + c];
  }
  __syncthreads();
  const size_t tot = (size_t)d.Co * d.Ci;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = p * 256 + threadIdx.x;
    const int a = idx >> 5, b = idx & 31;                       // b runs across the lanes
    // forward-side packs: co = a, ci = b (ci contiguous)
    {
      const float* w = &s[a][b * K];
      const size_t o = (size_t)(co0 + a) * d.Ci + ci0 + b;
      if (d.Wf)
        for (int k = 0; k < K; ++k) d.Wf[(size_t)k * tot + o] = w[k];
      if (d.Uf) {
        if (d.points == 16) {                                     // bf16 tap packs of conv_bf16.hip: [K][Co][Ci]
          __bf16* u = reinterpret_cast<__bf16*>(d.Uf);
          for (int k = 0; k < K; ++k) u[(size_t)k * tot + o] = (__bf16)w[k];
        } else if (d.points == 49) {                              // chunked split-bf16 packs of conv_x3p.hip (K == 3)
          if (K == 1) x3p_pack_put(reinterpret_cast<__bf16*>(d.Uf), 1, co0 + a, ci0 + b, d.Co, d.Ci, w[0]);   // 1x1: tap 1 only (conv_x3p_s2)
          else for (int k = 0; k < 3; ++k) x3p_pack_put(reinterpret_cast<__bf16*>(d.Uf), k, co0 + a, ci0 + b, d.Co, d.Ci, w[k]);
        } else {
          emit_wino_taps(d.Uf + o, tot, w[0], w[1], w[2], d.points);
        }
      }
    }
    // data-gradient packs: ci = a, co = b (co contiguous)
    {
      const float* w = &s[b][a * K];
      const size_t o = (size_t)(ci0 + a) * d.Co + co0 + b;
      if (d.Wd)
        for (int k = 0; k < K; ++k) d.Wd[(size_t)k * tot + o] = w[k];
      if (d.Ud) {                                               // taps reversed: g_t = w[..][2 - t]
        if (d.points == 16) {                                     // [K][Ci][Co] bf16, taps reversed
          __bf16* u = reinterpret_cast<__bf16*>(d.Ud);
          for (int k = 0; k < K; ++k) u[(size_t)k * tot + o] = (__bf16)w[K - 1 - k];
        } else if (d.points == 49) {
          if (K == 1) x3p_pack_put(reinterpret_cast<__bf16*>(d.Ud), 1, ci0 + a, co0 + b, d.Ci, d.Co, w[0]);
          else for (int k = 0; k < 3; ++k) x3p_pack_put(reinterpret_cast<__bf16*>(d.Ud), k, ci0 + a, co0 + b, d.Ci, d.Co, w[2 - k]);
        } else {
          emit_wino_taps(d.Ud + o, tot, w[2], w[1], w[0], d.points);
        }
      }
    }
  }
}

// out[pos][0:C1] = a[pos][0:C1]; out[pos][C1:C1+C2] = b[pos][0:C2]
__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, int lda, int C1,
                                                      const float* __restrict__ b, int ldb, int C2,
                                                      float* __restrict__ out, int ldo, size_t npos) {
  const int nq = (C1 + C2) >> 2;
  size_t total = npos * nq;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    int q = (int)(idx % nq);
    size_t pos = idx / nq;
    int c = q * 4;
    f32x4 v = c < C1 ? *reinterpret_cast<const f32x4*>(a + pos * lda + c)
                     : *reinterpret_cast<const f32x4*>(b + pos * ldb + (c - C1));
    *reinterpret_cast<f32x4*>(out + pos * ldo + c) = v;
  }
}

// strided channel-slice copy / add: dst[pos][0:C] (+)= src[pos][off:off+C]
__global__ __launch_bounds__(256) void slice_copy_kernel(const float* __restrict__ src, int lds, int off,
                                                         float* __restrict__ dst, int ldd, int C, size_t npos,
                                                         int accumulate) {
  const int nq = C >> 2;
  size_t total = npos * nq;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    int q = (int)(idx % nq);
    size_t pos = idx / nq;
    f32x4 v = *reinterpret_cast<const f32x4*>(src + pos * lds + off + q * 4);
    float* d = dst + pos * ldd + q * 4;
    if (accumulate) {
      f32x4 o = *reinterpret_cast<const f32x4*>(d);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += o[e];
    }
    *reinterpret_cast<f32x4*>(d) = v;
  }
}

// counter-based keep mask: keep iff hash(seed, idx) >= p * 2^32; y = x * keep / (1-p).  The mask is
// regenerated from (seed, idx) in backward, nothing is stored.
__device__ __forceinline__ uint32_t mix32(uint32_t a, uint32_t b) {
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                      const int64_t* __restrict__ seed_ptr, uint32_t salt, float p) {
  const uint32_t seed = (uint32_t)seed_ptr[0] ^ (uint32_t)(seed_ptr[0] >> 32);
  const uint32_t thr = (uint32_t)(p * 4294967296.0);
  const float scale = 1.0f / (1.0f - p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = mix32(mix32(seed, salt), (uint32_t)i ^ (uint32_t)(i >> 32) * 0x27d4eb2fu);
    y[i] = h >= thr ? x[i] * scale : 0.f;
  }
}

__device__ __forceinline__ float drop_keep(uint32_t key, size_t i, uint32_t thr, float scale, float v) {
  const uint32_t h = mix32(key, (uint32_t)i ^ (uint32_t)(i >> 32) * 0x27d4eb2fu);      // == dropout_kernel's mask of element i
  return h >= thr ? v * scale : 0.f;
}

// out[pos][0:C1] = a[pos][0:C1]; out[pos][C1:C1+C2] = dropout(b)[pos][0:C2]   (_DenseLayer: dropout then torch.cat,
// reference models/densenet.py:38-41) -- the mask is that of da_dropout on the contiguous [npos][C2] tensor b
__global__ __launch_bounds__(256) void concat2_dropout_kernel(const float* __restrict__ a, int lda, int C1,
                                                              const float* __restrict__ b, int ldb, int C2,
                                                              float* __restrict__ out, int ldo, size_t npos,
                                                              const int64_t* __restrict__ seed_ptr, uint32_t salt, float p) {
  const uint32_t seed = (uint32_t)seed_ptr[0] ^ (uint32_t)(seed_ptr[0] >> 32);
  const uint32_t key = mix32(seed, salt), thr = (uint32_t)(p * 4294967296.0);
  const float scale = 1.0f / (1.0f - p);
  const int nq = (C1 + C2) >> 2;
  const size_t total = npos * nq;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % nq);
    const size_t pos = idx / nq;
    const int c = q * 4;
    f32x4 v;
    if (c < C1) {
      v = *reinterpret_cast<const f32x4*>(a + pos * lda + c);
    } else {
      v = *reinterpret_cast<const f32x4*>(b + pos * ldb + (c - C1));
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = drop_keep(key, pos * C2 + (c - C1) + e, thr, scale, v[e]);
    }
    *reinterpret_cast<f32x4*>(out + pos * ldo + c) = v;
  }
}

// dst[pos][0:C] = dropout(src[pos][off:off+C]) with the mask of da_dropout on the contiguous [npos][C] tensor
// (backward of the above: the slice of the concatenated gradient that belongs to the new features)
__global__ __launch_bounds__(256) void slice_dropout_kernel(const float* __restrict__ src, int lds, int off,
                                                            float* __restrict__ dst, int ldd, int C, size_t npos,
                                                            const int64_t* __restrict__ seed_ptr, uint32_t salt, float p) {
  const uint32_t seed = (uint32_t)seed_ptr[0] ^ (uint32_t)(seed_ptr[0] >> 32);
  const uint32_t key = mix32(seed, salt), thr = (uint32_t)(p * 4294967296.0);
  const float scale = 1.0f / (1.0f - p);
  const int nq = C >> 2;
  const size_t total = npos * nq;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % nq);
    const size_t pos = idx / nq;
    f32x4 v = *reinterpret_cast<const f32x4*>(src + pos * lds + off + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = drop_keep(key, pos * C + q * 4 + e, thr, scale, v[e]);
    *reinterpret_cast<f32x4*>(dst + pos * ldd + q * 4) = v;
  }
}

// Device-resident window store: out[b][:] = float((tiles[idx[b]][:] - mu) / std), the
// reference's ARDSRawDataset.__getitem__ normalisation (dataset.py:1364,1379: float64 arithmetic) followed by
// the .float() cast of train_ards_detector.py:150-152 -- fused with the batch gather, bit-identical results.
__global__ __launch_bounds__(256) void gather_normalize_kernel(const double* __restrict__ tiles,
                                                               const int64_t* __restrict__ idx, double mu, double stdv,
                                                               float* __restrict__ out, int tile_elems) {
  const int b = blockIdx.x;
  const double* src = tiles + (size_t)idx[b] * tile_elems;
  float* dst = out + (size_t)b * tile_elems;
  for (int i = threadIdx.x; i < tile_elems; i += blockDim.x) dst[i] = (float)((src[i] - mu) / stdv);
}

// the same for windows with C <= 4 channels (flow + the Re / Im channels of its spectrum, dataset.py:1330-1341): every
// channel has its own (mu, std) -- the (NB, C, L) broadcast arrays of dataset.py:641,648 -- tile layout [NB][C][L]
struct ChanFactors {
  double mu[4];
  double stdv[4];
};
__global__ __launch_bounds__(256) void gather_normalize_ch_kernel(const double* __restrict__ tiles,
                                                                  const int64_t* __restrict__ idx, ChanFactors f,
                                                                  float* __restrict__ out, int tile_elems, int C, int L) {
  const int b = blockIdx.x;
  const double* src = tiles + (size_t)idx[b] * tile_elems;
  float* dst = out + (size_t)b * tile_elems;
  for (int i = threadIdx.x; i < tile_elems; i += blockDim.x) {
    const int c = (i / L) % C;
    dst[i] = (float)((src[i] - f.mu[c]) / f.stdv[c]);
  }
}

// one-hot targets gathered the same way (float32 [N][2] -> [B][2])
__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ out,
                                   int B, int width) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * width) out[i] = src[(size_t)idx[i / width] * width + i % width];
}

// Test-epoch reduction on the device: pred[b] = argmax(logits[b]) (first maximum, as
// torch.argmax: class 0 on a tie) and votes[group[b]][pred[b]] += 1 -- the per-patient vote table the reference
// builds on the host (train_ards_detector.py:932-936, metrics.py:572-604).  Integer atomics: exact, order-free.
__global__ void vote_kernel(const float* __restrict__ logits, const int64_t* __restrict__ group, int B, int n_groups,
                            int* __restrict__ votes, int* __restrict__ pred) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int p = logits[2 * b + 1] > logits[2 * b] ? 1 : 0;
  if (pred) pred[b] = p;
  int64_t g = group[b];
  if (g >= 0 && g < n_groups) atomicAdd(&votes[2 * g + p], 1);
}

static inline int grid_for(size_t total, int bs, int cap) {
  size_t g = (total + bs - 1) / bs;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// Lower median over the NB breath rows of every window (torch.median(outputs, dim=1)[0] of CNNLinearComprToRF,
// reference models/torch_cnn_linear_network.py:47): out[b][f] = the ((NB-1)/2)-th smallest of feat[b*NB + r][f],
// idx[b][f] = its row r (ties: the earliest row, a stable order).  One thread per (window, feature), NB <= 64.
__global__ __launch_bounds__(256) void window_median_fwd_kernel(const float* __restrict__ x, int ld, int B, int NB, int F,
                                                               float* __restrict__ out, int* __restrict__ idx) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * F) return;
  const int b = t / F, f = t - b * F;
  const float* col = x + (size_t)b * NB * ld + f;
  float v[64];
  for (int r = 0; r < NB; ++r) v[r] = col[(size_t)r * ld];
  const int k = (NB - 1) >> 1;
  int sel = 0;
  for (int i = 0; i < NB; ++i) {
    int rank = 0;
    for (int j = 0; j < NB; ++j) rank += (v[j] < v[i]) || (v[j] == v[i] && j < i);
    if (rank == k) sel = i;
  }
  out[t] = v[sel];
  idx[t] = sel;
}

// dfeat[b*NB + r][f] = (r == idx[b][f]) ? dout[b][f] : 0
__global__ __launch_bounds__(256) void window_median_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ idx,
                                                               int B, int NB, int F, float* __restrict__ dx, int ld) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * NB * F) return;
  const int f = (int)(t % F);
  const size_t row = t / F;
  const int b = (int)(row / NB), r = (int)(row - (size_t)b * NB);
  dx[row * ld + f] = idx[(size_t)b * F + f] == r ? dout[(size_t)b * F + f] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// LSTM head (CNNLSTMNetwork, reference models/torch_cnn_lstm_combo.py:6-50: nn.LSTM(F, H, 1 layer, batch_first) over
// the NB breath features of a window).  The input projection x W_ih^T for all time steps and the weight / input
// gradients are 1x1-conv GEMMs (da_conv_gemm / da_conv_wgrad); these two kernels run the recurrence: one block per
// window, one thread per gate unit (4H <= 1024), T steps inside the kernel.  Gate order i, f, g, o.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(1024) void lstm_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                        const float* __restrict__ bih, const float* __restrict__ bhh,
                                                        const float* __restrict__ h0, const float* __restrict__ c0,
                                                        float* __restrict__ hs, float* __restrict__ cs,
                                                        float* __restrict__ gates, float* __restrict__ hT,
                                                        float* __restrict__ cT, int T, int H) {
  __shared__ float hsm[256], zs[1024];
  const int b = blockIdx.x, j = threadIdx.x, G = 4 * H;
  float c = 0.f;
  if (j < H) {
    hsm[j] = h0 ? h0[(size_t)b * H + j] : 0.f;
    c = c0 ? c0[(size_t)b * H + j] : 0.f;
  }
  const float bias = bih[j] + bhh[j];
  const float* wrow = whh + (size_t)j * H;
  const bool is_g = j >= 2 * H && j < 3 * H;
  for (int t = 0; t < T; ++t) {
    __syncthreads();
    float z = gx[((size_t)b * T + t) * G + j] + bias;
    for (int k = 0; k < H; ++k) z = fmaf(wrow[k], hsm[k], z);
    const float a = is_g ? tanhf(z) : sigmoidf_(z);
    zs[j] = a;
    gates[((size_t)b * T + t) * G + j] = a;
    __syncthreads();
    if (j < H) {
      c = zs[H + j] * c + zs[j] * zs[2 * H + j];
      const float h = zs[3 * H + j] * tanhf(c);
      hsm[j] = h;
      hs[((size_t)b * T + t) * H + j] = h;
      cs[((size_t)b * T + t) * H + j] = c;
    }
  }
  if (j < H) {
    hT[(size_t)b * H + j] = hsm[j];
    cT[(size_t)b * H + j] = c;
  }
}

// backward through time: dgates [B][T][4H] (pre-activation gradients) and this window's dW_hh [4H][H] (summed over
// the windows afterwards, fixed order).  No gradient into the initial state (the reference detaches it).  H <= 64.
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ dh_all, const float* __restrict__ whh,
                                                       const float* __restrict__ hs, const float* __restrict__ cs,
                                                       const float* __restrict__ gates, const float* __restrict__ h0,
                                                       const float* __restrict__ c0, float* __restrict__ dgates,
                                                       float* __restrict__ dwhh_part, int T, int H) {
  __shared__ float dz[256], dhrec[64], hprev[64];
  const int b = blockIdx.x, j = threadIdx.x, G = 4 * H;
  float acc[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) acc[k] = 0.f;
  float dc_next = 0.f;
  if (j < H) dhrec[j] = 0.f;
  for (int t = T - 1; t >= 0; --t) {
    __syncthreads();
    const size_t bt = (size_t)b * T + t;
    if (j < H) {
      const float* ga = gates + bt * G;
      const float i = ga[j], f = ga[H + j], g = ga[2 * H + j], o = ga[3 * H + j];
      const float c = cs[bt * H + j], tc = tanhf(c);
      const float c_prev = t > 0 ? cs[(bt - 1) * H + j] : (c0 ? c0[(size_t)b * H + j] : 0.f);
      const float dh = dh_all[bt * H + j] + dhrec[j];
      const float dc = dh * o * (1.f - tc * tc) + dc_next;
      dz[j] = dc * g * i * (1.f - i);
      dz[H + j] = dc * c_prev * f * (1.f - f);
      dz[2 * H + j] = dc * i * (1.f - g * g);
      dz[3 * H + j] = dh * tc * o * (1.f - o);
      dc_next = dc * f;
      hprev[j] = t > 0 ? hs[(bt - 1) * H + j] : (h0 ? h0[(size_t)b * H + j] : 0.f);
    }
    __syncthreads();
    const float d = dz[j];
    dgates[bt * G + j] = d;
#pragma unroll
    for (int k = 0; k < 64; ++k)
      if (k < H) acc[k] = fmaf(d, hprev[k], acc[k]);
    float r = 0.f;
    if (j < H)
      for (int q = 0; q < G; ++q) r = fmaf(whh[(size_t)q * H + j], dz[q], r);
    __syncthreads();
    if (j < H) dhrec[j] = r;
  }
  for (int k = 0; k < H; ++k) dwhh_part[((size_t)b * G + j) * H + k] = acc[k];
}

// out[i] (+)= sum over the rows r of m[r][i] (fixed order): bias gradients and the fold of per-window partials
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ m, int rows, int n, float* __restrict__ out,
                                                          int accumulate) {
  __shared__ float red[32][8];
  const int o = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + o;
  float s = 0.f;
  if (i < n)
    for (int r = slot; r < rows; r += 32) s += m[(size_t)r * n + i];
  red[slot][o] = s;
  __syncthreads();
  if (threadIdx.x < 8 && i < n) {
    s = 0.f;
    for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
    out[i] = accumulate ? out[i] + s : s;
  }
}

int g_act_bf16 = 0;       // da_set_act_dtype

extern "C" {

int da_version(void) { return 200; }

// Activation storage type of every RLC activation / activation-gradient tensor that crosses the ABI from here on:
// 0 = float (default), 1 = bf16 (BASELINE's bf16 configs; needs the bf16-operand conv kernels).  Process-wide and
// not thread-safe: set it once, before the step is built (a captured graph keeps the kernels it was captured with).
// Entry points whose kernels exist only for float activations return DA_EINVAL while it is 1.
int da_set_act_dtype(int bf16) {
  if (bf16 != 0 && bf16 != 1) return DA_EINVAL;
  g_act_bf16 = bf16;
  return DA_OK;
}
int da_get_act_dtype(void) { return g_act_bf16; }



// Address of the HIP runtime entry point this library is bound to: the Python loader compares it with
// the runtime PyTorch uses, because two HIP runtimes in one process do not share streams or ordering.
const void* da_hip_runtime_symbol(void) { return (const void*)&hipGetLastError; }

// flat: [B][K] (K = NB*F, K % 4 == 0).  W: [2][K], bias: [2].  logits: [B][2].
int da_linear2_fwd(const float* flat, const float* W, const float* bias, float* logits, int B, int K,
                   hipStream_t stream) {
  DA_ENTER();
  if (!flat || !W || !bias || !logits || K % 4) return DA_EINVAL;
  if (B == 0) return DA_OK;
  hipLaunchKernelGGL(linear2_fwd_kernel, dim3(B), dim3(256), 0, stream, flat, W, bias, logits, K);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// loss: 1 float.  dlogits may be null (test epoch).  gscale multiplies the gradient (1 normally).
int da_bce_logits(const float* logits, const float* target, int n, float gscale, float* loss, float* dlogits,
                  hipStream_t stream) {
  DA_ENTER();
  if (!logits || !target || !loss || n < 1) return DA_EINVAL;
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, stream, logits, target, n, gscale, loss, dlogits);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_linear2_bwd(const float* dlogits, const float* flat, const float* W, float* dflat, float* dW, float* dbias,
                   int B, int K, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (!dlogits || !flat || !W || K % 4) return DA_EINVAL;
  if (B == 0) return DA_OK;
  if (dflat) {
    size_t total = (size_t)B * (K / 4);
    hipLaunchKernelGGL(linear2_bwd_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dlogits,
                       W, dflat, B, K);
    DA_CHECK_LAUNCH();
  }
  if (dW && dbias) {
    hipLaunchKernelGGL(linear2_bwd_weight_kernel, dim3((K / 4 + 15) / 16), dim3(256), 0, stream, dlogits, flat, dW,
                       dbias, B, K, accumulate);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

// The head of CNNLinearNetwork, forward: x [B * R][L][ldx] (the breath block's last map, activation storage type) -> flat
// [B][R * F] (global average pool) and part [B][G][2], the shares of the two dot products of G = da_head_groups(R, F) blocks of 256 (row, 4 features) items a window.
// finish != 0 (forward-only callers): also logits [B][2] and loss[0] (one more small launch); training callers continue
// with da_head_bwd, which derives everything from `part`.  replaces AvgPool1d(7,1) + view + linear_final (+ BCEWithLogitsLoss)
int da_head_groups(int R, int F) { return (R * (F / 4) + 255) / 256; }

int da_head_fwd(const void* x, int ldx, const float* W, const float* bias, const float* target, float* flat, float* part,
                float* logits, float* loss, int B, int R, int L, int F, int finish, hipStream_t stream) {
  DA_ENTER();
  if (!x || !W || !bias || !target || !flat || !part || F % 4 || ldx % 4 || R < 1 || L < 1 || (finish && (!logits || !loss)))
    return DA_EINVAL;
  if (B == 0) return DA_OK;
  const int G = da_head_groups(R, F);
  DA_ACT_DISPATCH(hipLaunchKernelGGL(head_pool_dot_kernel<AT>, dim3(B, G), dim3(256), 0, stream, (const AT*)x, ldx, W, flat, part,
                                     R, G, L, F));
  DA_CHECK_LAUNCH();
  if (finish) {
    hipLaunchKernelGGL(head_finish_kernel, dim3(1), dim3(256), 0, stream, part, bias, target, logits, loss, B, G,
                       1.0f / (2.0f * (float)B));
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

// ... and the loss + backward: logits / dlogits [B][2], terms [B] from `part`; dx [B * R][L][lddx] (linear backward + pool
// backward in one kernel); dW / dbias (+)= from dlogits and flat; loss[0] = mean of the terms.  gscale multiplies dlogits.
int da_head_bwd(const float* part, const float* bias, const float* target, const float* flat, const float* W, void* dx, int lddx,
                float* logits, float* dlogits, float* terms, float* dW, float* dbias, float* loss, int B, int R, int L, int F,
                float gscale, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (!part || !bias || !target || !flat || !W || !dx || !logits || !dlogits || !terms || !dW || !dbias || !loss || F % 4 ||
      lddx % 4 || R < 1 || L < 1)
    return DA_EINVAL;
  if (B == 0) return DA_OK;
  const int K = R * F, G = da_head_groups(R, F);
  const float inv_n = 1.0f / (2.0f * (float)B);
  const int wblocks = (K / 4 + 15) / 16;
  const bool one = B <= HEAD_MAXB;                      // the weight gradient as further blocks of the same launch
  const int wrows = one ? (wblocks + B - 1) / B : 0;
  DA_ACT_DISPATCH(hipLaunchKernelGGL(head_bwd_kernel<AT>, dim3(B, G + wrows), dim3(256), 0, stream, part, bias, target, W, (AT*)dx,
                                     lddx, logits, dlogits, terms, R, G, L, F, inv_n, gscale, flat, dW, dbias, loss, accumulate,
                                     one ? wblocks : 0));
  DA_CHECK_LAUNCH();
  if (!one) {
    hipLaunchKernelGGL(linear2_bwd_weight_kernel, dim3(wblocks), dim3(256), 0, stream, dlogits, flat, dW, dbias, B, K, accumulate,
                       terms, inv_n, loss);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

// The same two calls on features that are pooled already (da_bn_fwd_pool wrote them): flat_in [B * R][F] float takes the place
// of the map (a "map" of ONE position: the pool is the identity, bit for bit), dflat [B * R][F] float that of dx -- whatever
// the activation storage type.
int da_head_flat_fwd(const float* flat_in, const float* W, const float* bias, const float* target, float* flat, float* part,
                     float* logits, float* loss, int B, int R, int F, int finish, hipStream_t stream) {
  DA_ENTER();
  if (!flat_in || !W || !bias || !target || !flat || !part || F % 4 || R < 1 || (finish && (!logits || !loss))) return DA_EINVAL;
  if (B == 0) return DA_OK;
  const int G = da_head_groups(R, F);
  hipLaunchKernelGGL(head_pool_dot_kernel<float>, dim3(B, G), dim3(256), 0, stream, flat_in, F, W, flat, part, R, G, 1, F);
  DA_CHECK_LAUNCH();
  if (finish) {
    hipLaunchKernelGGL(head_finish_kernel, dim3(1), dim3(256), 0, stream, part, bias, target, logits, loss, B, G,
                       1.0f / (2.0f * (float)B));
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

int da_head_flat_bwd(const float* part, const float* bias, const float* target, const float* flat, const float* W, float* dflat,
                     float* logits, float* dlogits, float* terms, float* dW, float* dbias, float* loss, int B, int R, int F,
                     float gscale, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (!part || !bias || !target || !flat || !W || !dflat || !logits || !dlogits || !terms || !dW || !dbias || !loss || F % 4 ||
      R < 1)
    return DA_EINVAL;
  if (B == 0) return DA_OK;
  const int K = R * F, G = da_head_groups(R, F);
  const float inv_n = 1.0f / (2.0f * (float)B);
  const int wblocks = (K / 4 + 15) / 16;
  const bool one = B <= HEAD_MAXB;
  const int wrows = one ? (wblocks + B - 1) / B : 0;
  hipLaunchKernelGGL(head_bwd_kernel<float>, dim3(B, G + wrows), dim3(256), 0, stream, part, bias, target, W, dflat, F, logits,
                     dlogits, terms, R, G, 1, F, inv_n, gscale, flat, dW, dbias, loss, accumulate, one ? wblocks : 0);
  DA_CHECK_LAUNCH();
  if (!one) {
    hipLaunchKernelGGL(linear2_bwd_weight_kernel, dim3(wblocks), dim3(256), 0, stream, dlogits, flat, dW, dbias, B, K, accumulate,
                       terms, inv_n, loss);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

int da_clamp_sgd_nesterov(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay,
                          float clip, float gscale, int first, hipStream_t stream) {
  DA_ENTER();
  if (!p || !g || !buf) return DA_EINVAL;
  if (n == 0) return DA_OK;
  hipLaunchKernelGGL(clamp_sgd_nesterov_kernel, dim3(grid_for((n + 3) / 4, 256, 4096)), dim3(256), 0, stream, p, g, buf, n, lr,
                     momentum, weight_decay, clip, gscale, first);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_clamp_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                  int step, float clip, float gscale, hipStream_t stream) {
  DA_ENTER();
  if (!p || !g || !m || !v || step < 1) return DA_EINVAL;
  if (n == 0) return DA_OK;
  float bc1 = 1.0f - powf(beta1, (float)step);
  float bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(clamp_adam_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, bc1, bc2, clip, gscale);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// da_clamp_adam with the step count in device memory (int64, starts at 0): incremented here, then used -- the whole
// update is stream-ordered device work, so a captured graph can replay it (torch.optim.Adam, train_ards_detector.py:421).
int da_clamp_adam_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                      long long* step, float clip, float gscale, hipStream_t stream) {
  DA_ENTER();
  if (!p || !g || !m || !v || !step) return DA_EINVAL;
  hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(64), 0, stream, step);
  DA_CHECK_LAUNCH();
  if (n == 0) return DA_OK;
  hipLaunchKernelGGL(clamp_adam_dev_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, step, clip, gscale);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// W: [Co][Ci][K] torch layout; Wf: [K][Co][Ci]; Wd: [K][Ci][Co] (either may be null).
int da_repack_conv_weight(const float* W, float* Wf, float* Wd, int Co, int Ci, int K, hipStream_t stream) {
  DA_ENTER();
  if (!W || (!Wf && !Wd)) return DA_EINVAL;
  size_t total = (size_t)Co * Ci * K;
  hipLaunchKernelGGL(repack_conv_weight_kernel, dim3(grid_for(total, 256, 2048)), dim3(256), 0, stream, W, Wf, Wd, Co,
                     Ci, K);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// tiles: [N][tile_elems] float64 raw windows; idx: [B] int64 window indices; out: [B][tile_elems] float32.
int da_gather_normalize(const double* tiles, const int64_t* idx, double mu, double stdv, float* out, int B,
                        int tile_elems, hipStream_t stream) {
  DA_ENTER();
  if (!tiles || !idx || !out || tile_elems < 1 || stdv == 0.0) return DA_EINVAL;
  if (B == 0) return DA_OK;
  hipLaunchKernelGGL(gather_normalize_kernel, dim3(B), dim3(256), 0, stream, tiles, idx, mu, stdv, out, tile_elems);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// tiles: [N][NB][C][L] float64, C <= 4; mu / stdv: HOST arrays of C per-channel factors.
int da_gather_normalize_ch(const double* tiles, const int64_t* idx, const double* mu, const double* stdv, float* out, int B,
                           int NB, int C, int L, hipStream_t stream) {
  DA_ENTER();
  if (!tiles || !idx || !out || !mu || !stdv || NB < 1 || C < 1 || C > 4 || L < 1) return DA_EINVAL;
  ChanFactors f;
  for (int c = 0; c < 4; ++c) {
    f.mu[c] = c < C ? mu[c] : 0.0;
    f.stdv[c] = c < C ? stdv[c] : 1.0;
    if (f.stdv[c] == 0.0) return DA_EINVAL;
  }
  if (B == 0) return DA_OK;
  hipLaunchKernelGGL(gather_normalize_ch_kernel, dim3(B), dim3(256), 0, stream, tiles, idx, f, out, NB * C * L, C, L);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// votes: [n_groups][2] int32, accumulated across calls (zero it at the start of the epoch); pred: [B] int32 or NULL.
int da_vote_counts(const float* logits, const int64_t* group, int B, int n_groups, int* votes, int* pred,
                   hipStream_t stream) {
  DA_ENTER();
  if (!logits || !group || !votes || n_groups < 1) return DA_EINVAL;
  if (B == 0) return DA_OK;
  hipLaunchKernelGGL(vote_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, logits, group, B, n_groups, votes, pred);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// CNNLinearComprToRF head: lower median over the NB rows of each of the B windows (torch_cnn_linear_network.py:47).
// x: [B*NB][ld] (F features used), out: [B][F], idx: [B][F] selected row (kept for the backward).  NB <= 64.
int da_window_median_fwd(const float* x, int ld, int B, int NB, int F, float* out, int* idx, hipStream_t stream) {
  DA_ENTER();
  if (!x || !out || !idx || NB < 1 || NB > 64 || F < 1 || ld < F) return DA_EINVAL;
  if (B == 0) return DA_OK;
  hipLaunchKernelGGL(window_median_fwd_kernel, dim3((B * F + 255) / 256), dim3(256), 0, stream, x, ld, B, NB, F, out, idx);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// dx [B*NB][ld] = the median's gradient: dout[b][f] at row idx[b][f] of window b, 0 elsewhere.
int da_window_median_bwd(const float* dout, const int* idx, int B, int NB, int F, float* dx, int ld, hipStream_t stream) {
  DA_ENTER();
  if (!dout || !idx || !dx || NB < 1 || NB > 64 || F < 1 || ld < F) return DA_EINVAL;
  if (B == 0) return DA_OK;
  const size_t total = (size_t)B * NB * F;
  hipLaunchKernelGGL(window_median_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dout, idx, B,
                     NB, F, dx, ld);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// LSTM recurrence of CNNLSTMNetwork (torch_cnn_lstm_combo.py:17,46): gx [B][T][4H] = x W_ih^T (da_conv_gemm), whh [4H][H],
// bih / bhh [4H], h0 / c0 [B][H] or NULL (zeros).  Outputs hs, cs [B][T][H], gates [B][T][4H] (activated i,f,g,o; kept
// for the backward), hT, cT [B][H].  H % 8 == 0, H <= 256.
int da_lstm_fwd(const float* gx, const float* whh, const float* bih, const float* bhh, const float* h0, const float* c0,
                float* hs, float* cs, float* gates, float* hT, float* cT, int B, int T, int H, hipStream_t stream) {
  DA_ENTER();
  if (!gx || !whh || !bih || !bhh || !hs || !cs || !gates || !hT || !cT || T < 1 || H < 8 || H % 8 || H > 256)
    return DA_EINVAL;
  if (B == 0) return DA_OK;
  hipLaunchKernelGGL(lstm_fwd_kernel, dim3(B), dim3(4 * H), 0, stream, gx, whh, bih, bhh, h0, c0, hs, cs, gates, hT, cT, T, H);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// dh_all [B][T][H] -> dgates [B][T][4H] and dwhh_part [B][4H][H] (fold with da_reduce_rows).  H <= 64.
int da_lstm_bwd(const float* dh_all, const float* whh, const float* hs, const float* cs, const float* gates,
                const float* h0, const float* c0, float* dgates, float* dwhh_part, int B, int T, int H,
                hipStream_t stream) {
  DA_ENTER();
  if (!dh_all || !whh || !hs || !cs || !gates || !dgates || !dwhh_part || T < 1 || H < 8 || H % 8 || H > 64) return DA_EINVAL;
  if (B == 0) return DA_OK;
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(B), dim3(4 * H), 0, stream, dh_all, whh, hs, cs, gates, h0, c0, dgates, dwhh_part,
                     T, H);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// out[n] (+)= column sums of m [rows][n], fixed order.
int da_reduce_rows(const float* m, int rows, int n, float* out, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (!m || !out || rows < 0 || n < 1) return DA_EINVAL;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((n + 7) / 8), dim3(256), 0, stream, m, rows, n, out, accumulate);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_gather_rows(const float* src, const int64_t* idx, float* out, int B, int width, hipStream_t stream) {
  DA_ENTER();
  if (!src || !idx || !out || width < 1) return DA_EINVAL;
  if (B == 0) return DA_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((B * width + 255) / 256), dim3(256), 0, stream, src, idx, out, B, width);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

typedef struct {
  const float* W;
  float* Wf;
  float* Wd;
  float* Uf;
  float* Ud;
  int Co, Ci, K;
  int points;
} da_repack_desc;

// repack n conv weights (descs: HOST array) with one launch per 32.
int da_repack_multi(const da_repack_desc* descs, int n, hipStream_t stream) {
  DA_ENTER();
  if (n < 0 || (n && !descs)) return DA_EINVAL;
  for (int base = 0; base < n; base += 32) {
    RepackTable t;
    int m = n - base < 32 ? n - base : 32;
    for (int i = 0; i < m; ++i) {
      const da_repack_desc& s = descs[base + i];
      if (!s.W || (!s.Wf && !s.Wd && !s.Uf && !s.Ud) || ((s.Uf || s.Ud) && s.K != 3 && s.points != 16 && s.points != 49)) return DA_EINVAL;
      if ((s.points == 16 || s.points == 49) && (s.Co % 32 || s.Ci % 32)) return DA_EINVAL;   // bf16 packs: tiled kernel only
      if (s.points == 49 && s.K != 3 && s.K != 1) return DA_EINVAL;
      if (s.points == 49 && (s.Co % 64 || s.Ci % 64)) return DA_EINVAL;
      t.d[i] = {s.W, s.Wf, s.Wd, s.Uf, s.Ud, s.Co, s.Ci, s.K, s.points};
    }
    bool tiled = true;
    int maxtiles = 0;
    for (int i = 0; i < m; ++i) {
      const RepackDesc& r = t.d[i];
      tiled = tiled && r.Co % 32 == 0 && r.Ci % 32 == 0 && r.K >= 1 && r.K <= 3;
      const int tl = (r.Co / 32) * (r.Ci / 32);
      if (tl > maxtiles) maxtiles = tl;
    }
    if (tiled) hipLaunchKernelGGL(repack_tiled_kernel, dim3(maxtiles, m), dim3(256), 0, stream, t);
    else hipLaunchKernelGGL(repack_multi_kernel, dim3(256, m), dim3(256), 0, stream, t);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

int da_concat2(const float* a, int lda, int C1, const float* b, int ldb, int C2, float* out, int ldo, size_t npos,
               hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (!a || !b || !out || C1 % 4 || C2 % 4 || lda % 4 || ldb % 4 || ldo % 4) return DA_EINVAL;
  if (npos == 0) return DA_OK;
  hipLaunchKernelGGL(concat2_kernel, dim3(grid_for(npos * ((C1 + C2) / 4), 256, 8192)), dim3(256), 0, stream, a, lda,
                     C1, b, ldb, C2, out, ldo, npos);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_slice_copy(const float* src, int lds, int off, float* dst, int ldd, int C, size_t npos, int accumulate,
                  hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (!src || !dst || C % 4 || lds % 4 || ldd % 4 || off % 4) return DA_EINVAL;
  if (npos == 0) return DA_OK;
  hipLaunchKernelGGL(slice_copy_kernel, dim3(grid_for(npos * (C / 4), 256, 8192)), dim3(256), 0, stream, src, lds, off,
                     dst, ldd, C, npos, accumulate);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// da_concat2 with dropout (keep-prob 1-p, mask of da_dropout on b) applied to the b half on the way.
int da_concat2_dropout(const float* a, int lda, int C1, const float* b, int ldb, int C2, float* out, int ldo, size_t npos,
                       const int64_t* seed, unsigned salt, float p, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (!a || !b || !out || !seed || C1 % 4 || C2 % 4 || lda % 4 || ldb % 4 || ldo % 4 || p < 0.f || p >= 1.f) return DA_EINVAL;
  if (npos == 0) return DA_OK;
  hipLaunchKernelGGL(concat2_dropout_kernel, dim3(grid_for(npos * ((C1 + C2) / 4), 256, 8192)), dim3(256), 0, stream, a, lda,
                     C1, b, ldb, C2, out, ldo, npos, seed, salt, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// dst = dropout(src[:, off:off+C]) with the mask of da_dropout on a contiguous [npos][C] tensor.
int da_slice_dropout(const float* src, int lds, int off, float* dst, int ldd, int C, size_t npos, const int64_t* seed,
                     unsigned salt, float p, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (!src || !dst || !seed || C % 4 || lds % 4 || ldd % 4 || off % 4 || p < 0.f || p >= 1.f) return DA_EINVAL;
  if (npos == 0) return DA_OK;
  hipLaunchKernelGGL(slice_dropout_kernel, dim3(grid_for(npos * (C / 4), 256, 8192)), dim3(256), 0, stream, src, lds, off,
                     dst, ldd, C, npos, seed, salt, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// y = dropout(x) with keep-prob 1-p; the same (seed, salt) reproduces the mask (used by backward).
int da_dropout(const float* x, float* y, size_t n, const int64_t* seed, unsigned salt, float p, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (!x || !y || !seed || p < 0.f || p >= 1.f) return DA_EINVAL;
  if (n == 0) return DA_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, stream, x, y, n, seed, salt, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// sizeof of the host-side descriptor structs as THIS library was compiled with them, in the order
// {da_wgrad_job, da_conv_job, da_wgrad_reduce_desc, da_repack_desc, da_bn_running_desc, da_bn_pgrad_desc}:
// tests compare them with the public header (compiled by gcc) and with the ctypes mirror.
void da_abi_sizes(int* out) {
  out[0] = (int)sizeof(da_wgrad_job);
  out[1] = (int)sizeof(da_conv_job);
  out[2] = da_sizeof_wgrad_reduce_desc();
  out[3] = (int)sizeof(da_repack_desc);
  out[4] = da_sizeof_bn_running_desc();
  out[5] = da_sizeof_bn_pgrad_desc();
}

}  // extern "C"
