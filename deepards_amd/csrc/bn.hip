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

static int g_bn_two_stage = 0;   // da_bn_debug_two_stage(): force the two-stage kernels (tests compare both paths)

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

// Statistics in two fully parallel stages (a window slab is up to 573 KB: one block per (window,
// 32-channel group) would leave most CUs idle on the wide early layers):
//   stage 1  grid (W, C/32, P): chunk-local mean and centred M2 over <= `chunk` positions
//   stage 2  bn_stats_merge: Chan's pairwise update over the P chunks in order -> mean, invstd
template <typename AT>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const AT* __restrict__ x, int ld, int Wn, int C,
                                                               int chunk, float* __restrict__ part) {
  __shared__ float red[(SLOTS + 1) * CG];
  const int w = row_xcd_chunk(blockIdx.x, gridDim.x), cg = blockIdx.y, pc = blockIdx.z, P = gridDim.z;
  const int q = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int p_beg = pc * chunk, p_end = min(Wn, p_beg + chunk);
  const AT* base = x + (size_t)w * Wn * ld + cg * CG + q * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, m[4];
  for (int p = p_beg + slot; p < p_end; p += SLOTS) {
    f32x4 v = Act<AT>::ld4(base + (size_t)p * ld);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += v[e];
  }
  block_fold(s, red, slot, q, m);
  const float inv_n = 1.0f / (float)(p_end - p_beg);
#pragma unroll
  for (int e = 0; e < 4; ++e) m[e] *= inv_n;
  float s2[4] = {0.f, 0.f, 0.f, 0.f}, m2[4];
  for (int p = p_beg + slot; p < p_end; p += SLOTS) {
    f32x4 v = Act<AT>::ld4(base + (size_t)p * ld);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d = v[e] - m[e];
      s2[e] += d * d;
    }
  }
  block_fold(s2, red, slot, q, m2);
  if (slot == 0) {
    float* o = part + (((size_t)w * P + pc) * 2) * C + cg * CG + q * 4;   // [w][p][{mean,M2}][C]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = m[e];
      o[C + e] = m2[e];
    }
  }
}

// The same chunk records for the DEFAULT STEM's BatchNorm without its input in memory: position p of window w is output
// (row, l) = ((w Wn + p) / Lc, (w Wn + p) % Lc) of the k7 s2 conv on the raw rows, recomputed (common.h StemW4: bit for bit
// stem_conv_fwd_kernel's value) from the chunk's rows staged in LDS -- same slots, same folds: the records equal
// bn_stats_partial_kernel's on the stored map.  XSR = floats per staged row (3 zeros in front, zeros behind).
__global__ __launch_bounds__(256) void stem_stats_partial_kernel(const float* __restrict__ xrows, const float* __restrict__ wt,
                                                                 int Lin, int Lc, int Wn, int C, int chunk, int XSR,
                                                                 float* __restrict__ part) {
  extern __shared__ float sm[];                       // xs[rows of the chunk][XSR] | red[(SLOTS + 1) * CG]
  const int w = row_xcd_chunk(blockIdx.x, gridDim.x), cg = blockIdx.y, pc = blockIdx.z, P = gridDim.z;
  const int q = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int p_beg = pc * chunk, p_end = min(Wn, p_beg + chunk);
  const size_t g0 = (size_t)w * Wn + p_beg;
  const int row0 = (int)(g0 / Lc), l_first = (int)(g0 - (size_t)row0 * Lc);
  const int nrows = (l_first + (p_end - p_beg) + Lc - 1) / Lc;
  float* xs = sm;
  float* red = sm + nrows * XSR;
  for (int i = threadIdx.x; i < nrows * XSR; i += blockDim.x) {
    const int r = i / XSR, sx = i - r * XSR - 3;
    xs[i] = (sx >= 0 && sx < Lin) ? xrows[(size_t)(row0 + r) * Lin + sx] : 0.f;
  }
  StemW4 sw;
  sw.load(wt, cg * CG + q * 4);
  __syncthreads();
  // this thread's positions: p_beg + slot, + SLOTS, ...; (r, l) walks along (Lc >= SLOTS is not assumed)
  auto y_at = [&](int r, int l) {
    const float* xr = xs + r * XSR + 2 * l;             // input 2 l - 3 sits at xs[r][2 l]
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 7; ++k) acc = fmaf(sw.w[e][k], xr[k], acc);
      y[e] = acc;
    }
    return y;
  };
  float s[4] = {0.f, 0.f, 0.f, 0.f}, m[4];
  {
    int l = l_first + slot, r = 0;
    while (l >= Lc) { l -= Lc; ++r; }
    for (int p = p_beg + slot; p < p_end; p += SLOTS) {
      const f32x4 v = y_at(r, l);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += v[e];
      l += SLOTS;
      while (l >= Lc) { l -= Lc; ++r; }
    }
  }
  block_fold(s, red, slot, q, m);
  const float inv_n = 1.0f / (float)(p_end - p_beg);
#pragma unroll
  for (int e = 0; e < 4; ++e) m[e] *= inv_n;
  float s2[4] = {0.f, 0.f, 0.f, 0.f}, m2[4];
  {
    int l = l_first + slot, r = 0;
    while (l >= Lc) { l -= Lc; ++r; }
    for (int p = p_beg + slot; p < p_end; p += SLOTS) {
      const f32x4 v = y_at(r, l);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = v[e] - m[e];
        s2[e] += d * d;
      }
      l += SLOTS;
      while (l >= Lc) { l -= Lc; ++r; }
    }
  }
  block_fold(s2, red, slot, q, m2);
  if (slot == 0) {
    float* o = part + (((size_t)w * P + pc) * 2) * C + cg * CG + q * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = m[e];
      o[C + e] = m2[e];
    }
  }
}

// merge the P chunk records of one (window, channel) in chunk order (Chan's update)
__device__ __forceinline__ void bn_merge_chunks(const float* __restrict__ pp, int P, int C, int Wn, int chunk,
                                                float eps, float& mean, float& invstd) {
  float n = (float)min(chunk, Wn), mu = pp[0], m2 = pp[C];
  for (int p = 1; p < P; ++p) {
    float nb = (float)(min(Wn, (p + 1) * chunk) - p * chunk);
    float mb = pp[(size_t)p * 2 * C], m2b = pp[(size_t)p * 2 * C + C];
    float d = mb - mu, nt = n + nb;
    mu += d * (nb / nt);
    m2 += m2b + d * d * (n * nb / nt);
    n = nt;
  }
  mean = mu;
  invstd = 1.0f / sqrtf(m2 / (float)Wn + eps);
}

// stage 2 (standalone form): one thread per (window, channel)
__global__ __launch_bounds__(256) void bn_stats_merge_kernel(const float* __restrict__ part, int W, int P, int C, int Wn,
                                                             int chunk, float eps, float* __restrict__ mean,
                                                             float* __restrict__ invstd) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= W * C) return;
  int w = idx / C, c = idx - w * C;
  float mu, is;
  bn_merge_chunks(part + (size_t)w * P * 2 * C + c, P, C, Wn, chunk, eps, mu, is);
  mean[idx] = mu;
  invstd[idx] = is;
}

// Running statistics of up to 32 BatchNorms in one launch (blockIdx.y = which BN).  The reference
// updates them once per window, in window order (SURVEY.md finding 5); closed form of the W updates:
//   r_W = (1-mom)^W r_0 + mom * sum_w (1-mom)^(W-1-w) stat_w      (unbiased variance n/(n-1))
// block = 32 channels x 8 window slots, folded through LDS in a fixed order (deterministic).
// (BnRunningDesc and the block's body: common.h -- a training step's tail launch carries these blocks too)
struct BnRunningTable {
  BnRunningDesc d[32];
};

__global__ __launch_bounds__(256) void bn_running_multi_kernel(BnRunningTable t) {
  __shared__ float red[2][8][32];
  bn_running_block(t.d[blockIdx.y], blockIdx.x, red);
}

// out = act( (x-mean)*invstd*gamma + beta (+ res) )
template <typename AT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const AT* __restrict__ x, int ldx, const AT* __restrict__ res,
                                                       int ldr, AT* __restrict__ out, int ldo, int Wn, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int relu, int chunk, const float* __restrict__ part, int P,
                                                       int schunk, float eps, float* __restrict__ mean_out,
                                                       float* __restrict__ invstd_out) {
  const int w = row_xcd_chunk(blockIdx.x, gridDim.x), cg = blockIdx.y;
  const int q = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int c0 = cg * CG + q * 4;
  f32x4 mu, is;
  if (part) {   // statistics arrive as chunk records: merge them here, publish mean/invstd once
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float m_, i_;
      bn_merge_chunks(part + (size_t)w * P * 2 * C + c0 + e, P, C, Wn, schunk, eps, m_, i_);
      mu[e] = m_;
      is[e] = i_;
    }
    if (blockIdx.z == 0 && slot == 0) {
      *reinterpret_cast<f32x4*>(mean_out + (size_t)w * C + c0) = mu;
      *reinterpret_cast<f32x4*>(invstd_out + (size_t)w * C + c0) = is;
    }
  } else {
    mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * C + c0);
    is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * C + c0);
  }
  f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
  const int p_beg = blockIdx.z * chunk;
  const int p_end = min(Wn, p_beg + chunk);
  for (int p = p_beg + slot; p < p_end; p += SLOTS) {
    size_t pos = (size_t)w * Wn + p;
    f32x4 v = Act<AT>::ld4(x + pos * ldx + c0);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[e] - mu[e]) * is[e] * ga[e] + be[e];
    if (res) {
      f32x4 r = Act<AT>::ld4(res + pos * ldr + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += r[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    Act<AT>::st4(out + pos * ldo + c0, o);
  }
}

// Backward of out = act(bn(x) (+res)), in two fully parallel stages over (window, 32 channels, chunk):
//   mask_mode 0: no ReLU            g = dout
//             1: ReLU, no residual  g = dout * [bn(x) > 0]      (mask recomputed, `out` not read)
//             2: ReLU with residual g = dout * [out > 0]
//   reduce: part[w][p][{s1,s2}][C] = sum over the chunk of g, g*xhat
//   apply : dx = gamma*invstd*(g - mean_w(g) - xhat*mean_w(g*xhat));  gout (optional) = g;
//           chunk 0 also stores the window totals ds1/ds2 for dbeta/dgamma.
template <typename AT>
__device__ __forceinline__ f32x4 bn_masked_g(f32x4 g, const f32x4& xh, const f32x4& ga, const f32x4& be, int mask_mode,
                                             const AT* __restrict__ outp, size_t off) {
  if (mask_mode == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = (xh[e] * ga[e] + be[e] > 0.f) ? g[e] : 0.f;
  } else if (mask_mode == 2) {
    f32x4 o = Act<AT>::ld4(outp + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = (o[e] > 0.f) ? g[e] : 0.f;
  }
  return g;
}

template <typename AT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const AT* __restrict__ dout, int ldd,
                                                            const AT* __restrict__ x, int ldx,
                                                            const AT* __restrict__ outp, int ldo, int Wn, int C,
                                                            int chunk, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int mask_mode,
                                                            float* __restrict__ part) {
  __shared__ float red[(SLOTS + 1) * CG];
  const int w = row_xcd_chunk(blockIdx.x, gridDim.x), cg = blockIdx.y, pc = blockIdx.z, P = gridDim.z;
  const int q = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int c0 = cg * CG + q * 4;
  f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * C + c0);
  f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * C + c0);
  f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
  const int p_beg = pc * chunk, p_end = min(Wn, p_beg + chunk);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = p_beg + slot; p < p_end; p += SLOTS) {
    size_t pos = (size_t)w * Wn + p;
    f32x4 g = Act<AT>::ld4(dout + pos * ldd + c0);
    f32x4 v = Act<AT>::ld4(x + pos * ldx + c0);
    f32x4 xh;
#pragma unroll
    for (int e = 0; e < 4; ++e) xh[e] = (v[e] - mu[e]) * is[e];
    g = bn_masked_g(g, xh, ga, be, mask_mode, outp, pos * ldo + c0);
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
    float* o = part + (((size_t)w * P + pc) * 2) * C + c0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = t1[e];
      o[C + e] = t2[e];
    }
  }
}

template <typename AT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const AT* __restrict__ dout, int ldd,
                                                           const AT* __restrict__ x, int ldx,
                                                           const AT* __restrict__ outp, int ldo,
                                                           AT* __restrict__ dx, int lddx, AT* __restrict__ gout,
                                                           int ldg, int Wn, int C, int chunk,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int mask_mode,
                                                           const float* __restrict__ part, float* __restrict__ ds1,
                                                           float* __restrict__ ds2, const AT* __restrict__ add,
                                                           int ldadd) {
  const int w = row_xcd_chunk(blockIdx.x, gridDim.x), cg = blockIdx.y, pc = blockIdx.z, P = gridDim.z;
  const int q = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int c0 = cg * CG + q * 4;
  f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * C + c0);
  f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * C + c0);
  f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
  f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < P; ++p) {                       // chunk order: deterministic totals
    const float* o = part + (((size_t)w * P + p) * 2) * C + c0;
    f32x4 a = *reinterpret_cast<const f32x4*>(o);
    f32x4 b = *reinterpret_cast<const f32x4*>(o + C);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t1[e] += a[e];
      t2[e] += b[e];
    }
  }
  if (pc == 0 && slot == 0) {
    *reinterpret_cast<f32x4*>(ds1 + (size_t)w * C + c0) = t1;
    *reinterpret_cast<f32x4*>(ds2 + (size_t)w * C + c0) = t2;
  }
  const float inv_n = 1.0f / (float)Wn;
  const int p_beg = pc * chunk, p_end = min(Wn, p_beg + chunk);
  for (int p = p_beg + slot; p < p_end; p += SLOTS) {
    size_t pos = (size_t)w * Wn + p;
    f32x4 g = Act<AT>::ld4(dout + pos * ldd + c0);
    f32x4 v = Act<AT>::ld4(x + pos * ldx + c0);
    f32x4 xh, d;
#pragma unroll
    for (int e = 0; e < 4; ++e) xh[e] = (v[e] - mu[e]) * is[e];
    g = bn_masked_g(g, xh, ga, be, mask_mode, outp, pos * ldo + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = ga[e] * is[e] * (g[e] - t1[e] * inv_n - xh[e] * t2[e] * inv_n);
    if (add) {
      const f32x4 av = Act<AT>::ld4(add + pos * ldadd + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] += av[e];
    }
    Act<AT>::st4(dx + pos * lddx + c0, d);
    if (gout) Act<AT>::st4(gout + pos * ldg + c0, g);
  }
}

// -------------------------------------------------------------------------------------------------
// Single-pass forms.  A (window, 32- or 16-channel) slab fits the registers of one block (thread = one channel
// quad x NPOS positions; Wn <= 1280 positions with 32 channels per block, <= 2560 with 16), so every tensor crosses
// HBM exactly once:
//   forward : read x (+res), write out           (two-stage path: x is read by stats AND by apply)
//   backward: read dout, x (+out), write dx (+g) (two-stage path: both are read by reduce AND apply)
// block = NQ quads x P position slots, P = ceil(Wn / NPOS) rounded to whole waves; the fold over slots is wave
// shuffles + one LDS exchange between the waves, in a fixed order (deterministic).  16 channels per block are used
// when 32 would leave CUs without a block (the 64-channel layers at B = 64: the two halves of a 128-B line are read
// by blocks of the same window, which share an XCD and its L2) or the window is too long for one block.
// Both kernels read their whole slab before the first store, so out/dx may alias an input.
// -------------------------------------------------------------------------------------------------
#define FUSED_NPOS 10

// counter-based dropout keep mask of head_optim.hip (da_dropout): element i of the contiguous [npos][G] tensor
__device__ __forceinline__ uint32_t bn_mix32(uint32_t a, uint32_t b) {
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// Options of the dense-block BatchNorm backward (da_bn_bwd_ss): mean / invstd from a pitched table; the ReLU decision in
// the scale / shift form above; the upstream gradient at HALF resolution (a transition's AvgPool1d(2,2) folded in front of
// its 1x1 conv: dh[2j] = dh[2j+1] = dpooled[j] / 2); a dropout mask applied to the LAST drop_g channels of dx on store
// (those channels are the previous layer's new features: this kernel is the last to add to their gradient, and what
// its data / weight gradient convs want is the gradient in front of F.dropout, densenet.py:37-39).
struct BnBwdExt {
  const void* dout2;   // D2: the upstream gradient is dout + dout2 (activation storage type, pitch ldd2): a residual block's
  int ldd2;            //     input gradient left as its two terms (data-gradient conv output | identity branch), summed here
  int pool_L;          // DPOOL: dout is the gradient of the POOLED features, float [rows][ldd]: position p of a window reads
  FastDiv pool_div;    //        row p / pool_L, scaled by 1 / pool_L (the backward of bn_fwd_pool_kernel's average)
  int ldstat, half_dout, drop_c0, drop_g;
  const long long* seed;
  uint32_t salt;
  float p;
  void* hout;          // mask_mode 4: the activation relu(fma(x, sc, sh)) the forward never stored, written here (activation
  int ldh;             // storage type) for the weight gradient of the conv behind it -- or NULL
};

template <int NV, int NQ>
__device__ __forceinline__ void quad_block_sum(f32x4 (&v)[NV], float* red) {
  // lanes NQ*s+q of a wave hold the same channel quad q for 64/NQ slots s
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = v[i][e];
#pragma unroll
      for (int o = NQ; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
      v[i][e] = t;
    }
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6, q = threadIdx.x & (NQ - 1);
  __syncthreads();
  if (lane < NQ) {
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(&red[(wv * NV + i) * CG + lane * 4]) = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nw; ++k) {
      f32x4 r = *reinterpret_cast<const f32x4*>(&red[(k * NV + i) * CG + q * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += r[e];
    }
    v[i] = t;
  }
}

// mean / invstd of the window (two-pass from registers), published, and out = act(bn(x) (+res))
// RX3 / OX3: the residual is read / the output is stored in the x3 format (common.h: exact three-term bf16 split, 3 C bf16
// per position; float activations only) -- the producers of the k3 s1 convs' inputs under conv arithmetic 'f32x3'.
// POOL: the output is not stored -- only its average over the L positions of every row, pool_out[row][C] (float): the
// block-output BatchNorm of the LAST residual block, whose map only the head's AvgPool1d(7) reads (resnet.py:112,159-160).
// The values pooled are the ones the activation storage type would have held, summed in position order and scaled by 1 / L
// last (head_pool_dot_kernel's arithmetic: bit for bit the pooled features of the stored map).
#define BN_POOL_MAX_WN 160
template <typename AT, int NPOS, int QB, int RX3 = 0, int OX3 = 0, int POOL = 0>
__device__ __forceinline__ void bn_fwd_fused_body(const AT* __restrict__ x, int ldx, const AT* __restrict__ res, int ldr,
                                                  AT* __restrict__ out, int ldo, int Wn, int C,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                  float eps, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                  unsigned long long* __restrict__ mask, int ldstat, float* red,
                                                  float* __restrict__ pool_out = nullptr, int pool_L = 1, float* pool_s = nullptr) {
  // ldstat: pitch of mean_out / invstd_out ([W][ldstat], >= C: a dense block keeps ONE table for its whole buffer);
  // out == nullptr: statistics only (da_bn_stats_fused)
  constexpr int NQ = 1 << QB, CGB = 4 * NQ;            // channel quads / channels per block (32 or 16)
  const int w = row_xcd_chunk(blockIdx.x, gridDim.x), cg = blockIdx.y, P = blockDim.x >> QB;
  const int q = threadIdx.x & (NQ - 1), slot = threadIdx.x >> QB;
  const int c0 = cg * CGB + q * 4;
  const size_t base = (size_t)w * Wn;
  const AT* xb = x + base * ldx + cg * CGB;      // wave-uniform bases + 32-bit lane offsets
  const AT* rb = res ? res + base * ldr + cg * CGB : nullptr;
  AT* ob = out ? out + base * ldo + cg * CGB : nullptr;
  f32x4 v[NPOS];
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    const int p = slot + k * P;
    v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p < Wn) v[k] = Act<AT>::ld4(xb + (uint32_t)(p * ldx + q * 4));
  }
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int k = 0; k < NPOS; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[0][e] += v[k][e];
  quad_block_sum<1, NQ>(acc, red);
  const float inv_n = 1.0f / (float)Wn;
  f32x4 mu;
#pragma unroll
  for (int e = 0; e < 4; ++e) mu[e] = acc[0][e] * inv_n;
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    if (slot + k * P < Wn) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = v[k][e] - mu[e];
        acc[0][e] += d * d;
      }
    }
  }
  quad_block_sum<1, NQ>(acc, red);
  f32x4 is;
#pragma unroll
  for (int e = 0; e < 4; ++e) is[e] = 1.0f / sqrtf(acc[0][e] * inv_n + eps);
  if (slot == 0) {
    *reinterpret_cast<f32x4*>(mean_out + (size_t)w * ldstat + c0) = mu;
    *reinterpret_cast<f32x4*>(invstd_out + (size_t)w * ldstat + c0) = is;
  }
  if (!POOL && !out) return;                         // (block-uniform)
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
  unsigned long long bits = 0ull;
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    const int p = slot + k * P;
    if (p < Wn) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[k][e] - mu[e]) * is[e] * ga[e] + be[e];
      if (res) {
        f32x4 r;
        if constexpr (RX3) r = X3::ld4(reinterpret_cast<const __bf16*>(res) + (base + p) * (size_t)(3 * C), c0);
        else r = Act<AT>::ld4(rb + (uint32_t)(p * ldr + q * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += r[e];
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bits |= (unsigned long long)(o[e] > 0.f) << (k * 4 + e);
          o[e] = fmaxf(o[e], 0.f);
        }
      }
      if constexpr (POOL) *reinterpret_cast<f32x4*>(&pool_s[p * CGB + q * 4]) = act_round4<AT>(o);
      else if constexpr (OX3) X3::st4(reinterpret_cast<__bf16*>(out) + (base + p) * (size_t)(3 * C), c0, o);
      else Act<AT>::st4(ob + (uint32_t)(p * ldo + q * 4), o);
    }
  }
  if constexpr (POOL) {
    __syncthreads();
    const int R = Wn / pool_L;
    const float inv_l = 1.0f / (float)pool_L;
    for (int i = threadIdx.x; i < R * CGB; i += blockDim.x) {
      const int r = i / CGB, c = i - r * CGB;
      float acc = 0.f;
      for (int l = 0; l < pool_L; ++l) acc += pool_s[(r * pool_L + l) * CGB + c];
      pool_out[((size_t)w * R + r) * C + cg * CGB + c] = acc * inv_l;
    }
  }
  // ReLU decisions of this thread's 4 x NPOS elements: the backward kernel (same geometry, same thread -> element map)
  // reads 8 bytes per thread instead of the whole output tensor
  if (mask) mask[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x] = bits;
}

template <typename AT, int NPOS, int QB, int RX3 = 0, int OX3 = 0>
__global__ __launch_bounds__(1024) void bn_fwd_fused_kernel(const AT* __restrict__ x, int ldx,
                                                            const AT* __restrict__ res, int ldr,
                                                            AT* __restrict__ out, int ldo, int Wn, int C,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int relu, float eps,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ invstd_out,
                                                            unsigned long long* __restrict__ mask, int ldstat) {
  __shared__ float red[16 * CG];
  bn_fwd_fused_body<AT, NPOS, QB, RX3, OX3>(x, ldx, res, ldr, out, ldo, Wn, C, gamma, beta, relu, eps, mean_out, invstd_out, mask,
                                            ldstat, red);
}

template <typename AT, int NPOS, int QB>
__global__ __launch_bounds__(1024) void bn_fwd_pool_kernel(const AT* __restrict__ x, int ldx, const AT* __restrict__ res,
                                                           int ldr, int Wn, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                           unsigned long long* __restrict__ mask,
                                                           float* __restrict__ pool_out, int pool_L) {
  __shared__ float red[16 * CG];
  __shared__ float pool_s[BN_POOL_MAX_WN * 32];
  bn_fwd_fused_body<AT, NPOS, QB, 0, 0, 1>(x, ldx, res, ldr, (AT*)nullptr, 0, Wn, C, gamma, beta, 1, eps, mean_out, invstd_out,
                                           mask, C, red, pool_out, pool_L, pool_s);
}

// Two BatchNorms of ONE geometry (W, Wn, C) in one launch, blockIdx.z = which: the two that follow a stride-2 block entry's
// shared conv launch (bn1 + ReLU on the conv output, the downsample's BatchNorm on the 1x1 output; resnet.py:27-29,36-37)
template <typename AT>
struct BnFwdOne {
  const AT* x;
  const AT* res;
  AT* out;
  const float* gamma;
  const float* beta;
  float* mean;
  float* invstd;
  unsigned long long* mask;
  int ldx, ldr, ldo, relu;
};
template <typename AT, int NPOS, int QB>
__global__ __launch_bounds__(1024) void bn_fwd_pair_kernel(BnFwdOne<AT> a, BnFwdOne<AT> b, int Wn, int C, float eps) {
  __shared__ float red[16 * CG];
  const BnFwdOne<AT>& s = blockIdx.z ? b : a;
  bn_fwd_fused_body<AT, NPOS, QB, 0, 0>(s.x, s.ldx, s.res, s.ldr, s.out, s.ldo, Wn, C, s.gamma, s.beta, s.relu, eps, s.mean,
                                        s.invstd, s.mask, C, red);
}

// same arithmetic as bn_bwd_reduce_kernel + bn_bwd_apply_kernel with the slab held in registers
// DX3: dx (the gradient w.r.t. the BatchNorm input = the conv output: the data-gradient and weight-gradient convs' operand)
// is stored in the x3 format; gout stays float (the convs accumulate the branch gradient into it).
template <typename AT, int NPOS, int QB, int DX3 = 0, int EXT = 0, int DPOOL = 0, int D2 = 0>
__device__ __forceinline__ void bn_bwd_fused_body(const AT* __restrict__ dout, int ldd, const AT* __restrict__ x, int ldx,
                                                  const AT* __restrict__ outp, int ldo, AT* __restrict__ dx, int lddx,
                                                  AT* __restrict__ gout, int ldg, int Wn, int C,
                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  int mask_mode, float* __restrict__ ds1, float* __restrict__ ds2,
                                                  const AT* __restrict__ add, int ldadd,
                                                  const unsigned long long* __restrict__ mask, const BnBwdExt& ext, float* red) {
  constexpr int NQ = 1 << QB, CGB = 4 * NQ;            // channel quads / channels per block (32 or 16)
  const int w = row_xcd_chunk(blockIdx.x, gridDim.x), cg = blockIdx.y, P = blockDim.x >> QB;
  const int q = threadIdx.x & (NQ - 1), slot = threadIdx.x >> QB;
  const int c0 = cg * CGB + q * 4;
  // wave-uniform slab bases + 32-bit lane offsets (one SGPR pair + one VGPR per access instead of a 64-bit VGPR pair)
  const size_t base = (size_t)w * Wn;
  const AT* db = dout + (EXT && ext.half_dout ? base >> 1 : base) * ldd + cg * CGB;      // (Wn even with half_dout)
  const AT* xb = x + base * ldx + cg * CGB;
  const AT* ob = outp ? outp + base * ldo + cg * CGB : nullptr;
  AT* dxb = dx + base * lddx + cg * CGB;
  AT* gb = gout ? gout + base * ldg + cg * CGB : nullptr;
  const AT* ab = add ? add + base * ldadd + cg * CGB : nullptr;       // dx = bn_bwd(...) + add (pass-through gradient)
  const int lds_ = EXT ? ext.ldstat : C;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * lds_ + c0);
  const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * lds_ + c0);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
  f32x4 g[NPOS], xh[NPOS];
  f32x4 g2[D2 ? NPOS : 1];         // (40 more registers: the D2 kernels are bounded to 512 threads)
  const AT* d2b = D2 ? reinterpret_cast<const AT*>(ext.dout2) + base * ext.ldd2 + cg * CGB : nullptr;
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    const int p = slot + k * P;
    g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    xh[k] = mu;
    if constexpr (D2) {
      g2[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p < Wn) g2[k] = Act<AT>::ld4(d2b + (uint32_t)(p * ext.ldd2 + q * 4));
    }
    if (p < Wn) {
      if constexpr (DPOOL) {
        const float* df = reinterpret_cast<const float*>(dout) + ((size_t)w * (Wn / ext.pool_L)) * ldd + cg * CGB;
        g[k] = *reinterpret_cast<const f32x4*>(df + (size_t)fdiv((uint32_t)p, ext.pool_div) * ldd + q * 4);
      } else if (EXT && ext.half_dout) {
        g[k] = Act<AT>::ld4(db + (uint32_t)((p >> 1) * ldd + q * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) g[k][e] *= 0.5f;
      } else {
        g[k] = Act<AT>::ld4(db + (uint32_t)(p * ldd + q * 4));
      }
      xh[k] = Act<AT>::ld4(xb + (uint32_t)(p * ldx + q * 4));
    }
  }
  // EXT: gamma / beta live on only as sc = gamma invstd (the factor of dx as well) and sh -- the ReLU decision of the scale /
  // shift form (mask_mode 4); the slab already fills the 128 registers a 1024-thread block may hold
  f32x4 sc, sh;
  if (EXT) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a_, b_;
      bn_scale_shift(mu[e], is[e], ga[e], be[e], a_, b_);
      sc[e] = a_;
      sh[e] = b_;
    }
  }
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const unsigned long long mbits =
      mask ? mask[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x] : 0ull;
  const float pool_inv = DPOOL ? 1.0f / (float)ext.pool_L : 1.0f;
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    const int p = slot + k * P;
    if constexpr (D2) {            // (all loads are out by now)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[k][e] += g2[k][e];
    }
    if constexpr (DPOOL) {         // (here, not behind the load: a VALU op on a fresh register serialises the loads)
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = g[k][e] * pool_inv;
      g[k] = act_round4<AT>(t);    // what head_bwd_kernel would have stored in the activation type
    }
    if (EXT && mask_mode == 4) {   // before xh overwrites the raw value
      f32x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = fmaf(xh[k][e], sc[e], sh[e]);
        g[k][e] = (h[e] > 0.f) ? g[k][e] : 0.f;
        h[e] = fmaxf(h[e], 0.f);
      }
      if (EXT == 2 && p < Wn)       // (a template form of its own: the dense blocks' instantiation sits at the register limit)
        Act<AT>::st4(reinterpret_cast<AT*>(ext.hout) + (base + p) * (size_t)ext.ldh + cg * CGB + q * 4, h);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) xh[k][e] = (xh[k][e] - mu[e]) * is[e];
    if (EXT) {                            // modes 0, 2 (sign of the stored output) and 4 only
      if (mask_mode == 2 && p < Wn) {
        const f32x4 o = Act<AT>::ld4(ob + (uint32_t)(p * ldo + q * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) g[k][e] = (o[e] > 0.f) ? g[k][e] : 0.f;
      }
    } else if (mask_mode == 3) {          // ReLU decisions recorded by the forward kernel (bn_fwd_fused_kernel, same geometry)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[k][e] = ((mbits >> (k * 4 + e)) & 1ull) ? g[k][e] : 0.f;
    } else if (p < Wn) {
      g[k] = bn_masked_g(g[k], xh[k], ga, be, mask_mode, ob, (uint32_t)(p * ldo + q * 4));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0][e] += g[k][e];
      acc[1][e] += g[k][e] * xh[k][e];
    }
  }
  quad_block_sum<2, NQ>(acc, red);
  if (slot == 0) {
    *reinterpret_cast<f32x4*>(ds1 + (size_t)w * C + c0) = acc[0];
    *reinterpret_cast<f32x4*>(ds2 + (size_t)w * C + c0) = acc[1];
  }
  const float inv_n = 1.0f / (float)Wn;
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    const int p = slot + k * P;
    if (p < Wn) {
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        d[e] = (EXT ? sc[e] : ga[e] * is[e]) * (g[k][e] - acc[0][e] * inv_n - xh[k][e] * acc[1][e] * inv_n);
      if (ab) {
        const f32x4 av = Act<AT>::ld4(ab + (uint32_t)(p * ldadd + q * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] += av[e];
      }
      if (EXT && ext.p > 0.f && c0 >= ext.drop_c0) {  // (wave-uniform for 16- / 32-channel blocks and 32-channel segments)
        const long long sd = ext.seed[0];
        const uint32_t key = bn_mix32((uint32_t)sd ^ (uint32_t)(sd >> 32), ext.salt), thr = (uint32_t)(ext.p * 4294967296.0);
        const float scale = 1.0f / (1.0f - ext.p);
        const size_t i0 = (base + p) * (size_t)ext.drop_g + (c0 - ext.drop_c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const size_t i = i0 + e;
          const uint32_t hsh = bn_mix32(key, (uint32_t)i ^ (uint32_t)(i >> 32) * 0x27d4eb2fu);
          d[e] = hsh >= thr ? d[e] * scale : 0.f;
        }
      }
      if constexpr (DX3) X3::st4(reinterpret_cast<__bf16*>(dx) + (base + p) * (size_t)(3 * C), c0, d);
      else Act<AT>::st4(dxb + (uint32_t)(p * lddx + q * 4), d);
      if (gb) Act<AT>::st4(gb + (uint32_t)(p * ldg + q * 4), g[k]);
    }
  }
}

template <typename AT, int NPOS, int QB, int DX3 = 0, int EXT = 0>
__global__ __launch_bounds__(1024) void bn_bwd_fused_kernel(const AT* __restrict__ dout, int ldd,
                                                            const AT* __restrict__ x, int ldx,
                                                            const AT* __restrict__ outp, int ldo,
                                                            AT* __restrict__ dx, int lddx, AT* __restrict__ gout,
                                                            int ldg, int Wn, int C, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int mask_mode,
                                                            float* __restrict__ ds1, float* __restrict__ ds2,
                                                            const AT* __restrict__ add, int ldadd,
                                                            const unsigned long long* __restrict__ mask, BnBwdExt ext) {
  __shared__ float red[16 * 2 * CG];
  bn_bwd_fused_body<AT, NPOS, QB, DX3, EXT>(dout, ldd, x, ldx, outp, ldo, dx, lddx, gout, ldg, Wn, C, mean, invstd, gamma, beta,
                                            mask_mode, ds1, ds2, add, ldadd, mask, ext, red);
}

// the backward of bn_fwd_pool_kernel: dout = the gradient of the pooled features [rows][ldd] (float), ReLU decisions from the mask
template <typename AT, int NPOS, int QB>
__global__ __launch_bounds__(1024) void bn_bwd_pool_kernel(const float* __restrict__ dflat, int ldd, const AT* __restrict__ x,
                                                           int ldx, AT* __restrict__ dx, int lddx, AT* __restrict__ gout, int ldg,
                                                           int Wn, int C, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ ds1,
                                                           float* __restrict__ ds2, const unsigned long long* __restrict__ mask,
                                                           BnBwdExt ext) {
  __shared__ float red[16 * 2 * CG];
  bn_bwd_fused_body<AT, NPOS, QB, 0, 0, 1>(reinterpret_cast<const AT*>(dflat), ldd, x, ldx, (const AT*)nullptr, 0, dx, lddx, gout,
                                           ldg, Wn, C, mean, invstd, gamma, beta, 3, ds1, ds2, (const AT*)nullptr, 0, mask, ext,
                                           red);
}

// the block-output BatchNorm's backward with the upstream gradient in two terms, dout + dout2 (mask form, optional gout): a
// residual block behind it left its input gradient as (data-gradient conv output, identity branch) instead of accumulating
// the first onto the second in the conv's epilogue -- the accumulating launches cost 4 ... 8 us more than the plain ones
template <typename AT, int NPOS, int QB>
__global__ __launch_bounds__(512) void bn_bwd_d2_kernel(const AT* __restrict__ dout, int ldd, const AT* __restrict__ x, int ldx,
                                                        AT* __restrict__ dx, int lddx, AT* __restrict__ gout, int ldg, int Wn,
                                                        int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ ds1, float* __restrict__ ds2,
                                                        const unsigned long long* __restrict__ mask, BnBwdExt ext) {
  __shared__ float red[16 * 2 * CG];
  bn_bwd_fused_body<AT, NPOS, QB, 0, 0, 0, 1>(dout, ldd, x, ldx, (const AT*)nullptr, 0, dx, lddx, gout, ldg, Wn, C, mean, invstd,
                                              gamma, beta, 3, ds1, ds2, (const AT*)nullptr, 0, mask, ext, red);
}

// ... and their backward: the block-output BatchNorm (bn2) and the downsample's BatchNorm take the SAME masked gradient
// dout * [out > 0] (ReLU decisions as the bit mask of the forward), so neither waits for the other (resnet.py:33-38 backward)
template <typename AT>
struct BnBwdOne {
  const AT* x;
  AT* dx;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  float* ds;           // [2][W][C]
  int ldx, lddx;
};
template <typename AT, int NPOS, int QB>
__global__ __launch_bounds__(1024) void bn_bwd_pair_kernel(const AT* __restrict__ dout, int ldd, BnBwdOne<AT> a, BnBwdOne<AT> b,
                                                           int W, int Wn, int C, const unsigned long long* __restrict__ mask) {
  __shared__ float red[16 * 2 * CG];
  const BnBwdOne<AT>& s = blockIdx.z ? b : a;
  BnBwdExt ext = {};
  bn_bwd_fused_body<AT, NPOS, QB, 0, 0>(dout, ldd, s.x, s.ldx, (const AT*)nullptr, 0, s.dx, s.lddx, (AT*)nullptr, 0, Wn, C,
                                        s.mean, s.invstd, s.gamma, s.beta, 3, s.ds, s.ds + (size_t)W * C, (const AT*)nullptr, 0,
                                        mask, ext, red);
}

template <typename AT, int NPOS, int QB>
__global__ __launch_bounds__(512) void bn_bwd_pair_d2_kernel(const AT* __restrict__ dout, int ldd, BnBwdOne<AT> a, BnBwdOne<AT> b,
                                                             int W, int Wn, int C, const unsigned long long* __restrict__ mask,
                                                             BnBwdExt ext) {
  __shared__ float red[16 * 2 * CG];
  const BnBwdOne<AT>& s = blockIdx.z ? b : a;
  bn_bwd_fused_body<AT, NPOS, QB, 0, 0, 0, 1>(dout, ldd, s.x, s.ldx, (const AT*)nullptr, 0, s.dx, s.lddx, (AT*)nullptr, 0, Wn, C,
                                              s.mean, s.invstd, s.gamma, s.beta, 3, s.ds, s.ds + (size_t)W * C,
                                              (const AT*)nullptr, 0, mask, ext, red);
}

// geometry of the single-pass kernels for W windows of Wn positions: channels per block (32, or 16 when 32 would
// leave CUs without a block or the window too long for one block) and block size; 0: use the two-stage path
static int g_bn_target_blocks = 256;   // blocks a launch should have before the channel group per block stops shrinking
static int bn_fused_geometry(int W, int Wn, int C, int* cgb) {
  if (Wn < 1 || g_bn_two_stage) return 0;
  int nq = 8;                                                  // channel quads per block: 32, 16 or 8 channels
  while (nq > 2 && (Wn > FUSED_NPOS * (1024 / nq) || (long)W * (C / (4 * nq)) < g_bn_target_blocks)) nq >>= 1;
  if (nq == 2 && g_bn_target_blocks <= 256) nq = 4;           // 8-channel blocks only on request (32-byte row segments)
  if (Wn > FUSED_NPOS * (1024 / nq) || C % (4 * nq)) return 0;
  const int mult = 64 / nq;                                  // whole waves
  int P = (Wn + FUSED_NPOS - 1) / FUSED_NPOS;
  P = (P + mult - 1) / mult * mult;
  *cgb = 4 * nq;
  return nq * P;
}

// dbeta[c] (+)= sum_w s1[w][c];  dgamma[c] (+)= sum_w s2[w][c] for up to 32 BatchNorms in one launch
// (blockIdx.y = which BN); block = 32 channels x 8 window slots, fixed order (deterministic).  (BnPgradDesc and the block's
// body: common.h -- the step's slab reduction carries the same folds as extra blocks, da_wgrad_reduce_pgrad_multi.)
struct BnPgradTable {
  BnPgradDesc d[32];
};

__global__ __launch_bounds__(256) void bn_param_grad_multi_kernel(BnPgradTable t, int accumulate) {
  __shared__ float red[2][8][32];
  bn_param_grad_block(t.d[blockIdx.y], blockIdx.x, accumulate, red);
}

// out = max(fmaf(x, sc, sh), 0) with the statistics of a pitched table: the activation the dense-block path never stores
// (bn_scale_shift above), materialised for tests (their ReLU-decision taps) and for explainers.
__global__ __launch_bounds__(256) void bn_relu_ss_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo,
                                                         int Wn, int C, size_t npos, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, int ldstat,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         FastDiv divWn) {
  const int nq = C >> 2;
  const size_t total = npos * nq;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % nq);
    const size_t pos = idx / nq;
    const int c0 = q * 4;
    const uint32_t w = fdiv((uint32_t)pos, divWn);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * ldstat + c0);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * ldstat + c0);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + pos * ldx + c0);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sc, sh;
      bn_scale_shift(mu[e], is[e], ga[e], be[e], sc, sh);
      o[e] = fmaxf(fmaf(v[e], sc, sh), 0.f);
    }
    *reinterpret_cast<f32x4*>(out + pos * ldo + c0) = o;
  }
}

// chunking of a window's Wn positions so that (W * C/32 * P) fills the chip (~4 blocks per CU)
static void bn_chunks(int W, int Wn, int C, int* P, int* chunk) {
  long base = (long)W * (C / CG);
  int p = (int)((1024 + base - 1) / base);
  int maxp = (Wn + 127) / 128;                   // at least 128 positions (4 per slot) per chunk
  if (p > maxp) p = maxp;
  if (p < 1) p = 1;
  int ch = ((Wn + p - 1) / p + 31) / 32 * 32;
  *P = (Wn + ch - 1) / ch;
  *chunk = ch;
}

typedef struct {
  const void* x; int ldx;
  const void* res; int ldr;
  void* out; int ldo;
  float* mean; float* invstd;
  const float* gamma; const float* beta;
  int relu;
  unsigned long long* mask;
} da_bn_fwd_desc;
typedef struct {
  const void* x; int ldx;
  void* dx; int lddx;
  const float* mean; const float* invstd;
  const float* gamma; const float* beta;
  float* ds;
} da_bn_bwd_desc;

template <typename AT, int QB>
static void launch_bn_fwd_pair(const da_bn_fwd_desc* d, int W, int Wn, int C, int CH, int threads, float eps, hipStream_t stream) {
  BnFwdOne<AT> s[2];
  for (int i = 0; i < 2; ++i)
    s[i] = BnFwdOne<AT>{(const AT*)d[i].x, (const AT*)d[i].res, (AT*)d[i].out, d[i].gamma, d[i].beta, d[i].mean, d[i].invstd,
                        d[i].mask, d[i].ldx, d[i].ldr, d[i].ldo, d[i].relu};
  hipLaunchKernelGGL((bn_fwd_pair_kernel<AT, FUSED_NPOS, QB>), dim3(W, C / CH, 2), dim3(threads), 0, stream, s[0], s[1], Wn, C, eps);
}

template <typename AT, int QB>
static void launch_bn_bwd_pair(const void* dout, int ldd, const da_bn_bwd_desc* d, int W, int Wn, int C, int CH, int threads,
                               const unsigned long long* mask, hipStream_t stream, const void* dout2 = nullptr, int ldd2 = 0) {
  BnBwdOne<AT> s[2];
  for (int i = 0; i < 2; ++i)
    s[i] = BnBwdOne<AT>{(const AT*)d[i].x, (AT*)d[i].dx, d[i].mean, d[i].invstd, d[i].gamma, d[i].beta, d[i].ds, d[i].ldx, d[i].lddx};
  if (dout2) {
    BnBwdExt ext = {};
    ext.dout2 = dout2;
    ext.ldd2 = ldd2;
    hipLaunchKernelGGL((bn_bwd_pair_d2_kernel<AT, FUSED_NPOS, QB>), dim3(W, C / CH, 2), dim3(threads), 0, stream, (const AT*)dout,
                       ldd, s[0], s[1], W, Wn, C, mask, ext);
    return;
  }
  hipLaunchKernelGGL((bn_bwd_pair_kernel<AT, FUSED_NPOS, QB>), dim3(W, C / CH, 2), dim3(threads), 0, stream, (const AT*)dout, ldd,
                     s[0], s[1], W, Wn, C, mask);
}

extern "C" {

// Host-side descriptor of one BatchNorm for the batched small kernels (HOST arrays of these are passed).
typedef struct {
  const float* mean;
  const float* invstd;
  float* running_mean;
  float* running_var;
  long long* num_batches_tracked;
  int W, C, Wn;
  float eps, momentum;
} da_bn_running_desc;

typedef struct {
  const float* s1;
  const float* s2;
  float* dgamma;
  float* dbeta;
  int W, C;
} da_bn_pgrad_desc;

int da_sizeof_bn_running_desc(void) { return (int)sizeof(da_bn_running_desc); }
int da_sizeof_bn_pgrad_desc(void) { return (int)sizeof(da_bn_pgrad_desc); }

// chunk geometry shared by da_bn_stats_partial / da_bn_apply(part) / da_bn_bwd: P chunks of `chunk` positions
void da_bn_chunks(int W, int Wn, int C, int* P, int* chunk) { bn_chunks(W, Wn, C, P, chunk); }

// floats of scratch the statistics / backward stages need: 2*W*C*P
size_t da_bn_workspace(int W, int Wn, int C) {
  int P, chunk;
  bn_chunks(W, Wn, C, &P, &chunk);
  return (size_t)2 * W * C * P * sizeof(float);
}

// stage 1 of the statistics: part[w][p][{mean,M2}][C] chunk records (da_bn_workspace() bytes).
int da_bn_stats_partial(const void* x, int ld, int W, int Wn, int C, float* part, hipStream_t stream) {
  DA_ENTER();
  if (!x || !part || C % CG || ld % 4 || Wn < 1) return DA_EINVAL;
  if (W == 0) return DA_OK;
  int P, chunk;
  bn_chunks(W, Wn, C, &P, &chunk);
  DA_ACT_DISPATCH(hipLaunchKernelGGL(bn_stats_partial_kernel<AT>, dim3(W, C / CG, P), dim3(256), 0, stream, (const AT*)x, ld,
                                     Wn, C, chunk, part));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// stage 1 for the default stem's BatchNorm, its input recomputed from the raw rows (stem_stats_partial_kernel):
// xrows (rows, Lin) float, wt (C, 1, 7); window = R rows of Lc = Lin / 2 conv outputs.  Records as da_bn_stats_partial's.
int da_stem_stats_partial(const float* xrows, const float* wt, int rows, int R, int Lin, int C, float* part, hipStream_t stream) {
  DA_ENTER();
  if (!xrows || !wt || !part || C % CG || R < 1 || rows % R || Lin < 2 || (Lin & 1)) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const int W = rows / R, Lc = Lin / 2, Wn = R * Lc;
  int P, chunk;
  bn_chunks(W, Wn, C, &P, &chunk);
  const int XSR = Lin + 12;                           // 3 zeros in front, >= 9 behind (the last output reads inputs up to Lin + 1)
  const int max_rows = (chunk + Lc - 1) / Lc + 1;
  const size_t shm = ((size_t)max_rows * XSR + (SLOTS + 1) * CG) * sizeof(float);
  if (shm > 64 * 1024) return DA_EINVAL;
  hipLaunchKernelGGL(stem_stats_partial_kernel, dim3(W, C / CG, P), dim3(256), shm, stream, xrows, wt, Lin, Lc, Wn, C, chunk, XSR,
                     part);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// stage 2, standalone: mean/invstd [W][C] from the chunk records.
int da_bn_stats_merge(const float* part, int W, int Wn, int C, float eps, float* mean, float* invstd,
                      hipStream_t stream) {
  DA_ENTER();
  if (!part || !mean || !invstd || C % CG) return DA_EINVAL;
  if (W == 0) return DA_OK;
  int P, chunk;
  bn_chunks(W, Wn, C, &P, &chunk);
  hipLaunchKernelGGL(bn_stats_merge_kernel, dim3((W * C + 255) / 256), dim3(256), 0, stream, part, W, P, C, Wn, chunk,
                     eps, mean, invstd);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// running-stat updates of n BatchNorms (descs: HOST array), 32 per launch.
int da_bn_running_multi(const da_bn_running_desc* descs, int n, hipStream_t stream) {
  DA_ENTER();
  if (n < 0 || (n && !descs)) return DA_EINVAL;
  for (int base = 0; base < n; base += 32) {
    BnRunningTable t;
    int m = n - base < 32 ? n - base : 32, maxc = 0;
    for (int i = 0; i < m; ++i) {
      const da_bn_running_desc& s = descs[base + i];
      if (!s.mean || !s.invstd || !s.running_mean || !s.running_var || s.Wn < 1) return DA_EINVAL;
      t.d[i] = {s.mean, s.invstd, s.running_mean, s.running_var, s.num_batches_tracked, s.W, s.C, s.Wn, s.eps, s.momentum};
      if (s.C > maxc) maxc = s.C;
    }
    hipLaunchKernelGGL(bn_running_multi_kernel, dim3((maxc + 31) / 32, m), dim3(256), 0, stream, t);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

// dgamma/dbeta of n BatchNorms from their per-window totals (descs: HOST array), 32 per launch.
int da_bn_param_grad_multi(const da_bn_pgrad_desc* descs, int n, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (n < 0 || (n && !descs)) return DA_EINVAL;
  for (int base = 0; base < n; base += 32) {
    BnPgradTable t;
    int m = n - base < 32 ? n - base : 32, maxc = 0;
    for (int i = 0; i < m; ++i) {
      const da_bn_pgrad_desc& s = descs[base + i];
      if (!s.s1 || !s.s2 || !s.dgamma || !s.dbeta) return DA_EINVAL;
      t.d[i] = {s.s1, s.s2, s.dgamma, s.dbeta, s.W, s.C};
      if (s.C > maxc) maxc = s.C;
    }
    hipLaunchKernelGGL(bn_param_grad_multi_kernel, dim3((maxc + 31) / 32, m), dim3(256), 0, stream, t, accumulate);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

// out = act(bn(x) (+res)).  Statistics either as mean/invstd [W][C], or (part != NULL) as the chunk
// records of da_bn_stats_partial: they are merged on the fly and mean/invstd are WRITTEN as a by-product.
int da_bn_apply(const void* x, int ldx, const void* res, int ldr, void* out, int ldo, int W, int Wn, int C,
                float* mean, float* invstd, const float* gamma, const float* beta, int relu, const float* part,
                float eps, hipStream_t stream) {
  DA_ENTER();
  if (!x || !out || !mean || !invstd || !gamma || !beta || C % CG || ldx % 4 || ldo % 4 || (res && ldr % 4))
    return DA_EINVAL;
  if (W == 0) return DA_OK;
  int chunk = 256;
  int nz = (Wn + chunk - 1) / chunk;
  int P, schunk;
  bn_chunks(W, Wn, C, &P, &schunk);
  DA_ACT_DISPATCH(hipLaunchKernelGGL(bn_apply_kernel<AT>, dim3(W, C / CG, nz), dim3(256), 0, stream, (const AT*)x, ldx,
                                     (const AT*)res, ldr, (AT*)out, ldo, Wn, C, mean, invstd, gamma, beta, relu, chunk, part,
                                     P, schunk, eps, mean, invstd));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// Statistics + normalisation in one call: mean/invstd [W][C] are OUTPUTS, out = act(bn(x) (+res)).
// One single-pass kernel when a window slab fits a block's registers (Wn <= 1280), otherwise
// da_bn_stats_partial + da_bn_apply.  scratch: da_bn_workspace() bytes.
static int bn_fwd_impl(const void* x, int ldx, const void* res, int ldr, void* out, int ldo, int W, int Wn, int C,
                       float* mean, float* invstd, const float* gamma, const float* beta, int relu, float eps,
                       float* scratch, unsigned long long* mask, hipStream_t stream) {
  DA_ENTER();
  if (!x || !out || !mean || !invstd || !gamma || !beta || !scratch || C % CG || ldx % 4 || ldo % 4 ||
      (res && ldr % 4) || Wn < 1)
    return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  if (int threads = bn_fused_geometry(W, Wn, C, &cgb)) {
#define BN_FWD_LAUNCH(QB, CH)                                                                                        \
  DA_ACT_DISPATCH(hipLaunchKernelGGL((bn_fwd_fused_kernel<AT, FUSED_NPOS, QB>), dim3(W, C / CH), dim3(threads), 0, stream, \
                                     (const AT*)x, ldx, (const AT*)res, ldr, (AT*)out, ldo, Wn, C, gamma, beta, relu, eps,   \
                                     mean, invstd, mask, C))
    if (cgb == 32) BN_FWD_LAUNCH(3, 32);
    else if (cgb == 16) BN_FWD_LAUNCH(2, 16);
    else BN_FWD_LAUNCH(1, 8);
#undef BN_FWD_LAUNCH
    DA_CHECK_LAUNCH();
    return DA_OK;
  }
  if (mask) return DA_EINVAL;                    // masks exist only for the single-pass geometry (da_bn_mask_words() > 0)
  int rc = da_bn_stats_partial(x, ldx, W, Wn, C, scratch, stream);
  if (rc) return rc;
  return da_bn_apply(x, ldx, res, ldr, out, ldo, W, Wn, C, mean, invstd, gamma, beta, relu, scratch, eps, stream);
}

int da_bn_fwd(const void* x, int ldx, const void* res, int ldr, void* out, int ldo, int W, int Wn, int C,
              float* mean, float* invstd, const float* gamma, const float* beta, int relu, float eps, float* scratch,
              hipStream_t stream) {
  return bn_fwd_impl(x, ldx, res, ldr, out, ldo, W, Wn, C, mean, invstd, gamma, beta, relu, eps, scratch, nullptr, stream);
}

// da_bn_fwd / da_bn_fwd_mask with x3 operands (float activations, single-pass geometry only: returns -1 for a shape
// that would take the two-stage kernels -- ask da_bn_mask_words() > 0 first): res_x3 / out_x3 say which of `res`, `out`
// are in the x3 format (3 C bf16 per position, pitches ignored for those); mask may be NULL.
int da_bn_fwd_x(const float* x, int ldx, const void* res, int ldr, void* out, int ldo, int W, int Wn, int C, float* mean,
                float* invstd, const float* gamma, const float* beta, int relu, float eps, unsigned long long* mask,
                int res_x3, int out_x3, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;
  if (!x || !out || !mean || !invstd || !gamma || !beta || C % CG || ldx % 4 || Wn < 1) return DA_EINVAL;
  if ((!out_x3 && ldo % 4) || (res && !res_x3 && ldr % 4) || ((res_x3 || out_x3) && C % 16)) return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
  if (!threads) return DA_EINVAL;
  const int rx = res && res_x3 ? 1 : 0, ox = out_x3 ? 1 : 0;
#define BN_FWDX_LAUNCH(QB, CH, RX, OX)                                                                                  \
  hipLaunchKernelGGL((bn_fwd_fused_kernel<float, FUSED_NPOS, QB, RX, OX>), dim3(W, C / CH), dim3(threads), 0, stream, x, ldx,   \
                     (const float*)res, ldr, (float*)out, ldo, Wn, C, gamma, beta, relu, eps, mean, invstd, mask, C)
#define BN_FWDX_QB(QB, CH)                                  \
  do {                                                      \
    if (rx && ox) BN_FWDX_LAUNCH(QB, CH, 1, 1);             \
    else if (rx) BN_FWDX_LAUNCH(QB, CH, 1, 0);              \
    else if (ox) BN_FWDX_LAUNCH(QB, CH, 0, 1);              \
    else BN_FWDX_LAUNCH(QB, CH, 0, 0);                      \
  } while (0)
  if (cgb == 32) BN_FWDX_QB(3, 32);
  else if (cgb == 16) BN_FWDX_QB(2, 16);
  else BN_FWDX_QB(1, 8);
#undef BN_FWDX_QB
#undef BN_FWDX_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// da_bn_bwd / da_bn_bwd_mask with dx optionally stored in the x3 format (dx_x3); dout, x, gout float.  mask != NULL: the
// ReLU decisions of da_bn_fwd_x; else mask_mode 0 (no ReLU) or 1 (ReLU, recomputed from bn(x)).  Single-pass geometry only.
int da_bn_bwd_x(const float* dout, int ldd, const float* x, int ldx, void* dx, int lddx, float* gout, int ldg, int W, int Wn,
                int C, const float* mean, const float* invstd, const float* gamma, const float* beta, int mask_mode,
                float* ds, const unsigned long long* mask, int dx_x3, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;
  if (mask) mask_mode = 3;
  if (!dout || !x || !dx || !mean || !invstd || !gamma || !beta || !ds) return DA_EINVAL;
  if (C % CG || ldd % 4 || ldx % 4 || (!dx_x3 && lddx % 4) || (dx_x3 && C % 16) || (gout && ldg % 4)) return DA_EINVAL;
  if (mask_mode != 0 && mask_mode != 1 && mask_mode != 3) return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
  if (!threads) return DA_EINVAL;
  float* s1 = ds;
  float* s2 = ds + (size_t)W * C;
#define BN_BWDX_LAUNCH(QB, CH, DX)                                                                                        \
  hipLaunchKernelGGL((bn_bwd_fused_kernel<float, FUSED_NPOS, QB, DX>), dim3(W, C / CH), dim3(threads), 0, stream, dout, ldd, x,  \
                     ldx, (const float*)nullptr, 0, (float*)dx, lddx, gout, ldg, Wn, C, mean, invstd, gamma, beta, mask_mode, s1, \
                     s2, (const float*)nullptr, 0, mask, BnBwdExt{})
#define BN_BWDX_QB(QB, CH)                       \
  do {                                           \
    if (dx_x3) BN_BWDX_LAUNCH(QB, CH, 1);        \
    else BN_BWDX_LAUNCH(QB, CH, 0);              \
  } while (0)
  if (cgb == 32) BN_BWDX_QB(3, 32);
  else if (cgb == 16) BN_BWDX_QB(2, 16);
  else BN_BWDX_QB(1, 8);
#undef BN_BWDX_QB
#undef BN_BWDX_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// tests: 1 = always take the two-stage kernels (so both paths are checked against the oracle)
int da_bn_debug_two_stage(int on) {
  g_bn_two_stage = on != 0;
  return DA_OK;
}

// tuning: blocks per launch the single-pass geometry aims for (default 256; swept in DESIGN_APPENDIX.md section 8)
int da_bn_debug_target_blocks(int blocks) {
  if (blocks < 1) return DA_EINVAL;
  g_bn_target_blocks = blocks;
  return DA_OK;
}

// scratch: da_bn_workspace() bytes.  ds: [2][W][C] per-window totals (sum g, sum g*xhat), always written.
// dgamma/dbeta: [C]; when both are non-NULL they are computed here (accumulated when accumulate != 0),
// when NULL the caller folds ds later with da_bn_param_grad_multi.
static int bn_bwd_impl(const void* dout, int ldd, const void* x, int ldx, const void* out, int ldo, void* dx, int lddx,
                       void* gout, int ldg, int W, int Wn, int C, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, int mask_mode, float* scratch, float* ds, float* dgamma,
                       float* dbeta, int accumulate, const void* add, int ldadd, const unsigned long long* mask,
                       hipStream_t stream) {
  DA_ENTER();
  if (mask) mask_mode = 3;
  if (!dout || !x || !dx || !mean || !invstd || !gamma || !beta || !scratch || !ds) return DA_EINVAL;
  if (add && ldadd % 4) return DA_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return DA_EINVAL;
  if (C % CG || ldd % 4 || ldx % 4 || lddx % 4 || (gout && ldg % 4) || mask_mode < 0 || mask_mode > 3)
    return DA_EINVAL;
  if (mask_mode == 2 && (!out || ldo % 4)) return DA_EINVAL;
  if (W == 0) return DA_OK;
  int P, chunk;
  bn_chunks(W, Wn, C, &P, &chunk);
  float* s1 = ds;
  float* s2 = ds + (size_t)W * C;
  int cgb = 0;
  if (int threads = bn_fused_geometry(W, Wn, C, &cgb)) {
#define BN_BWD_LAUNCH(QB, CH)                                                                                         \
  DA_ACT_DISPATCH(hipLaunchKernelGGL((bn_bwd_fused_kernel<AT, FUSED_NPOS, QB>), dim3(W, C / CH), dim3(threads), 0, stream,  \
                                     (const AT*)dout, ldd, (const AT*)x, ldx, (const AT*)out, ldo, (AT*)dx, lddx, (AT*)gout,  \
                                     ldg, Wn, C, mean, invstd, gamma, beta, mask_mode, s1, s2, (const AT*)add, ldadd, mask, BnBwdExt{}))
    if (cgb == 32) BN_BWD_LAUNCH(3, 32);
    else if (cgb == 16) BN_BWD_LAUNCH(2, 16);
    else BN_BWD_LAUNCH(1, 8);
#undef BN_BWD_LAUNCH
    DA_CHECK_LAUNCH();
  } else {
    if (mask) return DA_EINVAL;
    DA_ACT_DISPATCH(hipLaunchKernelGGL(bn_bwd_reduce_kernel<AT>, dim3(W, C / CG, P), dim3(256), 0, stream, (const AT*)dout, ldd,
                                       (const AT*)x, ldx, (const AT*)out, ldo, Wn, C, chunk, mean, invstd, gamma, beta,
                                       mask_mode, scratch));
    DA_CHECK_LAUNCH();
    DA_ACT_DISPATCH(hipLaunchKernelGGL(bn_bwd_apply_kernel<AT>, dim3(W, C / CG, P), dim3(256), 0, stream, (const AT*)dout, ldd,
                                       (const AT*)x, ldx, (const AT*)out, ldo, (AT*)dx, lddx, (AT*)gout, ldg, Wn, C, chunk,
                                       mean, invstd, gamma, beta, mask_mode, scratch, s1, s2, (const AT*)add, ldadd));
    DA_CHECK_LAUNCH();
  }
  if (dgamma) {
    BnPgradTable t;
    t.d[0] = {s1, s2, dgamma, dbeta, W, C};
    hipLaunchKernelGGL(bn_param_grad_multi_kernel, dim3((C + 31) / 32, 1), dim3(256), 0, stream, t, accumulate);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

int da_bn_bwd(const void* dout, int ldd, const void* x, int ldx, const void* out, int ldo, void* dx, int lddx,
              void* gout, int ldg, int W, int Wn, int C, const float* mean, const float* invstd, const float* gamma,
              const float* beta, int mask_mode, float* scratch, float* ds, float* dgamma, float* dbeta, int accumulate,
              hipStream_t stream) {
  return bn_bwd_impl(dout, ldd, x, ldx, out, ldo, dx, lddx, gout, ldg, W, Wn, C, mean, invstd, gamma, beta, mask_mode,
                     scratch, ds, dgamma, dbeta, accumulate, nullptr, 0, nullptr, stream);
}

// The block-output BatchNorm of the LAST residual block with the head's global average pool folded in (resnet.py:33-38 into
// :112,159-160 AvgPool1d(7) + view): da_bn_fwd_mask(relu) whose output is not stored -- flat[W * Wn / L][C] (float) receives the
// average over the L positions of every row, bit for bit what da_head_fwd pools from the stored map; and its backward,
// da_bn_bwd_mask with dout = dflat[rows][ldd] (float), the gradient of those pooled features.  Single-pass geometry, L | Wn,
// Wn <= 160 (da_bn_pool_ok); mask: da_bn_mask_words() words.
int da_bn_pool_ok(int W, int Wn, int C, int L) {
  int cgb = 0;
  return L >= 1 && Wn % L == 0 && Wn <= BN_POOL_MAX_WN && C % CG == 0 && bn_fused_geometry(W, Wn, C, &cgb) != 0;
}

int da_bn_fwd_pool(const void* x, int ldx, const void* res, int ldr, float* flat, int W, int Wn, int C, int L, float* mean,
                   float* invstd, const float* gamma, const float* beta, float eps, unsigned long long* mask,
                   hipStream_t stream) {
  DA_ENTER();
  if (!x || !flat || !mean || !invstd || !gamma || !beta || !mask || ldx % 4 || (res && ldr % 4)) return DA_EINVAL;
  if (!da_bn_pool_ok(W, Wn, C, L)) return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
#define BN_FWDP_LAUNCH(QB, CH)                                                                                            \
  DA_ACT_DISPATCH(hipLaunchKernelGGL((bn_fwd_pool_kernel<AT, FUSED_NPOS, QB>), dim3(W, C / CH), dim3(threads), 0, stream,    \
                                     (const AT*)x, ldx, (const AT*)res, ldr, Wn, C, gamma, beta, eps, mean, invstd, mask, flat, L))
  if (cgb == 32) BN_FWDP_LAUNCH(3, 32);
  else if (cgb == 16) BN_FWDP_LAUNCH(2, 16);
  else BN_FWDP_LAUNCH(1, 8);
#undef BN_FWDP_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_bn_bwd_pool(const float* dflat, int ldd, const void* x, int ldx, void* dx, int lddx, void* gout, int ldg, int W, int Wn,
                   int C, int L, const float* mean, const float* invstd, const float* gamma, const float* beta, float* ds,
                   const unsigned long long* mask, hipStream_t stream) {
  DA_ENTER();
  if (!dflat || !x || !dx || !mean || !invstd || !gamma || !beta || !ds || !mask || ldd % 4 || ldx % 4 || lddx % 4 ||
      (gout && ldg % 4))
    return DA_EINVAL;
  if (!da_bn_pool_ok(W, Wn, C, L)) return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
  BnBwdExt ext = {};
  ext.pool_L = L;
  ext.pool_div = make_fastdiv((uint32_t)L);
  float* s1 = ds;
  float* s2 = ds + (size_t)W * C;
#define BN_BWDP_LAUNCH(QB, CH)                                                                                            \
  DA_ACT_DISPATCH(hipLaunchKernelGGL((bn_bwd_pool_kernel<AT, FUSED_NPOS, QB>), dim3(W, C / CH), dim3(threads), 0, stream,    \
                                     dflat, ldd, (const AT*)x, ldx, (AT*)dx, lddx, (AT*)gout, ldg, Wn, C, mean, invstd, gamma,   \
                                     beta, s1, s2, mask, ext))
  if (cgb == 32) BN_BWDP_LAUNCH(3, 32);
  else if (cgb == 16) BN_BWDP_LAUNCH(2, 16);
  else BN_BWDP_LAUNCH(1, 8);
#undef BN_BWDP_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// 64-bit words a ReLU mask of da_bn_fwd_mask / da_bn_bwd_mask has for this shape (one per thread of the single-pass
// kernel); 0: the shape takes the two-stage kernels, no mask form.
size_t da_bn_mask_words(int W, int Wn, int C) {
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
  return threads ? (size_t)W * (C / cgb) * threads : 0;
}

// da_bn_fwd (relu != 0) that also records the ReLU decisions, 1 bit per element (da_bn_mask_words() words).
int da_bn_fwd_mask(const void* x, int ldx, const void* res, int ldr, void* out, int ldo, int W, int Wn, int C,
                   float* mean, float* invstd, const float* gamma, const float* beta, float eps, float* scratch,
                   unsigned long long* mask, hipStream_t stream) {
  if (!mask) return DA_EINVAL;
  return bn_fwd_impl(x, ldx, res, ldr, out, ldo, W, Wn, C, mean, invstd, gamma, beta, 1, eps, scratch, mask, stream);
}

// da_bn_bwd of act = ReLU whose decisions come from the mask of da_bn_fwd_mask instead of the output tensor
// (mask_mode 2 reads `out`, 18 MB per layer at B = 64, only for its sign).
int da_bn_bwd_mask(const void* dout, int ldd, const void* x, int ldx, void* dx, int lddx, void* gout, int ldg, int W,
                   int Wn, int C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                   float* scratch, float* ds, float* dgamma, float* dbeta, int accumulate,
                   const unsigned long long* mask, hipStream_t stream) {
  if (!mask) return DA_EINVAL;
  return bn_bwd_impl(dout, ldd, x, ldx, nullptr, 0, dx, lddx, gout, ldg, W, Wn, C, mean, invstd, gamma, beta, 3, scratch,
                     ds, dgamma, dbeta, accumulate, nullptr, 0, mask, stream);
}

// da_bn_bwd with dx = (BatchNorm input gradient) + add[pos][0:C] (pitch ldadd): the pass-through gradient of a
// concatenation (densenet.py:41 torch.cat) joins in the same pass instead of a separate add kernel.
int da_bn_bwd_add(const void* dout, int ldd, const void* x, int ldx, const void* out, int ldo, void* dx, int lddx,
                  void* gout, int ldg, int W, int Wn, int C, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, int mask_mode, float* scratch, float* ds, float* dgamma, float* dbeta,
                  int accumulate, const void* add, int ldadd, hipStream_t stream) {
  if (!add) return DA_EINVAL;
  return bn_bwd_impl(dout, ldd, x, ldx, out, ldo, dx, lddx, gout, ldg, W, Wn, C, mean, invstd, gamma, beta, mask_mode,
                     scratch, ds, dgamma, dbeta, accumulate, add, ldadd, nullptr, stream);
}

// Two BatchNorm forwards of one geometry in ONE launch (a stride-2 block entry: bn1 + ReLU on conv1's output and the
// downsample's BatchNorm; resnet.py:27-29,36-37): d[0], d[1] as da_bn_fwd / da_bn_fwd_mask take them (mask may be NULL).
// Single-pass geometry only (da_bn_mask_words(W, Wn, C) > 0; -1 otherwise: the caller runs the two calls).
int da_bn_fwd_pair(const da_bn_fwd_desc* d, int W, int Wn, int C, float eps, hipStream_t stream) {
  DA_ENTER();
  if (!d || C % CG || Wn < 1) return DA_EINVAL;
  for (int i = 0; i < 2; ++i)
    if (!d[i].x || !d[i].out || !d[i].mean || !d[i].invstd || !d[i].gamma || !d[i].beta || d[i].ldx % 4 || d[i].ldo % 4 ||
        (d[i].res && d[i].ldr % 4) || (d[i].mask && !d[i].relu))
      return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
  if (!threads) return DA_EINVAL;
#define BN_FWDP_LAUNCH(QB, CH)                                                          \
  do {                                                                                  \
    if (g_act_bf16) launch_bn_fwd_pair<__bf16, QB>(d, W, Wn, C, CH, threads, eps, stream); \
    else launch_bn_fwd_pair<float, QB>(d, W, Wn, C, CH, threads, eps, stream);          \
  } while (0)
  if (cgb == 32) BN_FWDP_LAUNCH(3, 32);
  else if (cgb == 16) BN_FWDP_LAUNCH(2, 16);
  else BN_FWDP_LAUNCH(1, 8);
#undef BN_FWDP_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// ... and the two backwards that share one masked gradient (the block-output BatchNorm and the downsample's; the ReLU
// decisions of the block output as the bit mask of da_bn_fwd_mask): dx_i = BatchNorm_i backward of dout * mask, ds_i
// [2][W][C] the window sums for da_bn_param_grad_multi.  Single-pass geometry only.
int da_bn_bwd_pair(const void* dout, int ldd, const da_bn_bwd_desc* d, int W, int Wn, int C, const unsigned long long* mask,
                   hipStream_t stream) {
  DA_ENTER();
  if (!dout || !d || !mask || C % CG || Wn < 1 || ldd % 4) return DA_EINVAL;
  for (int i = 0; i < 2; ++i)
    if (!d[i].x || !d[i].dx || !d[i].mean || !d[i].invstd || !d[i].gamma || !d[i].beta || !d[i].ds || d[i].ldx % 4 ||
        d[i].lddx % 4)
      return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
  if (!threads) return DA_EINVAL;
#define BN_BWDP_LAUNCH(QB, CH)                                                                 \
  do {                                                                                         \
    if (g_act_bf16) launch_bn_bwd_pair<__bf16, QB>(dout, ldd, d, W, Wn, C, CH, threads, mask, stream); \
    else launch_bn_bwd_pair<float, QB>(dout, ldd, d, W, Wn, C, CH, threads, mask, stream);     \
  } while (0)
  if (cgb == 32) BN_BWDP_LAUNCH(3, 32);
  else if (cgb == 16) BN_BWDP_LAUNCH(2, 16);
  else BN_BWDP_LAUNCH(1, 8);
#undef BN_BWDP_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// da_bn_bwd_mask / da_bn_bwd_pair with the upstream gradient given as TWO terms, dout + dout2 (resnet.py:33-38 backward: the
// gradient of a block's output is the next block's data-gradient conv output plus its identity branch; summing them HERE
// spares the conv an accumulating epilogue).  Single-pass geometry of at most 512 threads (da_bn_two_ok); ds only (fold
// dgamma / dbeta with da_bn_param_grad_multi).
int da_bn_two_ok(int W, int Wn, int C) {
  int cgb = 0;
  const int threads = C % CG == 0 ? bn_fused_geometry(W, Wn, C, &cgb) : 0;
  return threads > 0 && threads <= 512;
}

int da_bn_bwd_mask2(const void* dout, int ldd, const void* dout2, int ldd2, const void* x, int ldx, void* dx, int lddx, void* gout,
                    int ldg, int W, int Wn, int C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                    float* ds, const unsigned long long* mask, hipStream_t stream) {
  DA_ENTER();
  if (!dout || !dout2 || !x || !dx || !mean || !invstd || !gamma || !beta || !ds || !mask || ldd % 4 || ldd2 % 4 || ldx % 4 ||
      lddx % 4 || (gout && ldg % 4) || !da_bn_two_ok(W, Wn, C))
    return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
  BnBwdExt ext = {};
  ext.dout2 = dout2;
  ext.ldd2 = ldd2;
  float* s1 = ds;
  float* s2 = ds + (size_t)W * C;
#define BN_BWD2_LAUNCH(QB, CH)                                                                                           \
  DA_ACT_DISPATCH(hipLaunchKernelGGL((bn_bwd_d2_kernel<AT, FUSED_NPOS, QB>), dim3(W, C / CH), dim3(threads), 0, stream,     \
                                     (const AT*)dout, ldd, (const AT*)x, ldx, (AT*)dx, lddx, (AT*)gout, ldg, Wn, C, mean,     \
                                     invstd, gamma, beta, s1, s2, mask, ext))
  if (cgb == 32) BN_BWD2_LAUNCH(3, 32);
  else if (cgb == 16) BN_BWD2_LAUNCH(2, 16);
  else BN_BWD2_LAUNCH(1, 8);
#undef BN_BWD2_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_bn_bwd_pair2(const void* dout, int ldd, const void* dout2, int ldd2, const da_bn_bwd_desc* d, int W, int Wn, int C,
                    const unsigned long long* mask, hipStream_t stream) {
  DA_ENTER();
  if (!dout || !dout2 || !d || !mask || Wn < 1 || ldd % 4 || ldd2 % 4 || !da_bn_two_ok(W, Wn, C)) return DA_EINVAL;
  for (int i = 0; i < 2; ++i)
    if (!d[i].x || !d[i].dx || !d[i].mean || !d[i].invstd || !d[i].gamma || !d[i].beta || !d[i].ds || d[i].ldx % 4 ||
        d[i].lddx % 4)
      return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
#define BN_BWDP2_LAUNCH(QB, CH)                                                                              \
  do {                                                                                                       \
    if (g_act_bf16) launch_bn_bwd_pair<__bf16, QB>(dout, ldd, d, W, Wn, C, CH, threads, mask, stream, dout2, ldd2); \
    else launch_bn_bwd_pair<float, QB>(dout, ldd, d, W, Wn, C, CH, threads, mask, stream, dout2, ldd2);      \
  } while (0)
  if (cgb == 32) BN_BWDP2_LAUNCH(3, 32);
  else if (cgb == 16) BN_BWDP2_LAUNCH(2, 16);
  else BN_BWDP2_LAUNCH(1, 8);
#undef BN_BWDP2_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// ---- dense-block forms (reference models/densenet.py:18-44,46-66,68-81; float activations, single-pass geometry only) ----
// statistics only: mean / invstd of x[:, 0:C] (pitch ldx) per window into the pitched tables [W][ldstat]
int da_bn_stats_fused(const float* x, int ldx, int W, int Wn, int C, float* mean, float* invstd, int ldstat, float eps,
                      hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;
  if (!x || !mean || !invstd || C % CG || ldx % 4 || ldx < C || ldstat % 4 || ldstat < C || Wn < 1) return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
  if (!threads) return DA_EINVAL;
#define BN_STATS_LAUNCH(QB, CH)                                                                                         \
  hipLaunchKernelGGL((bn_fwd_fused_kernel<float, FUSED_NPOS, QB>), dim3(W, C / CH), dim3(threads), 0, stream, x, ldx,           \
                     (const float*)nullptr, 0, (float*)nullptr, 0, Wn, C, (const float*)nullptr, (const float*)nullptr, 0, eps, \
                     mean, invstd, (unsigned long long*)nullptr, ldstat)
  if (cgb == 32) BN_STATS_LAUNCH(3, 32);
  else if (cgb == 16) BN_STATS_LAUNCH(2, 16);
  else BN_STATS_LAUNCH(1, 8);
#undef BN_STATS_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// out[:, 0:C] = max(fmaf(x, gamma invstd, beta - mean gamma invstd), 0): the activation relu(norm(x)) in the form the
// dense-block kernels apply on the fly (tests / explainers only; the hot path never materialises it)
int da_bn_relu_ss(const float* x, int ldx, float* out, int ldo, int W, int Wn, int C, const float* mean, const float* invstd,
                  int ldstat, const float* gamma, const float* beta, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;
  if (!x || !out || !mean || !invstd || !gamma || !beta || C % 4 || ldx % 4 || ldo % 4 || ldstat % 4 || Wn < 1) return DA_EINVAL;
  if (W == 0) return DA_OK;
  const size_t npos = (size_t)W * Wn;
  if (npos >= 0xffffffffull) return DA_EINVAL;
  size_t g = (npos * (C >> 2) + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(bn_relu_ss_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, ldx, out, ldo, Wn, C, npos, mean, invstd,
                     ldstat, gamma, beta, make_fastdiv((uint32_t)Wn));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// Backward of h = relu(norm(x)) -- relu = 1: the ReLU decision of the scale / shift form (a forward through da_conv1x1_bn);
// relu = 2: the sign of the stored output `out` (a forward through da_bn_fwd) -- or of norm(x) alone (relu = 0), x = the
// first C channels of a dense block's buffer (pitch ldx), statistics from its pitched table:
//   g = dout (half_dout: dout has Wn / 2 positions per window, g[p] = dout[p / 2] / 2) masked by the ReLU decision;
//   dx = BatchNorm input gradient (+ add[:, 0:C], pitch ldadd -- dx may alias add: the in-place accumulation into the
//   block's gradient buffer); then, with drop_p > 0, the dropout mask (seed, salt) of the contiguous [W Wn][drop_g]
//   tensor on the channels [C - drop_g, C) of dx.  ds [2][W][C]: the window sums for da_bn_param_grad_multi.
int da_bn_bwd_ss(const void* dout, int ldd, const void* x, int ldx, const void* out, int ldo, void* dx, int lddx,
                 const void* add, int ldadd, int W, int Wn, int C, const float* mean, const float* invstd, int ldstat,
                 const float* gamma, const float* beta, int relu, int half_dout, const long long* drop_seed, unsigned drop_salt, float drop_p, int drop_g, float* ds,
                 void* hout, int ldh, hipStream_t stream) {
  DA_ENTER();
  if (!dout || !x || !dx || !mean || !invstd || !gamma || !beta || !ds) return DA_EINVAL;
  if (hout && (relu != 1 || ldh % 4 || ldh < C)) return DA_EINVAL;
  if (C % CG || ldd % 4 || ldx % 4 || lddx % 4 || (add && ldadd % 4) || ldstat % 4 || ldstat < C || Wn < 1) return DA_EINVAL;
  if (half_dout && (Wn & 1)) return DA_EINVAL;
  if (relu < 0 || relu > 2 || (relu == 2 && (!out || ldo % 4))) return DA_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && (!drop_seed || drop_g < 4 || drop_g % 4 || drop_g > C))) return DA_EINVAL;
  if (W == 0) return DA_OK;
  int cgb = 0;
  const int threads = bn_fused_geometry(W, Wn, C, &cgb);
  if (!threads) return DA_EINVAL;
  BnBwdExt ext = {};
  ext.ldstat = ldstat; ext.half_dout = half_dout ? 1 : 0; ext.drop_c0 = C - drop_g; ext.drop_g = drop_g;
  ext.seed = drop_seed; ext.salt = drop_salt; ext.p = drop_p; ext.hout = hout; ext.ldh = ldh;
  float* s1 = ds;
  float* s2 = ds + (size_t)W * C;
#define BN_BWDSS_LAUNCH(QB, CH)                                                                                              \
  do {                                                                                                                              \
    if (hout)                                                                                                                       \
      DA_ACT_DISPATCH(hipLaunchKernelGGL((bn_bwd_fused_kernel<AT, FUSED_NPOS, QB, 0, 2>), dim3(W, C / CH), dim3(threads), 0, stream, \
                                         (const AT*)dout, ldd, (const AT*)x, ldx, (const AT*)out, ldo, (AT*)dx, lddx, (AT*)nullptr,  \
                                         0, Wn, C, mean, invstd, gamma, beta, 4, s1, s2, (const AT*)add, ldadd,                      \
                                         (const unsigned long long*)nullptr, ext));                                                  \
    else                                                                                                                            \
      DA_ACT_DISPATCH(hipLaunchKernelGGL((bn_bwd_fused_kernel<AT, FUSED_NPOS, QB, 0, 1>), dim3(W, C / CH), dim3(threads), 0, stream, \
                                         (const AT*)dout, ldd, (const AT*)x, ldx, (const AT*)out, ldo, (AT*)dx, lddx, (AT*)nullptr,  \
                                         0, Wn, C, mean, invstd, gamma, beta, relu == 1 ? 4 : (relu == 2 ? 2 : 0), s1, s2,           \
                                         (const AT*)add, ldadd, (const unsigned long long*)nullptr, ext));                           \
  } while (0)
  if (cgb == 32) BN_BWDSS_LAUNCH(3, 32);
  else if (cgb == 16) BN_BWDSS_LAUNCH(2, 16);
  else BN_BWDSS_LAUNCH(1, 8);
#undef BN_BWDSS_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

}  // extern "C"
