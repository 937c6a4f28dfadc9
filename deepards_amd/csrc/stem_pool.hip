// Stem convolution (C_in = 1, k7 s2 p3), the stem's BN+ReLU+pool(3,2,1) fusion, and the average pools.
//
// Replaces: reference models/resnet.py:86-87,100-104,141-153 (conv1 -> bn1 -> relu -> first_pool),
// models/densenet.py:118-124 (conv0/norm0/relu0/pool0), resnet.py:112,159 / densenet.py:167,183
// (AvgPool1d(7, stride=1) -> view) and densenet.py:79 (transition AvgPool1d(2,2)), fwd + bwd.
//
// The stem conv has C_in = 1: 7 MACs per output against 4 B written -- pure bandwidth, so it is a
// plain coalesced kernel (lane = output channel, the raw waveform row staged in LDS and read by
// broadcast), not a GEMM.
#include "common.h"

// y[row][l][co] = sum_ci sum_k w[co][ci][k] * x[row][ci][S*l + k - K/2];  block = one (rows, CIN, Lin) input row.
// CIN = 1, K = 7, S = 2 is the reference's default stem (resnet.py:86-87, densenet.py:118-119); CIN = 2 / 3 are the
// DenseNet FFT inputs (densenet.py:109-115: only_fft / with_fft), CIN = 1, K = 3, S = 1 is resnet's conv1_alt
// (resnet.py:88-89,145).  Sums run ci-outer, k-inner (for CIN = 1 the order the one-channel kernel always had).
template <typename AT, int CIN, int K, int S>
__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            AT* __restrict__ y, int Lin, int Lout, int C0, int ldy) {
  extern __shared__ float xs[];  // CIN x (Lin + K - 1)
  constexpr int PAD = K / 2;
  const int row = blockIdx.x, LP = Lin + K - 1;
  for (int i = threadIdx.x; i < CIN * LP; i += blockDim.x) {
    const int ci = i / LP, s = i - ci * LP - PAD;
    xs[i] = (s >= 0 && s < Lin) ? x[((size_t)row * CIN + ci) * Lin + s] : 0.f;
  }
  __syncthreads();
  const int co = threadIdx.x % C0, slot = threadIdx.x / C0, nslots = blockDim.x / C0;
  if (slot >= nslots) return;
  float wk[CIN * K];
#pragma unroll
  for (int k = 0; k < CIN * K; ++k) wk[k] = w[co * CIN * K + k];
  for (int l = slot; l < Lout; l += nslots) {
    float acc = 0.f;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int k = 0; k < K; ++k) acc = fmaf(wk[ci * K + k], xs[ci * LP + S * l + k], acc);
    Act<AT>::st1(y + ((size_t)row * Lout + l) * ldy + co, acc);
  }
}

// partial[blk][co][ci][k] = sum over this block's rows, positions of dy[row][l][co] * x[row][ci][S*l + k - K/2]
template <typename AT, int CIN, int K, int S>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const AT* __restrict__ dy, int lddy,
                                                         const float* __restrict__ x, float* __restrict__ partial,
                                                         int rows, int Lin, int Lout, int C0) {
  extern __shared__ float sm[];  // xs[CIN][Lin+K-1] then red[nslots][C0*CIN*K]
  constexpr int PAD = K / 2, KK = CIN * K;
  const int LP = Lin + K - 1;
  float* xs = sm;
  float* red = sm + CIN * LP;
  const int co = threadIdx.x % C0, slot = threadIdx.x / C0, nslots = blockDim.x / C0;
  float acc[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) acc[k] = 0.f;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    __syncthreads();
    for (int i = threadIdx.x; i < CIN * LP; i += blockDim.x) {
      const int ci = i / LP, s = i - ci * LP - PAD;
      xs[i] = (s >= 0 && s < Lin) ? x[((size_t)row * CIN + ci) * Lin + s] : 0.f;
    }
    __syncthreads();
    if (slot < nslots) {
      for (int l = slot; l < Lout; l += nslots) {
        float g = Act<AT>::ld1(dy + ((size_t)row * Lout + l) * lddy + co);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
          for (int k = 0; k < K; ++k) acc[ci * K + k] = fmaf(g, xs[ci * LP + S * l + k], acc[ci * K + k]);
      }
    }
  }
  __syncthreads();
  if (slot < nslots) {
#pragma unroll
    for (int k = 0; k < KK; ++k) red[(slot * C0 + co) * KK + k] = acc[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C0 * KK; i += blockDim.x) {
    float s = 0.f;
    for (int sl = 0; sl < nslots; ++sl) s += red[sl * C0 * KK + i];
    partial[(size_t)blockIdx.x * C0 * KK + i] = s;
  }
}

// dw[i] (+)= sum over the nblk partial blocks; block = 8 outputs x 32 partial slots, folded through LDS in a fixed
// order (448 outputs: 56 blocks -- the 14-block form took 19 us alone and 80-120 us beside other kernels)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, int n,
                                                                float* __restrict__ dw, int accumulate) {
  __shared__ float red[32][8];
  stem_wgrad_reduce_block(partial, nblk, n, dw, accumulate, blockIdx.x, red);      // (common.h: the step's tail launch runs these too)
}

// out[row][j][c] = pool_{l in {2j-1,2j,2j+1}} relu(bn(y[row][l][c]));  pool_mode 0 = max (-inf pad),
// 1 = avg (count_include_pad, zeros).  One thread per (output position, channel quad).
template <typename AT, int OX3 = 0>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const AT* __restrict__ y, int ldy,
                                                               AT* __restrict__ out, int ldo, int rows, int R,
                                                               int Lin, int Lout, int C, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int pool_mode) {
  const int nq = C >> 2;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)rows * Lout * nq;
  if (idx >= total) return;
  int q = (int)(idx % nq);
  size_t po = idx / nq;
  int j = (int)(po % Lout);
  int row = (int)(po / Lout);
  int w = row / R;
  int c0 = q * 4;
  f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * C + c0);
  f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * C + c0);
  f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
  f32x4 o;
  if (pool_mode == 0) o = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  else o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    int l = 2 * j - 1 + t;
    if (l < 0 || l >= Lin) continue;
    f32x4 v = Act<AT>::ld4(y + ((size_t)row * Lin + l) * ldy + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float z = fmaxf((v[e] - mu[e]) * is[e] * ga[e] + be[e], 0.f);
      o[e] = pool_mode == 0 ? fmaxf(o[e], z) : o[e] + z;
    }
  }
  if (pool_mode == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] *= (1.0f / 3.0f);
  }
  if constexpr (OX3) X3::st4(reinterpret_cast<__bf16*>(out) + po * (size_t)(3 * C), c0, o);   // x3 format (common.h)
  else Act<AT>::st4(out + po * ldo + c0, o);
}

// The same with the stem conv's output RECOMPUTED from the raw rows instead of read: the default stem never stores its
// 36.7 MB conv output.  Block = rows (grid-strided), the row staged in LDS behind STEM_XPAD zeros; a thread owns 4 channels
// (28 taps in registers) and a run of consecutive pool windows j: window j = conv positions 2j-1, 2j, 2j+1, of which 2j-1 is
// the previous step's last -- two new conv outputs per step from a 16-input register window that advances by ONE
// ds_read_b128.  Values: stem_conv_fwd_kernel's and bn_relu_pool_fwd_kernel's, bit for bit (same fmaf chain, same
// expression, max is order-free).  Lc = conv outputs per row, Lp = pooled outputs per row.
#define STEM_XPAD 8                                  // zeros in front of a staged row: xs[STEM_XPAD + s] = x[s]

__device__ __forceinline__ f32x4 stem_y4(const StemW4& sw, const float (&xw)[16], int o) {   // inputs xw[o .. o + 6]
  f32x4 y;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc = fmaf(sw.w[e][k], xw[o + k], acc);
    y[e] = acc;
  }
  return y;
}
// register window of step j: xw[i] = xs[4 j + i] = x[4 j - 8 + i], i = 0 .. 15; conv output l reads inputs 2 l - 3 .. 2 l + 3:
// position 2j-1 starts at xw[3], 2j at xw[5], 2j+1 at xw[7]
__device__ __forceinline__ void stem_xw_fill(float (&xw)[16], const float* xs, int j) {
#pragma unroll
  for (int i = 0; i < 16; i += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xs + 4 * j + i);
    xw[i] = v[0]; xw[i + 1] = v[1]; xw[i + 2] = v[2]; xw[i + 3] = v[3];
  }
}
__device__ __forceinline__ void stem_xw_next(float (&xw)[16], const float* xs, int j) {   // from step j - 1's window
#pragma unroll
  for (int i = 0; i < 12; ++i) xw[i] = xw[i + 4];
  const f32x4 v = *reinterpret_cast<const f32x4*>(xs + 4 * j + 12);
  xw[12] = v[0]; xw[13] = v[1]; xw[14] = v[2]; xw[15] = v[3];
}
__host__ __device__ __forceinline__ int stem_xs_floats(int Lin) { return (STEM_XPAD + Lin + 16 + 3) & ~3; }

struct StemBn4 {
  f32x4 mu, is, ga, be;
  __device__ __forceinline__ f32x4 z(const f32x4& y) const {        // bn_relu_pool_fwd_kernel's expression
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fmaxf((y[e] - mu[e]) * is[e] * ga[e] + be[e], 0.f);
    return r;
  }
};

template <typename AT, int OX3>
__global__ __launch_bounds__(256) void stem_bn_relu_pool_fwd_kernel(const float* __restrict__ xrows, const float* __restrict__ wt,
                                                                    AT* __restrict__ out, int ldo, int rows, int R, int Lin,
                                                                    int Lc, int Lp, int C, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int pool_mode) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // xs[stem_xs_floats(Lin)]
  const int nq = C >> 2, nruns = blockDim.x / nq, RJ = (Lp + nruns - 1) / nruns;
  const int XS = stem_xs_floats(Lin);
  float* xs = sm;
  const int q = threadIdx.x % nq, run = threadIdx.x / nq, c0 = q * 4;
  StemW4 sw;
  sw.load(wt, c0);
  StemBn4 bn;
  bn.ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  bn.be = *reinterpret_cast<const f32x4*>(beta + c0);
  const int j0 = run * RJ, j1 = min(Lp, j0 + RJ);
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int w = row / R;
    __syncthreads();
    for (int i = threadIdx.x; i < XS; i += blockDim.x) {
      const int sx = i - STEM_XPAD;
      xs[i] = (sx >= 0 && sx < Lin) ? xrows[(size_t)row * Lin + sx] : 0.f;
    }
    __syncthreads();
    if (j0 >= j1) continue;
    bn.mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * C + c0);
    bn.is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * C + c0);
    float xw[16];
    stem_xw_fill(xw, xs, j0);
    f32x4 zm = {0.f, 0.f, 0.f, 0.f};
    if (j0 > 0) zm = bn.z(stem_y4(sw, xw, 3));
    for (int j = j0; j < j1; ++j) {
      if (j > j0) stem_xw_next(xw, xs, j);
      const bool vm = j > 0, vp = 2 * j + 1 < Lc;
      const f32x4 z0 = bn.z(stem_y4(sw, xw, 5)), zp = bn.z(stem_y4(sw, xw, 7));
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (pool_mode == 0) {
          float m = vm ? fmaxf(-INFINITY, zm[e]) : -INFINITY;             // the order of bn_relu_pool_fwd_kernel: t = 0, 1, 2
          m = fmaxf(m, z0[e]);
          o[e] = vp ? fmaxf(m, zp[e]) : m;
        } else {
          float a = vm ? 0.f + zm[e] : 0.f;
          a += z0[e];
          if (vp) a += zp[e];
          o[e] = a * (1.0f / 3.0f);
        }
      }
      const size_t po = (size_t)row * Lp + j;
      if constexpr (OX3) X3::st4(reinterpret_cast<__bf16*>(out) + po * (size_t)(3 * C), c0, o);
      else Act<AT>::st4(out + po * ldo + c0, o);
      zm = zp;
    }
  }
}

// Gradient w.r.t. the ReLU output at stem resolution: dz[row][l][c] = sum over the (<= 2) pooling
// windows j that contain l of [argmax_j == l] * dout[row][j][c]  (max; first maximum wins, as ATen)
// or dout[row][j][c]/3 (avg).  The ReLU mask and BN backward are applied afterwards by da_bn_bwd
// (mask_mode 1), so no index tensor is ever stored: the argmax is recomputed from y.
template <typename AT>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const AT* __restrict__ dout, int ldd,
                                                       const AT* __restrict__ y, int ldy, AT* __restrict__ dz,
                                                       int lddz, int rows, int R, int Lin, int Lout, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int pool_mode) {
  const int nq = C >> 2;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)rows * Lin * nq;
  if (idx >= total) return;
  int q = (int)(idx % nq);
  size_t pi = idx / nq;
  int l = (int)(pi % Lin);
  int row = (int)(pi / Lin);
  int w = row / R;
  int c0 = q * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // windows containing l: j with 2j-1 <= l <= 2j+1
  int j_lo = (l) / 2;            // ceil((l-1)/2) for l >= 0
  int j_hi = (l + 1) / 2;        // floor((l+1)/2)
  if (pool_mode == 1) {
    for (int j = j_lo; j <= j_hi; ++j) {
      if (j >= Lout) continue;
      f32x4 g = Act<AT>::ld4(dout + ((size_t)row * Lout + j) * ldd + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += g[e] * (1.0f / 3.0f);
    }
  } else {
    f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * C + c0);
    f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * C + c0);
    f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
    f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0);
    for (int j = j_lo; j <= j_hi; ++j) {
      if (j >= Lout) continue;
      // argmax over t = 0..2 (position 2j-1+t), first max wins
      float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int barg[4] = {-1, -1, -1, -1};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        int ll = 2 * j - 1 + t;
        if (ll < 0 || ll >= Lin) continue;
        f32x4 v = Act<AT>::ld4(y + ((size_t)row * Lin + ll) * ldy + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float z = fmaxf((v[e] - mu[e]) * is[e] * ga[e] + be[e], 0.f);
          if (z > best[e]) {
            best[e] = z;
            barg[e] = ll;
          }
        }
      }
      f32x4 g = Act<AT>::ld4(dout + ((size_t)row * Lout + j) * ldd + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (barg[e] == l) acc[e] += g[e];
    }
  }
  Act<AT>::st4(dz + pi * lddz + c0, acc);
}

// ---- the recomputing stem, backward ------------------------------------------------------------------------------------
// pool backward + BN backward + the stem conv's weight gradient from dout (rows, Lp, C), the RAW rows and the statistics --
// the conv output y, the ReLU decisions and the pool's arg-maxima are recomputed (pool_bwd_kernel already recomputed the
// last two from a stored y), nothing of the 36.7 MB stem resolution is read or written:
//   stem_bwd_kernel<false>  per row: g = dReLU-out (max: the first maximum of each window takes its dout; avg: dout / 3),
//                           masked by the ReLU, and the row's sums of g and g xhat per channel -> rowpart[row][2][C]
//   stem_bwd_kernel<true>   per row: the window totals (its R rows' records, in row order; also stored as the BatchNorm's
//                           ds for dgamma / dbeta), dy = gamma invstd (g - s1 / n - xhat s2 / n) and the weight-gradient
//                           partials sum_l dy[l][c] x[2 l + k - 3] -> partial[block][c][k] (stem_wgrad_reduce_kernel folds them)
// A thread owns 4 channels and a run of consecutive POOL windows j of one row.  Window j covers conv positions 2j-1, 2j, 2j+1:
// per step two new conv outputs (2j, 2j+1; 2j-1 is the previous step's last) from a register window of 11 inputs that
// advances by one ds_read_b128; position 2j-1 is finished by the step of window j (it also sat in window j-1, whose choice
// the step before left in a carry -- a run recomputes the window in front of its first one for it), position 2j by its own.
template <typename AT, bool APPLY>
__global__ __launch_bounds__(256) void stem_bwd_kernel(const AT* __restrict__ dout, int ldd, const float* __restrict__ xrows,
                                                       const float* __restrict__ wt, int rows, int R, int Lin, int Lc, int Lp,
                                                       int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int pool_mode, int RPB, float* __restrict__ rowpart,
                                                       float* __restrict__ ds, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // xs[RS][stem_xs_floats(Lin)] | tot[2][C] | red[slots][APPLY ? 7 C : 2 C]
  // the block's thread slots (blockDim / quads) are RS rows side by side x NRUN runs of pool windows each: 8 runs keep the
  // run-start recompute (the window in front of the run) at 2 of 16 conv outputs
  const int nq = C >> 2, slots = blockDim.x / nq, NRUN = slots < 8 ? slots : 8, RS = slots / NRUN, RJ = (Lp + NRUN - 1) / NRUN;
  const int XS = stem_xs_floats(Lin);
  float* tot = sm + RS * XS;
  float* red = tot + 2 * C;
  const int q = threadIdx.x % nq, slot = threadIdx.x / nq, rs = slot / NRUN, run = slot - rs * NRUN, c0 = q * 4;
  float* xs = sm + rs * XS;                            // xs[STEM_XPAD + s] = x[s] of this thread's row
  StemW4 sw;
  sw.load(wt, c0);
  StemBn4 bn;
  bn.ga = *reinterpret_cast<const f32x4*>(gamma + c0);
  bn.be = *reinterpret_cast<const f32x4*>(beta + c0);
  float wacc[4][7];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 7; ++k) wacc[e][k] = 0.f;
  const float inv_n = 1.0f / (float)(R * Lc);
  const int j0 = run * RJ, j1 = min(Lp, j0 + RJ);
  // a block owns RPB consecutive rows of ONE window (RPB divides R): statistics, totals and taps are fetched once
  const int row_beg = blockIdx.x * RPB, w = row_beg / R;
  if (APPLY && (int)threadIdx.x < 2 * C) {            // the window's totals: its R row records in row order
    const int which = threadIdx.x / C, c = threadIdx.x - which * C;
    float t = 0.f;
    for (int r = 0; r < R; ++r) t += rowpart[((size_t)(w * R + r) * 2 + which) * C + c];
    tot[which * C + c] = t;
    if (ds && row_beg == w * R) ds[((size_t)which * (rows / R) + w) * C + c] = t;
  }
  bn.mu = *reinterpret_cast<const f32x4*>(mean + (size_t)w * C + c0);
  bn.is = *reinterpret_cast<const f32x4*>(invstd + (size_t)w * C + c0);
  f32x4 t1n = {0.f, 0.f, 0.f, 0.f}, t2n = t1n;
  for (int rb = row_beg; rb < row_beg + RPB; rb += RS) {       // RS divides RPB
    const int row = rb + rs;
    __syncthreads();
    for (int i = threadIdx.x; i < RS * XS; i += blockDim.x) {
      const int r = i / XS, sx = i - r * XS - STEM_XPAD;
      sm[i] = (sx >= 0 && sx < Lin) ? xrows[(size_t)(rb + r) * Lin + sx] : 0.f;
    }
    __syncthreads();
    if (APPLY && rb == row_beg) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t1n[e] = tot[c0 + e] * inv_n;
        t2n[e] = tot[C + c0 + e] * inv_n;
      }
    }
    const AT* drow = dout + (size_t)row * Lp * ldd + c0;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float xw[16];                                     // stem_xw_fill: position 2j-1 starts at xw[3], 2j at xw[5], 2j+1 at xw[7]
    // one finished position: its g (the windows' choices), its y, the window index of its first input
    auto finish = [&](const f32x4& gsum, const f32x4& y, int o) {
      f32x4 g, xh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[e] = (y[e] - bn.mu[e]) * bn.is[e];
        g[e] = (xh[e] * bn.ga[e] + bn.be[e] > 0.f) ? gsum[e] : 0.f;      // bn_masked_g, mode 1
      }
      if (!APPLY) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[e] += g[e];
          s2[e] += g[e] * xh[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = bn.ga[e] * bn.is[e] * (g[e] - t1n[e] - xh[e] * t2n[e]);
#pragma unroll
          for (int k = 0; k < 7; ++k) wacc[e][k] = fmaf(d, xw[o + k], wacc[e][k]);
        }
      }
    };
    // the choice of window j among its positions (2j-1: zm, 2j: z0, 2j+1: zp; validity flags): -> (gm, g0, gp) shares of dout[j]
    auto choose = [&](const f32x4& zm, const f32x4& z0, const f32x4& zp, bool vm, bool vp, const f32x4& d, f32x4& gm, f32x4& g0,
                      f32x4& gp) {
      if (pool_mode == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = d[e] * (1.0f / 3.0f);
          gm[e] = vm ? t : 0.f; g0[e] = t; gp[e] = vp ? t : 0.f;
        }
        return;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {                   // first maximum wins (strict >, from -inf), as pool_bwd_kernel / ATen
        float best = -INFINITY;
        int arg = -1;
        if (vm && zm[e] > best) { best = zm[e]; arg = 0; }
        if (z0[e] > best) { best = z0[e]; arg = 1; }
        if (vp && zp[e] > best) { best = zp[e]; arg = 2; }
        gm[e] = arg == 0 ? d[e] : 0.f;
        g0[e] = arg == 1 ? d[e] : 0.f;
        gp[e] = arg == 2 ? d[e] : 0.f;
      }
    };
    if (j0 < j1) {
      // the run starts one window early (when there is one): position 2 j0 - 1 also sits in window j0 - 1, whose choice
      // reaches it through the carry; that warm-up step finishes nothing
      const int js = j0 > 0 ? j0 - 1 : 0;
      stem_xw_fill(xw, xs, js);
      f32x4 carry = {0.f, 0.f, 0.f, 0.f}, ym = carry, zm = carry;
      if (js > 0) {
        ym = stem_y4(sw, xw, 3);
        zm = bn.z(ym);
      }
      for (int j = js; j < j1; ++j) {
        if (j > js) stem_xw_next(xw, xs, j);
        const bool vm = j > 0, vp = 2 * j + 1 < Lc;
        const f32x4 y0 = stem_y4(sw, xw, 5), yp = stem_y4(sw, xw, 7);
        const f32x4 z0 = bn.z(y0), zp = bn.z(yp);
        const f32x4 d = Act<AT>::ld4(drow + (size_t)j * ldd);
        f32x4 gm, g0, gp;
        choose(zm, z0, zp, vm, vp, d, gm, g0, gp);
        if (j >= j0) {
          if (vm) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gm[e] += carry[e];
            finish(gm, ym, 3);
          }
          finish(g0, y0, 5);
          if (j == Lp - 1 && vp) finish(gp, yp, 7);    // the row's last position sits in no further window
        }
        carry = gp; ym = yp; zm = zp;
      }
    }
    if (!APPLY) {                                     // the rows' records: each row's runs folded in run order
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(slot * 2 + 0) * C + c0 + e] = s1[e];
        red[(slot * 2 + 1) * C + c0 + e] = s2[e];
      }
      __syncthreads();
      for (int i = threadIdx.x; i < RS * 2 * C; i += blockDim.x) {
        const int r = i / (2 * C), t_ = i - r * 2 * C;
        float t = 0.f;
        for (int u = 0; u < NRUN; ++u) t += red[(r * NRUN + u) * 2 * C + t_];
        rowpart[(size_t)(rb + r) * 2 * C + t_] = t;
      }
    }
  }
  if (APPLY) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 7; ++k) red[(slot * C + c0 + e) * 7 + k] = wacc[e][k];
    __syncthreads();
    for (int i = threadIdx.x; i < C * 7; i += blockDim.x) {
      float t = 0.f;
      for (int r = 0; r < slots; ++r) t += red[r * C * 7 + i];
      partial[(size_t)blockIdx.x * C * 7 + i] = t;
    }
  }
}

// generic AvgPool1d(k, stride=k) (k=2 transition) and AvgPool1d(L, 1) on an L-long row (k = L -> 1).
template <typename IT, typename OT>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const IT* __restrict__ x, int ldx, OT* __restrict__ out,
                                                          int ldo, int rows, int Lin, int Lout, int k, int C) {
  const int nq = C >> 2;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)rows * Lout * nq;
  if (idx >= total) return;
  int q = (int)(idx % nq);
  size_t po = idx / nq;
  int j = (int)(po % Lout);
  int row = (int)(po / Lout);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < k; ++t) {
    f32x4 v = Act<IT>::ld4(x + ((size_t)row * Lin + j * k + t) * ldx + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += v[e];
  }
  const float inv = 1.0f / (float)k;
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] *= inv;
  Act<OT>::st4(out + po * ldo + q * 4, acc);
}

template <typename IT, typename OT>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const IT* __restrict__ dout, int ldd, OT* __restrict__ dx,
                                                          int lddx, int rows, int Lin, int Lout, int k, int C) {
  const int nq = C >> 2;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)rows * Lin * nq;
  if (idx >= total) return;
  int q = (int)(idx % nq);
  size_t pi = idx / nq;
  int l = (int)(pi % Lin);
  int row = (int)(pi / Lin);
  int j = l / k;
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
  if (j < Lout) {
    g = Act<IT>::ld4(dout + ((size_t)row * Lout + j) * ldd + q * 4);
    const float inv = 1.0f / (float)k;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] *= inv;
  }
  Act<OT>::st4(dx + pi * lddx + q * 4, g);
}

// AvgPool1d(k, stride=1) on a map longer than k, flattened the way `x.view(x.size(0), -1)` flattens (N, C, Lout):
// feature index c * Lout + j (resnet.py:159-160 / densenet.py:183-184 on seq_len > 224, e.g. BASELINE config C5's 512).
template <typename AT>
__global__ __launch_bounds__(256) void avgpool_slide_fwd_kernel(const AT* __restrict__ x, int ldx,
                                                                float* __restrict__ feat, int rows, int Lin, int Lout,
                                                                int k, int C) {
  const int nq = C >> 2;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)rows * Lout * nq;
  if (idx >= total) return;
  int q = (int)(idx % nq);
  size_t po = idx / nq;
  int j = (int)(po % Lout);
  int row = (int)(po / Lout);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < k; ++t) {
    f32x4 v = Act<AT>::ld4(x + ((size_t)row * Lin + j + t) * ldx + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += v[e];
  }
  const float inv = 1.0f / (float)k;
  float* o = feat + (size_t)row * C * Lout + (size_t)(q * 4) * Lout + j;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[(size_t)e * Lout] = acc[e] * inv;
}

template <typename AT>
__global__ __launch_bounds__(256) void avgpool_slide_bwd_kernel(const float* __restrict__ dfeat, AT* __restrict__ dx,
                                                                int lddx, int rows, int Lin, int Lout, int k, int C) {
  const int nq = C >> 2;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)rows * Lin * nq;
  if (idx >= total) return;
  int q = (int)(idx % nq);
  size_t pi = idx / nq;
  int l = (int)(pi % Lin);
  int row = (int)(pi / Lin);
  int j0 = l - k + 1 > 0 ? l - k + 1 : 0;
  int j1 = l < Lout - 1 ? l : Lout - 1;
  const float* d = dfeat + (size_t)row * C * Lout + (size_t)(q * 4) * Lout;
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
  for (int j = j0; j <= j1; ++j) {
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] += d[(size_t)e * Lout + j];
  }
  const float inv = 1.0f / (float)k;
#pragma unroll
  for (int e = 0; e < 4; ++e) g[e] *= inv;
  Act<AT>::st4(dx + pi * lddx + q * 4, g);
}

static inline int grid1d(size_t total, int bs) { return (int)((total + bs - 1) / bs); }

extern "C" {

// x: [rows][Lin] raw waveform (C_in = 1).  w: [C0][1][7] (torch layout).  y: [rows][Lin/2][ldy].
// the stem shapes this library instantiates: (Cin, K, stride) with pad = K / 2
#define DA_STEM_DISPATCH(CIN_, K_, S_, STMT)                                   \
  do {                                                                         \
    if ((CIN_) == 1 && (K_) == 7 && (S_) == 2) { constexpr int CIN = 1, KS = 7, SS = 2; STMT; }        \
    else if ((CIN_) == 2 && (K_) == 7 && (S_) == 2) { constexpr int CIN = 2, KS = 7, SS = 2; STMT; }   \
    else if ((CIN_) == 3 && (K_) == 7 && (S_) == 2) { constexpr int CIN = 3, KS = 7, SS = 2; STMT; }   \
    else if ((CIN_) == 1 && (K_) == 3 && (S_) == 1) { constexpr int CIN = 1, KS = 3, SS = 1; STMT; }   \
    else return DA_EINVAL;                                                     \
  } while (0)

int da_stem_conv_fwd_g(const float* x, const float* w, void* y, int rows, int Lin, int Cin, int K, int stride, int C0,
                       int ldy, hipStream_t stream) {
  DA_ENTER();
  if (!x || !w || !y || Lin < 2 || Lin % stride || C0 < 1 || C0 > 256 || 256 % C0 || stride < 1) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const int Lout = Lin / stride;                               // (Lin + 2 (K/2) - K) / stride + 1 for odd K, stride | Lin
  const size_t shm = (size_t)Cin * (Lin + K - 1) * sizeof(float);
  DA_STEM_DISPATCH(Cin, K, stride,
                   DA_ACT_DISPATCH(hipLaunchKernelGGL((stem_conv_fwd_kernel<AT, CIN, KS, SS>), dim3(rows), dim3(256), shm,
                                                      stream, x, w, (AT*)y, Lin, Lout, C0, ldy)));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_stem_conv_fwd(const float* x, const float* w, void* y, int rows, int Lin, int C0, int ldy,
                     hipStream_t stream) {
  return da_stem_conv_fwd_g(x, w, y, rows, Lin, 1, 7, 2, C0, ldy, stream);
}

// da_bn_relu_pool_fwd with the output stored in the x3 format (float activations; the input of layer1's k3 s1 convs under
// conv arithmetic 'f32x3'): out = [rows * Lout] positions of 3 C bf16.
int da_bn_relu_pool_fwd_x(const float* y, int ldy, void* out, int rows, int R, int Lin, int C, const float* mean,
                          const float* invstd, const float* gamma, const float* beta, int pool_mode, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;
  if (!y || !out || C % 16 || ldy % 4 || R < 1 || rows % R) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  int Lout = (Lin - 1) / 2 + 1;
  size_t total = (size_t)rows * Lout * (C / 4);
  hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<float, 1>), dim3(grid1d(total, 256)), dim3(256), 0, stream, y, ldy, (float*)out, 0,
                     rows, R, Lin, Lout, C, mean, invstd, gamma, beta, pool_mode);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

size_t da_stem_wgrad_workspace_g(int rows, int C0, int Cin, int K) {
  int nblk = rows < 512 ? rows : 512;
  return (size_t)nblk * C0 * Cin * K * sizeof(float);
}

size_t da_stem_wgrad_workspace(int rows, int C0) { return da_stem_wgrad_workspace_g(rows, C0, 1, 7); }

int da_stem_conv_wgrad_g(const void* dy, int lddy, const float* x, float* dw, float* workspace, int rows, int Lin,
                         int Cin, int K, int stride, int C0, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (!dy || !x || !dw || !workspace || stride < 1 || Lin % stride || C0 < 1 || C0 > 256 || 256 % C0) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  int nblk = rows < 512 ? rows : 512;
  int nslots = 256 / C0;
  const int Lout = Lin / stride;
  size_t shm = ((size_t)Cin * (Lin + K - 1) + (size_t)nslots * C0 * Cin * K) * sizeof(float);
  if (shm > 64 * 1024) return DA_EINVAL;
  DA_STEM_DISPATCH(Cin, K, stride,
                   DA_ACT_DISPATCH(hipLaunchKernelGGL((stem_wgrad_kernel<AT, CIN, KS, SS>), dim3(nblk), dim3(256), shm, stream,
                                                      (const AT*)dy, lddy, x, workspace, rows, Lin, Lout, C0)));
  DA_CHECK_LAUNCH();
  int n = C0 * Cin * K;
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((n + 7) / 8), dim3(256), 0, stream, workspace, nblk, n, dw,
                     accumulate);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_stem_conv_wgrad(const void* dy, int lddy, const float* x, float* dw, float* workspace, int rows, int Lin,
                       int C0, int accumulate, hipStream_t stream) {
  return da_stem_conv_wgrad_g(dy, lddy, x, dw, workspace, rows, Lin, 1, 7, 2, C0, accumulate, stream);
}

// Lout = (Lin + 2 - 3)/2 + 1.  R = rows per BN window.  pool_mode 0 max / 1 avg.
int da_bn_relu_pool_fwd(const void* y, int ldy, void* out, int ldo, int rows, int R, int Lin, int C,
                        const float* mean, const float* invstd, const float* gamma, const float* beta, int pool_mode,
                        hipStream_t stream) {
  DA_ENTER();
  if (!y || !out || C % 4 || ldy % 4 || ldo % 4 || R < 1 || rows % R) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  int Lout = (Lin - 1) / 2 + 1;
  size_t total = (size_t)rows * Lout * (C / 4);
  DA_ACT_DISPATCH(hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<AT>, dim3(grid1d(total, 256)), dim3(256), 0, stream, (const AT*)y,
                                     ldy, (AT*)out, ldo, rows, R, Lin, Lout, C, mean, invstd, gamma, beta, pool_mode));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// BN + ReLU + pool(3,2,1) of the default stem from the RAW rows (the conv output recomputed, never stored): xrows (rows, Lin),
// wt (C, 1, 7) -> out (rows, Lp, C) in the activation storage type, or the x3 format when out_x3 (float storage only).  mean / invstd: da_stem_stats_partial + da_bn_stats_merge.
int da_stem_bn_relu_pool_fwd(const float* xrows, const float* wt, void* out, int ldo, int rows, int R, int Lin, int C,
                             const float* mean, const float* invstd, const float* gamma, const float* beta, int pool_mode,
                             int out_x3, hipStream_t stream) {
  DA_ENTER();
  if (!xrows || !wt || !out || C % 4 || (out_x3 ? C % 16 : ldo % 4) || R < 1 || rows % R || Lin < 2 || (Lin & 1) ||
      (g_act_bf16 && out_x3))
    return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const int Lc = Lin / 2, Lp = (Lc - 1) / 2 + 1;
  if (256 % (C / 4)) return DA_EINVAL;
  const size_t shm = (size_t)stem_xs_floats(Lin) * sizeof(float);
  const int nblk = rows < 1024 ? rows : 1024;
  if (out_x3)
    hipLaunchKernelGGL((stem_bn_relu_pool_fwd_kernel<float, 1>), dim3(nblk), dim3(256), shm, stream, xrows, wt, (float*)out, ldo,
                       rows, R, Lin, Lc, Lp, C, mean, invstd, gamma, beta, pool_mode);
  else
    DA_ACT_DISPATCH(hipLaunchKernelGGL((stem_bn_relu_pool_fwd_kernel<AT, 0>), dim3(nblk), dim3(256), shm, stream, xrows, wt,
                                       (AT*)out, ldo, rows, R, Lin, Lc, Lp, C, mean, invstd, gamma, beta, pool_mode));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// where da_stem_bwd(dw = NULL) leaves its weight-gradient partials: offset (floats) into the workspace and their count --
// [nblk][C * 7], to be summed over nblk into dw (C, 1, 7) by da_step_tail_multi (or da_stem_wgrad_reduce)
int da_stem_bwd_partials(int rows, int R, int C, size_t* offset, int* nblk) {
  if (!offset || !nblk || C % 4 || C < 4 || 256 % (C / 4) || R < 1 || rows % R) return DA_EINVAL;
  const int nruns = 256 / (C / 4), NRUN = nruns < 8 ? nruns : 8, RS = nruns / NRUN;
  if (R % RS) return DA_EINVAL;
  int RPB = RS;
  for (int d = RS; d <= 5; d += RS)
    if (R % d == 0) RPB = d;
  *offset = (size_t)rows * 2 * C;
  *nblk = rows / RPB;
  return DA_OK;
}

// floats of scratch da_stem_bwd needs: the row records (rows x 2 C) and the weight-gradient partials (blocks x 7 C)
size_t da_stem_bwd_workspace(int rows, int C) {     // (one partial per block; a block owns >= 1 row)
  return ((size_t)rows * 2 * C + (size_t)rows * 7 * C) * sizeof(float);
}

// dw[n] (+)= sum over nblk of partial[nblk][n]: the fold da_stem_bwd(dw = NULL) left out
int da_stem_wgrad_reduce(const float* partial, int nblk, int n, float* dw, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (!partial || !dw || nblk < 1 || n < 1) return DA_EINVAL;
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((n + 7) / 8), dim3(256), 0, stream, partial, nblk, n, dw, accumulate);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// Backward of the default stem (conv k7 s2 p3 on one channel -> BN -> ReLU -> pool(3,2,1)) from the RAW rows: dout
// (rows, Lp, C) in the activation storage type -> dw (C, 1, 7) (+= when accumulate) and ds (2, W, C): the BatchNorm's window sums of g / g xhat
// (da_bn_param_grad_multi folds them into dgamma / dbeta).  Nothing at the stem's resolution is read or written.
int da_stem_bwd(const void* dout, int ldd, const float* xrows, const float* wt, int rows, int R, int Lin, int C,
                const float* mean, const float* invstd, const float* gamma, const float* beta, int pool_mode, float* ds,
                float* dw, int accumulate, float* workspace, hipStream_t stream) {
  DA_ENTER();
  if (!dout || !xrows || !wt || !mean || !invstd || !gamma || !beta || !ds || !workspace || C % 4 || C < 4 ||
      256 % (C / 4) || 2 * C > 256 || ldd % 4 || R < 1 || rows % R || Lin < 2 || (Lin & 1))
    return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const int Lc = Lin / 2, Lp = (Lc - 1) / 2 + 1, nruns = 256 / (C / 4);
  float* rowpart = workspace;
  float* partial = workspace + (size_t)rows * 2 * C;
  const int NRUN = nruns < 8 ? nruns : 8, RS = nruns / NRUN;       // the kernel's slot split (rows side by side x runs)
  if (R % RS) return DA_EINVAL;
  int RPB = RS;                                       // rows per block: a multiple of RS that divides R, up to 5 (or RS)
  for (int d = RS; d <= 5; d += RS)
    if (R % d == 0) RPB = d;
  const int nblk = rows / RPB;
  const size_t xsn = (size_t)RS * stem_xs_floats(Lin);
  const size_t shm1 = (xsn + 2 * C + (size_t)nruns * 2 * C) * sizeof(float);
  const size_t shm2 = (xsn + 2 * C + (size_t)nruns * 7 * C) * sizeof(float);
  if (shm2 > 64 * 1024) return DA_EINVAL;
  DA_ACT_DISPATCH(hipLaunchKernelGGL((stem_bwd_kernel<AT, false>), dim3(nblk), dim3(256), shm1, stream, (const AT*)dout, ldd, xrows,
                                     wt, rows, R, Lin, Lc, Lp, C, mean, invstd, gamma, beta, pool_mode, RPB, rowpart,
                                     (float*)nullptr, (float*)nullptr));
  DA_CHECK_LAUNCH();
  DA_ACT_DISPATCH(hipLaunchKernelGGL((stem_bwd_kernel<AT, true>), dim3(nblk), dim3(256), shm2, stream, (const AT*)dout, ldd, xrows,
                                     wt, rows, R, Lin, Lc, Lp, C, mean, invstd, gamma, beta, pool_mode, RPB, rowpart, ds, partial));
  DA_CHECK_LAUNCH();
  if (!dw) return DA_OK;       // the caller folds the partials later (da_step_tail_multi: da_stem_bwd_partials() says where they are)
  const int n = C * 7;
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((n + 7) / 8), dim3(256), 0, stream, partial, nblk, n, dw, accumulate);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_pool_bwd(const void* dout, int ldd, const void* y, int ldy, void* dz, int lddz, int rows, int R, int Lin,
                int C, const float* mean, const float* invstd, const float* gamma, const float* beta, int pool_mode,
                hipStream_t stream) {
  DA_ENTER();
  if (!dout || !y || !dz || C % 4 || ldd % 4 || ldy % 4 || lddz % 4 || R < 1 || rows % R) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  int Lout = (Lin - 1) / 2 + 1;
  size_t total = (size_t)rows * Lin * (C / 4);
  DA_ACT_DISPATCH(hipLaunchKernelGGL(pool_bwd_kernel<AT>, dim3(grid1d(total, 256)), dim3(256), 0, stream, (const AT*)dout, ldd,
                                     (const AT*)y, ldy, (AT*)dz, lddz, rows, R, Lin, Lout, C, mean, invstd, gamma, beta,
                                     pool_mode));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// AvgPool1d(k, stride=k) with Lout = Lin / k (floor); k == Lin gives the global pool.
int da_avgpool_fwd(const float* x, int ldx, float* out, int ldo, int rows, int Lin, int k, int C,
                   hipStream_t stream) {
  DA_ENTER();
  if (!x || !out || C % 4 || ldx % 4 || ldo % 4 || k < 1 || k > Lin) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  int Lout = Lin / k;
  size_t total = (size_t)rows * Lout * (C / 4);
  hipLaunchKernelGGL((avgpool_fwd_kernel<float, float>), dim3(grid1d(total, 256)), dim3(256), 0, stream, x, ldx, out, ldo,
                     rows, Lin, Lout, k, C);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// AvgPool1d(k, stride 1) + view(rows, -1): feat is (rows, C * (Lin - k + 1)) with the channel index slowest.
int da_avgpool_slide_fwd(const void* x, int ldx, float* feat, int rows, int Lin, int k, int C, hipStream_t stream) {
  DA_ENTER();
  if (!x || !feat || C % 4 || ldx % 4 || k < 1 || k > Lin) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  int Lout = Lin - k + 1;
  size_t total = (size_t)rows * Lout * (C / 4);
  DA_ACT_DISPATCH(hipLaunchKernelGGL(avgpool_slide_fwd_kernel<AT>, dim3(grid1d(total, 256)), dim3(256), 0, stream, (const AT*)x,
                                     ldx, feat, rows, Lin, Lout, k, C));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_avgpool_slide_bwd(const float* dfeat, void* dx, int lddx, int rows, int Lin, int k, int C,
                         hipStream_t stream) {
  DA_ENTER();
  if (!dfeat || !dx || C % 4 || lddx % 4 || k < 1 || k > Lin) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  int Lout = Lin - k + 1;
  size_t total = (size_t)rows * Lin * (C / 4);
  DA_ACT_DISPATCH(hipLaunchKernelGGL(avgpool_slide_bwd_kernel<AT>, dim3(grid1d(total, 256)), dim3(256), 0, stream, dfeat,
                                     (AT*)dx, lddx, rows, Lin, Lout, k, C));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_avgpool_bwd(const float* dout, int ldd, float* dx, int lddx, int rows, int Lin, int k, int C,
                   hipStream_t stream) {
  DA_ENTER();
  if (!dout || !dx || C % 4 || ldd % 4 || lddx % 4 || k < 1 || k > Lin) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  int Lout = Lin / k;
  size_t total = (size_t)rows * Lin * (C / 4);
  hipLaunchKernelGGL((avgpool_bwd_kernel<float, float>), dim3(grid1d(total, 256)), dim3(256), 0, stream, dout, ldd, dx, lddx,
                     rows, Lin, Lout, k, C);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// The activation -> feature boundary: AvgPool1d(L, stride 1) on an L-long map + view, x in the activation storage type,
// features always float (and back).  replaces reference models/resnet.py:112,159-160 / densenet.py:167,183-184
int da_global_avgpool_fwd(const void* x, int ldx, float* feat, int rows, int L, int C, hipStream_t stream) {
  DA_ENTER();
  if (!x || !feat || C % 4 || ldx % 4 || L < 1) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  size_t total = (size_t)rows * (C / 4);
  DA_ACT_DISPATCH(hipLaunchKernelGGL((avgpool_fwd_kernel<AT, float>), dim3(grid1d(total, 256)), dim3(256), 0, stream,
                                     (const AT*)x, ldx, feat, C, rows, L, 1, L, C));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_global_avgpool_bwd(const float* dfeat, void* dx, int lddx, int rows, int L, int C, hipStream_t stream) {
  DA_ENTER();
  if (!dfeat || !dx || C % 4 || lddx % 4 || L < 1) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  size_t total = (size_t)rows * L * (C / 4);
  DA_ACT_DISPATCH(hipLaunchKernelGGL((avgpool_bwd_kernel<float, AT>), dim3(grid1d(total, 256)), dim3(256), 0, stream, dfeat, C,
                                     (AT*)dx, lddx, rows, L, 1, L, C));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

}  // extern "C"
