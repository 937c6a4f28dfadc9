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
  const int o = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + o;
  float s = 0.f;
  if (i < n)
    for (int b = slot; b < nblk; b += 32) s += partial[(size_t)b * n + i];
  red[slot][o] = s;
  __syncthreads();
  if (threadIdx.x < 8 && i < n) {
    s = 0.f;
    for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
    dw[i] = accumulate ? dw[i] + s : s;
  }
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
