// Shared device/host helpers for the deepards MI355X (gfx950) kernels.
//
// Data layout used by every kernel in this directory ("RLC", channels-last rows):
//   activation[row][l][c]   row = window*rows_per_window + sub_batch_row, l = sample position,
//                           c = channel; channel pitch `ld` floats (>= C, multiple of 4).
// A *window* (the unit BatchNorm statistics are taken over, reference
// models/torch_cnn_linear_network.py:108-113) is rows_per_window consecutive rows, i.e. one
// contiguous [rows_per_window*L][ld] slab of HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

#define DA_OK 0
#define DA_EINVAL (-1)

// Row-partitioned kernels (BatchNorm, pools: one block per window of rows) take their window from the block index the way
// the convolutions take their tiles (conv_gemm.hip xcd_chunked): the blocks that the hardware places on XCD k (block id
// mod 8) work on the k-th EIGHTH of the rows, so the activation rows a convolution's XCD left in its 4 MB L2 are read by
// blocks of the same XCD, and what they write waits in the L2 the next convolution's tiles of those rows run on.
#ifndef DA_ROW_XCD
#define DA_ROW_XCD 1
#endif
__device__ __forceinline__ int row_xcd_chunk(int id, int total) {
#if DA_ROW_XCD
  if (total & 7) return id;
  return (id & 7) * (total >> 3) + (id >> 3);
#else
  return id;
#endif
}

// Activation storage type.  Every RLC activation / activation-gradient tensor that crosses a kernel boundary is either
// float (default) or bf16 (da_set_act_dtype(1): BASELINE's bf16 configs; statistics, sums, parameters and their
// gradients stay float).  Kernels that touch activations are templated on AT and launched through DA_ACT_DISPATCH;
// all arithmetic is fp32 either way -- bf16 is a storage format (round-to-nearest-even on store, exact widening on load).
template <typename T> struct Act;
template <> struct Act<float> {
  static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Act<__bf16> {
  static __device__ __forceinline__ f32x4 ld4(const __bf16* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                 __uint_as_float(r.y & 0xffff0000u)};
  }
  static __device__ __forceinline__ void st4(__bf16* p, const f32x4& v) {
    const f32x2v lo = {v[0], v[1]}, hi = {v[2], v[3]};
    const bf16x2v a = __builtin_convertvector(lo, bf16x2v), b = __builtin_convertvector(hi, bf16x2v);   // v_cvt_pk_bf16_f32
    *reinterpret_cast<f32x2v*>(p) = f32x2v{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
  }
  static __device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }
};
// The "x3" activation format (conv arithmetic 'f32x3'): an fp32 value stored as its EXACT three-term bf16 split
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)        (round-to-nearest-even, 3 x 8 bits)
// so that the convolutions that consume it can take fp32-equivalent products on the bf16 matrix cores (conv_x3p.hip)
// without splitting anything themselves.  Layout per position: C/16 groups of [h 16 ch | m 16 ch | l 16 ch] = 48 bf16;
// 3 C bf16 = 6 C bytes per position, no channel pitch.  Reading it back (residual adds, tests) is h + (m + l): exact for
// |x| >= 2^-109 (below that the third term underflows bf16's denormals: absolute error < 2^-133).  Only the BatchNorm / pool kernels in front of an x3
// consumer store it; statistics, sums and every conv OUTPUT stay plain fp32.
struct X3 {
  static __device__ __forceinline__ f32x2v cvt4(const f32x4& v) {      // 4 bf16 (nearest-even) as the bits of 2 floats
    const f32x2v lo = {v[0], v[1]}, hi = {v[2], v[3]};
    const bf16x2v a = __builtin_convertvector(lo, bf16x2v), b = __builtin_convertvector(hi, bf16x2v);
    return f32x2v{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
  }
  static __device__ __forceinline__ f32x4 widen4(const f32x2v& b) {    // the 4 bf16 back as floats (exact)
    const uint32_t u0 = __float_as_uint(b[0]), u1 = __float_as_uint(b[1]);
    return f32x4{__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 << 16),
                 __uint_as_float(u1 & 0xffff0000u)};
  }
  // element offset (bf16) of channel c0's h term inside a position; m at + 16, l at + 32
  static __device__ __forceinline__ int off(int c0) { return (c0 >> 4) * 48 + (c0 & 15); }
  static __device__ __forceinline__ void st4(__bf16* pos, int c0, const f32x4& v) {     // channels c0 .. c0 + 3, c0 % 4 == 0
    const f32x2v h = cvt4(v);
    const f32x4 r1 = v - widen4(h);
    const f32x2v m = cvt4(r1);
    const f32x4 r2 = r1 - widen4(m);
    __bf16* d = pos + off(c0);
    *reinterpret_cast<f32x2v*>(d) = h;
    *reinterpret_cast<f32x2v*>(d + 16) = m;
    *reinterpret_cast<f32x2v*>(d + 32) = cvt4(r2);
  }
  static __device__ __forceinline__ f32x4 ld4(const __bf16* pos, int c0) {
    const __bf16* d = pos + off(c0);
    const f32x4 h = widen4(*reinterpret_cast<const f32x2v*>(d)), m = widen4(*reinterpret_cast<const f32x2v*>(d + 16)),
                l = widen4(*reinterpret_cast<const f32x2v*>(d + 32));
    return h + (m + l);                              // m + l = x - h exactly (<= 17 bits), then h + (x - h) = x
  }
};

// a value as the activation storage type would hold it (float: itself; bf16: rounded to nearest even and widened back)
template <typename AT>
__device__ __forceinline__ f32x4 act_round4(const f32x4& v) {
  if constexpr (sizeof(AT) == 4) return v;
  else return X3::widen4(X3::cvt4(v));
}

// The default stem (conv k7 s2 p3 on ONE input channel: reference models/resnet.py:86-87, densenet.py:118-119) costs 7 FMAs per
// output and its output is 36.7 MB at B = 64 -- the "recomputing stem" kernels never store it: they take the raw rows and the
// 64 x 7 weights and recompute an output wherever one is needed.  This is stem_conv_fwd_kernel's value BIT FOR BIT: the same
// fmaf chain k = 0 .. 6 from zero, zero padding entering as fmaf(w, 0, acc) = acc.
struct StemW4 {
  float w[4][7];                       // taps of channels c0 .. c0 + 3
  __device__ __forceinline__ void load(const float* __restrict__ wt, int c0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 7; ++k) w[e][k] = wt[(c0 + e) * 7 + k];
  }
  // y[l][c0 .. c0 + 3] of one row; xr = the row (Lin floats, global or LDS), positions outside [0, Lin) are zero
  __device__ __forceinline__ f32x4 at(const float* __restrict__ xr, int Lin, int l) const {
    float xv[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int s = 2 * l + k - 3;
      xv[k] = (s >= 0 && s < Lin) ? xr[s] : 0.f;
    }
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 7; ++k) acc = fmaf(w[e][k], xv[k], acc);
      y[e] = acc;
    }
    return y;
  }
};

extern int g_act_bf16;                 // head_optim.hip; set by da_set_act_dtype
// run STMT once with `AT` = the current activation storage type
#define DA_ACT_DISPATCH(STMT)  \
  do {                         \
    if (g_act_bf16) {          \
      typedef __bf16 AT;       \
      STMT;                    \
    } else {                   \
      typedef float AT;        \
      STMT;                    \
    }                          \
  } while (0)

// hipGetLastError() is sticky per thread: clear whatever an earlier runtime call (ours or PyTorch's)
// left behind before judging our own launches.
#define DA_ENTER() (void)hipGetLastError()

#define DA_CHECK_LAUNCH()                          \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

// q = m / d for m*d < 2^32 via one v_mul_hi_u32 (magic = floor(2^32/d)+1, d >= 2; d == 1 handled).
struct FastDiv {
  uint32_t magic;
  uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.magic = d <= 1 ? 0u : (uint32_t)((0x100000000ull / d) + 1ull);
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t m, FastDiv f) {
  return f.d <= 1 ? m : __umulhi(m, f.magic);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// BatchNorm + ReLU as ONE fused multiply-add per element: h = max(fmaf(x, sc, sh), 0) with sc = gamma invstd,
// sh = beta - mean sc.  The dense-block path (models/densenet.py:18-44,68-81) never stores relu(norm(x)): the convolutions
// apply this while they stage their operand, the weight gradients do the same, and the BatchNorm backward recomputes the
// ReLU decision from it -- every one of them through THIS function on the same (x, mean, invstd, gamma, beta) floats, so
// they agree bit for bit on which elements are active.
__device__ __forceinline__ void bn_scale_shift(float mean, float invstd, float gamma, float beta, float& sc, float& sh) {
  sc = gamma * invstd;
  sh = fmaf(-mean, sc, beta);
}

// Statistics records (include/deepards_hip.h): (mean, invstd) of window w, channel c (of the record's nc) from the 64-unit
// tiles' records [tiles][2 slots][{mean, M2}][nc] | counts [tiles][2]: Chan's update in tile order.  Every block of a
// consuming conv runs this for its own windows: the records are loaded MERGE_B at a time, the update runs on registers.
// A tile whose first unit lies in front of the window started in the previous one: the window is its SECOND slot.
// (Measured alternatives, whole densenet18 step at B = 64 / B = 16: this form 1.287 / 0.791 ms; a two-pass weighted mean
// without divisions 1.311 / 0.813; the records staged through LDS by the whole block in one sweep 1.308 / 0.813.)
template <int MERGE_B = 4>
__device__ __forceinline__ void merge_stat_records(const float* __restrict__ part, int tiles, int nc, int Wu, int w, int c,
                                                   float eps, float& mean, float& invstd) {
  const int u0 = w * Wu;
  const int r0 = u0 >> 6, r1 = min((u0 + Wu - 1) >> 6, tiles - 1);
  const float* cnt = part + (size_t)tiles * 4 * nc;
  float n = 0.f, mu = 0.f, m2 = 0.f;
#pragma unroll 1
  for (int rb = r0; rb <= r1; rb += MERGE_B) {
    float cn[MERGE_B], mb[MERGE_B], qb[MERGE_B];
#pragma unroll
    for (int j = 0; j < MERGE_B; ++j) {
      const int r = min(rb + j, r1);
      const int sl = (r << 6) >= u0 ? 0 : 1;
      const float* rec = part + ((size_t)(r * 2 + sl) * 2) * nc + c;
      cn[j] = cnt[r * 2 + sl];
      mb[j] = rec[0];
      qb[j] = rec[nc];
    }
#pragma unroll
    for (int j = 0; j < MERGE_B; ++j) {
      const float nb = rb + j <= r1 ? cn[j] : 0.f;
      if (nb > 0.f) {
        const float d = mb[j] - mu, nt = n + nb;
        mu += d * (nb / nt);
        m2 += qb[j] + d * d * (n * nb / nt);
        n = nt;
      }
    }
  }
  mean = mu;
  invstd = 1.0f / sqrtf(m2 / fmaxf(n, 1.f) + eps);
}

// One block of the fold of the stem's weight-gradient partials [nblk][n] -> dw[n] (stem_pool.hip stem_wgrad_reduce_kernel):
// 8 outputs x 32 slots, fixed order.
__device__ __forceinline__ void stem_wgrad_reduce_block(const float* __restrict__ partial, int nblk, int n,
                                                        float* __restrict__ dw, int accumulate, int blk, float (*red)[8]) {
  const int o = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int i = blk * 8 + o;
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

// One block of the dgamma / dbeta fold of a BatchNorm (bn.hip bn_param_grad_multi_kernel; conv_gemm.hip's slab reduction runs
// the same blocks beside its own): channels [32 chunk, 32 chunk + 32), 8 window slots, fixed order.
struct BnPgradDesc {
  const float* s1;
  const float* s2;
  float* dgamma;
  float* dbeta;
  int W, C;
};
__device__ __forceinline__ void bn_param_grad_block(const BnPgradDesc& d, int chunk, int accumulate, float (*red)[8][32]) {
  if (chunk * 32 >= d.C) return;                       // (block-uniform)
  const int c = chunk * 32 + (threadIdx.x & 31), slot = threadIdx.x >> 5;
  float a = 0.f, b = 0.f;
  if (c < d.C) {
    for (int w = slot; w < d.W; w += 8) {
      a += d.s1[(size_t)w * d.C + c];
      b += d.s2[(size_t)w * d.C + c];
    }
  }
  red[0][slot][threadIdx.x & 31] = a;
  red[1][slot][threadIdx.x & 31] = b;
  __syncthreads();
  if (threadIdx.x < 32 && c < d.C) {
    a = 0.f;
    b = 0.f;
    for (int k = 0; k < 8; ++k) {
      a += red[0][k][threadIdx.x];
      b += red[1][k][threadIdx.x];
    }
    d.dbeta[c] = accumulate ? d.dbeta[c] + a : a;
    d.dgamma[c] = accumulate ? d.dgamma[c] + b : b;
  }
}

// One block of a BatchNorm's running-statistics update (bn.hip bn_running_multi_kernel): the reference updates them once per
// window, in window order (SURVEY.md finding 5); closed form of the W updates
//   r_W = (1-mom)^W r_0 + mom * sum_w (1-mom)^(W-1-w) stat_w      (unbiased variance n/(n-1))
// channels [32 chunk, 32 chunk + 32) x 8 window slots, folded through LDS in a fixed order (deterministic).
struct BnRunningDesc {
  const float* mean;
  const float* invstd;
  float* rmean;
  float* rvar;
  long long* nbt;
  int W, C, Wn;
  float eps, momentum;
};
__device__ __forceinline__ void bn_running_block(const BnRunningDesc& d, int chunk, float (*red)[8][32]) {
  if (chunk * 32 >= d.C) return;                       // (block-uniform)
  const int c = chunk * 32 + (threadIdx.x & 31), slot = threadIdx.x >> 5;
  const int W = d.W, C = d.C;
  const float keep = 1.f - d.momentum;
  const float unb = d.Wn > 1 ? (float)d.Wn / (float)(d.Wn - 1) : 1.f;
  float am = 0.f, av = 0.f;
  if (c < C) {
    for (int w = slot; w < W; w += 8) {
      float wt = d.momentum * powf(keep, (float)(W - 1 - w));
      float is = d.invstd[(size_t)w * C + c];
      am = fmaf(wt, d.mean[(size_t)w * C + c], am);
      av = fmaf(wt, (1.0f / (is * is) - d.eps) * unb, av);
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
    d.rmean[c] = fmaf(decay, d.rmean[c], sm);
    d.rvar[c] = fmaf(decay, d.rvar[c], sv);
  }
  if (d.nbt && chunk == 0 && threadIdx.x == 0) d.nbt[0] += W;
}

// Winograd F(4,3) taps G g of one (output, input) channel pair, written at u[j * stride] (conv_wino.hip)
__device__ __forceinline__ void wino4_taps(float g0, float g1, float g2, float* u, size_t stride) {
  const float s = g0 + g2;
  u[0] = g0 * 0.25f;
  u[stride] = -(s + g1) * (1.0f / 6.0f);
  u[2 * stride] = -(s - g1) * (1.0f / 6.0f);
  const float t = fmaf(g0, 1.0f / 24.0f, g2 * (1.0f / 6.0f));
  u[3 * stride] = fmaf(g1, 1.0f / 12.0f, t);
  u[4 * stride] = fmaf(g1, -1.0f / 12.0f, t);
  u[5 * stride] = g2;
}

// One weight-gradient GEMM of da_conv_wgrad_multi (include/deepards_hip.h); shared by conv_gemm.hip and conv_wino.hip.
typedef struct {
  const float* dy;
  const float* x;
  float* workspace;      // splits * ntaps*N*C floats: receives the split-K slabs
  int rows, Lm, Ldy, lddy, N, Lx, ldx, C, dy_stride, dy_off, src_stride, ntaps;
  int src_off[3];
  int winograd;          // k3 s1 p1, N and C multiples of 64 -- 1: Winograd F(2,3) form (fp32), 6: F(4,3) form (fp32), 16: bf16 operands /
                         // fp32 sums, 49: split-bf16 fp32-equivalent products on x3 operands (both conv_bf16.hip; also
                         // the stride-2 jobs); the matching da_conv_wgrad_plan(winograd = 1 / 16 / 49) sizes the workspace
  // dense-block operand forms of stride-1 jobs on the direct kernels (winograd == 0; conv_gemm.hip WgradArgs): xform = 1: X is
  // relu(BatchNorm(x)) recomputed while staged from the statistics tables [rows * Lm / Wn][ldstat]; dy_half = 1: dY has
  // Ldy = Lm / 2 positions per row and position j reads dy[j / 2] / 2 (a transition's pooling in front of its conv)
  int xform, dy_half, Wn, ldstat;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
} da_wgrad_job;

// dW[co][ci][k] (torch layout) (+)= sum over the split-K slabs slab[split][k][co][ci] of one convolution
struct WgradReduceDesc {
  const float* slab;
  float* dw;
  int splits, ntaps, N, C;
};

// One block of that reduction = 1 024 consecutive slab elements, a float4 per thread: every thread walks ALL splits of its four
// elements with eight 16-byte loads in flight and one fixed summation tree (deterministic; no LDS, no barrier).
__device__ __forceinline__ void wgrad_reduce_block(const WgradReduceDesc& d, int blk, int accumulate) {
  const int total = d.ntaps * d.N * d.C;          // a multiple of 4 (the host checks); of 1024 for N, C multiples of 32
  const int i = blk * 1024 + threadIdx.x * 4;
  if (i >= total) return;
  const float* p = d.slab + i;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int sp = 0;
  for (; sp + 8 <= d.splits; sp += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + j) * total);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += ((v[0][e] + v[1][e]) + (v[2][e] + v[3][e])) + ((v[4][e] + v[5][e]) + (v[6][e] + v[7][e]));
  }
  if (sp + 4 <= d.splits) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + j) * total);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += (v[0][e] + v[1][e]) + (v[2][e] + v[3][e]);
    sp += 4;
  }
  for (; sp < d.splits; ++sp) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)sp * total);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += v[e];
  }
  // slab order [tap][co][ci] -> torch order [co][ci][tap] (the four elements share a tap: N C is a multiple of 4)
  const int nc = d.N * d.C;
  const int tap = i / nc, rem = i - tap * nc;
  float* o = d.dw + (size_t)rem * d.ntaps + tap;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float* q = o + (size_t)e * d.ntaps;
    *q = accumulate ? *q + s[e] : s[e];
  }
}

// Slab reductions that ride on the next weight-gradient launch (da_conv_wgrad_multi_reduce): the slabs the previous launch
// of the same call wrote are folded by `nblocks` further blocks of this one -- memory-bound blocks beside matrix-bound ones
// (the step's one reduction launch behind all weight gradients found 165 MB of slabs cold: 54 us at B = 64).
#define WGRAD_PRE_MAX 14
struct WgradPreTable {
  WgradReduceDesc d[WGRAD_PRE_MAX];
  int first_block[WGRAD_PRE_MAX + 1];
  int n, nblocks, accumulate;      // nblocks = first_block[n] rounded up to a multiple of 8 (block id % 8 stays the XCD)
  int own;                         // the launch's own blocks, rounded up to a multiple of 8: the reductions come BEHIND them;
                                   // -1: in FRONT of them
};
// Where the reduction blocks sit in the launch, measured on the resnet18 step (B = 64; the F(2,3) launch carrying 75 MB of
// F(4,3) slabs, 267 us alone): in FRONT of the own blocks +15 us (they run alone before any matrix block starts); spread
// evenly in groups of 8 +53 us (the own blocks' whole-round balance is gone); BEHIND them: the own blocks end in a partly
// filled round (1 232 blocks on 1 024 slots) and the reductions fill its idle slots.
// -> -1: this block ran a reduction (or is padding); else its index among the launch's own blocks (may lie past their count:
// padding -- the caller returns)
__device__ __forceinline__ int wgrad_pre_dispatch(const WgradPreTable& p) {
  const int bid = blockIdx.x;
  if (p.nblocks == 0) return bid;
  if (p.own < 0) {                 // in front
    if (bid >= p.nblocks) return bid - p.nblocks;
  } else if (bid < p.own) {
    return bid;
  }
  const int pb = p.own < 0 ? bid : bid - p.own;
  if (pb < p.first_block[p.n]) {
    int i = 0;
    while (i + 1 < p.n && pb >= p.first_block[i + 1]) ++i;      // block-uniform
    wgrad_reduce_block(p.d[i], pb - p.first_block[i], p.accumulate);
  }
  return -1;
}
// grid of a launch of `blocks` own blocks that carries `pre` (sets pre.own); front: the reductions as the FIRST blocks -- for
// a launch whose own blocks already end in short ones (the direct kernels sort theirs longest first)
static inline int wgrad_pre_grid(WgradPreTable& pre, int blocks, bool front = false) {
  if (pre.nblocks == 0) return blocks;
  pre.own = front ? -1 : (blocks + 7) & ~7;
  return (front ? blocks : pre.own) + pre.nblocks;
}

// host side of the chain: `pre` = what the NEXT launch carries (with the jobs it belongs to), `next` = what the launch being
// assembled offers to the one after it
struct WgradChain {
  WgradPreTable pre, next;
  int pre_job[WGRAD_PRE_MAX], next_job[WGRAD_PRE_MAX];
  float* const* dws;       // per job: the gradient destination, or NULL (slabs only)
  int* reduced;            // per job: set to 1 once a launch carries its reduction
};
static inline void wgrad_chain_init(WgradChain* c, float* const* dws, int accumulate, int* reduced) {
  c->pre.n = c->pre.nblocks = c->pre.own = 0;
  c->pre.first_block[0] = 0;
  c->pre.accumulate = accumulate;
  c->next = c->pre;
  c->dws = dws;
  c->reduced = reduced;
}
// the launch being assembled writes job `job`'s slabs
static inline void wgrad_chain_offer(WgradChain* c, int job, const da_wgrad_job& j, int splits) {
  if (!c || !c->dws || !c->dws[job] || c->next.n >= WGRAD_PRE_MAX) return;
  WgradPreTable& t = c->next;
  t.d[t.n] = {j.workspace, c->dws[job], splits, j.ntaps, j.N, j.C};
  c->next_job[t.n] = job;
  t.first_block[t.n + 1] = t.first_block[t.n] + (j.ntaps * j.N * j.C + 1023) / 1024;
  ++t.n;
  t.nblocks = (t.first_block[t.n] + 7) & ~7;
}
// the table a launch carries (empty without a chain); call once per launch, right before it
static inline WgradPreTable wgrad_chain_take(WgradChain* c) {
  WgradPreTable t;
  t.n = t.nblocks = t.accumulate = t.own = 0;
  t.first_block[0] = 0;
  if (!c) return t;
  t = c->pre;
  for (int i = 0; i < t.n; ++i) c->reduced[c->pre_job[i]] = 1;
  c->pre = c->next;
  for (int i = 0; i < c->next.n; ++i) c->pre_job[i] = c->next_job[i];
  c->next.n = c->next.nblocks = 0;
  return t;
}

// conv_wino.hip
bool wino_wgrad_eligible(const da_wgrad_job& j);
void wino_wgrad_plan(int rows, int L, int* splits, int* pchunk, int f = 1);
int wino_wgrad_launch(const da_wgrad_job* jobs, int n, hipStream_t stream, WgradChain* chain = nullptr, int* factors = nullptr);
void wino4_wgrad_plan(int rows, int L, int* splits, int* qchunk);               // winograd == 6: the F(4,3) form (quads)
int wino4_wgrad_launch(const da_wgrad_job* jobs, int n, hipStream_t stream, WgradChain* chain = nullptr);

// conv_bf16.hip: jobs with winograd == 16 (the same eligibility; bf16 operands, padded-position K)
bool bf16_wgrad_eligible(const da_wgrad_job& j);
void bf16_wgrad_plan(int rows, int L, int* splits, int* pchunk);
int bf16_wgrad_launch(const da_wgrad_job* jobs, int n, int code, hipStream_t stream);   // code 16 (bf16) or 49 (x3 operands)
void bf16_wgrad_set_pchunk(int pchunk);

// One problem of da_conv_gemm_multi: the arguments of da_conv_gemm (include/deepards_hip.h).
typedef struct {
  const float* x;
  const float* w;
  float* y;
  int rows, Lm, Lsrc, ldx, C, Ldst, ldy, N, dst_stride, dst_off, src_stride, ntaps;
  int src_off[3];
  int wtap[3];
  int accumulate;
  const float* x2;       // optional second (source, weights) pair for the taps >= tap_split (NULL: one source)
  const float* w2;
  int tap_split;
} da_conv_job;

// sizeof helpers of descriptor structs that are private to one translation unit (da_abi_sizes, head_optim.hip)
extern "C" int da_sizeof_wgrad_reduce_desc(void);
extern "C" int da_sizeof_bn_running_desc(void);
extern "C" int da_sizeof_bn_pgrad_desc(void);
