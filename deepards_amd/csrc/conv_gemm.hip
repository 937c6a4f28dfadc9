// Conv1d forward / data-gradient / weight-gradient as implicit GEMMs on the fp32 matrix cores.
//
// Replaces the nn.Conv1d calls of reference models/resnet.py:5-8,16-19,126-128 (conv2x2, BasicBlock
// convs, 1x1 stride-2 downsample) and models/densenet.py:25-32,75-76 (1x1 bottleneck, k3 growth conv,
// transition conv), forward and backward.
//
// gfx950 has an exact-f32 MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, one rounding per
// product, same rate as the f32 VALU peak but one operand VGPR per lane and the VALU left free), so
// the contraction keeps the reference's fp32 arithmetic while running on the matrix pipe.
//
// Layout: activations RLC (see common.h).  Packed weights Wp[tap][n][c] (c contiguous): the forward
// conv uses Wp_f[k][co][ci], the data gradient Wp_d[k][ci][co] (repack kernels in elementwise.hip).
//
//   fwd/dgrad:  Y[m][n] (+)= sum_t sum_c X[src(m,t)][c] * Wp[wtap_t][n][c]
//               m -> (row, j) = (m / Lm, m % Lm);  src position = j*src_stride + src_off_t (zero
//               outside [0,Lsrc));  dst position = j*dst_stride + dst_off.
//   wgrad:      dWp[t][n][c] = sum_m dY[m][n] * X[src(m,t)][c], split over position chunks into
//               slabs that a reduce kernel sums deterministically (no float atomics).
//
// Tiling: 256 threads = 4 waves; a wave owns TM x TN tiles of 32x32; one K step = 32 reduction
// channels of one tap staged through LDS (pitch 36 floats: conflict-free ds_read_b128, each lane
// reads 4 consecutive k of its row, lane half h takes k = 8*c8 + 4*h + e, the same permutation for
// both operands, so each MFMA pairs identical k on A and B).
#include "common.h"
#include <algorithm>
#include <utility>
#include <vector>

#ifndef WGRAD_GLOAD_AT
#define WGRAD_GLOAD_AT 8
#endif
#ifndef GLOAD_AT
#define GLOAD_AT 3   // quarter of the K step's MFMAs (0..3; 4 = after them) before which the next step's global
                     // loads are issued.  Measured on MI355X: issuing them first (0) costs 5-8 % -- the wave's
                     // vector-memory instructions queue in front of its MFMAs; 3 leaves a quarter step of cover.
#endif

// XCD-aware block order (MI355X: 8 XCDs, blocks are dealt to them round-robin, each with its own 4 MiB L2):
// give every XCD a CONTIGUOUS chunk of the linear tile order, so that tiles which share an operand
// panel run on the same L2 close together in time.  Bijective for any block count; affects speed only.
__device__ __forceinline__ int xcd_linear_tile(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, s = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
}

struct ConvGemmArgs {
  const float* x;
  const float* w;
  const float* x2;  // taps >= tap_split read source x2 / weights w2 (same shapes and pitches): two convolutions that
  const float* w2;  // add into one output as ONE contraction (generic 64x64 kernel and half tiles only)
  int tap_split;
  float* y;
  int M, Lsrc, ldx, C;
  int Ldst, ldy, N;
  int dst_stride, dst_off, src_stride;
  int ntaps, so0, so1, so2, wt0, wt1, wt2;
  int accumulate;
  FastDiv divLm;  // Lm
};

template <int TM, int TN, int WGM, int WGN>
__device__ __forceinline__ void conv_gemm_body(const ConvGemmArgs& a, const int lin, float* lds) {
  constexpr int BM = TM * WGM * 32, BN = TN * WGN * 32, PITCH = 36;
  static_assert(WGM * WGN == 4, "4 waves");
  float* As = lds;
  float* Bs = lds + BM * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  // n-tile fastest, contiguous chunk per XCD: the N/BN tiles that re-read one A panel run back to back on one L2
  const int ntn = a.N / BN;
  const int m_blk = (lin / ntn) * BM, n_blk = (lin % ntn) * BN;
  const int lr = tid >> 3, lq = tid & 7;
  const int Lm = (int)a.divLm.d;

  constexpr int AP = BM / 32, BP = BN / 32;
  int a_rowoff[AP], a_j[AP];
  bool a_ok[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    int m = m_blk + lr + 32 * p;
    a_ok[p] = m < a.M;
    uint32_t row = fdiv((uint32_t)(a_ok[p] ? m : 0), a.divLm);
    int j = (a_ok[p] ? m : 0) - (int)row * Lm;
    a_rowoff[p] = (int)row * a.Lsrc;
    a_j[p] = j * a.src_stride;
  }

  const int kc = a.C >> 5;
  const int nk = a.ntaps * kc;
  f32x4 ra[AP], rb[BP];

  auto gload = [&](int it) {
    int t = it / kc;
    int c0 = (it - t * kc) << 5;
    int so = t == 0 ? a.so0 : (t == 1 ? a.so1 : a.so2);
    int wt = t == 0 ? a.wt0 : (t == 1 ? a.wt1 : a.wt2);
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      int ls = a_j[p] + so;
      bool ok = a_ok[p] && ls >= 0 && ls < a.Lsrc;
      const float* src = (t < a.tap_split ? a.x : a.x2) + (size_t)(a_rowoff[p] + (ok ? ls : 0)) * a.ldx + c0 + lq * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(src);
      ra[p] = v;
    }
    const float* wtp = (t < a.tap_split ? a.w : a.w2) + (size_t)wt * a.N * a.C;
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      int n = n_blk + lr + 32 * p;
      rb[p] = *reinterpret_cast<const f32x4*>(wtp + (size_t)n * a.C + c0 + lq * 4);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  const int frow = lane & 31, fh = lane >> 5;
  for (int it = 0; it < nk; ++it) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < AP; ++p) *reinterpret_cast<f32x4*>(&As[(lr + 32 * p) * PITCH + lq * 4]) = ra[p];
#pragma unroll
    for (int p = 0; p < BP; ++p) *reinterpret_cast<f32x4*>(&Bs[(lr + 32 * p) * PITCH + lq * 4]) = rb[p];
    __syncthreads();
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      if (c8 == GLOAD_AT) {
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nk) gload(it + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * TM + i) * 32 + frow) * PITCH + c8 * 8 + fh * 4]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * TN + j) * 32 + frow) * PITCH + c8 * 8 + fh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
    if (GLOAD_AT >= 4) {
      __builtin_amdgcn_sched_barrier(0);
      if (it + 1 < nk) gload(it + 1);
    }
  }

  // epilogue: lane holds output channel n (column), 16 positions (rows) per 32x32 tile.
  // Row offsets first, then (accumulate) ALL the old values in flight before the first add: the
  // naive per-row load -> wait -> store chain costs 16 dependent round trips.
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    size_t off[16];
    bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int mrow = (r & 3) + 8 * (r >> 2) + 4 * fh;
      int m = m_blk + (wm * TM + i) * 32 + mrow;
      ok[r] = m < a.M;
      uint32_t row = fdiv((uint32_t)(ok[r] ? m : 0), a.divLm);
      int jj = (ok[r] ? m : 0) - (int)row * Lm;
      off[r] = ((size_t)row * a.Ldst + (size_t)(jj * a.dst_stride + a.dst_off)) * a.ldy;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n_blk + (wn * TN + j) * 32 + frow;
      if (a.accumulate) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = ok[r] ? a.y[off[r] + n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ok[r]) a.y[off[r] + n] = acc[i][j][r] + old[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ok[r]) a.y[off[r] + n] = acc[i][j][r];
      }
    }
  }
}

template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvGemmArgs a) {
  __shared__ float lds[(TM * WGM * 32 + TN * WGN * 32) * 36];
  conv_gemm_body<TM, TN, WGM, WGN>(a, xcd_linear_tile(blockIdx.x, gridDim.x), lds);
}

// tuning knobs (benchmark use): 0 = automatic
static int g_force_conv_tile = 0;
static int g_wgrad_target_blocks = 0;

template <int TM, int TN, int WGM, int WGN>
static int launch_conv_gemm(const ConvGemmArgs& a, hipStream_t s) {
  constexpr int BM = TM * WGM * 32, BN = TN * WGN * 32;
  dim3 grid(((a.M + BM - 1) / BM) * (a.N / BN));
  hipLaunchKernelGGL((conv_gemm_kernel<TM, TN, WGM, WGN>), grid, dim3(256), 0, s, a);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// ---------------------------------------------------------------------------------------------
// 3-tap, stride-1 convolution (forward and data gradient of every k3 s1 conv: the bulk of the FLOPs) with
// the three taps sharing ONE staged A panel: for a channel chunk the 64 output positions need source
// positions m0-1 .. m0+64 of the flattened [rows*L] sequence axis -- 66 rows staged once, read three times
// with a row shift; positions whose neighbour belongs to another sequence (j == 0 for the -1 tap, j == L-1 for
// the +1 tap) are zeroed at fragment-read time.  Per 48 MFMAs a wave issues 9 vector loads instead of 12 and
// sits through one barrier pair instead of three.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void conv3_halo_body(const ConvGemmArgs& a, const int block_id, const int nblocks, float* lds) {
  constexpr int BM = 64, BN = 64, PITCH = 36, AROWS = BM + 2;
  float* As = lds;
  float* Bs = lds + AROWS * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = a.N / BN;
  const int lin = xcd_linear_tile(block_id, nblocks);
  const int m_blk = (lin / ntn) * BM, n_blk = (lin % ntn) * BN;
  const int lr = tid >> 3, lq = tid & 7;
  const int L = (int)a.divLm.d;
  const int kc = a.C >> 5;

  // A loader: rows e = lr + 32*p (p = 0,1) and, for the first 16 threads, the two halo-completing rows 64, 65
  // source position of row e: m_blk - 1 + e on the flattened axis (valid while 0 <= pos < M)
  f32x4 ra[3], rb[6];
  auto gload = [&](int kstep) {
    const int c0 = kstep << 5;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      int e = p < 2 ? lr + 32 * p : 64 + (tid >> 3);
      int pos = m_blk - 1 + e;
      bool ok = pos >= 0 && pos < a.M && (p < 2 || tid < 16);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(a.x + (size_t)pos * a.ldx + c0 + lq * 4);
      ra[p] = v;
    }
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      int t = p >> 1;
      int wt = t == 0 ? a.wt0 : (t == 1 ? a.wt1 : a.wt2);
      int n = n_blk + lr + 32 * (p & 1);
      rb[p] = *reinterpret_cast<const f32x4*>(a.w + ((size_t)wt * a.N + n) * a.C + c0 + lq * 4);
    }
  };

  // per-lane edge flags of its fragment row: which taps fall outside the row's own sequence
  const int frow = lane & 31, fh = lane >> 5;
  const int m_lane = m_blk + wm * 32 + frow;
  const uint32_t rowq = fdiv((uint32_t)(m_lane < a.M ? m_lane : 0), a.divLm);
  const int j_lane = (m_lane < a.M ? m_lane : 0) - (int)rowq * L;
  const bool at_first = j_lane == 0, at_last = j_lane == L - 1;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  gload(0);
  for (int ks = 0; ks < kc; ++ks) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<f32x4*>(&As[(lr + 32 * p) * PITCH + lq * 4]) = ra[p];
    if (tid < 16) *reinterpret_cast<f32x4*>(&As[(64 + (tid >> 3)) * PITCH + lq * 4]) = ra[2];
#pragma unroll
    for (int p = 0; p < 6; ++p)
      *reinterpret_cast<f32x4*>(&Bs[((p >> 1) * BN + lr + 32 * (p & 1)) * PITCH + lq * 4]) = rb[p];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int so = t == 0 ? a.so0 : (t == 1 ? a.so1 : a.so2);
      const bool dead = (so < 0 && at_first) || (so > 0 && at_last);
#pragma unroll
      for (int c8 = 0; c8 < 4; ++c8) {
        if (t == 2 && c8 == 1) {   // next chunk's loads late in the MFMA sequence (see GLOAD_AT)
          __builtin_amdgcn_sched_barrier(0);
          if (ks + 1 < kc) gload(ks + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 af = *reinterpret_cast<const f32x4*>(&As[(wm * 32 + frow + so + 1) * PITCH + c8 * 8 + fh * 4]);
        f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[(t * BN + wn * 32 + frow) * PITCH + c8 * 8 + fh * 4]);
        if (dead) af = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
      }
    }
  }

  // epilogue (dst position == flattened m: stride-1, same length)
  size_t off[16];
  bool ok[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int m = m_blk + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
    ok[r] = m < a.M;
    off[r] = (size_t)(ok[r] ? m : 0) * a.ldy;
  }
  const int n = n_blk + wn * 32 + frow;
  if (a.accumulate) {
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = ok[r] ? a.y[off[r] + n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok[r]) a.y[off[r] + n] = acc[r] + old[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok[r]) a.y[off[r] + n] = acc[r];
  }
}

#define HALO_LDS_FLOATS ((64 + 2) * 36 + 3 * 64 * 36)
__global__ __launch_bounds__(256) void conv3_halo_kernel(ConvGemmArgs a) {
  __shared__ float lds[HALO_LDS_FLOATS];
  conv3_halo_body(a, blockIdx.x, gridDim.x, lds);
}

// ---------------------------------------------------------------------------------------------
// Tail tiles.  A grid of T 64x64 tiles runs as rounds of 256 (one per CU and slot); the R = T % 256 tiles of the
// last, partly filled round cost a whole round (at the bench batch every layer has T = 1120: 4.375 rounds paid as 5).
// Those R tiles are cut into 2R half tiles -- 32 positions x 64 channels -- whose 4 waves also split every
// 64-channel K step in two (wave = (channel half wn, k half ks)); the two partial accumulators meet in LDS.
// A half tile occupies a CU for a quarter of a full tile's MFMA time, so the partial round costs ~0.5 instead of 1.
// They are the first blocks of the same launch (padded to a multiple of 8 so that block id % 8 stays the XCD of the
// full tiles) and share the CUs with the full tiles from the start.
// ---------------------------------------------------------------------------------------------
#define TAIL_LDS_FLOATS ((2 * 32 + 2 * 64) * 36)
__device__ __forceinline__ void conv_tail_body(const ConvGemmArgs& a, const int lin, const int half, float* lds) {
  constexpr int PITCH = 36;
  float* As = lds;                     // [2 k halves][32][PITCH]
  float* Bs = lds + 2 * 32 * PITCH;    // [2 k halves][64][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, ks = wave >> 1;
  const int ntn = a.N >> 6;
  const int m_blk = (lin / ntn) * 64 + 32 * half, n_blk = (lin % ntn) * 64;
  const int lr = tid >> 3, lq = tid & 7;
  const int Lm = (int)a.divLm.d;

  const int m_a = m_blk + lr;
  const bool a_ok = m_a < a.M;
  const uint32_t row_a = fdiv((uint32_t)(a_ok ? m_a : 0), a.divLm);
  const int a_rowoff = (int)row_a * a.Lsrc;
  const int a_j = ((a_ok ? m_a : 0) - (int)row_a * Lm) * a.src_stride;

  const int kc = a.C >> 6;
  const int nk = a.ntaps * kc;
  f32x4 ra[2], rb[4];
  auto gload = [&](int it) {
    const int t = it / kc;
    const int c0 = (it - t * kc) << 6;
    const int so = t == 0 ? a.so0 : (t == 1 ? a.so1 : a.so2);
    const int wt = t == 0 ? a.wt0 : (t == 1 ? a.wt1 : a.wt2);
    const int ls = a_j + so;
    const bool ok = a_ok && ls >= 0 && ls < a.Lsrc;
    const float* src = (t < a.tap_split ? a.x : a.x2) + (size_t)(a_rowoff + (ok ? ls : 0)) * a.ldx + c0 + lq * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(src + 32 * h);
      ra[h] = v;
    }
    const float* wtp = (t < a.tap_split ? a.w : a.w2) + (size_t)wt * a.N * a.C + c0 + lq * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        rb[2 * p + h] = *reinterpret_cast<const f32x4*>(wtp + (size_t)(n_blk + lr + 32 * p) * a.C + 32 * h);
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int frow = lane & 31, fh = lane >> 5;
  gload(0);
  for (int it = 0; it < nk; ++it) {
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(&As[(h * 32 + lr) * PITCH + lq * 4]) = ra[h];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<f32x4*>(&Bs[(h * 64 + lr + 32 * p) * PITCH + lq * 4]) = rb[2 * p + h];
    __syncthreads();
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      if (c8 == 3) {
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nk) gload(it + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4 af = *reinterpret_cast<const f32x4*>(&As[(ks * 32 + frow) * PITCH + c8 * 8 + fh * 4]);
      const f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[(ks * 64 + wn * 32 + frow) * PITCH + c8 * 8 + fh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
    }
  }

  // k half 1 hands its partial sums to k half 0 (fixed order: deterministic)
  __syncthreads();
  float* red = lds;                    // [2 channel halves][16][64 lanes]
  if (ks == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wn * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (ks == 1) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += red[(wn * 16 + r) * 64 + lane];

  size_t off[16];
  bool ok[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m_blk + (r & 3) + 8 * (r >> 2) + 4 * fh;
    ok[r] = m < a.M;
    const uint32_t row = fdiv((uint32_t)(ok[r] ? m : 0), a.divLm);
    const int jj = (ok[r] ? m : 0) - (int)row * Lm;
    off[r] = ((size_t)row * a.Ldst + (size_t)(jj * a.dst_stride + a.dst_off)) * a.ldy;
  }
  const int n = n_blk + wn * 32 + frow;
  if (a.accumulate) {
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = ok[r] ? a.y[off[r] + n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok[r]) a.y[off[r] + n] = acc[r] + old[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok[r]) a.y[off[r] + n] = acc[r];
  }
}

// full 64x64 tiles [0, full) + the half tiles of tiles [full, full + nmini / 2) in one launch
template <bool HALO>
__global__ __launch_bounds__(256) void conv_gemm_tailed_kernel(ConvGemmArgs a, int nmini, int nmini_pad, int full) {
  __shared__ float lds[HALO ? HALO_LDS_FLOATS : TAIL_LDS_FLOATS];
  if ((int)blockIdx.x < nmini_pad) {
    if ((int)blockIdx.x < nmini) conv_tail_body(a, full + ((int)blockIdx.x >> 1), blockIdx.x & 1, lds);
    return;
  }
  if (HALO) conv3_halo_body(a, blockIdx.x - nmini_pad, full, lds);
  else conv_gemm_body<1, 1, 2, 2>(a, xcd_linear_tile(blockIdx.x - nmini_pad, full), lds);
}

static int g_use_halo = 128;   // smallest channel count that takes the shared-panel kernel (0: never)
static int g_use_tail = 1;

// launch T = tiles 64x64 tiles, the partly filled last round of 256 as half tiles when that pays (see above)
template <bool HALO>
static int launch_conv64(const ConvGemmArgs& a, hipStream_t s) {
  const int tiles = ((a.M + 63) / 64) * (a.N / 64);
  const int R = tiles % 256;
  if (g_use_tail && a.C % 64 == 0 && tiles > 256 && R >= 1 && R <= 128) {
    const int full = tiles - R, nmini = 2 * R, nmini_pad = (nmini + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_gemm_tailed_kernel<HALO>), dim3(nmini_pad + full), dim3(256), 0, s, a, nmini, nmini_pad, full);
  } else if (HALO) {
    hipLaunchKernelGGL(conv3_halo_kernel, dim3(tiles), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((conv_gemm_kernel<1, 1, 2, 2>), dim3(tiles), dim3(256), 0, s, a);
  }
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// Tile choice.  Measured on MI355X: every tile shape below runs at about the same per-block MFMA
// efficiency, so the choice is about balance: blocks are work-conserving on a CU, the makespan is
// max-blocks-per-CU / mean-blocks-per-CU.  Pick the largest tile whose grid keeps that ratio high.
static double balance(long blocks) {
  double per = (double)blocks / 256.0;
  double mx = (double)((blocks + 255) / 256);
  return per / mx;
}

// Several independent conv GEMMs (64x64 tiles, generic kernel) in ONE launch: the stride-2 conv and the 1x1
// downsample of a block read the same input, the even / odd position sub-problems of a stride-2 data gradient write
// disjoint outputs -- alone each has only 2-4 tiles per CU.  Tiles of all problems form one sequence; its partly
// filled last round runs as half tiles like a single problem's.
struct ConvGemmTable {
  ConvGemmArgs d[4];
  int first_block[5];     // cumulative count of the problems' full-tile blocks
  int n;
  int tail_first;         // first tile (of the LAST problem) that runs as two half tiles
};

__global__ __launch_bounds__(256) void conv_gemm_multi_kernel(ConvGemmTable t, int nmini, int nmini_pad) {
  __shared__ float lds[TAIL_LDS_FLOATS];
  if ((int)blockIdx.x < nmini_pad) {
    if ((int)blockIdx.x < nmini) conv_tail_body(t.d[t.n - 1], t.tail_first + ((int)blockIdx.x >> 1), blockIdx.x & 1, lds);
    return;
  }
  const int g = blockIdx.x - nmini_pad;
  int i = 0;
  while (i + 1 < t.n && g >= t.first_block[i + 1]) ++i;      // wave-uniform
  // every problem spreads over all XCDs (a contiguous chunk of ITS tiles per XCD): problems differ in work per tile
  conv_gemm_body<1, 1, 2, 2>(t.d[i], xcd_linear_tile(g - t.first_block[i], t.first_block[i + 1] - t.first_block[i]), lds);
}

static int launch_conv_multi(const ConvGemmArgs* a, int n, hipStream_t s) {
  ConvGemmTable t;
  int tiles = 0, last = 0;
  bool tail_ok = g_use_tail != 0;
  for (int i = 0; i < n; ++i) {
    t.d[i] = a[i];
    t.first_block[i] = tiles;
    last = ((a[i].M + 63) / 64) * (a[i].N / 64);
    tiles += last;
    tail_ok = tail_ok && a[i].C % 64 == 0;
  }
  t.n = n;
  if (tiles == 0) return DA_OK;
  const int R = tiles % 256;
  int nmini = 0;
  if (tail_ok && tiles > 256 && R >= 1 && R <= 128 && R < last) nmini = 2 * R;   // the tail comes out of the last problem
  t.first_block[n] = tiles - nmini / 2;
  t.tail_first = last - nmini / 2;
  const int nmini_pad = (nmini + 7) / 8 * 8;
  hipLaunchKernelGGL(conv_gemm_multi_kernel, dim3(nmini_pad + t.first_block[n]), dim3(256), 0, s, t, nmini, nmini_pad);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

static int conv_gemm_dispatch(const ConvGemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return DA_OK;
  if (a.C % 32 || a.N % 32 || a.ldx % 4 || a.ntaps < 1 || a.ntaps > 3) return DA_EINVAL;
  if ((uint64_t)a.M * (uint64_t)a.divLm.d >= 0xffffffffull) return DA_EINVAL;
  // k3 stride-1 (forward / data gradient): shared-panel kernel (+2 % on C >= 128; its 37 KB of LDS cap a CU at 4
  // blocks, which costs more than it gains on the 2-chunk C = 64 layer).  Needs src and dst on the same flattened axis.
  if (g_use_halo && a.C >= g_use_halo && a.ntaps == 3 && a.src_stride == 1 && a.dst_stride == 1 && a.dst_off == 0 && a.N % 64 == 0 &&
      a.Lsrc == (int)a.divLm.d && a.Ldst == (int)a.divLm.d && a.so0 >= -1 && a.so0 <= 1 && a.so1 >= -1 && a.so1 <= 1 &&
      a.so2 >= -1 && a.so2 <= 1 && !g_force_conv_tile) {
    return launch_conv64<true>(a, s);
  }
  // candidates: id, BM, BN, relative per-block efficiency
  struct Cand { int id, bm, bn; double eff; };
  static const Cand cands[] = {{1, 128, 128, 1.00}, {2, 64, 128, 0.99}, {3, 128, 64, 0.97}, {4, 64, 64, 0.93},
                               {5, 128, 32, 0.85}, {6, 32, 128, 0.85}};
  int best = 0;
  double best_score = -1.0;
  for (const Cand& c : cands) {
    if (a.N % c.bn) continue;
    if (g_force_conv_tile && g_force_conv_tile != c.id) continue;
    long blocks = (long)((a.M + c.bm - 1) / c.bm) * (a.N / c.bn);
    double score = c.eff * balance(blocks);
    if (score > best_score) {
      best_score = score;
      best = c.id;
    }
  }
  switch (best) {
    case 1: return launch_conv_gemm<2, 2, 2, 2>(a, s);
    case 2: return launch_conv_gemm<1, 2, 2, 2>(a, s);
    case 3: return launch_conv_gemm<1, 2, 4, 1>(a, s);
    case 4: return launch_conv64<false>(a, s);
    case 5: return launch_conv_gemm<1, 1, 4, 1>(a, s);
    case 6: return launch_conv_gemm<1, 1, 1, 4>(a, s);
    default: return DA_EINVAL;
  }
}

// ---------------------------------------------------------------------------------------------
// Dense-block 1x1 convolution with BatchNorm + ReLU applied WHILE THE OPERAND IS STAGED (reference models/densenet.py:23-26
// norm1 -> relu1 -> conv1 of a _DenseLayer, :72-79 norm -> relu -> conv -> pool of a _Transition).  x is the first C
// channels of the block's buffer (pitch ldx); per-(window, channel) statistics come from the block's pitched table; the
// activation h = max(fmaf(x, sc, sh), 0) (bn_scale_shift) exists only in LDS -- "do not store what two FMAs recompute".
// POOL = 1: the transition's AvgPool1d(2,2) is applied IN FRONT of the (linear) conv: output row m contracts
// (h[2m] + h[2m+1]) / 2 -- half the MFMAs, and the full-resolution conv output is never written.
// 64 x 64 tiles (4 waves of 32 x 32), K step 32; a tile spans at most two windows (host: Wn >= 64), whose scale / shift
// vectors sit in LDS behind the operand panels.
// ---------------------------------------------------------------------------------------------
struct Conv1x1BnArgs {
  const float* x;
  const float* w;     // [N][C]
  float* y;
  int M, ldx, C, ldy, N, W, Wn, ldstat;
  float* mean;        // the block's statistics tables [W][ldstat]: read for the channels below pend_c0, WRITTEN for the rest
  float* invstd;
  const float* gamma;
  const float* beta;
  FastDiv divWn;
  // channels [pend_c0, C) have no table entry yet: their statistics arrive as the records their producer's epilogue wrote
  // (conv_wino.hip conv3_wino_stats_kernel / this kernel's OSTATS form) -- merged here, per block for its own two windows,
  // and published to the tables by the block that holds the window's first row (n tile 0): later consumers read the tables
  const float* pend;  // [tiles][2 slots][{mean, M2}][pend_nc] | counts [tiles][2]
  int pend_c0, pend_nc, pend_Wu, pend_tiles;
  float eps;
  float* out_part;    // OSTATS: records of THIS conv's output (N channels, 64-position tiles, windows of Wn positions)
};

template <int POOL, int OSTATS = 0>
__global__ __launch_bounds__(256) void conv1x1_bn_kernel(Conv1x1BnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];          // As | Bs | sc[2][C] | sh[2][C]
  constexpr int PITCH = 36;
  float* As = lds;
  float* Bs = lds + 64 * PITCH;
  float* scs = lds + 128 * PITCH;
  float* shs = scs + 2 * a.C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = a.N >> 6;
  const int lin = xcd_linear_tile(blockIdx.x, gridDim.x);
  const int m_blk = (lin / ntn) * 64, n_blk = (lin % ntn) * 64;
  const int lr = tid >> 3, lq = tid & 7;
  const int w0 = (int)fdiv((uint32_t)m_blk, a.divWn);

  int a_off[2], a_ws[2];
  bool a_ok[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int m = m_blk + lr + 32 * p;
    a_ok[p] = m < a.M;
    const int mm = a_ok[p] ? m : 0;
    a_ws[p] = ((int)fdiv((uint32_t)mm, a.divWn) - w0) * a.C + lq * 4;
    a_off[p] = (POOL ? 2 * mm : mm) * a.ldx + lq * 4;
  }
  const float* wb = a.w + (size_t)(n_blk + lr) * a.C + lq * 4;

  const int kc = a.C >> 5;
  f32x4 ra[2][POOL + 1], rb[2];
  auto gload = [&](int it) {
    const int c0 = it << 5;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int h = 0; h <= POOL; ++h) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[p]) v = *reinterpret_cast<const f32x4*>(a.x + a_off[p] + h * a.ldx + c0);
        ra[p][h] = v;
      }
#pragma unroll
    for (int p = 0; p < 2; ++p) rb[p] = *reinterpret_cast<const f32x4*>(wb + (size_t)(32 * p) * a.C + c0);
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  gload(0);
  // the scale / shift vectors of this tile's two windows (its first operand loads are in flight meanwhile)
  for (int i = tid; i < 2 * a.C; i += 256) {
    const int ws = i >= a.C ? 1 : 0, c = i - ws * a.C, w = w0 + ws;
    float sc = 0.f, sh = 0.f;
    if (w < a.W) {
      float mu, is;
      if (c >= a.pend_c0) {
        merge_stat_records(a.pend, a.pend_tiles, a.pend_nc, a.pend_Wu, w, c - a.pend_c0, a.eps, mu, is);
        const int mw = w * a.Wn;                          // the window's first output row: its tile publishes
        if (n_blk == 0 && mw >= m_blk && mw < m_blk + 64) {
          a.mean[(size_t)w * a.ldstat + c] = mu;
          a.invstd[(size_t)w * a.ldstat + c] = is;
        }
      } else {
        mu = a.mean[(size_t)w * a.ldstat + c];
        is = a.invstd[(size_t)w * a.ldstat + c];
      }
      bn_scale_shift(mu, is, a.gamma[c], a.beta[c], sc, sh);
    }
    scs[i] = sc;
    shs[i] = sh;
  }
  const int frow = lane & 31, fh = lane >> 5;
  for (int it = 0; it < kc; ++it) {
    __syncthreads();
    const int c0 = it << 5;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(&scs[a_ws[p] + c0]);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(&shs[a_ws[p] + c0]);
      f32x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = fmaxf(fmaf(ra[p][0][e], sc[e], sh[e]), 0.f);
        if (POOL) h[e] = 0.5f * (h[e] + fmaxf(fmaf(ra[p][POOL][e], sc[e], sh[e]), 0.f));
      }
      if (!a_ok[p]) h = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&As[(lr + 32 * p) * PITCH + lq * 4]) = h;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<f32x4*>(&Bs[(lr + 32 * p) * PITCH + lq * 4]) = rb[p];
    __syncthreads();
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      if (c8 == GLOAD_AT) {
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < kc) gload(it + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4 af = *reinterpret_cast<const f32x4*>(&As[(wm * 32 + frow) * PITCH + c8 * 8 + fh * 4]);
      const f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[(wn * 32 + frow) * PITCH + c8 * 8 + fh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
    }
  }
  const int n = n_blk + wn * 32 + frow;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m_blk + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
    if (m < a.M) a.y[(size_t)m * a.ldy + n] = acc[r];
  }
  if (OSTATS) {
    // records of the output (the next block's input channels): count, mean, centred M2 per (window slot, channel) of this
    // 64-row tile, two passes over the accumulators; lanes l and l + 32 hold a channel's two row halves, waves (wm = 0, 1)
    // its two 32-row halves -- fixed fold order
    const int m_edge = (w0 + 1) * a.Wn;                   // rows from here on: the tile's second window
    float* red = lds;                                     // [2 wm][2 slots][64 ch] | cnt [2 wm][2] | mean [2][64] | tot [2]
    float* rcnt = red + 256;
    float* rmean = rcnt + 4;
    float* rtot = rmean + 128;
    float ssum[2] = {0.f, 0.f}, scnt[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m_blk + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (m < a.M) {
        const int sl = m >= m_edge ? 1 : 0;
        ssum[sl] += acc[r];
        scnt[sl] += 1.f;
      }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      ssum[sl] += __shfl_xor(ssum[sl], 32, 64);
      scnt[sl] += __shfl_xor(scnt[sl], 32, 64);
    }
    __syncthreads();                                      // the operand panels are dead in every wave
    if (fh == 0) {
      red[(wm * 2 + 0) * 64 + wn * 32 + frow] = ssum[0];
      red[(wm * 2 + 1) * 64 + wn * 32 + frow] = ssum[1];
    }
    if (lane == 0 && wn == 0) {
      rcnt[wm * 2] = scnt[0];
      rcnt[wm * 2 + 1] = scnt[1];
    }
    __syncthreads();
    if (tid < 128) {
      const int sl = tid >> 6, ch = tid & 63;
      const float cn = rcnt[sl] + rcnt[2 + sl];
      rmean[sl * 64 + ch] = cn > 0.f ? (red[(0 * 2 + sl) * 64 + ch] + red[(1 * 2 + sl) * 64 + ch]) / cn : 0.f;
      if (ch == 0) rtot[sl] = cn;
    }
    __syncthreads();
    float sq[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m_blk + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (m < a.M) {
        const int sl = m >= m_edge ? 1 : 0;
        const float d = acc[r] - rmean[sl * 64 + wn * 32 + frow];
        sq[sl] += d * d;
      }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) sq[sl] += __shfl_xor(sq[sl], 32, 64);
    __syncthreads();
    if (fh == 0) {
      red[(wm * 2 + 0) * 64 + wn * 32 + frow] = sq[0];
      red[(wm * 2 + 1) * 64 + wn * 32 + frow] = sq[1];
    }
    __syncthreads();
    if (tid < 128) {
      const int sl = tid >> 6, ch = tid & 63;
      const int mt = m_blk >> 6, nmt = (a.M + 63) >> 6;
      float* rec = a.out_part + ((size_t)(mt * 2 + sl) * 2) * a.N + n_blk + ch;
      rec[0] = rmean[sl * 64 + ch];
      rec[a.N] = red[(0 * 2 + sl) * 64 + ch] + red[(1 * 2 + sl) * 64 + ch];
      if (ch == 0 && n_blk == 0) a.out_part[(size_t)nmt * 4 * a.N + mt * 2 + sl] = rtot[sl];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* dy;
  const float* x;
  float* slab;  // [splits][ntaps][N][C]
  int M, Ldy, lddy, N;
  int Lx, ldx, C;
  int dy_stride, dy_off, src_stride;
  int ntaps, so0, so1, so2;
  int kchunk;  // positions per split, multiple of 32
  FastDiv divLm;
  // dense-block forms: xform = 1: the X operand is max(fmaf(x, sc, sh), 0) of the stored tensor (the activation the 1x1
  // conv's forward applied while staging, see conv1x1_bn_kernel), statistics per window of Wn positions from the pitched
  // tables; dy_half = 1: dY sits at half resolution (a transition's pooling folded in front of its conv): position j
  // reads dy[j / 2] / 2
  int xform, dy_half, ldstat;
  int kind;              // conv_wgrad_any_kernel: which tile shape's body runs the job
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  FastDiv divWn;
};

// output tile (TM*WGM*32 co) x (TN*WGN*32 ci); K = positions, 32 per step; LDS tiles stored as they
// sit in HBM ([pos][channel]): lane (i, h) reads T[2*kk + h][i] -- 32 consecutive floats per half.
template <int TM, int TN, int WGM, int WGN, int XF = 0>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int block_id, const int nblocks, float* lds) {
  constexpr int BM = TM * WGM * 32, BN = TN * WGN * 32;
  static_assert(WGM * WGN == 4, "4 waves");
  float* Ys = lds;             // [32][BM]
  float* Xs = lds + 32 * BM;   // [32][BN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int ntn = a.C / BN, ntm = a.N / BM;
  const int tiles = ntm * ntn * a.ntaps;
  // tile fastest: the tiles that re-read one position chunk of dY / X are consecutive on one XCD
  const int lin = xcd_linear_tile(block_id, nblocks);
  int bx = lin % tiles;
  const int split = lin / tiles;
  const int t = bx / (ntm * ntn);
  bx -= t * ntm * ntn;
  const int n_blk = (bx / ntn) * BM, c_blk = (bx % ntn) * BN;
  // (values first: a conditional between the MEMBERS is an lvalue -- a computed address into `a`, which keeps a local copy of
  // the arguments in scratch memory, conv_wgrad_any_kernel)
  const int so0_ = a.so0, so1_ = a.so1, so2_ = a.so2;
  const int so = t == 0 ? so0_ : (t == 1 ? so1_ : so2_);
  const int Lm = (int)a.divLm.d;
  const int k_beg = split * a.kchunk;
  const int k_end = min(a.M, k_beg + a.kchunk);

  constexpr int YQ = BM / 4, XQ = BN / 4;          // float4 per tile row
  constexpr int YP = (32 * YQ) / 256, XP = (32 * XQ) / 256;
  static_assert((32 * YQ) % 256 == 0 && (32 * XQ) % 256 == 0, "tile/thread mapping");
  f32x4 ry[YP], rx[XP];

  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      int idx = tid + 256 * p;
      int pos = idx / YQ, q = idx % YQ;
      int m = k0 + pos;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < k_end) {
        uint32_t row = fdiv((uint32_t)m, a.divLm);
        int j = m - (int)row * Lm;
        // XF: a tap that falls outside the sequence must contribute nothing, but the recomputed X there is relu(sh), not the
        // zero padding of the stored activation -- so the dY row is zeroed instead (this block works on ONE tap)
        const bool tap_in = !XF || (j * a.src_stride + so >= 0 && j * a.src_stride + so < a.Lx);
        const int jd = (XF && a.dy_half) ? (j >> 1) : j * a.dy_stride + a.dy_off;
        if (tap_in) v = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)row * a.Ldy + (size_t)jd) * a.lddy + n_blk + q * 4);
        // (the factor 1/2 of dy_half goes onto the accumulators in the epilogue: a multiply HERE makes the compiler wait
        // for every load right behind its issue, and the prefetch of the next K step is gone -- +60 % measured)
      }
      ry[p] = v;
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      int idx = tid + 256 * p;
      int pos = idx / XQ, q = idx % XQ;
      int m = k0 + pos;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < k_end) {
        uint32_t row = fdiv((uint32_t)m, a.divLm);
        int j = m - (int)row * Lm;
        int ls = j * a.src_stride + so;
        if (ls >= 0 && ls < a.Lx)
          v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)row * a.Lx + ls) * a.ldx + c_blk + q * 4);
      }
      rx[p] = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // xform: X is transformed when its FRAGMENT is read -- max(fmaf(x, sc, sh), 0) with the scalars of THIS LANE's channels:
  // two VALU ops under the MFMAs, nothing between the barriers.  A lane keeps two sets: the window `wa` of the current K
  // step and wa + 1 (rows of a step that cross into it); they are loaded from the statistics tables right behind the first
  // operand loads, and when the chunk moves on a window the new "next" set is fetched a whole window ahead of its first use.
  // (A scale / shift table in LDS, or a transform of the staged tile, put a dependent memory round trip in front of the
  // first barrier of blocks that run only 5-10 K steps: +60 % at L = 56.)
  static_assert(256 % XQ == 0, "one channel quad per thread");
  const int frow = lane & 31, fh = lane >> 5;
  float xsc[TN][2], xsh[TN][2];
  int wa = 0;
  const int w_total = XF ? (int)fdiv((uint32_t)(a.M - 1), a.divWn) + 1 : 1;
  auto load_set = [&](int s_, int w) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = c_blk + (wn * TN + j) * 32 + frow;
      float sc_, sh_;
      bn_scale_shift(a.mean[(size_t)w * a.ldstat + c], a.invstd[(size_t)w * a.ldstat + c], a.gamma[c], a.beta[c], sc_, sh_);
      xsc[j][s_] = sc_;
      xsh[j][s_] = sh_;
    }
  };
  if (k_beg < k_end) gload(k_beg);
  if (XF && k_beg < k_end) {
    wa = (int)fdiv((uint32_t)k_beg, a.divWn);
    load_set(0, wa);
    load_set(1, min(wa + 1, w_total - 1));
  }
  for (int k0 = k_beg; k0 < k_end; k0 += 32) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      int idx = tid + 256 * p;
      *reinterpret_cast<f32x4*>(&Ys[(idx / YQ) * BM + (idx % YQ) * 4]) = ry[p];
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      int idx = tid + 256 * p;
      *reinterpret_cast<f32x4*>(&Xs[(idx / XQ) * BN + (idx % XQ) * 4]) = rx[p];
    }
    int e_row = 64;                                       // rows of this step from e_row on belong to window wa + 1
    if (XF) {
      const int ws = (int)fdiv((uint32_t)k0, a.divWn);
      if (ws != wa) {                                     // (block-uniform; once per window)
        if (ws == wa + 1) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            xsc[j][0] = xsc[j][1];
            xsh[j][0] = xsh[j][1];
          }
        } else {
          load_set(0, ws);
        }
        wa = ws;
        load_set(1, min(ws + 1, w_total - 1));
      }
      e_row = (ws + 1) * (int)a.divWn.d - k0;
    }
    const bool mixed = e_row < 32;                        // (block-uniform) this step crosses into the next window
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      if (kk == WGRAD_GLOAD_AT) {
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 32 < k_end) gload(k0 + 32);
        __builtin_amdgcn_sched_barrier(0);
      }
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = Ys[(2 * kk + fh) * BM + (wm * TM + i) * 32 + frow];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Xs[(2 * kk + fh) * BN + (wn * TN + j) * 32 + frow];
      if (XF) {                                           // (every job of an XF launch has xform: host)
        if (mixed) {
          const int hi = 2 * kk + fh >= e_row ? 1 : 0;
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[j] = fmaxf(fmaf(bf[j], xsc[j][hi], xsh[j][hi]), 0.f);
        } else {
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[j] = fmaxf(fmaf(bf[j], xsc[j][0], xsh[j][0]), 0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  float* out = a.slab + ((size_t)split * a.ntaps + t) * a.N * a.C;
  const float oscale = (XF && a.dy_half) ? 0.5f : 1.0f;  // exact: a power of two
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int n = n_blk + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        int c = c_blk + (wn * TN + j) * 32 + frow;
        out[(size_t)n * a.C + c] = acc[i][j][r] * oscale;
      }
}

template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  __shared__ float lds[32 * (TM * WGM * 32 + TN * WGN * 32)];
  wgrad_body<TM, TN, WGM, WGN>(a, blockIdx.x, gridDim.x, lds);
}

// All weight gradients of a step that share a tile shape in ONE launch.  They are independent of each other and of
// the rest of the backward pass, and each alone has only 1-2 blocks per CU at B = 64 -- too few to hide its own
// ramp-up and tail (the same kernel runs ~20 % faster with 8 tiles per CU than with 4, scripts/balance_probe.py).
#define WGRAD_TABLE_MAX 20
struct WgradTable {
  WgradArgs d[WGRAD_TABLE_MAX];
  int first_block[WGRAD_TABLE_MAX + 1];
  int n;
};
static_assert(sizeof(WgradTable) + sizeof(WgradPreTable) <= 4096, "kernel arguments: 4 KB");

// pre: the slab reductions of the previous launch's jobs, as this launch's first blocks (common.h WgradPreTable)
template <int TM, int TN, int WGM, int WGN, int XF = 0>
__global__ __launch_bounds__(256) void conv_wgrad_multi_kernel(WgradTable t, WgradPreTable pre) {
  __shared__ float lds[32 * (TM * WGM * 32 + TN * WGN * 32)];
  const int b = wgrad_pre_dispatch(pre);
  if (b < 0 || b >= t.first_block[t.n]) return;
  int i = 0;
  while (i + 1 < t.n && b >= t.first_block[i + 1]) ++i;      // wave-uniform
  wgrad_body<TM, TN, WGM, WGN, XF>(t.d[i], b - t.first_block[i], t.first_block[i + 1] - t.first_block[i], lds);
}

// dW[co][ci][k] (torch layout) (+)= sum_split slab[split][k][co][ci]
// block = 32 consecutive slab elements x 8 split slots, folded through LDS in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           int splits, int ntaps, int N, int C, int accumulate) {
  __shared__ float red[8][32];
  const int total = ntaps * N * C;
  const int i = blockIdx.x * 32 + (threadIdx.x & 31), slot = threadIdx.x >> 5;
  float s = 0.f;
  if (i < total)
    for (int sp = slot; sp < splits; sp += 8) s += slab[(size_t)sp * total + i];
  red[slot][threadIdx.x & 31] = s;
  __syncthreads();
  if (threadIdx.x < 32 && i < total) {
    s = 0.f;
    for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
    int t = i / (N * C);
    int rem = i - t * N * C;  // co*C + ci
    size_t o = (size_t)rem * ntaps + t;
    dw[o] = accumulate ? dw[o] + s : s;
  }
}

// ... and the jobs of DIFFERENT tile shapes in one launch (kind[i] = which body): a step's direct weight gradients as one
// grid of blocks sorted longest first, instead of one launch per tile shape each with its own ramp and partly filled last
// round (densenet18 at B = 64: five launches of 140 ... 3 180 blocks, 254 us).  The 128 x 128 tile stays a launch of its own
// (152 registers: it would halve the occupancy of all the others).
struct WgradAnyTable {
  WgradArgs d[WGRAD_TABLE_MAX];
  int first_block[WGRAD_TABLE_MAX + 1];
  int n;                                    // d[i].kind: 0: 128 x 64, 1: 64 x 128, 2: 64 x 64, 3: 128 x 32, 4: 32 x 128 (co x ci)
};
static_assert(sizeof(WgradAnyTable) + sizeof(WgradPreTable) <= 4096, "kernel arguments: 4 KB");

template <int XF>
__global__ __launch_bounds__(256, 4) void conv_wgrad_any_kernel(WgradAnyTable t, WgradPreTable pre) {
  __shared__ float lds[32 * 192];
  const int b = wgrad_pre_dispatch(pre);
  if (b < 0 || b >= t.first_block[t.n]) return;
  int i = 0;
  while (i + 1 < t.n && b >= t.first_block[i + 1]) ++i;      // wave-uniform
  const int blk = b - t.first_block[i], nblk = t.first_block[i + 1] - t.first_block[i];
  // a COPY of the job's arguments (with a reference to t.d[i] in five inlined bodies the compiler parked the table in scratch
  // memory and re-read it inside the K loops: 1 689 us for 254)
  const WgradArgs& j = t.d[i];
  WgradArgs a;
  a.dy = j.dy; a.x = j.x; a.slab = j.slab;
  a.M = j.M; a.Ldy = j.Ldy; a.lddy = j.lddy; a.N = j.N; a.Lx = j.Lx; a.ldx = j.ldx; a.C = j.C;
  a.dy_stride = j.dy_stride; a.dy_off = j.dy_off; a.src_stride = j.src_stride;
  a.ntaps = j.ntaps; a.so0 = j.so0; a.so1 = j.so1; a.so2 = j.so2; a.kchunk = j.kchunk; a.divLm = j.divLm;
  a.xform = j.xform; a.dy_half = j.dy_half; a.ldstat = j.ldstat; a.kind = j.kind;
  a.mean = j.mean; a.invstd = j.invstd; a.gamma = j.gamma; a.beta = j.beta; a.divWn = j.divWn;
  switch (a.kind) {                                           // block-uniform
    case 0: wgrad_body<2, 1, 2, 2, XF>(a, blk, nblk, lds); break;
    case 1: wgrad_body<1, 2, 2, 2, XF>(a, blk, nblk, lds); break;
    case 2: wgrad_body<1, 1, 2, 2, XF>(a, blk, nblk, lds); break;
    case 3: wgrad_body<1, 1, 4, 1, XF>(a, blk, nblk, lds); break;
    default: wgrad_body<1, 1, 1, 4, XF>(a, blk, nblk, lds); break;
  }
}

// the same reduction (common.h wgrad_reduce_block) for up to 32 convolutions in one launch
#define REDUCE_MAX_BN 24
// first_block[i]: the launch's first block of conv i (1 024 slab elements a block, ONE dimension: no block without work --
// with blockIdx.y = which conv and the largest conv's width, 24 k of the step's 67 k blocks found nothing to do)
struct WgradReduceTable {
  WgradReduceDesc d[32];
  int first_block[33];
  int n;
};
// the BatchNorm dgamma / dbeta folds, the running-statistics updates and the stem's weight-gradient fold of the step: three
// further runs of `nb` blocks behind the reductions
struct ReducePgradTable {
  BnPgradDesc p[REDUCE_MAX_BN];
  BnRunningDesc r[REDUCE_MAX_BN];
  int nb, nsmall, npg, nrun, chunks;   // nsmall = 2 or 3 runs of nb blocks; chunks = ceil(max C / 32): block x of those runs = (BatchNorm x / chunks, channel chunk x % chunks)
  const float* stem_partial;   // third run: the stem's weight-gradient partials [stem_nblk][stem_n] -> stem_dw (or NULL)
  float* stem_dw;
  int stem_nblk, stem_n;
};

__device__ __forceinline__ int wgrad_reduce_find(const WgradReduceTable& t, int b) {      // block-uniform
  int i = 0;
  while (i + 1 < t.n && b >= t.first_block[i + 1]) ++i;
  return i;
}

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WgradReduceTable t, int accumulate) {
  const int i = wgrad_reduce_find(t, blockIdx.x);
  wgrad_reduce_block(t.d[i], (int)blockIdx.x - t.first_block[i], accumulate);
}

__global__ __launch_bounds__(256) void wgrad_reduce_pgrad_kernel(WgradReduceTable t, ReducePgradTable g, int accumulate) {
  // the small folds FIRST: each is a dependent chain (load -> LDS fold -> store) of a few us that the slab reductions behind
  // them hide; as the launch's last blocks they were its tail
  const int nsmall = g.nsmall;
  if ((int)blockIdx.x < nsmall) {
    __shared__ float red[2][8][32];
    const int x = (int)blockIdx.x % g.nb, run = (int)blockIdx.x / g.nb;
    const int bn = x / g.chunks, chunk = x - bn * g.chunks;
    if (run == 0) {
      if (bn < g.npg) bn_param_grad_block(g.p[bn], chunk, accumulate, red);
    } else if (run == 1) {
      if (bn < g.nrun) bn_running_block(g.r[bn], chunk, red);
    } else if (x * 8 < g.stem_n) {
      stem_wgrad_reduce_block(g.stem_partial, g.stem_nblk, g.stem_n, g.stem_dw, accumulate, x,
                              reinterpret_cast<float (*)[8]>(&red[0][0][0]));
    }
    return;
  }
  const int b = (int)blockIdx.x - nsmall;
  const int i = wgrad_reduce_find(t, b);
  wgrad_reduce_block(t.d[i], b - t.first_block[i], accumulate);
}

template <int TM, int TN, int WGM, int WGN>
static int launch_wgrad(const WgradArgs& a, int splits, hipStream_t s) {
  constexpr int BM = TM * WGM * 32, BN = TN * WGN * 32;
  dim3 grid((a.N / BM) * (a.C / BN) * a.ntaps * splits);
  hipLaunchKernelGGL((conv_wgrad_kernel<TM, TN, WGM, WGN>), grid, dim3(256), 0, s, a);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// wgrad plan: output tile (tn x tc) and number of position splits.  Cost model (MI355X measurements):
// MFMA time at ~120 TF/s scaled by CU balance and a per-tile efficiency, plus the slab round trip
// (written once, read once by the reduce) at ~4 TB/s.  Small outputs (layer1/2) prefer small tiles and
// few splits; large outputs (layer3/4) prefer 128x128 tiles.
struct WgradPlan {
  int tn, tc, splits, kchunk;
};

// batch_target > 0: the job shares its launch with many others (launch_wgrad_any): that many blocks per job instead of a grid
// that fills the chip by itself
static WgradPlan wgrad_plan(int M, int N, int C, int ntaps, int batch_target = 0) {
  static const int opts[] = {128, 64, 32};
  WgradPlan best = {0, 0, 1, ((M + 31) / 32) * 32};
  double best_t = 1e30;
  static const int targets[] = {512, 768, 1024};
  for (int target0 : targets)
  for (int tn : opts) {
    if (N % tn) continue;
    for (int tc : opts) {
      if (C % tc) continue;
      int target = g_wgrad_target_blocks > 0 ? g_wgrad_target_blocks : (batch_target > 0 ? batch_target : target0);
      // kernels exist for these pairs only
      bool ok = (tn == 128 && tc == 128) || (tn == 128 && tc == 64) || (tn == 64 && tc == 128) ||
                (tn == 64 && tc == 64) || (tn == 128 && tc == 32) || (tn == 32 && tc == 128);
      if (!ok) continue;
      int tiles = (N / tn) * (C / tc) * ntaps;
      int sp = target / tiles;
      int maxs = (M + 127) / 128;
      if (sp > maxs) sp = maxs;
      if (sp < 1) sp = 1;
      int kc = ((M + sp - 1) / sp + 31) / 32 * 32;
      sp = (M + kc - 1) / kc;
      double eff = (tn * tc >= 128 * 128) ? 1.0 : (tn * tc >= 64 * 128 ? 0.95 : (tn * tc >= 64 * 64 ? 0.88 : 0.8));
      double flops = 2.0 * M * N * C * ntaps;
      double t = flops / (120e12 * eff * balance((long)tiles * sp)) +
                 2.0 * sp * (double)ntaps * N * C * 4.0 / 4e12 + 3e-6;
      if (t < best_t) {
        best_t = t;
        best = {tn, tc, sp, kc};
      }
    }
  }
  return best;
}

// =============================================================================================
// C ABI
// =============================================================================================

template <int TM, int TN, int WGM, int WGN, int XF = 0>
static int launch_wgrad_group(const da_wgrad_job* jobs, int n, int tn, int tc, hipStream_t s, WgradChain* chain = nullptr, int tgt = 0) {
  constexpr int BM = TM * WGM * 32, BN = TN * WGN * 32;
  WgradTable t;
  int cnt = 0, blocks = 0;
  auto flush = [&]() -> int {
    if (!cnt) return DA_OK;
    t.n = cnt;
    t.first_block[cnt] = blocks;
    WgradPreTable pre = wgrad_chain_take(chain);
    hipLaunchKernelGGL((conv_wgrad_multi_kernel<TM, TN, WGM, WGN, XF>), dim3(wgrad_pre_grid(pre, blocks, true)), dim3(256), 0, s, t, pre);
    DA_CHECK_LAUNCH();
    cnt = 0;
    blocks = 0;
    return DA_OK;
  };
  // longest blocks first (block time ~ kchunk * taps-independent tile work): short ones fill the launch's tail
  std::vector<std::pair<int, int>> order;
  for (int i = 0; i < n; ++i) {
    if (jobs[i].winograd) continue;                       // conv_wino.hip
    if ((XF != 0) != (jobs[i].xform != 0 || jobs[i].dy_half != 0)) continue;      // the operand forms have kernels of their own
    WgradPlan pl = wgrad_plan(jobs[i].rows * jobs[i].Lm, jobs[i].N, jobs[i].C, jobs[i].ntaps, tgt);
    if (pl.tn == tn && pl.tc == tc) order.push_back({-pl.kchunk, i});
  }
  std::stable_sort(order.begin(), order.end());
  for (const auto& o : order) {
    const da_wgrad_job& j = jobs[o.second];
    WgradPlan pl = wgrad_plan(j.rows * j.Lm, j.N, j.C, j.ntaps, tgt);
    WgradArgs& a = t.d[cnt];
    a.dy = j.dy; a.x = j.x; a.slab = j.workspace;
    a.M = j.rows * j.Lm; a.Ldy = j.Ldy; a.lddy = j.lddy; a.N = j.N;
    a.Lx = j.Lx; a.ldx = j.ldx; a.C = j.C;
    a.dy_stride = j.dy_stride; a.dy_off = j.dy_off; a.src_stride = j.src_stride;
    a.ntaps = j.ntaps;
    a.so0 = j.src_off[0]; a.so1 = j.ntaps > 1 ? j.src_off[1] : 0; a.so2 = j.ntaps > 2 ? j.src_off[2] : 0;
    a.kchunk = pl.kchunk;
    a.divLm = make_fastdiv((uint32_t)j.Lm);
    a.xform = j.xform; a.dy_half = j.dy_half; a.ldstat = j.ldstat;
    a.mean = j.mean; a.invstd = j.invstd; a.gamma = j.gamma; a.beta = j.beta;
    a.divWn = make_fastdiv((uint32_t)(j.Wn > 0 ? j.Wn : 1));
    t.first_block[cnt] = blocks;
    blocks += (j.N / BM) * (j.C / BN) * j.ntaps * pl.splits;
    wgrad_chain_offer(chain, o.second, j, pl.splits);
    if (++cnt == WGRAD_TABLE_MAX) {
      int rc = flush();
      if (rc) return rc;
    }
  }
  return flush();
}

static void wgrad_fill_args(WgradArgs& a, const da_wgrad_job& j, const WgradPlan& pl) {
  a.dy = j.dy; a.x = j.x; a.slab = j.workspace;
  a.M = j.rows * j.Lm; a.Ldy = j.Ldy; a.lddy = j.lddy; a.N = j.N;
  a.Lx = j.Lx; a.ldx = j.ldx; a.C = j.C;
  a.dy_stride = j.dy_stride; a.dy_off = j.dy_off; a.src_stride = j.src_stride;
  a.ntaps = j.ntaps;
  a.so0 = j.src_off[0]; a.so1 = j.ntaps > 1 ? j.src_off[1] : 0; a.so2 = j.ntaps > 2 ? j.src_off[2] : 0;
  a.kchunk = pl.kchunk;
  a.divLm = make_fastdiv((uint32_t)j.Lm);
  a.xform = j.xform; a.dy_half = j.dy_half; a.ldstat = j.ldstat;
  a.kind = 0;
  a.mean = j.mean; a.invstd = j.invstd; a.gamma = j.gamma; a.beta = j.beta;
  a.divWn = make_fastdiv((uint32_t)(j.Wn > 0 ? j.Wn : 1));
}

static int g_wgrad_any = 1;     // da_debug_set(key 7): 0 = one launch per tile shape (the form the tests compare with)

// every direct job of the operand kind XF whose plan is not the 128 x 128 tile, in one launch (conv_wgrad_any_kernel)
template <int XF>
static int launch_wgrad_any(const da_wgrad_job* jobs, int n, hipStream_t s, WgradChain* chain, int tgt) {
  WgradAnyTable t;
  int cnt = 0, blocks = 0;
  auto flush = [&]() -> int {
    if (!cnt) return DA_OK;
    t.n = cnt;
    t.first_block[cnt] = blocks;
    WgradPreTable pre = wgrad_chain_take(chain);
    hipLaunchKernelGGL((conv_wgrad_any_kernel<XF>), dim3(wgrad_pre_grid(pre, blocks, true)), dim3(256), 0, s, t, pre);
    DA_CHECK_LAUNCH();
    cnt = 0;
    blocks = 0;
    return DA_OK;
  };
  // longest blocks first (block time ~ positions per split x 32 x 32 products per wave): short ones fill the launch's tail
  std::vector<std::pair<long, int>> order;
  for (int i = 0; i < n; ++i) {
    if (jobs[i].winograd) continue;
    if ((XF != 0) != (jobs[i].xform != 0 || jobs[i].dy_half != 0)) continue;
    const WgradPlan pl = wgrad_plan(jobs[i].rows * jobs[i].Lm, jobs[i].N, jobs[i].C, jobs[i].ntaps, tgt);
    if (pl.tn == 128 && pl.tc == 128) continue;
    order.push_back({-(long)pl.kchunk * (pl.tn * pl.tc / 4096), i});
  }
  std::stable_sort(order.begin(), order.end());
  for (const auto& o : order) {
    const da_wgrad_job& j = jobs[o.second];
    const WgradPlan pl = wgrad_plan(j.rows * j.Lm, j.N, j.C, j.ntaps, tgt);
    wgrad_fill_args(t.d[cnt], j, pl);
    t.d[cnt].kind = pl.tn == 128 ? (pl.tc == 64 ? 0 : 3) : (pl.tn == 64 ? (pl.tc == 128 ? 1 : 2) : 4);
    t.first_block[cnt] = blocks;
    blocks += (j.N / pl.tn) * (j.C / pl.tc) * j.ntaps * pl.splits;
    wgrad_chain_offer(chain, o.second, j, pl.splits);
    if (++cnt == WGRAD_TABLE_MAX) {
      int rc = flush();
      if (rc) return rc;
    }
  }
  return flush();
}

extern "C" {

// Benchmark-only tuning knobs.  key 0: force the conv GEMM tile (0 auto, 1 128x128, 2 64x128, 3 128x64,
// 4 64x64, 5 128x32, 6 32x128).  key 1: wgrad target block count (0 = 1024).  key 7: 0 = the dense-block weight gradients as
// one launch per tile shape instead of one launch for all shapes (conv_wgrad_any_kernel).
int da_debug_set(int key, int value) {
  if (key == 0) g_force_conv_tile = value;
  else if (key == 1) g_wgrad_target_blocks = value;
  else if (key == 2) g_use_halo = value;
  else if (key == 3) g_use_tail = value;
  else if (key == 7) g_wgrad_any = value;
  else return DA_EINVAL;
  return DA_OK;
}

// Forward conv / generic implicit GEMM.  Every pointer is a device pointer; returns 0 on success.
//   rows: B*NB sequences.  x: [rows][Lsrc][ldx] first C channels used.  w: packed [ntaps_w][N][C].
//   y: [rows][Ldst][ldy] first N channels written.  Lm output positions per row are produced at
//   dst position j*dst_stride+dst_off from src positions j*src_stride+src_off[t] with weight tap wtap[t].
int da_conv_gemm(const float* x, const float* w, float* y, int rows, int Lm, int Lsrc, int ldx, int C, int Ldst,
                 int ldy, int N, int dst_stride, int dst_off, int src_stride, int ntaps, const int* src_off,
                 const int* wtap, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (!x || !w || !y || rows < 0 || Lm < 1 || ntaps < 1 || ntaps > 3) return DA_EINVAL;
  ConvGemmArgs a;
  a.x = x; a.w = w; a.y = y;
  a.x2 = x; a.w2 = w; a.tap_split = 3;
  a.M = rows * Lm; a.Lsrc = Lsrc; a.ldx = ldx; a.C = C;
  a.Ldst = Ldst; a.ldy = ldy; a.N = N;
  a.dst_stride = dst_stride; a.dst_off = dst_off; a.src_stride = src_stride;
  a.ntaps = ntaps;
  a.so0 = src_off[0]; a.so1 = ntaps > 1 ? src_off[1] : 0; a.so2 = ntaps > 2 ? src_off[2] : 0;
  a.wt0 = wtap[0]; a.wt1 = ntaps > 1 ? wtap[1] : 0; a.wt2 = ntaps > 2 ? wtap[2] : 0;
  a.accumulate = accumulate;
  a.divLm = make_fastdiv((uint32_t)Lm);
  return conv_gemm_dispatch(a, stream);
}

// n <= 4 independent problems of da_conv_gemm (jobs: HOST array) in one launch of the 64x64-tile kernel (N % 64 == 0).
// The problems must not write the same output elements.
int da_conv_gemm_multi(const da_conv_job* jobs, int n, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (n < 1 || n > 4 || !jobs) return DA_EINVAL;
  ConvGemmArgs a[4];
  for (int i = 0; i < n; ++i) {
    const da_conv_job& j = jobs[i];
    if (!j.x || !j.w || !j.y || j.rows < 0 || j.Lm < 1 || j.ntaps < 1 || j.ntaps > 3 || j.C % 32 || j.N % 64 || j.ldx % 4)
      return DA_EINVAL;
    if ((uint64_t)j.rows * j.Lm * (uint64_t)j.Lm >= 0xffffffffull) return DA_EINVAL;
    ConvGemmArgs& g = a[i];
    g.x = j.x; g.w = j.w; g.y = j.y;
    g.x2 = j.x2 ? j.x2 : j.x; g.w2 = j.w2 ? j.w2 : j.w; g.tap_split = j.x2 ? j.tap_split : 3;
    if (j.x2 && (!j.w2 || j.tap_split < 1 || j.tap_split >= j.ntaps)) return DA_EINVAL;
    g.M = j.rows * j.Lm; g.Lsrc = j.Lsrc; g.ldx = j.ldx; g.C = j.C;
    g.Ldst = j.Ldst; g.ldy = j.ldy; g.N = j.N;
    g.dst_stride = j.dst_stride; g.dst_off = j.dst_off; g.src_stride = j.src_stride;
    g.ntaps = j.ntaps;
    g.so0 = j.src_off[0]; g.so1 = j.ntaps > 1 ? j.src_off[1] : 0; g.so2 = j.ntaps > 2 ? j.src_off[2] : 0;
    g.wt0 = j.wtap[0]; g.wt1 = j.ntaps > 1 ? j.wtap[1] : 0; g.wt2 = j.ntaps > 2 ? j.wtap[2] : 0;
    g.accumulate = j.accumulate;
    g.divLm = make_fastdiv((uint32_t)j.Lm);
  }
  return launch_conv_multi(a, n, stream);
}

// y[m][0:N] (pitch ldy) = sum_c W[n][c] * h(m, c), h = relu(BatchNorm(x)) applied while x is staged (conv1x1_bn_kernel):
// x [rows * Lin][ldx] first C channels, statistics per window of R rows from the pitched tables [rows / R][ldstat];
// pool != 0: h(m, .) = (h[2m] + h[2m+1]) / 2, Lin even, the output has Lin / 2 positions per row (a _Transition with its
// AvgPool1d(2,2) in front of the conv).  w: [N][C] (the torch weight of a k = 1 conv as it lies).  N % 64 == 0, C % 32 == 0,
// R * Lout >= 64.  replaces norm1 -> relu1 -> conv1 (densenet.py:23-26) and norm -> relu -> conv -> pool (:72-79)
// pend != NULL: the channels [pend_c0, C) take their statistics from the records `pend` their producer wrote (pend_units units
// in all, pend_Wu per window: 64-unit tiles, da_stat_records_floats) and this call PUBLISHES them to the tables;
// out_part != NULL: this call writes the records of its own output (windows of R * Lout positions, N channels).
int da_conv1x1_bn(const float* x, int ldx, const float* w, float* y, int ldy, int rows, int R, int Lin, int C, int N, int pool,
                  float* mean, float* invstd, int ldstat, const float* gamma, const float* beta, const float* pend, int pend_c0,
                  long pend_units, int pend_Wu, float eps, float* out_part, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;
  if (!x || !w || !y || !mean || !invstd || !gamma || !beta || rows < 0 || R < 1 || rows % R || Lin < 1) return DA_EINVAL;
  if (C % 32 || C < 32 || C > 2048 || N % 64 || N < 64 || ldx % 4 || ldx < C || ldy < N || ldstat % 4 || ldstat < C) return DA_EINVAL;
  if (pool && (Lin & 1)) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const int Lout = pool ? Lin / 2 : Lin;
  const long M = (long)rows * Lout;
  if ((long)R * Lout < 64) return DA_EINVAL;                     // a tile must not span more than two windows
  if ((uint64_t)rows * Lin * (uint64_t)(ldx > ldy ? ldx : ldy) >= 0x7fffffffull) return DA_EINVAL;   // 32-bit element offsets
  if ((uint64_t)M * (uint64_t)(R * Lout) >= 0xffffffffull) return DA_EINVAL;
  Conv1x1BnArgs a;
  a.x = x; a.w = w; a.y = y;
  a.M = (int)M; a.ldx = ldx; a.C = C; a.ldy = ldy; a.N = N; a.W = rows / R; a.Wn = R * Lout; a.ldstat = ldstat;
  a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta;
  a.divWn = make_fastdiv((uint32_t)a.Wn);
  a.pend = pend; a.pend_c0 = C; a.pend_nc = 0; a.pend_Wu = 1; a.pend_tiles = 0; a.eps = eps; a.out_part = out_part;
  if (pend) {
    if (pend_c0 < 0 || pend_c0 >= C || pend_units < 1 || pend_Wu < 64 || pend_units % pend_Wu || pend_units / pend_Wu != rows / R)
      return DA_EINVAL;
    a.pend_c0 = pend_c0; a.pend_nc = C - pend_c0; a.pend_Wu = pend_Wu; a.pend_tiles = (int)((pend_units + 63) / 64);
  }
  const unsigned blocks = (unsigned)(((M + 63) / 64) * (N / 64));
  const size_t shm = (size_t)(128 * 36 + 4 * C) * sizeof(float);
  if (out_part) {
    if (pool) hipLaunchKernelGGL((conv1x1_bn_kernel<1, 1>), dim3(blocks), dim3(256), shm, stream, a);
    else hipLaunchKernelGGL((conv1x1_bn_kernel<0, 1>), dim3(blocks), dim3(256), shm, stream, a);
  } else if (pool) hipLaunchKernelGGL((conv1x1_bn_kernel<1, 0>), dim3(blocks), dim3(256), shm, stream, a);
  else hipLaunchKernelGGL((conv1x1_bn_kernel<0, 0>), dim3(blocks), dim3(256), shm, stream, a);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// Bytes of slab workspace da_conv_wgrad needs for this shape.
size_t da_conv_wgrad_workspace(int rows, int Lm, int N, int C, int ntaps) {
  WgradPlan p = wgrad_plan(rows * Lm, N, C, ntaps);
  return (size_t)p.splits * ntaps * N * C * sizeof(float);
}

// Slabs of n weight gradients (jobs: HOST array) with one launch per tile shape; reduce them afterwards with
// da_wgrad_reduce_multi (da_conv_wgrad_splits() slabs per job).
// dws != NULL: dws[i] is job i's gradient destination (or NULL); the slab reduction dws[i] (+)= sum of job i's slabs then rides
// in front of the NEXT launch of this call (common.h WgradPreTable) where there is one, and reduced[i] says whether it did
// (1) or the caller still owes it (0: da_wgrad_reduce_multi / da_step_tail_multi) -- the jobs of the call's last launch always.
// splits_out != NULL: the number of slabs every job wrote (what its reduction must be told) -- with it the dense-block jobs
// of a call that has 8 or more of them are planned as a batch (one launch, conv_wgrad_any_kernel: ~2 560 blocks over all of
// them instead of 512 ... 1 024 per job; never more slabs than da_conv_wgrad_plan's figure, which sizes the workspace).
// densenet18 at B = 64, per-job target 0 (the plan alone) / 64 / 128 / 192 / 256 / 384 / 512: 1.215 / 1.249 / 1.187 / 1.194 /
// 1.199 / 1.206 / 1.213 ms a step.
static int conv_wgrad_multi_impl(const da_wgrad_job* jobs, int n, float* const* dws, int accumulate, int* reduced,
                                 int* splits_out, hipStream_t stream) {
  DA_ENTER();
  if (n < 0 || (n && !jobs)) return DA_EINVAL;
  for (int i = 0; i < n; ++i) {
    const da_wgrad_job& j = jobs[i];
    if (!j.dy || !j.x || !j.workspace || j.ntaps < 1 || j.ntaps > 3 || j.C % 32 || j.N % 32 || j.lddy % 4 || j.ldx % 4 ||
        (j.winograd == 49 && (j.lddy != j.N || j.ldx != j.C)))
      return DA_EINVAL;
    if ((uint64_t)j.rows * j.Lm * (uint64_t)j.Lm >= 0xffffffffull) return DA_EINVAL;
    if (j.winograd != 0 && j.winograd != 1 && j.winograd != 6 && j.winograd != 16 && j.winograd != 49) return DA_EINVAL;
    if ((j.xform || j.dy_half) && (j.winograd || j.src_stride != 1 || j.dy_stride != 1 || j.dy_off || j.Lm != j.Lx ||
                                   (j.dy_half && j.ntaps != 1)))
      return DA_EINVAL;                                       // the dense-block operand forms: stride-1 jobs on the direct kernels
    if (j.dy_half && !j.xform) return DA_EINVAL;              // (the half-resolution dY only comes with the recomputed X)
    if (j.xform && (j.xform != 1 || !j.mean || !j.invstd || !j.gamma || !j.beta || j.Wn < 1 || j.ldstat < j.C || j.ldstat % 4 ||
                    j.Wn < 32 || (uint64_t)j.rows * j.Lm * (uint64_t)j.Wn >= 0xffffffffull))
      return DA_EINVAL;
    if (j.dy_half && (j.Lm & 1 || j.Ldy * 2 != j.Lm)) return DA_EINVAL;
    if (g_act_bf16 && j.winograd != 16) return DA_EINVAL;     // bf16 activations: only the bf16-operand kernels read them
    if (j.winograd == 16 || j.winograd == 49
            ? !bf16_wgrad_eligible(j)
            : (j.winograd ? !wino_wgrad_eligible(j) : !wgrad_plan(j.rows * j.Lm, j.N, j.C, j.ntaps).tn))
      return DA_EINVAL;
  }
  int rc;
  WgradChain chain_, *chain = nullptr;
  if (dws) {
    if (!reduced) return DA_EINVAL;
    for (int i = 0; i < n; ++i) reduced[i] = 0;
    wgrad_chain_init(&chain_, dws, accumulate, reduced);
    chain = &chain_;
  }
  if ((rc = wino4_wgrad_launch(jobs, n, stream, chain))) return rc;  // the heaviest blocks first
  std::vector<int> wino_f(n > 0 ? n : 1, 1);
  if ((rc = wino_wgrad_launch(jobs, n, stream, chain, splits_out ? wino_f.data() : nullptr))) return rc;
  if ((rc = bf16_wgrad_launch(jobs, n, 49, stream))) return rc;      // x3 operands (dy / x are x3 tensors; ld* = channel counts)
  if ((rc = bf16_wgrad_launch(jobs, n, 16, stream))) return rc;
  if ((rc = launch_wgrad_group<2, 2, 2, 2>(jobs, n, 128, 128, stream, chain))) return rc;
  if ((rc = launch_wgrad_group<2, 1, 2, 2>(jobs, n, 128, 64, stream, chain))) return rc;
  if ((rc = launch_wgrad_group<1, 2, 2, 2>(jobs, n, 64, 128, stream, chain))) return rc;
  if ((rc = launch_wgrad_group<1, 1, 2, 2>(jobs, n, 64, 64, stream, chain))) return rc;
  if ((rc = launch_wgrad_group<1, 1, 4, 1>(jobs, n, 128, 32, stream, chain))) return rc;
  if ((rc = launch_wgrad_group<1, 1, 1, 4>(jobs, n, 32, 128, stream, chain))) return rc;
  int n_xf = 0;
  for (int i = 0; i < n; ++i) n_xf += (jobs[i].xform || jobs[i].dy_half) ? 1 : 0;
  const int tgt = (splits_out && g_wgrad_any && n_xf >= 8) ? (2560 / n_xf < 96 ? 96 : 2560 / n_xf) : 0;
  if (splits_out)
    for (int i = 0; i < n; ++i) {
      const da_wgrad_job& j = jobs[i];
      int sp = 0, kc = 0;
      if (j.winograd == 16 || j.winograd == 49) bf16_wgrad_plan(j.rows, j.Lm, &sp, &kc);
      else if (j.winograd == 6) wino4_wgrad_plan(j.rows, j.Lm, &sp, &kc);
      else if (j.winograd) wino_wgrad_plan(j.rows, j.Lm, &sp, &kc, wino_f[i]);
      else sp = wgrad_plan(j.rows * j.Lm, j.N, j.C, j.ntaps, (j.xform || j.dy_half) ? tgt : 0).splits;
      splits_out[i] = sp;
    }
  if (n_xf) {                                             // dense-block operand forms (conv1x1_bn_kernel's weight gradients)
    if ((rc = launch_wgrad_group<2, 2, 2, 2, 1>(jobs, n, 128, 128, stream, chain, tgt))) return rc;
    if (g_wgrad_any) return launch_wgrad_any<1>(jobs, n, stream, chain, tgt);
    if ((rc = launch_wgrad_group<2, 1, 2, 2, 1>(jobs, n, 128, 64, stream, chain))) return rc;
    if ((rc = launch_wgrad_group<1, 2, 2, 2, 1>(jobs, n, 64, 128, stream, chain))) return rc;
    if ((rc = launch_wgrad_group<1, 1, 2, 2, 1>(jobs, n, 64, 64, stream, chain))) return rc;
    if ((rc = launch_wgrad_group<1, 1, 4, 1, 1>(jobs, n, 128, 32, stream, chain))) return rc;
    if ((rc = launch_wgrad_group<1, 1, 1, 4, 1>(jobs, n, 32, 128, stream, chain))) return rc;
  }
  return DA_OK;
}

int da_conv_wgrad_multi(const da_wgrad_job* jobs, int n, hipStream_t stream) {
  return conv_wgrad_multi_impl(jobs, n, nullptr, 0, nullptr, nullptr, stream);
}

int da_conv_wgrad_multi_reduce(const da_wgrad_job* jobs, int n, float* const* dws, int accumulate, int* reduced,
                               int* splits, hipStream_t stream) {
  if (n && (!dws || !reduced || !splits)) return DA_EINVAL;
  return conv_wgrad_multi_impl(jobs, n, dws, accumulate, reduced, splits, stream);
}

// number of slabs da_conv_wgrad writes for this shape (workspace = splits * ntaps*N*C floats)
int da_conv_wgrad_splits(int rows, int Lm, int N, int C, int ntaps) {
  return wgrad_plan(rows * Lm, N, C, ntaps).splits;
}

// the plan da_conv_wgrad / da_conv_wgrad_multi use for this shape: out = {tile_n, tile_c, splits, kchunk};
// a job's workspace is splits * ntaps*N*C floats.  winograd != 0: the plan of a job with that flag set (k3 s1 p1;
// kchunk counts output PAIRS).
int da_conv_wgrad_plan(int rows, int Lm, int N, int C, int ntaps, int winograd, int* out) {
  if (!out || ntaps < 1 || ntaps > 3 || N % 32 || C % 32) return DA_EINVAL;
  if (winograd) {
    if ((ntaps != 3 && winograd != 16 && winograd != 49) || N % 64 || C % 64) return DA_EINVAL;
    out[0] = 64; out[1] = 64;
    if (winograd == 16 || winograd == 49) bf16_wgrad_plan(rows, Lm, &out[2], &out[3]);     // kchunk counts padded positions
    else if (winograd == 6) wino4_wgrad_plan(rows, Lm, &out[2], &out[3]);                  // ... quads
    else wino_wgrad_plan(rows, Lm, &out[2], &out[3]);
    return DA_OK;
  }
  WgradPlan p = wgrad_plan(rows * Lm, N, C, ntaps);
  out[0] = p.tn; out[1] = p.tc; out[2] = p.splits; out[3] = p.kchunk;
  return p.tn ? DA_OK : DA_EINVAL;
}

typedef struct {
  const float* slab;
  float* dw;
  int splits, ntaps, N, C;
} da_wgrad_reduce_desc;

int da_sizeof_wgrad_reduce_desc(void) { return (int)sizeof(da_wgrad_reduce_desc); }

typedef struct {          // (bn.hip's definitions; include/deepards_hip.h)
  const float* s1;
  const float* s2;
  float* dgamma;
  float* dbeta;
  int W, C;
} da_bn_pgrad_desc;
typedef struct {
  const float* mean;
  const float* invstd;
  float* running_mean;
  float* running_var;
  long long* num_batches_tracked;
  int W, C, Wn;
  float eps, momentum;
} da_bn_running_desc;
int da_bn_param_grad_multi(const da_bn_pgrad_desc* descs, int n, int accumulate, hipStream_t stream);
int da_bn_running_multi(const da_bn_running_desc* descs, int n, hipStream_t stream);
int da_wgrad_reduce_multi(const da_wgrad_reduce_desc* descs, int n, int accumulate, hipStream_t stream);

// The tail of a training step in ONE launch: every slab reduction (da_wgrad_reduce_multi), every BatchNorm dgamma / dbeta
// fold (da_bn_param_grad_multi) and every running-statistics update (da_bn_running_multi).  Up to 32 reductions and 24
// BatchNorms of each kind share the launch; anything else runs as the three calls.
int da_stem_wgrad_reduce(const float* partial, int nblk, int n, float* dw, int accumulate, hipStream_t stream);
int da_step_tail_multi(const da_wgrad_reduce_desc* descs, int n, const da_bn_pgrad_desc* pg, int npg,
                       const da_bn_running_desc* run, int nrun, const float* stem_partial, int stem_nblk, int stem_n,
                       float* stem_dw, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (n < 0 || npg < 0 || nrun < 0 || (n && !descs) || (npg && !pg) || (nrun && !run)) return DA_EINVAL;
  if (stem_partial && (!stem_dw || stem_nblk < 1 || stem_n < 1)) return DA_EINVAL;
  if (n == 0 || n > 32 || npg > REDUCE_MAX_BN || nrun > REDUCE_MAX_BN || (npg + nrun == 0 && !stem_partial)) {
    int rc = nrun ? da_bn_running_multi(run, nrun, stream) : DA_OK;
    if (!rc && stem_partial) rc = da_stem_wgrad_reduce(stem_partial, stem_nblk, stem_n, stem_dw, accumulate, stream);
    if (!rc && npg) rc = da_bn_param_grad_multi(pg, npg, accumulate, stream);
    if (!rc && n) rc = da_wgrad_reduce_multi(descs, n, accumulate, stream);
    return rc;
  }
  WgradReduceTable t;
  ReducePgradTable g;
  int maxc = 0, nred = 0;
  for (int i = 0; i < n; ++i) {
    const da_wgrad_reduce_desc& s = descs[i];
    if (!s.slab || !s.dw || s.splits < 1) return DA_EINVAL;
    if (s.ntaps < 1 || s.N < 1 || s.C < 1 || (s.N * s.C) % 4) return DA_EINVAL;
    t.d[i] = {s.slab, s.dw, s.splits, s.ntaps, s.N, s.C};
    t.first_block[i] = nred;
    nred += (s.ntaps * s.N * s.C + 1023) / 1024;
  }
  t.first_block[n] = nred;
  t.n = n;
  for (int i = 0; i < npg; ++i) {
    const da_bn_pgrad_desc& s = pg[i];
    if (!s.s1 || !s.s2 || !s.dgamma || !s.dbeta) return DA_EINVAL;
    g.p[i] = {s.s1, s.s2, s.dgamma, s.dbeta, s.W, s.C};
    if (s.C > maxc) maxc = s.C;
  }
  for (int i = 0; i < nrun; ++i) {
    const da_bn_running_desc& s = run[i];
    if (!s.mean || !s.invstd || !s.running_mean || !s.running_var || s.Wn < 1) return DA_EINVAL;
    g.r[i] = {s.mean, s.invstd, s.running_mean, s.running_var, s.num_batches_tracked, s.W, s.C, s.Wn, s.eps, s.momentum};
    if (s.C > maxc) maxc = s.C;
  }
  g.npg = npg; g.nrun = nrun; g.chunks = (maxc + 31) / 32 > 0 ? (maxc + 31) / 32 : 1;
  g.stem_partial = stem_partial; g.stem_dw = stem_dw; g.stem_nblk = stem_nblk; g.stem_n = stem_n;
  int nb = (npg > nrun ? npg : nrun) * g.chunks;
  if (stem_partial && (stem_n + 7) / 8 > nb) nb = (stem_n + 7) / 8;
  g.nb = nb > 0 ? nb : 1;
  g.nsmall = (stem_partial ? 3 : 2) * g.nb;
  hipLaunchKernelGGL(wgrad_reduce_pgrad_kernel, dim3(nred + g.nsmall), dim3(256), 0, stream, t, g, accumulate);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// dW (+)= sum of slabs for n convolutions (descs: HOST array), 32 per launch.
int da_wgrad_reduce_multi(const da_wgrad_reduce_desc* descs, int n, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (n < 0 || (n && !descs)) return DA_EINVAL;
  for (int base = 0; base < n; base += 32) {
    WgradReduceTable t;
    int m = n - base < 32 ? n - base : 32, nred = 0;
    for (int i = 0; i < m; ++i) {
      const da_wgrad_reduce_desc& s = descs[base + i];
      if (!s.slab || !s.dw || s.splits < 1) return DA_EINVAL;
      if (s.ntaps < 1 || s.N < 1 || s.C < 1 || (s.N * s.C) % 4) return DA_EINVAL;
      t.d[i] = {s.slab, s.dw, s.splits, s.ntaps, s.N, s.C};
      t.first_block[i] = nred;
      nred += (s.ntaps * s.N * s.C + 1023) / 1024;
    }
    t.first_block[m] = nred;
    t.n = m;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(nred), dim3(256), 0, stream, t, accumulate);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

// dW[co][ci][k] (torch layout, k = ntaps) (+)= sum over positions dY[m][co] * X[src(m,k)][ci].
// dw == NULL: only the slabs are produced (deferred reduction).
int da_conv_wgrad(const float* dy, const float* x, float* dw, float* workspace, int rows, int Lm, int Ldy, int lddy,
                  int N, int Lx, int ldx, int C, int dy_stride, int dy_off, int src_stride, int ntaps,
                  const int* src_off, int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (!dy || !x || !workspace || ntaps < 1 || ntaps > 3) return DA_EINVAL;
  if (C % 32 || N % 32 || lddy % 4 || ldx % 4) return DA_EINVAL;
  WgradArgs a;
  a.dy = dy; a.x = x; a.slab = workspace;
  a.M = rows * Lm; a.Ldy = Ldy; a.lddy = lddy; a.N = N;
  a.Lx = Lx; a.ldx = ldx; a.C = C;
  a.dy_stride = dy_stride; a.dy_off = dy_off; a.src_stride = src_stride;
  a.ntaps = ntaps;
  a.so0 = src_off[0]; a.so1 = ntaps > 1 ? src_off[1] : 0; a.so2 = ntaps > 2 ? src_off[2] : 0;
  a.divLm = make_fastdiv((uint32_t)Lm);
  a.xform = 0; a.dy_half = 0; a.ldstat = 0;
  a.mean = a.invstd = a.gamma = a.beta = nullptr;
  a.divWn = make_fastdiv(1u);
  if ((uint64_t)a.M * (uint64_t)Lm >= 0xffffffffull) return DA_EINVAL;
  WgradPlan pl = wgrad_plan(a.M, N, C, ntaps);
  if (!pl.tn) return DA_EINVAL;
  a.kchunk = pl.kchunk;
  int sp = pl.splits;
  int tn = pl.tn, tc = pl.tc, rc;
  if (tn == 128 && tc == 128) rc = launch_wgrad<2, 2, 2, 2>(a, sp, stream);
  else if (tn == 128 && tc == 64) rc = launch_wgrad<2, 1, 2, 2>(a, sp, stream);
  else if (tn == 64 && tc == 128) rc = launch_wgrad<1, 2, 2, 2>(a, sp, stream);
  else if (tn == 64 && tc == 64) rc = launch_wgrad<1, 1, 2, 2>(a, sp, stream);
  else if (tn == 128 && tc == 32) rc = launch_wgrad<1, 1, 4, 1>(a, sp, stream);
  else if (tn == 32 && tc == 128) rc = launch_wgrad<1, 1, 1, 4>(a, sp, stream);
  else rc = DA_EINVAL;
  if (rc) return rc;
  if (!dw) return DA_OK;   // caller reduces the slabs later (da_wgrad_reduce_multi)
  int total = ntaps * N * C;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, stream, workspace, dw, sp, ntaps, N,
                     C, accumulate);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

}  // extern "C"
