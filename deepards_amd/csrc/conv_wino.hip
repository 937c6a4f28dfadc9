// k3 stride-1 pad-1 Conv1d (forward and data gradient) as Winograd F(2,3) on the fp32 matrix cores.
//
// Replaces the 3-tap nn.Conv1d calls of reference models/resnet.py:5-8,27-38 (BasicBlock conv1/conv2) and
// models/densenet.py:25-32 (growth conv) -- the bulk of the step's FLOPs -- forward and input gradient.
//
// Two neighbouring outputs of one sequence share their inputs: with d0..d3 = x[2i-1], x[2i], x[2i+1], x[2i+2]
// (zero outside the sequence) and taps g0, g1, g2,
//     m0 = (d0 - d2) g0            m1 = (d1 + d2) (g0 + g1 + g2)/2
//     m3 = (d1 - d3) g2            m2 = (d2 - d1) (g0 - g1 + g2)/2
//     y[2i] = m0 + m1 + m2         y[2i+1] = m1 - m2 - m3
// i.e. 4 channel contractions per output PAIR instead of 6: 2/3 of the direct convolution's MFMAs, all in fp32
// (measured error vs fp64 6e-7 of the output scale, direct fp32 4e-7).  The transformed taps U[4][N][C] come from
// the weight repack; the input transform is four VALU ops per fragment element at fragment-read time; the output
// transform runs on the accumulators in the epilogue.
//
// Rows of the GEMM are output pairs P = row * PL + i, PL = ceil(L / 2) (an odd L leaves the last pair of a sequence
// half used).  Staging keeps even and odd sequence positions in two LDS panels indexed by pair, so that the four
// inputs of a pair are unit-row-stride reads: d0 = O[P-1], d1 = E[P], d2 = O[P], d3 = E[P+1].
//
// Block = 64 pairs x 32 output channels, 4 waves of 16 pairs x 32 channels each: v_mfma_f32_16x16x4_f32 (same rate
// as 32x32x2), 8 accumulator tiles (4 products x 2 channel halves) = 32 VGPRs.  K step = 32 channels; LDS pitch 36.
#include "common.h"

struct WinoArgs {
  const float* x;   // [rows][L][ldx]
  const float* u;   // [4][N][C] transformed taps
  float* y;         // [rows][L][ldy]
  int MP, L, PL, ldx, C, ldy, N, accumulate;
  FastDiv divPL;
};

__device__ __forceinline__ int xcd_chunked(int id, int total) {   // same block order as conv_gemm.hip
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, s = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
}

#define WINO_PITCH 36
#define WINO_AROWS 66
#define WINO_LDS_FLOATS (2 * WINO_AROWS * WINO_PITCH + 4 * 32 * WINO_PITCH)

// MINI = false: the 64 pairs x 32 channels tile `tile`, wave = 16 pairs, whole K.
// MINI = true : half such a tile (32 pairs), wave = (16 pairs, one 16-channel half of every K step); the two partial
//               accumulator sets meet in LDS (fixed order).  Used for the partly filled last round of tiles, see
//               conv_gemm.hip "Tail tiles": a half tile holds its CU for a quarter of a full tile's time.
template <bool MINI>
__device__ __forceinline__ void conv3_wino_body(const WinoArgs& a, const int tile, const int sub, float* lds) {
  constexpr int PITCH = WINO_PITCH, PAIRS = MINI ? 32 : 64, AR = PAIRS + 2;
  float* Es = lds;                      // [AR][PITCH] even positions of pairs P0-1 .. P0+PAIRS
  float* Os = lds + AR * PITCH;         // [AR][PITCH] odd positions
  float* Us = lds + 2 * AR * PITCH;     // [4][32][PITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = a.N >> 5;
  const int P0 = (tile / ntn) * 64 + (MINI ? 32 * sub : 0), n_blk = (tile % ntn) * 32;
  const int lr = tid >> 3, lq = tid & 7;
  const int PL = a.PL;

  // loader: panel rows lr (+32) of both panels, the two rows past PAIRS by the first 32 threads (last slot)
  constexpr int NA = MINI ? 3 : 5;
  int aoff[NA];
  bool aok[NA];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const bool extra = p == NA - 1;
    const int e = extra ? PAIRS + (tid >> 4) : (MINI ? lr : lr + 32 * (p & 1));
    const int odd = extra ? ((tid >> 3) & 1) : (MINI ? p : (p >> 1));
    const int P = P0 - 1 + e;
    bool ok = P >= 0 && P < a.MP && (!extra || tid < 32);
    const uint32_t r = fdiv((uint32_t)(ok ? P : 0), a.divPL);
    const int i = (ok ? P : 0) - (int)r * PL;
    const int pos = 2 * i + odd;
    ok = ok && pos < a.L;
    aoff[p] = ((int)r * a.L + (ok ? pos : 0)) * a.ldx + lq * 4;
    aok[p] = ok;
  }
  const float* ub = a.u + (size_t)(n_blk + lr) * a.C + lq * 4;
  const size_t ustride = (size_t)a.N * a.C;

  f32x4 ra[NA], rb[4];
  auto gload = [&](int ks) {
    const int c0 = ks << 5;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (aok[p]) v = *reinterpret_cast<const f32x4*>(a.x + aoff[p] + c0);
      ra[p] = v;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const f32x4*>(ub + j * ustride + c0);
  };

  // fragment geometry (16x16x4: lane = (row l%16, k group l/16))
  const int prow = lane & 15, g = lane >> 4;
  const int wp = MINI ? (wave & 1) : wave, khalf = wave >> 1;
  const int pr = wp * 16 + prow + 1;                    // panel row of this lane's pair
  const int P_lane = P0 + wp * 16 + prow;
  const int Pc = P_lane < a.MP ? P_lane : 0;
  const int i_lane = Pc - (int)fdiv((uint32_t)Pc, a.divPL) * PL;
  const bool at_first = i_lane == 0, at_last = i_lane == PL - 1;

  f32x4 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kc = a.C >> 5;
  gload(0);
  for (int ks = 0; ks < kc; ++ks) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NA - 1; ++p) {
      const int e = MINI ? lr : lr + 32 * (p & 1);
      const int odd = MINI ? p : (p >> 1);
      *reinterpret_cast<f32x4*>(&(odd ? Os : Es)[e * PITCH + lq * 4]) = ra[p];
    }
    if (tid < 32)
      *reinterpret_cast<f32x4*>(&(((tid >> 3) & 1) ? Os : Es)[(PAIRS + (tid >> 4)) * PITCH + lq * 4]) = ra[NA - 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&Us[(j * 32 + lr) * PITCH + lq * 4]) = rb[j];
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int col = half * 16 + g * 4;
      if (half == 1) {                    // next chunk's loads late in the MFMA sequence (see GLOAD_AT in conv_gemm.hip)
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < kc) gload(ks + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MINI && half != khalf) continue;
      f32x4 d0 = *reinterpret_cast<const f32x4*>(&Os[(pr - 1) * PITCH + col]);
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(&Es[pr * PITCH + col]);
      const f32x4 d2 = *reinterpret_cast<const f32x4*>(&Os[pr * PITCH + col]);
      f32x4 d3 = *reinterpret_cast<const f32x4*>(&Es[(pr + 1) * PITCH + col]);
      if (at_first) d0 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (at_last) d3 = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 D[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        D[0][e] = d0[e] - d2[e];
        D[1][e] = d1[e] + d2[e];
        D[2][e] = d2[e] - d1[e];
        D[3][e] = d1[e] - d3[e];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 uf = *reinterpret_cast<const f32x4*>(&Us[(j * 32 + nt * 16 + prow) * PITCH + col]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(D[j][e], uf[e], acc[j][nt], 0, 0, 0);
        }
    }
  }

  if (MINI) {   // k half 1 hands its partial sums to k half 0
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds);          // [2 pair halves][8 tiles][64 lanes]
    if (khalf == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) red[((wp * 4 + j) * 2 + nt) * 64 + lane] = acc[j][nt];
    }
    __syncthreads();
    if (khalf == 1) return;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 o = red[((wp * 4 + j) * 2 + nt) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][nt][e] += o[e];
      }
  }

  // output transform + store: lane holds channel n = nt*16 + l%16 of pairs 4*(l/16) + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int P = P0 + wp * 16 + g * 4 + r;
    if (P >= a.MP) continue;
    const uint32_t rr = fdiv((uint32_t)P, a.divPL);
    const int i = P - (int)rr * PL;
    const bool has1 = 2 * i + 1 < a.L;
    float* y0p = a.y + ((size_t)rr * a.L + 2 * i) * a.ldy + n_blk + prow;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float y0 = acc[0][nt][r] + acc[1][nt][r] + acc[2][nt][r];
      float y1 = acc[1][nt][r] - acc[2][nt][r] - acc[3][nt][r];
      float* q0 = y0p + nt * 16;
      if (a.accumulate) {
        y0 += q0[0];
        if (has1) y1 += q0[a.ldy];
      }
      q0[0] = y0;
      if (has1) q0[a.ldy] = y1;
    }
  }
}

// tiles [0, full) as whole tiles; with nmini > 0 the tiles [full, full + nmini / 2) as half tiles in the first
// blocks of the launch (padded to a multiple of 8: block id % 8 stays the XCD of the full tiles)
__global__ __launch_bounds__(256) void conv3_wino_kernel(WinoArgs a, int nmini, int nmini_pad, int full) {
  __shared__ float lds[WINO_LDS_FLOATS];
  if ((int)blockIdx.x < nmini_pad) {
    if ((int)blockIdx.x < nmini) conv3_wino_body<true>(a, full + ((int)blockIdx.x >> 1), blockIdx.x & 1, lds);
    return;
  }
  conv3_wino_body<false>(a, xcd_chunked(blockIdx.x - nmini_pad, full), 0, lds);
}

static int g_wino_tail = 1;

// U[4][N][C] from torch-layout weights w[co][ci][3]:
//   forward  (transpose = 0): N = co, C = ci, taps g_t = w[n][c][t]
//   dgrad    (transpose = 1): N = ci, C = co, taps g_t = w[c][n][2 - t]   (dx[m] = sum_t dy[m + t - 1] w[..][2 - t])
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int co,
                                                          int ci, int transpose) {
  const int N = transpose ? ci : co, C = transpose ? co : ci;
  const size_t total = (size_t)N * C;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx / C), c = (int)(idx - (size_t)n * C);
  const float* src = transpose ? w + ((size_t)c * ci + n) * 3 : w + ((size_t)n * ci + c) * 3;
  const float g0 = transpose ? src[2] : src[0], g1 = src[1], g2 = transpose ? src[0] : src[2];
  u[idx] = g0;
  u[total + idx] = (g0 + g1 + g2) * 0.5f;
  u[2 * total + idx] = (g0 - g1 + g2) * 0.5f;
  u[3 * total + idx] = g2;
}

extern "C" {

// y (+)= conv1d(x, k = 3, stride 1, pad 1) per row with the transformed taps u (da_wino_weights).
// x: [rows][L][ldx] first C channels; y: [rows][L][ldy] first N channels.  replaces reference models/resnet.py:5-8
int da_conv3_winograd(const float* x, const float* u, float* y, int rows, int L, int ldx, int C, int ldy, int N,
                      int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (!x || !u || !y || rows < 0 || L < 1 || C % 32 || N % 32 || C < 32 || N < 32 || ldx % 4 || ldx < C || ldy < N)
    return DA_EINVAL;
  if (rows == 0) return DA_OK;
  if ((uint64_t)rows * L * (uint64_t)(ldx > ldy ? ldx : ldy) >= 0x7fffffffull) return DA_EINVAL;   // 32-bit element offsets
  WinoArgs a;
  a.x = x; a.u = u; a.y = y;
  a.L = L; a.PL = (L + 1) / 2; a.MP = rows * a.PL;
  a.ldx = ldx; a.C = C; a.ldy = ldy; a.N = N; a.accumulate = accumulate;
  a.divPL = make_fastdiv((uint32_t)a.PL);
  if ((uint64_t)a.MP * (uint64_t)a.PL >= 0xffffffffull) return DA_EINVAL;
  const int tiles = ((a.MP + 63) / 64) * (N / 32);
  const int R = tiles % 256;
  int nmini = 0, full = tiles;
  if (g_wino_tail && tiles > 256 && R >= 1 && R <= 128) {
    nmini = 2 * R;
    full = tiles - R;
  }
  const int nmini_pad = (nmini + 7) / 8 * 8;
  hipLaunchKernelGGL(conv3_wino_kernel, dim3(nmini_pad + full), dim3(256), 0, stream, a, nmini, nmini_pad, full);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// tuning / tests: 0 = no half tiles for the last round
int da_wino_debug_tail(int on) {
  g_wino_tail = on;
  return DA_OK;
}

// transformed taps for da_conv3_winograd from torch-layout weights w[co][ci][3]; u: 4*co*ci floats.
// transpose = 0: forward taps U[4][co][ci]; transpose = 1: data-gradient taps U[4][ci][co].
int da_wino_weights(const float* w, float* u, int co, int ci, int transpose, hipStream_t stream) {
  DA_ENTER();
  if (!w || !u || co < 1 || ci < 1) return DA_EINVAL;
  const size_t total = (size_t)co * ci;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, u, co, ci,
                     transpose);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

}  // extern "C"
