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
#include <type_traits>

struct WinoArgs {
  const float* x;   // [rows][L][ldx]
  const float* u;   // [4][N][C] transformed taps
  float* y;         // [rows][L][ldy]
  int MP, L, PL, ldx, C, ldy, N, accumulate;
  FastDiv divPL;
  // F.dropout on the conv's output in the epilogue (a _DenseLayer's growth conv writes its new features straight into the
  // block's buffer, densenet.py:36-40): the keep mask of da_dropout on the contiguous [rows * L][N] tensor; p = 0: off
  const long long* drop_seed;
  uint32_t drop_salt;
  float drop_p;
  // per-(tile, window) statistics records of the OUTPUT (after the dropout), for the BatchNorms that will normalise these
  // channels (common.h StatRecords): stat_part != NULL -> every 64-pair tile writes (count, mean, centred M2) per channel
  // for the <= 2 windows (stat_Wu pairs each, >= 64) it touches; F(2,3) full tiles only (the host launches no half tiles)
  float* stat_part;
  int stat_Wu;
  // XFW (conv3_wino_bn_kernel): x is the OUTPUT of the 1x1 conv in front (densenet.py:27-31 norm2 -> relu2 -> conv2) and
  // h = max(fmaf(x, sc, sh), 0) is applied while x is staged -- relu(norm2(.)) is never stored.  The statistics of x arrive
  // as the records the 1x1 conv's epilogue wrote (in_pend: in_tiles tiles of 64 POSITIONS, in_Wu positions per window);
  // every block merges them for its two windows, the block that holds a window's first pair publishes mean / invstd to
  // in_mean / in_invstd ([W][C]) for the backward kernels.  C <= 128.
  const float* in_pend;
  float* in_mean;
  float* in_invstd;
  const float* in_gamma;
  const float* in_beta;
  int in_tiles, in_Wu;
  float in_eps;
  int tapmod;      // timing experiment only (da_wino_debug_tapmod): F(4,3) K-16 kernel reads its taps modulo this many channels
};

__device__ __forceinline__ uint32_t wino_mix32(uint32_t a, uint32_t b) {      // head_optim.hip mix32
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

__device__ __forceinline__ int xcd_chunked(int id, int total) {   // same block order as conv_gemm.hip
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, s = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
}

// MFMA row (0..15) -> row of the 16-row tile it holds: rows 4..11 the even ones, 0..3 and 12..15 the odd ones
__device__ __forceinline__ int wino_row(int i) {
  return (i >= 4 && i < 12) ? 2 * (i - 4) : (i < 4 ? 2 * i + 1 : 2 * (i - 12) + 9);
}

#ifndef WINO_SPLIT_GLOAD
#define WINO_SPLIT_GLOAD 1   // tap loads in the first half of the MFMAs, activation loads in the second (+2..5 %)
#endif
#define WINO_PITCH 36
#define WINO_AROWS 66
#define WINO_LDS_FLOATS (2 * WINO_AROWS * WINO_PITCH + 4 * 32 * WINO_PITCH)

// MINI = false: the 64 pairs x 32 channels tile `tile`, wave = 16 pairs, whole K.
// MINI = true : half such a tile (32 pairs), wave = (16 pairs, one 16-channel half of every K step); the two partial
//               accumulator sets meet in LDS (fixed order).  Used for the partly filled last round of tiles, see
//               conv_gemm.hip "Tail tiles": a half tile holds its CU for a quarter of a full tile's time.
#define WINO_STAT_FLOATS (4 * 2 * 2 * 32 + 8)           // the records' fold area behind the operand panels
#define WINO_XF_FLOATS (2 * 2 * 128)                  // [2 window slots][{sc, sh}][C <= 128] behind that
template <bool MINI, bool STATS = false, bool XFW = false, bool DROP = false>
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
  int aoff[NA], aslot[NA];
  bool aok[NA];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const bool extra = p == NA - 1;
    const int e = extra ? PAIRS + (tid >> 4) : (MINI ? lr : lr + 32 * (p & 1));
    const int odd = extra ? ((tid >> 3) & 1) : (MINI ? p : (p >> 1));
    const int P = P0 - 1 + e;
    aslot[p] = 0;
    bool ok = P >= 0 && P < a.MP && (!extra || tid < 32);
    const uint32_t r = fdiv((uint32_t)(ok ? P : 0), a.divPL);
    const int i = (ok ? P : 0) - (int)r * PL;
    const int pos = 2 * i + odd;
    ok = ok && pos < a.L;
    aoff[p] = ((int)r * a.L + (ok ? pos : 0)) * a.ldx + lq * 4;
    aok[p] = ok;
    if (XFW) {   // window slot of the row's pair (a halo pair of another window is masked at the sequence edge anyway)
      const int sl = (P >= 0 ? P : 0) / a.stat_Wu - P0 / a.stat_Wu;
      aslot[p] = (sl < 0 ? 0 : (sl > 1 ? 1 : sl)) * 2 * a.C + lq * 4;
    }
  }
  const float* ub = a.u + (size_t)(n_blk + lr) * a.C + lq * 4;
  const size_t ustride = (size_t)a.N * a.C;
  float* xtab = lds + WINO_LDS_FLOATS + WINO_STAT_FLOATS;        // XFW: [2 slots][{sc, sh}][C]

  f32x4 ra[NA], rb[4];
  auto gload_a = [&](int ks) {
    const int c0 = ks << 5;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (aok[p]) v = *reinterpret_cast<const f32x4*>(a.x + aoff[p] + c0);
      ra[p] = v;
    }
  };
  auto gload_b = [&](int ks) {
    const int c0 = ks << 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const f32x4*>(ub + j * ustride + c0);
  };
  auto gload = [&](int ks) {
    gload_a(ks);
    gload_b(ks);
  };

  // fragment geometry (16x16x4: lane = (row l%16, k group l/16)).  ds_read_b128 is served in the lane groups
  // {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): half of a group reads k group g, the other
  // half g+1, from 16 different rows.  With pitch 36 a row step moves 9 16-byte slots, so a group is conflict-free
  // when (a) neighbouring k groups sit TWO slots apart -- k group g of 16-channel half h = slot 2g + h -- and
  // (b) MFMA rows 4..11 hold even tile rows and rows 0..3, 12..15 odd ones (wino_row): one half of the group then
  // covers the even slots, the other the odd slots.  The same permutation orders the output channels of a B tile.
  const int prow = wino_row(lane & 15), g = lane >> 4;
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
  if (XFW) {   // the scale / shift vectors of this tile's two windows, from the records (first operand loads in flight)
    const int w0 = P0 / a.stat_Wu, nwin = a.MP / a.stat_Wu;
    for (int i = tid; i < 2 * a.C; i += 256) {
      const int ws = i >= a.C ? 1 : 0, c = i - ws * a.C, w = w0 + ws;
      float sc = 0.f, sh = 0.f;
      if (w < nwin) {
        float mu, is;
        merge_stat_records(a.in_pend, a.in_tiles, a.C, a.in_Wu, w, c, a.in_eps, mu, is);
        const int wP = w * a.stat_Wu;                     // the window's first pair: its tile publishes
        if (n_blk == 0 && wP >= P0 && wP < P0 + 64) {
          a.in_mean[(size_t)w * a.C + c] = mu;
          a.in_invstd[(size_t)w * a.C + c] = is;
        }
        bn_scale_shift(mu, is, a.in_gamma[c], a.in_beta[c], sc, sh);
      }
      xtab[(ws * 2) * a.C + c] = sc;
      xtab[(ws * 2 + 1) * a.C + c] = sh;
    }
  }
  auto xf = [&](int p, int ks) -> f32x4 {                 // the staged value of loader slot p
    if (!XFW) return ra[p];
    const f32x4 sc = *reinterpret_cast<const f32x4*>(&xtab[aslot[p] + (ks << 5)]);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(&xtab[aslot[p] + a.C + (ks << 5)]);
    f32x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = fmaxf(fmaf(ra[p][e], sc[e], sh[e]), 0.f);
    return aok[p] ? h : f32x4{0.f, 0.f, 0.f, 0.f};        // padding and positions past the sequence stay zeros
  };
  for (int ks = 0; ks < kc; ++ks) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NA - 1; ++p) {
      const int e = MINI ? lr : lr + 32 * (p & 1);
      const int odd = MINI ? p : (p >> 1);
      *reinterpret_cast<f32x4*>(&(odd ? Os : Es)[e * PITCH + lq * 4]) = xf(p, ks);
    }
    if (tid < 32)
      *reinterpret_cast<f32x4*>(&(((tid >> 3) & 1) ? Os : Es)[(PAIRS + (tid >> 4)) * PITCH + lq * 4]) = xf(NA - 1, ks);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&Us[(j * 32 + lr) * PITCH + lq * 4]) = rb[j];
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int col = (2 * g + half) * 4;
#if WINO_SPLIT_GLOAD
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < kc) {
        if (half == 0) gload_b(ks + 1);
        else gload_a(ks + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#else
      if (half == 1) {                    // next chunk's loads late in the MFMA sequence (see GLOAD_AT in conv_gemm.hip)
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < kc) gload(ks + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
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

  // output transform + store: lane holds channel n = nt*16 + wino_row(l%16) of the pairs wino_row(4*(l/16) + r)
  const bool drop = (DROP || STATS || XFW) && a.drop_p > 0.f;      // (the plain kernel carries no dropout code: DROP = a form of its own)
  uint32_t dkey = 0u, dthr = 0u;
  float dscale = 1.f;
  if (drop) {
    const long long sd = a.drop_seed[0];
    dkey = wino_mix32((uint32_t)sd ^ (uint32_t)(sd >> 32), a.drop_salt);
    dthr = (uint32_t)(a.drop_p * 4294967296.0);
    dscale = 1.0f / (1.0f - a.drop_p);
  }
  constexpr bool stats = STATS && !MINI;                // (a kernel of its own: the records cost registers)
  const int edgeP = stats ? (P0 / a.stat_Wu + 1) * a.stat_Wu : 0x7fffffff;      // pairs from here on: the tile's 2nd window
  // records in ONE pass: sums of (y - pivot) and (y - pivot)^2 per (window slot, channel half), the pivot being this wave's
  // own first output of the channel -- a value of the distribution, so the variance below loses no digits to cancellation
  float piv[2] = {0.f, 0.f}, s1[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, scnt[2] = {0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int P = P0 + wp * 16 + wino_row(g * 4 + r);
    const bool live = P < a.MP;
    const uint32_t rr = fdiv((uint32_t)(live ? P : 0), a.divPL);
    const int i = (live ? P : 0) - (int)rr * PL;
    const bool has1 = 2 * i + 1 < a.L;
    float* y0p = a.y + ((size_t)rr * a.L + 2 * i) * a.ldy + n_blk + prow;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float y0 = acc[0][nt][r] + acc[1][nt][r] + acc[2][nt][r];
      float y1 = acc[1][nt][r] - acc[2][nt][r] - acc[3][nt][r];
      if (drop) {                                        // element (position, channel) of the contiguous [rows L][N] tensor
        const size_t e0 = ((size_t)rr * a.L + 2 * i) * (size_t)a.N + n_blk + prow + nt * 16, e1 = e0 + a.N;
        y0 = wino_mix32(dkey, (uint32_t)e0 ^ (uint32_t)(e0 >> 32) * 0x27d4eb2fu) >= dthr ? y0 * dscale : 0.f;
        y1 = wino_mix32(dkey, (uint32_t)e1 ^ (uint32_t)(e1 >> 32) * 0x27d4eb2fu) >= dthr ? y1 * dscale : 0.f;
      }
      float* q0 = y0p + nt * 16;
      if (live) {
        if (a.accumulate) {
          y0 += q0[0];
          if (has1) y1 += q0[a.ldy];
        }
        q0[0] = y0;
        if (has1) q0[a.ldy] = y1;
      }
      if (stats) {
        if (r == 0) piv[nt] = __shfl(live ? y0 : 0.f, (lane & 15) + 16, 64);      // pair row 0 of the wave's tile: g = 1, r = 0
        if (live) {
          const int sl = P >= edgeP ? 1 : 0;
          const float d0 = y0 - piv[nt], d1 = has1 ? y1 - piv[nt] : 0.f;
          s1[sl][nt] += d0 + d1;
          s2[sl][nt] += d0 * d0 + d1 * d1;
          if (nt == 0) scnt[sl] += has1 ? 2.f : 1.f;
        }
      }
    }
  }
  if (!stats) return;
  {
    // fold over the 4 lanes that share a channel, turn the wave's sums into (count, mean, centred M2), fold the 4 waves
    // with Chan's update in wave order through LDS (a region of its own behind the operand panels: no barrier to free it)
    float* wrec = lds + WINO_LDS_FLOATS;                 // [4 waves][2 slots][{mean, M2}][32 ch] | cnt [4][2]
    float* wcnt = wrec + 4 * 2 * 2 * 32;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        s1[sl][nt] += __shfl_xor(s1[sl][nt], 16, 64);
        s1[sl][nt] += __shfl_xor(s1[sl][nt], 32, 64);
        s2[sl][nt] += __shfl_xor(s2[sl][nt], 16, 64);
        s2[sl][nt] += __shfl_xor(s2[sl][nt], 32, 64);
      }
      scnt[sl] += __shfl_xor(scnt[sl], 16, 64);
      scnt[sl] += __shfl_xor(scnt[sl], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float cn = scnt[sl], inv = cn > 0.f ? 1.f / cn : 0.f;
          float* o = wrec + ((wave * 2 + sl) * 2) * 32 + nt * 16 + prow;
          o[0] = piv[nt] + s1[sl][nt] * inv;
          o[32] = fmaxf(s2[sl][nt] - s1[sl][nt] * s1[sl][nt] * inv, 0.f);
        }
    }
    if (lane == 0) {
      wcnt[wave * 2] = scnt[0];
      wcnt[wave * 2 + 1] = scnt[1];
    }
    __syncthreads();
    if (tid < 64) {
      const int sl = tid >> 5, ch = tid & 31;
      float n = 0.f, mu = 0.f, m2 = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        const float nb = wcnt[wv * 2 + sl];
        if (nb > 0.f) {
          const float* o = wrec + ((wv * 2 + sl) * 2) * 32 + ch;
          const float d = o[0] - mu, nt_ = n + nb;
          mu += d * (nb / nt_);
          m2 += o[32] + d * d * (n * nb / nt_);
          n = nt_;
        }
      }
      const int mt = P0 >> 6, nmt = (a.MP + 63) >> 6;
      float* rec = a.stat_part + ((size_t)(mt * 2 + sl) * 2) * a.N + n_blk + ch;      // [m tile][slot][{mean, M2}][N]
      rec[0] = mu;
      rec[a.N] = m2;
      if (ch == 0 && n_blk == 0) a.stat_part[(size_t)nmt * 4 * a.N + mt * 2 + sl] = n;
    }
  }
}

// ... and with relu(norm(x)) applied while x is staged (WinoArgs.in_pend): the growth conv of a _DenseLayer reading the 1x1
// conv's output directly
template <bool STATS>
__global__ __launch_bounds__(256, 3) void conv3_wino_bn_kernel(WinoArgs a, int full) {
  __shared__ float lds[WINO_LDS_FLOATS + WINO_STAT_FLOATS + WINO_XF_FLOATS];
  conv3_wino_body<false, STATS, true>(a, xcd_chunked(blockIdx.x, full), 0, lds);
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

// ... with F.dropout in the epilogue and no statistics records (a dense layer outside the fused block path)
__global__ __launch_bounds__(256) void conv3_wino_drop_kernel(WinoArgs a, int full) {
  __shared__ float lds[WINO_LDS_FLOATS];
  conv3_wino_body<false, false, false, true>(a, xcd_chunked(blockIdx.x, full), 0, lds);
}

// the same tiles (whole ones only) with the statistics records of the output written from the epilogue (WinoArgs.stat_part)
__global__ __launch_bounds__(256, 4) void conv3_wino_stats_kernel(WinoArgs a, int full) {
  __shared__ float lds[WINO_LDS_FLOATS + WINO_STAT_FLOATS];
  conv3_wino_body<false, true>(a, xcd_chunked(blockIdx.x, full), 0, lds);
}

// ---------------------------------------------------------------------------------------------
// The same convolution as Winograd F(4,3): four neighbouring outputs from six inputs d0..d5 = x[4i-1 .. 4i+4],
//     D0 = 4 d0 - 5 d2 + d4                  U0 = g0 / 4
//     D1 = (d4 - 4 d2) + (d3 - 4 d1)         U1 = -(g0 + g1 + g2) / 6
//     D2 = (d4 - 4 d2) - (d3 - 4 d1)         U2 = -(g0 - g1 + g2) / 6
//     D3 = (d4 - d2) + 2 (d3 - d1)           U3 = g0 / 24 + g1 / 12 + g2 / 6
//     D4 = (d4 - d2) - 2 (d3 - d1)           U4 = g0 / 24 - g1 / 12 + g2 / 6
//     D5 = 4 d1 - 5 d3 + d5                  U5 = g2
//     y0 = m0 + m1 + m2 + m3 + m4            y1 = (m1 - m2) + 2 (m3 - m4)
//     y2 = (m1 + m2) + 4 (m3 + m4)           y3 = (m1 - m2) + 8 (m3 - m4) + m5
// 6 channel contractions per output QUAD instead of 12: half the direct convolution's MFMAs (3/4 of F(2,3)'s), fp32
// error vs fp64 3e-6 of the output scale (F(2,3) 6e-7).  GEMM rows are output quads Q = row * QL + i, QL = ceil(L/4);
// staging keeps the four position phases in four quad-indexed LDS panels, so d0 = Ph3[Q-1], d1..d4 = Ph0..Ph3[Q],
// d5 = Ph0[Q+1] are unit-row-stride reads with the F(2,3) kernel's conflict-free lane geometry.
// Block = 64 quads x 32 output channels, 4 waves of 16 quads: 12 accumulator tiles = 48 VGPRs; 64.8 KB of LDS.
// ---------------------------------------------------------------------------------------------
// B^T d as whole-vector fp32 arithmetic (the compiler pairs it into v_pk_fma_f32 / v_pk_add_f32 where it can)
__device__ __forceinline__ void wino4_input_transform(const f32x4& d0, const f32x4& d1, const f32x4& d2,
                                                      const f32x4& d3, const f32x4& d4, const f32x4& d5, f32x4* D) {
  const f32x4 p = d4 - 4.f * d2, q = d3 - 4.f * d1;
  const f32x4 r = d4 - d2, t = d3 - d1;
  D[0] = 4.f * d0 + (d4 - 5.f * d2);
  D[1] = p + q;
  D[2] = p - q;
  D[3] = r + 2.f * t;
  D[4] = r - 2.f * t;
  D[5] = 4.f * d1 + (d5 - 5.f * d3);
}

#define W4_LDS_FLOATS ((4 * 64 + 2) * WINO_PITCH + 6 * 32 * WINO_PITCH)

template <bool MINI>
__device__ __forceinline__ void conv3_wino4_body(const WinoArgs& a, const int tile, const int sub, float* lds) {
  constexpr int PITCH = WINO_PITCH, QUADS = MINI ? 32 : 64;
  float* Ph0 = lds;                              // [QUADS + 1][PITCH] positions 4i     of quads Q0 .. Q0+QUADS
  float* Ph1 = Ph0 + (QUADS + 1) * PITCH;        // [QUADS][PITCH]     positions 4i + 1
  float* Ph2 = Ph1 + QUADS * PITCH;              // [QUADS][PITCH]     positions 4i + 2
  float* Ph3 = Ph2 + QUADS * PITCH;              // [QUADS + 1][PITCH] positions 4i + 3 of quads Q0-1 .. Q0+QUADS-1
  float* Us = Ph3 + (QUADS + 1) * PITCH;         // [6][32][PITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = a.N >> 5;
  const int Q0 = (tile / ntn) * 64 + (MINI ? 32 * sub : 0), n_blk = (tile % ntn) * 32;
  const int lr = tid >> 3, lq = tid & 7;
  const int QL = a.PL;

  constexpr int NA = MINI ? 5 : 9;
  int aoff[NA];
  bool aok[NA];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const bool extra = p == NA - 1;
    const int phase = extra ? (((tid >> 3) & 1) ? 3 : 0) : (MINI ? p : (p >> 1));
    const int e = extra ? (phase == 0 ? QUADS : -1) : (MINI ? lr : lr + 32 * (p & 1));
    const int Q = Q0 + e;
    bool ok = Q >= 0 && Q < a.MP && (!extra || tid < 16);
    const uint32_t r = fdiv((uint32_t)(ok ? Q : 0), a.divPL);
    const int i = (ok ? Q : 0) - (int)r * QL;
    const int pos = 4 * i + phase;
    ok = ok && pos < a.L;
    aoff[p] = ((int)r * a.L + (ok ? pos : 0)) * a.ldx + lq * 4;
    aok[p] = ok;
  }
  const float* ub = a.u + (size_t)(n_blk + lr) * a.C + lq * 4;
  const size_t ustride = (size_t)a.N * a.C;

  f32x4 ra[NA], rb[6];
  auto gload_a = [&](int ks) {
    const int c0 = ks << 5;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (aok[p]) v = *reinterpret_cast<const f32x4*>(a.x + aoff[p] + c0);
      ra[p] = v;
    }
  };
  auto gload_b = [&](int ks) {
    const int c0 = ks << 5;
#pragma unroll
    for (int j = 0; j < 6; ++j) rb[j] = *reinterpret_cast<const f32x4*>(ub + j * ustride + c0);
  };

  const int prow = wino_row(lane & 15), g = lane >> 4;
  const int wp = MINI ? (wave & 1) : wave, khalf = wave >> 1;
  const int pr = wp * 16 + prow;                        // quad of this lane within the block
  const int Q_lane = Q0 + pr;
  const int Qc = Q_lane < a.MP ? Q_lane : 0;
  const int i_lane = Qc - (int)fdiv((uint32_t)Qc, a.divPL) * QL;
  const bool at_first = i_lane == 0, at_last = i_lane == QL - 1;

  f32x4 acc[6][2];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kc = a.C >> 5;
  gload_a(0);
  gload_b(0);
  for (int ks = 0; ks < kc; ++ks) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NA - 1; ++p) {
      const int phase = MINI ? p : (p >> 1);
      const int e = MINI ? lr : lr + 32 * (p & 1);
      float* panel = phase == 0 ? Ph0 : (phase == 1 ? Ph1 : (phase == 2 ? Ph2 : Ph3 + PITCH));
      *reinterpret_cast<f32x4*>(&panel[e * PITCH + lq * 4]) = ra[p];
    }
    if (tid < 16) *reinterpret_cast<f32x4*>(&(((tid >> 3) & 1) ? Ph3 : Ph0 + QUADS * PITCH)[lq * 4]) = ra[NA - 1];
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(&Us[(j * 32 + lr) * PITCH + lq * 4]) = rb[j];
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int col = (2 * g + half) * 4;
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < kc) {
        if (half == 0) gload_b(ks + 1);
        else gload_a(ks + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MINI && half != khalf) continue;
      f32x4 d0 = *reinterpret_cast<const f32x4*>(&Ph3[pr * PITCH + col]);
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(&Ph0[pr * PITCH + col]);
      const f32x4 d2 = *reinterpret_cast<const f32x4*>(&Ph1[pr * PITCH + col]);
      const f32x4 d3 = *reinterpret_cast<const f32x4*>(&Ph2[pr * PITCH + col]);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(&Ph3[(pr + 1) * PITCH + col]);
      f32x4 d5 = *reinterpret_cast<const f32x4*>(&Ph0[(pr + 1) * PITCH + col]);
      if (at_first) d0 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (at_last) d5 = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 D[6];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = fmaf(-4.f, d2[e], d4[e]), q = fmaf(-4.f, d1[e], d3[e]);
        const float r = d4[e] - d2[e], t = 2.f * (d3[e] - d1[e]);
        D[0][e] = fmaf(4.f, d0[e], fmaf(-5.f, d2[e], d4[e]));
        D[1][e] = p + q;
        D[2][e] = p - q;
        D[3][e] = r + t;
        D[4][e] = r - t;
        D[5][e] = fmaf(4.f, d1[e], fmaf(-5.f, d3[e], d5[e]));
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
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
    f32x4* red = reinterpret_cast<f32x4*>(lds);          // [2 quad halves][12 tiles][64 lanes]
    if (khalf == 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) red[((wp * 6 + j) * 2 + nt) * 64 + lane] = acc[j][nt];
    }
    __syncthreads();
    if (khalf == 1) return;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 o = red[((wp * 6 + j) * 2 + nt) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][nt][e] += o[e];
      }
  }

#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int Q = Q0 + wp * 16 + wino_row(g * 4 + r);
    if (Q >= a.MP) continue;
    const uint32_t rr = fdiv((uint32_t)Q, a.divPL);
    const int i = Q - (int)rr * QL;
    const int nvalid = a.L - 4 * i;                     // outputs of this quad inside the sequence (>= 1)
    float* yp = a.y + ((size_t)rr * a.L + 4 * i) * a.ldy + n_blk + prow;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r], m4 = acc[4][nt][r],
                  m5 = acc[5][nt][r];
      const float sa = m1 + m2, da = m1 - m2, sb = m3 + m4, db = m3 - m4;
      float yv[4] = {m0 + sa + sb, fmaf(2.f, db, da), fmaf(4.f, sb, sa), fmaf(8.f, db, da) + m5};
      float* q0 = yp + nt * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < nvalid) {
          float v = yv[s];
          if (a.accumulate) v += q0[(size_t)s * a.ldy];
          q0[(size_t)s * a.ldy] = v;
        }
      }
    }
  }
}

// K step of 16 channels: 36 KB of LDS instead of 65 (pitch 20 = 5 slots a row keeps odd / even tile rows on odd / even
// slot residues like pitch 36; the lane groups of ds_read_b128 pair k groups (0,1) and (2,3), so k group g sits in slot
// {0,2,1,3}[g]).
#define W4K_PITCH 20
#define W4K_LDS_FLOATS ((4 * 64 + 2) * W4K_PITCH + 6 * 32 * W4K_PITCH)

template <bool MINI>
__device__ __forceinline__ void conv3_wino4k_body(const WinoArgs& a, const int tile, const int sub, float* lds) {
  constexpr int PITCH = W4K_PITCH, QUADS = MINI ? 32 : 64;
  float* Ph0 = lds;
  float* Ph1 = Ph0 + (QUADS + 1) * PITCH;
  float* Ph2 = Ph1 + QUADS * PITCH;
  float* Ph3 = Ph2 + QUADS * PITCH;              // row e + 1 = quad Q0 + e
  float* Us = Ph3 + (QUADS + 1) * PITCH;         // [6][32][PITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = a.N >> 5;
  const int Q0 = (tile / ntn) * 64 + (MINI ? 32 * sub : 0), n_blk = (tile % ntn) * 32;
  // loader rows: ds_write_b128 is served 8 lanes at a time on 32 banks, and with 5 slots a row two NEIGHBOURING rows
  // overlap in one bank group; rows 4 apart do not, so lanes 4..7 of every 8 take the row 4 below lanes 0..3
  const int lm = tid >> 3;
  const int lr = (lm >> 2) * 8 + (lm & 3) + 4 * ((tid >> 2) & 1), lq = tid & 3;
  const int QL = a.PL;

  constexpr int NP = MINI ? 2 : 4, NA = NP + 1;
  int aoff[NA];
  bool aok[NA];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const bool extra = p == NA - 1;
    const int idx = p * 64 + lr;
    const int phase = extra ? (((tid >> 2) & 1) ? 3 : 0) : idx / QUADS;
    const int e = extra ? (phase == 0 ? QUADS : -1) : idx % QUADS;
    const int Q = Q0 + e;
    bool ok = Q >= 0 && Q < a.MP && (!extra || tid < 8);
    const uint32_t r = fdiv((uint32_t)(ok ? Q : 0), a.divPL);
    const int i = (ok ? Q : 0) - (int)r * QL;
    const int pos = 4 * i + phase;
    ok = ok && pos < a.L;
    aoff[p] = ((int)r * a.L + (ok ? pos : 0)) * a.ldx + lq * 4;
    aok[p] = ok;
  }
  // taps: 192 rows (point j, channel n) in 3 passes of 64
  const float* ub[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int idx = p * 64 + lr;
    const int nrow = n_blk + (idx & 31);
    ub[p] = a.u + (size_t)(idx >> 5) * a.N * a.C + (size_t)(a.tapmod ? nrow % a.tapmod : nrow) * a.C + lq * 4;
  }

  f32x4 ra[NA], rb[3];
  auto gload_a = [&](int ks) {
    const int c0 = ks << 4;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (aok[p]) v = *reinterpret_cast<const f32x4*>(a.x + aoff[p] + c0);
      ra[p] = v;
    }
  };
  auto gload_b = [&](int ks) {
    const int c0 = ks << 4;
#pragma unroll
    for (int p = 0; p < 3; ++p) rb[p] = *reinterpret_cast<const f32x4*>(ub[p] + c0);
  };

  const int prow = wino_row(lane & 15), g = lane >> 4;
  const int col = (((g & 1) << 1) | (g >> 1)) * 4;       // slot {0,2,1,3}[g]
  const int wp = MINI ? (wave & 1) : wave, khalf = wave >> 1;
  const int pr = wp * 16 + prow;
  const int Q_lane = Q0 + pr;
  const int Qc = Q_lane < a.MP ? Q_lane : 0;
  const int i_lane = Qc - (int)fdiv((uint32_t)Qc, a.divPL) * QL;
  const bool at_first = i_lane == 0, at_last = i_lane == QL - 1;

  f32x4 acc[6][2];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kc = a.C >> 4;
  gload_a(0);
  gload_b(0);
  for (int ks = 0; ks < kc; ++ks) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int idx = p * 64 + lr;
      const int phase = idx / QUADS, e = idx % QUADS;
      float* panel = phase == 0 ? Ph0 : (phase == 1 ? Ph1 : (phase == 2 ? Ph2 : Ph3 + PITCH));
      *reinterpret_cast<f32x4*>(&panel[e * PITCH + lq * 4]) = ra[p];
    }
    if (tid < 8) *reinterpret_cast<f32x4*>(&(((tid >> 2) & 1) ? Ph3 : Ph0 + QUADS * PITCH)[lq * 4]) = ra[NA - 1];
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<f32x4*>(&Us[(p * 64 + lr) * PITCH + lq * 4]) = rb[p];
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < kc) {
      gload_b(ks + 1);
      gload_a(ks + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MINI && (ks & 1) != khalf) continue;
    f32x4 d0 = *reinterpret_cast<const f32x4*>(&Ph3[pr * PITCH + col]);
    const f32x4 d1 = *reinterpret_cast<const f32x4*>(&Ph0[pr * PITCH + col]);
    const f32x4 d2 = *reinterpret_cast<const f32x4*>(&Ph1[pr * PITCH + col]);
    const f32x4 d3 = *reinterpret_cast<const f32x4*>(&Ph2[pr * PITCH + col]);
    const f32x4 d4 = *reinterpret_cast<const f32x4*>(&Ph3[(pr + 1) * PITCH + col]);
    f32x4 d5 = *reinterpret_cast<const f32x4*>(&Ph0[(pr + 1) * PITCH + col]);
    if (at_first) d0 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (at_last) d5 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 D[6];
    wino4_input_transform(d0, d1, d2, d3, d4, d5, D);
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 uf = *reinterpret_cast<const f32x4*>(&Us[(j * 32 + nt * 16 + prow) * PITCH + col]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(D[j][e], uf[e], acc[j][nt], 0, 0, 0);
      }
  }

  if (MINI) {
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds);          // [2 quad halves][12 tiles][64 lanes] = 24.6 KB
    if (khalf == 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) red[((wp * 6 + j) * 2 + nt) * 64 + lane] = acc[j][nt];
    }
    __syncthreads();
    if (khalf == 1) return;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 o = red[((wp * 6 + j) * 2 + nt) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][nt][e] += o[e];
      }
  }

#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int Q = Q0 + wp * 16 + wino_row(g * 4 + r);
    if (Q >= a.MP) continue;
    const uint32_t rr = fdiv((uint32_t)Q, a.divPL);
    const int i = Q - (int)rr * QL;
    const int nvalid = a.L - 4 * i;
    float* yp = a.y + ((size_t)rr * a.L + 4 * i) * a.ldy + n_blk + prow;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r], m4 = acc[4][nt][r],
                  m5 = acc[5][nt][r];
      const float sa = m1 + m2, da = m1 - m2, sb = m3 + m4, db = m3 - m4;
      float yv[4] = {m0 + sa + sb, fmaf(2.f, db, da), fmaf(4.f, sb, sa), fmaf(8.f, db, da) + m5};
      float* q0 = yp + nt * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < nvalid) {
          float v = yv[s];
          if (a.accumulate) v += q0[(size_t)s * a.ldy];
          q0[(size_t)s * a.ldy] = v;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256, 3) void conv3_wino4k_kernel(WinoArgs a, int nmini, int nmini_pad, int full) {
  __shared__ float lds[W4K_LDS_FLOATS];
  if ((int)blockIdx.x < nmini_pad) {
    if ((int)blockIdx.x < nmini) conv3_wino4k_body<true>(a, full + ((int)blockIdx.x >> 1), blockIdx.x & 1, lds);
    return;
  }
  conv3_wino4k_body<false>(a, xcd_chunked(blockIdx.x - nmini_pad, full), 0, lds);
}

__global__ __launch_bounds__(256) void conv3_wino4_kernel(WinoArgs a, int nmini, int nmini_pad, int full) {
  __shared__ float lds[W4_LDS_FLOATS];
  if ((int)blockIdx.x < nmini_pad) {
    if ((int)blockIdx.x < nmini) conv3_wino4_body<true>(a, full + ((int)blockIdx.x >> 1), blockIdx.x & 1, lds);
    return;
  }
  conv3_wino4_body<false>(a, xcd_chunked(blockIdx.x - nmini_pad, full), 0, lds);
}

static int g_wino_tail = 1;
static int g_wino4_k16 = 1;
static int g_wino4_tapmod = 0;   // TIMING EXPERIMENT ONLY (da_wino_debug_tapmod): F(4,3) taps read modulo this many output
                                 // channels, so that the tap tensor fits one XCD's L2 -- results are then wrong by design

// ---------------------------------------------------------------------------------------------
// Weight gradient of the same convolution, Winograd form.  With dm = A dy = (dy0, dy0 + dy1, dy0 - dy1, -dy1) per
// output pair and D = B^T d as in the forward,
//     M_j[n][c] = sum over pairs  dm_j[n] * D_j[c]                 (4 contractions over PAIRS instead of 3 over positions)
//     dW0 = M0 + (M1 + M2)/2     dW1 = (M1 - M2)/2     dW2 = (M1 + M2)/2 + M3
// Block = 64 co x 64 ci, 4 waves of 32 x 32 with the four M_j accumulators (v_mfma_f32_32x32x2_f32, K = pairs);
// K step = WW_KP pairs: dY and X staged in pair-indexed even / odd panels ([pair][channel], as they sit in HBM),
// transforms at fragment-read time; sequence edges (d0 of a first pair, d3 of a last pair) from two 32-bit masks
// per K step.  The combination to dW happens on the accumulators, so the slabs have the direct kernel's layout
// [split][3][N][C] and share its reduction.
// ---------------------------------------------------------------------------------------------
struct WinoWgradArgs {
  const float* dy;
  const float* x;
  float* slab;
  int MP, L, PL, lddy, N, ldx, C, pchunk;
  FastDiv divPL;
};

#ifndef WW_KP
#define WW_KP 32          // pairs per K step (32 or 16)
#endif
#ifndef WW_MIN_WAVES
#define WW_MIN_WAVES 4    // waves per SIMD the register allocation must allow
#define WW_UNROLL 8
#endif
#define WW_LDS_FLOATS ((2 * WW_KP + 2 * (WW_KP + 2)) * 64)

__device__ __forceinline__ void wino_wgrad_body(const WinoWgradArgs& a, const int block_id, const int nblocks, float* lds) {
  constexpr int KP = WW_KP, NRB = KP / 16;
  float* YE = lds;                     // [KP][64]   dY at the even position of pairs k0 .. k0+KP-1
  float* YO = lds + KP * 64;           // [KP][64]   odd position (0 past the sequence end)
  float* XE = lds + 2 * KP * 64;       // [KP+2][64] X even, pairs k0-1 .. k0+KP
  float* XO = XE + (KP + 2) * 64;      // [KP+2][64] X odd

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntc = a.C >> 6, tiles = (a.N >> 6) * ntc;
  const int lin = xcd_chunked(block_id, nblocks);      // tile fastest: one pair chunk's readers share an XCD
  const int bx = lin % tiles, split = lin / tiles;
  const int n_blk = (bx / ntc) * 64, c_blk = (bx % ntc) * 64;
  const int PL = a.PL;
  const int k_beg = split * a.pchunk, k_end = min(a.MP, k_beg + a.pchunk);

  const int lrow = tid >> 4, lq = tid & 15;             // loader: 16 pair rows x 16 channel quads per pass
  f32x4 ry[2 * NRB], rx[2 * NRB + 1];
  auto gload = [&](int k0) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const int P = k0 + lrow + 16 * rb;
      const bool ok = P < a.MP;
      const uint32_t r = fdiv((uint32_t)(ok ? P : 0), a.divPL);
      const int i = (ok ? P : 0) - (int)r * PL;
      const size_t pos = (size_t)r * a.L + 2 * i;
      const bool ok1 = ok && 2 * i + 1 < a.L;
      const bool okk = ok && P < k_end;                 // dY only inside this split's pairs; X neighbours beyond it are real
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      ry[2 * rb] = okk ? *reinterpret_cast<const f32x4*>(a.dy + pos * a.lddy + n_blk + lq * 4) : z;
      ry[2 * rb + 1] = (okk && ok1) ? *reinterpret_cast<const f32x4*>(a.dy + (pos + 1) * a.lddy + n_blk + lq * 4) : z;
      rx[2 * rb] = ok ? *reinterpret_cast<const f32x4*>(a.x + pos * a.ldx + c_blk + lq * 4) : z;
      rx[2 * rb + 1] = ok1 ? *reinterpret_cast<const f32x4*>(a.x + (pos + 1) * a.ldx + c_blk + lq * 4) : z;
    }
    if (tid < 32) {                                     // halo: odd of pair k0-1 (tid < 16), even of pair k0+KP
      const int P = tid < 16 ? k0 - 1 : k0 + KP;
      const bool ok = P >= 0 && P < a.MP;
      const uint32_t r = fdiv((uint32_t)(ok ? P : 0), a.divPL);
      const int i = (ok ? P : 0) - (int)r * PL;
      const int pp = 2 * i + (tid < 16 ? 1 : 0);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok && pp < a.L) v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)r * a.L + pp) * a.ldx + c_blk + lq * 4);
      rx[2 * NRB] = v;
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int frow = lane & 31, fh = lane >> 5;
  if (k_beg < k_end) gload(k_beg);
  for (int k0 = k_beg; k0 < k_end; k0 += KP) {
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const int row = lrow + 16 * rb;
      *reinterpret_cast<f32x4*>(&YE[row * 64 + lq * 4]) = ry[2 * rb];
      *reinterpret_cast<f32x4*>(&YO[row * 64 + lq * 4]) = ry[2 * rb + 1];
      *reinterpret_cast<f32x4*>(&XE[(row + 1) * 64 + lq * 4]) = rx[2 * rb];
      *reinterpret_cast<f32x4*>(&XO[(row + 1) * 64 + lq * 4]) = rx[2 * rb + 1];
    }
    if (tid < 16) *reinterpret_cast<f32x4*>(&XO[lq * 4]) = rx[2 * NRB];
    else if (tid < 32) *reinterpret_cast<f32x4*>(&XE[(KP + 1) * 64 + lq * 4]) = rx[2 * NRB];
    // sequence-edge masks of this step's pairs (bit p: pair k0+p is the first / last of its sequence)
    const int Pm = k0 + (lane & 31);
    const int im = Pm - (int)fdiv((uint32_t)(Pm < a.MP ? Pm : 0), a.divPL) * PL;
    const uint32_t fmask = (uint32_t)__ballot(Pm < a.MP && im == 0);
    const uint32_t lmask = (uint32_t)__ballot(Pm < a.MP && im == PL - 1);
    __syncthreads();
#pragma unroll WW_UNROLL
    for (int kk = 0; kk < KP / 2; ++kk) {
      if (kk == KP / 4) {
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + KP < k_end) gload(k0 + KP);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int p = 2 * kk + fh;
      const float y0 = YE[p * 64 + wm * 32 + frow], y1 = YO[p * 64 + wm * 32 + frow];
      float d0 = XO[p * 64 + wn * 32 + frow];
      const float d1 = XE[(p + 1) * 64 + wn * 32 + frow], d2 = XO[(p + 1) * 64 + wn * 32 + frow];
      float d3 = XE[(p + 2) * 64 + wn * 32 + frow];
      if ((fmask >> p) & 1) d0 = 0.f;
      if ((lmask >> p) & 1) d3 = 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, d0 - d2, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0 + y1, d1 + d2, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0 - y1, d2 - d1, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(y1, d1 - d3, acc[3], 0, 0, 0);     // = -M3
    }
  }

  float* out = a.slab + (size_t)split * 3 * a.N * a.C;
  const size_t plane = (size_t)a.N * a.C;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n_blk + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
    const int c = c_blk + wn * 32 + frow;
    const float h = 0.5f * (acc[1][r] + acc[2][r]);
    float* o = out + (size_t)n * a.C + c;
    o[0] = acc[0][r] + h;
    o[plane] = 0.5f * (acc[1][r] - acc[2][r]);
    o[2 * plane] = h - acc[3][r];
  }
}

struct WinoWgradTable {
  WinoWgradArgs d[24];
  int first_block[25];
  int n;
};
static_assert(sizeof(WinoWgradTable) + sizeof(WgradPreTable) <= 4096, "kernel arguments: 4 KB");

// pre: the slab reductions of the previous weight-gradient launch's jobs, as this launch's first blocks (common.h)
__global__ __launch_bounds__(256, WW_MIN_WAVES) void wino_wgrad_multi_kernel(WinoWgradTable t, WgradPreTable pre) {
  __shared__ float lds[WW_LDS_FLOATS];
  const int b = wgrad_pre_dispatch(pre);
  if (b < 0 || b >= t.first_block[t.n]) return;
  int i = 0;
  while (i + 1 < t.n && b >= t.first_block[i + 1]) ++i;      // wave-uniform
  wino_wgrad_body(t.d[i], b - t.first_block[i], t.first_block[i + 1] - t.first_block[i], lds);
}

bool wino_wgrad_eligible(const da_wgrad_job& j) {
  return j.ntaps == 3 && j.dy_stride == 1 && j.dy_off == 0 && j.src_stride == 1 && j.src_off[0] == -1 &&
         j.src_off[1] == 0 && j.src_off[2] == 1 && j.Lm == j.Ldy && j.Lm == j.Lx && j.N % 64 == 0 && j.C % 64 == 0 &&
         j.N >= 64 && j.C >= 64;
}

#ifndef WW_PCHUNK
#define WW_PCHUNK 640      // (whole fp32 step at B = 64, ms, rows = pairs per split 416 / 512 / 640 / 768 / 1024 with 320 quads: 2.927 / 2.944 / 2.912 / 2.955 / 2.970)
#endif
static int g_ww_pchunk = WW_PCHUNK;

// pairs per split: every block of every job carries the same work (20 K steps), slab traffic 48 KB per block.  f = 2: half the
// pairs per split (the launcher's choice for the jobs of the launch's last, partly filled round -- see wino_wgrad_launch)
void wino_wgrad_plan(int rows, int L, int* splits, int* pchunk, int f) {
  const int MP = rows * ((L + 1) / 2);
  const int base = g_ww_pchunk / (f > 1 ? f : 1) >= 32 ? g_ww_pchunk / (f > 1 ? f : 1) : 32;
  int sp = (MP + base - 1) / base;
  if (sp < 1) sp = 1;
  int pc = ((MP + sp - 1) / sp + 31) / 32 * 32;
  if (pc < 32) pc = 32;
  *splits = (MP + pc - 1) / pc > 0 ? (MP + pc - 1) / pc : 1;
  *pchunk = pc;
}

// factors != NULL (the caller can be told how many slabs a job wrote: da_conv_wgrad_multi_reduce): the jobs whose blocks make
// up the launch's last, partly filled round of 1 024 slots (4 blocks a CU) run with HALF the pairs per split -- twice the
// blocks of half the length there: at B = 64 the step's 1 232 blocks are 1 008 (layer 3, layer 2) + 224 (layer 1), and with
// 448 half blocks behind the 1 008 the launch ends 13 us earlier (a quarter: 10 us; measured on the whole step).
// factors[i] = the divisor job i ran with (1 or 2); the workspace of a job must hold twice da_conv_wgrad_plan's slabs.
int wino_wgrad_launch(const da_wgrad_job* jobs, int n, hipStream_t s, WgradChain* chain, int* factors) {
  WinoWgradTable t;
  int cnt = 0, blocks = 0;
  auto flush = [&]() -> int {
    if (!cnt) return DA_OK;
    t.n = cnt;
    t.first_block[cnt] = blocks;
    WgradPreTable pre = wgrad_chain_take(chain);
    hipLaunchKernelGGL(wino_wgrad_multi_kernel, dim3(wgrad_pre_grid(pre, blocks)), dim3(256), 0, s, t, pre);
    DA_CHECK_LAUNCH();
    cnt = 0;
    blocks = 0;
    return DA_OK;
  };
  constexpr int SLOTS = 1024;
  long total = 0;
  int njobs = 0;
  for (int i = 0; i < n; ++i) {
    if (jobs[i].winograd != 1) continue;
    int splits, pchunk;
    wino_wgrad_plan(jobs[i].rows, jobs[i].Lm, &splits, &pchunk, 1);
    total += (long)(jobs[i].N / 64) * (jobs[i].C / 64) * splits;
    ++njobs;
  }
  // blocks in whole rounds (a launch of less than one round, or of more than one table, stays as planned)
  const long full = factors && njobs <= 24 && total > SLOTS && total % SLOTS ? total / SLOTS * SLOTS : total;
  long cum = 0;
  for (int i = 0; i < n; ++i) {
    const da_wgrad_job& j = jobs[i];
    if (j.winograd != 1) continue;
    int splits, pchunk;
    wino_wgrad_plan(j.rows, j.Lm, &splits, &pchunk, 1);
    cum += (long)(j.N / 64) * (j.C / 64) * splits;
    const int f = cum > full ? 2 : 1;
    if (f > 1) wino_wgrad_plan(j.rows, j.Lm, &splits, &pchunk, f);
    if (factors) factors[i] = f;
    WinoWgradArgs& a = t.d[cnt];
    a.dy = j.dy; a.x = j.x; a.slab = j.workspace;
    a.L = j.Lm; a.PL = (j.Lm + 1) / 2; a.MP = j.rows * a.PL;
    a.lddy = j.lddy; a.N = j.N; a.ldx = j.ldx; a.C = j.C; a.pchunk = pchunk;
    a.divPL = make_fastdiv((uint32_t)a.PL);
    t.first_block[cnt] = blocks;
    blocks += (j.N / 64) * (j.C / 64) * splits;
    wgrad_chain_offer(chain, i, j, splits);
    if (++cnt == 24) {
      int rc = flush();
      if (rc) return rc;
    }
  }
  return flush();
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient in the F(4,3) form (job flag winograd == 6; the 512-channel convs, like the forward's
// conv3_wino4k_kernel): K runs over output QUADS.  With dy0..dy3 the quad's outputs, d0..d5 = x[4q-1 .. 4q+4] and
//   dm = A dy = (dy0, dy0+dy1+dy2+dy3, dy0-dy1+dy2-dy3, dy0+2dy1+4dy2+8dy3, dy0-2dy1+4dy2-8dy3, dy3),   D = B^T d (as forward)
//   M_j[n][c] = sum over quads dm_j[n] D_j[c]              (6 contractions over quads: 3/4 of F(2,3)'s 8 over pairs)
//   dW0 = M0/4 - (M1+M2)/6 + (M3+M4)/24    dW1 = (M2-M1)/6 + (M3-M4)/12    dW2 = -(M1+M2)/6 + (M3+M4)/6 + M5      (= G^T M)
// Block = 64 co x 64 ci, 4 waves of 32 x 32 with six accumulators; K step = 16 quads: dY in four phase panels [quad][channel],
// X in four ([quad - 1 ..] for phase 3, [.. quad + 1] for phase 0: d0 = X3[q-1], d1..d4 = X0..X3[q], d5 = X0[q+1]); sequence
// edges (d0 of a row's first quad, d5 of its last) from two 16-bit masks per K step, positions past L are zeros of the
// loader.  Slabs in the direct kernel's layout, shared reduction.
// ---------------------------------------------------------------------------------------------
#ifndef WW4_KQ
#define WW4_KQ 16         // quads per K step (16 or 32)
#endif
#ifndef WW4_MIN_WAVES
#define WW4_MIN_WAVES 3   // waves per SIMD the register allocation must allow
#endif
#ifndef WW4_UNROLL
#define WW4_UNROLL 4      // (8: 39 spilled registers at 3 waves/SIMD, 346 us for the three 512-channel jobs at B = 64 against 280)
#endif
#define WW4_LDS_FLOATS ((4 * WW4_KQ + 4 * WW4_KQ + 2) * 64)

__device__ __forceinline__ void wino4_wgrad_body(const WinoWgradArgs& a, const int block_id, const int nblocks, float* lds) {
  constexpr int KQ = WW4_KQ, NRB = KQ / 16;
  float* Y = lds;                         // [4 phases][KQ][64]
  float* X0 = lds + 4 * KQ * 64;          // [KQ + 1][64]: quads k0 .. k0 + KQ
  float* X1 = X0 + (KQ + 1) * 64;         // [KQ][64]
  float* X2 = X1 + KQ * 64;               // [KQ][64]
  float* X3 = X2 + KQ * 64;               // [KQ + 1][64]: quads k0 - 1 .. k0 + KQ - 1

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntc = a.C >> 6, tiles = (a.N >> 6) * ntc;
  const int lin = xcd_chunked(block_id, nblocks);
  const int bx = lin % tiles, split = lin / tiles;
  const int n_blk = (bx / ntc) * 64, c_blk = (bx % ntc) * 64;
  const int QL = a.PL;                    // quads per row (the args' PL / MP / divPL count quads here)
  const int k_beg = split * a.pchunk, k_end = min(a.MP, k_beg + a.pchunk);

  const int lrow = tid >> 4, lq = tid & 15;             // loader: 16 quads x 16 channel quads per pass
  f32x4 ry[4 * NRB], rx[4 * NRB + 1];
  auto gload = [&](int k0) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const int Q = k0 + lrow + 16 * rb;
      const bool ok = Q < a.MP;
      const uint32_t r = fdiv((uint32_t)(ok ? Q : 0), a.divPL);
      const int i = (ok ? Q : 0) - (int)r * QL;
      const size_t pos = (size_t)r * a.L + 4 * i;
      const bool okk = ok && Q < k_end;                 // dY only inside this split's quads; X neighbours beyond it are real
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const bool in = 4 * i + ph < a.L;
        ry[4 * rb + ph] = (okk && in) ? *reinterpret_cast<const f32x4*>(a.dy + (pos + ph) * a.lddy + n_blk + lq * 4) : z;
        rx[4 * rb + ph] = (ok && in) ? *reinterpret_cast<const f32x4*>(a.x + (pos + ph) * a.ldx + c_blk + lq * 4) : z;
      }
    }
    if (tid < 32) {                                     // halo: phase 3 of quad k0 - 1 (tid < 16), phase 0 of quad k0 + KQ
      const int Qh = tid < 16 ? k0 - 1 : k0 + KQ;
      const bool okh = Qh >= 0 && Qh < a.MP;
      const uint32_t rh = fdiv((uint32_t)(okh ? Qh : 0), a.divPL);
      const int ih = (okh ? Qh : 0) - (int)rh * QL;
      const int pp = 4 * ih + (tid < 16 ? 3 : 0);
      f32x4 v = z;
      if (okh && pp < a.L) v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)rh * a.L + pp) * a.ldx + c_blk + lq * 4);
      rx[4 * NRB] = v;
    }
  };

  f32x16 acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int frow = lane & 31, fh = lane >> 5;
  if (k_beg < k_end) gload(k_beg);
  for (int k0 = k_beg; k0 < k_end; k0 += KQ) {
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const int row = lrow + 16 * rb;
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) *reinterpret_cast<f32x4*>(&Y[(ph * KQ + row) * 64 + lq * 4]) = ry[4 * rb + ph];
      *reinterpret_cast<f32x4*>(&X0[row * 64 + lq * 4]) = rx[4 * rb];
      *reinterpret_cast<f32x4*>(&X1[row * 64 + lq * 4]) = rx[4 * rb + 1];
      *reinterpret_cast<f32x4*>(&X2[row * 64 + lq * 4]) = rx[4 * rb + 2];
      *reinterpret_cast<f32x4*>(&X3[(row + 1) * 64 + lq * 4]) = rx[4 * rb + 3];
    }
    if (tid < 16) *reinterpret_cast<f32x4*>(&X3[lq * 4]) = rx[4 * NRB];
    else if (tid < 32) *reinterpret_cast<f32x4*>(&X0[KQ * 64 + lq * 4]) = rx[4 * NRB];
    // sequence-edge masks of this step's quads (bit p: quad k0 + p is the first / last of its sequence)
    const int Qm = k0 + (lane & (KQ - 1));
    const int im = Qm - (int)fdiv((uint32_t)(Qm < a.MP ? Qm : 0), a.divPL) * QL;
    const uint32_t fmask = (uint32_t)__ballot(Qm < a.MP && im == 0);          // (bits >= KQ repeat the low ones: never read)
    const uint32_t lmask = (uint32_t)__ballot(Qm < a.MP && im == QL - 1);
    __syncthreads();
#pragma unroll WW4_UNROLL
    for (int kk = 0; kk < KQ / 2; ++kk) {
      if (kk == KQ / 4) {
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + KQ < k_end) gload(k0 + KQ);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int p = 2 * kk + fh;
      const int yo = p * 64 + wm * 32 + frow, xo = p * 64 + wn * 32 + frow;
      const float y0 = Y[yo], y1 = Y[KQ * 64 + yo], y2 = Y[2 * KQ * 64 + yo], y3 = Y[3 * KQ * 64 + yo];
      float d0 = X3[xo];
      const float d1 = X0[xo], d2 = X1[xo], d3 = X2[xo], d4 = X3[xo + 64];
      float d5 = X0[xo + 64];
      if ((fmask >> p) & 1) d0 = 0.f;
      if ((lmask >> p) & 1) d5 = 0.f;
      const float ya = y0 + y2, yb = y1 + y3, yc = fmaf(4.f, y2, y0), yd = 2.f * fmaf(4.f, y3, y1);
      const float e1 = fmaf(-4.f, d2, d4), e2 = fmaf(-4.f, d1, d3), f1 = d4 - d2, f2 = d3 - d1;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, fmaf(4.f, d0, fmaf(-5.f, d2, d4)), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya + yb, e1 + e2, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya - yb, e1 - e2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(yc + yd, fmaf(2.f, f2, f1), acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(yc - yd, fmaf(-2.f, f2, f1), acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(y3, fmaf(4.f, d1, fmaf(-5.f, d3, d5)), acc[5], 0, 0, 0);
    }
  }

  float* out = a.slab + (size_t)split * 3 * a.N * a.C;
  const size_t plane = (size_t)a.N * a.C;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n_blk + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
    const int c = c_blk + wn * 32 + frow;
    const float s12 = acc[1][r] + acc[2][r], s34 = acc[3][r] + acc[4][r];
    float* o = out + (size_t)n * a.C + c;
    o[0] = fmaf(0.25f, acc[0][r], fmaf(1.0f / 24.0f, s34, -(1.0f / 6.0f) * s12));
    o[plane] = fmaf(1.0f / 6.0f, acc[2][r] - acc[1][r], (1.0f / 12.0f) * (acc[3][r] - acc[4][r]));
    o[2 * plane] = fmaf(1.0f / 6.0f, s34 - s12, acc[5][r]);
  }
}

__global__ __launch_bounds__(256, WW4_MIN_WAVES) void wino4_wgrad_multi_kernel(WinoWgradTable t, WgradPreTable pre) {
  __shared__ float lds[WW4_LDS_FLOATS];
  const int b = wgrad_pre_dispatch(pre);
  if (b < 0 || b >= t.first_block[t.n]) return;
  int i = 0;
  while (i + 1 < t.n && b >= t.first_block[i + 1]) ++i;      // wave-uniform
  wino4_wgrad_body(t.d[i], b - t.first_block[i], t.first_block[i + 1] - t.first_block[i], lds);
}

#ifndef WW4_QCHUNK
#define WW4_QCHUNK 320     // (quads per split 192 / 256 / 320 / 384 / 448 with 640 pairs: 2.959 / 2.934 / 2.912 / 2.990 (768 pairs) / 2.987: block counts that fill whole rounds)
#endif
static int g_ww4_qchunk = WW4_QCHUNK;

// quads per split (a multiple of the K step)
void wino4_wgrad_plan(int rows, int L, int* splits, int* qchunk) {
  const int MQ = rows * ((L + 3) / 4);
  int sp = (MQ + g_ww4_qchunk - 1) / g_ww4_qchunk;
  if (sp < 1) sp = 1;
  int qc = ((MQ + sp - 1) / sp + WW4_KQ - 1) / WW4_KQ * WW4_KQ;
  if (qc < WW4_KQ) qc = WW4_KQ;
  *splits = (MQ + qc - 1) / qc > 0 ? (MQ + qc - 1) / qc : 1;
  *qchunk = qc;
}

int wino4_wgrad_launch(const da_wgrad_job* jobs, int n, hipStream_t s, WgradChain* chain) {
  WinoWgradTable t;
  int cnt = 0, blocks = 0;
  auto flush = [&]() -> int {
    if (!cnt) return DA_OK;
    t.n = cnt;
    t.first_block[cnt] = blocks;
    WgradPreTable pre = wgrad_chain_take(chain);
    hipLaunchKernelGGL(wino4_wgrad_multi_kernel, dim3(wgrad_pre_grid(pre, blocks)), dim3(256), 0, s, t, pre);
    DA_CHECK_LAUNCH();
    cnt = 0;
    blocks = 0;
    return DA_OK;
  };
  for (int i = 0; i < n; ++i) {
    const da_wgrad_job& j = jobs[i];
    if (j.winograd != 6) continue;
    int splits, qchunk;
    wino4_wgrad_plan(j.rows, j.Lm, &splits, &qchunk);
    WinoWgradArgs& a = t.d[cnt];
    a.dy = j.dy; a.x = j.x; a.slab = j.workspace;
    a.L = j.Lm; a.PL = (j.Lm + 3) / 4; a.MP = j.rows * a.PL;
    a.lddy = j.lddy; a.N = j.N; a.ldx = j.ldx; a.C = j.C; a.pchunk = qchunk;
    a.divPL = make_fastdiv((uint32_t)a.PL);
    t.first_block[cnt] = blocks;
    blocks += (j.N / 64) * (j.C / 64) * splits;
    wgrad_chain_offer(chain, i, j, splits);
    if (++cnt == 24) {
      int rc = flush();
      if (rc) return rc;
    }
  }
  return flush();
}

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

// U[6][N][C] for F(4,3), same conventions.
__global__ __launch_bounds__(256) void wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int co,
                                                           int ci, int transpose) {
  const int N = transpose ? ci : co, C = transpose ? co : ci;
  const size_t total = (size_t)N * C;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx / C), c = (int)(idx - (size_t)n * C);
  const float* src = transpose ? w + ((size_t)c * ci + n) * 3 : w + ((size_t)n * ci + c) * 3;
  const float g0 = transpose ? src[2] : src[0], g1 = src[1], g2 = transpose ? src[0] : src[2];
  wino4_taps(g0, g1, g2, u + idx, total);
}

extern "C" {

// F(4,3) variant of da_conv3_winograd; u = 6 * N * C floats from da_wino4_weights.
int da_conv3_winograd4(const float* x, const float* u, float* y, int rows, int L, int ldx, int C, int ldy, int N,
                       int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (!x || !u || !y || rows < 0 || L < 1 || C % 32 || N % 32 || C < 32 || N < 32 || ldx % 4 || ldx < C || ldy < N)
    return DA_EINVAL;
  if (rows == 0) return DA_OK;
  if ((uint64_t)rows * L * (uint64_t)(ldx > ldy ? ldx : ldy) >= 0x7fffffffull) return DA_EINVAL;
  WinoArgs a;
  a.x = x; a.u = u; a.y = y;
  a.L = L; a.PL = (L + 3) / 4; a.MP = rows * a.PL;        // PL / MP count quads here
  a.ldx = ldx; a.C = C; a.ldy = ldy; a.N = N; a.accumulate = accumulate;
  a.drop_seed = nullptr; a.drop_salt = 0u; a.drop_p = 0.f;
  a.stat_part = nullptr; a.stat_Wu = 1;
  a.in_pend = nullptr; a.in_mean = a.in_invstd = nullptr; a.in_gamma = a.in_beta = nullptr; a.in_tiles = 0; a.in_Wu = 1; a.in_eps = 0.f;
  a.tapmod = g_wino4_tapmod;
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
  if (g_wino4_k16)
    hipLaunchKernelGGL(conv3_wino4k_kernel, dim3(nmini_pad + full), dim3(256), 0, stream, a, nmini, nmini_pad, full);
  else
    hipLaunchKernelGGL(conv3_wino4_kernel, dim3(nmini_pad + full), dim3(256), 0, stream, a, nmini, nmini_pad, full);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_wino4_weights(const float* w, float* u, int co, int ci, int transpose, hipStream_t stream) {
  DA_ENTER();
  if (!w || !u || co < 1 || ci < 1) return DA_EINVAL;
  const size_t total = (size_t)co * ci;
  hipLaunchKernelGGL(wino4_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, u, co, ci,
                     transpose);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// y (+)= conv1d(x, k = 3, stride 1, pad 1) per row with the transformed taps u (da_wino_weights).
// x: [rows][L][ldx] first C channels; y: [rows][L][ldy] first N channels.  replaces reference models/resnet.py:5-8
struct WinoBnIn {       // the XFW operands of conv3_winograd_impl (in_pend == NULL: a plain input)
  const float* pend; float* mean; float* invstd; const float* gamma; const float* beta; float eps;
};

static int conv3_winograd_impl(const float* x, const float* u, float* y, int rows, int L, int ldx, int C, int ldy, int N,
                               int accumulate, const long long* drop_seed, unsigned drop_salt, float drop_p, float* stat_part,
                               int stat_R, hipStream_t stream, const WinoBnIn* bn = nullptr) {
  DA_ENTER();
  if (g_act_bf16) return DA_EINVAL;              // float activations only
  if (!x || !u || !y || rows < 0 || L < 1 || C % 32 || N % 32 || C < 32 || N < 32 || ldx % 4 || ldx < C || ldy < N)
    return DA_EINVAL;
  if (rows == 0) return DA_OK;
  if ((uint64_t)rows * L * (uint64_t)(ldx > ldy ? ldx : ldy) >= 0x7fffffffull) return DA_EINVAL;   // 32-bit element offsets
  WinoArgs a;
  a.x = x; a.u = u; a.y = y;
  a.L = L; a.PL = (L + 1) / 2; a.MP = rows * a.PL;
  a.ldx = ldx; a.C = C; a.ldy = ldy; a.N = N; a.accumulate = accumulate;
  a.drop_seed = drop_seed; a.drop_salt = drop_salt; a.drop_p = drop_p;
  a.stat_part = stat_part; a.stat_Wu = (stat_part || bn) ? stat_R * a.PL : 1;
  if ((stat_part || bn) && (stat_R < 1 || rows % stat_R || a.stat_Wu < 64 || accumulate)) return DA_EINVAL;
  a.in_pend = nullptr; a.in_mean = a.in_invstd = nullptr; a.in_gamma = a.in_beta = nullptr; a.in_tiles = 0; a.in_Wu = 1; a.in_eps = 0.f;
  a.tapmod = 0;
  if (bn) {
    if (!bn->pend || !bn->mean || !bn->invstd || !bn->gamma || !bn->beta || C > 128 || ldx != C) return DA_EINVAL;
    a.in_pend = bn->pend; a.in_mean = bn->mean; a.in_invstd = bn->invstd; a.in_gamma = bn->gamma; a.in_beta = bn->beta;
    a.in_eps = bn->eps; a.in_Wu = stat_R * L; a.in_tiles = (int)(((long)rows * L + 63) / 64);
    if (a.in_Wu < 64) return DA_EINVAL;
  }
  a.divPL = make_fastdiv((uint32_t)a.PL);
  if ((uint64_t)a.MP * (uint64_t)a.PL >= 0xffffffffull) return DA_EINVAL;
  const int tiles = ((a.MP + 63) / 64) * (N / 32);
  const int R = tiles % 256;
  int nmini = 0, full = tiles;
  const bool dropping = drop_p > 0.f;
  if (g_wino_tail && !stat_part && !bn && !dropping && tiles > 256 && R >= 1 && R <= 128) {
    nmini = 2 * R;
    full = tiles - R;
  }
  const int nmini_pad = (nmini + 7) / 8 * 8;
  if (bn && stat_part) hipLaunchKernelGGL(conv3_wino_bn_kernel<true>, dim3(full), dim3(256), 0, stream, a, full);
  else if (bn) hipLaunchKernelGGL(conv3_wino_bn_kernel<false>, dim3(full), dim3(256), 0, stream, a, full);
  else if (stat_part) hipLaunchKernelGGL(conv3_wino_stats_kernel, dim3(full), dim3(256), 0, stream, a, full);
  else if (dropping) hipLaunchKernelGGL(conv3_wino_drop_kernel, dim3(full), dim3(256), 0, stream, a, full);
  else hipLaunchKernelGGL(conv3_wino_kernel, dim3(nmini_pad + full), dim3(256), 0, stream, a, nmini, nmini_pad, full);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_conv3_winograd(const float* x, const float* u, float* y, int rows, int L, int ldx, int C, int ldy, int N,
                      int accumulate, hipStream_t stream) {
  return conv3_winograd_impl(x, u, y, rows, L, ldx, C, ldy, N, accumulate, nullptr, 0u, 0.f, nullptr, 0, stream);
}

// da_conv3_winograd followed by F.dropout(p) in the epilogue: y = dropout(conv(x)) with the keep mask of da_dropout
// (seed, salt) on the contiguous [rows * L][N] tensor, written at pitch ldy (a _DenseLayer's growth conv + dropout storing
// its new features at their channel offset in the block's buffer, reference models/densenet.py:30-40)
// stat_part != NULL: also the statistics records of the output for windows of R rows (da_stat_records_floats(rows * ceil(L / 2),
// N) floats; R * ceil(L / 2) >= 64): what the consuming da_conv1x1_bn merges instead of a statistics pass over the new channels
int da_conv3_winograd_drop(const float* x, const float* u, float* y, int rows, int L, int ldx, int C, int ldy, int N,
                           const long long* drop_seed, unsigned drop_salt, float drop_p, float* stat_part, int R,
                           hipStream_t stream) {
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed)) return DA_EINVAL;
  return conv3_winograd_impl(x, u, y, rows, L, ldx, C, ldy, N, 0, drop_seed, drop_salt, drop_p, stat_part, R, stream);
}

// da_conv3_winograd_drop on the OUTPUT x of the 1x1 conv in front, with relu(norm2(x)) applied while x is staged
// (densenet.py:27-32 norm2 -> relu2 -> conv2; the activation is never stored): x [rows][L][C] contiguous, C <= 128; the
// statistics of x come as the records `in_pend` that conv's epilogue wrote (da_conv1x1_bn out_part: rows * L positions in
// windows of R * L) and are PUBLISHED to in_mean / in_invstd [rows / R][C] for the backward kernels.
int da_conv3_winograd_bn(const float* x, const float* u, float* y, int rows, int L, int C, int ldy, int N, int R,
                         const float* in_pend, float* in_mean, float* in_invstd, const float* gamma, const float* beta,
                         float eps, const long long* drop_seed, unsigned drop_salt, float drop_p, float* stat_part,
                         hipStream_t stream) {
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed)) return DA_EINVAL;
  WinoBnIn bn = {in_pend, in_mean, in_invstd, gamma, beta, eps};
  return conv3_winograd_impl(x, u, y, rows, L, C, C, ldy, N, 0, drop_seed, drop_salt, drop_p, stat_part, R, stream, &bn);
}

// floats of a statistics-record buffer for `units` record units (64 per tile) of N channels:
// [tiles][2 window slots][{mean, M2}][N] followed by the counts [tiles][2]
size_t da_stat_records_floats(long units, int N) {
  const size_t tiles = (size_t)((units + 63) / 64);
  return tiles * 4 * (size_t)N + tiles * 2;
}

// tuning / tests: 0 = no half tiles for the last round; pchunk > 0: pairs per weight-gradient split
int da_wino_debug_tail(int on) {
  if (on & ~1) {                 // 2 / 3: F(4,3) K step 32 / 16
    g_wino4_k16 = on & 1;
    return DA_OK;
  }
  g_wino_tail = on;
  return DA_OK;
}
// TIMING EXPERIMENT ONLY: the F(4,3) kernel reads its transformed taps modulo `mod` output channels (0: off) -- the tap
// tensor then fits one XCD's L2 and the launch shows what its re-reads through the fabric cost; results are wrong by design
int da_wino_debug_tapmod(int mod) {
  if (mod < 0 || mod % 32) return DA_EINVAL;
  g_wino4_tapmod = mod;
  return DA_OK;
}
int da_wino_debug_pchunk(int pchunk) {
  if (pchunk > 0) g_ww_pchunk = pchunk;
  if (pchunk < 0) bf16_wgrad_set_pchunk(-pchunk);          // negative: padded positions per split of the bf16 kernels
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
