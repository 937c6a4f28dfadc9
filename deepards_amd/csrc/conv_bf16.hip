// k3 stride-1 pad-1 Conv1d forward / data gradient with bf16 matrix arithmetic (BASELINE config C3: "resnet18-1D ...
// bf16"): replaces the same nn.Conv1d calls as conv_wino.hip (reference models/resnet.py:5-8,27-38) when the host
// selects dtype bf16.  Activations stay fp32 in HBM; they are rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32) while being staged into LDS, the weights come pre-packed in bf16 (da_pack_conv3_bf16), products
// run on v_mfma_f32_32x32x16_bf16 and accumulate in fp32; the output is fp32.  Direct 3-tap form (no Winograd: at the
// bf16 matrix rate the kernel is bound by the LDS / load path, not by multiplies).
//
// GEMM rows are flat positions P = row * L + l, so a tile's input panel is the contiguous range P0-1 .. P0+TM of
// positions; a tap that would cross a sequence edge (l = 0 with tap 0, l = L-1 with tap 2) is zeroed on the A
// fragment.  Block = 128 positions x 64 output channels, 4 waves of 64 x 32 (two 32x32 accumulators); K step = 32
// channels: X panel [130][32] and W panel [3][64][32] in bf16 with an 80-byte row pitch (5 16-byte slots: 16 rows of
// one k group, in the lane groups ds_read_b128 is served in, fall on 16 different slots mod 16).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// block id -> work item so that CONSECUTIVE work items share an XCD (blocks are dealt to the 8 XCDs round-robin): the
// channel tiles of one position tile, which read the same activation panel, then hit the same L2 (conv_gemm.hip)
__device__ __forceinline__ int xcd_chunked_bf(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, s = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
}

struct ConvBf16Args {
  const void* x;        // [M][ldx] activations (float or bf16: template AT), first C channels
  const __bf16* w;      // [3][N][C] bf16 taps
  void* y;              // [M][ldy] activations, first N channels
  int M, L, ldx, C, ldy, N, accumulate;
  FastDiv divL;
  // the BatchNorm in front of / behind the conv without a pass of its own (da_conv3_bf16_bn; resnet.py:27-33 conv1 -> bn1 ->
  // relu -> conv2): windows of Wn positions.
  //   stat_part != NULL: the statistics records (include/deepards_hip.h; units = positions, two records per 128-position tile)
  //     of the STORED output, written by the epilogue;
  //   in_pend != NULL: the input is a raw conv output whose records are in_pend -- every block merges the records of its
  //     tile's (<= 2) windows into scale / shift tables, the tile holding a window's first position publishes (mean, invstd),
  //     and relu(fma(x, sc, sh)) is what gets staged
  int Wn;
  float* stat_part;
  const float* in_pend;
  float* in_mean;
  float* in_invstd;
  const float* in_gamma;
  const float* in_beta;
  int in_tiles;
  float in_eps;
};

#define CB_TM 128
#define CB_TN 64
#define CB_PITCH 80                                   // bytes per LDS row (64 data + 16 pad)
#define CB_XROWS (CB_TM + 2)
#define CB_LDS_BYTES ((CB_XROWS + 3 * CB_TN) * CB_PITCH)

__device__ __forceinline__ f32x2v cvt4_bf16(const f32x4& v) {       // 4 bf16 (nearest-even) as the bits of 2 floats
  const f32x2v lo = {v[0], v[1]}, hi = {v[2], v[3]};
  const bf16x2 a = __builtin_convertvector(lo, bf16x2), b = __builtin_convertvector(hi, bf16x2);
  return f32x2v{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
}

// Four channels of one position on their way global memory -> registers -> LDS image (bf16 bits).  float activations
// are rounded while being staged (v_cvt_pk_bf16_f32); bf16 activations (da_set_act_dtype(1)) are already the image's
// format: 8-byte loads, no conversion -- half the bytes through the load path.
template <typename AT> struct Stage;
template <> struct Stage<float> {
  typedef f32x4 reg;
  static __device__ __forceinline__ reg zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ reg ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ f32x2v bits(const reg& v) { return cvt4_bf16(v); }
  static __device__ __forceinline__ f32x4 wide(const reg& v) { return v; }
};
template <> struct Stage<__bf16> {
  typedef f32x2v reg;
  static __device__ __forceinline__ reg zero() { return f32x2v{0.f, 0.f}; }
  static __device__ __forceinline__ reg ld(const __bf16* p) { return *reinterpret_cast<const f32x2v*>(p); }
  static __device__ __forceinline__ f32x2v bits(const reg& v) { return v; }
  static __device__ __forceinline__ f32x4 wide(const reg& v) { return X3::widen4(v); }
};

// activations in the x3 format (common.h): the three bf16 terms of 4 channels are 3 x 8 bytes, 32 bytes apart
struct X3T {};
template <> struct Stage<X3T> {
  struct reg { f32x2v h, m, l; };
  static __device__ __forceinline__ reg zero() { return reg{f32x2v{0.f, 0.f}, f32x2v{0.f, 0.f}, f32x2v{0.f, 0.f}}; }
};
// four channels (c0 .. c0 + 3, c0 % 4 == 0) of position `pos` of a [positions][ld] tensor of storage type AT (ld =
// channel pitch in elements; x3: the tensor has `ld` channels and 3 ld bf16 per position)
template <typename AT>
__device__ __forceinline__ typename Stage<AT>::reg stage_ld(const void* base, size_t pos, int ld, int c0) {
  if constexpr (__is_same(AT, X3T)) {
    const __bf16* d = reinterpret_cast<const __bf16*>(base) + pos * (size_t)(3 * ld) + X3::off(c0);
    return typename Stage<X3T>::reg{*reinterpret_cast<const f32x2v*>(d), *reinterpret_cast<const f32x2v*>(d + 16),
                                    *reinterpret_cast<const f32x2v*>(d + 32)};
  } else {
    return Stage<AT>::ld(reinterpret_cast<const AT*>(base) + pos * (size_t)ld + c0);
  }
}

template <typename AT, bool STATS, bool XF>
__device__ __forceinline__ void conv3_bf16_body(const ConvBf16Args& a, unsigned char* lds, float* xtab) {
  const AT* ax = reinterpret_cast<const AT*>(a.x);
  AT* ay = reinterpret_cast<AT*>(a.y);
  unsigned char* Xs = lds;                            // [130][80 B]: positions P0-1 .. P0+128
  unsigned char* Ws = lds + CB_XROWS * CB_PITCH;      // [3][64][80 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = a.N / CB_TN;
  const int tile = xcd_chunked_bf(blockIdx.x, gridDim.x);
  const int P0 = (tile / ntn) * CB_TM, n_blk = (tile % ntn) * CB_TN;

  // X loader: 32 rows x 8 channel quads per pass; lanes 8..15 of every 16 take the row 4 below lanes 0..7 so that a
  // ds_write_b64 group (16 lanes on 32 banks) covers 16 different 8-byte slots
  const int xm = tid >> 4, xq = tid & 7;
  const int xrow = (xm >> 2) * 8 + (xm & 3) + 4 * ((tid >> 3) & 1);
  constexpr int NXP = (CB_XROWS + 31) / 32;           // 5 passes, the last one 2 rows
  long xoff[NXP];
  bool xok[NXP];
  int xslot[NXP];                                     // XF: this row's scale / shift table (window w0 or w0 + 1), + the quad
  const int w0 = XF ? P0 / a.Wn : 0;
#pragma unroll
  for (int p = 0; p < NXP; ++p) {
    const int r = p * 32 + xrow;
    const long P = (long)P0 - 1 + r;
    xok[p] = r < CB_XROWS && P >= 0 && P < a.M;
    xoff[p] = (xok[p] ? P : 0) * a.ldx + xq * 4;
    xslot[p] = (XF && P >= (long)(w0 + 1) * a.Wn ? 2 * a.C : 0) + xq * 4;
  }
  // W loader: 64 rows x 4 slots of 8 channels per pass (rows 4 apart per 8 lanes: ds_write_b128), 3 passes = 3 taps
  const int wm_ = tid >> 3, ws = tid & 3;
  const int wrow = (wm_ >> 2) * 8 + (wm_ & 3) + 4 * ((tid >> 2) & 1);
  const __bf16* wsrc = a.w + (size_t)(n_blk + wrow) * a.C + ws * 8;
  const size_t wtap = (size_t)a.N * a.C;

  typename Stage<AT>::reg rx[NXP];
  f32x4 rw[3];                                        // 8 bf16 each, as bits
  auto gload = [&](int ks) {
    const int c0 = ks << 5;
#pragma unroll
    for (int t = 0; t < 3; ++t) rw[t] = *reinterpret_cast<const f32x4*>(wsrc + t * wtap + c0);
#pragma unroll
    for (int p = 0; p < NXP; ++p) {
      typename Stage<AT>::reg v = Stage<AT>::zero();
      if (xok[p]) v = Stage<AT>::ld(ax + xoff[p] + c0);
      rx[p] = v;
    }
  };

  // fragment geometry (32x32x16: lane = (row l%32, k group l/32 of 8 channels))
  const int frow = lane & 31, kg = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  bool at_first[2], at_last[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const long P = (long)P0 + wm * 64 + mt * 32 + frow;
    const uint32_t Pc = (uint32_t)(P < a.M ? P : 0);
    const int l = (int)(Pc - fdiv(Pc, a.divL) * (uint32_t)a.L);
    at_first[mt] = l == 0;
    at_last[mt] = l == a.L - 1;
  }
  const unsigned char* xfrag = Xs + (wm * 64 + frow) * CB_PITCH + kg * 16;
  const unsigned char* wfrag = Ws + (wn * 32 + frow) * CB_PITCH + kg * 16;

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  const int kc = a.C >> 5;
  gload(0);
  if (XF) {   // the scale / shift vectors of this tile's two windows, from the records (first operand loads in flight)
    const int nwin = a.M / a.Wn;
    for (int i = tid; i < 2 * a.C; i += 256) {
      const int ws_ = i >= a.C ? 1 : 0, c = i - ws_ * a.C, w = w0 + ws_;
      float sc = 0.f, sh = 0.f;
      if (w < nwin) {
        float mu, is;
        merge_stat_records<8>(a.in_pend, a.in_tiles, a.C, a.Wn, w, c, a.in_eps, mu, is);
        const long wP = (long)w * a.Wn;               // the window's first position: its tile publishes
        if (n_blk == 0 && wP >= P0 && wP < P0 + CB_TM) {
          a.in_mean[(size_t)w * a.C + c] = mu;
          a.in_invstd[(size_t)w * a.C + c] = is;
        }
        bn_scale_shift(mu, is, a.in_gamma[c], a.in_beta[c], sc, sh);
      }
      xtab[(ws_ * 2) * a.C + c] = sc;
      xtab[(ws_ * 2 + 1) * a.C + c] = sh;
    }
  }
  auto staged = [&](int p, int ks) -> f32x2v {            // the bf16 bits loader slot p puts into the image
    if (!XF) return Stage<AT>::bits(rx[p]);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(&xtab[xslot[p] + (ks << 5)]);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(&xtab[xslot[p] + a.C + (ks << 5)]);
    const f32x4 v = Stage<AT>::wide(rx[p]);
    f32x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
    return xok[p] ? cvt4_bf16(h) : f32x2v{0.f, 0.f};      // rows outside the tensor stay zeros
  };
  for (int ks = 0; ks < kc; ++ks) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NXP; ++p) {
      const int r = p * 32 + xrow;
      if (r < CB_XROWS) *reinterpret_cast<f32x2v*>(Xs + r * CB_PITCH + xq * 8) = staged(p, ks);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<f32x4*>(Ws + (t * CB_TN + wrow) * CB_PITCH + ws * 16) = rw[t];
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < kc) gload(ks + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(wfrag + t * CB_TN * CB_PITCH + kk * 32);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 av = *reinterpret_cast<const f32x4*>(xfrag + (mt * 32 + t) * CB_PITCH + kk * 32);   // 8 bf16 as bits
          if ((t == 0 && at_first[mt]) || (t == 2 && at_last[mt])) av = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), b, acc[mt], 0, 0, 0);
        }
      }
  }

  // lane holds output channel n_blk + wn*32 + l%32 of the positions (r & 3) + 8 (r >> 2) + 4 (l / 32) of each 32-row tile
  // STATS: the wave's 64 positions x 32 channels are one record tile's share -- per channel (count, mean, centred M2) of the
  // values AS STORED (rounded when the storage is bf16), single pass about a pivot (the chunk's first position), the window
  // that starts inside the chunk in the second slot
  const long Pf = (long)P0 + wm * 64;
  long bound = 0;
  float piv = 0.f, s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  if (STATS) {
    const uint32_t Pfc = (uint32_t)(Pf < a.M ? Pf : 0);
    bound = ((long)(Pfc / (uint32_t)a.Wn) + 1) * a.Wn;
    float v0 = acc[0][0];
    if constexpr (__is_same(AT, __bf16)) v0 = (float)(__bf16)v0;
    piv = __shfl(v0, frow, 64);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long P = Pf + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (P < a.M) {
        AT* o = ay + P * a.ldy + n_blk + wn * 32 + frow;
        float v = acc[mt][r];
        if (a.accumulate) v += Act<AT>::ld1(o);
        Act<AT>::st1(o, v);
        if (STATS) {
          if constexpr (__is_same(AT, __bf16)) v = (float)(__bf16)v;
          const float d = v - piv;
          const int sl = P >= bound ? 1 : 0;
          s1[sl] += d;
          s2[sl] = fmaf(d, d, s2[sl]);
        }
      }
    }
  if (STATS) {
    const long nrt = ((long)a.M + 63) >> 6, rt = Pf >> 6;
    if (rt < nrt) {
      const long lastP = Pf + 63 < a.M ? Pf + 63 : (long)a.M - 1;
      float cnt[2];
      cnt[1] = lastP >= bound ? (float)(lastP - bound + 1) : 0.f;
      cnt[0] = (float)(lastP - Pf + 1) - cnt[1];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        s1[sl] += __shfl_xor(s1[sl], 32, 64);
        s2[sl] += __shfl_xor(s2[sl], 32, 64);
      }
      if (kg == 0) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const float inv = cnt[sl] > 0.f ? 1.f / cnt[sl] : 0.f;
          float* rec = a.stat_part + ((size_t)(rt * 2 + sl) * 2) * a.N + n_blk + wn * 32 + frow;   // [tile][slot][{mean, M2}][N]
          rec[0] = piv + s1[sl] * inv;
          rec[a.N] = fmaxf(s2[sl] - s1[sl] * s1[sl] * inv, 0.f);
        }
        if (frow == 0 && n_blk == 0 && wn == 0) {
          a.stat_part[(size_t)nrt * 4 * a.N + rt * 2] = cnt[0];
          a.stat_part[(size_t)nrt * 4 * a.N + rt * 2 + 1] = cnt[1];
        }
      }
    }
  }
}

template <typename AT>
__global__ __launch_bounds__(256) void conv3_bf16_kernel(ConvBf16Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[CB_LDS_BYTES];
  conv3_bf16_body<AT, false, false>(a, lds, nullptr);
}

// ... with the output's statistics records from the epilogue and / or relu(BatchNorm(x)) applied while x is staged
template <typename AT, bool STATS, bool XF>
__global__ __launch_bounds__(256) void conv3_bf16_bn_kernel(ConvBf16Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[CB_LDS_BYTES];
  extern __shared__ __attribute__((aligned(16))) float xtab_dyn[];     // XF: [2 windows][{sc, sh}][C]
  conv3_bf16_body<AT, STATS, XF>(a, lds, xtab_dyn);
}

// ---------------------------------------------------------------------------------------------
// General form for the stride-2 block heads and 1x1 downsamples (forward and data gradient), the contract of
// da_conv_gemm with bf16 operands:
//     Y[row][j * dst_stride + dst_off][n] (+)= sum_t sum_c X[row][j * SS + src_off[t]][c] * Wp[wtap[t]][n][c],  j in [0, Lm)
// with Lsrc == SS * Lm, so that the source of flat output m = row * Lm + j is the flat position SS * m + src_off[t] and a
// tile's panel is one contiguous range of positions.  SS = 2 keeps even and odd panel rows in two halves: the rows a
// tap needs are then unit-stride in m (conflict-free ds_read_b128), whatever the tap's parity.
// ---------------------------------------------------------------------------------------------
struct ConvBf16GenArgs {
  const void* x;        // activations (template AT)
  const __bf16* w;      // [taps][N][C]
  void* y;
  int M, Lm, Lsrc, ldx, C, Ldst, ldy, N, dst_stride, dst_off, ntaps, accumulate;
  int src_off[3], wtap[3];
  long Msrc;            // rows * Lsrc
  FastDiv divLm;
  // two sources in one contraction (da_conv_job.x2 / w2 / tap_split): the taps >= tap_split read x2 with the weights w2 (same
  // pitch, channel count and geometry) -- the even input positions of a stride-2 block entry's data gradient take the conv's
  // tap 1 from dy1 AND the downsample's tap from dyd (resnet.py:36-38 backward); the K loop runs over x's channels for the
  // first taps, then over x2's for the others.  x2 == NULL: one source, tap_split = ntaps.
  const void* x2;
  const __bf16* w2;
  int tap_split;
};

template <int SS, typename AT>
__device__ __forceinline__ void conv_bf16_gen_body(const ConvBf16GenArgs& a, const int tile, unsigned char* lds) {
  const AT* ax = reinterpret_cast<const AT*>(a.x);
  AT* ay = reinterpret_cast<AT*>(a.y);
  constexpr int HROWS = CB_TM + 1;                               // rows per parity half (SS = 2)
  constexpr int NR = SS * (CB_TM - 1) + 3;                       // panel rows at most (span of the taps <= 2)
  constexpr int XBYTES = (SS == 2 ? 2 * HROWS : NR) * CB_PITCH;
  unsigned char* Xs = lds;
  unsigned char* Ws = lds + XBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = a.N / CB_TN;
  const int M0 = (tile / ntn) * CB_TM, n_blk = (tile % ntn) * CB_TN;
  int smin = a.src_off[0];
#pragma unroll
  for (int t = 1; t < 3; ++t)
    if (t < a.ntaps && a.src_off[t] < smin) smin = a.src_off[t];
  const long p0 = (long)SS * M0 + smin;                          // flat source position of panel row 0

  // X loader: 32 rows x 8 channel quads per pass; the two 8-lane halves of a ds_write_b64 group take rows 4 SS apart
  const int xm = tid >> 4, xq = tid & 7, xsub = (tid >> 3) & 1;
  const int xrow = SS == 1 ? (xm >> 2) * 8 + (xm & 3) + 4 * xsub : (xm >> 3) * 16 + (xm & 7) + 8 * xsub;
  constexpr int NXP = (NR + 31) / 32;
  long xoff[NXP];
  bool xok[NXP];
#pragma unroll
  for (int p = 0; p < NXP; ++p) {
    const int r = p * 32 + xrow;
    const long P = p0 + r;
    xok[p] = r < NR && P >= 0 && P < a.Msrc;
    xoff[p] = (xok[p] ? P : 0) * a.ldx + xq * 4;
  }
  const int wm_ = tid >> 3, ws = tid & 3;
  const int wrow = (wm_ >> 2) * 8 + (wm_ & 3) + 4 * ((tid >> 2) & 1);
  const size_t wtapsz = (size_t)a.N * a.C;
  const __bf16* wsrc = a.w + (size_t)(n_blk + wrow) * a.C + ws * 8;
  const AT* ax2 = reinterpret_cast<const AT*>(a.x2);
  const __bf16* w2src = a.w2 ? a.w2 + (size_t)(n_blk + wrow) * a.C + ws * 8 : wsrc;
  const int kc = a.C >> 5, ksteps = a.x2 ? 2 * kc : kc;          // second phase: the taps >= tap_split over x2's channels
  const int tsplit = a.x2 ? a.tap_split : a.ntaps;

  typename Stage<AT>::reg rx[NXP];
  f32x4 rw[3];
  auto gload = [&](int ks) {
    const bool second = ks >= kc;
    const int c0 = (second ? ks - kc : ks) << 5;
    const AT* xs = second ? ax2 : ax;
    const __bf16* wb = second ? w2src : wsrc;
    const int t0 = second ? tsplit : 0, t1 = second ? a.ntaps : tsplit;
#pragma unroll
    for (int t = 0; t < 3; ++t)
      if (t >= t0 && t < t1) rw[t] = *reinterpret_cast<const f32x4*>(wb + a.wtap[t] * wtapsz + c0);
#pragma unroll
    for (int p = 0; p < NXP; ++p) {
      typename Stage<AT>::reg v = Stage<AT>::zero();
      if (xok[p]) v = Stage<AT>::ld(xs + xoff[p] + c0);
      rx[p] = v;
    }
  };

  const int frow = lane & 31, kg = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  bool ok[2][3];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const long m = (long)M0 + wm * 64 + mt * 32 + frow;
    const uint32_t mc = (uint32_t)(m < a.M ? m : 0);
    const int j = (int)(mc - fdiv(mc, a.divLm) * (uint32_t)a.Lm);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int src = j * SS + a.src_off[t < a.ntaps ? t : 0];
      ok[mt][t] = m < a.M && src >= 0 && src < a.Lsrc;
    }
  }
  int tapoff[3];                                                 // LDS byte offset of tap t's row for output row 0
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int d = a.src_off[t < a.ntaps ? t : 0] - smin;
    tapoff[t] = SS == 1 ? d * CB_PITCH : (d & 1) * HROWS * CB_PITCH + (d >> 1) * CB_PITCH;
  }
  const unsigned char* xfrag = Xs + (wm * 64 + frow) * CB_PITCH + kg * 16;
  const unsigned char* wfrag = Ws + (wn * 32 + frow) * CB_PITCH + kg * 16;

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  gload(0);
  for (int ks = 0; ks < ksteps; ++ks) {
    const int t0 = ks >= kc ? tsplit : 0, t1 = ks >= kc ? a.ntaps : tsplit;      // this step's taps
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NXP; ++p) {
      const int r = p * 32 + xrow;
      if (r < NR) {
        const int off = SS == 1 ? r * CB_PITCH : (r & 1) * HROWS * CB_PITCH + (r >> 1) * CB_PITCH;
        *reinterpret_cast<f32x2v*>(Xs + off + xq * 8) = Stage<AT>::bits(rx[p]);
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
      if (t >= t0 && t < t1) *reinterpret_cast<f32x4*>(Ws + (t * CB_TN + wrow) * CB_PITCH + ws * 16) = rw[t];
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < ksteps) gload(ks + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t < t0 || t >= t1) continue;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(wfrag + t * CB_TN * CB_PITCH + kk * 32);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 av = *reinterpret_cast<const f32x4*>(xfrag + tapoff[t] + mt * 32 * CB_PITCH + kk * 32);
          if (!ok[mt][t]) av = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), b, acc[mt], 0, 0, 0);
        }
      }
    }
  }

#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long m = (long)M0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (m < a.M) {
        const uint32_t rq = fdiv((uint32_t)m, a.divLm);
        const int j = (int)((uint32_t)m - rq * (uint32_t)a.Lm);
        AT* o = ay + ((size_t)rq * a.Ldst + (size_t)j * a.dst_stride + a.dst_off) * a.ldy + n_blk + wn * 32 + frow;
        float v = acc[mt][r];
        if (a.accumulate) v += Act<AT>::ld1(o);
        Act<AT>::st1(o, v);
      }
    }
}

// up to 4 independent problems of one source stride in a launch (a block head's conv and its downsample read the same
// input; the even and the odd positions of a stride-2 data gradient are disjoint outputs): these launches are
// latency-bound, so sharing one costs the longest of them, not the sum
struct ConvBf16GenTable {
  ConvBf16GenArgs d[4];
  int first_block[5];
  int n;
};

template <int SS, typename AT>
__global__ __launch_bounds__(256) void conv_bf16_gen_kernel(ConvBf16GenTable t) {
  constexpr int XBYTES = (SS == 2 ? 2 * (CB_TM + 1) : SS * (CB_TM - 1) + 3) * CB_PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char lds[XBYTES + 3 * CB_TN * CB_PITCH];
  int i = 0;
  while (i + 1 < t.n && (int)blockIdx.x >= t.first_block[i + 1]) ++i;      // wave-uniform
  conv_bf16_gen_body<SS, AT>(t.d[i], xcd_chunked_bf(blockIdx.x - t.first_block[i], t.first_block[i + 1] - t.first_block[i]), lds);
}

// wf[t][co][ci] = bf16(w[co][ci][t]) (forward taps), wd[t][ci][co] = bf16(w[co][ci][2 - t]) (data-gradient taps)
__global__ __launch_bounds__(256) void pack_conv3_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wf,
                                                              __bf16* __restrict__ wd, int co, int ci) {
  const size_t total = (size_t)co * ci;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int o = (int)(idx / ci), i = (int)(idx - (size_t)o * ci);
  const float* src = w + idx * 3;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const __bf16 v = (__bf16)src[t];
    if (wf) wf[(size_t)t * total + idx] = v;
    if (wd) wd[(size_t)(2 - t) * total + (size_t)i * co + o] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of the same convolution with bf16 operands:  dW[t][n][c] = sum_P dy[P][n] * x[P + t - 1][c]  (fp32
// sums).  Both operands are contracted over POSITIONS but sit position-major in HBM, so the LDS images keep that layout
// ([position][channel] bf16, filled with plain 8-byte stores) and the fragments come out of ds_read_b64_tr_b16, the
// transposing read: per 16 lanes it takes a block of 4 positions x 16 channels and hands lane i channel i's 4 positions.
// Sequence edges need no masks: K runs over PADDED positions P' = row * (L + 1) + l whose l = L slot is a zero row in
// both images, so x[P' - 1] of a first and x[P' + 1] of a last position are zeros (1 / (L + 1) more K steps).
// Block = 64 n x 64 c, 4 waves of 32 x 32 with one accumulator per tap; K step = 64 padded positions; split over
// chunks of padded positions into slabs [split][3][N][C] = the direct kernel's layout, shared reduction.
// Image rows are 192 bytes (128 data + 64 pad): the 4 rows of a transposed read fall on disjoint bank ranges.
// ---------------------------------------------------------------------------------------------
struct WgradBf16Args {
  const void* dy;       // activations (template AT)
  const void* x;
  float* slab;
  int rows, L, Kpad, lddy, N, ldx, C, pchunk;
  FastDiv divL1;        // by L + 1
  int mode;             // 0: k3 s1 p1;  1: k3 s2 p1;  2: k1 s2 p0  (L = OUTPUT length; the input is 2 L long for 1 / 2)
  FastDiv divLx2;       // modes 1 / 2: by 2 L + 2
};

// NS = 1: operands rounded to bf16 (conv dtype 'bf16').  NS = 3: fp32-equivalent products ("f32x3", conv_x3.hip): every
// operand is split exactly into three bf16 terms kept in three channel planes of the image row, and a product is the six
// MFMA products h h' + h m' + m h' + h l' + l h' + m m' (the dropped ones are below one fp32 rounding).  The K step is
// 64 positions for NS = 1 and 32 for NS = 3 (three times the image bytes per position).
template <int NS> struct WB {
  static constexpr int KP = NS == 1 ? 64 : 32;
  static constexpr int PITCH = NS * 128 + 64;            // 192 / 448: = 192 mod 256, the 4 rows of a transposed read fall on disjoint bank ranges
  static constexpr int XPITCH2 = NS * 128 + 32;          // 160 / 416: stride-2 X image, rows 2 apart per transposed read
  static constexpr int X2ROWS = 2 * KP + 3;
  static constexpr int LDS_BYTES = KP * PITCH + ((KP + 2) * PITCH > X2ROWS * XPITCH2 ? (KP + 2) * PITCH : X2ROWS * XPITCH2);
};

// the NS bf16 terms of 4 channels (as bits), written to the planes of an image row at d
template <typename AT, int NS> struct WBStage;
template <typename AT> struct WBStage<AT, 1> {
  static __device__ __forceinline__ void put(unsigned char* d, const typename Stage<AT>::reg& v) {
    *reinterpret_cast<f32x2v*>(d) = Stage<AT>::bits(v);
  }
};
template <> struct WBStage<X3T, 3> {                       // pre-split operands: three plain 8-byte stores, no arithmetic
  static __device__ __forceinline__ void put(unsigned char* d, const Stage<X3T>::reg& v) {
    *reinterpret_cast<f32x2v*>(d) = v.h;
    *reinterpret_cast<f32x2v*>(d + 128) = v.m;
    *reinterpret_cast<f32x2v*>(d + 256) = v.l;
  }
};

// acc += A B over the NS x NS split terms that matter (small terms first)
template <int NS>
__device__ __forceinline__ void wb_mfma(f32x16& acc, const f32x4* a, const f32x4* b) {
#define WB_M(i, j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc, 0, 0, 0)
  if (NS == 3) {
    WB_M(2, 0); WB_M(0, 2); WB_M(1, 1); WB_M(1, 0); WB_M(0, 1);
  }
  WB_M(0, 0);
#undef WB_M
}

__device__ __forceinline__ f32x2v ds_read_tr16(const unsigned char* p) {
  f32x2v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
  return v;
}

template <typename AT, int NS>
__device__ __forceinline__ void wgrad_bf16_body(const WgradBf16Args& a, const int block_id, unsigned char* lds) {
  constexpr int KP = WB<NS>::KP, PITCH = WB<NS>::PITCH, NP = KP / 16;
  unsigned char* Ys = lds;                              // [KP][PITCH]      dY at padded positions k0 .. k0+KP-1
  unsigned char* Xs = lds + KP * PITCH;                 // [KP + 2][PITCH]  X at padded positions k0-1 .. k0+KP

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntc = a.C >> 6, tiles = (a.N >> 6) * ntc;
  const int bx = block_id % tiles, split = block_id / tiles;
  const int n_blk = (bx / ntc) * 64, c_blk = (bx % ntc) * 64;
  const int k_beg = split * a.pchunk, k_end = min(a.Kpad, k_beg + a.pchunk);
  const int L1 = a.L + 1;

  const int lq = tid & 15, lr = tid >> 4;               // loader: 16 rows x 16 channel quads per pass
  typename Stage<AT>::reg ry[NP], rx[NP + 1];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int Pp = k0 + lr + 16 * p;                  // padded position of dY row
      bool ok = Pp < k_end;
      const uint32_t sq = fdiv((uint32_t)(ok ? Pp : 0), a.divL1);
      const int l = (ok ? Pp : 0) - (int)sq * L1;
      ok = ok && l < a.L;
      typename Stage<AT>::reg v = Stage<AT>::zero();
      if (ok) v = stage_ld<AT>(a.dy, (size_t)sq * a.L + l, a.lddy, n_blk + lq * 4);
      ry[p] = v;
    }
#pragma unroll
    for (int p = 0; p < NP + 1; ++p) {
      const int r = p < NP ? lr + 16 * p : KP + lr;     // image row; rows KP, KP+1 by the first 32 threads
      const int Pp = k0 - 1 + r;
      bool ok = Pp >= 0 && Pp < a.Kpad && (p < NP || tid < 32);
      const uint32_t sq = fdiv((uint32_t)(ok ? Pp : 0), a.divL1);
      const int l = (ok ? Pp : 0) - (int)sq * L1;
      ok = ok && l < a.L;
      typename Stage<AT>::reg v = Stage<AT>::zero();
      if (ok) v = stage_ld<AT>(a.x, (size_t)sq * a.L + l, a.ldx, c_blk + lq * 4);
      rx[p] = v;
    }
  };

  // transposed-read geometry: group g = lane / 16 serves channels 16 (g & 1) .. + 15 at positions + 8 (g >> 1);
  // lane 4q + p of the group supplies the address of position row q, channels 4p .. 4p + 3
  const int wn = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const unsigned char* yfrag = Ys + (8 * (g >> 1) + tq) * PITCH + (wn * 32 + 16 * (g & 1) + 4 * tp) * 2;
  const unsigned char* xfrag = Xs + (8 * (g >> 1) + tq) * PITCH + (wc * 32 + 16 * (g & 1) + 4 * tp) * 2;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (k_beg < k_end) gload(k_beg);
  for (int k0 = k_beg; k0 < k_end; k0 += KP) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      WBStage<AT, NS>::put(Ys + (lr + 16 * p) * PITCH + lq * 8, ry[p]);
      WBStage<AT, NS>::put(Xs + (lr + 16 * p) * PITCH + lq * 8, rx[p]);
    }
    if (tid < 32) WBStage<AT, NS>::put(Xs + (KP + lr) * PITCH + lq * 8, rx[NP]);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (k0 + KP < k_end) gload(k0 + KP);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < (NS == 1 ? KP / 16 : 0); ++kk) {
      const unsigned char* yp = yfrag + kk * 16 * PITCH;
      const unsigned char* xp = xfrag + kk * 16 * PITCH;
      {
        f32x2v y0 = ds_read_tr16(yp), y1 = ds_read_tr16(yp + 4 * PITCH);
        f32x2v b0[3], b1[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {                   // X image row j + t holds the position dY row j meets at tap t
          b0[t] = ds_read_tr16(xp + t * PITCH);
          b1[t] = ds_read_tr16(xp + (t + 4) * PITCH);
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(y0), "+v"(y1), "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[1]), "+v"(b1[1]), "+v"(b0[2]), "+v"(b1[2]));
        const f32x4 av = {y0[0], y0[1], y1[0], y1[1]};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const f32x4 bv = {b0[t][0], b0[t][1], b1[t][0], b1[t][1]};
          wb_mfma<1>(acc[t], &av, &bv);
        }
      }
    }
    if constexpr (NS > 1) {
#pragma unroll
      for (int kk = 0; kk < KP / 16; ++kk) {
        const unsigned char* yp = yfrag + kk * 16 * PITCH;
        const unsigned char* xp = xfrag + kk * 16 * PITCH;
        f32x2v y0[NS], y1[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          y0[s] = ds_read_tr16(yp + s * 128);
          y1[s] = ds_read_tr16(yp + 4 * PITCH + s * 128);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          f32x2v b0[NS], b1[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            b0[s] = ds_read_tr16(xp + t * PITCH + s * 128);
            b1[s] = ds_read_tr16(xp + (t + 4) * PITCH + s * 128);
          }
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(y0[0]), "+v"(y1[0]), "+v"(y0[NS / 2]), "+v"(y1[NS / 2]), "+v"(y0[NS - 1]), "+v"(y1[NS - 1]),
                         "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[NS / 2]), "+v"(b1[NS / 2]), "+v"(b0[NS - 1]), "+v"(b1[NS - 1]));
          f32x4 av[NS], bv[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            av[s] = f32x4{y0[s][0], y0[s][1], y1[s][0], y1[s][1]};
            bv[s] = f32x4{b0[s][0], b0[s][1], b1[s][0], b1[s][1]};
          }
          wb_mfma<NS>(acc[t], av, bv);
        }
      }
    }
  }

  float* out = a.slab + (size_t)split * 3 * a.N * a.C;
  const size_t plane = (size_t)a.N * a.C;
  const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n_blk + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      out[t * plane + (size_t)n * a.C + c_blk + wc * 32 + frow] = acc[t][r];
    }
}

// Stride-2 convs (k3 s2 p1 block heads, NTAPS = 3; k1 s2 downsamples, NTAPS = 1):
//     dW[t][n][c] = sum_j dy[j][n] * x[2 j + t - 1][c]        (k1: the t = 1 term alone)
// K runs over padded OUTPUT positions P' = row (Lo + 1) + j; with the input padded to Lx + 2 = 2 (Lo + 1) slots a row
// (slot 0 and slot Lx + 1 zero) the input slot of (P', t) is Q' = 2 P' + t -- linear, so the X image of a K step is the
// contiguous range Q' = 2 k0 .. 2 k0 + 130 and a transposed read takes rows 2 apart (160-byte rows keep its 4 rows on
// disjoint banks).
template <int NTAPS, typename AT, int NS>
__device__ __forceinline__ void wgrad_bf16_s2_body(const WgradBf16Args& a, const int block_id, unsigned char* lds) {
  constexpr int KP = WB<NS>::KP, PITCH = WB<NS>::PITCH, XPITCH2 = WB<NS>::XPITCH2, X2ROWS = WB<NS>::X2ROWS, NP = KP / 16;
  unsigned char* Ys = lds;                              // [KP][PITCH]
  unsigned char* Xs = lds + KP * PITCH;                 // [2 KP + 3][XPITCH2]: input slots 2 k0 .. 2 k0 + 2 KP + 2

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntc = a.C >> 6, tiles = (a.N >> 6) * ntc;
  const int bx = block_id % tiles, split = block_id / tiles;
  const int n_blk = (bx / ntc) * 64, c_blk = (bx % ntc) * 64;
  const int k_beg = split * a.pchunk, k_end = min(a.Kpad, k_beg + a.pchunk);
  const int L1 = a.L + 1, Lx = 2 * a.L, Lx2 = Lx + 2;

  const int lq = tid & 15, lr = tid >> 4;
  constexpr int NXP = (X2ROWS + 15) / 16;               // passes of 16 rows, the last one 3 rows
  typename Stage<AT>::reg ry[NP], rx[NXP];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int Pp = k0 + lr + 16 * p;
      bool ok = Pp < k_end;
      const uint32_t sq = fdiv((uint32_t)(ok ? Pp : 0), a.divL1);
      const int j = (ok ? Pp : 0) - (int)sq * L1;
      ok = ok && j < a.L;
      typename Stage<AT>::reg v = Stage<AT>::zero();
      if (ok) v = stage_ld<AT>(a.dy, (size_t)sq * a.L + j, a.lddy, n_blk + lq * 4);
      ry[p] = v;
    }
#pragma unroll
    for (int p = 0; p < NXP; ++p) {
      const int r = lr + 16 * p;
      const long Qp = 2l * k0 + r;                      // padded input slot
      bool ok = r < X2ROWS && Qp < 2l * a.Kpad;
      const uint32_t sq = fdiv((uint32_t)(ok ? Qp : 0), a.divLx2);
      const int sl = (int)((ok ? Qp : 0) - (long)sq * Lx2);
      ok = ok && sl >= 1 && sl <= Lx;
      typename Stage<AT>::reg v = Stage<AT>::zero();
      if (ok) v = stage_ld<AT>(a.x, (size_t)sq * Lx + (sl - 1), a.ldx, c_blk + lq * 4);
      rx[p] = v;
    }
  };

  const int wn = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const unsigned char* yfrag = Ys + (8 * (g >> 1) + tq) * PITCH + (wn * 32 + 16 * (g & 1) + 4 * tp) * 2;
  const unsigned char* xfrag = Xs + 2 * (8 * (g >> 1) + tq) * XPITCH2 + (wc * 32 + 16 * (g & 1) + 4 * tp) * 2;

  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (k_beg < k_end) gload(k_beg);
  for (int k0 = k_beg; k0 < k_end; k0 += KP) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) WBStage<AT, NS>::put(Ys + (lr + 16 * p) * PITCH + lq * 8, ry[p]);
#pragma unroll
    for (int p = 0; p < NXP; ++p) {
      const int r = lr + 16 * p;
      if (r < X2ROWS) WBStage<AT, NS>::put(Xs + r * XPITCH2 + lq * 8, rx[p]);
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (k0 + KP < k_end) gload(k0 + KP);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < (NS == 1 ? KP / 16 : 0); ++kk) {
      const unsigned char* yp = yfrag + kk * 16 * PITCH;
      const unsigned char* xp = xfrag + kk * 32 * XPITCH2;
      {
        f32x2v y0 = ds_read_tr16(yp), y1 = ds_read_tr16(yp + 4 * PITCH);
        f32x2v b0[NTAPS], b1[NTAPS];
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {               // input slot of (output row j, tap t) = 2 j + t (k1: t = 1)
          const int tt = NTAPS == 1 ? 1 : t;
          b0[t] = ds_read_tr16(xp + tt * XPITCH2);
          b1[t] = ds_read_tr16(xp + (tt + 8) * XPITCH2);
        }
        if (NTAPS == 3)
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(y0), "+v"(y1), "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[NTAPS - 1]), "+v"(b1[NTAPS - 1]),
                         "+v"(b0[NTAPS / 2]), "+v"(b1[NTAPS / 2]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y0), "+v"(y1), "+v"(b0[0]), "+v"(b1[0]));
        const f32x4 av = {y0[0], y0[1], y1[0], y1[1]};
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
          const f32x4 bv = {b0[t][0], b0[t][1], b1[t][0], b1[t][1]};
          wb_mfma<1>(acc[t], &av, &bv);
        }
      }
    }
    if constexpr (NS > 1) {
#pragma unroll
      for (int kk = 0; kk < KP / 16; ++kk) {
        const unsigned char* yp = yfrag + kk * 16 * PITCH;
        const unsigned char* xp = xfrag + kk * 32 * XPITCH2;
        f32x2v y0[NS], y1[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          y0[s] = ds_read_tr16(yp + s * 128);
          y1[s] = ds_read_tr16(yp + 4 * PITCH + s * 128);
        }
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
          const int tt = NTAPS == 1 ? 1 : t;       // input slot of (output row j, tap t) = 2 j + t (k1: t = 1)
          f32x2v b0[NS], b1[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            b0[s] = ds_read_tr16(xp + tt * XPITCH2 + s * 128);
            b1[s] = ds_read_tr16(xp + (tt + 8) * XPITCH2 + s * 128);
          }
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(y0[0]), "+v"(y1[0]), "+v"(y0[NS / 2]), "+v"(y1[NS / 2]), "+v"(y0[NS - 1]), "+v"(y1[NS - 1]),
                         "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[NS / 2]), "+v"(b1[NS / 2]), "+v"(b0[NS - 1]), "+v"(b1[NS - 1]));
          f32x4 av[NS], bv[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            av[s] = f32x4{y0[s][0], y0[s][1], y1[s][0], y1[s][1]};
            bv[s] = f32x4{b0[s][0], b0[s][1], b1[s][0], b1[s][1]};
          }
          wb_mfma<NS>(acc[t], av, bv);
        }
      }
    }
  }

  float* out = a.slab + (size_t)split * NTAPS * a.N * a.C;
  const size_t plane = (size_t)a.N * a.C;
  const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n_blk + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      out[t * plane + (size_t)n * a.C + c_blk + wc * 32 + frow] = acc[t][r];
    }
}

struct WgradBf16Table {
  WgradBf16Args d[24];
  int first_block[25];
  int n;
};

template <typename AT, int NS>
__global__ __launch_bounds__(256) void wgrad_bf16_multi_kernel(WgradBf16Table t) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[WB<NS>::LDS_BYTES];
  int i = 0;
  while (i + 1 < t.n && (int)blockIdx.x >= t.first_block[i + 1]) ++i;      // wave-uniform
  const int b = xcd_chunked_bf(blockIdx.x - t.first_block[i], t.first_block[i + 1] - t.first_block[i]);   // a split's tiles share an XCD
  if (t.d[i].mode == 0) wgrad_bf16_body<AT, NS>(t.d[i], b, lds);
  else if (t.d[i].mode == 1) wgrad_bf16_s2_body<3, AT, NS>(t.d[i], b, lds);
  else wgrad_bf16_s2_body<1, AT, NS>(t.d[i], b, lds);
}

// the jobs on x3 operands (conv arithmetic 'f32x3p', job code 49: k3 s1 p1, and the k3 s2 p1 / k1 s2 p0 jobs of the
// stride-2 block entries): the bodies above with plain copies for staging
__global__ __launch_bounds__(256) void wgrad_x3p_multi_kernel(WgradBf16Table t) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[WB<3>::LDS_BYTES];
  int i = 0;
  while (i + 1 < t.n && (int)blockIdx.x >= t.first_block[i + 1]) ++i;      // wave-uniform
  const int b = xcd_chunked_bf(blockIdx.x - t.first_block[i], t.first_block[i + 1] - t.first_block[i]);
  if (t.d[i].mode == 0) wgrad_bf16_body<X3T, 3>(t.d[i], b, lds);
  else if (t.d[i].mode == 1) wgrad_bf16_s2_body<3, X3T, 3>(t.d[i], b, lds);
  else wgrad_bf16_s2_body<1, X3T, 3>(t.d[i], b, lds);
}

// jobs the bf16 kernels take: channel counts multiples of 64 and  k3 s1 p1 | k3 s2 p1 | k1 s2 p0 (even input length)
bool bf16_wgrad_eligible(const da_wgrad_job& j) {
  if (j.N % 64 || j.C % 64 || j.N < 64 || j.C < 64 || j.dy_stride != 1 || j.dy_off != 0 || j.Lm != j.Ldy) return false;
  if (j.src_stride == 1) return j.ntaps == 3 && j.src_off[0] == -1 && j.src_off[1] == 0 && j.src_off[2] == 1 && j.Lx == j.Lm;
  if (j.src_stride != 2 || j.Lx != 2 * j.Lm) return false;
  if (j.ntaps == 3) return j.src_off[0] == -1 && j.src_off[1] == 0 && j.src_off[2] == 1;
  return j.ntaps == 1 && j.src_off[0] == 0;
}

// (whole bf16 step at B = 64, ms: 1 536 / 1 792 / 2 048 / 2 112 / 2 176 / 2 240 / 2 304 / 2 432 / 2 560 / 3 072 padded positions per
// split: 1.475 / 1.458 / 1.460 / 1.464 / 1.447 / 1.450 / 1.451 / 1.458 / 1.465 / 1.480 -- 160 registers = 3 blocks a CU = 768
// slots; at 2 048 the step's 2 368 blocks are three rounds and 64 blocks, at 2 176 they fit three)
static int g_wb_pchunk = 2176;
void bf16_wgrad_set_pchunk(int pchunk) { g_wb_pchunk = pchunk; }

// padded positions per split (a multiple of the K step) and the number of slabs
void bf16_wgrad_plan(int rows, int L, int* splits, int* pchunk) {
  const long K = (long)rows * (L + 1);
  long sp = (K + g_wb_pchunk - 1) / g_wb_pchunk;
  if (sp < 1) sp = 1;
  long pc = ((K + sp - 1) / sp + 63) / 64 * 64;          // a multiple of both K steps (64 / 32 positions)
  if (pc < 64) pc = 64;
  *splits = (int)((K + pc - 1) / pc > 0 ? (K + pc - 1) / pc : 1);
  *pchunk = (int)pc;
}

// jobs flagged `code` (16: bf16 operands; 49: fp32-equivalent split-bf16 products on x3 operands, fp32 activations only)
int bf16_wgrad_launch(const da_wgrad_job* jobs, int n, int code, hipStream_t s) {
  WgradBf16Table t;
  int cnt = 0, blocks = 0;
  auto flush = [&]() -> int {
    if (!cnt) return DA_OK;
    t.n = cnt;
    t.first_block[cnt] = blocks;
    if (code == 49) hipLaunchKernelGGL(wgrad_x3p_multi_kernel, dim3(blocks), dim3(256), 0, s, t);
    else DA_ACT_DISPATCH(hipLaunchKernelGGL((wgrad_bf16_multi_kernel<AT, 1>), dim3(blocks), dim3(256), 0, s, t));
    DA_CHECK_LAUNCH();
    cnt = 0;
    blocks = 0;
    return DA_OK;
  };
  for (int i = 0; i < n; ++i) {
    const da_wgrad_job& j = jobs[i];
    if (j.winograd != code) continue;
    if (code == 49 && g_act_bf16) return DA_EINVAL;      // the split kernels belong to float activations
    int splits, pchunk;
    bf16_wgrad_plan(j.rows, j.Lm, &splits, &pchunk);
    WgradBf16Args& a = t.d[cnt];
    a.dy = j.dy; a.x = j.x; a.slab = j.workspace;
    a.rows = j.rows; a.L = j.Lm; a.Kpad = j.rows * (j.Lm + 1);
    a.lddy = j.lddy; a.N = j.N; a.ldx = j.ldx; a.C = j.C; a.pchunk = pchunk;
    a.divL1 = make_fastdiv((uint32_t)(j.Lm + 1));
    a.mode = j.src_stride == 1 ? 0 : (j.ntaps == 3 ? 1 : 2);
    a.divLx2 = make_fastdiv((uint32_t)(2 * j.Lm + 2));
    t.first_block[cnt] = blocks;
    blocks += (j.N / 64) * (j.C / 64) * splits;
    if (++cnt == 24) {
      int rc = flush();
      if (rc) return rc;
    }
  }
  return flush();
}

extern "C" {

// y (+)= conv1d(x, k = 3, stride 1, pad 1) per row of L positions, bf16 products / fp32 sums.  x: [rows][L][ldx] fp32
// (first C channels), wpk: [3][N][C] bf16 from da_pack_conv3_bf16, y: [rows][L][ldy] fp32 (first N channels).
// C % 32 == 0, N % 64 == 0.  replaces reference models/resnet.py:5-8 (conv2x2) under dtype bf16
int da_conv3_bf16(const void* x, const void* wpk, void* y, int rows, int L, int ldx, int C, int ldy, int N,
                  int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (!x || !wpk || !y || rows < 0 || L < 1 || C % 32 || N % CB_TN || C < 32 || N < CB_TN || ldx % 4 || ldx < C || ldy < N)
    return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const long M = (long)rows * L;
  if (M >= 0x7fffffffl) return DA_EINVAL;
  ConvBf16Args a;
  a.x = x; a.w = reinterpret_cast<const __bf16*>(wpk); a.y = y;
  a.M = (int)M; a.L = L; a.ldx = ldx; a.C = C; a.ldy = ldy; a.N = N; a.accumulate = accumulate;
  a.divL = make_fastdiv((uint32_t)L);
  a.Wn = 1; a.stat_part = nullptr; a.in_pend = nullptr; a.in_mean = a.in_invstd = nullptr; a.in_gamma = a.in_beta = nullptr;
  a.in_tiles = 0; a.in_eps = 0.f;
  const long tiles = ((M + CB_TM - 1) / CB_TM) * (N / CB_TN);
  if (tiles > 0x7fffffffl) return DA_EINVAL;
  DA_ACT_DISPATCH(hipLaunchKernelGGL(conv3_bf16_kernel<AT>, dim3((unsigned)tiles), dim3(256), 0, stream, a));
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// da_conv3_bf16 with a BatchNorm folded into either end (windows of R rows; resnet.py:27-33: conv1 -> bn1 -> relu -> conv2):
//   stat_part != NULL: the statistics records of y (da_stat_records_floats(rows * L, N) floats; units = positions) written by
//     the epilogue -- of the values as stored;
//   in_pend != NULL: x is a raw conv output with records in_pend (rows * L positions, C channels): relu(gamma (x - mean)
//     invstd + beta) is applied while x is staged, (mean, invstd) are published to in_mean / in_invstd [W][C].
// A window needs >= 130 positions (a 128-position tile and its halo then touch two windows at most).  No accumulate.
int da_conv3_bf16_bn(const void* x, const void* wpk, void* y, int rows, int L, int ldx, int C, int ldy, int N, int R,
                     const float* in_pend, float* in_mean, float* in_invstd, const float* gamma, const float* beta, float eps,
                     float* stat_part, hipStream_t stream) {
  DA_ENTER();
  if (!x || !wpk || !y || rows < 0 || L < 1 || C % 32 || N % CB_TN || C < 32 || N < CB_TN || ldx % 4 || ldx < C || ldy < N)
    return DA_EINVAL;
  if (R < 1 || rows % R || (long)R * L < CB_TM + 2 || (!stat_part && !in_pend)) return DA_EINVAL;
  if (in_pend && (!in_mean || !in_invstd || !gamma || !beta || C > 1024)) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const long M = (long)rows * L;
  if (M >= 0x7fffffffl) return DA_EINVAL;
  ConvBf16Args a;
  a.x = x; a.w = reinterpret_cast<const __bf16*>(wpk); a.y = y;
  a.M = (int)M; a.L = L; a.ldx = ldx; a.C = C; a.ldy = ldy; a.N = N; a.accumulate = 0;
  a.divL = make_fastdiv((uint32_t)L);
  a.Wn = R * L; a.stat_part = stat_part; a.in_pend = in_pend; a.in_mean = in_mean; a.in_invstd = in_invstd;
  a.in_gamma = gamma; a.in_beta = beta; a.in_tiles = (int)((M + 63) / 64); a.in_eps = eps;
  const long tiles = ((M + CB_TM - 1) / CB_TM) * (N / CB_TN);
  if (tiles > 0x7fffffffl) return DA_EINVAL;
  const size_t shm = in_pend ? (size_t)4 * C * sizeof(float) : 0;
#define CB_BN_LAUNCH(ST, XF_)                                                                                           \
  DA_ACT_DISPATCH(hipLaunchKernelGGL((conv3_bf16_bn_kernel<AT, ST, XF_>), dim3((unsigned)tiles), dim3(256), shm, stream, a))
  if (stat_part && in_pend) CB_BN_LAUNCH(true, true);
  else if (in_pend) CB_BN_LAUNCH(false, true);
  else CB_BN_LAUNCH(true, false);
#undef CB_BN_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// da_conv_gemm_multi's contract with bf16 operands (fp32 in / out / sums) for the cases whose sources are linear in the
// flat output index: per job Lsrc == src_stride * Lm, source offsets within a span of 2, one source (x2 == NULL); all
// jobs of a call share src_stride (1 or 2); w: bf16 [taps][N][C].  Up to 4 jobs per launch.
// Y[row][j*dst_stride+dst_off][n] (+)= sum_t sum_c X[row][j*src_stride+src_off[t]][c] * Wp[wtap[t]][n][c]
// replaces reference models/resnet.py:5-8,126-128 (stride-2 conv, 1x1 downsample) under dtype bf16
int da_conv_bf16_multi(const da_conv_job* jobs, int n, hipStream_t stream) {
  DA_ENTER();
  if (n < 0 || (n && !jobs)) return DA_EINVAL;
  for (int base = 0; base < n; base += 4) {
    ConvBf16GenTable t;
    const int m = n - base < 4 ? n - base : 4;
    int blocks = 0, cnt = 0;
    const int ss = jobs[base].src_stride;
    for (int i = 0; i < m; ++i) {
      const da_conv_job& j = jobs[base + i];
      if (!j.x || !j.w || !j.y || (j.x2 && (!j.w2 || j.tap_split < 1 || j.tap_split >= j.ntaps)) || j.rows < 0 || j.Lm < 1 || j.C % 32 || j.N % CB_TN || j.C < 32 || j.N < CB_TN ||
          j.ldx % 4 || j.ldx < j.C || j.ldy < j.N || j.ntaps < 1 || j.ntaps > 3 || (ss != 1 && ss != 2) ||
          j.src_stride != ss || j.Lsrc != ss * j.Lm || j.dst_stride < 1 || j.dst_off < 0 ||
          (j.Lm - 1) * j.dst_stride + j.dst_off >= j.Ldst)
        return DA_EINVAL;
      int lo = j.src_off[0], hi = j.src_off[0];
      for (int k = 0; k < j.ntaps; ++k) {
        if (j.wtap[k] < 0 || j.wtap[k] > 2) return DA_EINVAL;
        lo = j.src_off[k] < lo ? j.src_off[k] : lo;
        hi = j.src_off[k] > hi ? j.src_off[k] : hi;
      }
      if (hi - lo > 2) return DA_EINVAL;
      const long M = (long)j.rows * j.Lm;
      if (M >= 0x7fffffffl || (long)j.rows * j.Lsrc >= 0x7fffffffl || (long)j.rows * j.Ldst >= 0x7fffffffl) return DA_EINVAL;
      if (M == 0) continue;
      ConvBf16GenArgs& a = t.d[cnt];
      a.x = j.x; a.w = reinterpret_cast<const __bf16*>(j.w); a.y = j.y;
      a.M = (int)M; a.Lm = j.Lm; a.Lsrc = j.Lsrc; a.ldx = j.ldx; a.C = j.C; a.Ldst = j.Ldst; a.ldy = j.ldy; a.N = j.N;
      a.dst_stride = j.dst_stride; a.dst_off = j.dst_off; a.ntaps = j.ntaps; a.accumulate = j.accumulate;
      for (int k = 0; k < 3; ++k) {
        a.src_off[k] = k < j.ntaps ? j.src_off[k] : j.src_off[0];
        a.wtap[k] = k < j.ntaps ? j.wtap[k] : j.wtap[0];
      }
      a.Msrc = (long)j.rows * j.Lsrc;
      a.divLm = make_fastdiv((uint32_t)j.Lm);
      a.x2 = j.x2; a.w2 = reinterpret_cast<const __bf16*>(j.w2); a.tap_split = j.x2 ? j.tap_split : j.ntaps;
      t.first_block[cnt] = blocks;
      blocks += (int)(((M + CB_TM - 1) / CB_TM) * (j.N / CB_TN));
      ++cnt;
    }
    if (!cnt) continue;
    t.n = cnt;
    t.first_block[cnt] = blocks;
    if (ss == 1) DA_ACT_DISPATCH(hipLaunchKernelGGL((conv_bf16_gen_kernel<1, AT>), dim3((unsigned)blocks), dim3(256), 0, stream, t));
    else DA_ACT_DISPATCH(hipLaunchKernelGGL((conv_bf16_gen_kernel<2, AT>), dim3((unsigned)blocks), dim3(256), 0, stream, t));
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

// bf16 tap packs of one (Co, Ci, 3) fp32 conv weight: wf [3][Co][Ci] (forward), wd [3][Ci][Co] (data gradient; taps
// reversed); either may be NULL.  Round-to-nearest-even.
int da_pack_conv3_bf16(const float* w, void* wf, void* wd, int co, int ci, hipStream_t stream) {
  DA_ENTER();
  if (!w || (!wf && !wd) || co < 1 || ci < 1) return DA_EINVAL;
  const size_t total = (size_t)co * ci;
  hipLaunchKernelGGL(pack_conv3_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w,
                     reinterpret_cast<__bf16*>(wf), reinterpret_cast<__bf16*>(wd), co, ci);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

}  // extern "C"
