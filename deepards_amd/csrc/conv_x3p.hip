// fp32 convolutions on the bf16 matrix cores with PRE-SPLIT operands ("f32x3", storage form): the k3 s1 p1 Conv1d forward
// / data gradient of conv_x3.hip (reference models/resnet.py:5-8,27-38) fed by activations their PRODUCERS already wrote
// as exact three-term bf16 splits (common.h: the "x3" activation format), so this kernel has no VALU work in its K loop
// at all -- it moves bf16 bytes into LDS and issues MFMAs.
//
// Why: on gfx950 the fp32 matrix rate is 1/16 of the bf16 one.  x = h + m + l with h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m) is EXACT for finite fp32 (3 x 8 significand bits), each bf16 x bf16 product is exact in the MFMA's
// fp32 accumulate, and x w = hh' + hm' + mh' + hl' + lh' + mm' + (terms below 2^-23 |x w|, one fp32 rounding): fp32
// results at 6/16 of the fp32 MFMA cost, in DIRECT form (no Winograd error growth: measured error vs fp64 below the
// native fp32 direct kernel's).  Round 2 built this with the split done while staging (conv_x3.hip) and found it
// break-even: the split costs ~6 VALU instructions per staged element on the issue port the MFMAs share, and the tripled
// panel went through registers -> ds_write.  Here the BatchNorm / pool kernels (latency-bound: the 1.5x bytes are free
// there) store h | m | l, the weights come pre-split from the batched repack, and both operands reach the MFMAs
// through LDS with plain 16-byte copies.
//
// x3 activation format (common.h): per position C/16 groups of [h 16 ch | m 16 ch | l 16 ch] bf16 = 96 bytes per group,
// 6 C bytes per position: one K step (16 channels) of one panel row is ONE contiguous 96-byte run.
//
// Two forms of the same tiles, same MFMA order, bit-identical results: LDS-DMA staging into a 3-slot ring (the default,
// conv3_x3p_dma_kernel below) and register staging (conv3_x3p_kernel, build with -DXP_DEFAULT_KERNEL=1: the form of round 3's first
// ablations; kept for A/B).
//
// Block = (64 MT) positions x 64 output channels, 4 waves of (32 MT) x 32 (v_mfma_f32_32x32x16_bf16, MT accumulators);
// MT = 2 for the full tiles, MT = 1 for the tiles of the partly filled last round (see da_conv3_x3p).  K step = 16
// channels x 3 taps = 18 MT MFMAs per wave.  LDS per K step: the activation panel [64 MT + 2 rows + a zero row][112 B]
// (96 data + 16 pad: the 16 rows a ds_read_b128 lane group touches fall on 16 different 16-byte slots) and the weight
// chunk [3 taps][2 n halves][3 terms][64 lanes][16 B] = 18 KB exactly as the repack lays it out in HBM (fragment-major:
// a wave's B fragment is one conflict-free 1 KB read) -- both double-buffered, ONE barrier per K step; the loads of step
// k + 2 are in flight while step k multiplies.  Sequence edges: a tap that would cross one reads the zero row.
#include "common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef XP_DEFAULT_KERNEL
#define XP_DEFAULT_KERNEL 4
#endif
#define XP_TN 64
#define XP_PITCH 112
#define XP_BCHUNK (18 * 1024)                       // bytes of one (64-channel tile, K step) weight chunk

__device__ __forceinline__ int xcd_chunked_xp(int id, int total) {   // consecutive work items share an XCD (conv_gemm.hip)
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, s = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
}

// LDS fragment reads the compiler does not track (it would wait for ALL outstanding LDS reads in front of every MFMA group):
// the K loop issues the reads of the next group, then waits -- by count -- for the current group's only.
__device__ __forceinline__ f32x4 lds_read16(const unsigned char* p) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x4 (&a)[3], f32x4 (&b)[3]) {     // at most N LDS operations may still be in flight
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2])
               : "n"(N)
               : "memory");
}

struct ConvX3pArgs {
  const __bf16* x;      // [M] positions of x3 format, C channels: 3 C bf16 per position
  const __bf16* w;      // [N / 64][C / 16] chunks of XP_BCHUNK bytes (da_repack_desc.points = 49)
  float* y;             // [M][ldy] fp32, first N channels
  int M, L, C, ldy, N, accumulate;
  int full_m;           // M-tile rows (of 256 positions) that run as full tiles: full_m * (N / 64) blocks
  int tail_m;           // the remaining M-tile rows run as 4 * tail_m * (N / 64) tiles of 64 x 64, FIRST in the launch
  FastDiv divL;
};

// One tile of (32 MT WM) positions x 64 output channels by the calling block's NT threads: waves 0 .. 2 WM - 1 compute
// (wave = (wm, wn): rows wm * 32 MT .., channels wn * 32 ..), every thread of the block stages.
template <int MT, int WM, int NT>
__device__ __forceinline__ void conv3_x3p_body(const ConvX3pArgs& a, const int P0, const int n_blk, unsigned char* lds, const bool early) {
  constexpr int TM = 32 * MT * WM, XROWS = TM + 2, PROWS = XROWS + 1;     // + the zero row
  constexpr int XBYTES = PROWS * XP_PITCH;
  constexpr int NXP = (XROWS * 6 + NT - 1) / NT;                     // 16-byte pieces of the panel per thread
  constexpr int NBP = (1152 + NT - 1) / NT;                          // ... of the 18 KB weight chunk
  unsigned char* Xs = lds;                          // 2 x [PROWS][112]
  unsigned char* Bs = lds + 2 * XBYTES;             // 2 x 18 KB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kch = a.C >> 4;
  const size_t xrow_bytes = (size_t)a.C * 6;

  // panel loader: piece q of the panel = (row q / 6, 16-byte slot q % 6).  Rows outside [0, M) are read from a clamped
  // (valid) address: no output that is stored ever uses them (a tap that would reads the zero row).
  const unsigned char* xsrc[NXP];
  int xdst[NXP];
  bool xon[NXP];
#pragma unroll
  for (int p = 0; p < NXP; ++p) {
    const int q = tid + NT * p;
    xon[p] = q < XROWS * 6;
    const int r = xon[p] ? q / 6 : 0, s = xon[p] ? q - r * 6 : 0;
    long P = (long)P0 - 1 + r;
    P = P < 0 ? 0 : (P >= a.M ? a.M - 1 : P);
    xsrc[p] = reinterpret_cast<const unsigned char*>(a.x) + (size_t)P * xrow_bytes + s * 16;
    xdst[p] = r * XP_PITCH + s * 16;
  }
  // weight chunk loader: 1152 pieces of 16 bytes, linear
  const unsigned char* bsrc = reinterpret_cast<const unsigned char*>(a.w) + (size_t)(n_blk >> 6) * kch * XP_BCHUNK + tid * 16;
  bool bon[NBP];
#pragma unroll
  for (int i = 0; i < NBP; ++i) bon[i] = tid + NT * i < 1152;

  f32x4 rx[NXP], rb[NBP];
  auto gload = [&](int ks) {
#pragma unroll
    for (int p = 0; p < NXP; ++p)
      if (xon[p]) rx[p] = *reinterpret_cast<const f32x4*>(xsrc[p] + ks * 96);
    const unsigned char* b = bsrc + (size_t)ks * XP_BCHUNK;
#pragma unroll
    for (int i = 0; i < NBP; ++i)
      if (bon[i]) rb[i] = *reinterpret_cast<const f32x4*>(b + i * NT * 16);
  };
  auto stage = [&](int buf) {
    unsigned char* xs = Xs + buf * XBYTES;
    unsigned char* bs = Bs + buf * XP_BCHUNK + tid * 16;
#pragma unroll
    for (int p = 0; p < NXP; ++p)
      if (xon[p]) *reinterpret_cast<f32x4*>(xs + xdst[p]) = rx[p];
#pragma unroll
    for (int i = 0; i < NBP; ++i)
      if (bon[i]) *reinterpret_cast<f32x4*>(bs + i * NT * 16) = rb[i];
  };

  const bool computes = wave < 2 * WM;              // (the tail tiles of a 512-thread block: waves 4 .. 7 only stage)
  const int frow = lane & 31, kg = lane >> 5;
  const int wm = computes ? wave >> 1 : 0, wn = wave & 1;
  // LDS offset of the A fragment of (row tile mt, tap t): the panel row of position + t - 1, or the zero row when that
  // position lies across a sequence edge
  int aoff[MT][3];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long P = (long)P0 + (wm * MT + mt) * 32 + frow;
    const uint32_t Pc = (uint32_t)(P < a.M ? P : 0);
    const int l = (int)(Pc - fdiv(Pc, a.divL) * (uint32_t)a.L);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bool edge = (t == 0 && l == 0) || (t == 2 && l == a.L - 1);
      aoff[mt][t] = (edge ? XROWS : (wm * MT + mt) * 32 + t + frow) * XP_PITCH + kg * 16;
    }
  }
  const int boff = wn * 3 * 1024 + lane * 16;       // + tap * 6 KB + term * 1 KB

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  if (tid < 14) {                                   // the zero rows of both panel buffers (7 16-byte slots each)
    const int bsel = tid >= 7, piece = tid - 7 * bsel;
    *reinterpret_cast<f32x4*>(Xs + bsel * XBYTES + XROWS * XP_PITCH + piece * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  gload(0);
  stage(0);
  gload(kch > 1 ? 1 : 0);
  __syncthreads();
  for (int ks = 0; ks < kch; ++ks) {
    const int cur = ks & 1;
    const unsigned char* xs = Xs + cur * XBYTES;
    const unsigned char* bs = Bs + cur * XP_BCHUNK + boff;
    // One K step = 3 taps x MT row tiles = 3 MT groups of 6 MFMAs.  The fragments of group g + 1 are read from LDS BEFORE the
    // MFMAs of group g issue (register double buffer: an LDS round trip hides behind 192 MFMA cycles instead of stalling
    // both waves of a SIMD at the same moment), and the staging of step ks + 1 / the loads of step ks + 2 sit between the
    // first groups instead of in front of them, where the matrix pipe would idle.
    constexpr int G = 3 * MT;
    f32x4 av[2][3], bv[2][3];
    auto ld_frags = [&](int g) {
      const int t = g / MT, mt = g % MT;
      if (mt == 0) {
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) bv[t & 1][s_] = lds_read16(bs + t * 6144 + s_ * 1024);
      }
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) av[g & 1][s_] = lds_read16(xs + aoff[mt][t] + s_ * 32);
    };
    // The two waves that share a SIMD (waves w and w + 4) must not do the same thing at the same time, or the matrix pipe
    // idles while both stage and both queue for it afterwards (measured: MFMA, staging and load time simply ADDED UP):
    // waves 0-3 stage step ks + 1 and load step ks + 2 BEFORE their MFMAs, waves 4-7 AFTER theirs -- within one barrier
    // interval one half computes while the other moves bytes.
    if (early) {
      stage(cur ^ 1);                               // step ks + 1 (already in registers); its buffer was released by the last barrier
      gload(ks + 2 < kch ? ks + 2 : kch - 1);
    }
    if (computes) ld_frags(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (computes && g + 1 < G) ld_frags(g + 1);
      if (computes) {
        const int mt = g % MT, tb = (g / MT) & 1, sl = g & 1;
        // the reads of group g are older than everything issued since: the next group's 3 (+ 3 with a new tap's weights)
        if (g + 1 >= G) lds_wait<0>(av[sl], bv[tb]);
        else if ((g + 1) % MT == 0) lds_wait<6>(av[sl], bv[tb]);
        else lds_wait<3>(av[sl], bv[tb]);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bv[tb][0]), bm = __builtin_bit_cast(bf16x8, bv[tb][1]),
                     bl = __builtin_bit_cast(bf16x8, bv[tb][2]);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, av[sl][0]), am = __builtin_bit_cast(bf16x8, av[sl][1]),
                     al = __builtin_bit_cast(bf16x8, av[sl][2]);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mt], 0, 0, 0);      // small terms first
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!early) {
      stage(cur ^ 1);
      gload(ks + 2 < kch ? ks + 2 : kch - 1);
    }
    __syncthreads();
  }

  // lane holds output channel n_blk + wn*32 + l%32 of the positions (r & 3) + 8 (r >> 2) + 4 (l / 32) of each 32-row tile
  if (computes) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long P = (long)P0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (P < a.M) {
          float* o = a.y + P * a.ldy + n_blk + wn * 32 + frow;
          float v = acc[mt][r];
          if (a.accumulate) v += *o;
          *o = v;
        }
      }
  }
}

#define XP_TM 256                                    // full tile: 8 waves = 4 (rows) x 2 (channels), 2 accumulators each
#define XP_LDS_BYTES (2 * (XP_TM + 3) * XP_PITCH + 2 * XP_BCHUNK)

// Work items: full_m * (N / 64) tiles of 256 x 64 (8 waves; ONE resident block per CU: 95 KB of LDS -- a tile this tall
// is what brings the operand traffic down to what the L2 -> LDS path sustains: 18.7 bytes per clock and CU against 26.7
// for 128 x 64 tiles, of ~35 achievable) and, for the partly filled last round of a launch, 64 x 64 tiles (a quarter of
// the MFMA time on 4 of the block's 8 waves, the others help staging): the FIRST blocks of the launch.
__global__ __launch_bounds__(512, 1) void conv3_x3p_kernel(ConvX3pArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // XP_LDS_BYTES (> 64 KB: dynamic)
  const int ntn = a.N / XP_TN;
  const int tail_blocks = 4 * a.tail_m * ntn;
  if ((int)blockIdx.x < tail_blocks) {
    const int id = xcd_chunked_xp(blockIdx.x, tail_blocks);        // channel tile fastest: the tiles of one panel share an L2
    conv3_x3p_body<1, 2, 512>(a, a.full_m * XP_TM + (id / ntn) * 64, (id % ntn) * XP_TN, lds, threadIdx.x < 256);
  } else {
    const int tile = xcd_chunked_xp(blockIdx.x - tail_blocks, a.full_m * ntn);
    conv3_x3p_body<2, 4, 512>(a, (tile / ntn) * XP_TM, (tile % ntn) * XP_TN, lds, threadIdx.x < 256);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA form (variant 4): the same tiles and the same MFMA order, but the operands go global -> LDS by
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write) into a ring of THREE slots: the pieces of K step k + 2 are
// issued while step k multiplies and have a whole step to land (a counted vmcnt in front of the barrier leaves the newest
// ones in flight).  A slot = [activation panel: 258 rows x 7 16-byte granules (6 data + 1 pad), padded to whole 1 KB
// pieces][512 zero bytes][weight chunk 18 KB]; one wave instruction writes 64 consecutive granules from per-lane sources
// (the lanes that fall on a pad granule fetch their left neighbour again).  A tap across a sequence edge reads zeros at
// the address of the zero region that is congruent mod 256 to the lane's own panel address: no lane of its ds_read_b128
// lane group sits on those banks (one shared zero row cost a conflict cycle on every such group: 28 % of the LDS cycles
// at L = 7).  No staging phase is left, so the two waves of a SIMD are not staggered.
// ---------------------------------------------------------------------------------------------------------------------
#define XD_ZOFF 29696                               // 29 pieces of 1 KB hold the 258 x 112 bytes of the panel
#define XD_BOFF (XD_ZOFF + 512)
#define XD_SLOT (XD_BOFF + XP_BCHUNK)               // 48,640 = 190 x 256 bytes
#define XD_LDS_BYTES (3 * XD_SLOT)                  // 145,920

typedef __attribute__((address_space(3))) unsigned char lds_byte;

template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MT, int WM>
__device__ __forceinline__ void conv3_x3p_dma_body(const ConvX3pArgs& a, const int P0, const int n_blk, unsigned char* lds) {
  constexpr int TM = 32 * MT * WM, XROWS = TM + 2, NGX = XROWS * 7;
  constexpr int NJX = (NGX + 63) / 64, NJ = NJX + 18, NI = (NJ + 7) / 8;        // 1 KB pieces per K step; per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kch = a.C >> 4;
  const size_t xrow_bytes = (size_t)a.C * 6;

  // piece j = wave + 8 i of a K step: panel granules 64 j .. 64 j + 63 (j < NJX) or 1 KB of the weight chunk
  const unsigned char* src[NI];                     // this lane's source at K step 0
  int dst[NI], inc[NI];                             // LDS offset of the piece in a slot (wave-uniform); bytes per K step
  bool on[NI];
  const long Pb = P0 > 0 ? P0 - 1 : 0;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = wave + 8 * i;
    on[i] = j < NJ;
    if (j < NJX) {
      const int q = 64 * j + lane;
      int r = q / 7, sl = q - r * 7;
      r = r < XROWS ? r : XROWS - 1;
      sl = sl < 6 ? sl : 5;
      long P = (long)P0 - 1 + r;
      P = P < 0 ? 0 : (P >= a.M ? a.M - 1 : P);
      src[i] = reinterpret_cast<const unsigned char*>(a.x) + (size_t)P * xrow_bytes + sl * 16;
      dst[i] = j * 1024;
      inc[i] = 96;
    } else {
      const int jb = on[i] ? j - NJX : 0;
      src[i] = reinterpret_cast<const unsigned char*>(a.w) + (size_t)(n_blk >> 6) * kch * XP_BCHUNK + jb * 1024 + lane * 16;
      dst[i] = XD_BOFF + jb * 1024;
      inc[i] = XP_BCHUNK;
    }
  }
  (void)Pb;
  auto issue1 = [&](int i, int ks, int slot) {      // piece i of K step ks -> ring slot
    if (on[i])
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned char*)(src[i] + (size_t)ks * inc[i]),
                                       (lds_byte*)(lds + slot * XD_SLOT + dst[i]), 16, 0, 0);
  };
  auto issue = [&](int ks, int slot) {
#pragma unroll
    for (int i = 0; i < NI; ++i) issue1(i, ks, slot);
  };

  const bool computes = WM == 4 || wave < 2 * WM;   // (tail tiles: waves 4 .. 7 only move bytes)
  const int frow = lane & 31, kg = lane >> 5;
  const int wm = computes ? wave >> 1 : 0, wn = wave & 1;
  int aoff[MT][3];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long P = (long)P0 + (wm * MT + mt) * 32 + frow;
    const uint32_t Pc = (uint32_t)(P < a.M ? P : 0);
    const int l = (int)(Pc - fdiv(Pc, a.divL) * (uint32_t)a.L);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bool edge = (t == 0 && l == 0) || (t == 2 && l == a.L - 1);
      const int o = ((wm * MT + mt) * 32 + t + frow) * XP_PITCH + kg * 16;
      aoff[mt][t] = edge ? XD_ZOFF + (o & 255) : o;
    }
  }
  const int boff = XD_BOFF + wn * 3 * 1024 + lane * 16;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  if (tid < 96) {                                   // the zero regions of the three slots (32 granules each)
    const int sl = tid >> 5;
    *reinterpret_cast<f32x4*>(lds + sl * XD_SLOT + XD_ZOFF + (tid & 31) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  issue(0, 0);
  issue(kch > 1 ? 1 : 0, 1);
  // The counted wait is PER WAVE and comes from the same on[] that gates issue1(): a wave issues sum(on[i]) pieces per K
  // step, and since piece j = wave + 8 i grows with i only its LAST index can fall beyond the step's NJ pieces -- the
  // count is NI or NI - 1, nothing else (one shared constant let a wave with NI - 1 pieces pass the barrier with its
  // last piece of the step still in flight: the intermittent wrong result of round 3).
  static_assert(8 * (NI - 1) < NJ && NJ <= 8 * NI, "every wave issues NI or NI - 1 pieces per K step");
  int issued = 0;
#pragma unroll
  for (int i = 0; i < NI; ++i) issued += on[i] ? 1 : 0;
  auto land = [&]() {                               // everything but this wave's NEWEST step of pieces has landed
    if (issued == NI) vm_wait<NI>();
    else vm_wait<NI - 1>();
  };
  land();                                           // step 0 (this wave's pieces; the barrier covers the others')
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // The K loop is software-pipelined ACROSS the barrier: the barrier of step ks sits in front of the step's LAST MFMA group
  // (every fragment of the step is in registers by then, so the slot is free for the pieces of step ks + 3), and the
  // first fragments of step ks + 1 are read right behind it, under that last group's 192 MFMA cycles.  With the barrier
  // at the end of the step both waves of every SIMD started each step together with address arithmetic and an exposed
  // LDS round trip, matrix pipe idle (measured: 3,700 cycles per step against 2,200 with the waves left to drift apart).
  constexpr int G = 3 * MT;
  constexpr int NA = (G & 1) ? 3 : 2;               // A fragment buffers: the last group's and the next step's first differ
  constexpr int PER = (NI + G - 2) / (G - 1);       // pieces per group: all of a step's pieces go out before its barrier
  f32x4 av[NA][3], bv[3][3];
  int slot = 0;
  auto ld_frags = [&](const unsigned char* xs, int g) {
    const int t = g / MT, mt = g % MT;
    if (mt == 0) {
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) bv[t][s_] = lds_read16(xs + boff + t * 6144 + s_ * 1024);
    }
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) av[g % NA][s_] = lds_read16(xs + aoff[mt][t] + s_ * 32);
  };
  if (computes) ld_frags(lds, 0);
  for (int ks = 0; ks < kch; ++ks) {
    const unsigned char* xs = lds + slot * XD_SLOT;
    // K step ks + 2 -> the slot step ks - 1 was read from (released by that step's barrier), a piece or two in front of
    // each MFMA group: a burst of all of them holds every wave at its vector-memory issue, matrix pipe idle
    const int nslot = slot == 0 ? 2 : slot - 1;
    const int nks = ks + 2 < kch ? ks + 2 : kch - 1;
    const int slot1 = slot == 2 ? 0 : slot + 1;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int i = g * PER; i < (g + 1) * PER && i < NI; ++i) issue1(i, nks, nslot);
      if (computes && g + 1 < G) ld_frags(xs, g + 1);
      const int mt = g % MT, t = g / MT, sl = g % NA;
      if (g + 1 < G) {
        if (computes) {
          if ((g + 1) % MT == 0) lds_wait<6>(av[sl], bv[t]);      // younger than group g's reads: those of group g + 1
          else lds_wait<3>(av[sl], bv[t]);
        }
      } else {
        if (computes) lds_wait<0>(av[sl], bv[t]);   // every read of this slot has returned
        land();                                     // step ks + 1 has landed; step ks + 2 stays in flight across the barrier
        __builtin_amdgcn_s_barrier();
        if (computes) ld_frags(lds + slot1 * XD_SLOT, 0);        // (after the last step: a slot nobody writes any more)
        __builtin_amdgcn_sched_barrier(0);          // the reads go out BEFORE the group's MFMAs: they are its cover
      }
      if (computes) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bv[t][0]), bm = __builtin_bit_cast(bf16x8, bv[t][1]),
                     bl = __builtin_bit_cast(bf16x8, bv[t][2]);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, av[sl][0]), am = __builtin_bit_cast(bf16x8, av[sl][1]),
                     al = __builtin_bit_cast(bf16x8, av[sl][2]);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mt], 0, 0, 0);      // small terms first
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = slot1;
  }
  if (computes) lds_wait<0>(av[0], bv[0]);          // the reads issued behind the last barrier
  vm_wait<0>();

  if (computes) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long P = (long)P0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (P < a.M) {
          float* o = a.y + P * a.ldy + n_blk + wn * 32 + frow;
          float v = acc[mt][r];
          if (a.accumulate) v += *o;
          *o = v;
        }
      }
  }
}

__global__ __launch_bounds__(512, 1) void conv3_x3p_dma_kernel(ConvX3pArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];      // XD_LDS_BYTES
  const int ntn = a.N / XP_TN;
  const int tail_blocks = 4 * a.tail_m * ntn;
  if ((int)blockIdx.x < tail_blocks) {
    const int id = xcd_chunked_xp(blockIdx.x, tail_blocks);
    conv3_x3p_dma_body<1, 2>(a, a.full_m * XP_TM + (id / ntn) * 64, (id % ntn) * XP_TN, lds);
  } else {
    const int tile = xcd_chunked_xp(blockIdx.x - tail_blocks, a.full_m * ntn);
    conv3_x3p_dma_body<2, 4>(a, (tile / ntn) * XP_TM, (tile % ntn) * XP_TN, lds);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The stride-2 block entry on x3 operands: the k3 s2 p1 conv and the 1x1 s2 downsample conv of a BasicBlock read the same
// input (reference models/resnet.py:16-19,27-29,123-131), their data gradients add into the same dx -- ONE launch each
// way, same ring / DMA / pipelining as conv3_x3p_dma_body.  Units u = row * Lout + j (the stride-2 side, Lin = 2 Lout):
//   forward   y1[u] = W1[0] x[2u-1] + W1[1] x[2u] + W1[2] x[2u+1],   yd[u] = Wd x[2u]           (x[2u-1] = 0 at j = 0)
//   backward  dx[2u] = W1[1]' dy1[u] + Wd' dyd[u],   dx[2u+1] = W1[2]' dy1[u] + W1[0]' dy1[u+1]   (dy1[u+1] = 0 at j = Lout-1)
// Both are FOUR 32 x 32 products per wave and K step over THREE operand fragments -- A0 = class-A row i, A1 = class-B row
// i, A2 = class-A row i + 1 -- where class A / B are what the panel holds: forward the odd / even input positions
// (de-interleaved by the DMA's source addresses: every tap reads unit-stride LDS rows), backward dy1 / dyd.  Tile =
// 128 units x 64 output channels (two accumulators per wave); the weights of a K step are the k3 chunk (18 KB,
// repack code 49) and tap 1 of the chunk that holds the 1x1 weights (6 KB).
// ---------------------------------------------------------------------------------------------------------------------
#define S2_WOFF 29696                               // panel: (TU + 1) + TU rows x 112 bytes, padded to 29 KB
#define S2_SLOT (S2_WOFF + XP_BCHUNK + 6144)        // 54,272 = 212 x 256 bytes
#define S2_ZOFF (3 * S2_SLOT)                       // ONE zero region behind the ring (a lane that reads it takes no slot base)
#define S2_LDS_BYTES (S2_ZOFF + 512)                // 163,328 of 163,840

struct ConvX3pS2Args {
  const __bf16* xa;     // class A operand, x3 format: forward x, backward dy1
  const __bf16* xb;     // class B operand: forward x, backward dyd
  const __bf16* w1;     // k3 weights, repack code 49: forward pack / data-gradient pack
  const __bf16* wd;     // 1x1 weights in tap 1 of a code-49 pack
  float* y0;            // forward y1 [M][N]; backward dx [2 M][N]
  float* y1;            // forward yd [M][N]; backward unused
  int M, Lu, K, N;      // units, units per sequence, contraction channels, output channels of this launch
  int full_m, tail_m;   // as ConvX3pArgs: tiles of 128 units; the partly filled last round as tiles of 64, first in the launch
  FastDiv divLu;
};

template <bool DGRAD, int WM>
__device__ __forceinline__ void conv_x3p_s2_body(const ConvX3pS2Args& a, const int u0, const int n_blk, unsigned char* lds) {
  constexpr int TU = 32 * WM, RB = TU + 1, XROWS = 2 * TU + 1, NGX = XROWS * 7;   // class B rows start at RB
  constexpr int NJX = (NGX + 63) / 64, NJ = NJX + 24, NI = (NJ + 7) / 8;
  constexpr int G = 4, PER = (NI + G - 2) / (G - 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kch = a.K >> 4;
  const size_t row_bytes = (size_t)a.K * 6;
  const long pmax = DGRAD ? (long)a.M - 1 : 2l * a.M - 1;

  const unsigned char* src[NI];
  int dst[NI], inc[NI];
  bool on[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = wave + 8 * i;
    on[i] = j < NJ;
    if (j < NJX) {
      const int q = 64 * j + lane;
      int r = q / 7, sl = q - r * 7;
      r = r < XROWS ? r : XROWS - 1;
      sl = sl < 6 ? sl : 5;
      const bool cb = r >= RB;
      const long u = (long)u0 + (cb ? r - RB : r);
      long P = DGRAD ? u : (cb ? 2 * u : 2 * u - 1);
      P = P < 0 ? 0 : (P > pmax ? pmax : P);
      src[i] = reinterpret_cast<const unsigned char*>(cb ? a.xb : a.xa) + (size_t)P * row_bytes + sl * 16;
      dst[i] = j * 1024;
      inc[i] = 96;
    } else {
      const int jb = on[i] ? j - NJX : 0;             // 0 .. 17: the k3 chunk; 18 .. 23: tap 1 of the 1x1 chunk
      const size_t chunk0 = (size_t)(n_blk >> 6) * kch * XP_BCHUNK;
      src[i] = jb < 18 ? reinterpret_cast<const unsigned char*>(a.w1) + chunk0 + jb * 1024 + lane * 16
                       : reinterpret_cast<const unsigned char*>(a.wd) + chunk0 + 6144 + (jb - 18) * 1024 + lane * 16;
      dst[i] = S2_WOFF + jb * 1024;
      inc[i] = XP_BCHUNK;
    }
  }
  auto issue1 = [&](int i, int ks, int slot) {
    if (on[i])
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned char*)(src[i] + (size_t)ks * inc[i]),
                                       (lds_byte*)(lds + slot * S2_SLOT + dst[i]), 16, 0, 0);
  };
  static_assert(8 * (NI - 1) < NJ && NJ <= 8 * NI, "every wave issues NI or NI - 1 pieces per K step");
  int issued = 0;                                   // this wave's pieces per K step, from the on[] that gates issue1()
#pragma unroll
  for (int i = 0; i < NI; ++i) issued += on[i] ? 1 : 0;
  auto land = [&]() {
    if (issued == NI) vm_wait<NI>();
    else vm_wait<NI - 1>();
  };

  const bool computes = WM == 4 || wave < 2 * WM;
  const int frow = lane & 31, kg = lane >> 5;
  const int wm = computes ? wave >> 1 : 0, wn = wave & 1;
  const int iu = wm * 32 + frow;                    // this lane's unit inside the tile
  // fragment offsets inside a slot; the one fragment that can fall across a sequence edge reads the zero region instead
  // (absolute address, congruent mod 256 to its own: conflict-free), `zsel` = 0 for such a lane and ~0 otherwise
  const int aoffs[3] = {iu * XP_PITCH + kg * 16, (RB + iu) * XP_PITCH + kg * 16, (iu + 1) * XP_PITCH + kg * 16};
  const long um = (long)u0 + iu;
  const uint32_t uc = (uint32_t)(um < a.M ? um : 0);
  const int ju = (int)(uc - fdiv(uc, a.divLu) * (uint32_t)a.Lu);
  const bool edge = DGRAD ? ju == a.Lu - 1 : ju == 0;
  constexpr int EF = DGRAD ? 2 : 0;                 // the fragment the edge applies to
  const int eoff = edge ? S2_ZOFF + (aoffs[EF] & 255) : aoffs[EF];
  const unsigned zsel = edge ? 0u : ~0u;
  const int boff = S2_WOFF + wn * 3 * 1024 + lane * 16;
  // groups: (A fragment, weight tap: 0 .. 2 of the k3 chunk, 3 = the 1x1 tap, accumulator)
  constexpr int GA[2][4] = {{0, 1, 2, 1}, {0, 1, 0, 2}};
  constexpr int GB[2][4] = {{0, 1, 2, 3}, {1, 3, 0, 2}};
  constexpr int GC[2][4] = {{0, 0, 0, 1}, {0, 0, 1, 1}};
  constexpr int GLA[2][4] = {{1, 1, 1, 0}, {1, 1, 0, 1}};     // does the group read a NEW A fragment

  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  if (tid < 32) *reinterpret_cast<f32x4*>(lds + S2_ZOFF + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NI; ++i) issue1(i, 0, 0);
#pragma unroll
  for (int i = 0; i < NI; ++i) issue1(i, kch > 1 ? 1 : 0, 1);
  land();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 av[3][3], bv[2][3];
  auto ld_frags = [&](int sbase, int g) {           // sbase: byte offset of the slot
    constexpr int D = DGRAD ? 1 : 0;
    const int tb = GB[D][g];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) bv[g & 1][s_] = lds_read16(lds + sbase + boff + tb * 6144 + s_ * 1024);
    if (GLA[D][g]) {
      const int fa = GA[D][g];
      const int o = fa == EF ? eoff + (int)((unsigned)sbase & zsel) : aoffs[fa] + sbase;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) av[fa][s_] = lds_read16(lds + o + s_ * 32);
    }
  };
  int slot = 0;
  if (computes) ld_frags(0, 0);
  for (int ks = 0; ks < kch; ++ks) {
    const int sbase = slot * S2_SLOT;
    const int nslot = slot == 0 ? 2 : slot - 1;
    const int nks = ks + 2 < kch ? ks + 2 : kch - 1;
    const int slot1 = slot == 2 ? 0 : slot + 1;
    constexpr int D = DGRAD ? 1 : 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int i = g * PER; i < (g + 1) * PER && i < NI; ++i) issue1(i, nks, nslot);
      if (computes && g + 1 < G) ld_frags(sbase, g + 1);
      const int fa = GA[D][g], sl = g & 1;
      if (g + 1 < G) {
        if (computes) {
          if (GLA[D][g + 1]) lds_wait<6>(av[fa], bv[sl]);       // younger than group g's reads: those of group g + 1
          else lds_wait<3>(av[fa], bv[sl]);
        }
      } else {
        if (computes) lds_wait<0>(av[fa], bv[sl]);
        land();
        __builtin_amdgcn_s_barrier();
        if (computes) ld_frags(slot1 * S2_SLOT, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (computes) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bv[sl][0]), bm = __builtin_bit_cast(bf16x8, bv[sl][1]),
                     bl = __builtin_bit_cast(bf16x8, bv[sl][2]);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, av[fa][0]), am = __builtin_bit_cast(bf16x8, av[fa][1]),
                     al = __builtin_bit_cast(bf16x8, av[fa][2]);
        f32x16& c = acc[GC[D][g]];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);      // small terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = slot1;
  }
  if (computes) lds_wait<0>(av[0], bv[0]);
  vm_wait<0>();

  if (computes) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long u = (long)u0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (u < a.M) {
        const int n = n_blk + wn * 32 + frow;
        if (DGRAD) {
          a.y0[(size_t)(2 * u) * a.N + n] = acc[0][r];
          a.y0[(size_t)(2 * u + 1) * a.N + n] = acc[1][r];
        } else {
          a.y0[(size_t)u * a.N + n] = acc[0][r];
          a.y1[(size_t)u * a.N + n] = acc[1][r];
        }
      }
    }
  }
}

template <bool DGRAD>
__global__ __launch_bounds__(512, 1) void conv_x3p_s2_kernel(ConvX3pS2Args a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];      // S2_LDS_BYTES
  const int ntn = a.N / XP_TN;
  const int tail_blocks = 2 * a.tail_m * ntn;
  if ((int)blockIdx.x < tail_blocks) {
    const int id = xcd_chunked_xp(blockIdx.x, tail_blocks);
    conv_x3p_s2_body<DGRAD, 2>(a, a.full_m * 128 + (id / ntn) * 64, (id % ntn) * XP_TN, lds);
  } else {
    const int tile = xcd_chunked_xp(blockIdx.x - tail_blocks, a.full_m * ntn);
    conv_x3p_s2_body<DGRAD, 4>(a, (tile / ntn) * 128, (tile % ntn) * XP_TN, lds);
  }
}

// fp32 [npos][ld] (first C channels) -> x3 [npos][C/16][3][16] and back (tests, and the boundaries where a producer
// without an x3 store form meets an x3 consumer)
__global__ __launch_bounds__(256) void x3_split_kernel(const float* __restrict__ x, int ld, __bf16* __restrict__ out, size_t npos,
                                                       int C) {
  const int nq = C >> 2;
  const size_t total = npos * nq;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pos = i / nq;
    const int c0 = (int)(i - pos * nq) * 4;
    X3::st4(out + pos * 3 * C, c0, *reinterpret_cast<const f32x4*>(x + pos * ld + c0));
  }
}
__global__ __launch_bounds__(256) void x3_merge_kernel(const __bf16* __restrict__ x, float* __restrict__ out, int ld, size_t npos,
                                                       int C) {
  const int nq = C >> 2;
  const size_t total = npos * nq;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pos = i / nq;
    const int c0 = (int)(i - pos * nq) * 4;
    *reinterpret_cast<f32x4*>(out + pos * ld + c0) = X3::ld4(x + pos * 3 * C, c0);
  }
}

extern "C" {

// y (+)= conv1d(x, k = 3, stride 1, pad 1) per row of L positions; x: x3 format [rows * L][C/16][3][16] bf16, wpk: the
// chunked split-bf16 pack (da_repack_desc.points = 49 / da_pack_conv3_x3p), y: [rows][L][ldy] fp32 (first N channels).
// C % 16 == 0, N % 64 == 0.  replaces reference models/resnet.py:5-8 (conv2x2), forward and (wd pack) data gradient
// The stride-2 block entry, forward: y1 = conv(k3, s2, p1)(x; w1), yd = conv(1x1, s2)(x; wd) from ONE read of x.
// x3: x3 activation (rows, Lin, C), Lin even; w1pk / wdpk: forward packs of repack code 49 (the 1x1 weights in tap 1);
// y1, yd: (rows, Lin / 2, N) fp32.
int da_conv_x3p_s2_fwd(const void* x3, const void* w1pk, const void* wdpk, float* y1, float* yd, int rows, int Lin, int C, int N,
                       hipStream_t stream) {
  DA_ENTER();
  if (!x3 || !w1pk || !wdpk || !y1 || !yd || rows < 0 || Lin < 2 || (Lin & 1) || C % 16 || C < 16 || N % XP_TN || N < XP_TN)
    return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const long M = (long)rows * (Lin / 2);
  if (2 * M >= 0x7fffffffl) return DA_EINVAL;
  ConvX3pS2Args a;
  a.xa = a.xb = reinterpret_cast<const __bf16*>(x3);
  a.w1 = reinterpret_cast<const __bf16*>(w1pk); a.wd = reinterpret_cast<const __bf16*>(wdpk);
  a.y0 = y1; a.y1 = yd;
  a.M = (int)M; a.Lu = Lin / 2; a.K = C; a.N = N;
  a.divLu = make_fastdiv((uint32_t)a.Lu);
  const int ntn = N / XP_TN;
  const long mtiles = (M + 127) / 128, tiles = mtiles * ntn;
  const long tail_m = tiles < 256 ? mtiles : (tiles % 256) / ntn;
  a.tail_m = (int)tail_m; a.full_m = (int)(mtiles - tail_m);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3p_s2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            S2_LDS_BYTES) != hipSuccess)
      return DA_EINVAL;
    attr = true;
  }
  hipLaunchKernelGGL(conv_x3p_s2_kernel<false>, dim3((unsigned)((long)a.full_m * ntn + 2l * tail_m * ntn)), dim3(512), S2_LDS_BYTES,
                     stream, a);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// ... and its data gradient: dx (rows, 2 Lout, C) = dgrad(k3 s2)(dy1; w1) + dgrad(1x1 s2)(dyd; wd), every position written.
// dy1_3, dyd_3: x3 activations (rows, Lout, N); w1pk / wdpk: DATA-GRADIENT packs of repack code 49.
int da_conv_x3p_s2_dgrad(const void* dy1_3, const void* w1pk, const void* dyd_3, const void* wdpk, float* dx, int rows, int Lout, int N,
                         int C, hipStream_t stream) {
  DA_ENTER();
  if (!dy1_3 || !w1pk || !dyd_3 || !wdpk || !dx || rows < 0 || Lout < 1 || N % 16 || N < 16 || C % XP_TN || C < XP_TN) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const long M = (long)rows * Lout;
  if (2 * M >= 0x7fffffffl) return DA_EINVAL;
  ConvX3pS2Args a;
  a.xa = reinterpret_cast<const __bf16*>(dy1_3); a.xb = reinterpret_cast<const __bf16*>(dyd_3);
  a.w1 = reinterpret_cast<const __bf16*>(w1pk); a.wd = reinterpret_cast<const __bf16*>(wdpk);
  a.y0 = dx; a.y1 = nullptr;
  a.M = (int)M; a.Lu = Lout; a.K = N; a.N = C;
  a.divLu = make_fastdiv((uint32_t)Lout);
  const int ntn = C / XP_TN;
  const long mtiles = (M + 127) / 128, tiles = mtiles * ntn;
  const long tail_m = tiles < 256 ? mtiles : (tiles % 256) / ntn;
  a.tail_m = (int)tail_m; a.full_m = (int)(mtiles - tail_m);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3p_s2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            S2_LDS_BYTES) != hipSuccess)
      return DA_EINVAL;
    attr = true;
  }
  hipLaunchKernelGGL(conv_x3p_s2_kernel<true>, dim3((unsigned)((long)a.full_m * ntn + 2l * tail_m * ntn)), dim3(512), S2_LDS_BYTES,
                     stream, a);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_conv3_x3p(const void* x, const void* wpk, float* y, int rows, int L, int C, int ldy, int N, int accumulate,
                 hipStream_t stream) {
  DA_ENTER();
  if (!x || !wpk || !y || rows < 0 || L < 1 || C % 16 || N % XP_TN || C < 16 || N < XP_TN || ldy < N) return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const long M = (long)rows * L;
  if (M >= 0x7fffffffl) return DA_EINVAL;
  ConvX3pArgs a;
  a.x = reinterpret_cast<const __bf16*>(x); a.w = reinterpret_cast<const __bf16*>(wpk); a.y = y;
  a.M = (int)M; a.L = L; a.C = C; a.ldy = ldy; a.N = N; a.accumulate = accumulate;
  a.divL = make_fastdiv((uint32_t)L);
  const int ntn = N / XP_TN;
  const int g_kernel = XP_DEFAULT_KERNEL;
  const long mtiles = (M + XP_TM - 1) / XP_TM;
  const long tiles = mtiles * ntn;
  if (tiles > 0x3fffffffl) return DA_EINVAL;
  // the partly filled last round (256 resident blocks: 1 per CU) runs as 64 x 64 tiles -- whole M-tile rows of them
  long tail_m = 0;
  const int g_tail = 1;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_x3p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            XP_LDS_BYTES) != hipSuccess)
      return DA_EINVAL;
    attr_set = true;
  }
  if (g_tail) tail_m = tiles < 256 ? mtiles : (tiles % 256) / ntn;
  a.tail_m = (int)tail_m;
  a.full_m = (int)(mtiles - tail_m);
  const long blocks = (long)a.full_m * ntn + 4l * tail_m * ntn;
  if (g_kernel == 4) {
    static bool attr4 = false;
    if (!attr4) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_x3p_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              XD_LDS_BYTES) != hipSuccess)
        return DA_EINVAL;
      attr4 = true;
    }
    hipLaunchKernelGGL(conv3_x3p_dma_kernel, dim3((unsigned)blocks), dim3(512), XD_LDS_BYTES, stream, a);
    DA_CHECK_LAUNCH();
    return DA_OK;
  }
  hipLaunchKernelGGL(conv3_x3p_kernel, dim3((unsigned)blocks), dim3(512), XP_LDS_BYTES, stream, a);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_x3_split(const float* x, int ld, void* out, size_t npos, int C, hipStream_t stream) {
  DA_ENTER();
  if (!x || !out || C % 16 || ld < C || ld % 4) return DA_EINVAL;
  if (npos == 0) return DA_OK;
  size_t g = (npos * (C >> 2) + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, ld, reinterpret_cast<__bf16*>(out), npos, C);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int da_x3_merge(const void* x, float* out, int ld, size_t npos, int C, hipStream_t stream) {
  DA_ENTER();
  if (!x || !out || C % 16 || ld < C || ld % 4) return DA_EINVAL;
  if (npos == 0) return DA_OK;
  size_t g = (npos * (C >> 2) + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(x3_merge_kernel, dim3((unsigned)g), dim3(256), 0, stream, reinterpret_cast<const __bf16*>(x), out, ld, npos, C);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

}  // extern "C"
