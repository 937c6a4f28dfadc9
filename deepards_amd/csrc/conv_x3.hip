// fp32 convolutions on the bf16 matrix cores ("f32x3"): every fp32 operand is split EXACTLY into three bf16 terms
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)          (round-to-nearest-even, 3 x 8 bits)
// and a product x w is taken as the six bf16 products  h h' + h m' + m h' + h l' + l h' + m m'  (each exact in the
// MFMA's fp32 accumulate); the three dropped ones (m l', l m', l l') are below 2^-23 |x w|, the size of one fp32
// rounding.  On gfx950 the bf16 matrix rate is 16x the fp32 one (MI355X_MICROARCH.md: 2.5 PFLOP/s against 157 TFLOP/s),
// so six bf16 products cost 3/8 of one fp32 MFMA product -- less than Winograd F(2,3)'s 2/3 -- in DIRECT form, i.e.
// without Winograd's error growth: measured error against fp64 below the fp32 Winograd kernels' (tests/test_x3_gpu.py).
//
// Replaces the same nn.Conv1d calls as conv_wino.hip / conv_gemm.hip (reference models/resnet.py:5-8,27-38,126-128)
// when the host selects conv arithmetic 'f32x3'; storage, statistics and sums stay fp32.
//
// Block = 128 positions x 64 output channels, 4 waves of 64 x 32 (v_mfma_f32_32x32x16_bf16, two accumulators); K step
// = 16 channels.  Activations are split while being staged into LDS (once per block): panel rows are
// [h 16 ch | m 16 ch | l 16 ch | 16 B pad] = 112 bytes (7 16-byte slots: the 16 rows a ds_read_b128 lane group touches
// fall on 16 different slots).  The weight fragments never pass through LDS: da_pack_conv3_x3 stores them pre-split in
// the order the MFMA wants them -- [tap][N/32][C/16][term][lane 64][8 bf16], one coalesced 1 KB read per (tap, term)
// and wave -- and every wave fetches its nine fragments of the NEXT K step into registers while it multiplies the
// current ones.  The panel is double-buffered (2 x 131 rows x 112 B = 29 KB), so a K step has ONE barrier and the
// staging of step k+1 (VALU split, LDS stores) and the loads of step k+2 sit between step k's 36 MFMAs.
//
// Measured (scripts/bench_x3.py, B = 64): 8-16 % faster per launch than the fp32 Winograd kernels of conv_wino.hip, but
// no faster inside the captured step -- so 'f32' stays the default arithmetic and this one is opt-in.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define X3_TM 128
#define X3_TN 64
#define X3_PITCH 112
#define X3_XROWS (X3_TM + 2)

__device__ __forceinline__ int xcd_chunked_x3(int id, int total) {   // consecutive work items share an XCD (conv_gemm.hip)
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, s = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
}

__device__ __forceinline__ f32x2v x3_cvt4(const f32x4& v) {           // 4 bf16 (nearest-even) as the bits of 2 floats
  const f32x2v lo = {v[0], v[1]}, hi = {v[2], v[3]};
  const bf16x2 a = __builtin_convertvector(lo, bf16x2), b = __builtin_convertvector(hi, bf16x2);
  return f32x2v{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
}
__device__ __forceinline__ f32x4 x3_widen4(const f32x2v& b) {         // the 4 bf16 back as floats (exact)
  const uint32_t u0 = __float_as_uint(b[0]), u1 = __float_as_uint(b[1]);
  return f32x4{__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 << 16),
               __uint_as_float(u1 & 0xffff0000u)};
}
// v = h + m + l exactly for finite v (three 8-bit significands cover fp32's 24 bits)
__device__ __forceinline__ void x3_split4(const f32x4& v, f32x2v& h, f32x2v& m, f32x2v& l) {
  h = x3_cvt4(v);
  const f32x4 r1 = v - x3_widen4(h);
  m = x3_cvt4(r1);
  const f32x4 r2 = r1 - x3_widen4(m);
  l = x3_cvt4(r2);
}

struct ConvX3Args {
  const float* x;       // [M][ldx] activations, first C channels
  const __bf16* w;      // [3 taps][N / 32][C / 16][3 terms][64 lanes][8] from da_pack_conv3_x3
  float* y;             // [M][ldy], first N channels
  int M, L, ldx, C, ldy, N, accumulate;
  FastDiv divL;
};

#define X3P_ROWS (X3_XROWS + 1)                       // + one row of zeros (what a tap reads across a sequence edge)
#define X3P_XBYTES (X3P_ROWS * X3_PITCH)

struct X3BFrag {
  f32x4 v[9];           // [tap][term]: 8 bf16 each, as bits
};

__global__ __launch_bounds__(256, 2) void conv3_x3_kernel(ConvX3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * X3P_XBYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = a.N / X3_TN;
  const int tile = xcd_chunked_x3(blockIdx.x, gridDim.x);
  const int P0 = (tile / ntn) * X3_TM, n_blk = (tile % ntn) * X3_TN;

  // panel loader.  Rows outside [0, M) are loaded from a clamped (valid) address instead of being zeroed: no output
  // that is stored ever reads them (a tap that would reads the zero row, below).
  const int xq = tid & 3, xm = tid >> 2;
  const int xrow = 8 * (xm >> 3) + ((xm >> 2) & 1) + 2 * (xm & 3);
  const float* xsrc[3];
  int xdst[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int r = p < 2 ? p * 64 + xrow : 128 + ((tid >> 2) & 1);          // pass 2: rows 128, 129 (first 8 threads)
    long P = (long)P0 - 1 + r;
    P = P < 0 ? 0 : (P >= a.M ? a.M - 1 : P);
    xsrc[p] = a.x + P * a.ldx + xq * 4;
    xdst[p] = r * X3_PITCH + xq * 8;
  }
  const int kch = a.C >> 4;
  f32x4 rx[3];
  auto gload_x = [&](int ks) {
#pragma unroll
    for (int p = 0; p < 2; ++p) rx[p] = *reinterpret_cast<const f32x4*>(xsrc[p] + (ks << 4));
    if (tid < 8) rx[2] = *reinterpret_cast<const f32x4*>(xsrc[2] + (ks << 4));
  };
  auto stage_x = [&](unsigned char* Xs, int p) {
    f32x2v h, m, l;
    x3_split4(rx[p], h, m, l);
    unsigned char* d = Xs + xdst[p];
    *reinterpret_cast<f32x2v*>(d) = h;
    *reinterpret_cast<f32x2v*>(d + 32) = m;
    *reinterpret_cast<f32x2v*>(d + 64) = l;
  };

  const int frow = lane & 31, kg = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // fragment pack: ((((t * (N/32) + nb) * kch + ks) * 3 + term) * 64 + lane) * 8 bf16
  const size_t nb = (size_t)(n_blk >> 5) + wn;
  const __bf16* wbase = a.w + (nb * kch * 3 * 64 + lane) * 8;
  const size_t wtap = (size_t)(a.N >> 5) * kch * 3 * 64 * 8;
  auto gload_b = [&](X3BFrag& b, int ks) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        b.v[t * 3 + s] = *reinterpret_cast<const f32x4*>(wbase + t * wtap + ((size_t)ks * 3 + s) * 512);
  };

  // LDS offset of the A fragment of (row tile mt, tap t): the panel row of position + t - 1, or the zero row when that
  // position lies across a sequence edge
  int aoff[2][3];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const long P = (long)P0 + wm * 64 + mt * 32 + frow;
    const uint32_t Pc = (uint32_t)(P < a.M ? P : 0);
    const int l = (int)(Pc - fdiv(Pc, a.divL) * (uint32_t)a.L);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bool edge = (t == 0 && l == 0) || (t == 2 && l == a.L - 1);
      aoff[mt][t] = (edge ? X3_XROWS : wm * 64 + mt * 32 + t + frow) * X3_PITCH + kg * 16;
    }
  }

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // one K step: multiply buffer `cur` with bc; meanwhile fetch the next step's fragments into bn, stage the panel of
  // step ks + 1 (already in rx) into `nxt` and load the panel of step ks + 2 into rx.  Branch-free (the last steps
  // re-load and re-stage the last panel) so that the compiler can place all of it between the MFMAs.
  auto step = [&](const unsigned char* cur, unsigned char* nxt, const X3BFrag& bc, X3BFrag& bn, int ks) {
    const int k1 = ks + 1 < kch ? ks + 1 : kch - 1, k2 = ks + 2 < kch ? ks + 2 : kch - 1;
    gload_b(bn, k1);                                     // (the compiler sinks these loads to the end of the step: 122 registers, 4 waves per SIMD;
                                                         //  pinning them here -- a whole K step ahead -- costs 158 registers and measured 3 % slower)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 av[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) av[s] = *reinterpret_cast<const f32x4*>(cur + aoff[mt][t] + s * 32);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, av[0]), am = __builtin_bit_cast(bf16x8, av[1]),
                     al = __builtin_bit_cast(bf16x8, av[2]);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bc.v[t * 3]), bm = __builtin_bit_cast(bf16x8, bc.v[t * 3 + 1]),
                     bl = __builtin_bit_cast(bf16x8, bc.v[t * 3 + 2]);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mt], 0, 0, 0);      // small terms first
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mt], 0, 0, 0);
      }
      if (t < 2) stage_x(nxt, t);
    }
    if (tid < 8) stage_x(nxt, 2);
    gload_x(k2);
    __syncthreads();
  };

  X3BFrag b0, b1;
  gload_x(0);
  gload_b(b0, 0);
  if (tid < 28) {                                         // the zero rows of both buffers
    const int bsel = tid >= 14, piece = tid - 14 * bsel;
    *reinterpret_cast<f32x2v*>(lds + bsel * X3P_XBYTES + X3_XROWS * X3_PITCH + piece * 8) = f32x2v{0.f, 0.f};
  }
  stage_x(lds, 0);
  stage_x(lds, 1);
  if (tid < 8) stage_x(lds, 2);
  gload_x(kch > 1 ? 1 : 0);
  __syncthreads();
  for (int ks = 0; ks < kch; ks += 2) {
    step(lds, lds + X3P_XBYTES, b0, b1, ks);
    step(lds + X3P_XBYTES, lds, b1, b0, ks + 1);          // kch is even (C % 32 == 0)
  }

#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long P = (long)P0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (P < a.M) {
        float* o = a.y + P * a.ldy + n_blk + wn * 32 + frow;
        float v = acc[mt][r];
        if (a.accumulate) v += *o;
        *o = v;
      }
    }
}

// the fragment-major packs: element (t, n, c) term s sits at
//     ((((t * (N/32) + n/32) * (C/16) + c/16) * 3 + s) * 64 + (c%16 / 8) * 32 + n%32) * 8 + c%8
// wf: n = co, c = ci, tap t of the forward conv;  wd: n = ci, c = co, tap 2 - t
__global__ __launch_bounds__(256) void pack_conv3_x3_kernel(const float* __restrict__ w, __bf16* __restrict__ wf,
                                                             __bf16* __restrict__ wd, int co, int ci) {
  const size_t total = (size_t)co * ci;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int o = (int)(idx / ci), i = (int)(idx - (size_t)o * ci);
  const float* src = w + idx * 3;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const float v = src[t];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    if (wf) {
      __bf16* d = wf + ((((size_t)t * (co >> 5) + (o >> 5)) * (ci >> 4) + (i >> 4)) * 3 * 64 + ((i & 15) >> 3) * 32 + (o & 31)) * 8 + (i & 7);
      d[0] = h; d[512] = m; d[1024] = l;
    }
    if (wd) {
      __bf16* d = wd + ((((size_t)(2 - t) * (ci >> 5) + (i >> 5)) * (co >> 4) + (o >> 4)) * 3 * 64 + ((o & 15) >> 3) * 32 + (i & 31)) * 8 + (o & 7);
      d[0] = h; d[512] = m; d[1024] = l;
    }
  }
}

extern "C" {

// y (+)= conv1d(x, k = 3, stride 1, pad 1) per row of L positions with fp32-equivalent split-bf16 products (see the
// head of this file).  x: [rows][L][ldx] fp32 (first C channels), wpk from da_pack_conv3_x3, y: [rows][L][ldy] fp32
// (first N channels).  C % 32 == 0, N % 64 == 0.
// replaces reference models/resnet.py:5-8 (conv2x2), forward and (with the wd pack) data gradient
int da_conv3_x3(const float* x, const void* wpk, float* y, int rows, int L, int ldx, int C, int ldy, int N,
                 int accumulate, hipStream_t stream) {
  DA_ENTER();
  if (!x || !wpk || !y || rows < 0 || L < 1 || C % 32 || N % X3_TN || C < 32 || N < X3_TN || ldx % 4 || ldx < C || ldy < N)
    return DA_EINVAL;
  if (rows == 0) return DA_OK;
  const long M = (long)rows * L;
  if (M >= 0x7fffffffl) return DA_EINVAL;
  ConvX3Args a;
  a.x = x; a.w = reinterpret_cast<const __bf16*>(wpk); a.y = y;
  a.M = (int)M; a.L = L; a.ldx = ldx; a.C = C; a.ldy = ldy; a.N = N; a.accumulate = accumulate;
  a.divL = make_fastdiv((uint32_t)L);
  const long tiles = ((M + X3_TM - 1) / X3_TM) * (N / X3_TN);
  if (tiles > 0x7fffffffl) return DA_EINVAL;
  hipLaunchKernelGGL(conv3_x3_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, a);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// split-bf16 fragment packs of one (Co, Ci, 3) fp32 conv weight: wf (forward), wd (data gradient: channels swapped,
// taps reversed); either may be NULL; 3 * 3 * Co * Ci bf16 each.  Co, Ci multiples of 32.
int da_pack_conv3_x3(const float* w, void* wf, void* wd, int co, int ci, hipStream_t stream) {
  DA_ENTER();
  if (!w || (!wf && !wd) || co < 32 || ci < 32 || co % 32 || ci % 32) return DA_EINVAL;
  const size_t total = (size_t)co * ci;
  hipLaunchKernelGGL(pack_conv3_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w,
                     reinterpret_cast<__bf16*>(wf), reinterpret_cast<__bf16*>(wd), co, ci);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

}  // extern "C"
