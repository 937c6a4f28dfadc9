// microbenchmark: fp32 MFMA issue rate vs waves/SIMD and accumulator count, with/without LDS reads
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int LDSREAD>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[64 * 36 * 2];
  for (int i = threadIdx.x; i < 64 * 36 * 2; i += 256) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = threadIdx.x * 1e-3f, bv = 1.0f;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    if (LDSREAD) {
#pragma unroll
      for (int c8 = 0; c8 < 4; ++c8) {
        f32x4 af = *reinterpret_cast<const f32x4*>(&lds[(lane & 31) * 36 + c8 * 8 + (lane >> 5) * 4]);
        f32x4 bf = *reinterpret_cast<const f32x4*>(&lds[64 * 36 + (lane & 31) * 36 + c8 * 8 + (lane >> 5) * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[(c8 * 4 + e) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc[(c8 * 4 + e) % NACC], 0, 0, 0);
      }
      if (LDSREAD == 2) __syncthreads();
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[e % NACC], 0, 0, 0);
    }
  }
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// mimic of the conv K-step: MODE bit0: double barrier + LDS staging writes, bit1: 4 global float4 loads per step
template <int MODE>
__global__ __launch_bounds__(256) void kc(float* out, const float* __restrict__ src, int iters, size_t span) {
  __shared__ float lds[2 * 128 * 36];
  for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lane = threadIdx.x & 63, lr = threadIdx.x >> 3, lq = threadIdx.x & 7;
  const int wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  f32x4 ra[2], rb[2], ra2[2], rb2[2];
  for (int p = 0; p < 2; ++p) { ra[p] = f32x4{1, 2, 3, 4}; rb[p] = f32x4{1, 1, 1, 1}; ra2[p] = ra[p]; rb2[p] = rb[p]; }
  size_t base = ((size_t)blockIdx.x * 64 + lr) * 64 + lq * 4;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
      __syncthreads();
      for (int p = 0; p < 2; ++p) {
        *reinterpret_cast<f32x4*>(&lds[(lr + 32 * p) * 36 + lq * 4]) = ra[p];
        *reinterpret_cast<f32x4*>(&lds[64 * 36 + (lr + 32 * p) * 36 + lq * 4]) = rb[p];
      }
      __syncthreads();
    }
    if ((MODE & 2) && !(MODE & 4) && !(MODE & 8)) {
      size_t o = (base + (size_t)it * 32) & (span - 1);
      for (int p = 0; p < 2; ++p) {
        ra[p] = *reinterpret_cast<const f32x4*>(src + o + p * 2048);
        rb[p] = *reinterpret_cast<const f32x4*>(src + ((o + 4096 + p * 2048) & (span - 1)));
      }
    }
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      f32x4 af = *reinterpret_cast<const f32x4*>(&lds[(wm * 32 + (lane & 31)) * 36 + c8 * 8 + (lane >> 5) * 4]);
      f32x4 bf = *reinterpret_cast<const f32x4*>(&lds[64 * 36 + (wn * 32 + (lane & 31)) * 36 + c8 * 8 + (lane >> 5) * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
    }
    if ((MODE & 2) && (MODE & 4)) {   // loads issued after the MFMAs
      size_t o = (base + (size_t)it * 32) & (span - 1);
      for (int p = 0; p < 2; ++p) {
        ra[p] = *reinterpret_cast<const f32x4*>(src + o + p * 2048);
        rb[p] = *reinterpret_cast<const f32x4*>(src + ((o + 4096 + p * 2048) & (span - 1)));
      }
    }
    if (MODE & 16) {  // second register stage: what was loaded one step ago moves up, new loads go two steps ahead
      size_t o = (base + (size_t)it * 32) & (span - 1);
      for (int p = 0; p < 2; ++p) {
        ra[p] = ra2[p]; rb[p] = rb2[p];
        ra2[p] = *reinterpret_cast<const f32x4*>(src + o + p * 2048);
        rb2[p] = *reinterpret_cast<const f32x4*>(src + ((o + 4096 + p * 2048) & (span - 1)));
      }
    }
    if (MODE & 8) {   // async global->LDS (no VGPR staging), linear layout, other half of the LDS
      size_t o = (base + (size_t)it * 32) & (span - 1);
      float* dst = lds + 128 * 36 + wave * 1024;   // wave-uniform base; lane*16B appended by hardware
      for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_global_load_lds((const void*)(src + ((o + p * 2048) & (span - 1))), (__attribute__((address_space(3))) void*)(dst + p * 256), 16, 0, 0);
    }
  }
  float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s + ra[0][0] + rb[1][3] + ra2[0][0] + rb2[1][1];
}
// producer/consumer specialisation: NLOAD loader waves move global -> regs -> LDS[(it+1)&1], waves 0-3 only do
// LDS reads + MFMA on LDS[it&1]; one barrier per step; consumers never touch vector memory.
template <int NLOAD>
__global__ __launch_bounds__(256 + 64 * NLOAD) void kspec(float* out, const float* __restrict__ src, int iters, size_t span) {
  __shared__ float lds[2 * 128 * 36];
  for (int i = threadIdx.x; i < 2 * 128 * 36; i += blockDim.x) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int PER = 16 / NLOAD;          // float4 loads per loader lane per step (16 KB per block-step)
  f32x4 st[PER];
  for (int p = 0; p < PER; ++p) st[p] = f32x4{1, 2, 3, 4};
  size_t base = ((size_t)blockIdx.x * 64) * 64;
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  for (int it = 0; it < iters; ++it) {
    float* buf = lds + (it & 1) * 128 * 36;
    float* nbuf = lds + ((it + 1) & 1) * 128 * 36;
    if (wave >= 4) {
      const int lw = wave - 4;
      for (int p = 0; p < PER; ++p) {
        int row = (lw * PER + p) * 8 + (lane >> 3), q = lane & 7;
        *reinterpret_cast<f32x4*>(&nbuf[row * 36 + q * 4]) = st[p];
      }
      size_t o = (base + (size_t)it * 32) & (span - 1);
      for (int p = 0; p < PER; ++p) st[p] = *reinterpret_cast<const f32x4*>(src + ((o + (size_t)(lw * PER + p) * 512 + lane * 4) & (span - 1)));
    } else {
#pragma unroll
      for (int c8 = 0; c8 < 4; ++c8) {
        f32x4 af = *reinterpret_cast<const f32x4*>(&buf[(wm * 32 + (lane & 31)) * 36 + c8 * 8 + (lane >> 5) * 4]);
        f32x4 bf = *reinterpret_cast<const f32x4*>(&buf[64 * 36 + (wn * 32 + (lane & 31)) * 36 + c8 * 8 + (lane >> 5) * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + st[0][0] + st[PER - 1][3];
}
template <int NLOAD>
void runspec(int blocks_per_cu, size_t span_floats) {
  float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
  float* src; hipMalloc(&src, span_floats * 4 + (1 << 20)); hipMemset(src, 0, span_floats * 4 + (1 << 20));
  int iters = 2000, grid = 256 * blocks_per_cu, th = 256 + 64 * NLOAD;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kspec<NLOAD>), dim3(grid), dim3(th), 0, 0, out, src, iters, span_floats);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kspec<NLOAD>), dim3(grid), dim3(th), 0, 0, out, src, iters, span_floats);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 16 * 4096.0;
  printf("conv-mimic: SPECIALISED %d loader waves         blocks/CU %d: %.3f ms  %.1f TF/s\n", NLOAD, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out); hipFree(src);
}
template <int MODE>
void runc(const char* name, int blocks_per_cu, size_t span_floats) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  float* src; hipMalloc(&src, span_floats * 4 + (1 << 20)); hipMemset(src, 0, span_floats * 4 + (1 << 20));
  int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kc<MODE>), dim3(grid), dim3(256), 0, 0, out, src, iters, span_floats);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kc<MODE>), dim3(grid), dim3(256), 0, 0, out, src, iters, span_floats);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 16 * 4096.0;
  printf("%-44s blocks/CU %d: %.3f ms  %.1f TF/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out); hipFree(src);
}
template <int NACC, int LDSREAD>
void run(const char* name, int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, LDSREAD>), dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, LDSREAD>), dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 16 * 4096.0;
  printf("%-28s blocks/CU %d: %.3f ms  %.1f TF/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int b : {1, 2, 4}) { run<1, 0>("1 acc, regs", b); run<2, 0>("2 acc, regs", b); run<4, 0>("4 acc, regs", b); }
  for (int b : {1, 2, 4}) { run<1, 1>("1 acc, LDS b128 reads", b); run<2, 1>("2 acc, LDS reads", b); run<1, 2>("1 acc, LDS reads + barrier", b); }
  for (int b : {2, 4}) { runspec<1>(b, 1 << 22); runspec<2>(b, 1 << 22); runspec<4>(b, 1 << 22); }
  for (int b : {2, 4}) {
    runc<0>("conv-mimic: LDS reads only", b, 1 << 22);
    runc<1>("conv-mimic: + 2 barriers + staging writes", b, 1 << 22);
    runc<3>("conv-mimic: + 4 global loads (16 MB, L2)", b, 1 << 22);
    runc<7>("conv-mimic: loads AFTER the MFMAs (L2)", b, 1 << 22);
    runc<2>("conv-mimic: loads, no barriers/staging", b, 1 << 22);
    runc<17>("conv-mimic: staging + DISTANCE-2 reg prefetch", b, 1 << 22);
    runc<8>("conv-mimic: glds x4, no staging, no barrier", b, 1 << 22);
    runc<9>("conv-mimic: barriers+staging + glds x4 (L2)", b, 1 << 22);
  }
  return 0;
}
