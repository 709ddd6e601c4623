// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_f16 from ONE wave per SIMD in the access
// patterns the split-f16 row-MLP kernel uses.  Prints shader cycles per MFMA (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f4 mm(u4 a, u4 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f4 mm32(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// MODE 0: 32 accumulators, groups of 4, (hh x4, lh x4, hl x4) -- the kernel's order, operands in regs
// MODE 1: same but groups of 8 accumulators (dependent MFMAs 8 slots apart)
// MODE 2: 32 accumulators, every MFMA on a different accumulator round-robin (no near dependence)
// MODE 3: MODE 0 + 8 ds_read_b128 per group (fragments from LDS, prefetched one group ahead)
// MODE 4: fp32 16x16x4 reference: 32 accumulators round robin
// MODE 5/6/7: MODE 0 with 1 / 2 / 3 independent VALU ops (v_fma) wedged after EVERY MFMA
// MODE 8: MODE 0 with one transcendental (v_exp) + one v_fma after every second MFMA
// MODE 9: MODE 3's LDS fragment traffic, but ONE ds_read_b128 behind each of the first 8 MFMAs of a
//         group (hi fragments first) instead of a burst of 8 in front of the MFMAs
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const u4* __restrict__ src, f4* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  u4* lds = reinterpret_cast<u4*>(smem);
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
  __syncthreads();
  f4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  u4 ah[8], al[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ah[i] = src[lane + 64 * i]; al[i] = src[lane + 64 * (i + 8)]; }
  const u4 bh = src[lane + 1024], bl = src[lane + 1100];
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int T = 0; T < 8; ++T) {
        if (MODE == 3) {
          u4 nh[4], nl[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * T + q] = mm(ah[q], bh, acc[4 * T + q]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) { nh[q] = lds[((T + 1) & 7) * 512 + q * 128 + lane]; nl[q] = lds[((T + 1) & 7) * 512 + q * 128 + 64 + lane]; }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * T + q] = mm(ah[q], bl, acc[4 * T + q]);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * T + q] = mm(al[q], bh, acc[4 * T + q]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) { ah[q] = nh[q]; al[q] = nl[q]; }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * T + q] = mm(ah[q], bh, acc[4 * T + q]);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * T + q] = mm(ah[q], bl, acc[4 * T + q]);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * T + q] = mm(al[q], bh, acc[4 * T + q]);
        }
      }
    } else if (MODE == 9) {
#pragma unroll
      for (int T = 0; T < 8; ++T) {
        u4 nh[4], nl[4];
        const u4* base = lds + ((T + 1) & 7) * 512 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[4 * T + q] = mm(ah[q], bh, acc[4 * T + q]);
          nh[q] = base[q * 128];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[4 * T + q] = mm(ah[q], bl, acc[4 * T + q]);
          nl[q] = base[q * 128 + 64];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[4 * T + q] = mm(al[q], bh, acc[4 * T + q]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { ah[q] = nh[q]; al[q] = nl[q]; }
      }
    } else if (MODE >= 5 && MODE <= 8) {
      float va = __builtin_bit_cast(float, bh.x), vb = __builtin_bit_cast(float, bl.x);
      float v0 = va, v1 = vb, v2 = va + 1.f;
#pragma unroll
      for (int T = 0; T < 8; ++T) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[4 * T + q] = mm(r == 2 ? al[q] : ah[q], r == 1 ? bl : bh, acc[4 * T + q]);
            if (MODE == 8) {
              if (q & 1) { v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_fmaf(v1, va, vb); }
            } else {
              v0 = __builtin_fmaf(v0, va, vb);
              if (MODE >= 6) v1 = __builtin_fmaf(v1, vb, va);
              if (MODE >= 7) v2 = __builtin_fmaf(v2, va, va);
            }
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      acc[0].x += v0 + v1 + v2;
    } else if (MODE == 1) {
#pragma unroll
      for (int T = 0; T < 4; ++T) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[8 * T + q] = mm(ah[q], bh, acc[8 * T + q]);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[8 * T + q] = mm(ah[q], bl, acc[8 * T + q]);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[8 * T + q] = mm(al[q], bh, acc[8 * T + q]);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int n = 0; n < 32; ++n) acc[n] = mm(r == 2 ? al[n & 7] : ah[n & 7], r == 1 ? bl : bh, acc[n]);
    } else {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int n = 0; n < 32; ++n)
          acc[n] = mm32(__builtin_bit_cast(float, ah[n & 7].x), __builtin_bit_cast(float, bh.x), acc[n]);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  f4 s = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, const u4* src, f4* out, long long* cyc) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 65536, 0, src, out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 65536, 0, src, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += v;
  mean /= blocks;
  const double n_mfma = 96.0 * iters;
  printf("%-34s blocks=%4d  %.1f memtime-ticks/MFMA  %.2f ns/MFMA  (kernel %.3f ms)\n", name, blocks,
         mean / n_mfma, ms * 1e6 / n_mfma, ms);
}

int main() {
  u4* src; f4* out; long long* cyc;
  hipMalloc(&src, 4096 * sizeof(u4));
  hipMemset(src, 0, 4096 * sizeof(u4));
  hipMalloc(&out, 1024 * 256 * sizeof(f4));
  hipMalloc(&cyc, 1024 * sizeof(long long));
  for (int blocks : {1, 256}) {
    run<0>("f16 groups of 4 (kernel order)", blocks, src, out, cyc);
    run<1>("f16 groups of 8", blocks, src, out, cyc);
    run<2>("f16 round-robin 32 accumulators", blocks, src, out, cyc);
    run<3>("f16 groups of 4 + LDS fragments", blocks, src, out, cyc);
    run<4>("f32 16x16x4 round-robin", blocks, src, out, cyc);
    run<5>("f16 g4 + 1 VALU after each MFMA", blocks, src, out, cyc);
    run<6>("f16 g4 + 2 VALU after each MFMA", blocks, src, out, cyc);
    run<7>("f16 g4 + 3 VALU after each MFMA", blocks, src, out, cyc);
    run<8>("f16 g4 + exp,fma per 2 MFMA", blocks, src, out, cyc);
    run<9>("f16 g4 + 1 ds_read behind each MFMA", blocks, src, out, cyc);
  }
  return 0;
}
