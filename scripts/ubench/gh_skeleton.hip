// Micro-benchmark (round 4): does a wave that ONLY multiplies -- v_mfma_f32_16x16x32_f16 x 3 per product, its weight
// fragments from a quarter ring in LDS, one s_barrier per 16 KiB quarter -- sustain the matrix pipe when ANOTHER wave
// of the same SIMD does everything that touches memory (the LDS-DMA of the weight stream, further global loads,
// VALU work)?  Skeleton of a "GEMM waves + helper waves" workgroup: 512 threads, waves 0-3 multiply (one per SIMD),
// waves 4-7 help (one per SIMD; a workgroup's waves w and w + 4 share a SIMD).  Against it: the shipped structure
// in one-workgroup-per-CU form (four waves, each issuing its own DMA pieces).
//
//   hipcc --offload-arch=gfx950 -O3 gh_skeleton.hip -o /tmp/gh_skeleton && /tmp/gh_skeleton
//
// Prints shader cycles per quarter (24 MFMAs per multiplying wave: 384 cycles of pipe time) and ns per quarter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f4 mm(u4 a, u4 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int kQFloats = 4096;        // one quarter: 8 n-blocks x (hi, lo) x 1 KiB = 16 KiB
constexpr int kStream = 128;          // quarters in the weight image (2 MiB, L2-resident)

// one 1 KiB piece: wave-uniform LDS base in m0, per-lane global address
__device__ __forceinline__ void piece(const float* gsrc, float* lds, int p, int lane) {
  const unsigned m0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>(lds + p * 256)));
  const float* src = gsrc + p * 256 + lane * 4;
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(src) : "memory", "m0");
}

// One quarter of the multiplying wave: 8 n-blocks x 3 MFMAs in groups of NG n-blocks, the next group's fragment
// reads behind this group's MFMAs (hi fragments first), the next QUARTER's first group behind the barrier.
template <int NG>
__device__ __forceinline__ void g_quarter(f4 (&acc)[32], int nb0, const u4* wb, const u4* wb_next, u4 (&fh)[NG],
                                          u4 (&fl)[NG], u4 bh, u4 bl) {
  constexpr int kGroups = 8 / NG;
#pragma unroll
  for (int T = 0; T < kGroups; ++T) {
    const bool last = T + 1 == kGroups;
    u4 nh[NG], nl[NG];
    if (last) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < NG; ++q) acc[nb0 + NG * T + q] = mm(fh[q], bh, acc[nb0 + NG * T + q]);
      FENCE();
      asm volatile("s_barrier" ::: "memory");      // the next quarter is published; this one's buffer is free
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        acc[nb0 + NG * T + q] = mm(fh[q], bl, acc[nb0 + NG * T + q]);
        nh[q] = wb_next[q * 128];
        FENCE();
      }
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        acc[nb0 + NG * T + q] = mm(fl[q], bh, acc[nb0 + NG * T + q]);
        nl[q] = wb_next[q * 128 + 64];
        FENCE();
      }
    } else {
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        acc[nb0 + NG * T + q] = mm(fh[q], bh, acc[nb0 + NG * T + q]);
        nh[q] = wb[(NG * (T + 1) + q) * 128];
        FENCE();
      }
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        acc[nb0 + NG * T + q] = mm(fh[q], bl, acc[nb0 + NG * T + q]);
        nl[q] = wb[(NG * (T + 1) + q) * 128 + 64];
        FENCE();
      }
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        acc[nb0 + NG * T + q] = mm(fl[q], bh, acc[nb0 + NG * T + q]);
        FENCE();
      }
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      fh[q] = nh[q];
      fl[q] = nl[q];
    }
  }
}

// HELPERS = 1: waves 4-7 stage the ring (+ XV extra 16-byte global loads and XA VALU ops per quarter each);
// HELPERS = 0: four waves, each multiplies AND stages its share of every quarter (the shipped structure, one
// workgroup per CU).  RING = quarters in LDS (DMA runs RING - 1 quarters ahead).
// L2LIKE = 1: the multiplying wave's loop shaped like the real kernel's layer-2 pass -- 16 K steps unrolled, two quarters
// each into accumulator sets 0-7 / 8-15, the B operand of K step cc taken from register arrays hh[cc], hl[cc] (128
// VGPRs of packed hidden values) -- instead of four quarters with one B operand: does the unrolled 32-quarter body,
// or the register pressure, cost the ~110 cycles per quarter the real kernel lies above this skeleton?
template <int HELPERS, int NG, int RING, int XV, int XA, int L2LIKE = 0>
__global__ __launch_bounds__(HELPERS ? 512 : 256, HELPERS ? 2 : 1) void gh(const float* __restrict__ w,
                                                                           const f4* __restrict__ extra, f4* out,
                                                                           long long* cyc, int n_quarters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool helper = HELPERS && wave >= 4;
  const int hw = wave & 3;
  auto src = [&](int q) { return w + (size_t)(q % kStream) * kQFloats; };
  auto buf = [&](int q) { return smem + (q % RING) * kQFloats; };
  // prologue: quarters 0 .. RING - 2 in flight (4 pieces per staging wave and quarter)
  if (helper || !HELPERS) {
    for (int q = 0; q < RING - 1; ++q)
#pragma unroll
      for (int p = 0; p < 4; ++p) piece(src(q), buf(q), 4 * hw + p, lane);
  }
  if (helper) {
    f4 v = f4{1.f, 2.f, 3.f, 4.f}, v2 = v * 0.5f, v3 = v * 0.25f, v4 = v * 0.125f, ld[XV > 0 ? XV : 1];
#pragma unroll
    for (int k = 0; k < (XV > 0 ? XV : 1); ++k) ld[k] = v;
    const f4* ep = extra + (size_t)blockIdx.x * 4096 + tid;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int q = 0; q < n_quarters; ++q) {
      // quarter q has landed: everything but what was issued for the (RING - 2) younger quarters (the first
      // iterations, whose younger quarters came from the prologue without extras: everything)
      if (q < RING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * (4 + XV)) : "memory");
      asm volatile("s_barrier" ::: "memory");
#pragma unroll
      for (int p = 0; p < 4; ++p) piece(src(q + RING - 1), buf(q + RING - 1), 4 * hw + p, lane);
#pragma unroll
      for (int k = 0; k < XV; ++k)
        // ("+v": the destination registers stay THESE registers for the whole loop -- the compiler believes an asm
        //  output is there when the statement ends and would hand a dead one to the next DMA address; the
        //  load lands whenever it lands)
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(ld[k]) : "v"(ep + ((q * XV + k) & 15) * 256) : "memory");
#pragma unroll
      for (int k = 0; k < XA; k += 4) {        // XA x 4 VALU ops per quarter in four independent chains
        v = v * v.yzwx + v.wxyz;
        v2 = v2 * v2.yzwx + v2.wxyz;
        v3 = v3 * v3.yzwx + v3.wxyz;
        v4 = v4 * v4.yzwx + v4.wxyz;
        asm volatile("" : "+v"(v), "+v"(v2), "+v"(v3), "+v"(v4));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < XV; ++k) v += ld[k];
    out[(size_t)blockIdx.x * 512 + tid] = v + v2 + v3 + v4;
    if (tid == 256) cyc[2 * blockIdx.x + 1] = t1 - t0;
    return;
  }
  // ---- multiplying wave
  f4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  const u4 bh = reinterpret_cast<const u4*>(w)[lane + 64 * wave], bl = reinterpret_cast<const u4*>(w)[lane + 64 * wave + 256];
  u4 fh[NG], fl[NG];
  if (!HELPERS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * 4) : "memory");
  asm volatile("s_barrier" ::: "memory");            // quarter 0 is there (helpers: their first loop barrier)
  {
    const u4* wb = reinterpret_cast<const u4*>(buf(0)) + lane;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      fh[q] = wb[q * 128];
      fl[q] = wb[q * 128 + 64];
    }
  }
  if constexpr (L2LIKE) {
    u4 hh[16], hl[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      hh[c] = reinterpret_cast<const u4*>(w)[lane + 64 * ((wave + c) & 7) + 512];
      hl[c] = reinterpret_cast<const u4*>(w)[lane + 64 * ((wave + c) & 7) + 1024];
    }
    const long long t0 = __builtin_readcyclecounter();
    // (n_quarters = 32 k + 1)
#pragma unroll 1
    for (int q0 = 0; q0 + 32 < n_quarters; q0 += 32) {
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int q = q0 + 2 * cc + j;
          const u4* wb = reinterpret_cast<const u4*>(smem + ((2 * cc + j) % RING) * kQFloats) + lane;
          const u4* wn = reinterpret_cast<const u4*>(smem + ((2 * cc + j + 1) % RING) * kQFloats) + lane;
          (void)q;
          g_quarter<NG>(acc, 8 * j, wb, wn, fh, fl, hh[cc], hl[cc]);
        }
      }
    }
    const long long t1 = __builtin_readcyclecounter();
    f4 s2 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) s2 += acc[i];
    out[(size_t)blockIdx.x * 512 + tid] = s2;
    if (tid == 0) cyc[2 * blockIdx.x] = t1 - t0;
    return;
  }
  const long long t0 = __builtin_readcyclecounter();
  // (n_quarters = 4 k + 1: k rounds over the four accumulator groups, the accumulator index a compile-time constant)
#pragma unroll 1
  for (int q0 = 0; q0 + 4 < n_quarters; q0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = q0 + j;
      const u4* wb = reinterpret_cast<const u4*>(buf(q)) + lane;
      const u4* wn = reinterpret_cast<const u4*>(buf(q + 1)) + lane;
      if (!HELPERS) {
        // own staging: the pieces of quarter q + RING - 1 go out behind the barrier of quarter q - 1, i.e. here, into
        // the buffer quarter q - 1 has left (q = 0: the one buffer the prologue did not fill); the counted wait for
        // quarter q + 1 (younger: quarters q + 2 .. q + RING - 1) sits in front of this quarter's barrier
#pragma unroll
        for (int p = 0; p < 4; ++p) piece(src(q + RING - 1), buf(q + RING - 1), 4 * hw + p, lane);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * 4) : "memory");
      }
      g_quarter<NG>(acc, 8 * j, wb, wn, fh, fl, bh, bl);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (!HELPERS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f4 s = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[(size_t)blockIdx.x * 512 + tid] = s;
  if (tid == 0) cyc[2 * blockIdx.x] = t1 - t0;
}

template <int HELPERS, int NG, int RING, int XV, int XA, int L2LIKE = 0>
void run(const char* name, int blocks, const float* w, const f4* extra, f4* out, long long* cyc) {
  const int nq = 4097;                     // (= 32 * 128 + 1 = 4 * 1024 + 1: fits both loop shapes)
  const size_t lds = RING * kQFloats * sizeof(float);
  auto fn = gh<HELPERS, NG, RING, XV, XA, L2LIKE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int threads = HELPERS ? 512 : 256;
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), lds, 0, w, extra, out, cyc, 65);
  {
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(e)); exit(1); }
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), lds, 0, w, extra, out, cyc, nq);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(2 * blocks);
  hipMemcpy(h.data(), cyc, 2 * blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double g = 0, hh = 0;
  for (int b = 0; b < blocks; ++b) { g += h[2 * b]; hh += h[2 * b + 1]; }
  printf("%-58s blocks=%4d  %7.1f cyc/quarter (multiplying wave)  %7.1f (helper)  %6.1f ns/quarter  = %4.0f %% of the 384-cycle pipe time\n",
         name, blocks, g / blocks / (nq - 1), HELPERS ? hh / blocks / nq : 0.0, ms * 1e6 / nq, 100.0 * 384.0 * (nq - 1) / (g / blocks));
}

int main(int argc, char** argv) {
  float* w; f4* extra; f4* out; long long* cyc;
  const size_t wn = (size_t)kStream * kQFloats;
  hipMalloc(&w, wn * sizeof(float));
  {
    // non-trivial f16 operands (the clock under load depends on the data)
    std::vector<unsigned> h(wn);
    unsigned s = 12345u;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      const unsigned a = 0x3000u | ((s >> 8) & 0x0fffu), b = 0x3000u | ((s >> 20) & 0x0fffu);   // halves in [0.125, 0.5)
      v = a | (b << 16);
    }
    hipMemcpy(w, h.data(), wn * sizeof(float), hipMemcpyHostToDevice);
  }
  hipMalloc(&extra, (size_t)256 * 4096 * sizeof(f4) + 65536);
  hipMemset(extra, 0, (size_t)256 * 4096 * sizeof(f4) + 65536);
  hipMalloc(&out, (size_t)256 * 512 * sizeof(f4));
  hipMalloc(&cyc, 2 * 256 * sizeof(long long));
  hipMemset(cyc, 0, 2 * 256 * sizeof(long long));
  // one variant per process (argv[1]): a faulting variant loses only itself
  const int which = argc > 1 ? atoi(argv[1]) : -1;
  const int blocks = 256;
  setvbuf(stdout, nullptr, _IONBF, 0);
  switch (which) {
    case 0: run<0, 2, 4, 0, 0>("shipped structure, lone: own DMA, groups of 2, ring 4", blocks, w, extra, out, cyc); break;
    case 1: run<0, 4, 4, 0, 0>("own DMA, groups of 4, ring 4", blocks, w, extra, out, cyc); break;
    case 2: run<1, 2, 4, 0, 0>("helpers stage the ring; groups of 2, ring 4", blocks, w, extra, out, cyc); break;
    case 3: run<1, 4, 4, 0, 0>("helpers stage the ring; groups of 4, ring 4", blocks, w, extra, out, cyc); break;
    case 4: run<1, 4, 4, 1, 0>("helpers + 1 global load / quarter; g4, ring 4", blocks, w, extra, out, cyc); break;
    case 5: run<1, 4, 4, 2, 0>("helpers + 2 global loads / quarter", blocks, w, extra, out, cyc); break;
    case 6: run<1, 4, 4, 4, 0>("helpers + 4 global loads / quarter", blocks, w, extra, out, cyc); break;
    case 7: run<1, 4, 4, 0, 8>("helpers + 8 x 4 VALU / quarter", blocks, w, extra, out, cyc); break;
    case 8: run<1, 4, 4, 0, 16>("helpers + 16 x 4 VALU / quarter", blocks, w, extra, out, cyc); break;
    case 9: run<1, 4, 4, 0, 32>("helpers + 32 x 4 VALU / quarter", blocks, w, extra, out, cyc); break;
    case 10: run<1, 4, 4, 2, 16>("helpers + 2 loads + 16 x 4 VALU / quarter", blocks, w, extra, out, cyc); break;
    case 11: run<1, 4, 4, 2, 32>("helpers + 2 loads + 32 x 4 VALU / quarter", blocks, w, extra, out, cyc); break;
    case 12: run<1, 2, 4, 2, 16>("helpers + 2 loads + 16 x 4 VALU / quarter; g2", blocks, w, extra, out, cyc); break;
    case 13: run<1, 4, 6, 0, 0>("helpers; groups of 4, ring 6 (LDS offsets beyond 64 KiB)", blocks, w, extra, out, cyc); break;
    case 14: run<1, 2, 4, 0, 0, 1>("helpers; layer-2-like loop (32 quarters unrolled, B from 128 VGPRs); g2, ring 4", blocks, w, extra, out, cyc); break;
    case 15: run<0, 2, 4, 0, 0, 1>("own DMA; layer-2-like loop; g2, ring 4", blocks, w, extra, out, cyc); break;
    default: printf("usage: gh_skeleton <variant 0..15>\n");
  }
  return 0;
}
