// Micro-benchmark for the "column-owner" tile formulation of the split-f16 row-MLP (DESIGN.md,
// what comes next): every wave owns 128 output columns of a 64-row tile, its weight fragments
// go global/L2 -> VGPR directly (no LDS staging, no cross-wave redundancy), the 64 rows' (hi, lo)
// fragments are read from LDS by all four waves, products are v_mfma_f32_32x32x16_f16.
//   per K=16 step and wave: 8 x 1 KiB weight fragments (global_load_dwordx4), 4 x ds_read_b128,
//   24 MFMAs (8 accumulator blocks x {hh, lh, hl}) = 768 MFMA cycles.
// Prints microseconds per K=32 of one 64x512 tile layer ("chunk-equivalent": 64 KiB of weights,
// 1536 MFMA cycles per SIMD) so the number is directly comparable with kernel_probe.py's
// kcycles_per_chunk of the shipped kernel.
//   hipcc --offload-arch=gfx950 -O3 colown_stream.hip -o /tmp/colown_stream && /tmp/colown_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#ifndef COLOWN_ASM_LOADS
#define COLOWN_ASM_LOADS 0
#endif

typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16v mm(u4 a, u4 b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// hipcc's waitcnt insertion gives up at the loop back edge (vmcnt(0) at the loop header), so the
// weight loads are inline asm and the waits are placed by hand, as in csrc/gcast.hip.
__device__ __forceinline__ u4 gload(const u4* p) {
#if COLOWN_ASM_LOADS
  u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p));
  return v;
#else
  return *p;
#endif
}
template <int N> __device__ __forceinline__ void wait_vm() {
#if COLOWN_ASM_LOADS
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
#endif
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int STEPS = 64;                 // K=16 steps per tile: two layers of K=512
constexpr int FRAG_U4 = 64;               // one fragment = 64 lanes x 16 B
constexpr int STEP_U4 = 4 * 8 * FRAG_U4;  // 4 waves x 8 fragments per step = 32 KiB

// WHAT bit0: stream the weights (else keep the first step's fragments)
//      bit1: issue the MFMAs
//      bit2: read the row fragments from LDS every step (else once)
template <int WHAT, int DEPTH>
__global__ __launch_bounds__(256, 1) void k(const u4* __restrict__ wimg, const u4* __restrict__ rows,
                                            float* out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u4* lds = reinterpret_cast<u4*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 64 rows x 512 k (hi, lo) halves, padded row stride 1040 B: [plane][row][1040 B]
  for (int i = threadIdx.x; i < 2 * 64 * 65; i += 256) lds[i] = rows[i & 4095];
  __syncthreads();
  f16v acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const u4* wbase = wimg + wave * 8 * FRAG_U4 + lane;
  u4 w[DEPTH][8];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
    for (int f = 0; f < 8; ++f) w[d][f] = gload(wbase + d * STEP_U4 + f * FRAG_U4);
  const int rrow = lane & 31, rk = lane >> 5;
  const u4* lrow = lds + rrow * 65 + rk;
  u4 b[2][4];
#pragma unroll
  for (int f = 0; f < 4; ++f) b[0][f] = lrow[(f & 1) * 64 * 65 + (f >> 1) * 32 * 65];
  for (int t = 0; t < tiles; ++t) {
    for (int s0 = 0; s0 < STEPS; s0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int s = s0 + d;
        const int sp = (s + DEPTH - 1) & (STEPS - 1);       // step whose weights are fetched now
        const int kn = ((s + 1) & 31) * 2;                  // next step's k offset of the row fragments
        if (!(WHAT & 4)) {
#pragma unroll
          for (int f = 0; f < 4; ++f) b[(d + 1) & 1][f] = b[d & 1][f];
        }
        // in flight here: the fragments of steps s .. s+DEPTH-2; step s must have landed
        wait_vm<8 * (DEPTH - 2)>();
        // 24 MFMAs: i -> product p = i / 8 (hh, lh, hl), column block c = (i % 8) / 2, row block r = i % 2.
        // One weight fragment load rides behind every third MFMA, the next step's four row
        // fragments behind MFMAs 12..15.
#pragma unroll
        for (int i = 0; i < 24; ++i) {
          const int p = i / 8, c = (i % 8) / 2, r = i % 2;
          if (WHAT & 2) {
            acc[2 * c + r] = mm(w[d][2 * c + (p == 2 ? 1 : 0)], b[d & 1][2 * r + (p == 1 ? 1 : 0)], acc[2 * c + r]);
          }
          if ((WHAT & 1) && i % 3 == 0)
            w[(d + DEPTH - 1) % DEPTH][i / 3] = gload(wbase + sp * STEP_U4 + (i / 3) * FRAG_U4);
          if ((WHAT & 4) && i >= 12 && i < 16) {
            const int f = i - 12;
            b[(d + 1) & 1][f] = lrow[(f & 1) * 64 * 65 + (f >> 1) * 32 * 65 + kn];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!(WHAT & 2)) {
#pragma unroll
          for (int f = 0; f < 8; ++f) asm volatile("" ::"v"(w[d][f]));
#pragma unroll
          for (int f = 0; f < 4; ++f) asm volatile("" ::"v"(b[d & 1][f]));
        }
      }
    }
  }
  wait_vm<0>();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static int g_tiles = 40;

template <int WHAT, int DEPTH>
void run(const char* name, int blocks, const u4* wimg, const u4* rows, float* out) {
  const int tiles = g_tiles;
  const size_t lds_bytes = 2 * 64 * 65 * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<WHAT, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)lds_bytes);
  hipLaunchKernelGGL((k<WHAT, DEPTH>), dim3(blocks), dim3(256), lds_bytes, 0, wimg, rows, out, 2);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WHAT, DEPTH>), dim3(blocks), dim3(256), lds_bytes, 0, wimg, rows, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double chunks = double(tiles) * STEPS / 2;
  const double us = best * 1e3 / chunks;
  printf("%-52s depth=%d blocks=%4d  %.3f us per K32 chunk-equivalent  (%.1f GB/s weights per CU, %4.0f%% of MFMA rate at 2.4 GHz)  kernel %.3f ms\n",
         name, DEPTH, blocks, us, (WHAT & 1) ? 65536.0 / us * 1e-3 : 0.0, (WHAT & 2) ? 100.0 * 0.64 / us : 0.0, best);
}

// random halves in (-1, 1): with all-zero operands the matrix cores barely toggle and the chip
// never reaches its power limit -- the sustained, data-dependent rate is what a real kernel sees
static void fill_random_halves(u4* dev, size_t n_u4, unsigned seed) {
  std::vector<unsigned short> h(n_u4 * 8);
  unsigned x = seed;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    const unsigned mant = (x >> 9) & 0x3ff, sign = (x >> 31) << 15, exp = 10 + ((x >> 19) & 3);   // 2^-5 .. 2^-2
    v = (unsigned short)(sign | (exp << 10) | mant);
  }
  hipMemcpy(dev, h.data(), n_u4 * sizeof(u4), hipMemcpyHostToDevice);
}

int main(int argc, char** argv) {
  u4 *wimg, *rows; float* out;
  const size_t wbytes = size_t(STEPS) * STEP_U4 * sizeof(u4);      // 2 MiB: W1 and W2 (hi, lo)
  hipMalloc(&wimg, wbytes);
  hipMemset(wimg, 0, wbytes);
  hipMalloc(&rows, 4096 * sizeof(u4));
  hipMemset(rows, 0, 4096 * sizeof(u4));
  hipMalloc(&out, 1024 * 256 * sizeof(float));
  if (argc > 1) {
    // sustained mode: `colown_stream <tiles>`: random operands, long kernels, full chip only
    g_tiles = atoi(argv[1]);
    fill_random_halves(wimg, wbytes / sizeof(u4), 1u);
    fill_random_halves(rows, 4096, 2u);
    run<2, 2>("RANDOM DATA: MFMA only (operands resident)", 256, wimg, rows, out);
    run<7, 4>("RANDOM DATA: weights streamed + LDS rows + MFMA", 256, wimg, rows, out);
    hipMemset(wimg, 0, wbytes);
    hipMemset(rows, 0, 4096 * sizeof(u4));
    run<2, 2>("ZERO DATA:   MFMA only (operands resident)", 256, wimg, rows, out);
    run<7, 4>("ZERO DATA:   weights streamed + LDS rows + MFMA", 256, wimg, rows, out);
    return 0;
  }
  for (int blocks : {1, 256, 1024}) {
    run<2, 2>("MFMA only (operands resident)", blocks, wimg, rows, out);
    run<6, 2>("MFMA + row fragments from LDS", blocks, wimg, rows, out);
    run<1, 4>("weight stream only (global -> VGPR)", blocks, wimg, rows, out);
    run<7, 2>("weights streamed + LDS rows + MFMA", blocks, wimg, rows, out);
    run<7, 4>("weights streamed + LDS rows + MFMA", blocks, wimg, rows, out);
    run<7, 8>("weights streamed + LDS rows + MFMA", blocks, wimg, rows, out);
  }
  return 0;
}
