// Micro-benchmark (round 5): at the part's POWER limit, what does it buy to feed MORE MFMAs from each weight fragment
// read out of LDS?  The shipped f16x3 row-MLP kernels give a wave 16 rows: every 1 KiB fragment (ds_read_b128) feeds
// 1.5 MFMAs (hi and lo fragment -> three v_mfma_f32_16x16x32_f16), four waves of a workgroup read the same ring, two
// workgroups per CU stream it twice: per CU and 16 KiB quarter of weights 128 KiB of LDS reads + 32 KiB of L2 -> LDS.
// DESIGN.md section 9.14 names "more MACs per operand byte read" as one of the few things that can still move a
// power-bound step; this skeleton prices it before anyone builds the kernel.  Same MFMA count per CU in every variant:
//
//   0  pair      two 4-wave workgroups per CU, 16 rows per wave              DMA 2x  LDS reads 2x   (the shipped structure)
//   1  wide      one 8-wave workgroup per CU,  16 rows per wave              DMA 1x  LDS reads 2x
//   2  rows32    one 4-wave workgroup per CU,  32 rows per wave (2 B sets)   DMA 1x  LDS reads 1x   (needs ~384 registers in the
//                                                                                                    real kernel: one wave per SIMD)
//   3  m32       as 2 with v_mfma_f32_32x32x16_f16: half the register-file operand reads per MAC as well
//   4  helper    one 8-wave workgroup per CU: 4 multiplying + 4 weight-staging waves     DMA 2x  LDS reads 2x   (round 4's form
//                                                                                       of the big node-side launches)
//
// Every variant runs ~2 s on all 256 CUs with random f16 operands while the host polls `amd-smi` for socket power and
// shader clock; the figure of merit is wall time for the same MFMA work (= energy at the power limit).
//
//   hipcc --offload-arch=gfx950 -O3 rows_per_fragment.hip -o /tmp/rows_per_fragment && /tmp/rows_per_fragment <variant>
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
#include <chrono>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f4 mm(u4 a, u4 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f16v mm32(u4 a, u4 b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int kQFloats = 4096;        // one quarter: 8 fragment slots x (hi, lo) x 1 KiB = 16 KiB
constexpr int kStream = 128;          // quarters in the weight image (2 MiB, L2-resident)
constexpr int kRing = 4;
constexpr int NG = 4;                 // fragment slots per group

__device__ __forceinline__ void piece(const float* gsrc, float* lds, int p, int lane) {
  const unsigned m0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>(lds + p * 256)));
  const float* src = gsrc + p * 256 + lane * 4;
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(src) : "memory", "m0");
}

// One quarter, 16x16x32: 8 n-blocks x 3 RS MFMAs; the next group's fragment reads ride behind this group's MFMAs, the
// next quarter's first group behind the barrier.
template <int RS>
__device__ __forceinline__ void quarter16(f4 (&acc)[RS][16], int nb0, const u4* wb, const u4* wb_next, u4 (&fh)[NG],
                                          u4 (&fl)[NG], const u4 (&bh)[RS], const u4 (&bl)[RS]) {
#pragma unroll
  for (int T = 0; T < 8 / NG; ++T) {
    const bool last = T + 1 == 8 / NG;
    u4 nh[NG], nl[NG];
    if (last) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < NG; ++q) {
#pragma unroll
      for (int r = 0; r < RS; ++r) acc[r][nb0 + NG * T + q] = mm(fh[q], bh[r], acc[r][nb0 + NG * T + q]);
      if (!last) nh[q] = wb[(NG * (T + 1) + q) * 128];
      FENCE();
    }
    if (last) asm volatile("s_barrier" ::: "memory");      // the next quarter is published; this one's buffer is free
#pragma unroll
    for (int q = 0; q < NG; ++q) {
#pragma unroll
      for (int r = 0; r < RS; ++r) acc[r][nb0 + NG * T + q] = mm(fh[q], bl[r], acc[r][nb0 + NG * T + q]);
      if (!last) nl[q] = wb[(NG * (T + 1) + q) * 128 + 64];
      else nh[q] = wb_next[q * 128];
      FENCE();
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
#pragma unroll
      for (int r = 0; r < RS; ++r) acc[r][nb0 + NG * T + q] = mm(fl[q], bh[r], acc[r][nb0 + NG * T + q]);
      if (last) nl[q] = wb_next[q * 128 + 64];
      FENCE();
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      fh[q] = nh[q];
      fl[q] = nl[q];
    }
  }
}

// One quarter, 32x32x16: slot s = (n32 block s >> 1, K half s & 1); 3 MFMAs of 32 cycles per slot.
__device__ __forceinline__ void quarter32(f16v (&acc)[8], int nb0, const u4* wb, const u4* wb_next, u4 (&fh)[NG],
                                          u4 (&fl)[NG], const u4 (&bh)[2], const u4 (&bl)[2]) {
#pragma unroll
  for (int T = 0; T < 8 / NG; ++T) {
    const bool last = T + 1 == 8 / NG;
    u4 nh[NG], nl[NG];
    if (last) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int s = NG * T + q;
      acc[nb0 + (s >> 1)] = mm32(fh[q], bh[s & 1], acc[nb0 + (s >> 1)]);
      if (!last) nh[q] = wb[(NG * (T + 1) + q) * 128];
      FENCE();
    }
    if (last) asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int s = NG * T + q;
      acc[nb0 + (s >> 1)] = mm32(fh[q], bl[s & 1], acc[nb0 + (s >> 1)]);
      if (!last) nl[q] = wb[(NG * (T + 1) + q) * 128 + 64];
      else nh[q] = wb_next[q * 128];
      FENCE();
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int s = NG * T + q;
      acc[nb0 + (s >> 1)] = mm32(fl[q], bh[s & 1], acc[nb0 + (s >> 1)]);
      if (last) nl[q] = wb_next[q * 128 + 64];
      FENCE();
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      fh[q] = nh[q];
      fl[q] = nl[q];
    }
  }
}

// WAVES waves share one ring (each stages 16 / WAVES pieces of every quarter itself: the shipped structure);
// RS = B operand sets per wave (16 rows each); M32: the 32-row form with 32x32x16 MFMAs.
// HELP: waves 0-3 multiply (16 rows each, no DMA), waves 4-7 stage the whole ring for them (4 pieces per wave and
// quarter) -- the shipped rowmlp16d_kernel's split; 96 MFMAs per CU and quarter, so the host runs it over twice the quarters.
template <int WAVES, int RS, int M32, int WGS_PER_CU, int HELP = 0>
__global__ __launch_bounds__(64 * (HELP ? 8 : WAVES), HELP ? 2 : WGS_PER_CU) void rpf(const float* __restrict__ w, f4* out, int n_quarters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PPW = HELP ? 0 : 16 / WAVES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto src = [&](int q) { return w + (size_t)(q % kStream) * kQFloats; };
  auto buf = [&](int q) { return smem + (q % kRing) * kQFloats; };
  if constexpr (HELP != 0) {
    if (wave >= 4) {
      const int hw = wave & 3;
      for (int q = 0; q < kRing - 1; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) piece(src(q), buf(q), 4 * hw + p, lane);
#pragma unroll 1
      for (int q = 0; q < n_quarters; ++q) {
        if (q < kRing) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kRing - 2) * 4) : "memory");
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int p = 0; p < 4; ++p) piece(src(q + kRing - 1), buf(q + kRing - 1), 4 * hw + p, lane);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
  }
  for (int q = 0; q < kRing - 1; ++q)
#pragma unroll
    for (int p = 0; p < PPW; ++p) piece(src(q), buf(q), PPW * wave + p, lane);
  u4 bh[2], bl[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    bh[r] = reinterpret_cast<const u4*>(w)[lane + 64 * ((wave + 3 * r) & 7)];
    bl[r] = reinterpret_cast<const u4*>(w)[lane + 64 * ((wave + 3 * r) & 7) + 512];
  }
  u4 fh[NG], fl[NG];
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kRing - 2) * PPW) : "memory");       // (also this wave's B operand loads)
  asm volatile("s_barrier" ::: "memory");
  {
    const u4* wb = reinterpret_cast<const u4*>(buf(0)) + lane;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      fh[q] = wb[q * 128];
      fl[q] = wb[q * 128 + 64];
    }
  }
  f4 s = f4{0.f, 0.f, 0.f, 0.f};
  if constexpr (M32) {
    f16v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
#pragma unroll 1
    for (int q0 = 0; q0 + 2 < n_quarters; q0 += 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = q0 + j;
        const u4* wb = reinterpret_cast<const u4*>(buf(q)) + lane;
        const u4* wn = reinterpret_cast<const u4*>(buf(q + 1)) + lane;
#pragma unroll
        for (int p = 0; p < PPW; ++p) piece(src(q + kRing - 1), buf(q + kRing - 1), PPW * wave + p, lane);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kRing - 2) * PPW) : "memory");
        quarter32(acc, 4 * j, wb, wn, fh, fl, bh, bl);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < 16; k += 4) s += f4{acc[i][k], acc[i][k + 1], acc[i][k + 2], acc[i][k + 3]};
  } else {
    f4 acc[RS][16];
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][i] = f4{0.f, 0.f, 0.f, 0.f};
    u4 bhr[RS], blr[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) { bhr[r] = bh[r]; blr[r] = bl[r]; }
#pragma unroll 1
    for (int q0 = 0; q0 + 2 < n_quarters; q0 += 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = q0 + j;
        const u4* wb = reinterpret_cast<const u4*>(buf(q)) + lane;
        const u4* wn = reinterpret_cast<const u4*>(buf(q + 1)) + lane;
#pragma unroll
        for (int p = 0; p < PPW; ++p) piece(src(q + kRing - 1), buf(q + kRing - 1), PPW * wave + p, lane);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kRing - 2) * PPW) : "memory");
        quarter16<RS>(acc, 8 * j, wb, wn, fh, fl, bhr, blr);
      }
    }
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[r][i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[(size_t)blockIdx.x * 512 + tid] = s;
}

// ---- host: power / clock polling (amd-smi) while the kernel runs
static bool find_value(const std::string& txt, const char* key, size_t from, double* v, size_t* at = nullptr) {
  size_t p = txt.find(key, from);
  if (p == std::string::npos) return false;
  size_t q = txt.find("\"value\"", p);
  if (q == std::string::npos) return false;
  q = txt.find(':', q);
  if (q == std::string::npos) return false;
  *v = atof(txt.c_str() + q + 1);
  if (at) *at = q;
  return true;
}
static bool read_smi(double* watts, double* mhz) {
  FILE* f = popen("amd-smi metric -g 0 --power --clock --json 2>/dev/null", "r");
  if (!f) return false;
  std::string txt;
  char b[4096];
  size_t n;
  while ((n = fread(b, 1, sizeof(b), f)) > 0) txt.append(b, n);
  pclose(f);
  const bool a = find_value(txt, "\"socket_power\"", 0, watts);
  const bool c = find_value(txt, "\"gfx_0\"", 0, mhz);
  return a && c;
}
static double median(std::vector<double> v) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

template <int WAVES, int RS, int M32, int WGS_PER_CU, int HELP = 0>
void run(const char* name, const float* w, f4* out, int nq_in) {
  const int nq = HELP ? 2 * (nq_in - 1) + 1 : nq_in;          // (the helper form multiplies with four waves per CU: twice the quarters)
  const size_t lds = WGS_PER_CU == 2 ? kRing * kQFloats * sizeof(float) : 96 * 1024;   // (96 KiB: ONE workgroup per CU)
  auto fn = rpf<WAVES, RS, M32, WGS_PER_CU, HELP>;
  constexpr int kThreads = 64 * (HELP ? 8 : WAVES);
  hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = 256 * WGS_PER_CU;
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(kThreads), lds, 0, w, out, 65);
  {
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(e)); exit(1); }
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> pw, ck;
  std::atomic<bool> stop{false};
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(kThreads), lds, 0, w, out, nq);
  hipEventRecord(e1);
  std::thread poll([&] {
    std::this_thread::sleep_for(std::chrono::milliseconds(400));     // (the governor settles within ~0.3 s)
    while (!stop.load()) {
      double a = 0, c = 0;
      if (read_smi(&a, &c) && !stop.load()) { pw.push_back(a); ck.push_back(c); }
    }
  });
  hipEventSynchronize(e1);
  stop.store(true);
  poll.join();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // MFMA pipe cycles per SIMD: every variant issues 48 x 16 cycles per (quarter pair of the pair form | quarter of the others)
  const double mfma_cycles = (double)(nq_in - 1) * 768.0;
  const double macs = (double)(nq_in - 1) * 192.0 * 8192.0 * 256.0;    // per launch, all CUs
  const double mhz = median(ck);
  printf("{\"variant\": \"%s\", \"ms\": %.2f, \"mfma_tflops\": %.1f, \"power_w_median\": %.0f, \"sclk_mhz_median\": %.0f, \"samples\": %zu, "
         "\"mfma_pipe_busy_at_that_clock\": %.3f, \"joule_per_tmac\": %.3f}\n",
         name, ms, 2.0 * macs / (ms * 1e-3) * 1e-12, median(pw), mhz, pw.size(),
         mhz > 0 ? mfma_cycles / (ms * 1e-3 * mhz * 1e6) : 0.0, median(pw) * ms * 1e-3 / (macs * 1e-12));
}

int main(int argc, char** argv) {
  float* w; f4* out;
  const size_t wn = (size_t)kStream * kQFloats;
  hipMalloc(&w, wn * sizeof(float));
  {
    // non-trivial f16 operands (power under load depends on the data): halves with random sign, exponents 2^-3 .. 2^0,
    // random mantissas
    std::vector<unsigned> h(wn);
    unsigned s = 12345u;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      const unsigned a = (0x3000u + (((s >> 4) & 3u) << 10)) | ((s >> 8) & 0x03ffu) | ((s & 1u) << 15);
      const unsigned b = (0x3000u + (((s >> 6) & 3u) << 10)) | ((s >> 20) & 0x03ffu) | ((s & 2u) << 14);
      v = a | (b << 16);
    }
    hipMemcpy(w, h.data(), wn * sizeof(float), hipMemcpyHostToDevice);
  }
  hipMalloc(&out, (size_t)512 * 512 * sizeof(f4));
  const int which = argc > 1 ? atoi(argv[1]) : -1;
  const int nq = argc > 2 ? atoi(argv[2]) : 4000001;      // (odd: the loop takes quarters in twos)
  setvbuf(stdout, nullptr, _IONBF, 0);
  switch (which) {
    case 0: run<4, 1, 0, 2>("pair: 2 x 4 waves x 16 rows (DMA 2x, LDS reads 2x)", w, out, nq); break;
    case 1: run<8, 1, 0, 1>("wide: 8 waves x 16 rows, one ring (DMA 1x, LDS reads 2x)", w, out, nq); break;
    case 2: run<4, 2, 0, 1>("rows32: 4 waves x 32 rows (DMA 1x, LDS reads 1x)", w, out, nq); break;
    case 3: run<4, 2, 1, 1>("m32: 4 waves x 32 rows, 32x32x16 MFMAs (DMA 1x, LDS reads 1x, operand reads 1/2)", w, out, nq); break;
    case 4: run<4, 1, 0, 1, 1>("helper: 4 multiplying + 4 staging waves x 16 rows (DMA 2x, LDS reads 2x)", w, out, nq); break;
    default: printf("usage: rows_per_fragment <variant 0..4> [quarters]\n");
  }
  return 0;
}
