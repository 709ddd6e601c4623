// gcast.hip -- gfx950 (MI355X / CDNA4) kernels + C-ABI for GraphCast's
// encode-process-decode step.  See include/gcast.h for the interface and
// DESIGN.md for the design.  fp32 throughout (exact-fp32 MFMA, 157 TF peak).
//
// The one hot kernel is `rowmlp_kernel`: a 64-row tile (4 waves x 16 rows) is
// pushed through  z = A.W1 + addends -> swish -> .W2 + b2 -> LayerNorm
// [-> + residual] [-> receiver segment-sum]  without leaving the CU.
//
// "Transposed, register-chained" MFMA formulation (v_mfma_f32_16x16x4_f32):
//   C^T[n][m] += sum_k W^T[n][k] * X^T[k][m]
//   A operand  = weights: lane l supplies W[k = kb + 4*(l>>4) + j][n = 16*nb + (l&15)]
//   B operand  = rows:    lane l supplies X[row = l&15][k = kb + 4*(l>>4) + j]
//   C/D        : lane l, reg r holds out[row = l&15][n = 16*nb + 4*(l>>4) + r]
// so the 4 accumulator registers of block nb' are, as they sit, the B operands
// of the NEXT layer's k-steps kb = 16*nb' (j = r): the hidden activations never
// touch LDS or HBM.  Weights stream through LDS in 32-row K chunks laid out
// [k/4][n][k%4] (packed on the host), DMA'd linearly with global_load_lds and
// read as conflict-free ds_read_b128 (slot index == n mod 16 within a lane group).
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gcast.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#ifndef GC_SCHED_PIN
#define GC_SCHED_PIN 1
#endif
#ifndef GC_DMA_ASM
#define GC_DMA_ASM 1     // LDS-DMA as inline asm (see stage_piece)
#endif
#ifndef GC_ASM_FLUSH
#define GC_ASM_FLUSH 1   // segment-sum run flush as an inline-asm store (see finish_rows; -1.1 % per launch)
#endif
#ifndef GC_EXP
#define GC_EXP 0         // profiling experiments ONLY on the chunked kernels (results are wrong; round-1 probes):
#endif                   // bit0 no weight DMA after the prologue, bit1 no fragment reads, bit2 no MFMA,
                         // bit3 never wait for the DMA

// Profiling switches that make a kernel compute WRONG results (GC_EXP: pieces of the work
// removed) or write timestamps over result buffers (GC_H_TRACE) only compile in a build that
// says so: scripts/probes/build_probe_lib.sh passes -DGC_PROFILING_BUILD (scripts/half_probe.py loads such a
// library NEXT to the product one), the product build
// (graphcast_amd/_native.py: build) never does, gc_build_info() reports it and the Python binding
// refuses to load such a library as the product.
#if GC_EXP != 0 && !defined(GC_PROFILING_BUILD)
#error "GC_EXP is profiling-only: compile with -DGC_PROFILING_BUILD"
#endif

namespace {

constexpr int kD = GC_LATENT;            // 512
constexpr int kNB = kD / 16;             // 32 n-blocks of 16
constexpr int kBufFloats = 8 * kD * 4;   // one LDS weight buffer: 8 k4-groups x 512 n x 4 = 64 KiB
constexpr int kYld = kD + 4;             // padded row stride of the segment-sum staging tile
constexpr int kLdsFloats = GC_TILE_ROWS * kYld + GC_TILE_ROWS;   // Y tile (aliases both buffers) + seg ids
constexpr float kLnEps = 1e-5f;          // haiku LayerNorm default

thread_local char g_err[256] = "";

int fail(int code, const char* msg) {
  std::snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One 1 KiB piece (per wave) of a K chunk of packed weights -> LDS.  A chunk of NP
// columns is 8 * NP * 4 floats = NP / 32 pieces per wave (4 waves).
// LDS-DMA: destination = wave-uniform base (M0) + lane * 16 B (linear image), no VGPR round trip.
//
// GC_DMA_ASM (default): issued as inline asm.  Through the builtin, hipcc models the instruction
// as a FLAT access that may return out of order with LDS reads, and from then on every wait for a
// ds_read result becomes s_waitcnt lgkmcnt(0) -- which forbids keeping the NEXT group's
// fragment reads in flight across the current group's first MFMAs.  As asm the compiler does
// not see it at all, so the code waits for it explicitly (dma_wait) before the barrier that
// publishes a chunk; its over-conservative vmcnt accounting for ordinary loads stays safe
// (memory returns in order: it can only wait for more than it needs).
__device__ __forceinline__ void stage_piece(const float* __restrict__ gsrc, float* lds, int piece,
                                            int wave, int lane) {
  const int off = piece * 1024 + wave * 256;     // wave-uniform float offset of this 1 KiB piece
#if GC_DMA_ASM
  const unsigned m0 = __builtin_amdgcn_readfirstlane(
      static_cast<unsigned>(reinterpret_cast<size_t>(lds + off)));
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(m0), "v"(gsrc + off + lane * 4) : "memory", "m0");
#else
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)(gsrc + off + lane * 4),
      (__attribute__((address_space(3))) void*)(lds + off), 16, 0, 0);
#endif
}

// All LDS-DMA this wave has issued has landed (a no-op for the builtin path, where the barrier's
// own fence waits).
__device__ __forceinline__ void dma_wait() {
#if GC_DMA_ASM && !(GC_EXP & 8)       // (GC_EXP bit3: profiling only -- never wait for the DMA)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// Copies one whole K chunk of packed weights (8 * NP * 4 floats, contiguous) into LDS.
template <int NP>
__device__ __forceinline__ void stage_chunk(const float* __restrict__ gsrc, float* lds, int tid) {
  static_assert((8 * NP * 4) % 1024 == 0, "chunk must be a whole number of 4 KiB block copies");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
  for (int it = 0; it < NP / 32; ++it) stage_piece(gsrc, lds, it, wave, lane);
}

// acc[nb] += W-chunk(32 k) * B for all n-blocks; b0/b1 are the B operands of the two
// 16-k halves.  The chunk is walked in "groups" of two n-blocks (8 MFMAs = 256 issue
// cycles on this SIMD):
//   * the two A fragments (ds_read_b128) of group t+1 are requested BEFORE the MFMAs
//     of group t are issued, so the LDS latency hides behind 256 cycles of MFMA
//     instead of stalling every group (1 wave per SIMD: nobody else covers it);
//   * dependent MFMAs on one accumulator are 2 issue slots apart (40-cycle dependent
//     latency vs 32-cycle issue);
//   * the NEXT chunk's LDS-DMA (NP_NEXT / 32 pieces per wave) is issued one piece per
//     group over the first groups instead of as one burst in front of the MFMAs.
// Group T of a chunk (template recursion: the scheduling builtins need constants).
template <int NBLK, int NP, int NP_NEXT, int T>
__device__ __forceinline__ void mma_group(f4 (&acc)[kNB], const float* base0, const float* base1,
                                          f4 a0, f4 a1, f4 b0, f4 b1,
                                          const float* __restrict__ next_src, float* next_dst,
                                          int wave, int lane) {
  constexpr int kGroupsPerHalf = (NBLK + 1) / 2;
  constexpr int kGroups = 2 * kGroupsPerHalf;
  constexpr int kPieces = NP_NEXT / 32;
  static_assert(kPieces <= kGroups, "more staging pieces than MFMA groups");
  constexpr int s = T / kGroupsPerHalf;
  constexpr int nb = 2 * (T % kGroupsPerHalf);
  constexpr bool pair = nb + 1 < NBLK;
  constexpr bool more = T + 1 < kGroups;
  constexpr int s2 = (T + 1) / kGroupsPerHalf;
  constexpr int nb2 = 2 * ((T + 1) % kGroupsPerHalf);
  constexpr bool pair2 = nb2 + 1 < NBLK;
  // piece p of the next chunk is issued in group p: all pieces are in flight early in the
  // chunk, so the barrier that ends it does not wait on a freshly issued DMA
  constexpr bool piece = T < kPieces;
  const f4 b = s ? b1 : b0;
  // Issue order inside a group (held by full scheduling fences; left alone the scheduler
  // sinks the reads of group T+1 next to their first use and every group stalls on LDS):
  //   1. the LDS-DMA piece.  hipcc treats global_load_lds as a FLAT access that may touch
  //      LDS, so the next use of any ds_read result gets s_waitcnt lgkmcnt(0) ...
  //   2. ... which is why the first MFMA pair comes next: the only reads outstanding are
  //      this group's fragments, requested 6+ MFMAs (192+ cycles) ago -- the wait is free;
  //   3. then the A fragments of group T+1 are requested,
  //   4. and the remaining MFMAs of this group cover their latency.
  if constexpr (piece) stage_piece(next_src, next_dst, T, wave, lane);
#if GC_SCHED_PIN
  __builtin_amdgcn_sched_barrier(0);
#endif
  acc[nb] = mfma16(a0.x, b.x, acc[nb]);
  if constexpr (pair) acc[nb + 1] = mfma16(a1.x, b.x, acc[nb + 1]);
#if GC_SCHED_PIN
  __builtin_amdgcn_sched_barrier(0);
#endif
  f4 n0 = a0, n1 = a1;
  if constexpr (more) {
    const float* bs = s2 ? base1 : base0;
    n0 = *reinterpret_cast<const f4*>(bs + (nb2 * 16 << 2));
    if constexpr (pair2) n1 = *reinterpret_cast<const f4*>(bs + ((nb2 + 1) * 16 << 2));
  }
#if GC_SCHED_PIN
  __builtin_amdgcn_sched_barrier(0);
#endif
  if constexpr (pair) {
    acc[nb] = mfma16(a0.y, b.y, acc[nb]);
    acc[nb + 1] = mfma16(a1.y, b.y, acc[nb + 1]);
    acc[nb] = mfma16(a0.z, b.z, acc[nb]);
    acc[nb + 1] = mfma16(a1.z, b.z, acc[nb + 1]);
    acc[nb] = mfma16(a0.w, b.w, acc[nb]);
    acc[nb + 1] = mfma16(a1.w, b.w, acc[nb + 1]);
  } else {
    acc[nb] = mfma16(a0.y, b.y, acc[nb]);
    acc[nb] = mfma16(a0.z, b.z, acc[nb]);
    acc[nb] = mfma16(a0.w, b.w, acc[nb]);
  }
#if GC_SCHED_PIN
  __builtin_amdgcn_sched_barrier(0);
#endif
  if constexpr (more) {
    mma_group<NBLK, NP, NP_NEXT, T + 1>(acc, base0, base1, n0, n1, b0, b1, next_src, next_dst, wave, lane);
  }
}

template <int NBLK, int NP, int NP_NEXT>
__device__ __forceinline__ void mma_chunk(f4 (&acc)[kNB], const float* wb, f4 b0, f4 b1, int i, int g,
                                          const float* __restrict__ next_src, float* next_dst,
                                          int wave, int lane) {
  const float* base0 = wb + ((g * NP + i) << 2);
  const float* base1 = wb + (((4 + g) * NP + i) << 2);
  const f4 a0 = *reinterpret_cast<const f4*>(base0);
  const f4 a1 = *reinterpret_cast<const f4*>(base0 + (16 << 2));
  mma_group<NBLK, NP, NP_NEXT, 0>(acc, base0, base1, a0, a1, b0, b1, next_src, next_dst, wave, lane);
}

// ---------------------------------------------------------------------------------------
// GC_PREC_F16X3: fp32-grade GEMMs on the f16 matrix cores.  x = x_hi + x_lo with
// x_hi = fp16(x), x_lo = fp16(x - x_hi) (22 mantissa bits; the subtraction is exact in fp32),
// x.w ~= x_hi.w_hi + x_lo.w_hi + x_hi.w_lo, three v_mfma_f32_16x16x32_f16 accumulating in
// fp32.  Same transposed, register-chained formulation as the fp32 path:
//   A operand = weights: lane l = 16 g + n supplies W[kmap(g, j)][16 nb + n], j < 8 (one
//               ds_read_b128 of the pre-split hi (or lo) image, conflict-free: 64 lanes x 16 B
//               contiguous)
//   B operand = rows:    lane l = 16 g + i supplies X[row i][kmap(g, j)], split in registers
//   C/D       = lane l, reg r: out[row i][16 nb + 4 g + r]   (identical to the fp32 path)
// so two adjacent accumulator blocks (2c, 2c+1) are, after the split, the B operand of K step
// c of the next layer ("chained" kmap, see gcast.h).
__device__ __forceinline__ f4 mfma32h(u4 a, u4 b, f4 c) {
#if GC_EXP & 4
  c.x += __builtin_bit_cast(float, a.x ^ b.x);      // keeps the operands alive, no matrix core
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b),
                                                c, 0, 0, 0);
}

// (a, b) -> packed halves hi, lo with hi + lo = the input to 20+ bits.  Both conversions round
// toward zero (v_cvt_pkrtz_f16_f32: two values per instruction): hi is the input truncated to 11
// bits, the remainder a - hi is exact in fp32 and has the input's sign, lo is the remainder
// truncated to 11 bits => |a - (hi + lo)| < 2^-20 |a|.  Round-toward-zero also SATURATES
// instead of overflowing to inf: beyond +-65504 hi clamps and lo carries the excess (exact up
// to 1.3e5), no clamps needed.
typedef __fp16 p2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  const p2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
  const float ra = a - static_cast<float>(h.x);
  const float rb = b - static_cast<float>(h.y);
  const p2 l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ void split8(f4 a, f4 b, u4& hi, u4& lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  split2(a.x, a.y, h0, l0);
  split2(a.z, a.w, h1, l1);
  split2(b.x, b.y, h2, l2);
  split2(b.z, b.w, h3, l3);
  hi = u4{h0, h1, h2, h3};
  lo = u4{l0, l1, l2, l3};
}

#define GC_FENCE() __builtin_amdgcn_sched_barrier(0)      // pins the issue order of the MFMA / memory weave (rowmlp_half.inc)

__device__ __forceinline__ float swish1(float x) {
  // x * sigmoid(x); __expf/fast reciprocal are ~1-2 ulp, far inside the 1e-4 budget.
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

__device__ __forceinline__ float group_sum4(float v) {
  // sum over the 4 lanes {l, l^16, l^32, l^48} that share one row
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// ---- pieces shared by the two arithmetic modes ----------------------------------------

// Layer-1 accumulators start from the addends: b1 + d[row] + g0[idx0[row]] + g1[idx1[row]].
// `s` = the power of two the layer's packed weights carry (1 in fp32 mode): exact scaling.
__device__ __forceinline__ void init_addends(f4 (&acc)[kNB], const gc_rowmlp_desc& d, int rowc, int col0,
                                             float s) {
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
  if (d.b1) {
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) acc[nb] += s * *reinterpret_cast<const f4*>(d.b1 + nb * 16 + col0);
  }
  if (d.d) {
    const float* p = d.d + (size_t)rowc * d.ldd + col0;
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) acc[nb] += s * *reinterpret_cast<const f4*>(p + nb * 16);
  }
  if (d.g0) {
    int ix = d.idx0[rowc];
    ix = ix < 0 ? 0 : ix;
    const float* p = d.g0 + (size_t)ix * kD + col0;
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) acc[nb] += s * *reinterpret_cast<const f4*>(p + nb * 16);
  }
  if (d.g1) {
    int ix = d.idx1[rowc];
    ix = ix < 0 ? 0 : ix;
    const float* p = d.g1 + (size_t)ix * kD + col0;
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) acc[nb] += s * *reinterpret_cast<const f4*>(p + nb * 16);
  }
}

template <class Desc>
__device__ __forceinline__ void store_linear(const f4 (&acc)[kNB], const Desc& d, int row, int col0) {
  if (row < d.n_rows) {
    float* o = d.out + (size_t)row * d.ldo + col0;
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) *reinterpret_cast<f4*>(o + nb * 16) = acc[nb];
  }
}

__device__ __forceinline__ void swish_all(f4 (&acc)[kNB]) {
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) {
    acc[nb].x = swish1(acc[nb].x);
    acc[nb].y = swish1(acc[nb].y);
    acc[nb].z = swish1(acc[nb].z);
    acc[nb].w = swish1(acc[nb].w);
  }
}

// Everything after layer 2: decoder store, or LayerNorm -> [+ residual] store -> segment-sum.
template <int MODE>
__device__ __forceinline__ void finish_rows(f4 (&o2)[kNB], const gc_rowmlp_desc& d, float* smem, int tile,
                                            int row, int wave, int i, int col0, int tid) {
  constexpr int NB2 = MODE == GC_MODE_MLP_OUT ? 15 : 32;
  if (MODE == GC_MODE_MLP_OUT) {
    if (row < d.n_rows) {
      float* o = d.out + (size_t)row * d.ldo;
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) {
        const int n = nb * 16 + col0;
        if (n + 0 < d.n2) o[n + 0] = o2[nb].x;
        if (n + 1 < d.n2) o[n + 1] = o2[nb].y;
        if (n + 2 < d.n2) o[n + 2] = o2[nb].z;
        if (n + 3 < d.n2) o[n + 3] = o2[nb].w;
      }
    }
    return;
  }

  // ---- LayerNorm over the 512 outputs of each row (128 per lane x 4 lanes) -------
  if (d.ln_scale) {
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) s += (o2[nb].x + o2[nb].y) + (o2[nb].z + o2[nb].w);
    const float mean = group_sum4(s) * (1.0f / kD);
    float v = 0.f;
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) {
      o2[nb] -= mean;
      v += (o2[nb].x * o2[nb].x + o2[nb].y * o2[nb].y) + (o2[nb].z * o2[nb].z + o2[nb].w * o2[nb].w);
    }
    const float rstd = rsqrtf(group_sum4(v) * (1.0f / kD) + kLnEps);
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) {
      const f4 sc = *reinterpret_cast<const f4*>(d.ln_scale + nb * 16 + col0);
      const f4 of = *reinterpret_cast<const f4*>(d.ln_offset + nb * 16 + col0);
      o2[nb] = o2[nb] * rstd * sc + of;
    }
  }

  // ---- store (with residual) -------------------------------------------------------
  if (d.out && row < d.n_rows) {
    float* o = d.out + (size_t)row * d.ldo + col0;
    if (d.res) {
      const float* r = d.res + (size_t)row * d.ldres + col0;
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb)
        *reinterpret_cast<f4*>(o + nb * 16) = o2[nb] + *reinterpret_cast<const f4*>(r + nb * 16);
    } else {
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) *reinterpret_cast<f4*>(o + nb * 16) = o2[nb];
    }
  }

  // ---- deterministic segment-sum over the tile's receiver-sorted rows -------------
  if (d.seg) {
    float* ytile = smem;
    int* segs = reinterpret_cast<int*>(smem + GC_TILE_ROWS * kYld);
    __syncthreads();   // all waves are done with the weight buffers
    {
      float* y = ytile + (wave * 16 + i) * kYld + col0;
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) *reinterpret_cast<f4*>(y + nb * 16) = o2[nb];
    }
    if (tid < GC_TILE_ROWS) segs[tid] = d.seg[tile * GC_TILE_ROWS + tid];
    __syncthreads();
    const int flags = d.tile_flags[tile];
    const int c2 = tid * 2;          // this thread owns columns c2, c2+1
    float sx = 0.f, sy = 0.f;
    int cur = -1;
    int run_start = 0;
    auto flush = [&](int r) {
      if (cur >= 0) {
        float* dst;
        if (run_start == 0 && (flags & 1)) {
          dst = d.partial + (size_t)(2 * tile) * kD;
        } else if (r == GC_TILE_ROWS && (flags & 2)) {
          dst = d.partial + (size_t)(2 * tile + 1) * kD;
        } else {
          dst = d.agg + (size_t)cur * kD;
        }
#if GC_ASM_FLUSH
        // as inline asm: hipcc does not see the store, so it does not drain vmcnt at the join of
        // this conditional (one write-acknowledge latency per run otherwise); s_endpgm waits for it
        const unsigned long long bits = (unsigned long long)__float_as_uint(sx) |
                                        ((unsigned long long)__float_as_uint(sy) << 32);
        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst + c2), "v"(bits) : "memory");
#else
        *reinterpret_cast<float2*>(dst + c2) = make_float2(sx, sy);
#endif
      }
    };
#pragma unroll 1
    for (int r0 = 0; r0 < GC_TILE_ROWS; r0 += 8) {
      // eight rows per trip: ids (two broadcast reads) and values up front -- one LDS latency per
      // eight rows instead of one per row -- then pure VALU / branch logic, rows in order
      const int4 s0 = *reinterpret_cast<const int4*>(segs + r0);
      const int4 s1 = *reinterpret_cast<const int4*>(segs + r0 + 4);
      const int sg[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      float2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(ytile + (r0 + k) * kYld + c2);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int sid = sg[k];
        if (sid != cur) {
          flush(r0 + k);
          cur = sid;
          run_start = r0 + k;
          sx = 0.f;
          sy = 0.f;
        }
        if (sid >= 0) {
          sx += v[k].x;
          sy += v[k].y;
        }
      }
    }
    flush(GC_TILE_ROWS);
  }
}

// ---- GC_PREC_F32 ------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256, 1) void rowmlp_kernel(const gc_rowmlp_desc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool kLinear = MODE == GC_MODE_LINEAR;
  constexpr int NP2 = MODE == GC_MODE_MLP_OUT ? 256 : 512;
  constexpr int NB2 = MODE == GC_MODE_MLP_OUT ? 15 : 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i = lane & 15;          // row within the wave's 16 (B/C column), n within block (A row)
  const int g = lane >> 4;          // k sub-group (A/B), n sub-group (C)
  const int tile = blockIdx.x;
  const int row = tile * GC_TILE_ROWS + wave * 16 + i;
  const int rowc = row < d.n_rows ? row : d.n_rows - 1;
  const int col0 = 4 * g;           // this lane's first column inside a 16-wide n block
  const float* w1p = static_cast<const float*>(d.w1p);
  const float* w2p = static_cast<const float*>(d.w2p);

  const int n1 = (d.k0 + d.k1) >> 5;
  const int n1a = d.k0 >> 5;
  int q = 0;                        // position in the weight-chunk stream -> LDS buffer parity

  if (n1 > 0) {
    stage_chunk<512>(w1p, smem, tid);
  } else if (!kLinear) {
    stage_chunk<NP2>(w2p, smem, tid);
  }

  f4 acc[kNB];
  init_addends(acc, d, rowc, col0, 1.0f);

  // ---- layer 1: acc += A . W1, A rows streamed from global as B operands ---------
  if (n1 > 0) {
    const float* arow0 = d.a0 + (size_t)rowc * d.lda0 + col0;
    const float* arow1 = d.k1 ? d.a1 + (size_t)rowc * d.lda1 + col0 : arow0;
    f4 bc0, bc1, bn0, bn1;
    {
      const float* p = n1a > 0 ? arow0 : arow1;
      bc0 = *reinterpret_cast<const f4*>(p);
      bc1 = *reinterpret_cast<const f4*>(p + 16);
    }
    bn0 = bc0;
    bn1 = bc1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int c = 0; c + 1 < n1; ++c) {
      dma_wait();
      __syncthreads();   // chunk c landed in LDS; previous chunk's readers are done
      {
        const float* p = (c + 1 < n1a) ? arow0 + (c + 1) * 32 : arow1 + (c + 1 - n1a) * 32;
        bn0 = *reinterpret_cast<const f4*>(p);
        bn1 = *reinterpret_cast<const f4*>(p + 16);
      }
      mma_chunk<kNB, 512, 512>(acc, smem + (q & 1) * kBufFloats, bc0, bc1, i, g,
                               w1p + (size_t)(c + 1) * kBufFloats,
                               smem + ((q + 1) & 1) * kBufFloats, wave_u, lane);
      bc0 = bn0;
      bc1 = bn1;
      ++q;
    }
    dma_wait();
    __syncthreads();     // last layer-1 chunk; the first layer-2 chunk streams in behind it
    if (kLinear) {
      mma_chunk<kNB, 512, 0>(acc, smem + (q & 1) * kBufFloats, bc0, bc1, i, g, nullptr, nullptr,
                             wave_u, lane);
    } else {
      mma_chunk<kNB, 512, NP2>(acc, smem + (q & 1) * kBufFloats, bc0, bc1, i, g, w2p,
                               smem + ((q + 1) & 1) * kBufFloats, wave_u, lane);
    }
    ++q;
  }

  if (kLinear) {
    store_linear(acc, d, row, col0);
    return;
  }

  // ---- swish in place: acc becomes the hidden layer, already in B-operand layout -
  swish_all(acc);

  // ---- layer 2: out = hidden . W2 + b2 (fully unrolled: hidden regs are indexed by chunk)
  f4 o2[kNB];
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb) o2[nb] = *reinterpret_cast<const f4*>(d.b2 + nb * 16 + col0);
  {
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int cc = 0; cc < kD / 32; ++cc) {
      dma_wait();
      __syncthreads();
      if (cc + 1 < kD / 32) {
        mma_chunk<NB2, NP2, NP2>(o2, smem + (q & 1) * kBufFloats, acc[2 * cc], acc[2 * cc + 1], i, g,
                                 w2p + (size_t)(cc + 1) * (8 * NP2 * 4),
                                 smem + ((q + 1) & 1) * kBufFloats, wave_u, lane);
      } else {
        mma_chunk<NB2, NP2, 0>(o2, smem + (q & 1) * kBufFloats, acc[2 * cc], acc[2 * cc + 1], i, g,
                               nullptr, nullptr, wave_u, lane);
      }
      ++q;
    }
  }
  finish_rows<MODE>(o2, d, smem, tile, row, wave, i, col0, tid);
}

// An all-zero row: what an absent addend source reads in the kernels that take it unconditionally.
__device__ float g_zero_row[kD];


#include "rowmlp_half.inc"

// ---- bfloat16 MFMA / packing helpers of the GC_PREC_BF16 tier (rowmlp_bf16.inc) --------------------------------
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 mfma32b(u4 a, u4 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b),
                                                 c, 0, 0, 0);
}

__device__ __forceinline__ unsigned pack2bf(float a, float b) {     // v_cvt_pk_bf16_f32 (RNE)
  const f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

__device__ __forceinline__ u4 pack8bf(f4 a, f4 b) {
  return u4{pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w)};
}

#include "rowmlp_bf16.inc"

__global__ void seg_fixup_kernel(int n, const int* __restrict__ recv, const int* __restrict__ t0,
                                 const int* __restrict__ t1, const float* __restrict__ partial,
                                 float* __restrict__ agg) {
  const int e = blockIdx.x;
  if (e >= n) return;
  const int c = threadIdx.x * 4;    // 128 threads x 4 columns
  const int a = t0[e], b = t1[e];
  f4 s = *reinterpret_cast<const f4*>(partial + (size_t)(2 * a + 1) * kD + c);
  for (int t = a + 1; t <= b; ++t) s += *reinterpret_cast<const f4*>(partial + (size_t)(2 * t) * kD + c);
  *reinterpret_cast<f4*>(agg + (size_t)recv[e] * kD + c) = s;
}

__global__ void zero_rows_kernel(int n, const int* __restrict__ rows, float* __restrict__ agg) {
  const int e = blockIdx.x;
  if (e >= n) return;
  *reinterpret_cast<f4*>(agg + (size_t)rows[e] * kD + threadIdx.x * 4) = f4{0.f, 0.f, 0.f, 0.f};
}

// dst[rows[i], :] += src[rows[i], :]: the aggregate of the halo-sender edges of a partitioned edge update joins
// the aggregate of the sender-local ones (partition.py: the exchange runs under the local launch).
__global__ void add_rows_kernel(int n, const int* __restrict__ rows, const float* __restrict__ src,
                                float* __restrict__ dst) {
  const int e = blockIdx.x;
  if (e >= n) return;
  const size_t o = (size_t)rows[e] * kD + 4 * threadIdx.x;
  f4 a = *reinterpret_cast<const f4*>(dst + o);
  a += *reinterpret_cast<const f4*>(src + o);
  *reinterpret_cast<f4*>(dst + o) = a;
}

// The same rows, one float4 of the OUTPUT per thread (kp / 4 threads per row): the 32-column tail of the half-N
// path is eight threads per row and 32 rows per block instead of one half-empty wave per row (round 6: 0.16 ms of a
// 0.25 deg step were this kernel -- 0.27 GB of traffic).  Same values: element for element the picks of the kernel below.
__global__ __launch_bounds__(256) void prep_grid_rows4_kernel(int n_rows, int batch, int b, int c_in, int c0,
                                                              const float* __restrict__ x, int n_struct,
                                                              const float* __restrict__ node_struct, int kp,
                                                              float* __restrict__ xin) {
  const int per_row = kp >> 2;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = gid / per_row;
  if (row >= n_rows) return;
  const int q = (int)(gid - row * per_row);
  const float* src = x + ((size_t)row * batch + b) * c_in;
  const float* st = node_struct + (size_t)row * n_struct;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + 4 * q + j;
    v[j] = c < c_in ? src[c] : (c < c_in + n_struct ? st[c - c_in] : 0.f);
  }
  *reinterpret_cast<f4*>(xin + (size_t)row * kp + 4 * q) = f4{v[0], v[1], v[2], v[3]};
}

__global__ void prep_grid_input_kernel(int n_rows, int batch, int b, int c_in, int c0,
                                       const float* __restrict__ x, int n_struct,
                                       const float* __restrict__ node_struct, int kp,
                                       float* __restrict__ xin) {
  // one wave per row, lanes stride over the kp output columns = input columns c0 .. c0 + kp - 1
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float* src = x + ((size_t)row * batch + b) * c_in;
  const float* st = node_struct + (size_t)row * n_struct;
  float* dst = xin + (size_t)row * kp;
  for (int o = lane; o < kp; o += 64) {
    const int c = c0 + o;
    float v = 0.f;
    if (c < c_in) {
      v = src[c];
    } else if (c < c_in + n_struct) {
      v = st[c - c_in];
    }
    dst[o] = v;
  }
}

// A block owns kAdvRows consecutive rows.  Their x and y rows are ONE contiguous range each: copied into
// LDS with 16-byte loads (whole cache lines whatever the row length -- 474 floats = 1896 B), the next
// state and the prediction are formed in LDS (thread t owns channels t, t + 256, ...: its table entries
// are read once per block; the picks are 4-byte LDS reads) and leave as 16-byte stores of a contiguous
// range again.  Same arithmetic, element for element, as the one-wave-per-row kernel of rounds 2-3
// (2.33-2.43 ms per 0.25 deg step = 2.4 TB/s of useful bytes; this one 1.79 ms = 3.3 TB/s with 8 rows per block,
// 1.29 ms = 4.5 TB/s with 4 (round 4); two
// direct-to-global variants of this round were no faster than the old one, profiles/r03_s18_* .. r03_s21_*).
// (round 4, same-session A/B at the 0.25 deg shape, all bit-identical, profiles/r04_s14_*: 2 rows per block 1.90 ms -- 227
//  channels x 2 rows is no multiple of four floats: the y side falls back to 4-byte copies --, 4 rows 1.29 ms = 4.5 TB/s,
//  8 rows (rounds 3) 1.87 ms = 3.1 TB/s, 16 rows 2.32 ms: 18.5 KiB of LDS per block, eight blocks per CU in different phases)
#ifndef GC_ADV_ROWS
#define GC_ADV_ROWS 4
#endif
constexpr int kAdvRows = GC_ADV_ROWS;
__device__ __forceinline__ void adv_copy(float* dst, const float* src, int n, bool vec, int t) {
  if (vec) {
    const int n4 = n >> 2;
#pragma unroll 4
    for (int e = t; e < n4; e += 256) reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(src)[e];
    for (int e = (n4 << 2) + t; e < n; e += 256) dst[e] = src[e];
  } else {
    for (int e = t; e < n; e += 256) dst[e] = src[e];
  }
}
constexpr int kAdvMaxJ = 4;         // channels per thread: c_in, c_out <= 1024
__global__ __launch_bounds__(256) void advance_state_kernel(const gc_advance_desc d, int vec) {
  extern __shared__ __attribute__((aligned(16))) float adv_smem[];
  const int t = threadIdx.x;
  const int row0 = blockIdx.x * kAdvRows;
  const int rows = d.n_rows - row0 < kAdvRows ? d.n_rows - row0 : kAdvRows;
  const int xpad = (kAdvRows * d.c_in + 3) & ~3, ypad = (kAdvRows * d.c_out + 3) & ~3;
  float* xs = adv_smem;                 // [rows, c_in]   the current state
  float* xo = xs + xpad;                // [rows, c_in]   the next state
  float* ys = xo + xpad;                // [rows, c_out]  the step's output, then (in place) the prediction
  // this thread's table entries first: their round trip runs under the copies
  int sx[kAdvMaxJ], sy[kAdvMaxJ], sf[kAdvMaxJ], psx[kAdvMaxJ];
  float ax[kAdvMaxJ], ay[kAdvMaxJ], pax[kAdvMaxJ], pay[kAdvMaxJ], pb[kAdvMaxJ];
#pragma unroll
  for (int j = 0; j < kAdvMaxJ; ++j) {
    const int c = t + 256 * j;
    sx[j] = sy[j] = sf[j] = psx[j] = -1;
    ax[j] = ay[j] = pax[j] = pay[j] = pb[j] = 0.f;
    if (c < d.c_in) {
      sx[j] = d.src_x[c]; sy[j] = d.src_y[c]; sf[j] = d.src_f[c];
      ax[j] = d.ax[c]; ay[j] = d.ay[c];
    }
    if (d.pred && c < d.c_out) {
      psx[j] = d.p_src_x[c]; pax[j] = d.p_ax[c]; pay[j] = d.p_ay[c]; pb[j] = d.p_b[c];
    }
  }
  adv_copy(xs, d.x + (size_t)row0 * d.c_in, rows * d.c_in, vec & 1, t);
  adv_copy(ys, d.y + (size_t)row0 * d.c_out, rows * d.c_out, vec & 2, t);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kAdvMaxJ; ++j) {
    const int c = t + 256 * j;
    if (c < d.c_in) {
      const float* pf = nullptr;
      if (sf[j] >= 0)
        pf = (sf[j] < d.n_forc ? d.f_cur + sf[j] : d.f_next + (sf[j] - d.n_forc)) + (size_t)row0 * d.n_forc;
      for (int r = 0; r < rows; ++r) {
        float v = 0.f;
        if (sx[j] >= 0) v = ax[j] * xs[r * d.c_in + sx[j]];
        if (sy[j] >= 0) v = fmaf(ay[j], ys[r * d.c_out + sy[j]], v);
        if (pf) v += pf[(size_t)r * d.n_forc];
        xo[r * d.c_in + c] = v;
      }
    }
  }
  __syncthreads();                      // every pick of y is done: the prediction may overwrite it
  adv_copy(d.x_next + (size_t)row0 * d.c_in, xo, rows * d.c_in, vec & 4, t);
  if (d.pred) {
#pragma unroll
    for (int j = 0; j < kAdvMaxJ; ++j) {
      const int k = t + 256 * j;
      if (k < d.c_out) {
        for (int r = 0; r < rows; ++r) {
          float v = fmaf(pay[j], ys[r * d.c_out + k], pb[j]);
          if (psx[j] >= 0) v = fmaf(pax[j], xs[r * d.c_in + psx[j]], v);
          ys[r * d.c_out + k] = v;
        }
      }
    }
    __syncthreads();
    adv_copy(d.pred + (size_t)row0 * d.c_out, ys, rows * d.c_out, vec & 8, t);
  }
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return GC_ELAUNCH;
  }
  return 0;
}

bool g_attr_set[3] = {false, false, false};

// GC_LAYOUT_CHUNKED = GC_PREC_F32: the exact-fp32 kernel (round 1's formulation; the chunked GC_PREC_F16X3 kernel and
// the GC_PREC_BF16_GEMM tier that shared it were retired in round 5).
template <int MODE>
int launch_rowmlp(const gc_rowmlp_desc& d, hipStream_t s) {
  const size_t lds = kLdsFloats * sizeof(float);
  if (!g_attr_set[MODE]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowmlp_kernel<MODE>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      std::snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
      return GC_ELAUNCH;
    }
    g_attr_set[MODE] = true;
  }
  const int tiles = (d.n_rows + GC_TILE_ROWS - 1) / GC_TILE_ROWS;
  hipLaunchKernelGGL(rowmlp_kernel<MODE>, dim3(tiles), dim3(256), lds, s, d);
  return check_launch("rowmlp_kernel");
}

bool g_h_attr_set[3][4] = {};

// ONE tuning surface (include/gcast.h: gc_tuning).  The GCAST_* environment variables initialise the process default
// the first time it is needed and are never read again; gc_set_tuning replaces it; gc_plan_create snapshots it.  No
// launch function below calls getenv.
gc_tuning tuning_from_env() {
  gc_tuning t;
  std::memset(&t, 0, sizeof(t));
  auto env_int = [](const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; };
  const int cap = env_int("GCAST_GRID_CAP", 0);
  t.grid_cap = cap > 0 && cap <= GC_SCRATCH_SLOTS ? cap : GC_SCRATCH_SLOTS;
  { const char* e = std::getenv("GCAST_TILE_MAP"); t.tile_map_xcd = e && std::strcmp(e, "xcd") == 0; }
  t.prio_gemm = GC_PRIO_GEMM_DEFAULT; t.prio_other = GC_PRIO_OTHER_DEFAULT; t.prio_stage = GC_PRIO_STAGE_DEFAULT;
  if (const char* e = std::getenv("GCAST_PRIO")) {
    t.prio_set = 1;
    t.prio_gemm = t.prio_other = t.prio_stage = 0;
    std::sscanf(e, "%d,%d,%d", &t.prio_gemm, &t.prio_other, &t.prio_stage);
    t.prio_gemm &= 3; t.prio_other &= 3; t.prio_stage &= 3;
  }
  t.helpers = std::getenv("GCAST_HELPERS") ? (env_int("GCAST_HELPERS", 0) != 0) : (GC_HELPERS_DEFAULT ? 1 : -1);
  t.helpers_small = env_int("GCAST_HELPERS_SMALL", 1) != 0;
  t.helpers_edge = std::getenv("GCAST_HELPERS_EDGE") ? (env_int("GCAST_HELPERS_EDGE", 0) != 0 ? 2 : 0) : 1;
  { const int v = env_int("GCAST_HELPER_STORE", 2); t.helper_store = v < 0 ? 0 : v > 2 ? 2 : v; }
  { const int v = env_int("GCAST_HELPERS_MIN_ROWS", GC_HELPERS_MIN_ROWS_DEFAULT); t.helpers_min_rows = v < 0 ? 0 : v; }
  t.wide = env_int("GCAST_WIDE", 1) != 0;
  t.wide_edges = env_int("GCAST_WIDE_EDGES", GC_WIDE_EDGES_DEFAULT) & 3;
  { const int v = env_int("GCAST_BF16_ROWS", 0); t.bf16_rows = (v == 64 || v == 128) ? v : 0; }
  t.tile_queue = env_int("GCAST_TILE_QUEUE", 1) != 0;
  { const char* e = std::getenv("GCAST_FUSE"); t.fuse = !(e && std::strcmp(e, "0") == 0); }
  { const char* e = std::getenv("GCAST_ONEPASS"); t.onepass = !(e && std::strcmp(e, "0") == 0); }
  t.split_tail = env_int("GCAST_SPLIT_TAIL", GC_SPLIT_TAIL_DEFAULT) != 0;
  t.bf16_stream = env_int("GCAST_BF16_STREAM", GC_BF16_STREAM_DEFAULT) != 0;
  t.wide_late = env_int("GCAST_WIDE_LATE", GC_WIDE_LATE_DEFAULT) != 0;
  t.split_edges = env_int("GCAST_SPLIT_EDGES", GC_SPLIT_EDGES_DEFAULT) != 0;
  return t;
}
gc_tuning& tuning_mut() {
  static gc_tuning t = tuning_from_env();
  return t;
}
inline const gc_tuning& tuning() { return tuning_mut(); }
bool tuning_valid(const gc_tuning& t) {
  auto b = [](int v) { return v == 0 || v == 1; };
  return t.grid_cap >= 1 && t.grid_cap <= GC_SCRATCH_SLOTS && b(t.tile_map_xcd) && b(t.prio_set) && !(t.prio_gemm & ~3) &&
         !(t.prio_other & ~3) && !(t.prio_stage & ~3) && t.helpers >= -1 && t.helpers <= 1 && b(t.helpers_small) &&
         t.helpers_edge >= 0 && t.helpers_edge <= 2 && t.helper_store >= 0 && t.helper_store <= 2 && t.helpers_min_rows >= 0 &&
         b(t.wide) && !(t.wide_edges & ~3) && (t.bf16_rows == 0 || t.bf16_rows == 64 || t.bf16_rows == 128) && b(t.tile_queue) &&
         b(t.fuse) && b(t.onepass) && b(t.split_tail) && b(t.bf16_stream) && b(t.wide_late) && b(t.split_edges);
}
int half_grid_cap() { return tuning().grid_cap; }
bool half_tile_xcd() { return tuning().tile_map_xcd != 0; }

// Wave priorities of a GC_LAYOUT_HALF launch that does not carry its own GC_PRIO bits: GEMM phases, the other phases,
// staging waves (include/gcast.h GC_PRIO).  Default (round 5, same-session A/B of the whole 0.25 deg step,
// profiles/r05_s2_*): GEMM phases at priority 1 in the f16x3 kernels -- processor edge update 19.91 -> 19.43 ms per
// step, whole step 50.73 -> 50.47 ms; priorities 2 / 3 and a raised priority OUTSIDE the GEMM phases measured within
// noise of the default, the helper-wave form and the bf16 tier do not react at all (their defaults stay 0: the tuning's
// three values reach the bf16 kernels only when they were given explicitly, gc_tuning.prio_set).
int half_prio_flags(bool bf16) {
  const gc_tuning& t = tuning();
  if (bf16 && !t.prio_set) return GC_PRIO(0, 0, 0);
  return GC_PRIO(t.prio_gemm, t.prio_other, t.prio_stage);
}
inline void apply_prio(gc_rowmlp_desc& dd, bool bf16 = false) {
  if (!((dd.flags >> GC_PRIO_SHIFT) & 63)) dd.flags |= half_prio_flags(bf16);
}

template <int MODE, int ONEPASS>
int launch_rowmlp_half_d(const gc_rowmlp_desc& d, hipStream_t s);
int half_helpers_default();

// gc_rowmlp_desc.tile_queue: from GC_TILE_QUEUE_MIN_ROUNDS tiles per workgroup on (GC_TILE_QUEUE_ANY: whenever there is
// a second round at all).  With fewer the static walk is the better schedule: its few second-round tiles land on
// DISTINCT CUs (workgroups b and b + 256 share one), where a lone workgroup runs at 0.6 of the pair's tile time; the
// queue hands them to whoever finishes first -- often both workgroups of one CU.
inline bool tile_queue_pays(const gc_rowmlp_desc& d, int tiles, int grid) {
  return d.tile_queue && (tiles >= GC_TILE_QUEUE_MIN_ROUNDS * grid || ((d.flags & GC_TILE_QUEUE_ANY) && tiles > grid));
}

template <int MODE, int ONEPASS, int LATE = 0>
int launch_rowmlp_half_w(const gc_rowmlp_desc& d, hipStream_t s);
inline void apply_prio(gc_rowmlp_desc& dd, bool bf16);

template <int MODE, int ONEPASS = 0>
int launch_rowmlp_half(const gc_rowmlp_desc& d, hipStream_t s) {
  // The wide form (csrc/rowmlp_half.inc: rowmlp16w_kernel; eight multiplying waves per CU on ONE weight ring): asked for
  // per launch with GC_WG_WIDE -- the plan marks the big launches without gather or segment-sum (gcast_plan.inc: op_mlp).
  // Two-pass GC_MODE_MLP_LN launches without a segment-sum only: elsewhere the flag is ignored (a speed choice).
  // Round 6: the form also takes launches with a segment-sum (two 64-row sub-tiles per workgroup) and the one-pass edge
  // updates; gc_tuning.wide_edges asks for it on edge updates of at least GC_WIDE_EDGE_MIN_TILES tiles that do not pin
  // another form (bit 0: one-pass, bit 1: two-pass).
  // Round 6 (gc_tuning.split_edges): the processor's edge update of a SMALL graph -- 1 deg: 1,280 tiles, a rank of the
  // 8-way partition at 0.25 deg: 649 -- sits between the sizes the rules below were measured for: in the helper form it
  // is ceil(tiles / 256) rounds of one 64-row tile per CU (64 us each), in the wide form ceil(tiles / 512) rounds of
  // one 128-row tile per CU (104 us each: fewer joules AND less time per row, but a last round that is mostly empty).
  // Hence: every FULL round of 256 wide tiles in the wide form, and what is left -- if it is at most one tile per CU --
  // as a second launch in the helper form, the faster form of a lone tile (section 9.7); a remainder of more than 256
  // tiles is a wide round of its own.  Tiles are independent (a tile's outputs, straddle partials included, are
  // addressed by the tile) and both forms give the same bits as the four-wave kernel: nothing changes in the results.
  // Applies to launches of the shape the helpers_edge rule takes (two-pass, b1 + g0 + g1, segment-sum, rows stored)
  // that pin no form, above 512 tiles and below the wide_edges rule's GC_WIDE_EDGE_MIN_TILES.
  // MEASURED (profiles/r06_s16_*, same session, alternating): slower -- 1 deg step 8.38 / 8.39 against 8.09 / 8.09 ms, an
  // emulated 8-way rank 8.58 / 8.57 against 8.25 / 8.28 ms; the launch itself 5.13 against 5.08 ms per 1 deg step: a wide
  // tile of this launch WITHOUT late addends is no faster per row than two helper-form tiles, and the second launch
  // adds its own fill and drain.  Off by default (gc_tuning.split_edges = 0); kept as the measured alternative.
  if constexpr (MODE == GC_MODE_MLP_LN && ONEPASS == 0) {
    const gc_tuning& T = tuning();
    const int t64 = (d.n_rows + kHRows - 1) / kHRows;
    if (T.split_edges && T.helpers == -1 && T.helpers_edge == 1 && (T.wide_edges & 2) && d.seg && d.out && d.g0 && d.g1 &&
        !d.d && d.b1 && d.k0 + d.k1 > 0 && !(d.flags & (GC_WG_WIDE | GC_WG_HELPERS | GC_WG_NO_HELPERS | GC_LATE_ADDENDS)) &&
        t64 > GC_SCRATCH_SLOTS && t64 < GC_WIDE_EDGE_MIN_TILES && d.n_rows % kHRows == 0) {
      const int head_tiles = t64 / GC_SCRATCH_SLOTS * GC_SCRATCH_SLOTS, tail_tiles = t64 - head_tiles;
      gc_rowmlp_desc head = d;
      head.flags |= GC_WG_WIDE;
      if (tail_tiles == 0 || tail_tiles > GC_SCRATCH_SLOTS / 2) return launch_rowmlp_half_w<MODE, ONEPASS>(head, s);
      const int head_rows = head_tiles * kHRows;
      gc_rowmlp_desc tail = d;
      head.n_rows = head_rows;
      tail.n_rows = d.n_rows - head_rows;
      tail.flags |= GC_WG_HELPERS;
      auto rows = [&](const float* p, int ld) { return p ? p + (size_t)head_rows * ld : p; };
      tail.a0 = rows(d.a0, d.lda0); tail.a1 = rows(d.a1, d.lda1);
      tail.res = rows(d.res, d.ldres); tail.out = const_cast<float*>(rows(d.out, d.ldo));
      tail.idx0 = d.idx0 + head_rows; tail.idx1 = d.idx1 + head_rows;
      tail.seg = d.seg + head_rows; tail.tile_flags = d.tile_flags + head_tiles;
      if (d.partial) tail.partial = d.partial + (size_t)2 * head_tiles * kD;
      if (const int rc = launch_rowmlp_half_w<MODE, ONEPASS>(head, s)) return rc;
      return launch_rowmlp_half_d<MODE, ONEPASS>(tail, s);
    }
  }
  if constexpr (MODE == GC_MODE_MLP_LN) {
    const bool wide_edge = d.seg && !(d.flags & (GC_WG_HELPERS | GC_WG_NO_HELPERS)) &&
                           (tuning().wide_edges & (ONEPASS != 0 ? 1 : 2)) && tuning().helpers != 1 &&
                           (d.n_rows + kHRows - 1) / kHRows >= GC_WIDE_EDGE_MIN_TILES;
    // GC_LATE_ADDENDS (round 6): a two-pass edge update that adds its gathered rows when the hidden layer is formed -- an fp32
    // ASSOCIATION of its own, so it is a property of the launch: asked for by the flag (honoured by the wide and the
    // four-wave form, same bits in both), and by gc_tuning.wide_late for the launches the wide_edges RULE puts into the
    // wide form (where the gather has nothing to run under: -1.5 % of the step).  A launch pinned to a form by its own
    // flags keeps the association it asks for.
    const bool late_shape = ONEPASS == 0 && d.seg && d.g0 && !d.d && d.k0 + d.k1 > 0;
    const bool late = late_shape && ((d.flags & GC_LATE_ADDENDS) || (wide_edge && !(d.flags & GC_WG_WIDE) && tuning().wide_late));
    if ((d.flags & GC_WG_WIDE) || wide_edge) {
      if constexpr (ONEPASS == 0) {
        if (late) return launch_rowmlp_half_w<MODE, ONEPASS, 1>(d, s);
      }
      return launch_rowmlp_half_w<MODE, ONEPASS>(d, s);
    }
    if constexpr (ONEPASS == 0) {
      if (late) {               // (the four-wave form of the same association: what the tests compare the wide form with)
        static bool attr_set = false;
        const size_t lds = kHLdsFloats * sizeof(float);
        if (!attr_set) {
          const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowmlp16h_kernel<MODE, 0, 1>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          if (e != hipSuccess) {
            std::snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
            return GC_ELAUNCH;
          }
          attr_set = true;
        }
        const int tiles = (d.n_rows + kHRows - 1) / kHRows;
        const int cap = half_grid_cap();
        const int grid = tiles < cap ? tiles : cap;
        gc_rowmlp_desc dd = d;
        if (!tile_queue_pays(dd, tiles, grid)) dd.tile_queue = nullptr;
        apply_prio(dd);
        hipLaunchKernelGGL((rowmlp16h_kernel<MODE, 0, 1>), dim3(grid), dim3(256), lds, s, dd);
        return check_launch("rowmlp16h_kernel");
      }
    }
  }
  // Round 6: a node-side launch of 513 .. 768 tiles (the 0.25 deg processor's node updates: 641) is 1.6 rounds of
  // four-wave pairs -- a full round, then 129 lone workgroups on 129 CUs while the other 127 idle, each lone tile at
  // 0.6 of the pair's time.  Split in two launches: the first 512 tiles as ONE full round of 256 wide tiles (the wide
  // form's rate: ~7 % faster per row), the remaining <= 256 tiles in the helper form (the faster form of a LONE tile:
  // 142 k against 153 k cycles).  Rows are independent and every form gives the same bits.
  if constexpr (MODE == GC_MODE_MLP_LN && ONEPASS == 0) {
    const int t64 = (d.n_rows + kHRows - 1) / kHRows;
    if (tuning().split_tail && tuning().helpers == -1 && !d.seg && !d.g0 && !d.g1 && d.k0 + d.k1 > 0 &&
        !(d.flags & (GC_WG_WIDE | GC_WG_HELPERS | GC_WG_NO_HELPERS)) && t64 > GC_SCRATCH_SLOTS && t64 <= GC_SCRATCH_SLOTS * 3 / 2) {
      const int head_rows = GC_SCRATCH_SLOTS * kHRows;
      gc_rowmlp_desc head = d, tail = d;
      head.n_rows = head_rows;
      head.flags |= GC_WG_WIDE;
      tail.n_rows = d.n_rows - head_rows;
      tail.flags |= GC_WG_HELPERS;
      auto rows = [&](const float* p, int ld) { return p ? p + (size_t)head_rows * ld : p; };
      tail.a0 = rows(d.a0, d.lda0); tail.a1 = rows(d.a1, d.lda1); tail.d = rows(d.d, d.ldd);
      tail.res = rows(d.res, d.ldres); tail.out = const_cast<float*>(rows(d.out, d.ldo));
      for (int k = 0; k < d.n_chain; ++k)
        if (d.chain[k].out) tail.chain[k].out = d.chain[k].out + (size_t)head_rows * d.chain[k].ldo;
      if (const int rc = launch_rowmlp_half_w<MODE, ONEPASS>(head, s)) return rc;
      return launch_rowmlp_half_d<MODE, ONEPASS>(tail, s);
    }
  }
  // Which form.  Asked for per launch (GC_WG_HELPERS / GC_WG_NO_HELPERS), or per process (GCAST_HELPERS); otherwise:
  // a launch of at most one tile per CU runs one four-wave workgroup per CU anyway -- for the node-side launches (no
  // gather, no segment-sum) the eight-wave form with its staging waves is faster per lone tile (1 deg step: processor
  // node updates 2.24 -> 2.11 ms, profiles/r04_s13_*; the edge updates are NOT: 5.23 -> 5.37), so small node-side
  // launches (small grids, the 8-way partition's ranks) take it; GCAST_HELPERS_SMALL=0 switches the rule off (A/B).
  // GCAST_HELPERS=0 (the per-process "four-wave form everywhere" switch) turns it off as well; GC_WG_NO_HELPERS pins
  // the four-wave form per launch.
  const gc_tuning& T = tuning();
  const bool small_rule = T.helpers_small && T.helpers != 0;
  const bool small = small_rule && !d.g0 && !d.seg && (d.n_rows + kHRows - 1) / kHRows <= GC_SCRATCH_SLOTS / 2;
  // Round 5: the processor's edge update from step 1 on (two-pass, b1 + g0 + g1, segment-sum, rows stored) in the
  // eight-wave form -- its staging waves take residual + store and gather the next tile's addends (rowmlp_half.inc:
  // HST == 2), the same bits.  The launch is faster in it at every size (0.25 deg: 19.41 against 19.87 ms per step,
  // profiles/r05_s11_*), but only SMALL launches turn that into a faster step: 1 deg (1,280 tiles) 8.69 -> 8.33 ms per
  // step, an 8-way rank of the 0.25 deg partition (640 tiles) 11.79 -> 11.49 ms (profiles/r05_s14_*).  At the headline
  // size (5,120 tiles) the part runs at its power limit -- 1367 W at 1.94 GHz, profiles/r05_s12_* -- and the launch
  // behind a faster launch slows down by what was gained (5.69 -> 6.64 ms per step; whole step 53.02 -> 53.22 ms):
  // DESIGN.md section 9.14.  Hence the rule: by default for launches of more than one and at most
  // GC_HELPERS_EDGE_MAX_TILES tiles per ... launch; gc_tuning.helpers_edge = 2 at every size, 0 (or helpers = 0) never.
  const int edge_rule = T.helpers == 0 ? 0 : T.helpers_edge;
  const int edge_tiles = (d.n_rows + kHRows - 1) / kHRows;
  const bool edge = edge_rule != 0 && MODE == GC_MODE_MLP_LN && ONEPASS == 0 && d.seg && d.out && d.g0 && d.g1 && !d.d && d.b1 &&
                    d.k0 + d.k1 > 0 && edge_tiles > GC_SCRATCH_SLOTS / 2 && (edge_rule == 2 || edge_tiles <= GC_HELPERS_EDGE_MAX_TILES);
  if ((d.flags & GC_WG_HELPERS) || (!(d.flags & GC_WG_NO_HELPERS) && (half_helpers_default() || small || edge)))
    return launch_rowmlp_half_d<MODE, ONEPASS>(d, s);
  const size_t lds = kHLdsFloats * sizeof(float);
  if (!g_h_attr_set[MODE][ONEPASS]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowmlp16h_kernel<MODE, ONEPASS>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      std::snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
      return GC_ELAUNCH;
    }
    g_h_attr_set[MODE][ONEPASS] = true;
  }
  // persistent workgroups: two per CU on the 256 CUs of an MI355X, each walking tiles b, b + grid, ...
  const int tiles = (d.n_rows + kHRows - 1) / kHRows;
  const int cap = half_grid_cap();
  const int grid = tiles < cap ? tiles : cap;
  gc_rowmlp_desc dd = d;
  if (half_tile_xcd()) dd.flags |= GC_TILE_XCD;
  if (!tile_queue_pays(dd, tiles, grid)) dd.tile_queue = nullptr;
  apply_prio(dd);
  hipLaunchKernelGGL((rowmlp16h_kernel<MODE, ONEPASS>), dim3(grid), dim3(256), lds, s, dd);
  return check_launch("rowmlp16h_kernel");
}

// The eight-wave "helper waves" form of the same launch (csrc/rowmlp_half.inc: rowmlp16d_kernel): ONE persistent
// workgroup per CU.  gc_rowmlp_desc.flags GC_WG_HELPERS asks for it per launch, gc_tuning.helpers = 1 | 0 for every
// launch that does not pin a form.
int half_helpers_default() { return tuning().helpers == 1; }

template <int MODE, int ONEPASS, int HST>
int launch_rowmlp_half_d2(const gc_rowmlp_desc& d, hipStream_t s) {
  const size_t lds = (kHLdsFloats + kHSlotFloats) * sizeof(float);     // + the parked accumulators (LDS, not the scratch slots)
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowmlp16d_kernel<MODE, ONEPASS, HST>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      std::snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
      return GC_ELAUNCH;
    }
    attr_set = true;
  }
  const int tiles = (d.n_rows + kHRows - 1) / kHRows;
  const int cap = half_grid_cap() < GC_SCRATCH_SLOTS / 2 ? half_grid_cap() : GC_SCRATCH_SLOTS / 2;   // one workgroup per CU
  const int grid = tiles < cap ? tiles : cap;
  gc_rowmlp_desc dd = d;
  if (half_tile_xcd()) dd.flags |= GC_TILE_XCD;
  if (!tile_queue_pays(dd, tiles, grid)) dd.tile_queue = nullptr;
  apply_prio(dd);
  hipLaunchKernelGGL((rowmlp16d_kernel<MODE, ONEPASS, HST>), dim3(grid), dim3(512), lds, s, dd);
  return check_launch("rowmlp16d_kernel");
}

template <int MODE, int ONEPASS, int LATE>
int launch_rowmlp_half_w(const gc_rowmlp_desc& d, hipStream_t s) {
  const size_t lds = kWLdsFloats * sizeof(float);     // four-wave layout + the parking area (LDS share of the parked accumulators / second sub-tile)
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowmlp16w_kernel<MODE, ONEPASS, LATE>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      std::snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
      return GC_ELAUNCH;
    }
    attr_set = true;
  }
  const int tiles = (d.n_rows + 2 * kHRows - 1) / (2 * kHRows);          // 128-row tiles
  const int cap = half_grid_cap() < GC_SCRATCH_SLOTS / 2 ? half_grid_cap() : GC_SCRATCH_SLOTS / 2;   // one workgroup per CU, two slots each
  const int grid = tiles < cap ? tiles : cap;
  gc_rowmlp_desc dd = d;
  dd.flags &= ~GC_TILE_XCD;
  if (!tile_queue_pays(dd, tiles, grid)) dd.tile_queue = nullptr;
  apply_prio(dd);
  hipLaunchKernelGGL((rowmlp16w_kernel<MODE, ONEPASS, LATE>), dim3(grid), dim3(512), lds, s, dd);
  return check_launch("rowmlp16w_kernel");
}

// HST (round 5): an edge update that STORES its rows (segment-sum + out) hands residual + store to the staging waves
// (csrc/rowmlp_half.inc: hstore) -- its own instantiation, chosen here.
template <int MODE, int ONEPASS = 0>
int launch_rowmlp_half_d(const gc_rowmlp_desc& d, hipStream_t s) {
  if constexpr (MODE == GC_MODE_MLP_LN) {
    // gc_tuning.helper_store = 0|1|2 (A/B): 0 = the staging waves only stage, 1 = + residual / store, 2 (default)
    // = + the gather of the next tile's addends, where the launch has the shape for it (two-pass, b1 + g0 + g1)
    const int hst_max = tuning().helper_store;
    if (hst_max >= 1 && d.seg && d.out) {
      if constexpr (ONEPASS == 0)
        if (hst_max >= 2 && d.g0 && d.g1 && !d.d && d.b1 && d.k0 + d.k1 > 0)
          return launch_rowmlp_half_d2<MODE, ONEPASS, 2>(d, s);
      return launch_rowmlp_half_d2<MODE, ONEPASS, 1>(d, s);
    }
  }
  return launch_rowmlp_half_d2<MODE, ONEPASS, 0>(d, s);
}

// Rows per workgroup of a GC_PREC_BF16 launch: 64 (two workgroups per CU), or 128 (eight waves sharing
// one weight stream, one workgroup per CU) for the big node-side launches -- no gather, no segment-sum,
// at least kBfWideMinRows rows: measured 6-7 % faster there, 1-7 % slower on the edge updates
// (profiles/r03_s12_*).  gc_rowmlp_desc.flags GC_WG_ROWS_64 / _128 pin the choice per launch,
// gc_tuning.bf16_rows = 64 | 128 for every launch that does not pin one (A/B runs of bench.py).
constexpr int kBfWideMinRows = 128 * 256 * 2;
int bf16_rows_override() { return tuning().bf16_rows; }

template <bool F32ROWS, int NW, int VAR>
int launch_rowmlp_bf16(const gc_rowmlp_desc& d, hipStream_t s) {
  const size_t lds = BfLds<NW>::kFloats * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowmlpbf_kernel<F32ROWS, NW, VAR>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      std::snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
      return GC_ELAUNCH;
    }
    attr_set = true;
  }
  const int tiles = (d.n_rows + 16 * NW - 1) / (16 * NW);
  const int slots = GC_SCRATCH_SLOTS * 4 / NW;        // persistent: 8 / NW workgroups per CU
  gc_rowmlp_desc dd = d;
  if (!tile_queue_pays(dd, tiles, tiles < slots ? tiles : slots)) dd.tile_queue = nullptr;
  apply_prio(dd, true);
  hipLaunchKernelGGL((rowmlpbf_kernel<F32ROWS, NW, VAR>), dim3(tiles < slots ? tiles : slots), dim3(64 * NW), lds, s, dd);
  return check_launch("rowmlpbf_kernel");
}

template <bool F32ROWS>
int launch_rowmlp_bf16(const gc_rowmlp_desc& d, hipStream_t s) {
  const int pin = (d.flags & GC_WG_ROWS_128) ? 128 : (d.flags & GC_WG_ROWS_64) ? 64 : bf16_rows_override();
  const bool wide = pin == 128 || (pin != 64 && d.n_rows >= kBfWideMinRows && !d.seg && !d.g0 && !d.g1);
  // Round 6 (gc_tuning.bf16_stream): an edge update without a layer-1 GEMM -- the whole first layer folded into addend
  // rows -- forms every K step's hidden pair on the fly instead of gathering up front (rowmlp_bf16.inc: STREAM).
  if constexpr (!F32ROWS) {
    if (tuning().bf16_stream && d.k0 + d.k1 == 0 && d.g0 && d.n_chain == 0)
      return wide ? launch_rowmlp_bf16<F32ROWS, 8, 1>(d, s) : launch_rowmlp_bf16<F32ROWS, 4, 1>(d, s);
  }
  return wide ? launch_rowmlp_bf16<F32ROWS, 8, 0>(d, s) : launch_rowmlp_bf16<F32ROWS, 4, 0>(d, s);
}

bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// GC_PREC_BF16: MLP_LN launches only (the step's fused program has no others), rows bfloat16 in pi order.
int rowmlp_bf16(const gc_rowmlp_desc& d, hipStream_t s) {
  if (d.layout != GC_LAYOUT_HALF || d.mode != GC_MODE_MLP_LN)
    return fail(GC_EINVAL, "gc_rowmlp: GC_PREC_BF16 is built for GC_LAYOUT_HALF + GC_MODE_MLP_LN launches");
  if (d.n_rows <= 0) return fail(GC_EINVAL, "gc_rowmlp: n_rows must be positive");
  if ((d.w1_scale != 0.f && d.w1_scale != 1.f) || (d.w2_scale != 0.f && d.w2_scale != 1.f))
    return fail(GC_EINVAL, "gc_rowmlp: weight scales are not a GC_PREC_BF16 feature");
  if ((d.k0 | d.k1) & 31 || d.k0 < 0 || d.k1 < 0) return fail(GC_EINVAL, "gc_rowmlp: k0/k1 must be multiples of 32");
  if (d.k0 == 0 && d.k1 != 0) return fail(GC_EINVAL, "gc_rowmlp: k1 without k0");
  if (d.k0 + d.k1 > 0 && (!d.a0 || !d.w1p)) return fail(GC_EINVAL, "gc_rowmlp: layer-1 GEMM needs a0 and w1p");
  if (d.k1 && !d.a1) return fail(GC_EINVAL, "gc_rowmlp: k1 > 0 needs a1");
  const bool f32rows = (d.flags & GC_ROWS_F32) != 0;
  if (!f32rows && ((d.k0 && ((d.lda0 & 7) || !aligned16(d.a0))) || (d.k1 && ((d.lda1 & 7) || !aligned16(d.a1)))))
    return fail(GC_EINVAL, "gc_rowmlp: bfloat16 rows must be 16-byte aligned (strides multiples of 8 elements)");
  if ((d.d && ((d.ldd & 7) || !aligned16(d.d))) || !aligned16(d.g0) || !aligned16(d.g1) || !aligned16(d.b1) ||
      !aligned16(d.w1p) || !aligned16(d.w2p) || !aligned16(d.b2) || !aligned16(d.ln_scale) || !aligned16(d.ln_offset) ||
      (d.res && ((d.ldres & 7) || !aligned16(d.res))) || (d.out && ((d.ldo & 7) || !aligned16(d.out))) ||
      !aligned16(d.agg) || !aligned16(d.partial))
    return fail(GC_EINVAL, "gc_rowmlp: pointers must be 16-byte aligned, bfloat16 row strides multiples of 8");
  if ((d.g0 && !d.idx0) || (d.g1 && !d.idx1)) return fail(GC_EINVAL, "gc_rowmlp: gather without index array");
  if (d.k0 + d.k1 == 0 && !d.d && !d.g0 && !d.g1) return fail(GC_EINVAL, "gc_rowmlp: no layer-1 input at all");
  if (!d.w2p || !d.b2 || d.n2 != kD) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: needs w2p, b2, n2 == 512");
  if (d.ln_scale && !d.ln_offset) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: ln_scale without ln_offset");
  if (d.res && !d.out && d.n_chain == 0) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: residual needs out (or a chain)");
  if (d.n_chain < 0 || d.n_chain > GC_MAX_CHAIN) return fail(GC_EINVAL, "gc_rowmlp: n_chain out of range");
  if (d.seg) {
    if (d.n_chain) return fail(GC_EINVAL, "gc_rowmlp: a chain needs a launch without segment-sum");
    if (d.n_rows % GC_TILE_ROWS) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: segment-sum needs n_rows % 64 == 0");
    if (!d.tile_flags || !d.agg || !d.partial) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: segment-sum needs tile_flags, agg, partial");
  } else if (!d.out && d.n_chain == 0) {
    return fail(GC_EINVAL, "gc_rowmlp MLP_LN: nothing to produce (no out, no seg, no chain)");
  }
  for (int k = 0; k < d.n_chain; ++k) {
    const gc_chain_stage& c = d.chain[k];
    if (!c.wp || !aligned16(c.wp) || !aligned16(c.b) || (c.w_scale != 0.f && c.w_scale != 1.f))
      return fail(GC_EINVAL, "gc_rowmlp: chain stage needs 16-byte aligned wp / b (and no weight scale in GC_PREC_BF16)");
    if (c.kind == GC_CHAIN_ROWS) {
      if (!c.out || !aligned16(c.out) || (c.ldo & 7)) return fail(GC_EINVAL, "gc_rowmlp: GC_CHAIN_ROWS needs a 16-byte aligned out, ldo % 8 == 0");
    } else if (c.kind == GC_CHAIN_NARROW) {
      if (!c.out || c.n <= 0 || c.n > 240) return fail(GC_EINVAL, "gc_rowmlp: GC_CHAIN_NARROW needs out and 0 < n <= 240");
    } else if (c.kind == GC_CHAIN_SWISH) {
      if (k + 1 == d.n_chain) return fail(GC_EINVAL, "gc_rowmlp: GC_CHAIN_SWISH must feed a following stage");
    } else {
      return fail(GC_EINVAL, "gc_rowmlp: unknown chain kind");
    }
  }
  return f32rows ? launch_rowmlp_bf16<true>(d, s) : launch_rowmlp_bf16<false>(d, s);
}

}  // namespace

extern "C" {

static bool pow2_or_unset(float s) {
  if (s == 0.f) return true;
  int e = 0;
  return s > 0.f && std::frexp(s, &e) == 0.5f;
}

int gc_rowmlp(const gc_rowmlp_desc* dp, void* stream) {
  if (!dp) return fail(GC_EINVAL, "gc_rowmlp: null descriptor");
  gc_rowmlp_desc d = *dp;
  // gc_rowmlp_desc.tile_queue: the persistent kernels' dynamic tile queue (GC_LAYOUT_HALF launches and the
  // GC_PREC_BF16 tier); gc_tuning.tile_queue = 0 walks the tiles statically whatever the descriptor says (A/B).
  const bool queue_off = !tuning().tile_queue;
  if (d.tile_queue && (reinterpret_cast<size_t>(d.tile_queue) & 7))
    return fail(GC_EINVAL, "gc_rowmlp: tile_queue must be an 8-byte aligned pair of device words");
  if (d.tile_queue && d.prec != GC_PREC_BF16 && d.layout != GC_LAYOUT_HALF)
    return fail(GC_EINVAL, "gc_rowmlp: tile_queue is a feature of the persistent kernels (GC_LAYOUT_HALF, GC_PREC_BF16)");
  if (queue_off) d.tile_queue = nullptr;
  if (d.prec == GC_PREC_BF16) return rowmlp_bf16(d, static_cast<hipStream_t>(stream));
  if (!pow2_or_unset(d.w1_scale) || !pow2_or_unset(d.w2_scale))
    return fail(GC_EINVAL, "gc_rowmlp: w1_scale / w2_scale must be powers of two");
  if (d.w1_scale == 0.f || d.k0 + d.k1 == 0) d.w1_scale = 1.f;   // (no layer-1 weights: nothing is scaled)
  if (d.w2_scale == 0.f) d.w2_scale = 1.f;
  if (d.prec == GC_PREC_F32 && (d.w1_scale != 1.f || d.w2_scale != 1.f))
    return fail(GC_EINVAL, "gc_rowmlp: weight scales are not a GC_PREC_F32 feature");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d.n_rows <= 0) return fail(GC_EINVAL, "gc_rowmlp: n_rows must be positive");
  if (d.prec != GC_PREC_F32 && d.prec != GC_PREC_F16X3)   // (GC_PREC_BF16: above; 2 = the retired GC_PREC_BF16_GEMM tier)
    return fail(GC_EINVAL, "gc_rowmlp: unknown precision");
  if (d.layout != GC_LAYOUT_CHUNKED && d.layout != GC_LAYOUT_HALF)
    return fail(GC_EINVAL, "gc_rowmlp: unknown weight layout");
  if ((d.prec == GC_PREC_F16X3) != (d.layout == GC_LAYOUT_HALF))
    return fail(GC_EINVAL, "gc_rowmlp: GC_PREC_F16X3 runs in GC_LAYOUT_HALF, GC_PREC_F32 in GC_LAYOUT_CHUNKED (the chunked f16x3 "
                           "kernel of rounds 1-4 was retired)");
  if (d.layout == GC_LAYOUT_HALF) {
    if (d.prec != GC_PREC_F16X3) return fail(GC_EINVAL, "gc_rowmlp: GC_LAYOUT_HALF is built for GC_PREC_F16X3 only");
    if (d.mode == GC_MODE_MLP_LN && !(d.flags & GC_W2_NATURAL) && (!d.scratch || !aligned16(d.scratch)))
      return fail(GC_EINVAL, "gc_rowmlp: GC_LAYOUT_HALF + GC_MODE_MLP_LN needs a 16-byte aligned scratch");
  }
  if (d.range_flag && (d.layout != GC_LAYOUT_HALF || d.prec != GC_PREC_F16X3 || (reinterpret_cast<size_t>(d.range_flag) & 3)))
    return fail(GC_EINVAL, "gc_rowmlp: range_flag is a GC_PREC_F16X3 + GC_LAYOUT_HALF feature (4-byte aligned device word)");
  if (d.n_chain != 0) {
    if (d.layout != GC_LAYOUT_HALF || d.mode != GC_MODE_MLP_LN || d.seg)
      return fail(GC_EINVAL, "gc_rowmlp: a chain needs GC_LAYOUT_HALF + GC_MODE_MLP_LN without segment-sum");
    if (d.n_chain < 0 || d.n_chain > GC_MAX_CHAIN) return fail(GC_EINVAL, "gc_rowmlp: n_chain out of range");
    for (int k = 0; k < d.n_chain; ++k) {
      const gc_chain_stage& c = d.chain[k];
      if (!c.wp || !aligned16(c.wp) || !aligned16(c.b) || !pow2_or_unset(c.w_scale) || c.w_scale == 0.f)
        return fail(GC_EINVAL, "gc_rowmlp: chain stage needs 16-byte aligned wp / b and a power-of-two w_scale");
      if (c.kind == GC_CHAIN_ROWS) {
        if (!c.out || !aligned16(c.out) || (c.ldo & 3)) return fail(GC_EINVAL, "gc_rowmlp: GC_CHAIN_ROWS needs a 16-byte aligned out, ldo % 4 == 0");
      } else if (c.kind == GC_CHAIN_NARROW) {
        if (!c.out || c.n <= 0 || c.n > 240) return fail(GC_EINVAL, "gc_rowmlp: GC_CHAIN_NARROW needs out and 0 < n <= 240");
      } else if (c.kind == GC_CHAIN_SWISH) {
        if (k + 1 == d.n_chain) return fail(GC_EINVAL, "gc_rowmlp: GC_CHAIN_SWISH must feed a following stage");
      } else {
        return fail(GC_EINVAL, "gc_rowmlp: unknown chain kind");
      }
    }
  }
  if ((d.k0 | d.k1) & 31 || d.k0 < 0 || d.k1 < 0) return fail(GC_EINVAL, "gc_rowmlp: k0/k1 must be multiples of 32");
  if (d.k0 == 0 && d.k1 != 0) return fail(GC_EINVAL, "gc_rowmlp: k1 without k0");
  if (d.k0 + d.k1 > 0 && (!d.a0 || !d.w1p)) return fail(GC_EINVAL, "gc_rowmlp: layer-1 GEMM needs a0 and w1p");
  if (d.k1 && !d.a1) return fail(GC_EINVAL, "gc_rowmlp: k1 > 0 needs a1");
  // (GC_LAYOUT_HALF reads its layer-1 rows with 4-byte aligned vector loads: any float row stride)
  const bool rows_any = d.layout == GC_LAYOUT_HALF;
  if ((!rows_any && ((d.k0 && (d.lda0 & 3)) || (d.k1 && (d.lda1 & 3)))) || (d.d && (d.ldd & 3)))
    return fail(GC_EINVAL, "gc_rowmlp: row strides must be multiples of 4 floats");
  if ((!rows_any && (!aligned16(d.a0) || !aligned16(d.a1))) || !aligned16(d.w1p) || !aligned16(d.d) || !aligned16(d.g0) ||
      !aligned16(d.g1) || !aligned16(d.b1) || !aligned16(d.w2p) || !aligned16(d.b2) ||
      !aligned16(d.ln_scale) || !aligned16(d.ln_offset) || !aligned16(d.res) ||
      !aligned16(d.agg) || !aligned16(d.partial))
    return fail(GC_EINVAL, "gc_rowmlp: pointers must be 16-byte aligned");
  if ((d.g0 && !d.idx0) || (d.g1 && !d.idx1)) return fail(GC_EINVAL, "gc_rowmlp: gather without index array");
  if (d.k0 + d.k1 == 0 && !d.d && !d.g0 && !d.g1) return fail(GC_EINVAL, "gc_rowmlp: no layer-1 input at all");
  switch (d.mode) {
    case GC_MODE_LINEAR:
      if (!d.out || (d.ldo & 3) || !aligned16(d.out)) return fail(GC_EINVAL, "gc_rowmlp LINEAR: out must be 16B aligned, ldo % 4 == 0");
      if (d.layout == GC_LAYOUT_HALF) return launch_rowmlp_half<GC_MODE_LINEAR>(d, s);
      return launch_rowmlp<GC_MODE_LINEAR>(d, s);
    case GC_MODE_MLP_LN:
      if (!d.w2p || !d.b2 || d.n2 != kD) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: needs w2p, b2, n2 == 512");
      if (d.ln_scale && !d.ln_offset) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: ln_scale without ln_offset");
      if (d.out && ((d.ldo & 3) || !aligned16(d.out))) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: out alignment");
      if (d.res && ((!d.out && d.n_chain == 0) || (d.ldres & 3)))
        return fail(GC_EINVAL, "gc_rowmlp MLP_LN: residual needs out (or a chain), ldres % 4 == 0");
      if (d.seg) {
        if (d.n_rows % GC_TILE_ROWS) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: segment-sum needs n_rows % 64 == 0");
        if (!d.tile_flags || !d.agg || !d.partial) return fail(GC_EINVAL, "gc_rowmlp MLP_LN: segment-sum needs tile_flags, agg, partial");
      } else if (!d.out && d.n_chain == 0) {
        return fail(GC_EINVAL, "gc_rowmlp MLP_LN: nothing to produce (no out, no seg, no chain)");
      }
      if (d.layout == GC_LAYOUT_HALF) {
        if (d.flags & GC_W2_NATURAL) {        // the one-pass formulation of a launch without a layer-1 GEMM
          if (d.k0 + d.k1 != 0 || !d.d || !d.g0 || d.n_chain != 0)
            return fail(GC_EINVAL, "gc_rowmlp: GC_W2_NATURAL needs k0 + k1 == 0, d and g0, no chain");
          return d.g1 ? launch_rowmlp_half<GC_MODE_MLP_LN, 3>(d, s) : launch_rowmlp_half<GC_MODE_MLP_LN, 2>(d, s);
        }
        return launch_rowmlp_half<GC_MODE_MLP_LN>(d, s);
      }
      if (d.flags & GC_W2_NATURAL) return fail(GC_EINVAL, "gc_rowmlp: GC_W2_NATURAL is a GC_LAYOUT_HALF feature");
      return launch_rowmlp<GC_MODE_MLP_LN>(d, s);
    case GC_MODE_MLP_OUT:
      if (!d.w2p || !d.b2 || d.n2 <= 0 || d.n2 > 240 || !d.out) return fail(GC_EINVAL, "gc_rowmlp MLP_OUT: needs w2p, b2, out, 0 < n2 <= 240");
      if (d.layout == GC_LAYOUT_HALF) return launch_rowmlp_half<GC_MODE_MLP_OUT>(d, s);
      return launch_rowmlp<GC_MODE_MLP_OUT>(d, s);
    default:
      return fail(GC_EINVAL, "gc_rowmlp: unknown mode");
  }
}

int gc_seg_fixup(int n, const int* recv, const int* t0, const int* t1, const float* partial,
                 float* agg, void* stream) {
  if (n < 0) return fail(GC_EINVAL, "gc_seg_fixup: negative count");
  if (n == 0) return 0;
  if (!recv || !t0 || !t1 || !partial || !agg) return fail(GC_EINVAL, "gc_seg_fixup: null pointer");
  hipLaunchKernelGGL(seg_fixup_kernel, dim3(n), dim3(kD / 4), 0, static_cast<hipStream_t>(stream), n,
                     recv, t0, t1, partial, agg);
  return check_launch("seg_fixup_kernel");
}

int gc_seg_fixup_bf16(int n, const int* recv, const int* t0, const int* t1, const float* partial, void* agg,
                      void* stream) {
  if (n < 0) return fail(GC_EINVAL, "gc_seg_fixup_bf16: negative count");
  if (n == 0) return 0;
  if (!recv || !t0 || !t1 || !partial || !agg) return fail(GC_EINVAL, "gc_seg_fixup_bf16: null pointer");
  hipLaunchKernelGGL(seg_fixup_bf16_kernel, dim3(n), dim3(kD / 2), 0, static_cast<hipStream_t>(stream), n, recv, t0,
                     t1, partial, static_cast<unsigned*>(agg));
  return check_launch("seg_fixup_bf16_kernel");
}

int gc_zero_rows_bf16(int n, const int* rows, void* agg, void* stream) {
  if (n < 0) return fail(GC_EINVAL, "gc_zero_rows_bf16: negative count");
  if (n == 0) return 0;
  if (!rows || !agg) return fail(GC_EINVAL, "gc_zero_rows_bf16: null pointer");
  hipLaunchKernelGGL(zero_rows_bf16_kernel, dim3(n), dim3(kD / 2), 0, static_cast<hipStream_t>(stream), n, rows,
                     static_cast<unsigned*>(agg));
  return check_launch("zero_rows_bf16_kernel");
}

int gc_zero_rows(int n, const int* rows, float* agg, void* stream) {
  if (n < 0) return fail(GC_EINVAL, "gc_zero_rows: negative count");
  if (n == 0) return 0;
  if (!rows || !agg) return fail(GC_EINVAL, "gc_zero_rows: null pointer");
  hipLaunchKernelGGL(zero_rows_kernel, dim3(n), dim3(kD / 4), 0, static_cast<hipStream_t>(stream), n, rows, agg);
  return check_launch("zero_rows_kernel");
}

int gc_add_rows(int n, const int* rows, const float* src, float* dst, void* stream) {
  if (n < 0) return fail(GC_EINVAL, "gc_add_rows: negative count");
  if (n == 0) return 0;
  if (!rows || !src || !dst) return fail(GC_EINVAL, "gc_add_rows: null pointer");
  hipLaunchKernelGGL(add_rows_kernel, dim3(n), dim3(kD / 4), 0, static_cast<hipStream_t>(stream), n, rows, src, dst);
  return check_launch("add_rows_kernel");
}

int gc_prep_grid_input(int n_rows, int batch, int b, int c_in, const float* x, int n_struct,
                       const float* node_struct, int kp, float* xin, void* stream) {
  if (n_rows <= 0 || batch <= 0 || b < 0 || b >= batch || c_in <= 0 || n_struct < 0 ||
      kp < c_in + n_struct || (kp & 31))
    return fail(GC_EINVAL, "gc_prep_grid_input: bad sizes (kp must be a multiple of 32 >= c_in + n_struct)");
  if (!x || !xin || (n_struct && !node_struct)) return fail(GC_EINVAL, "gc_prep_grid_input: null pointer");
  const int rows_per_block = 4;
  hipLaunchKernelGGL(prep_grid_input_kernel, dim3((n_rows + rows_per_block - 1) / rows_per_block),
                     dim3(64 * rows_per_block), 0, static_cast<hipStream_t>(stream), n_rows, batch, b,
                     c_in, 0, x, n_struct, node_struct, kp, xin);
  return check_launch("prep_grid_input_kernel");
}

int gc_prep_grid_tail(int n_rows, int batch, int b, int c_in, int c0, const float* x, int n_struct,
                      const float* node_struct, int kt, float* xt, void* stream) {
  if (n_rows <= 0 || batch <= 0 || b < 0 || b >= batch || c_in <= 0 || n_struct < 0 || c0 < 0 || c0 > c_in ||
      (c0 & 31) || kt < c_in - c0 + n_struct || (kt & 31))
    return fail(GC_EINVAL, "gc_prep_grid_tail: bad sizes (c0, kt multiples of 32, kt >= c_in - c0 + n_struct)");
  if (!x || !xt || (n_struct && !node_struct)) return fail(GC_EINVAL, "gc_prep_grid_tail: null pointer");
  if ((reinterpret_cast<size_t>(xt) & 15) == 0) {          // (kt is a multiple of 32: whole float4s)
    const long long threads = (long long)n_rows * (kt >> 2);
    hipLaunchKernelGGL(prep_grid_rows4_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), n_rows, batch, b, c_in, c0, x, n_struct, node_struct, kt, xt);
    return check_launch("prep_grid_rows4_kernel");
  }
  const int rows_per_block = 4;
  hipLaunchKernelGGL(prep_grid_input_kernel, dim3((n_rows + rows_per_block - 1) / rows_per_block),
                     dim3(64 * rows_per_block), 0, static_cast<hipStream_t>(stream), n_rows, batch, b,
                     c_in, c0, x, n_struct, node_struct, kt, xt);
  return check_launch("prep_grid_input_kernel");
}

int gc_advance_state(const gc_advance_desc* dp, void* stream) {
  if (!dp) return fail(GC_EINVAL, "gc_advance_state: null descriptor");
  const gc_advance_desc& d = *dp;
  if (d.n_rows <= 0 || d.c_in <= 0 || d.c_out <= 0 || d.n_forc < 0)
    return fail(GC_EINVAL, "gc_advance_state: bad sizes");
  if (!d.x || !d.y || !d.x_next || !d.src_x || !d.ax || !d.src_y || !d.ay || !d.src_f)
    return fail(GC_EINVAL, "gc_advance_state: null pointer");
  if (d.x == d.x_next) return fail(GC_EINVAL, "gc_advance_state: x_next must not alias x");
  if (d.n_forc > 0 && (!d.f_cur || !d.f_next)) return fail(GC_EINVAL, "gc_advance_state: forcings missing");
  if (d.pred && (!d.p_src_x || !d.p_ax || !d.p_ay || !d.p_b))
    return fail(GC_EINVAL, "gc_advance_state: prediction tables missing");
  const size_t lds = ((((size_t)kAdvRows * d.c_in + 3) & ~(size_t)3) * 2 + (((size_t)kAdvRows * d.c_out + 3) & ~(size_t)3)) * sizeof(float);
  if (d.c_in > 256 * kAdvMaxJ || d.c_out > 256 * kAdvMaxJ) return fail(GC_EINVAL, "gc_advance_state: more than 1024 channels");
  static size_t lds_set = 0;
  if (lds > lds_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&advance_state_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      std::snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
      return GC_ELAUNCH;
    }
    lds_set = lds;
  }
  // 16-byte copies where base and row-block stride allow (bit 0 x, 1 y, 2 x_next, 3 pred)
  auto vec_ok = [](const void* p, int c) { return (reinterpret_cast<size_t>(p) & 15) == 0 && ((kAdvRows * c) & 3) == 0; };
  const int vec = (vec_ok(d.x, d.c_in) ? 1 : 0) | (vec_ok(d.y, d.c_out) ? 2 : 0) | (vec_ok(d.x_next, d.c_in) ? 4 : 0) |
                  (d.pred && vec_ok(d.pred, d.c_out) ? 8 : 0);
  hipLaunchKernelGGL(advance_state_kernel, dim3((d.n_rows + kAdvRows - 1) / kAdvRows), dim3(256), lds,
                     static_cast<hipStream_t>(stream), d, vec);
  return check_launch("advance_state_kernel");
}

static int run_op(const gc_op& op, void* stream) {
  switch (op.kind) {
    case GC_OP_ROWMLP:
      return gc_rowmlp(&op.mlp, stream);
    case GC_OP_FIXUP:
      if (op.mlp.prec == GC_PREC_BF16) return gc_seg_fixup_bf16(op.n, op.i0, op.i1, op.i2, op.src, op.dst, stream);
      return gc_seg_fixup(op.n, op.i0, op.i1, op.i2, op.src, op.dst, stream);
    case GC_OP_ZERO:
      if (op.mlp.prec == GC_PREC_BF16) return gc_zero_rows_bf16(op.n, op.i0, op.dst, stream);
      return gc_zero_rows(op.n, op.i0, op.dst, stream);
    case GC_OP_ADD:
      return gc_add_rows(op.n, op.i0, op.src, op.dst, stream);
    case GC_OP_PREP:
      if (op.c0 > 0)
        return gc_prep_grid_tail(op.n, op.batch, op.b, op.c_in, op.c0, op.x, op.n_struct, op.node_struct, op.kp,
                                 op.dst, stream);
      return gc_prep_grid_input(op.n, op.batch, op.b, op.c_in, op.x, op.n_struct, op.node_struct, op.kp,
                                op.dst, stream);
    default:
      return fail(GC_EINVAL, "gc_run_program: unknown op kind");
  }
}

int gc_run_program(const gc_op* ops, int n_ops, void* stream) {
  if (!ops || n_ops < 0) return fail(GC_EINVAL, "gc_run_program: bad arguments");
  for (int k = 0; k < n_ops; ++k) {
    const int rc = run_op(ops[k], stream);
    if (rc) return rc;
  }
  return 0;
}

int gc_time_program(const gc_op* ops, int n_ops, int iters, float* h_ms, void* stream) {
  if (!ops || n_ops <= 0 || iters <= 0 || !h_ms) return fail(GC_EINVAL, "gc_time_program: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t* ev = new hipEvent_t[n_ops + 1];
  for (int k = 0; k <= n_ops; ++k) hipEventCreate(&ev[k]);
  for (int k = 0; k < n_ops; ++k) h_ms[k] = 0.f;
  int rc = 0;
  for (int it = 0; it < iters && !rc; ++it) {
    hipEventRecord(ev[0], s);
    for (int k = 0; k < n_ops && !rc; ++k) {
      rc = run_op(ops[k], stream);
      hipEventRecord(ev[k + 1], s);
    }
    if (hipStreamSynchronize(s) != hipSuccess) rc = fail(GC_ELAUNCH, "gc_time_program: stream sync failed");
    for (int k = 0; k < n_ops && !rc; ++k) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
      h_ms[k] += ms / iters;
    }
  }
  for (int k = 0; k <= n_ops; ++k) hipEventDestroy(ev[k]);
  delete[] ev;
  return rc;
}

size_t gc_abi_sizeof(int what) {
  return what == 0 ? sizeof(gc_rowmlp_desc) : what == 1 ? sizeof(gc_op)
         : what == 2 ? sizeof(gc_advance_desc) : what == 3 ? sizeof(gc_model_desc) : what == 4 ? sizeof(gc_tuning) : 0;
}

const char* gc_last_error(void) { return g_err; }

int gc_get_tuning(gc_tuning* out) {
  if (!out) return fail(GC_EINVAL, "gc_get_tuning: null pointer");
  *out = tuning();
  return 0;
}
int gc_set_tuning(const gc_tuning* t) {
  if (!t) return fail(GC_EINVAL, "gc_set_tuning: null pointer");
  if (!tuning_valid(*t)) return fail(GC_EINVAL, "gc_set_tuning: a value is outside its range (include/gcast.h: gc_tuning)");
  tuning_mut() = *t;
  return 0;
}
const char* gc_tuning_string(const gc_tuning* tp) {
  static thread_local char buf[512];
  const gc_tuning& t = tp ? *tp : tuning();
  std::snprintf(buf, sizeof(buf),
                "grid_cap=%d;tile_map=%s;prio=%d,%d,%d%s;helpers=%d;helpers_small=%d;helpers_edge=%d;helper_store=%d;"
                "helpers_min_rows=%d;wide=%d;wide_edges=%d;bf16_rows=%d;tile_queue=%d;fuse=%d;onepass=%d;split_tail=%d;bf16_stream=%d;wide_late=%d;split_edges=%d",
                t.grid_cap, t.tile_map_xcd ? "xcd" : "rr", t.prio_gemm, t.prio_other, t.prio_stage, t.prio_set ? "(set)" : "",
                t.helpers, t.helpers_small, t.helpers_edge, t.helper_store, t.helpers_min_rows, t.wide, t.wide_edges, t.bf16_rows,
                t.tile_queue, t.fuse, t.onepass, t.split_tail, t.bf16_stream, t.wide_late, t.split_edges);
  return buf;
}

#include "gcast_plan.inc"

#define GC_STR2(x) #x
#define GC_STR(x) GC_STR2(x)
const char* gc_build_info(void) {
  return "gfx950;tile=64x512;mfma=f32_16x16x4|3xf16_16x16x32|bf16_16x16x32;tiers=bf16(Bfloat16Cast);"
         "layouts=chunked(f32)|half(f16x3: 2wg/cu,persistent,chain)|half+helpers(8 waves,1wg/cu)|half+wide(8 multiplying waves,128 rows/ring,10/16 parked n-blocks in LDS,seg+onepass);ring=4x16k"
         ";plan=latent<=512(padded parameters),hidden_layers>=1(one launch per further layer)"
         ";helpers_default=" GC_STR(GC_HELPERS_DEFAULT) ";wide_edges_default=" GC_STR(GC_WIDE_EDGES_DEFAULT)
#ifdef GC_SRC_HASH
         ";src=" GC_SRC_HASH
#endif
#ifdef GC_PROFILING_BUILD
         ";PROFILING_BUILD(results may be wrong)"
#endif
      ;
}

}  // extern "C"
