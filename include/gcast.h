/* gcast.h -- C-ABI of libgcast_hip.so: the MI355X (gfx950) device side of
 * GraphCast's encode-process-decode step.
 *
 * The reference (google-deepmind/graphcast, packaged as `weathernext`) has no
 * FFI: the path is Python calling jax/haiku/jraph.  Each entry point below
 * replaces the XLA-lowered form of the reference interface cited next to it
 * (paths relative to /root/reference/weathernext).  Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     the parameter name starts with `h_`; the caller owns all memory;
 *   - all work is enqueued on the caller's stream (`void* stream` is a
 *     hipStream_t; pass torch.cuda.current_stream().cuda_stream); nothing here
 *     allocates, frees or synchronises (except gc_time_program, a measurement
 *     helper that synchronises by design);
 *   - return 0 on success, a negative GC_E* code otherwise; the message is in
 *     gc_last_error() (thread-local);
 *   - fp32 tensors everywhere; the GEMMs run in one of two arithmetic modes
 *     (`gc_rowmlp_desc.prec`, below) that both deliver fp32-grade results;
 *     a launch's tile is 512 columns wide (GraphCast's `latent_size`,
 *     weathernext1_graph/graphcast.py:123) and fuses exactly one hidden layer
 *     (`hidden_layers`, :124): what every published GraphCast has.  The PLAN
 *     API runs narrower latents and deeper MLPs on these launches (padded
 *     parameters, one further launch per further hidden layer: see
 *     gc_plan_create); a latent size above 512 or a single-Linear MLP is
 *     rejected loudly.
 *
 * Packed weight layout ("k4-interleaved"): a haiku `w` of shape [K, N]
 * (x @ w, utils/legacy/deep_typed_graph_net.py:206-208) is stored as
 * Wp[K/4][NP][4] with Wp[q][n][j] = w[4q + j][n]; K is zero-padded to a
 * multiple of 32 and N to NP = 512 (or 256 for the decoder's last layer).
 * A 32-row K chunk is then one contiguous 8*NP*16-byte block that is DMA'd
 * linearly into LDS and read conflict-free as ds_read_b128 MFMA A-fragments.
 *
 * Split-f16 weight layout (prec == GC_PREC_F16X3): every weight is stored as the
 * pair (hi, lo) of IEEE halves with hi = fp16(w), lo = fp16(w - hi) (22 mantissa
 * bits).  A 32-row K chunk is [NP/16 n-blocks][2: hi, lo][64 lanes][8 halves]:
 * lane l = 16*g + n of n-block nb holds W[kmap(g, j)][16*nb + n], j = 0..7 --
 * exactly one v_mfma_f32_16x16x32_f16 A-fragment per 16 B.  kmap is
 *   natural (layer 1, rows come from memory):  32*c + 8*g + j
 *   chained (layer 2, rows are layer 1's accumulator registers):
 *                                              32*c + 4*g + j        (j < 4)
 *                                              32*c + 16 + 4*g + j-4 (j >= 4)
 * Chunk size in bytes is the same as in the fp32 layout (NP * 128).  The whole matrix may be
 * pre-multiplied by a power of two (gc_rowmlp_desc.w1_scale / w2_scale) so that lo = fp16(w - hi)
 * of a typical weight is a normal fp16 number (fp16 subnormals have an ABSOLUTE spacing of 2^-24).
 */
#ifndef GCAST_H_
#define GCAST_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GC_LATENT 512
#define GC_TILE_ROWS 64          /* rows per workgroup tile (4 waves x 16) */
#define GC_K_CHUNK 32            /* K rows per LDS weight chunk */

#define GC_SCRATCH_SLOTS 512     /* persistent workgroups of a GC_LAYOUT_HALF launch: 2 per CU x 256 CUs */
#define GC_SCRATCH_FLOATS ((size_t)GC_SCRATCH_SLOTS * GC_TILE_ROWS * 256)   /* floats in gc_rowmlp_desc.scratch */

#define GC_EINVAL (-1)
#define GC_ELAUNCH (-2)
#define GC_ERANGE (-3)       /* gc_plan_check_range: an input row held |x| > GC_F16X3_MAX (see gc_rowmlp_desc.range_flag) */
#define GC_F16X3_MAX 65504.0f  /* GC_PREC_F16X3 splits a row value into two halves, exact (20+ bits) for |x| <= this */

/* Arithmetic of the two GEMMs of a launch.  Inputs, outputs, accumulation, bias,
 * LayerNorm, residual and segment-sum are fp32 in every mode.
 *   GC_PREC_F32   v_mfma_f32_16x16x4_f32: exact fp32 products (157 TF peak).
 *   GC_PREC_F16X3 each fp32 operand x is split in registers into two halves
 *                 x = hi + lo (22 mantissa bits) and every product is formed as
 *                 x_hi.w_hi + x_lo.w_hi + x_hi.w_lo with three
 *                 v_mfma_f32_16x16x32_f16 accumulating in fp32 (dropped term
 *                 ~2^-22): fp32-grade results at 1/3 of the 2.5 PF f16 MFMA
 *                 peak.  |x| must stay below 1.3e5 (saturating split).
 *   GC_PREC_BF16_IMAGE  (= 2) NOT a launch precision any more: rounds 1-4 ran a "bf16gemm" tier under this value
 *                 (GEMM operands rounded to bfloat16, fp32 everywhere else -- numerics the reference does not have;
 *                 retired in round 5).  The value still names its weight image -- bfloat16, hi-only,
 *                 [NP/16 n-blocks][64 lanes][8 bf16] per 32-row K chunk (NP * 64 bytes) -- which GC_PREC_BF16
 *                 uses, for gc_host_pack_weight / gc_host_packed_weight_bytes.
 *   GC_PREC_BF16  the TIER that follows the reference's casting.Bfloat16Cast run (utils/casting.py:31-65,
 *                 155-205; fp32 aggregation only where graphcast.py:215 asks for it is over-fulfilled:
 *                 every aggregation accumulates in fp32): ALL row tensors -- a0 / a1 (unless
 *                 GC_ROWS_F32), d, g0, g1, res, out, agg, chained GC_CHAIN_ROWS outputs -- are
 *                 bfloat16 [rows, 512] in "pi order" (position 32 m + 8 g + 4 b + r holds logical
 *                 column 16 (2 m + b) + 4 g + r; strides in ELEMENTS, multiples of 8); vectors (b1,
 *                 b2, LayerNorm, chain biases) stay fp32 [512] in natural order and should hold
 *                 bfloat16-representable values; `partial` rows are fp32 in pi order; a
 *                 GC_CHAIN_NARROW output is fp32, natural order, bfloat16-rounded values.  Weights: the
 *                 GC_PREC_BF16_IMAGE image; every matrix whose K operand is a bfloat16 row tensor (or
 *                 a launch's own rows) in the CHAINED K order, one fed by GC_ROWS_F32 rows in the
 *                 natural one; no weight scales.  GC_LAYOUT_HALF + GC_MODE_MLP_LN launches only; no
 *                 `scratch`.  Values are rounded to bfloat16 (nearest even) where the reference's
 *                 program materialises an array (csrc/rowmlp_bf16.inc); ~1e-2 rel-RMSE from the fp32
 *                 step -- the reference's own bf16 run is as far away. */
enum gc_precision { GC_PREC_F32 = 0, GC_PREC_F16X3 = 1, GC_PREC_BF16_IMAGE = 2 /* not a launch precision, see above */, GC_PREC_BF16 = 3 };

/* gc_rowmlp_desc.flags */
#define GC_ROWS_F32 1            /* GC_PREC_BF16: a0 / a1 are external fp32 rows in natural column order */
#define GC_W2_NATURAL 2          /* GC_PREC_F16X3 + GC_LAYOUT_HALF + GC_MODE_MLP_LN launch WITHOUT a layer-1 GEMM
                                  * (k0 + k1 == 0; d and g0 given, g1 optional, no chain): w2p is packed in the
                                  * NATURAL K order and the launch runs in ONE pass -- every K chunk's hidden
                                  * columns are formed on the fly from the same columns of the addend rows, the
                                  * whole output row accumulates in registers, nothing is parked (no `scratch`).
                                  * Same arithmetic as the two-pass launch up to the summation order inside an
                                  * MFMA (csrc/rowmlp_half.inc, ONEPASS). */

#define GC_WG_ROWS_64 4          /* GC_PREC_BF16: pin the rows per workgroup -- 64 (four waves, two workgroups per CU) */
#define GC_WG_ROWS_128 8         /* or 128 (eight waves sharing one weight stream, one workgroup per CU).  Neither
                                  * flag: chosen per launch (128 for launches without gather / segment-sum from
                                  * 65,536 rows on).  Results are bit-identical either way. */
#define GC_WG_HELPERS 32         /* GC_LAYOUT_HALF: run the launch as ONE eight-wave workgroup per CU -- four multiplying
                                  * waves, four waves that stage the weight ring for them (csrc/rowmlp_half.inc:
                                  * rowmlp16d_kernel) -- instead of two four-wave workgroups.  Bit-identical results. */
#define GC_WG_NO_HELPERS 64      /* ... or pin the four-wave form (neither flag: the build's default, GC_HELPERS_DEFAULT) */
#define GC_WG_WIDE 256           /* GC_LAYOUT_HALF, GC_MODE_MLP_LN launches: ONE workgroup of eight MULTIPLYING waves per
                                  * CU -- 128 rows against one weight ring, half the L2 -> LDS stream per row
                                  * (csrc/rowmlp_half.inc: rowmlp16w_kernel).  Same bits.  Round 6: also launches with a
                                  * segment-sum and the one-pass (GC_W2_NATURAL) ones.  Ignored in other modes. */
#define GC_LATE_ADDENDS 512      /* GC_PREC_F16X3, two-pass GC_MODE_MLP_LN launches with a layer-1 GEMM and gathered addends (the
                                  * processor's edge update): the gathered rows are added when the hidden layer is formed --
                                  * loads pipelined three K steps ahead under the swish / split bursts -- instead of in a burst in
                                  * front of layer 1: ((b1 + products) + g0) + g1 instead of (b1 + g0 + g1) + products, another
                                  * fp32 ASSOCIATION (1e-7 apart).  Honoured by the four-wave and the wide form (the same bits
                                  * in both); a launch with the flag does not run in the helper form.  gc_tuning.wide_late sets
                                  * it on the launches the wide_edges rule puts into the wide form. */
#define GC_WIDE_MIN_ROWS 262144   /* the plan asks for GC_WG_WIDE from this many rows on (>= 8 rounds of 256 128-row tiles:
                                  * below, the doubled tail costs more than the shared ring saves) */
#define GC_TILE_QUEUE_ANY 128    /* gc_rowmlp_desc.tile_queue: hand the tiles out dynamically whenever the launch has more
                                  * tiles than workgroups (default: only from GC_TILE_QUEUE_MIN_ROUNDS tiles per
                                  * workgroup on -- below that the static walk places the few second-round tiles on
                                  * distinct CUs, which measured better: processor node updates, 641 tiles on 512
                                  * workgroups, 5.5 -> 6.1 ms per step with the queue, profiles/r04_s22_*) */
#define GC_TILE_QUEUE_MIN_ROUNDS 4
#ifndef GC_HELPERS_DEFAULT
#define GC_HELPERS_DEFAULT 0
#endif
#define GC_HELPERS_MIN_ROWS_DEFAULT 65536   /* the plan API asks for ONE eight-wave workgroup per CU on launches without
                                             * gather / segment-sum from this many rows on (0: never): GC_WG_WIDE where the
                                             * launch is a two-pass GC_MODE_MLP_LN launch of >= GC_WIDE_MIN_ROWS rows,
                                             * GC_WG_HELPERS elsewhere (and everywhere with GCAST_WIDE=0) */
/* GC_LAYOUT_HALF: wave issue priority by phase (s_setprio; round 5).  Two waves share a SIMD -- the two workgroups of
 * a CU, or a multiplying and a staging wave of the eight-wave form -- and the scheduler arbitrates their VALU / MFMA
 * issue by priority, then age (MI355X_MICROARCH.md "Two waves per SIMD").  Three 2-bit fields: the priority of a wave
 * inside its GEMM phases, outside them (gather, LayerNorm, segment-sum, residual / store), and of a staging wave.
 * 0 everywhere = the hardware default.  A speed choice only.  GCAST_PRIO="g,e,s" sets it for a whole process. */
#define GC_PRIO_SHIFT 16
#define GC_PRIO(gemm, other, stage) ((((gemm) & 3) | (((other) & 3) << 2) | (((stage) & 3) << 4)) << GC_PRIO_SHIFT)
#define GC_PRIO_GEMM_DEFAULT 1      /* (the GC_PREC_F16X3 kernels; GC_PREC_BF16: 0 -- measured, csrc/gcast.hip half_prio_flags) */
#define GC_PRIO_OTHER_DEFAULT 0
#define GC_PRIO_STAGE_DEFAULT 0
#define GC_HELPERS_EDGE_MAX_TILES 2048   /* the processor's two-pass edge update runs in the eight-wave form with WORKING staging
                                          * waves (residual + store, the next tile's addend gather) by default up to this
                                          * many tiles: beyond it the part is at its power limit and the step gains nothing
                                          * (csrc/gcast.hip: launch_rowmlp_half) */
#define GC_TILE_XCD 16           /* GC_LAYOUT_HALF: tile -> workgroup map in which each XCD walks a contiguous eighth
                                  * of the launch's tiles (csrc/rowmlp_half.inc).  A speed choice only. */

/* How w1p / w2p are packed, i.e. which tile formulation runs.
 *   GC_LAYOUT_CHUNKED  the layouts described above: 32-row K chunks staged through LDS, every wave
 *                      owns 16 rows of the tile and reads the whole chunk (all modes, all precisions).
 *   GC_LAYOUT_HALF     (GC_PREC_F16X3, all modes) the CHUNKED images unchanged, streamed as 16 KiB
 *                      quarter chunks (8 n-blocks of a chunk image) through a four-deep ring; layer 2
 *                      runs as two passes over the output columns with pass 0's accumulators parked
 *                      in `scratch`; <= 256 VGPRs and 75 KiB of LDS per workgroup, so that TWO
 *                      workgroups share a CU and one's non-GEMM phases run under the other's MFMAs
 *                      (csrc/rowmlp_half.inc).  Launches are PERSISTENT: at most GC_SCRATCH_SLOTS
 *                      workgroups, workgroup b walks the tiles b, b + grid, ...  MLP_LN launches
 *                      need `scratch`. */
enum gc_weight_layout { GC_LAYOUT_CHUNKED = 0, /* 1: retired (column-owner formulation, rounds 1-2) */ GC_LAYOUT_HALF = 2 };

/* What a fused row-MLP launch produces. */
enum gc_rowmlp_mode {
  /* out = A.W1 + addends            (no activation, no layer 2)            */
  GC_MODE_LINEAR = 0,
  /* out = [res +] LN(swish(A.W1 + addends).W2 + b2), N2 = 512;
   * optionally also segment-summed over receiver-sorted rows               */
  GC_MODE_MLP_LN = 1,
  /* out = swish(A.W1 + addends).W2 + b2, N2 <= 256, no LayerNorm (decoder) */
  GC_MODE_MLP_OUT = 2
};

/* GC_LAYOUT_HALF + GC_MODE_MLP_LN: further Linear layers applied to the rows a launch has just
 * produced (out = [res +] LN(...)), while they are still in registers -- the reference applies them
 * as separate layers of the NEXT module: the next edge update's sender / receiver products
 * (typed_graph_net.py:431-453 after the pre-gather split), the decoder's output MLP
 * (deep_typed_graph_net.py:313-322).  Stages run in order; their weights are packed like a layer-2
 * matrix (chained K order), NP = 512 (ROWS / SWISH) or 256 (NARROW).
 *   GC_CHAIN_ROWS    out[r, 0:512] = rows[r] . W + b          (row stride ldo); rows unchanged
 *   GC_CHAIN_SWISH   rows[r] <- swish(rows[r] . W + b)        nothing stored
 *   GC_CHAIN_NARROW  out[r, 0:n]  = rows[r] . W + b, n <= 240 (row stride ldo)
 * (GC_CHAIN_LN is the launch's own layer-2 epilogue; not a valid chain kind.) */
enum gc_chain_kind { GC_CHAIN_LN = 0, GC_CHAIN_ROWS = 1, GC_CHAIN_SWISH = 2, GC_CHAIN_NARROW = 3 };
typedef struct gc_chain_stage {
  const void* wp;          /* packed weights (16 K steps of NP * 128 bytes) */
  const float* b;          /* [NP] bias or NULL */
  float* out; int ldo;     /* ROWS / NARROW */
  int n;                   /* NARROW: real output width */
  int kind;                /* enum gc_chain_kind */
  float w_scale;           /* power of two the packed weights carry */
} gc_chain_stage;
#define GC_MAX_CHAIN 2

/* One fused "rows -> MLP (-> LayerNorm) (-> residual) (-> segment-sum)" launch.
 *
 * Replaces, per reference call site:
 *   - hk.nets.MLP + hk.LayerNorm built at deep_typed_graph_net.py:205-247
 *     (embedders :250-271, processor :294-311, decoder :313-322);
 *   - the sender/receiver row gathers of typed_graph_net.py:431-447 (`g0/idx0`,
 *     `g1/idx1`, applied AFTER the W1 product: x[idx].W == (x.W)[idx]);
 *   - jraph.concatenated_args (deep_typed_graph_net.py:209,247): the concat
 *     [a0 | a1] is realised as two K-slices of W1;
 *   - the residual of deep_typed_graph_net.py:380-392 (`res`);
 *   - jraph.segment_sum over receivers, typed_graph_net.py:532-538 (`seg`).
 *
 * layer-1 pre-activation  z[r,:] = a0[r,:k0].W1[:k0] + a1[r,:k1].W1[k0:k0+k1]
 *                                  + d[r,:] + g0[idx0[r],:] + g1[idx1[r],:] + b1
 */
typedef struct gc_rowmlp_desc {
  int mode;                /* enum gc_rowmlp_mode */
  int prec;                /* enum gc_precision: selects the layout w1p / w2p are packed in */
  int n_rows;              /* rows to process */
  int layout;              /* enum gc_weight_layout: the packing of w1p / w2p */
  /* Power-of-two factors the packed weights were multiplied by (GC_PREC_F16X3: chosen at pack
   * time so that the lo halves of typical weights are NORMAL fp16 numbers; the kernel scales the
   * layer's addends by the same factor and the accumulators by its inverse -- all exact).
   * Must be 1 (or 0 = unset) for GC_PREC_F32. */
  float w1_scale, w2_scale;
  /* layer-1 GEMM sources (row-major, 16-byte aligned rows); k0,k1 multiples of 32, may be 0 */
  const float* a0; int lda0; int k0;
  const float* a1; int lda1; int k1;
  const void* w1p;         /* packed (k0+k1)/32 chunks of 64 KiB; NULL iff k0+k1 == 0 */
  /* addends to the pre-activation (each may be NULL) */
  const float* d;  int ldd;          /* direct rows d[r] */
  const float* g0; const int* idx0;  /* gathered rows g0[idx0[r]], row stride 512 */
  const float* g1; const int* idx1;
  const float* b1;         /* [512] */
  /* layer 2 (modes MLP_LN / MLP_OUT) */
  const void* w2p;         /* packed 16 chunks of NP*128 B, NP = 512 (MLP_LN) or 256 (MLP_OUT) */
  const float* b2;         /* [NP] (zero padded) */
  int n2;                  /* real output width: 512 (MLP_LN), <= 240 (MLP_OUT) */
  /* LayerNorm (mode MLP_LN; NULL scale => skip) */
  const float* ln_scale; const float* ln_offset;
  /* residual and store */
  const float* res; int ldres;       /* added after LayerNorm; may alias out */
  float* out; int ldo;               /* NULL => nothing stored (segment-sum only) */
  /* segment-sum of the (pre-residual) rows over receiver-sorted, tile-packed rows
   * (mode MLP_LN only; n_rows must be a multiple of GC_TILE_ROWS) */
  const int* seg;          /* [n_rows] receiver id per row, -1 = padding row; NULL => off */
  const int* tile_flags;   /* [n_rows/64] bit0: first run continues the previous tile,
                                          bit1: last run continues into the next tile */
  float* agg;              /* [n_receivers][512] rows owned entirely by one tile */
  float* partial;          /* [2*n_rows/64][512] straddling partial sums */
  /* GC_LAYOUT_HALF + GC_MODE_MLP_LN: GC_SCRATCH_FLOATS floats (32 MiB, whatever n_rows is) the launch
   * may overwrite: one 64 KiB slot per persistent workgroup, in which a tile parks half of its
   * layer-2 accumulators between the two column passes.  Rewritten by every tile: cache resident.
   * Launches that run one after another on a stream may share it. */
  float* scratch;
  /* GC_LAYOUT_HALF + GC_MODE_MLP_LN, no segment-sum: chained stages (see gc_chain_stage); with a
   * chain `out` may be NULL (the rows are only consumed by the chain) */
  int n_chain;
  gc_chain_stage chain[GC_MAX_CHAIN];
  int flags;               /* GC_ROWS_F32 | ... (0 for everything but GC_PREC_BF16) */
  /* GC_PREC_F16X3 + GC_LAYOUT_HALF, optional: a device word the launch sets to 1 when a layer-1 row value (a0 / a1)
   * exceeds GC_F16X3_MAX in magnitude.  The split halves saturate there (exact to 6.5e4, 5e-4 up to 1.3e5, garbage
   * beyond) where the reference's fp32 does not care -- launches fed by EXTERNAL rows (the grid embedder: un-normalised
   * geopotential is ~5e5) pass it, and so do the node updates whose layer-1 operand is an AGGREGATE (a per-receiver
   * sum over up to 3,753 edge messages is not a LayerNorm output; the reference up-casts that sum to fp32 because it is
   * large, weathernext1_graph/graphcast.py:215); the host reads it at its next synchronisation point and raises.
   * Never cleared by a launch.  NULL: no check (rows a LayerNorm has produced). */
  int* range_flag;
  /* Persistent kernels (GC_LAYOUT_HALF, GC_PREC_BF16), optional: TWO device words (8-byte aligned), both ZERO before
   * the first launch that is given them.  A launch of at least GC_TILE_QUEUE_MIN_ROUNDS tiles per workgroup (any
   * launch of more tiles than workgroups with GC_TILE_QUEUE_ANY) then hands its tiles out dynamically -- a workgroup takes its first tile by its index and every further one from word 0 -- instead of
   * walking b, b + grid, ...: the launch ends when the work does, not when the slowest workgroup has done a fixed
   * share (the workgroups of one launch differ by +-13 % in speed on the MI355X, csrc/rowmlp_half.inc).  Results do
   * not depend on it.  Every launch leaves both words zero (word 1 counts the workgroups that have left; the last
   * one clears the pair), so launches that run ONE AFTER ANOTHER on a stream may share them -- the contract of
   * `scratch`; launches that may overlap need their own pair.  NULL: static walk. */
  int* tile_queue;
} gc_rowmlp_desc;

int gc_rowmlp(const gc_rowmlp_desc* desc, void* stream);

/* Deterministic combine of straddling segment partials (no float atomics):
 * agg[recv[i]] = partial[2*t0[i]+1] + sum_{t=t0[i]+1..t1[i]} partial[2*t].
 * Completes jraph.segment_sum (typed_graph_net.py:532-538). */
int gc_seg_fixup(int n_entries, const int* recv, const int* t0, const int* t1,
                 const float* partial, float* agg, void* stream);

/* agg[rows[i], :] = 0 for receivers without incoming edges (segment_sum's
 * "zeros for empty segments"). */
int gc_zero_rows(int n, const int* rows, float* agg, void* stream);

/* dst[rows[i], :] += src[rows[i], :] (fp32 rows of 512).  The spatially partitioned step runs an edge update
 * as TWO launches -- edges whose sender row is local (under which the halo exchange runs) and edges whose
 * sender row arrives with the exchange; this joins the second launch's aggregate rows into the first's
 * (GC_OP_ADD: n, i0 = rows, src, dst).  Completes jraph.segment_sum (typed_graph_net.py:532-538) there. */
int gc_add_rows(int n, const int* rows, const float* src, float* dst, void* stream);

/* The same two for a GC_PREC_BF16 launch: `partial` fp32 rows in pi order, `agg` bfloat16 rows
 * (gc_run_program picks them for GC_OP_FIXUP / GC_OP_ZERO ops whose mlp.prec is GC_PREC_BF16). */
int gc_seg_fixup_bf16(int n_entries, const int* recv, const int* t0, const int* t1, const float* partial,
                      void* agg, void* stream);
int gc_zero_rows_bf16(int n, const int* rows, void* agg, void* stream);

/* xin[r, :] = [ x[r, b, 0:c_in] | node_struct[r, 0:n_struct] | 0-pad ] (row stride kp).
 * Replaces the concat + batch broadcast of graphcast.py:561-568 for batch
 * element b of x [n_rows, batch, c_in]. */
int gc_prep_grid_input(int n_rows, int batch, int b, int c_in, const float* x,
                       int n_struct, const float* node_struct, int kp, float* xin,
                       void* stream);

/* The same for input columns c0 .. c0 + kt - 1 only (c0 a multiple of 32 <= c_in):
 * xt[r, :] = [ x[r, b, c0:c_in] | node_struct[r, :] | 0-pad ] (row stride kt).  With it a
 * GC_LAYOUT_HALF launch reads x[:, b, 0:c0] IN PLACE as its first K chunks (a0 = x + b * c_in,
 * lda0 = batch * c_in, k0 = c0; any float alignment) and this 32-column tail as the last one
 * (a1 = xt, k1 = kt) -- the [n_rows, kp] copy of gc_prep_grid_input is not made. */
int gc_prep_grid_tail(int n_rows, int batch, int b, int c_in, int c0, const float* x,
                      int n_struct, const float* node_struct, int kt, float* xt, void* stream);

/* One autoregressive state advance, entirely on the device: builds the NORMALISED stacked
 * inputs of step s+1 from those of step s, the step's normalised output and the forcings,
 * and (optionally) the DE-normalised prediction of step s.  Fuses, per channel,
 *   - the rolling window of rollout._get_next_inputs (utils/rollout.py:581-604) /
 *     autoregressive.Predictor._update_inputs (utils/autoregressive.py:114-125),
 *   - InputsAndResiduals' un-normalise + residual add and the re-normalisation of the next
 *     call (utils/normalization.py:113-132,148-160): in normalised space
 *         x'_last = x_last + y * (diffs_stddev / stddev),
 *   - the Dataset <-> stacked-channel round trip of graphcast.py:680-723 (never materialised).
 * Every input channel c of the next state is an affine pick:
 *   x_next[r,c] = ax[c]*x[r,src_x[c]] + ay[c]*y[r,src_y[c]] + f[r,src_f[c]]   (src < 0: term absent)
 * with f = [f_cur | f_next] (normalised forcings at the time just predicted / at the next
 * target time, n_forc channels each); and every output channel k
 *   pred[r,k] = p_ay[k]*y[r,k] + p_ax[k]*x[r,p_src_x[k]] + p_b[k].
 * x_next must not alias x.  Rows = grid nodes x batch.  HBM-bound. */
typedef struct gc_advance_desc {
  int n_rows, c_in, c_out, n_forc;
  const float* x;        /* [n_rows, c_in]  normalised stacked inputs of this step */
  const float* y;        /* [n_rows, c_out] normalised step output */
  const float* f_cur;    /* [n_rows, n_forc] or NULL */
  const float* f_next;   /* [n_rows, n_forc] or NULL */
  const int* src_x; const float* ax;   /* [c_in] */
  const int* src_y; const float* ay;   /* [c_in] */
  const int* src_f;                    /* [c_in] index into [f_cur | f_next] */
  float* x_next;         /* [n_rows, c_in] */
  const int* p_src_x; const float* p_ax; const float* p_ay; const float* p_b;   /* [c_out] */
  float* pred;           /* [n_rows, c_out] de-normalised prediction, or NULL */
} gc_advance_desc;

int gc_advance_state(const gc_advance_desc* desc, void* stream);

/* A recorded sequence of launches = one encode-process-decode step
 * (graphcast.py:306-323 between _inputs_to_grid_node_features and
 * _grid_node_outputs_to_prediction). */
enum gc_op_kind { GC_OP_ROWMLP = 0, GC_OP_FIXUP = 1, GC_OP_ZERO = 2, GC_OP_PREP = 3, GC_OP_ADD = 4 };

typedef struct gc_op {
  int kind;                /* enum gc_op_kind */
  int tag;                 /* caller-defined stage id, reported back by gc_time_program */
  gc_rowmlp_desc mlp;      /* GC_OP_ROWMLP */
  /* GC_OP_FIXUP / GC_OP_ZERO */
  int n; const int* i0; const int* i1; const int* i2;
  const float* src; float* dst;
  /* GC_OP_PREP */
  int batch, b, c_in, n_struct, kp;
  const float* x; const float* node_struct;
  int c0;                  /* > 0: gc_prep_grid_tail with kt = kp */
} gc_op;

int gc_run_program(const gc_op* h_ops, int n_ops, void* stream);

/* Measurement helper: runs the program `iters` times, bracketing EVERY op with
 * hipEvents on `stream`; h_ms[i] receives the mean milliseconds of op i.
 * Synchronises the stream.  Not used on the product path. */
int gc_time_program(const gc_op* h_ops, int n_ops, int iters, float* h_ms, void* stream);

/* ---- Plan API: the whole step behind two calls, for hosts without the Python packer ------------
 * A plan owns the static device data of one model on one device: the three edge sets in packed
 * (receiver-sorted, tile-packed) order, the packed weights of every MLP in the chosen arithmetic,
 * and the input-independent terms folded at creation.  It replaces what the
 * reference builds in GraphCast.__init__ / _maybe_init (graphcast.py:184-292,368-548) plus the
 * haiku parameter tree of a CheckPoint (graphcast.py:145-151).  All pointers in the descriptors
 * below are HOST pointers; arrays are in the reference's own order (edges in construction order,
 * weights as haiku stores them: w [in, out], row-major).
 *
 * gc_plan_create   packs on the host, uploads, runs the folding launches on `stream` and
 *                  synchronises it once (creation is not on the hot path).  Device memory of the
 *                  plan is allocated with hipMalloc and released by gc_plan_destroy.
 * gc_step_forward  x [n_grid, batch, c_in] -> y [n_grid, batch, c_out] (device, fp32, contiguous):
 *                  enqueues the launches of graphcast.py:311-319 on `stream`; never allocates,
 *                  never synchronises.  `workspace` (device, 256-byte aligned,
 *                  >= gc_plan_workspace_bytes) belongs to the caller; one workspace per stream in
 *                  flight -- the plan itself is immutable after creation.
 * Tensor names are "<haiku module>/<leaf>", e.g.
 *   "mesh_gnn/~_networks_builder/processor_edges_3_mesh_mlp/~/linear_0/w"
 *   "grid2mesh_gnn/~_networks_builder/encoder_nodes_grid_nodes_layer_norm/scale"  */
typedef struct gc_plan gc_plan;

typedef struct gc_edge_set {
  int n_edges;
  const int* h_senders;      /* [n_edges] */
  const int* h_receivers;    /* [n_edges] */
  const float* h_feat;       /* [n_edges, n_feat] structural edge features */
  int n_feat;                /* <= 32 */
} gc_edge_set;

typedef struct gc_model_desc {
  int n_grid, n_mesh;
  int c_in, c_out;           /* channels of x / y (structural node features not included) */
  int n_struct;              /* structural node features per node (3) */
  int num_steps;             /* processor message-passing steps (16) */
  int prec;                  /* enum gc_precision */
  const float* h_grid_node_feat;   /* [n_grid, n_struct] */
  const float* h_mesh_node_feat;   /* [n_mesh, n_struct] */
  gc_edge_set g2m, mesh, m2g;      /* grid->mesh, mesh->mesh, mesh->grid */
  int layout;                /* enum gc_weight_layout of the launches: GC_LAYOUT_CHUNKED, or
                                GC_LAYOUT_HALF (GC_PREC_F16X3: the workspace then includes the launches' scratch
                                rows; GC_PREC_BF16, the Bfloat16Cast tier, exists in this formulation only:
                                bfloat16 workspace rows, constants folded by the fp32-grade kernels at creation) */
  /* Spatially partitioned graphs (round 5; one rank's LOCAL graphs, graphcast_amd/partition.py): the node tables that
   * edges GATHER from carry a halo suffix of remote sender rows behind the owned rows -- sender indices >= n_grid /
   * n_mesh address it; the launches run over the owned prefix, the caller's halo exchange fills the suffix at the
   * program's exchange points (gc_plan_program: right in front of every edge update).  Rows of the grid-node table
   * of the encoder's senders, of the mesh-node tables of the processor's and of the decoder's senders; 0 = no halo
   * (= n_grid / n_mesh).  */
  int n_grid_senders, n_mesh_senders, n_mesh_senders_dec;
} gc_model_desc;

typedef struct gc_tensor_desc {
  const char* name;
  const float* h_data;       /* row-major [rows, cols]; vectors: rows == 1 */
  int rows, cols;
} gc_tensor_desc;

/* Sizes other than the published ones (round 6).  The kernels' tile is GC_LATENT = 512 columns and fuses ONE hidden
 * layer; gc_plan_create reads both sizes off the parameter tree (ModelConfig.latent_size / hidden_layers,
 * weathernext1_graph/graphcast.py:123-124; deep_typed_graph_net.py:205-209) --
 *   latent size L < 512: the parameters are re-shaped once (zero-padded latent axes; the output columns of every Linear
 *     that feeds a LayerNorm REPLICATED floor(512 / L) times -- and, where L does not divide 512, the remaining columns
 *     filled with the MEAN column and the Linear / the LayerNorm scale rescaled by sqrt(512 / (R L)) and its inverse --
 *     so that the statistics the kernels form over 512 columns are exactly those over the L real ones) and the same
 *     launches run: correct, at the 512-wide model's cost.  L > 512: GC_EINVAL;
 *   n > 1 hidden layers ("<stem>_mlp/~/linear_0" .. "linear_n"): one further launch of the same kernels per further
 *     hidden layer, pre-activation rows handed on through the workspace; the output MLP then runs as launches of
 *     its own instead of as chained stages.  n = 0 (a single Linear): GC_EINVAL. */
int gc_plan_create(const gc_model_desc* model, const gc_tensor_desc* tensors, int n_tensors,
                   void* stream, gc_plan** out);
/* The re-shaping for L < 512 on its own (host only, no device): tensor `index` of the tree as gc_plan_create would
 * see it -- *rows x *cols, copied to h_out when that is not NULL (capacity in floats).  Exported so that a binding can
 * check the rule (tests/test_native_abi.py: the oracle gives the same step on the re-shaped tree). */
int gc_host_pad_latent(const gc_tensor_desc* tensors, int n_tensors, int index, float* h_out, long long capacity,
                       int* rows, int* cols);
size_t gc_plan_workspace_bytes(const gc_plan* plan, int batch);
int gc_step_forward(const gc_plan* plan, const float* x, float* y, int batch, void* workspace,
                    size_t workspace_bytes, void* stream);
/* GC_PREC_F16X3 plans: gc_step_forward clears a range word in the workspace; the grid embedder sets it when an
 * input value exceeds GC_F16X3_MAX, the aggregate-fed node updates (encoder mesh nodes, processor nodes, decoder grid
 * nodes) when a per-receiver message sum does (gc_rowmlp_desc.range_flag).  gc_plan_check_range synchronises `stream`, reads
 * the word of the LAST gc_step_forward on this workspace and returns 0 or GC_ERANGE (message in gc_last_error):
 * call it wherever the host synchronises anyway, before trusting y.  Other precisions: always 0. */
int gc_plan_check_range(const gc_plan* plan, void* workspace, void* stream);

/* The plan's launch program, handed out instead of enqueued (round 5): the ops gc_step_forward(plan, x, y, batch,
 * workspace) would run, in order, each tagged with its stage (enum gc_stage_tag).  THE one program builder: the Python
 * engine (graphcast_amd/engine.py: stage-wise verification hooks, per-launch timing, the halo-exchange cut points of
 * the spatially partitioned step) drives this array through gc_run_program / gc_time_program instead of recording
 * its own.  `ops` has room for `capacity` entries; *n_ops receives the count (GC_EINVAL if it does not fit).  Unlike
 * gc_step_forward nothing is cleared: the caller zeroes the workspace's "tile_queue" (2 ints) before running a
 * program or a prefix of it, and owns the "range_flag" word (gc_plan_tensor).  Exchange points of a partitioned
 * model: immediately before each GC_OP_ROWMLP op tagged GC_TAG_ENC_EDGE / GC_TAG_PROC_EDGE / GC_TAG_DEC_EDGE the
 * halo suffix of "pre_grid" / "pre_s_mesh" / "pre_s_mesh" must have been filled. */
enum gc_stage_tag { GC_TAG_PREP = 0, GC_TAG_ENC_EMBED_GRID = 1, GC_TAG_ENC_PRE = 2, GC_TAG_ENC_EDGE = 3,
                    GC_TAG_ENC_NODE_MESH = 4, GC_TAG_ENC_NODE_GRID = 5, GC_TAG_PROC_PRE = 6, GC_TAG_PROC_EDGE = 7,
                    GC_TAG_PROC_NODE = 8, GC_TAG_DEC_PRE = 9, GC_TAG_DEC_EDGE = 10, GC_TAG_DEC_NODE = 11,
                    GC_TAG_DEC_OUT = 12, GC_TAG_FIXUP = 13 };
int gc_plan_program(const gc_plan* plan, const float* x, float* y, int batch, void* workspace,
                    size_t workspace_bytes, gc_op* ops, int capacity, int* n_ops);
/* A named tensor of the plan / its workspace: device pointer, rows, columns, bytes per element (4, or 2 for the
 * bfloat16 row tensors of a GC_PREC_BF16 plan).  Workspace tensors (need `workspace`): "xin", "h_grid", "pre_grid",
 * "h_grid2", "agg_grid", "h_mesh", "agg_mesh", "pre_s_mesh", "pre_r_mesh", "e_mesh_lat", "range_flag" (1 int),
 * "tile_queue" (2 ints); constants folded at creation: "h_mesh0", "d_g2m", "d_enc_mesh", "e_mesh0", "d_mesh0",
 * "d_m2g".  Returns GC_EINVAL for an unknown name (or a tensor this plan does not have). */
int gc_plan_tensor(const gc_plan* plan, void* workspace, const char* name, void** ptr, long long* rows, int* cols,
                   int* elem_bytes);
void gc_plan_destroy(gc_plan* plan);

/* Host-side packers the plan uses, exported so that a binding can check them bit for bit
 * against its own (tests/test_native_abi.py compares them with graphcast_amd/packing.py).
 * gc_host_pack_weight: w [k, n] -> the image of `prec` (chunked layouts above); `chained` selects
 *   the layer-2 K map; *scale_out receives the power of two chosen for GC_PREC_F16X3 (1 otherwise).
 *   Returns the image size in bytes (also with h_out == NULL), 0 on bad arguments.
 * gc_host_pack_edges: receiver-sorted tile packing; outputs (all optional) as documented for
 *   gc_rowmlp_desc.seg / tile_flags and gc_seg_fixup; returns n_rows (multiple of 64), < 0 on error.
 *   h_perm/h_snd/h_rcv need n_rows entries (<= 64 * ceil(n_edges / 21) is always enough),
 *   h_flags n_rows / 64, h_fix 3 * n_fix (recv, t0, t1 interleaved per entry), h_empty n_empty. */
size_t gc_host_pack_weight(int prec, int chained, const float* h_w, int k, int n, int np_cols,
                           void* h_out, float* scale_out);
int gc_host_pack_edges(int n_edges, const int* h_senders, const int* h_receivers, int n_receivers,
                       long long* h_perm, int* h_snd, int* h_rcv, int* h_flags,
                       int* h_fix, int* n_fix, int* h_empty, int* n_empty);

/* ONE tuning surface (round 6).  Every speed-only A/B switch of the library -- which kernel FORM a launch runs in, the
 * persistent grid, the tile schedule, wave priorities, which algebraic fusions the plan's program uses -- lives in this
 * struct.  None of them changes a result bit (tests/test_rowmlp_gpu.py, tests/test_step_gpu.py run the forms against
 * each other).  The GCAST_* environment variables named below are read ONCE, the first time the library needs a
 * tuning, and only INITIALISE the process default; after that the environment is never consulted again.
 *   gc_get_tuning / gc_set_tuning   the process default: what gc_rowmlp / gc_run_program use for a descriptor that does
 *                                   not pin its own form in gc_rowmlp_desc.flags, and what gc_plan_create snapshots.
 *                                   (Not synchronised against launches running on other host threads.)
 *   gc_plan_get_tuning              the snapshot a plan was created with: its program's fusions and the form pinned into
 *                                   every op of gc_plan_program / gc_step_forward -- later gc_set_tuning calls do not
 *                                   reach an existing plan's plan-level choices.
 *   gc_tuning_string                "grid_cap=512;tile_map=rr;prio=1,0,0;..." of a tuning (NULL: the process default),
 *                                   for logs and bench.py's line; thread-local storage, valid until the next call. */
typedef struct gc_tuning {
  int grid_cap;          /* GCAST_GRID_CAP: persistent workgroups per GC_LAYOUT_HALF launch, 1 .. GC_SCRATCH_SLOTS (512) */
  int tile_map_xcd;      /* GCAST_TILE_MAP=xcd: every XCD walks a contiguous eighth of the tiles (GC_TILE_XCD); default 0 */
  int prio_set;          /* GCAST_PRIO was given: its three values also apply to the GC_PREC_BF16 kernels (default there 0,0,0) */
  int prio_gemm, prio_other, prio_stage;   /* GCAST_PRIO="g,o,s": s_setprio levels (GC_PRIO); default 1,0,0 */
  int helpers;           /* GCAST_HELPERS: -1 unset (the rules below decide), 0 = four-wave form everywhere, 1 = eight-wave helper form everywhere */
  int helpers_small;     /* GCAST_HELPERS_SMALL: node-side launches of <= one tile per CU in the helper form; default 1 */
  int helpers_edge;      /* GCAST_HELPERS_EDGE: the processor edge update in the HST form: 0 never, 1 = 257 .. GC_HELPERS_EDGE_MAX_TILES tiles (default), 2 at every size */
  int helper_store;      /* GCAST_HELPER_STORE: what the staging waves of that form take over: 0 nothing, 1 residual + store, 2 + next tile's gather (default) */
  int helpers_min_rows;  /* GCAST_HELPERS_MIN_ROWS: plan: launches without gather / segment-sum from this many rows on run one eight-wave workgroup per CU; 0 never */
  int wide;              /* GCAST_WIDE: plan: ... in the WIDE form from GC_WIDE_MIN_ROWS rows on (default 1), else the helper form */
  int wide_edges;        /* GCAST_WIDE_EDGES (round 6): edge updates (segment-sum launches) in the WIDE form: bit 0 the one-pass ones
                            (encoder / decoder edge update, processor step 0), bit 1 the two-pass ones; from GC_WIDE_EDGE_MIN_TILES tiles on */
  int bf16_rows;         /* GCAST_BF16_ROWS: 0 = by rule, 64 | 128 = every GC_PREC_BF16 launch with that many rows per workgroup */
  int tile_queue;        /* GCAST_TILE_QUEUE: 0 = static tile walk whatever the descriptor says; default 1 */
  int fuse;              /* GCAST_FUSE: plan: the chained launch program (default 1); 0 = one launch per reference layer group */
  int onepass;           /* GCAST_ONEPASS: plan: one-pass edge updates where the first layer is all addends (default 1) */
  int split_tail;        /* GCAST_SPLIT_TAIL (round 6): a two-pass launch without gather / segment-sum of 513 .. 768 tiles (the
                            processor's node updates at 0.25 deg: 641) runs as TWO launches -- its first 512 tiles in the wide
                            form (one full round of 256 wide tiles), the rest in the helper form -- instead of 1.6 rounds of
                            four-wave pairs whose second round leaves half the chip idle.  Measured (profiles/r06_s4_*): that
                            stage 2 % faster, the power-limited step unchanged -- default 0 */
  int bf16_stream;       /* GCAST_BF16_STREAM (round 6): GC_PREC_BF16 edge updates without a layer-1 GEMM form every K step's hidden
                            pair on the fly from addend loads four K steps ahead instead of gathering up front (same bits);
                            default 1 */
  int wide_late;         /* GCAST_WIDE_LATE (round 6): the two-pass edge updates that the wide_edges rule puts into the WIDE form carry
                            GC_LATE_ADDENDS -- their gathered rows are added when the hidden layer is formed instead of in a burst
                            in front of layer 1 that nothing multiplies under in that form (processor edge update -3.3 %, step
                            -1.5 %).  Another fp32 association (1e-7), the same bits in every form that honours the flag.
                            default 1 */
  int split_edges;       /* GCAST_SPLIT_EDGES (round 6): the processor's two-pass edge update of 513 .. GC_WIDE_EDGE_MIN_TILES - 1 tiles
                            (the 1 deg model: 1,280; a rank of the 8-way partition at 0.25 deg: 649) runs its full rounds of 256
                            wide tiles in the WIDE form and a remainder of at most one tile per CU as a second launch in the
                            helper form (a bigger remainder: a wide round of its own) instead of ceil(tiles / 256) rounds of the
                            helper form.  The same bits (every form of a launch without GC_LATE_ADDENDS gives them).  Measured SLOWER
                            (profiles/r06_s16_*: 1 deg step 8.38 against 8.09 ms, an 8-way rank 8.58 against 8.25 ms): default 0 */
  int reserved[4];
} gc_tuning;
int gc_get_tuning(gc_tuning* out);
int gc_set_tuning(const gc_tuning* t);          /* GC_EINVAL (and no change) for a value outside its range */
int gc_plan_get_tuning(const gc_plan* plan, gc_tuning* out);
const char* gc_tuning_string(const gc_tuning* t);
#define GC_WIDE_EDGE_MIN_TILES 4096   /* 64-row tiles: >= 8 rounds of 256 wide tiles */
#ifndef GC_WIDE_LATE_DEFAULT
#define GC_WIDE_LATE_DEFAULT 1        /* gc_tuning.wide_late of a process that does not set GCAST_WIDE_LATE */
#endif
#ifndef GC_BF16_STREAM_DEFAULT
#define GC_BF16_STREAM_DEFAULT 1      /* gc_tuning.bf16_stream of a process that does not set GCAST_BF16_STREAM */
#endif
#ifndef GC_SPLIT_EDGES_DEFAULT
#define GC_SPLIT_EDGES_DEFAULT 0      /* gc_tuning.split_edges of a process that does not set GCAST_SPLIT_EDGES */
#endif
#ifndef GC_SPLIT_TAIL_DEFAULT
#define GC_SPLIT_TAIL_DEFAULT 0       /* gc_tuning.split_tail of a process that does not set GCAST_SPLIT_TAIL */
#endif
#ifndef GC_WIDE_EDGES_DEFAULT
#define GC_WIDE_EDGES_DEFAULT 3       /* gc_tuning.wide_edges of a process that does not set GCAST_WIDE_EDGES */
#endif

/* sizeof(gc_rowmlp_desc) for what == 0, sizeof(gc_op) for 1, sizeof(gc_advance_desc) for 2, sizeof(gc_model_desc) for 3, sizeof(gc_tuning) for 4, 0 otherwise:
 * lets a foreign-language binding verify its struct layout at load time. */
size_t gc_abi_sizeof(int what);

const char* gc_last_error(void);
/* Build fingerprint: "gfx950;tile=64x512;mfma=...;layouts=...;pipe=<1|2>;ring=4x16k" -- `ring` names the
 * weight ring of the GC_LAYOUT_HALF kernels (four 16 KiB quarter chunks); a profiling build appends
 * ";PROFILING_BUILD(...)" and is refused by the Python binding. */
const char* gc_build_info(void);

#ifdef __cplusplus
}
#endif
#endif  /* GCAST_H_ */
