/*
 * druglamp_hip.h — C ABI of libdruglamp_hip.so (gfx950 / MI355X).
 *
 * The reference (Lzcstan/DrugLAMP) has NO native / FFI layer: its hot path is Python nn.Module
 * composition over stock PyTorch ops.  Every entry point below therefore replaces a *sequence of
 * eager torch ops* inside one reference function; the reference file:line each one stands in for
 * is cited next to it.  The Python host side (druglamp_amd/model/…) keeps the reference's module
 * names, forward signatures and state_dict keys and calls these entry points through ctypes
 * (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers unless a
 *     parameter says "host".
 *   - the library owns nothing: the caller allocates inputs, outputs and workspaces
 *     (dl_*_workspace_bytes tells how much); no hipMalloc on any path below.
 *   - every launch goes to the hipStream_t passed in (void* here so that C callers need no HIP
 *     headers); nothing synchronises the device.
 *   - return value: 0 on success, negative dl_status otherwise; dl_last_error() gives a
 *     thread-local message.  Nothing aborts.
 *   - dtype: activations/weights are DL_F32 (exact-fp32 MFMA path, parity mode) or DL_BF16
 *     (bf16 MFMA, fp32 accumulate); statistics, biases, norm affine params, losses and all
 *     parameter gradients are fp32.
 */
#ifndef DRUGLAMP_HIP_H
#define DRUGLAMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { DL_OK = 0, DL_ERR_ARG = -1, DL_ERR_SHAPE = -2, DL_ERR_ALIGN = -3,
               DL_ERR_WORKSPACE = -4, DL_ERR_LAUNCH = -5, DL_ERR_UNSUPPORTED = -6 } dl_status;

typedef enum { DL_F32 = 0, DL_BF16 = 1 } dl_dtype;

typedef void* dl_stream;   /* hipStream_t */

const char* dl_last_error(void);
int dl_version(void);

/* ------------------------------------------------------------------------------------------
 * dl_gemm — C[m,n] = epilogue( sum_k X[m,k] * W[n,k] )
 *
 * Replaces torch.nn.Linear forward and its two autograd products wherever the hot path uses
 * them: Attention.forward query/key/value/out/fc (model/PMMA/attention.py:90-99,81-83,120),
 * Mlp.forward fc1/fc2 (model/PMMA/mlp.py:44-50), Embeddings.mol_embeddings
 * (model/PMMA/embed.py:40-42), PGCA in/out projections
 * (model/PGCA/guided_cross_attention_model.py:146-161,212,314), MHLA lin1/lin2
 * (model/PMMA/encoder.py:128-129).
 *
 * Operand storage: x_kslow == 0 -> X stored [M][K] (K contiguous, row pitch ldx elements);
 *                  x_kslow == 1 -> X stored [K][M] (M contiguous, pitch ldx).  Same for W/N.
 *   forward  y = x W^T      : X=x  (kslow 0), W=weight (kslow 0)
 *   dgrad    dx = dy W      : X=dy (kslow 0), W=weight viewed [contraction=N_out][K_in] (kslow 1)
 *   wgrad    dW = dy^T x    : X=dy (kslow 1, rows = N_out), W=x (kslow 1, rows = K_in)
 * Epilogue order: v = acc (+bias[n]); save pre_out; act; * gelu'(dact_pre); (+res if
 * res_before_dropout); dropout; (+res otherwise); store (or C += v when accumulate, f32 out).
 * split_k > 1: the contraction is cut into split_k slabs reduced by a second kernel; only the
 * plain epilogue (optionally accumulate) is allowed and out_dtype must be DL_F32.
 * ------------------------------------------------------------------------------------------ */
/* A pending second-stage reduction (see dl_reduce_batch): what dl_gemm's split-K path or dl_layernorm_bwd would have
 * launched right away, handed back to the caller instead so that several of them leave in ONE launch.  At the
 * strong-scaling batches (32-64 pairs per GPU) a step is bound by its launch count, and a third of those launches
 * were these reductions. */
typedef struct dl_reduce_item {
  int32_t kind;          /* DL_REDUCE_NONE (nothing pending), DL_REDUCE_SPLITK, DL_REDUCE_PARTIALS */
  int32_t out_dtype;     /* SPLITK: DL_F32 or DL_BF16 output */
  const float* src;      /* SPLITK: slabs [splits][M*N]; PARTIALS: partial [chunks][stride] */
  void* out;             /* SPLITK: C (row pitch ldc); PARTIALS: float [ncols] */
  int64_t mn;            /* SPLITK: M*N; PARTIALS: stride */
  int64_t ldc;
  int32_t N;             /* SPLITK: N; PARTIALS: ncols */
  int32_t splits;        /* SPLITK: slabs; PARTIALS: chunks */
  int32_t accumulate;
  int32_t M;             /* SPLITK: rows of the column sums that ride along (x_colsum), else 0 */
  const float* cs_slabs; /* SPLITK: [splits][M] or NULL */
  float* cs_out;
} dl_reduce_item;
enum { DL_REDUCE_NONE = 0, DL_REDUCE_SPLITK = 1, DL_REDUCE_PARTIALS = 2 };
enum { DL_REDUCE_BATCH_MAX = 24 };
/* Runs n <= DL_REDUCE_BATCH_MAX pending reductions in one launch; every item's result is bit-identical to what the
 * immediate path produces (same per-element summation order).  The buffers an item points to (the GEMM's workspace,
 * the LayerNorm partials) must stay untouched until this launch has run. */
int dl_reduce_batch(const dl_reduce_item* items, int32_t n, dl_stream s);

typedef struct {
  const void* X; int64_t ldx; int32_t x_kslow;
  const void* W; int64_t ldw; int32_t w_kslow;
  void* C; int64_t ldc;
  int64_t M, N, K;
  int32_t in_dtype, out_dtype;
  const float* bias;                 /* [N] or NULL */
  const void* residual; int64_t ldr; /* in_dtype, [M or res_row_mod][N] or NULL */
  int64_t res_row_mod;               /* >0: residual row = m % res_row_mod (positional table) */
  int32_t res_before_dropout;
  int32_t act;                       /* 0 none, 1 exact-erf GELU, 2 ReLU */
  void* pre_out; int64_t ldp;        /* in_dtype, pre-activation copy or NULL */
  const void* dact_pre; int64_t lddp;/* in_dtype, multiply by gelu'(dact_pre[m,n]) or NULL */
  float dropout_p; uint64_t dropout_seed;
  int32_t accumulate;
  int32_t split_k; void* workspace; size_t workspace_bytes;
  float* x_colsum;                   /* weight-gradient form only (x_kslow && w_kslow, plain epilogue): also writes
                                        x_colsum[m] = sum_k X[k][m], m < M (the bias gradient that accompanies
                                        dW = dY^T x, reference nn.Linear backward) from the X fragments the kernel
                                        streams anyway; fp32 [M] or NULL.  Counted in dl_gemm_workspace_bytes. */
  const uint64_t* dropout_seed_offset; /* DEVICE pointer or NULL: the mask is keyed by dropout_seed + *offset, read when
                                        the kernel runs — lets a hipGraph-captured step draw fresh masks per replay */
  int32_t algo;                      /* DL_GEMM_ALGO_AUTO, or DL_GEMM_ALGO_TILE128: never take the 256-wide persistent
                                        kernels (gemm_big.cuh) — outputs are bit-identical either way (tested) */
  int32_t* tile_tickets;             /* NULL, or a DEVICE array of 2 int32 that is ZERO on entry (and is zero again when the
                                        launch has finished): the one-workgroup-per-CU persistent kernel then takes its
                                        tiles beyond the first from an atomic ticket counter instead of a static stride, so
                                        a CU that starts late — e.g. because an RCCL channel workgroup of an overlapped
                                        gradient all-reduce was sitting on it — costs its share of tiles, not a second
                                        round.  Outputs are bit-identical with and without (tested). */
  dl_reduce_item* deferred;          /* NULL, or HOST pointer: a split-K call writes its slabs, describes the pending
                                        reduction here and does NOT launch it (kind = DL_REDUCE_NONE when the call took no
                                        split); the caller owns the workspace until dl_reduce_batch has run */
  int32_t prof_tag;                  /* sub-family of this product for the timing hooks (dl_prof_collect_tag): DL_TAG_*;
                                        no effect on the computation */
} dl_gemm_args;
enum { DL_TAG_OTHER = 0, DL_TAG_QKV_OUT = 1, DL_TAG_FFN = 2, DL_TAG_CONV = 3, DL_TAG_WGRAD = 4, DL_TAG_ADAPTOR = 5 };
enum { DL_GEMM_ALGO_AUTO = 0, DL_GEMM_ALGO_TILE128 = 1 };

size_t dl_gemm_workspace_bytes(const dl_gemm_args* a);
int dl_gemm(const dl_gemm_args* a, dl_stream s);
/* Two products of identical shape, layout and epilogue (the two streams of a paired PMMA block, reference
 * model/PMMA/block.py:33-62: the same layer applied to the drug stream and to the protein stream with separate
 * weights) in ONE launch when both are on the 128-tile path without split-K: at the strong-scaling batches each of them
 * fills half the CUs and the step is bound by its launch count.  Otherwise (large-tile / split-K shapes, x_colsum,
 * deferred reductions, different flags) the two run one after the other exactly as two dl_gemm calls.  Results are
 * bit-identical to two dl_gemm calls either way (tested). */
int dl_gemm_pair(const dl_gemm_args* a, const dl_gemm_args* b, dl_stream s);
/* n <= 16 weight-gradient products of DIFFERENT shapes in ONE launch (round 3): the dW = dY^T X products of one block's
 * backward (reference model/PMMA/block.py:33-62 as autograd differentiates it: fc1 / fc2 / out / fc / qkv of both streams).
 * At the strong-scaling batches each of them alone is a few 128 x 128 tiles over 8192-16384 rows: a launch, a prologue and a
 * 16- to 32-way split-K slab round trip for microseconds of matrix work.  Grouped, the tile lists are concatenated, the
 * parallelism comes from the number of products and the slab count drops to 1-4.  Every member is a dl_gemm_args in the
 * weight-gradient form (bf16 operands, x_kslow = w_kslow = 1, plain f32 output, split_k = 0, accumulate = 0; optional
 * x_colsum and `deferred`); dl_gemm_group_plan returns the slab count the library will use for each member (its workspace
 * must hold splits * M * (N + (x_colsum ? 1 : 0)) floats) or DL_ERR_UNSUPPORTED for a group it does not take (the caller
 * then issues dl_gemm calls).  A member's result is the fixed-order sum of its slabs like dl_gemm's, but NOT bit-identical
 * to the dl_gemm call (different slab count). */
int dl_gemm_group_plan(const dl_gemm_args* args, int32_t n, int32_t* splits_out);
int dl_gemm_group(const dl_gemm_args* args, int32_t n, dl_stream s);

/* column sums: out[n] (+)= sum_m X[m,n] — bias gradients of every Linear above. */
int dl_colsum(const void* X, int64_t ldx, int64_t M, int64_t N, int32_t dtype, float* out,
              int32_t accumulate, void* workspace, size_t workspace_bytes, dl_stream s);
size_t dl_colsum_workspace_bytes(int64_t M, int64_t N);

/* ------------------------------------------------------------------------------------------
 * BatchNorm over the rows of a channel-last [R][C] matrix (ProteinCNN's BatchNorm1d after each
 * Conv1d+ReLU, model/basic_model.py:176-178, with activations kept channel-last so that each Conv1d
 * is ONE dl_gemm over overlapping rows; see druglamp_amd/functional.py ProteinCNNFn).
 * Row validity: rows are grouped in windows of `win` rows; row r takes part iff
 * halo <= (r % win) < halo + valid  (win == 0: every row).  Invalid (halo) rows are written as zeros
 * by the apply kernels, so the next convolution sees zero padding.
 *   stats      : sums[0..C) = sum_r y, sums[C..2C) = sum_r y^2 over valid rows (fp32)
 *   apply_fwd  : z = (y - mean) * rstd * gamma + beta
 *   bwd_reduce : sums[0..C) = sum dz, sums[C..2C) = sum dz * yhat       (yhat = (y-mean)*rstd)
 *   bwd_apply  : dy = gamma*rstd*(dz - sums0/n - yhat*sums1/n), then * (y > 0) when relu_mask
 *                (y is the post-ReLU conv output that fed the BatchNorm)
 * ------------------------------------------------------------------------------------------ */
size_t dl_bn_workspace_bytes(int64_t R, int64_t C);
int dl_bn_stats(const void* y, int64_t R, int64_t C, int64_t win, int64_t halo, int64_t valid,
                int32_t dtype, float* sums, void* workspace, size_t workspace_bytes, dl_stream s);
/* sums (from dl_bn_stats: [sum | sum of squares]) -> batch mean, biased variance, rstd = rsqrt(var + eps), and
 * nn.BatchNorm1d's running-statistics update (momentum, unbiased running variance) in one launch; running_* may
 * be NULL. */
int dl_bn_finalize(const float* sums, int64_t n, float eps, float momentum, float* mean, float* var, float* rstd,
                   float* running_mean, float* running_var, int64_t C, dl_stream s);
int dl_bn_apply_fwd(const void* y, void* z, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, int64_t R, int64_t C, int64_t win, int64_t halo, int64_t valid,
                    int32_t dtype, dl_stream s);
int dl_bn_bwd_reduce(const void* dz, const void* y, const float* mean, const float* rstd, int64_t R,
                     int64_t C, int64_t win, int64_t halo, int64_t valid, int32_t dtype, float* sums,
                     void* workspace, size_t workspace_bytes, dl_stream s);
int dl_bn_bwd_apply(const void* dz, const void* y, const float* mean, const float* rstd,
                    const float* gamma, const float* sums, float inv_n, int32_t relu_mask, void* dy,
                    int64_t R, int64_t C, int64_t win, int64_t halo, int64_t valid, int32_t dtype,
                    dl_stream s);
/* Row-weight forms of the four BatchNorm passes (round 4: ProteinCNN on distinct rows, reference model/basic_model.py:155-180
 * over the tiled sequences of utils.py:392-412; tables: druglamp_amd/protein_plan.py).  row_w [R] fp32 replaces the window
 * rule: row_w[r] < 0 a halo row (excluded from every sum, written as zeros), 0 a context row (computed and normalised, not
 * part of the statistics), m >= 1 a row that stands for m identical rows of the reference's layout:
 *   stats_rw      sums = [sum m y | sum m y^2]                      (n of dl_bn_finalize = sum of the weights)
 *   bwd_reduce_rw sums = [sum dz | sum dz * yhat] over rows with row_w >= 0 (a row's dz already is the sum over its copies)
 *   bwd_apply_rw  dy = gamma * rstd * (dz - m * (S0 / n + yhat * S1 / n)), then the ReLU mask; halo rows zero. */
int dl_bn_stats_rw(const void* y, int64_t R, int64_t C, const float* row_w, int32_t dtype, float* sums, void* workspace,
                   size_t workspace_bytes, dl_stream s);
/* dl_bn_stats (row_w = NULL: the window rule) or dl_bn_stats_rw (row_w given) followed by dl_bn_finalize, in two launches
 * instead of three: the second stage of the column reduction also forms mean / biased var / rstd and updates the running
 * statistics (nn.BatchNorm1d, basic_model.py:160-178).  Same arithmetic in the same order: bit-identical results.  sums [2C]
 * may be NULL. */
int dl_bn_stats_finalize(const void* y, int64_t R, int64_t C, int64_t win, int64_t halo, int64_t valid, const float* row_w,
                         int32_t dtype, int64_t n, float eps, float momentum, float* sums, float* mean, float* var, float* rstd,
                         float* running_mean, float* running_var, void* workspace, size_t workspace_bytes, dl_stream s);
int dl_bn_apply_fwd_rw(const void* y, void* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       int64_t R, int64_t C, const float* row_w, int32_t dtype, dl_stream s);
int dl_bn_bwd_reduce_rw(const void* dz, const void* y, const float* mean, const float* rstd, int64_t R, int64_t C,
                        const float* row_w, int32_t dtype, float* sums, void* workspace, size_t workspace_bytes, dl_stream s);
int dl_bn_bwd_apply_rw(const void* dz, const void* y, const float* mean, const float* rstd, const float* gamma,
                       const float* sums, float inv_n, int32_t relu_mask, void* dy, int64_t R, int64_t C, const float* row_w,
                       int32_t dtype, dl_stream s);
/* Weighted tail rows (round 3: MolecularGCN's compact padding form, model/basic_model.py — the reference's virtual padding
 * nodes, handler/dataset.py:216-221, are identical in every layer and computed once).  Inside every window of `win` rows
 * the rows [lead, win) stand for w identical rows each; their incoming gradient is already the sum over those copies.
 * dl_bn_bwd_apply (run first, inv_n = 1 / the expanded row count) gave them the mean terms once; this adds the other
 * w - 1: dy[r] -= (w - 1) gamma rstd (S1 inv_n + xhat[r] S2 inv_n), in place, tail rows only. */
int dl_bn_tail_fix(void* dy, const void* y, const float* mean, const float* rstd, const float* gamma, const float* sums,
                   float inv_n, int32_t w, int64_t R, int64_t C, int64_t win, int64_t lead, int32_t dtype, dl_stream s);
/* BatchNorm1d followed by ReLU (round 5; the Linear -> BatchNorm1d -> ReLU stages of the SimSiam MLPs,
 * model/self_supervised_learning.py:126-166): z = max(0, (y - mean) rstd gamma + beta) over rows of [R][C]; the backward takes
 * the gradient dz with respect to z, recomputes the ReLU's open set from y, and gives sums [2C] = (d beta, d gamma) and dy
 * (three launches: masked partial sums, their reduction, the apply pass).  workspace: dl_bn_workspace_bytes(R, C). */
int dl_bn_apply_relu_fwd(const void* y, void* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         int64_t R, int64_t C, int32_t dtype, dl_stream s);
int dl_bn_relu_bwd(const void* dz, const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                   float inv_n, void* dy, float* sums, int64_t R, int64_t C, int32_t dtype, void* workspace, size_t workspace_bytes,
                   dl_stream s);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (eps 1e-6 inside PMMA: model/PMMA/block.py:23-27,
 * model/PMMA/encoder.py:31; eps 1e-5 for v/x_gca_norm: model/basic_model.py:115,118).
 * fwd: y = (x - mean) * rstd * gamma + beta ; mean/rstd [M] f32 saved for bwd (may be NULL).
 * bwd: dx = LN'(dy) (+ dres if given); dgamma/dbeta [D] f32 (+)= column reductions.  dy_share >= 1:
 *      row r of x takes row r / dy_share of dy (1 = one dy row per x row; L = the gradient of a mean over
 *      L tokens, `f.mean(dim=1)` at model/DrugLAMP.py:73, read in place instead of expanded to M rows).
 * ------------------------------------------------------------------------------------------ */
int dl_layernorm_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                     int64_t ldy, float* mean, float* rstd, int64_t M, int64_t D, float eps,
                     int32_t dtype, dl_stream s);
size_t dl_layernorm_bwd_workspace_bytes(int64_t M, int64_t D);
int dl_layernorm_bwd(const void* dy, int64_t lddy, int64_t dy_share, const void* x, int64_t ldx, const float* mean,
                     const float* rstd, const float* gamma, const void* dres, int64_t lddres,
                     void* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                     int64_t M, int64_t D, int32_t dtype, void* workspace, size_t workspace_bytes,
                     dl_reduce_item* deferred, dl_stream s);
/* deferred (HOST pointer or NULL): with dbeta == dgamma + D the final [2][D] column reduction is described there instead
 * of launched (dl_reduce_batch runs it; the workspace must stay untouched until then); kind = DL_REDUCE_NONE otherwise. */

/* ------------------------------------------------------------------------------------------
 * Fused attention: O = softmax(scale * Q K^T) V, no mask, no dropout
 * (attention_dropout_rate = 0, configs/default_config.py:81; PGCA dropout = 0).
 *
 * One launch covers `n_problems` independent (K,V) sets ("problems" = batch x streams), each
 * with n_heads heads.  Every problem has 1 or 2 query SEGMENTS of Lq rows each that attend to
 * the SAME K/V — this is Attention.paired_attention (model/PMMA/attention.py:44-88): segment 0
 * is the stream's own q, segment 1 the other stream's q ("guided" attention), and the two
 * outputs land side by side ([attn | attn_p], the torch.cat at attention.py:81).
 * n_segments == 1 gives plain self attention (attention.py:109-122, head dim 128) and PGCA's
 * cross attention (guided_cross_attention_model.py:290-311; Lq=256, Lk=512, 1 head).
 *
 * Addressing (element strides, all tensors dtype `dtype`, head dim contiguous):
 *   own q   (p, h, r, :)   = Q + p*q_ps + h*q_hs + r*q_rs
 *   segment 0 of problem p queries with Q(p); segment 1 with Q(partner(p)),
 *       partner(p) = (p + partner_shift) % n_problems      (paired streams: [stream][batch]
 *       ordering, partner_shift = batch)
 *   k/v     (p, h, r, :)   = K + p*k_ps + h*k_hs + r*k_rs  (same for V with v_*)
 *   o  (seg, p, h, r, :)   = O + seg*o_ss + p*o_ps + h*o_hs + r*o_rs
 *   lse(seg, p, h, r)      = LSE + ((seg*n_problems + p)*n_heads + h)*Lq + r  (f32, natural log)
 * raw_logits (optional, f32): scale*QK^T before softmax, [p][h][Lq][Lk] for segment 0 —
 * what GuidedCrossAttention returns as its 2nd output (guided_cross_attention_model.py:307,316-320).
 * head_dim in {64, 128}; strides multiples of 16 bytes; Lq, Lk arbitrary (tails are masked).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* Q; const void* K; const void* V; void* O; float* LSE; float* raw_logits;
  int64_t q_ps, q_hs, q_rs, k_ps, k_hs, k_rs, v_ps, v_hs, v_rs, o_ps, o_hs, o_rs, o_ss;
  int32_t n_problems, n_heads, n_segments, partner_shift, Lq, Lk, head_dim, dtype;
  float scale;
  int32_t algo;   /* DL_ATTN_ALGO_AUTO, or DL_ATTN_ALGO_STREAM: key tiles streamed through LDS whatever the lengths
                     (the form every long sequence takes; AUTO keeps K/V LDS-resident when Lk <= 256 at head_dim 64) */
  /* Key multiplicities (round 5; one segment only).  The last key_tail_rows keys each stand for key_tail_weight identical
   * keys of the attention the reference computes: PGCA's key / value rows are the 512 drug nodes / tokens of a molecule
   * (model/DrugLAMP.py:55-56), of which everything beyond the batch's block is one repeated padding row (virtual GCN nodes,
   * handler/dataset.py:216-221; zero token rows, utils.py:304-312).  softmax over the full key set = softmax over the distinct
   * keys with log(weight) added to a repeated key's logit; O, LSE and (backward) dQ are those of the full attention, dK / dV of
   * a tail key are the SUMS over the keys it stands for.  0 rows = plain attention. */
  int32_t key_tail_rows;
  float key_tail_weight;
} dl_attn_fwd_args;
enum { DL_ATTN_ALGO_AUTO = 0, DL_ATTN_ALGO_STREAM = 1, DL_ATTN_ALGO_TWO_PASS = 2, DL_ATTN_ALGO_ONE_PASS = 3 };
/* Backward at head_dim 64, bf16, Lk <= 256 (one segment, or two with partner(partner(p)) = p): AUTO takes the one-pass kernel
 * (dQ, dK, dV from one evaluation of P and dS) when it has a workgroup for every CU, the dQ + dK/dV kernel pair otherwise;
 * TWO_PASS / ONE_PASS force the choice (tests, A/B); ONE_PASS on an ineligible shape falls back to the pair. */
int dl_attn_fwd(const dl_attn_fwd_args* a, dl_stream s);

/* Backward.  dO is addressed like O (do_* strides); Delta has the shape of LSE and is scratch
 * filled by the call (rowsum(dO*O)).  dQ(p) receives BOTH contributions of q tensor p (its own
 * attention's segment 0 and its partner attention's segment 1) from one workgroup — nothing is
 * accumulated across launches and the result is deterministic.  dK/dV(p) sum over the segments
 * of attention p.  All three are overwritten. */
typedef struct {
  const void* Q; const void* K; const void* V; const void* O; const void* dO;
  const float* LSE; float* Delta;
  void* dQ; void* dK; void* dV;
  int64_t q_ps, q_hs, q_rs, k_ps, k_hs, k_rs, v_ps, v_hs, v_rs, o_ps, o_hs, o_rs, o_ss;
  int64_t do_ps, do_hs, do_rs, do_ss, dq_ps, dq_hs, dq_rs, dk_ps, dk_hs, dk_rs, dv_ps, dv_hs, dv_rs;
  int32_t n_problems, n_heads, n_segments, partner_shift, Lq, Lk, head_dim, dtype;
  float scale;
  int32_t algo;   /* as in dl_attn_fwd_args */
  int32_t key_tail_rows;     /* as in dl_attn_fwd_args (must match the forward call) */
  float key_tail_weight;
} dl_attn_bwd_args;
int dl_attn_bwd(const dl_attn_bwd_args* a, dl_stream s);


/* ------------------------------------------------------------------------------------------
 * MHLA token gate (MultiHeadLinearAttention.forward, model/PMMA/encoder.py:127-140):
 * logits [B][L][H] -> g = softmax over L (per b, per h); out = flat-reinterpreted product:
 * within sample b, flat index f = (j*L + l)*hd + c (j<H, hd = D/H) of the contiguous [L][D]
 * buffer is scaled by g[b][l][j].  (NOT a head split — the reference does a .view on the
 * contiguous (B,L,D) tensor.)  out = v * gate (+ v if add_residual, the caller's `mv + hv`).
 * bwd: dv, dlogits.
 * ------------------------------------------------------------------------------------------ */
int dl_token_gate_fwd(const void* v, const void* logits, void* out, float* gate_out, int64_t B,
                      int64_t L, int64_t D, int32_t H, int32_t add_residual, int32_t dtype,
                      dl_stream s);
int dl_token_gate_bwd(const void* dout, const void* v, const float* gate, void* dv, void* dlogits,
                      int64_t B, int64_t L, int64_t D, int32_t H, int32_t add_residual,
                      int32_t dtype, dl_stream s);
/* Gradient of MHLA's lin2 input through the GELU in front of it (encoder.py:127-140: logits = lin2(gelu(lin1 v))):
 * dpre[m][n] = gelu'(pre[m][n]) * sum_{h < H} dlogits[m][h] * w2[h][n], w2 = lin2.weight [H][d_diff] in the compute dtype.
 * An inner dimension of H = 8: one elementwise pass over `pre` instead of a zero-padded 64-deep GEMM (round 5). */
int dl_gate_dpre(const void* dlogits, const void* w2, const void* pre, void* dpre, int64_t M, int64_t d_diff, int32_t H,
                 int32_t dtype, dl_stream s);

/* ------------------------------------------------------------------------------------------
 * Stream concatenation of the PMMA encoder (model/PMMA/encoder.py:50: `cat((prot, mol), -1)` before
 * the self-attention layers) and its gradient.  src [S][R][row_bytes] -> dst [R][S * row_bytes];
 * inverse = 1 goes the other way (src [R][S * row_bytes] -> dst [S][R][row_bytes]).
 * ------------------------------------------------------------------------------------------ */
int dl_interleave_streams(const void* src, void* dst, int64_t R, int64_t row_bytes, int32_t S, int32_t inverse,
                          dl_stream s);
/* Symmetric-normalised, transposed adjacency of a batch of dense graphs: ahat[b][i][j] = din[i] * adj[b][j][i] * dout[j]
 * with clamped degrees^-1/2 (MolecularGCN: model/basic_model.py:137-153, 591-617 — dgl GraphConv norm='both').
 * adj (B, n, n) fp32, ahat (B, n, n) out_dtype; n <= 190. */
int dl_norm_adjacency(const float* adj, void* ahat, int64_t B, int32_t n, int32_t out_dtype, dl_stream s);
/* Neighbourhood aggregation of MolecularGCN on dense batched graphs (dgl GraphConv norm='both' `update_all(copy_u, sum)`,
 * model/basic_model.py:591-617): out[b][i][:] = sum_j A'[b][i][j] * feat[b][j][:] for the n real nodes (A' = ahat, or
 * ahat^T when transpose = 1: the gradient), out[b][i][:] = feat[b][i][:] for the virtual padding nodes n <= i < N (their
 * only edge is the self loop).  ahat (B, n, n) from dl_norm_adjacency, feat / out (B, N, C), all `dtype`; n <= 128, C = 128. */
int dl_graph_aggregate(const void* ahat, const void* feat, void* out, int64_t B, int32_t n, int32_t N, int32_t C,
                       int32_t transpose, int32_t dtype, dl_stream s);
/* cat[r] = [a[r] | b[r]] for two row-major buffers (model/DrugLAMP.py:57,66: `cat((prot_sites, guided), 2)` in front of
 * the MHLA blocks); inverse = 1 splits cat back into a and b (the gradient). */
int dl_concat2(void* a, void* b, void* cat, int64_t R, int64_t a_row_bytes, int64_t b_row_bytes, int32_t inverse,
               dl_stream s);

/* ------------------------------------------------------------------------------------------
 * Elementwise helpers on the path.
 * ------------------------------------------------------------------------------------------ */
/* y = dropout(x + pe[row % L]) — Embeddings.forward prot branch (model/PMMA/embed.py:51-52). */
int dl_add_rowmod_dropout(const void* x, const void* pe, void* y, int64_t M, int64_t D,
                          int64_t L, float p, uint64_t seed, const uint64_t* seed_offset, int32_t dtype, dl_stream s);
/* y = dropout_mask(seed)(x) / (1-p) — regenerates a forward mask for the backward pass.
 * seed_offset (here and above): DEVICE pointer or NULL, see dl_gemm_args.dropout_seed_offset. */
int dl_dropout_apply(const void* x, void* y, int64_t n_rows, int64_t D, int64_t ldx, int64_t ldy,
                     float p, uint64_t seed, const uint64_t* seed_offset, int32_t dtype, dl_stream s);
/* LLM feature ingest (input side of the path): ONE pass over the pre-extracted embedding tensor
 * x [B][S][F] produces the reference's fill bit (1 where the embedding row sums to exactly 0,
 * model/DrugLAMP.py:11-19) and the fill-bit-augmented, site-pooled features
 * pooled[b][j][:] = mean over the site_len chunks c of [x | fill][b][c*(S/site_len) + j][:]
 * (DrugLAMP.py:39-40; site_len = 1 just appends the fill bit), zero-padded to ceil8(F+1) columns so that
 * the adaptor GEMMs are aligned.  fill has x's dtype, pooled has out_dtype. */
int dl_fill_pool(const void* x, void* fill, void* pooled, int64_t B, int64_t S, int64_t F, int32_t site_len,
                 int32_t in_dtype, int32_t out_dtype, dl_stream s);
/* dx = dy * gelu'(pre) (exact-erf GELU): backward of the GELU that sits between a Linear and a LayerNorm in
 * the LLM adaptors (model/basic_model.py:189-193, DrugLAMP.py:46-52). */
int dl_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int32_t dtype, dl_stream s);
/* dst(bf16|f32) = src(f32|bf16) elementwise cast (weight casts, master fp32 -> compute dtype). */
int dl_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n,
            dl_stream s);
/* Batch assembly of the pre-extracted LLM embeddings from a DEVICE-RESIDENT packed store (replaces the
 * reference's per-sample torch.load + host loops, handler/dataset.py:186-195 and utils.py:304-334):
 *   store   [total_rows][F]  every unique protein / drug embedding, rows of one entity contiguous
 *   offsets [B] first row of sample b's entity, lengths [B] its row count
 *   out     [B][S][F]:  repeat = 1 -> the entity's rows written floor(S / len) times back to back, zeros after
 *                                     (repeat_pad, utils.py:314-324: protein side, S = 9 * 256)
 *                       repeat = 0 -> rows 0 .. min(len, S) - 1, zeros after (tail_pad, utils.py:304-312: drug side, S = 512)
 * F * sizeof(dtype) must be a multiple of 16. */
int dl_gather_pad(const void* store, const int64_t* offsets, const int32_t* lengths, void* out, int64_t B,
                  int64_t S, int64_t F, int32_t repeat, int32_t dtype, dl_stream s);
/* ProteinCNN head (model/basic_model.py:168-171): out[b][halo + l][:D] = weight[ids[b][l]][:] (nn.Embedding row
 * gather), out[b][halo + l][D] = fill[b][l] (the concatenated fill bit); `halo` zero rows on each side of every
 * sample are the conv 'same' padding of this library's channel-last layout.  weight is passed PADDED to
 * [V][D + 1] (last column ignored; 16-byte aligned rows); fill [B][L] and out [B][L + 2*halo][D + 1]; all `dtype`. */
int dl_embed_pad(const int64_t* ids, const void* weight, const void* fill, void* out, int64_t B, int64_t L,
                 int32_t V, int32_t D, int32_t halo, int32_t dtype, dl_stream s);
/* Device-side guard flags (sticky bits OR-ed into a caller-owned uint32 word; the trainer polls it):
 * the compact forms below are only valid for inputs with the padding structure of the reference's collate. */
enum { DL_FLAG_PROT_PERIOD = 1, DL_FLAG_DRUG_TOKEN_PAD = 2, DL_FLAG_GCN_NODE_PAD = 4, DL_FLAG_PLAN_ROWS = 8 };
/* ProteinCNN head on distinct rows (round 4; model/basic_model.py:168-171 over a sequence tiled by utils.py:392-412):
 * out[r][:D] = weight[ids[src[r]]], out[r][D] = fill[src[r]] for src[r] >= 0 (a flat index into ids / fill [B * L]), a zero
 * row for src[r] < 0.  weight padded to [V][D + 1] as for dl_embed_pad.  With `period` [B] given the same launch checks
 * every sample's (id, fill bit) sequence: equal at distance period[b] inside the last whole period's end E, constant on
 * [E, L) — what the row tables assume; a violation ORs DL_FLAG_PROT_PERIOD into *flags.  period[b] == 0: sample b's tables
 * keep every position (no periodic claim), nothing is checked for it. */
int dl_embed_rows(const int64_t* ids, const void* weight, const void* fill, const int32_t* src, void* out, int64_t R,
                  int32_t V, int32_t D, const int32_t* period, int64_t B, int64_t L, uint32_t* flags, int32_t dtype, dl_stream s);
/* out[i][:] = src[index[i]][:] (zeros where index[i] < 0), rows of row_bytes (a multiple of 16): the compact ProteinCNN
 * output expanded to all positions (the reference's full (B, 2304, 128) activation, basic_model.py:179). */
int dl_rows_gather(const void* src, const int32_t* index, void* out, int64_t N, int64_t row_bytes, dl_stream s);
/* out[r][:] = sum_{k < rep[r][2]} x[rep[r][0] + k * rep[r][1]][:] — the gradient of dl_rows_gather for index maps whose
 * preimages are arithmetic progressions (rep = (first, stride, count) per output row; count 0 gives a zero row); fixed
 * summation order, fp32 accumulation. */
int dl_rows_sum_strided(const void* x, const int32_t* rep, void* out, int64_t R, int64_t C, int32_t dtype, dl_stream s);
/* Guard of the drug branch's compact padding forms (reference handler/dataset.py:211-222 virtual nodes; utils.py:304-312
 * tail_pad zero rows): rows row0 .. N - 1 of every sample of x [B][N][row_bytes] must equal row row0 of sample 0 bit for bit,
 * else `code` is OR-ed into *flags. */
int dl_rows_equal_check(const void* x, int64_t B, int64_t N, int64_t row_bytes, int64_t row0, uint32_t code, uint32_t* flags,
                        dl_stream s);
/* The row tables of the ProteinCNN distinct-row layout, built on the device from the batch's residue counts (round 5).  The
 * reference tiles a protein of Lr residues with period P = Lr + 2 up to S positions (utils.py:392-412,
 * repeat_integer_label_protein) and the three 'same' convolutions (model/basic_model.py:155-180) see 7 positions to the left and
 * 8 to the right, so a sample has ~P + 31 distinct output rows; druglamp_amd/protein_plan.py states which (segments A / B / C,
 * halo rows, multiplicities) and builds the same tables on the host — the two are compared entry by entry in the tests.
 * lengths [B] int32 residue counts; R = row capacity of the tables (>= the rows the batch needs: rows behind them become
 * padding rows; fewer ORs DL_FLAG_PLAN_ROWS into *flags and writes nothing for the samples that do not fit);
 * src [R] flat input index or -1, w [R] -1 halo / 0 context / multiplicity, rep [R][3] (first, stride, count),
 * row_of [B * S] representative row of every position, period [B] (0 for samples kept in the plain layout). */
int dl_protein_plan_build(const int32_t* lengths, int64_t B, int64_t S, int64_t R, int32_t* src, float* w, int32_t* rep,
                          int32_t* row_of, int32_t* period, uint32_t* flags, dl_stream s);
/* ProteinCNN tail (model/basic_model.py:176-179 + DrugLAMP.py:39-40): the reference keeps the conv output
 * channel-first (B, C, L), REINTERPRETS that buffer with .view(B, L, C) and then site-pools it
 * (.view(B, site_len, n_site, C).mean(1)).  z is this library's channel-last conv output with `halo` zero rows
 * around every sample, [B][L + 2*halo][C]; pooled is [B][L / site_len][C].  fwd reproduces view + mean exactly in
 * one pass over z; bwd writes the full padded gradient dz (halo rows zeroed) from dpooled.  bf16 only.
 * site_len 1 (the masked-LM pass, model/self_supervised_learning.py:67-101: the reinterpretation alone) is a per-sample matrix
 * transpose and runs as one (64 x 64 tiles through LDS) when L and C are multiples of 64. */
int dl_cnn_sitepool_fwd(const void* z, void* pooled, int64_t B, int64_t L, int64_t C, int32_t halo,
                        int32_t site_len, int32_t dtype, dl_stream s);
int dl_cnn_sitepool_bwd(const void* dpooled, void* dz, int64_t B, int64_t L, int64_t C, int32_t halo,
                        int32_t site_len, int32_t dtype, dl_stream s);
/* The same through the row map of the distinct-row layout (round 4, druglamp_amd/protein_plan.py): z is the COMPACT conv
 * output [R][C]; position (b, l) is represented by row row_of[b * L + l].  fwd reads the pooling's operands through the map
 * (no expansion to [B][L][C]); bwd writes the compact gradient dz [R][C] directly: a row's gradient is the sum over the
 * positions it stands for (rep[r] = (first flat position, stride, count), count 0 = a zero row).  bf16 only. */
int dl_cnn_sitepool_rows_fwd(const void* z, const int32_t* row_of, void* pooled, int64_t B, int64_t L, int64_t C,
                             int32_t site_len, int32_t dtype, dl_stream s);
int dl_cnn_sitepool_rows_bwd(const void* dpooled, const int32_t* rep, const int32_t* row_of, void* dz, int64_t B, int64_t L,
                             int64_t C, int64_t R, int32_t site_len, int32_t dtype, dl_stream s);
/* Weight preparation: ONE launch refreshes every compute-dtype (and transposed) image of the fp32 master
 * parameters after an optimiser step — the per-parameter `.to(dtype)` / `.t().contiguous()` / `cat(q,k,v)` copies
 * a torch implementation of the reference's modules makes implicitly, batched.  items / block_map are DEVICE
 * arrays built once by the host: block_map[2b] = item index, block_map[2b+1] = 64x64 tile index in that item.
 * plain item (transpose = 0): dst[(row0 + r) * ld + c] = src[r][c];   transposed item (1): dst[c * ld + row0 + r] = src[r][c];
 * strided item (2, round 3): dst[row0 + r * ld + c * cs] = src[r][c] with any (also negative) column stride — the GEMM
 * layouts of a Conv1d weight [co][ci][k] (one item per output channel: forward Wg[co][j*ci + c] = w[co][c][j], data
 * gradient Wd[ci][j'*co + o] = w[o][ci][k-1-j']), which torch derived with flip / permute / contiguous / cast launches. */
typedef struct dl_wprep_item {
  const float* src;        /* fp32 master parameter, [rows][cols] contiguous */
  void* dst;               /* image base (out_dtype) */
  int64_t ld;              /* image leading dimension in elements (strided items: row stride) */
  int32_t rows, cols;
  int32_t row0;            /* offset of this parameter inside a concatenated image (strided items: element offset) */
  int32_t transpose;       /* 0 plain, 1 transposed, 2 strided */
  int64_t cs;              /* strided items: column stride in elements */
} dl_wprep_item;
int dl_weight_prep(const dl_wprep_item* items_dev, const int32_t* block_map_dev, int32_t n_blocks,
                   int32_t out_dtype, dl_stream s);
/* out[d] (+)= sum over rows r of x[r][d] where rows are grouped by (r % L): pe gradients. */
int dl_rowmod_sum(const void* x, float* out, int64_t M, int64_t D, int64_t L, int32_t accumulate,
                  int32_t dtype, dl_stream s);

/* ------------------------------------------------------------------------------------------
 * Loss kernels.
 * cos_rowloss: SimSiam loss_fn (model/self_supervised_learning.py:184-187):
 *   per row r: 2 - 2*cos(x_r, y_r); out_sum (+)= sum_r.  bwd gives dx (y is a detached target).
 * ntxent_stream: nt_xent_loss (self_supervised_learning.py:168-182): P = [q;k] (2n x d),
 *   logits = P P^T / T with the diagonal removed, positive of i is i+n (mod 2n), loss = sum CE / 2n,
 *   computed by streaming tiles with an online log-sum-exp; never materialises (2n)^2.
 * triplet_sigcos: ccpp_p_tri_loss (model/cross_modality.py:15-47) with distance
 *   1 - sigmoid(cos) (utils.py:571-574): all-pairs sigmoid(cos) matrix once, then the masked
 *   triplet reduction over the label matrix gt[n_p][n_d] (1 positive, 0 negative, -1 ignore).
 * ------------------------------------------------------------------------------------------ */
int dl_cos_rowloss_fwd(const float* x, const float* y, float* row_loss, float* loss_sum,
                       int64_t n_rows, int64_t D, dl_stream s);
int dl_cos_rowloss_bwd(const float* x, const float* y, float grad_scale, float* dx, int64_t n_rows,
                       int64_t D, dl_stream s);
/* Cross entropy over rows with few classes (round 5): F.cross_entropy(logits (N, C), labels, ignore_index) of the masked-LM
 * heads (model/self_supervised_learning.py:93-99; N = batch x 2304 tokens, C = 27).  logits [N][ld] (fp32 / bf16; ld >= C: the
 * GEMM in front pads the row), labels int64 [N].  fwd: lse [N] (log-sum-exp per row, kept for bwd), out2[0] = mean loss over the
 * rows whose label is not ignore_index, out2[1] = their number; fixed summation order.  As torch: NaN when no row is counted
 * (0 / 0); a label outside [0, C) other than ignore_index (torch: device-side assert) poisons the loss with NaN — never skipped silently.  bwd: dlogits [N][ldd],
 * columns < Cp written (softmax - onehot) * grad_out[0] / out2[1] for counted rows, zeros elsewhere; grad_out is a DEVICE scalar. */
size_t dl_ce_rows_workspace_bytes(int64_t N);
int dl_ce_rows_fwd(const void* logits, int64_t ld, const int64_t* labels, int64_t N, int32_t C, int64_t ignore_index, int32_t dtype,
                   float* lse, float* out2, void* workspace, size_t workspace_bytes, dl_stream s);
int dl_ce_rows_bwd(const void* logits, int64_t ld, const int64_t* labels, int64_t N, int32_t C, int64_t ignore_index, int32_t dtype,
                   const float* lse, const float* out2, const float* grad_out, void* dlogits, int64_t ldd, int32_t Cp, dl_stream s);
/* General form (round 3): the rows whose loss / gradient a call produces (side `a`, resident) are scored against a second
 * set of rows (side `b`, streamed); each side is [q rows ; k rows], n rows per half, row-major (n x d), dtype DL_F32 or
 * DL_BF16 (bf16 operands on v_mfma_f32_16x16x32_bf16, log-sum-exp / loss / gradients in fp32).  Rows are identified
 * by their index in the GLOBAL batch: row r of a side's q half is gid_offset + r, of its k half n_global + gid_offset + r;
 * a row's positive is the same sample in the other half, its own column is excluded (the reference removes the diagonal,
 * self_supervised_learning.py:172-176).  One process: a = b = {q, k, n, 0}, n_global = n (what dl_ntxent_fwd / _bwd do).
 * Data parallel (north star: the contrastive denominator sees the global batch; the reference is rank-local): a = this
 * rank's rows {q_loc, k_loc, n, rank * n}, b = the all-gathered rows {q_all, k_all, world * n, 0}, n_global = world * n.
 *   dl_ntxent_fwd_ex: row_lse[2 a.n], row_loss[2 a.n] (= lse_i - logit(i, positive)), *loss (optional) = mean row_loss.
 *   dl_ntxent_bwd_ex: da_q, da_k (fp32, a.n x d) = grad_scale / T * sum_j w_ij b_j with
 *     w_ij = [a.lse] softmax_i(j) + [b.lse] softmax_j(i) - (#lse given) [j positive of i]; a.lse / b.lse = that side's
 *     row_lse from the forward call, or NULL if the side's rows are not softmax rows of this call.  Both given = the
 *     whole gradient of the single-process loss in one pass; the data-parallel path calls it twice (local rows with
 *     their lse against the gathered rows, and the gathered rows against the local rows with their lse: the latter is
 *     the key-side gradient that a reduce-scatter returns to its owners). */
typedef struct dl_ntxent_side {
  const void* q;          /* n x d rows of the first half */
  const void* k;          /* n x d rows of the second half */
  int64_t n;
  int64_t gid_offset;
  const float* lse;       /* backward only; 2 n floats or NULL */
} dl_ntxent_side;
typedef struct dl_ntxent_args {
  dl_ntxent_side a, b;
  int64_t n_global;
  int64_t d;              /* 64 or 128 */
  int32_t dtype;          /* DL_F32 | DL_BF16, both sides */
  float temperature;
} dl_ntxent_args;
int dl_ntxent_fwd_ex(const dl_ntxent_args* a, float* row_lse, float* row_loss, float* loss, dl_stream s);
int dl_ntxent_bwd_ex(const dl_ntxent_args* a, float grad_scale, float* da_q, float* da_k, dl_stream s);
size_t dl_ntxent_workspace_bytes(int64_t n, int64_t d);
int dl_ntxent_fwd(const float* q, const float* k, int64_t n, int64_t d, float temperature,
                  float* loss, float* row_lse, void* workspace, size_t workspace_bytes,
                  dl_stream s);
int dl_ntxent_bwd(const float* q, const float* k, int64_t n, int64_t d, float temperature,
                  const float* row_lse, float grad_out, float* dq, float* dk, dl_stream s);
/* `dist` must hold dl_triplet_sigcos_buffer_floats(n_p, n_d) floats: the (n_p x n_d) distance matrix
 * first, then scratch that dl_triplet_sigcos_bwd reuses (pass the same buffer). */
size_t dl_triplet_sigcos_buffer_floats(int64_t n_p, int64_t n_d);
int dl_triplet_sigcos_fwd(const float* p_lats, const float* d_lats, const int8_t* gt, int64_t n_p,
                          int64_t n_d, int64_t dim, float margin, float* dist, float* loss,
                          float* n_tri, dl_stream s);
int dl_triplet_sigcos_bwd(const float* p_lats, const float* d_lats, const int8_t* gt,
                          const float* dist, int64_t n_p, int64_t n_d, int64_t dim, float margin,
                          const float* n_tri, float grad_out, float* dp, float* dd, dl_stream s);

/* ------------------------------------------------------------------------------------------
 * Fused AdamW over a flat fp32 parameter arena (torch.optim.AdamW semantics, decoupled weight
 * decay, bias correction; trainer.py:225-229 steps up to three of these per batch).
 * step is the 1-based step count AFTER increment.  Optionally emits a compute-dtype copy.
 * ------------------------------------------------------------------------------------------ */
int dl_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                  float grad_scale, void* param_lowp, int32_t lowp_dtype, dl_stream s);

/* ------------------------------------------------------------------------------------------
 * Kernel timing hooks used by bench.py for the roofline object: when enabled for a kernel
 * family, launches are bracketed by hipEventRecord on their own stream.
 * family: 0 gemm, 1 attn_fwd, 2 attn_bwd, 3 layernorm, 4 (unused since round 4), 5 ntxent_fwd, 6 ntxent_bwd.
 * These event lists are the library's ONLY process-global state; they exist while a family is enabled and are
 * never touched otherwise (dl_prof_enable(family, 0) frees them).
 * dl_prof_enable(family, on): on = 0 off; on = N >= 1 times one launch in N of the family, picked by a hash
 * of the launch index so that no launch slot of a periodic step is favoured (an
 * event pair costs ~6 us of stream time on gfx950: N = 1 times everything, a larger N keeps the
 * measurement from slowing the step it measures).
 * dl_prof_collect synchronises the recorded events and returns, over the TIMED launches,
 * launches / total ms / total algorithmic flops / total algorithmic bytes since dl_prof_enable.
 * dl_prof_totals returns the count and algorithmic work of ALL launches of the family since
 * dl_prof_enable, timed or not.
 * Algorithmic bytes of a dl_gemm launch: every DISTINCT operand byte once (an operand whose row pitch is smaller than
 * its row length — the implicit-im2col convolutions — spans (rows - 1) * pitch + length elements), the output, and
 * the epilogue's extra operands (pre_out, dact_pre, residual, bias, the read-back of an accumulating output).
 * ------------------------------------------------------------------------------------------ */
int dl_prof_enable(int32_t family, int32_t on);
int dl_prof_collect(int32_t family, int64_t* launches, double* total_ms, double* total_flops,
                    double* total_bytes);
int dl_prof_totals(int32_t family, int64_t* launches, double* total_flops, double* total_bytes);
/* The same per sub-family tag (dl_gemm_args.prof_tag) of a family: the timed launches' count / ms / algorithmic flops /
 * algorithmic bytes, and the count / flops / bytes of ALL the tag's launches since dl_prof_enable. */
int dl_prof_collect_tag(int32_t family, int32_t tag, int64_t* timed_launches, double* timed_ms, double* timed_flops,
                        double* timed_bytes, int64_t* all_launches, double* all_flops, double* all_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DRUGLAMP_HIP_H */
