/*
 * hg_aggr.h -- C ABI of libhgaggr.so, the MI355X (gfx950) backend for the fused
 * vertex -> hyperedge -> vertex aggregation of HGNNConv / UniGNNConv.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  It
 * is what the reference's operator layer binds instead of its CUDA launchers.
 * Each entry point cites the reference interface it replaces (paths relative
 * to the reference tree).  INTEGRATION.md shows the binding stubs.
 *
 * Conventions (reference: HyperGsys/include/util/check.cuh:11-12):
 *   Index = int32, DType = float32, feature matrices row-major [rows, F].
 *   H is the N x M vertex-by-hyperedge incidence matrix; the caller supplies
 *   H_T in CSR: csrptr_t[M+1], colind_t[nnz] (row = hyperedge, entries = member
 *   vertices), exactly the `H_T_csrptr` / `H_T_colind` tensors of
 *   HyperGsys/hypergraph.py:63-70.
 *
 * All device pointers must belong to the current HIP device.  Every function
 * returns HG_OK or a negative hg_status; nothing aborts, nothing throws.
 * hg_last_error() gives a thread-local message for the last failure.
 * Calls that take a stream only enqueue work on it; they do not synchronise.
 */
#ifndef HG_AGGR_H
#define HG_AGGR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_AGGR_VERSION 410 /* round 4: + hg_aggr_linear_res_dev_f32, HG_LIN_BF16X6 + hg_linear_pack_ex_f32 */

#if defined(__GNUC__)
#define HG_API __attribute__((visibility("default")))
#else
#define HG_API
#endif

typedef enum hg_status {
  HG_OK = 0,
  HG_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, bad CSR) */
  HG_ERR_NOMEM = -2,       /* host or device allocation failed */
  HG_ERR_HIP = -3,         /* a HIP runtime call or kernel launch failed */
  HG_ERR_WORKSPACE = -4,   /* caller workspace smaller than hg_plan_workspace_bytes */
  HG_ERR_UNSUPPORTED = -5  /* combination not implemented */
} hg_status;

typedef void *hg_stream_t; /* a hipStream_t; NULL = the legacy default stream */
typedef struct hg_plan hg_plan;

/* Kernel family used by hg_aggr_fused_f32. */
typedef enum hg_variant {
  HG_VARIANT_AUTO = 0,
  /* Atomic-free two-phase pull: Xe = S_e . (H^T X) over H_T CSR, then
   * Y = S_v . (H Xe) over H CSR.  Deterministic; bit-exact with the CPU
   * reference order for every row of at most `short_max` entries. */
  HG_VARIANT_PULL = 1,
  /* The reference's scheme (HGNNAggr_forward_kernel, hgnnaggr_cuda.cu:14-47):
   * hyperedge partial sum kept in registers, scattered with fp32 atomics. */
  HG_VARIANT_PUSH_ATOMIC = 2,
  /* Fused pull: vertex panels whose incident hyperedge sums are recomputed in
   * the workgroup and staged in LDS, so the M x F hyperedge feature matrix
   * never round-trips through HBM.  Hyperedges longer than t_big are materialised
   * by a pre-pass instead.  Vertices with more hyperedges than a panel holds are
   * cut into pieces (partial rows, summed by a fixup pass in a fixed order) or, the
   * heaviest ones of a large graph, served by the hub pass: persistent workgroups
   * that keep the hubs' running sums in registers while they stream the hyperedges,
   * each hyperedge sum computed once per pass however many hubs it feeds.
   * Rows of at most vdeg_max hyperedges keep HG_VARIANT_PULL's arithmetic order. */
  HG_VARIANT_FUSED = 3
} hg_variant;

typedef struct hg_plan_opts {
  int32_t short_max;  /* rows longer than this are cut out as wave tasks (default 32) */
  int32_t split_len;  /* a wave task covers at most this many entries (default 512)   */
  int32_t panel_rows; /* rows per row-panel workgroup (default 128)                   */
  int32_t panel_nnz;  /* index entries staged in LDS per panel (default 1024)         */
  int32_t flags;      /* HG_PLAN_* bits                                              */
  int32_t t_big;      /* fused: recompute hyperedges of at most this many members (8)  */
  int32_t fused_tile_bytes; /* fused: LDS tile budget per workgroup -> slots per panel (16384) */
  int32_t fused_steps; /* fused: a panel's hop-1 stream holds at most this many entries per lane group
                          (0 = 4 entries per slot on average); 8 = one batch of row loads in flight */
} hg_plan_opts;

#define HG_PLAN_HOST_ONLY 1 /* build the schedule on the host, upload nothing (tests) */
#define HG_PLAN_NO_XCD_REMAP 2 /* keep blockIdx -> panel identity mapping */
#define HG_PLAN_DFS_ORDER 4 /* fused: panel rows in plain depth-first order (no greedy growth) */
#define HG_PLAN_NO_ROW_STREAM 16 /* pull: always the general row-gather kernel (panels + wave tasks), never the streaming one */
#define HG_PLAN_NO_HUB_PASS 8 /* fused: no register-hub pass; every vertex too big for a panel is cut into pieces */

typedef struct hg_plan_info {
  int32_t N, M;
  int64_t nnz;
  int32_t short_max, split_len, panel_rows, panel_nnz, flags;
  /* hop 1 walks H_T (rows = hyperedges), hop 2 walks H (rows = vertices) */
  int32_t panels[2];   /* row panels                                   */
  int32_t tasks[2];    /* wave tasks for rows longer than short_max    */
  int32_t partials[2]; /* partial-sum slots of rows split over tasks   */
  int32_t fixups[2];   /* rows whose partials are summed in a 2nd pass */
  int32_t max_len[2];  /* longest row of each CSR                      */
  int64_t device_bytes; /* device memory held by the plan              */
} hg_plan_info;

/* Shape of the F-dependent fused schedule (hg_plan_prepare). */
typedef struct hg_fused_info {
  int32_t cap;      /* hyperedge slots (LDS tile rows) per vertex panel */
  int32_t t_big, vdeg_max;
  int32_t panels;
  int32_t n_mat;    /* materialised hyperedges */
  int32_t n_hub;    /* register hubs: running sums kept on chip by the hub pass */
  int64_t slots;    /* sum over panels of distinct hyperedges touched */
  int64_t member_entries; /* row gathers of one fused aggregation (panels only) */
  int32_t n_split;  /* vertices cut into pieces (panel rows that leave partial sums) */
  int32_t fixups;   /* rows summed from partial rows after the panels (hubs + split vertices) */
  int32_t hub_rounds, hub_workgroups; /* hub pass: rounds of hyperedge slots, persistent workgroups */
  int64_t hub_entries; /* row gathers of the hub pass */
  int64_t hub_pairs;   /* (hub, hyperedge) incidences served from the LDS tile */
  int64_t partial_rows; /* partial rows in the workspace (hub parts x workgroups + pieces) */
  int32_t record_words_max; /* largest panel record, 32-bit words (staged in LDS beside the tile) */
  int32_t stream_steps_max; /* longest entry stream of a panel: row gathers per lane group */
  int32_t lds_bytes;        /* LDS per panel workgroup without scale staging: tile + largest record */
  int32_t reserved;
} hg_fused_info;

HG_API int hg_version(void);
HG_API const char *hg_last_error(void);
HG_API const char *hg_status_string(int status);

/* ---- scheduler, reference-compatible ------------------------------------
 * Replaces balance_schedule (HyperGsys/balancer.py:4-33) and its C++ twin
 * hgnn_ef_full_balance_cpu (include/taskbalancer/balancer_kernel.cuh:229-259):
 * hyperedge r with n_r members is cut into w = ceil(n_r/ngs) partitions and
 * w*w (read j, write i) tasks, i outer / j inner; `key` holds the partition
 * starts plus a trailing nnz sentinel.  Host arrays in, host arrays out.
 * Two-call protocol: pass key == NULL to get *n_key / *n_group, then call
 * again with arrays of those lengths.  HG_ERR_INVALID when nnz == 0 (the
 * reference raises IndexError there). */
HG_API int hg_balance_schedule(int32_t nrow, int32_t ngs, const int32_t *csrptr_host,
                        int64_t *n_key, int64_t *n_group, int32_t *key,
                        int32_t *row, int32_t *group_st, int32_t *group_ed);

/* ---- input format ----------------------------------------------------------
 * MatrixMarket reader with the semantics of the reference's DataLoader
 * (include/dataloader/dataloader.hpp:22-180): 1-based `row col [value]` lines,
 * values dropped, entries sorted by (row, col); `symmetric` files are mirrored
 * and de-duplicated, `general` files keep duplicates.  Returns H (rows =
 * vertices) in CSR and its stable transpose H_T (rows = hyperedges, members
 * ascending).  The four arrays are malloc'ed; release each with hg_free. */
HG_API int hg_mtx_read(const char *path, int32_t *nrow, int32_t *ncol, int64_t *nnz,
                       int32_t **H_ptr, int32_t **H_ind, int32_t **HT_ptr, int32_t **HT_ind);
HG_API void hg_free(void *p);

/* ---- plan ------------------------------------------------------------------
 * The plan is this backend's own schedule (what hgnn_balancer,
 * include/taskbalancer/balancer.cuh:189-275, is to the reference kernels): it
 * derives H in CSR (vertex -> incident hyperedges, ascending) from H_T and
 * cuts both CSRs into row panels and wave tasks for 64-lane wavefronts.  It
 * depends only on the sparsity structure; build once per hypergraph, reuse for
 * every feature width and every call.  opts may be NULL (defaults). */
HG_API int hg_plan_create_host(hg_plan **out, int32_t N, int32_t M,
                        const int32_t *csrptr_t_host,
                        const int32_t *colind_t_host,
                        const hg_plan_opts *opts);
/* Same, from device arrays (copies them to the host on `stream`, synchronises
 * that stream once).  Not capturable into a hipGraph. */
HG_API int hg_plan_create_device(hg_plan **out, int32_t N, int32_t M, int64_t nnz,
                          const int32_t *csrptr_t_dev,
                          const int32_t *colind_t_dev,
                          const hg_plan_opts *opts, hg_stream_t stream);
HG_API void hg_plan_destroy(hg_plan *plan);
HG_API int hg_plan_get_info(const hg_plan *plan, hg_plan_info *info);
/* Host copy of the derived H CSR (tests / CLI): ptr_v[N+1], ind_v[nnz]. */
HG_API int hg_plan_get_vertex_csr(const hg_plan *plan, int32_t *ptr_v_host,
                           int32_t *ind_v_host);
/* Device pointers of the derived H CSR (valid for the plan's lifetime). */
HG_API int hg_plan_get_vertex_csr_device(const hg_plan *plan, const int32_t **ptr_v_dev,
                                  const int32_t **ind_v_dev);
/* Build (and upload) the part of the plan that depends on the feature width --
 * the fused variant's vertex panels -- ahead of time.  hg_aggr_fused_f32 does
 * this itself on first use of a width, but that first call allocates device
 * memory and so cannot be captured into a hipGraph.  info may be NULL. */
HG_API int hg_plan_prepare(const hg_plan *plan, int32_t F, hg_fused_info *info);
/* Optional: pre-gather the degree / weight vectors into the fused schedule's panel
 * order for feature width F.  Later hg_aggr_fused_f32 calls (fused variant) that pass
 * exactly these three pointers read the scales with coalesced loads instead of one
 * scattered 4-byte gather per hyperedge slot.  Re-bind after changing the vectors'
 * contents; passing other pointers simply bypasses the binding, and binding three NULL
 * pointers removes it (later calls gather their scales themselves: the safe choice for
 * vectors another library rewrites in place).  The binding is keyed on addresses only.
 * When W is given the call also checks once whether every W[e] is exactly 1.0f (what the
 * reference's models pass, model/ugsys/hgnn.py:12) and, if so, later calls skip that
 * multiplication -- the identity, bit for bit; this check synchronises `stream`.
 * Allocates on the first call per width (not capturable); enqueues one small kernel on
 * `stream` -- a caller that aggregates on a different stream orders the two itself. */
HG_API int hg_plan_bind_scales(const hg_plan *plan, int32_t F, const float *degE,
                               const float *degV, const float *W, hg_stream_t stream);
/* The variant HG_VARIANT_AUTO resolves to for feature width F (builds the
 * F-dependent schedule if the choice needs it); negative hg_status on error. */
HG_API int hg_plan_auto_variant(const hg_plan *plan, int32_t F);
/* Timed choice for feature width F, the counterpart of the reference's tuner (HyperGAggr_tune,
 * include/hgnnAgg.cuh:1115-1157: it times 20 partition sizes and keeps the fastest).  Runs the candidates on
 * the caller's buffers -- the fused schedule, and the pull variant with either kernel for each hop (streaming
 * row gather / panels + wave tasks) -- `iters` times each between two events on `stream`, waits for them, and
 * pins what HG_VARIANT_AUTO does for this width from then on.  Matters on launch-bound graphs (one dataset-sized
 * hypergraph), where the static rule cannot see which kernel's dependent round trips are shorter.  Same
 * arguments as hg_aggr_fused_f32 (Y ends up holding the result); workspace as after hg_plan_prepare.  The
 * candidates differ in summation order, so the low bits of later AUTO results depend on which one won; do not
 * call it while other threads use the plan.  info may be NULL. */
typedef struct hg_tune_info {
  int32_t variant;           /* HG_VARIANT_FUSED or HG_VARIANT_PULL */
  /* kernel of each pull hop (hop 0: vertices -> hyperedges), k0 + 3 * k1 with k = 0: streaming row gather,
   * 1: row panels + wave tasks, 2 (graphs of at most 2^18 incidences): the same kernel on the latency schedule --
   * every row of more than 8 entries is a wave task, its entries spread over the wave's lane groups and in
   * flight at once, instead of a chain of dependent load batches in one lane group */
  int32_t pull_hop_kernels;
  float us[10];              /* microseconds per call: fused, pull with hop kernels 0 .. 8 (negative: not run) */
  int32_t reserved;
} hg_tune_info;
HG_API int hg_plan_tune_f32(const hg_plan *plan, int32_t F, const int32_t *csrptr_t, const int32_t *colind_t,
                            const float *X, const float *degE, const float *degV, const float *W, float *Y,
                            void *workspace, size_t workspace_bytes, int32_t iters, hg_stream_t stream,
                            hg_tune_info *info);
/* Host copy of one hop's schedule as int32 quadruples (tests, tools): panels
 * {row0, nrows, nnz0, nnz_cnt}, tasks {row, beg, end, slot}, fixups {row,
 * first_slot, count, 0}; sizes from hg_plan_get_info.  Pointers may be NULL. */
HG_API int hg_plan_get_schedule(const hg_plan *plan, int32_t hop, int32_t *panels,
                                int32_t *tasks, int32_t *fixups);
/* Bytes of scratch hg_aggr_fused_f32 needs for feature width F, a multiple of 256: the larger
 * of the pull layout (the M x F hyperedge feature matrix plus partial-sum slots) and, once the
 * width's fused schedule exists, its layout (materialised rows, partial rows of hub vertices and
 * pieces).  The call resolves HG_VARIANT_AUTO for this width first -- building the fused schedule
 * on the host if that is the choice (once per width; milliseconds for dataset-sized graphs,
 * seconds for 10^7 incidences) -- so the size covers what an AUTO call will run.  A caller that
 * forces HG_VARIANT_FUSED on a plan whose AUTO choice is pull calls hg_plan_prepare first; a
 * fused call with a workspace sized before that returns HG_ERR_WORKSPACE, never writes past it. */
HG_API size_t hg_plan_workspace_bytes(const hg_plan *plan, int32_t F);

/* ---- the hot path ------------------------------------------------------------
 * Y[v,:] = degV[v] * sum_{e contains v} ( degE[e] * W[e] * sum_{u in e} X[u,:] )
 *
 * Replaces hgnnaggr_fp_cuda (source/hgnnaggr/hgnnaggr_cuda.cu:350-406),
 * unignnaggrdeg / unignnaggr launchers (source/unignnaggr/unignnaggr_cuda.cu:
 * 392-488) and HyperGAggr_device (include/hgnnAgg.cuh:985-1038).
 *   degE, W : [M] or NULL (factor 1);  degV : [N] or NULL.
 *     hgnnaggr      -> degE, degV, W      unignnaggrdeg -> degE, degV, W = NULL
 *     unignnaggr / aggr_proto kernels -> all three NULL
 *   Arithmetic order is the reference test's (test/hgnn_test.py:56-63):
 *     Xe = ((sum X) * degE) * W ;  Y = (sum Xe) * degV.
 *   Y is fully overwritten (no pre-zeroing needed); inputs are not modified.
 *   X, Y: row-major [N, F], any F >= 1, 4-byte aligned (16-byte alignment and F % 4 == 0 are not required:
 *   rows of more than 8 floats move as 16-byte lanes either way).
 *   workspace: device scratch of at least hg_plan_workspace_bytes(plan, F),
 *   256-byte aligned, private to this call until it completes on `stream`.
 *   csrptr_t / colind_t must be the arrays the plan was built from. */
HG_API int hg_aggr_fused_f32(const hg_plan *plan, int32_t F,
                      const int32_t *csrptr_t, const int32_t *colind_t,
                      const float *X, const float *degE, const float *degV,
                      const float *W, float *Y, void *workspace,
                      size_t workspace_bytes, int32_t variant,
                      hg_stream_t stream);

/* Aggregation with the layer's dense projection folded in: the
 * reference's HyperGsysHGNN / HyperGsysUinGINConv.forward run `X = self.W(X)` (nn.Linear
 * without bias, model/ugsys/hgnn.py:22-23, unigin.py:20-21) and then the aggregation on
 * the projected rows.  The aggregation is linear, so
 *     Y[N, F_out] = Aggr(X * Wlin^T) = Aggr(X) * Wlin^T
 * and this entry point aggregates the F_in-wide rows once and multiplies each finished
 * row by Wlin^T on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) before it is
 * stored: X is read once and Y written once, the projected matrix never exists in HBM.
 * Results equal the two-step form up to fp32 rounding order (both are fp32 fma chains).
 *   Wlin   : [F_out, F_in] row-major device array (nn.Linear.weight)
 *   wfrag  : F_out * F_in floats, 16-byte aligned: Wlin in MFMA fragment order.  Fill it with
 *            hg_linear_pack_f32 after every weight update.  A wave keeps the fragments of
 *            its output-column tile in registers; in this order they are K/16 coalesced
 *            16-byte reads per lane (read from the row-major matrix they would touch more
 *            cache lines per panel than the whole gather of X)
 *   F_in in {32, 64, 128}; F_out a positive multiple of 16 (else HG_ERR_UNSUPPORTED and
 *   the caller runs its own linear followed by hg_aggr_fused_f32)
 * Where the fused panels cannot take the epilogue (pull variant, hub vertices) the
 * aggregated rows go to the workspace and a standalone MFMA kernel projects them; the
 * workspace is therefore hg_aggr_linear_workspace_bytes, not hg_plan_workspace_bytes. */
HG_API int hg_linear_pack_f32(int32_t F_out, int32_t F_in, const float *Wlin, float *wfrag,
                              hg_stream_t stream);
/* Flags of the `relu` argument of hg_aggr_linear_res(_dev)_f32 (bit 0 is the activation, as before).
 * HG_LIN_BF16X6: the fused panels' matrix phase at F_in = 128 may compute every fp32 product as six bf16
 * products (x = h + m + l in bf16 up to 2^-24 |x|, bf16 x bf16 is exact in fp32, the pipe accumulates in fp32;
 * the three smallest cross terms, < 2^-23 |a b|, are dropped -- below one rounding of the fp32 product) on
 * v_mfma_f32_16x16x32_bf16: 3/8 of the matrix-pipe cycles of the fp32 form, the same error bound
 * (tests: both forms within 1e-5 x row mass of the float64 answer; measured maxima side by side in DESIGN.md).
 * The caller then passes a wfrag of hg_linear_pack_floats(F_out, F_in, HG_LIN_BF16X6) floats filled by
 * hg_linear_pack_ex_f32 with the same flag: the fp32 fragments (hub vertices, the other variants and every
 * other width still use them) followed by Wlin's three bf16 planes.  Without the flag nothing changes.
 * Range: bf16 keeps fp32's exponent, so finite operands below 2^127 in magnitude behave as in the fp32 form;
 * an infinite operand (or one that rounds to infinity in bf16) yields NaN where the fp32 form yields +-inf. */
#define HG_LIN_RELU 1
#define HG_LIN_BF16X6 2
HG_API size_t hg_linear_pack_floats(int32_t F_out, int32_t F_in, int32_t flags);
HG_API int hg_linear_pack_ex_f32(int32_t F_out, int32_t F_in, const float *Wlin, float *wfrag, int32_t flags,
                                 hg_stream_t stream);
HG_API size_t hg_aggr_linear_workspace_bytes(const hg_plan *plan, int32_t F_in);
/* The same pass with a whole UniGNN layer folded in.  For every vertex v
 *     t     = ca * Aggr(X)[v] + cb * R[v]          (R NULL: t = ca * Aggr(X)[v])
 *     T_out[v] = t                                  (if T_out != NULL; the backward pass needs it)
 *     Y[v]  = act(t * Wlin^T),  act = relu if (relu & HG_LIN_RELU) else identity; relu & HG_LIN_BF16X6: see above
 * R and T_out are [N, F_in] row-major, 16-byte aligned.  This is one HyperGsysUniGCNII layer
 * (model/ugsys/unigcnii.py:19-21 with the relu of model/gnn.py:199): Xi = (1-alpha) Xv + alpha X0,
 * out = (1-beta) Xi + beta W(Xi) = Xi * ((1-beta) I + beta W)^T -- pack that matrix as wfrag --
 * and one HyperGsysUinGINConv layer (unigin.py:20-22): (1+eps) W(X) + Aggr(W(X)) =
 * ((1+eps) X + Aggr(X)) * W^T.  Arithmetic: the two products and the sum of t are separate fp32
 * operations in that order, then the MFMA fma chain; the value equals the reference's layer up
 * to fp32 rounding order. */
HG_API int hg_aggr_linear_res_f32(const hg_plan *plan, int32_t F_in, int32_t F_out,
                                  const int32_t *csrptr_t, const int32_t *colind_t, const float *X,
                                  const float *degE, const float *degV, const float *W,
                                  const float *wfrag, const float *R, float ca, float cb, int32_t relu,
                                  float *T_out, float *Y, void *workspace, size_t workspace_bytes,
                                  int32_t variant, hg_stream_t stream);
/* hg_aggr_linear_res_f32 with cb read from device memory when the kernel runs (cb_dev: one float, not NULL): for a
 * LEARNED scalar -- UniGIN's 1 + eps (unigin.py:14,22) -- the caller neither reads the value back to the host
 * (a device-to-host sync per layer and step) nor bakes it into a captured hipGraph. */
HG_API int hg_aggr_linear_res_dev_f32(const hg_plan *plan, int32_t F_in, int32_t F_out,
                                      const int32_t *csrptr_t, const int32_t *colind_t, const float *X,
                                      const float *degE, const float *degV, const float *W,
                                      const float *wfrag, const float *R, float ca, const float *cb_dev,
                                      int32_t relu, float *T_out, float *Y, void *workspace,
                                      size_t workspace_bytes, int32_t variant, hg_stream_t stream);
/* The linear's weight gradient, C[F_a, F_b] = A^T B with A [nrows, F_a], B [nrows, F_b] row-major:
 * dWlin = dY^T T in the backward pass of the layers above (the contraction runs over the vertices).
 * Every element of A and B is read once, straight into fp32 MFMA operands; workgroups are reduced
 * in a fixed order (deterministic).  F_a, F_b multiples of 16 with F_a * F_b <= 4096 -- one workgroup's
 * accumulators hold the whole output -- or both multiples of 64 up to 512: 64 x 64 blocks of the output as
 * independent contractions over the same rows (128 x 128: 3 x rocBLAS).  Else
 * HG_ERR_UNSUPPORTED: use a BLAS.  workspace = hg_linear_wgrad_workspace_bytes (0 = unsupported shape). */
HG_API size_t hg_linear_wgrad_workspace_bytes(int64_t nrows, int32_t F_a, int32_t F_b);
HG_API int hg_linear_wgrad_f32(int64_t nrows, int32_t F_a, int32_t F_b, const float *A, const float *B,
                               float *C, void *workspace, size_t workspace_bytes, hg_stream_t stream);
/* The projection alone, Y[nrows, F_out] = T[nrows, F_in] * Wlin^T, on the same MFMA kernel the
 * fallback path of hg_aggr_linear_f32 uses (same width limits, same packed wfrag). */
HG_API int hg_linear_rows_f32(int64_t nrows, int32_t F_in, int32_t F_out, const float *T,
                              const float *wfrag, float *Y, hg_stream_t stream);
HG_API int hg_aggr_linear_f32(const hg_plan *plan, int32_t F_in, int32_t F_out,
                              const int32_t *csrptr_t, const int32_t *colind_t, const float *X,
                              const float *degE, const float *degV, const float *W,
                              const float *wfrag, float *Y, void *workspace, size_t workspace_bytes,
                              int32_t variant, hg_stream_t stream);

/* One hop only (CSR times dense with unit values, optional row scales):
 *   dst[r,:] = scaleB[r] * scaleA[r] * sum_{p in row r} src[ind[p],:]
 * hop = 0 walks H_T (nrows = M, src has N rows), hop = 1 walks the derived H
 * (nrows = N, src has M rows).  The two-step comparator and tests use it; it
 * is the own-kernel counterpart of csrspmm_cusparse (include/spmm/spmm.cuh:
 * 22-77). */
HG_API int hg_gather_rows_f32(const hg_plan *plan, int32_t hop, int32_t F,
                       const int32_t *csrptr_t, const int32_t *colind_t,
                       const float *src, const float *scaleA,
                       const float *scaleB, float *dst, void *workspace,
                       size_t workspace_bytes, hg_stream_t stream);

/* first_aggr = "max" pieces (hgnnaggr_max, source/hgnnaggr/hgnnaggr_cuda.cu:144-208).
 * hg_gather_max_f32: Xe[e,k] = (max_{u in e} X[u,k], start -1e5, strict >) * (degE[e]*W[e]),
 * record[e,k] = winning vertex (0 if none).  The second hop is hg_gather_rows_f32(hop = 1).
 * hg_scatter_record_f32 (backward): Y = 0; Y[record[e,k], k] += T[e,k] * degV[record[e,k]]. */
HG_API int hg_gather_max_f32(int32_t M, int32_t F, const int32_t *csrptr_t, const int32_t *colind_t,
                             const float *X, const float *degE, const float *W, float *Xe,
                             int32_t *record, hg_stream_t stream);
HG_API int hg_scatter_record_f32(int32_t N, int32_t M, int32_t F, const float *T,
                                 const int32_t *record, const float *degV, float *Y,
                                 hg_stream_t stream);

/* The reference's kernel driven by the reference's schedule tensors
 * (HGNNAggr_forward_kernel(+_sf), hgnnaggr_cuda.cu:14-84; unweighted twins
 * hgnnAgg.cuh:33-53, 98-167): task g reads partition group_st[g] and scatters
 * to partition group_ed[g] of hyperedge group_row[g] with fp32 atomics.  Pass
 * group_key == NULL for one task per hyperedge (needs csrptr_t).  Y is zeroed
 * on `stream` first (the reference's torch::zeros, hgnnaggr_cuda.cu:374). */
HG_API int hg_aggr_push_groups_f32(int32_t N, int32_t M, int32_t F, int64_t n_group,
                            const int32_t *group_key, const int32_t *group_row,
                            const int32_t *group_st, const int32_t *group_ed,
                            const int32_t *csrptr_t, const int32_t *colind_t,
                            const float *X, const float *degE,
                            const float *degV, const float *W, float *Y,
                            hg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HG_AGGR_H */
