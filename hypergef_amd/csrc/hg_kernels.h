// Kernel argument blocks and launchers shared by hg_kernels.hip and hg_api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hg_internal.h"

namespace hg {

struct GatherArgs {
  const int32_t *ptr;   // CSR row pointers [nrows + 1]
  const int32_t *ind;   // CSR column indices
  const float *src;     // gathered table [*, F]
  float *dst;           // output [nrows, F]
  const float *scaleA;  // per-row factor applied first, or null
  const float *scaleB;  // per-row factor applied second, or null
  float *partial;       // partial-sum slots [nslots, F]
  const int32_t *scale_map;  // row -> index into scaleA/scaleB, or null (identity)
  const int32_t *dst_map;    // row -> output row, or null (identity)
  const Panel *panels;
  const Task *tasks;
  int32_t npanels, ntasks, n_task_blocks;
  int32_t F;
  int32_t panel_rows, panel_nnz;  // LDS carve-up
  int32_t xcd_remap;
  int32_t nt_dst = 0;  // rows of dst are final output nothing reads again in this call chain: streaming (nt) stores
};

// What happens to an aggregated row t = Aggr(X)[v] on its way through the linear epilogue:
//   t' = ca * t + cb * R[v]   (R null: t' = ca * t),   T_out[v] = t' if wanted,
//   Y[v] = act(t' * Wlin^T),  act = relu or identity.
// One UniGCNII layer ((1-a) Xv + a X0, then (1-b) Xi + b W(Xi), relu: model/ugsys/unigcnii.py:19-21,
// model/gnn.py:196-199) or UniGIN layer ((1+eps) W(X) + Aggr(W(X)): unigin.py:20-22) is one such pass.
struct LinEpilogue {
  const float *R = nullptr;
  float ca = 1.f, cb = 0.f;
  const float *cb_dev = nullptr;  // non-null: cb is this device scalar, read when the kernel runs (a learned 1 + eps)
  int32_t relu = 0;
  float *T_out = nullptr;
  const void *wsplit = nullptr;  // Wlin as three bf16 planes in fragment order (launch_linear_pack_split): the fused panels' K = 128
                                 // matrix phase may then run as six bf16 products per fp32 product (HG_LIN_BF16X6)
};

struct FusedArgs {
  int32_t npanels;
  const int32_t *rec;     // packed per-panel records (hg_fused.cpp, pack_records)
  const FRec *rec_tab;    // per panel: record offset / length / list positions
  const int32_t *eid_all; // hyperedge id of every slot, record order (-1: materialised)
  int32_t max_rec_words;  // LDS space for one record
  int32_t ng;             // lane groups the records were packed for
  int32_t cap, rows_cap;  // LDS tile rows (hyperedge slots) and rows per panel
  int32_t dv_regs = 0;    // set by the launcher: bound degV travels in registers, not through LDS (fused_packed_kernel)
  const float *X, *Xe_mat, *degE, *W, *degV;
  const float *bsA, *bsB, *bsD;  // bound scales in panel order (or null: gather from degE/W/degV)
  float *Y;
  float *partial = nullptr;  // partial rows (pieces of split vertices): a row id with bit 31 set lands here
  int32_t F;
  int32_t xcd_remap;
  int32_t x_bytes;        // byte size of X if it fits a buffer descriptor (< 2 GiB), else 0
  int32_t mat_bytes;      // same for the materialised table
  int32_t nrows_x;        // rows of X
  int32_t nrows_mat = 0;  // rows of the materialised table
  int32_t debug = 0;      // ablation / stamp bits (experiments only)
  int32_t y_nt = 0;       // rows of Y leave with the streaming (nt) hint: set for rows of whole 64-byte units (F % 16 == 0)
  int32_t reverse_runs = 0;  // each XCD walks its run of panels backwards (set after a substantial materialisation pre-pass)
  // fused linear epilogue (hg_aggr_linear_f32): Y[N, F_out] = (aggregated rows) * Wlin^T
  const float *Wlin = nullptr;  // Wlin [F_out, F] in MFMA fragment order (launch_linear_pack), or null
  int32_t F_out = 0;
  LinEpilogue epi;
};

// Y[rowmap ? rowmap[r] : r, :] = T[r, :] * Wlin^T for r < nrows; T is [nrows, F_in] row-major.
struct LinearArgs {
  const float *T;
  const float *Wlin;  // [F_out, F_in] in MFMA fragment order (launch_linear_pack)
  const int32_t *rowmap;
  float *Y;
  int64_t nrows;
  int32_t F_in, F_out;
  LinEpilogue epi;
};

// The hub pass (HubPass, hg_internal.h): persistent workgroups, register accumulators.
struct HubArgs {
  const int32_t *rec;       // round records (hg_fused.cpp, build_hub_pass)
  const HubRec *rec_tab;
  const int32_t *wg_first;  // [nwg + 1] rounds of each workgroup
  const int32_t *vslot0;    // [ng * R] first partial row of each virtual row, -1 = unused
  int32_t nwg, ng, cap, max_rec_words;
  const float *X, *Xe_mat, *degE, *W;
  float *partial;
  int32_t F;
  int32_t x_bytes, mat_bytes, nrows_x, nrows_mat;
  int32_t n_heavy = 0;  // hubs fed by stream flags (bits 24..27 of a slot's last entry)
  int32_t hslot0[kHubHeavy] = {};  // their first partial rows
  int32_t debug = 0;    // ablation bits, diagnostic build only
};

// stream_rows_kernel (RowStream, hg_internal.h)
struct StreamArgs {
  const int32_t *rec;
  const SRec *rec_tab;
  int32_t nrec, ng, cap, max_rec_words;
  const float *src;             // gathered table [nrows_src, F]
  int32_t src_bytes, nrows_src;
  const float *scaleA, *scaleB; // indexed by the record's sidx lists, or null
  float *dst, *partial;
  int32_t F, xcd_remap;
  int32_t nt_dst = 0;  // as GatherArgs::nt_dst
};

struct PushArgs {
  int64_t n_group;
  const int32_t *group_key, *group_row, *group_st, *group_ed;
  const int32_t *csrptr_t, *colind_t;
  const float *X, *degE, *degV, *W;
  float *Y;
  int32_t F;
};

hipError_t launch_gather(const GatherArgs &a, int nfix, int nfix_l1, const Fixup *fixups, bool vec4,
                         hipStream_t stream);
hipError_t launch_fused(const FusedArgs &a, bool vec4, hipStream_t stream);
hipError_t launch_hub_pass(const HubArgs &a, bool vec4, hipStream_t stream);
size_t hub_pass_lds_bytes(int32_t cap, int32_t row_floats, int32_t max_rec_words);
// Y[fx.row] = scale[fx.row] * (sum of the fixup's partial rows); first-level fixups first
hipError_t launch_fixups(const Fixup *fixups, int nfix, int nfix_l1, int32_t F, float *partial, float *Y,
                         const float *scaleA, const float *scaleB, const int32_t *scale_map, bool vec4,
                         hipStream_t stream, bool nt_dst = false);
bool stream_rows_ok(const StreamArgs &a, bool vec4);  // buffer-addressable table, 16-byte lanes
hipError_t launch_stream_rows(const StreamArgs &a, hipStream_t stream);
bool fused_linear_ok(const FusedArgs &a);  // can launch_fused run this call's linear epilogue?
hipError_t launch_linear(const LinearArgs &a, hipStream_t stream);
int wgrad_parts(int64_t nrows, int32_t Fa, int32_t Fb);
bool wgrad_shape_ok(int32_t Fa, int32_t Fb);  // <= 16 tiles of 16 x 16, or both widths multiples of 64 up to 512 (64 x 64 blocks)
hipError_t launch_wgrad(int64_t nrows, int32_t Fa, int32_t Fb, const float *A, const float *B, float *C,
                        float *partial, hipStream_t stream);
hipError_t launch_linear_pack(int32_t F_out, int32_t F_in, const float *Wlin, float *wfrag, hipStream_t stream);
hipError_t launch_linear_pack_split(int32_t F_out, int32_t F_in, const float *Wlin, void *wsplit, hipStream_t stream);
int fused_tile_row_floats(int F, bool vec4);
hipError_t read_stamps(unsigned long long *out, bool reset);
hipError_t launch_mfma_rate(int blocks, int iters, float *sink, unsigned long long *ticks, hipStream_t stream);
hipError_t launch_push(const PushArgs &a, hipStream_t stream);
hipError_t launch_bind_scales(int64_t nslots, const int32_t *eid_all, const float *degE, const float *W,
                              float *bsA, float *bsB, int64_t nrows, const int32_t *prow, const float *degV,
                              float *bsD, hipStream_t stream);
// *flag (device int32, preset to 1) is cleared if any of W[0..n) differs from 1.0f
hipError_t launch_all_ones(int64_t n, const float *W, int32_t *flag, hipStream_t stream);
hipError_t launch_gather_max(int32_t M, int32_t F, const int32_t *ptr, const int32_t *ind, const float *X,
                             const float *degE, const float *W, float *Xe, int32_t *record,
                             hipStream_t stream);
hipError_t launch_scatter_record(int32_t M, int32_t F, const float *T, const int32_t *record,
                                 const float *degV, float *Y, hipStream_t stream);

}  // namespace hg
