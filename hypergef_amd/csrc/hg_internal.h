// Internal structures of libhgaggr (host side).  Not part of the C ABI.
#pragma once
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "hg_aggr.h"

namespace hg {

// One workgroup's share of the short rows: consecutive rows [row0, row0+nrows)
// whose index entries form the contiguous range [nnz0, nnz0+nnz_cnt).
struct Panel {
  int32_t row0, nrows, nnz0, nnz_cnt;
};

// One wavefront's share of a long row: entries [beg, end) of `row`.
// slot < 0: the wave owns the whole row and writes the scaled result itself.
// slot >= 0: the wave writes an unscaled partial sum to partial slot `slot`.
struct Task {
  int32_t row, beg, end, slot;
};

// A row split over several tasks: out[row] = scale * sum_{k<count} partial[first+k].
// pad > 0: first-level fixup of a two-level sum, partial[pad-1] = that sum, unscaled.
struct Fixup {
  int32_t row, first, count, pad;
};

// Schedule of one CSR matrix for the gather-rows kernel.
struct Sched {
  int32_t nrows = 0;
  int32_t max_len = 0;
  std::vector<Panel> panels;
  std::vector<Task> tasks;
  std::vector<Fixup> fixups;  // first-level fixups [0, n_fix_l1), then the final ones
  int32_t n_fix_l1 = 0;
  int32_t nslots = 0;
  // device copies
  Panel *d_panels = nullptr;
  Task *d_tasks = nullptr;
  Fixup *d_fixups = nullptr;
};

// ---- fused (LDS-staged) variant -------------------------------------------
// One workgroup's share of the non-hub vertices: the rows prow[r0 .. r0+nrows)
// (any vertex set; the plan clusters vertices that share hyperedges), their
// incidences as local slot ids pvs[v0 .. v0+nvs) with per-row end offsets
// pend[r0+i], plus the distinct hyperedges they touch ("slots").  Slot k
// of the panel owns entries [soff[sbase+k], soff[sbase+k+1]) of the panel's
// slice pmem[pm0 .. pm0+npm): member vertex ids of a recomputed hyperedge, or one
// entry with bit 31 set = row of the materialised table Xe_mat.
struct FPanel {
  int32_t r0, nrows, sbase, nslots, pm0, npm, eid0, v0, nvs, pad0, pad1, pad2;
};

struct FRec {
  int64_t off;  // word offset of the panel's record
  int32_t len;  // record length in words
  int32_t off_prow, pad;  // where the vertex ids sit inside the record
  int32_t nrows, nslots;
  int32_t slot_base, row_base;  // position of this panel's slots / rows in the bound scale arrays
};

struct FusedSched {
  int32_t cap = 0;        // slots per panel (LDS tile rows)
  int32_t rows_cap = 0;   // rows per panel
  int32_t mem_cap = 0;    // staged member entries per panel
  int32_t vslot_cap = 0;  // staged (vertex, slot) incidences per panel
  int32_t t_big = 0;      // hyperedges longer than this are materialised
  int32_t vdeg_max = 0;   // vertices with more incident hyperedges are "hubs"
  int32_t n_mat = 0, n_hub = 0;
  std::vector<FPanel> panels;
  std::vector<int32_t> soff, pmem, slot_eid;
  std::vector<int32_t> prow, pend;  // panel rows (vertex ids) and their local end offsets
  std::vector<uint16_t> pvs;        // panel-ordered incidences: local slot ids
  // materialised hyperedges (compact CSR over their members) and hub vertices
  // (compact CSR over their materialised hyperedges)
  std::vector<int32_t> mat_ptr, mat_ind, mat_eid, hub_ptr, hub_ind, hub_vid;
  Sched mat_sched, hub_sched;
  // device copies (the panel lists above stay on the host: the kernel reads the packed records)
  int32_t *d_prow = nullptr;
  int32_t *d_mat_ptr = nullptr, *d_mat_ind = nullptr, *d_mat_eid = nullptr;
  int32_t *d_hub_ptr = nullptr, *d_hub_ind = nullptr, *d_hub_vid = nullptr;
  int64_t pmem_entries = 0;
  // packed per-panel records (see pack_records in hg_fused.cpp)
  int32_t ng = 0;             // lane groups the hop-1 stream is packed for
  std::vector<int32_t> rec;   // all records, back to back
  std::vector<FRec> rec_tab;  // per panel: offset and length in words
  int32_t max_rec_words = 0;
  int32_t max_steps = 0;       // longest hop-1 stream of any panel
  int64_t stream_entries = 0;  // steps * ng summed over panels (incl. idle steps)
  int32_t *d_rec = nullptr;
  FRec *d_rec_tab = nullptr;
  // scales pre-gathered into panel order (hg_plan_bind_scales): per slot degE[e], W[e]
  // (1 for materialised slots), per panel row degV[v]
  std::vector<int32_t> eid_all;  // hyperedge id of every slot, record order
  int32_t *d_eid_all = nullptr;
  float *d_bsA = nullptr, *d_bsB = nullptr, *d_bsD = nullptr;
  const float *bound_degE = nullptr, *bound_W = nullptr, *bound_degV = nullptr;
};

struct Opts {
  int32_t short_max = 32;
  int32_t split_len = 512;
  int32_t panel_rows = 128;
  bool panel_rows_auto = true;  // not set by the caller: small schedules use smaller panels
  int32_t panel_nnz = 1024;
  int32_t flags = 0;
  int32_t t_big = 8;           // fused: recompute hyperedges of at most this many members
  int32_t fused_tile_bytes = 16384;  // fused: LDS tile budget -> hyperedge slots per panel
};

void set_error(const std::string &msg);

// Host algorithms (hg_schedule.cpp); all validate and return hg_status.
int balance_schedule(int32_t nrow, int32_t ngs, const int32_t *csrptr,
                     int64_t *n_key, int64_t *n_group, int32_t *key,
                     int32_t *row, int32_t *st, int32_t *ed);
int validate_csr(int32_t nrows, int32_t ncols, const int32_t *ptr,
                 const int32_t *ind);
void transpose_csr(int32_t nrows, int32_t ncols, const int32_t *ptr,
                   const int32_t *ind, std::vector<int32_t> &t_ptr,
                   std::vector<int32_t> &t_ind);
void build_sched(int32_t nrows, const int32_t *ptr, const Opts &o, Sched &s);
void classify_fused(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ptr_v, const int32_t *ind_v,
                    const Opts &o, int32_t cap, int32_t mem_cap, int64_t *n_mat, int64_t *n_hub);
void build_fused(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ind_t,
                 const int32_t *ptr_v, const int32_t *ind_v, const Opts &o, int32_t cap,
                 int32_t mem_cap, int32_t ng, FusedSched &f);

}  // namespace hg

struct hg_plan {
  int32_t N = 0, M = 0;
  int64_t nnz = 0;
  hg::Opts opts;
  std::vector<int32_t> ptr_t, ind_t;  // H_T CSR (hyperedge -> members), host copy
  std::vector<int32_t> ptr_v, ind_v;  // H CSR (vertex -> hyperedges), host
  int32_t *d_ptr_v = nullptr, *d_ind_v = nullptr;
  hg::Sched sched[2];  // [0]: H_T rows = hyperedges, [1]: H rows = vertices
  std::map<int64_t, hg::FusedSched> fused;  // keyed by (slot, entry) capacity: depends on F
  std::mutex fused_mu;
  std::map<int64_t, int32_t> auto_choice;  // what HG_VARIANT_AUTO resolved to, keyed by (F, vec4)
  std::mutex auto_mu;
  double small_nnz_frac = 0.0;  // share of incidences in hyperedges of <= t_big members
  int64_t device_bytes = 0;
  int device = -1;
};
