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

// ---- hub pass (register hubs) -----------------------------------------------
// The vertices with the most incident hyperedges ("register hubs") are not panel rows: one
// persistent 1024-thread workgroup per CU keeps a running sum for every one of them in VGPRs
// (R rows per lane group) while it streams a contiguous range of the hyperedges that contain a
// hub, in rounds of at most `cap` hyperedge slots: the round's hyperedge sums are computed once
// into an LDS tile (hop 1 exactly as in a vertex panel), then every lane group adds the tile rows
// its hubs belong to.  Each workgroup leaves one partial row per virtual row; fixups add them up.
// A hub with a large share of the incidences is cut into several virtual rows ("parts", its
// incidences dealt round-robin) so that no lane group becomes the critical path of a round.
constexpr int kHubRows = 12;   // virtual hub rows per lane group of the hub pass (accumulator registers)
constexpr int kHubChunk = 4;  // virtual rows a lane group walks together in hop 2 (kRows is a multiple of it)
constexpr int kHubHeavy = 4;  // heavy hubs fed by stream flags: four accumulators per lane (registers are what limits it)

struct HubRec {
  int64_t off;              // word offset of the round's record
  int32_t len;              // record length in words (multiple of 4)
  int32_t nslots, off_eid;  // slots of the round; where their hyperedge ids sit in the record
  int32_t pad;
};

struct HubPass {
  int32_t K = 0;    // register hubs
  // The first n_heavy of them (the heaviest, at most kHubHeavy) have no virtual rows: a flag bit in the last
  // stream entry of every hyperedge they belong to (bits 24..29) makes the lane group that finishes
  // that hyperedge sum add it to a register of its own; the lane groups' registers are added up
  // once, at the end of the launch.  No tile read, no list entry, perfectly balanced.
  int32_t n_heavy = 0;
  int32_t hslot0[kHubHeavy] = {};  // first partial row of each heavy hub (+ workgroup id)
  int32_t nv = 0;   // virtual rows in use (<= ng * R)
  int32_t bs = 1024, ng = 0, R = kHubRows;
  int32_t cap = 0, mem_cap = 0, pair_cap = 0;  // per round: slots, stream entries, (row, slot) pairs
  int32_t nwg = 0;  // persistent workgroups = partial rows per virtual row
  int32_t max_rec_words = 0, max_steps = 0;
  int64_t stream_entries = 0;  // row gathers of one hub pass (idle steps included)
  int64_t pairs = 0;           // hub incidences
  std::vector<int32_t> vid;     // [K] vertex ids, by degree descending
  std::vector<int32_t> vslot0;  // [ng * R] first partial slot of each virtual row (+ workgroup id), -1 = unused
  std::vector<int32_t> rec;     // round records, back to back (see pack_hub_rounds)
  std::vector<HubRec> rec_tab;  // per round
  std::vector<int32_t> wg_first;  // [nwg + 1] rounds of each workgroup
  int32_t *d_vslot0 = nullptr, *d_rec = nullptr, *d_wg_first = nullptr;
  HubRec *d_rec_tab = nullptr;
};

// ---- streaming row gather (materialisation pre-pass) ----------------------------------------------
// dst[r] = scaleB * (scaleA * sum of src rows listed in CSR row r), as gather_rows_kernel computes it, but
// scheduled like a panel's hop 1: a workgroup's rows are spread over the lane groups longest first, every
// group walks the same number of steps with unpredicated buffer loads, and the lane group that finishes a
// row scales and stores it straight from registers -- no tile, no second hop.  Rows of more than
// `chunk` entries are cut into chunks that leave partial rows (summed by fixups, two levels above 32).
// Record: [0] steps [1] nslots [4] off_gbase [5] off_stream [6] off_dst [7] off_sidx, then gbase[ng],
// stream[steps * ng] (entry words as in a panel record; an empty row is one idle entry with the last
// flag), dst[nslots] (output row, or bit 31 | partial row), sidx[nslots] (index of the row's scale
// factors, -1 = none: chunks and empty rows).
constexpr int kLatShortMax = 8;     // latency schedule (hg_plan::sched_lat): longer rows are wave tasks
constexpr int kRowStreamChunk = 64;  // rows longer than this are cut into chunks (partial rows + fixups)

struct SRec {
  int64_t off;
  int32_t len, nslots, off_sidx, pad;
};
struct RowStream {
  int32_t ng = 0, cap = 0, max_rec_words = 0, max_steps = 0;
  int32_t nslots = 0;  // partial rows
  int64_t entries = 0;
  std::vector<int32_t> rec;
  std::vector<SRec> rec_tab;
  std::vector<Fixup> fixups;  // rows = output rows; first-level ones first
  int32_t n_fix_l1 = 0;
  int32_t *d_rec = nullptr;
  SRec *d_rec_tab = nullptr;
  Fixup *d_fixups = nullptr;
};

struct FusedSched {
  int32_t cap = 0;        // slots per panel (LDS tile rows)
  int32_t rows_cap = 0;   // rows per panel
  int32_t mem_cap = 0;    // staged member entries per panel
  int32_t vslot_cap = 0;  // staged (vertex, slot) incidences per panel
  int32_t t_big = 0;      // hyperedges longer than this are materialised
  int32_t vdeg_max = 0;   // vertices with more incident hyperedges are hubs or cut into pieces
  int32_t n_mat = 0;
  int32_t n_split = 0;    // vertices cut into pieces (panel rows that write partial sums)
  bool invalid = false;   // slot_chunk schedule that does not fit (a vertex's sub-slots exceed a panel): discard
  std::vector<FPanel> panels;
  std::vector<int32_t> soff, pmem, slot_eid;
  // panel rows: vertex id, or 0x80000000 | partial slot for a piece of a split vertex
  std::vector<int32_t> prow, pend;
  std::vector<uint16_t> pvs;        // panel-ordered incidences: local slot ids
  // materialised hyperedges (compact CSR over their members)
  std::vector<int32_t> mat_ptr, mat_ind, mat_eid;
  Sched mat_sched;
  RowStream mat_stream;  // the same rows for stream_rows_kernel (buffer-addressable tables, 16-byte lanes)
  HubPass hub;
  // partial rows of the fused path: [hub: sum over hubs of parts * nwg][pieces][first-level sums]
  int32_t n_part = 0;
  std::vector<Fixup> fixups;  // rows = vertex ids (hubs and split vertices); first-level ones first
  int32_t n_fix_l1 = 0;
  Fixup *d_fixups = nullptr;
  // device copies (the panel lists above stay on the host: the kernel reads the packed records)
  int32_t *d_prow = nullptr;
  int32_t *d_mat_ptr = nullptr, *d_mat_ind = nullptr, *d_mat_eid = nullptr;
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
  bool bound_W_is_one = false;  // every W[e] was exactly 1.0f when bound: multiplying by it is the identity
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
  bool fused_tile_auto = true;       // not set by the caller: a launch-bound graph may get smaller panels
  // > 0 (set by the plan for launch-bound graphs only): nothing is materialised; a hyperedge with more
  // members than this is cut into sub-slots of at most this many members, every one a slot of its own that
  // the hyperedge's vertices all add -- the longest dependent gather chain of a panel is one chunk, and
  // the aggregation stays one launch.  At least t_big, so hyperedges the default schedule recomputes
  // whole keep their summation order.
  int32_t slot_chunk = 0;
  int32_t fused_steps = 0;           // fused: stream entries per lane group and panel (0: 4 per slot on average)
  // hub pass (not in the C ABI; the host tests lower them to reach that code on small graphs)
  int64_t hub_min_nnz = 1 << 20;  // smaller graphs are launch-bound: no extra pass
  int32_t hub_min_deg = 256;      // below this a hub's partial rows cost more than pieces do
  int32_t hub_tile_bytes = 80 * 1024;  // LDS tile of a hub-pass round (of the CU's 160 KiB; the rest: two records, scales)
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
                    const Opts &o, int32_t cap, int32_t mem_cap, int64_t *n_mat, int64_t *n_big);
// row_floats: floats per LDS tile row (the kernels' TW); allow_hub: the hub pass may be used
// (buffer-addressable tables).
// rows_cap: rows per panel (0 = as many as slots).  The linear epilogue's schedule asks for fewer rows than slots:
// whole 16-row MFMA tiles, reached before the slots run out.
void build_fused(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ind_t,
                 const int32_t *ptr_v, const int32_t *ind_v, const Opts &o, int32_t cap,
                 int32_t mem_cap, int32_t ng, int32_t row_floats, bool allow_hub, FusedSched &f,
                 int32_t rows_cap = 0);
// rows [0, nrows) of the CSR (ptr, ind) as a RowStream; scale_index[r] (or r itself if null) names the
// row's scale factors; idle = rows of the gathered table
void build_row_stream(int32_t nrows, const int32_t *ptr, const int32_t *ind, const int32_t *scale_index,
                      int32_t ng, int32_t idle, RowStream &rs);
// the two-level fixup rule shared by build_sched and build_fused
void add_fixups(int32_t row, int32_t first, int32_t count, int32_t &nslots, std::vector<Fixup> &level1,
                std::vector<Fixup> &finals);

}  // namespace hg

struct hg_plan {
  int32_t N = 0, M = 0;
  int64_t nnz = 0;
  hg::Opts opts;
  std::vector<int32_t> ptr_t, ind_t;  // H_T CSR (hyperedge -> members), host copy
  std::vector<int32_t> ptr_v, ind_v;  // H CSR (vertex -> hyperedges), host
  int32_t *d_ptr_v = nullptr, *d_ind_v = nullptr;
  hg::Sched sched[2];  // [0]: H_T rows = hyperedges, [1]: H rows = vertices
  // launch-bound graphs (nnz <= 2^18) only: the same rows with every row of more than kLatShortMax entries as a wave
  // task (hg_plan_tune_f32 times it against the other kernels; the static rule never picks it)
  hg::Sched sched_lat[2];
  bool has_lat = false;
  std::map<int64_t, hg::FusedSched> fused;  // keyed by (slot, entry) capacity: depends on F
  std::map<int64_t, const hg::FusedSched *> fused_by_width;  // (F, vec4, lin) -> the schedule built for it
  std::mutex fused_mu;
  // pull variant on the streaming row gather: schedules per (hop, lane groups), built on first use
  std::map<int64_t, hg::RowStream> row_streams;
  std::mutex stream_mu;
  int32_t stream_nslots[2] = {0, 0};  // partial rows a RowStream of each hop needs (independent of the lane layout)
  std::map<int64_t, int32_t> auto_choice;  // what HG_VARIANT_AUTO resolved to, keyed by (F, vec4)
  std::map<int32_t, int32_t> hop_kernel;   // per F, set by hg_plan_tune_f32: k0 + 3 * k1, kernel of each pull hop (hg_tune_info)
  std::mutex auto_mu;
  double small_nnz_frac = 0.0;  // share of incidences in hyperedges of <= t_big members
  int64_t device_bytes = 0;
  int device = -1;
};
