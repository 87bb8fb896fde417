// C ABI of libhgaggr.so (declared in include/hg_aggr.h): plan lifetime and the
// launch functions.  Launch functions only enqueue on the caller's stream: no
// allocation, no synchronisation, so they can be captured into a hipGraph.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>

#include "hg_kernels.h"

namespace {

int hip_fail(const char *what, hipError_t e) {
  hg::set_error(std::string(what) + ": " + hipGetErrorString(e));
  return HG_ERR_HIP;
}

#define HG_HIP(call)                                  \
  do {                                                \
    hipError_t e_ = (call);                           \
    if (e_ != hipSuccess) return hip_fail(#call, e_); \
  } while (0)

size_t round256(size_t x) { return (x + 255) & ~(size_t)255; }

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// rows of an output table are whole, aligned 64-byte units: where the streaming (nt) store hint pays (hg_kernels.hip, HG_Y_NT)
bool rows_whole_64(const void *base, int32_t F) { return F % 16 == 0 && (reinterpret_cast<uintptr_t>(base) & 63) == 0; }

int resolve_opts(const hg_plan_opts *in, hg::Opts &o) {
  if (in) {
    if (in->short_max > 0) o.short_max = in->short_max;
    if (in->split_len > 0) o.split_len = in->split_len;
    if (in->panel_rows > 0) {
      o.panel_rows = in->panel_rows;
      o.panel_rows_auto = false;
    }
    if (in->panel_nnz > 0) o.panel_nnz = in->panel_nnz;
    o.flags = in->flags;
    if (in->t_big > 0) o.t_big = in->t_big;
    if (in->fused_tile_bytes > 0) {
      o.fused_tile_bytes = in->fused_tile_bytes;
      o.fused_tile_auto = false;
    }
    if (in->fused_steps > 0) o.fused_steps = in->fused_steps;
  }
  if (o.short_max > o.panel_nnz || o.split_len < o.short_max || o.panel_rows > 4096 ||
      o.panel_nnz > 16384 || o.fused_tile_bytes > 131072) {
    hg::set_error("hg_plan_opts: need short_max <= panel_nnz <= 16384, split_len >= short_max, panel_rows <= 4096, fused_tile_bytes <= 131072");
    return HG_ERR_INVALID;
  }
  // what gather_rows_kernel stages per workgroup (launch_gather_t) must fit the CU's 160 KiB of LDS
  if ((size_t)(4 * o.panel_rows + 1 + o.panel_nnz) * sizeof(int32_t) > (size_t)160 * 1024) {
    hg::set_error("hg_plan_opts: panel_rows / panel_nnz need more than 160 KiB of LDS per workgroup");
    return HG_ERR_INVALID;
  }
  return HG_OK;
}

template <typename T>
int upload(const std::vector<T> &h, T **d, int64_t &bytes) {
  *d = nullptr;
  if (h.empty()) return HG_OK;
  const size_t n = h.size() * sizeof(T);
  hipError_t e = hipMalloc(reinterpret_cast<void **>(d), n);
  if (e != hipSuccess) {
    hg::set_error(std::string("hipMalloc(plan): ") + hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? HG_ERR_NOMEM : HG_ERR_HIP;
  }
  HG_HIP(hipMemcpy(*d, h.data(), n, hipMemcpyHostToDevice));
  bytes += (int64_t)n;
  return HG_OK;
}

int sched_upload(hg::Sched &s, int64_t &bytes) {
  int rc;
  if ((rc = upload(s.panels, &s.d_panels, bytes)) != HG_OK) return rc;
  if ((rc = upload(s.tasks, &s.d_tasks, bytes)) != HG_OK) return rc;
  return upload(s.fixups, &s.d_fixups, bytes);
}

void sched_free(hg::Sched &s) {
  if (s.d_panels) (void)hipFree(s.d_panels);
  if (s.d_tasks) (void)hipFree(s.d_tasks);
  if (s.d_fixups) (void)hipFree(s.d_fixups);
  s.d_panels = nullptr;
  s.d_tasks = nullptr;
  s.d_fixups = nullptr;
}

int plan_upload(hg_plan *p) {
  HG_HIP(hipGetDevice(&p->device));
  int rc;
  if ((rc = upload(p->ptr_v, &p->d_ptr_v, p->device_bytes)) != HG_OK) return rc;
  if ((rc = upload(p->ind_v, &p->d_ind_v, p->device_bytes)) != HG_OK) return rc;
  for (int h = 0; h < 2; h++)
    if ((rc = sched_upload(p->sched[h], p->device_bytes)) != HG_OK) return rc;
  if (p->has_lat)
    for (int h = 0; h < 2; h++)
      if ((rc = sched_upload(p->sched_lat[h], p->device_bytes)) != HG_OK) return rc;
  return HG_OK;
}

int fused_upload(hg::FusedSched &f, int64_t &bytes) {
  int rc;
#define UP(v, d) if ((rc = upload(f.v, &f.d, bytes)) != HG_OK) return rc
  UP(prow, d_prow);
  UP(mat_ptr, d_mat_ptr);
  UP(mat_ind, d_mat_ind);
  UP(mat_eid, d_mat_eid);
  UP(rec, d_rec);
  UP(rec_tab, d_rec_tab);
  UP(eid_all, d_eid_all);
  UP(fixups, d_fixups);
  UP(hub.vslot0, hub.d_vslot0);
  UP(hub.rec, hub.d_rec);
  UP(hub.rec_tab, hub.d_rec_tab);
  UP(hub.wg_first, hub.d_wg_first);
  UP(mat_stream.rec, mat_stream.d_rec);
  UP(mat_stream.rec_tab, mat_stream.d_rec_tab);
  UP(mat_stream.fixups, mat_stream.d_fixups);
#undef UP
  return sched_upload(f.mat_sched, bytes);
}

void fused_free(hg::FusedSched &f) {
  void *ptrs[] = {f.d_prow, f.d_mat_ptr, f.d_mat_ind, f.d_mat_eid, f.d_rec, f.d_rec_tab, f.d_eid_all, f.d_bsA,
                  f.d_bsB, f.d_bsD, f.d_fixups, f.hub.d_vslot0, f.hub.d_rec, f.hub.d_rec_tab, f.hub.d_wg_first, f.mat_stream.d_rec, f.mat_stream.d_rec_tab,
                  f.mat_stream.d_fixups};
  for (void *q : ptrs)
    if (q) (void)hipFree(q);
  sched_free(f.mat_sched);
}

// Capacities of a fused panel for feature width F: `cap` hyperedge slots (rows of the
// LDS tile, whole multiples of 16), four stream entries per slot on average.
// The fused chain (panels, streaming row gather, fixups) moves rows as 16-byte lanes whatever their width and
// base alignment when every table is addressable through a buffer descriptor (range-checked per dword, so
// the lane holding a row's last columns may load past the row's end): class-count widths (F = 3, 6, 7, 67)
// then run four floats per lane instead of one.  Mirrors `fast` in launch_fused_t.  Measured on the cora x1024
// batch (profiles/r02_experiments.md): F = 9 0.120 -> 0.076 ms, 33 0.340 -> 0.277, 67 0.766 -> 0.498; rows of up
// to 7 floats are faster one dword per lane (F = 7 0.072 vs 0.078, F = 3 0.047 vs 0.054) and stay there.
bool wide_rows_ok(const hg_plan *p, int32_t F) {
  return (F % 4 == 0 || F > 8) && p->N < (1 << 24) && p->M < (1 << 24) && F < (1 << 22) && (int64_t)p->N * F * 4 < ((int64_t)1 << 31) &&
         (int64_t)p->M * F * 4 < ((int64_t)1 << 31);
}
bool plan_vec4(const hg_plan *p, int32_t F) { return F % 4 == 0 || wide_rows_ok(p, F); }

void fused_caps(const hg_plan *p, int32_t F, bool vec4, int32_t &cap, int32_t &mem_cap, int32_t tile_bytes = 0) {
  const int row_bytes = hg::fused_tile_row_floats(F, vec4) * 4;
  const int c = std::max(16, std::min(256, (tile_bytes > 0 ? tile_bytes : p->opts.fused_tile_bytes) / row_bytes));
  cap = c / 16 * 16;
  mem_cap = cap * 4;
  if (p->opts.fused_steps > 0) {  // whole batches of the kernel's row loads: steps x lane groups
    const int32_t ng = 256 / (hg::fused_tile_row_floats(F, vec4) / (vec4 ? 4 : 1));
    mem_cap = std::max(p->opts.t_big, std::min(mem_cap, p->opts.fused_steps * ng));
  }
}

// A small hypergraph (nnz <= 2^18) is launch-bound: what one aggregation costs is the number of dependent
// launches and the longest dependent chain inside a workgroup, not bytes.  Model fitted on single cora /
// citeseer / pubmed-shape hypergraphs at F = 32 (profiles/r02_experiments.md, device us per aggregation):
// 5 us per launch (panels, + materialisation pre-pass, + fixup pass), 0.08 us per step of the longest hop-1
// stream, 0.4 us per row a lane group sums in hop 2.
double small_graph_cost(const hg::FusedSched &f) {
  int32_t max_rows = 0;
  for (const hg::FPanel &pn : f.panels) max_rows = std::max(max_rows, pn.nrows);
  const int launches = 1 + (f.n_mat > 0 ? 1 : 0) + (f.fixups.empty() ? 0 : 1);
  return 5.0 * launches + 0.08 * f.max_steps + 0.4 * ((max_rows + f.ng - 1) / std::max(1, f.ng));
}

// The F-dependent part of the plan, built on first use (guarded by the plan's mutex).
// The linear epilogue (hg_aggr_linear_f32) multiplies a panel's finished rows in 16-row MFMA tiles, at most four rows
// per lane group: a panel of 24 rows pays for 32.  Its schedule therefore caps the ROWS of a panel at 4 per lane group
// (F = 128: 32, F = 64: 64 -- whole tiles) and gives the panel half as many slots again, so that the rows run out before
// the slots do (pubmed-shape, F = 128: 32 slots hold 24 rows on average; 48 hold the 32).  lin_caps() returns false
// where the default schedule is already that shape (F = 32: 128 rows of 128 slots) or the caller fixed the tile.
bool lin_caps(const hg_plan *p, int32_t F, bool vec4, int32_t &cap, int32_t &mem_cap, int32_t &rows_cap) {
  // F = 128 only.  At F = 64 the default panels hold 53 of their 64 rows already, and every larger tile lost there (cora
  // x1024 64 -> 64: 0.463 ms default, 0.469-0.475 with 80 slots, 0.485-0.492 with 96).  At F = 32 a cap of 112 rows (seven
  // tiles) lets the operand rows fit the slot tile's 16 KB -- 19 600 instead of 21 648 bytes of LDS, eight resident
  // workgroups instead of seven -- and changed nothing: 32 -> 32 0.209-0.213 ms either way, 32 -> 64 -1 % (9 % more panels
  // eat the eighth workgroup; profiles/r04_experiments.md).
  if (!vec4 || F != 128 || !p->opts.fused_tile_auto || p->opts.fused_steps > 0 || p->nnz <= (1 << 18)) return false;
  int pct = 150;
#ifdef HG_TUNING
  if (const char *e = getenv("HG_LIN_SLOTS_PCT")) pct = atoi(e);  // diagnostic build: 0 = the default schedule
#endif
  if (pct <= 0) return false;
  fused_caps(p, F, vec4, cap, mem_cap);
  const int32_t ng = 256 / (hg::fused_tile_row_floats(F, vec4) / 4);
  rows_cap = std::min(cap, 4 * ng);
#ifdef HG_TUNING
  if (const char *e = getenv("HG_LIN_ROWS_CAP")) rows_cap = std::max(16, std::min(rows_cap, atoi(e) / 8 * 8));
#endif
  cap = std::max(cap, (rows_cap * pct / 100 + 7) / 8 * 8);
  mem_cap = cap * 4;
#ifdef HG_TUNING
  if (const char *e = getenv("HG_LIN_STEPS")) mem_cap = std::max(p->opts.t_big, std::min(mem_cap, atoi(e) * ng));  // whole batches of row loads
#endif
  return true;
}

int get_fused(const hg_plan *cp, int32_t F, bool vec4, const hg::FusedSched **out, bool lin = false) {
  hg_plan *p = const_cast<hg_plan *>(cp);
  int32_t cap, mem_cap, rows_cap = 0;
  if (lin && !lin_caps(p, F, vec4, cap, mem_cap, rows_cap)) lin = false;
  if (!lin) fused_caps(p, F, vec4, cap, mem_cap);
  const int32_t row_floats = hg::fused_tile_row_floats(F, vec4);
  const int32_t ng = 256 / (row_floats / (vec4 ? 4 : 1));  // lane groups per workgroup
  // The hub pass reads X and the materialised table through buffer descriptors (row index below 2^24,
  // tables below 2 GiB) and exists for 16-byte lanes of at least 16 floats per row.
  // (rows of the materialised table are masked to 24 bits in the round records too, and n_mat <= M)
  const bool allow_hub = vec4 && F % 4 == 0 && F >= 16 && p->N < (1 << 24) && p->M < (1 << 24) && (int64_t)p->N * F * 4 < ((int64_t)1 << 31) &&
                         (int64_t)p->M * F * 4 < ((int64_t)1 << 31);
  const int64_t key = ((((int64_t)cap * 1000000 + mem_cap) * 1000 + ng) * 2 + (allow_hub ? 1 : 0)) * 1024 + rows_cap;
  std::lock_guard<std::mutex> lock(p->fused_mu);
  auto it = p->fused.find(key);
  if (it == p->fused.end()) {
    hg::FusedSched f;
    try {
      hg::build_fused(p->N, p->M, p->ptr_t.data(), p->ind_t.data(), p->ptr_v.data(), p->ind_v.data(),
                      p->opts, cap, mem_cap, ng, row_floats, allow_hub, f, rows_cap);
      // A small hypergraph is launch-bound (small_graph_cost).  Two things can shorten it: recomputing the
      // few longish hyperedges too instead of a materialisation launch (one citeseer-shape hypergraph
      // 15.8 -> 8.2 us at F = 32), and -- unless the caller fixed the tile -- smaller panels: more, shorter
      // workgroups on a chip that 25 panels leave nine tenths empty (one cora-shape hypergraph at F = 32:
      // 128 / 64 / 32-slot panels 5.4 / 4.3 / 3.8 us).  All candidates are built, the cheapest is kept.
      if (p->nnz <= (1 << 18)) {
        double best = small_graph_cost(f);
        const int32_t tiles[3] = {p->opts.fused_tile_bytes, p->opts.fused_tile_bytes / 2, p->opts.fused_tile_bytes / 4};
        for (int t = 0; t < (p->opts.fused_tile_auto ? 3 : 1); t++) {
          int32_t c2, m2;
          fused_caps(p, F, vec4, c2, m2, tiles[t]);
          if (t > 0 && c2 == cap) continue;
          // mode 0: the default rule (materialise above t_big); 1: recompute every hyperedge whole;
          // 2, 3: cut hyperedges above 8 / 16 members into sub-slots (Opts::slot_chunk)
          for (int mode = 0; mode < 4; mode++) {
            if (t == 0 && mode == 0) continue;  // that is f
            hg::Opts o = p->opts;
            if (mode == 1) {
              if (p->sched[0].max_len * 4 > m2 || p->sched[0].max_len <= o.t_big) continue;
              o.t_big = p->sched[0].max_len;
            } else if (mode >= 2) {
              o.slot_chunk = std::max(o.t_big, mode == 2 ? 8 : 16);
              if (p->sched[0].max_len <= o.slot_chunk || (mode == 3 && o.t_big >= 16)) continue;
            }
            hg::FusedSched alt;
            hg::build_fused(p->N, p->M, p->ptr_t.data(), p->ind_t.data(), p->ptr_v.data(), p->ind_v.data(), o, c2, m2,
                            ng, row_floats, allow_hub, alt);
            if (alt.invalid) continue;
            const double c = small_graph_cost(alt);
            if (c < best) {
              best = c;
              f = std::move(alt);
            }
          }
        }
        // A tiny dense hypergraph (zoo: 43 hyperedges of ~28 of the 101 vertices) needs nearly every hyperedge in
        // every panel, so more panels do not shorten a workgroup's hop-1 stream -- 1221 gathers over the 32 lane
        // groups of a 256-thread workgroup are 38 dependent steps.  A 1024-thread workgroup has four times the lane
        // groups: the whole graph in one or a few such panels, nothing materialised, one launch.
        const int32_t lanes = row_floats / (vec4 ? 4 : 1);
        if (p->opts.fused_tile_auto && vec4 && F % 4 == 0 && (lanes == 8 || lanes == 16) && p->nnz <= 16384 && p->N <= 4096 &&
            wide_rows_ok(p, F)) {
          for (int32_t tile : {16384, 32768, 65536}) {
            int32_t c2, m2;
            fused_caps(p, F, vec4, c2, m2, tile);
            for (int mode = 1; mode < 4; mode++) {
              hg::Opts o = p->opts;
              if (mode == 1) {
                if (p->sched[0].max_len * 4 > m2) continue;
                o.t_big = std::max(o.t_big, p->sched[0].max_len);
              } else {
                o.slot_chunk = std::max(o.t_big, mode == 2 ? 8 : 16);
                if (p->sched[0].max_len <= o.slot_chunk) continue;
              }
              hg::FusedSched alt;
              hg::build_fused(p->N, p->M, p->ptr_t.data(), p->ind_t.data(), p->ptr_v.data(), p->ind_v.data(), o, c2, m2,
                              4 * ng, row_floats, false, alt);
              if (alt.invalid || alt.n_mat > 0 || !alt.fixups.empty()) continue;
              const double c = small_graph_cost(alt) + 0.3;  // barriers over sixteen waves
              if (c < best) {
                best = c;
                f = std::move(alt);
              }
            }
          }
        }
      }
    } catch (const std::bad_alloc &) {
      hg::set_error("fused schedule: host allocation failed");
      return HG_ERR_NOMEM;
    }
    // the kernel's LDS carve-up trusts these bounds: check them before anything can launch
    for (const hg::FPanel &pn : f.panels) {
      if (pn.nrows <= 0 || pn.nrows > f.rows_cap || pn.nslots > f.cap || pn.npm > f.mem_cap ||
          pn.nvs > f.vslot_cap || pn.r0 < 0 || (size_t)(pn.r0 + pn.nrows) > f.prow.size()) {
        hg::set_error("fused schedule: internal error, panel exceeds its LDS budget");
        return HG_ERR_INVALID;
      }
    }
    // tile (+ 4 pad floats per row in the linear epilogue) | record | scale staging: launch_fused_t's carve-up
    const size_t lds_need = std::max((size_t)f.cap * (row_floats + 4) * 4 + (size_t)f.max_rec_words * 4 +
                                         (size_t)(2 * f.cap + f.rows_cap) * 4 + 16,
                                     f.hub.K > 0 ? hg::hub_pass_lds_bytes(f.hub.cap, row_floats, f.hub.max_rec_words) : 0);
    if (lds_need > (size_t)160 * 1024) {
      hg::set_error("fused schedule: fused_tile_bytes = " + std::to_string(p->opts.fused_tile_bytes) + " needs " +
                    std::to_string(lds_need) + " bytes of LDS per workgroup at this feature width (limit 163840)");
      return HG_ERR_INVALID;
    }
    int rc = (p->opts.flags & HG_PLAN_HOST_ONLY) ? HG_OK : fused_upload(f, p->device_bytes);
    if (rc != HG_OK) {
      fused_free(f);
      return rc;
    }
    it = p->fused.emplace(key, std::move(f)).first;
  }
  p->fused_by_width[((int64_t)F * 2 + (vec4 ? 1 : 0)) * 2 + (lin ? 1 : 0)] = &it->second;
  *out = &it->second;
  return HG_OK;
}

// workspace carve-up.  Pull: [Xe: M*F][partials hop 0][partials hop 1].  Fused (the same buffer):
// [Xe_mat: n_mat*F][partials of the materialisation pre-pass][partial rows of hubs and pieces].
struct Carve {
  size_t xe, part[2], total;
};
Carve carve(const hg_plan *p, int32_t F) {
  Carve c;
  c.xe = 0;
  size_t off = round256((size_t)p->M * F * sizeof(float));
  for (int h = 0; h < 2; h++) {
    c.part[h] = off;
    off += round256((size_t)std::max(std::max(p->sched[h].nslots, p->sched_lat[h].nslots), p->stream_nslots[h]) * F * sizeof(float) + 16);  // + 16: see fused_carve
  }
  c.total = off;
  return c;
}
struct FusedCarve {
  size_t mat_part, part, total;
};
FusedCarve fused_carve(const hg::FusedSched &f, int32_t F) {
  FusedCarve c;
  c.mat_part = round256((size_t)f.n_mat * F * sizeof(float));
  c.part = c.mat_part + round256((size_t)std::max(f.mat_sched.nslots, f.mat_stream.nslots) * F * sizeof(float));
  c.total = c.part + round256((size_t)f.n_part * F * sizeof(float) + 16);  // + 16: a lane may load past the last partial row's end
  return c;
}
// The pull layout always fits a pull call; a fused schedule that exists for this width (built by
// hg_plan_prepare, hg_plan_auto_variant or an earlier call) may need more for its partial rows.
size_t workspace_need(const hg_plan *cp, int32_t F, bool pull_only = false) {
  hg_plan *p = const_cast<hg_plan *>(cp);
  size_t need = carve(p, F).total;
  if (pull_only) return need;
  std::lock_guard<std::mutex> lock(p->fused_mu);
  for (const auto &kv : p->fused_by_width)
    if (kv.first / 4 == F) need = std::max(need, fused_carve(*kv.second, F).total);
  return need;
}

// What HG_VARIANT_AUTO resolves to.  Measured over the 13 dataset shapes, single graphs and
// batches (profiles/r01_variant_choice.md): the fused panels win or tie wherever hub vertices
// are few -- by 3x on small graphs, where one launch instead of two is what counts, and by
// 0-35 % on large batches even when every hyperedge lands in four panels or 60 % of them are
// materialised -- and lose only where hubs drag nearly every hyperedge into the materialised
// table (yelp, the power-law config): the fused path is then the pull path plus overhead.
int pick_variant_uncached(const hg_plan *plan, int32_t F, bool vec4, int32_t *variant,
                          const hg::FusedSched **f);
int get_row_stream(const hg_plan *cp, int hop, int32_t ng, const hg::RowStream **out);

int pick_variant(const hg_plan *plan, int32_t F, bool vec4, int32_t *variant,
                 const hg::FusedSched **f) {
  // decided once per feature width: the classification below walks the whole graph on the host
  hg_plan *mp = const_cast<hg_plan *>(plan);
  const int64_t key = (int64_t)F * 2 + (vec4 ? 1 : 0);
  {
    std::lock_guard<std::mutex> lock(mp->auto_mu);
    auto it = mp->auto_choice.find(key);
    if (it != mp->auto_choice.end()) {
      *variant = it->second;
      return *variant == HG_VARIANT_FUSED ? get_fused(plan, F, vec4, f) : HG_OK;
    }
  }
  int rc = pick_variant_uncached(plan, F, vec4, variant, f);
  if (rc == HG_OK) {
    std::lock_guard<std::mutex> lock(mp->auto_mu);
    mp->auto_choice[key] = *variant;
  }
  return rc;
}

int pick_variant_uncached(const hg_plan *plan, int32_t F, bool vec4, int32_t *variant,
                          const hg::FusedSched **f) {
  *variant = HG_VARIANT_PULL;
  const bool small = plan->nnz <= (1 << 18);  // launch-bound: work per launch hardly matters
  if (!small && plan->small_nnz_frac < 0.2) return HG_OK;  // not worth building the schedule
  int32_t cap, mem_cap;
  fused_caps(plan, F, vec4, cap, mem_cap);
  if (small) {
    // a vertex too big for a panel means partial rows and a fixup launch: a third dependent launch,
    // which alone costs more than the pull path on a launch-bound graph
    int64_t n_mat = 0, n_big = 0;
    hg::classify_fused(plan->N, plan->M, plan->ptr_t.data(), plan->ptr_v.data(), plan->ind_v.data(), plan->opts,
                       cap, mem_cap, &n_mat, &n_big);
    if (n_big > 0) return HG_OK;
  }
  int rc = get_fused(plan, F, vec4, f);
  if (rc != HG_OK) return rc;
  // Row gathers of the fused path (panels + hub pass + materialisation) against the pull path's
  // 2 nnz: measured (profiles/r01_variant_choice.md, r02) the panels win up to about 5 nnz stream
  // entries -- one launch, no Xe round trip, re-gathered rows mostly L2 hits.
  const hg::FusedSched &fs = **f;
  const int64_t mat_nnz = fs.mat_ptr.empty() ? 0 : fs.mat_ptr.back();
  // small graphs: two dependent launches either way once the fused schedule needs a materialisation or fixup
  // pass, and then the pull kernels are the lighter pair (single pubmed / coauthor_cora / Mushroom / 20news /
  // house-committees shapes: pull 5-25 % ahead; profiles/r02_variant_choice.md)
  if (small && (fs.n_mat > 0 || !fs.fixups.empty())) return HG_OK;
  const bool work_ok = small || fs.pmem_entries + fs.hub.stream_entries + mat_nnz <= 5 * plan->nnz;
  // vertices cut into pieces pay for partial rows and a fixup pass: where they are more than 1/16 of the
  // vertices (yelp-shape: average degree 58, every second vertex above a panel's 64 hyperedges) the
  // pull path wins by 15-40 % (profiles/r02_variant_choice.md)
  const bool pieces_ok = small || (int64_t)fs.n_split * 16 <= plan->N;
  if (work_ok && pieces_ok) *variant = HG_VARIANT_FUSED;
  return HG_OK;
}

int run_sched(const hg_plan *p, const hg::Sched &s, int32_t F, const int32_t *ptr,
              const int32_t *ind, const float *src, const float *scaleA, const float *scaleB,
              const int32_t *scale_map, const int32_t *dst_map, float *dst, float *partial,
              hipStream_t stream, bool nt_dst = false) {
  hg::GatherArgs a;
  a.nt_dst = nt_dst ? 1 : 0;
  a.scale_map = scale_map;
  a.dst_map = dst_map;
  a.ptr = ptr;
  a.ind = ind;
  a.src = src;
  a.dst = dst;
  a.scaleA = scaleA;
  a.scaleB = scaleB;
  a.partial = partial;
  a.panels = s.d_panels;
  a.tasks = s.d_tasks;
  a.npanels = (int32_t)s.panels.size();
  a.ntasks = (int32_t)s.tasks.size();
  a.n_task_blocks = (a.ntasks + 3) / 4;
  a.F = F;
  a.panel_rows = p->opts.panel_rows;
  a.panel_nnz = p->opts.panel_nnz;
  a.xcd_remap = (p->opts.flags & HG_PLAN_NO_XCD_REMAP) ? 0 : 1;
  const bool vec4 = (F % 4 == 0) && aligned16(src) && aligned16(dst) && aligned16(partial);
  hipError_t e = hg::launch_gather(a, (int)s.fixups.size(), s.n_fix_l1, s.d_fixups, vec4, stream);
  if (e != hipSuccess) return hip_fail("gather_rows launch", e);
  return HG_OK;
}

// The pull hops on the streaming row gather where it applies (16-byte lanes, buffer-addressable table):
// the schedule for this hop and lane layout, built and uploaded on first use.
int get_row_stream(const hg_plan *cp, int hop, int32_t ng, const hg::RowStream **out) {
  hg_plan *p = const_cast<hg_plan *>(cp);
  std::lock_guard<std::mutex> lock(p->stream_mu);
  const int64_t key = (int64_t)hop * 1024 + ng;
  auto it = p->row_streams.find(key);
  if (it == p->row_streams.end()) {
    hg::RowStream rs;
    try {
      if (hop == 0) hg::build_row_stream(p->M, p->ptr_t.data(), p->ind_t.data(), nullptr, ng, p->N, rs);
      else hg::build_row_stream(p->N, p->ptr_v.data(), p->ind_v.data(), nullptr, ng, p->M, rs);
    } catch (const std::bad_alloc &) {
      hg::set_error("row stream schedule: host allocation failed");
      return HG_ERR_NOMEM;
    }
    if (rs.nslots > p->stream_nslots[hop]) {
      hg::set_error("row stream schedule: internal error, more partial rows than the workspace layout reserves");
      return HG_ERR_INVALID;
    }
    int rc;
    if ((rc = upload(rs.rec, &rs.d_rec, p->device_bytes)) != HG_OK) return rc;
    if ((rc = upload(rs.rec_tab, &rs.d_rec_tab, p->device_bytes)) != HG_OK) return rc;
    if ((rc = upload(rs.fixups, &rs.d_fixups, p->device_bytes)) != HG_OK) return rc;
    it = p->row_streams.emplace(key, std::move(rs)).first;
  }
  *out = &it->second;
  return HG_OK;
}

int run_hop(const hg_plan *p, int hop, int32_t F, const int32_t *ptr, const int32_t *ind,
            const float *src, const float *scaleA, const float *scaleB, float *dst,
            float *partial, hipStream_t stream) {
  // 16-byte lanes over rows of any width above 8 floats and any 4-byte alignment (wide_rows_ok): the stream
  // kernel loads through a range-checked descriptor and stores only the columns that exist
  const bool lanes16 = F % 4 == 0 || F > 8;
  const int64_t nsrc = hop == 0 ? p->N : p->M, sb = nsrc * F * 4;
  // Streaming (nt) stores for the hop's output: always for hop 2 (rows of Y); for hop 1 when Xe [M, F] is larger than
  // about three quarters of the 256 MiB Infinity Cache -- a smaller table is read straight back from it by hop 2 and plain stores keep it
  // there (same-box A/B, profiles/r03_experiments.md: 348-695 MB tables -3..-6 %, 2-143 MB tables +9..+16 % with nt)
  // -- and only for rows of whole 64-byte units (F % 16 == 0): others lose with the hint (hg_kernels.hip, HG_Y_NT)
  const bool nt_out = rows_whole_64(dst, F) && (hop == 1 || (int64_t)p->M * F * 4 >= ((int64_t)192 << 20));
  int kind = 0;  // hg_plan_tune_f32's choice for this hop and width: 0 streaming, 1 panels + tasks, 2 latency schedule
  {
    hg_plan *mp = const_cast<hg_plan *>(p);
    std::lock_guard<std::mutex> lock(mp->auto_mu);
    auto it = mp->hop_kernel.find(F);
    if (it != mp->hop_kernel.end()) kind = hop == 0 ? it->second % 3 : (it->second / 3) % 3;
  }
  if (kind == 0 && lanes16 && F < (1 << 22) && nsrc < (1 << 24) && sb > 0 && sb < ((int64_t)1 << 31) && !(p->opts.flags & HG_PLAN_NO_ROW_STREAM)) {
    const int32_t ng = 256 / (hg::fused_tile_row_floats(F, true) / 4);
    const hg::RowStream *rs = nullptr;
    int rc = get_row_stream(p, hop, ng, &rs);
    if (rc != HG_OK) return rc;
    hg::StreamArgs sa;
    sa.rec = rs->d_rec;
    sa.rec_tab = rs->d_rec_tab;
    sa.nrec = (int32_t)rs->rec_tab.size();
    sa.ng = rs->ng;
    sa.cap = rs->cap;
    sa.max_rec_words = rs->max_rec_words;
    sa.src = src;
    sa.src_bytes = (int32_t)sb;
    sa.nrows_src = (int32_t)nsrc;
    sa.scaleA = scaleA;
    sa.scaleB = scaleB;
    sa.dst = dst;
    sa.partial = partial;
    sa.F = F;
    sa.xcd_remap = (p->opts.flags & HG_PLAN_NO_XCD_REMAP) ? 0 : 1;
    sa.nt_dst = nt_out;
    if (sa.nrec == 0) return HG_OK;
    if (hg::stream_rows_ok(sa, true)) {
      hipError_t e = hg::launch_stream_rows(sa, stream);
      if (e == hipSuccess)
        e = hg::launch_fixups(rs->d_fixups, (int)rs->fixups.size(), rs->n_fix_l1, F, partial, dst, scaleA, scaleB,
                              nullptr, true, stream, nt_out);
      if (e != hipSuccess) return hip_fail("stream_rows launch", e);
      return HG_OK;
    }
  }
  return run_sched(p, (kind == 2 && p->has_lat) ? p->sched_lat[hop] : p->sched[hop], F, ptr, ind, src, scaleA, scaleB,
                   nullptr, nullptr, dst, partial, stream, nt_out);
}

// pull_only: the call runs the pull layout whatever schedules exist (a forced HG_VARIANT_PULL, a single hop): a
// caller-owned workspace that was large enough before a fused schedule was built stays valid for such calls.
enum { kSizeAny = 0, kSizePull = 1, kSizeLater = 2 };  // kSizeLater: the caller checks once it knows which layout runs
int check_call(const hg_plan *plan, int32_t F, const void *workspace, size_t workspace_bytes, int size_mode = kSizeAny) {
  if (!plan) {
    hg::set_error("null plan");
    return HG_ERR_INVALID;
  }
  if (plan->opts.flags & HG_PLAN_HOST_ONLY) {
    hg::set_error("plan was built with HG_PLAN_HOST_ONLY and holds no device schedule");
    return HG_ERR_INVALID;
  }
  if (F <= 0) {
    hg::set_error("feature width must be positive");
    return HG_ERR_INVALID;
  }
  if ((int64_t)std::max(plan->N, plan->M) * F >= ((int64_t)1 << 40)) {
    hg::set_error("feature matrix too large");
    return HG_ERR_INVALID;
  }
  const size_t need = size_mode == kSizeLater ? 0 : workspace_need(plan, F, size_mode == kSizePull);
  if (need > 0 && (!workspace || workspace_bytes < need)) {
    hg::set_error("workspace too small: need " + std::to_string(need) + " bytes, got " +
                  std::to_string(workspace_bytes));
    return HG_ERR_WORKSPACE;
  }
  if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 255)) {
    hg::set_error("workspace must be 256-byte aligned");
    return HG_ERR_INVALID;
  }
  return HG_OK;
}

int plan_build(hg_plan **out, int32_t N, int32_t M, const int32_t *csrptr_t,
               const int32_t *colind_t, const hg_plan_opts *opts) {
  if (!out) {
    hg::set_error("hg_plan_create: null output pointer");
    return HG_ERR_INVALID;
  }
  *out = nullptr;
  hg::Opts o;
  int rc = resolve_opts(opts, o);
  if (rc != HG_OK) return rc;
  if ((rc = hg::validate_csr(M, N, csrptr_t, colind_t)) != HG_OK) return rc;
  hg_plan *p = new (std::nothrow) hg_plan();
  if (!p) {
    hg::set_error("hg_plan_create: host allocation failed");
    return HG_ERR_NOMEM;
  }
  try {
    p->N = N;
    p->M = M;
    p->nnz = csrptr_t[M];
    p->opts = o;
    p->ptr_t.assign(csrptr_t, csrptr_t + M + 1);
    p->ind_t.assign(colind_t, colind_t + p->nnz);
    hg::transpose_csr(M, N, csrptr_t, colind_t, p->ptr_v, p->ind_v);
    hg::build_sched(M, csrptr_t, o, p->sched[0]);
    hg::build_sched(N, p->ptr_v.data(), o, p->sched[1]);
    if (p->nnz <= (1 << 18) && o.short_max > hg::kLatShortMax) {
      hg::Opts ol = o;
      ol.short_max = hg::kLatShortMax;
      hg::build_sched(M, csrptr_t, ol, p->sched_lat[0]);
      hg::build_sched(N, p->ptr_v.data(), ol, p->sched_lat[1]);
      p->has_lat = true;
    }
    int64_t small = 0;
    for (int32_t e = 0; e < M; e++) {
      const int32_t len = csrptr_t[e + 1] - csrptr_t[e];
      if (len <= o.t_big) small += len;
    }
    p->small_nnz_frac = p->nnz > 0 ? (double)small / (double)p->nnz : 0.0;
    for (int h = 0; h < 2; h++) {  // partial rows of the streaming row gather: chunks of rows beyond kRowStreamChunk,
      const std::vector<int32_t> &ptr = h ? p->ptr_v : p->ptr_t;  // plus the first-level sums of rows in more than 32 chunks
      int64_t ns = 0;
      for (size_t r = 0; r + 1 < ptr.size(); r++) {
        const int64_t len = ptr[r + 1] - ptr[r];
        if (len > hg::kRowStreamChunk) {
          const int64_t k = (len + hg::kRowStreamChunk - 1) / hg::kRowStreamChunk;
          ns += k;
          if (k > 32) {
            int64_t fan = 1;
            while (fan * fan < k) fan++;
            ns += (k + fan - 1) / fan;
          }
        }
      }
      p->stream_nslots[h] = (int32_t)std::min<int64_t>(ns, 0x7fffffff);
    }
  } catch (const std::bad_alloc &) {
    delete p;
    hg::set_error("hg_plan_create: host allocation failed");
    return HG_ERR_NOMEM;
  }
  if (!(o.flags & HG_PLAN_HOST_ONLY)) {
    rc = plan_upload(p);
    if (rc != HG_OK) {
      hg_plan_destroy(p);
      return rc;
    }
  }
  *out = p;
  return HG_OK;
}

}  // namespace

extern "C" {

int hg_plan_create_host(hg_plan **out, int32_t N, int32_t M, const int32_t *csrptr_t_host,
                        const int32_t *colind_t_host, const hg_plan_opts *opts) {
  return plan_build(out, N, M, csrptr_t_host, colind_t_host, opts);
}

int hg_plan_create_device(hg_plan **out, int32_t N, int32_t M, int64_t nnz,
                          const int32_t *csrptr_t_dev, const int32_t *colind_t_dev,
                          const hg_plan_opts *opts, hg_stream_t stream) {
  if (!out || N < 0 || M < 0 || nnz < 0 || !csrptr_t_dev || (nnz > 0 && !colind_t_dev)) {
    hg::set_error("hg_plan_create_device: bad argument");
    return HG_ERR_INVALID;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::vector<int32_t> ptr, ind;
  try {
    ptr.resize((size_t)M + 1);
    ind.resize((size_t)nnz);
  } catch (const std::bad_alloc &) {
    hg::set_error("hg_plan_create_device: host allocation failed");
    return HG_ERR_NOMEM;
  }
  HG_HIP(hipMemcpyAsync(ptr.data(), csrptr_t_dev, ptr.size() * sizeof(int32_t),
                        hipMemcpyDeviceToHost, s));
  if (nnz > 0)
    HG_HIP(hipMemcpyAsync(ind.data(), colind_t_dev, ind.size() * sizeof(int32_t),
                          hipMemcpyDeviceToHost, s));
  HG_HIP(hipStreamSynchronize(s));
  if (ptr[M] != nnz) {
    hg::set_error("hg_plan_create_device: csrptr_t[M] != nnz");
    return HG_ERR_INVALID;
  }
  return plan_build(out, N, M, ptr.data(), ind.data(), opts);
}

void hg_plan_destroy(hg_plan *p) {
  if (!p) return;
  if (p->d_ptr_v) (void)hipFree(p->d_ptr_v);
  if (p->d_ind_v) (void)hipFree(p->d_ind_v);
  for (int h = 0; h < 2; h++) sched_free(p->sched[h]);
  for (int h = 0; h < 2; h++) sched_free(p->sched_lat[h]);
  for (auto &kv : p->fused) fused_free(kv.second);
  for (auto &kv : p->row_streams) {
    void *ptrs[] = {kv.second.d_rec, kv.second.d_rec_tab, kv.second.d_fixups};
    for (void *q : ptrs)
      if (q) (void)hipFree(q);
  }
  delete p;
}

int hg_plan_get_info(const hg_plan *p, hg_plan_info *info) {
  if (!p || !info) {
    hg::set_error("hg_plan_get_info: null argument");
    return HG_ERR_INVALID;
  }
  std::memset(info, 0, sizeof(*info));
  info->N = p->N;
  info->M = p->M;
  info->nnz = p->nnz;
  info->short_max = p->opts.short_max;
  info->split_len = p->opts.split_len;
  info->panel_rows = p->opts.panel_rows;
  info->panel_nnz = p->opts.panel_nnz;
  info->flags = p->opts.flags;
  for (int h = 0; h < 2; h++) {
    info->panels[h] = (int32_t)p->sched[h].panels.size();
    info->tasks[h] = (int32_t)p->sched[h].tasks.size();
    info->partials[h] = p->sched[h].nslots;
    info->fixups[h] = (int32_t)p->sched[h].fixups.size();
    info->max_len[h] = p->sched[h].max_len;
  }
  info->device_bytes = p->device_bytes;
  return HG_OK;
}

int hg_plan_get_vertex_csr(const hg_plan *p, int32_t *ptr_v_host, int32_t *ind_v_host) {
  if (!p || !ptr_v_host || (p->nnz > 0 && !ind_v_host)) {
    hg::set_error("hg_plan_get_vertex_csr: null argument");
    return HG_ERR_INVALID;
  }
  std::memcpy(ptr_v_host, p->ptr_v.data(), p->ptr_v.size() * sizeof(int32_t));
  if (p->nnz > 0) std::memcpy(ind_v_host, p->ind_v.data(), p->ind_v.size() * sizeof(int32_t));
  return HG_OK;
}

int hg_plan_get_vertex_csr_device(const hg_plan *p, const int32_t **ptr_v_dev,
                                  const int32_t **ind_v_dev) {
  if (!p || !ptr_v_dev || !ind_v_dev || (p->opts.flags & HG_PLAN_HOST_ONLY)) {
    hg::set_error("hg_plan_get_vertex_csr_device: null argument or host-only plan");
    return HG_ERR_INVALID;
  }
  *ptr_v_dev = p->d_ptr_v;
  *ind_v_dev = p->d_ind_v;
  return HG_OK;
}

int hg_plan_get_schedule(const hg_plan *p, int32_t hop, int32_t *panels, int32_t *tasks,
                         int32_t *fixups) {
  if (!p || (hop != 0 && hop != 1)) {
    hg::set_error("hg_plan_get_schedule: bad argument");
    return HG_ERR_INVALID;
  }
  const hg::Sched &s = p->sched[hop];
  static_assert(sizeof(hg::Panel) == 16 && sizeof(hg::Task) == 16 && sizeof(hg::Fixup) == 16, "quad layout");
  if (panels && !s.panels.empty()) std::memcpy(panels, s.panels.data(), s.panels.size() * sizeof(hg::Panel));
  if (tasks && !s.tasks.empty()) std::memcpy(tasks, s.tasks.data(), s.tasks.size() * sizeof(hg::Task));
  if (fixups && !s.fixups.empty()) std::memcpy(fixups, s.fixups.data(), s.fixups.size() * sizeof(hg::Fixup));
  return HG_OK;
}

// degE[e] / W[e] per slot and degV[v] per panel row of schedule f, in record order (buffers allocated on first use)
static int gather_bound_scales(hg::FusedSched *f, const float *degE, const float *degV, const float *W, hipStream_t stream) {
  const size_t ns = f->eid_all.size(), nr = f->prow.size();
  if (!f->d_bsA && ns > 0) {
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_bsA), ns * sizeof(float)));
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_bsB), ns * sizeof(float)));
  }
  if (!f->d_bsD && nr > 0) HG_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_bsD), nr * sizeof(float)));
  hipError_t e = hg::launch_bind_scales((int64_t)ns, f->d_eid_all, degE, W, f->d_bsA, f->d_bsB, (int64_t)nr,
                                        f->d_prow, degV, f->d_bsD, stream);
  if (e != hipSuccess) return hip_fail("bind_scales launch", e);
  return HG_OK;
}

int hg_plan_bind_scales(const hg_plan *cp, int32_t F, const float *degE, const float *degV,
                        const float *W, hg_stream_t stream) {
  if (!cp || F <= 0 || (cp->opts.flags & HG_PLAN_HOST_ONLY)) {
    hg::set_error("hg_plan_bind_scales: bad argument or host-only plan");
    return HG_ERR_INVALID;
  }
  const hg::FusedSched *cf = nullptr;
  int rc = get_fused(cp, F, plan_vec4(cp, F), &cf);
  if (rc != HG_OK) return rc;
  hg::FusedSched *f = const_cast<hg::FusedSched *>(cf);
  std::lock_guard<std::mutex> lock(const_cast<hg_plan *>(cp)->fused_mu);
  // the linear epilogue's own schedule of this width (if one was built) follows: it is re-bound on its next use
  for (const auto &kv : cp->fused_by_width)
    if (kv.first / 4 == F && (kv.first & 1) && kv.second != cf) {
      hg::FusedSched *fl = const_cast<hg::FusedSched *>(kv.second);
      fl->bound_degE = fl->bound_W = fl->bound_degV = nullptr;
      fl->bound_W_is_one = false;
    }
  if (!degE && !degV && !W) {  // unbind: later calls gather their scales themselves
    f->bound_degE = f->bound_W = f->bound_degV = nullptr;
    f->bound_W_is_one = false;
    return HG_OK;
  }
  if ((rc = gather_bound_scales(f, degE, degV, W, static_cast<hipStream_t>(stream))) != HG_OK) return rc;
  // W = ones is what the reference's models pass (model/ugsys/hgnn.py:12): x * 1.0f is x, so a bound all-ones W
  // is left out of the kernels.  One 4-byte read-back: this call synchronises `stream` when W is given.
  bool w_one = false;
  if (W && cp->M > 0) {
    int32_t *d_flag = nullptr, h_flag = 1;
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&d_flag), sizeof(int32_t)));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e2 = hipMemcpyAsync(d_flag, &h_flag, sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e2 == hipSuccess) e2 = hg::launch_all_ones(cp->M, W, d_flag, st);
    if (e2 == hipSuccess) e2 = hipMemcpyAsync(&h_flag, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
    (void)hipFree(d_flag);
    if (e2 != hipSuccess) return hip_fail("bind_scales: all-ones check", e2);
    w_one = h_flag != 0;
  }
  f->bound_degE = degE;
  f->bound_W = W;
  f->bound_degV = degV;
  f->bound_W_is_one = w_one;
  return HG_OK;
}

int hg_plan_auto_variant(const hg_plan *p, int32_t F) {
  if (!p || F <= 0) {
    hg::set_error("hg_plan_auto_variant: bad argument");
    return HG_ERR_INVALID;
  }
  int32_t variant = HG_VARIANT_PULL;
  const hg::FusedSched *f = nullptr;
  int rc = pick_variant(p, F, plan_vec4(p, F), &variant, &f);
  return rc != HG_OK ? rc : variant;
}

int hg_plan_prepare(const hg_plan *p, int32_t F, hg_fused_info *info) {
  if (!p || F <= 0) {
    hg::set_error("hg_plan_prepare: bad argument");
    return HG_ERR_INVALID;
  }
  const hg::FusedSched *f = nullptr;
  int rc = get_fused(p, F, plan_vec4(p, F), &f);
  if (rc != HG_OK) return rc;
  // the pull variant's schedules for this lane layout too: after hg_plan_prepare a call of any variant
  // allocates nothing
  if (!(p->opts.flags & (HG_PLAN_HOST_ONLY | HG_PLAN_NO_ROW_STREAM)) && (F % 4 == 0 || F > 8)) {
    const int32_t ng = 256 / (hg::fused_tile_row_floats(F, true) / 4);
    for (int hop = 0; hop < 2; hop++) {
      const int64_t nsrc = hop == 0 ? p->N : p->M;
      if (nsrc >= (1 << 24) || nsrc * F * 4 >= ((int64_t)1 << 31)) continue;
      const hg::RowStream *rs = nullptr;
      if ((rc = get_row_stream(p, hop, ng, &rs)) != HG_OK) return rc;
    }
  }
  if (info) {
    info->cap = f->cap;
    info->t_big = f->t_big;
    info->vdeg_max = f->vdeg_max;
    info->panels = (int32_t)f->panels.size();
    info->n_mat = f->n_mat;
    info->n_hub = f->hub.K;
    info->member_entries = f->pmem_entries;
    int64_t slots = 0;
    for (const auto &pn : f->panels) slots += pn.nslots;
    info->slots = slots;
    info->n_split = f->n_split;
    info->fixups = (int32_t)f->fixups.size();
    info->hub_rounds = (int32_t)f->hub.rec_tab.size();
    info->hub_workgroups = f->hub.K > 0 ? f->hub.nwg : 0;
    info->hub_entries = f->hub.stream_entries;
    info->hub_pairs = f->hub.pairs;
    info->partial_rows = f->n_part;
    info->record_words_max = f->max_rec_words;
    info->stream_steps_max = f->max_steps;
    info->lds_bytes = (int32_t)((size_t)f->cap * hg::fused_tile_row_floats(F, plan_vec4(p, F)) * 4 + (size_t)f->max_rec_words * 4 + 16);
    info->reserved = 0;
  }
  return HG_OK;
}

// Diagnostic only (not declared in hg_aggr.h): per-phase cycle counters of the
// fused kernel when HG_FUSED_DEBUG has bit 32 set.
__attribute__((visibility("default"))) int hg_debug_read_stamps(unsigned long long *out16, int reset) {
  return hg::read_stamps(out16, reset != 0) == hipSuccess ? HG_OK : HG_ERR_HIP;
}

// Diagnostic only (not declared in hg_aggr.h): sustained rate of v_mfma_f32_16x16x4_f32 from registers.  out3 = {seconds,
// TFLOP/s, shader-clock ticks (s_memtime) of one wave}; blocks x 4 waves, iters x 16 MFMAs per wave.
__attribute__((visibility("default"))) int hg_debug_mfma_rate(int32_t blocks, int32_t iters, double *out3) {
  float *sink = nullptr;
  unsigned long long *ticks = nullptr, h_ticks = 0;
  HG_HIP(hipMalloc(reinterpret_cast<void **>(&sink), 16));
  HG_HIP(hipMalloc(reinterpret_cast<void **>(&ticks), 8));
  hipEvent_t e0, e1;
  HG_HIP(hipEventCreate(&e0));
  HG_HIP(hipEventCreate(&e1));
  HG_HIP(hg::launch_mfma_rate(blocks, iters, sink, ticks, nullptr));  // warm-up
  HG_HIP(hipEventRecord(e0, nullptr));
  HG_HIP(hg::launch_mfma_rate(blocks, iters, sink, ticks, nullptr));
  HG_HIP(hipEventRecord(e1, nullptr));
  HG_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  HG_HIP(hipEventElapsedTime(&ms, e0, e1));
  HG_HIP(hipMemcpy(&h_ticks, ticks, 8, hipMemcpyDeviceToHost));
  (void)hipFree(sink);
  (void)hipFree(ticks);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const double flop = (double)blocks * 4.0 * iters * 16.0 * 2048.0;
  out3[0] = ms * 1e-3;
  out3[1] = flop / (ms * 1e-3) / 1e12;
  out3[2] = (double)h_ticks;
  return HG_OK;
}

// Diagnostic only (not declared in hg_aggr.h): shape of the fused schedule of width F -- lin != 0: the linear epilogue's
// own one -- out = {panels, rows, rows padded to 16-row tiles, member entries, slots, cap, rows_cap, n_mat, hop-1 steps}.
__attribute__((visibility("default"))) int hg_debug_fused_shape(const hg_plan *p, int32_t F, int32_t lin, int64_t *out9) {
  const hg::FusedSched *f = nullptr;
  int rc = get_fused(p, F, plan_vec4(p, F), &f, lin != 0);
  if (rc != HG_OK) return rc;
  int64_t rows = 0, padded = 0, slots = 0;
  for (const auto &pn : f->panels) {
    rows += pn.nrows;
    padded += (pn.nrows + 15) / 16 * 16;
    slots += pn.nslots;
  }
  const int64_t v[9] = {(int64_t)f->panels.size(), rows, padded, f->pmem_entries, slots, f->cap, f->rows_cap, f->n_mat,
                        f->stream_entries / std::max(1, f->ng) * 100000 + f->max_rec_words};  // [8]: hop-1 steps summed over the panels * 1e5 + record words
  for (int i = 0; i < 9; i++) out9[i] = v[i];
  return HG_OK;
}

size_t hg_plan_workspace_bytes(const hg_plan *p, int32_t F) {
  if (!p || F <= 0) return 0;
  // size for what HG_VARIANT_AUTO will run: resolving it builds the fused schedule if that is the choice
  int32_t variant = HG_VARIANT_PULL;
  const hg::FusedSched *f = nullptr;
  (void)pick_variant(p, F, plan_vec4(p, F), &variant, &f);
  return workspace_need(p, F);
}

int hg_gather_rows_f32(const hg_plan *plan, int32_t hop, int32_t F, const int32_t *csrptr_t,
                       const int32_t *colind_t, const float *src, const float *scaleA,
                       const float *scaleB, float *dst, void *workspace, size_t workspace_bytes,
                       hg_stream_t stream) {
  int rc = check_call(plan, F, workspace, workspace_bytes, kSizePull);
  if (rc != HG_OK) return rc;
  if ((hop != 0 && hop != 1) || !src || !dst || (hop == 0 && (!csrptr_t || (plan->nnz > 0 && !colind_t)))) {
    hg::set_error("hg_gather_rows_f32: bad argument");
    return HG_ERR_INVALID;
  }
  const Carve c = carve(plan, F);
  float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + c.part[hop]);
  const int32_t *ptr = hop == 0 ? csrptr_t : plan->d_ptr_v;
  const int32_t *ind = hop == 0 ? colind_t : plan->d_ind_v;
  return run_hop(plan, hop, F, ptr, ind, src, scaleA, scaleB, dst, partial,
                 static_cast<hipStream_t>(stream));
}

int hg_aggr_push_groups_f32(int32_t N, int32_t M, int32_t F, int64_t n_group,
                            const int32_t *group_key, const int32_t *group_row,
                            const int32_t *group_st, const int32_t *group_ed,
                            const int32_t *csrptr_t, const int32_t *colind_t, const float *X,
                            const float *degE, const float *degV, const float *W, float *Y,
                            hg_stream_t stream) {
  if (N < 0 || M < 0 || F <= 0 || n_group < 0 || !X || !Y || !colind_t) {
    hg::set_error("hg_aggr_push_groups_f32: bad argument");
    return HG_ERR_INVALID;
  }
  if (group_key ? (!group_row || !group_st || !group_ed) : (!csrptr_t || n_group != M)) {
    hg::set_error("hg_aggr_push_groups_f32: need all four group arrays, or csrptr_t with n_group == M");
    return HG_ERR_INVALID;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  HG_HIP(hipMemsetAsync(Y, 0, (size_t)N * F * sizeof(float), s));
  hg::PushArgs a;
  a.n_group = n_group;
  a.group_key = group_key;
  a.group_row = group_row;
  a.group_st = group_st;
  a.group_ed = group_ed;
  a.csrptr_t = csrptr_t;
  a.colind_t = colind_t;
  a.X = X;
  a.degE = degE;
  a.degV = degV;
  a.W = W;
  a.Y = Y;
  a.F = F;
  hipError_t e = hg::launch_push(a, s);
  if (e != hipSuccess) return hip_fail("push_groups launch", e);
  return HG_OK;
}

int hg_gather_max_f32(int32_t M, int32_t F, const int32_t *csrptr_t, const int32_t *colind_t,
                      const float *X, const float *degE, const float *W, float *Xe, int32_t *record,
                      hg_stream_t stream) {
  if (M < 0 || F <= 0 || !csrptr_t || !X || !Xe || !record || (int64_t)M * F >= ((int64_t)1 << 39)) {
    hg::set_error("hg_gather_max_f32: bad argument");
    return HG_ERR_INVALID;
  }
  hipError_t e = hg::launch_gather_max(M, F, csrptr_t, colind_t, X, degE, W, Xe, record,
                                       static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("gather_max launch", e);
  return HG_OK;
}

int hg_scatter_record_f32(int32_t N, int32_t M, int32_t F, const float *T, const int32_t *record,
                          const float *degV, float *Y, hg_stream_t stream) {
  if (N < 0 || M < 0 || F <= 0 || !T || !record || !Y) {
    hg::set_error("hg_scatter_record_f32: bad argument");
    return HG_ERR_INVALID;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  HG_HIP(hipMemsetAsync(Y, 0, (size_t)N * F * sizeof(float), s));
  hipError_t e = hg::launch_scatter_record(M, F, T, record, degV, Y, s);
  if (e != hipSuccess) return hip_fail("scatter_record launch", e);
  return HG_OK;
}

// Shared body of hg_aggr_fused_f32 and hg_aggr_linear_f32.  With lin != nullptr the caller
// wants (aggregated rows) * Wlin^T in lin->Y: the fused panels do that in their epilogue when
// they can (lin->done = true); otherwise the aggregated rows go to Y as usual and the caller
// runs the standalone linear kernel over them.
struct LinReq {
  const float *Wlin;
  int32_t F_out;
  float *Y;
  bool done;
  hg::LinEpilogue epi;
};

static int aggr_impl(const hg_plan *plan, int32_t F, const int32_t *csrptr_t,
                     const int32_t *colind_t, const float *X, const float *degE,
                     const float *degV, const float *W, float *Y, void *workspace,
                     size_t workspace_bytes, int32_t variant, hg_stream_t stream, LinReq *lin) {
  if (variant == HG_VARIANT_PUSH_ATOMIC) {
    if (!plan) {
      hg::set_error("null plan");
      return HG_ERR_INVALID;
    }
    return hg_aggr_push_groups_f32(plan->N, plan->M, F, plan->M, nullptr, nullptr, nullptr, nullptr,
                                   csrptr_t, colind_t, X, degE, degV, W, Y, stream);
  }
  if (variant != HG_VARIANT_AUTO && variant != HG_VARIANT_PULL && variant != HG_VARIANT_FUSED) {
    hg::set_error("hg_aggr_fused_f32: unknown variant");
    return HG_ERR_UNSUPPORTED;
  }
  // the size is checked against the layout that actually runs, once the variant is resolved
  int rc = check_call(plan, F, workspace, workspace_bytes, kSizeLater);
  if (rc != HG_OK) return rc;
  if (!csrptr_t || (plan->nnz > 0 && !colind_t) || !X || !Y) {
    hg::set_error("hg_aggr_fused_f32: null array");
    return HG_ERR_INVALID;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Carve c = carve(plan, F);
  char *ws = static_cast<char *>(workspace);
  float *Xe = reinterpret_cast<float *>(ws + c.xe);
  const bool aligned = (F % 4 == 0) && aligned16(X) && aligned16(Y) && aligned16(Xe);
  const bool vec4 = aligned || wide_rows_ok(plan, F);  // the fused chain's lane layout; the pull kernels decide for themselves
  const hg::FusedSched *f = nullptr;
  if (variant == HG_VARIANT_AUTO) {
    if ((rc = pick_variant(plan, F, vec4, &variant, &f)) != HG_OK) return rc;
  }
  if (variant == HG_VARIANT_FUSED) {
    if (!f && (rc = get_fused(plan, F, vec4, &f)) != HG_OK) return rc;
    if (lin && aligned && f->fixups.empty()) {
      // the epilogue's own schedule (whole 16-row tiles per panel, get_fused); its scales follow the default
      // schedule's binding -- gathered here on first use, on this call's stream, which every later use is ordered behind
      // or waits for through the caller's binding event (plan.py, _bind_scales)
      const hg::FusedSched *fl = nullptr;
      if ((rc = get_fused(plan, F, vec4, &fl, true)) != HG_OK) return rc;
      if (fl != f && fl->fixups.empty()) {
        const bool dflt_bound = (degE || degV || W) && f->bound_degE == degE && f->bound_W == W && f->bound_degV == degV;
        if (dflt_bound && !(fl->bound_degE == degE && fl->bound_W == W && fl->bound_degV == degV)) {
          hg::FusedSched *m = const_cast<hg::FusedSched *>(fl);
          std::lock_guard<std::mutex> lock(const_cast<hg_plan *>(plan)->fused_mu);
          if ((rc = gather_bound_scales(m, degE, degV, W, s)) != HG_OK) return rc;
          // first use only (it also allocated): wait for the gather, so that a later call on ANY stream finds the arrays
          // complete -- the caller's binding event (plan.py) covers the default schedule's gather, not this one
          (void)hipStreamSynchronize(s);
          m->bound_degE = degE;
          m->bound_W = W;
          m->bound_degV = degV;
          m->bound_W_is_one = f->bound_W_is_one;
        }
        f = fl;
      }
    }
    const FusedCarve fc = fused_carve(*f, F);
    if (fc.total > 0 && (!workspace || fc.total > workspace_bytes)) {
      hg::set_error("workspace too small for the fused schedule: need " + std::to_string(fc.total) + " bytes, got " +
                    std::to_string(workspace_bytes) +
                    " (hg_plan_workspace_bytes covers it once hg_plan_prepare has built the schedule)");
      return HG_ERR_WORKSPACE;
    }
    float *partial = reinterpret_cast<float *>(ws + fc.part);
    // scales pre-gathered into panel order, if the caller bound exactly these arrays
    const bool bound = (degE || degV || W) && f->bound_degE == degE && f->bound_W == W && f->bound_degV == degV;
    if (bound && W && f->bound_W_is_one) W = nullptr;  // multiplying by exactly 1.0f is the identity: same bits, less work
    const int64_t xb = (int64_t)plan->N * F * 4, mb = (int64_t)f->n_mat * F * 4;
    const int32_t x_bytes = xb < ((int64_t)1 << 31) ? (int32_t)xb : 0;
    const int32_t mat_bytes = mb < ((int64_t)1 << 31) ? (int32_t)mb : 0;
    // (a) materialised hyperedges (more than t_big members): Xe_mat rows
    if (f->n_mat > 0) {
      hg::StreamArgs sa;
      sa.rec = f->mat_stream.d_rec;
      sa.rec_tab = f->mat_stream.d_rec_tab;
      sa.nrec = (int32_t)f->mat_stream.rec_tab.size();
      sa.ng = f->mat_stream.ng;
      sa.cap = f->mat_stream.cap;
      sa.max_rec_words = f->mat_stream.max_rec_words;
      sa.src = X;
      sa.src_bytes = x_bytes;
      sa.nrows_src = plan->N;
      sa.scaleA = degE;
      sa.scaleB = W;
      sa.dst = Xe;
      sa.partial = reinterpret_cast<float *>(ws + fc.mat_part);
      sa.F = F;
      sa.xcd_remap = (plan->opts.flags & HG_PLAN_NO_XCD_REMAP) ? 0 : 1;
      if (hg::stream_rows_ok(sa, vec4)) {  // the streaming row gather; else the general kernel
        hipError_t e = hg::launch_stream_rows(sa, s);
        if (e == hipSuccess)
          e = hg::launch_fixups(f->mat_stream.d_fixups, (int)f->mat_stream.fixups.size(), f->mat_stream.n_fix_l1, F,
                                sa.partial, Xe, degE, W, f->d_mat_eid, vec4, s);
        if (e != hipSuccess) return hip_fail("stream_rows launch", e);
      } else {
        rc = run_sched(plan, f->mat_sched, F, f->d_mat_ptr, f->d_mat_ind, X, degE, W, f->d_mat_eid,
                       nullptr, Xe, reinterpret_cast<float *>(ws + fc.mat_part), s);
        if (rc != HG_OK) return rc;
      }
    }
    // (b) register hubs: persistent workgroups stream the hyperedges, running sums in registers
    if (f->hub.K > 0) {
      hg::HubArgs h;
      h.rec = f->hub.d_rec;
      h.rec_tab = f->hub.d_rec_tab;
      h.wg_first = f->hub.d_wg_first;
      h.vslot0 = f->hub.d_vslot0;
      h.nwg = f->hub.nwg;
      h.ng = f->hub.ng;
      h.cap = f->hub.cap;
      h.max_rec_words = f->hub.max_rec_words;
      h.X = X;
      h.Xe_mat = f->n_mat > 0 ? Xe : nullptr;
      h.degE = degE;
      h.W = W;
      h.partial = partial;
      h.F = F;
      h.x_bytes = x_bytes;
      h.mat_bytes = mat_bytes;
      h.nrows_x = plan->N;
      h.nrows_mat = f->n_mat;
      h.n_heavy = f->hub.n_heavy;
      for (int q = 0; q < hg::kHubHeavy; q++) h.hslot0[q] = f->hub.hslot0[q];
      hipError_t e = hg::launch_hub_pass(h, vec4, s);
      if (e != hipSuccess) return hip_fail("hub_pass launch", e);
    }
    // (c) everything else: vertex panels with the hyperedge sums staged in LDS
    hg::FusedArgs a;
    a.npanels = (int32_t)f->panels.size();
    a.X = X;
    a.Xe_mat = f->n_mat > 0 ? Xe : nullptr;  // null = no materialised rows: kernels skip that path
    a.degE = degE;
    a.W = W;
    a.degV = degV;
    a.Y = Y;
    a.partial = partial;
    a.F = F;
    a.cap = f->cap;
    a.rows_cap = f->rows_cap;
    a.xcd_remap = (plan->opts.flags & HG_PLAN_NO_XCD_REMAP) ? 0 : 1;
    a.rec = f->d_rec;
    a.rec_tab = f->d_rec_tab;
    a.eid_all = f->d_eid_all;
    a.max_rec_words = f->max_rec_words;
    a.ng = f->ng;
    a.x_bytes = x_bytes;
    a.mat_bytes = mat_bytes;
    a.nrows_x = plan->N;
    a.nrows_mat = f->n_mat;
    a.y_nt = rows_whole_64(Y, F);
    // a pre-pass that read at least a quarter of the incidences' member rows: see fused_packed_kernel's run order
    a.reverse_runs = f->n_mat > 0 && !f->mat_ptr.empty() && (int64_t)f->mat_ptr.back() * 4 >= plan->nnz;
    a.bsA = bound ? f->d_bsA : nullptr;
    a.bsB = bound ? f->d_bsB : nullptr;
    a.bsD = bound ? f->d_bsD : nullptr;
    // rows produced outside the panels (hubs, split vertices) get no epilogue: two-step then
    if (lin && f->fixups.empty() && aligned) {
      a.Wlin = lin->Wlin;
      a.F_out = lin->F_out;
      a.epi = lin->epi;
      if (hg::fused_linear_ok(a)) {
        a.Y = lin->Y;
        lin->done = true;
      } else {
        a.Wlin = nullptr;
      }
    }
    hipError_t e = hg::launch_fused(a, vec4, s);
    if (e != hipSuccess) return hip_fail("fused_panel launch", e);
    // (d) hubs and split vertices: Y[v] = degV[v] * (sum of the vertex's partial rows), fixed order
    if (!f->fixups.empty()) {
      e = hg::launch_fixups(f->d_fixups, (int)f->fixups.size(), f->n_fix_l1, F, partial, Y, degV, nullptr, nullptr, vec4, s, rows_whole_64(Y, F));
      if (e != hipSuccess) return hip_fail("fixup launch", e);
    }
    return HG_OK;
  }
  if (c.total > 0 && (!workspace || workspace_bytes < c.total)) {
    hg::set_error("workspace too small: need " + std::to_string(c.total) + " bytes, got " + std::to_string(workspace_bytes));
    return HG_ERR_WORKSPACE;
  }
  // hop 1: Xe[e] = ((sum_{u in e} X[u]) * degE[e]) * W[e]
  rc = run_hop(plan, 0, F, csrptr_t, colind_t, X, degE, W, Xe,
               reinterpret_cast<float *>(ws + c.part[0]), s);
  if (rc != HG_OK) return rc;
  // hop 2: Y[v] = (sum_{e contains v} Xe[e]) * degV[v]
  return run_hop(plan, 1, F, plan->d_ptr_v, plan->d_ind_v, Xe, degV, nullptr, Y,
                 reinterpret_cast<float *>(ws + c.part[1]), s);
}

int hg_aggr_fused_f32(const hg_plan *plan, int32_t F, const int32_t *csrptr_t,
                      const int32_t *colind_t, const float *X, const float *degE,
                      const float *degV, const float *W, float *Y, void *workspace,
                      size_t workspace_bytes, int32_t variant, hg_stream_t stream) {
  return aggr_impl(plan, F, csrptr_t, colind_t, X, degE, degV, W, Y, workspace, workspace_bytes, variant,
                   stream, nullptr);
}

int hg_plan_tune_f32(const hg_plan *plan, int32_t F, const int32_t *csrptr_t, const int32_t *colind_t,
                     const float *X, const float *degE, const float *degV, const float *W, float *Y,
                     void *workspace, size_t workspace_bytes, int32_t iters, hg_stream_t stream,
                     hg_tune_info *info) {
  int rc = check_call(plan, F, workspace, workspace_bytes);
  if (rc != HG_OK) return rc;
  if (iters <= 0) iters = 20;
  hg_plan *mp = const_cast<hg_plan *>(plan);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char *ws = static_cast<const char *>(workspace);
  const bool aligned = (F % 4 == 0) && aligned16(X) && aligned16(Y) && aligned16(ws);
  const int64_t key = (int64_t)F * 2 + ((aligned || wide_rows_ok(plan, F)) ? 1 : 0);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    hg::set_error("hg_plan_tune_f32: hipEventCreate failed");
    return HG_ERR_HIP;
  }
  constexpr int NC = 10;  // fused + 3 x 3 kernels of the two pull hops
  float us[NC];
  for (int c = 0; c < NC; c++) us[c] = -1.f;
  auto set_choice = [&](int32_t variant, int32_t mask) {
    std::lock_guard<std::mutex> lock(mp->auto_mu);
    mp->auto_choice[key] = variant;
    mp->hop_kernel[F] = mask;
  };
  rc = HG_OK;
  for (int c = 0; c < NC && rc == HG_OK; c++) {
    const int32_t variant = c == 0 ? HG_VARIANT_FUSED : HG_VARIANT_PULL;
    if (c > 0 && !plan->has_lat && ((c - 1) % 3 == 2 || (c - 1) / 3 == 2)) continue;  // no latency schedule for this plan
    if (c == 0) {  // the fused schedule may not exist for this plan / width / workspace: skip it then
      const hg::FusedSched *f = nullptr;
      if (get_fused(plan, F, key & 1, &f) != HG_OK || fused_carve(*f, F).total > workspace_bytes) continue;
    }
    set_choice(variant, c == 0 ? 0 : c - 1);
    for (int i = 0; i < 3 && rc == HG_OK; i++)
      rc = aggr_impl(plan, F, csrptr_t, colind_t, X, degE, degV, W, Y, workspace, workspace_bytes, variant, stream, nullptr);
    if (rc != HG_OK) break;
    hipError_t he = hipEventRecord(e0, s);
    for (int i = 0; i < iters && rc == HG_OK; i++)
      rc = aggr_impl(plan, F, csrptr_t, colind_t, X, degE, degV, W, Y, workspace, workspace_bytes, variant, stream, nullptr);
    if (he == hipSuccess) he = hipEventRecord(e1, s);
    if (rc != HG_OK) break;
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    if (he != hipSuccess) {
      rc = hip_fail("hg_plan_tune_f32: event timing", he);
      break;
    }
    us[c] = ms * 1e3f / iters;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  int best = -1;
  for (int c = 0; c < NC; c++)
    if (us[c] >= 0.f && (best < 0 || us[c] < us[best])) best = c;
  if (rc != HG_OK || best < 0) {  // leave the static rule in place
    std::lock_guard<std::mutex> lock(mp->auto_mu);
    mp->auto_choice.erase(key);
    mp->hop_kernel.erase(F);
    if (rc == HG_OK) {
      hg::set_error("hg_plan_tune_f32: no candidate ran");
      rc = HG_ERR_INVALID;
    }
    return rc;
  }
  // among the pull candidates keep the fastest hop kernels even if fused won: a forced HG_VARIANT_PULL uses them
  int best_pull = 1;
  for (int c = 2; c < NC; c++)
    if (us[c] >= 0.f && us[c] < us[best_pull]) best_pull = c;
  set_choice(best == 0 ? HG_VARIANT_FUSED : HG_VARIANT_PULL, best_pull - 1);
  // Y holds the last candidate's result: run the winner once more so the caller gets what AUTO now computes
  rc = aggr_impl(plan, F, csrptr_t, colind_t, X, degE, degV, W, Y, workspace, workspace_bytes, HG_VARIANT_AUTO, stream, nullptr);
  if (info) {
    info->variant = best == 0 ? HG_VARIANT_FUSED : HG_VARIANT_PULL;
    info->pull_hop_kernels = best_pull - 1;
    for (int c = 0; c < NC; c++) info->us[c] = us[c];
    info->reserved = 0;
  }
  return rc;
}

int hg_linear_pack_f32(int32_t F_out, int32_t F_in, const float *Wlin, float *wfrag, hg_stream_t stream) {
  if (!Wlin || !wfrag || F_out <= 0 || (F_out % 16) || (F_in != 32 && F_in != 64 && F_in != 128)) {
    hg::set_error("hg_linear_pack_f32: need F_in in {32, 64, 128}, F_out a positive multiple of 16");
    return Wlin && wfrag ? HG_ERR_UNSUPPORTED : HG_ERR_INVALID;
  }
  hipError_t e = hg::launch_linear_pack(F_out, F_in, Wlin, wfrag, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("linear_pack launch", e);
  return HG_OK;
}

size_t hg_linear_pack_floats(int32_t F_out, int32_t F_in, int32_t flags) {
  if (F_out <= 0 || F_in <= 0) return 0;
  const size_t n = (size_t)F_out * F_in;
  return ((flags & HG_LIN_BF16X6) && F_in == 128) ? n + n * 3 / 2 : n;  // + three bf16 planes
}

int hg_linear_pack_ex_f32(int32_t F_out, int32_t F_in, const float *Wlin, float *wfrag, int32_t flags, hg_stream_t stream) {
  int rc = hg_linear_pack_f32(F_out, F_in, Wlin, wfrag, stream);
  if (rc != HG_OK || !(flags & HG_LIN_BF16X6) || F_in != 128) return rc;
  hipError_t e = hg::launch_linear_pack_split(F_out, F_in, Wlin, wfrag + (size_t)F_out * F_in, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("linear_pack_split launch", e);
  return HG_OK;
}

int hg_linear_rows_f32(int64_t nrows, int32_t F_in, int32_t F_out, const float *T, const float *wfrag, float *Y,
                       hg_stream_t stream) {
  if (nrows < 0 || !wfrag || (nrows > 0 && (!T || !Y))) {
    hg::set_error("hg_linear_rows_f32: bad argument");
    return HG_ERR_INVALID;
  }
  if (F_out <= 0 || (F_out % 16) || (F_in != 32 && F_in != 64 && F_in != 128)) {
    hg::set_error("hg_linear_rows_f32: need F_in in {32, 64, 128}, F_out a positive multiple of 16");
    return HG_ERR_UNSUPPORTED;
  }
  if (!aligned16(T) || !aligned16(wfrag) || !aligned16(Y)) {
    hg::set_error("hg_linear_rows_f32: arrays must be 16-byte aligned");
    return HG_ERR_INVALID;
  }
  hg::LinearArgs la;
  la.T = T;
  la.Wlin = wfrag;
  la.rowmap = nullptr;
  la.Y = Y;
  la.nrows = nrows;
  la.F_in = F_in;
  la.F_out = F_out;
  hipError_t e = hg::launch_linear(la, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("linear_rows launch", e);
  return HG_OK;
}

size_t hg_linear_wgrad_workspace_bytes(int64_t nrows, int32_t F_a, int32_t F_b) {
  if (nrows < 0 || !hg::wgrad_shape_ok(F_a, F_b)) return 0;
  return (size_t)(hg::wgrad_parts(nrows, F_a, F_b) + 32) * F_a * F_b * sizeof(float);  // + second-level partials
}

int hg_linear_wgrad_f32(int64_t nrows, int32_t F_a, int32_t F_b, const float *A, const float *B, float *C,
                        void *workspace, size_t workspace_bytes, hg_stream_t stream) {
  if (nrows < 0 || !C || (nrows > 0 && (!A || !B))) {
    hg::set_error("hg_linear_wgrad_f32: bad argument");
    return HG_ERR_INVALID;
  }
  if (!hg::wgrad_shape_ok(F_a, F_b)) {
    hg::set_error("hg_linear_wgrad_f32: need F_a, F_b multiples of 16 with F_a * F_b <= 4096, or multiples of 64 up to 512");
    return HG_ERR_UNSUPPORTED;
  }
  if (!workspace || workspace_bytes < hg_linear_wgrad_workspace_bytes(nrows, F_a, F_b)) {
    hg::set_error("hg_linear_wgrad_f32: workspace smaller than hg_linear_wgrad_workspace_bytes");
    return HG_ERR_WORKSPACE;
  }
  hipError_t e = hg::launch_wgrad(nrows, F_a, F_b, A, B, C, static_cast<float *>(workspace), static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("wgrad launch", e);
  return HG_OK;
}

// the base layout once the epilogue's own schedule exists too (it is what a fused call with a linear runs)
static size_t linear_base_bytes(const hg_plan *plan, int32_t F_in) {
  int32_t variant = HG_VARIANT_PULL;
  const hg::FusedSched *f = nullptr;
  const bool vec4 = plan_vec4(plan, F_in);
  if (pick_variant(plan, F_in, vec4, &variant, &f) == HG_OK && variant == HG_VARIANT_FUSED && f && f->fixups.empty())
    (void)get_fused(plan, F_in, vec4, &f, true);
  return workspace_need(plan, F_in);
}

size_t hg_aggr_linear_workspace_bytes(const hg_plan *plan, int32_t F_in) {
  if (!plan || F_in <= 0) return 0;
  return linear_base_bytes(plan, F_in) + round256((size_t)plan->N * F_in * sizeof(float));
}

static int aggr_linear_res(const hg_plan *plan, int32_t F_in, int32_t F_out, const int32_t *csrptr_t,
                           const int32_t *colind_t, const float *X, const float *degE, const float *degV,
                           const float *W, const float *wfrag, const float *R, float ca, float cb, const float *cb_dev,
                           int32_t relu, float *T_out, float *Y, void *workspace, size_t workspace_bytes, int32_t variant,
                           hg_stream_t stream) {
  if (!plan || !wfrag || !Y) {
    hg::set_error("hg_aggr_linear_f32: null argument");
    return HG_ERR_INVALID;
  }
  if (F_out <= 0 || (F_out % 16) || (F_in != 32 && F_in != 64 && F_in != 128)) {
    hg::set_error("hg_aggr_linear_f32: need F_in in {32, 64, 128}, F_out a positive multiple of 16");
    return HG_ERR_UNSUPPORTED;
  }
  if (variant == HG_VARIANT_PUSH_ATOMIC) {
    hg::set_error("hg_aggr_linear_f32: push-atomic variant not supported");
    return HG_ERR_UNSUPPORTED;
  }
  const size_t base = linear_base_bytes(plan, F_in);  // asked once: the query runs the AUTO rule under the plan's locks
  if (!workspace || workspace_bytes < base + round256((size_t)plan->N * F_in * sizeof(float))) {
    hg::set_error("hg_aggr_linear_f32: workspace smaller than hg_aggr_linear_workspace_bytes");
    return HG_ERR_WORKSPACE;
  }
  if (!aligned16(wfrag) || !aligned16(Y) || !aligned16(R) || !aligned16(T_out)) {
    hg::set_error("hg_aggr_linear_f32: wfrag, R, T_out and Y must be 16-byte aligned");
    return HG_ERR_INVALID;
  }
  float *T = reinterpret_cast<float *>(static_cast<char *>(workspace) + round256(base));
  LinReq lin{wfrag, F_out, Y, false, {}};
  lin.epi.R = R;
  lin.epi.ca = ca;
  lin.epi.cb = R ? cb : 0.f;
  lin.epi.cb_dev = R ? cb_dev : nullptr;
  lin.epi.relu = (relu & HG_LIN_RELU) ? 1 : 0;
  lin.epi.wsplit = ((relu & HG_LIN_BF16X6) && F_in == 128) ? wfrag + (size_t)F_out * F_in : nullptr;
  lin.epi.T_out = T_out;
  int rc = aggr_impl(plan, F_in, csrptr_t, colind_t, X, degE, degV, W, T, workspace, base, variant, stream, &lin);
  if (rc != HG_OK || lin.done) return rc;
  hg::LinearArgs la;
  la.T = T;
  la.Wlin = wfrag;
  la.rowmap = nullptr;
  la.Y = Y;
  la.nrows = plan->N;
  la.F_in = F_in;
  la.F_out = F_out;
  la.epi = lin.epi;
  hipError_t e = hg::launch_linear(la, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail("linear_rows launch", e);
  return HG_OK;
}

int hg_aggr_linear_res_f32(const hg_plan *plan, int32_t F_in, int32_t F_out, const int32_t *csrptr_t,
                           const int32_t *colind_t, const float *X, const float *degE, const float *degV,
                           const float *W, const float *wfrag, const float *R, float ca, float cb, int32_t relu,
                           float *T_out, float *Y, void *workspace, size_t workspace_bytes, int32_t variant,
                           hg_stream_t stream) {
  return aggr_linear_res(plan, F_in, F_out, csrptr_t, colind_t, X, degE, degV, W, wfrag, R, ca, cb, nullptr, relu, T_out, Y,
                         workspace, workspace_bytes, variant, stream);
}

int hg_aggr_linear_res_dev_f32(const hg_plan *plan, int32_t F_in, int32_t F_out, const int32_t *csrptr_t,
                               const int32_t *colind_t, const float *X, const float *degE, const float *degV,
                               const float *W, const float *wfrag, const float *R, float ca, const float *cb_dev,
                               int32_t relu, float *T_out, float *Y, void *workspace, size_t workspace_bytes,
                               int32_t variant, hg_stream_t stream) {
  if (!cb_dev) {
    hg::set_error("hg_aggr_linear_res_dev_f32: cb_dev is null (use hg_aggr_linear_res_f32 for a host scalar)");
    return HG_ERR_INVALID;
  }
  return aggr_linear_res(plan, F_in, F_out, csrptr_t, colind_t, X, degE, degV, W, wfrag, R, ca, 0.f, cb_dev, relu, T_out, Y,
                         workspace, workspace_bytes, variant, stream);
}

int hg_aggr_linear_f32(const hg_plan *plan, int32_t F_in, int32_t F_out, const int32_t *csrptr_t,
                       const int32_t *colind_t, const float *X, const float *degE, const float *degV,
                       const float *W, const float *wfrag, float *Y, void *workspace,
                       size_t workspace_bytes, int32_t variant, hg_stream_t stream) {
  return hg_aggr_linear_res_f32(plan, F_in, F_out, csrptr_t, colind_t, X, degE, degV, W, wfrag, nullptr, 1.f, 0.f,
                                0, nullptr, Y, workspace, workspace_bytes, variant, stream);
}

}  // extern "C"
