// Host-side schedule for the fused (LDS-staged) variant.  Pure C++.
//
// The vertex rows are cut into panels; for each panel the distinct hyperedges
// its vertices touch become "slots" of an LDS tile.  A slot is either
// recomputed inside the workgroup from its member rows of X (small hyperedges)
// or loaded from a materialised table Xe_mat (hyperedges with more than t_big
// members, and every hyperedge of a hub vertex), so the M x F hyperedge
// feature matrix never makes a round trip through HBM for the bulk of the graph.
#include <algorithm>

#include "hg_internal.h"

namespace hg {

void build_fused(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ind_t,
                 const int32_t *ptr_v, const int32_t *ind_v, const Opts &o, int32_t cap,
                 FusedSched &f) {
  f = FusedSched();
  f.cap = cap;
  f.rows_cap = cap;
  f.mem_cap = cap * 4;
  f.vslot_cap = cap * 2;
  f.t_big = std::max(1, std::min(o.t_big, f.mem_cap));
  // a single row must always fit into an empty panel
  f.vdeg_max = std::max(1, std::min(std::min(cap, f.vslot_cap), f.mem_cap / f.t_big));

  const int64_t nnz = ptr_t[M];
  std::vector<uint8_t> is_mat((size_t)M, 0), is_hub((size_t)N, 0);
  for (int32_t e = 0; e < M; e++)
    if (ptr_t[e + 1] - ptr_t[e] > f.t_big) is_mat[e] = 1;
  for (int32_t v = 0; v < N; v++)
    if (ptr_v[v + 1] - ptr_v[v] > f.vdeg_max) {
      is_hub[v] = 1;
      f.n_hub++;
      for (int32_t p = ptr_v[v]; p < ptr_v[v + 1]; p++) is_mat[ind_v[p]] = 1;
    }

  // compact CSR of the materialised hyperedges
  std::vector<int32_t> mat_id((size_t)M, -1);
  f.mat_ptr.push_back(0);
  for (int32_t e = 0; e < M; e++)
    if (is_mat[e]) {
      mat_id[e] = f.n_mat++;
      f.mat_eid.push_back(e);
      f.mat_ind.insert(f.mat_ind.end(), ind_t + ptr_t[e], ind_t + ptr_t[e + 1]);
      f.mat_ptr.push_back((int32_t)f.mat_ind.size());
    }
  // compact CSR of the hub vertices over materialised rows
  f.hub_ptr.push_back(0);
  for (int32_t v = 0; v < N; v++)
    if (is_hub[v]) {
      f.hub_vid.push_back(v);
      for (int32_t p = ptr_v[v]; p < ptr_v[v + 1]; p++) f.hub_ind.push_back(mat_id[ind_v[p]]);
      f.hub_ptr.push_back((int32_t)f.hub_ind.size());
    }
  build_sched(f.n_mat, f.mat_ptr.data(), o, f.mat_sched);
  build_sched(f.n_hub, f.hub_ptr.data(), o, f.hub_sched);

  // fused panels over the non-hub vertices
  f.vslot.assign((size_t)nnz, 0);
  std::vector<int32_t> stamp((size_t)M, -1), slot_of((size_t)M, 0);
  int32_t start = 0;
  FPanel cur{};
  int32_t vs_cnt = 0;
  auto open_panel = [&](int32_t row0) {
    cur = FPanel{};
    cur.row0 = row0;
    cur.sbase = (int32_t)f.soff.size();
    cur.pm0 = (int32_t)f.pmem.size();
    cur.eid0 = (int32_t)f.slot_eid.size();
    vs_cnt = 0;
    f.soff.push_back(0);
  };
  auto close_panel = [&](int32_t end_row) {
    cur.nrows = end_row - cur.row0;
    if (cur.nrows > 0) {
      cur.vs0 = ptr_v[cur.row0];
      cur.nvs = ptr_v[end_row] - ptr_v[cur.row0];
      cur.npm = (int32_t)f.pmem.size() - cur.pm0;
      f.panels.push_back(cur);
    } else {
      f.soff.pop_back();  // the opening 0 of an empty panel
    }
  };
  open_panel(0);
  start = 0;
  for (int32_t v = 0; v < N; v++) {
    if (is_hub[v]) {
      close_panel(v);
      open_panel(v + 1);
      start = v + 1;
      continue;
    }
    const int32_t pid = (int32_t)f.panels.size();
    // what would this row add?
    int32_t new_slots = 0, new_mem = 0;
    for (int32_t p = ptr_v[v]; p < ptr_v[v + 1]; p++) {
      const int32_t e = ind_v[p];
      if (stamp[e] != pid) {
        // a hyperedge listed twice for v (duplicate incidence) is counted twice here; harmless
        new_slots++;
        new_mem += is_mat[e] ? 1 : (ptr_t[e + 1] - ptr_t[e]);
      }
    }
    const int32_t deg = ptr_v[v + 1] - ptr_v[v];
    const int32_t cur_mem = (int32_t)f.pmem.size() - cur.pm0;
    if (v > start && (v - start == f.rows_cap || cur.nslots + new_slots > f.cap ||
                      cur_mem + new_mem > f.mem_cap || vs_cnt + deg > f.vslot_cap)) {
      close_panel(v);
      open_panel(v);
      start = v;
    }
    const int32_t pid2 = (int32_t)f.panels.size();
    for (int32_t p = ptr_v[v]; p < ptr_v[v + 1]; p++) {
      const int32_t e = ind_v[p];
      if (stamp[e] != pid2) {
        stamp[e] = pid2;
        slot_of[e] = cur.nslots++;
        if (is_mat[e]) {
          f.pmem.push_back((int32_t)(0x80000000u | (uint32_t)mat_id[e]));
          f.slot_eid.push_back(-1);
        } else {
          f.pmem.insert(f.pmem.end(), ind_t + ptr_t[e], ind_t + ptr_t[e + 1]);
          f.slot_eid.push_back(e);
        }
        f.soff.push_back((int32_t)f.pmem.size() - cur.pm0);
      }
      f.vslot[(size_t)p] = (uint16_t)slot_of[e];
    }
    vs_cnt += deg;
  }
  close_panel(N);
  f.pmem_entries = (int64_t)f.pmem.size();
}

}  // namespace hg
