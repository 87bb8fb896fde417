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
                 int32_t mem_cap, FusedSched &f) {
  f = FusedSched();
  f.cap = cap;
  f.rows_cap = cap;
  f.mem_cap = mem_cap;
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

  // Row order: depth-first post-order over the bipartite vertex/hyperedge graph,
  // so that a panel (a run of this order) holds vertices that share hyperedges:
  // each hyperedge a panel touches costs that panel one slot and one pass over
  // its members, whatever the number of the panel's vertices that use it.
  std::vector<int32_t> order;
  order.reserve((size_t)N);
  {
    std::vector<uint8_t> vis((size_t)N, 0), seen((size_t)M, 0);
    struct Frame {
      int32_t v, ep, up;  // vertex, position in its hyperedge list, position in that hyperedge
    };
    std::vector<Frame> st;
    for (int32_t root = 0; root < N; root++) {
      if (vis[root] || is_hub[root]) continue;
      vis[root] = 1;
      st.push_back(Frame{root, ptr_v[root], -1});
      while (!st.empty()) {
        Frame &fr = st.back();
        bool descended = false;
        while (fr.ep < ptr_v[fr.v + 1] && !descended) {
          const int32_t e = ind_v[fr.ep];
          if (fr.up < 0) {  // first look at this hyperedge from this vertex
            if (seen[e] || is_mat[e]) {
              fr.ep++;
              continue;
            }
            seen[e] = 1;
            fr.up = ptr_t[e];
          }
          while (fr.up < ptr_t[e + 1]) {
            const int32_t u = ind_t[fr.up++];
            if (!vis[u] && !is_hub[u]) {
              vis[u] = 1;
              st.push_back(Frame{u, ptr_v[u], -1});  // invalidates fr
              descended = true;
              break;
            }
          }
          if (!descended) {
            fr.ep++;
            fr.up = -1;
          }
        }
        if (!descended) {
          order.push_back(st.back().v);
          st.pop_back();
        }
      }
    }
  }

  // fused panels over that order
  std::vector<int32_t> stamp((size_t)M, -1), slot_of((size_t)M, 0);
  FPanel cur{};
  auto open_panel = [&]() {
    cur = FPanel{};
    cur.r0 = (int32_t)f.prow.size();
    cur.sbase = (int32_t)f.soff.size();
    cur.pm0 = (int32_t)f.pmem.size();
    cur.eid0 = (int32_t)f.slot_eid.size();
    cur.v0 = (int32_t)f.pvs.size();
    f.soff.push_back(0);
  };
  auto close_panel = [&]() {
    cur.nrows = (int32_t)f.prow.size() - cur.r0;
    if (cur.nrows > 0) {
      cur.npm = (int32_t)f.pmem.size() - cur.pm0;
      cur.nvs = (int32_t)f.pvs.size() - cur.v0;
      f.panels.push_back(cur);
    } else {
      f.soff.pop_back();  // the opening 0 of an empty panel
    }
  };
  open_panel();
  for (const int32_t v : order) {
    int32_t pid = (int32_t)f.panels.size();
    int32_t new_slots = 0, new_mem = 0;
    for (int32_t p = ptr_v[v]; p < ptr_v[v + 1]; p++) {
      const int32_t e = ind_v[p];
      if (stamp[e] != pid) {  // a duplicate incidence is counted twice here; harmless
        new_slots++;
        new_mem += is_mat[e] ? 1 : (ptr_t[e + 1] - ptr_t[e]);
      }
    }
    const int32_t deg = ptr_v[v + 1] - ptr_v[v];
    const int32_t rows = (int32_t)f.prow.size() - cur.r0;
    const int32_t cur_mem = (int32_t)f.pmem.size() - cur.pm0;
    const int32_t cur_vs = (int32_t)f.pvs.size() - cur.v0;
    if (rows > 0 && (rows == f.rows_cap || cur.nslots + new_slots > f.cap ||
                     cur_mem + new_mem > f.mem_cap || cur_vs + deg > f.vslot_cap)) {
      close_panel();
      open_panel();
      pid = (int32_t)f.panels.size();
    }
    for (int32_t p = ptr_v[v]; p < ptr_v[v + 1]; p++) {
      const int32_t e = ind_v[p];
      if (stamp[e] != pid) {
        stamp[e] = pid;
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
      f.pvs.push_back((uint16_t)slot_of[e]);
    }
    f.prow.push_back(v);
    f.pend.push_back((int32_t)f.pvs.size() - cur.v0);
  }
  close_panel();
  f.pmem_entries = (int64_t)f.pmem.size();
}

}  // namespace hg
