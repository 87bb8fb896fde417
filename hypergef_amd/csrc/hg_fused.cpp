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

void pack_records(FusedSched &f, int32_t ng, int32_t idle);

// The classification step of build_fused alone (what the AUTO rule needs first): how many
// vertices would be hubs and how many hyperedges materialised for these capacities.
void classify_fused(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ptr_v, const int32_t *ind_v,
                    const Opts &o, int32_t cap, int32_t mem_cap, int64_t *n_mat, int64_t *n_hub) {
  const int32_t t_big = std::max(1, std::min(o.t_big, mem_cap));
  const int32_t vdeg_max = std::max(1, std::min(std::min(cap, cap * 2), mem_cap / t_big));
  std::vector<uint8_t> is_mat((size_t)M, 0);
  *n_hub = 0;
  for (int32_t e = 0; e < M; e++)
    if (ptr_t[e + 1] - ptr_t[e] > t_big) is_mat[e] = 1;
  for (int32_t v = 0; v < N; v++)
    if (ptr_v[v + 1] - ptr_v[v] > vdeg_max) {
      (*n_hub)++;
      for (int32_t p = ptr_v[v]; p < ptr_v[v + 1]; p++) is_mat[ind_v[p]] = 1;
    }
  *n_mat = 0;
  for (int32_t e = 0; e < M; e++) *n_mat += is_mat[e];
}

void build_fused(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ind_t,
                 const int32_t *ptr_v, const int32_t *ind_v, const Opts &o, int32_t cap,
                 int32_t mem_cap, int32_t ng, FusedSched &f) {
  f = FusedSched();
  f.cap = cap;
  f.rows_cap = cap;
  f.mem_cap = mem_cap;
  f.vslot_cap = cap * 2;
  f.t_big = std::max(1, std::min(o.t_big, f.mem_cap));
  // a single row must always fit into an empty panel
  f.vdeg_max = std::max(1, std::min(std::min(cap, f.vslot_cap), f.mem_cap / f.t_big));

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
        while (!descended && fr.ep < ptr_v[fr.v + 1]) {  // (fr dangles once descended: test that first)
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

  // fused panels
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
  // Rows are taken greedily: after a vertex joins the panel, the vertices that share the
  // panel's recomputed hyperedges become candidates, and the next row is the candidate with
  // the largest share of its own hyperedges already in the panel (ties: most shared) -- it
  // adds the fewest new slots and member gathers per row.  The vertex that no longer fits
  // opens the next panel, so neighbouring panels also share rows of X in L2.  With no
  // candidate left, the depth-first order above supplies the next vertex.
  struct Cand {
    float frac;
    int32_t shared, v;
    bool operator<(const Cand &o) const {
      return frac != o.frac ? frac < o.frac : (shared != o.shared ? shared < o.shared : v > o.v);
    }
  };
  const bool greedy = !(o.flags & HG_PLAN_DFS_ORDER);
  std::vector<Cand> heap;
  std::vector<int32_t> score(greedy ? (size_t)N : 0, 0), touched;
  std::vector<uint8_t> done(greedy ? (size_t)N : 0, 0);
  size_t next_in_order = 0;
  auto reset_candidates = [&]() {
    for (const int32_t u : touched) score[u] = 0;
    touched.clear();
    heap.clear();
  };
  auto next_vertex = [&]() -> int32_t {
    if (greedy) {
      while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end());
        const Cand c = heap.back();
        heap.pop_back();
        if (!done[c.v] && score[c.v] == c.shared) return c.v;  // else: stale entry
      }
      while (next_in_order < order.size() && done[order[next_in_order]]) next_in_order++;
    }
    return next_in_order < order.size() ? order[next_in_order++] : -1;
  };
  open_panel();
  for (int32_t v = next_vertex(); v >= 0; v = next_vertex()) {
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
      if (greedy) reset_candidates();
    }
    if (greedy) done[v] = 1;
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
          if (greedy)
            for (int32_t q = ptr_t[e]; q < ptr_t[e + 1]; q++) {
              const int32_t u = ind_t[q];
              if (done[u] || is_hub[u]) continue;
              if (score[u]++ == 0) touched.push_back(u);
              heap.push_back(Cand{(float)score[u] / (float)(ptr_v[u + 1] - ptr_v[u]), score[u], u});
              std::push_heap(heap.begin(), heap.end());
            }
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
  pack_records(f, ng, N);
}

// One self-contained int32 record per panel for the packed kernel: a header, the
// hop-1 entry stream laid out [step][group] (the panel's slots are spread over
// the `ng` lane groups with longest-first greedy packing, so every group walks
// the same number of steps and the loop is wave-uniform), then the hop-2 lists.
//   header  : [0] steps  [1] nrows  [2] nslots  [3] nvs
//             [4] off_gbase [5] off_stream [6] off_pend [7] off_prow [9] off_pvs
//   gbase   : ng words, first slot id of each group (a group's slots are numbered in
//             the order it finishes them)
//   stream  : steps * ng words; `idle` (= N, one past the last row of X, no flags) = idle
//             step, else bits 0..29 row index, bit 30 = row of the materialised table,
//             bit 31 = last entry of its slot
//   prow    : nrows vertex ids
//   pend    : nrows local end offsets into pvs, two 16-bit values per word
//   pvs     : nvs slot ids, two 16-bit values per word
// The hyperedge id of every slot (-1 = materialised, already scaled) goes to eid_all in
// record order, outside the records: only unbound scaling reads it.
void pack_records(FusedSched &f, int32_t ng, int32_t idle) {
  f.ng = ng;
  f.rec.clear();
  f.rec_tab.clear();
  f.eid_all.clear();
  f.max_rec_words = 0;
  f.max_steps = 0;
  f.stream_entries = 0;
  std::vector<int32_t> order, load, gslots, newid, stream;
  for (const FPanel &pn : f.panels) {
    const int32_t *soff = f.soff.data() + pn.sbase;
    const int32_t *pm = f.pmem.data() + pn.pm0;
    // longest slot first, each to the least loaded group
    order.resize((size_t)pn.nslots);
    for (int32_t k = 0; k < pn.nslots; k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
      return (soff[x + 1] - soff[x]) > (soff[y + 1] - soff[y]);
    });
    load.assign((size_t)ng, 0);
    std::vector<std::vector<int32_t>> members((size_t)ng);
    for (int32_t k : order) {
      int32_t best = 0;
      for (int32_t g = 1; g < ng; g++)
        if (load[g] < load[best]) best = g;
      members[best].push_back(k);
      load[best] += soff[k + 1] - soff[k];
    }
    int32_t steps = 0;
    for (int32_t g = 0; g < ng; g++) steps = std::max(steps, load[g]);
    // slot ids in (group, completion order); streams
    newid.assign((size_t)pn.nslots, 0);
    gslots.assign((size_t)ng, 0);
    stream.assign((size_t)steps * ng, idle);
    int32_t next = 0;
    std::vector<int32_t> eid((size_t)pn.nslots);
    for (int32_t g = 0; g < ng; g++) {
      gslots[g] = next;
      int32_t s = 0;
      for (int32_t k : members[g]) {
        newid[k] = next;
        eid[next] = f.slot_eid[pn.eid0 + k];
        next++;
        for (int32_t p = soff[k]; p < soff[k + 1]; p++, s++) {
          uint32_t w = (uint32_t)pm[p];
          const uint32_t row = w & 0x3fffffffu;
          const uint32_t mat = (w & 0x80000000u) ? 0x40000000u : 0u;
          const uint32_t last = (p + 1 == soff[k + 1]) ? 0x80000000u : 0u;
          stream[(size_t)s * ng + g] = (int32_t)(row | mat | last);
        }
      }
    }
    const int32_t hdr = 16;
    const int32_t off_gbase = hdr, off_stream = off_gbase + ng, off_prow = off_stream + steps * ng;
    const int32_t off_pend = off_prow + pn.nrows, off_pvs = off_pend + (pn.nrows + 1) / 2;
    const int32_t words = (off_pvs + (pn.nvs + 1) / 2 + 3) & ~3;  // whole 16-byte units
    FRec rt;
    rt.off = (int64_t)f.rec.size();
    rt.len = words;
    rt.off_prow = off_prow;
    rt.pad = 0;
    rt.nrows = pn.nrows;
    rt.nslots = pn.nslots;
    rt.slot_base = (int32_t)f.eid_all.size();
    rt.row_base = pn.r0;
    f.eid_all.insert(f.eid_all.end(), eid.begin(), eid.end());
    f.rec_tab.push_back(rt);
    f.max_rec_words = std::max(f.max_rec_words, words);
    f.max_steps = std::max(f.max_steps, steps);
    f.stream_entries += (int64_t)steps * ng;
    const size_t base = f.rec.size();
    f.rec.resize(base + (size_t)words, 0);
    int32_t *r = f.rec.data() + base;
    r[0] = steps;
    r[1] = pn.nrows;
    r[2] = pn.nslots;
    r[3] = pn.nvs;
    r[4] = off_gbase;
    r[5] = off_stream;
    r[6] = off_pend;
    r[7] = off_prow;
    r[9] = off_pvs;
    for (int32_t g = 0; g < ng; g++) r[off_gbase + g] = gslots[g];
    std::copy(stream.begin(), stream.end(), r + off_stream);
    uint16_t *pe = reinterpret_cast<uint16_t *>(r + off_pend);
    for (int32_t i = 0; i < pn.nrows; i++) {
      pe[i] = (uint16_t)f.pend[pn.r0 + i];  // <= vslot_cap
      r[off_prow + i] = f.prow[pn.r0 + i];
    }
    uint16_t *pv = reinterpret_cast<uint16_t *>(r + off_pvs);
    for (int32_t i = 0; i < pn.nvs; i++) pv[i] = (uint16_t)newid[f.pvs[pn.v0 + i]];
  }
}

}  // namespace hg
