// Host-side schedule for the fused (LDS-staged) variant.  Pure C++.
//
// The vertex rows are cut into panels; for each panel the distinct hyperedges
// its vertices touch become "slots" of an LDS tile.  A slot is either
// recomputed inside the workgroup from its member rows of X (small hyperedges)
// or loaded from a materialised table Xe_mat (hyperedges with more than t_big
// members), so the M x F hyperedge feature matrix never makes a round trip
// through HBM for the bulk of the graph.
//
// Vertices with more incident hyperedges than a panel can hold take one of two roads:
//  * register hubs (HubPass, hg_internal.h): the heaviest vertices of a large graph keep their
//    running sums in registers of persistent workgroups that stream all hyperedges containing a
//    hub, each hyperedge sum computed once per pass however many hubs it feeds;
//  * split vertices: the incidence list is cut into pieces of at most vdeg_max hyperedges, every
//    piece is an ordinary panel row that writes a partial sum instead of a row of Y.
// Both leave partial rows that the fixup pass adds up in a fixed order (deterministic).
#include <algorithm>
#include <numeric>

#include "hg_internal.h"

namespace hg {

void pack_records(FusedSched &f, int32_t ng, int32_t idle);
static void build_hub_pass(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ind_t, const int32_t *ptr_v,
                           const std::vector<uint8_t> &is_mat, const std::vector<int32_t> &mat_id,
                           const std::vector<int32_t> &hub_of, const std::vector<int32_t> &parts,
                           int32_t t_big, FusedSched &f, std::vector<Fixup> &finals);

// The classification step of build_fused alone (what the AUTO rule needs first): how many
// hyperedges would be materialised and how many vertices exceed a panel (hubs or split rows).
void classify_fused(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ptr_v, const int32_t *ind_v,
                    const Opts &o, int32_t cap, int32_t mem_cap, int64_t *n_mat, int64_t *n_big) {
  (void)ind_v;
  const int32_t t_big = std::max(1, std::min(o.t_big, mem_cap));
  const int32_t vdeg_max = std::max(1, std::min(std::min(cap, cap * 2), mem_cap / t_big));
  *n_mat = 0;
  for (int32_t e = 0; e < M; e++)
    if (ptr_t[e + 1] - ptr_t[e] > t_big) (*n_mat)++;
  *n_big = 0;
  for (int32_t v = 0; v < N; v++)
    if (ptr_v[v + 1] - ptr_v[v] > vdeg_max) (*n_big)++;
}

// Geometry of the hub pass for a tile row of `row_floats` floats: 1024 threads, LPR lanes per row.
static void hub_geometry(int32_t ng_panel, int32_t row_floats, int32_t tile_bytes, HubPass &h) {
  h.bs = 1024;
  h.ng = ng_panel * 4;  // 1024 / LPR where ng_panel = 256 / LPR
  h.R = kHubRows;
  h.cap = std::max(16, std::min(1024, tile_bytes / (row_floats * 4))) / 16 * 16;
  h.cap = std::max(h.cap, (h.ng + 15) / 16 * 16);  // the end-of-launch reduction parks one row per lane group in the tile
  // stream entries per round: two full batches of the kernel's 8 row loads in flight per lane
  // (a third, nearly empty batch would cost a whole round trip), never less than one slot's worth
  h.mem_cap = std::min(h.cap * 4, 16 * h.ng);
  h.pair_cap = std::max(h.ng * h.R, std::min(h.cap * 8, 8192));  // (hub, slot) pairs per round
}

// The kernel prefetches the next round's record with two dwordx4 per thread (32 KB at most) and LDS
// holds two records beside the tile and the hot rows: a record stays below 24 KB.  Upper bound of a
// round's record: the longest-first packing puts at most ceil(entries / ng) + (longest slot) steps on a
// lane group.
constexpr int32_t kHubRecWords = 6144;
static int32_t hub_rec_bound(const HubPass &h, int32_t entries, int32_t longest, int32_t nslots, int32_t npairs) {
  return 16 + h.ng + ((entries + h.ng - 1) / h.ng + longest + 1) * h.ng + nslots + (h.ng * h.R + 1) / 2 + (npairs + 1) / 2 + 4;
}

// Hop-1 entry stream of one record: the slots are spread over `ng` lane groups, longest first, each
// to the least loaded group, so every group walks about the same number of steps; the stream is laid
// out [step][group].  Recomputed slots (member rows of X) come first; materialised slots (one row of
// the materialised table each) follow from step `steps_x` on, dealt round-robin -- the kernels walk
// the two phases with one buffer descriptor each instead of testing every entry.  Idle entries name
// the row one past the table of their phase (no flags).  Slot ids are renumbered in (group,
// completion order): newid[k] = new id of slot k, gslots[g] = first id of group g.
struct SlotRange {
  int32_t m0, m1;   // entry words in `mem`: row index, bit 31 = row of the materialised table
  uint32_t hmask;   // heavy-hub mask for the last entry (hub pass; bits 24..27)
};
struct PackedStream {
  int32_t steps = 0, steps_x = 0;
  std::vector<int32_t> stream, newid, gslots;
};
static void pack_stream(const std::vector<SlotRange> &slots, const int32_t *mem, int32_t ng, int32_t idle_x,
                        int32_t idle_m, uint32_t row_mask, PackedStream &out) {
  const int32_t ns = (int32_t)slots.size();
  std::vector<int32_t> order_x, mats;
  for (int32_t k = 0; k < ns; k++) {
    const bool mat = slots[k].m1 - slots[k].m0 == 1 && ((uint32_t)mem[slots[k].m0] & 0x80000000u);
    (mat ? mats : order_x).push_back(k);
  }
  std::stable_sort(order_x.begin(), order_x.end(), [&](int32_t a, int32_t b) {
    return (slots[a].m1 - slots[a].m0) > (slots[b].m1 - slots[b].m0);
  });
  std::vector<int32_t> load((size_t)ng, 0);
  std::vector<std::vector<int32_t>> members((size_t)ng), mmembers((size_t)ng);
  for (int32_t k : order_x) {
    int32_t best = 0;
    for (int32_t g = 1; g < ng; g++)
      if (load[g] < load[best]) best = g;
    members[best].push_back(k);
    load[best] += slots[k].m1 - slots[k].m0;
  }
  out.steps_x = 0;
  for (int32_t g = 0; g < ng; g++) out.steps_x = std::max(out.steps_x, load[g]);
  for (size_t i = 0; i < mats.size(); i++) mmembers[i % (size_t)ng].push_back(mats[i]);
  const int32_t mat_steps = ((int32_t)mats.size() + ng - 1) / ng;
  out.steps = out.steps_x + mat_steps;
  out.stream.assign((size_t)out.steps * ng, idle_x);
  for (size_t i = (size_t)out.steps_x * ng; i < out.stream.size(); i++) out.stream[i] = idle_m;
  out.newid.assign((size_t)ns, 0);
  out.gslots.assign((size_t)ng, 0);
  int32_t next = 0;
  auto emit = [&](int32_t g, int32_t k, int32_t &s) {
    out.newid[k] = next++;
    for (int32_t p = slots[k].m0; p < slots[k].m1; p++, s++) {
      const uint32_t w = (uint32_t)mem[p];
      const uint32_t mat = (w & 0x80000000u) ? 0x40000000u : 0u;
      const uint32_t last = (p + 1 == slots[k].m1) ? (0x80000000u | (slots[k].hmask << 24)) : 0u;
      out.stream[(size_t)s * ng + g] = (int32_t)((w & row_mask) | mat | last);
    }
  };
  for (int32_t g = 0; g < ng; g++) {
    out.gslots[g] = next;
    int32_t s = 0;
    for (int32_t k : members[g]) emit(g, k, s);
    s = out.steps_x;
    for (int32_t k : mmembers[g]) emit(g, k, s);
  }
}

void build_fused(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ind_t,
                 const int32_t *ptr_v, const int32_t *ind_v, const Opts &o, int32_t cap,
                 int32_t mem_cap, int32_t ng, int32_t row_floats, bool allow_hub, FusedSched &f,
                 int32_t rows_cap) {
  f = FusedSched();
  f.cap = cap;
  f.rows_cap = rows_cap > 0 ? std::min(rows_cap, cap) : cap;
  f.mem_cap = mem_cap;
  f.vslot_cap = cap * 2;
  f.t_big = std::max(1, std::min(o.t_big, f.mem_cap));
  // a single row must always fit into an empty panel
  f.vdeg_max = std::max(1, std::min(std::min(cap, f.vslot_cap), f.mem_cap / f.t_big));
  const int64_t nnz = ptr_t[M];

  // sub-slots per hyperedge (Opts::slot_chunk; 1 everywhere in the default schedule)
  const int32_t chunk = o.slot_chunk > 0 ? std::max(o.slot_chunk, f.t_big) : 0;
  auto nsub = [&](int32_t e) { return chunk > 0 ? std::max(1, (ptr_t[e + 1] - ptr_t[e] + chunk - 1) / chunk) : 1; };
  // materialised hyperedges: compact CSR over their members
  std::vector<uint8_t> is_mat((size_t)M, 0);
  std::vector<int32_t> mat_id((size_t)M, -1);
  f.mat_ptr.push_back(0);
  for (int32_t e = 0; e < M; e++)
    if (chunk == 0 && ptr_t[e + 1] - ptr_t[e] > f.t_big) {
      is_mat[e] = 1;
      mat_id[e] = f.n_mat++;
      f.mat_eid.push_back(e);
      f.mat_ind.insert(f.mat_ind.end(), ind_t + ptr_t[e], ind_t + ptr_t[e + 1]);
      f.mat_ptr.push_back((int32_t)f.mat_ind.size());
    }
  build_sched(f.n_mat, f.mat_ptr.data(), o, f.mat_sched);
  if (allow_hub) build_row_stream(f.n_mat, f.mat_ptr.data(), f.mat_ind.data(), f.mat_eid.data(), ng, N, f.mat_stream);

  // ---- vertices that do not fit a panel: register hubs, or pieces ------------------------------
  // big[v]: 1 = register hub, 2 = split into pieces
  std::vector<uint8_t> big((size_t)N, 0);
  std::vector<int32_t> cand;
  for (int32_t v = 0; v < N; v++) {
    int64_t w = ptr_v[v + 1] - ptr_v[v];
    if (chunk > 0) {
      w = 0;
      for (int32_t p = ptr_v[v]; p < ptr_v[v + 1]; p++) w += nsub(ind_v[p]);
      if (w > std::min(f.vdeg_max, f.mem_cap / chunk)) {  // would need pieces cut by sub-slot count: not worth it on a launch-bound graph
        f.invalid = true;
        return;
      }
    }
    if (w > f.vdeg_max) {
      big[v] = 2;
      cand.push_back(v);
    }
  }
  std::vector<int32_t> hub_of((size_t)N, -1), parts;
  HubPass &hp = f.hub;
  hub_geometry(ng, row_floats, o.hub_tile_bytes, hp);
  const int32_t kHubMinDeg = o.hub_min_deg;
  // one hyperedge of t_big members and a pair for every hub must fit an empty round's record
  allow_hub = allow_hub && hub_rec_bound(hp, f.t_big, f.t_big, 1, hp.ng * hp.R) <= kHubRecWords;
  if (allow_hub && !(o.flags & HG_PLAN_NO_HUB_PASS) && nnz >= o.hub_min_nnz && !cand.empty()) {
    auto deg = [&](int32_t v) { return ptr_v[v + 1] - ptr_v[v]; };
    std::stable_sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return deg(a) > deg(b); });
    const int32_t vmax = hp.ng * hp.R;
    int64_t top = 0;
    for (size_t i = 0; i < cand.size() && (int32_t)i < vmax && deg(cand[i]) >= kHubMinDeg; i++) top += deg(cand[i]);
    if (top * 5 >= nnz) {  // the pass streams (nearly) every hyperedge: worth it for a fifth of the incidences
      // the heaviest few (each at least 1/512 of all incidences) ride on stream flags, not on virtual rows
      // A flag bit says "this hyperedge contains the hub", once: a vertex listed twice in some hyperedge (duplicate
      // incidences are legal input) keeps virtual rows, whose pair lists carry one pair per incidence.  The heavy
      // ones move to the front of `cand`, so that hub ids [0, n_heavy) are exactly the flag-fed hubs.
      auto listed_twice = [&](int32_t v) {  // a vertex's hyperedges are ascending (stable transpose)
        for (int32_t p = ptr_v[v] + 1; p < ptr_v[v + 1]; p++)
          if (ind_v[p] == ind_v[p - 1]) return true;
        return false;
      };
      int64_t heavy_sum = 0;
      for (size_t i = 0; i < cand.size() && hp.n_heavy < kHubHeavy && deg(cand[i]) >= kHubMinDeg &&
                         (int64_t)deg(cand[i]) * 512 >= nnz; i++) {
        if (listed_twice(cand[i])) continue;
        heavy_sum += deg(cand[i]);
        std::rotate(cand.begin() + hp.n_heavy, cand.begin() + i, cand.begin() + i + 1);
        hp.n_heavy++;
      }
      // no virtual row heavier than half a lane group's fair share of a round.  (A wave's hop 2 costs, per chunk
      // of kHubChunk rows, the longest list among the chunk's rows of all its lane groups, so smaller parts shorten a
      // round's critical path -- modelled 75.8 k -> 34.2 k incidences with a sixth of a share -- but every hub that
      // loses its register row to them comes back as pieces in the panels: measured with the sorted assignment of
      // build_hub_pass, power-law F = 64: 1/2 share 0.924 ms, 1/3 0.926, 1/6 0.941.)
      const int64_t wmax = std::max<int64_t>(1, (top - heavy_sum) / (2 * (int64_t)hp.ng));
      int32_t nv = 0;
      for (size_t i = 0; i < cand.size() && deg(cand[i]) >= kHubMinDeg; i++) {
        const bool heavy = (int32_t)i < hp.n_heavy;
        const int32_t p = heavy ? 0 : (int32_t)std::min<int64_t>(64, (deg(cand[i]) + wmax - 1) / wmax);
        if (nv + p > vmax) break;
        nv += p;
        hub_of[cand[i]] = hp.K++;
        hp.vid.push_back(cand[i]);
        parts.push_back(p);
        big[cand[i]] = 1;
      }
      hp.nv = nv;
    }
  }

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
      if (vis[root] || big[root]) continue;
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
            if (!vis[u] && !big[u]) {
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
  // One panel row: the incidences [p0, p1) of vertex v (all of them, or one piece of a split
  // vertex), written to `dest` (vertex id, or 0x80000000 | partial slot).
  auto add_row = [&](int32_t v, int32_t p0, int32_t p1, int32_t dest, bool feed_greedy) {
    int32_t pid = (int32_t)f.panels.size();
    int32_t new_slots = 0, new_mem = 0, deg = 0;
    for (int32_t p = p0; p < p1; p++) {
      const int32_t e = ind_v[p];
      deg += nsub(e);
      if (stamp[e] != pid) {  // a duplicate incidence is counted twice here; harmless
        new_slots += nsub(e);
        new_mem += is_mat[e] ? 1 : (ptr_t[e + 1] - ptr_t[e]);
      }
    }
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
    if (greedy && feed_greedy) done[v] = 1;
    for (int32_t p = p0; p < p1; p++) {
      const int32_t e = ind_v[p];
      const int32_t k = nsub(e);
      if (stamp[e] != pid) {
        stamp[e] = pid;
        slot_of[e] = cur.nslots;
        cur.nslots += k;
        if (is_mat[e]) {
          f.pmem.push_back((int32_t)(0x80000000u | (uint32_t)mat_id[e]));
          f.slot_eid.push_back(-1);
          f.soff.push_back((int32_t)f.pmem.size() - cur.pm0);
        } else {
          const int32_t len = ptr_t[e + 1] - ptr_t[e];
          for (int32_t j = 0; j < k; j++) {  // k = 1: the whole hyperedge; else chunks of `chunk` members
            const int32_t b0 = ptr_t[e] + (k > 1 ? j * chunk : 0), b1 = k > 1 ? std::min(b0 + chunk, ptr_t[e] + len) : ptr_t[e] + len;
            f.pmem.insert(f.pmem.end(), ind_t + b0, ind_t + b1);
            f.slot_eid.push_back(e);
            f.soff.push_back((int32_t)f.pmem.size() - cur.pm0);
          }
          if (greedy && feed_greedy)
            for (int32_t q = ptr_t[e]; q < ptr_t[e + 1]; q++) {
              const int32_t u = ind_t[q];
              if (done[u] || big[u]) continue;
              if (score[u]++ == 0) touched.push_back(u);
              heap.push_back(Cand{(float)score[u] / (float)(ptr_v[u + 1] - ptr_v[u]), score[u], u});
              std::push_heap(heap.begin(), heap.end());
            }
        }
      }
      for (int32_t j = 0; j < k; j++) f.pvs.push_back((uint16_t)(slot_of[e] + j));
    }
    f.prow.push_back(dest);
    f.pend.push_back((int32_t)f.pvs.size() - cur.v0);
  };
  open_panel();
  for (int32_t v = next_vertex(); v >= 0; v = next_vertex()) add_row(v, ptr_v[v], ptr_v[v + 1], v, true);

  // partial rows: [hub parts x workgroups][pieces][first-level fixup sums]
  std::vector<Fixup> finals;
  if (hp.K > 0) build_hub_pass(N, M, ptr_t, ind_t, ptr_v, is_mat, mat_id, hub_of, parts, f.t_big, f, finals);
  // split vertices: pieces of vdeg_max incidences, each a panel row of its own
  if (greedy) reset_candidates();
  for (int32_t v = 0; v < N; v++) {
    if (big[v] != 2) continue;
    f.n_split++;
    const int32_t first = f.n_part;
    for (int32_t p0 = ptr_v[v]; p0 < ptr_v[v + 1]; p0 += f.vdeg_max) {
      const int32_t p1 = std::min(p0 + f.vdeg_max, ptr_v[v + 1]);
      add_row(v, p0, p1, (int32_t)(0x80000000u | (uint32_t)f.n_part), false);
      f.n_part++;
    }
    add_fixups(v, first, f.n_part - first, f.n_part, f.fixups, finals);
  }
  close_panel();
  f.n_fix_l1 = (int32_t)f.fixups.size();
  f.fixups.insert(f.fixups.end(), finals.begin(), finals.end());
  f.pmem_entries = (int64_t)f.pmem.size();
  pack_records(f, ng, N);
}

// Rounds of the hub pass: the hyperedges that contain a register hub, in ascending order, cut
// into rounds of at most cap slots / mem_cap stream entries / pair_cap (hub, slot) pairs; rounds
// dealt to the persistent workgroups as contiguous, equally heavy ranges.  Record of one round:
//   header  : [0] steps  [1] nslots  [2] npairs  [4] off_gbase  [5] off_stream  [6] off_pend
//             [8] steps_x (materialised slots start here)  [9] off_pvs  [10] off_eid
//   gbase   : ng words, first slot id of each lane group (as in a panel record)
//   stream  : steps * ng entry words (as in a panel record; indices below 2^24; the last entry of a slot
//             carries the slot's heavy-hub mask in bits 24..27)
//   eid     : nslots hyperedge ids in slot order (-1: materialised, already scaled)
//   pend    : ng * R 16-bit cumulative end offsets into pvs, one per virtual row (id = group * R + i)
//   pvs     : npairs 16-bit slot ids, grouped by virtual row, ascending hyperedge inside a row
static void build_hub_pass(int32_t N, int32_t M, const int32_t *ptr_t, const int32_t *ind_t, const int32_t *ptr_v,
                           const std::vector<uint8_t> &is_mat, const std::vector<int32_t> &mat_id,
                           const std::vector<int32_t> &hub_of, const std::vector<int32_t> &parts,
                           int32_t t_big, FusedSched &f, std::vector<Fixup> &finals) {
  (void)t_big;
  HubPass &hp = f.hub;
  const int32_t ng = hp.ng, R = hp.R, idle = N;
  // virtual rows -> (lane group, register): heaviest first, each to the least loaded group with a free register
  struct VRow {
    int32_t hub, part;
    int64_t w;
  };
  std::vector<VRow> vr;
  for (int32_t h = 0; h < hp.K; h++) {
    const int64_t d = ptr_v[hp.vid[h] + 1] - ptr_v[hp.vid[h]];
    for (int32_t j = 0; j < parts[h]; j++) vr.push_back(VRow{h, j, (d + parts[h] - 1) / parts[h]});
  }
  std::stable_sort(vr.begin(), vr.end(), [](const VRow &a, const VRow &b) { return a.w > b.w; });
  std::vector<int32_t> part0((size_t)hp.K + 1, 0);  // first index of each hub's parts in vrow_id
  for (int32_t h = 0; h < hp.K; h++) part0[h + 1] = part0[h] + parts[h];
  std::vector<int32_t> vrow_id((size_t)part0[hp.K], -1);  // (hub, part) -> group * R + i
  // Rows of like weight share a chunk, and neighbouring chunks go to neighbouring lane groups (the same wave):
  // kHubChunk consecutive rows of the sorted list fill chunk c of group k, the groups walked back and forth
  // from one chunk level to the next so that every group gets a heavy, a middling and a light chunk.
  static_assert(kHubRows % kHubChunk == 0, "chunks tile the rows of a lane group");
  for (size_t i = 0; i < vr.size(); i++) {
    const int32_t q = (int32_t)(i / kHubChunk), c = q / ng, k = q % ng;
    const int32_t g = (c & 1) ? ng - 1 - k : k;
    vrow_id[part0[vr[i].hub] + vr[i].part] = g * R + c * kHubChunk + (int32_t)(i % kHubChunk);
  }

  // rounds
  std::vector<int32_t> cnt((size_t)hp.K, 0), touched_h;
  std::vector<int64_t> round_w;
  struct Slot {
    int32_t e, m0, m1;  // hyperedge, range in `mem`
    uint32_t hmask;     // heavy hubs among its members (bit h = hub h)
  };
  std::vector<Slot> slots;
  std::vector<int32_t> mem;                         // stream entry words of the round's slots, unflagged
  std::vector<std::pair<int32_t, int32_t>> pairs;   // (virtual row, slot index)
  std::vector<int32_t> eh;                          // hub members of the current hyperedge
  int32_t longest = 0;                              // longest slot of the open round
  auto flush = [&]() {
    if (slots.empty()) return;
    const int32_t ns = (int32_t)slots.size();
    std::vector<SlotRange> sr((size_t)ns);
    for (int32_t k = 0; k < ns; k++) sr[k] = SlotRange{slots[k].m0, slots[k].m1, slots[k].hmask};
    PackedStream ps;
    // rows stay below 2^24 wherever the hub pass runs: bits 24..27 of a slot's last entry are its heavy-hub mask
    pack_stream(sr, mem.data(), ng, idle, f.n_mat, 0x00ffffffu, ps);
    const int32_t steps = ps.steps;
    const std::vector<int32_t> &newid = ps.newid, &stream = ps.stream, &gslots = ps.gslots;
    std::vector<int32_t> eid((size_t)ns, -1);
    for (int32_t k = 0; k < ns; k++) eid[newid[k]] = is_mat[slots[k].e] ? -1 : slots[k].e;
    // pairs by virtual row (stable: ascending hyperedge inside a row)
    std::stable_sort(pairs.begin(), pairs.end(),
                     [](const std::pair<int32_t, int32_t> &a, const std::pair<int32_t, int32_t> &b) { return a.first < b.first; });
    const int32_t nvr = ng * R, npairs = (int32_t)pairs.size();
    const int32_t hdr = 16;
    const int32_t off_gbase = hdr, off_stream = off_gbase + ng, off_eid = off_stream + steps * ng;
    const int32_t off_pend = off_eid + ns, off_pvs = off_pend + (nvr + 1) / 2;
    const int32_t words = (off_pvs + (npairs + 1) / 2 + 3) & ~3;
    HubRec rt;
    rt.off = (int64_t)hp.rec.size();
    rt.len = words;
    rt.nslots = ns;
    rt.off_eid = off_eid;
    rt.pad = 0;
    hp.rec_tab.push_back(rt);
    hp.max_rec_words = std::max(hp.max_rec_words, words);
    hp.max_steps = std::max(hp.max_steps, steps);
    hp.stream_entries += (int64_t)steps * ng;
    hp.pairs += npairs;
    round_w.push_back((int64_t)steps * ng + npairs);
    const size_t base = hp.rec.size();
    hp.rec.resize(base + (size_t)words, 0);
    int32_t *r = hp.rec.data() + base;
    r[0] = steps;
    r[1] = ns;
    r[2] = npairs;
    r[8] = ps.steps_x;
    r[4] = off_gbase;
    r[5] = off_stream;
    r[6] = off_pend;
    r[9] = off_pvs;
    r[10] = off_eid;
    for (int32_t g = 0; g < ng; g++) r[off_gbase + g] = gslots[g];
    std::copy(stream.begin(), stream.end(), r + off_stream);
    std::copy(eid.begin(), eid.end(), r + off_eid);
    uint16_t *pe = reinterpret_cast<uint16_t *>(r + off_pend);
    uint16_t *pv = reinterpret_cast<uint16_t *>(r + off_pvs);
    int32_t q = 0;
    for (int32_t v = 0; v < nvr; v++) {
      while (q < npairs && pairs[q].first == v) {
        pv[q] = (uint16_t)newid[pairs[q].second];
        q++;
      }
      pe[v] = (uint16_t)q;
    }
    for (int32_t h : touched_h) cnt[h] = 0;
    touched_h.clear();
    slots.clear();
    mem.clear();
    pairs.clear();
  };
  for (int32_t e = 0; e < M; e++) {
    eh.clear();
    for (int32_t p = ptr_t[e]; p < ptr_t[e + 1]; p++)
      if (hub_of[ind_t[p]] >= 0) eh.push_back(hub_of[ind_t[p]]);
    if (eh.empty()) continue;
    const int32_t cost = is_mat[e] ? 1 : ptr_t[e + 1] - ptr_t[e];
    if (!slots.empty() &&
        ((int32_t)slots.size() == hp.cap || (int32_t)mem.size() + cost > hp.mem_cap ||
         (int32_t)pairs.size() + (int32_t)eh.size() > hp.pair_cap ||
         hub_rec_bound(hp, (int32_t)mem.size() + cost, std::max(longest, cost), (int32_t)slots.size() + 1,
                       (int32_t)pairs.size() + (int32_t)eh.size()) > kHubRecWords)) {
      flush();
      longest = 0;
    }
    longest = std::max(longest, cost);
    const int32_t k = (int32_t)slots.size();
    const int32_t m0 = (int32_t)mem.size();
    if (is_mat[e]) mem.push_back((int32_t)(0x80000000u | (uint32_t)mat_id[e]));
    else mem.insert(mem.end(), ind_t + ptr_t[e], ind_t + ptr_t[e + 1]);
    uint32_t hmask = 0;
    for (int32_t h : eh)
      if (h < hp.n_heavy) hmask |= 1u << h;
    slots.push_back(Slot{e, m0, (int32_t)mem.size(), hmask});
    for (int32_t h : eh) {
      if (h < hp.n_heavy) continue;
      if (cnt[h] == 0) touched_h.push_back(h);
      const int32_t part = cnt[h]++ % parts[h];  // a hub's incidences are dealt round-robin to its parts
      pairs.emplace_back(vrow_id[part0[h] + part], k);
    }
  }
  flush();

  // contiguous, equally heavy round ranges for the persistent workgroups
  const int32_t nrounds = (int32_t)hp.rec_tab.size();
  hp.nwg = std::max(1, std::min(256, nrounds));
  int64_t total = 0;
  for (int64_t w : round_w) total += w;
  hp.wg_first.assign((size_t)hp.nwg + 1, nrounds);
  hp.wg_first[0] = 0;
  {
    int64_t acc = 0;
    int32_t w = 1;
    for (int32_t r = 0; r < nrounds && w < hp.nwg; r++) {
      acc += round_w[r];
      while (w < hp.nwg && acc * hp.nwg >= total * w) hp.wg_first[w++] = r + 1;
    }
  }
  for (int32_t w = 1; w <= hp.nwg; w++) hp.wg_first[w] = std::max(hp.wg_first[w], hp.wg_first[w - 1]);
  hp.wg_first[hp.nwg] = nrounds;

  // partial slots: hub h owns parts[h] * nwg consecutive rows; fixups add them (rows = vertex ids)
  hp.vslot0.assign((size_t)ng * R, -1);
  for (int32_t h = 0; h < hp.n_heavy; h++) {  // one partial row per workgroup
    hp.hslot0[h] = f.n_part;
    f.n_part += hp.nwg;
    add_fixups(hp.vid[h], hp.hslot0[h], hp.nwg, f.n_part, f.fixups, finals);
  }
  for (int32_t h = hp.n_heavy; h < hp.K; h++) {
    const int32_t base = f.n_part;
    for (int32_t j = 0; j < parts[h]; j++) hp.vslot0[vrow_id[part0[h] + j]] = base + j * hp.nwg;
    f.n_part += parts[h] * hp.nwg;
    add_fixups(hp.vid[h], base, parts[h] * hp.nwg, f.n_part, f.fixups, finals);
  }
}


void build_row_stream(int32_t nrows, const int32_t *ptr, const int32_t *ind, const int32_t *scale_index,
                      int32_t ng, int32_t idle, RowStream &rs) {
  rs = RowStream();
  rs.ng = ng;
  // A lane group walks at most kRowStreamChunk entries for one row, and a workgroup's record holds 64 steps'
  // worth of entries: with shorter records a single 64-entry row sets the step count of a record whose
  // other lane groups hold a quarter of that (power-law config, materialisation of 375 k rows: 258 us with
  // 16-step records, 136 us with 64-step ones; chunks of 32 or 128: 137 / 149 us).
  constexpr int32_t kChunk = kRowStreamChunk;
  const int32_t ent_cap = 64 * ng;
  rs.cap = 4 * ng;                    // rows (or chunks) per workgroup
  struct Unit {
    int32_t m0, m1, dst, sidx;
  };
  std::vector<Unit> units;
  std::vector<int32_t> mem;
  std::vector<Fixup> finals;
  auto flush = [&]() {
    if (units.empty()) return;
    const int32_t ns = (int32_t)units.size();
    std::vector<SlotRange> sr((size_t)ns);
    for (int32_t k = 0; k < ns; k++) sr[k] = SlotRange{units[k].m0, units[k].m1, 0u};
    PackedStream ps;
    pack_stream(sr, mem.data(), ng, idle, idle, 0x3fffffffu, ps);
    const int32_t hdr = 8;
    const int32_t off_gbase = hdr, off_stream = off_gbase + ng, off_dst = off_stream + ps.steps * ng;
    const int32_t off_sidx = off_dst + ns;
    const int32_t words = (off_sidx + ns + 3) & ~3;
    SRec rt;
    rt.off = (int64_t)rs.rec.size();
    rt.len = words;
    rt.nslots = ns;
    rt.off_sidx = off_sidx;
    rt.pad = 0;
    rs.rec_tab.push_back(rt);
    rs.max_rec_words = std::max(rs.max_rec_words, words);
    rs.max_steps = std::max(rs.max_steps, ps.steps);
    rs.entries += (int64_t)ps.steps * ng;
    const size_t base = rs.rec.size();
    rs.rec.resize(base + (size_t)words, 0);
    int32_t *r = rs.rec.data() + base;
    r[0] = ps.steps;
    r[1] = ns;
    r[4] = off_gbase;
    r[5] = off_stream;
    r[6] = off_dst;
    r[7] = off_sidx;
    for (int32_t g = 0; g < ng; g++) r[off_gbase + g] = ps.gslots[g];
    std::copy(ps.stream.begin(), ps.stream.end(), r + off_stream);
    for (int32_t k = 0; k < ns; k++) {
      r[off_dst + ps.newid[k]] = units[k].dst;
      r[off_sidx + ps.newid[k]] = units[k].sidx;
    }
    units.clear();
    mem.clear();
  };
  auto add_unit = [&](int32_t b, int32_t e, int32_t dst, int32_t sidx) {
    const int32_t n = std::max(1, e - b);
    if (!units.empty() && ((int32_t)units.size() == rs.cap || (int32_t)mem.size() + n > ent_cap)) flush();
    const int32_t m0 = (int32_t)mem.size();
    if (e > b) mem.insert(mem.end(), ind + b, ind + e);
    else mem.push_back(idle);  // an empty row: one idle entry (zeros) that carries the last flag
    units.push_back(Unit{m0, (int32_t)mem.size(), dst, sidx});
  };
  for (int32_t r = 0; r < nrows; r++) {
    const int32_t len = ptr[r + 1] - ptr[r];
    const int32_t si = scale_index ? scale_index[r] : r;
    if (len <= kChunk) {
      add_unit(ptr[r], ptr[r + 1], r, len > 0 ? si : -1);  // an empty row stays exactly 0 (its scale may be inf)
    } else {
      const int32_t k = (len + kChunk - 1) / kChunk, first = rs.nslots;
      for (int32_t c = 0; c < k; c++)
        add_unit(ptr[r] + c * kChunk, std::min(ptr[r] + (c + 1) * kChunk, ptr[r + 1]),
                 (int32_t)(0x80000000u | (uint32_t)(rs.nslots++)), -1);
      add_fixups(r, first, k, rs.nslots, rs.fixups, finals);
    }
  }
  flush();
  rs.n_fix_l1 = (int32_t)rs.fixups.size();
  rs.fixups.insert(rs.fixups.end(), finals.begin(), finals.end());
}

// One self-contained int32 record per panel for the packed kernel: a header, the
// hop-1 entry stream laid out [step][group] (the panel's slots are spread over
// the `ng` lane groups with longest-first greedy packing, so every group walks
// the same number of steps and the loop is wave-uniform), then the hop-2 lists.
//   header  : [0] steps  [1] nrows  [2] nslots  [3] nvs
//             [4] off_gbase [5] off_stream [6] off_pend [7] off_prow [8] steps_x [9] off_pvs
//   gbase   : ng words, first slot id of each group (a group's slots are numbered in
//             the order it finishes them)
//   stream  : steps * ng words (pack_stream): steps [0, steps_x) hold member rows of X, steps
//             [steps_x, steps) rows of the materialised table; an idle step names the row one past
//             the table of its phase (N / n_mat, no flags); else bits 0..29 row index, bit 30 = row
//             of the materialised table, bit 31 = last entry of its slot
//   prow    : nrows destinations: vertex id, or 0x80000000 | partial row (a piece of a split vertex)
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
  for (const FPanel &pn : f.panels) {
    const int32_t *soff = f.soff.data() + pn.sbase;
    const int32_t *pm = f.pmem.data() + pn.pm0;
    std::vector<SlotRange> sr((size_t)pn.nslots);
    for (int32_t k = 0; k < pn.nslots; k++) sr[k] = SlotRange{soff[k], soff[k + 1], 0u};
    PackedStream ps;
    pack_stream(sr, pm, ng, idle, f.n_mat, 0x3fffffffu, ps);
    const int32_t steps = ps.steps;
    const std::vector<int32_t> &newid = ps.newid, &stream = ps.stream, &gslots = ps.gslots;
    std::vector<int32_t> eid((size_t)pn.nslots);
    for (int32_t k = 0; k < pn.nslots; k++) eid[newid[k]] = f.slot_eid[pn.eid0 + k];
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
    r[8] = ps.steps_x;
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
