// Host-side schedule builders under AddressSanitizer / UBSan (CPU only; tests/test_host.py
// compiles and runs this).  Random hypergraphs of many shapes -- empty rows, hubs, duplicates,
// one hyperedge holding everything -- through transpose_csr, build_sched and build_fused (both
// row orders, several capacities); checks the structural invariants the kernels rely on.
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "hg_internal.h"

#define CHECK(c)                                                     \
  do {                                                               \
    if (!(c)) {                                                      \
      std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                  \
    }                                                                \
  } while (0)

// Interpret the fused schedule exactly as the kernels do -- materialisation table, hub-pass rounds
// (entry streams, tiles, 16-bit pair lists, register rows, partial rows per workgroup), panel records
// (piece rows with bit 31), fixups (first level, then final) -- on exact integer "features", and
// compare with Y = H H^T X computed directly.  Every vertex row must be written exactly once.
static long g_stream_rows = 0, g_stream_chunks = 0, g_chunked = 0, g_hub_graphs = 0, g_hub_rounds = 0, g_hub_parts = 0, g_split_rows = 0, g_l1_fixups = 0, g_heavy = 0, g_rows_capped = 0;

static void emulate_fused(const hg::FusedSched &f, int N, int M, const std::vector<int32_t> &ptr,
                          const std::vector<int32_t> &ind, const std::vector<int32_t> &ptr_v,
                          const std::vector<int32_t> &ind_v, std::mt19937 &rng) {
  std::vector<int64_t> X((size_t)N), Xe((size_t)M, 0), want((size_t)N, 0);
  for (auto &x : X) x = (int64_t)(rng() % 1000);
  for (int e = 0; e < M; e++)
    for (int p = ptr[e]; p < ptr[e + 1]; p++) Xe[e] += X[ind[p]];
  for (int v = 0; v < N; v++)
    for (int p = ptr_v[v]; p < ptr_v[v + 1]; p++) want[v] += Xe[ind_v[p]];
  // (a) materialised table
  std::vector<int64_t> mat((size_t)f.n_mat, 0);
  for (int i = 0; i < f.n_mat; i++) {
    CHECK(f.mat_eid[i] >= 0 && f.mat_eid[i] < M);
    for (int p = f.mat_ptr[i]; p < f.mat_ptr[i + 1]; p++) mat[i] += X[f.mat_ind[p]];
    CHECK(mat[i] == Xe[f.mat_eid[i]]);
  }
  // the same table through the streaming row gather's records (stream_rows_kernel): rows spread over the lane
  // groups, chunks of long rows into partial rows, two fixup levels
  const hg::RowStream &ms = f.mat_stream;
  if (!ms.rec_tab.empty()) {
    std::vector<int64_t> out((size_t)f.n_mat, INT64_MIN), part((size_t)ms.nslots, INT64_MIN);
    for (const hg::SRec &rt : ms.rec_tab) {
      CHECK(rt.off >= 0 && rt.off + rt.len <= (int64_t)ms.rec.size() && rt.len % 4 == 0 && rt.len <= ms.max_rec_words &&
            rt.nslots >= 1 && rt.nslots <= ms.cap);
      const int32_t *r = ms.rec.data() + rt.off;
      CHECK(r[1] == rt.nslots && r[7] == rt.off_sidx && r[0] <= ms.max_steps);
      const int32_t *gbase = r + r[4], *stream = r + r[5], *dstl = r + r[6], *sidx = r + r[7];
      std::vector<int> seen((size_t)rt.nslots, 0);
      for (int g = 0; g < ms.ng; g++) {
        int slot = gbase[g];
        int64_t sum = 0;
        for (int st = 0; st < r[0]; st++) {
          const uint32_t w = (uint32_t)stream[(size_t)st * ms.ng + g];
          if ((int32_t)w == N) continue;  // idle, no flags
          const uint32_t row = w & 0x3fffffffu;
          CHECK(!(w & 0x40000000u) && (int)row <= N);
          if ((int)row < N) sum += X[row];  // row N with the last flag: an empty row's single entry
          if (w & 0x80000000u) {
            CHECK(slot >= 0 && slot < rt.nslots && !seen[slot]);
            seen[slot] = 1;
            const int32_t d = dstl[slot];
            if (d < 0) {
              const int ps = d & 0x7fffffff;
              CHECK(ps < ms.nslots && part[ps] == INT64_MIN && sidx[slot] == -1);
              part[ps] = sum;
            } else {
              CHECK(d < f.n_mat && out[d] == INT64_MIN);
              CHECK(sidx[slot] == (sum == 0 && f.mat_ptr[d + 1] == f.mat_ptr[d] ? -1 : f.mat_eid[d]));
              out[d] = sum;
            }
            slot++;
            sum = 0;
          }
        }
        CHECK(sum == 0);
      }
      for (int k = 0; k < rt.nslots; k++) CHECK(seen[k]);
    }
    for (size_t i = 0; i < ms.fixups.size(); i++) {
      const hg::Fixup &fx = ms.fixups[i];
      CHECK(fx.first >= 0 && fx.count >= 1 && fx.first + fx.count <= ms.nslots && fx.row >= 0 && fx.row < f.n_mat);
      int64_t sum = 0;
      for (int k = 0; k < fx.count; k++) {
        CHECK(part[fx.first + k] != INT64_MIN);
        sum += part[fx.first + k];
      }
      if ((int)i < ms.n_fix_l1) {
        CHECK(fx.pad >= 1 && fx.pad <= ms.nslots && part[fx.pad - 1] == INT64_MIN);
        part[fx.pad - 1] = sum;
      } else {
        CHECK(fx.pad == 0 && out[fx.row] == INT64_MIN);
        out[fx.row] = sum;
      }
    }
    for (int i = 0; i < f.n_mat; i++) CHECK(out[i] == mat[i]);
    g_stream_rows += f.n_mat;
    g_stream_chunks += ms.nslots;
  }
  std::vector<int64_t> partial((size_t)f.n_part, INT64_MIN);
  std::vector<int64_t> tile;
  // heavy: if non-null, rows are 24 bits wide and bits 24..29 of a slot's last entry name the heavy
  // hubs the finished sum is added to (hub-pass records)
  auto run_stream = [&](const int32_t *r, int ng, int nslots, int cap, int64_t *heavy) {
    const int steps = r[0], steps_x = r[8];
    const int32_t *gbase = r + r[4], *stream = r + r[5];
    CHECK(0 <= steps_x && steps_x <= steps);
    tile.assign((size_t)nslots, INT64_MIN);
    CHECK(nslots <= cap);
    for (int g = 0; g < ng; g++) {
      int slot = gbase[g];
      int64_t sum = 0;
      for (int s = 0; s < steps; s++) {
        const uint32_t w = (uint32_t)stream[(size_t)s * ng + g];
        // two phases: rows of X, then rows of the materialised table; idle = one past the phase's table
        if ((int32_t)w == (s < steps_x ? N : f.n_mat)) continue;  // contributes zeros, never closes a slot
        CHECK(((w & 0x40000000u) != 0) == (s >= steps_x));
        const uint32_t row = heavy ? (w & 0x00ffffffu) : (w & 0x3fffffffu);
        if (w & 0x40000000u) {
          CHECK((int)row < f.n_mat);
          sum += mat[row];
        } else {
          CHECK((int)row < N);
          sum += X[row];
        }
        if (w & 0x80000000u) {
          CHECK(slot >= 0 && slot < nslots && tile[slot] == INT64_MIN);
          if (heavy)
            for (int h = 0; h < hg::kHubHeavy; h++)
              if ((w >> 24) & (1u << h)) {
                CHECK(h < f.hub.n_heavy);
                heavy[h] += sum;
              }
          tile[slot++] = sum;
          sum = 0;
        } else if (heavy) {
          CHECK(((w >> 24) & 0x3fu) == 0);  // flags only on a slot's last entry
        }
      }
      CHECK(sum == 0);  // a stream never ends inside a slot
    }
    for (int k = 0; k < nslots; k++) CHECK(tile[k] != INT64_MIN);
  };
  // (b) hub pass
  const hg::HubPass &hp = f.hub;
  if (hp.K > 0) {
    CHECK(hp.nwg >= 1 && (int)hp.wg_first.size() == hp.nwg + 1 && hp.wg_first[0] == 0 &&
          hp.wg_first[hp.nwg] == (int)hp.rec_tab.size());
    const int nvr = hp.ng * hp.R;
    CHECK((int)hp.vslot0.size() == nvr && hp.nv <= nvr);
    for (int w = 0; w < hp.nwg; w++) {
      std::vector<int64_t> acc((size_t)nvr, 0);
      int64_t heavy[hg::kHubHeavy] = {0};
      CHECK(hp.wg_first[w] <= hp.wg_first[w + 1] && N < (1 << 24));
      for (int rd = hp.wg_first[w]; rd < hp.wg_first[w + 1]; rd++) {
        const hg::HubRec &rt = hp.rec_tab[rd];
        CHECK(rt.off >= 0 && rt.off + rt.len <= (int64_t)hp.rec.size() && rt.len % 4 == 0 && rt.len <= hp.max_rec_words &&
              rt.len <= 8192);  // what hub_pass_kernel prefetches per round
        const int32_t *r = hp.rec.data() + rt.off;
        CHECK(r[1] == rt.nslots && r[10] == rt.off_eid && r[0] <= hp.max_steps);
        run_stream(r, hp.ng, rt.nslots, hp.cap, heavy);
        for (int k = 0; k < rt.nslots; k++) {  // slot ids vs hyperedge ids (what scaling reads)
          const int e = r[rt.off_eid + k];
          CHECK(e >= -1 && e < M);
          if (e >= 0) CHECK(tile[k] == Xe[e]);
        }
        const uint16_t *pend = reinterpret_cast<const uint16_t *>(r + r[6]);
        const uint16_t *pvs = reinterpret_cast<const uint16_t *>(r + r[9]);
        CHECK(pend[nvr - 1] == r[2] && r[2] <= hp.pair_cap);
        for (int vr = 0; vr < nvr; vr++) {
          const int pb = vr ? pend[vr - 1] : 0, pe = pend[vr];
          CHECK(pb <= pe);
          for (int p = pb; p < pe; p++) {
            CHECK(pvs[p] < rt.nslots && hp.vslot0[vr] >= 0);
            acc[vr] += tile[pvs[p]];
          }
        }
      }
      for (int vr = 0; vr < nvr; vr++)
        if (hp.vslot0[vr] >= 0) {
          const int s = hp.vslot0[vr] + w;
          CHECK(s < f.n_part && partial[s] == INT64_MIN);
          partial[s] = acc[vr];
        }
      CHECK(hp.n_heavy <= hg::kHubHeavy && hp.n_heavy <= hp.K && hp.cap >= hp.ng);
      for (int h = 0; h < hp.n_heavy; h++) {
        const int s = hp.hslot0[h] + w;
        CHECK(hp.hslot0[h] >= 0 && s < f.n_part && partial[s] == INT64_MIN);
        partial[s] = heavy[h];
      }
    }
  }
  // (c) panels
  std::vector<int> written((size_t)N, 0);
  std::vector<int64_t> Y((size_t)N, 0);
  for (size_t i = 0; i < f.panels.size(); i++) {
    const auto &pn = f.panels[i];
    const auto &rt = f.rec_tab[i];
    CHECK(pn.nrows >= 1 && pn.nrows <= f.rows_cap && pn.nslots <= f.cap && pn.npm <= f.mem_cap && pn.nvs <= f.vslot_cap);
    CHECK(rt.off >= 0 && rt.off + rt.len <= (int64_t)f.rec.size() && rt.len % 4 == 0 && rt.len <= f.max_rec_words);
    const int32_t *r = f.rec.data() + rt.off;
    CHECK(r[1] == pn.nrows && r[2] == pn.nslots && rt.nrows == pn.nrows && rt.off_prow == r[7]);
    run_stream(r, f.ng, pn.nslots, f.cap, nullptr);
    const uint16_t *pend = reinterpret_cast<const uint16_t *>(r + r[6]);
    const uint16_t *pvs = reinterpret_cast<const uint16_t *>(r + r[9]);
    const int32_t *prow = r + r[7];
    for (int k = 0; k < pn.nrows; k++) {
      const int pb = k ? pend[k - 1] : 0, pe = pend[k];
      int64_t sum = 0;
      for (int p = pb; p < pe; p++) {
        CHECK(pvs[p] < pn.nslots);
        sum += tile[pvs[p]];
      }
      if (prow[k] < 0) {
        const int s = prow[k] & 0x7fffffff;
        CHECK(s < f.n_part && partial[s] == INT64_MIN);
        partial[s] = sum;
      } else {
        CHECK(prow[k] < N);
        written[prow[k]]++;
        Y[prow[k]] = sum;
      }
    }
  }
  // (d) fixups: first level into slots of their own, then the final ones into Y
  CHECK(f.n_fix_l1 <= (int)f.fixups.size());
  for (size_t i = 0; i < f.fixups.size(); i++) {
    const hg::Fixup &fx = f.fixups[i];
    CHECK(fx.first >= 0 && fx.count >= 1 && fx.first + fx.count <= f.n_part && fx.row >= 0 && fx.row < N);
    int64_t sum = 0;
    for (int k = 0; k < fx.count; k++) {
      CHECK(partial[fx.first + k] != INT64_MIN);
      sum += partial[fx.first + k];
    }
    if ((int)i < f.n_fix_l1) {
      CHECK(fx.pad >= 1 && fx.pad <= f.n_part && partial[fx.pad - 1] == INT64_MIN);
      partial[fx.pad - 1] = sum;
    } else {
      CHECK(fx.pad == 0);
      written[fx.row]++;
      Y[fx.row] = sum;
    }
  }
  for (int v = 0; v < N; v++) CHECK(written[v] == 1 && Y[v] == want[v]);
  g_hub_graphs += hp.K > 0;
  g_hub_rounds += (long)hp.rec_tab.size();
  g_heavy += hp.n_heavy;
  g_hub_parts += hp.nv - hp.K;
  g_split_rows += f.n_split;
  g_l1_fixups += f.n_fix_l1;
}

static void one_graph(std::mt19937 &rng, int N, int M, double mean, int hub_every, bool mega, bool dups) {
  std::vector<int32_t> ptr(1, 0), ind;
  std::geometric_distribution<int> size(1.0 / (mean + 1.0));
  std::uniform_int_distribution<int> vert(0, N - 1);
  for (int e = 0; e < M; e++) {
    int s = std::min(size(rng), N);
    if (mega && e == 0) s = N;  // one hyperedge holds every vertex
    std::vector<char> seen((size_t)N, 0);
    for (int k = 0; k < s; k++) {
      int v = mega && e == 0 ? k : vert(rng);
      if (hub_every && k == 0 && e % hub_every) v = 0;  // vertex 0 is a hub
      // repeated members stay in for about one draw in ten, and always for the forced hub: validate_csr and the
      // MatrixMarket reader accept duplicate incidences, and every path must count them with multiplicity
      const bool keep_dup = dups && (v == 0 || rng() % 10 == 0);
      if (!seen[v] || keep_dup) {
        seen[v] = 1;
        ind.push_back(v);
      }
    }
    ptr.push_back((int32_t)ind.size());
  }
  CHECK(hg::validate_csr(M, N, ptr.data(), ind.data()) == HG_OK);
  std::vector<int32_t> ptr_v, ind_v;
  hg::transpose_csr(M, N, ptr.data(), ind.data(), ptr_v, ind_v);
  CHECK((int)ptr_v.size() == N + 1 && ptr_v.back() == ptr.back());
  for (int variant = 0; variant < 3; variant++) {
    hg::Opts o;
    if (variant == 1) {
      o.short_max = 4;
      o.split_len = 4;
      o.panel_rows = 3;
      o.panel_nnz = 8;
      o.panel_rows_auto = false;
      o.t_big = 2;
      o.flags = HG_PLAN_DFS_ORDER;
    } else if (variant == 2) {
      o.t_big = 64;
    }
    for (int hop = 0; hop < 2; hop++) {
      hg::Sched s;
      const int nrows = hop ? N : M;
      const int32_t *p = hop ? ptr_v.data() : ptr.data();
      hg::build_sched(nrows, p, o, s);
      std::vector<int> covered((size_t)nrows, 0);
      for (const auto &pn : s.panels)
        for (int r = pn.row0; r < pn.row0 + pn.nrows; r++) covered[r]++;
      std::vector<int64_t> task_len((size_t)nrows, 0);
      for (const auto &t : s.tasks) {
        CHECK(t.row >= 0 && t.row < nrows && p[t.row] <= t.beg && t.beg < t.end && t.end <= p[t.row + 1]);
        task_len[t.row] += t.end - t.beg;
        CHECK(t.slot < s.nslots);
      }
      for (int r = 0; r < nrows; r++) {
        const int len = p[r + 1] - p[r];
        if (len > o.short_max) CHECK(covered[r] == 0 && task_len[r] == len);
        else CHECK(covered[r] == 1 && task_len[r] == 0);
      }
      for (const auto &fx : s.fixups) CHECK(fx.first >= 0 && fx.count >= 1 && fx.first + fx.count <= s.nslots && fx.pad <= s.nslots);
    }
    for (int cap : {16, 64, 128}) {
      for (int hubs = 0; hubs < 3; hubs++) {
        hg::Opts oh = o;
        if (hubs == 2) {  // launch-bound schedules: nothing materialised, long hyperedges cut into sub-slots
          oh.flags |= HG_PLAN_NO_HUB_PASS;
          oh.slot_chunk = variant == 1 ? 2 : 8;
        } else if (hubs) {  // reach the hub pass on these small graphs
          oh.hub_min_nnz = 0;
          oh.hub_min_deg = 2;
          if (cap != 64) oh.hub_tile_bytes = 16384;  // 32-slot rounds: many rounds, many workgroups
        } else {
          oh.flags |= HG_PLAN_NO_HUB_PASS;
        }
        hg::FusedSched f;
        // 8 lanes x 4 floats per row, as F = 32; or (small hub tiles) 32 lanes x 4 floats, as F = 128
        // ... or (sub-slot schedules, 128-slot tiles) the 128 lane groups of a 1024-thread panel for tiny dense graphs
        const int ng = (hubs == 2 && cap == 128) ? 128 : (hubs && cap != 64) ? 8 : 32, row_floats = (hubs && cap != 64 && ng != 128) ? 128 : 32;
        // every third graph: the linear epilogue's panel shape -- fewer rows than slots (rows capped at 2/3 of the slots,
        // whole 16-row tiles where the cap allows)
        const int rows_cap = (N % 3 == 0 && hubs != 2) ? std::max(1, cap * 2 / 3 / 16 * 16) : 0;
        hg::build_fused(N, M, ptr.data(), ind.data(), ptr_v.data(), ind_v.data(), oh, cap, cap * 4, ng, row_floats, true, f, rows_cap);
        if (rows_cap) {
          CHECK(f.rows_cap == std::min(rows_cap, cap));
          for (const auto &pn : f.panels) CHECK(pn.nrows <= f.rows_cap);
          g_rows_capped++;
        }
        if (hubs == 2) {
          if (f.invalid) continue;  // a vertex's sub-slots exceed a panel: the plan discards such a schedule
          CHECK(f.n_mat == 0 && f.n_split == 0 && f.fixups.empty());
          emulate_fused(f, N, M, ptr, ind, ptr_v, ind_v, rng);
          g_chunked++;
          continue;
        }
        int64_t n_mat = 0, n_big = 0;
        hg::classify_fused(N, M, ptr.data(), ptr_v.data(), ind_v.data(), oh, cap, cap * 4, &n_mat, &n_big);
        CHECK(n_mat == f.n_mat && n_big == f.n_split + f.hub.K);
        CHECK(f.rec_tab.size() == f.panels.size());
        emulate_fused(f, N, M, ptr, ind, ptr_v, ind_v, rng);
      }
    }
  }
}

// hg_balance_schedule on random row pointers (size query, then fill) and hg_mtx_read on
// well-formed, odd and broken MatrixMarket text.
static void balance_and_mtx(std::mt19937 &rng, const char *tmp_path) {
  for (int it = 0; it < 60; it++) {
    const int nrow = (int)(rng() % 200);
    std::vector<int32_t> ptr(1, 0);
    for (int r = 0; r < nrow; r++) ptr.push_back(ptr.back() + (int32_t)(rng() % 4 == 0 ? 0 : rng() % 40));
    const int ngs = 1 + (int)(rng() % 12);
    int64_t nk = 0, ng = 0;
    CHECK(hg_balance_schedule(nrow, ngs, ptr.data(), &nk, &ng, nullptr, nullptr, nullptr, nullptr) == HG_OK);
    std::vector<int32_t> key((size_t)nk + 1), row((size_t)ng + 1), st((size_t)ng + 1), ed((size_t)ng + 1);
    int64_t nk2 = nk, ng2 = ng;
    CHECK(hg_balance_schedule(nrow, ngs, ptr.data(), &nk2, &ng2, key.data(), row.data(), st.data(), ed.data()) == HG_OK);
    CHECK(nk2 == nk && ng2 == ng);
    for (int64_t g = 0; g < ng; g++) CHECK(row[g] >= 0 && row[g] < nrow && st[g] >= 0 && st[g] + 1 < nk && ed[g] >= 0 && ed[g] + 1 < nk);
  }
  const char *texts[] = {
      "%%MatrixMarket matrix coordinate real general\n% comment\n3 2 4\n1 1 1.0\n2 1 1.0\n3 2 1.0\n1 2 1.0\n",
      "%%MatrixMarket matrix coordinate pattern symmetric\n3 3 2\n2 1\n3 3\n",
      "%%MatrixMarket matrix coordinate real general\n3 2 2\n1 1 1.0\n",          // fewer entries than declared
      "%%MatrixMarket matrix coordinate real general\n3 2 1\n9 1 1.0\n",          // row out of range
      "%%MatrixMarket matrix coordinate real general\n-3 2 1\n1 1 1.0\n",         // negative size
      "%%MatrixMarket matrix coordinate real general\n",                           // no size line
      "garbage\n1 2 3\n",
      "",
  };
  for (const char *t : texts) {
    FILE *fp = std::fopen(tmp_path, "w");
    CHECK(fp != nullptr);
    std::fputs(t, fp);
    std::fclose(fp);
    int32_t nr = 0, nc = 0, *hp = nullptr, *hi = nullptr, *tp = nullptr, *ti = nullptr;
    int64_t nnz = 0;
    const int rc = hg_mtx_read(tmp_path, &nr, &nc, &nnz, &hp, &hi, &tp, &ti);
    if (rc == HG_OK) {
      CHECK(hp && tp && hp[nr] == nnz && tp[nc] == nnz);
      for (int64_t p = 0; p < nnz; p++) CHECK(hi[p] >= 0 && hi[p] < nc && ti[p] >= 0 && ti[p] < nr);
      hg_free(hp);
      hg_free(hi);
      hg_free(tp);
      hg_free(ti);
    }
  }
  CHECK(hg_mtx_read("/nonexistent/file.mtx", nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != HG_OK);
}

int main(int argc, char **argv) {
  std::mt19937 rng(12345);
  balance_and_mtx(rng, argc > 1 ? argv[1] : "/tmp/sched_fuzz.mtx");
  for (int it = 0; it < 240; it++) {
    const int N = 1 + (int)(rng() % (it % 20 == 0 ? 4000 : 400)), M = (int)(rng() % (it % 20 == 0 ? 3000 : 300));
    one_graph(rng, N, M, 0.5 + (it % 7) * 2.0, it % 3 == 0 ? 2 : 0, it % 11 == 5, it % 2 == 1 || it % 6 == 0);
  }
  // the random graphs must actually have reached the hub pass, hub parts, split rows and two-level fixups
  std::printf("hub schedules %ld, hub rounds %ld, heavy hubs %ld, extra hub parts %ld, split vertices %ld, first-level fixups %ld\n",
              g_hub_graphs, g_hub_rounds, g_heavy, g_hub_parts, g_split_rows, g_l1_fixups);
  CHECK(g_rows_capped > 50);
  CHECK(g_stream_rows > 10000 && g_stream_chunks > 100 && g_chunked > 200 && g_heavy > 50 && g_hub_graphs > 50 && g_hub_rounds > 10 * g_hub_graphs && g_hub_parts > 0 && g_split_rows > 100 && g_l1_fixups > 0);
  std::puts("sched_fuzz ok");
  return 0;
}
