// Host-side schedule builders under AddressSanitizer / UBSan (CPU only; tests/test_host.py
// compiles and runs this).  Random hypergraphs of many shapes -- empty rows, hubs, duplicates,
// one hyperedge holding everything -- through transpose_csr, build_sched and build_fused (both
// row orders, several capacities); checks the structural invariants the kernels rely on.
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

static void one_graph(std::mt19937 &rng, int N, int M, double mean, int hub_every, bool mega) {
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
      if (!seen[v]) {
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
      hg::FusedSched f;
      hg::build_fused(N, M, ptr.data(), ind.data(), ptr_v.data(), ind_v.data(), o, cap, cap * 4, 32, f);
      std::vector<int> seen_v((size_t)N, 0);
      for (int32_t v : f.prow) seen_v[v]++;
      for (int32_t v : f.hub_vid) seen_v[v]++;
      for (int v = 0; v < N; v++) CHECK(seen_v[v] == 1);  // every vertex exactly once
      int64_t n_mat = 0, n_hub = 0;
      hg::classify_fused(N, M, ptr.data(), ptr_v.data(), ind_v.data(), o, cap, cap * 4, &n_mat, &n_hub);
      CHECK(n_mat == f.n_mat && n_hub == f.n_hub);
      CHECK(f.rec_tab.size() == f.panels.size());
      for (size_t i = 0; i < f.panels.size(); i++) {
        const auto &pn = f.panels[i];
        const auto &rt = f.rec_tab[i];
        CHECK(pn.nrows >= 1 && pn.nrows <= f.rows_cap && pn.nslots <= f.cap && pn.npm <= f.mem_cap);
        CHECK(rt.off >= 0 && rt.off + rt.len <= (int64_t)f.rec.size() && rt.len % 4 == 0 && rt.len <= f.max_rec_words);
        const int32_t *r = f.rec.data() + rt.off;
        CHECK(r[1] == pn.nrows && r[2] == pn.nslots);
        const int steps = r[0];
        for (int k = 0; k < steps * f.ng; k++) {
          const uint32_t w = (uint32_t)r[r[5] + k];
          if ((int32_t)w == N) continue;  // idle
          const uint32_t row = w & 0x3fffffffu;
          CHECK((w & 0x40000000u) ? (int)row < f.n_mat : (int)row < N);
        }
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
    one_graph(rng, N, M, 0.5 + (it % 7) * 2.0, it % 3 == 0 ? 2 : 0, it % 11 == 5);
  }
  std::puts("sched_fuzz ok");
  return 0;
}
