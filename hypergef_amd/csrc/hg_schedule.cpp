// Host-side scheduling for libhgaggr: the reference-compatible edge-group
// balancer and this backend's own wave64 schedule (row panels + wave tasks).
// Pure C++ (no HIP calls) so it is unit-tested on a machine without a GPU.
#include <algorithm>
#include <cstring>

#include "hg_internal.h"

namespace hg {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }
const char *last_error() { return g_last_error.c_str(); }

// Same contract as balance_schedule.balancer (HyperGsys/balancer.py:15-33) and
// hgnn_ef_full_balance_cpu (include/taskbalancer/balancer_kernel.cuh:229-259).
// Built directly in int32 (the reference round-trips through float32 tensors,
// HyperGsys/hypergraph.py:98-101, which corrupts indices above 2^24).
int balance_schedule(int32_t nrow, int32_t ngs, const int32_t *csrptr,
                     int64_t *n_key, int64_t *n_group, int32_t *key,
                     int32_t *row, int32_t *st, int32_t *ed) {
  if (nrow < 0 || ngs <= 0 || !csrptr || !n_key || !n_group) {
    set_error("hg_balance_schedule: bad argument");
    return HG_ERR_INVALID;
  }
  const bool fill = key != nullptr;
  if (fill && (!row || !st || !ed)) {
    set_error("hg_balance_schedule: key given without row/group_st/group_ed");
    return HG_ERR_INVALID;
  }
  int64_t nk = 0, ng = 0, parts = 0;
  int64_t last_key = -1;
  for (int32_t r = 0; r < nrow; r++) {
    const int64_t lb = csrptr[r], hb = csrptr[r + 1];
    if (hb < lb) {
      set_error("hg_balance_schedule: csrptr not monotone");
      return HG_ERR_INVALID;
    }
    const int64_t w = (hb - lb + ngs - 1) / ngs;
    for (int64_t k = lb; k < hb; k += ngs) {
      if (fill) key[nk] = (int32_t)k;
      last_key = k;
      nk++;
    }
    if (fill) {
      for (int64_t i = 0; i < w; i++)
        for (int64_t j = 0; j < w; j++) {
          st[ng] = (int32_t)(parts + j);
          ed[ng] = (int32_t)(parts + i);
          row[ng] = r;
          ng++;
        }
    } else {
      ng += w * w;
    }
    parts += w;
  }
  if (nk == 0) {
    set_error("hg_balance_schedule: incidence matrix has no entries");
    return HG_ERR_INVALID;
  }
  if (last_key != csrptr[nrow]) {
    if (fill) key[nk] = csrptr[nrow];
    nk++;
  }
  if (fill && (nk != *n_key || ng != *n_group)) {
    set_error("hg_balance_schedule: output arrays sized for a different schedule");
    return HG_ERR_INVALID;
  }
  *n_key = nk;
  *n_group = ng;
  return HG_OK;
}

int validate_csr(int32_t nrows, int32_t ncols, const int32_t *ptr,
                 const int32_t *ind) {
  if (nrows < 0 || ncols < 0 || !ptr) {
    set_error("CSR: bad dimensions or null row pointer");
    return HG_ERR_INVALID;
  }
  if (ptr[0] != 0) {
    set_error("CSR: csrptr[0] != 0");
    return HG_ERR_INVALID;
  }
  for (int32_t r = 0; r < nrows; r++)
    if (ptr[r + 1] < ptr[r]) {
      set_error("CSR: csrptr not monotone at row " + std::to_string(r));
      return HG_ERR_INVALID;
    }
  const int64_t nnz = ptr[nrows];
  if (nnz > 0 && !ind) {
    set_error("CSR: null column index array");
    return HG_ERR_INVALID;
  }
  for (int64_t p = 0; p < nnz; p++)
    if (ind[p] < 0 || ind[p] >= ncols) {
      set_error("CSR: column index out of range at entry " + std::to_string(p));
      return HG_ERR_INVALID;
    }
  return HG_OK;
}

// Stable counting sort by column: row ids stay ascending inside each output
// row, so hop 2 visits a vertex's hyperedges in the order the reference CPU
// path does (H CSR order, include/util/check.cuh:95-107).
void transpose_csr(int32_t nrows, int32_t ncols, const int32_t *ptr,
                   const int32_t *ind, std::vector<int32_t> &t_ptr,
                   std::vector<int32_t> &t_ind) {
  const int64_t nnz = ptr[nrows];
  t_ptr.assign((size_t)ncols + 1, 0);
  t_ind.resize((size_t)nnz);
  for (int64_t p = 0; p < nnz; p++) t_ptr[(size_t)ind[p] + 1]++;
  for (int32_t c = 0; c < ncols; c++) t_ptr[c + 1] += t_ptr[c];
  std::vector<int32_t> cursor(t_ptr.begin(), t_ptr.end() - 1);
  for (int32_t r = 0; r < nrows; r++)
    for (int32_t p = ptr[r]; p < ptr[r + 1]; p++) t_ind[cursor[ind[p]]++] = r;
}

// Row panels for short rows, wave tasks for long ones.
//  * a panel is a run of consecutive rows, none longer than short_max, with at
//    most panel_rows rows and panel_nnz index entries (what one workgroup
//    stages in LDS); its entries are one contiguous slice of the index array;
//  * a row longer than short_max ends the running panel and becomes
//    ceil(len/split_len) wave tasks; with more than one task the row gets
//    partial-sum slots and a fixup record (summed in slot order, so the result
//    does not depend on scheduling).  A row cut into more than kFixupFan tasks is
//    summed in two levels: ~sqrt(tasks) first-level fixups, each adding a run of
//    slots into a slot of its own (Fixup::pad = that slot + 1), then the final
//    fixup over those -- a hub row of 10^6 entries is not one serial chain of
//    thousands of dependent loads.  First-level fixups come first in the list.
// out[row] = sum of partial[first .. first + count): one final fixup, or -- above kFixupFan slots --
// ~sqrt(count) first-level fixups into slots of their own (allocated from nslots) and a final one.
void add_fixups(int32_t row, int32_t first, int32_t count, int32_t &nslots, std::vector<Fixup> &level1,
                std::vector<Fixup> &finals) {
  constexpr int32_t kFixupFan = 32;
  if (count <= kFixupFan) {
    finals.push_back(Fixup{row, first, count, 0});
    return;
  }
  int32_t fan = 1;
  while (fan * fan < count) fan++;
  const int32_t groups = (count + fan - 1) / fan;
  finals.push_back(Fixup{row, nslots, groups, 0});
  for (int32_t g = 0; g < groups; g++)
    level1.push_back(Fixup{row, first + g * fan, std::min(fan, count - g * fan), ++nslots});
}

void build_sched(int32_t nrows, const int32_t *ptr, const Opts &o, Sched &s) {
  s.nrows = nrows;
  s.max_len = 0;
  s.panels.clear();
  s.tasks.clear();
  s.fixups.clear();
  s.nslots = 0;
  s.n_fix_l1 = 0;
  std::vector<Fixup> finals;
  // A small schedule is latency-bound: a workgroup walks its panel's rows in 8..32 lane groups,
  // one dependent round of row loads after another, so 128-row panels turn a 2000-row graph into
  // 16 workgroups doing 16 rounds each (22 us per hop on a 2012-row kNN hypergraph at F = 128).
  // Unless the caller fixed the size, panels shrink until there are about a thousand of them.
  int32_t panel_rows = o.panel_rows;
  if (o.panel_rows_auto) panel_rows = std::max(8, std::min(o.panel_rows, (nrows + 1023) / 1024));
  int32_t start = 0;
  auto close = [&](int32_t end_row) {
    if (end_row > start) {
      Panel p;
      p.row0 = start;
      p.nrows = end_row - start;
      p.nnz0 = ptr[start];
      p.nnz_cnt = ptr[end_row] - ptr[start];
      s.panels.push_back(p);
    }
  };
  for (int32_t r = 0; r < nrows; r++) {
    const int32_t len = ptr[r + 1] - ptr[r];
    s.max_len = std::max(s.max_len, len);
    if (len > o.short_max) {
      close(r);
      start = r + 1;
      const int32_t chunks = (len + o.split_len - 1) / o.split_len;
      if (chunks == 1) {
        s.tasks.push_back(Task{r, ptr[r], ptr[r + 1], -1});
      } else {
        const int32_t first = s.nslots;
        for (int32_t c = 0; c < chunks; c++) {
          const int32_t b = ptr[r] + c * o.split_len;
          const int32_t e = std::min(b + o.split_len, ptr[r + 1]);
          s.tasks.push_back(Task{r, b, e, s.nslots++});
        }
        add_fixups(r, first, chunks, s.nslots, s.fixups, finals);
      }
    } else if (r - start == panel_rows ||
               ptr[r + 1] - ptr[start] > o.panel_nnz) {
      close(r);
      start = r;
    }
  }
  close(nrows);
  s.n_fix_l1 = (int32_t)s.fixups.size();
  s.fixups.insert(s.fixups.end(), finals.begin(), finals.end());
  // longest tasks first: the tail of the launch is made of short work
  std::stable_sort(s.tasks.begin(), s.tasks.end(), [](const Task &a, const Task &b) {
    return (a.end - a.beg) > (b.end - b.beg);
  });
}

}  // namespace hg

extern "C" {

const char *hg_last_error(void) { return hg::last_error(); }

int hg_version(void) { return HG_AGGR_VERSION; }

const char *hg_status_string(int status) {
  switch (status) {
    case HG_OK: return "ok";
    case HG_ERR_INVALID: return "invalid argument";
    case HG_ERR_NOMEM: return "out of memory";
    case HG_ERR_HIP: return "HIP runtime error";
    case HG_ERR_WORKSPACE: return "workspace too small";
    case HG_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
  }
}

int hg_balance_schedule(int32_t nrow, int32_t ngs, const int32_t *csrptr_host,
                        int64_t *n_key, int64_t *n_group, int32_t *key,
                        int32_t *row, int32_t *group_st, int32_t *group_ed) {
  return hg::balance_schedule(nrow, ngs, csrptr_host, n_key, n_group, key, row,
                              group_st, group_ed);
}

}  // extern "C"
