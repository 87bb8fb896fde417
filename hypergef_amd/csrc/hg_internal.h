// Internal structures of libhgaggr (host side).  Not part of the C ABI.
#pragma once
#include <cstdint>
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
struct Fixup {
  int32_t row, first, count, pad;
};

// Schedule of one CSR matrix for the gather-rows kernel.
struct Sched {
  int32_t nrows = 0;
  int32_t max_len = 0;
  std::vector<Panel> panels;
  std::vector<Task> tasks;
  std::vector<Fixup> fixups;
  int32_t nslots = 0;
  // device copies
  Panel *d_panels = nullptr;
  Task *d_tasks = nullptr;
  Fixup *d_fixups = nullptr;
};

struct Opts {
  int32_t short_max = 32;
  int32_t split_len = 512;
  int32_t panel_rows = 128;
  int32_t panel_nnz = 1024;
  int32_t flags = 0;
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

}  // namespace hg

struct hg_plan {
  int32_t N = 0, M = 0;
  int64_t nnz = 0;
  hg::Opts opts;
  std::vector<int32_t> ptr_v, ind_v;  // H CSR (vertex -> hyperedges), host
  int32_t *d_ptr_v = nullptr, *d_ind_v = nullptr;
  hg::Sched sched[2];  // [0]: H_T rows = hyperedges, [1]: H rows = vertices
  int64_t device_bytes = 0;
  int device = -1;
};
