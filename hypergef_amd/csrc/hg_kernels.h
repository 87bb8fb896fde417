// Kernel argument blocks and launchers shared by hg_kernels.hip and hg_api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hg_internal.h"

namespace hg {

struct GatherArgs {
  const int32_t *ptr;   // CSR row pointers [nrows + 1]
  const int32_t *ind;   // CSR column indices
  const float *src;     // gathered table [*, F]
  float *dst;           // output [nrows, F]
  const float *scaleA;  // per-row factor applied first, or null
  const float *scaleB;  // per-row factor applied second, or null
  float *partial;       // partial-sum slots [nslots, F]
  const Panel *panels;
  const Task *tasks;
  int32_t npanels, ntasks, n_task_blocks;
  int32_t F;
  int32_t panel_rows, panel_nnz;  // LDS carve-up
  int32_t xcd_remap;
};

struct PushArgs {
  int64_t n_group;
  const int32_t *group_key, *group_row, *group_st, *group_ed;
  const int32_t *csrptr_t, *colind_t;
  const float *X, *degE, *degV, *W;
  float *Y;
  int32_t F;
};

hipError_t launch_gather(const GatherArgs &a, int nfix, const Fixup *fixups, bool vec4,
                         hipStream_t stream);
hipError_t launch_push(const PushArgs &a, hipStream_t stream);

}  // namespace hg
