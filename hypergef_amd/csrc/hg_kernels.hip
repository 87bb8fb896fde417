// gfx950 (MI355X, CDNA4) kernels of libhgaggr: the two hops of the fused
// vertex -> hyperedge -> vertex aggregation as atomic-free row gathers, plus the
// reference-style register-fused push kernel with fp32 atomics.
//
// Layout: a feature row of F floats is spread over LPR consecutive lanes (VEC
// floats each), so a 64-lane wavefront holds G = 64/LPR rows at once and every
// row access is one contiguous LPR*VEC*4-byte segment (F = 32: 8 lanes x
// dwordx4 = one 128-byte line per row, 8 rows per wave instruction).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "hg_kernels.h"

namespace hg {

template <int VEC> struct Vec;
template <> struct Vec<1> {
  float x;
  __device__ __forceinline__ static Vec zero() { return Vec{0.f}; }
  __device__ __forceinline__ static Vec load(const float *p) { return Vec{*p}; }
  __device__ __forceinline__ void store(float *p) const { *p = x; }
  __device__ __forceinline__ void store_nt(float *p) const { __builtin_nontemporal_store(x, p); }
  __device__ __forceinline__ void add(const Vec &o) { x += o.x; }
  __device__ __forceinline__ void mul(float s) { x *= s; }
  __device__ __forceinline__ void xor_reduce(int off) { x += __shfl_xor(x, off, 64); }
};
template <> struct Vec<4> {
  float4 v;
  __device__ __forceinline__ static Vec zero() { return Vec{make_float4(0.f, 0.f, 0.f, 0.f)}; }
  __device__ __forceinline__ static Vec load(const float *p) {
    return Vec{*reinterpret_cast<const float4 *>(p)};
  }
  __device__ __forceinline__ void store(float *p) const { *reinterpret_cast<float4 *>(p) = v; }
  __device__ __forceinline__ void store_nt(float *p) const {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4 *>(p));
  }
  __device__ __forceinline__ void add(const Vec &o) {
    v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w;
  }
  __device__ __forceinline__ void mul(float s) { v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
  __device__ __forceinline__ void xor_reduce(int off) {
    v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64);
    v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
  }
};

// dst[r,:] = scaleB[r] * (scaleA[r] * sum_{p in row r} src[ind[p],:])
//
// Workgroups [0, n_task_blocks) run wave tasks (one long-row slice per wave, its
// entries strided over the wave's G row groups, then a cross-group shuffle
// reduction).  The remaining workgroups each own one row panel: the panel's
// row pointers, row scales and index slice are staged into LDS with coalesced
// loads, then every LPR-lane group walks a contiguous run of the panel's rows as
// one flat entry stream, U row loads in flight, adding in CSR order (so short
// rows reproduce the CPU reference's summation order exactly).
template <int LPR, int VEC, int U, bool PIPE>
__global__ __launch_bounds__(256) void gather_rows_kernel(const GatherArgs a) {
  constexpr int G = 64 / LPR;    // row groups per wave
  constexpr int NG = 256 / LPR;  // row groups per workgroup
  using V = Vec<VEC>;
  extern __shared__ int32_t smem[];

  const int tid = threadIdx.x;
  const int gl = tid & (LPR - 1);
  const int col = (blockIdx.y * LPR + gl) * VEC;
  const bool col_ok = col < a.F;
  const int64_t F = a.F;
  int b = blockIdx.x;

  if (b < a.n_task_blocks) {
    const int t = __builtin_amdgcn_readfirstlane(b * 4 + (tid >> 6));
    if (t >= a.ntasks) return;
    const Task tk = a.tasks[t];
    const int g = (tid & 63) / LPR;
    V acc = V::zero();
    for (int p = tk.beg + g; p < tk.end; p += G * U) {
      V v[U];
#pragma unroll
      for (int k = 0; k < U; k++) {
        const int q = p + k * G;
        const bool ok = col_ok && q < tk.end;
        const int64_t idx = ok ? a.ind[q] : 0;
        v[k] = ok ? V::load(a.src + idx * F + col) : V::zero();
      }
#pragma unroll
      for (int k = 0; k < U; k++) acc.add(v[k]);
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) acc.xor_reduce(off);
    if (g == 0 && col_ok) {
      if (tk.slot < 0) {
        const int srow = a.scale_map ? a.scale_map[tk.row] : tk.row;
        const int64_t drow = a.dst_map ? a.dst_map[tk.row] : tk.row;
        if (a.scaleA) acc.mul(a.scaleA[srow]);
        if (a.scaleB) acc.mul(a.scaleB[srow]);
        acc.store(a.dst + drow * F + col);
      } else {
        acc.store(a.partial + (int64_t)tk.slot * F + col);
      }
    }
    return;
  }

  b -= a.n_task_blocks;
  if (a.xcd_remap) {
    // workgroups are dealt round-robin to the 8 XCDs; give each XCD one
    // contiguous run of panels so neighbouring panels share an L2
    const int x = b & 7, i = b >> 3;
    const int cpx = a.npanels >> 3, rem = a.npanels & 7;
    b = x * cpx + (x < rem ? x : rem) + i;
  }
  const Panel pn = a.panels[b];
  int32_t *sptr = smem;                                         // [panel_rows + 1]
  float *sA = reinterpret_cast<float *>(smem + a.panel_rows + 1);  // [panel_rows]
  float *sB = sA + a.panel_rows;                                // [panel_rows]
  int32_t *sdst = reinterpret_cast<int32_t *>(sB + a.panel_rows);  // [panel_rows]
  int32_t *sind = sdst + a.panel_rows;                             // [panel_nnz]

  for (int i = tid; i <= pn.nrows; i += 256) sptr[i] = a.ptr[pn.row0 + i] - pn.nnz0;
  for (int i = tid; i < pn.nrows; i += 256) {
    const int srow = a.scale_map ? a.scale_map[pn.row0 + i] : pn.row0 + i;
    if (a.scaleA) sA[i] = a.scaleA[srow];
    if (a.scaleB) sB[i] = a.scaleB[srow];
    if (a.dst_map) sdst[i] = a.dst_map[pn.row0 + i];
  }
  for (int i = tid; i < pn.nnz_cnt; i += 256) sind[i] = a.ind[pn.nnz0 + i];
  __syncthreads();

  const int g = tid / LPR;
  const int rpg = (pn.nrows + NG - 1) / NG;
  int r = min(g * rpg, pn.nrows);
  const int re = min(r + rpg, pn.nrows);
  if (r >= re) return;

  auto flush = [&](int row, V acc) {
    if (sptr[row + 1] > sptr[row]) {  // an empty row stays exactly 0 (degE may be inf)
      if (a.scaleA) acc.mul(sA[row]);
      if (a.scaleB) acc.mul(sB[row]);
    }
    const int64_t drow = a.dst_map ? sdst[row] : pn.row0 + row;
    if (col_ok) acc.store(a.dst + drow * F + col);
  };

  int pos = sptr[r];
  const int stop = sptr[re];
  int row_end = sptr[r + 1];
  V acc = V::zero();

  auto issue = [&](int p0, int n, V(&v)[U]) {
#pragma unroll
    for (int k = 0; k < U; k++) {
      const int64_t idx = sind[p0 + min(k, n - 1)];
      v[k] = col_ok ? V::load(a.src + idx * F + col) : V::zero();
    }
  };
  auto consume = [&](int p0, int n, const V(&v)[U]) {
#pragma unroll
    for (int k = 0; k < U; k++) {
      if (k < n) {
        while (row_end <= p0 + k) {
          flush(r, acc);
          acc = V::zero();
          r++;
          row_end = sptr[r + 1];
        }
        acc.add(v[k]);
      }
    }
  };

  if constexpr (PIPE) {
    // two batches in flight: the next batch's loads are issued before the
    // current one is consumed, so the wave always has row loads outstanding
    V cur[U];
    int ncur = min(U, stop - pos);
    if (ncur > 0) issue(pos, ncur, cur);
    while (ncur > 0) {
      V nxt[U];
      const int nn = min(U, stop - pos - ncur);
      if (nn > 0) issue(pos + ncur, nn, nxt);
      consume(pos, ncur, cur);
      pos += ncur;
      ncur = nn;
#pragma unroll
      for (int k = 0; k < U; k++) cur[k] = nxt[k];
    }
  } else {
    while (pos < stop) {
      const int n = min(U, stop - pos);
      V v[U];
      issue(pos, n, v);
      consume(pos, n, v);
      pos += n;
    }
  }
  while (r < re) {
    flush(r, acc);
    acc = V::zero();
    r++;
  }
}

// out[row,:] = scaleB * (scaleA * sum_k partial[first+k,:]), slots in order.
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void fixup_rows_kernel(const GatherArgs a, const Fixup *fixups,
                                                         int nfix) {
  using V = Vec<VEC>;
  const int tid = threadIdx.x;
  const int gl = tid & (LPR - 1);
  const int col = (blockIdx.y * LPR + gl) * VEC;
  const int f = blockIdx.x * (256 / LPR) + tid / LPR;
  if (f >= nfix || col >= a.F) return;
  const Fixup fx = fixups[f];
  const int64_t F = a.F;
  V acc = V::zero();
  for (int k = 0; k < fx.count; k++) acc.add(V::load(a.partial + (int64_t)(fx.first + k) * F + col));
  const int srow = a.scale_map ? a.scale_map[fx.row] : fx.row;
  const int64_t drow = a.dst_map ? a.dst_map[fx.row] : fx.row;
  if (a.scaleA) acc.mul(a.scaleA[srow]);
  if (a.scaleB) acc.mul(a.scaleB[srow]);
  acc.store(a.dst + drow * F + col);
}

// Fused V->E->V for one vertex panel, hyperedge sums staged in LDS.
//  stage : the panel's slot offsets, member entries, (vertex,slot) incidences,
//          row pointers and scales -> LDS (coalesced loads)
//  hop 1 : every LPR-lane group walks a contiguous run of slots as one flat entry
//          stream (U row gathers in flight; an entry with bit 31 set reads the
//          materialised table instead of X), scales the finished sum by
//          degE*W and writes it to its row of the LDS tile
//  hop 2 : every group sums, for each of its vertices, the tile rows of the
//          vertex's hyperedges in H-CSR order, scales by degV and stores Y.
// Same arithmetic and order as the two-phase path (and the CPU reference), but
// the hyperedge feature rows never leave the CU.
// Diagnostic stamps (debug bit 32): lane 0 of every wave adds the cycles since
// the previous stamp to a global counter per phase.  Never on in production.
__device__ unsigned long long hg_stamps[8];
#define HG_STAMP(i)                                                        \
  do {                                                                     \
    if (stamp) {                                                           \
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();          \
      atomicAdd(&hg_stamps[i], t1 - t0);                                   \
      t0 = t1;                                                             \
    }                                                                      \
  } while (0)

template <int LPR, int VEC, int U, int BS>
__global__ __launch_bounds__(BS) void fused_panel_kernel(const FusedArgs a) {
  constexpr int NG = BS / LPR;
  constexpr int TW = LPR * VEC;  // tile row stride in floats
  using V = Vec<VEC>;
  extern __shared__ int32_t smem[];
  const int tid = threadIdx.x;
  const int gl = tid & (LPR - 1);
  const int lcol = gl * VEC;
  const int col = blockIdx.y * TW + lcol;
  const bool col_ok = col < a.F;
  const int64_t F = a.F;
  int b = blockIdx.x;
  if (a.xcd_remap) {
    const int x = b & 7, i = b >> 3;
    const int cpx = a.npanels >> 3, rem = a.npanels & 7;
    b = x * cpx + (x < rem ? x : rem) + i;
  }
  const bool stamp = (a.debug & 32) && (threadIdx.x & 63) == 0;
  unsigned long long t0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
  const FPanel pn = a.panels[b];
  HG_STAMP(0);

  float *tile = reinterpret_cast<float *>(smem);            // [cap * TW]
  int32_t *soff = smem + a.cap * TW;                         // [cap + 1]
  float *sA = reinterpret_cast<float *>(soff + a.cap + 1);   // [cap]
  float *sB = sA + a.cap;                                    // [cap]
  int32_t *spm = reinterpret_cast<int32_t *>(sB + a.cap);    // [mem_cap]
  int32_t *sptr = spm + a.mem_cap;                           // [rows_cap + 1]
  float *sdeg = reinterpret_cast<float *>(sptr + a.rows_cap + 1);  // [rows_cap]
  int32_t *srow = reinterpret_cast<int32_t *>(sdeg + a.rows_cap);  // [rows_cap]
  uint16_t *svs = reinterpret_cast<uint16_t *>(srow + a.rows_cap);  // [vslot_cap]

  for (int i = tid; i <= pn.nslots; i += BS) soff[i] = a.soff[pn.sbase + i];
  for (int i = tid; i < pn.npm; i += BS) spm[i] = a.pmem[pn.pm0 + i];
  if (tid == 0) sptr[0] = 0;
  for (int i = tid; i < pn.nrows; i += BS) {
    sptr[i + 1] = a.pend[pn.r0 + i];
    const int v = a.prow[pn.r0 + i];
    srow[i] = v;
    if (a.degV) sdeg[i] = a.degV[v];
  }
  for (int i = tid; i < pn.nvs; i += BS) svs[i] = a.pvs[pn.v0 + i];
  if (a.degE || a.W) {
    for (int i = tid; i < pn.nslots; i += BS) {
      const int e = a.slot_eid[pn.eid0 + i];  // -1: materialised row, already scaled
      sA[i] = (a.degE && e >= 0) ? a.degE[e] : 1.0f;
      sB[i] = (a.W && e >= 0) ? a.W[e] : 1.0f;
    }
  }
  HG_STAMP(1);
  __syncthreads();
  HG_STAMP(2);
  if (a.debug & 16) return;  // ablation: descriptor + lists only

  const int g = tid / LPR;
  if (!(a.debug & 4)) {  // ---- hop 1: slots -> LDS tile
    const int spg = (pn.nslots + NG - 1) / NG;
    int k = min(g * spg, pn.nslots);
    const int ke = min(k + spg, pn.nslots);
    if (k < ke) {
      auto flush = [&](int slot, V acc) {
        if (a.degE) acc.mul(sA[slot]);
        if (a.W) acc.mul(sB[slot]);
        acc.store(tile + slot * TW + lcol);
      };
      int pos = soff[k];
      const int stop = soff[ke];
      int slot_end = soff[k + 1];
      V acc = V::zero();
      while (pos < stop) {
        const int n = min(U, stop - pos);
        V v[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
          const int ent = spm[pos + min(j, n - 1)];
          const float *base = ent < 0 ? a.Xe_mat : a.X;
          const int64_t idx = ent & 0x7fffffff;
          v[j] = (col_ok && !(a.debug & 1)) ? V::load(base + idx * F + col) : V::zero();
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
          if (j < n) {
            while (slot_end <= pos + j) {
              flush(k, acc);
              acc = V::zero();
              k++;
              slot_end = soff[k + 1];
            }
            acc.add(v[j]);
          }
        }
        pos += n;
      }
      flush(k, acc);  // every slot has at least one entry, so the last one is still open
    }
  }
  HG_STAMP(3);
  __syncthreads();
  HG_STAMP(4);
  if (!(a.debug & 8)) {  // ---- hop 2: vertices <- LDS tile
    const int rpg = (pn.nrows + NG - 1) / NG;
    const int r0 = min(g * rpg, pn.nrows), r1 = min(r0 + rpg, pn.nrows);
    for (int r = r0; r < r1; r++) {
      V acc = V::zero();
      const int pb = sptr[r], pe = sptr[r + 1];
      for (int p = pb; p < pe; p++) acc.add(V::load(tile + (int)svs[p] * TW + lcol));
      if (a.degV && pe > pb) acc.mul(sdeg[r]);
      if (col_ok && !(a.debug & 2)) acc.store(a.Y + (int64_t)srow[r] * F + col);
    }
  }
  HG_STAMP(5);
}

// LDS-DMA form of the fused panel kernel.  Every member row of every slot of the
// panel is gathered by `global_load_lds` (per-lane source address, LDS
// destination = wave-uniform base + lane * 16 B, so one wave instruction lands
// 64/LPR consecutive entries as consecutive rows of the landing zone) with no
// VGPR staging: all of a panel's gathers are in flight at once instead of U per
// lane.  The chain per workgroup is descriptor -> lists -> rows -> store.
// Slot sums are then formed from LDS in entry order and written over the slot's
// first entry row, which doubles as the hyperedge tile for hop 2.
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void fused_dma_kernel(const FusedArgs a) {
  constexpr int BS = 256;
  constexpr int NG = BS / LPR;
  constexpr int TW = LPR * VEC;   // row stride of the landing zone in floats
  constexpr int EPI = 64 / LPR;   // entries per DMA wave instruction
  using V = Vec<VEC>;
  extern __shared__ int32_t smem[];
  const int tid = threadIdx.x;
  const int gl = tid & (LPR - 1);
  const int lcol = gl * VEC;
  const int col = blockIdx.y * TW + lcol;
  const bool col_ok = col < a.F;
  const int64_t F = a.F;
  int b = blockIdx.x;
  if (a.xcd_remap) {
    const int x = b & 7, i = b >> 3;
    const int cpx = a.npanels >> 3, rem = a.npanels & 7;
    b = x * cpx + (x < rem ? x : rem) + i;
  }
  const FPanel pn = a.panels[b];

  float *stage = reinterpret_cast<float *>(smem);             // [mem_cap * TW]
  int32_t *soff = smem + a.mem_cap * TW;                       // [cap + 1]
  float *sA = reinterpret_cast<float *>(soff + a.cap + 1);     // [cap]
  float *sB = sA + a.cap;                                      // [cap]
  int32_t *spm = reinterpret_cast<int32_t *>(sB + a.cap);      // [mem_cap]
  int32_t *sptr = spm + a.mem_cap;                             // [rows_cap + 1]
  float *sdeg = reinterpret_cast<float *>(sptr + a.rows_cap + 1);   // [rows_cap]
  int32_t *srow = reinterpret_cast<int32_t *>(sdeg + a.rows_cap);   // [rows_cap]
  int32_t *seid = srow + a.rows_cap;                           // [cap]
  uint16_t *svs = reinterpret_cast<uint16_t *>(seid + a.cap);  // [vslot_cap]

  for (int i = tid; i <= pn.nslots; i += BS) soff[i] = a.soff[pn.sbase + i];
  for (int i = tid; i < pn.npm; i += BS) spm[i] = a.pmem[pn.pm0 + i];
  if (tid == 0) sptr[0] = 0;
  for (int i = tid; i < pn.nrows; i += BS) {
    sptr[i + 1] = a.pend[pn.r0 + i];
    srow[i] = a.prow[pn.r0 + i];
  }
  for (int i = tid; i < pn.nvs; i += BS) svs[i] = a.pvs[pn.v0 + i];
  if (a.degE || a.W)
    for (int i = tid; i < pn.nslots; i += BS) seid[i] = a.slot_eid[pn.eid0 + i];
  __syncthreads();

  // ---- all row gathers of the panel, straight into LDS
  {
    const int wave = tid >> 6, lane = tid & 63;
    const int nchunks = (pn.npm + EPI - 1) / EPI;
    for (int c = wave; c < nchunks; c += BS / 64) {
      const int e = min(c * EPI + lane / LPR, pn.npm - 1);
      const int ent = spm[e];
      const float *base = ent < 0 ? a.Xe_mat : a.X;
      const float *src = base + (int64_t)(ent & 0x7fffffff) * F + col;
      float *dst = stage + (size_t)c * EPI * TW;  // hardware adds lane * VEC * 4 bytes
      if (col_ok) {
        // the size argument must be a literal (1, 2, 4, 12 or 16)
        if constexpr (VEC == 4)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                           (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                           (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
      }
    }
    // the scale gathers ride in the same round trip
    if (a.degE || a.W)
      for (int i = tid; i < pn.nslots; i += BS) {
        const int e = seid[i];  // -1: materialised row, already scaled
        sA[i] = (a.degE && e >= 0) ? a.degE[e] : 1.f;
        sB[i] = (a.W && e >= 0) ? a.W[e] : 1.f;
      }
    if (a.degV)
      for (int i = tid; i < pn.nrows; i += BS) sdeg[i] = a.degV[srow[i]];
  }
  __syncthreads();  // drains the DMA (vmcnt(0)) and publishes the scales

  const int g = tid / LPR;
  {  // ---- hop 1: slot sums, in entry order, written over the slot's first row
    const int spg = (pn.nslots + NG - 1) / NG;
    const int k0 = min(g * spg, pn.nslots), k1 = min(k0 + spg, pn.nslots);
    for (int k = k0; k < k1; k++) {
      const int pb = soff[k], pe = soff[k + 1];
      V acc = V::zero();
      for (int p = pb; p < pe; p++) acc.add(V::load(stage + p * TW + lcol));
      if (a.degE) acc.mul(sA[k]);
      if (a.W) acc.mul(sB[k]);
      acc.store(stage + pb * TW + lcol);
    }
  }
  __syncthreads();
  {  // ---- hop 2: vertices <- slot rows
    const int rpg = (pn.nrows + NG - 1) / NG;
    const int r0 = min(g * rpg, pn.nrows), r1 = min(r0 + rpg, pn.nrows);
    for (int r = r0; r < r1; r++) {
      V acc = V::zero();
      const int pb = sptr[r], pe = sptr[r + 1];
      for (int p = pb; p < pe; p++) acc.add(V::load(stage + soff[svs[p]] * TW + lcol));
      if (a.degV && pe > pb) acc.mul(sdeg[r]);
      if (col_ok) acc.store(a.Y + (int64_t)srow[r] * F + col);
    }
  }
}

// Packed form of the fused panel kernel.  The plan hands every panel over as ONE
// contiguous int32 record (hg_fused.cpp, pack_records): a single coalesced copy
// stages it, and hop 1 is a wave-uniform loop over a [step][group] entry stream in
// which the panel's hyperedge slots were packed longest-first over the lane
// groups -- no per-group offsets to look up, no divergent trip counts, the same
// number of row gathers for every group.  Entry word: -1 idle, else bits 0..29 =
// row, bit 30 = row of the materialised table, bit 31 = last member of its slot
// (scale, store to the LDS tile, start the next slot).  Members of a slot stay in
// their CSR order, so the arithmetic is still the CPU reference's.
typedef unsigned hg_u4 __attribute__((ext_vector_type(4)));
typedef int hg_i4 __attribute__((ext_vector_type(4)));

// FAST (VEC = 4 only): rows are fetched with buffer_load_dwordx4 through a buffer
// descriptor and a 32-bit byte offset formed by one v_mad_u32_u24 (row * row_bytes +
// column bytes) instead of 64-bit pointer arithmetic; needs N < 2^24, F*4 < 2^24 and
// tables below 2 GiB (the launcher checks; otherwise FAST = false runs the same loop
// on global loads).
template <int LPR, int VEC, int U, bool FAST>
__global__ __launch_bounds__(256) void fused_packed_kernel(const FusedArgs a) {
  constexpr int BS = 256;
  constexpr int NG = BS / LPR;
  constexpr int TW = LPR * VEC;
  using V = Vec<VEC>;
  extern __shared__ int32_t smem[];
  const int tid = threadIdx.x;
  const int gl = tid & (LPR - 1);
  const int lcol = gl * VEC;
  const int col = blockIdx.y * TW + lcol;
  const bool col_ok = col < a.F;
  const int64_t F = a.F;
  int b = blockIdx.x;
  if (a.xcd_remap) {
    const int x = b & 7, i = b >> 3;
    const int cpx = a.npanels >> 3, rem = a.npanels & 7;
    b = x * cpx + (x < rem ? x : rem) + i;
  }
  const bool stamp = (a.debug & 32) && (threadIdx.x & 63) == 0;
  unsigned long long t0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
  const FRec rt = a.rec_tab[b];
  HG_STAMP(0);
  const int32_t *grec = a.rec + rt.off;

  float *tile = reinterpret_cast<float *>(smem);        // [cap * TW]
  int32_t *rec = smem + a.cap * TW;                      // [max_rec_words], 16-byte aligned
  float *sA = reinterpret_cast<float *>(rec + a.max_rec_words);  // [cap]
  float *sB = sA + a.cap;                                // [cap]
  float *sdeg = sB + a.cap;                              // [rows_cap]

  // records are padded to whole 16-byte units: one dwordx4 per lane copies 4 KB per pass
  for (int i = tid; i < (rt.len >> 2); i += BS)
    reinterpret_cast<hg_i4 *>(rec)[i] = reinterpret_cast<const hg_i4 *>(grec)[i];
  // The scale gathers start from the ids in global memory, in the same round trip as the
  // record copy (the descriptor says where they are), not after it.
  if (a.degE || a.W)
    for (int i = tid; i < rt.nslots; i += BS) {
      if (a.bsA) {  // bound: one coalesced read instead of a scattered 4-byte gather per slot
        sA[i] = a.bsA[rt.slot_base + i];
        sB[i] = a.bsB[rt.slot_base + i];
      } else {
        const int e = grec[rt.off_eid + i];  // -1: materialised row, already scaled
        sA[i] = (a.degE && e >= 0) ? a.degE[e] : 1.f;
        sB[i] = (a.W && e >= 0) ? a.W[e] : 1.f;
      }
    }
  if (a.degV)
    for (int i = tid; i < rt.nrows; i += BS)
      sdeg[i] = a.bsD ? a.bsD[rt.row_base + i] : a.degV[grec[rt.off_prow + i]];
  HG_STAMP(1);
  __syncthreads();
  HG_STAMP(2);
  if (a.debug & 16) return;  // ablation (experiments): record copy only
  const int steps = rec[0], nrows = rec[1];
  const int32_t *gbase = rec + rec[4];
  const int32_t *stream = rec + rec[5];
  const int32_t *pend = rec + rec[6];
  const int32_t *prow = rec + rec[7];
  const uint16_t *pvs = reinterpret_cast<const uint16_t *>(rec + rec[9]);

  const int g = tid / LPR;
  if (!(a.debug & 4)) {  // ---- hop 1
    int slot = gbase[g];
    V acc = V::zero();
    [[maybe_unused]] const unsigned row_bytes = (unsigned)a.F * 4u, col_bytes = (unsigned)col * 4u;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rx, rm;
    if constexpr (FAST) {
      rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.X), 0, a.x_bytes, 0x00020000);
      rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.Xe_mat ? a.Xe_mat : a.X), 0,
                                             a.Xe_mat ? a.mat_bytes : 0, 0x00020000);
    }
    for (int s0 = 0; s0 < steps; s0 += U) {
      int ent[U];
#pragma unroll
      for (int j = 0; j < U; j++) ent[j] = (s0 + j < steps) ? stream[(s0 + j) * NG + g] : -1;
      V v[U];
#pragma unroll
      for (int j = 0; j < U; j++) {
        const bool on = col_ok && ent[j] != -1 && !(a.debug & 1);
        if constexpr (FAST) {
          const unsigned off = __umul24((unsigned)ent[j] & 0x3fffffffu, row_bytes) + col_bytes;
          const bool mat = a.Xe_mat && (ent[j] & 0x40000000);
          hg_u4 q = {0u, 0u, 0u, 0u};
          if (on && !mat) q = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
          if (a.Xe_mat) {  // wave-uniform: only graphs with materialised hyperedges pay for it
            if (on && mat) q = __builtin_amdgcn_raw_buffer_load_b128(rm, off, 0, 0);
          }
          v[j].v = __builtin_bit_cast(float4, q);
        } else {
          const int64_t idx = ent[j] & 0x3fffffff;
          const float *base = (a.Xe_mat && (ent[j] & 0x40000000)) ? a.Xe_mat : a.X;
          v[j] = on ? V::load(base + idx * F + col) : V::zero();
        }
      }
#pragma unroll
      for (int j = 0; j < U; j++) {
        acc.add(v[j]);        // an idle step contributed zeros
        if (ent[j] < -1) {    // bit 31 set and not the idle word: last member of this slot
          if (a.degE) acc.mul(sA[slot]);
          if (a.W) acc.mul(sB[slot]);
          acc.store(tile + slot * TW + lcol);
          slot++;
          acc = V::zero();
        }
      }
    }
  }
  HG_STAMP(3);
  __syncthreads();
  HG_STAMP(4);
  if (!(a.debug & 8)) {  // ---- hop 2
    const int rpg = (nrows + NG - 1) / NG;
    const int r0 = min(g * rpg, nrows), r1 = min(r0 + rpg, nrows);
    for (int r = r0; r < r1; r++) {
      V acc = V::zero();
      const int pb = r ? pend[r - 1] : 0, pe = pend[r];
      for (int p = pb; p < pe; p++) acc.add(V::load(tile + (int)pvs[p] * TW + lcol));
      if (a.degV && pe > pb) acc.mul(sdeg[r]);
      if (col_ok && !(a.debug & 2)) {
        if (a.debug & 64) acc.store_nt(a.Y + (int64_t)prow[r] * F + col);
        else acc.store(a.Y + (int64_t)prow[r] * F + col);
      }
    }
  }
  HG_STAMP(5);
}

__device__ __forceinline__ void dma_copy_dwords(const int32_t *src, int32_t *dst, int n, int lane);

// Wave-specialised persistent form of fused_packed_kernel: 4 compute waves + 1
// loader wave, two record buffers in LDS.  The loader wave draws the next panel of
// its XCD class from a per-class counter in global memory (dynamic, so the grid
// may be any size >= what the chip holds), copies that panel's record into the
// idle buffer by LDS-DMA and gathers its scales, all while the compute waves work
// on the current buffer: descriptor, record and scale latencies are off the
// compute waves' critical path, and because a wave's memory counter is its own,
// the loader's loads never queue ahead of the compute waves' row gathers.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the
// vector-memory counter, which on CDNA4 counts stores: in a persistent loop that would
// expose the full latency of the Y stores at every panel boundary.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int LPR, int VEC, int U>
__global__ __launch_bounds__(320) void fused_packed_ws_kernel(const FusedArgs a) {
  constexpr int CT = 256;
  constexpr int NG = CT / LPR;
  constexpr int TW = LPR * VEC;
  using V = Vec<VEC>;
  extern __shared__ int32_t smem[];
  const int tid = threadIdx.x;
  const bool loader = tid >= CT;
  const int64_t F = a.F;

  // this workgroup's class of panels (one contiguous eighth per XCD class)
  int cls, start, end;
  {
    const int x = blockIdx.x & 7;
    const int cpx = a.npanels >> 3, rem = a.npanels & 7;
    cls = x;
    start = x * cpx + (x < rem ? x : rem);
    end = start + cpx + (x < rem ? 1 : 0);
  }

  float *tile = reinterpret_cast<float *>(smem);  // [cap * TW]
  const int bufw = a.max_rec_words + 2 * a.cap + a.rows_cap;  // words per buffer
  int32_t *buf0 = smem + a.cap * TW, *buf1 = buf0 + bufw;
  const bool weighted = a.degE || a.W || a.degV;

  // loader wave: draw a panel, start copying its record into `buf`; returns the record length
  // (0 = class exhausted; rec[0] = -1 then tells the compute waves to stop)
  auto start_load = [&](int32_t *buf, int lane) -> int {
    int idx = 0;
    if (lane == 0) idx = atomicAdd(a.counters + cls * 16, 1);
    idx = __builtin_amdgcn_readfirstlane(idx) + start;
    if (idx >= end) {
      if (lane == 0) buf[0] = -1;
      return 0;
    }
    const FRec rt = a.rec_tab[idx];
    dma_copy_dwords(a.rec + rt.off, buf, rt.len, lane);
    return rt.len;
  };
  auto finish_load = [&](int32_t *buf, int len, int lane) {
    if (len == 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!weighted) return;
    const int nrows = buf[1], nslots = buf[2];
    const int32_t *prow = buf + buf[7];
    const int32_t *eid = buf + buf[8];
    float *sA = reinterpret_cast<float *>(buf + a.max_rec_words), *sB = sA + a.cap, *sdeg = sB + a.cap;
    if (a.degE || a.W)
      for (int i = lane; i < nslots; i += 64) {
        const int e = eid[i];
        sA[i] = (a.degE && e >= 0) ? a.degE[e] : 1.f;
        sB[i] = (a.W && e >= 0) ? a.W[e] : 1.f;
      }
    if (a.degV)
      for (int i = lane; i < nrows; i += 64) sdeg[i] = a.degV[prow[i]];
  };

  if (loader) {
    const int lane = tid - CT;
    const int len = start_load(buf0, lane);
    finish_load(buf0, len, lane);
  }
  __syncthreads();

  for (int it = 0;; it++) {
    int32_t *rec = (it & 1) ? buf1 : buf0;
    int32_t *nxt = (it & 1) ? buf0 : buf1;
    const int steps = rec[0];
    if (steps < 0) break;  // same LDS word for every wave: uniform exit
    const bool stamp = (a.debug & 32) && !loader && (threadIdx.x & 63) == 0;
    unsigned long long t0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    if (loader) {
      const int lane = tid - CT;
      const int len = start_load(nxt, lane);
      lds_barrier();  // (M)
      finish_load(nxt, len, lane);
      lds_barrier();  // (X)
      continue;
    }
    const int gl = tid & (LPR - 1);
    const int lcol = gl * VEC;
    const int col = blockIdx.y * TW + lcol;
    const bool col_ok = col < a.F;
    const int g = tid / LPR;
    const int nrows = rec[1];
    const int32_t *gbase = rec + rec[4];
    const int32_t *stream = rec + rec[5];
    const int32_t *pend = rec + rec[6];
    const int32_t *prow = rec + rec[7];
    const uint16_t *pvs = reinterpret_cast<const uint16_t *>(rec + rec[9]);
    const float *sA = reinterpret_cast<const float *>(rec + a.max_rec_words), *sB = sA + a.cap, *sdeg = sB + a.cap;
    {  // ---- hop 1
      int slot = gbase[g];
      V acc = V::zero();
      for (int s0 = 0; s0 < steps; s0 += U) {
        int ent[U];
#pragma unroll
        for (int j = 0; j < U; j++) ent[j] = (s0 + j < steps) ? stream[(s0 + j) * NG + g] : -1;
        V v[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
          const bool on = col_ok && ent[j] != -1 && !(a.debug & 1);
          const int64_t idx = ent[j] & 0x3fffffff;
          const float *base = (a.Xe_mat && (ent[j] & 0x40000000)) ? a.Xe_mat : a.X;
          v[j] = on ? V::load(base + idx * F + col) : V::zero();
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
          if (ent[j] != -1) {
            acc.add(v[j]);
            if (ent[j] < 0) {
              if (a.degE) acc.mul(sA[slot]);
              if (a.W) acc.mul(sB[slot]);
              acc.store(tile + slot * TW + lcol);
              slot++;
              acc = V::zero();
            }
          }
        }
      }
    }
    HG_STAMP(1);
    lds_barrier();  // (M)
    HG_STAMP(2);
    {  // ---- hop 2
      const int rpg = (nrows + NG - 1) / NG;
      const int r0 = min(g * rpg, nrows), r1 = min(r0 + rpg, nrows);
      for (int r = r0; r < r1; r++) {
        V acc = V::zero();
        const int pb = r ? pend[r - 1] : 0, pe = pend[r];
        for (int p = pb; p < pe; p++) acc.add(V::load(tile + (int)pvs[p] * TW + lcol));
        if (a.degV && pe > pb) acc.mul(sdeg[r]);
        if (col_ok && !(a.debug & 2)) acc.store(a.Y + (int64_t)prow[r] * F + col);
      }
    }
    HG_STAMP(3);
    lds_barrier();  // (X): the Y stores stay in flight across it
    HG_STAMP(4);
  }
}

// Persistent, software-pipelined form of fused_panel_kernel.  A workgroup walks a
// strided sequence of its XCD's panels; while it gathers and sums panel i, the
// lists of panel i+1 are already in flight (into registers, written to LDS when
// panel i is done) and the descriptor of panel i+2 is being fetched, so the
// descriptor -> lists -> rows dependency chain is paid once per workgroup, not
// once per panel.  Arithmetic identical to fused_panel_kernel.
template <int LPR, int VEC, int U>
__global__ __launch_bounds__(256) void fused_persist_kernel(const FusedArgs a) {
  constexpr int BS = 256;
  constexpr int NG = BS / LPR;
  constexpr int TW = LPR * VEC;
  // register prefetch capacity per thread (cap <= 256, mem_cap <= 1024, vslot_cap <= 512)
  constexpr int NK_SOFF = 2, NK_PM = 4, NK_ROW = 1, NK_VS = 2;
  using V = Vec<VEC>;
  extern __shared__ int32_t smem[];
  const int tid = threadIdx.x;
  const int gl = tid & (LPR - 1);
  const int lcol = gl * VEC;
  const int col = blockIdx.y * TW + lcol;
  const bool col_ok = col < a.F;
  const int64_t F = a.F;

  // this workgroup's panel sequence: first, first + step, ... < last
  int first, step, last;
  {
    const int w = blockIdx.x, G = gridDim.x;
    if (a.xcd_remap && G >= 8) {
      const int x = w & 7, j = w >> 3;
      const int J = (G - x + 7) >> 3;  // workgroups of this XCD class
      const int cpx = a.npanels >> 3, rem = a.npanels & 7;
      const int start = x * cpx + (x < rem ? x : rem);
      first = start + j;
      step = J;
      last = start + cpx + (x < rem ? 1 : 0);
    } else {
      first = w;
      step = G;
      last = a.npanels;
    }
  }
  if (first >= last) return;

  float *tile = reinterpret_cast<float *>(smem);            // [cap * TW]
  int32_t *soff = smem + a.cap * TW;                         // [cap + 1]
  float *sA = reinterpret_cast<float *>(soff + a.cap + 1);   // [cap]
  float *sB = sA + a.cap;                                    // [cap]
  int32_t *spm = reinterpret_cast<int32_t *>(sB + a.cap);    // [mem_cap]
  int32_t *sptr = spm + a.mem_cap;                           // [rows_cap + 1]
  float *sdeg = reinterpret_cast<float *>(sptr + a.rows_cap + 1);  // [rows_cap]
  int32_t *srow = reinterpret_cast<int32_t *>(sdeg + a.rows_cap);  // [rows_cap]
  uint16_t *svs = reinterpret_cast<uint16_t *>(srow + a.rows_cap);  // [vslot_cap]

  const bool weighted = a.degE || a.W || a.degV;

  struct Lists {  // one panel's lists, strided over the workgroup's threads
    int32_t off[NK_SOFF], pm[NK_PM], pend[NK_ROW], row[NK_ROW], eid[NK_SOFF];
    uint16_t vs[NK_VS];
  };
  auto fetch = [&](const FPanel &pn, Lists &L) {
#pragma unroll
    for (int k = 0; k < NK_SOFF; k++) {
      const int i = tid + k * BS;
      L.off[k] = i <= pn.nslots ? a.soff[pn.sbase + i] : 0;
      L.eid[k] = (weighted && i < pn.nslots) ? a.slot_eid[pn.eid0 + i] : -1;
    }
#pragma unroll
    for (int k = 0; k < NK_PM; k++) {
      const int i = tid + k * BS;
      L.pm[k] = i < pn.npm ? a.pmem[pn.pm0 + i] : 0;
    }
#pragma unroll
    for (int k = 0; k < NK_ROW; k++) {
      const int i = tid + k * BS;
      L.pend[k] = i < pn.nrows ? a.pend[pn.r0 + i] : 0;
      L.row[k] = i < pn.nrows ? a.prow[pn.r0 + i] : 0;
    }
#pragma unroll
    for (int k = 0; k < NK_VS; k++) {
      const int i = tid + k * BS;
      L.vs[k] = i < pn.nvs ? a.pvs[pn.v0 + i] : (uint16_t)0;
    }
  };
  // second-level gathers (scales), issued as soon as the ids are in registers
  struct Scales {
    float sa[NK_SOFF], sb[NK_SOFF], sd[NK_ROW];
  };
  auto fetch_scales = [&](const FPanel &pn, const Lists &L, Scales &S) {
#pragma unroll
    for (int k = 0; k < NK_SOFF; k++) {
      const int i = tid + k * BS;
      const int e = L.eid[k];
      S.sa[k] = (a.degE && i < pn.nslots && e >= 0) ? a.degE[e] : 1.f;
      S.sb[k] = (a.W && i < pn.nslots && e >= 0) ? a.W[e] : 1.f;
    }
#pragma unroll
    for (int k = 0; k < NK_ROW; k++) {
      const int i = tid + k * BS;
      S.sd[k] = (a.degV && i < pn.nrows) ? a.degV[L.row[k]] : 1.f;
    }
  };
  auto commit = [&](const FPanel &pn, const Lists &L, const Scales &S) {
#pragma unroll
    for (int k = 0; k < NK_SOFF; k++) {
      const int i = tid + k * BS;
      if (i <= pn.nslots) soff[i] = L.off[k];
      if (weighted && i < pn.nslots) {
        sA[i] = S.sa[k];
        sB[i] = S.sb[k];
      }
    }
#pragma unroll
    for (int k = 0; k < NK_PM; k++) {
      const int i = tid + k * BS;
      if (i < pn.npm) spm[i] = L.pm[k];
    }
    if (tid == 0) sptr[0] = 0;
#pragma unroll
    for (int k = 0; k < NK_ROW; k++) {
      const int i = tid + k * BS;
      if (i < pn.nrows) {
        sptr[i + 1] = L.pend[k];
        srow[i] = L.row[k];
        if (weighted) sdeg[i] = S.sd[k];
      }
    }
#pragma unroll
    for (int k = 0; k < NK_VS; k++) {
      const int i = tid + k * BS;
      if (i < pn.nvs) svs[i] = L.vs[k];
    }
  };

  const int g = tid / LPR;
  FPanel pn = a.panels[first];
  Lists L;
  Scales S;
  fetch(pn, L);
  if (weighted) fetch_scales(pn, L, S);
  commit(pn, L, S);
  int nxt = first + step;
  FPanel pn1 = a.panels[min(nxt, a.npanels - 1)];
  __syncthreads();

  for (;; nxt += step) {
    const bool has_next = nxt < last;
    // descriptor two panels ahead, lists one panel ahead
    const FPanel pn2 = a.panels[min(nxt + step, a.npanels - 1)];
    if (has_next) fetch(pn1, L);

    {  // ---- hop 1: slots -> LDS tile
      const int spg = (pn.nslots + NG - 1) / NG;
      int k = min(g * spg, pn.nslots);
      const int ke = min(k + spg, pn.nslots);
      if (k < ke) {
        auto flush = [&](int slot, V acc) {
          if (a.degE) acc.mul(sA[slot]);
          if (a.W) acc.mul(sB[slot]);
          acc.store(tile + slot * TW + lcol);
        };
        int pos = soff[k];
        const int stop = soff[ke];
        int slot_end = soff[k + 1];
        V acc = V::zero();
        while (pos < stop) {
          const int n = min(U, stop - pos);
          V v[U];
#pragma unroll
          for (int j = 0; j < U; j++) {
            const int ent = spm[pos + min(j, n - 1)];
            const float *base = ent < 0 ? a.Xe_mat : a.X;
            const int64_t idx = ent & 0x7fffffff;
            v[j] = col_ok ? V::load(base + idx * F + col) : V::zero();
          }
#pragma unroll
          for (int j = 0; j < U; j++) {
            if (j < n) {
              while (slot_end <= pos + j) {
                flush(k, acc);
                acc = V::zero();
                k++;
                slot_end = soff[k + 1];
              }
              acc.add(v[j]);
            }
          }
          pos += n;
        }
        flush(k, acc);
      }
    }
    if (has_next && weighted) fetch_scales(pn1, L, S);  // ids have landed by now
    __syncthreads();
    {  // ---- hop 2: vertices <- LDS tile
      const int rpg = (pn.nrows + NG - 1) / NG;
      const int r0 = min(g * rpg, pn.nrows), r1 = min(r0 + rpg, pn.nrows);
      for (int r = r0; r < r1; r++) {
        V acc = V::zero();
        const int pb = sptr[r], pe = sptr[r + 1];
        for (int p = pb; p < pe; p++) acc.add(V::load(tile + (int)svs[p] * TW + lcol));
        if (a.degV && pe > pb) acc.mul(sdeg[r]);
        if (col_ok) acc.store(a.Y + (int64_t)srow[r] * F + col);
      }
    }
    if (!has_next) break;
    __syncthreads();  // everyone is done with this panel's lists and tile
    commit(pn1, L, S);
    pn = pn1;
    pn1 = pn2;
    __syncthreads();
  }
}

// Wave-specialised persistent form: 4 compute waves + 1 loader wave per
// workgroup, two list buffers in LDS.  While the compute waves gather and sum
// panel i, the loader wave copies the lists of panel i+1 into the other buffer by
// LDS-DMA (global_load_lds, no registers) and gathers its scales.  The loader's
// memory counter is its own, so -- unlike a register prefetch issued by the
// compute waves, whose row gathers would queue behind it -- the descriptor ->
// lists latency is entirely off the compute waves' critical path.
struct WsLists {
  int32_t *soff;   // [cap + 1]
  float *sA, *sB;  // [cap]
  int32_t *seid;   // [cap]
  int32_t *spm;    // [mem_cap]
  int32_t *sptr;   // [rows_cap + 1]
  float *sdeg;     // [rows_cap]
  int32_t *srow;   // [rows_cap]
  uint16_t *svs;   // [vslot_cap]
  int32_t *hdr;    // [4]: nslots, nrows
};

__device__ __forceinline__ size_t ws_lists_dwords(const FusedArgs &a) {
  return (size_t)(a.cap + 1) + 3 * (size_t)a.cap + a.mem_cap + (a.rows_cap + 1) + 2 * (size_t)a.rows_cap +
         (a.vslot_cap + 1) / 2 + 4;
}

__device__ __forceinline__ WsLists ws_carve(int32_t *p, const FusedArgs &a) {
  WsLists L;
  L.soff = p;
  L.sA = reinterpret_cast<float *>(L.soff + a.cap + 1);
  L.sB = L.sA + a.cap;
  L.seid = reinterpret_cast<int32_t *>(L.sB + a.cap);
  L.spm = L.seid + a.cap;
  L.sptr = L.spm + a.mem_cap;
  L.sdeg = reinterpret_cast<float *>(L.sptr + a.rows_cap + 1);
  L.srow = reinterpret_cast<int32_t *>(L.sdeg + a.rows_cap);
  L.hdr = L.srow + a.rows_cap;
  L.svs = reinterpret_cast<uint16_t *>(L.hdr + 4);
  return L;
}

// one wave copies n dwords global -> LDS with global_load_lds (64 dwords per instruction)
__device__ __forceinline__ void dma_copy_dwords(const int32_t *src, int32_t *dst, int n, int lane) {
  for (int i0 = 0; i0 < n; i0 += 64) {
    if (i0 + lane < n)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i0 + lane),
                                       (__attribute__((address_space(3))) void *)(dst + i0), 4, 0, 0);
  }
}
__device__ __forceinline__ void dma_copy_u16(const uint16_t *src, uint16_t *dst, int n, int lane) {
  for (int i0 = 0; i0 < n; i0 += 64) {
    if (i0 + lane < n)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i0 + lane),
                                       (__attribute__((address_space(3))) void *)(dst + i0), 2, 0, 0);
  }
}

template <int LPR, int VEC, int U>
__global__ __launch_bounds__(320) void fused_ws_kernel(const FusedArgs a) {
  constexpr int CT = 256;  // compute threads
  constexpr int NG = CT / LPR;
  constexpr int TW = LPR * VEC;
  using V = Vec<VEC>;
  extern __shared__ int32_t smem[];
  const int tid = threadIdx.x;
  const bool loader = tid >= CT;
  const int64_t F = a.F;

  int first, step, last;
  {
    const int w = blockIdx.x, G = gridDim.x;
    if (a.xcd_remap && G >= 8) {
      const int x = w & 7, j = w >> 3;
      const int J = (G - x + 7) >> 3;
      const int cpx = a.npanels >> 3, rem = a.npanels & 7;
      const int start = x * cpx + (x < rem ? x : rem);
      first = start + j;
      step = J;
      last = start + cpx + (x < rem ? 1 : 0);
    } else {
      first = w;
      step = G;
      last = a.npanels;
    }
  }
  if (first >= last) return;

  float *tile = reinterpret_cast<float *>(smem);  // [cap * TW]
  const size_t ldw = ws_lists_dwords(a);
  const WsLists B0 = ws_carve(smem + a.cap * TW, a);
  const WsLists B1 = ws_carve(smem + a.cap * TW + ldw, a);
  const bool weighted = a.degE || a.W || a.degV;

  // loader wave: bring panel `idx` into buffer L
  auto load_panel = [&](int idx, const WsLists &L, int lane) {
    const FPanel pn = a.panels[idx];
    dma_copy_dwords(a.soff + pn.sbase, L.soff, pn.nslots + 1, lane);
    dma_copy_dwords(a.pmem + pn.pm0, L.spm, pn.npm, lane);
    dma_copy_dwords(a.pend + pn.r0, L.sptr + 1, pn.nrows, lane);
    dma_copy_dwords(a.prow + pn.r0, L.srow, pn.nrows, lane);
    if (a.degE || a.W) dma_copy_dwords(a.slot_eid + pn.eid0, L.seid, pn.nslots, lane);
    // 16-bit slot ids go through registers (sub-dword LDS-DMA is not relied upon)
    for (int i = lane; i < pn.nvs; i += 64) L.svs[i] = a.pvs[pn.v0 + i];
    if (lane == 0) {
      L.sptr[0] = 0;
      L.hdr[0] = pn.nslots;
      L.hdr[1] = pn.nrows;
    }
    return pn;
  };
  // second level (needs the ids that just landed)
  auto load_scales = [&](const FPanel &pn, const WsLists &L, int lane) {
    if (a.degE || a.W)
      for (int i = lane; i < pn.nslots; i += 64) {
        const int e = L.seid[i];  // -1: materialised row, already scaled
        L.sA[i] = (a.degE && e >= 0) ? a.degE[e] : 1.f;
        L.sB[i] = (a.W && e >= 0) ? a.W[e] : 1.f;
      }
    if (a.degV)
      for (int i = lane; i < pn.nrows; i += 64) L.sdeg[i] = a.degV[L.srow[i]];
  };

  if (loader) {
    const int lane = tid - CT;
    const FPanel pn = load_panel(first, B0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (weighted) load_scales(pn, B0, lane);
  }
  __syncthreads();

  int it = 0;
  for (int cur = first; cur < last; cur += step, it++) {
    const WsLists &L = (it & 1) ? B1 : B0;
    const WsLists &Ln = (it & 1) ? B0 : B1;
    const bool has_next = cur + step < last;
    if (loader) {
      const int lane = tid - CT;
      FPanel pn{};
      if (has_next) pn = load_panel(cur + step, Ln, lane);
      __syncthreads();  // (M) do not hold the compute waves back
      if (has_next) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (weighted) load_scales(pn, Ln, lane);
      }
      __syncthreads();  // (X) next buffer complete, this panel consumed
      continue;
    }
    const int gl = tid & (LPR - 1);
    const int lcol = gl * VEC;
    const int col = blockIdx.y * TW + lcol;
    const bool col_ok = col < a.F;
    const int g = tid / LPR;
    const int nslots = L.hdr[0], nrows = L.hdr[1];
    {  // ---- hop 1: slots -> LDS tile
      const int spg = (nslots + NG - 1) / NG;
      int k = min(g * spg, nslots);
      const int ke = min(k + spg, nslots);
      if (k < ke) {
        auto flush = [&](int slot, V acc) {
          if (a.degE) acc.mul(L.sA[slot]);
          if (a.W) acc.mul(L.sB[slot]);
          acc.store(tile + slot * TW + lcol);
        };
        int pos = L.soff[k];
        const int stop = L.soff[ke];
        int slot_end = L.soff[k + 1];
        V acc = V::zero();
        while (pos < stop) {
          const int n = min(U, stop - pos);
          V v[U];
#pragma unroll
          for (int j = 0; j < U; j++) {
            const bool on = col_ok && j < n;  // predicated off: no duplicate traffic
            const int ent = on ? L.spm[pos + j] : 0;
            const float *base = ent < 0 ? a.Xe_mat : a.X;
            const int64_t idx = ent & 0x7fffffff;
            v[j] = on ? V::load(base + idx * F + col) : V::zero();
          }
#pragma unroll
          for (int j = 0; j < U; j++) {
            if (j < n) {
              while (slot_end <= pos + j) {
                flush(k, acc);
                acc = V::zero();
                k++;
                slot_end = L.soff[k + 1];
              }
              acc.add(v[j]);
            }
          }
          pos += n;
        }
        flush(k, acc);
      }
    }
    __syncthreads();  // (M)
    {  // ---- hop 2: vertices <- LDS tile
      const int rpg = (nrows + NG - 1) / NG;
      const int r0 = min(g * rpg, nrows), r1 = min(r0 + rpg, nrows);
      for (int r = r0; r < r1; r++) {
        V acc = V::zero();
        const int pb = L.sptr[r], pe = L.sptr[r + 1];
        for (int p = pb; p < pe; p++) acc.add(V::load(tile + (int)L.svs[p] * TW + lcol));
        if (a.degV && pe > pb) acc.mul(L.sdeg[r]);
        if (col_ok) acc.store(a.Y + (int64_t)L.srow[r] * F + col);
      }
    }
    __syncthreads();  // (X)
  }
}

// The reference's register-fused scheme on wave64: LPR lanes = LPR feature
// columns of one task, 64/LPR tasks per wave.  Gather-sum the read partition,
// scale by degE*W, scatter acc*degV[v] to the write partition with hardware
// fp32 atomics (global_atomic_add_f32, one contiguous LPR*4-byte segment per
// destination row).
template <int LPR>
__global__ __launch_bounds__(256) void push_groups_kernel(const PushArgs a) {
  const int tid = threadIdx.x;
  const int64_t gid = (int64_t)blockIdx.x * (256 / LPR) + tid / LPR;
  const int k = blockIdx.y * LPR + (tid & (LPR - 1));
  if (gid >= a.n_group || k >= a.F) return;
  const int64_t F = a.F;
  int eid, rd_start, rd_end, wr_start, wr_end;
  if (a.group_key) {
    eid = a.group_row[gid];
    const int rid = a.group_st[gid], wid = a.group_ed[gid];
    rd_start = a.group_key[rid];
    rd_end = a.group_key[rid + 1];
    wr_start = a.group_key[wid];
    wr_end = a.group_key[wid + 1];
  } else {
    eid = (int)gid;
    rd_start = wr_start = a.csrptr_t[eid];
    rd_end = wr_end = a.csrptr_t[eid + 1];
  }
  float acc = 0.f;
  for (int p = rd_start; p < rd_end; p++) acc += a.X[(int64_t)a.colind_t[p] * F + k];
  const float degE_val = a.degE ? a.degE[eid] : 1.f;
  const float W_val = a.W ? a.W[eid] : 1.f;
  acc *= degE_val * W_val;
  for (int p = wr_start; p < wr_end; p++) {
    const int64_t v = a.colind_t[p];
    const float degV_val = a.degV ? a.degV[v] : 1.f;
    atomicAdd(a.Y + v * F + k, acc * degV_val);
  }
}

static inline int next_pow2(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

struct Tuning {
  int unroll = 4;
  int pipe = 0;
  int fused_bs = 256;
  int fused_u = 8;
  int fused_dma = 0;
  int fused_persist = 0;
  int fused_ws = 0;
  int fused_packed = 1;
  int fused_fast = 1;
  int fused_coltile = 0;
  int fused_grid = 0;
  int fused_debug = 0;  // ablation bits for fused_panel_kernel (timing experiments only)
};
// Experiment knobs (HG_UNROLL = 4|8, HG_PIPE = 0|1), read once.
static const Tuning &tuning() {
  static const Tuning t = [] {
    Tuning x;
    if (const char *e = getenv("HG_UNROLL")) x.unroll = atoi(e) == 8 ? 8 : 4;
    if (const char *e = getenv("HG_PIPE")) x.pipe = atoi(e) != 0;
    if (const char *e = getenv("HG_FUSED_BS")) x.fused_bs = atoi(e);
    if (const char *e = getenv("HG_FUSED_U")) x.fused_u = atoi(e);
    if (const char *e = getenv("HG_FUSED_DMA")) x.fused_dma = atoi(e) != 0;
    if (const char *e = getenv("HG_FUSED_PERSIST")) x.fused_persist = atoi(e) != 0;
    if (const char *e = getenv("HG_FUSED_WS")) x.fused_ws = atoi(e) != 0;
    if (const char *e = getenv("HG_FUSED_PACKED")) x.fused_packed = atoi(e);
    if (const char *e = getenv("HG_FUSED_FAST")) x.fused_fast = atoi(e) != 0;
    if (const char *e = getenv("HG_FUSED_COLTILE")) x.fused_coltile = atoi(e) != 0;
    if (const char *e = getenv("HG_FUSED_GRID")) x.fused_grid = atoi(e);
    if (const char *e = getenv("HG_FUSED_DEBUG")) x.fused_debug = atoi(e);
    return x;
  }();
  return t;
}

template <int LPR, int VEC>
static hipError_t launch_gather_t(const GatherArgs &a, int nfix, const Fixup *fixups,
                                  hipStream_t stream) {
  const int col_tiles = (a.F + LPR * VEC - 1) / (LPR * VEC);
  const int nblocks = a.n_task_blocks + a.npanels;
  if (nblocks > 0) {
    const size_t lds = (size_t)(4 * a.panel_rows + 1 + a.panel_nnz) * sizeof(int32_t);
    const dim3 grid(nblocks, col_tiles), block(256);
    const Tuning &t = tuning();
    if (t.unroll == 8) {
      if (t.pipe) hipLaunchKernelGGL((gather_rows_kernel<LPR, VEC, 8, true>), grid, block, lds, stream, a);
      else hipLaunchKernelGGL((gather_rows_kernel<LPR, VEC, 8, false>), grid, block, lds, stream, a);
    } else {
      if (t.pipe) hipLaunchKernelGGL((gather_rows_kernel<LPR, VEC, 4, true>), grid, block, lds, stream, a);
      else hipLaunchKernelGGL((gather_rows_kernel<LPR, VEC, 4, false>), grid, block, lds, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (nfix > 0) {
    const int per_block = 256 / LPR;
    hipLaunchKernelGGL((fixup_rows_kernel<LPR, VEC>), dim3((nfix + per_block - 1) / per_block, col_tiles),
                       dim3(256), 0, stream, a, fixups, nfix);
    return hipGetLastError();
  }
  return hipSuccess;
}

hipError_t launch_gather(const GatherArgs &a, int nfix, const Fixup *fixups, bool vec4,
                         hipStream_t stream) {
  const int lanes = vec4 ? a.F / 4 : a.F;
  const int lpr = std::min(64, next_pow2(std::max(lanes, 1)));
#define HG_CASE(L)                                                        \
  case L:                                                                 \
    return vec4 ? launch_gather_t<L, 4>(a, nfix, fixups, stream)          \
                : launch_gather_t<L, 1>(a, nfix, fixups, stream);
  switch (lpr) {
    HG_CASE(1)
    HG_CASE(2)
    HG_CASE(4)
    HG_CASE(8)
    HG_CASE(16)
    HG_CASE(32)
    HG_CASE(64)
  }
#undef HG_CASE
  return hipErrorInvalidValue;
}

template <int LPR, int VEC>
static hipError_t launch_fused_t(const FusedArgs &a, hipStream_t stream) {
  if (a.npanels == 0) return hipSuccess;
  constexpr int TW = LPR * VEC;
  const int col_tiles = (a.F + TW - 1) / TW;
  const size_t lds = (size_t)a.cap * TW * 4 + (size_t)(a.cap + 1 + 2 * a.cap + a.mem_cap + a.rows_cap + 1 + 2 * a.rows_cap) * 4 +
                     (size_t)a.vslot_cap * 2 + 16;
  const dim3 grid(a.npanels, col_tiles);
  const Tuning &t = tuning();
  if (a.dma) {
    const size_t lds_dma = (size_t)a.mem_cap * TW * 4 +
                           (size_t)(a.cap + 1 + 2 * a.cap + a.mem_cap + a.rows_cap + 1 + 2 * a.rows_cap + a.cap) * 4 +
                           (size_t)a.vslot_cap * 2 + 16;
    hipLaunchKernelGGL((fused_dma_kernel<LPR, VEC>), grid, dim3(256), lds_dma, stream, a);
    return hipGetLastError();
  }
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu <= 0)
      num_cu = 256;
  }
  if (t.fused_packed == 2 && a.ng == 256 / LPR) {
    // persistent + loader wave; work is drawn from per-class counters, so any grid that
    // covers the chip works (surplus workgroups find their class empty and leave)
    const size_t bufw = (size_t)a.max_rec_words + 2 * a.cap + a.rows_cap;
    const size_t lds_w = (size_t)a.cap * TW * 4 + 2 * bufw * 4 + 16;
    hipError_t e = hipMemsetAsync(a.counters, 0, 512, stream);
    if (e != hipSuccess) return e;
    int per_cu = (int)std::min<size_t>(6, (160 * 1024) / (lds_w + 256));
    if (per_cu < 1) per_cu = 1;
    const int want = t.fused_grid > 0 ? t.fused_grid : num_cu * per_cu;
    const int nwg = std::max(8, std::min((a.npanels + 7) / 8 * 8, want));
    FusedArgs aw = a;
    aw.debug = t.fused_debug;
    if (t.fused_u == 8)
      hipLaunchKernelGGL((fused_packed_ws_kernel<LPR, VEC, 8>), dim3(nwg, col_tiles), dim3(320), lds_w, stream, aw);
    else
      hipLaunchKernelGGL((fused_packed_ws_kernel<LPR, VEC, 4>), dim3(nwg, col_tiles), dim3(320), lds_w, stream, aw);
    return hipGetLastError();
  }
  if (t.fused_packed && a.ng == 256 / LPR) {
    const size_t lds_p = (size_t)a.cap * TW * 4 + (size_t)a.max_rec_words * 4 +
                         (size_t)(2 * a.cap + a.rows_cap) * 4 + 16;
    FusedArgs ap = a;
    ap.debug = t.fused_debug;
    if constexpr (VEC == 4) {
      const bool fast = t.fused_fast && a.x_bytes > 0 && a.nrows_x < (1 << 24) && a.F < (1 << 22) &&
                        (!a.Xe_mat || a.mat_bytes > 0);
      if (fast) {
        if (t.fused_u == 4)
          hipLaunchKernelGGL((fused_packed_kernel<LPR, VEC, 4, true>), grid, dim3(256), lds_p, stream, ap);
        else
          hipLaunchKernelGGL((fused_packed_kernel<LPR, VEC, 8, true>), grid, dim3(256), lds_p, stream, ap);
        return hipGetLastError();
      }
    }
    if (t.fused_u == 4)
      hipLaunchKernelGGL((fused_packed_kernel<LPR, VEC, 4, false>), grid, dim3(256), lds_p, stream, ap);
    else
      hipLaunchKernelGGL((fused_packed_kernel<LPR, VEC, 8, false>), grid, dim3(256), lds_p, stream, ap);
    return hipGetLastError();
  }
  if (t.fused_ws) {
    // wave-specialised persistent kernel: 4 compute waves + 1 loader wave, two list buffers.
    // Grid = workgroups resident at once, from the LDS footprint (the binding resource here);
    // a larger grid would run a second, mostly idle round.
    const size_t ldw = (size_t)(a.cap + 1) + 3 * (size_t)a.cap + a.mem_cap + (a.rows_cap + 1) +
                       2 * (size_t)a.rows_cap + (a.vslot_cap + 1) / 2 + 4;
    const size_t lds_ws = (size_t)a.cap * TW * 4 + 2 * ldw * 4 + 16;
    int per_cu = (int)std::min<size_t>(6, (160 * 1024) / (lds_ws + 256));  // 6 x 5 waves <= 32 waves/CU
    if (per_cu < 1) per_cu = 1;
    const int want = t.fused_grid > 0 ? t.fused_grid : num_cu * per_cu;
    const int nwg = std::min(a.npanels, want);
    if (t.fused_u == 16)
      hipLaunchKernelGGL((fused_ws_kernel<LPR, VEC, 16>), dim3(nwg, col_tiles), dim3(320), lds_ws, stream, a);
    else if (t.fused_u == 8)
      hipLaunchKernelGGL((fused_ws_kernel<LPR, VEC, 8>), dim3(nwg, col_tiles), dim3(320), lds_ws, stream, a);
    else
      hipLaunchKernelGGL((fused_ws_kernel<LPR, VEC, 4>), dim3(nwg, col_tiles), dim3(320), lds_ws, stream, a);
    return hipGetLastError();
  }
  if (t.fused_persist && a.cap <= 256 && a.mem_cap <= 1024 && a.rows_cap <= 256 && a.vslot_cap <= 512) {
    // persistent grid: as many workgroups as the chip holds at once (an oversized grid
    // would only queue: workgroups never wait for each other)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_persist_kernel<LPR, VEC, 4>, 256, lds) !=
            hipSuccess || per_cu <= 0)
      per_cu = 4;
    const int want = t.fused_grid > 0 ? t.fused_grid : num_cu * per_cu;
    const int nwg = std::min(a.npanels, want);
    hipLaunchKernelGGL((fused_persist_kernel<LPR, VEC, 4>), dim3(nwg, col_tiles), dim3(256), lds, stream, a);
    return hipGetLastError();
  }
  FusedArgs ad = a;
  ad.debug = t.fused_debug;
  if (t.fused_u == 8)
    hipLaunchKernelGGL((fused_panel_kernel<LPR, VEC, 8, 256>), grid, dim3(256), lds, stream, ad);
  else
    hipLaunchKernelGGL((fused_panel_kernel<LPR, VEC, 4, 256>), grid, dim3(256), lds, stream, ad);
  return hipGetLastError();
}

int fused_tile_row_floats(int F, bool vec4);

hipError_t launch_fused(const FusedArgs &a, bool vec4, hipStream_t stream) {
  const int lpr = fused_tile_row_floats(a.F, vec4) / (vec4 ? 4 : 1);
#define HG_CASE(L) \
  case L:          \
    return vec4 ? launch_fused_t<L, 4>(a, stream) : launch_fused_t<L, 1>(a, stream);
  switch (lpr) {
    HG_CASE(1)
    HG_CASE(2)
    HG_CASE(4)
    HG_CASE(8)
    HG_CASE(16)
    HG_CASE(32)
    HG_CASE(64)
  }
#undef HG_CASE
  return hipErrorInvalidValue;
}

bool fused_use_dma() { return tuning().fused_dma != 0; }

hipError_t read_stamps(unsigned long long *out, bool reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(hg_stamps), sizeof(hg_stamps));
  if (e == hipSuccess && reset) {
    unsigned long long z[8] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(hg_stamps), z, sizeof(z));
  }
  return e;
}

// floats per LDS tile row for feature width F (what launch_fused will use)
// Wide rows (F a multiple of 32 floats, above 32) are cut into 128-byte column tiles, one
// workgroup per (panel, tile): every workgroup then has the F = 32 shape -- 8 lanes per row,
// 32 row groups, 128 slots in a 16 KB tile -- instead of a few fat row groups and tiny panels.
int fused_tile_row_floats(int F, bool vec4) {
  if (vec4 && tuning().fused_coltile && F > 32 && F % 32 == 0) return 32;
  const int lanes = vec4 ? F / 4 : F;
  return std::min(64, next_pow2(std::max(lanes, 1))) * (vec4 ? 4 : 1);
}

hipError_t launch_push(const PushArgs &a, hipStream_t stream) {
  const int lpr = std::min(64, next_pow2(std::max(a.F, 1)));
  const int per_block = 256 / lpr;
  const int col_tiles = (a.F + lpr - 1) / lpr;
  const int64_t nblocks = (a.n_group + per_block - 1) / per_block;
  if (nblocks == 0) return hipSuccess;
  if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
#define HG_CASE(L)                                                                               \
  case L:                                                                                        \
    hipLaunchKernelGGL((push_groups_kernel<L>), dim3((unsigned)nblocks, col_tiles), dim3(256), 0, \
                       stream, a);                                                               \
    break;
  switch (lpr) {
    HG_CASE(1)
    HG_CASE(2)
    HG_CASE(4)
    HG_CASE(8)
    HG_CASE(16)
    HG_CASE(32)
    HG_CASE(64)
  }
#undef HG_CASE
  return hipGetLastError();
}

// first_aggr = max (HGNNAggr_f1max_forward_kernel, hgnnaggr_cuda.cu:144-177): per
// hyperedge and feature column the running maximum starts at -1e5, a member
// replaces it on strict >, the winning vertex id goes to record[e, k] (0 when no
// member beats -1e5), then Xe = max * (degE * W).  One lane per (hyperedge,
// column); the second hop is the ordinary row gather over H.
__global__ __launch_bounds__(256) void gather_max_kernel(int32_t M, int32_t F, const int32_t *ptr,
                                                         const int32_t *ind, const float *X,
                                                         const float *degE, const float *W, float *Xe,
                                                         int32_t *record) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)M * F) return;
  const int32_t e = (int32_t)(t / F), k = (int32_t)(t % F);
  float best = -1e5f;
  int32_t who = 0;
  for (int32_t p = ptr[e]; p < ptr[e + 1]; p++) {
    const int32_t u = ind[p];
    const float x = X[(int64_t)u * F + k];
    if (x > best) {
      best = x;
      who = u;
    }
  }
  const float degE_val = degE ? degE[e] : 1.f, W_val = W ? W[e] : 1.f;
  best *= degE_val * W_val;
  Xe[t] = best;
  record[t] = who;
}

// Backward of first_aggr = max (HGNNAggr_f1max_backward_kernel, hgnnaggr_cuda.cu:179-208):
// Y[record[e,k], k] += T[e,k] * degV[record[e,k]], T = the scaled hyperedge sums of grad.
__global__ __launch_bounds__(256) void scatter_record_kernel(int32_t M, int32_t F, const float *T,
                                                             const int32_t *record, const float *degV,
                                                             float *Y) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)M * F) return;
  const int32_t k = (int32_t)(t % F);
  const int64_t v = record[t];
  atomicAdd(Y + v * F + k, T[t] * (degV ? degV[v] : 1.f));
}

// hg_plan_bind_scales: degE[e], W[e] per slot and degV[v] per panel row, in record order.
__global__ __launch_bounds__(256) void bind_scales_kernel(int64_t nslots, const int32_t *eid_all,
                                                          const float *degE, const float *W, float *bsA,
                                                          float *bsB, int64_t nrows, const int32_t *prow,
                                                          const float *degV, float *bsD) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < nslots) {
    const int e = eid_all[t];  // -1: materialised row, already scaled
    bsA[t] = (degE && e >= 0) ? degE[e] : 1.f;
    bsB[t] = (W && e >= 0) ? W[e] : 1.f;
  }
  if (t < nrows) bsD[t] = degV ? degV[prow[t]] : 1.f;
}

hipError_t launch_bind_scales(int64_t nslots, const int32_t *eid_all, const float *degE, const float *W,
                              float *bsA, float *bsB, int64_t nrows, const int32_t *prow, const float *degV,
                              float *bsD, hipStream_t stream) {
  const int64_t n = std::max(nslots, nrows);
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(bind_scales_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, nslots,
                     eid_all, degE, W, bsA, bsB, nrows, prow, degV, bsD);
  return hipGetLastError();
}

hipError_t launch_gather_max(int32_t M, int32_t F, const int32_t *ptr, const int32_t *ind, const float *X,
                             const float *degE, const float *W, float *Xe, int32_t *record,
                             hipStream_t stream) {
  const int64_t n = (int64_t)M * F;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_max_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, M, F, ptr, ind,
                     X, degE, W, Xe, record);
  return hipGetLastError();
}

hipError_t launch_scatter_record(int32_t M, int32_t F, const float *T, const int32_t *record,
                                 const float *degV, float *Y, hipStream_t stream) {
  const int64_t n = (int64_t)M * F;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_record_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, M, F, T,
                     record, degV, Y);
  return hipGetLastError();
}

}  // namespace hg
