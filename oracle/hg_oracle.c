/*
 * hg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's algorithm for
 * the fused vertex->hyperedge->vertex aggregation path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (hypergef_amd/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - oracle_balance_schedule : PINNED by tests/golden/balancer_*.npz, which
 *     were produced by importing the reference's HyperGsys/balancer.py
 *     (tests/golden/make_golden.py).
 *   - aggregation functions   : PARITY UNPINNED.  The reference holds no golden
 *     vectors for this path, its Python checker needs
 *     dgl (absent) and its C++ host path includes <cuda_runtime.h> /
 *     <cusparse.h> (absent; stand-ins are not allowed), so nothing from the
 *     reference can be executed here to pin them.  They are line-by-line
 *     restatements of the cited functions and are cross-checked against two
 *     independent formulations (scipy CSR products and a pure-numpy
 *     index_add) in tests/test_oracle.py.
 *
 * Conventions follow the reference: Index = int32, DType = float32
 * (HyperGsys/include/util/check.cuh:11-12); H is N x M (vertex x hyperedge),
 * H_T is its transpose in CSR (row = hyperedge, entries = member vertices).
 * Offsets into feature matrices are computed in 64 bits (the reference's
 * int32 `v*F+k` is defect D9 and is not reproduced).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* util::spmm_reference_host, HyperGsys/include/util/check.cuh:61-79.         */
/* C_ref[i,:] += val * B[k,:] for every stored (i,k); the caller zeroes C_ref  */
/* (TwostepSpMM_host does, spmm.cuh:730-731).  csr_values may be NULL = ones.  */
ORACLE_API void oracle_spmm_csr(int32_t nrow, int32_t feature,
                                const int32_t *indptr, const int32_t *indices,
                                const float *values, const float *B,
                                float *C_ref) {
  for (int64_t i = 0; i < nrow; i++) {
    int32_t begin = indptr[i];
    int32_t end = indptr[i + 1];
    for (int32_t p = begin; p < end; p++) {
      int64_t k = indices[p];
      float val = values ? values[p] : 1.0f;
      for (int64_t j = 0; j < feature; j++) {
        C_ref[i * feature + j] += val * B[k * feature + j];
      }
    }
  }
}

/* TwostepSpMM_host, HyperGsys/include/spmm/spmm.cuh:724-740:                  */
/* tmp = H_T * X (M x F), out = H * tmp (N x F), both zeroed first.            */
ORACLE_API void oracle_twostep_host(int32_t N, int32_t M, int32_t feature,
                                    const int32_t *H_indptr,
                                    const int32_t *H_indices,
                                    const int32_t *HT_indptr,
                                    const int32_t *HT_indices, const float *X,
                                    float *tmp, float *out) {
  memset(tmp, 0, (size_t)M * feature * sizeof(float));
  memset(out, 0, (size_t)N * feature * sizeof(float));
  oracle_spmm_csr(M, feature, HT_indptr, HT_indices, NULL, X, tmp);
  oracle_spmm_csr(N, feature, H_indptr, H_indices, NULL, tmp, out);
}

/* util::hyperaggr_reference_host, HyperGsys/include/util/check.cuh:83-114.    */
/* Per vertex, per feature column, per incident hyperedge (H CSR order), per   */
/* member (H_T CSR order): fp32 B_acc then fp32 A_acc; result ASSIGNED.        */
ORACLE_API void oracle_hyperaggr_host(int32_t A_row, int32_t feature_size,
                                      const int32_t *A_indptr,
                                      const int32_t *A_indices,
                                      const int32_t *B_indptr,
                                      const int32_t *B_indices,
                                      const float *in_feature,
                                      float *out_feature) {
  for (int64_t A_row_idx = 0; A_row_idx < A_row; A_row_idx++) {
    int32_t A_lb = A_indptr[A_row_idx];
    int32_t A_hb = A_indptr[A_row_idx + 1];
    for (int64_t k_idx = 0; k_idx < feature_size; k_idx++) {
      float A_acc = 0;
      for (int32_t A_col_ptr = A_lb; A_col_ptr < A_hb; A_col_ptr++) {
        float B_acc = 0;
        int32_t B_row_idx = A_indices[A_col_ptr];
        int32_t B_lb = B_indptr[B_row_idx];
        int32_t B_hb = B_indptr[B_row_idx + 1];
        for (int32_t B_col_ptr = B_lb; B_col_ptr < B_hb; B_col_ptr++) {
          int64_t B_col_idx = B_indices[B_col_ptr];
          B_acc += in_feature[B_col_idx * feature_size + k_idx];
        }
        A_acc += B_acc;
      }
      out_feature[A_row_idx * feature_size + k_idx] = A_acc;
    }
  }
}

/* Same traversal parallelised over vertices: the "all host cores" CPU baseline */
/* of BASELINE.md section 3.  Per-vertex arithmetic is unchanged, so the result */
/* is bitwise equal to oracle_hyperaggr_host.                                   */
ORACLE_API void oracle_hyperaggr_host_omp(int32_t A_row, int32_t feature_size,
                                          const int32_t *A_indptr,
                                          const int32_t *A_indices,
                                          const int32_t *B_indptr,
                                          const int32_t *B_indices,
                                          const float *in_feature,
                                          float *out_feature) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t v = 0; v < A_row; v++) {
    int32_t A_lb = A_indptr[v];
    int32_t A_hb = A_indptr[v + 1];
    for (int64_t k = 0; k < feature_size; k++) {
      float A_acc = 0;
      for (int32_t p = A_lb; p < A_hb; p++) {
        float B_acc = 0;
        int32_t e = A_indices[p];
        for (int32_t q = B_indptr[e]; q < B_indptr[e + 1]; q++) {
          B_acc += in_feature[(int64_t)B_indices[q] * feature_size + k];
        }
        A_acc += B_acc;
      }
      out_feature[v * feature_size + k] = A_acc;
    }
  }
}

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* HGNN_check, test/hgnn_test.py:56-63 (the reference test's definition of     */
/* the weighted operator):                                                     */
/*   Xe = copy_u_sum(g1, X); Xe = Xe * degE; Xe *= W;                          */
/*   Xv = copy_u_sum(g2, Xe); Xv = Xv * degV                                   */
/* i.e. scale AFTER each sum.  degE / W / degV may be NULL (= factor skipped); */
/* that covers unignnaggrdeg (W NULL) and unignnaggr (all NULL).  Sums run in  */
/* CSR order (members ascending, incident hyperedges ascending).  Xe is the    */
/* caller's M x F scratch.                                                     */
ORACLE_API void oracle_hgnn_check(int32_t N, int32_t M, int32_t F,
                                  const int32_t *H_indptr,
                                  const int32_t *H_indices,
                                  const int32_t *HT_indptr,
                                  const int32_t *HT_indices, const float *X,
                                  const float *degE, const float *degV,
                                  const float *W, float *Xe, float *Y) {
  for (int64_t e = 0; e < M; e++) {
    for (int64_t k = 0; k < F; k++) {
      float acc = 0;
      for (int32_t p = HT_indptr[e]; p < HT_indptr[e + 1]; p++)
        acc += X[(int64_t)HT_indices[p] * F + k];
      if (degE) acc = acc * degE[e];
      if (W) acc *= W[e];
      Xe[e * F + k] = acc;
    }
  }
  for (int64_t v = 0; v < N; v++) {
    for (int64_t k = 0; k < F; k++) {
      float acc = 0;
      for (int32_t p = H_indptr[v]; p < H_indptr[v + 1]; p++)
        acc += Xe[(int64_t)H_indices[p] * F + k];
      if (degV) acc = acc * degV[v];
      Y[v * F + k] = acc;
    }
  }
}

/* Arithmetic of the reference's device kernel, HGNNAggr_forward_kernel,       */
/* HyperGsys/source/hgnnaggr/hgnnaggr_cuda.cu:28-45, with the atomics replayed  */
/* in ascending task order: B_acc *= degE*W; Y[v] += B_acc * degV[v].          */
/* Differs from oracle_hgnn_check only in where the roundings fall; kept so    */
/* tests can state the gap between the two reference definitions.              */
ORACLE_API void oracle_hgnn_kernel_order(int32_t N, int32_t M, int32_t F,
                                         const int32_t *HT_indptr,
                                         const int32_t *HT_indices,
                                         const float *X, const float *degE,
                                         const float *degV, const float *W,
                                         float *Y) {
  memset(Y, 0, (size_t)N * F * sizeof(float));
  for (int64_t e = 0; e < M; e++) {
    float degE_val = degE ? degE[e] : 1.0f;
    float W_val = W ? W[e] : 1.0f;
    for (int64_t k = 0; k < F; k++) {
      float B_acc = 0;
      for (int32_t p = HT_indptr[e]; p < HT_indptr[e + 1]; p++)
        B_acc += X[(int64_t)HT_indices[p] * F + k];
      B_acc *= degE_val * W_val;
      for (int32_t p = HT_indptr[e]; p < HT_indptr[e + 1]; p++) {
        int64_t v = HT_indices[p];
        float degV_val = degV ? degV[v] : 1.0f;
        Y[v * F + k] += B_acc * degV_val;
      }
    }
  }
}

/* first_aggr = mean: HGNNAggr_f1mean_forward_kernel,                          */
/* hgnnaggr_cuda.cu:86-113, with the hyperedge loop bounded by M (the          */
/* reference bounds it by N: defect D2).  B_acc *= degE*W/nnz.                 */
ORACLE_API void oracle_hgnn_mean(int32_t N, int32_t M, int32_t F,
                                 const int32_t *HT_indptr,
                                 const int32_t *HT_indices, const float *X,
                                 const float *degE, const float *degV,
                                 const float *W, float *Y) {
  memset(Y, 0, (size_t)N * F * sizeof(float));
  for (int64_t e = 0; e < M; e++) {
    int32_t start = HT_indptr[e], end = HT_indptr[e + 1];
    int32_t nnz = end - start;
    for (int64_t k = 0; k < F; k++) {
      float B_acc = 0;
      for (int32_t p = start; p < end; p++)
        B_acc += X[(int64_t)HT_indices[p] * F + k];
      B_acc *= degE[e] * W[e] / nnz;
      for (int32_t p = start; p < end; p++) {
        int64_t v = HT_indices[p];
        Y[v * F + k] += B_acc * degV[v];
      }
    }
  }
}

/* first_aggr = max: HGNNAggr_f1max_forward_kernel, hgnnaggr_cuda.cu:144-177   */
/* (init -1e5, strict >, record_max initial 0), hyperedge loop bounded by M.   */
ORACLE_API void oracle_hgnn_max(int32_t N, int32_t M, int32_t F,
                                const int32_t *HT_indptr,
                                const int32_t *HT_indices, const float *X,
                                const float *degE, const float *degV,
                                const float *W, float *Y,
                                int32_t *record_table) {
  memset(Y, 0, (size_t)N * F * sizeof(float));
  for (int64_t e = 0; e < M; e++) {
    int32_t start = HT_indptr[e], end = HT_indptr[e + 1];
    for (int64_t k = 0; k < F; k++) {
      float B_acc = -1e5;
      int32_t record_max = 0;
      for (int32_t p = start; p < end; p++) {
        int32_t u = HT_indices[p];
        float B_feat = X[(int64_t)u * F + k];
        if (B_feat > B_acc) {
          B_acc = B_feat;
          record_max = u;
        }
      }
      B_acc *= degE[e] * W[e];
      record_table[e * F + k] = record_max;
      for (int32_t p = start; p < end; p++) {
        int64_t v = HT_indices[p];
        Y[v * F + k] += B_acc * degV[v];
      }
    }
  }
}

/* balance_schedule.balancer, HyperGsys/balancer.py:15-33, and its C++ twin     */
/* hgnn_ef_full_balance_cpu, include/taskbalancer/balancer_kernel.cuh:229-259.  */
/* Two-call protocol: pass NULL arrays to get the counts, then call again with  */
/* arrays of that size.  n_key counts the trailing sentinel.  Returns 0, or -1  */
/* when nnz == 0 (the reference indexes an empty list there).                   */
ORACLE_API int oracle_balance_schedule(int32_t nrow, int32_t ngs,
                                       const int32_t *csrptr, int64_t *n_key,
                                       int64_t *n_group, int32_t *key,
                                       int32_t *row, int32_t *group_st,
                                       int32_t *group_ed) {
  int64_t nk = 0, ng = 0;
  int32_t work_p_sum = 0;
  int32_t last_key = -1;
  for (int32_t rid = 0; rid < nrow; rid++) {
    int32_t A_lb = csrptr[rid];
    int32_t A_hb = csrptr[rid + 1];
    int32_t workload = (A_hb - A_lb + ngs - 1) / ngs;
    int32_t tmp_key = A_lb;
    while (tmp_key < A_hb) {
      if (key) key[nk] = tmp_key;
      last_key = tmp_key;
      nk++;
      tmp_key += ngs;
    }
    for (int32_t i = 0; i < workload; i++) {
      for (int32_t j = 0; j < workload; j++) {
        if (row) {
          group_st[ng] = work_p_sum + j;
          group_ed[ng] = work_p_sum + i;
          row[ng] = rid;
        }
        ng++;
      }
    }
    work_p_sum += workload;
  }
  if (nk == 0) return -1;
  if (last_key != csrptr[nrow]) {
    if (key) key[nk] = csrptr[nrow];
    nk++;
  }
  *n_key = nk;
  *n_group = ng;
  return 0;
}

/* transpose + compressedRow, include/dataloader/dataloader.hpp:107-141: stable */
/* counting sort of H's COO by column -> H_T CSR with members ascending.        */
ORACLE_API void oracle_transpose_csr(int32_t nrow, int32_t ncol,
                                     const int32_t *indptr,
                                     const int32_t *indices, int32_t *t_indptr,
                                     int32_t *t_indices) {
  int32_t nnz = indptr[nrow];
  memset(t_indptr, 0, (size_t)(ncol + 1) * sizeof(int32_t));
  for (int32_t t = 0; t < nnz; t++) t_indptr[indices[t] + 1]++;
  for (int32_t c = 0; c < ncol; c++) t_indptr[c + 1] += t_indptr[c];
  int32_t *cursor = (int32_t *)malloc((size_t)(ncol + 1) * sizeof(int32_t));
  memcpy(cursor, t_indptr, (size_t)(ncol + 1) * sizeof(int32_t));
  for (int32_t r = 0; r < nrow; r++) {
    for (int32_t p = indptr[r]; p < indptr[r + 1]; p++) {
      t_indices[cursor[indices[p]]++] = r;
    }
  }
  free(cursor);
}

/* util::check_result, include/util/check.cuh:40-57: relative-only 1e-2 test,  */
/* stops at the first bad column of each row.  Returns 1 when passed.          */
ORACLE_API int oracle_check_result(int32_t M, int32_t N, const float *C,
                                   const float *C_ref) {
  int passed = 1;
  for (int64_t i = 0; i < M; i++) {
    for (int64_t j = 0; j < N; j++) {
      float c = C[i * N + j];
      float c_ref = C_ref[i * N + j];
      if (fabs(c - c_ref) > 1e-2 * fabs(c_ref)) {
        passed = 0;
        break;
      }
    }
  }
  return passed;
}

/* RamArray::fill_random_h, include/util/ramArray.cuh:72-76: (rand()%10)/10     */
/* from the process-default glibc sequence (the reference never seeds it).      */
ORACLE_API void oracle_fill_random(float *a, int64_t len, unsigned seed) {
  srand(seed);
  for (int64_t i = 0; i < len; i++) a[i] = (float)(rand() % 10) / 10;
}
