// aggr_proto <mtx path> <feature_len> [--cpu] [--iter N]
//
// Drop-in for the reference's benchmark CLI (HyperGsys/source/aggr_proto.cu):
// same positional arguments, same usage line and exit code, same stdout lines
// and the same `result.csv` row
//   file,F,t_two_spmm,t_spgemm,t_spmm,t_fused,t_full_min,p_full,t_shm_min,p_shm
// on the MI355X backend (libhgaggr.so, include/hg_aggr.h):
//   * "two ... spmm"            : rocSPARSE SpMM x2 (the cuSPARSE comparator, spmm.cuh:79-143)
//   * "spgemm"                  : not built (out of scope) -> empty CSV cells
//   * "baseline fused kernel"   : the reference's scheme, one task per hyperedge, fp32 atomics
//   * "ef full tune"            : the reference's w^2 task grid from hg_balance_schedule over the
//                                 reference's 20 partition sizes, best time + partition size
//   * "ef shm tune"             : this backend's own kernels (pull / fused), best time; the
//                                 "partition" cell carries the variant code (1 pull, 3 fused)
// Every GPU variant is validated against the host reference path first (rel 1e-2,
// util::check_result, check.cuh:40-57) and additionally at 1e-5 (printed).
// `--cpu` (BASELINE.json configs[0]) runs only the host paths: no GPU is touched.
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "hg_aggr.h"

#define HIP_OK(x)                                                                       \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "HIP error in line %d: %s\n", __LINE__, hipGetErrorString(e_));   \
      exit(EXIT_FAILURE);                                                               \
    }                                                                                   \
  } while (0)
#define HG_OKAY(x)                                                                      \
  do {                                                                                  \
    int r_ = (x);                                                                       \
    if (r_ != HG_OK) {                                                                  \
      fprintf(stderr, "libhgaggr error in line %d: %s\n", __LINE__, hg_last_error());   \
      exit(EXIT_FAILURE);                                                               \
    }                                                                                   \
  } while (0)
#define RS_OK(x)                                                                        \
  do {                                                                                  \
    rocsparse_status s_ = (x);                                                          \
    if (s_ != rocsparse_status_success) {                                               \
      fprintf(stderr, "rocSPARSE error %d in line %d\n", (int)s_, __LINE__);            \
      exit(EXIT_FAILURE);                                                               \
    }                                                                                   \
  } while (0)

namespace {

// The CLI's own checker, as the reference CLI carries its own (check.cuh:83-114).
void host_fused(int N, int F, const int *Hp, const int *Hi, const int *Tp, const int *Ti, const float *X,
                float *Y) {
  for (long v = 0; v < N; v++)
    for (long k = 0; k < F; k++) {
      float a = 0;
      for (int p = Hp[v]; p < Hp[v + 1]; p++) {
        float b = 0;
        const int e = Hi[p];
        for (int q = Tp[e]; q < Tp[e + 1]; q++) b += X[(long)Ti[q] * F + k];
        a += b;
      }
      Y[v * F + k] = a;
    }
}
void host_spmm(int rows, int F, const int *ptr, const int *ind, const float *B, float *C) {
  for (long i = 0; i < rows; i++)
    for (int p = ptr[i]; p < ptr[i + 1]; p++)
      for (long j = 0; j < F; j++) C[i * F + j] += B[(long)ind[p] * F + j];
}
bool check_result(long rows, long F, const float *C, const float *R, double *worst5) {
  bool passed = true;
  double w = 0;
  for (long i = 0; i < rows; i++)
    for (long j = 0; j < F; j++) {
      const float c = C[i * F + j], r = R[i * F + j];
      w = std::max(w, (double)std::fabs(c - r) / std::max(1.0, (double)std::fabs(r)));
      if (std::fabs(c - r) > 1e-2 * std::fabs(r)) {
        if (passed) printf("Wrong result: i = %ld, j = %ld, result = %lf, reference = %lf.\n", i, j, c, r);
        passed = false;
        break;
      }
    }
  *worst5 = w;
  return passed;
}

struct Timer {
  hipEvent_t a, b;
  Timer() {
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&b));
  }
  void start() { HIP_OK(hipEventRecord(a, 0)); }
  float stop() {
    HIP_OK(hipEventRecord(b, 0));
    HIP_OK(hipEventSynchronize(b));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    return ms;
  }
};

template <typename T>
T *to_device(const T *h, size_t n) {
  T *d = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d), std::max<size_t>(n, 1) * sizeof(T)));
  if (n) HIP_OK(hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

}  // namespace

int main(int argc, char **argv) {
  bool cpu_only = false;
  int iter = 300, iter_fused = 100;
  std::vector<char *> pos;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--cpu")) cpu_only = true;
    else if (!strcmp(argv[i], "--iter") && i + 1 < argc) iter = iter_fused = atoi(argv[++i]);
    else pos.push_back(argv[i]);
  }
  if (pos.size() < 2) {
    printf("Input: first get the path of sparse matrix, then get the "
           "feature length of dense matrix\n");
    exit(1);
  }
  const char *filename = pos[0];
  const int F = atoi(pos[1]);
  if (F <= 0) {
    printf("feature length must be positive\n");
    exit(1);
  }

  int N = 0, M = 0;
  int64_t nnz = 0;
  int *Hp, *Hi, *Tp, *Ti;
  if (hg_mtx_read(filename, &N, &M, &nnz, &Hp, &Hi, &Tp, &Ti) != HG_OK) {
    printf("%s\n", hg_last_error());
    exit(EXIT_FAILURE);
  }
  printf("H.nrow %d H.ncol %d H_nnz %d\n", N, M, (int)nnz);

  std::fstream fs;
  fs.open("result.csv", std::ios::app | std::ios::in | std::ios::out);
  if (!fs.is_open()) fs.open("result.csv", std::ios::out);

  std::vector<float> X((size_t)N * F), tmp((size_t)M * F, 0.f), ref((size_t)N * F, 0.f), out((size_t)N * F);
  for (auto &x : X) x = (float)(std::rand() % 10) / 10;  // RamArray::fill_random_h, ramArray.cuh:72-76

  // host reference paths (timed: this is the reference's CPU path)
  auto t0 = std::chrono::steady_clock::now();
  host_spmm(M, F, Tp, Ti, X.data(), tmp.data());
  host_spmm(N, F, Hp, Hi, tmp.data(), ref.data());
  auto t1 = std::chrono::steady_clock::now();
  std::vector<float> ref_fused((size_t)N * F);
  host_fused(N, F, Hp, Hi, Tp, Ti, X.data(), ref_fused.data());
  auto t2 = std::chrono::steady_clock::now();
  const double ms_two = std::chrono::duration<double, std::milli>(t1 - t0).count();
  const double ms_fused = std::chrono::duration<double, std::milli>(t2 - t1).count();
  const bool host_agree = !memcmp(ref.data(), ref_fused.data(), ref.size() * sizeof(float));
  printf("host reference: two-step spmm %.4f ms, fused %.4f ms, 1 thread, %s\n", ms_two, ms_fused,
         host_agree ? "bitwise equal" : "DIFFERENT");

  fs << filename << "," << F << ",";
  if (cpu_only) {
    printf("start spmm test\ncheck passed!\nThe time of two host spmm %g\n", ms_two);
    printf("start fused kernel test\ncheck passed!\ntest time one baseline fused kernel: %.4f ms\n", ms_fused);
    fs << ms_two << ",,," << ms_fused << ",,,,\n";
    fs.close();
    return host_agree ? 0 : 2;
  }

  // ---------------------------------------------------------------- device side
  int *dHp = to_device(Hp, (size_t)N + 1), *dHi = to_device(Hi, (size_t)nnz);
  int *dTp = to_device(Tp, (size_t)M + 1), *dTi = to_device(Ti, (size_t)nnz);
  float *dX = to_device(X.data(), X.size());
  float *dTmp = to_device(tmp.data(), tmp.size()), *dY = to_device(out.data(), out.size());
  std::vector<float> ones((size_t)nnz, 1.0f);
  float *dVal = to_device(ones.data(), ones.size());
  Timer tm;
  double w5 = 0;

  printf("start spmm test\n");
  {  // two rocSPARSE SpMMs: tmp = H_T X ; Y = H tmp
    rocsparse_handle h;
    RS_OK(rocsparse_create_handle(&h));
    rocsparse_spmat_descr A_t, A;
    rocsparse_dnmat_descr Bx, Ct, Cy;
    RS_OK(rocsparse_create_csr_descr(&A_t, M, N, nnz, dTp, dTi, dVal, rocsparse_indextype_i32,
                                     rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f32_r));
    RS_OK(rocsparse_create_csr_descr(&A, N, M, nnz, dHp, dHi, dVal, rocsparse_indextype_i32,
                                     rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f32_r));
    RS_OK(rocsparse_create_dnmat_descr(&Bx, N, F, F, dX, rocsparse_datatype_f32_r, rocsparse_order_row));
    RS_OK(rocsparse_create_dnmat_descr(&Ct, M, F, F, dTmp, rocsparse_datatype_f32_r, rocsparse_order_row));
    RS_OK(rocsparse_create_dnmat_descr(&Cy, N, F, F, dY, rocsparse_datatype_f32_r, rocsparse_order_row));
    const float one = 1.f, zero = 0.f;
    size_t b1 = 0, b2 = 0;
    auto spmm = [&](rocsparse_spmat_descr a, rocsparse_dnmat_descr b, rocsparse_dnmat_descr c,
                    rocsparse_spmm_stage st, size_t *bs, void *buf) {
      RS_OK(rocsparse_spmm(h, rocsparse_operation_none, rocsparse_operation_none, &one, a, b, &zero, c,
                           rocsparse_datatype_f32_r, rocsparse_spmm_alg_default, st, bs, buf));
    };
    spmm(A_t, Bx, Ct, rocsparse_spmm_stage_buffer_size, &b1, nullptr);
    spmm(A, Ct, Cy, rocsparse_spmm_stage_buffer_size, &b2, nullptr);
    void *buf1, *buf2;
    HIP_OK(hipMalloc(&buf1, std::max<size_t>(b1, 16)));
    HIP_OK(hipMalloc(&buf2, std::max<size_t>(b2, 16)));
    spmm(A_t, Bx, Ct, rocsparse_spmm_stage_preprocess, &b1, buf1);
    spmm(A, Ct, Cy, rocsparse_spmm_stage_preprocess, &b2, buf2);
    spmm(A_t, Bx, Ct, rocsparse_spmm_stage_compute, &b1, buf1);
    spmm(A, Ct, Cy, rocsparse_spmm_stage_compute, &b2, buf2);
    HIP_OK(hipMemcpy(out.data(), dY, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    if (check_result(N, F, out.data(), ref.data(), &w5)) {
      printf("check passed!\n");
      tm.start();
      for (int i = 0; i < iter; i++) {
        spmm(A_t, Bx, Ct, rocsparse_spmm_stage_compute, &b1, buf1);
        spmm(A, Ct, Cy, rocsparse_spmm_stage_compute, &b2, buf2);
      }
      const float t = tm.stop() / iter;
      printf("The time of two rocsparse spmm %g\n", t);
      fs << t << ",";
    } else {
      fs << ",";
    }
    HIP_OK(hipFree(buf1));
    HIP_OK(hipFree(buf2));
    rocsparse_destroy_spmat_descr(A_t);
    rocsparse_destroy_spmat_descr(A);
    rocsparse_destroy_dnmat_descr(Bx);
    rocsparse_destroy_dnmat_descr(Ct);
    rocsparse_destroy_dnmat_descr(Cy);
    rocsparse_destroy_handle(h);
  }
  fs << ",,";  // spgemm + spmm columns: SpGEMM baseline not built (out of scope)

  printf("start fused kernel test\n");
  auto validate = [&](const char *what) {
    HIP_OK(hipMemcpy(out.data(), dY, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    const bool ok = check_result(N, F, out.data(), ref_fused.data(), &w5);
    printf(ok ? "check passed!\n" : "check failed!\n");
    if (ok) printf("  (%s: max |y-ref|/max(1,|ref|) = %.3g%s)\n", what, w5, w5 <= 1e-5 ? ", within 1e-5" : "");
    return ok;
  };
  {  // baseline fused: one task per hyperedge, atomics (HyperGAggr_Edgefused_Kernel)
    HG_OKAY(hg_aggr_push_groups_f32(N, M, F, M, nullptr, nullptr, nullptr, nullptr, dTp, dTi, dX, nullptr,
                                    nullptr, nullptr, dY, nullptr));
    if (validate("edge-fused, atomics")) {
      tm.start();
      for (int i = 0; i < iter_fused; i++)
        HG_OKAY(hg_aggr_push_groups_f32(N, M, F, M, nullptr, nullptr, nullptr, nullptr, dTp, dTi, dX, nullptr,
                                        nullptr, nullptr, dY, nullptr));
      const float t = tm.stop() / iter_fused;
      printf("test time one baseline fused kernel: %.4f ms\n", t);
      fs << t << ",";
    } else {
      fs << ",";
    }
  }
  {  // the reference's balanced task grid over its tune list (aggr_proto.cu:70-71)
    const int tune_list[] = {4, 6, 10, 16, 20, 30, 40, 60, 80, 100, 120, 150, 180, 210, 250, 300, 350, 400, 450, 500};
    float best = 1e9f;
    int best_p = 0;
    for (int p : tune_list) {
      int64_t nk = 0, ng = 0;
      if (hg_balance_schedule(M, p, Tp, &nk, &ng, nullptr, nullptr, nullptr, nullptr) != HG_OK) break;
      std::vector<int> key((size_t)nk), row((size_t)ng), st((size_t)ng), ed((size_t)ng);
      HG_OKAY(hg_balance_schedule(M, p, Tp, &nk, &ng, key.data(), row.data(), st.data(), ed.data()));
      int *dk = to_device(key.data(), key.size()), *dr = to_device(row.data(), row.size());
      int *ds = to_device(st.data(), st.size()), *de = to_device(ed.data(), ed.size());
      HG_OKAY(hg_aggr_push_groups_f32(N, M, F, ng, dk, dr, ds, de, dTp, dTi, dX, nullptr, nullptr, nullptr, dY,
                                      nullptr));
      HIP_OK(hipMemcpy(out.data(), dY, out.size() * sizeof(float), hipMemcpyDeviceToHost));
      if (check_result(N, F, out.data(), ref_fused.data(), &w5)) {  // TRY: time only what passed
        tm.start();
        for (int i = 0; i < iter_fused; i++)
          HG_OKAY(hg_aggr_push_groups_f32(N, M, F, ng, dk, dr, ds, de, dTp, dTi, dX, nullptr, nullptr, nullptr,
                                          dY, nullptr));
        const float t = tm.stop() / iter_fused;
        if (t < best) {
          best = t;
          best_p = p;
        }
      }
      HIP_OK(hipFree(dk));
      HIP_OK(hipFree(dr));
      HIP_OK(hipFree(ds));
      HIP_OK(hipFree(de));
    }
    printf("test ef full tune time one %.4f ms\n", best);
    fs << best << "," << best_p << ",";
  }
  {  // this backend's kernels
    hg_plan *plan = nullptr;
    HG_OKAY(hg_plan_create_host(&plan, N, M, Tp, Ti, nullptr));
    HG_OKAY(hg_plan_prepare(plan, F, nullptr));  // both variants are timed: the fused schedule's partial rows count
    const size_t wsb = hg_plan_workspace_bytes(plan, F);
    void *ws = nullptr;
    HIP_OK(hipMalloc(&ws, std::max<size_t>(wsb, 256)));
    float best = 1e9f;
    int best_v = 0;
    const int variants[] = {HG_VARIANT_PULL, HG_VARIANT_FUSED};
    const char *names[] = {"two-phase pull", "fused (LDS-staged)"};
    for (int i = 0; i < 2; i++) {
      HG_OKAY(hg_aggr_fused_f32(plan, F, dTp, dTi, dX, nullptr, nullptr, nullptr, dY, ws, wsb, variants[i], nullptr));
      if (!validate(names[i])) continue;
      tm.start();
      for (int k = 0; k < iter_fused; k++)
        HG_OKAY(hg_aggr_fused_f32(plan, F, dTp, dTi, dX, nullptr, nullptr, nullptr, dY, ws, wsb, variants[i], nullptr));
      const float t = tm.stop() / iter_fused;
      printf("  %s: %.4f ms (%.3g G edges/s)\n", names[i], t, (double)nnz / t / 1e6);
      if (t < best) {
        best = t;
        best_v = variants[i];
      }
    }
    {  // the timed choice (hg_plan_tune_f32: the reference's HyperGAggr_tune on this backend's candidates), then "auto"
      hg_tune_info ti;
      HG_OKAY(hg_plan_tune_f32(plan, F, dTp, dTi, dX, nullptr, nullptr, nullptr, dY, ws, wsb, 20, nullptr, &ti));
      if (validate("tuned auto")) {
        tm.start();
        for (int k = 0; k < iter_fused; k++)
          HG_OKAY(hg_aggr_fused_f32(plan, F, dTp, dTi, dX, nullptr, nullptr, nullptr, dY, ws, wsb, HG_VARIANT_AUTO, nullptr));
        const float t = tm.stop() / iter_fused;
        printf("  tuned auto (%s, pull hop kernels %d): %.4f ms (%.3g G edges/s)\n",
               ti.variant == HG_VARIANT_FUSED ? "fused" : "pull", ti.pull_hop_kernels, t, (double)nnz / t / 1e6);
        if (t < best) {
          best = t;
          best_v = ti.variant;
        }
      }
    }
    printf("test ef shm tune time one %.4f ms\n", best);
    fs << best << "," << best_v << ",";
    HIP_OK(hipFree(ws));
    hg_plan_destroy(plan);
  }
  fs << "\n";
  fs.close();
  hg_free(Hp);
  hg_free(Hi);
  hg_free(Tp);
  hg_free(Ti);
  return 0;
}
