// MatrixMarket reader with the reference DataLoader's semantics
// (HyperGsys/include/dataloader/dataloader.hpp:22-180): banner, `%` comment
// lines, size line, then nnz coordinate lines `row col [value]` (1-based; the
// value is parsed and dropped for real / integer fields).  `symmetric` files get
// their off-diagonal entries mirrored, then sorted and de-duplicated; everything
// else is sorted by (row, col) and KEEPS duplicates.  Output: H in CSR (rows =
// vertices) and its stable counting-sort transpose H_T (rows = hyperedges, members
// ascending) -- the pair the reference CLI hands to its kernels.  Pure C++.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "hg_internal.h"

namespace {

int32_t *dup(const std::vector<int32_t> &v) {
  int32_t *p = static_cast<int32_t *>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(int32_t)));
  if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(int32_t));
  return p;
}

std::string lower(std::string s) {
  for (char &c : s) c = (char)std::tolower((unsigned char)c);
  return s;
}

struct FileGuard {  // closes on every exit path, exceptions included
  FILE *f;
  explicit FileGuard(FILE *f_) : f(f_) {}
  ~FileGuard() {
    if (f) std::fclose(f);
  }
  FileGuard(const FileGuard &) = delete;
  FileGuard &operator=(const FileGuard &) = delete;
};

}  // namespace

extern "C" {

void hg_free(void *p) { std::free(p); }

int hg_mtx_read(const char *path, int32_t *nrow, int32_t *ncol, int64_t *nnz, int32_t **H_ptr,
                int32_t **H_ind, int32_t **HT_ptr, int32_t **HT_ind) {
  if (!path || !nrow || !ncol || !nnz || !H_ptr || !H_ind || !HT_ptr || !HT_ind) {
    hg::set_error("hg_mtx_read: null argument");
    return HG_ERR_INVALID;
  }
  *H_ptr = *H_ind = *HT_ptr = *HT_ind = nullptr;
  FILE *f = std::fopen(path, "r");
  if (!f) {
    hg::set_error(std::string("File ") + path + " not found");
    return HG_ERR_INVALID;
  }
  FileGuard guard(f);
  char line[1024];
  char banner[64], mtx[64], crd[64], field[64], symm[64];
  if (!std::fgets(line, sizeof(line), f) ||
      std::sscanf(line, "%63s %63s %63s %63s %63s", banner, mtx, crd, field, symm) != 5 ||
      std::strcmp(banner, "%%MatrixMarket") != 0 || lower(mtx) != "matrix" || lower(crd) != "coordinate") {
    hg::set_error("Could not process this file.");
    return HG_ERR_INVALID;
  }
  const std::string fld = lower(field);
  const bool has_value = fld == "real" || fld == "integer";
  const bool has_two = fld == "complex";
  const bool symmetric = lower(symm) == "symmetric";
  do {
    if (!std::fgets(line, sizeof(line), f)) {
      hg::set_error("Could not process this file.");
      return HG_ERR_INVALID;
    }
  } while (line[0] == '%');
  long long R = 0, C = 0, NZ = 0;
  if (std::sscanf(line, "%lld %lld %lld", &R, &C, &NZ) != 3 || R < 0 || C < 0 || NZ < 0 ||
      R > 0x7fffffffLL || C > 0x7fffffffLL || NZ > 0x7fffffffLL) {
    hg::set_error("Could not process this file.");
    return HG_ERR_INVALID;
  }
  std::vector<std::pair<int32_t, int32_t>> coords;
  try {
    coords.reserve((size_t)NZ * (symmetric ? 2 : 1));
    for (long long i = 0; i < NZ; i++) {
      long long r, c;
      double dummy;
      if (std::fscanf(f, "%lld %lld", &r, &c) != 2) {
        hg::set_error("Error: not enough rows in mtx file.");
        return HG_ERR_INVALID;
      }
      if (has_value && std::fscanf(f, "%lf", &dummy) != 1) dummy = 0;
      if (has_two && std::fscanf(f, "%lf %lf", &dummy, &dummy) != 2) dummy = 0;
      if (r < 1 || r > R || c < 1 || c > C) {
        hg::set_error("mtx entry out of range at line " + std::to_string(i));
        return HG_ERR_INVALID;
      }
      coords.emplace_back((int32_t)(r - 1), (int32_t)(c - 1));
      if (symmetric && r != c) coords.emplace_back((int32_t)(c - 1), (int32_t)(r - 1));
    }
    if (symmetric && R != C) {
      hg::set_error("symmetric mtx file must be square");
      return HG_ERR_INVALID;
    }
    std::sort(coords.begin(), coords.end());
    if (symmetric) coords.erase(std::unique(coords.begin(), coords.end()), coords.end());
    if (coords.size() > (size_t)0x7fffffff) {  // mirrored entries count too: int32 row pointers below
      hg::set_error("mtx file holds more than 2^31 - 1 entries after symmetric expansion");
      return HG_ERR_INVALID;
    }
    std::vector<int32_t> ptr((size_t)R + 1, 0), ind(coords.size());
    for (size_t i = 0; i < coords.size(); i++) {
      ptr[(size_t)coords[i].first + 1]++;
      ind[i] = coords[i].second;
    }
    for (long long r = 0; r < R; r++) ptr[r + 1] += ptr[r];
    std::vector<int32_t> t_ptr, t_ind;
    hg::transpose_csr((int32_t)R, (int32_t)C, ptr.data(), ind.data(), t_ptr, t_ind);
    *nrow = (int32_t)R;
    *ncol = (int32_t)C;
    *nnz = (int64_t)ind.size();
    *H_ptr = dup(ptr);
    *H_ind = dup(ind);
    *HT_ptr = dup(t_ptr);
    *HT_ind = dup(t_ind);
  } catch (const std::bad_alloc &) {
    hg::set_error("hg_mtx_read: host allocation failed");
    return HG_ERR_NOMEM;
  }
  if (!*H_ptr || !*H_ind || !*HT_ptr || !*HT_ind) {
    hg_free(*H_ptr);
    hg_free(*H_ind);
    hg_free(*HT_ptr);
    hg_free(*HT_ind);
    *H_ptr = *H_ind = *HT_ptr = *HT_ind = nullptr;
    hg::set_error("hg_mtx_read: host allocation failed");
    return HG_ERR_NOMEM;
  }
  return HG_OK;
}

}  // extern "C"
