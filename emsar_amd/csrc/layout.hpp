// layout.hpp -- what every layout builder shares: CSR validation and the length classes of the row sort (pure C++, no HIP calls).
//
// Input is the incidence CT as CSR (what scan_rshbucket builds as ragged arrays,
// /root/reference/src/emsar_functions.c:2135-2192).  The EM only needs, per row, the multiset of its transcript ids, so rows
// may be stored in any order; layout_tiled.hpp chooses the order for the hardware.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace emsar {

constexpr int kLenClasses = 64;

inline int len_class(int64_t len) {
    if (len <= 32) return (int)len;
    int64_t c = 32 + (len - 32 + 7) / 8;
    return (int)std::min<int64_t>(c, kLenClasses - 1);
}

// returns 0 ok, -1 malformed CSR
inline int validate_csr(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx) {
    if (n_rows < 0 || n_tx <= 0 || !row_ptr) return -1;
    if (row_ptr[0] != 0) return -1;
    for (int64_t r = 0; r < n_rows; r++)
        if (row_ptr[r + 1] < row_ptr[r]) return -1;
    uint64_t nnz = row_ptr[n_rows];
    if (nnz && !col_idx) return -1;
    for (uint64_t k = 0; k < nnz; k++)
        if (col_idx[k] < 0 || col_idx[k] >= n_tx) return -1;
    return 0;
}

}  // namespace emsar
