#include "planner.h"

#include <algorithm>
#include <sstream>

#include "../../include/viprs_hip.h"

namespace viprs {

int plan_blocks(int64_t m, const int32_t* lb, const int64_t* ip, bool low_memory,
                std::vector<Block>& blocks, std::string& err) {
    blocks.clear();
    if (m < 0) { err = "m must be non-negative"; return VIPRS_EINVAL; }
    if (m == 0) return VIPRS_OK;
    if (!lb || !ip) { err = "null LD index array"; return VIPRS_EINVAL; }

    // ---- bit-exact integer validation of the window contract ------------
    if (ip[0] != 0) {
        std::ostringstream s; s << "ld_indptr[0] must be 0, got " << ip[0];
        err = s.str(); return VIPRS_ELAYOUT;
    }
    for (int64_t j = 0; j < m; ++j) {
        const int64_t len = ip[j + 1] - ip[j];
        if (len < 0) {
            std::ostringstream s; s << "ld_indptr is not monotone at row " << j;
            err = s.str(); return VIPRS_ELAYOUT;
        }
        const int64_t l = lb[j];
        if (l < 0 || l + len > m) {
            std::ostringstream s;
            s << "row " << j << ": window [" << l << ", " << l + len << ") falls outside [0, " << m << ")";
            err = s.str(); return VIPRS_ELAYOUT;
        }
    }

    // ---- connected components of the windows ------------------------------------------------
    // SNP j touches q on ext(j) = [min(j, lb_j), max(j + 1, lb_j + len_j)).  Merge overlapping
    // ext intervals; because ext(j) always contains j, components are contiguous SNP ranges.
    // A window may reach back into earlier components, hence the stack.
    std::vector<std::pair<int64_t, int64_t>> comp;  // [start, end)
    for (int64_t j = 0; j < m; ++j) {
        const int64_t len = ip[j + 1] - ip[j];
        int64_t lo = j, hi = j + 1;
        if (len > 0) {
            lo = std::min<int64_t>(lo, lb[j]);
            hi = std::max<int64_t>(hi, (int64_t)lb[j] + len);
        }
        while (!comp.empty() && comp.back().second > lo) {
            lo = std::min(lo, comp.back().first);
            hi = std::max(hi, comp.back().second);
            comp.pop_back();
        }
        comp.emplace_back(lo, hi);
    }
    // comp is now sorted, disjoint and covers [0, m) (every j belongs to its own ext(j)).

    blocks.reserve(comp.size());
    for (const auto& c : comp) {
        Block b;
        b.start = c.first;
        b.end = c.second;
        b.nnz = ip[b.end] - ip[b.start];
        bool dense = true;
        if (!low_memory) {
            for (int64_t j = b.start; j < b.end && dense; ++j) {
                const int64_t len = ip[j + 1] - ip[j];
                dense = (lb[j] == b.start) && (len == b.end - b.start);
            }
            b.kind = dense ? VIPRS_BLOCK_DENSE_SYM : VIPRS_BLOCK_RAGGED;
        } else {
            for (int64_t j = b.start; j < b.end && dense; ++j) {
                const int64_t len = ip[j + 1] - ip[j];
                // an empty last row carries no information in left_bound
                dense = (len == b.end - j - 1) && (len == 0 || lb[j] == j + 1);
            }
            b.kind = dense ? VIPRS_BLOCK_DENSE_UPPER : VIPRS_BLOCK_RAGGED;
        }
        blocks.push_back(b);
    }
    return VIPRS_OK;
}

}  // namespace viprs
