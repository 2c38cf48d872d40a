// Host-side LD block planner: validates the (left_bound, indptr) window arrays and partitions
// the SNPs into independent LD blocks.  Pure C++ (no HIP), so it runs -- and is tested -- on
// machines without a GPU.
//
// The reference has no explicit block loop: VIPRS.e_step() (VIPRS.py:393-422) hands a whole
// chromosome to e_step<T,U,I> (e_step.hpp:343-442), and independence between LDetect blocks is
// implicit in the row windows  win(j) = [left_bound[j], left_bound[j] + indptr[j+1]-indptr[j])
// (e_step.hpp:389-392).  SNP j reads q[j] and writes q[win(j)] (+ q[j] itself), so two SNPs can
// be processed independently iff they are not linked through a chain of overlapping windows.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace viprs {

struct Block {
    int64_t start;     // first SNP of the block
    int64_t end;       // one past the last SNP
    int32_t kind;      // viprs_block_kind
    int64_t nnz;       // LD entries stored for the rows of this block
};

// Returns 0 on success; VIPRS_EINVAL / VIPRS_ELAYOUT with a message in `err` otherwise.
int plan_blocks(int64_t m, const int32_t* left_bound, const int64_t* indptr, bool low_memory,
                std::vector<Block>& blocks, std::string& err);

}  // namespace viprs
