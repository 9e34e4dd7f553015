// nrt_rank_layout.h — layout constants of the rank stream (host builder: host/nrt_streams.cc; readers: kernels_nrt_rank.hip,
// kernels_nrt_fused.hip).  No HIP dependency: the host translation units include it too.
#pragma once

#include <cstdint>

namespace spx {

// rank-space Filter: chunk rows, comparison vectors per pod (pod-level, 8 containers, 4 sums), head dwords of a pod record
// (w0, w1, the slot sets of items 1..9, app containers a0 | a1 << 8 | a2 << 16 | count << 24, pad), largest chunk block
constexpr int kRkChunkRows = 32, kRkVectors = 13, kRkPodHead = 16;
// head dwords 12 / 13 of a pod record: one byte per container for the fused sweep — bits 0-2 the Filter status a misfit sets, then
constexpr uint32_t kRkOpMerge1 = 8;    // the second app container: its verdict for the zone a0 was charged to comes from vector 9
constexpr uint32_t kRkOpMerge3 = 16;   // the third: vectors 10 / 11 / 12 for the zones a0 / a1 / both were charged to
constexpr uint32_t kRkOpCharge0 = 32;  // the lowest fitting zone is remembered as a0's
constexpr uint32_t kRkOpCharge1 = 64;  // ... as a1's

}  // namespace spx
