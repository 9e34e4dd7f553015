// nrt_streams.hpp — the host-built streams of the NodeResourceTopologyMatch sweeps (pure host code, no device involved):
//   nrt_build_items        the pod record stream of the float64 formulation (kernels_nrt_fast.hip / kernels_nrt_fused.hip)
//   nrt_build_classes      pod equivalence classes over it
//   nrt_build_rank_stream  what a chunk's pods ask for as RANKS (kernels_nrt_rank.hip, kernels_nrt_fused.hip)
// Round 6: moved out of csrc/spx_engine.hip (the engine's translation unit had grown to 3 900 lines); the engine calls them from
// spx_upload_nrt_pods and from the lazily built every-row rank stream, tests reach them through spx_internal_nrt_pod_classes.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/spx.h"
#include "../csrc/nrt_rank_layout.h"

namespace spx_host {

// per slot, Value() form: OR and maximum of the zone capacities / of the requests in place (the packed float32 Score's preconditions,
// nrt_packed_score in the engine; delta uploads only ever add to them)
struct NrtQty {
  uint64_t bits[SPX_NRT_MAX_RES] = {0};
  int64_t most[SPX_NRT_MAX_RES] = {0};
  void add(int r, int64_t v) { bits[r] |= static_cast<uint64_t>(v), most[r] = v > most[r] ? v : most[r]; }
  void merge(const NrtQty& o) {
    for (int r = 0; r < SPX_NRT_MAX_RES; ++r) bits[r] |= o.bits[r], most[r] = o.most[r] > most[r] ? o.most[r] : most[r];
  }
};

// quantities the float64 NRT kernel may hold exactly, with room for x100 and the reciprocal trick
constexpr int64_t kNrtFastLimit = int64_t{1} << 42;
constexpr int64_t kNrtWeightLimit = int64_t{1} << 20;  // sum of the NRT scoring weights the float64 formulation accepts
inline bool nrt_fast_qty(int64_t v) { return v >= 0 && v < kNrtFastLimit; }
// RN(1/v) * (1 + 2^-49): floor(num * rc) == num / v for 0 <= num <= 101 * v, 0 < v < 2^42 (kernels_nrt_fast.hip)
inline double nrt_biased_rcp(double v) { return v > 0.0 ? (1.0 / v) * (1.0 + 0x1p-49) : 0.0; }
inline int64_t nrt_value_of(bool is_cpu, int64_t q) { return is_cpu ? (q + 999) / 1000 : q; }
// a quantity the float32 BalancedAllocation Score holds exactly: below 2^24, or any integer whose float32 image is itself (hugepage
// and device-memory quantities are small multiples of a power of two: 3 x 2^30 is as exact in float32 as 3).  Slots whose requests
// and capacities are all of that kind compare "request > capacity" exactly; the others (memory in bytes) are undecided near equality
inline bool nrt_exact_f32(double v) { return v >= 0.0 && v < 9.2e18 && static_cast<double>(static_cast<float>(v)) == v; }

void nrt_build_items(const spx_nrt_pods_soa* t, const uint8_t* slot_flags, int cpu_slot, const std::vector<double>& wtab, uint32_t* items, bool* ok_out,
                     uint32_t* big_out, uint64_t* hash_out, NrtQty* qty_out = nullptr);
void nrt_build_classes(const uint32_t* items, const uint64_t* hash, size_t p, size_t R, int32_t* rep);
void nrt_build_rank_stream(const uint32_t* items, const int32_t* list, size_t n_list, size_t R, std::vector<uint32_t>& words, std::vector<uint32_t>& off,
                           std::vector<uint32_t>& first_out, uint32_t* max_dwords_out, bool* ok_out, bool* all_narrow_out, bool narrow_ok = true);

}  // namespace spx_host
