// nrt_rank_device.h — the rank-space image of a node's zone table (NodeResourceTopologyMatch.Filter without compares), shared by the
// rank-space Filter launch (kernels_nrt_rank.hip) and the fused Filter + Score sweep (kernels_nrt_fused.hip).  What the counts are and
// why a subtraction decides "available >= request": the header of kernels_nrt_rank.hip.
#pragma once

#include "spx_internal.h"

namespace spx {
namespace nrtdev {

// Two layouts of the eight zones' counts (round 5).  WIDE: two zones per dword under guard bits 15 / 31 (zone j in the low half of dword j,
// zone j + 4 in the high half): lists of up to 32 767 quantities.  NARROW: four zones per dword under guard bits 7 / 15 / 23 / 31 (zones 0-3 in
// dword 0, zones 4-7 in dword 1) when every list of the chunk has at most 127 entries (the engine marks such chunks: header dword 9, and
// replicates the thresholds into four bytes) — half the subtract / and instructions per comparison vector.  A byte (halfword) holds
// count | guard >= 128 (32 768) and the subtrahend is at most 127 (32 767): no borrow crosses a field.
template <bool NARROW>
struct RkLayout {
  static constexpr int W = NARROW ? 2 : 4;
  static constexpr uint32_t G = NARROW ? 0x80808080u : 0x80008000u;
  static constexpr uint32_t kOne = NARROW ? 0x01010101u : 0x00010001u;  // "count >= 1" in every field
};

// the lowest zone of m as a packed one-zone set (same layout as m); all zero when m is empty
__device__ __forceinline__ void lowest_zone(const uint32_t (&m)[4], uint32_t (&z)[4]) {  // WIDE: zone z = dword z & 3, half z >> 2
  // bits 0..3 = zones 0..3, bits 16..19 = zones 4..7
  const uint32_t w = (m[0] >> 15) | (m[1] >> 14) | (m[2] >> 13) | (m[3] >> 12);
  const uint32_t m8 = (w | (w >> 12)) & 0xffu;
  const uint32_t low = m8 & (0u - m8);
  const uint32_t w2 = (low | (low << 12)) & 0x000f000fu;
#pragma unroll
  for (int j = 0; j < 4; ++j) z[j] = (w2 << (15 - j)) & 0x80008000u;
}
__device__ __forceinline__ void lowest_zone(const uint32_t (&m)[2], uint32_t (&z)[2]) {  // NARROW: zone z = dword z >> 2, byte z & 3
  // guard bits 7 / 15 / 23 / 31 -> bits 28..31 of the product (2^21 + 2^14 + 2^7 + 1: the four wanted partial products land there,
  // every other one elsewhere or outside the 32 bits, no two on one bit: no carries)
  constexpr uint32_t kGather = 0x00204081u;
  const uint32_t lo4 = ((m[0] & 0x80808080u) * kGather) >> 28, hi4 = ((m[1] & 0x80808080u) * kGather) >> 28;
  const uint32_t m8 = lo4 | (hi4 << 4);
  const uint32_t low = m8 & (0u - m8);
  // a one-hot nibble bit k -> bit 8 k + 7: k + 7 k is one of the product's four bits k + {0, 7, 14, 21}, the only one on a byte's bit 0
  z[0] = (((low & 0xfu) * kGather) & 0x01010101u) << 7;
  z[1] = (((low >> 4) * kGather) & 0x01010101u) << 7;
}

}  // namespace nrtdev
}  // namespace spx
