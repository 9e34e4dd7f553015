// kernels_nrt_rank.hip — NodeResourceTopologyMatch.Filter in rank space (round 4).
//
// Reference: pkg/noderesourcetopology/filter.go:42-245 — singleNUMAContainerLevelHandler (:42-91), resourcesAvailableInAnyNUMANodes
// (:93-163), singleNUMAPodLevelHandler (:165-184), Filter (:186-245) — and subtractResourcesFromNUMANodeList numaresources.go:145-182.
//
// Why.  The float64 Filter launch (k_nrt_fast<RM, *, kPhFilter>) is issue-bound on v_cmp_le_f64 + v_addc per (zone, resource) —
// compares issue at the float64 rate whatever their width — and on the container-scope handler's zone-table
// mutation (8 fma per charged resource, and again to undo it).  Nothing the Filter decides needs the quantities themselves, only
// their ORDER against the requests of the 32 pods the block walks.  So, per chunk of 32 pod rows, the engine sorts the distinct
// quantities the chunk's pods compare (per resource slot, behind a leading 0: nrt_build_rank_stream, spx_engine.hip); at block
// start every lane counts, for each of its node's (zone, resource) cells, how many of them the cell's available quantity reaches
// (a binary search in LDS: 2-9 steps) and packs the counts two zones per dword under guard bits; from then on
//     available >= request   <=>   count >= position(request) + 1   <=>   ((count | 0x8000) - (position + 1)) keeps bit 15
// — 4 v_sub_u32 + 4 v_and_b32 per resource for all eight zones, at the full VALU rate.
//
// Charging without mutation.  The reference subtracts an app container's request from the zone it was placed on before the next
// container is tested.  available - charged >= request  <=>  available >= charged + request (integers, exact), and every lane
// sees the same pod: the sums a zone may carry are pod-level quantities, so they are in the chunk's lists too — per pod 13
// comparison vectors: [0] the pod-level request, [1..8] the containers, [9] a1 + a0, [10] a2 + a0, [11] a2 + a1, [12] a2 + a0 + a1
// (a_i = the i-th app container; a charged container adds only the resources it was itself compared on).  The second app container
// takes its verdict for the zone a0 went to from vector 9 and for the others from its own; the third selects among four.  Only the
// last app container is never charged (nothing reads the table after it).  Pods with more than three app containers keep the
// float64 Filter (the engine then builds no stream).  The node tables are read once per block and never written: no undo pass.
//
// Output: the status table, byte for byte what k_nrt_fast writes (tests/test_gpu_exhaustive.py compares every cell of config #3 /
// #5 with the oracle; tests/test_gpu_nrt.py::test_rank_filter_equals_float64_filter compares the two launches directly).
#include <hip/hip_runtime.h>

#include "spx_internal.h"
#include "nrt_fast_device.h"
#include "nrt_rank_device.h"

namespace spx {

namespace {

using namespace nrtdev;

// the 8 zones' verdicts for one comparison vector: guard bits of m[]
template <int RM, bool NARROW>
__device__ __forceinline__ void rank_mask(const uint32_t (&qa)[RM][RkLayout<NARROW>::W], const uint32_t (&fillp)[RM], uint32_t host_level, const uint32_t* thr,
                                          uint32_t fit, uint32_t always, uint32_t (&m)[RkLayout<NARROW>::W]) {
  using L = RkLayout<NARROW>;
  const uint32_t need = fit | always;
#pragma unroll
  for (int j = 0; j < L::W; ++j) m[j] = L::G;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((need >> r) & 1u)) continue;  // uniform
    SPX_KEEP_BRANCH();
    // a non-Guaranteed pod's NUMA-affine request: any reporting zone suits (filter.go:120-129) — count >= 1
    const uint32_t tt = ((always >> r) & 1u) ? L::kOne : thr[r];
#pragma unroll
    for (int j = 0; j < L::W; ++j) {
      uint32_t x = qa[r][j] - tt;
      if ((host_level >> r) & 1u) x |= fillp[r];  // uniform: a host-level resource no zone reports is not checked (filter.go:110-116)
      m[j] &= x;
    }
  }
}

template <int W>
__device__ __forceinline__ bool any_zone(const uint32_t (&m)[W]) {
  uint32_t o = m[0];
#pragma unroll
  for (int j = 1; j < W; ++j) o |= m[j];
  return o != 0u;
}

// the pod loop of k_nrt_filter_rank in one of the two count layouts (chosen per chunk: block-uniform)
template <int RM, bool NARROW>
__device__ __forceinline__ void rank_walk(const uint32_t (&q4)[RM][4], uint32_t fill_bits, uint32_t host_level, const uint32_t* pods, int rows, int lane,
                                          bool w_pod, bool w_ctr, bool aligned, bool pod_scope, uint32_t node_present, uint32_t st_stale, bool in, int pos,
                                          uint32_t* stage) {
  using L = RkLayout<NARROW>;
  constexpr int W = L::W;
  constexpr int PWR = kRkPodHead + kRkVectors * RM;
  uint32_t qa[RM][W], fillp[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) {
#pragma unroll
    for (int j = 0; j < W; ++j) qa[r][j] = q4[r][j];
    fillp[r] = ((fill_bits >> r) & 1u) ? L::G : 0u;
  }
  uint32_t acc_status = 0;
  for (int p = 0; p < rows; ++p) {
    const uint32_t* rec = pods + p * PWR;
    // the head: lane l holds dword l & 15; the fields become scalars as they are needed
    const uint32_t hv = rec[lane & 15];
    auto head = [&](int i) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(hv), i)); };
    const uint32_t w0 = head(0);
    const int qos = w0 & 0xffu;
    const bool non_native = ((w0 >> 8) & 0xffu) != 0;
    const int n_ctr = (w0 >> 16) & 0xffu;
    const int last_app = static_cast<int>(w0 >> 24) == 0xff ? -1 : static_cast<int>(w0 >> 24);
    const bool filtered = !(qos == SPX_QOS_BESTEFFORT && !non_native);  // filter.go:186-190
    uint32_t status = filtered ? st_stale : 0u;
    if (filtered) {  // uniform
      if (w_pod && pod_scope && aligned) {  // singleNUMAPodLevelHandler
        const uint32_t s = head(2);
        const uint32_t fit = (s >> 8) & 0xffu, always = (s >> 16) & 0xffu;
        uint32_t m[W];
        rank_mask<RM, NARROW>(qa, fillp, host_level, rec + kRkPodHead, fit, always, m);
        const bool ok = ((fit | always) & ~node_present) == 0 && any_zone(m);
        if (!ok) status = SPX_NRT_ST_POD;
      }
      if (w_ctr && !pod_scope && aligned) {  // singleNUMAContainerLevelHandler, containers in order (init containers first)
        const uint32_t apps = head(11);
        const int a0 = apps & 0xffu, a1 = (apps >> 8) & 0xffu;
        uint32_t z0[W], z1[W];
#pragma unroll
        for (int j = 0; j < W; ++j) z0[j] = z1[j] = 0u;  // the zones app containers a0 / a1 were charged to (packed one-zone sets)
        for (int c = 0; c < n_ctr; ++c) {
          const uint32_t s = head(3 + c);
          const uint32_t fit = (s >> 8) & 0xffu, always = (s >> 16) & 0xffu, kind = s >> 24;
          const uint32_t* own = rec + kRkPodHead + (1 + c) * RM;
          uint32_t m[W];
          rank_mask<RM, NARROW>(qa, fillp, host_level, own, fit, always, m);
          if (kind == SPX_CTR_APP && c != a0 && fit != 0) {  // uniform: an earlier app container may have been charged to a zone
            if (c == a1) {
              uint32_t ms[W];
              rank_mask<RM, NARROW>(qa, fillp, host_level, rec + kRkPodHead + 9 * RM, fit, always, ms);
#pragma unroll
              for (int j = 0; j < W; ++j) m[j] = (m[j] & ~z0[j]) | (ms[j] & z0[j]);
            } else {  // the third app container: its own vector, + a0, + a1, + both
              uint32_t m0[W], m1[W], mb[W];
              rank_mask<RM, NARROW>(qa, fillp, host_level, rec + kRkPodHead + 10 * RM, fit, always, m0);
              rank_mask<RM, NARROW>(qa, fillp, host_level, rec + kRkPodHead + 11 * RM, fit, always, m1);
              rank_mask<RM, NARROW>(qa, fillp, host_level, rec + kRkPodHead + 12 * RM, fit, always, mb);
#pragma unroll
              for (int j = 0; j < W; ++j) {
                const uint32_t both = z0[j] & z1[j];
                uint32_t x = (m[j] & ~z1[j]) | (m1[j] & z1[j]);
                x = (x & ~z0[j]) | (m0[j] & z0[j]);
                m[j] = (x & ~both) | (mb[j] & both);
              }
            }
          }
          const bool ok = ((fit | always) & ~node_present) == 0 && any_zone(m);
          const bool live = status == 0;
          if (kind != SPX_CTR_APP) {
            if (live && !ok) status = kind == SPX_CTR_SIDECAR ? SPX_NRT_ST_SIDECAR_CONTAINER : SPX_NRT_ST_INIT_CONTAINER;
          } else {
            if (live && !ok) status = SPX_NRT_ST_CONTAINER;
            if (c != last_app && fit != 0) {  // uniform: the zone the container is charged to (the lowest that fits), for its successors
              uint32_t z[W];
              lowest_zone(m, z);
              const bool apply = live && ok;
#pragma unroll
              for (int j = 0; j < W; ++j) {
                if (c == a0) z0[j] = apply ? z[j] : 0u;
                else z1[j] = apply ? z[j] : 0u;
              }
            }
          }
        }
      }
    }
    acc_status |= status << (8 * (p & 3));
    if ((p & 3) == 3 || p + 1 == rows) {  // uniform
      if (in) stage[(p >> 2) * kWindow + pos] = acc_status;
      acc_status = 0;
    }
  }
}

// dynamic LDS: the chunk block (header, lists, pod records), then the staged status dwords [kPodsPerUnit / 4][kWindow]
template <int RM>
__global__ __launch_bounds__(256, 4) void k_nrt_filter_rank(NrtArgs a, int n_tiles) {
  extern __shared__ __align__(16) uint32_t lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_windows = n_tiles;
  int window;
  int64_t chunk;
  if (n_windows >= kXcdMapWindows) {  // as k_nrt_fast: every XCD walks its own node windows
    const int wpx = (n_windows + 7) >> 3;
    const int64_t seq = blockIdx.x >> 3;
    window = static_cast<int>(blockIdx.x & 7u) + 8 * static_cast<int>(seq % wpx);
    chunk = seq / wpx;
    if (window >= n_windows) return;
  } else {
    window = static_cast<int>(blockIdx.x % n_windows);
    chunk = blockIdx.x / n_windows;
  }
  if (chunk >= a.rk_chunks) return;
  const int64_t first = a.rk_first[chunk];  // a chunk: up to 32 consecutive rows of the list (nrt_build_rank_stream splits where the lists grow long)
  const int rows = static_cast<int>(a.rk_first[chunk + 1] - first);
  const int64_t base = static_cast<int64_t>(window) * kWindow;
  const int32_t pn = a.perm[base + threadIdx.x];
  const bool in = pn >= 0;
  const int64_t n = in ? pn : 0;
  const int pos = in ? static_cast<int>(n - base) : 0;
  const int R = a.n_res;

  // ---- the chunk block -> LDS (coalesced 16-byte pieces), the stage zeroed
  const uint32_t c0 = a.rk_off[chunk], c1 = a.rk_off[chunk + 1];
  const int n_quads = static_cast<int>((c1 - c0) >> 2);
  uint32_t* const stage = lds + a.rk_max_dwords;  // [kPodsPerUnit / 4][kWindow]
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.rk_stream + c0);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (int i = threadIdx.x; i < n_quads; i += 256) dst[i] = src[i];
    uint4* z = reinterpret_cast<uint4*>(stage) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < kPodsPerUnit / 4 * kWindow / 4 / 256; ++i) z[i * 256] = uint4{0, 0, 0, 0};
  }
  const uint32_t flags = in ? a.flags[n] : 0u;
  const uint32_t node_present = in ? a.node_present[n] : 0u;
  uint32_t host_level = 0;
#pragma unroll
  for (int r = 0; r < RM; ++r)
    if (r < R && (a.slot_flags[r] & SPX_NRT_SLOT_HOST_LEVEL)) host_level |= 1u << r;
  uint32_t fill_bits = 0;  // host-level resources no zone of the node reports
  const uint32_t nn = static_cast<uint32_t>(a.n_nodes), n32 = static_cast<uint32_t>(n);
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    const uint32_t rep = (in && r < R) ? ld_off(a.f_rep, static_cast<uint32_t>(r) * nn + n32) : 0u;
    fill_bits |= (((host_level >> r) & 1u) && rep == 0) ? 1u << r : 0u;
  }
  __syncthreads();
  const bool narrow = lds[9] != 0u;  // block-uniform: every list of the chunk has at most 127 entries (nrt_build_rank_stream)

  // ---- the node's cells as counts: for each resource the eight zones' quantities, ranked against the chunk's list
  uint32_t q4[RM][4];  // WIDE: four dwords; NARROW: the first two
  const double* lists = reinterpret_cast<const double*>(lds + 16);
  uint32_t list_doubles = 0;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) q4[r][j] = narrow ? RkLayout<true>::G : RkLayout<false>::G;  // (slots past the table: never requested)
    if (r >= R) continue;  // uniform
    const uint32_t hw = lds[r];
    const int steps = static_cast<int>(hw & 0xffu);
    const uint32_t lo = hw >> 8;
    list_doubles = lo + (1u << steps);
    double av[kZ];
#pragma unroll
    for (int z = 0; z < kZ; ++z) av[z] = in ? ld_off(a.f_av, (static_cast<uint32_t>(z * R + r) * nn + n32) * 8u) : -1.0;
    uint32_t cnt[kZ];
#pragma unroll
    for (int z = 0; z < kZ; ++z) cnt[z] = 0;
    for (int b = 1 << (steps - 1); b > 0; b >>= 1) {  // uniform trip count; the eight searches advance together
#pragma unroll
      for (int z = 0; z < kZ; ++z) {
        const double v = lists[lo + cnt[z] + static_cast<uint32_t>(b) - 1u];
        cnt[z] = v <= av[z] ? cnt[z] + static_cast<uint32_t>(b) : cnt[z];
      }
    }
    if (narrow) {
      q4[r][0] = RkLayout<true>::G | cnt[0] | (cnt[1] << 8) | (cnt[2] << 16) | (cnt[3] << 24);
      q4[r][1] = RkLayout<true>::G | cnt[4] | (cnt[5] << 8) | (cnt[6] << 16) | (cnt[7] << 24);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) q4[r][j] = RkLayout<false>::G | cnt[j] | (cnt[j + 4] << 16);
    }
  }
  const uint32_t* const pods = lds + 16 + 2 * list_doubles;
  const bool fresh = flags & SPX_NRT_F_FRESH;
  const bool has_nrt = flags & SPX_NRT_F_HAS_NRT;
  const bool single = flags & SPX_NRT_F_SINGLE_NUMA;
  const bool pod_scope = flags & SPX_NRT_F_POD_SCOPE;
  const bool aligned = fresh && has_nrt && single;
  const bool w_pod = __ballot(aligned && pod_scope) != 0, w_ctr = __ballot(aligned && !pod_scope) != 0;
  const uint32_t st_stale = fresh ? 0u : static_cast<uint32_t>(SPX_NRT_ST_INVALID_TOPOLOGY);
  if (narrow) rank_walk<RM, true>(q4, fill_bits, host_level, pods, rows, lane, w_pod, w_ctr, aligned, pod_scope, node_present, st_stale, in, pos, stage);
  else rank_walk<RM, false>(q4, fill_bits, host_level, pods, rows, lane, w_pod, w_ctr, aligned, pod_scope, node_present, st_stale, in, pos, stage);
  __syncthreads();
  // rows leave as whole 256-byte segments (as k_nrt_fast): lane l gathers byte (row & 3) of the four dwords of nodes 4l .. 4l+3
  const int64_t col = base + lane * 4;
  if (col < a.row_stride) {
    for (int i = wave; i < rows; i += 4) {
      const int64_t row = uload(a.row_list + first + i);
      const uint32_t b = static_cast<uint32_t>(i & 3);
      const uint32_t pick = 0x0c0c0000u | ((4u + b) << 8) | b;
      const u32x4 w = *reinterpret_cast<const u32x4*>(&stage[(i >> 2) * kWindow + lane * 4]);
      const uint32_t lo = __builtin_amdgcn_perm(w.y, w.x, pick), hi = __builtin_amdgcn_perm(w.w, w.z, pick);
      *reinterpret_cast<uint32_t*>(a.out_status + row * a.row_stride + col) = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
    }
  }
}

}  // namespace

// the Filter launch of a whole-batch sweep over pod classes; false = not launched (no stream, or the block does not fit in LDS)
bool launch_nrt_filter_rank(const NrtArgs& a, int n_tiles, hipStream_t s) {
  if (!a.rk_stream || !a.rk_off || !a.rk_first || !a.row_list || a.rk_max_dwords == 0 || !a.out_status) return false;
  const int64_t per_round = n_tiles >= kXcdMapWindows ? ((n_tiles + 7) / 8) * 8 : n_tiles;  // the kernel's block map
  const unsigned blocks = static_cast<unsigned>(static_cast<int64_t>(a.rk_chunks) * per_round);
  const size_t lds = static_cast<size_t>(a.rk_max_dwords) * 4 + static_cast<size_t>(kPodsPerUnit / 4) * kWindow * 4;
  if (lds > kRkMaxChunkBytes + 8192) return false;
  if (a.n_res <= 4) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nrt_filter_rank<4>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL((k_nrt_filter_rank<4>), dim3(blocks), dim3(256), lds, s, a, n_tiles);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nrt_filter_rank<8>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL((k_nrt_filter_rank<8>), dim3(blocks), dim3(256), lds, s, a, n_tiles);
  }
  return true;
}

}  // namespace spx
