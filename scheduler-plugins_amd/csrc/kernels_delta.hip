// kernels_delta.hip — snapshot deltas (SURVEY.md 8d "upload deltas", 8f-2): rows of the node tables replaced in place.
//
// Between two scheduling cycles a few nodes change — an NRT object is republished (pluginhelpers.go:105-161), the OverReserve
// cache charges an assumed pod to a node's zones (cache/overreserve.go:170-203), the trimaran collector's cache moves
// (collector.go:139-150, handler.go:131-139) — and a full upload re-flattens and re-transposes every column of every node
// (config #5's node tables: 14 ms on the host for 20k nodes).  Here the changed rows travel as one staged blob and are scattered
// into the device columns; the float64 NRT formulation's derived columns (kernels_nrt_fast.hip) are recomputed on the device for
// those nodes, with the host upload's own expressions (IEEE division, no contraction: the results are the full upload's bit for bit).
#include "spx_internal.h"

namespace spx {

namespace {

// dst column-major [inner][n_nodes]  <-  src row-major [n_rows][inner] at nodes idx[row]
template <typename T>
__global__ __launch_bounds__(256) void k_scatter_rows(T* __restrict__ dst, int64_t n_nodes, int inner, const int32_t* __restrict__ idx,
                                                      const T* __restrict__ src, int64_t n_rows) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_rows * inner) return;
  const int64_t row = i / inner;
  const int k = static_cast<int>(i - row * inner);
  dst[static_cast<int64_t>(k) * n_nodes + idx[row]] = src[i];
}

// dst row-major [n][inner]  <-  src row-major [n_rows][inner] at rows idx[row] (the ElasticQuota tables: one 64-byte row per namespace)
template <typename T>
__global__ __launch_bounds__(256) void k_scatter_rows_rm(T* __restrict__ dst, int inner, const int32_t* __restrict__ idx, const T* __restrict__ src, int64_t n_rows) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_rows * inner) return;
  const int64_t row = i / inner;
  dst[static_cast<int64_t>(idx[row]) * inner + (i - row * inner)] = src[i];
}

// NetworkOverhead: pairs appended to the workload keys' lists — the host has laid the new CSR out and numbered the positions
__global__ __launch_bounds__(256) void k_net_append(int64_t n, const int32_t* __restrict__ pos, const int32_t* __restrict__ node, const int64_t* __restrict__ cost,
                                                    int32_t* __restrict__ dst_node, int64_t* __restrict__ dst_max) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dst_node[pos[i]] = node[i];
  dst_max[pos[i]] = cost[i];
}

__global__ __launch_bounds__(256) void k_nrt_derive_rows(NrtDeltaArgs a) {
  constexpr int Z = SPX_NRT_MAX_ZONES;
  const int R = a.n_res;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;  // (row, zone)
  if (i >= a.n_rows * Z) return;
  const int64_t row = i / Z;
  const int z = static_cast<int>(i - row * Z);
  const int64_t n = a.idx[row];
  const bool zone = z < a.n_zones[row];
  const uint32_t present = zone ? a.zone_present[row * Z + z] : 0u;
  double cpuv = 0.0, braw = kNrtNoCap;
  for (int r = 0; r < R; ++r) {
    // (same expressions as spx_upload_nrt_nodes: reported ? available : -1; RN(100 / Value(capacity)); RN(1 / Value(capacity)))
    double av = -1.0, rcp = kNrtNoCap, rcv = 1.0;
    if ((present >> r) & 1u) {
      const int64_t cap = a.zone_avail[(row * Z + z) * R + r];
      const bool is_cpu = r == a.cpu_slot;
      const double cap_v = static_cast<double>(is_cpu ? (cap + 999) / 1000 : cap);
      av = static_cast<double>(cap);
      rcp = cap_v > 0.0 ? 100.0 / cap_v : kNrtNoCap;
      rcv = cap_v > 0.0 ? 1.0 / cap_v : 1.0;
      if (is_cpu) cpuv = cap_v;
      if (is_cpu && cap > 0) braw = 100.0 / static_cast<double>(cap);
    }
    const int64_t at = (static_cast<int64_t>(z) * R + r) * a.n_nodes + n;
    a.f_av[at] = av, a.f_rc[at] = rcp, a.f_rcv[at] = rcv;
  }
  a.f_cpu[static_cast<int64_t>(z) * a.n_nodes + n] = cpuv;
  a.f_braw[static_cast<int64_t>(z) * a.n_nodes + n] = braw;
  if (z == 0)  // per resource: the zones reporting it
    for (int r = 0; r < R; ++r) {
      uint32_t rep = 0;
      for (int q = 0; q < Z && q < a.n_zones[row]; ++q) rep |= ((a.zone_present[row * Z + q] >> r) & 1u) << q;
      a.f_rep[static_cast<int64_t>(r) * a.n_nodes + n] = static_cast<uint8_t>(rep);
    }
}

}  // namespace

void launch_scatter_rows(void* dst, int64_t n_nodes, int inner, const int32_t* idx, const void* src, int64_t n_rows, int elem_bytes, hipStream_t s) {
  const int64_t n = n_rows * inner;
  if (n <= 0) return;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256)), block(256);
  if (elem_bytes == 1) hipLaunchKernelGGL(k_scatter_rows<uint8_t>, grid, block, 0, s, static_cast<uint8_t*>(dst), n_nodes, inner, idx, static_cast<const uint8_t*>(src), n_rows);
  else if (elem_bytes == 4) hipLaunchKernelGGL(k_scatter_rows<uint32_t>, grid, block, 0, s, static_cast<uint32_t*>(dst), n_nodes, inner, idx, static_cast<const uint32_t*>(src), n_rows);
  else hipLaunchKernelGGL(k_scatter_rows<uint64_t>, grid, block, 0, s, static_cast<uint64_t*>(dst), n_nodes, inner, idx, static_cast<const uint64_t*>(src), n_rows);
}

void launch_scatter_rows_rowmajor(void* dst, int inner, const int32_t* idx, const void* src, int64_t n_rows, int elem_bytes, hipStream_t s) {
  const int64_t n = n_rows * inner;
  if (n <= 0) return;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256)), block(256);
  if (elem_bytes == 1) hipLaunchKernelGGL(k_scatter_rows_rm<uint8_t>, grid, block, 0, s, static_cast<uint8_t*>(dst), inner, idx, static_cast<const uint8_t*>(src), n_rows);
  else hipLaunchKernelGGL(k_scatter_rows_rm<uint64_t>, grid, block, 0, s, static_cast<uint64_t*>(dst), inner, idx, static_cast<const uint64_t*>(src), n_rows);
}

void launch_net_append(int64_t n, const int32_t* pos, const int32_t* node, const int64_t* cost, int32_t* dst_node, int64_t* dst_max, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_net_append, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, n, pos, node, cost, dst_node, dst_max);
}

void launch_nrt_derive_rows(const NrtDeltaArgs& a, hipStream_t s) {
  const int64_t n = a.n_rows * SPX_NRT_MAX_ZONES;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_nrt_derive_rows, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace spx
