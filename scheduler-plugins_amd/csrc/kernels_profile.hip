// kernels_profile.hip — profile-level passes over the per-plugin tables: feasibility-aware NormalizeScore for
// NodeResourcesAllocatable, and the per-pod weighted argmax ("selectHost" input).
//
// Upstream runs Score/NormalizeScore only on the nodes that passed every Filter plugin and then sums
// plugin_weight x score ("upstream framework runtime").  When a Filter plugin (NRT,
// NetworkOverhead) or a caller mask is part of the evaluation, Allocatable's min/max must run over each pod's
// feasible set (allocatable.go:143-168 normalises the list it is handed), so its rows stop being identical.
// One wavefront per pod row; the row is read twice (status bytes, raw scores from L2) and written once.
#include "spx_internal.h"

namespace spx {

namespace {

constexpr int kMaxRowWaves = 16;  // waves a single-row launch puts on its row
constexpr int kNpl = 4;

__device__ __forceinline__ int64_t shfl_xor_i64(int64_t v, int m) {
  const int lo = __shfl_xor(static_cast<int>(v & 0xffffffffLL), m, 64);
  const int hi = __shfl_xor(static_cast<int>(v >> 32), m, 64);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

__device__ __forceinline__ bool feasible_at(const ProfileArgs& a, int64_t pod, int64_t n) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (a.status[k]) ok &= a.status[k][pod * a.row_stride + n] == 0;
  return ok;
}

// floor(num / den) for num <= 101 * den
__device__ __forceinline__ uint64_t div_le100(uint64_t num, uint64_t den) {
  const float qf = static_cast<float>(num) * __frcp_rn(static_cast<float>(den));
  uint64_t q = static_cast<uint64_t>(static_cast<uint32_t>(qf));
  const uint64_t prod = q * den;
  if (prod > num) --q;
  else if (num - prod >= den) ++q;
  return q;
}

// OR of the Filter status bytes of nodes n0..n0+3 (one dword per table): a non-zero byte marks an infeasible node.
// Rows are 128-byte aligned and n0 is a multiple of 4; bytes past n_nodes are never written by the sweeps and
// must be ignored by the caller.
__device__ __forceinline__ uint32_t infeasible4(const ProfileArgs& a, int64_t pod, int64_t n0) {
  uint32_t w = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (a.status[k]) w |= *reinterpret_cast<const uint32_t*>(a.status[k] + pod * a.row_stride + n0);
  return w;
}

// compact path: raw scores as uint32 offsets from the global minimum (k_alloc_prepare) — 4 B instead of 8 B per node
// read from L2 per row and pass, 32-bit min/max, and the quotient as one float64 multiply (range < 2^32 < 2^42).
// The first pass leaves each lane's 4 feasibility bits per tile in LDS (one byte), so the status tables — whose rows
// do not survive in L2 between the passes at full occupancy — are read from HBM once.
constexpr int kNplCompact = 16;  // nodes per lane and tile of the compact path: one dwordx4 per status table, four of offsets
__device__ __forceinline__ void alloc_masked_compact(const ProfileArgs& a, int64_t pod, int lane, int wave, int n_waves, uint8_t* feas_bytes) {
  uint16_t* feas = reinterpret_cast<uint16_t*>(feas_bytes);  // [tiles][64]: the lane's 16 feasibility bits
  const int64_t tiles = (a.row_stride + 64 * kNplCompact - 1) / (64 * kNplCompact);
  const int64_t row = pod * a.row_stride;
  uint32_t lo = 0xffffffffu, hi = 0u;
  bool any = false;
#pragma unroll 2
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t n0 = (t * 64 + lane) * kNplCompact;
    uint32_t ok = 0;
    if (n0 < a.n_nodes) {  // rows are padded to a multiple of 16 bytes
      uint4 bad = uint4{0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (a.status[k]) {
          const uint4 v = *reinterpret_cast<const uint4*>(a.status[k] + row + n0);
          bad.x |= v.x, bad.y |= v.y, bad.z |= v.z, bad.w |= v.w;
        }
      const uint32_t badw[4] = {bad.x, bad.y, bad.z, bad.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 r4 = *reinterpret_cast<const uint4*>(a.alloc_rel + n0 + 4 * q);
        const uint32_t r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (n0 + 4 * q + j >= a.n_nodes || ((badw[q] >> (8 * j)) & 0xffu)) continue;
          lo = r[j] < lo ? r[j] : lo;
          hi = r[j] > hi ? r[j] : hi;
          ok |= 1u << (4 * q + j);
        }
      }
    }
    any |= ok != 0;
    feas[t * 64 + lane] = static_cast<uint16_t>(ok);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const uint32_t olo = __shfl_xor(lo, m, 64), ohi = __shfl_xor(hi, m, 64);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  bool some = __ballot(any) != 0;
  if (n_waves > 1) {  // a whole workgroup on one row (single-row launches of the sequential commit loop): combine the waves
    __shared__ uint32_t s_lo[kMaxRowWaves], s_hi[kMaxRowWaves], s_any[kMaxRowWaves];
    if (lane == 0) s_lo[wave] = lo, s_hi[wave] = hi, s_any[wave] = some ? 1u : 0u;
    __syncthreads();
    lo = 0xffffffffu, hi = 0u, some = false;
    for (int w = 0; w < n_waves; ++w) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
      some |= s_any[w] != 0;
    }
  }
  const uint32_t range = some ? hi - lo : 0u;
  const double b = range ? (100.0 / static_cast<double>(range)) * (1.0 + 0x1p-49) : 0.0;
#pragma unroll 2
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t n0 = (t * 64 + lane) * kNplCompact;
    if (n0 >= a.row_stride) continue;
    uint32_t w[4] = {0, 0, 0, 0};
    const uint32_t ok = feas[t * 64 + lane];  // written by this lane
    if (ok != 0 && range != 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 r4 = *reinterpret_cast<const uint4*>(a.alloc_rel + n0 + 4 * q);
        const uint32_t r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((ok >> (4 * q + j)) & 1u) w[q] |= static_cast<uint32_t>(static_cast<double>(r[j] - lo) * b) << (8 * j);  // infeasible cells hold 0
      }
    }
    *reinterpret_cast<uint4*>(a.out_alloc + row + n0) = uint4{w[0], w[1], w[2], w[3]};
  }
}

// Row mapping of the per-row kernels of this file: a batch launch gives every wave of a workgroup its own row (kRowsPerBlock
// rows per workgroup: 62.5k single-wave workgroups were bound by the workgroup launch rate), a single-row launch (the sequential
// commit loop) puts the whole workgroup on the row.  `lds_per_row`: bytes of dynamic LDS a row's wave owns.
constexpr int kRowsPerBlock = 4;
struct RowMap {
  int64_t pod;
  int wave, n_waves;
  uint8_t* lds;
};
__device__ __forceinline__ RowMap row_map(const ProfileArgs& a, uint8_t* lds, size_t lds_per_row) {
  const bool single = a.row_end - a.row_begin == 1 || a.block_per_row;  // uniform
  const int w = threadIdx.x >> 6;
  RowMap m;
  m.pod = single ? a.row_begin + static_cast<int64_t>(blockIdx.x) : a.row_begin + static_cast<int64_t>(blockIdx.x) * kRowsPerBlock + w;
  m.wave = single ? w : 0;
  m.n_waves = single ? static_cast<int>(blockDim.x >> 6) : 1;
  m.lds = single ? lds : lds + static_cast<size_t>(w) * lds_per_row;
  return m;
}
inline unsigned row_blocks(unsigned rows, bool whole = false) { return rows == 1 || whole ? rows : (rows + kRowsPerBlock - 1) / kRowsPerBlock; }
inline unsigned row_threads(unsigned rows, bool whole = false) { return rows == 1 || whole ? 64u * kMaxRowWaves : 64u * kRowsPerBlock; }
// kRowsPerBlock shares of dynamic LDS (plus the kernels' static arrays) must fit the 64 KiB a launch gets without opting in to more:
// rows wider than that take the one-workgroup-per-row mapping (the layout of a single-row launch, blockIdx.x = row)
constexpr size_t kLdsLaunchLimit = 64 * 1024 - 1024;
inline bool whole_block_rows(unsigned rows, size_t lds_per_row) { return rows > 1 && lds_per_row * kRowsPerBlock > kLdsLaunchLimit; }

__global__ __launch_bounds__(64 * kMaxRowWaves) void k_alloc_masked(ProfileArgs a, unsigned lds_per_row) {
  SPX_RESOLVE_ROWS(a);
  extern __shared__ uint8_t feas_all[];  // per row: [tiles][64]
  const RowMap rm = row_map(a, feas_all, lds_per_row);
  const int lane = threadIdx.x & 63, wave = rm.wave, n_waves = rm.n_waves;
  const int64_t pod = rm.pod;
  if (pod >= a.row_end) return;  // wave-uniform; a batch launch has no workgroup barrier
  const int64_t tiles = (a.row_stride + 64 * kNpl - 1) / (64 * kNpl);
  uint8_t* feas = rm.lds;
  if (a.alloc_rel != nullptr && a.alloc_rel[a.row_stride] != 0u && a.row_stride % kNplCompact == 0) {  // wave-uniform
    alloc_masked_compact(a, pod, lane, wave, n_waves, feas);
    return;
  }
  int64_t lo = INT64_MAX, hi = -INT64_MAX;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t n0 = (t * 64 + lane) * kNpl;
    if (n0 >= a.n_nodes) continue;
    const uint32_t bad = infeasible4(a, pod, n0);
#pragma unroll
    for (int j = 0; j < kNpl; ++j) {
      const int64_t n = n0 + j;
      if (n >= a.n_nodes || ((bad >> (8 * j)) & 0xffu)) continue;
      const int64_t s = a.alloc_raw[n];
      lo = s < lo ? s : lo;
      hi = s > hi ? s : hi;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int64_t olo = shfl_xor_i64(lo, m), ohi = shfl_xor_i64(hi, m);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  if (n_waves > 1) {
    __shared__ int64_t s_lo[kMaxRowWaves], s_hi[kMaxRowWaves];
    if (lane == 0) s_lo[wave] = lo, s_hi[wave] = hi;
    __syncthreads();
    lo = INT64_MAX, hi = -INT64_MAX;
    for (int w = 0; w < n_waves; ++w) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
    }
  }
  const uint64_t range = static_cast<uint64_t>(hi) - static_cast<uint64_t>(lo);
  // (s - lo) * 100 / range with a wave-uniform range: below 2^42 the quotient is floor(d * b) for
  // b = RN(100 / range) * (1 + 2^-49) — the float64 product lies in [x, x + 2^-42) and frac(x) <= 1 - 1/range, so the
  // floor is exact (same argument as kernels_nrt_fast.hip); wider ranges keep the int64 division
  const bool small = hi >= lo && range < (1ull << 42);
  const double b = small && range ? (100.0 / static_cast<double>(range)) * (1.0 + 0x1p-49) : 0.0;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t n0 = (t * 64 + lane) * kNpl;
    if (n0 >= a.row_stride) continue;
    uint32_t w = 0;
    if (n0 < a.n_nodes && range != 0 && hi >= lo) {
      const uint32_t bad = infeasible4(a, pod, n0);
#pragma unroll
      for (int j = 0; j < kNpl; ++j) {
        const int64_t n = n0 + j;
        if (n >= a.n_nodes || ((bad >> (8 * j)) & 0xffu)) continue;  // infeasible cells hold 0
        const uint64_t d = static_cast<uint64_t>(a.alloc_raw[n]) - static_cast<uint64_t>(lo);
        uint32_t v;
        if (small) {
          const double df = __builtin_fma(static_cast<double>(static_cast<uint32_t>(d >> 32)), 0x1p32, static_cast<double>(static_cast<uint32_t>(d)));
          v = static_cast<uint32_t>(df * b);
        } else {
          const uint64_t q = div_le100(d * 100ull, range);
          v = q > 255 ? 255u : static_cast<uint32_t>(q);
        }
        w |= v << (8 * j);
      }
    }
    *reinterpret_cast<uint32_t*>(a.out_alloc + pod * a.row_stride + n0) = w;
  }
}

// The common case of the argmax below — non-negative weights whose weighted sum fits 31 bits — in 32-bit arithmetic, the
// tables in play compacted by the launcher (no per-plugin NULL tests), 16 nodes per lane and table (one dwordx4 each).  The
// general kernel multiplied int64 weights per cell and plugin and read one dword per lane: config #2's two tables took 1.12 ms
// (1.8 TB/s), more than the sweep that wrote them.
struct BestFastArgs {
  int32_t n_tab;
  const uint8_t* tab[SPX_NUM_PLUGINS];
  int32_t w[SPX_NUM_PLUGINS];
};
constexpr int kNplBest = 16;

__global__ __launch_bounds__(64 * kMaxRowWaves) void k_best_fast(ProfileArgs a, BestFastArgs f) {
  SPX_RESOLVE_ROWS(a);
  const RowMap rm = row_map(a, nullptr, 0);
  const int lane = threadIdx.x & 63, wave = rm.wave, n_waves = rm.n_waves;
  const int64_t pod = rm.pod;
  if (pod >= a.row_end) return;  // wave-uniform
  int best = -1, best_n = -1, ties = 0, feas = 0;
  const bool pod_ok = !a.prefilter || a.prefilter[pod] == 0;
  const int64_t tiles = pod_ok ? (a.n_nodes + 64 * kNplBest - 1) / (64 * kNplBest) : 0;
  const int64_t row = pod * a.row_stride;
#pragma unroll 2
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t n0 = (t * 64 + lane) * kNplBest;
    if (n0 >= a.n_nodes) continue;  // rows are padded to a multiple of 16 bytes: a lane's 16 bytes are inside the row
    uint4 bad = uint4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (a.status[k]) {
        const uint4 v = *reinterpret_cast<const uint4*>(a.status[k] + row + n0);
        bad.x |= v.x, bad.y |= v.y, bad.z |= v.z, bad.w |= v.w;
      }
    int tot[kNplBest];
#pragma unroll
    for (int j = 0; j < kNplBest; ++j) tot[j] = 0;
    for (int k = 0; k < f.n_tab; ++k) {
      const uint4 v = *reinterpret_cast<const uint4*>(f.tab[k] + row + n0);
      const uint32_t words[4] = {v.x, v.y, v.z, v.w};
      const int w = f.w[k];
#pragma unroll
      for (int j = 0; j < kNplBest; ++j) tot[j] += w * static_cast<int>((words[j >> 2] >> (8 * (j & 3))) & 0xffu);
    }
    const uint32_t badw[4] = {bad.x, bad.y, bad.z, bad.w};
#pragma unroll
    for (int j = 0; j < kNplBest; ++j) {
      const bool ok = n0 + j < a.n_nodes && ((badw[j >> 2] >> (8 * (j & 3))) & 0xffu) == 0;
      const int total = ok ? tot[j] : -1;
      feas += ok ? 1 : 0;
      // a lane walks its nodes in increasing order: `>` keeps the lowest index among equals
      ties = total > best ? 1 : ties + ((total == best && ok) ? 1 : 0);
      best_n = total > best ? static_cast<int>(n0) + j : best_n;
      best = total > best ? total : best;
    }
  }
  // (best, lowest node, ties, feasible) over the wave; best_n < 0 marks "none"
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int ob = __shfl_xor(best, m, 64), on = __shfl_xor(best_n, m, 64), ot = __shfl_xor(ties, m, 64);
    feas += __shfl_xor(feas, m, 64);
    if (on >= 0 && (best_n < 0 || ob > best || (ob == best && on < best_n))) {
      ties = (best_n >= 0 && ob == best) ? ties + ot : ot;
      best = ob;
      best_n = on;
    } else if (on >= 0 && ob == best) {
      ties += ot;
    }
  }
  if (n_waves > 1) {
    __shared__ int s_best[kMaxRowWaves], s_n[kMaxRowWaves], s_t[kMaxRowWaves], s_f[kMaxRowWaves];
    if (lane == 0) s_best[wave] = best, s_n[wave] = best_n, s_t[wave] = ties, s_f[wave] = feas;
    __syncthreads();
    if (threadIdx.x != 0) return;
    best = -1, best_n = -1, ties = 0, feas = 0;
    for (int w = 0; w < n_waves; ++w) {
      const int ob = s_best[w], on = s_n[w], ot = s_t[w];
      feas += s_f[w];
      if (on < 0) continue;
      if (best_n < 0 || ob > best) best = ob, best_n = on, ties = ot;
      else if (ob == best) ties += ot, best_n = on < best_n ? on : best_n;
    }
  }
  if (lane == 0) {
    a.best_node[pod] = best_n;
    a.best_score[pod] = best_n >= 0 ? best : 0;
    a.best_ties[pod] = best_n >= 0 ? ties : 0;
    a.best_feasible[pod] = feas;
  }
}

// spx_decide for a profile with Filter plugins (round 3): alloc_masked_compact's normalisation and k_best_fast's argmax in one
// pass pair over the row, so that Allocatable's table is neither written nor read back and the Filter status rows are read from
// HBM once instead of three times.  First pass: as alloc_masked_compact (min/max of the feasible nodes' Allocatable offsets, the
// lanes' feasibility bits parked in LDS).  Second pass: the Allocatable byte of each feasible node in registers, the other
// plugins' bytes from their tables (f: every scoring table of the mask except Allocatable's), the weighted total and the running
// (best, lowest node, ties) — k_best_fast's rule, a lane walking its nodes in increasing order.  Same bytes, same totals,
// same decision as spx_eval + spx_eval_best (tests/test_gpu_decide.py compares every row).
__global__ __launch_bounds__(64 * kMaxRowWaves) void k_decide_masked(ProfileArgs a, BestFastArgs f, int w_alloc, unsigned lds_per_row) {
  SPX_RESOLVE_ROWS(a);
  extern __shared__ uint8_t feas_bytes[];
  const RowMap rm = row_map(a, feas_bytes, lds_per_row);
  const int lane = threadIdx.x & 63, wave = rm.wave, n_waves = rm.n_waves;
  const int64_t pod = rm.pod;
  if (pod >= a.row_end) return;  // wave-uniform
  uint16_t* feas_bits = reinterpret_cast<uint16_t*>(rm.lds);  // [tiles][64]: the lane's 16 feasibility bits
  const bool pod_ok = !a.prefilter || a.prefilter[pod] == 0;    // wave-uniform
  const int64_t tiles = pod_ok ? (a.n_nodes + 64 * kNplCompact - 1) / (64 * kNplCompact) : 0;
  const int64_t row = pod * a.row_stride;
  uint32_t lo = 0xffffffffu, hi = 0u;
  bool any = false;
#pragma unroll 2
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t n0 = (t * 64 + lane) * kNplCompact;
    uint32_t ok = 0;
    if (n0 < a.n_nodes) {
      uint4 bad = uint4{0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (a.status[k]) {
          const uint4 v = *reinterpret_cast<const uint4*>(a.status[k] + row + n0);
          bad.x |= v.x, bad.y |= v.y, bad.z |= v.z, bad.w |= v.w;
        }
      const uint32_t badw[4] = {bad.x, bad.y, bad.z, bad.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 r4 = *reinterpret_cast<const uint4*>(a.alloc_rel + n0 + 4 * q);
        const uint32_t r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (n0 + 4 * q + j >= a.n_nodes || ((badw[q] >> (8 * j)) & 0xffu)) continue;
          lo = r[j] < lo ? r[j] : lo;
          hi = r[j] > hi ? r[j] : hi;
          ok |= 1u << (4 * q + j);
        }
      }
    }
    any |= ok != 0;
    feas_bits[t * 64 + lane] = static_cast<uint16_t>(ok);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const uint32_t olo = __shfl_xor(lo, m, 64), ohi = __shfl_xor(hi, m, 64);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  bool some = __ballot(any) != 0;
  __shared__ uint32_t s_lo[kMaxRowWaves], s_hi[kMaxRowWaves], s_any[kMaxRowWaves];
  __shared__ int s_best[kMaxRowWaves], s_n[kMaxRowWaves], s_t[kMaxRowWaves], s_f[kMaxRowWaves];
  if (n_waves > 1) {
    if (lane == 0) s_lo[wave] = lo, s_hi[wave] = hi, s_any[wave] = some ? 1u : 0u;
    __syncthreads();
    lo = 0xffffffffu, hi = 0u, some = false;
    for (int w = 0; w < n_waves; ++w) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
      some |= s_any[w] != 0;
    }
  }
  const uint32_t range = some ? hi - lo : 0u;
  const double b = range ? (100.0 / static_cast<double>(range)) * (1.0 + 0x1p-49) : 0.0;
  int best = -1, best_n = -1, ties = 0, feas = 0;
#pragma unroll 2
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t n0 = (t * 64 + lane) * kNplCompact;
    if (n0 >= a.n_nodes) continue;
    const uint32_t ok = feas_bits[t * 64 + lane];  // written by this lane
    if (ok == 0) continue;                          // none of the lane's 16 nodes is feasible: nothing to count
    int tot[kNplCompact];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 r4 = *reinterpret_cast<const uint4*>(a.alloc_rel + n0 + 4 * q);
      const uint32_t r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)  // the byte alloc_masked_compact writes (infeasible cells are never read below)
        tot[4 * q + j] = range != 0 ? w_alloc * static_cast<int>(static_cast<uint32_t>(static_cast<double>(r[j] - lo) * b)) : 0;
    }
    for (int k = 0; k < f.n_tab; ++k) {
      const uint4 v = *reinterpret_cast<const uint4*>(f.tab[k] + row + n0);
      const uint32_t words[4] = {v.x, v.y, v.z, v.w};
      const int w = f.w[k];
#pragma unroll
      for (int j = 0; j < kNplCompact; ++j) tot[j] += w * static_cast<int>((words[j >> 2] >> (8 * (j & 3))) & 0xffu);
    }
#pragma unroll
    for (int j = 0; j < kNplCompact; ++j) {
      const bool okj = (ok >> j) & 1u;
      const int total = okj ? tot[j] : -1;
      feas += okj ? 1 : 0;
      ties = total > best ? 1 : ties + ((total == best && okj) ? 1 : 0);
      best_n = total > best ? static_cast<int>(n0) + j : best_n;
      best = total > best ? total : best;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int ob = __shfl_xor(best, m, 64), on = __shfl_xor(best_n, m, 64), ot = __shfl_xor(ties, m, 64);
    feas += __shfl_xor(feas, m, 64);
    if (on >= 0 && (best_n < 0 || ob > best || (ob == best && on < best_n))) {
      ties = (best_n >= 0 && ob == best) ? ties + ot : ot;
      best = ob;
      best_n = on;
    } else if (on >= 0 && ob == best) {
      ties += ot;
    }
  }
  if (n_waves > 1) {
    if (lane == 0) s_best[wave] = best, s_n[wave] = best_n, s_t[wave] = ties, s_f[wave] = feas;
    __syncthreads();
    if (threadIdx.x != 0) return;
    best = -1, best_n = -1, ties = 0, feas = 0;
    for (int w = 0; w < n_waves; ++w) {
      const int ob = s_best[w], on = s_n[w], ot = s_t[w];
      feas += s_f[w];
      if (on < 0) continue;
      if (best_n < 0 || ob > best) best = ob, best_n = on, ties = ot;
      else if (ob == best) ties += ot, best_n = on < best_n ? on : best_n;
    }
  }
  if (lane == 0) {
    a.best_node[pod] = best_n;
    a.best_score[pod] = best_n >= 0 ? best : 0;
    a.best_ties[pod] = best_n >= 0 ? ties : 0;
    a.best_feasible[pod] = feas;
  }
}

// per pod: argmax over feasible nodes of Σ_plugin weight x score; ties resolved to the lowest node index, the
// tie count is returned so that callers can compare tie SETS (upstream selectHost picks randomly among them)
__global__ __launch_bounds__(64 * kMaxRowWaves) void k_best(ProfileArgs a) {
  SPX_RESOLVE_ROWS(a);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const int64_t pod = a.row_begin + blockIdx.x;
  if (pod >= a.row_end) return;
  int64_t best = INT64_MIN;
  int best_n = -1, ties = 0, feas = 0;
  const bool pod_ok = !a.prefilter || a.prefilter[pod] == 0;
  const int64_t tiles = pod_ok ? (a.n_nodes + 64 * kNpl - 1) / (64 * kNpl) : 0;
  for (int64_t t = wave; t < tiles; t += n_waves) {  // 4 consecutive nodes per lane: one dword per table
    const int64_t n0 = (t * 64 + lane) * kNpl;
    if (n0 >= a.n_nodes) continue;
    const uint32_t bad = infeasible4(a, pod, n0);
    uint32_t sc[SPX_NUM_PLUGINS];
#pragma unroll
    for (int k = 0; k < SPX_NUM_PLUGINS; ++k)
      sc[k] = a.score[k] ? *reinterpret_cast<const uint32_t*>(a.score[k] + pod * a.row_stride + n0) : 0u;
#pragma unroll
    for (int j = 0; j < kNpl; ++j) {
      const int64_t n = n0 + j;
      if (n >= a.n_nodes || ((bad >> (8 * j)) & 0xffu)) continue;
      ++feas;
      int64_t total = 0;
#pragma unroll
      for (int k = 0; k < SPX_NUM_PLUGINS; ++k)
        if (a.score[k]) total += a.weight[k] * static_cast<int64_t>((sc[k] >> (8 * j)) & 0xffu);
      if (total > best) {  // a lane walks its nodes in increasing order: `>` keeps the lowest index among equals
        best = total;
        best_n = static_cast<int>(n);
        ties = 1;
      } else if (total == best) {
        ++ties;
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int64_t ob = shfl_xor_i64(best, m);
    const int on = __shfl_xor(best_n, m, 64);
    const int ot = __shfl_xor(ties, m, 64);
    feas += __shfl_xor(feas, m, 64);
    if (ob > best || (ob == best && on >= 0 && (best_n < 0 || on < best_n))) {
      ties = ob > best ? ot : ties + ot;
      best = ob;
      best_n = on;
    } else if (ob == best && on >= 0) {
      ties += ot;
    }
  }
  if (n_waves > 1) {  // merge the waves' (best, lowest node, ties, feasible) with the rule of the butterfly above
    __shared__ int64_t s_best[kMaxRowWaves];
    __shared__ int s_n[kMaxRowWaves], s_t[kMaxRowWaves], s_f[kMaxRowWaves];
    if (lane == 0) s_best[wave] = best, s_n[wave] = best_n, s_t[wave] = ties, s_f[wave] = feas;
    __syncthreads();
    if (threadIdx.x != 0) return;
    best = INT64_MIN, best_n = -1, ties = 0, feas = 0;
    for (int w = 0; w < n_waves; ++w) {
      const int64_t ob = s_best[w];
      const int on = s_n[w], ot = s_t[w];
      feas += s_f[w];
      if (on < 0) continue;
      if (best_n < 0 || ob > best) best = ob, best_n = on, ties = ot;
      else if (ob == best) ties += ot, best_n = on < best_n ? on : best_n;
    }
  }
  if (lane == 0) {
    a.best_node[pod] = best_n;
    a.best_score[pod] = best_n >= 0 ? best : 0;
    a.best_ties[pod] = best_n >= 0 ? ties : 0;
    a.best_feasible[pod] = feas;
  }
}

// Pod equivalence classes: the sweep evaluated one representative row per class; every other member's row is a copy.  A workgroup takes a
// TASK — up to kExpandFan copies of one representative (the engine sorts the (row, representative) pairs by representative and cuts the runs:
// expand_tasks, spx_engine.h) — reads the representative's row once, 16 bytes per lane and step, and writes it to each copy.  (Round 5 had a
// workgroup per copied row: Peaks' 88 000 copies read their 12 000 representatives' 121 MB seven times over — 0.26 ms for 0.88 GB of copies.)
constexpr int kExpandFan = kRowsExpandFan;
__global__ __launch_bounds__(256) void k_rows_expand(const int32_t* __restrict__ pairs, const int32_t* __restrict__ tasks, uint8_t* t0, uint8_t* t1,
                                                     int64_t row_stride) {
  const int64_t first = tasks[2 * static_cast<int64_t>(blockIdx.x)];
  const int count = tasks[2 * static_cast<int64_t>(blockIdx.x) + 1];
  uint8_t* t = blockIdx.y ? t1 : t0;
  const int64_t src = pairs[2 * first + 1];
  const uint4* from = reinterpret_cast<const uint4*>(t + src * row_stride);
  uint4* to[kExpandFan];
#pragma unroll
  for (int j = 0; j < kExpandFan; ++j) to[j] = reinterpret_cast<uint4*>(t + static_cast<int64_t>(pairs[2 * (first + (j < count ? j : 0))]) * row_stride);
  for (int64_t i = threadIdx.x; i < row_stride / 16; i += 256) {
    const uint4 v = from[i];
#pragma unroll
    for (int j = 0; j < kExpandFan; ++j)
      if (j < count) to[j][i] = v;  // block-uniform
  }
}

}  // namespace

void launch_rows_expand(const int32_t* pairs, const int32_t* tasks, int64_t n_tasks, uint8_t* t0, uint8_t* t1, int64_t row_stride, hipStream_t s) {
  if (n_tasks <= 0 || (!t0 && !t1)) return;
  if (!t0) t0 = t1, t1 = nullptr;
  hipLaunchKernelGGL(k_rows_expand, dim3(static_cast<unsigned>(n_tasks), t1 ? 2 : 1), dim3(256), 0, s, pairs, tasks, t0, t1, row_stride);
}

void launch_alloc_masked(const ProfileArgs& a, hipStream_t s) {
  if (a.row_end <= a.row_begin) return;
  // dynamic LDS = the larger of the two layouts the kernel may pick: uint8[tiles of 4 nodes per lane][64] (general path) and
  // uint16[tiles of 16 nodes per lane][64] (alloc_masked_compact) — for rows of <= 256 bytes the second is the larger one
  const size_t tiles4 = static_cast<size_t>((a.row_stride + 64 * kNpl - 1) / (64 * kNpl));
  const size_t tiles16 = static_cast<size_t>((a.row_stride + 64 * kNplCompact - 1) / (64 * kNplCompact));
  const size_t lds = tiles4 * 64 > tiles16 * 128 ? tiles4 * 64 : tiles16 * 128;
  const unsigned rows = static_cast<unsigned>(a.row_end - a.row_begin);
  ProfileArgs b = a;
  const bool whole = rows > 1 && (a.block_per_row || whole_block_rows(rows, lds));
  b.block_per_row = whole;
  hipLaunchKernelGGL(k_alloc_masked, dim3(row_blocks(rows, whole)), dim3(row_threads(rows, whole)), rows == 1 || whole ? lds : lds * kRowsPerBlock, s, b,
                     static_cast<unsigned>(lds));
}

bool decide_masked_ok(const ProfileArgs& a) {
  int64_t bound = 0;
  bool ok = a.row_stride % kNplCompact == 0 && a.alloc_rel != nullptr;
  for (int k = 0; k < SPX_NUM_PLUGINS; ++k)
    if (a.score[k] || k == SPX_PLUGIN_ALLOCATABLE) {
      ok &= a.weight[k] >= 0 && a.weight[k] < (int64_t{1} << 23);
      bound += a.weight[k] * 255;
    }
  return ok && bound < (int64_t{1} << 31);
}

void launch_decide_masked(const ProfileArgs& a, hipStream_t s) {
  if (a.row_end <= a.row_begin) return;
  BestFastArgs f{};
  for (int k = 0; k < SPX_NUM_PLUGINS; ++k)
    if (a.score[k] && k != SPX_PLUGIN_ALLOCATABLE) {
      f.tab[f.n_tab] = a.score[k];
      f.w[f.n_tab++] = static_cast<int32_t>(a.weight[k]);
    }
  const size_t tiles16 = static_cast<size_t>((a.row_stride + 64 * kNplCompact - 1) / (64 * kNplCompact));
  const unsigned rows = static_cast<unsigned>(a.row_end - a.row_begin);
  ProfileArgs b = a;
  const bool whole = rows > 1 && (a.block_per_row || whole_block_rows(rows, tiles16 * 128));
  b.block_per_row = whole;
  hipLaunchKernelGGL(k_decide_masked, dim3(row_blocks(rows, whole)), dim3(row_threads(rows, whole)), tiles16 * 128 * (rows == 1 || whole ? 1 : kRowsPerBlock), s, b, f,
                     static_cast<int>(a.weight[SPX_PLUGIN_ALLOCATABLE]), static_cast<unsigned>(tiles16 * 128));
}

void launch_best(const ProfileArgs& a, hipStream_t s) {
  if (a.row_end <= a.row_begin) return;
  const unsigned rows = static_cast<unsigned>(a.row_end - a.row_begin);
  BestFastArgs f{};
  int64_t bound = 0;
  bool fast = a.row_stride % 16 == 0;
  for (int k = 0; k < SPX_NUM_PLUGINS; ++k)
    if (a.score[k]) {
      fast &= a.weight[k] >= 0 && a.weight[k] < (int64_t{1} << 23);
      bound += a.weight[k] * 255;
      f.tab[f.n_tab] = a.score[k];
      f.w[f.n_tab++] = static_cast<int32_t>(a.weight[k]);
    }
  if (fast && bound < (int64_t{1} << 31)) {
    hipLaunchKernelGGL(k_best_fast, dim3(row_blocks(rows)), dim3(row_threads(rows)), 0, s, a, f);
    return;
  }
  hipLaunchKernelGGL(k_best, dim3(rows), dim3(rows == 1 ? 64 * kMaxRowWaves : 64), 0, s, a);
}

}  // namespace spx
