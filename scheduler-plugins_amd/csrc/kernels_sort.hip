// kernels_sort.hip — TopologicalSort as a batched device sort (topologicalsort.go:102-132).
//
// TopologicalSort.Less is not a strict weak order: pods of the same (non-empty) AppGroup compare by their index in the
// group's topology order (`orderP1 <= orderP2`, :131), everything else by upstream's PrioritySort (priority descending, then
// queue timestamp ascending, :109-113).  It is, however, a complete relation (for any two pods Less holds in at least one
// direction, counting PrioritySort ties), so an order in which every adjacent pair (x, y) satisfies Less(x, y) always
// exists.  This one is computed here, with two stable LSD radix sorts over pod indices:
//   1. sort all pods by K = (priority descending, queue timestamp ascending) — cross-group neighbours are then in order;
//   2. inside every maximal run of consecutive pods of one AppGroup, reorder by topology index (stable, so equal indexes
//      keep their K order).  A run's outside neighbours belong to other groups (or to none) and stay on the correct side of
//      every member of the run in K, so they remain in order whichever member ends up at the run's edge.
// Radix passes over bytes that are equal for all pods are skipped (a one-off histogram of every key byte decides that);
// keys stay where they are, only the 4-byte pod indices move.
//
// One pass = count (per 1024-key unit: LDS histogram of the digit) -> exclusive scan over [digit][unit] -> scatter (per unit,
// one wavefront walks its keys 64 at a time; lanes with equal digits find each other with 8 ballots, which gives a stable
// rank inside the step, and an LDS cursor per digit carries the position across steps).
#include <hip/hip_runtime.h>

#include "spx_internal.h"

namespace spx {
namespace {

constexpr int kWave = 64;
constexpr int kUnitKeys = 1024;   // keys per wavefront
constexpr int kWavesPerBlock = 4;

// digit `byte` of sort key `which` of pod p: 0..7 = queue timestamp (ascending, sign-biased), 8..11 = priority
// (descending, sign-biased), 12..15 = topology index (ascending, sign-biased), 16..19 = run id (ascending)
__device__ __forceinline__ unsigned key_digit(const SortArgs& a, int byte, int32_t p) {
  if (byte < 8) return static_cast<unsigned>((static_cast<uint64_t>(a.queue_ts[p]) ^ (1ull << 63)) >> (8 * byte)) & 0xffu;
  if (byte < 12) return (~(static_cast<uint32_t>(a.priority[p]) ^ 0x80000000u) >> (8 * (byte - 8))) & 0xffu;
  if (byte < 16) return ((static_cast<uint32_t>(a.topo_order[p]) ^ 0x80000000u) >> (8 * (byte - 12))) & 0xffu;
  return (static_cast<uint32_t>(a.run_of_pod[p]) >> (8 * (byte - 16))) & 0xffu;
}

// order-independent: how many pods carry each value of each of the 16 input key bytes -> hist[16][256]
__global__ __launch_bounds__(256) void k_sort_byte_hist(SortArgs a, unsigned* hist) {
  __shared__ unsigned h[16][256];
  for (int i = threadIdx.x; i < 16 * 256; i += blockDim.x) (&h[0][0])[i] = 0;
  __syncthreads();
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < a.n; p += static_cast<int64_t>(gridDim.x) * blockDim.x)
#pragma unroll
    for (int b = 0; b < 16; ++b) atomicAdd(&h[b][key_digit(a, b, static_cast<int32_t>(p))], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * 256; i += blockDim.x)
    if ((&h[0][0])[i]) atomicAdd(hist + i, (&h[0][0])[i]);
}

__global__ __launch_bounds__(256) void k_sort_iota(int32_t* idx, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = static_cast<int32_t>(i);
}

__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_radix_count(SortArgs a, const int32_t* in, int byte, unsigned* counts, int n_units) {
  __shared__ unsigned h[kWavesPerBlock][256];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * kWavesPerBlock + wave;
  for (int i = lane; i < 256; i += kWave) h[wave][i] = 0;
  __builtin_amdgcn_wave_barrier();
  if (unit < n_units) {
    const int64_t base = static_cast<int64_t>(unit) * kUnitKeys;
    for (int s = 0; s < kUnitKeys / kWave; ++s) {
      const int64_t i = base + s * kWave + lane;
      if (i < a.n) atomicAdd(&h[wave][key_digit(a, byte, in[i])], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < 256; d += kWave) counts[static_cast<int64_t>(d) * n_units + unit] = h[wave][d];
  }
}

// exclusive scan of counts[256 * n_units] (digit-major: all units of digit 0, then digit 1, ...), one block
__global__ __launch_bounds__(1024) void k_radix_scan(unsigned* counts, int64_t len) {
  __shared__ unsigned part[1024];
  const int t = threadIdx.x;
  const int64_t per = (len + 1023) / 1024;
  const int64_t b = t * per, e = b + per < len ? b + per : len;
  unsigned s = 0;
  for (int64_t i = b; i < e; ++i) s += counts[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned v = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned run = part[t] - s;  // exclusive prefix of this thread's segment
  for (int64_t i = b; i < e; ++i) {
    const unsigned c = counts[i];
    counts[i] = run;
    run += c;
  }
}

__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_radix_scatter(SortArgs a, const int32_t* in, int32_t* out, int byte, const unsigned* offsets,
                                                                           int n_units) {
  __shared__ unsigned cursor_s[kWavesPerBlock][256];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * kWavesPerBlock + wave;
  if (unit >= n_units) return;
  volatile unsigned* cursor = cursor_s[wave];
  for (int d = lane; d < 256; d += kWave) cursor[d] = offsets[static_cast<int64_t>(d) * n_units + unit];
  __builtin_amdgcn_wave_barrier();
  const int64_t base = static_cast<int64_t>(unit) * kUnitKeys;
  const uint64_t below = (1ull << lane) - 1ull;
  for (int s = 0; s < kUnitKeys / kWave; ++s) {
    const int64_t i = base + s * kWave + lane;
    const bool valid = i < a.n;
    const int32_t p = valid ? in[i] : 0;
    const unsigned d = valid ? key_digit(a, byte, p) : 0u;
    // lanes holding the same digit (among valid lanes): intersect, bit by bit, the ballots that agree with this lane
    uint64_t same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t m = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? m : ~m;
    }
    const unsigned rank = static_cast<unsigned>(__popcll(same & below));
    if (valid) {
      const unsigned pos = cursor[d] + rank;  // every lane reads before the group's first lane advances the cursor
      out[pos] = p;
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) cursor[d] += static_cast<unsigned>(__popcll(same));
    __builtin_amdgcn_wave_barrier();
  }
}

// heads of the maximal runs of consecutive same-AppGroup pods in the K-sorted order, counted per block of 1024 positions
__device__ __forceinline__ bool run_head(const SortArgs& a, const int32_t* perm, int64_t i) {
  if (i == 0) return true;
  const int32_t g = a.appgroup[perm[i]];
  return g < 0 || g != a.appgroup[perm[i - 1]];
}

__global__ __launch_bounds__(256) void k_run_count(SortArgs a, const int32_t* perm, unsigned* block_heads) {
  __shared__ unsigned cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  unsigned c = 0;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * 1024;
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i < a.n && run_head(a, perm, i)) ++c;
  }
  if (c) atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) block_heads[blockIdx.x] = cnt;
}

// run id of every position = (number of heads at or before it) - 1, stored per pod
__global__ __launch_bounds__(256) void k_run_assign(SortArgs a, const int32_t* perm, const unsigned* block_offsets, int32_t* run_of_pod) {
  __shared__ unsigned part[256];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * 1024 + threadIdx.x * 4;  // 4 consecutive positions per thread
  unsigned h[4], s = 0;
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k;
    h[k] = (i < a.n && run_head(a, perm, i)) ? 1u : 0u;
    s += h[k];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const unsigned v = threadIdx.x >= static_cast<unsigned>(off) ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = block_offsets[blockIdx.x] + part[threadIdx.x] - s;
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k;
    run += h[k];
    if (i < a.n) run_of_pod[perm[i]] = static_cast<int32_t>(run) - 1;
  }
}

}  // namespace

size_t sort_scratch_bytes(int64_t n) {
  const int64_t n_units = (n + kUnitKeys - 1) / kUnitKeys;
  const int64_t n_blocks = (n + 1023) / 1024;
  // [byte hist 16*256 u32 | counts 256*n_units u32 | block heads n_blocks u32 | idx A n i32 | idx B n i32 | run_of_pod n i32]
  return static_cast<size_t>(16 * 256 + 256 * n_units + n_blocks + 3 * n) * 4 + 256;
}

// Leaves the permutation in scratch (returned pointer, device memory).  `hist_host` is pinned host memory for the
// one-off byte histogram (16*256 u32); the call synchronises the stream once to read it.
const int32_t* launch_sort_keys(const SortArgs& a_in, void* scratch, unsigned* hist_host, hipStream_t s, hipError_t* err) {
  SortArgs a = a_in;
  const int64_t n = a.n;
  const int n_units = static_cast<int>((n + kUnitKeys - 1) / kUnitKeys);
  const int n_blocks = static_cast<int>((n + 1023) / 1024);
  unsigned* hist = static_cast<unsigned*>(scratch);
  unsigned* counts = hist + 16 * 256;
  unsigned* heads = counts + static_cast<int64_t>(256) * n_units;
  int32_t* idx_a = reinterpret_cast<int32_t*>(heads + n_blocks);
  int32_t* idx_b = idx_a + n;
  a.run_of_pod = idx_b + n;
  *err = hipMemsetAsync(hist, 0, 16 * 256 * sizeof(unsigned), s);
  if (*err != hipSuccess) return nullptr;
  hipLaunchKernelGGL(k_sort_byte_hist, dim3(static_cast<unsigned>(n_blocks < 512 ? n_blocks : 512)), dim3(256), 0, s, a, hist);
  hipLaunchKernelGGL(k_sort_iota, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, idx_a, n);
  if ((*err = hipMemcpyAsync(hist_host, hist, 16 * 256 * sizeof(unsigned), hipMemcpyDeviceToHost, s)) != hipSuccess) return nullptr;
  if ((*err = hipStreamSynchronize(s)) != hipSuccess) return nullptr;
  auto varies = [&](int byte) {
    for (int d = 0; d < 256; ++d)
      if (hist_host[byte * 256 + d] == static_cast<unsigned>(n)) return false;  // every pod holds digit d in this byte
    return true;
  };
  int32_t* in = idx_a;
  int32_t* out = idx_b;
  const unsigned cblocks = static_cast<unsigned>((n_units + kWavesPerBlock - 1) / kWavesPerBlock);
  auto pass = [&](int byte) {
    hipLaunchKernelGGL(k_radix_count, dim3(cblocks), dim3(kWave * kWavesPerBlock), 0, s, a, in, byte, counts, n_units);
    hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, s, counts, static_cast<int64_t>(256) * n_units);
    hipLaunchKernelGGL(k_radix_scatter, dim3(cblocks), dim3(kWave * kWavesPerBlock), 0, s, a, in, out, byte, counts, n_units);
    int32_t* t = in;
    in = out;
    out = t;
  };
  // 1. K = (priority descending, timestamp ascending): least significant digit first
  for (int byte = 0; byte < 12; ++byte)
    if (varies(byte)) pass(byte);
  // 2. runs of consecutive same-group pods, then (run, topology index) — stable, so K order survives inside equal keys
  hipLaunchKernelGGL(k_run_count, dim3(static_cast<unsigned>(n_blocks)), dim3(256), 0, s, a, in, heads);
  hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, s, heads, static_cast<int64_t>(n_blocks));
  hipLaunchKernelGGL(k_run_assign, dim3(static_cast<unsigned>(n_blocks)), dim3(256), 0, s, a, in, heads, a.run_of_pod);
  for (int byte = 12; byte < 16; ++byte)
    if (varies(byte)) pass(byte);
  for (int byte = 16; byte < 20; ++byte)
    if ((n - 1) >> (8 * (byte - 16))) pass(byte);  // run ids are below n
  *err = hipGetLastError();
  return in;
}

}  // namespace spx
