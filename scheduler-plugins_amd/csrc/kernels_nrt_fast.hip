// kernels_nrt_fast.hip — float64 formulation of the NodeResourceTopologyMatch sweep (Filter + Score for the
// LeastAllocated / MostAllocated / BalancedAllocation strategies).
//
// Same decomposition as kernels_nrt.hip (lane = node, pod record wave-uniform) and the same results, bit for
// bit; what changes is the arithmetic.  The generic kernel works on int64 quantities exactly like the
// reference (two VALU instructions per add/compare, ~25 per truncating division, 64-bit id bitmasks).  When
// the engine has verified at upload time that
//   * every NUMA zone's id equals its list position (createNUMANodeList, pluginhelpers.go:105-131, with the
//     usual node-0..node-(Z-1) zones), so "lowest NUMA id" == "lowest list position",
//   * every zone quantity and every request lies in [0, 2^42), and 100 * sum(weights) < 2^42,
// all of those integers are exact in float64 and:
//   * compare / subtract / add-back are single v_cmp_f64 / v_fma_f64 instructions;
//   * floor(num / cap) for 0 <= num <= 101*cap becomes floor(num * rc) with rc = RN(1/cap) * (1 + 2^-49)
//     precomputed per (node, zone, resource): the product lies in [x, x + 2^-42) for the true quotient x, and
//     frac(x) <= 1 - 1/cap < 1 - 2^-42, so the floor is exact with no fix-up;
//   * Quantity.Value() of the cpu capacity (ceil(milli / 1000)) is precomputed per (node, zone).
// Snapshots that fail the check (and the LeastNUMANodes strategy) run the generic kernel.
//
// Reference: pkg/noderesourcetopology/filter.go:42-245, score.go:62-191, least_allocated.go:25-55,
// most_allocated.go:25-54, balanced_allocation.go:27-54, numaresources.go:105-182.
#include "spx_internal.h"

namespace spx {

namespace {

constexpr int kZ = SPX_NRT_MAX_ZONES;
constexpr int kC = SPX_NRT_MAX_CTRS;
constexpr int kPodsPerUnit = 32;
constexpr int kSgAlloc = 0;
constexpr int kSgBalanced = 1;

template <int RM>
struct FastNode {
  double av[kZ][RM];  // zone reports the resource ? available : -1
  double rc[kZ][RM];  // biased reciprocal of Value(capacity); 0 when the capacity is not positive
  double cpu_v[kZ];   // Value() of the cpu slot's capacity
  uint32_t rep[RM / 4];  // per resource: 8-bit mask of the zones that report it
  uint32_t node_present;
  int nz;
  __device__ __forceinline__ uint32_t repmask(int r) const { return (rep[r >> 2] >> (8 * (r & 3))) & 0xffu; }
};

struct Q2 {
  double raw;    // the request as written (cpu in millicores)
  double value;  // Quantity.Value(): cpu rounded up to whole cores
};

// Wave-uniform read of immutable input through the constant address space: the backend may then use scalar
// loads (s_load_dwordxN into SGPRs).  Through a plain global pointer it cannot — the kernel's own table stores
// might alias — and every pod-record access becomes a vector load with a uniform address (measured: 54 VMEM
// reads per wave per pod, 56 % of wave cycles waiting).
template <typename T>
__device__ __forceinline__ T uload(const T* p) {
  typedef const T __attribute__((address_space(4))) CT;
  return *reinterpret_cast<CT*>(reinterpret_cast<uintptr_t>(p));
}

// per-pod header, 8 dwords (one s_load_dwordx8):
//   w0 = qos | non_native << 8 | n_ctr << 16 | pod_present << 24;  w1,w2 = ctr_kind[0..7];  w3,w4 = ctr_present[0..7]
typedef uint32_t PodHdr __attribute__((ext_vector_type(8)));
typedef double F64x2 __attribute__((ext_vector_type(2)));

// resourcesAvailableInAnyNUMANodes filter.go:93-163 with ids == positions
template <int RM>
__device__ __forceinline__ bool fits_fast(const FastNode<RM>& ns, const NrtArgs& a, bool non_g, uint32_t present,
                                          const Q2* __restrict__ q2, uint32_t* pos) {
  uint32_t mask = 0xffu;
  bool ok = true;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (r >= a.n_res || !((present >> r) & 1u)) continue;  // uniform
    const double q = uload(&q2[r].raw);
    if (__double_as_longlong(q) == 0) continue;  // "ignoring zero-qty resource request"
    const bool always = non_g && (a.slot_flags[r] & SPX_NRT_SLOT_AFFINE);
    const bool host_level = a.slot_flags[r] & SPX_NRT_SLOT_HOST_LEVEL;
    ok &= ((ns.node_present >> r) & 1u) != 0;
    const uint32_t rep = ns.repmask(r);
    uint32_t rb;
    if (always) {
      rb = rep;
    } else {
      rb = 0;
#pragma unroll
      for (int z = 0; z < kZ; ++z) rb |= ns.av[z][r] >= q ? (1u << z) : 0u;
    }
    mask &= (host_level && rep == 0) ? 0xffu : rb;
  }
  *pos = mask ? static_cast<uint32_t>(__builtin_ctz(mask)) : 0u;
  return ok && mask != 0;
}

// subtractResourcesFromNUMANodeList numaresources.go:145-182 (sign -1) / its inverse (+1).  Unreported cells
// hold a negative value and stay negative, which is all any reader tests.
template <int RM>
__device__ __forceinline__ void adjust_fast(FastNode<RM>& ns, const NrtArgs& a, bool non_g, uint32_t present,
                                            const Q2* __restrict__ q2, uint32_t pos, bool apply, double sign) {
  double sel[kZ];
#pragma unroll
  for (int z = 0; z < kZ; ++z) sel[z] = (apply && pos == static_cast<uint32_t>(z)) ? sign : 0.0;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (r >= a.n_res || !((present >> r) & 1u)) continue;
    if (non_g && (a.slot_flags[r] & SPX_NRT_SLOT_AFFINE)) continue;
    const double q = uload(&q2[r].raw);
    if (__double_as_longlong(q) == 0) continue;
#pragma unroll
    for (int z = 0; z < kZ; ++z) ns.av[z][r] = __builtin_fma(sel[z], q, ns.av[z][r]);
  }
}

// scoreForEachNUMANode score.go:110-124 over the zone strategy scores
template <int RM, int SG>
__device__ __forceinline__ int score_each_fast(const FastNode<RM>& ns, const NrtArgs& a, uint32_t present,
                                               const Q2* __restrict__ q2) {
  const uint32_t used = present & ((1u << a.n_res) - 1u);
  int min_score = 0;
  Q2 q[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    const F64x2 v = ((used >> r) & 1u) ? uload(reinterpret_cast<const F64x2*>(q2 + r)) : F64x2{0.0, 0.0};
    q[r] = Q2{v.x, v.y};
  }
  if constexpr (SG == kSgBalanced) {
    const double n = static_cast<double>(__builtin_popcount(used));
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      double fr[RM];
      bool over = false;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        fr[r] = 0.0;
        if (!((used >> r) & 1u)) continue;
        const double cap = ns.av[z][r];
        const double cap_v = r == a.cpu_slot ? ns.cpu_v[z] : cap;
        const double f = cap > 0.0 ? q[r].value / cap_v : 1.0;  // fractionOfCapacity balanced_allocation.go:49-54
        over |= f > 1.0;
        fr[r] = f;
      }
      // gonum stat.Variance (corrected two-pass, unbiased), fractions in ascending resource id
      double sum = 0.0;
#pragma unroll
      for (int r = 0; r < RM; ++r) sum += fr[r];
      const double mean = sum / n;
      double ss = 0.0, comp = 0.0;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const double d = ((used >> r) & 1u) ? fr[r] - mean : 0.0;
        ss += d * d;
        comp += d;
      }
      const double variance = (ss - comp * comp / n) / (n - 1.0);
      const int s = over ? 0 : static_cast<int>((1.0 - variance) * 100.0);
      if (z < ns.nz && (min_score == 0 || (s != 0 && s < min_score))) min_score = s;
    }
  } else {
    const bool least = a.strategy == SPX_NRT_LEAST_ALLOCATED;
    const double wsum = uload(a.wtab + 2 * used);
    const double wrc = uload(a.wtab + 2 * used + 1);
    if (wsum == 0.0) return 0;
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        if (!((used >> r) & 1u)) continue;
        const double cap = ns.av[z][r];
        const double cap_v = r == a.cpu_slot ? ns.cpu_v[z] : cap;
        const bool ok = q[r].raw <= cap && cap > 0.0;  // capacity == 0 or request > capacity -> 0
        const double num = least ? (cap_v - q[r].value) * 100.0 : q[r].value * 100.0;
        const double rs = __builtin_floor(num * ns.rc[z][r]);
        acc = __builtin_fma(ok ? rs : 0.0, a.slot_weight_f[r], acc);
      }
      const int s = static_cast<int>(acc * wrc);  // floor(acc / wsum), 0 <= quotient <= 100
      if (z < ns.nz && (min_score == 0 || (s != 0 && s < min_score))) min_score = s;
    }
  }
  return min_score;
}

__constant__ uint32_t kInv16[kC + 1] = {0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192};  // ceil(2^16 / n)

template <int RM, int SG>
__global__ __launch_bounds__(256, RM == 4 ? 2 : 1) void k_nrt_fast(NrtArgs a, int n_tiles) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  const int64_t pod0 = a.row_begin + chunk * kPodsPerUnit;
  if (pod0 >= a.row_end) return;
  const int64_t pod1 = pod0 + kPodsPerUnit < a.row_end ? pod0 + kPodsPerUnit : a.row_end;
  const int64_t n = static_cast<int64_t>(tile) * 64 + lane;
  const bool in = n < a.n_nodes;
  const int R = a.n_res;

  FastNode<RM> ns;
  const uint32_t flags = in ? a.flags[n] : 0u;
  ns.nz = in ? a.n_zones[n] : 0;
  ns.node_present = in ? a.node_present[n] : 0u;
#pragma unroll
  for (int i = 0; i < RM / 4; ++i) ns.rep[i] = 0;
#pragma unroll
  for (int r = 0; r < RM; ++r)
    if (r < R && in) ns.rep[r >> 2] |= static_cast<uint32_t>(a.f_rep[static_cast<int64_t>(r) * a.n_nodes + n]) << (8 * (r & 3));
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    ns.cpu_v[z] = (in && a.cpu_slot >= 0) ? a.f_cpu[static_cast<int64_t>(z) * a.n_nodes + n] : 0.0;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int64_t i = (static_cast<int64_t>(z) * R + r) * a.n_nodes + n;
      ns.av[z][r] = (in && r < R) ? a.f_av[i] : -1.0;
      ns.rc[z][r] = (in && r < R) ? a.f_rc[i] : 0.0;
    }
  }
  const bool fresh = flags & SPX_NRT_F_FRESH;
  const bool has_nrt = flags & SPX_NRT_F_HAS_NRT;
  const bool single = flags & SPX_NRT_F_SINGLE_NUMA;
  const bool pod_scope = flags & SPX_NRT_F_POD_SCOPE;

  for (int64_t pod = pod0; pod < pod1; ++pod) {
    // ---- wave-uniform pod record
    const PodHdr h = uload(reinterpret_cast<const PodHdr*>(a.pod_hdr) + pod);
    const int qos = h[0] & 0xffu;
    const bool non_native = ((h[0] >> 8) & 0xffu) != 0;
    const int n_ctr = (h[0] >> 16) & 0xffu;
    const uint32_t pod_present = h[0] >> 24;
    const uint64_t kinds = h[1] | (static_cast<uint64_t>(h[2]) << 32);
    const uint64_t press = h[3] | (static_cast<uint64_t>(h[4]) << 32);
    const Q2* __restrict__ preq = reinterpret_cast<const Q2*>(a.pod_q2) + pod * R;
    const Q2* __restrict__ creq = reinterpret_cast<const Q2*>(a.ctr_q2) + pod * kC * R;
    auto ckind_of = [&](int c) { return static_cast<uint32_t>(kinds >> (8 * c)) & 0xffu; };
    auto cpres_of = [&](int c) { return static_cast<uint32_t>(press >> (8 * c)) & 0xffu; };
    const bool non_g = qos != SPX_QOS_GUARANTEED;

    // ================= Filter (filter.go:179-245)
    uint32_t status = 0;
    if (!(qos == SPX_QOS_BESTEFFORT && !non_native)) {  // uniform
      if (!fresh) {
        status = SPX_NRT_ST_INVALID_TOPOLOGY;
      } else if (has_nrt && single) {
        if (pod_scope) {  // singleNUMAPodLevelHandler
          uint32_t pos;
          if (!fits_fast(ns, a, non_g, pod_present, preq, &pos)) status = SPX_NRT_ST_POD;
        } else {  // singleNUMAContainerLevelHandler
          int last_app = -1;
          for (int c = 0; c < n_ctr; ++c) {  // init containers: must fit, never subtracted
            if (ckind_of(c) == SPX_CTR_APP) {
              last_app = c;
              continue;
            }
            uint32_t pos;
            const bool ok = fits_fast(ns, a, non_g, cpres_of(c), creq + c * R, &pos);
            if (status == 0 && !ok) status = ckind_of(c) == SPX_CTR_SIDECAR ? SPX_NRT_ST_SIDECAR_CONTAINER : SPX_NRT_ST_INIT_CONTAINER;
          }
          uint32_t chosen = 0;  // list position picked per app container (for the undo), 4 bits each
          uint32_t placed = 0;  // bit c: container c was subtracted on this lane
          for (int c = 0; c < n_ctr; ++c) {
            if (ckind_of(c) != SPX_CTR_APP) continue;
            uint32_t pos;
            const bool ok = fits_fast(ns, a, non_g, cpres_of(c), creq + c * R, &pos);
            const bool live = status == 0;
            if (live && !ok) status = SPX_NRT_ST_CONTAINER;
            if (c == last_app) break;  // nothing reads the table after the last app container
            const bool apply = live && ok;
            adjust_fast(ns, a, non_g, cpres_of(c), creq + c * R, pos, apply, -1.0);
            chosen |= (apply ? pos : 0u) << (4 * c);
            placed |= (apply ? 1u : 0u) << c;
          }
          for (int c = 0; c < last_app; ++c) {  // undo: Filter works on a private copy in the reference
            if (ckind_of(c) != SPX_CTR_APP) continue;
            adjust_fast(ns, a, non_g, cpres_of(c), creq + c * R, (chosen >> (4 * c)) & 0xfu, (placed >> c) & 1u, 1.0);
          }
        }
      }
    }

    // ================= Score (score.go:62-102)
    int score;
    if (non_g) {
      score = 100;
    } else if (!fresh || !has_nrt || !single) {
      score = 0;
    } else if (pod_scope) {
      score = score_each_fast<RM, SG>(ns, a, pod_present, preq);
    } else {  // containerScopeScore: int64(mean) over init + app containers
      int sum = 0;
      for (int c = 0; c < n_ctr; ++c) sum += score_each_fast<RM, SG>(ns, a, cpres_of(c), creq + c * R);
      score = static_cast<int>((static_cast<uint32_t>(sum) * kInv16[n_ctr]) >> 16);  // sum / n_ctr for sum <= 800
    }

    if (in && a.out_raw != nullptr) {
      a.out_raw[n] = score;
    } else if (in) {
      const int64_t cell = pod * a.row_stride + n;
      a.out_status[cell] = static_cast<uint8_t>(status);
      a.out_score[cell] = static_cast<uint8_t>(score > 255 ? 255 : score);
    }
  }
}

}  // namespace

bool launch_nrt_fast(const NrtArgs& a, hipStream_t s) {
  if (!a.fast || a.strategy == SPX_NRT_LEAST_NUMA_NODES) return false;
  const int n_tiles = static_cast<int>((a.n_nodes + 63) / 64);
  const int64_t chunks = (a.row_end - a.row_begin + kPodsPerUnit - 1) / kPodsPerUnit;
  const unsigned blocks = static_cast<unsigned>((chunks * n_tiles + 3) / 4);
  const int sg = a.strategy == SPX_NRT_BALANCED_ALLOCATION ? kSgBalanced : kSgAlloc;
#define SPX_NRTF_CASE(RMV, SGV)                                                              \
  if ((a.n_res <= 4) == (RMV == 4) && sg == SGV) {                                           \
    hipLaunchKernelGGL((k_nrt_fast<RMV, SGV>), dim3(blocks), dim3(256), 0, s, a, n_tiles);  \
    return true;                                                                             \
  }
  SPX_NRTF_CASE(4, kSgAlloc)
  SPX_NRTF_CASE(4, kSgBalanced)
  SPX_NRTF_CASE(8, kSgAlloc)
  SPX_NRTF_CASE(8, kSgBalanced)
#undef SPX_NRTF_CASE
  return false;
}

}  // namespace spx
