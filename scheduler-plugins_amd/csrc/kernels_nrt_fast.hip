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
//   * the truncating divisions by a capacity c < 2^42 (quotient x <= 100, numerator an integer) become one
//     multiplication by b = RN(100 / c), precomputed per (node, zone, resource), biased so that the float64
//     value t lands in [x, x + 2^-42): since frac(x) <= 1 - 1/c < 1 - 2^-42 for a non-integer x, floor(t) ==
//     floor(x) with no fix-up.  LeastAllocated: (c - v) * 100 / c = 100 - v * (100 / c), t = fma(-v, b, 100 + 2^-43),
//     total rounding error < 2^-45.6; MostAllocated: t = (v * (1 + 2^-49)) * b, relative error within
//     2^-49 +- 3 * 2^-53; the final acc / sum(weights) uses the same biased reciprocal;
//   * Quantity.Value() of a cpu capacity (ceil(milli / 1000)) is folded into b.
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
constexpr int kSgLeast = 0;
constexpr int kSgMost = 1;
constexpr int kSgBalanced = 2;
constexpr double kNoCap = kNrtNoCap;  // b[][] of a cell whose capacity is not positive

template <int RM>
struct FastNode {
  double av[kZ][RM];     // zone reports the resource ? available : -1
  double b[kZ][RM];      // RN(100 / Value(capacity)); kNoCap when the capacity is not positive
  uint32_t rep[RM / 4];  // per resource: 8-bit mask of the zones that report it
  uint32_t fill[RM / 4]; // per resource: 0xff when no zone reports a host-level resource (the check is skipped), else 0
  uint32_t node_present;
  int nz;
  __device__ __forceinline__ uint32_t repmask(int r) const { return (rep[r >> 2] >> (8 * (r & 3))) & 0xffu; }
  __device__ __forceinline__ uint32_t fillmask(int r) const { return (fill[r >> 2] >> (8 * (r & 3))) & 0xffu; }
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

// per-pod header, 16 dwords (one s_load_dwordx16), built by the engine at upload (spx_engine.hip: nrt_pod_header)
typedef uint32_t PodHdr __attribute__((ext_vector_type(16)));
typedef double F64x2 __attribute__((ext_vector_type(2)));

// the request subsets of one container (or of the pod), precomputed per pod on the host from the QoS class,
// the slot flags and which quantities are zero
struct Sets {
  uint32_t used;    // requested resources (Score iterates these)
  uint32_t fit;     // non-zero requests compared per zone:            available >= quantity
  uint32_t always;  // non-zero requests of a non-Guaranteed pod for a NUMA-affine resource: any reporting zone suits
  uint32_t zero;    // explicit zero-quantity requests
};

// resourcesAvailableInAnyNUMANodes filter.go:93-163 with ids == positions
template <int RM>
__device__ __forceinline__ bool fits_fast(const FastNode<RM>& ns, const Sets& st, const Q2* __restrict__ q2, uint32_t* pos) {
  const uint32_t need = st.fit | st.always;
  const bool ok = (need & ~ns.node_present) == 0;  // requested but not reported at node level -> cannot meet request
  uint32_t mask = 0xffu;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((need >> r) & 1u)) continue;  // uniform
    uint32_t rb;
    if ((st.always >> r) & 1u) {
      rb = ns.repmask(r);
    } else {
      const double q = uload(&q2[r].raw);
      rb = 0;
#pragma unroll
      for (int z = 0; z < kZ; ++z) rb |= ns.av[z][r] >= q ? (1u << z) : 0u;
    }
    mask &= rb | ns.fillmask(r);
  }
  *pos = mask ? static_cast<uint32_t>(__builtin_ctz(mask)) : 0u;
  return ok && mask != 0;
}

// subtractResourcesFromNUMANodeList numaresources.go:145-182 (sign -1) / its inverse (+1).  Unreported cells
// hold a negative value and stay negative, which is all any reader tests.
template <int RM>
__device__ __forceinline__ void adjust_fast(FastNode<RM>& ns, const Sets& st, const Q2* __restrict__ q2, uint32_t pos,
                                            bool apply, double sign) {
  double sel[kZ];
#pragma unroll
  for (int z = 0; z < kZ; ++z) sel[z] = (apply && pos == static_cast<uint32_t>(z)) ? sign : 0.0;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((st.fit >> r) & 1u)) continue;
    const double q = uload(&q2[r].raw);
#pragma unroll
    for (int z = 0; z < kZ; ++z) ns.av[z][r] = __builtin_fma(sel[z], q, ns.av[z][r]);
  }
}

// scoreForEachNUMANode score.go:110-124: the minimum of the non-zero zone scores, 0 when there is none (the
// reference's running rule `min == 0 || (s != 0 && s < min)` is order-independent).  Zones past the node's
// count hold no capacity and score 0 under Least/MostAllocated, so they drop out by themselves.
template <int RM, int SG>
__device__ __forceinline__ int score_each_fast(const FastNode<RM>& ns, const NrtArgs& a, const Sets& st,
                                               const Q2* __restrict__ q2, const double* __restrict__ cpu_v) {
  const uint32_t used = st.used;
  uint32_t m = 0xffffffffu;  // min over zones of (score - 1) as unsigned: a zero score wraps to the maximum
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
        const double cap_v = r == a.cpu_slot ? cpu_v[z] : cap;
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
      const int s = (over || z >= ns.nz) ? 0 : static_cast<int>((1.0 - variance) * 100.0);
      const uint32_t s1 = static_cast<uint32_t>(s) - 1u;
      m = s1 < m ? s1 : m;
    }
  } else {
    const double wsum = uload(a.wtab + 2 * used);
    const double wrc = uload(a.wtab + 2 * used + 1);
    if (__double_as_longlong(wsum) == 0) return 0;
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        if (!((used >> r) & 1u)) continue;
        double rs;
        if constexpr (SG == kSgLeast) {
          // (cap_v - req_v) * 100 / cap_v == 100 - req_v * (100 / cap_v); see the header for why the floor is exact
          // cells without capacity hold b = +inf: -v * inf is -inf (v > 0) or NaN (explicit zero request), and
          // max(., 0) turns both into the reference's 0
          rs = __builtin_fmax(__builtin_floor(__builtin_fma(-q[r].value, ns.b[z][r], 100.0 + 0x1p-43)), 0.0);
        } else {
          const double t = __builtin_floor((q[r].value * (1.0 + 0x1p-49)) * ns.b[z][r]);
          rs = q[r].raw <= ns.av[z][r] ? t : 0.0;
        }
        acc = __builtin_fma(rs, a.slot_weight_f[r], acc);
      }
      const uint32_t s1 = static_cast<uint32_t>(static_cast<int>(acc * wrc)) - 1u;  // floor(acc / wsum) in 0..100
      m = s1 < m ? s1 : m;
    }
  }
  return static_cast<int>(m + 1u);
}

__constant__ uint32_t kInv16[kC + 1] = {0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192};  // ceil(2^16 / n)

template <int RM, int SG>
__global__ __launch_bounds__(256, RM == 4 ? 3 : 1) void k_nrt_fast(NrtArgs a, int n_tiles) {
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
  double cpu_v[kZ];
  const uint32_t flags = in ? a.flags[n] : 0u;
  ns.nz = in ? a.n_zones[n] : 0;
  ns.node_present = in ? a.node_present[n] : 0u;
#pragma unroll
  for (int i = 0; i < RM / 4; ++i) ns.rep[i] = ns.fill[i] = 0;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (r >= R) continue;
    const uint32_t rep = in ? a.f_rep[static_cast<int64_t>(r) * a.n_nodes + n] : 0u;
    ns.rep[r >> 2] |= rep << (8 * (r & 3));
    if ((a.slot_flags[r] & SPX_NRT_SLOT_HOST_LEVEL) && rep == 0) ns.fill[r >> 2] |= 0xffu << (8 * (r & 3));
  }
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    cpu_v[z] = (SG == kSgBalanced && in && a.cpu_slot >= 0) ? a.f_cpu[static_cast<int64_t>(z) * a.n_nodes + n] : 0.0;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int64_t i = (static_cast<int64_t>(z) * R + r) * a.n_nodes + n;
      ns.av[z][r] = (in && r < R) ? a.f_av[i] : -1.0;
      const double b = (SG != kSgBalanced && in && r < R) ? a.f_rc[i] : kNoCap;
      ns.b[z][r] = (SG == kSgLeast && b == kNoCap) ? __builtin_inf() : b;
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
    const int last_app = static_cast<int>(h[0] >> 24) == 0xff ? -1 : static_cast<int>(h[0] >> 24);
    const uint64_t kinds = h[1] | (static_cast<uint64_t>(h[2]) << 32);
    const uint64_t w_used = h[3] | (static_cast<uint64_t>(h[4]) << 32);
    const uint64_t w_fit = h[5] | (static_cast<uint64_t>(h[6]) << 32);
    const uint64_t w_always = h[7] | (static_cast<uint64_t>(h[8]) << 32);
    const uint64_t w_zero = h[9] | (static_cast<uint64_t>(h[10]) << 32);
    const Sets pod_sets{h[11] & 0xffu, (h[11] >> 8) & 0xffu, (h[11] >> 16) & 0xffu, h[11] >> 24};
    const Q2* __restrict__ preq = reinterpret_cast<const Q2*>(a.pod_q2) + pod * R;
    const Q2* __restrict__ creq = reinterpret_cast<const Q2*>(a.ctr_q2) + pod * kC * R;
    auto ckind_of = [&](int c) { return static_cast<uint32_t>(kinds >> (8 * c)) & 0xffu; };
    auto sets_of = [&](int c) {
      return Sets{static_cast<uint32_t>(w_used >> (8 * c)) & 0xffu, static_cast<uint32_t>(w_fit >> (8 * c)) & 0xffu,
                  static_cast<uint32_t>(w_always >> (8 * c)) & 0xffu, static_cast<uint32_t>(w_zero >> (8 * c)) & 0xffu};
    };
    const bool non_g = qos != SPX_QOS_GUARANTEED;

    // ================= Filter (filter.go:179-245)
    uint32_t status = 0;
    if (!(qos == SPX_QOS_BESTEFFORT && !non_native)) {  // uniform
      if (!fresh) {
        status = SPX_NRT_ST_INVALID_TOPOLOGY;
      } else if (has_nrt && single) {
        if (pod_scope) {  // singleNUMAPodLevelHandler
          uint32_t pos;
          if (!fits_fast(ns, pod_sets, preq, &pos)) status = SPX_NRT_ST_POD;
        } else {  // singleNUMAContainerLevelHandler
          for (int c = 0; c < n_ctr; ++c) {  // init containers: must fit, never subtracted
            if (ckind_of(c) == SPX_CTR_APP) continue;
            uint32_t pos;
            const bool ok = fits_fast(ns, sets_of(c), creq + c * R, &pos);
            if (status == 0 && !ok) status = ckind_of(c) == SPX_CTR_SIDECAR ? SPX_NRT_ST_SIDECAR_CONTAINER : SPX_NRT_ST_INIT_CONTAINER;
          }
          uint32_t chosen = 0;  // list position picked per app container (for the undo), 4 bits each
          uint32_t placed = 0;  // bit c: container c was subtracted on this lane
          for (int c = 0; c <= last_app; ++c) {
            if (ckind_of(c) != SPX_CTR_APP) continue;
            const Sets st = sets_of(c);
            uint32_t pos;
            const bool ok = fits_fast(ns, st, creq + c * R, &pos);
            const bool live = status == 0;
            if (live && !ok) status = SPX_NRT_ST_CONTAINER;
            if (c == last_app) break;  // nothing reads the table after the last app container
            const bool apply = live && ok;
            adjust_fast(ns, st, creq + c * R, pos, apply, -1.0);
            chosen |= (apply ? pos : 0u) << (4 * c);
            placed |= (apply ? 1u : 0u) << c;
          }
          for (int c = 0; c < last_app; ++c) {  // undo: Filter works on a private copy in the reference
            if (ckind_of(c) != SPX_CTR_APP) continue;
            adjust_fast(ns, sets_of(c), creq + c * R, (chosen >> (4 * c)) & 0xfu, (placed >> c) & 1u, 1.0);
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
      score = score_each_fast<RM, SG>(ns, a, pod_sets, preq, cpu_v);
    } else {  // containerScopeScore: int64(mean) over init + app containers
      int sum = 0;
      for (int c = 0; c < n_ctr; ++c) sum += score_each_fast<RM, SG>(ns, a, sets_of(c), creq + c * R, cpu_v);
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
  const int sg = a.strategy == SPX_NRT_BALANCED_ALLOCATION ? kSgBalanced : (a.strategy == SPX_NRT_LEAST_ALLOCATED ? kSgLeast : kSgMost);
#define SPX_NRTF_CASE(RMV, SGV)                                                              \
  if ((a.n_res <= 4) == (RMV == 4) && sg == SGV) {                                           \
    hipLaunchKernelGGL((k_nrt_fast<RMV, SGV>), dim3(blocks), dim3(256), 0, s, a, n_tiles);  \
    return true;                                                                             \
  }
  SPX_NRTF_CASE(4, kSgLeast)
  SPX_NRTF_CASE(4, kSgMost)
  SPX_NRTF_CASE(4, kSgBalanced)
  SPX_NRTF_CASE(8, kSgLeast)
  SPX_NRTF_CASE(8, kSgMost)
  SPX_NRTF_CASE(8, kSgBalanced)
#undef SPX_NRTF_CASE
  return false;
}

}  // namespace spx
