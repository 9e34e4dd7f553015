// kernels_commit_trimaran.hip — sequential commit loop for the Filter-less profile (Allocatable + TargetLoadPacking +
// LoadVariationRiskBalancing), SURVEY.md section 8f rank 1.
//
// Upstream schedules one pod at a time: Score every node, pick the best, bind; the bound pod enters trimaran's
// ScheduledPodsCache (handler.go:131-139) and from then on adds its predicted CPU utilisation to that node's "missing
// utilisation" (targetloadpacking.go:151-168) until the metrics catch up.  So pod i+1's row differs from what a frozen snapshot
// says in exactly one node — but which one depends on pod i's decision: the chain is inherently sequential and does not shard.
// One workgroup keeps the chain on the device; the per-pod decisions are the output, no table is read or written.
//
// k_commit_trimaran_reg (round 4; the node state lives in registers): 1024 threads x K cells.  Per pod every thread runs ONE
// branch-free pass over its cells — the float32 formula of k_tlp_fast2, the row's worst rounding margin and smallest |u| as
// running max/min, the weighted total folded into a 32-bit key (total << 14 | 16383 - node) by one v_mad_u32_u24 per plugin —
// then the keys are reduced: DPP inside the 16-lane rows, four v_readlane + scalar max across them, one LDS word per wave,
// ONE barrier, and every thread folds the 16 words itself.  Only a thread whose pass saw an ambiguous cell (or a pod that is
// not a float32 integer) walks its cells again with the reference's float64 sequence.  The winner's owner shifts that node's
// constant by the pod's integer millicores and keeps the committed millicores in a register (exact-path input).  Pod values
// and decisions move through LDS in chunks of 256 pods, so the loop body issues no global memory instruction at all.
// Round 3's kernel (512 threads x 20 cells, a divergent exact-path branch per cell, 10-30 spilled VGPRs, a global
// read-modify-write and three global stores per pod): 3.0 us per pod; this one: see DESIGN.md section 3.7.
//
// k_commit_trimaran_mem: float64 throughout, state re-read from global memory per pod (any size, any weights).
#include <cstdlib>

#include "spx_internal.h"
#include "trimaran_math.h"

namespace spx {

namespace {

using namespace trimath;

constexpr int kWave = 64;

__device__ __forceinline__ int64_t shfl_xor_i64(int64_t v, int m) {
  int lo = __shfl_xor(static_cast<int>(v & 0xffffffffLL), m, kWave);
  int hi = __shfl_xor(static_cast<int>(v >> 32), m, kWave);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

// max(v, v of the lane a DPP control pairs this lane with); the control's source is always a live lane here, and 0 — what a
// lane without a source would read — is the identity of an unsigned max, so the backend folds the move into v_max_u32_dpp
template <int CTRL>
__device__ __forceinline__ uint32_t max_dpp(uint32_t v) {
  const uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xf, 0xf, true));
  return o > v ? o : v;
}
// every lane of a 16-lane row ends with the row's maximum: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) {
  v = max_dpp<0xB1>(v);
  v = max_dpp<0x4E>(v);
  v = max_dpp<0x141>(v);
  return max_dpp<0x140>(v);
}
// maximum over the wavefront as a wave-uniform (scalar) value: the four rows are combined on the scalar unit
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = row_max_u32(v);
  const uint32_t r0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 0));
  const uint32_t r1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 16));
  const uint32_t r2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 32));
  const uint32_t r3 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 48));
  const uint32_t a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
  return a > b ? a : b;
}

// One cell of the float32 formulation, arranged for the issue rates of this part (compares, conversions, v_rndne, v_min/max and
// 24-bit multiplies issue at half the rate of add / fma / logic ops): the branch is picked on u's sign bit with two v_bfi (the
// +0.0 case lands in the ambiguity band |u| <= kTolU anyway); the value is computed scaled by 1/256 — the coefficients carry
// the factor, a power of two, so nothing rounds differently — with the fma's clamp modifier cutting it to [0, 1], i.e. the
// score to [0, 256], for free; adding 1.5 * 2^23 rounds 256 * xs to the nearest-even integer in the mantissa's low bits
// (what v_rndne + v_cvt_pk_u8 did in 4 issue slots).  `d` is the distance of the clamped value from that integer.
constexpr float kMagic = 12582912.0f;
__device__ __forceinline__ void tlp_cell32(const float4& rk, float pod_f, float tfs, float& u, float& d, uint32_t& tb) {
  u = (pod_f + rk.x) + rk.y;
  const int m = __float_as_int(u) >> 31;  // all ones: u < 0 or -0.0
  const float coef = __int_as_float((m & __float_as_int(rk.w)) | (~m & __float_as_int(rk.z)));
  const float off = __int_as_float((m & __float_as_int(100.0f / 256.0f)) | (~m & __float_as_int(tfs)));
  const float xs = __builtin_amdgcn_fmed3f(__builtin_fmaf(coef, u, off), 0.0f, 1.0f);
  const float y = __builtin_fmaf(xs, 256.0f, kMagic);
  const float rr = y - kMagic;
  d = __builtin_fabsf(__builtin_fmaf(xs, 256.0f, -rr));
  tb = static_cast<uint32_t>(__float_as_int(y)) & 0x1ffu;
}

// Two cells (a group's two nodes) at once in the packed float32 instructions: the same operations on the same values as
// tlp_cell32 — bit-identical u, d and score byte — in 11 instead of 2 x 8.5 instructions.  Both branches of a cell come out of ONE
// v_pk_fma_f32 ((coefficient for u > 0, for u <= 0) * (u, u) + (offsets), clamp modifier), the sign bit of u picks one; the
// rounding and its margin are packed across the two cells.  (The builtins do not produce the clamp modifier or the operand
// selectors on packed float32: those two instructions are written out.)
typedef float F32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tlp_cell_pair32(const F32x2& bh, const F32x2& bl, const F32x2& ca, const F32x2& cb, float pod_f, const F32x2& off2,
                                                float (&u)[2], float (&d)[2], uint32_t (&tb)[2]) {
  const F32x2 u2 = (F32x2{pod_f, pod_f} + bh) + bl;
  F32x2 xa, xb;  // (value on the u > 0 branch, on the u <= 0 branch), clamped to [0, 1]
  asm("v_pk_fma_f32 %0, %2, %4, %5 op_sel_hi:[1,0,1] clamp\n\t"  // (one statement: the compiler pads every asm statement with an s_nop)
      "v_pk_fma_f32 %1, %3, %4, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1] clamp"
      : "=&v"(xa), "=&v"(xb)
      : "v"(ca), "v"(cb), "v"(u2), "v"(off2));
  const F32x2 xs{__float_as_int(u2.x) < 0 ? xa.y : xa.x, __float_as_int(u2.y) < 0 ? xb.y : xb.x};  // sign bit set: u < 0 or -0.0
  const F32x2 y = __builtin_elementwise_fma(xs, F32x2{256.0f, 256.0f}, F32x2{kMagic, kMagic});
  const F32x2 rr = y - F32x2{kMagic, kMagic};
  const F32x2 dd = __builtin_elementwise_fma(xs, F32x2{256.0f, 256.0f}, -rr);
  u[0] = u2.x, u[1] = u2.y;
  d[0] = __builtin_fabsf(dd.x), d[1] = __builtin_fabsf(dd.y);
  tb[0] = static_cast<uint32_t>(__float_as_int(y.x)) & 0x1ffu, tb[1] = static_cast<uint32_t>(__float_as_int(y.y)) & 0x1ffu;
}

constexpr int kChunk = 256;  // pods whose values / decisions are staged in LDS at a time
constexpr int kGroup = 2;    // consecutive nodes per cell group (one 16-bit load of an LVRB row)

// cell k of thread tid = node ((k / 2) * T + tid) * 2 + k % 2
template <int K, int T, bool kHasL, bool kTies>
__global__ __launch_bounds__(T) void k_commit_trimaran_reg(CommitArgs c) {
  static_assert(K % kGroup == 0 && T % kWave == 0 && T >= kChunk, "cell groups / staging threads");
  constexpr int W = T / kWave;
  static_assert(W <= 16 && (W & (W - 1)) == 0, "the waves' keys are folded inside one 16-lane row");
  static_assert((T & (T - 1)) == 0, "owner thread = group index mod T as a mask");
  __shared__ __align__(16) uint32_t s_key[2][W];
  __shared__ int s_tie[2];
  __shared__ int64_t s_pod[2][kChunk];       // the chunk's pod values (owner / exact path)
  __shared__ float s_podf[2][kChunk + 1];    // ... as float32; NaN marks a value that is not a float32 integer (exact path for the row)
  __shared__ int32_t s_node[kChunk], s_score[kChunk], s_ties[kChunk];
  const TrimaranArgs& a = c.t;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool A = c.use_mask & 1u, Tl = c.use_mask & 2u;
  auto node_of = [&](int k) -> int { return ((k >> 1) * T + tid) * kGroup + (k & 1); };
  const double t = a.tlp_target;
  const double c1 = t / (100.0 - t), c2 = (100.0 - t) / t;
  const bool fast_ok = t >= 1.0 && t <= 99.0;
  const float tfs = static_cast<float>(t) * (1.0f / 256.0f);
  const F32x2 off2{tfs, 100.0f / 256.0f};  // the two branches' offsets (scaled)
  constexpr float kHalf = 0.5f - kTol32;
  const uint32_t wt14 = Tl ? static_cast<uint32_t>(c.w_tlp) << 14 : 0u;
  const uint32_t wl14 = kHasL ? static_cast<uint32_t>(c.w_lvrb) << 14 : 0u;

  // per cell (b2h, b2l, coefficient for u > 0, coefficient for u <= 0 — both scaled by 1/256); NaN b2h = the cell always takes the exact
  // path.  Held the way the packed pass reads them — b2h and b2l of a group's two cells side by side, a cell's two coefficients side by
  // side — so that no move stands between the registers and the v_pk instructions (a float4 per cell cost 12 v_mov per pass)
  F32x2 b_hi[K / kGroup], b_lo[K / kGroup], coef[K];
  uint32_t base[K];     // (w_alloc * Allocatable's normalised score) << 14 | 16383 - node; 0 for a cell past the node list
  // millicores committed to the node by this loop and not yet added to c.missing[] (exact-path input): one LDS word per cell —
  // touched by the owner at a commit and by the exact path only, so it need not cost the pass K registers
  __shared__ int32_t s_delta[K][T];
  uint32_t gmask[K / kGroup];  // kHasL: which bytes of the group's 16-bit load are nodes
  bool lane_nan = false;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int n = node_of(k);
    const bool in = n < a.n_nodes;
    float4 rk = (in && Tl) ? tlp_fast_consts(static_cast<double>(a.cap_cpu_milli[n]), a.tlp_cpu_util[n], static_cast<double>(c.missing[n]), a.tlp_valid[n] != 0, t, c1, c2)
                        : float4{1e30f, 0.0f, -1.0f, 0.0f};
    if (in && Tl && !fast_ok) rk.x = __builtin_nanf("");
    lane_nan |= rk.x != rk.x;
    if (k & 1) b_hi[k >> 1].y = rk.x, b_lo[k >> 1].y = rk.y;
    else b_hi[k >> 1].x = rk.x, b_lo[k >> 1].x = rk.y;
    coef[k] = F32x2{rk.z * (1.0f / 256.0f), rk.w * (1.0f / 256.0f)};
    base[k] = in ? (((A ? static_cast<uint32_t>(c.w_alloc) * a.alloc_norm[n] : 0u) << 14) | (16383u - static_cast<uint32_t>(n))) : 0u;
    s_delta[k][tid] = 0;
    if ((k & 1) == 0) gmask[k >> 1] = (in ? 0xffu : 0u) | (n + 1 < a.n_nodes ? 0xff00u : 0u);
  }
  const int64_t rows = a.row_end - a.row_begin;
  // chunk 0's pod values
  auto stage = [&](int b, int64_t v) {
    s_pod[b][tid] = v;
    s_podf[b][tid] = (v < 0 || v >= (1 << 23)) ? __builtin_nanf("") : static_cast<float>(v);
  };
  if (tid < kChunk) stage(0, (Tl && tid < rows) ? a.tlp_pod_milli[a.row_begin + tid] : 0);
  if (tid < 2) s_tie[tid] = 0;
  // may the 32-bit per-cell sums of committed millicores take unchecked adds?  Yes when every pod value of the row range is a
  // float32 integer and their total fits 31 bits (one node could receive them all) — a walk over the column, once per launch
  __shared__ unsigned long long s_sum;
  __shared__ int s_out_of_range;
  if (tid == 0) s_sum = 0, s_out_of_range = 0;
  __syncthreads();
  {
    unsigned long long mine = 0;
    bool odd_value = false;
    if (Tl)
      for (int64_t q = tid; q < rows; q += T) {
        const int64_t v = a.tlp_pod_milli[a.row_begin + q];
        odd_value |= v < 0 || v >= (1 << 23);
        mine += static_cast<unsigned long long>(v);
      }
    if (mine) atomicAdd(&s_sum, mine);
    if (odd_value) atomicOr(&s_out_of_range, 1);
  }
  uint32_t lv_next[K / kGroup];
#pragma unroll
  for (int g = 0; g < K / kGroup; ++g) {
    lv_next[g] = 0;
    if constexpr (kHasL) {
      const int n0 = (g * T + tid) * kGroup;
      if (n0 < a.n_nodes) lv_next[g] = *reinterpret_cast<const uint16_t*>(c.lv_table + a.row_begin * a.row_stride + n0);
    }
  }
  __syncthreads();
  const bool sum_fits = __builtin_amdgcn_readfirstlane(static_cast<int>(s_out_of_range == 0 && s_sum <= 0x7fffffffull)) != 0;

  for (int64_t chunk0 = 0; chunk0 < rows; chunk0 += kChunk) {
    const int buf = static_cast<int>((chunk0 / kChunk) & 1);
    const int n_here = static_cast<int>(rows - chunk0 < kChunk ? rows - chunk0 : kChunk);
    // the next chunk's pod values: requested now, parked in LDS when this chunk is done
    int64_t nxt = 0;
    if (tid < kChunk && Tl && chunk0 + kChunk + tid < rows) nxt = a.tlp_pod_milli[a.row_begin + chunk0 + kChunk + tid];
    float pod_f = s_podf[buf][0];
    for (int p = 0; p < n_here; ++p) {
      const float pod_next = s_podf[buf][p + 1];  // (the slot behind the chunk's last pod is never used)
      const bool pod_bad = pod_f != pod_f;         // not exact as a float32 integer: exact path for the row
      const int par = p & 1;
      // LVRB carries no commit state: its frozen-snapshot rows (swept just before this loop) stay valid.  The next pod's bytes
      // are requested now and used in the next iteration, so their latency is off the dependent chain.
      uint32_t lv_now[K / kGroup];
      if constexpr (kHasL) {
        const int64_t np = a.row_begin + chunk0 + (chunk0 + p + 1 < rows ? p + 1 : p);
#pragma unroll
        for (int g = 0; g < K / kGroup; ++g) {
          lv_now[g] = lv_next[g] & gmask[g];  // (masked where it is used: masking the load's result would wait for it right away)
          const int n0 = (g * T + tid) * kGroup;
          if (n0 < a.n_nodes) lv_next[g] = *reinterpret_cast<const uint16_t*>(c.lv_table + np * a.row_stride + n0);
        }
      }
      auto lv_part = [&](int k) -> uint32_t {
        if constexpr (kHasL) return __umul24((lv_now[k >> 1] >> (8 * (k & 1))) & 0xffu, wl14);
        return 0u;
      };
      // ---- the branch-free pass
      uint32_t kmax = 0, btot = 0;
      int ties = 0;
      float worst = 0.0f, minu = 1e30f;
#pragma unroll
      for (int g = 0; g < K / kGroup; ++g) {
        float u[2], d[2];
        uint32_t tb[2];
        tlp_cell_pair32(b_hi[g], b_lo[g], coef[2 * g], coef[2 * g + 1], pod_f, off2, u, d, tb);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = 2 * g + j;
          worst = __builtin_fmaxf(worst, d[j]);
          minu = __builtin_fminf(minu, __builtin_fabsf(u[j]));
          const uint32_t key = __umul24(tb[j], wt14) + base[k] + lv_part(k);
          kmax = key > kmax ? key : kmax;
          if constexpr (kTies) {
            const uint32_t tot = key >> 14;
            const bool real = key != 0;  // a cell past the node list carries key 0; a node's key never is (node < 16383)
            ties = (real && tot > btot) ? 1 : ((real && tot == btot) ? ties + 1 : ties);
            btot = (real && tot > btot) ? tot : btot;
          }
        }
      }
      const bool any = pod_bad || lane_nan || !(worst < kHalf) || !(minu > kTolU);
      if (__builtin_expect(any, 0)) {
        // rare per thread (~8e-5 of the cells are ambiguous; per pod, with 10^4 cells, every other one has such a thread): the
        // cells that are not provably the reference's result again, with its float64 sequence on the node's columns.
        // Measured alternatives (profiles/r04/commit_trimaran_ab.md): skipping cells that cannot reach the thread's best key, a
        // second reduction round only for contenders, the float64 sequence as a real function — all slower than this plain form:
        // the wave in here runs alone on its SIMD while fifteen others wait at the barrier, so what counts is its instruction count.
        float pf = pod_f;
        asm volatile("" : "+v"(pf));  // opaque copy: keeps the two passes apart (no flags carried across the branch)
        const double pod_milli = static_cast<double>(s_pod[buf][p]);
        kmax = 0, btot = 0, ties = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          float u, d;
          uint32_t tb;
          tlp_cell32(float4{(k & 1) ? b_hi[k >> 1].y : b_hi[k >> 1].x, (k & 1) ? b_lo[k >> 1].y : b_lo[k >> 1].x, coef[k].x, coef[k].y}, pf, tfs, u, d, tb);
          const bool unknown = pod_bad || !(__builtin_fabsf(u) > kTolU);  // (NaN constant: u is NaN)
          const bool tied = !(d < kHalf);
          uint32_t key = base[k] != 0u ? __umul24(tb, wt14) + base[k] + lv_part(k) : 0u;
          if (base[k] != 0u && Tl && (unknown || tied)) {
            int n = node_of(k);
            asm volatile("" : "+v"(n));  // opaque: keeps the K cells' column addresses from being precomputed outside the pod loop
            TlpNode tn;
            tn.cap = static_cast<double>(a.cap_cpu_milli[n]);
            tn.util_millis = (a.tlp_cpu_util[n] / 100.0) * tn.cap;
            tn.missing = static_cast<double>(c.missing[n] + s_delta[k][tid]);
            tn.valid = a.tlp_valid[n] != 0;
            bool zero;
            const double xe = tlp_unrounded(tn, pod_milli, t, &zero);
            tb = zero ? 0u : to_u8(xe);
            key = __umul24(tb, wt14) + base[k] + lv_part(k);
          }
          kmax = key > kmax ? key : kmax;
          if constexpr (kTies) {
            const uint32_t tot = key >> 14;
            const bool real = key != 0;
            ties = (real && tot > btot) ? 1 : ((real && tot == btot) ? ties + 1 : ties);
            btot = (real && tot > btot) ? tot : btot;
          }
        }
      }
      // ---- one key per wave, one barrier, every thread folds the waves' keys itself (double-buffered by pod parity)
      const uint32_t wkey = wave_max_u32(kmax);
      if (lane == 0) s_key[par][wave] = wkey;
      __syncthreads();
      // lane l reads wave (l mod W)'s key: every 16-lane row holds all of them, its maximum is the workgroup's
      const uint32_t gkey = row_max_u32(s_key[par][lane & (W - 1)]);
      const bool found = gkey != 0;
      const int win = found ? static_cast<int>(16383u - (gkey & 16383u)) : -1;
      if constexpr (kTies) {
        if (found && kmax != 0 && (kmax >> 14) == (gkey >> 14)) atomicAdd(&s_tie[par], ties);
        __syncthreads();
      }
      if (tid == 0) {
        s_node[p] = win;
        s_score[p] = found ? static_cast<int32_t>(gkey >> 14) : 0;
        if constexpr (kTies) {
          s_ties[p] = s_tie[par];
          s_tie[par] = 0;  // next written two pods from now, behind the next pod's barriers
        }
      }
      // the winner's owner advances the node: the real number b2h + b2l grows by exactly the pod's integer millicores (which
      // tracks the float64 b within ~1e-10), and the millicores join the exact path's missing utilisation.  This sits on the
      // chain — the owner's wave starts the next pod's pass after it while fifteen waves are already in theirs — so it is kept to
      // a scalar branch into the owner's wave, a scalar pick of the cell, one select, and an LDS add that returns nothing
      // (every gkey-derived value is the same in all lanes: made scalar where it steers a branch)
      if (Tl && found) {
        const int owner = (win >> 1) & (T - 1);
        const int owner_s = __builtin_amdgcn_readfirstlane(owner);
        if ((owner_s >> 6) == wave) {
          const int kk = __builtin_amdgcn_readfirstlane(((win >> 1) / T) * kGroup + (win & 1));
          const bool me = owner_s == tid;
#pragma unroll
          for (int k = 0; k < K; ++k)
            if (k == kk) {
              const float bx = (k & 1) ? b_hi[k >> 1].y : b_hi[k >> 1].x;  // (vector elements are not addressable: read, then written back below)
              const float nb = bx + pod_f;
              const float nv = (pod_bad || !(__builtin_fabsf(nb) < 8388607.0f)) ? __builtin_nanf("") : nb;
              if (k & 1) b_hi[k >> 1].y = me ? nv : bx;
              else b_hi[k >> 1].x = me ? nv : bx;
              lane_nan |= me && nv != nv;
            }
          if (me) {
            int32_t* cell = &s_delta[0][0] + kk * T + tid;
            if (sum_fits) {  // (pod_f is the pod's value, exactly)
              __hip_atomic_fetch_add(cell, static_cast<int32_t>(pod_f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {  // a queue whose values could overflow the 32-bit cell: checked adds, folded into the column when they would
              const int64_t nd = static_cast<int64_t>(*cell) + s_pod[buf][p];
              if (nd >= INT32_MIN && nd <= INT32_MAX) {
                *cell = static_cast<int32_t>(nd);
              } else {
                c.missing[win] += nd;
                *cell = 0;
              }
            }
          }
        }
      }
      pod_f = pod_next;
    }
    // ---- chunk boundary: decisions out, the next chunk's pod values in
    __syncthreads();
    if (tid < n_here) {
      c.out_node[chunk0 + tid] = s_node[tid];
      c.out_score[chunk0 + tid] = s_score[tid];
      if constexpr (kTies) {
        if (c.out_ties) c.out_ties[chunk0 + tid] = s_ties[tid];
      }
    }
    if (tid < kChunk) stage(buf ^ 1, nxt);
    __syncthreads();
  }
  // the committed millicores join the column (the caller copies it out)
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (s_delta[k][tid] != 0) c.missing[node_of(k)] += s_delta[k][tid];
}

// float64 throughout, state re-read from global memory per pod
template <int kCommitThreads>
__global__ __launch_bounds__(kCommitThreads) void k_commit_trimaran_mem(CommitArgs c) {
  __shared__ int64_t s_best[kCommitThreads / kWave];
  __shared__ int s_node[kCommitThreads / kWave];
  __shared__ int s_ties[kCommitThreads / kWave];
  const TrimaranArgs& a = c.t;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const bool A = c.use_mask & 1u, T = c.use_mask & 2u, L = c.use_mask & 4u;
  const double t = a.tlp_target;
  for (int64_t pod = a.row_begin; pod < a.row_end; ++pod) {
    const int64_t pod_i = T ? a.tlp_pod_milli[pod] : 0;
    const double pod_milli = static_cast<double>(pod_i);
    int64_t best = INT64_MIN;
    int best_n = INT32_MAX, ties = 0;
    for (int64_t n = tid; n < a.n_nodes; n += kCommitThreads) {
      int64_t total = 0;
      if (A) total += c.w_alloc * static_cast<int64_t>(a.alloc_norm[n]);
      if (T) {
        TlpNode tn;
        tn.cap = static_cast<double>(a.cap_cpu_milli[n]);
        tn.util_millis = (a.tlp_cpu_util[n] / 100.0) * tn.cap;
        tn.missing = static_cast<double>(c.missing[n]);
        tn.valid = a.tlp_valid[n] != 0;
        bool zero;
        const double x = tlp_unrounded(tn, pod_milli, t, &zero);
        total += c.w_tlp * static_cast<int64_t>(zero ? 0u : to_u8(x));
      }
      if (L) total += c.w_lvrb * static_cast<int64_t>(c.lv_table[pod * a.row_stride + n]);
      if (total > best) {  // a thread walks its nodes in increasing order: `>` keeps the lowest index among equals
        best = total;
        best_n = static_cast<int>(n);
        ties = 1;
      } else if (total == best) {
        ++ties;
      }
    }
    // wave-level then block-level argmax (every lane is live: no divergence around the shuffles)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const int64_t ob = shfl_xor_i64(best, m);
      const int on = __shfl_xor(best_n, m, 64);
      const int ot = __shfl_xor(ties, m, 64);
      if (ob > best || (ob == best && on < best_n)) {
        ties = ob > best ? ot : ties + ot;
        best = ob;
        best_n = on;
      } else if (ob == best) {
        ties += ot;
      }
    }
    if (lane == 0) {
      s_best[wave] = best;
      s_node[wave] = best_n;
      s_ties[wave] = ties;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < kCommitThreads / kWave; ++w) {
        if (s_best[w] > best || (s_best[w] == best && s_node[w] < best_n)) {
          ties = s_best[w] > best ? s_ties[w] : ties + s_ties[w];
          best = s_best[w];
          best_n = s_node[w];
        } else if (s_best[w] == best) {
          ties += s_ties[w];
        }
      }
      const bool any = best_n != INT32_MAX;
      c.out_node[pod - a.row_begin] = any ? best_n : -1;
      c.out_score[pod - a.row_begin] = any ? best : 0;
      if (c.out_ties) c.out_ties[pod - a.row_begin] = any ? ties : 0;
      if (any && T) {
        c.missing[best_n] += pod_i;  // the bound pod's predicted utilisation, from now on
        __threadfence_block();
      }
    }
    __syncthreads();
  }
}

template <int K>
void launch_reg(const CommitArgs& c, hipStream_t s) {
  constexpr int T = 1024;  // (512 threads x 20 cells: 2.5 us per pod against 1.7 — two waves per SIMD do not cover the pass's dependent issue)
  const bool l = (c.use_mask & 4u) != 0, ties = c.out_ties != nullptr;
  if (l && ties) hipLaunchKernelGGL((k_commit_trimaran_reg<K, T, true, true>), dim3(1), dim3(T), 0, s, c);
  else if (l) hipLaunchKernelGGL((k_commit_trimaran_reg<K, T, true, false>), dim3(1), dim3(T), 0, s, c);
  else if (ties) hipLaunchKernelGGL((k_commit_trimaran_reg<K, T, false, true>), dim3(1), dim3(T), 0, s, c);
  else hipLaunchKernelGGL((k_commit_trimaran_reg<K, T, false, false>), dim3(1), dim3(T), 0, s, c);
}

}  // namespace

void launch_commit_trimaran(const CommitArgs& c, hipStream_t s) {
  if (c.t.row_end <= c.t.row_begin) return;
  const bool from_memory = (c.t.opts & kOptCommitFromMemory) != 0;  // SPX_OPT_COMMIT_FROM_MEMORY
  // the register variant's 32-bit key: every weighted total below 2^18, every weight a 24-bit multiplicand after << 14
  const bool key_fits = c.w_alloc >= 0 && c.w_tlp >= 0 && c.w_lvrb >= 0 && c.w_alloc < 1024 && c.w_tlp < 1024 && c.w_lvrb < 1024 &&
                        (c.w_alloc + c.w_tlp + c.w_lvrb) * 255 < (int64_t{1} << 18);
  const int64_t n = c.t.n_nodes;
  if (!from_memory && key_fits && n <= 12 * 1024) {
    if (n <= 4 * 1024) launch_reg<4>(c, s);
    else if (n <= 8 * 1024) launch_reg<8>(c, s);
    else if (n <= 10 * 1024) launch_reg<10>(c, s);
    else launch_reg<12>(c, s);
  } else {
    hipLaunchKernelGGL((k_commit_trimaran_mem<1024>), dim3(1), dim3(1024), 0, s, c);
  }
}

}  // namespace spx
