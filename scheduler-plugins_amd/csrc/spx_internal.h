// spx_internal.h — shared between the engine (host) and the kernel translation units.
// Not part of the public ABI (that is include/spx.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/spx.h"
#include "nrt_rank_layout.h"

namespace spx {

// Score rows are written as 16-byte vectors, so every uint8 result row is padded to this.
constexpr int64_t kRowAlign = 16;
// Default row padding of engine-owned tables (rows start on a cache-line boundary).
constexpr int64_t kRowPad = 128;

inline int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// Row range of a sweep: [row_begin, row_end) from the launch, or — row_ptr set — the single row whose index sits in device memory.
// The sequential commit loop replays ONE captured graph per pod; the graph's kernels read the row from the counter k_commit_apply
// advances.  Call first thing in a kernel, on its by-value argument struct.
#define SPX_RESOLVE_ROWS(a)              \
  do {                                   \
    if ((a).row_ptr) {                   \
      (a).row_begin = *(a).row_ptr;      \
      (a).row_end = (a).row_begin + 1;   \
    }                                    \
  } while (0)

// spx_fetch_stats counters: kStatSlots per plugin, summed at fetch (same-address atomics serialise in the L2)
constexpr int kStatSlots = 64;
constexpr int kStatStride = 16;  // uint64 per slot: every slot on its own 128-byte line
constexpr size_t kStatBytes = static_cast<size_t>(SPX_NUM_PLUGINS) * kStatSlots * kStatStride * sizeof(unsigned long long);

// per-launch switches derived from the engine's options (spx_set_option); carried in every Args struct as `opts`
enum : uint32_t {
  kOptTrimaranExact = 1u << 0,     // TLP / LVRB: reference float64 sequence only
  kOptNrtGeneric = 1u << 1,        // NRT: int64 kernel
  kOptNetGeneric = 1u << 2,        // NetworkOverhead: per-node kernel
  kOptNrtSingleLaunch = 1u << 3,   // NRT Least/MostAllocated: Filter and Score in one launch
  kOptCommitFromMemory = 1u << 4,  // commit loop: node state in memory
  kOptPeaksWideA = 1u << 5,        // Peaks: 8 nodes per lane in the min/max pass
  kOptPeaksWideB = 1u << 6,        // Peaks: 8 nodes per lane in the write pass
  kOptTlpNoAmbTable = 1u << 7,     // TLP fast sweep: per-cell exactness bookkeeping in every row (SPX_OPT_TLP_AMB_TABLE 0)
  kOptPeaksEstimate = 1u << 8,     // Peaks: float32 interval estimates, raw_score only where they cannot decide (SPX_OPT_PEAKS_ESTIMATE)
  kOptPeaksEst8 = 1u << 9,         //   8 instead of 16 nodes per lane
};

// what the multi-device layer (spx_multi.hip) needs to see of an engine
struct EngineView {
  int device;
  hipStream_t stream;
  int64_t n_nodes, n_pods, row_stride;
  void* best;        // decision block [score int64 P | node int32 P | ties int32 P | feasible int32 P], NULL before the first argmax
  bool best_valid;
  uint32_t evaluated;  // bit per plugin: its tables describe the resident snapshot and pod rows
  hipEvent_t ev0, ev1;  // recorded around the last spx_eval / spx_decide
  bool timed;
};
EngineView engine_view(spx_engine* e);

// ---------------------------------------------------------------- Allocatable
struct AllocPrepArgs {
  int64_t n_nodes;
  int64_t row_stride;
  int32_t n_res;
  int32_t mode;
  const int64_t* alloc;    // [n_res][n_nodes]
  const int64_t* weight;   // [n_res] (device)
  int64_t* raw;            // [n_nodes] out: Allocatable.Score per node
  uint8_t* norm;           // [row_stride] out: NormalizeScore over the full node list
  uint32_t* rel;           // [row_stride + 1] out: raw - global min as uint32, then a flag: 1 when that form is exact
};
void launch_alloc_prepare(const AllocPrepArgs& a, hipStream_t s);

constexpr int32_t kTlpAmbSize = 1 << 16;  // pod values (millicores) k_tlp_amb_build's table covers; larger pods take the checked cell

// ---------------------------------------------------------------- fused Allocatable + TLP + LVRB sweep
struct TrimaranArgs {
  uint32_t opts;
  int64_t n_nodes;
  int64_t row_stride;
  int64_t row_begin;
  int64_t row_end;
  const int64_t* row_ptr;  // when set: evaluate the single row *row_ptr (sequential commit: the row counter lives on the device)
  // Allocatable: pod-independent normalized row (no per-row feasibility mask)
  const uint8_t* alloc_norm;
  // TargetLoadPacking
  const int64_t* cap_cpu_milli;
  const double* tlp_cpu_util;
  const int64_t* tlp_missing_milli;
  const uint8_t* tlp_valid;
  const int64_t* tlp_pod_milli;
  double tlp_target;
  // LoadVariationRiskBalancing
  const int64_t* lv_alloc_cpu_milli;
  const int64_t* lv_alloc_mem;
  const double* lv_cpu_avg;
  const double* lv_cpu_std;
  const double* lv_mem_avg;
  const double* lv_mem_std;
  const uint8_t* lv_flags;
  const int64_t* lv_req_cpu_milli;
  const int64_t* lv_req_mem;
  double lv_margin;
  double lv_sensitivity;
  double* lv_exact;  // scratch [n_nodes][8]: per-node exact LVRB state for the fast kernel's fallback
  float* lv_fast;    // scratch [ceil(row_stride/512)*512][8]: LVRB fast constants, tile-transposed (k_lvrb_prepare_fast)
  float* tlp_fast;   // scratch [ceil(row_stride/1024)*1024][4]: TLP fast constants, tile-transposed (k_tlp_prepare_fast)
  uint32_t* tlp_amb;     // scratch [tlp_amb_size]: per pod value, the node tiles (bit tile & 31) holding a cell the float32 sweep cannot prove (k_tlp_amb_build); NULL = checked cells everywhere
  int32_t tlp_amb_size;  // pod values at or above it take the checked cell
  uint32_t* lv_amb;      // scratch [lvrb_amb_bytes() / 4]: k_lvrb_amb_build's table (cpu millicores | memory MiB | always-checked tiles); NULL = checked cells everywhere
  // what the two tables were built for besides the columns (host, may be NULL): {nodes per tile, row stride, bits of the target / table size};
  // a launch whose geometry differs rebuilds them whatever the flags say (advisor, round 5: the flags alone relied on every launcher
  // using one tiling)
  int64_t* tlp_amb_geom;
  int64_t* lv_amb_geom;
  bool* lv_amb_built;    // as tlp_amb_built, for lv_exact / lv_fast / lv_amb: cleared by every writer of the LVRB node columns, margin or sensitivity
  bool* tlp_amb_built;   // host flag (may be NULL = always rebuild): true while tlp_amb describes the node columns / target in place; the launcher builds
                         // the table when it is false and sets it; the owner clears it whenever a column k_tlp_amb_build reads, or the target, changes
  unsigned long long* stats;  // [SPX_NUM_PLUGINS][kStatSlots][kStatStride] cells the fast sweeps re-evaluated with the reference sequence (spx_fetch_stats); may be NULL
  // outputs: uint8 [n_pods][row_stride] each (NULL = plugin not evaluated)
  uint8_t* out_alloc;
  uint8_t* out_tlp;
  uint8_t* out_lvrb;
};
// evaluates the plugins whose out_* pointer is non-NULL
void launch_trimaran(const TrimaranArgs& a, hipStream_t s);
size_t lvrb_amb_bytes();
// sequential commit loop over pod rows [t.row_begin, t.row_end) for Allocatable (bit 0) / TLP (bit 1) / LVRB (bit 2)
struct CommitArgs {
  TrimaranArgs t;       // inputs; alloc_norm must be prepared
  uint32_t use_mask;
  int64_t w_alloc, w_tlp, w_lvrb;
  const uint8_t* lv_table;  // LVRB's score table [pods][row_stride], evaluated for the rows beforehand (LVRB has no commit state)
  int64_t* missing;     // [n_nodes] mutable copy of tlp_missing_milli: advances with every commit
  int32_t* out_node;    // [rows]
  int64_t* out_score;
  int32_t* out_ties;    // may be NULL
};
void launch_commit_trimaran(const CommitArgs& c, hipStream_t s);
// decisions-only sweep of the Allocatable + TargetLoadPacking profile (no score tables written): per pod the best weighted
// total, the lowest node reaching it, the tie count — in the layout spx_fetch_best reads
struct DecideLaunch {
  TrimaranArgs t;        // inputs; alloc_norm prepared when use_alloc; tlp_fast scratch set
  bool use_alloc;
  int32_t w_alloc, w_tlp;
  int32_t n_extra;       // score tables of other Score-only plugins, evaluated for the rows beforehand, folded into the total
  int32_t w_extra[3];
  const uint8_t* extra[3];
  void* scratch;         // decide_scratch_bytes(row_stride, rows)
  int64_t* best_score;   // [n_pods] ...
  int32_t* best_node;
  int32_t* best_ties;
  int32_t* best_feasible;
};
size_t decide_scratch_bytes(int64_t row_stride, int64_t rows);
void launch_decide_trimaran(const DecideLaunch& d, hipStream_t s);
// raw int64 Score() of one pod row for `plugin` (SPX_PLUGIN_TLP / SPX_PLUGIN_LVRB)
void launch_trimaran_raw(const TrimaranArgs& a, int plugin, int64_t pod_row, int64_t* out, hipStream_t s);

// ---------------------------------------------------------------- LowRiskOverCommitment (kernels_lroc.hip)
constexpr int kLrocTabCols = 14;
struct LrocArgs {
  int64_t n_nodes;
  int64_t row_stride;
  int64_t row_begin;
  int64_t row_end;
  // node columns: the LVRB columns of spx_trimaran_nodes_soa ...
  const int64_t* alloc_cpu_milli;
  const int64_t* alloc_mem;
  const double* cpu_avg;
  const double* cpu_std;
  const double* mem_avg;
  const double* mem_std;
  const uint8_t* flags;
  // ... and spx_lroc_nodes_soa
  const int64_t* node_req_cpu;
  const int64_t* node_req_mem;
  const int64_t* node_lim_cpu;
  const int64_t* node_lim_mem;
  // pod columns (spx_lroc_pods_soa)
  const int64_t* pod_req_cpu;
  const int64_t* pod_req_mem;
  const int64_t* pod_lim_cpu;
  const int64_t* pod_lim_mem;
  double sqrt_window;   // sqrt(SmoothingWindowSize)
  double w_cpu, w_mem;  // RiskLimitWeights
  // per-node table written by launch_lroc_prepare: [kLrocTabCols][row_stride] doubles
  //   0/1: (1 - w) * riskLoad for cpu / memory (slot 0 is NaN for a node without metrics);
  //   2..7: float64 images of requested / limits / capacity for cpu, then memory;
  //   8..13: the fast form's constants (limits - capacity, limits - requested per resource; the two riskLoad terms as float32)
  double* node_tab;
  // float32 sweep only: [n_pods_total][8] float32 = pod limit as the sum of two float32 (high parts: cpu, memory; low parts: cpu, memory), limit - request
  // (cpu, memory), 1 for a pod without requests and limits, 0 — prepared by the engine at upload when its preconditions hold
  // (columns below 2^47, limits not below requests).  NULL = not available: the float64 / int64 sweep runs
  const float* pod_f32;
  int64_t n_pods_total;
  int32_t exact53;      // every integer the sweep touches is in [0, 2^52): float64 sums and differences are exact
  unsigned long long* stats;  // as TrimaranArgs::stats
  uint8_t* out_score;   // [n_pods][row_stride]
};
void launch_lroc_prepare(const LrocArgs& a, hipStream_t s);
void launch_lroc(const LrocArgs& a, hipStream_t s);

// ---------------------------------------------------------------- Peaks (kernels_peaks.hip)
struct PeaksArgs {
  uint32_t opts;
  int64_t n_nodes;
  int64_t row_stride;
  int64_t row_begin;
  int64_t row_end;
  const int64_t* cap_cpu_milli;  // [N] node.Status.Capacity
  const double* cpu_util;        // [N] percent
  const uint8_t* valid;          // [N]
  const double* k1;              // [N] power model
  const double* k2;
  const int64_t* pod_cpu_milli;  // [P]
  const uint8_t* other_status[3];  // Filter plugins' status tables [P][row_stride] (0 = passed), NULL = unused
  int64_t* row_min;              // [P] scratch: min / max of the raw scores over each pod's feasible nodes
  int64_t* row_max;
  float* row_c;                  // [P][4] scratch of the interval-estimate write pass (k_peaks_rowconst)
  double* node_tab;              // scratch of the interval-estimate passes: 96 bytes per column of a row (k_peaks_nodetab)
  void* seg;                     // the undecided cells' list: a segment per wave (peaks_est_scratch)
  int32_t* seg_n;                // entries per segment
  int32_t est_pods;              // pod rows per chunk of the interval-estimate sweeps (peaks_est_plan)
  uint8_t* out_score;            // [P][row_stride]
  int64_t* out_raw;              // when set: raw int64 scores of row_begin only, no table writes
  // when set: the sweep walks these n_list rows — the first row of each distinct pod cpu request, ascending — instead of
  // [row_begin, row_end); launch_rows_expand copies each to the rows that repeat it
  const int32_t* row_list;
  int64_t n_list;
};
void launch_peaks(const PeaksArgs& a, hipStream_t s);
int peaks_est_plan(uint32_t opts, int64_t row_stride, int64_t swept, size_t* seg_bytes, size_t* cnt_bytes);

// ---------------------------------------------------------------- NodeResourceTopologyMatch
struct NrtArgs {
  uint32_t opts;
  int64_t n_nodes;
  int64_t n_pods;
  int64_t row_stride;
  int64_t row_begin;
  int64_t row_end;
  const int64_t* row_ptr;  // when set: evaluate the single row *row_ptr (sequential commit: the row counter lives on the device)
  // pod equivalence classes (float64 formulation only): when set, the sweep evaluates the rows listed here — one representative
  // per class of pods whose NRT records agree, ascending — instead of [row_begin, row_end); launch_rows_expand copies each
  // representative's rows to the rest of its class afterwards
  const int32_t* row_list;
  int64_t n_list;
  int32_t n_res;
  int32_t strategy;
  uint8_t slot_flags[SPX_NRT_MAX_RES];
  int64_t slot_weight[SPX_NRT_MAX_RES];
  // node columns, zone/resource-major so that lane = node loads coalesce
  const uint8_t* flags;          // [N]
  const int32_t* max_numa;       // [N]
  const uint8_t* n_zones;        // [N]
  const uint8_t* zone_id;        // [Z][N]
  const uint8_t* zone_present;   // [Z][N]
  const int64_t* zone_avail;     // [Z][n_res][N]
  const int32_t* zone_cost;      // [Z][Z][N]
  const float* min_avg;          // [Z][N]  (subset size k-1)
  const uint8_t* node_present;   // [N]
  // pod records (wave-uniform reads)
  const uint8_t* qos;
  const uint8_t* non_native;
  const uint8_t* n_ctr;
  const uint8_t* ctr_kind;       // [P][8]
  const uint8_t* ctr_present;    // [P][8]
  const int64_t* ctr_req;        // [P][8][n_res]
  const uint8_t* pod_present;    // [P]
  const int64_t* pod_req;        // [P][n_res]
  uint8_t* out_status;           // [P][row_stride]
  uint8_t* out_score;            // [P][row_stride]
  int64_t* out_raw;              // when set: raw int64 scores of row_begin only, no table writes
  // float64 formulation (kernels_nrt_fast.hip); `fast` is set only when the engine verified its preconditions
  int32_t fast;
  int32_t cpu_slot;              // slot whose quantities are millicores (-1: none)
  double slot_weight_f[SPX_NRT_MAX_RES];
  const double* f_av;            // [Z][n_res][N] reported ? available : -1
  const double* f_rc;            // [Z][n_res][N] RN(100 / Value(capacity)), kNrtNoCap when the capacity is not positive
  const double* f_rcv;           // [Z][n_res][N] RN(1 / Value(capacity)) (BalancedAllocation's divisions), 1 when the capacity is not positive
  const double* f_cpu;           // [Z][N] Value() of the cpu capacity
  const double* f_braw;          // [Z][N] RN(100 / cpu capacity in millicores), kNrtNoCap when it is not positive
  const uint8_t* f_rep;          // [n_res][N] mask of the zones reporting the resource
  const uint32_t* ln_const;      // LeastNUMANodes: the block-constant tables (kLnConstWords dwords), behind ln_tab's rows
  const uint32_t* ln_tab;        // [LnLayout.rows][N] LeastNUMANodes: minimum-distance subsets + distance-rank planes; NULL when a
                                 // cost lies outside [0, 255] (the reference-arithmetic kernel then serves that strategy)
  const int32_t* perm;           // [ceil(N/256)*256] node index per slot, windows of 256 ordered by code path; -1 = empty
  const uint32_t* pod_items;     // [P][10][16 or 32] pod record stream (layout: spx_engine.hip nrt_pod_items)
  unsigned long long* stats;     // as TrimaranArgs::stats: BalancedAllocation cells recomputed in float64 (SPX_PLUGIN_NRT)
  uint32_t* redo_list;           // BalancedAllocation: [count, pad, (row - row_begin, node) x redo_cap] cells the float32 Score launch left to float64
  uint32_t redo_cap;
  // LeastNUMANodes (batch Score launch) uses the same buffer as node lists per (row, scope): [overflow flag, pad, count x 2 ln_rows,
  // node x 2 ln_rows x ln_per_row] — the cells whose subset search needs more than sizes 1-2 (k_nrt_ln_redo); ln_rows = 0: not in
  // use.  List 2 j + s: row slot j = position in row_list, or row - row_begin; s = 1 for the pod-scope nodes
  int64_t ln_rows;
  uint32_t ln_per_row;
  uint32_t* ln_rec;              // [N][kZ * RM * 2 + 16] scratch: the nodes' tables as one record each (k_nrt_ln_pack -> k_nrt_ln_redo)
  // rank-space Filter (kernels_nrt_rank.hip): the chunk stream of the listed rows; NULL = the float64 Filter launch
  const uint32_t* rk_stream;
  const uint32_t* rk_off;        // [chunks + 1] dword offsets of the chunk blocks
  uint32_t rk_max_dwords;        // largest chunk block (dynamic LDS)
  const uint32_t* rk_first;      // [rk_chunks + 1] position in the row list of each chunk's first row (a chunk holds up to 32 rows)
  uint32_t rk_chunks;
  bool rk_all_narrow;            // every chunk keeps four zones' counts per register: the fused walk (kernels_nrt_fused.hip) applies
  uint32_t exact32_slots;        // resource slots whose requests and capacities (Value() form) are all float32 values (below 2^24, or a multiple of a large power of two): compared exactly
  // LeastAllocated's Score-only launch in packed float32 (nrt_fast_device.h, score_least_packed): zone PAIRS per instruction, u16 zone totals
  uint32_t pk_mode;              // 0 = off (every other kernel / strategy / launch form)
  int32_t pk_tab_slot;           // the slot (memory in bytes) whose requests k_nrt_pk_tab_build replays; -1: every weighted slot is "small"
  uint32_t* pk_tab;              // [pk_tab_kmax + 1][pk_tab_words] by request / unit: bit w of the row = for some zone of node window w the packed form differs from the division
  uint32_t pk_tab_words;         // dwords per request value: ceil(node windows / 32)
  uint32_t pk_tab_kmax;          // largest request / unit of the pod batch
  double pk_tab_inv_unit;        // 1 / unit; unit = the largest power of two dividing every request of the slot
  bool* pk_tab_built;            // host flag: pk_tab describes the zone capacities and the unit in place (cleared by every writer of either)
  // the fused Filter + Score sweep (kernels_nrt_fused.hip): scratch for the packed Score items of the listed rows
  // (nrt_fused_item_words dwords); NULL = the Filter and Score launches
  uint32_t* fz_items;
  // the fused walk's block start for <= 4 slots: per (window, slot) the window's 2 048 cell quantities sorted, per node the cells' ranks in
  // them (k_nrt_window_sort; NULL = the per-cell list search)
  const double* wsort;
  const uint16_t* wrank;
  bool fz_pack;                  // the items must be (re)packed before the sweep: pods, slot table, row list or table slot changed
};
constexpr int64_t kNrtPkTabMaxK = (int64_t{1} << 17) - 1;   // request / unit above this: the float64 form
constexpr size_t kNrtPkTabMaxBytes = size_t{64} << 20;
constexpr int64_t kNrtPkMaxWeightSum = 320;  // 100 * sum(weights) must stay below 2^15 for the u16 zone totals
constexpr double kNrtNoCap = 1e200;
constexpr size_t kRkMaxChunkBytes = 56 * 1024;
bool launch_nrt_filter_rank(const NrtArgs& a, int n_tiles, hipStream_t s);
size_t nrt_fused_item_words(int n_res, int64_t n_list);
bool launch_nrt_fused(const NrtArgs& a, hipStream_t s);
bool launch_nrt_filter_fused(const NrtArgs& a, hipStream_t s);  // the same walk, Filter only
size_t nrt_window_sort_bytes(int64_t n_nodes, size_t* rank_bytes);
void launch_nrt_window_sort(const NrtArgs& a, double* wsort, uint16_t* wrank, hipStream_t s);
void launch_nrt_pk_tab_build(const NrtArgs& a, int n_tiles, hipStream_t s);  // kernels_nrt_fast.hip: the packed Score's table of exceptions

// combin.Combinations(8, k) for k = 1..8 as bitmasks over list positions, size-major then lexicographic — the order
// least_numa.go:167-208 walks.  Subsets of a node with fewer zones are the entries without high positions, in the same
// relative order.  Shared by the engine (per-node distance table) and the float64 LeastNUMANodes kernel.
struct Combo8 {
  uint8_t mask[256];
  uint8_t start[10];  // start[k-1] .. start[k]: subsets of size k
};
constexpr Combo8 make_combo8() {
  Combo8 t{};
  int idx = 0;
  for (int k = 1; k <= 8; ++k) {
    t.start[k - 1] = static_cast<uint8_t>(idx);
    int c[8] = {};
    for (int i = 0; i < k; ++i) c[i] = i;
    while (true) {
      int m = 0;
      for (int i = 0; i < k; ++i) m |= 1 << c[i];
      t.mask[idx++] = static_cast<uint8_t>(m);
      int i = k - 1;
      while (i >= 0 && c[i] == 8 - k + i) --i;
      if (i < 0) break;
      ++c[i];
      for (int j = i + 1; j < k; ++j) c[j] = c[j - 1] + 1;
    }
  }
  t.start[8] = static_cast<uint8_t>(idx);  // 255
  return t;
}
// Bit layout of "which zone subsets hold the request" in the float64 LeastNUMANodes search (kernels_nrt_fast.hip): subsets of
// size k occupy their own dwords — first[k] .. first[k] + nd[k] — in lexicographic order from bit 0 up, so "the smallest
// size with a fitting subset" is the first non-zero class and "the first in the reference's walk" the lowest set bit.
// Per node the engine stores, in the same layout, the subsets whose average distance is the node's minimum for their size
// (rows 0..11 of NrtArgs.ln_tab) and, per class, bit-planes of the RANK of each subset's distance among the node's distinct
// distances for that size (bits[k] planes of nd[k] dwords each, from row 12 + pbase[k]; plane b = bit b of the rank).
constexpr int kLnDwords = 12;
// what a LeastNUMANodes workgroup copies into LDS (NrtArgs::ln_const): [256][kLnDwords] dwords "zone set V -> the subsets inside V",
// then [kLnDwords][32] bytes "bit position -> zone mask"
constexpr int kLnConstWords = 256 * kLnDwords + kLnDwords * 32 / 4;
struct LnLayout {
  uint8_t subset[kLnDwords][32];  // zone mask of bit q of dword d; 0 = unused
  uint8_t cnt[kLnDwords];         // used bits of the dword
  uint8_t first[9], nd[9], bits[9], pbase[9];  // by subset size 1..8
  int rows;                       // 12 + plane rows
};
constexpr LnLayout make_ln_layout() {
  LnLayout l{};
  const Combo8 c = make_combo8();
  int d = 0, prow = 0;
  for (int k = 1; k <= 8; ++k) {
    const int n = c.start[k] - c.start[k - 1];
    l.first[k] = static_cast<uint8_t>(d);
    l.nd[k] = static_cast<uint8_t>((n + 31) / 32);
    int b = 0;
    while ((1 << b) < n) ++b;
    l.bits[k] = static_cast<uint8_t>(b);
    l.pbase[k] = static_cast<uint8_t>(prow);
    prow += b * l.nd[k];
    for (int p = 0; p < n; ++p) {
      l.subset[d + p / 32][p % 32] = c.mask[c.start[k - 1] + p];
      l.cnt[d + p / 32] = static_cast<uint8_t>(p % 32 + 1);
    }
    d += l.nd[k];
  }
  l.rows = kLnDwords + prow;
  return l;
}
bool launch_nrt(const NrtArgs& a, hipStream_t s);  // true = ran as the fused Filter + Score launch
// the reference-arithmetic kernel's per-container request column [P][8][n_res] int64, rebuilt from the pod record stream (whose
// quantities are exact doubles whenever the stream is valid): the engine does not ship that column — 256 bytes per pod — with every
// pod batch, only when a launch is going to read it
void launch_nrt_creq_from_items(const uint32_t* pod_items, int n_res, int64_t n_pods, int64_t* ctr_req, hipStream_t s);
// returns false when the float64 kernel does not apply (preconditions, LeastNUMANodes)
bool launch_nrt_fast(const NrtArgs& a, hipStream_t s);

// ---------------------------------------------------------------- snapshot deltas (kernels_delta.hip)
// dst column-major [inner][n_nodes] <- src row-major [n_rows][inner] at nodes idx[row]; elem_bytes 1, 4 or 8
void launch_scatter_rows(void* dst, int64_t n_nodes, int inner, const int32_t* idx, const void* src, int64_t n_rows, int elem_bytes, hipStream_t s);
void launch_scatter_rows_rowmajor(void* dst, int inner, const int32_t* idx, const void* src, int64_t n_rows, int elem_bytes, hipStream_t s);
void launch_net_append(int64_t n, const int32_t* pos, const int32_t* node, const int64_t* cost, int32_t* dst_node, int64_t* dst_max, hipStream_t s);
struct NrtDeltaArgs {
  int64_t n_rows, n_nodes;
  int32_t n_res, cpu_slot;
  const int32_t* idx;           // [n_rows] node of each row
  const uint8_t* n_zones;       // staged rows, as spx_nrt_nodes_soa holds them
  const uint8_t* zone_present;  // [n_rows][Z]
  const int64_t* zone_avail;    // [n_rows][Z][n_res]
  double* f_av;                 // the float64 formulation's derived columns (NrtArgs)
  double* f_rc;
  double* f_rcv;
  double* f_cpu;
  double* f_braw;
  uint8_t* f_rep;
};
void launch_nrt_derive_rows(const NrtDeltaArgs& a, hipStream_t s);

// ---------------------------------------------------------------- NetworkOverhead
struct NetArgs {
  uint32_t opts;
  int64_t n_nodes;
  int64_t row_stride;
  int64_t row_begin;
  int64_t row_end;
  const int64_t* row_ptr;  // when set: evaluate the single row *row_ptr (sequential commit: the row counter lives on the device)
  int32_t n_regions;
  int32_t n_zones;
  int32_t n_classes;            // 0 = no class table: every node takes the exact per-pair path
  const int32_t* region;        // [N] interned topology.kubernetes.io/region label, -1 = unset
  const int32_t* zone;          // [N]
  const int32_t* node_class;    // [N] index into cls_*
  const int32_t* cls_region;    // [n_classes]
  const int32_t* cls_zone;      // [n_classes]
  const int32_t* region_cost;   // [n_regions^2], -1 = no entry
  const int32_t* zone_cost;     // [n_zones^2]
  const uint16_t* node_class16; // [round_up(N, 4)] the same class ids, 16 bit (table sweep); NULL when they do not fit
  const int32_t* cls_size;      // [n_classes] nodes per class
  const int32_t* pod_key;       // [P]
  const uint8_t* key_flag;      // [K] 0 evaluate, 1 scoreEqually, 2 PreFilter error
  const int32_t* pair_ptr;      // [K+1]
  const int32_t* pair_end;      // [K] end of key k's list when the lists have slack and grow (sequential commit); NULL = pair_ptr[k+1]
  const int32_t* pair_node;
  const int64_t* pair_max;
  const uint8_t* other_status[2];  // other Filter plugins' status tables [P][row_stride] (0 = passed), NULL = unused
  uint8_t* out_status;
  uint8_t* out_score;
  int64_t* out_raw;             // when set: raw row (row_begin only), no table writes
  int32_t raw_which;
  // NodeResourcesAllocatable's feasibility-aware NormalizeScore carried by k_net_cls's two walks (batch launches; the engine sets
  // these when the evaluation would otherwise run k_alloc_masked's compact path over the same status tables)
  const uint32_t* alloc_rel;    // [row_stride + 1] raw scores as offsets from the global minimum (AllocPrepArgs.rel)
  uint8_t* out_alloc;           // Allocatable's score table; NULL = not fused
};
// true = the launch also wrote NetArgs::out_alloc (k_net_cls ran)
bool launch_net(const NetArgs& g, hipStream_t s);
size_t net_lds_bytes(int n_classes, int64_t n_nodes);

// ---------------------------------------------------------------- TopologicalSort (kernels_sort.hip)
struct SortArgs {
  int64_t n;
  const int32_t* priority;    // [P] pod.Spec.Priority
  const int64_t* queue_ts;    // [P] QueuedPodInfo.Timestamp
  const int32_t* appgroup;    // [P] AppGroup id, -1 = none
  const int32_t* topo_order;  // [P] FindPodOrder index (-1 = not found)
  int32_t* run_of_pod;        // scratch, set by launch_sort_keys
};
size_t sort_scratch_bytes(int64_t n);
const int32_t* launch_sort_keys(const SortArgs& a, void* scratch, unsigned* hist_host, hipStream_t s, hipError_t* err);

// ---------------------------------------------------------------- CapacityScheduling.PreFilter
struct QuotaArgs {
  int64_t row_begin;
  int64_t row_end;
  const int64_t* row_ptr;  // when set: evaluate the single row *row_ptr (sequential commit: the row counter lives on the device)
  int32_t n_namespaces;
  const int32_t* pod_ns;
  const int32_t* pod_priority;
  const int64_t* pod_req;          // [P][8]
  const uint8_t* pod_req_present;
  const uint8_t* has_quota;        // [NS]
  const int64_t* used;             // [NS][8]
  const int64_t* max;              // [NS][8]
  const uint8_t* max_present;
  int64_t agg_used[SPX_QUOTA_SLOTS];
  uint32_t agg_used_present;
  int64_t agg_min[SPX_QUOTA_SLOTS];
  uint32_t agg_min_present;
  const int64_t* agg_used_dyn;     // when set: [8] aggregate used + [1] its presence bits, in device memory (advances with commits)
  const int64_t* other_nominated;  // [NS][8]
  const uint8_t* other_nominated_present;
  const int32_t* nom_ptr;          // [NS+1]
  const int32_t* nom_priority;
  const int64_t* nom_pending_index;
  const int64_t* nom_req;          // [n_nominated][8]
  const uint8_t* nom_req_present;
  uint8_t* out_status;             // [P]
};
void launch_quota(const QuotaArgs& a, hipStream_t s);

// ---------------------------------------------------------------- sequential commit with Filter plugins (kernels_commit.hip)
// Bookkeeping of ONE bound pod (row `pod`, node = best_node[pod]) on the engine's device tables: what the reference's Reserve /
// assume-time hooks do between two scheduling cycles.  NULL table groups are skipped.
struct CommitApplyArgs {
  int64_t pod;                    // row just decided
  int64_t* row_counter;           // when set: the row is *row_counter, and the kernel advances it (graph replay)
  int64_t n_nodes, n_pods;
  const int32_t* best_node;       // [P] (spx_eval_best layout)
  // trimaran: handler.go:131-139 feeding targetloadpacking.go:151-168
  int64_t* tlp_missing;           // [N]
  const int64_t* tlp_pod_milli;   // [P]
  // NRT: OverReserve.ReserveNodeResources -> resourceStore.UpdateNRT (cache/overreserve.go:170-186, store.go:315-356)
  int32_t nrt_n_res, nrt_cpu_slot;
  const uint8_t* nrt_flags;       // [N]
  const uint8_t* nrt_zone_present;// [Z][N] bit r: zone reports slot r
  int64_t* nrt_avail;             // [Z][R][N]
  double* f_av;                   // derived float64 tables of the fast sweep, same cell order
  double* f_rc;
  double* f_rcv;
  double* f_cpu;                  // [Z][N]
  double* f_braw;                 // [Z][N]
  const uint8_t* nrt_pod_present; // [P] bit r: the pod's effective request lists slot r
  const int64_t* nrt_pod_req;     // [P][R]
  // CapacityScheduling: Reserve -> addPodIfNotPresent -> reserveResource (capacity_scheduling.go:350-364, elasticquota.go:89-98)
  int32_t q_n_namespaces;
  const int32_t* q_pod_ns;
  const int64_t* q_pod_req;       // [P][8]
  const uint8_t* q_pod_reqp;
  const uint8_t* q_has;           // [NS]
  int64_t* q_used;                // [NS][8]
  uint8_t* q_used_present;        // [NS]
  const int64_t* q_min;           // [NS][8]
  const uint8_t* q_min_present;
  int64_t* q_agg_used;            // [8] + [1] presence
  const int32_t* q_nom_ptr;       // [NS+1]
  const int64_t* q_nom_pending;   // per nominated entry: pending row, -1 none
  int64_t* q_nom_req;             // [n][8]  (zeroed when the nominated pod itself is bound: it left the nominator)
  uint8_t* q_nom_reqp;
  int64_t* q_other;               // [NS][8] nominated requests of OTHER namespaces whose quota is not over min: recomputed
  uint8_t* q_otherp;
  // NetworkOverhead: the bound pod joins its AppGroup's scheduled list (util.GetScheduledList over the pod lister)
  const int32_t* net_eff_ptr;     // [P+1]
  const int32_t* net_eff_key;     // workload key that sees the new pod
  const int64_t* net_eff_cost;    // dependency MaxNetworkCost, -1 = the key only stops scoring equally
  uint8_t* net_key_flag;          // [K]
  int32_t* net_pair_end;          // [K]
  int32_t* net_pair_node;
  int64_t* net_pair_max;
};
void launch_commit_apply(const CommitApplyArgs& a, hipStream_t s);

// ---------------------------------------------------------------- sequential commit, cooperative persistent kernel (kernels_commit_coop.hip)
// One launch schedules a whole row range one pod at a time for profiles with Filter plugins.  A workgroup owns a window of 256 nodes
// and keeps their state in registers; per pod the workgroups exchange two sets of self-tagged 8-byte granules (feasible-set
// minima / maxima, then the weighted argmax) — the only memory they share.  What Reserve changes (NRT zones, trimaran's missing
// utilisation, quota usage, AppGroup placement) is applied by the owner lane or replayed identically in every workgroup.
constexpr int kCoopWindow = 256;    // nodes per workgroup (= the NRT node order's window)
constexpr int kCoopMaxWg = 256;     // one thread polls one workgroup's granules
constexpr int kCoopMaxPairs = 512;  // (host, MaxNetworkCost) pairs of one workload key staged in LDS
constexpr int kCoopMaxEffects = 32; // workload keys one bound pod changes
constexpr int kCoopMaxClasses = 512;
constexpr int kCoopKinds = 8;       // granules per workgroup and parity: 4 for the feasible set, 4 for the argmax
struct CoopArgs {
  uint32_t use;                     // profile: bit SPX_PLUGIN_*
  int32_t w[SPX_NUM_PLUGINS];       // plugin weights (the launcher checks the 32-bit totals)
  int64_t n_nodes, n_pods, row_stride, row_begin, row_end;
  int32_t n_wg;
  int32_t nrt_sg;                   // 0 LeastAllocated, 1 MostAllocated (nrtdev::kSg*)
  const uint32_t* alloc_rel;        // Allocatable: raw scores as offsets from the global minimum (AllocPrepArgs.rel)
  TrimaranArgs t;                   // node / pod columns of TargetLoadPacking
  const uint8_t* lv_table;          // LVRB's rows, swept beforehand (no commit state)
  NrtArgs nrt;                      // float64 formulation's tables
  NetArgs net;                      // labels, cost matrices, classes, pod_key; pair_ptr = start of each key's list WITH slack
  const int32_t* net_init_end;      // [K] end of the initial pairs in that layout
  const uint8_t* net_init_flag;     // [K]
  const int32_t* net_init_node;     // [cap] the initial pairs in the slack layout: every workgroup copies them into its private lists
  const int64_t* net_init_max;
  int64_t net_cap;
  int32_t net_n_keys;
  int32_t* net_priv_node;           // [n_wg][cap]
  int64_t* net_priv_max;            // [n_wg][cap]
  const int32_t* eff_ptr;           // commit effects per pod (spx_upload_net_commit)
  const int32_t* eff_key;
  const int64_t* eff_cost;
  // CapacityScheduling: inputs; the mutable part (used, aggregate, nominated requests, other-namespace sums) is copied into LDS
  int32_t q_ns, q_n_nom;
  const int32_t* q_pod_ns;
  const int32_t* q_pod_prio;
  const int64_t* q_pod_req;
  const uint8_t* q_pod_reqp;
  const uint8_t* q_has;
  const int64_t* q_used;
  const uint8_t* q_usedp;
  const int64_t* q_max;
  const uint8_t* q_maxp;
  const int64_t* q_min;
  const uint8_t* q_minp;
  const int64_t* q_agg;             // [8] aggregate used + [1] presence bits
  int64_t q_agg_min[SPX_QUOTA_SLOTS];
  uint32_t q_agg_min_present;
  const int64_t* q_other;
  const uint8_t* q_otherp;
  const int32_t* q_nom_ptr;
  const int32_t* q_nom_prio;
  const int64_t* q_nom_pending;
  const int64_t* q_nom_req;
  const uint8_t* q_nom_reqp;
  unsigned long long* sync;         // [2 parities][kCoopKinds][kCoopMaxWg] granules: value | tag << 32, zeroed before the launch
  int64_t* best_score;              // [n_pods] decisions, spx_eval_best's layout
  int32_t* best_node;
  int32_t* best_ties;
  int32_t* best_feasible;
  int64_t* missing_out;             // [N] trimaran's missing utilisation after the last commit (NULL: not wanted)
  int32_t* err;                     // set when a workgroup gave up waiting for another one (the launch is then void)
};
// dynamic LDS of one workgroup, 0 when the profile does not fit the kernel's staging areas (the caller then runs the per-pod loop)
size_t commit_coop_lds_bytes(const CoopArgs& c);
int commit_coop_max_resident(const CoopArgs& c, int device);
void launch_commit_coop(const CoopArgs& c, hipStream_t s);
// key k's pairs src[src_ptr[k] .. src_ptr[k+1]) -> dst[dst_ptr[k] ..): the workload pair lists re-laid with room to grow
void launch_spread_pairs(int32_t n_keys, const int32_t* src_ptr, const int32_t* dst_ptr, const int32_t* src_node, const int64_t* src_max, int32_t* dst_node,
                         int64_t* dst_max, hipStream_t s);

// ---------------------------------------------------------------- profile-level passes
struct ProfileArgs {
  int64_t n_nodes;
  int64_t row_stride;
  int64_t row_begin;
  int64_t row_end;
  const int64_t* row_ptr;  // when set: evaluate the single row *row_ptr (sequential commit: the row counter lives on the device)
  const uint8_t* status[3];                // filter status tables in play (0 = passed); NULL = unused
  const uint8_t* prefilter;                // [P] CapacityScheduling.PreFilter status, NULL = unused
  const int64_t* alloc_raw;                // [N] Allocatable raw scores
  const uint32_t* alloc_rel;               // [row_stride + 1] compact form (AllocPrepArgs.rel)
  uint8_t* out_alloc;
  const uint8_t* score[SPX_NUM_PLUGINS];   // score tables to sum (NULL = not in the profile)
  int64_t weight[SPX_NUM_PLUGINS];
  int32_t* best_node;                      // [P]
  int64_t* best_score;
  int32_t* best_ties;
  int32_t* best_feasible;
  int32_t block_per_row;  // set by the launchers: a whole workgroup per row in a batch launch too (rows too wide for kRowsPerBlock LDS shares)
};
void launch_alloc_masked(const ProfileArgs& a, hipStream_t s);
// copies row pairs[2i+1] to row pairs[2i] in up to two uint8 tables of row_stride bytes per row (NULL = skip); tasks[2k], tasks[2k+1] = first pair and
// number of pairs (<= kRowsExpandFan) of a run of pairs that share their source row (expand_tasks, spx_engine.h)
// BalancedAllocation: the cells a float32 Score launch listed in NrtArgs::redo_list, recomputed in float64 (k_nrt_bal_scan + k_nrt_bal_redo, kernels_nrt_fast.hip)
void launch_nrt_bal_fixups(const NrtArgs& a, hipStream_t s);
constexpr int kRowsExpandFan = 8;
void launch_rows_expand(const int32_t* pairs, const int32_t* tasks, int64_t n_tasks, uint8_t* t0, uint8_t* t1, int64_t row_stride, hipStream_t s);
void launch_best(const ProfileArgs& a, hipStream_t s);
// spx_decide with Filter plugins in the mask: Allocatable's feasibility-aware normalisation and the weighted argmax in one kernel
// (no Allocatable table).  decide_masked_ok: weights fit the 32-bit totals and the rows the 16-byte tiles; the caller also needs
// the compact Allocatable form (AllocPrepArgs.rel valid: raw scores span less than 2^32)
bool decide_masked_ok(const ProfileArgs& a);
void launch_decide_masked(const ProfileArgs& a, hipStream_t s);

}  // namespace spx
