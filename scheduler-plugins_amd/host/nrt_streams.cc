// nrt_streams.cc — see nrt_streams.hpp
#include "nrt_streams.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <mutex>

#include "parallel.hpp"

namespace spx_host {

using spx::kRkChunkRows;
using spx::kRkOpCharge0;
using spx::kRkOpCharge1;
using spx::kRkOpMerge1;
using spx::kRkOpMerge3;
using spx::kRkPodHead;
using spx::kRkVectors;

namespace {
// The canonical view of a pod record, on which the pod equivalence classes are built (nrt_build_classes).  Two pods get the same
// NRT rows on every node when their records agree in everything the sweep reads, and a queue is full of such pods: replicas of
// one Deployment, and every pod whose verdict does not depend on quantities — a pod that is not filtered (BestEffort without
// non-native resources, filter.go:186-190) passes and scores 100 whatever it asks for; a non-Guaranteed pod scores 100
// (score.go:72-76) and its NUMA-affine requests suit any reporting zone (numaresources.go:137-142), so only their presence
// counts.  The view is never materialised: a header pair and, per live item (pod level + the n_ctr containers), IW dwords
// produced on the stack; everything past the last container is out of it (equal headers = equal n_ctr).
struct NrtCanon {
  size_t RMs, IW;
  struct Head {
    uint32_t w0, w1;
    size_t n_items;  // 0: a pod nothing but whose class is read
    bool guaranteed;
  };
  Head head_of(const uint32_t* w) const {
    const uint32_t qos = w[0] & 0xffu, n_ctr = (w[0] >> 16) & 0xffu;
    const bool non_native = ((w[0] >> 8) & 0xffu) != 0;
    if (qos == SPX_QOS_BESTEFFORT && !non_native) return Head{qos, 0u, 0, false};
    const bool g = qos == SPX_QOS_GUARANTEED;
    return Head{w[0], g ? w[1] : 0u, 1 + static_cast<size_t>(n_ctr), g};  // the mean over containers (w1) belongs to the Score
  }
  void item(const uint32_t* it, bool guaranteed, uint32_t* c) const {
    std::memcpy(c, it, IW * sizeof(uint32_t));
    if (guaranteed) return;
    const uint32_t sets = it[2 * RMs], fit = (sets >> 8) & 0xffu;
    for (size_t r = 0; r < RMs; ++r)
      if (!((fit >> r) & 1u)) c[2 * r] = c[2 * r + 1] = 0;  // only compared quantities matter
    c[2 * RMs] = sets & 0xffffff00u;                        // "requested" steers the Score only
    for (size_t k = 2 * RMs + 1; k < IW; ++k) c[k] = 0;      // weight sums, Value() of the cpu request
  }
  static uint64_t mix(uint64_t h, uint64_t v) {
    h ^= v;
    h *= 0xff51afd7ed558ccdull;
    return h ^ (h >> 29);
  }
  uint64_t hash(const uint32_t* w) const {
    uint32_t c[32];
    const Head hd = head_of(w);
    uint64_t h = mix(0x9e3779b97f4a7c15ull, (static_cast<uint64_t>(hd.w1) << 32) | hd.w0);
    for (size_t s = 1; s <= hd.n_items; ++s) {
      item(w + s * IW, hd.guaranteed, c);
      for (size_t k = 0; k < IW; k += 2) h = mix(h, (static_cast<uint64_t>(c[k + 1]) << 32) | c[k]);
    }
    return h;
  }
  bool equal(const uint32_t* wa, const uint32_t* wb) const {
    uint32_t ca[32], cb[32];
    const Head ha = head_of(wa), hb = head_of(wb);
    bool same = ha.w0 == hb.w0 && ha.w1 == hb.w1 && ha.n_items == hb.n_items;
    for (size_t s = 1; same && s <= ha.n_items; ++s) {
      item(wa + s * IW, ha.guaranteed, ca);
      item(wb + s * IW, hb.guaranteed, cb);
      same = std::memcmp(ca, cb, IW * sizeof(uint32_t)) == 0;
    }
    return same;
  }
};

}  // namespace

// The pod record stream of the float64 NRT formulation, built on the host (no device involved: spx_internal_nrt_pod_classes lets
// the CPU tests see it).
// nrt_pod_items: per pod 10 items of IW dwords (IW = 16 for <= 4 resource slots, else 32), RM = 4 or 8 slots:
//   item 0   header: w0 = qos | non_native << 8 | n_ctr << 16 | last app container << 24 (0xff: none),
//                    w1 = ceil(2^16 / n_ctr)
//   item 1   the pod-level effective request;  items 2..9  the containers, in order
//   request item: doubles raw[RM] (dwords 0..2RM-1); dword 2RM = requested slots | compared slots << 8 |
//                 "any reporting zone suits" slots << 16 | kind << 24; dword 2RM+1 = sum of the weights of the requested
//                 slots as an integer; then what only the Score reads: Value() of the
//                 cpu request (2RM+2), sum of the weights of the requested slots (2RM+4), its biased reciprocal (2RM+6)
// hash_out (optional): the hash of each record's canonical view, taken while the record is still in cache
void nrt_build_items(const spx_nrt_pods_soa* t, const uint8_t* slot_flags, int cpu_slot, const std::vector<double>& wtab, uint32_t* items,
                     bool* ok_out, uint32_t* big_out, uint64_t* hash_out, NrtQty* qty_out) {
  const size_t p = static_cast<size_t>(t->n_pods), R = static_cast<size_t>(t->n_res);
  constexpr size_t Cm = SPX_NRT_MAX_CTRS;
  const int RMs = R <= 4 ? 4 : 8;
  const size_t IW = R <= 4 ? 16 : 32;
  const uint32_t slot_mask = (1u << R) - 1u;
  std::atomic<bool> ok{wtab.size() == (static_cast<size_t>(2) << R)};
  std::atomic<uint32_t> big_pods{0};
  auto put_f64 = [](uint32_t* w, double v) { std::memcpy(w, &v, sizeof v); };
  const bool tab_ok = ok.load();
  const NrtCanon canon{static_cast<size_t>(RMs), IW};
  // bad / big: per calling thread, merged once per chunk (the shared flags would bounce between the cores otherwise)
  std::mutex qty_mu;
  auto fill = [&](uint32_t* w, uint32_t present, const int64_t* req, bool non_g, uint32_t kind, bool& bad, uint32_t& big, NrtQty& qty) {
    const uint32_t used = present & slot_mask;
    uint32_t fit = 0, always = 0;
    for (size_t r = 0; r < R; ++r) {
      if (!nrt_fast_qty(req[r])) bad = true;
      if (!nrt_exact_f32(static_cast<double>(nrt_value_of(static_cast<int>(r) == cpu_slot, req[r])))) big |= 1u << r;
      put_f64(w + 2 * r, static_cast<double>(req[r]));
      if (((used >> r) & 1u) && req[r] > 0) qty.add(static_cast<int>(r), nrt_value_of(static_cast<int>(r) == cpu_slot, req[r]));
      if (!((used >> r) & 1u) || req[r] == 0) continue;  // "ignoring zero-qty resource request" filter.go:103-106
      if (non_g && (slot_flags[r] & SPX_NRT_SLOT_AFFINE)) always |= 1u << r;
      else fit |= 1u << r;
    }
    const int64_t cpu_q = cpu_slot >= 0 ? req[cpu_slot] : 0;
    w[2 * RMs] = used | (fit << 8) | (always << 16) | (kind << 24);
    put_f64(w + 2 * RMs + 2, static_cast<double>(nrt_value_of(true, cpu_q)));
    if (tab_ok) {
      w[2 * RMs + 1] = static_cast<uint32_t>(wtab[2 * used]);  // the weight sum as an integer (< 2^20)
      put_f64(w + 2 * RMs + 4, wtab[2 * used]);
      put_f64(w + 2 * RMs + 6, wtab[2 * used + 1]);
    }
  };
  spx_host::parallel_rows(static_cast<int64_t>(p), [&](int64_t row0, int64_t row1) {
    bool bad = false;
    uint32_t big = 0;
    NrtQty qty;
    for (size_t i = static_cast<size_t>(row0); i < static_cast<size_t>(row1); ++i) {
      uint32_t* w = &items[i * 10 * IW];
      std::memset(w, 0, 10 * IW * sizeof(uint32_t));  // the record ends with the last container: zeros after it
      const bool non_g = t->qos[i] != SPX_QOS_GUARANTEED;
      const uint32_t n_ctr = t->n_ctr[i];
      uint32_t last_app = 0xffu;
      bool seen_app = false;
      for (size_t c = 0; c < Cm && c < n_ctr; ++c) {
        const uint32_t kind = t->ctr_kind[i * Cm + c];
        if (kind == SPX_CTR_APP) {
          last_app = static_cast<uint32_t>(c);
          seen_app = true;
        } else if (seen_app) {
          bad = true;  // the single-pass Filter needs init containers listed before app containers
        }
        fill(w + (2 + c) * IW, t->ctr_present[i * Cm + c], t->ctr_req + (i * Cm + c) * R, non_g, kind, bad, big, qty);
      }
      fill(w + IW, t->pod_present[i], t->pod_req + i * R, non_g, 0, bad, big, qty);
      w[0] = t->qos[i] | (static_cast<uint32_t>(t->non_native[i] != 0) << 8) | (n_ctr << 16) | (last_app << 24);
      w[1] = n_ctr ? (65536u + n_ctr - 1u) / n_ctr : 0u;
      if (hash_out) hash_out[i] = canon.hash(w);
    }
    if (bad) ok = false;
    if (big) big_pods.fetch_or(big, std::memory_order_relaxed);
    if (qty_out) {
      std::lock_guard<std::mutex> g(qty_mu);
      qty_out->merge(qty);
    }
  }, 4096);
  *ok_out = ok.load();
  *big_out = big_pods.load();
}

// Pod equivalence classes: rep[i] = the first row whose canonical record (NrtCanon) equals row i's (rep[i] == i: a
// representative).  hash[i] = NrtCanon::hash of row i (nrt_build_items); rows with equal hashes are verified word for word.
void nrt_build_classes(const uint32_t* items, const uint64_t* hash, size_t p, size_t R, int32_t* rep) {
  const NrtCanon canon{R <= 4 ? size_t{4} : size_t{8}, R <= 4 ? size_t{16} : size_t{32}};
  const size_t PW = 10 * canon.IW;
  // first row of each hash value: a flat open-addressing table, rows visited in order (serial: ~15 ns per row)
  {
    size_t cap = 64;
    while (cap < 2 * p) cap <<= 1;
    struct Slot {
      uint64_t h;
      int32_t row;
    };
    std::vector<Slot> tab(cap, Slot{0, -1});
    for (size_t i = 0; i < p; ++i) {
      size_t k = static_cast<size_t>(hash[i] >> 20) & (cap - 1);
      while (tab[k].row >= 0 && tab[k].h != hash[i]) k = (k + 1) & (cap - 1);
      if (tab[k].row < 0) tab[k] = Slot{hash[i], static_cast<int32_t>(i)};
      rep[i] = tab[k].row;
    }
  }
  spx_host::parallel_rows(static_cast<int64_t>(p), [&](int64_t row0, int64_t row1) {
    for (int64_t i = row0; i < row1; ++i) {
      const int32_t r0 = rep[static_cast<size_t>(i)];
      if (r0 != i && !canon.equal(items + static_cast<size_t>(i) * PW, items + static_cast<size_t>(r0) * PW))
        rep[static_cast<size_t>(i)] = static_cast<int32_t>(i);  // a hash collision: the row stands for itself
    }
  }, 2048);
}

// The rank-space Filter's input (kernels_nrt_rank.hip), built per chunk of 32 listed rows: what the chunk's pods ask for, as RANKS.
// For every resource slot the chunk's distinct compared quantities, sorted, behind a leading 0 ("any reporting zone"): a node's
// zone then needs one number per resource — how many of them its available quantity reaches — and "available >= request" becomes
// "that count >= the request's position + 1", an 11-bit integer comparison the kernel does with a subtract (two zones per dword).
// The container-scope handler charges an app container to the zone it chose before the next one is tested
// (filter.go:131-163 -> numaresources.go:145-182); instead of mutating the zone table, the later container is compared with the
// SUM of the requests a zone would have been charged — available - charged >= request  <=>  available >= charged + request, exact
// in integers — so the chunk's lists also hold those sums: per pod 13 comparison vectors (layout: kRk*, spx_internal.h):
// the pod-level request, the eight containers, and for the second / third app container the sums with the earlier app
// containers a zone may carry.  Pods with more than three app containers have no such finite list: *ok_out = false and the
// batch keeps the float64 Filter.  A chunk = up to 32 consecutive listed rows (first_out[c] .. first_out[c + 1]); chunk block: 16 header
// dwords (per slot: search steps | list offset << 8; [8] rows; [9] narrow),
// the lists (2^steps - 1 doubles each, padded with +inf), then per pod kRkPodHead + 13 x RM dwords.
void nrt_build_rank_stream(const uint32_t* items, const int32_t* list, size_t n_list, size_t R, std::vector<uint32_t>& words, std::vector<uint32_t>& off,
                           std::vector<uint32_t>& first_out, uint32_t* max_dwords_out, bool* ok_out, bool* all_narrow_out, bool narrow_ok) {
  const size_t RM = R <= 4 ? 4 : 8, IW = R <= 4 ? 16 : 32, PW = 10 * IW, PWR = kRkPodHead + kRkVectors * RM;
  const size_t n_groups = (n_list + kRkChunkRows - 1) / kRkChunkRows;
  struct Block {
    uint32_t first, rows;
    std::vector<uint32_t> w;
  };
  std::vector<std::vector<Block>> groups(n_groups);  // a group = 32 consecutive listed rows = one chunk, or the chunks it was split into
  std::atomic<bool> ok{true}, all_narrow{true};
  auto f64 = [](const uint32_t* w) { double v; std::memcpy(&v, w, sizeof v); return v; };
  spx_host::parallel_rows(static_cast<int64_t>(n_groups), [&](int64_t c0, int64_t c1) {
    std::vector<double> vals[SPX_NRT_MAX_RES];
    for (int64_t c = c0; c < c1; ++c) {
      const size_t first = static_cast<size_t>(c) * kRkChunkRows, rows = std::min<size_t>(kRkChunkRows, n_list - first);
      // pass 1: every pod's 13 vectors (value per slot, NaN = not compared)
      std::vector<double> vec(rows * kRkVectors * RM, std::numeric_limits<double>::quiet_NaN());
      std::vector<uint32_t> head(rows * kRkPodHead, 0u);
      std::vector<uint8_t> any_always(rows * kRkVectors, 0);  // per vector: the item's "any reporting zone suits" slots
      for (size_t i = 0; i < rows; ++i) {
        const uint32_t* w = items + static_cast<size_t>(list[first + i]) * PW;
        uint32_t* h = &head[i * kRkPodHead];
        h[0] = w[0], h[1] = w[1];
        const uint32_t n_ctr = (w[0] >> 16) & 0xffu;
        uint32_t app[3] = {0xffu, 0xffu, 0xffu}, n_app = 0;
        for (size_t k = 1; k <= 9; ++k) h[1 + k] = w[k * IW + 2 * RM];  // the items' slot sets (absent items are zero)
        for (uint32_t ctr = 0; ctr < n_ctr && ctr < SPX_NRT_MAX_CTRS; ++ctr)
          if ((h[3 + ctr] >> 24) == SPX_CTR_APP) {
            if (n_app < 3) app[n_app] = ctr;
            ++n_app;
          }
        if (n_app > 3) ok = false;
        h[11] = app[0] | (app[1] << 8) | (app[2] << 16) | (std::min<uint32_t>(n_app, 255u) << 24);
        auto fit_of = [&](size_t item) { return (w[item * IW + 2 * RM] >> 8) & 0xffu; };
        auto raw_of = [&](size_t item, size_t r) { return f64(w + item * IW + 2 * r); };
        double* v = &vec[i * kRkVectors * RM];
        auto put = [&](size_t vi, size_t item, std::initializer_list<uint32_t> charged) {
          const uint32_t fit = fit_of(item);
          any_always[i * kRkVectors + vi] = static_cast<uint8_t>((w[item * IW + 2 * RM] >> 16) & 0xffu);
          for (size_t r = 0; r < R; ++r) {
            if (!((fit >> r) & 1u)) continue;
            double q = raw_of(item, r);
            for (uint32_t j : charged)
              if ((fit_of(2 + j) >> r) & 1u) q += raw_of(2 + j, r);
            v[vi * RM + r] = q;
          }
        };
        put(0, 1, {});
        for (uint32_t ctr = 0; ctr < n_ctr && ctr < SPX_NRT_MAX_CTRS; ++ctr) put(1 + ctr, 2 + ctr, {});
        if (n_app >= 2 && n_app <= 3) put(9, 2 + app[1], {app[0]});
        if (n_app == 3) put(10, 2 + app[2], {app[0]}), put(11, 2 + app[2], {app[1]}), put(12, 2 + app[2], {app[0], app[1]});
        // per container one byte of what the fused sweep (kernels_nrt_fused.hip) does with it, so that its loop tests bits instead of
        // deriving them (h[12]: containers 0-3, h[13]: 4-7): bits 0-2 the status a misfit sets, kRkOp*
        const uint32_t last_app = w[0] >> 24;
        for (uint32_t ctr = 0; ctr < n_ctr && ctr < SPX_NRT_MAX_CTRS; ++ctr) {
          const uint32_t kind = h[3 + ctr] >> 24, fit = fit_of(2 + ctr);
          uint32_t op = kind == SPX_CTR_APP ? SPX_NRT_ST_CONTAINER : (kind == SPX_CTR_SIDECAR ? SPX_NRT_ST_SIDECAR_CONTAINER : SPX_NRT_ST_INIT_CONTAINER);
          if (kind == SPX_CTR_APP && fit != 0 && n_app <= 3) {
            if (ctr == app[1]) op |= kRkOpMerge1;
            else if (ctr == app[2]) op |= kRkOpMerge3;
            if (ctr != last_app) op |= ctr == app[0] ? kRkOpCharge0 : kRkOpCharge1;
          }
          h[12 + (ctr >> 2)] |= op << (8 * (ctr & 3));
        }
      }
      // pass 2: the chunk [lo, hi) of the group — its lists, then the thresholds.  A chunk whose lists all have at most 127 entries (leading 0
      // included) is "narrow": positions and counts fit 7 bits, the kernels pack four zones per dword (RkLayout<true>) and the thresholds are
      // replicated into four bytes instead of two halves.  With narrow_ok a chunk that is not is split in halves until it is (a single pod
      // compares at most 13 values per slot), so that every chunk of the stream is narrow — the fused sweep has no other layout.
      std::vector<Block>& out = groups[static_cast<size_t>(c)];
      auto emit = [&](auto&& self, size_t lo, size_t hi) -> void {
        for (size_t r = 0; r < R; ++r) {
          auto& a = vals[r];
          a.clear();
          a.push_back(0.0);
          for (size_t i = lo; i < hi; ++i)
            for (size_t vi = 0; vi < kRkVectors; ++vi) {
              const double q = vec[(i * kRkVectors + vi) * RM + r];
              if (q == q) a.push_back(q);
            }
          std::sort(a.begin(), a.end());
          a.erase(std::unique(a.begin(), a.end()), a.end());
        }
        bool narrow = true;
        for (size_t r = 0; r < R; ++r) narrow = narrow && vals[r].size() <= 127;
        if (narrow_ok && !narrow && hi - lo > 1) {
          const size_t mid = lo + (hi - lo) / 2;
          self(self, lo, mid);
          self(self, mid, hi);
          return;
        }
        narrow = narrow && narrow_ok;
        if (!narrow) all_narrow = false;
        uint32_t steps[SPX_NRT_MAX_RES] = {0}, loff[SPX_NRT_MAX_RES] = {0};
        size_t list_doubles = 0;
        for (size_t r = 0; r < R; ++r) {
          uint32_t k = 1;
          while ((size_t{1} << k) - 1 < vals[r].size()) ++k;
          steps[r] = k, loff[r] = static_cast<uint32_t>(list_doubles);
          list_doubles += (size_t{1} << k);  // 2^k - 1 entries and one pad: every list starts 16-byte aligned
        }
        out.emplace_back();
        Block& blk = out.back();
        blk.first = static_cast<uint32_t>(first + lo), blk.rows = static_cast<uint32_t>(hi - lo);
        std::vector<uint32_t>& b = blk.w;
        b.assign(16 + 2 * list_doubles + (hi - lo) * PWR, 0u);
        for (size_t r = 0; r < R; ++r) b[r] = steps[r] | (loff[r] << 8);
        b[8] = static_cast<uint32_t>(hi - lo);
        b[9] = narrow ? 1u : 0u;
        for (size_t r = 0; r < R; ++r) {
          double* dst = reinterpret_cast<double*>(&b[16]) + loff[r];
          const size_t n = size_t{1} << steps[r];
          for (size_t j = 0; j < n; ++j) dst[j] = j < vals[r].size() ? vals[r][j] : std::numeric_limits<double>::infinity();
        }
        for (size_t i = lo; i < hi; ++i) {
          uint32_t* dst = &b[16 + 2 * list_doubles + (i - lo) * PWR];
          std::memcpy(dst, &head[i * kRkPodHead], kRkPodHead * sizeof(uint32_t));
          for (size_t vi = 0; vi < kRkVectors; ++vi)
            for (size_t r = 0; r < R; ++r) {
              const double q = vec[(i * kRkVectors + vi) * RM + r];
              // a non-Guaranteed pod's NUMA-affine request: "count >= 1" (filter.go:120-129); k_nrt_filter_rank derives it from the slot
              // sets, the fused sweep reads it here; a slot the item does not compare keeps 0 ("count >= 0": every zone passes)
              if ((any_always[i * kRkVectors + vi] >> r) & 1u) dst[kRkPodHead + vi * RM + r] = narrow ? 0x01010101u : 0x00010001u;
              if (q != q) continue;
              const uint32_t t = static_cast<uint32_t>(std::lower_bound(vals[r].begin(), vals[r].end(), q) - vals[r].begin()) + 1u;
              dst[kRkPodHead + vi * RM + r] = narrow ? t * 0x01010101u : (t | (t << 16));
            }
        }
      };
      emit(emit, 0, rows);
    }
  }, 8);
  size_t n_chunks = 0;
  for (const auto& g : groups) n_chunks += g.size();
  off.assign(n_chunks + 1, 0u);
  first_out.assign(n_chunks + 1, static_cast<uint32_t>(n_list));
  std::vector<const Block*> flat;
  flat.reserve(n_chunks);
  for (const auto& g : groups)
    for (const Block& blk : g) flat.push_back(&blk);
  uint32_t max_dwords = 0;
  for (size_t c = 0; c < n_chunks; ++c) {
    off[c + 1] = off[c] + static_cast<uint32_t>((flat[c]->w.size() + 3) & ~size_t{3});
    first_out[c] = flat[c]->first;
    max_dwords = std::max<uint32_t>(max_dwords, off[c + 1] - off[c]);
  }
  words.assign(off[n_chunks], 0u);
  spx_host::parallel_rows(static_cast<int64_t>(n_chunks), [&](int64_t c0, int64_t c1) {
    for (int64_t c = c0; c < c1; ++c) std::memcpy(&words[off[static_cast<size_t>(c)]], flat[static_cast<size_t>(c)]->w.data(), flat[static_cast<size_t>(c)]->w.size() * sizeof(uint32_t));
  }, 64);
  *max_dwords_out = max_dwords;
  *ok_out = ok.load();
  *all_narrow_out = all_narrow.load();
}

}  // namespace spx_host
