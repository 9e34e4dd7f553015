// spx_multi.hip — one host process, several MI355X: the pods x nodes evaluation sharded by pod rows across the devices of
// one node, node tables replicated, no collective inside the evaluation; RCCL all-gather over xGMI afterwards to
// reassemble the per-pod decisions and, on request, a global score / feasibility table on every device.
//
// This is the shape BASELINE.json's north_star names: the (single) Go scheduler process binds this through cgo — it
// cannot join a torch.distributed job.  One spx_engine per device (include/spx.h), one persistent host thread per device so
// that the kernel launches of a step are issued concurrently (a single thread walking 8 devices would serialise 8 x the
// launch latency in front of a 0.4 ms sweep), HIP events per device for timing.
//
// RCCL is loaded with dlopen at spx_multi_create (librccl.so.1 is 570 MB; a scheduler that drives one GPU never maps it).
// The peer-copy transport moves the same bytes with hipMemcpyPeerAsync (SDMA over xGMI, or a plain device copy when two
// ranks share a device) — used when RCCL cannot be initialised and by the tests that run two ranks on a one-GPU box.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "spx_internal.h"

namespace {

thread_local std::string g_multi_create_error;

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;  // optional: only spx_multi_rccl_ranks asks

  bool load(std::string* why) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (so) break;
    }
    if (!so) {
      *why = std::string("dlopen(librccl.so.1): ") + dlerror();
      return false;
    }
    auto sym = [&](const char* n) { return dlsym(so, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
    if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) {
      *why = "librccl.so.1 lacks an expected symbol";
      return false;
    }
    return true;
  }
};

// one persistent host thread per rank; run_all() hands every thread the same closure (called with its rank) and waits
// until all of them have *issued* their work (the GPU work itself is asynchronous on each engine's stream)
class Workers {
 public:
  explicit Workers(int n) : n_(n), rc_(static_cast<size_t>(n), 0) {
    for (int r = 0; r < n; ++r) threads_.emplace_back([this, r] { loop(r); });
  }
  ~Workers() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  // returns the first non-zero code (by rank order), 0 when every rank succeeded
  int run_all(const std::function<int(int)>& fn, int* failed_rank = nullptr) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      pending_ = n_;
      ++gen_;
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
    for (int r = 0; r < n_; ++r)
      if (rc_[static_cast<size_t>(r)] != 0) {
        if (failed_rank) *failed_rank = r;
        return rc_[static_cast<size_t>(r)];
      }
    return 0;
  }

 private:
  void loop(int rank) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<int(int)>* fn;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        fn = fn_;
      }
      const int rc = (*fn)(rank);
      {
        std::lock_guard<std::mutex> lk(mu_);
        rc_[static_cast<size_t>(rank)] = rc;
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  int n_;
  std::vector<int> rc_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<int(int)>* fn_ = nullptr;
  int pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

struct GlobalTable {
  std::vector<void*> dptr;  // per rank: uint8 [size * rows_per][row_stride]
  int64_t rows_per = 0, row_stride = 0, n_pods_total = 0;
  size_t bytes = 0;         // of each rank's allocation
  bool gathered = false;
};

}  // namespace

struct spx_multi {
  int n = 0;
  int transport = SPX_MULTI_TRANSPORT_RCCL;
  std::vector<int> device;
  std::vector<spx_engine*> engine;
  std::vector<ncclComm_t> comm;
  Rccl rccl;
  bool rccl_ready = false;
  Workers* workers = nullptr;
  mutable std::string err;
  // decisions: per rank a buffer of size * slot bytes; rank r's own block sits in slot r
  std::vector<void*> d_best;
  size_t best_slot = 0;  // bytes per rank slot
  void* h_best = nullptr;
  size_t h_best_bytes = 0;
  GlobalTable table[2][SPX_NUM_PLUGINS];  // [which: 0 score, 1 status][plugin]
  std::vector<hipEvent_t> g0, g1;         // around the last gather, per rank
  bool gather_timed = false;
  std::vector<hipEvent_t> mark[2];        // spx_multi_mark: caller-placed region marks, per rank
};

namespace {

int mfail(const spx_multi* m, int code, const std::string& msg) {
  if (m) m->err = msg;
  else g_multi_create_error = msg;
  return code;
}

#define SPXM_HIP(m, call)                                                                  \
  do {                                                                                     \
    hipError_t _st = (call);                                                               \
    if (_st != hipSuccess) return mfail((m), SPX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_st)); \
  } while (0)

int engine_failed(const spx_multi* m, int rank, int rc) {
  const char* t = spx_last_error(m->engine[static_cast<size_t>(rank)]);
  return mfail(m, rc, "rank " + std::to_string(rank) + " (device " + std::to_string(m->device[static_cast<size_t>(rank)]) + "): " + (t ? t : ""));
}

// every rank all-gathers `bytes` from buf[r] + r * bytes into buf[r] (in place), on its engine's stream
int all_gather_in_place(spx_multi* m, const std::vector<void*>& buf, size_t bytes) {
  if (bytes == 0) return SPX_OK;
  if (!m->gather_timed) {
    m->g0.resize(static_cast<size_t>(m->n));
    m->g1.resize(static_cast<size_t>(m->n));
    for (int r = 0; r < m->n; ++r) {
      SPXM_HIP(m, hipSetDevice(m->device[static_cast<size_t>(r)]));
      SPXM_HIP(m, hipEventCreate(&m->g0[static_cast<size_t>(r)]));
      SPXM_HIP(m, hipEventCreate(&m->g1[static_cast<size_t>(r)]));
    }
    m->gather_timed = true;
  }
  std::vector<spx::EngineView> v;
  for (int r = 0; r < m->n; ++r) v.push_back(spx::engine_view(m->engine[static_cast<size_t>(r)]));
  for (int r = 0; r < m->n; ++r) {
    SPXM_HIP(m, hipSetDevice(v[static_cast<size_t>(r)].device));
    SPXM_HIP(m, hipEventRecord(m->g0[static_cast<size_t>(r)], v[static_cast<size_t>(r)].stream));
  }
  if (m->transport == SPX_MULTI_TRANSPORT_RCCL) {
    ncclResult_t st = m->rccl.GroupStart();
    for (int r = 0; r < m->n && st == ncclSuccess; ++r) {
      char* base = static_cast<char*>(buf[static_cast<size_t>(r)]);
      st = m->rccl.AllGather(base + static_cast<size_t>(r) * bytes, base, bytes, ncclUint8, m->comm[static_cast<size_t>(r)],
                             v[static_cast<size_t>(r)].stream);
    }
    const ncclResult_t st2 = m->rccl.GroupEnd();
    if (st == ncclSuccess) st = st2;
    if (st != ncclSuccess) return mfail(m, SPX_ERR_HIP, std::string("ncclAllGather: ") + m->rccl.GetErrorString(st));
  } else {
    // pull model: rank r copies every peer's slot out of the peer's buffer.  The producer must have finished: an event per
    // source rank, waited on by every consumer stream.
    std::vector<hipEvent_t> ready(static_cast<size_t>(m->n));
    for (int s = 0; s < m->n; ++s) {
      SPXM_HIP(m, hipSetDevice(v[static_cast<size_t>(s)].device));
      SPXM_HIP(m, hipEventCreateWithFlags(&ready[static_cast<size_t>(s)], hipEventDisableTiming));
      SPXM_HIP(m, hipEventRecord(ready[static_cast<size_t>(s)], v[static_cast<size_t>(s)].stream));
    }
    for (int r = 0; r < m->n; ++r) {
      SPXM_HIP(m, hipSetDevice(v[static_cast<size_t>(r)].device));
      for (int s = 0; s < m->n; ++s) {
        if (s == r) continue;
        SPXM_HIP(m, hipStreamWaitEvent(v[static_cast<size_t>(r)].stream, ready[static_cast<size_t>(s)], 0));
        char* dst = static_cast<char*>(buf[static_cast<size_t>(r)]) + static_cast<size_t>(s) * bytes;
        const char* src = static_cast<const char*>(buf[static_cast<size_t>(s)]) + static_cast<size_t>(s) * bytes;
        SPXM_HIP(m, hipMemcpyPeerAsync(dst, v[static_cast<size_t>(r)].device, src, v[static_cast<size_t>(s)].device, bytes,
                                       v[static_cast<size_t>(r)].stream));
      }
    }
    for (int s = 0; s < m->n; ++s) (void)hipEventDestroy(ready[static_cast<size_t>(s)]);  // deferred by the runtime until complete
  }
  for (int r = 0; r < m->n; ++r) {
    SPXM_HIP(m, hipSetDevice(v[static_cast<size_t>(r)].device));
    SPXM_HIP(m, hipEventRecord(m->g1[static_cast<size_t>(r)], v[static_cast<size_t>(r)].stream));
  }
  if (m->transport != SPX_MULTI_TRANSPORT_RCCL) {
    // the pulls read the producers' own slots from the consumers' streams: a producer's next write into its slot (the next
    // gather's copy, or an eval into a bound global table) must come after every consumer's pulls — each stream waits on the
    // other ranks' end-of-gather events
    for (int r = 0; r < m->n; ++r) {
      SPXM_HIP(m, hipSetDevice(v[static_cast<size_t>(r)].device));
      for (int s = 0; s < m->n; ++s)
        if (s != r) SPXM_HIP(m, hipStreamWaitEvent(v[static_cast<size_t>(r)].stream, m->g1[static_cast<size_t>(s)], 0));
    }
  }
  return SPX_OK;
}

int64_t rows_per_rank(int64_t n_pods_total, int n) { return (n_pods_total + n - 1) / n; }

}  // namespace

extern "C" {

const char* spx_multi_last_error(const spx_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

int spx_multi_create(const int* device_ids, int n_devices, int transport, spx_multi** out) {
  if (!out) return mfail(nullptr, SPX_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!device_ids || n_devices <= 0 || n_devices > 64) return mfail(nullptr, SPX_ERR_ARG, "device list must hold 1..64 ids");
  if (transport != SPX_MULTI_TRANSPORT_RCCL && transport != SPX_MULTI_TRANSPORT_PEER_COPY)
    return mfail(nullptr, SPX_ERR_ARG, "transport: SPX_MULTI_TRANSPORT_RCCL or SPX_MULTI_TRANSPORT_PEER_COPY");
  spx_multi* m = new spx_multi();
  m->n = n_devices;
  m->transport = transport;
  m->device.assign(device_ids, device_ids + n_devices);
  for (int r = 0; r < n_devices; ++r) {
    spx_engine* e = nullptr;
    const int rc = spx_create(device_ids[r], &e);
    if (rc != SPX_OK) {
      const std::string msg = std::string("rank ") + std::to_string(r) + ": " + spx_last_error(nullptr);
      for (spx_engine* p : m->engine) spx_destroy(p);
      delete m;
      return mfail(nullptr, rc, msg);
    }
    m->engine.push_back(e);
  }
  if (transport == SPX_MULTI_TRANSPORT_RCCL) {
    bool distinct = true;
    for (int a = 0; a < n_devices; ++a)
      for (int b = a + 1; b < n_devices; ++b) distinct &= device_ids[a] != device_ids[b];
    std::string why;
    if (!distinct) why = "RCCL needs distinct devices (use SPX_MULTI_TRANSPORT_PEER_COPY to run several ranks on one device)";
    else if (m->rccl.load(&why)) {
      m->comm.resize(static_cast<size_t>(n_devices));
      const ncclResult_t st = m->rccl.CommInitAll(m->comm.data(), n_devices, device_ids);
      if (st == ncclSuccess) m->rccl_ready = true;
      else why = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(st);
    }
    if (!m->rccl_ready) {
      for (spx_engine* p : m->engine) spx_destroy(p);
      delete m;
      return mfail(nullptr, SPX_ERR_HIP, why);
    }
  } else {
    for (int a = 0; a < n_devices; ++a)
      for (int b = 0; b < n_devices; ++b) {
        if (device_ids[a] == device_ids[b]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) == hipSuccess && can) {
          (void)hipSetDevice(device_ids[a]);
          const hipError_t st = hipDeviceEnablePeerAccess(device_ids[b], 0);
          if (st != hipSuccess && st != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // staged copies still work
          else (void)hipGetLastError();
        }
      }
  }
  m->workers = new Workers(n_devices);
  *out = m;
  return SPX_OK;
}

int spx_multi_destroy(spx_multi* m) {
  if (!m) return SPX_OK;
  delete m->workers;
  for (int r = 0; r < m->n; ++r) {
    (void)hipSetDevice(m->device[static_cast<size_t>(r)]);
    (void)hipDeviceSynchronize();
    for (int w = 0; w < 2; ++w)
      for (int p = 0; p < SPX_NUM_PLUGINS; ++p) {
        GlobalTable& t = m->table[w][p];
        if (!t.dptr.empty() && t.dptr[static_cast<size_t>(r)]) {
          if (w == 0) (void)spx_bind_score_table(m->engine[static_cast<size_t>(r)], p, nullptr, 0, 0);
          else (void)spx_bind_status_table(m->engine[static_cast<size_t>(r)], p, nullptr, 0, 0);
          (void)hipFree(t.dptr[static_cast<size_t>(r)]);
        }
      }
    if (!m->d_best.empty() && m->d_best[static_cast<size_t>(r)]) (void)hipFree(m->d_best[static_cast<size_t>(r)]);
    if (m->gather_timed) {
      (void)hipEventDestroy(m->g0[static_cast<size_t>(r)]);
      (void)hipEventDestroy(m->g1[static_cast<size_t>(r)]);
    }
    for (auto& ev : m->mark)
      if (!ev.empty()) (void)hipEventDestroy(ev[static_cast<size_t>(r)]);
  }
  if (m->rccl_ready)
    for (ncclComm_t c : m->comm) (void)m->rccl.CommDestroy(c);
  for (spx_engine* e : m->engine) (void)spx_destroy(e);
  if (m->h_best) (void)hipHostFree(m->h_best);
  delete m;
  return SPX_OK;
}

int spx_multi_size(const spx_multi* m) { return m ? m->n : SPX_ERR_ARG; }

int spx_multi_engine(spx_multi* m, int rank, spx_engine** out) {
  if (!m || !out) return SPX_ERR_ARG;
  if (rank < 0 || rank >= m->n) return mfail(m, SPX_ERR_ARG, "rank out of range");
  *out = m->engine[static_cast<size_t>(rank)];
  return SPX_OK;
}

int spx_multi_rccl_ranks(const spx_multi* m) {
  if (!m) return SPX_ERR_ARG;
  if (!m->rccl_ready || m->comm.empty()) return 0;  // peer-copy transport
  int n = static_cast<int>(m->comm.size());
  if (m->rccl.CommCount && m->rccl.CommCount(m->comm[0], &n) != ncclSuccess) return SPX_ERR_HIP;
  return n;
}

int spx_multi_shard(const spx_multi* m, int64_t n_pods_total, int rank, int64_t* row_begin, int64_t* row_end) {
  if (!m || !row_begin || !row_end) return SPX_ERR_ARG;
  if (rank < 0 || rank >= m->n || n_pods_total < 0) return mfail(m, SPX_ERR_ARG, "rank / pod count out of range");
  const int64_t per = rows_per_rank(n_pods_total, m->n);
  *row_begin = std::min<int64_t>(n_pods_total, per * rank);
  *row_end = std::min<int64_t>(n_pods_total, per * (rank + 1));
  return SPX_OK;
}

int spx_multi_eval(spx_multi* m, uint32_t plugin_mask) {
  if (!m) return SPX_ERR_ARG;
  int bad = -1;
  const int rc = m->workers->run_all([&](int r) {
    spx_engine* e = m->engine[static_cast<size_t>(r)];
    const spx::EngineView v = spx::engine_view(e);
    return v.n_pods > 0 ? spx_eval(e, plugin_mask, 0, v.n_pods) : SPX_OK;  // a rank whose shard is empty has nothing to do
  }, &bad);
  return rc ? engine_failed(m, bad, rc) : SPX_OK;
}

int spx_multi_eval_best(spx_multi* m, uint32_t plugin_mask) {
  if (!m) return SPX_ERR_ARG;
  int bad = -1;
  const int rc = m->workers->run_all([&](int r) {
    spx_engine* e = m->engine[static_cast<size_t>(r)];
    const spx::EngineView v = spx::engine_view(e);
    return v.n_pods > 0 ? spx_eval_best(e, plugin_mask, 0, v.n_pods) : SPX_OK;
  }, &bad);
  return rc ? engine_failed(m, bad, rc) : SPX_OK;
}

int spx_multi_decide(spx_multi* m, uint32_t plugin_mask) {
  if (!m) return SPX_ERR_ARG;
  int bad = -1;
  const int rc = m->workers->run_all([&](int r) {
    spx_engine* e = m->engine[static_cast<size_t>(r)];
    const spx::EngineView v = spx::engine_view(e);
    return v.n_pods > 0 ? spx_decide(e, plugin_mask, 0, v.n_pods) : SPX_OK;
  }, &bad);
  return rc ? engine_failed(m, bad, rc) : SPX_OK;
}

int spx_multi_sync(spx_multi* m) {
  if (!m) return SPX_ERR_ARG;
  for (int r = 0; r < m->n; ++r) {
    const int rc = spx_sync(m->engine[static_cast<size_t>(r)]);
    if (rc) return engine_failed(m, r, rc);
  }
  return SPX_OK;
}

int spx_multi_gather_best(spx_multi* m, int64_t n_pods_total, int32_t* node_idx, int64_t* weighted_score, int32_t* n_ties, int32_t* n_feasible) {
  if (!m || !node_idx || !weighted_score) return SPX_ERR_ARG;
  const int64_t per = rows_per_rank(n_pods_total, m->n);
  std::vector<spx::EngineView> v;
  for (int r = 0; r < m->n; ++r) {
    v.push_back(spx::engine_view(m->engine[static_cast<size_t>(r)]));
    int64_t b, en;
    spx_multi_shard(m, n_pods_total, r, &b, &en);
    const int64_t local = v.back().n_pods > 0 ? v.back().n_pods : 0;
    if (local != en - b) return mfail(m, SPX_ERR_STATE, "rank " + std::to_string(r) + " holds " + std::to_string(local) + " pod rows, its shard of the batch has " + std::to_string(en - b) + " (spx_multi_shard)");
    if (local > 0 && !v.back().best_valid) return mfail(m, SPX_ERR_STATE, "rank " + std::to_string(r) + ": no decisions (spx_multi_eval_best / spx_multi_decide first)");
  }
  const size_t slot = spx::round_up(per * 20, 256);
  if (m->d_best.empty() || m->best_slot != slot) {
    m->d_best.resize(static_cast<size_t>(m->n), nullptr);
    for (int r = 0; r < m->n; ++r) {
      SPXM_HIP(m, hipSetDevice(v[static_cast<size_t>(r)].device));
      if (m->d_best[static_cast<size_t>(r)]) SPXM_HIP(m, hipFree(m->d_best[static_cast<size_t>(r)]));
      m->d_best[static_cast<size_t>(r)] = nullptr;
      SPXM_HIP(m, hipMalloc(&m->d_best[static_cast<size_t>(r)], slot * static_cast<size_t>(m->n)));
    }
    m->best_slot = slot;
  }
  // own block -> own slot (the engine's block is laid out for its local pod count: 20 bytes per pod, arrays back to back)
  for (int r = 0; r < m->n; ++r) {
    if (v[static_cast<size_t>(r)].n_pods <= 0) continue;
    SPXM_HIP(m, hipSetDevice(v[static_cast<size_t>(r)].device));
    SPXM_HIP(m, hipMemcpyAsync(static_cast<char*>(m->d_best[static_cast<size_t>(r)]) + static_cast<size_t>(r) * slot, v[static_cast<size_t>(r)].best,
                               static_cast<size_t>(v[static_cast<size_t>(r)].n_pods) * 20, hipMemcpyDeviceToDevice, v[static_cast<size_t>(r)].stream));
  }
  int rc = all_gather_in_place(m, m->d_best, slot);
  if (rc) return rc;
  // every device now holds all slots; the host reads rank 0's copy
  const size_t total = slot * static_cast<size_t>(m->n);
  if (m->h_best_bytes < total) {
    if (m->h_best) SPXM_HIP(m, hipHostFree(m->h_best));
    m->h_best = nullptr;
    m->h_best_bytes = 0;
    SPXM_HIP(m, hipHostMalloc(&m->h_best, total, hipHostMallocDefault));
    m->h_best_bytes = total;
  }
  SPXM_HIP(m, hipSetDevice(v[0].device));
  SPXM_HIP(m, hipMemcpyAsync(m->h_best, m->d_best[0], total, hipMemcpyDeviceToHost, v[0].stream));
  SPXM_HIP(m, hipStreamSynchronize(v[0].stream));
  for (int r = 0; r < m->n; ++r) {
    int64_t b, en;
    spx_multi_shard(m, n_pods_total, r, &b, &en);
    const size_t P = static_cast<size_t>(en - b);
    if (P == 0) continue;
    const char* base = static_cast<const char*>(m->h_best) + static_cast<size_t>(r) * slot;
    const int64_t* hs = reinterpret_cast<const int64_t*>(base);
    const int32_t* hn = reinterpret_cast<const int32_t*>(hs + P);
    std::memcpy(weighted_score + b, hs, P * 8);
    std::memcpy(node_idx + b, hn, P * 4);
    if (n_ties) std::memcpy(n_ties + b, hn + P, P * 4);
    if (n_feasible) std::memcpy(n_feasible + b, hn + 2 * P, P * 4);
  }
  return SPX_OK;
}

int spx_multi_bind_global_table(spx_multi* m, int plugin, int which, int64_t n_pods_total) {
  if (!m || plugin < 0 || plugin >= SPX_NUM_PLUGINS || (which != 0 && which != 1)) return SPX_ERR_ARG;
  if (n_pods_total <= 0) return mfail(m, SPX_ERR_ARG, "n_pods_total must be positive");
  GlobalTable& t = m->table[which][plugin];
  const int64_t per = rows_per_rank(n_pods_total, m->n);
  int64_t stride = 0;
  for (int r = 0; r < m->n; ++r) {
    const spx::EngineView v = spx::engine_view(m->engine[static_cast<size_t>(r)]);
    if (v.row_stride <= 0) return mfail(m, SPX_ERR_STATE, "rank " + std::to_string(r) + ": upload the node table first");
    if (stride && v.row_stride != stride) return mfail(m, SPX_ERR_STATE, "ranks disagree on the row stride (different node counts or SPX_OPT_ROW_ALIGN)");
    stride = v.row_stride;
    if (v.n_pods > per) return mfail(m, SPX_ERR_STATE, "rank " + std::to_string(r) + " holds more pod rows than a shard of this batch");
  }
  const size_t slab = static_cast<size_t>(per) * static_cast<size_t>(stride);
  const size_t bytes = slab * static_cast<size_t>(m->n);
  const bool reuse = !t.dptr.empty() && t.bytes == bytes;  // same slab size (the usual per-batch re-bind): keep the allocations
  auto bind = [&](int r, void* p) {
    spx_engine* e = m->engine[static_cast<size_t>(r)];
    return which == 0 ? spx_bind_score_table(e, plugin, p, p ? stride : 0, p ? per : 0) : spx_bind_status_table(e, plugin, p, p ? stride : 0, p ? per : 0);
  };
  // every rank: wait for work that may still write the old slab, unbind it, free it unless it is reused
  auto release = [&](GlobalTable& g, bool keep) {
    for (int r = 0; r < m->n && !g.dptr.empty(); ++r) {
      if (!g.dptr[static_cast<size_t>(r)]) continue;
      const spx::EngineView v = spx::engine_view(m->engine[static_cast<size_t>(r)]);
      (void)hipSetDevice(v.device);
      (void)hipStreamSynchronize(v.stream);
      (void)bind(r, nullptr);
      if (!keep) {
        (void)hipFree(g.dptr[static_cast<size_t>(r)]);
        g.dptr[static_cast<size_t>(r)] = nullptr;
      }
    }
    if (!keep) g = GlobalTable{};
  };
  release(t, reuse);
  if (!reuse) t.dptr.assign(static_cast<size_t>(m->n), nullptr);
  t.bytes = bytes;
  t.rows_per = per;
  t.row_stride = stride;
  t.n_pods_total = n_pods_total;
  t.gathered = false;
  for (int r = 0; r < m->n; ++r) {
    int rc = SPX_OK;
    hipError_t he = hipSetDevice(m->device[static_cast<size_t>(r)]);
    if (he == hipSuccess && !t.dptr[static_cast<size_t>(r)]) he = hipMalloc(&t.dptr[static_cast<size_t>(r)], bytes);
    if (he == hipSuccess) {
      char* mine = static_cast<char*>(t.dptr[static_cast<size_t>(r)]) + static_cast<size_t>(r) * slab;
      rc = bind(r, mine);
    }
    if (he != hipSuccess || rc) {  // roll back: no rank stays bound to a table that is not recorded as complete
      std::string why = he != hipSuccess ? std::string(hipGetErrorString(he)) : std::string();
      if (rc) (void)engine_failed(m, r, rc);
      const std::string kept = m->err;
      release(t, false);
      if (he != hipSuccess) return mfail(m, SPX_ERR_HIP, "bind_global_table, rank " + std::to_string(r) + ": " + why);
      m->err = kept;
      return rc;
    }
  }
  return SPX_OK;
}

int spx_multi_allgather_table(spx_multi* m, int plugin, int which) {
  if (!m || plugin < 0 || plugin >= SPX_NUM_PLUGINS || (which != 0 && which != 1)) return SPX_ERR_ARG;
  GlobalTable& t = m->table[which][plugin];
  if (t.dptr.empty()) return mfail(m, SPX_ERR_STATE, "no global table bound for this plugin (spx_multi_bind_global_table)");
  // a rank whose shard of the table was not written by its last evaluation (spx_multi_decide on a Filter profile folds Allocatable's
  // normalisation into the argmax and writes no Allocatable table; a delta marks every table stale) would contribute old bytes
  for (int r = 0; r < m->n; ++r) {
    const spx::EngineView v = spx::engine_view(m->engine[static_cast<size_t>(r)]);
    if (v.n_pods > 0 && !(v.evaluated & (1u << plugin)))
      return mfail(m, SPX_ERR_STATE, "rank " + std::to_string(r) + ": this plugin's table was not written by the last evaluation (spx_multi_eval it first)");
  }
  const int rc = all_gather_in_place(m, t.dptr, static_cast<size_t>(t.rows_per) * static_cast<size_t>(t.row_stride));
  if (rc) return rc;
  t.gathered = true;
  return SPX_OK;
}

int spx_multi_global_table(spx_multi* m, int plugin, int which, int rank, void** dptr, int64_t* row_stride, int64_t* n_rows) {
  if (!m || plugin < 0 || plugin >= SPX_NUM_PLUGINS || (which != 0 && which != 1) || rank < 0 || rank >= m->n) return SPX_ERR_ARG;
  GlobalTable& t = m->table[which][plugin];
  if (t.dptr.empty()) return mfail(m, SPX_ERR_STATE, "no global table bound for this plugin");
  if (dptr) *dptr = t.dptr[static_cast<size_t>(rank)];
  if (row_stride) *row_stride = t.row_stride;
  if (n_rows) *n_rows = t.n_pods_total;
  return SPX_OK;
}

int spx_multi_fetch_global_rows(spx_multi* m, int plugin, int which, int rank, int64_t row_begin, int64_t row_end, uint8_t* out, int64_t out_stride) {
  if (!m || !out || plugin < 0 || plugin >= SPX_NUM_PLUGINS || (which != 0 && which != 1) || rank < 0 || rank >= m->n) return SPX_ERR_ARG;
  GlobalTable& t = m->table[which][plugin];
  if (t.dptr.empty() || !t.gathered) return mfail(m, SPX_ERR_STATE, "global table not gathered (spx_multi_allgather_table)");
  if (row_begin < 0 || row_end > t.n_pods_total || row_begin > row_end) return mfail(m, SPX_ERR_ARG, "row range out of bounds");
  const spx::EngineView v = spx::engine_view(m->engine[static_cast<size_t>(rank)]);
  if (out_stride < v.n_nodes) return mfail(m, SPX_ERR_ARG, "out_stride is smaller than n_nodes");
  if (row_begin == row_end) return SPX_OK;
  SPXM_HIP(m, hipSetDevice(v.device));
  SPXM_HIP(m, hipStreamSynchronize(v.stream));
  // global row g lives at slot (g / rows_per), local row (g % rows_per): contiguous because every slot holds rows_per rows
  SPXM_HIP(m, hipMemcpy2D(out, static_cast<size_t>(out_stride), static_cast<const char*>(t.dptr[static_cast<size_t>(rank)]) + row_begin * t.row_stride,
                          static_cast<size_t>(t.row_stride), static_cast<size_t>(v.n_nodes), static_cast<size_t>(row_end - row_begin), hipMemcpyDeviceToHost));
  return SPX_OK;
}

int spx_multi_mark(spx_multi* m, int which) {
  if (!m || (which != 0 && which != 1)) return SPX_ERR_ARG;
  std::vector<hipEvent_t>& ev = m->mark[which];
  if (ev.empty()) {
    ev.resize(static_cast<size_t>(m->n));
    for (int r = 0; r < m->n; ++r) {
      SPXM_HIP(m, hipSetDevice(m->device[static_cast<size_t>(r)]));
      SPXM_HIP(m, hipEventCreate(&ev[static_cast<size_t>(r)]));
    }
  }
  for (int r = 0; r < m->n; ++r) {
    const spx::EngineView v = spx::engine_view(m->engine[static_cast<size_t>(r)]);
    SPXM_HIP(m, hipSetDevice(v.device));
    SPXM_HIP(m, hipEventRecord(ev[static_cast<size_t>(r)], v.stream));
  }
  return SPX_OK;
}

int spx_multi_marked_ms(spx_multi* m, float* ms_max, float* ms_per_rank) {
  if (!m || !ms_max) return SPX_ERR_ARG;
  if (m->mark[0].empty() || m->mark[1].empty()) return mfail(m, SPX_ERR_STATE, "spx_multi_mark(0) and spx_multi_mark(1) first");
  float mx = 0.0f;
  for (int r = 0; r < m->n; ++r) {
    float ms = 0.0f;
    SPXM_HIP(m, hipSetDevice(m->device[static_cast<size_t>(r)]));
    SPXM_HIP(m, hipEventSynchronize(m->mark[1][static_cast<size_t>(r)]));
    SPXM_HIP(m, hipEventElapsedTime(&ms, m->mark[0][static_cast<size_t>(r)], m->mark[1][static_cast<size_t>(r)]));
    if (ms_per_rank) ms_per_rank[r] = ms;
    mx = ms > mx ? ms : mx;
  }
  *ms_max = mx;
  return SPX_OK;
}

int spx_multi_last_ms(spx_multi* m, float* eval_ms, float* gather_ms) {
  if (!m) return SPX_ERR_ARG;
  float ev = 0.0f, ga = 0.0f;
  for (int r = 0; r < m->n; ++r) {
    const spx::EngineView v = spx::engine_view(m->engine[static_cast<size_t>(r)]);
    SPXM_HIP(m, hipSetDevice(v.device));
    if (eval_ms && v.timed) {
      float ms = 0.0f;
      SPXM_HIP(m, hipEventSynchronize(v.ev1));
      SPXM_HIP(m, hipEventElapsedTime(&ms, v.ev0, v.ev1));
      ev = ms > ev ? ms : ev;
    }
    if (gather_ms && m->gather_timed) {
      float ms = 0.0f;
      SPXM_HIP(m, hipEventSynchronize(m->g1[static_cast<size_t>(r)]));
      SPXM_HIP(m, hipEventElapsedTime(&ms, m->g0[static_cast<size_t>(r)], m->g1[static_cast<size_t>(r)]));
      ga = ms > ga ? ms : ga;
    }
  }
  if (eval_ms) *eval_ms = ev;
  if (gather_ms) *gather_ms = ga;
  return SPX_OK;
}

}  // extern "C"
