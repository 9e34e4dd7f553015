// ingest.cc — wire format -> object tables: NodeResourceTopology, v1.Node, v1.Pod.
// Host-side product code: what the API server sends (JSON of topology.node.k8s.io/v1alpha2 NodeResourceTopology objects — a
// single object, a List with "items", or a bare array) decoded straight into the spx_nrt_objects columns the flattener
// consumes, without an intermediate object graph.  The schema is the reference's own CRD
// (manifests/crds/topology.node.k8s.io_noderesourcetopologies.yaml; examples manifests/noderesourcetopology/worker-node-A.yaml):
//   metadata.name, topologyPolicies[], attributes[]{name,value}, zones[]{name, type, resources[]{name, capacity,
//   allocatable, available}, costs[]{name, value}}.
// Field semantics follow the plugin's readers: TopologyPolicies[0] and the topologyManager* attributes as
// nodeconfig/topologymanager.go:78-162 interprets them, zones of type "Node" and their NUMA id from the "node-<id>" name as
// createNUMANodeList does (pluginhelpers.go:105-161), quantities as resource.Quantity (cpu in millicores = ceil(v*1000),
// everything else Value() = ceil(v)).
//
// Resource names are interned in first-seen order after the five fixed ids of spx.h; a caller that already interned names
// for its pod tables passes them in so that both sides share one id space.
//
// v1.Node and v1.Pod (core/v1 JSON as the API server serves it) fill spx_node_objects / spx_pod_objects the way the Go shim of
// INTEGRATION.md section 3 would: allocatable / capacity in canonical units, the scalar resources framework.Resource.Add keeps,
// the two topology labels; per pod the init containers first (restartPolicy Always = sidecar) then the app containers with
// their request / limit lists in document order, overhead, priority, namespace and the two AppGroup labels the network-aware
// plugins read (pkg/networkaware/util/util.go:67-75; key strings as in manifests/appgroup/deploy-onlineBoutique-*.yaml).
#include <cstdint>
#include <algorithm>
#include <cctype>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/spx.h"

namespace {

// ------------------------------------------------------------------ a small pull reader for JSON
// No document tree: the decoder walks the text once and asks for what it expects (object members, array elements, a string,
// a scalar as written), skipping everything else.  A 50k-node list is ~250 MB of JSON; building a DOM first made the decode
// allocation-bound (23 MB/s), the single pass runs at memory-copy-like speed for the parts it skips.
struct Reader {
  const char* p;
  const char* end;
  std::string err;
  std::string key;  // reused buffer for member names

  void ws() {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
  }
  bool fail(const char* m) {
    if (err.empty()) err = m;
    return false;
  }
  char peek() {
    ws();
    return p < end ? *p : '\0';
  }
  static void utf8(std::string& out, uint32_t c) {
    if (c < 0x80) out.push_back(static_cast<char>(c));
    else if (c < 0x800) out.push_back(static_cast<char>(0xC0 | (c >> 6))), out.push_back(static_cast<char>(0x80 | (c & 0x3F)));
    else if (c < 0x10000)
      out.push_back(static_cast<char>(0xE0 | (c >> 12))), out.push_back(static_cast<char>(0x80 | ((c >> 6) & 0x3F))),
          out.push_back(static_cast<char>(0x80 | (c & 0x3F)));
    else
      out.push_back(static_cast<char>(0xF0 | (c >> 18))), out.push_back(static_cast<char>(0x80 | ((c >> 12) & 0x3F))),
          out.push_back(static_cast<char>(0x80 | ((c >> 6) & 0x3F))), out.push_back(static_cast<char>(0x80 | (c & 0x3F)));
  }
  bool hex4(uint32_t* v) {
    if (end - p < 4) return fail("truncated \\u escape");
    uint32_t x = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = *p++;
      x <<= 4;
      if (c >= '0' && c <= '9') x |= static_cast<uint32_t>(c - '0');
      else if (c >= 'a' && c <= 'f') x |= static_cast<uint32_t>(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') x |= static_cast<uint32_t>(c - 'A' + 10);
      else return fail("bad \\u escape");
    }
    *v = x;
    return true;
  }
  // a JSON string, decoded into out (cleared first)
  bool str(std::string& out) {
    out.clear();
    ws();
    if (p >= end || *p != '"') return fail("expected string");
    ++p;
    for (;;) {
      const char* q = p;
      while (q < end && *q != '"' && *q != '\\') ++q;
      out.append(p, q);
      p = q;
      if (p >= end) return fail("unterminated string");
      if (*p == '"') return ++p, true;
      if (++p >= end) return fail("truncated escape");
      const char c = *p++;
      switch (c) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          uint32_t u;
          if (!hex4(&u)) return false;
          if (u >= 0xD800 && u < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
            p += 2;
            uint32_t lo;
            if (!hex4(&lo)) return false;
            u = 0x10000 + ((u - 0xD800) << 10) + (lo - 0xDC00);
          }
          utf8(out, u);
          break;
        }
        default: return fail("unknown escape");
      }
    }
  }
  // a string, or a number exactly as written (quantities may be either); anything else is an error
  bool scalar(std::string& out) {
    const char c = peek();
    if (c == '"') return str(out);
    if (c == '-' || (c >= '0' && c <= '9')) {
      const char* q = p;
      while (p < end && (*p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E' || (*p >= '0' && *p <= '9'))) ++p;
      out.assign(q, p);
      return true;
    }
    return fail("expected a string or a number");
  }
  bool skip(int depth = 0) {
    if (depth > 64) return fail("nesting too deep");
    const char c = peek();
    if (c == '"') {
      ++p;
      while (p < end && *p != '"') p += (*p == '\\' && p + 1 < end) ? 2 : 1;
      if (p >= end) return fail("unterminated string");
      return ++p, true;
    }
    if (c == '{' || c == '[') {
      const char close = c == '{' ? '}' : ']';
      ++p;
      if (peek() == close) return ++p, true;
      for (;;) {
        if (c == '{') {
          if (peek() != '"' || !skip(depth + 1)) return fail("expected member name");
          if (peek() != ':') return fail("expected ':'");
          ++p;
        }
        if (!skip(depth + 1)) return false;
        const char d = peek();
        if (d == ',') {
          ++p;
          continue;
        }
        if (d == close) return ++p, true;
        return fail("expected ',' or a closing bracket");
      }
    }
    if (c == 't' && end - p >= 4 && !std::memcmp(p, "true", 4)) return p += 4, true;
    if (c == 'f' && end - p >= 5 && !std::memcmp(p, "false", 5)) return p += 5, true;
    if (c == 'n' && end - p >= 4 && !std::memcmp(p, "null", 4)) return p += 4, true;
    if (c == '-' || (c >= '0' && c <= '9')) {
      while (p < end && (*p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E' || (*p >= '0' && *p <= '9'))) ++p;
      return true;
    }
    return fail(p >= end ? "unexpected end of input" : "unexpected character");
  }
  // for each member: fn(name) must consume the value (read it or skip()) and return false on error
  template <typename Fn>
  bool object(Fn&& fn) {
    if (peek() != '{') return fail("expected an object");
    ++p;
    if (peek() == '}') return ++p, true;
    for (;;) {
      if (!str(key)) return false;
      if (peek() != ':') return fail("expected ':'");
      ++p;
      if (!fn(key)) return err.empty() ? fail("rejected member") : false;
      const char d = peek();
      if (d == ',') {
        ++p;
        continue;
      }
      if (d == '}') return ++p, true;
      return fail("expected ',' or '}'");
    }
  }
  template <typename Fn>
  bool array(Fn&& fn) {
    if (peek() != '[') return fail("expected an array");
    ++p;
    if (peek() == ']') return ++p, true;
    for (;;) {
      if (!fn()) return err.empty() ? fail("rejected element") : false;
      const char d = peek();
      if (d == ',') {
        ++p;
        continue;
      }
      if (d == ']') return ++p, true;
      return fail("expected ',' or ']'");
    }
  }
};

// ------------------------------------------------------------------ resource.Quantity -> canonical int64
// text = <sign><digits>[.<digits>][suffix | e<exp>]; value * scale (1000 for cpu, 1 otherwise) rounded up, as MilliValue() /
// Value() do.  Exact: the digits are accumulated into an unsigned 128-bit integer together with a decimal exponent.
bool canonical_quantity(const std::string& t, bool milli, int64_t* out) {
  size_t i = 0, n = t.size();
  while (i < n && t[i] == ' ') ++i;
  bool neg = false;
  if (i < n && (t[i] == '+' || t[i] == '-')) neg = t[i++] == '-';
  unsigned __int128 mant = 0;
  int dec = 0;  // value = mant * 10^dec
  bool any = false, frac = false, sticky = false;
  for (; i < n; ++i) {
    const char c = t[i];
    if (c >= '0' && c <= '9') {
      if (mant > UINT64_MAX / 10) {
        // more than ~19 significant digits (apimachinery holds them in an inf.Dec): keep the leading ones exactly and remember
        // that something non-zero was dropped — enough for the round-up that Value() / MilliValue() apply
        if (c != '0') sticky = true;
        if (!frac) ++dec;
      } else {
        mant = mant * 10 + static_cast<unsigned>(c - '0');
        if (frac) --dec;
      }
      any = true;
    } else if (c == '.' && !frac) {
      frac = true;
    } else {
      break;
    }
  }
  if (!any) return false;
  std::string suf = t.substr(i);
  while (!suf.empty() && suf.back() == ' ') suf.pop_back();
  int bin = 0;  // additional factor 2^bin
  const bool exponent = suf.size() > 1 && (suf[0] == 'e' || suf[0] == 'E') && (suf[1] == '+' || suf[1] == '-' || (suf[1] >= '0' && suf[1] <= '9'));
  if (exponent) {  // a lone "E" is the exa suffix, "Ei" the exbi suffix
    size_t k = 1;
    bool eneg = false;
    if (suf[k] == '+' || suf[k] == '-') eneg = suf[k++] == '-';
    if (k >= suf.size()) return false;
    int e = 0;
    for (; k < suf.size(); ++k) {
      if (suf[k] < '0' || suf[k] > '9' || e > 1000) return false;
      e = e * 10 + (suf[k] - '0');
    }
    dec += eneg ? -e : e;
  } else if (suf == "Ki") bin = 10;
  else if (suf == "Mi") bin = 20;
  else if (suf == "Gi") bin = 30;
  else if (suf == "Ti") bin = 40;
  else if (suf == "Pi") bin = 50;
  else if (suf == "Ei") bin = 60;
  else if (suf == "n") dec -= 9;
  else if (suf == "u") dec -= 6;
  else if (suf == "m") dec -= 3;
  else if (suf == "k") dec += 3;
  else if (suf == "M") dec += 6;
  else if (suf == "G") dec += 9;
  else if (suf == "T") dec += 12;
  else if (suf == "P") dec += 15;
  else if (suf == "E") dec += 18;
  else if (!suf.empty()) return false;
  if (milli) dec += 3;
  const unsigned __int128 kMax = static_cast<unsigned __int128>(INT64_MAX);
  mant <<= bin;  // mant < 2^68 here, bin <= 60: fits 128 bits; the range check below decides
  bool inexact = sticky;
  while (dec > 0 && mant != 0) {
    if (mant > kMax) return false;
    mant *= 10;
    --dec;
  }
  while (dec < 0 && mant != 0) {
    if (mant % 10 != 0) inexact = true;
    mant /= 10;
    ++dec;
  }
  if (mant > kMax) return false;
  // inexact values round AWAY from zero for either sign (negativeScaleInt64 in apimachinery's amount.go: value++ / value--)
  if (inexact && mant == kMax) return false;
  int64_t v = static_cast<int64_t>(mant) + (inexact ? 1 : 0);
  if (neg) v = -v;
  *out = v;
  return true;
}

// ------------------------------------------------------------------ resource names -> spx.h ids + class flags
struct Interner {
  std::unordered_map<std::string, int32_t> ids;
  std::vector<std::string> names;  // index = id
  Interner() {
    names.resize(SPX_RES_FIRST_DYNAMIC);
    const std::pair<const char*, int32_t> fixed[] = {{"cpu", SPX_RES_CPU}, {"memory", SPX_RES_MEMORY}, {"ephemeral-storage", SPX_RES_EPHEMERAL},
                                                     {"pods", SPX_RES_PODS}, {"storage", SPX_RES_STORAGE}};
    for (const auto& f : fixed) ids[f.first] = f.second, names[static_cast<size_t>(f.second)] = f.first;
  }
  int32_t id(const std::string& name) {
    auto it = ids.find(name);
    if (it != ids.end()) return it->second;
    const int32_t v = static_cast<int32_t>(names.size());
    ids.emplace(name, v);
    names.push_back(name);
    return v;
  }
  static uint8_t flags(const std::string& n) {
    const bool native = n.find('/') == std::string::npos || n.find("kubernetes.io/") != std::string::npos;  // v1helper.IsNativeResource
    const bool huge = n.rfind("hugepages-", 0) == 0;                                                        // IsHugePageResourceName
    const bool scalar = (!native && n.rfind("requests.", 0) != 0) || huge || n.find("kubernetes.io/") != std::string::npos ||
                        n.rfind("attachable-volumes-", 0) == 0;                                             // schedutil.IsScalarResourceName
    return static_cast<uint8_t>((huge ? SPX_RC_HUGEPAGE : 0) | (native ? SPX_RC_NATIVE : 0) | (scalar ? SPX_RC_SCALAR : 0));
  }
};

int numa_id_of(const std::string& name) {  // numanode.NameToID: "node-<id>"
  if (name.rfind("node-", 0) != 0 || name.size() == 5) return -1;
  int64_t v = 0;
  size_t i = 5;
  bool neg = false;
  if (name[i] == '+' || name[i] == '-') neg = name[i++] == '-';  // strconv.Atoi accepts a sign
  if (i >= name.size()) return -1;
  for (; i < name.size(); ++i) {
    if (name[i] < '0' || name[i] > '9' || v > INT32_MAX) return -1;
    v = v * 10 + (name[i] - '0');
  }
  if (v > INT32_MAX) return -1;
  return neg ? -1 : static_cast<int>(v);  // negative ids never match a zone; the tables use -1 for "no id"
}

struct ZoneRow {
  bool is_node;
  int32_t numa_id;
  std::vector<int32_t> res;
  std::vector<int64_t> avail, alloc;
  std::vector<int32_t> cost_id;
  std::vector<int64_t> cost;
};
struct NodeRow {
  bool has = false;
  int8_t legacy = -1, scope = -1, policy = -1;
  int32_t max_numa = -1;
  std::vector<ZoneRow> zones;
};

}  // namespace

struct spx_ingest_quota;
static void free_quota(spx_ingest_quota* q);

// AppGroup CR as decoded (selectors still strings: their ids are assigned, and re-assigned, in lexicographic order)
struct DepRow {
  std::string selector;
  int64_t max_cost = 0;
};
struct WorkloadRow {
  std::string selector;
  std::vector<DepRow> deps;
};
struct GroupRow {
  std::string name;
  std::vector<WorkloadRow> workloads;
  std::vector<std::pair<std::string, int64_t>> topo;  // Status.TopologyOrder as written (the plugins binary-search it as is)
};

struct spx_ingest {
  spx_ingest_quota* quota = nullptr;  // ElasticQuota tables (end of this file)
  Interner res;
  std::vector<NodeRow> rows;
  std::string err;
  int64_t unknown = 0;  // objects whose metadata.name is not in the node list
  // frozen tables
  std::vector<uint8_t> has_nrt, fresh, zone_is_node, flags;
  std::vector<int8_t> legacy, scope, policy;
  std::vector<int32_t> max_numa, zone_ptr, zone_numa_id, zres_ptr, zres_res, zcost_ptr, zcost_id, assumed_ptr, arl_ptr;
  std::vector<int64_t> zres_avail, zres_alloc, zcost_val;
  spx_nrt_objects table{};
  spx_resource_classes classes{};
  std::unordered_map<std::string, int64_t> node_index;

  // ---- v1.Node / v1.Pod
  struct Names {
    std::unordered_map<std::string, int32_t> ids;
    std::vector<std::string> names;
    int32_t id(const std::string& n) {
      auto it = ids.find(n);
      if (it != ids.end()) return it->second;
      const int32_t v = static_cast<int32_t>(names.size());
      ids.emplace(n, v);
      names.push_back(n);
      return v;
    }
    int32_t find(const char* n) const {
      auto it = ids.find(n);
      return it == ids.end() ? -1 : it->second;
    }
  };
  Names regions, zones, namespaces, appgroups, selectors;
  struct NodeObj {
    int64_t alloc_cpu = 0, alloc_mem = 0, alloc_eph = 0, alloc_pods = 0, cap_cpu = 0;
    std::vector<std::pair<int32_t, int64_t>> scalars;
    int32_t region = -1, zone = -1;
  };
  std::vector<NodeObj> node_objs;
  std::vector<int64_t> n_alloc_cpu, n_alloc_mem, n_alloc_eph, n_alloc_pods, n_scalar_qty, n_cap_cpu;
  std::vector<int32_t> n_scalar_ptr, n_scalar_res, n_region, n_zone;
  spx_node_objects node_table{};
  // pods are appended in document order
  std::vector<int32_t> p_ctr_ptr{0}, p_req_ptr{0}, p_lim_ptr{0}, p_ovh_ptr{0}, p_req_res, p_lim_res, p_ovh_res, p_priority, p_appgroup, p_selector, p_ns;
  std::vector<uint8_t> p_kind;
  std::vector<int64_t> p_req_qty, p_lim_qty, p_ovh_qty, p_queue_ts;
  std::vector<int32_t> p_nominated;  // status.nominatedNodeName as a node index of the CURRENT node snapshot, -1 = none / unknown node:
                                     // derived from p_nominated_name by rebuild_nominated, so pods may be ingested before nodes and
                                     // the node snapshot may be replaced afterwards
  std::vector<std::string> p_nominated_name;  // status.nominatedNodeName as written ("" = none)
  spx_pod_objects pod_table{};
  // ---- AppGroup / NetworkTopology CRs
  std::vector<int32_t> g_wl_ptr{0}, g_wl_selector, g_dep_ptr{0}, g_dep_selector, g_topo_ptr{0}, g_topo_selector, g_topo_index, g_placed_ptr{0};
  std::vector<int64_t> g_dep_max_cost;
  // ---- load-watcher metrics (watcher.WatcherMetrics)
  std::vector<uint8_t> m_present, m_nil, m_type, m_op;
  std::vector<int32_t> m_ptr;
  std::vector<double> m_value;
  spx_metrics_objects metrics_table{};
  bool metrics_valid = false;
  std::vector<GroupRow> group_rows;  // indexed by AppGroup name id (kind 3) — the id pods carry; empty row = no CR seen
  spx_appgroup_objects group_table{};
  std::vector<std::vector<std::pair<int32_t, int64_t>>> nt_region, nt_zone;  // per origin id: (destination id, cost) in document order
  std::vector<int32_t> nt_rc_ptr{0}, nt_rc_dest, nt_zc_ptr{0}, nt_zc_dest;
  std::vector<int64_t> nt_rc_cost, nt_zc_cost;
  spx_nettopo_objects nettopo_table{};
};

namespace {

// one NodeResourceTopology object at the reader's position
bool decode_one(spx_ingest* h, Reader& r) {
  NodeRow row;
  row.has = true;
  std::string name, buf, an, av;
  bool have_name = false;
  auto fail = [&](const std::string& m) {
    if (h->err.empty()) h->err = m;
    return false;
  };
  const bool ok = r.object([&](const std::string& k) {
    if (k == "metadata") {
      return r.object([&](const std::string& mk) {
        if (mk != "name") return r.skip();
        have_name = true;
        return r.str(name);
      });
    }
    if (k == "topologyPolicies") {  // nodeconfig/topologymanager.go:131-162: only the first entry is read
      bool first = true;
      return r.array([&] {
        if (!first || r.peek() != '"') return r.skip();
        first = false;
        if (!r.str(buf)) return false;
        static const std::pair<const char*, int> kLegacy[] = {
            {"SingleNUMANodeContainerLevel", (3 << 1) | 0}, {"SingleNUMANodePodLevel", (3 << 1) | 1}, {"BestEffortContainerLevel", (1 << 1) | 0},
            {"BestEffortPodLevel", (1 << 1) | 1},           {"RestrictedContainerLevel", (2 << 1) | 0}, {"RestrictedPodLevel", (2 << 1) | 1}};
        for (const auto& e : kLegacy)
          if (buf == e.first) row.legacy = static_cast<int8_t>(e.second);
        return true;
      });
    }
    if (k == "attributes") {  // :98-115
      return r.array([&] {
        an.clear(), av.clear();
        bool has_n = false, has_v = false;
        if (!r.object([&](const std::string& ak) {
              if (ak == "name" && r.peek() == '"') return has_n = true, r.str(an);
              if (ak == "value" && r.peek() == '"') return has_v = true, r.str(av);
              return r.skip();
            }))
          return false;
        if (!has_n || !has_v) return true;
        if (an == "topologyManagerScope") {
          if (av == "container") row.scope = 0;
          else if (av == "pod") row.scope = 1;
        } else if (an == "topologyManagerPolicy") {
          if (av == "none") row.policy = 0;
          else if (av == "best-effort") row.policy = 1;
          else if (av == "restricted") row.policy = 2;
          else if (av == "single-numa-node") row.policy = 3;
        } else if (an == "topologyManagerMaxNUMANodes") {
          int64_t v = 0;
          size_t i = 0;
          bool good = !av.empty(), neg = false;
          if (good && (av[0] == '+' || av[0] == '-')) neg = av[i++] == '-';
          good = good && i < av.size();
          for (; good && i < av.size(); ++i) {
            if (av[i] < '0' || av[i] > '9' || v > INT32_MAX) good = false;
            else v = v * 10 + (av[i] - '0');
          }
          if (good && !neg && v > 1 && v <= INT32_MAX) row.max_numa = static_cast<int32_t>(v);  // values <= 1 are ignored (:108-113)
        }
        return true;
      });
    }
    if (k == "zones") {
      return r.array([&] {
        ZoneRow zr;
        zr.is_node = false;
        zr.numa_id = -1;
        if (!r.object([&](const std::string& zk) {
              if (zk == "name" && r.peek() == '"') {
                if (!r.str(buf)) return false;
                zr.numa_id = numa_id_of(buf);
                return true;
              }
              if (zk == "type" && r.peek() == '"') {
                if (!r.str(buf)) return false;
                zr.is_node = buf == "Node";
                return true;
              }
              if (zk == "resources") {
                return r.array([&] {
                  // members may come in any order: the name decides the unit, so keep the texts and convert afterwards
                  std::string rn, t_avail, t_alloc;
                  bool has_rn = false, has_av = false, has_al = false;
                  if (!r.object([&](const std::string& rk) {
                        if (rk == "name" && r.peek() == '"') return has_rn = true, r.str(rn);
                        if (rk == "available") return has_av = true, r.scalar(t_avail);
                        if (rk == "allocatable") return has_al = true, r.scalar(t_alloc);
                        return r.skip();
                      }))
                    return false;
                  if (!has_rn) return fail("zone resource without a name");
                  const bool milli = rn == "cpu";
                  int64_t avail = 0, alloc = 0;
                  if (!has_av || !canonical_quantity(t_avail, milli, &avail)) return fail("bad or missing 'available' quantity of " + rn);
                  if (!has_al || !canonical_quantity(t_alloc, milli, &alloc)) alloc = avail;
                  zr.res.push_back(h->res.id(rn));
                  zr.avail.push_back(avail);
                  zr.alloc.push_back(alloc);
                  return true;
                });
              }
              if (zk == "costs") {
                return r.array([&] {
                  std::string cn, cv;
                  bool has_cn = false, has_cv = false;
                  if (!r.object([&](const std::string& ck) {
                        if (ck == "name" && r.peek() == '"') return has_cn = true, r.str(cn);
                        if (ck == "value") return has_cv = true, r.scalar(cv);
                        return r.skip();
                      }))
                    return false;
                  int64_t v = 0;
                  if (!has_cn || !has_cv || !canonical_quantity(cv, false, &v)) return fail("bad zone cost entry");
                  zr.cost_id.push_back(numa_id_of(cn));
                  zr.cost.push_back(v);
                  return true;
                });
              }
              return r.skip();
            }))
          return false;
        row.zones.push_back(std::move(zr));
        return true;
      });
    }
    return r.skip();
  });
  if (!ok) return false;
  if (!have_name) return fail("NodeResourceTopology without metadata.name");
  auto it = h->node_index.find(name);
  if (it == h->node_index.end()) {
    ++h->unknown;
    return true;
  }
  h->rows[static_cast<size_t>(it->second)] = std::move(row);  // a later object of the same name replaces the earlier one
  return true;
}

void freeze(spx_ingest* h) {
  const size_t n = h->rows.size();
  h->has_nrt.assign(n, 0), h->fresh.assign(n, 1), h->legacy.assign(n, -1), h->scope.assign(n, -1), h->policy.assign(n, -1);
  h->max_numa.assign(n, -1);
  h->zone_ptr.assign(1, 0), h->zres_ptr.assign(1, 0), h->zcost_ptr.assign(1, 0);
  h->zone_is_node.clear(), h->zone_numa_id.clear(), h->zres_res.clear(), h->zres_avail.clear(), h->zres_alloc.clear();
  h->zcost_id.clear(), h->zcost_val.clear();
  for (size_t i = 0; i < n; ++i) {
    const NodeRow& r = h->rows[i];
    h->has_nrt[i] = r.has, h->legacy[i] = r.legacy, h->scope[i] = r.scope, h->policy[i] = r.policy, h->max_numa[i] = r.max_numa;
    for (const ZoneRow& z : r.zones) {
      h->zone_is_node.push_back(z.is_node);
      h->zone_numa_id.push_back(z.numa_id);
      h->zres_res.insert(h->zres_res.end(), z.res.begin(), z.res.end());
      h->zres_avail.insert(h->zres_avail.end(), z.avail.begin(), z.avail.end());
      h->zres_alloc.insert(h->zres_alloc.end(), z.alloc.begin(), z.alloc.end());
      h->zres_ptr.push_back(static_cast<int32_t>(h->zres_res.size()));
      h->zcost_id.insert(h->zcost_id.end(), z.cost_id.begin(), z.cost_id.end());
      h->zcost_val.insert(h->zcost_val.end(), z.cost.begin(), z.cost.end());
      h->zcost_ptr.push_back(static_cast<int32_t>(h->zcost_id.size()));
    }
    h->zone_ptr.push_back(static_cast<int32_t>(h->zone_is_node.size()));
  }
  h->assumed_ptr.assign(n + 1, 0);
  h->arl_ptr.assign(1, 0);
  h->flags.assign(h->res.names.size(), 0);
  for (size_t i = 0; i < h->res.names.size(); ++i)
    if (!h->res.names[i].empty()) h->flags[i] = Interner::flags(h->res.names[i]);
  spx_nrt_objects& t = h->table;
  t.n_nodes = static_cast<int64_t>(n);
  t.has_nrt = h->has_nrt.data(), t.fresh = h->fresh.data(), t.legacy_policy = h->legacy.data(), t.attr_scope = h->scope.data();
  t.attr_policy = h->policy.data(), t.attr_max_numa = h->max_numa.data(), t.zone_ptr = h->zone_ptr.data();
  t.zone_is_node = h->zone_is_node.data(), t.zone_numa_id = h->zone_numa_id.data(), t.zres_ptr = h->zres_ptr.data();
  t.zres_res = h->zres_res.data(), t.zres_avail = h->zres_avail.data(), t.zcost_ptr = h->zcost_ptr.data();
  t.zcost_numa_id = h->zcost_id.data(), t.zcost_value = h->zcost_val.data(), t.assumed_ptr = h->assumed_ptr.data();
  t.arl_ptr = h->arl_ptr.data(), t.arl_res = nullptr, t.arl_qty = nullptr, t.zres_allocatable = h->zres_alloc.data();
  h->classes.n_res = static_cast<int32_t>(h->flags.size());
  h->classes.flags = h->flags.data();
}

}  // namespace

static void freeze_nodes_initial(spx_ingest* h);
extern "C" int spx_ingest_create(const char* const* node_names, int64_t n_nodes, const char* const* resource_names, int32_t n_resource_names,
                                 spx_ingest** out) {
  if (!out || n_nodes <= 0 || !node_names) return SPX_ERR_ARG;
  auto* h = new spx_ingest();
  h->rows.resize(static_cast<size_t>(n_nodes));
  h->node_objs.resize(static_cast<size_t>(n_nodes));
  for (int64_t i = 0; i < n_nodes; ++i) {
    if (!node_names[i]) {
      delete h;
      return SPX_ERR_ARG;
    }
    h->node_index.emplace(node_names[i], i);  // first occurrence wins for duplicate names
  }
  for (int32_t i = 0; i < n_resource_names; ++i)
    if (resource_names && resource_names[i]) h->res.id(resource_names[i]);
  freeze(h);
  freeze_nodes_initial(h);
  *out = h;
  return SPX_OK;
}

extern "C" int spx_ingest_destroy(spx_ingest* h) {
  if (h) free_quota(h->quota);
  delete h;
  return SPX_OK;
}

extern "C" const char* spx_ingest_error(const spx_ingest* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int spx_ingest_nrt_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_objects_out, int64_t* n_unknown_out) {
  if (!h || !json || len < 0) return SPX_ERR_ARG;
  h->err.clear();
  h->unknown = 0;
  Reader r{json, json + len, {}, {}};
  int64_t n = 0;
  auto one = [&] {
    if (r.peek() != '{') return r.fail("list entry is not an object");
    ++n;
    return decode_one(h, r);
  };
  bool ok;
  const char c = r.peek();
  if (c == '[') {
    ok = r.array(one);
  } else if (c == '{') {
    // a List carries "items"; anything else is a single object.  Look ahead for a top-level "items" member without decoding.
    Reader probe{r.p, r.end, {}, {}};
    bool is_list = false;
    probe.object([&](const std::string& k) {
      if (k == "items") is_list = true;
      return probe.skip();
    });
    if (is_list) {
      ok = r.object([&](const std::string& k) { return k == "items" ? r.array(one) : r.skip(); });
    } else {
      ok = one();
    }
  } else {
    ok = r.fail("expected a NodeResourceTopology object, a List or an array");
  }
  if (ok && r.peek() != '\0') ok = r.fail("trailing characters");
  if (!ok) {
    if (h->err.empty()) h->err = "JSON: " + r.err;
    freeze(h);
    return SPX_ERR_ARG;
  }
  freeze(h);
  if (n_objects_out) *n_objects_out = n;
  if (n_unknown_out) *n_unknown_out = h->unknown;
  return SPX_OK;
}

extern "C" const spx_nrt_objects* spx_ingest_nrt_objects(const spx_ingest* h) { return h ? &h->table : nullptr; }
extern "C" const spx_resource_classes* spx_ingest_resource_classes(const spx_ingest* h) { return h ? &h->classes : nullptr; }
extern "C" int32_t spx_ingest_resource_id(const spx_ingest* h, const char* name) {
  if (!h || !name) return -1;
  auto it = h->res.ids.find(name);
  return it == h->res.ids.end() ? -1 : it->second;
}
extern "C" int spx_ingest_quantity(const char* text, int32_t milli, int64_t* out) {
  if (!text || !out) return SPX_ERR_ARG;
  return canonical_quantity(text, milli != 0, out) ? SPX_OK : SPX_ERR_ARG;
}

// ====================================================================== v1.Node / v1.Pod
namespace {

// top level: one object, a List ("items") or an array; calls one() with the reader positioned at each object
template <typename Fn>
bool for_each_object(Reader& r, Fn&& one) {
  const char c = r.peek();
  if (c == '[') return r.array([&] { return r.peek() == '{' ? one() : r.fail("list entry is not an object"); });
  if (c != '{') return r.fail("expected an object, a List or an array");
  Reader probe{r.p, r.end, {}, {}};
  bool is_list = false;
  probe.object([&](const std::string& k) {
    if (k == "items") is_list = true;
    return probe.skip();
  });
  if (!is_list) return one();
  return r.object([&](const std::string& k) {
    if (k != "items") return r.skip();
    return r.array([&] { return r.peek() == '{' ? one() : r.fail("list entry is not an object"); });
  });
}

// a v1.ResourceList object: fn(resource name, canonical quantity) per member, in document order
template <typename Fn>
bool resource_list(spx_ingest* h, Reader& r, std::string& buf, Fn&& fn) {
  if (r.peek() == 'n') return r.skip();  // null
  return r.object([&](const std::string& name) {
    const std::string rn = name;  // the reader reuses its key buffer
    if (!r.scalar(buf)) return false;
    int64_t q = 0;
    if (!canonical_quantity(buf, rn == "cpu", &q)) {
      if (h->err.empty()) h->err = "bad quantity for " + rn + ": " + buf;
      return false;
    }
    fn(rn, q);
    return true;
  });
}

bool scalar_resource_name(const std::string& n) { return (Interner::flags(n) & SPX_RC_SCALAR) != 0; }

bool decode_node(spx_ingest* h, Reader& r) {
  spx_ingest::NodeObj o;
  std::string name, buf, region, zone;
  bool have_name = false, has_region = false, has_zone = false;
  const bool ok = r.object([&](const std::string& k) {
    if (k == "metadata") {
      return r.object([&](const std::string& mk) {
        if (mk == "name" && r.peek() == '"') return have_name = true, r.str(name);
        if (mk == "labels" && r.peek() == '{') {
          return r.object([&](const std::string& lk) {
            if (lk == "topology.kubernetes.io/region" && r.peek() == '"') return has_region = true, r.str(region);
            if (lk == "topology.kubernetes.io/zone" && r.peek() == '"') return has_zone = true, r.str(zone);
            return r.skip();
          });
        }
        return r.skip();
      });
    }
    if (k == "status" && r.peek() == '{') {
      return r.object([&](const std::string& sk) {
        if (sk == "allocatable") {
          return resource_list(h, r, buf, [&](const std::string& rn, int64_t q) {
            if (rn == "cpu") o.alloc_cpu = q;
            else if (rn == "memory") o.alloc_mem = q;
            else if (rn == "ephemeral-storage") o.alloc_eph = q;
            else if (rn == "pods") o.alloc_pods = q;
            else if (scalar_resource_name(rn)) o.scalars.emplace_back(h->res.id(rn), q);  // framework.Resource.Add keeps scalar names only
          });
        }
        if (sk == "capacity") {
          return resource_list(h, r, buf, [&](const std::string& rn, int64_t q) {
            if (rn == "cpu") o.cap_cpu = q;
          });
        }
        return r.skip();
      });
    }
    return r.skip();
  });
  if (!ok) return false;
  if (!have_name) return h->err = "Node without metadata.name", false;
  auto it = h->node_index.find(name);
  if (it == h->node_index.end()) {
    ++h->unknown;
    return true;
  }
  // an empty label value counts as unset, as the plugins' `region != ""` tests treat it (networkoverhead.go:459, :479)
  o.region = (has_region && !region.empty()) ? h->regions.id(region) : -1;
  o.zone = (has_zone && !zone.empty()) ? h->zones.id(zone) : -1;
  h->node_objs[static_cast<size_t>(it->second)] = std::move(o);
  return true;
}

void freeze_nodes(spx_ingest* h) {
  const size_t n = h->node_objs.size();
  h->n_alloc_cpu.resize(n), h->n_alloc_mem.resize(n), h->n_alloc_eph.resize(n), h->n_alloc_pods.resize(n), h->n_cap_cpu.resize(n);
  h->n_region.resize(n), h->n_zone.resize(n);
  h->n_scalar_ptr.assign(1, 0), h->n_scalar_res.clear(), h->n_scalar_qty.clear();
  for (size_t i = 0; i < n; ++i) {
    const auto& o = h->node_objs[i];
    h->n_alloc_cpu[i] = o.alloc_cpu, h->n_alloc_mem[i] = o.alloc_mem, h->n_alloc_eph[i] = o.alloc_eph, h->n_alloc_pods[i] = o.alloc_pods;
    h->n_cap_cpu[i] = o.cap_cpu, h->n_region[i] = o.region, h->n_zone[i] = o.zone;
    for (const auto& s : o.scalars) h->n_scalar_res.push_back(s.first), h->n_scalar_qty.push_back(s.second);
    h->n_scalar_ptr.push_back(static_cast<int32_t>(h->n_scalar_res.size()));
  }
  spx_node_objects& t = h->node_table;
  t.n_nodes = static_cast<int64_t>(n);
  t.alloc_cpu_milli = h->n_alloc_cpu.data(), t.alloc_mem = h->n_alloc_mem.data(), t.alloc_eph = h->n_alloc_eph.data();
  t.alloc_pods = h->n_alloc_pods.data(), t.scalar_ptr = h->n_scalar_ptr.data(), t.scalar_res = h->n_scalar_res.data();
  t.scalar_qty = h->n_scalar_qty.data(), t.cap_cpu_milli = h->n_cap_cpu.data(), t.region = h->n_region.data(), t.zone = h->n_zone.data();
}

// RFC 3339 timestamp -> microseconds since the epoch (0 when malformed): metav1.Time as the API server writes it
int64_t rfc3339_micros(const std::string& t) {
  auto num = [&](size_t pos, size_t len, int* out) {
    if (pos + len > t.size()) return false;
    int v = 0;
    for (size_t i = pos; i < pos + len; ++i) {
      if (t[i] < '0' || t[i] > '9') return false;
      v = v * 10 + (t[i] - '0');
    }
    *out = v;
    return true;
  };
  int y, mo, d, hh, mm, ss;
  if (t.size() < 20 || !num(0, 4, &y) || t[4] != '-' || !num(5, 2, &mo) || t[7] != '-' || !num(8, 2, &d) || (t[10] != 'T' && t[10] != 't') ||
      !num(11, 2, &hh) || t[13] != ':' || !num(14, 2, &mm) || t[16] != ':' || !num(17, 2, &ss))
    return 0;
  size_t i = 19;
  int64_t frac = 0;
  if (i < t.size() && t[i] == '.') {
    int digits = 0;
    for (++i; i < t.size() && t[i] >= '0' && t[i] <= '9'; ++i)
      if (digits < 6) frac = frac * 10 + (t[i] - '0'), ++digits;
    for (; digits < 6; ++digits) frac *= 10;
  }
  int64_t offset = 0;
  if (i < t.size() && (t[i] == '+' || t[i] == '-')) {
    int oh, om;
    if (!num(i + 1, 2, &oh) || i + 3 >= t.size() || t[i + 3] != ':' || !num(i + 4, 2, &om)) return 0;
    offset = (t[i] == '-' ? -1 : 1) * (oh * 3600 + om * 60);
  } else if (i >= t.size() || (t[i] != 'Z' && t[i] != 'z')) {
    return 0;
  }
  // days from civil (proleptic Gregorian)
  const int yy = y - (mo <= 2);
  const int era = (yy >= 0 ? yy : yy - 399) / 400;
  const unsigned yoe = static_cast<unsigned>(yy - era * 400);
  const unsigned doy = static_cast<unsigned>((153 * (mo + (mo > 2 ? -3 : 9)) + 2) / 5 + d - 1);
  const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  const int64_t days = static_cast<int64_t>(era) * 146097 + static_cast<int64_t>(doe) - 719468;
  return ((days * 86400 + hh * 3600 + mm * 60 + ss) - offset) * 1000000 + frac;
}

struct CtrRow {
  uint8_t kind;
  std::vector<std::pair<int32_t, int64_t>> req, lim;
};

bool decode_container(spx_ingest* h, Reader& r, std::string& buf, bool init, CtrRow* c) {
  c->kind = init ? SPX_CTR_INIT : SPX_CTR_APP;
  return r.object([&](const std::string& k) {
    if (k == "resources" && r.peek() == '{') {
      return r.object([&](const std::string& rk) {
        if (rk == "requests") return resource_list(h, r, buf, [&](const std::string& rn, int64_t q) { c->req.emplace_back(h->res.id(rn), q); });
        if (rk == "limits") return resource_list(h, r, buf, [&](const std::string& rn, int64_t q) { c->lim.emplace_back(h->res.id(rn), q); });
        return r.skip();
      });
    }
    if (k == "restartPolicy" && init && r.peek() == '"') {  // a restartable init container is a sidecar (pkg/util/sidecar.go)
      if (!r.str(buf)) return false;
      if (buf == "Always") c->kind = SPX_CTR_SIDECAR;
      return true;
    }
    return r.skip();
  });
}

bool decode_pod(spx_ingest* h, Reader& r) {
  std::vector<CtrRow> init, app;
  std::vector<std::pair<int32_t, int64_t>> overhead;
  std::string buf, ns, group, selector, created, nominated;
  bool has_group = false, has_selector = false;
  int64_t priority = 0;
  const bool ok = r.object([&](const std::string& k) {
    if (k == "metadata" && r.peek() == '{') {
      return r.object([&](const std::string& mk) {
        if (mk == "namespace" && r.peek() == '"') return r.str(ns);
        if (mk == "creationTimestamp" && r.peek() == '"') return r.str(created);
        if (mk == "labels" && r.peek() == '{') {
          return r.object([&](const std::string& lk) {
            if (lk == "appgroup.diktyo.x-k8s.io" && r.peek() == '"') return has_group = true, r.str(group);
            if (lk == "appgroup.diktyo.x-k8s.io.workload" && r.peek() == '"') return has_selector = true, r.str(selector);
            return r.skip();
          });
        }
        return r.skip();
      });
    }
    if (k == "spec" && r.peek() == '{') {
      return r.object([&](const std::string& sk) {
        if (sk == "containers" && r.peek() == '[') {
          return r.array([&] {
            app.emplace_back();
            return decode_container(h, r, buf, false, &app.back());
          });
        }
        if (sk == "initContainers" && r.peek() == '[') {
          return r.array([&] {
            init.emplace_back();
            return decode_container(h, r, buf, true, &init.back());
          });
        }
        if (sk == "overhead") return resource_list(h, r, buf, [&](const std::string& rn, int64_t q) { overhead.emplace_back(h->res.id(rn), q); });
        if (sk == "priority" && r.peek() != 'n') {
          if (!r.scalar(buf) || !canonical_quantity(buf, false, &priority)) return r.fail("bad spec.priority");
          return true;
        }
        return r.skip();
      });
    }
    if (k == "status" && r.peek() == '{')
      return r.object([&](const std::string& sk) { return (sk == "nominatedNodeName" && r.peek() == '"') ? r.str(nominated) : r.skip(); });
    return r.skip();
  });
  if (!ok) return false;
  auto emit = [&](const CtrRow& c) {  // init containers first, then app containers: the order the reference walks them in
    h->p_kind.push_back(c.kind);
    for (const auto& e : c.req) h->p_req_res.push_back(e.first), h->p_req_qty.push_back(e.second);
    for (const auto& e : c.lim) h->p_lim_res.push_back(e.first), h->p_lim_qty.push_back(e.second);
    h->p_req_ptr.push_back(static_cast<int32_t>(h->p_req_res.size()));
    h->p_lim_ptr.push_back(static_cast<int32_t>(h->p_lim_res.size()));
  };
  for (const CtrRow& c : init) emit(c);
  for (const CtrRow& c : app) emit(c);
  h->p_ctr_ptr.push_back(static_cast<int32_t>(h->p_kind.size()));
  for (const auto& e : overhead) h->p_ovh_res.push_back(e.first), h->p_ovh_qty.push_back(e.second);
  h->p_ovh_ptr.push_back(static_cast<int32_t>(h->p_ovh_res.size()));
  h->p_priority.push_back(static_cast<int32_t>(priority));
  h->p_queue_ts.push_back(rfc3339_micros(created));
  h->p_appgroup.push_back((has_group && !group.empty()) ? h->appgroups.id(group) : -1);
  h->p_selector.push_back((has_selector && !selector.empty()) ? h->selectors.id(selector) : -1);
  h->p_ns.push_back(h->namespaces.id(ns));
  h->p_nominated.push_back(-1);  // resolved against the node snapshot in rebuild_nominated
  h->p_nominated_name.push_back(nominated);
  return true;
}

void freeze_pods(spx_ingest* h) {
  spx_pod_objects& t = h->pod_table;
  t.n_pods = static_cast<int64_t>(h->p_priority.size());
  t.ctr_ptr = h->p_ctr_ptr.data(), t.ctr_kind = h->p_kind.data();
  t.req_ptr = h->p_req_ptr.data(), t.req_res = h->p_req_res.data(), t.req_qty = h->p_req_qty.data();
  t.lim_ptr = h->p_lim_ptr.data(), t.lim_res = h->p_lim_res.data(), t.lim_qty = h->p_lim_qty.data();
  t.ovh_ptr = h->p_ovh_ptr.data(), t.ovh_res = h->p_ovh_res.data(), t.ovh_qty = h->p_ovh_qty.data();
  t.priority = h->p_priority.data(), t.queue_ts = h->p_queue_ts.data(), t.appgroup = h->p_appgroup.data();
  t.selector = h->p_selector.data(), t.ns = h->p_ns.data();
}

void refresh_classes(spx_ingest* h) {  // the resource interner may have grown
  h->flags.assign(h->res.names.size(), 0);
  for (size_t i = 0; i < h->res.names.size(); ++i)
    if (!h->res.names[i].empty()) h->flags[i] = Interner::flags(h->res.names[i]);
  h->classes.n_res = static_cast<int32_t>(h->flags.size());
  h->classes.flags = h->flags.data();
}

template <typename Fn>
int run_decoder(spx_ingest* h, const char* json, int64_t len, int64_t* n_out, Fn&& one) {
  h->err.clear();
  h->unknown = 0;
  Reader r{json, json + len, {}, {}};
  int64_t n = 0;
  bool ok = for_each_object(r, [&] {
    ++n;
    return one(r);
  });
  if (ok && r.peek() != '\0') ok = r.fail("trailing characters");
  if (!ok && h->err.empty()) h->err = "JSON: " + r.err;
  if (n_out) *n_out = n;
  return ok ? SPX_OK : SPX_ERR_ARG;
}

void rebuild_nominated(spx_ingest* h);

}  // namespace

extern "C" int spx_ingest_nodes_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_objects_out, int64_t* n_unknown_out) {
  if (!h || !json || len < 0) return SPX_ERR_ARG;
  const int rc = run_decoder(h, json, len, n_objects_out, [&](Reader& r) { return decode_node(h, r); });
  freeze_nodes(h);
  rebuild_nominated(h);  // nominatedNodeName is resolved against the node snapshot: pods may have arrived first
  refresh_classes(h);
  if (n_unknown_out) *n_unknown_out = h->unknown;
  return rc;
}

namespace {
void normalize_selectors(spx_ingest* h);
void freeze_groups(spx_ingest* h);
void rebuild_nominated(spx_ingest* h);
}  // namespace

extern "C" int spx_ingest_pods_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_objects_out) {
  if (!h || !json || len < 0) return SPX_ERR_ARG;
  // a failed document must not leave half a pod behind: remember the sizes and roll back
  const size_t n_pods = h->p_priority.size(), n_ctr = h->p_kind.size(), n_req = h->p_req_res.size(), n_lim = h->p_lim_res.size(), n_ovh = h->p_ovh_res.size();
  const int rc = run_decoder(h, json, len, n_objects_out, [&](Reader& r) { return decode_pod(h, r); });
  if (rc != SPX_OK) {
    h->p_priority.resize(n_pods), h->p_queue_ts.resize(n_pods), h->p_appgroup.resize(n_pods), h->p_selector.resize(n_pods), h->p_ns.resize(n_pods);
    h->p_nominated.resize(n_pods), h->p_nominated_name.resize(n_pods);
    h->p_ctr_ptr.resize(n_pods + 1), h->p_ovh_ptr.resize(n_pods + 1), h->p_kind.resize(n_ctr), h->p_req_ptr.resize(n_ctr + 1), h->p_lim_ptr.resize(n_ctr + 1);
    h->p_req_res.resize(n_req), h->p_req_qty.resize(n_req), h->p_lim_res.resize(n_lim), h->p_lim_qty.resize(n_lim);
    h->p_ovh_res.resize(n_ovh), h->p_ovh_qty.resize(n_ovh);
  }
  normalize_selectors(h);  // pod labels may have introduced selectors (first-seen ids) and AppGroup names
  freeze_groups(h);
  freeze_pods(h);
  rebuild_nominated(h);
  refresh_classes(h);
  return rc;
}

extern "C" int spx_ingest_pods_reset(spx_ingest* h) {
  if (!h) return SPX_ERR_ARG;
  h->p_ctr_ptr.assign(1, 0), h->p_req_ptr.assign(1, 0), h->p_lim_ptr.assign(1, 0), h->p_ovh_ptr.assign(1, 0);
  h->p_kind.clear(), h->p_req_res.clear(), h->p_req_qty.clear(), h->p_lim_res.clear(), h->p_lim_qty.clear(), h->p_ovh_res.clear(), h->p_ovh_qty.clear();
  h->p_priority.clear(), h->p_queue_ts.clear(), h->p_appgroup.clear(), h->p_selector.clear(), h->p_ns.clear();
  h->p_nominated.clear(), h->p_nominated_name.clear();
  freeze_pods(h);
  rebuild_nominated(h);
  return SPX_OK;
}

extern "C" const spx_node_objects* spx_ingest_node_objects(const spx_ingest* h) { return h ? &h->node_table : nullptr; }
extern "C" const spx_pod_objects* spx_ingest_pod_objects(const spx_ingest* h) { return h ? &h->pod_table : nullptr; }

// kind: 0 region, 1 zone, 2 namespace, 3 AppGroup name, 4 workload selector
extern "C" int spx_ingest_seed_names(spx_ingest* h, int32_t kind, const char* const* names, int32_t n) {
  if (!h || kind < 0 || kind > 4 || (n > 0 && !names)) return SPX_ERR_ARG;
  spx_ingest::Names* t[] = {&h->regions, &h->zones, &h->namespaces, &h->appgroups, &h->selectors};
  for (int32_t i = 0; i < n; ++i)
    if (names[i]) t[kind]->id(names[i]);
  return SPX_OK;
}
extern "C" int32_t spx_ingest_name_id(const spx_ingest* h, int32_t kind, const char* name) {
  if (!h || kind < 0 || kind > 4 || !name) return -1;
  const spx_ingest::Names* t[] = {&h->regions, &h->zones, &h->namespaces, &h->appgroups, &h->selectors};
  return t[kind]->find(name);
}


// ====================================================================== AppGroup / NetworkTopology CRs (network-aware plugins)
// Schemas: manifests/crds/appgroup.diktyo.x-k8s.io_appgroups.yaml, networktopology.diktyo.x-k8s.io_networktopologies.yaml.
namespace {

// {"workload": {"selector": "..."}} -> selector
bool workload_selector(Reader& r, std::string& out, bool* found) {
  return r.object([&](const std::string& k) {
    if (k == "selector" && r.peek() == '"') return *found = true, r.str(out);
    return r.skip();
  });
}

bool decode_appgroup(spx_ingest* h, Reader& r, GroupRow* g) {
  std::string buf;
  return r.object([&](const std::string& k) {
    if (k == "metadata" && r.peek() == '{') {
      return r.object([&](const std::string& mk) { return (mk == "name" && r.peek() == '"') ? r.str(g->name) : r.skip(); });
    }
    if (k == "spec" && r.peek() == '{') {
      return r.object([&](const std::string& sk) {
        if (sk != "workloads" || r.peek() != '[') return r.skip();
        return r.array([&] {
          WorkloadRow w;
          bool has_sel = false;
          if (!r.object([&](const std::string& wk) {
                if (wk == "workload" && r.peek() == '{') return workload_selector(r, w.selector, &has_sel);
                if (wk == "dependencies" && r.peek() == '[') {
                  return r.array([&] {
                    DepRow d;
                    bool dsel = false;
                    if (!r.object([&](const std::string& dk) {
                          if (dk == "workload" && r.peek() == '{') return workload_selector(r, d.selector, &dsel);
                          if (dk == "maxNetworkCost" && r.peek() != 'n') return r.scalar(buf) && (canonical_quantity(buf, false, &d.max_cost) || r.fail("bad maxNetworkCost"));
                          return r.skip();
                        }))
                      return false;
                    if (!dsel) return h->err = "AppGroup dependency without workload.selector", false;
                    w.deps.push_back(std::move(d));
                    return true;
                  });
                }
                return r.skip();
              }))
            return false;
          if (!has_sel) return h->err = "AppGroup workload without workload.selector", false;
          g->workloads.push_back(std::move(w));
          return true;
        });
      });
    }
    if (k == "status" && r.peek() == '{') {
      return r.object([&](const std::string& sk) {
        if (sk != "topologyOrder" || r.peek() != '[') return r.skip();
        return r.array([&] {
          std::string sel;
          int64_t index = 0;
          bool has_sel = false;
          if (!r.object([&](const std::string& tk) {
                if (tk == "workload" && r.peek() == '{') return workload_selector(r, sel, &has_sel);
                if (tk == "index" && r.peek() != 'n') return r.scalar(buf) && (canonical_quantity(buf, false, &index) || r.fail("bad topology index"));
                return r.skip();
              }))
            return false;
          if (!has_sel) return h->err = "topologyOrder entry without workload.selector", false;
          g->topo.emplace_back(std::move(sel), index);
          return true;
        });
      });
    }
    return r.skip();
  });
}


// Selector ids must preserve the lexicographic order of the selector strings (FindPodOrder binary-searches TopologyOrder with
// string comparisons, util.go:138-153).  Pods intern selectors first-seen and AppGroups may arrive in any order, so the table
// is re-sorted whenever it is out of order and the ids already handed out — the pod table's selector column — are remapped.
void normalize_selectors(spx_ingest* h) {
  std::vector<std::string>& names = h->selectors.names;
  if (std::is_sorted(names.begin(), names.end())) return;
  std::vector<int32_t> order(names.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int32_t>(i);
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return names[static_cast<size_t>(a)] < names[static_cast<size_t>(b)]; });
  std::vector<int32_t> remap(names.size());
  std::vector<std::string> sorted(names.size());
  for (size_t i = 0; i < order.size(); ++i) {
    remap[static_cast<size_t>(order[i])] = static_cast<int32_t>(i);
    sorted[i] = names[static_cast<size_t>(order[i])];
  }
  names = std::move(sorted);
  h->selectors.ids.clear();
  for (size_t i = 0; i < names.size(); ++i) h->selectors.ids.emplace(names[i], static_cast<int32_t>(i));
  for (int32_t& v : h->p_selector)
    if (v >= 0) v = remap[static_cast<size_t>(v)];
}

// group rows (indexed by AppGroup name id) -> the CSR columns of spx_appgroup_objects
void freeze_groups(spx_ingest* h) {
  h->group_rows.resize(h->appgroups.names.size());  // groups only named by pod labels so far: no workloads
  h->g_wl_ptr.assign(1, 0), h->g_dep_ptr.assign(1, 0), h->g_topo_ptr.assign(1, 0);
  h->g_wl_selector.clear(), h->g_dep_selector.clear(), h->g_dep_max_cost.clear(), h->g_topo_selector.clear(), h->g_topo_index.clear();
  for (const GroupRow& g : h->group_rows) {
    for (const WorkloadRow& w : g.workloads) {
      h->g_wl_selector.push_back(h->selectors.find(w.selector.c_str()));
      for (const DepRow& d : w.deps) h->g_dep_selector.push_back(h->selectors.find(d.selector.c_str())), h->g_dep_max_cost.push_back(d.max_cost);
      h->g_dep_ptr.push_back(static_cast<int32_t>(h->g_dep_selector.size()));
    }
    h->g_wl_ptr.push_back(static_cast<int32_t>(h->g_wl_selector.size()));
    for (const auto& t : g.topo) h->g_topo_selector.push_back(h->selectors.find(t.first.c_str())), h->g_topo_index.push_back(static_cast<int32_t>(t.second));
    h->g_topo_ptr.push_back(static_cast<int32_t>(h->g_topo_selector.size()));
  }
  spx_appgroup_objects& t = h->group_table;
  t.n_groups = static_cast<int32_t>(h->g_wl_ptr.size() - 1);
  t.wl_ptr = h->g_wl_ptr.data(), t.wl_selector = h->g_wl_selector.data();
  t.dep_ptr = h->g_dep_ptr.data(), t.dep_selector = h->g_dep_selector.data(), t.dep_max_cost = h->g_dep_max_cost.data();
  t.topo_ptr = h->g_topo_ptr.data(), t.topo_selector = h->g_topo_selector.data(), t.topo_index = h->g_topo_index.data();
  h->g_placed_ptr.assign(static_cast<size_t>(t.n_groups) + 1, 0);  // the scheduled list comes from the pod lister, not from the CR
  t.placed_ptr = h->g_placed_ptr.data(), t.placed_selector = nullptr, t.placed_node = nullptr;
}

void freeze_nettopo(spx_ingest* h) {
  auto flat = [](const std::vector<std::vector<std::pair<int32_t, int64_t>>>& per, size_t n, std::vector<int32_t>& ptr, std::vector<int32_t>& dest,
                 std::vector<int64_t>& cost) {
    ptr.assign(1, 0), dest.clear(), cost.clear();
    for (size_t o = 0; o < n; ++o) {
      if (o < per.size())
        for (const auto& e : per[o]) dest.push_back(e.first), cost.push_back(e.second);
      ptr.push_back(static_cast<int32_t>(dest.size()));
    }
  };
  spx_nettopo_objects& t = h->nettopo_table;
  t.n_regions = static_cast<int32_t>(h->regions.names.size());
  t.n_zones = static_cast<int32_t>(h->zones.names.size());
  flat(h->nt_region, h->regions.names.size(), h->nt_rc_ptr, h->nt_rc_dest, h->nt_rc_cost);
  flat(h->nt_zone, h->zones.names.size(), h->nt_zc_ptr, h->nt_zc_dest, h->nt_zc_cost);
  t.rc_ptr = h->nt_rc_ptr.data(), t.rc_dest = h->nt_rc_dest.data(), t.rc_cost = h->nt_rc_cost.data();
  t.zc_ptr = h->nt_zc_ptr.data(), t.zc_dest = h->nt_zc_dest.data(), t.zc_cost = h->nt_zc_cost.data();
}

}  // namespace

// AppGroup CRs -> spx_appgroup_objects (group id = position in the documents fed so far; also interned under kind 3).  The
// selectors of all groups of one call are interned in lexicographic order when the selector table is still empty; otherwise
// they must already be known (seeded by the caller in lexicographic order) — FindPodOrder compares selector strings.
extern "C" int spx_ingest_appgroups_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_objects_out) {
  if (!h || !json || len < 0) return SPX_ERR_ARG;
  std::vector<GroupRow> groups;
  const int rc = run_decoder(h, json, len, n_objects_out, [&](Reader& r) {
    groups.emplace_back();
    return decode_appgroup(h, r, &groups.back());
  });
  if (rc != SPX_OK) return rc;
  for (GroupRow& g : groups) {  // a group's row sits at its name id, whatever order CRs and pods arrive in; a later CR replaces an earlier one
    const int32_t id = h->appgroups.id(g.name);
    if (h->group_rows.size() <= static_cast<size_t>(id)) h->group_rows.resize(static_cast<size_t>(id) + 1);
    for (const WorkloadRow& w : g.workloads) {
      h->selectors.id(w.selector);
      for (const DepRow& d : w.deps) h->selectors.id(d.selector);
    }
    for (const auto& t : g.topo) h->selectors.id(t.first);
    h->group_rows[static_cast<size_t>(id)] = std::move(g);
  }
  normalize_selectors(h);
  freeze_groups(h);
  freeze_pods(h);  // the selector column may have been renumbered
  return SPX_OK;
}

extern "C" const spx_appgroup_objects* spx_ingest_appgroup_objects(const spx_ingest* h) { return h ? &h->group_table : nullptr; }

// one NetworkTopology CR -> spx_nettopo_objects for the weights set `weights_name` (NetworkOverheadArgs.WeightsName,
// populateCostMap networkoverhead.go:448-497): spec.weights[name == weights_name].topologyList[topologyKey == region | zone label]
// .originList[].{origin, costList[].{destination, networkCost}}.  Region / zone names share the id spaces of the node table.
extern "C" int spx_ingest_nettopo_json(spx_ingest* h, const char* json, int64_t len, const char* weights_name) {
  if (!h || !json || len < 0 || !weights_name) return SPX_ERR_ARG;
  h->nt_region.clear(), h->nt_zone.clear();
  std::string buf, wname, key, origin, dest;
  const int rc = run_decoder(h, json, len, nullptr, [&](Reader& r) {
    return r.object([&](const std::string& k) {
      if (k != "spec" || r.peek() != '{') return r.skip();
      return r.object([&](const std::string& sk) {
        if (sk != "weights" || r.peek() != '[') return r.skip();
        return r.array([&] {
          // the weights entry may list its name after its topologyList: decode into a scratch and intern names only on a match
          using Origins = std::vector<std::pair<std::string, std::vector<std::pair<std::string, int64_t>>>>;
          std::vector<std::pair<std::string, Origins>> topo;  // the entry's topologyList, in document order
          wname.clear();
          if (!r.object([&](const std::string& wk) {
                if (wk == "name" && r.peek() == '"') return r.str(wname);
                if (wk != "topologyList" || r.peek() != '[') return r.skip();
                return r.array([&] {
                  key.clear();
                  Origins origins;
                  if (!r.object([&](const std::string& tk) {
                        if (tk == "topologyKey" && r.peek() == '"') return r.str(key);
                        if (tk != "originList" || r.peek() != '[') return r.skip();
                        return r.array([&] {
                          origins.emplace_back();
                          auto& o = origins.back();
                          return r.object([&](const std::string& ok) {
                            if (ok == "origin" && r.peek() == '"') return r.str(o.first);
                            if (ok != "costList" || r.peek() != '[') return r.skip();
                            return r.array([&] {
                              dest.clear();
                              int64_t cost = 0;
                              if (!r.object([&](const std::string& ck) {
                                    if (ck == "destination" && r.peek() == '"') return r.str(dest);
                                    if (ck == "networkCost" && r.peek() != 'n') return r.scalar(buf) && (canonical_quantity(buf, false, &cost) || r.fail("bad networkCost"));
                                    return r.skip();
                                  }))
                                return false;
                              o.second.emplace_back(dest, cost);
                              return true;
                            });
                          });
                        });
                      }))
                    return false;
                  topo.emplace_back(key, std::move(origins));
                  return true;
                });
              }))
            return false;
          if (wname != weights_name) return true;
          // The reference does not look names up in a map: it binary-searches the lists (util.FindTopologyKey / FindOriginCosts,
          // util.go:156-191), after sorting them only when the weights are not the controller's ("NetperfCosts":
          // sortNetworkTopologyCosts networkoverhead.go:438-445, ByOrigin :462-465).  So a duplicated key or origin is found through
          // whichever entry the search lands on (the reference's own fixture lists "topology.kubernetes.io/region" twice,
          // networkoverhead_test.go:93,127: the search finds the first and never sees zones), and an unsorted NetperfCosts list
          // misses entries.  Reproduced here: the same search over the same order.  Go's sort.Sort is an insertion sort — stable —
          // up to 12 elements; beyond that its order among equal keys is unspecified, and such input is refused.
          const bool manual = std::strcmp(weights_name, "NetperfCosts") != 0;
          bool ambiguous = false;
          auto go_sort = [&](auto& list) {  // sort.Sort by .first
            if (!manual) return;
            std::stable_sort(list.begin(), list.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            if (list.size() > 12)
              for (size_t i = 1; i < list.size(); ++i) ambiguous |= list[i].first == list[i - 1].first;
          };
          auto go_find = [](const auto& list, const std::string& want) -> int {  // the loop of FindTopologyKey / FindOriginCosts
            int low = 0, high = static_cast<int>(list.size()) - 1;
            while (low <= high) {
              const int mid = (low + high) / 2;
              if (list[static_cast<size_t>(mid)].first == want) return mid;
              if (list[static_cast<size_t>(mid)].first < want) low = mid + 1;
              else high = mid - 1;
            }
            return -1;
          };
          for (const auto& t : topo) {  // name ids follow the document (first-seen order, as the object-table builders intern them)
            spx_ingest::Names* names = t.first == "topology.kubernetes.io/region" ? &h->regions : (t.first == "topology.kubernetes.io/zone" ? &h->zones : nullptr);
            if (!names) continue;
            for (const auto& o : t.second) {
              names->id(o.first);
              for (const auto& c : o.second) names->id(c.first);
            }
          }
          go_sort(topo);
          auto take = [&](const char* label, spx_ingest::Names& names, std::vector<std::vector<std::pair<int32_t, int64_t>>>& out) {
            const int at = go_find(topo, label);
            if (at < 0) return;
            Origins origins = topo[static_cast<size_t>(at)].second;
            go_sort(origins);
            for (const auto& o : origins) {  // every origin name a node may carry: what the search returns for it
              const int oi = go_find(origins, o.first);
              if (oi < 0 || &origins[static_cast<size_t>(oi)] != &o) continue;  // not reachable (or reached through its twin: once)
              const size_t oid = static_cast<size_t>(names.id(o.first));
              if (out.size() <= oid) out.resize(oid + 1);
              for (const auto& c : o.second) out[oid].emplace_back(names.id(c.first), c.second);  // costMap assignment: the last one wins
            }
          };
          take("topology.kubernetes.io/region", h->regions, h->nt_region);
          take("topology.kubernetes.io/zone", h->zones, h->nt_zone);
          if (ambiguous) return r.fail("NetworkTopology: a list of more than 12 entries repeats a key; the reference's pick depends on sort.Sort's unspecified order");
          return true;
        });
      });
    });
  });
  freeze_nettopo(h);
  return rc;
}

extern "C" const spx_nettopo_objects* spx_ingest_nettopo_objects(const spx_ingest* h) { return h ? &h->nettopo_table : nullptr; }


// ====================================================================== load-watcher response (trimaran Collector)
// What Collector.updateMetrics stores (collector.go:139-150): the JSON the load-watcher service returns, decoded into
// watcher.WatcherMetrics.  That struct lives in github.com/paypal/load-watcher v0.2.4 (pkg/watcher/api.go), which is NOT vendored
// under the reference; the reference's tests only round-trip the Go struct through encoding/json.  The field names below are the
// struct's json tags as published (timestamp, window{duration,start,end}, source, data{NodeMetricsMap{<node>{metrics[{name,type,
// operator,rollup,value}],tags,metadata}}}) — PARITY UNPINNED: no golden document exists in the reference.  Decoding follows
// encoding/json: member names match case-insensitively, unknown members are ignored, null leaves the zero value.  So a document
// without data.NodeMetricsMap (e.g. the draft payload in kep/61-Trimaran-real-load-aware-scheduling/README.md:301-352, which
// lists nodes directly under "data") yields a nil map: "Metrics not available from watcher" (collector.go:113-116).
// Metric.Type / Operator are compared by identity with the load-watcher constants (targetloadpacking.go:134-135,
// resourcestats.go:94-103): "CPU", "Memory"; "AVG", "STD", "Latest", "" — anything else is SPX_MT_OTHER / SPX_MO_OTHER.
namespace {

bool ieq(const std::string& a, const char* b) {
  size_t i = 0;
  for (; i < a.size() && b[i]; ++i)
    if (std::tolower(static_cast<unsigned char>(a[i])) != std::tolower(static_cast<unsigned char>(b[i]))) return false;
  return i == a.size() && !b[i];
}

struct MetricRow {
  uint8_t type = SPX_MT_OTHER, op = SPX_MO_EMPTY;
  double value = 0.0;
};

bool decode_metric(Reader& r, MetricRow* m, std::string& buf) {
  return r.object([&](const std::string& k) {
    if (ieq(k, "type") && r.peek() == '"') {
      if (!r.str(buf)) return false;
      m->type = buf == "CPU" ? SPX_MT_CPU : (buf == "Memory" ? SPX_MT_MEMORY : SPX_MT_OTHER);
      return true;
    }
    if (ieq(k, "operator") && r.peek() == '"') {
      if (!r.str(buf)) return false;
      m->op = buf == "AVG" ? SPX_MO_AVG : (buf == "STD" ? SPX_MO_STD : (buf == "Latest" ? SPX_MO_LATEST : (buf.empty() ? SPX_MO_EMPTY : SPX_MO_OTHER)));
      return true;
    }
    if (ieq(k, "value") && r.peek() != 'n' && r.peek() != '"') {
      if (!r.scalar(buf)) return false;
      char* endp = nullptr;
      m->value = std::strtod(buf.c_str(), &endp);
      return (endp && *endp == '\0') || r.fail("bad metric value");
    }
    return r.skip();
  });
}

}  // namespace

extern "C" int spx_ingest_metrics_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_nodes_out, int64_t* n_unknown_out) {
  if (!h || !json || len < 0) return SPX_ERR_ARG;
  h->err.clear();
  const size_t N = h->rows.size();
  std::vector<uint8_t> present(N, 0), isnil(N, 0);
  std::vector<std::vector<MetricRow>> lists(N);
  bool map_nil = true;
  int64_t window_end = 0, n_nodes = 0, unknown = 0;
  std::string buf;
  Reader r{json, json + len, {}, {}};
  bool ok = r.object([&](const std::string& k) {
    if (ieq(k, "window") && r.peek() == '{')
      return r.object([&](const std::string& wk) {
        if (ieq(wk, "end") && r.peek() != 'n' && r.peek() != '"') return r.scalar(buf) && (canonical_quantity(buf, false, &window_end) || r.fail("bad window.end"));
        return r.skip();
      });
    if (ieq(k, "data") && r.peek() == '{')
      return r.object([&](const std::string& dk) {
        if (!ieq(dk, "NodeMetricsMap") || r.peek() != '{') return r.skip();  // null / absent: the map stays nil
        map_nil = false;
        return r.object([&](const std::string& node) {
          ++n_nodes;
          const auto it = h->node_index.find(node);
          if (it == h->node_index.end()) {
            ++unknown;
            return r.skip();
          }
          const size_t ni = static_cast<size_t>(it->second);
          present[ni] = 1, isnil[ni] = 1, lists[ni].clear();  // a repeated key replaces the earlier entry (Go map assignment)
          if (r.peek() != '{') return r.skip();
          return r.object([&](const std::string& nk) {
            if (!ieq(nk, "metrics") || r.peek() != '[') return r.skip();  // null: Metrics stays a nil slice
            isnil[ni] = 0;
            return r.array([&] {
              MetricRow m;
              if (r.peek() != '{') return r.fail("metric is not an object");
              if (!decode_metric(r, &m, buf)) return false;
              lists[ni].push_back(m);
              return true;
            });
          });
        });
      });
    return r.skip();
  });
  if (ok && r.peek() != '\0') ok = r.fail("trailing characters");
  if (n_nodes_out) *n_nodes_out = n_nodes;
  if (n_unknown_out) *n_unknown_out = unknown;
  if (!ok) return h->err = "JSON: " + r.err, SPX_ERR_ARG;
  // a successful fetch replaces the whole snapshot (collector.metrics = *metrics); a failed one keeps the last (:140-150)
  h->m_present = std::move(present), h->m_nil = std::move(isnil);
  h->m_ptr.assign(1, 0), h->m_type.clear(), h->m_op.clear(), h->m_value.clear();
  for (size_t i = 0; i < N; ++i) {
    for (const MetricRow& m : lists[i]) h->m_type.push_back(m.type), h->m_op.push_back(m.op), h->m_value.push_back(m.value);
    h->m_ptr.push_back(static_cast<int32_t>(h->m_type.size()));
  }
  spx_metrics_objects& t = h->metrics_table;
  t.map_is_nil = map_nil ? 1 : 0;
  t.window_end = window_end;
  t.node_present = h->m_present.data(), t.node_metrics_nil = h->m_nil.data();
  t.m_ptr = h->m_ptr.data(), t.m_type = h->m_type.data(), t.m_op = h->m_op.data(), t.m_value = h->m_value.data();
  h->metrics_valid = true;
  return SPX_OK;
}

extern "C" const spx_metrics_objects* spx_ingest_metrics_objects(const spx_ingest* h) { return (h && h->metrics_valid) ? &h->metrics_table : nullptr; }

// ====================================================================== ElasticQuota CRs (CapacityScheduling)
// Schema: manifests/crds/scheduling.x-k8s.io_elasticquotas.yaml (spec.min, spec.max, status.used: ResourceLists; one quota per
// namespace).  Semantics as newElasticQuotaInfo / framework.NewResource read them (elasticquota.go:70-87): a nil Min becomes the
// zero bound, a nil Max the upper bound (math.MaxInt64 for cpu / memory / ephemeral-storage); cpu in millicores, memory,
// ephemeral-storage, pods, and scalar resources by slot.  `used` is taken from status.used — upstream keeps the same sum in
// memory from pod events; the nominated-pod list is scheduling-queue state and stays with the caller (empty here).
struct spx_ingest_quota {
  std::vector<int32_t> scalar_res;  // canonical id per scalar slot
  std::vector<uint8_t> has, min_p, max_p, used_p;
  std::vector<int64_t> min, max, used;
  spx_pod_objects no_pods{};
  int32_t zero_ptr[2] = {0, 0};
  spx_quota_objects table{};
  // nominated pods = the pending pods that carry status.nominatedNodeName for a node of the snapshot (what
  // PodNominator.NominatedPodsForNode yields over the node list, capacity_scheduling.go:231-253), as a compact pod table
  std::vector<int32_t> nom_ns, nom_priority, n_ctr_ptr, n_req_ptr, n_lim_ptr, n_ovh_ptr, n_req_res, n_lim_res, n_ovh_res;
  std::vector<int64_t> nom_pending, n_req_qty, n_lim_qty, n_ovh_qty;
  std::vector<uint8_t> n_kind;
  spx_pod_objects nom_pods{};
};

namespace {

struct QuotaRow {
  bool seen = false;
  bool has_min = false, has_max = false;
  std::vector<std::pair<std::string, int64_t>> min, max, used;
};

bool quota_vector(spx_ingest* h, spx_ingest_quota* q, const std::vector<std::pair<std::string, int64_t>>& rl, int64_t* v, uint8_t* present) {
  for (int i = 0; i < SPX_QUOTA_SLOTS; ++i) v[i] = 0;
  *present = 0;
  for (const auto& e : rl) {  // framework.Resource.Add
    if (e.first == "cpu") v[0] += e.second;
    else if (e.first == "memory") v[1] += e.second;
    else if (e.first == "ephemeral-storage") v[2] += e.second;
    else if (e.first == "pods") v[3] += e.second;
    else if (scalar_resource_name(e.first)) {
      const int32_t rid = h->res.id(e.first);
      size_t s = 0;
      while (s < q->scalar_res.size() && q->scalar_res[s] != rid) ++s;
      if (s == q->scalar_res.size()) {
        if (s >= SPX_QUOTA_SLOTS - 4) return h->err = "more scalar resources than quota slots", false;
        q->scalar_res.push_back(rid);
      }
      v[4 + s] += e.second;
      *present = static_cast<uint8_t>(*present | (1u << (4 + s)));
    }
  }
  return true;
}

}  // namespace

// namespaces fixes the namespace ids (= spx_pod_objects.ns; kind 2 of the name tables is seeded with them)
extern "C" int spx_ingest_quota_json(spx_ingest* h, const char* json, int64_t len, const char* const* namespaces, int32_t n_namespaces,
                                     int64_t* n_objects_out, int64_t* n_unknown_out) {
  if (!h || !json || len < 0 || n_namespaces <= 0 || !namespaces) return SPX_ERR_ARG;
  for (int32_t i = 0; i < n_namespaces; ++i) {
    if (!namespaces[i]) return SPX_ERR_ARG;
    h->namespaces.id(namespaces[i]);
  }
  std::vector<QuotaRow> rows(static_cast<size_t>(n_namespaces));
  std::unordered_map<std::string, int32_t> index;
  for (int32_t i = 0; i < n_namespaces; ++i) index.emplace(namespaces[i], i);
  std::string buf, ns;
  int64_t unknown = 0;
  const int rc = run_decoder(h, json, len, n_objects_out, [&](Reader& r) {
    QuotaRow row;
    row.seen = true;
    ns.clear();
    auto list = [&](std::vector<std::pair<std::string, int64_t>>* out, bool* present) {
      if (r.peek() == 'n') return r.skip();  // null: a nil list
      if (present) *present = true;
      return resource_list(h, r, buf, [&](const std::string& rn, int64_t q) { out->emplace_back(rn, q); });
    };
    if (!r.object([&](const std::string& k) {
          if (k == "metadata" && r.peek() == '{')
            return r.object([&](const std::string& mk) { return (mk == "namespace" && r.peek() == '"') ? r.str(ns) : r.skip(); });
          if (k == "spec" && r.peek() == '{')
            return r.object([&](const std::string& sk) {
              if (sk == "min") return list(&row.min, &row.has_min);
              if (sk == "max") return list(&row.max, &row.has_max);
              return r.skip();
            });
          if (k == "status" && r.peek() == '{')
            return r.object([&](const std::string& sk) { return sk == "used" ? list(&row.used, nullptr) : r.skip(); });
          return r.skip();
        }))
      return false;
    auto it = index.find(ns);
    if (it == index.end()) {
      ++unknown;
      return true;
    }
    rows[static_cast<size_t>(it->second)] = std::move(row);  // "Each namespace can only have one ElasticQuota": the last one wins
    return true;
  });
  if (n_unknown_out) *n_unknown_out = unknown;
  if (rc != SPX_OK) return rc;
  if (!h->quota) h->quota = new spx_ingest_quota();
  spx_ingest_quota* q = h->quota;
  q->scalar_res.clear();
  const size_t n = rows.size();
  q->has.assign(n, 0), q->min_p.assign(n, 0), q->max_p.assign(n, 0), q->used_p.assign(n, 0);
  q->min.assign(n * SPX_QUOTA_SLOTS, 0), q->max.assign(n * SPX_QUOTA_SLOTS, 0), q->used.assign(n * SPX_QUOTA_SLOTS, 0);
  for (size_t i = 0; i < n; ++i) {
    const QuotaRow& row = rows[i];
    q->has[i] = row.seen;
    // a namespace without a quota keeps the bounds a nil list would get (the flattener only looks at has_quota there)
    if (!quota_vector(h, q, row.has_min ? row.min : std::vector<std::pair<std::string, int64_t>>{}, &q->min[i * SPX_QUOTA_SLOTS], &q->min_p[i])) return SPX_ERR_ARG;
    if (row.has_max) {
      if (!quota_vector(h, q, row.max, &q->max[i * SPX_QUOTA_SLOTS], &q->max_p[i])) return SPX_ERR_ARG;
    } else {
      q->max[i * SPX_QUOTA_SLOTS + 0] = q->max[i * SPX_QUOTA_SLOTS + 1] = q->max[i * SPX_QUOTA_SLOTS + 2] = INT64_MAX;  // UpperBoundOfMax elasticquota.go:29
    }
    if (!quota_vector(h, q, row.used, &q->used[i * SPX_QUOTA_SLOTS], &q->used_p[i])) return SPX_ERR_ARG;
  }
  const int32_t slots = static_cast<int32_t>(q->scalar_res.size());
  q->scalar_res.resize(SPX_QUOTA_SLOTS - 4, 0);
  spx_quota_objects& t = q->table;
  t.n_namespaces = n_namespaces;
  t.n_scalar_slots = slots;
  t.scalar_res = q->scalar_res.data(), t.has_quota = q->has.data();
  t.min = q->min.data(), t.min_present = q->min_p.data(), t.max = q->max.data(), t.max_present = q->max_p.data();
  t.used = q->used.data(), t.used_present = q->used_p.data();
  t.n_nominated = 0, t.nom_ns = nullptr, t.nom_priority = nullptr, t.nom_pending_index = nullptr;
  q->no_pods = spx_pod_objects{};
  q->no_pods.ctr_ptr = q->no_pods.ovh_ptr = q->no_pods.req_ptr = q->no_pods.lim_ptr = q->zero_ptr;
  t.nom_pods = &q->no_pods;
  rebuild_nominated(h);
  refresh_classes(h);
  return SPX_OK;
}

extern "C" const spx_quota_objects* spx_ingest_quota_objects(const spx_ingest* h) { return (h && h->quota) ? &h->quota->table : nullptr; }

namespace {
void rebuild_nominated(spx_ingest* h) {
  // only pods nominated to a node of the snapshot count (capacity_scheduling.go:231-253 walks the snapshot's nodes and asks
  // NominatedPodsForNode for each): resolve the names against the node table as it is NOW
  for (size_t i = 0; i < h->p_nominated_name.size(); ++i) {
    const std::string& nm = h->p_nominated_name[i];
    const auto it = nm.empty() ? h->node_index.end() : h->node_index.find(nm);
    h->p_nominated[i] = it == h->node_index.end() ? -1 : static_cast<int32_t>(it->second);
  }
  spx_ingest_quota* q = h->quota;
  if (!q) return;  // no quota table yet: spx_ingest_quota_json calls this again
  q->nom_ns.clear(), q->nom_priority.clear(), q->nom_pending.clear();
  q->n_ctr_ptr.assign(1, 0), q->n_req_ptr.assign(1, 0), q->n_lim_ptr.assign(1, 0), q->n_ovh_ptr.assign(1, 0);
  q->n_kind.clear(), q->n_req_res.clear(), q->n_req_qty.clear(), q->n_lim_res.clear(), q->n_lim_qty.clear(), q->n_ovh_res.clear(), q->n_ovh_qty.clear();
  const size_t P = h->p_nominated.size();
  for (size_t i = 0; i < P; ++i) {
    if (h->p_nominated[i] < 0) continue;
    q->nom_ns.push_back(h->p_ns[i]);
    q->nom_priority.push_back(h->p_priority[i]);
    q->nom_pending.push_back(static_cast<int64_t>(i));  // the same pod object as pending row i: skipped for itself (p.UID == pod.UID, :236)
    for (int32_t c = h->p_ctr_ptr[i]; c < h->p_ctr_ptr[i + 1]; ++c) {
      q->n_kind.push_back(h->p_kind[static_cast<size_t>(c)]);
      for (int32_t k = h->p_req_ptr[static_cast<size_t>(c)]; k < h->p_req_ptr[static_cast<size_t>(c) + 1]; ++k)
        q->n_req_res.push_back(h->p_req_res[static_cast<size_t>(k)]), q->n_req_qty.push_back(h->p_req_qty[static_cast<size_t>(k)]);
      for (int32_t k = h->p_lim_ptr[static_cast<size_t>(c)]; k < h->p_lim_ptr[static_cast<size_t>(c) + 1]; ++k)
        q->n_lim_res.push_back(h->p_lim_res[static_cast<size_t>(k)]), q->n_lim_qty.push_back(h->p_lim_qty[static_cast<size_t>(k)]);
      q->n_req_ptr.push_back(static_cast<int32_t>(q->n_req_res.size()));
      q->n_lim_ptr.push_back(static_cast<int32_t>(q->n_lim_res.size()));
    }
    q->n_ctr_ptr.push_back(static_cast<int32_t>(q->n_kind.size()));
    for (int32_t k = h->p_ovh_ptr[i]; k < h->p_ovh_ptr[i + 1]; ++k)
      q->n_ovh_res.push_back(h->p_ovh_res[static_cast<size_t>(k)]), q->n_ovh_qty.push_back(h->p_ovh_qty[static_cast<size_t>(k)]);
    q->n_ovh_ptr.push_back(static_cast<int32_t>(q->n_ovh_res.size()));
  }
  spx_pod_objects& t = q->nom_pods;
  t = spx_pod_objects{};
  t.n_pods = static_cast<int64_t>(q->nom_ns.size());
  t.ctr_ptr = q->n_ctr_ptr.data(), t.ctr_kind = q->n_kind.data();
  t.req_ptr = q->n_req_ptr.data(), t.req_res = q->n_req_res.data(), t.req_qty = q->n_req_qty.data();
  t.lim_ptr = q->n_lim_ptr.data(), t.lim_res = q->n_lim_res.data(), t.lim_qty = q->n_lim_qty.data();
  t.ovh_ptr = q->n_ovh_ptr.data(), t.ovh_res = q->n_ovh_res.data(), t.ovh_qty = q->n_ovh_qty.data();
  t.priority = q->nom_priority.data(), t.ns = q->nom_ns.data();
  spx_quota_objects& qt = q->table;
  qt.n_nominated = t.n_pods;
  qt.nom_ns = q->nom_ns.data(), qt.nom_priority = q->nom_priority.data(), qt.nom_pending_index = q->nom_pending.data();
  qt.nom_pods = t.n_pods ? &q->nom_pods : &q->no_pods;
}
}  // namespace

static void free_quota(spx_ingest_quota* q) { delete q; }

static void freeze_nodes_initial(spx_ingest* h) {
  freeze_nodes(h);
  freeze_pods(h);
  freeze_groups(h);
  freeze_nettopo(h);
}
