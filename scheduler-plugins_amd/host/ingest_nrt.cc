// ingest_nrt.cc — wire format -> object tables for NodeResourceTopology (first slice).
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
#include <cstdint>
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
  bool any = false, frac = false;
  for (; i < n; ++i) {
    const char c = t[i];
    if (c >= '0' && c <= '9') {
      if (mant > UINT64_MAX / 10) return false;  // more than ~19 significant digits: not a quantity this engine can hold
      mant = mant * 10 + static_cast<unsigned>(c - '0');
      if (frac) --dec;
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
  bool inexact = false;
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
  int64_t v = static_cast<int64_t>(mant);
  if (neg) v = -v;                 // ceil of a negative value truncates toward zero
  else if (inexact) v += 1;        // ceil
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

struct spx_ingest {
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

extern "C" int spx_ingest_create(const char* const* node_names, int64_t n_nodes, const char* const* resource_names, int32_t n_resource_names,
                                 spx_ingest** out) {
  if (!out || n_nodes <= 0 || !node_names) return SPX_ERR_ARG;
  auto* h = new spx_ingest();
  h->rows.resize(static_cast<size_t>(n_nodes));
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
  *out = h;
  return SPX_OK;
}

extern "C" int spx_ingest_destroy(spx_ingest* h) {
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
