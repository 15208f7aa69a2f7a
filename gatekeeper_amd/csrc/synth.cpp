// Native synthetic-workload generator (bench / test plumbing; include/gksynth.h).
//
// Produces the SAME objects as gatekeeper_amd/synth.py (SURVEY.md section 8d's configs: SplitMix64, one generator per
// object seeded from (seed, index)) as JSON text plus the gk_review_in array gk_table_create takes, so that a
// 1 000 000-object audit set (BASELINE.json configs[2]) is ready in seconds instead of the minutes the Python generator
// and ctypes marshalling would need.  tests/test_synth.py pins it object by object against synth.py.
// Review shape: AugmentedUnstructured{Object, Namespace, Source: Original} as pkg/audit builds it
// (pkg/audit/manager.go:694-713): the Namespace is looked up by the object's metadata.namespace.
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <algorithm>
#include <chrono>
#include "../../include/gkgpu.h"
#include "../../include/gksynth.h"

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) / 9007199254740992.0; }
  uint64_t below(uint64_t n) { return next() % n; }
  bool chance(double p) { return uniform() < p; }
  template <size_t N>
  const char* pick(const char* const (&a)[N]) { return a[below(N)]; }
  // index of the chosen alternative of a weighted choice
  int weighted(const double* p, int n) {
    double u = uniform(), acc = 0.0;
    for (int i = 0; i < n; i++) { acc += p[i]; if (u < acc) return i; }
    return n - 1;
  }
};

const char* const LABEL_KEYS[] = {"app", "tier", "env", "team", "owner", "release", "track", "zone", "region", "cost-center", "app.kubernetes.io/name",
                                  "app.kubernetes.io/part-of", "app.kubernetes.io/managed-by", "chart", "heritage", "component", "role", "version",
                                  "stage", "project", "squad", "domain", "criticality", "pci", "gdpr", "tenant", "cluster", "shard", "canary",
                                  "backup", "monitored", "sidecar"};
const char* const LABEL_VALUES[] = {"a", "b", "prod", "dev", "web", "db", "cache", "true", "false", "blue", "green", "v1", "v2", "core", "edge", "x"};
const char* const HOST_PATHS[] = {"/tmp", "/foo", "/foo/bar", "/var/log", "/etc"};
const char* const IMAGES[] = {"nginx", "nginx:1.25", "openpolicyagent/opa:0.9.2", "gcr.io/proj/app:latest", "quay.io/org/tool:v3", "busybox"};
const char* const CPUS[] = {"100m", "200m", "1", "2"};
const char* const MEMS[] = {"128Mi", "1Gi", "2Gi"};

std::vector<std::string> namespace_names() {
  std::vector<std::string> out;
  for (const char* s : {"system", "public", "node-lease", "proxy", "dns"}) out.push_back(std::string("kube-") + s);
  char b[32];
  for (int i = 0; i < 30; i++) { snprintf(b, sizeof b, "prod-%02d", i); out.push_back(b); }
  for (int i = 0; i < 30; i++) { snprintf(b, sizeof b, "dev-%02d", i); out.push_back(b); }
  for (int i = 0; i < 35; i++) { snprintf(b, sizeof b, "team-%02d", i); out.push_back(b); }
  return out;
}

std::string q(const std::string& s) { return "\"" + s + "\""; }   // the generator's strings need no escaping
std::string num(uint64_t v) { return std::to_string(v); }
std::string fmt(const char* f, uint64_t v) { char b[48]; snprintf(b, sizeof b, f, (unsigned long long)v); return b; }

struct Labels { std::vector<std::pair<std::string, std::string>> kv; };   // insertion-ordered, overwrite keeps the position
Labels gen_labels(Rng& r) {
  Labels l;
  uint64_t n = 2 + r.below(5);
  for (uint64_t i = 0; i < n; i++) {
    std::string v = r.pick(LABEL_VALUES), k = r.pick(LABEL_KEYS);
    bool found = false;
    for (auto& e : l.kv) if (e.first == k) { e.second = v; found = true; }
    if (!found) l.kv.emplace_back(k, v);
  }
  return l;
}
std::string labels_json(const Labels& l) {
  std::string o = "{";
  for (size_t i = 0; i < l.kv.size(); i++) { if (i) o += ", "; o += q(l.kv[i].first) + ": " + q(l.kv[i].second); }
  return o + "}";
}

// HIGH CARDINALITY (gk_synth_batch_create, mixed | 16): every container carries an image tag and a name no other container of the stream
// has (a digest-like tag, as a registry that pins by digest produces; a generated name), so that the flattener's per-value memos of the
// dictionary expressions on `image` / `name` never hit -- the low-cardinality default (a dozen images) always does.  The random stream
// that decides everything else is untouched: the objects are the default ones with longer strings in those two places.
static thread_local uint64_t g_high_card = 0;   // 0: off; else a per-object salt
std::string gen_container(Rng& r, int idx, const std::vector<std::string>& vols, bool init) {
  std::string image = r.pick(IMAGES), cname = fmt(init ? "init-%llu" : "c%llu", (uint64_t)idx);
  if (g_high_card) {
    uint64_t h = (g_high_card + (uint64_t)idx * 0x9E3779B97F4A7C15ull + (init ? 0x5851F42D4C957F2Dull : 0)) * 0xD6E8FEB86659FD93ull;
    h ^= h >> 29;
    const size_t colon = image.rfind(':');
    image = (colon == std::string::npos || image.find('/', colon) != std::string::npos ? image : image.substr(0, colon)) + fmt(":sha-%016llx", (unsigned long long)h);
    cname += fmt("-%08llx", (unsigned long long)(h >> 32));
  }
  std::string o = "{\"name\": " + q(cname) + ", \"image\": " + q(image);
  if (r.chance(0.6)) {
    std::string sc;
    auto add = [&](const std::string& kv) { if (!sc.empty()) sc += ", "; sc += kv; };
    if (r.chance(0.05 / 0.6)) add("\"privileged\": true");
    else if (r.chance(0.3)) add("\"privileged\": false");
    if (r.chance(0.3)) add("\"runAsNonRoot\": true");
    if (r.chance(0.2)) add("\"allowPrivilegeEscalation\": false");
    o += ", \"securityContext\": {" + sc + "}";
  }
  static const double pp[] = {0.4, 0.4, 0.2};
  int nports = r.weighted(pp, 3);
  if (nports) {
    o += ", \"ports\": [";
    for (int k = 0; k < nports; k++) {
      if (k) o += ", ";
      o += "{\"containerPort\": " + num(1 + r.below(65535));
      if (r.chance(0.05)) o += ", \"hostPort\": " + num(1 + r.below(65535));
      if (r.chance(0.3)) o += ", \"protocol\": \"TCP\"";
      o += "}";
    }
    o += "]";
  }
  if (!vols.empty()) {
    std::string ms;
    for (const std::string& vn : vols) {
      if (!r.chance(0.6)) continue;
      if (!ms.empty()) ms += ", ";
      ms += "{\"name\": " + q(vn) + ", \"mountPath\": " + q("/mnt/" + vn);
      if (r.chance(0.5)) ms += std::string(", \"readOnly\": ") + (r.chance(0.8) ? "true" : "false");
      ms += "}";
    }
    if (!ms.empty()) o += ", \"volumeMounts\": [" + ms + "]";
  }
  if (r.chance(0.5)) {
    std::string cpu = r.pick(CPUS), mem = r.pick(MEMS);
    o += ", \"resources\": {\"limits\": {\"cpu\": " + q(cpu) + ", \"memory\": " + q(mem) + "}}";
  }
  if (r.chance(0.3)) {
    uint64_t n = 1 + r.below(3);
    o += ", \"env\": [";
    for (uint64_t k = 0; k < n; k++) { if (k) o += ", "; o += "{\"name\": " + q(fmt("E%llu", k)) + ", \"value\": " + q(r.pick(LABEL_VALUES)) + "}"; }
    o += "]";
  }
  return o + "}";
}

std::string gen_pod_spec(Rng& r) {
  static const double pv[] = {0.4, 0.3, 0.2, 0.1};
  int nvol = r.weighted(pv, 4);
  std::vector<std::string> names;
  std::string vols;
  for (int v = 0; v < nvol; v++) {
    static const double pt[] = {0.3, 0.25, 0.25, 0.1, 0.1};
    int vt = r.weighted(pt, 5);
    std::string name = fmt("vol-%llu", (uint64_t)v);
    if (v) vols += ", ";
    vols += "{\"name\": " + q(name) + ", ";
    switch (vt) {
      case 0: vols += "\"configMap\": {\"name\": " + q(fmt("cm-%llu", r.below(20))) + "}"; break;
      case 1: vols += "\"secret\": {\"secretName\": " + q(fmt("s-%llu", r.below(20))) + "}"; break;
      case 2: vols += "\"emptyDir\": {}"; break;
      case 3: vols += "\"hostPath\": {\"path\": " + q(r.pick(HOST_PATHS)) + "}"; break;
      default: vols += "\"persistentVolumeClaim\": {\"claimName\": " + q(fmt("pvc-%llu", r.below(20))) + "}"; break;
    }
    vols += "}";
    names.push_back(name);
  }
  static const double pc[] = {0.5, 0.3, 0.15, 0.05};
  int nc = 1 + r.weighted(pc, 4);
  std::string o = "{\"containers\": [";
  for (int i = 0; i < nc; i++) { if (i) o += ", "; o += gen_container(r, i, names, false); }
  o += "]";
  if (r.chance(0.2)) o += ", \"initContainers\": [" + gen_container(r, 0, names, true) + "]";
  if (nvol) o += ", \"volumes\": [" + vols + "]";
  if (r.chance(0.03)) o += ", \"hostNetwork\": true";
  if (r.chance(0.02)) o += ", \"hostPID\": true";
  if (r.chance(0.02)) o += ", \"hostIPC\": true";
  if (r.chance(0.3)) o += ", \"serviceAccountName\": " + q(fmt("sa-%llu", r.below(10)));
  if (r.chance(0.3)) o += ", \"restartPolicy\": \"Always\"";
  return o + "}";
}

struct Pod { std::string ns, labels, spec; bool annotated; };
Pod gen_pod(Rng& r, const std::vector<std::string>& nss) {
  Pod p;
  p.ns = nss[r.below(nss.size())];
  p.labels = labels_json(gen_labels(r));
  p.annotated = r.chance(0.2);
  p.spec = gen_pod_spec(r);
  return p;
}

// -> JSON text; *ns_index = index of the object's namespace in the namespace list, or -1 (cluster-scoped)
std::string gen_object(uint64_t seed, uint64_t i, bool mixed, const std::vector<std::string>& nss, int* ns_index) {
  Rng r(Rng(seed ^ (i * 0xD1342543DE82EF95ull)).next());
  int kind = 0;
  if (mixed) { static const double pk[] = {0.8, 0.1, 0.05, 0.025, 0.025}; kind = r.weighted(pk, 5); }
  auto ns_of = [&](const std::string& name) { for (size_t k = 0; k < nss.size(); k++) if (nss[k] == name) return (int)k; return -1; };
  *ns_index = -1;
  if (kind == 0 || kind == 1) {
    Pod p = gen_pod(r, nss);
    *ns_index = ns_of(p.ns);
    if (kind == 0)
      return "{\"apiVersion\": \"v1\", \"kind\": \"Pod\", \"metadata\": {\"name\": " + q(fmt("pod-%07llu", i)) + ", \"namespace\": " + q(p.ns) + ", \"labels\": " + p.labels +
             (p.annotated ? ", \"annotations\": {\"note\": \"generated\"}" : "") + "}, \"spec\": " + p.spec + "}";
    uint64_t replicas = 1 + r.below(5);
    return "{\"apiVersion\": \"apps/v1\", \"kind\": \"Deployment\", \"metadata\": {\"name\": " + q(fmt("dep-%07llu", i)) + ", \"namespace\": " + q(p.ns) +
           ", \"labels\": " + p.labels + "}, \"spec\": {\"replicas\": " + num(replicas) + ", \"template\": {\"metadata\": {\"labels\": " + p.labels +
           "}, \"spec\": " + p.spec + "}}}";
  }
  if (kind == 2)
    return "{\"apiVersion\": \"v1\", \"kind\": \"Namespace\", \"metadata\": {\"name\": " + q(fmt("gen-ns-%07llu", i)) + ", \"labels\": " + labels_json(gen_labels(r)) + "}}";
  if (kind == 3) {
    std::string ns = nss[r.below(nss.size())];
    *ns_index = ns_of(ns);
    uint64_t port = 80 + r.below(1000);
    std::string app = r.pick(LABEL_VALUES);
    return "{\"apiVersion\": \"v1\", \"kind\": \"Service\", \"metadata\": {\"name\": " + q(fmt("svc-%07llu", i)) + ", \"namespace\": " + q(ns) +
           "}, \"spec\": {\"ports\": [{\"port\": " + num(port) + "}], \"selector\": {\"app\": " + q(app) + "}}}";
  }
  std::string ns = nss[r.below(nss.size())];
  *ns_index = ns_of(ns);
  uint64_t n = 1 + r.below(4);
  std::string data;
  for (uint64_t k = 0; k < n; k++) { if (k) data += ", "; data += q(fmt("k%llu", k)) + ": " + q(r.pick(LABEL_VALUES)); }
  return "{\"apiVersion\": \"v1\", \"kind\": \"ConfigMap\", \"metadata\": {\"name\": " + q(fmt("cm-%07llu", i)) + ", \"namespace\": " + q(ns) + "}, \"data\": {" + data + "}}";
}

}  // namespace

struct gk_synth_batch {
  std::vector<std::string> json;          // one document per object
  std::vector<std::string> ns_json;       // the Namespace objects (by namespace-list index)
  std::vector<gk_review_in> reviews;
  uint64_t json_bytes = 0;
};

extern "C" {

int gk_synth_batch_create(uint64_t seed, uint64_t start, uint64_t n, int mixed, const char* const* namespace_jsons, size_t n_namespaces,
                          gk_synth_batch** out) {
  if (!out) return GK_ERR_INVALID;
  std::vector<std::string> nss = namespace_names();
  if (namespace_jsons && n_namespaces != nss.size()) return GK_ERR_INVALID;
  gk_synth_batch* b = new gk_synth_batch();
  if (namespace_jsons) for (size_t k = 0; k < n_namespaces; k++) b->ns_json.emplace_back(namespace_jsons[k]);
  b->json.resize(n);
  b->reviews.resize(n);
  size_t n_threads = std::max<size_t>(1, std::min<size_t>(gk_host_cpus(), n / 4096 + 1));
  std::vector<int> ns_idx(n, -1);
  const bool as_request = (mixed & 2) != 0;   // Pods wrapped in the AdmissionRequest the webhook receives (CREATE)
  if (as_request && (mixed & 1)) { delete b; return GK_ERR_INVALID; }
  auto work = [&](size_t w) {
    for (uint64_t k = w; k < n; k += n_threads) {
      g_high_card = (mixed & 16) ? ((seed * 0x2545F4914F6CDD1Dull) ^ ((start + k + 1) * 0x9E3779B97F4A7C15ull)) | 1ull : 0ull;
      std::string obj = gen_object(seed, start + k, (mixed & 1) != 0, nss, &ns_idx[k]);
      g_high_card = 0;
      if (as_request) {
        const std::string name = fmt("pod-%07llu", (unsigned long long)(start + k));
        obj = "{\"uid\": " + q(fmt("uid-%llu", (unsigned long long)(start + k))) + ", \"kind\": {\"group\": \"\", \"version\": \"v1\", \"kind\": \"Pod\"}, "
              "\"resource\": {\"group\": \"\", \"version\": \"v1\", \"resource\": \"pods\"}, \"name\": " + q(name) + ", \"namespace\": " +
              q(ns_idx[k] >= 0 ? nss[ns_idx[k]] : std::string()) + ", \"operation\": \"CREATE\", \"userInfo\": {\"username\": \"system:serviceaccount:ci:deployer\", "
              "\"groups\": [\"system:serviceaccounts\", \"system:authenticated\"]}, \"object\": " + obj + ", \"oldObject\": null, \"dryRun\": false, "
              "\"options\": {\"kind\": \"CreateOptions\", \"apiVersion\": \"meta.k8s.io/v1\"}}";
      }
      b->json[k] = std::move(obj);
    }
  };
  if (n_threads <= 1) work(0);
  else {
    std::vector<std::thread> th;
    for (size_t w = 0; w < n_threads; w++) th.emplace_back(work, w);
    for (auto& t : th) t.join();
  }
  for (uint64_t k = 0; k < n; k++) {
    gk_review_in& r = b->reviews[k];
    memset(&r, 0, sizeof r);
    r.kind = as_request ? GK_REVIEW_ADMISSION_REQUEST : GK_REVIEW_OBJECT;
    r.source = GK_SRC_ORIGINAL;
    r.json = b->json[k].data();
    r.json_len = b->json[k].size();
    b->json_bytes += r.json_len;
    if (ns_idx[k] >= 0 && !b->ns_json.empty()) { r.namespace_json = b->ns_json[ns_idx[k]].data(); r.namespace_len = b->ns_json[ns_idx[k]].size(); }
  }
  *out = b;
  return GK_OK;
}

const gk_review_in* gk_synth_batch_reviews(const gk_synth_batch* b) { return b ? b->reviews.data() : nullptr; }
size_t gk_synth_batch_size(const gk_synth_batch* b) { return b ? b->reviews.size() : 0; }
uint64_t gk_synth_batch_json_bytes(const gk_synth_batch* b) { return b ? b->json_bytes : 0; }
void gk_synth_batch_free(gk_synth_batch* b) { delete b; }

int gk_synth_query_storm(gk_engine* e, const gk_synth_batch* b, uint32_t threads, uint32_t per_thread, gk_storm_out* out) {
  return gk_synth_query_storm_ex(e, b, threads, per_thread, nullptr, 0, 0, out);
}
int gk_synth_query_storm_ex(gk_engine* e, const gk_synth_batch* b, uint32_t threads, uint32_t per_thread, const uint32_t* constraint_ids, size_t n_constraints,
                            uint32_t query_flags, gk_storm_out* out) {
  if (!e || !b || !out || !threads || !per_thread || b->reviews.empty()) return GK_ERR_INVALID;
  struct PerThread { std::vector<double> lat; double batch = 0, queue = 0, device = 0; uint64_t errors = 0, results = 0; };
  std::vector<PerThread> pt(threads);
  const size_t n = b->reviews.size();
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (uint32_t w = 0; w < threads; w++)
    th.emplace_back([&, w]() {
      PerThread& me = pt[w];
      me.lat.reserve(per_thread);
      for (uint32_t k = 0; k < per_thread; k++) {
        const gk_review_in& rv = b->reviews[((size_t)w * per_thread + k) % n];
        char* js = nullptr;
        gk_query_stats st;
        memset(&st, 0, sizeof st);
        // (constraint ids: Driver.Query as the Go shim calls it -- gk_query_ex2, pre-matched when the flag says so; none: gk_query)
        const int rc = constraint_ids ? gk_query_ex2(e, &rv, constraint_ids, n_constraints, query_flags, &js, nullptr, &st) : gk_query(e, &rv, &js, &st);
        if (rc != GK_OK) { me.errors++; continue; }
        me.lat.push_back(st.total_us);
        me.batch += st.batch_size; me.queue += st.queue_us; me.device += st.device_us;
        if (js) { for (const char* p = js; (p = strstr(p, "\"constraint\"")) != nullptr; p += 12) me.results++; gk_free(js); }
      }
    });
  for (auto& x : th) x.join();
  memset(out, 0, sizeof *out);
  out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::vector<double> all;
  double batch = 0, queue = 0, device = 0;
  for (auto& p : pt) { all.insert(all.end(), p.lat.begin(), p.lat.end()); batch += p.batch; queue += p.queue; device += p.device; out->errors += p.errors; out->results += p.results; }
  out->calls = all.size();
  if (!all.empty()) {
    std::sort(all.begin(), all.end());
    auto q = [&](double f) { return all[std::min(all.size() - 1, (size_t)(f * all.size()))]; };
    out->p50_us = q(0.50); out->p90_us = q(0.90); out->p99_us = q(0.99); out->max_us = all.back();
    out->mean_batch = batch / all.size(); out->mean_queue_us = queue / all.size(); out->mean_device_us = device / all.size();
  }
  return GK_OK;
}

}  // extern "C"
