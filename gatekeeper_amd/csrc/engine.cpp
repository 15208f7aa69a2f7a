// Host engine + C ABI (include/gkgpu.h).  Holds templates, constraints, the path dictionary, the Namespace cache
// and data.inventory; builds the device plan lazily; flattens reviews into HBM-resident tables; launches the HIP
// kernels through device.hpp; renders messages for the sparse violating pairs.
//
// Mirrors the state a reference drivers.Driver keeps (pkg/drivers/k8scel/driver.go:60-160) plus the target handler
// pieces that sit on the hot path (pkg/target/target.go:81-179, matcher.go:21-93, ns_cache.go:15-87).
#include <atomic>
#include <cstring>
#include <dirent.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <thread>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <sstream>

#include "../../include/gkgpu.h"
#include "codegen.hpp"
#include "device.hpp"
#include "flatten.hpp"
#include "lower.hpp"
#include "pe.hpp"
#include "regex.hpp"

using namespace gk;

static std::atomic<int> g_debug_group_max{0};   // gk_debug_set("group_max", n): plan groups of at most n constraints (0 = as many as fit)
namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

std::string lower_str(std::string s) { for (auto& c : s) if (c >= 'A' && c <= 'Z') c += 32; return s; }

struct ConstraintRec {
  uint32_t id;
  std::string kind, name;
  Value object, params, match;
  FP viol;
  MatchFormulas mf;
  std::shared_ptr<const PreparedConstraint> prep;   // simplified / folded once (lower.hpp), reused by every plan build
  // RESULT totals (gk_table_totals): "this review may yield MORE THAN ONE result" (Template::compile_multi) with the same match
  // formulas, lowered into the totals plans; null = not lowerable: every violating pair of the constraint is rendered
  std::shared_ptr<const PreparedConstraint> multi_prep;
  bool multi_ready = false;   // multi_prep is what it will be (possibly null); false: derived by the first gk_table_totals that needs it
                              //   (half of a K8sContainerLimits AddConstraint went into a formula only the audit's RESULT totals read)
  // RESULT COUNTS on the device (round 4; Template::count_forms): rows whose true values sum to the review's number of results
  // + a flag "cannot tell" (rendered).  `cforms` is derived with the violation formula (one evaluation of the template) so that
  // the message-key paths are registered with the flattener at AddConstraint; `count_rows` / `count_flag` are the prepared,
  // trial-lowered forms (first gk_table_totals; empty = this constraint is served by multi_prep as in round 3)
  std::shared_ptr<const Template::CountForms> cforms;
  std::vector<Pattern> count_reads;   // the paths the counting forms' atoms name (publish_read_set: pruned tables must hold their rows)
  std::vector<std::shared_ptr<const PreparedConstraint>> count_rows;
  std::shared_ptr<const PreparedConstraint> count_flag;
  bool count_ready = false;
  // referential template (reads data.inventory): compiled against a snapshot of the synced objects, again whenever they change
  // (refresh_referential); `broken`: why the current inventory does not compile -- every evaluation then fails (GK_ERR_UNSUPPORTED)
  bool referential = false;
  std::string broken;
  bool alive = true;
};

// ---- host-side matcher: only used to produce the exact autoreject message for pairs the device flagged -----------
// pkg/mutation/match/match.go:32-258, pkg/target/matcher.go:44-71
bool wildcard_matches(const std::string& w, const std::string& c) {
  bool pre = !w.empty() && w.front() == '*', suf = !w.empty() && w.back() == '*';
  if (pre && suf) { std::string in = w.substr(1); if (!in.empty() && in.back() == '*') in.pop_back(); return c.find(in) != std::string::npos; }
  if (pre) { std::string s = w.substr(1); return c.size() >= s.size() && c.compare(c.size() - s.size(), s.size(), s) == 0; }
  if (suf) { std::string p = w.substr(0, w.size() - 1); return c.compare(0, p.size(), p) == 0; }
  return w == c;
}

std::string selector_error(const Value& sel) { return selector_error_text(sel); }

// IsNamespaceExcluded for one review (excluder.go:96-105).  Bare objects: the object itself (audit).  AdmissionRequests:
// the webhook's view, common.go:149-189 -- oldObject on DELETE else object, its namespace overwritten by request.namespace;
// an object that does not decode is an error there, which the handler logs and then reviews the request anyway.
bool excluder_matches(const std::vector<std::string>& pats, const std::string& ns);
bool review_excluded(const std::vector<std::string>& pats, int kind, const Value& body) {
  if (!body.is_object()) return false;
  const Value* obj = &body;
  std::string ns;
  if (kind == GK_REVIEW_ADMISSION_REQUEST) {
    const Value* op = body.get("operation");
    const bool del = op && op->is_string() && op->str() == "DELETE";
    obj = body.get(del ? "oldObject" : "object");
    const Value* k = obj && obj->is_object() ? obj->get("kind") : nullptr;
    if (!k || !k->is_string() || k->str().empty()) return false;
    const Value* rns = body.get("namespace");
    if (rns && rns->is_string()) ns = rns->str();
  } else {
    ns = obj_string(*obj, "metadata", "namespace");
  }
  if (obj_is_namespace(*obj)) return excluder_matches(pats, obj_string(*obj, "metadata", "name"));
  return excluder_matches(pats, ns);
}

// exactOrWildcardMatch (excluder.go:120-128)
bool excluder_matches(const std::vector<std::string>& pats, const std::string& ns) {
  for (auto& w : pats) if (wildcard_matches(w, ns)) return true;
  return false;
}

// error text of match.Matches for one candidate object, "" if none (subset: the error sources of match.go)
std::string candidate_error(const Value& m, const Value& obj, const Value& ns, int source, bool* matched) {
  *matched = false;
  std::string g, v, k;
  obj_gvk(obj, &g, &v, &k);
  bool is_ns = k == "Namespace" && g.empty();
  std::string name = obj_string(obj, "metadata", "name"), nsfield = obj_string(obj, "metadata", "namespace");
  auto strs = [](const Value* l) { std::vector<std::string> o; if (l && l->is_array()) for (auto& x : l->items()) if (x.is_string()) o.push_back(x.str()); return o; };
  // kinds
  const Value* kinds = m.get("kinds");
  if (kinds && kinds->is_array() && kinds->size()) {
    bool any = false;
    for (auto& kk : kinds->items()) {
      auto ks = strs(kk.get("kinds")), gs = strs(kk.get("apiGroups"));
      auto has = [](const std::vector<std::string>& l, const std::string& x) { return l.empty() || std::find(l.begin(), l.end(), "*") != l.end() || std::find(l.begin(), l.end(), x) != l.end(); };
      if (has(ks, k) && has(gs, g)) any = true;
    }
    if (!any) return "";
  }
  std::string scope = obj_string(m, "scope");
  bool has_ns = !nsfield.empty() || ns.defined();
  if (scope == "Cluster" && !(is_ns || !has_ns)) return "";
  if (scope == "Namespaced" && !(!is_ns && has_ns)) return "";
  bool has_nsname = true;
  std::string nsname;
  if (is_ns) nsname = name; else if (ns.defined()) nsname = obj_string(ns, "metadata", "name"); else if (!nsfield.empty()) nsname = nsfield; else has_nsname = false;
  auto nss = strs(m.get("namespaces"));
  if (!nss.empty() && has_nsname) { bool any = false; for (auto& w : nss) if (wildcard_matches(w, nsname)) any = true; if (!any) return ""; }
  auto ex = strs(m.get("excludedNamespaces"));
  if (!ex.empty() && has_nsname) for (auto& w : ex) if (wildcard_matches(w, nsname)) return "";
  const Value* ls = m.get("labelSelector");
  if (ls && ls->is_object()) { std::string e = selector_error(*ls); if (!e.empty()) return e; }
  // (label selector truth is decided on the device; for error text we only need to know whether evaluation continues,
  //  which the device already established by flagging this pair)
  const Value* nsel = m.get("namespaceSelector");
  if (nsel && nsel->is_object() && !(!is_ns && !ns.defined() && nsfield.empty())) {
    std::string e = selector_error(*nsel);
    if (!e.empty()) return e;
    if (!is_ns && !ns.defined()) return "namespace selector for namespace-scoped object but missing Namespace";
  }
  std::string src = obj_string(m, "source");
  if (src.empty()) src = "All";
  if (src != "All" && src != "Original" && src != "Generated") return "invalid source field \"" + src + "\"";
  if (source == SRC_EMPTY && src != "All") return "source field not specified for resource " + name;
  if (src != "All" && source == SRC_INVALID) return "invalid source field";
  *matched = true;
  return "";
}

std::string autoreject_message(const Value& match, const ReviewDoc& doc) {
  const Value* obj = doc.request.get("object");
  const Value* old = doc.request.get("oldObject");
  // gkReviewToObject (matcher.go:73-93): object first, then oldObject; the raw bytes are echoed (here: compact,
  // key-sorted JSON of the parsed document)
  if (obj && obj->is_object() && obj_string(*obj, "kind").empty())
    return "unable to match constraints: invalid request object: failed to unmarshal gkReview object " + to_json(*obj);
  if (old && old->is_object() && obj_string(*old, "kind").empty())
    return "unable to match constraints: invalid request object: failed to unmarshal gkReview oldObject " + to_json(*old);
  bool any = false;
  for (const Value* c : {obj, old}) {
    if (!c || !c->is_object()) continue;
    any = true;
    bool matched;
    std::string e = candidate_error(match, *c, doc.match_ns, doc.source, &matched);
    if (!e.empty())
      return "unable to match constraints: error matching the requested object: " + obj_string(*c, "metadata", "name") +
             " :failed to run Match criteria: " + e;
    if (matched) break;
  }
  if (!any) return "unable to match constraints: invalid request object: neither object nor old object are defined";
  return "unable to match constraints: error matching the requested object";
}

}  // namespace

// Persistent host workers for table builds.  A 64k-review batch gives each of 128 threads ~3 ms of flattening: spawning and
// joining the threads per table cost more than the work (round 2: 4.6x on 256 threads).  The pool is process-wide and sized
// to the hardware; run(n, fn) executes fn(0..n-1) on the workers and the caller, returns when all are done.  Concurrent
// run() calls (the batcher's workers) are served one after the other.
class HostWorkers {
 public:
  static HostWorkers& get() { static HostWorkers w; return w; }
  void run(size_t n, const std::function<void(size_t)>& fn) {
    if (n <= 1) { if (n) fn(0); return; }
    std::lock_guard<std::mutex> serial(run_mu_);
    {
      std::lock_guard<std::mutex> l(mu_);
      while (threads_.size() + 1 < n && threads_.size() < max_) threads_.emplace_back([this] { loop(); });
      fn_ = &fn; n_ = n; next_ = 0; done_ = 0; gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> l(mu_);
    done_cv_.wait(l, [&] { return done_ == n_; });
    fn_ = nullptr;
  }
  ~HostWorkers() {
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
 private:
  HostWorkers() : max_(std::max<size_t>(1, std::thread::hardware_concurrency())) {}   // (a ceiling: callers ask for host_cpus() workers unless GK_HOST_THREADS says otherwise)
  void work() {
    for (;;) {
      size_t i;
      const std::function<void(size_t)>* fn;
      {
        std::lock_guard<std::mutex> l(mu_);
        if (!fn_ || next_ >= n_) return;
        i = next_++;
        fn = fn_;
      }
      (*fn)(i);
      std::lock_guard<std::mutex> l(mu_);
      if (++done_ == n_) done_cv_.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return stop_ || (gen_ != seen && fn_ && next_ < n_); });
        if (stop_) return;
        seen = gen_;
      }
      work();
    }
  }
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> threads_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t n_ = 0, next_ = 0, done_ = 0, max_;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

static std::atomic<uint64_t> g_engine_uid{1};
struct gk_engine {
  const uint64_t uid = g_engine_uid++;   // (per-thread caches are keyed by it: an address may be reused by a later engine)
  gk_opts opts{};
  std::set<std::string> disabled_builtins{"http.send"};   // rego.DisableBuiltins (gk_opts.disabled_builtins; the deployment's default)
  PathDict dict;
  DictRegistry dict_reg;     // leaf-local expressions of the loaded constraints (dexpr.hpp): evaluated by the flattener
  NsCache ns_cache;
  std::shared_mutex mu;   // templates / constraints / inventory
  std::map<std::string, std::shared_ptr<Template>> templates;   // lower(kind)
  std::vector<ConstraintRec> constraints;
  // data.inventory (pkg/target/target.go:40-79 ProcessData: namespace/<ns>/<gv>/<Kind>/<name>, cluster/<gv>/<Kind>/<name>).  Kept only
  // while a loaded constraint's template reads it (inv_tracking): inv_store mirrors gk_data_put / gk_data_remove, `inventory` is
  // the document built from it (generation inv_built), and the referential constraints are compiled against it as a CONSTANT --
  // again whenever it changes (inv_gen vs inv_compiled; refresh_referential, called by ensure_plan).
  Value inventory = Value::object({});
  bool inv_tracking = false;
  std::map<std::string, std::pair<std::vector<std::string>, std::string>> inv_store;   // key -> (path, JSON text)
  uint64_t inv_gen = 1, inv_built = 0, inv_compiled = 0;
  int next_quant = 0;
  // process excluder (pkg/controller/config/process/excluder.go): process -> namespace wildcards, replaced as a whole
  // from the Config resource's spec.match; `excluder_gen` lets resident chunks notice a change (guarded by mu)
  std::map<std::string, std::vector<std::string>> excluder;
  uint64_t excluder_gen = 1;
  // plan cache
  // Plans are read by every evaluation and replaced when the policy set (or the path dictionary) changes.  Evaluations of
  // DIFFERENT tables run side by side -- each holds plan_rw shared for as long as its launches use the plan (the batcher's
  // workers, a streamed batch being evaluated while the next one is uploaded) --; whoever replaces or walks the plans
  // takes it exclusively.  plan_gate is a turnstile in front of the shared side, so that a waiting writer is not starved by
  // a steady stream of readers; variants_mu guards the table-specialised variants, which evaluations create on demand.
  std::shared_mutex plan_rw;
  std::mutex plan_gate, variants_mu;
  // POLICY EPOCH: Driver.Query holds the driver's read lock for the whole call and AddTemplate / AddConstraint / Remove* its write
  // lock (pkg/drivers/k8scel/driver.go:61,131,140,168-169) -- a query never sees a policy set change between its two halves.  Here
  // an admission batch is flattened for the policy set loaded now (a pruned table) and evaluated a moment later: the batcher holds
  // this lock shared around the two steps, the four policy mutators take it exclusively (policy_gate: the turnstile that keeps a
  // waiting writer from being starved by back-to-back batches).  Nothing else takes it, so it nests with no other lock order.
  std::shared_mutex policy_rw;
  std::mutex policy_gate;
  bool plan_dirty = true;
  uint64_t plan_gen = 0;   // bumped whenever the device plan (and with it every variant) is rebuilt
  HostPlan fast, big;
  DevPlan* dev_plan = nullptr;
  std::vector<uint32_t> plan_ids;   // bitmap row -> constraint id
  // table-specialised variants of the fast plan (GK_TABLE_RESIDENT): same formulas, element capacities = what the
  // table's largest arrays need, so the per-review LDS footprint (hence occupancy) fits the data
  // A plan holds at most 64 distinct violation / match formulas (one result bit each) and 32 element scopes.  Larger
  // constraint sets are split: the first group is the primary plan above, the others are evaluated one after the
  // other over the same resident table (each on its own view: result buffers + path binding) and their bitmap rows
  // are appended, so callers see one [n_constraints][n_tiles] answer.
  struct Group { HostPlan fast, big; DevPlan* dev = nullptr; std::vector<uint32_t> ids; std::vector<uint8_t> roles; };   // roles: totals groups only (TR_*)
  std::vector<std::unique_ptr<Group>> extra;
  struct Variant { HostPlan fast; DevPlan* dev = nullptr; };
  std::map<std::pair<size_t, std::vector<uint16_t>>, std::unique_ptr<Variant>> variants;   // key: (plan group, capacities)
  // plans of the "more than one result" formulas (ConstraintRec::multi_prep), evaluated by gk_table_totals only; built on its
  // first call after a policy change (totals_gen = the plan generation they belong to; guarded by totals_mu, plan_rw shared)
  std::vector<std::unique_ptr<Group>> totals_groups;
  uint64_t totals_gen = ~0ull;
  std::mutex totals_mu;
  std::string last_dump;
  DevComm* comm = nullptr;   // multi-GPU exchange (gk_comm_init)
  // ---- resident set (row f2): every object synced through gk_data_put, flattened in HBM in chunks
  struct ResObj {
    std::vector<std::string> path;
    std::string key, json, ns;        // ns: the namespace the object lives in ("" = cluster scoped)
    bool is_namespace = false, alive = true;
    uint32_t chunk = UINT32_MAX, slot = 0;
  };
  struct ResChunk {
    gk_table* table = nullptr;
    std::vector<uint32_t> obj_of_slot;
    std::vector<uint64_t> live;       // bit per slot: still the current version of a live object
    std::vector<uint64_t> shown;      // live AND NOT excluded from the audit process by the excluder generation `excl_gen`
    uint64_t excl_gen = 0;
    gk_eval_out* ev = nullptr;        // bitmaps of the chunk's most recent evaluation
    uint64_t plan_gen = 0;            // ... and the plan generation they belong to
    uint64_t n_live = 0;
  };
  struct Resident {
    std::mutex mu;
    std::vector<ResObj> objs;
    std::unordered_map<std::string, uint32_t> by_key;
    std::unordered_map<std::string, std::vector<uint32_t>> by_ns;   // namespace -> objects living in it
    std::unordered_map<uint64_t, uint32_t> by_text;                 // hash of the JSON text -> object (gk_query's shortcut)
    std::vector<uint32_t> pending;                                   // objects to (re)flatten at the next sweep
    std::vector<ResChunk> chunks;
    bool swept = false;               // bitmaps are current (no put / remove / policy change since the last sweep)
    uint64_t n_live = 0, n_dead_slots = 0, flattened_total = 0;
  } resident;
  // ---- admission micro-batcher (gk_query): concurrent single-review calls coalesced into one table + one launch
  struct BatchResult;   // one evaluated batch: table + bitmaps, shared by its requests until the last one has rendered
  struct Request {
    const gk_review_in* in = nullptr;
    bool pre_matched = false;               // GK_QUERY_PRE_MATCHED: the review's RF_PREMATCHED bit in the batch's table
    std::chrono::steady_clock::time_point arrived;
    int status = GK_OK;
    std::string error;
    std::shared_ptr<BatchResult> batch;   // results are rendered by the CALLER's thread from its column of the bitmaps
    uint32_t index = 0;
    uint32_t batch_size = 0;
    double queue_us = 0, device_us = 0;
    bool done = false;
    std::condition_variable cv;
  };
  struct Batcher {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Request*> queue;
    std::vector<std::thread> workers;   // each takes whole batches: one flattens while another's launch is on the device
    bool running = false, stop = false;
    gk_batch_opts opts{64, 200, 0, 0};
    uint64_t batches = 0, reviews = 0;
  } batcher;
};

struct gk_table {
  gk_engine* eng;
  DevTable* dev = nullptr;
  HostTable host;               // rows/heap released after upload unless needed
  std::vector<ReviewDoc> docs;  // GK_TABLE_KEEP_DOCS
  std::vector<gk_review_in> texts;   // GK_TABLE_KEEP_TEXT: where each review's JSON lives (caller-owned text)
  std::shared_ptr<void> owned_text;  // gk_table_create_spool: the table owns the spooled text its `texts` point into
  std::vector<std::string> review_errors;
  uint64_t n_rejected = ~0ull;              // non-empty review_errors (counted by the first sharded sweep)
  uint64_t dir_bytes = 0, n_rows = 0;      // review-flag bytes (read by every launch); rows in the table
  std::vector<uint32_t> path_rows;          // rows per path: which rows a plan reads
  std::vector<uint32_t> path_max;           // per element path: largest array of one review
  std::vector<DevTable*> views;             // one per extra plan group (shares the device arrays of `dev`)
  std::vector<DevTable*> tviews;            // one per totals plan group (gk_table_totals)
  bool resident = false;
  bool any_prematched = false;              // some review carries RF_PREMATCHED (host.rflags says which)
  uint64_t cached_gen = 0;                  // plan generation the cached variant choices belong to
  std::vector<DevPlan*> cached_plan;        // per plan group (0 = the primary plan)
  std::vector<const HostPlan*> cached_host;
  std::vector<uint32_t> slot_path;          // path of each slot of the table's row-group index
  // per review: group \0 version \0 kind \0 namespace \0 name (audit order, manager.go:118-138) -- the bytes in one arena per host
  // thread's part, (offset, length) per review: a std::string per review was a malloc and a free per review
  std::vector<std::string> key_arena;
  std::vector<std::pair<uint32_t, uint32_t>> key_span;
  size_t key_per_part = 1;
  std::string_view obj_key(size_t i) const { const auto& sp = key_span[i]; return std::string_view(key_arena[i / key_per_part].data() + sp.first, sp.second); }
  std::vector<uint32_t> order, grp;         // reviews sorted by obj_keys / dense rank with ties equal (built by the first gk_table_topk)
  uint32_t last_nc = 0;
  std::vector<uint32_t> last_ids;
  // Reviews that may end up beyond the device's limits (RF_TOO_BIG / RF_REFUSE / RF_HOST_CAND at flatten time): their text is kept
  // whatever the table's flags, so that gk_table_eval can evaluate them on the host (complete_on_host) instead of refusing them
  struct OwnedReview {
    int32_t kind = 0, source = 0;
    std::string json, ns, nsobj, op;
    bool has_ns = false, has_nsobj = false, has_op = false;
    gk_review_in in() const {
      gk_review_in r{};
      r.kind = kind; r.source = source; r.json = json.data(); r.json_len = json.size();
      r.namespace_json = has_ns ? ns.data() : nullptr; r.namespace_len = has_ns ? ns.size() : 0;
      r.ns_object_json = has_nsobj ? nsobj.data() : nullptr; r.ns_object_len = has_nsobj ? nsobj.size() : 0;
      r.operation = has_op ? op.c_str() : nullptr;
      return r;
    }
  };
  std::map<uint32_t, OwnedReview> big_texts;
  std::mutex big_mu;
  // (row, review) pairs the host evaluation of the most recent gk_table_eval added: violations / autoreject errors the device
  // bitmaps do not hold (gk_table_totals and gk_table_topk read those)
  // (stamped with the constraint-id list of that evaluation: a reader whose own list differs -- a policy change renumbered the rows --
  //  drops them; host_mu orders the evaluation that writes them and the totals / top-k calls that read them)
  std::vector<std::pair<uint32_t, uint32_t>> host_viol, host_err;
  std::vector<uint32_t> host_ids;
  std::mutex host_mu;
  uint32_t n_reviews = 0;
  uint32_t rpt = GK_RPT_MIN;                // reviews per row group of this table
  uint64_t dict_gen = 0;                    // generation of the dictionary-predicate registry the rows were flattened under
  bool pruned = false;                      // GK_TABLE_PRUNED: rows of the registry's read set only ...
  uint64_t reads_gen = 0;                   // ... as it was when the table was built (DictRegistry::reads_gen)
  ShardInfo shard;                          // sharded sweeps: slot layout agreed with the other ranks
  std::vector<ShardInfo> group_shards;      // ... of the further plan groups (on their views)
  uint64_t shard_gen = 0;
  gk_table_stats stats{};
};

namespace {

Value set_in(const Value& root, const std::vector<std::string>& path, size_t i, const Value* leaf) {
  // returns a copy of root with path[i..] set to *leaf (or removed when leaf == nullptr)
  ValuePairs p = root.is_object() ? root.pairs() : ValuePairs{};
  Value key = Value::string(path[i]);
  ValuePairs out;
  bool done = false;
  for (auto& kv : p) {
    if (kv.first == key) {
      done = true;
      if (i + 1 == path.size()) { if (leaf) out.emplace_back(key, *leaf); }
      else out.emplace_back(key, set_in(kv.second, path, i + 1, leaf));
    } else out.push_back(kv);
  }
  if (!done && leaf) {
    if (i + 1 == path.size()) out.emplace_back(key, *leaf);
    else out.emplace_back(key, set_in(Value::object({}), path, i + 1, leaf));
  }
  return Value::object(out);
}

PlanCaps default_caps(const gk_engine* e) {
  PlanCaps caps;
  for (int i = 0; i < 3; i++) if (e->opts.elem_cap[i]) caps.level_cap[i] = e->opts.elem_cap[i];
  return caps;
}

// Plan for one table.  Resident tables (audit sets evaluated again and again) get a variant whose element capacities
// are what the table's largest arrays need: fewer accumulator words per review -> more tiles resident per CU, and
// reviews that would overflow the default capacities stay on the LDS kernel.  Caller holds plan_rw (shared suffices) and variants_mu; ensure_plan ran.
// `group`: 0 = the primary plan, g >= 1 = e->extra[g - 1] (a policy set of more than 64 formulas: round 5 -- the further groups ran
// with the default capacities (8, 8, 8, 12, 12 -> 81 accumulator words for the 200-template corpus, whose pods hold at most 4
// containers) until then).
DevPlan* plan_for_group(gk_engine* e, gk_table* t, size_t group, const HostPlan** host) {
  const HostPlan& base = group == 0 ? e->fast : e->extra[group - 1]->fast;
  DevPlan* base_dev = group == 0 ? e->dev_plan : e->extra[group - 1]->dev;
  *host = &base;
  if (!t->resident || base.scopes.empty()) return base_dev;
  if (t->cached_gen != e->plan_gen) { t->cached_plan.clear(); t->cached_host.clear(); t->cached_gen = e->plan_gen; }
  if (t->cached_plan.size() <= group) { t->cached_plan.resize(group + 1, nullptr); t->cached_host.resize(group + 1, nullptr); }
  if (t->cached_plan[group]) { *host = t->cached_host[group]; return t->cached_plan[group]; }   // per launch: no rescan
  std::vector<uint16_t> caps(base.scopes.size(), 1);
  for (size_t p = 0; p < base.ptab.size(); p++) {
    uint32_t ent = base.ptab[p];
    for (uint32_t j = 0; j < (ent & 0xFF); j++) {
      const Pred& pr = base.path_preds[(ent >> 8) + j];
      if (pr.op != P_PRESENT || pr.dst != D_ELEM) continue;
      uint32_t need = p < t->path_max.size() ? t->path_max[p] : 0;
      // (the marker rides on a member of the element -- plan.hpp T_ABSENT: the element count is the element path's, its parent)
      if (const PathDict::Info in = e->dict.info((uint32_t)p); !in.is_elem && in.parent < t->path_max.size()) need = std::max(need, t->path_max[in.parent]);
      if (need > caps[pr.scope]) caps[pr.scope] = (uint16_t)std::min<uint32_t>(need, 255);
    }
  }
  static const uint16_t steps[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 255};
  for (auto& c : caps) for (uint16_t s : steps) if (s >= c) { c = s; break; }
  const std::pair<size_t, std::vector<uint16_t>> key(group, caps);
  auto it = e->variants.find(key);
  if (it == e->variants.end()) {
    std::unique_ptr<gk_engine::Variant> v(new gk_engine::Variant());
    try {
      PlanBuilder pb(&e->dict, &e->dict_reg);
      const std::vector<uint32_t>& ids = group == 0 ? e->plan_ids : e->extra[group - 1]->ids;
      for (uint32_t id : ids) pb.add_constraint(e->constraints[id].prep);
      PlanCaps pc = default_caps(e);
      pc.scope_cap = caps;
      v->fast = pb.build(pc);
      if (v->fast.scopes.size() != base.scopes.size()) throw std::runtime_error("scope layout changed");
      v->dev = dev_plan_upload(e->opts.device, v->fast, group == 0 ? e->big : e->extra[group - 1]->big);
    } catch (const std::exception&) { v->dev = nullptr; }   // e.g. LDS limit: the default plan serves the table
    it = e->variants.emplace(key, std::move(v)).first;
  }
  if (!it->second->dev) { t->cached_plan[group] = base_dev; t->cached_host[group] = &base; return base_dev; }
  *host = &it->second->fast;
  t->cached_plan[group] = it->second->dev; t->cached_host[group] = &it->second->fast;
  return it->second->dev;
}
DevPlan* plan_for_table(gk_engine* e, gk_table* t, const HostPlan** host) { return plan_for_group(e, t, 0, host); }

void refresh_referential(gk_engine* e);
// The READ SET of pruned tables (GK_TABLE_PRUNED; flatten.hpp DictRegistry::set_reads): every path pattern some plan has a predicate
// on -- all plan groups, both capacities -- plus the paths the constraints' counting forms name (their plans are built at the
// first gk_table_totals, after the tables).  Caller holds plan_rw exclusively and mu shared (ensure_plan).
void publish_read_set(gk_engine* e) {
  std::vector<Pattern> pats;
  auto take = [&](const HostPlan& hp) { pats.insert(pats.end(), hp.pred_patterns.begin(), hp.pred_patterns.end()); };
  take(e->fast); take(e->big);
  for (auto& g : e->extra) { take(g->fast); take(g->big); }
  for (auto& c : e->constraints) if (c.alive) pats.insert(pats.end(), c.count_reads.begin(), c.count_reads.end());
  e->dict_reg.set_reads(pats);
}

void ensure_plan(gk_engine* e) {
  refresh_referential(e);   // referential constraints follow the synced inventory (no-op unless one is loaded and it changed)
  {   // the usual case: nothing changed -- decided without keeping other evaluations out
    std::shared_lock<std::shared_mutex> pl(e->plan_rw);
    std::shared_lock<std::shared_mutex> rl(e->mu);
    if (!e->plan_dirty && e->dev_plan && e->fast.dict_size == e->dict.size()) return;
  }
  std::lock_guard<std::mutex> gate(e->plan_gate);
  std::unique_lock<std::shared_mutex> l(e->plan_rw);
  std::shared_lock<std::shared_mutex> rl(e->mu);
  if (!e->plan_dirty && e->dev_plan && e->fast.dict_size == e->dict.size()) return;
  for (auto& v : e->variants) dev_plan_free(v.second->dev);
  e->variants.clear();
  PlanCaps bigcaps;
  bigcaps.level_cap[0] = bigcaps.level_cap[1] = bigcaps.level_cap[2] = 256;
  if (e->plan_dirty || !e->dev_plan) {
    for (auto& g : e->extra) dev_plan_free(g->dev);
    e->extra.clear();
    std::vector<const ConstraintRec*> alive;
    for (auto& c : e->constraints) if (c.alive) alive.push_back(&c);
    for (auto* c : alive) if (!c->broken.empty()) throw Unsupported(c->broken);   // (a referential constraint the current inventory does not compile for)
    // groups of constraints that fit one plan: everything if possible, else chunks of <= GK_MAX_VIOL constraints (a constraint
    // contributes one violation and one match formula), halved further while a chunk still does not lower
    std::vector<std::vector<const ConstraintRec*>> groups;
    auto builds = [&](const std::vector<const ConstraintRec*>& g, HostPlan* fast, HostPlan* big) {
      PlanBuilder pb(&e->dict, &e->dict_reg);
      for (auto* c : g) pb.add_constraint(c->prep);
      *fast = pb.build(default_caps(e));
      *big = pb.build(bigcaps);
    };
    std::vector<std::pair<HostPlan, HostPlan>> plans;
    std::function<void(const std::vector<const ConstraintRec*>&)> place = [&](const std::vector<const ConstraintRec*>& g) {
      HostPlan f, b;
      // (test / tuning aid, gk_debug_set("group_max", n): no plan group of more than n constraints -- the several-group machinery,
      //  which a set of more than 256 distinct violation formulas still takes, stays testable with a small corpus)
      if (const int forced = g_debug_group_max.load(); forced > 0 && g.size() > (size_t)forced) {
        for (size_t i = 0; i < g.size(); i += (size_t)forced)
          place(std::vector<const ConstraintRec*>(g.begin() + i, g.begin() + std::min(g.size(), i + (size_t)forced)));
        return;
      }
      try { builds(g, &f, &b); }
      catch (const Unsupported& u) {
        if (g.size() == 1 && g[0]->referential)   // (its formula follows the synced objects: say so -- the same words refresh_referential uses)
          throw Unsupported(std::string("referential constraint ") + g[0]->kind + "/" + g[0]->name + " does not compile against the synced inventory: " + u.what());
        if (g.size() <= 1) throw;
        size_t cap = GK_MAX_VIOL;   // (a constraint contributes at most one violation formula; round 6: 256 slots per plan, 64 before)
        if (const char* gm = getenv("GK_GROUP_MAX")) cap = (size_t)std::max(1, std::min((int)GK_MAX_VIOL, atoi(gm)));   // tuning aid: smaller plan groups (smaller code objects, fewer accumulator words, more walks of the table)
        size_t half = g.size() > cap ? cap : g.size() / 2;
        for (size_t i = 0; i < g.size(); i += half)
          place(std::vector<const ConstraintRec*>(g.begin() + i, g.begin() + std::min(g.size(), i + half)));
        return;
      }
      groups.push_back(g);
      plans.emplace_back(std::move(f), std::move(b));
    };
    place(alive);
    e->plan_ids.clear();
    if (groups.empty()) { groups.emplace_back(); HostPlan f, b; builds(groups[0], &f, &b); plans.emplace_back(std::move(f), std::move(b)); }
    for (auto* c : groups[0]) e->plan_ids.push_back(c->id);
    e->fast = std::move(plans[0].first);
    e->big = std::move(plans[0].second);
    for (size_t gi = 1; gi < groups.size(); gi++) {
      std::unique_ptr<gk_engine::Group> g(new gk_engine::Group());
      g->fast = std::move(plans[gi].first);
      g->big = std::move(plans[gi].second);
      for (auto* c : groups[gi]) g->ids.push_back(c->id);
      e->extra.push_back(std::move(g));
    }
  } else {
    try {
      e->fast.resolve_paths(e->dict);
      e->big.resolve_paths(e->dict);
      for (auto& g : e->extra) { g->fast.resolve_paths(e->dict); g->big.resolve_paths(e->dict); }
    } catch (const Unsupported& u) {
      // the plan was built before the table's key paths were known and does not fit them: with a referential constraint loaded
      // that is its unrolled inventory (one predicate per synced value on the joined path) -- say so
      for (auto& c : e->constraints) if (c.alive && c.referential)
        throw Unsupported(std::string("referential constraint ") + c.kind + "/" + c.name + " does not compile against the synced inventory: " + u.what());
      throw;
    }
  }
  for (auto& g : e->extra) {
    if (g->dev) { dev_plan_free(g->dev); g->dev = nullptr; }
    g->dev = dev_plan_upload(e->opts.device, g->fast, g->big);
  }
  if (e->dev_plan) { dev_plan_free(e->dev_plan); e->dev_plan = nullptr; }
  e->dev_plan = dev_plan_upload(e->opts.device, e->fast, e->big);
  e->plan_gen++;
  e->plan_dirty = false;
  publish_read_set(e);
}

Value parse_opt(const char* p, size_t n) { return (p && n) ? parse_json(p, n) : Value(); }

// thresholds per counted iteration: "at least k elements fire", k = 2 .. GK_COUNT_KMAX (one more flags the review); GK_COUNT_KMAX=0
// switches the device-side result counting off (round 3's "more than one result?" decision serves alone)
enum : uint8_t { TR_MULTI = 0, TR_FLAG = 1, TR_COUNT = 2 };   // what a row of a totals plan answers (gk_engine::Group::roles)
int count_kmax() { return getenv("GK_COUNT_KMAX") ? atoi(getenv("GK_COUNT_KMAX")) : 4; }   // (read per call: a test switches it within one process)

// The counting forms of a constraint from the evaluation that gave its violation formula; registers the message-key paths with
// the flattener (caller holds mu exclusively: a new key path makes the tables flattened so far stale, like a new value path)
std::shared_ptr<const Template::CountForms> derive_count_forms(gk_engine* e, const Template::CountInfo& ci, const MatchFormulas& mf, FP* merged_viol = nullptr,
                                                                 std::vector<Pattern>* reads = nullptr) {
  if (reads) reads->clear();
  if (count_kmax() < 2) return nullptr;
  try {
    const auto tf0 = std::chrono::steady_clock::now();
    auto cf = std::make_shared<Template::CountForms>(Template::count_forms(ci, count_kmax()));
    if (getenv("GK_DEBUG_LOAD")) fprintf(stderr, "[gkgpu load]   count_forms %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count());
    if (merged_viol && cf->viol && ci.br.size() > 8) *merged_viol = cf->viol;   // (a constraint with many unrolled alternatives: the merged form is the smaller formula)
    if (!cf->ok) return nullptr;
    // What the counting plans will read must be registered NOW, before tables are flattened: the merged bodies of the counted
    // branches fold their leaf-local parts into dictionary expressions of their own (the violation formula folds the union of
    // all alternatives).  One trial lowering of every branch's plain row and of the flag, against the live registry; the
    // threshold rows repeat those bodies.  A form that does not lower: no counting for this constraint (round 3's answer serves).
    // What the counting plans will read must be registered NOW, before tables are flattened: the merged bodies of the counted
    // branches fold their leaf-local parts into dictionary expressions of their own (the violation formula folds the union of all
    // alternatives).  The forms are prepared (simplified, folded) and their dictionary atoms interned in the registry's COUNTING
    // space -- no lowering: that happens once, at the first gk_table_totals, against the then frozen registry.
    {
      PrepMemoScope memo;
      std::vector<FP> probe;
      for (uint32_t i : cf->firsts) probe.push_back(cf->rows[i]);
      probe.push_back(cf->flag);
      auto pattern_of = [](const SPath& p) { Pattern pat; for (const Step& st : p) { PatStep ps; if (st.iter) ps.any = true; else ps.key = st.key; pat.push_back(ps); } return pat; };
      std::function<void(const FP&)> walk = [&](const FP& f) {
        if (f->kind == FNode::ATOM) {
          if (f->atom.kind == Atom::DICT) {
            // (the review facts live in the main space, whoever asks: lower.cpp REVIEW FACTS)
            const bool facts = review_fact_leaf(f->atom.path);
            if (!f->atom.alt) { if (facts) e->dict_reg.intern(pattern_of(f->atom.path), f->atom.dx, true, nullptr, false); else e->dict_reg.counting().intern(pattern_of(f->atom.path), f->atom.dx, true); return; }
            // (a promoted group of row predicates: when the counting dictionary cannot take it the totals plan reads the rows)
            try { if (facts) e->dict_reg.intern(pattern_of(f->atom.path), f->atom.dx, true, nullptr, true); else e->dict_reg.counting().intern(pattern_of(f->atom.path), f->atom.dx, true); } catch (const std::runtime_error&) {}
            walk(f->atom.alt);
            return;
          }
          // (every other atom reads rows of its path: a pruned table must hold them -- publish_read_set)
          if (reads && !f->atom.path.empty()) reads->push_back(pattern_of(f->atom.path));
          if (reads && !f->atom.path2.empty()) reads->push_back(pattern_of(f->atom.path2));
          return;
        }
        if (reads && f->kind == FNode::EXISTS) {   // (the element marker rows of a counted array -- and the container's own row: the plan-local guard of a counting loop looks at its type)
          SPath el = f->base; Step st; st.iter = true; st.q = f->q; el.push_back(st); reads->push_back(pattern_of(el));
          if (!f->base.empty()) reads->push_back(pattern_of(f->base));
        }
        for (auto& k : f->kids) walk(k);
      };
      for (auto& f : probe) walk(prepare_constraint(f, mf)->viol);
    }
    for (const SPath& key : cf->keys) {
      Pattern pat;
      for (const Step& st : key) { PatStep ps; if (st.iter) { ps.any = true; ps.elems_only = true; } else ps.key = st.key; pat.push_back(ps); }
      e->dict_reg.add_key(pat);
    }
    return cf;
  } catch (const std::exception& ex) {
    if (getenv("GK_DEBUG_MULTI")) fprintf(stderr, "[gkgpu totals] no counting forms: %s\n", ex.what());
    return nullptr;
  }
}

// the counting forms prepared and trial-lowered against the FROZEN registry (as prepare_multi): all or nothing
void prepare_counts(gk_engine* e, ConstraintRec& c) {
  c.count_rows.clear(); c.count_flag = nullptr; c.count_ready = true;
  if (!c.cforms || !c.cforms->ok) return;
  try {
    PrepMemoScope memo;   // (the rows and the flag share their bodies, node for node)
    std::vector<std::shared_ptr<const PreparedConstraint>> rows;
    for (const FP& r : c.cforms->rows) rows.push_back(prepare_constraint(r, c.mf));
    auto flag = prepare_constraint(c.cforms->flag, c.mf);
    std::vector<std::shared_ptr<const PreparedConstraint>> all = rows;
    all.push_back(flag);
    for (size_t i = 0; i < all.size(); i += 24) {   // (trial builds in batches that fit a plan's result slots)
      PlanBuilder pb(&e->dict, &e->dict_reg, true);
      pb.use_counting_space();
      for (size_t j = i; j < std::min(all.size(), i + 24); j++) pb.add_constraint(all[j]);
      PlanCaps caps;
      pb.build(caps);
    }
    c.count_rows = std::move(rows); c.count_flag = std::move(flag);
  } catch (const std::exception& ex) {
    if (getenv("GK_DEBUG_MULTI")) fprintf(stderr, "[gkgpu totals] %s/%s: counting forms do not lower (%s): the multi formula serves\n", c.kind.c_str(), c.name.c_str(), ex.what());
  }
}

// "may yield more than one result" formula of a constraint (Template::compile_multi), prepared and checked to lower like the
// violation formula is -- at AddConstraint time, so that the dictionary predicates / value-id paths it needs are registered
// before tables are flattened.  Null when it does not lower: gk_table_totals then renders every violating pair of it.
std::shared_ptr<const PreparedConstraint> prepare_multi(gk_engine* e, const Template& t, const Value& params, const MatchFormulas& mf, const Value& inventory = Value()) {
  try {
    FP multi = t.compile_multi(params, &e->next_quant, inventory);
    auto prep = prepare_constraint(multi, mf);
    // (frozen: the formula must be answerable from the rows, dictionary bits and value ids the violation formulas asked the
    //  flattener for -- a guard of its own, say, would refuse reviews the violation formulas can evaluate)
    PlanBuilder pb(&e->dict, &e->dict_reg, true);
    pb.add_constraint(prep);
    PlanCaps caps;
    pb.build(caps);
    return prep;
  } catch (const std::exception& ex) {
    if (getenv("GK_DEBUG_MULTI")) fprintf(stderr, "[gkgpu totals] no multi formula: %s\n", ex.what());
    return nullptr;
  }
}

// ---- referential templates (data.inventory) ----------------------------------------------------------------------------------
// data.inventory as one document, from the mirror of the synced objects (caller holds mu exclusively)
const Value& current_inventory(gk_engine* e) {
  if (e->inv_built == e->inv_gen) return e->inventory;
  struct Node { std::map<std::string, Node> kids; const std::string* json = nullptr; };
  Node root;
  for (auto& kv : e->inv_store) {
    Node* n = &root;
    for (auto& seg : kv.second.first) n = &n->kids[seg];
    n->json = &kv.second.second;
  }
  std::function<Value(const Node&)> build = [&](const Node& n) -> Value {
    if (n.json) return parse_json(n.json->data(), n.json->size());
    ValuePairs ps;
    for (auto& k : n.kids) ps.emplace_back(Value::string(k.first), build(k.second));
    return Value::object(ps);
  };
  e->inventory = build(root);
  e->inv_built = e->inv_gen;
  return e->inventory;
}

// start mirroring the synced objects: everything gk_data_put holds so far (caller holds resident.mu AND mu exclusively, in that order)
void start_inventory_tracking(gk_engine* e) {
  if (e->inv_tracking) return;
  for (auto& o : e->resident.objs) if (o.alive) e->inv_store[o.key] = {o.path, o.json};
  e->inv_tracking = true;
  e->inv_gen++;
}

// violation formula, prepared form and multi formula of a constraint against the current inventory (caller holds mu exclusively);
// throws what compile / the lowering throw
void compile_referential(gk_engine* e, const Template& t, ConstraintRec& c) {
  const Value& inv = current_inventory(e);
  const Template::CountInfo ci = t.compile_all(c.params, &e->next_quant, inv);
  FP viol = ci.viol;
  auto prep = prepare_constraint(viol, c.mf);
  {
    PlanBuilder pb(&e->dict, &e->dict_reg);
    pb.add_constraint(prep);
    PlanCaps caps;
    pb.build(caps);
  }
  c.viol = viol; c.prep = prep;
  c.multi_prep = prepare_multi(e, t, c.params, c.mf, inv);
  c.multi_ready = true;
  c.cforms = derive_count_forms(e, ci, c.mf, nullptr, &c.count_reads);
  c.count_ready = false;
}

// The synced objects changed: the constraints of referential templates are compiled against the new inventory.  One that no
// longer compiles (an inventory too large to unroll, say) is marked broken: evaluations fail with GK_ERR_UNSUPPORTED -- closed --
// until the inventory or the constraint changes.
void refresh_referential(gk_engine* e) {
  {
    std::shared_lock<std::shared_mutex> rl(e->mu);
    if (!e->inv_tracking || e->inv_compiled == e->inv_gen) return;
  }
  std::lock_guard<std::mutex> gate(e->plan_gate);
  std::unique_lock<std::shared_mutex> pl(e->plan_rw);
  std::unique_lock<std::shared_mutex> l(e->mu);
  if (!e->inv_tracking || e->inv_compiled == e->inv_gen) return;
  for (auto& c : e->constraints) {
    if (!c.alive || !c.referential) continue;
    auto it = e->templates.find(lower_str(c.kind));
    if (it == e->templates.end()) continue;
    try { compile_referential(e, *it->second, c); c.broken.clear(); }
    catch (const std::exception& ex) { c.broken = std::string("referential constraint ") + c.kind + "/" + c.name + " does not compile against the synced inventory: " + ex.what(); }
  }
  e->inv_compiled = e->inv_gen;
  e->plan_dirty = true;
}

// The totals plans: the alive constraints' multi formulas in groups that fit one plan each (as ensure_plan groups the violation
// formulas).  Caller holds plan_rw shared + mu shared + totals_mu; ensure_plan ran.
void ensure_totals_plans(gk_engine* e) {
  if (e->totals_gen == e->plan_gen) return;
  for (auto& g : e->totals_groups) dev_plan_free(g->dev);
  e->totals_groups.clear();
  PlanCaps bigcaps;
  bigcaps.level_cap[0] = bigcaps.level_cap[1] = bigcaps.level_cap[2] = 256;
  // the "more than one result" formulas still owed (AddConstraint leaves them to the first totals that need them; only this
  // function -- under totals_mu -- and the exclusive holders of mu touch multi_prep / multi_ready)
  for (auto& c : e->constraints) {
    if (!c.alive || c.multi_ready) continue;
    auto it = e->templates.find(lower_str(c.kind));
    c.multi_prep = it == e->templates.end() ? nullptr : prepare_multi(e, *it->second, c.params, c.mf);
    c.multi_ready = true;
  }
  // ... and the counting forms (round 4): a constraint whose forms lower is served by its count rows + flag row, any other by
  // its multi formula (round 3), one without either has every violating pair rendered
  const auto tc0 = std::chrono::steady_clock::now();
  for (auto& c : e->constraints) if (c.alive && !c.count_ready) prepare_counts(e, c);
  const auto tc1 = std::chrono::steady_clock::now();
  struct Item { std::shared_ptr<const PreparedConstraint> prep; uint32_t cid; uint8_t role; };
  std::vector<Item> have;
  for (auto& c : e->constraints) {
    if (!c.alive) continue;
    if (c.count_flag) {
      have.push_back({c.count_flag, c.id, TR_FLAG});
      for (auto& r : c.count_rows) have.push_back({r, c.id, TR_COUNT});
    } else if (c.multi_prep) have.push_back({c.multi_prep, c.id, TR_MULTI});
  }
  std::function<void(const std::vector<Item>&)> place = [&](const std::vector<Item>& g) {
    if (g.empty()) return;
    std::unique_ptr<gk_engine::Group> grp(new gk_engine::Group());
    try {
      PlanBuilder pb(&e->dict, &e->dict_reg, true);
      pb.use_counting_space(g[0].role != TR_MULTI);   // (a group holds rows of one kind: place() is called per kind)
      for (auto& it : g) pb.add_constraint(it.prep);
      grp->fast = pb.build(default_caps(e));
      grp->big = pb.build(bigcaps);
    } catch (const Unsupported&) {
      if (g.size() <= 1) {   // (checked at preparation; should it fail now: a flag / multi row that is missing means "render", a
        if (!g.empty() && g[0].role == TR_COUNT) throw;   //  missing count row would be an undercount -- never silently)
        return;
      }
      const size_t half = g.size() > 48 ? 48 : g.size() / 2;
      for (size_t i = 0; i < g.size(); i += half) place(std::vector<Item>(g.begin() + i, g.begin() + std::min(g.size(), i + half)));
      return;
    }
    for (auto& it : g) { grp->ids.push_back(it.cid); grp->roles.push_back(it.role); }
    grp->dev = dev_plan_upload(e->opts.device, grp->fast, grp->big);
    dev_plan_no_jit(grp->dev);   // (dozens of small plans, one sweep per audit each: the bytecode kernel serves them)
    e->totals_groups.push_back(std::move(grp));
  };
  {
    std::vector<Item> multis, counts;
    for (auto& it : have) (it.role == TR_MULTI ? multis : counts).push_back(it);
    place(multis);
    place(counts);
    if (getenv("GK_DEBUG_MULTI")) fprintf(stderr, "[gkgpu totals] %zu multi rows, %zu count / flag rows -> %zu plans; counting forms prepared in %.2f s, plans built in %.2f s\n", multis.size(), counts.size(),
                                          e->totals_groups.size(), std::chrono::duration<double>(tc1 - tc0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tc1).count());
  }
  e->totals_gen = e->plan_gen;
}

}  // namespace

extern "C" {

const char* gk_last_error(void) { return g_err.c_str(); }
const char* gk_version(void) { return "gkgpu 0.1 (gfx950)"; }

int gk_engine_create(const gk_opts* opts, gk_engine** out) {
  if (!out) return fail(GK_ERR_INVALID, "out is NULL");
  int dev = opts ? opts->device : 0;
  std::string err = dev_init(dev);
  if (!err.empty()) return fail(GK_ERR_DEVICE, err);
  gk_engine* e = new gk_engine();
  if (opts) {
    e->opts = *opts;
    if (opts->disabled_builtins) { e->disabled_builtins.clear(); for (size_t i = 0; i < opts->n_disabled_builtins; i++) if (opts->disabled_builtins[i]) e->disabled_builtins.insert(opts->disabled_builtins[i]); }
    e->opts.disabled_builtins = nullptr; e->opts.n_disabled_builtins = 0;   // (the caller's array is not kept)
  }
  *out = e;
  return GK_OK;
}

uint32_t gk_host_cpus(void) { return (uint32_t)host_cpus(); }

void gk_engine_destroy(gk_engine* e) {
  if (!e) return;
  gk_batcher_stop(e);
  dev_jit_quiesce();   // background builds of this engine's plans (a host that destroys its engines before exit() never exits under a running compile)
  if (e->comm) dev_comm_free(e->comm);
  for (auto& c : e->resident.chunks) { if (c.ev) gk_eval_free(c.ev); if (c.table) gk_table_free(c.table); }
  if (e->dev_plan) dev_plan_free(e->dev_plan);
  for (auto& v : e->variants) dev_plan_free(v.second->dev);
  for (auto& g : e->extra) dev_plan_free(g->dev);
  for (auto& g : e->totals_groups) dev_plan_free(g->dev);
  delete e;
}

int gk_template_add(gk_engine* e, const char* kind, const char* rego, const char* const* libs, size_t nlibs) {
  if (!e || !kind || !rego) return fail(GK_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> policy_turn(e->policy_gate);
  std::unique_lock<std::shared_mutex> policy_change(e->policy_rw);   // (no admission batch is between its flatten and its launch)
  try {
    std::vector<std::string> ls;
    for (size_t i = 0; i < nlibs; i++) ls.emplace_back(libs[i]);
    auto t = std::make_shared<Template>(rego, ls, &e->disabled_builtins);
    // rego.Externs() without "inventory" (GK_OPT_NO_REFERENTIAL, --enable-referential-rules=false): the frameworks' reference check
    // refuses a template that reads data.inventory when it is added (the wording of its error is the third-party driver's: unpinned)
    if ((e->opts.flags & GK_OPT_NO_REFERENTIAL) && t->references_inventory())
      return fail(GK_ERR_REGO, std::string("check refs failed on module {templates[\"admission.k8s.gatekeeper.sh\"][\"") + kind + "\"]}: disallowed ref data.inventory (referential rules are disabled)");
    std::unique_lock<std::mutex> res_lock(e->resident.mu, std::defer_lock);
    if (t->references_inventory()) res_lock.lock();   // (before mu: the order gk_resident_sweep uses)
    std::unique_lock<std::shared_mutex> l(e->mu);
    // Constraints of this kind are recompiled against the new template -- violation formula AND the prepared form every
    // plan is built from -- and put through the checks gk_constraint_add runs (referential template, lowering).  All or
    // nothing: when one of them does not compile the old template and its constraints stay as they are and the caller
    // gets the error (the reference reports it on the ConstraintTemplate's status and keeps serving the old one).
    const std::string k = lower_str(kind);
    struct Redo { ConstraintRec* c; FP viol; std::shared_ptr<const PreparedConstraint> prep, multi; bool referential; std::shared_ptr<const Template::CountForms> cforms; std::vector<Pattern> reads; Template::CountInfo ci; };
    std::vector<Redo> redo;
    for (auto& c : e->constraints) {
      if (!c.alive || lower_str(c.kind) != k) continue;
      const bool ref = t->references_inventory();
      if (ref) start_inventory_tracking(e);
      const Value inv = ref ? current_inventory(e) : Value();
      const Template::CountInfo ci = t->compile_all(c.params, &e->next_quant, inv);
      Redo r{&c, ci.viol, nullptr, nullptr, ref, nullptr};
      r.prep = prepare_constraint(r.viol, c.mf);
      PlanBuilder pb(&e->dict, &e->dict_reg);
      pb.add_constraint(r.prep);
      PlanCaps caps;
      pb.build(caps);
      r.multi = prepare_multi(e, *t, c.params, c.mf, inv);
      r.ci = ci;
      redo.push_back(std::move(r));
    }
    e->templates[k] = t;
    // the counting forms register message keys and dictionary atoms with the flattener (every table becomes stale): only now that
    // every constraint of the kind compiled -- a replacement that is refused leaves the registry as it was (round-4 advisor finding)
    for (auto& r : redo) r.cforms = derive_count_forms(e, r.ci, r.c->mf, nullptr, &r.reads);
    for (auto& r : redo) { r.c->viol = std::move(r.viol); r.c->prep = std::move(r.prep); r.c->multi_prep = std::move(r.multi); r.c->multi_ready = true; r.c->referential = r.referential; r.c->broken.clear(); r.c->cforms = std::move(r.cforms); r.c->count_reads = std::move(r.reads); r.c->count_ready = false; r.c->count_rows.clear(); r.c->count_flag = nullptr; }
    // (inv_compiled stays as it is: referential constraints of OTHER kinds may still hold an older inventory -- refresh_referential
    // recompiles every one of them at the next evaluation; marking the inventory compiled here left them stale, i.e. missed violations)
    e->plan_dirty = true;
    return GK_OK;
  } catch (const RegoError& ex) { return fail(GK_ERR_REGO, ex.what());
  } catch (const Unsupported& ex) { return fail(GK_ERR_UNSUPPORTED, ex.what());
  } catch (const RegexUnsupported& ex) { return fail(GK_ERR_UNSUPPORTED, std::string("unsupported on the device plan: ") + ex.what());
  } catch (const std::exception& ex) { return fail(GK_ERR_INTERNAL, ex.what()); }
}

int gk_template_remove(gk_engine* e, const char* kind) {
  if (!e || !kind) return fail(GK_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> policy_turn(e->policy_gate);
  std::unique_lock<std::shared_mutex> policy_change(e->policy_rw);   // (no admission batch is between its flatten and its launch)
  std::unique_lock<std::shared_mutex> l(e->mu);
  std::string k = lower_str(kind);
  if (!e->templates.erase(k)) return fail(GK_ERR_NOT_FOUND, "unknown template " + k);
  for (auto& c : e->constraints) if (lower_str(c.kind) == k) c.alive = false;
  e->plan_dirty = true;
  return GK_OK;
}

int gk_constraint_add(gk_engine* e, const char* json, size_t len, uint32_t* id_out) {
  if (!e || !json) return fail(GK_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> policy_turn(e->policy_gate);
  std::unique_lock<std::shared_mutex> policy_change(e->policy_rw);   // (no admission batch is between its flatten and its launch)
  try {
    Value c = parse_json(json, len);
    if (!c.is_object()) return fail(GK_ERR_INVALID, "constraint must be a JSON object");
    ConstraintRec rec;
    rec.kind = obj_string(c, "kind");
    rec.name = obj_string(c, "metadata", "name");
    rec.object = c;
    const Value* spec = c.get("spec");
    const Value* params = spec ? spec->get("parameters") : nullptr;
    rec.params = (params && !params->is_null()) ? *params : Value::object({});
    const Value* match = spec ? spec->get("match") : nullptr;
    rec.match = (match && match->is_object()) ? *match : Value();
    // (a referential template needs the synced objects: resident.mu is taken BEFORE mu, the order gk_resident_sweep uses)
    bool referential = false;
    {
      std::shared_lock<std::shared_mutex> rl(e->mu);
      auto it0 = e->templates.find(lower_str(rec.kind));
      referential = it0 != e->templates.end() && it0->second->references_inventory();
    }
    std::unique_lock<std::mutex> res_lock(e->resident.mu, std::defer_lock);
    if (referential) res_lock.lock();
    std::unique_lock<std::shared_mutex> l(e->mu);
    auto it = e->templates.find(lower_str(rec.kind));
    if (it == e->templates.end()) return fail(GK_ERR_NOT_FOUND, "unknown constraint template validator: " + rec.kind);
    rec.mf = compile_match(rec.match);
    if (it->second->references_inventory()) {
      // Referential (data.inventory): the synced objects are a CONSTANT of the compiled formula -- iterating them unrolls into
      // one alternative per object, so this serves inventories of the size the reference's fixtures have (an inventory that does
      // not fit a plan is refused: GK_ERR_UNSUPPORTED, the stock driver keeps the template); recompiled whenever they change
      if (!res_lock.owns_lock()) return fail(GK_ERR_INTERNAL, "the template of " + rec.kind + " changed while the constraint was added: try again");
      start_inventory_tracking(e);
      rec.referential = true;
      compile_referential(e, *it->second, rec);
    } else {
      const auto ta0 = std::chrono::steady_clock::now();
      PrepMemoScope memo;   // (the violation formula and the counting forms share their bodies node for node: simplified and folded once)
      const Template::CountInfo ci = it->second->compile_all(rec.params, &e->next_quant);
      rec.viol = ci.viol;
      const auto ta1 = std::chrono::steady_clock::now();
      rec.cforms = derive_count_forms(e, ci, rec.mf, &rec.viol, &rec.count_reads);
      const auto ta2 = std::chrono::steady_clock::now();
      if (getenv("GK_DEBUG_COUNTS")) {
        fprintf(stderr, "[gkgpu counts] %s/%s: %zu branches, forms %s\n", rec.kind.c_str(), rec.name.c_str(), ci.br.size(), rec.cforms ? "ok" : "none");
        for (auto& b : ci.br) fprintf(stderr, "   nq=%d keyed=%d const=%d head=%d pre=[%s] sep=[%s] tail=%d key=%s sig0=%s\n", b.nq, (int)b.keyed, (int)b.is_const, (int)b.head, b.pre.c_str(), b.sep.c_str(), (int)b.sep_tail,
                                    spath_to_string(b.key).c_str(), b.sig.empty() ? "" : b.sig[0].c_str());
        if (rec.cforms) fprintf(stderr, "   flag: %s\n", f_to_string(rec.cforms->flag).substr(0, 600).c_str());
      }
      // validate that it lowers (element scopes, register pressure) before accepting it
      {
        PlanBuilder pb(&e->dict, &e->dict_reg);
        rec.prep = prepare_constraint(rec.viol, rec.mf);
        pb.add_constraint(rec.prep);
        PlanCaps caps;
        pb.build(caps);
      }
      rec.multi_ready = false;   // (prepare_multi: on demand, ensure_totals_plans)
      if (getenv("GK_DEBUG_LOAD")) fprintf(stderr, "[gkgpu load] %s/%s: evaluate %.3f s, counting forms %.3f s, prepare + trial plan %.3f s\n", rec.kind.c_str(), rec.name.c_str(),
                                           std::chrono::duration<double>(ta1 - ta0).count(), std::chrono::duration<double>(ta2 - ta1).count(),
                                           std::chrono::duration<double>(std::chrono::steady_clock::now() - ta2).count());
    }
    for (auto& o : e->constraints) if (o.alive && o.kind == rec.kind && o.name == rec.name) o.alive = false;   // replace
    rec.id = (uint32_t)e->constraints.size();
    e->constraints.push_back(rec);
    e->plan_dirty = true;
    if (id_out) *id_out = rec.id;
    return GK_OK;
  } catch (const JsonError& ex) { return fail(GK_ERR_INVALID, ex.what());
  } catch (const RegoError& ex) { return fail(GK_ERR_REGO, ex.what());
  } catch (const Unsupported& ex) { return fail(GK_ERR_UNSUPPORTED, ex.what());
  } catch (const RegexUnsupported& ex) { return fail(GK_ERR_UNSUPPORTED, std::string("unsupported on the device plan: ") + ex.what());
  } catch (const std::exception& ex) { return fail(GK_ERR_INTERNAL, ex.what()); }
}

int gk_constraint_remove(gk_engine* e, const char* kind, const char* name) {
  if (!e || !kind || !name) return fail(GK_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> policy_turn(e->policy_gate);
  std::unique_lock<std::shared_mutex> policy_change(e->policy_rw);   // (no admission batch is between its flatten and its launch)
  std::unique_lock<std::shared_mutex> l(e->mu);
  bool found = false;
  for (auto& c : e->constraints) if (c.alive && c.kind == kind && c.name == name) { c.alive = false; found = true; }
  if (!found) return fail(GK_ERR_NOT_FOUND, std::string("unknown constraint ") + kind + "/" + name);
  e->plan_dirty = true;
  return GK_OK;
}

namespace {

uint64_t text_hash(const char* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++) { h ^= (uint8_t)p[i]; h *= 1099511628211ull; }
  return h ^ (h >> 29);
}

// an object leaves its slot (replaced, removed, or its Namespace changed): the slot is masked out of every answer
void resident_tombstone(gk_engine::Resident& R, gk_engine::ResObj& o) {
  if (o.chunk == UINT32_MAX) return;
  gk_engine::ResChunk& c = R.chunks[o.chunk];
  const uint64_t bit = 1ull << (o.slot % 64);
  if (c.live[o.slot / 64] & bit) { c.live[o.slot / 64] &= ~bit; c.n_live--; R.n_dead_slots++; }
  o.chunk = UINT32_MAX;
}

void resident_requeue(gk_engine::Resident& R, uint32_t id) {
  gk_engine::ResObj& o = R.objs[id];
  if (!o.alive) return;
  const bool queued = o.chunk == UINT32_MAX && std::find(R.pending.begin(), R.pending.end(), id) != R.pending.end();
  resident_tombstone(R, o);
  if (!queued) R.pending.push_back(id);
}

}  // namespace

int gk_data_put(gk_engine* e, const char* const* path, size_t npath, const char* json, size_t len) {
  if (!e || !path || !npath || !json) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    Value v = parse_json(json, len);
    std::vector<std::string> p;
    for (size_t i = 0; i < npath; i++) p.emplace_back(path[i]);
    const bool is_ns = v.is_object() && obj_is_namespace(v) && p.size() == 4 && p[0] == "cluster";
    {
      std::unique_lock<std::shared_mutex> l(e->mu);
      // nsCache.Add: cluster-scoped core/v1 Namespace objects (ns_cache.go:22-43)
      if (is_ns) e->ns_cache.put(p[3], v);
      if (e->inv_tracking) {   // data.inventory of the referential constraints
        std::string k;
        for (auto& x : p) { k += x; k.push_back('/'); }
        auto& slot = e->inv_store[k];
        if (slot.second.size() != len || memcmp(slot.second.data(), json, len) != 0) { slot = {p, std::string(json, len)}; e->inv_gen++; }
      }
    }
    // the resident set: this version of the object is flattened by the next gk_resident_sweep
    gk_engine::Resident& R = e->resident;
    std::lock_guard<std::mutex> rl(R.mu);
    std::string key;
    for (auto& x : p) { key += x; key.push_back('/'); }
    auto it = R.by_key.find(key);
    uint32_t id;
    if (it == R.by_key.end()) {
      id = (uint32_t)R.objs.size();
      R.objs.emplace_back();
      gk_engine::ResObj& o = R.objs.back();
      o.path = p; o.key = key;
      o.ns = (p.size() == 5 && p[0] == "namespace") ? p[1] : std::string();
      o.is_namespace = is_ns;
      R.by_key.emplace(key, id);
      if (!o.ns.empty()) R.by_ns[o.ns].push_back(id);
      R.n_live++;
    } else {
      id = it->second;
      gk_engine::ResObj& o = R.objs[id];
      if (o.alive && o.json.size() == len && memcmp(o.json.data(), json, len) == 0) return GK_OK;   // unchanged
      if (!o.alive) { o.alive = true; R.n_live++; }
      R.by_text.erase(text_hash(o.json.data(), o.json.size()));
    }
    gk_engine::ResObj& o = R.objs[id];
    o.json.assign(json, len);
    R.by_text[text_hash(json, len)] = id;
    resident_requeue(R, id);
    // a Namespace object is part of the review of every object living in it (Matchable.Namespace, namespaceObject)
    if (is_ns) { auto bn = R.by_ns.find(p[3]); if (bn != R.by_ns.end()) for (uint32_t d : bn->second) resident_requeue(R, d); }
    R.swept = false;
    return GK_OK;
  } catch (const JsonError& ex) { return fail(GK_ERR_INVALID, ex.what()); }
}

int gk_data_remove(gk_engine* e, const char* const* path, size_t npath) {
  if (!e || !path || !npath) return fail(GK_ERR_INVALID, "NULL argument");
  std::vector<std::string> p;
  for (size_t i = 0; i < npath; i++) p.emplace_back(path[i]);
  const bool is_ns = p.size() == 4 && p[0] == "cluster" && p[1] == "v1" && p[2] == "Namespace";
  {
    std::unique_lock<std::shared_mutex> l(e->mu);
    if (is_ns) e->ns_cache.remove(p[3]);
    if (e->inv_tracking) {
      std::string k;
      for (auto& x : p) { k += x; k.push_back('/'); }
      if (e->inv_store.erase(k)) e->inv_gen++;
    }
  }
  gk_engine::Resident& R = e->resident;
  std::lock_guard<std::mutex> rl(R.mu);
  std::string key;
  for (auto& x : p) { key += x; key.push_back('/'); }
  auto it = R.by_key.find(key);
  if (it != R.by_key.end() && R.objs[it->second].alive) {
    gk_engine::ResObj& o = R.objs[it->second];
    resident_tombstone(R, o);
    R.pending.erase(std::remove(R.pending.begin(), R.pending.end(), it->second), R.pending.end());
    R.by_text.erase(text_hash(o.json.data(), o.json.size()));
    o.alive = false; o.json.clear();
    R.n_live--;
    if (is_ns) { auto bn = R.by_ns.find(p[3]); if (bn != R.by_ns.end()) for (uint32_t d : bn->second) resident_requeue(R, d); }
    R.swept = false;
  }
  return GK_OK;
}

int gk_excluder_replace(gk_engine* e, const char* match_json, size_t len) {
  if (!e) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    static const char* all[] = {"audit", "webhook", "mutation-webhook", "sync"};   // allProcesses (excluder.go:31-36)
    std::map<std::string, std::vector<std::string>> next;
    auto put = [&](const std::string& proc, const std::string& ns) {
      auto& v = next[proc];
      if (std::find(v.begin(), v.end(), ns) == v.end()) v.push_back(ns);
    };
    if (match_json && len) {
      Value m = parse_json(match_json, len);
      if (!m.is_null() && !m.is_array()) return fail(GK_ERR_INVALID, "spec.match must be an array of {excludedNamespaces, processes}");
      if (m.is_array()) for (const Value& ent : m.items()) {   // Excluder.Add (excluder.go:52-76)
        if (!ent.is_object()) continue;
        const Value* nss = ent.get("excludedNamespaces");
        const Value* procs = ent.get("processes");
        if (!nss || !nss->is_array() || !procs || !procs->is_array()) continue;
        for (const Value& ns : nss->items()) {
          if (!ns.is_string()) continue;
          for (const Value& op : procs->items()) {
            if (!op.is_string()) continue;
            if (op.str() == "*") { for (const char* a : all) put(a, ns.str()); } else put(op.str(), ns.str());
          }
        }
      }
    }
    std::unique_lock<std::shared_mutex> l(e->mu);
    if (next != e->excluder) { e->excluder.swap(next); e->excluder_gen++; }   // Excluder.Replace
    return GK_OK;
  } catch (const JsonError& ex) { return fail(GK_ERR_INVALID, ex.what());
  } catch (const std::exception& ex) { return fail(GK_ERR_INTERNAL, ex.what()); }
}

int gk_excluder_excluded(gk_engine* e, const char* process, const gk_review_in* review, int32_t* excluded) {
  if (!e || !process || !review || !excluded) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    *excluded = 0;
    std::vector<std::string> pats;
    {
      std::shared_lock<std::shared_mutex> l(e->mu);
      auto it = e->excluder.find(process);
      if (it == e->excluder.end()) return GK_OK;
      pats = it->second;
    }
    Value body;
    try { body = parse_json(review->json, review->json_len); } catch (const std::exception&) { return GK_OK; }   // decode error: not excluded
    *excluded = review_excluded(pats, review->kind, body) ? 1 : 0;
    return GK_OK;
  } catch (const std::exception& ex) { return fail(GK_ERR_INTERNAL, ex.what()); }
}

// `prem` (may be NULL): per review, non-zero = the caller ran Matcher.Match itself (RF_PREMATCHED, plan.hpp); GK_TABLE_PRE_MATCHED = all of them
static int table_create_impl(gk_engine* e, const gk_review_in* reviews, size_t n, uint32_t flags, int32_t* statuses, const uint8_t* prem, gk_table** out);
int gk_table_create(gk_engine* e, const gk_review_in* reviews, size_t n, uint32_t flags, int32_t* statuses, gk_table** out) {
  return table_create_impl(e, reviews, n, flags, statuses, nullptr, out);
}
static int table_create_impl(gk_engine* e, const gk_review_in* reviews, size_t n, uint32_t flags, int32_t* statuses, const uint8_t* prem, gk_table** out) {
  if (!e || !out || (n && !reviews)) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    // referential constraints follow the synced inventory: compiled again -- and their dictionary expressions registered -- BEFORE
    // the rows are made, or the table would lack the <leaf>.$d rows the new plan reads (a plan that does not build is the
    // evaluation's error to report, not this call's)
    try { ensure_plan(e); } catch (const std::exception&) {}
    std::unique_ptr<gk_table> t(new gk_table());
    t->eng = e;
    bool keep = flags & GK_TABLE_KEEP_DOCS;
    t->review_errors.resize(n);
    t->key_span.assign(n, {0u, 0u});
    if (keep) t->docs.resize(n);
    if ((flags & GK_TABLE_KEEP_TEXT) && !keep) t->texts.assign(reviews, reviews + n);
    const auto t_begin = std::chrono::steady_clock::now();
    // Reviews are flattened by host threads, each on a contiguous range of whole tiles (the path dictionary is shared and
    // thread-safe).  Fast path: JSON text -> rows in one pass (Flattener::add_json); reviews it declines -- and every
    // review when the parsed documents must be kept for rendering -- go through parse_json + HandleReview normalisation +
    // Flattener::add, which produces the same rows.  The parts are then copied, in parallel, into the table's arrays.
    // reviews per row group: small batches (admission) keep 64-review groups; resident sets get large groups, whose
    // segments fill the lanes of the waves that stream them (plan.hpp / kernel_body.inc).  GK_RPT overrides (64|128|256|512).
    uint32_t rpt = n >= 8192 ? 256 : GK_RPT_MIN;
    if (rpt == 256) {
      // a large policy set (several plan groups, or accumulators too wide for three 256-review groups per CU) runs better on
      // 128-review groups: 4-wave workgroups of half the LDS footprint keep more waves resident and let the plan groups'
      // kernels share the CUs (configs[4]'s 200 templates: 0.81 against 1.11 ms per sweep, profiles/r03_variants_f_*.log)
      try {
        ensure_plan(e);
        std::shared_lock<std::shared_mutex> pl(e->plan_rw);
        if (!e->extra.empty() || (size_t)e->fast.dims.acc_words * 256 * 4 > 72 * 1024) rpt = 128;
      } catch (const std::exception&) {}
    }
    if (const char* rp = getenv("GK_RPT")) { int v = atoi(rp); if (v == 64 || v == 128 || v == 256 || v == 512) rpt = (uint32_t)v; }
    t->rpt = rpt;
    t->dict_gen = e->dict_reg.gen();
    t->pruned = ((flags & GK_TABLE_PRUNED) != 0 || getenv("GK_FORCE_PRUNE") != nullptr) && !getenv("GK_NO_PRUNE");   // (GK_FORCE_PRUNE: the test suites run every table pruned)
    t->reads_gen = e->dict_reg.reads_gen();
    const size_t n_tiles = (n + rpt - 1) / rpt;
    // host threads of a table build: at most 64 -- measured on the 256-thread GPU box (profiles/r03_phases_d_*.log): 1M objects
    // flatten in 0.45 s on 64 threads, 0.61 s on 128, 0.68 s on 256 (first-touch page faults and the shared dictionaries
    // contend; the GPU-side assembly does not care how many parts there are)
    // (ingest is a BURST: a 64 k-review batch is 20-25 ms of work on 64 threads, shorter than the 100 ms period of a cgroup CPU
    //  quota, so it may run wider than the quota -- measured on a 16-CPU-quota box: streaming 2.3-2.5 M reviews/s on 64 threads,
    //  1.8 M/s on 16; the 1 M-object build is the same either way.  Sustained host passes (gk_table_totals) size by host_cpus().)
    size_t n_threads = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(std::thread::hardware_concurrency(), 64), (n + 511) / 512));
    if (const char* ht = getenv("GK_HOST_THREADS")) n_threads = std::max(1, atoi(ht));
    n_threads = std::min(n_threads, std::max<size_t>(n_tiles, 1));
    const size_t tiles_per = (n_tiles + n_threads - 1) / std::max<size_t>(n_threads, 1);
    t->key_per_part = std::max<size_t>(1, tiles_per * rpt);
    t->key_arena.assign(std::max<size_t>(n_threads, 1), std::string());
    const bool slow_only = keep || getenv("GK_SLOW_INGEST") != nullptr;
    std::vector<HostTable> parts(n_threads);
    std::vector<std::string> part_err(n_threads);
    std::vector<uint64_t> part_fast(n_threads, 0);
    // process excluder: reviews of a table built for a process (audit sweep / validating webhook) whose namespace the
    // Config excludes for that process are skipped BEFORE evaluation -- pkg/audit/manager.go:530,599 (skipExcludedNamespace),
    // pkg/webhook/policy.go:197, pkg/webhook/common.go:149-189.  They keep their slot, hold no rows and report
    // GK_REVIEW_EXCLUDED.
    std::vector<std::string> excl;
    if (flags & (GK_TABLE_PROCESS_AUDIT | GK_TABLE_PROCESS_WEBHOOK)) {
      std::shared_lock<std::shared_mutex> l(e->mu);
      auto it = e->excluder.find((flags & GK_TABLE_PROCESS_AUDIT) ? "audit" : "webhook");
      if (it != e->excluder.end()) excl = it->second;
    }
    const Flattener::ExcludeFn excl_fn = [&](bool is_ns, const std::string& ns, const std::string& name) { return excluder_matches(excl, is_ns ? name : ns); };
    auto work = [&](size_t w) {
      try {
        // one Flattener per host thread and engine, kept across tables (Flattener::begin_table)
        thread_local std::unordered_map<uint64_t, std::unique_ptr<Flattener>> tl_flatteners;
        std::unique_ptr<Flattener>& slot = tl_flatteners[e->uid];
        if (!slot) { if (tl_flatteners.size() > 8) { tl_flatteners.clear(); } tl_flatteners[e->uid].reset(new Flattener(&e->dict, &e->dict_reg)); }
        Flattener& fl = *tl_flatteners[e->uid];
        fl.set_pruning(t->pruned);
        fl.begin_table();
        parts[w].rpt = rpt;
        const size_t lo = std::min(n, w * tiles_per * rpt), hi = std::min(n, (w + 1) * tiles_per * rpt);
        {
          // the part's arrays, sized ONCE from the JSON text it will flatten (a row per ~9 bytes of Kubernetes JSON, a
          // heap byte per ~5; a quarter on top): no growth by reallocation while 255 other threads do the same
          size_t jb = 0;
          for (size_t i = lo; i < hi; i++) jb += reviews[i].json_len + reviews[i].namespace_len / 4;
          if (jb >= (256u << 10)) {
            const size_t rows_est = jb / 7 + 4096;
            parts[w].rows.presize(rows_est);
            parts[w].shdr.presize(rows_est);
            parts[w].heap.presize(jb / 4 + 65536);
          }
        }
        std::string key_scratch;
        std::string& arena = t->key_arena[w];
        arena.reserve((hi - lo) * 40);
        auto put_key = [&](size_t i, const std::string& k) { t->key_span[i] = {(uint32_t)arena.size(), (uint32_t)k.size()}; arena += k; };
        // a review that may turn out to be beyond the device's limits keeps its text with the table (rare: a copy per such review)
        auto keep_if_big = [&](size_t i, const HostTable& part) {
          if (part.rflags.empty() || !(part.rflags.back() & (RF_TOO_BIG | RF_REFUSE | RF_HOST_CAND))) return;
          const gk_review_in& r = reviews[i];
          gk_table::OwnedReview o;
          o.kind = r.kind; o.source = r.source;
          o.json.assign(r.json ? r.json : "", r.json ? r.json_len : 0);
          if (r.namespace_json) { o.has_ns = true; o.ns.assign(r.namespace_json, r.namespace_len); }
          if (r.ns_object_json) { o.has_nsobj = true; o.nsobj.assign(r.ns_object_json, r.ns_object_len); }
          if (r.operation) { o.has_op = true; o.op = r.operation; }
          std::lock_guard<std::mutex> bl(t->big_mu);
          t->big_texts[(uint32_t)i] = std::move(o);
        };
        for (size_t i = lo; i < hi; i++) {
          const gk_review_in& r = reviews[i];
          if (i + 2 < hi && reviews[i + 2].json) {   // the text of the review after next: first touched by the scanner otherwise, a DRAM round trip per line
            const char* nj = reviews[i + 2].json;
            const size_t nl = std::min<size_t>(reviews[i + 2].json_len, 4096);
            for (size_t o = 0; o < nl; o += 64) __builtin_prefetch(nj + o, 0, 1);
          }
          if (!slow_only) {
            RawReview rr;
            rr.kind = r.kind; rr.source = r.source; rr.json = r.json; rr.json_len = r.json_len;
            rr.ns_json = r.namespace_json; rr.ns_len = r.namespace_len; rr.nsobj_json = r.ns_object_json; rr.nsobj_len = r.ns_object_len;
            rr.operation = r.operation;
            int rc = fl.add_json(rr, e->ns_cache, &parts[w], &key_scratch, excl.empty() ? nullptr : &excl_fn);
            if (rc == Flattener::ADDED) { put_key(i, key_scratch); if (statuses) statuses[i] = GK_OK; part_fast[w]++; keep_if_big(i, parts[w]); continue; }
            if (rc == Flattener::EXCLUDED) {
              if (statuses) statuses[i] = GK_REVIEW_EXCLUDED;
              fl.add_skipped(&parts[w]);
              part_fast[w]++;
              continue;
            }
          }
          ReviewDoc doc;
          int st = GK_OK;
          try {
            Value body = parse_json(r.json, r.json_len);
            Value mns = parse_opt(r.namespace_json, r.namespace_len);
            Value nso = parse_opt(r.ns_object_json, r.ns_object_len);
            if (!excl.empty() && review_excluded(excl, r.kind, body)) {
              if (statuses) statuses[i] = GK_REVIEW_EXCLUDED;
              doc.request = Value::object({});
              fl.add_skipped(&parts[w]);
              if (keep) t->docs[i] = doc;
              continue;
            }
            if (r.kind == GK_REVIEW_OBJECT) doc = normalize_object(body, mns, nso, r.source, r.operation ? r.operation : "", e->ns_cache);
            else doc = normalize_admission_request(body, mns, nso, r.source, e->ns_cache);
          } catch (const std::exception& ex) {
            st = GK_ERR_REVIEW;
            t->review_errors[i] = ex.what();
            doc = ReviewDoc();
            doc.request = Value::object({});
          }
          if (statuses) statuses[i] = st;
          {   // object identity in the order pkg/audit sorts violations by
            const Value* o = doc.request.get("object");
            if (!o || !o->is_object()) o = doc.request.get("oldObject");
            std::string g_, v_, k_, key;
            if (o && o->is_object()) {
              obj_gvk(*o, &g_, &v_, &k_);
              key = g_; key.push_back('\0'); key += v_; key.push_back('\0'); key += k_; key.push_back('\0');
              key += obj_string(*o, "metadata", "namespace"); key.push_back('\0'); key += obj_string(*o, "metadata", "name");
            }
            put_key(i, key);
          }
          if (st == GK_ERR_REVIEW) fl.add_skipped(&parts[w]);   // HandleReview's error is the caller's answer: nothing is evaluated
          else { fl.add(doc, &parts[w]); keep_if_big(i, parts[w]); }
          if (keep) t->docs[i] = doc;
        }
        fl.flush(&parts[w]);
      } catch (const std::exception& ex) { part_err[w] = ex.what(); }
    };
    auto run_threads = [&](const std::function<void(size_t)>& fn) {
      if (n_threads <= 1) { fn(0); return; }
      HostWorkers::get().run(n_threads, fn);
    };
    // content digest (test aid, GK_TABLE_DIGEST=1): per part while its rows are still on the host; parts are combined in
    // order, strings by value (not by heap offset), so the digest does not depend on the number of host threads
    const bool want_digest = getenv("GK_TABLE_DIGEST") != nullptr;
    std::vector<std::vector<uint64_t>> part_digest(n_threads);
    auto digest_part = [&](size_t w) {
      const HostTable& P = parts[w];
      for (size_t tl = 0; tl + 0 < P.tile_seg.size(); tl++) {
        uint64_t d = 1469598103934665603ull;
        auto mix = [&](uint64_t v) { d ^= v; d *= 1099511628211ull; d ^= d >> 31; };
        const uint32_t s0 = P.tile_seg[tl], s1 = tl + 1 < P.tile_seg.size() ? P.tile_seg[tl + 1] : (uint32_t)P.segs.size();
        for (uint32_t sg = s0; sg < s1; sg++) {
          const uint32_t a = P.segs[sg].start, b = sg + 1 < P.segs.size() ? P.segs[sg + 1].start : (uint32_t)P.rows.size();
          for (uint32_t i = a; i < b; i++) {
            const Row& r = P.rows[i];
            mix(P.segs[sg].path); mix(((uint64_t)r.rev << 32) | r.meta);
            if ((r.meta & ROW_TYPE_MASK) == T_STRING && !(r.meta & ROW_STR_INLINE)) {
              uint32_t len; memcpy(&len, &P.heap[r.lo - 4], 4);
              mix(len); mix(r.hi);
              for (uint32_t k = 0; k < len; k++) mix(P.heap[r.lo + k]);
              for (int k = 0; k < 4; k++) mix(P.shdr[i].w[k]);
            } else mix(((uint64_t)r.hi << 32) | r.lo);
          }
        }
        part_digest[w].push_back(d);
      }
    };
    std::vector<DevPart*> dev_parts(n_threads, nullptr);
    std::vector<double> part_upload_s(n_threads, 0);
    run_threads([&](size_t w) {
      work(w);
      if (!part_err[w].empty()) return;
      try {
        if (want_digest) digest_part(w);
        const auto u0 = std::chrono::steady_clock::now();
        dev_parts[w] = dev_part_upload(e->opts.device, parts[w]);   // each thread ships its part as soon as it is flattened
        part_upload_s[w] = std::chrono::duration<double>(std::chrono::steady_clock::now() - u0).count();
      } catch (const std::exception& ex) { part_err[w] = ex.what(); }
    });
    for (auto& pe_ : part_err) if (!pe_.empty()) { for (DevPart* dp : dev_parts) dev_part_free(dp); return fail(GK_ERR_INTERNAL, pe_); }
    // ---- what is global: row / heap bases by prefix sum -> segment starts, the slot index, review flags
    HostTable& H = t->host;
    H.rpt = rpt;
    {
      size_t rb = 0, hb = 0;
      for (size_t w = 0; w < n_threads; w++) {
        HostTable& P = parts[w];
        const uint32_t sb = (uint32_t)H.segs.size();
        for (const auto& sg : P.segs) H.segs.push_back({sg.path, (uint32_t)(sg.start + rb)});
        for (uint32_t ts : P.tile_seg) H.tile_seg.push_back(ts + sb);
        H.rflags.insert(H.rflags.end(), P.rflags.begin(), P.rflags.end());
        if (P.path_rows.size() > H.path_rows.size()) H.path_rows.resize(P.path_rows.size(), 0);
        for (size_t i = 0; i < P.path_rows.size(); i++) H.path_rows[i] += P.path_rows[i];
        if (P.path_max.size() > H.path_max.size()) H.path_max.resize(P.path_max.size(), 0);
        for (size_t i = 0; i < P.path_max.size(); i++) H.path_max[i] = std::max(H.path_max[i], P.path_max[i]);
        rb += P.n_rows_total; hb += P.heap_total;
        t->stats.fast_reviews += part_fast[w];
        H.n_reviews += P.n_reviews;
      }
      if (rb >= 0xFFFFFFF0ull || hb >= 0xFFFFFFF0ull) {
        for (DevPart* dp : dev_parts) dev_part_free(dp);
        return fail(GK_ERR_INVALID, "table too large for 32-bit row/heap offsets: split the batch");
      }
      t->n_rows = rb;
      t->stats.heap_bytes = hb;
      H.n_rows_total = rb;
    }
    if (prem || (flags & GK_TABLE_PRE_MATCHED))
      for (size_t i = 0; i < n && i < H.rflags.size(); i++) if ((flags & GK_TABLE_PRE_MATCHED) || prem[i]) { H.rflags[i] |= RF_PREMATCHED; t->any_prematched = true; }
    Flattener::build_index(&t->host);
    if (want_digest) {
      uint64_t d = 1469598103934665603ull;
      auto mix = [&](uint64_t v) { d ^= v; d *= 1099511628211ull; d ^= d >> 31; };
      for (auto& pd : part_digest) for (uint64_t x : pd) mix(x);
      for (uint32_t f : H.rflags) mix(f);
      for (uint32_t sp : H.slot_path) mix(sp);
      for (size_t i = 0; i < n; i++) for (unsigned char c : t->obj_key(i)) mix(c);
      t->stats.digest = d;
    }
    const auto t_indexed = std::chrono::steady_clock::now();
    t->dev = dev_table_assemble(e->opts.device, dev_parts, t->host);
    const auto t_up = std::chrono::steady_clock::now();
    t->n_reviews = (uint32_t)n;
    t->dir_bytes = t->host.rflags.size() * 4;
    t->slot_path = t->host.slot_path;
    t->path_rows = t->host.path_rows;
    t->path_max = t->host.path_max;
    t->resident = (flags & GK_TABLE_RESIDENT) != 0;
    t->stats.n_reviews = n; t->stats.n_rows = t->n_rows; t->stats.host_threads = (uint32_t)n_threads;
    t->stats.device_bytes = dev_table_bytes(t->dev);
    // the parts' transfers overlap the flattening of the other host threads: upload_s is what is NOT hidden (slot index,
    // review flags, placing the parts on the device), flatten_s the rest of the wall clock
    t->stats.upload_s = std::chrono::duration<double>(t_up - t_indexed).count();
    t->stats.flatten_s = std::chrono::duration<double>(t_indexed - t_begin).count();
    for (size_t i = 0; i < n; i++) t->stats.json_bytes += reviews[i].json_len;
    t->host.tile_idx.clear(); t->host.tile_idx.shrink_to_fit();
    *out = t.release();
    return GK_OK;
  } catch (const std::exception& ex) { return fail(GK_ERR_DEVICE, ex.what()); }
}

// ------------------------------------------------------------------------------------------------ audit spool (row f4)
namespace {
struct SpoolHolder {
  gk_spool_info pub;   // first member
  std::vector<std::string> names;
  std::vector<const char*> ptrs;
  std::vector<int32_t> statuses;
};
bool read_file(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  out->clear();
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  const bool ok = !ferror(f);
  fclose(f);
  return ok;
}
}  // namespace

int gk_table_create_spool(gk_engine* e, const char* api_cache_dir, const char* kind, uint32_t folders, uint32_t flags,
                          gk_spool_info** info, gk_table** out) {
  if (!e || !api_cache_dir || !kind || !out) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    std::unique_ptr<SpoolHolder> h(new SpoolHolder());
    memset(&h->pub, 0, sizeof h->pub);
    // 1. the files, folder by folder (reviewObjects, manager.go:676-686), numeric order within a folder
    struct Item { std::string name, text, ns_text; };
    std::vector<Item> items;
    for (uint32_t fo = 0; fo < folders; fo++) {
      const std::string sub = std::string(kind) + "_" + std::to_string(fo), dir = std::string(api_cache_dir) + "/" + sub;
      std::vector<std::string> files;
      if (DIR* d = opendir(dir.c_str())) {
        while (dirent* de = readdir(d)) { if (de->d_name[0] != '.') files.emplace_back(de->d_name); }
        closedir(d);
      } else h->pub.n_folders_missing++;   // (a folder that cannot be opened: "Unable to get files from directory", the loop goes on)
      std::sort(files.begin(), files.end(), [](const std::string& a, const std::string& b) { return a.size() != b.size() ? a.size() < b.size() : a < b; });
      for (const std::string& fn : files) {
        h->pub.n_files++;
        Item it;
        it.name = sub + "/" + fn;
        if (!read_file(dir + "/" + fn, &it.text)) { h->pub.n_unreadable++; continue; }
        h->pub.bytes += it.text.size();
        items.push_back(std::move(it));
      }
    }
    // 2. the Namespace of every object, from the driver's cache (nsCache.Get, manager.go:694-704): looked up by metadata.namespace
    std::unordered_map<std::string, std::string> ns_json;   // namespace name -> its JSON ("" = not cached)
    std::vector<Item> kept;
    for (Item& it : items) {
      std::string nsname;
      try {
        Value v = parse_json(it.text.data(), it.text.size());
        if (!v.is_object()) { h->pub.n_unreadable++; continue; }
        nsname = obj_string(v, "metadata", "namespace");
      } catch (const std::exception&) { h->pub.n_unreadable++; continue; }   // readUnstructured fails: logged, next file
      if (!nsname.empty()) {
        auto f = ns_json.find(nsname);
        if (f == ns_json.end()) {
          Value ns = e->ns_cache.get(nsname);
          f = ns_json.emplace(nsname, ns.defined() ? to_json(ns) : std::string()).first;
        }
        if (f->second.empty()) { h->pub.n_namespace_missing++; continue; }   // "Unable to look up object namespace": skipped
        it.ns_text = f->second;
      }
      kept.push_back(std::move(it));
    }
    // 3. one table of AugmentedUnstructured{Object, Namespace, Source: Original} + the namespaceObject option
    std::vector<gk_review_in> rins(kept.size());
    for (size_t i = 0; i < kept.size(); i++) {
      gk_review_in& r = rins[i];
      memset(&r, 0, sizeof r);
      r.kind = GK_REVIEW_OBJECT;
      r.source = GK_SRC_ORIGINAL;
      r.json = kept[i].text.data(); r.json_len = kept[i].text.size();
      if (!kept[i].ns_text.empty()) {
        r.namespace_json = kept[i].ns_text.data(); r.namespace_len = kept[i].ns_text.size();
        r.ns_object_json = kept[i].ns_text.data(); r.ns_object_len = kept[i].ns_text.size();
      }
      h->names.push_back(kept[i].name);
    }
    h->statuses.assign(std::max<size_t>(1, kept.size()), GK_OK);
    int rc = gk_table_create(e, rins.data(), rins.size(), flags, h->statuses.data(), out);
    if (rc != GK_OK) return rc;
    h->statuses.resize(kept.size());
    h->pub.n_reviews = kept.size();
    h->pub.statuses = h->statuses.data();
    // (HandleReview's verdict per spooled object, as gk_table_create reports it: the reference logs "Unable to review object from
    //  file" for a rejected one and goes on; GK_REVIEW_EXCLUDED = the process excluder skipped it)
    for (int32_t st : h->statuses) { if (st == GK_REVIEW_EXCLUDED) h->pub.n_excluded++; else if (st != GK_OK) h->pub.n_rejected++; }
    // GK_TABLE_KEEP_TEXT remembers WHERE the text lives: here it lives in `kept`, so the table takes it over (round-3 advisor
    // finding: gk_table_totals / gk_render read freed memory).  Moving the vector keeps every string where it is.
    if ((*out)->texts.size() == kept.size() && !kept.empty()) (*out)->owned_text = std::make_shared<std::vector<Item>>(std::move(kept));
    for (auto& nm : h->names) h->ptrs.push_back(nm.c_str());
    h->pub.names = h->ptrs.data();
    if (info) *info = &h.release()->pub;
    return GK_OK;
  } catch (const std::exception& ex) { return fail(GK_ERR_INTERNAL, ex.what()); }
}

void gk_spool_info_free(gk_spool_info* info) { if (info) delete reinterpret_cast<SpoolHolder*>(info); }

int gk_table_get_stats(const gk_table* t, gk_table_stats* out) {
  if (!t || !out) return fail(GK_ERR_INVALID, "NULL argument");
  *out = t->stats;
  return GK_OK;
}

void gk_table_free(gk_table* t) {
  if (!t) return;
  for (DevTable* v : t->views) dev_table_free(v);
  for (DevTable* v : t->tviews) dev_table_free(v);
  dev_table_free(t->dev);
  delete t;
}

struct EvalHolder {
  gk_eval_out pub;   // first member: gk_eval_free recovers the holder from the public pointer
  uint32_t lds_bytes = 0;
  EvalOut out;
  std::vector<uint32_t> ids;
  std::vector<uint32_t> host_evaluated;   // reviews beyond the device's limits that the host evaluator answered (complete_on_host)
};
static void complete_on_host(gk_engine* e, gk_table* t, EvalHolder* h, bool want_match, bool want_list);

int gk_table_eval(gk_engine* e, gk_table* t, uint32_t flags, gk_eval_out** out) {
  if (!e || !t || !out) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    ensure_plan(e);
    if (t->dict_gen != e->dict_reg.gen())
      return fail(GK_ERR_INVALID, "the table was flattened before a constraint with dictionary predicates was added (its <leaf>.$d rows are missing): create it again");
    if (t->pruned && t->reads_gen != e->dict_reg.reads_gen())
      return fail(GK_ERR_INVALID, "the pruned table was flattened before a constraint that reads other key paths was added (GK_TABLE_PRUNED): create it again");
    std::unique_ptr<EvalHolder> h(new EvalHolder());
    EvalOptions opt;
    opt.download = !(flags & GK_EVAL_NO_DOWNLOAD);
    opt.want_match = flags & GK_EVAL_WANT_MATCH;
    opt.time_each = (flags & GK_EVAL_TIME_EACH) != 0;
    opt.kernel_only = (flags & GK_EVAL_KERNEL_ONLY) != 0 && (flags & GK_EVAL_ASYNC) != 0;
    // an admission batch (small, evaluated once) never waits for a compiler; an audit-sized or resident table does
    opt.jit_wait = t->resident || t->n_reviews >= 8192;
    {
      { std::lock_guard<std::mutex> gate(e->plan_gate); }   // (a plan change that is waiting goes first)
      std::shared_lock<std::shared_mutex> l(e->plan_rw);
      h->ids = e->plan_ids;
      if (flags & GK_EVAL_WANT_LIST) opt.list_capacity = std::max<uint32_t>(1024, t->n_reviews * 4u);
      const HostPlan* hp = nullptr;
      DevPlan* dp = nullptr;
      { std::lock_guard<std::mutex> vl(e->variants_mu); dp = plan_for_table(e, t, &hp); }
      while (t->views.size() < e->extra.size()) t->views.push_back(dev_table_view(t->dev));
      std::vector<DevPlan*> gdev(e->extra.size(), nullptr);   // the further plan groups: table-sized variants as well
      { std::lock_guard<std::mutex> vl(e->variants_mu); for (size_t gi = 0; gi < e->extra.size(); gi++) { const HostPlan* gh = nullptr; gdev[gi] = plan_for_group(e, t, gi + 1, &gh); } }
      // every plan group is LAUNCHED before any is collected: the groups work on their own streams (views of the table) and
      // overlap on the device as far as their footprints allow
      if (!(flags & GK_EVAL_COLLECT)) {
        // (several plan groups: their plan-specialised builds are all started first and compile side by side)
        if (!e->extra.empty() && opt.jit_wait) { dev_jit_prefetch(dp, t->dev); for (size_t gi = 0; gi < e->extra.size(); gi++) dev_jit_prefetch(gdev[gi], t->views[gi]); }
        dev_eval_launch(dp, t->dev, opt);
        for (size_t gi = 0; gi < e->extra.size(); gi++) dev_eval_launch(gdev[gi], t->views[gi], opt);
      }
      if (flags & GK_EVAL_ASYNC) {   // enqueue only; a later call without GK_EVAL_ASYNC collects
        *out = nullptr;
        return GK_OK;
      }
      dev_eval_finish(dp, t->dev, opt, &h->out);
      h->lds_bytes = h->out.lds_bytes;
      // further plan groups: same table, their rows are appended below the primary group's
      for (size_t gi = 0; gi < e->extra.size(); gi++) {
        EvalOut og;
        dev_eval_finish(gdev[gi], t->views[gi], opt, &og);
        const uint32_t row_base = h->out.n_constraints;
        EvalOut& o = h->out;
        o.viol.insert(o.viol.end(), og.viol.begin(), og.viol.end());
        o.err.insert(o.err.end(), og.err.begin(), og.err.end());
        o.match.insert(o.match.end(), og.match.begin(), og.match.end());
        o.counts.insert(o.counts.end(), og.counts.begin(), og.counts.end());
        for (size_t k = 0; k + 1 < og.list.size(); k += 2) { o.list.push_back(og.list[k] + row_base); o.list.push_back(og.list[k + 1]); }
        for (size_t k = 0; k < og.too_big.size() && k < o.too_big.size(); k++) o.too_big[k] |= og.too_big[k];
        o.list_total += og.list_total;
        o.n_overflow += og.n_overflow;
        o.kernel_ms += og.kernel_ms; o.fast_kernel_ms += og.fast_kernel_ms;
        o.list_bytes += og.list_bytes;
        o.kernel_hash = o.kernel_hash * 1099511628211ull ^ og.kernel_hash;   // (several plan groups: one figure that names all their texts)
        o.n_constraints += og.n_constraints;
        o.d_viol = o.d_err = o.d_counts = nullptr;   // not one contiguous device buffer any more
        h->ids.insert(h->ids.end(), e->extra[gi]->ids.begin(), e->extra[gi]->ids.end());
      }
    }
    // reviews the device could not evaluate (too_big) are answered by the engine's own exact evaluator where their text is at hand
    if (opt.download && !(flags & GK_EVAL_DEVICE_ONLY)) complete_on_host(e, t, h.get(), opt.want_match, (flags & GK_EVAL_WANT_LIST) != 0);
    else if (opt.download) { std::lock_guard<std::mutex> hl(t->host_mu); t->host_viol.clear(); t->host_err.clear(); t->host_ids.clear(); }
    gk_eval_out& p = h->pub;
    memset(&p, 0, sizeof p);
    p.n_host_evaluated = (uint32_t)h->host_evaluated.size();
    p.host_evaluated = h->host_evaluated.empty() ? nullptr : h->host_evaluated.data();
    p.n_reviews = h->out.n_reviews; p.n_constraints = h->out.n_constraints; p.n_tiles = h->out.n_tiles;
    p.constraint_ids = h->ids.data();
    p.viol = h->out.viol.data(); p.err = h->out.err.data();
    p.match = h->out.match.empty() ? nullptr : h->out.match.data();
    p.too_big = h->out.too_big.data();
    p.counts = h->out.counts.data();
    p.list = h->out.list.data(); p.list_len = (uint32_t)(h->out.list.size() / 2); p.list_total = h->out.list_total;
    p.n_overflow = h->out.n_overflow;
    p.kernel_ms = h->out.kernel_ms; p.fast_kernel_ms = h->out.fast_kernel_ms; p.n_launches = h->out.n_launches;
    p.d_viol = h->out.d_viol; p.d_err = h->out.d_err; p.d_counts = h->out.d_counts;
    p.n_rows = t->n_rows;
    t->last_nc = p.n_constraints; t->last_ids = h->ids;
    p.lds_bytes = h->lds_bytes;
    p.kernel_text_hash = h->out.kernel_hash;
    // algorithmic bytes: rows of the segments whose path carries predicates (+ their string headers
    // where a predicate reads string bytes) + the groups' chunk lists (8 B per entry) + review flags, all read once;
    // plan tables read once; bitmaps written once; 8 B per list entry
    uint64_t rows_read = 0, hdrs_read = 0, bound = 0, plan_bytes = 0;
    std::vector<uint8_t> seen(t->slot_path.size(), 0);   // per slot: 1 = rows counted, 2 = string headers counted (table-once figure)
    uint64_t rows_once = 0, hdrs_once = 0;
    auto account = [&](const HostPlan& hp) {   // every plan group streams its own bound segments
      for (size_t si = 0; si < t->slot_path.size(); si++) {
        const uint32_t pth = t->slot_path[si];
        if (pth >= hp.ptab.size() || !hp.ptab[pth]) continue;
        bound++;
        uint64_t n = pth < t->path_rows.size() ? t->path_rows[pth] : 0;
        rows_read += n;
        uint32_t ent = hp.ptab[pth];
        bool str = false;
        for (uint32_t j = 0; j < (ent & 0xFF); j++) str = str || pred_needs_str(hp.path_preds[(ent >> 8) + j]);
        if (str) hdrs_read += n;
        static const bool dbg_path_rows = getenv("GK_DEBUG_PATH_ROWS") != nullptr;   // (read once)
        if (dbg_path_rows) fprintf(stderr, "[gkgpu paths] %-70s rows %9llu preds %3u%s\n", e->dict.to_string(pth).c_str(), (unsigned long long)n, ent & 0xFF, str ? " +hdr" : "");
        if (!(seen[si] & 1)) { seen[si] |= 1; rows_once += n; }
        if (str && !(seen[si] & 2)) { seen[si] |= 2; hdrs_once += n; }
      }
      plan_bytes += (uint64_t)hp.path_preds.size() * sizeof(Pred) + hp.code.size() * 4 + hp.cheap.size();
    };
    {
      std::shared_lock<std::shared_mutex> l(e->plan_rw);
      account(e->fast);
      for (auto& g : e->extra) account(g->fast);
    }
    p.n_rows_read = rows_read;
    (void)bound;
    p.algo_bytes = rows_read * sizeof(Row) + hdrs_read * sizeof(StrHdr) + h->out.list_bytes +
                   t->dir_bytes * (1 + e->extra.size()) + plan_bytes + (uint64_t)p.n_constraints * p.n_tiles * 16 + (uint64_t)p.list_total * 8;
    p.algo_bytes_once = p.algo_bytes - (rows_read - rows_once) * sizeof(Row) - (hdrs_read - hdrs_once) * sizeof(StrHdr) - t->dir_bytes * e->extra.size();
    p.n_plan_groups = 1 + (uint32_t)e->extra.size();
    *out = &h.release()->pub;
    return GK_OK;
  } catch (const Unsupported& ex) { return fail(GK_ERR_UNSUPPORTED, ex.what());
  } catch (const std::exception& ex) { return fail(GK_ERR_DEVICE, ex.what()); }
}

struct TopkHolder {
  gk_topk_out pub;   // first member
  std::vector<uint32_t> ids, counts, reviews, overflow;
};

int gk_table_topk(gk_engine* e, gk_table* t, uint32_t k, gk_topk_out** out) {
  if (!e || !t || !out) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    if (t->order.size() != t->n_reviews) {   // object-key order of the table, once
      t->order.resize(t->n_reviews);
      for (uint32_t i = 0; i < t->n_reviews; i++) t->order[i] = i;
      // field-wise order == order of the \0-joined keys (\0 sorts below every byte a name can contain)
      std::stable_sort(t->order.begin(), t->order.end(), [&](uint32_t a, uint32_t b) { return t->obj_key(a) < t->obj_key(b); });
      t->grp.resize(t->n_reviews);
      uint32_t gid = 0;
      for (uint32_t p = 0; p < t->n_reviews; p++) {
        if (p && t->obj_key(t->order[p]) != t->obj_key(t->order[p - 1])) gid++;
        t->grp[p] = gid;
      }
    }
    std::unique_ptr<TopkHolder> h(new TopkHolder());
    const uint32_t cap = k + 64;
    h->ids = t->last_ids;
    {
      std::shared_lock<std::shared_mutex> l(e->plan_rw);
      const uint32_t nc0 = (uint32_t)e->plan_ids.size();
      dev_topk(t->dev, nc0, t->order, t->grp, k, cap, &h->reviews, &h->counts, &h->overflow);
      for (size_t gi = 0; gi < e->extra.size() && gi < t->views.size(); gi++) {   // further plan groups: rows appended
        std::vector<uint32_t> r, c, o;
        dev_topk(t->views[gi], (uint32_t)e->extra[gi]->ids.size(), t->order, t->grp, k, cap, &r, &c, &o);
        h->reviews.insert(h->reviews.end(), r.begin(), r.end());
        h->counts.insert(h->counts.end(), c.begin(), c.end());
        h->overflow.insert(h->overflow.end(), o.begin(), o.end());
      }
    }
    // (violating pairs the host evaluation of the last gk_table_eval added are no part of the device bitmaps: they join the candidates
    //  -- a candidate too many costs a rendering, the host sorts and cuts the lists anyway; a full row says "walk the bitmap row")
    std::vector<std::pair<uint32_t, uint32_t>> hv;
    { std::lock_guard<std::mutex> hl(t->host_mu); if (t->host_ids == h->ids) hv = t->host_viol; }   // (another policy set's pairs: dropped)
    if (!hv.empty()) {
      std::vector<std::unordered_set<uint32_t>> have(h->counts.size());   // (a set per touched row, not a linear search per pair)
      for (auto& hp : hv) {
        const uint32_t row = hp.first;
        if (row >= h->counts.size()) continue;
        uint32_t* lst = &h->reviews[(size_t)row * cap];
        if (have[row].empty()) have[row].insert(lst, lst + h->counts[row]);
        if (!have[row].insert(hp.second).second) continue;
        if (h->counts[row] < cap) lst[h->counts[row]++] = hp.second; else h->overflow[row] = 1;
      }
    }
    h->pub.n_constraints = t->last_nc; h->pub.stride = cap;
    h->pub.constraint_ids = h->ids.data(); h->pub.counts = h->counts.data(); h->pub.reviews = h->reviews.data(); h->pub.overflow = h->overflow.data();
    *out = &h.release()->pub;
    return GK_OK;
  } catch (const std::exception& ex) { return fail(GK_ERR_DEVICE, ex.what()); }
}

struct TotalsHolder {
  gk_totals_out pub;   // first member
  std::vector<uint32_t> ids;
  std::vector<uint64_t> results, pairs;
};

// Result-level totals.  The device answers "does (constraint, review) violate" (one bit); the reference counts RESULTS
// (pkg/audit/manager.go:902: totalViolationsPerConstraint[key]++ per types.Result), and a violating pair yields as many
// results as the template's violation set has distinct {msg, details} members for that review.  The violating pairs of
// (the review document of table slot r: kept (GK_TABLE_KEEP_DOCS) or parsed now from the caller's text (GK_TABLE_KEEP_TEXT))
static ReviewDoc make_doc(gk_engine* e, const gk_review_in& in) {
  Value body = parse_json(in.json, in.json_len);
  Value mns = parse_opt(in.namespace_json, in.namespace_len);
  Value nso = parse_opt(in.ns_object_json, in.ns_object_len);
  if (in.kind == GK_REVIEW_OBJECT) return normalize_object(body, mns, nso, in.source, in.operation ? in.operation : "", e->ns_cache);
  return normalize_admission_request(body, mns, nso, in.source, e->ns_cache);
}
static const ReviewDoc* doc_for(gk_engine* e, const gk_table* t, uint32_t r, ReviewDoc* tmp) {
  if (r < t->docs.size()) return &t->docs[r];
  if (r < t->texts.size()) { *tmp = make_doc(e, t->texts[r]); return tmp; }
  auto it = t->big_texts.find(r);   // (written while the table was built, read-only afterwards)
  if (it != t->big_texts.end()) { const gk_review_in in = it->second.in(); *tmp = make_doc(e, in); return tmp; }
  return nullptr;
}

// ---- exact host evaluation of the reviews the device refuses (round 5).  A review with more than 255 elements in an iterated
// array, an object where predicates iterate array elements, or a non-empty container in a value join sets its bit in too_big:
// the kernels never guess.  The reference has no such limit (its audit loop reviews every object, pkg/audit/manager.go:591-642),
// and the engine owns an exact evaluator (ceval.cpp: the one that renders the messages).  So, for every too_big review whose
// text the table holds (kept at flatten time for exactly these reviews):
//   * the MATCH layer is answered by the device from a STRIPPED copy of the review -- object / oldObject cut down to apiVersion,
//     kind and metadata, which is all match.Matches reads (pkg/mutation/match/match.go:32-258) -- flattened into a one-review table
//     and evaluated with GK_EVAL_WANT_MATCH: the same compiled formulas, the same error conditions;
//   * for every constraint that matches, the template's violation set is evaluated on the whole document: a bit per non-empty set.
// The review's too_big bit is cleared and its index reported in gk_eval_out.host_evaluated.  Anything that goes wrong on the way
// (no text, the stripped review refused as well, an evaluation error, a policy change in between) leaves the review in too_big:
// fail closed, as before.
static Value strip_object(const Value& o) {
  if (!o.is_object()) return o;
  ValuePairs keep;
  for (auto& kv : o.pairs()) if (kv.first.is_string() && (kv.first.str() == "apiVersion" || kv.first.str() == "kind" || kv.first.str() == "metadata")) keep.push_back(kv);
  return Value::object(std::move(keep));
}
// (the nested evaluation of a stripped review must never complete on the host again: a stripped review that is still beyond the
//  device's limits -- 300 metadata.ownerReferences iterated by an element-scoped rule -- strips to itself and would recurse without
//  bound; with the guard it takes the "refused as well" branch below and the review stays in too_big: fail closed)
static thread_local int tl_host_completion_depth = 0;
// at most this many refused reviews are completed per evaluation; the rest stay in too_big (fail closed) -- the host evaluator is
// the exception path, not a second engine: a table of a million 300-container pods belongs on the reference driver
static constexpr size_t GK_HOST_COMPLETIONS_MAX = 4096;
static void complete_on_host(gk_engine* e, gk_table* t, EvalHolder* h, bool want_match, bool want_list) {
  EvalOut& o = h->out;
  static const bool env_off = [] { const char* off = getenv("GK_HOST_EVAL"); return off && atoi(off) == 0; }();   // (GK_HOST_EVAL=0, read once: the device's answer alone; per call: GK_EVAL_DEVICE_ONLY)
  static const bool dbg = getenv("GK_DEBUG_HOST") != nullptr;
  auto why = [&](uint32_t r, const char* what) { if (dbg) fprintf(stderr, "[gkgpu host] review %u stays refused: %s\n", r, what); };
  // the pairs this pass adds belong to THIS evaluation: collected locally, published under the table's lock with the constraint-id
  // list they index (gk_table_totals / gk_table_topk drop them when their own list differs)
  std::vector<std::pair<uint32_t, uint32_t>> add_viol, add_err;
  struct Publish {
    gk_table* t; EvalHolder* h; std::vector<std::pair<uint32_t, uint32_t>>*v, *er;
    ~Publish() { std::lock_guard<std::mutex> hl(t->host_mu); t->host_viol.swap(*v); t->host_err.swap(*er); t->host_ids = h->ids; }
  } publish{t, h, &add_viol, &add_err};
  if (env_off || tl_host_completion_depth > 0) return;
  struct Depth { Depth() { tl_host_completion_depth++; } ~Depth() { tl_host_completion_depth--; } } depth_guard;
  const uint32_t nt = o.n_tiles, nc = o.n_constraints;
  if (nc == 0 || h->ids.size() != nc) return;
  // ---- the candidates: refused reviews whose text is at hand
  struct Cand { uint32_t r; gk_review_in in; bool prem; std::string stext; int mrow = -1; };
  std::vector<Cand> cands;
  for (uint32_t w = 0; w < nt && w < o.too_big.size(); w++)
    for (uint64_t m = o.too_big[w]; m; m &= m - 1) {
      const uint32_t r = w * GK_TILE + (uint32_t)__builtin_ctzll(m);
      if (r >= t->n_reviews) continue;
      if (cands.size() >= GK_HOST_COMPLETIONS_MAX) { why(r, "more refused reviews than one evaluation completes"); continue; }
      Cand c;
      c.r = r;
      if (r < t->texts.size()) c.in = t->texts[r];
      else { auto it = t->big_texts.find(r); if (it == t->big_texts.end()) { why(r, "no text kept"); continue; } c.in = it->second.in(); }
      c.prem = r < t->host.rflags.size() && (t->host.rflags[r] & RF_PREMATCHED) != 0;
      cands.push_back(std::move(c));
    }
  if (cands.empty()) return;
  // ---- 1. match layer: the stripped reviews on the device, ONE table and ONE launch for all of them -- unless the caller ran
  //         Matcher.Match itself (RF_PREMATCHED: every constraint counts as matching, nothing is autorejected)
  std::vector<gk_review_in> sins;
  for (Cand& c : cands) {
    if (c.prem) continue;
    try {
      Value body = parse_json(c.in.json, c.in.json_len);
      if (!body.is_object()) { why(c.r, "not an object"); continue; }
      Value stripped;
      if (c.in.kind == GK_REVIEW_OBJECT) stripped = strip_object(body);
      else {
        ValuePairs kv = body.pairs();
        for (auto& pr : kv) if (pr.first.is_string() && (pr.first.str() == "object" || pr.first.str() == "oldObject")) pr.second = strip_object(pr.second);
        stripped = Value::object(std::move(kv));
      }
      c.stext = to_json(stripped);
      c.mrow = (int)sins.size();
      sins.push_back(c.in);
    } catch (const std::exception& ex) { why(c.r, ex.what()); }
  }
  for (Cand& c : cands) if (c.mrow >= 0) { sins[c.mrow].json = c.stext.data(); sins[c.mrow].json_len = c.stext.size(); }   // (after the loop: the strings no longer move)
  gk_table* mt = nullptr;
  gk_eval_out* mo = nullptr;
  std::vector<int32_t> mst(sins.size(), GK_OK);
  std::unique_ptr<gk_table, void (*)(gk_table*)> mt_guard(nullptr, gk_table_free);
  std::unique_ptr<gk_eval_out, void (*)(gk_eval_out*)> mo_guard(nullptr, gk_eval_free);
  bool match_ok = sins.empty();
  if (!sins.empty()) {
    if (gk_table_create(e, sins.data(), sins.size(), 0, mst.data(), &mt) == GK_OK && mt) {
      mt_guard.reset(mt);
      if (gk_table_eval(e, mt, GK_EVAL_WANT_MATCH | GK_EVAL_DEVICE_ONLY, &mo) == GK_OK && mo) {
        mo_guard.reset(mo);
        match_ok = mo->n_constraints == nc && mo->match != nullptr;
        for (uint32_t row = 0; row < nc && match_ok; row++) match_ok = mo->constraint_ids[row] == h->ids[row];   // (else: the policy set changed between the two evaluations)
      }
    }
    if (!match_ok && dbg) fprintf(stderr, "[gkgpu host] the stripped reviews' match table failed: %s\n", gk_last_error());
  }
  // ---- 2. violations: the template's violation set on the whole document, for the constraints that match; 3. the answer replaces the refusal
  for (Cand& c : cands) {
    const uint32_t r = c.r, w = r / GK_TILE;
    const uint64_t bit = 1ull << (r % GK_TILE);
    try {
      std::vector<uint8_t> v(nc, 0), er(nc, 0), ma(nc, c.prem ? 1 : 0);
      if (!c.prem) {
        if (c.mrow < 0 || !match_ok) { why(r, "no match answer"); continue; }
        if (mst[c.mrow] != GK_OK) { why(r, "stripped review rejected"); continue; }
        const uint32_t mw = (uint32_t)c.mrow / GK_TILE;
        const uint64_t mbit = 1ull << ((uint32_t)c.mrow % GK_TILE);
        if (mo->too_big[mw] & mbit) { why(r, "stripped review refused as well"); continue; }
        for (uint32_t row = 0; row < nc; row++) {
          er[row] = (mo->err[(size_t)row * mo->n_tiles + mw] & mbit) ? 1 : 0;
          ma[row] = (mo->match[(size_t)row * mo->n_tiles + mw] & mbit) ? 1 : 0;
        }
      }
      ReviewDoc tmp;
      const ReviewDoc* doc = doc_for(e, t, r, &tmp);
      if (!doc) { why(r, "no document"); continue; }
      {
        std::shared_lock<std::shared_mutex> l(e->mu);
        for (uint32_t row = 0; row < nc; row++) {
          if (!ma[row]) continue;
          const ConstraintRec& cr = e->constraints[h->ids[row]];
          auto it = e->templates.find(lower_str(cr.kind));
          if (it == e->templates.end()) throw std::runtime_error("no template");
          v[row] = it->second->render(doc->request, cr.params, e->inventory).empty() ? 0 : 1;
        }
      }
      for (uint32_t row = 0; row < nc; row++) {
        uint64_t& vw = o.viol[(size_t)row * nt + w];
        uint64_t& ew = o.err[(size_t)row * nt + w];
        if ((vw & bit) && row < o.counts.size() && o.counts[row]) o.counts[row]--;   // (never set for a refused review; kept consistent anyway)
        vw &= ~bit; ew &= ~bit;
        if (v[row]) { vw |= bit; if (row < o.counts.size()) o.counts[row]++; add_viol.emplace_back(row, r); if (want_list) { o.list.push_back(row); o.list.push_back(r); o.list_total++; } }
        if (er[row]) { ew |= bit; add_err.emplace_back(row, r); }
        if (want_match && !o.match.empty()) { uint64_t& mw = o.match[(size_t)row * nt + w]; mw = ma[row] ? (mw | bit) : (mw & ~bit); }
      }
      o.too_big[w] &= ~bit;
      h->host_evaluated.push_back(r);
    } catch (const std::exception& ex) {
      why(r, ex.what());
      // (an evaluation error, a document that does not parse: the review stays refused -- fail closed)
    }
  }
}

// Which violating pairs have to be RENDERED to know their result count: need[row][tile], a subset of viol.  The totals plans
// (ensure_totals_plans) answer "this review may yield more than one result" per constraint on the device; a violating pair
// they do not flag has exactly one result.  Everything they cannot answer stays in: constraints without a multi formula,
// reviews beyond the totals plans' limits, GK_TOTALS_RENDER_ALL=1 (test aid: the host pass over every violating pair, as
// before round 3).  Caller holds plan_rw shared and mu shared; ensure_plan ran; ids = constraint id per bitmap row.
// `counted[row]` (round 4): the device COUNTED the results of the row's unflagged pairs -- `sum[row]` of them -- instead of deciding
// "exactly one".
static std::vector<uint64_t> render_needed(gk_engine* e, gk_table* t, const std::vector<uint32_t>& ids, uint32_t nt, const std::vector<uint64_t>& viol,
                                           std::vector<uint8_t>* counted, std::vector<uint64_t>* sum) {
  std::vector<uint64_t> need = viol;
  counted->assign(ids.size(), 0); sum->assign(ids.size(), 0);
  if (getenv("GK_TOTALS_RENDER_ALL") || t->n_reviews == 0) return need;
  if (t->dict_gen != e->dict_reg.gen()) return need;   // (a table flattened before the constraint set changed: no totals plan can read it)
  if (t->pruned && t->reads_gen != e->dict_reg.reads_gen()) return need;
  std::lock_guard<std::mutex> tl(e->totals_mu);
  ensure_totals_plans(e);
  if (e->totals_groups.empty()) return need;
  std::map<uint32_t, uint32_t> row_of;
  for (uint32_t r = 0; r < ids.size(); r++) row_of[ids[r]] = r;
  EvalOptions opt;
  opt.download = true;
  opt.jit_wait = false;   // one pass per audit: the bytecode kernel serves it unless the specialised build is there already
  while (t->tviews.size() < e->totals_groups.size()) t->tviews.push_back(dev_table_view(t->dev));
  std::vector<uint64_t> beyond(nt, 0);
  std::vector<uint8_t> answered(ids.size(), 0);   // 1: a multi row, 2: a flag row (count rows follow)
  std::vector<uint64_t> multi((size_t)ids.size() * nt, 0);
  std::vector<std::vector<uint64_t>> count_rows;   // count rows of all groups, with the main row they belong to
  std::vector<uint32_t> count_main;
  // (the groups are launched in waves: each holds a view of the table -- a stream and result buffers of its own)
  const size_t wave = 8;
  // A PRUNED table holds rows of the published read set only (publish_read_set: the main plans and the counting forms' paths).  A
  // totals plan that has a predicate on any other path -- the lazily prepared "more than one result?" formulas are not published --
  // would read "absent" where the object has a value: its rows stay unanswered for this table and their pairs are rendered
  // (round-4 advisor finding: nothing enforced that compile_multi reads the violation formula's paths only)
  std::vector<uint8_t> usable(e->totals_groups.size(), 1);
  if (t->pruned)
    for (size_t gi = 0; gi < e->totals_groups.size(); gi++)
      for (const HostPlan* hp : {&e->totals_groups[gi]->fast, &e->totals_groups[gi]->big})
        for (const Pattern& pat : hp->pred_patterns) if (!e->dict_reg.reads_has(pat)) { usable[gi] = 0; if (getenv("GK_DEBUG_MULTI")) fprintf(stderr, "[gkgpu totals] plan %zu reads %s: not in the pruned table's read set\n", gi, pattern_to_string(pat).c_str()); }
  // (a constraint's flag row and its count rows may sit in different plans: with ANY of them unanswered its pairs are rendered -- a flag
  //  row that says "counted" with a count row missing would undercount)
  std::vector<uint8_t> blocked(ids.size(), 0);
  for (size_t gi = 0; gi < e->totals_groups.size(); gi++)
    if (!usable[gi]) for (uint32_t id : e->totals_groups[gi]->ids) { auto it = row_of.find(id); if (it != row_of.end()) blocked[it->second] = 1; }
  for (size_t g0 = 0; g0 < e->totals_groups.size(); g0 += wave) {
    const size_t g1 = std::min(e->totals_groups.size(), g0 + wave);
    for (size_t gi = g0; gi < g1; gi++) if (usable[gi]) dev_eval_launch(e->totals_groups[gi]->dev, t->tviews[gi], opt);
    for (size_t gi = g0; gi < g1; gi++) {
      if (!usable[gi]) continue;
      EvalOut og;
      dev_eval_finish(e->totals_groups[gi]->dev, t->tviews[gi], opt, &og);
      const gk_engine::Group& G = *e->totals_groups[gi];
      for (uint32_t w = 0; w < nt && w < og.too_big.size(); w++) beyond[w] |= og.too_big[w];
      for (uint32_t r = 0; r < G.ids.size() && r < og.n_constraints; r++) {
        auto it = row_of.find(G.ids[r]);
        if (it == row_of.end() || og.n_tiles != nt || blocked[it->second]) continue;
        const uint8_t role = r < G.roles.size() ? G.roles[r] : (uint8_t)TR_MULTI;
        const uint64_t* bits = &og.viol[(size_t)r * nt];
        // (autoreject pairs of the totals plan are the main plan's: they carry no violation bit either way)
        if (role == TR_COUNT) { count_rows.emplace_back(bits, bits + nt); count_main.push_back(it->second); continue; }
        answered[it->second] = role == TR_FLAG ? 2 : 1;
        for (uint32_t w = 0; w < nt; w++) multi[(size_t)it->second * nt + w] = bits[w];
      }
    }
  }
  for (uint32_t r = 0; r < ids.size(); r++) {
    if (!answered[r]) continue;
    for (uint32_t w = 0; w < nt; w++) need[(size_t)r * nt + w] = viol[(size_t)r * nt + w] & (multi[(size_t)r * nt + w] | beyond[w]);
    if (answered[r] == 2) (*counted)[r] = 1;
  }
  for (size_t k = 0; k < count_rows.size(); k++) {
    const uint32_t r = count_main[k];
    if (!(*counted)[r]) continue;   // (its flag row is missing: the pairs are rendered)
    uint64_t n = 0;
    for (uint32_t w = 0; w < nt; w++) n += (uint64_t)__builtin_popcountll(count_rows[k][w] & viol[(size_t)r * nt + w] & ~need[(size_t)r * nt + w]);
    (*sum)[r] += n;
  }
  if (getenv("GK_DEBUG_COUNT_CHECK")) {   // debugging aid: every counted pair is rendered as well and compared, review by review
    int shown = 0;
    for (uint32_t r = 0; r < ids.size() && shown < 12; r++) {
      if (!(*counted)[r]) continue;
      const ConstraintRec& c = e->constraints[ids[r]];
      auto itt = e->templates.find(lower_str(c.kind));
      for (uint32_t i = 0; i < t->n_reviews && shown < 12; i++) {
        const uint32_t w = i / 64; const uint64_t b = 1ull << (i % 64);
        if (!(viol[(size_t)r * nt + w] & b) || (need[(size_t)r * nt + w] & b)) continue;
        uint32_t dev = 0;
        for (size_t k = 0; k < count_rows.size(); k++) if (count_main[k] == r && (count_rows[k][w] & b)) dev++;
        ReviewDoc tmp;
        const ReviewDoc* doc = doc_for(e, t, i, &tmp);
        const size_t host = doc ? itt->second->render(doc->request, c.params, e->inventory).size() : 0;
        if (host != dev) { shown++; fprintf(stderr, "[gkgpu count check] %s/%s review %u: device %u, renderer %zu\n", c.kind.c_str(), c.name.c_str(), i, dev, host); }
      }
    }
  }
  return need;
}

// the table's most recent evaluation are rendered on host threads (the reference renders -- and logs -- every message
// as well, manager.go:926-928); only the counts are kept.
int gk_table_totals(gk_engine* e, gk_table* t, gk_totals_out** out) {
  if (!e || !t || !out) return fail(GK_ERR_INVALID, "NULL argument");
  if (t->docs.size() != t->n_reviews && t->texts.size() != t->n_reviews) return fail(GK_ERR_INVALID, "table was created without GK_TABLE_KEEP_DOCS / GK_TABLE_KEEP_TEXT");
  try {
    std::unique_ptr<TotalsHolder> h(new TotalsHolder());
    h->ids = t->last_ids;
    const uint32_t nc = t->last_nc, nt = (t->n_reviews + GK_TILE - 1) / GK_TILE;
    std::vector<uint64_t> viol;
    {
      std::shared_lock<std::shared_mutex> l(e->plan_rw);
      dev_last_viol(t->dev, (uint32_t)e->plan_ids.size(), &viol);
      for (size_t gi = 0; gi < e->extra.size() && gi < t->views.size(); gi++) {
        std::vector<uint64_t> v;
        dev_last_viol(t->views[gi], (uint32_t)e->extra[gi]->ids.size(), &v);
        viol.insert(viol.end(), v.begin(), v.end());
      }
    }
    if (viol.size() != (size_t)nc * nt) return fail(GK_ERR_INVALID, "gk_table_totals: evaluate the table first");
    h->results.assign(nc, 0); h->pairs.assign(nc, 0);
    ensure_plan(e);
    std::shared_lock<std::shared_mutex> pl(e->plan_rw);
    std::shared_lock<std::shared_mutex> rl(e->mu);
    // Round 3: the device decides which violating pairs CAN have more than one result; every other violating pair counts one
    // result without being rendered (configs[2]: 1.8 M violating pairs, ~1 % of them rendered)
    std::vector<uint8_t> counted;
    std::vector<uint64_t> counted_sum;
    // (pairs the host evaluation of the last gk_table_eval added are no part of the device bitmaps: they join here and are rendered)
    std::vector<std::pair<uint32_t, uint32_t>> hv;
    { std::lock_guard<std::mutex> hl(t->host_mu); if (t->host_ids == h->ids) hv = t->host_viol; }   // (another policy set's pairs: dropped)
    for (auto& hp : hv) if (hp.first < nc && hp.second / GK_TILE < nt) viol[(size_t)hp.first * nt + hp.second / GK_TILE] |= 1ull << (hp.second % GK_TILE);
    std::vector<uint64_t> need = render_needed(e, t, h->ids, nt, viol, &counted, &counted_sum);
    for (auto& hp : hv) if (hp.first < nc && hp.second / GK_TILE < nt) need[(size_t)hp.first * nt + hp.second / GK_TILE] |= 1ull << (hp.second % GK_TILE);
    for (uint32_t row = 0; row < nc; row++) {
      for (uint32_t w = 0; w < nt; w++) {
        const uint64_t v = viol[(size_t)row * nt + w];
        h->pairs[row] += (uint64_t)__builtin_popcountll(v);
        if (!counted[row]) h->results[row] += (uint64_t)__builtin_popcountll(v & ~need[(size_t)row * nt + w]);   // exactly one result each
      }
      if (counted[row]) h->results[row] += counted_sum[row];   // (round 4) counted on the device: as many as their count rows say
    }
    uint64_t n_rendered = 0;
    for (uint64_t x : need) n_rendered += (uint64_t)__builtin_popcountll(x);
    h->pub.rendered_pairs = n_rendered;
    struct CRef { const ConstraintRec* c; const Template* tm; };
    std::vector<CRef> cref(nc);
    for (uint32_t row = 0; row < nc; row++) {
      const ConstraintRec& c = e->constraints[h->ids[row]];
      auto it = e->templates.find(lower_str(c.kind));
      if (it == e->templates.end()) return fail(GK_ERR_NOT_FOUND, "unknown constraint template validator: " + c.kind);
      cref[row] = {&c, it->second.get()};
    }
    size_t n_threads = std::max<size_t>(1, std::min<size_t>(host_cpus(), nt / 4 + 1));
    if (const char* ht = getenv("GK_HOST_THREADS")) n_threads = std::max(1, atoi(ht));
    std::vector<std::vector<uint64_t>> part_r(n_threads, std::vector<uint64_t>(nc, 0));
    std::vector<std::string> errs(n_threads);
    std::atomic<uint32_t> next_tile{0};
    auto work = [&](size_t w) {
      try {
        for (;;) {
          const uint32_t tl = next_tile.fetch_add(1);
          if (tl >= nt) break;
          // review-major: a violating review is parsed (GK_TABLE_KEEP_TEXT) at most once, then rendered for each of its constraints
          uint64_t any = 0;
          for (uint32_t row = 0; row < nc; row++) any |= need[(size_t)row * nt + tl];
          for (uint64_t m = any; m; m &= m - 1) {
            const uint32_t b = (uint32_t)__builtin_ctzll(m), r = tl * GK_TILE + b;
            ReviewDoc tmp;
            const ReviewDoc* doc = doc_for(e, t, r, &tmp);
            if (!doc) throw std::runtime_error("review document unavailable");
            for (uint32_t row = 0; row < nc; row++) {
              if (!((need[(size_t)row * nt + tl] >> b) & 1ull)) continue;
              part_r[w][row] += cref[row].tm->render(doc->request, cref[row].c->params, e->inventory).size();
            }
          }
        }
      } catch (const std::exception& ex) { errs[w] = ex.what(); }
    };
    HostWorkers::get().run(n_threads, work);
    for (auto& er : errs) if (!er.empty()) return fail(GK_ERR_REGO, er);
    for (size_t w = 0; w < n_threads; w++) for (uint32_t row = 0; row < nc; row++) h->results[row] += part_r[w][row];
    h->pub.n_constraints = nc; h->pub.constraint_ids = h->ids.data(); h->pub.results = h->results.data(); h->pub.pairs = h->pairs.data();
    *out = &h.release()->pub;
    return GK_OK;
  } catch (const std::exception& ex) { return fail(GK_ERR_DEVICE, ex.what()); }
}

void gk_totals_free(gk_totals_out* o) {
  if (o) delete reinterpret_cast<TotalsHolder*>(o);
}

void gk_topk_free(gk_topk_out* o) {
  if (o) delete reinterpret_cast<TopkHolder*>(o);
}

void gk_eval_free(gk_eval_out* o) {
  if (!o) return;
  delete reinterpret_cast<EvalHolder*>(o);   // pub is the first member
}

int gk_render(gk_engine* e, gk_table* t, uint32_t constraint_id, uint32_t review, char** json_out) {
  if (!e || !t || !json_out) return fail(GK_ERR_INVALID, "NULL argument");
  if (t->docs.empty() && t->texts.empty()) return fail(GK_ERR_INVALID, "table was created without GK_TABLE_KEEP_DOCS / GK_TABLE_KEEP_TEXT");
  if (review >= std::max(t->docs.size(), t->texts.size())) return fail(GK_ERR_INVALID, "review index out of range");
  try {
    std::shared_lock<std::shared_mutex> l(e->mu);
    if (constraint_id >= e->constraints.size()) return fail(GK_ERR_NOT_FOUND, "unknown constraint id");
    const ConstraintRec& c = e->constraints[constraint_id];
    auto it = e->templates.find(lower_str(c.kind));
    if (it == e->templates.end()) return fail(GK_ERR_NOT_FOUND, "unknown constraint template validator: " + c.kind);
    ReviewDoc tmp;
    const ReviewDoc& doc = *doc_for(e, t, review, &tmp);
    ValueVec arr;
    auto vs = it->second->render(doc.request, c.params, e->inventory);
    for (auto& v : vs) {
      static const Value k_msg = Value::string("msg"), k_details = Value::string("details"), no_details = Value::object({});
      ValuePairs o{{k_details, v.details.defined() ? v.details : no_details}, {k_msg, Value::string(v.msg)}};
      arr.push_back(Value::object(o));
    }
    std::string s = to_json(Value::array(arr));
    char* buf = (char*)malloc(s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    *json_out = buf;
    return GK_OK;
  } catch (const std::exception& ex) { return fail(GK_ERR_REGO, ex.what()); }
}

int gk_render_error(gk_engine* e, gk_table* t, uint32_t constraint_id, uint32_t review, char** json_out) {
  if (!e || !t || !json_out) return fail(GK_ERR_INVALID, "NULL argument");
  if (review >= std::max(t->docs.size(), t->texts.size())) return fail(GK_ERR_INVALID, "review index out of range (needs GK_TABLE_KEEP_DOCS / GK_TABLE_KEEP_TEXT)");
  std::shared_lock<std::shared_mutex> l(e->mu);
  if (constraint_id >= e->constraints.size()) return fail(GK_ERR_NOT_FOUND, "unknown constraint id");
  ReviewDoc tmp;
  std::string msg;
  try { msg = autoreject_message(e->constraints[constraint_id].match, *doc_for(e, t, review, &tmp)); } catch (const std::exception& ex) { return fail(GK_ERR_REGO, ex.what()); }
  ValuePairs o{{Value::string("msg"), Value::string(msg)}, {Value::string("autoreject"), Value::boolean(true)}, {Value::string("details"), Value::object({})}};
  std::string s = to_json(Value::array({Value::object(o)}));
  char* buf = (char*)malloc(s.size() + 1);
  memcpy(buf, s.c_str(), s.size() + 1);
  *json_out = buf;
  return GK_OK;
}

void gk_free(void* p) { free(p); }

// ------------------------------------------------------------------------------------------------ micro-batcher (row f1)
namespace {

// results of ONE review of an evaluated table as the JSON gk_query returns; renders from `doc` (parsed lazily: most
// admission reviews violate nothing and never become a Value tree)
// `wanted` (may be NULL): sorted constraint ids the caller asked about (Driver.Query's `constraints`); others are not rendered
std::string query_results_json(gk_engine* e, gk_table* t, const gk_eval_out& ev, uint32_t r, const gk_review_in& in, bool* too_big,
                               const std::vector<uint32_t>* wanted = nullptr) {
  const uint32_t nt = ev.n_tiles, w = r / GK_TILE;
  const uint64_t bit = 1ull << (r % GK_TILE);
  *too_big = (ev.too_big[w] & bit) != 0;
  std::string out = "[";
  bool have_doc = false;
  ReviewDoc doc;
  auto need_doc = [&]() {
    if (have_doc) return;
    Value body = parse_json(in.json, in.json_len);
    Value mns = parse_opt(in.namespace_json, in.namespace_len);
    Value nso = parse_opt(in.ns_object_json, in.ns_object_len);
    if (in.kind == GK_REVIEW_OBJECT) doc = normalize_object(body, mns, nso, in.source, in.operation ? in.operation : "", e->ns_cache);
    else doc = normalize_admission_request(body, mns, nso, in.source, e->ns_cache);
    have_doc = true;
  };
  std::shared_lock<std::shared_mutex> l(e->mu);
  for (uint32_t row = 0; row < ev.n_constraints; row++) {
    const bool is_err = (ev.err[(size_t)row * nt + w] & bit) != 0, is_viol = (ev.viol[(size_t)row * nt + w] & bit) != 0;
    if (!is_err && !is_viol) continue;
    const uint32_t cid = ev.constraint_ids[row];
    if (wanted && !std::binary_search(wanted->begin(), wanted->end(), cid)) continue;
    const ConstraintRec& c = e->constraints[cid];
    need_doc();
    if (is_err) {
      ValuePairs o{{Value::string("constraint"), Value::integer(cid)}, {Value::string("msg"), Value::string(autoreject_message(c.match, doc))},
                   {Value::string("autoreject"), Value::boolean(true)}, {Value::string("details"), Value::object({})}};
      if (out.size() > 1) out += ",";
      out += to_json(Value::object(o));
      continue;
    }
    auto it = e->templates.find(lower_str(c.kind));
    // the device flagged this pair: a template that has gone, or a renderer that finds nothing, is a disagreement between
    // the plan and the evaluator -- an error for the caller (Client.ReviewBatch treats it the same way), never "no results"
    if (it == e->templates.end()) throw std::runtime_error("device / renderer disagree: constraint " + c.kind + "/" + c.name + " has no template");
    const auto vs = it->second->render(doc.request, c.params, e->inventory);
    if (vs.empty()) throw std::runtime_error("device / renderer disagree: the plan reports a violation of " + c.kind + "/" + c.name + " that the evaluator does not render");
    for (auto& v : vs) {
      static const Value k_constraint = Value::string("constraint"), k_msg = Value::string("msg"), k_details = Value::string("details"), no_details = Value::object({});
      ValuePairs o{{k_constraint, Value::integer(cid)}, {k_details, v.details.defined() ? v.details : no_details}, {k_msg, Value::string(v.msg)}};   // (in key order: Value::object has nothing to sort)
      if (out.size() > 1) out += ",";
      out += to_json(Value::object(o));
    }
  }
  return out + "]";
}

}  // namespace
struct gk_engine::BatchResult {
  gk_table* table = nullptr;
  gk_eval_out* ev = nullptr;
  ~BatchResult() { if (ev) gk_eval_free(ev); if (table) gk_table_free(table); }
};
namespace {

void batcher_loop(gk_engine* e) {
  gk_engine::Batcher& B = e->batcher;
  for (;;) {
    std::vector<gk_engine::Request*> batch;
    {
      std::unique_lock<std::mutex> l(B.mu);
      B.cv.wait(l, [&] { return B.stop || !B.queue.empty(); });
      if (B.stop && B.queue.empty()) return;
      // the first request of a batch waits up to window_us for company (or until max_batch requests are queued)
      const auto deadline = B.queue.front()->arrived + std::chrono::microseconds(B.opts.window_us);
      B.cv.wait_until(l, deadline, [&] { return B.stop || B.queue.size() >= B.opts.max_batch; });
      while (!B.queue.empty() && batch.size() < B.opts.max_batch) { batch.push_back(B.queue.front()); B.queue.pop_front(); }
    }
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<gk_review_in> ins;
    std::vector<uint8_t> prem;
    bool any_prem = false;
    for (auto* r : batch) { ins.push_back(*r->in); prem.push_back(r->pre_matched ? 1 : 0); any_prem = any_prem || r->pre_matched; }
    std::vector<int32_t> st(batch.size(), GK_OK);
    auto br = std::make_shared<gk_engine::BatchResult>();
    // an admission batch serves the policy set loaded now and is evaluated once: a PRUNED table (rows of the key paths that set reads,
    // the ingest walks past the rest).  A constraint that arrives between the two calls makes it stale: built again, once
    int rc = GK_OK;
    std::string err;
    { std::lock_guard<std::mutex> turn(e->policy_gate); }   // (a policy change that is waiting goes first)
    std::shared_lock<std::shared_mutex> policy_epoch(e->policy_rw);   // the policy set does not change between the flatten and the launch
    for (int attempt = 0; attempt < 3; attempt++) {   // (what can still make the table stale: a referential constraint recompiled against a changed inventory)
      if (br->table) { gk_table_free(br->table); br->table = nullptr; }
      rc = table_create_impl(e, ins.data(), ins.size(), attempt < 2 ? GK_TABLE_PRUNED : 0u, st.data(), any_prem ? prem.data() : nullptr, &br->table);
      err = rc == GK_OK ? "" : gk_last_error();
      if (rc != GK_OK) break;
      rc = gk_table_eval(e, br->table, 0, &br->ev);
      if (rc == GK_OK) break;
      err = gk_last_error();
      if (rc != GK_ERR_INVALID) break;
    }
    policy_epoch.unlock();
    const double dev_us = br->ev ? br->ev->kernel_ms * 1e3 : 0;
    for (size_t i = 0; i < batch.size(); i++) {
      gk_engine::Request* r = batch[i];
      r->batch_size = (uint32_t)batch.size();
      r->queue_us = std::chrono::duration<double, std::micro>(t_start - r->arrived).count();
      r->device_us = dev_us;
      r->index = (uint32_t)i;
      if (rc != GK_OK) { r->status = rc; r->error = err; }
      else if (st[i] != GK_OK) { r->status = GK_ERR_REVIEW; r->error = br->table->review_errors[i]; }
      else r->batch = br;   // the caller renders its own results (gk_query): rendering runs in parallel across callers
    }
    {
      std::lock_guard<std::mutex> l(B.mu);
      B.batches++; B.reviews += batch.size();
      for (auto* r : batch) { r->done = true; r->cv.notify_one(); }
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ multi-GPU (row e)
int gk_comm_unique_id(char id[GK_COMM_ID_BYTES]) {
  std::string err;
  if (!id) return fail(GK_ERR_INVALID, "NULL argument");
  if (!dev_comm_unique_id(id, &err)) return fail(GK_ERR_DEVICE, err);
  return GK_OK;
}

int gk_comm_init(gk_engine* e, const char id[GK_COMM_ID_BYTES], int rank, int world) {
  if (!e || !id || rank < 0 || world < 1 || rank >= world) return fail(GK_ERR_INVALID, "bad argument");
  std::string err;
  DevComm* c = dev_comm_init(e->opts.device, id, rank, world, &err);
  if (!c) return fail(GK_ERR_DEVICE, err);
  if (e->comm) dev_comm_free(e->comm);
  e->comm = c;
  return GK_OK;
}

// used by the test-only CPU emulation (tests/native/hostemu.cpp): adopt a communicator built elsewhere
int gk_comm_init_host_impl(gk_engine* e, DevComm* c) { if (!e || !c) return GK_ERR_INVALID; if (e->comm) dev_comm_free(e->comm); e->comm = c; return GK_OK; }

void gk_comm_destroy(gk_engine* e) { if (e && e->comm) { dev_comm_free(e->comm); e->comm = nullptr; } }

int gk_comm_info(gk_engine* e, int32_t* rank, int32_t* world) {
  if (!e || !rank || !world) return fail(GK_ERR_INVALID, "NULL argument");
  if (!e->comm) return fail(GK_ERR_INVALID, "gk_comm_init first");
  std::string err;
  int r = 0, w = 0;
  if (!dev_comm_query(e->comm, &r, &w, &err)) return fail(GK_ERR_DEVICE, err);
  *rank = r; *world = w;
  return GK_OK;
}

struct ShardHolder {
  gk_shard_out pub;   // first member
  std::vector<uint32_t> ids, shard_reviews;
  std::vector<int64_t> totals, err_totals;
  std::vector<uint64_t> gathered;
  uint64_t merged_slot = 0;
};

int gk_table_sweep_sharded(gk_engine* e, gk_table* t, uint32_t flags, gk_shard_out** out) {
  const bool enqueue = (flags & GK_SHARD_ENQUEUE) != 0;
  if (!e || !t || (!out && !enqueue)) return fail(GK_ERR_INVALID, "NULL argument");
  if (!e->comm) return fail(GK_ERR_INVALID, "gk_comm_init first");
  try {
    ensure_plan(e);
    if (t->dict_gen != e->dict_reg.gen()) return fail(GK_ERR_INVALID, "the table was flattened before a constraint with dictionary predicates was added: create it again");
    if (t->pruned && t->reads_gen != e->dict_reg.reads_gen())
      return fail(GK_ERR_INVALID, "the pruned table was flattened before a constraint that reads other key paths was added (GK_TABLE_PRUNED): create it again");
    std::unique_ptr<ShardHolder> h(new ShardHolder());
    std::shared_lock<std::shared_mutex> l(e->plan_rw);
    const HostPlan* hp = nullptr;
    DevPlan* dp = nullptr;
    { std::lock_guard<std::mutex> vl(e->variants_mu); dp = plan_for_table(e, t, &hp); }
    const uint32_t nc0 = (uint32_t)e->plan_ids.size();
    const size_t n_groups = 1 + e->extra.size();
    while (t->views.size() < e->extra.size()) t->views.push_back(dev_table_view(t->dev));
    if (t->shard_gen != e->plan_gen || t->shard.shard_reviews.empty() || t->group_shards.size() != e->extra.size()) {   // (collective) once per table and plan
      dev_shard_setup(t->dev, e->comm, nc0, &t->shard);
      t->group_shards.assign(e->extra.size(), ShardInfo());
      for (size_t gi = 0; gi < e->extra.size(); gi++) dev_shard_setup(t->views[gi], e->comm, (uint32_t)e->extra[gi]->ids.size(), &t->group_shards[gi]);
      t->shard_gen = e->plan_gen;
    }
    EvalOptions opt;
    opt.download = false;
    opt.shard = true;
    opt.jit_wait = true;   // a shard of the audit set is a resident table: the plan-specialised kernel or nothing
    if (t->n_rejected == ~0ull) {   // (a table's reviews never change: counted once, not per sweep -- a million strings)
      uint64_t nr = 0;
      for (const std::string& er : t->review_errors) if (!er.empty()) nr++;
      t->n_rejected = nr;
    }
    const uint64_t rejected = t->n_rejected;
    if (enqueue) {   // sweep + exchange of every plan group onto the stream(s); whoever collects waits for them
      // (four enqueues per plan group; GK_SHARD_GRAPH=1: a single plan group's step replays as one captured graph)
      dev_shard_enqueue(dp, t->dev, e->comm, opt, nc0, rejected, e->extra.empty());
      for (size_t gi = 0; gi < e->extra.size(); gi++)
        dev_shard_enqueue(e->extra[gi]->dev, t->views[gi], e->comm, opt, (uint32_t)e->extra[gi]->ids.size(), rejected, false);
      if (out) *out = nullptr;
      return GK_OK;
    }
    EvalOut eo;
    // local evaluation, finished (incl. the large-capacity pass for overflowing reviews) BEFORE the exchange, so that the
    // gathered bitmaps are complete; then the exchange step on the same stream.  A constraint set that needs several
    // plan groups runs one evaluation + exchange per group (each group has its own slot buffer on its view of the table).
    const void* d_all = nullptr;
    const bool want_host = (flags & GK_SHARD_DOWNLOAD) != 0;
    std::vector<uint64_t> g0;
    // GK_SHARD_COLLECT: the answer of the last enqueue-only pass, without sweeping again -- unless some review of that pass,
    // on any rank, was left to the large-capacity re-run (then this is an ordinary collecting sweep; every rank decides alike)
    const bool collected = (flags & GK_SHARD_COLLECT) && e->extra.empty() &&
                           dev_shard_collect(t->dev, e->comm, nc0, &h->totals, want_host ? &g0 : nullptr, &d_all);
    if (!collected) dev_eval(dp, t->dev, opt, &eo);
    // what the bitmaps cannot say travels with the totals (fail closed, like Client.AuditAggregate): autoreject pairs, reviews
    // beyond the engine's limits, reviews HandleReview rejected when this shard was built
    int64_t beyond = 0, not_eval = 0;
    auto split = [&](std::vector<int64_t>& raw, uint32_t ncg, bool first) {   // raw: [ncg] pairs | [ncg] autoreject | beyond | not evaluated | left to the re-run
      h->err_totals.insert(h->err_totals.end(), raw.begin() + ncg, raw.begin() + 2 * (size_t)ncg);
      beyond += raw[2 * (size_t)ncg];
      if (first) not_eval = raw[2 * (size_t)ncg + 1];
      raw.resize(ncg);
    };
    if (!collected) dev_shard_exchange(t->dev, e->comm, nc0, rejected, &h->totals, want_host ? &g0 : nullptr, &d_all);
    split(h->totals, nc0, true);
    uint32_t nc = nc0;
    if (n_groups == 1) h->gathered.swap(g0);
    else {
      // merged view for the caller: per rank [all groups' bitmap rows | all groups' counts], same stride; host copy only
      const int world = dev_comm_world(e->comm);
      const uint32_t stride = t->shard.stride_tiles;
      std::vector<std::vector<uint64_t>> parts(n_groups);
      std::vector<uint32_t> ncs{nc0};
      std::vector<uint64_t> slot_bytes{t->shard.slot_bytes};
      parts[0].swap(g0);
      for (size_t gi = 0; gi < e->extra.size(); gi++) {
        EvalOut og;
        dev_eval(e->extra[gi]->dev, t->views[gi], opt, &og);
        eo.kernel_ms += og.kernel_ms; eo.fast_kernel_ms += og.fast_kernel_ms; eo.n_overflow += og.n_overflow;
        std::vector<int64_t> tg;
        const void* d_g = nullptr;
        const uint32_t ncg = (uint32_t)e->extra[gi]->ids.size();
        dev_shard_exchange(t->views[gi], e->comm, ncg, rejected, &tg, want_host ? &parts[gi + 1] : nullptr, &d_g);
        split(tg, ncg, false);
        h->totals.insert(h->totals.end(), tg.begin(), tg.end());
        ncs.push_back(ncg); slot_bytes.push_back(t->group_shards[gi].slot_bytes);
        nc += ncg;
      }
      d_all = nullptr;   // several device buffers: only the merged host copy is handed out
      if (want_host) {
        const uint64_t merged_slot = (((uint64_t)nc * stride * 8 + (uint64_t)nc * 4) + 15) & ~(uint64_t)15;
        h->gathered.assign((size_t)world * merged_slot / 8, 0);
        for (int r = 0; r < world; r++) {
          uint8_t* dst = reinterpret_cast<uint8_t*>(h->gathered.data()) + (size_t)r * merged_slot;
          size_t row0 = 0;
          for (size_t g = 0; g < n_groups; g++) {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(parts[g].data()) + (size_t)r * slot_bytes[g];
            memcpy(dst + row0 * stride * 8, src, (size_t)ncs[g] * stride * 8);
            memcpy(dst + (size_t)nc * stride * 8 + row0 * 4, src + (size_t)ncs[g] * stride * 8, (size_t)ncs[g] * 4);
            row0 += ncs[g];
          }
        }
        h->merged_slot = merged_slot;
      }
    }
    h->ids = e->plan_ids;
    for (auto& g : e->extra) h->ids.insert(h->ids.end(), g->ids.begin(), g->ids.end());
    h->shard_reviews = t->shard.shard_reviews;
    gk_shard_out& p = h->pub;
    memset(&p, 0, sizeof p);
    p.world = (uint32_t)dev_comm_world(e->comm); p.rank = (uint32_t)dev_comm_rank(e->comm);
    p.n_constraints = nc; p.stride_tiles = t->shard.stride_tiles; p.slot_bytes = n_groups == 1 ? t->shard.slot_bytes : h->merged_slot;
    p.constraint_ids = h->ids.data(); p.shard_reviews = h->shard_reviews.data(); p.totals = h->totals.data();
    p.gathered = h->gathered.empty() ? nullptr : h->gathered.data();
    p.d_gathered = d_all;
    p.kernel_ms = eo.kernel_ms; p.fast_kernel_ms = eo.fast_kernel_ms; p.n_overflow = eo.n_overflow;
    p.err_totals = h->err_totals.data(); p.beyond_limits = beyond; p.not_evaluated = not_eval;
    {   // the exchange step's own figures (the primary plan group's; further groups exchange the same number of bytes per row)
      const ShardExchangeStats xs = dev_shard_exchange_stats(t->dev, e->comm);
      p.exchange_ms = collected ? 0.f : xs.exchange_ms;   // (a collected enqueue-only pass was not timed)
      p.exchange_bytes_inbound = xs.inbound_bytes;
      for (size_t gi = 0; gi < e->extra.size() && gi < t->views.size(); gi++) {
        const ShardExchangeStats xg = dev_shard_exchange_stats(t->views[gi], e->comm);
        if (!collected) p.exchange_ms += xg.exchange_ms;
        p.exchange_bytes_inbound += xg.inbound_bytes;
      }
      p.exchange_overlapped = xs.overlap ? 1u : 0u;
    }
    *out = &h.release()->pub;
    return GK_OK;
  } catch (const Unsupported& ex) { return fail(GK_ERR_UNSUPPORTED, ex.what());
  } catch (const std::exception& ex) { return fail(GK_ERR_DEVICE, ex.what()); }
}

void gk_shard_free(gk_shard_out* o) { if (o) delete reinterpret_cast<ShardHolder*>(o); }

// ------------------------------------------------------------------------------------------------ resident set (row f2)
namespace {

struct SweepHolder {
  gk_sweep_out pub;   // first member
  std::vector<uint32_t> ids;
  std::vector<uint64_t> results, pairs;
};

const gk_engine::ResObj* resident_namespace_of(const gk_engine::Resident& R, const gk_engine::ResObj& o) {
  if (o.ns.empty()) return nullptr;
  auto it = R.by_key.find("cluster/v1/Namespace/" + o.ns + "/");
  if (it == R.by_key.end() || !R.objs[it->second].alive) return nullptr;
  return &R.objs[it->second];
}

// the review pkg/audit builds for a cached object (auditFromCache, pkg/audit/manager.go:591-642): AugmentedUnstructured
// {Object, Namespace: the cached Namespace it lives in}, no Source, and the same Namespace as the namespaceObject option
gk_review_in resident_review_in(const gk_engine::Resident& R, const gk_engine::ResObj& o) {
  gk_review_in in;
  memset(&in, 0, sizeof in);
  in.kind = GK_REVIEW_OBJECT;
  in.source = GK_SRC_EMPTY;
  in.json = o.json.data(); in.json_len = o.json.size();
  if (const gk_engine::ResObj* ns = resident_namespace_of(R, o)) {
    in.namespace_json = ns->json.data(); in.namespace_len = ns->json.size();
    in.ns_object_json = ns->json.data(); in.ns_object_len = ns->json.size();
  }
  return in;
}

void resident_drop_chunk(gk_engine::ResChunk& c) {
  if (c.ev) gk_eval_free(c.ev);
  if (c.table) gk_table_free(c.table);
  c.ev = nullptr; c.table = nullptr;
}

}  // namespace

int gk_resident_sweep(gk_engine* e, uint32_t flags, gk_sweep_out** out) {
  if (!e || !out) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    gk_engine::Resident& R = e->resident;
    std::lock_guard<std::mutex> rl(R.mu);
    const auto t0 = std::chrono::steady_clock::now();
    ensure_plan(e);
    uint64_t flattened = 0;
    // compaction: when masked-out slots outnumber the live ones (or the chunks have piled up) everything is flattened
    // again into one chunk
    bool stale_rows = false;   // a constraint with dictionary predicates arrived: every chunk lacks its <leaf>.$d rows
    for (auto& c : R.chunks) stale_rows = stale_rows || c.table->dict_gen != e->dict_reg.gen() || (c.table->pruned && c.table->reads_gen != e->dict_reg.reads_gen());
    if (stale_rows || R.n_dead_slots > R.n_live + 1024 || R.chunks.size() > 16) {
      for (auto& c : R.chunks) resident_drop_chunk(c);
      R.chunks.clear();
      R.pending.clear();
      R.n_dead_slots = 0;
      for (uint32_t id = 0; id < R.objs.size(); id++) if (R.objs[id].alive) { R.objs[id].chunk = UINT32_MAX; R.pending.push_back(id); }
    }
    if (!R.pending.empty()) {
      std::vector<gk_review_in> ins;
      gk_engine::ResChunk c;
      for (uint32_t id : R.pending) {
        if (!R.objs[id].alive) continue;
        ins.push_back(resident_review_in(R, R.objs[id]));
        c.obj_of_slot.push_back(id);
      }
      R.pending.clear();
      if (!ins.empty()) {
        std::vector<int32_t> st(ins.size(), GK_OK);
        int rc = gk_table_create(e, ins.data(), ins.size(), GK_TABLE_RESIDENT | GK_TABLE_PRUNED, st.data(), &c.table);   // (stale chunks are flattened again: stale_rows above)
        if (rc != GK_OK) return rc;
        c.live.assign((ins.size() + 63) / 64, 0);
        for (size_t k = 0; k < ins.size(); k++) {
          gk_engine::ResObj& o = R.objs[c.obj_of_slot[k]];
          o.chunk = (uint32_t)R.chunks.size(); o.slot = (uint32_t)k;
          if (st[k] == GK_OK) { c.live[k / 64] |= 1ull << (k % 64); c.n_live++; }
        }
        flattened = ins.size();
        R.flattened_total += flattened;
        R.chunks.push_back(std::move(c));
      }
    }
    const auto t1 = std::chrono::steady_clock::now();
    ensure_plan(e);   // flattening may have interned new key paths: the plan is re-bound to the dictionary first
    uint64_t gen;
    { std::shared_lock<std::shared_mutex> l(e->plan_rw); gen = e->plan_gen; }
    std::unique_ptr<SweepHolder> h(new SweepHolder());
    for (auto& c : R.chunks) {
      if (c.ev && c.plan_gen == gen) continue;   // bitmaps still valid: same rows, same policies
      if (c.ev) { gk_eval_free(c.ev); c.ev = nullptr; }
      int rc = gk_table_eval(e, c.table, 0, &c.ev);
      if (rc != GK_OK) return rc;
      c.plan_gen = gen;
    }
    {   // audit-process excluder over the resident objects (manager.go:599): a mask beside `live`, refreshed when the Config changes
      std::shared_lock<std::shared_mutex> l(e->mu);
      auto ex = e->excluder.find("audit");
      for (auto& c : R.chunks) {
        if (c.excl_gen == e->excluder_gen && c.shown.size() == c.live.size()) {
          for (size_t w = 0; w < c.live.size(); w++) c.shown[w] &= c.live[w];
          continue;
        }
        c.shown = c.live;
        if (ex != e->excluder.end() && !ex->second.empty())
          for (size_t k = 0; k < c.obj_of_slot.size(); k++) {
            const gk_engine::ResObj& o = R.objs[c.obj_of_slot[k]];
            if (excluder_matches(ex->second, o.is_namespace ? o.path.back() : o.ns)) c.shown[k / 64] &= ~(1ull << (k % 64));
          }
        c.excl_gen = e->excluder_gen;
      }
    }
    // per-constraint totals over the live, not excluded slots of all chunks
    if (!R.chunks.empty()) {
      const gk_eval_out* ev0 = R.chunks[0].ev;
      h->ids.assign(ev0->constraint_ids, ev0->constraint_ids + ev0->n_constraints);
    } else {
      { std::lock_guard<std::mutex> gate(e->plan_gate); }   // (a plan change that is waiting goes first)
      std::shared_lock<std::shared_mutex> l(e->plan_rw);
      h->ids = e->plan_ids;
      for (auto& g : e->extra) h->ids.insert(h->ids.end(), g->ids.begin(), g->ids.end());
    }
    const uint32_t nc = (uint32_t)h->ids.size();
    h->pairs.assign(nc, 0); h->results.assign(nc, 0);
    uint64_t beyond = 0;
    for (auto& c : R.chunks) {
      const gk_eval_out* ev = c.ev;
      for (uint32_t row = 0; row < ev->n_constraints && row < nc; row++)
        for (uint32_t w = 0; w < ev->n_tiles; w++) h->pairs[row] += (uint64_t)__builtin_popcountll(ev->viol[(size_t)row * ev->n_tiles + w] & c.shown[w]);
      for (uint32_t w = 0; w < ev->n_tiles; w++) beyond += (uint64_t)__builtin_popcountll(ev->too_big[w] & c.shown[w]);
    }
    if (flags & GK_SWEEP_RESULT_TOTALS) {   // results, not pairs (pkg/audit/manager.go:902)
      // as gk_table_totals: the device says which violating live pairs can have more than one result; the others count one
      std::shared_lock<std::shared_mutex> pl(e->plan_rw);
      std::shared_lock<std::shared_mutex> l(e->mu);
      for (auto& c : R.chunks) {
        const gk_eval_out* ev = c.ev;
        const uint32_t nt = ev->n_tiles, ncc = std::min<uint32_t>(ev->n_constraints, nc);
        std::vector<uint64_t> shown_viol((size_t)nc * nt, 0);
        for (uint32_t row = 0; row < ncc; row++)
          for (uint32_t w = 0; w < nt; w++) shown_viol[(size_t)row * nt + w] = ev->viol[(size_t)row * nt + w] & c.shown[w];
        std::vector<uint8_t> counted;
        std::vector<uint64_t> counted_sum;
        const std::vector<uint64_t> need = render_needed(e, c.table, h->ids, nt, shown_viol, &counted, &counted_sum);
        for (uint32_t row = 0; row < ncc; row++) {
          if (counted[row]) { h->results[row] += counted_sum[row]; continue; }   // (round 4: counted on the device)
          for (uint32_t w = 0; w < nt; w++) h->results[row] += (uint64_t)__builtin_popcountll(shown_viol[(size_t)row * nt + w] & ~need[(size_t)row * nt + w]);
        }
        for (uint32_t slot = 0; slot < c.obj_of_slot.size(); slot++) {
          const uint64_t bit = 1ull << (slot % 64);
          bool any = false;
          for (uint32_t row = 0; row < ncc && !any; row++) any = (need[(size_t)row * nt + slot / 64] & bit) != 0;
          if (!any) continue;
          const gk_review_in in = resident_review_in(R, R.objs[c.obj_of_slot[slot]]);
          ReviewDoc doc = normalize_object(parse_json(in.json, in.json_len), parse_opt(in.namespace_json, in.namespace_len),
                                           parse_opt(in.ns_object_json, in.ns_object_len), in.source, "", e->ns_cache);
          for (uint32_t row = 0; row < ncc; row++) {
            if (!(need[(size_t)row * nt + slot / 64] & bit)) continue;
            const ConstraintRec& k = e->constraints[h->ids[row]];
            auto it = e->templates.find(lower_str(k.kind));
            if (it != e->templates.end()) h->results[row] += it->second->render(doc.request, k.params, e->inventory).size();
          }
        }
      }
    }
    R.swept = true;
    gk_sweep_out& p = h->pub;
    memset(&p, 0, sizeof p);
    p.n_objects = R.n_live; p.n_constraints = nc; p.constraint_ids = h->ids.data();
    p.pairs = h->pairs.data(); p.results = (flags & GK_SWEEP_RESULT_TOTALS) ? h->results.data() : nullptr;
    p.n_chunks = (uint32_t)R.chunks.size(); p.flattened = flattened; p.beyond_limits = beyond;
    p.sync_s = std::chrono::duration<double>(t1 - t0).count();
    p.eval_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    *out = &h.release()->pub;
    return GK_OK;
  } catch (const std::exception& ex) { return fail(GK_ERR_DEVICE, ex.what()); }
}

void gk_sweep_free(gk_sweep_out* o) { if (o) delete reinterpret_cast<SweepHolder*>(o); }

namespace {
// the cached answer for one resident object, as gk_query's JSON; false: not resident / not swept / its slot is stale
// audit: the answer pkg/audit would get -- an object the Config excludes from the audit process is never reviewed ("[]")
bool resident_answer(gk_engine* e, uint32_t id, std::string* json, int* status, std::string* err, bool audit = false, const std::vector<uint32_t>* wanted = nullptr) {
  gk_engine::Resident& R = e->resident;
  const gk_engine::ResObj& o = R.objs[id];
  if (!R.swept || !o.alive || o.chunk == UINT32_MAX) return false;
  const gk_engine::ResChunk& c = R.chunks[o.chunk];
  if (!c.ev || !(c.live[o.slot / 64] & (1ull << (o.slot % 64)))) return false;
  if (audit) {
    { std::shared_lock<std::shared_mutex> l(e->mu); if (c.excl_gen != e->excluder_gen) return false; }   // Config changed since the sweep
    if (!(c.shown[o.slot / 64] & (1ull << (o.slot % 64)))) { *json = "[]"; *status = GK_OK; return true; }
  }
  { std::shared_lock<std::shared_mutex> l(e->plan_rw); if (e->plan_dirty || c.plan_gen != e->plan_gen) return false; }
  const gk_review_in in = resident_review_in(R, o);
  bool too_big = false;
  *json = query_results_json(e, c.table, *c.ev, o.slot, in, &too_big, wanted);
  *status = GK_OK;
  if (too_big) { *status = GK_ERR_LIMIT; *err = "review is beyond the engine's limits (more than 255 elements in an array that constraint predicates iterate, or an object where they iterate array elements)"; }
  return true;
}
}  // namespace

// Driver.Query's `constraints` as a sorted id list; GK_ERR_NOT_FOUND for an id that is not loaded
static int wanted_ids(gk_engine* e, const uint32_t* ids, size_t n, std::vector<uint32_t>* out) {
  out->assign(ids, ids + n);
  std::sort(out->begin(), out->end());
  out->erase(std::unique(out->begin(), out->end()), out->end());
  std::shared_lock<std::shared_mutex> l(e->mu);
  for (uint32_t id : *out)
    if (id >= e->constraints.size() || !e->constraints[id].alive) return fail(GK_ERR_NOT_FOUND, "unknown constraint template validator: constraint id " + std::to_string(id) + " is not loaded");
  return GK_OK;
}
static int put_string(const std::string& js, char** out) {
  char* buf = (char*)malloc(js.size() + 1);
  if (!buf) return fail(GK_ERR_INTERNAL, "out of memory");
  memcpy(buf, js.c_str(), js.size() + 1);
  *out = buf;
  return GK_OK;
}

int gk_resident_review(gk_engine* e, const char* const* path, size_t npath, char** results_json) { return gk_resident_review_ex(e, path, npath, nullptr, 0, 0, results_json); }

int gk_resident_review_ex(gk_engine* e, const char* const* path, size_t npath, const uint32_t* constraint_ids, size_t n_constraints, uint32_t flags, char** results_json) {
  if (!e || !path || !npath || !results_json) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    std::vector<uint32_t> wanted;
    if (constraint_ids) { int rc = wanted_ids(e, constraint_ids, n_constraints, &wanted); if (rc != GK_OK) return rc; }
    gk_engine::Resident& R = e->resident;
    std::string key;
    for (size_t i = 0; i < npath; i++) { key += path[i]; key.push_back('/'); }
    if (flags & GK_QUERY_PRE_MATCHED) {
      // the sweep kept match AND violation: the caller's own match is answered by evaluating the resident text again, pre-matched
      // (copies: the resident set may change while the batcher works)
      std::string obj, ns;
      bool has_ns = false;
      {
        std::lock_guard<std::mutex> rl(R.mu);
        auto it = R.by_key.find(key);
        if (it == R.by_key.end() || !R.objs[it->second].alive) return fail(GK_ERR_NOT_FOUND, "object is not in the resident set");
        const gk_engine::ResObj& o = R.objs[it->second];
        obj = o.json;
        if (const gk_engine::ResObj* n = resident_namespace_of(R, o)) { has_ns = true; ns = n->json; }
      }
      gk_review_in in;
      memset(&in, 0, sizeof in);
      in.kind = GK_REVIEW_OBJECT; in.source = GK_SRC_EMPTY;
      in.json = obj.data(); in.json_len = obj.size();
      if (has_ns) { in.namespace_json = ns.data(); in.namespace_len = ns.size(); in.ns_object_json = ns.data(); in.ns_object_len = ns.size(); }
      return gk_query_ex2(e, &in, constraint_ids, n_constraints, GK_QUERY_PRE_MATCHED, results_json, nullptr, nullptr);
    }
    std::lock_guard<std::mutex> rl(R.mu);
    auto it = R.by_key.find(key);
    std::string js, err;
    int st = GK_OK;
    if (it == R.by_key.end() || !resident_answer(e, it->second, &js, &st, &err, true, constraint_ids ? &wanted : nullptr))
      return fail(GK_ERR_NOT_FOUND, "object is not in the swept resident set (unknown, changed since the last gk_resident_sweep, or policies / the process excluder changed)");
    if (st != GK_OK) return fail(st, err);
    char* buf = (char*)malloc(js.size() + 1);
    memcpy(buf, js.c_str(), js.size() + 1);
    *results_json = buf;
    return GK_OK;
  } catch (const std::exception& ex) { return fail(GK_ERR_REGO, ex.what()); }
}

int gk_batcher_start(gk_engine* e, const gk_batch_opts* opts) {
  if (!e) return fail(GK_ERR_INVALID, "NULL argument");
  gk_engine::Batcher& B = e->batcher;
  {
    std::unique_lock<std::mutex> l(B.mu);
    if (opts) { B.opts = *opts; if (!B.opts.max_batch) B.opts.max_batch = 64; }
    const uint32_t nw = std::max<uint32_t>(1, std::min<uint32_t>(16, B.opts.workers ? B.opts.workers : 2));
    if (B.running && B.workers.size() == nw) return GK_OK;
    if (B.running) { l.unlock(); gk_batcher_stop(e); l.lock(); }   // a different number of workers: restart them
    B.stop = false;
    B.running = true;
    for (uint32_t w = 0; w < nw; w++) B.workers.emplace_back(batcher_loop, e);
  }
  return GK_OK;
}

void gk_batcher_stop(gk_engine* e) {
  if (!e) return;
  gk_engine::Batcher& B = e->batcher;
  {
    std::lock_guard<std::mutex> l(B.mu);
    if (!B.running) return;
    B.stop = true;
  }
  B.cv.notify_all();
  for (auto& w : B.workers) w.join();
  std::lock_guard<std::mutex> l(B.mu);
  B.workers.clear();
  B.running = false;
  B.stop = false;
  for (auto* r : B.queue) { r->status = GK_ERR_INVALID; r->error = "the admission batcher was stopped"; r->done = true; r->cv.notify_one(); }
  B.queue.clear();
}

int gk_query(gk_engine* e, const gk_review_in* review, char** results_json, gk_query_stats* stats) { return gk_query_ex(e, review, 0, results_json, nullptr, stats); }

// the trace of one answered review (QueryResponse.Trace): where and how it was evaluated, what every constraint yielded
static std::string query_trace(gk_engine* e, const char* where, const std::string& results, const gk_query_stats* st, const std::vector<uint32_t>* wanted = nullptr) {
  std::ostringstream os;
  os << "gkgpu trace: evaluated " << where;
  if (st) os << "; batch of " << st->batch_size << " review(s), queued " << (long long)(st->queue_us * 1e3) << " ns, device " << (long long)(st->device_us * 1e3) << " ns";
  os << "\n";
  std::map<uint32_t, std::vector<std::string>> by;
  try {
    Value rows = parse_json(results.data(), results.size());
    if (rows.is_array()) for (const Value& r : rows.items()) {
      const Value* c = r.get("constraint"); const Value* m = r.get("msg"); const Value* ar = r.get("autoreject");
      if (c && c->is_number() && m && m->is_string()) by[(uint32_t)c->i].push_back(std::string(ar ? "autoreject: " : "violation: ") + m->str());
    }
  } catch (const std::exception&) {}
  std::shared_lock<std::shared_mutex> l(e->mu);
  for (const ConstraintRec& c : e->constraints) {
    if (!c.alive || (wanted && !std::binary_search(wanted->begin(), wanted->end(), c.id))) continue;
    os << "  constraint " << c.kind << "/" << c.name << " (id " << c.id << "): ";
    auto it = by.find(c.id);
    if (it == by.end()) { os << (wanted ? "no result\n" : "no result (does not match, or matches and is satisfied)\n"); continue; }
    os << it->second.size() << " result(s)\n";
    for (auto& m : it->second) os << "    " << m << "\n";
  }
  return os.str();
}

int gk_query_ex(gk_engine* e, const gk_review_in* review, uint32_t qflags, char** results_json, char** trace_out, gk_query_stats* stats) {
  return gk_query_ex2(e, review, nullptr, 0, qflags, results_json, trace_out, stats);
}

int gk_query_ex2(gk_engine* e, const gk_review_in* review, const uint32_t* constraint_ids, size_t n_constraints, uint32_t qflags,
                 char** results_json, char** trace_out, gk_query_stats* stats) {
  if (!e || !review || !results_json) return fail(GK_ERR_INVALID, "NULL argument");
  if (trace_out) *trace_out = nullptr;
  const bool pre_matched = (qflags & GK_QUERY_PRE_MATCHED) != 0;
  std::vector<uint32_t> wanted_v;
  const std::vector<uint32_t>* wanted = nullptr;
  if (constraint_ids) {
    int rc = wanted_ids(e, constraint_ids, n_constraints, &wanted_v);
    if (rc != GK_OK) return rc;
    wanted = &wanted_v;
    if (wanted_v.empty()) {   // Driver.Query with no constraints: nothing to evaluate
      if (stats) memset(stats, 0, sizeof *stats);
      return put_string("[]", results_json);
    }
  }
  const bool want_trace = trace_out && ((qflags & GK_QUERY_TRACE) || (e->opts.flags & GK_OPT_TRACE));
  auto put_trace = [&](const char* where, const std::string& js, const gk_query_stats* st) {
    if (!want_trace) return;
    const std::string t = query_trace(e, where, js, st, wanted);
    char* b = (char*)malloc(t.size() + 1);
    memcpy(b, t.c_str(), t.size() + 1);
    *trace_out = b;
  };
  // a review that is byte for byte a swept resident object (same object text, the Namespace the sweep used, no Source --
  // what pkg/audit's auditFromCache sends, manager.go:611-614) is answered from the sweep's bitmap column: no flatten, no launch
  if (!pre_matched && review->kind == GK_REVIEW_OBJECT && review->source == GK_SRC_EMPTY && review->json && !(review->operation && *review->operation)) {
    gk_engine::Resident& R = e->resident;
    std::unique_lock<std::mutex> rl(R.mu, std::try_to_lock);
    if (rl.owns_lock() && R.swept) {
      auto it = R.by_text.find(text_hash(review->json, review->json_len));
      if (it != R.by_text.end()) {
        const gk_engine::ResObj& o = R.objs[it->second];
        const gk_review_in want = resident_review_in(R, o);
        auto same = [](const char* a, size_t an, const char* b, size_t bn) { return (an == 0 && bn == 0) || (a && b && an == bn && memcmp(a, b, an) == 0); };
        if (o.alive && same(o.json.data(), o.json.size(), review->json, review->json_len) &&
            same(want.namespace_json, want.namespace_len, review->namespace_json, review->namespace_len) &&
            same(want.ns_object_json, want.ns_object_len, review->ns_object_json, review->ns_object_len)) {
          std::string js, err;
          int st = GK_OK;
          try {
            if (resident_answer(e, it->second, &js, &st, &err, false, wanted)) {
              if (stats) { stats->batch_size = 0; stats->queue_us = 0; stats->device_us = 0; stats->total_us = 0; }
              if (st != GK_OK) return fail(st, err);
              char* buf = (char*)malloc(js.size() + 1);
              memcpy(buf, js.c_str(), js.size() + 1);
              *results_json = buf;
              put_trace("from the resident sweep's bitmaps (the review is a swept object: no flatten, no launch)", js, nullptr);
              return GK_OK;
            }
          } catch (const std::exception&) { /* fall through to the regular path */ }
        }
      }
    }
  }
  gk_engine::Request req;
  req.in = review;
  req.pre_matched = pre_matched;
  req.arrived = std::chrono::steady_clock::now();
  gk_engine::Batcher& B = e->batcher;
  for (;;) {
    // enqueue only while workers that will serve the queue exist -- decided under the batcher's lock: a request queued
    // between gk_batcher_stop's `stop = true` and the moment the workers have gone would wait for ever
    std::unique_lock<std::mutex> l(B.mu);
    if (B.running && !B.stop) {
      B.queue.push_back(&req);
      B.cv.notify_all();
      req.cv.wait(l, [&] { return req.done; });
      break;
    }
    if (B.stop) return fail(GK_ERR_INVALID, "the admission batcher is shutting down");
    l.unlock();
    int rc = gk_batcher_start(e, nullptr);
    if (rc != GK_OK) return rc;
  }
  // render this review's results from its column of the batch's bitmaps -- in the caller's thread
  std::string results;
  bool on_host = false;
  if (req.status == GK_OK && req.batch) {
    try {
      bool too_big = false;
      for (uint32_t k = 0; k < req.batch->ev->n_host_evaluated; k++) on_host = on_host || req.batch->ev->host_evaluated[k] == req.index;
      results = query_results_json(e, req.batch->table, *req.batch->ev, req.index, *review, &too_big, wanted);
      if (too_big) { req.status = GK_ERR_LIMIT; req.error = "review is beyond the engine's limits (more than 255 elements in an array that constraint predicates iterate, or an object where they iterate array elements)"; }
    } catch (const RegoError& ex) { req.status = GK_ERR_REGO; req.error = ex.what();
    } catch (const std::exception& ex) { req.status = GK_ERR_INTERNAL; req.error = ex.what(); }
    req.batch.reset();
  }
  if (stats) {
    stats->batch_size = req.batch_size;
    stats->queue_us = req.queue_us;
    stats->device_us = req.device_us;
    stats->total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - req.arrived).count();
  }
  if (req.status != GK_OK) return fail(req.status, req.error);
  char* buf = (char*)malloc(results.size() + 1);
  memcpy(buf, results.c_str(), results.size() + 1);
  *results_json = buf;
  gk_query_stats mine{};
  mine.batch_size = req.batch_size; mine.queue_us = req.queue_us; mine.device_us = req.device_us;
  put_trace(on_host ? (pre_matched ? "by the host evaluator (the review is beyond the device's limits; pre-matched by the caller: violation sets by the concrete evaluator)"
                                   : "by the host evaluator (the review is beyond the device's limits: match layer from the stripped review on the device, violation sets by the concrete evaluator)")
                    : (pre_matched ? "on the device (pre-matched by the caller: violation bitmaps of the batch's one launch; messages rendered by the host evaluator for the flagged pairs)"
                                   : "on the device (match + violation bitmaps of the batch's one launch; messages rendered by the host evaluator for the flagged pairs)"), results, &mine);
  return GK_OK;
}

int gk_debug_set(const char* key, int64_t value) {
  if (!key) return fail(GK_ERR_INVALID, "NULL argument");
  if (strcmp(key, "dict_facts") == 0) { g_debug_dict_facts.store(value != 0); return GK_OK; }
  if (strcmp(key, "group_max") == 0) { g_debug_group_max.store((int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20))); return GK_OK; }
  if (strcmp(key, "fold_match_labels") == 0) { g_test_fold_match_labels.store(value != 0); return GK_OK; }
  return fail(GK_ERR_NOT_FOUND, std::string("gk_debug_set: unknown key ") + key);
}

void gk_jit_quiesce(void) { dev_jit_quiesce(); }
void gk_jit_cache_drop_memory(void) { dev_jit_cache_drop_memory(); }
const char* gk_jit_cache_dir(void) { return dev_jit_cache_dir(); }
void gk_jit_cache_stats(uint64_t* cache_hits, uint64_t* compiles) {
  uint64_t h = 0, c = 0;
  dev_jit_cache_stats(&h, &c);
  if (cache_hits) *cache_hits = h;
  if (compiles) *compiles = c;
}

int gk_dump(gk_engine* e, char** text_out) {
  if (!e || !text_out) return fail(GK_ERR_INVALID, "NULL argument");
  try {
    ensure_plan(e);
    std::ostringstream os;
    std::shared_lock<std::shared_mutex> l(e->plan_rw);
    const HostPlan& p = e->fast;
    os << "constraints=" << p.dims.n_constraints << " viol_formulas=" << p.n_viol << " match_formulas=" << p.n_match
       << " preds=" << p.dims.n_preds << " scopes=" << p.dims.n_scopes << " code_words=" << p.dims.n_code
       << " gwords=" << p.dims.n_gwords << " acc_words=" << p.dims.acc_words << " lds_bytes_per_64_reviews=" << p.dims.acc_words * 256
       << " paths=" << p.dims.n_paths << "\n";
    for (size_t i = 0; i < p.preds.size(); i++)
      os << "pred " << i << " op=" << (int)p.preds[i].op << " dst=" << (int)p.preds[i].dst << " scope=" << (int)p.preds[i].scope
         << " bit=" << p.preds[i].bit << " " << pattern_to_string(p.pred_patterns[i]) << "\n";
    {
      std::shared_lock<std::shared_mutex> rl(e->mu);
      for (auto& c : e->constraints) if (c.alive) os << "constraint " << c.id << " " << c.kind << "/" << c.name << " viol: " << f_to_string(c.viol) << "\n";
    }
    std::string s = os.str();
    char* buf = (char*)malloc(s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    *text_out = buf;
    return GK_OK;
  } catch (const std::exception& ex) { return fail(GK_ERR_INTERNAL, ex.what()); }
}

}  // extern "C"
