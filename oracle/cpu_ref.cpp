// ORACLE / CPU BASELINE (test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use
// it; the product path never loads this library).
//
// "Restated-reference CPU": the reference's serial audit loop in its own algorithmic shape, compiled, so that bench.py
// can time the CPU path beside the GPU path on the same box (the Go/OPA reference itself cannot
// be built here -- pure Go, no Go toolchain -- so there is no oracle/_ref).
//
//   for obj in objects:                                    pkg/audit/manager.go:591-642, 706-719 (one Review per object)
//     request := marshal(obj)                              pkg/target/target.go:159-179 unstructuredToAdmissionRequest
//     for constraint in constraints:                       frameworks Client.Review: Matcher.Match per constraint
//       obj', old' := unmarshal(request.object/oldObject)  pkg/target/matcher.go:73-93 (re-decoded for EVERY constraint)
//       match.Matches(constraint.match, obj', ns, source)  pkg/mutation/match/match.go:32-258
//     for matched constraint: evaluate the template        rego Driver.Query -> violation set
//
// The Match layer below is an independent C++ restatement of match.go / wildcard.go (a third implementation next to
// oracle/match.py and the device's compiled match formulas); HandleReview normalisation, the JSON reader and the
// tree-walking Rego evaluator are the engine's host-side components (gatekeeper_amd/csrc: flatten.cpp normalize_*,
// value.hpp, pe.cpp Template::render -- the evaluator that also renders the product's messages), linked in.
// tests/test_cpu_ref.py pins this loop against the pure-Python oracle.
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../gatekeeper_amd/csrc/flatten.hpp"
#include "../gatekeeper_amd/csrc/pe.hpp"
#include "../include/gkgpu.h"

using namespace gk;

namespace {

thread_local std::string g_err;

// ------------------------------------------------------------------------------------------------ wildcard.go:17-41
bool glob_matches(const std::string& w, const std::string& s) {
  const bool lead = !w.empty() && w.front() == '*', trail = !w.empty() && w.back() == '*';
  if (lead && trail) {
    std::string mid = w.size() >= 2 ? w.substr(1, w.size() - 2) : std::string();   // "*" and "**" -> ""
    return s.find(mid) != std::string::npos;
  }
  if (lead) { std::string suf = w.substr(1); return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0; }
  if (trail) { std::string pre = w.substr(0, w.size() - 1); return s.size() >= pre.size() && s.compare(0, pre.size(), pre) == 0; }
  return w == s;
}
bool glob_matches_generate_name(const std::string& w, const std::string& s) {
  const bool lead = !w.empty() && w.front() == '*', trail = !w.empty() && w.back() == '*';
  if (lead && trail) { std::string mid = w.size() >= 2 ? w.substr(1, w.size() - 2) : std::string(); return s.find(mid) != std::string::npos; }
  if (trail) { std::string pre = w.substr(0, w.size() - 1); return s.size() >= pre.size() && s.compare(0, pre.size(), pre) == 0; }
  return false;
}

// ------------------------------------------------------------------------------------------------ label selectors
// metav1.LabelSelectorAsSelector + labels.NewRequirement validation (apimachinery, third-party): qualified-name keys,
// label values, operator / values arity.
bool alnum(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'); }
bool name_part_ok(const std::string& n) {   // ([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9], 1..63
  if (n.empty() || n.size() > 63 || !alnum(n.front()) || !alnum(n.back())) return false;
  for (char c : n) if (!alnum(c) && c != '-' && c != '_' && c != '.') return false;
  return true;
}
bool dns_subdomain_ok(const std::string& p) {   // lower-case labels separated by dots, <= 253
  if (p.empty() || p.size() > 253) return false;
  size_t i = 0;
  while (i <= p.size()) {
    size_t j = p.find('.', i);
    if (j == std::string::npos) j = p.size();
    if (j == i) return false;
    for (size_t k = i; k < j; k++) {
      char c = p[k];
      bool lower = (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9');
      if (!lower && !(c == '-' && k != i && k + 1 != j)) return false;
    }
    i = j + 1;
  }
  return true;
}
bool label_key_ok(const std::string& k) {
  size_t slashes = 0;
  for (char c : k) slashes += c == '/';
  if (slashes > 1) return false;
  if (slashes == 1) { size_t s = k.find('/'); return dns_subdomain_ok(k.substr(0, s)) && name_part_ok(k.substr(s + 1)); }
  return name_part_ok(k);
}
bool label_value_ok(const std::string& v) { return v.empty() || name_part_ok(v); }

struct Requirement { std::string key; int op; std::vector<std::string> values; };   // 0 In, 1 NotIn, 2 Exists, 3 DoesNotExist
struct Selector { bool present = false, valid = true, everything = false; std::vector<Requirement> reqs; };

Selector compile_selector(const Value* sel) {
  Selector s;
  if (!sel || !sel->is_object()) return s;
  s.present = true;
  const Value* ml = sel->get("matchLabels");
  const Value* me = sel->get("matchExpressions");
  size_t n = (ml && ml->is_object() ? ml->size() : 0) + (me && me->is_array() ? me->size() : 0);
  if (n == 0) { s.everything = true; return s; }
  if (ml && ml->is_object())
    for (auto& kv : ml->pairs()) {
      if (!kv.first.is_string() || !label_key_ok(kv.first.str()) || !kv.second.is_string() || !label_value_ok(kv.second.str())) { s.valid = false; return s; }
      s.reqs.push_back({kv.first.str(), 0, {kv.second.str()}});
    }
  if (me && me->is_array())
    for (auto& e : me->items()) {
      std::string op = obj_string(e, "operator"), key = obj_string(e, "key");
      int o = op == "In" ? 0 : op == "NotIn" ? 1 : op == "Exists" ? 2 : op == "DoesNotExist" ? 3 : -1;
      const Value* vs = e.get("values");
      size_t nv = vs && vs->is_array() ? vs->size() : 0;
      if (o < 0 || !label_key_ok(key) || (o <= 1 && nv == 0) || (o >= 2 && nv != 0)) { s.valid = false; return s; }
      Requirement r{key, o, {}};
      if (vs && vs->is_array())
        for (auto& v : vs->items()) { if (!v.is_string() || !label_value_ok(v.str())) { s.valid = false; return s; } r.values.push_back(v.str()); }
      s.reqs.push_back(r);
    }
  return s;
}

// unstructured GetLabels(): NestedStringMap -- a labels field that is not a map of strings yields no labels at all
bool selector_matches(const Selector& s, const Value& obj) {
  if (s.everything) return true;
  const Value* md = obj.get("metadata");
  const Value* lb = md ? md->get("labels") : nullptr;
  bool usable = lb && lb->is_object();
  if (usable) for (auto& kv : lb->pairs()) if (!kv.second.is_string()) usable = false;
  for (const Requirement& r : s.reqs) {
    const Value* v = usable ? lb->get(r.key.c_str()) : nullptr;
    bool in = false;
    if (v) for (auto& x : r.values) if (x == v->str()) in = true;
    switch (r.op) {
      case 0: if (!in) return false; break;
      case 1: if (in) return false; break;
      case 2: if (!v) return false; break;
      default: if (v) return false; break;
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------ match.go:32-258
struct KindSel { std::vector<std::string> kinds, groups; };
struct MatchSpec {
  bool present = false;   // spec.match is an object (target.go:246-261: otherwise match everything)
  std::vector<KindSel> kinds;
  std::string scope, name, source;
  std::vector<std::string> namespaces, excluded;
  Selector label_sel, ns_sel;
};

std::vector<std::string> string_list(const Value* v) {
  std::vector<std::string> out;
  if (v && v->is_array()) for (auto& x : v->items()) if (x.is_string()) out.push_back(x.str());
  return out;
}
bool list_has(const std::vector<std::string>& l, const std::string& x) { for (auto& y : l) if (y == x) return true; return false; }

MatchSpec compile_match_spec(const Value* m) {
  MatchSpec s;
  if (!m || !m->is_object()) return s;
  s.present = true;
  const Value* ks = m->get("kinds");
  if (ks && ks->is_array()) for (auto& k : ks->items()) s.kinds.push_back({string_list(k.get("kinds")), string_list(k.get("apiGroups"))});
  s.scope = obj_string(*m, "scope");
  s.name = obj_string(*m, "name");
  s.source = obj_string(*m, "source");
  s.namespaces = string_list(m->get("namespaces"));
  s.excluded = string_list(m->get("excludedNamespaces"));
  s.label_sel = compile_selector(m->get("labelSelector"));
  s.ns_sel = compile_selector(m->get("namespaceSelector"));
  return s;
}

enum Tri { NO = 0, YES = 1, ERR = 2 };

Tri matches(const MatchSpec& m, const Value& obj, const Value* ns, int source) {
  std::string group, version, kind;
  obj_gvk(obj, &group, &version, &kind);
  const bool is_ns = kind == "Namespace" && group.empty();
  const std::string name = obj_string(obj, "metadata", "name"), ns_field = obj_string(obj, "metadata", "namespace");
  // kindsMatch
  if (!m.kinds.empty()) {
    bool any = false;
    for (const KindSel& k : m.kinds) {
      if (!(k.kinds.empty() || list_has(k.kinds, "*") || list_has(k.kinds, kind))) continue;
      if (k.groups.empty() || list_has(k.groups, "*") || list_has(k.groups, group)) { any = true; break; }
    }
    if (!any) return NO;
  }
  // scopeMatch
  const bool has_ns = !ns_field.empty() || ns != nullptr;
  if (m.scope == "Cluster" && !(is_ns || !has_ns)) return NO;
  if (m.scope == "Namespaced" && !(!is_ns && has_ns)) return NO;
  // namespacesMatch / excludedNamespacesMatch
  bool have_name = true;
  std::string ns_name;
  if (is_ns) ns_name = name;
  else if (ns) ns_name = obj_string(*ns, "metadata", "name");
  else if (!ns_field.empty()) ns_name = ns_field;
  else have_name = false;
  if (!m.namespaces.empty() && have_name) {
    bool any = false;
    for (auto& w : m.namespaces) if (glob_matches(w, ns_name)) { any = true; break; }
    if (!any) return NO;
  }
  if (!m.excluded.empty() && have_name)
    for (auto& w : m.excluded) if (glob_matches(w, ns_name)) return NO;
  // labelSelectorMatch
  if (m.label_sel.present) {
    if (!m.label_sel.valid) return ERR;
    if (!selector_matches(m.label_sel, obj)) return NO;
  }
  // namespaceSelectorMatch
  if (m.ns_sel.present && !(!is_ns && !ns && ns_field.empty())) {
    if (!m.ns_sel.valid) return ERR;
    if (is_ns) { if (!selector_matches(m.ns_sel, obj)) return NO; }
    else if (!ns) return ERR;   // "namespace selector for namespace-scoped object but missing Namespace"
    else if (!selector_matches(m.ns_sel, *ns)) return NO;
  }
  // namesMatch
  if (!m.name.empty() && !(glob_matches(m.name, name) || glob_matches_generate_name(m.name, obj_string(obj, "metadata", "generateName")))) return NO;
  // sourceMatch
  std::string want = m.source.empty() ? "All" : m.source;
  if (want != "All" && want != "Original" && want != "Generated") return ERR;
  if (source == GK_SRC_EMPTY && want != "All") return ERR;
  if (want == "All") return YES;
  if (source == GK_SRC_INVALID) return ERR;
  return ((want == "Original" && source == GK_SRC_ORIGINAL) || (want == "Generated" && source == GK_SRC_GENERATED)) ? YES : NO;
}

struct Constraint {
  std::string kind, name;
  Value params;
  MatchSpec match;
  std::shared_ptr<Template> tmpl;
};

}  // namespace

struct TemplateSrc { std::string kind, rego; std::vector<std::string> libs; };
struct cpuref {
  std::map<std::string, std::shared_ptr<Template>> templates;   // lower-cased kind
  std::vector<Constraint> constraints;
  // sources, so that every worker thread can build PRIVATE templates / parameters: the evaluator's values are
  // reference counted, and 256 threads sharing one AST would spend their time on the same cache lines
  std::vector<TemplateSrc> template_src;
  std::vector<std::string> constraint_src;
  NsCache ns_cache;
};

namespace {
std::string lower(std::string s) { for (auto& c : s) if (c >= 'A' && c <= 'Z') c += 32; return s; }
}

extern "C" {

const char* cpuref_last_error(void) { return g_err.c_str(); }
cpuref* cpuref_create(void) { return new cpuref(); }
void cpuref_destroy(cpuref* r) { delete r; }

int cpuref_add_template(cpuref* r, const char* kind, const char* rego, const char* const* libs, size_t nlibs) {
  try {
    std::vector<std::string> ls;
    for (size_t i = 0; i < nlibs; i++) ls.emplace_back(libs[i]);
    r->templates[lower(kind)] = std::make_shared<Template>(rego, ls);
    r->template_src.push_back({kind, rego, ls});
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// rows of the result bitmaps follow the order of these calls
int cpuref_add_constraint(cpuref* r, const char* json, size_t len) {
  try {
    Value c = parse_json(json, len);
    Constraint k;
    k.kind = obj_string(c, "kind");
    k.name = obj_string(c, "metadata", "name");
    auto it = r->templates.find(lower(k.kind));
    if (it == r->templates.end()) { g_err = "unknown template " + k.kind; return -1; }
    k.tmpl = it->second;
    const Value* spec = c.get("spec");
    const Value* p = spec ? spec->get("parameters") : nullptr;
    k.params = (p && !p->is_null()) ? *p : Value::object({});
    k.match = compile_match_spec(spec ? spec->get("match") : nullptr);
    r->constraints.push_back(k);
    r->constraint_src.emplace_back(json, len);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// The serial review loop over `n` reviews on `threads` host threads (contiguous slices; 1 = the reference's shape).
//   viol / err : [n_constraints][ceil(n / 64)] bitmaps (may be NULL): pair violates / Matcher.Match failed (autoreject)
//   results    : [n_constraints] results (distinct {msg, details} members of the violation sets) -- what
//                pkg/audit/manager.go:902 counts;  rejected[i] (may be NULL) = 1 if HandleReview rejects review i
//   seconds    : wall-clock of the loop
int cpuref_review(cpuref* r, const gk_review_in* reviews, size_t n, int threads, uint64_t* viol, uint64_t* err, uint64_t* results,
                  uint8_t* rejected, double* seconds) {
  const size_t nc = r->constraints.size(), nt = (n + 63) / 64;
  if (viol) memset(viol, 0, nc * nt * 8);
  if (err) memset(err, 0, nc * nt * 8);
  if (results) memset(results, 0, nc * 8);
  if (rejected) memset(rejected, 0, n);
  if (threads < 1) threads = 1;
  // slices of whole bitmap words so that threads never share a word
  const size_t words_per = (nt + threads - 1) / threads;
  std::vector<std::vector<uint64_t>> part_results(threads, std::vector<uint64_t>(nc, 0));
  std::vector<std::string> errors(threads);
  auto work = [&](int w) {
    try {
      const size_t lo = std::min(n, (size_t)w * words_per * 64), hi = std::min(n, ((size_t)w + 1) * words_per * 64);
      const Value inventory;
      // thread-private policy state (see cpuref): same sources, own objects
      std::unique_ptr<cpuref> mine;
      const cpuref* pol = r;
      if (threads > 1) {
        mine.reset(cpuref_create());
        for (auto& t : r->template_src) {
          std::vector<const char*> ls;
          for (auto& l : t.libs) ls.push_back(l.c_str());
          if (cpuref_add_template(mine.get(), t.kind.c_str(), t.rego.c_str(), ls.data(), ls.size())) throw std::runtime_error(g_err);
        }
        for (auto& c : r->constraint_src) if (cpuref_add_constraint(mine.get(), c.data(), c.size())) throw std::runtime_error(g_err);
        pol = mine.get();
      }
      for (size_t i = lo; i < hi; i++) {
        const gk_review_in& in = reviews[i];
        ReviewDoc doc;
        try {
          Value body = parse_json(in.json, in.json_len);
          Value mns = (in.namespace_json && in.namespace_len) ? parse_json(in.namespace_json, in.namespace_len) : Value();
          Value nso = (in.ns_object_json && in.ns_object_len) ? parse_json(in.ns_object_json, in.ns_object_len) : Value();
          if (in.kind == GK_REVIEW_OBJECT) doc = normalize_object(body, mns, nso, in.source, in.operation ? in.operation : "", r->ns_cache);
          else doc = normalize_admission_request(body, mns, nso, in.source, r->ns_cache);
        } catch (const std::exception&) { if (rejected) rejected[i] = 1; continue; }
        // target.go:159-179: the object travels inside the request as raw JSON ...
        const Value* o = doc.request.get("object");
        const Value* old = doc.request.get("oldObject");
        const std::string raw_obj = (o && o->is_object()) ? to_json(*o) : std::string();
        const std::string raw_old = (old && old->is_object()) ? to_json(*old) : std::string();
        const Value* ns = doc.match_ns.defined() ? &doc.match_ns : nullptr;
        for (size_t c = 0; c < nc; c++) {
          const Constraint& k = pol->constraints[c];
          Tri m = YES;
          if (k.match.present) {
            // ... and matcher.go:73-93 decodes it again for every constraint
            Value dobj = raw_obj.empty() ? Value() : parse_json(raw_obj.data(), raw_obj.size());
            Value dold = raw_old.empty() ? Value() : parse_json(raw_old.data(), raw_old.size());
            if ((dobj.defined() && obj_string(dobj, "kind").empty()) || (dold.defined() && obj_string(dold, "kind").empty())) m = ERR;   // ErrRequestObject
            else if (!dobj.defined() && !dold.defined()) m = ERR;   // neither object nor old object are defined
            else {
              m = NO;
              for (const Value* cand : {&dobj, &dold}) {
                if (!cand->defined()) continue;
                Tri t = matches(k.match, *cand, ns, in.source);
                if (t == ERR) { m = ERR; break; }
                if (t == YES) { m = YES; break; }
              }
            }
          }
          if (m == ERR) { if (err) err[c * nt + i / 64] |= 1ull << (i % 64); continue; }
          if (m != YES) continue;
          size_t nres = k.tmpl->render(doc.request, k.params, inventory).size();
          if (nres) { if (viol) viol[c * nt + i / 64] |= 1ull << (i % 64); part_results[w][c] += nres; }
        }
      }
    } catch (const std::exception& e) { errors[w] = e.what(); }
  };
  auto t0 = std::chrono::steady_clock::now();
  if (threads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int w = 0; w < threads; w++) th.emplace_back(work, w);
    for (auto& t : th) t.join();
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (auto& e : errors) if (!e.empty()) { g_err = e; return -1; }
  if (results) for (int w = 0; w < threads; w++) for (size_t c = 0; c < nc; c++) results[c] += part_results[w][c];
  return 0;
}

size_t cpuref_n_constraints(const cpuref* r) { return r->constraints.size(); }

}  // extern "C"
