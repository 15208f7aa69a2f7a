// See lower.hpp.
#include "lower.hpp"

#include "builtins.hpp"
#include "regex.hpp"
#include "vm_core.hpp"

#include <atomic>
#include <functional>
#include <regex>
#include <set>

namespace gk {
std::atomic<int> g_test_fold_match_labels{0};   // test aid, set through gk_debug_set("fold_match_labels", 1): nothing reads the environment while lowering
bool review_fact_leaf(const SPath& p) {
  Pattern pat;
  for (const Step& st : p) { PatStep ps; if (st.iter) ps.any = true; else ps.key = st.key; pat.push_back(ps); }
  return review_fact_pattern(pat);
}

// ================================================================================================ match blocks
namespace {

Step key_step(const std::string& k) { Step s; s.key = k; return s; }
SPath mk_path(std::initializer_list<const char*> keys) { SPath p; for (auto k : keys) p.push_back(key_step(k)); return p; }

int ctz32(uint32_t x) { int n = 0; while (!(x & 1)) { x >>= 1; n++; } return n; }
FP flag_f(uint32_t mask) { Atom a; a.kind = Atom::FLAG; a.flag = (uint32_t)ctz32(mask); return f_atom(a); }

FP atom_k(Atom::Kind kind, const SPath& p, const Value& k, int cmp = C_EQ) {
  Atom a; a.kind = kind; a.path = p; a.k = k; a.cmp = cmp;
  return f_atom(a);
}

// pkg/wildcard/wildcard.go:17-41
FP glob_f(const SPath& p, const std::string& w, bool generate_name) {
  bool pre = !w.empty() && w.front() == '*', suf = !w.empty() && w.back() == '*';
  if (pre && suf) {
    std::string inner = w.substr(1);
    if (!inner.empty() && inner.back() == '*') inner.pop_back();
    return atom_k(Atom::STR_CONTAINS, p, Value::string(inner));
  }
  if (pre) return generate_name ? f_false() : atom_k(Atom::STR_SUFFIX, p, Value::string(w.substr(1)));
  if (suf) return atom_k(Atom::STR_PREFIX, p, Value::string(w.substr(0, w.size() - 1)));
  return generate_name ? f_false() : atom_k(Atom::CMP, p, Value::string(w), C_EQ);
}

// apimachinery validation.IsQualifiedName / IsValidLabelValue (k8s.io/apimachinery v0.36.3, third-party)
bool valid_label_key(const std::string& k) {
  static const std::regex name_re("^([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9]$");
  static const std::regex sub_re("^[a-z0-9]([-a-z0-9]*[a-z0-9])?(\\.[a-z0-9]([-a-z0-9]*[a-z0-9])?)*$");
  size_t n = std::count(k.begin(), k.end(), '/');
  std::string name = k;
  if (n == 1) {
    std::string prefix = k.substr(0, k.find('/'));
    name = k.substr(k.find('/') + 1);
    if (prefix.empty() || prefix.size() > 253 || !std::regex_match(prefix, sub_re)) return false;
  } else if (n > 1) return false;
  return !name.empty() && name.size() <= 63 && std::regex_match(name, name_re);
}
bool valid_label_value(const std::string& v) {
  static const std::regex re("^(([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9])?$");
  return v.size() <= 63 && std::regex_match(v, re);
}

struct Req { std::string key; int op; std::vector<std::string> vals; };   // op: 0 In/Equals, 1 NotIn, 2 Exists, 3 DoesNotExist

// metav1.LabelSelectorAsSelector: false => conversion error; *everything => empty selector
// The conversion with its error text (labels.NewRequirement / LabelSelectorAsSelector validation, in evaluation order:
// matchLabels by sorted key, then matchExpressions as listed).  "" = valid.  The same function decides the device's
// error formula (compile_match) and words the host's autoreject message (engine.cpp), so the two cannot disagree.
static std::string selector_convert(const Value& sel, std::vector<Req>* reqs, bool* everything) {
  *everything = false;
  const Value* ml = sel.get("matchLabels");
  const Value* me = sel.get("matchExpressions");
  size_t n = (ml && ml->is_object() ? ml->size() : 0) + (me && me->is_array() ? me->size() : 0);
  if (n == 0) { *everything = true; return ""; }
  auto bad_key = [](const std::string& k) { return "key: Invalid value: \"" + k + "\": name part must be non-empty"; };
  auto bad_val = [](const std::string& k, const std::string& v) { return "values[0][" + k + "]: Invalid value: \"" + v + "\""; };
  if (ml && ml->is_object()) {
    std::vector<std::pair<std::string, const Value*>> kv;
    for (auto& p : ml->pairs()) kv.emplace_back(p.first.is_string() ? p.first.str() : std::string(), &p.second);
    std::sort(kv.begin(), kv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (auto& p : kv) {
      if (!valid_label_key(p.first)) return bad_key(p.first);
      if (!p.second->is_string() || !valid_label_value(p.second->str())) return bad_val(p.first, p.second->is_string() ? p.second->str() : to_json(*p.second));
      reqs->push_back({p.first, 0, {p.second->str()}});
    }
  }
  if (me && me->is_array())
    for (auto& e : me->items()) {
      std::string op = obj_string(e, "operator"), key = obj_string(e, "key");
      int o = op == "In" ? 0 : op == "NotIn" ? 1 : op == "Exists" ? 2 : op == "DoesNotExist" ? 3 : -1;
      if (o < 0) return "\"" + op + "\" is not a valid label selector operator";
      if (!valid_label_key(key)) return bad_key(key);
      std::vector<std::string> vals;
      const Value* vs = e.get("values");
      size_t nvals = vs && vs->is_array() ? vs->size() : 0;
      if (o <= 1 && nvals == 0) return "values: Invalid value: []: for 'in', 'notin' operators, values set can't be empty";
      if (o >= 2 && nvals != 0) return "values: Invalid value: values set must be empty for exists and does not exist";
      if (vs && vs->is_array())
        for (auto& v : vs->items()) {
          if (!v.is_string() || !valid_label_value(v.str())) return bad_val(key, v.is_string() ? v.str() : to_json(v));
          vals.push_back(v.str());
        }
      reqs->push_back({key, o, vals});
    }
  return "";
}

bool selector_reqs(const Value& sel, std::vector<Req>* reqs, bool* everything) { return selector_convert(sel, reqs, everything).empty(); }

FP selector_f(const std::vector<Req>& reqs, const SPath& labels, uint32_t bad_flag) {
  FP nb = f_not(flag_f(bad_flag));
  FP r = f_true();
  for (const Req& q : reqs) {
    SPath p = labels;
    p.push_back(key_step(q.key));
    ValueVec vals;
    for (auto& v : q.vals) vals.push_back(Value::string(v));
    FP in = f_and(atom_k(Atom::STR_IN_SET, p, Value::set(vals)), nb);
    Atom d; d.kind = Atom::DEFINED; d.path = p;
    FP has = f_and(f_atom(d), nb);
    switch (q.op) {
      case 0: r = f_and(r, in); break;
      case 1: r = f_and(r, f_not(in)); break;
      case 2: r = f_and(r, has); break;
      default: r = f_and(r, f_not(has)); break;
    }
  }
  return r;
}

std::vector<std::string> str_list(const Value* v) {
  std::vector<std::string> out;
  if (v && v->is_array()) for (auto& x : v->items()) if (x.is_string()) out.push_back(x.str());
  return out;
}

struct CandFlags { uint32_t has, is_ns, has_nsfield, has_nsname, labels_bad; const char* m; const char* obj; };

void match_candidate(const Value& m, const CandFlags& c, FP* out_match, FP* out_err) {
  FP p = f_true(), err = f_false();
  FP is_ns = flag_f(c.is_ns), ns_present = flag_f(RF_NS_PRESENT), has_nsfield = flag_f(c.has_nsfield);
  // 1. kinds (match.go:181-201)
  const Value* kinds = m.get("kinds");
  if (kinds && kinds->is_array() && kinds->size() > 0) {
    FP any = f_false();
    for (auto& kk : kinds->items()) {
      std::vector<std::string> ks = str_list(kk.get("kinds")), gs = str_list(kk.get("apiGroups"));
      auto member = [&](const std::vector<std::string>& l, const char* field) -> FP {
        if (l.empty()) return f_true();
        for (auto& x : l) if (x == "*") return f_true();
        ValueVec vals;
        for (auto& x : l) vals.push_back(Value::string(x));
        return atom_k(Atom::STR_IN_SET, mk_path({"$m", c.m, field}), Value::set(vals));
      };
      any = f_or(any, f_and(member(ks, "kind"), member(gs, "group")));
    }
    p = f_and(p, any);
  }
  // 2. scope (match.go:214-227)
  std::string scope = obj_string(m, "scope");
  FP has_ns = f_or(has_nsfield, ns_present);
  if (scope == "Cluster") p = f_and(p, f_or(is_ns, f_not(has_ns)));
  else if (scope == "Namespaced") p = f_and(p, f_and(f_not(is_ns), has_ns));
  // 3/4. namespaces, excludedNamespaces (match.go:118-179)
  SPath nsname = mk_path({"$m", c.m, "nsname"});
  FP no_nsname = f_not(flag_f(c.has_nsname));
  std::vector<std::string> nss = str_list(m.get("namespaces"));
  if (!nss.empty()) {
    FP any = f_false();
    for (auto& w : nss) any = f_or(any, glob_f(nsname, w, false));
    p = f_and(p, f_or(no_nsname, any));
  }
  std::vector<std::string> ex = str_list(m.get("excludedNamespaces"));
  if (!ex.empty()) {
    FP any = f_false();
    for (auto& w : ex) any = f_or(any, glob_f(nsname, w, false));
    p = f_and(p, f_or(no_nsname, f_not(any)));
  }
  // 5. labelSelector (match.go:103-116)
  const Value* ls = m.get("labelSelector");
  if (ls && ls->is_object()) {
    std::vector<Req> reqs;
    bool everything;
    if (!selector_reqs(*ls, &reqs, &everything)) { err = f_or(err, p); p = f_false(); }
    else if (!everything) p = f_and(p, selector_f(reqs, mk_path({c.obj, "metadata", "labels"}), c.labels_bad));
  }
  // 6. namespaceSelector (match.go:73-101)
  const Value* nsel = m.get("namespaceSelector");
  if (nsel && nsel->is_object()) {
    FP cluster_scoped = f_and(f_not(is_ns), f_and(f_not(ns_present), f_not(has_nsfield)));
    std::vector<Req> reqs;
    bool everything;
    if (!selector_reqs(*nsel, &reqs, &everything)) {
      err = f_or(err, f_and(p, f_not(cluster_scoped)));
      p = f_and(p, cluster_scoped);
    } else {
      FP missing = f_and(f_not(is_ns), f_and(f_not(ns_present), has_nsfield));   // ns-scoped but no Namespace
      err = f_or(err, f_and(p, missing));
      FP own = everything ? f_true() : selector_f(reqs, mk_path({c.obj, "metadata", "labels"}), c.labels_bad);
      FP nsl = everything ? f_true() : selector_f(reqs, mk_path({"$ns", "metadata", "labels"}), RF_NS_LABELS_BAD);
      FP ok = f_or(cluster_scoped, f_or(f_and(is_ns, own), f_and(f_and(f_not(is_ns), ns_present), nsl)));
      p = f_and(p, ok);
    }
  }
  // 7. name (match.go:203-212)
  std::string name = obj_string(m, "name");
  if (!name.empty())
    p = f_and(p, f_or(glob_f(mk_path({"$m", c.m, "name"}), name, false), glob_f(mk_path({"$m", c.m, "gname"}), name, true)));
  // 8. source (match.go:229-253)
  std::string src = obj_string(m, "source");
  if (src.empty()) src = "All";
  if (src != "All" && src != "Original" && src != "Generated") { err = f_or(err, p); p = f_false(); }
  else if (src != "All") {
    FP empty = f_and(f_and(f_not(flag_f(RF_SRC_ORIGINAL)), f_not(flag_f(RF_SRC_GENERATED))), f_and(f_not(flag_f(RF_SRC_ALL)), f_not(flag_f(RF_SRC_INVALID))));
    err = f_or(err, f_and(p, f_or(empty, flag_f(RF_SRC_INVALID))));
    p = f_and(p, flag_f(src == "Original" ? RF_SRC_ORIGINAL : RF_SRC_GENERATED));
  }
  FP has = flag_f(c.has);
  *out_match = f_and(has, p);
  *out_err = f_and(has, err);
}

}  // namespace

std::string selector_error_text(const Value& sel) {
  std::vector<Req> reqs;
  bool everything;
  return selector_convert(sel, &reqs, &everything);
}


MatchFormulas compile_match(const Value& m) {
  if (!m.defined() || !m.is_object()) return {f_true(), f_false()};
  CandFlags o{RF_HAS_OBJ, RF_OBJ_IS_NS, RF_OBJ_HAS_NSFIELD, RF_OBJ_HAS_NSNAME, RF_OBJ_LABELS_BAD, "o", "object"};
  CandFlags old{RF_HAS_OLD, RF_OLD_IS_NS, RF_OLD_HAS_NSFIELD, RF_OLD_HAS_NSNAME, RF_OLD_LABELS_BAD, "old", "oldObject"};
  FP mo, eo, mold, eold;
  match_candidate(m, o, &mo, &eo);
  match_candidate(m, old, &mold, &eold);
  FP none = f_and(f_not(flag_f(RF_HAS_OBJ)), f_not(flag_f(RF_HAS_OLD)));
  // gkReviewToObject runs before any matcher (matcher.go:32-35): an object / oldObject that does not decode into an
  // Unstructured makes every constraint WITH a match block fail with ErrRequestObject (target_test.go:690-703)
  FP bad = f_or(flag_f(RF_OBJ_BAD), flag_f(RF_OLD_BAD));
  MatchFormulas r;
  r.match = f_and(f_not(bad), f_or(mo, f_and(f_not(eo), mold)));
  r.error = f_or(bad, f_or(f_or(eo, f_and(f_and(f_not(mo), f_not(eo)), eold)), none));
  return r;
}

// ================================================================================================ lowering
namespace {

void conjuncts(const FP& f, std::vector<FP>& out) {
  if (f->kind == FNode::AND) { for (auto& k : f->kids) conjuncts(k, out); }
  else if (f->kind != FNode::T) out.push_back(f);
}

bool is_prefix(const SPath& a, const SPath& b) {   // a is a (non-strict) prefix of b
  if (a.size() > b.size()) return false;
  for (size_t i = 0; i < a.size(); i++) {
    if (a[i].iter != b[i].iter) return false;
    if (a[i].iter ? a[i].q != b[i].q : a[i].key != b[i].key) return false;
  }
  return true;
}

bool path_mentions(const SPath& p, int q) { for (auto& s : p) if (s.iter && s.q == q) return true; return false; }
bool mentions_q(const FP& f, int q) {
  switch (f->kind) {
    case FNode::T: case FNode::F: return false;
    case FNode::ATOM: return f->atom.q == q || path_mentions(f->atom.path, q) || path_mentions(f->atom.path2, q);
    case FNode::EXISTS: if (path_mentions(f->base, q)) return true;   // fallthrough to kids
    default: for (auto& k : f->kids) if (mentions_q(k, q)) return true; return false;
  }
}

// count(split(p)) >= n  &  for every i < n:  !(count(split(p)) > i & split(p)[i] != a_i)      (the PSP path_matches idiom)
//   ==>  one SPLIT_PREFIX predicate: the first n components of split(trim(p)) equal a_0..a_{n-1}
void fuse_split_prefix(std::vector<FP>& conj) {
  for (size_t ci = 0; ci < conj.size(); ci++) {
    const FP& c = conj[ci];
    if (!(c->kind == FNode::ATOM && c->atom.kind == Atom::SPLIT_COUNT && c->atom.cmp == C_GE && c->atom.k.is_int && c->atom.k.i >= 1 && c->atom.k.i <= 32)) continue;
    const Atom& cnt = c->atom;
    int n = (int)cnt.k.i;
    std::string pkey = spath_to_string(cnt.path);
    std::vector<Value> comps(n);
    std::vector<int> where(n, -1);
    for (size_t j = 0; j < conj.size(); j++) {
      const FP& t = conj[j];
      if (t->kind != FNode::NOT) continue;
      std::vector<FP> inner;
      conjuncts(t->kids[0], inner);
      const Atom* cmp = nullptr;
      bool ok = true;
      for (auto& x : inner) {
        if (x->kind != FNode::ATOM) { ok = false; break; }
        const Atom& a = x->atom;
        if (spath_to_string(a.path) != pkey) { ok = false; break; }
        if (a.kind == Atom::TYPE && a.mask == (1u << T_STRING)) continue;
        if (a.kind == Atom::SPLIT_COUNT && a.cut == cnt.cut && a.sep == cnt.sep && a.cmp == C_GT) continue;   // implied by count >= n when its bound < n (checked below)
        if (a.kind == Atom::SPLIT_CMP && a.cut == cnt.cut && a.sep == cnt.sep && a.cmp == C_NE && a.k.is_string() && !cmp) { cmp = &a; continue; }
        ok = false;
        break;
      }
      if (!ok || !cmp || cmp->idx < 0 || cmp->idx >= n || where[cmp->idx] >= 0) continue;
      for (auto& x : inner) if (x->atom.kind == Atom::SPLIT_COUNT && !(x->atom.k.is_int && x->atom.k.i == cmp->idx)) ok = false;
      if (!ok) continue;
      comps[cmp->idx] = cmp->k;
      where[cmp->idx] = (int)j;
    }
    bool all = true;
    for (int w : where) if (w < 0) all = false;
    if (!all) continue;
    Atom f;
    f.kind = Atom::SPLIT_PREFIX;
    f.path = cnt.path; f.cut = cnt.cut; f.sep = cnt.sep;
    f.k = Value::array(ValueVec(comps.begin(), comps.end()));
    std::vector<FP> out;
    for (size_t j = 0; j < conj.size(); j++) {
      if (j == ci) { out.push_back(f_atom(f)); continue; }
      if (std::find(where.begin(), where.end(), (int)j) != where.end()) continue;
      out.push_back(conj[j]);
    }
    conj.swap(out);
    ci = (size_t)-1;   // restart: indices moved
  }
}

// ---- memo of the three pure passes (simplify, pin_pass, fold_dict), keyed by NODE: the formulas prepared for one constraint's result
// counting -- a row per threshold, the flag with its pairwise terms -- share their (large, merged) bodies as nodes, and every
// prepare_constraint walked them afresh: 6 s per K8sContainerLimits constraint.  Scoped (PrepMemoScope): the nodes outlive it.
struct PrepMemo { std::unordered_map<const FNode*, FP> simp, pin, fold; std::vector<FP> keep; };
static thread_local PrepMemo* g_prep_memo = nullptr;
static FP simplify_impl(const FP& f);
static FP fold_dict_impl(const FP& f);
static FP pin_pass_impl(const FP& f);
#define GK_MEMO_PASS(name, field)                                                                       \
  FP name(const FP& f) {                                                                                \
    if (!g_prep_memo || f->kind == FNode::T || f->kind == FNode::F || f->kind == FNode::ATOM) return name##_impl(f); \
    auto it = g_prep_memo->field.find(f.get());                                                         \
    if (it != g_prep_memo->field.end()) return it->second;                                              \
    FP r = name##_impl(f);                                                                              \
    g_prep_memo->keep.push_back(f);   /* (the key stays a live node) */                                 \
    g_prep_memo->field.emplace(f.get(), r);                                                             \
    return r;                                                                                           \
  }
GK_MEMO_PASS(simplify, simp)
GK_MEMO_PASS(fold_dict, fold)
GK_MEMO_PASS(pin_pass, pin)
#undef GK_MEMO_PASS

static FP simplify_impl(const FP& f) {
  switch (f->kind) {
    case FNode::AND: {
      std::vector<FP> flat0, flat;
      for (auto& k : f->kids) conjuncts(simplify(k), flat0);
      for (auto& k : flat0) {   // De Morgan inside conjunctions: !(a | b)  ->  !a & !b
        if (k->kind == FNode::NOT && k->kids[0]->kind == FNode::OR) for (auto& x : k->kids[0]->kids) conjuncts(simplify(f_not(x)), flat);
        else flat.push_back(k);
      }
      // dedupe
      std::vector<FP> uniq;
      std::set<std::string> seen;
      for (auto& k : flat) { if (k->kind == FNode::F) return f_false(); if (seen.insert(f_to_string(k)).second) uniq.push_back(k); }
      // "the key is a member name" is implied by a positive string test on the same key (an index fails every such test)
      {
        std::vector<FP> u2;
        for (auto& k : uniq) {
          bool implied = false;
          if (k->kind == FNode::ATOM && k->atom.kind == Atom::KEYCMP && k->atom.cmp == KC_ISNAME)
            for (auto& o : uniq) if (o->kind == FNode::ATOM && o->atom.kind == Atom::KEYCMP && o->atom.q == k->atom.q && o->atom.cmp >= KC_PREFIX && o->atom.cmp < KC_ISNAME) implied = true;
          if (!implied) u2.push_back(k);
        }
        uniq.swap(u2);
      }
      // drop DEFINED(p) implied by another positive conjunct on p or below p
      std::vector<FP> keep;
      for (size_t i = 0; i < uniq.size(); i++) {
        const FP& k = uniq[i];
        bool implied = false;
        if (k->kind == FNode::ATOM && k->atom.kind == Atom::DEFINED) {
          for (size_t j = 0; j < uniq.size() && !implied; j++) {
            if (j == i) continue;
            const FP& o = uniq[j];
            if (o->kind == FNode::ATOM && o->atom.kind != Atom::KEYCMP && o->atom.kind != Atom::FLAG) {
              if (o->atom.kind == Atom::DEFINED && o->atom.path.size() == k->atom.path.size() && j > i) continue;
              if (is_prefix(k->atom.path, o->atom.path)) implied = true;
              if (o->atom.kind == Atom::VEQ && is_prefix(k->atom.path, o->atom.path2)) implied = true;
            } else if (o->kind == FNode::EXISTS && is_prefix(k->atom.path, o->base)) implied = true;
          }
        }
        if (!implied) keep.push_back(k);
      }
      fuse_split_prefix(keep);
      return f_all(keep);
    }
    case FNode::OR: {
      std::vector<FP> uniq;
      std::set<std::string> seen;
      for (auto& k : f->kids) { FP s = simplify(k); if (s->kind == FNode::T) return f_true(); if (s->kind != FNode::F && seen.insert(f_to_string(s)).second) uniq.push_back(s); }
      return f_any(uniq);
    }
    case FNode::NOT: return f_not(simplify(f->kids[0]));
    case FNode::EXISTS: {
      // hoist conjuncts that do not depend on q:   E q. (A(q) & G)  ==  (E q. A(q)) & G
      FP body = simplify(f->kids[0]);
      std::vector<FP> conj, dep, indep;
      conjuncts(body, conj);
      for (auto& c : conj) (mentions_q(c, f->q) ? dep : indep).push_back(c);
      if (indep.empty()) return f_exists_like(*f, body);   // (E2, "at least two children satisfy the body", hoists the same way)
      FP inner = f_exists_like(*f, f_all(dep));
      return simplify(f_and(inner, f_all(indep)));
    }
    default: return f;
  }
}

FP rename_f(const FP& f, const std::map<int, int>& m);

// ---- dictionary predicates (dexpr.hpp): every boolean sub-formula that talks about ONE leaf only and contains a DICT atom
// becomes a single DICT atom, evaluated by the flattener per distinct value of that leaf.
bool dict_foldable_atom(const Atom& a) {
  switch (a.kind) {
    case Atom::DICT: case Atom::TYPE: case Atom::TRUTHY: case Atom::DEFINED: case Atom::STR_PREFIX: case Atom::STR_SUFFIX: case Atom::STR_CONTAINS:
    case Atom::STR_IN_SET: case Atom::STR_REGEX: case Atom::COUNT_CMP: case Atom::SPLIT_CMP: case Atom::SPLIT_COUNT: case Atom::SPLIT_PREFIX: return true;
    case Atom::CMP: return !(a.k.is_array() || a.k.is_object() || a.k.is_set());
    default: return false;
  }
}
// leaf of a leaf-local formula ("" = not leaf-local); has_dict: it contains a DICT atom
static std::string leaf_of_uncached(const FP& f, bool* has_dict);
std::string leaf_of(const FP& f, bool* has_dict) {
  if (f->leaf_state == 0) {   // (asked again and again for the same sub-formulas while conjunctions are folded per leaf)
    bool d = false;
    f->leaf_text = leaf_of_uncached(f, &d);
    f->leaf_state = d ? 2 : 1;
  }
  if (f->leaf_state == 2 && !f->leaf_text.empty()) *has_dict = true;
  return f->leaf_text;
}
static std::string leaf_of_uncached(const FP& f, bool* has_dict) {
  switch (f->kind) {
    case FNode::ATOM:
      if (!dict_foldable_atom(f->atom)) return "";
      if (f->atom.kind == Atom::DICT) *has_dict = true;
      return spath_to_string(f->atom.path);
    case FNode::NOT: return leaf_of(f->kids[0], has_dict);
    case FNode::AND: case FNode::OR: {
      std::string l;
      for (auto& k : f->kids) { std::string x = leaf_of(k, has_dict); if (x.empty() || (!l.empty() && x != l)) return ""; l = x; }
      return l;
    }
    default: return "";
  }
}
// is the formula false whenever its leaf is absent? (device predicates only fire on rows that exist)
static bool needs_leaf_uncached(const FP& f);
bool needs_leaf(const FP& f) {
  if (f->needs_leaf_state == 0) f->needs_leaf_state = needs_leaf_uncached(f) ? 2 : 1;
  return f->needs_leaf_state == 2;
}
static bool needs_leaf_uncached(const FP& f) {
  switch (f->kind) {
    case FNode::ATOM: return true;
    case FNode::AND: for (auto& k : f->kids) if (needs_leaf(k)) return true; return false;
    case FNode::OR: for (auto& k : f->kids) if (!needs_leaf(k)) return false; return true;
    default: return false;
  }
}
const SPath* leaf_path_of(const FP& f) {
  if (f->kind == FNode::ATOM) return &f->atom.path;
  for (auto& k : f->kids) if (const SPath* p = leaf_path_of(k)) return p;
  return nullptr;
}
DX to_dx(const FP& f) {
  switch (f->kind) {
    case FNode::T: return dx_const(Value::boolean(true));
    case FNode::F: return dx_const(Value::boolean(false));
    case FNode::NOT: return dx_node(DExpr::NOT, {to_dx(f->kids[0])});
    case FNode::AND: case FNode::OR: { std::vector<DX> a; for (auto& k : f->kids) a.push_back(to_dx(k)); return dx_node(f->kind == FNode::AND ? DExpr::AND : DExpr::OR, a); }
    case FNode::ATOM: {
      const Atom& a = f->atom;
      switch (a.kind) {
        case Atom::DICT: return a.dx;
        case Atom::DEFINED: return dx_const(Value::boolean(true));   // the expression is only evaluated for leaves that exist
        case Atom::TRUTHY: return dx_node(DExpr::TRUTHY, {dx_leaf()});
        case Atom::TYPE: return dx_node(DExpr::TYPE_MASK, {dx_leaf()}, "", 0, a.mask);
        case Atom::CMP: return dx_node(DExpr::CMP, {dx_leaf(), dx_const(a.k)}, "", a.cmp);
        case Atom::STR_PREFIX: return dx_node(DExpr::TRUTHY, {dx_node(DExpr::CALL, {dx_leaf(), dx_const(a.k)}, "startswith")});
        case Atom::STR_SUFFIX: return dx_node(DExpr::TRUTHY, {dx_node(DExpr::CALL, {dx_leaf(), dx_const(a.k)}, "endswith")});
        case Atom::STR_CONTAINS: return dx_node(DExpr::TRUTHY, {dx_node(DExpr::CALL, {dx_leaf(), dx_const(a.k)}, "contains")});
        case Atom::STR_REGEX: return dx_node(DExpr::TRUTHY, {dx_node(DExpr::CALL, {dx_const(a.k), dx_leaf()}, "re_match")});
        case Atom::COUNT_CMP:   // member count of a container leaf (the flattener hands containers over with their size)
          return dx_node(DExpr::AND, {dx_node(DExpr::TYPE_MASK, {dx_leaf()}, "", 0, (1u << T_ARRAY) | (1u << T_OBJECT)), dx_node(DExpr::CMP, {dx_node(DExpr::CALL, {dx_leaf()}, "count"), dx_const(a.k)}, "", a.cmp)});
        case Atom::STR_IN_SET: { std::vector<DX> alts; for (auto& v : a.k.items()) alts.push_back(dx_node(DExpr::CMP, {dx_leaf(), dx_const(v)}, "", C_EQ)); return dx_node(DExpr::OR, alts); }
        case Atom::SPLIT_CMP: case Atom::SPLIT_COUNT: case Atom::SPLIT_PREFIX: {
          // split(trim(leaf, cut), sep) as the partial evaluator spells it (pe.cpp leaf_local); a leaf that is no string, a component
          // that does not exist: the builtin / $index is undefined and the comparison false -- what the row predicates answer
          DX base = dx_leaf();
          if (a.cut) base = dx_node(DExpr::CALL, {base, dx_const(Value::string(std::string(1, a.cut)))}, "trim");
          if (a.kind == Atom::SPLIT_PREFIX) {   // trim(s) == P  |  startswith(trim(s), P + sep)      (P = the components joined by sep)
            std::string joined;
            for (size_t i = 0; i < a.k.items().size(); i++) { if (i) joined.push_back(a.sep); joined += a.k.items()[i].str(); }
            return dx_node(DExpr::OR, {dx_node(DExpr::CMP, {base, dx_const(Value::string(joined))}, "", C_EQ),
                                       dx_node(DExpr::TRUTHY, {dx_node(DExpr::CALL, {base, dx_const(Value::string(joined + std::string(1, a.sep)))}, "startswith")})});
          }
          DX arr = dx_node(DExpr::CALL, {base, dx_const(Value::string(std::string(1, a.sep)))}, "split");
          if (a.kind == Atom::SPLIT_COUNT) return dx_node(DExpr::CMP, {dx_node(DExpr::CALL, {arr}, "count"), dx_const(a.k)}, "", a.cmp);
          return dx_node(DExpr::CMP, {dx_node(DExpr::CALL, {arr, dx_const(Value::integer(a.idx))}, "$index"), dx_const(a.k)}, "", a.cmp);
        }
        default: break;
      }
    }
    default: break;
  }
  throw Unsupported("unsupported on the device plan: formula is not leaf-local");
}
FP dict_atom(const SPath& leaf, DX dx, FP alt = nullptr) { Atom a; a.kind = Atom::DICT; a.path = leaf; a.dx = std::move(dx); a.alt = std::move(alt); return f_atom(a); }

// PROMOTION (round 5).  Tests that read a string leaf's BYTES -- prefix / suffix / contains / regex / split components -- cost the
// kernel tens of vector instructions per row and constraint (an image row of the 200-template corpus is tested against the
// constants of up to 22 constraints, one predicate after the other: 1 460 clocks per 64-row chunk against 230 in configs[2]'s
// kernel).  They are pure functions of ONE leaf: a leaf-local group that holds such a test becomes a dictionary expression like the
// quantity arithmetic of K8sContainerLimits always was -- evaluated by the flattener once per DISTINCT value of the leaf with the
// concrete builtins (memoised per engine), shipped as a bit of the <leaf>.$d row, tested on the device with one AND.  The atom
// keeps the group as `alt`: where the dictionary cannot take the expression the lowering evaluates the rows as before.
// GK_DICT_STRINGS=0 switches the promotion off (tuning / A-B aid).
static bool promote_strings() { static const bool on = !(getenv("GK_DICT_STRINGS") && atoi(getenv("GK_DICT_STRINGS")) == 0); return on; }
static bool reads_string_bytes(const FP& f) {
  switch (f->kind) {
    case FNode::ATOM:
      switch (f->atom.kind) {
        case Atom::STR_PREFIX: case Atom::STR_SUFFIX: case Atom::STR_CONTAINS: case Atom::STR_REGEX: case Atom::SPLIT_CMP: case Atom::SPLIT_COUNT: case Atom::SPLIT_PREFIX: return true;
        case Atom::STR_IN_SET: return true;
        default: return false;
      }
    case FNode::NOT: case FNode::AND: case FNode::OR: for (auto& k : f->kids) if (reads_string_bytes(k)) return true; return false;
    default: return false;
  }
}
// MATCH FACTS (round 5).  The five strings the match layer compares -- review.$m.<o|old>.{kind, group, name, gname, nsname}, a row with a
// string header each: 160 of configs[2]'s ~380 bytes per review -- are tested against constants only.  Every leaf-local group on one of
// them becomes a dictionary expression whatever it reads, and the five leaves of a candidate share ONE dictionary row,
// review.$m.<o|old>.$d (flatten.hpp DictRegistry "match group": their bits come from one 62-bit space): a sweep reads one 16-byte row
// per review where it read five rows and five headers.  GK_DICT_MATCH=0 keeps the rows (A/B aid).
bool match_fact_leaf(const SPath& p) {
  static const bool on = !(getenv("GK_DICT_MATCH") && atoi(getenv("GK_DICT_MATCH")) == 0);
  return on && p.size() == 3 && !p[0].iter && p[0].key == "$m" && !p[1].iter && !p[2].iter && !p[2].key.empty() && p[2].key[0] != '$';
}
// REVIEW FACTS (round 6).  A leaf that no iteration leads to -- review.object.metadata.name (every library template's message names the
// object: `def(review.object.metadata.name)`, a row per review and a chunk per row group for ONE bit), the labels a selector or a
// template names, review.$ns.metadata.labels.<k>, spec.hostNetwork ... -- occurs at most once per review.  Every leaf-local group on
// such a leaf becomes a dictionary expression whatever it reads, and all of them share ONE row per review, review.$r.$d
// (flatten.hpp review_fact_pattern; 62 bits for the loaded policy set: what does not fit keeps the rows).  configs[2]: a third of the
// rows a sweep still read after the element carriers.  GK_DICT_FACTS=0 keeps a row per leaf (A/B aid).
// (folding a MATCH formula promotes the match facts only: the labels a selector names keep their rows, which the counting plans --
//  frozen, in the counting space -- lower the same way.  gk_debug_set("fold_match_labels", 1), test aid: promote them as well -- the counting
//  plans then read label rows a pruned table does not hold, which is how tests/test_pruned.py reaches render_needed's unanswered plans)
static thread_local bool g_fold_match_only = false;
static bool promotable_leaf(const SPath& p) {   // (the other synthetic subtrees -- $ns -- keep their rows)
  if (p.empty()) return false;
  if (match_fact_leaf(p)) return true;
  if (review_fact_leaf(p)) return true;   // (in a match formula as well: the review facts live in the main space, whoever asks -- see the DICT lowering)
  if (g_fold_match_only && !g_test_fold_match_labels.load(std::memory_order_relaxed)) return false;
  for (auto& st : p) if (!st.iter && !st.key.empty() && st.key[0] == '$') return false;
  return true;
}
static bool promotable(const std::vector<FP>& g) {
  if (!promote_strings() || g.empty()) return false;
  const SPath* lp = leaf_path_of(g[0]);
  if (!lp || !promotable_leaf(*lp)) return false;
  if (match_fact_leaf(*lp) || review_fact_leaf(*lp)) return true;
  for (auto& k : g) if (reads_string_bytes(k)) return true;
  return false;
}

static FP fold_dict_impl(const FP& f) {
  switch (f->kind) {
    case FNode::NOT: return f_not(fold_dict(f->kids[0]));
    case FNode::EXISTS: return f_exists_like(*f, fold_dict(f->kids[0]));
    case FNode::OR: {
      // (E x in B. P(x)) | (E y in B. Q(y))  ==  E x in B. (P(x) | Q(x)): the alternatives of one template function over
      // the same array then meet in ONE body, where they fold per leaf
      std::vector<FP> kids;
      std::map<std::string, size_t> by_base;
      for (auto& k : f->kids) {
        if (k->kind != FNode::EXISTS || k->two) { kids.push_back(k); continue; }   // (E2 nodes count: they are not merged)
        const std::string bk = spath_to_string(k->base);
        auto it = by_base.find(bk);
        if (it == by_base.end()) { by_base[bk] = kids.size(); kids.push_back(k); continue; }
        const FP& first = kids[it->second];
        std::map<int, int> m{{k->q, first->q}};
        kids[it->second] = f_exists(first->q, first->base, f_or(first->kids[0], rename_f(k->kids[0], m)));
      }
      // (K & A) | (K & B)  ==  K & (A | B): conjuncts shared by every alternative of a merged body (the key test of
      // `spec[field][_]`, for one) go back in front, where the lowering expects them
      for (auto& k : kids) {
        if (k->kind != FNode::EXISTS || k->kids[0]->kind != FNode::OR) continue;
        std::vector<std::vector<FP>> alts;
        for (auto& d : k->kids[0]->kids) { alts.emplace_back(); conjuncts(d, alts.back()); }
        std::vector<FP> common;
        for (auto& c0 : alts[0]) {
          const std::string key = f_to_string(c0);
          bool all = true;
          for (size_t i = 1; i < alts.size() && all; i++) { bool has = false; for (auto& c : alts[i]) has = has || f_to_string(c) == key; all = has; }
          if (all) common.push_back(c0);
        }
        if (common.empty()) continue;
        FP body = f_false();
        for (auto& a : alts) {
          FP d = f_true();
          for (auto& c : a) { bool shared = false; for (auto& c0 : common) shared = shared || f_to_string(c0) == f_to_string(c); if (!shared) d = f_and(d, c); }
          body = f_or(body, d);
        }
        k = f_exists_like(*k, f_and(f_all(common), body));
      }
      // group the leaf-local alternatives by leaf: a group with a DICT atom whose members all need the leaf folds
      std::map<std::string, std::vector<FP>> groups;
      std::vector<std::string> order;
      std::vector<FP> rest;
      for (auto& k : kids) {
        bool hd = false;
        std::string l = leaf_of(k, &hd);
        if (l.empty() || !needs_leaf(k)) { rest.push_back(fold_dict(k)); continue; }
        if (!groups.count(l)) order.push_back(l);
        groups[l].push_back(k);
      }
      FP r = f_false();
      for (auto& l : order) {
        auto& g = groups[l];
        bool hd = false;
        for (auto& k : g) leaf_of(k, &hd);
        if (hd) {
          std::vector<DX> a;
          for (auto& k : g) a.push_back(to_dx(k));
          r = f_or(r, dict_atom(*leaf_path_of(g[0]), a.size() == 1 ? a[0] : dx_node(DExpr::OR, a)));
        } else if (promotable(g)) {
          std::vector<DX> a;
          for (auto& k : g) a.push_back(to_dx(k));
          r = f_or(r, dict_atom(*leaf_path_of(g[0]), a.size() == 1 ? a[0] : dx_node(DExpr::OR, a), f_any(g)));
        } else for (auto& k : g) r = f_or(r, fold_dict(k));
      }
      for (auto& k : rest) r = f_or(r, k);
      return r;
    }
    case FNode::AND: {
      // group the conjuncts by leaf; a group with a DICT atom and at least one conjunct that needs the leaf to exist folds
      std::vector<FP> rest;
      std::map<std::string, std::vector<FP>> groups;
      std::vector<std::string> order;
      for (auto& k : f->kids) {
        bool hd = false;
        std::string l = leaf_of(k, &hd);
        if (l.empty()) { rest.push_back(fold_dict(k)); continue; }
        if (!groups.count(l)) order.push_back(l);
        groups[l].push_back(k);
      }
      FP r = f_true();
      for (auto& l : order) {
        auto& g = groups[l];
        bool hd = false, needs = false;
        for (auto& k : g) { leaf_of(k, &hd); needs = needs || needs_leaf(k); }
        if (hd && needs) {
          std::vector<DX> a;
          for (auto& k : g) a.push_back(to_dx(k));
          r = f_and(r, dict_atom(*leaf_path_of(g[0]), a.size() == 1 ? a[0] : dx_node(DExpr::AND, a)));
        } else if (needs && promotable(g)) {
          std::vector<DX> a;
          for (auto& k : g) a.push_back(to_dx(k));
          r = f_and(r, dict_atom(*leaf_path_of(g[0]), a.size() == 1 ? a[0] : dx_node(DExpr::AND, a), f_all(g)));
        } else for (auto& k : g) r = f_and(r, fold_dict(k));
      }
      for (auto& k : rest) r = f_and(r, k);
      return r;
    }
    case FNode::ATOM: {
      // a lone DICT atom stays; a lone NOT(DICT) cannot be answered from a row that may not exist -- it reaches the
      // lowering as NOT(bit test), which is right: no row <=> leaf absent or expression false
      if (f->atom.kind != Atom::DICT && dict_foldable_atom(f->atom) && promotable({f})) return dict_atom(f->atom.path, to_dx(f), f);
      return f;
    }
    default: return f;
  }
}

// ---- alpha-normalised canonical text: quantifier ids renumbered in order of first appearance, so structurally
// identical (sub)formulas compiled for different constraints share predicates, derived bits and result slots.
void collect_q(const FP& f, std::vector<int>& order) {
  auto see = [&](int q) { if (q >= 0 && std::find(order.begin(), order.end(), q) == order.end()) order.push_back(q); };
  auto see_path = [&](const SPath& p) { for (auto& s : p) if (s.iter) see(s.q); };
  switch (f->kind) {
    case FNode::T: case FNode::F: return;
    case FNode::ATOM: see_path(f->atom.path); see_path(f->atom.path2); see(f->atom.q); return;
    case FNode::EXISTS: see(f->q); see_path(f->base);   // fallthrough
    default: for (auto& k : f->kids) collect_q(k, order);
  }
}
SPath rename_path(const SPath& p, const std::map<int, int>& m) {
  SPath o = p;
  for (auto& s : o) if (s.iter) { auto it = m.find(s.q); if (it != m.end()) s.q = it->second; }
  return o;
}
static bool path_renamed(const SPath& p, const std::map<int, int>& m) { for (auto& s : p) if (s.iter) { auto it = m.find(s.q); if (it != m.end() && it->second != s.q) return true; } return false; }
static bool q_renamed(int q, const std::map<int, int>& m) { if (q < 0) return false; auto it = m.find(q); return it != m.end() && it->second != q; }
FP rename_f(const FP& f, const std::map<int, int>& m) {
  switch (f->kind) {
    case FNode::T: case FNode::F: return f;
    case FNode::ATOM: {
      if (!path_renamed(f->atom.path, m) && !path_renamed(f->atom.path2, m) && !q_renamed(f->atom.q, m)) return f;   // (nothing to rename: shared, not copied)
      Atom a = f->atom;
      a.path = rename_path(a.path, m); a.path2 = rename_path(a.path2, m);
      if (a.q >= 0) { auto it = m.find(a.q); if (it != m.end()) a.q = it->second; }
      return f_atom(a);
    }
    default: {
      bool same = !path_renamed(f->base, m) && !q_renamed(f->q, m);
      std::vector<FP> kids;
      kids.reserve(f->kids.size());
      for (auto& k : f->kids) { kids.push_back(rename_f(k, m)); if (kids.back().get() != k.get()) same = false; }
      if (same) return f;
      FNode n = *f;
      n.kids = std::move(kids);
      n.base = rename_path(n.base, m);
      if (n.q >= 0) { auto it = m.find(n.q); if (it != m.end()) n.q = it->second; }
      n.canon_set = false; n.canon.clear(); n.leaf_state = 0; n.needs_leaf_state = 0; n.leaf_text.clear();   // (an edited copy: its text is derived afresh)
      return std::make_shared<const FNode>(n);
    }
  }
}
// `first`: quantifiers that must get the lowest numbers (free loop variables), in order
std::string canon(const FP& f, const std::vector<int>& first = {}) {
  std::vector<int> order = first;
  collect_q(f, order);
  std::map<int, int> m;
  for (size_t i = 0; i < order.size(); i++) m[order[i]] = 1000000 + (int)i;
  return f_to_string(rename_f(f, m));
}

// key of a constant string as the device compares it: packed bytes (len <= 7) or hash32
uint64_t string_key(const std::string& s) {
  if (s.size() <= 7) { uint64_t k = 0; memcpy(&k, s.data(), s.size()); return k; }
  return hash32(s);
}

struct Lowerer {
  PathDict* dict;
  DictRegistry* reg = nullptr;
  bool frozen = false;   // no new registry entries (PlanBuilder)
  bool counting = false; // dictionary predicates of the registry's counting space: <leaf>.$c (PlanBuilder::use_counting_space)
  HostPlan plan;
  PlanCaps caps;
  std::map<std::string, uint32_t> global_bits;            // canonical pred key -> global bit
  std::map<std::string, uint32_t> scope_ids;              // element pattern -> scope
  std::vector<Pattern> scope_patterns;
  std::vector<size_t> present_pred;   // per scope: index of its P_PRESENT predicate in plan.preds (SIZE_MAX: the root scope has none)
  std::vector<int> scope_level;
  std::vector<std::map<std::string, uint32_t>> elem_bits;  // per scope
  std::vector<std::map<std::string, uint32_t>> val_slots;  // per scope
  std::vector<uint32_t> scope_nbits;
  uint32_t n_gbits = 1;                                    // bit 0 = overflow
  uint64_t regs_used = 0;
  std::map<int, uint32_t> looped;                          // quantifier -> scope
  std::set<int> pass;                                      // pass-through quantifiers
  std::map<int, PatStep> wild;                             // wildcard quantifiers (global existentials)
  std::map<std::string, uint32_t> cheap_strings;
  std::map<std::string, uint32_t> derived_global;          // canonical closed EXISTS -> global bit
  std::vector<std::map<std::string, uint32_t>> derived_elem;   // per scope: canonical EXISTS over (elem) -> elem bit
  std::vector<uint32_t>* cur_code = nullptr;               // where emit() appends
  std::vector<std::vector<uint32_t>> prologue;             // derived-bit blocks, run before the formulas

  [[noreturn]] void unsupported(const std::string& what) { throw Unsupported("unsupported on the device plan: " + what); }

  int alloc() {
    for (int i = 0; i < 64; i++) if (!(regs_used >> i & 1)) { regs_used |= 1ull << i; return i; }
    unsupported("formula needs more than 64 boolean registers");
  }
  void release(int r) { regs_used &= ~(1ull << r); }
  void emit(uint32_t w) { (cur_code ? cur_code : &plan.code)->push_back(w); }

  uint32_t put_bytes(const std::string& s) {
    auto it = cheap_strings.find(s);
    if (it != cheap_strings.end()) return it->second;
    while (plan.cheap.size() & 15) plan.cheap.push_back(0);
    uint32_t off = (uint32_t)plan.cheap.size();
    plan.cheap.insert(plan.cheap.end(), s.begin(), s.end());
    plan.cheap.push_back(0);                       // device loops may read one zero word past short constants
    while (plan.cheap.size() & 15) plan.cheap.push_back(0);
    cheap_strings[s] = off;
    return off;
  }
  void put_u32(uint32_t v) { for (int i = 0; i < 4; i++) plan.cheap.push_back((uint8_t)(v >> (8 * i))); }

  Pattern pattern_of(const SPath& p) {
    Pattern out;
    for (const Step& s : p) {
      PatStep ps;
      if (!s.iter) { ps.key = s.key; out.push_back(ps); continue; }
      ps.any = true;
      if (looped.count(s.q) || pass.count(s.q)) ps.elems_only = true;
      else {
        auto it = wild.find(s.q);
        if (it == wild.end()) unsupported("free quantifier in a predicate path");
        ps.only = it->second.only;
        ps.except = it->second.except;
        ps.kpreds = it->second.kpreds;
      }
      out.push_back(ps);
    }
    return out;
  }

  std::string atom_key(const Atom& a, const Pattern& pat) {
    return std::to_string((int)a.kind) + "|" + pattern_to_string(pat) + "|" + std::to_string(a.cmp) + "|" + to_term_string(a.k) + "|" +
           std::to_string(a.mask) + "|" + std::to_string((int)a.cut) + "|" + std::to_string((int)a.sep) + "|" + std::to_string(a.idx);
  }

  Pred make_pred(const Atom& a) {
    Pred p{};
    p.cmp = (uint8_t)a.cmp;
    auto put_const_string = [&](const Value& v) { p.a = put_bytes(v.str()); p.b = (uint32_t)v.str().size(); p.k = string_key(v.str()); };
    switch (a.kind) {
      case Atom::DEFINED: p.op = P_DEFINED; break;
      case Atom::TRUTHY: p.op = P_TRUTHY; break;
      case Atom::TYPE: p.op = P_TYPE; p.ctype = (uint8_t)a.mask; break;
      case Atom::CMP:
        p.op = P_CMP;
        switch (a.k.kind) {
          case Value::Null: p.ctype = T_NULL; break;
          case Value::Bool: p.ctype = T_BOOL; p.k = a.k.b ? 1 : 0; break;
          case Value::Number:
            if (a.k.is_int && a.k.i >= (i128)INT64_MIN && a.k.i <= (i128)INT64_MAX) { p.ctype = T_INT; p.k = (uint64_t)(int64_t)a.k.i; }
            else { p.ctype = T_FLOAT; double d = a.k.as_double(); memcpy(&p.k, &d, 8); }
            break;
          case Value::String: p.ctype = T_STRING; put_const_string(a.k); break;
          default: unsupported("comparison with a composite constant");
        }
        break;
      case Atom::STR_PREFIX: p.op = P_STR_PREFIX; put_const_string(a.k); break;
      case Atom::STR_SUFFIX: p.op = P_STR_SUFFIX; put_const_string(a.k); break;
      case Atom::STR_CONTAINS: p.op = P_STR_CONTAINS; put_const_string(a.k); break;
      case Atom::STR_IN_SET: {
        p.op = P_STR_IN_SET;
        std::vector<std::pair<uint32_t, uint32_t>> ents;
        for (auto& v : a.k.items()) ents.emplace_back(put_bytes(v.str()), (uint32_t)v.str().size());
        while (plan.cheap.size() & 3) plan.cheap.push_back(0);
        p.a = (uint32_t)plan.cheap.size();
        p.b = (uint32_t)ents.size();
        size_t i = 0;
        for (auto& v : a.k.items()) {
          const std::string& str = v.str();
          if (str.size() <= 7) { uint64_t k = string_key(str); put_u32((uint32_t)k); put_u32((uint32_t)(k >> 32)); }
          else { put_u32(hash32(str)); put_u32(ents[i].first); }
          put_u32(ents[i].second);
          i++;
        }
        break;
      }
      case Atom::STR_REGEX: {
        auto re = get_regex(a.k.str());
        std::vector<uint8_t> table;
        if (!re || !re->to_dfa(&table)) unsupported("regular expression needs more than 255 DFA states: " + a.k.str());
        p.op = P_REGEX;
        p.a = put_bytes(std::string(table.begin(), table.end()));
        p.b = (uint32_t)table.size();
        break;
      }
      case Atom::SPLIT_CMP:
        p.op = P_SPLIT_CMP; put_const_string(a.k); p.idx = a.idx; p.pad = ((uint32_t)(uint8_t)a.cut << 8) | (uint8_t)a.sep;
        break;
      case Atom::SPLIT_COUNT:
        p.op = P_SPLIT_COUNT; p.k = (uint64_t)(int64_t)a.k.i; p.pad = ((uint32_t)(uint8_t)a.cut << 8) | (uint8_t)a.sep;
        break;
      case Atom::COUNT_CMP: p.op = P_COUNT_CMP; p.k = (uint64_t)(int64_t)a.k.i; break;
      case Atom::SPLIT_PREFIX: {
        std::string joined;
        for (size_t i = 0; i < a.k.items().size(); i++) { if (i) joined.push_back(a.sep); joined += a.k.items()[i].str(); }
        p.op = P_SPLIT_PREFIX; p.a = put_bytes(joined); p.b = (uint32_t)joined.size(); p.pad = ((uint32_t)(uint8_t)a.cut << 8) | (uint8_t)a.sep;
        break;
      }
      default: unsupported("predicate kind");
    }
    return p;
  }

  uint32_t scope_for(const SPath& elem_path) {
    Pattern pat = pattern_of(elem_path);
    for (auto& s : pat) if (s.any && !s.elems_only) unsupported("correlated iteration below an object-key iteration");
    std::string key = pattern_to_string(pat);
    auto it = scope_ids.find(key);
    if (it != scope_ids.end()) return it->second;
    int level = -1;
    for (auto& s : pat) if (s.any) level++;
    if (level > 2) unsupported("correlated iteration nested deeper than 3 arrays");
    if (scope_patterns.size() >= GK_MAX_SCOPES) unsupported("too many element scopes");
    uint32_t id = (uint32_t)scope_patterns.size();
    scope_ids[key] = id;
    scope_patterns.push_back(pat);
    scope_level.push_back(level);
    elem_bits.emplace_back();
    val_slots.emplace_back();
    derived_elem.emplace_back();
    scope_nbits.push_back(1);
    Pred p{};
    p.op = P_PRESENT; p.dst = D_ELEM; p.scope = (uint8_t)id; p.level = (uint8_t)level;
    present_pred.push_back(plan.preds.size());
    plan.preds.push_back(p);
    plan.pred_patterns.push_back(pat);
    return id;
  }

  // the one-element scope of review values that are compared with each other outside any iteration
  uint32_t root_scope() {
    auto it = scope_ids.find("$root");
    if (it != scope_ids.end()) return it->second;
    if (scope_patterns.size() >= GK_MAX_SCOPES) unsupported("too many element scopes");
    uint32_t id = (uint32_t)scope_patterns.size();
    scope_ids["$root"] = id;
    scope_patterns.push_back(Pattern());
    scope_level.push_back((int)GK_LEVEL_ROOT);
    elem_bits.emplace_back();
    val_slots.emplace_back();
    derived_elem.emplace_back();
    scope_nbits.push_back(1);
    present_pred.push_back(SIZE_MAX);
    return id;   // (no P_PRESENT predicate: a stored value marks the element, vm_core.hpp / codegen.cpp)
  }

  // innermost looped quantifier of a path: index of its ITER step, or -1
  int last_looped(const SPath& p) {
    for (int i = (int)p.size() - 1; i >= 0; i--) if (p[i].iter && looped.count(p[i].q)) return i;
    return -1;
  }

  int lower_atom(const Atom& a) {
    int r = alloc();
    if (a.kind == Atom::FLAG) { emit(finst(F_LDF, r, a.flag)); return r; }
    if (a.kind == Atom::KEYCMP) unsupported("key comparison outside a simple existential");
    if (a.kind == Atom::VEQ) {
      uint32_t sc[2], slot[2];
      const SPath* ps[2] = {&a.path, &a.path2};
      bool root_side = false;
      for (int k = 0; k < 2; k++) {
        int li = last_looped(*ps[k]);
        for (size_t j = li + 1; j < ps[k]->size(); j++) if ((*ps[k])[j].iter) unsupported("join on a nested iteration");
        // a value outside every iteration lives in the ROOT scope (one element per review)
        if (li < 0) { sc[k] = root_scope(); root_side = true; }
        else sc[k] = looped[(*ps[k])[li].q];
        Pattern pat = pattern_of(*ps[k]);
        std::string key = pattern_to_string(pat);
        auto it = val_slots[sc[k]].find(key);
        if (it == val_slots[sc[k]].end()) {
          slot[k] = (uint32_t)val_slots[sc[k]].size();
          if (slot[k] >= 8) unsupported("more than 8 joined values on one element scope");
          val_slots[sc[k]][key] = slot[k];
          Pred p{};
          p.op = P_STORE; p.dst = D_ELEM; p.scope = (uint8_t)sc[k]; p.level = (uint8_t)scope_level[sc[k]]; p.bit = (uint16_t)slot[k];
          plan.preds.push_back(p);
          plan.pred_patterns.push_back(pat);
        } else slot[k] = it->second;
      }
      if (root_side) {   // the comparison runs inside the (single-trip) loop over the root scope
        int acc = alloc();
        emit(finst(F_LOOP, root_scope(), 0, acc));
        emit(finst(F_VEQ, r));
        emit(sc[0] | (slot[0] << 8) | (sc[1] << 16) | (slot[1] << 24));
        emit(finst(F_ENDLOOP, acc, r));
        release(r);
        return acc;
      }
      emit(finst(F_VEQ, r));
      emit(sc[0] | (slot[0] << 8) | (sc[1] << 16) | (slot[1] << 24));
      return r;
    }
    if (a.kind == Atom::DICT) {
      // bit test on the leaf's <leaf>.$d row; the bit belongs to (pattern of the leaf, expression) in the engine's registry
      if (!reg) unsupported("dictionary predicate without a registry");
      Pattern leaf_pat = pattern_of(a.path);
      uint32_t bit;
      // (a match fact's expressions live in the main space, whoever asks: the constraint's own violation plan registered them -- the
      //  counting space is for what only the result counts read)
      // (... and so do the review facts: a leaf no iteration leads to keeps its expressions in the main space and its bits in review.$r.$d)
      const bool cnt = counting && !match_fact_leaf(a.path) && !review_fact_leaf(a.path);
      bool in_facts = false;
      if (a.alt) {   // a promoted group of row predicates: what the dictionary cannot take is evaluated from the rows, as before
        bool ok = true;
        for (auto& st : leaf_pat) if (st.any && !st.elems_only && (!st.only.empty() || !st.except.empty() || !st.kpreds.empty())) ok = false;
        if (ok) { try { bit = (cnt ? &reg->counting() : reg)->intern(leaf_pat, a.dx, !frozen, &in_facts, true); } catch (const std::runtime_error&) { ok = false; } }
        if (!ok) { release(r); return lower(a.alt); }
      } else {
      for (auto& st : leaf_pat) if (st.any && !st.elems_only && (!st.only.empty() || !st.except.empty() || !st.kpreds.empty())) unsupported("dictionary predicate under a filtered key iteration");
      try { bit = (cnt ? &reg->counting() : reg)->intern(leaf_pat, a.dx, !frozen, &in_facts, false); } catch (const std::runtime_error& ex) { unsupported(ex.what()); }
      }
      if (cnt) in_facts = false;
      Atom b;
      b.kind = Atom::DICT; b.path = a.path; b.dx = nullptr;
      Step st; st.key = cnt ? "$c" : "$d";
      if (match_fact_leaf(a.path)) b.path.pop_back();   // the candidate's five facts share review.$m.<o|old>.$d
      if (in_facts) { b.path.clear(); Step rs; rs.key = "$r"; b.path.push_back(rs); }   // the review facts share review.$r.$d
      b.path.push_back(st);
      b.mask = bit;
      Pattern pat = pattern_of(b.path);
      std::string key = "dict|" + pattern_to_string(pat) + "|" + std::to_string(bit);
      int li = last_looped(b.path);
      Pred p{};
      p.op = P_BITS; p.k = 1ull << bit;
      if (li < 0) {
        auto it = global_bits.find(key);
        uint32_t gb;
        if (it == global_bits.end()) {
          gb = n_gbits++;
          if (gb > 0xFFFF) unsupported("too many global predicates");
          global_bits[key] = gb;
          p.dst = D_GLOBAL; p.bit = (uint16_t)gb;
          plan.preds.push_back(p);
          plan.pred_patterns.push_back(pat);
        } else gb = it->second;
        emit(finst(F_LDG, r, gb & 0xFF, gb >> 8));
        return r;
      }
      uint32_t sc = looped[b.path[li].q];
      auto it = elem_bits[sc].find(key);
      uint32_t eb;
      if (it == elem_bits[sc].end()) {
        eb = scope_nbits[sc]++;
        if (eb >= ELEM_W0_BITS + 3 * 32) unsupported("too many predicates on one element scope");
        elem_bits[sc][key] = eb;
        p.dst = D_ELEM; p.scope = (uint8_t)sc; p.level = (uint8_t)scope_level[sc]; p.bit = (uint16_t)eb;
        plan.preds.push_back(p);
        plan.pred_patterns.push_back(pat);
      } else eb = it->second;
      emit(finst(F_LDE, r, sc, eb));
      return r;
    }
    Pattern pat = pattern_of(a.path);
    std::string key = atom_key(a, pat);
    int li = last_looped(a.path);
    if (li < 0) {
      auto it = global_bits.find(key);
      uint32_t bit;
      if (it == global_bits.end()) {
        bit = n_gbits++;
        global_bits[key] = bit;
        Pred p = make_pred(a);
        p.dst = D_GLOBAL; p.bit = (uint16_t)bit;
        plan.preds.push_back(p);
        plan.pred_patterns.push_back(pat);
      } else bit = it->second;
      if (bit > 0xFFFF) unsupported("too many global predicates");
      emit(finst(F_LDG, r, bit & 0xFF, bit >> 8));
      return r;
    }
    uint32_t sc = looped[a.path[li].q];
    auto it = elem_bits[sc].find(key);
    uint32_t bit;
    if (it == elem_bits[sc].end()) {
      bit = scope_nbits[sc]++;
      if (bit >= ELEM_W0_BITS + 3 * 32) unsupported("too many predicates on one element scope");
      elem_bits[sc][key] = bit;
      Pred p = make_pred(a);
      p.dst = D_ELEM; p.scope = (uint8_t)sc; p.level = (uint8_t)scope_level[sc]; p.bit = (uint16_t)bit;
      plan.preds.push_back(p);
      plan.pred_patterns.push_back(pat);
    } else bit = it->second;
    emit(finst(F_LDE, r, sc, bit));
    return r;
  }

  static bool path_has_q(const SPath& p, int q) { for (auto& s : p) if (s.iter && s.q == q) return true; return false; }

  // key constraints of quantifier q expressed by conjunct f; returns false if f is not such a constraint
  static bool key_constraint(const FP& f, int q, PatStep* ps) {
    if (f->kind == FNode::ATOM && f->atom.kind == Atom::KEYCMP && f->atom.q == q && f->atom.cmp >= KC_PREFIX) {   // string test on the member name
      ps->kpreds.push_back(KeyPred{(uint8_t)f->atom.cmp, false, f->atom.k.is_string() ? f->atom.k.str() : std::string()});
      return true;
    }
    if (f->kind == FNode::NOT && f->kids[0]->kind == FNode::ATOM && f->kids[0]->atom.kind == Atom::KEYCMP && f->kids[0]->atom.q == q && f->kids[0]->atom.cmp >= KC_PREFIX) {
      const Atom& a = f->kids[0]->atom;
      ps->kpreds.push_back(KeyPred{(uint8_t)a.cmp, true, a.k.is_string() ? a.k.str() : std::string()});
      return true;
    }
    if (f->kind == FNode::ATOM && f->atom.kind == Atom::KEYCMP && f->atom.q == q && f->atom.k.is_string()) {
      if (f->atom.cmp == C_EQ) ps->only.push_back(f->atom.k.str()); else ps->except.push_back(f->atom.k.str());
      return true;
    }
    if (f->kind == FNode::NOT) {
      const FP& in = f->kids[0];
      std::vector<FP> alts;
      if (in->kind == FNode::OR) alts = in->kids; else alts.push_back(in);
      for (auto& x : alts) if (!(x->kind == FNode::ATOM && x->atom.kind == Atom::KEYCMP && x->atom.q == q && x->atom.cmp == C_EQ && x->atom.k.is_string())) return false;
      for (auto& x : alts) ps->except.push_back(x->atom.k.str());
      return true;
    }
    if (f->kind == FNode::OR) {   // key == a | key == b
      for (auto& x : f->kids) if (!(x->kind == FNode::ATOM && x->atom.kind == Atom::KEYCMP && x->atom.q == q && x->atom.cmp == C_EQ && x->atom.k.is_string())) return false;
      if (!ps->only.empty()) return false;
      for (auto& x : f->kids) ps->only.push_back(x->atom.k.str());
      return true;
    }
    return false;
  }

  // f with quantifier q replaced by the constant member name `key`
  static FP pin_key(const FP& f, int q, const std::string& key) {
    auto pin_path = [&](SPath p) { for (auto& s : p) if (s.iter && s.q == q) { s.iter = false; s.q = -1; s.key = key; } return p; };
    switch (f->kind) {
      case FNode::T: case FNode::F: return f;
      case FNode::ATOM: {
        Atom a = f->atom;
        if (a.kind == Atom::KEYCMP && a.q == q) {
          if (a.cmp >= KC_PREFIX) return key_pred_holds(KeyPred{(uint8_t)a.cmp, false, a.k.is_string() ? a.k.str() : std::string()}, key, false) ? f_true() : f_false();
          if (!a.k.is_string()) return a.cmp == C_NE ? f_true() : f_false();
          int c = key.compare(a.k.str());
          bool r = a.cmp == C_EQ ? c == 0 : a.cmp == C_NE ? c != 0 : a.cmp == C_LT ? c < 0 : a.cmp == C_LE ? c <= 0 : a.cmp == C_GT ? c > 0 : c >= 0;
          return r ? f_true() : f_false();
        }
        a.path = pin_path(a.path);
        a.path2 = pin_path(a.path2);
        return f_atom(a);
      }
      case FNode::NOT: return f_not(pin_key(f->kids[0], q, key));
      case FNode::AND: { FP r = f_true(); for (auto& k : f->kids) r = f_and(r, pin_key(k, q, key)); return r; }
      case FNode::OR: { FP r = f_false(); for (auto& k : f->kids) r = f_or(r, pin_key(k, q, key)); return r; }
      case FNode::EXISTS: { FNode proto; proto.q = f->q; proto.base = pin_path(f->base); proto.two = f->two; proto.atleast = f->atleast; return f_exists_like(proto, pin_key(f->kids[0], q, key)); }
    }
    return f;
  }

  // EXISTS chain that reduces to one wildcard predicate
  bool try_flat(const FP& f, std::vector<std::pair<int, PatStep>>& wilds, Atom* out) {
    if (f->two) return false;   // a wildcard predicate cannot count
    int q = f->q;
    std::vector<FP> conj;
    conjuncts(f->kids[0], conj);
    PatStep ps;
    ps.any = true;
    std::vector<FP> rest;
    for (auto& c : conj) if (!key_constraint(c, q, &ps)) rest.push_back(c);
    if (rest.size() > 1) return false;
    wilds.emplace_back(q, ps);
    if (rest.empty()) {
      out->kind = Atom::DEFINED;
      out->path = f->base;
      Step st; st.iter = true; st.q = q;
      out->path.push_back(st);
      return true;
    }
    const FP& r0 = rest[0];
    if (r0->kind == FNode::ATOM) {
      const Atom& a = r0->atom;
      if (a.kind == Atom::KEYCMP || a.kind == Atom::FLAG || a.kind == Atom::VEQ) return false;
      if (!path_has_q(a.path, q)) return false;
      *out = a;
      return true;
    }
    if (r0->kind == FNode::EXISTS) {
      if (!try_flat(r0, wilds, out)) return false;
      return path_has_q(out->path, q);
    }
    return false;
  }

  bool uses_elem_directly(const FP& f, int q) {
    switch (f->kind) {
      case FNode::ATOM: {
        auto last_iter = [](const SPath& p) { for (int i = (int)p.size() - 1; i >= 0; i--) if (p[i].iter) return p[i].q; return -1; };
        if (f->atom.kind == Atom::KEYCMP) return f->atom.q == q;
        if (last_iter(f->atom.path) == q) return true;
        if (f->atom.kind == Atom::VEQ && last_iter(f->atom.path2) == q) return true;
        return false;
      }
      case FNode::T: case FNode::F: return false;
      default:
        for (auto& k : f->kids) if (uses_elem_directly(k, q)) return true;
        return false;
    }
  }

  int lower_exists(const FP& f) {
    int q = f->q;
    if (f->two) {
      // "at least two children": a counting loop when the children form an element scope; anything else is answered with the
      // plain EXISTS -- weaker, i.e. more pairs than necessary are rendered on the host, never fewer (Template::compile_multi)
      const uint64_t regs0 = regs_used;
      std::vector<uint32_t>* const code = cur_code ? cur_code : &plan.code;
      const size_t code0 = code->size();
      const std::map<int, uint32_t> looped0 = looped;
      const std::set<int> pass0 = pass;
      const std::map<int, PatStep> wild0 = wild;
      try { return lower_count2(f); }
      catch (const Unsupported&) { regs_used = regs0; code->resize(code0); looped = looped0; pass = pass0; wild = wild0; }
      if (f->atleast > 2) unsupported("counting loop (at least " + std::to_string(f->atleast) + " elements) where elements cannot be counted");   // (a threshold must be exact: no weaker answer)
      return lower_exists(f_exists(f->q, f->base, f->kids[0]));
    }
    {
      std::vector<std::pair<int, PatStep>> wilds;
      Atom flat;
      if (try_flat(f, wilds, &flat)) {
        bool ok = true;
        for (auto& s : flat.path) if (s.iter && !looped.count(s.q) && !pass.count(s.q)) { bool w = false; for (auto& x : wilds) if (x.first == s.q) w = true; if (!w) ok = false; }
        if (ok) {
          for (auto& w : wilds) wild[w.first] = w.second;
          int r = lower_atom(flat);
          for (auto& w : wilds) wild.erase(w.first);
          return r;
        }
      }
    }
    std::vector<FP> conj;
    conjuncts(f->kids[0], conj);
    {
      // key pinned to constants:  exists k. (k == "a" | k == "b") & body(base[k])   ==   OR_c  defined(base.c) & body(base.c)
      PatStep ps;
      std::vector<FP> rest;
      for (auto& c : conj) if (!key_constraint(c, q, &ps)) rest.push_back(c);
      if (!ps.only.empty() && ps.except.empty()) {
        FP any = f_false();
        for (const std::string& key : ps.only) {
          { bool ok = true; for (auto& kp : ps.kpreds) if (!key_pred_holds(kp, key, false)) ok = false; if (!ok) continue; }   // string tests on the pinned name
          SPath member = f->base;
          Step st; st.key = key;
          member.push_back(st);
          Atom d; d.kind = Atom::DEFINED; d.path = member;
          FP body = f_atom(d);
          for (auto& c : rest) body = f_and(body, pin_key(c, q, key));
          any = f_or(any, body);
        }
        return lower(simplify(any));
      }
    }
    for (auto& c : conj) { PatStep tmp; if (key_constraint(c, q, &tmp)) unsupported("correlated iteration over object keys"); }
    // pass-through: exists q. exists q2 in (.. q ..). body   with nothing else tied to q's element
    if (conj.size() == 1 && conj[0]->kind == FNode::EXISTS && path_has_q(conj[0]->base, q) && !uses_elem_directly(conj[0]->kids[0], q)) {
      pass.insert(q);
      int r = lower_exists(conj[0]);
      pass.erase(q);
      return r;
    }
    // ---- common-subformula cache: an EXISTS that depends on no enclosing loop variable is computed once per review
    // (derived global bit); one that depends on exactly one enclosing loop variable is computed once per element of
    // that scope (derived element bit).  Both run in a prologue, before the per-constraint formulas.
    std::vector<int> free;
    {
      std::vector<int> seen;
      collect_q(f, seen);
      for (int x : seen) if (looped.count(x)) free.push_back(x);
    }
    if (free.empty()) {
      std::string key = canon(f);
      auto it = derived_global.find(key);
      uint32_t bit;
      if (it == derived_global.end()) {
        bit = n_gbits++;
        if (bit > 0xFFFF) unsupported("too many global predicates");
        derived_global[key] = bit;
        std::vector<uint32_t> block;
        std::vector<uint32_t>* saved = cur_code;
        cur_code = &block;
        int r = lower_exists_loop(f);
        emit(finst(F_STG, r, bit & 0xFF, bit >> 8));
        release(r);
        cur_code = saved;
        prologue.push_back(block);
      } else bit = it->second;
      int r = alloc();
      emit(finst(F_LDG, r, bit & 0xFF, bit >> 8));
      return r;
    }
    if (free.size() == 1) {
      int q1 = free[0];
      uint32_t s1 = looped[q1];
      std::string key = canon(f, {q1});
      auto it = derived_elem[s1].find(key);
      uint32_t bit;
      if (it == derived_elem[s1].end()) {
        bit = scope_nbits[s1]++;
        if (bit >= ELEM_W0_BITS + 3 * 32) unsupported("too many predicates on one element scope");
        derived_elem[s1][key] = bit;
        std::vector<uint32_t> block;
        std::vector<uint32_t>* saved = cur_code;
        std::map<int, uint32_t> saved_looped = looped;
        cur_code = &block;
        looped.clear();
        looped[q1] = s1;
        int acc = alloc();
        emit(finst(F_LOOP, s1, 0, acc));
        int r = lower_exists_loop(f);
        emit(finst(F_STE, r, s1, bit));
        emit(finst(F_ENDLOOP, acc, r));
        release(r);
        release(acc);
        looped = saved_looped;
        cur_code = saved;
        prologue.push_back(block);
      } else bit = it->second;
      int r = alloc();
      emit(finst(F_LDE, r, s1, bit));
      return r;
    }
    return lower_exists_loop(f);
  }

  // EXISTS as an explicit loop over the elements of its scope
  int lower_exists_loop(const FP& f) {
    int q = f->q;
    SPath elem = f->base;
    Step st; st.iter = true; st.q = q;
    elem.push_back(st);
    // loop registration must precede pattern_of(elem)
    looped[q] = 0;
    uint32_t sc;
    try { sc = scope_for(elem); } catch (...) { looped.erase(q); throw; }
    looped[q] = sc;
    uint32_t parent = 0;
    for (int i = (int)f->base.size() - 1; i >= 0; i--)
      if (f->base[i].iter) { if (looped.count(f->base[i].q)) parent = looped[f->base[i].q] + 1; break; }
    int acc = alloc();
    emit(finst(F_LOOP, sc, parent, acc));
    int r = lower(f->kids[0]);
    emit(finst(F_ENDLOOP, acc, r));
    release(r);
    looped.erase(q);
    return acc;
  }

  // E2 as a loop with two accumulators: `once` = some element satisfied the body, `twice` = a second one did
  int lower_count2(const FP& f) {
    int q = f->q;
    {   // the children must be iterated as ELEMENTS: no key tests on q
      std::vector<FP> conj;
      conjuncts(f->kids[0], conj);
      for (auto& c : conj) { PatStep tmp; if (key_constraint(c, q, &tmp)) unsupported("counting over object keys"); }
    }
    SPath elem = f->base;
    Step st; st.iter = true; st.q = q;
    elem.push_back(st);
    looped[q] = 0;
    uint32_t sc;
    try { sc = scope_for(elem); } catch (...) { looped.erase(q); throw; }
    looped[q] = sc;
    uint32_t parent = 0;
    for (int i = (int)f->base.size() - 1; i >= 0; i--)
      if (f->base[i].iter) { if (looped.count(f->base[i].q)) parent = looped[f->base[i].q] + 1; break; }
    if (f->atleast > 2) {
      // E_k, k > 2: k accumulators c[1..k], c[j] = "at least j elements satisfied the body so far"; per element, from the top:
      // c[j] |= c[j-1] & hit, with hit = body & "the element exists" (its PRESENT bit -- top-level scopes only: a nested
      // element also has to belong to the enclosing one, which F_ENDLOOP checks and a plain register operation cannot)
      if (parent != 0) { looped.erase(q); unsupported("counting elements of a nested array"); }
      const int k = f->atleast;
      std::vector<int> c((size_t)k + 1, -1);
      for (int j = k; j >= 2; j--) { c[j] = alloc(); emit(finst(F_CONST, c[j], 0)); }
      c[1] = alloc();
      emit(finst(F_LOOP, sc, parent, c[1]));
      int r = lower(f->kids[0]);
      int pres = alloc();
      emit(finst(F_LDE, pres, sc, 0));       // bit 0 of the element word: P_PRESENT
      emit(finst(F_AND, r, r, pres));
      for (int j = k; j >= 2; j--) {
        emit(finst(F_AND, pres, c[j - 1], r));
        emit(finst(F_OR, c[j], c[j], pres));
      }
      release(pres);
      emit(finst(F_ENDLOOP, c[1], r));
      release(r);
      for (int j = 1; j < k; j++) release(c[j]);
      looped.erase(q);
      return c[k];
    }
    int twice = alloc(), once = alloc();
    emit(finst(F_CONST, twice, 0));
    emit(finst(F_LOOP, sc, parent, once));
    int r = lower(f->kids[0]);
    emit(finst(F_ENDLOOP2, once, r, twice));
    release(r);
    release(once);
    looped.erase(q);
    return twice;
  }

  int lower(const FP& f) {
    switch (f->kind) {
      case FNode::T: { int r = alloc(); emit(finst(F_CONST, r, 1)); return r; }
      case FNode::F: { int r = alloc(); emit(finst(F_CONST, r, 0)); return r; }
      case FNode::NOT: { int r = lower(f->kids[0]); emit(finst(F_NOT, r, r)); return r; }
      case FNode::AND: case FNode::OR: {
        int r = lower(f->kids[0]);
        for (size_t i = 1; i < f->kids.size(); i++) {
          int r2 = lower(f->kids[i]);
          emit(finst(f->kind == FNode::AND ? F_AND : F_OR, r, r, r2));
          release(r2);
        }
        return r;
      }
      case FNode::ATOM: return lower_atom(f->atom);
      case FNode::EXISTS: return lower_exists(f);
    }
    return -1;
  }
};

// Key iterations pinned to constants, as a formula rewrite (the lowering does the same for what is left):
//   E k in B. (k == "a" | k == "b") & body(B[k])   ==   OR_c  defined(B.c) & body(B.c)
// Run before fold_dict, so that `spec[field][_]` with field = "containers" becomes the concrete path spec.containers and
// the alternatives over it can be merged.
static FP pin_pass_impl(const FP& f) {
  switch (f->kind) {
    case FNode::NOT: return f_not(pin_pass(f->kids[0]));
    case FNode::AND: { FP r = f_true(); for (auto& k : f->kids) r = f_and(r, pin_pass(k)); return r; }
    case FNode::OR: { FP r = f_false(); for (auto& k : f->kids) r = f_or(r, pin_pass(k)); return r; }
    case FNode::EXISTS: {
      std::vector<FP> conj, rest;
      conjuncts(f->kids[0], conj);
      PatStep ps;
      for (auto& c : conj) if (!Lowerer::key_constraint(c, f->q, &ps)) rest.push_back(c);
      if (ps.only.empty() || !ps.except.empty()) return f_exists_like(*f, pin_pass(f->kids[0]));
      // (an E2 over keys pinned to constants is answered as the plain EXISTS: weaker, see lower_exists)
      FP any = f_false();
      for (const std::string& key : ps.only) {
        { bool ok = true; for (auto& kp : ps.kpreds) if (!key_pred_holds(kp, key, false)) ok = false; if (!ok) continue; }   // string tests on the pinned name
        SPath member = f->base;
        Step st; st.key = key;
        member.push_back(st);
        Atom d; d.kind = Atom::DEFINED; d.path = member;
        FP body = f_atom(d);
        for (auto& c : rest) body = f_and(body, Lowerer::pin_key(c, f->q, key));
        any = f_or(any, pin_pass(body));
      }
      return any;
    }
    default: return f;
  }
}

}  // namespace


// ---- result counting (pe.hpp Template::count_forms) ---------------------------------------------------------------------------
namespace {
bool is_prefix_of(const std::string& a, const std::string& b) { return a.size() <= b.size() && b.compare(0, a.size(), a) == 0; }
// can branches x and y be told apart by their messages alone, whatever they are bound to?  (pe.hpp, count_forms)
bool messages_differ(const Template::CountBranch& x, const Template::CountBranch& y) {
  if (x.is_const && y.is_const) return x.text != y.text;
  if (x.is_const != y.is_const) {
    const Template::CountBranch &c = x.is_const ? x : y, &h = x.is_const ? y : x;
    return h.head && !is_prefix_of(h.pre, c.text);
  }
  if (!x.head || !y.head) return false;
  if (!is_prefix_of(x.pre, y.pre) && !is_prefix_of(y.pre, x.pre)) return true;
  if (!(x.keyed && y.keyed) || x.pre != y.pre) return false;
  // same literal head, both keyed on string leaves that are pairwise different across the whole review unless they are the SAME
  // element of the same array (count_forms flags everything else): different leaves
  if (x.sig.size() < 2 || y.sig.size() < 2 || x.sig[1] != y.sig[1]) return true;
  // the same leaf pattern: possibly the same element -- then the text behind the key has to differ
  if (x.sep_tail && y.sep_tail) return x.sep != y.sep;
  if (!is_prefix_of(x.sep, y.sep) && !is_prefix_of(y.sep, x.sep)) return true;
  if (x.sig.size() != y.sig.size() || x.sig[0] != y.sig[0]) return false;
  int diff = 0;
  for (size_t i = 1; i < x.sig.size(); i++) {
    if (x.sig[i] == y.sig[i]) { if (x.sig[i] == "?") return false; continue; }   // (an operand nobody can compare)
    if (x.sig[i][0] != 'C' || y.sig[i][0] != 'C') return false;
    diff++;
  }
  return diff == 1;   // the same format and operands but for ONE constant: the same element prints two different texts
}

// ... or by their details: two constants that differ make two members whatever the messages say
bool results_differ(const Template::CountBranch& x, const Template::CountBranch& y) {
  if (x.dsig != "?" && y.dsig != "?" && x.dsig != y.dsig) return true;
  return messages_differ(x, y);
}

typedef Template::CountBranch CB;

std::string path_sig_q(const SPath& p) { std::string o; for (auto& st : p) { if (st.iter) o += "[]"; else { o += "."; o += st.key; } } return o; }

// A branch with TWO open iterations whose OUTER one walks the keys of an object and is pinned to constants by its body
// (K8sContainerLimits: spec[field][_] with field == "containers" | "initContainers") becomes one branch per constant with the
// inner iteration alone -- what pin_pass does to the violation formula.
void expand_pinned(const CB& b, std::vector<CB>* out) {
  if (b.nq != 2 || !b.body) { out->push_back(b); return; }
  // the generator's quantifiers: the outer one is b.q over b.base; the inner base is read off the key path / body: the first
  // EXISTS-free body talks about  base[q_outer][q_inner]...  -- find q_inner as the iter step behind q_outer in the key path
  int q_in = -1;
  SPath inner_base;
  const SPath* probe = b.head ? &b.key : nullptr;
  if (probe) {
    for (size_t i = 0; i + 1 < probe->size(); i++)
      if ((*probe)[i].iter && (*probe)[i].q == b.q && (*probe)[i + 1].iter) { q_in = (*probe)[i + 1].q; inner_base.assign(probe->begin(), probe->begin() + i + 1); break; }
  }
  if (getenv("GK_DEBUG_COUNTS") && (q_in < 0 || inner_base.size() != b.base.size() + 1)) {
    static int shown = 0;
    if (shown++ < 3) fprintf(stderr, "[gkgpu counts] no inner iteration: q=%d base=%s key=%s q_in=%d any=%s\n", b.q, spath_to_string(b.base).c_str(), spath_to_string(b.key).c_str(), q_in, f_to_string(b.any).substr(0, 1500).c_str());
  }
  if (q_in < 0 || inner_base.size() != b.base.size() + 1) { out->push_back(b); return; }
  for (size_t i = 0; i < b.base.size(); i++) if (b.base[i].iter) { out->push_back(b); return; }
  std::vector<FP> conj, rest;
  conjuncts(b.body, conj);
  PatStep ps;
  for (auto& c : conj) if (!Lowerer::key_constraint(c, b.q, &ps)) rest.push_back(c);
  if (getenv("GK_DEBUG_COUNTS") && (ps.only.empty() || !ps.except.empty())) {
    static int shown = 0;
    if (shown++ < 3) { fprintf(stderr, "[gkgpu counts] not pinned: q=%d only=%zu except=%zu kpreds=%zu body=%s\n", b.q, ps.only.size(), ps.except.size(), ps.kpreds.size(), f_to_string(b.body).substr(0, 700).c_str()); }
  }
  if (ps.only.empty() || !ps.except.empty()) { out->push_back(b); return; }
  auto pin_path = [&](SPath p, const std::string& key) { for (auto& s : p) if (s.iter && s.q == b.q) { s.iter = false; s.q = -1; s.key = key; } return p; };
  for (const std::string& key : ps.only) {
    { bool ok = true; for (auto& kp : ps.kpreds) if (!key_pred_holds(kp, key, false)) ok = false; if (!ok) continue; }
    CB n = b;
    n.nq = 1; n.q = q_in;
    n.base = pin_path(inner_base, key);
    FP body = f_true();
    for (auto& c : rest) body = f_and(body, Lowerer::pin_key(c, b.q, key));
    n.body = body;   // (f_and folds the constants pin_key makes of the key tests)
    if (n.body->kind == FNode::F) continue;
    n.any = f_exists(q_in, n.base, n.body);
    n.two = f_exists2(q_in, n.base, n.body);
    n.key = pin_path(b.key, key);
    for (auto& e : n.sig) if (e.size() > 1 && e[0] == 'P') { /* leaf signatures name the pinned member */ size_t at = e.find("[][]"); if (at != std::string::npos) e = e.substr(0, at) + "." + key + e.substr(at + 2); }
    // keyed as Template::compile_all decides it, now that one iteration over a top-level array is left
    n.keyed = n.head && n.key.size() > n.base.size() + 1 && n.key[n.base.size()].iter && n.key[n.base.size()].q == q_in && (!n.sep.empty() || n.sep_tail);
    for (size_t i = n.base.size() + 1; n.keyed && i < n.key.size(); i++) if (n.key[i].iter) n.keyed = false;
    out->push_back(std::move(n));
  }
}

// Branches that print the SAME message for the same binding (same format, every operand comparable and equal, same iteration)
// are one branch: the set keeps one member however many rule bodies / unrolled alternatives produce it.
// OR of conjunctions with what they have in common factored out:  (c & a1) | (c & a2) ..  ==  c & (a1 | a2 ..)  -- the unrolled
// alternatives of one message (K8sContainerLimits: hundreds of unit-suffix cases) then differ in sub-formulas about ONE leaf,
// which fold into one dictionary expression instead of one per alternative
FP or_factored(const std::vector<FP>& alts) {
  if (alts.size() == 1) return alts[0];
  std::vector<std::vector<FP>> cs(alts.size());
  std::map<std::string, size_t> seen;
  for (size_t i = 0; i < alts.size(); i++) {
    conjuncts(alts[i], cs[i]);
    std::set<std::string> mine;
    for (auto& c : cs[i]) if (mine.insert(f_to_string(c)).second) seen[f_to_string(c)]++;
  }
  FP common = f_true();
  std::set<std::string> is_common;
  for (auto& c : cs[0]) { const std::string t = f_to_string(c); if (seen[t] == alts.size() && is_common.insert(t).second) common = f_and(common, c); }
  FP any = f_false();
  for (auto& v : cs) {
    FP rest = f_true();
    for (auto& c : v) if (!is_common.count(f_to_string(c))) rest = f_and(rest, c);
    any = f_or(any, rest);
  }
  return f_and(common, any);
}

void merge_equal(std::vector<CB>* br) {
  std::vector<CB> out;
  std::vector<std::vector<FP>> alts;
  std::map<std::string, size_t> at;
  for (CB& b : *br) {
    std::string key;
    bool mergeable = b.nq <= 1 && (b.is_const || (b.head && !b.sig.empty()));
    if (b.is_const) key = "C|" + b.text;
    else if (mergeable) { for (auto& e : b.sig) { if (e == "?") mergeable = false; key += e; key.push_back('\x1f'); } }
    // ... and the same `details`: members that differ in details only stay apart in the set (one result each), so only branches
    // whose details are the same CONSTANT (or absent) are one branch; details that depend on the review are never merged
    if (b.dsig == "?") mergeable = false;
    if (mergeable) key += "|" + std::to_string(b.nq) + "|" + path_sig_q(b.base) + "|D" + b.dsig;
    auto it = mergeable ? at.find(key) : at.end();
    if (!mergeable || it == at.end()) { if (mergeable) at[key] = out.size(); alts.push_back({b.nq == 0 ? b.any : b.body}); out.push_back(std::move(b)); continue; }
    CB& a = out[it->second];
    if (a.nq == 0) { alts[it->second].push_back(b.any); continue; }
    std::map<int, int> m; m[b.q] = a.q;
    alts[it->second].push_back(rename_f(b.body, m));
    a.keyed = a.keyed && b.keyed;
  }
  for (size_t i = 0; i < out.size(); i++) {
    if (alts[i].size() < 2) continue;
    CB& a = out[i];
    if (a.nq == 0) { a.any = or_factored(alts[i]); continue; }
    a.body = or_factored(alts[i]);
    a.any = f_exists(a.q, a.base, a.body);
    a.two = f_exists2(a.q, a.base, a.body);
  }
  br->swap(out);
}
}  // namespace

Template::CountForms Template::count_forms(const CountInfo& ci, int kmax) {
  CountForms cf;
  cf.flag = f_false();
  std::vector<CB> br;
  const auto t0 = std::chrono::steady_clock::now();
  for (const CB& b : ci.br) expand_pinned(b, &br);
  const auto t1 = std::chrono::steady_clock::now();
  merge_equal(&br);
  if (getenv("GK_DEBUG_LOAD")) fprintf(stderr, "[gkgpu load]   count_forms: pinning %.3f s, merging %.3f s (%zu -> %zu branches)\n", std::chrono::duration<double>(t1 - t0).count(),
                                       std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count(), ci.br.size(), br.size());
  if (getenv("GK_DEBUG_COUNTS")) {
    fprintf(stderr, "[gkgpu counts] %zu branches -> %zu after pinning and merging\n", ci.br.size(), br.size());
    for (auto& b : br) fprintf(stderr, "   + nq=%d keyed=%d const=%d pre=[%s] sep=[%s] tail=%d key=%s sig=%s\n", b.nq, (int)b.keyed, (int)b.is_const, b.pre.c_str(), b.sep.c_str(), (int)b.sep_tail,
                               spath_to_string(b.key).c_str(), [&] { std::string o; for (auto& e : b.sig) o += e + " | "; return o; }().c_str());
  }
  Atom dup; dup.kind = Atom::DEFINED;
  { Step st; st.key = "$dup"; dup.path.push_back(st); }
  bool any_keyed = false;
  cf.viol = f_false();
  for (const CB& b : br) cf.viol = f_or(cf.viol, b.any);
  for (const CB& b : br) {
    if (b.nq == 0) cf.firsts.push_back((uint32_t)cf.rows.size());   // (the flag holds the body of every branch with an iteration)
    cf.rows.push_back(b.any);
    if (b.nq == 0) continue;
    if (!b.keyed) { cf.flag = f_or(cf.flag, b.two); continue; }
    any_keyed = true;
    cf.keys.push_back(b.key);
    // the key of every firing element is a string without the first character of the literal behind it
    Atom ty; ty.kind = Atom::TYPE; ty.path = b.key; ty.mask = 1u << T_STRING;
    FP good = f_atom(ty);
    if (!b.sep.empty()) { Atom c; c.kind = Atom::STR_CONTAINS; c.path = b.key; c.k = Value::string(b.sep.substr(0, 1)); good = f_and(good, f_not(f_atom(c))); }
    cf.flag = f_or(cf.flag, f_exists(b.q, b.base, f_and(b.body, f_not(good))));
    for (int k = 2; k <= kmax; k++) cf.rows.push_back(f_exists_k(b.q, b.base, b.body, k));
    cf.flag = f_or(cf.flag, f_exists_k(b.q, b.base, b.body, kmax + 1));   // more firing elements than thresholds
  }
  if (any_keyed) cf.flag = f_or(cf.flag, f_atom(dup));
  if (br.size() > 24) return cf;   // (too many alternatives for the pairwise terms: ok stays false, compile_multi's answer serves)
  for (size_t i = 0; i < br.size(); i++)
    for (size_t j = i + 1; j < br.size(); j++)
      if (!results_differ(br[i], br[j])) cf.flag = f_or(cf.flag, f_and(br[i].any, br[j].any));
  cf.ok = true;
  return cf;
}

PrepMemoScope::PrepMemoScope() { prev_ = g_prep_memo; mine_ = new PrepMemo(); g_prep_memo = (PrepMemo*)mine_; }
PrepMemoScope::~PrepMemoScope() { g_prep_memo = (PrepMemo*)prev_; delete (PrepMemo*)mine_; }

std::shared_ptr<const PreparedConstraint> prepare_constraint(const FP& violation, const MatchFormulas& mf) {
  auto pc = std::make_shared<PreparedConstraint>();
  pc->viol = simplify(fold_dict(simplify(pin_pass(simplify(violation)))));
  g_fold_match_only = true;
  try {
    pc->match = simplify(fold_dict(simplify(mf.match)));
    pc->error = simplify(fold_dict(simplify(mf.error)));
  } catch (...) { g_fold_match_only = false; throw; }
  g_fold_match_only = false;
  pc->viol_key = canon(pc->viol);
  pc->match_key = canon(pc->match) + "##" + canon(pc->error);
  return pc;
}

HostPlan PlanBuilder::build(const PlanCaps& caps) {
  Lowerer L;
  L.dict = dict_;
  L.reg = reg_;
  L.frozen = frozen_;
  L.counting = counting_;
  L.caps = caps;
  std::map<std::string, uint32_t> viol_ids, match_ids;
  std::vector<FP> viols, matches, errs;
  for (auto& c : cons_) {
    const FP &v = c->viol, &m = c->match, &e = c->error;
    const std::string &vk = c->viol_key, &mk = c->match_key;
    ConstraintSlot slot;
    auto it = viol_ids.find(vk);
    if (it == viol_ids.end()) { slot.viol = (uint16_t)viols.size(); viol_ids[vk] = slot.viol; viols.push_back(v); } else slot.viol = (uint16_t)it->second;
    auto jt = match_ids.find(mk);
    if (jt == match_ids.end()) { slot.match = (uint16_t)matches.size(); match_ids[mk] = slot.match; matches.push_back(m); errs.push_back(e); } else slot.match = (uint16_t)jt->second;
    L.plan.slots.push_back(slot);
  }
  if (viols.size() > GK_MAX_VIOL || matches.size() > GK_MAX_RES)
    throw Unsupported("more than 256 distinct violation or 64 distinct match formulas in one plan (split the constraint set)");
  std::vector<uint32_t> main_ends;
  for (size_t i = 0; i < viols.size(); i++) { int r = L.lower(viols[i]); L.emit(finst(F_RES, r, 0, (uint32_t)i)); L.release(r); main_ends.push_back((uint32_t)L.plan.code.size()); }
  for (size_t i = 0; i < matches.size(); i++) {
    int r = L.lower(matches[i]); L.emit(finst(F_RES, r, 1, (uint32_t)i)); L.release(r);
    main_ends.push_back((uint32_t)L.plan.code.size());
    r = L.lower(errs[i]); L.emit(finst(F_RES, r, 2, (uint32_t)i)); L.release(r);
    main_ends.push_back((uint32_t)L.plan.code.size());
  }
  L.emit(finst(F_END));
  // Guards: `x[_]` is lowered as iteration over ARRAY elements (rows carry dense element ordinals); Rego also iterates
  // the values of an OBJECT.  A review that holds a non-empty object where the plan iterates elements is therefore
  // refused, never guessed: the container patterns go to the registry, the flattener flags such reviews (RF_REFUSE) and
  // the kernels report them in too_big -- the caller fails closed.  No rows, no device predicate.
  if (reg_) {
    std::set<std::string> seen;
    std::vector<Pattern> local_guards;
    for (const Pattern& pat : L.plan.pred_patterns)
      for (size_t i = 0; i < pat.size(); i++) {
        if (!pat[i].any || !pat[i].elems_only) continue;
        Pattern prefix(pat.begin(), pat.begin() + i);
        if (!seen.insert(pattern_to_string(prefix)).second || reg_->add_guard(prefix, !frozen_)) continue;
        // frozen plan, guard not registered: a LOCAL guard -- a predicate of this plan that sets the review's overflow bit when
        // the container is an object; the big variant sets it again, so the review ends in too_big FOR THIS PLAN ONLY (the totals
        // plans: such a review is rendered; the violation formulas, which reach the members through wildcard predicates, are
        // not disturbed)
        local_guards.push_back(prefix);
      }
    for (const Pattern& prefix : local_guards) {
      Pred g{};
      g.op = P_TYPE; g.dst = D_GLOBAL; g.bit = 0; g.ctype = (uint8_t)(1u << T_OBJECT);
      L.plan.preds.push_back(g);
      L.plan.pred_patterns.push_back(prefix);
    }
    // rows that are compared with other review values need VALUE IDS: the flattener assigns them on the registered paths
    for (size_t i = 0; i < L.plan.preds.size(); i++)
      if (L.plan.preds[i].op == P_STORE && !reg_->add_value(L.plan.pred_patterns[i], !frozen_)) throw Unsupported("unsupported on the device plan: needs value ids no loaded constraint registered");
  }
  // ELEMENT CARRIERS (plan.hpp T_ABSENT, round 6).  A scope's element marker is read from the rows of ONE member of the element instead of
  // the element's own rows, when the registry names a carrier for the element pattern -- or this plan can offer one: a member of the
  // element it has a predicate on anyway (`name`, in every in-tree template that iterates containers / volumes / volumeMounts: half
  // of configs[2]'s rows were such pairs).  The flattener guarantees one row at the carrier's path per element.  GK_CARRIERS=0: off.
  static const bool carriers_on = !(getenv("GK_CARRIERS") && atoi(getenv("GK_CARRIERS")) == 0);
  if (reg_ && carriers_on) {
    for (size_t s = 0; s < L.scope_patterns.size(); s++) {
      const size_t pi = s < L.present_pred.size() ? L.present_pred[s] : SIZE_MAX;
      if (pi == SIZE_MAX) continue;
      const Pattern elem = L.scope_patterns[s];
      const std::string prefix = pattern_to_string(elem);
      std::string offer;
      for (size_t q = 0; q < L.plan.pred_patterns.size(); q++) {
        const Pattern& pat = L.plan.pred_patterns[q];
        if (q == pi || pat.size() != elem.size() + 1) continue;
        const PatStep& last = pat.back();
        if (last.any || last.key.empty() || last.key[0] == '$') continue;
        if (pattern_to_string(Pattern(pat.begin(), pat.end() - 1)) != prefix) continue;
        if (last.key == "name") { offer = "name"; break; }
        if (offer.empty() || last.key < offer) offer = last.key;
      }
      std::string chosen;
      if (!reg_->add_carrier(elem, offer, !frozen_, &chosen)) continue;
      PatStep step; step.key = chosen;
      L.plan.pred_patterns[pi].push_back(step);
    }
  }
  HostPlan& p = L.plan;
  {   // derived-bit prologue blocks first (inner blocks were completed, hence appended, before outer ones)
    std::vector<uint32_t> code;
    for (auto& b : L.prologue) { code.insert(code.end(), b.begin(), b.end()); p.seg_ends.push_back((uint32_t)code.size()); }
    for (uint32_t e : main_ends) p.seg_ends.push_back((uint32_t)code.size() + e);
    code.insert(code.end(), p.code.begin(), p.code.end());
    p.code.swap(code);
  }
  p.n_viol = (uint32_t)viols.size();
  p.n_match = (uint32_t)matches.size();
  // accumulator layout
  uint32_t gwords = (L.n_gbits + 31) / 32;
  uint32_t off = gwords;
  for (size_t s = 0; s < L.scope_patterns.size(); s++) {
    Scope sc{};
    uint32_t nb = L.scope_nbits[s];
    sc.wpe = (uint8_t)(nb <= ELEM_W0_BITS ? 1 : 1 + (nb - ELEM_W0_BITS + 31) / 32);
    sc.cap = L.scope_level[s] >= (int)GK_LEVEL_ROOT ? 1 : s < caps.scope_cap.size() && caps.scope_cap[s] ? caps.scope_cap[s] : caps.level_cap[L.scope_level[s]];
    sc.nvals = (uint8_t)L.val_slots[s].size();
    sc.count_off = off++;
    sc.word_off = off;
    off += (uint32_t)sc.cap * sc.wpe;
    // value slots hold VALUE IDS (plan.hpp): one word per slot and element -- or, for the usual scope with one joined value
    // and a handful of element bits, bits [23:8] of the element word itself (no accumulator words of their own)
    if (sc.nvals == 1 && nb <= ELEM_PACK_BITS) sc.val_off = GK_VAL_PACKED;
    else { sc.val_off = off; off += (uint32_t)sc.cap * val_stride(sc.nvals); }
    p.scopes.push_back(sc);
  }
  p.dims.n_preds = (uint32_t)p.preds.size();
  p.dims.n_scopes = (uint32_t)p.scopes.size();
  p.dims.n_code = (uint32_t)p.code.size();
  p.dims.n_constraints = (uint32_t)p.slots.size();
  p.dims.n_gwords = gwords;
  p.dims.acc_words = off;
  while (p.cheap.size() & 3) p.cheap.push_back(0);
  if (p.cheap.empty()) p.cheap.resize(4, 0);
  p.dims.const_bytes = (uint32_t)p.cheap.size();
  p.dims.n_viol = p.n_viol; p.dims.n_match = p.n_match;
  p.resolve_paths(*dict_);
  return p;
}

void HostPlan::resolve_paths(const PathDict& dict) {
  uint32_t n = dict.size();
  std::vector<PathDict::Info> infos(n);
  std::vector<std::vector<uint32_t>> children(n);
  for (uint32_t i = 0; i < n; i++) { infos[i] = dict.info(i); if (i) children[infos[i].parent].push_back(i); }
  std::vector<std::vector<uint32_t>> per_path(n);
  for (uint32_t pi = 0; pi < preds.size(); pi++) {
    const Pattern& pat = pred_patterns[pi];
    std::vector<uint32_t> cur{0};
    for (const PatStep& st : pat) {
      std::vector<uint32_t> nxt;
      for (uint32_t id : cur)
        for (uint32_t ch : children[id]) {
          const PathDict::Info& in = infos[ch];
          if (!st.any) { if (!in.is_elem && in.key == st.key) nxt.push_back(ch); continue; }
          { bool kp_ok = true; for (auto& kp : st.kpreds) if (!key_pred_holds(kp, in.key, in.is_elem)) kp_ok = false; if (!kp_ok) continue; }
          // array elements under a key iteration: the key is a numeric index, which differs from every string in an
          // `except` list and equals none in an `only` list
          if (in.is_elem) { if (st.only.empty()) nxt.push_back(ch); continue; }
          if (st.elems_only) continue;
          if (in.key == "$d" || in.key == "$c") continue;   // the flattener's dictionary rows under a leaf are not members of the document
          if (!st.only.empty() && std::find(st.only.begin(), st.only.end(), in.key) == st.only.end()) continue;
          if (std::find(st.except.begin(), st.except.end(), in.key) != st.except.end()) continue;
          nxt.push_back(ch);
        }
      cur.swap(nxt);
      if (cur.empty()) break;
    }
    for (uint32_t id : cur) per_path[id].push_back(pi);
  }
  ptab.assign(n, 0);
  path_preds.clear();
  path_preds.push_back(Pred{});   // keep entry 0 unused so that ptab == 0 means "none"
  for (uint32_t i = 0; i < n; i++) {
    if (per_path[i].empty()) continue;
    if (per_path[i].size() > 255) throw Unsupported("more than 255 predicates on one path");
    ptab[i] = ((uint32_t)path_preds.size() << 8) | (uint32_t)per_path[i].size();
    for (uint32_t pi : per_path[i]) path_preds.push_back(preds[pi]);
  }
  dict_size = n;
  dims.n_paths = n;
}

}  // namespace gk
