// Leaf-local expressions ("dictionary predicates").
//
// Some template logic is a pure function of ONE review leaf: the quantity parsing of K8sContainerLimits
// (demo/agilebank/templates/k8scontainterlimits_template.yaml: canonify_cpu / canonify_mem -- replace, substring, to_number,
// re_match, arithmetic, a dozen function bodies) decides "is this cpu limit above 200m" from the limit string alone.  The
// device has no string arithmetic -- and needs none: the partial evaluator records such a computation as an expression tree
// over the leaf (DExpr), the lowering folds every boolean sub-formula that talks about a single leaf into ONE expression,
// and the FLATTENER evaluates it with the concrete builtins once per distinct value of that column (memoised), shipping
// the answers as a bit mask in a synthetic integer row next to the leaf (path + "$d").  On the device the predicate is a
// bit test (P_BITS).  Exact by construction: the bits are computed by the same evaluator that renders the messages.
#pragma once
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "builtins.hpp"
#include "value.hpp"

namespace gk {

struct Atom;   // pe.hpp

struct DExpr;
typedef std::shared_ptr<const DExpr> DX;
struct DExpr {
  enum Kind { LEAF, CONST, CALL, ARITH, CMP, DEFINED, TRUTHY, AND, OR, NOT, TYPE_MASK } kind = LEAF;
  Value c;                  // CONST
  std::string name;         // CALL builtin name / ARITH operator
  int cmp = 0;              // CMP: CmpOp
  uint32_t mask = 0;        // TYPE_MASK: bit per RowType of the leaf-derived value
  std::vector<DX> args;
  std::string text;         // canonical text (dx_to_string), composed ONCE when the node is made from its children's: lowering keys
                            // formulas by it, and re-deriving it per lookup made a K8sContainerLimits constraint take 3 s to add
};

// DEEP expressions (round 3): a template's own helper function applied to ONE sub-document of the review -- closed (it reads
// nothing but its arguments) but beyond what a formula over rows expresses (K8sUniqueServiceSelector's flatten_selector:
// concat(",", sort([concat(":", [k, v]) | v = obj.spec.selector[k]]))).  The partial evaluator records the call as
// CALL "$u:<id>" over the leaf; the flattener hands such an expression the leaf's REAL value (containers included: parsed
// from the text span) and the registered closure runs the concrete evaluator on it -- the evaluator that renders the messages.
// CALL "$wrap" rebuilds the argument from the narrowest sub-document the function looks at: $wrap(leaf, ["spec", "selector"])
// = {"spec": {"selector": leaf}}.
typedef std::function<Value(const ValueVec&)> DxUserFn;
struct DxUserFns { std::shared_mutex mu; std::map<std::string, DxUserFn> fns; };
inline DxUserFns& dx_user_fns() { static DxUserFns r; return r; }
inline bool dx_is_user(const std::string& name) { return name.compare(0, 3, "$u:") == 0; }
inline void dx_register_user(const std::string& name, DxUserFn fn) { DxUserFns& r = dx_user_fns(); std::unique_lock<std::shared_mutex> l(r.mu); r.fns[name] = std::move(fn); }
inline void dx_unregister_user(const std::string& name) { DxUserFns& r = dx_user_fns(); std::unique_lock<std::shared_mutex> l(r.mu); r.fns.erase(name); }
inline Value dx_call_user(const std::string& name, const ValueVec& args) {
  DxUserFn fn;
  { DxUserFns& r = dx_user_fns(); std::shared_lock<std::shared_mutex> l(r.mu); auto it = r.fns.find(name); if (it == r.fns.end()) return Value(); fn = it->second; }
  return fn(args);
}

inline std::string dx_compose_text(const DExpr* e) {
  static const char* cmpn[] = {"==", "!=", "<", "<=", ">", ">="};
  switch (e->kind) {
    case DExpr::LEAF: return "$";
    case DExpr::CONST: return to_term_string(e->c);
    case DExpr::CALL: { std::string o = e->name + "("; for (size_t i = 0; i < e->args.size(); i++) { if (i) o += ","; o += e->args[i]->text; } return o + ")"; }
    case DExpr::ARITH: return "(" + e->args[0]->text + e->name + e->args[1]->text + ")";
    case DExpr::CMP: return "(" + e->args[0]->text + cmpn[e->cmp] + e->args[1]->text + ")";
    case DExpr::DEFINED: return "def(" + e->args[0]->text + ")";
    case DExpr::TRUTHY: return "truthy(" + e->args[0]->text + ")";
    case DExpr::NOT: return "!(" + e->args[0]->text + ")";
    case DExpr::TYPE_MASK: return "type(" + e->args[0]->text + ")&" + std::to_string(e->mask);
    case DExpr::AND: case DExpr::OR: {
      std::string o = "(";
      for (size_t i = 0; i < e->args.size(); i++) { if (i) o += e->kind == DExpr::AND ? " & " : " | "; o += e->args[i]->text; }
      return o + ")";
    }
  }
  return "?";
}

inline DX dx_leaf() { static thread_local DX l = [] { DExpr e; e.text = dx_compose_text(&e); return std::make_shared<const DExpr>(e); }(); return l; }
inline DX dx_const(const Value& v) { DExpr e; e.kind = DExpr::CONST; e.c = v; e.text = dx_compose_text(&e); return std::make_shared<const DExpr>(std::move(e)); }
inline DX dx_node(DExpr::Kind k, std::vector<DX> args, const std::string& name = "", int cmp = 0, uint32_t mask = 0) {
  DExpr e; e.kind = k; e.args = std::move(args); e.name = name; e.cmp = cmp; e.mask = mask;
  e.text = dx_compose_text(&e);
  return std::make_shared<const DExpr>(std::move(e));
}

inline const std::string& dx_to_string(const DX& e) { return e->text; }

inline bool dx_cmp_holds(int c, int op) {
  switch (op) { case 0: return c == 0; case 1: return c != 0; case 2: return c < 0; case 3: return c <= 0; case 4: return c > 0; default: return c >= 0; }
}

// value of the expression for a concrete leaf; Undefined propagates (a builtin error / undefined operand)
inline Value dx_eval(const DX& e, const Value& leaf) {
  switch (e->kind) {
    case DExpr::LEAF: return leaf;
    case DExpr::CONST: return e->c;
    case DExpr::CALL: {
      ValueVec av;
      for (auto& a : e->args) { Value v = dx_eval(a, leaf); if (!v.defined()) return Value(); av.push_back(v); }
      if (dx_is_user(e->name)) return dx_call_user(e->name, av);
      if (e->name == "$wrap") {   // the leaf under a constant key path: {"k0": {"k1": leaf}}
        if (av.size() != 2 || !av[1].is_array()) return Value();
        Value v = av[0];
        for (size_t i = av[1].size(); i-- > 0;) { ValuePairs p; p.emplace_back(av[1].items()[i], v); v = Value::object(std::move(p)); }
        return v;
      }
      if (e->name == "$index") {   // element of an array (a split component); negative index: from the end; out of range: undefined
        if (av.size() != 2 || !av[0].is_array() || !av[1].is_number() || !av[1].is_int) return Value();
        const long long n = (long long)av[0].size();
        long long i = (long long)av[1].i;
        if (i < 0) i += n;
        if (i < 0 || i >= n) return Value();
        return av[0].items()[(size_t)i];
      }
      return call_builtin(e->name, av);
    }
    case DExpr::ARITH: {
      Value a = dx_eval(e->args[0], leaf), b = dx_eval(e->args[1], leaf);
      if (!a.defined() || !b.defined()) return Value();
      return rego_arith(e->name, a, b);
    }
    case DExpr::CMP: {
      Value a = dx_eval(e->args[0], leaf), b = dx_eval(e->args[1], leaf);
      if (!a.defined() || !b.defined()) return Value::boolean(false);
      return Value::boolean(dx_cmp_holds(compare(a, b), e->cmp));
    }
    case DExpr::DEFINED: return Value::boolean(dx_eval(e->args[0], leaf).defined());
    case DExpr::TRUTHY: { Value v = dx_eval(e->args[0], leaf); return Value::boolean(v.defined() && !(v.is_bool() && !v.b)); }
    case DExpr::NOT: { Value v = dx_eval(e->args[0], leaf); return Value::boolean(!(v.is_bool() && v.b)); }
    case DExpr::TYPE_MASK: {
      Value v = dx_eval(e->args[0], leaf);
      if (!v.defined()) return Value::boolean(false);
      // RowType bits (plan.hpp): null 0, bool 1, int 2, float 3, string 4, object 5, array 6
      uint32_t t = v.is_null() ? 0 : v.is_bool() ? 1 : v.is_number() ? ((v.is_int && v.i >= (i128)INT64_MIN && v.i <= (i128)INT64_MAX) ? 2 : 3) : v.is_string() ? 4 : v.is_object() ? 5 : 6;
      uint32_t m = e->mask;
      if (m & ((1u << 2) | (1u << 3))) m |= (1u << 2) | (1u << 3);   // "number" masks carry both numeric row types
      return Value::boolean(((1u << t) & m) != 0);
    }
    case DExpr::AND: { for (auto& a : e->args) { Value v = dx_eval(a, leaf); if (!(v.is_bool() && v.b)) return Value::boolean(false); } return Value::boolean(true); }
    case DExpr::OR: { for (auto& a : e->args) { Value v = dx_eval(a, leaf); if (v.is_bool() && v.b) return Value::boolean(true); } return Value::boolean(false); }
  }
  return Value();
}
// does the expression look INSIDE a container leaf (a deep expression)?  Then the flattener must hand it the real value.
inline bool dx_deep(const DX& e) {
  if (e->kind == DExpr::CALL && dx_is_user(e->name)) return true;
  for (auto& a : e->args) if (dx_deep(a)) return true;
  return false;
}
inline bool dx_true(const DX& e, const Value& leaf) { Value v = dx_eval(e, leaf); return v.is_bool() && v.b; }

// dx_true for a STRING leaf given as bytes, for the shapes the lowering makes of string tests (comparisons with string constants,
// startswith / endswith / contains, type tests, and / or / not over them): no Value, no allocation.  *ok = false: the expression holds
// something else -- the caller asks dx_true.  (The match facts of a review -- names are as good as unique -- cannot go through a memo.)
// split($, "<sep>") with a non-empty constant separator over the leaf itself (Go strings.Split: the components between the occurrences)
inline bool dx_is_leaf_split(const DExpr& c, const std::string** sep) {
  if (c.kind != DExpr::CALL || c.name != "split" || c.args.size() != 2 || c.args[0]->kind != DExpr::LEAF || c.args[1]->kind != DExpr::CONST || !c.args[1]->c.is_string()) return false;
  *sep = &c.args[1]->c.str();
  return !(*sep)->empty();
}
// The components of split(<leaf bytes>, sep), worked out ONCE per leaf value and separator: the dictionary expressions of a policy set ask
// for them dozens of times per value (every banned-tag constraint its own `$index(split($, ":"), -1) == "..."`).  One slot per thread:
// the entries of a leaf are evaluated one after the other on the same bytes.
struct DxSplitCache { const char* s = nullptr; size_t n = 0; std::string sep; std::vector<uint32_t> cut; /* component k = [cut[2k], cut[2k+1]) */ };
inline const DxSplitCache& dx_split_of(const char* s, size_t n, const std::string& sep) {
  static thread_local DxSplitCache c;
  if (c.s == s && c.n == n && c.sep == sep && !c.cut.empty()) return c;
  c.s = s; c.n = n; c.sep = sep; c.cut.clear();
  uint32_t start = 0;
  for (size_t i = 0; i + sep.size() <= n;) {
    if (memcmp(s + i, sep.data(), sep.size()) == 0) { c.cut.push_back(start); c.cut.push_back((uint32_t)i); i += sep.size(); start = (uint32_t)i; } else i++;
  }
  c.cut.push_back(start); c.cut.push_back((uint32_t)n);
  return c;
}
inline void dx_split_forget() { const_cast<DxSplitCache&>(dx_split_of(nullptr, 0, std::string(1, '\0'))); }   // (the bytes behind a cached pointer may change: called per leaf value)
inline size_t dx_split_count(const char* s, size_t n, const std::string& sep) { return dx_split_of(s, n, sep).cut.size() / 2; }
// component `idx` (negative: from the end) of the split; false: out of range
inline bool dx_split_component(const char* s, size_t n, const std::string& sep, long long idx, const char** cp, size_t* cn) {
  const DxSplitCache& c = dx_split_of(s, n, sep);
  const long long cnt = (long long)(c.cut.size() / 2);
  if (idx < 0) idx += cnt;
  if (idx < 0 || idx >= cnt) return false;
  *cp = s + c.cut[2 * (size_t)idx]; *cn = c.cut[2 * (size_t)idx + 1] - c.cut[2 * (size_t)idx];
  return true;
}
inline bool dx_true_str(const DX& e, const char* s, size_t n, bool* ok) {
  switch (e->kind) {
    case DExpr::CONST: if (e->c.is_bool()) return e->c.b; *ok = false; return false;
    case DExpr::AND: { for (auto& a : e->args) { if (!dx_true_str(a, s, n, ok)) return false; if (!*ok) return false; } return true; }
    case DExpr::OR: { for (auto& a : e->args) { if (dx_true_str(a, s, n, ok)) return true; if (!*ok) return false; } return false; }
    case DExpr::NOT: { const bool v = dx_true_str(e->args[0], s, n, ok); return !v; }
    case DExpr::DEFINED: if (e->args.size() == 1 && e->args[0]->kind == DExpr::LEAF) return true; *ok = false; return false;
    case DExpr::TYPE_MASK: if (e->args.size() == 1 && e->args[0]->kind == DExpr::LEAF) return ((e->mask >> 4) & 1u) != 0; *ok = false; return false;
    case DExpr::CMP: {
      if (e->args.size() == 2 && e->args[0]->kind == DExpr::CALL && e->args[1]->kind == DExpr::CONST) {
        // the shapes `split(image, ":")` makes (the banned-tag templates of the library): count(split($, sep)) <op> int and
        // $index(split($, sep), i) <op> "string" -- on the bytes (round 6: with unique image tags the per-value memo never hits and the
        // generic evaluator, a Value tree per call, was 60 % of the ingest of the 200-template corpus)
        const DExpr& L = *e->args[0];
        const Value& K = e->args[1]->c;
        const std::string* sep = nullptr;
        if (L.name == "count" && L.args.size() == 1 && dx_is_leaf_split(*L.args[0], &sep) && K.is_number() && K.is_int) {
          const i128 c = (i128)dx_split_count(s, n, *sep);
          return dx_cmp_holds(c < K.i ? -1 : c > K.i ? 1 : 0, e->cmp);
        }
        if (L.name == "$index" && L.args.size() == 2 && dx_is_leaf_split(*L.args[0], &sep) && L.args[1]->kind == DExpr::CONST && L.args[1]->c.is_number() && L.args[1]->c.is_int && K.is_string()) {
          const char* cp = nullptr; size_t cn = 0;
          if (!dx_split_component(s, n, *sep, (long long)L.args[1]->c.i, &cp, &cn)) return false;   // (an undefined operand: the comparison is false)
          const std::string& k = K.str();
          const size_t m = cn < k.size() ? cn : k.size();
          int c = m ? memcmp(cp, k.data(), m) : 0;
          if (c == 0) c = cn < k.size() ? -1 : cn > k.size() ? 1 : 0;
          return dx_cmp_holds(c, e->cmp);
        }
        *ok = false; return false;
      }
      if (e->args.size() != 2 || e->args[0]->kind != DExpr::LEAF || e->args[1]->kind != DExpr::CONST || !e->args[1]->c.is_string()) { *ok = false; return false; }
      const std::string& k = e->args[1]->c.str();
      if (e->cmp == 0 || e->cmp == 1) { const bool eq = k.size() == n && memcmp(k.data(), s, n) == 0; return e->cmp == 0 ? eq : !eq; }
      const size_t m = n < k.size() ? n : k.size();
      int c = m ? memcmp(s, k.data(), m) : 0;
      if (c == 0) c = n < k.size() ? -1 : n > k.size() ? 1 : 0;
      return dx_cmp_holds(c, e->cmp);
    }
    case DExpr::TRUTHY: {
      if (e->args.size() != 1) { *ok = false; return false; }
      const DExpr& c = *e->args[0];
      if (c.kind == DExpr::LEAF) return true;   // a string is not `false`
      if (c.kind == DExpr::CALL && (c.name == "re_match" || c.name == "regex.match") && c.args.size() == 2 && c.args[0]->kind == DExpr::CONST && c.args[0]->c.is_string() && c.args[1]->kind == DExpr::LEAF) {
        bool valid = true;
        const bool hit = builtin_regex_search(c.args[0]->c.str(), s, n, &valid);
        return valid && hit;   // (an invalid pattern: the builtin is undefined, hence not truthy)
      }
      if (c.kind != DExpr::CALL || c.args.size() != 2 || c.args[0]->kind != DExpr::LEAF || c.args[1]->kind != DExpr::CONST || !c.args[1]->c.is_string()) { *ok = false; return false; }
      const std::string& k = c.args[1]->c.str();
      if (c.name == "startswith") return k.size() <= n && memcmp(s, k.data(), k.size()) == 0;
      if (c.name == "endswith") return k.size() <= n && memcmp(s + n - k.size(), k.data(), k.size()) == 0;
      if (c.name == "contains") {
        if (k.empty()) return true;
        if (k.size() > n) return false;
        for (size_t i = 0; i + k.size() <= n; i++) if (s[i] == k[0] && memcmp(s + i, k.data(), k.size()) == 0) return true;
        return false;
      }
      *ok = false; return false;
    }
    default: *ok = false; return false;
  }
}

// ---- the dictionary expressions of ONE leaf compiled for string values (round 6).  dx_true_str walks every entry's tree on its own: a
// leaf with 80 expressions (containers[].image of the 200-template corpus) re-tests the same prefixes and re-reads the same split
// component dozens of times per value, comparing builtin NAMES on the way -- 25 us per value once the values are unique and no memo
// hits.  DxStrProg flattens the entries into (a) the DISTINCT primitive tests -- each evaluated at most once per value, the split
// components shared -- and (b) a postfix boolean program per entry over their answers.  An entry with a node the compiler does not
// know stays generic (`generic[i]`: the caller asks dx_true_str / dx_true for it).
struct DxStrProg {
  enum PKind : uint8_t { P_TRUE, P_FALSE, P_ISSTR, P_EQ, P_CMP, P_PREFIX, P_SUFFIX, P_CONTAINS, P_SPLIT_COUNT, P_SPLIT_IDX, P_REGEX };
  struct Prim { PKind kind; int cmp = 0; std::string k, sep; long long idx = 0; i128 num = 0; bool yes = false; };
  enum : uint8_t { O_PRIM, O_AND, O_OR, O_NOT };
  struct Op { uint8_t op; uint32_t a; };   // O_PRIM: primitive id; O_AND / O_OR: operand count
  std::vector<Prim> prims;
  std::map<std::string, uint32_t> prim_ids;
  std::vector<std::vector<Op>> progs;      // per entry
  std::vector<uint8_t> generic;            // per entry: 1 = not compiled
  mutable std::vector<int8_t> val;         // per primitive: -1 unknown, 0 / 1 (scratch of one evaluation)
  mutable std::vector<uint8_t> stack;

  uint32_t prim(const Prim& p, const std::string& key) {
    auto it = prim_ids.find(key);
    if (it != prim_ids.end()) return it->second;
    prims.push_back(p);
    return prim_ids[key] = (uint32_t)prims.size() - 1;
  }
  bool compile(const DX& e, std::vector<Op>* out) {
    switch (e->kind) {
      case DExpr::CONST: { if (!e->c.is_bool()) return false; Prim p; p.kind = e->c.b ? P_TRUE : P_FALSE; out->push_back({O_PRIM, prim(p, e->c.b ? "T" : "F")}); return true; }
      case DExpr::AND: case DExpr::OR: {
        for (auto& a : e->args) if (!compile(a, out)) return false;
        out->push_back({(uint8_t)(e->kind == DExpr::AND ? O_AND : O_OR), (uint32_t)e->args.size()});
        return true;
      }
      case DExpr::NOT: { if (e->args.size() != 1 || !compile(e->args[0], out)) return false; out->push_back({O_NOT, 0}); return true; }
      case DExpr::DEFINED: { if (e->args.size() != 1 || e->args[0]->kind != DExpr::LEAF) return false; Prim p; p.kind = P_TRUE; out->push_back({O_PRIM, prim(p, "T")}); return true; }
      case DExpr::TYPE_MASK: { if (e->args.size() != 1 || e->args[0]->kind != DExpr::LEAF) return false; Prim p; p.kind = ((e->mask >> 4) & 1u) ? P_TRUE : P_FALSE; out->push_back({O_PRIM, prim(p, p.kind == P_TRUE ? "T" : "F")}); return true; }
      case DExpr::CMP: {
        if (e->args.size() != 2 || e->args[1]->kind != DExpr::CONST) return false;
        const DExpr& L = *e->args[0];
        const Value& K = e->args[1]->c;
        const std::string* sep = nullptr;
        Prim p; p.cmp = e->cmp;
        if (L.kind == DExpr::LEAF && K.is_string()) { p.kind = (e->cmp == 0 || e->cmp == 1) ? P_EQ : P_CMP; p.k = K.str(); out->push_back({O_PRIM, prim(p, e->text)}); return true; }
        if (L.kind != DExpr::CALL) return false;
        if (L.name == "count" && L.args.size() == 1 && dx_is_leaf_split(*L.args[0], &sep) && K.is_number() && K.is_int) { p.kind = P_SPLIT_COUNT; p.sep = *sep; p.num = K.i; out->push_back({O_PRIM, prim(p, e->text)}); return true; }
        if (L.name == "$index" && L.args.size() == 2 && dx_is_leaf_split(*L.args[0], &sep) && L.args[1]->kind == DExpr::CONST && L.args[1]->c.is_number() && L.args[1]->c.is_int && K.is_string()) {
          p.kind = P_SPLIT_IDX; p.sep = *sep; p.idx = (long long)L.args[1]->c.i; p.k = K.str(); out->push_back({O_PRIM, prim(p, e->text)}); return true;
        }
        return false;
      }
      case DExpr::TRUTHY: {
        if (e->args.size() != 1) return false;
        const DExpr& c = *e->args[0];
        Prim p;
        if (c.kind == DExpr::LEAF) { p.kind = P_TRUE; out->push_back({O_PRIM, prim(p, "T")}); return true; }
        if (c.kind != DExpr::CALL || c.args.size() != 2) return false;
        if ((c.name == "re_match" || c.name == "regex.match") && c.args[0]->kind == DExpr::CONST && c.args[0]->c.is_string() && c.args[1]->kind == DExpr::LEAF) { p.kind = P_REGEX; p.k = c.args[0]->c.str(); out->push_back({O_PRIM, prim(p, e->text)}); return true; }
        if (c.args[0]->kind != DExpr::LEAF || c.args[1]->kind != DExpr::CONST || !c.args[1]->c.is_string()) return false;
        p.k = c.args[1]->c.str();
        if (c.name == "startswith") p.kind = P_PREFIX; else if (c.name == "endswith") p.kind = P_SUFFIX; else if (c.name == "contains") p.kind = P_CONTAINS; else return false;
        out->push_back({O_PRIM, prim(p, e->text)});
        return true;
      }
      default: return false;
    }
  }
  template <class Entries> void build(const Entries& entries) {
    prims.clear(); prim_ids.clear(); progs.clear(); generic.clear();
    for (const auto& en : entries) {
      std::vector<Op> pr;
      const bool ok = compile(en.dx, &pr);
      if (!ok) pr.clear();
      progs.push_back(std::move(pr));
      generic.push_back(ok ? 0 : 1);
    }
    val.assign(prims.size(), -1);
    // a leaf whose expressions do not look at the string's bytes (def($), type tests: `def(review.object.metadata.name)` of every library
    // template's message) answers the same for EVERY string: worked out once, no hashing, no memo -- names are as good as unique
    constant = true;
    for (uint8_t g : generic) if (g) constant = false;
    for (const Prim& p : prims) if (p.kind != P_TRUE && p.kind != P_FALSE && p.kind != P_ISSTR) constant = false;
    constant_true.assign(progs.size(), 0);
    if (constant) { begin_value(); for (size_t i = 0; i < progs.size(); i++) constant_true[i] = eval(i, "", 0) ? 1 : 0; }
  }
  bool constant = false;                   // every entry's answer is the same for every string value
  std::vector<uint8_t> constant_true;
  bool eval_prim(uint32_t id, const char* s, size_t n) const {
    int8_t& v = val[id];
    if (v >= 0) return v != 0;
    const Prim& p = prims[id];
    bool r = false;
    switch (p.kind) {
      case P_TRUE: r = true; break;
      case P_FALSE: r = false; break;
      case P_ISSTR: r = true; break;
      case P_EQ: { const bool eq = p.k.size() == n && memcmp(p.k.data(), s, n) == 0; r = p.cmp == 0 ? eq : !eq; break; }
      case P_CMP: { const size_t m = n < p.k.size() ? n : p.k.size(); int c = m ? memcmp(s, p.k.data(), m) : 0; if (c == 0) c = n < p.k.size() ? -1 : n > p.k.size() ? 1 : 0; r = dx_cmp_holds(c, p.cmp); break; }
      case P_PREFIX: r = p.k.size() <= n && memcmp(s, p.k.data(), p.k.size()) == 0; break;
      case P_SUFFIX: r = p.k.size() <= n && memcmp(s + n - p.k.size(), p.k.data(), p.k.size()) == 0; break;
      case P_CONTAINS: {
        if (p.k.empty()) { r = true; break; }
        if (p.k.size() > n) break;
        for (size_t i = 0; i + p.k.size() <= n && !r; i++) r = s[i] == p.k[0] && memcmp(s + i, p.k.data(), p.k.size()) == 0;
        break;
      }
      case P_SPLIT_COUNT: { const i128 c = (i128)dx_split_count(s, n, p.sep); r = dx_cmp_holds(c < p.num ? -1 : c > p.num ? 1 : 0, p.cmp); break; }
      case P_SPLIT_IDX: {
        const char* cp = nullptr; size_t cn = 0;
        if (!dx_split_component(s, n, p.sep, p.idx, &cp, &cn)) break;
        const size_t m = cn < p.k.size() ? cn : p.k.size();
        int c = m ? memcmp(cp, p.k.data(), m) : 0;
        if (c == 0) c = cn < p.k.size() ? -1 : cn > p.k.size() ? 1 : 0;
        r = dx_cmp_holds(c, p.cmp);
        break;
      }
      case P_REGEX: { bool valid = true; const bool hit = builtin_regex_search(p.k, s, n, &valid); r = valid && hit; break; }
    }
    v = r ? 1 : 0;
    return r;
  }
  void begin_value() const { std::fill(val.begin(), val.end(), (int8_t)-1); dx_split_forget(); }
  bool eval(size_t entry, const char* s, size_t n) const {   // (every operand is evaluated: the primitives are memoised, the trees are small)
    stack.clear();
    for (const Op& o : progs[entry]) {
      if (o.op == O_PRIM) stack.push_back(eval_prim(o.a, s, n) ? 1 : 0);
      else if (o.op == O_NOT) stack.back() = !stack.back();
      else {
        bool r = o.op == O_AND;
        for (uint32_t k = 0; k < o.a; k++) { const bool x = stack.back() != 0; stack.pop_back(); r = o.op == O_AND ? (r && x) : (r || x); }
        stack.push_back(r ? 1 : 0);
      }
    }
    return !stack.empty() && stack.back() != 0;
  }
};

}  // namespace gk
