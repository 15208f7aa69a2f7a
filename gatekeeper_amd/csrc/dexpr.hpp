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
inline bool dx_true_str(const DX& e, const char* s, size_t n, bool* ok) {
  switch (e->kind) {
    case DExpr::CONST: if (e->c.is_bool()) return e->c.b; *ok = false; return false;
    case DExpr::AND: { for (auto& a : e->args) { if (!dx_true_str(a, s, n, ok)) return false; if (!*ok) return false; } return true; }
    case DExpr::OR: { for (auto& a : e->args) { if (dx_true_str(a, s, n, ok)) return true; if (!*ok) return false; } return false; }
    case DExpr::NOT: { const bool v = dx_true_str(e->args[0], s, n, ok); return !v; }
    case DExpr::DEFINED: if (e->args.size() == 1 && e->args[0]->kind == DExpr::LEAF) return true; *ok = false; return false;
    case DExpr::TYPE_MASK: if (e->args.size() == 1 && e->args[0]->kind == DExpr::LEAF) return ((e->mask >> 4) & 1u) != 0; *ok = false; return false;
    case DExpr::CMP: {
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

}  // namespace gk
