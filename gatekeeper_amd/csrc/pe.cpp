// Partial evaluator -- see pe.hpp.
#include "pe.hpp"

#include <functional>
#include <set>
#include <sstream>

#include "builtins.hpp"
#include "regex.hpp"
#include "plan.hpp"
#include <atomic>
#include <chrono>

namespace gk {

// ================================================================================================ formulas
namespace {
FP mkf(FNode n) { n.canon_set = false; n.canon.clear(); n.leaf_state = 0; n.needs_leaf_state = 0; n.leaf_text.clear(); return std::make_shared<const FNode>(std::move(n)); }   // (a copy that was edited carries no text)
}
// per-thread singletons: the reference count of a process-wide one would be the hottest cache line of every host thread
// that renders messages (gk_table_totals, the CPU baseline loop)
FP f_true() { static thread_local FP t = [] { FNode n; n.kind = FNode::T; return mkf(std::move(n)); }(); return t; }
FP f_false() { static thread_local FP f = [] { FNode n; n.kind = FNode::F; return mkf(std::move(n)); }(); return f; }
FP f_and(FP a, FP b) {
  if (a->kind == FNode::F || b->kind == FNode::F) return f_false();
  if (a->kind == FNode::T) return b;
  if (b->kind == FNode::T) return a;
  FNode n; n.kind = FNode::AND;
  if (a->kind == FNode::AND) n.kids = a->kids; else n.kids.push_back(a);
  if (b->kind == FNode::AND) n.kids.insert(n.kids.end(), b->kids.begin(), b->kids.end()); else n.kids.push_back(b);
  return mkf(std::move(n));
}
FP f_or(FP a, FP b) {
  if (a->kind == FNode::T || b->kind == FNode::T) return f_true();
  if (a->kind == FNode::F) return b;
  if (b->kind == FNode::F) return a;
  FNode n; n.kind = FNode::OR;
  if (a->kind == FNode::OR) n.kids = a->kids; else n.kids.push_back(a);
  if (b->kind == FNode::OR) n.kids.insert(n.kids.end(), b->kids.begin(), b->kids.end()); else n.kids.push_back(b);
  return mkf(std::move(n));
}
FP f_not(FP a) {
  if (a->kind == FNode::T) return f_false();
  if (a->kind == FNode::F) return f_true();
  if (a->kind == FNode::NOT) return a->kids[0];
  FNode n; n.kind = FNode::NOT; n.kids = {a};
  return mkf(std::move(n));
}
FP f_atom(const Atom& a) { FNode n; n.kind = FNode::ATOM; n.atom = a; return mkf(std::move(n)); }
FP f_exists(int q, const SPath& base, FP body) {
  if (body->kind == FNode::F) return f_false();
  FNode n; n.kind = FNode::EXISTS; n.q = q; n.base = base; n.kids = {body};
  return mkf(std::move(n));
}
FP f_exists2(int q, const SPath& base, FP body) {
  if (body->kind == FNode::F) return f_false();
  FNode n; n.kind = FNode::EXISTS; n.q = q; n.base = base; n.two = true; n.kids = {body};
  return mkf(std::move(n));
}
FP f_exists_k(int q, const SPath& base, FP body, int k) {
  if (k <= 1) return f_exists(q, base, body);
  if (body->kind == FNode::F) return f_false();
  FNode n; n.kind = FNode::EXISTS; n.q = q; n.base = base; n.two = true; n.atleast = k; n.kids = {body};
  return mkf(std::move(n));
}
FP f_exists_like(const FNode& proto, FP body) { return proto.two ? f_exists_k(proto.q, proto.base, body, proto.atleast) : f_exists(proto.q, proto.base, body); }
// (one node for the whole list: folding with f_and / f_or copies the growing child vector once per element)
static FP f_fold(const std::vector<FP>& v, FNode::Kind kind) {
  const FNode::Kind zero = kind == FNode::AND ? FNode::F : FNode::T, unit = kind == FNode::AND ? FNode::T : FNode::F;
  FNode n; n.kind = kind;
  for (auto& x : v) {
    if (x->kind == zero) return kind == FNode::AND ? f_false() : f_true();
    if (x->kind == unit) continue;
    if (x->kind == kind) n.kids.insert(n.kids.end(), x->kids.begin(), x->kids.end()); else n.kids.push_back(x);
  }
  if (n.kids.empty()) return kind == FNode::AND ? f_true() : f_false();
  if (n.kids.size() == 1) return n.kids[0];
  return mkf(std::move(n));
}
FP f_all(const std::vector<FP>& v) { return f_fold(v, FNode::AND); }
FP f_any(const std::vector<FP>& v) { return f_fold(v, FNode::OR); }

std::string spath_to_string(const SPath& p) {
  std::string o = "review";
  for (const Step& s : p) { if (s.iter) o += "[q" + std::to_string(s.q) + "]"; else o += "." + s.key; }
  return o;
}
static std::string f_compose_text(const FP& f);
const std::string& f_to_string(const FP& f) {
  if (!f->canon_set) { f->canon = f_compose_text(f); f->canon_set = true; }
  return f->canon;
}
static std::string f_compose_text(const FP& f) {
  static const char* cmpn[] = {"==", "!=", "<", "<=", ">", ">="};
  switch (f->kind) {
    case FNode::T: return "true";
    case FNode::F: return "false";
    case FNode::NOT: return "!(" + f_to_string(f->kids[0]) + ")";
    case FNode::AND: case FNode::OR: {
      std::string o = "(";
      for (size_t i = 0; i < f->kids.size(); i++) { if (i) o += f->kind == FNode::AND ? " & " : " | "; o += f_to_string(f->kids[i]); }
      return o + ")";
    }
    case FNode::EXISTS: return std::string(f->two ? "E" + std::to_string(f->atleast) + " q" : "E q") + std::to_string(f->q) + " in " + spath_to_string(f->base) + ". " + f_to_string(f->kids[0]);
    case FNode::ATOM: {
      const Atom& a = f->atom;
      std::string p = spath_to_string(a.path);
      switch (a.kind) {
        case Atom::DEFINED: return "def(" + p + ")";
        case Atom::TRUTHY: return "truthy(" + p + ")";
        case Atom::CMP: return p + " " + cmpn[a.cmp] + " " + to_term_string(a.k);
        case Atom::TYPE: return "type(" + p + ")&" + std::to_string(a.mask);
        case Atom::STR_PREFIX: return "prefix(" + p + "," + to_term_string(a.k) + ")";
        case Atom::STR_SUFFIX: return "suffix(" + p + "," + to_term_string(a.k) + ")";
        case Atom::STR_CONTAINS: return "contains(" + p + "," + to_term_string(a.k) + ")";
        case Atom::STR_IN_SET: return p + " in " + to_term_string(a.k);
        case Atom::STR_REGEX: return "re_match(" + to_term_string(a.k) + "," + p + ")";
        case Atom::SPLIT_CMP: return "split(" + p + ")[" + std::to_string(a.idx) + "] " + cmpn[a.cmp] + " " + to_term_string(a.k);
        case Atom::SPLIT_COUNT: return "count(split(" + p + ")) " + cmpn[a.cmp] + " " + to_term_string(a.k);
        case Atom::COUNT_CMP: return "count(" + p + ") " + cmpn[a.cmp] + " " + to_term_string(a.k);
        case Atom::FLAG: return "flag" + std::to_string(a.flag);
        case Atom::VEQ: return p + " === " + spath_to_string(a.path2);
        case Atom::SPLIT_PREFIX: return "splitprefix(" + p + "," + to_term_string(a.k) + ")";
        case Atom::KEYCMP: {
          static const char* kcn[] = {"startswith", "endswith", "contains", "isname"};
          if (a.cmp >= KC_PREFIX) return std::string(kcn[a.cmp - KC_PREFIX]) + "(key(q" + std::to_string(a.q) + ")," + to_term_string(a.k) + ")";
          return "key(q" + std::to_string(a.q) + ") " + cmpn[a.cmp] + " " + to_term_string(a.k);
        }
        case Atom::DICT: return "dict(" + p + ": " + dx_to_string(a.dx) + ")";
      }
    }
  }
  return "?";
}

// ================================================================================================ evaluator
namespace {

struct UnboundVar : std::runtime_error { using std::runtime_error::runtime_error; };

SVP mksv(SV s) { return std::make_shared<const SV>(std::move(s)); }
SVP sv_const(const Value& v) { SV s; s.kind = SV::CONST; s.c = v; return mksv(std::move(s)); }
SVP sv_path(const SPath& p) { SV s; s.kind = SV::PATH; s.path = p; return mksv(std::move(s)); }
SVP sv_bool(FP t, FP d) {
  if ((t->kind == FNode::T || t->kind == FNode::F) && d->kind == FNode::T) return sv_const(Value::boolean(t->kind == FNode::T));
  if (d->kind == FNode::F) return sv_const(Value());
  SV s; s.kind = SV::BOOLF; s.f = t; s.d = d; return mksv(std::move(s));
}

// Variable bindings of one evaluation state.  States are copied far more often than they are extended (every value an
// expression yields carries its state), so the bindings are shared between copies and cloned by the first write: a sorted
// vector behind a shared pointer (a std::map here was a node allocation per variable per copy -- a fifth of a host render).
class Env {
  typedef std::pair<std::string, SVP> E;
  std::shared_ptr<std::vector<E>> v_;
  std::vector<E>& own() {
    if (!v_) v_ = std::make_shared<std::vector<E>>();
    else if (v_.use_count() > 1) v_ = std::make_shared<std::vector<E>>(*v_);
    return *v_;
  }
  static bool before(const E& e, const std::string& k) { return e.first < k; }
 public:
  const SVP* get(const std::string& k) const {
    if (!v_) return nullptr;
    auto it = std::lower_bound(v_->begin(), v_->end(), k, before);
    return it != v_->end() && it->first == k ? &it->second : nullptr;
  }
  bool count(const std::string& k) const { return get(k) != nullptr; }
  void set(const std::string& k, const SVP& v) {
    std::vector<E>& o = own();
    auto it = std::lower_bound(o.begin(), o.end(), k, before);
    if (it != o.end() && it->first == k) it->second = v; else o.insert(it, E(k, v));
  }
  void erase(const std::string& k) {
    if (!get(k)) return;
    std::vector<E>& o = own();
    o.erase(std::lower_bound(o.begin(), o.end(), k, before));
  }
};
struct State {
  Env env;
  std::vector<FP> conds;
  std::vector<std::pair<int, SPath>> quants;
};
struct Val { SVP v; State s; };
typedef std::vector<Val> Vals;
typedef std::vector<State> States;

struct Alt { SVP v; std::vector<FP> conds; std::vector<std::pair<int, SPath>> quants; };

// ---- renaming of quantifier ids (fresh instances of generators / cached rule values)
typedef std::map<int, int> QMap;
SPath rn_path(const SPath& p, const QMap& m) {
  SPath o = p;
  for (Step& s : o) if (s.iter) { auto it = m.find(s.q); if (it != m.end()) s.q = it->second; }
  return o;
}
static bool path_mentions(const SPath& p, const QMap& m) { for (const Step& s : p) if (s.iter && m.count(s.q)) return true; return false; }
FP rn_f(const FP& f, const QMap& m) {
  if (m.empty()) return f;
  switch (f->kind) {
    case FNode::T: case FNode::F: return f;
    case FNode::ATOM: {
      // (most sub-formulas of a renaming mention none of the renamed quantifiers: they are shared, not copied)
      if (!path_mentions(f->atom.path, m) && !path_mentions(f->atom.path2, m) && !(f->atom.q >= 0 && m.count(f->atom.q))) return f;
      Atom a = f->atom;
      a.path = rn_path(a.path, m);
      a.path2 = rn_path(a.path2, m);
      if (a.q >= 0) { auto it = m.find(a.q); if (it != m.end()) a.q = it->second; }
      return f_atom(a);
    }
    default: {
      bool same = !path_mentions(f->base, m) && !(f->q >= 0 && m.count(f->q));
      std::vector<FP> kids;
      kids.reserve(f->kids.size());
      for (auto& k : f->kids) { kids.push_back(rn_f(k, m)); if (kids.back().get() != k.get()) same = false; }
      if (same) return f;
      FNode n = *f;
      n.kids = std::move(kids);
      n.base = rn_path(n.base, m);
      if (n.q >= 0) { auto it = m.find(n.q); if (it != m.end()) n.q = it->second; }
      return mkf(std::move(n));
    }
  }
}
SVP rn_sv(const SVP& v, const QMap& m) {
  if (m.empty() || !v || v->kind == SV::CONST) return v;
  SV s = *v;
  s.path = rn_path(s.path, m);
  s.hkeypath = rn_path(s.hkeypath, m);   // (the head of a formatted message names the same iterations as its operands)
  if (s.q >= 0) { auto it = m.find(s.q); if (it != m.end()) s.q = it->second; }
  for (auto& f : s.fields) f.second = rn_sv(f.second, m);
  for (auto& e : s.elems) { e.v = rn_sv(e.v, m); e.cond = rn_f(e.cond, m); }
  for (auto& g : s.gens) {
    g.elem = rn_sv(g.elem, m);
    g.cond = rn_f(g.cond, m);
    for (auto& q : g.quants) { auto it = m.find(q); if (it != m.end()) q = it->second; }
    for (auto& b : g.bases) b = rn_path(b, m);
  }
  if (s.f) s.f = rn_f(s.f, m);
  if (s.d) s.d = rn_f(s.d, m);
  return mksv(std::move(s));
}

Atom atom_path(Atom::Kind k, const SPath& p) { Atom a; a.kind = k; a.path = p; return a; }

// ---- leaf-local values (dexpr.hpp): a review leaf, its count(), or anything already derived from one leaf
bool leaf_local(const SVP& v, SPath* leaf, DX* dx) {
  if (v->kind == SV::PATH && !v->path.empty()) { *leaf = v->path; *dx = dx_leaf(); return true; }
  if (v->kind == SV::DERIVED) { if (v->idx == 1) return false;   // (a deep call narrowed to a sub-document: what it yields when the sub-document is ABSENT lives in v->c -- only comparisons / definedness / truth know about it)
                                *leaf = v->path; *dx = v->dx; return true; }
  if (v->kind == SV::COUNTOF) { *leaf = v->path; *dx = dx_node(DExpr::CALL, {dx_leaf()}, "count"); return true; }
  if (v->kind == SV::STRX && !v->path.empty()) {
    // trim(leaf, c) / split(.., sep) / a component / the component count: functions of the one string leaf
    DX base = dx_leaf();
    if (v->cut) base = dx_node(DExpr::CALL, {base, dx_const(Value::string(std::string(1, v->cut)))}, "trim");
    *leaf = v->path;
    if (v->xkind == SV::XTRIM) { *dx = base; return true; }
    DX arr = dx_node(DExpr::CALL, {base, dx_const(Value::string(std::string(1, v->sep)))}, "split");
    if (v->xkind == SV::XARR) { *dx = arr; return true; }
    if (v->xkind == SV::XCOMP) { *dx = dx_node(DExpr::CALL, {arr, dx_const(Value::integer(v->idx))}, "$index"); return true; }
    if (v->xkind == SV::XCOUNT) { *dx = dx_node(DExpr::ARITH, {dx_node(DExpr::CALL, {arr}, "count"), dx_const(Value::integer(v->idx))}, "+"); return true; }
  }
  return false;
}
SVP sv_derived(const SPath& leaf, DX dx) { SV s; s.kind = SV::DERIVED; s.path = leaf; s.dx = std::move(dx); return mksv(std::move(s)); }
FP f_dict(const SPath& leaf, DX dx) { Atom a; a.kind = Atom::DICT; a.path = leaf; a.dx = std::move(dx); return f_atom(a); }
// operands of one operation: constants and values derived from the SAME leaf -> their expressions
bool same_leaf_args(const std::vector<SVP>& args, SPath* leaf, std::vector<DX>* dxs) {
  bool have = false;
  dxs->clear();
  for (const SVP& a : args) {
    if (a->kind == SV::CONST) { if (!a->c.defined()) return false; dxs->push_back(dx_const(a->c)); continue; }
    SPath p; DX d;
    if (!leaf_local(a, &p, &d)) return false;
    if (have && spath_to_string(p) != spath_to_string(*leaf)) return false;
    *leaf = p; have = true;
    dxs->push_back(d);
  }
  return have;
}
constexpr uint32_t M_STRING = 1u << T_STRING, M_NUMBER = (1u << T_INT) | (1u << T_FLOAT), M_BOOL = 1u << T_BOOL,
                   M_NULL = 1u << T_NULL, M_ARRAY = 1u << T_ARRAY, M_OBJECT = 1u << T_OBJECT;
FP f_type(const SPath& p, uint32_t mask) { Atom a = atom_path(Atom::TYPE, p); a.mask = mask; return f_atom(a); }

int flip_cmp(int op) {
  switch (op) {
    case C_LT: return C_GT;
    case C_LE: return C_GE;
    case C_GT: return C_LT;
    case C_GE: return C_LE;
    default: return op;
  }
}
bool cmp_holds(int c, int op) {
  switch (op) {
    case C_EQ: return c == 0;
    case C_NE: return c != 0;
    case C_LT: return c < 0;
    case C_LE: return c <= 0;
    case C_GT: return c > 0;
    default: return c >= 0;
  }
}
int cmp_of(const std::string& op) {
  if (op == "==") return C_EQ;
  if (op == "!=") return C_NE;
  if (op == "<") return C_LT;
  if (op == "<=") return C_LE;
  if (op == ">") return C_GT;
  return C_GE;
}

}  // namespace

class PE {
 public:
  PE(const Template& t, const Value& params, SVP review, const Value& inventory, bool concrete, int* nq)
      : T(t), params_(params), review_(review), inventory_(inventory), concrete_(concrete), nq_(nq) {
    SV in;
    in.kind = SV::OBJ;
    in.fields.emplace_back(Value::string("parameters"), sv_const(params.defined() ? params : Value::object({})));
    in.fields.emplace_back(Value::string("review"), review);
    input_ = mksv(std::move(in));
  }

  // value of the main package's `violation` partial set
  SVP violation_set() {
    auto alts = rule_alts(T.pkg_name_, "violation");
    if (alts.empty()) return sv_const(Value::set({}));
    return alts[0].v;
  }

  FP defined_f(const SVP& v);
  FP truthy_f(const SVP& v);
  FP is_string_f(const SVP& v);

 private:
  const Template& T;
  Value params_;
  SVP review_, input_;
  Value inventory_;
  bool concrete_;
  int* nq_;
  int depth_ = 0;
  std::map<std::pair<std::string, std::string>, std::vector<Alt>> cache_;
  std::set<std::pair<std::string, std::string>> in_progress_;

  int fresh() { return (*nq_)++; }
  [[noreturn]] void unsupported(const std::string& what, int line = 0) {
    throw Unsupported("unsupported on the device plan: " + what + (line ? " (line " + std::to_string(line) + ")" : ""));
  }

  // ---------------------------------------------------------------------------------------------- rules
  const std::vector<const Rule*>* find_rules(const std::string& pkg, const std::string& name) const {
    auto it = T.rules_.find({pkg, name});
    return it == T.rules_.end() ? nullptr : &it->second;
  }

  std::vector<Alt> rule_alts(const std::string& pkg, const std::string& name) {
    auto key = std::make_pair(pkg, name);
    auto it = cache_.find(key);
    if (it != cache_.end()) return it->second;
    if (in_progress_.count(key)) throw RegoError("rego_recursion_error: rule " + name + " is recursive");
    in_progress_.insert(key);
    const auto& rules = *find_rules(pkg, name);
    std::vector<Alt> alts;
    Rule::Kind kind = rules[0]->kind;
    if (kind == Rule::Function) throw RegoError("rego_type_error: function " + name + " referenced without call");
    if (kind == Rule::PartialSet || kind == Rule::PartialObject) {
      SV out;
      out.kind = kind == Rule::PartialSet ? SV::SET : SV::OBJ;
      std::vector<std::pair<SVP, SVP>> obj_pairs;
      for (const Rule* r : rules) {
        State s0;
        States st = eval_body(r->body, s0, r);
        for (State& s : st) {
          Vals ks;
          eval_term(r->key, s, r, ks);
          for (Val& kv : ks) {
            if (kind == Rule::PartialObject) {
              Vals vs;
              eval_term(r->value, kv.s, r, vs);
              for (Val& vv : vs) {
                if (!kv.s.conds.empty() || !kv.s.quants.empty() || kv.v->kind != SV::CONST) unsupported("partial object rule over review data", r->line);
                obj_pairs.emplace_back(kv.v, vv.v);
              }
              continue;
            }
            add_member(out, kv.v, kv.s, 0, 0);
          }
        }
      }
      SVP v;
      if (kind == Rule::PartialObject) {
        SV o; o.kind = SV::OBJ;
        for (auto& p : obj_pairs) o.fields.emplace_back(p.first->c, p.second);
        v = fold(mksv(std::move(o)));
      } else v = fold(mksv(std::move(out)));
      alts.push_back({v, {}, {}});
    } else {
      Alt def;
      bool has_default = false;
      for (const Rule* r : rules) {
        if (r->is_default) {
          Vals vs; State s0;
          eval_term(r->value, s0, r, vs);
          if (!vs.empty()) { def = {vs[0].v, {}, {}}; has_default = true; }
          continue;
        }
        State s0;
        complete_def(r, s0, alts);
      }
      if (has_default) {
        // default applies when no other definition is defined
        FP any = f_false();
        for (Alt& a : alts) any = f_or(any, close_alt(a));
        FP none = f_not(any);
        if (none->kind != FNode::F) { def.conds = {none}; if (none->kind == FNode::T) def.conds.clear(); alts.push_back(def); }
      }
    }
    in_progress_.erase(key);
    cache_[key] = alts;
    return alts;
  }

  FP close_alt(const Alt& a) {
    FP f = f_all(a.conds);
    for (auto it = a.quants.rbegin(); it != a.quants.rend(); ++it) f = f_exists(it->first, it->second, f);
    return f;
  }

  // one complete-rule / function definition with its else chain, continuing from state `s`
  void complete_def(const Rule* r, const State& s, std::vector<Alt>& out) {
    std::vector<std::pair<TermP, const Body*>> chain;
    chain.emplace_back(r->value, &r->body);
    for (auto& e : r->elses) chain.emplace_back(e.first, &e.second);
    FP prior = f_false();   // some earlier link of the chain was defined
    for (auto& link : chain) {
      States st = eval_body(*link.second, s, r);
      FP here = f_false();
      for (State& b : st) {
        Vals vs;
        if (link.first) eval_term(link.first, b, r, vs);
        else vs.push_back({sv_const(Value::boolean(true)), b});
        for (Val& v : vs) {
          Alt a;
          a.v = v.v;
          a.conds.assign(v.s.conds.begin() + s.conds.size(), v.s.conds.end());
          a.quants.assign(v.s.quants.begin() + s.quants.size(), v.s.quants.end());
          FP dv = defined_f(v.v);
          if (dv->kind == FNode::F) continue;
          if (dv->kind != FNode::T) a.conds.push_back(dv);
          if (prior->kind != FNode::F) a.conds.push_back(f_not(prior));
          here = f_or(here, close_alt(a));
          out.push_back(a);
        }
      }
      if (chain.size() == 1) break;
      prior = f_or(prior, here);
      if (prior->kind == FNode::T) break;
    }
  }

  void call_function(const std::string& pkg, const std::string& name, const std::vector<SVP>& args, const State& s,
                     const Rule* /*caller*/, Vals& out) {
    const auto& rules = *find_rules(pkg, name);
    if (++depth_ > 64) { depth_--; throw RegoError("rego_recursion_error: call depth exceeded in " + name); }
    for (const Rule* r : rules) {
      if (r->kind != Rule::Function || r->args.size() != args.size()) continue;
      State fs;
      fs.conds = s.conds;
      fs.quants = s.quants;
      States cur{fs};
      for (size_t i = 0; i < args.size(); i++) {
        States nxt;
        for (State& c : cur) unify_value(r->args[i], args[i], c, r, nxt);
        cur.swap(nxt);
      }
      for (State& c : cur) {
        std::vector<Alt> alts;
        complete_def(r, c, alts);
        for (Alt& a : alts) {
          State o = s;
          o.conds = c.conds;
          o.quants = c.quants;
          o.conds.insert(o.conds.end(), a.conds.begin(), a.conds.end());
          o.quants.insert(o.quants.end(), a.quants.begin(), a.quants.end());
          out.push_back({a.v, o});
        }
      }
    }
    depth_--;
  }

  // ---------------------------------------------------------------------------------------------- sets
  // add `v` (under the extra conds/quants state s has beyond the given base sizes) to set/array SV `out`
  void add_member(SV& out, const SVP& v, const State& s, size_t base_conds, size_t base_quants) {
    FP cond = f_true();
    for (size_t i = base_conds; i < s.conds.size(); i++) cond = f_and(cond, s.conds[i]);
    cond = f_and(cond, defined_f(v));
    if (cond->kind == FNode::F) return;
    if (s.quants.size() == base_quants) { out.elems.push_back({v, cond}); return; }
    Gen g;
    g.elem = v;
    g.cond = cond;
    for (size_t i = base_quants; i < s.quants.size(); i++) { g.quants.push_back(s.quants[i].first); g.bases.push_back(s.quants[i].second); }
    out.gens.push_back(g);
  }

  // fold fully-constant composites back into CONST values
  SVP fold(const SVP& v) {
    if (v->kind == SV::OBJ) {
      ValuePairs p;
      for (auto& f : v->fields) { if (f.second->kind != SV::CONST || !f.second->c.defined()) return v; p.emplace_back(f.first, f.second->c); }
      return sv_const(Value::object(p));
    }
    if (v->kind == SV::ARR || v->kind == SV::SET) {
      if (!v->gens.empty()) return v;
      ValueVec items;
      for (auto& e : v->elems) { if (e.v->kind != SV::CONST || e.cond->kind != FNode::T) return v; items.push_back(e.v->c); }
      return sv_const(v->kind == SV::ARR ? Value::array(items) : Value::set(items));
    }
    return v;
  }

  // view any set-like SV as (elems, gens)
  bool as_setlike(const SVP& v, std::vector<CondElem>* elems, std::vector<Gen>* gens) {
    if (v->kind == SV::CONST && (v->c.is_set() || v->c.is_array())) {
      for (const Value& x : v->c.items()) elems->push_back({sv_const(x), f_true()});
      return true;
    }
    if (v->kind == SV::SET || v->kind == SV::ARR) { *elems = v->elems; *gens = v->gens; return true; }
    return false;
  }

  FP member_f(const SVP& x, const SVP& setlike) {
    if (setlike->kind == SV::CONST && x->kind == SV::CONST) {
      const Value& s = setlike->c;
      if (s.is_set()) return s.set_has(x->c) ? f_true() : f_false();
      if (s.is_array()) { for (auto& e : s.items()) if (e == x->c) return f_true(); return f_false(); }
      return f_false();
    }
    if (setlike->kind == SV::CONST && (setlike->c.is_set() || setlike->c.is_array()) && x->kind == SV::PATH) {
      bool all_str = true;
      for (auto& e : setlike->c.items()) if (!e.is_string()) all_str = false;
      if (all_str) {
        if (setlike->c.size() == 0) return f_false();
        Atom a = atom_path(Atom::STR_IN_SET, x->path);
        a.k = setlike->c;
        return f_atom(a);
      }
    }
    std::vector<CondElem> elems;
    std::vector<Gen> gens;
    if (!as_setlike(setlike, &elems, &gens)) unsupported("membership in a non-collection");
    FP r = f_false();
    for (auto& e : elems) r = f_or(r, f_and(e.cond, compare_f(e.v, C_EQ, x)));
    for (auto& g : gens) {
      QMap m;
      for (int q : g.quants) m[q] = fresh();
      FP body = f_and(rn_f(g.cond, m), compare_f(rn_sv(g.elem, m), C_EQ, x));
      for (size_t i = g.quants.size(); i-- > 0;) body = f_exists(m[g.quants[i]], rn_path(g.bases[i], m), body);
      r = f_or(r, body);
    }
    return r;
  }

  FP nonempty_f(const std::vector<CondElem>& elems, const std::vector<Gen>& gens) {
    FP r = f_false();
    for (auto& e : elems) r = f_or(r, e.cond);
    for (auto& g : gens) {
      FP body = g.cond;
      for (size_t i = g.quants.size(); i-- > 0;) body = f_exists(g.quants[i], g.bases[i], body);
      r = f_or(r, body);
    }
    return r;
  }

  FP at_least(const std::vector<FP>& c, size_t i, int k, std::map<std::pair<size_t, int>, FP>& memo) {
    if (k <= 0) return f_true();
    if (c.size() - i < (size_t)k) return f_false();
    auto key = std::make_pair(i, k);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    FP r = f_or(f_and(c[i], at_least(c, i + 1, k - 1, memo)), at_least(c, i + 1, k, memo));
    memo[key] = r;
    return r;
  }

  FP card_cmp(const SVP& card, int op, const Value& nval) {
    if (!nval.is_number()) {   // number vs non-number: rank order decides
      int c = compare(Value::integer(0), nval);
      return cmp_holds(c, op) ? f_true() : f_false();
    }
    double n = nval.as_double();
    FP ne = nonempty_f(card->elems, card->gens);
    auto is = [&](int o, double v) { return op == o && n == v; };
    if (is(C_GT, 0) || is(C_GE, 1) || is(C_NE, 0)) return ne;
    if (is(C_EQ, 0) || is(C_LT, 1) || is(C_LE, 0)) return f_not(ne);
    if (!card->gens.empty()) unsupported("count() of a set built from review data compared with a constant other than 0");
    for (auto& e : card->elems) if (e.v->kind != SV::CONST && card->idx == 0) unsupported("count() of a set with symbolic members");
    std::vector<FP> conds;
    for (auto& e : card->elems) conds.push_back(e.cond);
    if (conds.size() > 16) unsupported("count() over more than 16 conditional members");
    std::map<std::pair<size_t, int>, FP> memo;
    auto ge = [&](double k) { if (k > (double)conds.size()) return f_false(); if (k <= 0) return f_true(); memo.clear(); return at_least(conds, 0, (int)std::ceil(k), memo); };
    switch (op) {
      case C_GE: return ge(n);
      case C_GT: return ge(std::floor(n) + 1);
      case C_LE: return f_not(ge(std::floor(n) + 1));
      case C_LT: return f_not(ge(n));
      case C_EQ: if (std::floor(n) != n) return f_false(); return f_and(ge(n), f_not(ge(n + 1)));
      default: if (std::floor(n) != n) return f_true(); return f_not(f_and(ge(n), f_not(ge(n + 1))));
    }
  }

  // ---- formatted strings whose parts are known (sprintf over an array literal): [CONST text | PATH leaf] ...
  // false (parts left empty: a plain opaque string) for anything but literal text, %% and %v / %s verbs matched by the arguments
  static bool fmt_parts(const std::string& fmt, const std::vector<CondElem>& args, std::vector<CondElem>* parts) {
    std::vector<CondElem> o;
    std::string lit;
    size_t next = 0;
    auto flush = [&]() { if (!lit.empty()) { o.push_back({sv_const(Value::string(lit)), f_true()}); lit.clear(); } };
    for (size_t i = 0; i < fmt.size(); i++) {
      if (fmt[i] != '%') { lit += fmt[i]; continue; }
      if (i + 1 >= fmt.size()) return false;
      const char v = fmt[++i];
      if (v == '%') { lit += '%'; continue; }
      if ((v != 'v' && v != 's') || next >= args.size()) return false;
      const CondElem& e = args[next++];
      if (!e.cond || e.cond->kind != FNode::T) return false;
      if (e.v->kind == SV::CONST) {
        if (!e.v->c.is_string()) return false;   // (how a constant number prints is the concrete evaluator's business)
        lit += *e.v->c.s;
      } else if (e.v->kind == SV::PATH) { flush(); o.push_back({e.v, f_true()}); }
      else return false;
    }
    if (next != args.size()) return false;   // (surplus arguments print as %!(EXTRA ...))
    flush();
    *parts = o;
    return true;
  }
  // ---- the HEAD of a formatted string, for the result counting (pe.hpp SV::hhead): literal text, the first verb's operand when it
  // is a review leaf printed with %v / %s, the literal text behind it; plus a signature of the format and every operand
  static std::string path_sig(const SPath& p) { std::string o; for (auto& st : p) { if (st.iter) o += "[]"; else { o += "."; o += st.key; } } return o; }
  static void fmt_head(const std::string& fmt, const std::vector<CondElem>& args, SV* o) {
    o->hsig.clear();
    o->hsig.push_back(fmt);
    for (auto& e : args) {
      if (!e.cond || e.cond->kind != FNode::T) o->hsig.push_back("?");
      else if (e.v->kind == SV::CONST) o->hsig.push_back("C" + to_term_string(e.v->c));
      else if (e.v->kind == SV::PATH) o->hsig.push_back("P" + path_sig(e.v->path));
      else o->hsig.push_back("?");
    }
    std::string pre, sep;
    size_t i = 0;
    auto literal = [&](std::string* out) -> bool {   // literal text up to the next verb (%% is a literal %); false: malformed
      for (; i < fmt.size(); i++) {
        if (fmt[i] != '%') { out->push_back(fmt[i]); continue; }
        if (i + 1 >= fmt.size()) return false;
        if (fmt[i + 1] == '%') { out->push_back('%'); i++; continue; }
        return true;
      }
      return true;
    };
    if (!literal(&pre)) return;
    o->hhead = true;
    o->hpre = pre;
    o->hkeypath.clear();
    if (i >= fmt.size()) { o->hsep_tail = true; return; }   // (no verb at all: the text is a constant -- handled as literal head only)
    const char v = fmt[i + 1];
    if ((v != 'v' && v != 's') || args.empty() || !args[0].cond || args[0].cond->kind != FNode::T || args[0].v->kind != SV::PATH) return;   // head = literal text only
    i += 2;
    if (!literal(&sep)) { o->hhead = false; return; }
    o->hkeypath = args[0].v->path;
    o->hsep = sep;
    o->hsep_tail = i >= fmt.size();
  }
  // review leaves that are strings by construction: what gkReview / AdmissionRequest carry as Go strings
  static bool string_by_construction(const SPath& p) {
    for (auto& st : p) if (st.iter) return false;
    if (p.size() == 1) return p[0].key == "name" || p[0].key == "namespace" || p[0].key == "operation";
    if (p.size() == 2 && p[0].key == "kind") return p[1].key == "group" || p[1].key == "version" || p[1].key == "kind";
    return false;
  }
  // could `%v` of a NON-string JSON value print this text?  (numbers, booleans, null, [..], {..}, set(): then the piece
  // cannot be decided by a string comparison on the leaf alone)
  static bool prints_like_non_string(const std::string& t) {
    if (t.empty()) return false;
    if (t == "true" || t == "false" || t == "null" || t == "<nil>" || t[0] == '[' || t[0] == '{' || t.compare(0, 4, "map[") == 0 || t.compare(0, 4, "set(") == 0) return true;
    const char c = t[0];
    return (c >= '0' && c <= '9') || c == '-' || c == '+' || c == '.' || t == "NaN" || t == "Inf";
  }
  // parts[i ..] spell exactly text[pos ..]: OR over the ways of cutting the text, AND of "leaf == piece" per cut
  FP fmt_equals(const std::vector<CondElem>& parts, size_t i, const std::string& text, size_t pos, int* budget) {
    if (i == parts.size()) return pos == text.size() ? f_true() : f_false();
    const SVP& v = parts[i].v;
    if (v->kind == SV::CONST) {
      const std::string& lit = *v->c.s;
      if (text.compare(pos, lit.size(), lit) != 0 || pos + lit.size() > text.size()) return f_false();
      return fmt_equals(parts, i + 1, text, pos + lit.size(), budget);
    }
    FP r = f_false();
    const bool last = i + 1 == parts.size();
    for (size_t end = last ? text.size() : pos; end <= text.size(); end++) {
      if (--*budget < 0) unsupported("comparison of a formatted string with a constant: too many ways to cut the constant");
      FP rest = fmt_equals(parts, i + 1, text, end, budget);
      if (rest->kind == FNode::F) continue;
      const std::string piece = text.substr(pos, end - pos);
      if (!string_by_construction(v->path) && prints_like_non_string(piece))
        unsupported("comparison of a formatted review value with a constant that a number, boolean or container could print as");
      Atom c = atom_path(Atom::CMP, v->path);
      c.cmp = C_EQ; c.k = Value::string(piece);
      r = f_or(r, f_and(f_atom(c), rest));
    }
    return r;
  }

  // ---------------------------------------------------------------------------------------------- predicates
 public:
  FP compare_f(const SVP& a, int op, const SVP& b) {
    if (a->kind == SV::CONST && b->kind == SV::CONST) {
      if (!a->c.defined() || !b->c.defined()) return f_false();
      return cmp_holds(compare(a->c, b->c), op) ? f_true() : f_false();
    }
    if (a->kind == SV::CONST) return compare_f(b, flip_cmp(op), a);
    if (b->kind == SV::CONST && !b->c.defined()) return f_false();
    if (a->kind == SV::DERIVED && a->idx == 1 && b->kind == SV::CONST) {   // a deep call on a narrowed sub-document: present | absent
      FP present = f_dict(a->path, dx_node(DExpr::CMP, {a->dx, dx_const(b->c)}, "", op));
      FP absent = a->c.defined() && cmp_holds(compare(a->c, b->c), op) ? f_and(a->f, f_not(f_atom(atom_path(Atom::DEFINED, a->path)))) : f_false();
      return f_or(present, absent);
    }
    if (a->kind == SV::DERIVED || b->kind == SV::DERIVED) {   // a computation on one leaf against a constant / the same leaf
      SPath leaf; std::vector<DX> dx;
      if (same_leaf_args({a, b}, &leaf, &dx)) return f_dict(leaf, dx_node(DExpr::CMP, {dx[0], dx[1]}, "", op));
      unsupported("comparison between values derived from different review fields");
    }
    switch (a->kind) {
      case SV::OPAQUE:
        if (!a->elems.empty() && b->kind == SV::CONST && (op == C_EQ || op == C_NE)) {   // a formatted string with known parts
          if (!b->c.is_string()) return op == C_EQ ? f_false() : a->f;   // a string never equals a non-string
          int budget = 256;
          FP eq = fmt_equals(a->elems, 0, *b->c.s, 0, &budget);
          return op == C_EQ ? f_and(a->f, eq) : f_and(a->f, f_not(eq));
        }
        break;
      case SV::PATH:
        if (b->kind == SV::PATH && spath_to_string(a->path) == spath_to_string(b->path) && (op == C_EQ || op == C_NE)) {   // a value and itself
          FP d = f_atom(atom_path(Atom::DEFINED, a->path));
          return op == C_EQ ? d : f_false();
        }
        if (b->kind == SV::CONST) {
          const Value& k = b->c;
          if (k.is_array() || k.is_object() || k.is_set()) {
            if (k.size() == 0 && !k.is_set() && (op == C_EQ || op == C_NE)) {
              Atom c = atom_path(Atom::COUNT_CMP, a->path);
              c.cmp = C_EQ; c.k = Value::integer(0);
              FP eq = f_and(f_type(a->path, k.is_array() ? M_ARRAY : M_OBJECT), f_atom(c));
              return op == C_EQ ? eq : f_and(f_atom(atom_path(Atom::DEFINED, a->path)), f_not(eq));
            }
            unsupported("comparison of review data with a composite constant");
          }
          if (k.is_number() && !k.is_int && false) unsupported("");
          Atom c = atom_path(Atom::CMP, a->path);
          c.cmp = op; c.k = k;
          return f_atom(c);
        }
        if (b->kind == SV::PATH) {
          if (op != C_EQ && op != C_NE) unsupported("ordering comparison between two review values");
          Atom c = atom_path(Atom::VEQ, a->path);
          c.path2 = b->path;
          FP eq = f_atom(c);
          if (op == C_EQ) return eq;
          return f_and(f_and(f_atom(atom_path(Atom::DEFINED, a->path)), f_atom(atom_path(Atom::DEFINED, b->path))), f_not(eq));
        }
        break;
      case SV::KEYOF:
        if (b->kind == SV::CONST && (op == C_EQ || op == C_NE)) {
          // member names are strings, array indices numbers: anything else can never be equal
          if (!b->c.is_string() && !b->c.is_number()) return op == C_EQ ? f_false() : f_true();
          Atom c; c.kind = Atom::KEYCMP; c.q = a->q; c.cmp = op; c.k = b->c;
          return f_atom(c);
        }
        break;
      case SV::STRX:
        if (b->kind == SV::CONST) {
          if (a->xkind == SV::XCOMP) {
            if (!b->c.is_string()) {   // string vs non-string: rank order, needs the component to exist
              int c = compare(Value::string(""), b->c);
              return f_and(defined_f(a), cmp_holds(c, op) ? f_true() : f_false());
            }
            Atom c = atom_path(Atom::SPLIT_CMP, a->path);
            c.cut = a->cut; c.sep = a->sep; c.idx = a->idx; c.cmp = op; c.k = b->c;
            return f_atom(c);
          }
          if (a->xkind == SV::XCOUNT && b->c.is_number()) {
            Atom c = atom_path(Atom::SPLIT_COUNT, a->path);
            c.cut = a->cut; c.sep = a->sep; c.cmp = op;
            Value adj = rego_arith("-", b->c, Value::integer(a->idx));
            c.k = adj;
            if (!adj.is_int) unsupported("non-integer bound on split count");
            return f_atom(c);
          }
        }
        break;
      case SV::COUNTOF:
        if (b->kind == SV::CONST && b->c.is_number() && b->c.is_int) {
          Atom c = atom_path(Atom::COUNT_CMP, a->path);
          c.cmp = op; c.k = b->c;
          // count() of a string is its length in code points: a dictionary predicate on the leaf
          FP str = f_and(f_type(a->path, M_STRING), f_dict(a->path, dx_node(DExpr::CMP, {dx_node(DExpr::CALL, {dx_leaf()}, "count"), dx_const(b->c)}, "", op)));
          return f_or(f_and(f_type(a->path, M_ARRAY | M_OBJECT), f_atom(c)), str);
        }
        break;
      case SV::CARD:
        if (b->kind == SV::CONST) return card_cmp(a, op, b->c);
        break;
      case SV::BOOLF:
        if (b->kind == SV::CONST && (op == C_EQ || op == C_NE)) {
          if (!b->c.is_bool()) return op == C_EQ ? f_false() : a->d;
          bool want = b->c.b == (op == C_EQ);
          return f_and(a->d, want ? a->f : f_not(a->f));
        }
        break;
      case SV::OBJ: case SV::ARR: case SV::SET: {
        SVP fa = fold(a);
        if (fa->kind == SV::CONST) return compare_f(fa, op, b);
        break;
      }
      default: break;
    }
    unsupported("comparison between these symbolic operands");
  }

 private:
  // ---------------------------------------------------------------------------------------------- bodies
  States eval_body(const Body& lits, const State& s, const Rule* r) {
    std::vector<const Literal*> ptrs;
    for (auto& l : lits) ptrs.push_back(&l);
    States out;
    eval_lits(ptrs, s, r, out);
    return out;
  }

  void eval_lits(const std::vector<const Literal*>& lits, const State& s, const Rule* r, States& out) {
    if (lits.empty()) { out.push_back(s); return; }
    for (size_t k = 0; k < lits.size(); k++) {
      States next;
      try {
        eval_literal(*lits[k], s, r, next);
      } catch (const UnboundVar&) {
        continue;   // try a later literal first (OPA reorders bodies for safety at compile time)
      }
      std::vector<const Literal*> rest;
      for (size_t j = 0; j < lits.size(); j++) if (j != k) rest.push_back(lits[j]);
      for (State& n : next) eval_lits(rest, n, r, out);
      return;
    }
    throw UnboundVar("rego_unsafe_var_error: body has no evaluable literal");
  }

  FP close_states(const States& st, const State& base) {
    FP r = f_false();
    for (const State& s : st) {
      FP f = f_true();
      for (size_t i = base.conds.size(); i < s.conds.size(); i++) f = f_and(f, s.conds[i]);
      for (size_t i = s.quants.size(); i-- > base.quants.size();) f = f_exists(s.quants[i].first, s.quants[i].second, f);
      r = f_or(r, f);
    }
    return r;
  }

  // `dict(leaf: X == k)` with a constant k: the pieces (X by its canonical text)
  static bool dict_eq_const(const FP& c, const Atom** at, const DExpr** lhs, const Value** k) {
    if (c->kind != FNode::ATOM || c->atom.kind != Atom::DICT || !c->atom.dx) return false;
    const DExpr& d = *c->atom.dx;
    if (d.kind != DExpr::CMP || d.cmp != C_EQ || d.args.size() != 2) return false;
    const int ci = d.args[1]->kind == DExpr::CONST ? 1 : d.args[0]->kind == DExpr::CONST ? 0 : -1;
    if (ci < 0 || d.args[1 - ci]->kind == DExpr::CONST) return false;
    *at = &c->atom; *lhs = d.args[1 - ci].get(); *k = &d.args[ci]->c;
    return true;
  }
  void push_cond(State s, FP c, States& out) {
    if (c->kind == FNode::F) return;
    if (c->kind != FNode::T) {
      // A state that already holds X == k1 cannot also hold X == k2: the alternatives of a helper with constant heads applied
      // twice to the same value (K8sContainerLimits' mem_multiple(suffix): 14 x 14 states, 14 of them feasible) stop here
      // instead of travelling through every later stage as formulas that are false by construction.
      const Atom* at; const DExpr* lhs; const Value* k;
      if (dict_eq_const(c, &at, &lhs, &k)) {
        std::string leaf;
        for (const FP& o : s.conds) {
          const Atom* at2; const DExpr* lhs2; const Value* k2;
          if (!dict_eq_const(o, &at2, &lhs2, &k2) || lhs2->text != lhs->text) continue;
          if (leaf.empty()) leaf = spath_to_string(at->path);
          if (spath_to_string(at2->path) != leaf) continue;
          if (compare(*k, *k2) != 0) return;                     // contradiction
          out.push_back(std::move(s)); return;                    // the same condition again
        }
      }
      s.conds.push_back(c);
    }
    out.push_back(std::move(s));
  }

  void eval_literal(const Literal& l, const State& s, const Rule* r, States& out) {
    switch (l.kind) {
      case Literal::Expr: {
        Vals vs;
        eval_term(l.a, s, r, vs);
        for (Val& v : vs) push_cond(v.s, truthy_f(v.v), out);
        break;
      }
      case Literal::Assign: case Literal::Unify: unify_terms(l.a, l.b, s, r, out); break;
      case Literal::Not: {
        States inner;
        eval_literal(*l.inner, s, r, inner);
        push_cond(s, f_not(close_states(inner, s)), out);
        break;
      }
      case Literal::Some: {
        State n = s;
        for (auto& nm : l.names) n.env.erase(nm);
        out.push_back(n);
        break;
      }
      case Literal::SomeIn: {
        Vals cs;
        eval_term(l.c, s, r, cs);
        for (Val& c : cs)
          iterate(c.v, c.s, l.line, [&](const SVP& key, const SVP& val, const State& s2) {
            States s3;
            unify_value(l.b, val, s2, r, s3);
            for (State& x : s3) {
              if (!l.a) { out.push_back(x); continue; }
              if (!key) unsupported("key of a conditional array", l.line);
              unify_value(l.a, key, x, r, out);
            }
          });
        break;
      }
      case Literal::Every: {
        Vals cs;
        eval_term(l.c, s, r, cs);
        for (Val& c : cs) {
          FP all = f_true();
          if (c.v->kind == SV::CONST) {
            iterate(c.v, c.s, l.line, [&](const SVP& key, const SVP& val, const State& s2) {
              States s3, s4;
              unify_value(l.b, val, s2, r, s3);
              for (State& x : s3) { if (l.a) unify_value(l.a, key, x, r, s4); else s4.push_back(x); }
              States sat;
              for (State& x : s4) { States b = eval_body(*l.body, x, r); sat.insert(sat.end(), b.begin(), b.end()); }
              all = f_and(all, close_states(sat, c.s));
            });
          } else {
            // every x in P { body }  ==  not exists x in P: not body
            FP some_bad = f_false();
            iterate(c.v, c.s, l.line, [&](const SVP& key, const SVP& val, const State& s2) {
              States s3, s4;
              unify_value(l.b, val, s2, r, s3);
              for (State& x : s3) { if (l.a) { if (!key) unsupported("key of a conditional array", l.line); unify_value(l.a, key, x, r, s4); } else s4.push_back(x); }
              for (State& x : s4) {
                States b = eval_body(*l.body, x, r);
                State bad = x;
                FP nb = f_not(close_states(b, x));
                if (nb->kind == FNode::F) continue;
                if (nb->kind != FNode::T) bad.conds.push_back(nb);
                some_bad = f_or(some_bad, close_states({bad}, c.s));
              }
            });
            // the domain is bound before the quantifier runs (OPA rewrites `every x in <ref>` into `d = <ref>; every x in d`):
            // over an UNDEFINED domain the expression is undefined, not vacuously true
            all = f_and(defined_f(c.v), f_not(some_bad));
          }
          push_cond(c.s, all, out);
        }
        break;
      }
    }
  }

  // ---------------------------------------------------------------------------------------------- unification
  bool is_unbound(const TermP& t, const State& s, const Rule* r) {
    return t->kind == Term::Var && !s.env.count(t->name) && !is_global(t->name, r);
  }
  bool is_global(const std::string& name, const Rule* r) {
    if (name == "input" || name == "data") return true;
    if (find_rules(rule_pkg(r), name)) return true;
    return import_of(r, name) != nullptr;
  }
  const std::string& rule_pkg(const Rule* r) { return T.rule_pkgs_.at(r); }
  const std::vector<std::string>* import_of(const Rule* r, const std::string& alias) {
    const Module* m = T.rule_mods_.at(r);
    for (auto& im : m->imports) if (im.second == alias && im.first.size() > 1) return &im.first;
    return nullptr;
  }
  bool has_unbound(const TermP& t, const State& s, const Rule* r) {
    if (t->kind == Term::Var) return is_unbound(t, s, r);
    if (t->kind == Term::Array || t->kind == Term::Object) { for (auto& a : t->args) if (has_unbound(a, s, r)) return true; }
    return false;
  }

  void bind(State s, const std::string& name, const SVP& v, States& out) {
    FP d = defined_f(v);
    if (d->kind == FNode::F) return;
    if (d->kind != FNode::T) s.conds.push_back(d);
    if (name.compare(0, 2, "$w") != 0) s.env.set(name, v);
    out.push_back(std::move(s));
  }

  void unify_terms(const TermP& a, const TermP& b, const State& s, const Rule* r, States& out) {
    if (is_unbound(a, s, r)) {
      Vals vs;
      eval_term(b, s, r, vs);
      for (Val& v : vs) bind(v.s, a->name, v.v, out);
    } else if (is_unbound(b, s, r)) {
      Vals vs;
      eval_term(a, s, r, vs);
      for (Val& v : vs) bind(v.s, b->name, v.v, out);
    } else if ((a->kind == Term::Array || a->kind == Term::Object) && has_unbound(a, s, r)) {
      Vals vs;
      eval_term(b, s, r, vs);
      for (Val& v : vs) unify_value(a, v.v, v.s, r, out);
    } else if ((b->kind == Term::Array || b->kind == Term::Object) && has_unbound(b, s, r)) {
      Vals vs;
      eval_term(a, s, r, vs);
      for (Val& v : vs) unify_value(b, v.v, v.s, r, out);
    } else {
      Vals as;
      eval_term(a, s, r, as);
      for (Val& x : as) {
        Vals bs;
        eval_term(b, x.s, r, bs);
        for (Val& y : bs) push_cond(y.s, compare_f(x.v, C_EQ, y.v), out);
      }
    }
  }

  // unify pattern term against a value
  void unify_value(const TermP& pat, const SVP& val, const State& s, const Rule* r, States& out) {
    if (pat->kind == Term::Var && is_unbound(pat, s, r)) { bind(s, pat->name, val, out); return; }
    if (pat->kind == Term::Array && has_unbound(pat, s, r)) {
      std::vector<SVP> items;
      if (val->kind == SV::CONST && val->c.is_array()) for (auto& x : val->c.items()) items.push_back(sv_const(x));
      else if (val->kind == SV::ARR && val->gens.empty()) { for (auto& e : val->elems) { if (e.cond->kind != FNode::T) unsupported("destructuring a conditional array", pat->line); items.push_back(e.v); } }
      else if (val->kind == SV::CONST) return;
      else unsupported("destructuring review data", pat->line);
      if (items.size() != pat->args.size()) return;
      States cur{s};
      for (size_t i = 0; i < items.size(); i++) { States nxt; for (State& c : cur) unify_value(pat->args[i], items[i], c, r, nxt); cur.swap(nxt); }
      out.insert(out.end(), cur.begin(), cur.end());
      return;
    }
    if (pat->kind == Term::Object && has_unbound(pat, s, r)) {
      size_t n = pat->args.size() / 2;
      States cur{s};
      for (size_t i = 0; i < n; i++) {
        States nxt;
        for (State& c : cur) {
          Vals ks;
          eval_term(pat->args[2 * i], c, r, ks);
          for (Val& k : ks) {
            if (k.v->kind != SV::CONST) unsupported("symbolic key in object pattern", pat->line);
            SVP field;
            if (val->kind == SV::CONST) { if (!val->c.is_object() || val->c.size() != n) continue; const Value* f = val->c.get(k.v->c); if (f) field = sv_const(*f); }
            else if (val->kind == SV::OBJ) { if (val->fields.size() != n) continue; for (auto& f : val->fields) if (f.first == k.v->c) field = f.second; }
            else unsupported("object pattern against review data", pat->line);
            if (!field) continue;
            unify_value(pat->args[2 * i + 1], field, k.s, r, nxt);
          }
        }
        cur.swap(nxt);
      }
      out.insert(out.end(), cur.begin(), cur.end());
      return;
    }
    Vals vs;
    eval_term(pat, s, r, vs);
    for (Val& v : vs) push_cond(v.s, compare_f(v.v, C_EQ, val), out);
  }

  // ---------------------------------------------------------------------------------------------- terms
  void eval_term(const TermP& t, const State& s, const Rule* r, Vals& out) {
    switch (t->kind) {
      case Term::Scalar: out.push_back({sv_const(t->value), s}); break;
      case Term::Var: eval_var(t, s, r, out); break;
      case Term::Ref: eval_ref(t, s, r, out); break;
      case Term::Call: eval_call(t, s, r, out); break;
      case Term::BinOp: eval_binop(t, s, r, out); break;
      case Term::Array: case Term::SetLit: {
        eval_seq(t->args, 0, {}, s, r, [&](const std::vector<SVP>& vals, const State& s2) {
          SV v;
          v.kind = t->kind == Term::Array ? SV::ARR : SV::SET;
          for (auto& x : vals) v.elems.push_back({x, f_true()});
          out.push_back({fold(mksv(std::move(v))), s2});
        });
        break;
      }
      case Term::Object: {
        eval_seq(t->args, 0, {}, s, r, [&](const std::vector<SVP>& vals, const State& s2) {
          SV v;
          v.kind = SV::OBJ;
          for (size_t i = 0; i + 1 < vals.size(); i += 2) {
            if (vals[i]->kind != SV::CONST) unsupported("symbolic object key", t->line);
            v.fields.emplace_back(vals[i]->c, vals[i + 1]);
          }
          out.push_back({fold(mksv(std::move(v))), s2});
        });
        break;
      }
      case Term::ArrComp: case Term::SetComp: {
        SV v;
        v.kind = t->kind == Term::ArrComp ? SV::ARR : SV::SET;
        States st = eval_body(*t->body, s, r);
        for (State& b : st) {
          Vals hs;
          eval_term(t->head, b, r, hs);
          for (Val& h : hs) add_member(v, h.v, h.s, s.conds.size(), s.quants.size());
        }
        out.push_back({fold(mksv(std::move(v))), s});
        break;
      }
      case Term::ObjComp: {
        States st = eval_body(*t->body, s, r);
        ValuePairs pairs;
        for (State& b : st) {
          Vals ks;
          eval_term(t->head, b, r, ks);
          for (Val& k : ks) {
            Vals vs;
            eval_term(t->head2, k.s, r, vs);
            for (Val& v : vs) {
              if (k.v->kind != SV::CONST || v.v->kind != SV::CONST || v.s.conds.size() != s.conds.size() || v.s.quants.size() != s.quants.size())
                unsupported("object comprehension over review data", t->line);
              pairs.emplace_back(k.v->c, v.v->c);
            }
          }
        }
        out.push_back({sv_const(Value::object(pairs)), s});
        break;
      }
    }
  }

  void eval_seq(const std::vector<TermP>& ts, size_t i, std::vector<SVP> acc, const State& s, const Rule* r,
                const std::function<void(const std::vector<SVP>&, const State&)>& fn) {
    if (i == ts.size()) { fn(acc, s); return; }
    Vals vs;
    eval_term(ts[i], s, r, vs);
    for (Val& v : vs) {
      std::vector<SVP> a2 = acc;
      a2.push_back(v.v);
      eval_seq(ts, i + 1, a2, v.s, r, fn);
    }
  }

  void use_alts(const std::vector<Alt>& alts, const State& s, Vals& out) {
    for (const Alt& a : alts) {
      State n = s;
      QMap m;
      for (auto& q : a.quants) m[q.first] = fresh();
      for (auto& q : a.quants) n.quants.emplace_back(m[q.first], rn_path(q.second, m));
      for (auto& c : a.conds) n.conds.push_back(rn_f(c, m));
      out.push_back({rn_sv(a.v, m), n});
    }
  }

  void eval_var(const TermP& t, const State& s, const Rule* r, Vals& out) {
    if (const SVP* bound = s.env.get(t->name)) { out.push_back({*bound, s}); return; }
    if (t->name == "input") { out.push_back({input_, s}); return; }
    if (t->name == "data") { eval_data_ref({}, s, r, out, t->line); return; }
    if (find_rules(rule_pkg(r), t->name)) { use_alts(rule_alts(rule_pkg(r), t->name), s, out); return; }
    throw UnboundVar("rego_unsafe_var_error: var " + t->name + " is unsafe");
  }

  void eval_ref(const TermP& t, const State& s, const Rule* r, Vals& out) {
    const TermP& head = t->head;
    if (head->kind == Term::Var && !s.env.count(head->name)) {
      if (head->name == "data") { eval_data_ref(t->args, s, r, out, t->line); return; }
      if (const auto* imp = import_of(r, head->name)) {
        std::vector<TermP> ops;
        for (size_t i = 1; i < imp->size(); i++) { Term c; c.kind = Term::Scalar; c.value = Value::string((*imp)[i]); ops.push_back(std::make_shared<const Term>(c)); }
        ops.insert(ops.end(), t->args.begin(), t->args.end());
        if ((*imp)[0] == "data") { eval_data_ref(ops, s, r, out, t->line); return; }
        walk(input_, ops, 0, s, r, out, t->line);
        return;
      }
    }
    Vals hs;
    eval_term(head, s, r, hs);
    for (Val& h : hs) walk(h.v, t->args, 0, h.s, r, out, t->line);
  }

  void eval_data_ref(const std::vector<TermP>& ops, const State& s, const Rule* r, Vals& out, int line) {
    std::vector<std::string> consts;
    for (auto& o : ops) { if (o->kind == Term::Scalar && o->value.is_string()) consts.push_back(o->value.str()); else break; }
    for (size_t n = consts.size(); n-- > 0;) {
      std::string pkg;
      for (size_t i = 0; i < n; i++) { if (i) pkg += "."; pkg += consts[i]; }
      if (find_rules(pkg, consts[n])) {
        Vals vs;
        use_alts(rule_alts(pkg, consts[n]), s, vs);
        for (Val& v : vs) walk(v.v, ops, n + 1, v.s, r, out, line);
        return;
      }
    }
    // symbolic mode: data.inventory is a CONSTANT when the caller hands a snapshot of the synced objects over (engine.cpp: the
    // constraints of a referential template are recompiled when the inventory changes) -- iterating it unrolls into alternatives
    // like an iteration over input.parameters does
    if (!concrete_ && !inventory_.defined()) unsupported("reference to data.* (referential constraint; needs synced inventory)", line);
    ValuePairs root;
    if (inventory_.defined()) root.emplace_back(Value::string("inventory"), inventory_);
    walk(sv_const(Value::object(root)), ops, 0, s, r, out, line);
  }

  // iterate children of `cur`: fn(key-or-null, value, state)
  void iterate(const SVP& cur, const State& s, int line, const std::function<void(const SVP&, const SVP&, const State&)>& fn) {
    switch (cur->kind) {
      case SV::CONST: {
        const Value& c = cur->c;
        if (c.is_array()) { for (size_t i = 0; i < c.size(); i++) fn(sv_const(Value::integer((i128)i)), sv_const(c.items()[i]), s); }
        else if (c.is_set()) { for (auto& x : c.items()) fn(sv_const(x), sv_const(x), s); }
        else if (c.is_object()) { for (auto& kv : c.pairs()) fn(sv_const(kv.first), sv_const(kv.second), s); }
        break;
      }
      case SV::PATH: {
        int q = fresh();
        State n = s;
        n.quants.emplace_back(q, cur->path);
        SV k; k.kind = SV::KEYOF; k.q = q;
        SPath p = cur->path;
        Step st; st.iter = true; st.q = q;
        p.push_back(st);
        fn(mksv(std::move(k)), sv_path(p), n);
        break;
      }
      case SV::OBJ:
        for (auto& f : cur->fields) fn(sv_const(f.first), f.second, s);
        break;
      case SV::ARR: case SV::SET: {
        bool stable = cur->gens.empty();
        for (auto& e : cur->elems) if (e.cond->kind != FNode::T) stable = false;
        size_t idx = 0;
        for (auto& e : cur->elems) {
          State n = s;
          if (e.cond->kind != FNode::T) n.conds.push_back(e.cond);
          SVP key = cur->kind == SV::SET ? e.v : (stable ? sv_const(Value::integer((i128)idx)) : SVP());
          fn(key, e.v, n);
          idx++;
        }
        for (auto& g : cur->gens) {
          QMap m;
          for (int q : g.quants) m[q] = fresh();
          State n = s;
          for (size_t i = 0; i < g.quants.size(); i++) n.quants.emplace_back(m[g.quants[i]], rn_path(g.bases[i], m));
          FP c = rn_f(g.cond, m);
          if (c->kind != FNode::T) n.conds.push_back(c);
          SVP e = rn_sv(g.elem, m);
          fn(cur->kind == SV::SET ? e : SVP(), e, n);
        }
        break;
      }
      case SV::STRX: case SV::COUNTOF: case SV::CARD: unsupported("iteration over a derived value", line);
      default: break;   // scalars: nothing to iterate
    }
  }

  // cur[k]
  void index(const SVP& cur, const SVP& k, const State& s, int line, const std::function<void(const SVP&, const State&)>& fn) {
    switch (cur->kind) {
      case SV::CONST: {
        const Value& c = cur->c;
        if (k->kind == SV::CONST) {
          if (c.is_object()) { const Value* v = c.get(k->c); if (v) fn(sv_const(*v), s); }
          else if (c.is_array()) { if (k->c.is_number() && k->c.is_int && k->c.i >= 0 && (size_t)k->c.i < c.size()) fn(sv_const(c.items()[(size_t)k->c.i]), s); }
          else if (c.is_set()) { if (c.set_has(k->c)) fn(k, s); }
          return;
        }
        if (c.is_set()) {
          FP m = member_f(k, cur);
          if (m->kind == FNode::F) return;
          State n = s;
          if (m->kind != FNode::T) n.conds.push_back(m);
          fn(k, n);
          return;
        }
        if (c.is_object() || c.is_array()) {
          if (c.size() == 0) return;
          unsupported("constant collection indexed by review data", line);
        }
        return;
      }
      case SV::PATH:
        if (k->kind == SV::CONST) {
          if (k->c.is_string()) { SPath p = cur->path; Step st; st.key = k->c.str(); p.push_back(st); fn(sv_path(p), s); return; }
          if (k->c.is_number()) unsupported("numeric index into review data", line);
          return;
        }
        unsupported("review data indexed by a symbolic key", line);
      case SV::OBJ:
        if (k->kind != SV::CONST) unsupported("symbolic key into a composite value", line);
        for (auto& f : cur->fields) if (f.first == k->c) fn(f.second, s);
        return;
      case SV::SET: {
        FP m = member_f(k, cur);
        if (m->kind == FNode::F) return;
        State n = s;
        if (m->kind != FNode::T) n.conds.push_back(m);
        fn(k, n);
        return;
      }
      case SV::ARR: {
        if (k->kind != SV::CONST || !cur->gens.empty()) unsupported("index into a conditional array", line);
        for (auto& e : cur->elems) if (e.cond->kind != FNode::T) unsupported("index into a conditional array", line);
        if (k->c.is_number() && k->c.is_int && k->c.i >= 0 && (size_t)k->c.i < cur->elems.size()) fn(cur->elems[(size_t)k->c.i].v, s);
        return;
      }
      case SV::STRX:
        if (cur->xkind == SV::XARR) {
          SV c = *cur;
          c.xkind = SV::XCOMP;
          if (k->kind == SV::CONST && k->c.is_number() && k->c.is_int) {
            if (k->c.i < 0) return;
            c.idx = (int)k->c.i;
          } else if (k->kind == SV::STRX && k->xkind == SV::XCOUNT && k->idx < 0 && spath_to_string(k->path) == spath_to_string(cur->path) &&
                     k->cut == cur->cut && k->sep == cur->sep) {
            c.idx = k->idx;   // arr[count(arr) - m]
          } else unsupported("symbolic index into split()", line);
          fn(mksv(std::move(c)), s);
          return;
        }
        unsupported("index into a derived string", line);
      default: return;
    }
  }

  void walk(const SVP& cur, const std::vector<TermP>& ops, size_t i, const State& s, const Rule* r, Vals& out, int line) {
    if (i == ops.size()) { out.push_back({cur, s}); return; }
    const TermP& op = ops[i];
    if (op->kind == Term::Var && is_unbound(op, s, r)) {
      bool wild = op->name.compare(0, 2, "$w") == 0;
      iterate(cur, s, line, [&](const SVP& key, const SVP& val, const State& s2) {
        State n = s2;
        if (!wild) { if (!key) unsupported("key of a conditional array", line); n.env.set(op->name, key); }
        walk(val, ops, i + 1, n, r, out, line);
      });
      return;
    }
    if ((op->kind == Term::Array || op->kind == Term::Object) && has_unbound(op, s, r)) {
      iterate(cur, s, line, [&](const SVP& key, const SVP& val, const State& s2) {
        if (!key) unsupported("pattern key over a conditional array", line);
        States st;
        unify_value(op, key, s2, r, st);
        for (State& x : st) walk(val, ops, i + 1, x, r, out, line);
      });
      return;
    }
    Vals ks;
    eval_term(op, s, r, ks);
    for (Val& k : ks) index(cur, k.v, k.s, line, [&](const SVP& nxt, const State& s2) { walk(nxt, ops, i + 1, s2, r, out, line); });
  }

  void eval_binop(const TermP& t, const State& s, const Rule* r, Vals& out) {
    const std::string& op = t->name;
    Vals as;
    eval_term(t->args[0], s, r, as);
    for (Val& a : as) {
      Vals bs;
      eval_term(t->args[1], a.s, r, bs);
      for (Val& b : bs) {
        if (op == "==" || op == "!=" || op == "<" || op == "<=" || op == ">" || op == ">=") {
          FP d = f_and(defined_f(a.v), defined_f(b.v));
          out.push_back({sv_bool(compare_f(a.v, cmp_of(op), b.v), d), b.s});
        } else if (op == "in") {
          out.push_back({sv_bool(member_f(a.v, b.v), f_true()), b.s});
        } else {
          SVP v = arith(op, a.v, b.v, t->line);
          if (v) out.push_back({v, b.s});
        }
      }
    }
  }

  SVP arith(const std::string& op, const SVP& a, const SVP& b, int line) {
    if (a->kind == SV::CONST && b->kind == SV::CONST) {
      Value v = rego_arith(op, a->c, b->c);
      return v.defined() ? sv_const(v) : SVP();
    }
    {   // arithmetic on one review leaf: recorded as an expression of that leaf (dexpr.hpp)
      SPath leaf; std::vector<DX> dx;
      if (!(a->kind == SV::STRX) && !(b->kind == SV::STRX) && same_leaf_args({a, b}, &leaf, &dx)) return sv_derived(leaf, dx_node(DExpr::ARITH, {dx[0], dx[1]}, op));
    }
    if (a->kind == SV::STRX && a->xkind == SV::XCOUNT && b->kind == SV::CONST && b->c.is_number() && b->c.is_int && (op == "-" || op == "+")) {
      SV c = *a;
      c.idx += (int)(op == "-" ? -b->c.i : b->c.i);
      return mksv(std::move(c));
    }
    std::vector<CondElem> ae, be;
    std::vector<Gen> ag, bg;
    bool sa = (a->kind == SV::SET) || (a->kind == SV::CONST && a->c.is_set());
    bool sb = (b->kind == SV::SET) || (b->kind == SV::CONST && b->c.is_set());
    if (sa && sb && (op == "-" || op == "|" || op == "&")) {
      as_setlike(a, &ae, &ag);
      as_setlike(b, &be, &bg);
      SV o;
      o.kind = SV::SET;
      if (op == "|") {
        o.elems = ae; o.elems.insert(o.elems.end(), be.begin(), be.end());
        o.gens = ag; o.gens.insert(o.gens.end(), bg.begin(), bg.end());
        return fold(mksv(std::move(o)));
      }
      bool diff = op == "-";
      for (auto& e : ae) {
        FP m = member_f(e.v, b);
        FP c = f_and(e.cond, diff ? f_not(m) : m);
        if (c->kind != FNode::F) o.elems.push_back({e.v, c});
      }
      for (auto& g : ag) {
        Gen n = g;
        FP m = member_f(g.elem, b);
        n.cond = f_and(g.cond, diff ? f_not(m) : m);
        if (n.cond->kind != FNode::F) o.gens.push_back(n);
      }
      return fold(mksv(std::move(o)));
    }
    unsupported("arithmetic '" + op + "' on review data", line);
  }

  // ---------------------------------------------------------------------------------------------- calls
  void eval_call(const TermP& t, const State& s, const Rule* r, Vals& out) {
    std::string name;
    for (size_t i = 0; i < t->path.size(); i++) { if (i) name += "."; name += t->path[i]; }
    std::string fpkg, fname;
    bool user = false;
    if (t->path.size() == 1 && find_rules(rule_pkg(r), t->path[0])) { user = true; fpkg = rule_pkg(r); fname = t->path[0]; }
    else {
      std::vector<std::string> full = t->path;
      if (const auto* imp = import_of(r, t->path[0])) { full = *imp; full.insert(full.end(), t->path.begin() + 1, t->path.end()); }
      if (full[0] == "data" && full.size() >= 2) {
        std::string pkg;
        for (size_t i = 1; i + 1 < full.size(); i++) { if (i > 1) pkg += "."; pkg += full[i]; }
        if (find_rules(pkg, full.back())) { user = true; fpkg = pkg; fname = full.back(); }
      }
    }
    if (!user && !has_builtin(name)) {
      // (a DISABLED builtin never gets here: the template's static check refuses it when it is added, pe.cpp Template::Template)
      if (is_opa_builtin(name) || name == "http.send") unsupported("builtin " + name + " is not implemented by this engine", t->line);
      throw RegoError("rego_type_error: undefined function " + name);
    }
    eval_seq(t->args, 0, {}, s, r, [&](const std::vector<SVP>& args, const State& s2) {
      if (user) {
        if (concrete_) { call_function(fpkg, fname, args, s2, r, out); return; }
        const size_t mark = out.size();
        try { call_function(fpkg, fname, args, s2, r, out); return; }
        catch (const Unsupported&) {
          out.resize(mark);
          if (!deep_call(fpkg, fname, args, s2, out)) throw;   // (a closed helper over ONE sub-document goes to the flattener: dexpr.hpp)
          return;
        }
      }
      bool all_const = true;
      for (auto& a : args) if (a->kind != SV::CONST) all_const = false;
      if (all_const) {
        ValueVec av;
        for (auto& a : args) av.push_back(a->c);
        Value v = call_builtin(name, av);
        if (v.defined()) out.push_back({sv_const(v), s2});
        return;
      }
      // fold composite args that are in fact constant
      std::vector<SVP> fa;
      for (auto& a : args) fa.push_back(fold(a));
      bool folded_const = true;
      for (auto& a : fa) if (a->kind != SV::CONST) folded_const = false;
      if (folded_const) {   // e.g. an object / array literal argument of constants
        ValueVec av;
        for (auto& a : fa) av.push_back(a->c);
        Value v = call_builtin(name, av);
        if (v.defined()) out.push_back({sv_const(v), s2});
        return;
      }
      symbolic_builtin(name, fa, s2, t->line, out);
    });
  }

  FP all_defined(const SVP& v) { return defined_f(v); }

  // ---------------------------------------------------------------------------------------------- deep calls (dexpr.hpp)
  // Is the function CLOSED: do its bodies (and everything they call or refer to in the package) read nothing but their
  // arguments?  No `input`, no `data`, no rule that does.
  bool closed_term(const TermP& t, const std::string& pkg, std::set<std::string>& visiting) {
    if (!t) return true;
    if (t->kind == Term::Var) {
      if (t->name == "input" || t->name == "data") return false;
      if (find_rules(pkg, t->name)) return closed_rules(pkg, t->name, visiting);
      return true;
    }
    if (t->kind == Term::Call) {
      std::string name;
      for (size_t i = 0; i < t->path.size(); i++) { if (i) name += "."; name += t->path[i]; }
      if (t->path.size() == 1 && find_rules(pkg, t->path[0])) { if (!closed_rules(pkg, t->path[0], visiting)) return false; }
      else if (!has_builtin(name)) return false;   // (a function of another package, an unknown builtin: not our business here)
    }
    if (!closed_term(t->head, pkg, visiting) || !closed_term(t->head2, pkg, visiting)) return false;
    for (auto& a : t->args) if (!closed_term(a, pkg, visiting)) return false;
    if (t->body) for (const Literal& l : *t->body) if (!closed_literal(l, pkg, visiting)) return false;
    return true;
  }
  bool closed_literal(const Literal& l, const std::string& pkg, std::set<std::string>& visiting) {
    if (!closed_term(l.a, pkg, visiting) || !closed_term(l.b, pkg, visiting) || !closed_term(l.c, pkg, visiting)) return false;
    if (l.inner && !closed_literal(*l.inner, pkg, visiting)) return false;
    if (l.body) for (const Literal& x : *l.body) if (!closed_literal(x, pkg, visiting)) return false;
    return true;
  }
  bool closed_rules(const std::string& pkg, const std::string& name, std::set<std::string>& visiting) {
    if (!visiting.insert(name).second) return true;   // (already being checked further up: recursion decides nothing new)
    const auto* rules = find_rules(pkg, name);
    if (!rules) return false;
    for (const Rule* r : *rules) {
      for (auto& a : r->args) if (!closed_term(a, pkg, visiting)) return false;
      if (!closed_term(r->key, pkg, visiting) || !closed_term(r->value, pkg, visiting)) return false;
      for (const Literal& l : r->body) if (!closed_literal(l, pkg, visiting)) return false;
      for (auto& e : r->elses) { if (!closed_term(e.first, pkg, visiting)) return false; for (const Literal& l : e.second) if (!closed_literal(l, pkg, visiting)) return false; }
    }
    return true;
  }
  // The constant key prefix under which EVERY use of variable `var` in the function reads (obj.spec.selector[key] -> ["spec",
  // "selector"]); `whole` when the variable is also used as such (passed on, compared, iterated directly).
  static void narrow_term(const TermP& t, const std::string& var, std::vector<std::string>* prefix, bool* have, bool* whole) {
    if (!t || *whole) return;
    if (t->kind == Term::Var) { if (t->name == var) *whole = true; return; }
    if (t->kind == Term::Ref && t->head && t->head->kind == Term::Var && t->head->name == var) {
      std::vector<std::string> keys;
      for (auto& op : t->args) { if (op->kind == Term::Scalar && op->value.is_string()) keys.push_back(op->value.str()); else break; }
      if (!*have) { *prefix = keys; *have = true; }
      else { size_t n = 0; while (n < prefix->size() && n < keys.size() && (*prefix)[n] == keys[n]) n++; prefix->resize(n); }
      for (auto& op : t->args) narrow_term(op, var, prefix, have, whole);
      return;
    }
    narrow_term(t->head, var, prefix, have, whole);
    narrow_term(t->head2, var, prefix, have, whole);
    for (auto& a : t->args) narrow_term(a, var, prefix, have, whole);
    if (t->body) for (const Literal& l : *t->body) narrow_literal(l, var, prefix, have, whole);
  }
  static void narrow_literal(const Literal& l, const std::string& var, std::vector<std::string>* prefix, bool* have, bool* whole) {
    narrow_term(l.a, var, prefix, have, whole); narrow_term(l.b, var, prefix, have, whole); narrow_term(l.c, var, prefix, have, whole);
    if (l.inner) narrow_literal(*l.inner, var, prefix, have, whole);
    if (l.body) for (const Literal& x : *l.body) narrow_literal(x, var, prefix, have, whole);
    for (auto& n : l.names) if (n == var) *whole = true;   // (`some obj`: the name is rebound -- give up narrowing)
  }
  // f(.., <one review sub-document>, ..) of a closed helper that the formula language cannot express: a value DERIVED from that
  // sub-document by the flattener (the concrete evaluator runs the helper on the real value, once per distinct value).
  bool deep_call(const std::string& pkg, const std::string& name, const std::vector<SVP>& args, const State& s, Vals& out) {
    int sym = -1;
    for (size_t i = 0; i < args.size(); i++) {
      if (args[i]->kind == SV::CONST) { if (!args[i]->c.defined()) return false; continue; }
      if (args[i]->kind != SV::PATH || args[i]->path.empty() || sym >= 0) return false;
      sym = (int)i;
    }
    if (sym < 0) return false;
    std::set<std::string> visiting;
    if (!closed_rules(pkg, name, visiting)) return false;
    std::shared_ptr<const Template> keep;
    try { keep = T.shared_from_this(); } catch (const std::bad_weak_ptr&) { return false; }   // (a template on somebody's stack: no closure may outlive it)
    // how far down the argument does the helper look?  Every body must name the argument by a variable
    std::vector<std::string> prefix;
    bool have = false, whole = false;
    const auto* rules = find_rules(pkg, name);
    for (const Rule* r : *rules) {
      if (r->kind != Rule::Function || r->args.size() != args.size()) continue;
      const TermP& a = r->args[(size_t)sym];
      if (a->kind != Term::Var) { whole = true; break; }
      narrow_term(r->value, a->name, &prefix, &have, &whole);
      for (const Literal& l : r->body) narrow_literal(l, a->name, &prefix, &have, &whole);
      for (auto& e : r->elses) { narrow_term(e.first, a->name, &prefix, &have, &whole); for (const Literal& l : e.second) narrow_literal(l, a->name, &prefix, &have, &whole); }
    }
    if (whole || !have) prefix.clear();
    char idbuf[40];
    snprintf(idbuf, sizeof idbuf, "%llx", (unsigned long long)(uintptr_t)keep.get());
    const std::string fn_name = std::string("$u:") + idbuf + ":" + pkg + "." + name + "/" + std::to_string(args.size());
    const std::string fpkg = pkg, fname = name;
    // (the closure must not keep the template alive -- the registry outlives every engine --: a weak reference; the template's
    //  destructor takes the entry out again, and an expression of a template that is gone answers "undefined")
    const std::weak_ptr<const Template> weak = keep;
    if (std::find(T.deep_fns_.begin(), T.deep_fns_.end(), fn_name) == T.deep_fns_.end()) T.deep_fns_.push_back(fn_name);
    dx_register_user(fn_name, [weak, fpkg, fname](const ValueVec& av) -> Value {
      try {
        std::shared_ptr<const Template> keep = weak.lock();
        if (!keep) return Value();
        int nq = 0;
        PE pe(*keep, Value::object({}), sv_const(Value::object({})), Value(), true, &nq);
        pe.index_rules();
        std::vector<SVP> cargs;
        for (const Value& v : av) cargs.push_back(sv_const(v));
        Vals res;
        pe.call_function(fpkg, fname, cargs, State(), nullptr, res);
        Value got;
        for (const Val& x : res) {
          if (x.v->kind != SV::CONST || !x.v->c.defined()) continue;
          if (got.defined() && !(got == x.v->c)) return Value();   // (conflicting outputs: an evaluation error in OPA -- no value here)
          got = x.v->c;
        }
        return got;
      } catch (const std::exception&) { return Value(); }
    });
    std::vector<DX> dxs;
    ValueVec absent_args;
    for (size_t i = 0; i < args.size(); i++) {
      if ((int)i != sym) { dxs.push_back(dx_const(args[i]->c)); absent_args.push_back(args[i]->c); continue; }
      if (prefix.empty()) { dxs.push_back(dx_leaf()); absent_args.push_back(Value()); continue; }
      ValueVec keys;
      for (auto& k : prefix) keys.push_back(Value::string(k));
      dxs.push_back(dx_node(DExpr::CALL, {dx_leaf(), dx_const(Value::array(keys))}, "$wrap"));
      absent_args.push_back(Value::object({}));   // the helper reads nothing else of its argument: without the sub-document it sees an empty object
    }
    SV d;
    d.kind = SV::DERIVED;
    d.path = args[(size_t)sym]->path;
    for (auto& k : prefix) { Step st; st.key = k; d.path.push_back(st); }
    d.dx = dx_node(DExpr::CALL, dxs, fn_name);
    if (!prefix.empty()) {
      d.idx = 1;
      d.c = dx_call_user(fn_name, absent_args);
      d.f = defined_f(args[(size_t)sym]);
    }
    out.push_back({mksv(std::move(d)), s});
    return true;
  }

  void symbolic_builtin(const std::string& name, const std::vector<SVP>& a, const State& s, int line, Vals& out) {
    auto need = [&](size_t n) { if (a.size() != n) throw RegoError("rego_type_error: " + name + ": arity mismatch"); };
    auto push_bool = [&](FP t, FP d) { SVP v = sv_bool(t, d); if (!(v->kind == SV::CONST && !v->c.defined())) out.push_back({v, s}); };
    if (name == "print" || name == "trace") { out.push_back({sv_const(Value::boolean(true)), s}); return; }
    if (name == "sprintf") {
      need(2);
      if (a[0]->kind != SV::CONST || !a[0]->c.is_string()) unsupported("sprintf with a symbolic format", line);
      SV o; o.kind = SV::OPAQUE; o.f = defined_f(a[1]);
      if (a[1]->kind == SV::PATH) o.f = f_type(a[1]->path, M_ARRAY);
      // A format of literal text and %v / %s verbs over an array literal of constants and review leaves keeps its PARTS: such a
      // string can still be compared with a constant (K8sUniqueLabel's make_apiversion: sprintf("%v/%v", [g, v]) == obj.apiVersion)
      if (a[1]->kind == SV::ARR && a[1]->gens.empty()) fmt_parts(*a[0]->c.s, a[1]->elems, &o.elems);
      if (a[1]->kind == SV::ARR && a[1]->gens.empty()) fmt_head(*a[0]->c.s, a[1]->elems, &o);
      out.push_back({mksv(std::move(o)), s});
      return;
    }
    if (name == "array.concat") {   // of array values built from review data / conditional members: members of a, then of b
      need(2);
      std::vector<CondElem> ae, be;
      std::vector<Gen> ag, bg;
      const bool arr_a = a[0]->kind == SV::ARR || (a[0]->kind == SV::CONST && a[0]->c.is_array());
      const bool arr_b = a[1]->kind == SV::ARR || (a[1]->kind == SV::CONST && a[1]->c.is_array());
      if (arr_a && arr_b && as_setlike(a[0], &ae, &ag) && as_setlike(a[1], &be, &bg)) {
        SV o; o.kind = SV::ARR;
        o.elems = ae; o.elems.insert(o.elems.end(), be.begin(), be.end());
        o.gens = ag; o.gens.insert(o.gens.end(), bg.begin(), bg.end());
        out.push_back({fold(mksv(std::move(o))), s});
        return;
      }
      if ((a[0]->kind == SV::CONST && !a[0]->c.is_array()) || (a[1]->kind == SV::CONST && !a[1]->c.is_array())) return;   // not an array: undefined
      unsupported("array.concat with these operands on review data", line);
    }
    if (name == "count") {
      need(1);
      const SVP& x = a[0];
      if (x->kind == SV::PATH) { SV c; c.kind = SV::COUNTOF; c.path = x->path; out.push_back({mksv(std::move(c)), s}); return; }
      if (x->kind == SV::STRX && x->xkind == SV::XARR) { SV c = *x; c.xkind = SV::XCOUNT; c.idx = 0; out.push_back({mksv(std::move(c)), s}); return; }
      if (x->kind == SV::SET || x->kind == SV::ARR) {
        SV c; c.kind = SV::CARD; c.gens = x->gens; c.idx = x->kind == SV::ARR ? 1 : 0;
        if (x->kind == SV::ARR) c.elems = x->elems;
        else {   // merge equal constant members
          for (auto& e : x->elems) {
            bool merged = false;
            if (e.v->kind == SV::CONST) for (auto& o : c.elems) if (o.v->kind == SV::CONST && o.v->c == e.v->c) { o.cond = f_or(o.cond, e.cond); merged = true; break; }
            if (!merged) c.elems.push_back(e);
          }
        }
        out.push_back({mksv(std::move(c)), s});
        return;
      }
      if (x->kind == SV::DERIVED && x->idx != 1) { out.push_back({sv_derived(x->path, dx_node(DExpr::CALL, {x->dx}, "count")), s}); return; }
      unsupported("count() of this symbolic value", line);
    }
    if (name == "startswith" || name == "endswith" || name == "contains") {
      need(2);
      if (a[0]->kind == SV::PATH && a[1]->kind == SV::CONST) {
        if (!a[1]->c.is_string()) return;
        Atom at = atom_path(name == "startswith" ? Atom::STR_PREFIX : name == "endswith" ? Atom::STR_SUFFIX : Atom::STR_CONTAINS, a[0]->path);
        at.k = a[1]->c;
        push_bool(f_atom(at), f_type(a[0]->path, M_STRING));
        return;
      }
      if (a[0]->kind == SV::KEYOF && a[1]->kind == SV::CONST) {   // a string test on the member NAME of a key iteration
        if (!a[1]->c.is_string()) return;
        Atom t; t.kind = Atom::KEYCMP; t.q = a[0]->q; t.cmp = name == "startswith" ? KC_PREFIX : name == "endswith" ? KC_SUFFIX : KC_CONTAINS; t.k = a[1]->c;
        Atom d; d.kind = Atom::KEYCMP; d.q = a[0]->q; d.cmp = KC_ISNAME; d.k = Value::string("");   // an array index is a number: the builtin is undefined for it
        push_bool(f_atom(t), f_atom(d));
        return;
      }
      { SPath leaf; std::vector<DX> dx; if (same_leaf_args(a, &leaf, &dx)) { out.push_back({sv_derived(leaf, dx_node(DExpr::CALL, dx, name)), s}); return; } }
      unsupported(name + " with these symbolic operands", line);
    }
    if (name == "re_match" || name == "regex.match") {
      need(2);
      // constant pattern, review string: a DFA predicate on the device.  An invalid pattern or a non-string operand is a
      // builtin error, i.e. the expression is undefined (b_re_match in builtins.cpp mirrors this for concrete values).
      if (a[0]->kind == SV::CONST && a[1]->kind == SV::PATH) {
        if (!a[0]->c.is_string() || !get_regex(a[0]->c.str())) return;
        Atom at = atom_path(Atom::STR_REGEX, a[1]->path);
        at.k = a[0]->c;
        push_bool(f_atom(at), f_type(a[1]->path, M_STRING));
        return;
      }
      { SPath leaf; std::vector<DX> dx; if (same_leaf_args(a, &leaf, &dx)) { out.push_back({sv_derived(leaf, dx_node(DExpr::CALL, dx, name)), s}); return; } }
      unsupported(name + " with these symbolic operands", line);
    }
    if (name == "strings.any_prefix_match" || name == "strings.any_suffix_match") {
      need(2);
      if (a[0]->kind == SV::PATH && a[1]->kind == SV::CONST) {
        std::vector<Value> pats;
        if (a[1]->c.is_string()) pats.push_back(a[1]->c);
        else if (a[1]->c.is_array() || a[1]->c.is_set()) { for (auto& x : a[1]->c.items()) { if (!x.is_string()) return; pats.push_back(x); } }
        else return;
        FP t = f_false();
        for (auto& p : pats) { Atom at = atom_path(name == "strings.any_prefix_match" ? Atom::STR_PREFIX : Atom::STR_SUFFIX, a[0]->path); at.k = p; t = f_or(t, f_atom(at)); }
        push_bool(t, f_type(a[0]->path, M_STRING));
        return;
      }
      unsupported(name + " with these symbolic operands", line);
    }
    if (name == "is_string" || name == "is_number" || name == "is_boolean" || name == "is_array" || name == "is_object" || name == "is_null" || name == "is_set") {
      need(1);
      uint32_t m = name == "is_string" ? M_STRING : name == "is_number" ? M_NUMBER : name == "is_boolean" ? M_BOOL : name == "is_array" ? M_ARRAY : name == "is_object" ? M_OBJECT : name == "is_null" ? M_NULL : 0;
      if (a[0]->kind == SV::PATH) { push_bool(f_type(a[0]->path, m), f_atom(atom_path(Atom::DEFINED, a[0]->path))); return; }
      if (a[0]->kind == SV::OPAQUE || a[0]->kind == SV::STRX) { push_bool(name == "is_string" ? f_true() : f_false(), defined_f(a[0])); return; }
      if (a[0]->kind == SV::DERIVED && a[0]->idx != 1) { push_bool(f_dict(a[0]->path, dx_node(DExpr::TYPE_MASK, {a[0]->dx}, "", 0, m)), defined_f(a[0])); return; }
      if (a[0]->kind == SV::SET) { push_bool(name == "is_set" ? f_true() : f_false(), f_true()); return; }
      if (a[0]->kind == SV::ARR) { push_bool(name == "is_array" ? f_true() : f_false(), f_true()); return; }
      unsupported(name + " of this symbolic value", line);
    }
    if (name == "any" || name == "all") {
      need(1);
      std::vector<CondElem> elems;
      std::vector<Gen> gens;
      if (!as_setlike(a[0], &elems, &gens)) unsupported(name + " of a non-collection", line);
      if (!gens.empty()) unsupported(name + " over a comprehension of review data", line);
      auto is_true = [&](const SVP& v) -> FP {
        if (v->kind == SV::CONST) return (v->c.is_bool() && v->c.b) ? f_true() : f_false();
        if (v->kind == SV::BOOLF) return v->f;
        if (v->kind == SV::PATH) { Atom c = atom_path(Atom::CMP, v->path); c.cmp = C_EQ; c.k = Value::boolean(true); return f_atom(c); }
        return f_false();
      };
      FP t = name == "any" ? f_false() : f_true();
      for (auto& e : elems) {
        if (name == "any") t = f_or(t, f_and(e.cond, is_true(e.v)));
        else t = f_and(t, f_or(f_not(e.cond), is_true(e.v)));
      }
      push_bool(t, f_true());
      return;
    }
    if (name == "trim") {
      need(2);
      if (a[0]->kind == SV::PATH && a[1]->kind == SV::CONST && a[1]->c.is_string() && a[1]->c.str().size() == 1) {
        SV x; x.kind = SV::STRX; x.path = a[0]->path; x.cut = a[1]->c.str()[0]; x.xkind = SV::XTRIM;
        out.push_back({mksv(std::move(x)), s});
        return;
      }
      { SPath leaf; std::vector<DX> dx; if (same_leaf_args(a, &leaf, &dx)) { out.push_back({sv_derived(leaf, dx_node(DExpr::CALL, dx, name)), s}); return; } }
      unsupported("trim() with these operands on review data", line);
    }
    if (name == "split") {
      need(2);
      if (a[1]->kind == SV::CONST && a[1]->c.is_string() && a[1]->c.str().size() == 1) {
        SV x; x.kind = SV::STRX; x.sep = a[1]->c.str()[0]; x.xkind = SV::XARR;
        if (a[0]->kind == SV::PATH) { x.path = a[0]->path; out.push_back({mksv(std::move(x)), s}); return; }
        if (a[0]->kind == SV::STRX && a[0]->xkind == SV::XTRIM) { x.path = a[0]->path; x.cut = a[0]->cut; out.push_back({mksv(std::move(x)), s}); return; }
      }
      unsupported("split() with these operands on review data", line);
    }
    if (name == "object.get") {
      need(3);
      if (a[0]->kind == SV::PATH && a[1]->kind == SV::CONST && a[1]->c.is_string()) {
        SPath p = a[0]->path; Step st; st.key = a[1]->c.str(); p.push_back(st);
        FP d = f_atom(atom_path(Atom::DEFINED, p));
        FP isobj = f_type(a[0]->path, M_OBJECT);
        State s1 = s; s1.conds.push_back(d);
        out.push_back({sv_path(p), s1});
        State s2 = s; s2.conds.push_back(f_and(isobj, f_not(d)));
        out.push_back({a[2], s2});
        return;
      }
      if (a[0]->kind == SV::OBJ && a[1]->kind == SV::CONST) {   // e.g. object.get(input, "parameters", {}): constant keys
        for (auto& f : a[0]->fields) if (f.first == a[1]->c) { out.push_back({f.second, s}); return; }
        out.push_back({a[2], s});
        return;
      }
      if (a[0]->kind == SV::PATH && a[1]->kind == SV::CONST && a[1]->c.is_array()) {   // object.get(obj, ["a", "b"], default): the value at the path, else the default
        SPath p = a[0]->path;
        bool keys_ok = true;
        for (auto& k : a[1]->c.items()) { if (!k.is_string()) { keys_ok = false; break; } Step st; st.key = k.str(); p.push_back(st); }
        if (keys_ok) {
          if (a[1]->c.size() == 0) { out.push_back({a[0], s}); return; }
          FP d = f_atom(atom_path(Atom::DEFINED, p));
          State s1 = s; s1.conds.push_back(d);
          out.push_back({sv_path(p), s1});
          State s2 = s; s2.conds.push_back(f_and(f_type(a[0]->path, M_OBJECT), f_not(d)));
          out.push_back({a[2], s2});
          return;
        }
      }
      unsupported("object.get with these operands on review data", line);
    }
    if (name == "type_name") {   // total on defined operands; the result is only good for messages (comparing it is refused where it is compared)
      need(1);
      SV o; o.kind = SV::OPAQUE; o.f = defined_f(a[0]);
      out.push_back({mksv(std::move(o)), s});
      return;
    }
    if (name == "concat" && a.size() == 2 && a[0]->kind == SV::CONST && a[0]->c.is_string() && a[1]->kind == SV::ARR && a[1]->gens.empty()) {
      // concat(sep, [x, y, ..]) over an array literal: defined iff every element is a string; an opaque string otherwise
      FP d = f_true();
      bool plain = true;
      for (auto& e : a[1]->elems) { if (e.cond && e.cond->kind != FNode::T) { plain = false; break; } d = f_and(d, is_string_f(e.v)); }
      if (plain) {
        SV o; o.kind = SV::OPAQUE; o.f = d;
        out.push_back({mksv(std::move(o)), s});
        return;
      }
    }
    {   // a scalar builtin whose symbolic operands all derive from ONE leaf: an expression of that leaf.  Only builtins
        // that look at nothing but a scalar's value or a container's size: the flattener evaluates the expression on the
        // leaf's value, and hands a container leaf over as a placeholder of the same type and size
      static const std::set<std::string> scalar_fns = {"to_number", "replace", "substring", "lower", "upper", "trim", "trim_space", "trim_left", "trim_right",
          "trim_prefix", "trim_suffix", "startswith", "endswith", "contains", "re_match", "regex.match", "indexof", "abs", "round", "ceil", "floor", "format_int",
          "concat" /* of a constant separator and the split() of the leaf */};
      SPath leaf; std::vector<DX> dx;
      if (scalar_fns.count(name) && same_leaf_args(a, &leaf, &dx)) { out.push_back({sv_derived(leaf, dx_node(DExpr::CALL, dx, name)), s}); return; }
    }
    unsupported("builtin " + name + " applied to review data", line);
  }

 public:
  void index_rules() {}   // (the rule -> package / module index lives in the Template: built once, not per evaluation)
};

FP PE::defined_f(const SVP& v) {
  switch (v->kind) {
    case SV::CONST: return v->c.defined() ? f_true() : f_false();
    case SV::PATH: {
      if (!v->path.empty() && v->path.back().iter) return f_true();
      if (v->path.empty()) return f_true();
      return f_atom(atom_path(Atom::DEFINED, v->path));
    }
    case SV::KEYOF: return f_true();
    case SV::OBJ: { FP r = f_true(); for (auto& f : v->fields) r = f_and(r, defined_f(f.second)); return r; }
    case SV::ARR: {
      // an array LITERAL with symbolic members is defined iff all members are; comprehension results carry conds
      FP r = f_true();
      for (auto& e : v->elems) if (e.cond->kind == FNode::T) r = f_and(r, defined_f(e.v));
      return r;
    }
    case SV::SET: case SV::CARD: return f_true();
    case SV::OPAQUE: return v->f;
    case SV::BOOLF: return v->d;
    case SV::COUNTOF: return f_type(v->path, M_ARRAY | M_OBJECT | M_STRING);
    case SV::DERIVED: {
      FP present = f_dict(v->path, dx_node(DExpr::DEFINED, {v->dx}));
      if (v->idx == 1 && v->c.defined()) return f_or(present, f_and(v->f, f_not(f_atom(atom_path(Atom::DEFINED, v->path)))));
      return present;
    }
    case SV::STRX: {
      if (v->xkind == SV::XCOMP) {
        Atom c = atom_path(Atom::SPLIT_COUNT, v->path);
        c.cut = v->cut; c.sep = v->sep;
        if (v->idx >= 0) { c.cmp = C_GT; c.k = Value::integer(v->idx); }
        else { c.cmp = C_GE; c.k = Value::integer(-v->idx); }
        return f_atom(c);
      }
      return f_type(v->path, M_STRING);
    }
  }
  return f_true();
}

FP PE::truthy_f(const SVP& v) {
  switch (v->kind) {
    case SV::CONST: return (v->c.defined() && !(v->c.is_bool() && !v->c.b)) ? f_true() : f_false();
    case SV::PATH: return v->path.empty() ? f_true() : f_atom(atom_path(Atom::TRUTHY, v->path));
    case SV::BOOLF: return f_and(v->d, v->f);
    case SV::DERIVED: {
      FP present = f_dict(v->path, dx_node(DExpr::TRUTHY, {v->dx}));
      if (v->idx == 1 && v->c.defined() && !(v->c.is_bool() && !v->c.b)) return f_or(present, f_and(v->f, f_not(f_atom(atom_path(Atom::DEFINED, v->path)))));
      return present;
    }
    default: return defined_f(v);
  }
}

FP PE::is_string_f(const SVP& v) {
  switch (v->kind) {
    case SV::CONST: return v->c.is_string() ? f_true() : f_false();
    case SV::PATH: return f_type(v->path, M_STRING);
    case SV::KEYOF: {   // the key of an iteration: a member NAME is a string, an array index a number
      Atom d; d.kind = Atom::KEYCMP; d.q = v->q; d.cmp = KC_ISNAME; d.k = Value::string("");
      return f_atom(d);
    }
    case SV::OPAQUE: return v->f;
    case SV::STRX: return v->xkind == SV::XCOUNT ? f_false() : defined_f(v);
    case SV::DERIVED: {
      FP present = f_dict(v->path, dx_node(DExpr::TYPE_MASK, {v->dx}, "", 0, M_STRING));
      if (v->idx == 1 && v->c.is_string()) return f_or(present, f_and(v->f, f_not(f_atom(atom_path(Atom::DEFINED, v->path)))));
      return present;
    }
    default: return f_false();
  }
}

// ================================================================================================ Template
Template::~Template() { for (const std::string& n : deep_fns_) dx_unregister_user(n); drop_cindex(); }
Template::Template(const std::string& rego, const std::vector<std::string>& libs, const std::set<std::string>* disabled_in) {
  static const std::set<std::string> k_default_disabled = {"http.send"};
  const std::set<std::string>& disabled = disabled_in ? *disabled_in : k_default_disabled;
  modules_.push_back(parse_rego(rego));
  for (auto& l : libs) {
    modules_.push_back(parse_rego(l));
    if (modules_.back().package.empty() || modules_.back().package[0] != "lib")
      throw RegoError("invalid rego: libs must be declared under package lib");
  }
  for (const Module& m : modules_) {
    std::string pkg;
    for (size_t i = 0; i < m.package.size(); i++) { if (i) pkg += "."; pkg += m.package[i]; }
    if (&m == &modules_[0]) pkg_name_ = pkg;
    for (const Rule& r : m.rules) { rules_[{pkg, r.name}].push_back(&r); rule_pkgs_[&r] = pkg; rule_mods_[&r] = &m; }
  }
  if (!rules_.count({pkg_name_, "violation"})) throw RegoError("invalid rego: missing required rule violation");
  // static safety check + data usage scan
  std::function<void(const TermP&, std::set<std::string>&, bool&)> scan = [&](const TermP& t, std::set<std::string>& vars, bool& data) {
    if (!t) return;
    if (t->kind == Term::Var) { vars.insert(t->name); if (t->name == "data") data = true; }
    if (t->head) scan(t->head, vars, data);
    if (t->head2) scan(t->head2, vars, data);
    for (auto& a : t->args) scan(a, vars, data);
  };
  for (const Module& m : modules_)
    for (const Rule& r : m.rules) {
      std::function<void(const Body&)> sb = [&](const Body& b) {
        for (const Literal& l : b) {
          std::set<std::string> vars;
          const Literal* cur = &l;
          while (cur->kind == Literal::Not) cur = cur->inner.get();
          scan(cur->a, vars, uses_data_); scan(cur->b, vars, uses_data_); scan(cur->c, vars, uses_data_);
        }
      };
      sb(r.body);
      for (auto& e : r.elses) sb(e.second);
      std::set<std::string> vars;
      scan(r.key, vars, uses_data_);
      scan(r.value, vars, uses_data_);
    }
  // every called name resolves at AddTemplate time, as OPA's compiler has it: a rule of this template, a builtin this
  // engine implements, a builtin OPA defines (valid Rego that this engine refuses: unsupported) -- anything else is
  // `rego_type_error: undefined function` whether or not an evaluation would reach the call
  for (const Module& m : modules_) {
    std::string pkg;
    for (size_t i = 0; i < m.package.size(); i++) { if (i) pkg += "."; pkg += m.package[i]; }
    std::function<void(const TermP&)> calls;
    std::function<void(const Body&)> calls_body = [&](const Body& b) {
      for (const Literal& l : b) {
        const Literal* cur = &l;
        while (cur->kind == Literal::Not) cur = cur->inner.get();
        calls(cur->a); calls(cur->b); calls(cur->c);
        if (cur->body) calls_body(*cur->body);
      }
    };
    calls = [&](const TermP& t) {
      if (!t) return;
      calls(t->head); calls(t->head2);
      for (auto& a : t->args) calls(a);
      if (t->body) calls_body(*t->body);
      if (t->kind != Term::Call || t->path.empty()) return;
      std::string name;
      for (size_t i = 0; i < t->path.size(); i++) { if (i) name += "."; name += t->path[i]; }
      if (t->path.size() == 1 && rules_.count({pkg, t->path[0]})) return;
      std::vector<std::string> full = t->path;
      for (auto& imp : m.imports) if (imp.second == t->path[0]) { full = imp.first; full.insert(full.end(), t->path.begin() + 1, t->path.end()); break; }
      if (full[0] == "data" && full.size() >= 2) {
        std::string fp;
        for (size_t i = 1; i + 1 < full.size(); i++) { if (i > 1) fp += "."; fp += full[i]; }
        if (rules_.count({fp, full.back()})) return;
      }
      if (disabled.count(name)) throw RegoError("rego_type_error: undefined function " + name);   // (rego.DisableBuiltins: the capability is gone from the compiler)
      if (has_builtin(name)) return;
      if (is_opa_builtin(name) || name == "http.send") throw Unsupported("unsupported on the device plan: builtin " + name + " is not implemented by this engine (line " + std::to_string(t->line) + ")");
      throw RegoError("rego_type_error: undefined function " + name);
    };
    for (const Rule& r : m.rules) {
      for (auto& a : r.args) calls(a);
      calls(r.key); calls(r.value);
      calls_body(r.body);
      for (auto& e : r.elses) { calls(e.first); calls_body(e.second); }
    }
  }
  // a full smoke evaluation with empty inputs surfaces unsafe variables at AddTemplate time (after the static check of the called names: a disabled builtin is
  // `undefined function` whether or not an evaluation would reach it)
  try {
    render(Value::object({}), Value::object({}), Value());
  } catch (const UnboundVar& e) {
    throw RegoError(e.what());
  }
  // the scan above does not descend into comprehension bodies; a textual check is a sound over-approximation
  if (!uses_data_) {
    for (const std::string* src : {&rego}) if (src->find("data.inventory") != std::string::npos) uses_data_ = true;
    for (auto& l : libs) if (l.find("data.inventory") != std::string::npos) uses_data_ = true;
  }
}

FP Template::compile(const Value& parameters, int* next_quant, const Value& inventory) const {
  PE pe(*this, parameters, sv_path({}), inventory, false, next_quant);
  pe.index_rules();
  SVP set;
  try {
    set = pe.violation_set();
  } catch (const UnboundVar& e) {
    throw RegoError(e.what());
  }
  std::vector<CondElem> elems;
  std::vector<Gen> gens;
  if (set->kind == SV::CONST) { for (auto& x : set->c.items()) elems.push_back({sv_const(x), f_true()}); }
  else { elems = set->elems; gens = set->gens; }
  auto valid = [&](const SVP& e) -> FP {
    if (e->kind == SV::CONST) {
      const Value* m = e->c.get("msg");
      return (e->c.is_object() && m && m->is_string()) ? f_true() : f_false();
    }
    if (e->kind != SV::OBJ) return f_false();
    SVP msg;
    for (auto& f : e->fields) if (f.first == Value::string("msg")) msg = f.second;
    if (!msg) return f_false();
    return f_and(pe.defined_f(e), pe.is_string_f(msg));
  };
  FP r = f_false();
  for (auto& e : elems) r = f_or(r, f_and(e.cond, valid(e.v)));
  for (auto& g : gens) {
    FP body = f_and(g.cond, valid(g.elem));
    for (size_t i = g.quants.size(); i-- > 0;) body = f_exists(g.quants[i], g.bases[i], body);
    r = f_or(r, body);
  }
  return r;
}

FP Template::compile_multi(const Value& parameters, int* next_quant, const Value& inventory) const {
  PE pe(*this, parameters, sv_path({}), inventory, false, next_quant);
  pe.index_rules();
  SVP set;
  try {
    set = pe.violation_set();
  } catch (const UnboundVar& e) {
    throw RegoError(e.what());
  }
  std::vector<CondElem> elems;
  std::vector<Gen> gens;
  if (set->kind == SV::CONST) { for (auto& x : set->c.items()) elems.push_back({sv_const(x), f_true()}); }
  else { elems = set->elems; gens = set->gens; }
  auto valid = [&](const SVP& e) -> FP {
    if (e->kind == SV::CONST) {
      const Value* m = e->c.get("msg");
      return (e->c.is_object() && m && m->is_string()) ? f_true() : f_false();
    }
    if (e->kind != SV::OBJ) return f_false();
    SVP msg;
    for (auto& f : e->fields) if (f.first == Value::string("msg")) msg = f.second;
    if (!msg) return f_false();
    return f_and(pe.defined_f(e), pe.is_string_f(msg));
  };
  // one BRANCH per member expression of the set: `any` = it yields a result, `two` = it may yield two (two bindings of its
  // review iterations; a branch without iterations yields at most one)
  struct Branch { FP any, two; };
  std::vector<Branch> br;
  for (auto& e : elems) {
    FP c = f_and(e.cond, valid(e.v));
    if (c->kind != FNode::F) br.push_back({c, f_false()});
  }
  for (auto& g : gens) {
    const FP body = f_and(g.cond, valid(g.elem));
    if (body->kind == FNode::F) continue;
    const size_t k = g.quants.size();
    // inner[i] = E q_i .. E q_{k-1}. body   (inner[k] = body)
    std::vector<FP> inner(k + 1);
    inner[k] = body;
    for (size_t i = k; i-- > 0;) inner[i] = f_exists(g.quants[i], g.bases[i], inner[i + 1]);
    // two bindings differ first at some level i: two children of base_i with a satisfying rest, below a common prefix
    FP two = f_false();
    for (size_t i = k; i-- > 0;) {
      FP t = f_exists2(g.quants[i], g.bases[i], inner[i + 1]);
      for (size_t j = i; j-- > 0;) t = f_exists(g.quants[j], g.bases[j], t);
      two = f_or(two, t);
    }
    br.push_back({inner[0], two});
  }
  if (br.size() > 12) {   // too many alternatives for the pairwise terms: every violating pair is rendered
    FP r = f_false();
    for (auto& b : br) r = f_or(r, b.any);
    return r;
  }
  FP multi = f_false();
  for (auto& b : br) multi = f_or(multi, b.two);
  for (size_t i = 0; i < br.size(); i++)
    for (size_t j = i + 1; j < br.size(); j++) multi = f_or(multi, f_and(br[i].any, br[j].any));
  return multi;
}

Template::CountInfo Template::compile_all(const Value& parameters, int* next_quant, const Value& inventory) const {
  PE pe(*this, parameters, sv_path({}), inventory, false, next_quant);
  pe.index_rules();
  SVP set;
  try {
    set = pe.violation_set();
  } catch (const UnboundVar& e) {
    throw RegoError(e.what());
  }
  std::vector<CondElem> elems;
  std::vector<Gen> gens;
  if (set->kind == SV::CONST) { for (auto& x : set->c.items()) elems.push_back({sv_const(x), f_true()}); }
  else { elems = set->elems; gens = set->gens; }
  auto msg_of = [&](const SVP& e) -> SVP {
    if (e->kind == SV::CONST) { const Value* m = e->c.get("msg"); return (e->c.is_object() && m) ? sv_const(*m) : SVP(); }
    if (e->kind != SV::OBJ) return SVP();
    for (auto& f : e->fields) if (f.first == Value::string("msg")) return f.second;
    return SVP();
  };
  auto valid = [&](const SVP& e) -> FP {
    if (e->kind == SV::CONST) {
      const Value* m = e->c.get("msg");
      return (e->c.is_object() && m && m->is_string()) ? f_true() : f_false();
    }
    if (e->kind != SV::OBJ) return f_false();
    SVP msg = msg_of(e);
    if (!msg) return f_false();
    return f_and(pe.defined_f(e), pe.is_string_f(msg));
  };
  // a symbolic value that is a constant whatever the review holds (constants nested in object / array constructors)
  std::function<bool(const SVP&, Value*)> const_of = [&](const SVP& v, Value* out) -> bool {
    if (!v) return false;
    if (v->kind == SV::CONST) { *out = v->c; return true; }
    if (v->kind == SV::OBJ) {
      std::vector<std::pair<Value, Value>> kv;
      for (auto& f : v->fields) { Value x; if (!const_of(f.second, &x)) return false; kv.emplace_back(f.first, x); }
      *out = Value::object(std::move(kv));
      return true;
    }
    if (v->kind == SV::ARR && v->gens.empty()) {
      std::vector<Value> xs;
      for (auto& el : v->elems) { Value x; if (el.cond->kind != FNode::T || !const_of(el.v, &x)) return false; xs.push_back(x); }
      *out = Value::array(std::move(xs));
      return true;
    }
    return false;
  };
  auto details_of = [&](const SVP& e, CountBranch* b) {
    if (e->kind == SV::CONST) { const Value* d = e->c.is_object() ? e->c.get("details") : nullptr; if (d) b->dsig = "C" + to_term_string(*d); return; }
    if (e->kind != SV::OBJ) return;
    for (auto& f : e->fields) if (f.first == Value::string("details")) { Value d; b->dsig = const_of(f.second, &d) ? "C" + to_term_string(d) : "?"; }
  };
  auto head_of = [&](const SVP& e, CountBranch* b) {
    details_of(e, b);
    SVP m = msg_of(e);
    if (!m) return;
    if (m->kind == SV::CONST && m->c.is_string()) { b->is_const = true; b->text = m->c.str(); return; }
    if (m->kind != SV::OPAQUE || !m->hhead) return;
    b->head = true; b->pre = m->hpre; b->sep = m->hsep; b->sep_tail = m->hsep_tail; b->key = m->hkeypath; b->sig = m->hsig;
  };
  CountInfo ci;
  ci.viol = f_false();
  for (auto& e : elems) {
    FP c = f_and(e.cond, valid(e.v));
    ci.viol = f_or(ci.viol, c);
    if (c->kind == FNode::F) continue;
    CountBranch b;
    b.any = c; b.two = f_false();
    head_of(e.v, &b);
    ci.br.push_back(std::move(b));
  }
  for (auto& g : gens) {
    const FP body = f_and(g.cond, valid(g.elem));
    const size_t k = g.quants.size();
    std::vector<FP> inner(k + 1);
    inner[k] = body;
    for (size_t i = k; i-- > 0;) inner[i] = f_exists(g.quants[i], g.bases[i], inner[i + 1]);
    ci.viol = f_or(ci.viol, inner[0]);
    if (body->kind == FNode::F) continue;
    FP two = f_false();
    for (size_t i = k; i-- > 0;) {
      FP t = f_exists2(g.quants[i], g.bases[i], inner[i + 1]);
      for (size_t j = i; j-- > 0;) t = f_exists(g.quants[j], g.bases[j], t);
      two = f_or(two, t);
    }
    CountBranch b;
    b.any = inner[0]; b.two = two; b.body = body; b.nq = (int)k;
    if (k >= 1) { b.q = g.quants[0]; b.base = g.bases[0]; }
    head_of(g.elem, &b);
    // keyed: ONE iteration, and the key operand is a leaf of its element (base[q].x.y, no further iteration)
    if (k == 1 && b.head && b.key.size() > b.base.size() + 1) {
      bool ok = true;
      for (size_t i = 0; i < b.base.size(); i++) { const Step &x = b.key[i], &y = b.base[i]; if (x.iter != y.iter || x.key != y.key || x.q != y.q) ok = false; }
      if (ok && !(b.key[b.base.size()].iter && b.key[b.base.size()].q == b.q)) ok = false;
      for (size_t i = b.base.size() + 1; ok && i < b.key.size(); i++) if (b.key[i].iter) ok = false;
      for (size_t i = 0; ok && i < b.base.size(); i++) if (b.base[i].iter) ok = false;   // (top-level arrays only: the thresholds count elements of ONE array)
      b.keyed = ok && (!b.sep.empty() || b.sep_tail);
    }
    ci.br.push_back(std::move(b));
  }
  return ci;
}

std::vector<Violation> Template::render(const Value& review, const Value& parameters, const Value& inventory) const {
  // GK_RENDER=pe: the partial evaluator only (as before round 4); GK_RENDER_CHECK=1: both evaluators, a difference is an error
  // (read per call: the test suite switches the cross-check on for everything it renders, tests/conftest.py)
  const char* const mode = getenv("GK_RENDER");
  const bool pe_only = mode && strcmp(mode, "pe") == 0;
  const bool check = getenv("GK_RENDER_CHECK") != nullptr;
  Value setv;
  const bool fast = !pe_only && render_fast(review, parameters, inventory, &setv);
  static const bool stats = getenv("GK_RENDER_STATS") != nullptr;   // debugging aid: how many calls each evaluator served
  if (stats) {
    static std::atomic<long> n_fast{0}, n_pe{0};
    static const int reg = atexit([] {});
    (void)reg;
    const long a = fast ? ++n_fast : n_fast.load(), b = fast ? n_pe.load() : ++n_pe;
    if (((a + b) & 1023) == 0 || !fast) fprintf(stderr, "[gkgpu render] concrete evaluator %ld calls, partial evaluator %ld (last: %s)\n", a, b, pkg_name_.c_str());
  }
  if (!fast || check) {
    int nq = 0;
    PE pe(*this, parameters, sv_const(review), inventory, true, &nq);
    pe.index_rules();
    SVP set = pe.violation_set();
    if (set->kind != SV::CONST) throw RegoError("internal: concrete evaluation left a symbolic residue");
    if (fast && check && !(set->c == setv)) throw RegoError("internal: the concrete evaluators disagree: " + to_term_string(setv) + " against " + to_term_string(set->c));
    setv = set->c;
  }
  std::vector<Violation> out;
  if (!setv.is_set() && !setv.is_array()) return out;
  for (const Value& v : setv.items()) {
    if (!v.is_object()) continue;
    const Value* m = v.get("msg");
    if (!m || !m->is_string()) continue;
    Violation x;
    x.msg = m->str();
    const Value* d = v.get("details");
    if (d) x.details = *d;
    bool dup = false;
    // (a member without details and one with {} are the same result: the driver's hook reports object.get(r, "details", {}))
    auto empty_obj = [](const Value& d) { return !d.defined() || (d.is_object() && d.size() == 0); };
    for (auto& o : out) if (o.msg == x.msg && ((empty_obj(o.details) && empty_obj(x.details)) || (o.details.defined() && x.details.defined() && o.details == x.details))) dup = true;
    if (!dup) out.push_back(x);
  }
  return out;
}

}  // namespace gk
