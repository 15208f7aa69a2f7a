// Concrete evaluation of a template for ONE review: the violation set the host renders messages from.
//
// The reference gets this from OPA's topdown evaluator (frameworks/constraint pkg/client/drivers/rego Driver.Query, called
// from pkg/webhook/policy.go:826 and pkg/audit/manager.go:621,719).  The engine's partial evaluator (pe.cpp) can evaluate a
// concrete review as well -- every value a constant, every condition true or false -- and did all the rendering until round 4;
// but it carries its symbolic machinery along (a heap-allocated symbolic value per constant, a state copy per produced value:
// ~470 allocations and 55 us per rendered pair).  This evaluator is the same algorithm, function by function (the comments name
// the pe.cpp twin), over plain Values with one binding stack and continuations instead of copied states.  It does not report
// errors of its own: anything that is not a plain success -- an evaluation error, a construct it leaves out -- makes
// Template::render fall back to the partial evaluator, whose behaviour (messages included) stays the reference point;
// GK_RENDER_CHECK=1 runs both on every call and refuses a difference.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "builtins.hpp"
#include "pe.hpp"

namespace gk {

namespace {
struct CUnbound {};    // an unbound variable in a value position: the body tries another literal first (pe.cpp UnboundVar)
struct CFallback {};   // not handled here: the partial evaluator renders this call

// non-owning callable reference (continuations live on the stack of the caller)
template <class Sig> class fref;
template <class R, class... A>
class fref<R(A...)> {
  void* obj_;
  R (*call_)(void*, A...);

 public:
  template <class F, class = typename std::enable_if<!std::is_same<typename std::decay<F>::type, fref>::value>::type>
  fref(F&& f) : obj_((void*)std::addressof(f)), call_([](void* o, A... a) -> R { return (*(typename std::remove_reference<F>::type*)o)(static_cast<A>(a)...); }) {}
  R operator()(A... a) const { return call_(obj_, static_cast<A>(a)...); }
};
typedef fref<void()> K;
typedef fref<void(const Value&)> KV;
typedef fref<void(const Value&, const Value&)> KKV;

inline bool wildcard(const std::string& n) { return n.size() >= 2 && n[0] == '$' && n[1] == 'w'; }
}  // namespace

// What can be known about a template's terms before any review is seen: which rules a name or a call means.
struct Template::CIndex {
  typedef std::vector<const Rule*> RuleSet;
  struct CallInfo { const RuleSet* user = nullptr; std::string builtin; bool known = false; };
  struct DataInfo { const RuleSet* rules = nullptr; size_t skip = 0; };
  struct RuleInfo { const std::string* pkg = nullptr; const Module* mod = nullptr; };
  std::map<std::string, std::map<std::string, const RuleSet*>> by_pkg;
  std::unordered_map<const Rule*, RuleInfo> rule_info;
  std::unordered_map<const Term*, CallInfo> calls;
  std::unordered_map<const Term*, DataInfo> data_refs;          // Ref with head `data` (or an import of data.*)
  std::unordered_map<const Term*, std::vector<Value>> import_ops;   // Ref whose head is an import alias: the alias' path as keys
  const RuleSet* find(const std::string& pkg, const std::string& name) const {
    auto it = by_pkg.find(pkg);
    if (it == by_pkg.end()) return nullptr;
    auto jt = it->second.find(name);
    return jt == it->second.end() ? nullptr : jt->second;
  }
};

namespace {
class CEval {
 public:
  typedef Template::CIndex CIndex;
  typedef CIndex::RuleSet RuleSet;
  CEval(const CIndex& ix, const std::string& main_pkg, const Value& params, const Value& review, const Value& inventory)
      : ix_(ix), main_pkg_(main_pkg), inventory_(inventory), env_(scratch().env), sols_pool_(scratch().sols) {
    static const Value k_parameters = Value::string("parameters"), k_review = Value::string("review"), empty_object = Value::object({});
    ValuePairs in;
    in.reserve(2);
    in.emplace_back(k_parameters, params.defined() ? params : empty_object);
    in.emplace_back(k_review, review);
    input_ = Value::object(std::move(in));
    env_.clear();
    if (env_.capacity() < 64) env_.reserve(64);
  }
  ~CEval() { env_.clear(); for (auto& s : sols_pool_) { s->binds.clear(); s->ends.clear(); } }   // (the values go; the capacity stays for the thread's next call)
  Value violation_set() {
    const RuleSet* rs = ix_.find(main_pkg_, "violation");
    if (!rs) return Value::set({});
    const std::vector<Value>& alts = rule_values(rs);
    if (alts.empty()) return Value::set({});
    return alts[0];
  }

 private:
  const CIndex& ix_;
  const std::string& main_pkg_;
  Value input_, inventory_;
  struct Bind { const std::string* name; Value v; bool shadow; };
  struct Sols { std::vector<Bind> binds; std::vector<uint32_t> ends; };
  // the binding stack and the solution lists keep their capacity from one call of a thread to its next (an evaluation allocates
  // for the values it makes, not for its own bookkeeping)
  struct Scratch { std::vector<Bind> env; std::vector<std::unique_ptr<Sols>> sols; std::vector<std::unique_ptr<ValueVec>> vecs; };
  // a value list for the duration of a scope (call arguments, the members a comprehension collects), from the thread's pool
  struct Lease {
    CEval& e; ValueVec& v;
    explicit Lease(CEval& ev) : e(ev), v(ev.lease()) {}
    ~Lease() { v.clear(); e.vec_depth_--; }
  };
  ValueVec& lease() {
    std::vector<std::unique_ptr<ValueVec>>& p = scratch().vecs;
    if (vec_depth_ >= p.size()) p.emplace_back(new ValueVec());
    ValueVec& v = *p[vec_depth_++];
    v.clear();
    return v;
  }
  size_t vec_depth_ = 0;
  static Scratch& scratch() { static thread_local Scratch s; return s; }
  std::vector<Bind>& env_;
  size_t frame_ = 0;   // bindings below belong to callers: a rule body does not see them
  int depth_ = 0;
  struct RuleVal { bool done = false, in_progress = false; std::vector<Value> alts; };
  std::map<const RuleSet*, RuleVal> cache_;
  std::vector<std::unique_ptr<Sols>>& sols_pool_;
  size_t sols_depth_ = 0;

  // ---- bindings
  const Bind* lookup(const std::string& name) const {
    for (size_t i = env_.size(); i-- > frame_;) {
      const Bind& b = env_[i];
      if (b.name == &name || *b.name == name) return &b;
    }
    return nullptr;
  }
  const Value* bound(const std::string& name) const { const Bind* b = lookup(name); return b && !b->shadow ? &b->v : nullptr; }
  const CIndex::RuleInfo& info(const Rule* r) const { auto it = ix_.rule_info.find(r); if (it == ix_.rule_info.end()) throw CFallback(); return it->second; }
  const std::vector<std::string>* import_of(const Rule* r, const std::string& alias) const {   // (pe.cpp import_of)
    for (auto& im : info(r).mod->imports) if (im.second == alias && im.first.size() > 1) return &im.first;
    return nullptr;
  }
  bool is_global(const std::string& name, const Rule* r) const {   // (pe.cpp is_global)
    if (name == "input" || name == "data") return true;
    if (ix_.find(*info(r).pkg, name)) return true;
    return import_of(r, name) != nullptr;
  }
  bool is_unbound(const Term& t, const Rule* r) const { return t.kind == Term::Var && !bound(t.name) && !is_global(t.name, r); }
  bool has_unbound(const Term& t, const Rule* r) const {   // (pe.cpp has_unbound)
    if (t.kind == Term::Var) return is_unbound(t, r);
    if (t.kind == Term::Array || t.kind == Term::Object) { for (auto& a : t.args) if (has_unbound(*a, r)) return true; }
    return false;
  }
  void bind(const std::string& name, const Value& v, K k) {   // (pe.cpp bind)
    if (!v.defined()) return;
    if (wildcard(name)) { k(); return; }
    env_.push_back({&name, v, false});
    const size_t at = env_.size();
    k();
    env_.resize(at - 1);
  }

  // ---- bodies (pe.cpp eval_body / eval_lits: the first literal that can be evaluated goes first)
  void body(const Body& b, const Rule* r, K k) {
    if (b.size() > 64) throw CFallback();
    lits(b, b.empty() ? 0ull : (b.size() == 64 ? ~0ull : ((1ull << b.size()) - 1)), r, k);
  }
  void lits(const Body& b, uint64_t remaining, const Rule* r, K k) {
    if (!remaining) { k(); return; }
    for (uint64_t m = remaining; m; m &= m - 1) {
      const int idx = __builtin_ctzll(m);
      if (sols_depth_ >= sols_pool_.size()) sols_pool_.emplace_back(new Sols());
      Sols& S = *sols_pool_[sols_depth_++];
      S.binds.clear(); S.ends.clear();
      const size_t mark = env_.size();
      try {
        auto rec = [&]() { S.binds.insert(S.binds.end(), env_.begin() + mark, env_.end()); S.ends.push_back((uint32_t)S.binds.size()); };
        literal(b[(size_t)idx], r, rec);
      } catch (const CUnbound&) {
        env_.resize(mark);
        sols_depth_--;
        continue;   // try a later literal first
      }
      const uint64_t rest = remaining & ~(1ull << idx);
      uint32_t from = 0;
      for (size_t i = 0; i < S.ends.size(); i++) {
        env_.insert(env_.end(), S.binds.begin() + from, S.binds.begin() + S.ends[i]);
        from = S.ends[i];
        try { lits(b, rest, r, k); } catch (...) { sols_depth_--; throw; }
        env_.resize(mark);
      }
      sols_depth_--;
      return;
    }
    throw CUnbound();
  }

  static bool truthy(const Value& v) { return v.defined() && !(v.is_bool() && !v.b); }

  void literal(const Literal& l, const Rule* r, K k) {   // (pe.cpp eval_literal)
    switch (l.kind) {
      case Literal::Expr: term(*l.a, r, [&](const Value& v) { if (truthy(v)) k(); }); break;
      case Literal::Assign: case Literal::Unify: unify_terms(*l.a, *l.b, r, k); break;
      case Literal::Not: {
        bool any = false;
        literal(*l.inner, r, [&]() { any = true; });
        if (!any) k();
        break;
      }
      case Literal::Some: {
        const size_t mark = env_.size();
        for (auto& nm : l.names) env_.push_back({&nm, Value(), true});
        k();
        env_.resize(mark);
        break;
      }
      case Literal::SomeIn:
        term(*l.c, r, [&](const Value& coll) {
          iterate(coll, [&](const Value& key, const Value& val) {
            unify_value(*l.b, val, r, [&]() {
              if (!l.a) { k(); return; }
              unify_value(*l.a, key, r, k);
            });
          });
        });
        break;
      case Literal::Every:
        term(*l.c, r, [&](const Value& coll) {
          bool all = true;
          iterate(coll, [&](const Value& key, const Value& val) {
            bool sat = false;
            auto run = [&]() { body(*l.body, r, [&]() { sat = true; }); };
            unify_value(*l.b, val, r, [&]() {
              if (l.a) unify_value(*l.a, key, r, run); else run();
            });
            if (!sat) all = false;
          });
          if (all) k();
        });
        break;
    }
  }

  // ---- unification (pe.cpp unify_terms / unify_value)
  void unify_terms(const Term& a, const Term& b, const Rule* r, K k) {
    if (is_unbound(a, r)) term(b, r, [&](const Value& v) { bind(a.name, v, k); });
    else if (is_unbound(b, r)) term(a, r, [&](const Value& v) { bind(b.name, v, k); });
    else if ((a.kind == Term::Array || a.kind == Term::Object) && has_unbound(a, r)) term(b, r, [&](const Value& v) { unify_value(a, v, r, k); });
    else if ((b.kind == Term::Array || b.kind == Term::Object) && has_unbound(b, r)) term(a, r, [&](const Value& v) { unify_value(b, v, r, k); });
    else term(a, r, [&](const Value& x) { term(b, r, [&](const Value& y) { if (x == y) k(); }); });
  }
  void unify_array(const Term& pat, const Value& val, size_t i, const Rule* r, K k) {
    if (i == pat.args.size()) { k(); return; }
    unify_value(*pat.args[i], val.items()[i], r, [&]() { unify_array(pat, val, i + 1, r, k); });
  }
  void unify_object(const Term& pat, const Value& val, size_t i, const Rule* r, K k) {
    const size_t n = pat.args.size() / 2;
    if (i == n) { k(); return; }
    term(*pat.args[2 * i], r, [&](const Value& key) {
      if (!val.is_object() || val.size() != n) return;
      const Value* f = val.get(key);
      if (!f) return;
      unify_value(*pat.args[2 * i + 1], *f, r, [&]() { unify_object(pat, val, i + 1, r, k); });
    });
  }
  void unify_value(const Term& pat, const Value& val, const Rule* r, K k) {
    if (pat.kind == Term::Var && is_unbound(pat, r)) { bind(pat.name, val, k); return; }
    if (pat.kind == Term::Array && has_unbound(pat, r)) {
      if (!val.is_array() || val.size() != pat.args.size()) return;
      unify_array(pat, val, 0, r, k);
      return;
    }
    if (pat.kind == Term::Object && has_unbound(pat, r)) { unify_object(pat, val, 0, r, k); return; }
    term(pat, r, [&](const Value& v) { if (v == val) k(); });
  }

  // ---- terms (pe.cpp eval_term)
  void seq(const std::vector<TermP>& ts, size_t i, ValueVec& acc, const Rule* r, K k) {   // (pe.cpp eval_seq)
    if (i == ts.size()) { k(); return; }
    term(*ts[i], r, [&](const Value& v) {
      acc.push_back(v);
      seq(ts, i + 1, acc, r, k);
      acc.pop_back();
    });
  }
  void term(const Term& t, const Rule* r, KV kv) {
    switch (t.kind) {
      case Term::Scalar: kv(t.value); break;
      case Term::Var: {
        if (const Value* b = bound(t.name)) { const Value v = *b; kv(v); return; }   // (a copy: the continuation may grow the binding stack)
        if (t.name == "input") { kv(input_); return; }
        if (t.name == "data") { data_ref(t, nullptr, r, kv); return; }
        if (const RuleSet* rs = ix_.find(*info(r).pkg, t.name)) { for (const Value& v : rule_values(rs)) kv(v); return; }   // (a finished entry of the cache never changes)
        throw CUnbound();
      }
      case Term::Ref: {
        const Term& head = *t.head;
        if (head.kind == Term::Var && !bound(head.name)) {
          if (head.name == "data") { data_ref(t, &t.args, r, kv); return; }
          auto io = ix_.import_ops.find(&t);
          if (io != ix_.import_ops.end()) {
            const std::vector<std::string>* imp = import_of(r, head.name);
            if (!imp) throw CFallback();
            if ((*imp)[0] == "data") { data_ref(t, &t.args, r, kv); return; }
            walk_keys(input_, io->second, 0, [&](const Value& v) { walk(v, t.args, 0, r, kv); });
            return;
          }
        }
        term(head, r, [&](const Value& h) { walk(h, t.args, 0, r, kv); });
        break;
      }
      case Term::Call: call(t, r, kv); break;
      case Term::BinOp: {
        const std::string& op = t.name;
        term(*t.args[0], r, [&](const Value& a) {
          term(*t.args[1], r, [&](const Value& b) {
            if (op == "==") kv(Value::boolean(compare(a, b) == 0));
            else if (op == "!=") kv(Value::boolean(compare(a, b) != 0));
            else if (op == "<") kv(Value::boolean(compare(a, b) < 0));
            else if (op == "<=") kv(Value::boolean(compare(a, b) <= 0));
            else if (op == ">") kv(Value::boolean(compare(a, b) > 0));
            else if (op == ">=") kv(Value::boolean(compare(a, b) >= 0));
            else if (op == "in") {
              bool m = false;
              if (b.is_set()) m = b.set_has(a);
              else if (b.is_array()) { for (auto& e : b.items()) if (e == a) { m = true; break; } }
              kv(Value::boolean(m));
            } else {
              const Value v = rego_arith(op, a, b);
              if (v.defined()) kv(v);
            }
          });
        });
        break;
      }
      case Term::Array: case Term::SetLit: {
        Lease L(*this);
        ValueVec& acc = L.v;
        seq(t.args, 0, acc, r, [&]() { const Value v = t.kind == Term::Array ? Value::array(acc) : Value::set(acc); kv(v); });
        break;
      }
      case Term::Object: {
        Lease L(*this);
        ValueVec& acc = L.v;
        seq(t.args, 0, acc, r, [&]() {
          ValuePairs p;
          p.reserve(acc.size() / 2);
          for (size_t i = 0; i + 1 < acc.size(); i += 2) p.emplace_back(acc[i], acc[i + 1]);
          const Value v = Value::object(std::move(p));
          kv(v);
        });
        break;
      }
      case Term::ArrComp: case Term::SetComp: {
        Value v;
        {
          Lease L(*this);
          ValueVec& items = L.v;
          body(*t.body, r, [&]() { term(*t.head, r, [&](const Value& h) { items.push_back(h); }); });
          v = t.kind == Term::ArrComp ? Value::array(items) : Value::set(items);
        }
        kv(v);
        break;
      }
      case Term::ObjComp: {
        ValuePairs pairs;
        body(*t.body, r, [&]() { term(*t.head, r, [&](const Value& key) { term(*t.head2, r, [&](const Value& val) { pairs.emplace_back(key, val); }); }); });
        const Value v = Value::object(std::move(pairs));
        kv(v);
        break;
      }
    }
  }

  // ---- references (pe.cpp iterate / index / walk / eval_data_ref)
  void iterate(const Value& c, KKV fn) {
    if (c.is_array()) { for (size_t i = 0; i < c.size(); i++) fn(Value::integer((i128)i), c.items()[i]); }
    else if (c.is_set()) { for (auto& x : c.items()) fn(x, x); }
    else if (c.is_object()) { for (auto& kv : c.pairs()) fn(kv.first, kv.second); }
  }
  void index(const Value& c, const Value& k, KV fn) {
    if (c.is_object()) { const Value* v = c.get(k); if (v) fn(*v); }
    else if (c.is_array()) { if (k.is_number() && k.is_int && k.i >= 0 && (size_t)k.i < c.size()) fn(c.items()[(size_t)k.i]); }
    else if (c.is_set()) { if (c.set_has(k)) fn(k); }
  }
  void walk(const Value& cur, const std::vector<TermP>& ops, size_t i, const Rule* r, KV kv) {
    if (i == ops.size()) { kv(cur); return; }
    const Term& op = *ops[i];
    if (op.kind == Term::Var && is_unbound(op, r)) {
      const bool wild = wildcard(op.name);
      iterate(cur, [&](const Value& key, const Value& val) {
        if (wild) { walk(val, ops, i + 1, r, kv); return; }
        env_.push_back({&op.name, key, false});
        const size_t at = env_.size();
        walk(val, ops, i + 1, r, kv);
        env_.resize(at - 1);
      });
      return;
    }
    if ((op.kind == Term::Array || op.kind == Term::Object) && has_unbound(op, r)) {
      iterate(cur, [&](const Value& key, const Value& val) { unify_value(op, key, r, [&]() { walk(val, ops, i + 1, r, kv); }); });
      return;
    }
    term(op, r, [&](const Value& k) { index(cur, k, [&](const Value& nxt) { walk(nxt, ops, i + 1, r, kv); }); });
  }
  void walk_keys(const Value& cur, const std::vector<Value>& keys, size_t i, KV kv) {
    if (i == keys.size()) { kv(cur); return; }
    index(cur, keys[i], [&](const Value& nxt) { walk_keys(nxt, keys, i + 1, kv); });
  }
  // data.<...>: a rule of a loaded package (the longest package prefix that names one), else the base document {inventory: ..}
  void data_ref(const Term& t, const std::vector<TermP>* ops, const Rule* r, KV kv) {
    static const std::vector<TermP> none;
    const std::vector<TermP>& o = ops ? *ops : none;
    auto it = ix_.data_refs.find(&t);
    if (it == ix_.data_refs.end()) throw CFallback();
    const CIndex::DataInfo& d = it->second;
    auto io = ix_.import_ops.find(&t);
    if (io != ix_.import_ops.end()) throw CFallback();   // (data.* through an import alias with more operands: rare; the general evaluator does it)
    if (d.rules) { for (const Value& v : rule_values(d.rules)) walk(v, o, d.skip, r, kv); return; }
    ValuePairs root;
    if (inventory_.defined()) root.emplace_back(Value::string("inventory"), inventory_);
    const Value rootv = Value::object(std::move(root));
    walk(rootv, o, 0, r, kv);
  }

  // ---- calls (pe.cpp eval_call / call_function)
  void call(const Term& t, const Rule* r, KV kv) {
    auto it = ix_.calls.find(&t);
    if (it == ix_.calls.end() || !it->second.known) throw CFallback();
    const CIndex::CallInfo& ci = it->second;
    Lease L(*this);
    ValueVec& args = L.v;
    seq(t.args, 0, args, r, [&]() {
      if (ci.user) { call_function(ci.user, args, kv); return; }
      const Value v = call_builtin(ci.builtin, args);
      if (v.defined()) kv(v);
    });
  }
  void unify_args(const Rule* fr, const ValueVec& args, size_t i, K k) {
    if (i == args.size()) { k(); return; }
    unify_value(*fr->args[i], args[i], fr, [&]() { unify_args(fr, args, i + 1, k); });
  }
  void call_function(const RuleSet* rules, const ValueVec& args, KV kv) {   // (`args`: the caller's leased list; nothing below touches it -- nested calls lease their own)
    if (++depth_ > 64) throw CFallback();   // (the general evaluator words the recursion error)
    std::vector<Value> results;
    results.reserve(2);
    const size_t saved_frame = frame_, mark = env_.size();
    for (const Rule* fr : *rules) {
      if (fr->kind != Rule::Function || fr->args.size() != args.size()) continue;
      frame_ = env_.size();
      try { unify_args(fr, args, 0, [&]() { complete_def(fr, results); }); } catch (...) { frame_ = saved_frame; env_.resize(mark); depth_--; throw; }
      frame_ = saved_frame;
    }
    depth_--;
    for (const Value& v : results) kv(v);   // (the callee's bindings are gone: the continuation is the caller's)
  }

  // ---- rules (pe.cpp rule_alts / complete_def)
  void complete_def(const Rule* r, std::vector<Value>& out) {
    const size_t n_links = 1 + r->elses.size();
    for (size_t li = 0; li < n_links; li++) {
      const TermP& value = li == 0 ? r->value : r->elses[li - 1].first;
      const Body& b = li == 0 ? r->body : r->elses[li - 1].second;
      bool here = false;
      body(b, r, [&]() {
        if (value) term(*value, r, [&](const Value& v) { if (v.defined()) { out.push_back(v); here = true; } });
        else { out.push_back(Value::boolean(true)); here = true; }
      });
      if (n_links == 1 || here) break;
    }
  }
  const std::vector<Value>& rule_values(const RuleSet* rs) {
    RuleVal& rv = cache_[rs];
    if (rv.done) return rv.alts;
    if (rv.in_progress) throw CFallback();   // recursion: the general evaluator reports it
    rv.in_progress = true;
    const size_t saved_frame = frame_, mark = env_.size();
    std::vector<Value> alts;
    try {
      const Rule::Kind kind = (*rs)[0]->kind;
      if (kind == Rule::Function) throw CFallback();
      if (kind == Rule::PartialSet) {
        Lease L(*this);
        ValueVec& items = L.v;
        for (const Rule* r : *rs) {
          frame_ = env_.size();
          body(r->body, r, [&]() { term(*r->key, r, [&](const Value& k) { items.push_back(k); }); });
        }
        alts.push_back(Value::set(items));
      } else if (kind == Rule::PartialObject) {
        ValuePairs pairs;
        for (const Rule* r : *rs) {
          frame_ = env_.size();
          body(r->body, r, [&]() { term(*r->key, r, [&](const Value& k) { term(*r->value, r, [&](const Value& v) { pairs.emplace_back(k, v); }); }); });
        }
        alts.push_back(Value::object(std::move(pairs)));
      } else {
        Value def;
        for (const Rule* r : *rs) {
          frame_ = env_.size();
          if (r->is_default) { bool first = true; term(*r->value, r, [&](const Value& v) { if (first) { def = v; first = false; } }); continue; }
          complete_def(r, alts);
        }
        if (def.defined() && alts.empty()) alts.push_back(def);
      }
    } catch (...) { frame_ = saved_frame; env_.resize(mark); cache_.erase(rs); throw; }
    frame_ = saved_frame;
    RuleVal& rv2 = cache_[rs];
    rv2.alts = std::move(alts);
    rv2.done = true;
    rv2.in_progress = false;
    return rv2.alts;
  }
};
}  // namespace

// ---- the per-template index
static void index_term(Template::CIndex& ix, const TermP& tp, const Rule* r, const std::string& pkg, const Module* mod);
static void index_body(Template::CIndex& ix, const Body& b, const Rule* r, const std::string& pkg, const Module* mod) {
  for (const Literal& l : b) {
    const Literal* x = &l;
    while (x) {
      index_term(ix, x->a, r, pkg, mod); index_term(ix, x->b, r, pkg, mod); index_term(ix, x->c, r, pkg, mod);
      if (x->body) index_body(ix, *x->body, r, pkg, mod);
      x = x->inner.get();
    }
  }
}
static const std::vector<std::string>* import_path(const Module* mod, const std::string& alias) {
  for (auto& im : mod->imports) if (im.second == alias && im.first.size() > 1) return &im.first;
  return nullptr;
}
static void index_term(Template::CIndex& ix, const TermP& tp, const Rule* r, const std::string& pkg, const Module* mod) {
  if (!tp) return;
  const Term& t = *tp;
  index_term(ix, t.head, r, pkg, mod);
  index_term(ix, t.head2, r, pkg, mod);
  for (auto& a : t.args) index_term(ix, a, r, pkg, mod);
  if (t.body) index_body(ix, *t.body, r, pkg, mod);
  if (t.kind == Term::Call) {   // (pe.cpp eval_call: a function of this package, of an imported / named package, else a builtin)
    Template::CIndex::CallInfo ci;
    std::string name;
    for (size_t i = 0; i < t.path.size(); i++) { if (i) name += "."; name += t.path[i]; }
    if (t.path.size() == 1 && ix.find(pkg, t.path[0])) ci.user = ix.find(pkg, t.path[0]);
    else {
      std::vector<std::string> full = t.path;
      if (const auto* imp = import_path(mod, t.path[0])) { full = *imp; full.insert(full.end(), t.path.begin() + 1, t.path.end()); }
      if (full[0] == "data" && full.size() >= 2) {
        std::string p;
        for (size_t i = 1; i + 1 < full.size(); i++) { if (i > 1) p += "."; p += full[i]; }
        ci.user = ix.find(p, full.back());
      }
    }
    ci.known = ci.user != nullptr || has_builtin(name);
    ci.builtin = name;
    ix.calls[&t] = ci;
  }
  if ((t.kind == Term::Ref && t.head && t.head->kind == Term::Var) || (t.kind == Term::Var && t.name == "data")) {
    const std::string& head = t.kind == Term::Var ? t.name : t.head->name;
    const std::vector<std::string>* imp = head == "data" ? nullptr : import_path(mod, head);
    if (imp && t.kind == Term::Ref) {
      std::vector<Value> keys;
      for (size_t i = 1; i < imp->size(); i++) keys.push_back(Value::string((*imp)[i]));
      ix.import_ops[&t] = keys;
    }
    if (head == "data" || (imp && (*imp)[0] == "data")) {   // (pe.cpp eval_data_ref: the longest constant prefix that names a rule)
      Template::CIndex::DataInfo d;
      if (head == "data") {
        std::vector<std::string> consts;
        if (t.kind == Term::Ref) for (auto& o : t.args) { if (o->kind == Term::Scalar && o->value.is_string()) consts.push_back(o->value.str()); else break; }
        for (size_t n = consts.size(); n-- > 0;) {
          std::string p;
          for (size_t i = 0; i < n; i++) { if (i) p += "."; p += consts[i]; }
          if (const auto* rs = ix.find(p, consts[n])) { d.rules = rs; d.skip = n + 1; break; }
        }
      }
      ix.data_refs[&t] = d;
    }
  }
}

const Template::CIndex& Template::cindex() const {
  std::call_once(cindex_once_, [this]() {
    std::unique_ptr<CIndex> ix(new CIndex());
    for (auto& kv : rules_) ix->by_pkg[kv.first.first][kv.first.second] = &kv.second;
    for (auto& kv : rule_pkgs_) { ix->rule_info[kv.first].pkg = &kv.second; }
    for (auto& kv : rule_mods_) { ix->rule_info[kv.first].mod = kv.second; }
    for (auto& kv : rules_)
      for (const Rule* r : kv.second) {
        const std::string& pkg = rule_pkgs_.at(r);
        const Module* mod = rule_mods_.at(r);
        for (auto& a : r->args) index_term(*ix, a, r, pkg, mod);
        index_term(*ix, r->key, r, pkg, mod);
        index_term(*ix, r->value, r, pkg, mod);
        index_body(*ix, r->body, r, pkg, mod);
        for (auto& e : r->elses) { index_term(*ix, e.first, r, pkg, mod); index_body(*ix, e.second, r, pkg, mod); }
      }
    cindex_ = ix.release();
  });
  return *cindex_;
}

// The violation set of one review by the concrete evaluator; false: not evaluated here (the caller runs the general evaluator).
bool Template::render_fast(const Value& review, const Value& parameters, const Value& inventory, Value* set) const {
  try {
    CEval ev(cindex(), pkg_name_, parameters, review, inventory);
    *set = ev.violation_set();
    return true;
  } catch (const CFallback&) { return false; }
  catch (const CUnbound&) { return false; }
  catch (const std::exception&) { return false; }
}

void Template::drop_cindex() { delete cindex_; cindex_ = nullptr; }

}  // namespace gk
