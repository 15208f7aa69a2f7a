// Partial evaluator for the Rego subset: evaluates a template's `violation` rule with
//   input.parameters = constant (the constraint's spec.parameters)   and
//   input.review     = SYMBOLIC (AOT compile: result is a quantified boolean formula over review rows)
//                   or CONCRETE (host-side rendering of msg/details for the sparse violating pairs).
// One evaluator, two modes, so the compiled predicate plan and the rendered messages cannot drift apart.
//
// Semantics restated from the OPA language reference (the reference evaluates with OPA v1.17.1 topdown through the
// frameworks Rego driver; both third-party, go.mod:18-19): bodies are conjunctions, undefined propagates,
// `not` is negation as failure, partial-set rules are unions, functions may have several bodies.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "dexpr.hpp"
#include "rego_ast.hpp"
#include "value.hpp"

namespace gk {

// (Unsupported -- valid Rego this engine refuses -- is declared next to RegoError in rego_ast.hpp: the parser throws it too)

// ------------------------------------------------------------------------------------------------ formulas
struct Step {
  bool iter = false;     // true: iterate children (array elements / object members), bound to quantifier q
  std::string key;
  int q = -1;
};
typedef std::vector<Step> SPath;   // rooted at input.review

struct FNode;
struct Atom {
  enum Kind { DEFINED, TRUTHY, CMP, TYPE, STR_PREFIX, STR_SUFFIX, STR_CONTAINS, STR_IN_SET, STR_REGEX, SPLIT_CMP, SPLIT_COUNT,
              COUNT_CMP, FLAG, VEQ, KEYCMP, SPLIT_PREFIX,
              DICT /* leaf-local expression `dx` over the leaf at `path` is true (dexpr.hpp) */ } kind = DEFINED;
  SPath path;
  int cmp = 0;          // CmpOp
  Value k;              // constant operand (CMP / STR_* / SPLIT_* / COUNT_CMP / KEYCMP; STR_IN_SET: set/array of strings)
  uint32_t mask = 0;    // TYPE: bit per RowType
  char cut = 0, sep = 0;
  int idx = 0;          // SPLIT_CMP component (negative: from end)
  uint32_t flag = 0;    // FLAG: review flag bit index
  SPath path2;          // VEQ
  int q = -1;           // KEYCMP
  DX dx;                // DICT
  // DICT made by PROMOTING plain row predicates (string tests on one leaf, lower.cpp fold_dict): the formula it stands for, as the
  // device would evaluate it from the leaf's own rows.  The lowering falls back to it when the dictionary cannot take the expression
  // (62 bits per leaf pattern, overlapping patterns, a frozen registry); not part of the atom's identity (the text is path + dx)
  std::shared_ptr<const FNode> alt;
};

typedef std::shared_ptr<const FNode> FP;
struct FNode {
  enum Kind { T, F, AND, OR, NOT, EXISTS, ATOM } kind = T;
  std::vector<FP> kids;
  int q = -1;           // EXISTS: quantifier id; quantifies over children of `base`
  SPath base;           // EXISTS
  bool two = false;     // EXISTS: at least TWO children of `base` satisfy the body ("E2": instance counting for the RESULT totals)
  int atleast = 2;      // ... with `two`: at least THIS MANY children do ("E3", "E4", ..: the result COUNTS of round 4)
  Atom atom;
  // canonical text (f_to_string), derived once per node: lowering keys sub-formulas by it over and over.  Nodes are immutable
  // once made (mkf / f_* reset the cache of a copy they are handed); formulas are built and lowered under the engine's locks
  mutable std::string canon;
  mutable bool canon_set = false;
  // (lower.cpp leaf_of, cached the same way: the leaf of a leaf-local formula, "" = not leaf-local; whether it holds a DICT atom)
  mutable std::string leaf_text;
  mutable int leaf_state = 0;   // 0 unknown, 1 known without a DICT atom, 2 known with one
  mutable int needs_leaf_state = 0;   // (lower.cpp needs_leaf) 0 unknown, 1 no, 2 yes
};

FP f_true();
FP f_false();
FP f_and(FP a, FP b);
FP f_or(FP a, FP b);
FP f_not(FP a);
FP f_atom(const Atom& a);
FP f_exists(int q, const SPath& base, FP body);
FP f_exists2(int q, const SPath& base, FP body);              // at least two children of `base` satisfy the body
FP f_exists_k(int q, const SPath& base, FP body, int k);      // at least k (>= 1; 1 is the plain EXISTS)
FP f_exists_like(const FNode& proto, FP body);                // EXISTS / E2 with `proto`'s quantifier, base and flag
FP f_all(const std::vector<FP>& v);
FP f_any(const std::vector<FP>& v);
const std::string& f_to_string(const FP& f);   // (the node's cached canonical text)
std::string spath_to_string(const SPath& p);

// ------------------------------------------------------------------------------------------------ symbolic values
struct SV;
typedef std::shared_ptr<const SV> SVP;
struct CondElem { SVP v; FP cond; };
struct Gen { SVP elem; std::vector<int> quants; std::vector<SPath> bases; FP cond; };

struct SV {
  enum Kind { CONST, PATH, KEYOF, OBJ, ARR, SET, OPAQUE, BOOLF, STRX, COUNTOF, CARD,
              DERIVED /* value computed from ONE review leaf (`path`) by the expression `dx` */ } kind = CONST;
  Value c;                                        // CONST
  SPath path;                                     // PATH / STRX / COUNTOF
  int q = -1;                                     // KEYOF
  std::vector<std::pair<Value, SVP>> fields;      // OBJ (constant keys)
  std::vector<CondElem> elems;                    // ARR / SET / CARD
  std::vector<Gen> gens;                          // ARR / SET / CARD
  FP f;                                           // OPAQUE: definedness; BOOLF: truth value
  FP d;                                           // BOOLF: definedness
  char cut = 0, sep = 0;                          // STRX
  enum XK { XTRIM, XARR, XCOMP, XCOUNT } xkind = XTRIM;
  int idx = 0;                                    // XCOMP index / XCOUNT offset
  DX dx;                                          // DERIVED
  // OPAQUE from sprintf: the HEAD of the formatted text -- literal text `hpre`, then the first verb's operand `hkey` (a review
  // leaf), then literal text `hsep` up to the next verb (`hsep_tail`: up to the END of the format) -- and a signature of format
  // and operands (`hsig`: one entry per operand, "C<json>" for constants, "P<path>" for review leaves with quantifiers erased,
  // "?" otherwise; entry 0 is the format).  What the result counting needs to tell messages apart (Template::count_forms).
  bool hhead = false, hsep_tail = false;
  std::string hpre, hsep;
  SPath hkeypath;
  std::vector<std::string> hsig;
};

struct Violation {   // render mode output
  std::string msg;
  Value details;     // Undefined when the template gave none (driver reports {})
};

// One compiled template: main module + libs.
class Template : public std::enable_shared_from_this<Template> {
 public:
  // disabled: builtins the driver was constructed without (rego.DisableBuiltins, main.go:424): a call of one is `rego_type_error:
  // undefined function`, as for a name nobody defines.  nullptr = the reference deployment's default, {"http.send"}
  // (--disable-opa-builtin, test/bats/test.bats:492-498)
  Template(const std::string& rego, const std::vector<std::string>& libs, const std::set<std::string>* disabled = nullptr);   // throws RegoError
  const std::string& package_name() const { return pkg_name_; }

  // AOT: formula that is true iff the template yields >= 1 violation for a review, given constant parameters.
  // Throws Unsupported if the template uses constructs the device plan cannot express.
  // `inventory`: data.inventory as a CONSTANT (a snapshot of the synced objects) for referential templates; Undefined: a
  // reference to data.* is refused (Unsupported)
  FP compile(const Value& parameters, int* next_quant, const Value& inventory = Value()) const;

  // AOT, for the audit's RESULT totals (pkg/audit/manager.go:893-904 counts types.Results, not violating pairs): a formula
  // that is FALSE only if the template yields AT MOST ONE result for the review -- i.e. true whenever two different
  // members of the violation set may exist: two rule bodies / unrolled parameter alternatives hold at once, or one body holds
  // for two bindings of a review iteration.  Conservative by construction (it counts bindings, and equal messages of
  // different bindings collapse in the set): a pair it flags is rendered on the host and counted exactly; a violating pair
  // it does not flag has exactly one result.  Implies compile()'s formula.
  FP compile_multi(const Value& parameters, int* next_quant, const Value& inventory = Value()) const;

  // AOT, round 4: the result COUNT on the device.  compile() and compile_multi() in one evaluation, plus what is needed to count
  // the members of the violation set without rendering them: per branch of the set (rule body x unrolled parameter alternative)
  // its condition, its open review iteration and the head of its message.  count_forms() turns that into
  //     rows[]   boolean formulas whose TRUE values, summed over the rows, are the number of results of a review, and
  //     flag     "cannot tell": the pair is rendered on the host (as every flagged pair of compile_multi was)
  // -- exact whenever `flag` is false.  The argument: a branch with ONE open iteration over array elements whose message starts
  // "<literal><%v of a leaf of the element><literal>.." yields one result per firing element PROVIDED those leaves are strings
  // that do not hold the first character of the literal behind them (the message then parses back into the leaf: different
  // leaves, different messages) and no two of them are equal (the flattener marks a review in which a key-registered leaf value
  // repeats: review.$dup); thresholds "at least k elements fire" (E_k) count them.  Two branches never yield the same message
  // when their literal heads are not prefixes of one another, when they are constants that differ, when they differ in exactly
  // one constant operand of the same format, or when both are keyed on such leaves (all distinct, see above).  Anything else
  // that can hold together is flagged.
  struct CountBranch {
    FP any, two, body;            // yields a result | may yield two (compile_multi) | the iteration's body (nq == 1)
    int nq = 0, q = -1;           // open review iterations; the quantifier of the only one
    SPath base;                   // ... and what it iterates
    bool is_const = false, head = false, sep_tail = false, keyed = false;
    std::string text, pre, sep;   // constant message | literal head | literal behind the key operand
    SPath key;                    // the key operand's leaf (keyed: a leaf of the iteration's element)
    std::vector<std::string> sig;
    // the member's `details` (the set keeps {msg, details} members apart that differ in details only, pkg/audit/manager.go:902 counts
    // one result each): "C<term>" a constant ("C{}" when the member has none: the driver reports {}), "?" anything that depends on the review
    std::string dsig = "C{}";
  };
  struct CountInfo { FP viol; std::vector<CountBranch> br; };
  struct CountForms { bool ok = false; FP flag; FP viol /* the violation formula again, as the OR over the pinned + merged branches: the same truth table in a
                                                          fraction of the nodes when hundreds of unrolled alternatives print one message */; std::vector<FP> rows; std::vector<SPath> keys; std::vector<uint32_t> firsts /* rows[i] of the branches WITHOUT an iteration (the flag holds the bodies of the others) */; };
  CountInfo compile_all(const Value& parameters, int* next_quant, const Value& inventory = Value()) const;
  static CountForms count_forms(const CountInfo& ci, int kmax);

  // Host rendering: the violation set for a concrete review document (input.review) and parameters.
  // `inventory` is data.inventory (may be Undefined).  Throws RegoError on evaluation errors.
  std::vector<Violation> render(const Value& review, const Value& parameters, const Value& inventory) const;

  bool references_inventory() const { return uses_data_; }
  ~Template();   // drops the deep-expression closures registered for this template (dexpr.hpp)

  // (ceval.cpp) the concrete evaluator render() tries first: the violation set as a Value; false = not evaluated there
  struct CIndex;
  bool render_fast(const Value& review, const Value& parameters, const Value& inventory, Value* set) const;

 private:
  friend class PE;
  const CIndex& cindex() const;   // what is known about the terms before any review is seen, built on first use
  void drop_cindex();
  mutable std::once_flag cindex_once_;
  mutable CIndex* cindex_ = nullptr;
  std::vector<Module> modules_;
  std::map<std::pair<std::string, std::string>, std::vector<const Rule*>> rules_;   // (pkg, name)
  std::string pkg_name_;
  std::map<const Rule*, std::string> rule_pkgs_;        // rule -> its package / module (what every evaluation used to rebuild)
  std::map<const Rule*, const Module*> rule_mods_;
  bool uses_data_ = false;
  mutable std::vector<std::string> deep_fns_;   // names under which closures over this template sit in the deep-expression registry
};

}  // namespace gk
