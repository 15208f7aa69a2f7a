// Lowering: quantified boolean formulas (pe.hpp) -> device plan (plan.hpp): predicate table, element scopes,
// formula bytecode, constant heap; plus the Match-block compiler.
//
// Match blocks restate pkg/mutation/match/match.go:32-258 + pkg/target/matcher.go:44-71 as formulas over the
// flattener's `$m` / `$ns` rows and RF_* review flags, so that the match layer runs in the same kernels as the
// template predicates (K1 "match_filter" of SURVEY.md section 7.3 is fused into the predicate pass).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "flatten.hpp"
#include "pe.hpp"
#include "plan.hpp"

#include <atomic>
namespace gk {

struct MatchFormulas {
  FP match;   // constraint applies to the review (object OR oldObject, matcher.go:44-71)
  FP error;   // Matcher.Match returns an error (autoreject result)
};
// spec.match of a constraint (Undefined / null => match everything, target.go:246-261)
MatchFormulas compile_match(const Value& match_spec);
// error text of metav1.LabelSelectorAsSelector for a selector ("" = valid); same validation compile_match applies
std::string selector_error_text(const Value& selector);

struct PlanCaps {
  uint16_t level_cap[3] = {8, 12, 12};   // element capacity per array-nesting level
  std::vector<uint16_t> scope_cap;       // optional: capacity per element scope (table-specialised variants), overrides level_cap
};

struct HostPlan {
  std::vector<Pred> preds;
  std::vector<Pattern> pred_patterns;     // parallel to preds
  std::vector<Scope> scopes;
  std::vector<uint32_t> code;
  std::vector<uint32_t> seg_ends;         // code offsets closing each self-contained block (derived-bit blocks, then one per result)
  std::vector<uint8_t> cheap;
  std::vector<ConstraintSlot> slots;      // per constraint
  uint32_t n_viol = 0, n_match = 0;       // unique formulas
  PlanDims dims{};
  // path table (depends on the dictionary contents at build time)
  std::vector<uint32_t> ptab;
  std::vector<Pred> path_preds;           // predicates grouped by path id (what the device reads)
  uint32_t dict_size = 0;
  void resolve_paths(const PathDict& dict);   // (re)builds ptab / pred_list for the current dictionary
};

// A constraint's formulas after the capacity-independent passes (simplify, pin_pass, fold_dict) and their structural keys:
// computed once when the constraint is added, shared by every plan the constraint is lowered into (the default plan, the
// big-capacity plan, per-table variants, constraint groups).
// while one is alive on this thread, prepare_constraint's passes remember what they made of every formula NODE (lower.cpp PrepMemo)
struct PrepMemoScope { PrepMemoScope(); ~PrepMemoScope(); PrepMemoScope(const PrepMemoScope&) = delete; private: void* prev_; void* mine_; };
struct PreparedConstraint { FP viol, match, error; std::string viol_key, match_key; };
std::shared_ptr<const PreparedConstraint> prepare_constraint(const FP& violation, const MatchFormulas& m);

class PlanBuilder {
 public:
  // frozen: the plan must live with the registry entries (dictionary predicates, guards, value-id paths) the loaded constraints
  // made -- build() throws Unsupported instead of adding one (the totals plans: they must not change what the flattener does)
  explicit PlanBuilder(PathDict* dict, DictRegistry* reg = nullptr, bool frozen = false) : dict_(dict), reg_(reg), frozen_(frozen) {}
  // returns the constraint index; formulas are deduplicated structurally
  uint32_t add_constraint(const FP& violation, const MatchFormulas& m) { return add_constraint(prepare_constraint(violation, m)); }
  uint32_t add_constraint(std::shared_ptr<const PreparedConstraint> pc) { cons_.push_back(std::move(pc)); return (uint32_t)cons_.size() - 1; }
  HostPlan build(const PlanCaps& caps);   // throws Unsupported
  // the plan's dictionary predicates live in the registry's COUNTING space (<leaf>.$c rows; flatten.hpp): the result-counting plans
  void use_counting_space(bool on = true) { counting_ = on; }

 private:
  PathDict* dict_;
  DictRegistry* reg_;
  bool frozen_ = false, counting_ = false;
  std::vector<std::shared_ptr<const PreparedConstraint>> cons_;
};


bool review_fact_leaf(const SPath& p);   // a leaf of the review facts row (lower.cpp REVIEW FACTS)
// test aid (gk_debug_set "fold_match_labels"): match formulas' label tests become dictionary bits as well
extern std::atomic<int> g_test_fold_match_labels;

}  // namespace gk
