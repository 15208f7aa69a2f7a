// Plan-specialised source generation (see codegen.cpp).
#pragma once
#include <string>
#include <vector>

#include "lower.hpp"
#include "vm_core.hpp"

namespace gk {
// HIP/C++ text defining gk::jit_row and gk::jit_formulas for this plan (to be compiled after plan.hpp + vm_core.hpp).
std::string generate_plan_source(const HostPlan& plan);
// path table for the generated dispatch: simple paths -> (first << 8 | count) into the descriptor table `descs`,
// complex paths -> GK_ENT_COMPLEX | class id (classes[id] = its predicate list); 0 = no predicates
std::vector<uint32_t> jit_path_classes(const HostPlan& plan, std::vector<std::vector<Pred>>* classes, std::vector<uint32_t>* descs);
}  // namespace gk
