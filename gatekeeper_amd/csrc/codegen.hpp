// Plan-specialised source generation (see codegen.cpp).
#pragma once
#include <string>
#include <vector>

#include "lower.hpp"
#include "vm_core.hpp"

namespace gk {
// HIP/C++ text defining gk::jit_row and gk::jit_formulas for this plan (to be compiled after plan.hpp + vm_core.hpp).
// `parts`: formula shares per 64-review half of the kernel geometry the source is compiled for (gk_parts_of(rpt))
std::string generate_plan_source(const HostPlan& plan, uint32_t parts = GK_PARTS_MIN_RPT);
// result slots per kind the plan-specialised kernel keeps per 64-review half (16, then steps of four up to GK_MAX_RES)
uint32_t jit_res_k(const HostPlan& plan);
// path table for the generated dispatch: ptab_class[path] = predicate-list class id (0 = none)
std::vector<uint32_t> jit_path_classes(const HostPlan& plan, std::vector<std::vector<Pred>>* classes);
}  // namespace gk
