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
// result words the plan-specialised kernel keeps per 64-review half: jit_res_kv violation slots + 2 x jit_res_km match / error slots
// (each in steps of four; kernel_body.inc GK_RES_KV / GK_RES_KM)
uint32_t jit_res_kv(const HostPlan& plan);
uint32_t jit_res_km(const HostPlan& plan);
uint32_t jit_res_k(const HostPlan& plan);
// path table for the generated dispatch: ptab_class[path] = predicate-list class id (0 = none)
std::vector<uint32_t> jit_path_classes(const HostPlan& plan, std::vector<std::vector<Pred>>* classes);
}  // namespace gk
