// The text handed to hiprtc for a plan-specialised build of the dominant kernel (kernels.hip jit_build), and the
// row-group geometry it is built for.  Host-only and free of HIP calls, so the GPU-less test build can produce the very
// same text and put it through hiprtc in the build container (tests/test_jit_source.py): a plan whose generated source
// does not compile for gfx950 is caught before it reaches a GPU box.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "codegen.hpp"
#include "lower.hpp"
#include "plan.hpp"

namespace gk {

constexpr size_t GK_LDS_PER_CU = 160 * 1024;

// threads per row group of the plan-specialised kernel: the geometry table of plan.hpp, or GK_JIT_BLOCK (tuning aid: more
// waves per group, e.g. 512 threads for 128-review groups = three 8-wave groups per CU)
inline int jit_block_of(uint32_t rpt) {
  static const int forced = getenv("GK_JIT_BLOCK") ? atoi(getenv("GK_JIT_BLOCK")) : 0;
  if (forced >= (int)rpt && forced <= 1024 && forced % (int)rpt == 0 && forced % GK_TILE == 0) return forced;
  return gk_block_of((int)rpt);
}
// static LDS of the dominant kernel for a row-group geometry (kernel_body.inc: two chunk-list buffers; the result words of
// phase 2 alias one of them when they fit; the generic build also keeps the waves' loop bounds there).  res_k = 0: the
// generic bytecode build.
inline uint32_t list_cap_of(int block) { return (uint32_t)std::min(block / GK_TILE, 8) * GK_WAVE_CHUNKS; }
// list capacity a plan-specialised build is compiled for: what the table's longest list needs, in steps of 128 (round 6).  With the
// trimmed lists a FOURTH row group is resident per CU at configs[2] / [3] (34 KB of accumulators per 256-review group) and the kernel is
// built for 8 waves per SIMD (64 VGPRs).  History: round 3 measured that slower (0.133 against 0.1235 ms, spills), round 5 again
// (0.0695 against 0.0680); on round 6's kernel -- a third fewer rows per group, the list shorter -- it is faster: 1 M objects 0.0596 /
// 0.0588 / 0.0603 against 0.0605 / 0.0606 / 0.0602 ms per step, the 10 M-object table 0.530 against 0.568 ms
// (profiles/r06_variants_k_four_groups_per_cu.log).  GK_JIT_LIST_TRIM=0: the geometry's full capacity (A/B aid).
inline uint32_t jit_list_cap(int block, uint32_t need) {
  static const bool trim = !(getenv("GK_JIT_LIST_TRIM") && atoi(getenv("GK_JIT_LIST_TRIM")) == 0);
  const uint32_t full = list_cap_of(block);
  if (!trim || need == 0) return full;
  return std::min(full, std::max<uint32_t>(128u, (need + 127u) / 128u * 128u));
}
// words of the plan-specialised kernel's per-constraint totals in LDS (kernel_body.inc GK_TOT_K): the plan's constraints rounded up to
// 64; 0 = the build carries none (more than 256 constraints, or GK_FUSED_TOTALS=0 -- A/B aid: the popcount kernel behind every sweep)
inline uint32_t jit_tot_k(uint32_t n_constraints) {
  static const bool on = !(getenv("GK_FUSED_TOTALS") && atoi(getenv("GK_FUSED_TOTALS")) == 0);
  return on && n_constraints && n_constraints <= 256u ? (n_constraints + 63u) / 64u * 64u : 0u;
}
// ... unless the array costs the kernel a resident row group per CU.  MEASURED on the MI355X (profiles/r06_variants_a{q,r,u}_*.log):
// configs[2] -- 34 816 B of accumulators + 4 608 B of static LDS per 256-review group = 39 424 B -- keeps four groups per CU resident;
// EIGHT bytes more (39 432 B) and the fourth group of every CU runs behind the other three: the persistent grid of 1 024 workgroups
// takes 25 % longer (0.0442 -> 0.0540 ms; 10 M objects 0.410 -> 0.492), whatever the eight bytes are used for.  4 x 39 424 = 157 696 B
// is therefore what this file takes as a CU's LDS when it decides whether the totals array fits (the runtime's own limit, 160 KiB,
// sizes everything else as before): a plan whose exact footprint would lose a group to the array keeps the popcount kernel.
constexpr size_t GK_LDS_PER_CU_MEASURED = 157696;
inline size_t jit_static_lds_exact(uint32_t rpt, uint32_t res_k, uint32_t list_cap, uint32_t tot_k) {   // kernel_body.inc: s_work, s_masks_own (only when the result words do not alias a list buffer), s_slw, s_tot
  const size_t list = (size_t)list_cap * 8, masks = (size_t)(rpt / GK_TILE) * res_k * 8;
  return 2 * list + (masks <= list ? 0 : masks) + GK_TILE * 4 + (size_t)tot_k * 4;
}
inline uint32_t jit_tot_k(uint32_t n_constraints, size_t dyn_lds, uint32_t rpt, int block, uint32_t res_k, uint32_t list_cap) {
  const uint32_t tk = jit_tot_k(n_constraints);
  if (!tk || !list_cap) return tk;
  const size_t by_waves = std::max<size_t>(1, 32 / (size_t)std::max(1, block / GK_TILE));   // (8 waves per SIMD: more groups than that are never resident)
  const size_t with = std::min(by_waves, GK_LDS_PER_CU_MEASURED / (dyn_lds + jit_static_lds_exact(rpt, res_k, list_cap, tk)));
  const size_t without = std::min(by_waves, GK_LDS_PER_CU_MEASURED / (dyn_lds + jit_static_lds_exact(rpt, res_k, list_cap, 0)));
  return with < without ? 0u : tk;
}
inline size_t static_lds_of(uint32_t rpt, int block, uint32_t res_k, uint32_t list_cap = 0, uint32_t tot_k = 0) {
  const size_t list = (size_t)(list_cap ? list_cap : list_cap_of(block)) * 8;
  const size_t masks = (size_t)(rpt / GK_TILE) * (res_k ? res_k : (uint32_t)(GK_MAX_VIOL + 2 * GK_MAX_RES)) * 8;   // res_k: result words per half (jit_res_k)
  const size_t bounds = res_k ? 0 : (size_t)(block / GK_TILE) * GK_MAX_SCOPES * 4;
  return 2 * list + (masks <= list ? 8 : masks) + bounds + GK_TILE * 4 + 64 + (size_t)tot_k * 4;
}
inline size_t max_dyn_lds_of(uint32_t rpt) { return GK_LDS_PER_CU - static_lds_of(rpt, gk_block_of((int)rpt), 0) - 256; }
inline size_t max_dyn_lds_jit(uint32_t rpt, uint32_t res_k, uint32_t list_cap = 0, uint32_t tot_k = 0) { return GK_LDS_PER_CU - static_lds_of(rpt, jit_block_of(rpt), res_k, list_cap, tot_k) - 256; }

// A finished formula of the staged parts (generated GK_RES(kind, slot, b)): one ballot turns the 64 reviews' answers into the slot's
// bitmap word of this half.  The words stay in the WAVE until the part is through -- lane s of a register pair per kind holds slot s's
// word (v_writelane, two vector operations per result) -- and leave it with one predicated 8-byte LDS store per kind (GK_RES_FLUSH,
// the part's slots as constants from the generator): lane s = slot s is the layout the output stage reads (kernel_body.inc, s_masks).
// Before: every result was an exec-mask round trip around a one-lane LDS store (s_and_saveexec, two moves, ds_write_b64, s_or),
// ~25 times per wave and row group.  The text is shared with the test-only kernel emulator, which supplies its own GK_LANE_ID /
// GK_WRITELANE2 (tests/native/hostemu.cpp).  GK_JIT_RES_LANES=0 (A/B aid) keeps the one-lane stores.
inline std::string jit_res_macros() {
  static const bool lanes = !(getenv("GK_JIT_RES_LANES") && atoi(getenv("GK_JIT_RES_LANES")) == 0);
  // where a kind's slots start in the half's result words: kind 0 = violation slots 0..63, 1 = match, 2 = error, 3 / 4 / 5 = violation
  // slots 64.. / 128.. / 192.. (round 6: up to GK_MAX_VIOL violation formulas per plan, one register pair per bank of 64)
  // GK_BIT(b): bit 0 of a formula value, read through an opaque copy.  hiprtc for gfx950 (ROCm 7.2) folds the generated `& 1u` masks of a
  // chain like  b = !bit1 | (bit2 & bit1 & !bit3)  over (g >> k) terms into ONE v_bitop3_b32 over the UNMASKED shifts and then tests the
  // whole register (v_cmp_ne_u32 0, v): the higher bits of the accumulator word leak into the answer -- device fuzz seeds 9820 / 9833,
  // a template flagged for every review, right on the bytecode kernel, the CPU build and the emulator (DESIGN.md section 11,
  // profiles/r06_device_fuzz_ba_bk_*.log, tools/scratch/seed_9820_pair_plan_text.hip).  Behind the empty asm the compiler knows nothing
  // about the value: the mask and the test are real instructions.  The kernel emulator defines GK_BIT itself (tests/native/hostemu.cpp).
  const std::string bit =
      "#ifndef GK_BIT\nstatic __device__ inline uint32_t gk_bit(uint32_t b) { asm volatile(\"\" : \"+v\"(b)); return b & 1u; }\n#define GK_BIT(b) gk_bit(b)\n#endif\n";
  const std::string base = bit +
      "#define GK_RES_BASE(kind) ((kind) == 0 ? 0u : (kind) == 1 ? (uint32_t)GK_RES_KV : (kind) == 2 ? (uint32_t)(GK_RES_KV + GK_RES_KM) : ((uint32_t)(kind) - 2u) * 64u)\n";
  if (!lanes)
    return base +
           "#define GK_RES_PROLOGUE const bool gk_l0 = GK_LANE_ID() == 0u;\n"
           "#define GK_RES(kind, slot, b) do { const unsigned long long m_ = __ballot(GK_BIT(b) != 0u); if (gk_l0) masks[GK_RES_BASE(kind) + (slot)] = m_; } while (0)\n"
           "#define GK_RES_FLUSH(m0, m1, m2, m3, m4, m5)\n";
  std::string pro = "#define GK_RES_PROLOGUE uint32_t", flush = "#define GK_RES_FLUSH(m0, m1, m2, m3, m4, m5) do { const uint32_t l_ = GK_LANE_ID() & 63u; ", voids;
  for (int k = 0; k < 2 + GK_VIOL_WORDS; k++) {
    const std::string ks = std::to_string(k);
    pro += std::string(k ? "," : "") + " gk_rl" + ks + " = 0u, gk_rh" + ks + " = 0u";
    flush += "if (((unsigned long long)(m" + ks + ") >> l_) & 1ull) masks[GK_RES_BASE(" + ks + ") + l_] = ((unsigned long long)gk_rh" + ks + " << 32) | gk_rl" + ks + "; ";
    voids += "(void)gk_rl" + ks + "; (void)gk_rh" + ks + "; ";
  }
  return base + pro + ";\n"
         "#define GK_RES(kind, slot, b) do { const unsigned long long m_ = __ballot(GK_BIT(b) != 0u); GK_WRITELANE2(m_, slot, gk_rl##kind, gk_rh##kind); } while (0)\n" +
         flush + voids + "} while (0)\n";
}

// plan_hpp / vm_core_hpp / kernel_body: plan.hpp, vm_core.hpp and kernel_body.inc as text (build/jit_sources.inc)
inline std::string assemble_jit_source(const HostPlan& plan, uint32_t rpt, uint32_t rpp, const char* plan_hpp,
                                       const char* vm_core_hpp, const char* kernel_body, uint32_t list_cap = 0) {
  std::string src =
      "typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t; typedef unsigned long long uint64_t;\n"
      "typedef short int16_t; typedef int int32_t; typedef long long int64_t;\n";
  src += plan_hpp;
  src += vm_core_hpp;
  src += "#define GK_LANE_ID() __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))\n"
         // (v_writelane_b32: lane `l` -- an immediate -- of a register takes a wave-uniform value; hiprtc's clang has no builtin for it.
         //  gfx950 wants TWO wait states between a vector instruction that writes an SGPR -- the compare whose mask is the ballot --
         //  and a vector instruction that reads it; the compiler's hazard recogniser does not look into inline assembly, so the s_nop
         //  is part of the text.  Without it the device read stale masks now and then: 1 805 2xx instead of 1 809 882 violating pairs
         //  of configs[2], a different count every run -- profiles/r05_variants_t_result_lanes.log.)
         "static __device__ inline void gk_writelane2(unsigned long long m, const uint32_t l, uint32_t& lo, uint32_t& hi) {\n"
         "  asm(\"s_nop 1\\n\\tv_writelane_b32 %0, %2, %4\\n\\tv_writelane_b32 %1, %3, %4\" : \"+v\"(lo), \"+v\"(hi) : \"s\"((uint32_t)m), \"s\"((uint32_t)(m >> 32)), \"n\"(l)); }\n"
         "#define GK_WRITELANE2(m, l, lo, hi) gk_writelane2((m), (l), (lo), (hi))\n";
  src += jit_res_macros();
  const int block = jit_block_of(rpt);
  src += generate_plan_source(plan, (uint32_t)(block / GK_TILE / ((int)rpt / GK_TILE)));
  if (block != gk_block_of((int)rpt)) src += "#define GK_BLOCK_K " + std::to_string(block) + "\n";
  {
    // register budget: as many waves per SIMD as the LDS footprint lets groups be resident per CU (waves per SIMD =
    // groups per CU x waves per group / 4 SIMDs); measured on configs[1] with 64-review groups: 7 waves (72 VGPRs) edges
    // out 8 (64 VGPRs, twice the spill traffic) and clearly beats 5-6
    const size_t per_group = (size_t)plan.dims.acc_words * rpp * 4 + static_lds_of(rpt, block, jit_res_k(plan), list_cap, jit_tot_k(plan.dims.n_constraints, (size_t)plan.dims.acc_words * rpp * 4, rpt, block, jit_res_k(plan), list_cap ? list_cap : list_cap_of(block)));
    const size_t groups_per_cu = std::max<size_t>(1, GK_LDS_PER_CU / per_group);
    int waves = (int)std::min<size_t>(8, std::max<size_t>(block / 256, groups_per_cu * (block / GK_TILE) / 4));   // waves per SIMD the LDS allows
    if (const char* w = getenv("GK_JIT_WAVES")) waves = atoi(w);   // tuning aid
    src += "#define GK_TILES_BOUNDS __launch_bounds__(" + std::to_string(block) + ", " + std::to_string(waves) + ")\n";
  }
  {
    // chunks per batch (= loads in flight per wave) in phase 1.  Measured on configs[2] (profiles/r02_prefetch_ad.log): the
    // time to stream the rows does not depend on the depth (1..5) -- 0.111 ms with the row evaluation switched off in every
    // case -- and every extra chunk in flight costs registers and moves: 0.1755 / 0.186 / 0.190 / 0.196 / 0.210 ms for 1..5.
    // Phase 1 is bound by instruction issue, not by load latency: ONE chunk ahead.
    int depth = 1;
    if (const char* d = getenv("GK_JIT_PREFETCH")) depth = std::max(1, std::min(8, atoi(d)));   // tuning aid
    src += "#define GK_PREFETCH " + std::to_string(depth) + "\n";
  }
  {
    // wave priorities per phase (kernel_body.inc GK_PRIO_LEVELS, round 6): phase 1 and the output / request / clearing stage at
    // priority 3, bounds and formulas at 0 -- configs[2] 0.0526 -> 0.0499 ms per sweep, the 10 M-object table 0.499 -> 0.468
    // (profiles/r06_variants_x_*.log).  GK_JIT_PRIO: tuning aid (four digits; 0 = off).
    int prio = 3003;
    if (const char* pm = getenv("GK_JIT_PRIO")) prio = std::max(0, std::min(3333, atoi(pm)));
    if (prio) src += "#define GK_PRIO_LEVELS " + std::to_string(prio) + "\n";
    // ... and inside the formulas the share of wave 0 of every half -- the one that goes on to write the violation words -- one level
    // above the other share(s): 10 M objects 0.457 -> 0.440 ms, 3 M 0.140 -> 0.135, configs[2] 0.0482 -> 0.0465 (profiles/r06_variants_a{b,c,d}_*.log)
    int part0 = 1;
    if (const char* pp = getenv("GK_JIT_PRIO_PART0")) part0 = std::max(0, std::min(3, atoi(pp)));
    if (prio && part0) src += "#define GK_PRIO_PART0 " + std::to_string(part0) + "\n";
  }
  if (getenv("GK_KERNEL_PROF") || getenv("GK_DBG_PHASE")) src += "#define GK_WITH_PROF 1\n";   // (kernel_body.inc: the phase marks and switches, only when asked for)
  src += "#define GK_RPT_K " + std::to_string(rpt) + "\n#define GK_RPP_K " + std::to_string(rpp) + "\n";
  if (list_cap) src += "#define GK_LIST_CAP_K " + std::to_string(list_cap) + "\n";
  if (const uint32_t tk = jit_tot_k(plan.dims.n_constraints, (size_t)plan.dims.acc_words * rpp * 4, rpt, block, jit_res_k(plan), list_cap ? list_cap : list_cap_of(block))) src += "#define GK_TOT_K " + std::to_string(tk) + "\n";   // (kernel_body.inc: per-constraint totals left as one row per workgroup)
  if (const char* defs = getenv("GK_JIT_DEFINES")) {   // tuning aid: "A=1;B" -> #define A 1, #define B (kernel_body.inc variants)
    std::string d = defs, item;
    for (size_t i = 0; i <= d.size(); i++) {
      if (i < d.size() && d[i] != ';') { item.push_back(d[i]); continue; }
      if (!item.empty()) { size_t eq = item.find('='); src += "#define " + (eq == std::string::npos ? item : item.substr(0, eq) + " " + item.substr(eq + 1)) + "\n"; }
      item.clear();
    }
  }
  src += "namespace gk {\n#define GK_KERNEL_TILES gk_jit_tiles\n#define GK_KERNEL_BIG gk_jit_big\n#define GK_KERNEL_LINKAGE extern \"C\"\n#define GK_SKIP_BIG\n"
         "#define GK_ROW_FN(r, i, ent, h, pv, heap, acc, on) jit_row(r, ent, h, heap, acc, on)\n#define GK_BIND_ALWAYS_STR 0\n"
         "#define GK_FORMULA_FN(pv, acc, flags, rows, heap, bounds) jit_formulas(pv, acc, flags, rows, heap, bounds)\n";
  if (const char* bf = getenv("GK_JIT_BODY_FILE")) {   // tuning aid: A/B a variant of kernel_body.inc without rebuilding the library
    FILE* f = fopen(bf, "r");
    if (!f) throw std::runtime_error(std::string("GK_JIT_BODY_FILE: cannot open ") + bf);
    std::string body, line;
    char buf[4096];
    while (fgets(buf, sizeof buf, f)) { line = buf; if (line.rfind("#include", 0) != 0 && line.rfind("#pragma once", 0) != 0) body += line; }
    fclose(f);
    src += body;
  } else src += kernel_body;
  src += "}\n";
  return src;
}

}  // namespace gk
