// Plan -> HIP source: the predicate dispatch (phase 1) and the formulas (phase 2) of ONE compiled plan as straight-line
// code over the same primitives the interpreter uses (vm_core.hpp: eval_pred with a constexpr Pred folds to the single
// operation; boolean registers become locals; loops become real loops with constant LDS offsets).  kernels.hip
// compiles the result for gfx950 with hiprtc when the plan is uploaded; tests/native/hostemu.cpp can compile the same
// text with g++ to validate the generator in the GPU-less container.
#include "codegen.hpp"
#include "chunks.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <functional>
#include <sstream>

namespace gk {

namespace {

std::string u(uint64_t v) { return std::to_string(v) + "u"; }

std::string pred_literal(const Pred& p) {
  std::ostringstream o;
  o << "Pred{" << (int)p.op << "," << (int)p.dst << "," << (int)p.scope << "," << (int)p.level << "," << p.bit << "," << (int)p.cmp << ","
    << (int)p.ctype << "," << p.a << "u," << p.b << "u," << p.k << "ull," << p.idx << "," << p.pad << "u}";
  return o.str();
}

}  // namespace

std::vector<uint32_t> jit_path_classes(const HostPlan& plan, std::vector<std::vector<Pred>>* classes) {
  // distinct predicate lists -> class ids (1-based); entry[path] = class id | GK_ENT_NEEDS_STR, 0 = no predicates
  std::vector<uint32_t> out(plan.ptab.size(), 0);
  std::map<std::string, uint32_t> ids;
  classes->clear();
  classes->push_back({});
  for (size_t i = 0; i < plan.ptab.size(); i++) {
    uint32_t ent = plan.ptab[i];
    if (!ent) continue;
    uint32_t first = ent >> 8, cnt = ent & 0xFF;
    std::string key((const char*)&plan.path_preds[first], cnt * sizeof(Pred));
    bool str = false;
    for (uint32_t j = 0; j < cnt; j++) str = str || pred_needs_str(plan.path_preds[first + j]);
    auto it = ids.find(key);
    uint32_t id;
    if (it == ids.end()) {
      id = (uint32_t)classes->size();
      ids[key] = id;
      classes->emplace_back(plan.path_preds.begin() + first, plan.path_preds.begin() + first + cnt);
    } else id = it->second;
    out[i] = id | (str ? GK_ENT_NEEDS_STR : 0u);
  }
  // CANONICAL numbering: by the predicate lists themselves, not by the order in which key paths got their ids -- the host threads
  // of the first table's ingest intern paths in whatever order they meet them, and the generated text (hence the code-object
  // cache key, on disk too) must not depend on that: the same policy set over the same objects is the same kernel in every process
  std::vector<uint32_t> renum(classes->size(), 0);
  {
    uint32_t next = 1;
    for (auto& kv : ids) renum[kv.second] = next++;   // (std::map: ascending by the lists' bytes)
    std::vector<std::vector<Pred>> sorted(classes->size());
    for (size_t c = 1; c < classes->size(); c++) sorted[renum[c]] = std::move((*classes)[c]);
    classes->swap(sorted);
  }
  for (auto& e : out) if (e) e = renum[e & ~GK_ENT_NEEDS_STR] | (e & GK_ENT_NEEDS_STR);
  return out;
}

// result slots the plan-specialised kernel keeps per 64-review half: KV violation slots, KM match slots and KM error slots, in that
// order (kernel_body.inc s_masks).  In steps of four: 4 halves x (20 + 2 x 8) slots of configs[2] fit the 256-entry chunk-list buffer
// they alias.
uint32_t jit_res_kv(const HostPlan& plan) { return std::min<uint32_t>((uint32_t)GK_MAX_VIOL, std::max<uint32_t>(4u, (plan.n_viol + 3u) / 4u * 4u)); }
uint32_t jit_res_km(const HostPlan& plan) { return std::min<uint32_t>((uint32_t)GK_MAX_RES, std::max<uint32_t>(4u, (plan.n_match + 3u) / 4u * 4u)); }
uint32_t jit_res_k(const HostPlan& plan) { return jit_res_kv(plan) + 2u * jit_res_km(plan); }   // result words per half

std::string generate_plan_source(const HostPlan& plan, uint32_t parts) {
  std::ostringstream o;
  std::vector<std::vector<Pred>> classes;
  jit_path_classes(plan, &classes);
  // (GK_BIT: bit 0 of a formula value; the device text defines it as an opaque copy + mask BEFORE this source -- jit_source.hpp jit_res_macros,
  //  where the reason is written down; anything else that compiles the plan source gets the plain mask)
  o << "#ifndef GK_BIT\n#define GK_BIT(b) ((b) & 1u)\n#endif\n";
  o << "namespace gk {\n";
  // the plan's constant heap as a constant-initialised array: with constexpr predicates every constant-string load has
  // a compile-time address, so the optimiser folds the bytes into immediates (no memory traffic for constants)
  o << "GK_CONST_ARRAY unsigned char gk_plan_consts[" << plan.cheap.size() << "] = {";
  for (size_t i = 0; i < plan.cheap.size(); i++) o << (i ? "," : "") << (int)plan.cheap[i];
  o << "};\n";
  {   // accumulator words that must start at zero: all of them (an empty value slot is id 0)
    o << "#define GK_HAS_ZERO_RANGES 1\nconstexpr uint32_t GK_N_ZERO_RANGES = 1u;\n"
      << "GK_CONST_ARRAY uint32_t gk_zero_lo[1] = {0u};\nGK_CONST_ARRAY uint32_t gk_zero_hi[1] = {" << plan.dims.acc_words << "u};\n";
  }
  // result slots kept per 64-review half and kind (kernel_body.inc GK_RES_K), and where each scope's element count lives
  o << "#define GK_RES_KV " << jit_res_kv(plan) << "\n#define GK_RES_KM " << jit_res_km(plan) << "\n#define GK_N_SCOPES_K " << plan.scopes.size() << "\n"
    << "GK_CONST_ARRAY uint32_t gk_count_off[" << std::max<size_t>(1, plan.scopes.size()) << "] = {";
  for (size_t i = 0; i < plan.scopes.size(); i++) o << (i ? "," : "") << plan.scopes[i].count_off << "u";
  if (plan.scopes.empty()) o << "0u";
  o << "};\nGK_CONST_ARRAY uint32_t gk_scope_cap[" << std::max<size_t>(1, plan.scopes.size()) << "] = {";
  for (size_t i = 0; i < plan.scopes.size(); i++) o << (i ? "," : "") << plan.scopes[i].cap << "u";
  if (plan.scopes.empty()) o << "0u";
  o << "};\n";
  // ---------------------------------------------------------------------------------------------- phase 1
  // inlined into its single call site (the chunk loop): as a separate function every LDS atomic would first look the
  // dynamic-LDS base up in a table (s_getpc + s_load + full wait; seen in the gfx950 ISA) and the call frame costs scratch
  const bool inline_row = true;
  o << "template <class Acc>\nGK_HD __attribute__((" << (inline_row ? "always_inline" : "noinline") << ")) void jit_row(Row r, uint32_t cls, StrHdr h, const uint8_t* heap, Acc acc, bool on) {\n"
    << "  const uint8_t* cheap = gk_plan_consts;\n  (void)cheap; (void)h;\n  cls = GK_UNI(cls) & ~GK_ENT_NEEDS_STR;   // one class per call: the dispatch is a scalar branch\n";
  std::ostringstream& real_o = o;
  std::vector<std::string> case_body(classes.size());
  // One class = the predicates of one key path.  Results are gathered in one mask per destination word (a single LDS
  // atomic per word, not per predicate); integer comparisons share one type test; short string equalities compare
  // the packed payload; everything else goes through eval_pred with a constexpr predicate.
  static const char* kCmpOps[] = {"==", "!=", "<", "<=", ">", ">="};
  for (size_t c = 1; c < classes.size(); c++) {
    // the class dispatch is wave-uniform and comes FIRST; the per-lane "this lane holds a row of this pass" test sits inside
    // the case (around a divergent dispatch the structuriser threads every case exit through a chain of flow blocks)
    std::ostringstream o;   // (this class's body; assembled into the dispatch below)
    // (a branch-free form of these bodies -- predicates as selects, every LDS atomic unconditional with a neutral operand -- measured
    //  level with this one in round 3, 0.1299 against 0.1287 ms on configs[2], profiles/r03_variants_c_*.log, and was removed in round 5)
    o << "if (on) {\n      const uint32_t t = r.meta & 7u; (void)t;\n";
    const std::vector<Pred>& ps = classes[c];
    // a CARRIER class (plan.hpp T_ABSENT): the path's rows carry an element marker besides the member's own predicates, and an element
    // without the member has a row of type T_ABSENT there -- it exists for the marker alone: every other predicate of the class sees
    // "no row" (`real`), as eval_pred does
    bool mixed = false;
    for (const Pred& p : ps) if (p.op == P_PRESENT) mixed = true;
    if (mixed) o << "      const bool real = t != 7u; (void)real;\n";
    // (any other class: a T_ABSENT row may sit on its path all the same -- ANOTHER plan of the engine, or one loaded earlier, made the
    //  path a carrier -- and is no row to this class at all)
    else o << "      if (t != 7u) {\n";
    struct Group { int scope, level; bool always = false; std::vector<std::string> masks; std::vector<size_t> stores; bool present = false; };
    std::vector<Group> groups;          // element destinations by (scope, level)
    std::vector<std::string> gmasks;    // global destination words
    auto declare = [&](const std::string& name, std::vector<std::string>& list) {
      if (std::find(list.begin(), list.end(), name) == list.end()) { list.push_back(name); o << "      uint32_t " << name << " = 0u;\n"; }
    };
    auto group_of = [&](const Pred& p) -> Group& {
      for (auto& g : groups) if (g.scope == p.scope && g.level == p.level) return g;
      groups.push_back(Group{p.scope, p.level});
      return groups.back();
    };
    std::vector<std::string> target(ps.size());   // "mask |= bit" statement per predicate
    for (size_t i = 0; i < ps.size(); i++) {
      const Pred& p = ps[i];
      if (p.dst == D_GLOBAL) {
        std::string m = "mg" + std::to_string(p.bit >> 5);
        declare(m, gmasks);
        target[i] = m + " |= " + u(1u << (p.bit & 31)) + ";";
      } else {
        Group& g = group_of(p);
        if (p.op == P_STORE) { g.stores.push_back(i); g.always = true; if (p.level >= GK_LEVEL_ROOT) g.present = true; continue; }   // root scope: a store marks its element
        if (p.op == P_PRESENT) { g.present = true; g.always = true; continue; }
        std::string m = "me" + std::to_string(p.scope) + "_" + std::to_string(p.level) + "_" + std::to_string(elem_word_of_bit(p.bit));
        declare(m, g.masks);
        target[i] = m + " |= " + u(elem_mask_of_bit(p.bit)) + ";";
        if (p.op == P_DEFINED) g.always = true;
      }
    }
    // integer comparisons: one type test for all of them
    std::vector<size_t> icmp;
    for (size_t i = 0; i < ps.size(); i++) if (ps[i].op == P_CMP && ps[i].ctype == T_INT && !target[i].empty()) icmp.push_back(i);
    if (!icmp.empty()) {
      o << "      if (t == T_INT) {\n        const int64_t a = row_i64(r);\n";
      for (size_t i : icmp) o << "        if (a " << kCmpOps[ps[i].cmp] << " " << (long long)(int64_t)ps[i].k << "ll) " << target[i] << "\n";
      o << "      } else {\n";
      for (size_t i : icmp) o << "        { constexpr Pred P = " << pred_literal(ps[i]) << "; if (eval_pred(r, P, h, heap, cheap)) " << target[i] << " }\n";
      o << "      }\n";
    }
    // predicates on components of split(trim(row, cut), sep): the split itself -- where the separators are, vm_core.hpp
    // SplitMask -- is computed ONCE per (cut, sep) of the class and shared by all of them (seven "banned tag" predicates on
    // containers[].image used to scan the string seven times, a byte per memory round trip)
    std::vector<uint32_t> split_pads;
    for (size_t i = 0; i < ps.size(); i++)
      if (!target[i].empty() && (ps[i].op == P_SPLIT_CMP || ps[i].op == P_SPLIT_COUNT || ps[i].op == P_SPLIT_PREFIX) &&
          std::find(split_pads.begin(), split_pads.end(), ps[i].pad) == split_pads.end()) split_pads.push_back(ps[i].pad);
    if (!split_pads.empty()) {
      o << "      const bool isstr = t == T_STRING;\n      const StrRef s_ = make_str(r, h, heap);\n";
      for (uint32_t pad : split_pads)
        o << "      SplitMask sm_" << pad << "; sm_" << pad << ".seps = 0ull; sm_" << pad << ".lo = 0u; sm_" << pad << ".hi = 0u; sm_" << pad << ".fast = true;\n"
          << "      if (isstr) sm_" << pad << " = split_mask(s_, (uint8_t)" << (pad >> 8) << "u, (uint8_t)" << (pad & 0xFFu) << "u);\n";
    }
    for (size_t i = 0; i < ps.size(); i++) {
      const Pred& p = ps[i];
      if (target[i].empty() || (p.op == P_CMP && p.ctype == T_INT)) continue;
      if (p.op == P_SPLIT_CMP || p.op == P_SPLIT_COUNT || p.op == P_SPLIT_PREFIX) {
        const std::string call = std::string(p.op == P_SPLIT_PREFIX ? "eval_split_prefix" : "eval_split_pred") + "(s_, sm_" + std::to_string(p.pad) + ", P, cheap)";
        o << "      { constexpr Pred P = " << pred_literal(p) << "; if (isstr && " << call << ") " << target[i] << " }\n";
        continue;
      }
      std::string cond;
      switch (p.op) {
        case P_DEFINED: cond = "true"; break;
        case P_TRUTHY: cond = "!(t == T_BOOL && r.lo == 0u)"; break;
        case P_TYPE: cond = "((" + u(p.ctype) + " >> t) & 1u) != 0u"; break;
        case P_BITS: cond = "t == T_INT && (r.lo & " + u((uint32_t)p.k) + ") | (r.hi & " + u((uint32_t)(p.k >> 32)) + ")"; cond = "(t == T_INT) && (((r.lo & " + u((uint32_t)p.k) + ") | (r.hi & " + u((uint32_t)(p.k >> 32)) + ")) != 0u)"; break;
        case P_COUNT_CMP: cond = std::string("(t == T_OBJECT || t == T_ARRAY) && ((int64_t)r.lo ") + kCmpOps[p.cmp] + " " + std::to_string((long long)(int64_t)p.k) + "ll)"; break;
        case P_CMP:
          if (p.ctype == T_STRING && (p.cmp == C_EQ || p.cmp == C_NE) && p.b <= 7) {
            uint64_t bits = 0;
            for (uint32_t k = 0; k < p.b; k++) bits |= (uint64_t)plan.cheap[p.a + k] << (8 * k);
            uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32) | (p.b << 24);
            cond = std::string(p.cmp == C_NE ? "!" : "") + "((r.meta & (7u | ROW_STR_INLINE)) == (4u | ROW_STR_INLINE) && r.lo == " + u(lo) + " && r.hi == " + u(hi) + ")";
          }
          break;
        case P_STR_IN_SET: {   // all members short: compare the packed payload of an inline string row
          bool all_short = p.b > 0 && p.b <= 8;
          for (uint32_t k = 0; k < p.b && all_short; k++) {
            uint32_t len; memcpy(&len, &plan.cheap[p.a + 12 * k + 8], 4);
            if (len > 7) all_short = false;
          }
          if (!all_short) break;
          cond = "(r.meta & (7u | ROW_STR_INLINE)) == (4u | ROW_STR_INLINE) && (";
          for (uint32_t k = 0; k < p.b; k++) {
            uint32_t ea, eb, len;
            memcpy(&ea, &plan.cheap[p.a + 12 * k], 4); memcpy(&eb, &plan.cheap[p.a + 12 * k + 4], 4); memcpy(&len, &plan.cheap[p.a + 12 * k + 8], 4);
            cond += std::string(k ? " || " : "") + "(r.lo == " + u(ea) + " && r.hi == " + u(eb | (len << 24)) + ")";
          }
          cond += ")";
          break;
        }
        default: break;
      }
      if (mixed && !cond.empty()) cond = cond == "true" ? "real" : "real && (" + cond + ")";
      if (cond.empty()) o << "      { constexpr Pred P = " << pred_literal(p) << "; if (eval_pred(r, P, h, heap, cheap)) " << target[i] << " }\n";
      else if (cond == "true") o << "      " << target[i] << "\n";
      else o << "      if (" << cond << ") " << target[i] << "\n";
    }
    for (const std::string& m : gmasks) o << "      if (" << m << ") acc.or_word(" << m.substr(2) << "u, " << m << ");\n";
    for (const Group& g : groups) {
      const Scope& sc = plan.scopes[g.scope];
      std::string hit = g.always ? ((mixed && !(g.present && g.level < (int)GK_LEVEL_ROOT)) ? "real" : "true") : "";   // (only the marker's own group is written for a T_ABSENT row)
      if (!g.always) for (size_t k = 0; k < g.masks.size(); k++) hit += (k ? " | " : "") + g.masks[k];
      if (!g.always) hit = "(" + hit + ") != 0u";
      o << "      if (" << hit << ") {\n        const uint32_t ord = row_ordinal(r, " << g.level << "u);\n"
        << "        if (ord >= " << sc.cap << "u || (r.meta & ROW_ORD_OVERFLOW)) acc.or_word(0u, 1u);\n        else {\n";
      std::string extra;
      // a stored value = the row's VALUE ID (plan.hpp); a row without one (non-empty container, stale table) or with the
      // overflow id cannot be compared: the review goes beyond the limits (vm_core.hpp P_STORE)
      if (!g.stores.empty()) {
        if (mixed) o << "          const uint32_t vid = real ? row_vid(r) : 0u;\n          if (real && (vid == 0u || vid >= GK_VID_OVERFLOW)) acc.or_word(0u, 1u); else {\n";
        else o << "          const uint32_t vid = row_vid(r);\n          if (vid == 0u || vid >= GK_VID_OVERFLOW) acc.or_word(0u, 1u); else {\n";
      }
      for (size_t i : g.stores) {
        const Pred& p = ps[i];
        if (scope_packed(sc)) extra += " | (vid << " + std::to_string(ELEM_VID_SHIFT) + "u)";
        else o << "          " << (mixed ? "if (real) " : "") << "acc.store_word(" << sc.val_off << "u + ord * " << val_stride(sc.nvals) << "u + " << p.bit << "u, vid);\n";
      }
      if (g.present) {
        if (g.level > 0 && g.level < (int)GK_LEVEL_ROOT) extra += " | 1u | (row_ordinal(r, " + std::to_string(g.level - 1) + "u) << 24)";
        else extra += " | 1u";
        o << "          acc.max_word(" << sc.count_off << "u, ord + 1u);\n";
      }
      bool w0_done = false;
      for (const std::string& m : g.masks) {
        uint32_t wi = (uint32_t)atoi(m.substr(m.rfind('_') + 1).c_str());
        if (wi == 0) { o << "          acc.or_word(" << sc.word_off << "u + ord * " << (int)sc.wpe << "u, " << m << extra << ");\n"; w0_done = true; }
        else o << "          if (" << m << ") acc.or_word(" << sc.word_off << "u + ord * " << (int)sc.wpe << "u + " << wi << "u, " << m << ");\n";
      }
      if (!w0_done && !extra.empty()) o << "          acc.or_word(" << sc.word_off << "u + ord * " << (int)sc.wpe << "u, 0u" << extra << ");\n";
      if (!g.stores.empty()) o << "          }\n";
      o << "        }\n      }\n";
    }
    if (!mixed) o << "      }\n";
    o << "    }\n";
    case_body[c] = o.str();
  }
  {
    // The dispatch: one `switch` over the class (a balanced compare tree: the AMDGPU backend has no jump tables).  Testing the classes
    // that own most chunks first, in an if / else-if chain, measured slower (profiles/r02_variants_f_hot_dispatch.log,
    // r03_variants_l_*.log: 0.1148 ms with the plain switch, 0.1164 / 0.1177 / 0.1205 with chains of 2 / 4 / 6) and went in round 5 --
    // with it the generated text stopped depending on the table's chunk statistics: one policy set, one kernel, one cache entry.
    std::ostringstream& o = real_o;
    o << "  ";
    o << "switch (cls) {\n";
    for (size_t c = 1; c < classes.size(); c++) {
      o << "    case " << c << ": do { " << case_body[c] << "    } while (false);\n    break;\n";
    }
    o << "    default: break;\n  }\n";
    o << "\n}\n\n";
  }
  // ---------------------------------------------------------------------------------------------- phase 2
  o << "template <class Acc>\nGK_HD Results jit_formulas(const PlanView& pv, Acc& acc, uint32_t flags, const Row* rows, const uint8_t* heap, const uint32_t* bounds) {\n"
    << "  (void)pv; (void)rows; (void)heap; (void)flags;\n  Results res = {};\n  uint32_t";
  for (int i = 0; i < 64; i++) o << (i ? ", " : " ") << "b" << i << " = 0u";
  o << ";\n";
  // the global predicate words are read once; derived global bits (F_STG) update the register copy as well
  for (uint32_t w = 0; w < plan.dims.n_gwords; w++) o << "  uint32_t g" << w << " = acc.load(" << w << "u);\n";
  struct Loop { uint32_t scope; int depth; int lit; };   // lit: the element index as a literal (preloaded form), -1: a run-time loop variable
  std::vector<Loop> stack;
  // PRELOADED form of the staged parts (round 5).  The formulas are LDS-latency bound: every loop of every formula re-reads its
  // scope's element words (an LDS round trip in front of a handful of bit operations; ~450 instructions took 10 k clocks per
  // row group).  Here a part reads each element word it needs ONCE, up front -- all reads in flight together -- into registers
  // W<scope>_<element>; loops are unrolled by the generator (every iteration a copy of the body with the element index as a
  // literal; iterations beyond the wave's largest element count are skipped by a scalar branch), derived element bits update the
  // register copy as well as LDS.  Used when the unrolled text stays small (`pre_budget` operations per plan); GK_JIT_PRELOAD=0
  // keeps the loops.
  std::ostringstream* out_ = &o;
  // result slots (per KIND) the staged part being generated hands to GK_RES.  Kinds: 0 = violation slots 0..63, 1 = match, 2 = error,
  // 3.. = violation slots 64.., 128.., 192.. (one kind per bank of 64: a kind's words ride in one register pair, lane = slot & 63)
  uint64_t res_slots[2 + GK_VIOL_WORDS] = {};
  bool pre = false;                                    // generating the preloaded form
  std::set<std::pair<uint32_t, uint32_t>> pre_words;   // (scope, element) words the part being generated reads
  std::set<uint32_t> pre_bounds;                       // scopes whose run-time bound the part needs
  std::map<std::string, std::string> pre_vals;         // unpacked value slots: register name -> its load
  size_t pre_ops = 0;
  auto var_of = [&](uint32_t scope) -> int {
    for (size_t i = stack.size(); i-- > 0;) if (stack[i].scope == scope) return stack[i].depth;
    return -1;
  };
  const std::vector<uint32_t>& code = plan.code;
  auto loop_end = [&](size_t pc) -> size_t {   // pc: first instruction of a loop body -> the index of its F_ENDLOOP / F_ENDLOOP2
    int depth = 0;
    for (;;) {
      const uint32_t op = code[pc] & 0xFF;
      if (op == F_VEQ) { pc += 2; continue; }
      if (op == F_LOOP) depth++;
      if (op == F_ENDLOOP || op == F_ENDLOOP2) { if (depth == 0) return pc; depth--; }
      if (op == F_END) throw Unsupported("codegen: loop without an end");
      pc++;
    }
  };
  // CONJUNCTION bodies (round 5).  Most loops of a compiled policy set ask "does SOME element hold bits b1 & b2 & !b3 .." -- a
  // conjunction of literals of the element's own words (after the string tests became dictionary bits nearly every container loop
  // of the 200-template corpus has that shape).  Evaluated bit by bit that is an extract per literal, a combine per literal and two
  // operations to accumulate, per element; as ONE masked compare per element word -- (w & care) == want, the element's presence bit
  // among the literals, so that the zero word of an absent element fails by itself -- it is two vector operations and a scalar OR.
  // -> care / want per word of the element (index = word), false when the body is anything but such a conjunction.
  struct Conj { std::vector<uint32_t> care, want; bool never = false; };
  auto conj_body = [&](uint32_t scope, size_t pc, size_t end, uint32_t result_reg, Conj* out) -> bool {
    struct Lit { uint32_t bit; bool pos; };
    struct Val { int kind = 0; std::vector<Lit> lits; };   // kind 0: unknown, 1: conjunction of lits, 2: constant false, 3: constant true
    std::map<uint32_t, Val> regs;
    auto conj_and = [&](const Val& x, const Val& y) -> Val {
      Val r;
      if (x.kind == 0 || y.kind == 0) return r;
      if (x.kind == 2 || y.kind == 2) { r.kind = 2; return r; }
      if (x.kind == 3) return y;
      if (y.kind == 3) return x;
      r.kind = 1; r.lits = x.lits;
      for (const Lit& l : y.lits) {
        bool dup = false;
        for (const Lit& m : r.lits) if (m.bit == l.bit) { if (m.pos != l.pos) { r.kind = 2; r.lits.clear(); return r; } dup = true; }
        if (!dup) r.lits.push_back(l);
      }
      return r;
    };
    auto neg = [&](const Val& x) -> Val {
      Val r;
      if (x.kind == 2) r.kind = 3; else if (x.kind == 3) r.kind = 2;
      else if (x.kind == 1 && x.lits.size() == 1) { r.kind = 1; r.lits = {Lit{x.lits[0].bit, !x.lits[0].pos}}; }
      return r;
    };
    while (pc < end) {
      const uint32_t ins = code[pc++];
      const uint32_t op = ins & 0xFF, a = (ins >> 8) & 0xFF, b = (ins >> 16) & 0xFF, c = ins >> 24;
      switch (op) {
        case F_LDE: { if (b != scope) return false; Val v; v.kind = 1; v.lits = {Lit{c, true}}; regs[a] = v; break; }
        case F_AND: regs[a] = conj_and(regs[b], regs[c]); break;
        case F_ANDN: regs[a] = conj_and(regs[b], neg(regs[c])); break;
        case F_NOT: regs[a] = neg(regs[b]); break;
        case F_MOV: regs[a] = regs[b]; break;
        case F_CONST: { Val v; v.kind = (b & 1) ? 3 : 2; regs[a] = v; break; }
        default: return false;   // a nested loop, a join, a derived bit, a global / flag bit, a disjunction: the general form
      }
      if (regs[a].kind == 0) return false;
    }
    const Val& body = regs[result_reg];
    if (body.kind == 0) return false;
    const Scope& sc = plan.scopes[scope];
    out->care.assign(sc.wpe, 0u); out->want.assign(sc.wpe, 0u);
    out->care[0] = 1u; out->want[0] = 1u;   // the element is present
    if (body.kind == 2) { out->never = true; return true; }
    if (body.kind == 1) for (const Lit& l : body.lits) {
      const uint32_t w = elem_word_of_bit(l.bit), m = elem_mask_of_bit(l.bit);
      if (w >= sc.wpe) return false;
      if ((out->care[w] & m) && (((out->want[w] & m) != 0) != l.pos)) { out->never = true; return true; }
      out->care[w] |= m;
      if (l.pos) out->want[w] |= m;
    }
    return true;
  };
  std::function<void(size_t, size_t, bool, std::string)> gen = [&](size_t pc0, size_t pc1, bool staged, std::string ind) {
  std::ostringstream& o = *out_;
  for (size_t pc = pc0; pc < pc1;) {
    uint32_t ins = code[pc++];
    uint32_t op = ins & 0xFF, a = (ins >> 8) & 0xFF, b = (ins >> 16) & 0xFF, c = ins >> 24;
    pre_ops++;
    switch (op) {
      case F_LDG: { uint32_t bit = b | (c << 8); o << ind << "b" << a << " = (g" << (bit >> 5) << " >> " << (bit & 31) << ") & 1u;\n"; break; }
      case F_LDF: o << ind << "b" << a << " = (flags >> " << b << ") & 1u;\n"; break;
      case F_LDE: {
        const Scope& sc = plan.scopes[b];
        int d = var_of(b);
        if (d < 0) throw Unsupported("codegen: element load outside its loop");
        // booleans are 0/1 integers in vector registers (bitwise VALU ops), not wave masks in scalar registers
        const uint32_t sh = (uint32_t)__builtin_ctz(elem_mask_of_bit(c));
        if (elem_word_of_bit(c) == 0) o << ind << "b" << a << " = (w" << d << " >> " << sh << "u) & 1u;\n";
        else o << ind << "b" << a << " = (acc.load(" << sc.word_off << "u + e" << d << " * " << (int)sc.wpe << "u + " << elem_word_of_bit(c) << "u) >> " << sh << "u) & 1u;\n";
        break;
      }
      case F_AND: o << ind << "b" << a << " = b" << b << " & b" << c << ";\n"; break;
      case F_OR: o << ind << "b" << a << " = b" << b << " | b" << c << ";\n"; break;
      case F_NOT: o << ind << "b" << a << " = b" << b << " ^ 1u;\n"; break;
      case F_ANDN: o << ind << "b" << a << " = b" << b << " & (b" << c << " ^ 1u);\n"; break;
      case F_CONST: o << ind << "b" << a << " = " << ((b & 1) ? "1u" : "0u") << ";\n"; break;
      case F_MOV: o << ind << "b" << a << " = b" << b << ";\n"; break;
      case F_LOOP: {
        const Scope& sc = plan.scopes[a];
        int d = (int)stack.size();
        o << ind << "b" << c << " = 0u;\n";
        {
          // the conjunction form: one masked compare per element (word), accumulated as a wave mask
          const size_t end = loop_end(pc);
          const uint32_t endins = code[end];
          Conj cj;
          static const bool conj_on = !(getenv("GK_JIT_CONJ") && atoi(getenv("GK_JIT_CONJ")) == 0);   // (A/B aid)
          if (conj_on && (endins & 0xFF) == F_ENDLOOP && ((endins >> 8) & 0xFF) == c && sc.cap <= 16 && conj_body(a, pc, end, (endins >> 16) & 0xFF, &cj)) {
            int pd = -1;
            if (b) { pd = var_of(b - 1); if (pd < 0) throw Unsupported("codegen: parent loop not open"); }
            if (!cj.never) {
              const bool dyn = !(pre || (sc.cap <= 16 && [&] { uint64_t n = sc.cap; for (const Loop& l : stack) n *= plan.scopes[l.scope].cap; return n <= 4; }()));
              o << ind << "{ uint32_t t_ = 0u;\n";
              auto term = [&](const std::string& w0name, uint32_t e_lit, bool have_lit, const std::string& evar) {
                std::string t;
                for (uint32_t k = 0; k < sc.wpe; k++) {
                  if (!cj.care[k]) continue;
                  std::string wk;
                  if (k == 0) wk = w0name;
                  else if (have_lit) wk = "acc.load(" + std::to_string(sc.word_off + e_lit * sc.wpe + k) + "u)";
                  else wk = "acc.load(" + std::to_string(sc.word_off + k) + "u + " + evar + " * " + std::to_string((int)sc.wpe) + "u)";
                  if (!t.empty()) t += " & ";   // (bitwise on purpose: `&&` is control flow -- a divergent branch per element)
                  t += "(uint32_t)((" + wk + " & " + u(cj.care[k]) + ") == " + u(cj.want[k]) + ")";
                }
                if (b) t += " & (uint32_t)((" + w0name + " >> 24) == e" + std::to_string(pd) + ")";
                return t;
              };
              if (!dyn) {
                for (uint32_t e = 0; e < sc.cap; e++) {
                  std::string w0name;
                  const bool in_regs = pre && (sc.cap <= 8u || stack.empty());
                  if (in_regs) { pre_words.insert({a, e}); w0name = "W" + std::to_string(a) + "_" + std::to_string(e); }
                  else w0name = "acc.load(" + std::to_string(sc.word_off + e * sc.wpe) + "u)";
                  o << ind << "  t_ |= " << term(w0name, e, true, "") << ";\n";
                }
              } else {
                o << ind << "  const uint32_t nq_ = GK_UNI(bounds[" << a << "]);\n"
                  << ind << "  for (uint32_t eq_ = 0; eq_ < nq_; eq_++) { const uint32_t wq_ = acc.load(" << sc.word_off << "u + eq_ * " << (int)sc.wpe << "u); t_ |= " << term("wq_", 0, false, "eq_") << "; }\n";
              }
              o << ind << "  b" << c << " = t_; }\n";
            }
            pre_ops += (size_t)sc.cap * 3;
            pc = end + 1;
            break;
          }
        }
        if (pre) {
          // every element a copy of the body; the loop's own F_ENDLOOP closes each copy (below)
          const size_t end = loop_end(pc);
          uint64_t nest0 = sc.cap;
          for (const Loop& l : stack) nest0 *= plan.scopes[l.scope].cap;
          const bool guarded = !(sc.cap <= 4 && nest0 <= 4);   // small nests: every copy runs (absent elements hold zero words)
          if (guarded) pre_bounds.insert(a);
          int pd = -1;
          if (b) { pd = var_of(b - 1); if (pd < 0) throw Unsupported("codegen: parent loop not open"); }
          // (a large scope under another loop is read where it is used -- one LDS read with a constant address per copy -- instead of
          //  being kept in registers across the whole run: 12 words of a volumes array pushed the kernel past its 80-VGPR budget)
          constexpr uint32_t pre_cap = 8u;   // (16: 10 spilled dwords and 0.1167 against 0.1080 ms; 4: 0.1090 -- profiles/r05_variants_g_preload.log)
          const bool in_regs = sc.cap <= pre_cap || stack.empty();
          // ROLLING reads of a scope that is read where it is used: every copy is a basic block of its own (the scalar guard), so a
          // read at the top of the copy is an exposed LDS round trip in front of half a dozen bit operations -- 36 of them in the
          // volumeMounts x volumes join of configs[2].  Two registers carry the words of the next two elements instead: copy e takes
          // its word from one of them and requests element e + 2 into it (copy e + 2 runs only when copy e did: the guards are
          // thresholds of one count).  Not when the body stores derived bits into this scope's words (a later copy must see them).
          static const bool roll_on = !(getenv("GK_JIT_ROLL") && atoi(getenv("GK_JIT_ROLL")) == 0);   // (A/B aid)
          bool roll = roll_on && !in_regs && sc.cap >= 3;
          if (roll) for (size_t q = pc; q < end; q++) {
            const uint32_t qi = code[q], qop = qi & 0xFF;
            if (qop == F_VEQ) { q++; continue; }
            if (qop == F_STE && ((qi >> 16) & 0xFF) == a) { roll = false; break; }
          }
          const auto word_at = [&](uint32_t e) { return std::to_string(sc.word_off + e * sc.wpe) + "u"; };
          if (roll) o << ind << "{ uint32_t P" << d << "a = acc.load(" << word_at(0) << "), P" << d << "b = acc.load(" << word_at(1) << ");\n";
          for (uint32_t e = 0; e < sc.cap; e++) {
            if (in_regs) pre_words.insert({a, e});
            o << ind << (guarded ? "if (" + std::to_string(e) + "u < ns" + std::to_string(a) + ") " : std::string()) << "{\n";
            o << ind << "  constexpr uint32_t e" << d << " = " << e << "u; (void)e" << d << ";\n";
            if (in_regs) o << ind << "  const uint32_t w" << d << " = W" << a << "_" << e << ";\n";
            else if (roll) {
              const char* pn = (e & 1u) ? "b" : "a";
              o << ind << "  const uint32_t w" << d << " = P" << d << pn << ";\n";
              if (e + 2 < sc.cap) o << ind << "  P" << d << pn << " = acc.load(" << word_at(e + 2) << ");\n";
            }
            else o << ind << "  const uint32_t w" << d << " = acc.load(" << (sc.word_off + e * sc.wpe) << "u);\n";
            o << ind << "  uint32_t v" << d << " = w" << d << " & 1u;\n";
            if (b) o << ind << "  v" << d << " = v" << d << " & (uint32_t)((w" << d << " >> 24) == e" << pd << ");\n";
            stack.push_back({a, d, (int)e});
            gen(pc, end + 1, staged, ind + "  ");   // (its F_ENDLOOP pops the stack and closes the copy)
          }
          if (roll) o << ind << "}\n";
          pc = end + 1;
          break;
        }
        // small capacities: constant trip count, fully unrolled -- the element words of absent elements are zero, so
        // they contribute nothing, and the compiler can issue all LDS reads of the nest at once
        uint64_t nest = sc.cap;
        for (const Loop& l : stack) nest *= plan.scopes[l.scope].cap;
        constexpr uint64_t unroll_max = 4;
        if (sc.cap <= 16 && nest <= unroll_max) {
          o << ind << "{ _Pragma(\"unroll\")\n";
          o << ind << "  for (uint32_t e" << d << " = 0; e" << d << " < " << sc.cap << "u; e" << d << "++) {\n";
        } else {
          // run-time trip count (the wave's largest element count): partially unrolled, so that the LDS reads of several
          // iterations are in flight together instead of one exposed LDS latency per element
          constexpr int dyn_unroll = 1;   // (partial unrolling of the run-time-bounded loops measured slower in round 3: 0.127 / 0.140 against 0.122 ms)
          o << ind << "{ const uint32_t n" << d << " = GK_UNI(bounds[" << a << "]);\n";
          if (dyn_unroll > 1) o << ind << "  _Pragma(\"unroll " << dyn_unroll << "\")\n";
          o << ind << "  for (uint32_t e" << d << " = 0; e" << d << " < n" << d << "; e" << d << "++) {\n";
        }
        o << ind << "    const uint32_t w" << d << " = acc.load(" << sc.word_off << "u + e" << d << " * " << (int)sc.wpe << "u);\n";
        o << ind << "    uint32_t v" << d << " = w" << d << " & 1u;\n";
        if (b) {
          int pd = var_of(b - 1);
          if (pd < 0) throw Unsupported("codegen: parent loop not open");
          o << ind << "    v" << d << " = v" << d << " & (uint32_t)((w" << d << " >> 24) == e" << pd << ");\n";
        }
        stack.push_back({a, d, -1});
        ind += "    ";
        break;
      }
      case F_ENDLOOP: {
        int d = stack.back().depth;
        o << ind << "b" << a << " = b" << a << " | (b" << b << " & v" << d << ");\n";
        if (pre) { stack.pop_back(); o << ind.substr(0, ind.size() - 2) << "}\n"; return; }
        stack.pop_back();
        ind = ind.substr(0, ind.size() - 4);
        o << ind << "  }\n" << ind << "}\n";
        break;
      }
      case F_ENDLOOP2: {   // counting loop: a = once, b = body, c = twice
        int d = stack.back().depth;
        o << ind << "b" << c << " = b" << c << " | (b" << a << " & b" << b << " & v" << d << ");\n";
        o << ind << "b" << a << " = b" << a << " | (b" << b << " & v" << d << ");\n";
        if (pre) { stack.pop_back(); o << ind.substr(0, ind.size() - 2) << "}\n"; return; }
        stack.pop_back();
        ind = ind.substr(0, ind.size() - 4);
        o << ind << "  }\n" << ind << "}\n";
        break;
      }
      case F_VEQ: {
        uint32_t x = code[pc++];
        uint32_t sa = x & 0xFF, la = (x >> 8) & 0xFF, sb = (x >> 16) & 0xFF, lb = x >> 24;
        const Scope& A = plan.scopes[sa];
        const Scope& B = plan.scopes[sb];
        int da = var_of(sa), db = var_of(sb);
        if (da < 0 || db < 0) throw Unsupported("codegen: join outside its loops");
        auto vid = [&](const Scope& S, int d, uint32_t slot) {
          std::ostringstream x;
          if (scope_packed(S)) x << "((w" << d << " >> " << ELEM_VID_SHIFT << "u) & " << GK_VID_OVERFLOW << "u)";   // word0 of the loop's current element is in a register
          else if (pre) {
            int lit = -1;
            for (const Loop& l : stack) if (l.depth == d) lit = l.lit;
            const std::string name = "X" + std::to_string(&S - &plan.scopes[0]) + "_" + std::to_string(lit) + "_" + std::to_string(slot);
            pre_vals[name] = "acc.load(" + std::to_string(S.val_off + (uint32_t)lit * val_stride(S.nvals) + slot) + "u)";
            x << name;
          }
          else x << "acc.load(" << S.val_off << "u + e" << d << " * " << val_stride(S.nvals) << "u + " << slot << "u)";
          return x.str();
        };
        o << ind << "b" << a << " = (uint32_t)vid_eq(" << vid(A, da, la) << ", " << vid(B, db, lb) << ");\n";
        break;
      }
      case F_STE: {
        const Scope& sc = plan.scopes[b];
        int d = var_of(b);
        if (d < 0) throw Unsupported("codegen: element store outside its loop");
        if (pre && elem_word_of_bit(c) == 0 && pre_words.count({b, (uint32_t)[&] { int lit = -1; for (const Loop& l : stack) if (l.depth == d) lit = l.lit; return lit; }()})) {
          int lit = -1;
          for (const Loop& l : stack) if (l.depth == d) lit = l.lit;
          o << ind << "if (GK_BIT(b" << a << ")) { acc.or_word(" << sc.word_off << "u + e" << d << " * " << (int)sc.wpe << "u, " << u(elem_mask_of_bit(c)) << "); W" << b << "_" << lit << " |= " << u(elem_mask_of_bit(c)) << "; }\n";
          break;
        }
        o << ind << "if (GK_BIT(b" << a << ")) acc.or_word(" << sc.word_off << "u + e" << d << " * " << (int)sc.wpe << "u + " << elem_word_of_bit(c) << "u, " << u(elem_mask_of_bit(c)) << ");\n";
        break;
      }
      case F_STG: {
        uint32_t bit = b | (c << 8);
        if (staged) o << ind << "if (GK_BIT(b" << a << ")) { g" << (bit >> 5) << " |= " << u(1u << (bit & 31)) << "; acc.or_word(" << (bit >> 5) << "u, " << u(1u << (bit & 31)) << "); }\n";
        else o << ind << "if (GK_BIT(b" << a << ")) g" << (bit >> 5) << " |= " << u(1u << (bit & 31)) << ";\n";
        break;
      }
      case F_RES: {
        const char* f = b == 0 ? "viol" : b == 1 ? "match" : "err";
        // staged parts hand the result of slot c to GK_RES: on the device one ballot turns the 64 lanes' answers into the
        // slot's bitmap word (kernel_body.inc), elsewhere it accumulates into `res` like the monolithic function
        if (staged) {
          const uint32_t kind = (b == 0 && c >= 64) ? 2u + (c >> 6) : b, lane_ = (b == 0) ? (c & 63u) : c;
          if (kind >= 2u + GK_VIOL_WORDS || lane_ >= 64u) throw Unsupported("codegen: result slot out of range");
          o << ind << "GK_RES(" << kind << ", " << lane_ << ", b" << a << ");\n";
          res_slots[kind] |= 1ull << lane_;
        } else if (b == 0) o << ind << "res.viol[" << (c >> 6) << "] |= (uint64_t)b" << a << " << " << (c & 63u) << ";\n";
        else o << ind << "res." << f << " |= (uint64_t)b" << a << " << " << c << ";\n";
        break;
      }
      case F_END: pc = pc1; break;
      default: throw Unsupported("codegen: unknown formula op");
    }
  }
  };
  gen(0, code.size(), false, "  ");
  o << "  return res;\n}\n\n";
  // ---- the same formulas cut into self-contained blocks and spread over the tile's waves: blocks of one STAGE are
  // independent (they only read bits written by earlier stages); part = stage * parts + wave-within-half
  {
    const uint32_t NW = parts;   // waves that share the formulas of one 64-review half
    struct Blk { size_t pc0, pc1; uint32_t stage; uint64_t cost; std::vector<uint64_t> writes, reads; };
    std::vector<Blk> blks;
    size_t prev = 0;
    for (uint32_t e : plan.seg_ends) { blks.push_back({prev, e, 0, 0, {}, {}}); prev = e; }
    constexpr uint64_t loop_weight = 3;   // cost of a loop body relative to straight-line code
    std::map<uint64_t, size_t> writer;   // derived bit -> block
    for (size_t bi = 0; bi < blks.size(); bi++) {
      Blk& B = blks[bi];
      uint64_t weight = 1;
      for (size_t pc = B.pc0; pc < B.pc1;) {
        uint32_t ins = code[pc++];
        uint32_t op = ins & 0xFF, b = (ins >> 16) & 0xFF, c = ins >> 24;
        if (op == F_VEQ) { pc++; B.cost += 12 * weight; }
        else if (op == F_LOOP) { B.cost += 4 * weight; weight *= loop_weight; }
        else if (op == F_ENDLOOP || op == F_ENDLOOP2) { weight /= loop_weight; B.cost += (op == F_ENDLOOP2 ? 2 : 1) * weight; }
        else B.cost += weight;
        if (op == F_STG) B.writes.push_back(1ull << 40 | b | (c << 8));
        if (op == F_STE) B.writes.push_back(2ull << 40 | (uint64_t)b << 16 | c);
        if (op == F_LDG) B.reads.push_back(1ull << 40 | b | (c << 8));
        if (op == F_LDE) B.reads.push_back(2ull << 40 | (uint64_t)b << 16 | c);
      }
      for (uint64_t r : B.reads) { auto it = writer.find(r); if (it != writer.end() && it->second != bi) B.stage = std::max(B.stage, blks[it->second].stage + 1); }
      for (uint64_t w : B.writes) writer[w] = bi;
    }
    uint32_t n_stages = 0;
    for (auto& B : blks) n_stages = std::max(n_stages, B.stage + 1);
    std::vector<std::vector<size_t>> parts;
    // CHAINS (round 3).  A block of a later stage only waits for the blocks that write the derived bits it reads.  When those
    // run on the SAME wave, program order is all it needs (lane = review in every block: a wave reads back what its own lanes
    // OR-ed into LDS): such a block is appended to its producers' share of stage 0.  Producers that sit in another share
    // are DUPLICATED into this one when they are cheap (derived bits are ORs: writing one twice is harmless).  If every later
    // block can be placed that way the formulas take ONE stage -- one barrier and one call per item instead of one per level
    // of derived bits (configs[2]: three stages, the last two a dozen lines each, profiles/r03_*).  Otherwise: stages as before.
    bool chained = false;
    constexpr bool chain_on = true;
    if (chain_on && n_stages > 1) {
      std::vector<std::vector<size_t>> deps(blks.size());   // direct producers
      {
        std::map<uint64_t, size_t> w2;
        for (size_t bi = 0; bi < blks.size(); bi++) {
          for (uint64_t r : blks[bi].reads) { auto it = w2.find(r); if (it != w2.end() && it->second != bi) deps[bi].push_back(it->second); }
          for (uint64_t w : blks[bi].writes) w2[w] = bi;
        }
      }
      std::vector<std::vector<size_t>> closure(blks.size());   // transitive producers, ascending
      for (size_t bi = 0; bi < blks.size(); bi++) {
        std::vector<size_t> c;
        for (size_t d : deps[bi]) { c.push_back(d); c.insert(c.end(), closure[d].begin(), closure[d].end()); }
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        closure[bi] = c;
      }
      std::vector<std::vector<bool>> in_part(NW, std::vector<bool>(blks.size(), false));
      std::vector<uint64_t> load(NW, 0);
      uint64_t total = 0, dup_total = 0;
      for (auto& B : blks) total += B.cost;
      // (a) FAMILIES: blocks connected by derived bits go to one share as a whole (no duplicates), heaviest family to the
      //     lightest share -- as long as no family outweighs a fair share by much
      {
        std::vector<size_t> root(blks.size());
        for (size_t i = 0; i < blks.size(); i++) root[i] = i;
        auto find = [&](size_t x) { while (root[x] != x) x = root[x] = root[root[x]]; return x; };
        for (size_t bi = 0; bi < blks.size(); bi++) for (size_t d : deps[bi]) root[find(bi)] = find(d);
        std::map<size_t, uint64_t> fam_cost;
        for (size_t bi = 0; bi < blks.size(); bi++) fam_cost[find(bi)] += blks[bi].cost;
        uint64_t biggest = 0;
        for (auto& kv : fam_cost) biggest = std::max(biggest, kv.second);
        if (biggest * NW <= total + total / 8) {
          std::vector<size_t> fams;
          for (auto& kv : fam_cost) fams.push_back(kv.first);
          std::stable_sort(fams.begin(), fams.end(), [&](size_t x, size_t y) { return fam_cost[x] > fam_cost[y]; });
          for (size_t f : fams) {
            uint32_t w = 0;
            for (uint32_t k = 1; k < NW; k++) if (load[k] < load[w]) w = k;
            load[w] += fam_cost[f];
            for (size_t bi = 0; bi < blks.size(); bi++) if (find(bi) == f) in_part[w][bi] = true;
          }
          chained = true;
        }
      }
      // (b) a family too heavy for one share: stage-0 blocks dealt as the staged form deals them; a later block joins the share
      //     that holds most of its producers, the missing ones are duplicated if that is cheap
      if (!chained) {
        std::vector<size_t> ids;
        for (size_t bi = 0; bi < blks.size(); bi++) if (blks[bi].stage == 0) ids.push_back(bi);
        std::stable_sort(ids.begin(), ids.end(), [&](size_t x, size_t y) { return blks[x].cost > blks[y].cost; });
        for (size_t bi : ids) {
          uint32_t w = 0;
          for (uint32_t k = 1; k < NW; k++) if (load[k] < load[w]) w = k;
          load[w] += blks[bi].cost;
          in_part[w][bi] = true;
        }
        const uint64_t dup_max = std::max<uint64_t>(64, total / NW / 8);   // duplicated cost allowed per block
        chained = true;
        for (size_t bi = 0; bi < blks.size() && chained; bi++) {
          if (blks[bi].stage == 0) continue;
          uint32_t best = NW;
          uint64_t best_extra = 0;
          for (uint32_t k = 0; k < NW; k++) {
            uint64_t extra = 0;
            for (size_t d : closure[bi]) if (!in_part[k][d]) extra += blks[d].cost;
            if (extra > dup_max) continue;
            if (best == NW || extra + load[k] < best_extra + load[best]) { best = k; best_extra = extra; }
          }
          if (best == NW) { chained = false; break; }
          for (size_t d : closure[bi]) in_part[best][d] = true;
          in_part[best][bi] = true;
          load[best] += best_extra + blks[bi].cost;
          dup_total += best_extra;
        }
        if (chained && dup_total > total / 4) chained = false;   // (duplicates are work done twice)
      }
      if (chained) {
        n_stages = 1;
        parts.assign(NW, {});
        for (uint32_t k = 0; k < NW; k++) for (size_t bi = 0; bi < blks.size(); bi++) if (in_part[k][bi]) parts[k].push_back(bi);
      }
    }
    if (!chained) {
    parts.assign((size_t)n_stages * NW, {});
    for (uint32_t st = 0; st < n_stages; st++) {   // greedy balance: heaviest block to the lightest wave
      std::vector<size_t> ids;
      for (size_t bi = 0; bi < blks.size(); bi++) if (blks[bi].stage == st) ids.push_back(bi);
      std::stable_sort(ids.begin(), ids.end(), [&](size_t x, size_t y) { return blks[x].cost > blks[y].cost; });
      std::vector<uint64_t> load(NW, 0);
      for (size_t bi : ids) {
        uint32_t w = 0;
        for (uint32_t k = 1; k < NW; k++) if (load[k] < load[w]) w = k;
        load[w] += blks[bi].cost;
        parts[(size_t)st * NW + w].push_back(bi);
      }
    }
    }
    // the preloaded form when its unrolled text stays small: operations after unrolling, summed over the parts
    bool use_pre = !(getenv("GK_JIT_PRELOAD") && atoi(getenv("GK_JIT_PRELOAD")) == 0);
    if (use_pre) {
      constexpr size_t pre_budget = 12000;   // (100 000 -- every part of the corpus plans unrolled -- measured slower: 0.6045 against 0.5141 ms summed over the groups)
      std::function<uint64_t(size_t, size_t)> unrolled = [&](size_t pc, size_t pc1) -> uint64_t {
        uint64_t n = 0;
        while (pc < pc1) {
          const uint32_t ins = code[pc], op = ins & 0xFF;
          if (op == F_VEQ) { pc += 2; n += 2; continue; }
          if (op == F_LOOP) { const size_t end = loop_end(pc + 1); n += (uint64_t)plan.scopes[(ins >> 8) & 0xFF].cap * (4 + unrolled(pc + 1, end)); pc = end + 1; continue; }
          n++; pc++;
        }
        return n;
      };
      uint64_t total = 0;
      for (auto& part : parts) for (size_t bi : part) total += unrolled(blks[bi].pc0, blks[bi].pc1);
      for (const Scope& sc : plan.scopes) if (sc.cap > 16) total = ~0ull;   // (large capacities keep their loops)
      if (total > pre_budget) use_pre = false;
    }
    o << "#define GK_HAS_STAGES 1\nconstexpr uint32_t GK_N_STAGES = " << n_stages << "u;\nconstexpr uint32_t GK_GEN_PARTS = " << NW << "u;\n"
      << "#ifndef GK_RES\n#define GK_RES(kind, slot, b) do { if ((kind) == 0) res.viol[0] |= (uint64_t)(b) << (slot); else if ((kind) == 1) res.match |= (uint64_t)(b) << (slot); "
         "else if ((kind) == 2) res.err |= (uint64_t)(b) << (slot); else res.viol[(kind) - 2] |= (uint64_t)(b) << (slot); } while (0)\n#define GK_RES_PROLOGUE\n#endif\n"
         "#ifndef GK_RES_FLUSH\n#define GK_RES_FLUSH(m0, m1, m2, m3, m4, m5)\n#endif\n"
      << "template <class Acc>\nGK_HD void jit_formula_part(uint32_t part, Acc& acc, uint32_t flags, const uint8_t* heap, const uint32_t* bounds, Results& res, unsigned long long* masks) {\n"
      << "  (void)heap; (void)flags; (void)bounds; (void)res; (void)masks;\n  GK_RES_PROLOGUE\n  uint32_t";
    for (int i = 0; i < 64; i++) o << (i ? ", " : " ") << "b" << i << " = 0u";
    o << ";\n";
    for (uint32_t w = 0; w < plan.dims.n_gwords; w++) o << "  uint32_t g" << w << " = acc.load(" << w << "u);\n";
    o << "  switch (part) {\n";
    for (size_t p = 0; p < parts.size(); p++) {
      o << "    case " << p << ": {\n";
      for (auto& rs : res_slots) rs = 0;
      std::vector<size_t> order = parts[p];
      std::sort(order.begin(), order.end());
      if (use_pre) {
        // RUNS of consecutive blocks share one set of preloaded registers; a run ends where the words it keeps live would exceed
        // `pre_live` (the kernel runs at an 80-VGPR budget: everything preloaded at the top of the part spilled 19 dwords).  A later
        // run re-reads what an earlier one derived: the same wave's LDS operations complete in order.
        // (4 since round 6: at the 64-VGPR budget of four row groups per CU, and with the formulas running below the other phases'
        //  priority, short runs win -- 10 M objects 0.432 -> 0.417 ms, configs[2] 0.0470 -> 0.0462, the corpus level; 16 before:
        //  profiles/r06_variants_ae_*.log)
        static const size_t pre_live = getenv("GK_JIT_PRE_LIVE") ? (size_t)std::max(1, atoi(getenv("GK_JIT_PRE_LIVE"))) : 4;   // (tuning aid, read once)
        std::set<uint32_t> bounds_done;
        std::ostringstream run_body;
        std::set<std::pair<uint32_t, uint32_t>> run_words;
        std::map<std::string, std::string> run_vals;
        auto flush = [&]() {
          if (run_body.str().empty()) return;
          o << "      {\n";
          for (auto& we : run_words) {
            const Scope& sc = plan.scopes[we.first];
            o << "      uint32_t W" << we.first << "_" << we.second << " = acc.load(" << (sc.word_off + we.second * sc.wpe) << "u);\n";
          }
          for (auto& kv : run_vals) o << "      const uint32_t " << kv.first << " = " << kv.second << ";\n";
          o << run_body.str() << "      }\n";
          run_body.str(""); run_body.clear(); run_words.clear(); run_vals.clear();
        };
        for (size_t bi : order) {
          std::ostringstream body;
          out_ = &body; pre = true;
          pre_words.clear(); pre_bounds.clear(); pre_vals.clear();
          stack.clear();
          gen(blks[bi].pc0, blks[bi].pc1, true, "      ");
          out_ = &o; pre = false;
          for (uint32_t sidx : pre_bounds) if (bounds_done.insert(sidx).second) { flush(); o << "      const uint32_t ns" << sidx << " = GK_UNI(bounds[" << sidx << "]);\n"; }
          std::set<std::pair<uint32_t, uint32_t>> uw = run_words;
          uw.insert(pre_words.begin(), pre_words.end());
          std::map<std::string, std::string> uv = run_vals;
          uv.insert(pre_vals.begin(), pre_vals.end());
          if (uw.size() + uv.size() > pre_live && !run_body.str().empty()) { flush(); uw = pre_words; uv = pre_vals; }
          run_words.swap(uw); run_vals.swap(uv);
          run_body << body.str();
        }
        flush();
      } else
      for (size_t bi : order) { stack.clear(); gen(blks[bi].pc0, blks[bi].pc1, true, "      "); }
      // the part's finished slots leave the wave together (jit_source.hpp jit_res_macros: lane s holds slot s's word)
      {
        static_assert(GK_VIOL_WORDS == 4, "GK_RES_FLUSH takes the masks of six kinds");
        char fb[256];
        snprintf(fb, sizeof fb, "      GK_RES_FLUSH(0x%llxull, 0x%llxull, 0x%llxull, 0x%llxull, 0x%llxull, 0x%llxull);\n", (unsigned long long)res_slots[0], (unsigned long long)res_slots[1],
                 (unsigned long long)res_slots[2], (unsigned long long)res_slots[3], (unsigned long long)res_slots[4], (unsigned long long)res_slots[5]);
        o << fb;
      }
      o << "    } break;\n";
    }
    o << "    default: break;\n  }\n";
    for (uint32_t w = 0; w < plan.dims.n_gwords; w++) o << "  (void)g" << w << ";\n";
    o << "}\n";
  }
  o << "}  // namespace gk\n";
  return o.str();
}

}  // namespace gk
