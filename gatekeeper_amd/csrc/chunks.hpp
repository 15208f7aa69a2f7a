// Binding a plan to a table (host side, once per plan x table): the per-group CHUNK LISTS the dominant kernel walks.
// The table's slot index says where the rows of key path s start in row group g; the plan says which paths carry
// predicates.  Instead of letting every workgroup look its ~100 segments up again in every sweep (two dependent loads,
// an LDS atomic per path and two barriers before the first row is requested -- 13 % of a group's time, r02 clock
// profile), the segments are cut into 64-row chunks here and stored as one coalesced list per group.  Order: heaviest
// predicate class first, dealt to the group's waves round robin with every other round reversed (longest-processing-
// time-first: the waves of a group end phase 1 together instead of waiting for the one that drew the string classes).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "plan.hpp"

namespace gk {

// relative cost of a predicate list (only the ORDER matters): string predicates and element destinations (LDS atomics on
// shared words) weigh more.
template <class PredT>
inline uint32_t pred_list_cost(const PredT* preds, uint32_t n, bool (*needs_str)(const PredT&)) {
  uint32_t c = 1;
  for (uint32_t j = 0; j < n; j++) c += 1u + (needs_str(preds[j]) ? 3u : 0u) + (preds[j].dst == D_ELEM ? 1u : 0u);
  return c;
}

struct BoundPath { uint32_t slot, ent, cost; };   // ent: what the kernel's row function takes (path-table entry or class id), flag in GK_ENT_NEEDS_STR

struct ChunkLists {
  std::vector<ChunkDesc> d;   // [n_groups][capg]
  uint32_t capg = 1;          // entries per group: header + the longest list (<= list_cap)
  uint64_t n_chunks = 0;
  uint32_t n_overflow_groups = 0;
};

// tile_idx: [n_groups][n_slots + 1]; list_cap: entries a group's list may hold in LDS incl. the header; n_waves: waves
// that share a group's list
// The chunks of a group are ordered heaviest predicate class first and dealt to the group's waves round robin, every other round
// reversed.  (RUNS of one class per wave, dealt longest-processing-time-first with in-case loops in the row code, measured slower in
// round 3 -- configs[2] 0.144 against 0.122 ms, the corpus 1.89 against 0.80, profiles/r03_variants_g_class_runs.log -- and were
// removed in round 5.)
inline ChunkLists build_chunk_lists(const uint32_t* tile_idx, uint32_t n_groups, uint32_t n_slots, std::vector<BoundPath> bound, uint32_t list_cap,
                                    uint32_t n_waves) {
  ChunkLists out;
  // (ties by entry: the chunks of one predicate class are contiguous)
  std::stable_sort(bound.begin(), bound.end(), [](const BoundPath& a, const BoundPath& b) {
    return a.cost != b.cost ? a.cost > b.cost : (a.ent & GK_DESC_ENT_MASK) < (b.ent & GK_DESC_ENT_MASK);
  });
  // pass 1: every group's list (header excluded), in the order its waves take the entries
  std::vector<std::vector<ChunkDesc>> lists(n_groups);
  std::vector<ChunkDesc> tmp;
  std::vector<uint32_t> cost_of;
  uint32_t longest = 0;
  for (uint32_t g = 0; g < n_groups; g++) {
    const uint32_t* ix = tile_idx + (size_t)g * (n_slots + 1u);
    tmp.clear();
    for (const BoundPath& b : bound) {
      const uint32_t lo = ix[b.slot], hi = ix[b.slot + 1u];
      const uint32_t ent = (b.ent & GK_DESC_ENT_MASK) | ((b.ent & GK_ENT_NEEDS_STR) ? GK_DESC_NEEDS_STR : 0u);
      for (uint32_t st = lo; st < hi; st += (uint32_t)GK_TILE) {
        tmp.push_back(ChunkDesc{st, (std::min(hi - st, (uint32_t)GK_TILE) - 1u) | (ent << GK_DESC_ENT_SHIFT)});
      }
    }
    const uint32_t n = (uint32_t)tmp.size();
    out.n_chunks += n;
    std::vector<ChunkDesc>& L = lists[g];
    if (n + 1u > list_cap) { L.clear(); continue; }   // (overflow: decided again below, against capg)
    L.resize(n);
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t r = i / n_waves, k = i % n_waves;
      const bool full = (r + 1u) * n_waves <= n;
      L[(r & 1u) && full ? r * n_waves + (n_waves - 1u - k) : i] = tmp[i];
    }
    if (L.size() + 1u <= list_cap) longest = std::max<uint32_t>(longest, (uint32_t)L.size());
  }
  // pass 2: one block of capg entries per group: header + list (+ unused tail)
  out.capg = std::min(list_cap, longest + 1u);
  out.d.assign((size_t)n_groups * out.capg, ChunkDesc{0u, 0u});
  for (uint32_t g = 0; g < n_groups; g++) {
    ChunkDesc* D = &out.d[(size_t)g * out.capg];
    const std::vector<ChunkDesc>& L = lists[g];
    const uint32_t* ix = tile_idx + (size_t)g * (n_slots + 1u);
    bool any = false;
    for (const BoundPath& b : bound) any = any || ix[b.slot + 1u] > ix[b.slot];
    if ((L.empty() && any) || L.size() + 1u > out.capg) { D[0] = ChunkDesc{0u, GK_LIST_OVERFLOW}; out.n_overflow_groups++; continue; }
    D[0] = ChunkDesc{(uint32_t)L.size(), 0u};
    for (size_t i = 0; i < L.size(); i++) D[1u + i] = L[i];
  }
  return out;
}

}  // namespace gk
