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
inline ChunkLists build_chunk_lists(const uint32_t* tile_idx, uint32_t n_groups, uint32_t n_slots, std::vector<BoundPath> bound, uint32_t list_cap,
                                    uint32_t n_waves) {
  ChunkLists out;
  // (ties by entry: the chunks of one predicate class are contiguous)
  std::stable_sort(bound.begin(), bound.end(), [](const BoundPath& a, const BoundPath& b) {
    return a.cost != b.cost ? a.cost > b.cost : (a.ent & GK_DESC_ENT_MASK) < (b.ent & GK_DESC_ENT_MASK);
  });
  std::vector<uint32_t> count(n_groups, 0);
  uint32_t longest = 0;
  for (uint32_t g = 0; g < n_groups; g++) {
    const uint32_t* ix = tile_idx + (size_t)g * (n_slots + 1u);
    uint32_t n = 0;
    for (const BoundPath& b : bound) n += (ix[b.slot + 1u] - ix[b.slot] + (uint32_t)GK_TILE - 1u) / (uint32_t)GK_TILE;
    count[g] = n;
    if (n < list_cap) longest = std::max(longest, n);
  }
  out.capg = std::min(list_cap, longest + 1u);
  out.d.assign((size_t)n_groups * out.capg, ChunkDesc{0u, 0u});
  std::vector<ChunkDesc> tmp;
  for (uint32_t g = 0; g < n_groups; g++) {
    ChunkDesc* L = &out.d[(size_t)g * out.capg];
    if (count[g] >= out.capg) { L[0] = ChunkDesc{0u, GK_LIST_OVERFLOW}; out.n_overflow_groups++; continue; }
    const uint32_t* ix = tile_idx + (size_t)g * (n_slots + 1u);
    tmp.clear();
    for (const BoundPath& b : bound) {
      const uint32_t lo = ix[b.slot], hi = ix[b.slot + 1u];
      const uint32_t ent = (b.ent & GK_DESC_ENT_MASK) | ((b.ent & GK_ENT_NEEDS_STR) ? GK_DESC_NEEDS_STR : 0u);
      for (uint32_t st = lo; st < hi; st += (uint32_t)GK_TILE) tmp.push_back(ChunkDesc{st, (std::min(hi - st, (uint32_t)GK_TILE) - 1u) | (ent << GK_DESC_ENT_SHIFT)});
    }
    const uint32_t n = (uint32_t)tmp.size();
    L[0] = ChunkDesc{n, 0u};
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t r = i / n_waves, k = i % n_waves;
      const bool full = (r + 1u) * n_waves <= n;
      L[1u + ((r & 1u) && full ? r * n_waves + (n_waves - 1u - k) : i)] = tmp[i];
    }
    out.n_chunks += n;
  }
  return out;
}

}  // namespace gk
