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
// runs = true (plan-specialised kernel): every wave gets RUNS of chunks of one predicate class (up to GK_RUN_MAX in a row),
// so that its class dispatch -- a compare tree on the way in, a chain of join blocks on the way out, ~40 scalar
// instructions -- is paid once per run instead of once per chunk (kernel_body.inc GK_RUNS_K: the case body loops while the
// wave's next chunk has the same class).  Runs are dealt longest-processing-time-first by (fixed cost per chunk + class
// cost); wave w's k-th chunk sits at list position w + k * n_waves, a wave with fewer chunks than the longest is padded with
// NULL entries (info = GK_DESC_NULL: no load, no dispatch).
constexpr uint32_t GK_RUN_MAX = 4;
inline ChunkLists build_chunk_lists(const uint32_t* tile_idx, uint32_t n_groups, uint32_t n_slots, std::vector<BoundPath> bound, uint32_t list_cap,
                                    uint32_t n_waves, bool runs = false) {
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
    cost_of.clear();
    for (const BoundPath& b : bound) {
      const uint32_t lo = ix[b.slot], hi = ix[b.slot + 1u];
      const uint32_t ent = (b.ent & GK_DESC_ENT_MASK) | ((b.ent & GK_ENT_NEEDS_STR) ? GK_DESC_NEEDS_STR : 0u);
      for (uint32_t st = lo; st < hi; st += (uint32_t)GK_TILE) {
        tmp.push_back(ChunkDesc{st, (std::min(hi - st, (uint32_t)GK_TILE) - 1u) | (ent << GK_DESC_ENT_SHIFT)});
        cost_of.push_back(b.cost);
      }
    }
    const uint32_t n = (uint32_t)tmp.size();
    out.n_chunks += n;
    std::vector<ChunkDesc>& L = lists[g];
    if (n + 1u > list_cap) { L.clear(); continue; }   // (overflow: decided again below, against capg)
    if (!runs) {
      L.resize(n);
      for (uint32_t i = 0; i < n; i++) {
        const uint32_t r = i / n_waves, k = i % n_waves;
        const bool full = (r + 1u) * n_waves <= n;
        L[(r & 1u) && full ? r * n_waves + (n_waves - 1u - k) : i] = tmp[i];
      }
    } else {
      // runs of one class, at most GK_RUN_MAX chunks each (tmp is class-contiguous, heaviest class first), dealt LPT
      struct Run { uint32_t first, n; uint64_t load; };
      std::vector<Run> rs;
      for (uint32_t i = 0; i < n;) {
        const uint32_t cls = (tmp[i].info >> GK_DESC_ENT_SHIFT) & GK_DESC_ENT_MASK;
        uint32_t k = 1;
        while (i + k < n && k < GK_RUN_MAX && ((tmp[i + k].info >> GK_DESC_ENT_SHIFT) & GK_DESC_ENT_MASK) == cls) k++;
        uint64_t load = 0;
        for (uint32_t q = 0; q < k; q++) load += 8u + cost_of[i + q];
        rs.push_back(Run{i, k, load});
        i += k;
      }
      std::stable_sort(rs.begin(), rs.end(), [](const Run& a, const Run& b) { return a.load > b.load; });
      std::vector<std::vector<uint32_t>> per(n_waves);
      std::vector<uint64_t> load(n_waves, 0);
      for (const Run& r : rs) {
        uint32_t w = 0;
        for (uint32_t q = 1; q < n_waves; q++) if (load[q] < load[w] || (load[q] == load[w] && per[q].size() < per[w].size())) w = q;
        for (uint32_t q = 0; q < r.n; q++) per[w].push_back(r.first + q);
        load[w] += r.load;
      }
      uint32_t deepest = 0;
      for (auto& v : per) deepest = std::max<uint32_t>(deepest, (uint32_t)v.size());
      L.assign((size_t)deepest * n_waves, ChunkDesc{0u, GK_DESC_NULL});
      for (uint32_t w = 0; w < n_waves; w++) for (uint32_t k = 0; k < per[w].size(); k++) L[w + k * n_waves] = tmp[per[w][k]];
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
