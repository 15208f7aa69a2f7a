// See flatten.hpp.
#include "flatten.hpp"
#include <immintrin.h>

#include <sched.h>
#include <cstdio>
#include <thread>
#include <sys/mman.h>
#include <mutex>

#include <algorithm>
#include <cstring>
#include <mutex>

namespace gk {

// ------------------------------------------------------------------------------------------------ PathDict
size_t PathDict::KeyHash::operator()(const std::pair<uint32_t, std::string>& k) const {
  return std::hash<std::string>()(k.second) * 1000003u ^ (size_t)k.first * 0x9E3779B97F4A7C15ull;
}

PathDict::PathDict() { infos_.push_back({kNone, "", false, 0}); }

uint32_t PathDict::intern(uint32_t parent, const std::string& key, bool is_elem) {
  // the "[]" child is keyed by a flag beside the parent id (ids stay below 2^31), never by a reserved member name: an
  // object member may be called anything
  std::pair<uint32_t, std::string> k(is_elem ? (parent | 0x80000000u) : parent, is_elem ? std::string() : key);
  {
    std::shared_lock<std::shared_mutex> rl(mu_);
    auto it = map_.find(k);
    if (it != map_.end()) return it->second;
  }
  std::unique_lock<std::shared_mutex> wl(mu_);
  auto it = map_.find(k);
  if (it != map_.end()) return it->second;
  uint32_t id = (uint32_t)infos_.size();
  uint8_t ad = infos_[parent].adepth + (is_elem ? 1 : 0);
  infos_.push_back({parent, is_elem ? std::string() : key, is_elem, ad});
  map_.emplace(std::move(k), id);
  return id;
}
uint32_t PathDict::child(uint32_t parent, const std::string& key) { return intern(parent, key, false); }
uint32_t PathDict::elem(uint32_t parent) { return intern(parent, "", true); }
uint32_t PathDict::find_child(uint32_t parent, const std::string& key) const {
  std::shared_lock<std::shared_mutex> rl(mu_);
  auto it = map_.find(std::make_pair(parent, key));
  return it == map_.end() ? kNone : it->second;
}
PathDict::Info PathDict::info(uint32_t id) const {
  std::shared_lock<std::shared_mutex> rl(mu_);
  return infos_[id];
}
uint32_t PathDict::size() const {
  std::shared_lock<std::shared_mutex> rl(mu_);
  return (uint32_t)infos_.size();
}
std::string PathDict::to_string(uint32_t id) const {
  std::shared_lock<std::shared_mutex> rl(mu_);
  std::vector<std::string> parts;
  while (id != 0 && id != kNone) {
    const Info& in = infos_[id];
    parts.push_back(in.is_elem ? "[]" : in.key);
    id = in.parent;
  }
  std::string o = "review";
  for (auto it = parts.rbegin(); it != parts.rend(); ++it) { if (*it != "[]") o += "."; o += *it; }
  return o;
}

uint32_t hash32(const uint8_t* p, size_t n) {
  // only ever compared for equality as a fast reject (bytes decide): eight bytes per multiply (a byte-at-a-time FNV was 4 cycles
  // per byte of every kept string), a murmur-style finalizer down to 32 bits
  uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)n * 0xD6E8FEB86659FD93ull);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0x9FB21C651E98DF25ull; h ^= h >> 32; }
  if (i < n) { uint64_t w = 0; memcpy(&w, p + i, n - i); h = (h ^ w) * 0x9FB21C651E98DF25ull; h ^= h >> 32; }
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  return (uint32_t)h;
}

// ------------------------------------------------------------------------------------------------ patterns / registry
std::string pattern_to_string(const Pattern& p) {
  std::string o = "review";
  for (const PatStep& s : p) {
    if (!s.any) { o += "." + s.key; continue; }
    o += s.elems_only ? "[]" : "[*";
    for (auto& k : s.only) o += "=" + k;
    for (auto& k : s.except) o += "!" + k;
    for (auto& kp : s.kpreds) o += std::string(kp.neg ? "~!" : "~") + (char)('0' + kp.op) + kp.s;
    if (!s.elems_only) o += "]";
  }
  return o;
}

// does the concrete dictionary path `id` match the pattern (same rules as HostPlan::resolve_paths)?
bool pattern_matches(const Pattern& pat, const PathDict& dict, uint32_t id) {
  std::vector<PathDict::Info> chain;
  while (id != 0 && id != PathDict::kNone) { chain.push_back(dict.info(id)); id = chain.back().parent; }
  if (chain.size() != pat.size()) return false;
  for (size_t i = 0; i < pat.size(); i++) {
    const PathDict::Info& in = chain[chain.size() - 1 - i];
    const PatStep& st = pat[i];
    if (!st.any) { if (in.is_elem || in.key != st.key) return false; continue; }
    bool kp_ok = true;
    for (auto& kp : st.kpreds) if (!key_pred_holds(kp, in.key, in.is_elem)) kp_ok = false;
    if (!kp_ok) return false;
    if (in.is_elem) { if (!st.only.empty()) return false; continue; }
    if (st.elems_only) return false;
    if (!st.only.empty() && std::find(st.only.begin(), st.only.end(), in.key) == st.only.end()) return false;
    if (std::find(st.except.begin(), st.except.end(), in.key) != st.except.end()) return false;
  }
  return true;
}

bool pattern_reaches(const Pattern& pat, const PathDict& dict, uint32_t id, bool* full) {
  std::vector<PathDict::Info> chain;
  while (id != 0 && id != PathDict::kNone) { chain.push_back(dict.info(id)); id = chain.back().parent; }
  *full = chain.size() == pat.size();
  if (chain.size() > pat.size()) return false;
  for (size_t i = 0; i < chain.size(); i++) {
    const PathDict::Info& in = chain[chain.size() - 1 - i];
    const PatStep& st = pat[i];
    if (!st.any) { if (in.is_elem || in.key != st.key) return false; continue; }
    bool kp_ok = true;
    for (auto& kp : st.kpreds) if (!key_pred_holds(kp, in.key, in.is_elem)) kp_ok = false;
    if (!kp_ok) return false;
    if (in.is_elem) { if (!st.only.empty()) return false; continue; }
    if (st.elems_only) return false;
    if (!st.only.empty() && std::find(st.only.begin(), st.only.end(), in.key) == st.only.end()) return false;
    if (std::find(st.except.begin(), st.except.end(), in.key) != st.except.end()) return false;
  }
  return true;
}

// can two leaf patterns cover the same concrete path?
static bool patterns_overlap(const Pattern& a, const Pattern& b) {
  if (a.size() != b.size()) return false;
  auto admits = [](const PatStep& any, const std::string& key) {
    if (any.elems_only) return false;
    for (auto& kp : any.kpreds) if (!key_pred_holds(kp, key, false)) return false;
    if (!any.only.empty() && std::find(any.only.begin(), any.only.end(), key) == any.only.end()) return false;
    return std::find(any.except.begin(), any.except.end(), key) == any.except.end();
  };
  for (size_t i = 0; i < a.size(); i++) {
    const PatStep &x = a[i], &y = b[i];
    if (!x.any && !y.any) { if (x.key != y.key) return false; }
    else if (!x.any) { if (!admits(y, x.key)) return false; }
    else if (!y.any) { if (!admits(x, y.key)) return false; }
  }
  return true;
}

bool match_group_pattern(const Pattern& p) {
  return p.size() == 3 && !p[0].any && p[0].key == "$m" && !p[1].any && !p[2].any && !p[2].key.empty() && p[2].key[0] != '$';
}

std::atomic<int> g_debug_dict_facts{1};   // gk_debug_set("dict_facts", 0): a row per leaf, as before round 6 (test aid)
bool review_fact_pattern(const Pattern& p) {
  static const bool on = !(getenv("GK_DICT_FACTS") && atoi(getenv("GK_DICT_FACTS")) == 0);   // (A/B aid, read once: 0 keeps a row per leaf)
  if (!on || !g_debug_dict_facts.load(std::memory_order_relaxed) || p.empty() || match_group_pattern(p)) return false;
  for (const PatStep& st : p) if (st.any || st.key.empty()) return false;
  // what lies INSIDE the candidate objects and the labels of the review's Namespace: every such leaf goes through the subtree parsers,
  // which evaluate the dictionary expressions of whatever they meet.  (The members of the request envelope -- kind, operation, name,
  // userInfo ... -- and the roots `object` / `oldObject` themselves are written by the envelope code: their tests stay row predicates.)
  if (p.size() >= 2 && (p[0].key == "object" || p[0].key == "oldObject")) { for (size_t i = 1; i < p.size(); i++) if (p[i].key[0] == '$') return false; return true; }
  return p.size() == 4 && p[0].key == "$ns" && p[1].key == "metadata" && p[2].key == "labels" && p[3].key[0] != '$';
}

uint32_t DictRegistry::intern(const Pattern& leaf_in, const DX& dx, bool add, bool* is_facts, bool has_fallback) {
  // Canonical form: an unfiltered iteration step is registered as "any child", whether the lowering reached the leaf through
  // an explicit element loop (elements only) or through a flat wildcard predicate (members and elements) -- the same leaf
  // must own ONE pattern, because bit numbers are per pattern and a path has one $d row.  (The device predicates keep their
  // own, exact patterns; a $d row on a path only the wider form covers is merely never read.)
  Pattern leaf = leaf_in;
  for (PatStep& st : leaf) if (st.any && st.only.empty() && st.except.empty() && st.kpreds.empty()) st.elems_only = false;
  const std::string pk = pattern_to_string(leaf), dk = dx_to_string(dx);
  std::unique_lock<std::shared_mutex> l(mu_);
  Pat* p = nullptr;
  for (auto& x : pats_) if (x.key == pk) p = &x;
  if (!p && !add) throw std::runtime_error("needs a dictionary predicate no loaded constraint registered");
  if (!p) {
    // a different pattern that covers some of the same paths (a constant member next to an iteration over the members of the
    // same object) would need two $d rows on one path: refused here, at AddConstraint -- never at table creation
    for (auto& x : pats_) if (patterns_overlap(x.pat, leaf)) throw std::runtime_error("dictionary predicates on overlapping leaf patterns (" + x.key + " and " + pk + ")");
    pats_.push_back({leaf, pk, {}}); p = &pats_.back();
    // REVIEW FACTS: a leaf without an iteration on the way joins the shared row -- when its first expression comes with a
    // row-predicate fallback (a promoted test: if the 62 shared bits run out the lowering reads the leaf's rows instead).  A leaf whose
    // first expression has none (quantity arithmetic, a deep expression) keeps its own row and its own 62 bits: it must never
    // become unloadable because other leaves took the shared ones.
    p->facts = has_fallback && review_fact_pattern(leaf);
  }
  if (is_facts) *is_facts = p->facts;
  for (auto& e : p->entries) if (e.key == dk) return e.bit;
  if (!add) throw std::runtime_error("needs a dictionary predicate no loaded constraint registered");
  // MATCH GROUP: the facts of one candidate (review.$m.<o|old>.<fact>) travel in ONE row, review.$m.<o|old>.$d -- their
  // expressions take their bits from one space (the flattener ORs the facts' masks, match_group_row)
  size_t used = p->entries.size();
  if (match_group_pattern(leaf)) { used = 0; for (auto& x : pats_) if (match_group_pattern(x.pat) && x.pat[1].key == leaf[1].key) used += x.entries.size(); }
  if (p->facts) { used = 0; for (auto& x : pats_) if (x.facts) used += x.entries.size(); }   // (one bit space for every leaf of the review facts row)
  if (used >= 62) {
    if (p->entries.empty()) { pats_.pop_back(); }   // (a pattern created for this expression alone does not stay behind empty)
    throw std::runtime_error("more than 62 dictionary predicates on " + pk);
  }
  { static const bool dbg = getenv("GK_DEBUG_DICT") != nullptr; if (dbg) fprintf(stderr, "[gkgpu dict] %s%s bit %zu: %s\n", pk.c_str(), p->facts ? " (review facts)" : "", used, dk.substr(0, 300).c_str()); }
  p->entries.push_back({dx, dk, (uint32_t)used});
  p->memo.clear();
  gen_++;
  return p->entries.back().bit;
}
uint64_t DictRegistry::gen() const {
  uint64_t g = gen_.load(std::memory_order_acquire);
  if (const DictRegistry* c = counting_if_any()) g += c->gen() << 32;   // (a new counting expression makes earlier tables stale as well)
  return g;
}
bool DictRegistry::memo_get(int pi, size_t n_entries, const std::string& key, uint64_t* mask) const {
  std::shared_lock<std::shared_mutex> l(mu_);
  if (pi < 0 || (size_t)pi >= pats_.size() || pats_[pi].entries.size() != n_entries) return false;
  auto it = pats_[pi].memo.find(key);
  if (it == pats_[pi].memo.end()) return false;
  *mask = it->second;
  return true;
}
void DictRegistry::memo_put(int pi, size_t n_entries, const std::string& key, uint64_t mask) {
  std::unique_lock<std::shared_mutex> l(mu_);
  if (pi < 0 || (size_t)pi >= pats_.size() || pats_[pi].entries.size() != n_entries) return;   // the pattern gained an expression meanwhile
  if (pats_[pi].memo.size() < 262144) pats_[pi].memo.emplace(key, mask);
}
bool DictRegistry::facts_get(uint64_t st, uint32_t path, uint8_t bit, Facts* out) const {
  std::shared_lock<std::shared_mutex> l(fmu_);
  if (facts_stamp_ != st || path >= facts_.size() || !(facts_[path].known & bit)) return false;
  *out = facts_[path];
  return true;
}
void DictRegistry::match(const PathDict& dict, uint32_t path_id, std::vector<DictEntry>* out, int* pat_index, bool* is_facts) const {
  const uint64_t st = stamp();
  Facts f;
  if (is_facts) *is_facts = false;
  if (facts_get(st, path_id, F_PAT, &f)) {
    std::shared_lock<std::shared_mutex> l(mu_);
    out->clear();
    if (pat_index) *pat_index = -1;
    if (f.pat >= 0 && (size_t)f.pat < pats_.size()) { *out = pats_[f.pat].entries; if (pat_index) *pat_index = f.pat; if (is_facts) *is_facts = pats_[f.pat].facts; }
    return;
  }
  int found = -1;
  struct Put { const DictRegistry* r; uint64_t st; uint32_t path; int* found; ~Put() { const int v = *found; r->facts_put(st, path, F_PAT, [v](Facts& x) { x.pat = v; }); } } put{this, st, path_id, &found};
  std::shared_lock<std::shared_mutex> l(mu_);
  out->clear();
  if (pat_index) *pat_index = -1;
  // several patterns may cover one concrete path: their bit numbers are per PATTERN, so only one pattern may own a path's
  // $d row -- the lowering registers element / key iterations in canonical form, which makes overlapping patterns equal
  for (size_t i = 0; i < pats_.size(); i++) {
    const Pat& p = pats_[i];
    if (!pattern_matches(p.pat, dict, path_id)) continue;
    if (!out->empty()) throw std::runtime_error("overlapping dictionary patterns on " + dict.to_string(path_id));
    *out = p.entries;
    if (pat_index) *pat_index = (int)i;
    if (is_facts) *is_facts = p.facts;
    found = (int)i;
  }
}

bool DictRegistry::add_guard(const Pattern& container, bool add) {
  const std::string k = pattern_to_string(container);
  std::unique_lock<std::shared_mutex> l(mu_);
  for (auto& g : guards_) if (g.first == k) return true;
  if (!add) return false;
  guards_.emplace_back(k, container);
  gen_++;
  return true;
}
bool DictRegistry::guarded(const PathDict& dict, uint32_t path_id) const {
  const uint64_t st = stamp();
  Facts f;
  if (facts_get(st, path_id, F_GUARD, &f)) return f.guarded;
  bool r = false;
  {
    std::shared_lock<std::shared_mutex> l(mu_);
    for (const auto& g : guards_) if (pattern_matches(g.second, dict, path_id)) { r = true; break; }
  }
  facts_put(st, path_id, F_GUARD, [r](Facts& x) { x.guarded = r; });
  return r;
}

bool DictRegistry::add_carrier(const Pattern& elem, const std::string& member, bool add, std::string* chosen) {
  const std::string k = pattern_to_string(elem);
  std::unique_lock<std::shared_mutex> l(mu_);
  for (auto& c : carriers_) if (c.key == k) { if (chosen) *chosen = c.member; return true; }
  if (!add || member.empty()) return false;
  carriers_.push_back(Carrier{k, elem, member});
  gen_++;   // tables flattened before this lack the T_ABSENT rows: they are stale (engine.cpp dict_gen)
  if (chosen) *chosen = member;
  return true;
}
bool DictRegistry::carrier_of(const PathDict& dict, uint32_t elem_path_id, std::string* member) const {
  std::shared_lock<std::shared_mutex> l(mu_);
  for (const auto& c : carriers_) if (pattern_matches(c.elem, dict, elem_path_id)) { if (member) *member = c.member; return true; }
  return false;
}

bool DictRegistry::add_value(const Pattern& leaf, bool add) {
  const std::string k = pattern_to_string(leaf);
  std::unique_lock<std::shared_mutex> l(mu_);
  for (auto& g : values_) if (g.first == k) return true;
  if (!add) return false;
  values_.emplace_back(k, leaf);
  gen_++;   // tables flattened before this do not carry the ids: they are stale (engine.cpp dict_gen)
  return true;
}
bool DictRegistry::set_reads(const std::vector<Pattern>& pats) {
  std::vector<std::pair<std::string, Pattern>> v;
  for (const Pattern& p : pats) v.emplace_back(pattern_to_string(p), p);
  std::sort(v.begin(), v.end(), [](const std::pair<std::string, Pattern>& a, const std::pair<std::string, Pattern>& b) { return a.first < b.first; });
  v.erase(std::unique(v.begin(), v.end(), [](const std::pair<std::string, Pattern>& a, const std::pair<std::string, Pattern>& b) { return a.first == b.first; }), v.end());
  std::unique_lock<std::shared_mutex> l(mu_);
  bool same = v.size() == reads_.size();
  for (size_t i = 0; same && i < v.size(); i++) same = v[i].first == reads_[i].first;
  if (same) return false;
  reads_.swap(v);
  reads_gen_++;
  return true;
}
// every concrete path `p` matches is matched by `r` as well (sufficient, step by step; not necessary)
static bool pattern_covers(const Pattern& r, const Pattern& p) {
  if (r.size() != p.size()) return false;
  for (size_t i = 0; i < r.size(); i++) {
    const PatStep &x = r[i], &y = p[i];
    if (!x.any) { if (y.any || x.key != y.key) return false; continue; }
    const bool plain = x.only.empty() && x.except.empty() && x.kpreds.empty();
    if (plain && !x.elems_only) continue;                       // any child at all
    if (plain && x.elems_only && y.any && y.elems_only) continue;   // any element: whatever the other one asks of elements
    Pattern a{x}, b{y};
    if (pattern_to_string(a) != pattern_to_string(b)) return false;   // (the same filters)
  }
  return true;
}
bool DictRegistry::reads_has(const Pattern& pat) const {
  // (dictionary rows <leaf>.$d / <leaf>.$c and review.$dup are made by the flattener for every registered entry, pruned table or not)
  if (!pat.empty() && !pat.back().any && (pat.back().key == "$d" || pat.back().key == "$c" || pat.back().key == "$dup")) return true;
  std::shared_lock<std::shared_mutex> l(mu_);
  for (const auto& r : reads_) if (pattern_covers(r.second, pat)) return true;
  return false;
}
void DictRegistry::interest(std::vector<const Pattern*>* out) const {
  for (auto& p : pats_) out->push_back(&p.pat);
  for (auto& g : guards_) out->push_back(&g.second);
  for (auto& g : values_) out->push_back(&g.second);
  for (auto& g : keys_) out->push_back(&g.second);
}
uint32_t DictRegistry::read_state(const PathDict& dict, uint32_t path_id) const {
  const uint64_t stp = stamp();
  Facts f;
  if (facts_get(stp, path_id, F_READ, &f)) return f.read;
  const uint32_t r = read_state_now(dict, path_id);
  facts_put(stp, path_id, F_READ, [r](Facts& x) { x.read = (uint8_t)r; });
  return r;
}
uint32_t DictRegistry::read_state_now(const PathDict& dict, uint32_t path_id) const {
  uint32_t st = 0;
  bool full = false;
  std::shared_lock<std::shared_mutex> l(mu_);
  for (const auto& r : reads_) if (pattern_reaches(r.second, dict, path_id, &full)) { st |= 2u; if (full) st |= 1u; }
  if (st == 3u) return st;
  std::vector<const Pattern*> more;
  interest(&more);
  if (counting_) { std::shared_lock<std::shared_mutex> l2(counting_->mu_); counting_->interest(&more); for (const Pattern* p : more) if (pattern_reaches(*p, dict, path_id, &full)) st |= 2u; return st; }
  for (const Pattern* p : more) if (pattern_reaches(*p, dict, path_id, &full)) st |= 2u;
  return st;
}
bool DictRegistry::child_names(const PathDict& dict, uint32_t path_id, std::vector<std::string>* names) const {
  const uint64_t st = stamp();
  {
    std::shared_lock<std::shared_mutex> l(fmu_);
    if (facts_stamp_ == st) { auto it = names_.find(path_id); if (it != names_.end()) { *names = it->second.second; return it->second.first; } }
  }
  const bool listed = child_names_now(dict, path_id, names);
  if (st == stamp()) {
    std::unique_lock<std::shared_mutex> l(fmu_);
    if (facts_stamp_ != st) { facts_.clear(); names_.clear(); facts_stamp_ = st; }
    names_[path_id] = std::make_pair(listed, *names);
  }
  return listed;
}
bool DictRegistry::child_names_now(const PathDict& dict, uint32_t path_id, std::vector<std::string>* names) const {
  size_t depth = 0;
  for (uint32_t id = path_id; id != 0 && id != PathDict::kNone; id = dict.info(id).parent) depth++;
  std::shared_lock<std::shared_mutex> l(mu_);
  std::vector<const Pattern*> all;
  for (const auto& r : reads_) all.push_back(&r.second);
  interest(&all);
  std::unique_ptr<std::shared_lock<std::shared_mutex>> l2;
  if (counting_) { l2.reset(new std::shared_lock<std::shared_mutex>(counting_->mu_)); counting_->interest(&all); }
  bool full = false;
  for (const Pattern* p : all) {
    if (p->size() <= depth || !pattern_reaches(*p, dict, path_id, &full) || full) continue;
    const PatStep& st = (*p)[depth];
    if (st.any) return false;
    if (std::find(names->begin(), names->end(), st.key) == names->end()) names->push_back(st.key);
  }
  return true;
}
bool DictRegistry::add_key(const Pattern& leaf, bool add) {
  const std::string k = pattern_to_string(leaf);
  std::unique_lock<std::shared_mutex> l(mu_);
  for (auto& g : keys_) if (g.first == k) return true;
  if (!add) return false;
  keys_.emplace_back(k, leaf);
  gen_++;   // tables flattened before this do not carry review.$dup for these paths: they are stale (engine.cpp dict_gen)
  return true;
}
bool DictRegistry::keyed(const PathDict& dict, uint32_t path_id) const {
  const uint64_t st = stamp();
  Facts f;
  if (facts_get(st, path_id, F_KEY, &f)) return f.keyed;
  bool r = false;
  {
    std::shared_lock<std::shared_mutex> l(mu_);
    for (const auto& g : keys_) if (pattern_matches(g.second, dict, path_id)) { r = true; break; }
  }
  facts_put(st, path_id, F_KEY, [r](Facts& x) { x.keyed = r; });
  return r;
}
bool DictRegistry::valued(const PathDict& dict, uint32_t path_id) const {
  const uint64_t st = stamp();
  Facts f;
  if (facts_get(st, path_id, F_VALUE, &f)) return f.valued;
  bool r = false;
  {
    std::shared_lock<std::shared_mutex> l(mu_);
    for (const auto& g : values_) if (pattern_matches(g.second, dict, path_id)) { r = true; break; }
  }
  facts_put(st, path_id, F_VALUE, [r](Facts& x) { x.valued = r; });
  return r;
}

// ------------------------------------------------------------------------------------------------ host staging pool
namespace {
struct HostPool {
  std::mutex mu;
  std::unordered_map<size_t, std::vector<void*>> free_;   // size class -> blocks
  size_t cached = 0, limit = 0;
  HostPool() {
    const char* mb = getenv("GK_HOST_POOL_MB");
    limit = (size_t)(mb ? atoll(mb) : 8192) << 20;
  }
  ~HostPool() { for (auto& kv : free_) for (void* p : kv.second) free(p); }
};
HostPool& host_pool() { static HostPool p; return p; }
}  // namespace

size_t host_cpus() {
  static const size_t n = [] {
    size_t hw = std::max<size_t>(1, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) hw = std::min<size_t>(hw, (size_t)CPU_COUNT(&set));
    long long quota = 0, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
      char q[32] = {0};
      if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
      fclose(f);
    } else {
      if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = 0; fclose(fq); }
      if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
    }
    if (quota > 0 && period > 0) hw = std::min<size_t>(hw, (size_t)std::max<long long>(1, (quota + period - 1) / period));
    return hw;
  }();
  return n;
}

void* host_block_alloc(size_t bytes, size_t* cap_bytes) {
  size_t cls = kHostBlockMin;
  while (cls < bytes) cls <<= 1;
  *cap_bytes = cls;
  HostPool& P = host_pool();
  {
    std::lock_guard<std::mutex> l(P.mu);
    auto it = P.free_.find(cls);
    if (it != P.free_.end() && !it->second.empty()) { void* p = it->second.back(); it->second.pop_back(); P.cached -= cls; return p; }
  }
  void* p = nullptr;
  // (no MADV_HUGEPAGE: with defrag=madvise -- the usual setting -- the first touch compacts memory synchronously; measured
  //  here: table builds of 1 s turned into 3-13 s, erratically)
  if (posix_memalign(&p, 4096, cls) != 0) return nullptr;
  return p;
}

void host_block_free(void* p, size_t cap_bytes) {
  if (!p) return;
  HostPool& P = host_pool();
  {
    std::lock_guard<std::mutex> l(P.mu);
    if (P.cached + cap_bytes <= P.limit) { P.free_[cap_bytes].push_back(p); P.cached += cap_bytes; return; }
  }
  free(p);
}

// ------------------------------------------------------------------------------------------------ NsCache
void NsCache::put(const std::string& name, const Value& ns) { std::unique_lock<std::shared_mutex> l(mu_); m_[name] = ns; }
void NsCache::remove(const std::string& name) { std::unique_lock<std::shared_mutex> l(mu_); m_.erase(name); }
Value NsCache::get(const std::string& name) const {
  std::shared_lock<std::shared_mutex> l(mu_);
  auto it = m_.find(name);
  return it == m_.end() ? Value() : it->second;
}

// ------------------------------------------------------------------------------------------------ unstructured
std::string obj_string(const Value& obj, const char* a, const char* b) {
  const Value* v = obj.get(a);
  if (v && b) v = v->get(b);
  return (v && v->is_string()) ? v->str() : std::string();
}

void obj_gvk(const Value& obj, std::string* group, std::string* version, std::string* kind) {
  std::string av = obj_string(obj, "apiVersion");
  *kind = obj_string(obj, "kind");
  group->clear();
  version->clear();
  size_t n = std::count(av.begin(), av.end(), '/');   // schema.ParseGroupVersion
  if (av.empty() || av == "/") return;
  if (n == 0) *version = av;
  else if (n == 1) { size_t i = av.find('/'); *group = av.substr(0, i); *version = av.substr(i + 1); }
}

bool obj_is_namespace(const Value& obj) {
  std::string g, v, k;
  obj_gvk(obj, &g, &v, &k);
  return k == "Namespace" && g.empty();
}

// ------------------------------------------------------------------------------------------------ normalisation
namespace {
Value S(const std::string& s) { return Value::string(s); }
// a string LITERAL as a Value: made once (member names of the normalised request: a dozen per review otherwise)
#define SK(lit) ([]() -> const Value& { static const Value v_ = Value::string(lit); return v_; }())

Value str_field(const Value& o, const char* k) {
  const Value* v = o.get(k);
  return (v && v->is_string()) ? *v : SK("");
}

Value triple(const Value* o, const char* a, const char* b, const char* c) {
  ValuePairs p;
  Value empty = Value::object({});
  const Value& src = (o && o->is_object()) ? *o : empty;
  p.emplace_back(S(a), str_field(src, a));
  p.emplace_back(S(b), str_field(src, b));
  p.emplace_back(S(c), str_field(src, c));
  return Value::object(std::move(p));
}
}  // namespace

// Canonical JSON encoding of admissionv1.AdmissionRequest (k8s.io/api/admission/v1, third-party struct tags):
// uid/kind/resource/operation/userInfo always present; RawExtension fields encode as null when empty; the
// `omitempty` fields are dropped when zero.  namespaceObject comes from reviews.Namespace (pkg/util/namespace.go:15).
ReviewDoc normalize_admission_request(const Value& request, const Value& match_ns, const Value& ns_object, int source,
                                      const NsCache& cache) {
  if (!request.is_object()) throw ReviewError("invalid request object: AdmissionRequest must be a JSON object");
  ValuePairs p;
  p.emplace_back(SK("uid"), str_field(request, "uid"));
  p.emplace_back(SK("kind"), triple(request.get("kind"), "group", "version", "kind"));
  p.emplace_back(SK("resource"), triple(request.get("resource"), "group", "version", "resource"));
  Value op = str_field(request, "operation");
  p.emplace_back(SK("operation"), op);
  const Value* ui = request.get("userInfo");
  p.emplace_back(SK("userInfo"), (ui && ui->is_object()) ? *ui : Value::object({}));
  const Value* obj = request.get("object");
  const Value* old = request.get("oldObject");
  Value vobj = (obj && obj->is_object()) ? *obj : Value::null();
  Value vold = (old && old->is_object()) ? *old : Value::null();
  if (op.str() == "DELETE") {   // setObjectOnDelete
    if (!vold.is_object()) throw ReviewError("oldObject cannot be nil for DELETE operations");
    vobj = vold;
  }
  p.emplace_back(SK("object"), vobj);
  p.emplace_back(SK("oldObject"), vold);
  const Value* opts = request.get("options");
  p.emplace_back(SK("options"), opts ? *opts : Value::null());
  for (const char* k : {"subResource", "requestSubResource", "name", "namespace"}) {
    const Value* v = request.get(k);
    if (v && v->is_string() && !v->str().empty()) p.emplace_back(S(k), *v);
  }
  for (const char* k : {"requestKind", "requestResource", "dryRun"}) {
    const Value* v = request.get(k);
    if (v && !v->is_null()) p.emplace_back(S(k), *v);
  }
  if (ns_object.defined() && !ns_object.is_null()) p.emplace_back(SK("namespaceObject"), ns_object);
  ReviewDoc d;
  d.request = Value::object(std::move(p));
  d.source = source;
  d.match_ns = (match_ns.defined() && !match_ns.is_null()) ? match_ns : Value();
  if (!d.match_ns.defined()) {
    std::string rns = obj_string(d.request, "namespace");
    if (!rns.empty()) d.match_ns = cache.get(rns);   // matcher.go:37-39
  }
  return d;
}

ReviewDoc normalize_object(const Value& object, const Value& match_ns, const Value& ns_object, int source,
                           const std::string& operation, const NsCache& cache) {
  if (!object.is_object()) throw ReviewError("invalid request object: object must be a JSON object");
  std::string g, v, k;
  obj_gvk(object, &g, &v, &k);
  ValuePairs kind{{SK("group"), S(g)}, {SK("version"), S(v)}, {SK("kind"), S(k)}};
  ValuePairs req;
  req.emplace_back(SK("kind"), Value::object(std::move(kind)));
  req.emplace_back(SK("name"), S(obj_string(object, "metadata", "name")));
  req.emplace_back(SK("namespace"), S(obj_string(object, "metadata", "namespace")));
  if (!operation.empty()) req.emplace_back(SK("operation"), S(operation));
  if (operation == "DELETE") req.emplace_back(SK("oldObject"), object);   // target.go:151-154
  else req.emplace_back(SK("object"), object);
  return normalize_admission_request(Value::object(std::move(req)), match_ns, ns_object, source, cache);
}

// ------------------------------------------------------------------------------------------------ Flattener
Flattener::Flattener(PathDict* dict, const DictRegistry* reg) : dict_(dict), reg_(reg) {
  id_object_ = dict_->child(0, "object");
  id_old_ = dict_->child(0, "oldObject");
  id_m_ = dict_->child(0, "$m");
  id_ns_ = dict_->child(0, "$ns");
  for (int w = 0; w < 2; w++) {   // paths of the strings the match layer reads from object / oldObject (fast ingest)
    const uint32_t root = w ? id_old_ : id_object_;
    CapIds& c = cap_[w];
    c.api_version = dict_->child(root, "apiVersion");
    c.kind = dict_->child(root, "kind");
    c.metadata = dict_->child(root, "metadata");
    c.name = dict_->child(c.metadata, "name");
    c.ns = dict_->child(c.metadata, "namespace");
    c.gname = dict_->child(c.metadata, "generateName");
    c.labels = dict_->child(c.metadata, "labels");
  }
}

void Flattener::begin_table() {
  use_index_ = ix_supported() && !getenv("GK_NO_INDEX");
  ns_memo_.clear();
  ns_memo_name_.clear();
  stage_.clear();
  order_.clear();
  if (reg_) {
    const uint64_t g = reg_->gen();
    const uint64_t rg = reg_->reads_gen();
    if (g != reg_gen_ || rg != reads_gen_seen_) { dict_paths_.clear(); pbits_.clear(); kid_filter_.clear(); kid_arena_.clear(); carrier_cache_.clear(); reg_gen_ = g; reads_gen_seen_ = rg; }
  }
}

// the carrier member's path of an element path (plan.hpp T_ABSENT), 0 = none
uint32_t Flattener::carrier_slow(uint32_t elem_path) {
  if (elem_path >= carrier_cache_.size()) carrier_cache_.resize((size_t)elem_path * 2 + 64, 0);
  std::string member;
  uint32_t v = 1u;
  if (reg_ && reg_->carrier_of(*dict_, elem_path, &member)) v = 2u + child(elem_path, member);
  if (elem_path >= carrier_cache_.size()) carrier_cache_.resize((size_t)elem_path * 2 + 64, 0);   // (child() may have grown the dictionary, not this vector: kept for symmetry)
  carrier_cache_[elem_path] = v;
  return v == 1u ? 0u : v - 2u;
}

// is a dictionary predicate registered for this leaf path? (cached per path; the cache follows the registry's generation)
bool Flattener::dict_wanted(uint32_t path) {
  if (!reg_) return false;
  if (path >= dict_paths_.size()) dict_paths_.resize((size_t)path * 2 + 64);
  DictPath& d = dict_paths_[path];
  if (d.state == 0) {
    reg_->match(*dict_, path, &d.entries, &d.pat, &d.facts);
    d.centries.clear(); d.cpat = -1;
    if (const DictRegistry* c = reg_->counting_if_any()) c->match(*dict_, path, &d.centries, &d.cpat);
    d.state = d.entries.empty() && d.centries.empty() ? 1 : 2;
    d.deep = false;
    for (const DictEntry& e : d.entries) if (dx_deep(e.dx)) d.deep = true;
    for (const DictEntry& e : d.centries) if (dx_deep(e.dx)) d.deep = true;
    if (!d.entries.empty() && !d.facts) d.dpath = child(path, "$d");
    if (!d.centries.empty()) d.cpath = child(path, "$c");
  }
  return dict_paths_[path].state == 2;
}

bool Flattener::guard_wanted(uint32_t path) {
  if (!reg_) return false;
  if (path >= dict_paths_.size()) dict_paths_.resize((size_t)path * 2 + 64);
  DictPath& d = dict_paths_[path];
  if (d.gstate == 0) d.gstate = reg_->guarded(*dict_, path) ? 2 : 1;
  return d.gstate == 2;
}

void Flattener::dict_row(uint32_t path, uint32_t meta, const Value& leaf, uint64_t* masks_out) {
  DictPath& d = dict_paths_[path];
  // containers count by type and size only (the registered expressions cannot look inside them: pe.cpp scalar_fns) -- unless
  // an expression of this path is DEEP (dexpr.hpp): then the leaf is the real sub-document and the memo goes by its text
  std::string key = ((leaf.is_array() || leaf.is_object() || leaf.is_set()) && !d.deep) ? std::to_string(leaf.size()) : to_term_string(leaf);
  key.push_back((char)('0' + (int)leaf.kind));
  // (both masks are worked out BEFORE a row is emitted: emit() may grow dict_paths_ -- the new row's path id asks value_wanted /
  //  key_wanted -- and `d` would dangle)
  uint64_t masks[2] = {0, 0};
  const uint32_t dpaths[2] = {d.dpath, d.cpath};
  auto one = [&](const DictRegistry* reg, int pat, const std::vector<DictEntry>& entries, std::unordered_map<std::string, uint64_t>& memo, uint64_t* out) {
    if (entries.empty()) return;
    auto it = memo.find(key);
    uint64_t mask;
    if (it != memo.end()) mask = it->second;
    else {
      if (!reg->memo_get(pat, entries.size(), key, &mask)) {   // first sight of this value in the whole engine
        mask = 0;
        for (const DictEntry& e : entries) if (dx_true(e.dx, leaf)) mask |= 1ull << e.bit;
        const_cast<DictRegistry*>(reg)->memo_put(pat, entries.size(), key, mask);
      }
      if (memo.size() < 65536) memo.emplace(key, mask);
    }
    *out = mask;
  };
  one(reg_, d.pat, d.entries, d.memo, &masks[0]);
  if (!d.centries.empty()) one(reg_->counting_if_any(), d.cpat, d.centries, d.cmemo, &masks[1]);   // (<leaf>.$c: the counting plans' expressions)
  if (masks_out) { masks_out[0] = masks[0]; masks_out[1] = masks[1]; return; }
  if (dict_paths_[path].facts) { facts_acc_ |= masks[0]; masks[0] = 0; }   // (a leaf of the review facts row: review.$r.$d, finish_tail)
  for (int k = 0; k < 2; k++)
    if (masks[k]) emit(dpaths[k], (meta & ~(uint32_t)ROW_TYPE_MASK & ~(uint32_t)ROW_STR_INLINE) | T_INT, (uint32_t)masks[k], (uint32_t)(masks[k] >> 32), true);
}

// dict_row for a STRING leaf given as bytes: the answers are remembered per path under the bytes themselves (no Value, no quoted
// term text, no std::string key per row -- the dictionary leaves of a policy set are mostly strings: images, quantities, names)
void Flattener::dict_row_str(uint32_t path, uint32_t meta, const char* s, uint32_t n, uint64_t* masks_out, bool use_memo) {
  if (!use_memo) {   // a leaf whose values are as good as unique (a match candidate's name): straight evaluation where the expressions allow
    DictPath& d = dict_paths_[path];
    bool ok = true;
    uint64_t m[2] = {0, 0};
    dx_split_forget();
    for (const DictEntry& e : d.entries) { if (dx_true_str(e.dx, s, n, &ok)) m[0] |= 1ull << e.bit; if (!ok) break; }
    if (ok) for (const DictEntry& e : d.centries) { if (dx_true_str(e.dx, s, n, &ok)) m[1] |= 1ull << e.bit; if (!ok) break; }
    if (ok && masks_out) { masks_out[0] = m[0]; masks_out[1] = m[1]; return; }
  }
  {
    // (the entries compiled for string values: built once per path and registry generation)
    DictPath& d = dict_paths_[path];
    bool all_const = true;
    for (int k = 0; k < 2; k++) {
      const std::vector<DictEntry>& es = k ? d.centries : d.entries;
      if (es.empty()) continue;
      if (!d.sprog[k]) { d.sprog[k].reset(new DxStrProg()); d.sprog[k]->build(es); }
      all_const = all_const && d.sprog[k]->constant;
    }
    if (all_const) {   // no expression of this leaf looks at the bytes: one answer for every string
      uint64_t m[2] = {0, 0};
      for (int k = 0; k < 2; k++) { const std::vector<DictEntry>& es = k ? d.centries : d.entries; for (size_t i = 0; i < es.size(); i++) if (d.sprog[k]->constant_true[i]) m[k] |= 1ull << es[i].bit; }
      if (masks_out) { masks_out[0] = m[0]; masks_out[1] = m[1]; return; }
      if (d.facts) { facts_acc_ |= m[0]; m[0] = 0; }
      const uint32_t dp[2] = {d.dpath, d.cpath};
      for (int k = 0; k < 2; k++)
        if (m[k]) emit(dp[k], (meta & ~(uint32_t)ROW_TYPE_MASK & ~(uint32_t)ROW_STR_INLINE) | T_INT, (uint32_t)m[k], (uint32_t)(m[k] >> 32), true);
      return;
    }
  }
  uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)n * 0xD6E8FEB86659FD93ull);
  {
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, s + i, 8); h = (h ^ w) * 0x9FB21C651E98DF25ull; h ^= h >> 32; }
    if (i < n) { uint64_t w = 0; memcpy(&w, s + i, n - i); h = (h ^ w) * 0x9FB21C651E98DF25ull; h ^= h >> 32; }
    h |= 1;   // (0 marks an empty slot)
  }
  uint64_t masks[2] = {0, 0};
  bool found = false, bypass = false;
  {
    DictPath& d = dict_paths_[path];
    if (!d.smemo) { d.smemo.reset(new StrMemo()); d.smemo->tab.resize(256); }
    StrMemo& M = *d.smemo;
    if (M.bypass_left) { M.bypass_left--; bypass = true; }
    else {
      size_t mask = M.tab.size() - 1, i = (size_t)(h >> 7) & mask;
      for (; M.tab[i].hash; i = (i + 1) & mask) {
        const StrMemo::Ent& e = M.tab[i];
        if (e.hash == h && e.len == n && memcmp(M.arena.data() + e.off, s, n) == 0) { masks[0] = e.m[0]; masks[1] = e.m[1]; found = true; break; }
      }
      M.hits += found ? 1u : 0u;
      if (++M.seen == 8192u) { if (M.hits * 8u < M.seen) M.bypass_left = 7u * 8192u; M.seen = M.hits = 0; }
    }
  }
  if (!found) {
    {
      // string tests against constants are evaluated here, on the bytes (dx_true_str); anything else -- quantity arithmetic, regular
      // expressions, split components -- goes through the engine-wide memo of the pattern (dict_row), evaluated once per engine
      // (PER ENTRY since round 6: one expression the byte evaluator does not know no longer sends the leaf's whole list through the
      //  generic evaluator -- and through the engine-wide memo, which a column of unique values only fills)
      DictPath& d0 = dict_paths_[path];
      Value leaf;
      for (int k = 0; k < 2; k++) {
        const std::vector<DictEntry>& es = k ? d0.centries : d0.entries;
        if (es.empty()) continue;
        if (!d0.sprog[k]) { d0.sprog[k].reset(new DxStrProg()); d0.sprog[k]->build(es); }   // (dict_paths_ is dropped with the registry's generation: the program follows the entries)
        const DxStrProg& P = *d0.sprog[k];
        P.begin_value();
        for (size_t i = 0; i < es.size(); i++) {
          bool hit;
          if (!P.generic[i]) hit = P.eval(i, s, n);
          else { if (!leaf.defined()) leaf = Value::string(std::string(s, n)); hit = dx_true(es[i].dx, leaf); }
          if (hit) masks[k] |= 1ull << es[i].bit;
        }
      }
    }
    DictPath& d = dict_paths_[path];
    StrMemo& M = *d.smemo;
    if (M.count < 65536 && !bypass) {
      if ((M.count + 1) * 2 > M.tab.size()) {
        std::vector<StrMemo::Ent> old;
        old.swap(M.tab);
        M.tab.resize(old.size() * 2);
        const size_t mk = M.tab.size() - 1;
        for (const StrMemo::Ent& e : old) if (e.hash) { size_t j = (size_t)(e.hash >> 7) & mk; while (M.tab[j].hash) j = (j + 1) & mk; M.tab[j] = e; }
      }
      const size_t mk = M.tab.size() - 1;
      size_t j = (size_t)(h >> 7) & mk;
      while (M.tab[j].hash) j = (j + 1) & mk;
      M.tab[j] = StrMemo::Ent{h, (uint32_t)M.arena.size(), n, {masks[0], masks[1]}};
      M.arena.append(s, n);
      M.count++;
    }
  }
  if (masks_out) { masks_out[0] = masks[0]; masks_out[1] = masks[1]; return; }
  if (dict_paths_[path].facts) { facts_acc_ |= masks[0]; masks[0] = 0; }   // (a leaf of the review facts row)
  const uint32_t dpaths[2] = {dict_paths_[path].dpath, dict_paths_[path].cpath};
  for (int k = 0; k < 2; k++)
    if (masks[k]) emit(dpaths[k], (meta & ~(uint32_t)ROW_TYPE_MASK & ~(uint32_t)ROW_STR_INLINE) | T_INT, (uint32_t)masks[k], (uint32_t)(masks[k] >> 32), true);
}

// One match fact of a candidate (review.$m.<o|old>.<fact>): its string row (where some plan still reads it) and, OR-ed into `acc`,
// the answers of the dictionary expressions registered on it -- the candidate's facts share one dictionary row (match_group_row)
void Flattener::match_fact(uint32_t path, const char* s, uint32_t n, uint64_t* acc, bool use_memo) {
  emit_str_n(path, 0, s, n);
  if (dict_wanted(path)) {
    uint64_t m[2] = {0, 0};
    dict_row_str(path, 0, s, n, m, use_memo);
    acc[0] |= m[0]; acc[1] |= m[1];
  }
}
void Flattener::match_group_row(int w, const uint64_t* acc) {
  if (acc[0]) emit(env_.m_d[w], T_INT, (uint32_t)acc[0], (uint32_t)(acc[0] >> 32), true);
  if (acc[1]) emit(env_.m_c[w], T_INT, (uint32_t)acc[1], (uint32_t)(acc[1] >> 32), true);
}

bool Flattener::value_wanted(uint32_t path) {
  if (!reg_) return false;
  if (path >= dict_paths_.size()) dict_paths_.resize((size_t)path * 2 + 64);
  DictPath& d = dict_paths_[path];
  if (d.vstate == 0) d.vstate = reg_->valued(*dict_, path) ? 2 : 1;
  return d.vstate == 2;
}

// VALUE ID of a row (plan.hpp): equal ids <=> equal Rego values, within one review.  Numbers are interned by numeric value
// (an integral float is the integer: 1 == 1.0), strings by their bytes (an inline string's payload IS its bytes; heap strings
// compare hash, length and bytes), the five valueless kinds have fixed ids.  A non-empty container gets none (0): its
// equality would need a deep comparison, the predicate that wants the id then refuses the review.
uint32_t Flattener::value_id(uint32_t meta, uint32_t lo, uint32_t hi) {
  const uint32_t t = meta & ROW_TYPE_MASK;
  VidEnt want{0, 0, 0, 0};
  switch (t) {
    case T_NULL: return GK_VID_NULL;
    case T_BOOL: return lo ? GK_VID_TRUE : GK_VID_FALSE;
    case T_ARRAY: return lo == 0 ? GK_VID_EMPTY_ARRAY : 0u;
    case T_OBJECT: return lo == 0 ? GK_VID_EMPTY_OBJECT : 0u;
    case T_INT: want.tag = 1; want.key = ((uint64_t)hi << 32) | lo; break;
    case T_FLOAT: {
      const uint64_t bits = ((uint64_t)hi << 32) | lo;
      double d;
      memcpy(&d, &bits, 8);
      if (d >= -9223372036854775808.0 && d < 9223372036854775808.0 && d == (double)(int64_t)d && !(meta & ROW_INEXACT)) { want.tag = 1; want.key = (uint64_t)(int64_t)d; }
      else { want.tag = 2; want.key = bits; }
      break;
    }
    case T_STRING:
      if (meta & ROW_STR_INLINE) { want.tag = 3; want.key = ((uint64_t)hi << 32) | lo; }
      else { uint32_t n; memcpy(&n, &t_->heap[lo - 4], 4); want.tag = 4; want.key = ((uint64_t)n << 32) | hi; want.off = lo; }
      break;
    default: return 0u;
  }
  for (const VidEnt& e : vids_) {
    if (e.tag != want.tag || e.key != want.key) continue;
    if (want.tag != 4 || e.off == want.off || memcmp(&t_->heap[e.off], &t_->heap[want.off], (size_t)(want.key >> 32)) == 0) return e.id;
  }
  want.id = GK_VID_FIRST + (uint32_t)vids_.size();
  if (want.id >= GK_VID_OVERFLOW) return GK_VID_OVERFLOW;   // more distinct compared values than ids: the review is refused where one is needed
  vids_.push_back(want);
  return want.id;
}

bool Flattener::key_wanted(uint32_t path) {
  if (!reg_) return false;
  if (path >= dict_paths_.size()) dict_paths_.resize((size_t)path * 2 + 64);
  DictPath& d = dict_paths_[path];
  if (d.kstate == 0) d.kstate = reg_->keyed(*dict_, path) ? 2 : 1;
  return d.kstate == 2;
}

uint32_t Flattener::read_state(uint32_t path) {
  if (path >= dict_paths_.size()) dict_paths_.resize((size_t)path * 2 + 64);
  DictPath& d = dict_paths_[path];
  if (d.rstate == 0) d.rstate = 4u | reg_->read_state(*dict_, path);
  return d.rstate & 3u;
}

const Flattener::KidFilter* Flattener::kid_filter(uint32_t path) {
  if (path >= kid_filter_.size()) kid_filter_.resize((size_t)path * 2 + 64);
  if (!kid_filter_[path]) {
    std::vector<std::string> names;
    const bool listed = reg_->child_names(*dict_, path, &names) && names.size() <= 24;
    std::unique_ptr<KidFilter> f(new KidFilter());
    if (listed) for (const std::string& n : names) { f->kids.push_back({child(path, n), (uint32_t)kid_arena_.size(), (uint32_t)n.size()}); kid_arena_ += n; }
    f->state = listed ? 2 : 1;
    kid_filter_[path] = std::move(f);
  }
  return kid_filter_[path]->state == 2 ? kid_filter_[path].get() : nullptr;
}

uint8_t Flattener::pbits_slow(uint32_t path) {
  uint8_t b = PB_KNOWN;
  if (!pruning_) b |= PB_ROW | PB_BELOW;
  else { const uint32_t r = read_state(path); if (r & 1u) b |= PB_ROW; if (r & 2u) b |= PB_BELOW; }
  if (value_wanted(path)) b |= PB_VALUE;
  if (key_wanted(path)) b |= PB_KEY;
  if (dict_wanted(path)) { b |= PB_DICT; if (dict_paths_[path].deep) b |= PB_DEEP; }
  if (guard_wanted(path)) b |= PB_GUARD;
  if (path >= pbits_.size()) pbits_.resize((size_t)path * 2 + 64, 0);
  pbits_[path] = b;
  return b;
}

bool Flattener::emit(uint32_t path, uint32_t meta, uint32_t lo, uint32_t hi, bool always) {
  const uint8_t pb = pbits(path);
  if (!always && !(pb & PB_ROW)) {
    // a pruned table holds no row of this path -- what the flattener itself derives from the value still happens: a message key
    // is still compared with the review's other keys (review.$dup)
    if (pb & PB_KEY) {
      emit_side_effects_ = true;
      const uint32_t id = value_id(meta, lo, hi);
      if (id == 0u || id == GK_VID_OVERFLOW || std::find(key_ids_.begin(), key_ids_.end(), id) != key_ids_.end()) dup_seen_ = true;
      else key_ids_.push_back(id);
    }
    return false;
  }
  uint32_t rev = rev_cur_;
  if (pb & PB_VALUE) {
    const uint32_t vid = value_id(meta, lo, hi);
    if (vid == 0u || vid >= GK_VID_OVERFLOW) review_flags_ |= RF_HOST_CAND;
    rev |= vid << ROW_VID_SHIFT; emit_side_effects_ = true;
  }
  if (pb & PB_KEY) {   // a message key: equal values within the review (or one without an id) -> review.$dup (finish_review)
    emit_side_effects_ = true;
    const uint32_t id = value_id(meta, lo, hi);
    if (id == 0u || id == GK_VID_OVERFLOW || std::find(key_ids_.begin(), key_ids_.end(), id) != key_ids_.end()) dup_seen_ = true;
    else key_ids_.push_back(id);
  }
  stage_.push_back({path, Row{rev, meta, lo, hi}, StrHdr{{0, 0, 0, 0}}});
  return true;
}

// the row that exists for the element marker alone (plan.hpp T_ABSENT): no value, hence no value id, no message key, no dictionary row
void Flattener::emit_absent(uint32_t path, uint32_t meta) {
  if (!(pbits(path) & PB_ROW)) return;
  stage_.push_back({path, Row{rev_cur_, (meta & ~ROW_TYPE_MASK) | T_ABSENT, 0u, 0u}, StrHdr{{0, 0, 0, 0}}});
}

uint32_t Flattener::put_string(const std::string& s, uint32_t* hash) {
  // 16-byte aligned, zero-padded entry [u32 len][bytes][pad]: one aligned 16 B load fetches len + the first 12 bytes
  PodVec<uint8_t>& h = t_->heap;
  uint32_t n = (uint32_t)s.size();
  size_t at = (h.size() + 15) & ~(size_t)15;
  const size_t old_n = h.size(), end = at + ((4 + (size_t)n + 15) & ~(size_t)15);
  h.resize(end);
  memset(&h[old_n], 0, at - old_n);
  memset(&h[end - 16], 0, 16);     // zero padding: the device compares whole words
  memcpy(&h[at], &n, 4);
  memcpy(&h[at + 4], s.data(), n);
  *hash = hash32((const uint8_t*)s.data(), n);
  return (uint32_t)(at + 4);
}

void Flattener::emit_string_row(uint32_t path, uint32_t meta, const std::string& s) {
  if (pruning_ && !(read_state(path) & 1u) && !key_wanted(path)) return;   // (no row of this path in a pruned table: no heap entry either)
  if (s.size() <= 7) {   // inline: no heap entry, no memory access on the device
    uint64_t bits = 0;
    memcpy(&bits, s.data(), s.size());
    emit(path, meta | T_STRING | ROW_STR_INLINE, (uint32_t)bits, (uint32_t)(bits >> 32) | ((uint32_t)s.size() << 24));
    return;
  }
  uint32_t hsh, off = put_string(s, &hsh);
  if (emit(path, meta | T_STRING, off, hsh)) memcpy(&stage_.back().hdr, &t_->heap[off - 4], 16);   // entry header: length + first 12 bytes
}

void Flattener::emit_str(uint32_t parent, const char* key, const std::string& s) { emit_string_row(child(parent, key), 0, s); }

// Flattener-local memo of the shared dictionary: the hot lookups take no lock (one Flattener per host thread).
uint32_t Flattener::child(uint32_t parent, const std::string& key) { return fast_child(parent, key.data(), (uint32_t)key.size()); }
uint32_t Flattener::elem(uint32_t parent) {
  if (parent >= elem_cache_.size()) elem_cache_.resize((size_t)parent * 2 + 64, PathDict::kNone);
  uint32_t& id = elem_cache_[parent];
  if (id == PathDict::kNone) id = dict_->elem(parent);
  return id;
}

void Flattener::walk(const Value& v, uint32_t path, uint32_t ords, int adepth, uint32_t extra) {
  uint32_t meta = ords | extra;
  switch (v.kind) {
    case Value::Null: emit(path, meta | T_NULL, 0, 0); if (dict_wanted(path)) dict_row(path, meta, v); break;
    case Value::Bool: emit(path, meta | T_BOOL, v.b ? 1 : 0, 0); if (dict_wanted(path)) dict_row(path, meta, v); break;
    case Value::Number:
      if (dict_wanted(path)) dict_row(path, meta, v);
      if (v.is_int && v.i >= (i128)INT64_MIN && v.i <= (i128)INT64_MAX) {
        uint64_t u = (uint64_t)(int64_t)v.i;
        emit(path, meta | T_INT, (uint32_t)u, (uint32_t)(u >> 32));
      } else {
        double d = v.as_double();
        uint64_t u;
        memcpy(&u, &d, 8);
        emit(path, meta | T_FLOAT | (v.is_int ? ROW_INEXACT : 0), (uint32_t)u, (uint32_t)(u >> 32));
      }
      break;
    case Value::String: emit_string_row(path, meta, v.str()); if (dict_wanted(path)) dict_row(path, meta, v); break;
    case Value::Object: {
      emit(path, meta | T_OBJECT, (uint32_t)v.size(), 0);
      if (v.size() && guard_wanted(path)) review_flags_ |= RF_REFUSE;
      if (dict_wanted(path)) dict_row(path, meta, v);
      for (const auto& kv : v.pairs()) {
        const uint32_t ch = child(path, kv.first.str());
        // GK_TABLE_PRUNED, as the one-pass parser: a member nothing reaches is not walked (metadata and its members always are)
        if (pruning_ && !(read_state(ch) & 2u) && !walk_always(path, ch)) continue;
        walk(kv.second, ch, ords, adepth, extra);
      }
      break;
    }
    case Value::Array: case Value::Set: {
      emit(path, meta | T_ARRAY, (uint32_t)v.size(), 0);
      if (dict_wanted(path)) dict_row(path, meta, v);
      uint32_t ep = elem(path);
      Ctr* c = nullptr;
      for (auto& x : ctrs_) if (x.path == ep) { c = &x; break; }
      if (!c) { ctrs_.push_back({ep, 0}); c = &ctrs_.back(); }
      size_t ci = c - &ctrs_[0];
      for (const Value& e : v.items()) {
        uint32_t ord = ctrs_[ci].n++;
        uint32_t ex = extra, o2 = ords;
        if (adepth < 3) {
          if (ord >= 255) { ord = 255; ex |= ROW_ORD_OVERFLOW; review_flags_ |= RF_TOO_BIG; }
          o2 |= ord << (ROW_E_SHIFT0 + 8 * adepth);
        } else ex |= ROW_DEEP;
        if (pruning_ && !(read_state(ep) & 2u)) continue;   // (the ordinal is taken: the elements that ARE walked count as in the parser)
        walk(e, ep, o2, adepth + 1, ex);
        // element carrier (plan.hpp T_ABSENT): exactly one row at the carrier member's path per element
        if (const uint32_t cp = carrier_child(ep)) {
          bool has = false;
          if (e.is_object()) { const std::string& mk = dict_->info(cp).key; for (const auto& kv : e.pairs()) if (kv.first.is_string() && kv.first.str() == mk) { has = true; break; } }
          if (!has) emit_absent(cp, o2 | ex);
        }
      }
      break;
    }
    default: break;
  }
}

// members the pruned walk never skips: metadata of the candidate objects and what is directly in it (the one-pass parser captures the
// match layer's facts there)
bool Flattener::walk_always(uint32_t parent, uint32_t ch) {
  for (int k = 0; k < 2; k++) {
    const CapIds& c = cap_[k];
    if ((parent == (k ? id_old_ : id_object_) && ch == c.metadata) || parent == c.metadata) return true;
  }
  return false;
}

void Flattener::match_facts(const Value& obj, const Value& ns, bool is_old, uint32_t m_parent) {
  // facts of pkg/mutation/match/match.go needed per candidate object (object, then oldObject; matcher.go:44-71)
  std::string g, ver, k;
  obj_gvk(obj, &g, &ver, &k);
  bool is_ns = (k == "Namespace" && g.empty());
  std::string name = obj_string(obj, "metadata", "name");
  std::string nsfield = obj_string(obj, "metadata", "namespace");
  if (!env_.ready) env_init();
  const int w = is_old ? 1 : 0;
  uint32_t sub = env_.m_sub[w];
  (void)m_parent;
  emit(sub, T_OBJECT, 0, 0);
  uint64_t acc[2] = {0, 0};
  const std::string gname = obj_string(obj, "metadata", "generateName");
  match_fact(env_.m_group[w], g.data(), (uint32_t)g.size(), acc);
  match_fact(env_.m_kind[w], k.data(), (uint32_t)k.size(), acc);
  match_fact(env_.m_name[w], name.data(), (uint32_t)name.size(), acc, false);
  match_fact(env_.m_gname[w], gname.data(), (uint32_t)gname.size(), acc, false);
  bool has_nsname = true;
  std::string nsname;
  if (is_ns) nsname = name;
  else if (ns.defined()) nsname = obj_string(ns, "metadata", "name");
  else if (!nsfield.empty()) nsname = nsfield;
  else has_nsname = false;
  if (has_nsname) match_fact(env_.m_nsname[w], nsname.data(), (uint32_t)nsname.size(), acc);
  match_group_row(w, acc);
  review_flags_ |= is_old ? RF_HAS_OLD : RF_HAS_OBJ;
  if (is_ns) review_flags_ |= is_old ? RF_OLD_IS_NS : RF_OBJ_IS_NS;
  if (!nsfield.empty()) review_flags_ |= is_old ? RF_OLD_HAS_NSFIELD : RF_OBJ_HAS_NSFIELD;
  if (has_nsname) review_flags_ |= is_old ? RF_OLD_HAS_NSNAME : RF_OBJ_HAS_NSNAME;
  // unstructured GetLabels(): NestedStringMap fails (=> no labels) when any value is not a string
  const Value* md = obj.get("metadata");
  const Value* lb = md ? md->get("labels") : nullptr;
  bool bad = false;
  if (lb) {
    if (!lb->is_object()) bad = true;
    else for (const auto& kv : lb->pairs()) if (!kv.second.is_string()) bad = true;
  }
  if (bad) review_flags_ |= is_old ? RF_OLD_LABELS_BAD : RF_OBJ_LABELS_BAD;
}

void Flattener::add(const ReviewDoc& doc, HostTable* out) {
  t_ = out;
  rev_cur_ = out->n_reviews % out->rpt;
  ctrs_.clear();
  ctr_touched_.clear();
  review_flags_ = 0;
  vids_.clear();
  begin_review_keys();
  const Value& req = doc.request;
  // root + request members (input.review.*)
  emit(0, T_OBJECT, (uint32_t)req.size(), 0);
  for (const auto& kv : req.pairs()) walk(kv.second, child(0, kv.first.str()), 0, 0, 0);
  // $m: per-candidate match facts
  const Value& ns = doc.match_ns;
  emit(id_m_, T_OBJECT, 2, 0);
  const Value* obj = req.get("object");
  const Value* old = req.get("oldObject");
  if (obj && obj->is_object()) match_facts(*obj, ns, false, id_m_);
  if (old && old->is_object()) match_facts(*old, ns, true, id_m_);
  // gkReviewToObject (matcher.go:73-93): Unstructured.UnmarshalJSON rejects a document without a non-empty string `kind`
  if (obj && obj->is_object() && obj_string(*obj, "kind").empty()) review_flags_ |= RF_OBJ_BAD;
  if (old && old->is_object() && obj_string(*old, "kind").empty()) review_flags_ |= RF_OLD_BAD;
  finish_review(ns, doc.source, out);
}

void Flattener::add_skipped(HostTable* out) {
  ReviewDoc none;
  none.request = Value::object({});
  add(none, out);
  out->rflags.back() |= RF_SKIP;
}

// $ns rows (only what the match layer reads from Matchable.Namespace: name + labels), source flags, bookkeeping
void Flattener::finish_review(const Value& ns, int source, HostTable* out) {
  ns_rows(ns);
  finish_tail(source, out);
}
void Flattener::finish_review_memo(NsMemo* m, int source, HostTable* out) {
  if (m->rows_state == 1 && m->owner == out) {
    const uint32_t cur = out->n_reviews % out->rpt;
    for (const Staged& s : m->rows) { stage_.push_back(s); stage_.back().row.rev = cur; }
    review_flags_ |= m->flags;
    facts_acc_ |= m->facts;
  } else if (m->rows_state == 2) ns_rows(m->ns);
  else {   // record: the rows depend on the Namespace alone unless an emit interned a value id / compared a key or an array was counted
    const size_t s0 = stage_.size(), c0 = ctrs_.size(), t0 = ctr_touched_.size();
    const uint32_t f0 = review_flags_;
    const uint64_t a0 = facts_acc_;
    review_flags_ = 0;
    facts_acc_ = 0;
    emit_side_effects_ = false;
    ns_rows(m->ns);
    bool ok = !emit_side_effects_ && ctrs_.size() == c0 && ctr_touched_.size() == t0;
    for (size_t i = s0; i < stage_.size() && ok; i++) if (stage_[i].row.rev & ~ROW_REV_MASK) ok = false;
    if (ok) { m->rows.assign(stage_.begin() + s0, stage_.end()); m->flags = review_flags_; m->facts = facts_acc_; m->owner = out; m->rows_state = 1; }
    else m->rows_state = 2;
    review_flags_ |= f0;
    facts_acc_ |= a0;
  }
  finish_tail(source, out);
}
void Flattener::ns_rows(const Value& ns) {
  if (ns.defined()) {
    review_flags_ |= RF_NS_PRESENT;
    emit(id_ns_, T_OBJECT, 1, 0);
    uint32_t md = child(id_ns_, "metadata");
    emit(md, T_OBJECT, 2, 0);
    emit_str(md, "name", obj_string(ns, "metadata", "name"));
    const Value* m = ns.get("metadata");
    const Value* lb = m ? m->get("labels") : nullptr;
    if (lb && lb->is_object()) {
      bool bad = false;
      for (const auto& kv : lb->pairs()) if (!kv.second.is_string()) bad = true;
      if (bad) review_flags_ |= RF_NS_LABELS_BAD;
      walk(*lb, child(md, "labels"), 0, 0, 0);
    } else if (lb) review_flags_ |= RF_NS_LABELS_BAD;
  }
}
void Flattener::finish_tail(int source, HostTable* out) {
  switch (source) {
    case SRC_ORIGINAL: review_flags_ |= RF_SRC_ORIGINAL; break;
    case SRC_GENERATED: review_flags_ |= RF_SRC_GENERATED; break;
    case SRC_ALL: review_flags_ |= RF_SRC_ALL; break;
    case SRC_INVALID: review_flags_ |= RF_SRC_INVALID; break;
    default: break;
  }
  if (dup_seen_) {   // (round 4) two message keys of this review are equal: the counting plans leave it to the renderer
    if (!id_dup_) id_dup_ = child(0, "$dup");
    emit(id_dup_, T_BOOL, 1, 0, true);
  }
  if (facts_acc_) {   // (round 6) the review facts row: the answers of the dictionary expressions on the review's non-iterated leaves
    if (!id_facts_d_) id_facts_d_ = child(child(0, "$r"), "$d");
    emit(id_facts_d_, T_INT, (uint32_t)facts_acc_, (uint32_t)(facts_acc_ >> 32), true);
  }
  out->rflags.push_back(review_flags_);
  for (const Ctr& c : ctrs_) {
    if (c.path >= out->path_max.size()) out->path_max.resize(c.path + 1, 0);
    out->path_max[c.path] = std::max(out->path_max[c.path], c.n);
  }
  for (uint32_t ep : ctr_touched_) {
    if (ep >= out->path_max.size()) out->path_max.resize(ep + 1, 0);
    out->path_max[ep] = std::max(out->path_max[ep], ctr_val_[ep]);
  }
  out->n_reviews++;
  if (out->n_reviews % out->rpt == 0) flush_tile(out);
}

// Close the current tile: stable sort of its rows by path (keeps review order, then document order, inside a
// segment) and one segment record per distinct path (turned into the slot index by build_index).
void Flattener::flush_tile(HostTable* out) {
  out->tile_seg.push_back((uint32_t)out->segs.size());
  // counting sort by path (stable): paths are dense ids, a tile touches a few hundred of them
  const uint32_t n = (uint32_t)stage_.size();
  uint32_t max_path = 0;
  for (const Staged& s : stage_) max_path = std::max(max_path, s.path);
  if (sort_count_.size() <= max_path) sort_count_.resize(max_path + 1, 0);
  sort_paths_.clear();
  for (const Staged& s : stage_) if (sort_count_[s.path]++ == 0) sort_paths_.push_back(s.path);
  std::sort(sort_paths_.begin(), sort_paths_.end());
  const size_t base = out->rows.size();
  uint32_t at = 0;
  for (uint32_t p : sort_paths_) {
    out->segs.push_back({p, (uint32_t)(base + at)});
    if (p >= out->path_rows.size()) out->path_rows.resize(p + 1, 0);
    out->path_rows[p] += sort_count_[p];
    const uint32_t c = sort_count_[p];
    sort_count_[p] = at;   // becomes the write cursor of the path
    at += c;
  }
  out->rows.resize(base + n);
  out->shdr.resize(base + n);
  for (const Staged& s : stage_) {
    const uint32_t k = sort_count_[s.path]++;
    out->rows[base + k] = s.row;
    out->shdr[base + k] = s.hdr;
  }
  for (uint32_t p : sort_paths_) sort_count_[p] = 0;
  stage_.clear();
}

void Flattener::flush(HostTable* out) {
  if (!stage_.empty() || out->n_reviews % out->rpt != 0) flush_tile(out);
  out->n_rows_total = out->rows.size();
  out->heap_total = out->heap.size();
}

// Appends `part` (whole tiles flattened by another Flattener over the same dictionary; the receiving table must end
// on a tile boundary) -- row starts and heap offsets are relocated.
void HostTable::append(const HostTable& part) {
  const uint32_t row_base = (uint32_t)rows.size(), heap_base = (uint32_t)heap.size(), seg_base = (uint32_t)segs.size();
  rows.append(part.rows.data(), part.rows.size());
  for (size_t i = row_base; i < rows.size(); i++)
    if ((rows[i].meta & ROW_TYPE_MASK) == T_STRING && !(rows[i].meta & ROW_STR_INLINE)) rows[i].lo += heap_base;
  shdr.append(part.shdr.data(), part.shdr.size());
  heap.append(part.heap.data(), part.heap.size());
  for (const SegRec& s : part.segs) segs.push_back({s.path, s.start + row_base});
  for (uint32_t ts : part.tile_seg) tile_seg.push_back(ts + seg_base);
  rflags.insert(rflags.end(), part.rflags.begin(), part.rflags.end());
  if (part.path_rows.size() > path_rows.size()) path_rows.resize(part.path_rows.size(), 0);
  for (size_t i = 0; i < part.path_rows.size(); i++) path_rows[i] += part.path_rows[i];
  if (part.path_max.size() > path_max.size()) path_max.resize(part.path_max.size(), 0);
  for (size_t i = 0; i < part.path_max.size(); i++) path_max[i] = std::max(path_max[i], part.path_max[i]);
  n_reviews += part.n_reviews;
  n_rows_total = rows.size(); heap_total = heap.size();
}

// ------------------------------------------------------------------------------------------------ fast ingest
// JSON text -> rows in one pass (see flatten.hpp).  Grammar and value semantics are those of value.hpp's JsonParser;
// anything unusual bails out (returns false / -1) so that the general path decides.
// the first eight bytes of a member name (zero-padded): one load when eight bytes can be read at the name (it lies inside the
// document text, which goes on after the closing quote), else a copy
inline uint64_t Flattener::name8(const char* key, uint32_t len) const {
  uint64_t w = 0;
  if (len >= 8) { memcpy(&w, key, 8); return w; }
  if (key >= ix_json_ && key + 8 <= ix_json_ + ix_len_) { memcpy(&w, key, 8); return len ? (w & (~0ull >> (64 - 8 * len))) : 0; }
  memcpy(&w, key, len);
  return w;
}
uint32_t Flattener::fast_child_at(uint32_t parent, uint32_t pos, const char* key, uint32_t len) {
  const uint64_t w8 = name8(key, len);
  if (parent < pred_.size() && pos < pred_[parent].size()) {
    const PredEnt& pe = pred_[parent][pos];
    if (pe.len == len && pe.first8 == w8 && pe.id != 0xFFFFFFFFu && (len <= 8 || memcmp(key_arena_.data() + pe.off + 8, key + 8, len - 8) == 0)) return pe.id;
  }
  const uint32_t id = fast_child_w(parent, key, len, w8);
  if (pos < 64) {   // (positions beyond 64 -- label maps, annotation maps -- are not worth remembering: their names vary)
    if (parent >= pred_.size()) pred_.resize((size_t)parent * 2 + 64);
    std::vector<PredEnt>& v = pred_[parent];
    if (pos >= v.size()) v.resize(pos + 1);
    // the name's bytes live in key_arena_ (fast_child interned them): fast_child left the slot it used in last_slot_
    v[pos].id = id; v[pos].off = key_tab_[last_slot_].off; v[pos].len = len; v[pos].first8 = w8;
  }
  return id;
}

uint32_t Flattener::fast_child(uint32_t parent, const char* key, uint32_t len) {
  uint64_t w = 0;
  memcpy(&w, key, len < 8 ? len : 8);
  return fast_child_w(parent, key, len, w);
}
uint32_t Flattener::fast_child_w(uint32_t parent, const char* key, uint32_t len, uint64_t w8) {
  uint64_t h = 1469598103934665603ull ^ ((uint64_t)parent * 0x9E3779B97F4A7C15ull);
  h = (h ^ w8) * 0x9E3779B97F4A7C15ull; h ^= h >> 32;
  if (len > 8) {   // eight bytes of the name per round, the tail byte by byte
    uint32_t i = 8;
    for (; i + 8 <= len; i += 8) { uint64_t w; memcpy(&w, key + i, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 32; }
    uint64_t w = 0;
    memcpy(&w, key + i, len - i);
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
  }
  h = (h ^ ((uint64_t)len << 56)) * 1099511628211ull;
  h ^= h >> 29;
  if (key_tab_.empty()) key_tab_.resize(2048);
  size_t mask = key_tab_.size() - 1, i = (size_t)h & mask;
  while (key_tab_[i].used) {
    const KeySlot& k = key_tab_[i];
    if (k.hash == h && k.first8 == w8 && k.parent == parent && k.len == len && (len <= 8 || memcmp(key_arena_.data() + k.off + 8, key + 8, len - 8) == 0)) { last_slot_ = i; return k.id; }
    i = (i + 1) & mask;
  }
  const uint32_t id = dict_->child(parent, std::string(key, len));
  if ((key_count_ + 1) * 2 > key_tab_.size()) {   // grow + rehash
    std::vector<KeySlot> old;
    old.swap(key_tab_);
    key_tab_.resize(old.size() * 2);
    mask = key_tab_.size() - 1;
    for (const KeySlot& k : old) if (k.used) { size_t j = (size_t)k.hash & mask; while (key_tab_[j].used) j = (j + 1) & mask; key_tab_[j] = k; }
    i = (size_t)h & mask;
    while (key_tab_[i].used) i = (i + 1) & mask;
  }
  KeySlot& k = key_tab_[i];
  k.used = true; k.hash = h; k.first8 = w8; k.parent = parent; k.id = id; k.off = (uint32_t)key_arena_.size(); k.len = len;
  key_arena_.append(key, len);
  key_count_++;
  last_slot_ = i;
  return id;
}

// p_ is at the opening quote.  On success *s/*n view the decoded bytes: the JSON text itself when the string has no
// escapes, else scratch_ (valid until the next decoded string).
bool Flattener::fast_string(const char** s, uint32_t* n) {
  const char* q = p_ + 1;
  const char* start = q;
  while (q < e_ && *q != '"' && *q != '\\') q++;
  if (q >= e_) return false;
  if (*q == '"') { *s = start; *n = (uint32_t)(q - start); p_ = q + 1; return true; }
  // escapes: decode (same rules as JsonParser::str)
  scratch_.assign(start, q - start);
  p_ = q;
  auto hex4 = [&](uint32_t* v) -> bool {
    if (e_ - p_ < 4) return false;
    uint32_t x = 0;
    for (int k = 0; k < 4; k++) {
      char c = *p_++;
      x <<= 4;
      if (c >= '0' && c <= '9') x |= c - '0';
      else if (c >= 'a' && c <= 'f') x |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') x |= c - 'A' + 10;
      else return false;
    }
    *v = x;
    return true;
  };
  auto utf8 = [&](uint32_t cp) {
    if (cp < 0x80) scratch_.push_back((char)cp);
    else if (cp < 0x800) { scratch_.push_back((char)(0xC0 | (cp >> 6))); scratch_.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { scratch_.push_back((char)(0xE0 | (cp >> 12))); scratch_.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); scratch_.push_back((char)(0x80 | (cp & 0x3F))); }
    else { scratch_.push_back((char)(0xF0 | (cp >> 18))); scratch_.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); scratch_.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); scratch_.push_back((char)(0x80 | (cp & 0x3F))); }
  };
  for (;;) {
    if (p_ >= e_) return false;
    const char* r = p_;
    while (r < e_ && *r != '"' && *r != '\\') r++;
    scratch_.append(p_, r - p_);
    p_ = r;
    if (p_ >= e_) return false;
    if (*p_ == '"') { p_++; *s = scratch_.data(); *n = (uint32_t)scratch_.size(); return true; }
    p_++;
    if (p_ >= e_) return false;
    char c = *p_++;
    switch (c) {
      case '"': scratch_.push_back('"'); break;
      case '\\': scratch_.push_back('\\'); break;
      case '/': scratch_.push_back('/'); break;
      case 'b': scratch_.push_back('\b'); break;
      case 'f': scratch_.push_back('\f'); break;
      case 'n': scratch_.push_back('\n'); break;
      case 'r': scratch_.push_back('\r'); break;
      case 't': scratch_.push_back('\t'); break;
      case 'u': {
        uint32_t cp;
        if (!hex4(&cp)) return false;
        if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
          p_ += 2;
          uint32_t lo;
          if (!hex4(&lo)) return false;
          if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          else { utf8(0xFFFD); cp = lo; }
        }
        utf8(cp);
        break;
      }
      default: return false;
    }
  }
}

void Flattener::emit_str_n(uint32_t path, uint32_t meta, const char* s, uint32_t n) {
  if (!(pbits(path) & (PB_ROW | PB_KEY))) return;
  if (n <= 7) {   // inline: no heap entry, no memory access on the device
    uint64_t bits = 0;
    memcpy(&bits, s, n);
    emit(path, meta | T_STRING | ROW_STR_INLINE, (uint32_t)bits, (uint32_t)(bits >> 32) | (n << 24));
    return;
  }
  PodVec<uint8_t>& h = t_->heap;
  size_t at = (h.size() + 15) & ~(size_t)15;
  const size_t old_n = h.size(), end = at + ((4 + (size_t)n + 15) & ~(size_t)15);
  h.resize(end);
  memset(&h[old_n], 0, at - old_n);
  memset(&h[end - 16], 0, 16);     // zero padding: the device compares whole words
  memcpy(&h[at], &n, 4);
  memcpy(&h[at + 4], s, n);
  if (emit(path, meta | T_STRING, (uint32_t)(at + 4), hash32((const uint8_t*)s, n))) memcpy(&stage_.back().hdr, &h[at], 16);   // entry header: length + first 12 bytes
}

// one JSON value at p_ -> rows under `path`; returns its RowType, -1 to bail out
int Flattener::fast_value(uint32_t path, uint32_t ords, int adepth, uint32_t extra, int depth) {
  if (depth > 96) return -1;
  ws();
  if (p_ >= e_) return -1;
  const uint32_t meta = ords | extra;
  const char c = *p_;
  if (c == '{') {
    const char* const span0 = p_;   // (a deep dictionary expression wants the container's text: dexpr.hpp)
    p_++;
    const size_t row = stage_.size();
    const bool has_row = emit(path, meta | T_OBJECT, 0, 0);
    const uint32_t inst = ++obj_instance_;
    uint32_t count = 0;
    ws();
    if (p_ < e_ && *p_ == '}') { p_++; if (dict_wanted(path)) dict_row(path, meta, Value::object({})); return T_OBJECT; }
    const CapIds* cap = nullptr;
    if (cur_facts_ && depth <= 1) cap = &cap_[cur_root_ == id_old_ ? 1 : 0];
    for (;;) {
      ws();
      if (p_ >= e_ || *p_ != '"') return -1;
      const char* k; uint32_t kn;
      if (!fast_string(&k, &kn)) return -1;
      const uint32_t ch = fast_child_at(path, count, k, kn);
      if (ch >= dup_gen_.size()) dup_gen_.resize((size_t)ch * 2 + 64, 0);
      if (dup_gen_[ch] == inst) return -1;   // duplicate member name: the general path applies "last one wins"
      dup_gen_[ch] = inst;
      ws();
      if (p_ >= e_ || *p_ != ':') return -1;
      p_++;
      // strings the match layer needs from the candidate object (unstructured accessors: apiVersion, kind,
      // metadata.{name,namespace,generateName}; GetLabels() needs metadata.labels to be a map of strings)
      Captured* want = nullptr;
      if (cap) {
        if (depth == 0) { if (ch == cap->api_version) want = &cur_facts_->api_version; else if (ch == cap->kind) want = &cur_facts_->kind; }
        else if (path == cap->metadata) { if (ch == cap->name) want = &cur_facts_->name; else if (ch == cap->ns) want = &cur_facts_->ns; else if (ch == cap->gname) want = &cur_facts_->gname; }
      }
      int t;
      // GK_TABLE_PRUNED: no pattern of any kind reaches this member or anything below it -- validated and walked past.  (metadata and
      // its members are always visited: the match layer's facts are captured there; metadata.labels needs the TYPE of
      // every value, which the skipper reports.)
      if (pruning_ && !want && !(cap && ((depth == 0 && ch == cap->metadata) || (depth == 1 && path == cap->metadata))) && !(read_state(ch) & 2u)) {
        t = skip_value(depth + 1);
      } else if (want) {
        ws();
        if (p_ < e_ && *p_ == '"') {
          const char* v; uint32_t vn;
          if (!fast_string(&v, &vn)) return -1;
          emit_str_n(ch, meta, v, vn);
          if (dict_wanted(ch)) dict_row(ch, meta, Value::string(std::string(v, vn)));
          if (v == scratch_.data()) { scratch_keep_.emplace_back(new std::string(v, vn)); v = scratch_keep_.back()->data(); }
          want->p = v; want->n = vn; want->set = true;
          t = T_STRING;
        } else t = fast_value(ch, ords, adepth, extra, depth + 1);
      } else t = fast_value(ch, ords, adepth, extra, depth + 1);
      if (t < 0) return -1;
      if (cap && depth == 1 && path == cap->metadata && ch == cap->labels && t != T_OBJECT) cur_facts_->labels_bad = true;
      if (cur_facts_ && depth == 2 && path == cap_[cur_root_ == id_old_ ? 1 : 0].labels && t != T_STRING) cur_facts_->labels_bad = true;
      count++;
      ws();
      if (p_ < e_ && *p_ == ',') { p_++; continue; }
      if (p_ < e_ && *p_ == '}') { p_++; break; }
      return -1;
    }
    if (has_row) {
      stage_[row].row.lo = count;
      if (count) { if (stage_[row].row.rev & ~ROW_REV_MASK) review_flags_ |= RF_HOST_CAND; stage_[row].row.rev &= ROW_REV_MASK; }   // a NON-EMPTY container has no value id (emit saw it empty)
    }
    if (count && guard_wanted(path)) review_flags_ |= RF_REFUSE;
    if (dict_deep(path)) dict_row(path, meta, parse_json(span0, (size_t)(p_ - span0)));
    else if (dict_wanted(path)) { ValuePairs ph; for (uint32_t k = 0; k < count; k++) ph.emplace_back(Value::integer((i128)k), Value::null()); dict_row(path, meta, Value::object(std::move(ph))); }
    return T_OBJECT;
  }
  if (c == '[') {
    const char* const span0 = p_;
    p_++;
    const size_t row = stage_.size();
    const bool has_row = emit(path, meta | T_ARRAY, 0, 0);
    uint32_t count = 0;
    ws();
    if (p_ < e_ && *p_ == ']') { p_++; if (dict_wanted(path)) dict_row(path, meta, Value::array({})); return T_ARRAY; }
    const uint32_t ep = elem(path);
    if (ep >= ctr_gen_.size()) { ctr_gen_.resize((size_t)ep * 2 + 64, 0); ctr_val_.resize(ctr_gen_.size(), 0); }
    if (ctr_gen_[ep] != review_gen_) { ctr_gen_[ep] = review_gen_; ctr_val_[ep] = 0; ctr_touched_.push_back(ep); }
    for (;;) {
      uint32_t ord = ctr_val_[ep]++;
      uint32_t ex = extra, o2 = ords;
      if (adepth < 3) {
        if (ord >= 255) { ord = 255; ex |= ROW_ORD_OVERFLOW; review_flags_ |= RF_TOO_BIG; }
        o2 |= ord << (ROW_E_SHIFT0 + 8 * adepth);
      } else ex |= ROW_DEEP;
      if (pruning_ && !(read_state(ep) & 2u)) { if (skip_value(depth + 1) < 0) return -1; }
      else {
        // element carrier (plan.hpp T_ABSENT): an object that held the carrier member marked its path with an instance number drawn
        // while this element was parsed (dup_gen_); anything else gets the row that exists for the element marker alone
        const uint32_t cp = carrier_child(ep), inst0 = obj_instance_;
        if (fast_value(ep, o2, adepth + 1, ex, depth + 1) < 0) return -1;
        if (cp && !(cp < dup_gen_.size() && dup_gen_[cp] > inst0)) emit_absent(cp, o2 | ex);
      }
      count++;
      ws();
      if (p_ < e_ && *p_ == ',') { p_++; continue; }
      if (p_ < e_ && *p_ == ']') { p_++; break; }
      return -1;
    }
    if (has_row) {
      stage_[row].row.lo = count;
      if (count) { if (stage_[row].row.rev & ~ROW_REV_MASK) review_flags_ |= RF_HOST_CAND; stage_[row].row.rev &= ROW_REV_MASK; }   // a NON-EMPTY container has no value id (emit saw it empty)
    }
    if (dict_deep(path)) dict_row(path, meta, parse_json(span0, (size_t)(p_ - span0)));
    else if (dict_wanted(path)) dict_row(path, meta, Value::array(ValueVec(count, Value::null())));
    return T_ARRAY;
  }
  if (c == '"') {
    const char* v; uint32_t vn;
    if (!fast_string(&v, &vn)) return -1;
    emit_str_n(path, meta, v, vn);
    if (dict_wanted(path)) dict_row(path, meta, Value::string(std::string(v, vn)));
    return T_STRING;
  }
  if (c == 't') { if (e_ - p_ < 4 || memcmp(p_, "true", 4) != 0) return -1; p_ += 4; emit(path, meta | T_BOOL, 1, 0); if (dict_wanted(path)) dict_row(path, meta, Value::boolean(true)); return T_BOOL; }
  if (c == 'f') { if (e_ - p_ < 5 || memcmp(p_, "false", 5) != 0) return -1; p_ += 5; emit(path, meta | T_BOOL, 0, 0); if (dict_wanted(path)) dict_row(path, meta, Value::boolean(false)); return T_BOOL; }
  if (c == 'n') { if (e_ - p_ < 4 || memcmp(p_, "null", 4) != 0) return -1; p_ += 4; emit(path, meta | T_NULL, 0, 0); if (dict_wanted(path)) dict_row(path, meta, Value::null()); return T_NULL; }
  // number: JsonParser::number + Flattener::walk
  const char* s = p_;
  bool is_int = true, neg = false;
  if (p_ < e_ && *p_ == '-') { neg = true; p_++; }
  if (p_ >= e_ || !(*p_ >= '0' && *p_ <= '9')) return -1;
  const char* digits = p_;
  while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++;
  const size_t nd = p_ - digits;
  if (p_ < e_ && *p_ == '.') { is_int = false; p_++; while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++; }
  if (p_ < e_ && (*p_ == 'e' || *p_ == 'E')) {
    is_int = false; p_++;
    if (p_ < e_ && (*p_ == '+' || *p_ == '-')) p_++;
    while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++;
  }
  if (is_int && nd <= 18) {   // fits int64 for certain
    int64_t x = 0;
    for (size_t k = 0; k < nd; k++) x = x * 10 + (digits[k] - '0');
    if (neg) x = -x;
    emit(path, meta | T_INT, (uint32_t)(uint64_t)x, (uint32_t)((uint64_t)x >> 32));
    if (dict_wanted(path)) dict_row(path, meta, Value::integer((i128)x));
    return T_INT;
  }
  {   // the general number rules, through the same Value code the tree walk uses
    Value v = parse_json(s, p_ - s);
    if (dict_wanted(path)) dict_row(path, meta, v);
    if (v.is_int && v.i >= (i128)INT64_MIN && v.i <= (i128)INT64_MAX) {
      uint64_t u = (uint64_t)(int64_t)v.i;
      emit(path, meta | T_INT, (uint32_t)u, (uint32_t)(u >> 32));
      return T_INT;
    }
    double d = v.as_double();
    uint64_t u;
    memcpy(&u, &d, 8);
    emit(path, meta | T_FLOAT | (v.is_int ? ROW_INEXACT : 0), (uint32_t)u, (uint32_t)(u >> 32));
    return T_FLOAT;
  }
}

// p_ at a value nothing reads: the same syntax checks as fast_value (a malformed value must still send the review down the general
// path, which rejects it the way the reference's decoder does), no path lookups, no rows
int Flattener::skip_value(int depth) {
  if (depth > 96) return -1;
  ws();
  if (p_ >= e_) return -1;
  const char c = *p_;
  if (c == '{') {
    p_++;
    ws();
    if (p_ < e_ && *p_ == '}') { p_++; return T_OBJECT; }
    for (;;) {
      ws();
      if (p_ >= e_ || *p_ != '"') return -1;
      const char* k; uint32_t kn;
      if (!fast_string(&k, &kn)) return -1;
      ws();
      if (p_ >= e_ || *p_ != ':') return -1;
      p_++;
      if (skip_value(depth + 1) < 0) return -1;
      ws();
      if (p_ < e_ && *p_ == ',') { p_++; continue; }
      if (p_ < e_ && *p_ == '}') { p_++; return T_OBJECT; }
      return -1;
    }
  }
  if (c == '[') {
    p_++;
    ws();
    if (p_ < e_ && *p_ == ']') { p_++; return T_ARRAY; }
    for (;;) {
      if (skip_value(depth + 1) < 0) return -1;
      ws();
      if (p_ < e_ && *p_ == ',') { p_++; continue; }
      if (p_ < e_ && *p_ == ']') { p_++; return T_ARRAY; }
      return -1;
    }
  }
  if (c == '"') { const char* v; uint32_t vn; return fast_string(&v, &vn) ? (int)T_STRING : -1; }
  if (c == 't') { if (e_ - p_ < 4 || memcmp(p_, "true", 4) != 0) return -1; p_ += 4; return T_BOOL; }
  if (c == 'f') { if (e_ - p_ < 5 || memcmp(p_, "false", 5) != 0) return -1; p_ += 5; return T_BOOL; }
  if (c == 'n') { if (e_ - p_ < 4 || memcmp(p_, "null", 4) != 0) return -1; p_ += 4; return T_NULL; }
  const char* s0 = p_;
  bool is_int = true;
  if (p_ < e_ && *p_ == '-') p_++;
  if (p_ >= e_ || !(*p_ >= '0' && *p_ <= '9')) return -1;
  const char* digits = p_;
  while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++;
  const size_t nd = p_ - digits;
  if (p_ < e_ && *p_ == '.') { is_int = false; p_++; while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++; }
  if (p_ < e_ && (*p_ == 'e' || *p_ == 'E')) {
    is_int = false; p_++;
    if (p_ < e_ && (*p_ == '+' || *p_ == '-')) p_++;
    while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++;
  }
  if (is_int && nd <= 18) return T_INT;
  Value v = parse_json(s0, p_ - s0);   // (the general number rules decide, as in fast_value)
  return v.is_int && v.i >= (i128)INT64_MIN && v.i <= (i128)INT64_MAX ? (int)T_INT : (int)T_FLOAT;
}

// ------------------------------------------------------------------------------------------------ structural index
// Stage 1.  Per 64-byte block: B = backslashes, Q = quotes not escaped by an odd run of backslashes (the carry of a run crossing the
// block edge travels in `prev_escaped`), S = prefix-xor of Q (carry-less multiply by all-ones) = opening quote .. the byte before
// the closing quote, OP = { } [ ] : , outside strings, SC = any other byte that is neither white space nor a quote outside strings;
// tokens = OP | Q | first byte of every run of SC.
bool Flattener::ix_supported() {
  static const bool ok = __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("bmi2");
  return ok;
}
__attribute__((target("avx512f,avx512bw,pclmul,bmi,bmi2,lzcnt,popcnt")))
static uint32_t ix_stage1(const char* s, uint32_t len, uint32_t* out, uint64_t* bs_out, bool* any_bs) {
  uint64_t prev_in_string = 0, prev_escaped = 0, prev_scalar = 0, bs_or = 0;
  uint32_t n = 0;
  const __m512i c_bs = _mm512_set1_epi8('\\'), c_q = _mm512_set1_epi8('"'), c_20 = _mm512_set1_epi8(0x20), c_ob = _mm512_set1_epi8('{'), c_cb = _mm512_set1_epi8('}'),
                c_col = _mm512_set1_epi8(':'), c_com = _mm512_set1_epi8(','), c_nl = _mm512_set1_epi8('\n'), c_tab = _mm512_set1_epi8('\t'), c_cr = _mm512_set1_epi8('\r');
  for (uint32_t base = 0; base < len; base += 64) {
    const uint32_t rem = len - base;
    const uint64_t valid = rem >= 64 ? ~0ull : (~0ull >> (64 - rem));
    const __m512i v = rem >= 64 ? _mm512_loadu_si512((const void*)(s + base)) : _mm512_maskz_loadu_epi8((__mmask64)valid, (const void*)(s + base));
    uint64_t bs = _mm512_cmpeq_epi8_mask(v, c_bs);
    uint64_t quote = _mm512_cmpeq_epi8_mask(v, c_q);
    bs_out[base >> 6] = bs;
    bs_or |= bs;
    if (bs | prev_escaped) {   // bytes escaped by a backslash: the byte after every ODD-length run's last backslash
      bs &= ~prev_escaped;
      const uint64_t follows = (bs << 1) | prev_escaped;
      const uint64_t even = 0x5555555555555555ull;
      const uint64_t odd_starts = bs & ~even & ~follows;
      uint64_t on_even;
      prev_escaped = __builtin_add_overflow(odd_starts, bs, &on_even) ? 1 : 0;
      const uint64_t escaped = (even ^ (on_even << 1)) & follows;
      quote &= ~escaped;
    }
    const uint64_t in_string = (uint64_t)_mm_cvtsi128_si64(_mm_clmulepi64_si128(_mm_set_epi64x(0, (long long)quote), _mm_set1_epi8((char)0xFF), 0)) ^ prev_in_string;
    prev_in_string = (uint64_t)((int64_t)in_string >> 63);
    const __m512i lower = _mm512_or_si512(v, c_20);   // '[' | 0x20 = '{', ']' | 0x20 = '}'
    const uint64_t op = _mm512_cmpeq_epi8_mask(lower, c_ob) | _mm512_cmpeq_epi8_mask(lower, c_cb) | _mm512_cmpeq_epi8_mask(v, c_col) | _mm512_cmpeq_epi8_mask(v, c_com);
    const uint64_t wsm = _mm512_cmpeq_epi8_mask(v, c_20) | _mm512_cmpeq_epi8_mask(v, c_nl) | _mm512_cmpeq_epi8_mask(v, c_tab) | _mm512_cmpeq_epi8_mask(v, c_cr);
    const uint64_t scalar = ~(op | wsm | quote) & ~in_string & valid;
    const uint64_t scalar_start = scalar & ~((scalar << 1) | prev_scalar);
    prev_scalar = scalar >> 63;
    uint64_t bits = ((op & ~in_string) | quote | scalar_start) & valid;
    while (bits) { out[n++] = base + (uint32_t)_tzcnt_u64(bits); bits = _blsr_u64(bits); }
  }
  *any_bs = bs_or != 0;
  return n;
}
__attribute__((target("avx512f,avx512bw,avx512vbmi,avx512vbmi2,pclmul,bmi,bmi2,lzcnt,popcnt")))
static uint32_t ix_stage1_compress(const char* s, uint32_t len, uint32_t* out, uint64_t* bs_out, bool* any_bs) {
  uint64_t prev_in_string = 0, prev_escaped = 0, prev_scalar = 0, bs_or = 0;
  uint32_t n = 0;
  const __m512i c_bs = _mm512_set1_epi8('\\'), c_q = _mm512_set1_epi8('"'), c_20 = _mm512_set1_epi8(0x20), c_ob = _mm512_set1_epi8('{'), c_cb = _mm512_set1_epi8('}'),
                c_col = _mm512_set1_epi8(':'), c_com = _mm512_set1_epi8(','), c_nl = _mm512_set1_epi8('\n'), c_tab = _mm512_set1_epi8('\t'), c_cr = _mm512_set1_epi8('\r');
  alignas(64) static const uint8_t iota_b[64] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
                                                 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63};
  const __m512i iota = _mm512_load_si512((const void*)iota_b);
  for (uint32_t base = 0; base < len; base += 64) {
    const uint32_t rem = len - base;
    const uint64_t valid = rem >= 64 ? ~0ull : (~0ull >> (64 - rem));
    const __m512i v = rem >= 64 ? _mm512_loadu_si512((const void*)(s + base)) : _mm512_maskz_loadu_epi8((__mmask64)valid, (const void*)(s + base));
    uint64_t bs = _mm512_cmpeq_epi8_mask(v, c_bs);
    uint64_t quote = _mm512_cmpeq_epi8_mask(v, c_q);
    bs_out[base >> 6] = bs;
    bs_or |= bs;
    if (bs | prev_escaped) {   // bytes escaped by a backslash: the byte after every ODD-length run's last backslash
      bs &= ~prev_escaped;
      const uint64_t follows = (bs << 1) | prev_escaped;
      const uint64_t even = 0x5555555555555555ull;
      const uint64_t odd_starts = bs & ~even & ~follows;
      uint64_t on_even;
      prev_escaped = __builtin_add_overflow(odd_starts, bs, &on_even) ? 1 : 0;
      const uint64_t escaped = (even ^ (on_even << 1)) & follows;
      quote &= ~escaped;
    }
    const uint64_t in_string = (uint64_t)_mm_cvtsi128_si64(_mm_clmulepi64_si128(_mm_set_epi64x(0, (long long)quote), _mm_set1_epi8((char)0xFF), 0)) ^ prev_in_string;
    prev_in_string = (uint64_t)((int64_t)in_string >> 63);
    const __m512i lower = _mm512_or_si512(v, c_20);   // '[' | 0x20 = '{', ']' | 0x20 = '}'
    const uint64_t op = _mm512_cmpeq_epi8_mask(lower, c_ob) | _mm512_cmpeq_epi8_mask(lower, c_cb) | _mm512_cmpeq_epi8_mask(v, c_col) | _mm512_cmpeq_epi8_mask(v, c_com);
    const uint64_t wsm = _mm512_cmpeq_epi8_mask(v, c_20) | _mm512_cmpeq_epi8_mask(v, c_nl) | _mm512_cmpeq_epi8_mask(v, c_tab) | _mm512_cmpeq_epi8_mask(v, c_cr);
    const uint64_t scalar = ~(op | wsm | quote) & ~in_string & valid;
    const uint64_t scalar_start = scalar & ~((scalar << 1) | prev_scalar);
    prev_scalar = scalar >> 63;
    const uint64_t bits = ((op & ~in_string) | quote | scalar_start) & valid;
    // positions of the set bits: VPCOMPRESSB packs the byte numbers 0..63 under the mask, widened 16 at a time (no loop over
    // bits, no data-dependent branch per token; the stores run up to 64 entries past n: ix_build leaves that room)
    const uint32_t cnt = (uint32_t)_mm_popcnt_u64(bits);
    const __m512i packed = _mm512_maskz_compress_epi8((__mmask64)bits, iota);
    const __m512i vb = _mm512_set1_epi32((int)base);
    uint32_t* o = out + n;
    _mm512_storeu_si512((void*)o, _mm512_add_epi32(vb, _mm512_cvtepu8_epi32(_mm512_castsi512_si128(packed))));
    if (cnt > 16) {
      _mm512_storeu_si512((void*)(o + 16), _mm512_add_epi32(vb, _mm512_cvtepu8_epi32(_mm512_extracti32x4_epi32(packed, 1))));
      if (cnt > 32) {
        _mm512_storeu_si512((void*)(o + 32), _mm512_add_epi32(vb, _mm512_cvtepu8_epi32(_mm512_extracti32x4_epi32(packed, 2))));
        if (cnt > 48) _mm512_storeu_si512((void*)(o + 48), _mm512_add_epi32(vb, _mm512_cvtepu8_epi32(_mm512_extracti32x4_epi32(packed, 3))));
      }
    }
    n += cnt;
  }
  *any_bs = bs_or != 0;
  return n;
}
void Flattener::ix_build(const char* json, size_t len) {
  if (ix_.size() < len + 80) ix_.resize(len + 80 + len / 2);
  if (ix_bs_.size() < len / 64 + 2) ix_bs_.resize(len / 64 + 2 + len / 128);
  ix_json_ = json; ix_len_ = (uint32_t)len;
  static const bool compress = __builtin_cpu_supports("avx512vbmi2") && __builtin_cpu_supports("avx512vbmi") && !getenv("GK_NO_VBMI2");
  ix_n_ = compress ? ix_stage1_compress(json, (uint32_t)len, ix_.data(), ix_bs_.data(), &ix_any_bs_) : ix_stage1(json, (uint32_t)len, ix_.data(), ix_bs_.data(), &ix_any_bs_);
  ix_[ix_n_] = (uint32_t)len;   // sentinel: where the last token's text ends at the latest
  ixp_ = 0;
  p_ = json; e_ = json + len;   // (strings with escapes are decoded by fast_string)
}
bool Flattener::ix_has_bs(uint32_t a, uint32_t b) const {
  if (!ix_any_bs_ || a >= b) return false;
  const uint32_t wa = a >> 6, wb = (b - 1) >> 6;
  const uint64_t ma = ~0ull << (a & 63), mb = ~0ull >> (63 - ((b - 1) & 63));
  if (wa == wb) return (ix_bs_[wa] & ma & mb) != 0;
  if (ix_bs_[wa] & ma) return true;
  for (uint32_t w = wa + 1; w < wb; w++) if (ix_bs_[w]) return true;
  return (ix_bs_[wb] & mb) != 0;
}
// token ixp_ is a quote that opens a string (the caller looked): the next token is the quote that closes it, or the text ends inside
bool Flattener::ix_string(const char** s, uint32_t* n) {
  if (ixp_ + 1 >= ix_n_) return false;
  const uint32_t a = ix_[ixp_] + 1, b = ix_[ixp_ + 1];
  if (ix_has_bs(a, b)) {
    p_ = ix_json_ + a - 1;
    if (!fast_string(s, n) || p_ != ix_json_ + b + 1) return false;
  } else { *s = ix_json_ + a; *n = b - a; }
  ixp_ += 2;
  return true;
}
bool Flattener::ix_skip_string() {
  if (ixp_ + 1 >= ix_n_) return false;
  const uint32_t a = ix_[ixp_] + 1, b = ix_[ixp_ + 1];
  if (ix_has_bs(a, b)) { const char* s; uint32_t n; p_ = ix_json_ + a - 1; if (!fast_string(&s, &n) || p_ != ix_json_ + b + 1) return false; }   // (the escapes must be valid ones)
  ixp_ += 2;
  return true;
}
// token ixp_ starts a scalar that is not a string.  Its text runs to the next token at most, and what follows it up to there is
// white space by construction (any other byte would have started a token) -- so "the grammar consumed the whole run" is one look
// at the byte after.
int Flattener::ix_scalar(bool rows, uint32_t path, uint32_t meta) {
  const char* s = ix_json_ + ix_[ixp_];
  const char* lim = ix_json_ + ix_[ixp_ + 1];
  auto ends = [&](const char* q) { return q == lim || *q == ' ' || *q == '\n' || *q == '\t' || *q == '\r'; };
  const char c = *s;
  if (c == 't') { if (lim - s < 4 || memcmp(s, "true", 4) != 0 || !ends(s + 4)) return -1; ixp_++; if (rows) { emit(path, meta | T_BOOL, 1, 0); if (pbits(path) & PB_DICT) dict_row(path, meta, Value::boolean(true)); } return T_BOOL; }
  if (c == 'f') { if (lim - s < 5 || memcmp(s, "false", 5) != 0 || !ends(s + 5)) return -1; ixp_++; if (rows) { emit(path, meta | T_BOOL, 0, 0); if (pbits(path) & PB_DICT) dict_row(path, meta, Value::boolean(false)); } return T_BOOL; }
  if (c == 'n') { if (lim - s < 4 || memcmp(s, "null", 4) != 0 || !ends(s + 4)) return -1; ixp_++; if (rows) { emit(path, meta | T_NULL, 0, 0); if (pbits(path) & PB_DICT) dict_row(path, meta, Value::null()); } return T_NULL; }
  const char* q = s;
  bool is_int = true, neg = false;
  if (q < lim && *q == '-') { neg = true; q++; }
  if (q >= lim || !(*q >= '0' && *q <= '9')) return -1;
  const char* digits = q;
  while (q < lim && *q >= '0' && *q <= '9') q++;
  const size_t nd = q - digits;
  if (q < lim && *q == '.') { is_int = false; q++; while (q < lim && *q >= '0' && *q <= '9') q++; }
  if (q < lim && (*q == 'e' || *q == 'E')) {
    is_int = false; q++;
    if (q < lim && (*q == '+' || *q == '-')) q++;
    while (q < lim && *q >= '0' && *q <= '9') q++;
  }
  if (!ends(q)) return -1;
  ixp_++;
  if (is_int && nd <= 18) {
    if (rows) {
      int64_t x = 0;
      for (size_t k = 0; k < nd; k++) x = x * 10 + (digits[k] - '0');
      if (neg) x = -x;
      emit(path, meta | T_INT, (uint32_t)(uint64_t)x, (uint32_t)((uint64_t)x >> 32));
      if (pbits(path) & PB_DICT) dict_row(path, meta, Value::integer((i128)x));
    }
    return T_INT;
  }
  Value v = parse_json(s, q - s);   // the general number rules, through the same Value code the tree walk uses
  const bool as_int = v.is_int && v.i >= (i128)INT64_MIN && v.i <= (i128)INT64_MAX;
  if (rows) {
    if (pbits(path) & PB_DICT) dict_row(path, meta, v);
    if (as_int) { const uint64_t u = (uint64_t)(int64_t)v.i; emit(path, meta | T_INT, (uint32_t)u, (uint32_t)(u >> 32)); }
    else { const double d = v.as_double(); uint64_t u; memcpy(&u, &d, 8); emit(path, meta | T_FLOAT | (v.is_int ? ROW_INEXACT : 0), (uint32_t)u, (uint32_t)(u >> 32)); }
  }
  return as_int ? (int)T_INT : (int)T_FLOAT;
}

// fast_value over tokens: the same rows, facts, flags and bail-outs, member for member
int Flattener::ix_value(uint32_t path, uint32_t ords, int adepth, uint32_t extra, int depth) {
  if (depth > 96 || ixp_ >= ix_n_) return -1;
  const uint32_t meta = ords | extra;
  const uint32_t at = ix_[ixp_];
  const char* const js = ix_json_;
  const char c = js[at];
  if (c == '{') {
    ixp_++;
    const size_t row = stage_.size();
    const bool has_row = emit(path, meta | T_OBJECT, 0, 0);
    const uint32_t inst = ++obj_instance_;
    uint32_t count = 0;
    if (ixp_ < ix_n_ && js[ix_[ixp_]] == '}') { ixp_++; if (pbits(path) & PB_DICT) dict_row(path, meta, Value::object({})); return T_OBJECT; }
    const CapIds* cap = nullptr;
    if (cur_facts_ && depth <= 1) cap = &cap_[cur_root_ == id_old_ ? 1 : 0];
    // (the object's own row and its dictionary row carry the member COUNT, which the general path takes after "last one wins" on
    //  duplicate names: with either kept, every name gets its id and duplicates are seen)
    const KidFilter* const kf = (pruning_ && !(cap && (depth == 0 || path == cap->metadata)) && !has_row && !(pbits(path) & PB_DICT)) ? kid_filter(path) : nullptr;
    for (;;) {
      if (ixp_ >= ix_n_ || js[ix_[ixp_]] != '"') return -1;
      const char* k; uint32_t kn;
      if (!ix_string(&k, &kn)) return -1;
      uint32_t ch;
      if (kf) {
        ch = PathDict::kNone;
        for (const KidEnt& ke : kf->kids) if (ke.len == kn && memcmp(kid_arena_.data() + ke.off, k, kn) == 0) { ch = ke.id; break; }
        if (ch == PathDict::kNone) {   // a member no pattern names: its value is checked and walked past
          if (ixp_ >= ix_n_ || js[ix_[ixp_]] != ':') return -1;
          ixp_++;
          const int ts = ix_skip(depth + 1);
          if (ts < 0) return -1;
          if (cur_facts_ && depth == 2 && path == cap_[cur_root_ == id_old_ ? 1 : 0].labels && ts != T_STRING) cur_facts_->labels_bad = true;   // (GetLabels() needs every value to be a string, read or not)
          count++;
          if (ixp_ >= ix_n_) return -1;
          const char d = js[ix_[ixp_]];
          if (d == ',') { ixp_++; continue; }
          if (d == '}') { ixp_++; break; }
          return -1;
        }
      } else ch = fast_child_at(path, count, k, kn);
      if (ch >= dup_gen_.size()) dup_gen_.resize((size_t)ch * 2 + 64, 0);
      if (dup_gen_[ch] == inst) return -1;   // duplicate member name: the general path applies "last one wins"
      dup_gen_[ch] = inst;
      if (ixp_ >= ix_n_ || js[ix_[ixp_]] != ':') return -1;
      ixp_++;
      if (ixp_ >= ix_n_) return -1;
      Captured* want = nullptr;
      if (cap) {
        if (depth == 0) { if (ch == cap->api_version) want = &cur_facts_->api_version; else if (ch == cap->kind) want = &cur_facts_->kind; }
        else if (path == cap->metadata) { if (ch == cap->name) want = &cur_facts_->name; else if (ch == cap->ns) want = &cur_facts_->ns; else if (ch == cap->gname) want = &cur_facts_->gname; }
      }
      int t;
      if (!want && !(pbits(ch) & PB_BELOW) && !(cap && ((depth == 0 && ch == cap->metadata) || (depth == 1 && path == cap->metadata)))) {
        t = ix_skip(depth + 1);
      } else if (want && js[ix_[ixp_]] == '"') {
        const char* v; uint32_t vn;
        if (!ix_string(&v, &vn)) return -1;
        emit_str_n(ch, meta, v, vn);
        if (pbits(ch) & PB_DICT) dict_row_str(ch, meta, v, vn);
        if (v == scratch_.data()) { scratch_keep_.emplace_back(new std::string(v, vn)); v = scratch_keep_.back()->data(); }
        want->p = v; want->n = vn; want->set = true;
        t = T_STRING;
      } else t = ix_value(ch, ords, adepth, extra, depth + 1);
      if (t < 0) return -1;
      if (cap && depth == 1 && path == cap->metadata && ch == cap->labels && t != T_OBJECT) cur_facts_->labels_bad = true;
      if (cur_facts_ && depth == 2 && path == cap_[cur_root_ == id_old_ ? 1 : 0].labels && t != T_STRING) cur_facts_->labels_bad = true;
      count++;
      if (ixp_ >= ix_n_) return -1;
      const char d = js[ix_[ixp_]];
      if (d == ',') { ixp_++; continue; }
      if (d == '}') { ixp_++; break; }
      return -1;
    }
    if (has_row) {
      stage_[row].row.lo = count;
      if (count) { if (stage_[row].row.rev & ~ROW_REV_MASK) review_flags_ |= RF_HOST_CAND; stage_[row].row.rev &= ROW_REV_MASK; }   // a NON-EMPTY container has no value id (emit saw it empty)
    }
    if (count && (pbits(path) & PB_GUARD)) review_flags_ |= RF_REFUSE;
    if (pbits(path) & PB_DEEP) dict_row(path, meta, parse_json(js + at, (size_t)(ix_[ixp_ - 1] + 1 - at)));
    else if (pbits(path) & PB_DICT) { ValuePairs ph; for (uint32_t k = 0; k < count; k++) ph.emplace_back(Value::integer((i128)k), Value::null()); dict_row(path, meta, Value::object(std::move(ph))); }
    return T_OBJECT;
  }
  if (c == '[') {
    ixp_++;
    const size_t row = stage_.size();
    const bool has_row = emit(path, meta | T_ARRAY, 0, 0);
    uint32_t count = 0;
    if (ixp_ < ix_n_ && js[ix_[ixp_]] == ']') { ixp_++; if (pbits(path) & PB_DICT) dict_row(path, meta, Value::array({})); return T_ARRAY; }
    const uint32_t ep = elem(path);
    if (ep >= ctr_gen_.size()) { ctr_gen_.resize((size_t)ep * 2 + 64, 0); ctr_val_.resize(ctr_gen_.size(), 0); }
    if (ctr_gen_[ep] != review_gen_) { ctr_gen_[ep] = review_gen_; ctr_val_[ep] = 0; ctr_touched_.push_back(ep); }
    const bool skip_elems = !(pbits(ep) & PB_BELOW);
    for (;;) {
      uint32_t ord = ctr_val_[ep]++;
      uint32_t ex = extra, o2 = ords;
      if (adepth < 3) {
        if (ord >= 255) { ord = 255; ex |= ROW_ORD_OVERFLOW; review_flags_ |= RF_TOO_BIG; }
        o2 |= ord << (ROW_E_SHIFT0 + 8 * adepth);
      } else ex |= ROW_DEEP;
      if (skip_elems) { if (ix_skip(depth + 1) < 0) return -1; }
      else {
        const uint32_t cp = carrier_child(ep), inst0 = obj_instance_;   // (element carrier: see fast_value)
        if (ix_value(ep, o2, adepth + 1, ex, depth + 1) < 0) return -1;
        if (cp && !(cp < dup_gen_.size() && dup_gen_[cp] > inst0)) emit_absent(cp, o2 | ex);
      }
      count++;
      if (ixp_ >= ix_n_) return -1;
      const char d = js[ix_[ixp_]];
      if (d == ',') { ixp_++; continue; }
      if (d == ']') { ixp_++; break; }
      return -1;
    }
    if (has_row) {
      stage_[row].row.lo = count;
      if (count) { if (stage_[row].row.rev & ~ROW_REV_MASK) review_flags_ |= RF_HOST_CAND; stage_[row].row.rev &= ROW_REV_MASK; }
    }
    if (pbits(path) & PB_DEEP) dict_row(path, meta, parse_json(js + at, (size_t)(ix_[ixp_ - 1] + 1 - at)));
    else if (pbits(path) & PB_DICT) dict_row(path, meta, Value::array(ValueVec(count, Value::null())));
    return T_ARRAY;
  }
  if (c == '"') {
    const char* v; uint32_t vn;
    if (!ix_string(&v, &vn)) return -1;
    emit_str_n(path, meta, v, vn);
    if (pbits(path) & PB_DICT) dict_row_str(path, meta, v, vn);
    return T_STRING;
  }
  if (c == '}' || c == ']' || c == ':' || c == ',') return -1;
  return ix_scalar(true, path, meta);
}

// skip_value over tokens: a value nothing reads is a run of tokens; the same grammar, checked by a state machine with an explicit
// stack (bit per open container), no path lookups, no rows
int Flattener::ix_skip(int depth) {
  if (ixp_ >= ix_n_) return -1;
  const char* const js = ix_json_;
  uint8_t stack[100];
  int sp = 0, top = -1;
  char c;
value:
  if (depth + sp > 96 || ixp_ >= ix_n_) return -1;
  c = js[ix_[ixp_]];
  if (c == '{') {
    if (sp == 0) top = T_OBJECT;
    ixp_++;
    if (ixp_ < ix_n_ && js[ix_[ixp_]] == '}') { ixp_++; goto after; }
    stack[sp++] = 1;
    goto key;
  }
  if (c == '[') {
    if (sp == 0) top = T_ARRAY;
    ixp_++;
    if (ixp_ < ix_n_ && js[ix_[ixp_]] == ']') { ixp_++; goto after; }
    stack[sp++] = 0;
    goto value;
  }
  if (c == '"') { if (!ix_skip_string()) return -1; if (sp == 0) top = T_STRING; goto after; }
  if (c == '}' || c == ']' || c == ':' || c == ',') return -1;
  { const int t = ix_scalar(false, 0, 0); if (t < 0) return -1; if (sp == 0) top = t; }
after:
  if (sp == 0) return top;
  if (ixp_ >= ix_n_) return -1;
  c = js[ix_[ixp_]];
  if (c == ',') { ixp_++; if (stack[sp - 1]) goto key; goto value; }
  if (c == (stack[sp - 1] ? '}' : ']')) { ixp_++; sp--; goto after; }
  return -1;
key:
  if (depth + sp > 96) return -1;
  if (ixp_ >= ix_n_ || js[ix_[ixp_]] != '"' || !ix_skip_string()) return -1;
  if (ixp_ >= ix_n_ || js[ix_[ixp_]] != ':') return -1;
  ixp_++;
  goto value;
}

bool Flattener::fast_tree(const char* json, size_t len, uint32_t root, ObjFacts* facts, int* type) {
  cur_facts_ = facts; cur_root_ = root;
  int t;
  if (use_index_ && len < 0x7FFFFF00u) {
    ix_build(json, len);
    t = ix_value(root, 0, 0, 0, 0);
    cur_facts_ = nullptr;
    if (t < 0 || ixp_ != ix_n_) return false;
    *type = t;
    if (facts) facts->present = true;
    return true;
  }
  p_ = json; e_ = json + len;
  ix_json_ = json; ix_len_ = (uint32_t)len;   // (name8 reads eight bytes at a member name when they lie inside the document)
  t = fast_value(root, 0, 0, 0, 0);
  cur_facts_ = nullptr;
  if (t < 0) return false;
  ws();
  if (p_ != e_) return false;
  *type = t;
  if (facts) facts->present = true;
  return true;
}

namespace {
struct Span { const char* p = nullptr; size_t n = 0; bool set() const { return p != nullptr; } };
inline void skip_ws(const char*& p, const char* e) { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
// advances p over one JSON value without interpreting it (the spans that matter are parsed properly afterwards)
bool skip_value(const char*& p, const char* e) {
  skip_ws(p, e);
  if (p >= e) return false;
  auto skip_string = [&]() { p++; while (p < e && *p != '"') { if (*p == '\\') p++; p++; } if (p >= e) return false; p++; return true; };
  if (*p == '"') return skip_string();
  if (*p == '{' || *p == '[') {
    int depth = 0;
    while (p < e) {
      const char c = *p;
      if (c == '"') { if (!skip_string()) return false; continue; }
      if (c == '{' || c == '[') depth++;
      else if (c == '}' || c == ']') { depth--; if (depth == 0) { p++; return true; } }
      p++;
    }
    return false;
  }
  const char* s = p;
  while (p < e && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') p++;
  return p > s;
}
// members of a JSON object text: calls fn(key, key_len, value span); false on malformed text or an escaped / duplicate-prone key
template <class F> bool scan_members(const char* p, const char* e, F fn) {
  skip_ws(p, e);
  if (p >= e || *p != '{') return false;
  p++;
  skip_ws(p, e);
  if (p < e && *p == '}') { p++; skip_ws(p, e); return p == e; }
  for (;;) {
    skip_ws(p, e);
    if (p >= e || *p != '"') return false;
    const char* k = ++p;
    while (p < e && *p != '"' && *p != '\\') p++;
    if (p >= e || *p != '"') return false;   // escaped key: the general path words it
    const size_t kn = (size_t)(p - k);
    p++;
    skip_ws(p, e);
    if (p >= e || *p != ':') return false;
    p++;
    skip_ws(p, e);
    const char* v = p;
    if (!skip_value(p, e)) return false;
    if (!fn(k, kn, Span{v, (size_t)(p - v)})) return false;
    skip_ws(p, e);
    if (p < e && *p == ',') { p++; continue; }
    if (p < e && *p == '}') { p++; skip_ws(p, e); return p == e; }
    return false;
  }
}
inline bool span_is(const Span& s, char c) { return s.set() && s.n && *s.p == c; }
inline bool span_null(const Span& s) { return !s.set() || (s.n == 4 && memcmp(s.p, "null", 4) == 0); }
}  // namespace

// Matchable.Namespace of a review: the one handed along with it, else the nsCache entry of `nsfield` (matcher.go:37-39); nullptr:
// neither.  What a review takes from it is remembered per table part (NsMemo).  *bad: the Namespace text does not parse.
Flattener::NsMemo* Flattener::ns_memo_for(const RawReview& r, const char* nsp, uint32_t nsn, const NsCache& cache, bool* bad) {
  NsMemo* memo = nullptr;
  if (r.ns_json && r.ns_len) {
    auto it = ns_memo_.find(r.ns_json);
    if (it == ns_memo_.end() || it->second.len != r.ns_len) {
      NsMemo m;
      m.len = r.ns_len;
      try { Value v = parse_json(r.ns_json, r.ns_len); if (!v.is_null()) m.ns = v; } catch (const std::exception&) { *bad = true; return nullptr; }
      if (m.ns.defined()) m.nsname = obj_string(m.ns, "metadata", "name");
      it = ns_memo_.insert_or_assign(r.ns_json, std::move(m)).first;
    }
    if (it->second.ns.defined()) memo = &it->second;
  }
  if (!memo && nsn) {
    const std::string nsfield(nsp, nsn);
    auto it = ns_memo_name_.find(nsfield);
    if (it == ns_memo_name_.end()) {
      NsMemo m;
      m.ns = cache.get(nsfield);
      if (m.ns.defined()) m.nsname = obj_string(m.ns, "metadata", "name");
      it = ns_memo_name_.emplace(nsfield, std::move(m)).first;
    }
    memo = &it->second;
  }
  return memo;
}

int Flattener::add_json_request(const RawReview& r, const NsCache& cache, HostTable* out, std::string* obj_key, const ExcludeFn* excluded) {
  t_ = out;
  rev_cur_ = out->n_reviews % out->rpt;
  const size_t stage0 = stage_.size(), heap0 = out->heap.size();
  auto bail = [&]() { stage_.resize(stage0); out->heap.resize(heap0); return (int)DECLINED; };
  ctrs_.clear();
  ctr_touched_.clear();
  scratch_keep_.clear();
  review_flags_ = 0;
  vids_.clear();
  begin_review_keys();
  if (++review_gen_ == 0) { std::fill(ctr_gen_.begin(), ctr_gen_.end(), 0); review_gen_ = 1; }
  // 1. the envelope: spans of the members normalize_admission_request reads; anything else is dropped (Go decodes into a
  // struct) after a syntax check by the subtree parser
  static const char* const kNames[] = {"uid", "kind", "resource", "operation", "userInfo", "object", "oldObject", "options", "subResource",
                                       "requestSubResource", "name", "namespace", "requestKind", "requestResource", "dryRun"};
  enum { M_UID, M_KIND, M_RESOURCE, M_OP, M_USER, M_OBJ, M_OLD, M_OPTS, M_SUBRES, M_REQSUBRES, M_NAME, M_NS, M_REQKIND, M_REQRES, M_DRYRUN, M_COUNT };
  Span m[M_COUNT];
  std::vector<Span> unknown;
  bool ok = scan_members(r.json, r.json + r.json_len, [&](const char* k, size_t kn, Span v) {
    for (int i = 0; i < M_COUNT; i++)
      if (strlen(kNames[i]) == kn && memcmp(kNames[i], k, kn) == 0) { if (m[i].set()) return false; m[i] = v; return true; }   // duplicate member: general path
    unknown.push_back(v);
    return true;
  });
  if (!ok) return bail();
  const uint32_t scratch_path = child(0, "$skip");
  for (const Span& u : unknown) {   // syntax check only: parsed into a scratch subtree that is dropped again
    const size_t sb = stage_.size(), hb = out->heap.size();
    int tt = -1;
    if (!fast_tree(u.p, u.n, scratch_path, nullptr, &tt)) return bail();
    stage_.resize(sb); out->heap.resize(hb);
  }
  // a string member without escapes, viewed in place ("" for absent / non-string members: str_field)
  bool declined = false;
  auto str_of = [&](const Span& s, const char** p, uint32_t* n) {
    *p = ""; *n = 0;
    if (!span_is(s, '"')) {   // not a string: its syntax still has to be valid
      if (s.set()) { const size_t sb = stage_.size(), hb = out->heap.size(); int tt = -1; if (!fast_tree(s.p, s.n, scratch_path, nullptr, &tt)) declined = true; stage_.resize(sb); out->heap.resize(hb); }
      return;
    }
    if (s.n < 2 || memchr(s.p, '\\', s.n)) { declined = true; return; }   // escapes: general path
    for (size_t i = 1; i + 1 < s.n; i++) if ((unsigned char)s.p[i] < 0x20) { declined = true; return; }
    *p = s.p + 1; *n = (uint32_t)(s.n - 2);
  };
  auto triple = [&](const Span& s, const char* a, const char* b, const char* c, uint32_t path) {
    Span f[3];
    if (span_is(s, '{')) {
      if (!scan_members(s.p, s.p + s.n, [&](const char* k, size_t kn, Span v) {
            const char* names[3] = {a, b, c};
            for (int i = 0; i < 3; i++) if (strlen(names[i]) == kn && memcmp(names[i], k, kn) == 0) { if (f[i].set()) return false; f[i] = v; return true; }
            const size_t sb = stage_.size(), hb = out->heap.size(); int tt = -1;
            const bool good = fast_tree(v.p, v.n, scratch_path, nullptr, &tt);
            stage_.resize(sb); out->heap.resize(hb);
            return good;
          })) { declined = true; return; }
    } else if (s.set()) { const char* dp; uint32_t dn; str_of(s, &dp, &dn); }   // (syntax check of a non-object member)
    emit(path, T_OBJECT, 3, 0);
    const char* names[3] = {a, b, c};
    for (int i = 0; i < 3; i++) { const char* sp; uint32_t sn; str_of(f[i], &sp, &sn); emit_str_n(child(path, names[i]), 0, sp, sn); }
  };
  uint32_t members = 8;   // uid kind resource operation userInfo object oldObject options
  const char* sp; uint32_t sn;
  if (!env_.ready) env_init();
  str_of(m[M_UID], &sp, &sn); emit_str_n(env_.uid, 0, sp, sn);
  triple(m[M_KIND], "group", "version", "kind", env_.kind);
  triple(m[M_RESOURCE], "group", "version", "resource", env_.resource);
  const char* op_p; uint32_t op_n;
  str_of(m[M_OP], &op_p, &op_n); emit_str_n(env_.operation, 0, op_p, op_n);
  if (declined) return bail();
  const bool del = op_n == 6 && memcmp(op_p, "DELETE", 6) == 0;
  int type = -1;
  if (span_is(m[M_USER], '{')) { if (!fast_tree(m[M_USER].p, m[M_USER].n, env_.user_info, nullptr, &type)) return bail(); }
  else { if (m[M_USER].set()) { str_of(m[M_USER], &sp, &sn); if (declined) return bail(); } emit(env_.user_info, T_OBJECT, 0, 0); }
  const bool has_obj = span_is(m[M_OBJ], '{'), has_old = span_is(m[M_OLD], '{');
  for (const Span* s : {&m[M_OBJ], &m[M_OLD]}) if (s->set() && !span_is(*s, '{')) { str_of(*s, &sp, &sn); if (declined) return bail(); }   // non-object: null, after a syntax check
  if (del && !has_old) return bail();   // ErrOldObjectIsNil: worded by the general path
  ObjFacts fobj, fold;
  const Span& osrc = del ? m[M_OLD] : m[M_OBJ];   // setObjectOnDelete (target.go:269-287)
  const bool obj_present = del || has_obj;
  if (obj_present) { if (!fast_tree(osrc.p, osrc.n, id_object_, &fobj, &type) || type != T_OBJECT) return bail(); }
  else emit(id_object_, T_NULL, 0, 0);
  if (has_old) { if (!fast_tree(m[M_OLD].p, m[M_OLD].n, id_old_, &fold, &type) || type != T_OBJECT) return bail(); }
  else emit(id_old_, T_NULL, 0, 0);
  if (m[M_OPTS].set()) { if (!fast_tree(m[M_OPTS].p, m[M_OPTS].n, env_.options, nullptr, &type)) return bail(); }
  else emit(env_.options, T_NULL, 0, 0);
  const char* rns_p = ""; uint32_t rns_n = 0;
  for (int i : {M_SUBRES, M_REQSUBRES, M_NAME, M_NS}) {
    str_of(m[i], &sp, &sn);
    if (declined) return bail();
    if (i == M_NS) { rns_p = sp; rns_n = sn; }
    if (sn) { emit_str_n(child(0, kNames[i]), 0, sp, sn); members++; }
  }
  for (int i : {M_REQKIND, M_REQRES, M_DRYRUN}) {
    if (span_null(m[i])) { if (m[i].set()) continue; else continue; }
    if (!fast_tree(m[i].p, m[i].n, child(0, kNames[i]), nullptr, &type)) return bail();
    members++;
  }
  // the webhook's process excluder looks at oldObject on DELETE else object, with the REQUEST's namespace (common.go:149-189)
  const ObjFacts& fx = del ? fold : fobj;
  if (excluded && (del ? has_old : has_obj) && fx.kind.set && fx.kind.n) {
    std::string av = fx.api_version.set ? std::string(fx.api_version.p, fx.api_version.n) : std::string();
    const bool core = std::count(av.begin(), av.end(), '/') != 1;   // group "" (schema.ParseGroupVersion)
    const std::string kind(fx.kind.p, fx.kind.n);
    if ((*excluded)(kind == "Namespace" && core, std::string(rns_p, rns_n), fx.name.set ? std::string(fx.name.p, fx.name.n) : std::string())) { bail(); return EXCLUDED; }
  }
  if (r.nsobj_json && r.nsobj_len) {
    const size_t before = stage_.size(), hb = out->heap.size();
    int t2 = -1;
    if (!fast_tree(r.nsobj_json, r.nsobj_len, env_.nsobj, nullptr, &t2)) return bail();
    if (t2 == T_NULL) { stage_.resize(before); out->heap.resize(hb); } else members++;
  }
  emit(0, T_OBJECT, members, 0);
  // Matchable.Namespace: the review's, else the nsCache entry of the REQUEST namespace (matcher.go:37-39)
  bool ns_bad = false;
  NsMemo* memo = ns_memo_for(r, rns_p, rns_n, cache, &ns_bad);
  if (ns_bad) return bail();
  static const Value no_ns;
  static const std::string no_name;
  const bool ns_defined = memo && memo->ns.defined();
  emit(id_m_, T_OBJECT, 2, 0);
  if (obj_present) fast_match_facts_n(fobj, ns_defined, ns_defined ? memo->nsname : no_name, false);
  if (has_old) fast_match_facts_n(fold, ns_defined, ns_defined ? memo->nsname : no_name, true);
  if (obj_key) {   // the audit sort key: object (after setObjectOnDelete), else oldObject
    const ObjFacts& k = obj_present ? fobj : fold;
    std::string& key = *obj_key;
    key.clear();
    if (obj_present || has_old) {
      std::string av = k.api_version.set ? std::string(k.api_version.p, k.api_version.n) : std::string(), group, version;
      const size_t nsl = std::count(av.begin(), av.end(), '/');
      if (!(av.empty() || av == "/")) { if (nsl == 0) version = av; else if (nsl == 1) { size_t i = av.find('/'); group = av.substr(0, i); version = av.substr(i + 1); } }
      key = group; key.push_back('\0'); key += version; key.push_back('\0'); if (k.kind.set) key.append(k.kind.p, k.kind.n); key.push_back('\0');
      if (k.ns.set) key.append(k.ns.p, k.ns.n);
      key.push_back('\0');
      if (k.name.set) key.append(k.name.p, k.name.n);
    }
  }
  if (memo) finish_review_memo(memo, r.source, out);
  else finish_review(no_ns, r.source, out);
  return ADDED;
}

int Flattener::add_json(const RawReview& r, const NsCache& cache, HostTable* out, std::string* obj_key, const ExcludeFn* excluded) {
  if (!r.json) return DECLINED;
  if (r.kind == 0) return add_json_request(r, cache, out, obj_key, excluded);
  if (r.kind != 1) return DECLINED;
  t_ = out;
  rev_cur_ = out->n_reviews % out->rpt;
  const size_t stage0 = stage_.size(), heap0 = out->heap.size();
  auto bail = [&]() { stage_.resize(stage0); out->heap.resize(heap0); return (int)DECLINED; };
  ctrs_.clear();
  ctr_touched_.clear();
  scratch_keep_.clear();
  review_flags_ = 0;
  vids_.clear();
  begin_review_keys();
  if (++review_gen_ == 0) { std::fill(ctr_gen_.begin(), ctr_gen_.end(), 0); review_gen_ = 1; }
  const std::string op = r.operation ? r.operation : "";
  const bool del = op == "DELETE";   // target.go:151-154 + setObjectOnDelete: the object is both oldObject and object
  ObjFacts fobj, fold;
  int type = -1;
  if (!fast_tree(r.json, r.json_len, id_object_, &fobj, &type) || type != T_OBJECT) return bail();
  if (del && (!fast_tree(r.json, r.json_len, id_old_, &fold, &type) || type != T_OBJECT)) return bail();
  // the request around it (normalize_object + normalize_admission_request); schema.ParseGroupVersion on the captured text
  if (!env_.ready) env_init();
  const char* const avp = fobj.api_version.set ? fobj.api_version.p : ""; const uint32_t avn = fobj.api_version.set ? fobj.api_version.n : 0;
  const char* const kindp = fobj.kind.set ? fobj.kind.p : ""; const uint32_t kindn = fobj.kind.set ? fobj.kind.n : 0;
  const char* groupp = ""; uint32_t groupn = 0; const char* verp = ""; uint32_t vern = 0;
  {
    uint32_t nsl = 0, first = 0;
    for (uint32_t i = 0; i < avn; i++) if (avp[i] == '/') { if (!nsl) first = i; nsl++; }
    if (!(avn == 0 || (avn == 1 && avp[0] == '/'))) {
      if (nsl == 0) { verp = avp; vern = avn; }
      else if (nsl == 1) { groupp = avp; groupn = first; verp = avp + first + 1; vern = avn - first - 1; }
    }
  }
  const char* const namep = fobj.name.set ? fobj.name.p : ""; const uint32_t namen = fobj.name.set ? fobj.name.n : 0;
  const char* const nsp = fobj.ns.set ? fobj.ns.p : ""; const uint32_t nsn = fobj.ns.set ? fobj.ns.n : 0;
  if (excluded && (*excluded)(kindn == 9 && memcmp(kindp, "Namespace", 9) == 0 && groupn == 0, std::string(nsp, nsn), std::string(namep, namen))) { bail(); return EXCLUDED; }
  uint32_t members = 8;   // uid kind resource operation userInfo object oldObject options
  emit_str_n(env_.uid, 0, "", 0);
  emit(env_.kind, T_OBJECT, 3, 0);
  emit_str_n(env_.k_group, 0, groupp, groupn); emit_str_n(env_.k_version, 0, verp, vern); emit_str_n(env_.k_kind, 0, kindp, kindn);
  emit(env_.resource, T_OBJECT, 3, 0);
  emit_str_n(env_.r_group, 0, "", 0); emit_str_n(env_.r_version, 0, "", 0); emit_str_n(env_.r_resource, 0, "", 0);
  emit_str_n(env_.operation, 0, op.data(), (uint32_t)op.size());
  emit(env_.user_info, T_OBJECT, 0, 0);
  if (!del) emit(id_old_, T_NULL, 0, 0);
  emit(env_.options, T_NULL, 0, 0);
  if (namen) { emit_str_n(env_.name, 0, namep, namen); members++; }
  if (nsn) { emit_str_n(env_.ns, 0, nsp, nsn); members++; }
  if (r.nsobj_json && r.nsobj_len) {
    const size_t before = stage_.size(), hb = out->heap.size();
    int t2 = -1;
    if (!fast_tree(r.nsobj_json, r.nsobj_len, env_.nsobj, nullptr, &t2)) return bail();
    if (t2 == T_NULL) { stage_.resize(before); out->heap.resize(hb); } else members++;
  }
  emit(0, T_OBJECT, members, 0);
  // Matchable.Namespace: the review's, else the nsCache entry of the request namespace (matcher.go:37-39)
  bool ns_bad = false;
  NsMemo* memo = ns_memo_for(r, nsp, nsn, cache, &ns_bad);
  if (ns_bad) return bail();
  static const Value no_ns;
  static const std::string no_name;
  const bool ns_defined = memo && memo->ns.defined();
  emit(id_m_, T_OBJECT, 2, 0);
  fast_match_facts_n(fobj, ns_defined, ns_defined ? memo->nsname : no_name, false);
  if (del) fast_match_facts_n(fold, ns_defined, ns_defined ? memo->nsname : no_name, true);
  if (obj_key) {
    std::string& k = *obj_key;
    k.clear();
    k.reserve(groupn + vern + kindn + nsn + namen + 4);
    k.append(groupp, groupn); k.push_back('\0'); k.append(verp, vern); k.push_back('\0'); k.append(kindp, kindn); k.push_back('\0'); k.append(nsp, nsn); k.push_back('\0'); k.append(namep, namen);
  }
  if (memo) finish_review_memo(memo, r.source, out);
  else finish_review(no_ns, r.source, out);
  return ADDED;
}

void Flattener::env_init() {
  EnvIds& v = env_;
  v.uid = child(0, "uid");
  v.kind = child(0, "kind"); v.k_group = child(v.kind, "group"); v.k_version = child(v.kind, "version"); v.k_kind = child(v.kind, "kind");
  v.resource = child(0, "resource"); v.r_group = child(v.resource, "group"); v.r_version = child(v.resource, "version"); v.r_resource = child(v.resource, "resource");
  v.operation = child(0, "operation"); v.user_info = child(0, "userInfo"); v.options = child(0, "options");
  v.name = child(0, "name"); v.ns = child(0, "namespace"); v.nsobj = child(0, "namespaceObject");
  for (int w = 0; w < 2; w++) {
    v.m_sub[w] = child(id_m_, w ? "old" : "o");
    v.m_group[w] = child(v.m_sub[w], "group"); v.m_kind[w] = child(v.m_sub[w], "kind"); v.m_name[w] = child(v.m_sub[w], "name");
    v.m_gname[w] = child(v.m_sub[w], "gname"); v.m_nsname[w] = child(v.m_sub[w], "nsname");
    v.m_d[w] = child(v.m_sub[w], "$d"); v.m_c[w] = child(v.m_sub[w], "$c");
  }
  v.ready = true;
}
// match_facts() on the captured strings, Matchable.Namespace reduced to its name
void Flattener::fast_match_facts_n(const ObjFacts& f, bool ns_defined, const std::string& ns_name, bool is_old) {
  if (!env_.ready) env_init();
  const int w = is_old ? 1 : 0;
  const char* const avp = f.api_version.set ? f.api_version.p : ""; const uint32_t avn = f.api_version.set ? f.api_version.n : 0;
  const char* const kindp = f.kind.set ? f.kind.p : ""; const uint32_t kindn = f.kind.set ? f.kind.n : 0;
  uint32_t nsl = 0, first = 0;
  for (uint32_t i = 0; i < avn; i++) if (avp[i] == '/') { if (!nsl) first = i; nsl++; }
  const uint32_t groupn = (!(avn == 0 || (avn == 1 && avp[0] == '/')) && nsl == 1) ? first : 0;
  const bool is_ns = kindn == 9 && memcmp(kindp, "Namespace", 9) == 0 && groupn == 0;
  emit(env_.m_sub[w], T_OBJECT, 0, 0);
  uint64_t acc[2] = {0, 0};
  match_fact(env_.m_group[w], avp, groupn, acc);
  match_fact(env_.m_kind[w], kindp, kindn, acc);
  const char* const namep = f.name.set ? f.name.p : ""; const uint32_t namen = f.name.set ? f.name.n : 0;
  const uint32_t nsn = f.ns.set ? f.ns.n : 0;
  match_fact(env_.m_name[w], namep, namen, acc, false);
  match_fact(env_.m_gname[w], f.gname.set ? f.gname.p : "", f.gname.set ? f.gname.n : 0, acc, false);
  bool has_nsname = true;
  if (is_ns) match_fact(env_.m_nsname[w], namep, namen, acc);
  else if (ns_defined) match_fact(env_.m_nsname[w], ns_name.data(), (uint32_t)ns_name.size(), acc);
  else if (nsn) match_fact(env_.m_nsname[w], f.ns.p, nsn, acc);
  else has_nsname = false;
  match_group_row(w, acc);
  review_flags_ |= is_old ? RF_HAS_OLD : RF_HAS_OBJ;
  if (is_ns) review_flags_ |= is_old ? RF_OLD_IS_NS : RF_OBJ_IS_NS;
  if (nsn) review_flags_ |= is_old ? RF_OLD_HAS_NSFIELD : RF_OBJ_HAS_NSFIELD;
  if (has_nsname) review_flags_ |= is_old ? RF_OLD_HAS_NSNAME : RF_OBJ_HAS_NSNAME;
  if (f.labels_bad) review_flags_ |= is_old ? RF_OLD_LABELS_BAD : RF_OBJ_LABELS_BAD;
  if (kindn == 0) review_flags_ |= is_old ? RF_OLD_BAD : RF_OBJ_BAD;
}

void Flattener::finish(HostTable* out) {
  flush(out);
  build_index(out);
}

void Flattener::build_index(HostTable* out) {
  out->tile_seg.push_back((uint32_t)out->segs.size());
  // slots = distinct paths of the table in path-id order; dense index [tile][slot] of first rows
  std::vector<uint32_t> paths;
  for (const auto& s : out->segs) paths.push_back(s.path);
  std::sort(paths.begin(), paths.end());
  paths.erase(std::unique(paths.begin(), paths.end()), paths.end());
  out->slot_path = paths;
  const uint32_t S = (uint32_t)paths.size(), T = out->n_tiles();
  out->tile_idx.assign((size_t)T * (S + 1), 0);
  for (uint32_t t = 0; t < T; t++) {
    uint32_t* ix = &out->tile_idx[(size_t)t * (S + 1)];
    const uint32_t s0 = out->tile_seg[t], s1 = out->tile_seg[t + 1];
    const uint32_t tile_end = s1 < out->segs.size() ? out->segs[s1].start : (uint32_t)out->n_rows_total;
    uint32_t k = s1;            // walk the tile's segments from the right; absent slots start where the next present one does
    uint32_t next = tile_end;
    for (uint32_t s = S; s-- > 0;) {
      if (k > s0 && out->segs[k - 1].path == paths[s]) { k--; next = out->segs[k].start; }
      ix[s] = next;
    }
    ix[S] = tile_end;
  }
  out->segs.clear(); out->segs.shrink_to_fit();
  out->tile_seg.clear();
}

}  // namespace gk
