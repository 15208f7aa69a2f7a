// Review normalisation + flattening (host).
//
//  * normalize_*: what K8sValidationTarget.HandleReview does before matching/evaluation
//      pkg/target/target.go:81-138 (8 input shapes -> gkReview), :140-179 (unstructuredToAdmissionRequest),
//      :269-287 (setObjectOnDelete, ErrOldObjectIsNil), pkg/target/matcher.go:37-39 (nsCache fallback).
//  * Flattener: key-path -> value SoA rows (plan.hpp Row) + string heap + per-review header with the match-layer
//    facts (RF_*) that pkg/mutation/match/match.go:73-258 derives from object/namespace/source.
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <shared_mutex>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "dexpr.hpp"
#include "plan.hpp"
#include "value.hpp"

namespace gk {

// ------------------------------------------------------------------------------------------------ path dictionary
// Interns wildcarded key paths ("object.spec.containers[].image") to dense ids; exact, append-only, thread-safe.
class PathDict {
 public:
  struct Info {
    uint32_t parent;
    std::string key;     // member name; empty for array elements
    bool is_elem;        // "[]" step
    uint8_t adepth;      // number of "[]" steps on the path (including this one)
  };
  static constexpr uint32_t kNone = 0xFFFFFFFFu;

  PathDict();
  uint32_t root() const { return 0; }
  uint32_t child(uint32_t parent, const std::string& key);   // interns
  uint32_t elem(uint32_t parent);                            // interns the "[]" child
  uint32_t find_child(uint32_t parent, const std::string& key) const;   // kNone if unknown
  Info info(uint32_t id) const;
  uint32_t size() const;
  std::string to_string(uint32_t id) const;

 private:
  struct KeyHash { size_t operator()(const std::pair<uint32_t, std::string>& k) const; };
  mutable std::shared_mutex mu_;
  std::unordered_map<std::pair<uint32_t, std::string>, uint32_t, KeyHash> map_;
  std::vector<Info> infos_;
  uint32_t intern(uint32_t parent, const std::string& key, bool is_elem);
};

// ------------------------------------------------------------------------------------------------ path patterns
// string test on the member NAME of a key iteration (`startswith(key, "x")` ..): decided when patterns are resolved against
// the table's key paths, never on the device.  An array index is a number: it fails every positive test (the builtin is
// undefined for it) and passes every negated one.
struct KeyPred { uint8_t op = 0; bool neg = false; std::string s; };   // op: 6 prefix, 7 suffix, 8 contains, 9 "is a member name"
inline bool key_pred_holds(const KeyPred& p, const std::string& key, bool is_elem) {
  bool t = false;
  if (!is_elem) {
    if (p.op == KC_PREFIX) t = key.compare(0, p.s.size(), p.s) == 0 && key.size() >= p.s.size();
    else if (p.op == KC_SUFFIX) t = key.size() >= p.s.size() && key.compare(key.size() - p.s.size(), p.s.size(), p.s) == 0;
    else if (p.op == KC_CONTAINS) t = key.find(p.s) != std::string::npos;
    else t = true;   // KC_ISNAME
  }
  return p.neg ? !t : t;
}

struct PatStep {
  bool any = false;                 // true: any single step
  bool elems_only = false;          // any: only "[]" children (array-element scopes)
  std::string key;                  // !any: exact member name
  std::vector<std::string> only;    // any: member name must be one of (key iteration with ==)
  std::vector<std::string> except;  // any: member name must not be one of
  std::vector<KeyPred> kpreds;      // any: string tests the member name must pass
};
typedef std::vector<PatStep> Pattern;
std::string pattern_to_string(const Pattern& p);
bool pattern_matches(const Pattern& pat, const PathDict& dict, uint32_t path_id);
// the first steps of `pat` match the whole path (the pattern reaches this path or something below it); *full: it matches exactly
bool pattern_reaches(const Pattern& pat, const PathDict& dict, uint32_t path_id, bool* full);
// review.$m.<o|old>.<fact>: a leaf of a candidate's MATCH GROUP (the group's dictionary expressions share one row, see DictRegistry::intern)
bool match_group_pattern(const Pattern& pat);
// a leaf of the REVIEW FACTS group (round 6): no iteration step on the way (review.object.metadata.name, review.object.metadata.labels.env,
// review.$ns.metadata.labels.pci, review.object.spec.hostNetwork ...), outside the match group.  Such a leaf occurs at most once per
// review, so the dictionary expressions on ALL of them share one bit space and travel in ONE row per review, review.$r.$d (the
// flattener ORs the leaves' masks) -- where every such leaf used to cost a row (and most a string header) and a chunk per row group.
bool review_fact_pattern(const Pattern& pat);
extern std::atomic<int> g_debug_dict_facts;

// ------------------------------------------------------------------------------------------------ dictionary predicates
// Leaf-local expressions (dexpr.hpp) registered by the loaded constraints: pattern of the leaf -> expressions, each with a
// bit.  The flattener evaluates them per distinct leaf value and ships the answers as an integer row at <leaf>.$d.
struct DictEntry { DX dx; std::string key; uint32_t bit; };
class DictRegistry {
 public:
  // bit of the expression on leaves matching `leaf`; throws std::runtime_error beyond 62 bits -- and, with add = false (a plan
  // that must live with what the loaded constraints registered: the totals plans), when the entry does not exist yet
  // is_facts (may be NULL): the bit lives in the review facts row review.$r.$d, not in <leaf>.$d (review_fact_pattern; a leaf whose
  // first expression arrived without a row-predicate fallback keeps a row of its own -- see intern)
  uint32_t intern(const Pattern& leaf, const DX& dx, bool add = true, bool* is_facts = nullptr, bool has_fallback = false);
  uint64_t gen() const;                                  // bumped by every new entry
  void match(const PathDict& dict, uint32_t path_id, std::vector<DictEntry>* out, int* pat_index = nullptr, bool* is_facts = nullptr) const;   // entries for a concrete leaf path
  // answers shared by every flattener of the engine, per pattern: distinct value -> bit mask.  A value is evaluated once
  // per engine (not once per table part and thread); the memo of a pattern is dropped when it gains an expression.
  bool memo_get(int pat_index, size_t n_entries, const std::string& key, uint64_t* mask) const;
  void memo_put(int pat_index, size_t n_entries, const std::string& key, uint64_t mask);
  // Guards: container paths under which the loaded constraints iterate ARRAY elements.  A review holding a non-empty
  // OBJECT there is refused (RF_REFUSE): Rego's `x[_]` would walk the object's values, the compiled plan would not.
  bool add_guard(const Pattern& container, bool add = true);   // false: not registered (and add = false)
  bool guarded(const PathDict& dict, uint32_t path_id) const;
  // Compared values: leaf patterns whose rows the loaded constraints compare with other review values (P_STORE).  The
  // flattener gives the rows of matching paths a VALUE ID (plan.hpp ROW_VID_*), unique per distinct value within the review.
  bool add_value(const Pattern& leaf, bool add = true);        // false: not registered (and add = false)
  bool valued(const PathDict& dict, uint32_t path_id) const;
  // Message keys (round 4, result counting: pe.hpp Template::count_forms): leaf patterns whose values lead the messages of a
  // per-element violation.  A review in which two rows of such paths hold the SAME value -- or one that is no scalar -- gets the
  // synthetic row  review.$dup : true ; the counting plans then leave the review's pairs to the host renderer.
  bool add_key(const Pattern& leaf, bool add = true);
  bool keyed(const PathDict& dict, uint32_t path_id) const;
  // Element carriers (plan.hpp T_ABSENT): `elem` = an array-element pattern X[] the plans iterate, `member` = the member of the element
  // whose rows carry the element marker.  ONE carrier per element pattern, first come first served: *chosen = the registered member
  // (the caller's or an earlier one).  false: none registered (and add = false, or no member offered).  A new carrier makes the
  // tables flattened so far stale (gen): they lack the T_ABSENT rows of elements without the member.
  bool add_carrier(const Pattern& elem, const std::string& member, bool add, std::string* chosen);
  bool carrier_of(const PathDict& dict, uint32_t elem_path_id, std::string* member) const;
  // The COUNTING space (round 4): the dictionary expressions the result-counting plans read live in a registry of their own and
  // travel in a row of their own, <leaf>.$c -- the 62 bits per leaf of <leaf>.$d belong to the violation formulas alone (a
  // constraint must never become unloadable because the totals of another one took its bits).
  // The READ SET (round 4, GK_TABLE_PRUNED): the patterns of every path some loaded plan has a predicate on (engine.cpp ensure_plan)
  // plus the paths the counting forms name.  A pruned table holds rows for these paths only and its parser walks past subtrees no
  // pattern of any kind (reads, dictionary leaves, guards, value and key paths) reaches into.  A change of the set makes pruned
  // tables stale (reads_gen), as a new dictionary predicate makes every table stale (gen).
  bool set_reads(const std::vector<Pattern>& pats);   // true: the set changed
  bool reads_has(const Pattern& pat) const;            // some pattern of the published read set matches every path this one matches
  uint64_t reads_gen() const { return reads_gen_.load(std::memory_order_acquire); }
  // bit 0: rows of `path_id` are read; bit 1: some pattern reaches `path_id` or below it (the parser must visit it)
  uint32_t read_state(const PathDict& dict, uint32_t path_id) const;
  // the member names through which some pattern (of the read set or of anything else the flattener acts on) goes on below
  // `path_id`; false: some pattern continues with a step that takes any name -- every member must be looked at
  bool child_names(const PathDict& dict, uint32_t path_id, std::vector<std::string>* names) const;
  DictRegistry& counting() { std::unique_lock<std::shared_mutex> l(mu_); if (!counting_) counting_.reset(new DictRegistry()); return *counting_; }
  const DictRegistry* counting_if_any() const { std::shared_lock<std::shared_mutex> l(mu_); return counting_.get(); }
 private:
  struct Pat { Pattern pat; std::string key; std::vector<DictEntry> entries; std::unordered_map<std::string, uint64_t> memo; bool facts = false; /* bits of review.$r.$d */ };
  mutable std::shared_mutex mu_;
  std::vector<Pat> pats_;
  std::vector<std::pair<std::string, Pattern>> guards_, values_, keys_;
  struct Carrier { std::string key; Pattern elem; std::string member; };
  std::vector<Carrier> carriers_;
  std::atomic<uint64_t> gen_{0};
  std::unique_ptr<DictRegistry> counting_;
  std::vector<std::pair<std::string, Pattern>> reads_;
  std::atomic<uint64_t> reads_gen_{0};
  void interest(std::vector<const Pattern*>* out) const;   // every pattern that makes the flattener do something below a path (caller holds mu_)
  // What the patterns say about a concrete path is the same for every flattener of the engine: worked out once per path and state
  // of the registry (a table part per host thread, 64 threads: each of them matching every path against every pattern was half
  // the wall clock of a 200-template table's first ingest).  Dropped when gen() / reads_gen() move.
  struct Facts { uint8_t known = 0, read = 0; bool guarded = false, valued = false, keyed = false; int pat = -1; };
  enum : uint8_t { F_READ = 1, F_GUARD = 2, F_VALUE = 4, F_KEY = 8, F_PAT = 16 };
  mutable std::shared_mutex fmu_;
  mutable std::vector<Facts> facts_;
  mutable std::unordered_map<uint32_t, std::pair<bool, std::vector<std::string>>> names_;
  mutable uint64_t facts_stamp_ = ~0ull;
  uint32_t read_state_now(const PathDict& dict, uint32_t path_id) const;
  bool child_names_now(const PathDict& dict, uint32_t path_id, std::vector<std::string>* names) const;
  uint64_t stamp() const { return gen() * 0x9E3779B97F4A7C15ull + reads_gen(); }
  bool facts_get(uint64_t st, uint32_t path, uint8_t bit, Facts* out) const;
  template <class F> void facts_put(uint64_t st, uint32_t path, uint8_t bit, F set) const {
    if (st != stamp()) return;   // the registry moved while the fact was being worked out
    std::unique_lock<std::shared_mutex> l(fmu_);
    if (facts_stamp_ != st) { facts_.clear(); names_.clear(); facts_stamp_ = st; }
    if (path >= facts_.size()) facts_.resize((size_t)path * 2 + 64);
    set(facts_[path]);
    facts_[path].known |= bit;
  }
};

uint32_t hash32(const uint8_t* p, size_t n);
inline uint32_t hash32(const std::string& s) { return hash32((const uint8_t*)s.data(), s.size()); }

// ------------------------------------------------------------------------------------------------ reviews
enum SourceType : int { SRC_EMPTY = 0, SRC_ORIGINAL = 1, SRC_GENERATED = 2, SRC_ALL = 3, SRC_INVALID = 4 };

struct ReviewDoc {
  Value request;        // canonical input.review object (AdmissionRequest JSON incl. namespaceObject when provided)
  Value match_ns;       // Matchable.Namespace (Undefined = nil)
  int source = SRC_EMPTY;
};

struct ReviewError : std::runtime_error { using std::runtime_error::runtime_error; };

class NsCache {   // pkg/target/ns_cache.go:15-87
 public:
  void put(const std::string& name, const Value& ns);
  void remove(const std::string& name);
  Value get(const std::string& name) const;   // Undefined if absent
 private:
  mutable std::shared_mutex mu_;
  std::unordered_map<std::string, Value> m_;
};

// AdmissionRequest JSON (+ gkReview.namespace, + reviews.Namespace option) -> ReviewDoc. Throws ReviewError.
ReviewDoc normalize_admission_request(const Value& request, const Value& match_ns, const Value& ns_object, int source,
                                      const NsCache& cache);
// Unstructured / AugmentedUnstructured -> ReviewDoc (target.go:140-179).
ReviewDoc normalize_object(const Value& object, const Value& match_ns, const Value& ns_object, int source,
                           const std::string& operation, const NsCache& cache);

// unstructured accessors shared with the host matcher / renderer
std::string obj_string(const Value& obj, const char* a, const char* b = nullptr);
void obj_gvk(const Value& obj, std::string* group, std::string* version, std::string* kind);
bool obj_is_namespace(const Value& obj);

// Host staging memory.  A table part is a few to a few hundred MB of rows, string headers and heap bytes that live until the
// part has been copied to the device.  On a many-core host (256 threads building one table) the allocator traffic of such
// blocks -- mmap, page-fault in, grow by mremap, munmap with a TLB shoot-down on every core -- serialises the threads on the
// process's address-space lock: round 2 measured 4.6x on 256 threads.  Blocks of 1 MiB and more therefore come from a
// process-wide pool: size classes of powers of two, handed back to the pool
// (not to the kernel) when a part is released, up to GK_HOST_POOL_MB (default 8192) kept.  Smaller blocks use malloc / realloc.
// CPUs this process may actually use: the hardware threads, cut down to the scheduler affinity mask and to the cgroup CPU
// bandwidth quota (cpu.max, or cfs_quota_us / cfs_period_us under cgroup v1).  A container that sees 256 hardware threads
// behind a 16-CPU quota is throttled for most of every period when 256 workers run (measured in round 3: RESULT totals of
// 200 k objects 0.78 s on 16 threads, 2.6 s on 256; profiles/r03_host_threads_under_cpu_quota.json).  Every host pool sizes
// itself by this number; GK_HOST_THREADS overrides it.
size_t host_cpus();
void* host_block_alloc(size_t bytes, size_t* cap_bytes);   // bytes >= kHostBlockMin; *cap_bytes = the size class
void host_block_free(void* p, size_t cap_bytes);
constexpr size_t kHostBlockMin = 1u << 20;

// Growable array of PODs WITHOUT value-initialisation: the table arrays hold gigabytes, std::vector::resize would zero
// every byte before it is overwritten and re-copy everything on growth.
template <class T>
struct PodVec {
  T* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
  size_t pooled_ = 0;   // size class in bytes when p_ is a pooled block, 0: malloc'ed
  PodVec() {}
  PodVec(const PodVec& o) { assign(o.p_, o.n_); }
  PodVec(PodVec&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), pooled_(o.pooled_) { o.p_ = nullptr; o.n_ = o.cap_ = o.pooled_ = 0; }
  PodVec& operator=(const PodVec& o) { if (this != &o) assign(o.p_, o.n_); return *this; }
  PodVec& operator=(PodVec&& o) noexcept { if (this != &o) { release(); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; pooled_ = o.pooled_; o.p_ = nullptr; o.n_ = o.cap_ = o.pooled_ = 0; } return *this; }
  ~PodVec() { release(); }
  void release() { if (pooled_) host_block_free(p_, pooled_); else free(p_); p_ = nullptr; n_ = cap_ = pooled_ = 0; }
  void assign(const T* src, size_t n) { resize(n); if (n) memcpy(p_, src, n * sizeof(T)); }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T* data() { return p_; }
  const T* data() const { return p_; }
  T& operator[](size_t i) { return p_[i]; }
  const T& operator[](size_t i) const { return p_[i]; }
  T& back() { return p_[n_ - 1]; }
  T* begin() { return p_; }
  T* end() { return p_ + n_; }
  const T* begin() const { return p_; }
  const T* end() const { return p_ + n_; }
  // capacity for c elements from the host staging pool, for a vector that is sized up front (a part of a table build)
  void presize(size_t c) {
    if (c <= cap_ || c * sizeof(T) < kHostBlockMin) { reserve(c); return; }
    size_t cap_bytes = 0;
    T* q = static_cast<T*>(host_block_alloc(c * sizeof(T), &cap_bytes));
    if (!q) throw std::bad_alloc();
    if (n_) memcpy(static_cast<void*>(q), p_, n_ * sizeof(T));
    if (pooled_) host_block_free(p_, pooled_); else free(p_);
    p_ = q; cap_ = cap_bytes / sizeof(T); pooled_ = cap_bytes;
  }
  void reserve(size_t c) {
    if (c <= cap_) return;
    if (pooled_) { presize(c); return; }   // (a pooled block is never realloc'ed)
    T* q = static_cast<T*>(realloc(p_, c * sizeof(T)));   // organic growth: realloc moves pages instead of copying them
    if (!q) throw std::bad_alloc();
    p_ = q; cap_ = c;
  }
  void resize(size_t n) {   // new elements are UNINITIALISED
    if (n > cap_) reserve(std::max(n, cap_ + cap_ / 2 + 1024));
    n_ = n;
  }
  void resize_zero(size_t n) { size_t old = n_; resize(n); if (n > old) memset(static_cast<void*>(p_ + old), 0, (n - old) * sizeof(T)); }
  void push_back(const T& v) { if (n_ == cap_) reserve(cap_ + cap_ / 2 + 1024); p_[n_++] = v; }
  void append(const T* src, size_t n) { size_t old = n_; resize(old + n); if (n) memcpy(static_cast<void*>(p_ + old), src, n * sizeof(T)); }
  void clear() { n_ = 0; }
  void shrink_to_fit() { if (n_ == 0) release(); }
};

struct HostTable {
  PodVec<Row> rows;                 // row groups: per tile, sorted by path (plan.hpp)
  PodVec<StrHdr> shdr;              // parallel to rows
  std::vector<uint32_t> tile_idx;   // [n_tiles][n_slots + 1]
  std::vector<uint32_t> slot_path;  // [n_slots] path id of each slot, increasing
  std::vector<uint32_t> rflags;     // n_reviews
  PodVec<uint8_t> heap;
  std::vector<uint32_t> path_rows;  // rows per path over the whole table (algorithmic-byte accounting per plan)
  std::vector<uint32_t> path_max;   // per array-element path: largest element count of one review (plan specialisation)
  uint32_t n_reviews = 0;
  uint32_t rpt = GK_RPT_MIN;        // reviews per row group
  size_t n_rows_total = 0, heap_total = 0;   // sizes of rows / heap (kept when the arrays themselves have gone to the device)
  uint32_t n_tiles() const { return (n_reviews + rpt - 1) / rpt; }   // row groups
  uint32_t n_slots() const { return (uint32_t)slot_path.size(); }
  // build-time only: per-tile segment lists, turned into tile_idx by Flattener::finish
  struct SegRec { uint32_t path, start; };
  std::vector<SegRec> segs;
  std::vector<uint32_t> tile_seg;
  void append(const HostTable& part);
};

// One review as the C ABI hands it over (gk_review_in without the C types).
struct RawReview {
  int kind = 1;                    // 0 AdmissionRequest JSON, 1 bare object (gk_review_kind)
  int source = SRC_EMPTY;
  const char* json = nullptr; size_t json_len = 0;
  const char* ns_json = nullptr; size_t ns_len = 0;          // gkReview.namespace
  const char* nsobj_json = nullptr; size_t nsobj_len = 0;    // reviews.Namespace option -> input.review.namespaceObject
  const char* operation = nullptr;
};

class Flattener {
 public:
  explicit Flattener(PathDict* dict, const DictRegistry* reg = nullptr);
  void set_pruning(bool on) { const bool p = on && reg_ != nullptr; if (p != pruning_) { pbits_.clear(); kid_filter_.clear(); kid_arena_.clear(); } pruning_ = p; }
  // A Flattener may serve many tables one after the other (engine.cpp keeps one per host worker thread: its member-name
  // table, its path caches and its memo of dictionary answers then survive from batch to batch instead of being rebuilt --
  // through the engine's shared, locked structures -- by every thread for every table).  begin_table() starts a table:
  // drops what is only valid within one (Namespace documents cached by text pointer) and, when the registry of dictionary
  // predicates / guards / compared values has changed since the last table, the per-path answers derived from it.
  void begin_table();
  void add(const ReviewDoc& doc, HostTable* out);
  void add_skipped(HostTable* out);   // a slot that holds no rows and is never evaluated (RF_SKIP)
  // Fast ingest: ONE pass over the JSON text of a review straight into rows -- no Value
  // tree -- including HandleReview's normalisation (target.go:81-179, 269-287) and the match-layer facts.  Produces
  // exactly the rows add(normalize_*(parse_json(..))) produces (tests/test_ingest.py compares table digests).
  // Returns false -- with nothing added -- when the text needs the general path (malformed JSON, duplicate object keys,
  // nesting beyond the fast parser's depth ...): the caller then runs parse_json + normalize_* + add, which also words
  // the errors.  obj_key: the audit sort key of the object (group \0 version \0 kind \0 namespace \0 name).
  // `excluded` (may be NULL): called with (is a core Namespace, metadata.namespace, metadata.name) once the object's
  // identity is known; true -> the review is dropped again and EXCLUDED is returned (process excluder, engine.cpp).
  enum { DECLINED = 0, ADDED = 1, EXCLUDED = 2 };
  typedef std::function<bool(bool, const std::string&, const std::string&)> ExcludeFn;
  int add_json(const RawReview& r, const NsCache& cache, HostTable* out, std::string* obj_key, const ExcludeFn* excluded = nullptr);
  // ... of an admissionv1.AdmissionRequest document (the validating webhook's wire shape): the envelope members are
  // located with a skipping scan, object / oldObject / userInfo / options go through the same one-pass subtree parser
  int add_json_request(const RawReview& r, const NsCache& cache, HostTable* out, std::string* obj_key, const ExcludeFn* excluded);
  void finish(HostTable* out);   // flush + build_index
  void flush(HostTable* out);    // closes the tile being built (parallel table builds flush per part, then append)
  static void build_index(HostTable* out);   // slots + dense [tile][slot] index from the per-tile segment lists

 private:
  PathDict* dict_;
  const DictRegistry* reg_ = nullptr;
  uint64_t reg_gen_ = ~0ull, reads_gen_seen_ = ~0ull;
  struct StrMemo { struct Ent { uint64_t hash = 0; uint32_t off = 0, len = 0; uint64_t m[2] = {0, 0}; }; std::vector<Ent> tab; size_t count = 0; std::string arena;
                   // a column of (nearly) unique values -- image tags pinned by digest, generated names -- only fills the memo: after a window of
                   // 8192 look-ups with fewer than one hit in eight the next seven windows evaluate straight (round 6)
                   uint32_t seen = 0, hits = 0, bypass_left = 0; };
  struct DictPath { std::unique_ptr<StrMemo> smemo; int state = 0; int gstate = 0; int vstate = 0; int kstate = 0; uint32_t rstate = 0 /* 4 | read_state once known */; int pat = -1; std::unique_ptr<DxStrProg> sprog[2] /* the entries / centries compiled for string values (dexpr.hpp) */; bool facts = false /* the main-space answers go to the review facts row, review.$r.$d */; bool deep = false /* some expression looks inside a container leaf */; std::vector<DictEntry> entries; uint32_t dpath = 0; std::unordered_map<std::string, uint64_t> memo;   // state 0 unknown, 1 none, 2 has entries
                    int cpat = -1; std::vector<DictEntry> centries; uint32_t cpath = 0; std::unordered_map<std::string, uint64_t> cmemo; /* the counting space: <leaf>.$c */ };
  std::vector<DictPath> dict_paths_;
  void dict_row(uint32_t path, uint32_t meta, const Value& leaf, uint64_t* masks_out = nullptr);   // emits <leaf>.$d / .$c when some registered expression is true (masks_out: hands the two masks back instead)
  void dict_row_str(uint32_t path, uint32_t meta, const char* s, uint32_t n, uint64_t* masks_out = nullptr, bool use_memo = true);
  void match_fact(uint32_t path, const char* s, uint32_t n, uint64_t* acc, bool use_memo = true);   // one fact of a match candidate: string row + dictionary answers
  void match_group_row(int w, const uint64_t* acc);                            // review.$m.<o|old>.$d / .$c
  bool dict_wanted(uint32_t path);
  bool dict_deep(uint32_t path) { return dict_wanted(path) && dict_paths_[path].deep; }
  bool guard_wanted(uint32_t path);   // is `path` a container under which element predicates iterate? (cached per path)
  bool value_wanted(uint32_t path);   // are the rows of `path` compared with other review values? (cached per path)
  bool key_wanted(uint32_t path);     // do the rows of `path` lead per-element messages? (DictRegistry::add_key)
  // GK_TABLE_PRUNED: rows only for the registry's read set, subtrees nothing reaches into are validated and walked past
  bool pruning_ = false;
  // everything the hot path asks about a path, in one byte (lazily derived from the caches above; dropped with them)
  enum : uint8_t { PB_KNOWN = 1, PB_ROW = 2 /* rows of the path are kept */, PB_BELOW = 4 /* the parser must visit it */, PB_VALUE = 8, PB_KEY = 16, PB_DICT = 32, PB_GUARD = 64, PB_DEEP = 128 };
  std::vector<uint8_t> pbits_;
  // per element path: 0 = not looked up yet, 1 = no carrier, else 2 + the path id of the carrier member (follows the registry's generation)
  std::vector<uint32_t> carrier_cache_;
  uint32_t carrier_slow(uint32_t elem_path);
  uint32_t carrier_child(uint32_t elem_path) {   // 0: the element pattern has no carrier
    if (elem_path < carrier_cache_.size()) { const uint32_t c = carrier_cache_[elem_path]; if (c) return c == 1u ? 0u : c - 2u; }
    return carrier_slow(elem_path);
  }
  // WANTED MEMBERS of an object path in a pruned table (round 4): when no row of the object itself is kept, only the members some
  // pattern names can matter -- the others are walked past without a path id (no hashing of label keys, env names, port fields)
  struct KidEnt { uint32_t id, off, len; };
  struct KidFilter { uint8_t state = 0 /* 0 unknown, 1 every member, 2 the list */; std::vector<KidEnt> kids; };
  std::vector<std::unique_ptr<KidFilter>> kid_filter_;   // (by path; the entries stay where they are when the vector grows: a parse holds one per open object)
  std::string kid_arena_;
  const KidFilter* kid_filter(uint32_t path);   // nullptr: every member is looked up
  uint32_t rev_cur_ = 0;              // the current review's number inside its row group (every row carries it)
  uint8_t pbits_slow(uint32_t path);
  uint8_t pbits(uint32_t path) { if (path < pbits_.size()) { const uint8_t b = pbits_[path]; if (b) return b; } return pbits_slow(path); }
  uint32_t read_state(uint32_t path);   // DictRegistry::read_state, cached per path
  bool emit_row_wanted(uint32_t path) { return !pruning_ || (read_state(path) & 1u); }
  bool walk_always(uint32_t parent, uint32_t ch);
  int skip_value(int depth);            // p_ at a value: validates it as fast_value would, emits nothing; its RowType or -1
  std::vector<uint32_t> key_ids_;     // value ids of the key rows of the current review
  bool dup_seen_ = false;             // ... two of them were equal (or not a scalar): review.$dup
  uint32_t id_dup_ = 0;
  void begin_review_keys() { key_ids_.clear(); dup_seen_ = false; facts_acc_ = 0; }
  uint64_t facts_acc_ = 0;   // the review facts row of the review being flattened (review.$r.$d), emitted by finish_tail
  uint32_t id_facts_d_ = 0;
  // per-review interning of compared values -> value ids (plan.hpp ROW_VID_*)
  struct VidEnt { uint64_t key; uint32_t tag, off, id; };   // tag 1 number-as-int64, 2 float bits, 3 inline string, 4 heap string (key = hash32 | len << 32, off = heap offset)
  std::vector<VidEnt> vids_;
  uint32_t value_id(uint32_t meta, uint32_t lo, uint32_t hi);
  uint32_t id_object_, id_old_, id_m_, id_ns_;
  struct Ctr { uint32_t path, n; };
  std::vector<Ctr> ctrs_;
  HostTable* t_ = nullptr;
  uint32_t review_flags_ = 0;
  struct Staged { uint32_t path; Row row; StrHdr hdr; };
  std::vector<Staged> stage_;       // rows of the tile being built, review order / document order
  std::vector<uint32_t> order_;
  std::vector<uint32_t> sort_count_, sort_paths_;   // counting sort of a tile's rows by path
  void flush_tile(HostTable* out);
  std::vector<uint32_t> elem_cache_;   // path -> its "[]" child
  uint32_t child(uint32_t parent, const std::string& key);
  uint32_t elem(uint32_t parent);
  void walk(const Value& v, uint32_t path, uint32_t meta_ords, int adepth, uint32_t extra);
  void emit_absent(uint32_t path, uint32_t meta);   // the carrier row of an element without the carrier member (plan.hpp T_ABSENT)
  bool emit(uint32_t path, uint32_t meta, uint32_t lo, uint32_t hi, bool always = false);   // false: a pruned table does not hold rows of this path
  uint32_t put_string(const std::string& s, uint32_t* hash);
  void emit_str(uint32_t parent, const char* key, const std::string& s);
  void emit_string_row(uint32_t path, uint32_t meta, const std::string& s);
  void match_facts(const Value& obj, const Value& ns, bool is_old, uint32_t m_parent);
  void finish_review(const Value& ns, int source, HostTable* out);   // $ns rows, source flags, per-review bookkeeping

  // ---- fast ingest state
  struct Captured { const char* p = nullptr; uint32_t n = 0; bool set = false; };   // a string value seen at a known path
  struct ObjFacts { Captured api_version, kind, name, ns, gname; bool labels_bad = false; bool present = false; };
  struct KeySlot { uint64_t hash = 0, first8 = 0 /* the name's first eight bytes, zero-padded */; uint32_t parent = 0, id = 0, off = 0, len = 0; bool used = false; };
  std::vector<KeySlot> key_tab_;      // open addressing: (parent path, member name) -> child path
  std::string key_arena_;
  size_t key_count_ = 0, last_slot_ = 0;
  // PREDICTED children (round 4): objects of one kind list their members in one order, so the i-th member of the object at path P
  // is, nearly always, the member that was i-th there last time -- one length compare + memcmp instead of hashing the name.
  // pred_[P] = what the members of the most recent object at P were, by position (id kNone: nothing predicted yet)
  struct PredEnt { uint64_t first8 = 0; uint32_t id = 0xFFFFFFFFu, off = 0, len = 0; };
  std::vector<std::vector<PredEnt>> pred_;
  uint32_t fast_child_at(uint32_t parent, uint32_t pos, const char* key, uint32_t len);
  std::vector<uint32_t> ctr_gen_, ctr_val_;   // per element path: review generation / running ordinal
  std::vector<uint32_t> ctr_touched_;
  std::vector<uint32_t> dup_gen_;             // per path: id of the object instance that last produced it (duplicate keys)
  uint32_t review_gen_ = 0, obj_instance_ = 0;
  std::string scratch_;                       // decoded strings with escapes
  std::vector<std::unique_ptr<std::string>> scratch_keep_;     // ... that are captured (must outlive the review)
  const char* p_ = nullptr; const char* e_ = nullptr;
  ObjFacts* cur_facts_ = nullptr;             // facts of the object / oldObject tree being parsed
  uint32_t cur_root_ = 0;                     // its root path (object / oldObject)
  struct CapIds { uint32_t api_version, kind, metadata, name, ns, gname, labels; } cap_[2];   // [0] object, [1] oldObject
  // (round 4) what a review takes from its Namespace is the same for every review of that namespace: the name the match layer
  // compares and the $ns rows.  Kept per table part (begin_table drops it): the rows hold offsets into the part's heap.
  struct NsMemo {
    size_t len = 0; Value ns; std::string nsname;
    int rows_state = 0;               // 0 not recorded yet, 1 `rows` / `flags` replay, 2 not replayable (value ids, message keys, element counters)
    std::vector<Staged> rows; uint32_t flags = 0; uint64_t facts = 0; const HostTable* owner = nullptr;
  };
  std::unordered_map<const char*, NsMemo> ns_memo_;
  std::unordered_map<std::string, NsMemo> ns_memo_name_;
  bool emit_side_effects_ = false;    // an emit() interned a value id or compared a message key since this was cleared
  NsMemo* ns_memo_for(const RawReview& r, const char* nsp, uint32_t nsn, const NsCache& cache, bool* bad);
  void ns_rows(const Value& ns);                                        // the $ns rows of finish_review
  void finish_review_memo(NsMemo* m, int source, HostTable* out);       // finish_review with the $ns rows replayed
  void finish_tail(int source, HostTable* out);
  // paths of the request envelope add_json writes around an object, resolved once
  struct EnvIds { bool ready = false; uint32_t uid, kind, k_group, k_version, k_kind, resource, r_group, r_version, r_resource, operation, user_info, options, name, ns, nsobj,
                  m_sub[2], m_group[2], m_kind[2], m_name[2], m_gname[2], m_nsname[2], m_d[2], m_c[2]; } env_;
  void env_init();
  void fast_match_facts_n(const ObjFacts& f, bool ns_defined, const std::string& ns_name, bool is_old);
  uint32_t fast_child(uint32_t parent, const char* key, uint32_t len);
  uint32_t fast_child_w(uint32_t parent, const char* key, uint32_t len, uint64_t w8);
  uint64_t name8(const char* key, uint32_t len) const;
  int fast_value(uint32_t path, uint32_t ords, int adepth, uint32_t extra, int depth);   // -> RowType of the value, -1 = bail
  bool fast_string(const char** s, uint32_t* n);   // decodes the string at p_ (views the text when it has no escapes)
  bool fast_tree(const char* json, size_t len, uint32_t root, ObjFacts* facts, int* type);
  void emit_str_n(uint32_t path, uint32_t meta, const char* s, uint32_t n);
  // ---- structural index (round 4): stage 1 classifies the document 64 bytes at a time (AVX-512: unescaped quotes, the in-string
  // mask by carry-less multiply, the six structural characters, the first character of every other scalar) into an array of token
  // positions; stage 2 (ix_value / ix_skip: the grammar and the row semantics of fast_value / skip_value) walks TOKENS, not bytes: no
  // white-space loops, a string's length is the distance of two tokens, an unread subtree is a run of tokens checked by a small
  // state machine.  Hosts without AVX-512BW + PCLMUL keep the byte-at-a-time path (GK_NO_INDEX=1 forces it: the tests compare both).
  std::vector<uint32_t> ix_;          // token positions (+ one sentinel: the document's length)
  std::vector<uint64_t> ix_bs_;       // backslash bits per 64-byte block
  const char* ix_json_ = nullptr;
  uint32_t ix_n_ = 0, ixp_ = 0, ix_len_ = 0;
  bool ix_any_bs_ = false;
  bool use_index_ = false;            // decided per table (begin_table): the CPU has the instructions and GK_NO_INDEX is not set
  static bool ix_supported();
  void ix_build(const char* json, size_t len);
  bool ix_has_bs(uint32_t a, uint32_t b) const;            // a backslash in [a, b)?
  bool ix_string(const char** s, uint32_t* n);             // the string whose opening quote is token ixp_; advances past its closing quote
  bool ix_skip_string();
  int ix_scalar(bool emit_rows, uint32_t path, uint32_t meta);   // literal / number at token ixp_: its RowType (rows + dictionary row when asked), -1 = bail
  int ix_value(uint32_t path, uint32_t ords, int adepth, uint32_t extra, int depth);
  int ix_skip(int depth);
  void ws() { if (p_ < e_ && (unsigned char)*p_ > ' ') return; while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) p_++; }
};

}  // namespace gk
