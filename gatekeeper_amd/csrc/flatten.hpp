// Review normalisation + flattening (host).
//
//  * normalize_*: what K8sValidationTarget.HandleReview does before matching/evaluation
//      pkg/target/target.go:81-138 (8 input shapes -> gkReview), :140-179 (unstructuredToAdmissionRequest),
//      :269-287 (setObjectOnDelete, ErrOldObjectIsNil), pkg/target/matcher.go:37-39 (nsCache fallback).
//  * Flattener: key-path -> value SoA rows (plan.hpp Row) + string heap + per-review header with the match-layer
//    facts (RF_*) that pkg/mutation/match/match.go:73-258 derives from object/namespace/source.
#pragma once
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

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

struct HostTable {
  std::vector<Row> rows;            // row groups: per tile, sorted by path (plan.hpp)
  std::vector<StrHdr> shdr;         // parallel to rows
  std::vector<uint32_t> tile_idx;   // [n_tiles][n_slots + 1]
  std::vector<uint32_t> slot_path;  // [n_slots] path id of each slot, increasing
  std::vector<uint32_t> rflags;     // n_reviews
  std::vector<uint8_t> heap;
  std::vector<uint32_t> path_rows;  // rows per path over the whole table (algorithmic-byte accounting per plan)
  std::vector<uint32_t> path_max;   // per array-element path: largest element count of one review (plan specialisation)
  uint32_t n_reviews = 0;
  uint32_t n_tiles() const { return (n_reviews + GK_RPT - 1) / GK_RPT; }   // row groups
  uint32_t n_slots() const { return (uint32_t)slot_path.size(); }
  // build-time only: per-tile segment lists, turned into tile_idx by Flattener::finish
  struct SegRec { uint32_t path, start; };
  std::vector<SegRec> segs;
  std::vector<uint32_t> tile_seg;
  void append(const HostTable& part);
};

class Flattener {
 public:
  explicit Flattener(PathDict* dict);
  void add(const ReviewDoc& doc, HostTable* out);
  void finish(HostTable* out);   // flush + build_index
  void flush(HostTable* out);    // closes the tile being built (parallel table builds flush per part, then append)
  static void build_index(HostTable* out);   // slots + dense [tile][slot] index from the per-tile segment lists

 private:
  PathDict* dict_;
  uint32_t id_object_, id_old_, id_m_, id_ns_;
  struct Ctr { uint32_t path, n; };
  std::vector<Ctr> ctrs_;
  HostTable* t_ = nullptr;
  uint32_t review_flags_ = 0;
  struct Staged { uint32_t path; Row row; StrHdr hdr; };
  std::vector<Staged> stage_;       // rows of the tile being built, review order / document order
  std::vector<uint32_t> order_;
  void flush_tile(HostTable* out);
  std::unordered_map<uint32_t, std::unordered_map<std::string, uint32_t>> memo_;
  std::unordered_map<uint32_t, uint32_t> memo_elem_;
  uint32_t child(uint32_t parent, const std::string& key);
  uint32_t elem(uint32_t parent);
  void walk(const Value& v, uint32_t path, uint32_t meta_ords, int adepth, uint32_t extra);
  void emit(uint32_t path, uint32_t meta, uint32_t lo, uint32_t hi);
  uint32_t put_string(const std::string& s, uint32_t* hash);
  void emit_str(uint32_t parent, const char* key, const std::string& s);
  void emit_string_row(uint32_t path, uint32_t meta, const std::string& s);
  void match_facts(const Value& obj, const Value& ns, bool is_old, uint32_t m_parent);
};

}  // namespace gk
