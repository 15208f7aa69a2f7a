// See flatten.hpp.
#include "flatten.hpp"

#include <algorithm>

#include <mutex>

namespace gk {

// ------------------------------------------------------------------------------------------------ PathDict
size_t PathDict::KeyHash::operator()(const std::pair<uint32_t, std::string>& k) const {
  return std::hash<std::string>()(k.second) * 1000003u ^ (size_t)k.first * 0x9E3779B97F4A7C15ull;
}

PathDict::PathDict() { infos_.push_back({kNone, "", false, 0}); }

uint32_t PathDict::intern(uint32_t parent, const std::string& key, bool is_elem) {
  std::pair<uint32_t, std::string> k(parent, is_elem ? std::string("\x01[]") : key);
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
  // FNV-1a with a murmur-style finalizer; only ever compared for equality as a fast reject (bytes decide).
  uint32_t h = 2166136261u;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
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

Value str_field(const Value& o, const char* k) {
  const Value* v = o.get(k);
  return (v && v->is_string()) ? *v : S("");
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
  p.emplace_back(S("uid"), str_field(request, "uid"));
  p.emplace_back(S("kind"), triple(request.get("kind"), "group", "version", "kind"));
  p.emplace_back(S("resource"), triple(request.get("resource"), "group", "version", "resource"));
  Value op = str_field(request, "operation");
  p.emplace_back(S("operation"), op);
  const Value* ui = request.get("userInfo");
  p.emplace_back(S("userInfo"), (ui && ui->is_object()) ? *ui : Value::object({}));
  const Value* obj = request.get("object");
  const Value* old = request.get("oldObject");
  Value vobj = (obj && obj->is_object()) ? *obj : Value::null();
  Value vold = (old && old->is_object()) ? *old : Value::null();
  if (op.str() == "DELETE") {   // setObjectOnDelete
    if (!vold.is_object()) throw ReviewError("oldObject cannot be nil for DELETE operations");
    vobj = vold;
  }
  p.emplace_back(S("object"), vobj);
  p.emplace_back(S("oldObject"), vold);
  const Value* opts = request.get("options");
  p.emplace_back(S("options"), opts ? *opts : Value::null());
  for (const char* k : {"subResource", "requestSubResource", "name", "namespace"}) {
    const Value* v = request.get(k);
    if (v && v->is_string() && !v->str().empty()) p.emplace_back(S(k), *v);
  }
  for (const char* k : {"requestKind", "requestResource", "dryRun"}) {
    const Value* v = request.get(k);
    if (v && !v->is_null()) p.emplace_back(S(k), *v);
  }
  if (ns_object.defined() && !ns_object.is_null()) p.emplace_back(S("namespaceObject"), ns_object);
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
  ValuePairs kind{{S("group"), S(g)}, {S("version"), S(v)}, {S("kind"), S(k)}};
  ValuePairs req;
  req.emplace_back(S("kind"), Value::object(std::move(kind)));
  req.emplace_back(S("name"), S(obj_string(object, "metadata", "name")));
  req.emplace_back(S("namespace"), S(obj_string(object, "metadata", "namespace")));
  if (!operation.empty()) req.emplace_back(S("operation"), S(operation));
  if (operation == "DELETE") req.emplace_back(S("oldObject"), object);   // target.go:151-154
  else req.emplace_back(S("object"), object);
  return normalize_admission_request(Value::object(std::move(req)), match_ns, ns_object, source, cache);
}

// ------------------------------------------------------------------------------------------------ Flattener
Flattener::Flattener(PathDict* dict) : dict_(dict) {
  id_object_ = dict_->child(0, "object");
  id_old_ = dict_->child(0, "oldObject");
  id_m_ = dict_->child(0, "$m");
  id_ns_ = dict_->child(0, "$ns");
}

void Flattener::emit(uint32_t path, uint32_t meta, uint32_t lo, uint32_t hi) {
  stage_.push_back({path, Row{t_->n_reviews % GK_RPT, meta, lo, hi}, StrHdr{{0, 0, 0, 0}}});
}

uint32_t Flattener::put_string(const std::string& s, uint32_t* hash) {
  // 16-byte aligned, zero-padded entry [u32 len][bytes][pad]: one aligned 16 B load fetches len + the first 12 bytes
  std::vector<uint8_t>& h = t_->heap;
  uint32_t n = (uint32_t)s.size();
  size_t at = (h.size() + 15) & ~(size_t)15;
  h.resize(at + ((4 + (size_t)n + 15) & ~(size_t)15));
  memcpy(&h[at], &n, 4);
  memcpy(&h[at + 4], s.data(), n);
  *hash = hash32((const uint8_t*)s.data(), n);
  return (uint32_t)(at + 4);
}

void Flattener::emit_string_row(uint32_t path, uint32_t meta, const std::string& s) {
  if (s.size() <= 7) {   // inline: no heap entry, no memory access on the device
    uint64_t bits = 0;
    memcpy(&bits, s.data(), s.size());
    emit(path, meta | T_STRING | ROW_STR_INLINE, (uint32_t)bits, (uint32_t)(bits >> 32) | ((uint32_t)s.size() << 24));
    return;
  }
  uint32_t hsh, off = put_string(s, &hsh);
  emit(path, meta | T_STRING, off, hsh);
  memcpy(&stage_.back().hdr, &t_->heap[off - 4], 16);   // entry header: length + first 12 bytes
}

void Flattener::emit_str(uint32_t parent, const char* key, const std::string& s) { emit_string_row(child(parent, key), 0, s); }

// Flattener-local memo of the shared dictionary: the hot lookups take no lock (one Flattener per host thread).
uint32_t Flattener::child(uint32_t parent, const std::string& key) {
  auto& m = memo_[parent];
  auto it = m.find(key);
  if (it != m.end()) return it->second;
  uint32_t id = dict_->child(parent, key);
  m.emplace(key, id);
  return id;
}
uint32_t Flattener::elem(uint32_t parent) {
  auto it = memo_elem_.find(parent);
  if (it != memo_elem_.end()) return it->second;
  uint32_t id = dict_->elem(parent);
  memo_elem_.emplace(parent, id);
  return id;
}

void Flattener::walk(const Value& v, uint32_t path, uint32_t ords, int adepth, uint32_t extra) {
  uint32_t meta = ords | extra;
  switch (v.kind) {
    case Value::Null: emit(path, meta | T_NULL, 0, 0); break;
    case Value::Bool: emit(path, meta | T_BOOL, v.b ? 1 : 0, 0); break;
    case Value::Number:
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
    case Value::String: emit_string_row(path, meta, v.str()); break;
    case Value::Object: {
      emit(path, meta | T_OBJECT, (uint32_t)v.size(), 0);
      for (const auto& kv : v.pairs()) walk(kv.second, child(path, kv.first.str()), ords, adepth, extra);
      break;
    }
    case Value::Array: case Value::Set: {
      emit(path, meta | T_ARRAY, (uint32_t)v.size(), 0);
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
        walk(e, ep, o2, adepth + 1, ex);
      }
      break;
    }
    default: break;
  }
}

void Flattener::match_facts(const Value& obj, const Value& ns, bool is_old, uint32_t m_parent) {
  // facts of pkg/mutation/match/match.go needed per candidate object (object, then oldObject; matcher.go:44-71)
  std::string g, ver, k;
  obj_gvk(obj, &g, &ver, &k);
  bool is_ns = (k == "Namespace" && g.empty());
  std::string name = obj_string(obj, "metadata", "name");
  std::string nsfield = obj_string(obj, "metadata", "namespace");
  uint32_t sub = dict_->child(m_parent, is_old ? "old" : "o");
  emit(sub, T_OBJECT, 0, 0);
  emit_str(sub, "group", g);
  emit_str(sub, "kind", k);
  emit_str(sub, "name", name);
  emit_str(sub, "gname", obj_string(obj, "metadata", "generateName"));
  bool has_nsname = true;
  std::string nsname;
  if (is_ns) nsname = name;
  else if (ns.defined()) nsname = obj_string(ns, "metadata", "name");
  else if (!nsfield.empty()) nsname = nsfield;
  else has_nsname = false;
  if (has_nsname) emit_str(sub, "nsname", nsname);
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
  ctrs_.clear();
  review_flags_ = 0;
  const Value& req = doc.request;
  // root + request members (input.review.*)
  emit(0, T_OBJECT, (uint32_t)req.size(), 0);
  for (const auto& kv : req.pairs()) walk(kv.second, child(0, kv.first.str()), 0, 0, 0);
  // $ns: only what the match layer reads from Matchable.Namespace (name + labels)
  const Value& ns = doc.match_ns;
  if (ns.defined()) {
    review_flags_ |= RF_NS_PRESENT;
    emit(id_ns_, T_OBJECT, 1, 0);
    uint32_t md = dict_->child(id_ns_, "metadata");
    emit(md, T_OBJECT, 2, 0);
    emit_str(md, "name", obj_string(ns, "metadata", "name"));
    const Value* m = ns.get("metadata");
    const Value* lb = m ? m->get("labels") : nullptr;
    if (lb && lb->is_object()) {
      bool bad = false;
      for (const auto& kv : lb->pairs()) if (!kv.second.is_string()) bad = true;
      if (bad) review_flags_ |= RF_NS_LABELS_BAD;
      walk(*lb, dict_->child(md, "labels"), 0, 0, 0);
    } else if (lb) review_flags_ |= RF_NS_LABELS_BAD;
  }
  // $m: per-candidate match facts
  emit(id_m_, T_OBJECT, 2, 0);
  const Value* obj = req.get("object");
  const Value* old = req.get("oldObject");
  if (obj && obj->is_object()) match_facts(*obj, ns, false, id_m_);
  if (old && old->is_object()) match_facts(*old, ns, true, id_m_);
  // gkReviewToObject (matcher.go:73-93): Unstructured.UnmarshalJSON rejects a document without a non-empty string `kind`
  if (obj && obj->is_object() && obj_string(*obj, "kind").empty()) review_flags_ |= RF_OBJ_BAD;
  if (old && old->is_object() && obj_string(*old, "kind").empty()) review_flags_ |= RF_OLD_BAD;
  switch (doc.source) {
    case SRC_ORIGINAL: review_flags_ |= RF_SRC_ORIGINAL; break;
    case SRC_GENERATED: review_flags_ |= RF_SRC_GENERATED; break;
    case SRC_ALL: review_flags_ |= RF_SRC_ALL; break;
    case SRC_INVALID: review_flags_ |= RF_SRC_INVALID; break;
    default: break;
  }
  out->rflags.push_back(review_flags_);
  for (const Ctr& c : ctrs_) {
    if (c.path >= out->path_max.size()) out->path_max.resize(c.path + 1, 0);
    out->path_max[c.path] = std::max(out->path_max[c.path], c.n);
  }
  out->n_reviews++;
  if (out->n_reviews % GK_RPT == 0) flush_tile(out);
}

// Close the current tile: stable sort of its rows by path (keeps review order, then document order, inside a
// segment) and one segment record per distinct path (turned into the slot index by build_index).
void Flattener::flush_tile(HostTable* out) {
  out->tile_seg.push_back((uint32_t)out->segs.size());
  order_.resize(stage_.size());
  for (uint32_t i = 0; i < order_.size(); i++) order_[i] = i;
  std::stable_sort(order_.begin(), order_.end(), [&](uint32_t a, uint32_t b) { return stage_[a].path < stage_[b].path; });
  uint32_t prev = PathDict::kNone;
  for (uint32_t i : order_) {
    const Staged& s = stage_[i];
    if (s.path != prev) { out->segs.push_back({s.path, (uint32_t)out->rows.size()}); prev = s.path; }
    if (s.path >= out->path_rows.size()) out->path_rows.resize(s.path + 1, 0);
    out->path_rows[s.path]++;
    out->rows.push_back(s.row);
    out->shdr.push_back(s.hdr);
  }
  stage_.clear();
}

void Flattener::flush(HostTable* out) {
  if (!stage_.empty() || out->n_reviews % GK_RPT != 0) flush_tile(out);
}

// Appends `part` (whole tiles flattened by another Flattener over the same dictionary; the receiving table must end
// on a tile boundary) -- row starts and heap offsets are relocated.
void HostTable::append(const HostTable& part) {
  const uint32_t row_base = (uint32_t)rows.size(), heap_base = (uint32_t)heap.size(), seg_base = (uint32_t)segs.size();
  rows.insert(rows.end(), part.rows.begin(), part.rows.end());
  for (size_t i = row_base; i < rows.size(); i++)
    if ((rows[i].meta & ROW_TYPE_MASK) == T_STRING && !(rows[i].meta & ROW_STR_INLINE)) rows[i].lo += heap_base;
  shdr.insert(shdr.end(), part.shdr.begin(), part.shdr.end());
  heap.insert(heap.end(), part.heap.begin(), part.heap.end());
  for (const SegRec& s : part.segs) segs.push_back({s.path, s.start + row_base});
  for (uint32_t ts : part.tile_seg) tile_seg.push_back(ts + seg_base);
  rflags.insert(rflags.end(), part.rflags.begin(), part.rflags.end());
  if (part.path_rows.size() > path_rows.size()) path_rows.resize(part.path_rows.size(), 0);
  for (size_t i = 0; i < part.path_rows.size(); i++) path_rows[i] += part.path_rows[i];
  if (part.path_max.size() > path_max.size()) path_max.resize(part.path_max.size(), 0);
  for (size_t i = 0; i < part.path_max.size(); i++) path_max[i] = std::max(path_max[i], part.path_max[i]);
  n_reviews += part.n_reviews;
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
    const uint32_t tile_end = s1 < out->segs.size() ? out->segs[s1].start : (uint32_t)out->rows.size();
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
