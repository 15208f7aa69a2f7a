// CDNA4 (gfx950) kernels for the constraint x review cross-product.
//
// gk_eval_tiles  (dominant kernel; HBM-bound, integer/byte work, no MFMA)
//   one 64-lane wave per tile of 64 consecutive reviews.
//   phase 1: the wave streams the tile's rows (16 B each, coalesced dwordx4), looks each row's interned path up in
//            the path table and evaluates the few predicates attached to it; result bits are OR-ed (LDS atomics) into
//            per-review accumulators laid out [word][lane] so phase 2 reads are bank-conflict free.
//   phase 2: lane = review.  A wave-uniform bytecode interpreter (scalar fetch/decode, per-lane boolean registers)
//            evaluates every distinct match / violation formula over the accumulators; wave ballots turn the 64
//            per-lane answers into one bitmap word per constraint, and ballot + popcount prefix compacts the
//            (constraint, review) violation list.
// gk_eval_big    (rare path) one wave per review whose arrays overflow the LDS element capacity; accumulators in HBM.
//
// Replaces the serial loop  for obj { for constraint { Matcher.Match (pkg/target/matcher.go:21) ; Driver.Query } }
// of pkg/audit/manager.go:591-642 / pkg/webhook/policy.go:826.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <stdexcept>
#include <string>

#include "device.hpp"
#include "vm_core.hpp"

namespace gk {

#define HIP_OK(expr)                                                                                          \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); \
  } while (0)

struct Work;
struct DevTable {
  Work* work = nullptr;
  Row* rows = nullptr;
  ReviewHdr* hdrs = nullptr;
  uint8_t* heap = nullptr;
  uint32_t n_reviews = 0, n_rows = 0;
  uint64_t bytes = 0;
};

struct DevPlanVariant {
  uint32_t* ptab = nullptr;
  uint32_t* pred_list = nullptr;
  Pred* preds = nullptr;
  Scope* scopes = nullptr;
  uint32_t* code = nullptr;
  uint8_t* cheap = nullptr;
  PlanDims dims{};
  PlanView view() const { return PlanView{ptab, pred_list, preds, scopes, code, cheap, dims}; }
};

struct DevPlan {
  DevPlanVariant fast, big;
  ConstraintSlot* slots = nullptr;
  uint32_t n_constraints = 0;
};

// ------------------------------------------------------------------------------------------------ accumulators
struct LdsAcc {   // phase 1: review-local lane `rl`; words are [w][64]
  uint32_t* base;
  uint32_t rl;
  __device__ void or_word(uint32_t w, uint32_t m) { atomicOr(&base[w * GK_TILE + rl], m); }
  __device__ void max_word(uint32_t w, uint32_t v) { atomicMax(&base[w * GK_TILE + rl], v); }
  __device__ void store_word(uint32_t w, uint32_t v) { base[w * GK_TILE + rl] = v; }
  __device__ uint32_t load(uint32_t w) const { return base[w * GK_TILE + rl]; }
};
struct GlobalAcc {   // big variant: contiguous words of one review in HBM scratch
  uint32_t* base;
  __device__ void or_word(uint32_t w, uint32_t m) { atomicOr(&base[w], m); }
  __device__ void max_word(uint32_t w, uint32_t v) { atomicMax(&base[w], v); }
  __device__ void store_word(uint32_t w, uint32_t v) { base[w] = v; }
  __device__ uint32_t load(uint32_t w) const { return base[w]; }
};

__device__ inline uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

struct OutPtrs {
  uint64_t* viol;
  uint64_t* err;
  uint64_t* match;       // may be null
  uint64_t* overflow;    // [n_tiles] reviews that must be re-run in the big variant
  uint64_t* too_big;     // [n_tiles] reviews beyond engine limits (reported, never guessed)
  uint32_t* counts;      // [n_constraints]
  uint32_t* list;        // pairs
  uint32_t* list_count;  // [0] = entries wanted, [1] = overflowed-review count
  uint32_t list_capacity;
};

// ------------------------------------------------------------------------------------------------ dominant kernel
__global__ __launch_bounds__(GK_TILE) void gk_eval_tiles(PlanView pv, const Row* __restrict__ rows,
                                                         const ReviewHdr* __restrict__ hdrs, const uint8_t* __restrict__ heap,
                                                         uint32_t n_reviews, uint32_t n_tiles, const ConstraintSlot* __restrict__ slots,
                                                         OutPtrs out) {
  extern __shared__ uint32_t lds[];          // [acc_words][64]
  __shared__ uint32_t s_bounds[GK_MAX_SCOPES];
  const uint32_t lane = threadIdx.x;
  const uint32_t tile = blockIdx.x;
  const uint32_t r0 = tile * GK_TILE;
  const uint32_t nrev = min((uint32_t)GK_TILE, n_reviews - r0);
  const uint32_t acc_words = pv.dims.acc_words;

  for (uint32_t w = lane; w < acc_words * GK_TILE; w += GK_TILE) lds[w] = 0;
  __syncthreads();

  // ---- phase 1: stream the tile's rows
  const uint32_t row_lo = hdrs[r0].row_start;
  const uint32_t row_hi = hdrs[r0 + nrev].row_start;
  uint32_t seen = 0;   // reviews started before this iteration (wave-uniform)
  for (uint32_t base = row_lo; base < row_hi; base += GK_TILE) {
    const uint32_t i = base + lane;
    const bool active = i < row_hi;
    Row r;
    if (active) {
      const uint4 v = reinterpret_cast<const uint4*>(rows)[i];
      r.path = v.x; r.meta = v.y; r.lo = v.z; r.hi = v.w;
    } else { r.path = 0xFFFFFFFFu; r.meta = 0; r.lo = 0; r.hi = 0; }
    const uint64_t firsts = __ballot(active && (r.meta & ROW_FIRST));
    const uint32_t before = __popcll(firsts & ((2ull << lane) - 1ull));   // FIRST flags at lanes <= mine
    const uint32_t rl = seen + before - 1u;
    seen += __popcll(firsts);
    if (active && r.path < pv.dims.n_paths && pv.ptab[r.path] != 0) {
      LdsAcc acc{lds, rl};
      eval_row(r, pv, heap, acc);
    }
  }
  __syncthreads();

  // ---- phase 2: lane = review
  for (uint32_t s = 0; s < pv.dims.n_scopes; s++) {
    uint32_t c = lds[pv.scopes[s].count_off * GK_TILE + lane];
    for (int off = 32; off > 0; off >>= 1) c = max(c, (uint32_t)__shfl_xor((int)c, off));
    if (lane == 0) s_bounds[s] = c;
  }
  __syncthreads();

  const bool live = lane < nrev;
  LdsAcc acc{lds, lane};
  Results res = {0, 0, 0};
  const uint32_t flags = live ? hdrs[r0 + lane].flags : 0u;
  res = eval_formulas(pv, acc, flags, heap, s_bounds);
  const bool too_big = live && (flags & RF_TOO_BIG);
  const bool ovf = live && !too_big && (acc.load(0) & 1u);
  const uint64_t ovf_mask = __ballot(ovf);
  const uint64_t big_mask = __ballot(too_big);
  if (lane == 0) {
    out.overflow[tile] = ovf_mask;
    if (ovf_mask) atomicAdd(&out.list_count[1], (uint32_t)__popcll(ovf_mask));
    if (big_mask) atomicOr((unsigned long long*)&out.too_big[tile], (unsigned long long)big_mask);
  }
  const bool usable = live && !ovf && !too_big;
  const uint32_t nc = pv.dims.n_constraints;
  for (uint32_t c = 0; c < nc; c++) {
    const ConstraintSlot sl = slots[c];
    const bool m = usable && ((res.match >> sl.match) & 1ull);
    const bool e = usable && ((res.err >> sl.match) & 1ull);
    const bool v = m && ((res.viol >> sl.viol) & 1ull);
    const uint64_t vb = __ballot(v), eb = __ballot(e);
    if (out.match) { const uint64_t mb = __ballot(m); if (lane == 0) out.match[(size_t)c * n_tiles + tile] = mb; }
    if (lane == 0) {
      out.viol[(size_t)c * n_tiles + tile] = vb;
      out.err[(size_t)c * n_tiles + tile] = eb;
    }
    if (vb) {
      const uint32_t n = (uint32_t)__popcll(vb);
      uint32_t slot0 = 0;
      if (lane == 0) {
        atomicAdd(&out.counts[c], n);
        if (out.list_capacity) slot0 = atomicAdd(&out.list_count[0], n);
      }
      if (out.list_capacity) {
        slot0 = (uint32_t)__shfl((int)slot0, 0);
        if (v) {
          const uint32_t k = slot0 + (uint32_t)__popcll(vb & ((1ull << lane) - 1ull));
          if (k < out.list_capacity) { out.list[2 * k] = c; out.list[2 * k + 1] = r0 + lane; }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ big variant
__global__ __launch_bounds__(GK_TILE) void gk_eval_big(PlanView pv, const Row* __restrict__ rows, const ReviewHdr* __restrict__ hdrs,
                                                       const uint8_t* __restrict__ heap, const uint32_t* __restrict__ review_ids,
                                                       uint32_t n_list, uint32_t n_tiles, const ConstraintSlot* __restrict__ slots,
                                                       uint32_t* scratch, OutPtrs out) {
  __shared__ uint32_t s_bounds[GK_MAX_SCOPES];
  const uint32_t lane = threadIdx.x;
  const uint32_t r = review_ids[blockIdx.x];
  uint32_t* accw = scratch + (size_t)blockIdx.x * pv.dims.acc_words;
  for (uint32_t w = lane; w < pv.dims.acc_words; w += GK_TILE) accw[w] = 0;
  __threadfence_block();
  __syncthreads();
  GlobalAcc acc{accw};
  const uint32_t row_lo = hdrs[r].row_start, row_hi = hdrs[r + 1].row_start;
  for (uint32_t i = row_lo + lane; i < row_hi; i += GK_TILE) {
    const uint4 v = reinterpret_cast<const uint4*>(rows)[i];
    Row rw{v.x, v.y, v.z, v.w};
    if (rw.path < pv.dims.n_paths && pv.ptab[rw.path] != 0) eval_row(rw, pv, heap, acc);
  }
  __threadfence_block();
  __syncthreads();
  if (lane < pv.dims.n_scopes) s_bounds[lane] = accw[pv.scopes[lane].count_off];
  __syncthreads();
  if (lane != 0) return;
  const uint32_t tile = r / GK_TILE, bit = r % GK_TILE;
  if (acc.load(0) & 1u) {   // still overflowing: report, never guess
    atomicOr((unsigned long long*)&out.too_big[tile], 1ull << bit);
    return;
  }
  Results res = eval_formulas(pv, acc, hdrs[r].flags, heap, s_bounds);
  for (uint32_t c = 0; c < pv.dims.n_constraints; c++) {
    const ConstraintSlot sl = slots[c];
    const bool m = (res.match >> sl.match) & 1ull;
    const bool e = (res.err >> sl.match) & 1ull;
    const bool v = m && ((res.viol >> sl.viol) & 1ull);
    if (out.match && m) atomicOr((unsigned long long*)&out.match[(size_t)c * n_tiles + tile], 1ull << bit);
    if (e) atomicOr((unsigned long long*)&out.err[(size_t)c * n_tiles + tile], 1ull << bit);
    if (v) {
      atomicOr((unsigned long long*)&out.viol[(size_t)c * n_tiles + tile], 1ull << bit);
      atomicAdd(&out.counts[c], 1u);
      if (out.list_capacity) {
        const uint32_t k = atomicAdd(&out.list_count[0], 1u);
        if (k < out.list_capacity) { out.list[2 * k] = c; out.list[2 * k + 1] = r; }
      }
    }
  }
  (void)n_list;
}

// ------------------------------------------------------------------------------------------------ host side
static int g_device = -1;

std::string dev_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return std::string("no HIP device available: ") + hipGetErrorString(e);
  if (device < 0 || device >= n) return "HIP device index out of range";
  e = hipSetDevice(device);
  if (e != hipSuccess) return std::string("hipSetDevice failed: ") + hipGetErrorString(e);
  g_device = device;
  // allow a full CU's LDS per workgroup for large plans
  hipFuncSetAttribute(reinterpret_cast<const void*>(gk_eval_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  return "";
}

int dev_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

template <class T>
static T* upload(const T* src, size_t n) {
  T* d = nullptr;
  size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  HIP_OK(hipMalloc(&d, bytes));
  if (n) HIP_OK(hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

DevTable* dev_table_upload(const HostTable& t) {
  DevTable* d = new DevTable();
  d->n_reviews = t.n_reviews;
  d->n_rows = (uint32_t)t.rows.size();
  d->rows = upload(t.rows.data(), t.rows.size());
  d->hdrs = upload(t.hdrs.data(), t.hdrs.size());
  // 16 B of slack so byte loops may over-read safely
  std::vector<uint8_t> heap = t.heap;
  heap.resize(heap.size() + 16, 0);
  d->heap = upload(heap.data(), heap.size());
  d->bytes = t.rows.size() * sizeof(Row) + t.hdrs.size() * sizeof(ReviewHdr) + heap.size();
  return d;
}

void dev_table_free_work(DevTable* t);
void dev_table_free(DevTable* t) {
  if (!t) return;
  dev_table_free_work(t);
  hipFree(t->rows); hipFree(t->hdrs); hipFree(t->heap);
  delete t;
}
uint64_t dev_table_bytes(const DevTable* t) { return t->bytes; }

static void upload_variant(const HostPlan& h, DevPlanVariant* v) {
  v->ptab = upload(h.ptab.data(), h.ptab.size());
  v->pred_list = upload(h.pred_list.data(), h.pred_list.size());
  v->preds = upload(h.preds.data(), h.preds.size());
  v->scopes = upload(h.scopes.data(), h.scopes.size());
  v->code = upload(h.code.data(), h.code.size());
  v->cheap = upload(h.cheap.data(), h.cheap.size());
  v->dims = h.dims;
}
static void free_variant(DevPlanVariant* v) {
  hipFree(v->ptab); hipFree(v->pred_list); hipFree(v->preds); hipFree(v->scopes); hipFree(v->code); hipFree(v->cheap);
}

DevPlan* dev_plan_upload(const HostPlan& fast, const HostPlan& big) {
  DevPlan* p = new DevPlan();
  upload_variant(fast, &p->fast);
  upload_variant(big, &p->big);
  p->slots = upload(fast.slots.data(), fast.slots.size());
  p->n_constraints = (uint32_t)fast.slots.size();
  size_t lds = (size_t)fast.dims.acc_words * GK_TILE * 4;
  if (lds > 160 * 1024 - 256) {
    dev_plan_free(p);
    throw Unsupported("plan needs " + std::to_string(lds) + " B of LDS per tile (limit 160 KiB): lower the element capacities");
  }
  return p;
}

void dev_plan_free(DevPlan* p) {
  if (!p) return;
  free_variant(&p->fast); free_variant(&p->big);
  hipFree(p->slots);
  delete p;
}

// Per-table workspace: output buffers and timing events are allocated once and reused by every launch.
struct Work {
  uint32_t nc = 0, n_tiles = 0, list_cap = 0;
  bool has_match = false;
  uint64_t *d_viol = nullptr, *d_err = nullptr, *d_match = nullptr, *d_ovf = nullptr, *d_big = nullptr;
  uint32_t *d_counts = nullptr, *d_list = nullptr, *d_lc = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;   // one pair per launch since the last finish
  size_t pending = 0;
  void release() {
    hipFree(d_viol); hipFree(d_err); hipFree(d_match); hipFree(d_ovf); hipFree(d_big); hipFree(d_counts); hipFree(d_list); hipFree(d_lc);
    d_viol = d_err = d_match = d_ovf = d_big = nullptr; d_counts = d_list = d_lc = nullptr;
  }
  ~Work() { release(); for (auto& e : evs) { hipEventDestroy(e.first); hipEventDestroy(e.second); } }
};

static Work* get_work(const DevTable* t, uint32_t nc, uint32_t n_tiles, const EvalOptions& opt) {
  DevTable* mt = const_cast<DevTable*>(t);
  if (!mt->work) mt->work = new Work();
  Work* w = mt->work;
  bool want_match = opt.want_match;
  if (w->nc != nc || w->n_tiles != n_tiles || w->list_cap < opt.list_capacity || (want_match && !w->has_match)) {
    w->release();
    const size_t bm = std::max<size_t>((size_t)nc * n_tiles, 1);
    HIP_OK(hipMalloc(&w->d_viol, bm * 8));
    HIP_OK(hipMalloc(&w->d_err, bm * 8));
    if (want_match) HIP_OK(hipMalloc(&w->d_match, bm * 8));
    HIP_OK(hipMalloc(&w->d_ovf, std::max<size_t>(n_tiles, 1) * 8));
    HIP_OK(hipMalloc(&w->d_big, std::max<size_t>(n_tiles, 1) * 8));
    HIP_OK(hipMalloc(&w->d_counts, std::max<size_t>(nc, 1) * 4));
    HIP_OK(hipMalloc(&w->d_lc, 8));
    if (opt.list_capacity) HIP_OK(hipMalloc(&w->d_list, (size_t)opt.list_capacity * 8));
    w->nc = nc; w->n_tiles = n_tiles; w->list_cap = opt.list_capacity; w->has_match = want_match;
  }
  return w;
}

void dev_table_free_work(DevTable* t) { delete t->work; t->work = nullptr; }

void dev_eval_launch(const DevPlan* p, const DevTable* t, const EvalOptions& opt) {
  const uint32_t n = t->n_reviews, nc = p->n_constraints;
  const uint32_t n_tiles = (n + GK_TILE - 1) / GK_TILE;
  if (n == 0 || nc == 0) return;
  Work* w = get_work(t, nc, n_tiles, opt);
  hipStream_t stream = 0;
  HIP_OK(hipMemsetAsync(w->d_counts, 0, (size_t)nc * 4, stream));
  HIP_OK(hipMemsetAsync(w->d_lc, 0, 8, stream));
  HIP_OK(hipMemsetAsync(w->d_big, 0, (size_t)n_tiles * 8, stream));
  OutPtrs out{w->d_viol, w->d_err, opt.want_match ? w->d_match : nullptr, w->d_ovf, w->d_big, w->d_counts, w->d_list, w->d_lc,
              opt.list_capacity};
  if (w->pending == w->evs.size()) {
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
    w->evs.emplace_back(a, b);
  }
  auto& ev = w->evs[w->pending++];
  const size_t lds = (size_t)p->fast.dims.acc_words * GK_TILE * 4;
  HIP_OK(hipEventRecord(ev.first, stream));
  hipLaunchKernelGGL(gk_eval_tiles, dim3(n_tiles), dim3(GK_TILE), lds, stream, p->fast.view(), t->rows, t->hdrs, t->heap, n, n_tiles,
                     p->slots, out);
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(ev.second, stream));
}

void dev_eval_finish(const DevPlan* p, const DevTable* t, const EvalOptions& opt, EvalOut* o) {
  const uint32_t n = t->n_reviews, nc = p->n_constraints;
  const uint32_t n_tiles = (n + GK_TILE - 1) / GK_TILE;
  o->n_reviews = n; o->n_constraints = nc; o->n_tiles = n_tiles;
  o->kernel_ms = o->fast_kernel_ms = 0;
  o->n_overflow = 0; o->list_total = 0; o->n_launches = 0;
  if (n == 0 || nc == 0) {
    o->viol.assign((size_t)nc * n_tiles, 0); o->err = o->viol; o->match = o->viol; o->too_big.assign(n_tiles, 0); o->counts.assign(nc, 0); o->list.clear();
    return;
  }
  Work* w = t->work;
  if (!w || w->pending == 0) throw std::runtime_error("dev_eval_finish without a pending launch");
  hipStream_t stream = 0;
  const size_t bm_words = (size_t)nc * n_tiles;
  uint32_t lc[2] = {0, 0};
  HIP_OK(hipMemcpyAsync(lc, w->d_lc, 8, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  float total = 0;
  for (size_t i = 0; i < w->pending; i++) { float ms = 0; HIP_OK(hipEventElapsedTime(&ms, w->evs[i].first, w->evs[i].second)); total += ms; }
  o->n_launches = (uint32_t)w->pending;
  o->fast_kernel_ms = total / (float)w->pending;   // average duration of the dominant kernel per launch
  o->kernel_ms = o->fast_kernel_ms;
  w->pending = 0;
  o->n_overflow = lc[1];
  OutPtrs out{w->d_viol, w->d_err, opt.want_match ? w->d_match : nullptr, w->d_ovf, w->d_big, w->d_counts, w->d_list, w->d_lc,
              opt.list_capacity};
  if (lc[1]) {   // rare: reviews whose arrays exceed the LDS element capacity
    std::vector<uint64_t> ovf(n_tiles);
    HIP_OK(hipMemcpy(ovf.data(), w->d_ovf, (size_t)n_tiles * 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> ids;
    for (uint32_t tl = 0; tl < n_tiles; tl++) for (uint64_t m = ovf[tl]; m; m &= m - 1) ids.push_back(tl * GK_TILE + (uint32_t)__builtin_ctzll(m));
    uint32_t* d_ids = upload(ids.data(), ids.size());
    uint32_t* d_scratch;
    HIP_OK(hipMalloc(&d_scratch, ids.size() * (size_t)p->big.dims.acc_words * 4));
    hipEvent_t e1, e2;
    HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&e2));
    HIP_OK(hipEventRecord(e1, stream));
    hipLaunchKernelGGL(gk_eval_big, dim3((uint32_t)ids.size()), dim3(GK_TILE), 0, stream, p->big.view(), t->rows, t->hdrs, t->heap, d_ids,
                       (uint32_t)ids.size(), n_tiles, p->slots, d_scratch, out);
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(e2, stream));
    HIP_OK(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e1, e2));
    o->kernel_ms += ms;
    hipEventDestroy(e1); hipEventDestroy(e2);
    hipFree(d_ids); hipFree(d_scratch);
    HIP_OK(hipMemcpy(lc, w->d_lc, 8, hipMemcpyDeviceToHost));
  }
  o->list_total = lc[0];
  o->d_viol = w->d_viol; o->d_err = w->d_err; o->d_counts = w->d_counts;
  if (opt.download) {
    o->viol.resize(bm_words); o->err.resize(bm_words); o->counts.resize(nc); o->too_big.resize(n_tiles);
    HIP_OK(hipMemcpy(o->viol.data(), w->d_viol, bm_words * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(o->err.data(), w->d_err, bm_words * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(o->counts.data(), w->d_counts, (size_t)nc * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(o->too_big.data(), w->d_big, (size_t)n_tiles * 8, hipMemcpyDeviceToHost));
    if (opt.want_match) { o->match.resize(bm_words); HIP_OK(hipMemcpy(o->match.data(), w->d_match, bm_words * 8, hipMemcpyDeviceToHost)); }
    if (opt.list_capacity) {
      uint32_t k = std::min(lc[0], opt.list_capacity);
      o->list.resize((size_t)k * 2);
      if (k) HIP_OK(hipMemcpy(o->list.data(), w->d_list, (size_t)k * 8, hipMemcpyDeviceToHost));
    }
  }
}

void dev_eval(const DevPlan* p, const DevTable* t, const EvalOptions& opt, EvalOut* o) {
  dev_eval_launch(p, t, opt);
  dev_eval_finish(p, t, opt, o);
}

}  // namespace gk
