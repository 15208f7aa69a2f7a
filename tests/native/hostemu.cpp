// TEST-ONLY CPU implementation of device.hpp (see vm_core.hpp).  Built into libgkgpu_hostemu.so, which exists so the
// AOT compiler, the flattener and the formula VM can be checked against the oracle in the GPU-less build container.
// It executes exactly the same per-row / per-review code as kernels.hip, lane by lane.  The product library
// (libgkgpu.so) never contains this file; gatekeeper_amd/_lib.py refuses to load it outside tests.
#include <array>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include <dlfcn.h>
#include <unistd.h>

#include <fstream>
#include <sstream>
#include <map>

#include "codegen.hpp"
#include "device.hpp"
#include "jit_source.hpp"
#include "shard_pipe.hpp"
#include <memory>
#include <mutex>
#include "jit_sources.inc"   // kPlanHpp, kVmCoreHpp, kKernelBody (the text the product embeds)
#include "vm_core.hpp"

namespace gk {

struct DevTable { HostTable t; int pending = 0; std::vector<uint64_t> last_viol, last_err, last_big; std::vector<uint8_t> shard_all; size_t shard_slot = 0; uint32_t shard_stride = 0, shard_nc = 0; float exchange_ms = 0; int exchange_world = 0; };
typedef void (*HeRowFn)(const Row*, uint32_t, uint32_t, const StrHdr*, const PlanView*, const uint8_t*, std::vector<uint32_t>*);
typedef void (*HeFormFn)(const PlanView*, std::vector<uint32_t>*, uint32_t, const Row*, const uint8_t*, const uint32_t*, Results*);
struct DevPlan {
  HostPlan fast, big;
  // GK_HOSTEMU_JIT=1: the plan-specialised source (codegen.cpp, the text hiprtc compiles on the GPU) built with g++
  void* dl = nullptr;
  HeRowFn row = nullptr;
  HeFormFn form = nullptr;
  std::vector<uint32_t> cls;
  std::map<std::string, std::pair<void*, void*>> emu_jit;   // GK_HOSTEMU_KERNEL=jit: geometry -> (dl handle, launcher) of the emulated plan-specialised kernel
};

struct VecAcc {
  std::vector<uint32_t>* w;
  void or_word(uint32_t i, uint32_t m) { (*w)[i] |= m; }
  void max_word(uint32_t i, uint32_t v) { if ((*w)[i] < v) (*w)[i] = v; }
  void store_word(uint32_t i, uint32_t v) { (*w)[i] = v; }
  uint32_t load(uint32_t i) const { return (*w)[i]; }
};

std::string dev_init(int) { return ""; }
int dev_count() { return 0; }
struct DevPart { HostTable t; };
DevPart* dev_part_upload(int, HostTable& part) { DevPart* p = new DevPart(); p->t.rows = std::move(part.rows); p->t.shdr = std::move(part.shdr); p->t.heap = std::move(part.heap); return p; }
void dev_part_free(DevPart* p) { delete p; }
DevTable* dev_table_assemble(int, std::vector<DevPart*>& parts, const HostTable& meta) {
  DevTable* d = new DevTable();
  d->t = meta;
  for (DevPart* p : parts) {
    const uint32_t row_base = (uint32_t)d->t.rows.size(), heap_base = (uint32_t)d->t.heap.size();
    d->t.rows.append(p->t.rows.data(), p->t.rows.size());
    for (size_t i = row_base; i < d->t.rows.size(); i++)
      if ((d->t.rows[i].meta & ROW_TYPE_MASK) == T_STRING && !(d->t.rows[i].meta & ROW_STR_INLINE)) d->t.rows[i].lo += heap_base;
    d->t.shdr.append(p->t.shdr.data(), p->t.shdr.size());
    d->t.heap.append(p->t.heap.data(), p->t.heap.size());
    delete p;
  }
  parts.clear();
  d->t.heap.resize_zero(d->t.heap.size() + 16);
  return d;
}
void hostemu_table_gone(const DevTable* t);
void dev_table_free(DevTable* t) { hostemu_table_gone(t); delete t; }
DevTable* dev_table_view(DevTable* base) { DevTable* v = new DevTable(*base); v->pending = 0; v->last_viol.clear(); return v; }
uint64_t dev_table_bytes(const DevTable* t) { return t->t.rows.size() * 32 + t->t.tile_idx.size() * 4 + t->t.heap.size(); }
DevPlan* dev_plan_upload(int, const HostPlan& fast, const HostPlan& big) {
  DevPlan* p = new DevPlan();
  p->fast = fast; p->big = big;
  if (getenv("GK_HOSTEMU_JIT")) {
    static int counter = 0;
    std::string base = "/tmp/gkjit_hostemu_" + std::to_string(getpid()) + "_" + std::to_string(counter++);
    std::ostringstream text;
    {
      std::ostringstream& f = text;
      f << "#include <vector>\n#include \"" << GK_CSRC_DIR << "/vm_core.hpp\"\n" << generate_plan_source(fast)
        << "struct VecAcc { std::vector<uint32_t>* w; void or_word(uint32_t i, uint32_t m) { (*w)[i] |= m; } void max_word(uint32_t i, uint32_t v) { if ((*w)[i] < v) (*w)[i] = v; }\n"
           "  void store_word(uint32_t i, uint32_t v) { (*w)[i] = v; } uint32_t load(uint32_t i) const { return (*w)[i]; } };\n"
           "extern \"C\" void gk_he_row(const gk::Row* r, uint32_t i, uint32_t cls, const gk::StrHdr* h, const gk::PlanView* pv, const uint8_t* heap, std::vector<uint32_t>* w) { VecAcc acc{w}; (void)i; (void)pv; gk::jit_row(*r, cls, *h, heap, acc, true); }\n"
           "extern \"C\" void gk_he_form(const gk::PlanView* pv, std::vector<uint32_t>* w, uint32_t flags, const gk::Row* rows, const uint8_t* heap, const uint32_t* bounds, gk::Results* out) {\n"
           "  VecAcc acc{w}; std::vector<uint32_t> w2 = *w; VecAcc acc2{&w2};\n"
           "  gk::Results mono = gk::jit_formulas(*pv, acc2, flags, rows, heap, bounds);\n"
           "  gk::Results r = {};\n"
           "  for (uint32_t st = 0; st < gk::GK_N_STAGES; st++) for (uint32_t wv = 0; wv < gk::GK_GEN_PARTS; wv++) gk::jit_formula_part(st * gk::GK_GEN_PARTS + (gk::GK_GEN_PARTS - 1 - wv), acc, flags, heap, bounds, r, nullptr);\n"
           "  bool same = r.match == mono.match && r.err == mono.err; for (int k = 0; k < gk::GK_VIOL_WORDS; k++) same = same && r.viol[k] == mono.viol[k];\n"
           "  if (!same) { for (int k = 0; k < gk::GK_VIOL_WORDS; k++) r.viol[k] = ~0ull; r.match = ~0ull; r.err = ~0ull; }   // staged and monolithic code must agree\n"
           "  *out = r; }\n";
    }
    // The same policy sets are loaded by many tests: the g++ run (about a second) is skipped when this exact text -- and the
    // header it includes -- was built before (a cache under /tmp keyed by their hash; written under a private name, then renamed)
    const std::string src = text.str();
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](const std::string& t) { for (unsigned char ch : t) { hsh ^= ch; hsh *= 1099511628211ull; } };
    mix(src);
    { std::ifstream hdr(std::string(GK_CSRC_DIR) + "/vm_core.hpp"); std::stringstream hs; hs << hdr.rdbuf(); mix(hs.str()); }
    { std::ifstream hdr(std::string(GK_CSRC_DIR) + "/plan.hpp"); std::stringstream hs; hs << hdr.rdbuf(); mix(hs.str()); }
    char hx[32];
    snprintf(hx, sizeof hx, "%016llx", (unsigned long long)hsh);
    const std::string cached = std::string("/tmp/gk_hostemu_cache/plan_") + hx + ".so";
    if (!getenv("GK_HOSTEMU_NO_CACHE")) p->dl = dlopen(cached.c_str(), RTLD_NOW);
    if (!p->dl) {
      { std::ofstream f(base + ".cpp"); f << src; }
      std::string cmd = "g++ -std=c++17 -O1 -shared -fPIC -o " + base + ".so " + base + ".cpp 2> " + base + ".log";
      if (system(cmd.c_str()) != 0) throw std::runtime_error("hostemu: generated plan source does not compile, see " + base + ".log");
      p->dl = dlopen((base + ".so").c_str(), RTLD_NOW);
      if (!p->dl) throw std::runtime_error(std::string("hostemu: dlopen failed: ") + dlerror());
      if (!getenv("GK_HOSTEMU_NO_CACHE")) { if (system("mkdir -p /tmp/gk_hostemu_cache") == 0) (void)rename((base + ".so").c_str(), cached.c_str()); }
    }
    p->row = (HeRowFn)dlsym(p->dl, "gk_he_row");
    p->form = (HeFormFn)dlsym(p->dl, "gk_he_form");
    std::vector<std::vector<Pred>> classes;
    p->cls = jit_path_classes(fast, &classes);
    if (const char* keep = getenv("GK_HOSTEMU_KEEP_SRC")) { static int n_kept = 0; std::ofstream f(std::string(keep) + "/plan_" + std::to_string(getpid()) + "_" + std::to_string(n_kept++) + ".cpp"); f << src; }   // (debug aid)
    unlink((base + ".cpp").c_str()); unlink((base + ".so").c_str()); unlink((base + ".log").c_str());
  }
  return p;
}
void dev_plan_no_jit(DevPlan*) {}
void dev_plan_free(DevPlan* p) {
  if (p && p->dl) dlclose(p->dl);
  // (the emulated plan-specialised kernels stay loaded: test-only, and the emulator's state is shared between objects)
  delete p;
}

static PlanView view_of(const HostPlan& h) {
  return PlanView{h.ptab.data(), h.path_preds.data(), h.scopes.data(), h.code.data(), h.cheap.data(), h.dims};
}

static bool eval_review(const HostPlan& hp, const HostTable& t, uint32_t r, Results* res, const DevPlan* jit = nullptr) {
  PlanView pv = view_of(hp);
  std::vector<uint32_t> words(hp.dims.acc_words, 0);
  VecAcc acc{&words};
  const uint32_t tile = r / t.rpt, rl = r % t.rpt;
  const uint32_t S = t.n_slots();
  const uint32_t* ix = &t.tile_idx[(size_t)tile * (S + 1)];
  for (uint32_t s = 0; s < S; s++) {
    const uint32_t path = t.slot_path[s];
    if (path >= pv.dims.n_paths || pv.ptab[path] == 0) continue;
    for (uint32_t i = ix[s]; i < ix[s + 1]; i++) {
      if ((t.rows[i].rev & ROW_REV_MASK) != rl) continue;
      if (jit) {
        uint32_t c = path < jit->cls.size() ? jit->cls[path] : 0;
        // like the device: the string header is only fetched for classes flagged as reading string bytes
        StrHdr h = (c & GK_ENT_NEEDS_STR) ? t.shdr[i] : StrHdr{{0, 0, 0, 0}};
        if (c) jit->row(&t.rows[i], i, c, &h, &pv, t.heap.data(), &words);
      } else eval_row_ent(t.rows[i], i, pv.ptab[path], t.shdr[i], pv, t.heap.data(), acc);
    }
  }
  if (words[0] & 1u) return false;   // overflow
  uint32_t bounds[GK_MAX_SCOPES] = {0};
  for (uint32_t s = 0; s < hp.dims.n_scopes; s++) bounds[s] = words[hp.scopes[s].count_off];
  if (jit) jit->form(&pv, &words, t.rflags[r], t.rows.data(), t.heap.data(), bounds, res);
  else *res = eval_formulas(pv, acc, t.rflags[r], t.rows.data(), t.heap.data(), bounds);
  return true;
}

void dev_topk(const DevTable* t, uint32_t nc, const std::vector<uint32_t>& order, const std::vector<uint32_t>& grp, uint32_t k, uint32_t cap,
              std::vector<uint32_t>* idx, std::vector<uint32_t>* n, std::vector<uint32_t>* ovf) {
  idx->assign((size_t)nc * cap, 0xFFFFFFFFu); n->assign(nc, 0); ovf->assign(nc, 0);
  const uint32_t nt = (t->t.n_reviews + GK_TILE - 1) / GK_TILE;
  if (t->last_viol.size() != (size_t)nc * nt) throw std::runtime_error("dev_topk: evaluate the table first");
  for (uint32_t c = 0; c < nc && k; c++) {
    uint32_t count = 0, gk_ = 0;
    for (uint32_t p = 0; p < order.size(); p++) {
      uint32_t rev = order[p];
      if (!((t->last_viol[(size_t)c * nt + rev / GK_TILE] >> (rev % GK_TILE)) & 1)) continue;
      if (count >= k && grp[p] != gk_) break;
      if (count == k - 1) gk_ = grp[p];
      if (count < cap) (*idx)[(size_t)c * cap + count] = rev; else (*ovf)[c] = 1;
      count++;
    }
    (*n)[c] = count < cap ? count : cap;
  }
}
// ---- sharded exchange on the CPU: the collectives are callbacks supplied by the test (torch.distributed / gloo), the slot
// layout is kernels.hip's and the ORDER OF OPERATIONS of the passes is the product's own: csrc/shard_pipe.hpp, instantiated here
// with a worker THREAD for the exchange stream and condition variables for the events (the table's stream is the calling thread).
// GK_SHARD_OVERLAP=1 therefore runs the two-buffer overlapped exchange for real -- pass k's all-gather on the exchange thread
// while pass k + 1 is evaluated -- at any world size gloo offers.
typedef void (*HeAllGather)(void* ctx, void* buf, uint64_t slot_bytes);          // in place: buf = [world][slot_bytes]
typedef void (*HeAllReduce)(void* ctx, long long* buf, uint64_t n);              // sum, in place
struct DevComm {
  int rank = 0, world = 1; HeAllGather gather = nullptr; HeAllReduce reduce = nullptr; void* ctx = nullptr;
  // collectives of one communicator run in ISSUE order on every rank, whichever stream (thread) carries them -- as RCCL's do
  std::mutex mu; std::condition_variable cv; uint64_t issued = 0, serving = 0;
  uint64_t ticket() { std::lock_guard<std::mutex> l(mu); return issued++; }
  void gather_in_order(uint64_t tk, void* buf, uint64_t slot_bytes) {
    { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return serving == tk; }); }
    gather(ctx, buf, slot_bytes);
    { std::lock_guard<std::mutex> l(mu); serving++; }
    cv.notify_all();
  }
};
bool dev_comm_unique_id(char id[128], std::string*) { memset(id, 0, 128); return true; }
DevComm* dev_comm_init(int, const char*, int, int, std::string* err) { *err = "the CPU emulation takes its collectives from gk_comm_init_host"; return nullptr; }
DevComm* hostemu_comm(int rank, int world, HeAllGather g, HeAllReduce r, void* ctx) { DevComm* c = new DevComm(); c->rank = rank; c->world = world; c->gather = g; c->reduce = r; c->ctx = ctx; return c; }
void dev_comm_free(DevComm* c) { delete c; }
int dev_comm_rank(const DevComm* c) { return c->rank; }
int dev_comm_world(const DevComm* c) { return c->world; }
bool dev_comm_query(const DevComm* c, int* rank, int* world, std::string*) { *rank = c->rank; *world = c->world; return true; }   // (what the test's gloo group said)
static size_t shard_tail_off(uint32_t nc, uint32_t stride) { return (size_t)nc * stride * 8; }

// a stream: closures run in order on a thread of their own; an event: "everything recorded up to generation g has run"
struct EmuStream {
  std::thread th; std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; bool stop = false, busy = false; std::string error;
  EmuStream() { th = std::thread([this] {
    for (;;) {
      std::function<void()> fn;
      { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return stop || !q.empty(); }); if (q.empty()) return; fn = std::move(q.front()); q.pop_front(); busy = true; }
      try { fn(); } catch (const std::exception& ex) { std::lock_guard<std::mutex> l(mu); if (error.empty()) error = ex.what(); }
      { std::lock_guard<std::mutex> l(mu); busy = false; }
      cv.notify_all();
    } }); }
  ~EmuStream() { { std::lock_guard<std::mutex> l(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); }
  void push(std::function<void()> fn) { { std::lock_guard<std::mutex> l(mu); q.push_back(std::move(fn)); } cv.notify_all(); }
  void sync() { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return q.empty() && !busy; }); if (!error.empty()) { std::string e = error; error.clear(); throw std::runtime_error("exchange stream: " + e); } }
};
struct EmuEvent { std::mutex mu; std::condition_variable cv; uint64_t issued = 0, done = 0; };
struct ShardState {   // what Work holds for a shard in kernels.hip
  std::vector<uint8_t> buf[2];
  std::vector<long long> totals[2];   // totals over the gathered tails, per buffer (the product keeps ONE d_totals: written in pass order)
  std::unique_ptr<EmuStream> comm;
  EmuEvent ev[2][2];   // [event kind][buffer]
  struct Backend;
  ShardPipe<Backend> pipe;
  bool enqueued = false;   // an enqueue-only pass ran since the setup: its answer can be collected
};
struct DevTableShard { std::shared_ptr<ShardState> st; };
static std::map<const DevTable*, std::shared_ptr<ShardState>>& shard_states() { static std::map<const DevTable*, std::shared_ptr<ShardState>> m; return m; }
static std::mutex& shard_states_mu() { static std::mutex m; return m; }
static std::shared_ptr<ShardState> shard_state_of(const DevTable* t, bool make = false) {
  std::lock_guard<std::mutex> l(shard_states_mu());
  auto it = shard_states().find(t);
  if (it != shard_states().end()) return it->second;
  if (!make) return nullptr;
  return shard_states()[t] = std::make_shared<ShardState>();
}
void hostemu_table_gone(const DevTable* t) { std::shared_ptr<ShardState> st; { std::lock_guard<std::mutex> l(shard_states_mu()); auto it = shard_states().find(t); if (it != shard_states().end()) { st = it->second; shard_states().erase(it); } } }

void dev_shard_setup(DevTable* t, DevComm* c, uint32_t nc, ShardInfo* info) {
  std::vector<unsigned long long> sizes(c->world, 0);
  sizes[c->rank] = t->t.n_reviews;
  c->gather_in_order(c->ticket(), sizes.data(), 8);
  info->shard_reviews.clear();
  uint32_t stride = 1;
  for (int r = 0; r < c->world; r++) { info->shard_reviews.push_back((uint32_t)sizes[r]); stride = std::max<uint32_t>(stride, (uint32_t)((sizes[r] + GK_TILE - 1) / GK_TILE)); }
  info->stride_tiles = stride;
  // slot (kernels.hip): [nc][stride] u64 bitmap | [nc] u32 violating pairs | [nc] u32 autoreject pairs | u64 beyond limits | u64 not evaluated | pad
  info->slot_bytes = (shard_tail_off(nc, stride) + (size_t)nc * 8 + 16 + 15) & ~(size_t)15;
  t->shard_stride = stride; t->shard_slot = info->slot_bytes; t->shard_nc = nc;
  hostemu_table_gone(t);
  auto st = shard_state_of(t, true);
  st->pipe.overlap = getenv("GK_SHARD_OVERLAP") && atoi(getenv("GK_SHARD_OVERLAP")) != 0;   // (opt-in, as in kernels.hip)
  for (int b = 0; b < (st->pipe.overlap ? 2 : 1); b++) st->buf[b].assign((size_t)c->world * info->slot_bytes, 0);
  if (st->pipe.overlap) st->comm.reset(new EmuStream());
  t->shard_all.clear();
}

struct ShardState::Backend {
  const DevPlan* p; DevTable* t; DevComm* c; EvalOptions opt; uint32_t nc; uint64_t not_evaluated; ShardState* st;
  int sel = 0;
  void on(ShardStream s, std::function<void()> fn) { if (s == SS_TABLE) fn(); else st->comm->push(std::move(fn)); }   // (the table's stream is the calling thread)
  void select(int b) { sel = b; }
  void sweep() { EvalOut tmp; EvalOptions o = opt; o.download = true; dev_eval_launch(p, t, o); dev_eval_finish(p, t, o, &tmp); }   // -> t->last_viol / last_err / last_big
  void tail() {
    const uint32_t nt = (t->t.n_reviews + GK_TILE - 1) / GK_TILE;
    uint8_t* slot = st->buf[sel].data() + (size_t)c->rank * t->shard_slot;
    memset(slot, 0, t->shard_slot);
    uint8_t* tl = slot + shard_tail_off(nc, t->shard_stride);
    const bool have = t->last_viol.size() == (size_t)nc * nt && t->last_err.size() == (size_t)nc * nt;
    for (uint32_t k = 0; k < nc; k++) {
      uint32_t cnt = 0, ecnt = 0;
      for (uint32_t w = 0; w < nt; w++) { const uint64_t v = have ? t->last_viol[(size_t)k * nt + w] : 0; memcpy(slot + ((size_t)k * t->shard_stride + w) * 8, &v, 8); cnt += (uint32_t)__builtin_popcountll(v); }
      for (uint32_t w = 0; w < nt && have; w++) ecnt += (uint32_t)__builtin_popcountll(t->last_err[(size_t)k * nt + w]);
      memcpy(tl + (size_t)k * 4, &cnt, 4);
      memcpy(tl + ((size_t)nc + k) * 4, &ecnt, 4);
    }
    unsigned long long beyond = 0, ne = not_evaluated;
    for (uint32_t w = 0; w < nt && w < t->last_big.size(); w++) beyond += (unsigned long long)__builtin_popcountll(t->last_big[w]);
    memcpy(tl + (size_t)nc * 8, &beyond, 8);
    memcpy(tl + (size_t)nc * 8 + 8, &ne, 8);
  }
  void gather(ShardStream s) {
    const uint64_t tk = c->ticket();   // (issue order = the calling thread's order, the same on every rank)
    uint8_t* buf = st->buf[sel].data(); const uint64_t sb = t->shard_slot; DevComm* cc = c;
    on(s, [cc, tk, buf, sb] { cc->gather_in_order(tk, buf, sb); });
  }
  void totals(ShardStream s) {
    ShardState* S = st; const int b = sel; const uint32_t n = nc, stride = t->shard_stride; const size_t sb = t->shard_slot; const int world = c->world;
    on(s, [S, b, n, stride, sb, world] {
      std::vector<long long>& T = S->totals[b];
      T.assign(2 * (size_t)n + 2, 0);   // [nc] pairs | [nc] autoreject pairs | beyond limits | not evaluated: sums over the gathered tails
      for (int r = 0; r < world; r++) {
        const uint8_t* tl = S->buf[b].data() + (size_t)r * sb + shard_tail_off(n, stride);
        for (uint32_t i = 0; i < 2 * n; i++) { uint32_t v; memcpy(&v, tl + (size_t)i * 4, 4); T[i] += v; }
        for (uint32_t i = 0; i < 2; i++) { unsigned long long v; memcpy(&v, tl + (size_t)n * 8 + (size_t)i * 8, 8); T[2 * (size_t)n + i] += (long long)v; }
      }
    });
  }
  void record(ShardEvent e, int b, ShardStream s) {
    EmuEvent* ev = &st->ev[e][b];
    uint64_t g; { std::lock_guard<std::mutex> l(ev->mu); g = ++ev->issued; }
    on(s, [ev, g] { { std::lock_guard<std::mutex> l(ev->mu); if (ev->done < g) ev->done = g; } ev->cv.notify_all(); });
  }
  void wait(ShardStream s, ShardEvent e, int b) {
    EmuEvent* ev = &st->ev[e][b];
    uint64_t g; { std::lock_guard<std::mutex> l(ev->mu); g = ev->issued; }   // (the most recent record at the time of the call, as hipStreamWaitEvent)
    on(s, [ev, g] { std::unique_lock<std::mutex> l(ev->mu); ev->cv.wait(l, [&] { return ev->done >= g; }); });
  }
  void sync(ShardStream s) { if (s == SS_COMM && st->comm) st->comm->sync(); }
};

static void shard_read(DevTable* t, ShardState* st, uint32_t nc, std::vector<int64_t>* totals, std::vector<uint64_t>* gathered, const void** d_gathered) {
  const int b = st->pipe.cur;
  totals->assign(st->totals[b].begin(), st->totals[b].end());
  totals->resize(2 * (size_t)nc + 2, 0);
  if (gathered) { gathered->resize(st->buf[b].size() / 8); memcpy(gathered->data(), st->buf[b].data(), st->buf[b].size()); }
  if (d_gathered) *d_gathered = st->buf[b].data();
}
// the exchange of a collecting sweep: on the table's stream, into the CURRENT buffer, after whatever the exchange stream still runs
void dev_shard_exchange(DevTable* t, DevComm* c, uint32_t nc, uint64_t not_evaluated, std::vector<int64_t>* totals, std::vector<uint64_t>* gathered,
                        const void** d_gathered) {
  auto st = shard_state_of(t);
  if (t->shard_nc != nc || !st) throw std::runtime_error("dev_shard_exchange without dev_shard_setup");
  ShardState::Backend be{nullptr, t, c, EvalOptions(), nc, not_evaluated, st.get()};
  st->pipe.drain(be);
  be.select(st->pipe.cur);
  be.tail();
  const auto x0 = std::chrono::steady_clock::now();
  be.gather(SS_TABLE);
  if (totals) { t->exchange_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - x0).count(); t->exchange_world = c->world; }
  be.totals(SS_TABLE);
  if (!totals) return;
  shard_read(t, st.get(), nc, totals, gathered, d_gathered);
}
ShardExchangeStats dev_shard_exchange_stats(const DevTable* t, const DevComm* c) {
  ShardExchangeStats s;
  if (!t || !c) return s;
  s.exchange_ms = t->exchange_ms;
  s.inbound_bytes = (uint64_t)(c->world > 0 ? c->world - 1 : 0) * t->shard_slot;
  s.overlap = false;
  return s;
}
// the answer of the LAST enqueue-only pass without sweeping again (kernels.hip dev_shard_collect)
bool dev_shard_collect(DevTable* t, DevComm* c, uint32_t nc, std::vector<int64_t>* totals, std::vector<uint64_t>* gathered, const void** d_gathered) {
  auto st = shard_state_of(t);
  if (!st || t->shard_nc != nc || !st->enqueued) return false;
  ShardState::Backend be{nullptr, t, c, EvalOptions(), nc, 0, st.get()};
  st->pipe.drain(be);
  shard_read(t, st.get(), nc, totals, gathered, d_gathered);
  return true;
}
void dev_shard_enqueue(const DevPlan* p, DevTable* t, DevComm* c, const EvalOptions& opt, uint32_t nc, uint64_t not_evaluated, bool) {
  auto st = shard_state_of(t);
  if (!st || t->shard_nc != nc) throw std::runtime_error("dev_shard_enqueue without dev_shard_setup");
  ShardState::Backend be{p, t, c, opt, nc, not_evaluated, st.get()};
  st->pipe.enqueue(be);
  st->enqueued = true;
}

// GK_HOSTEMU_KERNEL=1 | jit: additionally run the dominant kernel's HIP source (kernel_body.inc) through the kernel emulator
// (kernel_emu.hpp; generic bytecode build | plan-specialised build compiled with g++) and require bit-identical outputs
static void emu_kernel_check(const DevPlan* p, const DevTable* dt, const EvalOptions& opt, const EvalOut& want);

void dev_last_viol(const DevTable* t, uint32_t nc, std::vector<uint64_t>* viol) { (void)nc; *viol = t->last_viol; }
void dev_jit_quiesce() {}
void dev_jit_cache_stats(uint64_t* hits, uint64_t* compiles) { *hits = 0; *compiles = 0; }
void dev_jit_cache_drop_memory() {}
const char* dev_jit_cache_dir() { return ""; }
void dev_jit_prefetch(const DevPlan*, const DevTable*) {}
void dev_eval_launch(const DevPlan*, const DevTable* dt, const EvalOptions&) { const_cast<DevTable*>(dt)->pending++; }
void dev_eval(const DevPlan* p, const DevTable* dt, const EvalOptions& opt, EvalOut* o) { dev_eval_launch(p, dt, opt); dev_eval_finish(p, dt, opt, o); }
void dev_eval_finish(const DevPlan* p, const DevTable* dt, const EvalOptions& opt, EvalOut* o) {
  // as strict as the device (kernels.hip dev_eval_finish): collecting needs a launch -- a test that forgets it must fail here too
  if (dt->pending == 0 && dt->t.n_reviews != 0 && !p->fast.slots.empty()) throw std::runtime_error("dev_eval_finish without a pending launch");
  o->n_launches = (uint32_t)dt->pending; const_cast<DevTable*>(dt)->pending = 0;
  o->lds_bytes = p->fast.dims.acc_words * dt->t.rpt * 4;
  const HostTable& t = dt->t;
  uint32_t n = t.n_reviews, nc = (uint32_t)p->fast.slots.size();
  uint32_t nt = (n + GK_TILE - 1) / GK_TILE;
  o->n_reviews = n; o->n_constraints = nc; o->n_tiles = nt;
  o->viol.assign((size_t)nc * nt, 0); o->err.assign((size_t)nc * nt, 0);
  o->match.clear();
  if (opt.want_match) o->match.assign((size_t)nc * nt, 0);
  o->too_big.assign(nt, 0); o->counts.assign(nc, 0); o->list.clear(); o->list_total = 0; o->n_overflow = 0;
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t r = 0; r < n; r++) {
    uint32_t tile = r / GK_TILE; uint64_t bit = 1ull << (r % GK_TILE);
    Results res{};
    if (t.rflags[r] & RF_SKIP) continue;   // like the kernel's `usable` mask
    if (t.rflags[r] & RF_REFUSE) { o->too_big[tile] |= bit; continue; }
    if (!eval_review(p->fast, t, r, &res, p->row ? p : nullptr)) {
      o->n_overflow++;
      if (!eval_review(p->big, t, r, &res)) { o->too_big[tile] |= bit; continue; }
    }
    for (uint32_t c = 0; c < nc; c++) {
      ConstraintSlot sl = p->fast.slots[c];
      const bool prem = (t.rflags[r] & RF_PREMATCHED) != 0;   // (kernel_body.inc: the caller matched, nothing is autorejected)
      bool m = prem || ((res.match >> sl.match) & 1), e = !prem && ((res.err >> sl.match) & 1), v = m && res.viol_bit(sl.viol);
      if (opt.want_match && m) o->match[(size_t)c * nt + tile] |= bit;
      if (e) o->err[(size_t)c * nt + tile] |= bit;
      if (v) {
        o->viol[(size_t)c * nt + tile] |= bit;
        o->counts[c]++;
        o->list_total++;
        if (opt.list_capacity && o->list.size() / 2 < opt.list_capacity) { o->list.push_back(c); o->list.push_back(r); }
      }
    }
  }
#ifdef GK_COUNT_OPS
  if (getenv("GK_PRINT_OPS")) fprintf(stderr, "[hostemu] formula ops per review: %.1f\n", (double)gk_op_counter / (n ? n : 1));
  gk_op_counter = 0;
#endif
  const_cast<DevTable*>(dt)->last_viol = o->viol;
  const_cast<DevTable*>(dt)->last_err = o->err;
  const_cast<DevTable*>(dt)->last_big = o->too_big;
  if (getenv("GK_HOSTEMU_KERNEL")) emu_kernel_check(p, dt, opt, *o);
  o->kernel_ms = o->fast_kernel_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace gk

// ------------------------------------------------------------------------------------------------ kernel emulation
// (kept at the end of the file: kernel_emu.hpp defines HIP's names as macros)
#include <map>
#include <mutex>

#include "chunks.hpp"
#include "kernel_emu.hpp"

namespace gk {
#define GK_SKIP_BIG
#define GK_KERNEL_BIG gk_emu_big
#define GK_KERNEL_LINKAGE static
#define GK_ROW_FN(r, i, ent, h, pv, heap, acc, on) do { if (on) eval_row_ent(r, i, ent, h, pv, heap, acc); } while (0)
#define GK_FORMULA_FN(pv, acc, flags, rows, heap, bounds) eval_formulas(pv, acc, flags, rows, heap, bounds)
#define GK_RPT_K 64
#define GK_KERNEL_TILES gk_emu_tiles_64
#include "kernel_body.inc"
#undef GK_RPT_K
#undef GK_KERNEL_TILES
#undef GK_TILES_BOUNDS
#define GK_RPT_K 128
#define GK_KERNEL_TILES gk_emu_tiles_128
#include "kernel_body.inc"
#undef GK_RPT_K
#undef GK_KERNEL_TILES
#undef GK_TILES_BOUNDS
#define GK_RPT_K 256
#define GK_KERNEL_TILES gk_emu_tiles_256
#include "kernel_body.inc"
#undef GK_RPT_K
#undef GK_KERNEL_TILES
#undef GK_TILES_BOUNDS
#define GK_RPT_K 512
#define GK_KERNEL_TILES gk_emu_tiles_512
#include "kernel_body.inc"
#undef GK_RPT_K
#undef GK_KERNEL_TILES

typedef void (*EmuJitLaunch)(unsigned, unsigned, size_t, const PlanView*, const Row*, const StrHdr*, const ChunkDesc*, uint32_t, const uint32_t*, const uint8_t*,
                             uint32_t, uint32_t, const ConstraintSlot*, const OutPtrs*, uint32_t, uint32_t);
static EmuJitLaunch emu_jit_for(const DevPlan* p, uint32_t rpt, uint32_t rpp, int block) {
  char key[96];
  snprintf(key, sizeof key, "%u/%u/%d", rpt, rpp, block);
  DevPlan* mp = const_cast<DevPlan*>(p);
  auto it = mp->emu_jit.find(key);
  if (it != mp->emu_jit.end()) return (EmuJitLaunch)it->second.second;
  static int counter = 0;
  std::string base = "/tmp/gkemu_jit_" + std::to_string(getpid()) + "_" + std::to_string(counter++);
  {
    std::ofstream f(base + ".cpp");
    const char* pf = getenv("GK_JIT_PREFETCH");
    f << "#include \"" << GK_CSRC_DIR << "/../../tests/native/kernel_emu.hpp\"\n#include \"" << GK_CSRC_DIR << "/vm_core.hpp\"\n"
      << "#define GK_LANE_ID() (threadIdx.x & 63u)\n"
      << "#define GK_BIT(b) ((b) & 1u)\n"   // (jit_source.hpp: on the device an opaque copy in front of the mask)
      << "#define GK_WRITELANE2(m, l, lo, hi) do { if ((threadIdx.x & 63u) == (uint32_t)(l)) { (lo) = (uint32_t)(m); (hi) = (uint32_t)((m) >> 32); } } while (0)\n"
      << jit_res_macros()
      << generate_plan_source(p->fast, (uint32_t)(block / GK_TILE / ((int)rpt / GK_TILE)))
      << "#define GK_RPT_K " << rpt << "\n#define GK_RPP_K " << rpp << "\n#define GK_BLOCK_K " << block << "\n#define GK_PREFETCH " << (pf ? pf : "1") << "\n#define GK_SKIP_BIG\n"
      << (jit_tot_k(p->fast.dims.n_constraints) ? "#define GK_TOT_K " + std::to_string(jit_tot_k(p->fast.dims.n_constraints)) + "\n" : std::string())   // (the emulator has no CU to fill: the array whenever the plan allows it)
      << "namespace gk {\n#define GK_KERNEL_TILES gk_jit_tiles\n#define GK_KERNEL_BIG gk_jit_big\n#define GK_KERNEL_LINKAGE static\n"
         "#define GK_ROW_FN(r, i, ent, h, pv, heap, acc, on) jit_row(r, ent, h, heap, acc, on)\n#define GK_BIND_ALWAYS_STR 0\n"
         "#define GK_FORMULA_FN(pv, acc, flags, rows, heap, bounds) jit_formulas(pv, acc, flags, rows, heap, bounds)\n"
      << "#include \"" << GK_CSRC_DIR << "/kernel_body.inc\"\n}\n"
      << "extern \"C\" void gk_emu_jit_launch(unsigned grid, unsigned block, size_t lds, const gk::PlanView* pv, const gk::Row* rows, const gk::StrHdr* shdr,\n"
         "    const gk::ChunkDesc* lists, uint32_t capg, const uint32_t* rflags, const uint8_t* heap, uint32_t n, uint32_t nt, const gk::ConstraintSlot* slots,\n"
         "    const gk::OutPtrs* out, uint32_t dbg, uint32_t rpp) {\n"
         "  gkemu::launch(grid, block, lds, [&] { gk::gk_jit_tiles(*pv, rows, shdr, lists, capg, rflags, heap, n, nt, slots, *out, dbg, rpp); });\n}\n";
  }
  // (GK_EMU_CXXFLAGS, e.g. -fsanitize=undefined: the generated text under a sanitizer -- what g++ and the device compiler may disagree on)
  std::string cmd = std::string(getenv("GK_EMU_CXX") ? getenv("GK_EMU_CXX") : "g++") + std::string(" -std=c++17 -O1 -shared -fPIC -w ") + (getenv("GK_EMU_CXXFLAGS") ? getenv("GK_EMU_CXXFLAGS") : "") + " -o " + base + ".so " + base + ".cpp 2> " + base + ".log";
  if (system(cmd.c_str()) != 0) throw std::runtime_error("kernel_emu: the plan-specialised kernel does not compile, see " + base + ".log");
  void* dl = dlopen((base + ".so").c_str(), RTLD_NOW);
  if (!dl) throw std::runtime_error(std::string("kernel_emu: dlopen failed: ") + dlerror());
  EmuJitLaunch fn = (EmuJitLaunch)dlsym(dl, "gk_emu_jit_launch");
  if (!getenv("GK_EMU_KEEP")) { unlink((base + ".cpp").c_str()); unlink((base + ".so").c_str()); unlink((base + ".log").c_str()); }
  mp->emu_jit[key] = std::make_pair(dl, (void*)fn);
  return fn;
}

static void emu_kernel_check(const DevPlan* p, const DevTable* dt, const EvalOptions& opt, const EvalOut& want) {
  static std::mutex mu;   // the emulator keeps one workgroup's state in statics
  std::lock_guard<std::mutex> lock(mu);
  const bool jit = std::string(getenv("GK_HOSTEMU_KERNEL")) == "jit";
  const HostTable& t = dt->t;
  const HostPlan& hp = p->fast;
  const uint32_t n = t.n_reviews, nc = (uint32_t)hp.slots.size(), nt = (n + GK_TILE - 1) / GK_TILE, rpt = t.rpt;
  if (!n || !nc) return;
  int block = gk_block_of((int)rpt);
  if (jit && getenv("GK_JIT_BLOCK")) { const int f = atoi(getenv("GK_JIT_BLOCK")); if (f >= (int)rpt && f <= 1024 && f % (int)rpt == 0) block = f; }
  // bound paths, costs and chunk lists exactly as kernels.hip builds them
  std::vector<std::vector<Pred>> classes;
  const std::vector<uint32_t> entries = jit ? jit_path_classes(hp, &classes) : hp.ptab;
  std::vector<BoundPath> bound;
  for (uint32_t s = 0; s < t.n_slots(); s++) {
    const uint32_t path = t.slot_path[s];
    if (path >= entries.size() || !entries[path]) continue;
    const uint32_t ent = hp.ptab[path];
    const uint32_t c = pred_list_cost<Pred>(&hp.path_preds[ent >> 8], ent & 0xFF, [](const Pred& q) { return pred_needs_str(q); });
    bound.push_back(BoundPath{s, entries[path], c});
  }
  const uint32_t n_groups = (n + rpt - 1) / rpt;
  uint32_t list_cap = (uint32_t)std::min(block / GK_TILE, 8) * GK_WAVE_CHUNKS;
  if (const char* lc = getenv("GK_EMU_LIST_CAP")) list_cap = std::min<uint32_t>(list_cap, (uint32_t)atoi(lc));   // test aid: list overflow
  ChunkLists cl = build_chunk_lists(t.tile_idx.data(), n_groups, t.n_slots(), bound, list_cap, (uint32_t)(block / GK_TILE));
  if (getenv("GK_EMU_CHUNK_STATS")) {   // diagnostic: chunks / rows per predicate class of this table's lists
    std::map<uint32_t, std::array<uint64_t, 5>> st;   // class -> {chunks, rows, needs_str, chunks in segments of >= 128 rows, segments}
    for (const BoundPath& b : bound) {
      auto& e = st[b.ent & GK_DESC_ENT_MASK];
      e[2] = (b.ent & GK_ENT_NEEDS_STR) ? 1 : 0;
      for (uint32_t g = 0; g < n_groups; g++) {
        const uint32_t lo = t.tile_idx[(size_t)g * (t.n_slots() + 1) + b.slot], hi = t.tile_idx[(size_t)g * (t.n_slots() + 1) + b.slot + 1];
        if (hi == lo) continue;
        const uint32_t ch = (hi - lo + GK_TILE - 1) / GK_TILE;
        e[0] += ch; e[1] += hi - lo; e[4]++;
        if (hi - lo >= 128) e[3] += (hi - lo) / 128 * 2;
      }
    }
    uint64_t tc = 0, tr = 0, td = 0, ts = 0;
    for (auto& kv : st) {
      fprintf(stderr, "[chunk_stats] class %u str %llu chunks/group %.2f rows/chunk %.1f pairable %.2f segs/group %.2f cost %u\n", kv.first, (unsigned long long)kv.second[2], (double)kv.second[0] / n_groups,
              kv.second[0] ? (double)kv.second[1] / kv.second[0] : 0.0, (double)kv.second[3] / n_groups, (double)kv.second[4] / n_groups, 0u);
      tc += kv.second[0]; tr += kv.second[1]; if (!kv.second[2]) td += kv.second[3]; if (kv.second[2]) ts += kv.second[0];
    }
    fprintf(stderr, "[chunk_stats] groups %u chunks/group %.1f rows/group %.1f str chunks/group %.1f pairable non-str chunks/group %.1f\n", n_groups, (double)tc / n_groups, (double)tr / n_groups, (double)ts / n_groups, (double)td / n_groups);
  }
  // reviews per pass
  uint32_t rpp = rpt;
  while (rpp > (uint32_t)GK_TILE && (size_t)hp.dims.acc_words * rpp * 4 > 140 * 1024) rpp /= 2;
  if (const char* f = getenv("GK_FORCE_RPP")) rpp = std::min<uint32_t>(rpt, std::max<uint32_t>(GK_TILE, (uint32_t)atoi(f)));
  const size_t lds = (size_t)hp.dims.acc_words * rpp * 4;
  // outputs (garbage-filled: the kernel must write every word it owns)
  std::vector<uint64_t> viol((size_t)nc * nt, 0xABABABABABABABABull), err((size_t)nc * nt, 0xABABABABABABABABull), match((size_t)nc * nt, 0xABABABABABABABABull);
  std::vector<uint64_t> ovf(nt, 0xABABABABABABABABull), big(nt, 0xABABABABABABABABull);
  std::vector<uint32_t> counts(nc, 0), list((size_t)std::max<uint32_t>(opt.list_capacity, 1) * 2, 0), lcnt(2, 0);
  OutPtrs out{viol.data(), err.data(), opt.want_match ? match.data() : nullptr, ovf.data(), big.data(), counts.data(), list.data(), lcnt.data(), opt.list_capacity, nullptr};
  PlanView pv = view_of(hp);
  unsigned grid = (n_groups + 7u) / 8u * 8u;
  if (const char* g = getenv("GK_EMU_GRID")) grid = std::min<unsigned>(grid, (unsigned)std::max(8, atoi(g) / 8 * 8));   // persistent workgroups: several groups each
  else grid = std::min<unsigned>(grid, 16u);
  // per-workgroup violation counts (kernel_body.inc GK_TOT_K, the plan-specialised text): every workgroup of the grid writes its row
  const uint32_t tot_k = jit ? jit_tot_k(hp.dims.n_constraints) : 0u;
  const bool emu_partial = tot_k != 0;
  std::vector<uint32_t> partial((size_t)grid * tot_k, 0xABABABABu);
  if (emu_partial) out.partial = partial.data();
  const Row* rows = t.rows.data(); const StrHdr* shdr = t.shdr.data();
  const uint32_t emu_dbg = getenv("GK_EMU_STAGGER") ? ((uint32_t)atoi(getenv("GK_EMU_STAGGER")) & 0xFFFu) << 8 : 0u;   // the launch word's stagger field, as dev_eval_launch sets it for >= 1024 groups
  if (jit) {
    if (const char* dir = getenv("GK_EMU_HIP_SOURCE_DIR")) {
      // test aid (tests/test_jit_source.py): the text kernels.hip would hand to hiprtc for this plan, geometry and table
      static int n_dumped = 0;
      std::ofstream f(std::string(dir) + "/gk_plan_" + std::to_string(getpid()) + "_" + std::to_string(n_dumped++) + "_" + std::to_string(rpt) + "_" + std::to_string(rpp) + ".hip");
      f << assemble_jit_source(hp, rpt, rpp, kPlanHpp, kVmCoreHpp, kKernelBody);
    }
    EmuJitLaunch fn = emu_jit_for(p, rpt, rpp, block);
    fn(grid, (unsigned)block, lds, &pv, rows, shdr, cl.d.data(), cl.capg, t.rflags.data(), t.heap.data(), n, nt, hp.slots.data(), &out, emu_dbg, rpp);
  } else {
    auto fn = rpt == 64 ? gk_emu_tiles_64 : rpt == 128 ? gk_emu_tiles_128 : rpt == 256 ? gk_emu_tiles_256 : gk_emu_tiles_512;
    gkemu::launch(grid, (unsigned)block, lds, [&] { fn(pv, rows, shdr, cl.d.data(), cl.capg, t.rflags.data(), t.heap.data(), n, nt, hp.slots.data(), out, emu_dbg, rpp); });
  }
  // the workgroups' own counts add up to the popcount of the rows they wrote (what gk_sum_partials hands out as the totals)
  if (emu_partial)
    for (uint32_t c = 0; c < nc; c++) {
      uint64_t by_rows = 0, by_wgs = 0;
      for (uint32_t w = 0; w < nt; w++) by_rows += (uint64_t)__builtin_popcountll(viol[(size_t)c * nt + w]);
      for (unsigned g = 0; g < grid; g++) by_wgs += partial[(size_t)g * tot_k + c];
      if (by_rows != by_wgs) throw std::runtime_error("kernel_emu: per-workgroup violation counts of constraint " + std::to_string(c) + " sum to " + std::to_string(by_wgs) + ", its bitmap row holds " + std::to_string(by_rows));
    }
  // overflowed reviews: the big variant (per review, as above)
  uint32_t n_ovf = 0;
  for (uint32_t w = 0; w < nt; w++)
    for (uint64_t m = ovf[w]; m; m &= m - 1) {
      const uint32_t r = w * GK_TILE + (uint32_t)__builtin_ctzll(m);
      const uint64_t bit = 1ull << (r % GK_TILE);
      n_ovf++;
      Results res{};
      if (!eval_review(p->big, t, r, &res)) { big[w] |= bit; continue; }
      for (uint32_t c = 0; c < nc; c++) {
        ConstraintSlot sl = hp.slots[c];
        const bool m_ = (res.match >> sl.match) & 1, e = (res.err >> sl.match) & 1, v = m_ && res.viol_bit(sl.viol);
        if (opt.want_match && m_) match[(size_t)c * nt + w] |= bit;
        if (e) err[(size_t)c * nt + w] |= bit;
        if (v) { viol[(size_t)c * nt + w] |= bit; lcnt[0]++; }   // (the big kernel appends its pairs to the same list)
      }
    }
  if (lcnt[1] != n_ovf) throw std::runtime_error("kernel_emu: overflow counter and overflow bitmap disagree");
  auto fail = [&](const char* what, uint32_t c, uint32_t w, uint64_t got, uint64_t exp) {
    char buf[256];
    snprintf(buf, sizeof buf, "kernel_emu (%s, rpt %u rpp %u grid %u): %s differs at constraint %u word %u: kernel %016llx, per-review %016llx", jit ? "jit" : "generic", rpt, rpp,
             grid, what, c, w, (unsigned long long)got, (unsigned long long)exp);
    throw std::runtime_error(buf);
  };
  for (uint32_t c = 0; c < nc; c++)
    for (uint32_t w = 0; w < nt; w++) {
      if (viol[(size_t)c * nt + w] != want.viol[(size_t)c * nt + w]) fail("violation bitmap", c, w, viol[(size_t)c * nt + w], want.viol[(size_t)c * nt + w]);
      if (err[(size_t)c * nt + w] != want.err[(size_t)c * nt + w]) fail("autoreject bitmap", c, w, err[(size_t)c * nt + w], want.err[(size_t)c * nt + w]);
      if (opt.want_match && match[(size_t)c * nt + w] != want.match[(size_t)c * nt + w]) fail("match bitmap", c, w, match[(size_t)c * nt + w], want.match[(size_t)c * nt + w]);
    }
  for (uint32_t w = 0; w < nt; w++) if (big[w] != want.too_big[w]) fail("too_big", 0, w, big[w], want.too_big[w]);
  if (opt.list_capacity) {
    if (lcnt[0] != want.list_total) throw std::runtime_error("kernel_emu: violation list length differs");
  }
  if (getenv("GK_EMU_VERBOSE")) fprintf(stderr, "[kernel_emu] %s rpt %u rpp %u grid %u groups %u capg %u chunks %llu overflowed %u: identical\n", jit ? "jit" : "generic", rpt, rpp, grid, n_groups, cl.capg, (unsigned long long)cl.n_chunks, n_ovf);
}
}  // namespace gk

// TEST-ONLY entry point of libgkgpu_hostemu.so: join an engine to a "communicator" whose collectives are callbacks
// (tests/test_sweep_dist.py wires them to torch.distributed / gloo)
struct gk_engine;
extern "C" int gk_comm_init_host_impl(gk_engine* e, gk::DevComm* c);
extern "C" int gk_comm_init_host(gk_engine* e, int rank, int world, gk::HeAllGather g, gk::HeAllReduce r, void* ctx) {
  return gk_comm_init_host_impl(e, gk::hostemu_comm(rank, world, g, r, ctx));
}
