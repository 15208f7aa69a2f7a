// Device backend interface between the host engine (engine.cpp) and the HIP kernels (kernels.hip).
// The product library links kernels.hip.  tests/native/hostemu.cpp implements the same interface on the CPU for
// the build container's GPU-less unit tests ONLY (libgkgpu_hostemu.so; never loaded by the product path).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "flatten.hpp"
#include "lower.hpp"

namespace gk {

struct DevTable;
struct DevPlan;

struct EvalOptions {
  bool download = true;      // copy bitmaps / list back to the host
  bool want_match = false;   // also produce the match-only bitmap
  uint32_t list_capacity = 0;   // max violation-list entries (0 = no list)
  bool shard = false;           // results go to the shard slot prepared by dev_shard_setup (bitmap stride = the largest shard's)
  bool detached = false;        // (kernels.hip, internal) an enqueue-only pass that dev_eval_finish never collects: see dev_shard_enqueue
  bool kernel_only = false;     // GK_EVAL_KERNEL_ONLY: no totals kernel behind the dominant one (back-to-back timing of that kernel)
  bool time_each = false;       // GK_EVAL_TIME_EACH: an event pair around EVERY launch (isolated kernel durations) instead of one around all pending ones
  bool jit_wait = true;         // wait for the plan-specialised build of the dominant kernel; false (admission batches): never
                                //   block -- the bytecode kernel serves until the background build has been loaded
};

struct EvalOut {
  uint32_t n_reviews = 0, n_constraints = 0, n_tiles = 0;
  std::vector<uint64_t> viol;     // [n_constraints][n_tiles]  bit r%64 of word r/64: constraint matched AND violated
  std::vector<uint64_t> err;      // [n_constraints][n_tiles]  Matcher.Match returned an error (autoreject)
  std::vector<uint64_t> match;    // [n_constraints][n_tiles]  (only if want_match)
  std::vector<uint64_t> too_big;  // [n_tiles] reviews that exceed engine limits even in the large variant
  std::vector<uint32_t> counts;   // [n_constraints] violating reviews per constraint
  std::vector<uint32_t> list;     // pairs (constraint, review) -- compacted violation list
  uint32_t list_total = 0;        // entries the kernel wanted to emit (may exceed capacity)
  uint32_t n_overflow = 0;        // reviews re-run in the large-capacity variant
  float kernel_ms = 0;            // device time of the evaluation kernels (HIP events on the launch stream)
  float fast_kernel_ms = 0;       // average duration of the dominant (LDS) kernel per launch since the last finish
  uint32_t n_launches = 0;        // launches averaged in fast_kernel_ms
  uint32_t lds_bytes = 0;         // accumulator LDS per workgroup of the dominant kernel's most recent launch
  uint64_t list_bytes = 0;        // chunk lists the dominant kernel reads per launch (chunks.hpp)
  uint64_t kernel_hash = 0;       // FNV-64 of the plan-specialised kernel's source text (0: the bytecode kernel ran)
  const void *d_viol = nullptr, *d_err = nullptr, *d_counts = nullptr;   // device-resident results (valid until the table's next launch)
};

std::string dev_init(int device);                       // "" on success, else error text
int dev_count();
// A table goes to the device in PARTS: every host thread uploads the part it flattened as soon as it is done
// (dev_part_upload: thread-safe, overlaps the other threads' flattening and their transfers); dev_table_assemble then
// places the parts one after the other in the table's arrays ON THE DEVICE, relocating the rows' heap offsets there -- the
// host never merges the gigabytes.  `meta` carries what is global: tile_idx, slot_path, rflags, n_reviews.
struct DevPart;
DevPart* dev_part_upload(int device, HostTable& part);   // releases the part's host arrays (rows / shdr / heap)
void dev_part_free(DevPart* p);
DevTable* dev_table_assemble(int device, std::vector<DevPart*>& parts, const HostTable& meta);   // consumes the parts
void dev_table_free(DevTable* t);
// A second handle on the same resident table with its own result buffers and path binding (plan groups beyond the
// first evaluate through views); freeing a view leaves the table's arrays alone.  Free views before the table.
DevTable* dev_table_view(DevTable* base);
uint64_t dev_table_bytes(const DevTable* t);
DevPlan* dev_plan_upload(int device, const HostPlan& fast, const HostPlan& big);
void dev_plan_free(DevPlan* p);
void dev_plan_no_jit(DevPlan* p);   // this plan is served by the bytecode kernel only (the small one-sweep-per-audit plans of the RESULT totals)
void dev_eval(const DevPlan* p, const DevTable* t, const EvalOptions& opt, EvalOut* out);   // launch + finish; throws std::runtime_error
// first-k violating reviews per bitmap row in `order` (reviews sorted by object key; grp = dense key rank, ties equal);
// uses the bitmaps of the table's most recent evaluation.  idx: [nc][cap], n: [nc], ovf: [nc]
void dev_topk(const DevTable* t, uint32_t nc, const std::vector<uint32_t>& order, const std::vector<uint32_t>& grp, uint32_t k, uint32_t cap,
              std::vector<uint32_t>* idx, std::vector<uint32_t>* n, std::vector<uint32_t>* ovf);
// the violation bitmap [nc][n_tiles] of the table's most recent evaluation (device -> host copy)
void dev_last_viol(const DevTable* t, uint32_t nc, std::vector<uint64_t>* viol);
// ---- multi-GPU exchange step of a sharded sweep: one communicator per engine (= per GPU / rank)
struct DevComm;
bool dev_comm_unique_id(char id[128], std::string* err);                                  // rank 0: ncclGetUniqueId
DevComm* dev_comm_init(int device, const char id[128], int rank, int world, std::string* err);   // ncclCommInitRank (RCCL over xGMI)
void dev_comm_free(DevComm* c);
int dev_comm_rank(const DevComm* c);
int dev_comm_world(const DevComm* c);
bool dev_comm_query(const DevComm* c, int* rank, int* world, std::string* err);   // ncclCommUserRank / ncclCommCount of the live communicator
// A table evaluated as one SHARD: its violation bitmap and counts live in this rank's slot of an all-gather buffer
// ([world] x ([nc][stride_tiles] u64 | tail | pad), tail below), stride_tiles = the largest shard's words per row.
struct ShardInfo {
  uint32_t stride_tiles = 0;            // bitmap words per row in every slot
  uint64_t slot_bytes = 0;
  std::vector<uint32_t> shard_reviews;  // [world]
};
void dev_shard_setup(DevTable* t, DevComm* c, uint32_t nc, ShardInfo* info);   // collective: exchanges the shard sizes
// after a finished local evaluation of the shard (dev_eval): the slot's tail <- counts, ONE in-place all-gather of the slots,
// int64 totals <- sums over the gathered tails, on the evaluation stream; then synchronises and copies what was asked for to
// the host.  Slot: [nc][stride_tiles] u64 bitmap | [nc] u32 violating pairs | [nc] u32 autoreject pairs | u64 beyond | u64 not evaluated | pad.
// totals: [nc] violating pairs | [nc] autoreject pairs (match errors) | reviews beyond the engine's limits | reviews not
// evaluated (`not_evaluated` of this rank: rejected by HandleReview when the table was built), each summed over ALL shards --
// what the gathered violation bitmaps cannot say, so that a sharded audit fails closed like the single-GPU one.
// totals == nullptr: ENQUEUE ONLY -- counts, all-gather and totals go onto the stream behind the launches of
// dev_eval_launch and the call returns without waiting (back-to-back sweeps; the collecting call comes last)
void dev_shard_exchange(DevTable* t, DevComm* c, uint32_t nc, uint64_t not_evaluated, std::vector<int64_t>* totals,
                        std::vector<uint64_t>* gathered /* may be null */, const void** d_gathered);
// the exchange step of the most recent COLLECTING dev_shard_exchange of this table (enqueue-only passes are not timed): duration of the
// all-gather alone (events around it on its stream; 0 when no collecting exchange ran), the bytes this rank RECEIVES in it
// ((world - 1) x slot), and whether consecutive enqueue-only passes overlap their exchange with the next sweep
struct ShardExchangeStats { float exchange_ms = 0; uint64_t inbound_bytes = 0; bool overlap = false; };
ShardExchangeStats dev_shard_exchange_stats(const DevTable* t, const DevComm* c);
// one enqueue-only sweep + exchange step of a resident shard (launch + exchange; captured into a graph from the second pass on)
void dev_shard_enqueue(const DevPlan* p, DevTable* t, DevComm* c, const EvalOptions& opt, uint32_t nc, uint64_t not_evaluated, bool allow_graph);
// the answer of the most recent enqueue-only pass, without sweeping again; false when there is none or when that pass left
// reviews to the large-capacity re-run (which only a collecting sweep does)
bool dev_shard_collect(DevTable* t, DevComm* c, uint32_t nc, std::vector<int64_t>* totals, std::vector<uint64_t>* gathered, const void** d_gathered);
// Plan-specialised builds of the dominant kernel run in the background for admission batches (kernels.hip jit_for): wait for
// every build in flight / code-object cache counters (hits = builds served without hiprtc)
void dev_jit_quiesce();
void dev_jit_cache_stats(uint64_t* hits, uint64_t* compiles);
void dev_jit_cache_drop_memory();        // forget the code objects held in memory; the disk cache stays (what a restarted process sees)
const char* dev_jit_cache_dir();         // "" = no disk cache
void dev_jit_prefetch(const DevPlan* p, const DevTable* t);   // start the build this (plan, table) will ask for; never waits
void dev_eval_launch(const DevPlan* p, const DevTable* t, const EvalOptions& opt);          // asynchronous on the default stream
void dev_eval_finish(const DevPlan* p, const DevTable* t, const EvalOptions& opt, EvalOut* out);   // sync, overflow re-run, download

}  // namespace gk
