/* gkgpu.h -- C ABI of the MI355X-native constraint-evaluation engine (libgkgpu.so).
 *
 * This is the drop-in boundary a Go `drivers.Driver` (Name()=="Rego") binds through cgo; INTEGRATION.md shows the
 * shim.  Each entry point cites the reference interface it replaces.  Plain pointers and sizes only; the callee
 * copies whatever it retains; outputs are callee-allocated and released with the matching *_free.  All entry points
 * are re-entrant; errors are negative gk_status codes plus gk_last_error() text (thread-local), never exceptions.
 */
#ifndef GKGPU_H
#define GKGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gk_engine gk_engine;   /* opaque, thread-safe */
typedef struct gk_table gk_table;     /* opaque: a flattened, HBM-resident set of reviews */

typedef enum {
  GK_OK = 0,
  GK_ERR_INVALID = -1,      /* bad argument / malformed JSON */
  GK_ERR_REGO = -2,         /* template does not parse / compile (AddTemplate error, driver.go:74-137) */
  GK_ERR_UNSUPPORTED = -3,  /* template/constraint uses constructs the device plan cannot express: the caller keeps
                               it on the reference CPU driver -- this engine never approximates and never falls back */
  GK_ERR_NOT_FOUND = -4,    /* unknown template kind / constraint (driver.go:198-200) */
  GK_ERR_DEVICE = -5,       /* no MI355X / HIP failure */
  GK_ERR_REVIEW = -6,       /* review rejected by HandleReview (target.go:81-138, e.g. ErrOldObjectIsNil) */
  GK_ERR_INTERNAL = -7,
  GK_ERR_LIMIT = -8         /* a review is beyond the engine's limits (an array that element predicates iterate has more
                               than 255 elements, or an object sits where they iterate array elements): reported per
                               review in gk_eval_out.too_big, NEVER evaluated to "no violations" -- the caller must fail
                               closed or take that review to the reference CPU driver */
} gk_status;

/* gk_opts.flags -- the rego.Arg surface the reference constructs its driver with (main.go:424,479-484; pkg/gator/test/test.go:175-190);
 * all zero = the reference deployment's defaults (referential rules on, no stats, no trace) */
#define GK_OPT_NO_REFERENTIAL 1u   /* rego.Externs() instead of rego.Externs("inventory") (--enable-referential-rules=false): a template that
                                      refers to data.inventory is refused by gk_template_add (GK_ERR_REGO), never compiled */
#define GK_OPT_GATHER_STATS 2u     /* rego.GatherStats(): gk_query_ex fills its stats for every call (the caller turns them into
                                      instrumentation.StatsEntry) */
#define GK_OPT_TRACE 4u            /* rego.Tracing(true): gk_query_ex returns a trace text for every call (QueryResponse.Trace) */
typedef struct {
  int32_t device;          /* HIP device ordinal (one engine per GPU; one process per GPU under torch.distributed) */
  uint16_t elem_cap[3];    /* LDS element capacity per array-nesting level; 0 = default (8, 12, 12); larger reviews take the HBM-accumulator kernel variant */
  uint16_t reserved;
  uint32_t flags;          /* GK_OPT_* */
  uint32_t reserved2;
  /* rego.DisableBuiltins(names...): a template that calls one of them is `rego_type_error: undefined function <name>` at
   * gk_template_add.  NULL = the deployment's default {"http.send"} (--disable-opa-builtin, test/bats/test.bats:492-498); an array
   * (n_disabled_builtins may be 0) replaces it.  Copied by gk_engine_create. */
  const char* const* disabled_builtins;
  size_t n_disabled_builtins;
} gk_opts;

/* ---- lifecycle ------------------------------------------------------------------------------------------- */
/* replaces rego.New(args...) at main.go:486, pkg/gator/opa.go:32, pkg/gator/test/test.go:48 */
int gk_engine_create(const gk_opts* opts, gk_engine** out);
void gk_engine_destroy(gk_engine* e);
const char* gk_last_error(void);
const char* gk_version(void);
/* CPUs the engine's host pools size themselves by: hardware threads, cut down to the affinity mask and the cgroup CPU quota
 * (a container that shows 256 hardware threads behind a 16-CPU quota gets 16); GK_HOST_THREADS overrides the pools. */
uint32_t gk_host_cpus(void);

/* ---- policy state (drivers.Driver Add/Remove*, boundary exemplar pkg/drivers/k8scel/driver.go:74-160) ----- */
/* Driver.AddTemplate: kind as in spec.crd.spec.names.kind; rego + libs from Source.Value{"rego","libs"}
 * (pkg/fakes/fixtures.go:35-41).  Parses and statically checks the Rego. */
int gk_template_add(gk_engine* e, const char* kind, const char* rego, const char* const* libs, size_t nlibs);
/* Driver.RemoveTemplate */
int gk_template_remove(gk_engine* e, const char* kind);
/* Driver.AddConstraint: constraint object JSON AFTER the client applied CRD defaults.
 * Compiles spec.match (pkg/target/target.go:246-261, pkg/mutation/match/match.go:32-258) and the template's
 * violation rule specialised to spec.parameters.  Returns GK_ERR_UNSUPPORTED if it cannot run on the device. */
int gk_constraint_add(gk_engine* e, const char* constraint_json, size_t len, uint32_t* id_out);
/* Driver.RemoveConstraint */
int gk_constraint_remove(gk_engine* e, const char* kind, const char* name);
/* Driver.AddData / RemoveData with the path of K8sValidationTarget.ProcessData (pkg/target/target.go:40-66):
 * ["cluster",gv,kind,name] or ["namespace",ns,gv,kind,name].  Namespace objects feed the nsCache
 * (pkg/target/ns_cache.go:22-43); everything is kept as data.inventory for host-side rendering. */
int gk_data_put(gk_engine* e, const char* const* path, size_t npath, const char* json, size_t len);
int gk_data_remove(gk_engine* e, const char* const* path, size_t npath);

/* ---- reviews --------------------------------------------------------------------------------------------- */
typedef enum { GK_SRC_EMPTY = 0, GK_SRC_ORIGINAL = 1, GK_SRC_GENERATED = 2, GK_SRC_ALL = 3, GK_SRC_INVALID = 4 } gk_source;
typedef enum {
  GK_REVIEW_ADMISSION_REQUEST = 0,  /* json = admissionv1.AdmissionRequest (gkReview / AugmentedReview shapes) */
  GK_REVIEW_OBJECT = 1              /* json = bare object (Unstructured / AugmentedUnstructured shapes, target.go:140-179) */
} gk_review_kind;

/* one input of K8sValidationTarget.HandleReview (pkg/target/target.go:81-138, pkg/target/review.go:9-29) */
typedef struct {
  int32_t kind;                 /* gk_review_kind */
  int32_t source;               /* gk_source: gkReview.source / AugmentedUnstructured.Source */
  const char* json;             size_t json_len;
  const char* namespace_json;   size_t namespace_len;    /* gkReview.namespace (*corev1.Namespace) or NULL */
  const char* ns_object_json;   size_t ns_object_len;    /* reviews.Namespace(nsMap) option -> input.review.namespaceObject, or NULL */
  const char* operation;        /* AugmentedUnstructured.Operation ("" / NULL = none) */
} gk_review_in;

/* ---- process excluder (row f3) ---------------------------------------------------------------------------- */
/* Excluder.Replace(New().Add(config.spec.match)) -- pkg/controller/config/process/excluder.go:52-82, fed by the Config
 * controller (pkg/controller/config/config_controller.go): match_json = the Config's spec.match, a JSON array of
 * {"excludedNamespaces": [wildcard...], "processes": ["audit"|"webhook"|"mutation-webhook"|"sync"|"*"...]}; NULL / "[]"
 * clears it.  Tables built with GK_TABLE_PROCESS_* and the resident sweep (process "audit") skip excluded objects before
 * evaluation: pkg/audit/manager.go:530,599, pkg/webhook/policy.go:197. */
int gk_excluder_replace(gk_engine* e, const char* match_json, size_t len);
/* Excluder.IsNamespaceExcluded(process, obj) (excluder.go:96-105) for one review; AdmissionRequests are looked at the way
 * the webhook does (pkg/webhook/common.go:149-189: oldObject on DELETE, namespace = request.namespace). */
int gk_excluder_excluded(gk_engine* e, const char* process, const gk_review_in* review, int32_t* excluded);


/* Flatten n reviews into key-path -> value SoA rows and upload them to HBM (they stay resident until freed).
 * A review that HandleReview rejects gets status GK_ERR_REVIEW in statuses[i] (may be NULL) and evaluates to nothing. */
int gk_table_create(gk_engine* e, const gk_review_in* reviews, size_t n, uint32_t flags, int32_t* statuses, gk_table** out);
void gk_table_free(gk_table* t);
/* what building the table cost on the host (the PCIe-inclusive, end-to-end leg of SURVEY.md section 8(d)) */
typedef struct {
  uint64_t n_reviews, n_rows, json_bytes, heap_bytes, device_bytes;
  double flatten_s;        /* parse + HandleReview normalisation + flatten + row-group index, wall clock over host_threads */
  double upload_s;         /* host -> HBM */
  uint32_t host_threads, reserved;
  uint64_t fast_reviews;   /* reviews flattened by the one-pass JSON -> rows path (the others took parse + normalise + flatten) */
  uint64_t digest;         /* content digest of the table, only with GK_TABLE_DIGEST=1 in the environment (test aid) */
} gk_table_stats;
int gk_table_get_stats(const gk_table* t, gk_table_stats* out);
#define GK_TABLE_KEEP_DOCS 1u   /* keep parsed reviews on the host so violations can be rendered to messages */
#define GK_TABLE_PROCESS_AUDIT 4u     /* apply the process excluder (gk_excluder_replace) for process "audit" / "webhook":     */
#define GK_TABLE_PROCESS_WEBHOOK 8u   /* excluded reviews keep their slot, hold no rows and get status GK_REVIEW_EXCLUDED      */
#define GK_REVIEW_EXCLUDED 1          /* statuses[i]: skipped before evaluation (not an error)                                 */
#define GK_TABLE_KEEP_TEXT 16u  /* remember WHERE the reviews' JSON text lives (the gk_review_in array is copied, the text is NOT: the
                                   caller keeps it alive as long as the table) so that gk_table_totals / gk_render / gk_render_error
                                   can parse the few reviews they need on demand -- no parsed copy of a million objects */
#define GK_TABLE_PRUNED 32u     /* the table serves the policy set loaded NOW: it holds rows only for the key paths some loaded constraint
                                   reads, and the ingest walks past sub-documents nothing reads (syntax still checked).  A constraint that
                                   arrives later and reads another path makes it stale: gk_table_eval / gk_table_sweep_sharded / gk_table_totals
                                   then fail with GK_ERR_INVALID ("create it again"), as after a new dictionary predicate.  What pkg/audit's
                                   per-sweep list and the webhook's per-request decode amount to: the objects are read again anyway */
#define GK_TABLE_PRE_MATCHED 64u /* the CALLER ran Matcher.Match (pkg/target/matcher.go:21-42) for every review of the table and wants the
                                   violation sets alone -- Driver.Query's contract (pkg/drivers/k8scel/driver.go:162-251: the constraints handed
                                   over are evaluated, never matched again): every constraint counts as matching every usable review, no
                                   autoreject bit is written, namespace_json / source of the reviews are not looked at.  The caller keeps the
                                   (constraint, review) pairs it matched.  Per call of the admission path: GK_QUERY_PRE_MATCHED */
#define GK_TABLE_RESIDENT 2u    /* the table is evaluated again and again (audit set): the engine may compile a plan variant
                                   whose LDS layout fits this table's array sizes (first evaluation pays the compile) */

/* ---- audit spool (row f4) ---------------------------------------------------------------------------------------------
 * pkg/audit's auditResources lists every kind in chunks and writes each object, as JSON, to
 *     <apiCacheDir>/<Kind>_<folder>/<index>                                   (manager.go:519-551)
 * reviewObjects then re-reads the folders of one kind file by file, looks the object's Namespace up and reviews it as
 * AugmentedUnstructured{Object, Namespace, Source: Original} with the Namespace also as the namespaceObject option
 * (manager.go:667-776).  gk_table_create_spool is that loop's front end: it reads folders <kind>_0 .. <kind>_<folders-1>
 * of `api_cache_dir` (files of a folder in numeric order of their names), attaches to every object the Namespace synced
 * through gk_data_put (Driver.AddData) for its metadata.namespace, and builds ONE table of them (flags as gk_table_create;
 * with GK_TABLE_KEEP_TEXT the TABLE owns the spooled text until gk_table_free).
 * A file that cannot be read or is not a JSON object, and an object whose Namespace is not in the cache, is skipped and
 * counted -- the reference logs the error and continues with the next file (manager.go:688-704); so is a folder that cannot
 * be opened ("Unable to get files from directory", manager.go:680-684: getFilesFromDir's error is logged, the loop goes on).
 * info->names[i] = "<Kind>_<folder>/<index>" of review i. */
typedef struct {
  uint64_t n_files, n_reviews, n_unreadable, n_namespace_missing, bytes;
  const char* const* names;   /* [n_reviews] */
  uint64_t n_folders_missing;  /* folders <kind>_<i>, i < folders, that do not exist / cannot be opened */
  const int32_t* statuses;     /* [n_reviews] as gk_table_create's: GK_OK, GK_REVIEW_EXCLUDED (process excluder) or HandleReview's error */
  uint64_t n_rejected, n_excluded;   /* how many of them are errors ("Unable to review object from file") / GK_REVIEW_EXCLUDED */
} gk_spool_info;
int gk_table_create_spool(gk_engine* e, const char* api_cache_dir, const char* kind, uint32_t folders, uint32_t flags,
                          gk_spool_info** info, gk_table** out);
void gk_spool_info_free(gk_spool_info* info);

typedef struct {
  uint32_t n_reviews, n_constraints, n_tiles;   /* n_tiles = ceil(n_reviews / 64) */
  const uint32_t* constraint_ids;   /* [n_constraints] ids as returned by gk_constraint_add, in bitmap-row order */
  const uint64_t* viol;      /* [n_constraints][n_tiles]: bit (r%64) of word (r/64): constraint applies AND is violated */
  const uint64_t* err;       /* same layout: Matcher.Match error (autoreject, "unable to match constraints: ...") */
  const uint64_t* match;     /* same layout, only with GK_EVAL_WANT_MATCH */
  const uint64_t* too_big;   /* [n_tiles]: reviews beyond engine limits (reported, never guessed) */
  const uint32_t* counts;    /* [n_constraints] violating reviews per constraint */
  const uint32_t* list;      /* list_len pairs (bitmap row, review): ballot/prefix-sum compacted violation list */
  uint32_t list_len, list_total;
  uint32_t n_overflow;       /* reviews evaluated by the large-capacity kernel variant */
  float kernel_ms;           /* device time of the evaluation kernels (HIP events on the launch stream) */
  float fast_kernel_ms;      /* ... of the dominant LDS kernel alone */
  uint64_t algo_bytes;       /* algorithmic bytes of this launch (DESIGN.md "Roofline accounting") */
  uint64_t n_rows;           /* rows in the table */
  uint32_t n_launches;       /* launches averaged in fast_kernel_ms (GK_EVAL_ASYNC enqueues without collecting) */
  uint32_t lds_bytes;        /* accumulator LDS bytes per tile of the plan variant that ran */
  const void* d_viol;        /* device pointers to the same bitmaps / counts, valid until the table's next launch;     */
                             /* d_counts == d_viol + n_constraints*n_tiles*8: one contiguous [bitmap | counts] range:  */
  const void* d_err;         /* lets the caller hand them to RCCL (all-gather of per-shard violation bitmaps)      */
  const void* d_counts;
  uint64_t n_rows_read;      /* rows in the segments whose key path carries predicates of the current plan */
  uint64_t algo_bytes_once;  /* algo_bytes with the TABLE counted once: a constraint set that needs several plan groups walks the
                                table once per group today; rows / string headers / review flags bound by ANY group count once
                                here (chunk lists, plan tables and bitmaps are per group).  == algo_bytes for a single group */
  uint32_t n_plan_groups;
  /* Reviews beyond the device's limits that the engine answered with its own exact host evaluator instead of refusing them (their bits
   * in viol / err / match are set from that evaluation, their too_big bit is clear): the match layer from a stripped copy of the review
   * on the device, the template's violation set by the evaluator that renders the messages.  The reference has no such limit
   * (pkg/audit/manager.go:591-642 reviews every object).  What is still set in too_big could not be completed (no text at hand, an
   * evaluation error): the caller fails closed for those. */
  uint32_t n_host_evaluated;
  const uint32_t* host_evaluated;   /* [n_host_evaluated] review indices */
  uint64_t kernel_text_hash;        /* FNV-64 of the source text of the plan-specialised dominant kernel that ran (several plan groups: their
                                       hashes combined; 0: the bytecode kernel ran).  Names the kernel a measurement was taken on: bench.py
                                       reports PMC traffic only from passes stamped with the hash of the kernel it timed */
} gk_eval_out;

#define GK_EVAL_WANT_MATCH 1u
#define GK_EVAL_NO_DOWNLOAD 2u   /* leave results on the device (timing runs) */
#define GK_EVAL_WANT_LIST 4u
#define GK_EVAL_ASYNC 8u         /* enqueue one launch on the default stream and return (out = NULL); the next call
                                    without this flag synchronises, and reports the average kernel time per launch */
#define GK_EVAL_COLLECT 16u      /* do not launch: synchronise and collect the results of the pending GK_EVAL_ASYNC launches */
#define GK_EVAL_KERNEL_ONLY 64u  /* with GK_EVAL_ASYNC: enqueue the dominant kernel alone, without the per-constraint totals kernel behind it (timing runs:
                                    consecutive launches of that kernel under ONE event pair -- fast_kernel_ms of the collecting call is then the kernel's
                                    back-to-back average, free of the event records a pair per launch puts between the launches); the totals the
                                    collecting call reports are not those of these launches */
#define GK_EVAL_DEVICE_ONLY 128u /* the device's answer alone: reviews beyond the device's limits stay in too_big instead of being answered by the
                                    engine's host evaluator (gk_eval_out.host_evaluated stays empty); GK_HOST_EVAL=0 in the environment of the
                                    process does the same for every call */
#define GK_EVAL_TIME_EACH 32u    /* with GK_EVAL_ASYNC: an event pair around THIS launch (fast_kernel_ms of the collecting call = the sum of the
                                    isolated kernel durations / launches) instead of one pair around all pending launches, gaps included */

/* The hot path: every loaded constraint x every review of the table -- Match (a3-a7) + violation predicate (a8/a9).
 * Replaces the per-object Client.Review loops at pkg/audit/manager.go:591-642,706-719 and pkg/webhook/policy.go:826. */
int gk_table_eval(gk_engine* e, gk_table* t, uint32_t flags, gk_eval_out** out);
void gk_eval_free(gk_eval_out* o);

/* Driver.Query result rendering for one (constraint, review) pair flagged in `viol`: evaluates the template on the
 * host for that pair only and returns a JSON array [{"msg": "...", "details": ...}] (types.Result.Msg / Metadata).
 * Needs GK_TABLE_KEEP_DOCS or GK_TABLE_KEEP_TEXT. */
int gk_render(gk_engine* e, gk_table* t, uint32_t constraint_id, uint32_t review, char** json_out);
/* The autoreject result for a pair flagged in `err` (frameworks Client.Review turns a Matcher.Match error into a
 * result; message pinned by test/gator/test/test.bats:301): [{"msg": "unable to match constraints: ...", ...}] */
int gk_render_error(gk_engine* e, gk_table* t, uint32_t constraint_id, uint32_t review, char** json_out);
void gk_free(void* p);

/* ---- audit aggregation (row a11) ---------------------------------------------------------------------------------
 * pkg/audit/manager.go:885-941 keeps, per constraint, the `--constraint-violations-limit` (default 20) SMALLEST
 * violations by (group, version, kind, namespace, name, message, action) (LimitQueue, manager.go:112-203).  After a
 * gk_table_eval, gk_table_topk selects on the device, for every bitmap row, the `k` violating reviews with the smallest
 * object key (ties on the k-th key included, so the host-side message tie-break stays exact): the caller renders only
 * those k x C pairs instead of every violating pair.  `reviews` is [n_constraints][stride], counts[c] entries valid,
 * ascending by object key.  overflow[c] != 0: more ties than `stride` holds -- walk that bitmap row on the host. */
typedef struct {
  uint32_t n_constraints, stride;
  const uint32_t* constraint_ids;
  const uint32_t* counts;
  const uint32_t* reviews;
  const uint32_t* overflow;
} gk_topk_out;
int gk_table_topk(gk_engine* e, gk_table* t, uint32_t k, gk_topk_out** out);
void gk_topk_free(gk_topk_out* o);

/* Result-level totals of the table's most recent evaluation (row a11): pkg/audit/manager.go:902 increments
 * totalViolationsPerConstraint once per types.Result, and one violating (constraint, object) pair yields as many
 * results as the template's violation set has distinct {msg, details} members.  `pairs` = popcount of the bitmap row,
 * `results` = what the reference's counters hold.  The device answers, per violating pair, whether the template CAN yield more
 * than one result for that review (two rule bodies / parameter alternatives hold, or one holds for two elements of an
 * iterated collection); a pair it does not flag counts exactly one result, the flagged ones (`rendered_pairs`) are rendered on
 * the host workers and counted.  Needs GK_TABLE_KEEP_DOCS or GK_TABLE_KEEP_TEXT (then only the rendered reviews are parsed,
 * once each). */
typedef struct {
  uint32_t n_constraints;
  const uint32_t* constraint_ids;
  const uint64_t* results;
  const uint64_t* pairs;
  uint64_t rendered_pairs;   /* violating pairs whose result count needed the host renderer */
} gk_totals_out;
int gk_table_totals(gk_engine* e, gk_table* t, gk_totals_out** out);
void gk_totals_free(gk_totals_out* o);

/* ---- multi-GPU audit sweep (row e) ----------------------------------------------------------------------------------
 * One process / engine per GPU; the audited objects are block-sharded over the ranks (each rank flattens its own shard
 * into a resident table); policies are replicated.  gk_comm_init joins the engine to an RCCL communicator (rank 0 makes
 * the id with gk_comm_unique_id and distributes it out of band, e.g. through the launcher's store).  One sweep step
 * (gk_table_sweep_sharded) = local evaluation of the shard, then ONE collective on the same stream: an in-place
 * ncclAllGather of every shard's slot [violation bitmap | counts | what the bitmaps cannot say], so that every rank ends
 * with the full constraints x objects bitmap; the global int64 totals are the sums over the gathered slot tails (a kernel on
 * every rank, no second collective).  Shards may differ in size: all slots use the bitmap stride of the largest shard
 * (`stride_tiles`), a shard's own words come first.
 * A constraint set that needs several plan groups (more than 64 distinct formulas) runs one evaluation + exchange per
 * group; `gathered` is then the merged host copy (all groups' rows in constraint_ids order) and `d_gathered` is NULL.
 * Collective: every rank must call it. */
#define GK_COMM_ID_BYTES 128
int gk_comm_unique_id(char id[GK_COMM_ID_BYTES]);
int gk_comm_init(gk_engine* e, const char id[GK_COMM_ID_BYTES], int rank, int world);
void gk_comm_destroy(gk_engine* e);
/* rank and size of the engine's communicator as RCCL itself reports them (ncclCommUserRank / ncclCommCount): a launcher that believes
 * it started N ranks can check that the exchange really spans N GPUs.  GK_ERR_INVALID before gk_comm_init. */
int gk_comm_info(gk_engine* e, int32_t* rank, int32_t* world);
typedef struct {
  uint32_t world, rank, n_constraints, stride_tiles;
  uint64_t slot_bytes;               /* bytes per rank in the gathered buffer: [n_constraints][stride_tiles] u64 | [n_constraints] u32 violating
                                        objects | tail (autoreject counts, fail-closed counts; engine-internal) | pad */
  const uint32_t* constraint_ids;    /* [n_constraints] bitmap-row order */
  const uint32_t* shard_reviews;     /* [world] objects per shard */
  const int64_t* totals;             /* [n_constraints] violating (constraint, object) pairs over ALL shards */
  const uint64_t* gathered;          /* host copy of the gathered buffer (only with GK_SHARD_DOWNLOAD) */
  const void* d_gathered;            /* the same on the device, valid until the table's next sweep */
  float kernel_ms, fast_kernel_ms;
  uint32_t n_overflow;
  /* Fail closed (pkg/audit/manager.go:622-625 logs and skips a failed Review; it never counts it as clean): what the
   * violation bitmaps cannot say, summed over ALL shards (it travels in the slot tails of the same all-gather).  A caller that wants the reference's
   * answer reviews these objects on the stock driver. */
  const int64_t* err_totals;         /* [n_constraints] autoreject pairs: Matcher.Match returned an error (one Result each) */
  int64_t beyond_limits;             /* reviews a plan group could not evaluate (beyond the engine's limits; counted per plan group) */
  int64_t not_evaluated;             /* reviews HandleReview rejected when their shard's table was built (GK_ERR_REVIEW) */
  /* the exchange step, so that a scaling run can be read: */
  float exchange_ms;                 /* duration of the all-gather alone (HIP events around it on its stream) in THIS call's collecting exchange;
                                        0 when the call handed out an enqueue-only pass (GK_SHARD_COLLECT: those passes carry no events) */
  uint32_t exchange_overlapped;      /* 1: consecutive enqueue-only passes run their all-gather on the exchange stream, under the next sweep */
  uint64_t exchange_bytes_inbound;   /* bytes this rank receives per sweep: (world - 1) x slot_bytes (summed over plan groups) */
} gk_shard_out;
#define GK_SHARD_DOWNLOAD 1u
/* GK_SHARD_ENQUEUE: put one sweep + its exchange step onto the table's stream and return at once (`out` is not written and may
 * be NULL) -- back-to-back sweeps of a resident shard without a host round trip per sweep.  A call without the flag collects:
 * it runs one more sweep, waits, and returns the answer.  Only that collecting sweep re-runs reviews that overflowed the
 * dominant kernel's element capacity; the engine's limits apply as before.  Collective like every call of this function. */
#define GK_SHARD_ENQUEUE 2u
/* GK_SHARD_COLLECT (without GK_SHARD_ENQUEUE): wait for the enqueue-only passes and return the answer of the LAST of them
 * instead of sweeping once more.  Falls back to an ordinary collecting sweep when nothing was enqueued, when the constraint
 * set needs several plan groups, or when that pass left reviews -- on any rank -- to the large-capacity re-run.  An enqueue-only
 * pass is four enqueues (sweep, slot-tail kernel, all-gather, totals); GK_SHARD_GRAPH=1 replays a single-plan-group pass as one
 * captured graph instead.  With GK_SHARD_OVERLAP=1 (opt-in: tested on the host-side emulation only, it has never met a second GPU)
 * consecutive enqueue-only passes alternate between two slot buffers and their all-gathers run on an exchange stream: the exchange
 * of one pass overlaps the sweep of the next; `d_gathered` / `gathered` of the collecting call are those of the last pass. */
#define GK_SHARD_COLLECT 4u
int gk_table_sweep_sharded(gk_engine* e, gk_table* t, uint32_t flags, gk_shard_out** out);
void gk_shard_free(gk_shard_out* o);

/* ---- resident-set audit (row f2) -----------------------------------------------------------------------------------
 * pkg/audit's auditFromCache reviews, one by one, exactly the objects that were synced into the driver through
 * Driver.AddData (pkg/audit/manager.go:591-642; pkg/cachemanager/cachemanager.go:310-343).  gk_data_put / gk_data_remove
 * therefore also maintain a RESIDENT SET: every synced object, flattened in HBM.  gk_resident_sweep brings the device
 * tables up to date -- only objects added or changed since the last sweep are flattened (into a new chunk); replaced
 * and removed objects are masked out; a changed Namespace re-flattens the objects living in it; chunks are compacted when
 * the masked-out slots outnumber the live ones -- and evaluates every constraint over the set in one launch per chunk.
 * The review of a resident object is the one auditFromCache builds: AugmentedUnstructured{Object, Namespace: the synced
 * Namespace it lives in (or none)}, no Source, the same Namespace as the namespaceObject option.
 * Afterwards the per-object answers are a column of the bitmaps: gk_resident_review (by path) and gk_query (for a review
 * that is byte for byte such an object + Namespace) return them without flattening or launching anything. */
typedef struct {
  uint64_t n_objects;              /* live objects in the set */
  uint32_t n_constraints, n_chunks;
  const uint32_t* constraint_ids;  /* [n_constraints] */
  const uint64_t* pairs;           /* [n_constraints] violating (constraint, object) pairs */
  const uint64_t* results;         /* [n_constraints] results (manager.go:902 counts these), only with GK_SWEEP_RESULT_TOTALS */
  uint64_t flattened;              /* objects (re)flattened by THIS sweep */
  uint64_t beyond_limits;          /* live objects the engine refuses to evaluate (fail closed: review them on the CPU driver) */
  double sync_s, eval_s;           /* host flatten + upload of the new chunk / device evaluation + download of the bitmaps */
} gk_sweep_out;
#define GK_SWEEP_RESULT_TOTALS 1u
int gk_resident_sweep(gk_engine* e, uint32_t flags, gk_sweep_out** out);
void gk_sweep_free(gk_sweep_out* o);
/* results of one swept object, in gk_query's JSON; GK_ERR_NOT_FOUND when the object is unknown or its answer is stale */
int gk_resident_review(gk_engine* e, const char* const* path, size_t npath, char** results_json);
/* the same for a caller that matched itself (gk_query_ex2's arguments).  With GK_QUERY_PRE_MATCHED the sweep's bitmaps -- match AND
 * violation -- cannot answer: the object's resident text is evaluated again through the admission batcher (one flatten + its share of
 * a launch), pre-matched, with the review auditFromCache builds (the synced Namespace as namespaceObject); GK_ERR_NOT_FOUND when the
 * object is not resident.  Without the flag: gk_resident_review restricted to the listed constraints. */
int gk_resident_review_ex(gk_engine* e, const char* const* path, size_t npath, const uint32_t* constraint_ids, size_t n_constraints,
                          uint32_t flags, char** results_json);

/* ---- admission path: micro-batched Driver.Query (row f1) ----------------------------------------------------------
 * Driver.Query evaluates ONE review (pkg/drivers/k8scel/driver.go:162-251) and the validating webhook calls it from up
 * to GOMAXPROCS request goroutines at once (pkg/webhook/policy.go:142-146, 748-757).  gk_query may be called from any
 * number of threads: the engine's batcher thread coalesces the calls that arrive within `window_us` (or `max_batch` of
 * them) into ONE flattened table and ONE launch; each caller then renders its own results from its column of the batch's
 * bitmaps, in its own thread:
 *   [{"constraint": <id from gk_constraint_add>, "msg": "...", "details": {...}[, "autoreject": true]}, ...]
 * for every loaded constraint that matches the review and is violated (or whose Matcher.Match failed: autoreject).  The
 * caller keeps the results of the constraints it asked about (Driver.Query's `constraints` argument).
 * GK_ERR_REVIEW: HandleReview rejected the review; GK_ERR_LIMIT: beyond the engine's limits (fail closed). */
typedef struct {
  uint32_t max_batch;    /* reviews per launch (0 = 64) */
  uint32_t window_us;    /* how long the first call of a batch waits for company */
  uint32_t workers;      /* batches in progress at once: one is flattened while another's launch is on the device (0 = 2) */
  uint32_t reserved;
} gk_batch_opts;
typedef struct {
  uint32_t batch_size;   /* reviews that shared the launch */
  uint32_t reserved;
  double queue_us;       /* arrival -> batch start */
  double device_us;      /* device time of the batch's kernels */
  double total_us;       /* arrival -> results ready */
} gk_query_stats;
int gk_batcher_start(gk_engine* e, const gk_batch_opts* opts);   /* optional: gk_query starts it with the defaults (64, 200 us) */
void gk_batcher_stop(gk_engine* e);
int gk_query(gk_engine* e, const gk_review_in* review, char** results_json, gk_query_stats* stats);
/* gk_query plus QueryResponse.Trace (pkg/drivers/k8scel/driver.go:162-251: `Trace *string` when the review asks for tracing or the
 * driver was built with it): flags & GK_QUERY_TRACE, or an engine created with GK_OPT_TRACE, makes *trace_out a text (gk_free) that
 * says how the review was answered -- the batch it shared, where it was evaluated (device / host evaluator / resident sweep) and,
 * per loaded constraint, whether it applied and what it yielded; NULL otherwise.  trace_out may be NULL. */
#define GK_QUERY_TRACE 1u
int gk_query_ex(gk_engine* e, const gk_review_in* review, uint32_t flags, char** results_json, char** trace_out, gk_query_stats* stats);
/* Driver.Query as the reference defines it (pkg/drivers/k8scel/driver.go:162-251; frameworks drivers/rego Query): the caller --
 * Client.Review -- has ALREADY run Matcher.Match (pkg/target/matcher.go:21-42) and hands over exactly the constraints that matched;
 * the driver evaluates those and never matches again.  A Go driver cannot do anything else: gkReview.namespace and gkReview.source,
 * which Match reads, are unexported and have no accessor (pkg/target/review.go:16-21), so it cannot pass them down.
 *   constraint_ids / n_constraints: the ids (gk_constraint_add) of Driver.Query's `constraints`; results of other constraints are
 *     not returned.  NULL = every loaded constraint.  n_constraints == 0 with a non-NULL pointer: "[]" at once (the Rego driver
 *     returns early on an empty list).  An id that is not loaded: GK_ERR_NOT_FOUND ("unknown constraint template validator").
 *   GK_QUERY_PRE_MATCHED: for the listed constraints the VIOLATION sets are returned whatever this engine's own match layer says
 *     about the review (a `match.source: Generated` constraint, a namespaceSelector on a Namespace the engine has never been
 *     synced: the caller matched them, the caller knows); no autoreject rows; review->namespace_json and review->source are not
 *     looked at; the review is never answered from the resident sweep's match-and-violation bitmaps.  Without the flag the call is
 *     gk_query_ex restricted to the listed constraints (the engine matches: the batch / audit callers that own namespace + source).
 * Calls with and without the flag share batches: the mode is a per-review bit of the batch's table (GK_TABLE_PRE_MATCHED). */
#define GK_QUERY_PRE_MATCHED 2u
int gk_query_ex2(gk_engine* e, const gk_review_in* review, const uint32_t* constraint_ids, size_t n_constraints, uint32_t flags,
                 char** results_json, char** trace_out, gk_query_stats* stats);

/* ---- plan specialisation in the background -----------------------------------------------------------------------------
 * After AddTemplate / AddConstraint (drivers.Driver, pkg/drivers/k8scel/driver.go:74-160) the next Query must not wait for a
 * compiler: the reference's webhook gives a review 3 s by default and turns a Query error into an HTTP 500
 * (pkg/webhook/policy.go:208-211).  The plan-specialised build of the dominant kernel (hiprtc, ~2 s) therefore runs on a
 * background thread for admission batches; until its module is loaded the generic bytecode kernel -- same answers, parity
 * tested -- serves them.  Resident (audit) tables wait for the build.  Code objects are cached by the hash of their source
 * text (in memory; on disk under $GK_JIT_CACHE_DIR).
 * gk_jit_quiesce waits for every build in flight (tests, orderly shutdown); gk_jit_cache_stats reports builds served from
 * the cache / compiled by hiprtc since the process started. */
void gk_jit_quiesce(void);
void gk_jit_cache_stats(uint64_t* cache_hits, uint64_t* compiles);
/* The disk cache is on by default: $GK_JIT_CACHE_DIR, else $XDG_CACHE_HOME/gkgpu-jit, else $HOME/.cache/gkgpu-jit, else
 * /tmp/gkgpu-jit-<uid>; GK_JIT_CACHE_DIR="" (or "off") disables it.  Files are named by the hash of the kernel's source text, the
 * target (gfx950) and the hiprtc version.  gk_jit_cache_dir: the directory in use ("" = none).  gk_jit_cache_drop_memory forgets the
 * code objects held in memory, so that the next build of a known text is served from the file, as after a restart of the process. */
const char* gk_jit_cache_dir(void);
void gk_jit_cache_drop_memory(void);

/* Test and tuning hooks that used to be environment switches read on the compile path (nothing there reads the environment any more).
 * Not for deployments.  Keys: "fold_match_labels" (0 | 1: the match formulas' label tests become dictionary bits as well -- how
 * tests/test_pruned.py reaches the totals plans a pruned table cannot answer); "group_max" (n > 0: no plan group of more than n
 * constraints, takes effect at the next policy change -- the several-group path, which a set of more than 256 distinct violation
 * formulas takes by itself, tested with small corpora; 0 = as many as fit); "dict_facts" (0: the tests on a review's non-iterated leaves
 * keep a row per leaf instead of sharing the review facts row, for the policies loaded while it is 0).  GK_ERR_NOT_FOUND for an unknown key. */
int gk_debug_set(const char* key, int64_t value);

/* Debug: the compiled plan as text (Driver.Dump, pkg/drivers/k8scel/driver.go:253). */
int gk_dump(gk_engine* e, char** text_out);

#ifdef __cplusplus
}
#endif
#endif /* GKGPU_H */
