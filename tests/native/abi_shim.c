/* abi_shim.c -- the call order of the cgo shim of INTEGRATION.md, in plain C99 against include/gkgpu.h.
 *
 * What cgo would bind is exactly this: no C++ types, no Python, every input BORROWED for the call (the buffers below are freed or
 * overwritten the moment a call returns), every output released with its own *_free.  The harness walks the shim's life cycle
 *
 *     gk_engine_create(gk_opts: flags + disabled builtins)        rego.New(args...)            main.go:424-486
 *     gk_template_add / gk_constraint_add / gk_data_put           Driver.AddTemplate / AddConstraint / AddData
 *     N threads x gk_query_ex2 (blocking, micro-batched)          Driver.Query from the webhook's goroutines, pkg/webhook/policy.go:142-146,826:
 *       odd threads the way the Go shim calls it -- the constraint ids Client.Review matched, GK_QUERY_PRE_MATCHED, no namespace, no
 *       source (a Go driver cannot read either: pkg/target/review.go:16-21) --, even threads let the engine match (gk_query_ex)
 *       ... while another thread REPLACES the template and adds / removes a constraint      (Query must be re-entrant and never see
 *           a half-replaced policy set: pkg/drivers/k8scel/driver.go:61,131,140,168-169 guards the same with a RWMutex)
 *     gk_table_create / gk_table_eval / gk_render                 the audit's batch path, pkg/audit/manager.go:706-719
 *     gk_template_remove / gk_batcher_stop / gk_engine_destroy
 *
 * and checks every answer.  Built by tests/test_abi.py with  gcc -std=c99 -Wall -Wextra -pedantic -Werror  against the TEST-ONLY CPU
 * build of the engine (no GPU in the build container) and, on the GPU box, against libgkgpu.so itself.
 * usage: abi_shim [threads] [queries per thread]      exit 0 = every check passed */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gkgpu.h"

#define CHECK(cond, what)                                                                  \
  do {                                                                                     \
    if (!(cond)) { fprintf(stderr, "abi_shim: %s (line %d): %s\n", what, __LINE__, gk_last_error()); exit(1); } \
  } while (0)

static const char* REGO_V1 =
    "package k8srequiredlabels\n"
    "violation[{\"msg\": msg, \"details\": {\"missing\": l}}] {\n"
    "  l := input.parameters.labels[_]\n"
    "  not input.review.object.metadata.labels[l]\n"
    "  msg := sprintf(\"v1: label %v is missing\", [l])\n"
    "}\n";
static const char* REGO_V2 =
    "package k8srequiredlabels\n"
    "violation[{\"msg\": msg, \"details\": {\"missing\": l}}] {\n"
    "  l := input.parameters.labels[_]\n"
    "  not input.review.object.metadata.labels[l]\n"
    "  msg := sprintf(\"v2: label %v is missing\", [l])\n"
    "}\n";
static const char* REGO_PRIV =
    "package k8spriv\n"
    "violation[{\"msg\": msg}] {\n"
    "  c := input.review.object.spec.containers[_]\n"
    "  c.securityContext.privileged\n"
    "  msg := sprintf(\"privileged container %v\", [c.name])\n"
    "}\n";

static gk_engine* E;
static volatile int g_stop;
static uint32_t g_ids[3];   /* need-team, no-priv, gen-only (match.source: Generated) */

/* a heap copy of a string that is POISONED and freed after the call it was lent to: the engine must have copied what it keeps */
static char* lend(const char* s) { char* p = (char*)malloc(strlen(s) + 1); strcpy(p, s); return p; }
static void take_back(char* p) { memset(p, '#', strlen(p)); free(p); }

static void add_template(const char* kind, const char* rego) {
  char *k = lend(kind), *r = lend(rego);
  CHECK(gk_template_add(E, k, r, NULL, 0) == GK_OK, "gk_template_add");
  take_back(k); take_back(r);
}
static uint32_t add_constraint(const char* json) {
  char* j = lend(json);
  uint32_t id = 0;
  CHECK(gk_constraint_add(E, j, strlen(j), &id) == GK_OK, "gk_constraint_add");
  take_back(j);
  return id;
}

typedef struct { int id, n_queries, failures; long results; } worker_arg;

static void* query_worker(void* p) {
  worker_arg* a = (worker_arg*)p;
  int q;
  for (q = 0; q < a->n_queries; q++) {
    char obj[512];
    const int labelled = (a->id + q) % 3 == 0, priv = (a->id + q) % 2 == 0;
    gk_review_in in;
    char *js = NULL, *trace = NULL, *text;
    gk_query_stats st;
    int rc;
    snprintf(obj, sizeof obj,
             "{\"apiVersion\": \"v1\", \"kind\": \"Pod\", \"metadata\": {\"name\": \"p-%d-%d\", \"namespace\": \"prod\"%s}, "
             "\"spec\": {\"containers\": [{\"name\": \"c0\", \"image\": \"nginx\"%s}]}}",
             a->id, q, labelled ? ", \"labels\": {\"team\": \"a\"}" : "", priv ? ", \"securityContext\": {\"privileged\": true}" : "");
    text = lend(obj);
    memset(&in, 0, sizeof in);
    in.kind = GK_REVIEW_OBJECT; in.json = text; in.json_len = strlen(text);
    if (a->id % 2) {
      /* Driver.Query: the caller matched (say: an expansion resultant, Source Generated -- all three constraints apply); the driver
       * is handed their ids and nothing the match layer reads */
      rc = gk_query_ex2(E, &in, g_ids, 3, GK_QUERY_PRE_MATCHED | ((q % 8 == 0) ? GK_QUERY_TRACE : 0u), &js, &trace, &st);
    } else {
      in.source = GK_SRC_ORIGINAL;
      rc = gk_query_ex(E, &in, (q % 8 == 0) ? GK_QUERY_TRACE : 0u, &js, &trace, &st);
    }
    take_back(text);                                   /* borrowed for the call only */
    if (rc != GK_OK || !js) { a->failures++; fprintf(stderr, "abi_shim: thread %d query %d: rc %d: %s\n", a->id, q, rc, gk_last_error()); continue; }
    /* whatever the replacing thread is doing, the answer is one of the two templates' -- never a mixture, never nothing */
    {
      const int has_v1 = strstr(js, "v1: label team is missing") != NULL, has_v2 = strstr(js, "v2: label team is missing") != NULL;
      const char* first_priv = strstr(js, "privileged container c0");
      const int has_priv = first_priv != NULL;
      /* pre-matched callers handed over `gen-only` as well: the privileged pod violates it too, whatever its match block says;
       * callers that let the engine match sent Source Original: `gen-only` does not apply */
      const int n_priv = has_priv + (first_priv && strstr(first_priv + 1, "privileged container c0") != NULL);
      int bad = 0;
      if (labelled ? (has_v1 || has_v2) : (has_v1 == has_v2)) bad |= 1;
      if (n_priv != (priv ? (a->id % 2 ? 2 : 1) : 0)) bad |= 2;
      if (strstr(js, "autoreject") != NULL) bad |= 32;
      if ((q % 8 == 0) && (!trace || !strstr(trace, "gkgpu trace"))) bad |= 4;
      if ((q % 8 != 0) && trace) bad |= 8;
      if (st.batch_size < 1) bad |= 16;
      if (bad) { a->failures++; fprintf(stderr, "abi_shim: thread %d query %d: check %d failed (labelled %d, privileged %d): %s\n", a->id, q, bad, labelled, priv, js); }
      a->results += has_v1 + has_v2 + has_priv;
    }
    gk_free(js);
    if (trace) gk_free(trace);
  }
  return NULL;
}

static void* policy_worker(void* p) {   /* Driver.AddTemplate on a kind that is serving, AddConstraint / RemoveConstraint beside it */
  int round = 0;
  (void)p;
  while (!g_stop) {
    uint32_t id;
    add_template("K8sRequiredLabels", (round & 1) ? REGO_V1 : REGO_V2);
    id = add_constraint("{\"apiVersion\": \"constraints.gatekeeper.sh/v1beta1\", \"kind\": \"K8sRequiredLabels\", \"metadata\": {\"name\": \"extra\"}, "
                        "\"spec\": {\"match\": {\"namespaces\": [\"nowhere\"]}, \"parameters\": {\"labels\": [\"x\"]}}}");
    (void)id;
    CHECK(gk_constraint_remove(E, "K8sRequiredLabels", "extra") == GK_OK, "gk_constraint_remove");
    round++;
  }
  return NULL;
}

int main(int argc, char** argv) {
  const int n_threads = argc > 1 ? atoi(argv[1]) : 32, n_queries = argc > 2 ? atoi(argv[2]) : 40;
  const char* disabled[] = {"http.send", "net.lookup_ip_addr"};
  gk_opts opts;
  gk_batch_opts bo;
  pthread_t* th;
  pthread_t pol;
  worker_arg* args;
  int i, failures = 0;
  long results = 0;

  /* ---- rego.New(rego.Tracing(false), rego.DisableBuiltins(...), rego.Externs("inventory")) */
  memset(&opts, 0, sizeof opts);
  opts.device = 0;
  opts.flags = GK_OPT_GATHER_STATS;
  opts.disabled_builtins = disabled; opts.n_disabled_builtins = 2;
  CHECK(gk_engine_create(&opts, &E) == GK_OK, "gk_engine_create");
  CHECK(gk_host_cpus() >= 1 && gk_version() != NULL, "gk_version");

  /* ---- policies; a template that calls a disabled builtin is a type error, an unknown kind is refused */
  add_template("K8sRequiredLabels", REGO_V1);
  add_template("K8sPriv", REGO_PRIV);
  CHECK(gk_template_add(E, "K8sHttp", "package h\nviolation[{\"msg\": \"x\"}] { http.send({\"method\": \"get\", \"url\": \"u\"}) }\n", NULL, 0) == GK_ERR_REGO, "disabled builtin");
  CHECK(strstr(gk_last_error(), "undefined function http.send") != NULL, "disabled builtin: message");
  g_ids[0] = add_constraint("{\"apiVersion\": \"constraints.gatekeeper.sh/v1beta1\", \"kind\": \"K8sRequiredLabels\", \"metadata\": {\"name\": \"need-team\"}, "
                            "\"spec\": {\"match\": {\"kinds\": [{\"apiGroups\": [\"\"], \"kinds\": [\"Pod\"]}]}, \"parameters\": {\"labels\": [\"team\"]}}}");
  g_ids[1] = add_constraint("{\"apiVersion\": \"constraints.gatekeeper.sh/v1beta1\", \"kind\": \"K8sPriv\", \"metadata\": {\"name\": \"no-priv\"}, \"spec\": {}}");
  g_ids[2] = add_constraint("{\"apiVersion\": \"constraints.gatekeeper.sh/v1beta1\", \"kind\": \"K8sPriv\", \"metadata\": {\"name\": \"gen-only\"}, "
                            "\"spec\": {\"match\": {\"source\": \"Generated\"}}}");
  {   /* Driver.Query with an empty constraint list answers at once; an id that is not loaded is "unknown constraint template validator" */
    gk_review_in in;
    char* js = NULL;
    const uint32_t nobody = 4242;
    const char* pod = "{\"apiVersion\": \"v1\", \"kind\": \"Pod\", \"metadata\": {\"name\": \"p\"}}";
    memset(&in, 0, sizeof in);
    in.kind = GK_REVIEW_OBJECT; in.json = pod; in.json_len = strlen(pod);
    CHECK(gk_query_ex2(E, &in, g_ids, 0, GK_QUERY_PRE_MATCHED, &js, NULL, NULL) == GK_OK && js && strcmp(js, "[]") == 0, "gk_query_ex2 with no constraints");
    gk_free(js);
    CHECK(gk_query_ex2(E, &in, &nobody, 1, GK_QUERY_PRE_MATCHED, &js, NULL, NULL) == GK_ERR_NOT_FOUND, "gk_query_ex2 with an unknown id");
  }
  {
    uint32_t id = 0;
    const char* orphan = "{\"apiVersion\": \"constraints.gatekeeper.sh/v1beta1\", \"kind\": \"K8sNoSuchKind\", \"metadata\": {\"name\": \"o\"}, \"spec\": {}}";
    CHECK(gk_constraint_add(E, orphan, strlen(orphan), &id) == GK_ERR_NOT_FOUND, "constraint without a template");
  }
  {   /* Driver.AddData: a Namespace (feeds the namespace cache), path as K8sValidationTarget.ProcessData builds it */
    const char* path[] = {"cluster", "v1", "Namespace", "prod"};
    char* ns = lend("{\"apiVersion\": \"v1\", \"kind\": \"Namespace\", \"metadata\": {\"name\": \"prod\", \"labels\": {\"env\": \"prod\"}}}");
    CHECK(gk_data_put(E, path, 4, ns, strlen(ns)) == GK_OK, "gk_data_put");
    take_back(ns);
  }

  /* ---- the webhook: n_threads callers in gk_query_ex while the policy thread replaces the template under them */
  memset(&bo, 0, sizeof bo);
  bo.max_batch = 64; bo.window_us = 200; bo.workers = 2;
  CHECK(gk_batcher_start(E, &bo) == GK_OK, "gk_batcher_start");
  th = (pthread_t*)calloc((size_t)n_threads, sizeof *th);
  args = (worker_arg*)calloc((size_t)n_threads, sizeof *args);
  CHECK(pthread_create(&pol, NULL, policy_worker, NULL) == 0, "pthread_create");
  for (i = 0; i < n_threads; i++) { args[i].id = i; args[i].n_queries = n_queries; CHECK(pthread_create(&th[i], NULL, query_worker, &args[i]) == 0, "pthread_create"); }
  for (i = 0; i < n_threads; i++) { pthread_join(th[i], NULL); failures += args[i].failures; results += args[i].results; }
  g_stop = 1;
  pthread_join(pol, NULL);
  CHECK(failures == 0, "a query returned a wrong or mixed answer");
  CHECK(results > 0, "no results at all");

  /* ---- the audit's batch path: a table of reviews whose text goes away right after gk_table_create */
  {
    enum { N = 70 };
    gk_review_in in[N];
    char* texts[N];
    int32_t statuses[N];
    gk_table* t = NULL;
    gk_eval_out* ev = NULL;
    gk_topk_out* top = NULL;
    uint32_t row, viol_pairs = 0;
    memset(in, 0, sizeof in);
    for (i = 0; i < N; i++) {
      char obj[384];
      snprintf(obj, sizeof obj, "{\"apiVersion\": \"v1\", \"kind\": \"Pod\", \"metadata\": {\"name\": \"a-%d\", \"namespace\": \"prod\"%s}, \"spec\": {\"containers\": [{\"name\": \"c0\", \"image\": \"i\"%s}]}}",
               i, i % 2 ? ", \"labels\": {\"team\": \"t\"}" : "", i % 5 == 0 ? ", \"securityContext\": {\"privileged\": true}" : "");
      texts[i] = lend(obj);
      in[i].kind = GK_REVIEW_OBJECT; in[i].source = GK_SRC_ORIGINAL; in[i].json = texts[i]; in[i].json_len = strlen(texts[i]);
    }
    CHECK(gk_table_create(E, in, N, GK_TABLE_KEEP_DOCS | GK_TABLE_PROCESS_AUDIT, statuses, &t) == GK_OK, "gk_table_create");
    for (i = 0; i < N; i++) { take_back(texts[i]); CHECK(statuses[i] == GK_OK, "review status"); }
    memset(in, 0, sizeof in);
    CHECK(gk_table_eval(E, t, GK_EVAL_WANT_MATCH, &ev) == GK_OK, "gk_table_eval");
    CHECK(ev->n_reviews == N && ev->n_constraints == 3 && ev->n_tiles == 2, "gk_eval_out shape");
    for (row = 0; row < ev->n_constraints; row++) viol_pairs += ev->counts[row];
    CHECK(viol_pairs == 35 + 14, "violating pairs");          /* 35 pods without the label, 14 privileged */
    for (row = 0; row < ev->n_constraints; row++) {
      uint32_t r;
      for (r = 0; r < N; r++) {
        if ((ev->viol[(size_t)row * ev->n_tiles + r / 64] >> (r % 64)) & 1u) {
          char* msg = NULL;
          CHECK(gk_render(E, t, ev->constraint_ids[row], r, &msg) == GK_OK && msg && msg[0] == '[', "gk_render");
          CHECK(strstr(msg, "\"msg\"") != NULL, "rendered message");
          gk_free(msg);
        }
      }
    }
    CHECK(gk_table_topk(E, t, 20, &top) == GK_OK && top->n_constraints == 3, "gk_table_topk");
    gk_topk_free(top);
    gk_eval_free(ev);
    gk_table_free(t);
  }

  /* ---- Driver.RemoveTemplate takes the kind's constraints with it; the engine goes last */
  CHECK(gk_template_remove(E, "K8sPriv") == GK_OK, "gk_template_remove");
  CHECK(gk_template_remove(E, "K8sPriv") == GK_ERR_NOT_FOUND, "gk_template_remove twice");
  {
    char* dump = NULL;
    CHECK(gk_dump(E, &dump) == GK_OK && dump, "gk_dump");
    gk_free(dump);
  }
  gk_batcher_stop(E);
  gk_jit_quiesce();
  gk_engine_destroy(E);
  free(th); free(args);
  printf("abi_shim ok: %d threads x %d queries under template replacement, %ld results\n", n_threads, n_queries, results);
  return 0;
}
