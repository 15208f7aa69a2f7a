/* gksynth.h -- native synthetic-workload generator exported by libgkgpu.so (bench / test plumbing, NOT part of the
 * drivers.Driver boundary of gkgpu.h).  Generates SURVEY.md section 8(d)'s synthetic cluster objects -- the same objects
 * as gatekeeper_amd/synth.py, pinned object by object in tests/test_synth.py -- as JSON text, already laid out as the
 * gk_review_in array gk_table_create takes (AugmentedUnstructured{Object, Namespace, Source: Original}, the shape
 * pkg/audit builds at pkg/audit/manager.go:694-713). */
#ifndef GKSYNTH_H
#define GKSYNTH_H
#include <stddef.h>
#include <stdint.h>

#include "gkgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gk_synth_batch gk_synth_batch;

/* objects [start, start + n) of stream `seed`; mixed = 0: Pods only (configs[1]); 1: Pod / Deployment / Namespace /
 * Service / ConfigMap mix (configs[2]); 2: the Pods of 0, each wrapped in the admissionv1.AdmissionRequest (CREATE) the
 * validating webhook receives (review kind GK_REVIEW_ADMISSION_REQUEST, synth.py admission_request_for).  namespace_jsons: the 100 Namespace objects in synth.py NAMESPACES order (the
 * review's Namespace is looked up by the object's metadata.namespace) or NULL for none.
 * mixed | 16: HIGH CARDINALITY -- every container gets an image tag and a name unique in the stream (the per-value memos of the ingest's
 * dictionary expressions then never hit; the default vocabulary of a dozen images always does).  Ingest measurements only: the Python
 * generator (synth.py) does not mirror it. */
int gk_synth_batch_create(uint64_t seed, uint64_t start, uint64_t n, int mixed, const char* const* namespace_jsons, size_t n_namespaces,
                          gk_synth_batch** out);
const gk_review_in* gk_synth_batch_reviews(const gk_synth_batch* b);
size_t gk_synth_batch_size(const gk_synth_batch* b);
uint64_t gk_synth_batch_json_bytes(const gk_synth_batch* b);
void gk_synth_batch_free(gk_synth_batch* b);

/* Admission load generator (row f1 measurement): `threads` native threads each call gk_query `per_thread` times on the
 * batch's reviews (round robin) -- the shape of the validating webhook's request goroutines, pkg/webhook/policy.go:142-146.
 * Latency = arrival -> results ready as reported by gk_query_stats.total_us. */
typedef struct {
  uint64_t calls, errors;
  double seconds;                      /* wall clock of the whole storm */
  double p50_us, p90_us, p99_us, max_us;
  double mean_batch;                   /* reviews per launch, averaged over calls */
  double mean_queue_us, mean_device_us;
  uint64_t results;                    /* violations + autoreject results returned */
} gk_storm_out;
int gk_synth_query_storm(gk_engine* e, const gk_synth_batch* b, uint32_t threads, uint32_t per_thread, gk_storm_out* out);
/* the same through gk_query_ex2: Driver.Query the way the Go shim calls it -- the listed constraints, `query_flags` (GK_QUERY_PRE_MATCHED) */
int gk_synth_query_storm_ex(gk_engine* e, const gk_synth_batch* b, uint32_t threads, uint32_t per_thread, const uint32_t* constraint_ids, size_t n_constraints,
                            uint32_t query_flags, gk_storm_out* out);

#ifdef __cplusplus
}
#endif
#endif /* GKSYNTH_H */
