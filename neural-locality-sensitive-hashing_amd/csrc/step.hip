// One pipelined query batch per C-ABI call (nlsh_query_step_enqueue, ABI v3).
//
// A batch of the reference's `Indexer.query` (nlsh/indexer.py:56-96) is encode_hash + the PLAN, SCAN and MERGE phases of the scan:
// seven kernel launches.  Pipelined over HIP streams (front: encode + PLAN | mid: SCAN | tail: MERGE [+ the shard exchange]) it
// also needs eight to ten event records / waits, and through r04 the Python facade issued all of them one ctypes / torch call at a
// time: ~0.09 ms of host time per batch, which on small shards (8 GPUs: scan 0.055-0.068 ms) WAS the step.  Here the slot -- the
// validated encode launch, the scan call, the four streams and the slot's events -- is built once (nlsh_step_create) and a batch is
// ONE call that only swaps the batch pointer, its row stride and the Philox seed.  The library owns the events (it is linked against
// the HIP runtime the kernels use; ADVICE r04: the facade used to dlopen "libamdhip64.so" by name for them); the streams and every
// buffer stay the caller's.  Same kernels, same arguments, same results as the phase calls: only where the launches are issued from
// changes.
//
// r06 (ABI v4): the batch is FIVE launches -- the bucket lookup of the PLAN phase runs in encode_hash's epilogue (scan_plan.h) whenever
// the schedule is a bucket-major one, and the rest of the PLAN phase is two launches instead of three -- and the same batch exists
// as a one-stream call for callers that do not pipeline (nlsh_query_batch: what `Indexer.query_tensors` issues).
#include <cstdlib>
#include <new>

#include "step_nodes.h"

using namespace nlsh;

struct nlsh_step {
    EncPlan enc;
    nlsh_step_desc_t d;
    int dims[NLSH_MAX_LAYERS + 1];
    hipEvent_t ready, encoded, planned, scanned, done;
    bool done_pending;   // hold_done: the caller still has to release the batch (nlsh_step_release)
    // graph slots (nlsh_step_create_graph): the batch's five launches captured once, replayed on `lane`
    hipStream_t lane;
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipGraphNode_t enc_node, scan_node;
    bool graph_failed;   // the capture or the instantiation did not work on this runtime: the slot launches its batches eagerly on its stream
};

// One batch's scan call as the bucket-major descriptor (validated), or -- query-major schedule -- nothing (fused = false).
static int scan_call(const nlsh_step_desc_t &d, const float *queries, int64_t q_stride, void *ev0, void *ev1, nlsh_stream_t stream, int phases,
                     int plan_blocks, BucketScanCall *call_out) {
    return scan_topk_cells_phase_checked(d.corpus_sorted, d.row_stride, d.d, d.gid, d.uniq_keys, d.offsets, d.bucket_order, d.n_buckets, d.cell_of,
                                         d.cell_offsets, d.n_cells, d.inv_norm, queries, q_stride, d.Q, d.qkeys, d.nkeys, d.n_probes, d.k, d.metric,
                                         d.algo, d.seg_rows, d.out_dist, d.out_idx, d.out_keys, d.out_ncand, d.status, d.workspace, d.workspace_bytes,
                                         d.max_tasks, ev0, ev1, stream, phases, plan_blocks, call_out);
}

static bool fuses_lookup(const nlsh_step_desc_t &d) { return d.algo != NLSH_SCAN_QUERY_MAJOR && d.n_probes <= 64 && d.n_buckets > 0; }

// encode_hash with the bucket lookup in its epilogue when the schedule allows it; *plan_phase = what is left of the PLAN phase
static int launch_encode(EncPlan &enc, const nlsh_step_desc_t &d, const float *queries, int64_t q_stride, uint64_t seed, hipStream_t s, int *plan_phase,
                         int *plan_blocks) {
    *plan_phase = NLSH_PHASE_PLAN;
    *plan_blocks = 0;
    if (!fuses_lookup(d)) return encode_plan_launch(enc, queries, q_stride, seed, s);
    BucketScanCall c;
    int rc = scan_call(d, queries, q_stride, nullptr, nullptr, (nlsh_stream_t)s, NLSH_PHASE_PLAN_REST, 0, &c);
    if (rc != NLSH_OK) return rc;
    PlanArgs pa;
    rc = bucket_scan_plan_args(c, &pa);
    if (rc == NLSH_OK) rc = encode_plan_fuse_lookup(enc, pa);   // idempotent: the first call of a slot sizes the coarse table's LDS
    if (rc != NLSH_OK) return rc;
    *plan_phase = NLSH_PHASE_PLAN_REST;
    *plan_blocks = (int)enc.grid;
    return encode_plan_launch(enc, queries, q_stride, seed, s, &pa);
}

static int make_event(hipEvent_t *e) {
    NLSH_CHECK_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return NLSH_OK;
}

static int step_create(const nlsh_step_desc_t *desc, size_t desc_bytes, hipStream_t lane, nlsh_step_t **out) {
    NLSH_REQUIRE(desc && out, NLSH_E_INVALID, "step_create: null pointer");
    NLSH_REQUIRE(desc_bytes == sizeof(nlsh_step_desc_t), NLSH_E_INVALID, "step_create: descriptor of %zu bytes, this library's is %zu (ABI %d)",
                 desc_bytes, sizeof(nlsh_step_desc_t), NLSH_ABI_VERSION);
    if (lane == nullptr) {
        NLSH_REQUIRE(desc->front && desc->mid && desc->tail, NLSH_E_INVALID, "step_create: the front, mid and tail streams must be real streams (not the default stream)");
        NLSH_REQUIRE(desc->front != desc->mid && desc->mid != desc->tail && desc->front != desc->tail && desc->plan != desc->front && desc->plan != desc->mid &&
                         desc->plan != desc->tail,
                     NLSH_E_INVALID, "step_create: front, plan, mid and tail must be DIFFERENT streams (the stages overlap only across streams, and an event is recorded on one and waited for on the next)");
    }
    NLSH_REQUIRE(desc->n_layers >= 1 && desc->n_layers <= NLSH_MAX_LAYERS && desc->dims, NLSH_E_INVALID, "step_create: n_layers=%d", desc->n_layers);
    NLSH_REQUIRE(desc->n_probes >= 1 && desc->n_probes <= NLSH_MAX_PROBES, NLSH_E_UNSUPPORTED, "step_create: n_probes=%d not in [1,%d] (one scan call per batch)",
                 desc->n_probes, NLSH_MAX_PROBES);
    NLSH_REQUIRE(desc->Q >= 1 && desc->dims[0] == desc->d, NLSH_E_INVALID, "step_create: Q=%lld, encoder input %d vs corpus dimension %d", (long long)desc->Q,
                 desc->dims[0], desc->d);
    nlsh_step *s = new (std::nothrow) nlsh_step();
    NLSH_REQUIRE(s != nullptr, NLSH_E_INVALID, "step_create: out of host memory");
    s->d = *desc;
    for (int l = 0; l <= desc->n_layers; ++l) s->dims[l] = desc->dims[l];
    s->d.dims = s->dims;
    s->ready = s->encoded = s->planned = s->scanned = s->done = nullptr;
    s->done_pending = false;
    s->lane = lane; s->graph = nullptr; s->exec = nullptr; s->enc_node = s->scan_node = nullptr;
    s->graph_failed = getenv("NLSH_STEP_NO_GRAPH") != nullptr;    // diagnostic: a graph slot without its graph (what a failed capture leaves)
    const hipStream_t last = lane ? lane : (hipStream_t)desc->tail;    // the stream a batch ends on
    int rc = encode_plan_fill(s->enc, desc->Q, desc->n_layers, s->dims, desc->packed, desc->act, desc->key_mode, desc->n_probes, desc->n_multi_rows, 0,
                              nullptr, nullptr, nullptr, desc->qkeys, desc->nkeys);
    // the scan call's own argument checks (workspace alignment and size, strides, limits) run HERE, on a stand-in batch pointer, so
    // that a bad descriptor fails before anything is enqueued (they used to fail inside the first enqueue, behind its encode)
    if (rc == NLSH_OK) {
        BucketScanCall c;
        rc = scan_call(s->d, desc->corpus_sorted ? desc->corpus_sorted : (const float *)desc->workspace, desc->row_stride >= desc->d ? desc->row_stride : desc->d,
                       nullptr, nullptr, (nlsh_stream_t)last, NLSH_PHASE_SCAN, 0, desc->algo != NLSH_SCAN_QUERY_MAJOR ? &c : nullptr);
        if (rc == NLSH_OK && desc->algo != NLSH_SCAN_QUERY_MAJOR) {
            PlanArgs pa;
            rc = bucket_scan_plan_args(c, &pa);
            if (rc == NLSH_OK && fuses_lookup(s->d)) rc = encode_plan_fuse_lookup(s->enc, pa);
        }
    }
    if (rc == NLSH_OK && lane != nullptr && !(fuses_lookup(s->d) && desc->max_tasks > 0)) {
        set_error("step_create_graph: graph slots take the bucket-major schedules (algo 1, 2) on a non-empty index and task table");
        rc = NLSH_E_UNSUPPORTED;
    }
    if (rc == NLSH_OK) rc = encode_plan_prepare(s->enc);     // the dynamic-LDS permission of the encode's form: not inside a capture
    if (rc == NLSH_OK) rc = make_event(&s->ready);
    if (rc == NLSH_OK && !lane) rc = make_event(&s->encoded);
    if (rc == NLSH_OK && !lane) rc = make_event(&s->planned);
    if (rc == NLSH_OK && !lane) rc = make_event(&s->scanned);
    if (rc == NLSH_OK) rc = make_event(&s->done);
    if (rc == NLSH_OK && hipEventRecord(s->done, last) != hipSuccess) {   // the slot starts out free
        set_error("step_create: hipEventRecord failed");
        rc = NLSH_E_HIP;
    }
    if (rc != NLSH_OK) {
        nlsh_step_destroy(s);
        return rc;
    }
    *out = s;
    return NLSH_OK;
}

extern "C" int nlsh_step_create(const nlsh_step_desc_t *desc, size_t desc_bytes, nlsh_step_t **out) {
    return step_create(desc, desc_bytes, nullptr, out);
}

extern "C" int nlsh_step_create_graph(const nlsh_step_desc_t *desc, size_t desc_bytes, nlsh_stream_t lane, nlsh_step_t **out) {
    NLSH_REQUIRE(lane != nullptr, NLSH_E_INVALID, "step_create_graph: the slot's stream must be a real stream (not the default stream: it is captured)");
    return step_create(desc, desc_bytes, (hipStream_t)lane, out);
}

extern "C" int nlsh_step_destroy(nlsh_step_t *s) {
    if (!s) return NLSH_OK;
    for (hipEvent_t e : {s->ready, s->encoded, s->planned, s->scanned, s->done})
        if (e) (void)hipEventDestroy(e);
    if (s->exec) (void)hipGraphExecDestroy(s->exec);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    delete s;
    return NLSH_OK;
}

extern "C" int nlsh_step_set_weights(nlsh_step_t *s, const float *packed) {
    NLSH_REQUIRE(s && packed, NLSH_E_INVALID, "step_set_weights: null pointer");
    s->enc.a.packed = packed;
    s->d.packed = packed;
    return NLSH_OK;
}

static int scan_phase(const nlsh_step *s, const float *queries, int64_t q_stride, void *ev0, void *ev1, nlsh_stream_t stream, int phases, int plan_blocks) {
    return scan_call(s->d, queries, q_stride, ev0, ev1, stream, phases, plan_blocks, nullptr);
}

// A graph slot's batch: the five launches replayed from the slot's captured graph on the slot's own stream.  Batches of different slots
// overlap because their streams do (the slot's previous batch precedes this one on the same stream: no event between them).  Per batch the
// host makes ONE graph launch, two node updates (the encode's batch pointer / stride / seed / lookup arguments, the scan's query
// pointer) and the `done` record -- r05's staged slots made seven launches and eight to ten event calls (~3.5 us each inside the runtime).
// With scan events requested the batch is launched eagerly on the same stream (an event record cannot be re-pointed in a graph).
static int graph_capture(nlsh_step *s, const float *queries, int64_t q_stride, uint64_t seed) {
    NLSH_CHECK_HIP(hipStreamBeginCapture(s->lane, hipStreamCaptureModeThreadLocal));
    int plan_phase, plan_blocks;
    int rc = launch_encode(s->enc, s->d, queries, q_stride, seed, s->lane, &plan_phase, &plan_blocks);
    if (rc == NLSH_OK) rc = scan_call(s->d, queries, q_stride, nullptr, nullptr, (nlsh_stream_t)s->lane, plan_phase | NLSH_PHASE_SCAN | NLSH_PHASE_MERGE, plan_blocks, nullptr);
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(s->lane, &g);
    if (rc != NLSH_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    NLSH_CHECK_HIP(ce);
    s->graph = g;
    // the two nodes whose arguments change per batch, found by their kernels
    EncNode en;
    ScanNode sn;
    BucketScanCall c;
    rc = scan_call(s->d, queries, q_stride, nullptr, nullptr, (nlsh_stream_t)s->lane, NLSH_PHASE_SCAN, 0, &c);
    if (rc == NLSH_OK) rc = bucket_scan_node(c, &sn);
    if (rc == NLSH_OK) rc = encode_plan_node(s->enc, queries, q_stride, seed, nullptr, &en);
    if (rc != NLSH_OK) return rc;
    size_t n = 0;
    NLSH_CHECK_HIP(hipGraphGetNodes(g, nullptr, &n));
    hipGraphNode_t nodes[16];
    NLSH_REQUIRE(n >= 2 && n <= 16, NLSH_E_HIP, "step graph: %zu nodes captured", n);
    NLSH_CHECK_HIP(hipGraphGetNodes(g, nodes, &n));
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        NLSH_CHECK_HIP(hipGraphNodeGetType(nodes[i], &t));
        if (t != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams kp;
        NLSH_CHECK_HIP(hipGraphKernelNodeGetParams(nodes[i], &kp));
        if (kp.func == en.p.func) s->enc_node = nodes[i];
        else if (kp.func == sn.p.func) s->scan_node = nodes[i];
    }
    NLSH_REQUIRE(s->enc_node && s->scan_node, NLSH_E_HIP, "step graph: the encode / scan nodes were not found among the %zu captured", n);
    NLSH_CHECK_HIP(hipGraphInstantiate(&s->exec, g, nullptr, nullptr, 0));
    return NLSH_OK;
}

static int graph_enqueue(nlsh_step *s, const float *queries, int64_t q_stride, uint64_t seed, nlsh_stream_t producer, void *ev0, void *ev1) {
    const hipStream_t lane = s->lane;
    if ((hipStream_t)producer != lane) {    // the batch may still be in flight on the stream that produced it (as for the staged slots)
        const hipError_t qe = hipStreamQuery((hipStream_t)producer);
        if (qe == hipErrorNotReady) {
            (void)hipGetLastError();
            NLSH_CHECK_HIP(hipEventRecord(s->ready, (hipStream_t)producer));
            NLSH_CHECK_HIP(hipStreamWaitEvent(lane, s->ready, 0));
        } else {
            NLSH_CHECK_HIP(qe);
        }
    }
    int rc;
    if (!s->exec && !s->graph_failed && !(ev0 || ev1)) {
        // Captured once.  A runtime that cannot capture or instantiate the batch (nothing has been launched by a failed capture) leaves the
        // slot what a graph slot is without its graph: the same five launches, eagerly, on the slot's own stream -- same results, same
        // overlap between slots, ~0.005 ms more host time per batch (profiles/r06_shard_step_profile_eager_on_lane_vs_staged.jsonl).
        if (graph_capture(s, queries, q_stride, seed) != NLSH_OK) {
            s->graph_failed = true;
            if (s->exec) { (void)hipGraphExecDestroy(s->exec); s->exec = nullptr; }
            if (s->graph) { (void)hipGraphDestroy(s->graph); s->graph = nullptr; }
            (void)hipGetLastError();
        }
    }
    if (ev0 || ev1 || !s->exec) {
        int plan_phase, plan_blocks;
        rc = launch_encode(s->enc, s->d, queries, q_stride, seed, lane, &plan_phase, &plan_blocks);
        if (rc == NLSH_OK) rc = scan_call(s->d, queries, q_stride, ev0, ev1, (nlsh_stream_t)lane, plan_phase | NLSH_PHASE_SCAN | NLSH_PHASE_MERGE, plan_blocks, nullptr);
        if (rc != NLSH_OK) return rc;
    } else {
        // this batch's arguments of the two nodes that carry them
        BucketScanCall c;
        rc = scan_call(s->d, queries, q_stride, nullptr, nullptr, (nlsh_stream_t)lane, NLSH_PHASE_SCAN, 0, &c);
        if (rc != NLSH_OK) return rc;
        PlanArgs pa;
        EncNode en;
        ScanNode sn;
        rc = bucket_scan_plan_args(c, &pa);
        if (rc == NLSH_OK) rc = encode_plan_fuse_lookup(s->enc, pa);
        if (rc == NLSH_OK) rc = encode_plan_node(s->enc, queries, q_stride, seed, &pa, &en);
        if (rc == NLSH_OK) rc = bucket_scan_node(c, &sn);
        if (rc != NLSH_OK) return rc;
        NLSH_CHECK_HIP(hipGraphExecKernelNodeSetParams(s->exec, s->enc_node, &en.p));
        NLSH_CHECK_HIP(hipGraphExecKernelNodeSetParams(s->exec, s->scan_node, &sn.p));
        NLSH_CHECK_HIP(hipGraphLaunch(s->exec, lane));
    }
    if (s->d.hold_done) s->done_pending = true;     // the caller queues more work on the slot's stream (the shard exchange) first
    else NLSH_CHECK_HIP(hipEventRecord(s->done, lane));
    return NLSH_OK;
}

extern "C" int nlsh_query_step_enqueue(nlsh_step_t *s, const float *queries, int64_t q_stride, uint64_t seed, nlsh_stream_t producer,
                                       void *ev_scan_begin, void *ev_scan_end) {
    NLSH_REQUIRE(s && queries, NLSH_E_INVALID, "query_step_enqueue: null pointer");
    NLSH_REQUIRE(!s->done_pending, NLSH_E_INVALID, "query_step_enqueue: the slot's previous batch was not released (nlsh_step_release)");
    if (s->lane) return graph_enqueue(s, queries, q_stride, seed, producer, ev_scan_begin, ev_scan_end);
    const hipStream_t front = (hipStream_t)s->d.front, mid = (hipStream_t)s->d.mid, tail = (hipStream_t)s->d.tail;
    NLSH_CHECK_HIP(hipStreamWaitEvent(front, s->done, 0));            // the slot's previous batch has left the tail: its buffers are free
    // The batch may still be in flight on the stream that produced it.  NULL is a stream like any other here -- the default stream, where
    // torch produces a tensor unless told otherwise -- and the stage streams are non-blocking ones that do NOT order themselves behind it.
    // A producer stream with nothing in flight (one query, ~1 us: the steady state of a loop over batches made earlier) needs no event.
    bool wait_producer = (hipStream_t)producer != front;
    if (wait_producer) {
        const hipError_t qe = hipStreamQuery((hipStream_t)producer);
        if (qe == hipSuccess) wait_producer = false;
        else if (qe == hipErrorNotReady) (void)hipGetLastError();     // an answer, not a failure
        else NLSH_CHECK_HIP(qe);
    }
    if (wait_producer) {
        NLSH_CHECK_HIP(hipEventRecord(s->ready, (hipStream_t)producer));
        NLSH_CHECK_HIP(hipStreamWaitEvent(front, s->ready, 0));
    }
    int plan_phase, plan_blocks;
    int rc = launch_encode(s->enc, s->d, queries, q_stride, seed, front, &plan_phase, &plan_blocks);
    if (rc != NLSH_OK) return rc;
    hipStream_t hp = front;
    if (s->d.plan) {                                                  // four stages: the PLAN phase on a stream of its own behind the encode
        hp = (hipStream_t)s->d.plan;
        NLSH_CHECK_HIP(hipEventRecord(s->encoded, front));
        NLSH_CHECK_HIP(hipStreamWaitEvent(hp, s->encoded, 0));
    }
    rc = scan_phase(s, queries, q_stride, nullptr, nullptr, hp, plan_phase, plan_blocks);
    if (rc != NLSH_OK) return rc;
    NLSH_CHECK_HIP(hipEventRecord(s->planned, hp));
    NLSH_CHECK_HIP(hipStreamWaitEvent(mid, s->planned, 0));
    rc = scan_phase(s, queries, q_stride, ev_scan_begin, ev_scan_end, mid, NLSH_PHASE_SCAN, 0);
    if (rc != NLSH_OK) return rc;
    NLSH_CHECK_HIP(hipEventRecord(s->scanned, mid));
    NLSH_CHECK_HIP(hipStreamWaitEvent(tail, s->scanned, 0));
    rc = scan_phase(s, queries, q_stride, nullptr, nullptr, tail, NLSH_PHASE_MERGE, 0);
    if (rc != NLSH_OK) return rc;
    if (s->d.hold_done) s->done_pending = true;                       // the caller queues more work on the tail stream (the shard exchange) first
    else NLSH_CHECK_HIP(hipEventRecord(s->done, tail));
    return NLSH_OK;
}

extern "C" int nlsh_step_release(nlsh_step_t *s) {
    NLSH_REQUIRE(s, NLSH_E_INVALID, "step_release: null pointer");
    NLSH_CHECK_HIP(hipEventRecord(s->done, s->lane ? s->lane : (hipStream_t)s->d.tail));
    s->done_pending = false;
    return NLSH_OK;
}

extern "C" int nlsh_step_busy(nlsh_step_t *s) {
    NLSH_REQUIRE(s, NLSH_E_INVALID, "step_busy: null pointer");
    if (s->done_pending) return 1;
    const hipError_t e = hipEventQuery(s->done);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();   // "not ready" is an answer, not a failure: do not leave it behind as the thread's last HIP error
        return 1;
    }
    set_error("hipEventQuery failed: %s", hipGetErrorString(e));
    return NLSH_E_HIP;
}

// The same batch on ONE stream, nothing kept between calls (the streams and hold_done of `desc` are ignored): encode_hash with the
// bucket lookup in its epilogue, then bscan, bscatter, the scan kernel and bmerge -- five launches.  `lookup_done` != 0 repeats the
// scan part only, on keys (and their lookup) an earlier call of the same batch left in `qkeys` / the workspace: the retry after a
// task-table overflow (status[1] = 1; the lookup's records sit at workspace offsets that do not depend on max_tasks), with the
// lookup redone by the stand-alone kernel.
extern "C" int nlsh_query_batch(const nlsh_step_desc_t *desc, size_t desc_bytes, const float *queries, int64_t q_stride, uint64_t seed, int64_t row0,
                                int lookup_done, void *ev_scan_begin, void *ev_scan_end, nlsh_stream_t stream) {
    NLSH_REQUIRE(desc, NLSH_E_INVALID, "query_batch: null pointer");
    NLSH_REQUIRE(desc_bytes == sizeof(nlsh_step_desc_t), NLSH_E_INVALID, "query_batch: descriptor of %zu bytes, this library's is %zu (ABI %d)", desc_bytes,
                 sizeof(nlsh_step_desc_t), NLSH_ABI_VERSION);
    NLSH_REQUIRE(desc->Q >= 0, NLSH_E_INVALID, "query_batch: Q=%lld", (long long)desc->Q);
    if (desc->Q == 0) return NLSH_OK;
    NLSH_REQUIRE(queries, NLSH_E_INVALID, "query_batch: null pointer");
    NLSH_REQUIRE(desc->n_layers >= 1 && desc->n_layers <= NLSH_MAX_LAYERS && desc->dims && desc->dims[0] == desc->d, NLSH_E_INVALID,
                 "query_batch: n_layers=%d, encoder input vs corpus dimension %d", desc->n_layers, desc->d);
    NLSH_REQUIRE(desc->n_probes >= 1 && desc->n_probes <= NLSH_MAX_PROBES, NLSH_E_UNSUPPORTED, "query_batch: n_probes=%d not in [1,%d] (one scan call per batch)",
                 desc->n_probes, NLSH_MAX_PROBES);
    hipStream_t s = (hipStream_t)stream;
    int plan_phase = NLSH_PHASE_PLAN, plan_blocks = 0;
    if (!lookup_done) {
        EncPlan enc;
        int rc = encode_plan_fill(enc, desc->Q, desc->n_layers, desc->dims, desc->packed, desc->act, desc->key_mode, desc->n_probes, desc->n_multi_rows, row0, nullptr,
                                  nullptr, nullptr, desc->qkeys, desc->nkeys);
        if (rc == NLSH_OK) rc = launch_encode(enc, *desc, queries, q_stride, seed, s, &plan_phase, &plan_blocks);
        if (rc != NLSH_OK) return rc;
    }
    return scan_call(*desc, queries, q_stride, ev_scan_begin, ev_scan_end, stream, plan_phase | NLSH_PHASE_SCAN | NLSH_PHASE_MERGE, plan_blocks, nullptr);
}
