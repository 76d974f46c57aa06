/*
 * nlsh_hip.h -- C ABI of the MI355X (gfx950) implementation of nlsh's query-time hot path.
 *
 * The reference exposes this path as duck-typed Python; the only native
 * boundary it has is the Cython module nlsh/utils.pyx.  The entry points below are what a
 * Python/ctypes (or cgo/JNI) binding of that path would bind; each cites the reference
 * interface it replaces.  INTEGRATION.md shows the reference-side ctypes stub.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer marked [dev] is a DEVICE pointer
 *     (hipMalloc / torch tensor.data_ptr()), [host] is host memory;
 *   - the caller allocates every buffer (workspace sizes come from the *_workspace queries; a workspace starts on a 16-byte
 *     boundary -- any hipMalloc'ed or torch tensor does -- and is refused otherwise);
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream); no call synchronises the device or allocates device memory;
 *   - return 0 on success, a negative NLSH_E_* code on failure; never throws; the message of
 *     the last failure on the calling thread is available from nlsh_last_error();
 *   - integer outputs are bit-exact w.r.t. the oracle; fp32 outputs within the tolerance
 *     stated in DESIGN.md.
 */
#ifndef NLSH_HIP_H
#define NLSH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLSH_ABI_VERSION 4   /* 4 (r06): nlsh_query_batch; the workspace counters are handed back by the PLAN phase; 3 (r05): pipelined batch slots (nlsh_step_*, nlsh_query_step_enqueue); 2 (r04): cells */

typedef void *nlsh_stream_t; /* hipStream_t */

enum {
    NLSH_OK = 0,
    NLSH_E_INVALID = -1,     /* bad argument (null pointer, negative size, k or P out of range) */
    NLSH_E_UNSUPPORTED = -2, /* shape outside what the kernels are built for (see each call) */
    NLSH_E_HIP = -3,         /* a HIP runtime call failed (message has hipGetErrorString) */
    NLSH_E_WORKSPACE = -4    /* workspace too small */
};

enum { NLSH_ACT_SIGMOID = 0, NLSH_ACT_TANH = 1 };      /* nlsh/hashings.py:22-26 (tanh_output) */
enum { NLSH_KEY_REF_INT16 = 0, NLSH_KEY_FULL = 1 };    /* nlsh/utils.pyx:7-15 (int16 wrap) | eval.py:49-53 */
/* nlsh/data.py:191-201 (F.pairwise_distance: sqrt(sum(((q - c) + 1e-6)^2)), the operation order the oracle restates bit for bit) |
 * nlsh/data.py:99-109 | OPT-IN: the same L2 distance evaluated as sqrt(sum(((q + 1e-6) - c)^2)) -- one rounding per element
 * differs from the reference's order (relative 1e-7; |d - d_exact| <= 1e-4 * max(1, d), the tolerance BASELINE.json states), in
 * exchange for 2 instead of 3 vector operations per element in the LDS-tiled schedule; schedules 0 and 1 answer it with the
 * exact form.  Never the default: NLSH_METRIC_L2_EPS is bit-identical to the oracle. */
enum { NLSH_METRIC_L2_EPS = 0, NLSH_METRIC_COSINE = 1, NLSH_METRIC_L2_EPS_FOLDED = 2 };
/* schedules of nlsh_scan_topk: one wave per (query, segment) | one wave per (bucket segment, <= 8 queries) |
 * one workgroup per (bucket segment, <= 16 queries) with the row tile staged through LDS.  0 and 1 give
 * bit-identical results; 2 sums each distance in k order (bit-identical to the oracle), ids agree except fp32 near-ties. */
enum { NLSH_SCAN_QUERY_MAJOR = 0, NLSH_SCAN_BUCKET_MAJOR = 1, NLSH_SCAN_BUCKET_TILED = 2 };

#define NLSH_MAX_LAYERS 8   /* Linear layers incl. the output layer */
#define NLSH_MAX_HASH_BITS 32
#define NLSH_MAX_PROBES 64  /* keys per query one nlsh_scan_topk call takes */
#define NLSH_MAX_ENCODE_PROBES 128  /* hash_times nlsh_encode_hash generates (eval.py:148 sweeps 1..100); scan in slices of 64 */
#define NLSH_MAX_K 64
#define NLSH_MAX_DIM 1024   /* vector dimension of corpus / queries for the scan */
#define NLSH_MAX_WIDTH 632  /* widest HIDDEN encoder layer the LDS-resident MLP supports (the input may be NLSH_MAX_DIM wide) */

int nlsh_abi_version(void);
const char *nlsh_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Hash function forward + bit packing + multi-probe key generation.
 * Replaces: encoder forward (encoders.py:18-21, 39-55) -> output Linear + sigmoid/tanh
 * (nlsh/hashings.py:13-27) -> hard bits / Bernoulli samples (hashings.py:66-81) -> D2H ->
 * binarr_to_int + set() (nlsh/utils.pyx:6-32), i.e. everything `MultivariateBernoulli.hash`
 * does, without leaving the device.
 * ------------------------------------------------------------------------------------------- */

/* Number of floats of the packed (MFMA-fragment-ordered) weight blob for a layer stack: every layer in the fragment order of the
 * 32-row-tile forms, then the hidden layers once more in 16x16x4 order (the 16-row form batches of <= 4096 rows take).
 * dims [host] = {d_in, h_1, ..., H}, n_layers = number of Linear layers (len(dims) - 1). */
int64_t nlsh_encoder_packed_floats(int n_layers, const int *dims);

/* Pack nn.Linear-layout weights (W[l] is [dims[l+1], dims[l]] row-major, b[l] is [dims[l+1]] or
 * NULL) into `packed` [dev].  W, b: [host] arrays of n_layers [dev] pointers.
 * BatchNorm1d in eval mode is an affine map: fold it into W/b before packing. */
int nlsh_encoder_pack(int n_layers, const int *dims, const float *const *W, const float *const *b,
                      float *packed, nlsh_stream_t stream);

/* x [dev] fp32 [n, d] with row stride x_stride (floats).  Outputs (all [dev], nullable unless noted):
 *   z_out     [n, H] pre-activation of the output layer
 *   probs_out [n, H] module output: sigmoid(z) or tanh(z)             (hashings.py:39-40 predict)
 *   code_out  [n]    hard code, bit h of the hasher = bit (H-1-h): MSB-first (utils.pyx:12-14)
 *   keys_out  [n, n_probes] (required) distinct keys in first-occurrence order, slot 0 = hard key;
 *             key_mode REF_INT16: sign-extended low 16 bits; FULL: the H-bit code as int32 bits
 *   nkeys_out [n]    (required) number of valid slots in keys_out (>= 1)
 * Probes 1..n_probes-1 are Bernoulli(p) draws from a Philox4x32-10 stream keyed by `seed`,
 * counter (row0 + row, probe, word): reproducible across devices and ranks.  Rows with index
 * >= n_multi_rows are single-probe (Indexer.hash's trailing-batch rule, nlsh/indexer.py:51-53).
 * Limits: H <= 32, n_probes <= NLSH_MAX_ENCODE_PROBES, dims[0] <= NLSH_MAX_DIM, hidden dims[l] <= NLSH_MAX_WIDTH. */
int nlsh_encode_hash(const float *x, int64_t n, int64_t x_stride, int n_layers, const int *dims,
                     const float *packed, int act, int key_mode, int n_probes, int64_t n_multi_rows,
                     uint64_t seed, int64_t row0, float *z_out, float *probs_out, uint32_t *code_out,
                     int32_t *keys_out, int32_t *nkeys_out, nlsh_stream_t stream);

/* Standalone bit packing: codes [dev] int32 [B, n, H] (0/1, C-contiguous) -> keys_out [dev] int32
 * [B, n].  Replaces binarr_to_int + the inner loop of hash_codes (nlsh/utils.pyx:6-15, 22-31);
 * the set() of each row is formed by the host binding.  key_mode as above (FULL: eval.py:49-53). */
int nlsh_pack_codes(const int32_t *codes, int64_t B, int n, int H, int key_mode, int32_t *keys_out,
                    nlsh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Index build.  Replaces build_index (nlsh/indexer.py:6-24): key -> ascending row list, as CSR.
 * ------------------------------------------------------------------------------------------- */
size_t nlsh_build_csr_workspace(int64_t n);

/* keys [dev] int32 [n] (one key per row).  Outputs [dev]: perm [n] row ids grouped by bucket
 * (buckets in ascending signed key order, rows ascending inside a bucket), uniq_keys [n] (first
 * *n_buckets valid), offsets [n + 1] (first *n_buckets + 1 valid), n_buckets [1]. */
int nlsh_build_csr(const int32_t *keys, int64_t n, int32_t *perm, int32_t *uniq_keys, int32_t *offsets,
                   int32_t *n_buckets, void *workspace, size_t workspace_bytes, nlsh_stream_t stream);

/* Schedule order of the buckets for the bucket-major scans: order_out [dev] int32 [n_buckets] = bucket
 * indices by DESCENDING size (ties: ascending index), from offsets [dev] [n_buckets + 1].  Static per index:
 * nlsh_scan_topk numbers its (bucket segment, query group) tasks in this order, so the heavy tasks of the big
 * buckets are dispatched first and the launch does not end on a tail of late-started long tasks.  No reference
 * counterpart (the reference walks `for key in index_keys`, nlsh/indexer.py:66); changes speed, never results. */
size_t nlsh_bucket_order_workspace(int64_t n_buckets);
int nlsh_bucket_order(const int32_t *offsets, int64_t n_buckets, int32_t *order_out, void *workspace,
                      size_t workspace_bytes, nlsh_stream_t stream);

/* Cells: small-bucket packing for the LDS-tiled scan (NLSH_SCAN_BUCKET_TILED).  A task of that schedule is one workgroup on
 * (<= 256 consecutive rows of corpus_sorted) x (<= 16 queries) and costs ~9 us before its first distance; with one task per
 * (bucket, query group) a balanced hash with tiny buckets (GloVe-1.2M-shaped: 104 k buckets, ~1 probing query per touched bucket and
 * batch) pays that latency once per (query, bucket) pair.  The sorted corpus is bucket-contiguous, so consecutive small buckets are
 * consecutive rows: a CELL is either one bucket of more than window_rows rows, or a greedy run (CSR order, restarted every 1024
 * buckets so that the packing is a pure function of (offsets, window_rows)) of consecutive buckets of <= window_rows rows whose rows
 * total <= window_rows.  The scan then counts (query, probe) pairs and lays out tasks per CELL: the queries probing any bucket of a
 * window share its tasks, and each (task, query) carries the row range of ITS bucket inside the window, applied when the top-k is
 * selected -- every distance is still the same k-ascending fmaf chain over the same row, so results are bit-identical with and without
 * cells, for every window_rows.  What changes is the number of tasks (fewer, fuller) against rows scored for queries that do not own
 * them (window_rows 64 adds none: a lone tiny bucket already costs a 64-row tile).
 * Outputs [dev]: cell_of [n_buckets] bucket -> cell; cell_offsets [n_buckets + 1] first sorted row of each cell (entries from
 * *n_cells on hold N); cell_order [n_buckets] cells by descending rows (first *n_cells entries valid) = the bucket_order argument
 * of the scan when cells are passed; n_cells [1].  window_rows in [1, 256].  No reference counterpart (the reference walks
 * `for key in index_keys`, nlsh/indexer.py:66, one gather per key); changes speed, never results. */
size_t nlsh_build_cells_workspace(int64_t n_buckets);
int nlsh_build_cells(const int32_t *offsets, int64_t n_buckets, int window_rows, int32_t *cell_of, int32_t *cell_offsets,
                     int32_t *cell_order, int32_t *n_cells, void *workspace, size_t workspace_bytes, nlsh_stream_t stream);

/* Re-order the corpus bucket-contiguously: sorted[i, :] = corpus[perm[i], :], zero padded to
 * dst_stride floats (dst_stride % 4 == 0, >= d).  Replaces the per-(query,key) index_select
 * gather of nlsh/indexer.py:77-82 by a one-time permutation.  inv_norm (nullable) [n] receives
 * 1 / max(||row||, 1e-8) (cosine).  gid (nullable) [n] receives perm[i] + id_base: the global
 * row id the scan reports (corpus shards pass their row offset as id_base). */
int nlsh_gather_rows(const float *corpus, int64_t src_stride, int d, const int32_t *perm, int64_t n,
                     float *sorted, int64_t dst_stride, float *inv_norm, int32_t *gid, int32_t id_base,
                     nlsh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Candidate scan + top-k.  Replaces the per-query loop of Indexer.query (nlsh/indexer.py:62-95):
 * bucket lookup (:68), gather (:77-82), distance (:84-87, nlsh/data.py:99-109,191-201),
 * cat (:88), topk + id map (:90-91), n_candidates (:71,94).
 * ------------------------------------------------------------------------------------------- */
size_t nlsh_scan_workspace(int64_t Q, int P, int k, int64_t max_tasks, int64_t n_buckets, int d);

/* corpus_sorted [dev] fp32 [N, row_stride] bucket-contiguous (nlsh_gather_rows), gid [dev] [N],
 * uniq_keys [dev] [n_buckets] ascending, offsets [dev] [n_buckets+1], bucket_order (nullable) [dev]
 * [n_buckets] from nlsh_bucket_order (NULL = CSR order), inv_norm [dev] [N] (cosine only), queries [dev] [Q, d] stride q_stride, qkeys [dev] [Q, P] with nkeys [dev] [Q] valid slots
 * (a query's keys are a set, nlsh/utils.pyx:27-31: a key repeated within a row probes its bucket once; unknown keys are
 * empty buckets, never an error: indexer.py:61,68).
 * Outputs [dev]: out_dist [Q, k] ascending, +inf padded; out_idx [Q, k] global row ids, -1 padded;
 * out_keys (nullable) [Q, k] the 64-bit sort keys (monotone(dist) << 32 | id; ~0 padded) used by
 * nlsh_merge_topk; out_ncand [Q] candidates per query; status [2] = {tasks needed, flag}: flag 0 = complete,
 * 1 = task table overflow (repeat with max_tasks >= status[0]), 2 = workspace contract violated (below), 3 = the cells handed
 * to nlsh_scan_topk_cells_phase hold a shared window of more than 256 rows (not from nlsh_build_cells: results incomplete).
 * Order is (distance asc, row id asc): deterministic refinement of torch.topk's tie order.
 * A query's candidate list is cut into segments of `seg_rows` rows (0 = default), one wavefront
 * each; max_tasks bounds the number of segments the workspace holds: if status[1] != 0 the
 * results are incomplete and the call must be repeated with max_tasks >= status[0].
 * Workspace contract of the bucket-major schedules (algo 1, 2): the per-bucket pair counters live in the first
 * 4 * n_buckets bytes of the workspace (an offset that does not depend on Q, P or max_tasks).  Those bytes must be ZERO
 * before the first call that uses the buffer and must not be written by anything else afterwards (do not lend the buffer
 * to algo 0 calls in between); every PLAN phase leaves them zero again (the step that turns the counts into list offsets reads
 * and resets each counter), which is what saves a clearing launch per batch.  The contract is CHECKED on the device: the PLAN
 * phase holds the sum of the counters against the (query, probe) pairs it counted itself, rejects negative counters and rejects
 * any pair that drew a negative slot from its counter -- stale counts that cancel in the sum include a negative one, so every
 * way a stale count can enter a batch is covered --; on a violation the batch gets no task at all, every query's result is
 * empty and status[1] = 2 (the Python facade raises NLSH_E_WORKSPACE).  The counters are zero again after the refused call.
 * ev_scan_begin / ev_scan_end (nullable hipEvent_t): recorded on `stream` immediately before and
 * after the scan kernel, so a caller can time the HBM-bound kernel alone (bench.py roofline).
 * Limits: d <= NLSH_MAX_DIM, k <= NLSH_MAX_K, P <= NLSH_MAX_PROBES. */
int nlsh_scan_topk(const float *corpus_sorted, int64_t row_stride, int d, const int32_t *gid,
                   const int32_t *uniq_keys, const int32_t *offsets, const int32_t *bucket_order, int32_t n_buckets,
                   const float *inv_norm, const float *queries, int64_t q_stride, int64_t Q,
                   const int32_t *qkeys, const int32_t *nkeys, int P, int k, int metric, int algo, int seg_rows,
                   float *out_dist, int32_t *out_idx, uint64_t *out_keys, int32_t *out_ncand,
                   int32_t *status, void *workspace, size_t workspace_bytes, int64_t max_tasks,
                   void *ev_scan_begin, void *ev_scan_end, nlsh_stream_t stream);

/* Diagnostic, for tests of the bucket-major schedules (algo 1, 2): byte offsets, inside a workspace of this shape, of the
 * task table the PLAN phase leaves there -- int32 [status[0]][4] = {first pair of the query group, queries in the group,
 * first corpus row of the segment, rows in the segment} -- and (algo 2 only) of the tasks' per-slot records, ONE interleaved table
 * int32 [max_tasks][16][2] = {query id, row range lo | hi << 16}: *task_queries_offset is the table's offset (the id of (task, slot)
 * sits at + (task * 16 + slot) * 8), *task_ranges_offset = *task_queries_offset + 4 (the same stride of 8 bytes); the range is the
 * rows of the query's own bucket inside the task's rows: the whole segment without cells, the bucket's slice of a shared window
 * with them.  Any out pointer may be NULL.
 * No reference counterpart: the reference has no schedule (it walks `for key in index_keys`, nlsh/indexer.py:66). */
int nlsh_scan_workspace_layout(int64_t Q, int P, int k, int64_t max_tasks, int64_t n_buckets, int d, int algo,
                               size_t *task_table_offset, size_t *task_queries_offset, size_t *task_ranges_offset);

/* The same call cut in three, for callers that pipeline batches over streams: NLSH_PHASE_PLAN runs everything up
 * to the scan kernel (bucket lookup, task table; touches status and the workspace, the query-major schedule also out_ncand),
 * NLSH_PHASE_SCAN the scan kernel (reads what PLAN left in the workspace, leaves per-task partial top-k lists there),
 * NLSH_PHASE_MERGE the per-query merge of those lists (writes out_dist / out_idx / out_keys / out_ncand).  All take the identical
 * argument list and may be combined; each must be ordered after the previous one (an event when they run on
 * different streams), and a workspace belongs to one batch until its MERGE has finished.  All three bits =
 * nlsh_scan_topk.  The plan of batch i+1 and the merge of batch i-1 (latency-bound, small kernels) then overlap the
 * scan of batch i (nlsh_amd/pipeline.py). */
#define NLSH_PHASE_PLAN 1
#define NLSH_PHASE_SCAN 2
#define NLSH_PHASE_MERGE 4
int nlsh_scan_topk_phase(const float *corpus_sorted, int64_t row_stride, int d, const int32_t *gid,
                         const int32_t *uniq_keys, const int32_t *offsets, const int32_t *bucket_order, int32_t n_buckets,
                         const float *inv_norm, const float *queries, int64_t q_stride, int64_t Q,
                         const int32_t *qkeys, const int32_t *nkeys, int P, int k, int metric, int algo, int seg_rows,
                         float *out_dist, int32_t *out_idx, uint64_t *out_keys, int32_t *out_ncand,
                         int32_t *status, void *workspace, size_t workspace_bytes, int64_t max_tasks,
                         void *ev_scan_begin, void *ev_scan_end, nlsh_stream_t stream, int phases);

/* nlsh_scan_topk_phase on an index with cells (nlsh_build_cells): cell_of [dev] [n_buckets], cell_offsets [dev] [n_cells + 1],
 * and `bucket_order` = the cell order [n_cells] (or NULL).  n_cells = 0 (pointers NULL) is nlsh_scan_topk_phase.  Cells only
 * change the task layout of algo NLSH_SCAN_BUCKET_TILED; the other schedules ignore them.  Same outputs, bit for bit. */
int nlsh_scan_topk_cells_phase(const float *corpus_sorted, int64_t row_stride, int d, const int32_t *gid,
                               const int32_t *uniq_keys, const int32_t *offsets, const int32_t *bucket_order, int32_t n_buckets,
                               const int32_t *cell_of, const int32_t *cell_offsets, int32_t n_cells,
                               const float *inv_norm, const float *queries, int64_t q_stride, int64_t Q,
                               const int32_t *qkeys, const int32_t *nkeys, int P, int k, int metric, int algo, int seg_rows,
                               float *out_dist, int32_t *out_idx, uint64_t *out_keys, int32_t *out_ncand,
                               int32_t *status, void *workspace, size_t workspace_bytes, int64_t max_tasks,
                               void *ev_scan_begin, void *ev_scan_end, nlsh_stream_t stream, int phases);

/* ---------------------------------------------------------------------------------------------
 * One pipelined query batch per call.  Replaces, for a caller that streams batches, the whole body of Indexer.query
 * (nlsh/indexer.py:56-96: hashing.hash on the batch :59 via Indexer.hash :40-54, then the per-query loop :62-95) by ONE
 * enqueue: nlsh_encode_hash + the PLAN, SCAN and MERGE phases of nlsh_scan_topk_cells_phase (five launches since r06: the bucket
 * lookup of the PLAN phase runs in the encode's epilogue) and the events that order them across three or four streams.  A step (slot) is built once from everything that does not change between
 * batches of one shape; nlsh_query_step_enqueue then takes the batch pointer, its row stride and the Philox seed.  The library
 * owns the slot's events; streams and buffers are the caller's.  Results are those of the separate calls, bit for bit.
 *
 *   front:  encode(i+1) plan(i+1)         | ...          (plan on its own stream when `plan` is not NULL: four stages)
 *   mid  :  scan(i)                       | ...
 *   tail :  merge(i-1) [caller's work]    | ...
 *
 * A slot is reused when its previous batch has left the tail stream (event, waited for on the device by the next enqueue).
 * hold_done != 0: the caller queues more work on the tail stream after the merge (a sharded index's all-gather + shard merge)
 * and then calls nlsh_step_release, which marks the slot's batch as finished at that point of the tail stream.
 * A failed enqueue leaves the slot's workspace in an undefined state (workspace contract above): destroy the step.
 * Not thread-safe: one host thread per step.
 * ------------------------------------------------------------------------------------------- */
typedef struct nlsh_step nlsh_step_t;
typedef struct nlsh_step_desc {
    /* hash function: the arguments of nlsh_encode_hash that do not change from batch to batch (dims [host] is copied) */
    int32_t n_layers, act, key_mode, n_probes;
    const int *dims;
    const float *packed;
    int64_t n_multi_rows;
    /* index: the arguments of nlsh_scan_topk_cells_phase in its order */
    const float *corpus_sorted;
    int64_t row_stride;
    const int32_t *gid, *uniq_keys, *offsets, *bucket_order, *cell_of, *cell_offsets;
    const float *inv_norm;
    int32_t d, n_buckets, n_cells, k, metric, algo, seg_rows, hold_done;
    /* batch shape and the slot's own buffers (qkeys [Q, n_probes] and nkeys [Q] are written by the encode, read by the scan) */
    int64_t Q;
    int32_t *qkeys, *nkeys;
    float *out_dist;
    int32_t *out_idx;
    uint64_t *out_keys;
    int32_t *out_ncand, *status;
    void *workspace;
    size_t workspace_bytes;
    int64_t max_tasks;
    /* streams (hipStream_t): front, mid, tail must be three different non-default streams; plan may be NULL, else a fourth different
     * one.  Equal handles are refused by nlsh_step_create, as is every argument the scan call itself would refuse. */
    nlsh_stream_t front, plan, mid, tail;
} nlsh_step_desc_t;

/* desc_bytes = sizeof(nlsh_step_desc_t) of the CALLER's header: a binding built against another layout is refused. */
int nlsh_step_create(const nlsh_step_desc_t *desc, size_t desc_bytes, nlsh_step_t **step_out);
/* A GRAPH slot (r06): the batch's five launches are captured once into a hipGraph and replayed on `lane`, the slot's OWN stream (a real
 * stream, different for every slot; the four stream handles of `desc` are ignored).  nlsh_query_step_enqueue is then one graph launch,
 * two kernel-node updates (the batch pointer / row stride / seed of the encode, the query pointer of the scan) and one event record --
 * a third of the runtime calls of a staged slot.  Batches of different slots overlap because their lanes do; consecutive batches of one
 * slot are ordered by its lane.  hold_done: the caller's extra work goes on `lane`.  A call with scan events is launched eagerly on the
 * lane (same kernels), and so is every batch of a slot whose capture or instantiation the runtime refused (or NLSH_STEP_NO_GRAPH set in the
 * environment when the slot is created: diagnostic).  Bucket-major schedules only (algo 1, 2).  Same results as the staged slots and the
 * separate calls, bit for bit. */
int nlsh_step_create_graph(const nlsh_step_desc_t *desc, size_t desc_bytes, nlsh_stream_t lane, nlsh_step_t **step_out);
int nlsh_step_destroy(nlsh_step_t *step);
/* New packed weights (nlsh_encoder_pack) for the batches enqueued from now on (a training step between two batches). */
int nlsh_step_set_weights(nlsh_step_t *step, const float *packed);
/* queries [dev] fp32 [Q, d] with row stride q_stride.  producer: the stream the batch was produced on (NULL = the default stream, as
 * for every nlsh_stream_t of this header) -- the front stream waits for what is queued there now (one stream query; an event only
 * when the producer has work in flight); pass the step's own front stream when there is never anything to wait for.  ev_scan_begin / ev_scan_end (nullable hipEvent_t): recorded around the scan kernel. */
int nlsh_query_step_enqueue(nlsh_step_t *step, const float *queries, int64_t q_stride, uint64_t seed, nlsh_stream_t producer,
                            void *ev_scan_begin, void *ev_scan_end);
int nlsh_step_release(nlsh_step_t *step);   /* hold_done steps only: the batch ends HERE on the tail stream */
/* The same batch for a caller that does not pipeline: everything on ONE stream, nothing kept between calls (the stream handles and
 * hold_done of `desc` are ignored; desc_bytes as for nlsh_step_create).  Replaces one whole Indexer.query body (nlsh/indexer.py:56-96)
 * by five launches: nlsh_encode_hash with the bucket lookup of :68 in its epilogue (the keys never leave the workgroup that made them
 * before they are looked up), two launches that lay out the (row segment, query group) tasks, the scan kernel, the merge.
 * row0: index of the batch's first row in the Philox counter of the multi-probe draws (as nlsh_encode_hash's row0: a row range of a
 * larger batch draws what it would have drawn inside it; desc->n_multi_rows counts from the range's first row).
 * lookup_done != 0: skip the hashing and repeat the scan part on the keys the previous call of this batch left in desc->qkeys --
 * the retry after status[1] = 1 (task table too small), with a larger max_tasks / workspace. */
int nlsh_query_batch(const nlsh_step_desc_t *desc, size_t desc_bytes, const float *queries, int64_t q_stride, uint64_t seed,
                     int64_t row0, int lookup_done, void *ev_scan_begin, void *ev_scan_end, nlsh_stream_t stream);
/* 1 while the slot's last batch has not left the tail stream, 0 once it has, < 0 on error.  Never blocks. */
int nlsh_step_busy(nlsh_step_t *step);

/* Merge G per-shard top-k lists per query (keys_in [dev] [G, Q, row_stride] u64, the first k of each
 * row as all-gathered from nlsh_scan_topk's out_keys) into the global top-k; same comparator, so the
 * result equals the single-GPU result.  Candidate counts are summed into out_ncand from ncand_in
 * (nullable) [G, Q], or, when ncand_in is NULL and row_stride > k, from element k of every row (so one
 * collective can carry keys and counts together). */
int nlsh_merge_topk(const uint64_t *keys_in, int64_t row_stride, int G, int64_t Q, int k, const int32_t *ncand_in,
                    float *out_dist, int32_t *out_idx, int32_t *out_ncand, nlsh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NLSH_HIP_H */
