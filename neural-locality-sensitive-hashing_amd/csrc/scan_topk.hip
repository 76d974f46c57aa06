// scan_topk: bucket lookup + candidate distance scan + top-k for a whole query batch.
//
// Replaces the host-driven per-query loop of Indexer.query (nlsh/indexer.py:62-95): dict lookup
// (:68), one index_select launch per (query, key) (:77-82), one distance launch per query
// (:84-87; nlsh/data.py:99-109, 191-201), cat (:88), topk + .tolist() with a device sync per
// query (:90-91).
//
// Three schedules, one contract (selected by `algo`; Indexer.choose_algo picks per batch):
//   0 QUERY-MAJOR (this file): one wavefront per (query, <= seg_rows candidates); every candidate row is
//     fetched once per query that probes it: the HBM-roofline design.  Best when buckets are small and few
//     queries share a bucket (balanced hash): a query's <= P buckets are walked by one wave.
//   1 BUCKET-MAJOR, wave level (scan_bucket.hip): one wavefront per (bucket segment, <= 8 queries held in
//     registers); same lane-partial + DPP tree arithmetic as 0, so 0 and 1 are bit-identical.
//   2 BUCKET-MAJOR, LDS-tiled (scan_bucket.hip): one workgroup per (256-row segment, <= 16 queries), lane
//     owns a row, k-ordered fmaf chain (bit-identical to the oracle), up to 16x less HBM traffic.
//
// gfx950 mapping of the query-major kernel (HBM-bound: 4*d bytes, 3*d flop per candidate)
//   * the corpus is bucket-contiguous (nlsh_gather_rows), so a query's candidate list is a
//     concatenation of <= P contiguous row ranges: streaming, not gathering;
//   * plan kernel: thread per query; binary search of each key in uniq_keys, prefix of bucket
//     sizes, candidates cut into segments of seg_rows -> one task per segment;
//   * scan kernel: ONE wavefront per task, one task per wave of the grid, so the hardware
//     workgroup dispatcher load-balances skewed buckets.  A wave walks its segment in tiles of
//     64 candidates.  LPR lanes cover one row with 16-byte loads (a wave-instruction fetches
//     64/LPR whole rows = 1 KiB, fully coalesced), U wave-loads are kept in flight, the
//     (q-c+eps)^2 / dot partials are reduced across the LPR lanes with DPP adds, and lane l ends
//     up owning exactly one candidate of the tile;
//   * per-wave top-k: the running best-64 (distance,id) keys live sorted across the 64 lanes;
//     a tile is filtered against the k-th key with one ballot and only survivors are inserted
//     (ballot + popcount + one lane shift), so the steady state costs ~3 instructions per tile;
//   * queries whose candidates span several tasks are combined by a small merge kernel with the
//     same comparator; nlsh_merge_topk does the same across corpus shards (multi-GPU).
#include "scan_common.h"

namespace nlsh {

struct ScanArgs {
    const float *corpus;
    long long row_stride;
    int d;
    const int32_t *gid;
    const int32_t *uniq;
    const int32_t *offsets;
    int nb;
    const float *inv_norm;
    const float *queries;
    long long q_stride;
    long long Q;
    const int32_t *qkeys;
    const int32_t *nkeys;
    int P, k, metric, seg;
    float *out_dist;
    int32_t *out_idx;
    uint64_t *out_keys;
    int32_t *out_ncand;
    int32_t *status;
    // workspace
    int32_t *pstart, *pcum, *nseg, *tbase, *task_q, *task_s;
    uint64_t *partial;
    long long max_tasks;
};

// ------------------------------------------------------------------------------------ plan
__global__ __launch_bounds__(256) void plan_kernel(ScanArgs a) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long q = (long long)blockIdx.x * 256 + tid;
    int32_t C = 0, ns = 0;
    if (q < a.Q) {
        int nk = a.nkeys[q];
        nk = nk < 0 ? 0 : (nk > a.P ? a.P : nk);
        for (int p = 0; p < a.P; ++p) {
            int32_t st = 0, sz = 0;
            if (p < nk) {
                const int32_t key = a.qkeys[q * a.P + p];
                bool dup = false;       // a query's keys are a set (nlsh/utils.pyx:27-31): a repeated key probes its bucket once
                for (int pp = 0; pp < p; ++pp) dup |= a.qkeys[q * a.P + pp] == key;
                int lo = 0, hi = dup ? 0 : a.nb;  // lower_bound in the ascending bucket keys
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (a.uniq[mid] < key) lo = mid + 1; else hi = mid;
                }
                if (!dup && lo < a.nb && a.uniq[lo] == key) {  // unknown key = empty bucket (indexer.py:61,68)
                    st = a.offsets[lo];
                    sz = a.offsets[lo + 1] - st;
                }
            }
            a.pstart[q * a.P + p] = st;
            a.pcum[q * a.P + p] = C;
            C += sz;
        }
        ns = (C + a.seg - 1) / a.seg;
        a.out_ncand[q] = C;  // n_candidates, indexer.py:71,94
        if (C == 0)  // no task will touch this query: write the empty result here
            for (int i = 0; i < a.k; ++i) store_topk(a.out_dist, a.out_idx, a.out_keys, q, a.k, KEY_NONE, i);
    }
    // block-exclusive scan of ns, one atomic per block for the global task base
    int32_t incl = ns;
    for (int m = 1; m < 64; m <<= 1) {
        int32_t t = __shfl_up(incl, m);
        if (lane >= m) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    if (tid == 0) {
        int32_t tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        base_s = tot ? atomicAdd(&a.status[0], tot) : 0;
    }
    __syncthreads();
    if (q < a.Q) {
        const int32_t tb = base_s + woff + incl - ns;
        a.nseg[q] = ns;
        a.tbase[q] = tb;
        for (int s = 0; s < ns; ++s) {
            long long t = (long long)tb + s;
            if (t < a.max_tasks) { a.task_q[t] = (int32_t)q; a.task_s[t] = s; }
        }
    }
}

// ------------------------------------------------------------------------------------ scan
// LPR lanes cover one row, VPL 16-byte words per lane (d4 = ceil(d/4) <= LPR*VPL).
template <int LPR, int VPL, int METRIC>
__global__ __launch_bounds__(256) void scan_kernel(ScanArgs a) {
    constexpr int RPI = 64 / LPR;              // rows per wave-load
    constexpr int U = (VPL == 1) ? 8 : (VPL == 2 ? 4 : 2);  // wave-loads in flight
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    long long ntasks = a.status[0];
    if (ntasks > a.max_tasks) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.status[1] = 1;  // incomplete: caller must retry
        ntasks = a.max_tasks;
    }
    if (t >= ntasks) return;

    const long long q = __builtin_amdgcn_readfirstlane(a.task_q[t]);
    const int s = __builtin_amdgcn_readfirstlane(a.task_s[t]);
    const int C = __builtin_amdgcn_readfirstlane(a.out_ncand[q]);
    const int nk = __builtin_amdgcn_readfirstlane(a.nkeys[q]);
    const int v0 = s * a.seg;
    const int v1 = min(C, v0 + a.seg);

    // probe table, one probe per lane
    int cum_l = 0x7FFFFFFF, st_l = 0;
    if (lane < a.P) { cum_l = a.pcum[q * a.P + lane]; st_l = a.pstart[q * a.P + lane]; }
    const int np = nk < a.P ? nk : a.P;

    const int li = lane % LPR, sub = lane / LPR;
    float4 qv[VPL];
    bool act[VPL];
    load_query<LPR, VPL, METRIC>(a.queries + q * a.q_stride, a.d, li, qv, act);

    const float4 *corpus4 = reinterpret_cast<const float4 *>(a.corpus);
    const long long stride4 = a.row_stride >> 2;
    uint64_t top = KEY_NONE, tau = KEY_NONE;

    for (int tile0 = v0; tile0 < v1; tile0 += 64) {
        const int ntile = min(64, v1 - tile0);            // wave-uniform
        const int myc = li * RPI + sub;                   // candidate of the tile this lane owns
        const bool valid = myc < ntile;
        const int myv = tile0 + myc;
        int pidx = -1;                                    // last probe whose first candidate <= myv
        for (int p = 0; p < np; ++p) pidx += (myv >= __builtin_amdgcn_readlane(cum_l, p)) ? 1 : 0;
        pidx = valid ? pidx : 0;
        const int stp = __shfl(st_l, pidx), cmp = __shfl(cum_l, pidx);
        const int prow = valid ? stp + (myv - cmp) : 0;   // row in the bucket-sorted corpus
        const int32_t mygid = valid ? a.gid[prow] : -1;
        float myinv = 0.0f;
        if (METRIC == NLSH_METRIC_COSINE) myinv = valid ? a.inv_norm[prow] : 0.0f;
        float mydist = __builtin_inff();

        for (int j0 = 0; j0 * RPI < ntile; j0 += U) {
            float4 cv[U][VPL];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u;
                const int row = __shfl(prow, sub * LPR + j);
                const bool ok = j * RPI + sub < ntile;
                const float4 *rp = corpus4 + (long long)row * stride4 + li;
#pragma unroll
                for (int v = 0; v < VPL; ++v)
                    cv[u][v] = (ok && act[v]) ? rp[v * LPR] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float tot = group_sum<LPR>(row_partial<VPL, METRIC>(qv, act, cv[u]));
                if (li == j0 + u) mydist = tot;
            }
        }
        const float dist = finish_distance<METRIC>(mydist, myinv);
        const uint64_t key = valid ? make_key(dist, mygid) : KEY_NONE;
        topk_offer(top, tau, key, a.k, lane);
    }

    const int ns = __builtin_amdgcn_readfirstlane(a.nseg[q]);
    if (ns == 1) store_topk(a.out_dist, a.out_idx, a.out_keys, q, a.k, top, lane);
    else if (lane < a.k) a.partial[t * a.k + lane] = top;
}

// ------------------------------------------------------------------------------------ merge
__global__ __launch_bounds__(256) void merge_segments_kernel(ScanArgs a) {
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= a.Q) return;
    const int ns = __builtin_amdgcn_readfirstlane(a.nseg[q]);
    if (ns <= 1) return;
    const long long tb = __builtin_amdgcn_readfirstlane(a.tbase[q]);
    uint64_t top = KEY_NONE, tau = KEY_NONE;
    for (int s = 0; s < ns; ++s) {
        const long long t = tb + s;
        if (t >= a.max_tasks) break;  // overflow: status[1] already set by scan_kernel
        const uint64_t key = lane < a.k ? a.partial[t * a.k + lane] : KEY_NONE;
        topk_offer(top, tau, key, a.k, lane);
    }
    store_topk(a.out_dist, a.out_idx, a.out_keys, q, a.k, top, lane);
}

__global__ __launch_bounds__(256) void merge_shards_kernel(const uint64_t *keys_in, long long row_stride, int G, long long Q, int k,
                                                            const int32_t *ncand_in, float *out_dist, int32_t *out_idx,
                                                            int32_t *out_ncand) {
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    // G shard lists of k keys: 3 x (64 / k) lists per round, the k best of them and of the best so far SELECTED
    // (merge_round), ranked and stored at the end -- no ordered insertion
    const int R = 64 / k, r = lane / k, e = lane - r * k;
    __shared__ uint64_t scratch[4][64];
    uint64_t *sc = scratch[threadIdx.x >> 6];
    uint64_t carry = KEY_NONE;
    for (int base = 0; base < G; base += 3 * R) {
        uint64_t key[4];
        key[0] = carry;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int g = base + s * R + r;
            key[s + 1] = (r < R && g < G) ? keys_in[((long long)g * Q + q) * row_stride + e] : KEY_NONE;
        }
        carry = merge_round<4>(key, k, lane, sc);
    }
    merge_finish(carry, k, lane, out_dist, out_idx, nullptr, q);
    if (out_ncand && (ncand_in || row_stride > k)) {
        int32_t nc = 0;
        for (int g = lane; g < G; g += 64) nc += ncand_in ? ncand_in[(long long)g * Q + q] : (int32_t)keys_in[((long long)g * Q + q) * row_stride + k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) nc += __shfl_xor(nc, m);
        if (lane == 0) out_ncand[q] = nc;
    }
}

struct ScanWs {
    size_t pstart, pcum, nseg, tbase, task_q, task_s, partial, total;
};
static void scan_layout(long long Q, int P, int k, long long max_tasks, ScanWs *w) {
    size_t o = 0;
    w->pstart = o; o += ws_align((size_t)Q * P * 4);
    w->pcum = o;   o += ws_align((size_t)Q * P * 4);
    w->nseg = o;   o += ws_align((size_t)Q * 4);
    w->tbase = o;  o += ws_align((size_t)Q * 4);
    w->task_q = o; o += ws_align((size_t)max_tasks * 4);
    w->task_s = o; o += ws_align((size_t)max_tasks * 4);
    w->partial = o; o += ws_align((size_t)max_tasks * k * 8);
    w->total = o;
}

template <int METRIC>
static void launch_scan(const ScanArgs &a, int d4, unsigned grid, hipStream_t s) {
    if (d4 <= 16) hipLaunchKernelGGL((scan_kernel<16, 1, METRIC>), dim3(grid), dim3(256), 0, s, a);
    else if (d4 <= 32) hipLaunchKernelGGL((scan_kernel<32, 1, METRIC>), dim3(grid), dim3(256), 0, s, a);
    else if (d4 <= 64) hipLaunchKernelGGL((scan_kernel<64, 1, METRIC>), dim3(grid), dim3(256), 0, s, a);
    else if (d4 <= 128) hipLaunchKernelGGL((scan_kernel<64, 2, METRIC>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((scan_kernel<64, 4, METRIC>), dim3(grid), dim3(256), 0, s, a);
}

}  // namespace nlsh

using namespace nlsh;

extern "C" size_t nlsh_scan_workspace(int64_t Q, int P, int k, int64_t max_tasks, int64_t n_buckets, int d) {
    if (Q < 0 || P < 1 || k < 1 || max_tasks < 0 || n_buckets < 0 || d < 1) { set_error("scan_workspace: bad sizes"); return 0; }
    ScanWs w;
    scan_layout(Q, P, k, max_tasks, &w);
    const size_t b = bucket_scan_workspace(Q, P, k, max_tasks, n_buckets, d);
    return w.total > b ? w.total : b;
}

extern "C" int nlsh_scan_topk_cells_phase(const float *corpus_sorted, int64_t row_stride, int d, const int32_t *gid,
                              const int32_t *uniq_keys, const int32_t *offsets, const int32_t *bucket_order, int32_t n_buckets,
                              const int32_t *cell_of, const int32_t *cell_offsets, int32_t n_cells,
                              const float *inv_norm, const float *queries, int64_t q_stride, int64_t Q, const int32_t *qkeys, const int32_t *nkeys,
                              int P, int k, int metric, int algo, int seg_rows, float *out_dist, int32_t *out_idx,
                              uint64_t *out_keys, int32_t *out_ncand, int32_t *status, void *workspace, size_t workspace_bytes,
                              int64_t max_tasks, void *ev_scan_begin, void *ev_scan_end, nlsh_stream_t stream, int phases) {
    NLSH_REQUIRE(phases >= 1 && phases <= (NLSH_PHASE_PLAN | NLSH_PHASE_SCAN | NLSH_PHASE_MERGE), NLSH_E_INVALID, "scan_topk: phases=%d", phases);
    return nlsh::scan_topk_cells_phase_checked(corpus_sorted, row_stride, d, gid, uniq_keys, offsets, bucket_order, n_buckets, cell_of, cell_offsets, n_cells,
                                               inv_norm, queries, q_stride, Q, qkeys, nkeys, P, k, metric, algo, seg_rows, out_dist, out_idx, out_keys,
                                               out_ncand, status, workspace, workspace_bytes, max_tasks, ev_scan_begin, ev_scan_end, stream, phases, 0,
                                               nullptr);
}

int nlsh::scan_topk_cells_phase_checked(const float *corpus_sorted, int64_t row_stride, int d, const int32_t *gid, const int32_t *uniq_keys,
                                        const int32_t *offsets, const int32_t *bucket_order, int32_t n_buckets, const int32_t *cell_of,
                                        const int32_t *cell_offsets, int32_t n_cells, const float *inv_norm, const float *queries,
                                        int64_t q_stride, int64_t Q, const int32_t *qkeys, const int32_t *nkeys, int P, int k, int metric,
                                        int algo, int seg_rows, float *out_dist, int32_t *out_idx, uint64_t *out_keys, int32_t *out_ncand,
                                        int32_t *status, void *workspace, size_t workspace_bytes, int64_t max_tasks, void *ev_scan_begin,
                                        void *ev_scan_end, nlsh_stream_t stream, int phases, int plan_blocks, BucketScanCall *call_out) {
    NLSH_REQUIRE(phases >= 1 && phases <= 15 && !((phases & NLSH_PHASE_PLAN) && (phases & NLSH_PHASE_PLAN_REST)), NLSH_E_INVALID, "scan_topk: phases=%d", phases);
    NLSH_REQUIRE(!(phases & NLSH_PHASE_PLAN_REST) || algo != NLSH_SCAN_QUERY_MAJOR, NLSH_E_INVALID, "scan_topk: the query-major schedule has no fused lookup");
    NLSH_REQUIRE(Q >= 0 && Q < (1ll << 31), NLSH_E_INVALID, "scan_topk: Q=%lld", (long long)Q);
    NLSH_REQUIRE(d >= 1 && d <= NLSH_MAX_DIM, NLSH_E_UNSUPPORTED, "scan_topk: d=%d not in [1,%d]", d, NLSH_MAX_DIM);
    NLSH_REQUIRE(k >= 1 && k <= NLSH_MAX_K, NLSH_E_UNSUPPORTED, "scan_topk: k=%d not in [1,%d]", k, NLSH_MAX_K);
    NLSH_REQUIRE(P >= 1 && P <= NLSH_MAX_PROBES, NLSH_E_UNSUPPORTED, "scan_topk: P=%d not in [1,%d]", P, NLSH_MAX_PROBES);
    NLSH_REQUIRE(metric == NLSH_METRIC_L2_EPS || metric == NLSH_METRIC_COSINE || metric == NLSH_METRIC_L2_EPS_FOLDED, NLSH_E_INVALID, "scan_topk: metric=%d", metric);
    NLSH_REQUIRE(algo == NLSH_SCAN_QUERY_MAJOR || algo == NLSH_SCAN_BUCKET_MAJOR || algo == NLSH_SCAN_BUCKET_TILED, NLSH_E_INVALID,
                 "scan_topk: algo=%d", algo);
    NLSH_REQUIRE(n_buckets >= 0 && max_tasks >= 0 && seg_rows >= 0, NLSH_E_INVALID, "scan_topk: negative size");
    NLSH_REQUIRE(n_cells >= 0 && n_cells <= n_buckets && (n_cells == 0 || (cell_of && cell_offsets)), NLSH_E_INVALID,
                 "scan_topk: n_cells=%d needs cell_of and cell_offsets and cannot exceed n_buckets=%d", (int)n_cells, (int)n_buckets);
    if (Q == 0) return NLSH_OK;
    NLSH_REQUIRE(queries && qkeys && nkeys && out_dist && out_idx && out_ncand && status && workspace, NLSH_E_INVALID, "scan_topk: null pointer");
    NLSH_REQUIRE(n_buckets == 0 || (corpus_sorted && gid && uniq_keys && offsets), NLSH_E_INVALID, "scan_topk: null index pointer");
    NLSH_REQUIRE(metric != NLSH_METRIC_COSINE || n_buckets == 0 || inv_norm, NLSH_E_INVALID, "scan_topk: cosine needs inv_norm");
    NLSH_REQUIRE((row_stride & 3) == 0 && row_stride >= d && ((uintptr_t)corpus_sorted & 15) == 0, NLSH_E_INVALID,
                 "scan_topk: row_stride=%lld must be a multiple of 4 and >= d, corpus 16-byte aligned", (long long)row_stride);
    NLSH_REQUIRE(q_stride >= d, NLSH_E_INVALID, "scan_topk: q_stride < d");
    // the task table and the partial lists are read with 16- and 8-byte accesses at 256-byte offsets of the workspace
    NLSH_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)out_keys & 7) == 0, NLSH_E_INVALID,
                 "scan_topk: the workspace must be 16-byte aligned and out_keys 8-byte aligned");
    if (seg_rows == 0) seg_rows = 512;
    seg_rows = (seg_rows + 63) / 64 * 64;
    hipStream_t s = (hipStream_t)stream;

    if (algo != NLSH_SCAN_QUERY_MAJOR) {
        BucketScanCall c = {corpus_sorted, row_stride, d, gid, uniq_keys, offsets, n_buckets, inv_norm, queries, q_stride, Q,
                            qkeys, nkeys, P, k, metric, seg_rows, out_dist, out_idx, out_keys, out_ncand, status, workspace,
                            workspace_bytes, max_tasks, ev_scan_begin, ev_scan_end, s, algo == NLSH_SCAN_BUCKET_TILED, bucket_order, phases,
                            cell_of, cell_offsets, (int)n_cells, plan_blocks};
        if (call_out) { *call_out = c; return NLSH_OK; }
        return bucket_scan_run(c);
    }
    NLSH_REQUIRE(call_out == nullptr, NLSH_E_INVALID, "scan_topk: the query-major schedule has no call descriptor");

    if (metric == NLSH_METRIC_L2_EPS_FOLDED) metric = NLSH_METRIC_L2_EPS;   // the folded form exists in the tiled schedule only; the exact form is inside its tolerance
    ScanWs w;
    scan_layout(Q, P, k, max_tasks, &w);
    NLSH_REQUIRE(workspace_bytes >= w.total, NLSH_E_WORKSPACE, "scan_topk: workspace %zu < %zu", workspace_bytes, w.total);
    ScanArgs a;
    a.corpus = corpus_sorted; a.row_stride = row_stride; a.d = d; a.gid = gid; a.uniq = uniq_keys; a.offsets = offsets;
    a.nb = n_buckets; a.inv_norm = inv_norm; a.queries = queries; a.q_stride = q_stride; a.Q = Q; a.qkeys = qkeys;
    a.nkeys = nkeys; a.P = P; a.k = k; a.metric = metric; a.seg = seg_rows; a.out_dist = out_dist; a.out_idx = out_idx;
    a.out_keys = out_keys; a.out_ncand = out_ncand; a.status = status; a.max_tasks = max_tasks;
    char *base = (char *)workspace;
    a.pstart = (int32_t *)(base + w.pstart); a.pcum = (int32_t *)(base + w.pcum); a.nseg = (int32_t *)(base + w.nseg);
    a.tbase = (int32_t *)(base + w.tbase); a.task_q = (int32_t *)(base + w.task_q); a.task_s = (int32_t *)(base + w.task_s);
    a.partial = (uint64_t *)(base + w.partial);

    if (phases & NLSH_PHASE_PLAN) {
        NLSH_CHECK_HIP(hipMemsetAsync(status, 0, 2 * sizeof(int32_t), s));
        hipLaunchKernelGGL(plan_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s, a);
    }
    if ((phases & NLSH_PHASE_SCAN) && max_tasks > 0) {
        const unsigned grid = (unsigned)((max_tasks + 3) / 4);
        const int d4 = (d + 3) / 4;
        if (ev_scan_begin) NLSH_CHECK_HIP(hipEventRecord((hipEvent_t)ev_scan_begin, s));
        if (metric == NLSH_METRIC_L2_EPS) launch_scan<NLSH_METRIC_L2_EPS>(a, d4, grid, s);
        else launch_scan<NLSH_METRIC_COSINE>(a, d4, grid, s);
        if (ev_scan_end) NLSH_CHECK_HIP(hipEventRecord((hipEvent_t)ev_scan_end, s));
    }
    if ((phases & NLSH_PHASE_MERGE) && max_tasks > 0)
        hipLaunchKernelGGL(merge_segments_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, s, a);
    NLSH_CHECK_HIP(hipGetLastError());
    return NLSH_OK;
}

extern "C" int nlsh_scan_topk_phase(const float *corpus_sorted, int64_t row_stride, int d, const int32_t *gid,
                              const int32_t *uniq_keys, const int32_t *offsets, const int32_t *bucket_order, int32_t n_buckets,
                              const float *inv_norm, const float *queries, int64_t q_stride, int64_t Q, const int32_t *qkeys, const int32_t *nkeys,
                              int P, int k, int metric, int algo, int seg_rows, float *out_dist, int32_t *out_idx,
                              uint64_t *out_keys, int32_t *out_ncand, int32_t *status, void *workspace, size_t workspace_bytes,
                              int64_t max_tasks, void *ev_scan_begin, void *ev_scan_end, nlsh_stream_t stream, int phases) {
    return nlsh_scan_topk_cells_phase(corpus_sorted, row_stride, d, gid, uniq_keys, offsets, bucket_order, n_buckets, nullptr, nullptr, 0,
                                      inv_norm, queries, q_stride, Q, qkeys, nkeys, P, k, metric, algo, seg_rows, out_dist, out_idx, out_keys,
                                      out_ncand, status, workspace, workspace_bytes, max_tasks, ev_scan_begin, ev_scan_end, stream, phases);
}

extern "C" int nlsh_scan_topk(const float *corpus_sorted, int64_t row_stride, int d, const int32_t *gid,
                              const int32_t *uniq_keys, const int32_t *offsets, const int32_t *bucket_order, int32_t n_buckets,
                              const float *inv_norm, const float *queries, int64_t q_stride, int64_t Q, const int32_t *qkeys,
                              const int32_t *nkeys, int P, int k, int metric, int algo, int seg_rows, float *out_dist,
                              int32_t *out_idx, uint64_t *out_keys, int32_t *out_ncand, int32_t *status, void *workspace,
                              size_t workspace_bytes, int64_t max_tasks, void *ev_scan_begin, void *ev_scan_end,
                              nlsh_stream_t stream) {
    return nlsh_scan_topk_phase(corpus_sorted, row_stride, d, gid, uniq_keys, offsets, bucket_order, n_buckets, inv_norm, queries,
                                q_stride, Q, qkeys, nkeys, P, k, metric, algo, seg_rows, out_dist, out_idx, out_keys, out_ncand,
                                status, workspace, workspace_bytes, max_tasks, ev_scan_begin, ev_scan_end, stream,
                                NLSH_PHASE_PLAN | NLSH_PHASE_SCAN | NLSH_PHASE_MERGE);
}

extern "C" int nlsh_merge_topk(const uint64_t *keys_in, int64_t row_stride, int G, int64_t Q, int k, const int32_t *ncand_in,
                               float *out_dist, int32_t *out_idx, int32_t *out_ncand, nlsh_stream_t stream) {
    NLSH_REQUIRE(G >= 1 && Q >= 0 && k >= 1 && k <= NLSH_MAX_K && row_stride >= k, NLSH_E_INVALID,
                 "merge_topk: G=%d Q=%lld k=%d row_stride=%lld", G, (long long)Q, k, (long long)row_stride);
    if (Q == 0) return NLSH_OK;
    NLSH_REQUIRE(keys_in && out_dist && out_idx, NLSH_E_INVALID, "merge_topk: null pointer");
    hipLaunchKernelGGL(merge_shards_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, (hipStream_t)stream, keys_in, (long long)row_stride, G,
                       (long long)Q, k, ncand_in, out_dist, out_idx, out_ncand);
    NLSH_CHECK_HIP(hipGetLastError());
    return NLSH_OK;
}
