// Bucket-major candidate scan (algos NLSH_SCAN_BUCKET_MAJOR and NLSH_SCAN_BUCKET_TILED of nlsh_scan_topk).
//
// Same contract as the query-major kernel (scan_topk.hip; reference nlsh/indexer.py:62-95), different
// schedule: when many queries of a batch probe the same buckets (SIFT1M-shaped run: 45k distinct
// (query, bucket) pairs over 5k buckets -> every corpus row is a candidate of ~24 queries) the
// query-major kernel re-reads each row once per query and sits on the HBM roofline.  Here the
// (query, probe) pairs are inverted into per-bucket query lists on the device and a row tile is scored
// against a GROUP of queries per fetch:
//   bscan2 (wave-level):  one wavefront = (bucket segment, <= 8 queries in VGPRs); lanes span a row,
//                         cross-lane DPP/permlane reduction per (row, query); same bits as query-major;
//   bscan3 (LDS-tiled):   one workgroup = (256-row segment, <= 16 queries); every lane owns a row of each
//                         tile, queries arrive as scalar loads, no cross-lane traffic, k-ordered fmaf chain
//                         (bit-identical to the oracle), selection by bisection instead of insertion.
//
// Per batch, all on the caller's stream, no host round trip (r06: five launches with the lookup fused into encode_hash, six without):
//   bplan    thread/(query,probe): binary search key -> bucket -> cell; the pair takes its slot in the cell's list (atomicAdd);
//            scan_plan.h -- in the epilogue of encode_hash_kernel when the keys come from there, bplan_kernel for caller-supplied keys
//            (tiled, cosine / folded L2 / d % 4 != 0 only: the padded / pre-normalised query copy is written by the same launch)
//   bscan    thread/cell: reads AND resets the cell's pair counter; block scan + decoupled look-back over the per-block totals ->
//            disjoint pair/task ranges; task = (group of <= QB pairs, segment of <= seg rows), segment-major ids
//   bscatter thread/(query,probe): the pair's record into its cell's query list and its tasks' slot records (no atomics)
//   bscan2 | bscan3: partial top-k per (task, query)
//   bmerge   wave/query: merge the partial lists of its (probe, segment) pairs -> final top-k
// Results do not depend on slot/task order: every list is merged with the (distance, id) comparator.
#include <new>

#include "step_nodes.h"

// Diagnostic build only (make EXTRA=-DNLSH_SCAN_TRACE, tools/scan_trace.py): wave 0 of every bscan3 workgroup
// leaves its phase durations (100 MHz wall_clock64 ticks) in g_scan_trace; the shipped library has neither.
#ifdef NLSH_SCAN_TRACE
#define NLSH_TRACE_SLOTS (1 << 16)
__device__ float g_scan_trace[NLSH_TRACE_SLOTS * 8];
extern "C" int nlsh_debug_scan_trace(float *host, int n_floats) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_scan_trace), (size_t)n_floats * 4);
}
#define SCAN_NOW() wall_clock64()
#else
#define SCAN_NOW() 0ull
#endif

// 16-byte chunks per k-block of the tiled schedule.  4 (64 bytes of every row per stage): 20 KB of LDS and 64 VGPRs
// per workgroup -> 6-7 workgroups per CU; 8 measured 0.404 ms against 0.350 ms at 4 waves/SIMD, 2 the same as 4.
#ifndef NLSH_TILED_KB
#define NLSH_TILED_KB 4
#endif

#ifndef NLSH_ABLATE_NQ
#define NLSH_ABLATE_NQ 4
#endif
// Instruction-arbitration priority of a wave outside / inside the distance loop (s_setprio, 0..3; -1: leave it alone).
#ifndef NLSH_PRIO_OUT
#define NLSH_PRIO_OUT -1
#endif
#ifndef NLSH_PRIO_MATH
#define NLSH_PRIO_MATH -1
#endif
#ifndef NLSH_WARM_QLINES
#define NLSH_WARM_QLINES 1
#endif
#ifndef NLSH_COS_SGPR
#define NLSH_COS_SGPR 1   // 1: the cosine block reads the query chunk straight from SGPRs (68 VGPRs, 7 waves per SIMD, every v_fmac at the SGPR-operand rate: 0.195 ms on the skewed cosine run); 0: from VGPR copies made once per chunk (84 VGPRs, 5 waves: 0.207 ms)
#endif
#ifndef NLSH_FAST_COSINE
#define NLSH_FAST_COSINE 1   // cosine tasks through the hand-scheduled k-blocks too (0: the compiler-scheduled generic loop)
#endif
#ifndef NLSH_FAT_STAGES
#define NLSH_FAT_STAGES 1
#endif
// (NLSH_ABLATE / NLSH_NO_STAGE_BARRIER: diagnostic switches, defined in scan_common.h, live only under -DNLSH_DIAG)
#define NLSH_STAGE_SYNC() do { if (!NLSH_NO_STAGE_BARRIER) __syncthreads(); } while (0)

#ifndef NLSH_TILED_MIN_WAVES
#define NLSH_TILED_MIN_WAVES 1  // min waves per SIMD hint of the tiled kernel (8 = 64 VGPRs / 80 SGPRs: measured equal, 43 SGPR spills)
#endif

#ifndef NLSH_EPS_VGPR
#define NLSH_EPS_VGPR 0  // hand-scheduled L2 block: 0 = eps as a 32-bit literal (8-byte v_add), 1 = eps in a VGPR (4-byte)
#endif

#ifndef NLSH_DEAL_BLOCKS
#define NLSH_DEAL_BLOCKS 0  // tiled one-shot kernel: 0 = a task's queries dealt round-robin over the 4 waves, 1 = in blocks of 4
#endif
#define NLSH_SLOT(wave, jq) (NLSH_DEAL_BLOCKS ? (wave) * QW + (jq) : (wave) + NW * (jq))

#ifndef NLSH_LEAN_EPILOGUE
#define NLSH_LEAN_EPILOGUE 1   // r05: square roots without the range scaling (one wave-uniform guard per list), sign-free key build for L2
#endif
#ifndef NLSH_FAST_KBLOCK
#define NLSH_FAST_KBLOCK 1  // hand-scheduled k-blocks for full L2 tasks (0: compiler-scheduled loop everywhere, for A/B)
#endif


namespace nlsh {

struct BArgs {
    const float *corpus;
    long long row_stride;
    int d;
    const int32_t *gid;
    const int32_t *uniq;
    const int32_t *offsets;
    int nb;
    // Cells (nlsh_build_cells; tiled schedule): the unit the PLAN phase counts pairs and lays out tasks by.  A cell is a big bucket on
    // its own, or a run of consecutive small buckets of at most `window_rows` (<= 256) rows that share one row window -- the
    // queries probing ANY of them share the window's tasks, each with its own row range.  Without cells: cell == bucket.
    const int32_t *cell_of;    // [nb] bucket -> cell, or nullptr
    const int32_t *coffsets;   // [nc + 1] first sorted row of every cell (== offsets without cells)
    int nc;                    // cells (== nb without cells)
    const float *inv_norm;
    const float *queries;
    long long q_stride;
    long long Q;
    const int32_t *qkeys;
    const int32_t *nkeys;
    int P, k, seg, QB;
    float *out_dist;
    int32_t *out_idx;
    uint64_t *out_keys;
    int32_t *out_ncand;
    int32_t *status;
    int32_t *inv_q, *bcount;   // bcount: per CELL
    int4 *ppair;    // [Q*P] what the lookup left per (query, probe) pair (scan_plan.h): {cell, slot in the cell's list, bucket rows, first row inside the cell}
    int4 *cellrec;  // [nc] per cell, written by bscan: {first pair of its list, first task, query groups, 0}
    int4 *prec;  // [Q*P] per (query, probe): {first task of its query group, slot in the group, bucket rows, query groups}; .z = 0: no bucket
    int4 *task;
    int2 *task_qr;     // tiled schedule: [max_tasks][16] {query id, row range lo | hi << 16} of every (task, slot): the task's group's slice of
                       // inv_q, repeated per row segment, and the rows of the query's bucket inside the task's rows (a whole segment of a
                       // big bucket; the bucket's slice of a shared window).  ONE 8-byte record: one store in bscatter, one load in the scan
    uint64_t *partial;
    long long max_tasks;
    const int32_t *border;     // [nc] schedule order of the cells (largest first) or nullptr = CSR order
    unsigned long long *lookback;   // [blocks of bscan] per-block (pairs, tasks, negative counter seen) totals + ready bit, zeroed by the plan launch
    int32_t *hits;             // [blocks of the plan launch] (query, probe) pairs each block counted | PLAN_VIOL_* flags
    unsigned long long *tauq;  // [Q] running upper bound of each query's k-th best key (atomicMin), KEY_NONE-initialised
    const float *qpad;  // tiled variant: queries padded to d4p*4 floats (L2: pad = -eps; cosine: pre-normalised, pad = 0)
    float *qpad_w;      // same buffer, writable (prep_query); qpad aliases `queries` when no padding/normalisation is needed
    long long qpad_stride;
    int d4p;
};

// Bucket lookup of every (query, probe) of a CALLER-SUPPLIED key table (keys from encode_hash are looked up in its own epilogue):
// scan_plan.h.  The coarse table (every `stride`-th key, <= 1024 entries) is loaded once per workgroup.
// Workgroups past `plan_blocks` prepare the tiled schedule's query copy instead (prep_metric >= 0), four queries each.
__global__ __launch_bounds__(256) void bplan_kernel(PlanArgs a, unsigned plan_blocks) {
    __shared__ int32_t coarse[1024];
    __shared__ int whits[4], wviol[4];
    if (blockIdx.x >= plan_blocks) {   // uniform per workgroup: no barrier below is reached by these
        const long long q = (long long)(blockIdx.x - plan_blocks) * 4 + (threadIdx.x >> 6);
        if (q < a.Q) prep_query(a, q, threadIdx.x & 63);
        return;
    }
    for (int i = threadIdx.x; i < a.nco; i += 256) coarse[i] = a.uniq[(long long)i * a.stride];
    if (blockIdx.x == 0) plan_batch_init(a, threadIdx.x, 256);   // per-batch initialisation rides along (no separate launch)
    __syncthreads();
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    int viol = 0;
    if (idx < a.Q * a.P) {
        const long long q = idx / a.P;
        const int p = (int)(idx - q * a.P);
        if (p == 0) a.tauq[q] = KEY_NONE;        // running bound of the query
        int nk = a.qnkeys[q];
        nk = nk < 0 ? 0 : (nk > a.P ? a.P : nk);
        int32_t key = 0;
        bool live = p < nk;
        if (live) {
            key = a.qkeys[idx];
            // a query's keys are a SET (nlsh/utils.pyx:27-31): a repeated key probes its bucket once.  encode_hash never
            // emits one; a C caller's table might, and the selection-based merges assume distinct (distance, id) keys.
            for (int pp = 0; pp < p; ++pp) live &= a.qkeys[idx - p + pp] != key;
        }
        hit = plan_pair(a, coarse, idx, key, live, viol);
    }
    // pairs this block added to the counters: bscan_kernel holds the counters' sum against the sum of these, which is how
    // a workspace head that was not zero on entry (workspace contract, nlsh_hip.h) is caught instead of trusted
    const unsigned long long m = __ballot(hit);
    const unsigned long long v1 = __ballot(viol & PLAN_VIOL_COUNTER), v2 = __ballot(viol & PLAN_VIOL_CELLS);
    if ((threadIdx.x & 63) == 0) {
        whits[threadIdx.x >> 6] = __popcll(m);
        wviol[threadIdx.x >> 6] = (v1 ? PLAN_VIOL_COUNTER : 0) | (v2 ? PLAN_VIOL_CELLS : 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) a.hits[blockIdx.x] = (whits[0] + whits[1] + whits[2] + whits[3]) | wviol[0] | wviol[1] | wviol[2] | wviol[3];
}

// exclusive scan of a 64-bit value over the 256 threads of a block (two packed 32-bit sums scanned together: the low word must not
// carry into the high one, which the callers' ranges guarantee: low = pairs <= Q * P < 2^31)
__device__ __forceinline__ unsigned long long block_excl_scan64(unsigned long long v, unsigned long long *wsum, unsigned long long *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = v;
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned lo = __shfl_up((unsigned)incl, m), hi = __shfl_up((unsigned)(incl >> 32), m);
        if (lane >= m) incl += ((unsigned long long)hi << 32) | lo;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return woff + incl - v;
}

// Tasks are numbered in SCHEDULE order: cell `border[i]` (cells by descending size, fixed at index build)
// is handled by thread i, so the heavy (segment x query-group) tasks of the big buckets get the low ids and are
// dispatched first, the single small tasks of the small buckets last: the kernel no longer ends on a tail of
// 40-us tasks started in its last microseconds (measured: machine full until 370 us, drained until 440 us).
// Numbering is deterministic (no dependence on block timing): block j publishes its totals, block i sums the
// totals of the blocks before it.
//
// r06: ONE launch (r02-r05: bcount_kernel left the per-block totals, a second launch summed them: a dependent launch costs ~4.5 us
// on this part whatever it does).  Decoupled look-back without the chain: a block publishes its OWN totals -- they depend on nothing
// but its own 256 counters -- as one 64-bit word {tasks : 32 | pairs : 30 | negative counter seen : 1 | ready : 1} with a
// device-scope store, then thread t waits for the words of blocks t, t + 256, ... < blockIdx.x and the block adds them up.  Blocks
// are dispatched in index order and a block only ever waits for LOWER indices, which wait for nothing unfinished: the lowest
// unfinished block always runs, so the wait cannot deadlock whatever the grid size.  The slots are zeroed by the plan launch in
// front of this one (plan_batch_init).  Device-scope (sc1) accesses to the slots only: no fence, no L2 write-back (r02 measured an
// in-kernel grid barrier with agent-scope fences at 127 us -- each writes back an XCD's L2).
//
// The pair counters are READ AND RESET here (thread per cell): the slot of every pair was fixed when the lookup incremented the
// counter (scan_plan.h), so nobody needs the counts after this kernel, and the head of the workspace is zero again for the next
// batch (workspace contract) without the scatter step's atomicSub of r01-r05.
//
// status[1] = 2: the pair counters were not zero when the batch's lookup started (an uninitialised buffer, one lent to another
// schedule).  The LAST block knows every total: the counters must sum to the pairs the lookup counted this batch, none may be
// negative and no pair may have drawn a negative slot.  On a violation status[0] = 0 -- the scan launches no task -- and bscatter,
// which starts after this kernel, drops every pair: nothing is ever addressed through a stale count (negative counts are clamped
// to zero before they enter a prefix; descriptors are written inside the table only; the pair lists are only written by bscatter).
// The facade turns the flag into NLSH_E_WORKSPACE.  status[1] = 3: the lookup refused a pair of a foreign cell layout (scan_plan.h).
constexpr unsigned long long LB_READY = 1ull, LB_NEG = 2ull;
__device__ __forceinline__ void lookback_publish(unsigned long long *slot, unsigned long long v) {
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lookback_wait(unsigned long long *slot) {
    unsigned long long v;
    do { v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(v & LB_READY));
    return v;
}

__global__ __launch_bounds__(256) void bscan_kernel(BArgs a, int plan_blocks) {
    __shared__ unsigned long long wsum[4];
    __shared__ unsigned long long base_s;
    __shared__ int neg_s, flags_s;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) { neg_s = 0; flags_s = 0; }
    const bool last_block = blockIdx.x == gridDim.x - 1;
    // the last block's verdict needs the lookup's per-block pair counts: requested NOW, with the first loads of the kernel, not behind the
    // look-back (the last block ends the kernel: a round trip on its path is a round trip of the launch)
    int hsum = 0, hflags = 0;
    if (last_block)
        for (int j = threadIdx.x; j < plan_blocks; j += 256) {
            const int h = a.hits[j];
            hsum += h & PLAN_HITS_MASK;
            hflags |= h & ~PLAN_HITS_MASK;
        }
    int b = 0, m = 0, s = 0, ns = 0, ng = 0;
    bool neg = false;
    if (i < a.nc) {
        b = a.border ? a.border[i] : i;   // a CELL (a bucket when the index has no cells)
        m = a.bcount[b];
        if (m != 0) a.bcount[b] = 0;      // handed back: zero again for the next batch
        neg = m < 0;
        m = neg ? 0 : m;                  // a stale negative count enters no prefix (the batch is refused below)
        s = a.coffsets[b + 1] - a.coffsets[b];
        ns = (s + a.seg - 1) / a.seg;
        ng = (m + a.QB - 1) / a.QB;
    }
    const int nt = ng * ns;
    unsigned long long tot;
    const unsigned long long ex = block_excl_scan64(((unsigned long long)(unsigned)nt << 32) | (unsigned)m, wsum, &tot);   // (barriers inside: neg_s is initialised)
    if (neg) neg_s = 1;
    __syncthreads();
    if (threadIdx.x == 0)
        lookback_publish(a.lookback + blockIdx.x, (tot & 0xFFFFFFFF00000000ull) | ((tot & 0x3FFFFFFFull) << 2) | (neg_s ? LB_NEG : 0ull) | LB_READY);
    // totals of the blocks before this one
    unsigned long long prev = 0;
    int pneg = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) {
        const unsigned long long v = lookback_wait(a.lookback + j);
        prev += (v & 0xFFFFFFFF00000000ull) | ((v >> 2) & 0x3FFFFFFFull);
        pneg |= (v & LB_NEG) ? 1 : 0;
    }
    {
        unsigned long long ptot;
        block_excl_scan64(prev, wsum, &ptot);
        if (threadIdx.x == 0) base_s = ptot;
        if (pneg) neg_s = 1;
    }
    if (last_block) {   // the verdict: every total is known here
        unsigned long long htot;
        __syncthreads();
        block_excl_scan64((unsigned long long)(unsigned)hsum, wsum, &htot);
        if (hflags) atomicOr(&flags_s, hflags);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long all = base_s + tot;
            const bool bad = (all & 0xFFFFFFFFull) != htot || neg_s != 0 || (flags_s & PLAN_VIOL_COUNTER);
            a.status[0] = bad ? 0 : (int)(all >> 32);   // tasks needed (may exceed max_tasks: the caller retries)
            a.status[1] = bad ? 2 : ((flags_s & PLAN_VIOL_CELLS) ? 3 : 0);
        }
    }
    __syncthreads();
    // A cell's descriptors are written by its own thread (writing a block's tasks with all its threads, one task per thread and
    // round, measured 15.6 us against 9.5 us on the SIFT1M-shaped batch) -- unless it has more than HEAVY of them: a 180 k-row
    // bucket probed by thousands of queries (Deep100M-shaped: 707 segments x hundreds of query groups) is 10^5 descriptors, and one
    // thread writing them made this kernel 1.0 ms of a 37-ms step (r03).  Heavy cells are queued in LDS and written by the whole block.
    constexpr int HEAVY = 256, HEAVY_SLOTS = 256;
    __shared__ int heavy_n;
    __shared__ int heavy_b[HEAVY_SLOTS][6];   // po, to, ng, nt, m, cell (row0 and size are re-read)
    if (threadIdx.x == 0) heavy_n = 0;
    __syncthreads();
    if (i < a.nc) {
        const unsigned long long off = base_s + ex;
        const int po = (int)(off & 0xFFFFFFFFull), to = (int)(off >> 32);
        a.cellrec[b] = make_int4(po, to, ng, 0);
        if (nt > HEAVY) {
            const int slot = atomicAdd(&heavy_n, 1);   // <= 256 threads, so a slot always exists
            heavy_b[slot][0] = po; heavy_b[slot][1] = to; heavy_b[slot][2] = ng; heavy_b[slot][3] = nt; heavy_b[slot][4] = m; heavy_b[slot][5] = b;
        } else {
            const int row0 = a.coffsets[b];
            for (int t = 0; t < nt; ++t) {
                const long long tt = (long long)to + t;
                if (tt >= a.max_tasks) break;
                // segment-major: the query groups of one row segment get consecutive task ids, so they run
                // at about the same time (and, with the chunked XCD map of bscan3, on one XCD's L2)
                const int si = t / ng, gi = t - si * ng;
                a.task[tt] = make_int4(po + gi * a.QB, min(a.QB, m - gi * a.QB), row0 + si * a.seg, min(a.seg, s - si * a.seg));
            }
        }
    }
    __syncthreads();
    for (int h = 0; h < heavy_n; ++h) {
        const int po = heavy_b[h][0], to = heavy_b[h][1], hng = heavy_b[h][2], hnt = heavy_b[h][3], hm = heavy_b[h][4], hb = heavy_b[h][5];
        const int row0 = a.coffsets[hb], hs = a.coffsets[hb + 1] - row0;
        for (int t = threadIdx.x; t < hnt; t += 256) {
            const long long tt = (long long)to + t;
            if (tt >= a.max_tasks) break;
            const int si = t / hng, gi = t - si * hng;
            a.task[tt] = make_int4(po + gi * a.QB, min(a.QB, hm - gi * a.QB), row0 + si * a.seg, min(a.seg, hs - si * a.seg));
        }
    }
}

// Every counted pair into its cell's query list and its tasks' slot records.  r06: no atomics and one dependent load level -- the
// lookup left {cell, slot, bucket rows, first row inside the cell} per pair (scan_plan.h) and bscan one {first pair, first task,
// query groups} record per cell (r01-r05: atomicSub on the cell's counter for the slot, then five loads behind it).
__global__ __launch_bounds__(256) void bscatter_kernel(BArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.Q * a.P) return;
    const int4 pp = a.ppair[idx];
    if (pp.x < 0 || a.status[1] == 2) {   // no bucket, or a poisoned workspace (bscan_kernel): nothing is addressed through the counters
        a.prec[idx] = make_int4(0, 0, 0, 0);
        return;
    }
    const int rel = pp.y, size = pp.z, lo0 = pp.w;   // slot in the cell's pair list; rows of the BUCKET; its first row inside the cell (0 for a bucket that is its own cell)
    const int4 cr = a.cellrec[pp.x];
    a.inv_q[cr.x + rel] = (int32_t)(idx / a.P);
    // what bmerge needs to find this probe's partial lists, resolved here so that it has one load level less:
    // task of (segment si, group gi) = first task of the cell + si * ngroups + gi
    const int gi = rel / a.QB;
    // `size` = rows of the BUCKET: the query's candidate count, and (size + seg - 1) / seg = its partial lists -- one per row segment of
    // a big bucket, exactly one for a bucket inside a shared window (window_rows <= seg; the lookup refuses cells that break this)
    const int t0 = cr.y + gi, ng = cr.z;
    a.prec[idx] = make_int4(t0, rel - gi * a.QB, size, ng);
    if (a.task_qr) {
        // the tiled scan reads a task's query ids from the task's own record (address known from the task id alone: the
        // ids arrive with the descriptor instead of one dependent round trip later); one copy per row segment
        const int ns = (size + a.seg - 1) / a.seg;
        for (int si = 0; si < ns; ++si) {
            const long long tt = (long long)t0 + (long long)si * ng;
            if (tt >= a.max_tasks) break;
            const int lo = max(lo0 - si * a.seg, 0), hi = min(lo0 + size - si * a.seg, a.seg);   // the bucket's rows inside segment si
            a.task_qr[tt * a.QB + (rel - gi * a.QB)] = make_int2((int32_t)(idx / a.P), lo | (hi << 16));
        }
    }
}

// One task: stream `nrows` rows starting at row0 once, score them against the nq <= QB queries of
// the group.  FULL (nq == QB, the common case in hot buckets) compiles without per-query branches.
template <int LPR, int VPL, int METRIC, int QB, bool FULL>
__device__ __forceinline__ void bscan2_task(const BArgs &a, long long t, int pair0, int nq, int row0, int nrows, int lane) {
    constexpr int RPI = 64 / LPR;
    constexpr int U = (VPL == 1) ? 4 : (VPL == 2 ? 2 : 1);  // wave-loads per pipeline stage (two stages in flight)
    const int li = lane % LPR, sub = lane / LPR;
    float4 qv[QB][VPL];
    bool act[VPL];
    uint64_t top[QB], tau[QB];
#pragma unroll
    for (int jq = 0; jq < QB; ++jq) {
        top[jq] = KEY_NONE;
        tau[jq] = KEY_NONE;
        // clamped into [0, Q): a slot that a stale counter invented (workspace contract violated; bmerge flags it) holds whatever
        // the buffer held, and nothing may be addressed through it
        const int qi = min(max(__builtin_amdgcn_readfirstlane(a.inv_q[pair0 + ((FULL || jq < nq) ? jq : 0)]), 0), (int)a.Q - 1);
        load_query<LPR, VPL, METRIC>(a.queries + (long long)qi * a.q_stride, a.d, li, qv[jq], act);
    }

    // The segment is walked in groups of U wave-loads (U*RPI rows); groups are software-pipelined
    // through two register sets so the next group's HBM/L2 latency hides under the current group's
    // QB*U distance evaluations.  LPR/U groups make one 64-row tile (one candidate per lane).
    constexpr int GPT = LPR / U;
    const float4 *seg4 = reinterpret_cast<const float4 *>(a.corpus) + (long long)row0 * (a.row_stride >> 2) + li;
    const long long stride4 = a.row_stride >> 2;
    const int G = (nrows + U * RPI - 1) / (U * RPI);
    const int myc = li * RPI + sub;  // candidate of a tile this lane owns
    float mydist[QB];
    int32_t mygid = -1;
    float myinv = 0.0f;
    bool valid = false;

    auto load_group = [&](float4 (&cv)[U][VPL], int g) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = (g * U + u) * RPI + sub;  // row of the segment this lane group covers
            const bool ok = r < nrows;
            const float4 *rp = seg4 + (long long)r * stride4;
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                cv[u][v] = (ok && act[v]) ? rp[v * LPR] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto compute_group = [&](const float4 (&cv)[U][VPL], int g) {
        const int gi = g % GPT;
        if (gi == 0) {  // tile begin
            const int tile0 = (g / GPT) * 64;
            valid = tile0 + myc < nrows;
            const int prow = row0 + tile0 + (valid ? myc : 0);
            mygid = valid ? a.gid[prow] : -1;
            if (METRIC == NLSH_METRIC_COSINE) myinv = valid ? a.inv_norm[prow] : 0.0f;
#pragma unroll
            for (int jq = 0; jq < QB; ++jq) mydist[jq] = __builtin_inff();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool mine = li == gi * U + u;
#pragma unroll
            for (int jq = 0; jq < QB; ++jq) {
                if (FULL || jq < nq) {
                    const float tot = group_sum<LPR>(row_partial<VPL, METRIC>(qv[jq], act, cv[u]));
                    mydist[jq] = mine ? tot : mydist[jq];
                }
            }
        }
        if (gi == GPT - 1 || g == G - 1) {  // tile end: offer this lane's candidate to every query's list
#pragma unroll
            for (int jq = 0; jq < QB; ++jq) {
                if (FULL || jq < nq) {
                    const float dist = finish_distance<METRIC>(mydist[jq], myinv);
                    const uint64_t key = valid ? make_key(dist, mygid) : KEY_NONE;
                    topk_offer(top[jq], tau[jq], key, a.k, lane);
                }
            }
        }
    };

    float4 cvA[U][VPL], cvB[U][VPL];
    if (G > 0) load_group(cvA, 0);
    for (int g = 0; g < G; g += 2) {
        if (g + 1 < G) load_group(cvB, g + 1);
        compute_group(cvA, g);
        if (g + 1 >= G) break;
        if (g + 2 < G) load_group(cvA, g + 2);
        compute_group(cvB, g + 1);
    }
#pragma unroll
    for (int jq = 0; jq < QB; ++jq)
        if ((FULL || jq < nq) && lane < a.k) a.partial[((long long)t * QB + jq) * a.k + lane] = top[jq];
}

template <int LPR, int VPL, int METRIC, int QB>
__global__ __launch_bounds__(256) void bscan2_kernel(BArgs a) {
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    long long ntasks = a.status[0];
    if (ntasks > a.max_tasks) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&a.status[1], 1);  // incomplete: caller must retry (a refusal of the PLAN phase -- 2, 3 -- stays)
        ntasks = a.max_tasks;
    }
    if (t >= ntasks) return;
    const int4 desc = a.task[t];
    const int pair0 = __builtin_amdgcn_readfirstlane(desc.x);
    const int nq = __builtin_amdgcn_readfirstlane(desc.y);
    const int row0 = __builtin_amdgcn_readfirstlane(desc.z);
    const int nrows = __builtin_amdgcn_readfirstlane(desc.w);
    if (nq == QB) bscan2_task<LPR, VPL, METRIC, QB, true>(a, t, pair0, nq, row0, nrows, lane);
    else bscan2_task<LPR, VPL, METRIC, QB, false>(a, t, pair0, nq, row0, nrows, lane);
}

// ------------------------------------------------------------------------------------ tiled variant
// NLSH_SCAN_BUCKET_TILED: one WORKGROUP per task = (256-row bucket segment, group of <= 16 queries).
// The segment goes through LDS one k-block (KB 16-byte chunks of every row) at a time (coalesced 16-byte
// global loads -> ds_write_b128, odd row stride = conflict-free column reads) and every lane OWNS ONE ROW of
// each 64-row tile: it walks the row in k order and updates QW query accumulators per tile, the query values
// arriving as wave-uniform scalar loads (s_load from the queries, or from a padded / pre-normalised copy when
// the prepared query copy is needed).  No cross-lane reduction at all: 3 VALU per element and query for L2 ((q-c), +eps, fma),
// 1 for cosine; the distance of lane l's row is a k-ascending fmaf chain, bit-identical to the oracle's scalar
// loop.  Four waves share the tile, so a row is fetched from HBM/L2 once per 16 queries.  The next k-block's
// global loads are issued before the current one is computed (load early, ds_write late).
//
// What bounds it (r01 traces, tools/scan_trace.py, tools/probe_l2_loop.hip): the inner loop's instruction mix
// sustains 1 VALU / 2.5-2.9 cycles per SIMD in isolation; the kernel reaches ~60 % of that because a wave spends
// ~35 % of a task outside the distance loop (stage barriers, top-k selection) and only resident waves of OTHER
// workgroups fill those gaps -- occupancy is the lever that paid (KB 8 -> 4: 4 -> 6-7 workgroups per CU, 0.40 ->
// 0.34 ms).  Measured and dropped: branch-free loop bodies specialised on (queries, tiles) per wave, with and
// without hand-placed LDS/SMEM double buffering (the scheduler keeps the scalar query chunks in VGPRs: 110-150
// VGPRs, 3-4 waves/SIMD, 0.43-0.49 ms); reading the next step's row chunk one step ahead (+3 %); reading all
// tiles' chunks of a step up front (+-0); a second copy of the loop without the per-query guards for waves that
// hold all QW queries (79 VGPRs -> 6 waves/SIMD: 0.363 vs 0.344 ms); requesting a k-block's first query chunk before
// the stage barriers (0.346 vs 0.337 ms: the barrier's lgkmcnt(0) then also waits for the scalar loads); touching the
// next 64-byte query line with a dummy scalar load one line ahead (SGPR spills 28 -> 50: 0.355 vs 0.331 ms); one
// s_load_dwordx16 per query line instead of four x4 (64 VGPRs kept, +-0: the scalar loads are not what waves wait for).
typedef const __attribute__((address_space(4))) float *const_f32p;

template <int QW>
struct QChunk { float v[QW][4]; };  // wave-uniform: lives in SGPRs

template <int QW, bool FULL>
__device__ __forceinline__ void load_qchunk(QChunk<QW> &qc, const const_f32p (&qs)[QW], int nqw, int c) {
#pragma unroll
    for (int jq = 0; jq < QW; ++jq)
        if (FULL || jq < nqw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) qc.v[jq][e] = NLSH_ABLATE == 4 ? (float)(c + e) : qs[jq][4 * c + e];  // one s_load_dwordx4 per query
        }
}

template <int METRIC, int QW, bool FULL>
__device__ __forceinline__ void apply_qchunk(const QChunk<QW> &qc, const float4 rv, int nqw, float (&acc)[QW]) {
#pragma unroll
    for (int jq = 0; jq < QW; ++jq) {
        if (FULL || jq < nqw) {
            const float q0 = qc.v[jq][0], q1 = qc.v[jq][1], q2 = qc.v[jq][2], q3 = qc.v[jq][3];
            if (METRIC == NLSH_METRIC_L2_EPS) {
                // F.pairwise_distance: || (x1 - x2) + eps ||, summed in k order (nlsh/data.py:201)
                const float t0 = (q0 - rv.x) + 1e-6f, t1 = (q1 - rv.y) + 1e-6f, t2 = (q2 - rv.z) + 1e-6f, t3 = (q3 - rv.w) + 1e-6f;
                acc[jq] = fmaf(t3, t3, fmaf(t2, t2, fmaf(t1, t1, fmaf(t0, t0, acc[jq]))));
            } else if (METRIC == NLSH_METRIC_L2_EPS_FOLDED) {
                const float t0 = q0 - rv.x, t1 = q1 - rv.y, t2 = q2 - rv.z, t3 = q3 - rv.w;   // q already carries + eps (prep_query)
                acc[jq] = fmaf(t3, t3, fmaf(t2, t2, fmaf(t1, t1, fmaf(t0, t0, acc[jq]))));
            } else {
                acc[jq] = fmaf(q3, rv.w, fmaf(q2, rv.z, fmaf(q1, rv.y, fmaf(q0, rv.x, acc[jq]))));
            }
        }
    }
}

// ---- hand-scheduled k-block of a FULL L2 task (4 queries per wave x 4 row tiles x 4 chunks) ------------------------------
// r02 finding (ISA + SQ counters of the compiler-scheduled loop): every (tile, query) block sat behind two or three
// uniform branches and an `s_waitcnt lgkmcnt(0)` placed directly after its `ds_read_b128` -- the LDS round trip was exposed
// 16 times per chunk step and the scalar loads of the next chunk (same counter) were waited for as soon as they were
// issued; waves spent as many cycles stalled at issue as executing.  Full tasks are 29 % of the tasks and most of the
// arithmetic, so their k-blocks run this straight-line form instead: VALU in inline asm (the compiler cannot re-order or
// re-guard it), the row chunk of tile t+1 and the query chunk c+1 requested one block (48 VALU) ahead, no branches.
// Same arithmetic in the same order as apply_qchunk: (q - c) + eps, fmaf chain in ascending k -> bit-identical results.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// eps is the 32-bit LITERAL, not an SGPR: tools/probe_l2_block.hip measured 2.2 cycles per VALU for this block with the
// literal against 2.9-3.4 with eps in an SGPR (a VALU instruction that reads an SGPR issues slower on gfx950: sub/add
// with SGPR operands only, 4.1 cycles) -- only the v_sub reads one (the query value).
__device__ __forceinline__ void l2_query_block(float &a, const float4 r, const f32x4 q) {
    float t0, t1, t2, t3;
#if NLSH_EPS_VGPR
    float veps = 1e-6f;
    asm volatile("" : "+v"(veps));   // eps in a VGPR: 4-byte encodings (48 instead of 64 bytes of instruction stream per block)
    asm volatile(
        "v_sub_f32 %[t0], %[q0], %[r0]\n\t"
        "v_sub_f32 %[t1], %[q1], %[r1]\n\t"
        "v_sub_f32 %[t2], %[q2], %[r2]\n\t"
        "v_sub_f32 %[t3], %[q3], %[r3]\n\t"
        "v_add_f32 %[t0], %[e], %[t0]\n\t"
        "v_add_f32 %[t1], %[e], %[t1]\n\t"
        "v_add_f32 %[t2], %[e], %[t2]\n\t"
        "v_add_f32 %[t3], %[e], %[t3]\n\t"
        "v_fmac_f32 %[a], %[t0], %[t0]\n\t"
        "v_fmac_f32 %[a], %[t1], %[t1]\n\t"
        "v_fmac_f32 %[a], %[t2], %[t2]\n\t"
        "v_fmac_f32 %[a], %[t3], %[t3]"
        : [a] "+v"(a), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
        : [r0] "v"(r.x), [r1] "v"(r.y), [r2] "v"(r.z), [r3] "v"(r.w), [q0] "s"(q.x), [q1] "s"(q.y), [q2] "s"(q.z), [q3] "s"(q.w), [e] "v"(veps));
#else
    asm volatile(
        "v_sub_f32 %[t0], %[q0], %[r0]\n\t"
        "v_sub_f32 %[t1], %[q1], %[r1]\n\t"
        "v_sub_f32 %[t2], %[q2], %[r2]\n\t"
        "v_sub_f32 %[t3], %[q3], %[r3]\n\t"
        "v_add_f32 %[t0], 0x358637bd, %[t0]\n\t"
        "v_add_f32 %[t1], 0x358637bd, %[t1]\n\t"
        "v_add_f32 %[t2], 0x358637bd, %[t2]\n\t"
        "v_add_f32 %[t3], 0x358637bd, %[t3]\n\t"
        "v_fmac_f32 %[a], %[t0], %[t0]\n\t"
        "v_fmac_f32 %[a], %[t1], %[t1]\n\t"
        "v_fmac_f32 %[a], %[t2], %[t2]\n\t"
        "v_fmac_f32 %[a], %[t3], %[t3]"
        : [a] "+v"(a), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
        : [r0] "v"(r.x), [r1] "v"(r.y), [r2] "v"(r.z), [r3] "v"(r.w), [q0] "s"(q.x), [q1] "s"(q.y), [q2] "s"(q.z), [q3] "s"(q.w));
#endif
}

struct QSet { f32x4 v[4]; };   // one 16-byte chunk of each of the wave's (up to) 4 queries: 16 SGPRs

// NQ s_load_dwordx4 (query pointer + byte offset in an SGPR) that the compiler can neither merge into wider loads (x8
// pairs need 64 SGPRs for a double buffer and spilled) nor move; the caller waits for them with s_waitcnt lgkmcnt(0)
// before the first VALU block that reads them.
// `after`: a VGPR the loads pretend to read -- the row chunk the NEXT VALU block consumes.  LDS and scalar loads share
// one counter and scalar loads return out of order, so any wait for LDS data also waits for scalar loads in flight: the
// compiler's wait for that row chunk is thereby placed BEFORE these loads are issued, and they get a whole VALU block
// of cover before the next wait.
template <int NQ>
__device__ __forceinline__ void load_qset(QSet &q, const const_f32p (&qk)[4], int byte_off, float after) {
    if (NQ == 4)
        asm volatile("s_load_dwordx4 %0, %4, %8\n\ts_load_dwordx4 %1, %5, %8\n\ts_load_dwordx4 %2, %6, %8\n\ts_load_dwordx4 %3, %7, %8"
                     : "=&s"(q.v[0]), "=&s"(q.v[1]), "=&s"(q.v[2]), "=&s"(q.v[3])
                     : "s"(qk[0]), "s"(qk[1]), "s"(qk[2]), "s"(qk[3]), "s"(byte_off), "v"(after));
    else if (NQ == 3)
        asm volatile("s_load_dwordx4 %0, %3, %6\n\ts_load_dwordx4 %1, %4, %6\n\ts_load_dwordx4 %2, %5, %6"
                     : "=&s"(q.v[0]), "=&s"(q.v[1]), "=&s"(q.v[2])
                     : "s"(qk[0]), "s"(qk[1]), "s"(qk[2]), "s"(byte_off), "v"(after));
    else if (NQ == 2)
        asm volatile("s_load_dwordx4 %0, %2, %4\n\ts_load_dwordx4 %1, %3, %4"
                     : "=&s"(q.v[0]), "=&s"(q.v[1])
                     : "s"(qk[0]), "s"(qk[1]), "s"(byte_off), "v"(after));
    else
        asm volatile("s_load_dwordx4 %0, %1, %2" : "=&s"(q.v[0]) : "s"(qk[0]), "s"(byte_off), "v"(after));
}

// One (tile, chunk) block = NQ query blocks in ONE asm statement: the compiler pads every inline-asm statement with an
// s_nop (it cannot see the hazards inside), and per-query statements left 25 of them per chunk pair in the hot loop.
#define NLSH_QBLK(J)                                                                                               \
    "v_sub_f32 %[t0], %[q" #J "0], %[r0]\n\tv_sub_f32 %[t1], %[q" #J "1], %[r1]\n\t"                                 \
    "v_sub_f32 %[t2], %[q" #J "2], %[r2]\n\tv_sub_f32 %[t3], %[q" #J "3], %[r3]\n\t"                                 \
    "v_add_f32 %[t0], 0x358637bd, %[t0]\n\tv_add_f32 %[t1], 0x358637bd, %[t1]\n\t"                                   \
    "v_add_f32 %[t2], 0x358637bd, %[t2]\n\tv_add_f32 %[t3], 0x358637bd, %[t3]\n\t"                                   \
    "v_fmac_f32 %[a" #J "], %[t0], %[t0]\n\tv_fmac_f32 %[a" #J "], %[t1], %[t1]\n\t"                                 \
    "v_fmac_f32 %[a" #J "], %[t2], %[t2]\n\tv_fmac_f32 %[a" #J "], %[t3], %[t3]\n\t"
// The 2-op form (NLSH_METRIC_L2_EPS_FOLDED): eps is folded into the query copy prep_query writes, a block is v_sub + v_fmac -- 8
// instead of 12 VALU per chunk and query.  (q + eps) - c rounds differently from (q - c) + eps, so it is NOT the oracle's bits:
// an opt-in within the north_star's 1e-4 tolerance, never the default.
#define NLSH_QBLK2(J)                                                                                              \
    "v_sub_f32 %[t0], %[q" #J "0], %[r0]\n\tv_sub_f32 %[t1], %[q" #J "1], %[r1]\n\t"                                 \
    "v_sub_f32 %[t2], %[q" #J "2], %[r2]\n\tv_sub_f32 %[t3], %[q" #J "3], %[r3]\n\t"                                 \
    "v_fmac_f32 %[a" #J "], %[t0], %[t0]\n\tv_fmac_f32 %[a" #J "], %[t1], %[t1]\n\t"                                 \
    "v_fmac_f32 %[a" #J "], %[t2], %[t2]\n\tv_fmac_f32 %[a" #J "], %[t3], %[t3]\n\t"
#define NLSH_QIN(J) [q##J##0] "s"(q.v[J].x), [q##J##1] "s"(q.v[J].y), [q##J##2] "s"(q.v[J].z), [q##J##3] "s"(q.v[J].w)
#define NLSH_TMP [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
#define NLSH_RIN [r0] "v"(r.x), [r1] "v"(r.y), [r2] "v"(r.z), [r3] "v"(r.w)
template <int NQ>
__device__ __forceinline__ void l2f_tile_block(float (&acc)[4], const float4 r, const QSet &q) {
    float t0, t1, t2, t3;
    if (NQ == 4)
        asm volatile(NLSH_QBLK2(0) NLSH_QBLK2(1) NLSH_QBLK2(2) NLSH_QBLK2(3)
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), NLSH_TMP
                     : NLSH_RIN, NLSH_QIN(0), NLSH_QIN(1), NLSH_QIN(2), NLSH_QIN(3));
    else if (NQ == 3)
        asm volatile(NLSH_QBLK2(0) NLSH_QBLK2(1) NLSH_QBLK2(2)
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), NLSH_TMP
                     : NLSH_RIN, NLSH_QIN(0), NLSH_QIN(1), NLSH_QIN(2));
    else if (NQ == 2)
        asm volatile(NLSH_QBLK2(0) NLSH_QBLK2(1) : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), NLSH_TMP : NLSH_RIN, NLSH_QIN(0), NLSH_QIN(1));
    else
        asm volatile(NLSH_QBLK2(0) : [a0] "+v"(acc[0]), NLSH_TMP : NLSH_RIN, NLSH_QIN(0));
}
template <int NQ>
__device__ __forceinline__ void l2_tile_block(float (&acc)[4], const float4 r, const QSet &q) {
    float t0, t1, t2, t3;
    if (NQ == 4)
        asm volatile(NLSH_QBLK(0) NLSH_QBLK(1) NLSH_QBLK(2) NLSH_QBLK(3)
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), NLSH_TMP
                     : NLSH_RIN, NLSH_QIN(0), NLSH_QIN(1), NLSH_QIN(2), NLSH_QIN(3));
    else if (NQ == 3)
        asm volatile(NLSH_QBLK(0) NLSH_QBLK(1) NLSH_QBLK(2)
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), NLSH_TMP
                     : NLSH_RIN, NLSH_QIN(0), NLSH_QIN(1), NLSH_QIN(2));
    else if (NQ == 2)
        asm volatile(NLSH_QBLK(0) NLSH_QBLK(1) : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), NLSH_TMP : NLSH_RIN, NLSH_QIN(0), NLSH_QIN(1));
    else
        asm volatile(NLSH_QBLK(0) : [a0] "+v"(acc[0]), NLSH_TMP : NLSH_RIN, NLSH_QIN(0));
}
#undef NLSH_QBLK
#undef NLSH_QBLK2
#undef NLSH_QIN
#undef NLSH_TMP

// Cosine form of the block: acc += q_k * c_k in ascending k (the same fmaf chain as apply_qchunk), one VALU per element.  The
// query chunk is read from VGPR COPIES made once per chunk (`QCopy`, 4 v_mov per query) and used by all NT tile blocks of the chunk:
// a v_fmac that reads its multiplier from an SGPR issues at about half the rate of one that reads a VGPR (tools/probe_l2_block.hip),
// and here EVERY instruction would read one.
struct QCopy { float4 v[4]; };
#define NLSH_CBLK(J)                                                                                              \
    "v_fmac_f32 %[a" #J "], %[q" #J "0], %[r0]\n\tv_fmac_f32 %[a" #J "], %[q" #J "1], %[r1]\n\t"                     \
    "v_fmac_f32 %[a" #J "], %[q" #J "2], %[r2]\n\tv_fmac_f32 %[a" #J "], %[q" #J "3], %[r3]\n\t"
#if NLSH_COS_SGPR
#define NLSH_CIN(J) [q##J##0] "s"(q.v[J].x), [q##J##1] "s"(q.v[J].y), [q##J##2] "s"(q.v[J].z), [q##J##3] "s"(q.v[J].w)
#else
#define NLSH_CIN(J) [q##J##0] "v"(q.v[J].x), [q##J##1] "v"(q.v[J].y), [q##J##2] "v"(q.v[J].z), [q##J##3] "v"(q.v[J].w)
#endif
template <int NQ, typename QT>
__device__ __forceinline__ void cos_tile_block(float (&acc)[4], const float4 r, const QT &q) {
    if (NQ == 4)
        asm volatile(NLSH_CBLK(0) NLSH_CBLK(1) NLSH_CBLK(2) NLSH_CBLK(3)
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                     : NLSH_RIN, NLSH_CIN(0), NLSH_CIN(1), NLSH_CIN(2), NLSH_CIN(3));
    else if (NQ == 3)
        asm volatile(NLSH_CBLK(0) NLSH_CBLK(1) NLSH_CBLK(2)
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2])
                     : NLSH_RIN, NLSH_CIN(0), NLSH_CIN(1), NLSH_CIN(2));
    else if (NQ == 2)
        asm volatile(NLSH_CBLK(0) NLSH_CBLK(1) : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]) : NLSH_RIN, NLSH_CIN(0), NLSH_CIN(1));
    else
        asm volatile(NLSH_CBLK(0) : [a0] "+v"(acc[0]) : NLSH_RIN, NLSH_CIN(0));
}
template <int NQ>
__device__ __forceinline__ void copy_qset(QCopy &dst, const QSet &src) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) dst.v[j] = make_float4(src.v[j].x, src.v[j].y, src.v[j].z, src.v[j].w);
}
#undef NLSH_CBLK
#undef NLSH_CIN
#undef NLSH_RIN

// One k-block (nchunk 16-byte chunks of every row, LDS row stride RSt slots) for a wave that holds NQ queries, on NT
// 64-row tiles.  Blocks = (chunk, tile) pairs in chunk-major order; two chunks are unrolled so that the row-chunk
// registers (rr[0], rr[1]) and the query sets (qa, qb) alternate statically: block j reads rr[j & 1] while the row
// chunk of block j + 1 is on its way into rr[(j + 1) & 1], and the query chunk c + 1 is requested during the first block
// of chunk c.  The main loop has NO branch but its back edge: prefetches past the end of the k-block are clamped to its
// last chunk (valid addresses, values unused) instead of being guarded -- guarded, the loop carried 8 branches, 22
// scalar-ALU instructions and 25 s_nops per 384 VALU, and on this machine instruction issue is what the kernel is
// bound by (r02: kernel time tracks VALU x 2.3 + scalar x 2..4 cycles per SIMD across every variant measured).
template <int NQ, int NT, int METRIC = NLSH_METRIC_L2_EPS>
__device__ __forceinline__ void l2_kblock(const float4 *col, int RSt, int nchunk, const const_f32p (&qk)[4], float (&acc)[4][4]) {
    constexpr bool COS = METRIC == NLSH_METRIC_COSINE;
    QSet qa, qb;
    [[maybe_unused]] QCopy qv;   // cosine: VGPR copies of the current chunk's queries
    float4 rr[2];
    const int TS = 64 * RSt;   // tile stride in float4 slots
    const int last = nchunk - 1;
    rr[0] = col[0];
    load_qset<NQ>(qa, qk, 0, 0.0f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    int c = 0;
    for (; c + 1 < nchunk; c += 2) {
        const int c2 = min(c + 2, last);   // first chunk of the next pair (clamped on the last pair)
#pragma unroll
        for (int j = 0; j < 2 * NT; ++j) {
            const int tl = j % NT, cc = j / NT;             // compile-time after unrolling
            const int jn = j + 1, tn = jn % NT, cn = jn / NT;
            rr[jn & 1] = col[tn * TS + (cn == 2 ? c2 : c + cn)];
            if (tl == 0) {
                if (cc == 0) load_qset<NQ>(qb, qk, 16 * (c + 1), rr[j & 1].x);
                else load_qset<NQ>(qa, qk, 16 * c2, rr[j & 1].x);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (COS) {
                if (NLSH_COS_SGPR) {
                    cos_tile_block<NQ>(acc[tl], rr[j & 1], cc ? qb : qa);
                } else {
                    if (tl == 0) copy_qset<NQ>(qv, cc ? qb : qa);   // this chunk's set was waited for behind the previous chunk's last block
                    cos_tile_block<NQ>(acc[tl], rr[j & 1], qv);
                }
            } else if (METRIC == NLSH_METRIC_L2_EPS_FOLDED) {
                l2f_tile_block<NQ>(acc[tl], rr[j & 1], cc ? qb : qa);
            } else {
                l2_tile_block<NQ>(acc[tl], rr[j & 1], cc ? qb : qa);
            }
            if (tl == NT - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next chunk's queries (and first row chunk)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (c < nchunk) {   // odd chunk count (d / 4 not a multiple of the stage width): one more chunk, queries in qa, tile 0 in rr[0]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j + 1 < NT) rr[(j + 1) & 1] = col[(j + 1) * TS + c];
            __builtin_amdgcn_sched_barrier(0);
            if (COS) {
                if (NLSH_COS_SGPR) {
                    cos_tile_block<NQ>(acc[j], rr[j & 1], qa);
                } else {
                    if (j == 0) copy_qset<NQ>(qv, qa);
                    cos_tile_block<NQ>(acc[j], rr[j & 1], qv);
                }
            } else if (METRIC == NLSH_METRIC_L2_EPS_FOLDED) {
                l2f_tile_block<NQ>(acc[j], rr[j & 1], qa);
            } else {
                l2_tile_block<NQ>(acc[j], rr[j & 1], qa);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Scalar-cache warm-up of the query lines a k-block will read.  Every 64-byte line of a query is read by exactly one wave exactly once
// per task, so its first `s_load` always misses the scalar cache (SQC_DCACHE: 5.2 M requests, 1.08 M misses per launch = one per line) and
// the k-block's first wait -- directly behind the load -- sat through an L2 round trip eight times per task.  A throw-away one-dword load
// of each line, issued a stage earlier (the result register is never read), moves that round trip under the barriers and the LDS
// write of the stage in between.  gfx950 has no scalar prefetch instruction.
// `sink` is the destination of every throw-away load and MUST stay allocated until a `s_waitcnt lgkmcnt(0)` behind them (the loads
// complete asynchronously: a destination the compiler has already handed to another value is overwritten when they land -- the first
// version of this did exactly that and faulted).  It is threaded through the statements as a read-write operand and released by
// `warm_query_lines_done` after the wait.
template <int NQ>
__device__ __forceinline__ void warm_query_lines(const const_f32p (&qs)[4], int byte_off, int byte_end, float &sink) {
#if NLSH_WARM_QLINES
    for (int off = byte_off; off < byte_end; off += 64) {
#pragma unroll
        for (int jq = 0; jq < NQ; ++jq) asm volatile("s_load_dword %0, %1, %2" : "+s"(sink) : "s"(qs[jq]), "s"(off));
    }
#endif
}
__device__ __forceinline__ void warm_query_lines_done(float &sink) {
#if NLSH_WARM_QLINES
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sink)::"memory");
#endif
}

// All k-blocks of one L2 task for a wave that holds NQ (0..4) of its queries, NTL = tiles of the task (1..4): staging
// (global -> registers -> LDS, next k-block's loads in flight during the current one) + the hand-scheduled k-blocks.
// The (NQ, NTL) pair is chosen ONCE per task, outside the k-block loop: chosen per k-block, the 16 accumulators crossed
// a 16-way switch every k-block and the register allocator copied all of them in and out each time (466 v_mov in the
// kernel, +67 % instructions on small shapes).  NTL also fixes the fat-stage geometry at compile time.
template <int NW, int NQ, int NTL, int METRIC = NLSH_METRIC_L2_EPS>
__device__ __forceinline__ void l2_task(float4 *tile, const float4 *corpus4, long long stride4, int d4, int row0, int nrows,
                                        const const_f32p (&qs)[4], int tid, int lane, float (&acc)[4][4],
                                        [[maybe_unused]] unsigned long long (&tr)[3]) {
    constexpr int NTH = 64 * NW, KB = NLSH_TILED_KB;
    constexpr int kshift = NLSH_FAT_STAGES ? (NTL <= 1 ? 2 : (NTL == 2 ? 1 : 0)) : 0;
    constexpr int KBt = KB << kshift, RSt = KBt + 1, RPPt = (NTH / KB) >> kshift, SPT = 256 * KB / NTH;
    const int nkb = (d4 + KBt - 1) / KBt;
    const int sc = tid & (KBt - 1), sr = tid / KBt;   // staging map: KBt threads cover 16*KBt contiguous bytes of a row
    float4 stg[SPT];
    // Rows past the end of the segment and chunks past the end of a row are CLAMPED, not zero-filled: the clamped loads read valid
    // memory, rows >= nrows are masked at the epilogue (`valid`) and a k-block only evaluates its `nchunk` real chunks.  Guarded,
    // every staged word sat behind its own exec mask + branch + zero fill: ~22 VALU, 12 SALU and 4 branches per stage and wave.
    // (r04: a SECOND register set -- two k-blocks of a wave's rows in flight, for the waves that hold <= 0 / 1 / 2 / 4 of the task's
    // queries -- measured equal on all three workloads at 76 / 80 VGPRs and 3-6 % slower at 88: DESIGN.md appendix A.)
    const float4 *rowp[SPT];
#pragma unroll
    for (int i = 0; i < SPT; ++i) rowp[i] = corpus4 + (long long)(row0 + min(sr + RPPt * i, nrows - 1)) * stride4;
    auto stage_load = [&](int kb) {
        const int gc = min(kb * KBt + sc, d4 - 1);
#pragma unroll
        for (int i = 0; i < SPT; ++i) stg[i] = (NLSH_ABLATE != 2 && NLSH_ABLATE != 12 && NLSH_ABLATE != 13) ? rowp[i][gc] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    stage_load(0);
    float qsink = 0.0f;
    asm volatile("" : "+s"(qsink));
    if (NQ > 0) warm_query_lines<(NQ > 0 ? NQ : 1)>(qs, 0, min(KBt, d4) * 16, qsink);
    for (int kb = 0; kb < nkb; ++kb) {
        [[maybe_unused]] const unsigned long long ta = SCAN_NOW();
        NLSH_STAGE_SYNC();  // everyone has finished reading the previous k-block
        [[maybe_unused]] const unsigned long long tb = SCAN_NOW();
#pragma unroll
        for (int i = 0; i < SPT; ++i) tile[(sr + RPPt * i) * RSt + sc] = stg[i];
        NLSH_STAGE_SYNC();
        if (kb + 1 < nkb) stage_load(kb + 1);  // in flight while this k-block is computed
        // (r05: the queries' published bounds requested HERE in front of the last k-block, whose staging registers are free, instead of
        // behind it: the eight registers stay live across the 20 task bodies' joins -- 80 VGPRs, 6 waves per SIMD)
        [[maybe_unused]] const unsigned long long tc = SCAN_NOW();
        tr[0] += tb - ta;   // first barrier: the slowest wave's previous k-block
        tr[1] += tc - tb;   // own stage data (vmcnt) + LDS write + second barrier
        if (NQ > 0) warm_query_lines_done(qsink);   // behind the two barriers: the lines of this k-block are in the scalar cache
        if (NQ > 0 && NLSH_ABLATE != 1 && NLSH_ABLATE != 6 && NLSH_ABLATE != 13) {
            const int nchunk = min(KBt, d4 - kb * KBt);
            const_f32p qk[4];
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) qk[jq] = qs[jq] + kb * KBt * 4;
            if (NLSH_PRIO_MATH >= 0) __builtin_amdgcn_s_setprio(NLSH_PRIO_MATH);
            l2_kblock<(NQ > 0 ? NQ : 1), NTL, METRIC>(tile + lane * RSt, RSt, nchunk, qk, acc);
            if (NLSH_PRIO_OUT >= 0) __builtin_amdgcn_s_setprio(NLSH_PRIO_OUT);
            if (kb + 1 < nkb) warm_query_lines<(NQ > 0 ? NQ : 1)>(qs, (kb + 1) * KBt * 16, min((kb + 2) * KBt, d4) * 16, qsink);
#ifdef NLSH_SCAN_TRACE
            asm volatile("" : "+v"(acc[0][0]));
            tr[2] += SCAN_NOW() - tc;
#endif
        }
    }
    // No warm-up load is in flight here (the last k-block issues none: its range is empty), but only the loop bounds say so.
    // One wait makes it a property of the control-flow graph, which is what tools/isa_lint.py checks: the sink register is
    // released on EVERY path into the epilogue, whatever a future compiler makes of the loop.
    if (NQ > 0) warm_query_lines_done(qsink);
}

// Single-stage task (r05): a task whose rows x 16-byte chunks fit the workgroup's LDS stage at once (rows * d4 <= 1024 slots: <= 40 rows
// of a 100-d corpus, <= 32 of a 128-d one) is staged in ONE pass -- four loads per thread over the rows' contiguous bytes, one barrier --
// and scored by ONE k-block call over all d4 chunks.  The fat two-stage form it replaces for such tasks cut every row at byte 256: rows
// of 400 bytes are not aligned to the 128-byte lines of the L2, so nearly every line held bytes of both stages and was requested twice,
// a whole stage apart -- on the balanced workloads, whose scan is bound by the memory system (the load skeleton alone is 0.10 of GloVe's
// 0.12 ms), the counters saw 1.33x the bytes the task table accounts for (profiles/r05_traffic_tally.txt).  Here every line of the task
// is requested once, and the task has one exposed round trip less.  Same k-ascending fmaf chain per (row, query): same bits.
#ifndef NLSH_SINGLE_STAGE
#define NLSH_SINGLE_STAGE 1
#endif
constexpr int SINGLE_STAGE_SLOTS = 1024;   // float4 slots one pass of the 256 threads stages (4 each)
template <int NW, int NQ, int METRIC = NLSH_METRIC_L2_EPS>
__device__ __forceinline__ void l2_task_single(float4 *tile, const float4 *corpus4, long long stride4, int d4, int row0, int nrows,
                                               const const_f32p (&qs)[4], int tid, int lane, float (&acc)[4][4]) {
    constexpr int NTH = 64 * NW, SPT = SINGLE_STAGE_SLOTS / NTH;
    const int RS = d4 | 1;                   // odd LDS row stride (16-byte slots): conflict-free column reads
    const int total = nrows * d4;
    const float inv = 1.0f / (float)d4;
    // (r05: the pass started at the 128-byte line below the task's first byte, so that every wave-load covered whole lines of the L2: the
    // counters did not move -- 289.6 against 289.5 K FETCH_SIZE units per GloVe launch -- and the shift was removed; DESIGN.md appendix A.)
    float4 stg[SPT];
    int dst[SPT];
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
        const int idx = min(tid + NTH * i, total - 1);   // slots past the task's last chunk re-read it (valid address, value never written)
        int r = (int)((float)idx * inv);                  // idx / d4 without an integer division: off by at most one, put right below
        r -= (r * d4 > idx) ? 1 : 0;
        r += ((r + 1) * d4 <= idx) ? 1 : 0;
        const int c = idx - r * d4;
        stg[i] = (NLSH_ABLATE != 2 && NLSH_ABLATE != 12 && NLSH_ABLATE != 13) ? corpus4[(long long)(row0 + r) * stride4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
        dst[i] = r * RS + c;
    }
    float qsink = 0.0f;
    asm volatile("" : "+s"(qsink));
    if (NQ > 0) warm_query_lines<(NQ > 0 ? NQ : 1)>(qs, 0, d4 * 16, qsink);
    // one task per workgroup: nobody has read the tile before, so the writes need no barrier in front of them
#pragma unroll
    for (int i = 0; i < SPT; ++i)
        if (tid + NTH * i < total) tile[dst[i]] = stg[i];
    NLSH_STAGE_SYNC();
    if (NQ > 0) {
        warm_query_lines_done(qsink);
        if (NLSH_ABLATE != 1 && NLSH_ABLATE != 6 && NLSH_ABLATE != 13) {
            const_f32p qk[4];
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) qk[jq] = qs[jq];
            // lanes past the task's last row walk its last row (staged data; their results are masked at the epilogue)
            l2_kblock<(NQ > 0 ? NQ : 1), 1, METRIC>(tile + min(lane, nrows - 1) * RS, RS, d4, qk, acc);
        }
    }
}

template <int NW, int NQ, int METRIC = NLSH_METRIC_L2_EPS>
__device__ __forceinline__ void l2_task_nt(int ntile, float4 *tile, const float4 *corpus4, long long stride4, int d4, int row0, int nrows,
                                           const const_f32p (&qs)[4], int tid, int lane, float (&acc)[4][4], unsigned long long (&tr)[3]) {
    switch (ntile) {
        case 0: l2_task_single<NW, NQ, METRIC>(tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc); break;   // "0 tiles": the single-stage body
        case 1: l2_task<NW, NQ, 1, METRIC>(tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc, tr); break;
        case 2: l2_task<NW, NQ, 2, METRIC>(tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc, tr); break;
        case 3: l2_task<NW, NQ, 3, METRIC>(tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc, tr); break;
        default: l2_task<NW, NQ, 4, METRIC>(tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc, tr); break;
    }
}

// Merge of ONE query's partial lists into its final top-k (one wavefront; `sc` = 64 u64 of LDS scratch owned by the wave).
__device__ __forceinline__ void merge_query(const BArgs &a, long long q, int lane, uint64_t *sc, int2 *ltab) {
    // lane p holds probe p's record (written by bscatter): where its partial lists are.  bscatter writes a record for EVERY slot of
    // the [Q, P] table -- zeros for slots past the query's key count, for repeated keys and for keys without a bucket -- so the key
    // count is not needed here (r04: its load sat in front of the record load, one dependent round trip per wave)
    int ns_l = 0, j_l = 0, ng_l = 0;
    long long t0_l = 0;
    int size_l = 0;
    if (lane < a.P) {
        const int4 rec = a.prec[q * a.P + lane];
        t0_l = rec.x; j_l = rec.y; size_l = rec.z; ng_l = rec.w;
        ns_l = (size_l + a.seg - 1) / a.seg;
    }
    {   // n_candidates of the query (indexer.py:71,94) = rows of its probed buckets
        const int c = __builtin_amdgcn_readlane(wave_incl_scan_i32(size_l), 63);
        if (lane == 0) a.out_ncand[q] = c;
    }
    // The query's partial lists (one per probe and row segment) are numbered 0..L-1 by an inclusive scan of the
    // per-probe segment counts.  A round fetches 3 x R lists (R = 64/k per load instruction: lane -> (list, entry)) and
    // SELECTS the k best of them and the best so far (merge_round); typical queries (<= 18 lists at k = 10) take one round.
    const int incl = wave_incl_scan_i32(ns_l);
    const int L = __builtin_amdgcn_readlane(incl, 63);
    const int R = 64 / a.k;
    const int r = lane / a.k, e = lane - r * a.k;
    // r06: where list `li` lives -- (task, slot) -- comes from a table in LDS that lane p fills for its probe's ns_l lists (task of
    // segment si = first task + si * query groups), instead of a 6-step shuffle search + 6 more shuffles per fetched list: the merge is
    // bound by its instruction count (10^4 waves x ~10 per SIMD), and a typical query has 5-18 lists of 1-2 segments per probe.  Queries
    // with a probe of more than LIST_TAB_MAX_SEG segments (a giant bucket) or more than LIST_TAB lists keep the search.
    constexpr int LIST_TAB = 128, LIST_TAB_MAX_SEG = 8;
    const bool tabbed = ltab != nullptr && L <= LIST_TAB && (int)wave_minmax_u32<true>((uint32_t)ns_l) <= LIST_TAB_MAX_SEG;   // wave-uniform
    if (tabbed) {
        const int first = incl - ns_l;
        for (int si = 0; si < LIST_TAB_MAX_SEG; ++si)
            if (si < ns_l) ltab[first + si] = make_int2((int)(t0_l + (long long)si * ng_l), j_l);
    }
    // keys of list slot `li` (one list per group of k lanes): which probe it belongs to, which segment of that probe's bucket
    auto fetch = [&](int li) -> uint64_t {
        long long t;
        int j;
        if (tabbed) {
            const int2 ent = ltab[(r < R && li < L) ? li : 0];   // same wave wrote it: LDS operations of a wave complete in order
            t = ent.x; j = ent.y;
        } else {
            int lo = 0, hi = 63;  // probe of list li = first lane whose inclusive count exceeds li
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int mid = (lo + hi) >> 1;
                if (__shfl(incl, mid) > li) hi = mid; else lo = mid + 1;
            }
            const int p = lo > 63 ? 63 : lo;
            const int si = li - (__shfl(incl, p) - __shfl(ns_l, p));
            t = (long long)(((unsigned long long)(unsigned)__shfl((int)(t0_l >> 32), p) << 32) | (unsigned)__shfl((int)t0_l, p)) + (long long)si * __shfl(ng_l, p);
            j = __shfl(j_l, p);
        }
        // t >= max_tasks: table overflow, status[1] was set by the scan kernel and the caller repeats the call
        const bool live = r < R && li < L && t < a.max_tasks;
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.partial) + ((live ? t : 0) * a.QB + j) * a.k + e;
        return live ? (uint64_t)*src : KEY_NONE;
    };
    uint64_t carry = KEY_NONE;
    if (L <= R) {
        // r06: a query with at most R lists (GloVe-shaped: 6.5 probed buckets of a few rows each) selects from ONE key per lane -- a
        // quarter of the ballots per bisection step of the general round and one list lookup instead of three; <= 2R lists: two
        uint64_t key[1] = {fetch(r)};
        carry = merge_round<1>(key, a.k, lane, sc);
    } else if (L <= 2 * R) {
        uint64_t key[2] = {fetch(r), fetch(R + r)};
        carry = merge_round<2>(key, a.k, lane, sc);
    } else {
        for (int base = 0; base < L; base += 3 * R) {
            uint64_t key[4];
            key[0] = carry;
#pragma unroll
            for (int s = 0; s < 3; ++s) key[s + 1] = fetch(base + s * R + r);
            carry = merge_round<4>(key, a.k, lane, sc);
        }
    }
    merge_finish(carry, a.k, lane, a.out_dist, a.out_idx, a.out_keys, q);
}

// QW queries per wave, NW waves per workgroup (QW*NW queries per task), TPS 64-row tiles per task.
// k-blocks of KB chunks are the OUTER loop: one stage holds the KB-chunk slice of ALL 64*TPS rows of
// the segment in LDS, so every scalar-loaded query chunk is applied to TPS row tiles (TPS x fewer
// scalar loads and SALU per VALU than a tile-outer loop) and the accumulators of all tiles live in
// registers until the last k-block.  Chunks go through two scalar register sets: the s_loads of
// chunk c+1 are issued before chunk c is evaluated.
// One task of the tiled schedule, start to finish (operands of the task already requested by the caller: descriptor and
// the wave's query ids).  `tile` = the workgroup's LDS stage.
template <int METRIC, int QW, int NW, int TPS>
__device__ __forceinline__ void tiled_task_body(const BArgs &a, float4 *tile, long long t, const int4 desc, const int2 qr_all, int tid, int lane,
                                                int wave, [[maybe_unused]] unsigned long long ts_entry) {
    constexpr int NT = 64 * NW;              // threads per workgroup
    constexpr int KB = NLSH_TILED_KB;        // 16-byte chunks per k-block
    constexpr int ROWS = 64 * TPS;
    constexpr int SPT = ROWS * KB / NT;      // staged 16-byte words per thread and stage
    constexpr int RPP = NT / KB;             // rows covered by one pass of the workgroup
    [[maybe_unused]] const unsigned long long ts0 = SCAN_NOW();
    [[maybe_unused]] unsigned long long ts_stage = 0, ts_comp = 0;
#ifdef NLSH_SCAN_TRACE_CLOCK
    const unsigned long long core0 = __builtin_amdgcn_s_memtime();   // shader-clock counter beside the 100 MHz stamps: the clock held
#endif
    const int nq = __builtin_amdgcn_readfirstlane(desc.y);       // desc.x (first pair of the group) is the wave-level schedule's: the tiled tasks carry their query ids
    // The task's rows, narrowed to the hull of the rows its queries own: a 64-row window shared by several small buckets is staged and
    // scored from the first row of its first PROBED bucket to the last row of its last one (GloVe-1.2M: 1.27x -> 1.10x the rows of the
    // probed buckets, profiles/r05_traffic_tally.txt; the balanced workloads' scan is bound by the bytes it moves).  The hull is the
    // min / max over the task's <= 16 slot ranges (two wave-wide DPP reductions per task; a segment of a big bucket gives itself).
    // Same rows per query, same chains: same bits.
    const bool slot_live = (lane & (QW * NW - 1)) < nq;
    const int h_lo = (int)wave_minmax_u32<false>(slot_live ? (uint32_t)(qr_all.y & 0xFFFF) : 0xFFFFu);
    const int h_hi = (int)wave_minmax_u32<true>(slot_live ? (uint32_t)(qr_all.y >> 16) : 0u);
    const int row0 = __builtin_amdgcn_readfirstlane(desc.z) + h_lo;
    const int nrows = min(h_hi, __builtin_amdgcn_readfirstlane(desc.w)) - h_lo;  // <= ROWS (the host fixes seg = ROWS)
    if (nrows <= 0) return;   // wave-uniform, in front of every barrier: slot records a stale counter invented (workspace contract; bmerge flags it)
    // queries are dealt round-robin over the waves (slot = wave + NW*jq): a group of 5 queries costs the
    // workgroup 2 query-times per stage (2,1,1,1) instead of 4 (4,1,0,0); the stage barrier waits for the slowest wave
    int nqw = NLSH_DEAL_BLOCKS ? nq - wave * QW : (nq - wave + NW - 1) / NW;
    nqw = __builtin_amdgcn_readfirstlane(nqw < 0 ? 0 : (nqw > QW ? QW : nqw));

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const_f32p qs[QW];
    int qid[QW];
#pragma unroll
    for (int jq = 0; jq < QW; ++jq) {
        // ids clamped into [0, Q) so that a slot a stale counter invented (bmerge flags it) addresses nothing outside the queries
        qid[jq] = min(max(__builtin_amdgcn_readlane(qr_all.x, NLSH_SLOT(wave_u, jq)), 0), (int)a.Q - 1);
        qs[jq] = (const_f32p)(a.qpad + (long long)qid[jq] * a.qpad_stride);
    }

    const float4 *corpus4 = reinterpret_cast<const float4 *>(a.corpus);
    const long long stride4 = a.row_stride >> 2;
    const int d4 = a.d4p;
    const int ntile = (nrows + 63) >> 6;
    // A task costs ~15 us before it does any work (r01 trace: 16 us for 1 query x <= 64 rows, 55 us for 16 x 256): one
    // exposed global-load round trip + two barriers per k-block.  Short segments therefore take FATTER k-blocks --
    // the LDS tile holds 64*TPS rows x KB chunks = 64 rows x TPS*KB chunks: 1 tile -> 4*KB chunks per stage, 2 tiles ->
    // 2*KB -- and go through a quarter / half of the stages (a third of the tasks of the headline run are <= 128 rows).
    const int kshift = NLSH_FAT_STAGES ? (ntile <= 1 ? 2 : (ntile == 2 ? 1 : 0)) : 0;
    const int KBt = KB << kshift;            // chunks per k-block of THIS task
    const int RSt = KBt + 1;                 // odd row stride of its tile
    const int RPPt = RPP >> kshift;          // rows covered by one pass of the workgroup
    const int nkb = (d4 + KBt - 1) / KBt;
    // staging map: KBt threads cover 16*KBt contiguous bytes of a row
    const int sc = tid & (KBt - 1), sr = tid / KBt;
    float4 stg[SPT];
    // Rows past the end of the segment and chunks past the end of a row are CLAMPED, not zero-filled: the clamped loads read valid
    // memory, rows >= nrows are masked at the epilogue (`valid`) and a k-block only evaluates its `nchunk` real chunks.  Guarded,
    // every staged word sat behind its own exec mask + branch + zero fill: ~22 VALU, 12 SALU and 4 branches per stage and wave.
    const float4 *rowp[SPT];
#pragma unroll
    for (int i = 0; i < SPT; ++i) rowp[i] = corpus4 + (long long)(row0 + min(sr + RPPt * i, nrows - 1)) * stride4;
    auto stage_load = [&](int kb) {
        const int gc = min(kb * KBt + sc, d4 - 1);
#pragma unroll
        for (int i = 0; i < SPT; ++i) stg[i] = (NLSH_ABLATE != 2 && NLSH_ABLATE != 12 && NLSH_ABLATE != 13) ? rowp[i][gc] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float acc[TPS][QW];
#pragma unroll
    for (int tl = 0; tl < TPS; ++tl)
#pragma unroll
        for (int jq = 0; jq < QW; ++jq) acc[tl][jq] = 0.0f;

    // global row ids (and cosine norms) of the rows this lane owns: requested up front, consumed by the epilogue --
    // issued there, the load was an exposed round trip at the end of every task (r02 trace: ~2 us of a 28-us task)
    int32_t mygid[TPS];
    float myinv[TPS];
    bool valid[TPS];
#pragma unroll
    for (int tl = 0; tl < TPS; ++tl) {
        valid[tl] = tl * 64 + lane < nrows;
        const int prow = row0 + (valid[tl] ? tl * 64 + lane : 0);
        mygid[tl] = valid[tl] ? a.gid[prow] : -1;
        myinv[tl] = (METRIC == NLSH_METRIC_COSINE && valid[tl]) ? a.inv_norm[prow] : 0.0f;
    }
    constexpr bool FAST = NLSH_FAST_KBLOCK && (METRIC != NLSH_METRIC_COSINE || NLSH_FAST_COSINE) && QW == 4 && TPS == 4;
    [[maybe_unused]] unsigned long long trl[3] = {0, 0, 0};
    [[maybe_unused]] const unsigned long long ts_in = SCAN_NOW();
    // a task whose rows x chunks fit one stage takes the single-stage body (l2_task_single): selected as "0 tiles" of the same switch
    const int nt_sel = (NLSH_SINGLE_STAGE && nrows * d4 <= SINGLE_STAGE_SLOTS && nrows <= 64) ? 0 : ntile;   // wave-uniform (task shape)
    if (FAST) {   // hand-scheduled form, specialised per (queries of this wave, tiles of the task); same barrier count on every path
        switch (nqw) {
            case 0: l2_task_nt<NW, 0, METRIC>(nt_sel, tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc, trl); break;
            case 1: l2_task_nt<NW, 1, METRIC>(nt_sel, tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc, trl); break;
            case 2: l2_task_nt<NW, 2, METRIC>(nt_sel, tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc, trl); break;
            case 3: l2_task_nt<NW, 3, METRIC>(nt_sel, tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc, trl); break;
            default: l2_task_nt<NW, 4, METRIC>(nt_sel, tile, corpus4, stride4, d4, row0, nrows, qs, tid, lane, acc, trl); break;
        }
    }
    if (!FAST) stage_load(0);
    [[maybe_unused]] const unsigned long long ts1 = SCAN_NOW();
    for (int kb = 0; !FAST && kb < nkb; ++kb) {
        const unsigned long long ta = SCAN_NOW();
        __syncthreads();  // everyone has finished reading the previous k-block
#pragma unroll
        for (int i = 0; i < SPT; ++i) tile[(sr + RPPt * i) * RSt + sc] = stg[i];
        __syncthreads();
        if (kb + 1 < nkb) stage_load(kb + 1);  // in flight while this k-block is computed
        const unsigned long long tb = SCAN_NOW();
        ts_stage += tb - ta;
        if (NLSH_ABLATE != 1 && nqw > 0) {
            const int nchunk = min(KBt, d4 - kb * KBt);
            const_f32p qk[QW];
#pragma unroll
            for (int jq = 0; jq < QW; ++jq) qk[jq] = qs[jq] + kb * KBt * 4;
            const float4 *col = tile + lane * RSt;
            QChunk<QW> qa, qb;
            load_qchunk<QW, false>(qa, qk, nqw, 0);
            for (int c = 0; c < nchunk; c += 2) {
                const bool has1 = c + 1 < nchunk;
                load_qchunk<QW, false>(qb, qk, nqw, has1 ? c + 1 : c);
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl)
                    if (tl < ntile) apply_qchunk<METRIC, QW, false>(qa, col[tl * 64 * RSt + c], nqw, acc[tl]);
                if (!has1) break;
                load_qchunk<QW, false>(qa, qk, nqw, c + 2 < nchunk ? c + 2 : c);
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl)
                    if (tl < ntile) apply_qchunk<METRIC, QW, false>(qb, col[tl * 64 * RSt + c + 1], nqw, acc[tl]);
            }
        }
#ifdef NLSH_SCAN_TRACE
        {   // the accumulators must exist before the stamp: make the stamp depend on one of them
            float sink = 0.f;
#pragma unroll
            for (int tl = 0; tl < TPS; ++tl)
#pragma unroll
                for (int jq = 0; jq < QW; ++jq) sink += acc[tl][jq];
            asm volatile("" ::"v"(sink));
            ts_comp += SCAN_NOW() - tb;
        }
#endif
    }
    [[maybe_unused]] const unsigned long long ts2 = SCAN_NOW();
    if (nqw == 0) return;
    if (NLSH_ABLATE == 5 || NLSH_ABLATE == 6 || NLSH_ABLATE == 12 || NLSH_ABLATE == 13) {   // diagnostic: no epilogue at all (the accumulators are kept alive)
#pragma unroll
        for (int tl = 0; tl < TPS; ++tl)
#pragma unroll
            for (int jq = 0; jq < QW; ++jq) asm volatile("" ::"v"(acc[tl][jq]));
        return;
    }
    // lane = row of each tile -> one candidate per lane, tile and query
    // Lists of the same query in other tasks publish their k-th best key to tauq[q] (atomicMin): no
    // candidate above it can reach the final top-k, so it pre-filters this list (fewer insertions).
    // Which partial entries survive depends on timing; the merged result does not.
    // All <= 64*TPS candidates of a list exist at once here (TPS keys per lane), so the k best are
    // SELECTED (bisection + compaction, select_k_smallest) instead of inserted one by one.
    [[maybe_unused]] const unsigned long long ts3 = SCAN_NOW();
    // the running bounds of the wave's queries are requested together (one exposed round trip, not one per query)
    uint64_t tau_w[QW];
#pragma unroll
    for (int jq = 0; jq < QW; ++jq) tau_w[jq] = jq < nqw ? global_tau_load(a.tauq + qid[jq]) : KEY_NONE;
    constexpr bool LEAN = NLSH_LEAN_EPILOGUE && METRIC != NLSH_METRIC_COSINE;
#pragma unroll
    for (int jq = 0; jq < QW; ++jq) {
        if (jq < nqw) {
            const uint64_t tau_g = tau_w[jq];
            // rows of the task that belong to THIS query's bucket: all of them for a segment of a big bucket, the bucket's slice of a
            // window shared by several small buckets (the other rows were scored for nothing: the arithmetic of a shared window is what
            // a task of its own would have cost each of those buckets in fixed latency)
            const int rng = __builtin_amdgcn_readlane(qr_all.y, NLSH_SLOT(wave_u, jq));
            const unsigned r_lo = (unsigned)((rng & 0xFFFF) - h_lo), r_n = (unsigned)(rng >> 16) - (unsigned)(rng & 0xFFFF);   // relative to the first row staged
            uint64_t key[TPS];
            // LEAN: every accumulator of the list at or above 2^-96 (wave-uniform test; NaN compares false and takes the general path)
            // -> square roots without the range scaling, and a non-negative distance's order-preserving word is its bits with the sign set
            bool lean = LEAN;
            if (LEAN) {   // tiles the task does not have hold zeros and lanes past its last row another row's (or nobody's) sums: neither is asked
                bool ok = true;
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl) ok = ok && (tl >= ntile || !valid[tl] || acc[tl][jq] >= 0x1p-96f);   // a NaN fails the comparison
                lean = __ballot(!ok) == 0ull;
            }
#pragma unroll
            for (int tl = 0; tl < TPS; ++tl) {
                const bool mine = valid[tl] && (unsigned)(tl * 64 + lane) - r_lo < r_n;
                uint64_t kk;
                if (lean) {
                    const float dist = sqrt_rn_unscaled(acc[tl][jq]);
                    kk = ((uint64_t)(__builtin_bit_cast(uint32_t, dist) | 0x80000000u) << 32) | (uint32_t)mygid[tl];   // == make_key for dist >= +0
                } else {
                    kk = make_key(finish_distance<METRIC>(acc[tl][jq], myinv[tl]), mygid[tl]);
                }
                kk = mine ? kk : KEY_NONE;
                key[tl] = kk < tau_g ? kk : KEY_NONE;  // beyond another list's k-th best: cannot reach the final top-k
            }
            uint64_t *out = a.partial + ((long long)t * (QW * NW) + NLSH_SLOT(wave, jq)) * a.k;
#ifdef NLSH_SCAN_TRACE_EPILOGUE
            {   // diagnostic: how many (task, query) lists reach the selection with a published bound, and with how many survivors
                int n_all = 0, n_live = 0;
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl) { n_all += __popcll(__ballot(valid[tl])); n_live += __popcll(__ballot(key[tl] != KEY_NONE)); }
                if (lane == 0) {
                    float *c = g_scan_trace + (NLSH_TRACE_SLOTS - 1) * 8;
                    atomicAdd(c + 0, 1.0f);
                    if (tau_g != KEY_NONE) atomicAdd(c + 1, 1.0f);
                    if (n_live == 0) atomicAdd(c + 2, 1.0f);
                    else if (n_live < a.k) atomicAdd(c + 3, 1.0f);
                    atomicAdd(c + 4, (float)n_all);
                    atomicAdd(c + 5, (float)n_live);
                }
            }
#endif
            if (NLSH_ABLATE != 3) {
                // (r04: one-tile tasks selecting from ONE key per lane instead of TPS with three absent -- a quarter of the ballots per
                // bisection step -- measured equal on all three workloads, profiles/r04_select_nk1_ab.txt; not kept)
                const uint64_t bound = select_k_smallest<TPS>(key, a.k, lane, out);
                if (bound != KEY_NONE && lane == 0) atomicMin(a.tauq + qid[jq], (unsigned long long)bound);
            } else if (lane < a.k) out[lane] = key[0];
        }
    }
#ifdef NLSH_SCAN_TRACE
    if (tid == 0 && t < NLSH_TRACE_SLOTS) {
        const unsigned long long ts4 = SCAN_NOW();
        float *o = g_scan_trace + t * 8;
        o[0] = (float)(ts4 - ts_entry); o[1] = FAST ? (float)(ts_in - ts_entry) : (float)(ts1 - ts0); o[2] = FAST ? (float)trl[0] : (float)ts_stage;
        o[3] = FAST ? (float)trl[2] : (float)ts_comp;
        o[4] = FAST ? (float)trl[1] : 0.f; o[5] = (float)(ts4 - ts3); o[6] = (float)(nq * 1000 + nrows); o[7] = (float)(ts_entry & 0xFFFFFFull);
#ifdef NLSH_SCAN_TRACE_HWID   // (r05's placement analysis: overwrites the barrier-1 and compute columns)
        // where it ran: HW_ID (wave/simd/cu/sh/se) and XCC_ID, as exact small integers
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        o[2] = (float)(((xcc & 0xF) << 12) | (((hw >> 13) & 0x7) << 9) | (((hw >> 12) & 0x1) << 8) | (((hw >> 8) & 0xF) << 4) | (((hw >> 4) & 0x3) << 2));
        o[3] = (float)(hw & 0xF);
#endif
#ifdef NLSH_SCAN_TRACE_CLOCK
        o[4] = (float)(__builtin_amdgcn_s_memtime() - core0); o[1] = (float)(ts4 - ts0);   // core cycles and 100 MHz ticks of the same interval
#endif
    }
#endif
}


template <int METRIC, int QW, int NW, int TPS>
__global__ __launch_bounds__(64 * NW, NLSH_TILED_MIN_WAVES) void bscan3_kernel(BArgs a) {
    constexpr int KB = NLSH_TILED_KB;        // 16-byte chunks per k-block
    constexpr int RS = KB + 1;               // odd LDS row stride (16-byte slots) -> conflict-free column reads
    constexpr int ROWS = 64 * TPS;
    __shared__ float4 tile[ROWS * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    [[maybe_unused]] const unsigned long long ts_entry = SCAN_NOW();
    if (NLSH_PRIO_OUT >= 0) __builtin_amdgcn_s_setprio(NLSH_PRIO_OUT);
    long long ntasks = a.status[0];
    if (ntasks > a.max_tasks) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&a.status[1], 1);  // incomplete: caller must retry (a refusal of the PLAN phase -- 2, 3 -- stays)
        ntasks = a.max_tasks;
    }
    // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Task ids are dealt to the XCDs
    // in CHUNKS of 16 consecutive ids: the query groups of one row segment (consecutive ids) mostly land on one
    // XCD and re-read its rows from that L2 instead of HBM, while every XCD still walks the size-ordered task
    // list front to back (a contiguous 1/8 range per XCD would hand all the heavy tasks to XCD 0).
    // Placement only changes speed, never results.
    constexpr int XC = 16;
    const long long j = blockIdx.x >> 3;
    const long long t = ((j / XC) * 8 + (blockIdx.x & 7)) * XC + (j % XC);
    // the descriptor is requested BEFORE the task count is known (index clamped into the table): one dependent round trip
    // less in front of every task
    const long long tc = t < a.max_tasks ? t : a.max_tasks - 1;
    const int4 desc = a.task[tc];
    // all 16 {query id, row range} records of the task in ONE load, one record per lane (& 15) -- address known from the task id alone,
    // requested with the descriptor.  A wave picks its own slots out of it with v_readlane (r04: four 8-byte loads per wave), and the
    // hull of the rows the task's queries own at all is taken over all sixteen (r05).  Slots >= nq hold garbage, never used.
    const int2 qr_all = a.task_qr[tc * (QW * NW) + (lane & (QW * NW - 1))];
    // (r06: cross-task prefetch into the XCD's L2 -- a finished workgroup touching the rows of the task 1792-3072 ids ahead on its XCD with
    // loads nobody waits for -- was built and is NOT here: every run of it ended in a GPU exception (a wave may not end with its loads in
    // flight on this part), and with the wait in front of s_endpgm the workgroup's slot is held for exactly the HBM miss the prefetch was
    // meant to hide.  DESIGN.md appendix A.)
    if (t >= ntasks) return;
    if (NLSH_ABLATE == 9) return;   // diagnostic: every workgroup leaves after its descriptor loads (what dispatching the grid costs)
    if (NLSH_ABLATE == 8 && desc.y <= NLSH_ABLATE_NQ) return;   // diagnostic: tasks with few queries vanish (what the low-density tasks cost)
    if (NLSH_ABLATE == 7 && desc.w <= 64) return;   // diagnostic: tasks of <= 64 rows vanish (what a kernel without the tail of tiny tasks would take)
    tiled_task_body<METRIC, QW, NW, TPS>(a, tile, t, desc, qr_all, tid, lane, wave, ts_entry);
}

// (r01-r05 re-checked here that every pair counter was back at zero; since r06 bscan_kernel resets the counters itself and its
// verdict covers every way a stale count can enter a batch -- positive ones through the sum, negative ones directly.)
__global__ __launch_bounds__(256) void bmerge_kernel(BArgs a) {
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    __shared__ uint64_t scratch[4][64];
    __shared__ int2 list_tab[4][128];    // merge_query's LIST_TAB entries per wave
    if (q < a.Q) merge_query(a, q, lane, scratch[threadIdx.x >> 6], list_tab[threadIdx.x >> 6]);
}

#ifndef NLSH_TILED_QB
#define NLSH_TILED_QB 16
#endif
constexpr int TILED_QB = NLSH_TILED_QB;  // queries per task of the tiled schedule (4 per wave)
#ifndef NLSH_TILED_TPS
#define NLSH_TILED_TPS 4
#endif
constexpr int TILED_TPS = NLSH_TILED_TPS;  // 64-row tiles per task of the tiled schedule (segment = 64*TPS rows)
// a shared window (nlsh_build_cells: window_rows <= 256) must fit ONE segment: bscatter derives a window bucket's partial-list count from
// its own size and clamps its row range to the segment, so a window wider than a segment would silently lose rows
static_assert(64 * TILED_TPS >= 256, "the tiled schedule's segment must hold the widest row window nlsh_build_cells accepts (256 rows)");

struct BWs {
    size_t ppair, prec, inv_q, bcount, cellrec, lookback, hits, task, task_qr, partial, qpad, tauq, total;
};
// blocks of the launch that does the lookup: bplan_kernel's 256 pairs per block, or encode_hash's workgroups (>= 16 rows each)
static inline long long plan_blocks_max(long long Q, int P) {
    const long long a = (Q * P + 255) / 256, b = (Q + 15) / 16;
    return (a > b ? a : b) + 1;
}
static void blayout(long long Q, int P, int k, long long max_tasks, long long nb, int d, bool tiled, BWs *w) {
    size_t o = 0;
    const size_t qp = (size_t)Q * P * 4, nb4 = (size_t)(nb > 0 ? nb : 1) * 4;
    w->bcount = o;   o += ws_align(nb4);   // first, at an offset that does not depend on the batch: ZERO between calls (workspace contract)
    w->ppair = o;    o += ws_align(qp * 4);
    w->prec = o;     o += ws_align(qp * 4);
    w->inv_q = o;    o += ws_align(qp);
    w->cellrec = o;  o += ws_align(nb4 * 4);
    w->lookback = o; o += ws_align((size_t)((nb + 255) / 256 + 1) * 8);
    w->hits = o;     o += ws_align((size_t)plan_blocks_max(Q, P) * 4);
    // everything above sits at offsets that do not depend on max_tasks: a lookup done for one table size (encode_hash's epilogue)
    // stays valid when the call is repeated with a larger table
    w->task = o;     o += ws_align((size_t)max_tasks * sizeof(int4));
    w->task_qr = o;  o += tiled ? ws_align((size_t)max_tasks * TILED_QB * 8) : 0;
    w->partial = o;  o += ws_align((size_t)max_tasks * (tiled ? TILED_QB : 8) * k * 8);
    w->qpad = o;     o += tiled ? ws_align((size_t)Q * ((d + 3) / 4) * 16) : 0;
    w->tauq = o;     o += ws_align((size_t)Q * 8);
    w->total = o;
}

size_t bucket_scan_workspace(long long Q, int P, int k, long long max_tasks, long long n_buckets, int d) {
    BWs w, wt;
    blayout(Q, P, k, max_tasks, n_buckets, d, false, &w);
    blayout(Q, P, k, max_tasks, n_buckets, d, true, &wt);
    return w.total > wt.total ? w.total : wt.total;
}

}  // namespace nlsh

extern "C" int nlsh_scan_workspace_layout(int64_t Q, int P, int k, int64_t max_tasks, int64_t n_buckets, int d, int algo,
                                          size_t *task_table_offset, size_t *task_queries_offset, size_t *task_ranges_offset) {
    NLSH_REQUIRE(algo == NLSH_SCAN_BUCKET_MAJOR || algo == NLSH_SCAN_BUCKET_TILED, NLSH_E_INVALID, "scan_workspace_layout: algo=%d has no task table of this form", algo);
    NLSH_REQUIRE(Q >= 0 && P >= 1 && k >= 1 && max_tasks >= 0 && n_buckets >= 0 && d >= 1, NLSH_E_INVALID, "scan_workspace_layout: bad sizes");
    nlsh::BWs w;
    nlsh::blayout(Q, P, k, max_tasks, n_buckets, d, algo == NLSH_SCAN_BUCKET_TILED, &w);
    if (task_table_offset) *task_table_offset = w.task;
    if (task_queries_offset) *task_queries_offset = w.task_qr;        // interleaved: {query id, range} pairs, 8 bytes per (task, slot)
    if (task_ranges_offset) *task_ranges_offset = w.task_qr + 4;
    return NLSH_OK;
}

namespace nlsh {

template <int METRIC>
static const void *bscan2_of(int d4) {
    if (d4 <= 16) return (const void *)bscan2_kernel<16, 1, METRIC, 8>;
    if (d4 <= 32) return (const void *)bscan2_kernel<32, 1, METRIC, 8>;
    if (d4 <= 64) return (const void *)bscan2_kernel<64, 1, METRIC, 8>;
    if (d4 <= 128) return (const void *)bscan2_kernel<64, 2, METRIC, 4>;
    return (const void *)bscan2_kernel<64, 4, METRIC, 2>;
}

// the scan kernel of a call and its launch shape
static void scan_kernel_of(const BucketScanCall &c, int metric, int d4, const void **fn, dim3 *grid, dim3 *block) {
    if (c.tiled) {
        // QW = 4 queries per wave (SGPR budget: two chunks x QW x 4 scalar values in flight), NW = 4 waves
        // one workgroup per task; the chunked XCD map works on 8 x 16 ids.  (Persistent workgroups pulling tasks from a
        // per-XCD queue were measured three ways in r02 -- 0.447 / 0.423 / 0.350 ms against 0.283 ms; r04: 2 / 4 / 8 consecutive
        // task ids per workgroup, 79 VGPRs, +3 / +10 / +25 % on the headline and no better on the small-bucket workloads:
        // profiles/r04_tasks_per_workgroup_ab.txt, DESIGN.md appendix A.)
        *grid = dim3((unsigned)((c.max_tasks + 127) / 128 * 128));
        *block = dim3(64 * (TILED_QB / 4));
        if (metric == NLSH_METRIC_L2_EPS) *fn = (const void *)bscan3_kernel<NLSH_METRIC_L2_EPS, 4, TILED_QB / 4, TILED_TPS>;
        else if (metric == NLSH_METRIC_L2_EPS_FOLDED) *fn = (const void *)bscan3_kernel<NLSH_METRIC_L2_EPS_FOLDED, 4, TILED_QB / 4, TILED_TPS>;
        else *fn = (const void *)bscan3_kernel<NLSH_METRIC_COSINE, 4, TILED_QB / 4, TILED_TPS>;
    } else {
        *grid = dim3((unsigned)((c.max_tasks + 3) / 4));  // one wavefront per task
        *block = dim3(256);
        *fn = metric == NLSH_METRIC_L2_EPS ? bscan2_of<NLSH_METRIC_L2_EPS>(d4) : bscan2_of<NLSH_METRIC_COSINE>(d4);
    }
}

// BArgs (what the scan kernels take) and PlanArgs (what the lookup takes) of one call: pointers into the caller's workspace.
static int bucket_scan_args(const BucketScanCall &c, BArgs &a, PlanArgs &pa, int &metric_out, bool &prep_out) {
    BWs w;
    blayout(c.Q, c.P, c.k, c.max_tasks, c.nb, c.d, c.tiled != 0, &w);
    NLSH_REQUIRE(c.workspace_bytes >= w.total, NLSH_E_WORKSPACE, "scan_topk(bucket-major): workspace %zu < %zu", c.workspace_bytes, w.total);
    const int d4 = (c.d + 3) / 4;
    a.corpus = c.corpus; a.row_stride = c.row_stride; a.d = c.d; a.gid = c.gid; a.uniq = c.uniq; a.offsets = c.offsets; a.nb = c.nb;
    // cells exist for the tiled schedule only (a window is one 256-row segment of its task shape); the wave-level schedule ignores them
    const bool cells = c.tiled && c.cell_of && c.cell_offsets && c.n_cells > 0;
    a.cell_of = cells ? c.cell_of : nullptr; a.coffsets = cells ? c.cell_offsets : c.offsets; a.nc = cells ? c.n_cells : c.nb;
    a.inv_norm = c.inv_norm; a.queries = c.queries; a.q_stride = c.q_stride; a.Q = c.Q; a.qkeys = c.qkeys; a.nkeys = c.nkeys;
    a.P = c.P; a.k = c.k; a.seg = c.tiled ? 64 * TILED_TPS : c.seg; a.QB = c.tiled ? TILED_QB : (d4 <= 64 ? 8 : (d4 <= 128 ? 4 : 2));
    a.qpad_w = (float *)((char *)c.workspace + w.qpad); a.qpad = a.qpad_w; a.qpad_stride = (long long)d4 * 4; a.d4p = d4;
    // L2 with d % 4 == 0 needs neither padding nor normalisation: read the caller's queries directly
    // the folded L2 form exists in the tiled schedule only (it always needs the prepared query copy: q + eps); the other two
    // schedules answer NLSH_METRIC_L2_EPS_FOLDED with the exact form, which is inside the same tolerance
    const int metric = (!c.tiled && c.metric == NLSH_METRIC_L2_EPS_FOLDED) ? NLSH_METRIC_L2_EPS : c.metric;
    const bool prep = c.tiled && !(metric == NLSH_METRIC_L2_EPS && (c.d & 3) == 0 && (c.q_stride & 3) == 0 && ((uintptr_t)c.queries & 15) == 0);
    if (c.tiled && !prep) { a.qpad = c.queries; a.qpad_stride = c.q_stride; }
    a.out_dist = c.out_dist; a.out_idx = c.out_idx; a.out_keys = c.out_keys; a.out_ncand = c.out_ncand; a.status = c.status;
    char *base = (char *)c.workspace;
    a.ppair = (int4 *)(base + w.ppair); a.prec = (int4 *)(base + w.prec); a.inv_q = (int32_t *)(base + w.inv_q);
    a.bcount = (int32_t *)(base + w.bcount); a.cellrec = (int4 *)(base + w.cellrec); a.lookback = (unsigned long long *)(base + w.lookback);
    a.hits = (int32_t *)(base + w.hits); a.border = c.bucket_order; a.task = (int4 *)(base + w.task);
    a.task_qr = c.tiled ? (int2 *)(base + w.task_qr) : nullptr; a.partial = (uint64_t *)(base + w.partial);
    a.max_tasks = c.max_tasks;
    a.tauq = (unsigned long long *)(base + w.tauq);

    pa.uniq = a.uniq; pa.offsets = a.offsets; pa.cell_of = a.cell_of; pa.coffsets = a.coffsets; pa.nb = a.nb;
    pa.stride = 1; pa.nco = a.nb;   // the launcher of the lookup sizes the coarse table for the LDS it has (plan_coarse)
    pa.seg = a.seg; pa.P = a.P; pa.Q = a.Q; pa.qkeys = a.qkeys; pa.qnkeys = a.nkeys;
    pa.bcount = a.bcount; pa.ppair = a.ppair; pa.hits = a.hits; pa.status = a.status; pa.tauq = a.tauq;
    pa.lookback = a.lookback; pa.n_lookback = (a.nc + 255) / 256;
    pa.queries = c.queries; pa.q_stride = c.q_stride; pa.qpad = a.qpad_w; pa.qpad_stride = (long long)d4 * 4; pa.d = c.d; pa.d4p = d4;
    pa.prep_metric = prep ? metric : -1;
    pa.enabled = 1;
    metric_out = metric;
    prep_out = prep;
    return NLSH_OK;
}

int bucket_scan_node(const BucketScanCall &c, ScanNode *out) {
    static_assert(sizeof(BArgs) <= sizeof(out->args) && alignof(BArgs) <= 16, "ScanNode::args must hold the scan kernels' argument struct");
    NLSH_REQUIRE(c.max_tasks > 0, NLSH_E_INVALID, "scan node: empty task table");
    BArgs *a = new (out->args) BArgs;
    PlanArgs pa;
    int metric;
    bool prep;
    const int rc = bucket_scan_args(c, *a, pa, metric, prep);
    if (rc != NLSH_OK) return rc;
    const void *fn;
    out->p = hipKernelNodeParams{};
    scan_kernel_of(c, metric, (c.d + 3) / 4, &fn, &out->p.gridDim, &out->p.blockDim);
    out->p.func = const_cast<void *>(fn);
    out->argv[0] = out->args;
    out->p.kernelParams = out->argv; out->p.sharedMemBytes = 0; out->p.extra = nullptr;
    return NLSH_OK;
}

int bucket_scan_plan_args(const BucketScanCall &c, PlanArgs *pa) {
    BArgs a;
    int metric;
    bool prep;
    return bucket_scan_args(c, a, *pa, metric, prep);
}

int bucket_scan_run(const BucketScanCall &c) {
    BArgs a;
    PlanArgs pa;
    int metric;
    bool prep;
    const int rc = bucket_scan_args(c, a, pa, metric, prep);
    if (rc != NLSH_OK) return rc;
    const int d4 = (c.d + 3) / 4;

    hipStream_t s = c.stream;
    const unsigned gp = (unsigned)((c.Q * c.P + 255) / 256);
    if (c.phases & NLSH_PHASE_PLAN) {
        // The lookup as a launch of its own (caller-supplied keys; keys made by encode_hash are looked up in ITS epilogue and the
        // caller passes NLSH_PHASE_PLAN_REST instead).  (One fused launch with grid barriers between the steps of this counting
        // sort was built and measured in r02: 127 us with agent-scope fences -- each writes back / invalidates an XCD's L2 --, 63 us
        // with device-coherent sc1 accesses instead, against 35 us for the separate launches: crossing XCDs inside a kernel costs as
        // much as a kernel boundary on this part.)  The per-cell pair counters need no clearing launch: bscan_kernel hands every
        // count back, so they are zero again after every call (workspace contract, nlsh_hip.h).
        plan_coarse(pa, 1024);
        const unsigned gprep = prep ? (unsigned)((c.Q + 3) / 4) : 0u;   // the query copy rides in the same launch
        hipLaunchKernelGGL(bplan_kernel, dim3(gp + gprep), dim3(256), 0, s, pa, gp);
    }
    if (c.phases & (NLSH_PHASE_PLAN | NLSH_PHASE_PLAN_REST)) {
        // (r03: bcount + bscan as ONE single-workgroup launch for indexes of <= 8192 buckets took 23 us against 16.4 us for the two
        // launches of r02-r05: one CU writes 12 k task descriptors slower than 23 workgroups do.  r06: one launch of all the
        // workgroups with a look-back over their totals, see bscan_kernel.)
        const int plan_blocks = (c.phases & NLSH_PHASE_PLAN) ? (int)gp : c.plan_blocks;
        if (a.nc > 0) hipLaunchKernelGGL(bscan_kernel, dim3((unsigned)((a.nc + 255) / 256)), dim3(256), 0, s, a, plan_blocks);
        hipLaunchKernelGGL(bscatter_kernel, dim3(gp), dim3(256), 0, s, a);
    }
    if ((c.phases & NLSH_PHASE_SCAN) && c.max_tasks > 0) {
        if (c.ev_begin) NLSH_CHECK_HIP(hipEventRecord((hipEvent_t)c.ev_begin, s));
        const void *fn;
        dim3 grid, block;
        scan_kernel_of(c, metric, d4, &fn, &grid, &block);
        void *argv[1] = {&a};
        NLSH_CHECK_HIP(hipLaunchKernel(fn, grid, block, argv, 0, s));
        if (c.ev_end) NLSH_CHECK_HIP(hipEventRecord((hipEvent_t)c.ev_end, s));
    }
    if (c.phases & NLSH_PHASE_MERGE) hipLaunchKernelGGL(bmerge_kernel, dim3((unsigned)((c.Q + 3) / 4)), dim3(256), 0, s, a);
    NLSH_CHECK_HIP(hipGetLastError());
    return NLSH_OK;
}

}  // namespace nlsh
