// The bucket lookup of the bucket-major PLAN phase, shared by its two homes (r06): `bplan_kernel` (scan_bucket.hip; caller-supplied key
// tables) and the epilogue of `encode_hash_kernel` (encode_hash.hip; the keys are in the workgroup's LDS there, so the lookup rides in
// the launch that made them and the batch has one dependent launch less).  Replaces, per (query, probe) pair, the reference's
// `self.index2row.get(key, empty)` (nlsh/indexer.py:68) and the bookkeeping of its `for key in index_keys` loop (:66-83).
//
// What a pair leaves behind: ONE 16-byte record {cell, slot in the cell's pair list, rows of its bucket, first row of the bucket inside
// the cell} -- the slot is the value `atomicAdd` returned on the cell's pair counter, so the scatter step (bscatter_kernel) needs no
// atomic of its own and no second lookup -- and the cell's counter incremented.  The counters are read AND reset by bscan_kernel
// (thread per cell), which is how they are zero again after every batch (workspace contract, include/nlsh_hip.h).
#pragma once
#include "common.h"

namespace nlsh {

struct PlanArgs {
    const int32_t *uniq;       // [nb] bucket keys ascending
    const int32_t *offsets;    // [nb + 1]
    const int32_t *cell_of;    // [nb] bucket -> cell, or nullptr (cell == bucket)
    const int32_t *coffsets;   // [nc + 1] first sorted row of each cell (== offsets without cells)
    int nb;
    int stride, nco;           // coarse table = every stride-th key, nco entries (in the workgroup's LDS)
    int seg;                   // rows of a task segment (a shared window must fit one)
    int P;                     // probes per query = columns of the key table
    long long Q;
    const int32_t *qkeys;      // [Q, P] key table and [Q] valid slots per row (bplan_kernel only: encode_hash has them in LDS)
    const int32_t *qnkeys;
    int32_t *bcount;           // [nc] pair counters (head of the workspace)
    int4 *ppair;               // [Q * P] the pair records; .x < 0: no bucket
    int32_t *hits;             // [blocks] pairs each block counted | PLAN_VIOL_* flags
    int32_t *status;           // [2], zeroed here
    unsigned long long *tauq;  // [Q] running bound of each query, KEY_NONE-initialised here
    unsigned long long *lookback;   // [n_lookback] per-block aggregates of bscan_kernel's look-back, zeroed here
    int n_lookback;
    // the tiled schedule's padded / pre-normalised query copy (prep_metric < 0: not needed)
    const float *queries;
    long long q_stride;
    float *qpad;
    long long qpad_stride;
    int d, d4p, prep_metric;
    int enabled;               // encode_hash: 0 = plain encode (index builds, hash())
};

// coarse table of the lookup: every stride-th key, at most `cap` entries (what the launching kernel has LDS for)
inline void plan_coarse(PlanArgs &pa, int cap) {
    int stride = 1;
    while ((long long)stride * cap < pa.nb) stride <<= 1;
    pa.stride = stride;
    pa.nco = (pa.nb + stride - 1) / stride;
}

// a block's entry of `hits`: pairs counted in the low 24 bits, violations above them (the status words are initialised by the same
// launch, so a violation cannot be written there without a race: bscan_kernel's last block turns the flags into status[1])
constexpr int PLAN_HITS_MASK = (1 << 24) - 1;
constexpr int PLAN_VIOL_COUNTER = 1 << 24;   // a pair counter was negative on entry (workspace contract) -> status[1] = 2
constexpr int PLAN_VIOL_CELLS = 1 << 25;     // a shared window wider than one segment (cells not from nlsh_build_cells) -> status[1] = 3

// Padded / pre-normalised copy of one query for the tiled schedule (one wavefront per query): L2 pads with -eps ((q - 0) + eps == 0 on
// padding), the folded L2 form stores q + eps, cosine stores x1 / max(||x1||, 1e-8) as cosine_similarity does (nlsh/data.py:109).
__device__ __forceinline__ void prep_query(const PlanArgs &a, long long q, int lane) {
    const float *qp = a.queries + q * a.q_stride;
    float *dst = a.qpad + q * a.qpad_stride;
    const int n = a.d4p * 4;
    if (a.prep_metric == NLSH_METRIC_L2_EPS) {
        for (int e = lane; e < n; e += 64) dst[e] = e < a.d ? qp[e] : -1e-6f;  // (q - 0) + eps == 0 on padding
    } else if (a.prep_metric == NLSH_METRIC_L2_EPS_FOLDED) {
        for (int e = lane; e < n; e += 64) dst[e] = e < a.d ? qp[e] + 1e-6f : 0.0f;   // eps folded into the query: (q + eps) - c; 0 - 0 on padding
    } else {
        float ss = 0.0f;
        for (int e = lane; e < a.d; e += 64) ss = fmaf(qp[e], qp[e], ss);
        for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
        const float nrm = fmaxf(sqrtf(ss), 1e-8f);  // x1 / max(||x1||, eps), as cosine_similarity does
        for (int e = lane; e < n; e += 64) dst[e] = e < a.d ? qp[e] / nrm : 0.0f;
    }
}

// Per-batch initialisation that rides in the plan launch (no launch of its own): the status words and bscan_kernel's look-back slots.
// Called by every thread of block 0.
__device__ __forceinline__ void plan_batch_init(const PlanArgs &a, int tid, int nthreads) {
    if (tid == 0) { a.status[0] = 0; a.status[1] = 0; }
    for (int i = tid; i < a.n_lookback; i += nthreads) a.lookback[i] = 0ull;
}

// One (query, probe) pair: binary search of `key` in uniq[nb] -- the first steps on the coarse table in LDS (every stride-th key),
// the last ones on the stride-long run in global memory: 1-5 dependent global round trips instead of 13-17 -- then the bucket's
// rows, its cell, and the slot the pair takes in the cell's list.  `live` = the slot holds a key at all (p < nkeys[q], not a repeat).
// Returns true when the pair was counted; `viol` collects PLAN_VIOL_* flags.
__device__ __forceinline__ bool plan_pair(const PlanArgs &a, const int32_t *coarse, long long idx, int32_t key, bool live, int &viol) {
    int4 rec = make_int4(-1, 0, 0, 0);
    bool hit = false;
    if (live) {
        int lo = 0, hi = a.nco;
        while (lo < hi) {  // first coarse entry > key
            const int mid = (lo + hi) >> 1;
            if (coarse[mid] <= key) lo = mid + 1; else hi = mid;
        }
        if (lo > 0) {  // the key, if present, lies in the run that starts at coarse entry lo-1
            lo = (lo - 1) * a.stride;
            hi = min(a.nb, lo + a.stride);
            // r06: the run is narrowed over its BLOCKS of 8 keys (first key of each) and the last block's 8 keys are requested
            // together -- one dependent round trip for the final three bisection steps and the equality test behind them
            // (headline index: stride 8, so the whole fine search is ONE round trip instead of four).  The lookup sits in the tail of
            // a latency-bound workgroup (encode_hash's epilogue): dependent round trips are what it costs.
            int blo = lo >> 3, bhi = (hi + 7) >> 3;     // runs start at multiples of the stride, a power of two >= 8 or the whole table
            while (bhi - blo > 16) {                     // (runs of more than 128 keys: indexes of more than 131,072 buckets) last block whose first key is <= key
                const int mid = (blo + bhi) >> 1;
                if (a.uniq[mid << 3] <= key) blo = mid; else bhi = mid;
            }
            if (bhi - blo > 1) {
                // the first keys of the run's <= 16 blocks, requested together: ONE round trip where the bisection took up to four
                // (GloVe-shaped: 104 k buckets, runs of 128 keys -- the lookup's seven dependent round trips were 8 us of a 51-us launch)
                int32_t head[16];
#pragma unroll
                for (int j = 1; j < 16; ++j) head[j] = a.uniq[min(blo + j, bhi - 1) << 3];
                int adv = 0;
#pragma unroll
                for (int j = 1; j < 16; ++j) adv += (blo + j < bhi && head[j] <= key) ? 1 : 0;    // heads ascend: the count IS the block
                blo += adv;
            }
            const int base = blo << 3;
            int32_t kk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) kk[j] = a.uniq[min(base + j, a.nb - 1)];
            lo = a.nb;
#pragma unroll
            for (int j = 7; j >= 0; --j) lo = (kk[j] == key && base + j < a.nb) ? base + j : lo;
            if (lo < a.nb) {  // unknown key = empty bucket (indexer.py:61,68)
                const int row0 = a.offsets[lo], size = a.offsets[lo + 1] - row0;
                if (size > 0) {
                    const int c = a.cell_of ? a.cell_of[lo] : lo;
                    const int c0 = a.coffsets[c], crows = a.coffsets[c + 1] - c0;
                    // A bucket that is only PART of its cell lives in a shared window, and a shared window must fit one segment
                    // (nlsh_build_cells: window_rows <= 256 = the segment): the scatter step takes the bucket's partial-list count
                    // from its own size and clamps its row range to the segment.  Cells from elsewhere that break the rule are
                    // refused (status[1] = 3), never scanned short.
                    if (a.cell_of && size < crows && crows > a.seg) {
                        viol |= PLAN_VIOL_CELLS;
                    } else {
                        const int rel = atomicAdd(&a.bcount[c], 1);   // slot of this query in the cell's pair list
                        if (rel < 0) {
                            viol |= PLAN_VIOL_COUNTER;   // the counter started below zero: a stale count (workspace contract); no slot exists
                        } else {
                            rec = make_int4(c, rel, size, row0 - c0);
                            hit = true;
                        }
                    }
                }
            }
        }
    }
    a.ppair[idx] = rec;
    return hit;
}

}  // namespace nlsh
