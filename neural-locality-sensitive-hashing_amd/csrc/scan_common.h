// Wave-level building blocks shared by the query-major and bucket-major scan kernels.
#pragma once
#include "common.h"

// Diagnostic timing builds (`make VARIANT=diag EXTRA="-DNLSH_DIAG -DNLSH_ABLATE=n"`, tools/scan_bench.py): every switch below
// produces WRONG results by design (it removes a piece of the kernel to time the rest) and exists only under NLSH_DIAG, so that a
// stray -D on the shipped build cannot turn one on.  NLSH_ABLATE: 1 no distance math, 2 no global loads, 3 no top-k selection,
// 4 no scalar loads (generic loop), 5 no epilogue, 6 neither math nor epilogue (staging skeleton), 7 tasks of <= 64 rows vanish,
// 8 tasks of <= NLSH_ABLATE_NQ queries vanish, 9 every workgroup leaves after its descriptor loads, 11 no bisection in the
// selection, 12 = 2 + 5 (no global loads, no epilogue), 13 = 2 + 6 (barriers and LDS writes only).  NLSH_NO_STAGE_BARRIER=1: the stage barriers of the hand-scheduled task body are removed (the four waves of a
// workgroup race on the tile): what the barriers' straggler coupling costs (r03: 3-5 %).
// (Two r03 experiments lived here behind switches and were deleted in r04, measured slower and never shipped: LDS-DMA staging of the
// k-blocks, NLSH_TILED_GLDS, and the per-query merge inside the scan launch, NLSH_MERGE_IN_SCAN.  DESIGN.md appendix A and
// profiles/r03_glds_ab.txt / r03_merge_in_scan_ab.txt keep the findings; git history keeps the code.)
#ifdef NLSH_DIAG
#ifndef NLSH_ABLATE
#define NLSH_ABLATE 0
#endif
#ifndef NLSH_NO_STAGE_BARRIER
#define NLSH_NO_STAGE_BARRIER 0
#endif
#else
#if defined(NLSH_ABLATE) || defined(NLSH_NO_STAGE_BARRIER)
#error "NLSH_ABLATE / NLSH_NO_STAGE_BARRIER produce wrong results by design: diagnostic builds only (add -DNLSH_DIAG)"
#endif
#define NLSH_ABLATE 0
#define NLSH_NO_STAGE_BARRIER 0
#endif

#ifndef NLSH_SELECT_SHORT_PATHS
#define NLSH_SELECT_SHORT_PATHS 1   // select_k_smallest: lists with no / fewer than k present keys skip the cut search and its compaction (0: r04's single path, for A/B; same results)
#endif

namespace nlsh {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

// sum over aligned groups of LPR lanes; every lane of a group receives the same bits
template <int LPR>
__device__ __forceinline__ float group_sum(float x) {
    x += dpp_f<0xB1>(x);                    // quad_perm [1,0,3,2]
    x += dpp_f<0x4E>(x);                    // quad_perm [2,3,0,1]
    x += dpp_f<0x141>(x);                   // row_half_mirror
    x += dpp_f<0x140>(x);                   // row_mirror -> 16-lane sums
    // gfx950 v_permlane16_swap / v_permlane32_swap exchange 16-lane rows / 32-lane halves between
    // TWO registers (semantics probed on hardware: tools/probe_permlane.hip): with a = b = x,
    // a' = {x0,x0,x2,x2}, b' = {x1,x1,x3,x3}, so a' + b' is the cross-row sum in every lane with no
    // LDS round trip.  Written as asm: through the builtin hipcc (ROCm 7.2) folded the second result
    // into the first when both inputs carry the same value and emitted a' + a'.  The s_nop covers
    // the VALU-write -> permlane-read wait states (hipcc pads nothing inside asm).
    if (LPR >= 32) {
        float a = x, b = x;
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        x = a + b;
    }
    if (LPR >= 64) {
        float a = x, b = x;
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        x = a + b;
    }
    return x;
}

// 64-bit value of lane `src` (wave-uniform index) as a wave-uniform value: two v_readlane_b32,
// no LDS round trip (a generic __shfl compiles to ds_bpermute).
__device__ __forceinline__ uint64_t read_lane64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// value of lane-1 (lane 0 keeps its own): DPP wave_shr:1, two v_mov_b32_dpp
__device__ __forceinline__ uint64_t lane_shift_up64(uint64_t v) {
    const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    const uint32_t slo = (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xF, 0xF, false);
    const uint32_t shi = (uint32_t)__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xF, 0xF, false);
    return ((uint64_t)shi << 32) | slo;
}

// best-64 list, sorted ascending across lanes; insert one wave-uniform key
__device__ __forceinline__ void topk_insert(uint64_t &top, uint64_t c, int lane) {
    const int posn = __popcll(__ballot(top < c));
    const uint64_t up = lane_shift_up64(top);
    top = lane < posn ? top : (lane == posn ? c : up);
}

// offer one key per lane; only keys below the current k-th best (tau) are inserted
__device__ __forceinline__ void topk_offer(uint64_t &top, uint64_t &tau, uint64_t key, int k, int lane) {
    unsigned long long m = __ballot(key < tau);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint64_t c = read_lane64(key, src);
        if (c < tau) {
            topk_insert(top, c, lane);
            const uint64_t kth = read_lane64(top, k - 1);
            tau = kth < tau ? kth : tau;  // never rises: tau may start from a bound published by another list
        }
    }
}

// Per-query running bound shared by all lists of a query (agent-scope relaxed atomics: sc1 load / atomic umin).
// tau semantics in topk_offer: a key is inserted iff key < tau, so the published bound is (k-th best key + 1).
__device__ __forceinline__ uint64_t global_tau_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void global_tau_publish(unsigned long long *p, uint64_t top, int k, int lane) {
    const uint64_t kth = read_lane64(top, k - 1);
    if (kth != KEY_NONE && lane == 0) atomicMin(p, (unsigned long long)kth + 1ull);
}

// wave-wide min / max of a u32 (every lane gets the result): DPP row operations, no LDS, no scalar ALU
template <bool MAX>
__device__ __forceinline__ uint32_t wave_minmax_u32(uint32_t x) {
    auto op = [](uint32_t a, uint32_t b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
    const uint32_t id = MAX ? 0u : 0xFFFFFFFFu;
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)x, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x141, 0xF, 0xF, false));  // row_half_mirror
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x140, 0xF, 0xF, false));  // row_mirror -> 16-lane result
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x142, 0xA, 0xF, false));  // row_bcast15 into rows 1 and 3
    x = op(x, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x143, 0xC, 0xF, false));  // row_bcast31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// wave-wide inclusive prefix sum of an i32 (lane l gets x_0 + ... + x_l): four row_shr steps inside each 16-lane row (zero fill), then the
// row totals carried across with row_bcast15 / row_bcast31 -- six DPP adds, no LDS crossbar (`__shfl_up` is a ds_bpermute per step)
__device__ __forceinline__ int wave_incl_scan_i32(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);    // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);    // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);    // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);    // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast15: lane 15 of rows 0 / 2 into rows 1 / 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast31: lane 31 into rows 2 and 3
    return x;
}

// k smallest of the wave's NK*64 keys (NK per lane, KEY_NONE = absent), written UNSORTED to out[0..k)
// (KEY_NONE padded), without any ordered insertion.  A bisection on the distance word looks for ANY threshold that
// separates exactly k keys (ballots + scalar popcounts): it starts from the [min, max] bracket of the present
// distances (two DPP reductions) and stops as soon as a pivot has exactly k keys at or below it -- with 256 keys
// spread over the bracket that takes ~9 steps instead of the 32 a search for the exact k-th value needs.  This matters
// because a step is ~17 scalar-ALU instructions and a CU has ONE scalar unit for its four SIMDs (r02 ablation: with the
// 32-step form the selection was HALF of the tiled scan's time, 95 M of its SALU instructions per launch).  Exact
// distance ties at the k-th place (no pivot separates k keys) fall back to the exact search plus a second bisection
// on the id word.  Survivors are compacted with mbcnt prefix counts.  Returns an exclusive upper bound of this list's
// k-th best key to publish for the query (every key at or above it is beyond the list's k best) or KEY_NONE when the
// list holds fewer than k keys.
// CLAMP (the merges): keys that reach a merge are distinct when every (query, bucket) pair was scanned once -- the plan
// kernels de-duplicate a query's probe keys, corpus shards are disjoint -- but nlsh_merge_topk takes whatever lists a C
// caller hands it: with repeated keys the tie search can select more than k, so the compaction never writes past out[k).
template <int NK, bool CLAMP = false>
__device__ __forceinline__ uint64_t select_k_smallest(const uint64_t (&key)[NK], int k, int lane, uint64_t *out) {
    uint32_t hi[NK], lo[NK];
    int n = 0;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        hi[i] = (uint32_t)(key[i] >> 32);
        lo[i] = (uint32_t)key[i];
        n += __popcll(__ballot(key[i] != KEY_NONE));
    }
    // r05: the two short cases first.  18-36 % of the tiled scan's lists reach this point with NO key below the query's published bound and
    // 23-41 % with fewer than k (profiles/r05_epilogue_counters_and_ablations.txt): neither needs a cut, and the general compaction below
    // spends six compares per key slot re-deriving "present" from a cut that takes everything.
    if (NLSH_SELECT_SHORT_PATHS && n == 0) {
        if (lane < k) out[lane] = KEY_NONE;
        return KEY_NONE;
    }
    if (NLSH_SELECT_SHORT_PATHS && n < k) {
        int base0 = 0;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const bool sel = key[i] != KEY_NONE;
            const unsigned long long m = __ballot(sel);
            if (m) {   // wave-uniform: an absent tile costs one scalar compare
                const int pos = base0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (sel) out[pos] = key[i];
                base0 += __popcll(m);
            }
        }
        if (lane >= base0 && lane < k) out[lane] = KEY_NONE;
        return KEY_NONE;
    }
    uint32_t dk = 0xFFFFFFFFu, idk = 0xFFFFFFFFu;  // take everything present (hi of a present key < 0xFFFFFFFF)
    if (n >= k) {
        uint32_t mn = hi[0], mx = hi[0] == 0xFFFFFFFFu ? 0u : hi[0];
#pragma unroll
        for (int i = 1; i < NK; ++i) {
            mn = hi[i] < mn ? hi[i] : mn;
            const uint32_t h = hi[i] == 0xFFFFFFFFu ? 0u : hi[i];
            mx = h > mx ? h : mx;
        }
        uint32_t a = wave_minmax_u32<false>(mn), b = wave_minmax_u32<true>(mx);
        // invariants: #{hi <= b} >= k (b = max: all n > k keys) and #{hi < a} < k (a = min: none)
        bool split = false;
        if (n == k) { a = b; split = true; }   // all present keys are taken: the cut is the largest of them
#if NLSH_ABLATE == 11   // diagnostic timing build (WRONG results): no bisection at all -- what the search for the cut costs
        a = b; split = true;
#endif
        while (a < b) {
            const uint32_t mid = a + ((b - a) >> 1);
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < NK; ++i) cnt += __popcll(__ballot(hi[i] <= mid));
            if (cnt == k) { a = mid; split = true; break; }   // exactly k keys at or below the pivot: done
            if (cnt > k) b = mid; else a = mid + 1;
        }
        dk = a;
        if (!split) {   // a == b = the k-th distance, shared by keys on both sides of the cut (or n has exactly k at/below max)
            int c_less = 0, c_eq = 0;
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                c_less += __popcll(__ballot(hi[i] < dk));
                c_eq += __popcll(__ballot(hi[i] == dk));
            }
            const int need = k - c_less;  // >= 1 keys to take among those at distance d_k
            if (c_eq > need) {            // exact distance ties: smallest ids win
                a = 0; b = 0xFFFFFFFFu;
                while (a < b) {
                    const uint32_t mid = a + ((b - a) >> 1);
                    int cnt = 0;
#pragma unroll
                    for (int i = 0; i < NK; ++i) cnt += __popcll(__ballot(hi[i] == dk && lo[i] <= mid));
                    if (cnt >= need) b = mid; else a = mid + 1;
                }
                idk = a;
            }
        }
    }
    int base = 0;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const bool sel = key[i] != KEY_NONE && (hi[i] < dk || (hi[i] == dk && lo[i] <= idk));
        const unsigned long long m = __ballot(sel);
        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#if NLSH_ABLATE == 11
        if (sel && pos < k) {
#else
        if (sel && (!CLAMP || pos < k)) {
#endif
            out[pos] = key[i];
        }
        base += __popcll(m);
    }
    if (lane >= base && lane < k) out[lane] = KEY_NONE;
    return n >= k ? ((uint64_t)dk + 1ull) << 32 : KEY_NONE;
}

// Merging many short lists by SELECTION instead of ordered insertion: a wave keeps its current best k keys unsorted in
// lanes 0..k-1 (`carry`), takes NK-1 new keys per lane and round, lets select_k_smallest pick the k best of all NK*64 and
// reads them back from a per-wave LDS scratch (k keys, compacted: real keys first, KEY_NONE after).  The ordered
// insertion it replaces (topk_offer) cost ~100 instructions per inserted key and the first list of every query inserts
// all of its k keys.  merge_finish ranks the k survivors (k uniform compare steps) and stores them in ascending order.
template <int NK>
__device__ __forceinline__ uint64_t merge_round(uint64_t (&key)[NK], int k, int lane, uint64_t *scratch) {
    select_k_smallest<NK, true>(key, k, lane, scratch);
    return lane < k ? scratch[lane] : KEY_NONE;   // same wave wrote it: LDS operations of a wave complete in order
}

__device__ __forceinline__ void merge_finish(uint64_t carry, int k, int lane, float *out_dist, int32_t *out_idx, uint64_t *out_keys, long long q) {
    int rank = 0;
    for (int j = 0; j < k; ++j) {   // keys are distinct ((distance, id) with distinct ids), KEY_NONE sits behind the real ones;
        const uint64_t kj = read_lane64(carry, j);   // equal keys (a C caller's repeated lists) are ordered by lane: every slot is written
        rank += (kj < carry || (kj == carry && j < lane)) ? 1 : 0;
    }
    if (lane < k) {
        const bool none = carry == KEY_NONE;
        const int pos = none ? lane : rank;
        out_dist[q * k + pos] = none ? __builtin_inff() : float_from_mono((uint32_t)(carry >> 32));
        out_idx[q * k + pos] = none ? -1 : (int32_t)(uint32_t)carry;
        if (out_keys) out_keys[q * k + pos] = carry;
    }
}

__device__ __forceinline__ void store_topk(float *out_dist, int32_t *out_idx, uint64_t *out_keys, long long q, int k,
                                           uint64_t top, int lane) {
    if (lane < k) {
        const bool none = top == KEY_NONE;
        out_dist[q * k + lane] = none ? __builtin_inff() : float_from_mono((uint32_t)(top >> 32));
        out_idx[q * k + lane] = none ? -1 : (int32_t)(uint32_t)top;
        if (out_keys) out_keys[q * k + lane] = top;
    }
}

// Query fragment of one lane: columns 4*(li + v*LPR) .. +3, padded so that padding contributes 0
// (L2: (q - 0) + eps == 0 with q = -eps; cosine: 0).  Cosine fragments are pre-divided by
// max(||q||, 1e-8) like cosine_similarity does with x1.
template <int LPR, int VPL, int METRIC>
__device__ __forceinline__ void load_query(const float *qp, int d, int li, float4 (&qv)[VPL], bool (&act)[VPL]) {
    const float padv = METRIC == NLSH_METRIC_L2_EPS ? -1e-6f : 0.0f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int c0 = 4 * (li + v * LPR);
        act[v] = c0 < d;
        qv[v].x = c0 + 0 < d ? qp[c0 + 0] : padv;
        qv[v].y = c0 + 1 < d ? qp[c0 + 1] : padv;
        qv[v].z = c0 + 2 < d ? qp[c0 + 2] : padv;
        qv[v].w = c0 + 3 < d ? qp[c0 + 3] : padv;
    }
    if (METRIC == NLSH_METRIC_COSINE) {
        float ss = 0.0f;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
            if (act[v]) { ss = fmaf(qv[v].x, qv[v].x, ss); ss = fmaf(qv[v].y, qv[v].y, ss); ss = fmaf(qv[v].z, qv[v].z, ss); ss = fmaf(qv[v].w, qv[v].w, ss); }
        ss = group_sum<LPR>(ss);
        const float nrm = fmaxf(sqrtf(ss), 1e-8f);
#pragma unroll
        for (int v = 0; v < VPL; ++v) { qv[v].x /= nrm; qv[v].y /= nrm; qv[v].z /= nrm; qv[v].w /= nrm; }
    }
}

// lane-partial of one row against one query fragment
template <int VPL, int METRIC>
__device__ __forceinline__ float row_partial(const float4 (&qv)[VPL], const bool (&act)[VPL], const float4 (&cv)[VPL]) {
    float sacc = 0.0f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        if (METRIC == NLSH_METRIC_L2_EPS) {
            // F.pairwise_distance: || (x1 - x2) + eps ||  (nlsh/data.py:201)
            float t0 = (qv[v].x - cv[v].x) + 1e-6f, t1 = (qv[v].y - cv[v].y) + 1e-6f;
            float t2 = (qv[v].z - cv[v].z) + 1e-6f, t3 = (qv[v].w - cv[v].w) + 1e-6f;
            float part = fmaf(t3, t3, fmaf(t2, t2, fmaf(t1, t1, t0 * t0)));
            sacc += act[v] ? part : 0.0f;
        } else {
            sacc += fmaf(qv[v].w, cv[v].w, fmaf(qv[v].z, cv[v].z, fmaf(qv[v].y, cv[v].y, qv[v].x * cv[v].x)));
        }
    }
    return sacc;
}

// Correctly rounded sqrtf for x >= 2^-96 (and for +0, +inf and NaN, which come out as sqrtf gives them): the core of the sequence hipcc
// emits for sqrtf -- v_sqrt_f32 (1 ulp), then the two one-ulp neighbours are tried against the residual x - s'*s computed by ONE fma each
// -- without the x * 2^32 / * 2^-16 range scaling it wraps around that core for arguments below 2^-96 and without the class test that
// passes zeros and infinities through (27 VALU per call as compiled, 11 here).  The tiled scan takes 16 square roots per lane and task;
// callers hold the wave's arguments against 2^-96 first and fall back to sqrtf when any lies below (never on real distances: the smallest
// is sqrt(d) * 1e-6 for a query that IS a corpus row).
__device__ __forceinline__ float sqrt_rn_unscaled(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) - 1u), sp = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) + 1u);
    const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
    float r = rm <= 0.0f ? sm : s;
    r = rp > 0.0f ? sp : r;
    return r;
}

template <int METRIC>
__device__ __forceinline__ float finish_distance(float acc, float inv_norm) {
    if (METRIC != NLSH_METRIC_COSINE) return sqrtf(acc);   // both L2 forms
    return 1.0f - acc * inv_norm;  // 1 - cos (nlsh/data.py:109)
}

static inline size_t ws_align(size_t x) { return (x + 255) & ~(size_t)255; }

// entry point of the bucket-major algorithm (scan_bucket.hip)
struct BucketScanCall {
    const float *corpus; long long row_stride; int d; const int32_t *gid; const int32_t *uniq; const int32_t *offsets; int nb;
    const float *inv_norm; const float *queries; long long q_stride; long long Q; const int32_t *qkeys; const int32_t *nkeys;
    int P, k, metric, seg; float *out_dist; int32_t *out_idx; uint64_t *out_keys; int32_t *out_ncand; int32_t *status;
    void *workspace; size_t workspace_bytes; long long max_tasks; void *ev_begin; void *ev_end; hipStream_t stream; int tiled;
    const int32_t *bucket_order;  // nlsh_bucket_order output (nlsh_build_cells' cell_order when cells are given) or nullptr
    int phases;                   // NLSH_PHASE_PLAN | NLSH_PHASE_SCAN | NLSH_PHASE_MERGE
    const int32_t *cell_of;       // nlsh_build_cells outputs, or nullptr / 0: every bucket is its own cell
    const int32_t *cell_offsets;
    int n_cells;
    int plan_blocks;              // NLSH_PHASE_PLAN_REST: workgroups of the encode_hash launch that did the lookup (entries of `hits`)
};
// internal phase bit (not part of the C ABI's phase mask): the PLAN phase WITHOUT the bucket lookup, which the batch's encode_hash
// launch already did in its epilogue (scan_plan.h, encode_plan_fuse_lookup): bscan + bscatter only
#define NLSH_PHASE_PLAN_REST 8
struct PlanArgs;
size_t bucket_scan_workspace(long long Q, int P, int k, long long max_tasks, long long n_buckets, int d);
int bucket_scan_run(const BucketScanCall &c);
int bucket_scan_plan_args(const BucketScanCall &c, PlanArgs *pa);   // the lookup's arguments for this call's workspace

// nlsh_scan_topk_cells_phase after argument validation, with the internal phase bit allowed; `call_out` (nullable) receives the
// bucket-major call descriptor instead of running it (step.hip builds the fused lookup's arguments from it)
int scan_topk_cells_phase_checked(const float *corpus_sorted, int64_t row_stride, int d, const int32_t *gid, const int32_t *uniq_keys,
                                  const int32_t *offsets, const int32_t *bucket_order, int32_t n_buckets, const int32_t *cell_of,
                                  const int32_t *cell_offsets, int32_t n_cells, const float *inv_norm, const float *queries, int64_t q_stride,
                                  int64_t Q, const int32_t *qkeys, const int32_t *nkeys, int P, int k, int metric, int algo, int seg_rows,
                                  float *out_dist, int32_t *out_idx, uint64_t *out_keys, int32_t *out_ncand, int32_t *status, void *workspace,
                                  size_t workspace_bytes, int64_t max_tasks, void *ev_scan_begin, void *ev_scan_end, nlsh_stream_t stream,
                                  int phases, int plan_blocks, BucketScanCall *call_out);

}  // namespace nlsh
