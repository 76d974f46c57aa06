// Index build: keys[n] -> CSR (perm, uniq_keys, offsets) and the bucket-contiguous corpus copy.
//
// Replaces build_index (nlsh/indexer.py:6-24: Python dict of row lists + one tiny H2D per
// bucket) and, once per index instead of once per (query, key), the index_select gather of
// nlsh/indexer.py:77-82.  The stable LSD radix sort of (key, row) pairs is rocPRIM's (a native
// ROCm primitive; this step is SURVEY.md §8(f) row N1, outside the graded scan/encode kernels);
// bucket heads, offsets and the row permutation/gather are hand-written.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace nlsh {

__global__ void iota_kernel(int32_t *v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        v[i] = (int32_t)i;
}

__global__ void head_flags_kernel(const int32_t *sk, long long n, int32_t *flags) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        flags[i] = (i == 0 || sk[i] != sk[i - 1]) ? 1 : 0;
}

// rank[i] = inclusive prefix sum of flags -> bucket index + 1 of row i (sorted order).
__global__ void emit_buckets_kernel(const int32_t *sk, const int32_t *flags, const int32_t *rank, long long n,
                                    int32_t *uniq, int32_t *offsets, int32_t *n_buckets) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (flags[i]) {
            int b = rank[i] - 1;
            uniq[b] = sk[i];
            offsets[b] = (int32_t)i;
        }
        if (i == n - 1) {
            int nb = rank[i];
            offsets[nb] = (int32_t)n;
            *n_buckets = nb;
        }
    }
}

// One group of LPR lanes per destination row; 16-byte loads when the source allows it.
template <bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *corpus, long long src_stride, int d, const int32_t *perm,
                                                           long long n, float *sorted, long long dst_stride, float *inv_norm,
                                                           int32_t *gid, int32_t id_base) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int d4 = (int)(dst_stride >> 2);
    for (long long i = wave; i < n; i += nwaves) {
        const long long src = perm[i];
        const float *s = corpus + src * src_stride;
        float4 *dst = reinterpret_cast<float4 *>(sorted + i * dst_stride);
        float ss = 0.0f;
        for (int c = lane; c < d4; c += 64) {
            float4 v;
            if (VEC) {
                v = (c * 4 + 3 < d) ? *reinterpret_cast<const float4 *>(s + c * 4) : make_float4(0, 0, 0, 0);
                if (!(c * 4 + 3 < d)) {
                    v.x = c * 4 + 0 < d ? s[c * 4 + 0] : 0.0f;
                    v.y = c * 4 + 1 < d ? s[c * 4 + 1] : 0.0f;
                    v.z = c * 4 + 2 < d ? s[c * 4 + 2] : 0.0f;
                    v.w = 0.0f;
                }
            } else {
                v.x = c * 4 + 0 < d ? s[c * 4 + 0] : 0.0f;
                v.y = c * 4 + 1 < d ? s[c * 4 + 1] : 0.0f;
                v.z = c * 4 + 2 < d ? s[c * 4 + 2] : 0.0f;
                v.w = c * 4 + 3 < d ? s[c * 4 + 3] : 0.0f;
            }
            dst[c] = v;
            ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
        }
        if (inv_norm) {
            for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
            if (lane == 0) inv_norm[i] = 1.0f / fmaxf(sqrtf(ss), 1e-8f);  // cosine_similarity eps (nlsh/data.py:109)
        }
        if (gid && lane == 0) gid[i] = (int32_t)src + id_base;
    }
}

// bucket sizes as sort keys + identity values for the size-descending schedule order
__global__ void bucket_sizes_kernel(const int32_t *offsets, long long nb, uint32_t *sizes, int32_t *idx) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (long long)gridDim.x * blockDim.x) {
        sizes[i] = (uint32_t)(offsets[i + 1] - offsets[i]);
        idx[i] = (int32_t)i;
    }
}


// ---- cells: runs of consecutive small buckets that share one row window of the tiled scan ------------------------------------
// Greedy packing in CSR order: a bucket of more than `W` rows is a cell of its own; consecutive buckets of <= W rows are packed
// into one cell while the cell's rows stay <= W.  One WAVEFRONT packs CELL_SPAN consecutive buckets (the packing restarts at every
// span boundary, so the result is a pure function of (offsets, W) and the spans are independent): it holds 64 buckets at a time,
// one per lane, and takes one ballot per CELL, not per bucket -- the lanes past the open cell's last fitting bucket vote, the
// first of them starts the next cell.
constexpr int CELL_SPAN = 1024;

__global__ __launch_bounds__(256) void cell_flags_kernel(const int32_t *offsets, long long nb, int W, int32_t *flags) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long b0 = wave * CELL_SPAN;
    if (b0 >= nb) return;
    int cur = -1;   // first row of the open cell; -1: none (span start, or the previous bucket was a big one)
    for (int ch = 0; ch < CELL_SPAN / 64; ++ch) {
        const long long cb = b0 + (long long)ch * 64;
        if (cb >= nb) break;
        const long long b = cb + lane;
        const int nvalid = (int)min((long long)64, nb - cb);
        const int s = b < nb ? offsets[b] : 0, e = b < nb ? offsets[b + 1] : 0;
        const bool small = (e - s) <= W;
        unsigned long long starts = 0;
        int pos = 0;
        while (pos < nvalid) {
            const int s_p = __builtin_amdgcn_readlane(s, pos), e_p = __builtin_amdgcn_readlane(e, pos);
            if (e_p - s_p > W) {          // big bucket: a cell of its own, and nothing is open behind it
                starts |= 1ull << pos;
                cur = -1;
                ++pos;
                continue;
            }
            if (cur < 0 || e_p - cur > W) {   // does not fit the open cell: it opens the next one
                starts |= 1ull << pos;
                cur = s_p;
            }
            const unsigned long long stop = __ballot(lane > pos && lane < nvalid && (!small || e - cur > W));
            pos = stop ? __ffsll((long long)stop) - 1 : nvalid;
        }
        if (b < nb) flags[b] = (int32_t)((starts >> lane) & 1ull);
    }
}

// rank = inclusive prefix sum of the flags: cell of bucket b = rank[b] - 1.  cell_offsets is written for ALL nb + 1 slots (slots past
// the last cell hold N: zero-row cells), so the size sort below can run on nb entries without the host knowing the cell count.
__global__ void emit_cells_kernel(const int32_t *offsets, const int32_t *flags, const int32_t *rank, long long nb, int32_t *cell_of,
                                  int32_t *cell_offsets, int32_t *n_cells) {
    const int32_t nc = rank[nb - 1], n_rows = offsets[nb];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= nb; i += (long long)gridDim.x * blockDim.x) {
        if (i < nb) {
            cell_of[i] = rank[i] - 1;
            if (flags[i]) cell_offsets[rank[i] - 1] = offsets[i];
        }
        if (i >= nc) cell_offsets[i] = n_rows;
        if (i == 0) *n_cells = nc;
    }
}

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct OrderWs {
    size_t sizes, sorted_sizes, idx, tmp, tmp_bytes, total;
};

static int order_layout(long long nb, OrderWs *w, hipStream_t s) {
    const size_t n4 = align_up((size_t)(nb > 0 ? nb : 1) * 4);
    size_t t_sort = 0;
    uint32_t *nulk = nullptr;
    int32_t *nulv = nullptr;
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, t_sort, nulk, nulk, nulv, nulv, (size_t)nb, 0, 32, s);
    if (e != hipSuccess) { set_error("rocprim::radix_sort_pairs_desc size query: %s", hipGetErrorString(e)); return NLSH_E_HIP; }
    w->sizes = 0;
    w->sorted_sizes = n4;
    w->idx = 2 * n4;
    w->tmp = 3 * n4;
    w->tmp_bytes = align_up(t_sort);
    w->total = w->tmp + w->tmp_bytes;
    return NLSH_OK;
}

struct CsrWs {
    size_t sk, iota, flags, rank, tmp, tmp_bytes, total;
};

static int csr_layout(long long n, CsrWs *w, hipStream_t s) {
    size_t n4 = align_up((size_t)(n > 0 ? n : 1) * 4);
    size_t t_sort = 0, t_scan = 0;
    int32_t *nul = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, t_sort, nul, nul, nul, nul, (size_t)n, 0, 32, s);
    if (e != hipSuccess) { set_error("rocprim::radix_sort_pairs size query: %s", hipGetErrorString(e)); return NLSH_E_HIP; }
    e = rocprim::inclusive_scan(nullptr, t_scan, nul, nul, (size_t)n, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) { set_error("rocprim::inclusive_scan size query: %s", hipGetErrorString(e)); return NLSH_E_HIP; }
    w->sk = 0;
    w->iota = w->sk + n4;
    w->flags = w->iota + n4;
    w->rank = w->flags + n4;
    w->tmp = w->rank + n4;
    w->tmp_bytes = align_up(t_sort > t_scan ? t_sort : t_scan);
    w->total = w->tmp + w->tmp_bytes;
    return NLSH_OK;
}

}  // namespace nlsh

using namespace nlsh;

extern "C" size_t nlsh_build_csr_workspace(int64_t n) {
    if (n < 0) { set_error("build_csr_workspace: n=%lld", (long long)n); return 0; }
    CsrWs w;
    if (csr_layout(n, &w, nullptr) != NLSH_OK) return 0;
    return w.total;
}

extern "C" int nlsh_build_csr(const int32_t *keys, int64_t n, int32_t *perm, int32_t *uniq_keys, int32_t *offsets,
                              int32_t *n_buckets, void *workspace, size_t workspace_bytes, nlsh_stream_t stream) {
    NLSH_REQUIRE(n >= 0 && n < (1ll << 31), NLSH_E_INVALID, "build_csr: n=%lld", (long long)n);
    NLSH_REQUIRE(offsets && n_buckets, NLSH_E_INVALID, "build_csr: null output");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        NLSH_CHECK_HIP(hipMemsetAsync(offsets, 0, sizeof(int32_t), s));
        NLSH_CHECK_HIP(hipMemsetAsync(n_buckets, 0, sizeof(int32_t), s));
        return NLSH_OK;
    }
    NLSH_REQUIRE(keys && perm && uniq_keys && workspace, NLSH_E_INVALID, "build_csr: null pointer");
    CsrWs w;
    int rc = csr_layout(n, &w, s);
    if (rc != NLSH_OK) return rc;
    NLSH_REQUIRE(workspace_bytes >= w.total, NLSH_E_WORKSPACE, "build_csr: workspace %zu < %zu", workspace_bytes, w.total);
    NLSH_REQUIRE(((uintptr_t)workspace & 15) == 0, NLSH_E_INVALID, "build_csr: the workspace must be 16-byte aligned");
    char *base = (char *)workspace;
    int32_t *sk = (int32_t *)(base + w.sk), *iota = (int32_t *)(base + w.iota);
    int32_t *flags = (int32_t *)(base + w.flags), *rank = (int32_t *)(base + w.rank);
    void *tmp = base + w.tmp;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(256), 0, s, iota, (long long)n);
    size_t tb = w.tmp_bytes;
    // stable: equal keys keep ascending row order (the insertion order of indexer.py:8-13)
    NLSH_CHECK_HIP(rocprim::radix_sort_pairs(tmp, tb, keys, sk, iota, perm, (size_t)n, 0, 32, s));
    hipLaunchKernelGGL(head_flags_kernel, dim3(grid), dim3(256), 0, s, sk, (long long)n, flags);
    tb = w.tmp_bytes;
    NLSH_CHECK_HIP(rocprim::inclusive_scan(tmp, tb, flags, rank, (size_t)n, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(emit_buckets_kernel, dim3(grid), dim3(256), 0, s, sk, flags, rank, (long long)n, uniq_keys, offsets, n_buckets);
    NLSH_CHECK_HIP(hipGetLastError());
    return NLSH_OK;
}

extern "C" size_t nlsh_bucket_order_workspace(int64_t n_buckets) {
    if (n_buckets < 0) { set_error("bucket_order_workspace: n_buckets=%lld", (long long)n_buckets); return 0; }
    OrderWs w;
    if (order_layout(n_buckets, &w, nullptr) != NLSH_OK) return 0;
    return w.total;
}

extern "C" int nlsh_bucket_order(const int32_t *offsets, int64_t n_buckets, int32_t *order_out, void *workspace,
                                 size_t workspace_bytes, nlsh_stream_t stream) {
    NLSH_REQUIRE(n_buckets >= 0 && n_buckets < (1ll << 31), NLSH_E_INVALID, "bucket_order: n_buckets=%lld", (long long)n_buckets);
    if (n_buckets == 0) return NLSH_OK;
    NLSH_REQUIRE(offsets && order_out && workspace, NLSH_E_INVALID, "bucket_order: null pointer");
    hipStream_t s = (hipStream_t)stream;
    OrderWs w;
    int rc = order_layout(n_buckets, &w, s);
    if (rc != NLSH_OK) return rc;
    NLSH_REQUIRE(workspace_bytes >= w.total, NLSH_E_WORKSPACE, "bucket_order: workspace %zu < %zu", workspace_bytes, w.total);
    NLSH_REQUIRE(((uintptr_t)workspace & 15) == 0, NLSH_E_INVALID, "bucket_order: the workspace must be 16-byte aligned");
    char *base = (char *)workspace;
    uint32_t *sizes = (uint32_t *)(base + w.sizes), *sorted_sizes = (uint32_t *)(base + w.sorted_sizes);
    int32_t *idx = (int32_t *)(base + w.idx);
    int grid = (int)((n_buckets + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(bucket_sizes_kernel, dim3(grid), dim3(256), 0, s, offsets, (long long)n_buckets, sizes, idx);
    size_t tb = w.tmp_bytes;
    // stable: buckets of equal size keep ascending key order -> the order is a pure function of the index
    NLSH_CHECK_HIP(rocprim::radix_sort_pairs_desc(base + w.tmp, tb, sizes, sorted_sizes, idx, order_out, (size_t)n_buckets, 0, 32, s));
    NLSH_CHECK_HIP(hipGetLastError());
    return NLSH_OK;
}

struct CellWs {
    size_t flags, rank, tmp, tmp_bytes, order, total;
};
static int cell_layout(long long nb, CellWs *w, hipStream_t s) {
    const size_t n4 = align_up((size_t)(nb > 0 ? nb : 1) * 4);
    size_t t_scan = 0;
    int32_t *nul = nullptr;
    hipError_t e = rocprim::inclusive_scan(nullptr, t_scan, nul, nul, (size_t)nb, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) { set_error("rocprim::inclusive_scan size query: %s", hipGetErrorString(e)); return NLSH_E_HIP; }
    OrderWs ow;
    int rc = order_layout(nb, &ow, s);
    if (rc != NLSH_OK) return rc;
    w->flags = 0;
    w->rank = n4;
    w->tmp = 2 * n4;
    w->tmp_bytes = align_up(t_scan);
    w->order = w->tmp + w->tmp_bytes;
    w->total = w->order + ow.total;
    return NLSH_OK;
}

extern "C" size_t nlsh_build_cells_workspace(int64_t n_buckets) {
    if (n_buckets < 0) { set_error("build_cells_workspace: n_buckets=%lld", (long long)n_buckets); return 0; }
    CellWs w;
    if (cell_layout(n_buckets, &w, nullptr) != NLSH_OK) return 0;
    return w.total;
}

extern "C" int nlsh_build_cells(const int32_t *offsets, int64_t n_buckets, int window_rows, int32_t *cell_of, int32_t *cell_offsets,
                                int32_t *cell_order, int32_t *n_cells, void *workspace, size_t workspace_bytes, nlsh_stream_t stream) {
    NLSH_REQUIRE(n_buckets >= 0 && n_buckets < (1ll << 31), NLSH_E_INVALID, "build_cells: n_buckets=%lld", (long long)n_buckets);
    NLSH_REQUIRE(window_rows >= 1 && window_rows <= 256, NLSH_E_UNSUPPORTED, "build_cells: window_rows=%d not in [1,256] (one segment of the tiled scan)", window_rows);
    NLSH_REQUIRE(n_cells, NLSH_E_INVALID, "build_cells: null output");
    hipStream_t s = (hipStream_t)stream;
    if (n_buckets == 0) {
        NLSH_CHECK_HIP(hipMemsetAsync(n_cells, 0, sizeof(int32_t), s));
        return NLSH_OK;
    }
    NLSH_REQUIRE(offsets && cell_of && cell_offsets && cell_order && workspace, NLSH_E_INVALID, "build_cells: null pointer");
    CellWs w;
    int rc = cell_layout(n_buckets, &w, s);
    if (rc != NLSH_OK) return rc;
    NLSH_REQUIRE(workspace_bytes >= w.total, NLSH_E_WORKSPACE, "build_cells: workspace %zu < %zu", workspace_bytes, w.total);
    NLSH_REQUIRE(((uintptr_t)workspace & 15) == 0, NLSH_E_INVALID, "build_cells: the workspace must be 16-byte aligned");
    char *base = (char *)workspace;
    int32_t *flags = (int32_t *)(base + w.flags), *rank = (int32_t *)(base + w.rank);
    const long long spans = (n_buckets + CELL_SPAN - 1) / CELL_SPAN;
    hipLaunchKernelGGL(cell_flags_kernel, dim3((unsigned)((spans + 3) / 4)), dim3(256), 0, s, offsets, (long long)n_buckets, window_rows, flags);
    size_t tb = w.tmp_bytes;
    NLSH_CHECK_HIP(rocprim::inclusive_scan(base + w.tmp, tb, flags, rank, (size_t)n_buckets, rocprim::plus<int32_t>(), s));
    int grid = (int)((n_buckets + 256) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(emit_cells_kernel, dim3(grid), dim3(256), 0, s, offsets, flags, rank, (long long)n_buckets, cell_of, cell_offsets, n_cells);
    NLSH_CHECK_HIP(hipGetLastError());
    // schedule order of the cells: by descending rows over all n_buckets slots (the zero-row slots past the last cell sort behind
    // every real cell, so the first *n_cells entries are the order of the cells)
    return nlsh_bucket_order(cell_offsets, n_buckets, cell_order, base + w.order, workspace_bytes - w.order, stream);
}

extern "C" int nlsh_gather_rows(const float *corpus, int64_t src_stride, int d, const int32_t *perm, int64_t n,
                                float *sorted, int64_t dst_stride, float *inv_norm, int32_t *gid, int32_t id_base,
                                nlsh_stream_t stream) {
    NLSH_REQUIRE(n >= 0 && d >= 1 && d <= NLSH_MAX_DIM, NLSH_E_INVALID, "gather_rows: n=%lld d=%d", (long long)n, d);
    if (n == 0) return NLSH_OK;
    NLSH_REQUIRE(corpus && perm && sorted, NLSH_E_INVALID, "gather_rows: null pointer");
    NLSH_REQUIRE(dst_stride >= d && (dst_stride & 3) == 0 && ((uintptr_t)sorted & 15) == 0, NLSH_E_INVALID,
                 "gather_rows: dst_stride=%lld must be >= d, a multiple of 4, and `sorted` 16-byte aligned", (long long)dst_stride);
    NLSH_REQUIRE(src_stride >= d, NLSH_E_INVALID, "gather_rows: src_stride < d");
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    bool vec = ((src_stride & 3) == 0) && (((uintptr_t)corpus & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(gather_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, corpus, (long long)src_stride, d, perm,
                           (long long)n, sorted, (long long)dst_stride, inv_norm, gid, id_base);
    else
        hipLaunchKernelGGL(gather_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, corpus, (long long)src_stride, d, perm,
                           (long long)n, sorted, (long long)dst_stride, inv_norm, gid, id_base);
    NLSH_CHECK_HIP(hipGetLastError());
    return NLSH_OK;
}
