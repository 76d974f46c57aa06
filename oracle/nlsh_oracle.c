/*
 * nlsh_oracle.c -- CPU restatement of the reference's query-time hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (neural-locality-sensitive-hashing_amd/)
 * may link, load or call this file: only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg use it, as the checker / the timed CPU baseline -- never as a fallback.
 *
 * Parity pin: every function here is checked against golden vectors produced by running the
 * UNMODIFIED reference in the build container (tests/golden/make_golden.py ->
 * tests/golden/*.npz|json; checked by tests/test_oracle_golden.py).
 *
 * Reference lines restated (paths relative to the reference repo):
 *   - binarr_to_int / hash_codes ............ nlsh/utils.pyx:6-15, 18-32   (int16 wrap: F2)
 *   - _binarr_to_int (untruncated) .......... eval.py:49-53
 *   - encoder + head forward ................ encoders.py:18-21,39-55 ; nlsh/hashings.py:13-27
 *   - hard bits / Bernoulli multi-probe ..... nlsh/hashings.py:66-92
 *   - build_index ........................... nlsh/indexer.py:6-24
 *   - gather + distance + top-k ............. nlsh/indexer.py:56-96 ; nlsh/data.py:99-109,191-201
 *
 * Plain C99 + libm (+ optional OpenMP for the timed baseline).  No reference source is copied.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ bit packing */

/* nlsh/utils.pyx:6-15 : MSB-first accumulate in int32, RETURN TYPE int16 (two's-complement wrap). */
int32_t oracle_pack_ref_int16(const int32_t *bits, int n_bits, int stride) {
    int32_t out = 0;
    for (int i = 0; i < n_bits; ++i) {
        int32_t bit = bits[(size_t)i * stride];
        out = (int32_t)(((uint32_t)out << 1) | (uint32_t)bit);
    }
    return (int32_t)(int16_t)(uint16_t)((uint32_t)out & 0xFFFFu);
}

/* eval.py:49-53 : same shift/or on an unbounded Python int (no truncation). H <= 63 here. */
int64_t oracle_pack_full(const int32_t *bits, int n_bits, int stride) {
    int64_t out = 0;
    for (int i = 0; i < n_bits; ++i) out = (out << 1) | (int64_t)bits[(size_t)i * stride];
    return out;
}

/* nlsh/utils.pyx:18-32 minus the Python set: keys[b][j] for codes[B][n][H] (C-contiguous int32).
 * mode 0 = ref_int16, 1 = full (returned in int64). The set() is formed by the Python wrapper. */
void oracle_hash_codes(const int32_t *codes, int64_t B, int n, int H, int mode, int64_t *keys_out) {
    for (int64_t b = 0; b < B; ++b)
        for (int j = 0; j < n; ++j) {
            const int32_t *p = codes + ((size_t)b * n + j) * H;
            keys_out[b * n + j] = mode == 0 ? (int64_t)oracle_pack_ref_int16(p, H, 1) : oracle_pack_full(p, H, 1);
        }
}

/* ------------------------------------------------------------------ MLP forward
 * y = act_l( x W_l^T + b_l ), ReLU on hidden layers, identity on the head (z), fp32.
 * Summation is a k-ordered fmaf chain starting from 0, bias added afterwards: this is the order
 * the HIP kernel's fp32 MFMA chain uses (DESIGN.md "encode_hash"), so kernel-vs-oracle is bit
 * exact; oracle-vs-reference (BLAS order) is pinned through tests/golden/g2_hasher.npz with the
 * |z|-gated flip policy (SURVEY.md F5 / hard part 3).
 */
void oracle_mlp_forward(const float *x, int64_t n, int n_layers, const float *const *W, const float *const *b,
                        const int *dims, float *z_out) {
    int maxd = 0;
    for (int l = 0; l <= n_layers; ++l) if (dims[l] > maxd) maxd = dims[l];
    float *cur = (float *)malloc(sizeof(float) * (size_t)maxd);
    float *nxt = (float *)malloc(sizeof(float) * (size_t)maxd);
    for (int64_t r = 0; r < n; ++r) {
        memcpy(cur, x + (size_t)r * dims[0], sizeof(float) * (size_t)dims[0]);
        for (int l = 0; l < n_layers; ++l) {
            int K = dims[l], N = dims[l + 1];
            for (int j = 0; j < N; ++j) {
                const float *w = W[l] + (size_t)j * K;
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) acc = fmaf(cur[k], w[k], acc);
                if (b[l]) acc = acc + b[l][j];
                if (l + 1 < n_layers) acc = acc > 0.0f ? acc : 0.0f;   /* ReLU, encoders.py:19-20 */
                nxt[j] = acc;
            }
            float *t = cur; cur = nxt; nxt = t;
        }
        memcpy(z_out + (size_t)r * dims[n_layers], cur, sizeof(float) * (size_t)dims[n_layers]);
    }
    free(cur); free(nxt);
}

/* nlsh/hashings.py:22-26,67-69: probability the Bernoulli is built from.
 * act 0: sigmoid(z); act 1: tanh(z)/2 + 0.5.  `probs_raw` gets the module output (sigmoid or tanh). */
static inline float oracle_sigmoid(float z) { return 1.0f / (1.0f + expf(-z)); }

void oracle_head_probs(const float *z, int64_t count, int act, float *probs_raw, float *probs01) {
    for (int64_t i = 0; i < count; ++i) {
        float raw = act == 0 ? oracle_sigmoid(z[i]) : tanhf(z[i]);
        float p = act == 0 ? raw : raw / 2.0f + 0.5f;
        if (probs_raw) probs_raw[i] = raw;
        if (probs01) probs01[i] = p;
    }
}

/* ------------------------------------------------------------------ Philox4x32-10 (multi-probe sampler)
 * The reference samples with torch's global RNG (nlsh/hashings.py:80), which no other
 * implementation can reproduce; the build defines its own counter-based stream
 * (key = seed, counter = (row, probe, word, 0)) and the HIP kernel uses the same one, so
 * sampling is bit-identical between kernel and oracle given the same probabilities.
 */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void oracle_philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t c[4] = {c0, c1, c2, c3};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

static inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

/* Keys of one row: probe 0 = hard bits (p > 0.5), probes 1..n-1 = Bernoulli(p) draws (u < p).
 * Deduplicated in first-occurrence order (set semantics of nlsh/utils.pyx:26-31).
 * Returns the number of distinct keys written to keys_out[0..n). */
int oracle_row_keys(const float *p01, int H, int n_probes, int key_mode, uint64_t seed, int64_t row,
                    int64_t *keys_out) {
    int cnt = 0;
    for (int j = 0; j < n_probes; ++j) {
        uint64_t code = 0;
        uint32_t rnd[4];
        for (int h = 0; h < H; ++h) {
            int bit;
            if (j == 0) bit = p01[h] > 0.5f;
            else {
                if ((h & 3) == 0) oracle_philox4x32(seed, (uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)j, (uint32_t)(h >> 2), rnd);
                bit = u01(rnd[h & 3]) < p01[h];
            }
            code = (code << 1) | (uint64_t)bit;
        }
        int64_t key = key_mode == 0 ? (int64_t)(int16_t)(uint16_t)(code & 0xFFFFu) : (int64_t)code;
        int dup = 0;
        for (int t = 0; t < cnt; ++t) if (keys_out[t] == key) { dup = 1; break; }
        if (!dup) keys_out[cnt++] = key;
    }
    return cnt;
}

/* All rows at once (rows >= n_multi_rows single-probe: nlsh/indexer.py:51-53). keys_out [n][n_probes]. */
void oracle_rows_keys(const float *p01, int64_t n, int H, int n_probes, int key_mode, uint64_t seed, int64_t row0,
                      int64_t n_multi_rows, int64_t *keys_out, int32_t *nkeys_out) {
    for (int64_t r = 0; r < n; ++r) {
        int npr = r < n_multi_rows ? n_probes : 1;
        for (int j = 0; j < n_probes; ++j) keys_out[r * n_probes + j] = 0;
        nkeys_out[r] = oracle_row_keys(p01 + (size_t)r * H, H, npr, key_mode, seed, row0 + r, keys_out + (size_t)r * n_probes);
    }
}

/* ------------------------------------------------------------------ index build (CSR)
 * nlsh/indexer.py:6-24 with one key per row: bucket -> ascending row list.  Output: buckets in
 * ascending key order; perm = rows grouped by bucket (ascending inside); offsets[nb+1]. */
typedef struct { int64_t key; int32_t row; } kr_t;
static int kr_cmp(const void *a, const void *b) {
    const kr_t *x = (const kr_t *)a, *y = (const kr_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row);
}

int64_t oracle_build_csr(const int64_t *keys, int64_t n, int32_t *perm, int64_t *uniq_keys, int64_t *offsets) {
    kr_t *v = (kr_t *)malloc(sizeof(kr_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { v[i].key = keys[i]; v[i].row = (int32_t)i; }
    qsort(v, (size_t)n, sizeof(kr_t), kr_cmp);
    int64_t nb = 0;
    for (int64_t i = 0; i < n; ++i) {
        perm[i] = v[i].row;
        if (i == 0 || v[i].key != v[i - 1].key) { uniq_keys[nb] = v[i].key; offsets[nb] = i; ++nb; }
    }
    offsets[nb] = n;
    free(v);
    return nb;
}

/* ------------------------------------------------------------------ distances
 * L2  (nlsh/data.py:191-201): F.pairwise_distance = sqrt(sum_j ((q_j - c_j) + 1e-6)^2)
 * cos (nlsh/data.py:99-109) : 1 - sum_j (q_j / max(|q|,1e-8)) (c_j / max(|c|,1e-8))
 */
float oracle_l2(const float *q, const float *c, int d) {
    float s = 0.0f;
    for (int j = 0; j < d; ++j) { float t = (q[j] - c[j]) + 1e-6f; s = fmaf(t, t, s); }
    return sqrtf(s);
}
float oracle_cosine(const float *q, const float *c, int d) {
    float qq = 0.0f, cc = 0.0f;
    for (int j = 0; j < d; ++j) { qq = fmaf(q[j], q[j], qq); cc = fmaf(c[j], c[j], cc); }
    float qn = fmaxf(sqrtf(qq), 1e-8f), cn = fmaxf(sqrtf(cc), 1e-8f);
    float s = 0.0f;
    for (int j = 0; j < d; ++j) s = fmaf(q[j] / qn, c[j] / cn, s);
    return 1.0f - s;
}
double oracle_l2_f64(const float *q, const float *c, int d) {
    double s = 0.0;
    for (int j = 0; j < d; ++j) { double t = ((double)q[j] - (double)c[j]) + 1e-6; s += t * t; }
    return sqrt(s);
}
double oracle_cosine_f64(const float *q, const float *c, int d) {
    double qq = 0, cc = 0, s = 0;
    for (int j = 0; j < d; ++j) { qq += (double)q[j] * q[j]; cc += (double)c[j] * c[j]; s += (double)q[j] * c[j]; }
    return 1.0 - s / (fmax(sqrt(qq), 1e-8) * fmax(sqrt(cc), 1e-8));
}

void oracle_distances(const float *q, const float *corpus, int d, const int32_t *rows, int64_t n_rows, int metric,
                      float *out32, double *out64) {
    for (int64_t i = 0; i < n_rows; ++i) {
        const float *c = corpus + (size_t)rows[i] * d;
        if (out32) out32[i] = metric == 0 ? oracle_l2(q, c, d) : oracle_cosine(q, c, d);
        if (out64) out64[i] = metric == 0 ? oracle_l2_f64(q, c, d) : oracle_cosine_f64(q, c, d);
    }
}

/* ------------------------------------------------------------------ batched query (gather + distance + top-k)
 * nlsh/indexer.py:62-95 for every query, on a CSR index.  Candidate order = keys in the given
 * order, rows ascending in a bucket (what torch.cat of index2row tensors yields).  Top-k order is
 * (distance asc, row id asc): the build's deterministic refinement of torch.topk's unspecified
 * tie order.  Fewer than k candidates -> all of them in that order with +inf/-1
 * padding here; the F7 fallback ("last key's rows") is applied by the Python wrapper.
 */
static int64_t find_bucket(const int64_t *uniq, int64_t nb, int64_t key) {
    int64_t lo = 0, hi = nb;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (uniq[mid] < key) lo = mid + 1; else hi = mid; }
    return (lo < nb && uniq[lo] == key) ? lo : -1;
}

void oracle_query_batch(const float *corpus, int d, const int32_t *perm, const int64_t *uniq_keys,
                        const int64_t *offsets, int64_t nb, const float *queries, int64_t Q,
                        const int64_t *qkeys, const int32_t *nkeys, int P, int k, int metric,
                        float *out_dist, int32_t *out_idx, int64_t *out_ncand) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t q = 0; q < Q; ++q) {
        const float *qv = queries + (size_t)q * d;
        float *bd = out_dist + (size_t)q * k;
        int32_t *bi = out_idx + (size_t)q * k;
        int have = 0;
        int64_t nc = 0;
        for (int t = 0; t < k; ++t) { bd[t] = INFINITY; bi[t] = -1; }
        for (int p = 0; p < nkeys[q]; ++p) {
            int64_t bkt = find_bucket(uniq_keys, nb, qkeys[(size_t)q * P + p]);
            if (bkt < 0) continue;
            for (int64_t i = offsets[bkt]; i < offsets[bkt + 1]; ++i) {
                int32_t row = perm[i];
                const float *c = corpus + (size_t)row * d;
                float dist = metric == 0 ? oracle_l2(qv, c, d) : oracle_cosine(qv, c, d);
                ++nc;
                if (have == k && !(dist < bd[k - 1] || (dist == bd[k - 1] && row < bi[k - 1]))) continue;
                int pos = have < k ? have : k - 1;
                while (pos > 0 && (dist < bd[pos - 1] || (dist == bd[pos - 1] && row < bi[pos - 1]))) {
                    bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos;
                }
                bd[pos] = dist; bi[pos] = row;
                if (have < k) ++have;
            }
        }
        out_ncand[q] = nc;
    }
}

/* ------------------------------------------------------------------ the same scan, 8 candidates per AVX2 lane set
 * bench.py's cpu_baseline times THIS form (the scalar loop above is a 4-cycle-latency dependent fmaf chain per
 * candidate, i.e. latency- not bandwidth-bound, which is not what a tuned CPU implementation would run).  Lane l of a
 * vector follows candidate l's OWN k-ascending chain ((q_j - c_j) + eps, fmaf), so every distance has the same bits as
 * oracle_l2 / oracle_cosine and the result is identical to oracle_query_batch (tests/test_oracle_golden.py checks
 * that); rows of 8 candidates are brought in as 8x8 blocks and transposed in registers. */
#include <immintrin.h>

static inline void transpose8(__m256 r[8]) {
    __m256 t0 = _mm256_unpacklo_ps(r[0], r[1]), t1 = _mm256_unpackhi_ps(r[0], r[1]);
    __m256 t2 = _mm256_unpacklo_ps(r[2], r[3]), t3 = _mm256_unpackhi_ps(r[2], r[3]);
    __m256 t4 = _mm256_unpacklo_ps(r[4], r[5]), t5 = _mm256_unpackhi_ps(r[4], r[5]);
    __m256 t6 = _mm256_unpacklo_ps(r[6], r[7]), t7 = _mm256_unpackhi_ps(r[6], r[7]);
    __m256 u0 = _mm256_shuffle_ps(t0, t2, 0x44), u1 = _mm256_shuffle_ps(t0, t2, 0xEE);
    __m256 u2 = _mm256_shuffle_ps(t1, t3, 0x44), u3 = _mm256_shuffle_ps(t1, t3, 0xEE);
    __m256 u4 = _mm256_shuffle_ps(t4, t6, 0x44), u5 = _mm256_shuffle_ps(t4, t6, 0xEE);
    __m256 u6 = _mm256_shuffle_ps(t5, t7, 0x44), u7 = _mm256_shuffle_ps(t5, t7, 0xEE);
    r[0] = _mm256_permute2f128_ps(u0, u4, 0x20); r[1] = _mm256_permute2f128_ps(u1, u5, 0x20);
    r[2] = _mm256_permute2f128_ps(u2, u6, 0x20); r[3] = _mm256_permute2f128_ps(u3, u7, 0x20);
    r[4] = _mm256_permute2f128_ps(u0, u4, 0x31); r[5] = _mm256_permute2f128_ps(u1, u5, 0x31);
    r[6] = _mm256_permute2f128_ps(u2, u6, 0x31); r[7] = _mm256_permute2f128_ps(u3, u7, 0x31);
}

/* distances of 8 candidates (row pointers c[0..7]) to one query; qn = q / max(|q|, 1e-8) for cosine (else unused) */
static void dist8(const float *q, const float *qn, const float *const c[8], int d, int metric, float out[8]) {
    const __m256 eps = _mm256_set1_ps(1e-6f);
    __m256 acc = _mm256_setzero_ps(), cc = _mm256_setzero_ps(), cn = _mm256_setzero_ps();
    const int d8 = d & ~7;
    if (metric != 0) {   /* cosine: the candidates' norms first (their own k-ascending chains) */
        for (int j = 0; j < d8; j += 8) {
            __m256 r[8];
            for (int l = 0; l < 8; ++l) r[l] = _mm256_loadu_ps(c[l] + j);
            transpose8(r);
            for (int t = 0; t < 8; ++t) cc = _mm256_fmadd_ps(r[t], r[t], cc);
        }
        for (int j = d8; j < d; ++j) {
            __m256 col = _mm256_set_ps(c[7][j], c[6][j], c[5][j], c[4][j], c[3][j], c[2][j], c[1][j], c[0][j]);
            cc = _mm256_fmadd_ps(col, col, cc);
        }
        cn = _mm256_max_ps(_mm256_sqrt_ps(cc), _mm256_set1_ps(1e-8f));
    }
    for (int j = 0; j < d8; j += 8) {
        __m256 r[8];
        for (int l = 0; l < 8; ++l) r[l] = _mm256_loadu_ps(c[l] + j);
        transpose8(r);
        for (int t = 0; t < 8; ++t) {
            if (metric == 0) {
                __m256 v = _mm256_add_ps(_mm256_sub_ps(_mm256_set1_ps(q[j + t]), r[t]), eps);
                acc = _mm256_fmadd_ps(v, v, acc);
            } else {
                acc = _mm256_fmadd_ps(_mm256_set1_ps(qn[j + t]), _mm256_div_ps(r[t], cn), acc);
            }
        }
    }
    for (int j = d8; j < d; ++j) {
        __m256 col = _mm256_set_ps(c[7][j], c[6][j], c[5][j], c[4][j], c[3][j], c[2][j], c[1][j], c[0][j]);
        if (metric == 0) {
            __m256 v = _mm256_add_ps(_mm256_sub_ps(_mm256_set1_ps(q[j]), col), eps);
            acc = _mm256_fmadd_ps(v, v, acc);
        } else {
            acc = _mm256_fmadd_ps(_mm256_set1_ps(qn[j]), _mm256_div_ps(col, cn), acc);
        }
    }
    if (metric == 0) acc = _mm256_sqrt_ps(acc);
    else acc = _mm256_sub_ps(_mm256_set1_ps(1.0f), acc);
    _mm256_storeu_ps(out, acc);
}

static inline void topk_push(float *bd, int32_t *bi, int *have, int k, float dist, int32_t row) {
    if (*have == k && !(dist < bd[k - 1] || (dist == bd[k - 1] && row < bi[k - 1]))) return;
    int pos = *have < k ? *have : k - 1;
    while (pos > 0 && (dist < bd[pos - 1] || (dist == bd[pos - 1] && row < bi[pos - 1]))) {
        bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos;
    }
    bd[pos] = dist; bi[pos] = row;
    if (*have < k) ++*have;
}

void oracle_query_batch_simd(const float *corpus, int d, const int32_t *perm, const int64_t *uniq_keys,
                             const int64_t *offsets, int64_t nb, const float *queries, int64_t Q,
                             const int64_t *qkeys, const int32_t *nkeys, int P, int k, int metric,
                             float *out_dist, int32_t *out_idx, int64_t *out_ncand) {
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        float *qn = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int64_t q = 0; q < Q; ++q) {
            const float *qv = queries + (size_t)q * d;
            float *bd = out_dist + (size_t)q * k;
            int32_t *bi = out_idx + (size_t)q * k;
            int have = 0;
            int64_t nc = 0;
            for (int t = 0; t < k; ++t) { bd[t] = INFINITY; bi[t] = -1; }
            if (metric != 0) {
                float qq = 0.0f;
                for (int j = 0; j < d; ++j) qq = fmaf(qv[j], qv[j], qq);
                const float nrm = fmaxf(sqrtf(qq), 1e-8f);
                for (int j = 0; j < d; ++j) qn[j] = qv[j] / nrm;
            }
            for (int p = 0; p < nkeys[q]; ++p) {
                int64_t bkt = find_bucket(uniq_keys, nb, qkeys[(size_t)q * P + p]);
                if (bkt < 0) continue;
                int64_t i = offsets[bkt];
                const int64_t end = offsets[bkt + 1];
                nc += end - i;
                for (; i + 8 <= end; i += 8) {
                    const float *c[8];
                    float dist[8];
                    for (int l = 0; l < 8; ++l) c[l] = corpus + (size_t)perm[i + l] * d;
                    dist8(qv, qn, c, d, metric, dist);
                    for (int l = 0; l < 8; ++l) topk_push(bd, bi, &have, k, dist[l], perm[i + l]);
                }
                for (; i < end; ++i) {
                    const int32_t row = perm[i];
                    const float *c = corpus + (size_t)row * d;
                    topk_push(bd, bi, &have, k, metric == 0 ? oracle_l2(qv, c, d) : oracle_cosine(qv, c, d), row);
                }
            }
            out_ncand[q] = nc;
        }
        free(qn);
    }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
