// encode_hash: learned hash function forward (MLP) + bit packing + multi-probe keys, one kernel.
//
// Replaces (reference): encoders.py:18-21,39-55 (Linear+ReLU stack) -> nlsh/hashings.py:13-27
// (output Linear + sigmoid/tanh) -> hashings.py:66-81 (hard bits, Bernoulli samples) ->
// .cpu().numpy() -> nlsh/utils.pyx:6-32 (binarr_to_int, set()).  Nothing leaves the device.
//
// gfx950 mapping
//   * three forms, same arithmetic: index builds 128 rows per workgroup, query batches 32, batches of <= 4096 rows 16 (H16, all
//     layers as 16x16x4 tiles); described here for the 32-row-tile forms:
//   * one workgroup (NW = 8 wavefronts, 512 threads) owns M = 32*RT rows; their activations never
//     leave LDS (two ping-pong [M][S] fp32 images, 133 KB at width 256, so ONE workgroup per CU: the
//     two wavefronts per SIMD are what overlaps one wave's LDS/L2 waits, write-back and epilogue VALU
//     work with the other's MFMA chain -- with 4 waves a workgroup took 42 us, MFMA-busy for 24);
//   * a wavefront owns one 32-column tile of a hidden layer (8 tiles at width 256) for all RT row tiles,
//     so a B fragment is fetched once per RT MFMAs;
//   * every Linear layer is an fp32 MFMA chain, v_mfma_f32_32x32x2_f32: exact fp32, result is
//     bit-for-bit a k-ascending fmaf chain (guide §3 "FP32-input MFMA"), which is what the oracle
//     computes -> z is bit-exact against oracle_mlp_forward;
//   * weights are pre-packed in B-fragment order so that each lane fetches the B operands of four
//     consecutive MFMA k-steps with ONE coalesced global_load_dwordx4 (1 KiB per wave-instruction;
//     the 410 KB blob stays L2-resident); the A operands of the same four steps are ONE
//     ds_read_b128 thanks to an even/odd de-interleaved LDS row layout (pos()); row stride
//     S = width+4 floats with S/4 odd makes those reads conflict-free;
//   * bias + ReLU are fused into the accumulator write-back; sigmoid/tanh, the `> 0.5` bit rule,
//     MSB-first packing, Philox Bernoulli probes and per-row de-duplication are the epilogue.
#include <atomic>

#include "step_nodes.h"

namespace nlsh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// position of logical column k inside an LDS row: within each group of 8, evens first then odds
// so that lane half h reads k = 8c+h, 8c+2+h, 8c+4+h, 8c+6+h as one 16-byte word at 8c+4h.
__device__ __forceinline__ int pos(int k) { return (k & ~7) + ((k & 1) << 2) + ((k & 7) >> 1); }

__device__ __forceinline__ void philox4x32_10(unsigned long long seed, uint32_t c0, uint32_t c1, uint32_t c2,
                                              uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// dynamic LDS a workgroup may ask for: the CU's 160 KB less the kernel's static variables (the lookup's two block counters, r06)
constexpr size_t ENC_LDS_LIMIT = 160 * 1024 - 256;
// Batches up to this many rows take the single-image 32-row form (2 x 256 CUs x 32 rows: every workgroup resident at once)
#ifndef NLSH_ENC_BUILD_128
#define NLSH_ENC_BUILD_128 1  // index-build launches: 1 = 128-row workgroups on one LDS image, 0 = 64-row workgroups on a ping-pong pair
#endif
#ifndef NLSH_ENC_SINGLE_MAX_ROWS
#define NLSH_ENC_SINGLE_MAX_ROWS 16384
#endif
// Batches of at most this many rows take the 16-row form (H16): as long as its workgroups are at most one per CU (256 x 16 rows) it is
// the shorter critical path (24.5 us against 27.2 for the 32-row form, 64 ... 4096 rows); beyond that it loses -- see the kernel.
#ifndef NLSH_ENC_HET_PRIO
#define NLSH_ENC_HET_PRIO 3   // wave priority of the 16-row workgroups of the balanced query-batch launch (-1: leave it alone)
#endif
#ifndef NLSH_ENC_HET
#define NLSH_ENC_HET 1   // 0: every query batch of more than 4096 rows as 32-row workgroups (r02-r05), for A/B
#endif
#ifndef NLSH_ENC_H16_MAX_ROWS
#define NLSH_ENC_H16_MAX_ROWS 4096
#endif
// (r05: 48-row workgroups of THREE 16-row tiles sharing every B fragment -- one workgroup per CU for 8192 < rows <= 12288 instead of two 32-row
// ones on some CUs -- measured 44.4 us against 37.2 us on the 10^4-query batch and were removed: profiles/r05_encoder_48row_form_ab.txt.)

// Diagnostic build only (make EXTRA=-DNLSH_ENC_TRACE, tools/enc_trace.py): thread 0 of every workgroup
// leaves the 100 MHz wall_clock64 stamp of each phase boundary in the first floats of its z_out rows.
#ifdef NLSH_ENC_TRACE
// r06: the stamps of every workgroup also land in a global table (16 floats per workgroup: stamps 0..11 relative to stamp 0, [12] the
// workgroup's start on the launch's clock, [13] rows), readable whatever outputs the call asked for (tools/enc_step_trace.py: the
// fused encode + lookup launch of a query batch has no z_out)
#define NLSH_ENC_TRACE_SLOTS 4096
__device__ float g_enc_trace[NLSH_ENC_TRACE_SLOTS * 16];
extern "C" int nlsh_debug_enc_trace(float *host, int n_floats) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_enc_trace), (size_t)n_floats * 4); }
#define ENC_STAMP(i) do { if (tid == 0) stamp[i] = wall_clock64(); } while (0)
#else
#define ENC_STAMP(i) do { } while (0)
#endif

// SINGLE: one LDS image instead of the ping-pong pair.  A wave keeps the accumulators of ALL its column tiles
// (<= MT) in registers until every wave has finished reading the layer's input, then the outputs overwrite the image.
// With RT = 1 that is 33 KB at width 256 -- small enough to sit on a CU BESIDE six resident workgroups of the scan
// kernel (20 KB each), which is what lets the batch pipeline (nlsh_amd/pipeline.py) run a query batch's encode under
// the previous batch's scan; the 133 KB ping-pong form only finds a CU once the scan's dispatch queue has drained.
// H16 (r04): 16-row workgroups whose layers are ALL 16x16x4 tiles (MT = 16-column tiles a wave may own); same k-ascending fmaf
// chains (guide, FP32-input MFMA), so z keeps its bits.  Built to balance the 10^4-query batch -- 313 32-row workgroups on 256 CUs:
// the 57 CUs that host two of them take 42 us where one workgroup alone takes 28 (8192 rows 28.0 us, 8224 rows 40.9 us), and the
// launch lasts as long as its busiest CU -- and measured the other way round there: every workgroup streams the whole 410 KB of
// weights whatever its rows, a 16x16x4 tile needs twice the operand bytes per flop of a 32x32x2 one, and each further 16-row
// workgroup on a CU adds 12.5 us (24.5 / 36.9 / 47.4 / 62.0 us at 1 / 2 / 3 / 4 per CU) where a second 32-row one adds 14 for twice
// the rows.  It IS the shorter critical path while there is at most one workgroup per CU: batches of <= 4096 rows (24.5 vs 27.2 us).
// The workgroup body: rows [row_base, row_base + M) of the batch.  `blk` = the workgroup's index in the launch (its entry of the
// lookup's `hits`; workgroup 0 does the batch's initialisation).
template <int RT, int NW, bool SINGLE, int MT = 1, bool H16 = false>  // MT: column tiles a wave may own in SINGLE mode (1: width <= 32*NW)
__device__ __forceinline__ void encode_hash_body(const EncArgs &a, const PlanArgs &pa, float *smem, const long long row_base, const unsigned blk) {
    constexpr int M = H16 ? 16 * RT : 32 * RT;   // H16: RT row tiles of 16 sharing every B fragment (shipped: RT = 1; RT = 3 measured and dropped in r05)
    static_assert(!H16 || SINGLE, "the 16-row-tile form runs on a single LDS image");
    constexpr int NTH = NW * 64;
    const int S = a.S;
    float *in = smem;
    float *out = SINGLE ? smem : smem + (size_t)M * S;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = lane & 31;   // row (A) / column (B, C) inside a 32x32 tile
    const int lh = lane >> 5;   // k parity (A, B) / row-quad select (C)
#ifdef NLSH_ENC_TRACE
    __shared__ unsigned long long stamp[12];
    if (tid == 0) for (int i = 0; i < 12; ++i) stamp[i] = 0;   // LDS, not registers: twelve 64-bit values in thread 0 changed the 128-row form's allocation
#endif
    ENC_STAMP(0);
#ifdef NLSH_ENC_TRACE
    const unsigned long long core0 = __builtin_amdgcn_s_memtime();   // shader-clock counter: with the 100 MHz stamps gives the clock held
#endif

    // ---- stage the input rows (zero padded to Kp, zero rows past n) in the de-interleaved layout
    {
        const int K0 = a.L[0].K, Kp0 = a.L[0].Kp;
        const bool vec = ((K0 | (int)a.x_stride) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
        if (vec) {
            // 16-byte loads, 8 per thread issued back to back (the scalar loop exposed one HBM
            // round trip per element: ~30 us of a 60 us workgroup)
            const int Kq = Kp0 >> 2, total = M * Kq;
            const float4 *x4 = reinterpret_cast<const float4 *>(a.x);
            const long long xs4 = a.x_stride >> 2;
            for (int e0 = 0; e0 < total; e0 += NTH * 8) {
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = e0 + i * NTH + tid;
                    const int r = e / Kq, c4 = e - r * Kq;
                    const long long grow = row_base + r;
                    v[i] = (e < total && grow < a.n && 4 * c4 < K0) ? x4[grow * xs4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = e0 + i * NTH + tid;
                    if (e < total) {
                        const int r = e / Kq, c4 = e - r * Kq;
                        float *dst = in + r * S + ((4 * c4) & ~7) + ((c4 & 1) << 1);  // pos(4*c4 + j) = base + {0,4,1,5}
                        dst[0] = v[i].x; dst[4] = v[i].y; dst[1] = v[i].z; dst[5] = v[i].w;
                    }
                }
            }
        } else {
            for (int e = tid; e < M * Kp0; e += NTH) {
                int r = e / Kp0, k = e - r * Kp0;
                long long grow = row_base + r;
                float v = (grow < a.n && k < K0) ? a.x[grow * a.x_stride + k] : 0.0f;
                in[r * S + pos(k)] = v;
            }
        }
    }
    __syncthreads();
    ENC_STAMP(1);

    for (int l = 0; l < a.n_layers; ++l) {
        const LayerDesc L = a.L[l];
        const int nch = L.Kp >> 3;
        const float4 *Wp = reinterpret_cast<const float4 *>(a.packed + L.w_off);
        const float *Bp = a.packed + L.b_off;
        const bool last = (l + 1 == a.n_layers);
        if (H16) {
            // every layer as 16x16x4 tiles: the workgroup's ONE 16-row tile x (width / 16) column tiles, dealt over the waves (tile ct =
            // wave + m * NW); a wave walks its tiles in LOCK STEP over k, so the four A values of a 16-k group are read from LDS once for
            // all of them and their MFMA chains interleave.  Lane l holds A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15].
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            const int ct16 = last ? (a.H + 15) >> 4 : L.Np >> 4, ngr = (L.K + 15) >> 4;
            const float4 *W16 = reinterpret_cast<const float4 *>(a.packed + L.w16_off);
            const int g = lane >> 4, l16 = lane & 15;
            const int pa = ((g & 1) << 2) + (g >> 1);   // pos(g) inside a chunk; pos(4 + g) = pa + 2
            const float *arow = in + (size_t)l16 * S + pa;
            constexpr int OR = 4;                        // B ring per tile: groups of 16 k in flight
            f32x4v acc[MT][RT];                          // RT row tiles share every B fragment: MT * RT independent MFMA chains per wave
            float4 B[MT][OR];
            const float4 *w0[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[m][rt] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
                const int ct = min(wave + m * NW, ct16 - 1);   // tiles past the layer's width redo its last one (valid addresses), results dropped
                w0[m] = W16 + (size_t)ct * ngr * 64 + lane;
#pragma unroll
                for (int j = 0; j < OR; ++j) B[m][j] = w0[m][(size_t)min(j, ngr - 1) * 64];
            }
            if (wave < ct16) {
                for (int j0 = 0; j0 < ngr; j0 += OR) {
#pragma unroll
                    for (int jj = 0; jj < OR; ++jj) {
                        const int j = j0 + jj;
                        if (j < ngr) {
                            float4 bq[MT];
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                bq[m] = B[m][jj];
                                B[m][jj] = w0[m][(size_t)min(j + OR, ngr - 1) * 64];
                            }
                            const int c0 = 2 * j, c1 = 2 * j + 1;
                            const bool two = c1 < nch;   // Kp is a multiple of 8, not of 16: the last group may hold one chunk only
                            float a0[RT], a1[RT], a2[RT], a3[RT];
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) {
                                const float *ar = arow + (size_t)rt * 16 * S;
                                a0[rt] = ar[c0 * 8]; a1[rt] = ar[c0 * 8 + 2];
                                a2[rt] = two ? ar[c1 * 8] : 0.0f; a3[rt] = two ? ar[c1 * 8 + 2] : 0.0f;
                            }
#pragma unroll
                            for (int m = 0; m < MT; ++m)
#pragma unroll
                                for (int rt = 0; rt < RT; ++rt) acc[m][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[rt], bq[m].x, acc[m][rt], 0, 0, 0);
#pragma unroll
                            for (int m = 0; m < MT; ++m)
#pragma unroll
                                for (int rt = 0; rt < RT; ++rt) acc[m][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[rt], bq[m].y, acc[m][rt], 0, 0, 0);
                            if (two) {
#pragma unroll
                                for (int m = 0; m < MT; ++m)
#pragma unroll
                                    for (int rt = 0; rt < RT; ++rt) acc[m][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[rt], bq[m].z, acc[m][rt], 0, 0, 0);
#pragma unroll
                                for (int m = 0; m < MT; ++m)
#pragma unroll
                                    for (int rt = 0; rt < RT; ++rt) acc[m][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[rt], bq[m].w, acc[m][rt], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __syncthreads();   // one image: every wave has finished reading this layer's input before it is overwritten
            const int ncol_keep = last ? a.H : a.L[l + 1].Kp;   // columns the next layer reads (>= N, zero padded) / the H code bits
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int ct = wave + m * NW;
                const int col = ct * 16 + l16;
                if (ct < ct16 && col < (last ? 32 : ncol_keep)) {
                    const float bias = Bp[col];
                    if (last) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int i = 0; i < 4; ++i) out[(rt * 16 + 4 * g + i) * 33 + col] = acc[m][rt][i] + bias;   // z, natural column order
                    } else {
                        const int pc = pos(col);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {   // bias + ReLU (encoders.py:19-20); C/D map: col = lane & 15, row = 4 * (lane >> 4) + i
                                const float v = acc[m][rt][i] + bias;
                                out[(size_t)(rt * 16 + 4 * g + i) * S + pc] = v > 0.0f ? v : 0.0f;
                            }
                    }
                }
            }
            __syncthreads();
            ENC_STAMP(2 + (l < 4 ? l : 4));
        } else if (!last) {
            const int NT = L.Np >> 5;
            const int ncol_keep = a.L[l + 1].Kp;  // columns the next layer reads (>= N, zero padded)
            f32x16 accs[MT][RT];
            auto write_back = [&](int nt0, const f32x16 (&acc)[RT]) {
                // bias + ReLU (encoders.py:19-20), C/D map: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
                const int col = nt0 * 32 + lr;
                if (col < ncol_keep) {
                    const float bias = Bp[col];
                    const int pc = pos(col);
                    // opaque copy of the row stride: with the plain S the compiler hoists all 16*RT store addresses out of the
                    // LAYER loop (they are invariant in SINGLE mode) and spills them -- 32 scratch round trips per write-back
                    int S = a.S;
                    asm volatile("" : "+s"(S));
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int row = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                            float v = acc[rt][i] + bias;
                            out[(size_t)row * S + pc] = v > 0.0f ? v : 0.0f;
                        }
                }
            };
#pragma unroll
            for (int m = 0; m < (SINGLE ? MT : 1); ++m)
            for (int nt0 = wave + m * NW; nt0 < (SINGLE ? min(NT, wave + m * NW + 1) : NT); nt0 += NW) {
                f32x16 (&acc)[RT] = accs[m];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[rt][i] = 0.0f;
                const float4 *w0 = Wp + (size_t)nt0 * nch * 64 + lane;
                const float *arow = in + (size_t)lr * S + 4 * lh;
                // B fragments ride a ring of four registers, fetched THREE k-chunks (24 MFMAs) ahead of their use
                // (r02: a ring of eight / seven chunks ahead measured 2.25 ms per 1M rows and 49.7 us per 10k queries against
                // 2.2 ms / 44 us -- the ring is not what the MFMA pipe waits for; a 64-row SINGLE image for index builds needs
                // 172 VGPRs, so two workgroups per CU only fit with 42 spilled registers: 2.41 ms; the next layer's first three
                // fragments requested before the write-back and barrier of the current one: 2.23 ms / 42.5 us, noise).
                // The ring is indexed statically (main loop unrolled by 4, branch-free: indices past the end are
                // clamped and their data unused), so neither a register rotation nor a control-flow join makes
                // the compiler wait for the youngest load; A fragments (LDS) are fetched one chunk ahead.
                float4 B[4];
#pragma unroll
                for (int j = 0; j < 3; ++j) B[j] = w0[(size_t)min(j, nch - 1) * 64];
                float4 an[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) an[rt] = *reinterpret_cast<const float4 *>(arow + (size_t)rt * 32 * S);
                const int nch4 = nch & ~3;
                for (int c = 0; c < nch4; c += 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cc = c + j;
                        B[(j + 3) & 3] = w0[(size_t)min(cc + 3, nch - 1) * 64];
                        float4 av[RT];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) av[rt] = an[rt];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            an[rt] = *reinterpret_cast<const float4 *>(arow + (size_t)rt * 32 * S + min(cc + 1, nch - 1) * 8);
                        // keep the fetches HERE: the machine scheduler otherwise sinks them next to their use
                        // (seen: load; s_waitcnt vmcnt(0); mfma -- an exposed L2 round trip every four chunks)
                        __builtin_amdgcn_sched_barrier(0);
                        const float bb0[4] = {B[j].x, B[j].y, B[j].z, B[j].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) {
                                const float aa = i == 0 ? av[rt].x : i == 1 ? av[rt].y : i == 2 ? av[rt].z : av[rt].w;
                                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, bb0[i], acc[rt], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {  // the nch % 4 trailing chunks sit in ring slots 0..2
                    const int cc = nch4 + j;
                    if (cc < nch) {
                        float4 av[RT];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) av[rt] = an[rt];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            an[rt] = *reinterpret_cast<const float4 *>(arow + (size_t)rt * 32 * S + min(cc + 1, nch - 1) * 8);
                        const float bb0[4] = {B[j].x, B[j].y, B[j].z, B[j].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) {
                                const float aa = i == 0 ? av[rt].x : i == 1 ? av[rt].y : i == 2 ? av[rt].z : av[rt].w;
                                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, bb0[i], acc[rt], 0, 0, 0);
                            }
                        }
                    }
                }
                if (!SINGLE) write_back(nt0, acc);
            }
            if (SINGLE) {
#ifdef NLSH_ENC_FINE
                if (l == 1) ENC_STAMP(4);
#endif
                __syncthreads();  // every wave has finished reading this layer's input: the image may be overwritten
#ifdef NLSH_ENC_FINE
                if (l == 1) ENC_STAMP(5);
#endif
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (wave + m * NW < NT) write_back(wave + m * NW, accs[m]);
#ifdef NLSH_ENC_FINE
                if (l == 1) ENC_STAMP(6);
#endif
            }
            __syncthreads();
            if (!SINGLE) { float *t = in; in = out; out = t; }
            ENC_STAMP(2 + (l < 4 ? l : 4));
        } else {
            // output layer: H <= 32 columns.  r01 ran it as ONE 32x32x2 column tile per 32-row tile on `RT` of the NW waves
            // (5 us of a 37-us workgroup with six of eight waves idle and, at H = 16, half of every MFMA's columns
            // padding).  r02: 16x16x4 tiles -- (M / 16) row tiles x (H / 16) column tiles dealt over ALL waves; same
            // k-ascending fmaf chain (guide, FP32-input MFMA), so z keeps its bits.  Lane l holds A[row l & 15][k = l >> 4] and
            // B[k = l >> 4][col l & 15]; with the even/odd de-interleaved LDS rows the two A values of an 8-wide k chunk sit
            // two floats apart (pos(8c + g) and pos(8c + 4 + g)); B is packed per 16 k (4 MFMA steps) as one float4 per lane.
            {
                typedef float f32x4v __attribute__((ext_vector_type(4)));
                const int ct16 = (a.H + 15) >> 4, ngr = (L.K + 15) >> 4;
                const int T = (M >> 4) * ct16;
                const int g = lane >> 4, l16 = lane & 15;
                const int pa = ((g & 1) << 2) + (g >> 1);   // pos(g) inside a chunk; pos(4 + g) = pa + 2
                constexpr int MAXT = (M / 16 * 2 + NW - 1) / NW;   // tiles a wave may own (H <= 32: at most two column tiles)
                f32x4v accs[MAXT];
#pragma unroll
                for (int m = 0; m < MAXT; ++m) {
                    const int t = wave + m * NW;
                    const int rt = t / ct16, ct = t - rt * ct16;
                    f32x4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (t < T) {
                        const float4 *w0 = Wp + (size_t)ct * ngr * 64 + lane;
                        const float *arow = in + (size_t)(rt * 16 + l16) * S + pa;
                        constexpr int OR = 8;   // B ring: groups of 16 k in flight (each is only 4 short MFMAs of cover)
                        float4 B[OR];
#pragma unroll
                        for (int j = 0; j < OR; ++j) B[j] = w0[(size_t)min(j, ngr - 1) * 64];
                        for (int j0 = 0; j0 < ngr; j0 += OR) {
#pragma unroll
                            for (int jj = 0; jj < OR; ++jj) {
                                const int j = j0 + jj;
                                if (j < ngr) {
                                    const float4 bq = B[jj];
                                    if (j + OR < ngr) B[jj] = w0[(size_t)(j + OR) * 64];
                                    const int c0 = 2 * j, c1 = 2 * j + 1;
                                    const float a0 = arow[c0 * 8], a1 = arow[c0 * 8 + 2];
                                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bq.x, acc, 0, 0, 0);
                                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bq.y, acc, 0, 0, 0);
                                    if (c1 < nch) {   // Kp is a multiple of 8, not of 16: the last group may hold one chunk only
                                        const float a2 = arow[c1 * 8], a3 = arow[c1 * 8 + 2];
                                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bq.z, acc, 0, 0, 0);
                                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bq.w, acc, 0, 0, 0);
                                    }
                                }
                            }
                        }
                    }
                    accs[m] = acc;
                }
                if (SINGLE) __syncthreads();   // one image: z lands on activation rows other waves were still reading
#pragma unroll
                for (int m = 0; m < MAXT; ++m) {
                    const int t = wave + m * NW;
                    if (t < T) {
                        const int rt = t / ct16, ct = t - rt * ct16;
                        const int col = ct * 16 + l16;
                        const float bias = Bp[col];
#pragma unroll
                        for (int i = 0; i < 4; ++i) out[(rt * 16 + 4 * g + i) * 33 + col] = accs[m][i] + bias;  // z, natural column order
                    }
                }
            }
            __syncthreads();
        }
    }

    ENC_STAMP(7);
    // ---- epilogue.  `out` holds z [M][33]; `in` is free.
    const int H = a.H;
    const int NP = a.n_probes;
    float *zbuf = out;
    float *pbuf = SINGLE ? smem + M * 33 : in;  // Bernoulli probability [M][33]
    // [M][n_probes] key table: over z (dead now) in the ping-pong form, behind z and p in the single image; behind it one 64-bit
    // first-occurrence mask per row and (lookup fused, scan_plan.h) the coarse key table of the bucket search
    int32_t *kbuf = reinterpret_cast<int32_t *>(SINGLE ? smem + 2 * M * 33 : out);
    unsigned long long *fmask = reinterpret_cast<unsigned long long *>(kbuf + M * NP);
    int32_t *coarse = reinterpret_cast<int32_t *>(fmask + M);
    int32_t *cbuf = reinterpret_cast<int32_t *>(SINGLE ? smem : in);   // the rows' distinct keys, compacted (over z / p, dead by then)
    __shared__ int plan_hits, plan_viol;
    // r06: the bucket lookup of the scan's PLAN phase rides here when the caller asks for it (pa.enabled; wave-uniform): its coarse
    // key table is requested now, lands in registers under the sigmoid pass and in LDS under the key pass
    int32_t cpre[2] = {0, 0};
    if (pa.enabled) {
        if (tid < pa.nco) cpre[0] = pa.uniq[(long long)tid * pa.stride];
        if (tid + NTH < pa.nco) cpre[1] = pa.uniq[(long long)(tid + NTH) * pa.stride];
        if (tid == 0) { plan_hits = 0; plan_viol = 0; }
        if (blk == 0) plan_batch_init(pa, tid, NTH);
        if (pa.prep_metric >= 0)   // the tiled schedule's padded / pre-normalised query copy of this workgroup's rows
            for (int r = wave; r < M; r += NW)
                if (row_base + r < a.n) prep_query(pa, row_base + r, lane);
    }
    for (int e = tid; e < M * H; e += NTH) {
        int r = e / H, h = e - r * H;
        long long grow = row_base + r;
        float z = zbuf[r * 33 + h];
        float raw, p;
        if (a.act == NLSH_ACT_SIGMOID) {  // hashings.py:26
            raw = 1.0f / (1.0f + expf(-z));
            p = raw;
        } else {  // hashings.py:24 and :68-69
            raw = tanhf(z);
            p = raw / 2.0f + 0.5f;
        }
        pbuf[r * 33 + h] = p;
        if (grow < a.n) {
            if (a.z_out) a.z_out[grow * H + h] = z;
            if (a.probs_out) a.probs_out[grow * H + h] = raw;
        }
    }
    __syncthreads();
    ENC_STAMP(8);
    // (r02 measured a parallel form of the key pass -- a thread per (row, probe, 4-bit word) OR-ing its bits into the code -- at
    // 38.8 us per 10k queries against 37.5 us: four more passes and barriers cost what the idle threads had cost.)
    for (int e = tid; e < M * NP; e += NTH) {
        int r = e / NP, j = e - r * NP;
        long long grow = row_base + r;
        if (grow >= a.n || (j > 0 && grow >= a.n_multi_rows)) continue;
        const unsigned long long gidx = (unsigned long long)(a.row0 + grow);
        uint32_t code = 0;
        uint32_t rnd[4];
        for (int h = 0; h < H; ++h) {
            const float p = pbuf[r * 33 + h];
            int bit;
            if (j == 0) {
                bit = p > 0.5f;  // hashings.py:72: strict, on the probability
            } else {
                if ((h & 3) == 0) philox4x32_10(a.seed, (uint32_t)gidx, (uint32_t)(gidx >> 32), (uint32_t)j, (uint32_t)(h >> 2), rnd);
                const float u = (float)(rnd[h & 3] >> 8) * (1.0f / 16777216.0f);
                bit = u < p;  // Bernoulli(p) draw, hashings.py:80
            }
            code = (code << 1) | (uint32_t)bit;  // MSB-first, utils.pyx:12-14
        }
        int32_t key = a.key_mode == NLSH_KEY_REF_INT16 ? (int32_t)(int16_t)(uint16_t)(code & 0xFFFFu) : (int32_t)code;
        kbuf[r * NP + j] = key;
        if (j == 0 && a.code_out) a.code_out[grow] = code;
    }
    if (NP <= 64) {
        if (tid < M) fmask[tid] = 0ull;
        if (pa.enabled) {   // (z is dead: in the ping-pong form the table lies over it)
            if (tid < pa.nco) coarse[tid] = cpre[0];
            if (tid + NTH < pa.nco) coarse[tid + NTH] = cpre[1];
        }
    }
    __syncthreads();
    ENC_STAMP(9);
    if (NP <= 64) {
        // Set semantics (utils.pyx:26-31), first-occurrence order, a thread per (row, probe) (r01-r05: one thread per ROW walked its
        // probes with a nested loop -- 32 of 512 threads for 3.7 us of a 28-us workgroup): pass 1 marks the first occurrences in the
        // row's 64-bit mask, pass 2 moves every first occurrence to its rank among them, pass 3 stores the row -- and, fused, looks
        // every distinct key up (scan_plan.h).  Same table as the serial form, bit for bit.
        for (int e = tid; e < M * NP; e += NTH) {
            const int r = e / NP, j = e - r * NP;
            const long long grow = row_base + r;
            if (grow >= a.n || (j > 0 && grow >= a.n_multi_rows)) continue;
            const int32_t key = kbuf[e];
            bool first = true;
            for (int t = 0; t < j; ++t) first &= kbuf[r * NP + t] != key;
            if (first) atomicOr(&fmask[r], 1ull << j);
        }
        __syncthreads();
        for (int e = tid; e < M * NP; e += NTH) {
            const int r = e / NP, j = e - r * NP;
            const long long grow = row_base + r;
            if (grow >= a.n || (j > 0 && grow >= a.n_multi_rows)) continue;
            const unsigned long long m = fmask[r];
            if ((m >> j) & 1ull) cbuf[r * NP + __popcll(m & ((1ull << j) - 1ull))] = kbuf[e];
        }
        __syncthreads();
        ENC_STAMP(10);
        int nhit = 0, viol = 0;
        for (int e = tid; e < M * NP; e += NTH) {
            const int r = e / NP, j = e - r * NP;
            const long long grow = row_base + r;
            if (grow >= a.n) continue;
            const int cnt = __popcll(fmask[r]);
            const int32_t key = j < cnt ? cbuf[r * NP + j] : 0;
            a.keys_out[grow * NP + j] = key;
            if (j == 0) a.nkeys_out[grow] = cnt;
            if (pa.enabled) {
                if (j == 0) pa.tauq[grow] = KEY_NONE;        // running bound of the query
                nhit += plan_pair(pa, coarse, grow * NP + j, key, j < cnt, viol) ? 1 : 0;
            }
        }
        if (pa.enabled) {
            // pairs this workgroup added to the counters (| violations): bscan_kernel holds the counters' sum against the sum of these
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { nhit += __shfl_xor(nhit, m); viol |= __shfl_xor(viol, m); }
            if (lane == 0 && (nhit | viol)) { atomicAdd(&plan_hits, nhit); atomicOr(&plan_viol, viol); }
            __syncthreads();
            if (tid == 0) pa.hits[blk] = plan_hits | plan_viol;
        }
    } else if (tid < M) {   // more than 64 probes (eval.py:148 sweeps to 100; never scanned in one call): the serial form
        const int r = tid;
        long long grow = row_base + r;
        if (grow < a.n) {
            const int npr = grow < a.n_multi_rows ? NP : 1;
            int cnt = 0;
            for (int j = 0; j < npr; ++j) {  // set semantics (utils.pyx:26-31), first-occurrence order
                int32_t key = kbuf[r * NP + j];
                bool dup = false;
                for (int t = 0; t < cnt; ++t) dup |= (kbuf[r * NP + t] == key);
                if (!dup) kbuf[r * NP + cnt++] = key;
            }
            for (int t = 0; t < NP; ++t) a.keys_out[grow * NP + t] = t < cnt ? kbuf[r * NP + t] : 0;
            a.nkeys_out[grow] = cnt;
        }
    }
#ifdef NLSH_ENC_TRACE
    __syncthreads();
    ENC_STAMP(11);
    if (tid == 0 && a.z_out && row_base + M <= a.n) {
        for (int i = 0; i <= 9; ++i) a.z_out[row_base * H + i] = (float)(stamp[i] - stamp[0]);
        a.z_out[row_base * H + 10] = (float)(stamp[11] - stamp[0]);
        a.z_out[row_base * H + 11] = (float)(__builtin_amdgcn_s_memtime() - core0);
    }
    if (tid == 0 && blk < NLSH_ENC_TRACE_SLOTS) {
        for (int i = 0; i <= 11; ++i) g_enc_trace[blk * 16 + i] = (float)(stamp[i] - stamp[0]);
        g_enc_trace[blk * 16 + 12] = (float)(stamp[0] & 0xFFFFFFull);
        g_enc_trace[blk * 16 + 13] = (float)M;
    }
#endif
}

template <int RT, int NW, bool SINGLE, int MT = 1, int WPE = 1, bool H16 = false>  // WPE: waves per SIMD the registers must allow
__global__ __launch_bounds__(NW * 64, WPE) void encode_hash_kernel(EncArgs a, PlanArgs pa) {
    extern __shared__ float4 smem4[];
    encode_hash_body<RT, NW, SINGLE, MT, H16>(a, pa, reinterpret_cast<float *>(smem4), (long long)blockIdx.x * (H16 ? 16 * RT : 32 * RT), blockIdx.x);
}

// r06, the balanced query batch: 313 32-row workgroups on 256 CUs left 57 CUs with two of them, and the launch lasts as long as its
// busiest CU (8192 rows 27 us, 8193 rows 36 us).  Here the rows of the batch's last, partial round of 32-row workgroups -- at most half
// a round -- are dealt as 16-ROW workgroups of the all-16x16x4 form IN THE SAME GRID: workgroups [0, n32) take 32 rows each, the rest
// 16 each, so the busiest CUs host 32 + 16 rows instead of 32 + 32 (the layers are bound by the CU's MFMA pipes: 3.7 + 7.4 us of
// pipe time per 32-row workgroup at width 256, half of that per 16-row one).  Under the reference's batching rule (F6) the tail rows
// are also the single-probe ones: no Philox draws, one lookup per row.  Same arithmetic per row in both bodies, so the split changes
// no bit; the register allocation is the larger of the two bodies (87 VGPRs: two workgroups per CU, which is all the launch wants).
__global__ __launch_bounds__(512, 1) void encode_hash_het_kernel(EncArgs a, PlanArgs pa, unsigned n32) {
    extern __shared__ float4 smem4[];
    if (blockIdx.x < n32) {
        encode_hash_body<1, 8, true, 1, false>(a, pa, reinterpret_cast<float *>(smem4), (long long)blockIdx.x * 32, blockIdx.x);
    } else {
        // A 16-row workgroup shares its CU with a 32-row one whose waves are older: at equal priority they keep the MFMA pipes, the
        // 16-row workgroup's first layer ends when the other's SECOND does (traced: 19 us for 1.9 us of pipe time) and its serial
        // tail -- output layer, key pass, lookup: 11 us -- then runs alone at the end of the launch.  With the higher issue priority
        // the short workgroup goes first and the long one's layers absorb the delay (tools/enc_step_trace.py).
#if NLSH_ENC_HET_PRIO >= 0
        __builtin_amdgcn_s_setprio(NLSH_ENC_HET_PRIO);
#endif
        encode_hash_body<1, 8, true, 2, true>(a, pa, reinterpret_cast<float *>(smem4), (long long)n32 * 32 + (long long)(blockIdx.x - n32) * 16, blockIdx.x);
    }
}

// packed[w_off + ((nt*nch + c)*64 + lane)*4 + i] = W[nt*32 + (lane&31)][8c + 2i + (lane>>5)]
__global__ void pack_weights_kernel(const float *W, const float *b, int K, int N, int Kp, int Np, float *wdst, float *bdst) {
    const long long total = (long long)Np * Kp;
    const int nch = Kp >> 3;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int i = (int)(e & 3);
        int lane = (int)((e >> 2) & 63);
        long long rest = e >> 8;
        int c = (int)(rest % nch);
        int nt = (int)(rest / nch);
        int col = nt * 32 + (lane & 31);
        int k = 8 * c + 2 * i + (lane >> 5);
        wdst[e] = (col < N && k < K) ? W[(size_t)col * K + k] : 0.0f;
    }
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < Np; e += (long long)gridDim.x * blockDim.x)
        bdst[e] = (b != nullptr && e < N) ? b[e] : 0.0f;
}

// output layer, 16x16x4 order: packed[((ct*ngr + j)*64 + lane)*4 + s] = W[ct*16 + (lane&15)][16j + 4s + (lane>>4)]
__global__ void pack_out_weights_kernel(const float *W, const float *b, int K, int N, int ngr, int ct16, int Np, float *wdst, float *bdst) {
    const long long total = (long long)ct16 * ngr * 256;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int st = (int)(e & 3), lane = (int)((e >> 2) & 63);
        const long long rest = e >> 8;
        const int j = (int)(rest % ngr), ct = (int)(rest / ngr);
        const int col = ct * 16 + (lane & 15), k = 16 * j + 4 * st + (lane >> 4);
        wdst[e] = (col < N && k < K) ? W[(size_t)col * K + k] : 0.0f;
    }
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < Np; e += (long long)gridDim.x * blockDim.x)
        bdst[e] = (b != nullptr && e < N) ? b[e] : 0.0f;
}

// nlsh/utils.pyx:6-15: out = (out << 1) | bit over H bits, returned as int16 (or untruncated)
__global__ void pack_codes_kernel(const int32_t *codes, long long total, int H, int key_mode, int32_t *keys) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int32_t *p = codes + e * H;
        uint32_t out = 0;
        for (int h = 0; h < H; ++h) out = (out << 1) | (uint32_t)p[h];
        keys[e] = key_mode == NLSH_KEY_REF_INT16 ? (int32_t)(int16_t)(uint16_t)(out & 0xFFFFu) : (int32_t)out;
    }
}

static int fill_layers(int n_layers, const int *dims, LayerDesc *L, long long *total) {
    long long off = 0;
    for (int l = 0; l < n_layers; ++l) {
        L[l].K = dims[l];
        L[l].N = dims[l + 1];
        L[l].Kp = round_up(dims[l], 8);
        L[l].Np = round_up(dims[l + 1], 32);
        L[l].w_off = off;
        // hidden layers: [Np/32 column tiles][Kp/8 chunks][64 lanes][4]; the output layer (l == n_layers - 1) is packed for
        // 16x16x4 tiles: [ceil(N/16)][ceil(K/16) groups][64][4], which can exceed Np * Kp by one k group
        const long long hidden = (long long)L[l].Np * L[l].Kp, outl = (long long)((dims[l + 1] + 15) / 16) * ((dims[l] + 15) / 16) * 256;
        off += (l + 1 == n_layers && outl > hidden) ? outl : hidden;
        L[l].b_off = off;
        off += L[l].Np;
    }
    for (int l = 0; l < n_layers; ++l) {   // the 16x16x4 packing of the hidden layers (the 16-row query-batch form), behind everything else
        if (l + 1 == n_layers) { L[l].w16_off = L[l].w_off; continue; }
        L[l].w16_off = off;
        off += (long long)(L[l].Np / 16) * ((dims[l] + 15) / 16) * 256;
    }
    *total = off;
    return 0;
}

static int check_dims(int n_layers, const int *dims) {
    NLSH_REQUIRE(dims != nullptr && n_layers >= 1 && n_layers <= NLSH_MAX_LAYERS, NLSH_E_INVALID,
                 "encoder: n_layers=%d out of range [1,%d]", n_layers, NLSH_MAX_LAYERS);
    for (int l = 0; l <= n_layers; ++l) NLSH_REQUIRE(dims[l] >= 1, NLSH_E_INVALID, "encoder: dims[%d]=%d", l, dims[l]);
    for (int l = 0; l < n_layers; ++l)
        NLSH_REQUIRE(dims[l] <= (l == 0 ? NLSH_MAX_DIM : NLSH_MAX_WIDTH), NLSH_E_UNSUPPORTED,
                     "encoder: layer input width %d > %d (LDS-resident MLP)", dims[l], l == 0 ? NLSH_MAX_DIM : NLSH_MAX_WIDTH);
    NLSH_REQUIRE(dims[n_layers] <= NLSH_MAX_HASH_BITS, NLSH_E_UNSUPPORTED, "encoder: hash_size %d > %d", dims[n_layers], NLSH_MAX_HASH_BITS);
    return NLSH_OK;
}

}  // namespace nlsh

using namespace nlsh;

extern "C" int64_t nlsh_encoder_packed_floats(int n_layers, const int *dims) {
    if (check_dims(n_layers, dims) != NLSH_OK) return -1;
    LayerDesc L[NLSH_MAX_LAYERS];
    long long total;
    fill_layers(n_layers, dims, L, &total);
    return total;
}

extern "C" int nlsh_encoder_pack(int n_layers, const int *dims, const float *const *W, const float *const *b,
                                 float *packed, nlsh_stream_t stream) {
    int rc = check_dims(n_layers, dims);
    if (rc != NLSH_OK) return rc;
    NLSH_REQUIRE(W != nullptr && b != nullptr && packed != nullptr, NLSH_E_INVALID, "encoder_pack: null pointer");
    LayerDesc L[NLSH_MAX_LAYERS];
    long long total;
    fill_layers(n_layers, dims, L, &total);
    hipStream_t s = (hipStream_t)stream;
    for (int l = 0; l < n_layers; ++l) {
        NLSH_REQUIRE(W[l] != nullptr, NLSH_E_INVALID, "encoder_pack: W[%d] is null", l);
        long long tot = (long long)L[l].Np * L[l].Kp;
        int grid = (int)((tot + 255) / 256);
        if (grid > 4096) grid = 4096;
        if (l + 1 == n_layers)
            hipLaunchKernelGGL(pack_out_weights_kernel, dim3(grid), dim3(256), 0, s, W[l], b[l], L[l].K, L[l].N, (L[l].K + 15) / 16,
                               (L[l].N + 15) / 16, L[l].Np, packed + L[l].w_off, packed + L[l].b_off);
        else {
            hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, s, W[l], b[l], L[l].K, L[l].N, L[l].Kp, L[l].Np,
                               packed + L[l].w_off, packed + L[l].b_off);
            // second copy for the 16-row form: Np / 16 column tiles (columns >= N zero), same bias vector
            hipLaunchKernelGGL(pack_out_weights_kernel, dim3(grid), dim3(256), 0, s, W[l], b[l], L[l].K, L[l].N, (L[l].K + 15) / 16,
                               L[l].Np / 16, L[l].Np, packed + L[l].w16_off, packed + L[l].b_off);
        }
        NLSH_CHECK_HIP(hipGetLastError());
    }
    return NLSH_OK;
}

extern "C" int nlsh_pack_codes(const int32_t *codes, int64_t B, int n, int H, int key_mode, int32_t *keys_out,
                               nlsh_stream_t stream) {
    NLSH_REQUIRE(B >= 0 && n >= 0 && H >= 1 && H <= NLSH_MAX_HASH_BITS, NLSH_E_INVALID, "pack_codes: B=%lld n=%d H=%d", (long long)B, n, H);
    NLSH_REQUIRE(key_mode == NLSH_KEY_REF_INT16 || key_mode == NLSH_KEY_FULL, NLSH_E_INVALID, "pack_codes: key_mode=%d", key_mode);
    const long long total = (long long)B * n;
    if (total == 0) return NLSH_OK;
    NLSH_REQUIRE(codes && keys_out, NLSH_E_INVALID, "pack_codes: null pointer");
    long long grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_codes_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, codes, total, H, key_mode, keys_out);
    NLSH_CHECK_HIP(hipGetLastError());
    return NLSH_OK;
}

namespace nlsh {

// compute units of the current device (256 on MI355X), asked once
static int device_cus() {
    static std::atomic<int> cached{0};
    int v = cached.load(std::memory_order_relaxed);
    if (v > 0) return v;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        n = 256;     // no device visible (a CPU-only build check): the part this library is built for
    }
    cached.store(n, std::memory_order_relaxed);
    return n;
}

int encode_plan_fill(EncPlan &p, int64_t n, int n_layers, const int *dims, const float *packed, int act, int key_mode, int n_probes,
                     int64_t n_multi_rows, int64_t row0, float *z_out, float *probs_out, uint32_t *code_out, int32_t *keys_out,
                     int32_t *nkeys_out) {
    int rc = check_dims(n_layers, dims);
    if (rc != NLSH_OK) return rc;
    NLSH_REQUIRE(n >= 0, NLSH_E_INVALID, "encode_hash: n=%lld", (long long)n);
    NLSH_REQUIRE(n == 0 || (packed && keys_out && nkeys_out), NLSH_E_INVALID, "encode_hash: null pointer");
    NLSH_REQUIRE(act == NLSH_ACT_SIGMOID || act == NLSH_ACT_TANH, NLSH_E_INVALID, "encode_hash: act=%d", act);
    NLSH_REQUIRE(key_mode == NLSH_KEY_REF_INT16 || key_mode == NLSH_KEY_FULL, NLSH_E_INVALID, "encode_hash: key_mode=%d", key_mode);
    // hashings.py:83: "`n` should be positive integer"
    NLSH_REQUIRE(n_probes >= 1 && n_probes <= NLSH_MAX_ENCODE_PROBES, NLSH_E_INVALID, "encode_hash: n_probes=%d not in [1,%d]", n_probes, NLSH_MAX_ENCODE_PROBES);

    EncArgs &a = p.a;
    a.x = nullptr; a.n = n; a.x_stride = 0; a.n_layers = n_layers; a.packed = packed;
    long long total;
    fill_layers(n_layers, dims, a.L, &total);
    int maxKp = 64;
    for (int l = 0; l < n_layers; ++l) if (a.L[l].Kp > maxKp) maxKp = a.L[l].Kp;
    if (round_up(n_probes, 8) > maxKp) maxKp = round_up(n_probes, 8);  // the per-row key table [M][n_probes] reuses an activation image
    a.S = maxKp + 4;  // S/4 odd -> conflict-free ds_read_b128 of A fragments
    a.H = dims[n_layers]; a.act = act; a.key_mode = key_mode; a.n_probes = n_probes;
    a.n_multi_rows = n_multi_rows; a.row0 = row0; a.seed = 0;
    a.z_out = z_out; a.probs_out = probs_out; a.code_out = code_out; a.keys_out = keys_out; a.nkeys_out = nkeys_out;
    p.form = ENC_FORM_SINGLE; p.grid = 0; p.lds = 0; p.n32 = 0;
    if (n == 0) return NLSH_OK;

    const size_t lds_limit = ENC_LDS_LIMIT;
    int max_np = 0;  // widest hidden layer decides how many column tiles a wave owns
    for (int l = 0; l + 1 < n_layers; ++l) if (a.L[l].Np > max_np) max_np = a.L[l].Np;
    // Query-sized batches (and encoders too wide for two 64-row images) run 32-row workgroups on a SINGLE LDS image;
    // its rows must also hold the epilogue's z, p and key tables side by side (66 + n_probes floats per row).
    const bool single = n <= NLSH_ENC_SINGLE_MAX_ROWS || (size_t)2 * 64 * a.S * 4 > lds_limit;
    if (single) {
        if (72 + round_up(n_probes, 8) > maxKp) a.S = 72 + round_up(n_probes, 8) + 4;
        p.lds = (size_t)32 * a.S * 4;
        NLSH_REQUIRE(p.lds <= lds_limit, NLSH_E_UNSUPPORTED, "encode_hash: width %d needs %zu B of LDS", maxKp, p.lds);
        p.grid = (unsigned)((n + 31) / 32);
        const bool h16_ok = max_np <= 16 * 8 * 2;   // two 16-column tiles per wave: wider encoders would need five = 175 VGPRs and stay on the 32-row form
        if (n <= NLSH_ENC_H16_MAX_ROWS && h16_ok) {
            // 16-row workgroups, every layer as 16x16x4 tiles (78 VGPRs, three workgroups per CU)
            p.form = ENC_FORM_H16; p.lds = (size_t)16 * a.S * 4; p.grid = (unsigned)((n + 15) / 16);
        } else if (max_np <= 32 * 8) {
            p.form = ENC_FORM_SINGLE;
            // the rows behind the last FULL round of 32-row workgroups (one per CU), when they are at most half a round, as 16-row
            // workgroups of the same launch (encode_hash_het_kernel): 10^4 rows on 256 CUs = 256 x 32 + 113 x 16
            const long long round_rows = 32ll * device_cus(), rem = n % round_rows;
            if (NLSH_ENC_HET && h16_ok && n > round_rows && rem > 0 && rem <= round_rows / 2) {
                p.form = ENC_FORM_HET;
                p.n32 = (unsigned)((n - rem) / 32);
                p.grid = p.n32 + (unsigned)((rem + 15) / 16);
            }
        } else {
            p.form = ENC_FORM_SINGLE_WIDE;
        }
    } else {
        if (72 + round_up(n_probes, 8) > maxKp) a.S = 72 + round_up(n_probes, 8) + 4;
        const size_t lds128 = (size_t)128 * a.S * 4;
        // r02 also measured two ways of putting a second workgroup on the CU so that one's staging, write-backs and epilogue run
        // under the other's MFMAs: two 64-row single-image workgroups of eight waves (128 VGPRs, spills outside the k loop): 1.96 ms
        // per 1M rows; of four waves owning two column tiles each (201 VGPRs): 1.92 ms at an in-kernel clock of 2.23 GHz -- against
        // 1.895 ms at 2.16 GHz for the form below.  The phases did overlap (tools/enc_trace.py) and the chip gave the gain back as
        // clock: clock x MFMA duty stayed put.
        if (NLSH_ENC_BUILD_128 && max_np <= 32 * 8 && lds128 <= lds_limit) {
            // index builds: 128 rows per workgroup on ONE image (accumulators of the four row tiles held in registers across the
            // layer barrier): a B fragment feeds 16 MFMAs instead of 8 and the per-layer fixed cost (write-back, barriers, ring
            // prologue: ~2.5-3.8 us) is paid once per 128 rows instead of once per 64
            p.form = ENC_FORM_BUILD128; p.lds = lds128; p.grid = (unsigned)((n + 127) / 128);
        } else {
            a.S = maxKp + 4;
            p.form = ENC_FORM_PINGPONG; p.lds = (size_t)2 * 64 * a.S * 4; p.grid = (unsigned)((n + 63) / 64);
        }
    }
    return NLSH_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel form) and process instead of once per launch (r04: a
// runtime call of its own in front of every encode, on a path whose pipelined step is bound by the host's enqueue time): the
// attribute is a permission, so it is raised to the whole 160 KB the first time a form runs on a device.
static int allow_lds(int form, const void *fn) {
    static std::atomic<unsigned> done[16];
    int dev = 0;
    NLSH_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 16 && (done[dev].load(std::memory_order_acquire) >> form) & 1u) return NLSH_OK;
    NLSH_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ENC_LDS_LIMIT));
    if (dev >= 0 && dev < 16) done[dev].fetch_or(1u << form, std::memory_order_release);
    return NLSH_OK;
}

// rows per workgroup of a kernel form
static int form_rows(int form) {
    return (form == ENC_FORM_H16 || form == ENC_FORM_HET) ? 16 : form == ENC_FORM_BUILD128 ? 128 : form == ENC_FORM_PINGPONG ? 64 : 32;   // HET: its smaller workgroups
}

// The bucket lookup of the scan's PLAN phase in this launch's epilogue (scan_plan.h): sizes the coarse key table for the LDS the form
// has behind its key table -- growing the rows' stride when an encoder is so narrow that fewer than 256 entries would fit -- and
// returns the workgroups of the launch (= entries of `hits` the look-back verdict of bscan_kernel sums).
int encode_plan_fuse_lookup(EncPlan &p, PlanArgs &pa) {
    NLSH_REQUIRE(p.a.n == pa.Q && p.a.n_probes == pa.P, NLSH_E_INVALID, "encode_hash + lookup: the encode makes a [%lld, %d] key table, the scan expects [%lld, %d]",
                 (long long)p.a.n, p.a.n_probes, (long long)pa.Q, pa.P);
    NLSH_REQUIRE(pa.P <= 64, NLSH_E_UNSUPPORTED, "encode_hash + lookup: n_probes=%d > 64", pa.P);
    if (p.a.n == 0) return NLSH_OK;
    const int M = form_rows(p.form), NP = p.a.n_probes;
    const bool pingpong = p.form == ENC_FORM_PINGPONG;
    const int used = (pingpong ? 0 : 2 * 33) + NP + 2;            // floats per row in front of the coarse table (z, p | keys, mask)
    int cap = M * (p.a.S - used);
    const int want = pa.nb < 256 ? (pa.nb > 0 ? pa.nb : 1) : 256;
    if (cap < want) {
        p.a.S = round_up(used + (want + M - 1) / M, 8) + 4;       // S / 4 stays odd (conflict-free A-fragment reads)
        p.lds = (size_t)(pingpong ? 2 : 1) * (p.form == ENC_FORM_HET ? 32 : M) * p.a.S * 4;
        NLSH_REQUIRE(p.lds <= ENC_LDS_LIMIT, NLSH_E_UNSUPPORTED, "encode_hash + lookup: %zu B of LDS", p.lds);
        cap = M * (p.a.S - used);
    }
    plan_coarse(pa, cap < 1024 ? cap : 1024);                      // two entries per thread of the 512 at most
    pa.enabled = 1;
    return NLSH_OK;
}

// the kernel of a form (one instantiation each)
static const void *form_kernel(int form) {
    switch (form) {
        case ENC_FORM_H16: return (const void *)encode_hash_kernel<1, 8, true, 2, 1, true>;
        case ENC_FORM_SINGLE: return (const void *)encode_hash_kernel<1, 8, true, 1>;
        case ENC_FORM_SINGLE_WIDE: return (const void *)encode_hash_kernel<1, 8, true, 3>;
        case ENC_FORM_BUILD128: return (const void *)encode_hash_kernel<4, 8, true, 1>;
        case ENC_FORM_PINGPONG: return (const void *)encode_hash_kernel<2, 8, false, 1>;
        case ENC_FORM_HET: return (const void *)encode_hash_het_kernel;
        default: return nullptr;
    }
}

int encode_plan_prepare(const EncPlan &p) {
    const void *fn = form_kernel(p.form);
    NLSH_REQUIRE(fn != nullptr, NLSH_E_INVALID, "encode_hash: form %d", p.form);
    return p.a.n == 0 ? NLSH_OK : allow_lds(p.form, fn);
}

int encode_plan_node(const EncPlan &p, const float *x, int64_t x_stride, uint64_t seed, const PlanArgs *lookup, EncNode *out) {
    NLSH_REQUIRE(p.a.n > 0 && x != nullptr && out != nullptr, NLSH_E_INVALID, "encode_hash: null pointer");
    NLSH_REQUIRE(x_stride >= p.a.L[0].K, NLSH_E_INVALID, "encode_hash: x_stride %lld < d %d", (long long)x_stride, p.a.L[0].K);
    out->a = p.a;
    out->a.x = x; out->a.x_stride = x_stride; out->a.seed = seed;
    if (lookup) {
        out->pa = *lookup;
        out->pa.queries = x; out->pa.q_stride = x_stride;   // the batch itself (the padded query copy is made from it)
    } else {
        out->pa = PlanArgs{};
        out->pa.enabled = 0; out->pa.prep_metric = -1;
    }
    out->n32 = p.n32;
    out->argv[0] = &out->a; out->argv[1] = &out->pa; out->argv[2] = &out->n32;
    out->p = hipKernelNodeParams{};
    out->p.func = const_cast<void *>(form_kernel(p.form));
    NLSH_REQUIRE(out->p.func != nullptr, NLSH_E_INVALID, "encode_hash: form %d", p.form);
    out->p.gridDim = dim3(p.grid); out->p.blockDim = dim3(512); out->p.sharedMemBytes = (unsigned)p.lds;
    out->p.kernelParams = out->argv; out->p.extra = nullptr;
    return NLSH_OK;
}

int encode_plan_launch(const EncPlan &p, const float *x, int64_t x_stride, uint64_t seed, hipStream_t s, const PlanArgs *lookup) {
    if (p.a.n == 0) return NLSH_OK;
    EncNode node;
    int rc = encode_plan_node(p, x, x_stride, seed, lookup, &node);
    if (rc == NLSH_OK) rc = encode_plan_prepare(p);
    if (rc != NLSH_OK) return rc;
    NLSH_CHECK_HIP(hipLaunchKernel(node.p.func, node.p.gridDim, node.p.blockDim, node.p.kernelParams, node.p.sharedMemBytes, s));
    return NLSH_OK;
}

}  // namespace nlsh

extern "C" int nlsh_encode_hash(const float *x, int64_t n, int64_t x_stride, int n_layers, const int *dims,
                                const float *packed, int act, int key_mode, int n_probes, int64_t n_multi_rows,
                                uint64_t seed, int64_t row0, float *z_out, float *probs_out, uint32_t *code_out,
                                int32_t *keys_out, int32_t *nkeys_out, nlsh_stream_t stream) {
    EncPlan p;
    int rc = encode_plan_fill(p, n, n_layers, dims, packed, act, key_mode, n_probes, n_multi_rows, row0, z_out, probs_out, code_out,
                              keys_out, nkeys_out);
    if (rc != NLSH_OK) return rc;
    return encode_plan_launch(p, x, x_stride, seed, (hipStream_t)stream);
}
