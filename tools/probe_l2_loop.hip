// Hardware probe: sustained VALU rate of the tiled scan's inner loop shape on gfx950 --
// per chunk: TPS row float4 (VGPR or LDS) x QW queries (SGPR, scalar loads) x 4 elements x (v_sub, v_add literal, v_fmac).
//   mode 0: math only (rows and queries loop-invariant registers)
//   mode 1: + one ds_read_b128 per tile and chunk
//   mode 2: + one s_load_dwordx4 per query and chunk (double buffered like bscan3)
//   mode 3: both
//   mode 7: both + the uniform guards `if (t < ntile)` / `if (j < nqw)` of the generic loop (runtime 4 / 4)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef const __attribute__((address_space(4))) float *const_f32p;
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *q, int iters, int nch, int ntile, int nqw) {
    __shared__ float4 tile[256 * 9];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 256 * 9; i += 256) tile[i] = make_float4(i * 0.001f, 1.f, 2.f, 3.f);
    __syncthreads();
    const float4 *col = tile + lane * 9;
    const_f32p qs[4];
    for (int j = 0; j < 4; ++j) qs[j] = (const_f32p)(q + ((blockIdx.x * 4 + j) & 1023) * 128);
    float acc[4][4] = {};
    float4 rv[4];
    for (int t = 0; t < 4; ++t) rv[t] = col[t * 64 * 9];
    float qa[4][4], qb[4][4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) { qa[j][e] = qs[j][e]; qb[j][e] = qs[j][4 + e]; }
    for (int it = 0; it < iters; ++it) {
        for (int c = 0; c < nch; c += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int cc = c + half;
                if (MODE & 2) {
                    for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) {
                        if (half == 0) qb[j][e] = qs[j][4 * ((cc + 1) & 7) + e]; else qa[j][e] = qs[j][4 * ((cc + 1) & 7) + e];
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if ((MODE & 4) && t >= ntile) continue;
                    float4 r = (MODE & 1) ? col[t * 64 * 9 + (cc & 7)] : rv[t];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if ((MODE & 4) && j >= nqw) continue;
                        const float *qq = half == 0 ? qa[j] : qb[j];
                        const float t0 = (qq[0] - r.x) + 1e-6f, t1 = (qq[1] - r.y) + 1e-6f, t2 = (qq[2] - r.z) + 1e-6f, t3 = (qq[3] - r.w) + 1e-6f;
                        acc[t][j] = fmaf(t3, t3, fmaf(t2, t2, fmaf(t1, t1, fmaf(t0, t0, acc[t][j]))));
                    }
                }
            }
        }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 4; ++j) s += acc[t][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *d, *q;
    hipMalloc(&d, 256 * 4096 * 4);
    hipMalloc(&q, 1024 * 128 * 4);
    hipMemset(q, 0, 1024 * 128 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 64, nch = 8, blocks = 256 * 4 * 4;  // 4 resident workgroups per CU (4 waves/SIMD), 4 rounds
    for (int mode = 1; mode < 8; mode += (mode == 3 ? 4 : 1))
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 1) k<1><<<blocks, 256>>>(d, q, iters, nch, 4, 4);
            if (mode == 2) k<2><<<blocks, 256>>>(d, q, iters, nch, 4, 4);
            if (mode == 3) k<3><<<blocks, 256>>>(d, q, iters, nch, 4, 4);
            if (mode == 7) k<7><<<blocks, 256>>>(d, q, iters, nch, 4, 4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double winst = (double)blocks * 4 * iters * nch * 4 * 4 * 12;
            if (rep) printf("mode %d: %.3f ms, %.2f cycles(@2.4GHz)/VALU wave-instr/SIMD\n", mode, ms, ms * 1e-3 * 2.4e9 / (winst / 1024.0));
        }
    return 0;
}
