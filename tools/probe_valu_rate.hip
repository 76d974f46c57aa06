// Hardware probe: issue rate of v_fma_f32 vs v_pk_fma_f32 (and v_pk_add_f32) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16]; f2 y[16];
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], x[i] + 1}; }
    f2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
            if (MODE == 2) y[i] = y[i] + a2;
            if (MODE == 3) x[i] = x[i] + a;
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;  // 8 blocks/CU -> 8 waves/SIMD
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 1) k<1><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 2) k<2><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            if (mode == 3) k<3><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double winst = (double)blocks * 4 * iters * 16;  // wave-instructions
            double per_simd_cyc = ms * 1e-3 * 2.4e9 / (winst / 1024.0);
            if (rep) printf("mode %d (%s): %.3f ms, %.2f cycles(@2.4GHz)/wave-instr/SIMD, %.1f Tlane-ops/s\n", mode,
                            mode == 0 ? "v_fma_f32" : mode == 1 ? "v_pk_fma_f32" : mode == 2 ? "v_pk_add_f32" : "v_add_f32", ms, per_simd_cyc,
                            winst * 64 * (mode == 1 || mode == 2 ? 2 : 1) / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
