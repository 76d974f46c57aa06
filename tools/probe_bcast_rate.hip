// Hardware probe: cost of getting wave-uniform operands: v_readlane_b32 vs broadcast ds_read_b128.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(threadIdx.x, 1, 2, 3);
    __syncthreads();
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float v = threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {  // 4 readlanes feeding 12 VALU
                float q0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (it + j) & 63));
                float q1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (it + j + 1) & 63));
                float q2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (it + j + 2) & 63));
                float q3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (it + j + 3) & 63));
                float t0 = (q0 - v) + 1e-6f, t1 = (q1 - v) + 1e-6f, t2 = (q2 - v) + 1e-6f, t3 = (q3 - v) + 1e-6f;
                acc[j] = fmaf(t3, t3, fmaf(t2, t2, fmaf(t1, t1, fmaf(t0, t0, acc[j]))));
            } else if (MODE == 1) {  // one broadcast ds_read_b128 feeding 12 VALU
                float4 q = lds[(it + j) & 255];
                float t0 = (q.x - v) + 1e-6f, t1 = (q.y - v) + 1e-6f, t2 = (q.z - v) + 1e-6f, t3 = (q.w - v) + 1e-6f;
                acc[j] = fmaf(t3, t3, fmaf(t2, t2, fmaf(t1, t1, fmaf(t0, t0, acc[j]))));
            } else {  // 12 VALU only
                float q0 = it, q1 = it + 1, q2 = it + 2, q3 = it + 3;
                float t0 = (q0 - v) + 1e-6f, t1 = (q1 - v) + 1e-6f, t2 = (q2 - v) + 1e-6f, t3 = (q3 - v) + 1e-6f;
                acc[j] = fmaf(t3, t3, fmaf(t2, t2, fmaf(t1, t1, fmaf(t0, t0, acc[j]))));
            }
        }
    }
    float s = 0; for (int j = 0; j < 8; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048, blocks = 256 * 4;  // 4 blocks/CU -> 4 waves/SIMD
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) k<0><<<blocks, 256>>>(d, iters);
        if (mode == 1) k<1><<<blocks, 256>>>(d, iters);
        if (mode == 2) k<2><<<blocks, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double units = (double)blocks * 4 * iters * 8;  // (4 operands + 12 VALU) units per wave
        if (rep) printf("mode %d (%s): %.3f ms, %.1f ns-cycles@2.1GHz per unit per SIMD\n", mode,
                        mode == 0 ? "4 readlane + 12 VALU" : mode == 1 ? "1 bcast ds_read_b128 + 12 VALU" : "12 VALU", ms,
                        ms * 1e-3 * 2.1e9 / (units / 1024.0));
    }
    return 0;
}
