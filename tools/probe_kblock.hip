// Hardware probe: the tiled scan's hand-scheduled k-block (nlsh::l2_kblock<NQ, NT> from csrc/scan_bucket.hip, the very code the
// kernel runs) in isolation -- LDS tile resident, no staging, no barriers, no epilogue -- at the kernel's occupancy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe_kblock.hip neural-locality-sensitive-hashing_amd/csrc/capi.hip -o /tmp/probe_kblock
#include "../neural-locality-sensitive-hashing_amd/csrc/scan_bucket.hip"

template <int NQ, int NT, bool MOVING>
__global__ __launch_bounds__(256) void probe(float *out, const float *q, unsigned long long *clk, int iters, int nq_rows) {
    __shared__ float4 tile[256 * 5];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 256 * 5; i += 256) tile[i] = make_float4(i * 0.001f, 1.f, 2.f, 3.f);
    __syncthreads();
    float acc[4][4] = {};
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        nlsh::const_f32p qk[4];
        for (int j = 0; j < 4; ++j) {
            const int row = MOVING ? (blockIdx.x * 16 + wave * 4 + j + it * 97) % nq_rows : (blockIdx.x * 16 + wave * 4 + j) % nq_rows;
            qk[j] = (nlsh::const_f32p)(q + (long long)__builtin_amdgcn_readfirstlane(row) * 128 + (MOVING ? (it & 7) * 16 : 0));
        }
        nlsh::l2_kblock<NQ, NT>(tile + lane * 5, 5, 4, qk, acc);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 4; ++j) s += acc[t][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NQ, int NT, bool MOVING>
static void run(const char *name, float *d, const float *q, unsigned long long *c, int wgs_per_cu) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2048, blocks = 256 * wgs_per_cu;
    float ms = 0; unsigned long long h[2];
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        probe<NQ, NT, MOVING><<<blocks, 256>>>(d, q, c, iters, 10000);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    }
    const double winst = (double)blocks * 4 * iters * 4 * NQ * NT * 12;
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);
    printf("%-40s %d WG/CU: %.3f ms  clock %.2f GHz  %.2f cycles per useful VALU per SIMD\n", name, wgs_per_cu, ms, ghz, ms * 1e-3 * ghz * 1e9 / (winst / 1024.0));
}

int main() {
    float *d, *q; unsigned long long *c;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    (void)hipMalloc(&q, 10000 * 128 * 4);
    (void)hipMemset(q, 0, 10000 * 128 * 4);
    (void)hipMalloc(&c, 16);
    for (int w : {2, 4, 7}) {
        run<4, 4, false>("kblock<4,4> queries fixed (cache hits)", d, q, c, w);
        run<4, 4, true>("kblock<4,4> queries moving", d, q, c, w);
        run<2, 4, true>("kblock<2,4> queries moving", d, q, c, w);
        run<1, 4, true>("kblock<1,4> queries moving", d, q, c, w);
        run<4, 1, true>("kblock<4,1> (5-slot rows) queries moving", d, q, c, w);
    }
    return 0;
}
