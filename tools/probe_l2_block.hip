// Hardware probe: issue rate of the tiled scan's L2 block ((q - c) + eps, fmac) on gfx950 per SIMD at 2/4/8 waves per
// SIMD, in several formulations, plus the shader clock actually sustained (s_memtime vs the 100 MHz wall clock).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_l2_block.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define SUB4(Q0, Q1, Q2, Q3) "v_sub_f32 %[t0], " Q0 ", %[r0]\n\tv_sub_f32 %[t1], " Q1 ", %[r1]\n\tv_sub_f32 %[t2], " Q2 ", %[r2]\n\tv_sub_f32 %[t3], " Q3 ", %[r3]\n\t"
#define ADD4(E) "v_add_f32 %[t0], " E ", %[t0]\n\tv_add_f32 %[t1], " E ", %[t1]\n\tv_add_f32 %[t2], " E ", %[t2]\n\tv_add_f32 %[t3], " E ", %[t3]\n\t"
#define FMA4 "v_fmac_f32 %[a], %[t0], %[t0]\n\tv_fmac_f32 %[a], %[t1], %[t1]\n\tv_fmac_f32 %[a], %[t2], %[t2]\n\tv_fmac_f32 %[a], %[t3], %[t3]"
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *clk, int iters, float q0, float q1, float q2, float q3) {
    float acc[16];
    float r[4] = {threadIdx.x * 0.001f, threadIdx.x * 0.002f, 1.5f, 2.5f};
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float eps;
    asm volatile("s_mov_b32 %0, 0x358637bd" : "=s"(eps));
    float vq0 = q0 + threadIdx.x * 1e-9f, vq1 = q1, vq2 = q2, vq3 = q3, veps = eps;   // VGPR copies
    asm volatile("" : "+v"(vq0), "+v"(vq1), "+v"(vq2), "+v"(vq3), "+v"(veps));
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            float t0, t1, t2, t3, u0, u1, u2, u3;
            if (MODE == 0) {          // shipped block: q and eps in SGPRs, 4 dependent fmacs
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    asm volatile(SUB4("%[q0]", "%[q1]", "%[q2]", "%[q3]") ADD4("%[eps]") FMA4
                                 : [a] "+v"(acc[i + h]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                                 : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3), [eps] "s"(eps));
            } else if (MODE == 1) {   // q and eps in VGPRs
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    asm volatile(SUB4("%[q0]", "%[q1]", "%[q2]", "%[q3]") ADD4("%[eps]") FMA4
                                 : [a] "+v"(acc[i + h]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                                 : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [q0] "v"(vq0), [q1] "v"(vq1), [q2] "v"(vq2), [q3] "v"(vq3), [eps] "v"(veps));
            } else if (MODE == 2) {   // q SGPR, eps literal
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    asm volatile(SUB4("%[q0]", "%[q1]", "%[q2]", "%[q3]") ADD4("0x358637bd") FMA4
                                 : [a] "+v"(acc[i + h]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                                 : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3));
            } else if (MODE == 3) {   // two queries interleaved (fmac chains alternate), SGPR operands
                asm volatile(
                    "v_sub_f32 %[t0], %[q0], %[r0]\n\tv_sub_f32 %[u0], %[q1], %[r0]\n\tv_sub_f32 %[t1], %[q1], %[r1]\n\tv_sub_f32 %[u1], %[q2], %[r1]\n\t"
                    "v_sub_f32 %[t2], %[q2], %[r2]\n\tv_sub_f32 %[u2], %[q3], %[r2]\n\tv_sub_f32 %[t3], %[q3], %[r3]\n\tv_sub_f32 %[u3], %[q0], %[r3]\n\t"
                    "v_add_f32 %[t0], %[eps], %[t0]\n\tv_add_f32 %[u0], %[eps], %[u0]\n\tv_add_f32 %[t1], %[eps], %[t1]\n\tv_add_f32 %[u1], %[eps], %[u1]\n\t"
                    "v_add_f32 %[t2], %[eps], %[t2]\n\tv_add_f32 %[u2], %[eps], %[u2]\n\tv_add_f32 %[t3], %[eps], %[t3]\n\tv_add_f32 %[u3], %[eps], %[u3]\n\t"
                    "v_fmac_f32 %[a], %[t0], %[t0]\n\tv_fmac_f32 %[b], %[u0], %[u0]\n\tv_fmac_f32 %[a], %[t1], %[t1]\n\tv_fmac_f32 %[b], %[u1], %[u1]\n\t"
                    "v_fmac_f32 %[a], %[t2], %[t2]\n\tv_fmac_f32 %[b], %[u2], %[u2]\n\tv_fmac_f32 %[a], %[t3], %[t3]\n\tv_fmac_f32 %[b], %[u3], %[u3]"
                    : [a] "+v"(acc[i]), [b] "+v"(acc[i + 1]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3)
                    : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3), [eps] "s"(eps));
            } else if (MODE == 4) {   // fmac chains only (24 dependent-in-fours fmacs)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    asm volatile(FMA4 "\n\t" FMA4 "\n\t" FMA4
                                 : [a] "+v"(acc[i + h]) : [t0] "v"(r[0]), [t1] "v"(r[1]), [t2] "v"(r[2]), [t3] "v"(r[3]));
            } else if (MODE == 5) {   // sub + add only, SGPR operands (no fmac)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    asm volatile(SUB4("%[q0]", "%[q1]", "%[q2]", "%[q3]") ADD4("%[eps]") "v_add_f32 %[t0], %[eps], %[t0]\n\tv_add_f32 %[t1], %[eps], %[t1]\n\tv_add_f32 %[t2], %[eps], %[t2]\n\tv_add_f32 %[t3], %[eps], %[t3]"
                                 : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                                 : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3), [eps] "s"(eps));
                    acc[i + h] = t0;
                }
            } else if (MODE == 7 || MODE == 8) {   // packed f32 (VOP3P): two elements per instruction, two partial sums per accumulator pair
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 qa = {q0, q1}, qb = {q2, q3}, e2 = {eps, eps};
                f2 ra = {r[0], r[1]}, rb = {r[2], r[3]};
                f2 a2 = {acc[i], acc[i + 1]};
                f2 ta, tb;
                if (MODE == 7) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        asm volatile("v_pk_add_f32 %[ta], %[qa], %[ra] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[tb], %[qb], %[rb] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                                     "v_pk_add_f32 %[ta], %[ta], %[e]\n\tv_pk_add_f32 %[tb], %[tb], %[e]\n\t"
                                     "v_pk_fma_f32 %[a], %[ta], %[ta], %[a]\n\tv_pk_fma_f32 %[a], %[tb], %[tb], %[a]"
                                     : [a] "+v"(a2), [ta] "=&v"(ta), [tb] "=&v"(tb) : [qa] "s"(qa), [qb] "s"(qb), [ra] "v"(ra), [rb] "v"(rb), [e] "s"(e2));
                } else {
                    f2 vqa = {vq0, vq1}, vqb = {vq2, vq3}, ve = {veps, veps};
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        asm volatile("v_pk_add_f32 %[ta], %[qa], %[ra] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[tb], %[qb], %[rb] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                                     "v_pk_add_f32 %[ta], %[ta], %[e]\n\tv_pk_add_f32 %[tb], %[tb], %[e]\n\t"
                                     "v_pk_fma_f32 %[a], %[ta], %[ta], %[a]\n\tv_pk_fma_f32 %[a], %[tb], %[tb], %[a]"
                                     : [a] "+v"(a2), [ta] "=&v"(ta), [tb] "=&v"(tb) : [qa] "v"(vqa), [qb] "v"(vqb), [ra] "v"(ra), [rb] "v"(rb), [e] "v"(ve));
                }
                acc[i] = a2.x; acc[i + 1] = a2.y;
            } else if (MODE == 9) {   // packed sub and add (same roundings per element), scalar fmac chain in k order: bit-identical to mode 2
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 qa = {q0, q1}, qb = {q2, q3}, e2 = {eps, eps};
                f2 ra = {r[0], r[1]}, rb = {r[2], r[3]};
                f2 ta, tb;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                {
                    asm volatile("v_pk_add_f32 %[ta], %[qa], %[ra] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[tb], %[qb], %[rb] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                                 "v_pk_add_f32 %[ta], %[ta], %[e]\n\tv_pk_add_f32 %[tb], %[tb], %[e]"
                                 : [ta] "=&v"(ta), [tb] "=&v"(tb) : [qa] "s"(qa), [qb] "s"(qb), [ra] "v"(ra), [rb] "v"(rb), [e] "s"(e2));
                    const float t0 = ta.x, t1 = ta.y, t2 = tb.x, t3 = tb.y;   // sub-registers of the pairs: no copies
                    asm volatile("v_fmac_f32 %[a], %[t0], %[t0]\n\tv_fmac_f32 %[a], %[t1], %[t1]\n\tv_fmac_f32 %[a], %[t2], %[t2]\n\tv_fmac_f32 %[a], %[t3], %[t3]"
                                 : [a] "+v"(acc[i + h]) : [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [t3] "v"(t3));
                }
            } else if (MODE == 6) {   // VOP3 fma with SGPR q: t = fma(1.0, q, -c) is not the reference rounding -- rate probe only
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    asm volatile("v_fma_f32 %[t0], %[r0], -1.0, %[q0]\n\tv_fma_f32 %[t1], %[r1], -1.0, %[q1]\n\tv_fma_f32 %[t2], %[r2], -1.0, %[q2]\n\tv_fma_f32 %[t3], %[r3], -1.0, %[q3]\n\t"
                                 ADD4("%[eps]") FMA4
                                 : [a] "+v"(acc[i + h]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                                 : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3), [eps] "s"(eps));
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int MODE>
void run(const char *name, float *d, unsigned long long *c) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 1024;
    for (int wps = 2; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;   // 256 CUs x wps blocks of 4 waves -> wps waves per SIMD
        float ms = 0; unsigned long long h[2];
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            k<MODE><<<blocks, 256>>>(d, c, iters, 1.f, 2.f, 3.f, 4.f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        }
        const double winst = (double)blocks * 4 * iters * 16 * 12;
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);
        printf("%-34s %d waves/SIMD: %.3f ms  clock %.2f GHz  %.2f cycles/VALU/SIMD\n", name, wps, ms, ghz, ms * 1e-3 * ghz * 1e9 / (winst / 1024.0));
    }
}
int main() {
    float *d; unsigned long long *c;
    (void)hipMalloc(&d, 256 * 4096 * 4);
    (void)hipMalloc(&c, 16);
    run<0>("0 q,eps SGPR; fmac chain", d, c);
    run<1>("1 q,eps VGPR; fmac chain", d, c);
    run<2>("2 q SGPR, eps literal", d, c);
    run<3>("3 two queries interleaved, SGPR", d, c);
    run<4>("4 fmac chains only", d, c);
    run<5>("5 sub/add only, SGPR", d, c);
    run<6>("6 fma(-c,1,q) + add + fmac, SGPR", d, c);
    run<7>("7 packed f32, q/eps SGPR pairs (per VALU-equivalent: 12 per 4 elements)", d, c);
    run<8>("8 packed f32, q/eps VGPR pairs", d, c);
    run<9>("9 packed sub+add, scalar fmac chain (bit-identical to 2)", d, c);
    return 0;
}
