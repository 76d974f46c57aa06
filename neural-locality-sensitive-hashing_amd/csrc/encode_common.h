// Host-visible description of one encode_hash launch (shared by encode_hash.hip and step.hip).
#pragma once
#include "common.h"

namespace nlsh {

struct LayerDesc {
    int K, N;      // logical in/out width
    int Kp, Np;    // padded: Kp % 8 == 0, Np % 32 == 0
    long long w_off, b_off;  // float offsets into the packed blob
    long long w16_off;       // the same weights packed for 16x16x4 tiles (hidden layers: a second copy behind the blob; output layer: == w_off)
};

struct EncArgs {
    const float *x;
    long long n, x_stride;
    int n_layers;
    LayerDesc L[NLSH_MAX_LAYERS];
    const float *packed;
    int S;  // LDS row stride (floats)
    int H, act, key_mode, n_probes;
    long long n_multi_rows, row0;
    unsigned long long seed;
    float *z_out, *probs_out;
    uint32_t *code_out;
    int32_t *keys_out, *nkeys_out;
};

// A validated launch of encode_hash with everything that does not change from batch to batch resolved once: kernel form, grid,
// LDS bytes, layer table.  nlsh_encode_hash builds one per call; a pipelined batch slot (step.hip) keeps one and only swaps the
// batch pointer, its row stride and the Philox seed.
struct EncPlan {
    EncArgs a;
    int form;        // ENC_FORM_*
    unsigned grid;
    size_t lds;
    unsigned n32;    // ENC_FORM_HET: workgroups [0, n32) take 32 rows each, the rest 16 each
};
enum { ENC_FORM_H16 = 0, ENC_FORM_SINGLE, ENC_FORM_SINGLE_WIDE, ENC_FORM_BUILD128, ENC_FORM_PINGPONG, ENC_FORM_HET, ENC_FORM_COUNT };

int encode_plan_fill(EncPlan &p, int64_t n, int n_layers, const int *dims, const float *packed, int act, int key_mode, int n_probes,
                     int64_t n_multi_rows, int64_t row0, float *z_out, float *probs_out, uint32_t *code_out, int32_t *keys_out,
                     int32_t *nkeys_out);
struct PlanArgs;   // scan_plan.h: the bucket lookup of the scan's PLAN phase, optionally run in encode_hash's epilogue
// `lookup` (nullable): PlanArgs of the scan call this batch's keys go to (bucket_scan_plan_args), made ready for this launch by
// encode_plan_fuse_lookup -- the scan call then runs NLSH_PHASE_PLAN_REST instead of NLSH_PHASE_PLAN.
int encode_plan_launch(const EncPlan &p, const float *x, int64_t x_stride, uint64_t seed, hipStream_t s, const PlanArgs *lookup = nullptr);
int encode_plan_fuse_lookup(EncPlan &p, PlanArgs &pa);
// One-time, per (device, kernel form) set-up of the launch (the dynamic-LDS permission): called by encode_plan_launch itself; callers
// that CAPTURE the launch into a hipGraph call it before the capture starts.
int encode_plan_prepare(const EncPlan &p);

}  // namespace nlsh
