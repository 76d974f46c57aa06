// Kernel-node descriptions of the two launches of a query batch whose arguments change from batch to batch -- encode_hash (batch
// pointer, row stride, Philox seed) and the scan kernel (query pointer) -- for slots that replay a captured hipGraph of the batch
// (step.hip, nlsh_step_create_graph): per batch the slot refreshes these two nodes (hipGraphExecKernelNodeSetParams) and launches.
#pragma once
#include "encode_common.h"
#include "scan_common.h"
#include "scan_plan.h"

namespace nlsh {

struct EncNode {
    EncArgs a;
    PlanArgs pa;
    unsigned n32;
    void *argv[3];
    hipKernelNodeParams p;
};
// what encode_plan_launch would launch for this batch, as node parameters (func, grid, block, dynamic LDS, argument pointers into `out`)
int encode_plan_node(const EncPlan &p, const float *x, int64_t x_stride, uint64_t seed, const PlanArgs *lookup, EncNode *out);

struct ScanNode {
    alignas(16) unsigned char args[640];   // the scan kernels' argument struct (scan_bucket.hip: BArgs), opaque here
    void *argv[1];
    hipKernelNodeParams p;
};
// the scan kernel of NLSH_PHASE_SCAN of this call (bucket-major schedules), as node parameters
int bucket_scan_node(const BucketScanCall &c, ScanNode *out);

}  // namespace nlsh
