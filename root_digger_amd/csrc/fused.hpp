// Device-side records of the fused full-traversal evaluator (kernels_fused.hip)
// and the host objects that own them (evaluate.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace rdamd {

enum : uint32_t {
  kFusedTT = 0,   // both children are tips
  kFusedRT = 1,   // X = running CLV (register), Y = tip
  kFusedRP = 2,   // X = running CLV (register), Y = popped from the LDS stack
};

// One step of a compiled traversal ("program"), 16 bytes, read with scalar loads.
struct FusedOp {
  uint32_t mats;         // P-matrix indices of the two child branches: X | Y << 16
  uint32_t tipX, tipY;   // tip rows (0 when the operand is not a tip)
  uint32_t flags;        // kind (kFused*) | spill << 8 (push the running CLV first)
};

struct FusedJob {
  const FusedOp *prog;   // device
  const double  *brlen;  // device, indexed by P-matrix index
  uint32_t n_ops, depth;
};

struct FusedArgs {
  const FusedJob *jobs;
  const uint8_t  *tipcodes;          // [tips][sites]
  const unsigned *pattern_weights;   // [sites]
  const double   *pmat;              // [job][matrix][rate][16]
  const double   *tiptab;            // [job][matrix][rate][16 codes][4]
  const double   *freqs;             // [job][4]
  const double   *rate_weights;      // [job][R]
  double         *partials;          // [job][blocks_x]
  double         *persite;           // [job][sites] or null
  size_t   pmat_job_stride;
  unsigned sites, rate_cats;
  unsigned tipcodes_bytes;           // tips * sites
};

hipError_t launch_fused_pmatrix(const FusedArgs &a, const double *d_q, const double *d_rates,
                                unsigned n_jobs, unsigned n_mat, hipStream_t stream);
hipError_t launch_fused_eval(const FusedArgs &a, unsigned n_jobs, unsigned max_depth,
                             unsigned blocks_x, double *d_out, hipStream_t stream);

}  // namespace rdamd
