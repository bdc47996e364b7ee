// 20-state CLV updates on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Same contract as kernels_clv.hip (replaces corax_update_clvs,
// /root/reference/src/model.cpp:402, :440, :461, :851) for the protein shape
// of BASELINE config c3.  For one rate category the child term of an operation
// is a small GEMM,  T[i][s] = sum_j P[i][j] * c[s][j], tiled as
//     A = P   (20 -> 2 row tiles of 16, padded with zero rows),
//     B = the child CLV of 16 sites (k = child state, 5 steps of 4),
//     D = T   (16 states x 16 sites per tile, 4 doubles per lane),
// i.e. 10 MFMAs per child per 16 sites.  The D layout of this instruction
// (row = lane/16 + 4*reg, column = lane%16) is exactly the B layout of the
// next operation (k = lane/16 + 4*step), so a CLV never needs a transpose.
//
// One wave = (32 sites, one rate); the waves of a workgroup are the R rates of
// the same 32 sites, so the per-site "all entries < 2^-256" rule is one LDS
// exchange.  A-operands come from an MFMA-ready copy of the P-matrices that
// the P-matrix kernel writes ([matrix][rate][tile][step][lane], fully
// coalesced 512-B rows).  Tips are expanded from their state masks in
// registers (a 0/1 B operand), so no tip table and no tip CLV is read.
// Like the 4-state kernel, a whole operation list is one launch: every
// dependency is site-local and each wave owns its sites for the whole list.
#include "common.hpp"

namespace rdamd {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kMfmaK = 20;        // states
constexpr int kMfmaSteps = 5;     // k steps of 4
constexpr int kMfmaTiles = 2;     // row tiles of 16 (rows 20..31 are padding)
constexpr int kMfmaNT = 2;        // 16-site column tiles per wave

// MFMA-ready copy of one 20x20 P-matrix: element (tile t, step s, lane l) =
// P[16 t + l%16][4 s + l/16]  (0 for padded rows)
__global__ void __launch_bounds__(256)
pmat_to_mfma_kernel(const double *__restrict__ pmat, double *__restrict__ out,
                    const unsigned *__restrict__ mat_idx, unsigned count, unsigned R) {
  const unsigned slot_in_list = blockIdx.x / R, r = blockIdx.x % R;
  if (slot_in_list >= count) return;
  const size_t slot = (size_t)mat_idx[slot_in_list] * R + r;
  const double *p = pmat + slot * (kMfmaK * kMfmaK);
  double *o = out + slot * (kMfmaTiles * kMfmaSteps * 64);
  for (unsigned e = threadIdx.x; e < kMfmaTiles * kMfmaSteps * 64; e += blockDim.x) {
    const unsigned l = e & 63, s = (e >> 6) % kMfmaSteps, t = (e >> 6) / kMfmaSteps;
    const unsigned i = 16 * t + (l & 15), j = 4 * s + (l >> 4);
    o[e] = i < kMfmaK ? p[i * kMfmaK + j] : 0.0;
  }
}

__global__ void __launch_bounds__(1024)
clv_k20_traversal_kernel(DeviceView v, const double *__restrict__ pmfma,
                         const LevelOp *__restrict__ ops, unsigned nops) {
  // flags[rate][site in block]: "this rate's 20 entries are all < 2^-256"
  __shared__ unsigned flags[16][16 * kMfmaNT];
  const unsigned R = v.rate_cats, S = v.sites;
  const unsigned lane = threadIdx.x & 63, r = threadIdx.x >> 6;   // wave = rate
  const unsigned col = lane & 15, grp = lane >> 4;
  const unsigned site0 = blockIdx.x * (16 * kMfmaNT);

  for (unsigned oi = 0; oi < nops; ++oi) {
    const LevelOp op = ops[oi];
    const bool tip1 = op.src1 == 0, tip2 = op.src2 == 0;
    // A operands of both children for this wave's rate
    double a1[kMfmaTiles][kMfmaSteps], a2[kMfmaTiles][kMfmaSteps];
    {
      const double *p1 = pmfma + ((size_t)op.child1_mat * R + r) * (kMfmaTiles * kMfmaSteps * 64) + lane;
      const double *p2 = pmfma + ((size_t)op.child2_mat * R + r) * (kMfmaTiles * kMfmaSteps * 64) + lane;
#pragma unroll
      for (int t = 0; t < kMfmaTiles; ++t)
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) {
          a1[t][s] = p1[(t * kMfmaSteps + s) * 64];
          a2[t][s] = p2[(t * kMfmaSteps + s) * 64];
        }
    }
    const double *c1 = tip1 ? nullptr : v.clv + (size_t)(op.child1_clv - v.tips) * v.clv_stride;
    const double *c2 = tip2 ? nullptr : v.clv + (size_t)(op.child2_clv - v.tips) * v.clv_stride;
    double *pc = v.clv + (size_t)(op.parent_clv - v.tips) * v.clv_stride;

    v4d res[kMfmaNT][kMfmaTiles];
#pragma unroll
    for (int nt = 0; nt < kMfmaNT; ++nt) {
      const unsigned site = site0 + nt * 16 + col;
      const unsigned ls = site < S ? site : S - 1;   // clamped for loads
      // B operands: child state (4 step + grp) of site `col`
      double b1[kMfmaSteps], b2[kMfmaSteps];
      if (tip1) {
        const uint64_t mask = v.codemask[v.tipcodes[(size_t)op.child1_clv * S + ls]];
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) b1[s] = ((mask >> (4 * s + grp)) & 1) ? 1.0 : 0.0;
      } else {
        const double *c = c1 + ((size_t)ls * R + r) * kMfmaK + grp;
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) b1[s] = c[4 * s];
      }
      if (tip2) {
        const uint64_t mask = v.codemask[v.tipcodes[(size_t)op.child2_clv * S + ls]];
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) b2[s] = ((mask >> (4 * s + grp)) & 1) ? 1.0 : 0.0;
      } else {
        const double *c = c2 + ((size_t)ls * R + r) * kMfmaK + grp;
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) b2[s] = c[4 * s];
      }
      bool small = true;
#pragma unroll
      for (int t = 0; t < kMfmaTiles; ++t) {
        v4d d1 = {0, 0, 0, 0}, d2 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) {
          d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t][s], b1[s], d1, 0, 0, 0);
          d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[t][s], b2[s], d2, 0, 0, 0);
        }
        v4d o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = d1[q] * d2[q];   // rows >= 20 are exact zeros (padded A rows)
          small = small && (o[q] < kScaleThreshold);
        }
        res[nt][t] = o;
      }
      // across the 4 lane groups that hold the other rows of this site
      int sm = small ? 1 : 0;
      sm &= __shfl_xor(sm, 16);
      sm &= __shfl_xor(sm, 32);
      if (grp == 0) flags[r][nt * 16 + col] = (unsigned)sm;
    }
    __syncthreads();
    const bool scaled_buffer = op.parent_sc >= 0;
#pragma unroll
    for (int nt = 0; nt < kMfmaNT; ++nt) {
      const unsigned site = site0 + nt * 16 + col;
      bool all_small = scaled_buffer;
      for (unsigned q = 0; q < R; ++q) all_small = all_small && flags[q][nt * 16 + col];
      if (site < S) {
        if (scaled_buffer && r == 0 && grp == 0) {
          const unsigned sc = (op.child1_sc >= 0 ? v.scaler[(size_t)op.child1_sc * S + site] : 0u) +
                              (op.child2_sc >= 0 ? v.scaler[(size_t)op.child2_sc * S + site] : 0u) +
                              (all_small ? 1u : 0u);
          v.scaler[(size_t)op.parent_sc * S + site] = sc;
        }
        double *dst = pc + ((size_t)site * R + r) * kMfmaK + grp;
#pragma unroll
        for (int t = 0; t < kMfmaTiles; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 16 * t + 4 * q;   // + grp
            if (row + (int)grp < kMfmaK)
              dst[row] = all_small ? res[nt][t][q] * kScaleFactor : res[nt][t][q];
          }
      }
    }
    // stores of this operation are ordered before the next operation's loads
    // of them (same workgroup), and the flags array may be rewritten
    __threadfence_block();
    __syncthreads();
  }
}

hipError_t launch_pmat_to_mfma(rdamd_partition *p, const unsigned *d_matrix_indices,
                               unsigned count) {
  if (!count) return hipSuccess;
  pmat_to_mfma_kernel<<<count * p->rate_cats, 256, 0, p->stream>>>(
      p->d_pmat, p->d_pmat_mfma, d_matrix_indices, count, p->rate_cats);
  return hipGetLastError();
}

hipError_t launch_clv_k20_traversal(rdamd_partition *p, const LevelOp *d_ops, unsigned nops) {
  DeviceView v = p->view();
  const unsigned per_block = 16 * kMfmaNT;
  const unsigned gx = (p->sites + per_block - 1) / per_block;
  clv_k20_traversal_kernel<<<gx, 64 * p->rate_cats, 0, p->stream>>>(v, p->d_pmat_mfma, d_ops, nops);
  return hipGetLastError();
}

}  // namespace rdamd
