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
// exchange (one barrier per operation).  A child produced by the operation just
// before is consumed straight from the D registers; A operands, tip masks and
// older-sibling CLVs of operation i+1 are fetched while operation i computes.  A-operands come from an MFMA-ready copy of the P-matrices that
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
constexpr int kMfmaNT = 1;        // 16-site column tiles per wave

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

// B operands of one child for the kMfmaNT site tiles of a wave
struct ChildB {
  double b[kMfmaNT][kMfmaSteps];
};

template <int MAXT>   // 64 * rate categories, rounded up to 256 or 1024
__global__ void __launch_bounds__(MAXT)
clv_k20_traversal_kernel(DeviceView v, const double *__restrict__ pmfma,
                         const LevelOp *__restrict__ ops, unsigned nops) {
  // flags[parity][rate][site in block]: "this rate's 20 entries are all < 2^-256";
  // double-buffered by operation parity so one barrier per operation suffices
  __shared__ unsigned flags[2][16][16 * kMfmaNT];
  __shared__ uint64_t masks[256];   // code -> state mask
  const unsigned R = v.rate_cats, S = v.sites;
  const unsigned lane = threadIdx.x & 63, r = threadIdx.x >> 6;   // wave = rate
  const unsigned col = lane & 15, grp = lane >> 4;
  const unsigned site0 = blockIdx.x * (16 * kMfmaNT);
  for (unsigned e = threadIdx.x; e < 256; e += blockDim.x) masks[e] = v.codemask[e];
  __syncthreads();

  unsigned site[kMfmaNT], ls[kMfmaNT];
#pragma unroll
  for (int nt = 0; nt < kMfmaNT; ++nt) {
    site[nt] = site0 + nt * 16 + col;
    ls[nt] = site[nt] < S ? site[nt] : S - 1;   // clamped for loads
  }
  const bool sc_lane = r == 0 && grp == 0;      // the lanes that own the per-site scalers

  // everything operation `op` needs from memory, one operation ahead:
  // A operands (MFMA-ready P), B operands of tip / older-sibling children, scalers
  auto load_a = [&](const LevelOp &op, double (&a1)[kMfmaTiles][kMfmaSteps],
                    double (&a2)[kMfmaTiles][kMfmaSteps]) {
    const double *p1 = pmfma + ((size_t)op.child1_mat * R + r) * (kMfmaTiles * kMfmaSteps * 64) + lane;
    const double *p2 = pmfma + ((size_t)op.child2_mat * R + r) * (kMfmaTiles * kMfmaSteps * 64) + lane;
#pragma unroll
    for (int t = 0; t < kMfmaTiles; ++t)
#pragma unroll
      for (int s = 0; s < kMfmaSteps; ++s) {
        a1[t][s] = p1[(t * kMfmaSteps + s) * 64];
        a2[t][s] = p2[(t * kMfmaSteps + s) * 64];
      }
  };
  auto load_b = [&](unsigned src, unsigned clv, int scb, ChildB &cb, unsigned (&sc)[kMfmaNT]) {
    if (src == 2u) return;   // produced by the previous operation: stays in registers
#pragma unroll
    for (int nt = 0; nt < kMfmaNT; ++nt) {
      if (src == 0u) {       // tip: a 0/1 operand straight from the state mask
        const uint64_t mask = masks[v.tipcodes[(size_t)clv * v.tip_stride + ls[nt]]];
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) cb.b[nt][s] = ((mask >> (4 * s + grp)) & 1) ? 1.0 : 0.0;
        sc[nt] = 0;
      } else {               // an older sibling from HBM / L2
        const double *c = v.clv + (size_t)(clv - v.tips) * v.clv_stride +
                          ((size_t)ls[nt] * R + r) * kMfmaK + grp;
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) cb.b[nt][s] = c[4 * s];
        sc[nt] = (sc_lane && scb >= 0) ? v.scaler[(size_t)scb * S + ls[nt]] : 0u;
      }
    }
  };

  double a1[kMfmaTiles][kMfmaSteps], a2[kMfmaTiles][kMfmaSteps];
  ChildB b1, b2;
  unsigned sc1[kMfmaNT] = {0}, sc2[kMfmaNT] = {0};
  v4d res[kMfmaNT][kMfmaTiles];       // the CLV this wave produced last (D layout = B layout)
  unsigned osc[kMfmaNT] = {0};
#pragma unroll
  for (int nt = 0; nt < kMfmaNT; ++nt)
#pragma unroll
    for (int t = 0; t < kMfmaTiles; ++t) res[nt][t] = v4d{0, 0, 0, 0};
  {
    const LevelOp op0 = ops[0];
    load_a(op0, a1, a2);
    load_b(op0.src1, op0.child1_clv, op0.child1_sc, b1, sc1);
    load_b(op0.src2, op0.child2_clv, op0.child2_sc, b2, sc2);
  }

  for (unsigned oi = 0; oi < nops; ++oi) {
    const LevelOp op = ops[oi];
    const bool more = oi + 1 < nops;
    const LevelOp nx = ops[more ? oi + 1 : oi];
    // a child produced by the previous operation: its D registers ARE the B operand
    if (op.src1 == 2u) {
#pragma unroll
      for (int nt = 0; nt < kMfmaNT; ++nt) {
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) b1.b[nt][s] = res[nt][s / 4][s % 4];
        sc1[nt] = osc[nt];
      }
    }
    if (op.src2 == 2u) {
#pragma unroll
      for (int nt = 0; nt < kMfmaNT; ++nt) {
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) b2.b[nt][s] = res[nt][s / 4][s % 4];
        sc2[nt] = osc[nt];
      }
    }
    // the MFMAs of this operation
#pragma unroll
    for (int nt = 0; nt < kMfmaNT; ++nt) {
      bool small = true;
#pragma unroll
      for (int t = 0; t < kMfmaTiles; ++t) {
        v4d d1 = {0, 0, 0, 0}, d2 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < kMfmaSteps; ++s) {
          d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t][s], b1.b[nt][s], d1, 0, 0, 0);
          d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[t][s], b2.b[nt][s], d2, 0, 0, 0);
        }
        v4d o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = d1[q] * d2[q];   // rows >= 20 are exact zeros (padded A rows)
          small = small && (o[q] < kScaleThreshold);
        }
        res[nt][t] = o;
      }
      int sm = small ? 1 : 0;   // across the 4 lane groups that hold the other rows of this site
      sm &= __shfl_xor(sm, 16);
      sm &= __shfl_xor(sm, 32);
      if (grp == 0) flags[oi & 1][r][nt * 16 + col] = (unsigned)sm;
    }
    __syncthreads();
    const bool scaled_buffer = op.parent_sc >= 0;
    double *pc = v.clv + (size_t)(op.parent_clv - v.tips) * v.clv_stride;
#pragma unroll
    for (int nt = 0; nt < kMfmaNT; ++nt) {
      bool all_small = scaled_buffer;
      for (unsigned q = 0; q < R; ++q) all_small = all_small && flags[oi & 1][q][nt * 16 + col];
      if (all_small) {
#pragma unroll
        for (int t = 0; t < kMfmaTiles; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) res[nt][t][q] *= kScaleFactor;
      }
      osc[nt] = scaled_buffer ? sc1[nt] + sc2[nt] + (all_small ? 1u : 0u) : 0u;
      if (site[nt] < S) {
        if (scaled_buffer && sc_lane) v.scaler[(size_t)op.parent_sc * S + site[nt]] = osc[nt];
        double *dst = pc + ((size_t)site[nt] * R + r) * kMfmaK + grp;
#pragma unroll
        for (int t = 0; t < kMfmaTiles; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 16 * t + 4 * q;   // + grp
            if (row + (int)grp < kMfmaK) dst[row] = res[nt][t][q];
          }
      }
    }
    if (more) {   // operands of the next operation (other waves of the CU cover the latency)
      load_a(nx, a1, a2);
      load_b(nx.src1, nx.child1_clv, nx.child1_sc, b1, sc1);
      load_b(nx.src2, nx.child2_clv, nx.child2_sc, b2, sc2);
    }
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
  if (p->rate_cats <= 4)
    clv_k20_traversal_kernel<256><<<gx, 64 * p->rate_cats, 0, p->stream>>>(v, p->d_pmat_mfma, d_ops, nops);
  else
    clv_k20_traversal_kernel<1024><<<gx, 64 * p->rate_cats, 0, p->stream>>>(v, p->d_pmat_mfma, d_ops, nops);
  return hipGetLastError();
}

}  // namespace rdamd
