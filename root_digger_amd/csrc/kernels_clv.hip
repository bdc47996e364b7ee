// Conditional-likelihood-vector (CLV) updates.
//
// Replaces corax_update_clvs (called at /root/reference/src/model.cpp:402, :440,
// :461, :851).  For every operation of a traversal
//   parent[s][r][i] = (sum_j P1[r][i][j] c1[s][r][j]) * (sum_j P2[r][i][j] c2[s][r][j])
// followed by the per-site 2^256 rescale rule:
// the parent scaler starts as the sum of the children's scalers and is
// incremented (and the site multiplied by 2^256) when EVERY entry of the site
// is below 2^-256.
//
// HBM layout: CLV [site][rate][state] contiguous doubles, so consecutive lanes
// read consecutive 32-byte (DNA) records.  Tips are never expanded to CLVs:
// a tip child contributes tiptab[matrix][rate][code][i] (built next to the
// P-matrices), read from LDS.
//
// A whole operation list runs in ONE launch.  Every dependency of the
// traversal is site-local -- parent[s][r] needs only child[s][r] -- and each
// lane owns one (site, rate) pair for the whole list, so there is no
// synchronisation at all between waves.  A child produced by the operation
// just before (every inner child of a tip+inner node, the second child of an
// inner+inner node) is taken from the lane's registers instead of being read
// back; an older sibling is prefetched one operation ahead; every CLV is
// still written (the reference's state contract).  What is left on the memory
// pipe is an almost pure store stream.
#include "common.hpp"

#include <algorithm>

namespace rdamd {

// ---------------------------------------------------------------------------
// 4-state path: one lane per (site, rate); the R lanes of a site are adjacent
// so the "all entries below threshold" test is an in-wave AND.
// ---------------------------------------------------------------------------
// LevelOp::src* values
enum : unsigned { kSrcTip = 0, kSrcMem = 1, kSrcReg = 2, kSrcPark = 3 };

template <int R>
__global__ void __launch_bounds__(256)
clv_dna_traversal_kernel(DeviceView v, const LevelOp *__restrict__ ops, unsigned nops,
                         unsigned slots) {
  // Per WAVE, double-buffered: the P-matrices [R][4][4] of both children.
  // Waves never synchronise with each other: each stages the matrices of
  // operation i+1 itself while operation i computes.  (A tip child is expanded
  // from its code to a 0/1 vector and goes through the same product: this kernel
  // is bound by HBM, not by FMAs, and P is a quarter of the tip table to stage.)
  __shared__ double smat_all[4][2][2][R * 16];
  // LDS parking: `slots` CLVs (+ scaler counts) per lane, [slot][half][lane] so
  // every 16-byte access of a wave is conflict-free.  An older sibling waits
  // here for its operation instead of being read back from HBM.
  extern __shared__ double2 park_lds[];
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double (*smat)[2][R * 16] = smat_all[wave];
  const unsigned S = v.sites;
  const size_t total = (size_t)S * R;
  const size_t idx = (size_t)blockIdx.x * 256 + tid;   // one (site, rate) pair per lane
  const bool active = idx < total;
  const size_t cidx = active ? idx : total - 1;        // clamped for loads
  const unsigned s = (unsigned)(cidx / R), r = (unsigned)(cidx % R);

  // Staging is split (load early into registers, write to LDS late) so that no
  // wave ever sits on a global load it has just issued: while operation i
  // computes, the matrices, tip codes and older-sibling CLVs of operation i+1
  // are all in flight.
  constexpr int kStageRegs = (R * 16 + 63) / 64;   // doubles per lane per child
  auto stage_load = [&](const LevelOp &op, double (&s1)[kStageRegs], double (&s2)[kStageRegs]) {
    const double *src1 = v.pmat + (size_t)op.child1_mat * R * 16;
    const double *src2 = v.pmat + (size_t)op.child2_mat * R * 16;
#pragma unroll
    for (int k = 0; k < kStageRegs; ++k) {
      const unsigned e = lane + 64 * k;
      s1[k] = e < R * 16 ? src1[e] : 0.0;
      s2[k] = e < R * 16 ? src2[e] : 0.0;
    }
  };
  auto stage_write = [&](unsigned buf, const double (&s1)[kStageRegs], const double (&s2)[kStageRegs]) {
#pragma unroll
    for (int k = 0; k < kStageRegs; ++k) {
      const unsigned e = lane + 64 * k;
      if (e < R * 16) {
        smat[buf][0][e] = s1[k];
        smat[buf][1][e] = s2[k];
      }
    }
  };
  // an older sibling (not produced by the operation just before) comes from
  // HBM/L2; it is fetched one operation ahead
  auto fetch = [&](const LevelOp &op, int which, double (&x)[4], unsigned &sc) {
    const unsigned clv = which ? op.child2_clv : op.child1_clv;
    const int scb = which ? op.child2_sc : op.child1_sc;
    const double2 *c = reinterpret_cast<const double2 *>(v.clv + (size_t)(clv - v.tips) * v.clv_stride);
    const double2 a = c[cidx * 2], b = c[cidx * 2 + 1];
    x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
    sc = scb >= 0 ? v.scaler[(size_t)scb * S + s] : 0u;
  };
  // everything operation `op` needs from global memory, into registers
  auto prefetch = [&](const LevelOp &op, double (&m1)[4], unsigned &m1sc, double (&m2)[4],
                      unsigned &m2sc, unsigned &cx, unsigned &cy) {
    if (op.src1 == kSrcTip) cx = v.tipcodes[(size_t)op.child1_clv * S + s];
    else if (op.src1 == kSrcMem && !(op.late & 1u)) fetch(op, 0, m1, m1sc);
    if (op.src2 == kSrcTip) cy = v.tipcodes[(size_t)op.child2_clv * S + s];
    else if (op.src2 == kSrcMem && !(op.late & 2u)) fetch(op, 1, m2, m2sc);
  };
  unsigned *park_sc = reinterpret_cast<unsigned *>(park_lds + (size_t)slots * 512);
  auto unpark = [&](unsigned slot, double (&x)[4], unsigned &sc) {
    const double2 a = park_lds[(slot * 2) * 256 + tid], b = park_lds[(slot * 2 + 1) * 256 + tid];
    x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
    sc = park_sc[slot * 256 + tid];
  };

  double o[4] = {0, 0, 0, 0};       // the CLV this lane produced last
  unsigned osc = 0;
  double m1[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0};   // prefetched memory operands
  unsigned m1sc = 0, m2sc = 0, cx = 0, cy = 0;
  {
    double s1[kStageRegs], s2[kStageRegs];
    const LevelOp op0 = ops[0];
    stage_load(op0, s1, s2);
    prefetch(op0, m1, m1sc, m2, m2sc, cx, cy);
    stage_write(0, s1, s2);
  }
  // (A prefetch distance of two was tried: an older sibling may be the parent of
  // operation i, which is not stored yet when operation i+2 would fetch it, and
  // the conditional loads defeat counted vmcnt waits; one operation ahead is
  // both correct by construction and faster.)
  for (unsigned i = 0; i < nops; ++i) {
    const unsigned buf = i & 1;
    const LevelOp op = ops[i];
    const bool more = i + 1 < nops;
    const LevelOp nx = ops[more ? i + 1 : i];
    // operands of THIS operation out of the prefetch registers
    double x[4], y[4];
    unsigned xsc = 0, ysc = 0;
    const unsigned ccx = cx, ccy = cy;
    if (op.src1 == kSrcReg) {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = o[k];
      xsc = osc;
    } else if (op.src1 == kSrcMem) {
      if (op.late & 1u) fetch(op, 0, m1, m1sc);
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = m1[k];
      xsc = m1sc;
    } else if (op.src1 == kSrcTip) {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = ((ccx >> k) & 1u) ? 1.0 : 0.0;
    } else {
      unpark(op.src1 - kSrcPark, x, xsc);
    }
    if (op.src2 == kSrcReg) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = o[k];
      ysc = osc;
    } else if (op.src2 == kSrcMem) {
      if (op.late & 2u) fetch(op, 1, m2, m2sc);
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = m2[k];
      ysc = m2sc;
    } else if (op.src2 == kSrcTip) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = ((ccy >> k) & 1u) ? 1.0 : 0.0;
    } else {
      unpark(op.src2 - kSrcPark, y, ysc);
    }
    // issue everything the NEXT operation needs
    double s1[kStageRegs], s2[kStageRegs];
    if (more) {
      stage_load(nx, s1, s2);
      prefetch(nx, m1, m1sc, m2, m2sc, cx, cy);
    }
    double t1[4], t2[4];
    {
      const double *m = &smat[buf][0][r * 16];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        t1[k] = m[k * 4 + 0] * x[0] + m[k * 4 + 1] * x[1] + m[k * 4 + 2] * x[2] + m[k * 4 + 3] * x[3];
      m = &smat[buf][1][r * 16];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        t2[k] = m[k * 4 + 0] * y[0] + m[k * 4 + 1] * y[1] + m[k * 4 + 2] * y[2] + m[k * 4 + 3] * y[3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = t1[k] * t2[k];
    osc = 0;
    if (op.parent_sc >= 0) {
      int small = (o[0] < kScaleThreshold) & (o[1] < kScaleThreshold) &
                  (o[2] < kScaleThreshold) & (o[3] < kScaleThreshold);
#pragma unroll
      for (int off = 1; off < R; off <<= 1) small &= __shfl_xor(small, off);
      osc = xsc + ysc;
      if (small) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] *= kScaleFactor;
        osc += 1;
      }
      if (r == 0 && active) v.scaler[(size_t)op.parent_sc * S + s] = osc;
    }
    if (active) {
      double2 *pc = reinterpret_cast<double2 *>(v.clv + (size_t)(op.parent_clv - v.tips) * v.clv_stride);
      pc[idx * 2] = make_double2(o[0], o[1]);
      pc[idx * 2 + 1] = make_double2(o[2], o[3]);
    }
    if (op.park) {
      const unsigned slot = op.park - 1;
      park_lds[(slot * 2) * 256 + tid] = make_double2(o[0], o[1]);
      park_lds[(slot * 2 + 1) * 256 + tid] = make_double2(o[2], o[3]);
      park_sc[slot * 256 + tid] = osc;
    }
    if (more) stage_write(buf ^ 1, s1, s2);
  }
}

// ---------------------------------------------------------------------------
// Generic path (any K <= 64, any R): one lane per site, looping over rates and
// states.  Correct for every shape; the 20-state MFMA kernel supersedes it for
// protein data.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
clv_generic_traversal_kernel(DeviceView v, const LevelOp *__restrict__ ops, unsigned nops) {
 for (unsigned oi = 0; oi < nops; ++oi) {
  const LevelOp op = ops[oi];
  const unsigned K = v.states, R = v.rate_cats, S = v.sites, cap = v.ncodes_cap;
  const bool tip1 = op.child1_clv < v.tips, tip2 = op.child2_clv < v.tips;
  const double *c1 = tip1 ? nullptr : v.clv + (size_t)(op.child1_clv - v.tips) * v.clv_stride;
  const double *c2 = tip2 ? nullptr : v.clv + (size_t)(op.child2_clv - v.tips) * v.clv_stride;
  double *pc = v.clv + (size_t)(op.parent_clv - v.tips) * v.clv_stride;
  const double *p1 = v.pmat + (size_t)op.child1_mat * R * K * K;
  const double *p2 = v.pmat + (size_t)op.child2_mat * R * K * K;
  const double *tt1 = v.tiptab + (size_t)op.child1_mat * R * cap * K;
  const double *tt2 = v.tiptab + (size_t)op.child2_mat * R * cap * K;
  unsigned *psc = op.parent_sc >= 0 ? v.scaler + (size_t)op.parent_sc * S : nullptr;
  const unsigned *lsc = op.child1_sc >= 0 ? v.scaler + (size_t)op.child1_sc * S : nullptr;
  const unsigned *rsc = op.child2_sc >= 0 ? v.scaler + (size_t)op.child2_sc * S : nullptr;
  const size_t span = (size_t)R * K;

  for (unsigned s = blockIdx.x * blockDim.x + threadIdx.x; s < S;
       s += gridDim.x * blockDim.x) {
    const unsigned code1 = tip1 ? v.tipcodes[(size_t)op.child1_clv * S + s] : 0;
    const unsigned code2 = tip2 ? v.tipcodes[(size_t)op.child2_clv * S + s] : 0;
    bool small = psc != nullptr;
    for (unsigned r = 0; r < R; ++r) {
      const double *l = tip1 ? nullptr : c1 + s * span + (size_t)r * K;
      const double *q = tip2 ? nullptr : c2 + s * span + (size_t)r * K;
      for (unsigned i = 0; i < K; ++i) {
        double ta, tb;
        if (tip1) {
          ta = tt1[((size_t)r * cap + code1) * K + i];
        } else {
          ta = 0.0;
          const double *row = p1 + ((size_t)r * K + i) * K;
          for (unsigned j = 0; j < K; ++j) ta += row[j] * l[j];
        }
        if (tip2) {
          tb = tt2[((size_t)r * cap + code2) * K + i];
        } else {
          tb = 0.0;
          const double *row = p2 + ((size_t)r * K + i) * K;
          for (unsigned j = 0; j < K; ++j) tb += row[j] * q[j];
        }
        const double o = ta * tb;
        pc[s * span + (size_t)r * K + i] = o;
        small = small && (o < kScaleThreshold);
      }
    }
    if (psc) {
      unsigned sc = (lsc ? lsc[s] : 0u) + (rsc ? rsc[s] : 0u);
      if (small) {
        for (size_t e = 0; e < span; ++e) pc[s * span + e] *= kScaleFactor;
        sc += 1;
      }
      psc[s] = sc;
    }
  }
  __threadfence_block();   // a lane re-reads only its own site's earlier stores
 }
}

static inline bool dna_fast_ok(unsigned K, unsigned R, unsigned cap) {
  return K == 4 && cap == 16 && (R == 1 || R == 2 || R == 4 || R == 8);
}

constexpr size_t kComputeUnits = 256;   // MI355X: 8 XCDs x 32 CUs
constexpr size_t kMaxParkSlots = 6;   // static + dynamic LDS stays under 64 KB per block

// Slots are sized so that every block of the launch is resident at once (the
// blocks of one CU share its 160 KB of LDS): a second dispatch round would cost
// more than the read-backs the extra slots save.
unsigned clv_traversal_slots(const rdamd_partition *p) {
  if (!dna_fast_ok(p->states, p->rate_cats, p->ncodes_cap) || p->sites == 0) return 0;
  const size_t total = (size_t)p->sites * p->rate_cats;
  const size_t blocks = (total + 255) / 256;
  const size_t per_cu = (blocks + kComputeUnits - 1) / kComputeUnits;
  const size_t lds_static = 8u * 1024, per_slot = 256 * (32 + 4);
  const size_t budget = (size_t)(160 * 1024) / std::max<size_t>(per_cu, 1);
  if (budget <= lds_static) return 0;
  return (unsigned)std::min<size_t>((budget - lds_static) / per_slot, kMaxParkSlots);
}

hipError_t launch_clv_traversal(rdamd_partition *p, const LevelOp *d_ops, unsigned nops,
                                unsigned slots) {
  if (nops == 0 || p->sites == 0) return hipSuccess;
  DeviceView v = p->view();
  const unsigned K = p->states, R = p->rate_cats;
  if (dna_fast_ok(K, R, p->ncodes_cap)) {
    // one (site, rate) pair per lane where possible: maximum memory-level
    // parallelism; the lane -> pair map is identical for every operation
    size_t total = (size_t)p->sites * R;
    unsigned gx = (unsigned)((total + 255) / 256);
    const size_t lds = (size_t)slots * 256 * (32 + 4);
    switch (R) {
      case 1: clv_dna_traversal_kernel<1><<<gx, 256, lds, p->stream>>>(v, d_ops, nops, slots); break;
      case 2: clv_dna_traversal_kernel<2><<<gx, 256, lds, p->stream>>>(v, d_ops, nops, slots); break;
      case 4: clv_dna_traversal_kernel<4><<<gx, 256, lds, p->stream>>>(v, d_ops, nops, slots); break;
      default: clv_dna_traversal_kernel<8><<<gx, 256, lds, p->stream>>>(v, d_ops, nops, slots); break;
    }
  } else {
    unsigned gx = (p->sites + 255) / 256;
    clv_generic_traversal_kernel<<<gx, 256, 0, p->stream>>>(v, d_ops, nops);
  }
  return hipGetLastError();
}

}  // namespace rdamd
