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

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned uni(unsigned x) {   // assert wave-uniformity
  return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}
// Buffer descriptor over [p, p+bytes); bytes == 0 makes every access through it
// a no-op (loads return 0, stores are dropped) -- how a wave-uniform "this
// operand is not needed" is expressed WITHOUT a branch around the memory
// instruction: every iteration issues the same instruction sequence, so the
// s_waitcnt vmcnt counts are exact and no wait ever covers a store.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = uni((unsigned)u), hi = uni((unsigned)(u >> 32));
  void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)uni(bytes), 0x00020000);
}
constexpr unsigned kOutOfRange = 0x80000000u;   // offset no descriptor here reaches

template <int R>
__global__ void __launch_bounds__(256)
clv_dna_traversal_kernel(DeviceView v, const LevelOp *__restrict__ ops, unsigned nops,
                         unsigned slots, unsigned pmat_bytes) {
  // Per WAVE, double-buffered: the P-matrices [R][4][4] of both children.
  // Waves never synchronise with each other: each stages the matrices of
  // operation i+1 itself while operation i computes.  (A tip child is expanded
  // from its code to a 0/1 vector and goes through the same product: this kernel
  // is bound by memory, not by FMAs, and P is a quarter of the tip table to stage.)
  __shared__ double smat_all[4][2][2][R * 16];
  // LDS parking: `slots` CLVs (+ scaler counts) per lane, [slot][half][lane] so
  // every 16-byte access of a wave is conflict-free.  An older sibling waits
  // here for its operation instead of being read back from HBM.
  extern __shared__ double2 park_lds[];
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double (*smat)[2][R * 16] = smat_all[wave];
  const unsigned S = v.sites;
  const unsigned total = S * R;                      // < 2^26 (launch_clv_traversal)
  const unsigned idx = blockIdx.x * 256 + tid;       // one (site, rate) pair per lane
  const bool active = idx < total;
  const unsigned cidx = active ? idx : total - 1;    // clamped for loads
  const unsigned s = cidx / R, r = cidx % R;

  const unsigned clv_bytes = (unsigned)(v.clv_stride * sizeof(double));
  const unsigned off_clv_ld = cidx * 32u;
  const unsigned off_clv_st = active ? idx * 32u : kOutOfRange;
  const unsigned off_sc_st = (active && r == 0) ? s * 4u : kOutOfRange;
  const __amdgpu_buffer_rsrc_t pmat_rs = make_rsrc(v.pmat, pmat_bytes);
  auto clv_rs = [&](unsigned clv, bool on) {
    return make_rsrc(v.clv + (size_t)(on ? clv - v.tips : 0u) * v.clv_stride, on ? clv_bytes : 0u);
  };
  auto sc_rs = [&](int scb, bool on) {
    on = on && scb >= 0;
    return make_rsrc(v.scaler + (size_t)(on ? scb : 0) * S, on ? S * 4u : 0u);
  };
  auto tip_rs = [&](unsigned clv, bool on) {
    return make_rsrc(v.tipcodes + (size_t)(on ? clv : 0u) * S, on ? S : 0u);
  };

  struct Child {          // one child operand in flight
    v4u lo, hi;           // CLV entries 0-1 / 2-3 (memory child)
    unsigned sc, code;    // its scaler count / tip code
  };
  constexpr int kStageRegs = (R * 16 + 63) / 64;     // doubles per lane per child matrix
  // everything operation `op` needs from memory, issued as ONE fixed
  // instruction sequence (descriptors with 0 bytes switch parts of it off)
  auto issue = [&](const LevelOp &op, Child &c1, Child &c2, double (&s1)[kStageRegs],
                   double (&s2)[kStageRegs]) {
#pragma unroll
    for (int k = 0; k < kStageRegs; ++k) {
      const unsigned e = lane + 64 * k;
      const unsigned off = e < R * 16 ? e * 8u : kOutOfRange;
      s1[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                  pmat_rs, off, (int)uni(op.child1_mat * (R * 128u)), 0));
      s2[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                  pmat_rs, off, (int)uni(op.child2_mat * (R * 128u)), 0));
    }
    c1.code = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(tip_rs(op.child1_clv, op.src1 == kSrcTip), s, 0, 0);
    c2.code = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(tip_rs(op.child2_clv, op.src2 == kSrcTip), s, 0, 0);
    const bool mem1 = op.src1 == kSrcMem, mem2 = op.src2 == kSrcMem;
    const __amdgpu_buffer_rsrc_t r1 = clv_rs(op.child1_clv, mem1), r2 = clv_rs(op.child2_clv, mem2);
    c1.lo = __builtin_amdgcn_raw_buffer_load_b128(r1, off_clv_ld, 0, 0);
    c1.hi = __builtin_amdgcn_raw_buffer_load_b128(r1, off_clv_ld + 16, 0, 0);
    c2.lo = __builtin_amdgcn_raw_buffer_load_b128(r2, off_clv_ld, 0, 0);
    c2.hi = __builtin_amdgcn_raw_buffer_load_b128(r2, off_clv_ld + 16, 0, 0);
    c1.sc = __builtin_amdgcn_raw_buffer_load_b32(sc_rs(op.child1_sc, mem1), s * 4u, 0, 0);
    c2.sc = __builtin_amdgcn_raw_buffer_load_b32(sc_rs(op.child2_sc, mem2), s * 4u, 0, 0);
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
  unsigned *park_sc = reinterpret_cast<unsigned *>(park_lds + (size_t)slots * 512);
  // operand of THIS operation: registers / prefetched memory / tip code / LDS slot
  auto operand = [&](unsigned src, const Child &c, const double (&o)[4], unsigned osc,
                     double (&x)[4], unsigned &xsc) {
    xsc = 0;
    if (src == kSrcReg) {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = o[k];
      xsc = osc;
    } else if (src == kSrcMem) {
      const double2 a = __builtin_bit_cast(double2, c.lo), b = __builtin_bit_cast(double2, c.hi);
      x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
      xsc = c.sc;
    } else if (src == kSrcTip) {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = ((c.code >> k) & 1u) ? 1.0 : 0.0;
    } else {
      const unsigned slot = src - kSrcPark;
      const double2 a = park_lds[(slot * 2) * 256 + tid], b = park_lds[(slot * 2 + 1) * 256 + tid];
      x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
      xsc = park_sc[slot * 256 + tid];
    }
  };

  double o[4] = {0, 0, 0, 0};       // the CLV this lane produced last
  unsigned osc = 0;
  Child c1, c2;
  {
    double s1[kStageRegs], s2[kStageRegs];
    issue(ops[0], c1, c2, s1, s2);
    // three no-op stores (0-byte descriptor): the loop body ends with three
    // stores after its loads, so entering the loop with the same sequence in
    // flight lets the compiler count its vmcnt waits past them exactly
    const __amdgpu_buffer_rsrc_t none = make_rsrc(v.clv, 0u);
    __builtin_amdgcn_raw_buffer_store_b32(0u, none, kOutOfRange, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v4u{0, 0, 0, 0}, none, kOutOfRange, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v4u{0, 0, 0, 0}, none, kOutOfRange + 16, 0, 0);
    stage_write(0, s1, s2);
  }
  // (A prefetch distance of two was tried: an older sibling may be the parent of
  // operation i, which is not stored yet when operation i+2 would fetch it;
  // one operation ahead is correct by construction.)
  for (unsigned i = 0; i < nops; ++i) {
    const unsigned buf = i & 1;
    const LevelOp op = ops[i];
    const LevelOp nx = ops[i + 1 < nops ? i + 1 : i];
    double x[4], y[4];
    unsigned xsc, ysc;
    operand(op.src1, c1, o, osc, x, xsc);
    operand(op.src2, c2, o, osc, y, ysc);
    // everything the NEXT operation needs (after the last one: a harmless repeat)
    double s1[kStageRegs], s2[kStageRegs];
    issue(nx, c1, c2, s1, s2);

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
    }
    __builtin_amdgcn_raw_buffer_store_b32(osc, sc_rs(op.parent_sc, true), off_sc_st, 0, 0);
    const __amdgpu_buffer_rsrc_t prs = clv_rs(op.parent_clv, true);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, make_double2(o[0], o[1])), prs, off_clv_st, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, make_double2(o[2], o[3])), prs, off_clv_st + 16, 0, 0);
    if (op.park) {
      const unsigned slot = op.park - 1;
      park_lds[(slot * 2) * 256 + tid] = make_double2(o[0], o[1]);
      park_lds[(slot * 2 + 1) * 256 + tid] = make_double2(o[2], o[3]);
      park_sc[slot * 256 + tid] = osc;
    }
    stage_write(buf ^ 1, s1, s2);
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

// 4-state kernel: 32-bit buffer offsets, so one CLV must stay under 2 GB
// (64 M (site, rate) pairs); beyond that the generic kernel takes over
static inline bool dna_fast_ok(const rdamd_partition *p) {
  const unsigned R = p->rate_cats;
  return p->states == 4 && p->ncodes_cap == 16 && (R == 1 || R == 2 || R == 4 || R == 8) &&
         (size_t)p->sites * R < ((size_t)1 << 26);
}

constexpr size_t kComputeUnits = 256;   // MI355X: 8 XCDs x 32 CUs
constexpr size_t kMaxParkSlots = 6;   // static + dynamic LDS stays under 64 KB per block

// Slots are sized so that every block of the launch is resident at once (the
// blocks of one CU share its 160 KB of LDS): a second dispatch round would cost
// more than the read-backs the extra slots save.
unsigned clv_traversal_slots(const rdamd_partition *p) {
  if (!dna_fast_ok(p) || p->sites == 0) return 0;
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
  const unsigned R = p->rate_cats;
  if (dna_fast_ok(p)) {
    // one (site, rate) pair per lane where possible: maximum memory-level
    // parallelism; the lane -> pair map is identical for every operation
    size_t total = (size_t)p->sites * R;
    unsigned gx = (unsigned)((total + 255) / 256);
    const size_t lds = (size_t)slots * 256 * (32 + 4);
    const unsigned pmat_bytes = (unsigned)((size_t)(p->prob_matrices + kExtraMatrices) * R * 16 * sizeof(double));
    switch (R) {
      case 1: clv_dna_traversal_kernel<1><<<gx, 256, lds, p->stream>>>(v, d_ops, nops, slots, pmat_bytes); break;
      case 2: clv_dna_traversal_kernel<2><<<gx, 256, lds, p->stream>>>(v, d_ops, nops, slots, pmat_bytes); break;
      case 4: clv_dna_traversal_kernel<4><<<gx, 256, lds, p->stream>>>(v, d_ops, nops, slots, pmat_bytes); break;
      default: clv_dna_traversal_kernel<8><<<gx, 256, lds, p->stream>>>(v, d_ops, nops, slots, pmat_bytes); break;
    }
  } else {
    unsigned gx = (p->sites + 255) / 256;
    clv_generic_traversal_kernel<<<gx, 256, 0, p->stream>>>(v, d_ops, nops);
  }
  return hipGetLastError();
}

}  // namespace rdamd
