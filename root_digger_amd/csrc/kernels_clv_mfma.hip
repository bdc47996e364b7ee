// 20-state CLV updates on the FP64 matrix cores (v_mfma_f64_4x4x4_4b_f64).
//
// Same contract as kernels_clv.hip (replaces corax_update_clvs,
// /root/reference/src/model.cpp:402, :440, :461, :851) for the protein shape
// of BASELINE config c3.  For one rate category the child term of an operation
// is a small GEMM,  T[i][s] = sum_j P[i][j] * c[s][j].
//
// Which MFMA: profiles/micro/mfma_f64_rate.hip measures v_mfma_f64_16x16x4_f64
// at 112-210 cycles per instruction (<= 45 TFLOP/s) but the four-block
// v_mfma_f64_4x4x4_4b_f64 at 17-18 cycles (70-73 TFLOP/s), and 20 = 5 x 4
// wastes nothing where 20 -> 32 rows of the 16-wide tile wasted 37 %.  So the
// product is tiled as 5 row groups x 5 k-steps of 4x4x4 blocks:
//     A = P[4 rg + i][4 ks + k]   the same 4x4 block in all four MFMA blocks,
//     B = c[site][4 ks + k]       block b = sites 4b .. 4b+3 of the wave's 16,
//     D = T[4 rg + i][site]       one double per lane,
// 25 MFMAs per child per 16 sites, five independent accumulator chains.
// Lane layout (profiles/micro/mfma_f64_4x4x4_layout.hip): with col = lane % 16
// and grp = lane / 16,  A: i = col % 4, k = grp;  B: site = col, k = grp;
// D: site = col, i = grp.  So a lane holds states {4 s + grp} of its site both
// as a B operand and as a result: a CLV never needs a transpose and a child
// produced by the operation just before is consumed straight from the D
// registers.
//
// One wave = (16 sites, one rate); the waves of a workgroup are the R rates of
// the same 16 sites, so the per-site "all entries < 2^-256" rule is one ballot
// folded on the scalar unit plus one LDS exchange (one barrier per operation).
// A operands come from an MFMA-ready permutation of the P-matrices that the
// P-matrix step writes ([matrix][rate][rg][ks][k][i]: the 16 values of one MFMA
// are one cache line).  A tip child costs no MFMA at all: P . (0/1 vector) is a
// row of the partition's tip table (tiptab[matrix][rate][code][state], written by
// the P-matrix step), and a lane's five entries {4 s + grp} of that row arrive
// through the same five 8-byte loads an older sibling's CLV would use -- only the
// descriptor and the per-lane offset differ.  Its A copy is not fetched (empty
// descriptor), staged or multiplied (wave-uniform branch around LDS + MFMA work
// only, so the vector-memory instruction stream stays fixed).  Tip codes are
// requested two operations ahead so that the table row can be requested one ahead.
//
// Memory pipeline.  With ~2.4 waves per SIMD on the c3 shape nothing hides a
// memory round trip, so everything operation i+1 needs (A operands, tip codes,
// older-sibling CLVs, scalers) is requested at the START of operation i, a
// whole operation ahead.  The A operands (3.2 KB per child) are fetched as
// contiguous 16-byte pieces (4 loads per child instead of 25), parked in
// registers for one iteration and redistributed through a wave-private LDS
// area (no barrier) right before the MFMAs that use them.  Vector memory
// operations retire in issue order, so every one of them is issued
// unconditionally through buffer descriptors -- a load that is not needed, or
// a store of a lane past the last site, gets a zero-size descriptor or an
// out-of-range offset and is dropped by the hardware.  The instruction stream
// per operation is therefore fixed, the compiler's `s_waitcnt vmcnt(N)` are
// exact (never below the six stores an iteration ends with), and no wave waits
// for its own CLV stores.
// Like the 4-state kernel, a whole operation list is one launch: every
// dependency is site-local and each wave owns its sites for the whole list
// (the host cuts the list where an operation reads, from memory, what the
// operation just before wrote -- partition.hip).
#include "common.hpp"

namespace rdamd {

namespace {

constexpr int kMfmaK = 20;        // states
constexpr int kMfmaSteps = 5;     // k steps of 4
constexpr int kMfmaGroups = 5;    // row groups of 4
constexpr int kMfmaBlocks = kMfmaGroups * kMfmaSteps;   // 4x4 blocks of P
constexpr int kMfmaCopy = kMfmaBlocks * 16;             // doubles per (matrix, rate): a permutation of P
constexpr unsigned kOob = 0x80000000u;     // offset no descriptor here reaches

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) unsigned *const_u32_ptr;

__device__ __forceinline__ unsigned uni(unsigned x) {
  return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}
// buffer descriptor over [p, p + bytes); bytes == 0 drops every access
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = uni((unsigned)u), hi = uni((unsigned)(u >> 32));
  void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)uni(bytes), 0x00020000);
}
__device__ __forceinline__ double as_f64(u32x2 w) { return __builtin_bit_cast(double, w); }

// the leading 32-bit fields of a LevelOp (common.hpp), read through the scalar cache
struct OpHead {
  unsigned parent_clv, child1_clv, child2_clv, child1_mat, child2_mat;
  int parent_sc, child1_sc, child2_sc;
  unsigned src1, src2;
};
static_assert(sizeof(LevelOp) % 4 == 0 && offsetof(LevelOp, src2) == 36, "OpHead mirrors LevelOp");
__device__ __forceinline__ OpHead load_op(const_u32_ptr ops, unsigned i) {
  const const_u32_ptr w = ops + (size_t)i * (sizeof(LevelOp) / 4);
  return OpHead{w[0], w[1], w[2], w[3], w[4], (int)w[5], (int)w[6], (int)w[7], w[8], w[9]};
}

// what one child of the next operation brings from memory
struct ChildLoad {
  u32x4 raw[4];               // this lane's 16-byte pieces of the MFMA-ready copy (3200 B / 64 lanes)
  double b[kMfmaSteps];       // B operands of an older sibling, or a tip's finished term (its
                              // tip-table entries); 0 when the descriptor is empty
  unsigned sc;                // scaler of an older sibling
};
constexpr int kMfmaLdsChild = 4 * 64 * 16;   // bytes of LDS one child's A copy occupies per wave

}  // namespace

// MFMA-ready copy of one 20x20 P-matrix: element (rg, ks, k, i) = P[4 rg + i][4 ks + k], so
// the 16 values one MFMA's A operand needs are 128 contiguous bytes (one cache line)
__global__ void __launch_bounds__(256)
pmat_to_mfma_kernel(const double *__restrict__ pmat, double *__restrict__ out,
                    const unsigned *__restrict__ mat_idx, unsigned count, unsigned R) {
  const unsigned slot_in_list = blockIdx.x / R, r = blockIdx.x % R;
  if (slot_in_list >= count) return;
  const size_t slot = (size_t)mat_idx[slot_in_list] * R + r;
  const double *p = pmat + slot * (kMfmaK * kMfmaK);
  double *o = out + slot * kMfmaCopy;
  for (unsigned e = threadIdx.x; e < (unsigned)kMfmaCopy; e += blockDim.x) {
    const unsigned i = e & 3, k = (e >> 2) & 3, blk = e >> 4, ks = blk % kMfmaSteps, rg = blk / kMfmaSteps;
    o[e] = p[(4 * rg + i) * kMfmaK + 4 * ks + k];
  }
}

template <int MAXT>   // 64 * rate categories, rounded up to 256 or 1024
__global__ void __launch_bounds__(MAXT)
clv_k20_traversal_kernel(DeviceView v, const double *__restrict__ pmfma,
                         const LevelOp *__restrict__ ops_generic, unsigned nops) {
  // flags[parity][rate]: bit c = "this rate's 20 entries of site c are all < 2^-256";
  // double-buffered by operation parity so one barrier per operation suffices
  __shared__ unsigned flags[2][16];
  const unsigned R = v.rate_cats, S = v.sites;
  const unsigned lane = threadIdx.x & 63, r = uni(threadIdx.x >> 6);   // wave = rate
  const unsigned col = lane & 15, grp = lane >> 4;
  const unsigned site = blockIdx.x * 16 + col;
  const unsigned ls = site < S ? site : S - 1;   // clamped for loads

  const bool sc_lane = r == 0 && grp == 0;       // the lanes that own the per-site scalers
  const unsigned clv_bytes = uni((unsigned)(v.clv_stride * sizeof(double)));
  const unsigned sc_bytes = S * 4u;
  // loop-invariant per-lane offsets
  const unsigned ld_clv = ((ls * R + r) * kMfmaK + grp) * 8u;                 // + 32 s
  const unsigned st_clv = site < S ? ld_clv : kOob;
  const unsigned ld_sc = sc_lane ? ls * 4u : kOob;
  const unsigned st_sc = (sc_lane && site < S) ? site * 4u : kOob;
  const unsigned a_off = (grp * 4 + (col & 3)) * 8u;                          // my element of every 4x4 block
  const unsigned long long ops_u = reinterpret_cast<unsigned long long>(ops_generic);
  const const_u32_ptr ops =   // wave-uniform, constant address space: op heads come by s_load
      (const_u32_ptr)(((unsigned long long)uni((unsigned)(ops_u >> 32)) << 32) | uni((unsigned)ops_u));
  const char *clv_base = reinterpret_cast<const char *>(v.clv);
  const char *sc_base = reinterpret_cast<const char *>(v.scaler);
  const char *tab_base = reinterpret_cast<const char *>(v.tiptab);
  const unsigned tab_row = kMfmaK * 8u;                              // bytes per code
  const unsigned tab_slot = uni(v.ncodes_cap * tab_row);             // bytes per (matrix, rate)

  // Wave-private LDS: the A copies of both children of the NEXT operation,
  // written as loaded (contiguous), read back as [block][my role].
  extern __shared__ char a_lds_all[];
  char *a_lds = a_lds_all + (threadIdx.x >> 6) * (2 * kMfmaLdsChild);

  // The loads a child of the next operation may need: the descriptors of the
  // ones it does not need are empty, so the instruction stream never changes.
  // (B side first, A pieces last: the B side is wanted first.)
  auto load_code = [&](unsigned src, unsigned clv) -> unsigned {
    const bool tip = src == 0u;
    const __amdgpu_buffer_rsrc_t code_rs =
        make_rsrc(v.tipcodes + (size_t)(tip ? clv : 0u) * v.tip_stride, tip ? v.tip_stride : 0u);
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(code_rs, (int)ls, 0, 0);
  };
  auto load_child_b = [&](unsigned src, unsigned clv, unsigned mat, int scb, unsigned code, ChildLoad &c) {
    const bool tip = src == 0u, mem = src == 1u;
    // tip: row `code` of the (matrix, rate) tip table; memory: the CLV; else nothing
    const char *base = tip ? tab_base + (size_t)(mat * R + r) * tab_slot
                           : clv_base + (size_t)(mem ? clv - v.tips : 0u) * clv_bytes;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, tip ? tab_slot : (mem ? clv_bytes : 0u));
    const unsigned off = tip ? (code & 255u) * tab_row + grp * 8u : ld_clv;
#pragma unroll
    for (int s = 0; s < kMfmaSteps; ++s)
      c.b[s] = as_f64(__builtin_amdgcn_raw_buffer_load_b64(rs, (int)(off + 32u * s), 0, 0));
    const bool has_sc = mem && scb >= 0;
    const __amdgpu_buffer_rsrc_t sc_rs =
        make_rsrc(sc_base + (size_t)(has_sc ? scb : 0) * sc_bytes, has_sc ? sc_bytes : 0u);
    c.sc = __builtin_amdgcn_raw_buffer_load_b32(sc_rs, (int)ld_sc, 0, 0);
  };
  auto load_child_a = [&](unsigned src, unsigned mat, ChildLoad &c) {
    // one (matrix, rate) copy = 3200 contiguous bytes; pieces past its end are dropped;
    // a tip child needs none of it
    const __amdgpu_buffer_rsrc_t rs =
        make_rsrc(reinterpret_cast<const char *>(pmfma) + (size_t)(mat * R + r) * (kMfmaCopy * 8),
                  src == 0u ? 0u : kMfmaCopy * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      c.raw[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lane * 16u + 1024u * k), 0, 0);
  };
  auto stage_child_a = [&](int child, const ChildLoad &c) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      *reinterpret_cast<u32x4 *>(a_lds + child * kMfmaLdsChild + lane * 16u + 1024u * k) = c.raw[k];
  };
  // the five accumulator chains of one child: D[rg] = sum_ks A[rg][ks] . B[ks]
  auto child_product = [&](int child, const double (&b)[kMfmaSteps], double (&d)[kMfmaGroups]) {
    const char *ap = a_lds + child * kMfmaLdsChild + a_off;
    double a[kMfmaBlocks];
#pragma unroll
    for (int j = 0; j < kMfmaBlocks; ++j) a[j] = *reinterpret_cast<const double *>(ap + 128 * j);
#pragma unroll
    for (int t = 0; t < kMfmaGroups; ++t) {
      d[t] = 0.0;
#pragma unroll
      for (int s = 0; s < kMfmaSteps; ++s)
        d[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t * kMfmaSteps + s], b[s], d[t], 0, 0, 0);
    }
  };

  // Iteration -1 only issues operation 0's loads (its arithmetic runs on zeros
  // and its stores go to empty descriptors): the loop header then has a single
  // memory-queue state, so the compiler's vmcnt waits are exact.
  ChildLoad c1, c2;
#pragma unroll
  for (int k = 0; k < 4; ++k) c1.raw[k] = c2.raw[k] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int s = 0; s < kMfmaSteps; ++s) c1.b[s] = c2.b[s] = 0.0;
  c1.sc = c2.sc = 0u;
  // The results of the last two operations stay in registers (D layout = B
  // layout): `res` belongs to operation oi - 1, `prev` to operation oi - 2.
  // Stores are issued one operation LATE and behind the loads of the same
  // iteration: a load then only ever queues behind stores that are a whole
  // iteration old (vector memory operations return in order, and a store's
  // acknowledgement takes as long as an operation).  The price is that an
  // operation may not read from memory what the two operations before it
  // produced: those children are forwarded from `res` (source 2) and `prev`
  // (source 3), or the host cuts the list there.
  double res[kMfmaGroups], prev[kMfmaGroups];
  unsigned osc = 0, oscp = 0;
#pragma unroll
  for (int t = 0; t < kMfmaGroups; ++t) res[t] = prev[t] = 0.0;
  OpHead op = load_op(ops, 0);
  OpHead nx = op;              // operation oi + 1 (for oi = -1: operation 0)
  OpHead pop = op;             // operation oi - 1: its stores are still due
  op.src1 = op.src2 = 2u;      // iteration -1 computes nothing: no tip term, no A copy
  // tip codes of operation oi + 1 (requested an iteration ago; here: up front)
  unsigned code1 = load_code(nx.src1, nx.child1_clv), code2 = load_code(nx.src2, nx.child2_clv);

  auto store_result = [&](const OpHead &h, bool valid, const double (&val)[kMfmaGroups], unsigned sc) {
    const bool scaled = valid && h.parent_sc >= 0;
    const __amdgpu_buffer_rsrc_t psc_rs =
        make_rsrc(sc_base + (size_t)(scaled ? h.parent_sc : 0) * sc_bytes, scaled ? sc_bytes : 0u);
    __builtin_amdgcn_raw_buffer_store_b32(sc, psc_rs, (int)st_sc, 0, 0);
    const __amdgpu_buffer_rsrc_t pclv_rs =
        make_rsrc(clv_base + (size_t)(h.parent_clv - v.tips) * clv_bytes, valid ? clv_bytes : 0u);
#pragma unroll
    for (int t = 0; t < kMfmaGroups; ++t)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, val[t]), pclv_rs,
                                            (int)(st_clv + 32u * t), 0, 0);
  };

  for (int oi = -1; oi < (int)nops; ++oi) {
    const bool live = oi >= 0;
    const unsigned nx2i = (unsigned)(oi + 2) < nops ? (unsigned)(oi + 2) : nops - 1;
    const OpHead nx2 = load_op(ops, nx2i);   // wanted one iteration from now
    // tip codes two operations ahead: they address next iteration's table loads
    const unsigned ncode1 = load_code(nx2.src1, nx2.child1_clv);
    const unsigned ncode2 = load_code(nx2.src2, nx2.child2_clv);
    // B operands of an inner child without a branch: what the loads brought (zeros
    // for a forwarded child), or the registers of one of the two operations before
    const bool tip1 = op.src1 == 0u, tip2 = op.src2 == 0u;
    double b1[kMfmaSteps], b2[kMfmaSteps];
    {
      const unsigned ra1 = op.src1 == 2u ? ~0u : 0u, ra2 = op.src2 == 2u ? ~0u : 0u;
      const unsigned rb1 = op.src1 == 3u ? ~0u : 0u, rb2 = op.src2 == 3u ? ~0u : 0u;
#pragma unroll
      for (int s = 0; s < kMfmaSteps; ++s) {
        const u32x2 l1 = __builtin_bit_cast(u32x2, c1.b[s]), l2 = __builtin_bit_cast(u32x2, c2.b[s]);
        const u32x2 p = __builtin_bit_cast(u32x2, res[s]), q = __builtin_bit_cast(u32x2, prev[s]);
        const u32x2 w1 = {l1[0] | (p[0] & ra1) | (q[0] & rb1), l1[1] | (p[1] & ra1) | (q[1] & rb1)};
        const u32x2 w2 = {l2[0] | (p[0] & ra2) | (q[0] & rb2), l2[1] | (p[1] & ra2) | (q[1] & rb2)};
        b1[s] = as_f64(w1);
        b2[s] = as_f64(w2);
      }
    }
    // (an empty descriptor loaded 0)
    const unsigned sc1 = op.src1 == 2u ? osc : (op.src1 == 3u ? oscp : c1.sc);
    const unsigned sc2 = op.src2 == 2u ? osc : (op.src2 == 3u ? oscp : c2.sc);
    // this operation's A copies, requested a whole iteration ago, go to LDS ...
    if (!tip1) stage_child_a(0, c1);
    if (!tip2) stage_child_a(1, c2);
    // ... everything the NEXT operation needs from memory is requested ...
    load_child_b(nx.src1, nx.child1_clv, nx.child1_mat, nx.child1_sc, code1, c1);
    load_child_b(nx.src2, nx.child2_clv, nx.child2_mat, nx.child2_sc, code2, c2);
    load_child_a(nx.src1, nx.child1_mat, c1);
    load_child_a(nx.src2, nx.child2_mat, c2);
    // ... and the PREVIOUS operation's result goes out behind those loads
    store_result(pop, oi >= 1, res, osc);
    // the MFMAs of this operation
    // (a tip's loads brought its finished term: b = P . indicator, from the tip table)
    double d1[kMfmaGroups], d2[kMfmaGroups], out[kMfmaGroups];
    if (tip1) {
#pragma unroll
      for (int t = 0; t < kMfmaGroups; ++t) d1[t] = b1[t];
    } else {
      child_product(0, b1, d1);
    }
    if (tip2) {
#pragma unroll
      for (int t = 0; t < kMfmaGroups; ++t) d2[t] = b2[t];
    } else {
      child_product(1, b2, d2);
    }
    bool small = true;
#pragma unroll
    for (int t = 0; t < kMfmaGroups; ++t) {
      out[t] = d1[t] * d2[t];
      small = small && (out[t] < kScaleThreshold);
    }
    // across the 4 lane groups that hold the other rows of a site: fold the
    // ballot on the scalar unit (bit c of the result = site c of this wave)
    unsigned long long bm = __ballot(small);
    bm &= bm >> 32;
    bm &= bm >> 16;
    if (lane == 0) flags[oi & 1][r] = (unsigned)bm & 0xFFFFu;
    __syncthreads();
    const bool scaled_buffer = live && op.parent_sc >= 0;
    unsigned all_bits = scaled_buffer ? 0xFFFFu : 0u;
    for (unsigned q = 0; q < R; ++q) all_bits &= flags[oi & 1][q];
    const bool all_small = (all_bits >> col) & 1u;
    const double f = all_small ? kScaleFactor : 1.0;
    oscp = osc;
    osc = scaled_buffer ? sc1 + sc2 + (all_small ? 1u : 0u) : 0u;
#pragma unroll
    for (int t = 0; t < kMfmaGroups; ++t) {
      prev[t] = res[t];
      res[t] = out[t] * f;
    }
    pop = op;
    op = nx;
    nx = nx2;
    code1 = ncode1;
    code2 = ncode2;
  }
  store_result(pop, nops >= 1, res, osc);   // the last operation's result
}

hipError_t launch_pmat_to_mfma(rdamd_partition *p, const unsigned *d_matrix_indices,
                               unsigned count) {
  if (!count) return hipSuccess;
  pmat_to_mfma_kernel<<<count * p->rate_cats, 256, 0, p->stream>>>(
      p->d_pmat, p->d_pmat_mfma, d_matrix_indices, count, p->rate_cats);
  return hipGetLastError();
}

// one CLV must stay under 2 GB (32-bit buffer offsets, kOob above every offset)
bool k20_mfma_ok(const rdamd_partition *p) {
  return p->d_pmat_mfma != nullptr &&
         (size_t)p->sites * p->rate_cats * kMfmaK * sizeof(double) < ((size_t)1 << 31) &&
         (size_t)p->prob_matrices * p->rate_cats * kMfmaCopy * sizeof(double) < ((size_t)1 << 31);
}
size_t k20_mfma_copy_doubles() { return kMfmaCopy; }

hipError_t launch_clv_k20_traversal(rdamd_partition *p, const LevelOp *d_ops, unsigned nops) {
  if (nops == 0 || p->sites == 0) return hipSuccess;
  DeviceView v = p->view();
  const unsigned gx = (p->sites + 15) / 16;
  const size_t lds = (size_t)p->rate_cats * 2 * kMfmaLdsChild;   // 8 KB per wave
  if (p->rate_cats <= 4)
    clv_k20_traversal_kernel<256><<<gx, 64 * p->rate_cats, lds, p->stream>>>(v, p->d_pmat_mfma, d_ops, nops);
  else
    clv_k20_traversal_kernel<1024><<<gx, 64 * p->rate_cats, lds, p->stream>>>(v, p->d_pmat_mfma, d_ops, nops);
  return hipGetLastError();
}

}  // namespace rdamd
