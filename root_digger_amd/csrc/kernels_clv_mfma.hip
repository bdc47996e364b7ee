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
//
// Data layouts (all private to this library; rdamd_get_clv hands out the ABI's
// [site][rate][state]):
//   * a CLV is kept in the OPERAND layout, [rate][tile of 16 sites][2 560 B]
//     (common.hpp, k20_tile_index): a tile holds, lane by lane, exactly what the
//     wave keeps of it in registers -- 16 + 16 + 8 bytes per lane in three pieces.
//     Every CLV load and store instruction covers contiguous memory (1 KB, 1 KB,
//     512 B) and needs no transposition; D layout = B layout, so a child produced
//     by one of the two operations before is consumed straight from registers;
//   * A operands come from an MFMA-ready permutation of the P-matrices that the
//     P-matrix step writes ([matrix][rate][rg][ks][k][i]), fetched as four contiguous
//     16-byte pieces per lane (3.2 KB per child instead of 25 scattered loads),
//     parked in registers for one iteration and redistributed through a
//     wave-private LDS area (no barrier) right before the MFMAs that use them;
//   * a tip child costs no MFMA and no A copy: P . (0/1 vector) is a row of the
//     partition's tip table (tiptab[matrix][rate][code][...], rows in operand order,
//     k20_row_state: the four lanes of a site read 64 + 64 + 32 contiguous bytes).
//
// Memory pipeline.  With ~2.4 waves per SIMD on the c3 shape nothing hides a
// memory round trip, so everything operation i+1 needs (A operands, table rows,
// older-sibling CLVs, scalers) is requested at the START of operation i, and
// the result of operation i-1 is stored behind those loads.  Only what a child
// needs is requested (wave-uniform branches by its source: the CU's one address
// unit serves every wave, an instruction dropped through an empty descriptor
// still costs it a slot); the four stores of an iteration are always issued
// (out-of-range offsets where there is nothing to keep).  Tip codes and operation
// heads come through the scalar cache two operations ahead, the tip-code
// address from the head in front (LevelOp::ahead*), so that no scalar load waits
// for another one.
// Like the 4-state kernel, a whole operation list is one launch: every
// dependency is site-local and each wave owns its sites for the whole list
// (the host cuts the list where an operation reads, from memory, what the
// operation just before wrote -- partition.hip).
#include <cstdlib>
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
  unsigned ahead1, ahead2;   // tip indices of the next operation's children
};
constexpr unsigned kAheadWord = 10;
static_assert(sizeof(LevelOp) % 4 == 0 && offsetof(LevelOp, src2) == 36 &&
              offsetof(LevelOp, ahead1) == 4 * kAheadWord && offsetof(LevelOp, ahead2) == 4 * kAheadWord + 4,
              "OpHead mirrors LevelOp");
__device__ __forceinline__ OpHead load_op(const_u32_ptr ops, unsigned i) {
  const const_u32_ptr w = ops + (size_t)i * (sizeof(LevelOp) / 4);
  return OpHead{w[0], w[1], w[2], w[3], w[4], (int)w[5], (int)w[6], (int)w[7], w[8], w[9],
                w[kAheadWord], w[kAheadWord + 1]};
}

// what one child of the next operation brings from memory
struct ChildLoad {
  u32x4 raw[4];               // this lane's 16-byte pieces of the MFMA-ready copy (3200 B / 64 lanes)
  double b[kMfmaSteps];       // B operands of an older sibling, or a tip's finished term (its
                              // tip-table entries); 0 when the descriptor is empty
  unsigned sc;                // scaler of an older sibling
};
constexpr int kMfmaLdsChild = 4 * 64 * 16;   // bytes of LDS one child's A copy occupies per wave
constexpr int kMfmaLdsWave = 2 * kMfmaLdsChild;

// (which state a lane group holds as operand t: k20_state_of, common.hpp)
__host__ __device__ constexpr unsigned state_of(unsigned g, unsigned t) { return k20_state_of(g, t); }

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
    o[e] = p[state_of(i, rg) * kMfmaK + state_of(k, ks)];
  }
}

template <int MAXT, int VAR = 0>   // 64 * rate categories, rounded up to 256 or 1024
__global__ void __launch_bounds__(MAXT)
clv_k20_traversal_kernel(DeviceView v, const double *__restrict__ pmfma,
                         const LevelOp *__restrict__ ops_all, ListPieces pieces) {
  // blockIdx.y = an independent piece of the operation list (a subtree: rdamd_update_clvs cuts
  // the list so that a shape with too few 16-site tiles for the chip still fills it)
  const LevelOp *__restrict__ ops_generic = ops_all + pieces.start[blockIdx.y];
  const unsigned nops = pieces.len[blockIdx.y];
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
  // Loop-invariant per-lane offsets.  A CLV of these partitions is kept in the operand
  // layout (common.hpp, k20_tile_index): [rate][tile][piece 0: 64 lanes x 16 B][piece 1:
  // 64 x 16 B][piece 2: 64 x 8 B] -- a lane's five operands are its 16 + 16 + 8 bytes of
  // the tile's three pieces, every load and store instruction covers contiguous memory.
  // (A tip-table row has the same operands as [group][t = 0, 1] [group][t = 2, 3]
  // [group][t = 4]: k20_row_state.)
  const unsigned tiles = (S + 15u) / 16u;
  const unsigned tile_off = (uni(r * tiles) + blockIdx.x) * (16u * kMfmaK * 8u);
  const unsigned ll = grp * 16u + (ls & 15u);               // the lane whose data this lane loads (itself unless past S)
  const unsigned ld_clv = tile_off + ll * 16u;              // piece 0; piece 1 at + 1024, piece 2 at tile_off + 2048 + ll * 8
  const unsigned ld_clv4 = tile_off + 2048u + ll * 8u;
  const unsigned st_clv = site < S ? tile_off + lane * 16u : kOob;
  const unsigned st_clv4 = site < S ? tile_off + 2048u + lane * 8u : kOob;
  const unsigned tab_g = grp * 16u;                         // a tip-table row: + code * 160
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
  // the 16 tip codes of this wave's sites are 16 contiguous bytes of a tip row: they
  // come through the scalar cache (no vector-memory instruction), two operations ahead
  const unsigned long long codes_u = reinterpret_cast<unsigned long long>(v.tipcodes);
  const unsigned long long codes_lo =
      (((unsigned long long)uni((unsigned)(codes_u >> 32)) << 32) | uni((unsigned)codes_u)) +
      (unsigned long long)blockIdx.x * 16u;
  const bool upper_half = (col & 8u) != 0u;
  const unsigned code_shift = (col & 7u) * 8u;
  struct Codes { unsigned w0, w1, w2, w3; };   // (named members: an array here ends up in scratch memory)
  auto load_codes = [&](unsigned src, unsigned clv) -> Codes {   // (prologue only)
    Codes c{0u, 0u, 0u, 0u};
    if (src == 0u) {
      const const_u32_ptr row = (const_u32_ptr)(codes_lo + (unsigned long long)clv * v.tip_stride);
      c.w0 = row[0]; c.w1 = row[1]; c.w2 = row[2]; c.w3 = row[3];
    }
    return c;
  };
  // In the loop the 16 bytes are fetched unconditionally (row 0 for a child that is no
  // tip) from an address the PREVIOUS operation's head supplies (LevelOp::ahead*): no
  // branch, no scalar load waiting for another one -- nothing touches the result until
  // the table loads of the next iteration need it.
  auto load_codes_ahead = [&](unsigned tip) -> Codes {
    const const_u32_ptr row = (const_u32_ptr)(codes_lo + (unsigned long long)tip * v.tip_stride);
    return Codes{row[0], row[1], row[2], row[3]};
  };
  auto my_code = [&](const Codes &c) -> unsigned {
    // byte `col` of the 16: pick the 8-byte half, shift (a chain of selects over the
    // four words is turned into an indexed load from scratch memory by the compiler)
    const unsigned long long lo = ((unsigned long long)c.w1 << 32) | c.w0;
    const unsigned long long hi = ((unsigned long long)c.w3 << 32) | c.w2;
    return (unsigned)((upper_half ? hi : lo) >> code_shift) & 255u;
  };

  // Wave-private LDS: the A copies of both children of the NEXT operation,
  // written as loaded (contiguous), read back as [block][my role].
  extern __shared__ char a_lds_all[];
  char *a_lds = a_lds_all + (threadIdx.x >> 6) * kMfmaLdsWave;
  // What a child of the next operation needs from memory, by where it comes from
  // (wave-uniform branches; the kernel is bound by the number of vector-memory
  // instructions the CU's address unit has to process, so none is issued in vain):
  //   tip      three pieces of its tip-table row: the child's finished term
  //   memory   three pieces of its CLV, its scaler, the A copy of its matrix
  //   register (result of one of the two operations before) only the A copy
  auto load_pieces = [&](const __amdgpu_buffer_rsrc_t rs, unsigned o01, unsigned o23, unsigned o4, ChildLoad &c) {
    if (VAR & 16) return;   // VAR bit 4 (timing only): no CLV / tip-table loads
    const u32x4 p = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o01, 0, 0);
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o23, 0, 0);
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)o4, 0, 0);
    c.b[0] = as_f64(u32x2{p[0], p[1]}); c.b[1] = as_f64(u32x2{p[2], p[3]});
    c.b[2] = as_f64(u32x2{q[0], q[1]}); c.b[3] = as_f64(u32x2{q[2], q[3]});
    c.b[4] = as_f64(t);
  };
  // (Measured alternative: the same eight load instructions for every child, descriptor and
  // offsets chosen by its source, what is not needed dropped through a zero-size
  // descriptor -- every wait becomes an exact count, 20 instead of 11.6 vector-memory
  // instructions per operation: 4 - 7 % slower.)
  auto load_child = [&](unsigned src, unsigned clv, unsigned mat, int scb, const Codes &codes, ChildLoad &c) {
    if (src == 0u) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(tab_base + (size_t)(mat * R + r) * tab_slot, tab_slot);
      const unsigned row = my_code(codes) * tab_row;
      load_pieces(rs, row + tab_g, row + 64u + tab_g, row + 128u + grp * 8u, c);
      return;
    }
    if (src == 1u) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(clv_base + (size_t)(clv - v.tips) * clv_bytes, clv_bytes);
      load_pieces(rs, ld_clv, ld_clv + 1024u, ld_clv4, c);
      c.sc = 0u;
      if (scb >= 0) {
        const __amdgpu_buffer_rsrc_t sc_rs = make_rsrc(sc_base + (size_t)scb * sc_bytes, sc_bytes);
        c.sc = __builtin_amdgcn_raw_buffer_load_b32(sc_rs, (int)ld_sc, 0, 0);
      }
    }
    if (VAR & 8) return;   // VAR bit 3 (timing only): no A loads
    // one (matrix, rate) copy = 3200 contiguous bytes; pieces past its end are dropped
    const __amdgpu_buffer_rsrc_t rs =
        make_rsrc(reinterpret_cast<const char *>(pmfma) + (size_t)(mat * R + r) * (kMfmaCopy * 8), kMfmaCopy * 8);
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
  op.src1 = op.src2 = 2u;      // iteration -1 computes nothing: it only issues operation 0's loads
  // tip codes of operation oi + 1 (requested an iteration ago; here: up front)
  Codes cw1 = load_codes(nx.src1, nx.child1_clv), cw2 = load_codes(nx.src2, nx.child2_clv);

  // Always the same four instructions (a scaler that is not kept, or the iterations before
  // there is a result, go to out-of-range offsets and are dropped by the hardware): with a
  // fixed number of stores behind an iteration's loads the wait for those loads is a count
  // (vmcnt(4)) and never a drain that would include the stores' acknowledgements.
  auto store_result = [&](const OpHead &h, const double (&val)[kMfmaGroups], unsigned sc, bool wanted) {
    const bool keep_sc = h.parent_sc >= 0;
    const __amdgpu_buffer_rsrc_t psc_rs =
        make_rsrc(sc_base + (size_t)(keep_sc ? (unsigned)h.parent_sc : 0u) * sc_bytes, sc_bytes);
    __builtin_amdgcn_raw_buffer_store_b32(sc, psc_rs, (int)(wanted && keep_sc ? st_sc : kOob), 0, 0);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(clv_base + (size_t)(h.parent_clv - v.tips) * clv_bytes, clv_bytes);
    const u32x2 x0 = __builtin_bit_cast(u32x2, val[0]), x1 = __builtin_bit_cast(u32x2, val[1]);
    const u32x2 x2 = __builtin_bit_cast(u32x2, val[2]), x3 = __builtin_bit_cast(u32x2, val[3]);
    constexpr int aux = (VAR & 2) ? 0 : 2;   // nt (streamed): VAR bit 1 = default policy
    const unsigned o = wanted ? st_clv : kOob, o4 = wanted ? st_clv4 : kOob;
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{x0[0], x0[1], x1[0], x1[1]}, rs, (int)o, 0, aux);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{x2[0], x2[1], x3[0], x3[1]}, rs, (int)(o + 1024u), 0, aux);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, val[4]), rs, (int)o4, 0, aux);
  };

  for (int oi = -1; oi < (int)nops; ++oi) {
    const bool live = oi >= 0;
    const unsigned nx2i = (unsigned)(oi + 2) < nops ? (unsigned)(oi + 2) : nops - 1;
    const OpHead nx2 = load_op(ops, nx2i);   // wanted one iteration from now
    // tip codes two operations ahead: they address next iteration's table loads
    const Codes ncw1 = load_codes_ahead(nx.ahead1), ncw2 = load_codes_ahead(nx.ahead2);
    // this operation's operands: what the loads brought, or the registers of one of
    // the two operations before
    const bool tip1 = op.src1 == 0u, tip2 = op.src2 == 0u;
    double b1[kMfmaSteps], b2[kMfmaSteps];
    unsigned sc1 = c1.sc, sc2 = c2.sc;
#pragma unroll
    for (int s = 0; s < kMfmaSteps; ++s) { b1[s] = c1.b[s]; b2[s] = c2.b[s]; }
    if (op.src1 == 2u) {
#pragma unroll
      for (int s = 0; s < kMfmaSteps; ++s) b1[s] = res[s];
      sc1 = osc;
    } else if (op.src1 == 3u) {
#pragma unroll
      for (int s = 0; s < kMfmaSteps; ++s) b1[s] = prev[s];
      sc1 = oscp;
    } else if (tip1) {
      sc1 = 0u;
    }
    if (op.src2 == 2u) {
#pragma unroll
      for (int s = 0; s < kMfmaSteps; ++s) b2[s] = res[s];
      sc2 = osc;
    } else if (op.src2 == 3u) {
#pragma unroll
      for (int s = 0; s < kMfmaSteps; ++s) b2[s] = prev[s];
      sc2 = oscp;
    } else if (tip2) {
      sc2 = 0u;
    }
    // this operation's A copies, requested a whole iteration ago, go to LDS ...
    if (!tip1 && !(VAR & 8)) stage_child_a(0, c1);
    if (!tip2 && !(VAR & 8)) stage_child_a(1, c2);
    // ... everything the NEXT operation needs from memory is requested ...
    if ((unsigned)(oi + 1) < nops) {
      load_child(nx.src1, nx.child1_clv, nx.child1_mat, nx.child1_sc, cw1, c1);
      load_child(nx.src2, nx.child2_clv, nx.child2_mat, nx.child2_sc, cw2, c2);
    }
    // ... and the PREVIOUS operation's result goes out behind those loads
    if (!(VAR & 1)) store_result(pop, res, osc, oi >= 1);   // VAR bit 0 (timing only): no stores
    // the MFMAs of this operation
    // (a tip's loads brought its finished term: b = P . indicator, from the tip table)
    double d1[kMfmaGroups], d2[kMfmaGroups], out[kMfmaGroups];
    if (tip1 || (VAR & 4)) {   // VAR bit 2 (timing only): no LDS reads, no MFMAs
#pragma unroll
      for (int t = 0; t < kMfmaGroups; ++t) d1[t] = b1[t];
    } else {
      child_product(0, b1, d1);
    }
    if (tip2 || (VAR & 4)) {
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
    cw1 = ncw1;
    cw2 = ncw2;
  }
  if (nops >= 1 && !(VAR & 1)) store_result(pop, res, osc, true);   // the last operation's result
}

hipError_t launch_pmat_to_mfma(rdamd_partition *p, const unsigned *d_matrix_indices,
                               unsigned count) {
  if (!count) return hipSuccess;
  pmat_to_mfma_kernel<<<count * p->rate_cats, 256, 0, p->stream>>>(
      p->d_pmat, p->d_pmat_mfma, d_matrix_indices, count, p->rate_cats);
  return hipGetLastError();
}

size_t k20_mfma_copy_doubles() { return kMfmaCopy; }

// How a list is cut for this kernel.  Round 3: 8 pieces, one level (c3, 625 blocks per row: 593 -> 494 us).
// Round 5, with the levels of list_levels: about 5 000 blocks per launch -- c3 keeps its 8 pieces (16: 514 - 520 us,
// 32: 527 - 566), c3's tree on 1 000 sites (63 blocks per row) 227 -> 135 us with 32 pieces in three launches
// (profiles/r5_clv_pieces_ab.txt, section 5).
void clv_k20_traversal_cut(const rdamd_partition *p, unsigned count, unsigned *pieces, unsigned *small, unsigned *min_count) {
  const unsigned blocks = (p->sites + 15) / 16;
  *pieces = std::min(std::max(5000u / std::max(blocks, 1u), 8u), kMaxListPieces);
  *small = 24;
  *min_count = count < 24 ? 24 : 12;   // (lists under 24 operations run whole, the joining lists of a cut from 12)
#ifdef RDAMD_ABLATION
  if (getenv("RDAMD_CLV_PIECES")) *pieces = (unsigned)atoi(getenv("RDAMD_CLV_PIECES"));
#endif
  *pieces = std::min(*pieces, kMaxListPieces);
}

hipError_t launch_clv_k20_traversal(rdamd_partition *p, const LevelOp *d_ops, const ListPieces &pieces) {
  if (pieces.n == 0 || p->sites == 0) return hipSuccess;
  DeviceView v = p->view();
  const dim3 grid((p->sites + 15) / 16, pieces.n);
  const size_t lds = (size_t)p->rate_cats * kMfmaLdsWave;   // 12 KB per wave
#ifdef RDAMD_ABLATION
  // timing-only variants (stores / loads / arithmetic switched off: results are garbage).
  // They exist only in the ablation build (`make ablation` -> ../lib/librdamd_ablation.so,
  // used by profiles/k20_ab.sh); the product library ignores RDAMD_K20_VAR.
  static const int var = getenv("RDAMD_K20_VAR") ? atoi(getenv("RDAMD_K20_VAR")) : 0;
#define RDAMD_K20_CASE(V) case V: clv_k20_traversal_kernel<256, V><<<grid, 64 * p->rate_cats, lds, p->stream>>>(v, p->d_pmat_mfma, d_ops, pieces); break;
  if (p->rate_cats <= 4)
    switch (var) {
      RDAMD_K20_CASE(1) RDAMD_K20_CASE(2) RDAMD_K20_CASE(4) RDAMD_K20_CASE(5) RDAMD_K20_CASE(8)
      RDAMD_K20_CASE(13) RDAMD_K20_CASE(16) RDAMD_K20_CASE(17) RDAMD_K20_CASE(29) RDAMD_K20_CASE(31)
      default: clv_k20_traversal_kernel<256><<<grid, 64 * p->rate_cats, lds, p->stream>>>(v, p->d_pmat_mfma, d_ops, pieces);
    }
#else
  if (p->rate_cats <= 4)
    clv_k20_traversal_kernel<256><<<grid, 64 * p->rate_cats, lds, p->stream>>>(v, p->d_pmat_mfma, d_ops, pieces);
#endif
  else
    clv_k20_traversal_kernel<1024><<<grid, 64 * p->rate_cats, lds, p->stream>>>(v, p->d_pmat_mfma, d_ops, pieces);
  return hipGetLastError();
}

}  // namespace rdamd
