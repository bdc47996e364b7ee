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
// read consecutive 32-byte (DNA) records.  Tips are never expanded to CLVs in
// memory: a tip child is its 1-byte code per site (4-state kernel: turned into
// a 0/1 vector in registers; generic kernel: a row of the tip table built next
// to the P-matrices).
//
// A whole operation list runs in ONE launch.  Every dependency of the
// traversal is site-local -- parent[s][r] needs only child[s][r] -- and each
// lane owns one (site, rate) pair for the whole list.  A child produced by the
// operation just before stays in the lane's registers, an older sibling waits
// in an LDS parking slot (host-side liveness analysis in rdamd_update_clvs);
// every CLV is still written (the reference's state contract).  What is left
// on the vector-memory pipe is an almost pure store stream -- see the comment
// in clv_dna_traversal_kernel for why that matters.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>

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
// Buffer descriptor over [p, p+bytes): lanes whose offset lies outside are
// dropped by the hardware, so stores need no branch around them.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = uni((unsigned)u), hi = uni((unsigned)(u >> 32));
  void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)uni(bytes), 0x00020000);
}
constexpr unsigned kOutOfRange = 0x80000000u;   // offset no descriptor here reaches
#if defined(RDAMD_ABLATION) && defined(RDAMD_CLV_STORE_AUX)
constexpr int kClvStoreAux = RDAMD_CLV_STORE_AUX;
#else
constexpr int kClvStoreAux = 0;
#endif

// read-only data through the scalar cache: a load from the constant address
// space at a wave-uniform address is an s_load
typedef const __attribute__((address_space(4))) unsigned *const_u32_ptr;

// The tip codes of the 64/R sites of one wave for one tip, fetched by scalar loads.
template <int NW>
struct TipCodes {
  unsigned w[NW];
};

// Operations per staging chunk: the block stages the P-matrices of kChunk
// operations at a time (double-buffered, one barrier per chunk).
constexpr unsigned kChunk = 4;

template <int R>
__global__ void __launch_bounds__(256)
clv_dna_traversal_kernel(DeviceView v, const LevelOp *__restrict__ all_ops, ListPieces pieces,
                         unsigned slots) {
  // Independent PIECES of the list side by side (grid.y; round 5): all of c2's waves are resident
  // from the start, three per SIMD, so a launch lasts as long as ONE wave needs for its whole
  // list -- and a wave that waits (LDS reads, scalar loads, its turn on the SIMD) does not store
  // (profiles/micro/clv_store_pattern.hip: the stores alone 126 us, with half a microsecond
  // between a wave's stores 143 - 161 us).  Subtrees of the list are independent chains: the
  // host cuts it (k20_split.hpp), the pieces run as the rows of this launch, the few operations
  // that join them follow as a second launch.
  const LevelOp *__restrict__ ops = all_ops + pieces.start[blockIdx.y];
  const unsigned nops = pieces.len[blockIdx.y];
  // Why the memory pipe sees (almost) only stores inside the loop: vector-memory
  // operations retire in issue order, so a load queued behind the CLV stores of
  // the previous operation cannot return before they are acknowledged, and with
  // three waves per SIMD that round trip is exposed on every operation.  So:
  //   * P-matrices are staged by the whole block kChunk operations at a time,
  //     loaded one chunk ahead;
  //   * tip codes come through the SCALAR cache (s_load, its own counter);
  //   * a child produced by the previous operation stays in registers, an older
  //     sibling waits in an LDS parking slot (host: rdamd_update_clvs);
  //   * only a sibling that found no slot is read back (one operation ahead).
  // (a rate's matrix starts kRateStride doubles after the previous one: with a
  // stride of 16 the R lanes of a site would hit the same LDS banks on every read)
  constexpr unsigned kRateStride = 18;
  __shared__ double smat[2][kChunk][2][R * kRateStride];
  // LDS parking: `slots` CLVs (+ scaler counts) per lane, [slot][half][lane] so
  // every 16-byte access of a wave is conflict-free.
  extern __shared__ double2 park_lds[];
  const unsigned tid = threadIdx.x, lane = tid & 63;
  const unsigned S = v.sites;
  const unsigned total = S * R;                      // < 2^26 (launch_clv_traversal)
  const unsigned idx = blockIdx.x * 256 + tid;       // one (site, rate) pair per lane
  const bool active = idx < total;
  const unsigned cidx = active ? idx : total - 1;    // clamped for loads
  const unsigned s = cidx / R, r = cidx % R;

  const unsigned clv_bytes = (unsigned)(v.clv_stride * sizeof(double));
  const unsigned off_sc_st = (active && r == 0) ? s * 4u : kOutOfRange;
  // Stores.  A lane's result is a 32-byte record, written as the lane's own two 16-byte halves: a
  // store instruction touches half of every 64-byte sector of the wave's 2 KB (round 4: the store
  // stream ran at 3.99 TB/s of real HBM traffic, 62 % of what plain stores reach).  Round 5 tried
  // whole sectors: the four lanes of a QUAD exchange halves (DPP quad_perm: no LDS) so that one
  // instruction writes the quad's records 0 and 1 -- 64 contiguous bytes -- and the other its
  // records 2 and 3 (ablation builds, -DRDAMD_ABL_QUAD_STORES; bit-identical CLVs).  Same box,
  // profiles/r5_clv_store_ab.txt: c2 176.7 -> 174.2 us (+1.4 %), c5's shard +0.7 %, 125.phy
  // 132.3 -> 136.8 us (-3.4 %): the sector fill of a store instruction is not what holds the
  // stream back.  RDAMD_CLV_STORE_AUX: the cache-policy bits of the stores, for the same A/B.
#if !(defined(RDAMD_ABLATION) && defined(RDAMD_ABL_QUAD_STORES))
  const unsigned off_clv_st = active ? idx * 32u : kOutOfRange;
#else
  const unsigned ql = lane & 3u, rec_a = (idx & ~3u) + (ql >> 1), rec_b = rec_a + 2u;
  const unsigned off_st_a = rec_a < total ? rec_a * 32u + (ql & 1u) * 16u : kOutOfRange;
  const unsigned off_st_b = rec_b < total ? rec_b * 32u + (ql & 1u) * 16u : kOutOfRange;
  const bool odd_lane = (ql & 1u) != 0u;
#endif

  // ---- P-matrix staging, one chunk ahead ------------------------------------
  // wave w stages the (operation, child) pairs w, w+4, ... of the chunk: the
  // matrix index is wave-uniform (scalar load), the matrix one coalesced read
  constexpr int kPairsPerWave = (2 * kChunk) / 4;
  constexpr int kMatRegs = (R * 16 + 63) / 64;
  const unsigned wave = uni(tid >> 6);
  auto stage_load = [&](unsigned first_op, double (&st)[kPairsPerWave][kMatRegs]) {
#pragma unroll
    for (int q = 0; q < kPairsPerWave; ++q) {
      const unsigned pair = wave + 4 * q, oi = first_op + pair / 2;
      const unsigned oc = oi < nops ? oi : nops - 1;
      const unsigned mat = (pair & 1u) ? ops[oc].child2_mat : ops[oc].child1_mat;
#pragma unroll
      for (int k = 0; k < kMatRegs; ++k) {
        const unsigned e = lane + 64 * k;
        st[q][k] = e < R * 16 ? v.pmat[(size_t)mat * (R * 16) + e] : 0.0;
      }
    }
  };
  auto stage_write = [&](unsigned buf, const double (&st)[kPairsPerWave][kMatRegs]) {
#pragma unroll
    for (int q = 0; q < kPairsPerWave; ++q) {
      const unsigned pair = wave + 4 * q;
#pragma unroll
      for (int k = 0; k < kMatRegs; ++k) {
        const unsigned e = lane + 64 * k;
        if (e < R * 16) smat[buf][pair / 2][pair & 1u][(e / 16) * kRateStride + (e % 16)] = st[q][k];
      }
    }
  };

  // ---- tip codes through the scalar cache -------------------------------------
  // a wave covers 64/R consecutive sites starting at a multiple of 64/R >= 8 and
  // tip rows are dword-aligned (tip_stride), so its codes are NW aligned dwords;
  // which dword and which byte a lane wants never changes
  constexpr int NW = 64 / R / 4;
  const unsigned wave_first = uni((blockIdx.x * 256 + (tid & ~63u)) / R);   // first site of this wave
  const unsigned wave_site0 = wave_first < S ? wave_first : 0u;
  const unsigned lane_word = (lane / R) >> 2, lane_shift = ((lane / R) & 3u) * 8u;
  const unsigned long long tip_base = reinterpret_cast<unsigned long long>(v.tipcodes) + wave_site0;
  auto load_codes = [&](uint64_t row_off, bool on, TipCodes<NW> &t) {
    const unsigned long long a = tip_base + (on ? row_off : 0ull);
    const unsigned lo = uni((unsigned)a), hi = uni((unsigned)(a >> 32));
    const_u32_ptr p = (const_u32_ptr)(((unsigned long long)hi << 32) | lo);
#pragma unroll
    for (int k = 0; k < NW; ++k) t.w[k] = p[k];
  };
  auto lane_code = [&](const TipCodes<NW> &t) {
    unsigned word = 0;   // (masks, not selects: a select chain over w[] is turned into a scratch array)
#pragma unroll
    for (int k = 0; k < NW; ++k) word |= t.w[k] & (lane_word == (unsigned)k ? ~0u : 0u);
    return (word >> lane_shift) & 0xffu;
  };

  unsigned *park_sc = reinterpret_cast<unsigned *>(park_lds + (size_t)slots * 512);
  // operand of THIS operation: registers / prefetched memory / tip code / LDS slot
  const char *clv_bytes_base = reinterpret_cast<const char *>(v.clv);
  const char *sc_bytes_base = reinterpret_cast<const char *>(v.scaler);
  auto operand = [&](unsigned src, uint64_t off, uint64_t sc_off, const TipCodes<NW> &t,
                     const double (&o)[4], unsigned osc, double (&x)[4], unsigned &xsc) {
    xsc = 0;
    if (src == kSrcReg) {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = o[k];
      xsc = osc;
    } else if (src == kSrcMem) {
      // a sibling that found no parking slot (or was left by an earlier call):
      // read back here, after every earlier store of this lane in program order
      const double2 *p = reinterpret_cast<const double2 *>(clv_bytes_base + off);
      const double2 a = p[(size_t)cidx * 2], b = p[(size_t)cidx * 2 + 1];
      x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
      xsc = sc_off != kNoOffset ? reinterpret_cast<const unsigned *>(sc_bytes_base + sc_off)[s] : 0u;
      // wait HERE (vmcnt(0)): left to the compiler the wait sinks to the join
      // below and every other route would pay for it
      __builtin_amdgcn_s_waitcnt(0x0F70);
    } else if (src == kSrcTip) {
      // bit k of the code -> 1.0 / 0.0: the sign-extended bit masks the high
      // word of 1.0 (two integer instructions per entry)
      const unsigned code = lane_code(t);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int bit = (int)(code << (31 - k)) >> 31;
        x[k] = __hiloint2double(bit & 0x3FF00000, 0);
      }
    } else {
      const unsigned slot = src - kSrcPark;
      const double2 a = park_lds[(slot * 2) * 256 + tid], b = park_lds[(slot * 2 + 1) * 256 + tid];
      x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
      xsc = park_sc[slot * 256 + tid];
    }
  };
  // lanes of the wave whose SITE (R adjacent lanes) is small in every rate:
  // folded on the scalar unit from the ballot, no cross-lane traffic
  constexpr unsigned long long kGroupMask = R == 1 ? ~0ull : R == 2 ? 0x5555555555555555ull
                                          : R == 4 ? 0x1111111111111111ull : 0x0101010101010101ull;
  auto site_small = [&](bool lane_small) {
    unsigned long long m = __builtin_amdgcn_ballot_w64(lane_small);
#pragma unroll
    for (int off = 1; off < R; off <<= 1) m &= m >> off;
    m &= kGroupMask;
#pragma unroll
    for (int off = 1; off < R; off <<= 1) m |= m << off;
    return ((m >> lane) & 1ull) != 0;
  };

  double o[4] = {0, 0, 0, 0};       // the CLV this lane produced last
  unsigned osc = 0;
  TipCodes<NW> t1, t2;
  {
    double st[kPairsPerWave][kMatRegs];
    stage_load(0, st);
    const LevelOp op0 = ops[0];
    load_codes(op0.child1_off, op0.src1 == kSrcTip, t1);
    load_codes(op0.child2_off, op0.src2 == kSrcTip, t2);
    stage_write(0, st);
    __syncthreads();
  }
  const unsigned sc_bytes = S * 4u;
  for (unsigned base = 0; base < nops; base += kChunk) {
    const unsigned buf = (base / kChunk) & 1u;
    double st[kPairsPerWave][kMatRegs];
    stage_load(base + kChunk, st);                    // the NEXT chunk's matrices
    // (the host ends the list with a terminator: chunk_ops[kChunk] always exists)
    const LevelOp *chunk_ops = ops + base;
    // nops is a multiple of kChunk (the host pads with no-ops whose stores go
    // through 0-byte descriptors): the chunk is straight-line code with a fixed
    // number of stores, so the wait for the staged matrices below is counted
    // past them instead of draining the store queue
#pragma unroll
    for (unsigned j = 0; j < kChunk; ++j) {
      const LevelOp op = chunk_ops[j];
      const LevelOp nx = chunk_ops[j + 1];
      double x[4], y[4];
      unsigned xsc, ysc;
      operand(op.src1, op.child1_off, op.child1_sc_off, t1, o, osc, x, xsc);
      operand(op.src2, op.child2_off, op.child2_sc_off, t2, o, osc, y, ysc);
      // what the NEXT operation needs
      load_codes(nx.child1_off, nx.src1 == kSrcTip, t1);
      load_codes(nx.child2_off, nx.src2 == kSrcTip, t2);

      double p1[4], p2[4];
      {
        const double *m = &smat[buf][j][0][r * kRateStride];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          p1[k] = m[k * 4 + 0] * x[0] + m[k * 4 + 1] * x[1] + m[k * 4 + 2] * x[2] + m[k * 4 + 3] * x[3];
        m = &smat[buf][j][1][r * kRateStride];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          p2[k] = m[k * 4 + 0] * y[0] + m[k * 4 + 1] * y[1] + m[k * 4 + 2] * y[2] + m[k * 4 + 3] * y[3];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = p1[k] * p2[k];
      osc = 0;
      const bool has_sc = op.parent_sc_off != kNoOffset && !op.noop;
      if (op.parent_sc_off != kNoOffset) {
        // entries are non-negative, so o < 2^-256 is a comparison of high words
        const unsigned hmax = max(max((unsigned)__double2hiint(o[0]), (unsigned)__double2hiint(o[1])),
                                  max((unsigned)__double2hiint(o[2]), (unsigned)__double2hiint(o[3])));
        osc = xsc + ysc;
        if (site_small(hmax < 0x2FF00000u)) {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] *= kScaleFactor;
          osc += 1;
        }
      }
      __builtin_amdgcn_raw_buffer_store_b32(
          osc, make_rsrc(sc_bytes_base + (has_sc ? op.parent_sc_off : 0ull), has_sc ? sc_bytes : 0u),
          off_sc_st, 0, 0);
      const __amdgpu_buffer_rsrc_t prs =
          make_rsrc(clv_bytes_base + op.parent_off, op.noop ? 0u : clv_bytes);
#if !(defined(RDAMD_ABLATION) && defined(RDAMD_ABL_QUAD_STORES))
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, make_double2(o[0], o[1])), prs, off_clv_st, 0, kClvStoreAux);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, make_double2(o[2], o[3])), prs, off_clv_st + 16, 0, kClvStoreAux);
#else
      {
        const v4u lo = __builtin_bit_cast(v4u, make_double2(o[0], o[1]));
        const v4u hi = __builtin_bit_cast(v4u, make_double2(o[2], o[3]));
        v4u wa, wb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // quad_perm [0,0,1,1] = 0x50: lanes 0, 1 read lane 0 and lanes 2, 3 read lane 1; [2,2,3,3] = 0xFA
          const unsigned la = (unsigned)__builtin_amdgcn_mov_dpp((int)lo[k], 0x50, 0xF, 0xF, true);
          const unsigned ha = (unsigned)__builtin_amdgcn_mov_dpp((int)hi[k], 0x50, 0xF, 0xF, true);
          const unsigned lb = (unsigned)__builtin_amdgcn_mov_dpp((int)lo[k], 0xFA, 0xF, 0xF, true);
          const unsigned hb = (unsigned)__builtin_amdgcn_mov_dpp((int)hi[k], 0xFA, 0xF, 0xF, true);
          wa[k] = odd_lane ? ha : la;
          wb[k] = odd_lane ? hb : lb;
        }
        __builtin_amdgcn_raw_buffer_store_b128(wa, prs, off_st_a, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(wb, prs, off_st_b, 0, 0);
      }
#endif
      if (op.park) {
        const unsigned slot = op.park - 1;
        park_lds[(slot * 2) * 256 + tid] = make_double2(o[0], o[1]);
        park_lds[(slot * 2 + 1) * 256 + tid] = make_double2(o[2], o[3]);
        park_sc[slot * 256 + tid] = osc;
      }
    }
    stage_write(buf ^ 1u, st);
    __syncthreads();
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
    const unsigned code1 = tip1 ? v.tipcodes[(size_t)op.child1_clv * v.tip_stride + s] : 0;
    const unsigned code2 = tip2 ? v.tipcodes[(size_t)op.child2_clv * v.tip_stride + s] : 0;
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
  const size_t lds_static = 10u * 1024, per_slot = 256 * (32 + 4);
  const size_t budget = (size_t)(160 * 1024) / std::max<size_t>(per_cu, 1);
  if (budget <= lds_static) return 0;
  return (unsigned)std::min<size_t>((budget - lds_static) / per_slot, kMaxParkSlots);
}

unsigned clv_traversal_chunk(const rdamd_partition *p) { return dna_fast_ok(p) ? kChunk : 1u; }

// A list whose one row of blocks leaves most of the device's wave slots empty is cut into pieces
// (k20_split.hpp, list_levels).  How many per launch, measured on one box (profiles/
// r5_clv_pieces_ab.txt; us per traversal, whole list -> best cut):
//   c2 / 8 (98 blocks, 99 operations)      102 ->   65 with 16 (32: 69)
//   125.phy (304 blocks, 124)              133 ->   91 with 16 (8: 110, 32: 100)
//   c2 (782 blocks, 99)                    175 ->  158 with  8 (16: 188 -- a third of the list is the
//                                                  tree's spine then, a chain no cut shortens)
//   c5's shard (782 blocks, 999)          1848 -> 1610 with 16 - 32 (8: 1800)
//   c4's shard (977 blocks, 499)          1013 ~  1000 with  8; c5 (1 563 blocks), c4 (7 813): no gain
// i.e. about 6 000 blocks per launch, at least a piece per 64 operations, no piece under 6.
unsigned clv_traversal_pieces(const rdamd_partition *p, unsigned count) {
  if (!dna_fast_ok(p) || p->sites == 0 || count < 24) return 0;
  const size_t blocks = ((size_t)p->sites * p->rate_cats + 255) / 256;
  size_t want = blocks <= 896 ? std::min<size_t>(std::max<size_t>(6144 / blocks, count / 64), count / 6) : 0;
#ifdef RDAMD_ABLATION
  if (getenv("RDAMD_CLV_PIECES")) want = (size_t)atoi(getenv("RDAMD_CLV_PIECES"));
#endif
  return want >= 2 ? (unsigned)std::min<size_t>(want, kMaxListPieces) : 0u;
}

hipError_t launch_clv_traversal(rdamd_partition *p, const LevelOp *d_ops, const ListPieces &pieces,
                                unsigned slots) {
  if (pieces.n == 0 || p->sites == 0) return hipSuccess;
  DeviceView v = p->view();
  const unsigned R = p->rate_cats;
  if (dna_fast_ok(p)) {
    // one (site, rate) pair per lane where possible: maximum memory-level
    // parallelism; the lane -> pair map is identical for every operation
    size_t total = (size_t)p->sites * R;
    const dim3 grid((unsigned)((total + 255) / 256), pieces.n);
    const size_t lds = (size_t)slots * 256 * (32 + 4);
    switch (R) {
      case 1: clv_dna_traversal_kernel<1><<<grid, 256, lds, p->stream>>>(v, d_ops, pieces, slots); break;
      case 2: clv_dna_traversal_kernel<2><<<grid, 256, lds, p->stream>>>(v, d_ops, pieces, slots); break;
      case 4: clv_dna_traversal_kernel<4><<<grid, 256, lds, p->stream>>>(v, d_ops, pieces, slots); break;
      default: clv_dna_traversal_kernel<8><<<grid, 256, lds, p->stream>>>(v, d_ops, pieces, slots); break;
    }
  } else {
    unsigned gx = (p->sites + 255) / 256;
    for (unsigned k = 0; k < pieces.n; ++k)   // (this path never cuts its lists: one piece)
      clv_generic_traversal_kernel<<<gx, 256, 0, p->stream>>>(v, d_ops + pieces.start[k], pieces.len[k]);
  }
  return hipGetLastError();
}

}  // namespace rdamd
