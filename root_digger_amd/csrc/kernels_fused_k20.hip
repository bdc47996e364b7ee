// Fused full-traversal evaluator, 20-state data.
//
// The 20-state sibling of kernels_fused.hip: one launch evaluates a batch of
// (schedule, parameter set) jobs -- the body of model_t::compute_lh_partition
// (/root/reference/src/model.cpp:454-476: corax_update_clvs followed by
// corax_compute_root_loglikelihood) -- without materialising any CLV in HBM,
// so it has no store stream at all (the materialising 20-state kernel,
// kernels_clv_mfma.hip, is held back by exactly that: DESIGN.md 4.2).
//
// Arithmetic: the same v_mfma_f64_4x4x4_4b_f64 tiling and lane layout as
// kernels_clv_mfma.hip -- one wave = (16 sites, one rate), a lane holds states
// {4 s + grp} of site `col` both as MFMA B operand and as MFMA result, a
// matrix-vector product is 25 MFMAs in five independent chains.  A operands
// come from the job's MFMA-ready P copies as contiguous 16-byte pieces,
// requested one step ahead and redistributed through wave-private LDS.  A tip
// child costs no MFMA: its term is a row of the job's tip table (written next
// to the P copies: per code the sums of P over the code's states, laid out so
// that a lane's five entries are 48 contiguous bytes), looked up by the tip
// code that was fetched two steps ahead.  So a step runs at most ONE product:
// the running CLV times its branch matrix.
//
// Control: a wave walks the job's compiled program (evaluate.hip) with the
// running CLV in registers and pending siblings on a wave-private LDS stack
// whose depth the host minimised (level 0 of it in registers).  Nothing is shared between waves until the
// root, so there is no barrier in the loop: each (site, rate) keeps its own
// 2^256 rescale count (decided by a ballot folded on the scalar unit) and the
// root sum aligns the rate terms to the smallest count -- the per-rate-scaler
// form of the reference rule, as in kernels_fused.hip.
#include "common.hpp"
#include "fused.hpp"

namespace rdamd {

namespace {

constexpr int kK = 20;
constexpr int kSteps = 5;            // k steps of 4 = states held per lane
constexpr int kGroups = 5;           // row groups of 4
constexpr int kBlocks = kGroups * kSteps;
constexpr int kCopy = kBlocks * 16;  // doubles per (matrix, rate) copy
constexpr int kCopyLds = 4 * 64 * 16;   // bytes one copy occupies in LDS (4 pieces per lane)
constexpr int kLevelBytes = kSteps * 64 * 8 + 64 * 4;   // one stack level: 5 doubles + 1 count per lane

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) unsigned *const_u32_ptr;

__device__ __forceinline__ unsigned uni(unsigned x) {
  return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = uni((unsigned)u), hi = uni((unsigned)(u >> 32));
  void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)uni(bytes), 0x00020000);
}
__device__ __forceinline__ const_u32_ptr scalar_ptr(const void *p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  return (const_u32_ptr)(((unsigned long long)uni((unsigned)(u >> 32)) << 32) | uni((unsigned)u));
}
__device__ __forceinline__ double pow2_neg256(int d) {   // exact 2^(-256 d), 0 once it underflows
  return d == 0 ? 1.0 : (d == 1 ? kScaleThreshold
       : (d == 2 ? kScaleThreshold * kScaleThreshold
       : (d == 3 ? kScaleThreshold * kScaleThreshold * kScaleThreshold : 0.0)));
}

struct Step {   // a FusedOp: pM byte offset of the running child's A copy, cX / cY tip rows,
  unsigned pM, tX, tY, cX, cY, flags, tabX, tabY;   // tabX / tabY the tip children's table offsets
};
__device__ __forceinline__ Step load_step(const_u32_ptr prog, unsigned i) {
  const const_u32_ptr w = prog + (size_t)i * (sizeof(FusedOp) / 4);
  return Step{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
}

}  // namespace

// ---- P-matrices of a batch, straight into the MFMA-ready layout ---------------
// exp(Q t) by scaling and squaring around the degree-16 Taylor polynomial of pmatrix_generic_kernel
// (||Q t / 2^s||_1 <= 1/4).  All (matrix, rate) problems of a job share Q, so its powers
// Q^1 .. Q^16 are made ONCE per job (fused20_qpowers_kernel, 15 products) and a problem's
// polynomial is
//     p = I + sum_k (tau^k / k!) Q^k,   tau = t / 2^s
// -- 16 multiply-adds per element, the powers read from L2 (51 KB per job, read by its ~1 600
// problems), no matrix product -- followed by the s squarings, which are the LDS-tiled 20x20
// products of before: a thread owns a 4x4 tile of the product, per inner step four 8-byte reads
// of A (a row each, shared by the five threads of a tile row) and two 16-byte reads of B feed 16
// FMAs.  25 threads of a 32-lane half wave work on a (job, matrix, rate) problem, a 128-thread
// workgroup holds four.  (Until round 4 every problem evaluated the polynomial itself,
// Paterson-Stockmeyer style: 6 products + the squarings, 2.64 ms per c3 batch.)
namespace {

constexpr int kLd = 20;             // row stride in LDS (doubles).  Unpadded: the tile rows of A collide on banks (stride 22 avoids
                                    // that), but less LDS per workgroup is another workgroup per CU
constexpr int kMatLds = kK * kLd;   // doubles per matrix in LDS
constexpr unsigned kQPow20Stride = 16 * kK * kK + 8;   // doubles per job: Q^1 .. Q^16, ||Q||_1 at [6400]

// c = A . B for this thread's 4x4 tile (rows 4 ti .., columns 4 tj ..)
__device__ __forceinline__ void tile_product(const double *A, const double *B, unsigned ti, unsigned tj,
                                             double (&c)[16]) {
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.0;
  const double *a = A + (4 * ti) * kLd, *b = B + 4 * tj;
#pragma unroll 4
  for (int l = 0; l < kK; ++l) {
    const double2 b0 = *reinterpret_cast<const double2 *>(b + l * kLd);
    const double2 b1 = *reinterpret_cast<const double2 *>(b + l * kLd + 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double x = a[r * kLd + l];
      c[4 * r + 0] = fma(x, b0.x, c[4 * r + 0]); c[4 * r + 1] = fma(x, b0.y, c[4 * r + 1]);
      c[4 * r + 2] = fma(x, b1.x, c[4 * r + 2]); c[4 * r + 3] = fma(x, b1.y, c[4 * r + 3]);
    }
  }
}
__device__ __forceinline__ void tile_store(double *C, unsigned ti, unsigned tj, const double (&c)[16]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double *row = C + (4 * ti + r) * kLd + 4 * tj;
    *reinterpret_cast<double2 *>(row) = make_double2(c[4 * r], c[4 * r + 1]);
    *reinterpret_cast<double2 *>(row + 2) = make_double2(c[4 * r + 2], c[4 * r + 3]);
  }
}

}  // namespace

// The powers of every job's Q: 32 lanes (25 workers) per job, four jobs per workgroup;
// Q^(k+1) = Q^k . Q through two LDS buffers, each power also written to qpow[job][k][i][j]
// (row-major, unpadded); ||Q||_1 behind them.
__global__ void __launch_bounds__(128)
fused20_qpowers_kernel(const double *__restrict__ q, unsigned n_jobs, double *__restrict__ qpow) {
  __shared__ __attribute__((aligned(16))) double lds[4][3][kMatLds];
  const unsigned sub = threadIdx.x >> 5, w = threadIdx.x & 31u;
  const unsigned job_raw = blockIdx.x * 4 + sub;
  const bool live = job_raw < n_jobs;
  const unsigned job = live ? job_raw : n_jobs - 1;
  const bool worker = w < 25;
  const unsigned ti = worker ? w / 5 : 0, tj = worker ? w % 5 : 0;
  double *Q = lds[sub][0], *A = lds[sub][1], *B = lds[sub][2];
  const double *qq = q + (size_t)job * (kK * kK);
  double *out = qpow + (size_t)job * kQPow20Stride;
  for (unsigned e = w; e < (unsigned)(kK * kK); e += 32) {
    const double v = qq[e];
    Q[(e / kK) * kLd + e % kK] = v;
    A[(e / kK) * kLd + e % kK] = v;
    if (live) out[e] = v;
  }
  __syncthreads();
  if (w == 0 && live) {   // ||Q||_1: the largest column sum of absolute values, columns in order
    double norm = 0.0;
    for (unsigned j = 0; j < (unsigned)kK; ++j) {
      double cs = 0.0;
      for (unsigned i = 0; i < (unsigned)kK; ++i) cs += fabs(Q[i * kLd + j]);
      norm = fmax(norm, cs);
    }
    out[16 * kK * kK] = norm;
  }
  for (int k = 1; k < 16; ++k) {   // A = Q^k  ->  B = Q^(k+1)
    double c[16];
    if (worker) {
      tile_product(A, Q, ti, tj, c);
      tile_store(B, ti, tj, c);
      if (live) {
        double *o = out + (size_t)k * (kK * kK);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double *row = o + (4 * ti + r) * kK + 4 * tj;
          *reinterpret_cast<double2 *>(row) = make_double2(c[4 * r], c[4 * r + 1]);
          *reinterpret_cast<double2 *>(row + 2) = make_double2(c[4 * r + 2], c[4 * r + 3]);
        }
      }
    }
    __syncthreads();
    double *swap = A; A = B; B = swap;
  }
}

__global__ void __launch_bounds__(128)
fused20_pmatrix_kernel(const double *__restrict__ qpow, const double *__restrict__ rates,
                       const FusedJob *__restrict__ jobs, unsigned n_mat, unsigned R, unsigned total,
                       double *__restrict__ pmat, size_t pmat_job_stride,
                       double *__restrict__ tiptab, size_t tiptab_job_stride,
                       const uint64_t *__restrict__ codemask, unsigned ncodes) {
  __shared__ __attribute__((aligned(16))) double lds[4][2][kMatLds];
  __shared__ __attribute__((aligned(16))) double stage_lds[2][kK * kK];   // the power in use / the next one
  __shared__ int sq[4];
  const unsigned sub = threadIdx.x >> 5, w = threadIdx.x & 31u;
  const unsigned prob = blockIdx.x * 4 + sub;
  const bool live = prob < total;
  const unsigned pr = live ? prob : total - 1;
  const unsigned per_job = n_mat * R;
  const unsigned job = pr / per_job, rem = pr % per_job;
  const unsigned m = rem / R, r = rem % R;
  const bool worker = w < 25;
  const unsigned ti = worker ? w / 5 : 0, tj = worker ? w % 5 : 0;

  const double t = jobs[job].brlen[m] * rates[(size_t)job * R + r];
  const double *__restrict__ qp = qpow + (size_t)job * kQPow20Stride;
  // scaling: the smallest s with ||Q||_1 t / 2^s <= 1/4 (every lane works it out for itself)
  int s_mine = 0;
  double scale = 1.0;
  {
    const double norm = qp[16 * kK * kK] * t;
    while (norm * scale > 0.25 && s_mine < 60) { scale *= 0.5; ++s_mine; }
  }
  if (w == 0) sq[sub] = s_mine;
  const double tau = t * scale;
  // 1/k!, k = 0..16
  constexpr double f[17] = {1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040,
                            1.0 / 40320, 1.0 / 362880, 1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600,
                            1.0 / 6227020800.0, 1.0 / 87178291200.0, 1.0 / 1307674368000.0,
                            1.0 / 20922789888000.0};
  double *H = lds[sub][0], *T = lds[sub][1];
  double c[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = (4 * ti + (e >> 2) == 4 * tj + (e & 3)) ? 1.0 : 0.0;
  // The polynomial, power by power.  The four problems of a workgroup belong to ONE job whenever
  // n_mat R is a multiple of 4 (always with 4 rate categories): the workgroup then fetches each
  // power once into LDS (double-buffered: power k + 1 is on its way while power k is used) and the
  // four problems read it there -- a quarter of the L2 traffic, which is what bounds this loop
  // (51 KB of powers per problem).  A workgroup that straddles two jobs reads them from L2 directly.
  const unsigned first = blockIdx.x * 4, last = min(first + 3u, total - 1u);
  const bool one_job = first / per_job == last / per_job;   // (workgroup-uniform)
  double pw = 1.0;
  if (one_job) {
    double *stage = &stage_lds[0][0];
    constexpr unsigned kPer = (kK * kK + 127) / 128;   // elements of a power per thread (4, the last one partly)
    double nxt[kPer];
#pragma unroll
    for (unsigned u = 0; u < kPer; ++u) {
      const unsigned e = threadIdx.x + 128 * u;
      nxt[u] = e < (unsigned)(kK * kK) ? qp[e] : 0.0;
    }
    for (int k = 1; k <= 16; ++k) {
      double *buf = stage + (k & 1) * (kK * kK);
#pragma unroll
      for (unsigned u = 0; u < kPer; ++u) {
        const unsigned e = threadIdx.x + 128 * u;
        if (e < (unsigned)(kK * kK)) buf[e] = nxt[u];
      }
      __syncthreads();   // (also: everybody is done with the buffer that power k + 1 will go into)
      if (k < 16) {
#pragma unroll
        for (unsigned u = 0; u < kPer; ++u) {
          const unsigned e = threadIdx.x + 128 * u;
          nxt[u] = e < (unsigned)(kK * kK) ? qp[(size_t)k * (kK * kK) + e] : 0.0;
        }
      }
      pw *= tau;
      const double ck = f[k] * pw;
      const double *qk = buf + (4 * ti) * kK + 4 * tj;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const double2 lo = *reinterpret_cast<const double2 *>(qk + rr * kK);
        const double2 hi = *reinterpret_cast<const double2 *>(qk + rr * kK + 2);
        c[4 * rr + 0] = fma(ck, lo.x, c[4 * rr + 0]); c[4 * rr + 1] = fma(ck, lo.y, c[4 * rr + 1]);
        c[4 * rr + 2] = fma(ck, hi.x, c[4 * rr + 2]); c[4 * rr + 3] = fma(ck, hi.y, c[4 * rr + 3]);
      }
    }
  } else if (worker) {
    for (int k = 1; k <= 16; ++k) {
      pw *= tau;
      const double ck = f[k] * pw;
      const double *__restrict__ qk = qp + (size_t)(k - 1) * (kK * kK) + (4 * ti) * kK + 4 * tj;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const double2 lo = *reinterpret_cast<const double2 *>(qk + rr * kK);
        const double2 hi = *reinterpret_cast<const double2 *>(qk + rr * kK + 2);
        c[4 * rr + 0] = fma(ck, lo.x, c[4 * rr + 0]); c[4 * rr + 1] = fma(ck, lo.y, c[4 * rr + 1]);
        c[4 * rr + 2] = fma(ck, hi.x, c[4 * rr + 2]); c[4 * rr + 3] = fma(ck, hi.y, c[4 * rr + 3]);
      }
    }
  }
  if (worker) tile_store(H, ti, tj, c);
  __syncthreads();
  const int s_max = max(max(sq[0], sq[1]), max(sq[2], sq[3]));
  for (int k = 0; k < s_max; ++k) {   // squarings
    if (worker && k < s_mine) { tile_product(H, H, ti, tj, c); tile_store(T, ti, tj, c); }
    __syncthreads();
    if (k < s_mine) { double *swap = H; H = T; T = swap; }
  }
  if (!live) return;
  const double *out = H;
  // element (rg, ks, k, i) of the copy = P[4 rg + i][4 ks + k]  (kernels_clv_mfma.hip)
  double *o = pmat + (size_t)job * pmat_job_stride + ((size_t)m * R + r) * kCopy;
  for (unsigned e = w; e < (unsigned)kCopy; e += 32) {
    const unsigned i = e & 3, k = (e >> 2) & 3, blk = e >> 4, ks = blk % kSteps, rg = blk / kSteps;
    const double v = out[(4 * rg + i) * kLd + 4 * ks + k];
    o[e] = v <= 0.0 ? 0.0 : v;   // (<=: no -0.0 either -- the rescale test reads high words)
  }
  // tip table, one 192-byte row per code: positions 0-7 = (s, g) for s = 0, 1 as pairs
  // [g][s], 8-15 the same for s = 2, 3, 16-19 = s = 4 by g, 20-23 padding; entry (s, g) =
  // sum over the states j of the code of P[4 s + g][j].  The four lanes of a site then
  // read 64 + 64 + 32 CONTIGUOUS bytes with three instructions (one cache line per site
  // and instruction instead of two).
  // (only where the evaluator will look: a branch that ends in a tip -- half of them, and the table
  // is four fifths of what this kernel writes)
  const unsigned *tipmask = reinterpret_cast<const unsigned *>(jobs[job].clade_steps);
  if (tipmask && !((tipmask[m >> 5] >> (m & 31u)) & 1u)) return;
  double *tt = tiptab + (size_t)job * tiptab_job_stride + ((size_t)m * R + r) * kFused20TabDoubles;
  for (unsigned e = w; e < ncodes * kFused20TabRow; e += 32) {
    const unsigned cc = e / kFused20TabRow, ww = e % kFused20TabRow;
    const unsigned g = ww < 16 ? (ww & 7u) >> 1 : ww - 16, sidx = ww < 16 ? 2 * (ww >> 3) + (ww & 1u) : 4;
    double acc = 0.0;
    if (ww < 20) {
      // (the states of the code in ascending order, as a loop over all twenty would add them --
      // adding the zeros of the others changes nothing --: an unambiguous residue is ONE read
      // instead of twenty, and most codes of real alignments are)
      uint64_t mask = codemask[cc] & ((1ull << kK) - 1);
      const double *row = out + (4 * sidx + g) * kLd;
      while (mask) {
        const unsigned j = (unsigned)__builtin_ctzll(mask);
        mask &= mask - 1;
        const double v = row[j] <= 0.0 ? 0.0 : row[j];
        acc += v;
      }
    }
    tt[e] = acc;
  }
}

// ---- the evaluator ---------------------------------------------------------------
// grid = (groups of NT 16-site tiles, jobs); workgroup = R waves, wave r = rate
// category r.  A wave carries NT tiles: the A copy of a step is fetched, staged
// and read from LDS once and multiplies NT running CLVs (2 NT x 5 independent MFMA
// chains), so the per-step traffic through the CU's address unit, the LDS reads
// and the scalar control flow are shared by NT x 16 sites.
// Dynamic LDS: [root exchange R x NT x 16 x (8 + 4) B][per wave: one A copy (4 KB) +
// `depth` stack levels of NT tiles].
// EXPORT (rdamd_evaluate_root_children): the steps the host flagged (0x8000 / 0x10000: they compute
// the root operation's two children) also store the running CLV -- in the partition's operand
// layout -- and its rescale count: what the root-only steps of the search read afterwards
// (src/model.cpp:415-446), instead of a traversal that materialises every CLV.
// THREADS: 256 for up to four rate categories (the shape every measurement of DESIGN 4.6 is about),
// 512 for five to eight -- a workgroup is R waves (round 6: `rd --rate-cats N` takes any N,
// src/main.cpp:256-266; beyond eight the traversal kernels serve).
template <int NT, bool EXPORT, int THREADS>
__global__ void __launch_bounds__(THREADS)
fused20_eval_kernel(Fused20Args a, unsigned depth) {
  extern __shared__ char lds_raw[];
  const unsigned R = a.rate_cats, S = a.sites;
  const unsigned lane = threadIdx.x & 63, r = uni(threadIdx.x >> 6);
  const unsigned col = lane & 15, grp = lane >> 4;
  const unsigned job = blockIdx.y;
  unsigned site[NT], ls[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    site[q] = (blockIdx.x * NT + q) * 16 + col;
    ls[q] = site[q] < S ? site[q] : S - 1;   // clamped for loads
  }

  double *root_f = reinterpret_cast<double *>(lds_raw);
  int *root_sc = reinterpret_cast<int *>(lds_raw + R * NT * 16 * 8);
  const unsigned wave_bytes = kCopyLds + depth * NT * kLevelBytes;
  char *a_lds = lds_raw + R * NT * 16 * 12 + r * wave_bytes;   // [4 pieces][64 lanes][16 B]
  char *stack = a_lds + kCopyLds;                               // [level][tile][5][64] doubles + [64] counts

  const FusedJob jb = a.jobs[job];
  const const_u32_ptr prog = scalar_ptr(jb.prog);   // n_ops + 4 entries (tail padded)
  const unsigned nops = jb.n_ops;
  const char *pm_job = reinterpret_cast<const char *>(a.pmat + (size_t)job * a.pmat_job_stride);
  const char *tt_job = reinterpret_cast<const char *>(a.tiptab + (size_t)job * a.tiptab_job_stride);
  const unsigned a_off = (grp * 4 + (col & 3)) * 8u;   // my element of every 4x4 block
  const unsigned tt_rate = r * (kFused20TabDoubles * 8u);
  const unsigned tt_lane = grp * 16u;                  // + code * 192: my share of a table row (see the P-matrix kernel)

  // What a step needs from memory, by its kind (wave-uniform branches: the CU's one
  // address unit serves every wave of the CU, so no vector-memory instruction is
  // issued in vain -- an empty descriptor still costs it a slot), requested so
  // that no loaded register has to be copied before it is used:
  //   * the A copy of step i+1 (unless it is a tip-tip step) at the top of step i
  //     (its registers were emptied into LDS at the end of step i-1), staged into
  //     LDS at the end of step i;
  //   * the table rows of step i+1's tip children right after step i has used its
  //     own rows, i.e. behind its MFMAs;
  //   * the 16 tip codes of a tile are 16 contiguous bytes of a tip row: they come
  //     through the scalar cache, two steps ahead.
  struct Rows { u32x4 a, b; u32x2 c; };   // table entries 0-1, 2-3, 4 of one tip child of one tile
  u32x4 raw[4];
  Rows t1[NT], t2[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    t1[q].a = t1[q].b = t2[q].a = t2[q].b = u32x4{0u, 0u, 0u, 0u};
    t1[q].c = t2[q].c = u32x2{0u, 0u};
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) raw[k] = u32x4{0u, 0u, 0u, 0u};
  struct Codes { unsigned w0, w1, w2, w3; };
  struct TileCodes { Codes t[NT]; };
  const unsigned long long codes_u = reinterpret_cast<unsigned long long>(a.tipcodes);
  const unsigned long long codes_lo =
      (((unsigned long long)uni((unsigned)(codes_u >> 32)) << 32) | uni((unsigned)codes_u)) +
      (unsigned long long)blockIdx.x * (16u * NT);
  const bool upper_half = (col & 8u) != 0u;
  const unsigned code_shift = (col & 7u) * 8u;
  auto load_codes = [&](bool wanted, unsigned row_off) -> TileCodes {
    TileCodes c;
#pragma unroll
    for (int q = 0; q < NT; ++q) c.t[q] = Codes{0u, 0u, 0u, 0u};
    if (wanted) {
      const const_u32_ptr row = (const_u32_ptr)(codes_lo + row_off);
#pragma unroll
      for (int q = 0; q < NT; ++q) c.t[q] = Codes{row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]};
    }
    return c;
  };
  // In the loop the codes are fetched unconditionally (a step without that tip child
  // carries row offset 0: a valid row, the result is not used) from the offsets of a
  // step head that arrived an iteration ago: no branch, no scalar load that waits for
  // another one -- nothing touches the result before next iteration's table requests.
  auto load_codes_ahead = [&](unsigned row_off) -> TileCodes {
    TileCodes c;
    const const_u32_ptr row = (const_u32_ptr)(codes_lo + row_off);
#pragma unroll
    for (int q = 0; q < NT; ++q) c.t[q] = Codes{row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]};
    return c;
  };
  auto my_code = [&](const Codes &c) -> unsigned {
    const unsigned long long lo = ((unsigned long long)c.w1 << 32) | c.w0;
    const unsigned long long hi = ((unsigned long long)c.w3 << 32) | c.w2;
    return (unsigned)((upper_half ? hi : lo) >> code_shift) & 255u;
  };
  // ONE descriptor for the job's A copies and one for its tables, built once; a request adds
  // its (matrix, rate) as the scalar offset of the load.  (Round 2 built a descriptor per
  // request -- eight scalar instructions each, three requests per step.)  The last 16-byte piece
  // of a 3 200-byte copy runs 896 bytes into the next copy: staged and never read.
  const __amdgpu_buffer_rsrc_t pm_rs = make_rsrc(pm_job, (unsigned)(a.pmat_job_stride * 8));
  const __amdgpu_buffer_rsrc_t tt_rs = make_rsrc(tt_job, (unsigned)(a.tiptab_job_stride * 8));
  const unsigned pm_rate = r * (kCopy * 8u);
  auto request_a = [&](const Step &st) {
    const int so = (int)(st.pM + pm_rate);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      raw[k] = __builtin_amdgcn_raw_buffer_load_b128(pm_rs, (int)(lane * 16u + 1024u * k), so, 0);
  };
  auto request_tab = [&](unsigned tab_off, const TileCodes &codes, Rows (&t)[NT]) {
    const int so = (int)(tab_off + tt_rate);
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int o = (int)(my_code(codes.t[q]) * 192u + tt_lane);
      t[q].a = __builtin_amdgcn_raw_buffer_load_b128(tt_rs, o, so, 0);          // s = 0, 1
      t[q].b = __builtin_amdgcn_raw_buffer_load_b128(tt_rs, o + 64, so, 0);     // s = 2, 3
      t[q].c = __builtin_amdgcn_raw_buffer_load_b64(tt_rs, (int)(my_code(codes.t[q]) * 192u + 128u + grp * 8u), so, 0);   // s = 4
    }
  };
  auto has_tip1 = [](const Step &st) { return (st.flags & 3u) == kFusedTT; };
  auto has_tip2 = [](const Step &st) { return (st.flags & 3u) == kFusedTT || (st.flags & 3u) == kFusedRT; };
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4 *>(a_lds + lane * 16u + 1024u * k) = raw[k];
  };
  // d[q] = A(the copy in LDS) . b[q] : NT x five chains of five MFMAs, each A operand read once
  auto product = [&](const double (&b)[NT][kSteps], double (&d)[NT][kGroups]) {
    const char *ap = a_lds + a_off;
#pragma unroll
    for (int s = 0; s < kSteps; ++s)
#pragma unroll
      for (int t = 0; t < kGroups; ++t) {
        const double aa = *reinterpret_cast<const double *>(ap + 128 * (t * kSteps + s));
#pragma unroll
        for (int q = 0; q < NT; ++q)   // (first k step: the constant 0 as C, no register to clear)
          d[q][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(aa, b[q][s], s == 0 ? 0.0 : d[q][t], 0, 0, 0);
      }
  };
  auto f64 = [](unsigned lo, unsigned hi) { return __builtin_bit_cast(double, u32x2{lo, hi}); };
  auto row_of = [&](const Rows &t, double (&o)[kSteps]) {
    o[0] = f64(t.a[0], t.a[1]); o[1] = f64(t.a[2], t.a[3]);
    o[2] = f64(t.b[0], t.b[1]); o[3] = f64(t.b[2], t.b[3]);
    o[4] = f64(t.c[0], t.c[1]);
  };

  double v[NT][kSteps];   // the running CLVs: states 4 s + grp of site col of each tile
  int sc[NT];             // their 2^256 rescale counts
  double s0[NT][kSteps];  // stack level 0 (the most used) stays in registers
  int s0sc[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    sc[q] = s0sc[q] = 0;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) v[q][s] = s0[q][s] = 0.0;
  }
  unsigned sp = 0;

  // prologue: step 0's operands and the codes of step 1
  Step cur = load_step(prog, 0);
  Step nxt = load_step(prog, 1);
  Step nx2 = load_step(prog, 2);   // (the program is padded: heads up to n_ops + 3 exist)
  {
    const TileCodes k1 = load_codes(has_tip1(cur), cur.cX), k2 = load_codes(has_tip2(cur), cur.cY);
    if (!has_tip1(cur)) request_a(cur);
    if (has_tip1(cur)) request_tab(cur.tabX, k1, t1);
    if (has_tip2(cur)) request_tab(cur.tabY, k2, t2);
    if (!has_tip1(cur)) stage();
  }
  TileCodes cw1 = load_codes(has_tip1(nxt), nxt.cX), cw2 = load_codes(has_tip2(nxt), nxt.cY);

  for (unsigned i = 0; i < nops; ++i) {
    const Step nx3 = load_step(prog, i + 3);   // first looked at an iteration from now
    const unsigned kind = cur.flags & 3u;
    const bool next_product = !has_tip1(nxt) && i + 1 < nops;
    // tip codes of the step after next (scalar loads; the offsets came with nx2 an iteration ago)
    const TileCodes ncw1 = load_codes_ahead(nx2.cX), ncw2 = load_codes_ahead(nx2.cY);
    if (kind == kFusedTT) {   // both children are table rows: use them at once ...
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        double x[kSteps], y[kSteps];
        row_of(t1[q], x);
        row_of(t2[q], y);
#pragma unroll
        for (int s = 0; s < kSteps; ++s) v[q][s] = x[s] * y[s];
        sc[q] = 0;
      }
    }
    // ... so that, unless this step still needs its own row after its product (RT), the
    // next step's rows are requested a whole step ahead (the codes arrived a step ago)
    const bool rows_early = kind != kFusedRT;
    if (rows_early && has_tip1(nxt)) request_tab(nxt.tabX, cw1, t1);
    if (rows_early && has_tip2(nxt)) request_tab(nxt.tabY, cw2, t2);
    if (next_product) request_a(nxt);  // the A copy of the next step
    double d1[NT][kGroups];
    if (kind != kFusedTT) product(v, d1);
    if (kind == kFusedPark) {          // push M . (running CLV); the next step is a TT
      if (cur.flags & 0x200u) {        // stack level 0 lives in registers
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
          for (int s = 0; s < kSteps; ++s) s0[q][s] = d1[q][s];
          s0sc[q] = sc[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          char *level = stack + (sp * NT + q) * kLevelBytes;
          double *lv = reinterpret_cast<double *>(level) + lane;
#pragma unroll
          for (int s = 0; s < kSteps; ++s) lv[s * 64] = d1[q][s];
          reinterpret_cast<int *>(level + kSteps * 64 * 8)[lane] = sc[q];
        }
        ++sp;
      }
    } else {
      if (kind == kFusedRP && !(cur.flags & 0x400u)) --sp;
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        if (kind == kFusedRP && (cur.flags & 0x400u)) {   // the sibling waits in the register slot
#pragma unroll
          for (int s = 0; s < kSteps; ++s) v[q][s] = d1[q][s] * s0[q][s];
          sc[q] += s0sc[q];
        } else if (kind == kFusedRP) {   // ... or on the LDS stack (multiplied by its matrix when parked)
          const char *level = stack + (sp * NT + q) * kLevelBytes;
          const double *lv = reinterpret_cast<const double *>(level) + lane;
#pragma unroll
          for (int s = 0; s < kSteps; ++s) v[q][s] = d1[q][s] * lv[s * 64];
          sc[q] += reinterpret_cast<const int *>(level + kSteps * 64 * 8)[lane];
        } else if (kind == kFusedRT) {   // this step's table rows, used in place
          double y[kSteps];
          row_of(t2[q], y);
#pragma unroll
          for (int s = 0; s < kSteps; ++s) v[q][s] = d1[q][s] * y[s];
        }   // (tip-tip: v was formed at the top of the step)
        // all five entries < 2^-256: entries are non-negative and never -0.0 (P-matrix and
        // table entries are clamped with `<= 0 ? +0`, and sums / products of such values
        // keep the sign bit clear), so the largest high word decides (two v_max3_u32 + one compare instead of five FP64 compares; a NaN
        // compares as large and never rescales, as with `<`)
        unsigned hmax = 0u;
#pragma unroll
        for (int s = 0; s < kSteps; ++s) hmax = max(hmax, (unsigned)__double2hiint(v[q][s]));
        const bool small = hmax < 0x2FF00000u;
        // all 20 entries of a (site, rate) sit in the four lanes col + 16 g
        unsigned long long bm = __ballot(small);
        bm &= bm >> 32;
        bm &= bm >> 16;
        if ((bm >> col) & 1ull) {
#pragma unroll
          for (int s = 0; s < kSteps; ++s) v[q][s] *= kScaleFactor;
          sc[q] += 1;
        }
      }
    }
    // (an RT step: the next step's table rows only now;) then the next step's A copy
    // replaces this one's in LDS
    if (!rows_early && has_tip1(nxt)) request_tab(nxt.tabX, cw1, t1);
    if (!rows_early && has_tip2(nxt)) request_tab(nxt.tabY, cw2, t2);
    if (next_product) stage();
    if (EXPORT && (cur.flags & 0x18000u)) {   // a child of the root operation: leave it behind
      double *ec = (cur.flags & 0x8000u) ? a.export_clv[0] : a.export_clv[1];
      unsigned *en = (cur.flags & 0x8000u) ? a.export_cnt[0] : a.export_cnt[1];
#pragma unroll
      for (int q = 0; q < NT; ++q)
        if (site[q] < S) {
          double *tile = ec + ((size_t)r * a.tiles + (blockIdx.x * NT + q)) * (16 * kK);
#pragma unroll
          for (int s = 0; s < kSteps; ++s) tile[k20_tile_index(col, 4u * s + grp)] = v[q][s];
          if (grp == 0) en[(size_t)site[q] * R + r] = (unsigned)sc[q];
        }
    }
    cur = nxt;
    nxt = nx2;
    nx2 = nx3;
    cw1 = ncw1;
    cw2 = ncw2;
  }

  // root: f_r = sum_k pi_k v[k] for my site and rate, then the rate sum
  const double *freqs = a.freqs + (size_t)job * kK;
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    double f = 0.0;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) f += freqs[4 * s + grp] * v[q][s];
    f += __shfl_xor(f, 16);
    f += __shfl_xor(f, 32);
    if (grp == 0) {
      root_f[(r * NT + q) * 16 + col] = f * a.rate_weights[(size_t)job * R + r];
      root_sc[(r * NT + q) * 16 + col] = sc[q];
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      double l = 0.0;
      if (lane < 16) {
        double term = root_f[q * 16 + lane];
        int smin = root_sc[q * 16 + lane];
        for (unsigned w = 1; w < R; ++w) {
          const double fq = root_f[(w * NT + q) * 16 + lane];
          const int sq = root_sc[(w * NT + q) * 16 + lane];
          if (sq >= smin) {
            term += fq * pow2_neg256(sq - smin);
          } else {
            term = term * pow2_neg256(smin - sq) + fq;
            smin = sq;
          }
        }
        l = log(term) + (double)smin * kLogScaleThreshold;
        l *= (double)a.pattern_weights[ls[q]];
        if (site[q] >= S) l = 0.0;
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) l += __shfl_down(l, off);
      const unsigned tile = blockIdx.x * NT + q;
      if (lane == 0 && tile < a.tiles) a.partials[(size_t)job * a.tiles + tile] = l;
    }
  }
}

// fixed-order finish, one workgroup per job
__global__ void __launch_bounds__(256)
fused20_finish_kernel(const double *__restrict__ partials, unsigned per_job,
                      double *__restrict__ out) {
  __shared__ double lds[4];
  const double *p = partials + (size_t)blockIdx.x * per_job;
  double acc = 0.0;
  for (unsigned i = threadIdx.x; i < per_job; i += 256) acc += p[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ((lds[0] + lds[1]) + lds[2]) + lds[3];
}

// the exporting evaluator left a child's CLV with one rescale count per (site, rate); the root
// kernels (and rdamd_get_clv) read the reference's form, one count per site: the smallest count of
// the site's rates, a rate that was rescaled d times more goes back by 2^(-256 d)
// (fused_export_fixup_kernel's 20-state sibling; one-wave workgroups, both children in one launch)
struct Export20FixupArgs {
  double *clv[2];
  const unsigned *cnt[2];
  unsigned *scaler[2];
  unsigned sites, R, tiles;
};
__global__ void __launch_bounds__(64)
fused20_export_fixup_kernel(Export20FixupArgs x) {
  const unsigned k = blockIdx.y;
  double *__restrict__ clv = x.clv[k];
  const unsigned *__restrict__ cnt = x.cnt[k];
  if (!clv) return;
  const unsigned R = x.R, site = blockIdx.x * 64u + threadIdx.x;
  if (site >= x.sites) return;
  unsigned smin = cnt[(size_t)site * R];
  for (unsigned r = 1; r < R; ++r) smin = min(smin, cnt[(size_t)site * R + r]);
  for (unsigned r = 0; r < R; ++r) {
    const unsigned d = cnt[(size_t)site * R + r] - smin;
    if (d) {
      const double f = d < 4u ? pow2_neg256((int)d) : (d == 4u ? 0x1p-1024 : 0.0);
      double *tile = clv + ((size_t)r * x.tiles + site / 16u) * (16 * kK);
      for (unsigned st = 0; st < (unsigned)kK; ++st) tile[k20_tile_index(site % 16u, st)] *= f;
    }
  }
  x.scaler[k][site] = smin;
}

size_t fused20_qpow_doubles() { return kQPow20Stride; }

hipError_t launch_fused20_pmatrix(const Fused20Args &a, const double *d_q, double *d_qpow, const double *d_rates,
                                  unsigned n_jobs, unsigned n_mat, hipStream_t stream) {
  const size_t total = (size_t)n_jobs * n_mat * a.rate_cats;
  if (!total) return hipSuccess;
  fused20_qpowers_kernel<<<(n_jobs + 3) / 4, 128, 0, stream>>>(d_q, n_jobs, d_qpow);
  fused20_pmatrix_kernel<<<(unsigned)((total + 3) / 4), 128, 0, stream>>>(
      d_qpow, d_rates, a.jobs, n_mat, a.rate_cats, (unsigned)total, const_cast<double *>(a.pmat), a.pmat_job_stride,
      const_cast<double *>(a.tiptab), a.tiptab_job_stride, a.codemask, a.ncodes);
  return hipGetLastError();
}

constexpr int kFused20Tiles = 1;   // 16-site tiles per wave

size_t fused20_lds_bytes(unsigned R, unsigned depth) {
  return (size_t)R * kFused20Tiles * 16 * 12 +
         (size_t)R * (kCopyLds + (size_t)depth * kFused20Tiles * kLevelBytes);
}

// one instantiation: its LDS limit raised when a launch needs more than 64 KB (raised, never lowered:
// partitions launch from their own host threads), then the launch
template <bool EXPORT, int THREADS>
static hipError_t launch_fused20_variant(const Fused20Args &a, unsigned n_jobs, unsigned max_depth, hipStream_t stream) {
  const size_t lds = fused20_lds_bytes(a.rate_cats, max_depth);
  if (lds > 160u * 1024u) return hipErrorInvalidValue;   // (deep stacks at many rate categories)
  {
    static std::mutex lds_mu;
    static size_t lds_allowed = 64 * 1024;
    std::lock_guard<std::mutex> guard(lds_mu);
    if (lds > lds_allowed) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused20_eval_kernel<kFused20Tiles, EXPORT, THREADS>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      lds_allowed = lds;
    }
  }
  const unsigned groups = (a.tiles + kFused20Tiles - 1) / kFused20Tiles;
  fused20_eval_kernel<kFused20Tiles, EXPORT, THREADS><<<dim3(groups, n_jobs), 64 * a.rate_cats, lds, stream>>>(a, max_depth);
  return hipGetLastError();
}

hipError_t launch_fused20_eval(const Fused20Args &a, unsigned n_jobs, unsigned max_depth,
                               double *d_out, hipStream_t stream) {
  if (!n_jobs) return hipSuccess;
  if (a.rate_cats > 8) return hipErrorInvalidValue;
  hipError_t e = a.rate_cats <= 4 ? launch_fused20_variant<false, 256>(a, n_jobs, max_depth, stream)
                                  : launch_fused20_variant<false, 512>(a, n_jobs, max_depth, stream);
  if (e != hipSuccess) return e;
  fused20_finish_kernel<<<n_jobs, 256, 0, stream>>>(a.partials, a.tiles, d_out);
  return hipGetLastError();
}

hipError_t launch_fused20_export(const Fused20Args &a, unsigned max_depth, unsigned *const d_scaler[2],
                                 double *d_out, hipStream_t stream) {
  if (a.rate_cats > 8) return hipErrorInvalidValue;
  hipError_t e = a.rate_cats <= 4 ? launch_fused20_variant<true, 256>(a, 1, max_depth, stream)
                                  : launch_fused20_variant<true, 512>(a, 1, max_depth, stream);
  if (e != hipSuccess) return e;
  fused20_finish_kernel<<<1, 256, 0, stream>>>(a.partials, a.tiles, d_out);
  if (a.export_clv[0] || a.export_clv[1]) {
    Export20FixupArgs x;
    for (int k = 0; k < 2; ++k) { x.clv[k] = a.export_clv[k]; x.cnt[k] = a.export_cnt[k]; x.scaler[k] = d_scaler[k]; }
    x.sites = a.sites; x.R = a.rate_cats; x.tiles = a.tiles;
    fused20_export_fixup_kernel<<<dim3((a.sites + 63u) / 64u, 2), 64, 0, stream>>>(x);
  }
  return hipGetLastError();
}

}  // namespace rdamd
