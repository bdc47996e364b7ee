// Root log-likelihood reduction.
//
// Replaces corax_compute_root_loglikelihood (called at
// /root/reference/src/model.cpp:406, :441, :466):
//   lnL = sum_s w_s * [ log( sum_r omega_r * sum_k pi_k * root[s][r][k] )
//                       + scaler_s * log(2^-256) ]
// (the invariant-sites branch is never active in the
// reference, src/model.cpp:292-300).  The reduction has a fixed shape
// (per-lane strided sums -> wave shuffle tree -> LDS -> one finishing
// workgroup), so a repeated call returns the bit-identical value the
// reference's test demands (test/src/model.cpp:73); no floating-point atomics.
#include "common.hpp"
#include "expm_k4.hpp"

namespace rdamd {

constexpr unsigned kRootBlocks = 1024;

__device__ inline double block_sum_256(double v, double *lds) {
  // wave tree
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = ((lds[0] + lds[1]) + lds[2]) + lds[3];
  return r;
}

// sum_k c[k] * f[k] with a pinned evaluation order (one fma chain), so the
// separate and the fused root kernels produce bit-identical terms
__device__ __forceinline__ double dot_freq(const double *c, const double *f, unsigned K) {
  double tr = 0.0;
  for (unsigned k = 0; k < K; ++k) tr = __builtin_fma(c[k], f[k], tr);
  return tr;
}

// the same chain over a record kept in the 20-state operand layout (common.hpp:
// k20_tile_index); `tile` = the 16-site tile of the record's rate, c = site % 16
__device__ __forceinline__ double dot_freq_k20_tile(const double *tile, unsigned c, const double *f) {
  double tr = 0.0;
  for (unsigned k = 0; k < 20; ++k) tr = __builtin_fma(tile[k20_tile_index(c, k)], f[k], tr);
  return tr;
}

// one lane per (site, rate); R in {1,2,4,8,16}
template <int R>
__global__ void __launch_bounds__(256)
root_lnl_group_kernel(const double *__restrict__ clv, const unsigned *__restrict__ scaler,
                      const double *__restrict__ freqs, const unsigned *__restrict__ fidx,
                      const double *__restrict__ rate_w, const unsigned *__restrict__ pw,
                      unsigned S, unsigned K, double *__restrict__ persite,
                      double *__restrict__ partials) {
  __shared__ double lds[4];
  const size_t total = (size_t)S * R;
  const size_t stride = (size_t)gridDim.x * 256;
  double acc = 0.0;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const unsigned s = (unsigned)(idx / R), r = (unsigned)(idx % R);
    const double *c = clv + idx * K;
    const double *f = freqs + (size_t)fidx[r] * K;
    double tr = dot_freq(c, f, K) * rate_w[r];
    // sum the R rate terms in rate order (same order as the reference loop)
    double term = 0.0;
    const int base = (int)(threadIdx.x & 63) & ~(R - 1);
#pragma unroll
    for (int q = 0; q < R; ++q) term += __shfl(tr, base + q);
    if (r == 0) {
      double l = log(term);
      if (scaler) {
        unsigned sc = scaler[s];
        if (sc) l += (double)sc * kLogScaleThreshold;
      }
      l *= (double)pw[s];
      if (persite) persite[s] = l;
      acc += l;
    }
  }
  double b = block_sum_256(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = b;
}

// one lane per site (any R, K)
__global__ void __launch_bounds__(256)
root_lnl_site_kernel(const double *__restrict__ clv, const unsigned *__restrict__ scaler,
                     const double *__restrict__ freqs, const unsigned *__restrict__ fidx,
                     const double *__restrict__ rate_w, const unsigned *__restrict__ pw,
                     unsigned S, unsigned R, unsigned K, unsigned tiles, double *__restrict__ persite,
                     double *__restrict__ partials) {
  __shared__ double lds[4];
  double acc = 0.0;
  for (unsigned s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
    const double *c = clv + (size_t)s * R * K;
    double term = 0.0;
    for (unsigned r = 0; r < R; ++r) {
      const double *f = freqs + (size_t)fidx[r] * K;
      // tiles != 0: 20-state CLV in the operand layout, [rate][tile][320 doubles]
      const double tr = tiles ? dot_freq_k20_tile(clv + ((size_t)r * tiles + (s >> 4)) * 320, s & 15u, f)
                              : dot_freq(c + (size_t)r * K, f, K);
      term += tr * rate_w[r];
    }
    double l = log(term);
    if (scaler) {
      unsigned sc = scaler[s];
      if (sc) l += (double)sc * kLogScaleThreshold;
    }
    l *= (double)pw[s];
    if (persite) persite[s] = l;
    acc += l;
  }
  double b = block_sum_256(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = b;
}

// The same reduction for MANY root CLVs of one partition in one launch
// (blockIdx.y = which): the grid shape in x, the per-lane order and the block
// tree are those of the single-root kernels, so every value is bit-identical
// to a separate rdamd_compute_root_loglikelihood call.
template <int R>
__global__ void __launch_bounds__(256)
root_lnl_group_batch_kernel(const double *__restrict__ clv_base, const unsigned *__restrict__ scaler_base,
                            const unsigned *__restrict__ clv_rel, const int *__restrict__ scaler_idx,
                            const double *__restrict__ freqs, const unsigned *__restrict__ fidx,
                            const double *__restrict__ rate_w, const unsigned *__restrict__ pw,
                            unsigned S, unsigned K, double *__restrict__ partials) {
  __shared__ double lds[4];
  const unsigned which = blockIdx.y;
  const double *clv = clv_base + (size_t)clv_rel[which] * S * R * K;
  const int sci = scaler_idx[which];
  const unsigned *scaler = sci >= 0 ? scaler_base + (size_t)sci * S : nullptr;
  const size_t total = (size_t)S * R;
  const size_t stride = (size_t)gridDim.x * 256;
  double acc = 0.0;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const unsigned s = (unsigned)(idx / R), r = (unsigned)(idx % R);
    const double *c = clv + idx * K;
    const double *f = freqs + (size_t)fidx[r] * K;
    double tr = dot_freq(c, f, K) * rate_w[r];
    double term = 0.0;
    const int base = (int)(threadIdx.x & 63) & ~(R - 1);
#pragma unroll
    for (int q = 0; q < R; ++q) term += __shfl(tr, base + q);
    if (r == 0) {
      double l = log(term);
      if (scaler) {
        unsigned sc = scaler[s];
        if (sc) l += (double)sc * kLogScaleThreshold;
      }
      l *= (double)pw[s];
      acc += l;
    }
  }
  double b = block_sum_256(acc, lds);
  if (threadIdx.x == 0) partials[(size_t)which * gridDim.x + blockIdx.x] = b;
}

__global__ void __launch_bounds__(256)
root_lnl_site_batch_kernel(const double *__restrict__ clv_base, const unsigned *__restrict__ scaler_base,
                           const unsigned *__restrict__ clv_rel, const int *__restrict__ scaler_idx,
                           const double *__restrict__ freqs, const unsigned *__restrict__ fidx,
                           const double *__restrict__ rate_w, const unsigned *__restrict__ pw,
                           unsigned S, unsigned R, unsigned K, unsigned tiles, size_t clv_doubles,
                           double *__restrict__ partials) {
  __shared__ double lds[4];
  const unsigned which = blockIdx.y;
  const double *clv = clv_base + (size_t)clv_rel[which] * clv_doubles;
  const int sci = scaler_idx[which];
  const unsigned *scaler = sci >= 0 ? scaler_base + (size_t)sci * S : nullptr;
  double acc = 0.0;
  for (unsigned s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
    const double *c = clv + (size_t)s * R * K;
    double term = 0.0;
    for (unsigned r = 0; r < R; ++r) {
      const double *f = freqs + (size_t)fidx[r] * K;
      const double tr = tiles ? dot_freq_k20_tile(clv + ((size_t)r * tiles + (s >> 4)) * 320, s & 15u, f)
                              : dot_freq(c + (size_t)r * K, f, K);
      term += tr * rate_w[r];
    }
    double l = log(term);
    if (scaler) {
      unsigned sc = scaler[s];
      if (sc) l += (double)sc * kLogScaleThreshold;
    }
    l *= (double)pw[s];
    acc += l;
  }
  double b = block_sum_256(acc, lds);
  if (threadIdx.x == 0) partials[(size_t)which * gridDim.x + blockIdx.x] = b;
}

// fixed-order finish: 256 lanes stride over the partials, then the block tree
__global__ void __launch_bounds__(256)
finish_sum_kernel(const double *__restrict__ partials, unsigned n, double *__restrict__ out) {
  __shared__ double lds[4];
  const double *p = partials + (size_t)blockIdx.x * n;   // one workgroup per result
  double acc = 0.0;
  for (unsigned i = threadIdx.x; i < n; i += 256) acc += p[i];
  double b = block_sum_256(acc, lds);
  if (threadIdx.x == 0) out[blockIdx.x] = b;
}

// ---------------------------------------------------------------------------
// Root-only evaluation for 4-state data: the root operation (both child CLVs
// read ONCE) and the log-likelihood reduction for up to four root positions on
// the same branch -- the body of model_t::compute_lh_root / compute_dlh
// (/root/reference/src/model.cpp:415-452, :481-519).  Grid shape, per-lane
// accumulation order and reduction tree are those of root_lnl_group_kernel, so
// each value is bit-identical to update_prob_matrices + update_clvs + that
// kernel.  ONE launch, nothing else on the stream: the two
// P-matrices per position are exponentiated inside the kernel (every block
// repeats the few hundred flops; expm_k4 is the code of pmatrix_k4_kernel, so
// the bits are the same), tip tables are built from them in LDS, the block
// that finishes last folds the partial sums in finish_sum_kernel's order, and
// the result lands in pinned host memory.  A root-only evaluation is
// launch-latency, not bandwidth: one launch + one stream wait instead of three
// copies, three launches and a copy back.
// ---------------------------------------------------------------------------
// (the body: wave `w` of virtual block `bid` of `nblocks` works on one root operation -- the
// whole grid of root_single_dna_kernel, one row of root_multi_dna_kernel's)
//
// ONE WAVE PER WORKGROUP.  The arithmetic is laid out over "virtual blocks" of 256 lanes --
// grid shape, per-lane accumulation order and reduction tree of root_lnl_group_kernel, so every
// value has that kernel's bits -- but each of a block's four waves is a workgroup of its own
// (blockIdx.x = 4 bid + w) and the four wave sums meet in the final fold instead of in LDS.
// Why: these launches run BESIDE the fused evaluator's (the lock-stepped search, model.cpp),
// whose one-wave workgroups fill every wave slot of the device; a 256-lane workgroup starts
// only once four slots of one CU are free at the same moment -- the scheduler holds freed
// slots idle until then -- and a root step that takes 24 us alone cost the evaluator seven
// times that (profiles/r4_e2e_*).  A one-wave workgroup takes the first slot that opens.
template <int R>
__device__ __forceinline__ void
root_single_body(const DeviceView &v, const LevelOp &op, const RootSingleArgs &ra, const double *__restrict__ q,
                 const double *__restrict__ rates, const double *__restrict__ freqs,
                 const double *__restrict__ rate_w, const unsigned *__restrict__ pw,
                 const uint64_t *__restrict__ codemask, double *__restrict__ partials,
                 unsigned *__restrict__ counter, double *__restrict__ result, const unsigned phys,
                 const unsigned n_phys, const unsigned nblocks) {
  constexpr unsigned kMaxPos = R <= 4 ? 8u : 4u;   // root_single_max_positions(R)
  __shared__ double spm[kMaxPos][2][R][16];        // P-matrices, 8 KB
  const unsigned lane = threadIdx.x, S = v.sites;
  const unsigned NA = (unsigned)__builtin_amdgcn_readfirstlane((int)ra.n_positions);
  const bool tip1 = op.child1_clv < v.tips, tip2 = op.child2_clv < v.tips;
  // (`ra` may be a kernel argument: a per-lane index into its arrays would make the compiler
  // copy the whole record into every lane's private memory -- selects over constant indices
  // read it where it is)
  auto pidx_of = [&](unsigned r) {
    unsigned x = ra.params_idx[0];
#pragma unroll
    for (int k = 1; k < R; ++k) x = r == (unsigned)k ? ra.params_idx[k] : x;
    return x;
  };
  auto len_of = [&](unsigned a, unsigned c) {
    double x = c ? ra.len2[0] : ra.len1[0];
#pragma unroll
    for (unsigned k = 1; k < kMaxPos; ++k) x = a == k ? (c ? ra.len2[k] : ra.len1[k]) : x;
    return x;
  };
  // the NA x 2 x R matrices, four at a time: 16 lanes per matrix, a lane per element
  // (expm_k4_coop16: pmatrix_k4_kernel's arithmetic without its 128 registers)
#ifdef RDAMD_ABLATION
  if (ra.abl & 1u) {   // timing only: garbage matrices, no exponentiation
    for (unsigned e = lane; e < NA * 2 * R * 16; e += 64) (&spm[0][0][0][0])[e] = 0.25;
  } else
#endif
  for (unsigned m0 = 0; m0 < NA * 2 * R; m0 += 4) {
    const unsigned m = m0 + (lane >> 4);
    const bool have = m < NA * 2 * R;
    const unsigned mm = have ? m : 0u;
    const unsigned a = mm / (2 * R), c = (mm / R) & 1u, r = mm % R;
    const double e = expm_k4_coop16(q + (size_t)pidx_of(r) * 16, have ? len_of(a, c) * rates[r] : 0.0);
    if (have) spm[a][c][r][lane & 15u] = e <= 0.0 ? 0.0 : e;
  }
  __syncthreads();
  // the state contract: the LAST position's matrices (and their tip tables) are
  // what rdamd_update_prob_matrices would have left in the partition
  if (phys == 0) {
    for (unsigned e = lane; e < 2 * R * 16; e += 64) {
      const unsigned c = e / (R * 16), ww = e % (R * 16);
      const unsigned m = c ? op.child2_mat : op.child1_mat;
      v.pmat[(size_t)m * R * 16 + ww] = spm[NA - 1][c][ww / 16][ww % 16];
    }
    for (unsigned e = lane; e < 2 * R * 64; e += 64) {
      const unsigned c = e / (R * 64), ww = e % (R * 64);
      const unsigned m = c ? op.child2_mat : op.child1_mat;
      const unsigned r = ww / 64, code = (ww / 4) & 15u, i = ww & 3u;
      const uint64_t mask = codemask[code];
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += ((mask >> j) & 1) ? spm[NA - 1][c][r][i * 4 + j] : 0.0;
      v.tiptab[(size_t)m * R * 64 + ww] = acc;
    }
  }

  const size_t total = (size_t)S * R, stride = (size_t)nblocks * 256;
  const double2 *c1 = tip1 ? nullptr
      : reinterpret_cast<const double2 *>(v.clv + (size_t)(op.child1_clv - v.tips) * v.clv_stride);
  const double2 *c2 = tip2 ? nullptr
      : reinterpret_cast<const double2 *>(v.clv + (size_t)(op.child2_clv - v.tips) * v.clv_stride);
  double2 *pc = reinterpret_cast<double2 *>(v.clv + (size_t)(op.parent_clv - v.tips) * v.clv_stride);
  const uint8_t *code1 = tip1 ? v.tipcodes + (size_t)op.child1_clv * v.tip_stride : nullptr;
  const uint8_t *code2 = tip2 ? v.tipcodes + (size_t)op.child2_clv * v.tip_stride : nullptr;
  unsigned *psc = v.scaler + (size_t)op.parent_sc * S;
  const unsigned *lsc = op.child1_sc >= 0 ? v.scaler + (size_t)op.child1_sc * S : nullptr;
  const unsigned *rsc = op.child2_sc >= 0 ? v.scaler + (size_t)op.child2_sc * S : nullptr;

  // One pass over this wave's (site, rate) pairs PER POSITION (the children are re-read: they
  // come from L2, and a lane holds one pair in all but the largest alignments): the registers
  // the pass needs do not grow with the number of positions, and one kernel serves any number.
#pragma unroll 1
  for (unsigned a = 0; a < NA; ++a) {
    const bool last = a + 1 == NA;
    // the virtual waves this workgroup plays, one after the other: wave w of virtual block bid
#pragma unroll 1
    for (unsigned vw = phys; vw < nblocks * 4; vw += n_phys) {
    const unsigned bid = vw >> 2, w = vw & 3u, tid = w * 64u + lane;   // tid: lane of the virtual block
    double acc = 0.0;
    for (size_t idx = (size_t)bid * 256 + tid; idx < total; idx += stride) {
      // (the P-matrices in LDS do not change inside this loop, which usually runs once: left
      // alone the compiler reads all 32 entries ahead of it and keeps them in 64 registers)
      asm volatile("" ::: "memory");
      const unsigned s = (unsigned)(idx / R), r = (unsigned)(idx % R);
      // A tip child is the 0/1 vector of its code's states and goes through the same product as
      // an inner child: P . (0/1 vector) adds the selected entries of a row in state order --
      // the tip table's sum (pmatrix_k4_kernel) term for term, products with 0 and 1 being
      // exact, so the bits are the table's -- and the kernel needs no table in LDS.
      double x[4], y[4];
      if (tip1) {
        const uint64_t mask = codemask[code1[s]];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = (double)((mask >> j) & 1);
      } else { const double2 u = c1[idx * 2], b = c1[idx * 2 + 1]; x[0] = u.x; x[1] = u.y; x[2] = b.x; x[3] = b.y; }
      if (tip2) {
        const uint64_t mask = codemask[code2[s]];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (double)((mask >> j) & 1);
      } else { const double2 u = c2[idx * 2], b = c2[idx * 2 + 1]; y[0] = u.x; y[1] = u.y; y[2] = b.x; y[3] = b.y; }
      const unsigned sc0 = (lsc ? lsc[s] : 0u) + (rsc ? rsc[s] : 0u);
      const double *f = freqs + (size_t)pidx_of(r) * 4;
      const double wgt = rate_w[r];
      const double weight = (double)pw[s];
      const int base = (int)lane & ~(R - 1);
      double t1[4], t2[4], o[4];
      {
        const double *m = &spm[a][0][r][0];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          t1[k] = m[k * 4 + 0] * x[0] + m[k * 4 + 1] * x[1] + m[k * 4 + 2] * x[2] + m[k * 4 + 3] * x[3];
      }
      {
        const double *m = &spm[a][1][r][0];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          t2[k] = m[k * 4 + 0] * y[0] + m[k * 4 + 1] * y[1] + m[k * 4 + 2] * y[2] + m[k * 4 + 3] * y[3];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = t1[k] * t2[k];
      int small = (o[0] < kScaleThreshold) & (o[1] < kScaleThreshold) &
                  (o[2] < kScaleThreshold) & (o[3] < kScaleThreshold);
#pragma unroll
      for (int off = 1; off < R; off <<= 1) small &= __shfl_xor(small, off);
      unsigned sc = sc0;
      if (small) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] *= kScaleFactor;
        sc += 1;
      }
      const double tr = dot_freq(o, f, 4) * wgt;
      double term = 0.0;
#pragma unroll
      for (int q2 = 0; q2 < R; ++q2) term += __shfl(tr, base + q2);
      if (r == 0) {
        double l = log(term);
        if (sc) l += (double)sc * kLogScaleThreshold;
        acc += l * weight;
      }
      if (last) {
        if (r == 0) psc[s] = sc;
        pc[idx * 2] = make_double2(o[0], o[1]);
        pc[idx * 2 + 1] = make_double2(o[2], o[3]);
      }
    }
    // this wave's share of block_sum_256: the wave tree; the four wave sums of a virtual block
    // are combined -- ((w0 + w1) + w2) + w3, block_sum_256's order -- in the final fold
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) partials[((size_t)a * nblocks + bid) * 4 + w] = acc;
    }
  }
  // the wave that arrives last folds the partials (finish_sum_kernel's order: lane t of a
  // 256-lane block adds the block sums t, t + 256, ..., then block_sum_256 over the lanes --
  // one wave plays the block's four in turn)
  __threadfence();
  unsigned ticket = 0;
  if (lane == 0) ticket = atomicAdd(counter, 1u);
  ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
  if (ticket != n_phys - 1) return;
  __threadfence();
  for (unsigned a = 0; a < NA; ++a) {
    const volatile double *p = partials + (size_t)a * nblocks * 4;
    double wave_sum[4];
#pragma unroll
    for (unsigned vw = 0; vw < 4; ++vw) {
      double sum = 0.0;
      for (unsigned i = vw * 64 + lane; i < nblocks; i += 256)
        sum += ((p[i * 4] + p[i * 4 + 1]) + p[i * 4 + 2]) + p[i * 4 + 3];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
      wave_sum[vw] = sum;
    }
    if (lane == 0) result[a] = ((wave_sum[0] + wave_sum[1]) + wave_sum[2]) + wave_sum[3];
  }
  if (lane == 0) *counter = 0u;   // ready for the next launch on this stream
}

// (at least four waves per SIMD = at most 128 registers: a wave of these kernels must fit the
// slot a wave of the fused evaluator leaves)
template <int R>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
root_single_dna_kernel(DeviceView v, LevelOp op, RootSingleArgs ra, const double *__restrict__ q,
                       const double *__restrict__ rates, const double *__restrict__ freqs,
                       const double *__restrict__ rate_w, const unsigned *__restrict__ pw,
                       const uint64_t *__restrict__ codemask, double *__restrict__ partials,
                       unsigned *__restrict__ counter, double *__restrict__ result, unsigned nblocks) {
  root_single_body<R>(v, op, ra, q, rates, freqs, rate_w, pw, codemask, partials, counter, result,
                      blockIdx.x, gridDim.x, nblocks);
}

// The same for SEVERAL partitions at once (grid.y): the root-only steps of the candidates that
// a lock-stepped search has in flight -- every one on its own model replica, hence its own
// CLVs, parameters and scratch -- as one launch instead of one launch per candidate and step
// (rdamd_root_loglikelihood_fused_multi).  Every row keeps the grid shape it would have had
// alone, so each value has the bits of the single launch.
template <int R>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
root_multi_dna_kernel(const RootItem *__restrict__ items) {
  const RootItem &it = items[blockIdx.y];
  // (a row with fewer virtual waves than the launch has workgroups per row uses as many as it has)
  const unsigned n_phys = min(gridDim.x, it.blocks * 4u);
  if (blockIdx.x >= n_phys) return;
  root_single_body<R>(it.v, it.op, it.ra, it.q, it.rates, it.freqs, it.rate_w, it.pw, it.codemask,
                      it.partials, it.counter, it.result, blockIdx.x, n_phys, it.blocks);
}

hipError_t launch_root_lnl(rdamd_partition *p, unsigned clv_index, int scaler_index,
                           const unsigned *d_fidx, double *d_persite, double *d_out) {
  const unsigned S = p->sites, R = p->rate_cats, K = p->states;
  const double *clv = p->d_clv + (size_t)(clv_index - p->tips) * p->clv_doubles();
  const unsigned *sc = scaler_index >= 0 ? p->d_scaler + (size_t)scaler_index * S : nullptr;
  const unsigned tiles = p->mfma_layout ? p->clv_tiles() : 0u;
  unsigned blocks;
  // (a CLV in the 20-state operand layout goes through the one-lane-per-site kernel)
  const bool group = !p->mfma_layout && (R == 1 || R == 2 || R == 4 || R == 8 || R == 16);
  if (group) {
    size_t total = (size_t)S * R;
    blocks = (unsigned)((total + 255) / 256);
  } else {
    blocks = (S + 255) / 256;
  }
  if (blocks > kRootBlocks) blocks = kRootBlocks;
  if (blocks == 0) blocks = 1;
#define RDAMD_ROOT_ARGS clv, sc, p->d_freqs, d_fidx, p->d_rate_weights, p->d_pattern_weights
  if (group) {
    switch (R) {
      case 1: root_lnl_group_kernel<1><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
      case 2: root_lnl_group_kernel<2><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
      case 4: root_lnl_group_kernel<4><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
      case 8: root_lnl_group_kernel<8><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
      default: root_lnl_group_kernel<16><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
    }
  } else {
    root_lnl_site_kernel<<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, R, K, tiles, d_persite, p->d_partials);
  }
#undef RDAMD_ROOT_ARGS
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  finish_sum_kernel<<<1, 256, 0, p->stream>>>(p->d_partials, blocks, d_out);
  return hipGetLastError();
}

unsigned root_lnl_blocks(const rdamd_partition *p) {
  const unsigned S = p->sites, R = p->rate_cats;
  const bool group = !p->mfma_layout && (R == 1 || R == 2 || R == 4 || R == 8 || R == 16);
  unsigned blocks = group ? (unsigned)(((size_t)S * R + 255) / 256) : (S + 255) / 256;
  if (blocks > kRootBlocks) blocks = kRootBlocks;
  return blocks ? blocks : 1;
}

// d_clv_rel: CLV indices minus `tips`; d_partials: count * root_lnl_blocks(p) doubles
hipError_t launch_root_lnl_batch(rdamd_partition *p, unsigned count, const unsigned *d_clv_rel,
                                 const int *d_scaler_idx, const unsigned *d_fidx,
                                 double *d_partials, double *d_out) {
  if (!count) return hipSuccess;
  const unsigned S = p->sites, R = p->rate_cats, K = p->states;
  const unsigned blocks = root_lnl_blocks(p);
  const dim3 grid(blocks, count);
  const unsigned tiles = p->mfma_layout ? p->clv_tiles() : 0u;
#define RDAMD_RB_ARGS p->d_clv, p->d_scaler, d_clv_rel, d_scaler_idx, p->d_freqs, d_fidx, p->d_rate_weights, p->d_pattern_weights
  switch (p->mfma_layout ? 0u : R) {
    case 1: root_lnl_group_batch_kernel<1><<<grid, 256, 0, p->stream>>>(RDAMD_RB_ARGS, S, K, d_partials); break;
    case 2: root_lnl_group_batch_kernel<2><<<grid, 256, 0, p->stream>>>(RDAMD_RB_ARGS, S, K, d_partials); break;
    case 4: root_lnl_group_batch_kernel<4><<<grid, 256, 0, p->stream>>>(RDAMD_RB_ARGS, S, K, d_partials); break;
    case 8: root_lnl_group_batch_kernel<8><<<grid, 256, 0, p->stream>>>(RDAMD_RB_ARGS, S, K, d_partials); break;
    case 16: root_lnl_group_batch_kernel<16><<<grid, 256, 0, p->stream>>>(RDAMD_RB_ARGS, S, K, d_partials); break;
    default: root_lnl_site_batch_kernel<<<grid, 256, 0, p->stream>>>(RDAMD_RB_ARGS, S, R, K, tiles, p->clv_doubles(), d_partials); break;
  }
#undef RDAMD_RB_ARGS
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  finish_sum_kernel<<<count, 256, 0, p->stream>>>(d_partials, blocks, d_out);
  return hipGetLastError();
}

// Workgroups (= waves) a root step is launched with.  Its arithmetic is laid out over
// blocks x 4 VIRTUAL waves (above); how many physical waves play them changes no bit.  With the
// device to itself a step wants them all side by side (24 us on c2).  BESIDE the objective
// launches of a lock-stepped search (the replicas' streams run on high priority then, model_c_api.cpp)
// every workgroup that is dispatched between the evaluator's costs that kernel ~27 ns of the whole
// device, whatever the workgroup does -- measured, profiles/root_interference.py: 3 128 workgroups
// 85 us per step, 782: 22 us, 196: 5 us, with or without the exponentiation in them -- so there a
// wave plays 16 virtual ones.
static unsigned root_workgroups(unsigned blocks, bool beside) {
  const unsigned vw = blocks * 4u;
  unsigned per_wave = 16u;
#ifdef RDAMD_ABLATION
  if (getenv("RDAMD_ROOT_PER_WAVE")) per_wave = (unsigned)std::max(1, atoi(getenv("RDAMD_ROOT_PER_WAVE")));
#endif
  return beside ? std::max(1u, (vw + per_wave - 1u) / per_wave) : vw;
}

template <int R>
static hipError_t launch_root_single_r(rdamd_partition *p, const DeviceView &v, const LevelOp &op,
                                       const RootSingleArgs &ra, unsigned blocks, unsigned *d_counter,
                                       double *result) {
  root_single_dna_kernel<R><<<root_workgroups(blocks, p->stream_priority < 0), 64, 0, p->stream>>>(
      v, op, ra, p->d_q, p->d_rates, p->d_freqs, p->d_rate_weights, p->d_pattern_weights,
      p->d_codemask, p->d_partials, d_counter, result, blocks);
  return hipGetLastError();
}

// lengths / parameter indices travel as kernel arguments; `result` must be
// device-visible (the partition's pinned host block)
hipError_t launch_root_single(rdamd_partition *p, const LevelOp &op, const double *len1,
                              const double *len2, unsigned n_positions,
                              const unsigned *params_indices, unsigned *d_counter, double *result) {
  const unsigned S = p->sites, R = p->rate_cats;
  if (n_positions == 0 || n_positions > root_single_max_positions(R) || R > 8) return hipErrorInvalidValue;
  RootSingleArgs ra;
  for (unsigned a = 0; a < kRootMaxPositions; ++a) {
    ra.len1[a] = len1[a < n_positions ? a : n_positions - 1];
    ra.len2[a] = len2[a < n_positions ? a : n_positions - 1];
  }
  for (unsigned r = 0; r < 8; ++r) ra.params_idx[r] = r < R ? params_indices[r] : 0u;
  ra.n_positions = n_positions;
  size_t total = (size_t)S * R;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > kRootBlocks) blocks = kRootBlocks;   // same shape as launch_root_lnl
  if (blocks == 0) blocks = 1;
#ifdef RDAMD_ABLATION
  ra.abl = getenv("RDAMD_ROOT_NOEXPM") ? 1u : 0u;
#endif
  const DeviceView v = p->view();
  switch (R) {
    case 1: return launch_root_single_r<1>(p, v, op, ra, blocks, d_counter, result);
    case 2: return launch_root_single_r<2>(p, v, op, ra, blocks, d_counter, result);
    case 4: return launch_root_single_r<4>(p, v, op, ra, blocks, d_counter, result);
    case 8: return launch_root_single_r<8>(p, v, op, ra, blocks, d_counter, result);
    default: return hipErrorInvalidValue;
  }
}

unsigned root_single_blocks(const rdamd_partition *p) {
  size_t total = (size_t)p->sites * p->rate_cats;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > kRootBlocks) blocks = kRootBlocks;   // same shape as launch_root_lnl
  return blocks ? blocks : 1u;
}

hipError_t launch_root_multi(const RootItem *d_items, unsigned n_items, unsigned R, unsigned max_positions,
                             unsigned max_blocks, hipStream_t stream) {
  if (!n_items) return hipSuccess;
  const dim3 grid(root_workgroups(max_blocks, true), n_items);   // (the lock-stepped search's call: always beside)
  if (max_positions > root_single_max_positions(R)) return hipErrorInvalidValue;
  switch (R) {
    case 1: root_multi_dna_kernel<1><<<grid, 64, 0, stream>>>(d_items); break;
    case 2: root_multi_dna_kernel<2><<<grid, 64, 0, stream>>>(d_items); break;
    case 4: root_multi_dna_kernel<4><<<grid, 64, 0, stream>>>(d_items); break;
    case 8: root_multi_dna_kernel<8><<<grid, 64, 0, stream>>>(d_items); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace rdamd
