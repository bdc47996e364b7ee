// Fused full-traversal evaluator (4-state data).
//
// One launch evaluates a BATCH of candidate-root / parameter-set jobs: for each
// job the whole post-order traversal (n-1 CLV operations) plus the root
// log-likelihood, i.e. the body of model_t::compute_lh_partition
// (/root/reference/src/model.cpp:454-476: corax_update_clvs followed by
// corax_compute_root_loglikelihood), WITHOUT materialising any CLV in HBM.
//
// Mapping: one wave = 64 site patterns, one lane = one site.  A lane walks the
// traversal once per rate category; the running CLV (4 doubles) lives in
// registers, pending sibling CLVs on a per-wave LDS stack whose depth the host
// minimises (Sethi-Ullman order).  Because the rate is wave-uniform, the 4x4
// P-matrix of an inner operand is a set of scalar (SGPR) operands of the FMAs:
// no LDS or VGPR traffic for it.  A tip operand costs no FMA at all: its 16x4
// table (one row per ambiguity code, built next to the P-matrices) is exactly
// 64 doubles, one per lane, dropped into LDS and read back by code.
// HBM traffic per evaluation drops from ~(2n-2) CLVs to n bytes per site, so
// the kernel is bound by FP64 FMA issue, not by HBM.
//
// Scaling: each (site, rate) lane keeps its own 2^256 rescale count (rescale
// when all four entries drop below 2^-256) and the root sum aligns the rate
// terms to the smallest count -- the per-rate-scaler form of the reference
// rule; every factor is an exact power of two, so the
// result differs from the per-site rule only where that rule would already
// have lost the category to underflow.
#include "common.hpp"
#include "fused.hpp"

namespace rdamd {

// exact 2^(-256 * d) for d >= 0 (0 once it underflows)
__device__ __forceinline__ double pow2_neg256(int d) {
  return d == 0 ? 1.0 : (d == 1 ? kScaleThreshold
       : (d == 2 ? kScaleThreshold * kScaleThreshold
       : (d == 3 ? kScaleThreshold * kScaleThreshold * kScaleThreshold : 0.0)));
}

__device__ __forceinline__ unsigned uni(unsigned x) {   // assert wave-uniformity
  return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}

// LDS per wave: [2][64] doubles of tip-table slots, then the CLV stack
// [depth][64 lanes][4] doubles, then the rescale-count stack [depth][64] ints.
constexpr unsigned kTabDoubles = 128;

typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
  // descriptor inputs made provably wave-uniform (cdna_hip_programming.md T20)
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = uni((unsigned)u), hi = uni((unsigned)(u >> 32));
  void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)uni(bytes), 0x00020000);
}

__global__ void __launch_bounds__(64)
fused_dna_eval_kernel(FusedArgs a) {
  extern __shared__ double lds[];
  const unsigned lane = threadIdx.x;
  const unsigned job = blockIdx.y;
  const unsigned S = a.sites, R = a.rate_cats;
  unsigned site = blockIdx.x * 64 + lane;
  const bool valid = site < S;
  if (!valid) site = S - 1;

  const FusedJob jb = a.jobs[job];
  const FusedOp *__restrict__ prog = jb.prog;   // n_ops + 2 entries (tail padded)
  const unsigned nops = jb.n_ops;
  const double *__restrict__ pm = a.pmat + (size_t)job * a.pmat_job_stride;
  const double *__restrict__ freqs = a.freqs + (size_t)job * 4;
  const double *__restrict__ rw = a.rate_weights + (size_t)job * R;
  // tip codes and this job's tip tables through buffer descriptors: the
  // per-operation part of every address is a scalar offset, the per-lane part
  // a loop-invariant VGPR, so address generation costs no vector instruction
  const __amdgpu_buffer_rsrc_t tips_rs = make_rsrc(a.tipcodes, a.tipcodes_bytes);
  const __amdgpu_buffer_rsrc_t tab_rs =
      make_rsrc(a.tiptab + (size_t)job * a.pmat_job_stride * 4, (unsigned)(a.pmat_job_stride * 32));
  const int lane8 = (int)lane * 8;
  const int site_off = (int)site;
  double *tabx = lds, *taby = lds + 64;
  double *stk = lds + kTabDoubles;
  int *stk_sc = reinterpret_cast<int *>(stk + (size_t)jb.depth * 64 * 4);

  double term = 0.0;   // sum_r w_r f_r 2^(-256 (s_r - smin))
  int smin = 0;

#define RDAMD_LOAD_CODE(tiprow) \
  ((unsigned)__builtin_amdgcn_raw_buffer_load_b8(tips_rs, site_off, (int)(uni(tiprow) * S), 0))
#define RDAMD_LOAD_TAB(mat) \
  __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64( \
      tab_rs, lane8, (int)((uni(mat) * R + r) * 512u), 0))

  for (unsigned r = 0; r < R; ++r) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    int sc = 0;
    unsigned sp = 0;
    // software pipeline: descriptor i+2 (scalar load), tip codes and tip-table
    // entries of op i+1 (vector loads) are in flight while op i computes
    FusedOp cur = prog[0];
    FusedOp nxt = prog[1];
    unsigned cx = RDAMD_LOAD_CODE(cur.tipX), cy = RDAMD_LOAD_CODE(cur.tipY);
    double ex = RDAMD_LOAD_TAB(cur.mats & 0xffffu), ey = RDAMD_LOAD_TAB(cur.mats >> 16);
    for (unsigned i = 0; i < nops; ++i) {
      const FusedOp nn = prog[i + 2];
      const unsigned kind = uni(cur.flags & 0xffu), spill = uni(cur.flags >> 8);
      // stage A: hand the prefetched tip data of THIS op to LDS
      const double *rowx = tabx + cx * 4, *rowy = taby + cy * 4;
      if (kind == kFusedTT) tabx[lane] = ex;
      if (kind != kFusedRP) taby[lane] = ey;
      // stage B: refill the prefetch registers with the NEXT op's tip data
      cx = RDAMD_LOAD_CODE(nxt.tipX);
      cy = RDAMD_LOAD_CODE(nxt.tipY);
      ex = RDAMD_LOAD_TAB(nxt.mats & 0xffffu);
      ey = RDAMD_LOAD_TAB(nxt.mats >> 16);
      // stage C: the operation itself
      if (spill) {   // wave-uniform: park the running CLV for a later pop
        double *d = stk + ((size_t)sp * 64 + lane) * 4;
        reinterpret_cast<double2 *>(d)[0] = make_double2(v[0], v[1]);
        reinterpret_cast<double2 *>(d)[1] = make_double2(v[2], v[3]);
        stk_sc[sp * 64 + lane] = sc;
        ++sp;
      }
      double tx[4], ty[4];
      int scx, scy;
      if (kind == kFusedTT) {
        // tip operand: row `code` of the 16x4 table of this (matrix, rate)
        const double2 lo = reinterpret_cast<const double2 *>(rowx)[0];
        const double2 hi = reinterpret_cast<const double2 *>(rowx)[1];
        tx[0] = lo.x; tx[1] = lo.y; tx[2] = hi.x; tx[3] = hi.y;
        scx = 0;
      } else {
        const double *__restrict__ px = pm + (size_t)(uni(cur.mats & 0xffffu) * R + r) * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          tx[k] = px[k * 4 + 0] * v[0] + px[k * 4 + 1] * v[1] + px[k * 4 + 2] * v[2] + px[k * 4 + 3] * v[3];
        scx = sc;
      }
      if (kind == kFusedRP) {
        --sp;
        const double *d = stk + ((size_t)sp * 64 + lane) * 4;
        const double2 lo = reinterpret_cast<const double2 *>(d)[0];
        const double2 hi = reinterpret_cast<const double2 *>(d)[1];
        const double *__restrict__ py = pm + (size_t)(uni(cur.mats >> 16) * R + r) * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          ty[k] = py[k * 4 + 0] * lo.x + py[k * 4 + 1] * lo.y + py[k * 4 + 2] * hi.x + py[k * 4 + 3] * hi.y;
        scy = stk_sc[sp * 64 + lane];
      } else {
        const double2 lo = reinterpret_cast<const double2 *>(rowy)[0];
        const double2 hi = reinterpret_cast<const double2 *>(rowy)[1];
        ty[0] = lo.x; ty[1] = lo.y; ty[2] = hi.x; ty[3] = hi.y;
        scy = 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = tx[k] * ty[k];
      sc = scx + scy;
      // all four entries < 2^-256  <=>  the largest high word < that of 2^-256
      // (entries are non-negative; a NaN compares as large and never rescales)
      const unsigned hmax = max(max((unsigned)__double2hiint(v[0]), (unsigned)__double2hiint(v[1])),
                                max((unsigned)__double2hiint(v[2]), (unsigned)__double2hiint(v[3])));
      if (hmax < 0x2FF00000u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= kScaleFactor;
        sc += 1;
      }
      cur = nxt;
      nxt = nn;
    }
    // root: f_r = sum_k pi_k v[k]; fold into the running rate sum
    double f = v[0] * freqs[0] + v[1] * freqs[1] + v[2] * freqs[2] + v[3] * freqs[3];
    f *= rw[r];
    if (r == 0) {
      term = f;
      smin = sc;
    } else if (sc >= smin) {
      term += f * pow2_neg256(sc - smin);
    } else {
      term = term * pow2_neg256(smin - sc) + f;
      smin = sc;
    }
  }
#undef RDAMD_LOAD_CODE
#undef RDAMD_LOAD_TAB

  double l = log(term) + (double)smin * kLogScaleThreshold;
  l *= (double)a.pattern_weights[site];
  if (!valid) l = 0.0;
  if (a.persite && valid) a.persite[(size_t)job * S + site] = l;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) l += __shfl_down(l, off);
  if (lane == 0) a.partials[(size_t)job * gridDim.x + blockIdx.x] = l;
}

// fixed-order finish, one workgroup per job
__global__ void __launch_bounds__(256)
fused_finish_kernel(const double *__restrict__ partials, unsigned per_job,
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

// P-matrices for a batch of jobs: one thread per (job, matrix, rate).
// Same scaling-and-squaring / 16-term Taylor core as pmatrix_k4_kernel.
__global__ void __launch_bounds__(64)
fused_pmatrix_k4_kernel(const double *__restrict__ q, const double *__restrict__ rates,
                        const FusedJob *__restrict__ jobs, unsigned n_jobs,
                        unsigned n_mat, unsigned R, double *__restrict__ pmat,
                        double *__restrict__ tiptab, size_t pmat_job_stride) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per_job = (size_t)n_mat * R;
  if (gid >= per_job * n_jobs) return;
  const unsigned job = (unsigned)(gid / per_job);
  const unsigned rem = (unsigned)(gid % per_job);
  const unsigned m = rem / R, r = rem % R;
  const double t = jobs[job].brlen[m] * rates[(size_t)job * R + r];
  const double *qq = q + (size_t)job * 16;
  double x[16], term[16], out[16], tmp[16];
  double norm = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double cs = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[i * 4 + j] = qq[i * 4 + j] * t;
      cs += fabs(x[i * 4 + j]);
    }
    norm = fmax(norm, cs);
  }
  int s = 0;
  double scale = 1.0;
  while (norm * scale > 0.25 && s < 60) { scale *= 0.5; ++s; }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x[i] *= scale;
    term[i] = out[i] = (i % 5 == 0) ? 1.0 : 0.0;
  }
  for (int k = 1; k <= 16; ++k) {
    const double inv = 1.0 / (double)k;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l) acc += term[i * 4 + l] * x[l * 4 + j];
        tmp[i * 4 + j] = acc * inv;
      }
#pragma unroll
    for (int i = 0; i < 16; ++i) { term[i] = tmp[i]; out[i] += tmp[i]; }
  }
  for (int k = 0; k < s; ++k) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l) acc += out[i * 4 + l] * out[l * 4 + j];
        tmp[i * 4 + j] = acc;
      }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = tmp[i];
  }
  double *pmo = pmat + (size_t)job * pmat_job_stride + ((size_t)m * R + r) * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    out[i] = out[i] < 0.0 ? 0.0 : out[i];
    pmo[i] = out[i];
  }
  // tip table of this (matrix, rate): row c = sum over the states in code c
  double *tto = tiptab + ((size_t)job * pmat_job_stride + ((size_t)m * R + r) * 16) * 4;
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((c >> j) & 1) acc += out[i * 4 + j];
      tto[c * 4 + i] = acc;
    }
}

hipError_t launch_fused_pmatrix(const FusedArgs &a, const double *d_q, const double *d_rates,
                                unsigned n_jobs, unsigned n_mat, hipStream_t stream) {
  const size_t total = (size_t)n_jobs * n_mat * a.rate_cats;
  if (!total) return hipSuccess;
  fused_pmatrix_k4_kernel<<<(unsigned)((total + 63) / 64), 64, 0, stream>>>(
      d_q, d_rates, a.jobs, n_jobs, n_mat, a.rate_cats, const_cast<double *>(a.pmat),
      const_cast<double *>(a.tiptab), a.pmat_job_stride);
  return hipGetLastError();
}

hipError_t launch_fused_eval(const FusedArgs &a, unsigned n_jobs, unsigned max_depth,
                             unsigned blocks_x, double *d_out, hipStream_t stream) {
  if (!n_jobs) return hipSuccess;
  const size_t lds = kTabDoubles * sizeof(double) +
                     (size_t)(max_depth ? max_depth : 1) * 64 * (4 * sizeof(double) + sizeof(int));
  static size_t lds_limit_set = 0;
  if (lds > 48 * 1024 && lds > lds_limit_set) {
    hipError_t e = hipFuncSetAttribute((const void *)fused_dna_eval_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_limit_set = lds;
  }
  dim3 grid(blocks_x, n_jobs);
  fused_dna_eval_kernel<<<grid, 64, lds, stream>>>(a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  fused_finish_kernel<<<n_jobs, 256, 0, stream>>>(a.partials, blocks_x, d_out);
  return hipGetLastError();
}

}  // namespace rdamd
