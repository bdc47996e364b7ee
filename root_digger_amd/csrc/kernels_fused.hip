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
// minimises (Sethi-Ullman order), tips are expanded from their 1-byte codes.
// Because the rate is wave-uniform, both 4x4 P-matrices of an operation are
// scalar (SGPR) operands of the FMAs: no LDS or VGPR traffic for them.
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

__device__ __forceinline__ void expand_tip(unsigned code, double (&x)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = ((code >> j) & 1u) ? 1.0 : 0.0;
}

// exact 2^(-256 * d) for d >= 0 (0 once it underflows)
__device__ __forceinline__ double pow2_neg256(int d) {
  return d == 0 ? 1.0 : (d == 1 ? kScaleThreshold
       : (d == 2 ? kScaleThreshold * kScaleThreshold
       : (d == 3 ? kScaleThreshold * kScaleThreshold * kScaleThreshold : 0.0)));
}

__global__ void __launch_bounds__(64)
fused_dna_eval_kernel(FusedArgs a) {
  extern __shared__ double lds[];   // [depth][64 lanes][4] doubles, then [depth][64] ints
  const unsigned lane = threadIdx.x;
  const unsigned job = blockIdx.y;
  const unsigned S = a.sites, R = a.rate_cats;
  unsigned site = blockIdx.x * 64 + lane;
  const bool valid = site < S;
  if (!valid) site = S - 1;

  const FusedJob jb = a.jobs[job];
  const FusedOp *__restrict__ prog = jb.prog;
  const unsigned nops = jb.n_ops;
  const double *__restrict__ pm = a.pmat + (size_t)job * a.pmat_job_stride;
  const double *__restrict__ freqs = a.freqs + (size_t)job * 4;
  const double *__restrict__ rw = a.rate_weights + (size_t)job * R;
  const uint8_t *__restrict__ tips = a.tipcodes + site;
  double *stk = lds;
  int *stk_sc = reinterpret_cast<int *>(lds + (size_t)jb.depth * 64 * 4);

  double term = 0.0;   // sum_r w_r f_r 2^(-256 (s_r - smin))
  int smin = 0;

  for (unsigned r = 0; r < R; ++r) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    int sc = 0;
    unsigned sp = 0;
    // software prefetch of the next operation's tip codes
    unsigned cx = tips[(size_t)prog[0].tipX * S], cy = tips[(size_t)prog[0].tipY * S];
    for (unsigned i = 0; i < nops; ++i) {
      const FusedOp op = prog[i];
      const unsigned nxt = i + 1 < nops ? i + 1 : i;
      const unsigned ncx = tips[(size_t)prog[nxt].tipX * S];
      const unsigned ncy = tips[(size_t)prog[nxt].tipY * S];
      const double *__restrict__ px = pm + ((size_t)op.matX * R + r) * 16;
      const double *__restrict__ py = pm + ((size_t)op.matY * R + r) * 16;

      if (op.spill) {   // wave-uniform: park the running CLV for a later pop
        double *d = stk + ((size_t)sp * 64 + lane) * 4;
        reinterpret_cast<double2 *>(d)[0] = make_double2(v[0], v[1]);
        reinterpret_cast<double2 *>(d)[1] = make_double2(v[2], v[3]);
        stk_sc[sp * 64 + lane] = sc;
        ++sp;
      }
      double x[4], y[4];
      int scx, scy;
      if (op.kind == kFusedTT) {
        expand_tip(cx, x);
        scx = 0;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = v[j];
        scx = sc;
      }
      if (op.kind == kFusedRP) {
        --sp;
        const double *d = stk + ((size_t)sp * 64 + lane) * 4;
        const double2 lo = reinterpret_cast<const double2 *>(d)[0];
        const double2 hi = reinterpret_cast<const double2 *>(d)[1];
        y[0] = lo.x; y[1] = lo.y; y[2] = hi.x; y[3] = hi.y;
        scy = stk_sc[sp * 64 + lane];
      } else {
        expand_tip(cy, y);
        scy = 0;
      }
      double o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double tx = px[k * 4 + 0] * x[0] + px[k * 4 + 1] * x[1] + px[k * 4 + 2] * x[2] + px[k * 4 + 3] * x[3];
        const double ty = py[k * 4 + 0] * y[0] + py[k * 4 + 1] * y[1] + py[k * 4 + 2] * y[2] + py[k * 4 + 3] * y[3];
        o[k] = tx * ty;
      }
      sc = scx + scy;
      if ((o[0] < kScaleThreshold) & (o[1] < kScaleThreshold) & (o[2] < kScaleThreshold) &
          (o[3] < kScaleThreshold)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] *= kScaleFactor;
        sc += 1;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = o[k];
      cx = ncx;
      cy = ncy;
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
                        size_t pmat_job_stride) {
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
  for (int i = 0; i < 16; ++i) pmo[i] = out[i] < 0.0 ? 0.0 : out[i];
}

hipError_t launch_fused_pmatrix(const FusedArgs &a, const double *d_q, const double *d_rates,
                                unsigned n_jobs, unsigned n_mat, hipStream_t stream) {
  const size_t total = (size_t)n_jobs * n_mat * a.rate_cats;
  if (!total) return hipSuccess;
  fused_pmatrix_k4_kernel<<<(unsigned)((total + 63) / 64), 64, 0, stream>>>(
      d_q, d_rates, a.jobs, n_jobs, n_mat, a.rate_cats, const_cast<double *>(a.pmat), a.pmat_job_stride);
  return hipGetLastError();
}

hipError_t launch_fused_eval(const FusedArgs &a, unsigned n_jobs, unsigned max_depth,
                             unsigned blocks_x, double *d_out, hipStream_t stream) {
  if (!n_jobs) return hipSuccess;
  const size_t lds = (size_t)(max_depth ? max_depth : 1) * 64 * (4 * sizeof(double) + sizeof(int));
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
