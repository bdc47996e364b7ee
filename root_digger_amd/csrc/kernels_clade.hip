// Clade tables: the per-job look-up tables of the fused evaluator's pseudo-tips, and the
// per-partition cache of directed subtrees behind them (clades.hpp has the idea).
// Match: coraxlib's site repeats, switched on by the reference for every 4-state run
// (/root/reference/src/model.cpp:145-149).
#include <algorithm>
#include <cstring>

#include "clade_classes.hpp"
#include "clades.hpp"
#include "common.hpp"
#include "fused.hpp"

namespace rdamd {

CladeCache::~CladeCache() {
  if (d_maps) (void)hipFree(d_maps);
}
void clade_cache_free(CladeCache *c) { delete c; }

// ---- host: classes of a directed subtree (clade_classes.hpp) -------------------------------
unsigned clade_intern(rdamd_partition *p, unsigned child0, unsigned child1, unsigned mat0, unsigned mat1) {
  if (!p->clades) p->clades = new CladeCache();
  CladeCache &c = *p->clades;
  const std::array<unsigned, 4> key = {child0, child1, mat0, mat1};
  auto hit = c.intern.find(key);
  if (hit != c.intern.end()) return hit->second;
  const unsigned tips = p->tips;
  CladeNode n;
  n.child[0] = child0; n.child[1] = child1; n.mat[0] = mat0; n.mat[1] = mat1;
  const size_t S = p->sites;
  const uint8_t *cls[2];
  unsigned cnt[2];
  bool small = true;
  for (int k = 0; k < 2; ++k) {
    const unsigned ch = k ? child1 : child0;
    if (ch < tips) {
      cls[k] = p->tipcodes.data() + (size_t)ch * S;
      cnt[k] = 16;   // (4-state codes are the state masks themselves)
      n.n_tips += 1;
    } else {
      const CladeNode &cn = c.nodes[ch - tips];
      n.n_tips += cn.n_tips;
      if (cn.n_classes == 0) small = false;
      cls[k] = cn.cls.data();
      cnt[k] = cn.n_classes;
    }
  }
  if (small) n.n_classes = clade_classes(cls[0], cnt[0], cls[1], cnt[1], S, c.max_classes, n.cls, n.cmap);
  else n.n_classes = 0;
  const unsigned id = tips + (unsigned)c.nodes.size();
  c.nodes.push_back(std::move(n));
  c.intern.emplace(key, id);
  return id;
}

// the code arenas: rows of tip_stride() entries, grown geometrically (programs hold row
// OFFSETS).  wide = false: the 8-bit arena every 4-state partition has (d_tipcodes16);
// wide = true: its 16-bit twin for schedules with 64-row tables.
// Returns hipErrorOutOfMemory with p->code_arena_full set when `rows` rows cannot be addressed
// with 32-bit byte offsets (the caller then compiles without pseudo-tips); growth is geometric
// but never asks for more than the offsets can reach.
static hipError_t ensure_code_rows(rdamd_partition *p, unsigned rows, bool wide) {
  uint8_t *&arena = wide ? p->d_codes_wide : p->d_tipcodes16;
  unsigned &used = wide ? p->wide_rows : p->code_rows;
  unsigned &have = wide ? p->wide_rows_cap : p->code_rows_cap;
  if (rows <= have) return hipSuccess;
  const size_t stride = (size_t)p->tip_stride() * (wide ? 2 : 1);
  const size_t max_rows = (0xffffffffull - kTipcodePad) / stride;   // 32-bit offsets
  if (rows > max_rows) {
    p->code_arena_full = true;
    return hipErrorOutOfMemory;
  }
  const unsigned cap = (unsigned)std::min<size_t>(std::max<size_t>(rows, (size_t)have + have / 2 + 16), max_rows);
  hipError_t e = sync_streams(p);
  if (e != hipSuccess) return e;
  uint8_t *fresh = nullptr;
  e = hipMalloc(&fresh, (size_t)cap * stride + kTipcodePad);
  if (e != hipSuccess) return e;
  if (used) e = hipMemcpy(fresh, arena, (size_t)used * stride, hipMemcpyDeviceToDevice);
  if (e == hipSuccess)
    e = hipMemset(fresh + (size_t)used * stride, 0, (size_t)(cap - used) * stride + kTipcodePad);
  if (e != hipSuccess) { (void)hipFree(fresh); return e; }
  if (arena) (void)hipFree(arena);
  arena = fresh;
  have = cap;
  return hipSuccess;
}

static hipError_t upload_wide_row(rdamd_partition *p, unsigned row, const uint8_t *classes) {
  std::vector<uint16_t> wide(p->tip_stride(), 0);
  for (size_t s = 0; s < p->sites; ++s) wide[s] = (uint16_t)(classes[s] << 4);
  return hipMemcpy(p->d_codes_wide + (size_t)row * p->tip_stride() * 2, wide.data(), wide.size() * 2,
                   hipMemcpyHostToDevice);
}

// The 16-bit arena with every tip's row in place.  rdamd_set_tip_states drops the arena; it is
// rebuilt from the host copy here, by the next schedule that is compiled or the next batch that
// runs a 64-row schedule.  A failure half way leaves NO arena (never one with missing tip rows).
hipError_t ensure_wide_arena(rdamd_partition *p) {
  if (p->d_codes_wide && p->wide_rows >= p->tips) return hipSuccess;
  hipError_t e = ensure_code_rows(p, p->tips, true);
  for (unsigned t = 0; t < p->tips && e == hipSuccess; ++t)
    e = upload_wide_row(p, t, p->tipcodes.data() + (size_t)t * p->sites);
  if (e == hipSuccess) {
    p->wide_rows = p->tips;
  } else {
    if (p->d_codes_wide) (void)hipFree(p->d_codes_wide);
    p->d_codes_wide = nullptr;
    p->wide_rows = p->wide_rows_cap = 0;
  }
  return e;
}

hipError_t clade_upload_codes(rdamd_partition *p, unsigned id, bool wide) {
  CladeNode &n = p->clades->nodes[id - p->tips];
  if (n.code_row[wide] >= 0) return hipSuccess;
  hipError_t e = wide ? ensure_wide_arena(p) : hipSuccess;
  if (e != hipSuccess) return e;
  unsigned &used = wide ? p->wide_rows : p->code_rows;
  e = ensure_code_rows(p, used + 1, wide);
  if (e != hipSuccess) return e;
  if (wide) {
    e = upload_wide_row(p, used, n.cls.data());
  } else {
    std::vector<uint8_t> row(p->tip_stride(), 0);
    for (size_t s = 0; s < p->sites; ++s) row[s] = (uint8_t)(n.cls[s] << 4);   // the LDS row offset, as for tips
    e = hipMemcpy(p->d_tipcodes16 + (size_t)used * p->tip_stride(), row.data(), row.size(), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) return e;
  n.code_row[wide] = (int)used++;
  return hipSuccess;
}

hipError_t clade_upload_map(rdamd_partition *p, unsigned id) {
  CladeCache &c = *p->clades;
  CladeNode &n = c.nodes[id - p->tips];
  if (n.map_off >= 0) return hipSuccess;
  const size_t bytes = (n.cmap.size() + 3) & ~(size_t)3;
  if (c.maps_used + bytes > c.maps_cap) {
    const size_t cap = std::max<size_t>(c.maps_used + bytes, c.maps_cap * 2 + 4096);
    hipError_t e = sync_streams(p);
    if (e != hipSuccess) return e;
    uint8_t *fresh = nullptr;
    e = hipMalloc(&fresh, cap);
    if (e != hipSuccess) return e;
    if (c.maps_used) e = hipMemcpy(fresh, c.d_maps, c.maps_used, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { (void)hipFree(fresh); return e; }
    if (c.d_maps) (void)hipFree(c.d_maps);
    c.d_maps = fresh;
    c.maps_cap = cap;
  }
  hipError_t e = hipMemcpy(c.d_maps + c.maps_used, n.cmap.data(), n.cmap.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) return e;
  n.map_off = (long)c.maps_used;
  c.maps_used += bytes;
  return hipSuccess;
}

// ---- device: the tables of one launch -----------------------------------------------------
// One workgroup per (pseudo-tip, job); a thread owns (class, rate) pairs.  The nodes of the
// pseudo-tip's clade are walked in post-order (a barrier between nodes): the CLV of a class
// is the product of the two child rows the class map points at -- rows of the job's tip
// tables for tip children, rows this workgroup wrote a moment ago (global scratch) for nested
// clades -- and the row that is kept is P(branch above the node) . CLV, which is what the
// node's parent consumes.  The last node's rows go into the job's table slot of that branch
// (the slot its tip table would occupy if the child were a tip: a branch has one child), in
// the tip tables' layout, so the evaluator reads a pseudo-tip exactly as it reads a tip.
//
// Scaling: a pseudo-tip carries no rescale count.  That is exact as long as no class of any
// node of the clade meets the rescale condition (all entries < 2^-256);
// if one does -- or if a table entry is small enough for a later tip-tip product to meet it --
// the job's flag goes up and the evaluator runs the job's plain program instead.
// ROWS: rows per table slot of the launch (16 or 64) = the scratch tables' row count.  A
// pseudo-tip of up to 16 classes lands in its branch's 16-row table, in the tip tables'
// layout ([class][state]); one of up to 64 classes in its own 64-row table, stored as the
// evaluator's LDS slot image ([half][class][2 states]: the table goes to LDS by DMA).
// THREADS: 256 lanes per workgroup for 64-row tables when the launch has the device to itself;
// 64 when it runs beside the evaluator of the batch in front of it (pipelined batches), whose
// one-wave workgroups fill the device -- a four-wave workgroup waits until four slots of one
// CU are free at once, a wave takes the first that opens.
template <int ROWS, int THREADS>
__global__ void __launch_bounds__(THREADS)
clade_table_kernel(FusedJob *__restrict__ jobs, const uint8_t *__restrict__ maps,
                   const double *__restrict__ pmat, double *__restrict__ tiptab, size_t pmat_job_stride,
                   size_t tiptab_job_stride, double *__restrict__ scratch, size_t scratch_job_stride, unsigned R,
                   unsigned *__restrict__ any_unsafe) {
  const unsigned job = blockIdx.y, grp = blockIdx.x;
  const FusedJob jb = jobs[job];
  if (grp >= jb.n_groups) return;
  const CladeGroup g = jb.clade_groups[grp];
  const double *__restrict__ pm = pmat + (size_t)job * pmat_job_stride;        // [matrix][rate][16]
  double *__restrict__ tt = tiptab + (size_t)job * tiptab_job_stride;           // [matrix][rate][16 rows][4]
  double *__restrict__ wide = tt + pmat_job_stride * 4;                         // [slot][rate][2][64][2]
  double *__restrict__ sc = scratch + (size_t)job * scratch_job_stride;         // [step][rate][ROWS][4]
  bool unsafe = false;
  for (unsigned k = 0; k < g.count; ++k) {
    const CladeStep st = jb.clade_steps[g.first + k];
    const uint8_t *__restrict__ map = maps + st.map_off;
    for (unsigned idx = threadIdx.x; idx < st.n_classes * R; idx += blockDim.x) {
      const unsigned c = idx / R, r = idx % R;
      double v[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const unsigned cc = map[2 * c + ch];
        if (st.src[ch] & 0x80000000u) {
          const double *row = sc + (((size_t)(g.first + (st.src[ch] & 0x7fffffffu)) * R + r) * ROWS + cc) * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] *= row[i];
        } else {   // a tip's table (ROWS = 64: stored [half][code][2], kernels_fused.hip)
          const double *tab = tt + ((size_t)st.src[ch] * R + r) * 64;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] *= tab[ROWS > 16 ? (i >> 1) * 32 + cc * 2 + (i & 1) : cc * 4 + i];
        }
      }
      const double vmax = fmax(fmax(v[0], v[1]), fmax(v[2], v[3]));
      if (vmax > 0.0 && vmax < kScaleThreshold) unsafe = true;   // the reference rule would rescale here
      const double *__restrict__ m = pm + ((size_t)st.out_mat * R + r) * 16;
      double t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t[i] = m[i * 4 + 0] * v[0] + m[i * 4 + 1] * v[1] + m[i * 4 + 2] * v[2] + m[i * 4 + 3] * v[3];
        if (t[i] > 0.0 && t[i] < 0x1p-128) unsafe = true;
      }
      if (ROWS > 16 && st.last && st.wide_slot != 0xffffffffu) {
        double *out = wide + ((size_t)st.wide_slot * R + r) * (4 * ROWS) + c * 2;
        out[0] = t[0]; out[1] = t[1];
        out[2 * ROWS] = t[2]; out[2 * ROWS + 1] = t[3];
      } else {
        if (st.last && ROWS > 16) {   // a 16-row table in the DMA layout
          double *out = tt + ((size_t)st.out_mat * R + r) * 64 + c * 2;
          out[0] = t[0]; out[1] = t[1]; out[32] = t[2]; out[33] = t[3];
        } else {
          double *out = st.last ? tt + (((size_t)st.out_mat * R + r) * 16 + c) * 4
                                : sc + (((size_t)(g.first + k) * R + r) * ROWS + c) * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) out[i] = t[i];
        }
      }
    }
    __syncthreads();   // (also orders this workgroup's scratch writes before the next node's reads)
  }
  if (unsafe) {
    jobs[job].tt_unsafe = 1u;   // the job's flag (every writer stores the same value)
    *any_unsafe = 1u;           // ... and the batch's: the evaluator's second pass is needed
  }
}

hipError_t launch_clade_tables(const FusedArgs &a, const uint8_t *d_maps, double *d_scratch,
                               size_t scratch_job_stride, unsigned n_jobs, unsigned max_groups,
                               bool slim, hipStream_t stream) {
  if (!n_jobs || !max_groups) return hipSuccess;
  if (a.table_rows > 16 && slim)
    clade_table_kernel<64, 64><<<dim3(max_groups, n_jobs), 64, 0, stream>>>(
        const_cast<FusedJob *>(a.jobs), d_maps, a.pmat, const_cast<double *>(a.tiptab), a.pmat_job_stride,
        a.tiptab_job_stride, d_scratch, scratch_job_stride, a.rate_cats, a.any_unsafe);
  else if (a.table_rows > 16)
    clade_table_kernel<64, 256><<<dim3(max_groups, n_jobs), 256, 0, stream>>>(
        const_cast<FusedJob *>(a.jobs), d_maps, a.pmat, const_cast<double *>(a.tiptab), a.pmat_job_stride,
        a.tiptab_job_stride, d_scratch, scratch_job_stride, a.rate_cats, a.any_unsafe);
  else
    clade_table_kernel<16, 64><<<dim3(max_groups, n_jobs), 64, 0, stream>>>(
        const_cast<FusedJob *>(a.jobs), d_maps, a.pmat, const_cast<double *>(a.tiptab), a.pmat_job_stride,
        a.tiptab_job_stride, d_scratch, scratch_job_stride, a.rate_cats, a.any_unsafe);
  return hipGetLastError();
}

}  // namespace rdamd
