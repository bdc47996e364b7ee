// C ABI: partition life cycle, setters and the three hot-path entry points.
// Mirrors the coraxlib subset RootDigger calls; each
// function cites the call site it replaces in include/root_digger_amd.h.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include <dlfcn.h>

#include "clades.hpp"
#include "k20_split.hpp"
#include "common.hpp"

namespace rdamd {

static thread_local int  g_errno = 0;
static thread_local char g_errmsg[512] = "";

void set_error(int code, const char *fmt, ...) {
  g_errno = code;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_errmsg, sizeof(g_errmsg), fmt, ap);
  va_end(ap);
}
void clear_error() { g_errno = 0; g_errmsg[0] = 0; }

// Small host->device copies go through a pinned ring so they can be queued on
// the partition's stream without a blocking pageable copy.
static void *stage_alloc(rdamd_partition *p, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (p->stage_off + bytes > p->stage_bytes) {
    (void)hipStreamSynchronize(p->stream);
    p->stage_off = 0;
  }
  void *r = p->h_stage + p->stage_off;
  p->stage_off += bytes;
  return r;
}

// upload `bytes` from host memory to `dst` on the partition stream
static hipError_t upload(rdamd_partition *p, void *dst, const void *src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  if (bytes > p->stage_bytes / 2) {  // large: plain blocking copy
    hipError_t e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) return e;
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
  }
  void *st = stage_alloc(p, bytes);
  memcpy(st, src, bytes);
  p->stream_dirty = true;
  return hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, p->stream);
}

// device scratch carve-out (bump; reset per API call)
struct Scratch {
  rdamd_partition *p;
  size_t off = 0;
  void *take(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (off + bytes > p->scratch_bytes) return nullptr;
    void *r = (char *)p->d_scratch + off;
    off += bytes;
    return r;
  }
};

static hipError_t ensure_scratch(rdamd_partition *p, size_t bytes) {
  if (bytes <= p->scratch_bytes) return hipSuccess;
  hipError_t e = hipStreamSynchronize(p->stream);
  if (e != hipSuccess) return e;
  if (p->d_scratch) (void)hipFree(p->d_scratch);
  p->d_scratch = nullptr;
  p->scratch_bytes = 0;
  size_t want = std::max(bytes * 2, (size_t)1 << 20);
  e = hipMalloc(&p->d_scratch, want);
  if (e == hipSuccess) p->scratch_bytes = want;
  return e;
}

static hipError_t flush_q(rdamd_partition *p) {
  const unsigned K = p->states;
  std::vector<double> q((size_t)K * K);
  for (unsigned i = 0; i < p->rate_matrices; ++i) {
    if (!p->q_dirty[i]) continue;
    build_q_host(K, p->subst[i].data(), p->freqs[i].data(), q.data());
    hipError_t e = upload(p, p->d_q + (size_t)i * K * K, q.data(), sizeof(double) * K * K);
    if (e != hipSuccess) return e;
    e = upload(p, p->d_freqs + (size_t)i * K, p->freqs[i].data(), sizeof(double) * K);
    if (e != hipSuccess) return e;
    p->q_dirty[i] = 0;
  }
  return hipSuccess;
}

}  // namespace rdamd

hipEvent_t rdamd_partition::prof_begin(int kind) {
  stream_dirty = true;   // (every launch of the partition passes here)
  if (!profiling) return nullptr;
  hipEvent_t ev[2];
  for (auto &e : ev) {
    if (!prof_pool.empty()) { e = prof_pool.back(); prof_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) return nullptr;
  }
  (void)hipEventRecord(ev[0], stream);
  prof_spans.push_back({ev[0], ev[1], kind});
  return ev[1];
}
void rdamd_partition::prof_end() {
  if (!profiling || prof_spans.empty()) return;
  (void)hipEventRecord(prof_spans.back().b, stream);
}

using namespace rdamd;


namespace rdamd {
// one more slot of a sparse pool: a bigger block, the live slots copied over (everything queued
// on the partition has drained first -- kernels in flight hold the old addresses)
static hipError_t grow_pool(rdamd_partition *p, bool scalers) {
  unsigned &cap = scalers ? p->sc_slots_cap : p->clv_slots_cap;
  const unsigned used = scalers ? p->sc_slots_used : p->clv_slots_used;
  const unsigned limit = scalers ? p->scale_buffers : p->clv_buffers;
  const size_t slot_bytes = scalers ? (size_t)p->sites * sizeof(unsigned) : p->clv_doubles() * sizeof(double);
  const unsigned fresh_cap = std::min(limit, std::max(cap * 2u, 4u));
  if (fresh_cap <= cap) return hipErrorOutOfMemory;   // (cannot happen: a pool never holds more slots than indices)
  hipError_t e = sync_streams(p);
  if (e != hipSuccess) return e;
  void *fresh = nullptr;
  e = hipMalloc(&fresh, std::max<size_t>(8, (size_t)fresh_cap * slot_bytes));
  if (e != hipSuccess) return e;
  void *old = scalers ? (void *)p->d_scaler : (void *)p->d_clv;
  if (used && slot_bytes) e = hipMemcpy(fresh, old, (size_t)used * slot_bytes, hipMemcpyDeviceToDevice);
  if (e != hipSuccess) { (void)hipFree(fresh); return e; }
  (void)hipFree(old);
  if (scalers) p->d_scaler = (unsigned *)fresh; else p->d_clv = (double *)fresh;
  cap = fresh_cap;
  return hipSuccess;
}

hipError_t clv_phys(rdamd_partition *p, unsigned clv_index, unsigned *phys) {
  if (clv_index >= p->tips + p->clv_buffers) return hipErrorInvalidValue;
  *phys = clv_index;
  if (!p->sparse || clv_index < p->tips) return hipSuccess;
  int &slot = p->clv_slot[clv_index - p->tips];
  if (slot < 0) {
    if (p->clv_slots_used == p->clv_slots_cap) {
      hipError_t e = grow_pool(p, false);
      if (e != hipSuccess) return e;
    }
    slot = (int)p->clv_slots_used++;
  }
  *phys = p->tips + (unsigned)slot;
  return hipSuccess;
}

hipError_t scaler_phys(rdamd_partition *p, int scaler_index, int *phys) {
  if (scaler_index >= (int)p->scale_buffers) return hipErrorInvalidValue;
  *phys = scaler_index;
  if (!p->sparse || scaler_index < 0) return hipSuccess;
  int &slot = p->sc_slot[(size_t)scaler_index];
  if (slot < 0) {
    if (p->sc_slots_used == p->sc_slots_cap) {
      hipError_t e = grow_pool(p, true);
      if (e != hipSuccess) return e;
    }
    slot = (int)p->sc_slots_used++;
    // a scale buffer nobody has written reads as zeros, as in a dense partition
    p->stream_dirty = true;
    hipError_t e = hipMemsetAsync(p->d_scaler + (size_t)slot * p->sites, 0, (size_t)p->sites * sizeof(unsigned), p->stream);
    if (e != hipSuccess) return e;
  }
  *phys = slot;
  return hipSuccess;
}

hipError_t op_phys(rdamd_partition *p, const rdamd_operation_t &o, rdamd_operation_t *out) {
  *out = o;
  if (!p->sparse) return hipSuccess;
  if (o.parent_clv_index < p->tips) return hipErrorInvalidValue;
  hipError_t e = clv_phys(p, o.parent_clv_index, &out->parent_clv_index);
  if (e == hipSuccess) e = clv_phys(p, o.child1_clv_index, &out->child1_clv_index);
  if (e == hipSuccess) e = clv_phys(p, o.child2_clv_index, &out->child2_clv_index);
  if (e == hipSuccess) e = scaler_phys(p, o.parent_scaler_index, &out->parent_scaler_index);
  if (e == hipSuccess) e = scaler_phys(p, o.child1_scaler_index, &out->child1_scaler_index);
  if (e == hipSuccess) e = scaler_phys(p, o.child2_scaler_index, &out->child2_scaler_index);
  return e;
}
}  // namespace rdamd

extern "C" {

int rdamd_errno(void) { return g_errno; }
const char *rdamd_errmsg(void) { return g_errmsg; }
const char *rdamd_version(void) { return "root_digger_amd 0.1 (gfx950)"; }

// the libamdhip64 that serves this library (a process may hold a second one, e.g. the copy
// PyTorch bundles): device pointers are only good between code on the SAME runtime
const char *rdamd_hip_runtime_path(void) {
  static std::string path = [] {
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDevice), &info) && info.dli_fname) return std::string(info.dli_fname);
    return std::string();
  }();
  return path.c_str();
}

int rdamd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rdamd_set_device(int device) {
  clear_error();
  RDAMD_HIP_TRY(hipSetDevice(device), RDAMD_FAILURE);
  return RDAMD_SUCCESS;
}

int rdamd_device_memory(uint64_t *free_bytes, uint64_t *total_bytes) {
  clear_error();
  size_t f = 0, t = 0;
  RDAMD_HIP_TRY(hipMemGetInfo(&f, &t), RDAMD_FAILURE);
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return RDAMD_SUCCESS;
}

// device bytes rdamd_partition_create takes for a partition of this shape (the
// large buffers; what a replica of a model costs)
uint64_t rdamd_partition_footprint(unsigned int tips, unsigned int clv_buffers, unsigned int states,
                                   unsigned int sites, unsigned int prob_matrices,
                                   unsigned int rate_cats, unsigned int scale_buffers) {
  const uint64_t K = states == 2 ? 4 : states, R = rate_cats, S = sites;
  const uint64_t codes = K == 4 ? 16 : 64;
  const uint64_t Sclv = (K == 20 && R <= 8) ? (S + 15) / 16 * 16 : S;   // operand layout: whole 16-site tiles
  // (4 states: the tip codes twice -- as codes and as LDS row offsets for the fused evaluator)
  uint64_t b = (uint64_t)tips * ((S + 3) / 4 * 4) * (K == 4 ? 2 : 1) + (uint64_t)clv_buffers * Sclv * R * K * 8 +
               (uint64_t)scale_buffers * S * 4 + (uint64_t)prob_matrices * R * K * K * 8 +
               (uint64_t)prob_matrices * R * codes * K * 8 + S * 4 + ((uint64_t)6 << 20);
  if (K == 20 && R <= 8) b += (uint64_t)prob_matrices * R * k20_mfma_copy_doubles() * 8;
  return b;
}

#define NT(ch, v) [ch] = v, [ch + 32] = v
const uint64_t rdamd_map_nt[256] = {
    NT('A', 1),  NT('C', 2),  NT('G', 4),  NT('T', 8),  NT('U', 8),  NT('R', 5),
    NT('Y', 10), NT('S', 6),  NT('W', 9),  NT('K', 12), NT('M', 3),  NT('B', 14),
    NT('D', 13), NT('H', 11), NT('V', 7),  NT('N', 15), NT('O', 15), NT('X', 15),
    ['-'] = 15,  ['?'] = 15,
};
#undef NT
const uint64_t rdamd_map_bin[256] = {
    ['0'] = 1, ['1'] = 2, ['-'] = 3, ['?'] = 3,
};

rdamd_partition_t *rdamd_partition_create(unsigned int tips, unsigned int clv_buffers,
                                          unsigned int states, unsigned int sites,
                                          unsigned int rate_matrices,
                                          unsigned int prob_matrices,
                                          unsigned int rate_cats,
                                          unsigned int scale_buffers,
                                          unsigned int attributes) {
  clear_error();
  if (states < 2 || states > 64 || rate_cats < 1 || rate_matrices < 1 || tips < 1) {
    set_error(1, "rdamd_partition_create: unsupported sizes (states=%u rate_cats=%u "
                 "rate_matrices=%u tips=%u)", states, rate_cats, rate_matrices, tips);
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error(2, "rdamd_partition_create: no HIP device available; this library has "
                 "no CPU fallback");
    return nullptr;
  }
  rdamd_partition *p = new rdamd_partition();
  p->api_states = states;
  if (states == 2) states = 4;   // embedded, see common.hpp
  p->tips = tips; p->clv_buffers = clv_buffers; p->states = states; p->sites = sites;
  p->rate_matrices = rate_matrices; p->prob_matrices = prob_matrices;
  p->rate_cats = rate_cats; p->scale_buffers = scale_buffers; p->attributes = attributes;
  // (rdamd_partition_set_rescale_speculation's mode for partitions a caller does not reach itself --
  // model_t's, its replicas' --: RDAMD_RESCALE_SPECULATION=0|1 in the environment, read here)
  if (const char *e = getenv("RDAMD_RESCALE_SPECULATION")) {
    const int mode = atoi(e);
    p->rescale_speculation = mode <= 0 ? 0 : 1;
  }
  p->ncodes_cap = states == 4 ? 16 : 64;
  const unsigned K = states, R = rate_cats;
  const size_t S = sites;

#define TRY(expr) RDAMD_HIP_TRY(expr, (rdamd_partition_destroy(p), nullptr))
  TRY(hipGetDevice(&p->device));
  TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  // + kTipcodePad: the traversal kernel reads tip codes with dword-wide scalar
  // loads that may run a few bytes past the last row
  TRY(hipMalloc(&p->d_tipcodes, (size_t)tips * p->tip_stride() + kTipcodePad));
  if (K == 4) {
    TRY(hipMalloc(&p->d_tipcodes16, (size_t)tips * p->tip_stride() + kTipcodePad));
    p->code_rows = p->code_rows_cap = tips;
  }
  // (the 20-state matrix-core kernel keeps CLVs in its operand layout, whole 16-site tiles:
  // decided here, before the CLV buffers are sized -- common.hpp)
  p->mfma_layout = K == 20 && R <= 8 &&
                   (size_t)p->clv_tiles() * 16u * R * K * sizeof(double) < ((size_t)1 << 31) &&
                   (size_t)prob_matrices * R * k20_mfma_copy_doubles() * sizeof(double) < ((size_t)1 << 31);
  p->sparse = (attributes & RDAMD_ATTRIB_SPARSE_CLVS) != 0;
  if (p->sparse) {   // a small pool to start with (common.hpp); it grows when more buffers are live at once
    p->clv_slot.assign(clv_buffers, -1);
    p->sc_slot.assign(scale_buffers, -1);
    p->clv_slots_cap = std::min(clv_buffers, 4u);
    p->sc_slots_cap = std::min(scale_buffers, 4u);
  }
  const size_t clv_alloc = p->sparse ? p->clv_slots_cap : clv_buffers, sc_alloc = p->sparse ? p->sc_slots_cap : scale_buffers;
  TRY(hipMalloc(&p->d_clv, std::max<size_t>(8, clv_alloc * p->clv_doubles() * sizeof(double))));
  TRY(hipMalloc(&p->d_scaler, std::max<size_t>(4, sc_alloc * S * sizeof(unsigned))));
  TRY(hipMalloc(&p->d_pmat, (size_t)prob_matrices * R * K * K * sizeof(double)));
  TRY(hipMalloc(&p->d_tiptab, (size_t)prob_matrices * R * p->ncodes_cap * K * sizeof(double)));
  if (p->mfma_layout)
    TRY(hipMalloc(&p->d_pmat_mfma, (size_t)prob_matrices * R * k20_mfma_copy_doubles() * sizeof(double)));
  TRY(hipMalloc(&p->d_codemask, 256 * sizeof(uint64_t)));
  TRY(hipMalloc(&p->d_q, (size_t)rate_matrices * K * K * sizeof(double)));
  TRY(hipMalloc(&p->d_freqs, (size_t)rate_matrices * K * sizeof(double)));
  TRY(hipMalloc(&p->d_rates, R * sizeof(double)));
  TRY(hipMalloc(&p->d_rate_weights, R * sizeof(double)));
  TRY(hipMalloc(&p->d_pattern_weights, std::max<size_t>(4, S * sizeof(unsigned))));
  TRY(hipMalloc(&p->d_partials, 32768 * sizeof(double)));   // (8 root positions x 1024 blocks x 4 waves)
  TRY(hipMalloc(&p->d_counter, sizeof(unsigned)));
  TRY(hipMemsetAsync(p->d_counter, 0, sizeof(unsigned), p->stream));
  TRY(hipMalloc(&p->d_result, 64 * sizeof(double)));
  p->stage_bytes = (size_t)4 << 20;
  TRY(hipHostMalloc(&p->h_stage, p->stage_bytes, hipHostMallocDefault));
  TRY(hipHostMalloc(&p->h_result, 64 * sizeof(double), hipHostMallocDefault));
  TRY(ensure_scratch(p, (size_t)1 << 20));
  TRY(hipMemsetAsync(p->d_scaler, 0, std::max<size_t>(4, sc_alloc * S * sizeof(unsigned)), p->stream));
  TRY(hipMemsetAsync(p->d_tipcodes, 0, (size_t)tips * p->tip_stride() + kTipcodePad, p->stream));
  if (K == 4) TRY(hipMemsetAsync(p->d_tipcodes16, 0, (size_t)tips * p->tip_stride() + kTipcodePad, p->stream));

  // defaults as corax_partition_create leaves them: weights 1, rates 1, 1/R
  p->subst.assign(rate_matrices, std::vector<double>((size_t)K * K - K, 1.0));
  p->freqs.assign(rate_matrices, std::vector<double>(K, 1.0 / K));
  if (p->embedded()) {   // defaults of a 2-state partition, through the embedding setters
    p->api_subst.assign(rate_matrices, std::vector<double>(2, 1.0));
    p->api_freqs.assign(rate_matrices, std::vector<double>(2, 0.5));
    for (unsigned i = 0; i < rate_matrices; ++i) {
      p->subst[i].assign(12, 0.0);
      p->subst[i][0] = p->subst[i][3] = 1.0;
      p->freqs[i] = {0.5, 0.5, 0.0, 0.0};
    }
  }
  p->rates.assign(R, 1.0);
  p->rate_weights.assign(R, 1.0 / R);
  p->prop_invar.assign(rate_matrices, 0.0);
  p->pattern_weights.assign(S, 1u);
  p->tipcodes.assign((size_t)tips * S, 0);
  p->q_dirty.assign(rate_matrices, 1);
  p->codemask.assign(256, 0);
  if (K == 4) {
    for (unsigned c = 0; c < 16; ++c) p->codemask[c] = c;
    p->ncodes = 16;
  } else {
    p->ncodes = 0;
  }
  TRY(upload(p, p->d_codemask, p->codemask.data(), 256 * sizeof(uint64_t)));
  TRY(upload(p, p->d_rates, p->rates.data(), R * sizeof(double)));
  TRY(upload(p, p->d_rate_weights, p->rate_weights.data(), R * sizeof(double)));
  if (S) TRY(upload(p, p->d_pattern_weights, p->pattern_weights.data(), S * sizeof(unsigned)));
  TRY(hipStreamSynchronize(p->stream));
#undef TRY
  return p;
}

void rdamd_partition_destroy(rdamd_partition_t *p) {
  if (!p) return;
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  void *dev[] = {p->d_tipcodes, p->d_tipcodes16, p->d_codes_wide, p->d_clv, p->d_scaler, p->d_pmat, p->d_tiptab, p->d_pmat_mfma,
                 p->d_codemask, p->d_q, p->d_freqs, p->d_rates, p->d_rate_weights,
                 p->d_pattern_weights, p->d_tipclv_scratch, p->d_scratch,
                 p->d_partials, p->d_result, p->d_persite, p->d_counter};
  for (void *d : dev)
    if (d) (void)hipFree(d);
  rdamd::fused_workspace_free(p->fused);
  rdamd::fused_workspace_free(p->fused1);
  for (auto &b : p->sched_pool) (void)hipFree(b.ptr);
  for (auto &b : p->sched_retired) (void)hipFree(b.ptr);
  if (p->stream_pre) (void)hipStreamDestroy(p->stream_pre);
  rdamd::clade_cache_free(p->clades);
  for (auto &sp : p->prof_spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  for (auto &e : p->prof_pool) (void)hipEventDestroy(e);
  if (p->h_stage) (void)hipHostFree(p->h_stage);
  if (p->h_result) (void)hipHostFree(p->h_result);
  if (p->d_root_items) (void)hipFree(p->d_root_items);
  if (p->h_root_items) (void)hipHostFree(p->h_root_items);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

void rdamd_partition_discard_clvs(rdamd_partition_t *p) {
  if (!p || !p->sparse) return;
  // (slots are reused by later calls on the partition's stream, i.e. after everything queued so
  // far; a combined root step of ANOTHER partition's stream has returned before its caller can
  // get here)
  std::fill(p->clv_slot.begin(), p->clv_slot.end(), -1);
  std::fill(p->sc_slot.begin(), p->sc_slot.end(), -1);
  p->clv_slots_used = p->sc_slots_used = 0;
}

unsigned int rdamd_update_clvs_launches(const rdamd_partition_t *p) { return p->last_clv_launches; }

uint64_t rdamd_partition_clv_bytes(const rdamd_partition_t *p) {
  const uint64_t c = p->sparse ? p->clv_slots_cap : p->clv_buffers, s = p->sparse ? p->sc_slots_cap : p->scale_buffers;
  return c * p->clv_doubles() * sizeof(double) + s * (uint64_t)p->sites * sizeof(unsigned);
}

int rdamd_set_tip_states(rdamd_partition_t *p, unsigned int tip_index,
                         const uint64_t *map, const char *sequence) {
  clear_error();
  if (tip_index >= p->tips) {
    set_error(3, "rdamd_set_tip_states: tip index %u out of range", tip_index);
    return RDAMD_FAILURE;
  }
  const size_t S = p->sites;
  uint8_t *row = p->tipcodes.data() + (size_t)tip_index * S;
  const uint64_t full = p->api_states == 64 ? ~0ull : ((1ull << p->api_states) - 1);
  bool new_code = false;
  for (size_t s = 0; s < S; ++s) {
    uint64_t st = map[(unsigned char)sequence[s]];
    if (!st || (st & ~full)) {
      set_error(4, "rdamd_set_tip_states: character '%c' (site %zu) has no valid state "
                   "in the map", sequence[s], s);
      return RDAMD_FAILURE;
    }
    if (p->states == 4) {
      row[s] = (uint8_t)st;
    } else {
      unsigned c = 0;
      for (; c < p->ncodes; ++c)
        if (p->codemask[c] == st) break;
      if (c == p->ncodes) {
        if (p->ncodes >= p->ncodes_cap) {
          set_error(5, "rdamd_set_tip_states: more than %u distinct state codes",
                    p->ncodes_cap);
          return RDAMD_FAILURE;
        }
        p->codemask[p->ncodes++] = st;
        new_code = true;
      }
      row[s] = (uint8_t)c;
    }
  }
  if (new_code) {
    RDAMD_HIP_TRY(upload(p, p->d_codemask, p->codemask.data(), 256 * sizeof(uint64_t)),
                  RDAMD_FAILURE);
    p->tiptab_stale = true;
  }
  // class codes of subtrees were worked out from the old characters: forget them (schedules
  // compiled with pseudo-tips notice through the generation count)
  p->tip_generation += 1;
  if (p->clades) {
    RDAMD_HIP_TRY(sync_streams(p), RDAMD_FAILURE);
    const unsigned keep = p->clades->max_classes;
    rdamd::clade_cache_free(p->clades);
    p->clades = new rdamd::CladeCache();
    p->clades->max_classes = keep;
    p->code_rows = p->tips;
  }
  if (p->d_codes_wide) {   // (rebuilt from the host copy when the next wide schedule is compiled or run)
    RDAMD_HIP_TRY(sync_streams(p), RDAMD_FAILURE);
    (void)hipFree(p->d_codes_wide);
    p->d_codes_wide = nullptr;
    p->wide_rows = p->wide_rows_cap = 0;
  }
  RDAMD_HIP_TRY(upload(p, p->d_tipcodes + (size_t)tip_index * p->tip_stride(), row, S), RDAMD_FAILURE);
  if (p->d_tipcodes16) {   // the same row as LDS row offsets (kernels_fused.hip)
    std::vector<uint8_t> row16(S);
    for (size_t s = 0; s < S; ++s) row16[s] = (uint8_t)(row[s] << 4);
    RDAMD_HIP_TRY(upload(p, p->d_tipcodes16 + (size_t)tip_index * p->tip_stride(), row16.data(), S), RDAMD_FAILURE);
  }
  return RDAMD_SUCCESS;
}

void rdamd_set_pattern_weights(rdamd_partition_t *p, const unsigned int *w) {
  p->pattern_weights.assign(w, w + p->sites);
  (void)upload(p, p->d_pattern_weights, w, sizeof(unsigned) * p->sites);
}

void rdamd_set_subst_params(rdamd_partition_t *p, unsigned int idx, const double *v) {
  if (idx >= p->rate_matrices) return;
  if (p->embedded()) {   // (q01, q10) into the row-major off-diagonals of the 4x4 matrix
    p->api_subst[idx].assign(v, v + 2);
    p->subst[idx].assign(12, 0.0);
    p->subst[idx][0] = v[0];
    p->subst[idx][3] = v[1];
  } else {
    p->subst[idx].assign(v, v + (size_t)p->states * p->states - p->states);
  }
  p->q_dirty[idx] = 1;
}

void rdamd_set_frequencies(rdamd_partition_t *p, unsigned int idx, const double *f) {
  if (idx >= p->rate_matrices) return;
  if (p->embedded()) {
    p->api_freqs[idx].assign(f, f + 2);
    p->freqs[idx] = {f[0], f[1], 0.0, 0.0};
  } else {
    p->freqs[idx].assign(f, f + p->states);
  }
  p->q_dirty[idx] = 1;
}

void rdamd_set_category_rates(rdamd_partition_t *p, const double *r) {
  p->rates.assign(r, r + p->rate_cats);
  (void)upload(p, p->d_rates, r, sizeof(double) * p->rate_cats);
}

void rdamd_set_category_weights(rdamd_partition_t *p, const double *w) {
  p->rate_weights.assign(w, w + p->rate_cats);
  (void)upload(p, p->d_rate_weights, w, sizeof(double) * p->rate_cats);
}

int rdamd_update_invariant_sites_proportion(rdamd_partition_t *p, unsigned int idx,
                                            double prop) {
  clear_error();
  if (idx >= p->rate_matrices || prop != 0.0) {
    set_error(6, "rdamd_update_invariant_sites_proportion: only 0.0 is supported (the "
                 "reference never sets anything else, src/model.cpp:292-300)");
    return RDAMD_FAILURE;
  }
  p->prop_invar[idx] = 0.0;
  return RDAMD_SUCCESS;
}

double *rdamd_msa_empirical_frequencies(rdamd_partition_t *p) {
  const unsigned K = p->api_states;
  double *f = (double *)calloc(K, sizeof(double));
  if (!f) return nullptr;
  double total = 0.0;
  for (unsigned s = 0; s < p->sites; ++s) total += p->pattern_weights[s];
  for (unsigned t = 0; t < p->tips; ++t) {
    const uint8_t *row = p->tipcodes.data() + (size_t)t * p->sites;
    for (unsigned s = 0; s < p->sites; ++s) {
      uint64_t mask = p->codemask[row[s]];
      double cnt = (double)__builtin_popcountll(mask);
      if (cnt == 0.0) continue;  // tip never set
      for (unsigned j = 0; j < K; ++j)
        if ((mask >> j) & 1) f[j] += p->pattern_weights[s] * 1.0 / cnt;
    }
  }
  for (unsigned j = 0; j < K; ++j) f[j] /= total * p->tips;
  return f;
}

unsigned int rdamd_partition_states(const rdamd_partition_t *p) { return p->api_states; }
unsigned int rdamd_partition_rate_cats(const rdamd_partition_t *p) { return p->rate_cats; }
unsigned int rdamd_partition_sites(const rdamd_partition_t *p) { return p->sites; }
unsigned int rdamd_partition_tips(const rdamd_partition_t *p) { return p->tips; }
double rdamd_partition_weight_sum(const rdamd_partition_t *p) {
  double total = 0.0;
  for (unsigned s = 0; s < p->sites; ++s) total += p->pattern_weights[s];
  return total;
}
void *rdamd_partition_stream(const rdamd_partition_t *p) {
  const_cast<rdamd_partition_t *>(p)->stream_external = true;
  return (void *)p->stream;
}

int rdamd_partition_set_stream_priority(rdamd_partition_t *p, int level) {
  clear_error();
  std::lock_guard<std::mutex> guard(p->launch_mu);
  RDAMD_HIP_TRY(sync_streams(p), RDAMD_FAILURE);
  int least = 0, greatest = 0;   // (numerically: greatest priority = lowest number)
  RDAMD_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest), RDAMD_FAILURE);
  const int prio = level < 0 ? greatest : (level > 0 ? least : (least + greatest) / 2);
  if ((level < 0 ? -1 : (level > 0 ? 1 : 0)) == p->stream_priority && p->stream) return RDAMD_SUCCESS;   // (nothing to re-create)
  hipStream_t fresh = nullptr;
  RDAMD_HIP_TRY(hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, prio), RDAMD_FAILURE);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  p->stream = fresh;
  p->stream_priority = level < 0 ? -1 : (level > 0 ? 1 : 0);
  return RDAMD_SUCCESS;
}
const double *rdamd_partition_subst_params(const rdamd_partition_t *p, unsigned int i) {
  if (i >= p->rate_matrices) return nullptr;
  return p->embedded() ? p->api_subst[i].data() : p->subst[i].data();
}
const double *rdamd_partition_frequencies(const rdamd_partition_t *p, unsigned int i) {
  if (i >= p->rate_matrices) return nullptr;
  return p->embedded() ? p->api_freqs[i].data() : p->freqs[i].data();
}

int rdamd_update_prob_matrices(rdamd_partition_t *p, const unsigned int *params_indices,
                               const unsigned int *matrix_indices,
                               const double *branch_lengths, unsigned int count) {
  clear_error();
  if (count == 0) return RDAMD_SUCCESS;
  const unsigned R = p->rate_cats;
  for (unsigned r = 0; r < R; ++r)
    if (params_indices[r] >= p->rate_matrices) {
      set_error(7, "rdamd_update_prob_matrices: params index %u out of range",
                params_indices[r]);
      return RDAMD_FAILURE;
    }
  for (unsigned m = 0; m < count; ++m) {
    if (matrix_indices[m] >= p->prob_matrices) {
      set_error(8, "rdamd_update_prob_matrices: matrix index %u out of range",
                matrix_indices[m]);
      return RDAMD_FAILURE;
    }
    if (!(branch_lengths[m] >= 0.0) || !std::isfinite(branch_lengths[m])) {
      set_error(9, "rdamd_update_prob_matrices: invalid branch length %g for matrix %u",
                branch_lengths[m], matrix_indices[m]);
      return RDAMD_FAILURE;
    }
  }
  RDAMD_HIP_TRY(flush_q(p), RDAMD_FAILURE);
  // the three small input arrays travel as ONE host-to-device copy: lengths
  // first (8-byte aligned), then the two index arrays
  const size_t bl_bytes = sizeof(double) * count, mi_bytes = sizeof(unsigned) * count,
               pi_bytes = sizeof(unsigned) * R, packed = bl_bytes + mi_bytes + pi_bytes;
  RDAMD_HIP_TRY(ensure_scratch(p, packed + 512), RDAMD_FAILURE);
  Scratch sc{p};
  char *d_block = (char *)sc.take(packed);
  double *d_bl = (double *)d_block;
  unsigned *d_mi = (unsigned *)(d_block + bl_bytes);
  unsigned *d_pi = (unsigned *)(d_block + bl_bytes + mi_bytes);
  {
    std::vector<char> host(packed);
    memcpy(host.data(), branch_lengths, bl_bytes);
    memcpy(host.data() + bl_bytes, matrix_indices, mi_bytes);
    memcpy(host.data() + bl_bytes + mi_bytes, params_indices, pi_bytes);
    RDAMD_HIP_TRY(upload(p, d_block, host.data(), packed), RDAMD_FAILURE);
  }
  p->prof_begin(1);
  hipError_t le = launch_pmatrix(p, d_pi, d_mi, d_bl, count);
  if (le == hipSuccess && p->d_pmat_mfma) le = launch_pmat_to_mfma(p, d_mi, count);
  p->prof_end();
  RDAMD_HIP_TRY(le, RDAMD_FAILURE);
  return RDAMD_SUCCESS;
}

void rdamd_update_clvs(rdamd_partition_t *p, const rdamd_operation_t *ops,
                       unsigned int count) {
  clear_error();
  if (count == 0) return;
  const unsigned nclv = p->tips + p->clv_buffers;
  std::vector<rdamd_operation_t> phys_ops;
  if (p->sparse) {   // from here on the list names pool slots (common.hpp)
    phys_ops.resize(count);
    for (unsigned i = 0; i < count; ++i) {
      const hipError_t pe = op_phys(p, ops[i], &phys_ops[i]);
      if (pe != hipSuccess) {
        set_error(pe == hipErrorInvalidValue ? 10 : 100 + (int)pe, "rdamd_update_clvs: operation %u: %s", i,
                  pe == hipErrorInvalidValue ? "an index out of range" : hipGetErrorString(pe));
        return;
      }
    }
    ops = phys_ops.data();
  }
  // Independent subtrees side by side, level by level (k20_split.hpp, list_levels): the 20-state
  // kernel always (kernels_clv_mfma.hip), the 4-state one where one row of blocks leaves the device's
  // wave slots empty (kernels_clv.hip).
  ListLevels lv;
  {
    unsigned rows = 0, small = 8, min_count = 16;
    if (p->mfma_layout) clv_k20_traversal_cut(p, count, &rows, &small, &min_count);
    else rows = clv_traversal_pieces(p, count);
#ifdef RDAMD_ABLATION
    if (getenv("RDAMD_CLV_PIECE_OPS")) small = (unsigned)atoi(getenv("RDAMD_CLV_PIECE_OPS"));
    if (getenv("RDAMD_CLV_MIN_SPLIT")) min_count = (unsigned)atoi(getenv("RDAMD_CLV_MIN_SPLIT"));
#endif
    if (rows >= 2) list_levels(p->tips, p->clv_buffers, ops, count, rows, small, min_count, lv, p->mfma_layout);
  }
  if (!lv.order.empty()) ops = lv.order.data();
  std::vector<LevelOp> lops(count);
  for (unsigned i = 0; i < count; ++i) {
    const rdamd_operation_t &o = ops[i];
    if (o.parent_clv_index < p->tips || o.parent_clv_index >= nclv ||
        o.child1_clv_index >= nclv || o.child2_clv_index >= nclv ||
        o.child1_matrix_index >= p->prob_matrices ||
        o.child2_matrix_index >= p->prob_matrices ||
        o.parent_scaler_index >= (int)p->scale_buffers ||
        o.child1_scaler_index >= (int)p->scale_buffers ||
        o.child2_scaler_index >= (int)p->scale_buffers) {
      set_error(10, "rdamd_update_clvs: operation %u has an index out of range", i);
      return;
    }
    LevelOp &d = lops[i];
    d.parent_clv = o.parent_clv_index; d.child1_clv = o.child1_clv_index;
    d.child2_clv = o.child2_clv_index; d.child1_mat = o.child1_matrix_index;
    d.child2_mat = o.child2_matrix_index; d.parent_sc = o.parent_scaler_index;
    d.child1_sc = o.child1_scaler_index; d.child2_sc = o.child2_scaler_index;
    d.src1 = o.child1_clv_index < p->tips ? 0u : 1u;
    d.src2 = o.child2_clv_index < p->tips ? 0u : 1u;
    d.park = d.noop = 0;
    const uint64_t clv_bytes = (uint64_t)p->clv_doubles() * sizeof(double);
    auto sc_off = [&](int scb) {
      return scb >= 0 ? (uint64_t)scb * p->sites * sizeof(unsigned) : kNoOffset;
    };
    auto child_off = [&](unsigned clv) {
      return clv < p->tips ? (uint64_t)clv * p->tip_stride() : (uint64_t)(clv - p->tips) * clv_bytes;
    };
    d.parent_off = (uint64_t)(d.parent_clv - p->tips) * clv_bytes;
    d.parent_sc_off = sc_off(d.parent_sc);
    d.child1_off = child_off(d.child1_clv); d.child1_sc_off = sc_off(d.child1_sc);
    d.child2_off = child_off(d.child2_clv); d.child2_sc_off = sc_off(d.child2_sc);
  }
  // Where does each inner child come from?  The parent of the operation just
  // before stays in the lane's registers; an older sibling waits in one of the
  // kernel's LDS parking slots when one is free (shortest wait wins: when the
  // slots are full the value needed furthest in the future gives its slot up
  // and is read back from HBM instead -- every CLV is written there anyway).
  // A child is forwarded only when its scaler index is the producer's.
  // (`cuts` splits a list into segments, each analysed on its own: the pieces of a split list, which
  // run side by side with the slots their row count leaves them, then the operations that join them;
  // the 4-state kernel needs no other cut -- memory children are read at use, after every earlier
  // store of the lane.)
  const unsigned slots_whole = clv_traversal_slots(p);
  // segment boundaries, and the first segment of every launch (segments [levels[l], levels[l + 1])
  // run side by side)
  std::vector<unsigned> cuts{0u}, levels{0u};
  if (!lv.order.empty()) {
    cuts.assign(lv.seg.begin(), lv.seg.end() - 1);
    levels.assign(lv.level.begin(), lv.level.end() - 1);
  }
  // The 20-state kernel requests the operands of operation i+1 a whole
  // operation ahead and stores the result of operation i one operation late:
  // operation i may not read from memory what i-1 or i-2 wrote.  Their parents
  // are forwarded in registers (sources 2 and 3 below) -- except where the
  // value cannot be forwarded (same CLV under another scaler index, or the
  // other way round, or both earlier operations wrote it): there the list is
  // cut into two launches.
  // (No piece of a cut list has such a place -- list_levels has checked --; the list that is left
  // over, or the whole list, may.)
  const bool use_k20 = p->mfma_layout;   // fixed at creation, with the CLV layout
  if (use_k20)
    for (unsigned i = cuts.back() + 1; i < count; ++i)
      if (k20_hazard(p->tips, ops, i, cuts.back())) {
        cuts.push_back(i);
        levels.push_back((unsigned)cuts.size() - 1);
      }
  cuts.push_back(count);
  levels.push_back((unsigned)cuts.size() - 1);
  std::vector<int> producer(nclv, -1), consumer(count), which(count);
  // one segment with `nslots` parking slots; returns the number of children it reads back from memory
  // (producer: clv -> op of this segment that wrote it; all -1 between calls)
  auto analyse = [&](unsigned lo, unsigned hi, unsigned nslots) {
    for (unsigned i = lo; i < hi; ++i) {
      consumer[i] = -1;   // op -> first later op reading its parent
      lops[i].src1 = ops[i].child1_clv_index < p->tips ? 0u : 1u;
      lops[i].src2 = ops[i].child2_clv_index < p->tips ? 0u : 1u;
      if (!use_k20) lops[i].park = 0;
    }
    for (unsigned i = lo; i < hi; ++i) {
      const rdamd_operation_t &o = ops[i];
      const unsigned ch[2] = {o.child1_clv_index, o.child2_clv_index};
      const int chsc[2] = {o.child1_scaler_index, o.child2_scaler_index};
      for (int c = 0; c < 2; ++c) {
        if (ch[c] < p->tips || (!use_k20 && c == 1 && ch[1] == ch[0])) continue;
        const int j = producer[ch[c]];
        if (use_k20) {
          // the 20-state kernel keeps the results of the last TWO operations in
          // registers and forwards them to every reader (source 2: the operation
          // just before, 3: the one before that)
          if (j >= 0 && (int)i - j <= 2 && ops[j].parent_scaler_index == chsc[c])
            (c ? lops[i].src2 : lops[i].src1) = (int)i - j == 1 ? 2u : 3u;
          continue;
        }
        if (j >= 0 && consumer[j] < 0 && ops[j].parent_scaler_index == chsc[c]) {
          consumer[j] = (int)i;
          which[j] = c;
        }
      }
      producer[o.parent_clv_index] = (int)i;
    }
    std::vector<int> slot_owner(nslots, -1);
    auto set_src = [&](int j, unsigned kind) {
      LevelOp &c = lops[consumer[j]];
      (which[j] ? c.src2 : c.src1) = kind;
    };
    for (unsigned i = lo; i < hi; ++i) {
      for (unsigned sl = 0; sl < nslots; ++sl)      // slots whose value is consumed now
        if (slot_owner[sl] >= 0 && consumer[slot_owner[sl]] == (int)i) slot_owner[sl] = -1;
      const int c = consumer[i];
      if (c < 0) continue;
      // (consumer[i] was taken from producer[] at the time the consumer was
      // scanned, i.e. op i is the LAST writer of that CLV before it: nothing
      // in between can have overwritten the value)
      if (c == (int)i + 1) {
        set_src((int)i, 2u);
        // the same CLV as both children: both come from the registers
        if (ops[c].child1_clv_index == ops[c].child2_clv_index &&
            ops[c].child1_scaler_index == ops[c].child2_scaler_index)
          lops[c].src1 = lops[c].src2 = 2u;
        continue;
      }
      if (nslots == 0) continue;
      int take = -1, far = -1;
      for (unsigned sl = 0; sl < nslots; ++sl) {
        if (slot_owner[sl] < 0) { take = (int)sl; far = -1; break; }
        if (far < 0 || consumer[slot_owner[sl]] > consumer[slot_owner[far]]) far = (int)sl;
      }
      if (take < 0 && far >= 0 && consumer[slot_owner[far]] > c) {
        const int ev = slot_owner[far];             // give the slot to the shorter wait
        set_src(ev, 1u);
        lops[ev].park = 0;
        take = far;
      }
      if (take >= 0) {
        slot_owner[take] = (int)i;
        lops[i].park = 1u + (unsigned)take;
        set_src((int)i, 3u + (unsigned)take);
      }
    }
    unsigned readbacks = 0;
    for (unsigned i = lo; i < hi; ++i) {
      if (consumer[i] >= 0 && (which[i] ? lops[consumer[i]].src2 : lops[consumer[i]].src1) == 1u) ++readbacks;
      producer[ops[i].parent_clv_index] = -1;
    }
    return readbacks;
  };
  // The pieces of a launch share one slot count: the fewest slots that leave no more read-backs than
  // the whole-list count would, plus 6 in 100 operations (every slot less is LDS for another resident
  // block, a read-back is one exposed round trip of one piece; measured, profiles/r5_clv_pieces_ab.txt:
  // c2 in 8 pieces 162 / 172 / 189 us with 1 / 2 / 3 slots and 2 / 0 / 0 read-backs; c5's shard in
  // 32 pieces 1.67 / 1.60 / 1.73 ms with 74 / 31 / 13).
  std::vector<unsigned> seg_slots(cuts.size() - 1, slots_whole);
  for (size_t l = 0; !use_k20 && l + 1 < levels.size(); ++l) {
    const unsigned s0 = levels[l], s1 = levels[l + 1];
    if (s1 - s0 < 2) continue;
    auto level_readbacks = [&](unsigned nslots) {
      unsigned n = 0;
      for (unsigned seg = s0; seg < s1; ++seg) n += analyse(cuts[seg], cuts[seg + 1], nslots);
      return n;
    };
    unsigned tolerance = 6;
#ifdef RDAMD_ABLATION
    if (getenv("RDAMD_CLV_READBACK_PCT")) tolerance = (unsigned)atoi(getenv("RDAMD_CLV_READBACK_PCT"));
#endif
    const unsigned allowed = level_readbacks(slots_whole) + (cuts[s1] - cuts[s0]) * tolerance / 100;
    unsigned chosen = slots_whole;
    while (chosen > 0 && level_readbacks(chosen - 1) <= allowed) --chosen;
#ifdef RDAMD_ABLATION
    if (getenv("RDAMD_CLV_PIECE_SLOTS")) chosen = std::min<unsigned>((unsigned)atoi(getenv("RDAMD_CLV_PIECE_SLOTS")), 6u);
    if (getenv("RDAMD_CLV_DEBUG")) {
      fprintf(stderr, "[clv pieces] launch %zu: %u operations in %u pieces (longest %u); read-backs by slots:", l,
              cuts[s1] - cuts[s0], s1 - s0, cuts[s0 + 1] - cuts[s0]);
      for (unsigned sl = 0; sl <= slots_whole; ++sl) fprintf(stderr, " %u:%u", sl, level_readbacks(sl));
      fprintf(stderr, "; chosen %u\n", chosen);
    }
#endif
    for (unsigned seg = s0; seg < s1; ++seg) seg_slots[seg] = chosen;
  }
#ifdef RDAMD_ABLATION
  if (getenv("RDAMD_CLV_DEBUG") && !use_k20 && levels.size() > 2)
    fprintf(stderr, "[clv pieces] last launch: %u operations\n", count - cuts[levels[levels.size() - 2]]);
#endif
  for (size_t seg = 0; seg + 1 < cuts.size(); ++seg) analyse(cuts[seg], cuts[seg + 1], seg_slots[seg]);
  // Each segment (normally the whole list) runs as one launch in the caller's
  // order: every dependency is site-local, so the kernel needs no level
  // structure (kernels_clv.hip).
  if (!use_k20) {   // every segment padded to whole chunks with no-ops (a copy of its last op, stores off)
    const unsigned chunk = clv_traversal_chunk(p);
    std::vector<LevelOp> padded_ops;
    std::vector<unsigned> padded_cuts{0u};
    padded_ops.reserve(count + cuts.size() * chunk + 1);
    LevelOp pad{};
    for (size_t seg = 0; seg + 1 < cuts.size(); ++seg) {
      padded_ops.insert(padded_ops.end(), lops.begin() + cuts[seg], lops.begin() + cuts[seg + 1]);
      pad = lops[cuts[seg + 1] - 1];
      pad.src1 = pad.src2 = 2u;
      pad.park = 0;
      pad.noop = 1;
      while ((padded_ops.size() - padded_cuts.back()) % chunk) padded_ops.push_back(pad);
      padded_cuts.push_back((unsigned)padded_ops.size());
    }
    padded_ops.push_back(pad);   // terminator: the kernel looks one operation ahead (a segment's last
                                 // operation looks at the next segment's first: tip codes it never uses)
    lops.swap(padded_ops);
    cuts.swap(padded_cuts);
  }
  if (use_k20)   // tip-code look-ahead of the 20-state kernel (LevelOp::ahead*)
    for (unsigned i = 0; i < count; ++i) {
      const bool more = i + 1 < count;
      lops[i].ahead1 = more && lops[i + 1].src1 == 0u ? lops[i + 1].child1_clv : 0u;
      lops[i].ahead2 = more && lops[i + 1].src2 == 0u ? lops[i + 1].child2_clv : 0u;
    }
  const size_t padded = lops.size();
  hipError_t e = ensure_scratch(p, sizeof(LevelOp) * padded + 256);
  if (e == hipSuccess && p->tiptab_stale) {
    e = (p->stream_dirty = true, launch_tiptab_all(p));
    p->tiptab_stale = false;
  }
  Scratch sc{p};
  if (e == hipSuccess) {
    LevelOp *d_ops = (LevelOp *)sc.take(sizeof(LevelOp) * padded);
    e = upload(p, d_ops, lops.data(), sizeof(LevelOp) * padded);
    p->prof_begin(0);
    // one launch per level of the cut: its pieces side by side
    for (size_t l = 0; e == hipSuccess && l + 1 < levels.size(); ++l) {
      ListPieces pc;
      for (unsigned seg = levels[l]; seg < levels[l + 1]; ++seg) {
        pc.start[pc.n] = cuts[seg];
        pc.len[pc.n++] = cuts[seg + 1] - cuts[seg];
      }
      e = use_k20 ? launch_clv_k20_traversal(p, d_ops, pc) : launch_clv_traversal(p, d_ops, pc, seg_slots[levels[l]]);
    }
    p->last_clv_launches = (unsigned)levels.size() - 1;
    p->prof_end();
  }
  if (e != hipSuccess)
    set_error(100 + (int)e, "rdamd_update_clvs: %s", hipGetErrorString(e));
}

double rdamd_compute_root_loglikelihood(rdamd_partition_t *p, unsigned int clv_index,
                                        int scaler_index,
                                        const unsigned int *freqs_indices,
                                        double *persite_lnl) {
  clear_error();
  const double nan = std::nan("");
  if (clv_index < p->tips || clv_index >= p->tips + p->clv_buffers ||
      scaler_index >= (int)p->scale_buffers) {
    set_error(11, "rdamd_compute_root_loglikelihood: index out of range (clv %u, scaler %d)",
              clv_index, scaler_index);
    return nan;
  }
  for (unsigned r = 0; r < p->rate_cats; ++r)
    if (freqs_indices[r] >= p->rate_matrices) {
      set_error(7, "rdamd_compute_root_loglikelihood: freqs index out of range");
      return nan;
    }
  if (p->sites == 0) return 0.0;   // an empty alignment has likelihood 1
  RDAMD_HIP_TRY(clv_phys(p, clv_index, &clv_index), nan);
  RDAMD_HIP_TRY(scaler_phys(p, scaler_index, &scaler_index), nan);
  RDAMD_HIP_TRY(flush_q(p), nan);
  RDAMD_HIP_TRY(ensure_scratch(p, 1024 + sizeof(unsigned) * p->rate_cats), nan);
  Scratch sc{p};
  unsigned *d_fi = (unsigned *)sc.take(sizeof(unsigned) * p->rate_cats);
  RDAMD_HIP_TRY(upload(p, d_fi, freqs_indices, sizeof(unsigned) * p->rate_cats), nan);
  if (persite_lnl && !p->d_persite)
    RDAMD_HIP_TRY(hipMalloc(&p->d_persite, std::max<size_t>(8, sizeof(double) * p->sites)), nan);
  p->prof_begin(2);
  // the finishing workgroup writes straight into the pinned host block
  hipError_t le = launch_root_lnl(p, clv_index, scaler_index, d_fi,
                                  persite_lnl ? p->d_persite : nullptr, p->h_result);
  p->prof_end();
  RDAMD_HIP_TRY(le, nan);
  RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), nan);
  if (persite_lnl)
    RDAMD_HIP_TRY(hipMemcpy(persite_lnl, p->d_persite, sizeof(double) * p->sites,
                            hipMemcpyDeviceToHost), nan);
  return p->h_result[0];
}

int rdamd_compute_root_loglikelihoods(rdamd_partition_t *p, unsigned int count,
                                      const unsigned int *clv_indices, const int *scaler_indices,
                                      const unsigned int *freqs_indices, double *lnl_out) {
  clear_error();
  if (count == 0) return RDAMD_SUCCESS;
  std::vector<unsigned> rel(count);
  for (unsigned i = 0; i < count; ++i) {
    if (clv_indices[i] < p->tips || clv_indices[i] >= p->tips + p->clv_buffers ||
        scaler_indices[i] >= (int)p->scale_buffers) {
      set_error(11, "rdamd_compute_root_loglikelihoods: index out of range (entry %u)", i);
      return RDAMD_FAILURE;
    }
    rel[i] = clv_indices[i] - p->tips;
  }
  std::vector<int> phys_sc;
  if (p->sparse && p->sites) {
    phys_sc.assign(scaler_indices, scaler_indices + count);
    for (unsigned i = 0; i < count; ++i) {
      unsigned c = 0;
      RDAMD_HIP_TRY(clv_phys(p, clv_indices[i], &c), RDAMD_FAILURE);
      RDAMD_HIP_TRY(scaler_phys(p, scaler_indices[i], &phys_sc[i]), RDAMD_FAILURE);
      rel[i] = c - p->tips;
    }
    scaler_indices = phys_sc.data();
  }
  for (unsigned r = 0; r < p->rate_cats; ++r)
    if (freqs_indices[r] >= p->rate_matrices) {
      set_error(7, "rdamd_compute_root_loglikelihoods: freqs index out of range");
      return RDAMD_FAILURE;
    }
  if (p->sites == 0) {
    std::fill(lnl_out, lnl_out + count, 0.0);
    return RDAMD_SUCCESS;
  }
  RDAMD_HIP_TRY(flush_q(p), RDAMD_FAILURE);
  const unsigned blocks = root_lnl_blocks(p);
  const size_t need = 4096 + sizeof(unsigned) * p->rate_cats + (size_t)count * (8 + 8) +
                      sizeof(double) * ((size_t)count * blocks + count);
  RDAMD_HIP_TRY(ensure_scratch(p, need), RDAMD_FAILURE);
  Scratch sc{p};
  unsigned *d_fi = (unsigned *)sc.take(sizeof(unsigned) * p->rate_cats);
  unsigned *d_rel = (unsigned *)sc.take(sizeof(unsigned) * count);
  int *d_sci = (int *)sc.take(sizeof(int) * count);
  double *d_partials = (double *)sc.take(sizeof(double) * (size_t)count * blocks);
  double *d_out = (double *)sc.take(sizeof(double) * count);
  RDAMD_HIP_TRY(upload(p, d_fi, freqs_indices, sizeof(unsigned) * p->rate_cats), RDAMD_FAILURE);
  RDAMD_HIP_TRY(upload(p, d_rel, rel.data(), sizeof(unsigned) * count), RDAMD_FAILURE);
  RDAMD_HIP_TRY(upload(p, d_sci, scaler_indices, sizeof(int) * count), RDAMD_FAILURE);
  p->prof_begin(2);
  hipError_t le = launch_root_lnl_batch(p, count, d_rel, d_sci, d_fi, d_partials, d_out);
  p->prof_end();
  RDAMD_HIP_TRY(le, RDAMD_FAILURE);
  RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), RDAMD_FAILURE);
  RDAMD_HIP_TRY(hipMemcpy(lnl_out, d_out, sizeof(double) * count, hipMemcpyDeviceToHost),
                RDAMD_FAILURE);
  return RDAMD_SUCCESS;
}

int rdamd_root_loglikelihood_fused(rdamd_partition_t *p, const rdamd_operation_t *root_op,
                                   const unsigned int *params_indices,
                                   const double *lengths1, const double *lengths2,
                                   unsigned int n_alpha, double *lnl_out) {
  clear_error();
  if (n_alpha == 0) return RDAMD_SUCCESS;
  const unsigned R = p->rate_cats, K = p->states;
  if (p->sites == 0) {
    std::fill(lnl_out, lnl_out + n_alpha, 0.0);
    return RDAMD_SUCCESS;
  }
  const bool fast = K == 4 && p->ncodes_cap == 16 &&
                    (R == 1 || R == 2 || R == 4 || R == 8) &&
                    root_op->parent_scaler_index >= 0;
  if (!fast) {
    // generic shapes: the three calls queued back to back on the partition stream
    for (unsigned a = 0; a < n_alpha; ++a) {
      unsigned mi[2] = {root_op->child1_matrix_index, root_op->child2_matrix_index};
      double bl[2] = {lengths1[a], lengths2[a]};
      if (rdamd_update_prob_matrices(p, params_indices, mi, bl, 2) != RDAMD_SUCCESS)
        return RDAMD_FAILURE;
      rdamd_update_clvs(p, root_op, 1);
      if (rdamd_errno()) return RDAMD_FAILURE;
      lnl_out[a] = rdamd_compute_root_loglikelihood(
          p, root_op->parent_clv_index, root_op->parent_scaler_index, params_indices, nullptr);
      if (rdamd_errno()) return RDAMD_FAILURE;
    }
    return RDAMD_SUCCESS;
  }
  const unsigned nclv = p->tips + p->clv_buffers;
  if (root_op->parent_clv_index < p->tips || root_op->parent_clv_index >= nclv ||
      root_op->child1_clv_index >= nclv || root_op->child2_clv_index >= nclv ||
      root_op->child1_matrix_index >= p->prob_matrices ||
      root_op->child2_matrix_index >= p->prob_matrices ||
      root_op->parent_scaler_index >= (int)p->scale_buffers ||
      root_op->child1_scaler_index >= (int)p->scale_buffers ||
      root_op->child2_scaler_index >= (int)p->scale_buffers) {
    set_error(10, "rdamd_root_loglikelihood_fused: index out of range");
    return RDAMD_FAILURE;
  }
  for (unsigned r = 0; r < R; ++r)
    if (params_indices[r] >= p->rate_matrices) {
      set_error(7, "rdamd_root_loglikelihood_fused: params index out of range");
      return RDAMD_FAILURE;
    }
  rdamd_operation_t phys_root;
  RDAMD_HIP_TRY(op_phys(p, *root_op, &phys_root), RDAMD_FAILURE);
  root_op = &phys_root;
  RDAMD_HIP_TRY(flush_q(p), RDAMD_FAILURE);
  if (p->tiptab_stale) {
    RDAMD_HIP_TRY((p->stream_dirty = true, launch_tiptab_all(p)), RDAMD_FAILURE);
    p->tiptab_stale = false;
  }
  // One launch per chunk of up to eight positions (four at 8 rate categories;
  // root_single_dna_kernel): branch
  // lengths and parameter indices travel as kernel arguments, the P-matrices are
  // exponentiated inside the kernel, the result lands in the pinned host block.
  // The LAST position of the call leaves its matrices, root CLV and scaler in the
  // partition, exactly as the unfused call sequence would.
  LevelOp op;
  op.parent_clv = root_op->parent_clv_index; op.child1_clv = root_op->child1_clv_index;
  op.child2_clv = root_op->child2_clv_index; op.child1_mat = root_op->child1_matrix_index;
  op.child2_mat = root_op->child2_matrix_index; op.parent_sc = root_op->parent_scaler_index;
  op.child1_sc = root_op->child1_scaler_index; op.child2_sc = root_op->child2_scaler_index;
  op.src1 = op.src2 = 0;
  for (unsigned a = 0; a < n_alpha; ++a)
    if (!(lengths1[a] >= 0.0) || !(lengths2[a] >= 0.0) || !std::isfinite(lengths1[a]) ||
        !std::isfinite(lengths2[a])) {
      set_error(9, "rdamd_root_loglikelihood_fused: invalid branch length");
      return RDAMD_FAILURE;
    }
  const unsigned chunk = root_single_max_positions(R);
  for (unsigned base = 0; base < n_alpha; base += chunk) {
    const unsigned n = std::min(chunk, n_alpha - base);
    p->prof_begin(2);
    hipError_t e = launch_root_single(p, op, lengths1 + base, lengths2 + base, n, params_indices,
                                      p->d_counter, p->h_result);
    p->prof_end();
    RDAMD_HIP_TRY(e, RDAMD_FAILURE);
    RDAMD_HIP_TRY(sync_main(p), RDAMD_FAILURE);
    for (unsigned a = 0; a < n; ++a) lnl_out[base + a] = p->h_result[a];
  }
  return RDAMD_SUCCESS;
}

// Root-only evaluations of SEVERAL partitions in one launch (root_multi_dna_kernel): the
// Brent / finite-difference steps of the candidates a lock-stepped search has in flight, each
// on its own replica.  Item i: partition parts[i], its root operation ops[i], n_positions[i]
// <= 8 root positions (<= 4 at 8 rate categories) with branch lengths len1[8 i + a],
// len2[8 i + a]; out[8 i + a] = lnL.
// Every partition is left exactly as rdamd_root_loglikelihood_fused leaves it, and every value
// has that call's bits.  Shapes the one-launch kernel does not take fall back to it item by item.
int rdamd_root_loglikelihood_fused_multi(unsigned int n_items, rdamd_partition_t *const *parts,
                                         const rdamd_operation_t *ops,
                                         const unsigned int *const *params_indices,
                                         const double *len1, const double *len2,
                                         const unsigned int *n_positions, double *out) {
  clear_error();
  if (n_items == 0) return RDAMD_SUCCESS;
  bool fast = true;
  unsigned max_pos = 0, max_blocks = 0;
  for (unsigned i = 0; i < n_items; ++i) {
    const rdamd_partition *p = parts[i];
    fast = fast && p->states == 4 && p->ncodes_cap == 16 && p->rate_cats == parts[0]->rate_cats &&
           (p->rate_cats == 1 || p->rate_cats == 2 || p->rate_cats == 4 || p->rate_cats == 8) &&
           ops[i].parent_scaler_index >= 0 && p->sites > 0 && p->device == parts[0]->device &&
           n_positions[i] >= 1 && n_positions[i] <= root_single_max_positions(p->rate_cats);
  }
  if (!fast || n_items == 1) {
    for (unsigned i = 0; i < n_items; ++i)
      if (rdamd_root_loglikelihood_fused(parts[i], &ops[i], params_indices[i], len1 + 8 * i, len2 + 8 * i,
                                         n_positions[i], out + 8 * i) != RDAMD_SUCCESS)
        return RDAMD_FAILURE;
    return RDAMD_SUCCESS;
  }
  rdamd_partition *lead = parts[0];
  const unsigned R = lead->rate_cats;
  if (lead->root_items_cap < n_items) {
    if (lead->d_root_items) (void)hipFree(lead->d_root_items);
    if (lead->h_root_items) (void)hipHostFree(lead->h_root_items);
    lead->d_root_items = lead->h_root_items = nullptr;
    lead->root_items_cap = std::max(64u, n_items * 2);
    const size_t bytes = (size_t)lead->root_items_cap * (sizeof(RootItem) + kRootMaxPositions * sizeof(double));
    RDAMD_HIP_TRY(hipMalloc(&lead->d_root_items, bytes), RDAMD_FAILURE);
    RDAMD_HIP_TRY(hipHostMalloc(&lead->h_root_items, bytes, hipHostMallocDefault), RDAMD_FAILURE);
  }
  RootItem *h_items = (RootItem *)lead->h_root_items, *d_items = (RootItem *)lead->d_root_items;
  double *h_res = (double *)(h_items + lead->root_items_cap);
  for (unsigned i = 0; i < n_items; ++i) {
    rdamd_partition *p = parts[i];
    const unsigned nclv = p->tips + p->clv_buffers;
    if (ops[i].parent_clv_index < p->tips || ops[i].parent_clv_index >= nclv || ops[i].child1_clv_index >= nclv ||
        ops[i].child2_clv_index >= nclv || ops[i].child1_matrix_index >= p->prob_matrices ||
        ops[i].child2_matrix_index >= p->prob_matrices || ops[i].parent_scaler_index >= (int)p->scale_buffers ||
        ops[i].child1_scaler_index >= (int)p->scale_buffers || ops[i].child2_scaler_index >= (int)p->scale_buffers) {
      set_error(10, "rdamd_root_loglikelihood_fused_multi: item %u: index out of range", i);
      return RDAMD_FAILURE;
    }
    rdamd_operation_t o;
    RDAMD_HIP_TRY(op_phys(p, ops[i], &o), RDAMD_FAILURE);
    RootItem &it = h_items[i];
    memset(&it, 0, sizeof it);
    for (unsigned a = 0; a < kRootMaxPositions; ++a) {   // (unused positions repeat the last one: same state left behind)
      const unsigned src = std::min(a, n_positions[i] - 1);
      it.ra.len1[a] = len1[8 * i + src];
      it.ra.len2[a] = len2[8 * i + src];
      if (!(it.ra.len1[a] >= 0.0) || !(it.ra.len2[a] >= 0.0) || !std::isfinite(it.ra.len1[a]) ||
          !std::isfinite(it.ra.len2[a])) {
        set_error(9, "rdamd_root_loglikelihood_fused_multi: item %u: invalid branch length", i);
        return RDAMD_FAILURE;
      }
    }
    for (unsigned r = 0; r < 8; ++r) {
      it.ra.params_idx[r] = r < R ? params_indices[i][r] : 0u;
      if (it.ra.params_idx[r] >= p->rate_matrices) {
        set_error(7, "rdamd_root_loglikelihood_fused_multi: item %u: params index out of range", i);
        return RDAMD_FAILURE;
      }
    }
    RDAMD_HIP_TRY(flush_q(p), RDAMD_FAILURE);
    if (p->tiptab_stale) {
      RDAMD_HIP_TRY((p->stream_dirty = true, launch_tiptab_all(p)), RDAMD_FAILURE);
      p->tiptab_stale = false;
    }
    // whatever this partition's own stream still has queued (parameter uploads just now) must
    // be done before the leader's stream reads it
    if (p != lead && (p->stream_dirty || p->stream_external)) RDAMD_HIP_TRY(sync_main(p), RDAMD_FAILURE);
    it.v = p->view();
    it.op.parent_clv = o.parent_clv_index; it.op.child1_clv = o.child1_clv_index;
    it.op.child2_clv = o.child2_clv_index; it.op.child1_mat = o.child1_matrix_index;
    it.op.child2_mat = o.child2_matrix_index; it.op.parent_sc = o.parent_scaler_index;
    it.op.child1_sc = o.child1_scaler_index; it.op.child2_sc = o.child2_scaler_index;
    it.q = p->d_q; it.rates = p->d_rates; it.freqs = p->d_freqs; it.rate_w = p->d_rate_weights;
    it.pw = p->d_pattern_weights; it.codemask = p->d_codemask;
    it.partials = p->d_partials; it.counter = p->d_counter;
    it.result = h_res + kRootMaxPositions * i;   // (pinned host memory: the folding wave writes it there, no copy launch)
    it.blocks = root_single_blocks(p);
    it.ra.n_positions = n_positions[i];
    max_pos = std::max(max_pos, n_positions[i]);
    max_blocks = std::max(max_blocks, it.blocks);
  }
  RDAMD_HIP_TRY(hipMemcpyAsync(d_items, h_items, sizeof(RootItem) * n_items, hipMemcpyHostToDevice, lead->stream),
                RDAMD_FAILURE);
  lead->prof_begin(2);
  hipError_t e = launch_root_multi(d_items, n_items, R, max_pos, max_blocks, lead->stream);
  lead->prof_end();
  RDAMD_HIP_TRY(e, RDAMD_FAILURE);
  RDAMD_HIP_TRY(sync_main(lead), RDAMD_FAILURE);
  for (unsigned i = 0; i < n_items; ++i)
    for (unsigned a = 0; a < n_positions[i]; ++a) out[8 * i + a] = h_res[8 * i + a];
  return RDAMD_SUCCESS;
}

int rdamd_get_clv(rdamd_partition_t *p, unsigned int clv_index, double *out) {
  clear_error();
  const size_t S = p->sites, R = p->rate_cats, K = p->states, KA = p->api_states;
  if (clv_index < p->tips) {  // tips live as codes; expand on the host
    const uint8_t *row = p->tipcodes.data() + (size_t)clv_index * S;
    for (size_t s = 0; s < S; ++s) {
      uint64_t mask = p->codemask[row[s]];
      for (size_t r = 0; r < R; ++r)
        for (size_t j = 0; j < KA; ++j) out[(s * R + r) * KA + j] = (double)((mask >> j) & 1);
    }
    return RDAMD_SUCCESS;
  }
  if (clv_index >= p->tips + p->clv_buffers) {
    set_error(11, "rdamd_get_clv: index out of range");
    return RDAMD_FAILURE;
  }
  RDAMD_HIP_TRY(clv_phys(p, clv_index, &clv_index), RDAMD_FAILURE);
  RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), RDAMD_FAILURE);
  const double *src = p->d_clv + (size_t)(clv_index - p->tips) * p->clv_doubles();
  if (p->mfma_layout) {   // device: [rate][tile][operand layout] -> the ABI's [site][rate][state]
    std::vector<double> dev(p->clv_doubles());
    RDAMD_HIP_TRY(hipMemcpy(dev.data(), src, sizeof(double) * dev.size(), hipMemcpyDeviceToHost),
                  RDAMD_FAILURE);
    const size_t tiles = p->clv_tiles();
    for (size_t s = 0; s < S; ++s)
      for (size_t r = 0; r < R; ++r) {
        const double *tile = dev.data() + (r * tiles + s / 16) * (16 * K);
        for (size_t j = 0; j < KA; ++j)
          out[(s * R + r) * KA + j] = tile[k20_tile_index((unsigned)(s % 16), (unsigned)j)];
      }
    return RDAMD_SUCCESS;
  }
  if (!p->embedded()) {
    RDAMD_HIP_TRY(hipMemcpy(out, src, sizeof(double) * S * R * K, hipMemcpyDeviceToHost), RDAMD_FAILURE);
    return RDAMD_SUCCESS;
  }
  std::vector<double> wide(S * R * K);   // the caller's states are the first KA of every K
  RDAMD_HIP_TRY(hipMemcpy(wide.data(), src, sizeof(double) * S * R * K, hipMemcpyDeviceToHost),
                RDAMD_FAILURE);
  for (size_t e = 0; e < S * R; ++e)
    for (size_t j = 0; j < KA; ++j) out[e * KA + j] = wide[e * K + j];
  return RDAMD_SUCCESS;
}

int rdamd_get_scaler(rdamd_partition_t *p, unsigned int scaler_index, unsigned int *out) {
  clear_error();
  if (scaler_index >= p->scale_buffers) {
    set_error(11, "rdamd_get_scaler: index out of range");
    return RDAMD_FAILURE;
  }
  int phys_sc = (int)scaler_index;
  RDAMD_HIP_TRY(scaler_phys(p, (int)scaler_index, &phys_sc), RDAMD_FAILURE);
  scaler_index = (unsigned)phys_sc;
  RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), RDAMD_FAILURE);
  RDAMD_HIP_TRY(hipMemcpy(out, p->d_scaler + (size_t)scaler_index * p->sites,
                          sizeof(unsigned) * p->sites, hipMemcpyDeviceToHost), RDAMD_FAILURE);
  return RDAMD_SUCCESS;
}

int rdamd_get_pmatrix(rdamd_partition_t *p, unsigned int matrix_index, double *out) {
  clear_error();
  if (matrix_index >= p->prob_matrices) {
    set_error(8, "rdamd_get_pmatrix: index out of range");
    return RDAMD_FAILURE;
  }
  const size_t K = p->states, KA = p->api_states, n = (size_t)p->rate_cats * K * K;
  RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), RDAMD_FAILURE);
  if (!p->embedded()) {
    RDAMD_HIP_TRY(hipMemcpy(out, p->d_pmat + (size_t)matrix_index * n, sizeof(double) * n,
                            hipMemcpyDeviceToHost), RDAMD_FAILURE);
    return RDAMD_SUCCESS;
  }
  std::vector<double> wide(n);
  RDAMD_HIP_TRY(hipMemcpy(wide.data(), p->d_pmat + (size_t)matrix_index * n, sizeof(double) * n,
                          hipMemcpyDeviceToHost), RDAMD_FAILURE);
  for (size_t r = 0; r < p->rate_cats; ++r)
    for (size_t i = 0; i < KA; ++i)
      for (size_t j = 0; j < KA; ++j) out[(r * KA + i) * KA + j] = wide[(r * K + i) * K + j];
  return RDAMD_SUCCESS;
}

void rdamd_profile_enable(rdamd_partition_t *p, int on) {
  (void)hipStreamSynchronize(p->stream);
  for (auto &sp : p->prof_spans) { p->prof_pool.push_back(sp.a); p->prof_pool.push_back(sp.b); }
  p->prof_spans.clear();
  p->profiling = on != 0;
}

int rdamd_profile_read(rdamd_partition_t *p, double ms_out[8], unsigned int launches_out[8]) {
  clear_error();
  RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), RDAMD_FAILURE);
  for (int k = 0; k < 8; ++k) { ms_out[k] = 0.0; launches_out[k] = 0; }
  for (auto &sp : p->prof_spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess && sp.kind >= 0 && sp.kind < 8) {
      ms_out[sp.kind] += ms;
      launches_out[sp.kind] += 1;
    }
    p->prof_pool.push_back(sp.a);
    p->prof_pool.push_back(sp.b);
  }
  p->prof_spans.clear();
  return RDAMD_SUCCESS;
}

void rdamd_partition_sync(rdamd_partition_t *p) {
  if (p && p->stream) (void)hipStreamSynchronize(p->stream);
}

}  // extern "C"
