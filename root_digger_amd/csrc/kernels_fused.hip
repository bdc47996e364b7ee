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
// registers, pending sibling CLVs on a stack whose depth the host minimises
// (Sethi-Ullman order): level 0 in registers, deeper levels in per-wave LDS.  Because the rate is wave-uniform, the 4x4
// P-matrix of an inner operand is a set of scalar (SGPR) operands of the FMAs:
// no LDS or VGPR traffic for it.  A tip operand costs no FMA at all: its 16x4
// table (one row per ambiguity code, built next to the P-matrices) is exactly
// 64 doubles, one per lane, dropped into LDS and read back by code.
// HBM traffic per evaluation drops from ~(2n-2) CLVs to n bytes per site, so
// the kernel is bound by instruction issue (FP64 FMAs first), not by HBM --
// DESIGN.md 4.1 has the counters and the ablations.
//
// Scaling: each (site, rate) lane keeps its own 2^256 rescale count (rescale
// when all four entries drop below 2^-256) and the root sum aligns the rate
// terms to the smallest count -- the per-rate-scaler form of the reference
// rule; every factor is an exact power of two, so the
// result differs from the per-site rule only where that rule would already
// have lost the category to underflow.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "expm_k4.hpp"
#include "fused.hpp"

namespace rdamd {

// A job's value must not depend on the kernel variant that happens to run it -- the sites per
// lane follow the launch size, the stack form the deepest program of the launch, and the
// lock-stepped search (batch_combiner.hpp) must walk the sequential search's trajectories bit
// for bit whatever its launches look like.  Left to itself the compiler contracts a * b + c
// into fused multiply-adds per instantiation, and not always the same pairs (round 3: six of
// them in the variant with the tip-tip rescale test; round 4: one job in 160 between one and
// two sites per lane).  So the evaluator's arithmetic is spelled out -- every fused
// multiply-add is an explicit fma() -- and contraction is off from here to the end of the
// evaluator kernel.
#pragma clang fp contract(off)

// exact 2^(-256 * d) for d >= 0 (0 once it underflows)
__device__ __forceinline__ double pow2_neg256(int d) {
  return d == 0 ? 1.0 : (d == 1 ? kScaleThreshold
       : (d == 2 ? kScaleThreshold * kScaleThreshold
       : (d == 3 ? kScaleThreshold * kScaleThreshold * kScaleThreshold : 0.0)));
}

__device__ __forceinline__ unsigned uni(unsigned x) {   // assert wave-uniformity
  return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}

// LDS per wave: two table slots (X and Y operand) of TR rows each -- a slot holds a table as
// two half tables with 16-byte rows, states 0-1 then states 2-3, 32 TR bytes --, then the
// CLV stack [depth][site slot][2 halves][64 lanes] double2, then the rescale-count stack
// [depth][site slot][64] ints.  NS = sites per lane.  TR = 16: tips and pseudo-tips of up to
// 16 classes (one double per lane fills a table); TR = 64: pseudo-tips of up to 64 classes as
// well, whose 2 KB tables go from memory straight into their slot (LDS-DMA, two
// `buffer_load_dwordx4 ... lds` per table: no VGPR holds them on the way).
template <int TR>
constexpr unsigned tab_doubles() { return 8u * TR; }

typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
  // descriptor inputs made provably wave-uniform (cdna_hip_programming.md T20)
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = uni((unsigned)u), hi = uni((unsigned)(u >> 32));
  void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)uni(bytes), 0x00020000);
}

template <int NS>
struct LaneState {
  double v[NS][4];
  int sc[NS];
};

// t = P . x  with the 4x4 P-matrix as scalar operands
__device__ __forceinline__ void matvec(const double *__restrict__ p, const double (&x)[4],
                                       double (&t)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    t[k] = __builtin_fma(p[k * 4 + 3], x[3],
                         __builtin_fma(p[k * 4 + 2], x[2], __builtin_fma(p[k * 4 + 1], x[1], p[k * 4 + 0] * x[0])));
}

// A tip-table row, addressed by its LDS byte offset.  A table (16 codes x 4 states)
// sits in LDS as two half tables with 16-byte rows -- states 0-1 at BASE, states 2-3
// at BASE + 256 -- so that the row offset is code x 16, which is what the tip-code
// array of the fused evaluator stores: the loaded byte IS the address, the table
// bases are instruction offsets, and no vector instruction is spent on addressing.
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f64x2 *lds_row_ptr;
typedef __attribute__((address_space(3))) double *lds_f64_ptr;
typedef __attribute__((address_space(3))) void *lds_void_ptr;
// Wave-uniform operands (step descriptors, P-matrices, frequencies) are read through the
// CONSTANT address space: such loads stay scalar (s_load into SGPRs) whatever else the kernel
// does.  Through plain global pointers the compiler demotes them to vector loads as soon as
// the kernel contains a memory-writing intrinsic it cannot see through -- the LDS-DMA table
// loads of the TR = 64 variants did exactly that (sgpr 106 -> 54, vgpr 127 -> 196).  The data
// are written by earlier launches and constant for this one.
template <typename T>
using const_as = const __attribute__((address_space(4))) T *;
template <typename T>
__device__ __forceinline__ const_as<T> to_const(const T *p) {
  return (const_as<T>)(unsigned long long)p;
}
template <typename T>   // a whole record (dword by dword: the compiler merges them into s_load_dwordx8 / x16)
__device__ __forceinline__ T load_const(const T *p) {
  static_assert(sizeof(T) % 4 == 0, "records of whole dwords");
  T out;
  const const_as<unsigned> src = to_const(reinterpret_cast<const unsigned *>(p));
  unsigned *dst = reinterpret_cast<unsigned *>(&out);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) dst[i] = src[i];
  return out;
}
template <unsigned BASE, int TR>
__device__ __forceinline__ void read_row(unsigned off, double (&t)[4]) {
  const lds_row_ptr row = (lds_row_ptr)(size_t)(off + BASE);
  const f64x2 lo = row[0], hi = row[TR];   // + 16 TR bytes: the other half table
  t[0] = lo[0]; t[1] = lo[1]; t[2] = hi[0]; t[3] = hi[1];
}
// the class / tip code of a site, stored as the LDS byte offset of its table row: a byte
// (code x 16 <= 240) in the arena of 16-row launches, 16 bits where tables have 64 rows
#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_CODES8)
// timing only (profiles/r6_codes8_ab.txt): what 8-bit codes in a 64-row launch could buy at most --
// every code is ONE byte at a byte stride (half the arena lines per rate pass); the byte that
// arrives is not the code, its upper four bits are a valid row offset
constexpr bool kAblCodes8 = true;
#else
constexpr bool kAblCodes8 = false;
#endif
template <int TR>
__device__ __forceinline__ unsigned load_code(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  if constexpr (TR == 16) return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs, voff, soff, 0);
  else if constexpr (kAblCodes8) return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs, voff, soff, 0) & 0xF0u;
  else return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0);
}

// v = tx * ty for the NS sites of a lane, then the 2^256 rescale of a site whose four entries
// are all < 2^-256 (entries are non-negative, so comparing the high words is exact; a NaN
// compares as large and never rescales) -- with ONE wave-uniform branch around the
// rescaling: the masked multiplies of a site whose test fails nowhere in the wave (the
// ordinary case: a 100-taxon tree never comes near 2^-256) are jumped over instead of being
// issued with an empty EXEC mask -- which costs their issue cycles all the same: ten vector
// instructions per step at two sites per lane, c2 70.5k -> 75.2k - 76.7k evaluations/s.
// (RDAMD_ABL_NOCHECK, ablation builds only -- timing, results are garbage where a rescale is due:
// 1 = no rescale test on the running-CLV x tip steps, 2 = none at all; profiles/rescale_ab.sh)
template <int NS, bool TEST = true>
__device__ __forceinline__ void combine_sites(const double (&tx)[NS][4], const double (&ty)[NS][4],
                                              double (&v)[NS][4], int (&sc)[NS]) {
  if constexpr (!TEST) {
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[q][k] = tx[q][k] * ty[q][k];
    return;
  }
  unsigned hmax[NS];
  bool any_small = false;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[q][k] = tx[q][k] * ty[q][k];
    hmax[q] = max(max((unsigned)__double2hiint(v[q][0]), (unsigned)__double2hiint(v[q][1])),
                  max((unsigned)__double2hiint(v[q][2]), (unsigned)__double2hiint(v[q][3])));
    any_small = any_small || hmax[q] < 0x2FF00000u;
  }
  if (__builtin_amdgcn_ballot_w64(any_small) != 0ull) {
#pragma unroll
    for (int q = 0; q < NS; ++q)
      if (hmax[q] < 0x2FF00000u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[q][k] *= kScaleFactor;
        sc[q] += 1;
      }
  }
}

#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_NOCHECK)
constexpr bool kTestRT = false, kTestRP = RDAMD_ABL_NOCHECK < 2;
#else
constexpr bool kTestRT = true, kTestRP = true;
#endif

// TTCHECK: rescale test on tip-tip steps too.  A variant is queued over all jobs of a batch
// (the one with the test only if the batch raised a flag at all, launch_fused_eval_ns); a
// workgroup looks at the flag the P-matrix / clade-table steps left in ITS job
// (FusedJob::tt_unsafe: some table entry of the job in (0, 2^-128), or a pseudo-tip class
// that would have been rescaled) and returns at once in the variant the job does not belong
// to.  Per job, not per launch: a job's value then does not depend on what else shares its
// launch -- the lock-stepped search combines the candidates' batches and must reproduce the
// sequential trajectories bit for bit --, and one candidate on the edge of its parameter
// range does not send the others to the slower plain programs.
// RL: stack levels that live in registers (1 or 2).  The second one costs 18 VGPRs per
// lane at two sites per lane (142: three waves per SIMD) and is taken when it frees
// enough LDS to more than pay for that (deep stacks of 500- and 1000-taxon trees:
// three LDS levels allow 10 waves per CU, two LDS levels 15).
// RW ("rates across waves"): a workgroup is R waves, wave w walks rate category w of the same
// 64 NS sites instead of one wave looping over the categories.  Nothing is shared between
// them until the root -- each has its own table slots and stack in LDS, and they run without a
// barrier --, but they start together and execute the same program, so the tip / class codes
// of a site are fetched once per workgroup (the other waves hit L1 / L2) instead of once per
// rate pass, a whole traversal apart: the 500- and 1000-taxon shapes, whose code arena
// (850 MB / 340 MB) no longer fits any cache level, spend their time on exactly those
// re-reads.  The rate terms meet in LDS and wave 0 folds them in rate order with
// the loop's own arithmetic: the same bits as the one-wave form.
// Stack entries of the SP > 0 kernels: a parked entry lives in the wave's one LDS slot if its step
// says so (flags 0x20000 park / 0x40000 pop: the host places every park on its own,
// traversal_compiler.hpp), in the wave's private segment otherwise -- entry `sp` of a stack that
// holds only those.  The choice is a scalar branch INSIDE one asm
// statement, both sides writing the same registers: as an `if` in the source the two-sites
// kernel grows from 124 to 146-164 VGPRs (register copies where the paths join) and loses
// the fourth wave per SIMD that the whole exercise is about.  The compiler does not count
// these memory instructions: its own counted waits only get stricter by them, and the pop
// waits for its data itself.
typedef double f64x2_t __attribute__((ext_vector_type(2)));
// CNT = false (the speculative variant: every rescale count is zero): the entry without its count
template <int NS, bool CNT = true>
__device__ __forceinline__ void stack_push(unsigned sp, unsigned flags, int q, unsigned stk_lds,
                                           unsigned stksc_lds, unsigned spill_off, const double (&v)[4], int sc) {
  const f64x2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
  const unsigned la = stk_lds + q * 2048u, lsc = stksc_lds + q * 256u;
  sp = uni(sp);   // (wave-uniform by construction; not every variant's compiler pass sees it)
  const unsigned so = spill_off + (sp * NS + q) * 48u;
  if constexpr (!CNT) {
    asm volatile(
        "s_bitcmp1_b32 %[fl], 17\n\t"   /* 0x20000: into the LDS slot */
        "s_cbranch_scc1 1f\n\t"
        "scratch_store_dwordx4 off, %[lo], %[so]\n\t"
        "scratch_store_dwordx4 off, %[hi], %[so] offset:16\n\t"
        "s_branch 2f\n"
        "1:\n\t"
        "ds_write_b128 %[la], %[lo]\n\t"
        "ds_write_b128 %[la], %[hi] offset:1024\n"
        "2:\n\t"
        "s_nop 0"
        :
        : [fl] "s"(uni(flags)), [lo] "v"(lo), [hi] "v"(hi), [so] "s"(so), [la] "v"(la)
        : "memory", "scc");
    return;
  }
  asm volatile(
      "s_bitcmp1_b32 %[fl], 17\n\t"   /* 0x20000: into the LDS slot */
      "s_cbranch_scc1 1f\n\t"
      "scratch_store_dwordx4 off, %[lo], %[so]\n\t"
      "scratch_store_dwordx4 off, %[hi], %[so] offset:16\n\t"
      "scratch_store_dword off, %[sc], %[so] offset:32\n\t"
      "s_branch 2f\n"
      "1:\n\t"
      "ds_write_b128 %[la], %[lo]\n\t"
      "ds_write_b128 %[la], %[hi] offset:1024\n\t"
      "ds_write_b32 %[lsc], %[sc]\n"
      "2:\n\t"
      "s_nop 0"   /* (a VALU write to 16-byte store data needs one wait state behind the store) */
      :
      : [fl] "s"(uni(flags)), [lo] "v"(lo), [hi] "v"(hi), [sc] "v"(sc), [so] "s"(so),
        [la] "v"(la), [lsc] "v"(lsc)
      : "memory", "scc");
}
template <int NS>
__device__ __forceinline__ void stack_pop(unsigned sp, unsigned flags, int q, unsigned stk_lds,
                                          unsigned stksc_lds, unsigned spill_off, double (&v)[4], int &sc) {
  f64x2_t lo, hi;
  const unsigned la = stk_lds + q * 2048u, lsc = stksc_lds + q * 256u;
  sp = uni(sp);   // (wave-uniform by construction; not every variant's compiler pass sees it)
  const unsigned so = spill_off + (sp * NS + q) * 48u;
  asm volatile(
      "s_bitcmp1_b32 %[fl], 18\n\t"   /* 0x40000: from the LDS slot */
      "s_cbranch_scc1 1f\n\t"
      "scratch_load_dwordx4 %[lo], off, %[so]\n\t"
      "scratch_load_dwordx4 %[hi], off, %[so] offset:16\n\t"
      "scratch_load_dword %[sc], off, %[so] offset:32\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_branch 2f\n"
      "1:\n\t"
      "ds_read_b128 %[lo], %[la]\n\t"
      "ds_read_b128 %[hi], %[la] offset:1024\n\t"
      "ds_read_b32 %[sc], %[lsc]\n\t"
      "s_waitcnt lgkmcnt(0)\n"
      "2:"
      : [lo] "=&v"(lo), [hi] "=&v"(hi), [sc] "=&v"(sc)
      : [fl] "s"(uni(flags)), [so] "s"(so), [la] "v"(la), [lsc] "v"(lsc)
      : "memory", "scc");
  v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
}

// The pop in two halves: the loads are ISSUED at the top of the step and awaited where the popped
// sibling is first needed, behind the step's matrix-vector product -- a scratch entry that was
// evicted from L2 takes a microsecond to come back, and with the wait inside the pop nothing of
// the wave's own work covered it.  (The wait statement takes the values as in / out operands:
// that is what orders their uses behind it.)
template <int NS, bool CNT = true>
__device__ __forceinline__ void stack_pop_issue(unsigned sp, unsigned flags, int q, unsigned stk_lds,
                                                unsigned stksc_lds, unsigned spill_off, f64x2_t &lo, f64x2_t &hi,
                                                int &sc) {
  const unsigned la = stk_lds + q * 2048u, lsc = stksc_lds + q * 256u;
  sp = uni(sp);
  const unsigned so = spill_off + (sp * NS + q) * 48u;
  if constexpr (!CNT) {
    sc = 0;
    asm volatile(
        "s_bitcmp1_b32 %[fl], 18\n\t"   /* 0x40000: from the LDS slot */
        "s_cbranch_scc1 1f\n\t"
        "scratch_load_dwordx4 %[lo], off, %[so]\n\t"
        "scratch_load_dwordx4 %[hi], off, %[so] offset:16\n\t"
        "s_branch 2f\n"
        "1:\n\t"
        "ds_read_b128 %[lo], %[la]\n\t"
        "ds_read_b128 %[hi], %[la] offset:1024\n"
        "2:"
        : [lo] "=&v"(lo), [hi] "=&v"(hi)
        : [fl] "s"(uni(flags)), [so] "s"(so), [la] "v"(la)
        : "memory", "scc");
    return;
  }
  asm volatile(
      "s_bitcmp1_b32 %[fl], 18\n\t"   /* 0x40000: from the LDS slot */
      "s_cbranch_scc1 1f\n\t"
      "scratch_load_dwordx4 %[lo], off, %[so]\n\t"
      "scratch_load_dwordx4 %[hi], off, %[so] offset:16\n\t"
      "scratch_load_dword %[sc], off, %[so] offset:32\n\t"
      "s_branch 2f\n"
      "1:\n\t"
      "ds_read_b128 %[lo], %[la]\n\t"
      "ds_read_b128 %[hi], %[la] offset:1024\n\t"
      "ds_read_b32 %[sc], %[lsc]\n"
      "2:"
      : [lo] "=&v"(lo), [hi] "=&v"(hi), [sc] "=&v"(sc)
      : [fl] "s"(uni(flags)), [so] "s"(so), [la] "v"(la), [lsc] "v"(lsc)
      : "memory", "scc");
}
template <int NS, bool CNT = true>
__device__ __forceinline__ void stack_pop_wait(f64x2_t (&lo)[NS], f64x2_t (&hi)[NS], int (&sc)[NS]) {
  if constexpr (!CNT) {
    if constexpr (NS == 2)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0])::"memory");
  } else if constexpr (NS == 2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(lo[0]), "+v"(hi[0]), "+v"(sc[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(sc[1])::"memory");
  else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(sc[0])::"memory");
}

// RDAMD_ABL_STAMPS (ablation builds only, profiles/step_timeline.py): WHERE INSIDE A STEP A WAVE WAITS.
// A few waves of one job (rdamd_abl_stamps_config: job, every stride-th workgroup) read the shader
// clock (s_memtime) at the phase boundaries of every step and leave the five readings per step in
// `rdamd_stamp_buf`; every wave executes the reads (the instrument is the same for all of them),
// only the chosen ones store.  A scheduling barrier on both sides of a reading keeps the compiler
// from moving a phase's instructions across it.
//   0 top of the step | 1 the operand tables have landed (vmcnt(0); tip-tip and running x tip
//   steps) | 2 the table rows have come back from LDS and the next step's codes / tables / descriptor
//   are requested | 3 the matrix-vector product(s) issued | 4 the step's last instruction issued
#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_STAMPS)
constexpr unsigned kStampWaves = 64, kStampSteps = 4096;
__device__ unsigned long long rdamd_stamp_buf[kStampWaves][kStampSteps][4];   // [t0 | t1 t2 | t3 t4 | kind rate]
__device__ unsigned rdamd_stamp_cfg[2] = {0xffffffffu, 1u};   // job, workgroup stride
// (registers: the evaluator has 125 of the 128 VGPRs four waves per SIMD allow and no free SGPR.
// The readings keep their low words only -- a step is far shorter than 2^32 ticks --, and the one
// lane that stores them does so inside one asm statement with EXEC = 1: two scratch VGPRs, live
// only there)
// RDAMD_ABL_STAMPS is a MASK of the readings that are taken (31: all five): a reading waits for
// its own value -- and with it for every scalar load and LDS read in flight --, so builds with
// fewer readings cross-check what the full set says about a phase.
#define RDAMD_STAMP(k)                                                                          \
  if ((RDAMD_ABL_STAMPS >> (k)) & 1) {                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    tstamp[k] = __builtin_readcyclecounter();   /* (64 bits, untouched until the store: no wait here) */ \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  }
#define RDAMD_STAMP_STORE(kind)                                                                 \
  if (stamped) {                                                                                \
    if (stamp_step < kStampSteps) {                                                             \
      unsigned long long *rec_ = &rdamd_stamp_buf[stamp_slot][stamp_step][0];                   \
      unsigned d0_, d1_, z_;                                                                    \
      unsigned long long sv_;                                                                   \
      asm volatile(                                                                             \
          "s_mov_b64 %[sv], exec\n\t"                                                           \
          "s_mov_b64 exec, 1\n\t"                                                               \
          "v_mov_b32 %[z], 0\n\t"                                                               \
          "v_mov_b32 %[d0], %[t0]\n\t"                                                          \
          "v_mov_b32 %[d1], %[th]\n\t"                                                          \
          "global_store_dword %[z], %[d0], %[rec]\n\t"                                          \
          "global_store_dword %[z], %[d1], %[rec] offset:4\n\t"                                 \
          "s_nop 1\n\t"                                                                         \
          "v_mov_b32 %[d0], %[t1]\n\t"                                                          \
          "v_mov_b32 %[d1], %[t2]\n\t"                                                          \
          "global_store_dword %[z], %[d0], %[rec] offset:8\n\t"                                 \
          "global_store_dword %[z], %[d1], %[rec] offset:12\n\t"                                \
          "s_nop 1\n\t"                                                                         \
          "v_mov_b32 %[d0], %[t3]\n\t"                                                          \
          "v_mov_b32 %[d1], %[t4]\n\t"                                                          \
          "global_store_dword %[z], %[d0], %[rec] offset:16\n\t"                                \
          "global_store_dword %[z], %[d1], %[rec] offset:20\n\t"                                \
          "s_nop 1\n\t"                                                                         \
          "v_mov_b32 %[d0], %[kd]\n\t"                                                          \
          "v_mov_b32 %[d1], %[rr]\n\t"                                                          \
          "global_store_dword %[z], %[d0], %[rec] offset:24\n\t"                                \
          "global_store_dword %[z], %[d1], %[rec] offset:28\n\t"                                \
          "s_mov_b64 exec, %[sv]"                                                                \
          : [d0] "=&v"(d0_), [d1] "=&v"(d1_), [z] "=&v"(z_), [sv] "=&s"(sv_)                     \
          : [rec] "s"(rec_), [t0] "s"((unsigned)tstamp[0]), [th] "s"((unsigned)(tstamp[0] >> 32)),       \
            [t1] "s"((unsigned)tstamp[1]), [t2] "s"((unsigned)tstamp[2]),                       \
            [t3] "s"((unsigned)tstamp[3]), [t4] "s"((unsigned)tstamp[4]), [kd] "s"(kind), [rr] "s"(r) \
          : "memory");                                                                          \
    }                                                                                           \
    ++stamp_step;                                                                               \
  }
#else
#define RDAMD_STAMP(k)
#define RDAMD_STAMP_STORE(kind)
#endif

// SP ("spill levels"): the LDS stack is what limits the resident waves once a program needs
// two levels of it (13.2 KB per two-sites wave: 12 waves per CU instead of 16 -- and with
// every wave waiting on its own dependent chain, throughput follows the wave count: c5 and
// 125.phy measured +22 % with the second level simply taken away).  With SP > 0 a wave keeps
// ONE stack entry in LDS and has a slot for each of its up to SP in-memory entries in its
// PRIVATE segment (scratch: memory the hardware hands out per wave slot, so the same few MB
// serve every wave that passes through and stay in L2): a park there is three stores, the pop
// three loads whose latency the other waves cover.  Which parks go where the host decides, park
// by park (traversal_compiler.hpp, place_levels; rounds 3 - 4: level by level, the busiest level
// in the register slot, the runner-up in the LDS slot): as many as fit the two single slots --
// c5's plain programs 237 of 250 where the two busiest levels hold 171 --, the rest here.
// EXPORT (rdamd_evaluate_root_children): the steps the host flagged (0x8000 / 0x10000: they compute the
// root operation's two children) also store the running CLV and its rescale count -- the one thing
// of a traversal the root-only steps of the search need afterwards (a6, src/model.cpp:415-446).
// SPEC ("speculative", round 6; FusedArgs::speculate, evaluate.hip decides): NO rescale test at all in
// the variant on the programs with pseudo-tips.  FP64 reaches down to 2^-1022 and the largest entry of
// a (site, rate) vector can only shrink on the way to the root (the rows of a P-matrix sum to 1, every
// entry lies in [0, 1]): if a site's rate sum AT THE ROOT is still >= 2^-900, nothing that matters to
// it has come near the end of the range -- a category that did underflow on the way is < 2^-1021 in
// truth and as computed, 2^-121 of the sum -- and the 2^256 rule, whose factors are exact powers of
// two, would have changed exponents only.  A wave that finds a smaller sum raises its job's
// tt_unsafe flag and the batch's word, exactly as a small table entry does: the job then runs in the
// second pass, plain program, every test (the flag is per job, whatever the launch looks like, so a
// job's value still depends on the job alone).  What it buys: the tests are 6 of a step's 59 vector
// instructions, and without them the step loses a branch and the rescale arm behind every product
// (profiles/r6_speculative_rescale.md: c2 2.45 -> 2.25 ms per launch).
// (amdgpu_waves_per_eu: left alone, the two-sites 16-row variant comes out at 129 VGPRs without the
// tests -- 126 with them -- and loses its fourth wave per SIMD; asked for four it takes 120.)
template <int NS, bool TTCHECK, int RL, int TR, bool RW, int SP, bool EXPORT = false, bool SPEC = false>
__global__ void __launch_bounds__(RW ? 512 : 64) __attribute__((amdgpu_waves_per_eu((SPEC && NS == 2 && RL == 1 && TR == 16) ? 4 : 1)))
fused_dna_eval_kernel(FusedArgs a) {
  static_assert(!SPEC || (!TTCHECK && !RW && !EXPORT), "the speculative variant: first pass, one wave per workgroup");
  constexpr bool kRT = kTestRT && !SPEC, kRP = kTestRP && !SPEC;
  extern __shared__ double lds[];
  // (read_row<> and the table writes address LDS bytes 0 and 32 TR absolutely: that
  // is the dynamic block only while this kernel has no static __shared__ in front of
  // it -- launch_fused_eval_ns checks the kernel's static LDS size on the host; a
  // device-side test here cost a factor 2.6, its trap path changes the whole kernel)
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = RW ? uni(threadIdx.x >> 6) : 0u;
  // (RW: the table slots of wave w sit at LDS bytes w x 64 TR: below every code -- a row offset
  // < 16 TR -- so the wave's base is OR-ed into the codes as they arrive, one v_and_or instead
  // of the v_and the compiler needs there anyway)
  const unsigned wbase = wave * (64u * TR);
  // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2); grid.x is
  // a multiple of 8, so an XCD owns a fixed eighth of the sites and sees every job's
  // tables.  (Measured alternative: every XCD walks whole jobs, so that its L2 holds one
  // job's tables -- c2 -4 %, c4 +2 %, c5 +1 %: not worth a second mapping.)
  unsigned job = blockIdx.y, bx = blockIdx.x;
  if (a.job_major) {   // (wave-uniform) workgroup L of the launch runs on XCD L % 8: give it a block of job (..) * 8 + L % 8
    const unsigned gx = gridDim.x, L = job * gx + bx, seq = L >> 3;
    job = uni((seq / gx) * 8u + (L & 7u));
    bx = uni(seq % gx);
    if (job >= a.n_jobs) return;
  }
  const unsigned S = a.sites, R = a.rate_cats;
  unsigned site[NS];
  bool valid[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    site[q] = (bx * NS + q) * 64 + lane;
    valid[q] = site[q] < S;
    if (!valid[q]) site[q] = S - 1;
  }

  // a tip-tip step multiplies two tip-table rows: with every non-zero table entry of
  // the job >= 2^-128 (checked where the tables are built) the product is 0 or
  // >= 2^-256 and the rescale test cannot fire on a non-zero vector: it is compiled
  // out of the variant that runs then (+3.5 % on c2; a run-time branch gave nothing)
  if ((uni(to_const(&a.jobs[job].tt_unsafe)[0]) != 0u) != TTCHECK) return;
  // A padding block of the grid (blocks_x is a multiple of 16 so that an XCD keeps its eighth of
  // the sites: up to 15 blocks of 64 sites beyond the alignment's end -- 12.5 % of the workgroups
  // of a 6 250-site shard, 0.25 % of c2's): nothing to evaluate, its partial sums are the zeros a
  // wave of clamped sites would have arrived at after a whole traversal (round 5).
  if (bx * (64u * NS) >= S) {
    if (threadIdx.x == 0)
#pragma unroll
      for (int q = 0; q < NS; ++q) a.partials[(size_t)job * (gridDim.x * NS) + bx * NS + q] = 0.0;
    return;
  }
  constexpr bool tt_safe = !TTCHECK;
  // (the variant with the test walks the PLAIN programs -- no pseudo-tips, every rescale
  // where the reference rule has it; fused.hpp)
  const FusedJob jb = load_const(a.jobs + job);
  const FusedOp *__restrict__ prog = TTCHECK ? jb.prog_plain : jb.prog;   // n_ops + 4 entries (tail padded)
  const unsigned nops = TTCHECK ? jb.n_ops_plain : jb.n_ops;
  const unsigned job_levels = TTCHECK ? jb.depth_plain : jb.depth;
  const unsigned lds_levels = SP > 0 ? (job_levels < 1u ? job_levels : 1u) : job_levels;
  const const_as<char> pm = to_const(reinterpret_cast<const char *>(a.pmat + (size_t)job * a.pmat_job_stride));
  const const_as<double> freqs = to_const(a.freqs + (size_t)job * 4);
  const const_as<double> rw = to_const(a.rate_weights + (size_t)job * R);
  // tip codes and this job's tip tables through buffer descriptors: the
  // per-operation part of every address is a scalar offset, the per-lane part
  // a loop-invariant VGPR, so address generation costs no vector instruction
  const __amdgpu_buffer_rsrc_t tips_rs = make_rsrc(a.tipcodes, a.tipcodes_bytes);
  // (TR = 64: kFusedDmaBias bytes in front of the job's tables, see RDAMD_LOAD_TABS64; the workspace has that pad)
  const __amdgpu_buffer_rsrc_t tab_rs =
      make_rsrc(reinterpret_cast<const char *>(a.tiptab + (size_t)job * a.tiptab_job_stride) - (TR == 16 ? 0 : kFusedDmaBias),
                (unsigned)(a.tiptab_job_stride * 8) + (TR == 16 ? 0u : kFusedDmaBias));
  // (the tables' descriptor as four dwords, for the asm block in RDAMD_LOAD_TABS64)
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const unsigned long long tab_base = reinterpret_cast<unsigned long long>(
      reinterpret_cast<const char *>(a.tiptab + (size_t)job * a.tiptab_job_stride) - (TR == 16 ? 0 : kFusedDmaBias));
  const u32x4_t tab_desc = {uni((unsigned)tab_base), uni((unsigned)(tab_base >> 32)),
                            uni((unsigned)(a.tiptab_job_stride * 8) + (TR == 16 ? 0u : kFusedDmaBias)), 0x00020000u};
  const int lane8 = (int)lane * 8, lane16 = (int)lane * 16;
  const int lane4 = (int)lane * 4;
  static_assert(TR == 16 || TR == 64, "RDAMD_LOAD_TABS64's asm spells out the slot offsets of 64-row slots");
  static_assert(kFusedDmaYSlot == 32u * 64u, "... the Y slot's among them");
  static_assert(kFusedTT == 0 && kFusedRP == 2, "... and the step kinds it skips an operand for");
  // where this lane's entry of a 16-row table (code lane / 4, state lane % 4) goes in LDS: see read_row
  const unsigned tab_wr = wbase + ((lane & 2u) ? 16u * TR : 0u) + (lane >> 2) * 16u + (lane & 1u) * 8u;
  constexpr unsigned kYSlot = 32u * TR;   // the X / Y table slots sit at LDS bytes 0 and 32 TR
  int site_off[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) site_off[q] = (int)site[q] * (TR == 16 || kAblCodes8 ? 1 : 2);
  // LDS: the table slots of all waves first, then each wave's stack
  const unsigned n_waves = RW ? R : 1u;
  double *my_stack = lds + n_waves * tab_doubles<TR>() + (size_t)wave * lds_levels * NS * 288;
  double2 *stk = reinterpret_cast<double2 *>(my_stack) + lane;
  int *stk_sc = reinterpret_cast<int *>(my_stack + (size_t)lds_levels * NS * 256) + lane;
  // SP > 0: LDS byte addresses of this lane's stack entries and the private-segment offset of
  // the deep levels, for stack_push / stack_pop (the array is only ever touched by their
  // scratch instructions; handing its address to them keeps it allocated)
  const unsigned stk_lds = (unsigned)(size_t)stk, stksc_lds = (unsigned)(size_t)stk_sc;
  char spill_mem[SP > 0 ? SP * NS * 48 : 4] __attribute__((aligned(16)));
  const unsigned spill_off = SP > 0 ? uni((unsigned)(size_t)(__attribute__((address_space(5))) char *)spill_mem) : 0u;

  double term[NS];   // sum_r w_r f_r 2^(-256 (s_r - smin))
  int smin[NS];
#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_STAMPS)
  const unsigned stamp_stride = uni(to_const(rdamd_stamp_cfg)[1]);
  const bool stamped = uni(to_const(rdamd_stamp_cfg)[0]) == job && stamp_stride && bx % stamp_stride == 0 &&
                       bx / stamp_stride < kStampWaves;
  const unsigned stamp_slot = stamped ? bx / stamp_stride : 0u;
  unsigned stamp_step = 0;
  unsigned long long tstamp[5] = {0, 0, 0, 0, 0};
#endif

  for (unsigned r = RW ? wave : 0u; r < (RW ? wave + 1u : R); ++r) {
    LaneState<NS> st;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      st.sc[q] = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) st.v[q][k] = 0.0;
    }
    unsigned sp = 0;
    const unsigned roff = uni(r * 128u);   // byte offset of rate r inside a matrix slot
    const unsigned roff4 = uni(r * 512u), roff16 = uni(r * 2048u);   // ... inside a 16-row / a 64-row table
    // Software pipeline, unrolled by two with ping-pong register sets (A/B) so
    // that no prefetched value is ever copied at the loop edge (a copy would
    // force the wait for the prefetch into the same iteration): while op i
    // computes, the tip codes / tip-table entries of op i+1 and the descriptor
    // of op i+2 are in flight.
    // Operand tables of the NEXT operation.  TR = 16: a table is one double per lane,
    // prefetched into a register and dropped into its slot when the operation starts.
    // TR = 64: every table goes from memory straight into its slot (LDS-DMA, no register on
    // the way): one piece per half table -- 64 lanes x 16 bytes for a 64-row table (flags
    // 0x2000 X / 0x4000 Y), 64 lanes x 4 bytes for a 16-row one.  The choice is a scalar branch
    // INSIDE one asm statement: as an `if` in the source it costs the two-sites-per-lane kernel
    // 21 VGPRs and 90 register copies at the joins (145 against 124 VGPRs: three waves per
    // SIMD instead of four), although neither side loads a register.  Measured alternatives on
    // c2 with the 16-row kernel's schedules (class limit 17; that kernel: 2.98 ms per launch):
    // 16-byte pieces for both sizes, the lanes behind a 16-row half table out of range -- they
    // WRITE ZEROS, 4 x the bytes into LDS (profiles/micro/lds_dma_oob.hip) -- 3.44 ms; the extra
    // pieces of a 64-row table issued with EXEC = 0 otherwise: 5.94 ms (an instruction without
    // lanes still costs the address unit its ~10 cycles); this form 3.17 ms.  The same statement
    // skips an operand the step does not have (X unless it is tip-tip, Y if it pops; as an
    // `if` in the source that measured c2 -9 %).  The compiler does not count these loads: its own counted
    // waits only get stricter by that, and the step that reads the tables waits for everything.
    // A 16-row table's second half sits 256 bytes behind the first in memory but 16 TR bytes
    // behind it in the slot (the instruction offset moves source AND destination,
    // profiles/micro/lds_dma_offset.hip; the scalar offset makes up the difference, which is
    // why the descriptor starts kFusedDmaBias bytes in front of the job's tables).  The wait in front: this
    // operation's own rows must have left the slots before new tables land in them.
#define RDAMD_LOAD_TAB16(op, tOFF, e)                                                           \
  e = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(                          \
      tab_rs, lane8, (int)((uni(op.tOFF) + roff) * 4u), 0));
    // TR = 64, BOTH operands in one statement (round 5: the scalar side on a diet -- the step
    // issued 40 scalar instructions for 40 FP64 ones, two thirds of them here).  M0, the LDS base
    // of an LDS-DMA load, is set ONCE, to the wave's slot base: the Y slot and the second half of
    // a table are reached through the instruction offset, which moves source and destination
    // alike -- the host has taken the Y slot's 2 KB off the Y operand's scalar offset
    // (FusedOp::pad, traversal_compiler.hpp; the descriptor starts kFusedDmaBias bytes in front of
    // the job's tables so that no offset goes negative).  M0 is a reserved register to the
    // compiler (never allocated: it sets it right in front of each of its own uses), so it is
    // neither saved nor restored.  The per-rate part of the offset is one s_add inside the arm
    // that knows the table's size.
#define RDAMD_LOAD_TABS64(op)                                                                   \
  {                                                                                             \
    unsigned t0;                                                                                \
    asm volatile(                                                                               \
        "s_cmp_eq_u32 %[kind], 2\n\t"   /* the step pops: no table at all */                  \
        "s_cbranch_scc1 9f\n\t"                                                               \
        "s_mov_b32 m0, %[wb]\n\t"                                                             \
        "s_bitcmp1_b32 %[fl], 14\n\t"   /* 0x4000: Y has 64 rows */                           \
        "s_cbranch_scc1 1f\n\t"                                                               \
        "s_add_u32 %[t0], %[soy], %[r4]\n\t"                                                  \
        "buffer_load_dword %[vo4], %[rs], %[t0] offen offset:2048 lds\n\t"                    \
        "s_sub_u32 %[t0], %[t0], 0x300\n\t"                                                   \
        "buffer_load_dword %[vo4], %[rs], %[t0] offen offset:3072 lds\n\t"                    \
        "s_branch 2f\n"                                                                        \
        "1:\n\t"                                                                              \
        "s_add_u32 %[t0], %[soy], %[r16]\n\t"                                                 \
        "buffer_load_dwordx4 %[vo16], %[rs], %[t0] offen offset:2048 lds\n\t"                 \
        "buffer_load_dwordx4 %[vo16], %[rs], %[t0] offen offset:3072 lds\n"                    \
        "2:\n\t"                                                                              \
        "s_cmp_lg_u32 %[kind], 0\n\t"   /* X: tip-tip steps only */                           \
        "s_cbranch_scc1 9f\n\t"                                                               \
        "s_bitcmp1_b32 %[fl], 13\n\t"   /* 0x2000: X has 64 rows */                           \
        "s_cbranch_scc1 3f\n\t"                                                               \
        "s_add_u32 %[t0], %[sox], %[r4]\n\t"                                                  \
        "buffer_load_dword %[vo4], %[rs], %[t0] offen lds\n\t"                                \
        "s_sub_u32 %[t0], %[t0], 0x300\n\t"                                                   \
        "buffer_load_dword %[vo4], %[rs], %[t0] offen offset:1024 lds\n\t"                    \
        "s_branch 9f\n"                                                                        \
        "3:\n\t"                                                                              \
        "s_add_u32 %[t0], %[sox], %[r16]\n\t"                                                 \
        "buffer_load_dwordx4 %[vo16], %[rs], %[t0] offen lds\n\t"                             \
        "buffer_load_dwordx4 %[vo16], %[rs], %[t0] offen offset:1024 lds\n"                    \
        "9:"                                                                                    \
        : [t0] "=&s"(t0)                                                                        \
        : [kind] "s"(uni(op.flags) & 3u), [fl] "s"(uni(op.flags)), [wb] "s"(wbase),             \
          [sox] "s"(uni(op.pad[0])), [soy] "s"(uni(op.pad[1])), [r4] "s"(roff4), [r16] "s"(roff16), \
          [vo4] "v"(lane4), [vo16] "v"(lane16), [rs] "s"(tab_desc)                              \
        : "memory", "scc");                                                                     \
  }
    // THE FORM OF THE KERNELS WITH THE RESCALE TESTS: one statement per operand, M0 saved and restored
    // around it, the offsets worked out in the loop -- 15 scalar instructions per step more than
    // RDAMD_LOAD_TABS64 above and 1.5 - 2 % FASTER on c2 and c5's shard, equal on c4's and 125.phy (same
    // box, two rounds each: profiles/r5_tabload_ab.txt; the ISA budget: profiles/r5_fused_step_isa.md):
    // there the vector unit is oversubscribed and the scalar unit is not what a step waits for.  The
    // SPECULATIVE kernels (no tests: 48 vector instructions per step instead of 59, issue slots 65 %
    // full) are bound by a wave's own chain instead, and there RDAMD_LOAD_TABS64 wins: c2 +2.6 %, 125.phy
    // +2 % alone, c2 +7.7 % / 125.phy +4.4 % / c2's 6 250-site shard +5.6 % with the early descriptor
    // (round 6, profiles/r6_speculative_rescale.md).  -DRDAMD_ABL_ONE_TABLOAD (ablation builds) forces it
    // everywhere for the A/B.
#define RDAMD_LOAD_TAB(op, SLOT, WIDE, tOFF, e, SKIP_IF)                                        \
  if (TR == 16) {                                                                               \
    e = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(                         \
        tab_rs, lane8, (int)((uni(op.tOFF) + roff) * 4u), 0));                                  \
  } else {                                                                                      \
    const unsigned wide = uni(op.flags) & WIDE;                                                 \
    const int so = (int)((uni(op.tOFF) + (wide ? roff * 4u : roff)) * 4u) + (int)kFusedDmaBias; \
    unsigned m0_saved;                                                                          \
    asm volatile(                                                                               \
        SKIP_IF "\n\t"   /* an operand the step does not have: nothing to load */               \
        "s_cbranch_scc1 3f\n\t"                                                               \
        "s_mov_b32 %[sv], m0\n\t"                                                             \
        "s_mov_b32 m0, %[m]\n\t"                                                              \
        "s_cmp_lg_u32 %[w], 0\n\t"                                                            \
        "s_cbranch_scc1 1f\n\t"                                                               \
        "buffer_load_dword %[vo4], %[rs], %[so] offen lds\n\t"                                \
        "buffer_load_dword %[vo4], %[rs], %[so2] offen offset:1024 lds\n\t"                   \
        "s_branch 2f\n"                                                                        \
        "1:\n\t"                                                                              \
        "buffer_load_dwordx4 %[vo16], %[rs], %[so] offen lds\n\t"                             \
        "buffer_load_dwordx4 %[vo16], %[rs], %[so] offen offset:1024 lds\n"                    \
        "2:\n\t"                                                                              \
        "s_mov_b32 m0, %[sv]\n"                                                                \
        "3:"                                                                                    \
        : [sv] "=&s"(m0_saved)                                                                  \
        : [m] "s"((unsigned)(SLOT) + wbase), [w] "s"(wide), [vo4] "v"(lane4), [vo16] "v"(lane16), [rs] "s"(tab_desc), \
          [so] "s"(so), [so2] "s"(so - 768), [kind] "s"(uni(op.flags) & 3u)                     \
        : "memory", "scc");                                                                     \
  }
#if !(defined(RDAMD_ABLATION) && defined(RDAMD_ABL_ONE_TABLOAD))
#define RDAMD_LOAD_TABS(op, ex, ey) \
  if (TR > 16 && SPEC) {                                                                        \
    __builtin_amdgcn_s_waitcnt(0xc07f);   /* lgkmcnt(0) */                                      \
    RDAMD_LOAD_TABS64(op)                                                                       \
  } else {                                                                                      \
    if (TR > 16) { __builtin_amdgcn_s_waitcnt(0xc07f);   /* lgkmcnt(0) */ RDAMD_WARM_M(op) }    \
    RDAMD_LOAD_TAB(op, 0u, 0x2000u, tX, ex, "s_cmp_lg_u32 %[kind], 0")       /* X: tip-tip steps only */ \
    RDAMD_LOAD_TAB(op, kYSlot, 0x4000u, tY, ey, "s_cmp_eq_u32 %[kind], 2")   /* Y: not when the step pops */ \
  }
#else
#define RDAMD_LOAD_TABS(op, ex, ey) \
  if (TR > 16) {                                                                                \
    __builtin_amdgcn_s_waitcnt(0xc07f);   /* lgkmcnt(0) */                                      \
    RDAMD_LOAD_TABS64(op)                                                                       \
  } else {                                                                                      \
    RDAMD_LOAD_TAB16(op, tX, ex) RDAMD_LOAD_TAB16(op, tY, ey)                                   \
  }
#endif
#define RDAMD_LOAD_TIPS_NOW(op, cx, cy, ex, ey)                                                 \
  _Pragma("unroll") for (int q = 0; q < NS; ++q) {                                              \
    cx[q] = load_code<TR>(tips_rs, site_off[q], (int)uni(op.cX)) | wbase;                       \
    cy[q] = load_code<TR>(tips_rs, site_off[q], (int)uni(op.cY)) | wbase;                       \
  }                                                                                             \
  RDAMD_LOAD_TABS(op, ex, ey)
#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_NOTIPS)
    // timing only: the codes and tables of the program's FIRST operation serve every step -- no code
    // load, no table DMA inside the loop: what the step costs when nothing it waits for comes from
    // memory (the LDS row reads and the matrix's scalar load stay)
#define RDAMD_LOAD_TIPS(op, cx, cy, ex, ey)                                                     \
  _Pragma("unroll") for (int q = 0; q < NS; ++q) { cx[q] = cxA[q] | cxB[q]; cy[q] = cyA[q] | cyB[q]; } \
  if (TR > 16) __builtin_amdgcn_s_waitcnt(0xc07f);
#else
#define RDAMD_LOAD_TIPS(op, cx, cy, ex, ey) RDAMD_LOAD_TIPS_NOW(op, cx, cy, ex, ey)
#endif
    // (TR = 64: the descriptor of the operation after next only now -- fetched at the top of
    // the step it would sit in front of the wait above)
#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_EARLY_DESC)   /* A/B: the 64-row kernels fetch it at the top too */
#define RDAMD_LATE_DESC(cur, idx2)
#define RDAMD_EARLY_DESC(cur, idx2) cur = load_const(prog + (idx2));
#else
    // (the speculative 64-row kernels fetch it at the top like the 16-row ones: without the rescale
    // tests a step waits more than it computes, and the earlier request pays -- c2 +3.8 % alone, +7.7 %
    // together with the one-statement table loads below; the kernels WITH the tests lose 6 % on the
    // deep shapes that way: profiles/r6_speculative_rescale.md, section 4)
#define RDAMD_LATE_DESC(cur, idx2) if (TR > 16 && !SPEC) cur = load_const(prog + (idx2));
#define RDAMD_EARLY_DESC(cur, idx2) if (TR == 16 || SPEC) cur = load_const(prog + (idx2));
#endif
    // the ONE matrix an operation applies to the running CLV, into SGPRs
#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_MSAME)   /* timing only: every matrix load hits the scalar cache */
#define RDAMD_M_OFFSET(op) 0u
#else
#define RDAMD_M_OFFSET(op) uni(op.pM)
#endif
#define RDAMD_LOAD_M(op, M)                                                                     \
  {                                                                                             \
    const const_as<double> pp = (const_as<double>)(pm + RDAMD_M_OFFSET(op) + roff);             \
    _Pragma("unroll") for (int k = 0; k < 16; ++k) M[k] = pp[k];                                \
  }
    // WARM: the 128 bytes of the NEXT operation's matrix are touched (two dword loads) as soon as
    // its descriptor is known to have arrived -- behind the step's one lgkmcnt(0) wait --, so that
    // the 16-double load behind the matrix-vector product finds them in the scalar cache instead of
    // in L2.  The loads land in s101: the register allocator never hands out s100 / s101 on gfx9
    // (its limit is 100 SGPRs beside VCC; "reserved registers" to an asm clobber list) while the
    // wave's allocation (100 + 6 rounded up to 112) holds them -- a sink that nothing else reads
    // or writes, so no live range has to cover the loads' flight.
#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_WARM_M)
#define RDAMD_WARM_M(op)                                                                        \
  if (TR > 16) {                                                                                \
    asm volatile("s_load_dword s101, %0, 0x0\n\ts_load_dword s101, %0, 0x40"                    \
                 :: "s"(pm + uni(op.pM) + roff) : "memory");                                    \
  }
#else
#define RDAMD_WARM_M(op)
#endif

    // one traversal step: `cur`/c?/e?/M hold op i (all arrived), `nxt` is the
    // descriptor of op i+1 whose tip data and matrix are fetched into
    // nc?/ne?/M (M is consumed before it is refilled); finally the `cur` slot is
    // refilled with the descriptor of op i+2.
#define RDAMD_STEP(cur, nxt, cx, cy, ex, ey, ncx, ncy, nex, ney, idx2)                          \
  {                                                                                             \
    const unsigned kind = uni(cur.flags);                                                       \
    RDAMD_STAMP(0)                                                                              \
    RDAMD_EARLY_DESC(cur, idx2)                                                                 \
    unsigned rowx[NS], rowy[NS];   /* byte offsets of the rows inside the X / Y table = the codes */ \
    _Pragma("unroll") for (int q = 0; q < NS; ++q) { rowx[q] = cx[q]; rowy[q] = cy[q]; }        \
    double tx[NS][4], ty[NS][4];                                                                \
    const unsigned k3 = kind & 3u;                                                              \
    if (k3 == kFusedTT) {                                                                       \
      if (TR == 16) {                                                                           \
        ((lds_f64_ptr)(size_t)tab_wr)[0] = ex;                                                  \
        ((lds_f64_ptr)(size_t)tab_wr)[4 * TR] = ey;                                             \
      } else {   /* the tables came by DMA: the compiler does not order the reads below behind it */ \
        __builtin_amdgcn_s_waitcnt(0x0f70);   /* vmcnt(0) */                                    \
      }                                                                                         \
      RDAMD_STAMP(1)                                                                            \
      _Pragma("unroll") for (int q = 0; q < NS; ++q) { read_row<0, TR>(rowx[q], tx[q]); read_row<kYSlot, TR>(rowy[q], ty[q]); } \
      RDAMD_LOAD_TIPS(nxt, ncx, ncy, nex, ney) RDAMD_LATE_DESC(cur, idx2)                       \
      RDAMD_STAMP(2)                                                                            \
      if (kind & 0x100u) { /* park M . (running CLV) for the later inner-inner node */          \
        if (kind & 0x200u) { /* stack level 0 lives in registers: the product lands there */    \
          _Pragma("unroll") for (int q = 0; q < NS; ++q) {                                      \
            matvec(M, st.v[q], s0[q]);                                                          \
            if (!SPEC) s0sc[q] = st.sc[q];                                                      \
          }                                                                                     \
        } else if (RL >= 2 && (kind & 0x800u)) { /* ... and so does level 1 */                  \
          _Pragma("unroll") for (int q = 0; q < NS; ++q) {                                      \
            matvec(M, st.v[q], s1[q]);                                                          \
            if (!SPEC) s1sc[q] = st.sc[q];                                                      \
          }                                                                                     \
        } else {                                                                                \
          _Pragma("unroll") for (int q = 0; q < NS; ++q) {                                      \
            double tp[4];                                                                       \
            matvec(M, st.v[q], tp);                                                             \
            if (SP > 0) {                                                                       \
              stack_push<NS, !SPEC>(sp, kind, q, stk_lds, stksc_lds, spill_off, tp, st.sc[q]);  \
            } else {                                                                            \
              double2 *d = stk + (size_t)(sp * NS + q) * 128;                                   \
              d[0] = make_double2(tp[0], tp[1]);                                                \
              d[64] = make_double2(tp[2], tp[3]);                                               \
              if (!SPEC) stk_sc[(sp * NS + q) * 64] = st.sc[q];                                 \
            }                                                                                   \
          }                                                                                     \
          if (SP == 0 || !(kind & 0x20000u)) ++sp;   /* (SP > 0: the private segment's entries) */ \
        }                                                                                       \
      }                                                                                         \
      RDAMD_STAMP(3)                                                                            \
      RDAMD_LOAD_M(nxt, M)                                                                      \
      _Pragma("unroll") for (int q = 0; q < NS; ++q) {                                          \
        st.sc[q] = 0;                                                                           \
        if (tt_safe) { _Pragma("unroll") for (int k = 0; k < 4; ++k) st.v[q][k] = tx[q][k] * ty[q][k]; } \
      }                                                                                         \
      if (!tt_safe) combine_sites<NS>(tx, ty, st.v, st.sc);                                     \
    } else if (k3 == kFusedRT) {                                                                \
      if (TR == 16) ((lds_f64_ptr)(size_t)tab_wr)[4 * TR] = ey;                                 \
      else __builtin_amdgcn_s_waitcnt(0x0f70);   /* vmcnt(0): the table came by DMA */          \
      RDAMD_STAMP(1)                                                                            \
      _Pragma("unroll") for (int q = 0; q < NS; ++q) read_row<kYSlot, TR>(rowy[q], ty[q]);      \
      RDAMD_LOAD_TIPS(nxt, ncx, ncy, nex, ney) RDAMD_LATE_DESC(cur, idx2)                       \
      RDAMD_STAMP(2)                                                                            \
      _Pragma("unroll") for (int q = 0; q < NS; ++q) matvec(M, st.v[q], tx[q]);                 \
      RDAMD_STAMP(3)                                                                            \
      RDAMD_LOAD_M(nxt, M)                                                                      \
      combine_sites<NS, kRT>(tx, ty, st.v, st.sc);                                          \
    } else { /* kFusedRP: running CLV times M, sibling already multiplied when parked */        \
      if (kind & 0x400u) { /* the sibling waits in the register slot: used in place */          \
        RDAMD_STAMP(1)                                                                          \
        RDAMD_LOAD_TIPS(nxt, ncx, ncy, nex, ney) RDAMD_LATE_DESC(cur, idx2)                     \
        RDAMD_STAMP(2)                                                                          \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) matvec(M, st.v[q], tx[q]);               \
        RDAMD_STAMP(3)                                                                          \
        RDAMD_LOAD_M(nxt, M)                                                                    \
        if (!SPEC) { _Pragma("unroll") for (int q = 0; q < NS; ++q) st.sc[q] += s0sc[q]; }      \
        combine_sites<NS, kRP>(tx, s0, st.v, st.sc);                                                 \
      } else if (RL >= 2 && (kind & 0x1000u)) {                                                 \
        RDAMD_STAMP(1)                                                                          \
        RDAMD_LOAD_TIPS(nxt, ncx, ncy, nex, ney) RDAMD_LATE_DESC(cur, idx2)                     \
        RDAMD_STAMP(2)                                                                          \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) matvec(M, st.v[q], tx[q]);               \
        RDAMD_STAMP(3)                                                                          \
        RDAMD_LOAD_M(nxt, M)                                                                    \
        if (!SPEC) { _Pragma("unroll") for (int q = 0; q < NS; ++q) st.sc[q] += s1sc[q]; }      \
        combine_sites<NS, kRP>(tx, s1, st.v, st.sc);                                                 \
      } else {                                                                                  \
        int scy[NS];                                                                            \
        f64x2_t plo[NS], phi[NS];                                                               \
        if (SP == 0 || !(kind & 0x40000u)) --sp;                                                \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) {                                        \
          if (SP > 0) {   /* issued here, awaited behind the product (stack_pop_issue) */       \
            stack_pop_issue<NS, !SPEC>(sp, kind, q, stk_lds, stksc_lds, spill_off, plo[q], phi[q], scy[q]); \
          } else {                                                                              \
            const double2 *d = stk + (size_t)(sp * NS + q) * 128;                               \
            const double2 lo = d[0], hi = d[64];                                                \
            ty[q][0] = lo.x; ty[q][1] = lo.y; ty[q][2] = hi.x; ty[q][3] = hi.y;                 \
            scy[q] = SPEC ? 0 : stk_sc[(sp * NS + q) * 64];                                     \
          }                                                                                     \
        }                                                                                       \
        RDAMD_STAMP(1)                                                                          \
        RDAMD_LOAD_TIPS(nxt, ncx, ncy, nex, ney) RDAMD_LATE_DESC(cur, idx2)                     \
        RDAMD_STAMP(2)                                                                          \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) matvec(M, st.v[q], tx[q]);               \
        RDAMD_STAMP(3)                                                                          \
        RDAMD_LOAD_M(nxt, M)                                                                    \
        if (SP > 0) {                                                                           \
          stack_pop_wait<NS, !SPEC>(plo, phi, scy);                                             \
          _Pragma("unroll") for (int q = 0; q < NS; ++q) {                                      \
            ty[q][0] = plo[q].x; ty[q][1] = plo[q].y; ty[q][2] = phi[q].x; ty[q][3] = phi[q].y; \
          }                                                                                     \
        }                                                                                       \
        if (!SPEC) { _Pragma("unroll") for (int q = 0; q < NS; ++q) st.sc[q] += scy[q]; }       \
        combine_sites<NS, kRP>(tx, ty, st.v, st.sc);                                        \
      }                                                                                         \
    }                                                                                           \
    RDAMD_STAMP(4)                                                                              \
    RDAMD_STAMP_STORE(kind)                                                                     \
    if (EXPORT && (kind & 0x18000u)) {   /* a child of the root operation: leave it behind */  \
      double *ec = (kind & 0x8000u) ? a.export_clv[0] : a.export_clv[1];                        \
      unsigned *en = (kind & 0x8000u) ? a.export_cnt[0] : a.export_cnt[1];                      \
      _Pragma("unroll") for (int q = 0; q < NS; ++q)                                            \
        if (valid[q]) {                                                                         \
          const size_t at = (size_t)site[q] * R + r;                                            \
          reinterpret_cast<double2 *>(ec + at * 4)[0] = make_double2(st.v[q][0], st.v[q][1]);   \
          reinterpret_cast<double2 *>(ec + at * 4)[1] = make_double2(st.v[q][2], st.v[q][3]);   \
          en[at] = (unsigned)st.sc[q];                                                          \
        }                                                                                       \
    }                                                                                           \
  }

    double s0[NS][4];   // stack level 0 (the most frequently used) stays in registers
    int s0sc[NS];
    double s1[NS][4];   // level 1 too when RL = 2 (unused, and gone from the code, when RL = 1)
    int s1sc[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      s0sc[q] = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s0[q][k] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      s1sc[q] = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s1[q][k] = 0.0;
    }
    FusedOp dA = load_const(prog);
    FusedOp dB = load_const(prog + 1);
    unsigned cxA[NS], cyA[NS], cxB[NS], cyB[NS];
    double exA = 0.5, eyA = 0.25, exB = 0.5, eyB = 0.25;
#pragma unroll
    for (int q = 0; q < NS; ++q) { cxA[q] = cxB[q] = 1; cyA[q] = cyB[q] = 2; }
    RDAMD_LOAD_TIPS_NOW(dA, cxA, cyA, exA, eyA)
#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_NOTIPS)
#pragma unroll
    for (int q = 0; q < NS; ++q) { cxB[q] = cxA[q]; cyB[q] = cyA[q]; }
#endif
    double M[16];
    RDAMD_LOAD_M(dA, M)
    unsigned i = 0;
    for (; i + 1 < nops; i += 2) {
      RDAMD_STEP(dA, dB, cxA, cyA, exA, eyA, cxB, cyB, exB, eyB, i + 2)
      RDAMD_STEP(dB, dA, cxB, cyB, exB, eyB, cxA, cyA, exA, eyA, i + 3)
    }
    if (i < nops) {   // odd tail (the program is padded, so the prefetches stay in bounds)
      RDAMD_STEP(dA, dB, cxA, cyA, exA, eyA, cxB, cyB, exB, eyB, i + 2)
    }
#undef RDAMD_STEP
    // root: f_r = sum_k pi_k v[k]; fold into the running rate sum
    const double w = rw[r];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      double f = __builtin_fma(st.v[q][3], freqs[3],
                               __builtin_fma(st.v[q][2], freqs[2], __builtin_fma(st.v[q][1], freqs[1], st.v[q][0] * freqs[0])));
      f *= w;
      if (RW) {   // (folded below, by wave 0, once all rates have arrived)
        term[q] = f;
        smin[q] = st.sc[q];
      } else if (r == 0) {
        term[q] = f;
        smin[q] = st.sc[q];
      } else if (st.sc[q] >= smin[q]) {
        term[q] = __builtin_fma(f, pow2_neg256(st.sc[q] - smin[q]), term[q]);
      } else {
        term[q] = __builtin_fma(term[q], pow2_neg256(smin[q] - st.sc[q]), f);
        smin[q] = st.sc[q];
      }
    }
  }

  // (the last step of every rate pass has requested the tables of the padding entry behind
  // the program: no LDS-DMA may still be on its way when the wave gives its LDS back)
  if (TR > 16) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  if (RW) {
    // the rate terms meet in LDS (every wave is done with its tables and stack: the space is
    // free after the first barrier), then wave 0 alone folds them, rate 0 first, with the
    // arithmetic of the loop above
    __syncthreads();
    double *xf = lds + (size_t)wave * NS * 64 + lane;
    int *xs = reinterpret_cast<int *>(lds + (size_t)R * NS * 64) + (size_t)wave * NS * 64 + lane;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      xf[q * 64] = term[q];
      xs[q * 64] = smin[q];
    }
    __syncthreads();
    if (wave != 0u) return;
    for (unsigned r = 1; r < R; ++r) {
      const double *yf = lds + (size_t)r * NS * 64 + lane;
      const int *ys = reinterpret_cast<const int *>(lds + (size_t)R * NS * 64) + (size_t)r * NS * 64 + lane;
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const double f = yf[q * 64];
        const int sc = ys[q * 64];
        if (sc >= smin[q]) {
          term[q] = __builtin_fma(f, pow2_neg256(sc - smin[q]), term[q]);
        } else {
          term[q] = __builtin_fma(term[q], pow2_neg256(smin[q] - sc), f);
          smin[q] = sc;
        }
      }
    }
  }
  if constexpr (SPEC) {   // a site whose sum is below 2^-900 (or 0): the job goes to the second pass
    bool low = false;
#pragma unroll
    for (int q = 0; q < NS; ++q) low = low || (valid[q] && (unsigned)__double2hiint(term[q]) < 0x07B00000u);
    if (__builtin_amdgcn_ballot_w64(low) != 0ull && lane == 0) {
      const_cast<FusedJob *>(a.jobs)[job].tt_unsafe = 1u;   // (every writer stores the same value)
      *a.any_unsafe = 1u;
    }
  }
  // One partial sum per 64-SITE BLOCK, whatever the number of sites per lane: a job's value
  // must not depend on the size of the launch it rides in (NS is chosen by that), so a wave
  // with two blocks reduces them separately and the finishing kernel adds the same partials
  // in the same order either way.
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    double l = __builtin_fma((double)smin[q], kLogScaleThreshold, log(term[q]));
    l *= (double)a.pattern_weights[site[q]];
    if (!valid[q]) l = 0.0;
    if (a.persite && valid[q]) a.persite[(size_t)job * S + site[q]] = l;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) l += __shfl_down(l, off);
    if (lane == 0) a.partials[(size_t)job * (gridDim.x * NS) + bx * NS + q] = l;
  }
}

#pragma clang fp contract(fast)   // (the kernels below keep the compiler's default)

// fixed-order finish, one workgroup per job.  A job belongs to exactly one of the two passes
// (FusedJob::tt_unsafe): the other pass's workgroups returned at once and left no partials, so
// its result is not touched -- `out` never holds a value made from stale partials.
__global__ void __launch_bounds__(256)
fused_finish_kernel(const double *__restrict__ partials, unsigned per_job,
                    const FusedJob *__restrict__ jobs, unsigned unsafe_pass,
                    double *__restrict__ out, double *__restrict__ host_out,
                    const unsigned *__restrict__ any_unsafe, unsigned *__restrict__ host_flag,
                    double *__restrict__ flag_f64) {
  __shared__ double lds[4];
  // (host_out / host_flag: pinned host memory, written from here -- the results and the word
  // that says whether the second pass is needed arrive with the kernel's end, no copy launches.
  // flag_f64: the same word as a double BEHIND a device-side copy of the results (host_out then
  // points at device memory): a collective queued behind this kernel sums it with them, and the
  // ranks of a site group learn together whether any of them needs the second pass)
  if (host_flag && blockIdx.x == 0 && threadIdx.x == 0) *host_flag = *any_unsafe;
  if (flag_f64 && blockIdx.x == 0 && threadIdx.x == 0) *flag_f64 = *any_unsafe ? 1.0 : 0.0;
  if ((jobs[blockIdx.x].tt_unsafe != 0u) != (unsafe_pass != 0u)) return;
  const double *p = partials + (size_t)blockIdx.x * per_job;
  double acc = 0.0;
  for (unsigned i = threadIdx.x; i < per_job; i += 256) acc += p[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double v = ((lds[0] + lds[1]) + lds[2]) + lds[3];
    out[blockIdx.x] = v;
    if (host_out) host_out[blockIdx.x] = v;
  }
}

// rdamd_evaluate_root_children: the exporting evaluator left a child's CLV with one rescale count
// per (site, rate); the root kernels (and rdamd_get_clv) read the reference's form, one count per
// site.  s = the smallest count of the site's rates, a rate that was rescaled d times more goes
// back by 2^(-256 d) -- an exact power of two, the value the per-site rule would hold there
// (or 0 / a denormal where that rule would have lost the rate to underflow as well).
// One-wave workgroups, both children in one launch (blockIdx.y): the call runs beside the
// objective launches of a lock-stepped search, where a four-wave workgroup waits for the end of
// the launch that fills the device (profiles/micro/side_kernel_latency.hip; measured here: 360 us
// per launch as 256-lane workgroups against 4 us alone).
struct ExportFixupArgs {
  double *clv[2];
  const unsigned *cnt[2];
  unsigned *scaler[2];
  unsigned sites, R;
};
__global__ void __launch_bounds__(64)
fused_export_fixup_kernel(ExportFixupArgs x) {
  const unsigned k = blockIdx.y;
  double *__restrict__ clv = x.clv[k];
  const unsigned *__restrict__ cnt = x.cnt[k];
  if (!clv) return;
  const unsigned R = x.R;
  for (unsigned site = blockIdx.x * 1024u + threadIdx.x; site < min(x.sites, (blockIdx.x + 1) * 1024u); site += 64u) {
  unsigned smin = cnt[(size_t)site * R];
  for (unsigned r = 1; r < R; ++r) smin = min(smin, cnt[(size_t)site * R + r]);
  for (unsigned r = 0; r < R; ++r) {
    const unsigned d = cnt[(size_t)site * R + r] - smin;
    if (d) {
      // (2^-1024 is still a number -- a denormal --, and so may be the product: what the per-site
      // rule's chain of small products arrives at, up to the denormals' rounding)
      const double f = d < 4u ? pow2_neg256((int)d) : (d == 4u ? 0x1p-1024 : 0.0);
      double *v = clv + ((size_t)site * R + r) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= f;
    }
  }
  x.scaler[k][site] = smin;
  }
}

// fused_finish_kernel for ONE job as a single wave (same reason): lane l plays threads l, l + 64,
// l + 128, l + 192 of that kernel's workgroup and the four wave sums meet in its order -- the
// same bits.
__global__ void __launch_bounds__(64)
fused_finish_wave_kernel(const double *__restrict__ partials, unsigned per_job, double *__restrict__ out,
                         double *__restrict__ host_out) {
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (unsigned w = 0; w < 4; ++w)
    for (unsigned i = w * 64u + threadIdx.x; i < per_job; i += 256) acc[w] += partials[i];
#pragma unroll
  for (unsigned w = 0; w < 4; ++w)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[w] += __shfl_down(acc[w], off);
  if (threadIdx.x == 0) {
    const double v = ((acc[0] + acc[1]) + acc[2]) + acc[3];
    out[0] = v;
    if (host_out) host_out[0] = v;
  }
}

// P-matrices and tip tables for a batch of jobs: 16 lanes per (job, matrix, rate) problem, a
// lane per matrix element (expm_k4_coop16: pmatrix_k4_kernel's arithmetic, element by
// element); a wave works off 64 problems, four at a time.  No LDS and ~40 registers: the
// launch runs beside the evaluator of the batch in front of it (evaluate.hip, pipelined
// batches) and its waves must fit the slots that launch's waves leave -- the one-lane-per-
// problem form (128 registers for the matrix, 9.7 KB of LDS to turn the results into
// contiguous stores) did not, and sat between the evaluators instead of beside them
// (profiles/micro/side_kernel_latency.hip).  Every store instruction writes 128 contiguous
// bytes per problem; the values are the same bits as before.
// (problems per wave = 4 x kSlimPasses.  Four times fewer, four times longer workgroups here and
// in the clade-table launch -- what pays for the root-only steps, kernels_root.hip -- measured no
// gain in the search: 21.9 s against 22.2 s for 66 candidates, three runs each on one box)
constexpr unsigned kSlimPasses = 16;
__global__ void __launch_bounds__(64)
fused_pmatrix_k4_kernel(const double *__restrict__ q, const double *__restrict__ rates,
                        FusedJob *__restrict__ jobs, unsigned n_jobs,
                        unsigned n_mat, unsigned R, double *__restrict__ pmat,
                        double *__restrict__ tiptab, size_t pmat_job_stride, size_t tiptab_job_stride,
                        unsigned table_rows, unsigned *__restrict__ any_unsafe) {
  const unsigned lane = threadIdx.x, e = lane & 15u;
  const int base = (int)(lane & ~15u);
  const size_t per_job = (size_t)n_mat * R;
  const size_t total = per_job * n_jobs;
  for (unsigned pass = 0; pass < kSlimPasses; ++pass) {
    const size_t first = (size_t)blockIdx.x * (kSlimPasses * 4) + pass * 4;
    if (first >= total) break;   // (wave-uniform)
    const size_t gid_raw = first + (lane >> 4);
    const bool live = gid_raw < total;
    const size_t gid = live ? gid_raw : total - 1;   // (idle groups of the last wave repeat its last problem)
    const unsigned job = (unsigned)(gid / per_job);
    const unsigned rem = (unsigned)(gid % per_job);
    const unsigned m = rem / R, r = rem % R;
    const double t = jobs[job].brlen[m] * rates[(size_t)job * R + r];
    double v = expm_k4_coop16(q + (size_t)job * 16, t);
    v = v <= 0.0 ? 0.0 : v;   // (<=: a -0.0 becomes +0.0 -- the rescale tests read high words)
    if (live) pmat[(size_t)job * pmat_job_stride + ((size_t)m * R + r) * 16 + e] = v;
    // tip table of the (matrix, rate): row c = sum over the states in code c; this lane's four
    // of its 64 entries (64-row launches: the evaluator's LDS image, [half][code][2 states] -- it
    // goes there by DMA; 16-row launches: [code][state])
    double *tt = tiptab + (size_t)job * tiptab_job_stride + ((size_t)m * R + r) * 64;
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
      const unsigned L = e + 16u * k;
      const unsigned c = table_rows > 16 ? (L >> 1) & 15u : L >> 2;
      const unsigned i = table_rows > 16 ? (L >> 5) * 2 + (L & 1u) : L & 3u;
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double pj = __shfl(v, base + (int)i * 4 + j);
        if ((c >> j) & 1) acc += pj;
      }
      if (live) tt[L] = acc;
    }
    // every table entry is a sum of P entries, so the smallest non-zero P entry bounds
    // them all from below (FusedJob::tt_unsafe; 2^-128 = 0x1p-128)
    if (live && v > 0.0 && v < 0x1p-128) {
      jobs[job].tt_unsafe = 1u;   // (every writer stores the same value)
      *any_unsafe = 1u;
    }
  }
}

// P-matrices for a batch of jobs, the form for a launch that has the device to itself
// (rdamd_evaluate_batch: everything on one stream): one thread per (job, matrix, rate).
// Same scaling-and-squaring / 16-term Taylor core as pmatrix_k4_kernel.
__global__ void __launch_bounds__(64)
fused_pmatrix_k4_wide_kernel(const double *__restrict__ q, const double *__restrict__ rates,
                        FusedJob *__restrict__ jobs, unsigned n_jobs,
                        unsigned n_mat, unsigned R, double *__restrict__ pmat,
                        double *__restrict__ tiptab, size_t pmat_job_stride, size_t tiptab_job_stride,
                        unsigned table_rows, unsigned *__restrict__ any_unsafe) {
  // the wave's 64 results, for the cooperative stores below ([problem][17]: no bank conflict
  // when every lane reads its own row)
  __shared__ double sh[64 * 17];
  __shared__ unsigned long long base_pm[64], base_tt[64];
  const unsigned lane = threadIdx.x;
  const size_t per_job = (size_t)n_mat * R;
  const size_t total = per_job * n_jobs;
  const size_t gid_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = gid_raw < total;
  const size_t gid = live ? gid_raw : total - 1;   // (idle lanes of the last wave repeat its last problem)
  const unsigned job = (unsigned)(gid / per_job);
  const unsigned rem = (unsigned)(gid % per_job);
  const unsigned m = rem / R, r = rem % R;
  const double t = jobs[job].brlen[m] * rates[(size_t)job * R + r];
  const double *qq = q + (size_t)job * 16;
  double out[16];
  expm_k4(qq, t, out);   // (the one definition of the arithmetic, expm_k4.hpp)
  // Stores: a thread's own 16 + 64 doubles would go out as 80 instructions of 64 lanes x 8
  // bytes, 512 bytes apart -- 64 cache lines each (235 MB of HBM traffic for 100 MB of data
  // on c2).  Instead the wave parks its results in LDS and writes problem by problem, a lane
  // per element: 512 contiguous bytes per instruction.
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    out[i] = out[i] <= 0.0 ? 0.0 : out[i];   // (<=: a -0.0 becomes +0.0 -- the rescale tests read high words)
    sh[lane * 17 + i] = out[i];
  }
  base_pm[lane] = (unsigned long long)job * pmat_job_stride + ((size_t)m * R + r) * 16;
  base_tt[lane] = (unsigned long long)job * tiptab_job_stride + ((size_t)m * R + r) * 64;
  __syncthreads();
  const unsigned n_live = (unsigned)(total - (size_t)blockIdx.x * 64 < 64 ? total - (size_t)blockIdx.x * 64 : 64);
#pragma unroll 4
  for (unsigned it = 0; it < 16; ++it) {   // P: four problems per instruction
    const unsigned e = it * 64 + lane, pr = e >> 4, k = e & 15u;
    if (pr < n_live) pmat[base_pm[pr] + k] = sh[pr * 17 + k];
  }
  // tip table of a (matrix, rate): row c = sum over the states in code c; my element of every
  // problem's table (64-row launches: the evaluator's LDS image, [half][code][2 states] -- it
  // goes there by DMA; 16-row launches: [code][state])
  const unsigned c = table_rows > 16 ? (lane >> 1) & 15u : lane >> 2;
  const unsigned i = table_rows > 16 ? (lane >> 5) * 2 + (lane & 1u) : lane & 3u;
  for (unsigned pr = 0; pr < n_live; ++pr) {
    const double *o = sh + pr * 17 + i * 4;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((c >> j) & 1) acc += o[j];
    tiptab[base_tt[pr] + lane] = acc;
  }
  // every table entry is a sum of P entries, so the smallest non-zero P entry bounds
  // them all from below (FusedJob::tt_unsafe; 2^-128 = 0x1p-128)
  bool tiny = false;
#pragma unroll
  for (int k = 0; k < 16; ++k) tiny = tiny || (out[k] > 0.0 && out[k] < 0x1p-128);
  if (tiny) {
    jobs[job].tt_unsafe = 1u;   // (every writer stores the same value)
    *any_unsafe = 1u;
  }
}

// slim: the launch runs BESIDE an evaluator (pipelined batches): one-wave workgroups that fit
// the wave slots it leaves; otherwise the one-thread-per-problem form, which is ~60 us per
// 197-job batch cheaper when it has the device to itself.  Same bits either way.
hipError_t launch_fused_pmatrix(const FusedArgs &a, const double *d_q, const double *d_rates,
                                unsigned n_jobs, unsigned n_mat, bool slim, hipStream_t stream) {
  const size_t total = (size_t)n_jobs * n_mat * a.rate_cats;
  if (!total) return hipSuccess;
  if (slim)
    fused_pmatrix_k4_kernel<<<(unsigned)((total + kSlimPasses * 4 - 1) / (kSlimPasses * 4)), 64, 0, stream>>>(
        d_q, d_rates, const_cast<FusedJob *>(a.jobs), n_jobs, n_mat, a.rate_cats, const_cast<double *>(a.pmat),
        const_cast<double *>(a.tiptab), a.pmat_job_stride, a.tiptab_job_stride, a.table_rows, a.any_unsafe);
  else
    fused_pmatrix_k4_wide_kernel<<<(unsigned)((total + 63) / 64), 64, 0, stream>>>(
        d_q, d_rates, const_cast<FusedJob *>(a.jobs), n_jobs, n_mat, a.rate_cats, const_cast<double *>(a.pmat),
        const_cast<double *>(a.tiptab), a.pmat_job_stride, a.tiptab_job_stride, a.table_rows, a.any_unsafe);
  return hipGetLastError();
}

template <int NS, bool TTCHECK, int RL, int TR, bool RW, int SP, bool EXPORT = false, bool SPEC = false>
static hipError_t launch_fused_variant(const FusedArgs &a, unsigned n_jobs, unsigned max_depth, unsigned gx,
                                       hipStream_t stream) {
  const unsigned n_waves = RW ? a.rate_cats : 1u;
  const size_t per_wave = tab_doubles<TR>() * sizeof(double) +
                          (size_t)(SP > 0 || !max_depth ? 1 : max_depth) * NS * 64 * (4 * sizeof(double) + sizeof(int));
  // (RW: at least the room the rate terms need when they meet: R x NS x 64 x 12 bytes)
  const size_t lds = std::max<size_t>(per_wave * n_waves, (size_t)n_waves * NS * 64 * 12);
  {   // deep stacks (very unbalanced 10^3-taxon trees), or R of them: raise the limit -- never lower it:
      // partitions launch from their own host threads (the replicas of a lock-stepped search)
    static std::mutex lds_mu;
    static size_t lds_allowed = 48 * 1024;
    std::lock_guard<std::mutex> guard(lds_mu);
    if (lds > lds_allowed) {
      hipError_t e = hipFuncSetAttribute((const void *)fused_dna_eval_kernel<NS, TTCHECK, RL, TR, RW, SP, EXPORT, SPEC>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      lds_allowed = lds;
    }
  }
  static const bool lds_starts_at_zero = [] {   // see the note at the top of the kernel
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, (const void *)fused_dna_eval_kernel<NS, TTCHECK, RL, TR, RW, SP, EXPORT, SPEC>) == hipSuccess &&
           attr.sharedSizeBytes == 0;
  }();
  if (!lds_starts_at_zero) return hipErrorInvalidValue;
  fused_dna_eval_kernel<NS, TTCHECK, RL, TR, RW, SP, EXPORT, SPEC><<<dim3(gx, a.job_major ? (n_jobs + 7u) / 8u * 8u : n_jobs), 64 * n_waves, lds, stream>>>(a);
  return hipGetLastError();
}

template <int NS, bool TTCHECK, int TR>
static hipError_t launch_fused_variant_rl(const FusedArgs &a, unsigned n_jobs, unsigned max_depth, unsigned gx,
                                          unsigned reg_levels, hipStream_t stream) {
  // two or more stack levels behind the register one: one in LDS, the rest in the private segment
  // -- for the kernels with 64-row table slots.  Their steps wait for table DMAs and want the
  // waves; the 16-row kernels (partitions without site repeats) prefetch through registers, are
  // bound by instruction issue at 12 waves per CU already and only pay for the asm-fenced stack
  // accesses: c2 without repeats 47.5k evaluations/s on two LDS levels, 41.3k this way.
#ifdef RDAMD_ABLATION   // A/B runs only: RDAMD_FUSED_SPILL_MIN = smallest in-memory depth that takes the SP kernels
  static const unsigned spill_min = getenv("RDAMD_FUSED_SPILL_MIN") ? (unsigned)atoi(getenv("RDAMD_FUSED_SPILL_MIN")) : 2u;
#else
  constexpr unsigned spill_min = 2u;
#endif
  const bool spill = TR > 16 && reg_levels < 2 && max_depth >= spill_min && max_depth <= kFusedSpillLevels;
  if (a.rates_across_waves) {
    if (spill) return launch_fused_variant<NS, TTCHECK, 1, TR, true, kFusedSpillLevels>(a, n_jobs, max_depth, gx, stream);
    return reg_levels >= 2 ? launch_fused_variant<NS, TTCHECK, 2, TR, true, 0>(a, n_jobs, max_depth, gx, stream)
                           : launch_fused_variant<NS, TTCHECK, 1, TR, true, 0>(a, n_jobs, max_depth, gx, stream);
  }
  if constexpr (!TTCHECK)
    if (a.speculate) {   // (never with rates across waves: evaluate.hip)
      if (spill) return launch_fused_variant<NS, false, 1, TR, false, kFusedSpillLevels, false, true>(a, n_jobs, max_depth, gx, stream);
      return reg_levels >= 2 ? launch_fused_variant<NS, false, 2, TR, false, 0, false, true>(a, n_jobs, max_depth, gx, stream)
                             : launch_fused_variant<NS, false, 1, TR, false, 0, false, true>(a, n_jobs, max_depth, gx, stream);
    }
  if (spill) return launch_fused_variant<NS, TTCHECK, 1, TR, false, kFusedSpillLevels>(a, n_jobs, max_depth, gx, stream);
  return reg_levels >= 2 ? launch_fused_variant<NS, TTCHECK, 2, TR, false, 0>(a, n_jobs, max_depth, gx, stream)
                         : launch_fused_variant<NS, TTCHECK, 1, TR, false, 0>(a, n_jobs, max_depth, gx, stream);
}

// sites_per_lane: 1 or 2.  Two sites per lane share every scalar operand (P-matrix
// SGPRs, descriptors, control flow) and every tip-table row fetch between two
// sites: +10 % on c2 (197-job launches: 40.6k -> 44.8k evaluations/s, 124 VGPRs, four
// waves per SIMD), +5 % on c5; three and four sites per lane lose (register
// pressure: 31k and 22k).  One site per lane is kept for launches too small to
// fill the chip with half the waves (evaluate.hip picks).
// Two variants -- the one on the programs with pseudo-tips and no tip-tip rescale test
// ([0]), the one on the plain programs with it ([1]) --, each with the LDS and the register
// stack levels ITS programs need; a workgroup of the variant its job does not belong to
// returns at once.  A pass queues one of them over all jobs and the finishing kernel; the
// second pass is only queued when a flag went up in the batch (evaluate.hip reads
// FusedArgs::any_unsafe with the results: on ordinary data it never does, and a launch of
// 77 000 workgroups that all return at once still costs 20 us).
template <int NS, int TR>
static hipError_t launch_fused_eval_ns(const FusedArgs &a, unsigned n_jobs, const unsigned max_depth[2],
                                       unsigned blocks_x, const unsigned reg_levels[2], bool unsafe_pass,
                                       double *d_out, double *h_out, unsigned *h_flag, double *flag_f64,
                                       hipStream_t stream) {
  const unsigned gx = (blocks_x + NS - 1) / NS;   // blocks_x counts 64-site blocks
  hipError_t e = unsafe_pass ? launch_fused_variant_rl<NS, true, TR>(a, n_jobs, max_depth[1], gx, reg_levels[1], stream)
                             : launch_fused_variant_rl<NS, false, TR>(a, n_jobs, max_depth[0], gx, reg_levels[0], stream);
  if (e != hipSuccess) return e;
  fused_finish_kernel<<<n_jobs, 256, 0, stream>>>(a.partials, gx * NS, a.jobs, unsafe_pass ? 1u : 0u, d_out, h_out, a.any_unsafe, h_flag, flag_f64);   // 64-site blocks per job
  return hipGetLastError();
}

hipError_t launch_fused_eval(const FusedArgs &a, unsigned n_jobs, const unsigned max_depth[2],
                             unsigned blocks_x, unsigned sites_per_lane, const unsigned reg_levels[2],
                             bool unsafe_pass, double *d_out, double *h_out, unsigned *h_flag, hipStream_t stream,
                             double *flag_f64) {
  if (!n_jobs) return hipSuccess;
  if (a.table_rows > 16)   // 16-bit code arena, 64-row table slots
    return sites_per_lane == 2 ? launch_fused_eval_ns<2, 64>(a, n_jobs, max_depth, blocks_x, reg_levels, unsafe_pass, d_out, h_out, h_flag, flag_f64, stream)
                               : launch_fused_eval_ns<1, 64>(a, n_jobs, max_depth, blocks_x, reg_levels, unsafe_pass, d_out, h_out, h_flag, flag_f64, stream);
  return sites_per_lane == 2 ? launch_fused_eval_ns<2, 16>(a, n_jobs, max_depth, blocks_x, reg_levels, unsafe_pass, d_out, h_out, h_flag, flag_f64, stream)
                             : launch_fused_eval_ns<1, 16>(a, n_jobs, max_depth, blocks_x, reg_levels, unsafe_pass, d_out, h_out, h_flag, flag_f64, stream);
}

// One site per lane (a single job is a few hundred waves either way), all-LDS stack behind the
// register level(s), plain program with every rescale test; then the finishing kernel and, per
// exported child, the counts -> per-site scalers step.
hipError_t launch_fused_export(const FusedArgs &a, unsigned max_depth, unsigned blocks_x, unsigned reg_levels,
                               unsigned *const d_scaler[2], double *d_out, double *h_out, hipStream_t stream) {
  if (a.table_rows != 16 || a.rates_across_waves || a.job_major || a.n_jobs != 1) return hipErrorInvalidValue;
  hipError_t e = reg_levels >= 2 ? launch_fused_variant<1, true, 2, 16, false, 0, true>(a, 1, max_depth, blocks_x, stream)
                                 : launch_fused_variant<1, true, 1, 16, false, 0, true>(a, 1, max_depth, blocks_x, stream);
  if (e != hipSuccess) return e;
  fused_finish_wave_kernel<<<1, 64, 0, stream>>>(a.partials, blocks_x, d_out, h_out);
  if (a.export_clv[0] || a.export_clv[1]) {
    ExportFixupArgs x;
    for (int k = 0; k < 2; ++k) { x.clv[k] = a.export_clv[k]; x.cnt[k] = a.export_cnt[k]; x.scaler[k] = d_scaler[k]; }
    x.sites = a.sites; x.R = a.rate_cats;
    fused_export_fixup_kernel<<<dim3((a.sites + 1023u) / 1024u, 2), 64, 0, stream>>>(x);
  }
  return hipGetLastError();
}

}  // namespace rdamd

#if defined(RDAMD_ABLATION) && defined(RDAMD_ABL_STAMPS)
// profiles/step_timeline.py: which waves stamp, and their readings ([64][4096][8] u32; zeroed by _config)
extern "C" int rdamd_abl_stamps_config(unsigned job, unsigned stride) {
  const unsigned cfg[2] = {job, stride};
  void *buf = nullptr;
  if (hipGetSymbolAddress(&buf, HIP_SYMBOL(rdamd::rdamd_stamp_buf)) != hipSuccess) return 0;
  if (hipMemset(buf, 0, sizeof(rdamd::rdamd_stamp_buf)) != hipSuccess) return 0;
  return hipMemcpyToSymbol(HIP_SYMBOL(rdamd::rdamd_stamp_cfg), cfg, sizeof cfg) == hipSuccess;
}
extern "C" int rdamd_abl_stamps_read(unsigned long long *host) {
  if (hipDeviceSynchronize() != hipSuccess) return 0;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(rdamd::rdamd_stamp_buf), sizeof(rdamd::rdamd_stamp_buf)) == hipSuccess;
}
#endif
