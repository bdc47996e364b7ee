// Subtree site repeats for the fused evaluator: "clade tables".
//
// The reference turns coraxlib's site repeats on for every 4-state run
// (/root/reference/src/model.cpp:145-149, CORAX_ATTRIB_SITE_REPEATS): below an inner node
// the columns of the alignment fall into CLASSES of identical tip patterns, and the CLV of
// the node is computed once per class instead of once per site.  The fused evaluator keeps
// no CLV in memory, so it uses the repeats the way it already uses tips: a directed subtree
// ("clade") whose class count is small becomes a PSEUDO-TIP -- a per-site class code (one
// more row of the tip-code array, worked out once per partition on the host) plus, per job,
// a table  row[class] = P(branch above the clade) . CLV_clade(class)  that a small kernel
// fills next to the tip tables (clade_table_kernel, kernels_clade.hip).  The traversal
// compiler (evaluate.hip) then drops every operation inside the clade and hands the
// pseudo-tip to the clade's parent operation as if it were a tip: no matrix-vector product,
// no stack traffic, one table look-up.
//
// This file: the per-partition cache of directed subtrees (hash-consed by
// (child, child, matrix, matrix), so the 2n-3 schedules of one tree share their clades),
// their class codes, and the device-side records of the table kernel.
#pragma once

#include <hip/hip_runtime.h>

#include <array>
#include <cstdint>
#include <map>
#include <vector>

struct rdamd_partition;

namespace rdamd {

// One directed subtree.  Node ids: a tip is its tip index, an inner node is tips + its
// index in CladeCache::nodes.
struct CladeNode {
  unsigned child[2] = {0, 0};
  unsigned mat[2] = {0, 0};        // P-matrix indices of the two child branches
  unsigned n_tips = 0;
  unsigned n_classes = 0;          // 0: more than the cache's max_classes (never a pseudo-tip)
  std::vector<uint8_t> cls;        // [sites] class of every site (n_classes <= 256)
  std::vector<uint8_t> cmap;       // [n_classes][2]: class -> (class of child 0, class of child 1)
  int code_row[2] = {-1, -1};      // row in the fused evaluator's code arena ([0] 8-bit, [1] 16-bit entries;
                                   // uploaded on first use)
  long map_off = -1;               // byte offset of cmap in the device map arena (uploaded on first use)
};

struct CladeCache {
  unsigned max_classes = 64;   // the attribute's default (rdamd_partition_set_site_repeats changes it)
  std::map<std::array<unsigned, 4>, unsigned> intern;   // (child0, child1, mat0, mat1) -> node id
  std::vector<CladeNode> nodes;
  // device: class -> child-class maps of every node a schedule has used
  uint8_t *d_maps = nullptr;
  size_t maps_used = 0, maps_cap = 0;
  ~CladeCache();
};

// One node of one pseudo-tip of one schedule, in evaluation (post-) order; device record.
struct CladeStep {
  uint32_t map_off;        // byte offset of the node's class map in CladeCache::d_maps
  uint32_t n_classes;
  uint32_t src[2];         // child i: a tip -> its branch's matrix index (the job's tip table of
                           // that matrix); a nested clade -> 0x80000000 | step index inside the group
  uint32_t out_mat;        // the branch above this node: row = P[out_mat] . CLV
  uint32_t last;           // 1: the pseudo-tip itself -> the row goes to the job's table of out_mat
                           // (<= 16 classes) or to its 64-row table `wide_slot`; 0: nested -> scratch
                           // slot of this step
  uint32_t wide_slot;      // 0xffffffff: none
  uint32_t pad;
};
struct CladeGroup {        // one pseudo-tip = steps [first, first + count) of the schedule
  uint32_t first, count;
};

// id of the directed subtree (child0, child1 over branches mat0, mat1); computes its classes
// on first sight.  Children must be ids handed out earlier (or tips).
unsigned clade_intern(rdamd_partition *p, unsigned child0, unsigned child1, unsigned mat0, unsigned mat1);
inline const CladeNode *clade_node(const CladeCache &c, unsigned tips, unsigned id) {
  return id < tips ? nullptr : &c.nodes[id - tips];
}
// make sure the node's class codes sit in the code arena / its map in the map arena
hipError_t clade_upload_codes(rdamd_partition *p, unsigned id, bool wide);
hipError_t ensure_wide_arena(rdamd_partition *p);   // the 16-bit arena with the tips' rows in place
hipError_t clade_upload_map(rdamd_partition *p, unsigned id);

struct FusedArgs;
// scratch: [job][step][rate][rows][4] with rows = a.table_rows
hipError_t launch_clade_tables(const FusedArgs &a, const uint8_t *d_maps, double *d_scratch,
                               size_t scratch_job_stride, unsigned n_jobs, unsigned max_groups,
                               bool slim, hipStream_t stream);

}  // namespace rdamd
