// The traversal compiler of the fused evaluators (pure host code: no HIP call, no device
// memory -- tests/cpp/host_logic_check.cpp runs it on the CPU and replays its programs
// symbolically).  rdamd_schedule_create (evaluate.hip) is its only caller in the library.
#pragma once

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/root_digger_amd.h"
#include "fused.hpp"

namespace rdamd {

// ---- traversal compiler -----------------------------------------------------
// Input: operations in dependency order, the last one being the root.  Output:
// the same operations re-ordered so that, at every inner-inner node, the child
// needing the deeper stack is evaluated first (its result is parked in LDS
// while the other child runs), plus the flags the kernel interprets.
struct Compiler {
  const rdamd_operation_t *ops;
  unsigned n_ops, tips, sites, tip_stride, rate_cats;
  unsigned unit = 0;         // bytes between the [rate 0] entries of consecutive matrices
  bool split_park = false;   // 20-state programs: parking is a step of its own
  unsigned reg_levels = 1;   // stack levels the kernel keeps in registers (4 states: 1 or 2)
  unsigned reg_level = 0;    // reg_levels == 1: WHICH level that is (the busiest one, see the callers)
  std::unordered_map<unsigned, unsigned> producer;   // clv -> op index
  // pseudo-tips (clades.hpp): clv of a collapsed clade -> row of its class codes in the code
  // arena; to the compiler such a child is a tip whose table sits in its branch's slot
  std::unordered_map<unsigned, unsigned> pseudo_row;
  std::unordered_map<unsigned, unsigned> pseudo_wide;   // ... and, for a 64-row table, its slot among the job's
  unsigned wide_base = 0;                               // tX of wide slot 0 (behind the 16-row tables)
  bool dma_offsets = false;                             // 4-state programs for 64-row slots: FusedOp::pad = the
                                                        // scalar offsets of the table DMAs (fused.hpp)
  unsigned mark_clv[2] = {~0u, ~0u};                 // the root operation's children: the steps that compute them are
                                                     // flagged 0x8000 / 0x10000 (fused.hpp; the exporting variant)
  unsigned matvecs = 0;                              // inner children = matrix-vector products per (site, rate)
  std::vector<unsigned> need;                        // stack slots a subtree needs
  std::vector<FusedOp> out;
  unsigned depth = 0, max_depth = 0;
  unsigned parks_at[16] = {0};   // parks by stack level (0 = the first level, a register slot)
  bool ok = true;
  // Kernels with ONE register slot, ONE LDS slot and a private-segment stack behind them (4 states,
  // 64-row table slots: kernels_fused.hip, SP): every park is placed on its own instead of level by
  // level (place_levels).  park_class: by operation -- the inner-inner node whose first child waits
  // -- 1 = register slot (flags 0x200 / 0x400), 2 = LDS slot (0x20000 / 0x40000), 3 = private
  // segment; empty: by level.  mem_depth: most private-segment entries at a time.
  bool place_parks = false;
  std::vector<unsigned char> park_class;
  unsigned mem_now = 0, mem_depth = 0;
  std::vector<int> place_memo;

  bool is_inner(unsigned clv) const { return clv >= tips && !pseudo_row.count(clv); }
  unsigned row_of(unsigned clv) const { return clv < tips ? clv : pseudo_row.at(clv); }

  unsigned compute_need(unsigned i) {
    const rdamd_operation_t &o = ops[i];
    unsigned n1 = 0, n2 = 0;
    const bool i1 = is_inner(o.child1_clv_index), i2 = is_inner(o.child2_clv_index);
    if (i1) n1 = compute_need(producer.at(o.child1_clv_index));
    if (i2) n2 = compute_need(producer.at(o.child2_clv_index));
    unsigned r;
    if (i1 && i2) r = std::max(std::max(n1, n2), std::min(n1, n2) + 1);
    else r = i1 ? n1 : (i2 ? n2 : 0);
    need[i] = r;
    return r;
  }

  // park_mat: the matrix the CURRENTLY running CLV will meet at its parent if it
  // has to be parked while this subtree is evaluated
  // park_cls: where the CURRENTLY running CLV goes if it has to be parked (0: its level decides)
  void emit(unsigned i, bool live, unsigned park_mat, unsigned park_cls = 0) {
    const rdamd_operation_t &o = ops[i];
    const bool i1 = is_inner(o.child1_clv_index), i2 = is_inner(o.child2_clv_index);
    FusedOp f;
    memset(&f, 0, sizeof(f));
    unsigned matM = 0, matX = 0, matY = 0, kind = 0, spill = 0, tipX_row = 0, tipY_row = 0, placed = 0;
    if (!i1 && !i2) {
      kind = kFusedTT;
      spill = live ? 1 : 0;
      tipX_row = row_of(o.child1_clv_index); matX = o.child1_matrix_index;
      tipY_row = row_of(o.child2_clv_index); matY = o.child2_matrix_index;
      if (live) {
        matM = park_mat;              // pre-multiply the parked CLV
        if (park_cls) {               // a park with a place of its own
          if (park_cls == 1) spill |= 2;
          else if (park_cls == 2) placed = 0x20000u;
          else mem_depth = std::max(mem_depth, ++mem_now);
        } else if (reg_levels >= 2) { // levels 0 and 1 are register slots in the kernel
          if (depth == 0) spill |= 2;
          else if (depth == 1) spill |= 8;
        } else if (depth == reg_level) {
          spill |= 2;                 // the one register slot
        }
        ++parks_at[depth < 15 ? depth : 15];
        ++depth;
        max_depth = std::max(max_depth, depth);
        if (split_park) {
          FusedOp park;
          memset(&park, 0, sizeof(park));
          park.pM = matM * unit;
          park.flags = kFusedPark | ((spill & 2) ? 0x200u : 0u);   // 0x200: into the register slot
          out.push_back(park);
          matM = 0;
          spill = 0;
        }
      }
    } else if (i1 != i2) {
      const bool first_inner = i1;
      emit(producer.at(first_inner ? o.child1_clv_index : o.child2_clv_index), live, park_mat, park_cls);
      kind = kFusedRT;
      matM = first_inner ? o.child1_matrix_index : o.child2_matrix_index;
      tipY_row = row_of(first_inner ? o.child2_clv_index : o.child1_clv_index);
      matY = first_inner ? o.child2_matrix_index : o.child1_matrix_index;
    } else {
      const unsigned a = producer.at(o.child1_clv_index), b = producer.at(o.child2_clv_index);
      const bool a_first = need[a] >= need[b];
      const unsigned first = a_first ? a : b, second = a_first ? b : a;
      const unsigned mat_first = a_first ? o.child1_matrix_index : o.child2_matrix_index;
      const unsigned mine = park_class.empty() ? 0u : park_class[i];
      emit(first, live, park_mat, park_cls);   // parked (times mat_first) by the first TT op of `second`
      emit(second, true, mat_first, mine);
      kind = kFusedRP;                // running CLV = second; popped = mat_first . first
      matM = a_first ? o.child2_matrix_index : o.child1_matrix_index;
      --depth;
      if (mine) {
        if (mine == 1) spill |= 4;
        else if (mine == 2) placed = 0x40000u;
        else --mem_now;
      } else if (reg_levels >= 2) {
        if (depth == 0) spill |= 4;   // the popped sibling sits in a register slot
        else if (depth == 1) spill |= 16;
      } else if (depth == reg_level) {
        spill |= 4;
      }
    }
    matvecs += (i1 ? 1u : 0u) + (i2 ? 1u : 0u);
    f.pM = matM * unit;
    f.tX = matX * unit;
    f.tY = matY * unit;
    // a leaf whose table has 64 rows: its own slot, flagged for the kernel (0x2000 X, 0x4000 Y)
    unsigned wide_flags = 0;
    if (kind == kFusedTT || kind == kFusedRT) {
      const unsigned leafX = o.child1_clv_index;
      const unsigned leafY = kind == kFusedTT ? o.child2_clv_index : (i1 ? o.child2_clv_index : o.child1_clv_index);
      if (kind == kFusedTT && pseudo_wide.count(leafX)) {
        f.tX = wide_base + pseudo_wide.at(leafX) * rate_cats * 512u;
        wide_flags |= 0x2000u;
      }
      if (pseudo_wide.count(leafY)) {
        f.tY = wide_base + pseudo_wide.at(leafY) * rate_cats * 512u;
        wide_flags |= 0x4000u;
      }
    }
    // 20 states: the tip tables' byte offsets ([matrix][rate 0], 12288 B per (matrix, rate))
    f.pad[0] = matX * rate_cats * (kFused20TabDoubles * 8u);
    f.pad[1] = matY * rate_cats * (kFused20TabDoubles * 8u);
    if (dma_offsets) {
      f.pad[0] = f.tX * 4u + kFusedDmaBias;
      f.pad[1] = f.tY * 4u + kFusedDmaBias - kFusedDmaYSlot;
    }
    f.cX = tipX_row * tip_stride;
    f.cY = tipY_row * tip_stride;
    f.flags = kind | (spill << 8) | wide_flags | placed;   // (a 20-state TT never parks: its spill bits were moved to the park step)
    if (o.parent_clv_index == mark_clv[0]) f.flags |= 0x8000u;
    if (o.parent_clv_index == mark_clv[1]) f.flags |= 0x10000u;
    out.push_back(f);
  }
  // place_parks: park by park.  The parks that are live together nest (a stack), so a set of them
  // fits two single slots exactly when no three of it are live at once: the largest such set is a
  // tree recursion over the operations -- below an inner-inner node the first child sees the
  // slots its parent sees, the second child those the node's own park leaves -- and the level
  // rule is one of its candidates.  c5's plain programs: 250 parks, 171 on the two busiest levels,
  // 237 placed this way; 125.phy 33 / 26 / 31 (round 5).  The register slot counts a little more
  // than the LDS slot.  What is left forms a stack of its own in the private segment.
  int place_value(unsigned i, unsigned free_slots) {
    int &memo = place_memo[4 * i + free_slots];
    if (memo >= 0) return memo;
    const rdamd_operation_t &o = ops[i];
    const bool i1 = is_inner(o.child1_clv_index), i2 = is_inner(o.child2_clv_index);
    int v = 0;
    if (i1 != i2) {
      v = place_value(producer.at(i1 ? o.child1_clv_index : o.child2_clv_index), free_slots);
    } else if (i1 && i2) {
      const unsigned a = producer.at(o.child1_clv_index), b = producer.at(o.child2_clv_index);
      const bool a_first = need[a] >= need[b];
      const unsigned first = a_first ? a : b, second = a_first ? b : a;
      const int base = place_value(first, free_slots);
      v = base + place_value(second, free_slots);
      if (free_slots & 1u) v = std::max(v, base + 1025 + place_value(second, free_slots & ~1u));
      if (free_slots & 2u) v = std::max(v, base + 1024 + place_value(second, free_slots & ~2u));
    }
    return memo = v;
  }
  void place_assign(unsigned i, unsigned free_slots) {
    const rdamd_operation_t &o = ops[i];
    const bool i1 = is_inner(o.child1_clv_index), i2 = is_inner(o.child2_clv_index);
    if (i1 != i2) {
      place_assign(producer.at(i1 ? o.child1_clv_index : o.child2_clv_index), free_slots);
    } else if (i1 && i2) {
      const unsigned a = producer.at(o.child1_clv_index), b = producer.at(o.child2_clv_index);
      const bool a_first = need[a] >= need[b];
      const unsigned first = a_first ? a : b, second = a_first ? b : a;
      const int base = place_value(first, free_slots), v = place_value(i, free_slots);
      unsigned cls = 3, left = free_slots;
      if ((free_slots & 1u) && v == base + 1025 + place_value(second, free_slots & ~1u)) { cls = 1; left = free_slots & ~1u; }
      else if ((free_slots & 2u) && v == base + 1024 + place_value(second, free_slots & ~2u)) { cls = 2; left = free_slots & ~2u; }
      park_class[i] = (unsigned char)cls;
      place_assign(first, free_slots);
      place_assign(second, left);
    }
  }
  // Second pass.  By LEVEL (20 states; 4 states without private-segment levels): the first pass
  // (emit) counted the parks per level, and in a big tree the BOTTOM of the stack is the quiet end
  // (an entry parked near the root waits for half the traversal; the churn is two or three levels
  // up: c5's plain programs park 24 / 79 / 112 / 36 times on levels 0 - 3), so the register slot
  // goes to the busiest level.  What is parked in memory is a sub-sequence of a stack, i.e. a
  // stack: the kernel's count of in-memory entries addresses it whichever level sits in the
  // register.  4 states (`two_reg_beyond` > 0): a program with more levels than that is compiled
  // for TWO register levels (0 and 1) and an all-LDS stack instead (the caller passes
  // 1 + kFusedSpillLevels -- a balanced tree of more than 256 taxa -- for the kernels that have
  // private-segment levels, 3 for those that do not).  Park by park (`place_parks`, above) where
  // the kernel has its one LDS slot and the private segment; `mem_limit`: private-segment entries
  // a wave has room for.  Returns the rank of the runner-up level among the in-memory levels
  // (rounds 3 - 4: FusedJob::lds_pos; no kernel reads it any more).
  unsigned place_levels(unsigned two_reg_beyond, unsigned mem_limit = 0) {
    const bool two_reg = two_reg_beyond > 0 && max_depth > two_reg_beyond;
    if (!two_reg && max_depth < 2) return 0;
    bool again = false;
    if (place_parks && !two_reg) {
      place_memo.assign((size_t)4 * n_ops, -1);
      park_class.assign(n_ops, 0);
      place_value(n_ops - 1, 3u);
      place_assign(n_ops - 1, 3u);
      out.clear();
      depth = max_depth = mem_now = mem_depth = 0;
      matvecs = 0;
      memset(parks_at, 0, sizeof parks_at);
      emit(n_ops - 1, false, 0);
      if (mem_depth <= mem_limit) return 0;
      // (a chain of parks nobody placed, longer than the private segment has room for: the level rule
      // serves -- the kernel then finds no LDS flags and keeps every in-memory entry in the private segment)
      park_class.clear();
      mem_now = mem_depth = 0;
      again = true;
    }
    unsigned busiest = 0;
    for (unsigned l = 1; l < max_depth && l < 16; ++l)
      if (parks_at[l] > parks_at[busiest]) busiest = l;
    unsigned second = busiest == 0 ? 1 : 0;
    for (unsigned l = 0; l < max_depth && l < 16; ++l)
      if (l != busiest && parks_at[l] > parks_at[second]) second = l;
    if (two_reg || busiest != 0 || again) {
      out.clear();
      depth = max_depth = 0;
      matvecs = 0;
      memset(parks_at, 0, sizeof parks_at);
      if (two_reg) reg_levels = 2;
      else reg_level = busiest;
      emit(n_ops - 1, false, 0);
    }
    return two_reg ? 0u : second - (second > busiest ? 1u : 0u);
  }
};

}  // namespace rdamd
