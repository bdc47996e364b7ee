// Host-side check of the 20-state matrix-core layouts (root_digger_amd/csrc/common.hpp):
// the CLV tile index is a bijection, a lane's five operands sit where the traversal
// kernel's three load / store instructions put them, and a tip-table row lists every
// state once in the order the four lanes of a site read it.  No device call.
#include <cstdio>
#include <set>
#include "common.hpp"

int main() {
  using namespace rdamd;
  int bad = 0;
  std::set<unsigned> seen;
  for (unsigned c = 0; c < 16; ++c)
    for (unsigned j = 0; j < 20; ++j) {
      const unsigned e = k20_tile_index(c, j);
      if (e >= 320 || !seen.insert(e).second) { std::printf("tile index clash: site %u state %u -> %u\n", c, j, e); ++bad; }
    }
  // lane = 16 g + c holds operand t = state k20_state_of(g, t): pieces 0 / 1 are 64 x 16 bytes
  // (two doubles per lane), piece 2 is 64 x 8 bytes
  for (unsigned g = 0; g < 4; ++g)
    for (unsigned c = 0; c < 16; ++c)
      for (unsigned t = 0; t < 5; ++t) {
        const unsigned lane = 16 * g + c;
        const unsigned want = t < 4 ? (t / 2) * 128 + lane * 2 + (t & 1) : 256 + lane;
        if (k20_tile_index(c, k20_state_of(g, t)) != want) { std::printf("operand (%u, %u, %u) misplaced\n", g, c, t); ++bad; }
      }
  // every lane group owns five contiguous states
  for (unsigned g = 0; g < 4; ++g) {
    std::set<unsigned> st;
    for (unsigned t = 0; t < 5; ++t) st.insert(k20_state_of(g, t));
    if (st.size() != 5 || *st.begin() != 5 * g || *st.rbegin() != 5 * g + 4) { std::printf("group %u states\n", g); ++bad; }
  }
  // tip-table row: positions 2g, 2g+1 = operands 0, 1 of group g; 8 + 2g, 9 + 2g = operands 2, 3; 16 + g = operand 4
  std::set<unsigned> row;
  for (unsigned w = 0; w < 20; ++w) row.insert(k20_row_state(w));
  if (row.size() != 20) { std::printf("row order is no permutation\n"); ++bad; }
  for (unsigned g = 0; g < 4; ++g)
    for (unsigned t = 0; t < 5; ++t) {
      const unsigned w = t < 4 ? 8 * (t / 2) + 2 * g + (t & 1) : 16 + g;
      if (k20_row_state(w) != k20_state_of(g, t)) { std::printf("row position %u\n", w); ++bad; }
    }
  std::printf(bad ? "FAILED\n" : "k20 layouts OK\n");
  return bad ? 1 : 0;
}
