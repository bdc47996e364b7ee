// The pattern classes of a directed subtree (clades.hpp), as a pure host function: no HIP, so
// that tests/cpp/host_logic_check.cpp can hold it against a brute-force count without a GPU.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace rdamd {

// The classes of a node are the distinct pairs (class of child 0, class of child 1) over the
// sites, numbered in order of first appearance -- the partition of the sites by the pattern
// of the tips below the node, whichever way the subtree is split.  a / b: the children's
// classes per site (values < na / nb; a tip's class is its code).  Returns the number of
// classes and fills cls[site] and cmap[class] = (class of child 0, class of child 1); returns 0
// (and leaves the outputs empty) as soon as there are more than max_classes.
inline unsigned clade_classes(const uint8_t *a, unsigned na, const uint8_t *b, unsigned nb, size_t sites,
                              unsigned max_classes, std::vector<uint8_t> &cls, std::vector<uint8_t> &cmap) {
  std::vector<int> seen((size_t)na * nb, -1);
  cls.assign(sites, 0);
  cmap.clear();
  unsigned count = 0;
  for (size_t s = 0; s < sites; ++s) {
    int &slot = seen[(size_t)a[s] * nb + b[s]];
    if (slot < 0) {
      if (count == max_classes) {
        cls.clear();
        cls.shrink_to_fit();
        cmap.clear();
        return 0;
      }
      slot = (int)count++;
      cmap.push_back(a[s]);
      cmap.push_back(b[s]);
    }
    cls[s] = (uint8_t)slot;
  }
  return count;
}

}  // namespace rdamd
