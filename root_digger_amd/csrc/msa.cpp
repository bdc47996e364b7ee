// Alignment ingest: PHYLIP (interleaved or sequential)
// and FASTA readers plus site-pattern compression, the subset of
// /root/reference/src/msa.cpp that feeds the likelihood path
// (parse_msa_file :19-60, msa_t::compress :621-633, constiency_check :641-668,
// valid_data :670-686).  Partition files and model strings are not part of it.
#include <algorithm>
#include <cctype>
#include <fstream>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <unordered_set>

#include "model.hpp"

namespace rdamd {

namespace {
std::string squeeze(const std::string &s) {
  std::string o;
  for (char c : s)
    if (!isspace((unsigned char)c)) o.push_back(c);
  return o;
}

bool parse_fasta(std::istream &in, msa_t &m) {
  std::string line;
  bool any = false;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    if (line[0] == '>') {
      std::istringstream ls(line.substr(1));
      std::string name;
      ls >> name;
      m.labels.push_back(name);
      m.sequences.emplace_back();
      any = true;
    } else {
      if (!any) return false;
      m.sequences.back() += squeeze(line);
    }
  }
  return any;
}

bool parse_phylip(std::istream &in, msa_t &m) {
  std::string header;
  while (std::getline(in, header))
    if (!squeeze(header).empty()) break;
  std::istringstream hs(header);
  long n = 0, len = 0;
  if (!(hs >> n >> len) || n <= 0 || len <= 0) return false;
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(in, line))
    if (!squeeze(line).empty()) lines.push_back(line);
  if ((long)lines.size() < n) return false;
  auto split_named = [](const std::string &l, std::string &name, std::string &rest) {
    std::istringstream ls(l);
    ls >> name;
    std::string tail;
    std::getline(ls, tail);
    rest = squeeze(tail);
  };
  // interleaved (also covers one-line sequential): n named lines, then blocks of n
  {
    msa_t t;
    bool ok = true;
    for (long i = 0; i < n; ++i) {
      std::string name, rest;
      split_named(lines[(size_t)i], name, rest);
      t.labels.push_back(name);
      t.sequences.push_back(rest);
    }
    for (size_t k = (size_t)n; k < lines.size(); ++k) t.sequences[k % (size_t)n] += squeeze(lines[k]);
    for (auto &s : t.sequences) ok = ok && (long)s.size() == len;
    if (ok) { m.labels = t.labels; m.sequences = t.sequences; return true; }
  }
  // sequential: a named line followed by continuation lines until `len`
  {
    msa_t t;
    size_t k = 0;
    for (long i = 0; i < n; ++i) {
      if (k >= lines.size()) return false;
      std::string name, rest;
      split_named(lines[k++], name, rest);
      while ((long)rest.size() < len && k < lines.size()) rest += squeeze(lines[k++]);
      if ((long)rest.size() != len) return false;
      t.labels.push_back(name);
      t.sequences.push_back(rest);
    }
    m.labels = t.labels; m.sequences = t.sequences;
    return true;
  }
}
}  // namespace

// msa_t(filename, map, states, compress), src/msa.hpp:23-37
msa_t msa_t::from_file(const std::string &filename, const uint64_t *map, unsigned int states,
                       bool compress_patterns) {
  msa_t m;
  m.set_map(map);
  m.states = states;
  {
    std::ifstream in(filename);
    if (!in) throw std::invalid_argument("Could not open the MSA file " + filename);
    // the reference tries PHYLIP first, then FASTA (src/msa.cpp:19-60)
    if (!parse_phylip(in, m)) {
      std::ifstream in2(filename);
      m.labels.clear(); m.sequences.clear();
      if (!parse_fasta(in2, m)) throw std::invalid_argument("Could not parse the MSA file " + filename);
    }
  }
  const size_t len = m.length();
  for (auto &s : m.sequences)
    if (s.size() != len) throw std::invalid_argument("MSA sequences differ in length");
  m.valid_data();
  if (compress_patterns) m.compress();
  else m.weights.assign(len, 1u);
  return m;
}

// corax_compress_site_patterns as used at src/msa.cpp:621-633: characters with
// the same state set are merged, identical columns collapse into one pattern
// with a weight; patterns come out in sorted column order.
void msa_t::compress() {
  const size_t n = sequences.size(), len = length();
  if (!n || !len) return;
  // canonical character per state mask
  char canon[256];
  for (int c = 0; c < 256; ++c) canon[c] = (char)c;
  for (int c = 0; c < 256; ++c) {
    if (!map[c]) continue;
    for (int d = 0; d < c; ++d)
      if (map[d] == map[c]) { canon[c] = (char)d; break; }
  }
  std::vector<std::string> cols(len, std::string(n, ' '));
  for (size_t i = 0; i < n; ++i)
    for (size_t s = 0; s < len; ++s) cols[s][i] = canon[(unsigned char)sequences[i][s]];
  std::vector<size_t> order(len);
  std::iota(order.begin(), order.end(), 0);
  const std::vector<unsigned int> old_w = weights.size() == len ? weights : std::vector<unsigned int>(len, 1u);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cols[a] < cols[b]; });
  std::vector<std::string> out(n);
  std::vector<unsigned int> w;
  for (size_t k = 0; k < len; ++k) {
    const size_t s = order[k];
    if (k && cols[s] == cols[order[k - 1]]) {
      w.back() += old_w[s];
    } else {
      for (size_t i = 0; i < n; ++i) out[i].push_back(cols[s][i]);
      w.push_back(old_w[s]);
    }
  }
  sequences = out;
  weights = w;
}

// src/msa.cpp:641-668
msa_t msa_t::columns(size_t lo, size_t hi) const {
  for (auto w : weights)
    if (w != 1u) throw std::runtime_error("msa_t::columns: the alignment is already compressed");
  if (lo > hi || hi > length()) throw std::out_of_range("msa_t::columns: block outside the alignment");
  msa_t out;
  out.labels = labels;
  out.states = states;
  out.map = map;
  out.map_store = map_store;
  for (const auto &s : sequences) out.sequences.push_back(s.substr(lo, hi - lo));
  if (!weights.empty()) out.weights.assign(hi - lo, 1u);
  return out;
}

bool msa_t::constiency_check(const std::unordered_set<std::string> &tree_labels) const {
  std::unordered_set<std::string> taxa(labels.begin(), labels.end());
  for (const auto &k : tree_labels)
    if (!taxa.count(k)) return false;
  for (const auto &k : taxa)
    if (!tree_labels.count(k)) return false;
  return true;
}

// src/msa.cpp:670-686
void msa_t::valid_data() const {
  for (size_t i = 0; i < sequences.size(); ++i)
    for (size_t j = 0; j < sequences[i].size(); ++j) {
      const char c = sequences[i][j];
      if (c < 0)
        throw std::runtime_error("Encountered an invalid character in sequence " +
                                 std::to_string(i) + " at position " + std::to_string(j) + ".");
      if (map[(unsigned char)c] == 0)
        throw std::runtime_error("Found unrecognized character sequence " + std::to_string(i) +
                                 " position " + std::to_string(j) + ".");
    }
}

unsigned int msa_t::total_weight() const {
  if (weights.empty()) return (unsigned int)length();
  unsigned int t = 0;
  for (auto w : weights) t += w;
  return t;
}

}  // namespace rdamd
