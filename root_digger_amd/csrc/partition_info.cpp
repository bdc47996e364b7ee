// See partition_info.hpp.
#include "partition_info.hpp"

#include <cctype>
#include <fstream>
#include <stdexcept>

namespace rdamd {
namespace {

// A cursor over one line.  peek() past the end is '\0', so no scan can run off.
class cursor_t {
public:
  explicit cursor_t(const std::string &text) : _s(text) {}
  char peek() const { return _at < _s.size() ? _s[_at] : '\0'; }
  char lower() const { return (char)std::tolower((unsigned char)peek()); }
  bool done() const { return _at >= _s.size(); }
  void advance() { if (!done()) ++_at; }
  void skip_space() { while (std::isspace((unsigned char)peek())) ++_at; }

  // the next non-blank character must be c (case-insensitive); blanks after it are eaten
  void expect(char c) {
    skip_space();
    if (std::tolower((unsigned char)c) != lower())
      throw std::runtime_error(std::string("Failed to parse partition file, expected '") + c +
                               "' got '" + peek() + "' instead");
    ++_at;
    skip_space();
  }
  template <typename Pred> std::string take_while(Pred &&keep) {
    const size_t from = _at;
    while (!done() && keep(peek())) ++_at;
    return _s.substr(from, _at - from);
  }
  // everything up to and including the next `c`
  void skip_past(char c) {
    while (!done() && peek() != c) ++_at;
    if (done()) throw std::runtime_error("Could not skip, found end of string intead");
    ++_at;
  }
  size_t integer(const char *what) {
    const std::string digits = take_while([](char ch) { return std::isdigit((unsigned char)ch) != 0; });
    if (digits.empty()) throw std::runtime_error(std::string("Expected ") + what + ", but found something else");
    return (size_t)std::stoull(digits);
  }
  // digits, then any mix of '.', 'e', '+', '-' groups followed by digits (the
  // reference's loose float syntax, src/msa.cpp:113-147)
  double real() {
    const size_t from = _at;
    if (!std::isdigit((unsigned char)peek()))
      throw std::runtime_error("Expected float, but found something else");
    while (std::isdigit((unsigned char)peek())) ++_at;
    for (;;) {
      const char c = peek();
      if (c != '.' && c != 'e' && c != '+' && c != '-') break;
      ++_at;
      const size_t digits_from = _at;
      while (std::isdigit((unsigned char)peek())) ++_at;
      if (c != '.' && digits_from == _at)
        throw std::runtime_error("Encountered a malformed floating point number");
    }
    return std::stod(_s.substr(from, _at - from));
  }

private:
  const std::string &_s;
  size_t _at = 0;
};

bool word_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == ':'; }

freq_opts_t freq_option(cursor_t &in) {   // +F, +FC, +FO, +FE, +FU{a/b/c/d}
  freq_opts_t f;
  in.advance();
  switch (in.lower()) {
    case 'c': f.type = param_type::emperical; in.advance(); break;
    case 'o': f.type = param_type::estimate; in.advance(); break;
    case 'e': f.type = param_type::equal; in.advance(); break;
    case 'u': f.type = param_type::user; in.skip_past('}'); break;   // the values are not used
    default: f.type = param_type::emperical; break;
  }
  in.skip_space();
  return f;
}

invar_opts_t invar_option(cursor_t &in) {   // +I, +IO, +IC, +IU{p}
  invar_opts_t v;
  v.present = true;
  in.advance();
  switch (in.lower()) {
    case 'o': v.type = param_type::estimate; in.advance(); break;
    case 'c': v.type = param_type::emperical; in.advance(); break;
    case 'u':
      in.advance();
      v.type = param_type::user;
      in.expect('{');
      v.user_prop = (float)in.real();
      in.expect('}');
      break;
    default: v.type = param_type::estimate; break;
  }
  in.skip_space();
  return v;
}

ratehet_opts_t gamma_option(cursor_t &in) {   // +G, +Gn, +Gn{alpha}, +GA
  ratehet_opts_t g(4);
  g.type = param_type::estimate;
  g.rate_category_type = rate_category::MEAN;
  in.advance();
  if (in.lower() == 'a') {
    g.rate_category_type = rate_category::MEDIAN;
    in.advance();
  } else if (std::isdigit((unsigned char)in.peek())) {
    g.rate_cats = in.integer("a number");
    if (in.peek() == '{') {
      in.advance();
      g.alpha = in.real();
      g.alpha_init = true;
      g.type = param_type::user;
      in.expect('}');
    }
  }
  in.skip_space();
  return g;
}

ratehet_opts_t free_rates_option(cursor_t &in) {   // +Rn, +Rn{rates}{weights} (values ignored)
  ratehet_opts_t r(0);
  r.type = param_type::estimate;
  r.rate_category_type = rate_category::FREE;
  in.advance();
  r.rate_cats = in.integer("a number");
  if (in.peek() == '{') {
    in.skip_past('}');
    in.skip_past('}');
  }
  in.skip_space();
  return r;
}

asc_bias_opts_t asc_option(cursor_t &in) {   // +ASC_L, +ASC_F{w}, +ASC_S{w/w/...}
  asc_bias_opts_t a;
  a.present = true;
  in.advance();   // 'A'
  in.expect('S');
  in.expect('C');
  in.expect('_');
  switch (in.lower()) {
    case 'l': a.type = asc_bias_type::lewis; in.advance(); break;
    case 'f':
      a.type = asc_bias_type::fels;
      in.advance();
      in.expect('{');
      a.fels_weight = in.real();
      in.expect('}');
      break;
    case 's':
      a.type = asc_bias_type::stam;
      in.advance();
      in.expect('{');
      for (;;) {
        a.stam_weights.push_back(in.real());
        if (in.peek() == '}') { in.advance(); break; }
        in.expect('/');
      }
      break;
    default: break;
  }
  in.skip_space();
  return a;
}

}  // namespace

model_info_t parse_model_info(const std::string &model_string) {
  model_info_t info;
  cursor_t in(model_string);
  info.subst_str = in.take_while(word_char);
  if (info.subst_str.empty()) throw std::runtime_error("Failed to find a word when scanning");
  while (!in.done()) {
    in.expect('+');
    switch (in.lower()) {
      case 'f': info.freq_opts = freq_option(in); break;
      case 'i': info.invar_opts = invar_option(in); break;
      case 'g': info.ratehet_opts = gamma_option(in); break;
      case 'r': info.ratehet_opts = free_rates_option(in); break;
      case 'a': info.asc_opts = asc_option(in); break;
      case 'm': in.advance(); break;   // "+M": not supported by root digger, ignored
      default:
        throw std::runtime_error(std::string("Unknown option '") + in.peek() + "' in model string " +
                                 model_string);
    }
  }
  return info;
}

partition_info_t parse_partition_info(const std::string &line) {
  partition_info_t pi;
  cursor_t in(line);
  in.skip_space();
  pi.model_name = in.take_while([](char c) {
    return std::isalnum((unsigned char)c) || c == '+' || c == '{' || c == '}' || c == '/' || c == '.' ||
           c == '_';
  });
  if (pi.model_name.empty()) throw std::runtime_error("Error, partition is missing a model name");
  pi.model = parse_model_info(pi.model_name);
  in.expect(',');
  pi.partition_name = in.take_while([](char c) { return std::isalnum((unsigned char)c) || c == '_'; });
  in.expect('=');
  for (;;) {
    in.skip_space();
    const size_t begin = in.integer("a partition range");
    size_t end = begin;
    in.skip_space();
    if (in.peek() != ',') {   // "<BEGIN>," alone is a one-column range; anything else needs "-<END>"
      in.expect('-');
      end = in.integer("the end of a partition range");
      if (end < begin)
        throw std::runtime_error("The end index of the partition '" + pi.partition_name +
                                 "' comes before the beginning");
    }
    pi.parts.emplace_back(begin, end);
    in.skip_space();
    if (in.peek() != ',') break;
    in.advance();
  }
  return pi;   // (whatever follows the last range is ignored, as in the reference)
}

msa_partitions_t parse_partition_file(const std::string &filename) {
  std::ifstream file(filename);
  if (!file) throw std::runtime_error("Failed to open the partition file");
  msa_partitions_t parts;
  for (std::string line; std::getline(file, line);) {
    bool blank = true;
    for (char c : line) blank = blank && std::isspace((unsigned char)c);
    if (blank) continue;
    parts.push_back(parse_partition_info(line));
  }
  return parts;
}

std::vector<msa_t> partition_msa(const msa_t &whole, const msa_partitions_t &parts,
                                 bool compress_patterns) {
  for (unsigned int w : whole.weights)
    if (w != 1) throw std::runtime_error("partition_msa: partition the alignment before compressing it");
  std::vector<msa_t> out;
  for (const auto &pi : parts) {
    msa_t m;
    m.labels = whole.labels;
    m.states = whole.states;
    m.map = whole.map;
    m.map_store = whole.map_store;
    m.sequences.assign(whole.sequences.size(), std::string());
    for (const auto &range : pi.parts) {
      if (range.first == 0)
        throw std::runtime_error("Partition ranges start at 1, but we encountered a 0");
      if (range.second > whole.length())
        throw std::runtime_error("Partition '" + pi.partition_name + "' reaches past the end of the alignment");
      for (size_t t = 0; t < whole.sequences.size(); ++t)
        m.sequences[t].append(whole.sequences[t], range.first - 1, range.second - range.first + 1);
    }
    if (compress_patterns) m.compress();
    out.push_back(std::move(m));
  }
  return out;
}

}  // namespace rdamd
