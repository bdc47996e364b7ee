// Partition file / model string front end (the reference's src/msa.cpp:91-522,
// src/util.hpp:37-108): which columns of the alignment form each partition and
// which model options (+F.., +I.., +G.., +R.., +ASC_..) its model string asks
// for.  Grammar of one line:
//   <MODEL> , <NAME> = <BEGIN>[-<END>] [, <BEGIN>[-<END>]]*       (1-based, inclusive)
//   <MODEL> = <SUBST> [+ <OPTION>]*
// Pinned by the known answers of the reference's test/src/msa.cpp:17-245
// (tests/test_partition_info.py).
#pragma once

#include <string>
#include <utility>
#include <vector>

#include "model.hpp"

namespace rdamd {

struct freq_opts_t {   // src/util.hpp:39-41
  param_type type = param_type::emperical;
};
struct invar_opts_t {   // :43-46
  param_type type = param_type::estimate;
  float      user_prop = 0.0f;
  bool       present = false;   // "+I" appeared at all (not in the reference: its struct is left uninitialised)
};
enum class asc_bias_type { lewis, fels, stam };   // :72
struct asc_bias_opts_t {   // :80-84
  asc_bias_type       type = asc_bias_type::lewis;
  double              fels_weight = 0.0;
  std::vector<double> stam_weights;
  bool                present = false;
};
struct model_info_t {   // :86-93
  size_t          states = 4;
  std::string     subst_str;
  freq_opts_t     freq_opts;
  invar_opts_t    invar_opts;
  ratehet_opts_t  ratehet_opts{0};   // rate_cats == 0: no rate heterogeneity asked for
  asc_bias_opts_t asc_opts;
};
struct partition_info_t {   // :95-100
  std::vector<std::pair<size_t, size_t>> parts;
  std::string  model_name, partition_name;
  model_info_t model;
};
using msa_partitions_t = std::vector<partition_info_t>;

model_info_t     parse_model_info(const std::string &model_string);   // src/msa.cpp:364-415
partition_info_t parse_partition_info(const std::string &line);       // :417-506
msa_partitions_t parse_partition_file(const std::string &filename);   // :512-522

// msa_t::partition (src/msa.cpp:522-591, :635-639): one alignment per
// partition, made of its column ranges, each compressed to site patterns
std::vector<msa_t> partition_msa(const msa_t &whole, const msa_partitions_t &parts,
                                 bool compress_patterns = true);

}  // namespace rdamd
