// Result log compatible with the reference's `<prefix>.ckp` file
// (/root/reference/src/checkpoint.hpp:231-300, src/checkpoint.cpp): same
// bytes, same method names, so a run can be resumed by either program and
// several processes (one per GPU) can share one log the way the reference's
// MPI ranks do -- every append happens under an fcntl write lock.
//
// File layout (native little-endian, no alignment padding between fields):
//   header   cli_options_t, field by field (src/checkpoint.cpp:60-91), then a
//            u32 success flag (bit 0)
//   records  rd_result_t as its 24 raw bytes + u32 checksum, then
//            vector<partition_parameters_t> + u32 checksum, repeated
//   string   u64 length + bytes;   vector<T>   u64 count + elements
// The checksum is the reference's Adler-32 variant INCLUDING its quirks (the
// second sum is never reduced, and a parameter vector's fold ends with one
// extra step over the running first sum), restated in checkpoint.cpp.
//
// Parity status: the raw-written structs (rd_result_t, ratehet_opts_t), enum
// values and defaults are pinned against the reference's own util.hpp
// (oracle/ref_layout.cpp -> tests/golden/ref_layout.json).  The byte stream as a
// whole is UNPINNED: the reference's checkpoint code cannot be built here (it
// includes tree.hpp -> coraxlib) and its tests hold no golden file;
// tests/test_checkpoint.py checks it against an independent restatement
// (oracle/ckp_oracle.py) and hand-assembled bytes.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "model.hpp"

namespace rdamd {

enum class initial_root_strategy_t : int32_t { random, midpoint, modified_mad };   // src/util.hpp:74-78
enum class early_stop_t : int32_t { uninitalized, initialized_true, initialized_false };   // :115-121

struct cli_options_t {   // the serialised subset of src/util.hpp:149-177, in file order
  std::string msa_filename, tree_filename, prefix, prefix_dir, model_filename, freqs_filename,
      partition_filename, data_type, model_string;
  std::vector<ratehet_opts_t> rate_cats{ratehet_opts_t{}};
  uint64_t seed = 0;
  uint64_t min_roots = 1, threads = 0;
  double root_ratio = 0.01, abs_tolerance = 1e-7, factor = 1e4, br_tolerance = 1e-12,
         bfgs_tol = 1e-7;
  bool silent = false, exhaustive = false, echo = false, invariant_sites = false;
  early_stop_t early_stop = early_stop_t::uninitalized;
  initial_root_strategy_t initial_root_strategy = initial_root_strategy_t::modified_mad;
};

using checkpoint_record_t = std::pair<rd_result_t, std::vector<partition_parameters_t>>;

uint32_t checkpoint_checksum(const rd_result_t &);
uint32_t checkpoint_checksum(const std::vector<partition_parameters_t> &);

class checkpoint_t {
public:
  explicit checkpoint_t(const std::string &prefix);   // opens/creates <prefix>.ckp
  ~checkpoint_t();
  checkpoint_t(const checkpoint_t &) = delete;
  checkpoint_t &operator=(const checkpoint_t &) = delete;

  bool existing_checkpoint() const { return _existing_results; }
  std::string get_filename() const { return _checkpoint_filename; }
  int  get_inode();
  void reload();

  void save_options(const cli_options_t &);   // only into a new file (src/checkpoint.cpp:212-217)
  void load_options(cli_options_t &);         // only from an existing one (:219-230)
  void write(const rd_result_t &, const std::vector<partition_parameters_t> &);

  std::vector<checkpoint_record_t> read_results();
  std::vector<rd_result_t> current_progress();
  std::vector<size_t>      completed_indicies();
  bool needs_cleaning();   // a truncated or corrupt tail
  void clean();            // rewrite the file with the header and the intact records

private:
  // parses the file; `intact` = every byte belonged to a well-formed record
  std::vector<checkpoint_record_t> parse(bool &intact, cli_options_t *header);

  std::string _checkpoint_filename;
  int  _file_descriptor = -1;
  bool _existing_results = false;
};

}  // namespace rdamd
