// checkpoint_t: see checkpoint.hpp for the file layout and its provenance.
#include "checkpoint.hpp"

#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <stdexcept>
#include <sys/stat.h>
#include <unistd.h>

namespace rdamd {
namespace {

// fcntl locks exclude other PROCESSES; the worker threads of one process
// (rdamd_model_exhaustive_search_parallel) are serialised here
std::mutex g_append_mutex;

constexpr uint32_t kSuccessFlag = 1u << 0;   // CHECKPOINT_WRITE_SUCCESS_FLAG
constexpr uint32_t kModAdler = 65521;

// ---- checksum (src/checkpoint.hpp:33-91) ------------------------------------------
// One fold over raw bytes.  The reference writes `b = b + a % MOD_ADLER`, i.e.
// the second sum is never reduced and wraps at 2^32; kept, the file depends on it.
struct adler_t {
  uint32_t a = 1, b = 0;
  void fold(const void *data, size_t len) {
    const uint8_t *p = static_cast<const uint8_t *>(data);
    for (size_t i = 0; i < len; ++i) {
      a = (a + p[i]) % kModAdler;
      b = b + a % kModAdler;
    }
  }
  uint32_t value() const { return (b << 16) | a; }
};

// The reference folds the four vectors of a parameter set through a variadic
// template whose recursion ends in the GENERIC overload called with
// (running a, running b): that call reads the 4 bytes of `a` as the value,
// starts from a = the running b, and restarts b at 0.  Kept for the same reason.
void fold_parameters(adler_t &ck, const partition_parameters_t &pp) {
  for (const model_params_t *v : {&pp.subst_rates, &pp.freqs, &pp.gamma_alpha, &pp.gamma_weights})
    for (double d : *v) ck.fold(&d, sizeof d);
  const uint32_t value = ck.a;
  ck.a = ck.b;
  ck.b = 0;
  ck.fold(&value, sizeof value);
}

// ---- serialisation ----------------------------------------------------------------
struct sink_t {
  std::vector<uint8_t> bytes;
  template <typename T> void raw(const T &v) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&v);
    bytes.insert(bytes.end(), p, p + sizeof(T));
  }
  void str(const std::string &s) {
    raw<uint64_t>(s.size());
    bytes.insert(bytes.end(), s.begin(), s.end());
  }
  void doubles(const model_params_t &v) {
    raw<uint64_t>(v.size());
    for (double d : v) raw(d);
  }
};

struct short_read : std::runtime_error {
  short_read() : std::runtime_error("checkpoint: the file ends inside a field") {}
};

struct source_t {
  const std::vector<uint8_t> &bytes;
  size_t pos = 0;
  template <typename T> T raw() {
    if (pos + sizeof(T) > bytes.size()) throw short_read{};
    T v;
    std::memcpy(&v, bytes.data() + pos, sizeof(T));
    pos += sizeof(T);
    return v;
  }
  size_t count(size_t element_bytes) {   // a u64 length that must fit in what is left
    const uint64_t n = raw<uint64_t>();
    if (n > (bytes.size() - pos) / (element_bytes ? element_bytes : 1)) throw short_read{};
    return (size_t)n;
  }
  std::string str() {
    const size_t n = count(1);
    std::string s(reinterpret_cast<const char *>(bytes.data() + pos), n);
    pos += n;
    return s;
  }
  model_params_t doubles() {
    const size_t n = count(sizeof(double));
    model_params_t v(n);
    for (auto &d : v) d = raw<double>();
    return v;
  }
  bool done() const { return pos >= bytes.size(); }
};

void put_options(sink_t &out, const cli_options_t &o) {   // src/checkpoint.cpp:60-91
  out.str(o.msa_filename);   out.str(o.tree_filename);  out.str(o.prefix);
  out.str(o.prefix_dir);     out.str(o.model_filename); out.str(o.freqs_filename);
  out.str(o.partition_filename); out.str(o.data_type);  out.str(o.model_string);
  out.raw<uint64_t>(o.rate_cats.size());
  for (const auto &rc : o.rate_cats) {   // raw struct; its padding bytes are written as zeros
    uint8_t image[sizeof(ratehet_opts_t)] = {0};
    std::memcpy(image + 0, &rc.type, 4);
    std::memcpy(image + 4, &rc.rate_category_type, 4);
    std::memcpy(image + 8, &rc.rate_cats, 8);
    image[16] = rc.alpha_init ? 1 : 0;
    std::memcpy(image + 24, &rc.alpha, 8);
    out.bytes.insert(out.bytes.end(), image, image + sizeof image);
  }
  out.raw(o.seed); out.raw(o.min_roots); out.raw(o.threads);
  out.raw(o.root_ratio); out.raw(o.abs_tolerance); out.raw(o.factor);
  out.raw(o.br_tolerance); out.raw(o.bfgs_tol);
  out.raw<uint8_t>(o.silent); out.raw<uint8_t>(o.exhaustive); out.raw<uint8_t>(o.echo);
  out.raw<uint8_t>(o.invariant_sites);
  out.raw(o.early_stop);
  out.raw(o.initial_root_strategy);
  out.raw(kSuccessFlag);
}

cli_options_t get_options(source_t &in) {   // src/checkpoint.cpp:93-124
  cli_options_t o;
  o.msa_filename = in.str();   o.tree_filename = in.str();  o.prefix = in.str();
  o.prefix_dir = in.str();     o.model_filename = in.str(); o.freqs_filename = in.str();
  o.partition_filename = in.str(); o.data_type = in.str();  o.model_string = in.str();
  const size_t n = in.count(sizeof(ratehet_opts_t));
  o.rate_cats.assign(n, ratehet_opts_t{});
  for (auto &rc : o.rate_cats) {
    const size_t at = in.pos;
    rc.type = in.raw<param_type>();
    rc.rate_category_type = in.raw<rate_category>();
    rc.rate_cats = in.raw<uint64_t>();
    rc.alpha_init = in.raw<uint8_t>() != 0;
    in.pos = at + 24;
    rc.alpha = in.raw<double>();
  }
  o.seed = in.raw<uint64_t>(); o.min_roots = in.raw<uint64_t>(); o.threads = in.raw<uint64_t>();
  o.root_ratio = in.raw<double>(); o.abs_tolerance = in.raw<double>(); o.factor = in.raw<double>();
  o.br_tolerance = in.raw<double>(); o.bfgs_tol = in.raw<double>();
  o.silent = in.raw<uint8_t>() != 0; o.exhaustive = in.raw<uint8_t>() != 0;
  o.echo = in.raw<uint8_t>() != 0;   o.invariant_sites = in.raw<uint8_t>() != 0;
  o.early_stop = in.raw<early_stop_t>();
  o.initial_root_strategy = in.raw<initial_root_strategy_t>();
  if (!(in.raw<uint32_t>() & kSuccessFlag))
    throw std::runtime_error("checkpoint: the options header was not written completely");
  return o;
}

void put_record(sink_t &out, const rd_result_t &r, const std::vector<partition_parameters_t> &pps) {
  out.raw<uint64_t>(r.root_id);
  out.raw(r.llh);
  out.raw(r.alpha);
  out.raw(checkpoint_checksum(r));
  out.raw<uint64_t>(pps.size());
  for (const auto &pp : pps) {
    out.doubles(pp.subst_rates);
    out.doubles(pp.freqs);
    out.doubles(pp.gamma_alpha);
    out.doubles(pp.gamma_weights);
  }
  out.raw(checkpoint_checksum(pps));
}

// advisory whole-file lock for the lifetime of the object (fcntl_lock_t,
// src/checkpoint.hpp:191-229; blocking)
class file_lock_t {
public:
  file_lock_t(int fd, short type) : _fd(fd) {
    struct flock fl;
    std::memset(&fl, 0, sizeof fl);
    fl.l_type = type;
    if (fcntl(_fd, F_SETLKW, &fl) == -1) throw std::runtime_error("checkpoint: failed to obtain the lock");
  }
  ~file_lock_t() {
    struct flock fl;
    std::memset(&fl, 0, sizeof fl);
    fl.l_type = F_UNLCK;
    fcntl(_fd, F_SETLK, &fl);
  }
  file_lock_t(const file_lock_t &) = delete;
  file_lock_t &operator=(const file_lock_t &) = delete;

private:
  int _fd;
};

void write_all(int fd, const std::vector<uint8_t> &bytes) {
  size_t done = 0;
  while (done < bytes.size()) {
    const ssize_t n = ::write(fd, bytes.data() + done, bytes.size() - done);
    if (n < 0) {
      if (errno == EINTR) continue;
      throw std::runtime_error("checkpoint: failed to write all data to the file");
    }
    done += (size_t)n;
  }
}

std::vector<uint8_t> read_all(int fd) {
  std::vector<uint8_t> bytes;
  struct stat st;
  if (fstat(fd, &st) == 0 && st.st_size > 0) bytes.reserve((size_t)st.st_size);
  uint8_t buf[1 << 16];
  off_t at = 0;
  for (;;) {
    const ssize_t n = pread(fd, buf, sizeof buf, at);
    if (n < 0) {
      if (errno == EINTR) continue;
      throw std::runtime_error("checkpoint: failed to read the file");
    }
    if (n == 0) break;
    bytes.insert(bytes.end(), buf, buf + n);
    at += n;
  }
  return bytes;
}

}  // namespace

uint32_t checkpoint_checksum(const rd_result_t &r) {
  static_assert(sizeof(rd_result_t) == 24, "rd_result_t is written as its raw bytes");
  adler_t ck;
  ck.fold(&r, sizeof r);
  return ck.value();
}

uint32_t checkpoint_checksum(const std::vector<partition_parameters_t> &pps) {
  adler_t ck;
  for (const auto &pp : pps) fold_parameters(ck, pp);
  return ck.value();
}

checkpoint_t::checkpoint_t(const std::string &prefix) : _checkpoint_filename(prefix + ".ckp") {
  _existing_results = access(_checkpoint_filename.c_str(), F_OK) != -1;
  _file_descriptor = open(_checkpoint_filename.c_str(), O_RDWR | O_APPEND | O_CREAT, 0640);
  if (_file_descriptor == -1) throw std::runtime_error("Failed to open the checkpoint file");
}

checkpoint_t::~checkpoint_t() {
  if (_file_descriptor != -1) close(_file_descriptor);
}

void checkpoint_t::reload() {
  close(_file_descriptor);
  _file_descriptor = open(_checkpoint_filename.c_str(), O_RDWR | O_APPEND | O_CREAT, 0640);
  if (_file_descriptor == -1) throw std::runtime_error("Failed to reload the checkpoint file");
}

int checkpoint_t::get_inode() {
  struct stat st;
  if (fstat(_file_descriptor, &st) == -1)
    throw std::runtime_error("There was an error getting the INODE of the checkpoint");
  return (int)st.st_ino;
}

void checkpoint_t::save_options(const cli_options_t &options) {
  if (_existing_results) return;
  sink_t out;
  put_options(out, options);
  file_lock_t lock(_file_descriptor, F_WRLCK);
  write_all(_file_descriptor, out.bytes);   // O_APPEND: one write, one record
}

void checkpoint_t::load_options(cli_options_t &options) {
  if (!_existing_results) return;
  file_lock_t lock(_file_descriptor, F_WRLCK);
  const auto bytes = read_all(_file_descriptor);
  source_t in{bytes};
  options = get_options(in);
}

void checkpoint_t::write(const rd_result_t &result,
                         const std::vector<partition_parameters_t> &parameters) {
  sink_t out;
  put_record(out, result, parameters);
  std::lock_guard<std::mutex> guard(g_append_mutex);
  file_lock_t lock(_file_descriptor, F_WRLCK);
  write_all(_file_descriptor, out.bytes);
  fsync(_file_descriptor);
}

std::vector<checkpoint_record_t> checkpoint_t::parse(bool &intact, cli_options_t *header) {
  std::vector<uint8_t> bytes;
  {
    file_lock_t lock(_file_descriptor, F_WRLCK);
    bytes = read_all(_file_descriptor);
  }
  std::vector<checkpoint_record_t> records;
  intact = true;
  source_t in{bytes};
  cli_options_t opts = get_options(in);   // a file without a valid header is an error, as in the reference
  if (header) *header = std::move(opts);
  while (!in.done()) {
    try {   // a record that is cut short or fails a checksum ends the usable part
      checkpoint_record_t rec;
      rec.first.root_id = (size_t)in.raw<uint64_t>();
      rec.first.llh = in.raw<double>();
      rec.first.alpha = in.raw<double>();
      if (in.raw<uint32_t>() != checkpoint_checksum(rec.first)) throw short_read{};
      const size_t n = in.count(4 * sizeof(uint64_t));
      rec.second.resize(n);
      for (auto &pp : rec.second) {
        pp.subst_rates = in.doubles();
        pp.freqs = in.doubles();
        pp.gamma_alpha = in.doubles();
        pp.gamma_weights = in.doubles();
      }
      if (in.raw<uint32_t>() != checkpoint_checksum(rec.second)) throw short_read{};
      records.push_back(std::move(rec));
    } catch (const short_read &) {
      intact = false;
      break;
    }
  }
  return records;
}

std::vector<checkpoint_record_t> checkpoint_t::read_results() {
  bool intact;
  return parse(intact, nullptr);
}

bool checkpoint_t::needs_cleaning() {
  bool intact;
  parse(intact, nullptr);
  return !intact;
}

std::vector<rd_result_t> checkpoint_t::current_progress() {
  std::vector<rd_result_t> out;
  for (auto &rec : read_results()) out.push_back(rec.first);
  return out;
}

std::vector<size_t> checkpoint_t::completed_indicies() {
  std::vector<size_t> out;
  for (auto &rec : read_results()) out.push_back(rec.first.root_id);
  return out;
}

void checkpoint_t::clean() {   // src/checkpoint.cpp:168-193
  if (!_existing_results) return;
  bool intact;
  cli_options_t header;
  const auto records = parse(intact, &header);
  sink_t out;
  put_options(out, header);
  for (const auto &rec : records) put_record(out, rec.first, rec.second);
  const std::string backup = _checkpoint_filename + ".bak";
  {
    file_lock_t lock(_file_descriptor, F_WRLCK);
    const int fd = open(backup.c_str(), O_RDWR | O_CREAT | O_APPEND | O_EXCL, 0640);
    if (fd == -1)
      throw std::runtime_error("Failed to open the new checkpoint when cleaning the checkpoint");
    try {
      write_all(fd, out.bytes);
    } catch (...) {
      close(fd);
      unlink(backup.c_str());
      throw;
    }
    fsync(fd);
    close(fd);
    if (rename(backup.c_str(), _checkpoint_filename.c_str()) != 0)
      throw std::runtime_error("Failed to replace the checkpoint when cleaning it");
  }
  reload();   // this descriptor still points at the replaced file
}

}  // namespace rdamd
