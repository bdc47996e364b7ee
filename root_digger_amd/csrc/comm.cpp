// RCCL communicator of one site group (include/root_digger_amd.h, "site-sharded
// runs").  The per-block log-likelihoods of a batch are summed over the ranks of
// the group with one ncclAllReduce(f64, sum) queued on the partition's stream
// (north star: "an RCCL all-reduce of per-block log-likelihoods over xGMI").  The
// message is 8 x jobs bytes, i.e. latency-bound: there is nothing to bucket or
// overlap, the lever is the batch size of the launch in front of it.
//
// librccl is opened on first use (dlopen) so that single-GPU users carry no
// dependency on it; only the entry points used here are resolved.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <cstring>
#include <mutex>
#include <string>

#include "common.hpp"

namespace {
struct rccl_api {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

rccl_api *rccl() {
  static rccl_api api;
  static std::once_flag once;
  std::call_once(once, [] {
    // RCCL must sit on the SAME HIP/HSA runtime instance this library runs on.  A
    // process may hold two ROCm installations (e.g. PyTorch's bundled one next to
    // /opt/rocm): look next to the libamdhip64 that is actually serving us first.
    std::string beside;
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDevice), &info) && info.dli_fname) {
      beside = info.dli_fname;
      const size_t slash = beside.rfind('/');
      beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1);
    }
    // absolute candidates (beside the runtime, /opt/rocm) are tried only where the file
    // exists; the bare sonames go through the loader's search path
    const std::string candidates[] = {beside.empty() ? std::string() : beside + "librccl.so.1",
                                      beside.empty() ? std::string() : beside + "librccl.so",
                                      "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const std::string &name : candidates) {
      if (name.empty()) continue;
      if (name[0] == '/' && access(name.c_str(), R_OK) != 0) continue;
      api.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
      dlclose(api.lib);
      api.lib = nullptr;
    }
  });
  if (!api.lib) {
    rdamd::set_error(60, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
    return nullptr;
  }
  return &api;
}

bool ok(rccl_api *a, ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return true;
  rdamd::set_error(61, "%s: %s", what, a->GetErrorString ? a->GetErrorString(r) : "RCCL error");
  return false;
}
}  // namespace

struct rdamd_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
};

static_assert(sizeof(ncclUniqueId) == 128, "rdamd_comm_unique_id hands out 128 bytes");

extern "C" {

int rdamd_comm_unique_id(char id[128]) {
  rdamd::clear_error();
  rccl_api *a = rccl();
  if (!a) return RDAMD_FAILURE;
  ncclUniqueId u;
  if (!ok(a, a->GetUniqueId(&u), "ncclGetUniqueId")) return RDAMD_FAILURE;
  std::memcpy(id, &u, sizeof u);
  return RDAMD_SUCCESS;
}

rdamd_comm_t *rdamd_comm_create(const char id[128], int rank, int n_ranks) {
  rdamd::clear_error();
  rccl_api *a = rccl();
  if (!a) return nullptr;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) {
    rdamd::set_error(62, "rdamd_comm_create: rank %d of %d", rank, n_ranks);
    return nullptr;
  }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  auto *c = new rdamd_comm();
  c->rank = rank;
  c->n_ranks = n_ranks;
  if (!ok(a, a->CommInitRank(&c->comm, n_ranks, u, rank), "ncclCommInitRank")) {
    delete c;
    return nullptr;
  }
  return c;
}

int rdamd_comm_allreduce_sum(rdamd_comm_t *c, double *device_values, unsigned int n, void *stream) {
  rccl_api *a = rccl();
  if (!a || !c) return RDAMD_FAILURE;
  if (n == 0) return RDAMD_SUCCESS;
  return ok(a, a->AllReduce(device_values, device_values, n, ncclDouble, ncclSum, c->comm,
                            (hipStream_t)stream), "ncclAllReduce")
             ? RDAMD_SUCCESS : RDAMD_FAILURE;
}

int rdamd_comm_reducer(double *values, unsigned int n, void *stream, void *user) {
  return rdamd_comm_allreduce_sum((rdamd_comm_t *)user, values, n, stream);
}

void rdamd_comm_destroy(rdamd_comm_t *c) {
  if (!c) return;
  rccl_api *a = rccl();
  if (a && c->comm) (void)a->CommDestroy(c->comm);
  delete c;
}

}  // extern "C"
