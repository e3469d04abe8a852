// node.cpp -- one node, several GPUs, behind the C boundary (include/sipnet_amd.h, sipnet_node_*).
//
// The reference runs one process per ensemble member (frontend.c:212-250 is its whole host side);
// the north star asks for the ensemble axis sharded over the GPUs of one node with ONE RCCL
// all-gather over xGMI of the output block, issued from the C host.  A sipnet_node is that host
// object: one sipnet_batch, one HIP stream and one RCCL communicator rank per device, all in one
// process (ncclCommInitAll; the per-device calls of a collective are fused with ncclGroupStart /
// ncclGroupEnd, RCCL's single-process multi-GPU idiom), members sharded contiguously.  The
// forward model has no coupling between members, so the step kernels never exchange anything; the
// collective is the all-gather of what the devices computed:
//   sipnet_node_gather_stats   the per-(variable, step, site) sum / sum of squares block of every
//                              device (0.84 MB per device and year) -- the default exchange, after
//                              which every device and the host hold the ensemble statistics;
//   sipnet_node_gather_planes  the north star's exchange as written: the member-resolved NEE / GPP /
//                              ET planes of every device on every device.
// RCCL is loaded on first use (dlopen of librccl.so.1: a process that already holds one -- PyTorch
// ships its own -- keeps using that one; a single-GPU batch never pays for the 570 MB library).
// Without a usable RCCL sipnet_node_create fails loudly; there is no fallback path.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sipnet_amd.h"

namespace sipnet {
void setError(const std::string& s);
}
using sipnet::setError;

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) commInitAll = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclAllGather) allGather = nullptr;
  decltype(&ncclGroupStart) groupStart = nullptr;
  decltype(&ncclGroupEnd) groupEnd = nullptr;
  decltype(&ncclGetErrorString) errorString = nullptr;
  decltype(&ncclGetVersion) getVersion = nullptr;
  std::string path;
};

Rccl* loadRccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) {
      r.path = n;
      break;
    }
  }
  if (!r.handle) return nullptr;
#define RCCL_SYM(field, name)                                         \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name)); \
  if (!r.field) {                                                     \
    r.handle = nullptr;                                               \
    return nullptr;                                                   \
  }
  RCCL_SYM(commInitAll, "ncclCommInitAll")
  RCCL_SYM(commDestroy, "ncclCommDestroy")
  RCCL_SYM(allGather, "ncclAllGather")
  RCCL_SYM(groupStart, "ncclGroupStart")
  RCCL_SYM(groupEnd, "ncclGroupEnd")
  RCCL_SYM(errorString, "ncclGetErrorString")
  RCCL_SYM(getVersion, "ncclGetVersion")
#undef RCCL_SYM
  return &r;
}

}  // namespace

struct sipnet_node {
  int32_t flags[SIPNET_NFLAGS];
  int32_t n_sites = 0, n_members = 0, precision = 0;
  std::vector<int32_t> devices, first, count;
  std::vector<sipnet_batch*> batches;
  std::vector<hipStream_t> streams;
  std::vector<ncclComm_t> comms;
  Rccl* rccl = nullptr;
  int32_t maxCount = 0;   // members per device, rounded up: the planes' leading dimension / n_sites
  int64_t ld = 0;         // n_sites * maxCount: every device's planes have this leading dimension
  // per device: planes [3][n_alloc][ld] (element type by precision), statistics [3][n_alloc][n_sites][2],
  // gathered statistics [n_dev][3][n_run][n_sites][2], gathered planes [n_dev][3][n_run][ld] (on request)
  int32_t nAlloc = 0, nRun = 0, step0 = 0;
  std::vector<void*> planes, gatheredPlanes;
  std::vector<double*> stats, gatheredStats;
  size_t gatheredPlanesCap = 0, gatheredStatsCap = 0;
  size_t elem() const { return precision == SIPNET_F64 ? 8 : 4; }
  int n() const { return (int)devices.size(); }
};

#define NODE_HIP(expr)                                                        \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      setError(std::string("sipnet_node: ") + #expr + ": " + hipGetErrorString(e_)); \
      return SIPNET_ERR_NO_DEVICE;                                            \
    }                                                                         \
  } while (0)
#define NODE_RCCL(nd, expr)                                                   \
  do {                                                                        \
    ncclResult_t r_ = (expr);                                                 \
    if (r_ != ncclSuccess) {                                                  \
      setError(std::string("sipnet_node: ") + #expr + ": " + (nd)->rccl->errorString(r_)); \
      return SIPNET_ERR_NO_DEVICE;                                            \
    }                                                                         \
  } while (0)

// f on every device, one host thread each (the batch calls block on uploads); first failure wins
template <class F>
static int onEveryDevice(sipnet_node* nd, F f) {
  const int n = nd->n();
  std::vector<int> rc(n, SIPNET_OK);
  std::vector<std::string> msg(n);
  auto body = [&](int k) {
    if (hipSetDevice(nd->devices[k]) != hipSuccess) {
      rc[k] = SIPNET_ERR_NO_DEVICE;
      msg[k] = "hipSetDevice failed";
      return;
    }
    rc[k] = f(k);
    if (rc[k] != SIPNET_OK) msg[k] = sipnet_last_error();   // thread-local: carry it to the caller's thread
  };
  if (n == 1) {
    body(0);
  } else {
    std::vector<std::thread> th;
    for (int k = 0; k < n; k++) th.emplace_back(body, k);
    for (auto& t : th) t.join();
  }
  for (int k = 0; k < n; k++)
    if (rc[k] != SIPNET_OK) {
      setError("device " + std::to_string(nd->devices[k]) + ": " + msg[k]);
      return rc[k];
    }
  return SIPNET_OK;
}

extern "C" {

int sipnet_node_create(const int32_t* flags, int32_t n_sites, int32_t n_members, int32_t precision,
                       const int32_t* devices, int32_t n_devices, sipnet_node** out) {
  if (!flags || !out || !devices || n_devices <= 0 || n_devices > 64 || n_sites <= 0 || n_members < n_devices) {
    setError("sipnet_node_create: bad argument (needs at least one member per device)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int have = sipnet_device_count();
  for (int k = 0; k < n_devices; k++) {
    if (devices[k] < 0 || devices[k] >= have) {
      setError("sipnet_node_create: no usable HIP device " + std::to_string(devices[k]) +
               " (this engine has no CPU path)");
      return SIPNET_ERR_NO_DEVICE;
    }
    for (int j = 0; j < k; j++)
      if (devices[j] == devices[k]) {
        setError("sipnet_node_create: a device is listed twice");
        return SIPNET_ERR_BAD_ARGUMENT;
      }
  }
  Rccl* r = loadRccl();
  if (!r) {
    const char* why = dlerror();
    setError(std::string("sipnet_node_create: RCCL (librccl.so.1) cannot be loaded: ") + (why ? why : "missing symbol"));
    return SIPNET_ERR_NO_DEVICE;
  }
  sipnet_node* nd = new sipnet_node();
  memcpy(nd->flags, flags, sizeof nd->flags);
  nd->n_sites = n_sites;
  nd->n_members = n_members;
  nd->precision = precision;
  nd->rccl = r;
  nd->devices.assign(devices, devices + n_devices);
  nd->batches.assign(n_devices, nullptr);
  nd->streams.assign(n_devices, nullptr);
  nd->planes.assign(n_devices, nullptr);
  nd->gatheredPlanes.assign(n_devices, nullptr);
  nd->stats.assign(n_devices, nullptr);
  nd->gatheredStats.assign(n_devices, nullptr);
  for (int k = 0; k < n_devices; k++) {  // contiguous member ranges, sizes differ by at most one
    const int32_t a = (int32_t)((int64_t)n_members * k / n_devices), z = (int32_t)((int64_t)n_members * (k + 1) / n_devices);
    nd->first.push_back(a);
    nd->count.push_back(z - a);
    if (z - a > nd->maxCount) nd->maxCount = z - a;
  }
  nd->maxCount = (nd->maxCount + 1) & ~1;  // even: 16-byte aligned fp64 rows
  nd->ld = (int64_t)n_sites * nd->maxCount;
  int rc = SIPNET_OK;
  for (int k = 0; k < n_devices && rc == SIPNET_OK; k++) {
    rc = sipnet_batch_create(flags, n_sites, nd->count[k], precision, devices[k], &nd->batches[k]);
    if (rc == SIPNET_OK && (hipSetDevice(devices[k]) != hipSuccess ||
                            hipStreamCreateWithFlags(&nd->streams[k], hipStreamNonBlocking) != hipSuccess)) {
      setError("sipnet_node_create: hipStreamCreate failed");
      rc = SIPNET_ERR_NO_DEVICE;
    }
  }
  if (rc == SIPNET_OK) {
    nd->comms.assign(n_devices, nullptr);
    ncclResult_t nr = r->commInitAll(nd->comms.data(), n_devices, nd->devices.data());
    if (nr != ncclSuccess) {
      setError(std::string("sipnet_node_create: ncclCommInitAll: ") + r->errorString(nr));
      nd->comms.clear();
      rc = SIPNET_ERR_NO_DEVICE;
    }
  }
  if (rc != SIPNET_OK) {
    const std::string keep = sipnet_last_error();
    sipnet_node_destroy(nd);
    setError(keep);
    return rc;
  }
  *out = nd;
  return SIPNET_OK;
}

void sipnet_node_destroy(sipnet_node* nd) {
  if (!nd) return;
  for (int k = 0; k < nd->n(); k++) {
    (void)hipSetDevice(nd->devices[k]);
    if (nd->streams[k]) (void)hipStreamSynchronize(nd->streams[k]);
    if (k < (int)nd->comms.size() && nd->comms[k]) nd->rccl->commDestroy(nd->comms[k]);
    if (nd->planes[k]) (void)hipFree(nd->planes[k]);
    if (nd->gatheredPlanes[k]) (void)hipFree(nd->gatheredPlanes[k]);
    if (nd->stats[k]) (void)hipFree(nd->stats[k]);
    if (nd->gatheredStats[k]) (void)hipFree(nd->gatheredStats[k]);
    if (nd->batches[k]) sipnet_batch_destroy(nd->batches[k]);
    if (nd->streams[k]) (void)hipStreamDestroy(nd->streams[k]);
  }
  delete nd;
}

int32_t sipnet_node_n_devices(const sipnet_node* nd) { return nd ? nd->n() : 0; }
sipnet_batch* sipnet_node_batch(sipnet_node* nd, int32_t k) {
  return (nd && k >= 0 && k < nd->n()) ? nd->batches[k] : nullptr;
}
int sipnet_node_member_range(const sipnet_node* nd, int32_t k, int32_t* first, int32_t* count) {
  if (!nd || k < 0 || k >= nd->n()) return SIPNET_ERR_BAD_ARGUMENT;
  if (first) *first = nd->first[k];
  if (count) *count = nd->count[k];
  return SIPNET_OK;
}
int64_t sipnet_node_ld(const sipnet_node* nd) { return nd ? nd->ld : 0; }
const char* sipnet_node_collective_library(const sipnet_node* nd) {
  static thread_local std::string s;
  if (!nd || !nd->rccl) return "";
  int v = 0;
  nd->rccl->getVersion(&v);
  s = nd->rccl->path + " (RCCL " + std::to_string(v) + ")";
  return s.c_str();
}

int sipnet_node_set_climate(sipnet_node* nd, int32_t site, int32_t n_steps, const double* clim,
                            const int32_t* year, const int32_t* day) {
  if (!nd) return SIPNET_ERR_BAD_ARGUMENT;
  for (int k = 0; k < nd->n(); k++) {
    int rc = sipnet_batch_set_climate(nd->batches[k], site, n_steps, clim, year, day);
    if (rc) return rc;
  }
  return SIPNET_OK;
}
int sipnet_node_set_events(sipnet_node* nd, int32_t site, int32_t n_events, const sipnet_event* events) {
  if (!nd) return SIPNET_ERR_BAD_ARGUMENT;
  for (int k = 0; k < nd->n(); k++) {
    int rc = sipnet_batch_set_events(nd->batches[k], site, n_events, events);
    if (rc) return rc;
  }
  return SIPNET_OK;
}
int sipnet_node_set_params(sipnet_node* nd, int32_t site, int32_t first_member, int32_t count, const double* raw) {
  if (!nd || !raw || first_member < 0 || count <= 0 || first_member + count > nd->n_members) {
    setError("sipnet_node_set_params: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  return onEveryDevice(nd, [&](int k) -> int {
    const int32_t a = std::max(first_member, nd->first[k]);
    const int32_t z = std::min(first_member + count, nd->first[k] + nd->count[k]);
    if (z <= a) return SIPNET_OK;
    return sipnet_batch_set_params(nd->batches[k], site, a - nd->first[k], z - a,
                                   raw + (size_t)(a - first_member) * SIPNET_NPARAMS);
  });
}
int sipnet_node_set_math(sipnet_node* nd, int32_t policy) {
  if (!nd) return SIPNET_ERR_BAD_ARGUMENT;
  for (int k = 0; k < nd->n(); k++) {
    int rc = sipnet_batch_set_math(nd->batches[k], policy);
    if (rc) return rc;
  }
  return SIPNET_OK;
}
int sipnet_node_set_kernel(sipnet_node* nd, int32_t kernel, int32_t options) {
  if (!nd) return SIPNET_ERR_BAD_ARGUMENT;
  for (int k = 0; k < nd->n(); k++) {
    int rc = sipnet_batch_set_kernel(nd->batches[k], kernel, options);
    if (rc) return rc;
  }
  return SIPNET_OK;
}

int sipnet_node_setup(sipnet_node* nd) {
  if (!nd) return SIPNET_ERR_BAD_ARGUMENT;
  return onEveryDevice(nd, [&](int k) -> int { return sipnet_batch_setup(nd->batches[k], nd->streams[k]); });
}

int sipnet_node_run(sipnet_node* nd, int32_t step0, int32_t n_steps) {
  if (!nd || n_steps <= 0) {
    setError("sipnet_node_run: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const size_t planeBytes = (size_t)3 * n_steps * nd->ld * nd->elem();
  const size_t statDoubles = (size_t)3 * n_steps * nd->n_sites * 2;
  const bool grow = n_steps > nd->nAlloc;
  int rc = onEveryDevice(nd, [&](int k) -> int {
    if (grow) {
      NODE_HIP(hipStreamSynchronize(nd->streams[k]));
      if (nd->planes[k]) NODE_HIP(hipFree(nd->planes[k]));
      if (nd->stats[k]) NODE_HIP(hipFree(nd->stats[k]));
      nd->planes[k] = nullptr;
      nd->stats[k] = nullptr;
      NODE_HIP(hipMalloc(&nd->planes[k], planeBytes));
      NODE_HIP(hipMalloc(&nd->stats[k], statDoubles * sizeof(double)));
      // columns past a device's own members (the padding up to the common leading dimension) stay zero
      NODE_HIP(hipMemsetAsync(nd->planes[k], 0, planeBytes, nd->streams[k]));
    }
    char* p = (char*)nd->planes[k];
    const size_t one = (size_t)n_steps * nd->ld * nd->elem();
    return sipnet_batch_run_stats(nd->batches[k], step0, n_steps, p, p + one, p + 2 * one, nd->ld, nd->stats[k],
                                  nd->streams[k]);
  });
  if (rc) return rc;
  if (grow) nd->nAlloc = n_steps;
  nd->nRun = n_steps;
  nd->step0 = step0;
  return SIPNET_OK;
}

int sipnet_node_sync(sipnet_node* nd) {
  if (!nd) return SIPNET_ERR_BAD_ARGUMENT;
  for (int k = 0; k < nd->n(); k++) {
    NODE_HIP(hipSetDevice(nd->devices[k]));
    NODE_HIP(hipStreamSynchronize(nd->streams[k]));
  }
  return SIPNET_OK;
}

void* sipnet_node_planes(sipnet_node* nd, int32_t k) { return (nd && k >= 0 && k < nd->n()) ? nd->planes[k] : nullptr; }
double* sipnet_node_stats(sipnet_node* nd, int32_t k) { return (nd && k >= 0 && k < nd->n()) ? nd->stats[k] : nullptr; }
double* sipnet_node_gathered_stats(sipnet_node* nd, int32_t k) {
  return (nd && k >= 0 && k < nd->n()) ? nd->gatheredStats[k] : nullptr;
}
void* sipnet_node_gathered_planes(sipnet_node* nd, int32_t k) {
  return (nd && k >= 0 && k < nd->n()) ? nd->gatheredPlanes[k] : nullptr;
}

// ONE all-gather: every device's statistics block of the last sipnet_node_run to every device
int sipnet_node_gather_stats(sipnet_node* nd, double* host_total) {
  if (!nd || nd->nRun <= 0) {
    setError("sipnet_node_gather_stats: nothing has run");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int n = nd->n();
  const size_t block = (size_t)3 * nd->nRun * nd->n_sites * 2;  // doubles per device
  if (block * n > nd->gatheredStatsCap) {
    for (int k = 0; k < n; k++) {
      NODE_HIP(hipSetDevice(nd->devices[k]));
      NODE_HIP(hipStreamSynchronize(nd->streams[k]));
      if (nd->gatheredStats[k]) NODE_HIP(hipFree(nd->gatheredStats[k]));
      nd->gatheredStats[k] = nullptr;
      NODE_HIP(hipMalloc(&nd->gatheredStats[k], block * n * sizeof(double)));
    }
    nd->gatheredStatsCap = block * n;
  }
  NODE_RCCL(nd, nd->rccl->groupStart());
  for (int k = 0; k < n; k++) {
    NODE_HIP(hipSetDevice(nd->devices[k]));
    NODE_RCCL(nd, nd->rccl->allGather(nd->stats[k], nd->gatheredStats[k], block, ncclDouble, nd->comms[k], nd->streams[k]));
  }
  NODE_RCCL(nd, nd->rccl->groupEnd());
  if (host_total) {  // the ensemble's totals: the devices' blocks added up in device order (deterministic)
    std::vector<double> all(block * n);
    NODE_HIP(hipSetDevice(nd->devices[0]));
    NODE_HIP(hipStreamSynchronize(nd->streams[0]));
    NODE_HIP(hipMemcpy(all.data(), nd->gatheredStats[0], all.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < block; i++) {
      double s = 0.0;
      for (int k = 0; k < n; k++) s += all[(size_t)k * block + i];
      host_total[i] = s;
    }
  }
  return SIPNET_OK;
}

// ONE all-gather of the member-resolved planes of the last run: on every device
// gathered[(k * 3 + v) * n_steps + t][ld] = device k's plane v (columns past its members are zero)
int sipnet_node_gather_planes(sipnet_node* nd) {
  if (!nd || nd->nRun <= 0) {
    setError("sipnet_node_gather_planes: nothing has run");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (nd->nRun != nd->nAlloc) {
    setError("sipnet_node_gather_planes: the last run must fill the plane buffers (run the longest segment last)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int n = nd->n();
  const size_t count = (size_t)3 * nd->nRun * nd->ld;  // elements per device
  if (count * n > nd->gatheredPlanesCap) {
    for (int k = 0; k < n; k++) {
      NODE_HIP(hipSetDevice(nd->devices[k]));
      NODE_HIP(hipStreamSynchronize(nd->streams[k]));
      if (nd->gatheredPlanes[k]) NODE_HIP(hipFree(nd->gatheredPlanes[k]));
      nd->gatheredPlanes[k] = nullptr;
      NODE_HIP(hipMalloc(&nd->gatheredPlanes[k], count * n * nd->elem()));
    }
    nd->gatheredPlanesCap = count * n;
  }
  NODE_RCCL(nd, nd->rccl->groupStart());
  for (int k = 0; k < n; k++) {
    NODE_HIP(hipSetDevice(nd->devices[k]));
    NODE_RCCL(nd, nd->rccl->allGather(nd->planes[k], nd->gatheredPlanes[k], count,
                                      nd->precision == SIPNET_F64 ? ncclDouble : ncclFloat, nd->comms[k], nd->streams[k]));
  }
  NODE_RCCL(nd, nd->rccl->groupEnd());
  return SIPNET_OK;
}

}  // extern "C"
