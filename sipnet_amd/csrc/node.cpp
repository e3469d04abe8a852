// node.cpp -- one node, several GPUs, behind the C boundary (include/sipnet_amd.h, sipnet_node_*).
//
// The reference runs one process per ensemble member (frontend.c:212-250 is its whole host side);
// the north star asks for the ensemble axis sharded over the GPUs of one node with ONE RCCL
// all-gather over xGMI of the output block, issued from the C host.  A sipnet_node is that host
// object: per listed device one SHARD = one sipnet_batch, one HIP stream, one RCCL communicator rank
// (ncclCommInitAll) and one host thread that lives as long as the node and enqueues the shard's work.
// Two ways to cut the batch (SURVEY 8(e) "Partitioning"):
//   SIPNET_SHARD_MEMBERS  every shard holds a contiguous range of every site's members (c10k, c3, c5);
//   SIPNET_SHARD_SITES    every shard holds whole sites with all their members, so a site's forcing,
//                         events and plan exist on ONE device only (c4: 32 of 256 sites per GPU).
// The forward model has no coupling between members, so the step kernels never exchange anything;
// the collectives are all-gathers of what the shards computed:
//   sipnet_node_gather_stats   the per-(variable, step, site) sum / sum of squares block of every
//                              shard (0.84 MB per device and year) -- the default exchange;
//   sipnet_node_gather_planes  the north star's exchange as written: the member-resolved planes;
//   sipnet_node_pf_analysis    the particle filter's one exchange step: ONE all-gather of the
//                              log-weight blocks, after which every shard reads the ancestors it needs
//                              straight out of its peers' HBM (pf.hip: sipnet_batch_pf_resample_peers).
// RCCL is loaded on first use (dlopen of librccl.so.1: a process that already holds one -- PyTorch
// ships its own -- keeps using that one; a single-GPU batch never pays for the 570 MB library).
// Without a usable RCCL sipnet_node_create fails loudly; there is no fallback path.
// A device listed more than once (sipnet_node_create_sharded only) puts several shards on it -- the
// rehearsal of an N-shard run on a smaller machine.  RCCL refuses two ranks on one device, so such a
// node's all-gathers are device-to-device copies ordered by HIP events between the shards' streams;
// sharding, uploads, kernels, the peer tables and the gathered layouts are the same code.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sipnet_amd.h"
#include "shard_pool.h"

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
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;     // (sipnet_comm_*: ranks that are processes)
  decltype(&ncclCommInitRank) commInitRank = nullptr;
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
  RCCL_SYM(getUniqueId, "ncclGetUniqueId")
  RCCL_SYM(commInitRank, "ncclCommInitRank")
#undef RCCL_SYM
  return &r;
}

}  // namespace

struct sipnet_node {
  int32_t flags[SIPNET_NFLAGS];
  int32_t n_sites = 0, n_members = 0, precision = 0, mode = SIPNET_SHARD_MEMBERS;
  std::vector<int32_t> devices;
  // shard k owns members [first, first + count) of every site (member mode: sites0 = 0, nSites = n_sites)
  // or sites [sites0, sites0 + nSites) with all members (site mode: first = 0, count = n_members)
  std::vector<int32_t> first, count, sites0, nSites;
  std::vector<sipnet_batch*> batches;
  std::vector<hipStream_t> streams;
  std::vector<ncclComm_t> comms;   // empty: event-ordered copies (a device is listed twice)
  Rccl* rccl = nullptr;
  int32_t maxCount = 0;   // member mode: members per shard, rounded up to even; site mode: n_members
  int32_t maxSites = 0;   // site mode: sites per shard at most; member mode: n_sites
  int64_t ld = 0;         // maxSites * maxCount: every shard's planes have this leading dimension
  // per shard: planes [3][n_run][ld] (element type by precision), statistics [3][n_run][maxSites][2],
  // gathered statistics [n][3][n_run][maxSites][2], gathered planes [n][3][n_run][ld] (on request)
  int32_t nAlloc = 0, nRun = 0, step0 = 0;
  std::vector<void*> planes, gatheredPlanes;
  std::vector<double*> stats, gatheredStats, statsCompact;
  std::vector<size_t> statsCompactCap;
  size_t gatheredPlanesCap = 0, gatheredStatsCap = 0;
  // particle filter: per shard the gathered log-weight blocks, its particles' ancestors, a ring of total weights
  bool pfConnected = false;
  int64_t pfBlock = 0;
  int32_t pfCycles = 0;
  std::vector<double*> pfGathered;
  std::vector<int32_t*> pfAnc;
  std::vector<int64_t*> pfTotals;
  static constexpr int kPfTotals = 64;
  // sipnet_node_run_gathering: the run cut into segments, segment j at rows [3 * segCuts[j], 3 * segCuts[j + 1]) of
  // every shard's planes ([3][len_j][ld] each) and at [n][3][len_j][ld] from element n * 3 * ld * segCuts[j] of the
  // gathered planes; the all-gathers run on the shards' second streams
  bool segmented = false;
  std::vector<int32_t> segCuts;
  // sipnet_node_run_gathering_reduced: what travelled instead of the planes -- per shard [3][R][ld] (segment by segment, like
  // the planes), gathered [n][3][R_j][ld] per segment; R rows = steps (SIPNET_GATHER_F32, floats) or groups of sum_steps steps
  // (SIPNET_GATHER_SUMS, doubles); redCuts: the segments' first rows
  int32_t reducedForm = 0, reducedSumSteps = 0;
  bool reducedInKernel = false;   // the sums came out of the step kernels' own launches (no planes were written)
  std::vector<int32_t> redCuts;
  std::vector<void*> reduced, gatheredReduced;
  size_t reducedCap = 0;   // bytes per shard
  std::vector<hipStream_t> gatherStreams;
  std::vector<hipEvent_t> evSeg, evGathered;
  // event-ordered transport
  std::vector<hipEvent_t> evReady, evCopied;
  std::vector<const void*> agSend;
  // the shards' host threads, the task hand-over and the barrier they meet at (shard_pool.h: no HIP in it)
  sipnet::ShardPool pool;

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

// f(k) for every shard k, each on the shard's own host thread (a node of one shard: on the caller's);
// returns when all have returned, the first failure wins
template <class F>
static int onEveryShard(sipnet_node* nd, F f) {
  const int bad = nd->pool.run(f);
  if (bad < 0) return SIPNET_OK;
  setError("device " + std::to_string(nd->devices[bad]) + " (shard " + std::to_string(bad) + "): " + nd->pool.msg[bad]);
  return nd->pool.rc[bad];
}

// All-gather of `bytes` per shard among the shards, called by shard k's thread from inside an onEveryShard
// task: recv = [n][bytes] on shard k's device, send = this shard's block (may be its own slice of recv).
// RCCL when every shard has a device of its own; otherwise copies ordered by events (see the file header).
static int allGatherShard(sipnet_node* nd, int k, const void* send, void* recv, size_t bytes, hipStream_t stream = nullptr) {
  const int n = nd->n();
  if (!stream) stream = nd->streams[k];
  if (!nd->comms.empty()) {
    // A collective nobody may be missing from: a shard whose task has failed (its launch, an upload) returns WITHOUT
    // enqueuing its part, and the others' ncclAllGather would then never complete -- the next synchronisation of their
    // streams would hang instead of reporting the error.  So the shards' host threads first agree that all of them
    // have come this far (the host barrier; a failed shard breaks it): either everybody enqueues or nobody does.
    if (!nd->pool.bar.arrive()) {
      setError("sipnet_node: another shard failed");
      return SIPNET_ERR_INTERNAL;
    }
    NODE_RCCL(nd, nd->rccl->allGather(send, recv, bytes, ncclChar, nd->comms[k], stream));
    return SIPNET_OK;
  }
  nd->agSend[k] = send;
  NODE_HIP(hipEventRecord(nd->evReady[k], stream));
  if (!nd->pool.bar.arrive()) {
    setError("sipnet_node: another shard failed");
    return SIPNET_ERR_INTERNAL;
  }
  for (int s = 0; s < n; s++) {
    char* dst = (char*)recv + (size_t)s * bytes;
    if (s != k) NODE_HIP(hipStreamWaitEvent(stream, nd->evReady[s], 0));
    if ((const void*)dst != nd->agSend[s])
      NODE_HIP(hipMemcpyAsync(dst, nd->agSend[s], bytes, hipMemcpyDeviceToDevice, stream));
  }
  NODE_HIP(hipEventRecord(nd->evCopied[k], stream));
  if (!nd->pool.bar.arrive()) {
    setError("sipnet_node: another shard failed");
    return SIPNET_ERR_INTERNAL;
  }
  // what follows on this stream may overwrite the block the others copy from
  for (int s = 0; s < n; s++)
    if (s != k) NODE_HIP(hipStreamWaitEvent(stream, nd->evCopied[s], 0));
  return SIPNET_OK;
}

static int createNode(const int32_t* flags, int32_t n_sites, int32_t n_members, int32_t precision,
                      const int32_t* devices, int32_t n_devices, int32_t mode, bool allowShared, sipnet_node** out) {
  const char* fn = allowShared ? "sipnet_node_create_sharded" : "sipnet_node_create";
  if (!flags || !out || !devices || n_devices <= 0 || n_devices > 64 || n_sites <= 0 || n_members <= 0 ||
      (mode != SIPNET_SHARD_MEMBERS && mode != SIPNET_SHARD_SITES) ||
      (mode == SIPNET_SHARD_MEMBERS ? n_members : n_sites) < n_devices) {
    setError(std::string(fn) + ": bad argument (needs at least one member -- or, sharding sites, one site -- per device)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int have = sipnet_device_count();
  bool shared = false;
  for (int k = 0; k < n_devices; k++) {
    if (devices[k] < 0 || devices[k] >= have) {
      setError(std::string(fn) + ": no usable HIP device " + std::to_string(devices[k]) + " (this engine has no CPU path)");
      return SIPNET_ERR_NO_DEVICE;
    }
    for (int j = 0; j < k; j++)
      if (devices[j] == devices[k]) shared = true;
  }
  if (shared && !allowShared) {
    setError("sipnet_node_create: a device is listed twice");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  Rccl* r = nullptr;
  if (!shared) {
    r = loadRccl();
    if (!r) {
      const char* why = dlerror();
      setError(std::string(fn) + ": RCCL (librccl.so.1) cannot be loaded: " + (why ? why : "missing symbol"));
      return SIPNET_ERR_NO_DEVICE;
    }
  }
  sipnet_node* nd = new sipnet_node();
  memcpy(nd->flags, flags, sizeof nd->flags);
  nd->n_sites = n_sites;
  nd->n_members = n_members;
  nd->precision = precision;
  nd->mode = mode;
  nd->rccl = r;
  nd->devices.assign(devices, devices + n_devices);
  nd->batches.assign(n_devices, nullptr);
  nd->streams.assign(n_devices, nullptr);
  nd->planes.assign(n_devices, nullptr);
  nd->gatheredPlanes.assign(n_devices, nullptr);
  nd->reduced.assign(n_devices, nullptr);
  nd->gatheredReduced.assign(n_devices, nullptr);
  nd->stats.assign(n_devices, nullptr);
  nd->gatheredStats.assign(n_devices, nullptr);
  nd->statsCompact.assign(n_devices, nullptr);
  nd->statsCompactCap.assign(n_devices, 0);
  nd->pfGathered.assign(n_devices, nullptr);
  nd->pfAnc.assign(n_devices, nullptr);
  nd->pfTotals.assign(n_devices, nullptr);
  nd->evReady.assign(n_devices, nullptr);
  nd->evCopied.assign(n_devices, nullptr);
  nd->gatherStreams.assign(n_devices, nullptr);
  nd->evSeg.assign(n_devices, nullptr);
  nd->evGathered.assign(n_devices, nullptr);
  nd->agSend.assign(n_devices, nullptr);
  for (int k = 0; k < n_devices; k++) {  // contiguous ranges, sizes differ by at most one
    const int32_t total = mode == SIPNET_SHARD_MEMBERS ? n_members : n_sites;
    const int32_t a = (int32_t)((int64_t)total * k / n_devices), z = (int32_t)((int64_t)total * (k + 1) / n_devices);
    if (mode == SIPNET_SHARD_MEMBERS) {
      nd->first.push_back(a);
      nd->count.push_back(z - a);
      nd->sites0.push_back(0);
      nd->nSites.push_back(n_sites);
      nd->maxCount = std::max(nd->maxCount, z - a);
    } else {
      nd->first.push_back(0);
      nd->count.push_back(n_members);
      nd->sites0.push_back(a);
      nd->nSites.push_back(z - a);
      nd->maxSites = std::max(nd->maxSites, z - a);
    }
  }
  if (mode == SIPNET_SHARD_MEMBERS) {
    nd->maxCount = (nd->maxCount + 1) & ~1;  // even: 16-byte aligned fp64 rows
    nd->maxSites = n_sites;
  } else {
    nd->maxCount = n_members;
  }
  nd->ld = (int64_t)nd->maxSites * nd->maxCount;
  int rc = SIPNET_OK;
  for (int k = 0; k < n_devices && rc == SIPNET_OK; k++) {
    rc = sipnet_batch_create(flags, nd->nSites[k], nd->count[k], precision, devices[k], &nd->batches[k]);
    if (rc == SIPNET_OK) {
      // shards that share a device analyse at the same time: each may keep 1 / (their number) of it spinning (pf.hip fusedBudget)
      int32_t sameDevice = 0;
      for (int q = 0; q < n_devices; q++) sameDevice += devices[q] == devices[k];
      rc = sipnet_batch_set_device_share(nd->batches[k], sameDevice);
    }
    if (rc == SIPNET_OK && (hipSetDevice(devices[k]) != hipSuccess ||
                            hipStreamCreateWithFlags(&nd->streams[k], hipStreamNonBlocking) != hipSuccess ||
                            hipEventCreateWithFlags(&nd->evReady[k], hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&nd->evCopied[k], hipEventDisableTiming) != hipSuccess ||
                            hipStreamCreateWithFlags(&nd->gatherStreams[k], hipStreamNonBlocking) != hipSuccess ||
                            hipEventCreateWithFlags(&nd->evSeg[k], hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&nd->evGathered[k], hipEventDisableTiming) != hipSuccess)) {
      setError(std::string(fn) + ": hipStreamCreate / hipEventCreate failed");
      rc = SIPNET_ERR_NO_DEVICE;
    }
  }
  if (rc == SIPNET_OK && !shared) {
    nd->comms.assign(n_devices, nullptr);
    ncclResult_t nr = r->commInitAll(nd->comms.data(), n_devices, nd->devices.data());
    if (nr != ncclSuccess) {
      setError(std::string(fn) + ": ncclCommInitAll: " + r->errorString(nr));
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
  // one host thread per shard from here on (none for a node of one shard: its tasks run on the caller's thread)
  nd->pool.start(n_devices, [nd](int k) { return hipSetDevice(nd->devices[k]) == hipSuccess; },
                 [] { return std::string(sipnet_last_error()); }, SIPNET_ERR_NO_DEVICE);
  *out = nd;
  return SIPNET_OK;
}

// kernels of sipnet_node_run_gathering_reduced (below)
namespace {
__global__ __launch_bounds__(256) void planesToF32Kernel(const double* __restrict__ src, float* __restrict__ dst, size_t n2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // pairs (ld is even)
  if (i >= n2) return;
  const double2 v = ((const double2*)src)[i];
  ((float2*)dst)[i] = make_float2((float)v.x, (float)v.y);
}
// out[(v * groups + g) * ld + c] = sum over the steps t of group g, in step order, of plane v's [t][c]   (a thread = a column)
template <typename T>
__global__ __launch_bounds__(256) void sumStepsKernel(const T* __restrict__ planes, size_t planeStride, int32_t rows, int32_t k,
                                                      int32_t groups, int64_t ld, double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  const int g = blockIdx.y, v = blockIdx.z;
  const T* __restrict__ p = planes + (size_t)v * planeStride + (size_t)g * k * ld + c;
  const int cnt = (g + 1) * k <= rows ? k : rows - g * k;
  double acc = 0.0;
  int t = 0;
  for (; t + 8 <= cnt; t += 8) {   // eight loads in flight, added in step order
    T x[8];
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = p[(size_t)(t + q) * ld];
#pragma unroll
    for (int q = 0; q < 8; q++) acc += (double)x[q];
  }
  for (; t < cnt; t++) acc += (double)p[(size_t)t * ld];
  out[((size_t)v * groups + g) * ld + c] = acc;
}
}  // namespace

extern "C" {

int sipnet_node_create(const int32_t* flags, int32_t n_sites, int32_t n_members, int32_t precision,
                       const int32_t* devices, int32_t n_devices, sipnet_node** out) {
  return createNode(flags, n_sites, n_members, precision, devices, n_devices, SIPNET_SHARD_MEMBERS, false, out);
}
int sipnet_node_create_sharded(const int32_t* flags, int32_t n_sites, int32_t n_members, int32_t precision,
                               const int32_t* devices, int32_t n_devices, int32_t shard_mode, sipnet_node** out) {
  return createNode(flags, n_sites, n_members, precision, devices, n_devices, shard_mode, true, out);
}

void sipnet_node_destroy(sipnet_node* nd) {
  if (!nd) return;
  nd->pool.stop();
  for (int k = 0; k < nd->n(); k++) {
    (void)hipSetDevice(nd->devices[k]);
    if (nd->streams[k]) (void)hipStreamSynchronize(nd->streams[k]);
    if (nd->gatherStreams[k]) (void)hipStreamSynchronize(nd->gatherStreams[k]);
  }
  for (int k = 0; k < nd->n(); k++) {
    (void)hipSetDevice(nd->devices[k]);
    if (k < (int)nd->comms.size() && nd->comms[k]) nd->rccl->commDestroy(nd->comms[k]);
    if (nd->planes[k]) (void)hipFree(nd->planes[k]);
    if (nd->gatheredPlanes[k]) (void)hipFree(nd->gatheredPlanes[k]);
    if (nd->reduced[k]) (void)hipFree(nd->reduced[k]);
    if (nd->gatheredReduced[k]) (void)hipFree(nd->gatheredReduced[k]);
    if (nd->stats[k]) (void)hipFree(nd->stats[k]);
    if (nd->gatheredStats[k]) (void)hipFree(nd->gatheredStats[k]);
    if (nd->statsCompact[k]) (void)hipFree(nd->statsCompact[k]);
    if (nd->pfGathered[k]) (void)hipFree(nd->pfGathered[k]);
    if (nd->pfAnc[k]) (void)hipFree(nd->pfAnc[k]);
    if (nd->pfTotals[k]) (void)hipFree(nd->pfTotals[k]);
    if (nd->evReady[k]) (void)hipEventDestroy(nd->evReady[k]);
    if (nd->evCopied[k]) (void)hipEventDestroy(nd->evCopied[k]);
    if (nd->evSeg[k]) (void)hipEventDestroy(nd->evSeg[k]);
    if (nd->evGathered[k]) (void)hipEventDestroy(nd->evGathered[k]);
    if (nd->gatherStreams[k]) (void)hipStreamDestroy(nd->gatherStreams[k]);
    if (nd->batches[k]) sipnet_batch_destroy(nd->batches[k]);
    if (nd->streams[k]) (void)hipStreamDestroy(nd->streams[k]);
  }
  delete nd;
}

int32_t sipnet_node_n_devices(const sipnet_node* nd) { return nd ? nd->n() : 0; }
int32_t sipnet_node_shard_mode(const sipnet_node* nd) { return nd ? nd->mode : -1; }
sipnet_batch* sipnet_node_batch(sipnet_node* nd, int32_t k) {
  return (nd && k >= 0 && k < nd->n()) ? nd->batches[k] : nullptr;
}
void* sipnet_node_stream(sipnet_node* nd, int32_t k) { return (nd && k >= 0 && k < nd->n()) ? (void*)nd->streams[k] : nullptr; }
int sipnet_node_member_range(const sipnet_node* nd, int32_t k, int32_t* first, int32_t* count) {
  if (!nd || k < 0 || k >= nd->n()) return SIPNET_ERR_BAD_ARGUMENT;
  if (first) *first = nd->first[k];
  if (count) *count = nd->count[k];
  return SIPNET_OK;
}
int sipnet_node_site_range(const sipnet_node* nd, int32_t k, int32_t* first, int32_t* count) {
  if (!nd || k < 0 || k >= nd->n()) return SIPNET_ERR_BAD_ARGUMENT;
  if (first) *first = nd->sites0[k];
  if (count) *count = nd->nSites[k];
  return SIPNET_OK;
}
int64_t sipnet_node_ld(const sipnet_node* nd) { return nd ? nd->ld : 0; }
// ---- a RCCL communicator for ranks that are processes (include/sipnet_amd.h: sipnet_comm_*) -------------------------------
struct sipnet_comm {
  Rccl* rccl = nullptr;
  ncclComm_t comm = nullptr;
  int32_t world = 0, rank = 0, device = 0;
};
#define COMM_RCCL(r, expr)                                                            \
  do {                                                                                \
    ncclResult_t r_ = (expr);                                                         \
    if (r_ != ncclSuccess) {                                                          \
      setError(std::string("sipnet_comm: ") + #expr + ": " + (r)->errorString(r_));   \
      return SIPNET_ERR_NO_DEVICE;                                                    \
    }                                                                                 \
  } while (0)
int sipnet_comm_unique_id(uint8_t id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  Rccl* r = loadRccl();
  if (!id || !r) {
    setError("sipnet_comm_unique_id: no usable RCCL (librccl.so.1)");
    return SIPNET_ERR_NO_DEVICE;
  }
  ncclUniqueId u;
  COMM_RCCL(r, r->getUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return SIPNET_OK;
}
int sipnet_comm_create(const uint8_t id[128], int32_t world, int32_t rank, int32_t device, sipnet_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) {
    setError("sipnet_comm_create: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  Rccl* r = loadRccl();
  if (!r) {
    setError("sipnet_comm_create: no usable RCCL (librccl.so.1)");
    return SIPNET_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    setError("sipnet_comm_create: no usable HIP device " + std::to_string(device));
    return SIPNET_ERR_NO_DEVICE;
  }
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  sipnet_comm* c = new sipnet_comm();
  c->rccl = r;
  c->world = world;
  c->rank = rank;
  c->device = device;
  ncclResult_t nr = r->commInitRank(&c->comm, world, u, rank);
  if (nr != ncclSuccess) {
    setError(std::string("sipnet_comm_create: ncclCommInitRank: ") + r->errorString(nr));
    delete c;
    return SIPNET_ERR_NO_DEVICE;
  }
  *out = c;
  return SIPNET_OK;
}
int sipnet_comm_all_gather(sipnet_comm* c, const void* d_send, void* d_recv, int64_t bytes_per_rank, void* hip_stream) {
  if (!c || !d_send || !d_recv || bytes_per_rank <= 0) {
    setError("sipnet_comm_all_gather: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (hipSetDevice(c->device) != hipSuccess) {
    (void)hipGetLastError();
    return SIPNET_ERR_NO_DEVICE;
  }
  COMM_RCCL(c->rccl, c->rccl->allGather(d_send, d_recv, (size_t)bytes_per_rank, ncclChar, c->comm, (hipStream_t)hip_stream));
  return SIPNET_OK;
}
int32_t sipnet_comm_world(const sipnet_comm* c) { return c ? c->world : 0; }
void sipnet_comm_destroy(sipnet_comm* c) {
  if (!c) return;
  if (c->comm) (void)c->rccl->commDestroy(c->comm);
  delete c;
}
#undef COMM_RCCL

const char* sipnet_node_collective_library(const sipnet_node* nd) {
  static thread_local std::string s;
  if (!nd) return "";
  if (!nd->rccl) return "event-ordered device copies (shards share a device; no RCCL communicator)";
  int v = 0;
  nd->rccl->getVersion(&v);
  s = nd->rccl->path + " (RCCL " + std::to_string(v) + ")";
  return s.c_str();
}

// the shard that owns `site` and the site's index inside that shard's batch (member mode: every shard, same index)
static int ownerOf(const sipnet_node* nd, int32_t site) {
  for (int k = 0; k < nd->n(); k++)
    if (site >= nd->sites0[k] && site < nd->sites0[k] + nd->nSites[k]) return k;
  return -1;
}

int sipnet_node_set_climate(sipnet_node* nd, int32_t site, int32_t n_steps, const double* clim,
                            const int32_t* year, const int32_t* day) {
  if (!nd || site < 0 || site >= nd->n_sites) {
    setError("sipnet_node_set_climate: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (nd->mode == SIPNET_SHARD_SITES) {   // the forcing of a site goes to the ONE shard that owns the site
    const int k = ownerOf(nd, site);
    return sipnet_batch_set_climate(nd->batches[k], site - nd->sites0[k], n_steps, clim, year, day);
  }
  for (int k = 0; k < nd->n(); k++) {
    int rc = sipnet_batch_set_climate(nd->batches[k], site, n_steps, clim, year, day);
    if (rc) return rc;
  }
  return SIPNET_OK;
}
int sipnet_node_set_events(sipnet_node* nd, int32_t site, int32_t n_events, const sipnet_event* events) {
  if (!nd || site < 0 || site >= nd->n_sites) {
    setError("sipnet_node_set_events: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (nd->mode == SIPNET_SHARD_SITES) {
    const int k = ownerOf(nd, site);
    return sipnet_batch_set_events(nd->batches[k], site - nd->sites0[k], n_events, events);
  }
  for (int k = 0; k < nd->n(); k++) {
    int rc = sipnet_batch_set_events(nd->batches[k], site, n_events, events);
    if (rc) return rc;
  }
  return SIPNET_OK;
}
int sipnet_node_set_params(sipnet_node* nd, int32_t site, int32_t first_member, int32_t count, const double* raw) {
  if (!nd || !raw || first_member < 0 || count <= 0 || first_member + count > nd->n_members ||
      (site != SIPNET_ALL_SITES && (site < 0 || site >= nd->n_sites))) {
    setError("sipnet_node_set_params: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  return onEveryShard(nd, [&](int k) -> int {
    int32_t local = site;
    if (nd->mode == SIPNET_SHARD_SITES && site != SIPNET_ALL_SITES) {
      if (site < nd->sites0[k] || site >= nd->sites0[k] + nd->nSites[k]) return SIPNET_OK;
      local = site - nd->sites0[k];
    }
    const int32_t a = std::max(first_member, nd->first[k]);
    const int32_t z = std::min(first_member + count, nd->first[k] + nd->count[k]);
    if (z <= a) return SIPNET_OK;
    return sipnet_batch_set_params(nd->batches[k], local, a - nd->first[k], z - a,
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
  return onEveryShard(nd, [&](int k) -> int { return sipnet_batch_setup(nd->batches[k], nd->streams[k]); });
}

// shard k's planes and statistics block for runs of up to n_steps records (called on the shard's thread)
static int growRunBuffers(sipnet_node* nd, int k, int32_t n_steps) {
  const size_t planeBytes = (size_t)3 * n_steps * nd->ld * nd->elem();
  const size_t statDoubles = (size_t)3 * n_steps * nd->maxSites * 2;
  NODE_HIP(hipStreamSynchronize(nd->streams[k]));
  if (nd->planes[k]) NODE_HIP(hipFree(nd->planes[k]));
  if (nd->stats[k]) NODE_HIP(hipFree(nd->stats[k]));
  nd->planes[k] = nullptr;
  nd->stats[k] = nullptr;
  NODE_HIP(hipMalloc(&nd->planes[k], planeBytes));
  NODE_HIP(hipMalloc(&nd->stats[k], statDoubles * sizeof(double)));
  // columns past a shard's own (the padding up to the common leading dimension) and the statistics of
  // sites it does not have stay zero: no kernel ever writes them, whatever the length of a run
  NODE_HIP(hipMemsetAsync(nd->planes[k], 0, planeBytes, nd->streams[k]));
  NODE_HIP(hipMemsetAsync(nd->stats[k], 0, statDoubles * sizeof(double), nd->streams[k]));
  return SIPNET_OK;
}

// does a site of shard k end before record `upTo`?  Then the kernels leave rows of the shard's planes unwritten, and
// what stands there depends on the layout of the run before (plain or segmented): such a shard's planes are cleared
// at the start of every run
static bool shardEndsEarly(const sipnet_node* nd, int k, int32_t upTo) {
  for (int32_t s = 0; s < nd->nSites[k]; s++)
    if (sipnet_batch_site_nsteps(nd->batches[k], s) < upTo) return true;
  return false;
}

static int runShards(sipnet_node* nd, int32_t step0, int32_t n_steps, bool withStats) {
  if (!nd || n_steps <= 0) {
    setError("sipnet_node_run: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const size_t statDoubles = (size_t)3 * n_steps * nd->maxSites * 2;
  const bool grow = n_steps > nd->nAlloc;
  int rc = onEveryShard(nd, [&](int k) -> int {
    if (grow) {
      int rcg = growRunBuffers(nd, k, n_steps);
      if (rcg) return rcg;
    }
    char* p = (char*)nd->planes[k];
    const size_t one = (size_t)n_steps * nd->ld * nd->elem();
    if (!grow && shardEndsEarly(nd, k, step0 + n_steps)) NODE_HIP(hipMemsetAsync(p, 0, 3 * one, nd->streams[k]));
    // (site shards may hold forcings of different lengths: a shard runs to the end of ITS longest site; the rows
    // past it stay what they are -- zero -- in the common [n_steps] layout)
    const int32_t have = sipnet_batch_nsteps(nd->batches[k]);
    const int32_t nLoc = step0 + n_steps <= have ? n_steps : have - step0;
    if (nLoc <= 0) return SIPNET_OK;
    if (!withStats)
      return sipnet_batch_run(nd->batches[k], step0, nLoc, p, p + one, p + 2 * one, nullptr, nd->ld, nd->streams[k]);
    if (nd->nSites[k] == nd->maxSites && nLoc == n_steps)
      return sipnet_batch_run_stats(nd->batches[k], step0, n_steps, p, p + one, p + 2 * one, nd->ld, nd->stats[k],
                                    nd->streams[k]);
    // a shard with fewer sites than the largest, or a shorter run: its block [3][nLoc][nSites][2] is produced
    // compactly and spread out to the common shape [3][n_steps][maxSites][2] (everything else stays zero)
    const size_t compactDoubles = (size_t)3 * nLoc * nd->nSites[k] * 2;
    if (compactDoubles > nd->statsCompactCap[k]) {
      NODE_HIP(hipStreamSynchronize(nd->streams[k]));
      if (nd->statsCompact[k]) NODE_HIP(hipFree(nd->statsCompact[k]));
      nd->statsCompact[k] = nullptr;
      NODE_HIP(hipMalloc(&nd->statsCompact[k], compactDoubles * sizeof(double)));
      nd->statsCompactCap[k] = compactDoubles;
    }
    int rc2 = sipnet_batch_run_stats(nd->batches[k], step0, nLoc, p, p + one, p + 2 * one, nd->ld, nd->statsCompact[k],
                                     nd->streams[k]);
    if (rc2) return rc2;
    if (nLoc != n_steps) NODE_HIP(hipMemsetAsync(nd->stats[k], 0, statDoubles * sizeof(double), nd->streams[k]));
    for (int v = 0; v < 3; v++)
      NODE_HIP(hipMemcpy2DAsync(nd->stats[k] + (size_t)v * n_steps * nd->maxSites * 2, (size_t)nd->maxSites * 2 * sizeof(double),
                                nd->statsCompact[k] + (size_t)v * nLoc * nd->nSites[k] * 2,
                                (size_t)nd->nSites[k] * 2 * sizeof(double), (size_t)nd->nSites[k] * 2 * sizeof(double),
                                (size_t)nLoc, hipMemcpyDeviceToDevice, nd->streams[k]));
    return SIPNET_OK;
  });
  if (rc) return rc;
  if (grow) nd->nAlloc = n_steps;
  nd->nRun = n_steps;
  nd->step0 = step0;
  nd->segmented = false;
  return SIPNET_OK;
}

// The north star's exchange as written, overlapped (SURVEY 8(e) "Collective": "gather per chunk ... overlapped on a
// side stream"): the run is cut into segments -- one launch each -- and the member-resolved planes of segment j
// travel (one all-gather per segment, on every shard's SECOND stream) under the step kernel of segment j + 1.
// Every segment has its own place in the shard's planes and in the gathered block, so nothing is double-buffered
// and nothing waits for a consumer (288 GB of HBM: the gathered year of c4's shape is 8 x 13.8 GB).
// ---- member-resolved outputs that fit under the kernel (round 6) ------------------------------------------------------
// The raw fp64 planes of a year are 4.3 GB per rank at 10 240 members: ~100 ms of xGMI time against 8 ms of compute.  What a
// consumer of the reference's per-step output (sipnet.c:453-473) aggregates anyway travels instead: the same planes as floats
// (half the bytes), or every member's sums over groups of sum_steps steps (daily sums of a half-hourly year: 1 / 48 of the
// bytes, 90 MB per rank) -- produced from segment j's planes on the shard's SECOND stream while segment j + 1 computes (the
// step kernel of such a shape leaves compute units idle and uses a few per cent of the HBM bandwidth), then all-gathered there.

// form: 0 the planes themselves (sipnet_node_run_gathering), SIPNET_GATHER_F32, SIPNET_GATHER_SUMS (sumSteps steps per group)
static int runGathering(sipnet_node* nd, int32_t step0, int32_t n_steps, int32_t n_segments, int32_t form = 0, int32_t sumSteps = 0) {
  if (!nd || n_steps <= 0 || n_segments <= 0 || n_segments > n_steps || step0 < 0) {
    setError("sipnet_node_run_gathering: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (form != 0 && form != SIPNET_GATHER_F32 && form != SIPNET_GATHER_SUMS) {
    setError("sipnet_node_run_gathering_reduced: form must be SIPNET_GATHER_F32 or SIPNET_GATHER_SUMS");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (form == SIPNET_GATHER_F32 && nd->precision != SIPNET_F64) {
    setError("sipnet_node_run_gathering_reduced: the planes of an fp32-mixed node are floats already (sipnet_node_run_gathering)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int32_t nGroups = form == SIPNET_GATHER_SUMS ? (sumSteps > 0 ? (n_steps + sumSteps - 1) / sumSteps : 0) : n_steps;
  if (form == SIPNET_GATHER_SUMS && (sumSteps <= 0 || n_segments > nGroups)) {
    setError("sipnet_node_run_gathering_reduced: sum_steps > 0 and at most one segment per group of steps");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int n = nd->n();
  const size_t count = (size_t)3 * n_steps * nd->ld;   // elements per shard
  const bool grow = n_steps > nd->nAlloc;
  const bool growGathered = count * n > nd->gatheredPlanesCap;
  // cuts at whole 16-step tiles of the site plan where the segments are long enough for that
  std::vector<int32_t> cuts(n_segments + 1, 0);
  cuts[n_segments] = n_steps;
  for (int j = 1; j < n_segments; j++) {
    const int32_t raw = (int32_t)((int64_t)n_steps * j / n_segments);
    const int32_t tile = ((step0 + raw) & ~15) - step0;
    cuts[j] = tile > cuts[j - 1] ? tile : raw;
    if (form == SIPNET_GATHER_SUMS) cuts[j] = (int32_t)((int64_t)nGroups * j / n_segments) * sumSteps;   // whole groups
  }
  // rows of the reduced block per segment (steps, or groups of steps) and its element size
  std::vector<int32_t> redCuts(n_segments + 1, 0);
  for (int j = 0; j <= n_segments; j++) redCuts[j] = form == SIPNET_GATHER_SUMS ? (cuts[j] + sumSteps - 1) / sumSteps : cuts[j];
  const size_t redElem = form == SIPNET_GATHER_F32 ? sizeof(float) : sizeof(double);
  const size_t redBytes = form ? (size_t)3 * nGroups * nd->ld * redElem : 0;
  // sums: inside the step kernel's own launch where every shard's batch has such a kernel (sipnet_batch_run_sums: no planes
  // are written at all), else the planes summed by a pass on the second stream
  bool sumsInKernel = form == SIPNET_GATHER_SUMS;
  for (int k = 0; k < n && sumsInKernel; k++) sumsInKernel = sipnet_batch_sums_in_kernel(nd->batches[k]) != 0;
  nd->reducedInKernel = sumsInKernel;
  const bool growReduced = redBytes > nd->reducedCap;
  int rc = onEveryShard(nd, [&](int k) -> int {
    if (growReduced) {
      NODE_HIP(hipStreamSynchronize(nd->streams[k]));
      NODE_HIP(hipStreamSynchronize(nd->gatherStreams[k]));
      if (nd->reduced[k]) NODE_HIP(hipFree(nd->reduced[k]));
      if (nd->gatheredReduced[k]) NODE_HIP(hipFree(nd->gatheredReduced[k]));
      nd->reduced[k] = nd->gatheredReduced[k] = nullptr;
      NODE_HIP(hipMalloc(&nd->reduced[k], redBytes));
      NODE_HIP(hipMalloc(&nd->gatheredReduced[k], redBytes * n));
    }
    if (grow) {
      NODE_HIP(hipStreamSynchronize(nd->gatherStreams[k]));
      int rcg = growRunBuffers(nd, k, n_steps);
      if (rcg) return rcg;
    }
    if (growGathered && !form) {
      NODE_HIP(hipStreamSynchronize(nd->streams[k]));
      NODE_HIP(hipStreamSynchronize(nd->gatherStreams[k]));
      if (nd->gatheredPlanes[k]) NODE_HIP(hipFree(nd->gatheredPlanes[k]));
      nd->gatheredPlanes[k] = nullptr;
      NODE_HIP(hipMalloc(&nd->gatheredPlanes[k], count * n * nd->elem()));
    }
    return SIPNET_OK;
  });
  if (rc) return rc;
  if (grow) nd->nAlloc = n_steps;
  if (growGathered && !form) nd->gatheredPlanesCap = count * n;
  if (growReduced) nd->reducedCap = redBytes;
  // every shard's host thread walks the segments on its own: launch, hand over to the second stream, all-gather there
  // (RCCL: one communicator per thread, the multi-thread idiom; shards sharing a device: event-ordered copies, the
  // threads meeting at a host barrier per segment)
  rc = onEveryShard(nd, [&](int k) -> int {
    const int32_t have = sipnet_batch_nsteps(nd->batches[k]);
    if (!grow && shardEndsEarly(nd, k, step0 + n_steps))
      NODE_HIP(hipMemsetAsync(nd->planes[k], 0, count * nd->elem(), nd->streams[k]));
    for (int j = 0; j < n_segments; j++) {
      const int32_t a = cuts[j], len = cuts[j + 1] - a;
      const size_t one = (size_t)len * nd->ld * nd->elem();              // one variable of the segment
      const size_t segOff = (size_t)3 * a * nd->ld * nd->elem();         // the segment in a shard's planes
      char* p = (char*)nd->planes[k] + segOff;
      const int32_t nLoc = step0 + a + len <= have ? len : have - (step0 + a);
      if (sumsInKernel) {
        const int32_t rows = redCuts[j + 1] - redCuts[j];
        const size_t redOff = (size_t)3 * redCuts[j] * nd->ld * sizeof(double), oneRed = (size_t)rows * nd->ld;
        double* r = (double*)((char*)nd->reduced[k] + redOff);
        if (nLoc < len) NODE_HIP(hipMemsetAsync(r, 0, 3 * oneRed * sizeof(double), nd->streams[k]));   // (groups past a shard's last record: zero)
        if (nLoc > 0) {
          int rcr = sipnet_batch_run_sums(nd->batches[k], step0 + a, nLoc, sumSteps, r, r + oneRed, r + 2 * oneRed, nd->ld, nd->streams[k]);
          if (rcr) return rcr;
        }
        NODE_HIP(hipEventRecord(nd->evSeg[k], nd->streams[k]));
        NODE_HIP(hipStreamWaitEvent(nd->gatherStreams[k], nd->evSeg[k], 0));
        int rcg = allGatherShard(nd, k, r, (char*)nd->gatheredReduced[k] + (size_t)n * redOff, 3 * oneRed * sizeof(double), nd->gatherStreams[k]);
        if (rcg) return rcg;
        continue;
      }
      // (a site shard whose forcings end inside the segment: the rows past its end travel as zeros)
      if (nLoc > 0) {
        int rcr = sipnet_batch_run(nd->batches[k], step0 + a, nLoc, p, p + one, p + 2 * one, nullptr, nd->ld, nd->streams[k]);
        if (rcr) return rcr;   // (the barrier is told: the other shards give up at their next arrival)
      }
      NODE_HIP(hipEventRecord(nd->evSeg[k], nd->streams[k]));
      NODE_HIP(hipStreamWaitEvent(nd->gatherStreams[k], nd->evSeg[k], 0));
      if (!form) {
        int rcg = allGatherShard(nd, k, p, (char*)nd->gatheredPlanes[k] + (size_t)n * segOff, 3 * one, nd->gatherStreams[k]);
        if (rcg) return rcg;
        continue;
      }
      // the segment's reduced block [3][rows][ld], made on the second stream, then gathered there
      const int32_t rows = redCuts[j + 1] - redCuts[j];
      const size_t redOff = (size_t)3 * redCuts[j] * nd->ld * redElem, redLen = (size_t)3 * rows * nd->ld * redElem;
      char* r = (char*)nd->reduced[k] + redOff;
      hipStream_t gs = nd->gatherStreams[k];
      if (form == SIPNET_GATHER_F32) {
        const size_t n2 = (size_t)3 * len * nd->ld / 2;   // (ld is even)
        hipLaunchKernelGGL(planesToF32Kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, gs, (const double*)p, (float*)r, n2);
      } else {
        const dim3 grid((unsigned)((nd->ld + 255) / 256), (unsigned)rows, 3);
        if (nd->precision == SIPNET_F64)
          hipLaunchKernelGGL(sumStepsKernel<double>, grid, dim3(256), 0, gs, (const double*)p, (size_t)len * nd->ld, len, sumSteps, rows, nd->ld, (double*)r);
        else
          hipLaunchKernelGGL(sumStepsKernel<float>, grid, dim3(256), 0, gs, (const float*)p, (size_t)len * nd->ld, len, sumSteps, rows, nd->ld, (double*)r);
      }
      NODE_HIP(hipGetLastError());
      int rcg = allGatherShard(nd, k, r, (char*)nd->gatheredReduced[k] + (size_t)n * redOff, redLen, gs);
      if (rcg) return rcg;
    }
    return SIPNET_OK;
  });
  if (rc) return rc;
  // whatever follows on a shard's stream (the next run writes the same planes) comes after its gathers
  for (int k = 0; k < n; k++) {
    NODE_HIP(hipSetDevice(nd->devices[k]));
    NODE_HIP(hipEventRecord(nd->evGathered[k], nd->gatherStreams[k]));
    NODE_HIP(hipStreamWaitEvent(nd->streams[k], nd->evGathered[k], 0));
  }
  nd->nRun = n_steps;
  nd->step0 = step0;
  nd->segmented = true;
  nd->segCuts = cuts;
  nd->reducedForm = form;
  nd->reducedSumSteps = sumSteps;
  nd->redCuts = redCuts;
  return SIPNET_OK;
}

int sipnet_node_run_gathering(sipnet_node* nd, int32_t step0, int32_t n_steps, int32_t n_segments) {
  return runGathering(nd, step0, n_steps, n_segments);
}
int sipnet_node_run_gathering_reduced(sipnet_node* nd, int32_t step0, int32_t n_steps, int32_t n_segments, int32_t form,
                                      int32_t sum_steps) {
  if (form != SIPNET_GATHER_F32 && form != SIPNET_GATHER_SUMS) {
    setError("sipnet_node_run_gathering_reduced: form must be SIPNET_GATHER_F32 or SIPNET_GATHER_SUMS");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  return runGathering(nd, step0, n_steps, n_segments, form, sum_steps);
}
int32_t sipnet_node_reduced_in_kernel(const sipnet_node* nd) { return (nd && nd->segmented && nd->reducedForm && nd->reducedInKernel) ? 1 : 0; }
void* sipnet_node_gathered_reduced(sipnet_node* nd, int32_t k, int32_t segment, int32_t* first_row, int32_t* n_rows, int32_t* elem_bytes) {
  if (!nd || !nd->segmented || !nd->reducedForm || k < 0 || k >= nd->n() || segment < 0 || segment + 1 >= (int32_t)nd->redCuts.size())
    return nullptr;
  const size_t redElem = nd->reducedForm == SIPNET_GATHER_F32 ? sizeof(float) : sizeof(double);
  if (first_row) *first_row = nd->redCuts[segment];
  if (n_rows) *n_rows = nd->redCuts[segment + 1] - nd->redCuts[segment];
  if (elem_bytes) *elem_bytes = (int32_t)redElem;
  return (char*)nd->gatheredReduced[k] + (size_t)nd->n() * 3 * nd->redCuts[segment] * nd->ld * redElem;
}
int32_t sipnet_node_n_segments(const sipnet_node* nd) { return (nd && nd->segmented) ? (int32_t)nd->segCuts.size() - 1 : 0; }
void* sipnet_node_gathered_segment(sipnet_node* nd, int32_t k, int32_t segment, int32_t* first_step, int32_t* n_steps) {
  if (!nd || !nd->segmented || k < 0 || k >= nd->n() || segment < 0 || segment + 1 >= (int32_t)nd->segCuts.size()) return nullptr;
  if (first_step) *first_step = nd->step0 + nd->segCuts[segment];
  if (n_steps) *n_steps = nd->segCuts[segment + 1] - nd->segCuts[segment];
  return (char*)nd->gatheredPlanes[k] + (size_t)nd->n() * 3 * nd->segCuts[segment] * nd->ld * nd->elem();
}

int sipnet_node_run(sipnet_node* nd, int32_t step0, int32_t n_steps) { return runShards(nd, step0, n_steps, true); }
int sipnet_node_forecast(sipnet_node* nd, int32_t step0, int32_t n_steps) { return runShards(nd, step0, n_steps, false); }

int sipnet_node_sync(sipnet_node* nd) {
  if (!nd) return SIPNET_ERR_BAD_ARGUMENT;
  for (int k = 0; k < nd->n(); k++) {
    NODE_HIP(hipSetDevice(nd->devices[k]));
    NODE_HIP(hipStreamSynchronize(nd->streams[k]));
  }
  return SIPNET_OK;
}

int sipnet_node_get_status(sipnet_node* nd, int32_t* status) {
  if (!nd || !status) return SIPNET_ERR_BAD_ARGUMENT;
  std::vector<int32_t> loc;
  for (int k = 0; k < nd->n(); k++) {
    NODE_HIP(hipSetDevice(nd->devices[k]));
    loc.resize((size_t)nd->nSites[k] * nd->count[k]);
    int rc = sipnet_batch_get_status(nd->batches[k], loc.data(), nd->streams[k]);   // synchronises the shard's stream
    if (rc) return rc;
    for (int32_t s = 0; s < nd->nSites[k]; s++)
      memcpy(status + (size_t)(nd->sites0[k] + s) * nd->n_members + nd->first[k], loc.data() + (size_t)s * nd->count[k],
             (size_t)nd->count[k] * sizeof(int32_t));
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

// ONE all-gather: every shard's statistics block of the last sipnet_node_run to every shard
int sipnet_node_gather_stats(sipnet_node* nd, double* host_total) {
  if (!nd || nd->nRun <= 0 || nd->segmented) {
    setError("sipnet_node_gather_stats: nothing has run (sipnet_node_run_gathering leaves no statistics)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int n = nd->n();
  const size_t block = (size_t)3 * nd->nRun * nd->maxSites * 2;  // doubles per shard
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
  if (!nd->comms.empty()) {   // one thread, the per-device calls of the collective fused (RCCL's single-process idiom)
    NODE_RCCL(nd, nd->rccl->groupStart());
    for (int k = 0; k < n; k++) {
      NODE_HIP(hipSetDevice(nd->devices[k]));
      NODE_RCCL(nd, nd->rccl->allGather(nd->stats[k], nd->gatheredStats[k], block, ncclDouble, nd->comms[k], nd->streams[k]));
    }
    NODE_RCCL(nd, nd->rccl->groupEnd());
  } else {
    int rc = onEveryShard(nd, [&](int k) -> int {
      return allGatherShard(nd, k, nd->stats[k], nd->gatheredStats[k], block * sizeof(double));
    });
    if (rc) return rc;
  }
  if (host_total) {
    // the ensemble's totals [3][n_run][n_sites][2]: members sharded -- the shards' blocks added up in shard order
    // (deterministic); sites sharded -- shard k's sites stand at sites0[k] .. (concatenation along the site axis)
    std::vector<double> all(block * n);
    NODE_HIP(hipSetDevice(nd->devices[0]));
    NODE_HIP(hipStreamSynchronize(nd->streams[0]));   // shard 0's copy of everybody's block: complete when its all-gather is
    NODE_HIP(hipMemcpy(all.data(), nd->gatheredStats[0], all.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (nd->mode == SIPNET_SHARD_MEMBERS) {
      for (size_t i = 0; i < block; i++) {
        double s = 0.0;
        for (int k = 0; k < n; k++) s += all[(size_t)k * block + i];
        host_total[i] = s;
      }
    } else {
      for (int k = 0; k < n; k++)
        for (size_t vt = 0; vt < (size_t)3 * nd->nRun; vt++)
          memcpy(host_total + (vt * nd->n_sites + nd->sites0[k]) * 2, all.data() + (size_t)k * block + vt * nd->maxSites * 2,
                 (size_t)nd->nSites[k] * 2 * sizeof(double));
    }
  }
  return SIPNET_OK;
}

// ONE all-gather of the member-resolved planes of the last run: on every shard
// gathered[(k * 3 + v) * n_run + t][ld] = shard k's plane v; inside a row shard k's column of (its local site s,
// its member m) is s * count_k + m -- the site stride is the SHARD's member count, not the common maximum --
// and the columns from n_sites_k * count_k to ld are zero
int sipnet_node_gather_planes(sipnet_node* nd) {
  if (!nd || nd->nRun <= 0 || nd->segmented) {
    setError("sipnet_node_gather_planes: nothing has run (after sipnet_node_run_gathering the planes are gathered already)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int n = nd->n();
  const size_t count = (size_t)3 * nd->nRun * nd->ld;  // elements per shard
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
  if (!nd->comms.empty()) {
    NODE_RCCL(nd, nd->rccl->groupStart());
    for (int k = 0; k < n; k++) {
      NODE_HIP(hipSetDevice(nd->devices[k]));
      NODE_RCCL(nd, nd->rccl->allGather(nd->planes[k], nd->gatheredPlanes[k], count,
                                        nd->precision == SIPNET_F64 ? ncclDouble : ncclFloat, nd->comms[k], nd->streams[k]));
    }
    NODE_RCCL(nd, nd->rccl->groupEnd());
    return SIPNET_OK;
  }
  return onEveryShard(nd, [&](int k) -> int {
    return allGatherShard(nd, k, nd->planes[k], nd->gatheredPlanes[k], count * nd->elem());
  });
}

// ---- particle filter over the node's shards (BASELINE config C5; SURVEY 8(e) "PF extra exchange") --------------
int sipnet_node_pf_connect(sipnet_node* nd, int32_t with_params) {
  if (!nd || nd->mode != SIPNET_SHARD_MEMBERS || nd->n_sites != 1 || nd->n() > 16) {
    setError("sipnet_node_pf_connect: a filter needs ONE site, its particles sharded as members over at most 16 devices");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int n = nd->n();
  std::vector<sipnet_pf_peer> peers(n);
  int rc = onEveryShard(nd, [&](int k) -> int { return sipnet_batch_pf_publish(nd->batches[k], with_params, &peers[k]); });
  if (rc) return rc;
  rc = onEveryShard(nd, [&](int k) -> int {
    int r2 = sipnet_batch_pf_connect(nd->batches[k], n, k, peers.data());
    if (r2) return r2;
    NODE_HIP(hipStreamSynchronize(nd->streams[k]));
    if (nd->pfGathered[k]) NODE_HIP(hipFree(nd->pfGathered[k]));
    if (nd->pfAnc[k]) NODE_HIP(hipFree(nd->pfAnc[k]));
    if (nd->pfTotals[k]) NODE_HIP(hipFree(nd->pfTotals[k]));
    nd->pfGathered[k] = nullptr;
    nd->pfAnc[k] = nullptr;
    nd->pfTotals[k] = nullptr;
    const int64_t L = sipnet_batch_pf_block_len(nd->batches[k]);
    NODE_HIP(hipMalloc(&nd->pfGathered[k], (size_t)n * L * sizeof(double)));
    NODE_HIP(hipMalloc(&nd->pfAnc[k], (size_t)nd->count[k] * sizeof(int32_t)));
    NODE_HIP(hipMalloc(&nd->pfTotals[k], sipnet_node::kPfTotals * sizeof(int64_t)));
    NODE_HIP(hipMemset(nd->pfTotals[k], 0xff, sipnet_node::kPfTotals * sizeof(int64_t)));   // -1: "no cycle wrote this slot"
    return SIPNET_OK;
  });
  if (rc) return rc;
  nd->pfBlock = sipnet_batch_pf_block_len(nd->batches[0]);
  nd->pfConnected = true;
  nd->pfCycles = 0;
  return SIPNET_OK;
}

int sipnet_node_pf_arm(sipnet_node* nd, double obs, double sigma) {
  if (!nd || !nd->pfConnected || !(sigma > 0)) {
    setError("sipnet_node_pf_arm: needs sipnet_node_pf_connect and sigma > 0");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  // (host-only: every shard's next forecast launch is told where its log-weight block lies -- its slice of the all-gather's
  // buffer -- and what it will be weighed against)
  for (int k = 0; k < nd->n(); k++) {
    int rc = sipnet_batch_pf_arm(nd->batches[k], obs, sigma, nd->pfGathered[k] + (size_t)k * nd->pfBlock);
    if (rc) return rc;
  }
  return SIPNET_OK;
}

int sipnet_node_pf_analysis(sipnet_node* nd, int32_t variable, double obs, double sigma, double u0) {
  if (!nd || !nd->pfConnected || nd->nRun <= 0 || variable < 0 || variable > 2) {
    setError("sipnet_node_pf_analysis: needs sipnet_node_pf_connect and a forecast (sipnet_node_forecast / _run)");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (nd->segmented) {   // the planes lie segment by segment ([3][len_j][ld] each): a plane is not one [n_run][ld] block
    setError("sipnet_node_pf_analysis: the last run was sipnet_node_run_gathering (segmented planes); forecast with "
             "sipnet_node_forecast / sipnet_node_run");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const int slot = nd->pfCycles % sipnet_node::kPfTotals;
  int rc = onEveryShard(nd, [&](int k) -> int {
    const char* plane = (const char*)nd->planes[k] + (size_t)variable * nd->nRun * nd->ld * nd->elem();
    double* mine = nd->pfGathered[k] + (size_t)k * nd->pfBlock;
    int r2 = sipnet_batch_pf_local_weights(nd->batches[k], plane, nd->precision == SIPNET_F32_MIXED, nd->nRun, nd->ld, obs,
                                           sigma, mine, nd->streams[k]);
    if (r2) return r2;
    r2 = allGatherShard(nd, k, mine, nd->pfGathered[k], (size_t)nd->pfBlock * sizeof(double));   // in place
    if (r2) return r2;
    return sipnet_batch_pf_resample_peers(nd->batches[k], nd->pfGathered[k], u0, nd->pfAnc[k], nd->pfTotals[k] + slot,
                                          nd->streams[k]);
  });
  if (rc) return rc;
  nd->pfCycles++;
  return SIPNET_OK;
}

int sipnet_node_pf_check(sipnet_node* nd, int32_t* n_cycles_checked) {
  if (!nd || !nd->pfConnected) return SIPNET_ERR_BAD_ARGUMENT;
  const int m = std::min<int>(nd->pfCycles, sipnet_node::kPfTotals);
  std::vector<int64_t> t0(sipnet_node::kPfTotals), tk(sipnet_node::kPfTotals);
  for (int k = 0; k < nd->n(); k++) {
    NODE_HIP(hipSetDevice(nd->devices[k]));
    NODE_HIP(hipStreamSynchronize(nd->streams[k]));
    NODE_HIP(hipMemcpy(tk.data(), nd->pfTotals[k], tk.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (k == 0) t0 = tk;
    for (int c = 0; c < m; c++) {
      if (tk[c] == SIPNET_PF_VOID_TOTAL) {
        setError("sipnet_node_pf_check: a cycle's analysis kernel gave up at its grid barrier on shard " + std::to_string(k) +
                 " (its workgroups were not all resident: something else held device " + std::to_string(nd->devices[k]) +
                 "); that cycle's resampling is void");
        return SIPNET_ERR_INTERNAL;
      }
      if (tk[c] != t0[c]) {
        setError("sipnet_node_pf_check: the shards disagree on a cycle's total weight (the gathered blocks differ)");
        return SIPNET_ERR_INTERNAL;
      }
      if (tk[c] <= 0) {
        setError("sipnet_node_pf_check: a cycle ended with every particle at zero weight");
        return SIPNET_ERR_BAD_PARAMETER;
      }
    }
  }
  if (n_cycles_checked) *n_cycles_checked = m;
  nd->pfCycles = 0;
  return SIPNET_OK;
}

int32_t* sipnet_node_pf_ancestors(sipnet_node* nd, int32_t k) { return (nd && k >= 0 && k < nd->n()) ? nd->pfAnc[k] : nullptr; }
int64_t sipnet_node_pf_block_len(const sipnet_node* nd) { return nd ? nd->pfBlock : 0; }

}  // extern "C"
