// sipnet_cli.cpp -- `sipnet`-compatible command line over the C-ABI (drop-in at the process
// level, the boundary PEcAn uses: one working directory with sipnet.in / <prefix>.param /
// <prefix>.clim / <events>.in, producing <prefix>.out, <events>.out, <prefix>.config).
//
// Behaviour mirrored from the reference (paths relative to /root/reference/src/):
//   option table and precedence     sipnet/cli.c:20-57,144-229; common/context.c:17-25,143-170
//   sipnet.in syntax                sipnet/frontend.c:35-128
//   flag coupling rules             common/context.c:195-223
//   derived file names, run order   sipnet/frontend.c:130-253
//   --dump-config format            common/context.c:225-268
//   single-variable outputs         sipnet/outputItems.c:126-150, sipnet/sipnet.c:1993-1998
//   restart checkpoints             sipnet/sipnet.c:1963-1989, sipnet/restart.c:932-996
// The model itself runs on the GPU through libsipnet_amd.so; there is no CPU model here.
//
// Additive extension (never changes single-run behaviour):
//   --devices LIST           HIP devices the ensemble shards across (0-7, 0,2,3; default 0): member
//                            rows are split into contiguous ranges, one host thread + one batch
//                            per device, every shard writes its own members' files; with --sites:
//                            the sites of a flag set in contiguous ranges, one batch per device
//   --math strict|fast|auto  step-kernel arithmetic: strict = the reference's operation order (the
//                            default for a single run), fast = the throughput kernels (the default
//                            with --ensemble-params)
//   --ensemble-params FILE   whitespace table, first line = parameter names, one row per
//                            member overriding those parameters; every member runs in ONE
//                            batch and writes <prefix>.<m>.out (m = 0..M-1); restart paths
//                            get the same .<m> suffix; the members' text files are written
//                            by a pool of host threads
//   --ensemble-stats FILE    (with --ensemble-params) the ensemble's per-step statistics instead of
//                            the members' files: the members shard over --devices as ONE sipnet_node
//                            (one RCCL rank per device), every device runs its members on the
//                            throughput kernels, ONE all-gather of the statistics block joins them,
//                            and FILE gets `year day time n mean/sd of NEE, GPP, ET` per step
//   --ensemble-out FILE      (with --ensemble-params or --sites) every member's outputs as ONE NetCDF-3 block
//                            (include/sipnet_amd.h "ensemble output block": dimensions time x member, the `.out`
//                            columns' names and units) INSTEAD of the members' text files: nee, gpp and
//                            evapotranspiration by default -- the lean throughput kernels, three planes -- or the
//                            `.out` columns named with --ensemble-out-columns a,b,c|all (the 44-column record);
//                            --ensemble-out-f32 stores floats; --ensemble-text writes the text files as well;
//                            --ensemble-out-sums K (the three planes only): every member's SUMS over groups of K steps
//                            instead of the steps (K = 48: daily NEE / GPP / ET of a half-hourly forcing; the block's
//                            time axis holds each group's first record, its step length the group's) -- summed inside
//                            the step kernel's launch where the batch has such a kernel (sipnet_batch_run_sums), else
//                            from the planes on the host;
//                            --ensemble-out-segment N holds N steps of the record on the device at a time (default:
//                            about 3 GB worth -- the whole record of 10 240 members x a year would be 63 GB).
//                            Device shards stream their member ranges into the one file.  With --sites one block
//                            per distinct forcing: FILE itself when there is one, else <stem>.<k><ext> with k = the
//                            position in the list of the site's first run; member = position in the list, the global
//                            attribute run_dirs names the directories
//   --bounded-waits          (throughput kernels) the cooperative kernels' build whose hand-over waits have a budget of polls:
//                            a wait that never ends is reported (SIPNET_ERR_INTERNAL, exit 7, naming the wait and step)
//                            instead of hanging the GPU -- ~10 % slower, same bits; for boxes where a hang costs the machine
//   --sites FILE             stacking at the process boundary PEcAn uses: FILE lists run directories (one per
//                            line, `#` comments), each with its own sipnet.in / <prefix>.param / <prefix>.clim /
//                            <events>.in.  Every directory is resolved exactly like a run started inside it (the
//                            options of THIS command line take precedence over each sipnet.in, cli.c:144-229), runs
//                            whose forcing (climate + events) is identical become members of one site, runs with the
//                            same model flags share ONE batch (whatever their lengths), and every directory gets the files its
//                            own run would have written (<prefix>.out, <events>.out, <prefix>.config, single-variable
//                            outputs) -- byte for byte with --math auto / strict (the strict-order kernel), to the last
//                            printed digit with --math fast (the throughput kernels).  RESTART_IN / RESTART_OUT of a
//                            directory's sipnet.in are honoured (PEcAn's assimilation cycles: one directory per member
//                            and cycle, sipnet.c:1963-1989, restart.c:932-996): runs share a site only when their
//                            checkpoints agree in what the site plan owns (year-to-date GDD, year counters, tillage
//                            modifier, ring layout, processed steps), every member resumes from its own checkpoint
//                            and is checkpointed at the end of ITS forcing
#include <getopt.h>
#include <unistd.h>
#include <cmath>
#include <sched.h>
#include <strings.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sipnet_amd.h"

namespace {

enum Source { SRC_DEFAULT = 0, SRC_FILE, SRC_CLI, SRC_CALCULATED };
const char* sourceName(int s) {
  static const char* n[] = {"DEFAULT", "INPUT_FILE", "COMMAND_LINE", "CALCULATED"};
  return n[s];
}

struct Entry {
  std::string printName;
  bool isInt;
  int ival;
  std::string sval;
  int source;
};

// context.c:76-90
std::string keyOf(const std::string& name) {
  std::string k;
  for (char c : name)
    if (isalnum((unsigned char)c)) k += (char)tolower((unsigned char)c);
  if (k == "filename") k = "fileprefix";
  return k;
}

struct Context {
  std::map<std::string, Entry> m;  // ordered by key == HASH_SORT(by_name)
  void addInt(const char* name, const char* print, int v) { m[keyOf(name)] = {print, true, v, "", SRC_DEFAULT}; }
  void addStr(const char* name, const char* print, const char* v) { m[keyOf(name)] = {print, false, 0, v, SRC_DEFAULT}; }
  Entry* find(const std::string& name) {
    auto it = m.find(keyOf(name));
    return it == m.end() ? nullptr : &it->second;
  }
  void setInt(const std::string& name, int v, int src) {
    Entry* e = find(name);
    if (e && e->source <= src) { e->ival = v; e->source = src; }
  }
  void setStr(const std::string& name, const std::string& v, int src) {
    Entry* e = find(name);
    if (e && e->source <= src) { e->sval = v; e->source = src; }
  }
  int i(const char* name) { return find(name)->ival; }
  const std::string& s(const char* name) { return find(name)->sval; }
};

bool g_quiet = false;
int32_t g_kernelOptions = 0;   // SIPNET_KOPT_* for every batch this process creates (--bounded-waits)
void logInfo(const std::string& s) { if (!g_quiet) printf("[INFO   ] %s", s.c_str()); }
void logError(const std::string& s) { printf("[ERROR  ] %s", s.c_str()); }
void logWarning(const std::string& s) { printf("[WARNING] %s", s.c_str()); }

void initContext(Context& c) {  // context.c:26-71
  c.addInt("events", "EVENTS", 1);
  c.addInt("gdd", "GDD", 1);
  c.addInt("growthResp", "GROWTH_RESP", 0);
  c.addInt("leafWater", "LEAF_WATER", 0);
  c.addInt("litterPool", "LITTER_POOL", 0);
  c.addInt("snow", "SNOW", 1);
  c.addInt("soilPhenol", "SOIL_PHENOL", 0);
  c.addInt("waterHResp", "WATER_HRESP", 1);
  c.addInt("nitrogenCycle", "NITROGEN_CYCLE", 0);
  c.addInt("anaerobic", "ANAEROBIC", 0);
  c.addInt("flooding", "FLOODING", 0);
  c.addInt("carbonSaturation", "CARBON_SATURATION", 0);
  c.addInt("doMainOutput", "DO_MAIN_OUTPUT", 1);
  c.addInt("doSingleOutputs", "DO_SINGLE_OUTPUT", 0);
  c.addInt("dumpConfig", "DUMP_CONFIG", 0);
  c.addInt("printHeader", "PRINT_HEADER", 1);
  c.addInt("quiet", "QUIET", 0);
  c.addStr("paramFile", "PARAM_FILE", "");
  c.addStr("climFile", "CLIM_FILE", "");
  c.addStr("outFile", "OUT_FILE", "");
  c.addStr("outConfigFile", "OUT_CONFIG_FILE", "");
  c.addStr("eventsPrefix", "EVENTS_PREFIX", "events");
  c.addStr("inputFile", "INPUT_FILE", "sipnet.in");
  c.addStr("restartIn", "RESTART_IN", "");
  c.addStr("restartOut", "RESTART_OUT", "");
  c.addStr("debugLogPrefix", "DEBUG_LOG_PREFIX", "");
  c.addStr("filePrefix", "FILE_PREFIX", "sipnet");
}

const char* kFlagOpts[][2] = {  // cli.c:20-44 option name -> context field
    {"events", "events"}, {"gdd", "gdd"}, {"growth-resp", "growthResp"},
    {"leaf-water", "leafWater"}, {"litter-pool", "litterPool"}, {"snow", "snow"},
    {"soil-phenol", "soilPhenol"}, {"water-hresp", "waterHResp"},
    {"nitrogen-cycle", "nitrogenCycle"}, {"anaerobic", "anaerobic"}, {"flooding", "flooding"},
    {"carbon-saturation", "carbonSaturation"}, {"do-main-output", "doMainOutput"},
    {"do-single-outputs", "doSingleOutputs"}, {"dump-config", "dumpConfig"},
    {"print-header", "printHeader"}, {"quiet", "quiet"}};
constexpr int kNumFlagOpts = sizeof(kFlagOpts) / sizeof(kFlagOpts[0]);

void usage(const char* prog) {
  printf("Usage: %s [OPTIONS]\n\n", prog);
  printf("Run SIPNET model for one site with configured options (MI355X engine).\n\n");
  printf("  -i, --input-file <path>     Name of input config file ('sipnet.in')\n");
  printf("  -f, --file-prefix <name>    Prefix of climate and parameter files ('sipnet')\n");
  printf("      --file-name <name>      Backward-compatible alias for --file-prefix\n");
  printf("  -e, --events-prefix <name>  Prefix of events input/output files ('events')\n");
  printf("Model flags (prepend 'no-' to force off): --anaerobic --events --flooding --gdd\n");
  printf("  --growth-resp --leaf-water --litter-pool --nitrogen-cycle --snow --soil-phenol\n");
  printf("  --water-hresp --carbon-saturation\n");
  printf("Output flags: --do-main-output --do-single-outputs --dump-config --print-header --quiet\n");
  printf("      --restart-in <path>     Read a restart checkpoint from path\n");
  printf("      --restart-out <path>    Write a restart checkpoint to path at end of run\n");
  printf("  --ensemble-params <file>    run one member per row of a parameter table in one batch\n");
  printf("  --math strict|fast|auto     arithmetic of the step kernel (auto: strict for one run, fast for an ensemble)\n");
  printf("  --devices <list>            HIP devices the ensemble shards across, e.g. 0-7 or 0,2,3 ('0')\n");
  printf("  --ensemble-stats <file>     per-step ensemble mean / sd of NEE, GPP, ET instead of the members' files\n");
  printf("  --sites <file>              run every directory listed in <file> (one per line) in shared batches\n");
  printf("  --ensemble-out <file.nc>    (--ensemble-params / --sites) all members' outputs as one NetCDF-3 block instead of\n");
  printf("                              the members' text files: nee, gpp, evapotranspiration -- or, with\n");
  printf("      --ensemble-out-columns <a,b,..|all>  the named .out columns; --ensemble-out-f32 stores floats;\n");
  printf("      --ensemble-out-sums <K>  (planes only) every member's sums over groups of K steps (48: daily sums);\n");
  printf("      --ensemble-text         writes the text files as well\n");
  printf("  --bounded-waits             cooperative kernels with bounded hand-over waits (a stuck wait is reported, not a hang)\n");
  printf("  -h, --help   -v, --version\n");
}

// A fatal condition ends the process with the reference's exit code -- from the main thread.  Host threads (device
// shards, file writers) must not call exit() while other threads hold batches: they throw, the first error is kept, and
// the main thread reports it after the joins.
struct Fatal {
  int code;
  std::string msg;
};
thread_local bool t_worker = false;
[[noreturn]] void die(int code, const std::string& msg) {
  if (t_worker) throw Fatal{code, msg};
  logError(msg);
  exit(code);
}
struct FirstFatal {
  std::mutex mu;
  bool set = false;
  Fatal f{0, ""};
  void take(const Fatal& x) {
    std::lock_guard<std::mutex> lock(mu);
    if (!set) {
      set = true;
      f = x;
    }
  }
  void exitIfSet() {   // main thread, after the joins
    if (set) {
      logError(f.msg);
      exit(f.code);
    }
  }
};
template <class F>
void guarded(FirstFatal& sink, F&& fn) {   // body of a host thread
  t_worker = true;
  try {
    fn();
  } catch (const Fatal& f) {
    sink.take(f);
  }
}
// run `fn` on n host threads (inline when n == 1); a fatal condition in any of them ends the process afterwards
template <class F>
void runThreads(int n, F&& fn) {
  if (n <= 1) {
    fn();
    return;
  }
  FirstFatal sink;
  std::vector<std::thread> pool;
  for (int i = 0; i < n; i++) pool.emplace_back([&]() { guarded(sink, fn); });
  for (auto& th : pool) th.join();
  if (sink.set) die(sink.f.code, sink.f.msg);   // (rethrown when this is itself a worker thread)
}
// a path named in a sipnet.in, seen from outside its directory
std::string joinPath(const std::string& dir, const std::string& p) {
  return (!p.empty() && p[0] == '/') ? p : dir + "/" + p;
}

// frontend.c:35-128
void readInputFile(Context& c) {
  const std::string path = c.s("inputFile");
  logInfo("Reading config from file " + path + "\n");
  std::ifstream in(path);
  if (!in) die(6, "Error opening " + path + " for reading\n");
  std::string line;
  while (std::getline(in, line)) {
    const size_t bang = line.find('!');
    if (bang != std::string::npos) line.erase(bang);
    // name: up to " \t=:" ; value: up to " \t=:\n\r"
    std::vector<std::string> tok;
    size_t i = 0;
    while (i < line.size() && tok.size() < 2) {
      while (i < line.size() && strchr(" \t=:\r\n", line[i])) i++;
      size_t j = i;
      while (j < line.size() && !strchr(" \t=:\r\n", line[j])) j++;
      if (j > i) tok.push_back(line.substr(i, j - i));
      i = j;
    }
    if (tok.empty()) continue;
    if (strcasecmp(tok[0].c_str(), "runtype") == 0) {
      if (tok.size() > 1 && strcasecmp(tok[1].c_str(), "standard") != 0)
        die(3, "RUNTYPE is obsolete; only 'standard' is accepted. Please fix " + path + " and re-run\n");
      continue;
    }
    Entry* e = c.find(tok[0]);
    if (!e) {
      logInfo("ignoring input file parameter " + tok[0] + "\n");
      continue;
    }
    if (tok.size() < 2)
      die(3, "Error in input file: No value given for input item " + tok[0] + "\n");
    if (e->isInt) {
      char* end = nullptr;
      const long v = strtol(tok[1].c_str(), &end, 0);
      if (*end) die(3, "ERROR in input file: Invalid value for " + tok[0] + ": " + tok[1] + "\n");
      c.setInt(tok[0], (int)v, SRC_FILE);
    } else {
      c.setStr(tok[0], tok[1] == "none" ? "" : tok[1], SRC_FILE);
    }
  }
}

// context.c:225-268
void printConfig(Context& c, FILE* f) {
  unsigned width = 0;
  for (auto& kv : c.m)
    if (!kv.second.isInt) width = std::max<unsigned>(width, (unsigned)kv.second.printName.size());
  if (c.i("printHeader")) {
    char ts[100];
    time_t now;
    time(&now);
    strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S UTC", gmtime(&now));
    fprintf(f, "Final config for SIPNET run at %s\n", ts);
    fprintf(f, "%21s %13s %*s\n", "Name", "Source", width, "Value");
  }
  for (auto& kv : c.m) {
    const Entry& e = kv.second;
    if (e.isInt)
      fprintf(f, "%21s %13s %*d\n", e.printName.c_str(), sourceName(e.source), width, e.ival);
    else
      fprintf(f, "%21s %13s %*s\n", e.printName.c_str(), sourceName(e.source), width, e.sval.c_str());
  }
}

// "--devices 0-7", "0,2,3", "0,0" (two shards on one device): HIP device ordinals, one shard each
std::vector<int> parseDevices(const std::string& arg) {
  std::vector<int> out;
  std::istringstream in(arg);
  for (std::string tok; std::getline(in, tok, ',');) {
    if (tok.empty()) continue;
    const size_t dash = tok.find('-', 1);
    char* end = nullptr;
    const long a = strtol(tok.c_str(), &end, 10);
    long z = a;
    if (dash != std::string::npos) {
      if (end != tok.c_str() + dash) die(8, "bad --devices list: " + arg + "\n");
      z = strtol(tok.c_str() + dash + 1, &end, 10);
    }
    if (*end || a < 0 || z < a || z - a > 1023) die(8, "bad --devices list: " + arg + "\n");
    for (long d = a; d <= z; d++) out.push_back((int)d);
  }
  if (out.empty()) die(8, "bad --devices list: " + arg + "\n");
  return out;
}

double nowSeconds() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void check(int rc, const char* what) {
  if (rc != SIPNET_OK) die(rc >= 100 ? 1 : rc, std::string(what) + ": " + sipnet_last_error() + "\n");
}

// ---- the ensemble output block (--ensemble-out) ---------------------------------------------------------------
struct BlockSpec {
  std::string path;                // empty: no block
  std::vector<int> cols;           // `.out` column indices (sipnet_io_out_column); empty: the three planes
  bool f32 = false, text = false;  // store floats; write the members' text files as well
  int segment = 0;                 // --ensemble-out-segment: steps of the record held on the device at a time (0: ~3 GB worth)
  int sums = 0;                    // --ensemble-out-sums K: sums over groups of K steps instead of the steps (planes only)
  bool on() const { return !path.empty(); }
  bool planesOnly() const { return cols.empty(); }
};
void parseBlockColumns(BlockSpec& spec, const std::string& arg) {
  if (arg == "all") {
    for (int k = 0; k < sipnet_io_out_column_count(); k++) spec.cols.push_back(k);
    return;
  }
  std::istringstream in(arg);
  for (std::string tok; std::getline(in, tok, ',');) {
    if (tok.empty()) continue;
    const int k = sipnet_io_out_column_index(tok.c_str());
    if (k < 0) die(8, "--ensemble-out-columns: unknown .out column " + tok + "\n");
    spec.cols.push_back(k);
  }
  if (spec.cols.empty()) die(8, "--ensemble-out-columns: no columns\n");
}
sipnet_ensemble_file* createBlock(const BlockSpec& spec, const std::string& path, int T, int M, const sipnet_clim_table* clim,
                                  const int32_t* memberIds, const std::string& attrs) {
  std::vector<const char*> names;
  if (spec.planesOnly()) {
    names = {"nee", "gpp", "evapotranspiration"};
  } else {
    for (int k : spec.cols) {
      const char* nm = nullptr;
      check(sipnet_io_out_column(k, &nm, nullptr, nullptr, nullptr), "column table");
      names.push_back(nm);
    }
  }
  sipnet_ensemble_file* f = nullptr;
  check(sipnet_io_ensemble_create(path.c_str(), T, M, sipnet_clim_year(clim), sipnet_clim_day(clim), sipnet_clim_data(clim),
                                  memberIds, (int32_t)names.size(), names.data(), nullptr, spec.f32 ? SIPNET_NC_F32 : SIPNET_NC_F64,
                                  attrs.c_str(), &f), "creating the ensemble block");
  return f;
}
// members [col0, col0 + n) of a device result -> members [member0, member0 + n) of the block, one variable at a time
// (a dense [T][n] host array per variable: nothing the size of the record ever exists on the host).
// planes: dPlanes[3][Tld][ld] (NEE, GPP, ET); records: dRec[Tld][SIPNET_NREC][ld].
// (step0: the file rows the device arrays' T rows go to -- a run cut into segments fills the block segment by segment)
void putBlock(sipnet_ensemble_file* f, const BlockSpec& spec, int T, int64_t ld, int64_t Tld, int64_t col0, int n, int member0,
              const double* dPlanes, const double* dRec, int step0 = 0) {
  std::vector<double> host((size_t)T * n), second;
  const size_t w = (size_t)n * sizeof(double), dense = (size_t)T * w;
  double tFetch = 0.0, tPut = 0.0;
  // a column of the device result is T pieces of n doubles, ld (planes) or SIPNET_NREC x ld (records) doubles apart: gathered
  // into a dense device array first (a device-side copy), then ONE dense copy to the host -- row pieces straight over
  // PCIe took 3 s per column of 10 240 members x 17 520 steps, this takes 0.15 s
  const bool strided = spec.planesOnly() ? ld != n : true;
  double* dDense = strided ? (double*)sipnet_dev_alloc(dense) : nullptr;
  if (strided && !dDense) die(1, std::string(sipnet_last_error()) + "\n");
  auto fetch = [&](std::vector<double>& dst, const double* src, size_t pitch) {
    const double t0 = nowSeconds();
    if (!strided) {
      check(sipnet_dev_to_host(dst.data(), src, dense, nullptr), "copy back");
    } else {
      check(sipnet_dev_to_dev_2d(dDense, w, src, pitch, w, (size_t)T, nullptr), "gathering a column");
      check(sipnet_dev_to_host(dst.data(), dDense, dense, nullptr), "copy back");
    }
    tFetch += nowSeconds() - t0;
  };
  auto put = [&](int v) {
    const double t0 = nowSeconds();
    check(sipnet_io_ensemble_put(f, v, step0, T, member0, n, host.data(), n, 0), "writing the ensemble block");
    tPut += nowSeconds() - t0;
  };
  if (spec.planesOnly()) {
    for (int v = 0; v < 3; v++) {
      fetch(host, dPlanes + ((size_t)v * Tld) * ld + col0, (size_t)ld * sizeof(double));
      put(v);
    }
  } else {
    const size_t pitch = (size_t)SIPNET_NREC * ld * sizeof(double);
    for (size_t v = 0; v < spec.cols.size(); v++) {
      int32_t r0 = 0, r1 = -1;
      check(sipnet_io_out_column(spec.cols[v], nullptr, &r0, &r1, nullptr), "column table");
      fetch(host, dRec + (size_t)r0 * ld + col0, pitch);
      if (r1 >= 0) {   // total wood = plantWoodC + accounting delta (state.c:17-19)
        second.resize(host.size());
        fetch(second, dRec + (size_t)r1 * ld + col0, pitch);
        for (size_t i = 0; i < host.size(); i++) host[i] += second[i];
      }
      put((int)v);
    }
  }
  if (dDense) sipnet_dev_free(dDense);
  char msg[200];
  snprintf(msg, sizeof msg, "ensemble block: members %d..%d, steps %d..%d, %zu variable(s): device -> host %.2f s, conversion + file %.2f s\n",
           member0, member0 + n - 1, step0, step0 + T - 1, spec.planesOnly() ? (size_t)3 : spec.cols.size(), tFetch, tPut);
  logInfo(msg);
}

// ---- --sites: many run directories, few batches ---------------------------------------------------------------
struct SiteRun {
  std::string dir;           // absolute
  Context ctx;               // resolved like a run started inside dir
  int32_t flags[SIPNET_NFLAGS];
  std::vector<double> params;
  sipnet_clim_table* clim = nullptr;
  sipnet_event* events = nullptr;
  int32_t nEvents = 0;
  int T = 0;
  bool hasResume = false;    // RESTART_IN: the checkpoint this run resumes from (restartLoadCheckpoint, restart.c:968-996)
  sipnet_restart resume;
  std::string restartOut;    // RESTART_OUT as an absolute path ("" = none)
};

// what the site plan takes from a checkpoint (sipnet_batch_set_resume): members of one site must agree in it
bool samePlanCarry(const sipnet_restart& a, const sipnet_restart& b) {
  return a.trackers[SIPNET_RT_GDD] == b.trackers[SIPNET_RT_GDD] && a.trackers_last_year == b.trackers_last_year &&
         a.phenology_last_year == b.phenology_last_year && a.d_till_mod == b.d_till_mod && a.mean_start == b.mean_start &&
         a.mean_last == b.mean_last && a.processed_steps == b.processed_steps &&
         memcmp(a.mean_weights, b.mean_weights, sizeof a.mean_weights) == 0;
}

bool sameForcing(const SiteRun& a, const SiteRun& b) {
  if (a.T != b.T || a.nEvents != b.nEvents || a.hasResume != b.hasResume) return false;
  if (a.hasResume && !samePlanCarry(a.resume, b.resume)) return false;
  if (memcmp(sipnet_clim_data(a.clim), sipnet_clim_data(b.clim), (size_t)a.T * SIPNET_NCLIM * sizeof(double)) != 0) return false;
  if (memcmp(sipnet_clim_year(a.clim), sipnet_clim_year(b.clim), (size_t)a.T * sizeof(int32_t)) != 0) return false;
  if (memcmp(sipnet_clim_day(a.clim), sipnet_clim_day(b.clim), (size_t)a.T * sizeof(int32_t)) != 0) return false;
  for (int k = 0; k < a.nEvents; k++) {
    const sipnet_event &x = a.events[k], &y = b.events[k];
    if (x.type != y.type || x.year != y.year || x.day != y.day || memcmp(x.p, y.p, sizeof x.p) != 0) return false;
  }
  return true;
}

// restartLoadCheckpoint's checks (restart.c:968-996) with the reference's log lines
void checkResume(const sipnet_restart& r, const std::string& path, const int32_t* flags, const sipnet_clim_table* clim) {
  const int T = sipnet_clim_nsteps(clim);
  const double* c0 = sipnet_clim_data(clim);
  int32_t warn = 0;
  check(sipnet_restart_check(&r, flags, T > 0, T > 0 ? sipnet_clim_year(clim)[0] : 0, T > 0 ? sipnet_clim_day(clim)[0] : 0,
                             T > 0 ? c0[10] : 0.0, T > 0 ? c0[0] : 0.0, &warn), "restart checkpoint");
  if (warn & SIPNET_RESTART_WARN_BOUNDARY_NOT_MIDNIGHT)
    logWarning("Restart checkpoint boundary in " + path + " is more than one timestep before "
               "midnight; there is a time gap on resume.\n");
  if (warn & SIPNET_RESTART_WARN_BUILD_INFO)
    logInfo(std::string("Restart build info mismatch: checkpoint=") + r.build_info + "\n");
  if (warn & SIPNET_RESTART_WARN_TIME_GAP)
    logWarning("Restart resumed segment starts more than one timestep after midnight "
               "checkpoint boundary; there is a time gap\n");
}

// restartWriteCheckpoint (restart.c:932-996) of one member of a batch that has run its site to the end
void writeCheckpoint(sipnet_batch* b, std::mutex& gpuMutex, std::mutex& logMutex, int site, int member, int T, const double* lastRec,
                     const double* prevPools, const std::string& path) {
  sipnet_restart ck;
  {
    std::lock_guard<std::mutex> lock(gpuMutex);   // (the export talks to the GPU through the one batch handle)
    check(sipnet_batch_export_restart(b, site, member, T, lastRec, prevPools, &ck, nullptr), "restart checkpoint");
  }
  int32_t warn = 0;
  check(sipnet_restart_check_boundary_for_write(&ck, &warn), "restart checkpoint");
  if (warn & SIPNET_RESTART_WARN_BOUNDARY_NOT_MIDNIGHT) {
    std::lock_guard<std::mutex> lock(logMutex);
    logWarning("Restart checkpoint " + path + " ends more than one timestep before midnight; "
               "there will be a time gap if this file is used to resume.\n");
  }
  check(sipnet_io_write_restart(path.c_str(), &ck), "writing restart checkpoint");
}

// the part of main() between the command line and the run, for one directory (the process is inside it)
void resolveRun(SiteRun& r) {
  Context& ctx = r.ctx;
  if (ctx.s("filePrefix").empty()) die(3, "filePrefix must be set for SIPNET to run\n");
  readInputFile(ctx);
  bool bad = false;
  if (ctx.i("soilPhenol") && ctx.i("gdd")) { logError("soil-phenol and gdd may not both be turned on\n"); bad = true; }
  if (ctx.i("nitrogenCycle") && !(ctx.i("litterPool") && ctx.i("anaerobic"))) {
    logError("nitrogen-cycle requires both litter-pool and anaerobic to be turned on\n"); bad = true; }
  if (ctx.i("anaerobic") && !ctx.i("waterHResp")) { logError("anaerobic requires water-hresp to be turned on\n"); bad = true; }
  if (ctx.i("carbonSaturation") && !ctx.i("litterPool")) { logError("carbon-saturation requires litter-pool to be turned on\n"); bad = true; }
  if (bad) exit(3);
  if (!ctx.s("debugLogPrefix").empty()) die(8, "--sites does not combine with --debug-log (" + r.dir + ")\n");
  const std::string prefix = ctx.s("filePrefix");
  ctx.setStr("paramFile", prefix + ".param", SRC_CALCULATED);
  ctx.setStr("climFile", prefix + ".clim", SRC_CALCULATED);
  if (ctx.i("doMainOutput")) ctx.setStr("outFile", prefix + ".out", SRC_CALCULATED);
  if (ctx.i("dumpConfig")) {
    ctx.setStr("outConfigFile", prefix + ".config", SRC_CALCULATED);
    FILE* f = fopen((prefix + ".config").c_str(), "w");
    if (!f) die(6, "Error opening " + prefix + ".config for writing\n");
    printConfig(ctx, f);
    fclose(f);
  }
  const int32_t fl[SIPNET_NFLAGS] = {ctx.i("events"), ctx.i("gdd"), ctx.i("growthResp"), ctx.i("leafWater"), ctx.i("litterPool"),
                                     ctx.i("snow"), ctx.i("soilPhenol"), ctx.i("waterHResp"), ctx.i("nitrogenCycle"),
                                     ctx.i("anaerobic"), ctx.i("flooding"), ctx.i("carbonSaturation")};
  memcpy(r.flags, fl, sizeof fl);
  r.params.resize(SIPNET_NPARAMS);
  check(sipnet_io_read_params(ctx.s("paramFile").c_str(), r.flags, r.params.data(), nullptr), "reading parameters");
  check(sipnet_io_read_clim(ctx.s("climFile").c_str(), r.flags[SIPNET_F_GDD], &r.clim), "reading climate");
  r.T = sipnet_clim_nsteps(r.clim);
  if (ctx.i("events")) {
    check(sipnet_io_read_events((ctx.s("eventsPrefix") + ".in").c_str(), r.flags, r.params.data(), &r.events, &r.nEvents),
          "reading events");
    if (r.nEvents == 0) logInfo("No event file found, assuming no input events\n");
  }
  if (!ctx.s("restartIn").empty()) {
    check(sipnet_io_read_restart(ctx.s("restartIn").c_str(), &r.resume), "reading restart checkpoint");
    checkResume(r.resume, ctx.s("restartIn"), r.flags, r.clim);
    r.hasResume = true;
  }
  if (!ctx.s("restartOut").empty()) r.restartOut = joinPath(r.dir, ctx.s("restartOut"));
}

int runSites(const Context& cliCtx, const std::string& listFile, const std::string& mathArg, const std::vector<int>& devices,
             const BlockSpec& block) {
  std::vector<std::string> dirs;
  {
    std::ifstream in(listFile);
    if (!in) die(6, "Error opening " + listFile + " for reading\n");
    for (std::string line; std::getline(in, line);) {
      const size_t hash = line.find('#');
      if (hash != std::string::npos) line.erase(hash);
      const size_t a = line.find_first_not_of(" \t\r"), z = line.find_last_not_of(" \t\r");
      if (a == std::string::npos) continue;
      dirs.push_back(line.substr(a, z - a + 1));
    }
  }
  if (dirs.empty()) die(5, "no run directories in " + listFile + "\n");
  char cwd0[4096];
  if (!getcwd(cwd0, sizeof cwd0)) die(1, "getcwd failed\n");
  std::vector<SiteRun> runs(dirs.size());
  for (size_t k = 0; k < dirs.size(); k++) {
    if (chdir(cwd0) != 0 || chdir(dirs[k].c_str()) != 0) die(6, "cannot enter run directory " + dirs[k] + "\n");
    char here[4096];
    if (!getcwd(here, sizeof here)) die(1, "getcwd failed\n");
    runs[k].dir = here;
    runs[k].ctx = cliCtx;
    resolveRun(runs[k]);
  }
  if (chdir(cwd0) != 0) die(1, "cannot return to the start directory\n");
  for (int device : devices)
    if (device >= sipnet_device_count())
      die(1, "--devices names device " + std::to_string(device) + " but only " + std::to_string(sipnet_device_count()) +
                 " HIP device(s) are visible (this engine has no CPU path)\n");
  const bool fastMath = mathArg == "fast" || (mathArg == "auto" && block.on() && !block.text);
  const bool wantText = !block.on() || block.text;
  // batches: same model flags; inside a batch, runs with identical forcing (and resume state) are members of ONE site
  std::vector<char> done(runs.size(), 0);
  std::vector<std::vector<std::vector<int>>> groups;   // [batch][site][member] -> run index
  size_t nSitesTotal = 0;
  for (size_t lead = 0; lead < runs.size(); lead++) {
    if (done[lead]) continue;
    std::vector<std::vector<int>> sites;
    for (size_t k = lead; k < runs.size(); k++) {
      if (done[k] || memcmp(runs[k].flags, runs[lead].flags, sizeof runs[k].flags) != 0) continue;   // (any length)
      done[k] = 1;
      bool placed = false;
      for (auto& st : sites)
        if (sameForcing(runs[st[0]], runs[k])) {
          st.push_back((int)k);
          placed = true;
          break;
        }
      if (!placed) sites.push_back({(int)k});
    }
    nSitesTotal += sites.size();
    groups.push_back(std::move(sites));
  }
  auto blockPathOf = [&](int firstRun) -> std::string {   // one block per distinct forcing
    if (nSitesTotal == 1) return block.path;
    const size_t slash = block.path.find_last_of('/'), dot = block.path.find_last_of('.');
    const bool hasExt = dot != std::string::npos && (slash == std::string::npos || dot > slash);
    const std::string stem = hasExt ? block.path.substr(0, dot) : block.path, ext = hasExt ? block.path.substr(dot) : "";
    return stem + "." + std::to_string(firstRun) + ext;
  };
  int worst = 0, nBatches = 0;
  for (const auto& allSites : groups) {
    const size_t lead = (size_t)allSites[0][0];
    // the sites of a flag set are dealt to the listed devices in contiguous ranges (whole sites per device, as
    // SIPNET_SHARD_SITES does: a site's forcing, events and plan exist on one device only); each range is one batch,
    // driven by a host thread of its own
    const int nParts = std::max(1, std::min((int)devices.size(), (int)allSites.size()));
    std::vector<int> partWorst(nParts, 0);
    std::mutex batchMutex;
    int hostThreads = 1;
    {
      cpu_set_t cpus;
      if (sched_getaffinity(0, sizeof cpus, &cpus) == 0) hostThreads = CPU_COUNT(&cpus);
    }
    auto runPart = [&](int part) {
    const std::vector<std::vector<int>> sites(allSites.begin() + (size_t)allSites.size() * part / nParts,
                                              allSites.begin() + (size_t)allSites.size() * (part + 1) / nParts);
    const int device = devices[part];
    const int S = (int)sites.size();
    int M = 0, T = 0;   // T: the longest forcing of the batch (its sites may be shorter)
    bool anyCheckpoint = false;
    for (auto& st : sites) {
      M = std::max(M, (int)st.size());
      T = std::max(T, runs[st[0]].T);
      for (int k : st) anyCheckpoint = anyCheckpoint || !runs[k].restartOut.empty();
    }
    {
      std::lock_guard<std::mutex> lock(batchMutex);
      nBatches++;
      logInfo("batch " + std::to_string(nBatches) + " (device " + std::to_string(device) + "): " + std::to_string(S) +
              " site(s) x up to " + std::to_string(M) + " member(s), up to " + std::to_string(T) + " steps\n");
    }
    sipnet_batch* b = nullptr;
    check(sipnet_batch_create(runs[lead].flags, S, M, SIPNET_F64, device, &b), "creating batch");
    check(sipnet_batch_set_math(b, fastMath ? SIPNET_MATH_FAST : SIPNET_MATH_STRICT), "math policy");
    if (g_kernelOptions) check(sipnet_batch_set_kernel(b, SIPNET_KERNEL_AUTO, g_kernelOptions), "kernel options");
    std::vector<double> rows((size_t)M * SIPNET_NPARAMS);
    for (int s = 0; s < S; s++) {
      const SiteRun& r0 = runs[sites[s][0]];
      check(sipnet_batch_set_events(b, s, r0.nEvents, r0.events), "events");
      check(sipnet_batch_set_climate(b, s, r0.T, sipnet_clim_data(r0.clim), sipnet_clim_year(r0.clim), sipnet_clim_day(r0.clim)), "climate");
      for (int m = 0; m < M; m++) {   // (a site with fewer runs than the widest one: its first run's parameters fill the rest)
        const SiteRun& r = runs[sites[s][m < (int)sites[s].size() ? m : 0]];
        memcpy(rows.data() + (size_t)m * SIPNET_NPARAMS, r.params.data(), SIPNET_NPARAMS * sizeof(double));
      }
      check(sipnet_batch_set_params(b, s, 0, M, rows.data()), "parameters");
      if (r0.hasResume) check(sipnet_batch_set_resume(b, s, &r0.resume), "restart checkpoint");
    }
    check(sipnet_batch_setup(b, nullptr), "setupModel");
    for (int s = 0; s < S; s++) {   // every member from its own checkpoint (the filling columns repeat the first run's)
      if (!runs[sites[s][0]].hasResume) continue;
      std::vector<sipnet_restart> cks(M);
      for (int m = 0; m < M; m++) cks[m] = runs[sites[s][m < (int)sites[s].size() ? m : 0]].resume;
      check(sipnet_batch_import_restart(b, s, 0, M, cks.data(), nullptr), "restart checkpoint");
    }
    const int64_t ncol = (int64_t)S * M;
    std::vector<double> state0;
    if (wantText || anyCheckpoint) {
      state0.resize((size_t)ncol * SIPNET_NSTATE);
      check(sipnet_batch_get_state(b, state0.data(), nullptr), "state");
    }
    // the block alone, planes: the lean kernels and three planes; anything else needs the 44-column record
    const bool needRec = wantText || anyCheckpoint || (block.on() && !block.planesOnly());
    const size_t recElems = needRec ? (size_t)T * SIPNET_NREC * ncol : 0, planeElems = needRec ? 0 : (size_t)3 * T * ncol;
    double* dRec = recElems ? (double*)sipnet_dev_alloc(recElems * sizeof(double)) : nullptr;
    double* dPlanes = planeElems ? (double*)sipnet_dev_alloc(planeElems * sizeof(double)) : nullptr;
    if ((recElems && !dRec) || (planeElems && !dPlanes)) die(1, std::string(sipnet_last_error()) + "\n");
    if (needRec)
      check(sipnet_batch_run(b, 0, T, nullptr, nullptr, nullptr, dRec, ncol, nullptr), "run");
    else
      check(sipnet_batch_run(b, 0, T, dPlanes, dPlanes + (size_t)T * ncol, dPlanes + (size_t)2 * T * ncol, nullptr, ncol, nullptr), "run");
    std::vector<int32_t> status(ncol);
    check(sipnet_batch_get_status(b, status.data(), nullptr), "status");
    std::atomic<int> worstA{0};
    std::mutex logMutex, gpuMutex;
    auto reportStatus = [&](const SiteRun& r, int st) {
      std::lock_guard<std::mutex> lock(logMutex);
      logError(r.dir + ": status " + std::to_string(st) +
               " (NPP allocation params must be less than one individually and add to less than one)\n");
      int w = worstA.load();
      while (st > w && !worstA.compare_exchange_weak(w, st)) {}
    };
    if (block.on()) {   // one block per site: its runs are the members
      for (int s = 0; s < S; s++) {
        const SiteRun& r0 = runs[sites[s][0]];
        const int n = (int)sites[s].size();
        std::vector<int32_t> ids(sites[s].begin(), sites[s].end());
        std::string attrs = "run_dirs=";
        for (int m = 0; m < n; m++) attrs += (m ? " " : "") + runs[sites[s][m]].dir;
        attrs += "\nmath=" + std::string(fastMath ? "fast" : "strict");
        sipnet_ensemble_file* f = createBlock(block, blockPathOf(sites[s][0]), r0.T, n, r0.clim, ids.data(), attrs);
        putBlock(f, block, r0.T, ncol, T, (int64_t)s * M, n, 0, dPlanes, dRec);
        check(sipnet_io_ensemble_close(f), "closing the ensemble block");
        if (!wantText)
          for (int m = 0; m < n; m++)
            if (status[(size_t)s * M + m] != 0) reportStatus(runs[sites[s][m]], status[(size_t)s * M + m]);
      }
    }
    if (wantText || anyCheckpoint) {
    std::vector<double> rec(recElems);
    check(sipnet_dev_to_host(rec.data(), dRec, recElems * sizeof(double), nullptr), "copy back");
    // every run's files, into its own directory (absolute paths), by a pool of host threads
    struct Job { int run, site, member; int64_t col; };
    std::vector<Job> jobs;
    for (int s = 0; s < S; s++)
      for (int m = 0; m < (int)sites[s].size(); m++) jobs.push_back({sites[s][m], s, m, (int64_t)s * M + m});
    std::atomic<int> next{0};
    auto worker = [&]() {
      std::vector<double> one((size_t)T * SIPNET_NREC);
      for (int j = next.fetch_add(1); j < (int)jobs.size(); j = next.fetch_add(1)) {
        SiteRun& r = runs[jobs[j].run];
        const int64_t c = jobs[j].col;
        const int T = r.T;   // (this run's own length: shadows the batch's longest)
        if (status[c] != 0) {
          if (wantText) reportStatus(r, status[c]);
          continue;
        }
        for (int t = 0; t < T; t++)
          for (int k = 0; k < SIPNET_NREC; k++) one[(size_t)t * SIPNET_NREC + k] = rec[((size_t)t * SIPNET_NREC + k) * ncol + c];
        Context& ctx = r.ctx;
        const std::string prefix = joinPath(r.dir, ctx.s("filePrefix"));
        if (wantText && ctx.i("doMainOutput"))
          check(sipnet_io_write_out((prefix + ".out").c_str(), ctx.i("printHeader"), T, sipnet_clim_year(r.clim), sipnet_clim_day(r.clim),
                                    sipnet_clim_data(r.clim), one.data()), "writing output");
        if (wantText && ctx.i("events"))
          check(sipnet_io_write_events_out((joinPath(r.dir, ctx.s("eventsPrefix")) + ".out").c_str(), ctx.i("printHeader"), r.flags,
                                           r.params.data(), T, sipnet_clim_year(r.clim), sipnet_clim_day(r.clim),
                                           sipnet_clim_data(r.clim), r.nEvents, r.events, one.data(),
                                           state0.data() + (size_t)c * SIPNET_NSTATE), "writing events.out");
        if (!r.restartOut.empty()) {
          const double* prevPools = T >= 2 ? one.data() + (size_t)(T - 2) * SIPNET_NREC + 14 : state0.data() + (size_t)c * SIPNET_NSTATE;
          writeCheckpoint(b, gpuMutex, logMutex, jobs[j].site, jobs[j].member, T, one.data() + (size_t)(T - 1) * SIPNET_NREC, prevPools,
                          r.restartOut);
        }
        if (wantText && ctx.i("doSingleOutputs")) {  // sipnet.c:1993-1998, outputItems.c:126-150
          const struct { const char* name; int col; } items[] = {{"NEE", 0}, {"NEE_cum", 3}, {"GPP", 1}, {"GPP_cum", 35}};
          for (const auto& it : items) {
            FILE* f = fopen((prefix + "." + it.name).c_str(), "w");
            if (!f) die(6, std::string("Error opening single output file for ") + it.name + "\n");
            for (int t = 0; t < T; t++) fprintf(f, "%f ", one[(size_t)t * SIPNET_NREC + it.col]);
            fprintf(f, "\n");
            fclose(f);
          }
        }
      }
    };
    runThreads(std::max(1, std::min({hostThreads / nParts, (int)jobs.size(), 64})), worker);
    }
    if (dRec) sipnet_dev_free(dRec);
    if (dPlanes) sipnet_dev_free(dPlanes);
    sipnet_batch_destroy(b);
    partWorst[part] = worstA.load();
    };   // runPart
    if (nParts == 1) {
      runPart(0);
    } else {
      FirstFatal sink;
      std::vector<std::thread> parts;
      for (int p = 0; p < nParts; p++) parts.emplace_back([&, p]() { guarded(sink, [&]() { runPart(p); }); });
      for (auto& th : parts) th.join();
      sink.exitIfSet();
    }
    for (int w : partWorst) worst = std::max(worst, w);
  }
  logInfo(std::to_string(runs.size()) + " run(s) in " + std::to_string(nBatches) + " batch(es)\n");
  for (auto& r : runs) {
    sipnet_clim_free(r.clim);
    sipnet_io_free(r.events);
  }
  return worst;
}

}  // namespace

int main(int argc, char** argv) {
  Context ctx;
  initContext(ctx);

  // ---- command line (cli.c:144-229) ----
  std::vector<option> opts;
  static int tmpFlag = 0;
  for (int k = 0; k < kNumFlagOpts; k++) {
    opts.push_back({kFlagOpts[k][0], no_argument, &tmpFlag, 1});
    opts.push_back({strdup((std::string("no-") + kFlagOpts[k][0]).c_str()), no_argument, &tmpFlag, 0});
  }
  enum { OPT_RIN = 1001, OPT_ROUT, OPT_DBG, OPT_ENS, OPT_DEV, OPT_MATH, OPT_ESTATS, OPT_SITES, OPT_EOUT, OPT_ECOLS, OPT_EF32, OPT_ETEXT, OPT_BOUNDED, OPT_ESEG, OPT_ESUMS };
  opts.push_back({"input-file", required_argument, nullptr, 'i'});
  opts.push_back({"file-prefix", required_argument, nullptr, 'f'});
  opts.push_back({"file-name", required_argument, nullptr, 'f'});
  opts.push_back({"events-prefix", required_argument, nullptr, 'e'});
  opts.push_back({"restart-in", required_argument, nullptr, OPT_RIN});
  opts.push_back({"restart-out", required_argument, nullptr, OPT_ROUT});
  opts.push_back({"debug-log", required_argument, nullptr, OPT_DBG});
  opts.push_back({"ensemble-params", required_argument, nullptr, OPT_ENS});
  opts.push_back({"devices", required_argument, nullptr, OPT_DEV});
  opts.push_back({"math", required_argument, nullptr, OPT_MATH});
  opts.push_back({"ensemble-stats", required_argument, nullptr, OPT_ESTATS});
  opts.push_back({"sites", required_argument, nullptr, OPT_SITES});
  opts.push_back({"ensemble-out", required_argument, nullptr, OPT_EOUT});
  opts.push_back({"ensemble-out-columns", required_argument, nullptr, OPT_ECOLS});
  opts.push_back({"ensemble-out-f32", no_argument, nullptr, OPT_EF32});
  opts.push_back({"ensemble-text", no_argument, nullptr, OPT_ETEXT});
  opts.push_back({"bounded-waits", no_argument, nullptr, OPT_BOUNDED});
  opts.push_back({"ensemble-out-segment", required_argument, nullptr, OPT_ESEG});
  opts.push_back({"ensemble-out-sums", required_argument, nullptr, OPT_ESUMS});
  opts.push_back({"help", no_argument, nullptr, 'h'});
  opts.push_back({"version", no_argument, nullptr, 'v'});
  opts.push_back({nullptr, 0, nullptr, 0});
  std::string ensembleFile, devicesArg = "0", mathArg = "auto", ensembleStats, sitesFile, blockColumns;
  BlockSpec block;
  int longIndex = 0, ch;
  while ((ch = getopt_long(argc, argv, "he:f:i:v", opts.data(), &longIndex)) != -1) {
    switch (ch) {
      case 0: ctx.setInt(kFlagOpts[longIndex / 2][1], tmpFlag, SRC_CLI); break;
      case 'f': ctx.setStr("filePrefix", optarg, SRC_CLI); break;
      case 'e': ctx.setStr("eventsPrefix", optarg, SRC_CLI); break;
      case 'i': ctx.setStr("inputFile", optarg, SRC_CLI); break;
      case OPT_RIN: ctx.setStr("restartIn", optarg, SRC_CLI); break;
      case OPT_ROUT: ctx.setStr("restartOut", optarg, SRC_CLI); break;
      case OPT_DBG: ctx.setStr("debugLogPrefix", optarg, SRC_CLI); break;
      case OPT_ENS: ensembleFile = optarg; break;
      case OPT_DEV: devicesArg = optarg; break;
      case OPT_MATH: mathArg = optarg; break;
      case OPT_ESTATS: ensembleStats = optarg; break;
      case OPT_SITES: sitesFile = optarg; break;
      case OPT_EOUT: block.path = optarg; break;
      case OPT_ECOLS: blockColumns = optarg; break;
      case OPT_EF32: block.f32 = true; break;
      case OPT_ETEXT: block.text = true; break;
      case OPT_BOUNDED: g_kernelOptions |= SIPNET_KOPT_BOUNDED_WAITS; break;
      case OPT_ESEG: block.segment = atoi(optarg); break;
      case OPT_ESUMS: block.sums = atoi(optarg); if (block.sums <= 0) die(8, "--ensemble-out-sums needs a positive number of steps\n"); break;
      case 'h': usage(argv[0]); return 0;
      case 'v': printf("SIPNET version 2.1.0 (%s)\n", sipnet_version()); return 0;
      default: usage(argv[0]); return 8;  // EXIT_CODE_BAD_CLI_ARGUMENT
    }
  }
  std::vector<int> devices = parseDevices(devicesArg);  // syntax errors are CLI errors (exit 8)
  if (mathArg != "auto" && mathArg != "strict" && mathArg != "fast") {
    logError("--math takes strict, fast or auto\n");
    return 8;
  }
  if (!ensembleStats.empty() && ensembleFile.empty()) {
    logError("--ensemble-stats needs --ensemble-params\n");
    return 8;
  }
  if (block.on() ? (ensembleFile.empty() && sitesFile.empty()) || !ensembleStats.empty()
                 : (!blockColumns.empty() || block.f32 || block.text)) {
    logError("--ensemble-out needs --ensemble-params or --sites (not --ensemble-stats); "
             "--ensemble-out-columns / -f32 / --ensemble-text need --ensemble-out\n");
    return 8;
  }
  if (!blockColumns.empty()) parseBlockColumns(block, blockColumns);
  if (block.sums > 0 && (!block.on() || !block.planesOnly() || block.text || !sitesFile.empty())) {
    logError("--ensemble-out-sums sums the three planes of --ensemble-params ... --ensemble-out (no --ensemble-out-columns, "
             "--ensemble-text or --sites)\n");
    return 8;
  }
  g_quiet = ctx.i("quiet") != 0;
  if (!sitesFile.empty()) {
    if (!ensembleFile.empty() || !ensembleStats.empty()) {
      logError("--sites does not combine with --ensemble-params / --ensemble-stats\n");
      return 8;
    }
    return runSites(ctx, sitesFile, mathArg, devices, block);
  }
  if (ctx.s("filePrefix").empty()) die(3, "filePrefix must be set for SIPNET to run\n");
  readInputFile(ctx);
  g_quiet = ctx.i("quiet") != 0;

  // ---- validateContext (context.c:195-223) ----
  bool bad = false;
  if (ctx.i("soilPhenol") && ctx.i("gdd")) { logError("soil-phenol and gdd may not both be turned on\n"); bad = true; }
  if (ctx.i("nitrogenCycle") && !(ctx.i("litterPool") && ctx.i("anaerobic"))) {
    logError("nitrogen-cycle requires both litter-pool and anaerobic to be turned on\n"); bad = true; }
  if (ctx.i("anaerobic") && !ctx.i("waterHResp")) { logError("anaerobic requires water-hresp to be turned on\n"); bad = true; }
  if (ctx.i("carbonSaturation") && !ctx.i("litterPool")) { logError("carbon-saturation requires litter-pool to be turned on\n"); bad = true; }
  if (bad) return 3;
  const std::string debugLog = ctx.s("debugLogPrefix");

  // ---- derived names (frontend.c:164-209) ----
  const std::string prefix = ctx.s("filePrefix");
  ctx.setStr("paramFile", prefix + ".param", SRC_CALCULATED);
  ctx.setStr("climFile", prefix + ".clim", SRC_CALCULATED);
  const bool useEvents = ctx.i("events") != 0;
  const std::string eventsIn = useEvents ? ctx.s("eventsPrefix") + ".in" : "";
  const std::string eventsOut = useEvents ? ctx.s("eventsPrefix") + ".out" : "";
  if (ctx.i("doMainOutput")) ctx.setStr("outFile", prefix + ".out", SRC_CALCULATED);
  if (ctx.i("dumpConfig")) {
    ctx.setStr("outConfigFile", prefix + ".config", SRC_CALCULATED);
    FILE* f = fopen((prefix + ".config").c_str(), "w");
    if (!f) die(6, "Error opening " + prefix + ".config for writing\n");
    printConfig(ctx, f);
    fclose(f);
  }

  int32_t flags[SIPNET_NFLAGS] = {ctx.i("events"), ctx.i("gdd"), ctx.i("growthResp"),
                                  ctx.i("leafWater"), ctx.i("litterPool"), ctx.i("snow"),
                                  ctx.i("soilPhenol"), ctx.i("waterHResp"), ctx.i("nitrogenCycle"),
                                  ctx.i("anaerobic"), ctx.i("flooding"), ctx.i("carbonSaturation")};

  // ---- initModel (sipnet.c:2001-2009) ----
  std::vector<double> base(SIPNET_NPARAMS);
  check(sipnet_io_read_params(ctx.s("paramFile").c_str(), flags, base.data(), nullptr), "reading parameters");
  sipnet_clim_table* clim = nullptr;
  check(sipnet_io_read_clim(ctx.s("climFile").c_str(), flags[SIPNET_F_GDD], &clim), "reading climate");
  const int T = sipnet_clim_nsteps(clim);
  sipnet_event* events = nullptr;
  int32_t nEvents = 0;
  if (useEvents) {
    check(sipnet_io_read_events(eventsIn.c_str(), flags, base.data(), &events, &nEvents), "reading events");
    if (nEvents == 0) logInfo("No event file found, assuming no input events\n");
  }

  // ---- members ----
  std::vector<double> members(base);
  int M = 1;
  if (!ensembleFile.empty()) {
    std::ifstream in(ensembleFile);
    if (!in) die(6, "Error opening " + ensembleFile + " for reading\n");
    std::string line;
    std::getline(in, line);
    std::istringstream hs(line);
    std::vector<int> cols;
    for (std::string name; hs >> name;) {
      const int idx = sipnet_param_index(name.c_str());
      if (idx < 0) die(5, "unknown parameter " + name + " in " + ensembleFile + "\n");
      cols.push_back(idx);
    }
    members.clear();
    M = 0;
    while (std::getline(in, line)) {
      std::istringstream ls(line);
      std::vector<double> row(base);
      size_t k = 0;
      for (double v; ls >> v && k < cols.size(); k++) row[cols[k]] = v;
      if (k == 0) continue;
      if (k != cols.size()) die(5, "short row in " + ensembleFile + "\n");
      members.insert(members.end(), row.begin(), row.end());
      M++;
    }
    if (M == 0) die(5, "no members in " + ensembleFile + "\n");
    logInfo("ensemble of " + std::to_string(M) + " members in one batch\n");
  }

  // ---- restart checkpoint to resume from (restartLoadCheckpoint, restart.c:968-996) ----
  const std::string restartIn = ctx.s("restartIn"), restartOut = ctx.s("restartOut");
  std::vector<sipnet_restart> resume;
  if (!restartIn.empty()) {
    resume.resize(M);
    for (int m = 0; m < M; m++) {
      const std::string path = ensembleFile.empty() ? restartIn : restartIn + "." + std::to_string(m);
      check(sipnet_io_read_restart(path.c_str(), &resume[m]), "reading restart checkpoint");
      checkResume(resume[m], path, flags, clim);
    }
  }

  // ---- run on the GPU(s) ----
  // The ensemble axis shards across the devices given with --devices: contiguous member ranges,
  // one host thread and one sipnet_batch per device, every shard writing its own members'
  // files (members are independent, so no exchange is needed for file output).
  {
    const int have = sipnet_device_count();
    for (int d : devices)
      if (d >= have)
        die(1, "--devices names device " + std::to_string(d) + " but only " + std::to_string(have) +
                   " HIP device(s) are visible (this engine has no CPU path)\n");
  }
  if ((int)devices.size() > M) devices.resize(M);
  if (!ensembleStats.empty()) {
    // ---- the ensemble as ONE node object: shards, RCCL ranks and the all-gather behind the C-ABI ----
    if (!restartIn.empty() || !restartOut.empty() || !debugLog.empty())
      die(8, "--ensemble-stats does not combine with restart checkpoints or --debug-log\n");
    std::vector<int32_t> devs(devices.begin(), devices.end());
    sipnet_node* nd = nullptr;
    check(sipnet_node_create(flags, 1, M, SIPNET_F64, devs.data(), (int32_t)devs.size(), &nd), "creating the node");
    logInfo("ensemble of " + std::to_string(M) + " members on " + std::to_string(devs.size()) +
            " device(s), collectives: " + sipnet_node_collective_library(nd) + "\n");
    check(sipnet_node_set_math(nd, mathArg == "strict" ? SIPNET_MATH_STRICT : SIPNET_MATH_FAST), "math policy");
    if (g_kernelOptions) check(sipnet_node_set_kernel(nd, SIPNET_KERNEL_AUTO, g_kernelOptions), "kernel options");
    check(sipnet_node_set_events(nd, 0, nEvents, events), "events");
    check(sipnet_node_set_climate(nd, 0, T, sipnet_clim_data(clim), sipnet_clim_year(clim), sipnet_clim_day(clim)),
          "climate");
    check(sipnet_node_set_params(nd, 0, 0, M, members.data()), "parameters");
    check(sipnet_node_setup(nd), "setupModel");
    check(sipnet_node_run(nd, 0, T), "run");
    std::vector<double> total((size_t)3 * T * 2);
    check(sipnet_node_gather_stats(nd, total.data()), "all-gather of the statistics");
    int worstStatus = 0, skipped = 0;
    {
      std::vector<int32_t> status(M);   // (synchronises every shard's stream: the runs are complete when it returns)
      check(sipnet_node_get_status(nd, status.data()), "status");
      for (int m = 0; m < M; m++)
        if (status[m] != 0) {
          logError("member " + std::to_string(m) + ": status " + std::to_string(status[m]) +
                   " (NPP allocation params must be less than one individually and add to less than one)\n");
          worstStatus = std::max(worstStatus, (int)status[m]);
          skipped++;
        }
    }
    if (skipped) die(worstStatus, "the statistics would include members that did not run\n");
    FILE* f = fopen(ensembleStats.c_str(), "w");
    if (!f) die(6, "Error opening " + ensembleStats + " for writing\n");
    if (ctx.i("printHeader")) fprintf(f, "year day time n meanNEE sdNEE meanGPP sdGPP meanET sdET\n");
    const double* cd = sipnet_clim_data(clim);
    for (int t = 0; t < T; t++) {
      fprintf(f, "%4d %3d %5.2f %d", sipnet_clim_year(clim)[t], sipnet_clim_day(clim)[t], cd[(size_t)t * SIPNET_NCLIM + 10], M);
      for (int v = 0; v < 3; v++) {
        const double s1 = total[((size_t)v * T + t) * 2], s2 = total[((size_t)v * T + t) * 2 + 1];
        // (s2 - s1^2 / M) / M in extended precision: s2 / M - mean^2 cancels badly when |mean| >> sd
        const long double s1l = s1, s2l = s2;
        const double mean = s1 / M, var = (double)((s2l - s1l * s1l / (long double)M) / (long double)M);
        fprintf(f, " %.10g %.10g", mean, var > 0 ? sqrt(var) : 0.0);
      }
      fprintf(f, "\n");
    }
    fclose(f);
    sipnet_node_destroy(nd);
    sipnet_clim_free(clim);
    sipnet_io_free(events);
    return 0;
  }
  const int nShards = (int)devices.size();
  if (nShards > 1)
    logInfo("ensemble sharded over " + std::to_string(nShards) + " device(s)\n");
  if (block.on() && !debugLog.empty()) die(8, "--ensemble-out does not combine with --debug-log\n");
  // the block is created once; every shard streams its own member range into it
  const bool wantText = !block.on() || block.text;
  const bool needRec = wantText || !restartOut.empty() || (block.on() && !block.planesOnly());
  const bool blockFast = mathArg == "fast" || (mathArg == "auto" && !ensembleFile.empty());
  // --ensemble-out-sums K: the block's rows are the groups of K steps -- each group's first record on the time axis, its step
  // length the group's
  const int sumK = block.on() ? block.sums : 0;
  const int nGroups = sumK > 0 ? (T + sumK - 1) / sumK : 0;
  sipnet_ensemble_file* blockFile = nullptr;
  if (block.on() && sumK > 0) {
    if (!restartOut.empty() || !debugLog.empty()) die(8, "--ensemble-out-sums does not combine with restart output / --debug-log\n");
    std::vector<int32_t> gy(nGroups), gd(nGroups);
    std::vector<double> gc((size_t)nGroups * SIPNET_NCLIM);
    const double* cd = sipnet_clim_data(clim);
    for (int g = 0; g < nGroups; g++) {
      const int t0 = g * sumK, t1 = std::min(T, t0 + sumK);
      gy[g] = sipnet_clim_year(clim)[t0];
      gd[g] = sipnet_clim_day(clim)[t0];
      memcpy(&gc[(size_t)g * SIPNET_NCLIM], cd + (size_t)t0 * SIPNET_NCLIM, SIPNET_NCLIM * sizeof(double));
      double len = 0.0;
      for (int t = t0; t < t1; t++) len += cd[(size_t)t * SIPNET_NCLIM];
      gc[(size_t)g * SIPNET_NCLIM] = len;
    }
    const char* names[3] = {"nee", "gpp", "evapotranspiration"};
    const std::string attrs = "parameter_table=" + ensembleFile + "\nmath=" + (blockFast ? "fast" : "strict") +
                              "\nsums_over_steps=" + std::to_string(sumK);
    check(sipnet_io_ensemble_create(block.path.c_str(), nGroups, M, gy.data(), gd.data(), gc.data(), nullptr, 3, names, nullptr,
                                    block.f32 ? SIPNET_NC_F32 : SIPNET_NC_F64, attrs.c_str(), &blockFile), "creating the ensemble block");
  } else if (block.on()) {
    blockFile = createBlock(block, block.path, T, M, clim, nullptr,
                            "parameter_table=" + ensembleFile + "\nmath=" + (blockFast ? "fast" : "strict"));
  }
  std::atomic<int> worst{0};
  std::mutex logMutex;
  int hostThreads = 1;
  {
    cpu_set_t cpus;
    if (sched_getaffinity(0, sizeof cpus, &cpus) == 0) hostThreads = CPU_COUNT(&cpus);
  }
  auto runShard = [&](int shard) {
    const int m0 = (int)((int64_t)M * shard / nShards), m1 = (int)((int64_t)M * (shard + 1) / nShards);
    const int Ms = m1 - m0;
    const double* shardParams = members.data() + (size_t)m0 * SIPNET_NPARAMS;
    sipnet_batch* b = nullptr;
    check(sipnet_batch_create(flags, 1, Ms, SIPNET_F64, devices[shard], &b), "creating batch");
    // A single run writes the reference's bytes: strict operation order (never the environment's
    // choice).  An ensemble runs on the throughput kernels (their Full instantiations write the
    // same 44-column record; <= 2e-14 from the strict kernel on the fluxes, invisible at the
    // precision `.out` prints) unless --math strict asks otherwise; --debug-log needs strict.
    const bool fastMath = debugLog.empty() && (mathArg == "fast" || (mathArg == "auto" && !ensembleFile.empty()));
    check(sipnet_batch_set_math(b, fastMath ? SIPNET_MATH_FAST : SIPNET_MATH_STRICT), "math policy");
    if (g_kernelOptions) check(sipnet_batch_set_kernel(b, SIPNET_KERNEL_AUTO, g_kernelOptions), "kernel options");
    check(sipnet_batch_set_events(b, 0, nEvents, events), "events");
    check(sipnet_batch_set_climate(b, 0, T, sipnet_clim_data(clim), sipnet_clim_year(clim),
                                   sipnet_clim_day(clim)), "climate");
    check(sipnet_batch_set_params(b, 0, 0, Ms, shardParams), "parameters");
    if (!resume.empty()) check(sipnet_batch_set_resume(b, 0, &resume[m0]), "restart checkpoint");
    check(sipnet_batch_setup(b, nullptr), "setupModel");
    if (!resume.empty())
      check(sipnet_batch_import_restart(b, 0, 0, Ms, resume.data() + m0, nullptr), "restart checkpoint");
    std::vector<double> state0((size_t)Ms * SIPNET_NSTATE);
    check(sipnet_batch_get_state(b, state0.data(), nullptr), "state");
    if (!needRec && sumK > 0) {
      // the block of sums: out of the step kernel's own launch where the batch has such a kernel (1 / K of the bytes ever
      // leave the kernel), else the planes summed on the host in step order
      const bool inKernel = sipnet_batch_sums_in_kernel(b) != 0;
      const size_t rows = inKernel ? (size_t)nGroups : (size_t)T;
      double* dOut = (double*)sipnet_dev_alloc((size_t)3 * rows * Ms * sizeof(double));
      if (!dOut) die(1, std::string(sipnet_last_error()) + "\n");
      if (inKernel)
        check(sipnet_batch_run_sums(b, 0, T, sumK, dOut, dOut + rows * Ms, dOut + 2 * rows * Ms, Ms, nullptr), "run");
      else
        check(sipnet_batch_run(b, 0, T, dOut, dOut + rows * Ms, dOut + 2 * rows * Ms, nullptr, Ms, nullptr), "run");
      std::vector<int32_t> status(Ms);
      check(sipnet_batch_get_status(b, status.data(), nullptr), "status");
      for (int m = 0; m < Ms; m++)
        if (status[m] != 0) {
          std::lock_guard<std::mutex> lock(logMutex);
          logError("member " + std::to_string(m0 + m) + ": status " + std::to_string(status[m]) +
                   " (NPP allocation params must be less than one individually and add to less than one)\n");
          int w = worst.load();
          while (status[m] > w && !worst.compare_exchange_weak(w, status[m])) {}
        }
      std::vector<double> host(rows * Ms), sums;
      for (int v = 0; v < 3; v++) {
        check(sipnet_dev_to_host(host.data(), dOut + (size_t)v * rows * Ms, rows * Ms * sizeof(double), nullptr), "copy back");
        const double* src = host.data();
        if (!inKernel) {
          sums.assign((size_t)nGroups * Ms, 0.0);
          for (int t = 0; t < T; t++) {
            double* acc = &sums[(size_t)(t / sumK) * Ms];
            const double* row = &host[(size_t)t * Ms];
            for (int m = 0; m < Ms; m++) acc[m] += row[m];
          }
          src = sums.data();
        }
        check(sipnet_io_ensemble_put(blockFile, v, 0, nGroups, m0, Ms, src, Ms, 0), "writing the ensemble block");
      }
      if (shard == 0)
        logInfo(std::string("ensemble block: sums over groups of ") + std::to_string(sumK) + " steps, " +
                (inKernel ? "from the step kernel's launch (" : "from the planes, on the host (") + sipnet_batch_last_kernel_name(b) + ")\n");
      sipnet_dev_free(dOut);
      sipnet_batch_destroy(b);
      return;
    }
    if (!needRec) {   // the block alone, three planes: the lean kernels, nothing but the planes leaves the device
      double* dPlanes = (double*)sipnet_dev_alloc((size_t)3 * T * Ms * sizeof(double));
      if (!dPlanes) die(1, std::string(sipnet_last_error()) + "\n");
      check(sipnet_batch_run(b, 0, T, dPlanes, dPlanes + (size_t)T * Ms, dPlanes + (size_t)2 * T * Ms, nullptr, Ms, nullptr), "run");
      std::vector<int32_t> status(Ms);
      check(sipnet_batch_get_status(b, status.data(), nullptr), "status");
      for (int m = 0; m < Ms; m++)
        if (status[m] != 0) {
          std::lock_guard<std::mutex> lock(logMutex);
          logError("member " + std::to_string(m0 + m) + ": status " + std::to_string(status[m]) +
                   " (NPP allocation params must be less than one individually and add to less than one)\n");
          int w = worst.load();
          while (status[m] > w && !worst.compare_exchange_weak(w, status[m])) {}
        }
      putBlock(blockFile, block, T, Ms, T, 0, Ms, m0, dPlanes, nullptr);
      sipnet_dev_free(dPlanes);
      sipnet_batch_destroy(b);
      return;
    }
    if (!wantText && restartOut.empty() && debugLog.empty()) {
      // the block alone, named `.out` columns: the 44-column record exists on the device only, a segment of steps at a
      // time (the whole record of 10 240 members x a half-hourly year is 63 GB: allocating it took 25 s of a 27 s run)
      int64_t segSteps = (int64_t)(3.0e9 / ((double)SIPNET_NREC * Ms * sizeof(double)));
      segSteps = std::max<int64_t>(16, segSteps & ~(int64_t)15);
      if (block.segment > 0) segSteps = block.segment;
      if (segSteps > T) segSteps = T;
      double* dSeg = (double*)sipnet_dev_alloc((size_t)segSteps * SIPNET_NREC * Ms * sizeof(double));
      if (!dSeg) die(1, std::string(sipnet_last_error()) + "\n");
      for (int t0 = 0; t0 < T; t0 += (int)segSteps) {
        const int len = (int)std::min<int64_t>(segSteps, T - t0);
        check(sipnet_batch_run(b, t0, len, nullptr, nullptr, nullptr, dSeg, Ms, nullptr), "run");
        putBlock(blockFile, block, len, Ms, len, 0, Ms, m0, nullptr, dSeg, t0);
      }
      std::vector<int32_t> status(Ms);
      check(sipnet_batch_get_status(b, status.data(), nullptr), "status");
      for (int m = 0; m < Ms; m++)
        if (status[m] != 0) {
          std::lock_guard<std::mutex> lock(logMutex);
          logError("member " + std::to_string(m0 + m) + ": status " + std::to_string(status[m]) +
                   " (NPP allocation params must be less than one individually and add to less than one)\n");
          int w = worst.load();
          while (status[m] > w && !worst.compare_exchange_weak(w, status[m])) {}
        }
      sipnet_dev_free(dSeg);
      sipnet_batch_destroy(b);
      return;
    }
    const size_t recElems = (size_t)T * SIPNET_NREC * Ms;
    double* dRec = (double*)sipnet_dev_alloc(recElems * sizeof(double));
    if (!dRec) die(1, std::string(sipnet_last_error()) + "\n");
    const size_t dbgElems = debugLog.empty() ? 0 : (size_t)T * SIPNET_NDBG * Ms;
    double* dDbg = nullptr;
    std::vector<double> dbg(dbgElems);
    if (dbgElems) {  // --debug-log: the per-step flux / tracker dump of debug_log.c
      dDbg = (double*)sipnet_dev_alloc(dbgElems * sizeof(double));
      if (!dDbg) die(1, std::string(sipnet_last_error()) + "\n");
      check(sipnet_batch_run_debug(b, 0, T, dRec, dDbg, Ms, nullptr), "run");
    } else {
      const double t0 = nowSeconds();
      check(sipnet_batch_run(b, 0, T, nullptr, nullptr, nullptr, dRec, Ms, nullptr), "run");
      check(sipnet_stream_sync(nullptr), "run");
      char msg[160];
      snprintf(msg, sizeof msg, "members %d..%d: %d steps with the 44-column record in %.3f s\n", m0, m1 - 1, T, nowSeconds() - t0);
      logInfo(msg);
    }
    if (blockFile && !block.planesOnly()) putBlock(blockFile, block, T, Ms, T, 0, Ms, m0, nullptr, dRec);
    if (blockFile && block.planesOnly()) {   // (with text or checkpoints: the three planes are record columns 0, 1, 2)
      BlockSpec asCols = block;
      asCols.cols = {sipnet_io_out_column_index("nee"), sipnet_io_out_column_index("gpp"), sipnet_io_out_column_index("evapotranspiration")};
      putBlock(blockFile, asCols, T, Ms, T, 0, Ms, m0, nullptr, dRec);
    }
    // (what follows needs every record on the host: the members' text files and checkpoints)
    std::vector<double> rec(recElems);
    check(sipnet_dev_to_host(rec.data(), dRec, recElems * sizeof(double), nullptr), "copy back");
    if (dbgElems) check(sipnet_dev_to_host(dbg.data(), dDbg, dbgElems * sizeof(double), nullptr), "copy back");
    std::vector<int32_t> status(Ms);
    check(sipnet_batch_get_status(b, status.data(), nullptr), "status");

    // ---- outputs ----
    // The text of an ensemble is the slow part of the job (about 120 MB/s of printf per host
    // thread against tens of ms of GPU time), so members are formatted by a pool of host threads;
    // only the checkpoint export, which talks to the GPU through the one batch handle, is
    // serialised.
    for (int m = 0; m < Ms; m++) {
      if (status[m] != 0) {
        std::lock_guard<std::mutex> lock(logMutex);
        logError("member " + std::to_string(m0 + m) + ": status " + std::to_string(status[m]) +
                 " (NPP allocation params must be less than one individually and add to less than one)\n");
        int w = worst.load();
        while (status[m] > w && !worst.compare_exchange_weak(w, status[m])) {}
      }
    }
    std::mutex gpuMutex;
    auto writeMember = [&](int m, std::vector<double>& one, std::vector<double>& oneDbg) {
      if (status[m] != 0) return;
      for (int t = 0; t < T; t++)
        for (int k = 0; k < SIPNET_NREC; k++)
          one[(size_t)t * SIPNET_NREC + k] = rec[((size_t)t * SIPNET_NREC + k) * Ms + m];
      const std::string tag = ensembleFile.empty() ? "" : "." + std::to_string(m0 + m);
      if (dbgElems) {
        for (int t = 0; t < T; t++)
          for (int k = 0; k < SIPNET_NDBG; k++)
            oneDbg[(size_t)t * SIPNET_NDBG + k] = dbg[((size_t)t * SIPNET_NDBG + k) * Ms + m];
        check(sipnet_io_write_debug_logs((debugLog + tag).c_str(), ctx.i("printHeader"), T, sipnet_clim_year(clim),
                                         sipnet_clim_day(clim), sipnet_clim_data(clim), one.data(),
                                         oneDbg.data()), "writing debug logs");
      }
      if (wantText && ctx.i("doMainOutput"))
        check(sipnet_io_write_out((prefix + tag + ".out").c_str(), ctx.i("printHeader"), T,
                                  sipnet_clim_year(clim), sipnet_clim_day(clim), sipnet_clim_data(clim),
                                  one.data()), "writing output");
      if (wantText && useEvents)
        check(sipnet_io_write_events_out((ctx.s("eventsPrefix") + tag + ".out").c_str(),
                                         ctx.i("printHeader"), flags, shardParams + (size_t)m * SIPNET_NPARAMS,
                                         T, sipnet_clim_year(clim), sipnet_clim_day(clim),
                                         sipnet_clim_data(clim), nEvents, events, one.data(),
                                         state0.data() + (size_t)m * SIPNET_NSTATE), "writing events.out");
      if (!restartOut.empty()) {  // restartWriteCheckpoint, restart.c:932-996
        const double* prevPools = T >= 2 ? one.data() + (size_t)(T - 2) * SIPNET_NREC + 14
                                         : state0.data() + (size_t)m * SIPNET_NSTATE;
        writeCheckpoint(b, gpuMutex, logMutex, 0, m, T, one.data() + (size_t)(T - 1) * SIPNET_NREC, prevPools, restartOut + tag);
      }
      if (wantText && ctx.i("doSingleOutputs")) {  // sipnet.c:1993-1998, outputItems.c:126-150
        const struct { const char* name; int col; } items[] = {{"NEE", 0}, {"NEE_cum", 3}, {"GPP", 1}, {"GPP_cum", 35}};
        for (const auto& it : items) {
          FILE* f = fopen((prefix + tag + "." + it.name).c_str(), "w");
          if (!f) die(6, std::string("Error opening single output file for ") + it.name + "\n");
          for (int t = 0; t < T; t++) fprintf(f, "%f ", one[(size_t)t * SIPNET_NREC + it.col]);
          fprintf(f, "\n");
          fclose(f);
        }
      }
    };
    {
      std::atomic<int> next{0};
      auto worker = [&]() {
        std::vector<double> one((size_t)T * SIPNET_NREC), oneDbg(dbgElems ? (size_t)T * SIPNET_NDBG : 0);
        for (int m = next.fetch_add(1); m < Ms; m = next.fetch_add(1)) writeMember(m, one, oneDbg);
      };
      runThreads(std::max(1, std::min({hostThreads / nShards, Ms, 64})), worker);
    }
    sipnet_dev_free(dRec);
    if (dDbg) sipnet_dev_free(dDbg);
    sipnet_batch_destroy(b);
  };
  if (nShards == 1) {
    runShard(0);
  } else {
    FirstFatal sink;
    std::vector<std::thread> shards;
    for (int k = 0; k < nShards; k++) shards.emplace_back([&, k]() { guarded(sink, [&]() { runShard(k); }); });
    for (auto& th : shards) th.join();
    sink.exitIfSet();
  }
  if (blockFile) check(sipnet_io_ensemble_close(blockFile), "closing the ensemble block");
  sipnet_clim_free(clim);
  sipnet_io_free(events);
  return worst.load();
}
