// Mutation fuzzer of everything on the host side that parses untrusted files or lays out bytes -- host_io.cpp
// (.clim, .param, events.in), restart_io.cpp (SIPNET_RESTART checkpoints), plan.cpp (the site plan built from a parsed
// forcing), ensemble_io.cpp (the NetCDF block) -- compiled WITHOUT HIP by g++ under AddressSanitizer +
// UndefinedBehaviourSanitizer (`make -C sipnet_amd/csrc san`).  Reference parsers: modelParams.c:136-230,
// sipnet.c:128-277, events.c:263-367, restart.c:590-756.
//
//   fuzz_host_io ITERATIONS SCRATCH_DIR SEED_FILE...       seed files by extension: .clim .param .in .restart
//
// Every seed is parsed as it is (must succeed), then ITERATIONS mutated copies (deterministic: the PRNG is seeded by
// the seed file's index and the iteration) are parsed; whatever the parsers return is fine, what the sanitizers
// report is not (they abort the process).  A forcing that still parses also goes through buildSitePlan (both record
// types, narrow and wide) -- with the mutated events when an events file has been seen -- and through buildSitePlanLight,
// the host's share of a device-built plan, whose every output must equal the full builder's (abort otherwise).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/sipnet_amd.h"
#include "../../sipnet_amd/csrc/plan.h"

namespace sipnet {
static thread_local std::string g_err;
void setError(const std::string& s) { g_err = s; }
}  // namespace sipnet
extern "C" const char* sipnet_last_error(void) { return sipnet::g_err.c_str(); }
extern "C" const char* sipnet_version(void) { return "sipnet_amd sanitizer build"; }

static std::string slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
static void spit(const std::string& p, const std::string& s) {
  std::ofstream f(p, std::ios::binary | std::ios::trunc);
  f.write(s.data(), (std::streamsize)s.size());
}
static bool endsWith(const std::string& s, const char* e) {
  const size_t n = strlen(e);
  return s.size() >= n && s.compare(s.size() - n, n, e) == 0;
}

static std::string mutate(std::string s, std::mt19937& rng) {
  static const char* const dict[] = {"nan", "inf", "-inf", "-1", "0", "1e309", "-1e-320", "9999999999999999999999", "", "\n", "!",
                                     "\t", " ", "=", ":", "0x10", "1.", ".", "-", "+", "e", "#", "\r\n", "\0x", "mean.npp.values",
                                     "end_restart", "SIPNET_RESTART", "plantWoodInit", "irrig", "harv", "till", "plant", "fert", "leafon"};
  const int nOps = 1 + (int)(rng() % 4);
  for (int k = 0; k < nOps; k++) {
    const size_t n = s.size();
    const size_t at = n ? rng() % n : 0;
    switch (rng() % 7) {
      case 0:   // flip a byte
        if (n) s[at] = (char)(rng() & 0xff);
        break;
      case 1: {   // delete a range
        if (n) s.erase(at, 1 + rng() % (1 + (n - at < 40 ? n - at : 40)));
        break;
      }
      case 2: {   // duplicate a range
        if (n) {
          const size_t len = 1 + rng() % (n - at < 200 ? n - at : 200);
          s.insert(at, s.substr(at, len));
        }
        break;
      }
      case 3:   // insert a token
        s.insert(at, dict[rng() % (sizeof dict / sizeof dict[0])]);
        break;
      case 4:   // truncate
        s.resize(at);
        break;
      case 5: {   // replace the token at `at`
        size_t a = at, z = at;
        while (a > 0 && !isspace((unsigned char)s[a - 1])) a--;
        while (z < n && !isspace((unsigned char)s[z])) z++;
        s.replace(a, z - a, dict[rng() % (sizeof dict / sizeof dict[0])]);
        break;
      }
      default: {   // swap two lines
        const size_t l0 = s.rfind('\n', at), l1 = s.find('\n', at);
        if (l0 != std::string::npos && l1 != std::string::npos && l1 + 1 < n) {
          const std::string line = s.substr(l0 + 1, l1 - l0);
          s.erase(l0 + 1, l1 - l0);
          s.insert(rng() % (s.size() + 1), line);
        }
      }
    }
  }
  return s;
}

static int32_t g_flags[SIPNET_NFLAGS] = {1, 1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0};
static std::vector<double> g_params(SIPNET_NPARAMS, 1.0);
static std::vector<sipnet_event> g_events;
static long g_ok = 0, g_rejected = 0;

static sipnet::PlanCarry g_carry;
static long g_lightChecked = 0;
static void planFrom(const sipnet_clim_table* t) {
  const int32_t n = sipnet_clim_nsteps(t);
  if (n <= 0 || n > 20000) return;
  std::vector<sipnet::StepRec> steps((size_t)n);
  std::vector<sipnet::FastRec> fast((size_t)n);
  for (int narrow = 0; narrow < 2; narrow++) {
    sipnet::PlanCarry fin;
    sipnet::SitePlan p = sipnet::buildSitePlan(g_flags, n, sipnet_clim_data(t), sipnet_clim_year(t), sipnet_clim_day(t),
                                               (int32_t)g_events.size(), g_events.empty() ? nullptr : g_events.data(), nullptr, &fin,
                                               narrow == 0, narrow == 0 ? steps.data() : nullptr, fast.data(), narrow != 0);
    // the light pass a device-built plan gets from the host (plan_device.h) must say what the full builder says: the GDD
    // chain and the tillage series bit for bit, the same events on the same records, the same verdict on the site -- for a
    // fresh start and, once a checkpoint has been parsed, for a segment resumed from it
    if (narrow == 0) {
      for (int resumed = 0; resumed < (g_carry.set ? 2 : 1); resumed++) {
        const sipnet::PlanCarry* init = resumed ? &g_carry : nullptr;
        sipnet::SitePlan pf = p;
        if (resumed)
          pf = sipnet::buildSitePlan(g_flags, n, sipnet_clim_data(t), sipnet_clim_year(t), sipnet_clim_day(t), (int32_t)g_events.size(),
                                     g_events.empty() ? nullptr : g_events.data(), init, nullptr, true, steps.data(), nullptr, false);
        std::vector<double> gdd((size_t)n), dTill((size_t)n), tillAfter((size_t)n);
        std::vector<int32_t> evFirst((size_t)n), evCount((size_t)n);
        sipnet::PlanLight l = sipnet::buildSitePlanLight(g_flags, n, sipnet_clim_data(t), sipnet_clim_year(t), sipnet_clim_day(t),
                                                         (int32_t)g_events.size(), g_events.empty() ? nullptr : g_events.data(), init,
                                                         0.0202, 48, gdd.data(), evFirst.data(), evCount.data(), dTill.data(), tillAfter.data());
        bool same = (l.status == SIPNET_OK) == (pf.status == SIPNET_OK);
        if (same && pf.status == SIPNET_OK) {
          same = memcmp(gdd.data(), pf.gddAfter.data(), (size_t)n * sizeof(double)) == 0 && l.events.size() == pf.events.size() &&
                 (l.events.empty() || memcmp(l.events.data(), pf.events.data(), l.events.size() * sizeof(sipnet::EvRec)) == 0) &&
                 memcmp(&l.startCumGdd, &pf.startCumGdd, sizeof(double)) == 0 && memcmp(&l.startDayTime, &pf.startDayTime, sizeof(double)) == 0;
          for (int k = 0; same && k < n; k++) {
            const sipnet::StepRec& sr = steps[(size_t)k];
            if (l.hasEvents)
              same = evFirst[(size_t)k] == sr.evFirst && evCount[(size_t)k] == sr.evCount &&
                     memcmp(&dTill[(size_t)k], &sr.dTill, sizeof(double)) == 0 && memcmp(&tillAfter[(size_t)k], &sr.tillAfter, sizeof(double)) == 0;
            else
              same = sr.evCount == 0 && sr.dTill == 0.0 && sr.tillAfter == 0.0;
          }
        }
        if (!same) {
          fprintf(stderr, "buildSitePlanLight disagrees with buildSitePlan (status %d vs %d, resumed %d)\n", l.status, pf.status, resumed);
          abort();
        }
        // the room a DEVICE-built site gets for its eviction list (engine.hip devRingOpRoom: 2 n + the live entries the ring starts
        // with + 8) bounds the list the host builder produces -- for every ring the device is handed: one that carries the 5-day
        // window (engine.hip devicePrepass)
        if (pf.status == SIPNET_OK) {
          size_t preK = 1;
          double sum = 5.0;
          if (init) {
            const sipnet::RingSched& r = init->ring;
            preK = (size_t)((r.last - r.start + SIPNET_RING_SLOTS) % SIPNET_RING_SLOTS) + 1;
            sum = 0.0;
            for (int i = r.start;; i = (i + 1) % SIPNET_RING_SLOTS) {
              sum += r.w[i];
              if (i == r.last) break;
            }
          }
          if (sum - 5.0 <= 1e-9 && 5.0 - sum <= 1e-9 && pf.ringOps.size() > 2 * (size_t)n + preK + 8) {
            fprintf(stderr, "eviction list of %zu entries for %d steps and %zu initial ring entries: over the device's room\n",
                    pf.ringOps.size(), (int)n, preK);
            abort();
          }
        }
        g_lightChecked++;
      }
    }
  }
}

static void parse(const std::string& kind, const std::string& path) {
  int rc = 0;
  if (kind == "clim") {
    sipnet_clim_table* t = nullptr;
    rc = sipnet_io_read_clim(path.c_str(), g_flags[SIPNET_F_GDD], &t);
    if (rc == 0 && t) planFrom(t);
    if (t) sipnet_clim_free(t);
  } else if (kind == "param") {
    std::vector<double> out(SIPNET_NPARAMS);
    std::vector<int32_t> seen(SIPNET_NPARAMS);
    rc = sipnet_io_read_params(path.c_str(), g_flags, out.data(), seen.data());
    if (rc == 0) g_params = out;
  } else if (kind == "events") {
    sipnet_event* ev = nullptr;
    int32_t n = 0;
    rc = sipnet_io_read_events(path.c_str(), g_flags, g_params.data(), &ev, &n);
    if (rc == 0 && n >= 0 && n < 4096) g_events.assign(ev, ev + n);
    sipnet_io_free(ev);
  } else {
    sipnet_restart r;
    rc = sipnet_io_read_restart(path.c_str(), &r);
    if (rc == 0) {   // what a resume does with it next, and the way back to text
      int32_t warn = 0;
      (void)sipnet_restart_check(&r, g_flags, 1, r.boundary_year, r.boundary_day + 1, 0.0, 0.5, &warn);
      (void)sipnet_restart_check_boundary_for_write(&r, &warn);
      (void)sipnet_io_write_restart((path + ".rewritten").c_str(), &r);
      // what sipnet_batch_set_resume keeps of it for the site plan (engine.hip): the plans of the forcings that follow are
      // also built as resumed segments
      if (r.mean_length == SIPNET_RING_SLOTS && r.mean_start >= 0 && r.mean_start < SIPNET_RING_SLOTS && r.mean_last >= 0 &&
          r.mean_last < SIPNET_RING_SLOTS) {
        g_carry = sipnet::PlanCarry{};
        g_carry.set = true;
        g_carry.gdd = r.trackers[SIPNET_RT_GDD];
        g_carry.trackLastYear = r.trackers_last_year;
        g_carry.phenLastYear = r.phenology_last_year;
        g_carry.dTill = r.d_till_mod;
        g_carry.ring.start = r.mean_start;
        g_carry.ring.last = r.mean_last;
        for (int i = 0; i < SIPNET_RING_SLOTS; i++) {
          g_carry.ring.w[i] = r.mean_weights[i];
          g_carry.ring.insStep[i] = 0;
        }
      }
    }
  }
  (rc == 0 ? g_ok : g_rejected)++;
}

static void fuzzEnsembleBlock(const std::string& scratch, std::mt19937& rng) {
  const int T = 1 + (int)(rng() % 40), M = 1 + (int)(rng() % 9);
  std::vector<int32_t> year(T, 2000), day(T, 1);
  std::vector<double> clim((size_t)T * SIPNET_NCLIM, 0.5), rec((size_t)T * SIPNET_NREC * (M + 3), 1.25);
  const char* cols[] = {nullptr, "", "nee", "nee,gpp,plantWoodC", "all?", "soilWater,,snow", ",", "nee,nee"};
  (void)sipnet_io_write_ensemble_block((scratch + "/blk.nc").c_str(), T, M, year.data(), day.data(), clim.data(), nullptr, nullptr,
                                       rec.data(), M + (int)(rng() % 4), cols[rng() % 8], (int)(rng() % 4), (rng() & 1) ? "k=v\nbad\n=x\nq=" : nullptr);
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s ITERATIONS SCRATCH_DIR SEED_FILE...\n", argv[0]);
    return 2;
  }
  const int iters = atoi(argv[1]);
  const std::string scratch = argv[2];
  for (int a = 3; a < argc; a++) {
    const std::string seedPath = argv[a];
    const std::string kind = endsWith(seedPath, ".clim") ? "clim" : endsWith(seedPath, ".param") ? "param"
                             : endsWith(seedPath, ".restart") ? "restart" : "events";
    const std::string seed = slurp(seedPath);
    const std::string tmp = scratch + "/mut." + kind;
    const long ok0 = g_ok;
    parse(kind, seedPath);
    if (g_ok != ok0 + 1) {
      fprintf(stderr, "the unmutated seed %s was rejected: %s\n", seedPath.c_str(), sipnet_last_error());
      return 1;
    }
    // a forcing of thousands of records costs a plan build per mutation: cut long seeds down to their head
    std::string base = seed;
    if (kind == "clim") {
      size_t pos = 0;
      for (int lines = 0; lines < 400 && pos != std::string::npos; lines++) pos = base.find('\n', pos + 1);
      if (pos != std::string::npos) base.resize(pos + 1);
    }
    for (int it = 0; it < iters; it++) {
      std::mt19937 rng((uint32_t)(a * 1000003 + it));
      spit(tmp, mutate(base, rng));
      parse(kind, tmp);
      if (it % 16 == 0) fuzzEnsembleBlock(scratch, rng);
    }
  }
  printf("fuzz_host_io: %ld inputs parsed, %ld rejected, no sanitizer finding\n", g_ok, g_rejected);
  printf("light plan pass held against the full builder %ld times (resumed segments included: %d)\n", g_lightChecked, g_carry.set ? 1 : 0);
  return 0;
}
