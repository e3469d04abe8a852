// restart_io.cpp -- the `SIPNET_RESTART` checkpoint text at the drop-in boundary (host only).
//
// Format and checks of the reference's sipnet/restart.c (citations relative to
// /root/reference/src/), built around one table of (key, kind, offset into sipnet_restart)
// rows and status returns instead of exit().
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sipnet_amd.h"

namespace sipnet {
void setError(const std::string& s);

namespace {
constexpr const char* kMagic = "SIPNET_RESTART";   // restart.c:22
constexpr const char* kModelVersion = "2.1.0";     // sipnet/version.h:4
constexpr double kEps = 1e-8;                      // restart.c:23
// compiled struct sizes the reference pins the schema with (restart.c:30-38)
constexpr int kSchemaEnvi = 104, kSchemaTrackers = 264, kSchemaPhenology = 12,
              kSchemaSurvival = 4, kSchemaEvent = 24;

enum Kind { K_STR, K_I64, K_INT, K_DBL, K_SCHEMA };
struct Key {
  const char* name;
  Kind kind;
  size_t off;   // offset into sipnet_restart (K_SCHEMA: unused)
  int aux;      // K_STR: buffer size; K_SCHEMA: expected value
  int group;    // blank line after the last key of a group when writing
};
#define OFF(f) offsetof(sipnet_restart, f)
#define TRK(i) (offsetof(sipnet_restart, trackers) + sizeof(double) * (i))
#define ENV(i) (offsetof(sipnet_restart, envi) + sizeof(double) * (i))
#define FLG(i) (offsetof(sipnet_restart, flags) + sizeof(int32_t) * (i))
// File order of writeRestartState (restart.c:787-828); groups: 0 meta+schema, 1 flags,
// 2 boundary, 3 envi, 4 trackers, 5 phenology, 6 survival, 7 event trackers, 8 mean meta.
const Key kKeys[] = {
    {"meta_info.model_version", K_STR, OFF(model_version), 32, 0},
    {"meta_info.build_info", K_STR, OFF(build_info), 96, 0},
    {"meta_info.checkpoint_utc_epoch", K_I64, OFF(checkpoint_utc_epoch), 0, 0},
    {"meta_info.processed_steps", K_I64, OFF(processed_steps), 0, 0},
    {"schema_layout.envi_size", K_SCHEMA, 0, kSchemaEnvi, 0},
    {"schema_layout.trackers_size", K_SCHEMA, 0, kSchemaTrackers, 0},
    {"schema_layout.phenology_trackers_size", K_SCHEMA, 0, kSchemaPhenology, 0},
    {"schema_layout.survival_trackers_size", K_SCHEMA, 0, kSchemaSurvival, 0},
    {"schema_layout.event_trackers_size", K_SCHEMA, 0, kSchemaEvent, 0},
    {"flags.events", K_INT, FLG(SIPNET_F_EVENTS), 0, 1},
    {"flags.gdd", K_INT, FLG(SIPNET_F_GDD), 0, 1},
    {"flags.growthResp", K_INT, FLG(SIPNET_F_GROWTH_RESP), 0, 1},
    {"flags.leafWater", K_INT, FLG(SIPNET_F_LEAF_WATER), 0, 1},
    {"flags.litterPool", K_INT, FLG(SIPNET_F_LITTER_POOL), 0, 1},
    {"flags.snow", K_INT, FLG(SIPNET_F_SNOW), 0, 1},
    {"flags.soilPhenol", K_INT, FLG(SIPNET_F_SOIL_PHENOL), 0, 1},
    {"flags.waterHResp", K_INT, FLG(SIPNET_F_WATER_HRESP), 0, 1},
    {"flags.nitrogenCycle", K_INT, FLG(SIPNET_F_NITROGEN_CYCLE), 0, 1},
    {"flags.anaerobic", K_INT, FLG(SIPNET_F_ANAEROBIC), 0, 1},
    {"flags.flooding", K_INT, FLG(SIPNET_F_FLOODING), 0, 1},
    {"flags.carbonSaturation", K_INT, FLG(SIPNET_F_CARBON_SATURATION), 0, 1},
    {"boundary.year", K_INT, OFF(boundary_year), 0, 2},
    {"boundary.day", K_INT, OFF(boundary_day), 0, 2},
    {"boundary.time", K_DBL, OFF(boundary_time), 0, 2},
    {"boundary.length", K_DBL, OFF(boundary_length), 0, 2},
    {"envi.plantWoodC", K_DBL, ENV(0), 0, 3},
    {"envi.plantLeafC", K_DBL, ENV(1), 0, 3},
    {"envi.soilC", K_DBL, ENV(2), 0, 3},
    {"envi.soilWater", K_DBL, ENV(3), 0, 3},
    {"envi.litterC", K_DBL, ENV(4), 0, 3},
    {"envi.snow", K_DBL, ENV(5), 0, 3},
    {"envi.coarseRootC", K_DBL, ENV(6), 0, 3},
    {"envi.fineRootC", K_DBL, ENV(7), 0, 3},
    {"envi.minN", K_DBL, ENV(8), 0, 3},
    {"envi.soilOrgN", K_DBL, ENV(9), 0, 3},
    {"envi.litterN", K_DBL, ENV(10), 0, 3},
    {"envi.plantStorageN", K_DBL, ENV(11), 0, 3},
    {"envi.plantCAccountingDelta", K_DBL, ENV(12), 0, 3},
    {"trackers.gpp", K_DBL, TRK(SIPNET_RT_GPP), 0, 4},
    {"trackers.rtot", K_DBL, TRK(SIPNET_RT_RTOT), 0, 4},
    {"trackers.ra", K_DBL, TRK(SIPNET_RT_RA), 0, 4},
    {"trackers.rh", K_DBL, TRK(SIPNET_RT_RH), 0, 4},
    {"trackers.rRoot", K_DBL, TRK(SIPNET_RT_RROOT), 0, 4},
    {"trackers.rSoil", K_DBL, TRK(SIPNET_RT_RSOIL), 0, 4},
    {"trackers.rAboveground", K_DBL, TRK(SIPNET_RT_RABOVEGROUND), 0, 4},
    {"trackers.npp", K_DBL, TRK(SIPNET_RT_NPP), 0, 4},
    {"trackers.nee", K_DBL, TRK(SIPNET_RT_NEE), 0, 4},
    {"trackers.woodCreation", K_DBL, TRK(SIPNET_RT_WOODCREATION), 0, 4},
    {"trackers.gdd", K_DBL, TRK(SIPNET_RT_GDD), 0, 4},
    {"trackers.evapotranspiration", K_DBL, TRK(SIPNET_RT_ET), 0, 4},
    {"trackers.soilWetnessFrac", K_DBL, TRK(SIPNET_RT_SOILWETNESSFRAC), 0, 4},
    {"trackers.yearlyGpp", K_DBL, TRK(SIPNET_RT_YEARLYGPP), 0, 4},
    {"trackers.yearlyRtot", K_DBL, TRK(SIPNET_RT_YEARLYRTOT), 0, 4},
    {"trackers.yearlyRa", K_DBL, TRK(SIPNET_RT_YEARLYRA), 0, 4},
    {"trackers.yearlyRh", K_DBL, TRK(SIPNET_RT_YEARLYRH), 0, 4},
    {"trackers.yearlyNpp", K_DBL, TRK(SIPNET_RT_YEARLYNPP), 0, 4},
    {"trackers.yearlyNee", K_DBL, TRK(SIPNET_RT_YEARLYNEE), 0, 4},
    {"trackers.yearlyLitter", K_DBL, TRK(SIPNET_RT_YEARLYLITTER), 0, 4},
    {"trackers.totGpp", K_DBL, TRK(SIPNET_RT_TOTGPP), 0, 4},
    {"trackers.totRtot", K_DBL, TRK(SIPNET_RT_TOTRTOT), 0, 4},
    {"trackers.totRa", K_DBL, TRK(SIPNET_RT_TOTRA), 0, 4},
    {"trackers.totRh", K_DBL, TRK(SIPNET_RT_TOTRH), 0, 4},
    {"trackers.totNpp", K_DBL, TRK(SIPNET_RT_TOTNPP), 0, 4},
    {"trackers.totNee", K_DBL, TRK(SIPNET_RT_TOTNEE), 0, 4},
    {"trackers.lastYear", K_INT, OFF(trackers_last_year), 0, 4},
    {"trackers.methane", K_DBL, TRK(SIPNET_RT_METHANE), 0, 4},
    {"trackers.n2o", K_DBL, TRK(SIPNET_RT_N2O), 0, 4},
    {"trackers.nLeaching", K_DBL, TRK(SIPNET_RT_NLEACHING), 0, 4},
    {"trackers.nFixation", K_DBL, TRK(SIPNET_RT_NFIXATION), 0, 4},
    {"trackers.nUptake", K_DBL, TRK(SIPNET_RT_NUPTAKE), 0, 4},
    {"trackers.meanNPP", K_DBL, TRK(SIPNET_RT_MEANNPP), 0, 4},
    {"phenology.didLeafGrowth", K_INT, OFF(did_leaf_growth), 0, 5},
    {"phenology.didLeafFall", K_INT, OFF(did_leaf_fall), 0, 5},
    {"phenology.lastYear", K_INT, OFF(phenology_last_year), 0, 5},
    {"survival.isAlive", K_INT, OFF(is_alive), 0, 6},
    {"event_trackers.d_till_mod", K_DBL, OFF(d_till_mod), 0, 7},
    {"event_trackers.harvestFracRemoved", K_DBL, OFF(harvest_frac_removed), 0, 7},
    {"event_trackers.harvestFracTransferred", K_DBL, OFF(harvest_frac_transferred), 0, 7},
    {"mean.npp.length", K_INT, OFF(mean_length), 0, 8},
    {"mean.npp.totWeight", K_DBL, OFF(mean_tot_weight), 0, 8},
    {"mean.npp.start", K_INT, OFF(mean_start), 0, 8},
    {"mean.npp.last", K_INT, OFF(mean_last), 0, 8},
    {"mean.npp.sum", K_DBL, OFF(mean_sum), 0, 8},
};
constexpr int kNumKeys = (int)(sizeof(kKeys) / sizeof(kKeys[0]));
#undef OFF
#undef TRK
#undef ENV
#undef FLG

int fail(const std::string& path, const std::string& msg) {
  setError("Restart parse error in " + path + ": " + msg);
  return SIPNET_ERR_RESTART;
}
std::string badValue(const char* key, const char* value) {
  return std::string("invalid value '") + value + "' for key '" + key + "'";
}

// restart.c:418-452: whole-token numbers only, no overflow, finite
bool parseI64(const char* v, long long* out) {
  char* end = nullptr;
  errno = 0;
  const long long x = strtoll(v, &end, 10);
  if (end == v || *end != '\0' || errno == ERANGE) return false;
  *out = x;
  return true;
}
bool parseInt(const char* v, int32_t* out) {
  long long x;
  if (!parseI64(v, &x) || x < INT_MIN || x > INT_MAX) return false;
  *out = (int32_t)x;
  return true;
}
bool parseDbl(const char* v, double* out) {
  char* end = nullptr;
  const double x = strtod(v, &end);
  if (end == v || *end != '\0' || !std::isfinite(x)) return false;
  *out = x;
  return true;
}
bool startsWith(const char* s, const char* prefix) {
  return strncmp(s, prefix, strlen(prefix)) == 0;
}

bool isLeap(int y) { return ((y % 4 == 0) && (y % 100 != 0)) || (y % 400 == 0); }
}  // namespace
}  // namespace sipnet

using namespace sipnet;

extern "C" {

int sipnet_io_read_restart(const char* path, sipnet_restart* out) {
  if (!path || !out) return SIPNET_ERR_BAD_ARGUMENT;
  FILE* in = fopen(path, "r");
  if (!in) {
    setError(std::string("Error opening ") + path + " for reading");
    return SIPNET_ERR_FILE_OPEN;
  }
  struct Closer {
    FILE* f;
    ~Closer() { fclose(f); }
  } closer{in};
  const std::string p(path);
  memset(out, 0, sizeof(*out));
  out->mean_length = SIPNET_RING_SLOTS;  // the compiled ring length (sipnet.c:2008)

  // magic line, restart.c:596-607
  char first[256];
  if (!fgets(first, sizeof first, in)) return fail(p, "missing header line");
  {
    const size_t n = strlen(first);
    if (n > 0 && first[n - 1] != '\n' && !feof(in)) return fail(p, "line too long or truncated");
  }
  first[strcspn(first, "\r\n")] = '\0';
  if (strcmp(first, kMagic) != 0) {
    setError("Restart file " + p + " has invalid magic header");
    return SIPNET_ERR_RESTART;
  }

  std::vector<char> seen(kNumKeys, 0);
  std::vector<char> seenVal(SIPNET_RING_SLOTS, 0), seenWgt(SIPNET_RING_SLOTS, 0);
  bool seenEnd = false;
  char line[4096], key[128], value[2048], extra[32];
  while (fgets(line, sizeof line, in)) {
    const size_t n = strlen(line);
    if (n > 0 && line[n - 1] != '\n' && !feof(in)) return fail(p, "line too long or truncated");
    if (seenEnd) break;  // lines after end_restart are ignored, restart.c:626-630
    const int got = sscanf(line, " %127s %2047s %31s", key, value, extra);
    if (got <= 0) continue;  // blank line
    if (got != 2) return fail(p, "line must contain exactly '<key> <value>'");

    int hit = -1;
    for (int k = 0; k < kNumKeys; k++) {
      if (strcmp(kKeys[k].name, key) == 0) {
        hit = k;
        break;
      }
    }
    if (hit >= 0) {
      const Key& K = kKeys[hit];
      if (seen[hit]) return fail(p, std::string("duplicate key '") + key + "'");
      char* base = reinterpret_cast<char*>(out) + K.off;
      switch (K.kind) {
        case K_STR: {
          memset(base, 0, (size_t)K.aux);
          strncpy(base, value, (size_t)K.aux - 1);
        } break;
        case K_I64: {
          long long x;
          if (!parseI64(value, &x)) return fail(p, badValue(key, value));
          *reinterpret_cast<int64_t*>(base) = (int64_t)x;
        } break;
        case K_INT: {
          int32_t x;
          if (!parseInt(value, &x)) return fail(p, badValue(key, value));
          *reinterpret_cast<int32_t*>(base) = x;
        } break;
        case K_DBL: {
          double x;
          if (!parseDbl(value, &x)) return fail(p, badValue(key, value));
          *reinterpret_cast<double*>(base) = x;
        } break;
        case K_SCHEMA: {  // restart.c:454-464
          int32_t x;
          if (!parseInt(value, &x)) return fail(p, badValue(key, value));
          if (x != K.aux) {
            setError("Restart schema layout mismatch in " + p + ": key=" + key + " found=" +
                     std::to_string(x) + " expected=" + std::to_string(K.aux));
            return SIPNET_ERR_RESTART;
          }
        } break;
      }
      seen[hit] = 1;
      continue;
    }
    if (strcmp(key, "end_restart") == 0) {
      int32_t x;
      if (!parseInt(value, &x)) return fail(p, badValue(key, value));
      seenEnd = true;
      continue;
    }
    // ring arrays, restart.c:694-719
    const bool isVal = startsWith(key, "mean.npp.values.");
    const bool isWgt = !isVal && startsWith(key, "mean.npp.weights.");
    if (isVal || isWgt) {
      const char* idxText = key + strlen(isVal ? "mean.npp.values." : "mean.npp.weights.");
      int32_t idx;
      if (!parseInt(idxText, &idx)) return fail(p, badValue(key, idxText));
      if (idx < 0 || idx >= SIPNET_RING_SLOTS)
        return fail(p, std::string(isVal ? "mean.npp.values" : "mean.npp.weights") +
                           " index out of range (" + key + ")");
      std::vector<char>& sv = isVal ? seenVal : seenWgt;
      if (sv[idx]) return fail(p, std::string("duplicate key '") + key + "'");
      sv[idx] = 1;
      double x;
      if (!parseDbl(value, &x)) return fail(p, badValue(key, value));
      (isVal ? out->mean_values : out->mean_weights)[idx] = x;
      continue;
    }
    return fail(p, std::string("unknown key '") + key + "'");
  }

  // restart.c:727-733: the file may not resize the ring
  if (seen[kNumKeys - 5] && out->mean_length != SIPNET_RING_SLOTS) {
    setError("Restart schema mismatch in " + p + ": mean.npp.length (" +
             std::to_string(out->mean_length) + ") does not match the compiled model length (" +
             std::to_string(SIPNET_RING_SLOTS) + ")");
    return SIPNET_ERR_RESTART;
  }
  // every key present, restart.c:735-745, in the reference's group order
  static const int groupOrder[] = {0, 1, 2, 8, 3, 4, 5, 6, 7};
  for (int g : groupOrder) {
    for (int k = 0; k < kNumKeys; k++) {
      if (kKeys[k].group == g && !seen[k])
        return fail(p, std::string("missing required key (") + kKeys[k].name + ")");
    }
  }
  if (!seenEnd) return fail(p, "missing required key (end_restart)");
  for (int i = 0; i < SIPNET_RING_SLOTS; i++)
    if (!seenVal[i]) return fail(p, "mean.npp.values array is incomplete");
  for (int i = 0; i < SIPNET_RING_SLOTS; i++)
    if (!seenWgt[i]) return fail(p, "mean.npp.weights array is incomplete");
  return SIPNET_OK;
}

int sipnet_io_write_restart(const char* path, const sipnet_restart* in) {
  if (!path || !in) return SIPNET_ERR_BAD_ARGUMENT;
  FILE* f = fopen(path, "w");
  if (!f) {
    setError(std::string("Error opening ") + path + " for writing");
    return SIPNET_ERR_FILE_OPEN;
  }
  fprintf(f, "%s\n", kMagic);
  for (int k = 0; k < kNumKeys; k++) {
    const Key& K = kKeys[k];
    const char* base = reinterpret_cast<const char*>(in) + K.off;
    switch (K.kind) {
      case K_STR: {
        // one whitespace-free token (restart.c:391-404)
        std::string s(base, strnlen(base, (size_t)K.aux - 1));
        for (char& c : s)
          if (c == ' ' || c == '\t' || c == '\r' || c == '\n') c = '_';
        if (s.empty()) s = "unknown";
        fprintf(f, "%s %s\n", K.name, s.c_str());
      } break;
      case K_I64:
        fprintf(f, "%s %lld\n", K.name, (long long)*reinterpret_cast<const int64_t*>(base));
        break;
      case K_INT:
        fprintf(f, "%s %d\n", K.name, *reinterpret_cast<const int32_t*>(base));
        break;
      case K_DBL:
        fprintf(f, "%s %.17g\n", K.name, *reinterpret_cast<const double*>(base));
        break;
      case K_SCHEMA:
        fprintf(f, "%s %d\n", K.name, K.aux);
        break;
    }
    if (k + 1 == kNumKeys || kKeys[k + 1].group != K.group) fprintf(f, "\n");
  }
  for (int i = 0; i < SIPNET_RING_SLOTS; i++)
    fprintf(f, "mean.npp.values.%d %.17g\n", i, in->mean_values[i]);
  fprintf(f, "\n");
  for (int i = 0; i < SIPNET_RING_SLOTS; i++)
    fprintf(f, "mean.npp.weights.%d %.17g\n", i, in->mean_weights[i]);
  fprintf(f, "\n");
  fprintf(f, "end_restart 1\n");
  const bool bad = ferror(f) != 0;
  if (fclose(f) != 0 || bad) {
    setError(std::string("Error writing ") + path);
    return SIPNET_ERR_FILE_OPEN;
  }
  return SIPNET_OK;
}

int sipnet_restart_check_boundary_for_write(const sipnet_restart* r, int32_t* warnings) {
  if (!r) return SIPNET_ERR_BAD_ARGUMENT;
  if (warnings) *warnings = 0;
  const double stepHours = r->boundary_length * 24.0;
  if (stepHours <= kEps) {  // restart.c:347-354
    char buf[200];
    snprintf(buf, sizeof buf,
             "Cannot write restart checkpoint: non-positive timestep length at boundary "
             "(year=%d day=%d time=%.8f length=%.8f)",
             r->boundary_year, r->boundary_day, r->boundary_time, r->boundary_length);
    setError(buf);
    return SIPNET_ERR_RESTART;
  }
  if (warnings && 24.0 - r->boundary_time > stepHours + kEps)
    *warnings |= SIPNET_RESTART_WARN_BOUNDARY_NOT_MIDNIGHT;
  return SIPNET_OK;
}

int sipnet_restart_check(const sipnet_restart* r, const int32_t* flags, int32_t has_climate,
                         int32_t year0, int32_t day0, double time0, double length0,
                         int32_t* warnings) {
  if (!r || !flags) return SIPNET_ERR_BAD_ARGUMENT;
  int32_t warn = 0;
  char buf[320];
  // validateCheckpointBoundaryForLoad, restart.c:368-389
  const double stepHours = r->boundary_length * 24.0;
  if (stepHours <= kEps) {
    snprintf(buf, sizeof buf,
             "Restart boundary mismatch: checkpoint boundary has non-positive timestep length "
             "(year=%d day=%d time=%.8f length=%.8f)",
             r->boundary_year, r->boundary_day, r->boundary_time, r->boundary_length);
    setError(buf);
    return SIPNET_ERR_RESTART;
  }
  if (24.0 - r->boundary_time > stepHours + kEps) warn |= SIPNET_RESTART_WARN_BOUNDARY_NOT_MIDNIGHT;
  // checkRestartContextCompatibility, restart.c:830-852
  for (int i = 0; i < SIPNET_NFLAGS; i++) {
    if (r->flags[i] != flags[i]) {
      setError("Restart context mismatch: model flags must match checkpoint exactly");
      return SIPNET_ERR_RESTART;
    }
  }
  // validateRestartModelBuild, restart.c:854-867
  if (strcmp(r->model_version, kModelVersion) != 0) {
    setError(std::string("Restart model version mismatch: checkpoint=") + r->model_version +
             " current=" + kModelVersion);
    return SIPNET_ERR_RESTART;
  }
  {
    std::string mine = sipnet_version();
    for (char& c : mine)
      if (c == ' ' || c == '\t' || c == '\r' || c == '\n') c = '_';
    if (mine != r->build_info) warn |= SIPNET_RESTART_WARN_BUILD_INFO;
  }
  // validateRestartBoundary, restart.c:869-910
  if (!has_climate) {
    setError("Cannot restart: climate forcing has no records");
    return SIPNET_ERR_INPUT_FILE;
  }
  bool after;
  if (year0 != r->boundary_year) {
    after = year0 > r->boundary_year;
  } else if (day0 != r->boundary_day) {
    after = day0 > r->boundary_day;
  } else {
    after = time0 > r->boundary_time + kEps;
  }
  if (!after) {
    snprintf(buf, sizeof buf,
             "Restart boundary mismatch: first climate timestamp does not follow checkpoint "
             "boundary timestamp (checkpoint year=%d day=%d time=%.8f; found year=%d day=%d "
             "time=%.8f)",
             r->boundary_year, r->boundary_day, r->boundary_time, year0, day0, time0);
    setError(buf);
    return SIPNET_ERR_RESTART;
  }
  const double firstStepHours = length0 * 24.0;
  if (firstStepHours <= kEps) {
    snprintf(buf, sizeof buf,
             "Cannot restart: first climate timestep length is non-positive (year=%d day=%d "
             "time=%.8f length=%.8f)",
             year0, day0, time0, length0);
    setError(buf);
    return SIPNET_ERR_RESTART;
  }
  int ey = r->boundary_year, ed = r->boundary_day + 1;
  if (ed > (isLeap(ey) ? 366 : 365)) {
    ed = 1;
    ey++;
  }
  if (year0 != ey || day0 != ed || time0 > firstStepHours + kEps) warn |= SIPNET_RESTART_WARN_TIME_GAP;
  // ring cursors, restart.c:987-992
  if (r->mean_start < 0 || r->mean_start >= SIPNET_RING_SLOTS || r->mean_last < 0 ||
      r->mean_last >= SIPNET_RING_SLOTS) {
    setError("Restart mean-tracker cursor out of range");
    return SIPNET_ERR_RESTART;
  }
  if (warnings) *warnings = warn;
  return SIPNET_OK;
}

}  // extern "C"
