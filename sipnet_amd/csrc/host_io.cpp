// host_io.cpp -- SIPNET's text formats at the drop-in boundary (host only).
//
// Readers for `<prefix>.clim`, `<prefix>.param`, `events.in` and writers for
// `<prefix>.out`, behaving like the reference's (citations relative to
// /root/reference/src/), re-written around std::string / std::vector with
// status returns instead of exit().
#include <strings.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/sipnet_amd.h"

namespace sipnet {
void setError(const std::string& s);

namespace {
constexpr double kTiny = 0.000001;  // common/util.h:14

struct ParamDef {
  const char* name;
  int rule;
};
enum Rule {
  ALWAYS, NEVER, GROWTH_RESP, LEAF_DAY, GDD, SOIL_PHENOL, LITTER_POOL, WATER_HRESP,
  LEAF_WATER, SNOW, NCYCLE, ANAEROBIC_OR_NCYCLE, ANAEROBIC, FLOODING, CSAT
};
const ParamDef kParams[SIPNET_NPARAMS] = {
#define SIPNET_PARAM(idx, field, fname, rule) {fname, rule},
#include "../../include/sipnet_params.def"
#undef SIPNET_PARAM
};

// required-ness of a parameter under a flag set, sipnet.c:300-396
bool isRequired(int rule, const int32_t* f) {
  switch (rule) {
    case ALWAYS: return true;
    case NEVER: return false;
    case GROWTH_RESP: return f[SIPNET_F_GROWTH_RESP] != 0;
    case LEAF_DAY: return !(f[SIPNET_F_GDD] || f[SIPNET_F_SOIL_PHENOL]);
    case GDD: return f[SIPNET_F_GDD] != 0;
    case SOIL_PHENOL: return f[SIPNET_F_SOIL_PHENOL] != 0;
    case LITTER_POOL: return f[SIPNET_F_LITTER_POOL] != 0;
    case WATER_HRESP: return f[SIPNET_F_WATER_HRESP] != 0;
    case LEAF_WATER: return f[SIPNET_F_LEAF_WATER] != 0;
    case SNOW: return f[SIPNET_F_SNOW] != 0;
    case NCYCLE: return f[SIPNET_F_NITROGEN_CYCLE] != 0;
    case ANAEROBIC_OR_NCYCLE: return f[SIPNET_F_ANAEROBIC] || f[SIPNET_F_NITROGEN_CYCLE];
    case ANAEROBIC: return f[SIPNET_F_ANAEROBIC] != 0;
    case FLOODING: return f[SIPNET_F_FLOODING] != 0;
    case CSAT: return f[SIPNET_F_CARBON_SATURATION] != 0;
  }
  return false;
}

std::vector<std::string> splitFields(const std::string& line) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < line.size()) {
    while (i < line.size() && strchr(" \t\n\r", line[i])) i++;
    if (i >= line.size()) break;
    size_t j = i;
    while (j < line.size() && !strchr(" \t\n\r", line[j])) j++;
    out.push_back(line.substr(i, j - i));
    i = j;
  }
  return out;
}

// strtod over a whole token; false when trailing garbage remains (fscanf %lf
// semantics: a partial match fails the field)
bool parseDouble(const std::string& tok, double& v) {
  char* end = nullptr;
  v = strtod(tok.c_str(), &end);
  return end != tok.c_str() && *end == '\0';
}
bool parseInt(const std::string& tok, int& v) {
  char* end = nullptr;
  long x = strtol(tok.c_str(), &end, 10);
  v = (int)x;
  return end != tok.c_str() && *end == '\0';
}
}  // namespace
}  // namespace sipnet

using namespace sipnet;

struct sipnet_clim_table {
  std::vector<double> data;  // [n][SIPNET_NCLIM]
  std::vector<int32_t> year, day;
};

extern "C" {

// ---------------------------------------------------------------- .clim
// sipnet.c:128-277.  The reference reads the remainder of the file with fscanf,
// i.e. as one whitespace-separated token stream; line breaks are not significant
// after the first line, so the same is done here.
int sipnet_io_read_clim(const char* path, int32_t gdd_flag, sipnet_clim_table** out) {
  if (!path || !out) return SIPNET_ERR_BAD_ARGUMENT;
  std::ifstream in(path);
  if (!in) {
    setError(std::string("Error opening ") + path + " for reading");
    return SIPNET_ERR_FILE_OPEN;
  }
  std::string first;
  if (!std::getline(in, first)) {
    setError(std::string("no climate data in ") + path);
    return SIPNET_ERR_INPUT_FILE;
  }
  const std::vector<std::string> f0 = splitFields(first);
  int ncols;
  bool legacy;
  if (f0.size() == 12) {
    ncols = 12;
    legacy = false;
  } else if (f0.size() == 14) {
    ncols = 14;
    legacy = true;
  } else {
    setError("format unrecognized in climate file " + std::string(path) + "; " +
             std::to_string(f0.size()) + " columns found, expected 12 or 14 (legacy format)");
    return SIPNET_ERR_INPUT_FILE;
  }
  std::vector<std::string> toks = f0;
  {
    std::string rest((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::vector<std::string> more = splitFields(rest);
    toks.insert(toks.end(), more.begin(), more.end());
  }
  auto* tab = new sipnet_clim_table();
  const size_t nrec = toks.size() / ncols;
  int firstLoc = 0;
  int lastYear = 0, lastDay = 0;
  for (size_t r = 0; r <= nrec; r++) {
    const size_t base = r * ncols;
    if (base >= toks.size()) break;
    if (toks.size() - base < (size_t)ncols) {  // partial trailing record
      setError("while reading climate file: bad data near year " + std::to_string(lastYear) +
               " day " + std::to_string(lastDay));
      delete tab;
      return SIPNET_ERR_INPUT_FILE;
    }
    size_t k = base;
    int loc = 0, year, day;
    double v[10];  // time length tair tsoil par precip vpd vpdSoil vPress wspd
    bool ok = true;
    if (legacy) ok = ok && parseInt(toks[k++], loc);
    ok = ok && parseInt(toks[k++], year) && parseInt(toks[k++], day);
    for (int j = 0; j < 10 && ok; j++) ok = parseDouble(toks[k++], v[j]);
    if (legacy && ok) {
      double soilWetness;
      ok = parseDouble(toks[k++], soilWetness);
    }
    if (!ok) {
      setError(r == 0 ? std::string("while reading climate file: bad data on first line")
                      : "while reading climate file: bad data near year " +
                            std::to_string(lastYear) + " day " + std::to_string(lastDay));
      delete tab;
      return SIPNET_ERR_INPUT_FILE;
    }
    if (legacy) {
      if (r == 0) {
        firstLoc = loc;
      } else if (loc != firstLoc) {  // sipnet.c:258-264
        setError("while reading legacy climate file " + std::string(path) +
                 ": multiple locations not supported");
        delete tab;
        return SIPNET_ERR_INPUT_FILE;
      }
    }
    lastYear = year;
    lastDay = day;
    double time = v[0], length = v[1];
    const double tair = v[2], tsoil = v[3], par = v[4], precip = v[5];
    double vpd = v[6];
    const double vpdSoil = v[7], vPress = v[8];
    double wspd = v[9];
    // unit conversions, sipnet.c:209-238
    if (length < 0) length = length / -86400.;
    const double parRate = par * (1.0 / length);
    vpd = vpd * 0.001;
    if (vpd < kTiny) vpd = kTiny;
    if (wspd < kTiny) wspd = kTiny;
    double gdd = 0.0;
    if (gdd_flag) {
      gdd = tair * length;
      if (gdd < 0) gdd = 0;
    }
    const double rec[SIPNET_NCLIM] = {length, tair, tsoil, parRate, precip * 0.1, vpd,
                                      vpdSoil * 0.001, vPress * 0.001, wspd, gdd, time};
    tab->data.insert(tab->data.end(), rec, rec + SIPNET_NCLIM);
    tab->year.push_back(year);
    tab->day.push_back(day);
  }
  *out = tab;
  return SIPNET_OK;
}
int32_t sipnet_clim_nsteps(const sipnet_clim_table* t) { return t ? (int32_t)t->year.size() : 0; }
const double* sipnet_clim_data(const sipnet_clim_table* t) { return t ? t->data.data() : nullptr; }
const int32_t* sipnet_clim_year(const sipnet_clim_table* t) { return t ? t->year.data() : nullptr; }
const int32_t* sipnet_clim_day(const sipnet_clim_table* t) { return t ? t->day.data() : nullptr; }
void sipnet_clim_free(sipnet_clim_table* t) { delete t; }

// ---------------------------------------------------------------- .param
const char* sipnet_param_name(int32_t index) {
  if (index < 0 || index >= SIPNET_NPARAMS) return "";
  return kParams[index].name;
}
int32_t sipnet_param_index(const char* name) {  // modelParams.c:240-258
  if (!name || !*name) return -1;
  for (int i = 0; i < SIPNET_NPARAMS; i++)
    if (kParams[i].name[0] && strcasecmp(name, kParams[i].name) == 0) return i;
  return -1;
}

// modelParams.c:136-230 + sipnet.c:290-427
int sipnet_io_read_params(const char* path, const int32_t* flags, double* out,
                          int32_t* is_read) {
  if (!path || !flags || !out) return SIPNET_ERR_BAD_ARGUMENT;
  std::ifstream in(path);
  if (!in) {
    setError(std::string("Error opening ") + path + " for reading");
    return SIPNET_ERR_FILE_OPEN;
  }
  bool seen[SIPNET_NPARAMS] = {false};
  for (int i = 0; i < SIPNET_NPARAMS; i++) out[i] = 0.0;  // globals start at zero
  std::string line;
  while (std::getline(in, line)) {
    const size_t bang = line.find('!');
    if (bang != std::string::npos) line.erase(bang);
    const std::vector<std::string> tok = splitFields(line);
    if (tok.empty()) continue;
    if (tok.size() < 2) {
      setError("reading parameter file: no value for " + tok[0]);
      return SIPNET_ERR_INPUT_FILE;
    }
    if (tok[1] == "*") {
      setError("reading parameter " + tok[0] + "; '*' is no longer supported");
      return SIPNET_ERR_BAD_PARAMETER;
    }
    const int idx = sipnet_param_index(tok[0].c_str());
    if (idx < 0) continue;  // unknown names are ignored
    if (seen[idx]) {
      setError("reading parameter file: read " + tok[0] +
               ", but this parameter has already been set");
      return SIPNET_ERR_INPUT_FILE;
    }
    out[idx] = strtod(tok[1].c_str(), nullptr);
    seen[idx] = true;
  }
  std::string missing;
  for (int i = 0; i < SIPNET_NPARAMS; i++) {
    if (is_read) is_read[i] = seen[i] ? 1 : 0;
    if (!seen[i] && isRequired(kParams[i].rule, flags)) {
      if (!missing.empty()) missing += ", ";
      missing += kParams[i].name;
    }
  }
  if (!missing.empty()) {
    setError("Did not find required parameter(s): " + missing);
    return SIPNET_ERR_INPUT_FILE;
  }
  // divisor clamps, sipnet.c:404-424
  const int clamp[] = {25 /*cFracLeaf*/, 12 /*halfSatPar*/, 23 /*soilWHC*/, 24 /*leafCSpWt*/,
                       66 /*leafCN*/,    67 /*woodCN*/,     68 /*fineRootCN*/};
  for (int idx : clamp)
    if (out[idx] < kTiny) out[idx] = kTiny;
  return SIPNET_OK;
}

// ---------------------------------------------------------------- events.in
// events.c:39-184, :210-367
int sipnet_io_read_events(const char* path, const int32_t* flags, const double* params,
                          sipnet_event** out, int32_t* n_events) {
  if (!path || !flags || !out || !n_events) return SIPNET_ERR_BAD_ARGUMENT;
  *out = nullptr;
  *n_events = 0;
  std::ifstream in(path);
  if (!in) return SIPNET_OK;  // no file: no events (events.c:276-281)
  std::vector<sipnet_event> evs;
  std::string line;
  bool checkedLeaf = false;
  int curYear = 0, curDay = 0;
  while (std::getline(in, line)) {
    if (line.size() >= 1023) {  // events.c:246-252
      setError("Event line too long (exceeds 1024 chars), data may be truncated");
      return SIPNET_ERR_INPUT_FILE;
    }
    const std::vector<std::string> tok = splitFields(line);
    int year, day;
    if (tok.size() < 3 || !parseInt(tok[0], year) || !parseInt(tok[1], day)) {
      setError(evs.empty() ? std::string("reading event file: bad data on first line")
                           : "reading event file: bad data on line after year " +
                                 std::to_string(curYear) + " day " + std::to_string(curDay));
      return SIPNET_ERR_INPUT_FILE;
    }
    static const struct { const char* s; int t; int n; } kinds[] = {
        {"irrig", SIPNET_EV_IRRIG, 2}, {"fert", SIPNET_EV_FERT, 3},
        {"plant", SIPNET_EV_PLANT, 4}, {"till", SIPNET_EV_TILL, 1},
        {"harv", SIPNET_EV_HARVEST, 4}, {"leafon", SIPNET_EV_LEAFON, 0},
        {"leafoff", SIPNET_EV_LEAFOFF, 0}};
    int type = -1, nreq = 0;
    for (const auto& k : kinds)
      if (tok[2] == k.s) {
        type = k.t;
        nreq = k.n;
      }
    if (type < 0) {
      if (tok[2] == "plantdeath") {
        setError("PLANTDEATH event found, but not implemented as an input event");
        return SIPNET_ERR_INPUT_FILE;
      }
      setError("reading event file: unknown event type " + tok[2]);
      return SIPNET_ERR_UNKNOWN_EVENT;
    }
    if ((type == SIPNET_EV_LEAFON || type == SIPNET_EV_LEAFOFF) && !checkedLeaf) {
      // events.c:254-261: computed and scheduled leaf events are exclusive
      const double leafOnDay = params ? params[14] : 0.0, leafOffDay = params ? params[15] : 0.0;
      if (flags[SIPNET_F_GDD] || flags[SIPNET_F_SOIL_PHENOL] || leafOnDay > 0 || leafOffDay > 0) {
        setError("calculated leaf events are not compatible with user-specified leaf events "
                 "in event file");
        return SIPNET_ERR_BAD_PARAMETER;
      }
      checkedLeaf = true;
    }
    sipnet_event ev;
    memset(&ev, 0, sizeof ev);
    ev.type = type;
    ev.year = year;
    ev.day = day;
    int got = 0;
    for (size_t k = 3; k < tok.size() && got < 4; k++) {
      double v;
      if (type == SIPNET_EV_IRRIG && got == 1) {
        int m;
        if (!parseInt(tok[k], m)) break;
        v = m;
      } else if (!parseDouble(tok[k], v)) {
        break;
      }
      ev.p[got++] = v;
    }
    if (nreq == 0 ? got > 0 : got < nreq) {
      setError(std::string("parsing ") + tok[2] + " params for year " + std::to_string(year) +
               " day " + std::to_string(day));
      return SIPNET_ERR_INPUT_FILE;
    }
    if (type == SIPNET_EV_HARVEST && ((ev.p[0] + ev.p[2] > 1) || (ev.p[1] + ev.p[3] > 1))) {
      setError("invalid harvest event for year " + std::to_string(year) + " day " +
               std::to_string(day) + "; above and below must each add to 1 or less");
      return SIPNET_ERR_BAD_PARAMETER;
    }
    if (!evs.empty() && ((year < curYear) || ((year == curYear) && (day < curDay)))) {
      setError("event records must be in time-ascending order");
      return SIPNET_ERR_INPUT_FILE;
    }
    evs.push_back(ev);
    curYear = year;
    curDay = day;
  }
  if (!evs.empty()) {
    *out = (sipnet_event*)malloc(evs.size() * sizeof(sipnet_event));
    memcpy(*out, evs.data(), evs.size() * sizeof(sipnet_event));
    *n_events = (int32_t)evs.size();
  }
  return SIPNET_OK;
}
void sipnet_io_free(void* p) { free(p); }

// ---------------------------------------------------------------- .out
// outputHeader(), sipnet.c:434-444
int sipnet_io_format_out_header(char* buf, size_t cap) {
  static const char hdr[] =
      "year day  time plantWoodC plantLeafC woodCreation     "
      "soil coarseRootC fineRootC   "
      "litter  soilWater soilWetnessFrac     snow      "
      "npp      nee   cumNEE      gpp rAboveground    rSoil    "
      "rRoot       ra       rh     rtot evapotranspiration "
      "fluxestranspiration     minN  soilOrgN    litterN  "
      "plantStorageN       n2o nLeaching  nFixation  nUptake      ch4  "
      "nppStorage\n";
  const size_t n = sizeof(hdr) - 1;
  if (n + 1 > cap) return -1;
  memcpy(buf, hdr, n + 1);
  return (int)n;
}

// outputState(), sipnet.c:453-473
int sipnet_io_format_out_row(char* buf, size_t cap, int32_t year, int32_t day, double time,
                             const double* rec, int64_t st) {
  auto R = [&](int k) { return rec[(int64_t)k * st]; };
  const double totalWood = R(14) + R(26);  // state.c:17-19
  const int n = snprintf(
      buf, cap,
      "%4d %3d %5.2f %10.2f %10.2f %12.2f "
      "%8.2f "
      "%11.2f %9.2f "
      "%8.2f %10.3f %15.3f %8.2f "
      "%8.3f %8.3f %8.3f %8.3f %12.3f %8.3f %8.3f %8.3f %8.3f %8.3f %18.8f "
      "%19.4f %8.4f %9.4f %10.4f %14.4f "
      "%9.6f %9.4f %10.4f %8.4f %8.4f"
      "%12.4f\n",
      year, day, time, totalWood, R(15), R(11),  //
      R(16),                                      // soilC
      R(20), R(21),                               // coarse, fine roots
      R(18), R(17), R(12), R(19),                 // litter soilWater wetness snow
      R(4), R(0), R(3), R(1), R(5), R(6), R(7), R(8), R(9), R(10), R(2),  //
      R(13), R(22), R(23), R(24), R(25),          // transpiration + N pools
      R(27), R(28), R(29), R(30), R(31),          // n2o nLeaching nFixation nUptake ch4
      R(26));
  if (n < 0 || (size_t)n >= cap) return -1;
  return n;
}

int sipnet_io_write_out(const char* path, int32_t print_header, int32_t n_steps,
                        const int32_t* year, const int32_t* day, const double* clim,
                        const double* rec) {
  FILE* f = fopen(path, "w");
  if (!f) {
    setError(std::string("Error opening ") + path + " for writing");
    return SIPNET_ERR_FILE_OPEN;
  }
  char buf[1024];
  if (print_header) {
    const int n = sipnet_io_format_out_header(buf, sizeof buf);
    fwrite(buf, 1, (size_t)n, f);
  }
  for (int t = 0; t < n_steps; t++) {
    const int n = sipnet_io_format_out_row(buf, sizeof buf, year[t], day[t],
                                           clim[(size_t)t * SIPNET_NCLIM + 10],
                                           rec + (size_t)t * SIPNET_NREC, 1);
    if (n < 0) {
      fclose(f);
      return SIPNET_ERR_INTERNAL;
    }
    fwrite(buf, 1, (size_t)n, f);
  }
  fclose(f);
  return SIPNET_OK;
}


// ---------------------------------------------------------------- --debug-log files
// debug_log.c:40-166 (field tables), :181-212 (formats), :258-312 (what goes in which file)
int sipnet_io_write_debug_logs(const char* prefix, int32_t print_header, int32_t n_steps,
                               const int32_t* year, const int32_t* day, const double* clim,
                               const double* rec, const double* dbg) {
  static const char* const enviNames[13] = {
      "plantWoodC", "plantLeafC", "soilC", "soilWater", "litterC", "snow", "coarseRootC",
      "fineRootC", "minN", "soilOrgN", "litterN", "plantStorageN", "plantCAccountingDelta"};
  static const char* const fluxNames[56] = {
      "photosynthesis", "leafLitter", "woodLitter", "rVeg", "rSoil", "rain", "transpiration",
      "drainage", "litterToSoil", "rLitter", "snowFall", "snowMelt", "sublimation",
      "immedEvap", "fastFlow", "evaporation", "fineRootLoss", "coarseRootLoss",
      "fineRootCreation", "coarseRootCreation", "rCoarseRoot", "rFineRoot", "leafCreation",
      "woodCreation", "leafOnCreation", "leafOnCreationFromWood", "nVolatilization",
      "nLeaching", "nOrgSoil", "nOrgLitter", "nMin", "nFixation", "nUptake",
      "leafOffNResorption", "reductionNResorption", "eventLeafC", "eventWoodC",
      "eventFineRootC", "eventCoarseRootC", "eventEvap", "eventSoilWater", "eventSoilC",
      "eventLitterC", "eventMinN", "eventSoilOrgN", "eventLitterN", "eventInputC",
      "eventOutputC", "eventInputN", "eventOutputN", "eventLeafOnCreation",
      "eventLeafOnCreationFromWood", "eventLeafOffLitter", "eventLeafOffNResorption",
      "soilMethane", "litterMethane"};
  // tracker fields: name, source (record column r, debug column d, or the year y)
  struct TF { const char* name; char src; int col; };
  static const TF trk[33] = {
      {"gpp", 'r', 1}, {"rtot", 'r', 10}, {"ra", 'r', 8}, {"rh", 'r', 9}, {"rRoot", 'r', 7},
      {"rSoil", 'r', 6}, {"rAboveground", 'r', 5}, {"npp", 'r', 4}, {"nee", 'r', 0},
      {"woodCreation", 'r', 11}, {"gdd", 'r', 33}, {"evapotranspiration", 'r', 2},
      {"soilWetnessFrac", 'r', 12}, {"yearlyGpp", 'd', 56}, {"yearlyRtot", 'd', 57},
      {"yearlyRa", 'd', 58}, {"yearlyRh", 'd', 59}, {"yearlyNpp", 'd', 60},
      {"yearlyNee", 'd', 61}, {"yearlyLitter", 'd', 62}, {"totGpp", 'r', 35},
      {"totRtot", 'd', 63}, {"totRa", 'd', 64}, {"totRh", 'd', 65}, {"totNpp", 'd', 66},
      {"totNee", 'r', 3}, {"lastYear", 'y', 0}, {"methane", 'r', 31}, {"n2o", 'r', 27},
      {"nLeaching", 'r', 28}, {"nFixation", 'r', 29}, {"nUptake", 'r', 30},
      {"meanNPP", 'r', 32}};
  const std::string pre(prefix ? prefix : "");
  if (pre.size() + 13 >= 256) {  // FILENAME_MAXLEN, debug_log.c:169-178
    setError("debug-log prefix '" + pre + "' is too long");
    return SIPNET_ERR_BAD_PARAMETER;
  }
  FILE* fe = fopen((pre + "_envi.log").c_str(), "w");
  FILE* ff = fopen((pre + "_fluxes.log").c_str(), "w");
  FILE* ft = fopen((pre + "_trackers.log").c_str(), "w");
  if (!fe || !ff || !ft) {
    if (fe) fclose(fe);
    if (ff) fclose(ff);
    if (ft) fclose(ft);
    setError("Error opening debug log files with prefix " + pre);
    return SIPNET_ERR_FILE_OPEN;
  }
  if (print_header) {  // sipnet.c:1959-1961
    fprintf(fe, "year day time");
    for (const char* n : enviNames) fprintf(fe, " %s", n);
    fprintf(fe, "\n");
    fprintf(ff, "year day time");
    for (const char* n : fluxNames) fprintf(ff, " %s", n);
    fprintf(ff, "\n");
    fprintf(ft, "year day time");
    for (const TF& t : trk) fprintf(ft, " t.%s", t.name);
    fprintf(ft, " pt.didLeafGrowth pt.didLeafFall pt.lastYear s.isAlive\n");
  }
  for (int t = 0; t < n_steps; t++) {
    const double* r = rec + (size_t)t * SIPNET_NREC;
    const double* d = dbg + (size_t)t * SIPNET_NDBG;
    const double time = clim[(size_t)t * SIPNET_NCLIM + 10];
    fprintf(fe, "%4d %3d %5.2f", year[t], day[t], time);
    for (int k = 0; k < 13; k++) fprintf(fe, " %.15g", r[14 + k]);
    fprintf(fe, "\n");
    fprintf(ff, "%4d %3d %5.2f", year[t], day[t], time);
    for (int k = 0; k < 56; k++) fprintf(ff, " %.15g", d[k]);
    fprintf(ff, "\n");
    fprintf(ft, "%4d %3d %5.2f", year[t], day[t], time);
    for (const TF& f : trk) {
      if (f.src == 'y') fprintf(ft, " %d", year[t]);
      else fprintf(ft, " %.15g", f.src == 'r' ? r[f.col] : d[f.col]);
    }
    fprintf(ft, " %d %d %d %d\n", (int)d[67], (int)d[68], year[t], (int)d[69]);
  }
  fclose(fe);
  fclose(ff);
  fclose(ft);
  return SIPNET_OK;
}

// ---------------------------------------------------------------- events.out
namespace {
struct EvLine {
  FILE* f;
  // events.c:381-407: "%4d  %3d  %-7s  " then name=%-.2f pairs joined by commas
  void write(int year, int day, const char* type, int n, const char* const* names,
             const double* vals) const {
    fprintf(f, "%4d  %3d  %-7s  ", year, day, type);
    for (int i = 0; i < n - 1; i++) fprintf(f, "%s=%-.2f,", names[i], vals[i]);
    fprintf(f, "%s=%-.2f\n", names[n - 1], vals[n - 1]);
  }
};
const char* evTypeName(int t) {  // events.c:186-208
  switch (t) {
    case SIPNET_EV_IRRIG: return "irrig";
    case SIPNET_EV_PLANT: return "plant";
    case SIPNET_EV_HARVEST: return "harv";
    case SIPNET_EV_FERT: return "fert";
    case SIPNET_EV_TILL: return "till";
    case SIPNET_EV_LEAFON: return "leafon";
    case SIPNET_EV_LEAFOFF: return "leafoff";
  }
  return "plantdeath";
}
}  // namespace

// Regenerates what the reference prints while stepping (events.c:484-742 for input events,
// sipnet.c:829-841 and :1230-1247 for computed leaf events, :1759-1765 for plant death),
// in the reference's order within a step.
int sipnet_io_write_events_out(const char* path, int32_t print_header, const int32_t* flags,
                               const double* P, int32_t n_steps, const int32_t* year,
                               const int32_t* day, const double* clim, int32_t n_events,
                               const sipnet_event* events, const double* rec,
                               const double* init_pools) {
  if (!path || !flags || !P || !year || !day || !clim || !rec || !init_pools)
    return SIPNET_ERR_BAD_ARGUMENT;
  FILE* f = fopen(path, "w");
  if (!f) {
    setError(std::string("Error opening ") + path + " for writing");
    return SIPNET_ERR_FILE_OPEN;
  }
  if (print_header)  // events.c:371-378
    fprintf(f, "%4s  %3s  %-7s  %s", "year", "day", "type",
            "param_name=delta[,param_name=delta,...]\n");
  const EvLine out{f};
  const bool litter = flags[SIPNET_F_LITTER_POOL] != 0, ncyc = flags[SIPNET_F_NITROGEN_CYCLE] != 0;
  const double immedEvapFrac = P[31], leafCN = P[66], woodCN = P[67], fineRootCN = P[68];
  const double fracLeafFall = P[56], leafNResorptionFrac = P[73];
  int evNext = 0;
  if (!flags[SIPNET_F_EVENTS]) n_events = 0;
  for (int t = 0; t < n_steps; t++) {
    const double* r = rec + (size_t)t * SIPNET_NREC;
    const double* pools = t == 0 ? init_pools : rec + (size_t)(t - 1) * SIPNET_NREC + 14;
    const double plantWoodC = pools[0], plantLeafC = pools[1], fineRootC = pools[7],
                 coarseRootC = pools[6], delta = pools[12];
    const double len = clim[(size_t)t * SIPNET_NCLIM];
    double harvRemoved = 0, harvTransferred = 0;
    while (evNext < n_events && events[evNext].year <= year[t] && events[evNext].day <= day[t]) {
      const sipnet_event& ev = events[evNext++];
      const char* ty = evTypeName(ev.type);
      switch (ev.type) {
        case SIPNET_EV_IRRIG: {
          const double amount = ev.p[0];
          const double evap = ((int)ev.p[1] == 0) ? immedEvapFrac * amount : 0.0;
          const char* n[] = {"eventSoilWater", "eventEvap"};
          const double v[] = {amount - evap, evap};
          out.write(ev.year, ev.day, ty, 2, n, v);
        } break;
        case SIPNET_EV_PLANT: {
          const double inC = ev.p[0] + ev.p[1] + ev.p[2] + ev.p[3];
          const double inN = ncyc ? ev.p[0] / leafCN + ev.p[1] / woodCN + ev.p[2] / fineRootCN +
                                        ev.p[3] / woodCN
                                  : 0.0;
          const char* n[] = {"eventLeafC", "eventWoodC", "eventFineRootC", "eventCoarseRootC",
                             "eventInputC", "eventInputN"};
          const double v[] = {ev.p[0], ev.p[1], ev.p[2], ev.p[3], inC, inN};
          out.write(ev.year, ev.day, ty, 6, n, v);
        } break;
        case SIPNET_EV_HARVEST: {
          const double fRA = ev.p[0], fRB = ev.p[1], fTA = ev.p[2], fTB = ev.p[3];
          const double woodC = plantWoodC + delta;
          const double above = woodC + plantLeafC, below = fineRootC + coarseRootC;
          if (above + below > kTiny) {
            harvRemoved += (fRA * above + fRB * below) / (above + below);
            harvTransferred += (fTA * above + fTB * below) / (above + below);
          }
          double litterAdd = fTA * (plantLeafC + woodC), soilAdd = fTB * (fineRootC + coarseRootC);
          if (!litter) {
            soilAdd += litterAdd;
            litterAdd = 0.0;
          }
          double soilN = 0, litterN = 0, outN = 0;
          if (ncyc) {
            litterN = fTA * ((plantLeafC / leafCN) + (plantWoodC / woodCN));
            soilN = fTB * ((fineRootC / fineRootCN) + (coarseRootC / woodCN));
            outN = (plantWoodC / woodCN + plantLeafC / leafCN) * fRA +
                   (fineRootC / fineRootCN + coarseRootC / woodCN) * fRB;
          }
          const double outC = (woodC + plantLeafC) * fRA + (fineRootC + coarseRootC) * fRB;
          const char* n[] = {"eventSoilC", "eventLitterC", "eventLeafC", "eventWoodC",
                             "eventFineRootC", "eventCoarseRootC", "eventSoilOrgN",
                             "eventLitterN", "eventOutputC", "eventOutputN"};
          const double v[] = {soilAdd, litterAdd, -plantLeafC * (fRA + fTA), -woodC * (fRA + fTA),
                              -fineRootC * (fRB + fTB), -coarseRootC * (fRB + fTB), soilN, litterN,
                              outC, outN};
          out.write(ev.year, ev.day, ty, 10, n, v);
        } break;
        case SIPNET_EV_TILL: {
          const char* n[] = {"eventTrackers.d_till_mod"};
          const double v[] = {ev.p[0]};
          out.write(ev.year, ev.day, ty, 1, n, v);
        } break;
        case SIPNET_EV_FERT: {
          const double orgC = ev.p[1], orgN = ncyc ? ev.p[0] : 0.0, minN = ncyc ? ev.p[2] : 0.0;
          const char* n[] = {"eventLitterC", "eventSoilC", "eventMinN", "eventLitterN",
                             "eventInputC", "eventInputN"};
          const double v[] = {litter ? orgC : 0.0, litter ? 0.0 : orgC, minN, orgN, orgC,
                              orgN + minN};
          out.write(ev.year, ev.day, ty, 6, n, v);
        } break;
        case SIPNET_EV_LEAFOFF: {
          const double leafOff = plantLeafC * fracLeafFall;
          double resorb = 0, litterN = 0;
          if (ncyc) {
            const double leafN = leafOff / leafCN;
            resorb = leafN * leafNResorptionFrac;
            litterN = leafN - resorb;
          }
          const char* n[] = {"eventLeafOffLitter", "eventLeafOffNResorption", "eventLitterN"};
          const double v[] = {leafOff, resorb, litterN};
          out.write(ev.year, ev.day, ty, 3, n, v);
        } break;
        default:  // leaf-on is written after the N-limitation check (below)
          break;
      }
    }
    // computed leaf-off (sipnet.c:836-840), then the delayed leaf-on lines (sipnet.c:1230-1247)
    if (flags[SIPNET_F_EVENTS]) {
      if (r[38] / len > kTiny) {
        const char* n[] = {"leafLitter"};
        const double v[] = {r[38]};
        out.write(year[t], day[t], "leafoff", 1, n, v);
      }
      if (r[36] / len > kTiny) {
        const char* n[] = {"leafOnCreation", "leafOnCreationFromWood"};
        const double v[] = {r[36], r[37]};
        out.write(year[t], day[t], "leafon", 2, n, v);
      }
      if (r[39] / len > kTiny) {
        const char* n[] = {"eventLeafOnCreation", "eventLeafOnCreationFromWood"};
        const double v[] = {r[39], r[40]};
        out.write(year[t], day[t], "leafon", 2, n, v);
      }
      if (r[43] != 0.0) {
        const char* n[] = {"harvestFracRemoved", "harvestFracTransferred", "totalWoodC",
                           "totalRootC"};
        const double v[] = {harvRemoved, harvTransferred, r[41], r[42]};
        out.write(year[t], day[t], "plantdeath", 4, n, v);
      }
    }
  }
  fclose(f);
  return SIPNET_OK;
}

}  // extern "C"
