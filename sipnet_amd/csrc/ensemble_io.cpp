// ensemble_io.cpp -- the ensemble output block: every member's outputs in ONE
// self-describing file instead of one `<prefix>.out` text file per member (sipnet.c:434-473,
// outputItems.c:126-150 write text per process; a 10 240-member year is 20 GB of it).
//
// NetCDF-3 "classic" written by hand (no library; the format is a fixed header followed by the
// variables' big-endian arrays): CDF-2 (64-bit offsets), or CDF-5 (64-bit sizes) when a variable
// would exceed CDF-2's 4 GiB.  Dimensions (time, member); coordinate variables year(time) i4,
// day(time) i4, hour(time) f8, length(time) f8, member(member) i4; data variables
// <name>(time, member) f8 or f4 with a `units` attribute -- names and units of the `.out` header
// (sipnet.c:434-452).  Variables are fixed-size and contiguous, so a writer may fill them in any
// order, a range of members or of steps at a time, from several threads (pwrite): device shards
// stream their own member ranges, one variable at a time, and nothing the size of the whole block
// is ever held in host memory.  Host only: no HIP in this file.
#include <fcntl.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sipnet_amd.h"

namespace sipnet {
void setError(const std::string& s);
}
using sipnet::setError;

namespace {

// `.out` column -> record column(s) and units; order of outputHeader(), sipnet.c:434-444
struct OutColumn {
  const char* name;
  int rec0, rec1;   // value = rec[rec0] (+ rec[rec1] when rec1 >= 0)
  const char* units;
};
const OutColumn kOutColumns[] = {
    {"plantWoodC", 14, 26, "g C m-2"},   // printed as total wood = plantWoodC + accounting delta (state.c:17-19)
    {"plantLeafC", 15, -1, "g C m-2"},
    {"woodCreation", 11, -1, "g C m-2 step-1"},
    {"soil", 16, -1, "g C m-2"},
    {"coarseRootC", 20, -1, "g C m-2"},
    {"fineRootC", 21, -1, "g C m-2"},
    {"litter", 18, -1, "g C m-2"},
    {"soilWater", 17, -1, "cm"},
    {"soilWetnessFrac", 12, -1, "1"},
    {"snow", 19, -1, "cm water equiv."},
    {"npp", 4, -1, "g C m-2 step-1"},
    {"nee", 0, -1, "g C m-2 step-1"},
    {"cumNEE", 3, -1, "g C m-2"},
    {"gpp", 1, -1, "g C m-2 step-1"},
    {"rAboveground", 5, -1, "g C m-2 step-1"},
    {"rSoil", 6, -1, "g C m-2 step-1"},
    {"rRoot", 7, -1, "g C m-2 step-1"},
    {"ra", 8, -1, "g C m-2 step-1"},
    {"rh", 9, -1, "g C m-2 step-1"},
    {"rtot", 10, -1, "g C m-2 step-1"},
    {"evapotranspiration", 2, -1, "cm step-1"},
    {"fluxestranspiration", 13, -1, "cm day-1"},
    {"minN", 22, -1, "g N m-2"},
    {"soilOrgN", 23, -1, "g N m-2"},
    {"litterN", 24, -1, "g N m-2"},
    {"plantStorageN", 25, -1, "g N m-2"},
    {"n2o", 27, -1, "g N m-2 step-1"},
    {"nLeaching", 28, -1, "g N m-2 step-1"},
    {"nFixation", 29, -1, "g N m-2 step-1"},
    {"nUptake", 30, -1, "g N m-2 step-1"},
    {"ch4", 31, -1, "g C m-2 step-1"},
    {"nppStorage", 26, -1, "g C m-2"},
};
constexpr int kNumOutColumns = sizeof(kOutColumns) / sizeof(kOutColumns[0]);

enum : uint32_t { NC_CHAR = 2, NC_INT = 4, NC_FLOAT = 5, NC_DOUBLE = 6, NC_DIMENSION = 10, NC_VARIABLE = 11, NC_ATTRIBUTE = 12 };

struct Header {   // big-endian byte stream
  std::vector<unsigned char> b;
  bool wide;      // CDF-5: sizes and counts are 64-bit
  void u32(uint32_t v) {
    for (int k = 3; k >= 0; k--) b.push_back((unsigned char)(v >> (8 * k)));
  }
  void u64(uint64_t v) {
    for (int k = 7; k >= 0; k--) b.push_back((unsigned char)(v >> (8 * k)));
  }
  void count(uint64_t v) { wide ? u64(v) : u32((uint32_t)v); }   // NON_NEG
  void text(const std::string& s) {
    b.insert(b.end(), s.begin(), s.end());
    while (b.size() % 4) b.push_back(0);
  }
  void name(const std::string& s) {
    count(s.size());
    text(s);
  }
  void charAttr(const std::string& key, const std::string& val) {
    name(key);
    u32(NC_CHAR);
    count(val.size());
    text(val);
  }
};

struct Var {
  std::string name, units;
  uint32_t type;
  int dim0, dim1;     // dimension ids (0 time, 1 member); dim1 < 0: one dimension
  uint64_t bytes;     // unpadded
  uint64_t begin;
};

uint64_t pad4(uint64_t n) { return (n + 3) & ~(uint64_t)3; }

bool writeAll(int fd, const void* p, size_t n, uint64_t off) {
  const unsigned char* c = (const unsigned char*)p;
  while (n) {
    const ssize_t w = pwrite(fd, c, n, (off_t)off);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    c += w;
    off += (uint64_t)w;
    n -= (size_t)w;
  }
  return true;
}

}  // namespace

struct sipnet_ensemble_file {
  int fd = -1;
  std::string path;
  int32_t nSteps = 0, nMembers = 0, nVars = 0;
  bool storeF32 = false;
  std::vector<Var> vars;   // coordinate variables first, then the nVars data variables
  int firstData = 0;
};

extern "C" {

int32_t sipnet_io_out_column_count(void) { return kNumOutColumns; }

int sipnet_io_out_column(int32_t k, const char** name, int32_t* rec0, int32_t* rec1, const char** units) {
  if (k < 0 || k >= kNumOutColumns) {
    setError("sipnet_io_out_column: index out of range");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (name) *name = kOutColumns[k].name;
  if (rec0) *rec0 = kOutColumns[k].rec0;
  if (rec1) *rec1 = kOutColumns[k].rec1;
  if (units) *units = kOutColumns[k].units;
  return SIPNET_OK;
}

int32_t sipnet_io_out_column_index(const char* name) {
  if (!name) return -1;
  for (int k = 0; k < kNumOutColumns; k++)
    if (strcmp(kOutColumns[k].name, name) == 0) return k;
  return -1;
}

int sipnet_io_ensemble_create(const char* path, int32_t n_steps, int32_t n_members, const int32_t* year, const int32_t* day,
                              const double* clim, const int32_t* member_ids, int32_t n_vars, const char* const* names,
                              const char* const* units, int32_t store_f32, const char* attrs, sipnet_ensemble_file** out) {
  if (!path || n_steps <= 0 || n_members <= 0 || !year || !day || !clim || n_vars <= 0 || !names || !out) {
    setError("sipnet_io_ensemble_create: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const bool forceWide = (store_f32 & SIPNET_NC_FORCE_CDF5) != 0;
  store_f32 &= SIPNET_NC_F32;
  const uint64_t T = (uint64_t)n_steps, M = (uint64_t)n_members, esize = store_f32 ? 4 : 8;
  std::vector<Var> vars;
  vars.push_back({"year", "year", NC_INT, 0, -1, T * 4, 0});
  vars.push_back({"day", "day of year", NC_INT, 0, -1, T * 4, 0});
  vars.push_back({"hour", "hour of day at step start", NC_DOUBLE, 0, -1, T * 8, 0});
  vars.push_back({"length", "days", NC_DOUBLE, 0, -1, T * 8, 0});
  vars.push_back({"member", "", NC_INT, 1, -1, M * 4, 0});
  const int firstData = (int)vars.size();
  for (int v = 0; v < n_vars; v++) {
    if (!names[v] || !names[v][0]) {
      setError("sipnet_io_ensemble_create: a variable without a name");
      return SIPNET_ERR_BAD_ARGUMENT;
    }
    for (const Var& o : vars)
      if (o.name == names[v]) {
        setError(std::string("sipnet_io_ensemble_create: variable name used twice: ") + names[v]);
        return SIPNET_ERR_BAD_ARGUMENT;
      }
    std::string u = (units && units[v]) ? units[v] : "";
    if (!(units && units[v])) {
      const int k = sipnet_io_out_column_index(names[v]);
      if (k >= 0) u = kOutColumns[k].units;
    }
    vars.push_back({names[v], u, store_f32 ? NC_FLOAT : NC_DOUBLE, 0, 1, T * M * esize, 0});
  }
  // CDF-2 holds a fixed-size variable of up to 2^32 - 4 bytes; beyond that the 64-bit-data format
  bool wide = forceWide;
  for (const Var& v : vars)
    if (pad4(v.bytes) > 0xFFFFFFFCull) wide = true;
  // global attributes: "key=value" lines
  std::vector<std::pair<std::string, std::string>> gatts = {{"title", "SIPNET ensemble outputs (sipnet_amd)"},
                                                            {"model_version", "2.1.0"},
                                                            {"Conventions", "member-resolved SIPNET .out columns, sipnet.c:434-473"}};
  if (attrs) {
    std::string s(attrs);
    size_t pos = 0;
    while (pos < s.size()) {
      size_t nl = s.find('\n', pos);
      if (nl == std::string::npos) nl = s.size();
      const std::string line = s.substr(pos, nl - pos);
      pos = nl + 1;
      const size_t eq = line.find('=');
      if (eq == std::string::npos || eq == 0) continue;
      gatts.push_back({line.substr(0, eq), line.substr(eq + 1)});
    }
  }
  // the header is laid out twice: once to learn its size, once with the variables' offsets
  Header h;
  for (int pass = 0; pass < 2; pass++) {
    h.b.clear();
    h.wide = wide;
    h.b.push_back('C');
    h.b.push_back('D');
    h.b.push_back('F');
    h.b.push_back(wide ? 5 : 2);
    h.count(0);   // numrecs: no record dimension
    h.u32(NC_DIMENSION);
    h.count(2);
    h.name("time");
    h.count(T);
    h.name("member");
    h.count(M);
    h.u32(NC_ATTRIBUTE);
    h.count(gatts.size());
    for (auto& kv : gatts) h.charAttr(kv.first, kv.second);
    h.u32(NC_VARIABLE);
    h.count(vars.size());
    for (const Var& v : vars) {
      h.name(v.name);
      h.count(v.dim1 >= 0 ? 2 : 1);
      h.count((uint64_t)v.dim0);
      if (v.dim1 >= 0) h.count((uint64_t)v.dim1);
      if (v.units.empty()) {
        h.u32(0);   // ABSENT
        h.count(0);
      } else {
        h.u32(NC_ATTRIBUTE);
        h.count(1);
        h.charAttr("units", v.units);
      }
      h.u32(v.type);
      h.count(pad4(v.bytes));
      h.u64(v.begin);
    }
    uint64_t off = pad4(h.b.size());
    for (Var& v : vars) {
      v.begin = off;
      off += pad4(v.bytes);
    }
  }
  const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
  if (fd < 0) {
    setError(std::string("Error opening ") + path + " for writing");
    return SIPNET_ERR_FILE_OPEN;
  }
  auto fail = [&](const char* what) {
    setError(std::string("sipnet_io_ensemble_create: ") + what + " " + path + ": " + strerror(errno));
    close(fd);
    return SIPNET_ERR_FILE_OPEN;
  };
  const Var& lastVar = vars.back();
  if (ftruncate(fd, (off_t)(lastVar.begin + pad4(lastVar.bytes))) != 0) return fail("cannot size");
  if (!writeAll(fd, h.b.data(), h.b.size(), 0)) return fail("cannot write the header of");
  // coordinate variables
  {
    Header c;
    c.wide = false;
    for (uint64_t t = 0; t < T; t++) c.u32((uint32_t)year[t]);
    if (!writeAll(fd, c.b.data(), c.b.size(), vars[0].begin)) return fail("cannot write");
    c.b.clear();
    for (uint64_t t = 0; t < T; t++) c.u32((uint32_t)day[t]);
    if (!writeAll(fd, c.b.data(), c.b.size(), vars[1].begin)) return fail("cannot write");
    for (int which = 0; which < 2; which++) {   // hour = clim column 10, length = clim column 0
      c.b.clear();
      for (uint64_t t = 0; t < T; t++) {
        uint64_t bits;
        const double x = clim[t * SIPNET_NCLIM + (which == 0 ? 10 : 0)];
        memcpy(&bits, &x, 8);
        c.u64(bits);
      }
      if (!writeAll(fd, c.b.data(), c.b.size(), vars[2 + which].begin)) return fail("cannot write");
    }
    c.b.clear();
    for (uint64_t m = 0; m < M; m++) c.u32((uint32_t)(member_ids ? member_ids[m] : (int32_t)m));
    if (!writeAll(fd, c.b.data(), c.b.size(), vars[4].begin)) return fail("cannot write");
  }
  sipnet_ensemble_file* f = new sipnet_ensemble_file;
  f->fd = fd;
  f->path = path;
  f->nSteps = n_steps;
  f->nMembers = n_members;
  f->nVars = n_vars;
  f->storeF32 = store_f32 != 0;
  f->vars = std::move(vars);
  f->firstData = firstData;
  *out = f;
  return SIPNET_OK;
}

int sipnet_io_ensemble_put(sipnet_ensemble_file* f, int32_t var, int32_t step0, int32_t n_steps, int32_t member0,
                           int32_t n_members, const void* data, int64_t ld, int32_t data_is_f32) {
  if (!f || f->fd < 0 || var < 0 || var >= f->nVars || step0 < 0 || n_steps <= 0 || step0 + n_steps > f->nSteps ||
      member0 < 0 || n_members <= 0 || member0 + n_members > f->nMembers || !data || ld < n_members) {
    setError("sipnet_io_ensemble_put: bad argument");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  const Var& v = f->vars[f->firstData + var];
  const size_t esize = f->storeF32 ? 4 : 8;
  const size_t M = (size_t)f->nMembers;
  // rows are converted (precision, byte order) in blocks of a few MB; a full-width put is one contiguous run
  const bool fullWidth = member0 == 0 && n_members == f->nMembers;
  const size_t rowBytes = (size_t)n_members * esize;
  size_t rowsPerBlock = (4u << 20) / rowBytes;
  if (rowsPerBlock < 1) rowsPerBlock = 1;
  if (!fullWidth) rowsPerBlock = 1;
  std::vector<unsigned char> buf(rowsPerBlock * rowBytes);
  for (int32_t t0 = 0; t0 < n_steps; t0 += (int32_t)rowsPerBlock) {
    const int32_t nr = (int32_t)((size_t)(n_steps - t0) < rowsPerBlock ? (size_t)(n_steps - t0) : rowsPerBlock);
    for (int32_t r = 0; r < nr; r++) {
      unsigned char* dst = buf.data() + (size_t)r * rowBytes;
      const size_t rowOff = (size_t)(t0 + r) * (size_t)ld;
      if (f->storeF32) {
        uint32_t* o = (uint32_t*)dst;
        if (data_is_f32) {
          const uint32_t* s = (const uint32_t*)data + rowOff;
          for (int32_t m = 0; m < n_members; m++) o[m] = __builtin_bswap32(s[m]);
        } else {
          const double* s = (const double*)data + rowOff;
          for (int32_t m = 0; m < n_members; m++) {
            const float x = (float)s[m];
            uint32_t bits;
            memcpy(&bits, &x, 4);
            o[m] = __builtin_bswap32(bits);
          }
        }
      } else {
        uint64_t* o = (uint64_t*)dst;
        if (data_is_f32) {
          const float* s = (const float*)data + rowOff;
          for (int32_t m = 0; m < n_members; m++) {
            const double x = (double)s[m];
            uint64_t bits;
            memcpy(&bits, &x, 8);
            o[m] = __builtin_bswap64(bits);
          }
        } else {
          const uint64_t* s = (const uint64_t*)data + rowOff;
          for (int32_t m = 0; m < n_members; m++) o[m] = __builtin_bswap64(s[m]);
        }
      }
    }
    const uint64_t off = v.begin + ((uint64_t)(step0 + t0) * M + (uint64_t)member0) * esize;
    if (!writeAll(f->fd, buf.data(), (size_t)nr * rowBytes, off)) {
      setError(std::string("sipnet_io_ensemble_put: write to ") + f->path + " failed: " + strerror(errno));
      return SIPNET_ERR_FILE_OPEN;
    }
  }
  return SIPNET_OK;
}

int sipnet_io_ensemble_close(sipnet_ensemble_file* f) {
  if (!f) return SIPNET_OK;
  int rc = SIPNET_OK;
  if (f->fd >= 0 && close(f->fd) != 0) {
    setError(std::string("sipnet_io_ensemble_close: ") + f->path + ": " + strerror(errno));
    rc = SIPNET_ERR_FILE_OPEN;
  }
  delete f;
  return rc;
}

int sipnet_io_write_ensemble_block(const char* path, int32_t n_steps, int32_t n_members, const int32_t* year,
                                   const int32_t* day, const double* clim, const int32_t* member_ids, const double* planes,
                                   const double* rec, int64_t ld, const char* columns, int32_t store_f32, const char* attrs) {
  if (!planes && !rec) {
    setError("sipnet_io_write_ensemble_block: neither planes nor records");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  if (ld < n_members) {
    setError("sipnet_io_write_ensemble_block: ld < n_members");
    return SIPNET_ERR_BAD_ARGUMENT;
  }
  // which columns: the three planes' names without records; with records the named ones (default: all)
  std::vector<int> cols;
  if (rec) {
    if (columns && columns[0]) {
      std::string s(columns);
      size_t pos = 0;
      while (pos <= s.size()) {
        size_t c = s.find(',', pos);
        if (c == std::string::npos) c = s.size();
        const std::string nm = s.substr(pos, c - pos);
        pos = c + 1;
        if (nm.empty()) continue;
        const int k = sipnet_io_out_column_index(nm.c_str());
        if (k < 0) {
          setError("sipnet_io_write_ensemble_block: unknown .out column " + nm);
          return SIPNET_ERR_BAD_ARGUMENT;
        }
        cols.push_back(k);
      }
    } else {
      for (int k = 0; k < kNumOutColumns; k++) cols.push_back(k);
    }
  } else {
    cols = {sipnet_io_out_column_index("nee"), sipnet_io_out_column_index("gpp"), sipnet_io_out_column_index("evapotranspiration")};
  }
  std::vector<const char*> names;
  for (int k : cols) names.push_back(kOutColumns[k].name);
  sipnet_ensemble_file* f = nullptr;
  int rc = sipnet_io_ensemble_create(path, n_steps, n_members, year, day, clim, member_ids, (int32_t)names.size(), names.data(),
                                     nullptr, store_f32, attrs, &f);
  if (rc) return rc;
  std::vector<double> col;
  for (size_t v = 0; v < cols.size() && rc == SIPNET_OK; v++) {
    const OutColumn& oc = kOutColumns[cols[v]];
    if (!rec) {   // planes[3][n_steps][ld]: NEE, GPP, ET
      rc = sipnet_io_ensemble_put(f, (int32_t)v, 0, n_steps, 0, n_members, planes + (size_t)v * (size_t)n_steps * (size_t)ld, ld, 0);
      continue;
    }
    // rec[n_steps][SIPNET_NREC][ld]: gather the column (sum of two for total wood)
    col.resize((size_t)n_steps * (size_t)n_members);
    for (int32_t t = 0; t < n_steps; t++) {
      const double* r0 = rec + ((size_t)t * SIPNET_NREC + (size_t)oc.rec0) * (size_t)ld;
      double* o = col.data() + (size_t)t * (size_t)n_members;
      if (oc.rec1 >= 0) {
        const double* r1 = rec + ((size_t)t * SIPNET_NREC + (size_t)oc.rec1) * (size_t)ld;
        for (int32_t m = 0; m < n_members; m++) o[m] = r0[m] + r1[m];
      } else {
        memcpy(o, r0, (size_t)n_members * sizeof(double));
      }
    }
    rc = sipnet_io_ensemble_put(f, (int32_t)v, 0, n_steps, 0, n_members, col.data(), n_members, 0);
  }
  const int rc2 = sipnet_io_ensemble_close(f);
  return rc ? rc : rc2;
}

}  // extern "C"
