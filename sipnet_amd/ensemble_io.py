"""Ensemble output block: every member's outputs in ONE self-describing file.

The reference writes one `<prefix>.out` text file per process, so a 10 k-member PEcAn ensemble
means 10 k directories of text that `model2netcdf.SIPNET` parses again.  This writes the
batch's planes / full records as one NetCDF-3 (classic, 64-bit offset) file with dimensions
(time, member): variables carry SIPNET's own column names (sipnet.c:434-452) and units, the
time axis carries year / day-of-year / hour / step length exactly as the `.clim` rows had them.
scipy's pure-Python NetCDF-3 writer is used (no C library needed); the CLI's text files stay
the drop-in path, this is the bulk path.
"""
import numpy as np

# `.out` column -> (record index, units); names as printed by outputHeader (sipnet.c:434-452)
OUT_COLUMNS = {
    "plantWoodC": ((14, 26), "g C m-2"),   # printed as total wood = plantWoodC + accounting delta
    "plantLeafC": (15, "g C m-2"), "woodCreation": (11, "g C m-2 step-1"),
    "soil": (16, "g C m-2"), "coarseRootC": (20, "g C m-2"), "fineRootC": (21, "g C m-2"),
    "litter": (18, "g C m-2"), "soilWater": (17, "cm"), "soilWetnessFrac": (12, "1"),
    "snow": (19, "cm water equiv."), "npp": (4, "g C m-2 step-1"), "nee": (0, "g C m-2 step-1"),
    "cumNEE": (3, "g C m-2"), "gpp": (1, "g C m-2 step-1"), "rAboveground": (5, "g C m-2 step-1"),
    "rSoil": (6, "g C m-2 step-1"), "rRoot": (7, "g C m-2 step-1"), "ra": (8, "g C m-2 step-1"),
    "rh": (9, "g C m-2 step-1"), "rtot": (10, "g C m-2 step-1"),
    "evapotranspiration": (2, "cm step-1"), "fluxestranspiration": (13, "cm day-1"),
    "minN": (22, "g N m-2"), "soilOrgN": (23, "g N m-2"), "litterN": (24, "g N m-2"),
    "plantStorageN": (25, "g N m-2"),
    "n2o": (27, "g N m-2 step-1"), "nLeaching": (28, "g N m-2 step-1"),
    "nFixation": (29, "g N m-2 step-1"), "nUptake": (30, "g N m-2 step-1"),
    "ch4": (31, "g C m-2 step-1"), "nppStorage": (26, "g C m-2"),
}
PLANE_NAMES = ("nee", "gpp", "evapotranspiration")


def write_ensemble_netcdf(path, clim, planes=None, rec=None, columns=None, member_ids=None,
                          attrs=None, dtype="f8"):
    """planes[3][T][M] (NEE, GPP, ET) and/or rec[T][>=36][M] (full records; `columns` selects
    names from OUT_COLUMNS, default all) -> NetCDF-3 file.  Arrays may be numpy or torch."""
    from scipy.io import netcdf_file
    to_np = lambda x: x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
    T = clim.n_steps
    data = {}
    if rec is not None:
        rec = to_np(rec)
        assert rec.shape[0] == T and rec.shape[1] >= 36
        for name in (columns or OUT_COLUMNS):
            idx, units = OUT_COLUMNS[name]
            col = rec[:, idx[0], :] + rec[:, idx[1], :] if isinstance(idx, tuple) else rec[:, idx, :]
            data[name] = (col, units)
    if planes is not None:
        planes = to_np(planes)
        assert planes.shape[0] == 3 and planes.shape[1] == T
        for k, name in enumerate(PLANE_NAMES):
            data.setdefault(name, (planes[k], OUT_COLUMNS[name][1]))
    assert data, "nothing to write"
    M = next(iter(data.values()))[0].shape[1]
    with netcdf_file(str(path), "w", version=2) as f:
        f.title = "SIPNET ensemble outputs (sipnet_amd)"
        f.model_version = "2.1.0"
        for k, v in (attrs or {}).items():
            setattr(f, k, v)
        f.createDimension("time", T)
        f.createDimension("member", M)
        for name, arr, typ, units in (("year", clim.year, "i4", "year"), ("day", clim.day, "i4", "day of year"),
                                      ("hour", clim.data[:, 10], "f8", "hour of day at step start"),
                                      ("length", clim.data[:, 0], "f8", "days")):
            v = f.createVariable(name, typ, ("time",))
            v[:] = arr
            v.units = units
        v = f.createVariable("member", "i4", ("member",))
        v[:] = np.arange(M) if member_ids is None else np.asarray(member_ids)
        for name, (arr, units) in data.items():
            v = f.createVariable(name, dtype, ("time", "member"))
            v[:] = arr.astype(dtype)
            v.units = units


def read_ensemble_netcdf(path):
    """-> dict name -> array (copied out of the file)"""
    from scipy.io import netcdf_file
    with netcdf_file(str(path), "r", mmap=False) as f:
        return {k: np.array(v[:]).astype(v[:].dtype.newbyteorder("=")) for k, v in f.variables.items()}
