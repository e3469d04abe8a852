"""Model flags and the `sipnet.in` configuration file.

Flag order and defaults: /root/reference/src/common/context.h:46-57 and
context.c:35-53.  `sipnet.in` syntax: /root/reference/src/sipnet/frontend.c:35-128
(`KEY[ =:\\t]VALUE`, `!` starts a comment, names are matched after dropping
`_`/`-` and lower-casing, context.c:76-90).
"""
import re

from ._lib import lib, NPARAMS

FLAG_NAMES = ["events", "gdd", "growthResp", "leafWater", "litterPool", "snow",
              "soilPhenol", "waterHResp", "nitrogenCycle", "anaerobic", "flooding",
              "carbonSaturation"]
DEFAULT_FLAGS = dict(events=1, gdd=1, growthResp=0, leafWater=0, litterPool=0, snow=1,
                     soilPhenol=0, waterHResp=1, nitrogenCycle=0, anaerobic=0, flooding=0,
                     carbonSaturation=0)
_IO_DEFAULTS = dict(doMainOutput=1, doSingleOutputs=0, dumpConfig=0, printHeader=1, quiet=0,
                    filePrefix="sipnet", eventsPrefix="events", restartIn="", restartOut="",
                    debugLogPrefix="")
# aliases the reference accepts (cli.c:48-50, context metadata keys)
_ALIASES = {"filename": "filePrefix", "doSingleOutput": "doSingleOutputs"}


def _key(name):
    return re.sub(r"[-_]", "", name).lower()


_KEYMAP = {_key(n): n for n in list(DEFAULT_FLAGS) + list(_IO_DEFAULTS)}
_KEYMAP.update({_key(k): v for k, v in _ALIASES.items()})


def flags_from(**overrides):
    """List of the 12 model flags in C-ABI order, defaults plus overrides."""
    f = dict(DEFAULT_FLAGS)
    for k, v in overrides.items():
        if k not in f:
            raise KeyError(k)
        f[k] = int(v)
    return [f[n] for n in FLAG_NAMES]


def read_config(path):
    """Parse a `sipnet.in` file -> dict of every context value (defaults filled in)."""
    cfg = dict(DEFAULT_FLAGS)
    cfg.update(_IO_DEFAULTS)
    with open(path) as fh:
        for line in fh:
            line = line.split("!")[0].strip()
            if not line:
                continue
            toks = [t for t in re.split(r"[ \t=:]+", line) if t]
            name = _KEYMAP.get(_key(toks[0]))
            if name is None:
                if _key(toks[0]) == "runtype":
                    if len(toks) > 1 and toks[1].lower() != "standard":
                        raise ValueError("RUNTYPE must be 'standard'")
                continue  # unknown keys are ignored (frontend.c:78-82)
            if len(toks) < 2:
                raise ValueError(f"No value given for input item {toks[0]}")
            if isinstance(cfg[name], int):
                cfg[name] = int(toks[1], 0)
            else:
                cfg[name] = "" if toks[1] == "none" else toks[1]
    return cfg


PARAM_NAMES = None


def _load_names():
    global PARAM_NAMES
    if PARAM_NAMES is None:
        L = lib()
        PARAM_NAMES = [L.sipnet_param_name(i).decode() for i in range(NPARAMS)]
    return PARAM_NAMES


def param_index(name):
    """Index of a `.param` name in the raw parameter vector (-1 if unknown)."""
    return lib().sipnet_param_index(name.encode())
