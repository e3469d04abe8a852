"""Dev tool (build container only): compare oracle/liboracle.so with the real
reference step loop (oracle/_ref/libsipnet_ref.so) on the reference's smoke cases.
Run each case in a subprocess because the reference keeps process-global state."""
import ctypes as C, numpy as np, os, sys, subprocess, json

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMOKE = "/root/reference/tests/smoke"
FLAG_NAMES = ["events","gdd","growthResp","leafWater","litterPool","snow","soilPhenol",
              "waterHResp","nitrogenCycle","anaerobic","flooding","carbonSaturation"]
DEFAULT_FLAGS = dict(events=1,gdd=1,growthResp=0,leafWater=0,litterPool=0,snow=1,soilPhenol=0,
                     waterHResp=1,nitrogenCycle=0,anaerobic=0,flooding=0,carbonSaturation=0)
KEYMAP = {"events":"events","gdd":"gdd","growthresp":"growthResp","leafwater":"leafWater",
          "litterpool":"litterPool","snow":"snow","soilphenol":"soilPhenol","waterhresp":"waterHResp",
          "nitrogencycle":"nitrogenCycle","anaerobic":"anaerobic","flooding":"flooding",
          "carbonsaturation":"carbonSaturation"}

def read_flags(path):
    f = dict(DEFAULT_FLAGS)
    for line in open(path):
        line = line.split("!")[0].strip()
        if not line: continue
        import re
        toks = re.split(r"[ \t=:]+", line)
        k = toks[0].lower().replace("_","").replace("-","")
        if k in KEYMAP and len(toks) > 1:
            f[KEYMAP[k]] = int(toks[1])
    return [f[n] for n in FLAG_NAMES]

EV_TYPES = {"fert":0,"harv":1,"irrig":2,"plant":3,"till":4,"leafon":5,"leafoff":6}
class Ev(C.Structure):
    _fields_ = [("type",C.c_int),("year",C.c_int),("day",C.c_int),("pad",C.c_int),("p",C.c_double*4)]
def read_events(path):
    evs = []
    if not os.path.exists(path): return evs
    for line in open(path):
        t = line.split()
        if len(t) < 3: continue
        e = Ev(); e.year=int(t[0]); e.day=int(t[1]); e.type=EV_TYPES[t[2]]
        for i,v in enumerate(t[3:7]): e.p[i]=float(v)
        evs.append(e)
    return evs

def run_case(case):
    d = os.path.join(SMOKE, case)
    flags = read_flags(os.path.join(d,"sipnet.in"))
    ref = C.CDLL(os.path.join(REPO,"oracle/_ref/libsipnet_ref.so"))
    ora = C.CDLL(os.path.join(REPO,"oracle/liboracle.so"))
    fl = (C.c_int*12)(*flags)
    n = ref.ref_init(fl, os.path.join(d,"sipnet.param").encode(), os.path.join(d,"sipnet.clim").encode(),
                     os.path.join(d,"events.in").encode(), b"/tmp/ref_events_%s.out" % case.encode())
    NP = ref.ref_num_params(); NR = ref.ref_rec_len()
    base = np.zeros(NP); ref.ref_get_base_params(base.ctypes.data_as(C.c_void_p))
    clim = np.zeros((n,11)); yr = np.zeros(n,dtype=np.int32); dy = np.zeros(n,dtype=np.int32)
    ref.ref_get_climate(clim.ctypes.data_as(C.c_void_p), yr.ctypes.data_as(C.c_void_p), dy.ctypes.data_as(C.c_void_p))
    rec_ref = np.zeros((n,NR))
    ref.ref_run_member(base.ctypes.data_as(C.c_void_p), rec_ref.ctypes.data_as(C.c_void_p), None,None,None)
    evs = read_events(os.path.join(d,"events.in"))
    evarr = (Ev*max(1,len(evs)))(*evs)
    rec_or = np.zeros((n,NR))
    ora.sipo_run_member.restype = C.c_int
    st = ora.sipo_run_member(fl, base.ctypes.data_as(C.c_void_p), n, clim.ctypes.data_as(C.c_void_p),
                             yr.ctypes.data_as(C.c_void_p), dy.ctypes.data_as(C.c_void_p), len(evs), evarr,
                             rec_or.ctypes.data_as(C.c_void_p), None,None,None,
                             b"/tmp/ora_events_%s.out" % case.encode(), None)
    diff = np.abs(rec_or-rec_ref)
    print(case, "flags",flags,"steps",n,"status",st,"max|d|",diff.max(), "argmax col", diff.max(0).argmax(),
          "nee max|d|", diff[:,0].max(), "identical", bool((rec_or==rec_ref).all()))
    # events.out parity
    import filecmp
    print("  events.out vs golden:", filecmp.cmp("/tmp/ora_events_%s.out"%case, os.path.join(d,"events.out"), shallow=False))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in ["niwot","russell_1","russell_2","russell_3"]:
            subprocess.run([sys.executable, __file__, c])
