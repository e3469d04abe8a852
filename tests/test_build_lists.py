"""Book-keeping that nothing else checks (no GPU, no compiler): every device source under sipnet_amd/csrc is in the
Makefile's build, every file the step kernels are made of is in the hash that guards the PMC profiles
(sipnet_amd._lib.kernel_source_sha16: a counter collected on older kernel sources must not be quoted under a fresh kernel
time), and the experimental-variant builder links every object the engine refers to."""
import os
import re

from tests import helpers

CSRC = os.path.join(helpers.REPO, "sipnet_amd", "csrc")


def test_every_source_is_built_and_every_step_kernel_file_is_hashed():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    srcs = re.search(r"^SRCS := (.*)$", mk, re.M).group(1)
    objs = re.search(r"^OBJS := (.*)$", mk, re.M).group(1)
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") or (f.endswith(".cpp") and f != "sipnet_cli.cpp"):
            assert "$(HERE)" + f in srcs, f + " is not in the Makefile's SRCS"
            assert "$(HERE)" + f.rsplit(".", 1)[0] + ".o" in objs, f + " has no object in the Makefile's OBJS"
    hdrs = re.search(r"^HDRS := (.*?)\n\n", mk, re.M | re.S).group(1)
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".inc") and not f.startswith("coop_"):      # (coop_*.inc: a wildcard)
            assert f in hdrs, f + " is not a prerequisite of the objects"
    lib = open(os.path.join(helpers.REPO, "sipnet_amd", "_lib.py")).read()
    hashed = re.search(r"def kernel_source_sha16\(\):.*?for name in \((.*?)\):", lib, re.S).group(1)
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".inc") or f in ("step_kernel.hip", "step_fast.hip", "step_coop.hip", "step_kernel.h", "fast_math.h"):
            assert '"%s"' % f in hashed, f + " is not in kernel_source_sha16's list"
    # (step_coop_bounded.hip / step_coop_sums.hip / step_fast_sums.hip are two-line wrappers around hashed sources)
    for f in ("step_coop_bounded.hip", "step_coop_sums.hip", "step_fast_sums.hip"):
        body = [l for l in open(os.path.join(CSRC, f)).read().splitlines() if l.strip() and not l.startswith("//")]
        assert len(body) == 2 and body[0].startswith("#define ") and body[1].startswith('#include "step_'), (f, body)


def test_the_variant_builder_links_every_object():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    objs = {os.path.basename(o) for o in re.search(r"^OBJS := (.*)$", mk, re.M).group(1).replace("$(HERE)", "").split()}
    vb = open(os.path.join(helpers.REPO, "tools", "build_variants.py")).read()
    fast = set(re.search(r"^FAST_SRCS = \[(.*?)\]", vb, re.M).group(1).replace('"', "").replace(" ", "").split(","))
    others = set(re.search(r"^OTHER_OBJS = \[(.*?)\]", vb, re.M).group(1).replace('"', "").replace(" ", "").split(","))
    assert {s.replace(".hip", ".o") for s in fast} | others == objs
