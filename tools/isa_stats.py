#!/usr/bin/env python3
"""Static instruction mix of the step loop of stepFastKernel<double|float> (dev tool)."""
import re, subprocess, sys
from collections import Counter
flags = sys.argv[1:] or []
subprocess.check_call(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-fPIC","-ffp-contract=fast","-S","--cuda-device-only","-w",
                       "/root/repo/sipnet_amd/csrc/step_fast.hip","-o","/tmp/sf.s"]+flags, stderr=subprocess.DEVNULL)
lines = open('/tmp/sf.s').read().split('\n')
for ty in ("Id","If"):
    i = [k for k,l in enumerate(lines) if re.match(r'^_ZN.*stepFastKernel%s.*:' % ty, l)][0]
    j = i
    while 's_endpgm' not in lines[j]: j+=1
    body = lines[i:j]
    lab = {}
    for n,l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: lab[m.group(1)] = n
    spans=[]
    for n,l in enumerate(body):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
        if m:
            t = m.group(1) or m.group(2)
            if lab.get(t, 10**9) < n: spans.append((n-lab[t], lab[t], n))
    spans.sort(reverse=True)
    _, a, b = spans[1]
    loop = body[a:b+1]
    if ty=="Id": open('/tmp/loop.s','w').write('\n'.join(loop))
    ins = [x.strip().split()[0] for x in loop if x.startswith('\t') and not x.strip().startswith('.') and not x.strip().startswith(';')]
    c = Counter(ins)
    valu = sum(v for k,v in c.items() if k.startswith('v_'))
    salu = sum(v for k,v in c.items() if k.startswith('s_'))
    meta = [l for l in lines[j:j+400] if 'vgpr_count' in l or 'sgpr_count' in l or 'agpr_count' in l][:3]
    print(ty, "loop instrs", len(ins), "VALU", valu, "SALU", salu, "| mov_b64", c.get('v_mov_b64_e32',0), "mov_b32", c.get('v_mov_b32_e32',0),
          "accrd", c.get('v_accvgpr_read_b32',0), "accwr", c.get('v_accvgpr_write_b32',0), "readlane", c.get('v_readlane_b32',0), "writelane", c.get('v_writelane_b32',0),
          "cndmask", c.get('v_cndmask_b32_e32',0)+c.get('v_cndmask_b32_e64',0), "ds", sum(v for k,v in c.items() if k.startswith('ds_')),
          "branches", sum(v for k,v in c.items() if 'branch' in k))
    print("   ", [m.strip() for m in meta])
