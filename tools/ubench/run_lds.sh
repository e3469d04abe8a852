#!/bin/bash
cd "$GRAFT_REPO_ROOT/tools/ubench" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w lds_overlap.hip -o /tmp/lds_overlap || exit 1
for i in $(seq 0 4); do timeout 20 /tmp/lds_overlap $i || echo "test $i: timeout/fail rc=$?"; done
