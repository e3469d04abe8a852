#!/bin/bash
cd "$GRAFT_REPO_ROOT/tools/ubench" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w pingpong.hip -o /tmp/pingpong || exit 1
for i in $(seq 0 9); do timeout 20 /tmp/pingpong $i || echo "test $i: timeout/fail rc=$?"; done
