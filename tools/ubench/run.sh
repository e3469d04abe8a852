#!/bin/bash
cd "$GRAFT_REPO_ROOT/tools/ubench" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w issue_cost.hip -o /tmp/issue_cost || exit 1
for i in $(seq 0 14); do timeout 20 /tmp/issue_cost $i || echo "test $i: timeout/fail rc=$?"; done
