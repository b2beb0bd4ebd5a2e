#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/b6; mkdir -p $O
( cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -o /tmp/march_twostep $GRAFT_REPO_ROOT/profiles/micro/march_twostep.hip && timeout 600 /tmp/march_twostep > $GRAFT_REPO_ROOT/$O/march_twostep.txt 2>&1 )
cat $O/march_twostep.txt
