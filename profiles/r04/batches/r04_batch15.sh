#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/b15; mkdir -p $O
( cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -o /tmp/march_stencil $GRAFT_REPO_ROOT/profiles/micro/march_stencil.hip && timeout 600 /tmp/march_stencil 2>&1 | head -6 > $GRAFT_REPO_ROOT/$O/march_prefetch2.txt )
cat $O/march_prefetch2.txt
