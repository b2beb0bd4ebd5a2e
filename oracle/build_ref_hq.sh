#!/bin/bash
# TEST INFRASTRUCTURE (not product).  Builds oracle/_ref/psolve_hq: the REAL reference program
# (CMU-Quake/hercules psolve: its main(), mesher, solver_init, source, stations, checkpoints) with the
# physics + communication block of solver_run() (psolve.c:4286-4316) running on libhq_solver.so -- the
# drop-in boundary of INTEGRATION.md exercised by the reference itself.
#
# As oracle/build_ref.sh: a direct gcc invocation over the reference's sources where they lie; only
# quake/forward/psolve.c is first passed through oracle/patch_psolve_hq.py (the stub's four edits) into a
# scratch directory under /tmp, compiled from there and deleted.  Nothing from /root/reference is copied
# into the repo; only the binary lands in oracle/_ref/ (git-ignored, travels to the GPU box with gpurun).
set -euo pipefail
REF=${HERC_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
OUT=$HERE/_ref
MPI=${HERC_MPI_DIR:-/opt/conda}
LIB=$ROOT/hercules_amd/csrc
if [ ! -d "$REF/quake/forward" ]; then
    echo "build_ref_hq: $REF absent - keeping prebuilt oracle/_ref/psolve_hq (if any)"; exit 0
fi
if [ ! -f "$MPI/include/mpi.h" ]; then
    echo "build_ref_hq: no MPI in $MPI - reference unbuildable here"; exit 0
fi
if [ ! -f "$LIB/libhq_solver.so" ]; then
    echo "build_ref_hq: $LIB/libhq_solver.so not built yet (python -m hercules_amd.build)"; exit 1
fi
TMP=$(mktemp -d /tmp/psolve_hq.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT" "$TMP/obj"
python3 "$HERE/patch_psolve_hq.py" "$REF/quake/forward/psolve.c" "$ROOT/examples/psolve_hq_stub.inc" > "$TMP/psolve_hq.c"
CC=${CC:-gcc}
CFLAGS="-O2 -g -fno-stack-protector -w -std=gnu99 -D_FILE_OFFSET_BITS=64 -D_LARGEFILE_SOURCE"
INC="-I$MPI/include -I$REF/etree -I$REF/quake/cvm -I$REF/octor -I$REF/quake/forward -I$ROOT/include"
DEFS="-DHALFSPACE -DBOUNDARY -DUSECVMDB -DSCEC -DPROCPERNODE=4000"
FWD="nrutila quakesource geometrics nonlinear commutil util output io_planes io_checkpoint stiffness damping quake_util timers buildings meshformatlab drm"
ETREE="btree buffer code dlink etree schema wrapper xplatform"
objs="$TMP/obj/psolve_hq.o"
$CC $CFLAGS $DEFS $INC -c "$TMP/psolve_hq.c" -o "$TMP/obj/psolve_hq.o" &
for f in $FWD;   do $CC $CFLAGS $DEFS $INC -c "$REF/quake/forward/$f.c" -o "$TMP/obj/fwd_$f.o" & objs="$objs $TMP/obj/fwd_$f.o"; done
eobjs=""
for f in $ETREE; do $CC $CFLAGS $DEFS $INC -c "$REF/etree/$f.c"        -o "$TMP/obj/et_$f.o"  & eobjs="$eobjs $TMP/obj/et_$f.o";  done
$CC $CFLAGS $DEFS $INC -c "$REF/octor/octor.c"   -o "$TMP/obj/octor.o" & objs="$objs $TMP/obj/octor.o"
$CC $CFLAGS $DEFS $INC -c "$REF/quake/cvm/cvm.c" -o "$TMP/obj/cvm.o"   & objs="$objs $TMP/obj/cvm.o"
wait
ar rcs "$TMP/obj/libetree.a" $eobjs
# libhq_solver.so (hipcc) needs the system's libstdc++; $MPI/lib (conda) carries an older one, so that directory is
# neither a -L nor an rpath here: libmpi by path, the system libstdc++ named first, and at run time
# LD_LIBRARY_PATH = <system lib dir>:$MPI/lib (tests/test_gpu_reference_link.py)
SYSLIB=$(dirname "$($CC -print-file-name=libstdc++.so.6)")
$CC -o "$OUT/psolve_hq" $objs "$TMP/obj/libetree.a" "$SYSLIB/libstdc++.so.6" -L"$LIB" -lhq_solver \
    "$MPI/lib/libmpi.so" -Wl,-rpath-link,"$MPI/lib" -Wl,-rpath,'$ORIGIN/../../hercules_amd/csrc' -Wl,-rpath,/opt/rocm/lib -lm
# the same program in the reference's single precision (psolve.h:60-64) on libhq_solver_f32.so: only the files that see
# solver_float are compiled again (everything under quake/forward includes psolve.h)
objs32="$TMP/obj/f32_psolve_hq.o"
D32="-DSINGLE_PRECISION_SOLVER -DHQ_SINGLE_PRECISION_SOLVER"
$CC $CFLAGS $DEFS $D32 $INC -c "$TMP/psolve_hq.c" -o "$TMP/obj/f32_psolve_hq.o" &
for f in $FWD;   do $CC $CFLAGS $DEFS $D32 $INC -c "$REF/quake/forward/$f.c" -o "$TMP/obj/f32_$f.o" & objs32="$objs32 $TMP/obj/f32_$f.o"; done
wait
if [ -f "$LIB/libhq_solver_f32.so" ]; then
    $CC -o "$OUT/psolve_hq_f32" $objs32 "$TMP/obj/octor.o" "$TMP/obj/cvm.o" "$TMP/obj/libetree.a" "$SYSLIB/libstdc++.so.6" -L"$LIB" -lhq_solver_f32 \
        "$MPI/lib/libmpi.so" -Wl,-rpath-link,"$MPI/lib" -Wl,-rpath,'$ORIGIN/../../hercules_amd/csrc' -Wl,-rpath,/opt/rocm/lib -lm
    echo "build_ref_hq: built $OUT/psolve_hq_f32"
fi
echo "build_ref_hq: built $OUT/psolve_hq"
