#!/bin/bash
# TEST INFRASTRUCTURE (not product).  Builds the *real* reference solver
# (CMU-Quake/hercules `psolve`) from its own sources, in place, into
# oracle/_ref/psolve.  Nothing from /root/reference is copied into the repo:
# the compiler reads the sources where they lie and only the binary lands in
# oracle/_ref/ (git-ignored, travels to the GPU box with gpurun).
#
# We do NOT run the reference's Makefiles: this is a direct gcc invocation
# over the file lists in quake/forward/Makefile:53-54, octor/, etree/ and
# quake/cvm/cvm.c, with the reference's default switches
# (-DHALFSPACE -DBOUNDARY, quake/forward/Makefile:19; -DUSECVMDB, :36).
# MPI comes from the image's MPICH (/opt/conda), which the reference needs
# (psolve.h includes mpi.h).  -fno-stack-protector: the reference overruns a
# stack buffer at exit under Ubuntu's default hardening.
set -euo pipefail
REF=${HERC_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
MPI=${HERC_MPI_DIR:-/opt/conda}
if [ ! -d "$REF/quake/forward" ]; then
    echo "build_ref: $REF absent - keeping prebuilt oracle/_ref (if any)"; exit 0
fi
if [ ! -f "$MPI/include/mpi.h" ]; then
    echo "build_ref: no MPI in $MPI - reference unbuildable here"; exit 0
fi
mkdir -p "$OUT/obj"
CC=${CC:-gcc}
CFLAGS="-O2 -g -fno-stack-protector -w -std=gnu99 -D_FILE_OFFSET_BITS=64 -D_LARGEFILE_SOURCE"
INC="-I$MPI/include -I$REF/etree -I$REF/quake/cvm -I$REF/octor -I$REF/quake/forward"
DEFS="-DHALFSPACE -DBOUNDARY -DUSECVMDB -DSCEC -DPROCPERNODE=4000"
FWD="psolve nrutila quakesource geometrics nonlinear commutil util output io_planes io_checkpoint stiffness damping quake_util timers buildings meshformatlab drm"
ETREE="btree buffer code dlink etree schema wrapper xplatform"
objs=""
objs32=""   # the same program with -DSINGLE_PRECISION_SOLVER (psolve.h:60-64: solver_float = float): oracle/_ref/psolve_f32
for f in $FWD;   do $CC $CFLAGS $DEFS $INC -c "$REF/quake/forward/$f.c" -o "$OUT/obj/fwd_$f.o" & objs="$objs $OUT/obj/fwd_$f.o"; done
for f in $FWD;   do $CC $CFLAGS $DEFS -DSINGLE_PRECISION_SOLVER $INC -c "$REF/quake/forward/$f.c" -o "$OUT/obj/f32_$f.o" & objs32="$objs32 $OUT/obj/f32_$f.o"; done
eobjs=""   # etree goes into an archive, as in etree/Makefile:15-17 (wrapper.o is never pulled)
for f in $ETREE; do $CC $CFLAGS $DEFS $INC -c "$REF/etree/$f.c"        -o "$OUT/obj/et_$f.o"  & eobjs="$eobjs $OUT/obj/et_$f.o";  done
$CC $CFLAGS $DEFS $INC -c "$REF/octor/octor.c"   -o "$OUT/obj/octor.o" & objs="$objs $OUT/obj/octor.o"; objs32="$objs32 $OUT/obj/octor.o"
$CC $CFLAGS $DEFS $INC -c "$REF/quake/cvm/cvm.c" -o "$OUT/obj/cvm.o"   & objs="$objs $OUT/obj/cvm.o"; objs32="$objs32 $OUT/obj/cvm.o"
wait
ar rcs "$OUT/obj/libetree.a" $eobjs
$CC -o "$OUT/psolve" $objs "$OUT/obj/libetree.a" -L"$MPI/lib" -Wl,-rpath,"$MPI/lib" -lmpi -lm
$CC -o "$OUT/psolve_f32" $objs32 "$OUT/obj/libetree.a" -L"$MPI/lib" -Wl,-rpath,"$MPI/lib" -lmpi -lm
# fixture helper: a layered CVM database writer on top of the reference's etree + cvm libraries
$CC $CFLAGS $INC -o "$OUT/make_cvm" "$HERE/make_cvm.c" "$OUT/obj/cvm.o" "$OUT/obj/libetree.a" -lm
rm -rf "$OUT/obj"
echo "build_ref: built $OUT/psolve $OUT/psolve_f32 $OUT/make_cvm"
