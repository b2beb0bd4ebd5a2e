"""INTEGRATION.md's psolve.c stub (examples/psolve_hq_stub.inc) is compiled in the place it is meant
for -- behind the reference's own psolve.c, included where it lies under /root/reference (nothing is
copied) -- with gcc -fsyntax-only and the switches of oracle/build_ref.sh, so every reference name it
uses (Global.myMesh, dnode_t.lanid, messenger_t.mapping, comm_solver, solver_abort ...) is checked
against the real headers.  Skipped where the reference tree is absent (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HERC_REFERENCE", "/root/reference")
MPI = os.environ.get("HERC_MPI_DIR", "/opt/conda")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "quake", "forward")) or
                    not os.path.exists(os.path.join(MPI, "include", "mpi.h")),
                    reason="reference tree / MPI headers not present")
def test_stub_compiles_against_the_reference_headers(tmp_path):
    tu = tmp_path / "psolve_with_stub.c"
    tu.write_text('#include "%s"\n#include "%s"\n'
                  'void hq_stub_is_used(void) { hq_attach(8); hq_steps(0, 1); hq_refresh_host(0); }\n'
                  % (os.path.join(REF, "quake", "forward", "psolve.c"), os.path.join(ROOT, "examples", "psolve_hq_stub.inc")))
    cmd = ["gcc", "-fsyntax-only", "-std=gnu99", "-Wall", "-Wno-unused", "-Wno-format-truncation",
           "-D_FILE_OFFSET_BITS=64", "-D_LARGEFILE_SOURCE", "-DHALFSPACE", "-DBOUNDARY", "-DUSECVMDB", "-DSCEC",
           "-DPROCPERNODE=4000", "-I", os.path.join(MPI, "include"), "-I", os.path.join(REF, "etree"),
           "-I", os.path.join(REF, "quake", "cvm"), "-I", os.path.join(REF, "octor"),
           "-I", os.path.join(REF, "quake", "forward"), "-I", os.path.join(ROOT, "include"), str(tu)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    ours = [l for l in out.stdout.splitlines() if "psolve_hq_stub.inc" in l and ("error" in l or "warning" in l)]
    assert out.returncode == 0 and not ours, "\n".join(ours) or out.stdout[-3000:]
