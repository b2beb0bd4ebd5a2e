#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not product; build container only).  Writes, to stdout, the translation unit of
`psolve_hq`: the reference's quake/forward/psolve.c with INTEGRATION.md's stub applied -- the four edits a
Hercules maintainer makes to run solver_run()'s physics + communication block (psolve.c:4286-4316) on
libhq_solver.so.  The output goes to a scratch directory under /tmp (oracle/build_ref_hq.sh), is compiled
there and deleted; no reference source enters the repository.

    patch_psolve_hq.py <reference>/quake/forward/psolve.c <repo>/examples/psolve_hq_stub.inc

Edits (each anchor must occur exactly once, else the script fails -- a changed reference is not patched
blindly):
  1. the stub (hq_attach / hq_steps / hq_refresh_host) is included in front of solver_run();
  2. main(): hq_attach() behind output_init(), i.e. after solver_init(), source_init(), stiffness_init();
  3. solver_run(): behind the tm1/tm2 pointer swap the host arrays are refreshed from the device AT THE
     REFERENCE'S OWN OUTPUT CADENCE (hq_refresh_for_outputs: the whole field in the post-swap view where a
     checkpoint, the 4D output or a plane is due, the stations' 8 nodes each where only stations are due,
     nothing otherwise), so that checkpoints, stations, planes and the 4D output read what they always read;
  4. solver_run(): the block from solver_nonlinear_state() to solver_send_displacement_dangling() becomes
     hq_steps( step, n ) -- n = hq_batch_length( step ), the steps up to the next due output, enqueued in one
     go, the loop counter advanced by n - 1 -- between Timer_Start / Timer_Stop of the timers
     solver_run_collect_timers() reduces; behind the loop the device-side timing split is printed
     (hq_print_timing) and the context destroyed.
"""
import sys


def once(text, anchor):
    n = text.count(anchor)
    if n != 1:
        sys.exit("patch_psolve_hq: anchor occurs %d times, expected once: %r" % (n, anchor[:60]))
    return text.index(anchor)


def main():
    src, stub = sys.argv[1], sys.argv[2]
    t = open(src).read()
    # 1
    a = "static void solver_run()\n{"
    i = once(t, a)
    t = t[:i] + '#include "%s"\n\n' % stub + t[i:]
    # 2
    a = "    output_init (Param.parameters_input_file, &Param.theOutputParameters);\n"
    i = once(t, a) + len(a)
    t = t[:i] + "    hq_attach( 8 );\n" + t[i:]
    # 3
    a = "        Global.mySolver->tm1 = tmpvector;\n"
    i = once(t, a) + len(a)
    t = t[:i] + "        hq_refresh_for_outputs( step, startingStep );\n" + t[i:]
    # 4
    a = '        Timer_Start( "Compute Physics" );\n        solver_nonlinear_state('
    b = '        solver_send_displacement_dangling( Global.mySolver );\n        Timer_Stop( "Communication" );\n'
    i, j = once(t, a), once(t, b) + len(b)
    if not i < j:
        sys.exit("patch_psolve_hq: the physics block's anchors are out of order")
    # solver_run_collect_timers() reduces the timers of the replaced phases: they must exist
    names = ["Compute Physics", "Communication", "Compute addforces s", "Compute addforces e", "Damping addforce",
             "1st schedule send data (contribution)", "1st compute adjust (distribution)",
             "2nd schedule send data (contribution)", "Compute new displacement", "3rd schedule send data (sharing)",
             "2nd compute adjust (assignment)", "4th schadule send data (sharing)"]
    for n in names:
        if '"%s"' % n not in t:
            sys.exit("patch_psolve_hq: the reference has no timer %r any more" % n)
    block = "".join('        Timer_Start( "%s" );\n' % n for n in names) + \
        "        { int hq_n = hq_batch_length( step ); hq_steps( step, hq_n ); step += hq_n - 1; }\n" + \
        "".join('        Timer_Stop( "%s" );\n' % n for n in reversed(names))
    t = t[:i] + block + t[j:]
    a = "    solver_drm_close();\n    solver_output_wavefield_close();\n    solver_run_collect_timers();\n"
    i = once(t, a)
    t = t[:i] + "    hq_sync( theHq );\n    hq_print_timing();\n    hq_destroy( theHq );\n" + t[i:]
    sys.stdout.write(t)


if __name__ == "__main__":
    main()
