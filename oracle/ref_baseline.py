"""TEST INFRASTRUCTURE -- time the REAL reference (oracle/_ref/psolve, built in place by
oracle/build_ref.sh) on the host cores: bench.py's cpu_baseline leg, kind "reference".

Workload: the reference's own examples/simple material database refined by its own
mesher to 128 x 128 x 64 = 1 048 576 elements (simulation_wave_max_freq_hz = 40,
SURVEY.md s8c item 7), Rayleigh damping, effective stiffness, MPI ranks = host cores.
Two runs of different length are differenced so that meshing/set-up and the quiet
start (the reference skips quiescent elements, quake_util.c:49-68) cancel out of the
per-step time, as SURVEY.md s6 does.

Inputs are data fixtures under tests/golden/ref_inputs (the reference's material
database and source description); the parameter file is written here.
"""
import os
import re
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PSOLVE = os.path.join(HERE, "_ref", "psolve")
INPUTS = os.path.join(ROOT, "tests", "golden", "ref_inputs")

PARAMS = """
region_origin_latitude_deg  = 0.0
region_origin_longitude_deg = 0.0
region_depth_shallow_m      = 0
region_length_east_m        = 1000
region_length_north_m       = 1000
region_depth_deep_m         = 500
region_azimuth_leftface_deg = 0
type_of_damping             = rayleigh
output_mesh                 = 0
source_directory            = sourcefiles
source_directory_output     = out/srctmp
monitor_file                = out/monitor.txt
simulation_wave_max_freq_hz    = {freq}
simulation_start_time_sec      = 0
simulation_end_time_sec        = {end_time}
simulation_delta_time_sec      = {dt}
simulation_node_per_wavelength = 8
simulation_shear_velocity_min  = 3400
simulation_output_rate         = 10000000
the_threshold_damping          = 0.05
the_threshold_Vp_over_Vs       = 3.0
do_damping_statistics          = 0
simulation_displacement_out    = 0
simulation_velocity_out        = 0
use_checkpoint     = 0
checkpointing_rate = 100000000
checkpoint_path    = out/checkpoints
number_output_planes     = 0
output_planes_print_rate = 50
output_planes_directory  = out/planes
domain_surface_corners =
  0.0      0.0
  0.0      1000.0
  1000.0   1000.0
  1000.0   0.0
number_output_stations        = 1
output_stations_print_rate    = 100000
output_stations_directory     = out/stations
output_stations =
500.0 500.0 100.0
softening_factor = 0
use_progressive_meshing = 0
4D_output_file = out/disp.q4d
cvmdb_input_file = simple_case.e
mesh_etree_output_file = out/mesh.e
planes_input_file = planes.in
include_nonlinear_analysis = no
stiffness_calculation_method = effective
print_matrix_k = no
print_station_velocities = no
print_station_accelerations = no
include_buildings = no
mesh_coordinates_for_matlab = no
implement_drm = no
simulation_velocity_profile_freq_hz = 0
use_infinite_qk = no
"""


def available():
    mpi = os.environ.get("HERC_MPI_DIR", "/opt/conda")
    return (os.path.exists(PSOLVE) and os.path.exists(os.path.join(INPUTS, "simple_case.e")) and
            os.path.exists(os.path.join(mpi, "bin", "mpiexec")))


def _run(nranks, steps, freq, dt, timeout):
    mpi = os.environ.get("HERC_MPI_DIR", "/opt/conda")
    run = tempfile.mkdtemp(prefix="herc_refbase_", dir="/tmp")
    try:
        shutil.copy(os.path.join(INPUTS, "simple_case.e"), run)
        shutil.copytree(os.path.join(INPUTS, "sourcefiles"), os.path.join(run, "sourcefiles"))
        for d in ("checkpoints", "planes", "srctmp", "stations"):
            os.makedirs(os.path.join(run, "out", d))
        open(os.path.join(run, "parameters.in"), "w").write(
            PARAMS.format(freq=freq, dt=dt, end_time=repr(steps * dt)))
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(mpi, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        out = subprocess.run([os.path.join(mpi, "bin", "mpiexec"), "-np", str(nranks), PSOLVE, "parameters.in"],
                             cwd=run, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             universal_newlines=True, timeout=timeout)
        if out.returncode != 0:
            raise RuntimeError("psolve failed: " + out.stdout[-500:])
        m = re.search(r"TOTAL SOLVER\s*:\s*([0-9.]+)", out.stdout)
        e = re.search(r"Total elements:\s+(\d+)", out.stdout)
        s = re.search(r"Simulation duration\s*:.*?\n.*?Total number of steps\s*:\s*(\d+)", out.stdout, re.S)
        nsteps = int(s.group(1)) if s else steps
        return float(m.group(1)), int(e.group(1)), nsteps
    finally:
        shutil.rmtree(run, ignore_errors=True)


SOURCE_IN = """
source_is_filtered     = 0
threshold_frequency    = 4.5
number_of_poles        = 14
source_function_type = discrete
number_of_time_windows = 1
time_windows =
0
type_of_source = srfh
number_of_point_sources = {n}
domain_surface_corners =
  0.0      0.0
  0.0      1000.0
  1000.0   1000.0
  1000.0   0.0
"""


def _grid_sources(run, nx=4, ny=4, nz=2):
    """Replace the single point source by a grid of nx x ny x nz double couples with a short ramp as
    slip function (srfh input files, quakesource.c:2313-2390), so that every element of the box is
    active after a few dozen steps (the reference skips quiescent elements, quake_util.c:49-68; activity
    spreads one element per step)."""
    sd = os.path.join(run, "sourcefiles")
    pts = [((i + 0.5) * 1000.0 / nx + 1.7, (j + 0.5) * 1000.0 / ny + 2.3, (k + 0.5) * 500.0 / nz + 1.1)
           for k in range(nz) for j in range(ny) for i in range(nx)]
    n = len(pts)
    open(os.path.join(sd, "source.in"), "w").write(SOURCE_IN.format(n=n))
    open(os.path.join(sd, "coords.in"), "w").write("".join("%.3f %.3f %.3f\n" % p for p in pts))
    for name, val in (("area.in", "30866000.0"), ("strike.in", "0.0"), ("dip.in", "90.0"), ("rake.in", "0.0"),
                      ("slip.in", "1")):
        open(os.path.join(sd, name), "w").write((val + "\n") * n)
    ramp = "".join("%.6f\n" % min(1.0, t / 20.0) for t in range(64))
    open(os.path.join(sd, "slipfunction.in"), "w").write(("64\n0.0\n0.001\n" + ramp) * n)
    return n


def measure_box(nranks, freq=80.0, dt=0.00025, steps=101, timeout=420):
    """ONE run of the reference on the SURVEY s8d box for BASELINE config 2 (f = 80 Hz refines the
    examples/simple material to 256 x 256 x 128 = 8 388 608 elements; f = 40 Hz: 1 048 576), a grid of
    32 point sources, and the wall clock the solver itself prints every 50 steps (solver_update_status,
    psolve.c:3810-3838: "WC="): element-updates/s over steps 50..100, when every element is active."""
    mpi = os.environ.get("HERC_MPI_DIR", "/opt/conda")
    run = tempfile.mkdtemp(prefix="herc_refbase_", dir="/tmp")
    try:
        shutil.copy(os.path.join(INPUTS, "simple_case.e"), run)
        shutil.copytree(os.path.join(INPUTS, "sourcefiles"), os.path.join(run, "sourcefiles"))
        nsrc = _grid_sources(run)
        for d in ("checkpoints", "planes", "srctmp", "stations"):
            os.makedirs(os.path.join(run, "out", d))
        open(os.path.join(run, "parameters.in"), "w").write(
            PARAMS.format(freq=freq, dt=dt, end_time=repr(steps * dt)))
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(mpi, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        out = subprocess.run([os.path.join(mpi, "bin", "mpiexec"), "-np", str(nranks), PSOLVE, "parameters.in"],
                             cwd=run, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             universal_newlines=True, timeout=timeout)
        if out.returncode != 0:
            raise RuntimeError("psolve failed: " + out.stdout[-500:])
        wc = [float(v) for v in re.findall(r"WC=\s*([0-9.]+)", out.stdout)]
        e = re.search(r"Total elements:\s+(\d+)", out.stdout)
        if len(wc) < 2 or not e:
            raise RuntimeError("no wall-clock lines in the solver's output: " + out.stdout[-300:])
        E = int(e.group(1))
        per_step = (wc[-1] - wc[-2]) / 50.0
        return {"value": E / per_step, "elements": E, "ranks": nranks, "s_per_step": per_step,
                "steps": (steps - 51, steps - 1), "sources": nsrc, "wall_clock": wc}
    finally:
        shutil.rmtree(run, ignore_errors=True)


def measure(nranks, steps_short=150, steps_long=450, freq=40.0, dt=0.0005, timeout=600):
    """-> dict(value element-updates/s, elements, ranks, ...) from the solver's own
    'TOTAL SOLVER' timer (psolve.c:6065-6081), long run minus short run."""
    t1, E, n1 = _run(nranks, steps_short, freq, dt, timeout)
    t2, E2, n2 = _run(nranks, steps_long, freq, dt, timeout)
    assert E == E2 and n2 > n1 and t2 > t1
    per_step = (t2 - t1) / (n2 - n1)
    return {"value": E / per_step, "elements": E, "ranks": nranks, "s_per_step": per_step,
            "steps": (n1, n2), "solver_s": (t1, t2)}
