"""ctypes binding of include/hq_host.h (C host side: uniform layered boxes,
point source, stations, solver_run)."""
import ctypes
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
# HQ_HOST_LIB: an instrumented build of hq_host.c (tests/sanitize_host.sh), never a fallback
_LIBPATH = os.environ.get("HQ_HOST_LIB") or os.path.join(_HERE, "csrc", "libhq_host.so")

DAMPING = {"none": 0, "rayleigh": 1, "mass": 2}
EXPORTS = ["hqh_box_create", "hqh_box_destroy", "hqh_box_get_info", "hqh_box_desc", "hqh_box_lnid",
           "hqh_box_node_ijk", "hqh_box_etable", "hqh_box_ntable", "hqh_box_owner", "hqh_box_material", "hqh_ntable_to_float",
           "hqh_point_source", "hqh_stations", "hqh_solver_run", "hqh_source_table",
           "hqh_forcefile_info", "hqh_forcefile_read", "hqh_forcefile_write",
           "hqh_checkpoint_write", "hqh_checkpoint_read", "hqh_station_format", "hqh_station_format_derivs",
           "hqh_station_kinematics", "hqh_station_header", "hqh_wavefield_create", "hqh_wavefield_write",
           "hqh_octbox_create", "hqh_octbox_destroy", "hqh_octbox_desc", "hqh_octbox_view",
           "hqh_cvm_open", "hqh_cvm_close", "hqh_cvm_info", "hqh_cvm_query", "hqh_cvm_grid"]


class _BoxParams(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nz", ctypes.c_int32),
                ("h", ctypes.c_double), ("nlayers", ctypes.c_int32),
                ("layer_ztop", ctypes.c_void_p), ("layer_vp", ctypes.c_void_p),
                ("layer_vs", ctypes.c_void_p), ("layer_rho", ctypes.c_void_p),
                ("deltaT", ctypes.c_double), ("freq", ctypes.c_double), ("damping", ctypes.c_int32),
                ("threshold_damping", ctypes.c_double), ("threshold_vpvs", ctypes.c_double),
                ("halfspace", ctypes.c_int32), ("rank", ctypes.c_int32), ("nranks", ctypes.c_int32),
                ("lateral_classes", ctypes.c_int32), ("lateral_amp", ctypes.c_double), ("origin", ctypes.c_int32 * 3),
                ("solver_float", ctypes.c_int32)]


class _BoxInfo(ctypes.Structure):
    _fields_ = [("total_elements", ctypes.c_int64), ("total_nodes", ctypes.c_int64),
                ("lenum", ctypes.c_int32), ("nharbored", ctypes.c_int32), ("nowned", ctypes.c_int32),
                ("nneighbors", ctypes.c_int32), ("shared_nodes", ctypes.c_int64)]


STATION_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                              ctypes.POINTER(ctypes.c_double))


class _RunParams(ctypes.Structure):
    _fields_ = [("force_file", ctypes.c_char_p), ("nloaded", ctypes.c_int32), ("loaded_lnid", ctypes.c_void_p),
                ("pattern", ctypes.c_void_p), ("moment", ctypes.c_double),
                ("rise_time", ctypes.c_double), ("source_window", ctypes.c_int32),
                ("nstations", ctypes.c_int32), ("station_ids", ctypes.c_void_p),
                ("station_phi", ctypes.c_void_p), ("station_rate", ctypes.c_int32),
                ("station_fn", STATION_FN), ("station_user", ctypes.c_void_p),
                ("nplanes", ctypes.c_int32), ("plane_npoints", ctypes.c_void_p),
                ("plane_ids", ctypes.c_void_p), ("plane_phi", ctypes.c_void_p),
                ("plane_mine", ctypes.c_void_p), ("plane_rate", ctypes.c_int32),
                ("plane_dir", ctypes.c_char_p), ("checkpoint_rate", ctypes.c_int32),
                ("checkpoint_dir", ctypes.c_char_p), ("station_derivs", ctypes.c_int32),
                ("wavefield_rate", ctypes.c_int32), ("wavefield_disp_file", ctypes.c_char_p),
                ("wavefield_vel_file", ctypes.c_char_p), ("wavefield_total_nodes", ctypes.c_int64),
                ("wavefield_base_gnid", ctypes.c_int64), ("wavefield_first_owned", ctypes.c_int32),
                ("wavefield_count", ctypes.c_int32)]


class _WavefieldInfo(ctypes.Structure):
    _fields_ = [("total_nodes", ctypes.c_int64), ("total_elements", ctypes.c_int64),
                ("domain_x", ctypes.c_double), ("domain_y", ctypes.c_double), ("domain_z", ctypes.c_double),
                ("mesh_ticksize", ctypes.c_double), ("delta_t", ctypes.c_double),
                ("output_rate", ctypes.c_int32), ("total_time_steps", ctypes.c_int32)]


def wavefield_create(path, quantity, total_nodes, total_elements, domain, ticksize, dt, rate, total_steps):
    """hqh_wavefield_create: the reference's 4D output file with its 136-byte header; quantity
    "displacement" or "velocity"."""
    w = _WavefieldInfo(int(total_nodes), int(total_elements), float(domain[0]), float(domain[1]), float(domain[2]),
                       float(ticksize), float(dt), int(rate), int(total_steps))
    capi._check(load_library().hqh_wavefield_create(os.fsencode(path), ctypes.byref(w),
                                                    ctypes.c_int32({"displacement": 1, "velocity": 2}[quantity])))


class _Plane(ctypes.Structure):
    _fields_ = [("origin", ctypes.c_double * 3), ("step_strike", ctypes.c_double), ("n_strike", ctypes.c_int32),
                ("step_dip", ctypes.c_double), ("n_dip", ctypes.c_int32), ("strike_deg", ctypes.c_double),
                ("dip_deg", ctypes.c_double)]


def plane_points(origin, step_strike, n_strike, step_dip, n_dip, strike_deg, dip_deg):
    """Grid points of an output plane, [n_strike * n_dip, 3] (hqh_plane_points)."""
    lib = load_library()
    pl = _Plane((ctypes.c_double * 3)(*[float(v) for v in origin]), float(step_strike), int(n_strike),
                float(step_dip), int(n_dip), float(strike_deg), float(dip_deg))
    out = np.zeros((int(n_strike) * int(n_dip), 3))
    capi._check(lib.hqh_plane_points(ctypes.byref(pl), out.ctypes.data_as(ctypes.c_void_p)))
    return out


def domain_coords(lon, lat, lon_corners, lat_corners, len_x, len_y):
    """(longitude, latitude) -> domain (x, y) through the four surface corners (hqh_domain_coords)."""
    lib = load_library()
    lc = (ctypes.c_double * 4)(*[float(v) for v in lon_corners])
    la = (ctypes.c_double * 4)(*[float(v) for v in lat_corners])
    x, y = ctypes.c_double(), ctypes.c_double()
    capi._check(lib.hqh_domain_coords(ctypes.c_double(lon), ctypes.c_double(lat), lc, la, ctypes.c_double(len_x),
                                      ctypes.c_double(len_y), ctypes.byref(x), ctypes.byref(y)))
    return x.value, y.value


_lib = None


def load_library():
    global _lib
    if _lib is None:
        capi.load_library()                      # libhq_solver.so first (RTLD_GLOBAL)
        if not os.path.exists(_LIBPATH):
            raise capi.HqError("native library %s is missing: run `python -m hercules_amd.build`" % _LIBPATH)
        lib = ctypes.CDLL(_LIBPATH)
        for n in ("hqh_box_lnid", "hqh_box_node_ijk", "hqh_box_etable", "hqh_box_ntable", "hqh_box_owner"):
            getattr(lib, n).restype = ctypes.c_void_p
            getattr(lib, n).argtypes = [ctypes.c_void_p]
        lib.hqh_box_destroy.restype = None
        lib.hqh_box_destroy.argtypes = [ctypes.c_void_p]
        lib.hqh_source_table.restype = None
        _lib = lib
    return _lib


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def run_params(loaded=None, pattern=None, moment=1.0, rise_time=0.1, source_window=256,
               station_ids=None, station_phi=None, station_rate=0, station_fn=None, force_file=None,
               planes=None, plane_rate=0, plane_dir=None, checkpoint_rate=0, checkpoint_dir=None,
               station_derivs=0, wavefield_rate=0, wavefield_disp_file=None, wavefield_vel_file=None,
               wavefield_total_nodes=0, wavefield_owned=None):
    """wavefield_*: every wavefield_rate steps the owned nodes' displacement / velocity go to their
    place in the 4D file(s) made by wavefield_create; wavefield_owned = (base_gnid, first_owned,
    count) for a partition (None: the whole mesh).
    station_derivs: 0 = the station callback gets displacements [n, 3]; 1 = + velocities [n, 6];
    2 = + accelerations [n, 9] (print_station_velocities / _accelerations).
    planes: list of (ids [n,8], phi [n,8]) or (ids, phi, mine [n]) per output plane, written
    every plane_rate steps to <plane_dir>/planedisplacements.<i> (the reference's format)."""
    rp = _RunParams()
    keep = []
    if checkpoint_rate > 0 and checkpoint_dir is not None:
        rp.checkpoint_rate, rp.checkpoint_dir = int(checkpoint_rate), os.fsencode(checkpoint_dir)
    if wavefield_rate > 0 and (wavefield_disp_file or wavefield_vel_file):
        rp.wavefield_rate = int(wavefield_rate)
        if wavefield_disp_file:
            rp.wavefield_disp_file = os.fsencode(wavefield_disp_file)
        if wavefield_vel_file:
            rp.wavefield_vel_file = os.fsencode(wavefield_vel_file)
        rp.wavefield_total_nodes = int(wavefield_total_nodes)
        if wavefield_owned is not None:
            rp.wavefield_base_gnid, rp.wavefield_first_owned, rp.wavefield_count = [int(v) for v in wavefield_owned]
    if planes and plane_rate > 0 and plane_dir is not None:
        npts = np.array([len(p[0]) for p in planes], np.int32)
        pid = np.ascontiguousarray(np.concatenate([np.asarray(p[0]).reshape(-1, 8) for p in planes]), np.int32)
        pph = np.ascontiguousarray(np.concatenate([np.asarray(p[1]).reshape(-1, 8) for p in planes]), np.float64)
        keep += [npts, pid, pph]
        rp.nplanes, rp.plane_npoints, rp.plane_ids, rp.plane_phi = len(planes), npts.ctypes.data, pid.ctypes.data, pph.ctypes.data
        if any(len(p) > 2 for p in planes):
            pm = np.ascontiguousarray(np.concatenate([np.asarray(p[2]) if len(p) > 2 else np.ones(len(p[0]))
                                                      for p in planes]), np.int32)
            keep.append(pm)
            rp.plane_mine = pm.ctypes.data
        rp.plane_rate, rp.plane_dir = int(plane_rate), os.fsencode(plane_dir)
    if force_file is not None:
        rp.force_file = os.fsencode(force_file)
        l = np.ascontiguousarray(loaded, np.int32)
        keep.append(l)
        rp.nloaded, rp.loaded_lnid = len(l), l.ctypes.data
    elif loaded is not None and len(loaded):
        l = np.ascontiguousarray(loaded, np.int32)
        pt = np.ascontiguousarray(pattern, np.float64)
        keep += [l, pt]
        rp.nloaded, rp.loaded_lnid, rp.pattern = len(l), l.ctypes.data, pt.ctypes.data
    rp.moment, rp.rise_time, rp.source_window = moment, rise_time, source_window
    if station_ids is not None and len(station_ids) and station_fn is not None:
        si = np.ascontiguousarray(station_ids, np.int32)
        sp = np.ascontiguousarray(station_phi, np.float64)
        n = len(si)

        ncol = 3 * (1 + int(station_derivs))

        def _cb(user, step, ns, disp):
            station_fn(step, np.ctypeslib.as_array(disp, (ns, ncol)).copy())
        cb = STATION_FN(_cb)
        keep += [si, sp, cb]
        rp.nstations, rp.station_ids, rp.station_phi = n, si.ctypes.data, sp.ctypes.data
        rp.station_rate, rp.station_fn = station_rate, cb
        rp.station_derivs = int(station_derivs)
    rp._keep = keep
    return rp



class Box:
    """One partition of a uniform layered box (hqh_box)."""

    def __init__(self, nx, ny, nz, h, dt, freq, vp=6000.0, vs=3464.0, rho=2700.0, layers=None,
                 damping="rayleigh", threshold_damping=0.05, threshold_vpvs=3.0, halfspace=True,
                 rank=0, nranks=1, lateral_classes=0, lateral_amp=0.0, origin=(0, 0, 0), solver_float=8):
        """solver_float = 4: the n_t rows as the reference's -DSINGLE_PRECISION_SOLVER build sums them (hq_host.h)."""
        lib = load_library()
        if layers is None:
            layers = [(0.0, vp, vs, rho)]
        zt = np.array([l[0] for l in layers], np.float64)
        lvp = np.array([l[1] for l in layers], np.float32)
        lvs = np.array([l[2] for l in layers], np.float32)
        lrho = np.array([l[3] for l in layers], np.float32)
        p = _BoxParams(nx, ny, nz, h, len(layers), zt.ctypes.data, lvp.ctypes.data, lvs.ctypes.data,
                       lrho.ctypes.data, dt, freq, DAMPING[damping], threshold_damping, threshold_vpvs,
                       int(halfspace), rank, nranks, int(lateral_classes), float(lateral_amp),
                       (ctypes.c_int32 * 3)(*[int(v) for v in origin]), int(solver_float))
        self.solver_float = int(solver_float)
        self._h = ctypes.c_void_p()
        rc = lib.hqh_box_create(ctypes.byref(p), ctypes.byref(self._h))
        if rc != 0:
            raise capi.HqError("hqh_box_create failed: %d" % rc)
        self._lib = lib
        self.nx, self.ny, self.nz, self.h, self.dt = nx, ny, nz, h, dt
        self.rank, self.nranks = rank, nranks
        i = _BoxInfo()
        lib.hqh_box_get_info(self._h, ctypes.byref(i))
        self.info = {k: getattr(i, k) for k, _ in _BoxInfo._fields_}
        E, N = i.lenum, i.nharbored
        self.lnid = _view(lib.hqh_box_lnid(self._h), (E, 8), np.int32)
        self.node_ijk = _view(lib.hqh_box_node_ijk(self._h), (N, 3), np.int32)
        self.etable = _view(lib.hqh_box_etable(self._h), (E, 4), np.float64)
        self.ntable = _view(lib.hqh_box_ntable(self._h), (N, 7), np.float64)
        self.owner = _view(lib.hqh_box_owner(self._h), (N,), np.int32)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            for k in ("lnid", "node_ijk", "etable", "ntable", "owner"):
                setattr(self, k, None)
            self._lib.hqh_box_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    def material(self):
        """[lenum][3] float32 Vp, Vs, rho of this partition's elements (edata_t as solver_init reads it)."""
        out = np.empty((self.info["lenum"], 3), np.float32)
        rc = self._lib.hqh_box_material(self._h, out.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise capi.HqError("hqh_box_material failed: %d" % rc)
        return out

    def schedule(self):
        """an_sched as {"c": [(procid, mapping)], "s": [...]} (copies)."""
        d = capi._Desc()
        self._lib.hqh_box_desc(self._h, ctypes.byref(d))
        out = {}
        for key, cnt, first in (("c", d.an_sched.c_count, d.an_sched.first_c),
                                ("s", d.an_sched.s_count, d.an_sched.first_s)):
            out[key] = [(first[i].procid,
                         _view(first[i].mapping, (first[i].nodecount,), np.int32).copy()) for i in range(cnt)]
        return out

    def plan_check(self):
        """The patch planner's host-only self-check on this box (capi.plan_check)."""
        d = capi._Desc()
        rc = self._lib.hqh_box_desc(self._h, ctypes.byref(d))
        if rc != 0:
            raise capi.HqError("hqh_box_desc failed: %d" % rc)
        return capi.plan_check(d)

    def stencil_plan_check(self):
        """The stencil kernel's tables against this box (capi.stencil_plan_check, host only)."""
        d = capi._Desc()
        rc = self._lib.hqh_box_desc(self._h, ctypes.byref(d))
        if rc != 0:
            raise capi.HqError("hqh_box_desc failed: %d" % rc)
        return capi.stencil_plan_check(d)

    def brick_plan_check(self):
        """The brick planner against this box's connectivity (capi.brick_plan_check, host only)."""
        d = capi._Desc()
        rc = self._lib.hqh_box_desc(self._h, ctypes.byref(d))
        if rc != 0:
            raise capi.HqError("hqh_box_desc failed: %d" % rc)
        return capi.brick_plan_check(d)

    def create_solver(self, variant=capi.HQ_VARIANT_AUTO, device=0, tm1=None, tm2=None, options=None, precision="f64"):
        """hq_create[_opts] on the arrays the C host side built (no copies through Python)."""
        d = capi._Desc()
        rc = self._lib.hqh_box_desc(self._h, ctypes.byref(d))
        if rc != 0:
            raise capi.HqError("hqh_box_desc failed: %d" % rc)
        return _solver_from_desc(d, self.ntable, variant, device, tm1, tm2, options, precision)

    def point_source(self, x, y, z, strike=0.0, dip=90.0, rake=0.0):
        n = ctypes.c_int32()
        ids = np.zeros(8, np.int32)
        pat = np.zeros((8, 3))
        rc = self._lib.hqh_point_source(self._h, ctypes.c_double(x), ctypes.c_double(y), ctypes.c_double(z),
                                        ctypes.c_double(strike), ctypes.c_double(dip), ctypes.c_double(rake),
                                        ctypes.byref(n), ids.ctypes.data_as(ctypes.c_void_p),
                                        pat.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise capi.HqError("hqh_point_source failed: %d" % rc)
        return (ids, pat) if n.value == 8 else (ids[:0], pat[:0])

    def stations(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        n = len(xyz)
        ids = np.zeros((n, 8), np.int32)
        phi = np.zeros((n, 8))
        mine = np.zeros(n, np.int32)
        rc = self._lib.hqh_stations(self._h, ctypes.c_int32(n), xyz.ctypes.data_as(ctypes.c_void_p),
                                    ids.ctypes.data_as(ctypes.c_void_p), phi.ctypes.data_as(ctypes.c_void_p),
                                    mine.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise capi.HqError("hqh_stations failed: %d" % rc)
        return ids, phi, mine

    def run_params(self, *args, **kw):
        """See host.run_params."""
        return run_params(*args, **kw)

    def source_table(self, rp, step0, nsteps):
        F = np.zeros((nsteps, rp.nloaded, 3))
        self._lib.hqh_source_table(ctypes.byref(rp), ctypes.c_double(self.dt), ctypes.c_int32(step0),
                                   ctypes.c_int32(nsteps), F.ctypes.data_as(ctypes.c_void_p))
        return F

    def solver_run(self, solver, rp, step0, nsteps):
        capi._check(self._lib.hqh_solver_run(solver._h, self._h, ctypes.byref(rp), ctypes.c_int32(step0),
                                             ctypes.c_int32(nsteps)))


class _OctParams(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nz_fine", ctypes.c_int32),
                ("nz_coarse", ctypes.c_int32), ("h", ctypes.c_double),
                ("vp_top", ctypes.c_float), ("vs_top", ctypes.c_float), ("rho_top", ctypes.c_float),
                ("vp_bot", ctypes.c_float), ("vs_bot", ctypes.c_float), ("rho_bot", ctypes.c_float),
                ("deltaT", ctypes.c_double), ("freq", ctypes.c_double), ("damping", ctypes.c_int32),
                ("threshold_damping", ctypes.c_double), ("threshold_vpvs", ctypes.c_double),
                ("halfspace", ctypes.c_int32), ("rank", ctypes.c_int32), ("nranks", ctypes.c_int32),
                ("solver_float", ctypes.c_int32)]


class _Layered(ctypes.Structure):
    _fields_ = [("nlayers", ctypes.c_int32), ("ztop", ctypes.c_void_p), ("vp", ctypes.c_void_p),
                ("vs", ctypes.c_void_p), ("rho", ctypes.c_void_p)]


def layered_column(layers, h0, ncoarse, factor, vscut=0.0):
    """Leaves (edge, vp, vs, rho) from the top of the column the reference's mesher makes of the
    layered model layers = [(ztop, vp, vs, rho), ...] (hqh_layered_column: Vs rule + 2:1 balance)."""
    lib = load_library()
    zt = np.array([l[0] for l in layers], np.float64)
    m = [np.array([l[1 + c] for l in layers], np.float32) for c in range(3)]
    mod = _Layered(len(layers), zt.ctypes.data, m[0].ctypes.data, m[1].ctypes.data, m[2].ctypes.data)
    cap = 4096
    edge = np.zeros(cap)
    out = [np.zeros(cap, np.float32) for _ in range(3)]
    n = ctypes.c_int32()
    capi._check(lib.hqh_layered_column(ctypes.byref(mod), ctypes.c_double(h0), ctypes.c_int32(ncoarse),
                                       ctypes.c_double(factor), ctypes.c_double(vscut), ctypes.c_int32(cap),
                                       edge.ctypes.data_as(ctypes.c_void_p), out[0].ctypes.data_as(ctypes.c_void_p),
                                       out[1].ctypes.data_as(ctypes.c_void_p), out[2].ctypes.data_as(ctypes.c_void_p),
                                       ctypes.byref(n)))
    return [(float(edge[i]), float(out[0][i]), float(out[1][i]), float(out[2][i])) for i in range(n.value)]


def levels_from_column(column):
    """(finest edge, levels for OctBox(levels=...)) from layered_column(); the column must coarsen
    monotonically with depth."""
    h = min(c[0] for c in column)
    levels = []
    for e, vp, vs, rho in column:
        L = int(round(np.log2(e / h)))
        if levels and L < len(levels) - 1:
            raise ValueError("column refines again with depth")
        while len(levels) <= L:
            levels.append([0, []])
        levels[L][0] += 1
        levels[L][1].append((vp, vs, rho))
    return h, [(n, mats) for n, mats in levels]


class _OctLevels(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nlevels", ctypes.c_int32),
                ("layers", ctypes.c_void_p), ("h", ctypes.c_double),
                ("vp", ctypes.c_void_p), ("vs", ctypes.c_void_p), ("rho", ctypes.c_void_p),
                ("deltaT", ctypes.c_double), ("freq", ctypes.c_double), ("damping", ctypes.c_int32),
                ("threshold_damping", ctypes.c_double), ("threshold_vpvs", ctypes.c_double),
                ("halfspace", ctypes.c_int32), ("rank", ctypes.c_int32), ("nranks", ctypes.c_int32),
                ("solver_float", ctypes.c_int32)]


class OctBox:
    """Layered box on two (or, with `levels`, any number of) octree levels with hanging nodes
    (hqh_octbox), whole or one of nranks partitions.  levels = [(layers, vp, vs, rho), ...] or
    [(layers, [(vp, vs, rho) per element layer]), ...] from the top, level L with elements of edge
    h * 2^L; nz_fine / nz_coarse / top / bottom are ignored then."""

    def __init__(self, nx, ny, nz_fine, nz_coarse, h, dt, freq, top=(3000.0, 1732.0, 2200.0),
                 bottom=(6000.0, 3464.0, 2700.0), damping="rayleigh", threshold_damping=0.05,
                 threshold_vpvs=3.0, halfspace=True, rank=0, nranks=1, levels=None, solver_float=8):
        lib = load_library()
        lib.hqh_octbox_view.restype = ctypes.c_void_p
        lib.hqh_octbox_view.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)]
        lib.hqh_octbox_destroy.restype = None
        lib.hqh_octbox_destroy.argtypes = [ctypes.c_void_p]
        p = _OctParams(nx, ny, nz_fine, nz_coarse, h, top[0], top[1], top[2], bottom[0], bottom[1], bottom[2],
                       dt, freq, DAMPING[damping], threshold_damping, threshold_vpvs, int(halfspace),
                       int(rank), int(nranks), int(solver_float))
        self.rank, self.nranks, self.solver_float = int(rank), int(nranks), int(solver_float)
        self._h = ctypes.c_void_p()
        if levels is not None:
            lay = np.array([l[0] for l in levels], np.int32)
            # a level's material: one (vp, vs, rho) for all its element layers, or a list of them
            per = []
            for l in levels:
                per += [tuple(l[1:4])] * int(l[0]) if not isinstance(l[1], (list, tuple)) else list(l[1])
            assert len(per) == int(lay.sum())
            mats = [np.array([m[c] for m in per], np.float32) for c in range(3)]
            q = _OctLevels(nx, ny, len(levels), lay.ctypes.data, h, mats[0].ctypes.data, mats[1].ctypes.data,
                           mats[2].ctypes.data, dt, freq, DAMPING[damping], threshold_damping, threshold_vpvs,
                           int(halfspace), int(rank), int(nranks), int(solver_float))
            rc = lib.hqh_octbox_create_levels(ctypes.byref(q), ctypes.byref(self._h))
        else:
            rc = lib.hqh_octbox_create(ctypes.byref(p), ctypes.byref(self._h))
        if rc != 0:
            raise capi.HqError("hqh_octbox_create failed: %d" % rc)
        self._lib = lib
        self.dt = dt

        self._load_views()

    def _load_views(self):
        lib = self._lib

        def view(which, dtype, cols):
            n = ctypes.c_int64()
            ptr = lib.hqh_octbox_view(self._h, which, ctypes.byref(n))
            a = _view(ptr, (n.value,), dtype)
            return a.reshape(-1, cols) if cols > 1 else a
        self.lnid = view(0, np.int32, 8)
        self.node_xyz = view(1, np.int32, 3)
        self.dangling = (view(2, np.int32, 1), view(3, np.int32, 1), view(4, np.int32, 1))
        self.etable = view(5, np.float64, 4)
        self.ntable = view(6, np.float64, 7)
        self.owner = view(7, np.int32, 1)
        self.gid = view(8, np.int32, 1)
        self.E, self.N, self.ldnnum = len(self.lnid), len(self.node_xyz), len(self.dangling[0])

    @classmethod
    def from_leaves(cls, elem_ticks, elem_edge, edata, far_ticks, dt, freq, damping="rayleigh",
                    threshold_damping=0.05, threshold_vpvs=3.0, halfspace=True, rank=0, nranks=1, solver_float=8):
        """An octree mesh from its leaves in pre-order (hqh_mesh_from_leaves), e.g. those of a
        mesh.e read with etree_read."""
        lib = load_library()
        lib.hqh_octbox_view.restype = ctypes.c_void_p
        lib.hqh_octbox_view.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)]
        lib.hqh_octbox_destroy.restype = None
        lib.hqh_octbox_destroy.argtypes = [ctypes.c_void_p]
        et = np.ascontiguousarray(elem_ticks, np.uint32).reshape(-1, 3)
        ee = np.ascontiguousarray(elem_edge, np.uint32).reshape(-1)
        ed = np.ascontiguousarray(edata, np.float32).reshape(-1, 4)
        far = (ctypes.c_uint32 * 3)(*[int(v) for v in far_ticks])
        ip = _InitParams(dt, freq, DAMPING[damping], threshold_damping, threshold_vpvs, int(halfspace),
                         int(rank), int(nranks), int(solver_float))
        self = cls.__new__(cls)
        self.rank, self.nranks, self.solver_float = int(rank), int(nranks), int(solver_float)
        self._h = ctypes.c_void_p()
        rc = lib.hqh_mesh_from_leaves(ctypes.c_int64(len(et)), et.ctypes.data_as(ctypes.c_void_p),
                                      ee.ctypes.data_as(ctypes.c_void_p), ed.ctypes.data_as(ctypes.c_void_p), far,
                                      ctypes.byref(ip), ctypes.byref(self._h))
        if rc != 0:
            raise capi.HqError("hqh_mesh_from_leaves failed: %d" % rc)
        self._lib = lib
        self.dt = dt
        self._load_views()
        return self

    def run_params(self, *args, **kw):
        """See host.run_params."""
        return run_params(*args, **kw)

    def solver_run(self, solver, rp, step0, nsteps):
        capi._check(self._lib.hqh_octbox_solver_run(solver._h, self._h, ctypes.byref(rp), ctypes.c_int32(step0),
                                                    ctypes.c_int32(nsteps)))

    def schedules(self):
        """{"an": {"c": [(procid, mapping)], "s": [...]}, "dn": {...}} (copies)."""
        d = capi._Desc()
        self._lib.hqh_octbox_desc(self._h, ctypes.byref(d))
        out = {}
        for name, sch in (("an", d.an_sched), ("dn", d.dn_sched)):
            out[name] = {}
            for key, cnt, first in (("c", sch.c_count, sch.first_c), ("s", sch.s_count, sch.first_s)):
                out[name][key] = [(first[i].procid,
                                   _view(first[i].mapping, (first[i].nodecount,), np.int32).copy())
                                  for i in range(cnt)]
        return out

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lnid = self.node_xyz = self.dangling = self.etable = self.ntable = self.owner = self.gid = None
            self._lib.hqh_octbox_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    def plan_check(self):
        """The patch planner's host-only self-check on this box (capi.plan_check)."""
        d = capi._Desc()
        rc = self._lib.hqh_octbox_desc(self._h, ctypes.byref(d))
        if rc != 0:
            raise capi.HqError("hqh_octbox_desc failed: %d" % rc)
        return capi.plan_check(d)

    def stencil_plan_check(self):
        """The stencil kernel's tables against this box (capi.stencil_plan_check, host only)."""
        d = capi._Desc()
        rc = self._lib.hqh_octbox_desc(self._h, ctypes.byref(d))
        if rc != 0:
            raise capi.HqError("hqh_octbox_desc failed: %d" % rc)
        return capi.stencil_plan_check(d)

    def brick_plan_check(self):
        """The brick planner against this box's connectivity (capi.brick_plan_check, host only)."""
        d = capi._Desc()
        rc = self._lib.hqh_octbox_desc(self._h, ctypes.byref(d))
        if rc != 0:
            raise capi.HqError("hqh_octbox_desc failed: %d" % rc)
        return capi.brick_plan_check(d)

    def create_solver(self, variant=capi.HQ_VARIANT_AUTO, device=0, tm1=None, tm2=None, options=None, precision="f64"):
        d = capi._Desc()
        rc = self._lib.hqh_octbox_desc(self._h, ctypes.byref(d))
        if rc != 0:
            raise capi.HqError("hqh_octbox_desc failed: %d" % rc)
        return _solver_from_desc(d, self.ntable, variant, device, tm1, tm2, options, precision)


def _solver_from_desc(d, ntable, variant, device, tm1, tm2, options, precision):
    """capi.Solver on a description the C host side filled.  precision "f32": libhq_solver_f32.so -- the n_t rows go over as a
    float array: exact where the box was made with solver_float=4 (the float reference's own sums, hq_host.h), ROUNDED from
    the double build's rows otherwise; tm1 / tm2 are taken as float32."""
    d.variant = variant
    real = np.float32 if precision == "f32" else np.float64
    keep = []
    if precision == "f32":
        nt = np.empty(ntable.shape, np.float32)
        src = np.ascontiguousarray(ntable, np.float64)
        rc = load_library().hqh_ntable_to_float(src.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(src)),
                                                nt.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise capi.HqError("hqh_ntable_to_float failed: %d" % rc)
        keep.append(nt)
        d.nTable = nt.ctypes.data
    for name, a in (("tm1", tm1), ("tm2", tm2)):
        if a is not None:
            a = np.ascontiguousarray(a, real)
            keep.append(a)
            setattr(d, name, a.ctypes.data)
    s = capi.Solver.__new__(capi.Solver)
    s._lib = capi.load_library(precision=precision)
    s.real = real
    s._h = ctypes.c_void_p()
    s.N, s.E = d.nharbored, d.lenum
    if isinstance(options, dict):
        options = capi.Options(**options)
    if options is None:
        capi._check(s._lib.hq_create(ctypes.byref(d), ctypes.c_int(device), ctypes.byref(s._h)), s._lib)
    else:
        capi._check(s._lib.hq_create_opts(ctypes.byref(d), ctypes.c_int(device), ctypes.byref(options), ctypes.byref(s._h)), s._lib)
    return s


class _GridModel(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nz", ctypes.c_int32), ("cell", ctypes.c_double),
                ("vp", ctypes.c_void_p), ("vs", ctypes.c_void_p), ("rho", ctypes.c_void_p)]


class _MesherParams(ctypes.Structure):
    _fields_ = [("domain", ctypes.c_double * 3), ("factor", ctypes.c_double), ("vscut", ctypes.c_double),
                ("max_level", ctypes.c_int32)]


def octree_generate(vp, vs, rho, cell, domain, factor, vscut=0.0, max_level=0):
    """hqh_octree_generate: the leaves the reference's mesher makes of a material model given on a regular grid
    (vp / vs / rho [nz][ny][nx] in the MESH's axes, cell edge in metres): Vs rule on the 27-sample record + 2:1
    balancing.  -> elem_ticks [E,3] uint32, elem_edge [E] uint32, edata [E,4] float32, far_ticks (3,), ticksize."""
    lib = load_library()
    vp, vs, rho = [np.ascontiguousarray(a, np.float32) for a in (vp, vs, rho)]
    assert vp.ndim == 3 and vp.shape == vs.shape == rho.shape
    m = _GridModel(vp.shape[2], vp.shape[1], vp.shape[0], float(cell), vp.ctypes.data, vs.ctypes.data, rho.ctypes.data)
    p = _MesherParams((ctypes.c_double * 3)(*[float(v) for v in domain]), float(factor), float(vscut), int(max_level))
    E = ctypes.c_int64()
    t, e, d = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    far = (ctypes.c_uint32 * 3)()
    ts = ctypes.c_double()
    lib.hqh_free.restype = None
    lib.hqh_free.argtypes = [ctypes.c_void_p]
    rc = lib.hqh_octree_generate(ctypes.byref(m), ctypes.byref(p), ctypes.byref(E), ctypes.byref(t), ctypes.byref(e),
                                 ctypes.byref(d), far, ctypes.byref(ts))
    if rc != 0:
        raise capi.HqError("hqh_octree_generate failed: %d" % rc)
    n = E.value
    ticks = np.ctypeslib.as_array(ctypes.cast(t, ctypes.POINTER(ctypes.c_uint32)), (n, 3)).copy()
    edge = np.ctypeslib.as_array(ctypes.cast(e, ctypes.POINTER(ctypes.c_uint32)), (n,)).copy()
    edata = np.ctypeslib.as_array(ctypes.cast(d, ctypes.POINTER(ctypes.c_float)), (n, 4)).copy()
    for q in (t, e, d):
        lib.hqh_free(q)
    return ticks, edge, edata, tuple(int(v) for v in far), ts.value


def etree_read(path):
    """Leaves of an etree file in key order (hqh_etree_read): ticks [n,3] uint32, level [n],
    raw payloads [n, value_size] uint8."""
    lib = load_library()
    n, vs = ctypes.c_int64(), ctypes.c_int32()
    pt, pl, pv = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    capi._check(lib.hqh_etree_read(os.fsencode(path), ctypes.byref(n), ctypes.byref(vs), ctypes.byref(pt),
                                   ctypes.byref(pl), ctypes.byref(pv)))
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    try:
        ticks = _view(pt.value, (n.value, 3), np.uint32).copy()
        level = _view(pl.value, (n.value,), np.int32).copy()
        vals = _view(pv.value, (n.value, vs.value), np.uint8).copy()
    finally:
        for p in (pt, pl, pv):
            libc.free(p)
    return ticks, level, vals


class Cvm:
    """A CVM etree (hqh_cvm_*): the material database the reference's mesher queries -- cvm_query's point location on
    this library's own etree reader.  grid() -> (vp, vs, rho [nz, ny, nx] float32 in the MESH's axes, cell edge in m):
    the model hqh_octree_generate takes."""

    def __init__(self, path):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        capi._check(self._lib.hqh_cvm_open(os.fsencode(path), ctypes.byref(self._h)))
        n, lv, reg, tick = ctypes.c_int64(), (ctypes.c_int32 * 2)(), (ctypes.c_double * 3)(), ctypes.c_double()
        capi._check(self._lib.hqh_cvm_info(self._h, ctypes.byref(n), lv, reg, ctypes.byref(tick)))
        self.nleaves, self.levels, self.region, self.ticksize = n.value, (lv[0], lv[1]), tuple(reg), tick.value

    def query(self, east_m, north_m, depth_m):
        """cvm_query: (Vp, Vs, density) of the leaf octant that holds the point, or None outside the database."""
        out = (ctypes.c_float * 3)()
        rc = self._lib.hqh_cvm_query(self._h, ctypes.c_double(east_m), ctypes.c_double(north_m), ctypes.c_double(depth_m), out)
        return None if rc != 0 else (out[0], out[1], out[2])

    def grid(self):
        dims, cell = (ctypes.c_int32 * 3)(), ctypes.c_double()
        p = [ctypes.c_void_p() for _ in range(3)]
        capi._check(self._lib.hqh_cvm_grid(self._h, dims, ctypes.byref(cell), *[ctypes.byref(q) for q in p]))
        self._lib.hqh_free.argtypes = [ctypes.c_void_p]
        self._lib.hqh_free.restype = None
        try:
            out = [_view(q.value, (dims[2], dims[1], dims[0]), np.float32).copy() for q in p]
        finally:
            for q in p:
                self._lib.hqh_free(q)
        return out[0], out[1], out[2], cell.value

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.hqh_cvm_close.argtypes = [ctypes.c_void_p]
            self._lib.hqh_cvm_close.restype = None
            self._lib.hqh_cvm_close(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close


def mesh_payload(values):
    """mdata_t (psolve.h:84-87) of mesh.e payloads: global node ids [n,8] int64, edata [n,4] float32."""
    v = np.ascontiguousarray(values)
    return v[:, :64].copy().view("<i8").reshape(-1, 8), v[:, 64:80].copy().view("<f4").reshape(-1, 4)


class _InitParams(ctypes.Structure):
    _fields_ = [("deltaT", ctypes.c_double), ("freq", ctypes.c_double), ("damping", ctypes.c_int32),
                ("threshold_damping", ctypes.c_double), ("threshold_vpvs", ctypes.c_double),
                ("halfspace", ctypes.c_int32), ("rank", ctypes.c_int32), ("nranks", ctypes.c_int32),
                ("solver_float", ctypes.c_int32)]


def forcefile_info(path):
    lib = load_library()
    n, ns = ctypes.c_int32(), ctypes.c_int32()
    rc = lib.hqh_forcefile_info(os.fsencode(path), ctypes.byref(n), ctypes.byref(ns), None, 0)
    if rc != 0:
        raise capi.HqError("hqh_forcefile_info failed: %d" % rc)
    ids = np.zeros(n.value, np.int32)
    lib.hqh_forcefile_info(os.fsencode(path), ctypes.byref(n), ctypes.byref(ns),
                           ids.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(n.value))
    return ids, ns.value


def forcefile_read(path, nloaded, step0, nsteps):
    F = np.empty((nsteps, nloaded, 3))
    rc = load_library().hqh_forcefile_read(os.fsencode(path), ctypes.c_int32(step0), ctypes.c_int32(nsteps),
                                           F.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise capi.HqError("hqh_forcefile_read failed: %d" % rc)
    return F


def forcefile_write(path, lnid, F):
    ids = np.ascontiguousarray(lnid, np.int32)
    F = np.ascontiguousarray(F, np.float64)
    rc = load_library().hqh_forcefile_write(os.fsencode(path), ctypes.c_int32(len(ids)),
                                            ids.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(F.shape[0]),
                                            F.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise capi.HqError("hqh_forcefile_write failed: %d" % rc)


def checkpoint_write(solver, path, step, rank=0, nranks=1, nharboredmax=None):
    nh = solver.N
    rc = load_library().hqh_checkpoint_write(solver._h, os.fsencode(path), ctypes.c_int32(step), ctypes.c_int32(rank),
                                             ctypes.c_int32(nranks), ctypes.c_int32(nh),
                                             ctypes.c_int32(nh if nharboredmax is None else nharboredmax))
    if rc != 0:
        raise capi.HqError("hqh_checkpoint_write failed: %d" % rc)


def checkpoint_read(solver, path, rank=0, nranks=1):
    step = ctypes.c_int32()
    rc = load_library().hqh_checkpoint_read(solver._h, os.fsencode(path), ctypes.c_int32(rank), ctypes.c_int32(nranks),
                                            ctypes.c_int32(solver.N), ctypes.byref(step))
    if rc != 0:
        raise capi.HqError("hqh_checkpoint_read failed: %d" % rc)
    return step.value


def station_header(derivs=0):
    buf = ctypes.create_string_buffer(256)
    capi._check(load_library().hqh_station_header(buf, 256, ctypes.c_int32(derivs)))
    return buf.value.decode()


def station_format(time, disp):
    """One station line as the reference prints it; disp of 3, 6 or 9 values (+ velocity, + acceleration)."""
    buf = ctypes.create_string_buffer(256)
    d = np.ascontiguousarray(disp, np.float64).reshape(-1)
    if len(d) == 3:
        capi._check(load_library().hqh_station_format(buf, 256, ctypes.c_double(time), d.ctypes.data_as(ctypes.c_void_p)))
    else:
        capi._check(load_library().hqh_station_format_derivs(buf, 256, ctypes.c_double(time),
                                                           d.ctypes.data_as(ctypes.c_void_p),
                                                           ctypes.c_int32(len(d) // 3 - 1)))
    return buf.value.decode()
