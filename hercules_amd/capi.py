"""ctypes binding of include/hq_solver.h (the stub a Python host would use).

Fails loudly when the native library is missing or no gfx950 device is present:
there is no fallback implementation.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("HQ_SOLVER_LIB") or os.path.join(_HERE, "csrc", "libhq_solver.so")   # (HQ_SOLVER_LIB: experiment builds, profiles/tools)

HQ_VARIANT_AUTO, HQ_VARIANT_SCATTER, HQ_VARIANT_PATCH = 0, 1, 2
IPC_BLOB_BYTES = 4096                   # HQ_IPC_BLOB_BYTES

EXPORTS = ["hq_device_count", "hq_last_error", "hq_create", "hq_destroy", "hq_get_info", "hq_get_info_sized", "hq_abi_version", "hq_comm_init_ipc_n",
           "hq_options_init", "hq_create_opts", "hq_get_options", "hq_real_bytes",
           "hq_comm_unique_id", "hq_comm_init", "hq_comm_selftest", "hq_group_link", "hq_group_run", "hq_set_source", "hq_run", "hq_sync", "hq_gather", "hq_gather3",
           "hq_download", "hq_upload", "hq_phase_force", "hq_phase_update", "hq_download_force",
           "hq_run_timed", "hq_dominant_kernel", "hq_plan_check", "hq_stencil_plan_check", "hq_check_finite",
           "hq_stencil_coefficients", "hq_brick_plan_check", "hq_brick_plan_check_n", "hq_comm_init_host",
           "hq_comm_ipc_export", "hq_comm_init_ipc", "hq_comm_init_loopback"]


class HqError(RuntimeError):
    pass


# hq_host_exchange_fn (include/hq_solver.h)
HOST_EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                    ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                    ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64),
                                    ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32)


class _Messenger(ctypes.Structure):
    _fields_ = [("procid", ctypes.c_int32), ("nodecount", ctypes.c_int32),
                ("mapping", ctypes.c_void_p)]


class _Schedule(ctypes.Structure):
    _fields_ = [("c_count", ctypes.c_int32), ("first_c", ctypes.POINTER(_Messenger)),
                ("s_count", ctypes.c_int32), ("first_s", ctypes.POINTER(_Messenger))]


class _Desc(ctypes.Structure):
    _fields_ = [("lenum", ctypes.c_int32), ("nharbored", ctypes.c_int32), ("ldnnum", ctypes.c_int32),
                ("lnid", ctypes.c_void_p), ("node_xyz", ctypes.c_void_p),
                ("dn_ldnid", ctypes.c_void_p), ("dn_ptr", ctypes.c_void_p), ("dn_lanid", ctypes.c_void_p),
                ("eTable", ctypes.c_void_p), ("nTable", ctypes.c_void_p),
                ("tm1", ctypes.c_void_p), ("tm2", ctypes.c_void_p),
                ("an_sched", _Schedule), ("dn_sched", _Schedule),
                ("deltaT", ctypes.c_double), ("rank", ctypes.c_int32), ("nranks", ctypes.c_int32),
                ("variant", ctypes.c_int32), ("reserved", ctypes.c_int32), ("node_gnid", ctypes.c_void_p),
                ("edata", ctypes.c_void_p), ("mat_bbase", ctypes.c_double), ("mat_threshold_damping", ctypes.c_double),
                ("mat_threshold_vpvs", ctypes.c_double)]


class _Info(ctypes.Structure):
    _fields_ = [("variant", ctypes.c_int32), ("npatches", ctypes.c_int32),
                ("patch_pairs", ctypes.c_int64), ("device_bytes", ctypes.c_int64),
                ("step", ctypes.c_int32), ("nranks", ctypes.c_int32),
                ("lattice_patches", ctypes.c_int32), ("stencil_patches", ctypes.c_int32),
                ("ragged_patches", ctypes.c_int32), ("brick_units", ctypes.c_int32),
                ("brick_nodes", ctypes.c_int64), ("brick_units_pernode", ctypes.c_int32),
                ("brick_units_het", ctypes.c_int32), ("pcie_h2d_bytes", ctypes.c_int64),
                ("pcie_d2h_bytes", ctypes.c_int64), ("transport", ctypes.c_int32), ("ipc_arena_coarse", ctypes.c_int32),
                ("ipc_arena_kind", ctypes.c_int32), ("debug_halo", ctypes.c_int32),
                ("brick_units_packed", ctypes.c_int32), ("brick_units_ragged", ctypes.c_int32),
                ("brick_stream", ctypes.c_int32), ("brick_units_ragged_het", ctypes.c_int32), ("timed_steps", ctypes.c_int64),
                ("t_step_us", ctypes.c_double), ("t_shell_us", ctypes.c_double), ("t_interior_us", ctypes.c_double),
                ("t_chain_us", ctypes.c_double), ("t_chain_exposed_us", ctypes.c_double)]


# hq_options (include/hq_solver.h): int32 fields in the header's order, two doubles, two more int32
OPTION_FIELDS = ["no_bricks", "brick_cz", "brick_minz", "brick_minnodes", "brick_no_het", "brick_no_ntsame", "brick_by_component",
                 "brick_stream", "brick_no_faces", "brick_half_tiles", "brick_no_pack", "patch_pipe", "patch_threads", "patch_pmax", "patch_pmerge", "patch_psplit", "patch_nlmax",
                 "patch_vmax", "patch_ragged", "patch_no_lattice", "patch_no_stencil", "patch_no_uniform", "patch_no_iso",
                 "patch_no_ntsame", "patch_no_dedup", "patch_wform", "patch_merge_rounds", "overlap", "no_overlap", "reserve_cus",
                 "cu_mask", "no_fused_share", "group_copies", "debug_halo", "ipc_arena"]


class Options(ctypes.Structure):
    """hq_options.  Options(brick_cz=16, debug_halo=1): every other field stays -1 = library default."""
    _fields_ = ([("size", ctypes.c_uint64)] + [(n, ctypes.c_int32) for n in OPTION_FIELDS] +
                [("ipc_timeout_ms", ctypes.c_double), ("loopback_delay_us", ctypes.c_double),
                 ("verbose", ctypes.c_int32), ("quiet", ctypes.c_int32),
                 ("brick_ragged", ctypes.c_int32), ("brick_ragged_minfill", ctypes.c_int32),
                 ("allow_env", ctypes.c_int32), ("brick_ragged_het", ctypes.c_int32), ("reserved0", ctypes.c_int32),
                 ("phase_timing", ctypes.c_int32)])

    def __init__(self, **kw):
        super().__init__()
        load_library().hq_options_init(ctypes.byref(self), ctypes.c_uint64(ctypes.sizeof(self)))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError("hq_options has no field %r" % k)
            setattr(self, k, v)

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "size"}


_lib = None
_lib_f32 = None
# the float build stays anchored in the package (HQ_SOLVER_LIB names an experiment build of the fp64 library only);
# HQ_SOLVER_LIB_F32 names an experiment build of the float one
_LIBPATH_F32 = os.environ.get("HQ_SOLVER_LIB_F32") or os.path.join(_HERE, "csrc", "libhq_solver_f32.so")


def load_library(path=None, precision="f64"):
    """dlopen libhq_solver.so; raises HqError if it was not built.
    precision="f32": libhq_solver_f32.so -- the same sources with -DHQ_SINGLE_PRECISION_SOLVER (hq_real = float: the
    reference's -DSINGLE_PRECISION_SOLVER, psolve.h:60-64); a separately named build, never the default."""
    global _lib, _lib_f32
    if precision == "f32" and path is None:
        if _lib_f32 is None:
            _lib_f32 = load_library(_LIBPATH_F32)
            if _lib_f32.hq_real_bytes() != 4:
                raise HqError("%s was not built with -DHQ_SINGLE_PRECISION_SOLVER" % _LIBPATH_F32)
        return _lib_f32
    if _lib is not None and path is None:
        return _lib
    p = path or _LIBPATH
    if not os.path.exists(p):
        raise HqError("native library %s is missing: run `python -m hercules_amd.build` "
                      "(there is no Python/CPU fallback)" % p)
    # libhq_solver.so and its _f32 build export the same names: only the fp64 one joins the global scope (libhq_host.so's
    # references must never bind to the float build, whichever was loaded first); both are linked -Bsymbolic, so calls
    # between their own entry points stay inside the library they belong to
    # (which of the two a file is, is asked of the file -- hq_real_bytes -- not read off its name)
    probe = ctypes.CDLL(p, mode=ctypes.RTLD_LOCAL)
    lib = probe if probe.hq_real_bytes() == 4 else ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
    lib.hq_last_error.restype = ctypes.c_char_p
    lib.hq_dominant_kernel.restype = ctypes.c_char_p
    lib.hq_dominant_kernel.argtypes = [ctypes.c_void_p]
    for name in EXPORTS:
        getattr(lib, name)          # AttributeError if the ABI is incomplete
    if path is None:
        _lib = lib
    return lib


def device_count():
    return int(load_library().hq_device_count())


def _check(rc, lib=None):
    if rc != 0:
        raise HqError("hq error %d: %s" % (rc, (lib or load_library()).hq_last_error().decode()))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _schedule(sched, keep):
    """sched = {"c": [(procid, mapping ndarray), ...], "s": [...]} or None."""
    out = _Schedule()
    if not sched:
        return out
    for key, cnt, first in (("c", "c_count", "first_c"), ("s", "s_count", "first_s")):
        items = sched.get(key, [])
        arr = (_Messenger * max(len(items), 1))()
        for i, (procid, mapping) in enumerate(items):
            m = np.ascontiguousarray(mapping, np.int32)
            keep.append(m)
            arr[i].procid = int(procid)
            arr[i].nodecount = len(m)
            arr[i].mapping = m.ctypes.data
        keep.append(arr)
        setattr(out, cnt, len(items))
        setattr(out, first, ctypes.cast(arr, ctypes.POINTER(_Messenger)))
    return out


class Solver:
    """One device-resident partition (hq_ctx)."""

    def __init__(self, lnid, etable, ntable, dt, tm1=None, tm2=None, node_xyz=None,
                 dangling=None, an_sched=None, dn_sched=None, rank=0, nranks=1,
                 variant=HQ_VARIANT_AUTO, device=0, options=None, edata=None, material=None, precision="f64"):
        """options: an Options (hq_options) or a dict of its fields; None = hq_create's defaults.
        precision: "f64", or "f32" = libhq_solver_f32.so (hq_real = float: ntable, tm1, tm2 and everything gathered or
        downloaded are float32 arrays, as the reference's -DSINGLE_PRECISION_SOLVER arrays are).
        edata [E,4] float32 (edgesize, Vp, Vs, rho as solver_init left them) + material = (bBase, threshold_damping,
        threshold_vpvs): hq_desc.edata / mat_* -- lets hq_k_brick_het keep 12 bytes per element."""
        lib = load_library(precision=precision)
        self.real = np.float32 if precision == "f32" else np.float64
        if lib.hq_real_bytes() != np.dtype(self.real).itemsize:
            raise HqError("library and precision %r disagree about sizeof(hq_real)" % precision)
        if isinstance(options, dict):
            options = Options(**options)
        keep = []
        lnid = np.ascontiguousarray(lnid, np.int32)
        etable = np.ascontiguousarray(etable, np.float64)
        ntable = np.ascontiguousarray(ntable, self.real)
        d = _Desc()
        d.lenum, d.nharbored = lnid.shape[0], ntable.shape[0]
        d.lnid, d.eTable, d.nTable = _ptr(lnid), _ptr(etable), _ptr(ntable)
        keep += [lnid, etable, ntable]
        if node_xyz is not None:
            node_xyz = np.ascontiguousarray(node_xyz, np.int32)
            keep.append(node_xyz)
            d.node_xyz = _ptr(node_xyz)
        for name, a in (("tm1", tm1), ("tm2", tm2)):
            if a is not None:
                a = np.ascontiguousarray(a, self.real)
                keep.append(a)
                setattr(d, name, _ptr(a))
        if dangling is not None:
            ids, ptr, anchors = [np.ascontiguousarray(x, np.int32) for x in dangling]
            keep += [ids, ptr, anchors]
            d.ldnnum = len(ids)
            d.dn_ldnid, d.dn_ptr, d.dn_lanid = _ptr(ids), _ptr(ptr), _ptr(anchors)
        if edata is not None:
            edata = np.ascontiguousarray(edata, np.float32)
            assert edata.shape == (lnid.shape[0], 4) and material is not None
            keep.append(edata)
            d.edata = _ptr(edata)
            d.mat_bbase, d.mat_threshold_damping, d.mat_threshold_vpvs = [float(v) for v in material]
        d.an_sched = _schedule(an_sched, keep)
        d.dn_sched = _schedule(dn_sched, keep)
        d.deltaT, d.rank, d.nranks, d.variant = dt, rank, nranks, variant
        self._h = ctypes.c_void_p()
        self.N, self.E = d.nharbored, d.lenum
        self._lib = lib
        if options is None:
            _check(lib.hq_create(ctypes.byref(d), ctypes.c_int(device), ctypes.byref(self._h)), lib)
        else:
            _check(lib.hq_create_opts(ctypes.byref(d), ctypes.c_int(device), ctypes.byref(options), ctypes.byref(self._h)), lib)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.hq_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    def info(self):
        i = _Info()
        _check(self._lib.hq_get_info_sized(self._h, ctypes.byref(i), ctypes.c_uint64(ctypes.sizeof(i))), self._lib)
        return {k: getattr(i, k) for k, _ in _Info._fields_}

    def options(self):
        """hq_get_options: what the context runs with, as resolved at hq_create_opts."""
        o = Options()
        _check(self._lib.hq_get_options(self._h, ctypes.byref(o), ctypes.c_uint64(ctypes.sizeof(o))), self._lib)
        return o.as_dict()

    def set_source(self, loaded_lnid, forces, step0=0):
        ids = np.ascontiguousarray(loaded_lnid, np.int32)
        F = np.ascontiguousarray(forces, np.float64)
        nsteps = F.shape[0] if len(ids) else 0
        _check(self._lib.hq_set_source(self._h, ctypes.c_int32(len(ids)), _ptr(ids),
                                       ctypes.c_int32(step0), ctypes.c_int32(nsteps), _ptr(F)))

    def comm_init(self, id128):
        buf = (ctypes.c_char * 128).from_buffer_copy(bytes(id128))
        _check(self._lib.hq_comm_init(self._h, buf), self._lib)

    def comm_init_host(self, exchange):
        """hq_comm_init_host: `exchange(recvs, sends, tag)` is called at every halo exchange with
        recvs = [(peer, float64 array to fill)], sends = [(peer, float64 array)] -- views of the engine's pinned
        staging buffers -- and must return when everything has arrived (the caller's MPI / gloo / ...)."""
        import numpy as np

        def view(ptr, n):
            return np.ctypeslib.as_array(ctypes.cast(ctypes.c_void_p(ptr), ctypes.POINTER(ctypes.c_double)), shape=(int(n),))

        def trampoline(user, nrecv, rpeer, rcount, rbuf, nsend, speer, scount, sbuf, tag):
            try:
                exchange([(rpeer[i], view(rbuf[i], rcount[i])) for i in range(nrecv)],
                         [(speer[i], view(sbuf[i], scount[i])) for i in range(nsend)], tag)
                return 0
            except Exception:                       # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._host_exchange = HOST_EXCHANGE_FN(trampoline)          # keep the thunk alive with the context
        _check(self._lib.hq_comm_init_host(self._h, self._host_exchange, None), self._lib)

    def comm_ipc_export(self):
        """hq_comm_ipc_export: this rank's HQ_IPC_BLOB_BYTES blob (bytes) for the host to all-gather."""
        buf = (ctypes.c_char * IPC_BLOB_BYTES)()
        _check(self._lib.hq_comm_ipc_export(self._h, buf), self._lib)
        return bytes(buf)

    def comm_init_ipc(self, blobs):
        """hq_comm_init_ipc: `blobs` = every rank's export, in rank order (list of bytes or one bytes object)."""
        raw = blobs if isinstance(blobs, (bytes, bytearray)) else b"".join(blobs)
        if len(raw) % IPC_BLOB_BYTES:
            raise HqError("comm_init_ipc: %d bytes is not a whole number of %d-byte blobs" % (len(raw), IPC_BLOB_BYTES))
        buf = (ctypes.c_char * len(raw)).from_buffer_copy(bytes(raw))
        _check(self._lib.hq_comm_init_ipc_n(self._h, buf, ctypes.c_int32(len(raw) // IPC_BLOB_BYTES)), self._lib)

    def comm_init_loopback(self):
        _check(self._lib.hq_comm_init_loopback(self._h), self._lib)

    def comm_selftest(self, count=1024):
        _check(self._lib.hq_comm_selftest(self._h, ctypes.c_int32(count)), self._lib)

    def run(self, nsteps):
        _check(self._lib.hq_run(self._h, ctypes.c_int32(nsteps)), self._lib)

    def sync(self):
        _check(self._lib.hq_sync(self._h), self._lib)

    def check_finite(self):
        """Count of NaN / infinite values in tm1, tm2 (solver_check_nan, psolve.c:3769-3782)."""
        n = ctypes.c_int64()
        _check(self._lib.hq_check_finite(self._h, ctypes.byref(n)), self._lib)
        return int(n.value)

    def run_timed(self, nsteps):
        tot, ker = ctypes.c_double(), ctypes.c_double()
        _check(self._lib.hq_run_timed(self._h, ctypes.c_int32(nsteps), ctypes.byref(tot), ctypes.byref(ker)), self._lib)
        return tot.value, ker.value

    def dominant_kernel(self):
        return self._lib.hq_dominant_kernel(self._h).decode()

    def download(self, want_tm2=True):
        """(tm1, tm2) = u(step dt), u((step - 1) dt); tm2 is None with want_tm2=False."""
        tm1 = np.empty((self.N, 3), self.real)
        tm2 = np.empty((self.N, 3), self.real) if want_tm2 else None
        _check(self._lib.hq_download(self._h, _ptr(tm1), _ptr(tm2)), self._lib)
        return tm1, tm2

    def upload(self, tm1, tm2, step):
        a = np.ascontiguousarray(tm1, self.real)
        b = np.ascontiguousarray(tm2, self.real)
        _check(self._lib.hq_upload(self._h, _ptr(a), _ptr(b), ctypes.c_int32(step)), self._lib)

    def gather(self, lnid):
        ids = np.ascontiguousarray(np.asarray(lnid).reshape(-1), np.int32)
        o1 = np.empty((len(ids), 3), self.real)
        o2 = np.empty((len(ids), 3), self.real)
        _check(self._lib.hq_gather(self._h, ctypes.c_int32(len(ids)), _ptr(ids), _ptr(o1), _ptr(o2)), self._lib)
        return o1, o2

    def gather3(self, lnid):
        """tm1, tm2 and tm3 = u((step - 2) dt) at the given nodes (patch variant)."""
        ids = np.ascontiguousarray(np.asarray(lnid).reshape(-1), np.int32)
        o = [np.empty((len(ids), 3), self.real) for _ in range(3)]
        _check(self._lib.hq_gather3(self._h, ctypes.c_int32(len(ids)), _ptr(ids), _ptr(o[0]), _ptr(o[1]), _ptr(o[2])), self._lib)
        return tuple(o)

    def phase_force(self):
        _check(self._lib.hq_phase_force(self._h), self._lib)

    def phase_update(self):
        _check(self._lib.hq_phase_update(self._h), self._lib)

    def download_force(self):
        f = np.empty((self.N, 3))
        _check(self._lib.hq_download_force(self._h, _ptr(f)), self._lib)
        return f


PLAN_REPORT = ("patches", "lattice_patches", "pairs", "distinct_row_blocks", "gather_passes", "gather_instructions",
               "lattice_gather_passes", "faults")


def plan_check(desc):
    """hq_plan_check on a filled _Desc: the planner's host-only self-check (no device needed)."""
    rep = (ctypes.c_int64 * 8)()
    _check(load_library().hq_plan_check(ctypes.byref(desc), rep))
    return dict(zip(PLAN_REPORT, [int(v) for v in rep]))


STENCIL_PLAN_REPORT = ("patches", "tables", "full_lattices", "boundary_nodes", "corners_checked", "faults")


def stencil_plan_check(desc):
    """hq_stencil_plan_check on a filled _Desc: the stencil kernel's tables against the mesh (no device needed)."""
    rep = (ctypes.c_int64 * 6)()
    _check(load_library().hq_stencil_plan_check(ctypes.byref(desc), rep))
    return dict(zip(STENCIL_PLAN_REPORT, [int(v) for v in rep]))


BRICK_PLAN_REPORT = ("brick_nodes", "columns", "units", "units_one_nt_row", "het_units", "neighbours_checked",
                     "patch_nodes", "faults", "ragged_units", "ragged_nodes", "ragged_het_units", "ragged_het_nodes")


def brick_plan_check(desc):
    """hq_brick_plan_check on a filled _Desc: the brick planner against the mesh's connectivity (no device needed)."""
    rep = (ctypes.c_int64 * len(BRICK_PLAN_REPORT))()
    _check(load_library().hq_brick_plan_check_n(ctypes.byref(desc), rep, ctypes.c_int32(len(BRICK_PLAN_REPORT))))
    return dict(zip(BRICK_PLAN_REPORT, [int(v) for v in rep]))


def comm_unique_id():
    buf = (ctypes.c_char * 128)()
    _check(load_library().hq_comm_unique_id(buf))
    return bytes(buf)


def group_link(solvers):
    """hq_group_link: solvers[i] must hold rank i of len(solvers)."""
    arr = (ctypes.c_void_p * len(solvers))(*[s._h for s in solvers])
    lib = solvers[0]._lib                       # (the group's library: libhq_solver.so or its _f32 build)
    _check(lib.hq_group_link(arr, ctypes.c_int32(len(solvers))), lib)


def group_run(solvers, nsteps):
    arr = (ctypes.c_void_p * len(solvers))(*[s._h for s in solvers])
    lib = solvers[0]._lib
    _check(lib.hq_group_run(arr, ctypes.c_int32(len(solvers)), ctypes.c_int32(nsteps)), lib)
    for s in solvers:
        s.sync()
