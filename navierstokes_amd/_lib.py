"""ctypes binding of libnsk.so (the C ABI declared in include/nsk.h).

There is no CPU fallback: importing this module without the built library, or
calling into it without a GPU, raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NSK_LIB") or os.path.join(_HERE, "libnsk.so")   # NSK_LIB: another build of the same ABI

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class NskError(RuntimeError):
    """A nonzero return of the library; `code` is the NSK_ERR_* value of include/nsk.h."""

    def __init__(self, message: str, code: int = 0):
        super().__init__(message)
        self.code = code


ERR_HIP, ERR_ARGUMENT, ERR_FRONT_CAPACITY, ERR_FRONT_OUTSIDE, ERR_TIMEOUT = 1, 2, 3, 4, 5


class CNSParams(C.Structure):
    """struct nsk_ns_params (include/nsk.h)."""
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("iy0", C.c_int), ("nyg", C.c_int),
        ("niter", C.c_int), ("nchk", C.c_int), ("full_rows", C.c_int),
        ("ld_pr", C.c_int), ("ld_dp", C.c_int), ("dp_embedded", C.c_int),
        ("lx", C.c_double), ("ly", C.c_double), ("rho", C.c_double), ("vin", C.c_double),
        ("psc", C.c_double), ("mu", C.c_double), ("g", C.c_double), ("a2", C.c_double),
        ("b2", C.c_double), ("ox", C.c_double), ("oy", C.c_double), ("sinb", C.c_double),
        ("cosb", C.c_double), ("eps_it", C.c_double), ("dx", C.c_double), ("dy", C.c_double),
        ("dt", C.c_double), ("damp", C.c_double), ("dtau", C.c_double),
    ]


class CNSState(C.Structure):
    """struct nsk_ns_state (include/nsk.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "Pr", "dPrdtau", "C", "C_o", "txx", "tyy", "txy", "Vx", "Vy", "Vx_o", "Vy_o",
        "divV", "Rp", "Vprof")]


class CBubParams(C.Structure):
    """struct nsk_bub_params (include/nsk.h)."""
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("maxit", C.c_int)] + \
               [(n, C.c_double) for n in ("Lx", "Ly", "gx", "gy", "rho1", "rho2", "m1", "m2", "sigma", "rro", "usouth",
                                          "unorth", "vwest", "veast", "dt", "dx", "dy", "maxError", "beta")]


class CDnsState(C.Structure):
    """struct nsk_bub_dns_state (include/nsk.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("u", "v", "p", "ut", "vt", "tmp1", "tmp2", "r", "ro", "rt", "uu", "vv")]


class CCode3State(C.Structure):
    """struct nsk_bub_code3_state (include/nsk.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("u", "v", "p", "ut", "vt", "tmp1", "tmp2", "fx", "fy", "un", "vn", "r", "m",
                                          "rn", "mn", "rt", "xf", "yf", "xfn", "yfn", "uf", "vf", "tx", "ty", "xs", "ys")] + \
               [("Nf", C.c_int), ("cap", C.c_int)]


_D, _I, _P, _S = C.c_double, C.c_int, C.c_void_p, C.c_size_t

# name -> argtypes (after the leading context pointer); all return int
_SIGS = {
    "nsk_update_tau": [_P, _P, _P, _P, _P, _D, _D, _D, _I, _I],
    "nsk_predict_V": [_P, _P, _P, _P, _P, _D, _D, _D, _D, _D, _I, _I],
    "nsk_set_sphere": [_P, _P, _P] + [_D] * 10 + [_I, _I, _I],
    "nsk_update_divV": [_P, _P, _P, _D, _D, _I, _I],
    "nsk_update_divV_ld": [_P, _I, _P, _P, _D, _D, _I, _I],
    "nsk_correct_V_ld": [_P, _P, _P, _I, _D, _D, _D, _D, _I, _I],
    "nsk_update_dPrdtau": [_P, _P, _P, _D, _D, _D, _D, _D, _D, _I, _I],
    "nsk_update_Pr": [_P, _P, _D, _I, _I],
    "nsk_set_bc_Pr": [_P, _D, _I, _I, _I, _I, _I],
    "nsk_compute_res": [_P, _P, _P, _D, _D, _D, _D, _I, _I],
    "nsk_maxabs": [_P, _S, c_double_p],
    "nsk_correct_V": [_P, _P, _P, _D, _D, _D, _D, _I, _I],
    "nsk_set_bc_Vel": [_P, _P, _P, _I, _I, _I, _I, _I],
    "nsk_advect": [_P, _P, _P, _P, _P, _P, _D, _D, _D, _I, _I, _I, _I],
    "nsk_copy": [_P, _P, _S],
    "nsk_upload": [_P, _P, _S],
    "nsk_download": [_P, _P, _S],
    "nsk_free": [_P],
    "nsk_zeros": [_I, _I, C.POINTER(_P)],
    "nsk_set_stream": [_P],
    "nsk_use_own_stream": [],
    "nsk_set_option": [C.c_char_p, _I],
    "nsk_get_option": [C.c_char_p, c_int_p],
    "nsk_sync": [],
    # bubble / drop solvers
    "nsk_bub_set_density": [_P, _P, _D, _D, _P, _P, _D, _D, _D, _I, _I],
    "nsk_bub_set_tangential": [_P, _P, _D, _D, _D, _D, _I, _I],
    "nsk_bub_temp_vel": [_P, _P, _P, _P, _P, _D, _D, _D, _D, _D, _D, _I, _I],
    "nsk_bub_temp_vel_adv": [_P, _P, _P, _P, _P, _D, _D, _D, _P, _P, _D, _D, _D, _I, _I],
    "nsk_bub_temp_vel_diff": [_P, _P, _P, _P, _P, _D, _D, _D, _P, _I, _I],
    "nsk_bub_set_boundary": [_P, _P, _D, _I, _I],
    "nsk_bub_compute_sources": [_P, _P, _P, _P, _P, _D, _D, _D, _I, _I],
    "nsk_bub_solve_pressure": [_P, _P, _P, _P, _D, _D, _D, _I, _I, _I, _D, c_int_p, c_double_p],
    "nsk_bub_solve_pressure_rb": [_P, _P, _P, _P, _D, _D, _D, _I, _I, _I, _D, c_int_p, c_double_p],
    "nsk_bub_pt_pressure": [_P, _P, _P, _P, _D, _D, _D, _D, _I, _I, _I, _D, c_int_p, c_double_p],
    "nsk_bub_correct_vel": [_P, _P, _P, _P, _P, _P, _D, _D, _D, _I, _I],
    "nsk_bub_advect_density": [_P, _P, _P, _P, _D, _D, _D, _D, _I, _I],
    "nsk_bub_velocity": [_P, _P, _P, _P, _I, _I],
    "nsk_bub_surface_tension": [_P, _P, _P, _P, _P, _P, _D, _D, _D, _I, _I, _I],
    "nsk_bub_advect_move_front": [_P, _P, _P, _P, _P, _P, _D, _D, _D, _D, _D, _I, _I, _I, c_int_p],
    "nsk_bub_distribute": [_P, _P, _P, _P, _D, _D, _D, _D, _I, _I, _I],
    "nsk_bub_construct_den": [_P, _P, _P, _D, _I, _D, _D, _D, _I, _I, c_int_p],
    "nsk_bub_update_visco": [_P, _P, _D, _D, _D, _D, _I, _I],
    "nsk_bub_average": [_P, _P, _S],
    "nsk_bub_regrid_front": [_P, _P, _P, _P, _D, _D, _I, _I, c_int_p],
    "nsk_ns_poisson_iters": [_P, _P, _P, C.POINTER(CNSParams), _I],
    "nsk_bub_dns_step": [C.POINTER(CDnsState), C.POINTER(CBubParams), C.POINTER(C.c_int)],
    "nsk_bub_code3_step": [C.POINTER(CCode3State), C.POINTER(CBubParams), C.POINTER(C.c_int)],
    "nsk_bub_dns_run": [C.POINTER(CDnsState), C.POINTER(CBubParams), _I, C.POINTER(C.c_int)],
    "nsk_bub_code3_run": [C.POINTER(CCode3State), C.POINTER(CBubParams), _I, C.POINTER(C.c_int)],
    "nsk_bub_code3_run_status": [c_int_p],
    "nsk_ns_poisson_iters_static": [_P, _P, _P, C.POINTER(CNSParams), _I],
    "nsk_ns_poisson_iters_edges_first": [_P, _P, _P, C.POINTER(CNSParams), _I, _I, _I, _I, _I, _P],
    "nsk_ns_residual": [_P, _P, _P, C.POINTER(CNSParams), _I, _I, c_double_p],
    "nsk_ns_poisson": [_P, _P, _P, _P, C.POINTER(CNSParams), c_int_p, c_double_p, c_int_p, _I, c_int_p],
    "nsk_ns_step": [C.POINTER(CNSState), C.POINTER(CNSParams), c_int_p, c_double_p, c_int_p, _I, c_int_p],
}

# slab-group entry points (after the leading group pointer)
_SLAB_SIGS = {
    "nsk_slabs_destroy": [],
    "nsk_slabs_window": [_I] + [c_int_p] * 7,
    "nsk_slabs_fields": [_I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)],
    "nsk_slabs_upload": [_I, _P],
    "nsk_slabs_download": [_I, _P],
    "nsk_slabs_exchange": [_I],
    "nsk_slabs_poisson_iters": [_I],
    "nsk_slabs_residual": [c_double_p],
    "nsk_slabs_poisson": [c_int_p, c_double_p, c_int_p, _I, c_int_p],
    "nsk_slabs_sync": [],
    "nsk_slabs_init_flow": [_P],
    "nsk_slabs_step": [c_int_p, c_double_p, c_int_p, _I, c_int_p],
    "nsk_slabs_rccl_self_exchange": [_I],
    "nsk_slabs_info": [c_int_p] * 4,
    "nsk_slabs_set_option": [C.c_char_p, _I],
    "nsk_slabs_get_option": [C.c_char_p, c_int_p],
}

_lib = None


def load() -> C.CDLL:
    """Load libnsk.so; raises NskError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NskError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C navierstokes_amd/csrc` (there is no CPU fallback)")
    # PyTorch ships its own HIP runtime; libnsk must bind to that same instance (streams and
    # device pointers are shared with torch), so torch has to be loaded before the library.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.nsk_last_error.restype = C.c_char_p
    lib.nsk_version.restype = C.c_char_p
    lib.nsk_create.argtypes = [_I, C.POINTER(_P)]
    lib.nsk_create.restype = _I
    lib.nsk_destroy.argtypes = [_P]
    lib.nsk_destroy.restype = _I
    lib.nsk_poisson_plan.argtypes = [_I] * 8 + [c_int_p]
    lib.nsk_poisson_plan.restype = _I
    lib.nsk_division_form.argtypes = [_D, c_int_p, c_double_p, c_double_p]
    lib.nsk_division_form.restype = _I
    lib.nsk_rccl_unique_id.argtypes = [_P]
    lib.nsk_rccl_unique_id.restype = _I
    lib.nsk_slabs_create.argtypes = [_I, _I, _I, c_int_p, C.POINTER(CNSParams), _I, _I, _I, _P, C.POINTER(_P)]
    lib.nsk_slabs_create.restype = _I
    for name, args in _SLAB_SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = [_P] + args
        fn.restype = _I
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = [_P] + args
        fn.restype = _I
    _lib = lib
    return lib


def poisson_kernel_hash() -> str:
    """Identifies the build of the multi-pass Poisson kernel (its source, the instantiation file and
    the compiler flags): counter measurements are stored with it and ignored when it has changed."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for name in ("ns_poisson_kernel.h", "ns_poisson_inst.hip", "Makefile"):
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def exported_names() -> list[str]:
    return ["nsk_create", "nsk_destroy", "nsk_last_error", "nsk_version", "nsk_poisson_plan",
            "nsk_division_form", "nsk_rccl_unique_id", "nsk_slabs_create"] + list(_SIGS) + list(_SLAB_SIGS)


def poisson_plan(nx: int, ny: int, num_simds: int = 1024, staged: bool = True, k: int = 0, k_limit: int = 8,
                 rows: int = 0, edge_percent: int = 0) -> dict:
    """nsk_poisson_plan: the launch geometry the fused Poisson entry points would choose (host only)."""
    out = (C.c_int * 6)()
    lib = load()
    if lib.nsk_poisson_plan(nx, ny, num_simds, int(staged), k, k_limit, rows, edge_percent, out) != 0:
        raise NskError(lib.nsk_last_error().decode())
    return dict(zip(("k", "rows", "rows_edge", "strips", "tasks", "est_ns_per_pass"), out))


def division_form(d: float) -> tuple[int, float, float]:
    """nsk_division_form: (form, rh, rl) for the divisor d (host only)."""
    form, rh, rl = C.c_int(0), C.c_double(0.0), C.c_double(0.0)
    lib = load()
    if lib.nsk_division_form(float(d), C.byref(form), C.byref(rh), C.byref(rl)) != 0:
        raise NskError(lib.nsk_last_error().decode())
    return form.value, rh.value, rl.value


def rccl_unique_id() -> bytes:
    """nsk_rccl_unique_id: 128 bytes to hand to every process of a slab group."""
    buf = C.create_string_buffer(128)
    lib = load()
    if lib.nsk_rccl_unique_id(buf) != 0:
        raise NskError(lib.nsk_last_error().decode())
    return buf.raw


class Slabs:
    """nsk_slabs: the slabs of this process (include/nsk.h).  transport: 0 auto, 1 RCCL, 2 device copies."""

    FIELDS = {"Pr": 0, "dPrdtau": 1, "divV": 2, "Vx": 3, "Vy": 4, "C": 5}

    def __init__(self, world, first_rank, devices, cparams: CNSParams, ny_loc, halo, transport=0, unique_id=None):
        self.lib = load()
        h = _P()
        dev = (C.c_int * len(devices))(*devices)
        uid = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        rc = self.lib.nsk_slabs_create(world, first_rank, len(devices), dev, C.byref(cparams), ny_loc, halo, transport,
                                       uid, C.byref(h))
        if rc != 0:
            raise NskError(self.lib.nsk_last_error().decode())
        self.handle, self.nlocal = h, len(devices)

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.handle, *args)
        if rc != 0:
            raise NskError(f"{name}: {self.lib.nsk_last_error().decode()}", rc)

    def window(self, local=0) -> dict:
        v = [C.c_int(0) for _ in range(7)]
        self.call("nsk_slabs_window", local, *[C.byref(x) for x in v])
        return dict(zip(("rank", "device", "iy0", "ny_win", "own_lo", "own_hi", "ld"), (x.value for x in v)))

    def fields(self, local=0) -> dict:
        p = [_P() for _ in range(3)]
        self.call("nsk_slabs_fields", local, *[C.byref(x) for x in p])
        return dict(zip(("Pr", "dPrdtau", "divV"), (x.value for x in p)))

    def upload(self, field: str, host) -> None:
        self.call("nsk_slabs_upload", self.FIELDS[field], host.ctypes.data_as(_P))

    def download(self, field: str, host) -> None:
        self.call("nsk_slabs_download", self.FIELDS[field], host.ctypes.data_as(_P))

    def exchange(self, mask=3) -> None:
        self.call("nsk_slabs_exchange", mask)

    def poisson_iters(self, count) -> None:
        self.call("nsk_slabs_poisson_iters", int(count))

    def residual(self) -> float:
        m = C.c_double(0.0)
        self.call("nsk_slabs_residual", C.byref(m))
        return m.value

    def poisson(self, cap=1024):
        iters, nh = C.c_int(0), C.c_int(0)
        he, hi = (C.c_double * cap)(), (C.c_int * cap)()
        self.call("nsk_slabs_poisson", C.byref(iters), he, hi, cap, C.byref(nh))
        n = min(nh.value, cap)
        return iters.value, list(hi[:n]), list(he[:n])

    def init_flow(self, vprof) -> None:
        self.call("nsk_slabs_init_flow", vprof.ctypes.data_as(_P))

    def step(self, cap=1024):
        """nsk_slabs_step: (iterations, iter history, err history) of one time step."""
        iters, nh = C.c_int(0), C.c_int(0)
        he, hi = (C.c_double * cap)(), (C.c_int * cap)()
        self.call("nsk_slabs_step", C.byref(iters), he, hi, cap, C.byref(nh))
        n = min(nh.value, cap)
        return iters.value, list(hi[:n]), list(he[:n])

    def sync(self) -> None:
        self.call("nsk_slabs_sync")

    def set_option(self, name: str, value: int) -> None:
        self.call("nsk_slabs_set_option", name.encode(), int(value))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        self.call("nsk_slabs_get_option", name.encode(), C.byref(v))
        return v.value

    def info(self) -> dict:
        v = [C.c_int(0) for _ in range(4)]
        self.call("nsk_slabs_info", *[C.byref(x) for x in v])
        return dict(zip(("world", "nlocal", "transport", "exchanges"), (x.value for x in v)))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.nsk_slabs_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One nsk_ctx bound to a HIP device; methods are the C entry points without the nsk_ prefix."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load()
        h = _P()
        rc = self.lib.nsk_create(int(device), C.byref(h))
        if rc != 0:
            raise NskError(self.lib.nsk_last_error().decode())
        self.handle = h
        if stream is not None:
            self.call("nsk_set_stream", _P(stream))

    def call(self, name: str, *args) -> None:
        rc = getattr(self.lib, name)(self.handle, *args)
        if rc != 0:
            raise NskError(f"{name}: {self.lib.nsk_last_error().decode()}", rc)

    def __getattr__(self, name):
        if name.startswith("_") or name in ("set_option", "get_option") or ("nsk_" + name) not in _SIGS:
            raise AttributeError(name)
        full = "nsk_" + name
        return lambda *a: self.call(full, *a)

    def set_option(self, name: str, value: int) -> None:
        self.call("nsk_set_option", name.encode(), int(value))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        self.call("nsk_get_option", name.encode(), C.byref(v))
        return v.value

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.nsk_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
