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
    pass


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
    "nsk_ns_poisson_iters_static": [_P, _P, _P, C.POINTER(CNSParams), _I],
    "nsk_ns_residual": [_P, _P, _P, C.POINTER(CNSParams), _I, _I, c_double_p],
    "nsk_ns_poisson": [_P, _P, _P, _P, C.POINTER(CNSParams), c_int_p, c_double_p, c_int_p, _I, c_int_p],
    "nsk_ns_step": [C.POINTER(CNSState), C.POINTER(CNSParams), c_int_p, c_double_p, c_int_p, _I, c_int_p],
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
            "nsk_division_form"] + list(_SIGS)


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
            raise NskError(f"{name}: {self.lib.nsk_last_error().decode()}")

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
