"""Helpers shared by the parity tests."""
import ctypes as C

import numpy as np
import torch


def rng_field(rng, n1, n2, scale=1.0):
    return np.asfortranarray(rng.standard_normal((n1, n2)) * scale)


class Dev:
    """Dense column-major device copy of a host (n1,n2) array, optionally pitched."""

    def __init__(self, host: np.ndarray, ld: int = 0, pad_value: float = 0.0):
        self.n1, self.n2 = host.shape
        self.ld = ld or self.n1
        self.t = torch.full((self.n2, self.ld), pad_value, dtype=torch.float64, device="cuda:0")
        self.t[:, : self.n1].copy_(torch.from_numpy(np.ascontiguousarray(host.T)))

    @property
    def ptr(self):
        return C.c_void_p(self.t.data_ptr())

    def ptr_at(self, ix0: int, iy0: int):
        """pointer to 0-based element (ix0, iy0)"""
        return C.c_void_p(self.t.data_ptr() + 8 * (iy0 * self.ld + ix0))

    def get(self) -> np.ndarray:
        torch.cuda.synchronize()
        return np.asfortranarray(self.t[:, : self.n1].cpu().numpy().T)


def same_bits(a: np.ndarray, b: np.ndarray) -> bool:
    """Bitwise equality except that +0.0 == -0.0 and any NaN == any NaN."""
    return bool(np.array_equal(a, b, equal_nan=True))


def strict_bits(a: np.ndarray, b: np.ndarray) -> bool:
    """Equal bit patterns (the sign of a zero counts), NaNs in the same places."""
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a.view(np.uint64)[~na], b.view(np.uint64)[~nb]))


def rel_l2(a, b) -> float:
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return float(d / n) if n > 0 else float(d)
