"""TEST INFRASTRUCTURE -- ctypes view of oracle/libcdl_oracle.so (see cdl_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libcdl_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcdl_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def geom(N, C, M, dims, P, pad, stride):
    """15 ints: N,C,M, D,H,W, Pd,Ph,Pw, pd,ph,pw, sd,sh,sw (2-D: leading 1s / 0 pad / stride 1)."""
    def three(v, fill):
        v = list(v)
        return [fill] * (3 - len(v)) + v
    arr = [N, C, M] + three(dims, 1) + three(P, 1) + three(pad, 0) + three(stride, 1)
    return (ctypes.c_int * 15)(*arr)


def forward(yp, mask, wA, wB, tau, pad, stride):
    """yp (N,C,*sp); wA,wB (K,M,C,*P); tau (K,N,M) -> (z_K, xp) as numpy arrays."""
    yp = np.asarray(yp, dtype=np.float32)
    wA = np.asarray(wA, dtype=np.float32)
    K, M, C = wA.shape[:3]
    N = yp.shape[0]
    sp, P = yp.shape[2:], wA.shape[3:]
    stride = [stride] * len(sp) if isinstance(stride, int) else list(stride)
    zsp = tuple(d // s for d, s in zip(sp, stride))
    z = np.empty((N, M) + zsp, dtype=np.float32)
    xp = np.empty_like(yp)
    g = geom(N, C, M, sp, P, pad, stride)
    keep = [_f(yp), _f(wA), _f(wB), _f(tau)]
    mp = None
    if mask is not None:
        keep.append(_f(mask))
        mp = keep[-1][1]
    rc = lib().cdl_oracle_forward(K, g, keep[0][1], mp, keep[1][1], keep[2][1], keep[3][1],
                                  z.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                  xp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc:
        raise MemoryError("cdl_oracle_forward")
    return z, xp
