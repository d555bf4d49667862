"""Deterministic synthetic observation files (ctypes front end of csrc/mdc_gen.c).

The byte format is the one the reference's producer writes
(tools/dminingoracle.cpp:212-219, :243).  Seed 20240519 by default.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

SEED = 20240519
_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libmdc_gen.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(path)
        vp, sz, u64, i64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int64
        lib.mdc_gen_mind_xml.restype = sz
        lib.mdc_gen_mind_xml.argtypes = [vp, sz, u64, i64, i64, ctypes.POINTER(i64)]
        lib.mdc_gen_mind_csv.restype = sz
        lib.mdc_gen_mind_csv.argtypes = [vp, sz, u64, i64, i64, ctypes.c_char_p, ctypes.POINTER(i64)]
        lib.mdc_gen_code_xml.restype = sz
        lib.mdc_gen_code_xml.argtypes = [vp, sz, u64, i64, ctypes.POINTER(i64)]
        _lib = lib
    return _lib


def mind_xml(nbytes: int = 0, rows: int = 0, seed: int = SEED, first_row: int = 0):
    """One ZHOBTMIND file: either exactly `rows` rows or as many as fit in `nbytes`.

    Returns (uint8 array, rows)."""
    lib = _load()
    cap = nbytes if nbytes else rows * 256 + 1024
    buf = np.empty(cap + 1024 if rows else cap, dtype=np.uint8)
    n = ctypes.c_int64(0)
    used = lib.mdc_gen_mind_xml(buf.ctypes.data, buf.size, seed, first_row, rows, ctypes.byref(n))
    return buf[:used], int(n.value)


def mind_csv(nbytes: int = 0, rows: int = 0, seed: int = SEED, first_row: int = 0, sep: str = ","):
    lib = _load()
    cap = nbytes if nbytes else rows * 128 + 1024
    buf = np.empty(cap + 1024 if rows else cap, dtype=np.uint8)
    n = ctypes.c_int64(0)
    used = lib.mdc_gen_mind_csv(buf.ctypes.data, buf.size, seed, first_row, rows, sep.encode(), ctypes.byref(n))
    return buf[:used], int(n.value)


def code_xml(rows: int = 0, seed: int = SEED):
    lib = _load()
    buf = np.empty(839 * 512 + 1024, dtype=np.uint8)
    n = ctypes.c_int64(0)
    used = lib.mdc_gen_code_xml(buf.ctypes.data, buf.size, seed, rows, ctypes.byref(n))
    return buf[:used], int(n.value)
