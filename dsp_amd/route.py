"""ctypes front end of include/mdc_route.h: routing entries of the parameter file, file-name
matching, SQL text and bind order (reference tools/xmltodb.cpp:102-129, :311-477; matchstr,
public/_public.cpp:213-263).  Host code of libmdc_parse.so; no GPU is needed."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

from . import capi
from .schema import Col, c_cols

# every symbol include/mdc_route.h declares
ROUTE_EXPORTS = [
    "mdc_route_load", "mdc_route_load_buffer", "mdc_route_free", "mdc_route_count", "mdc_route_truncated",
    "mdc_route_entry", "mdc_route_find", "mdc_matchstr", "mdc_crtsql", "mdc_bind_plan",
    "mdc_batch_bind_slabs", "mdc_ingest_rows",
]


class CBindSlab(ctypes.Structure):
    """mdc_bind_slab: one bound column as fixed-stride NUL-terminated slots."""

    _fields_ = [("base", ctypes.c_void_p), ("stride", ctypes.c_size_t), ("col", ctypes.c_int), ("pad", ctypes.c_int)]


class CIngestCounts(ctypes.Structure):
    _fields_ = [("totalcount", ctypes.c_int64), ("inscount", ctypes.c_int64), ("uptcount", ctypes.c_int64),
                ("insfailed", ctypes.c_int64), ("uptfailed", ctypes.c_int64), ("duplicates", ctypes.c_int64),
                ("result", ctypes.c_int), ("fatalcode", ctypes.c_int)]


ARRAY_EXEC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(CBindSlab), ctypes.c_int,
                              ctypes.c_int64, ctypes.POINTER(ctypes.c_int))


class CXmlToTable(ctypes.Structure):
    """C layout of one routing entry (st_xmltotable, reference tools/xmltodb.cpp:35-41)."""

    _fields_ = [
        ("filename", ctypes.c_char * 128),
        ("tname", ctypes.c_char * 32),
        ("uptbz", ctypes.c_int),
        ("execsql", ctypes.c_char * 256),
    ]


_ready = False


def _lib():
    global _ready
    L = capi.lib()
    if not _ready:
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        pp = ctypes.POINTER(vp)
        L.mdc_route_load.argtypes = [ctypes.c_char_p, pp]
        L.mdc_route_load_buffer.argtypes = [vp, sz, pp]
        L.mdc_route_free.argtypes = [vp]
        L.mdc_route_free.restype = None
        L.mdc_route_count.argtypes = [vp]
        L.mdc_route_truncated.argtypes = [vp]
        L.mdc_route_entry.argtypes = [vp, ci]
        L.mdc_route_entry.restype = ctypes.POINTER(CXmlToTable)
        L.mdc_route_find.argtypes = [vp, ctypes.c_char_p]
        L.mdc_matchstr.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.mdc_crtsql.argtypes = [ctypes.POINTER(CXmlToTable), vp, ci, ctypes.c_char_p, sz, ctypes.POINTER(sz),
                                 ctypes.c_char_p, sz, ctypes.POINTER(sz)]
        L.mdc_bind_plan.argtypes = [ctypes.POINTER(CXmlToTable), vp, ci, vp, ctypes.POINTER(ci), vp,
                                    ctypes.POINTER(ci)]
        L.mdc_batch_bind_slabs.argtypes = [vp, ctypes.POINTER(CXmlToTable), vp, ci, ctypes.POINTER(CBindSlab),
                                           ctypes.POINTER(ci), ctypes.POINTER(CBindSlab), ctypes.POINTER(ci)]
        L.mdc_ingest_rows.argtypes = [ctypes.POINTER(CBindSlab), ci, ctypes.POINTER(CBindSlab), ci, ctypes.c_int64, ci,
                                      ARRAY_EXEC, vp, ctypes.POINTER(CIngestCounts)]
        _ready = True
    return L


def _enc(x) -> bytes:
    return x.encode() if isinstance(x, str) else bytes(x)


def matchstr(name, rules) -> bool:
    return bool(_lib().mdc_matchstr(_enc(name), _enc(rules)))


def entry(filename=b"", tname=b"", uptbz=0, execsql=b"") -> CXmlToTable:
    e = CXmlToTable()
    e.filename, e.tname, e.uptbz, e.execsql = _enc(filename), _enc(tname), int(uptbz), _enc(execsql)
    return e


class Route:
    """The entries of a parameter file (loadxmltotable) and the lookup by file name."""

    def __init__(self, data: Optional[bytes] = None, path: Optional[str] = None):
        self.h = None
        h = ctypes.c_void_p()
        if path is not None:
            rc = _lib().mdc_route_load(_enc(path), ctypes.byref(h))
        else:
            data = bytes(data or b"")
            rc = _lib().mdc_route_load_buffer(data, len(data), ctypes.byref(h))
        if rc != capi.MDC_OK:
            raise capi.MdcError(rc, _lib().mdc_last_error().decode(errors="replace"))
        self.h = h

    def close(self):
        if self.h:
            _lib().mdc_route_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __len__(self) -> int:
        return int(_lib().mdc_route_count(self.h))

    @property
    def truncated(self) -> int:
        return int(_lib().mdc_route_truncated(self.h))

    def entries(self) -> List[Tuple[bytes, bytes, int, bytes]]:
        out = []
        for i in range(len(self)):
            e = _lib().mdc_route_entry(self.h, i).contents
            out.append((e.filename, e.tname, e.uptbz, e.execsql))
        return out

    def raw(self, i: int) -> CXmlToTable:
        p = _lib().mdc_route_entry(self.h, i)
        if not p:
            raise IndexError(i)
        return p.contents

    def find(self, xmlfile) -> int:
        return int(_lib().mdc_route_find(self.h, _enc(xmlfile)))


def crtsql(e: CXmlToTable, cols: Sequence[Col], cap: int = 1 << 16):
    """(insertsql, updatesql or None) as bytes."""
    cc = c_cols(cols)
    ins, upd = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(cap)
    il, ul = ctypes.c_size_t(0), ctypes.c_size_t(0)
    capi.check(_lib().mdc_crtsql(ctypes.byref(e), cc, len(cols), ins, cap, ctypes.byref(il), upd, cap, ctypes.byref(ul)))
    return ins.raw[:il.value], (upd.raw[:ul.value] if e.uptbz == 1 else None)


def bind_plan(e: CXmlToTable, cols: Sequence[Col]):
    cc = c_cols(cols)
    n = max(1, len(cols))
    ins, upd = (ctypes.c_int * n)(), (ctypes.c_int * n)()
    ni, nu = ctypes.c_int(0), ctypes.c_int(0)
    capi.check(_lib().mdc_bind_plan(ctypes.byref(e), cc, len(cols), ins, ctypes.byref(ni), upd, ctypes.byref(nu)))
    return list(ins[:ni.value]), list(upd[:nu.value])


def slab_rows(slab: CBindSlab, nrows: int) -> List[bytes]:
    """The texts of one bound column, row by row (each slot up to its NUL)."""
    import numpy as np
    if nrows == 0:
        return []
    buf = (ctypes.c_char * (nrows * slab.stride)).from_address(slab.base)
    a = np.frombuffer(buf, dtype=np.uint8).reshape(nrows, slab.stride)
    return [bytes(r).split(b"\0")[0] for r in a]


def ingest_rows(ins: Sequence[CBindSlab], upd: Sequence[CBindSlab], rows: int, uptbz: int, execute) -> CIngestCounts:
    """mdc_ingest_rows: `execute(statement, columns, nrows)` gets the bound texts as a list of columns (each a
    list of nrows byte strings) and returns the per-row database codes (0 = done, 1 = ORA-00001 ...), or None
    when the statement could not be executed at all."""
    L = _lib()

    def thunk(ctx, statement, slabs, nslabs, nrows, rowcode):
        cols = [slab_rows(slabs[k], nrows) for k in range(nslabs)]
        codes = execute(statement, cols, nrows)
        if codes is None:
            return 1
        for r in range(nrows):
            rowcode[r] = int(codes[r])
        return 0

    cb = ARRAY_EXEC(thunk)
    ia = (CBindSlab * max(1, len(ins)))(*ins)
    ua = (CBindSlab * max(1, len(upd)))(*upd)
    counts = CIngestCounts()
    capi.check(L.mdc_ingest_rows(ia, len(ins), ua if len(upd) else None, len(upd), rows, uptbz, cb, None, ctypes.byref(counts)))
    return counts


def batch_bind_slabs(batch, entry, cols: Sequence[Col]):
    """mdc_batch_bind_slabs: (insert slabs, update slabs) of a parsed batch."""
    L = _lib()
    n = len(cols)
    ins, upd = (CBindSlab * n)(), (CBindSlab * n)()
    ni, nu = ctypes.c_int(0), ctypes.c_int(0)
    ent = entry
    capi.check(L.mdc_batch_bind_slabs(batch.h, ctypes.byref(ent), c_cols(cols), n, ins, ctypes.byref(ni), upd, ctypes.byref(nu)))
    return list(ins[: ni.value]), list(upd[: nu.value])
