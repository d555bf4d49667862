"""ctypes front end of the two CPU checkers (TEST INFRASTRUCTURE ONLY).

``load("port")`` is oracle/libmdc_oracle.so, the plain-C restatement; ``load("reference")``
is oracle/_ref/libmdc_ref.so, the reference's own public/_public.cpp compiled in place
(present only where oracle/Makefile found the reference tree, or where the prebuilt file
travelled with the repository).  Both expose the same calls and return `HostBatch`.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

T_NONE, T_I32, T_U32, T_I64, T_U64, T_F64, T_F32, T_BOOL, T_TIME = range(9)


class _CCol(ctypes.Structure):
    _fields_ = [
        ("colname", ctypes.c_char * 32),
        ("datatype", ctypes.c_char * 32),
        ("collen", ctypes.c_int),
        ("pkseq", ctypes.c_int),
    ]


class _CXmlToTable(ctypes.Structure):
    _fields_ = [
        ("filename", ctypes.c_char * 128),
        ("tname", ctypes.c_char * 32),
        ("uptbz", ctypes.c_int),
        ("execsql", ctypes.c_char * 256),
    ]


class _CBatch(ctypes.Structure):
    _fields_ = [
        ("rows", ctypes.c_int64),
        ("ncols", ctypes.c_int),
        ("words_per_row", ctypes.c_int),
        ("rec_off", ctypes.POINTER(ctypes.c_uint64)),
        ("off", ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))),
        ("bytes", ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8))),
        ("nbytes", ctypes.POINTER(ctypes.c_uint64)),
        ("val", ctypes.POINTER(ctypes.POINTER(ctypes.c_uint64))),
        ("valid", ctypes.POINTER(ctypes.c_uint32)),
        ("found", ctypes.POINTER(ctypes.c_uint32)),
    ]


@dataclass
class HostBatch:
    """Columnar parse result on the host (same layout as the GPU batch)."""

    rows: int
    ncols: int
    rec_off: np.ndarray                      # uint64 [rows+1]
    off: List[np.ndarray]                    # uint32 [rows+1] per column
    data: List[np.ndarray]                   # uint8 arena per column
    val: List[Optional[np.ndarray]]          # uint64 [rows] per typed column
    valid: np.ndarray                        # uint32 [rows, words]
    found: np.ndarray                        # uint32 [rows, words]

    def cell(self, row: int, col: int) -> bytes:
        o = self.off[col]
        return bytes(self.data[col][int(o[row]):int(o[row + 1])])

    def is_valid(self, row: int, col: int) -> bool:
        return bool((int(self.valid[row, col >> 5]) >> (col & 31)) & 1)

    def is_found(self, row: int, col: int) -> bool:
        return bool((int(self.found[row, col >> 5]) >> (col & 31)) & 1)


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _ccols(cols):
    arr = (_CCol * max(1, len(cols)))()
    for i, c in enumerate(cols):
        nm = c.colname.encode() if isinstance(c.colname, str) else bytes(c.colname)
        arr[i].colname = nm
        arr[i].datatype = c.datatype.encode()
        arr[i].collen = c.collen
        arr[i].pkseq = c.pkseq
    return arr


class Checker:
    def __init__(self, path: str, prefix: str):
        self.path = path
        self.lib = ctypes.CDLL(path)
        self.prefix = prefix
        L, P = self.lib, prefix
        vp, sz, i64, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_int
        pb = ctypes.POINTER(_CBatch)
        f = lambda n: getattr(L, P + n)
        f("split_records").restype = i64
        f("split_records").argtypes = [vp, sz, ctypes.c_char_p, vp, vp, i64]
        f("getxmlbuffer").restype = ci
        f("getxmlbuffer").argtypes = [vp, sz, ctypes.c_char_p, ci, vp, sz, ctypes.POINTER(sz)]
        f("picknumber").restype = sz
        f("picknumber").argtypes = [vp, sz, vp, ci, ci]
        f("convert").restype = ci
        f("convert").argtypes = [vp, sz, ci, ctypes.POINTER(ctypes.c_uint64)]
        f("trim").restype = sz
        f("trim").argtypes = [vp, sz, ci, ci]
        f("parse_xml").restype = pb
        f("parse_xml").argtypes = [vp, sz, ctypes.c_char_p, vp, ci, vp, ci]
        f("parse_xml_file").restype = pb
        f("parse_xml_file").argtypes = [ctypes.c_char_p, ctypes.c_char_p, vp, ci, vp, ci]
        f("parse_csv").restype = pb
        f("parse_csv").argtypes = [vp, sz, ctypes.c_char_p, ci, ci, vp, vp]
        f("format_xml").restype = i64
        f("format_xml").argtypes = [vp, vp, ci, vp, vp, i64, i64, vp, vp, i64, ctypes.POINTER(i64)]
        f("format_xml_files").restype = i64
        f("format_xml_files").argtypes = [vp, vp, ci, vp, vp, i64, i64] + [ctypes.c_char_p] * 4 + [ci]
        f("timetostr").restype = ci
        f("timetostr").argtypes = [i64, ctypes.c_char_p, ctypes.c_char_p, sz]
        f("matchstr").restype = ci
        f("matchstr").argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        f("load_xmltotable").restype = i64
        f("load_xmltotable").argtypes = [vp, sz, vp, i64]
        f("crtsql").restype = ci
        f("crtsql").argtypes = [vp, vp, ci, ctypes.c_char_p, sz, ctypes.c_char_p, sz]
        f("bind_plan").restype = ci
        f("bind_plan").argtypes = [vp, vp, ci, vp, vp, ctypes.POINTER(ci)]
        f("batch_free").restype = None
        f("batch_free").argtypes = [pb]
        f("kind").restype = ctypes.c_char_p
        self._f = f

    @property
    def kind(self) -> str:
        return self._f("kind")().decode()

    # -- scalar entry points ------------------------------------------------------------
    def split_records(self, data: bytes, endbz: str = "<endl/>"):
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        ptr = buf.ctypes.data if buf.size else None
        n = self._f("split_records")(ptr, buf.size, endbz.encode(), None, None, 0)
        st = np.zeros(max(n, 1), dtype=np.uint64)
        ln = np.zeros(max(n, 1), dtype=np.uint64)
        self._f("split_records")(ptr, buf.size, endbz.encode(), st.ctypes.data, ln.ctypes.data, n)
        return st[:n], ln[:n]

    def getxmlbuffer(self, rec: bytes, name: str, ilen: int = 0):
        out = ctypes.create_string_buffer(len(rec) + 1)
        ol = ctypes.c_size_t(0)
        nm = name.encode() if isinstance(name, str) else name
        ok = self._f("getxmlbuffer")(rec, len(rec), nm, ilen, out, len(rec), ctypes.byref(ol))
        return (True, out.raw[: ol.value]) if ok else (False, None)

    def picknumber(self, s: bytes, bsigned=False, bdot=False) -> bytes:
        out = ctypes.create_string_buffer(len(s) + 1)
        n = self._f("picknumber")(s, len(s), out, int(bsigned), int(bdot))
        return out.raw[:n]

    def convert(self, s: bytes, typ: int):
        v = ctypes.c_uint64(0)
        ok = self._f("convert")(s, len(s), typ, ctypes.byref(v))
        return (True, v.value) if ok else (False, None)

    def trim(self, s: bytes, which: int, cc: str = " ") -> bytes:
        buf = ctypes.create_string_buffer(s, len(s) + 1)
        n = self._f("trim")(buf, len(s), ord(cc), which)
        return buf.raw[:n]

    # -- bulk entry points --------------------------------------------------------------
    def _take(self, pb, typed: Sequence[int]) -> HostBatch:
        if not pb:
            raise RuntimeError("checker returned NULL")
        b = pb.contents
        rows, ncols, w = int(b.rows), int(b.ncols), int(b.words_per_row)
        hb = HostBatch(
            rows=rows,
            ncols=ncols,
            rec_off=_arr(b.rec_off, rows + 1, np.uint64),
            off=[_arr(b.off[c], rows + 1, np.uint32) for c in range(ncols)],
            data=[_arr(b.bytes[c], int(b.nbytes[c]), np.uint8) for c in range(ncols)],
            val=[_arr(b.val[c], rows, np.uint64) if b.val[c] else None for c in range(ncols)],
            valid=_arr(b.valid, rows * w, np.uint32).reshape(rows, w),
            found=_arr(b.found, rows * w, np.uint32).reshape(rows, w),
        )
        self._f("batch_free")(pb)
        return hb

    def parse_xml(self, data, cols, typed=None, endbz: str = "<endl/>", carry: bool = True) -> HostBatch:
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        ty = (ctypes.c_int * max(1, len(cols)))(*typed) if typed is not None else None
        cc = _ccols(cols)
        pb = self._f("parse_xml")(buf.ctypes.data if buf.size else None, buf.size, endbz.encode(), cc, len(cols), ty, int(carry))
        return self._take(pb, typed)

    def parse_xml_file(self, path: str, cols, typed=None, endbz: str = "<endl/>", carry: bool = True) -> HostBatch:
        ty = (ctypes.c_int * max(1, len(cols)))(*typed) if typed is not None else None
        cc = _ccols(cols)
        pb = self._f("parse_xml_file")(path.encode(), endbz.encode(), cc, len(cols), ty, int(carry))
        return self._take(pb, typed)

    def parse_csv(self, data, sep: str, nfields: int, ilen=None, typed=None, bdelspace: bool = False) -> HostBatch:
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        il = (ctypes.c_int * max(1, nfields))(*ilen) if ilen is not None else None
        ty = (ctypes.c_int * max(1, nfields))(*typed) if typed is not None else None
        pb = self._f("parse_csv")(buf.ctypes.data if buf.size else None, buf.size, sep.encode(), int(bdelspace), nfields, il, ty)
        return self._take(pb, typed)


    # -- writer side --------------------------------------------------------------------
    @staticmethod
    def _columns(names, fieldlen, off, data):
        n = len(names)
        enc = [nm.encode() if isinstance(nm, str) else bytes(nm) for nm in names]
        c_names = (ctypes.c_char_p * max(1, n))(*enc)
        c_len = (ctypes.c_int * max(1, n))(*fieldlen) if fieldlen is not None else None
        offs = [np.ascontiguousarray(o, dtype=np.uint32) for o in off]
        dats = [np.ascontiguousarray(d, dtype=np.uint8) if len(d) else np.zeros(1, np.uint8) for d in data]
        c_off = (ctypes.c_void_p * max(1, n))(*[o.ctypes.data for o in offs])
        c_dat = (ctypes.c_void_p * max(1, n))(*[d.ctypes.data for d in dats])
        return c_names, c_len, c_off, c_dat, (enc, offs, dats)

    def format_xml(self, names, off, data, rows: int, maxcount: int = 0, fieldlen=None):
        """All files back to back and their offsets, or None where the reference fails."""
        c_names, c_len, c_off, c_dat, keep = self._columns(names, fieldlen, off, data)
        nf = ctypes.c_int64(0)
        fn = self._f("format_xml")
        total = fn(c_names, c_len, len(names), c_off, c_dat, rows, maxcount, None, None, 0, ctypes.byref(nf))
        if total < 0:
            return None
        out = np.zeros(max(total, 1), dtype=np.uint8)
        fo = np.zeros(nf.value + 1, dtype=np.uint64)
        fn(c_names, c_len, len(names), c_off, c_dat, rows, maxcount, out.ctypes.data, fo.ctypes.data,
           nf.value + 1, ctypes.byref(nf))
        return out[:total], fo

    def format_xml_files(self, names, off, data, rows: int, maxcount: int, outpath: str, bfilename: str,
                         timestamp: str, efilename: str, first_seq: int = 1, fieldlen=None) -> int:
        c_names, c_len, c_off, c_dat, keep = self._columns(names, fieldlen, off, data)
        return self._f("format_xml_files")(c_names, c_len, len(names), c_off, c_dat, rows, maxcount,
                                           outpath.encode(), bfilename.encode(), timestamp.encode(),
                                           efilename.encode(), first_seq)


    # -- times as text -------------------------------------------------------------------
    def timetostr(self, t: int, fmt: str = ""):
        """timetostr(t, fmt) in the process's time zone: the text as bytes, None for a text that is
        none of the formats."""
        buf = ctypes.create_string_buffer(64)
        n = self._f("timetostr")(int(t), fmt.encode(), buf, 64)
        return None if n < 0 else buf.raw[:n]

    # -- routing and SQL text -----------------------------------------------------------
    def matchstr(self, name, rules) -> bool:
        enc = lambda x: x.encode() if isinstance(x, str) else bytes(x)
        return bool(self._f("matchstr")(enc(name), enc(rules)))

    def load_xmltotable(self, data: bytes):
        """[(filename, tname, uptbz, execsql)] of a parameter file, fields as bytes."""
        data = bytes(data)
        n = self._f("load_xmltotable")(data, len(data), None, 0)
        if n < 0:
            raise RuntimeError("checker could not read the parameter file")
        arr = (_CXmlToTable * max(1, n))()
        self._f("load_xmltotable")(data, len(data), arr, n)
        return [(e.filename, e.tname, e.uptbz, e.execsql) for e in arr[:n]]

    @staticmethod
    def _entry(entry):
        e = _CXmlToTable()
        enc = lambda x: x.encode() if isinstance(x, str) else bytes(x)
        e.filename, e.tname, e.uptbz, e.execsql = enc(entry[0]), enc(entry[1]), int(entry[2]), enc(entry[3])
        return e

    def crtsql(self, entry, cols):
        """(insertsql, updatesql or None) as bytes."""
        e, cc = self._entry(entry), _ccols(cols)
        ins, upd = ctypes.create_string_buffer(1 << 16), ctypes.create_string_buffer(1 << 16)
        has = self._f("crtsql")(ctypes.byref(e), cc, len(cols), ins, len(ins), upd, len(upd))
        return ins.value, (upd.value if has else None)

    def bind_plan(self, entry, cols):
        e, cc = self._entry(entry), _ccols(cols)
        n = max(1, len(cols))
        ins, upd, nu = (ctypes.c_int * n)(), (ctypes.c_int * n)(), ctypes.c_int(0)
        ni = self._f("bind_plan")(ctypes.byref(e), cc, len(cols), ins, upd, ctypes.byref(nu))
        return list(ins[:ni]), list(upd[:nu.value])


_cache = {}


def path_of(kind: str) -> str:
    if kind == "port":
        return os.path.join(_HERE, "libmdc_oracle.so")
    if kind == "reference":
        return os.path.join(_HERE, "_ref", "libmdc_ref.so")
    raise ValueError(kind)


def available(kind: str) -> bool:
    return os.path.exists(path_of(kind))


def load(kind: str = "port") -> Checker:
    if kind not in _cache:
        p = path_of(kind)
        if not os.path.exists(p):
            raise RuntimeError(f"{p} missing: run `make -C oracle`")
        _cache[kind] = Checker(p, "orc_" if kind == "port" else "ref_")
    return _cache[kind]
