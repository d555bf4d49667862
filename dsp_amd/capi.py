"""ctypes binding of libmdc_parse.so (include/mdc_parse.h).

Used by the tests, bench.py and the examples; everything goes through the C ABI, exactly as a
C/C++ caller would.  There is no Python or CPU implementation of the parsing path here: if the
shared library is missing, or no GPU is usable, calls fail loudly.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from .schema import CCol, Col, c_cols

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmdc_parse.so")

MDC_OK = 0
MDC_STREAM_END = 1
MDC_FILE_OPEN_FAILED = 5
MDC_ERR_ARG = -1
MDC_ERR_NO_DEVICE = -2
MDC_ERR_HIP = -3
MDC_ERR_NOMEM = -4
MDC_ERR_RECORD_TOO_LONG = -5
MDC_ERR_TOO_LARGE = -6
MDC_ERR_NUMERIC = -7
MDC_ERR_IO = -8
MDC_ERR_INTERNAL = -9
MDC_ERR_BUSY = -10
MDC_F_NO_RECOFF = 1
MDC_F_ONE_PASS = 2
MDC_F_RECORDS_ONLY = 4

# every symbol include/mdc_parse.h declares
EXPORTS = [
    "mdc_version", "mdc_device_count", "mdc_strerror", "mdc_last_error",
    "mdc_parser_create_xml", "mdc_parser_create_csv", "mdc_parser_destroy",
    "mdc_parse_device", "mdc_parse_host", "mdc_parse_file", "mdc_parse_device_noalloc", "mdc_parser_last_timing",
    "mdc_parse_xml_buffer", "mdc_parse_xml_file", "mdc_parse_csv_buffer",
    "mdc_stream_open", "mdc_stream_next", "mdc_stream_bytes", "mdc_stream_close", "mdc_stream_release_pinned",
    "mdc_host_buffer_get", "mdc_host_buffer_put",
    "mdc_batch_rows", "mdc_batch_ncols", "mdc_batch_words_per_row",
    "mdc_batch_device_column", "mdc_batch_device_values", "mdc_batch_device_masks", "mdc_batch_device_recoff",
    "mdc_batch_host_column", "mdc_batch_host_values", "mdc_batch_host_masks", "mdc_batch_host_recoff",
    "mdc_batch_cell", "mdc_batch_column_fixed", "mdc_batch_free",
    "mdc_device_alloc", "mdc_device_free", "mdc_device_upload", "mdc_device_download",
    "mdc_device_synchronize",
]
# every symbol include/mdc_format.h declares
FORMAT_EXPORTS = [
    "mdc_formatter_create", "mdc_formatter_destroy", "mdc_format_device", "mdc_format_host", "mdc_format_batch",
    "mdc_formatter_last_timing", "mdc_text_files", "mdc_text_bytes", "mdc_text_file_range", "mdc_text_device",
    "mdc_text_host", "mdc_text_write_files", "mdc_text_free", "mdc_timetostr_device", "mdc_timetostr_width",
]


class MdcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mdc error {code}: {msg}")
        self.code = code


class COpts(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int),
        ("endbz", ctypes.c_char_p),
        ("carry", ctypes.c_int),
        ("typed", ctypes.POINTER(ctypes.c_int)),
        ("flags", ctypes.c_uint),
        ("tz_offset", ctypes.c_int),
    ]


class CTiming(ctypes.Structure):
    _fields_ = [
        ("count_ms", ctypes.c_float),
        ("scan_ms", ctypes.c_float),
        ("write_ms", ctypes.c_float),
        ("total_ms", ctypes.c_float),
        ("bytes_in", ctypes.c_uint64),
        ("bytes_out", ctypes.c_uint64),
        ("rows", ctypes.c_int64),
        ("tiles", ctypes.c_int),
        ("full_write", ctypes.c_int),
        ("lookup_tiles", ctypes.c_int),
        ("long_records", ctypes.c_int),
        ("one_pass", ctypes.c_int),
        ("one_pass_redone", ctypes.c_int),
    ]


class CFormatTiming(ctypes.Structure):
    _fields_ = [
        ("format_ms", ctypes.c_float),
        ("total_ms", ctypes.c_float),
        ("bytes_in", ctypes.c_uint64),
        ("bytes_out", ctypes.c_uint64),
        ("exact_pass", ctypes.c_int),
    ]


_lib = None


def lib():
    """Loads libmdc_parse.so; raises if it has not been built (there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m dsp_amd.build` "
                               "(the parsing path exists only as gfx950 code)")
        L = ctypes.CDLL(LIB_PATH)
        vp, sz, ci, i64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int64
        pp = ctypes.POINTER(vp)
        L.mdc_version.restype = ctypes.c_char_p
        L.mdc_strerror.restype = ctypes.c_char_p
        L.mdc_strerror.argtypes = [ci]
        L.mdc_last_error.restype = ctypes.c_char_p
        L.mdc_device_count.restype = ci
        L.mdc_parser_create_xml.argtypes = [vp, ci, ctypes.POINTER(COpts), pp]
        L.mdc_parser_create_csv.argtypes = [ctypes.c_char_p, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(COpts), pp]
        L.mdc_parser_destroy.argtypes = [vp]
        L.mdc_parser_destroy.restype = None
        L.mdc_parse_device.argtypes = [vp, vp, sz, pp]
        L.mdc_parse_host.argtypes = [vp, vp, sz, pp]
        L.mdc_parse_file.argtypes = [vp, ctypes.c_char_p, pp]
        L.mdc_parser_last_timing.argtypes = [vp, ctypes.POINTER(CTiming)]
        L.mdc_parse_xml_buffer.argtypes = [vp, sz, vp, ci, ctypes.POINTER(COpts), pp]
        L.mdc_parse_xml_file.argtypes = [ctypes.c_char_p, vp, ci, ctypes.POINTER(COpts), pp]
        L.mdc_parse_csv_buffer.argtypes = [vp, sz, ctypes.c_char_p, ci, ci, ctypes.POINTER(ci),
                                           ctypes.POINTER(COpts), pp]
        L.mdc_stream_open.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ci, ci, pp]
        L.mdc_stream_next.argtypes = [vp, pp, ctypes.POINTER(ci)]
        L.mdc_stream_bytes.argtypes = [vp]
        L.mdc_stream_bytes.restype = ctypes.c_uint64
        L.mdc_stream_close.argtypes = [vp]
        L.mdc_stream_close.restype = None
        L.mdc_stream_release_pinned.argtypes = []
        L.mdc_stream_release_pinned.restype = None
        L.mdc_batch_rows.restype = i64
        L.mdc_batch_rows.argtypes = [vp]
        L.mdc_batch_ncols.argtypes = [vp]
        L.mdc_batch_words_per_row.argtypes = [vp]
        L.mdc_batch_device_column.argtypes = [vp, ci, pp, pp, ctypes.POINTER(ctypes.c_uint64)]
        L.mdc_batch_device_values.argtypes = [vp, ci, pp]
        L.mdc_batch_device_masks.argtypes = [vp, pp, pp]
        L.mdc_batch_device_recoff.argtypes = [vp, pp]
        L.mdc_batch_host_column.argtypes = [vp, ci, pp, pp, ctypes.POINTER(ctypes.c_uint64)]
        L.mdc_batch_host_values.argtypes = [vp, ci, pp]
        L.mdc_batch_host_masks.argtypes = [vp, pp, pp]
        L.mdc_batch_host_recoff.argtypes = [vp, pp]
        L.mdc_batch_cell.argtypes = [vp, i64, ci, pp, ctypes.POINTER(ctypes.c_uint32)]
        L.mdc_batch_column_fixed.argtypes = [vp, ci, pp, ctypes.POINTER(sz)]
        L.mdc_batch_free.argtypes = [vp]
        L.mdc_batch_free.restype = None
        L.mdc_device_alloc.argtypes = [ci, sz, pp]
        L.mdc_device_free.argtypes = [ci, vp]
        L.mdc_device_upload.argtypes = [ci, vp, vp, sz]
        L.mdc_device_download.argtypes = [ci, vp, vp, sz]
        L.mdc_device_synchronize.argtypes = [ci]
        u64 = ctypes.c_uint64
        L.mdc_formatter_create.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ci), ci, ci, pp]
        L.mdc_formatter_destroy.argtypes = [vp]
        L.mdc_formatter_destroy.restype = None
        L.mdc_format_device.argtypes = [vp, pp, pp, i64, i64, pp]
        L.mdc_format_host.argtypes = [vp, pp, pp, i64, i64, pp]
        L.mdc_format_batch.argtypes = [vp, vp, i64, pp]
        L.mdc_formatter_last_timing.argtypes = [vp, ctypes.POINTER(CFormatTiming)]
        L.mdc_text_files.argtypes = [vp]
        L.mdc_text_files.restype = i64
        L.mdc_text_bytes.argtypes = [vp]
        L.mdc_text_bytes.restype = u64
        L.mdc_text_file_range.argtypes = [vp, i64, ctypes.POINTER(u64), ctypes.POINTER(u64)]
        L.mdc_text_device.argtypes = [vp, pp, ctypes.POINTER(u64)]
        L.mdc_text_host.argtypes = [vp, pp, ctypes.POINTER(u64)]
        L.mdc_text_write_files.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                           ci, ctypes.POINTER(ci)]
        L.mdc_text_free.argtypes = [vp]
        L.mdc_text_free.restype = None
        L.mdc_timetostr_device.argtypes = [ci, vp, i64, ctypes.c_char_p, ci, vp, vp]
        L.mdc_timetostr_width.argtypes = [ctypes.c_char_p]
        _lib = L
    return _lib


def check(rc: int):
    if rc != MDC_OK:
        raise MdcError(rc, lib().mdc_last_error().decode(errors="replace"))


def _np_from(ptr, count, dtype):
    if count == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (ctypes.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


@dataclass
class HostColumns:
    """Host copy of a batch (same fields as the CPU checker's result, so tests can compare)."""

    rows: int
    ncols: int
    rec_off: Optional[np.ndarray]
    off: List[np.ndarray]
    data: List[np.ndarray]
    val: List[Optional[np.ndarray]]
    valid: np.ndarray
    found: np.ndarray

    def cell(self, row: int, col: int) -> bytes:
        o = self.off[col]
        return bytes(self.data[col][int(o[row]):int(o[row + 1])])


class Batch:
    def __init__(self, handle: int):
        self.h = ctypes.c_void_p(handle)

    def __del__(self):
        self.free()

    def free(self):
        if self.h:
            lib().mdc_batch_free(self.h)
            self.h = None

    @property
    def rows(self) -> int:
        return int(lib().mdc_batch_rows(self.h))

    @property
    def ncols(self) -> int:
        return int(lib().mdc_batch_ncols(self.h))

    @property
    def words(self) -> int:
        return int(lib().mdc_batch_words_per_row(self.h))

    def column(self, col: int):
        off, data, nb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        check(lib().mdc_batch_host_column(self.h, col, ctypes.byref(off), ctypes.byref(data), ctypes.byref(nb)))
        return _np_from(off.value, self.rows + 1, np.uint32), _np_from(data.value, nb.value, np.uint8)

    def values(self, col: int) -> Optional[np.ndarray]:
        v = ctypes.c_void_p()
        rc = lib().mdc_batch_host_values(self.h, col, ctypes.byref(v))
        if rc == MDC_ERR_ARG:
            return None
        check(rc)
        return _np_from(v.value, self.rows, np.uint64)

    def masks(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        check(lib().mdc_batch_host_masks(self.h, ctypes.byref(a), ctypes.byref(b)))
        w = self.words
        return (_np_from(a.value, self.rows * w, np.uint32).reshape(self.rows, w),
                _np_from(b.value, self.rows * w, np.uint32).reshape(self.rows, w))

    def rec_off(self) -> Optional[np.ndarray]:
        p = ctypes.c_void_p()
        rc = lib().mdc_batch_host_recoff(self.h, ctypes.byref(p))
        if rc == MDC_ERR_ARG:
            return None
        check(rc)
        return _np_from(p.value, self.rows + 1, np.uint64)

    def cell(self, row: int, col: int) -> bytes:
        p, n = ctypes.c_void_p(), ctypes.c_uint32()
        check(lib().mdc_batch_cell(self.h, row, col, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p.value, n.value) if n.value else b""

    def column_fixed(self, col: int):
        p, st = ctypes.c_void_p(), ctypes.c_size_t()
        check(lib().mdc_batch_column_fixed(self.h, col, ctypes.byref(p), ctypes.byref(st)))
        return _np_from(p.value, self.rows * st.value, np.uint8).reshape(self.rows, st.value), st.value

    def device_column(self, col: int):
        off, data, nb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        check(lib().mdc_batch_device_column(self.h, col, ctypes.byref(off), ctypes.byref(data), ctypes.byref(nb)))
        return off.value, data.value, nb.value

    def to_host(self) -> HostColumns:
        cols = [self.column(c) for c in range(self.ncols)]
        valid, found = self.masks()
        return HostColumns(self.rows, self.ncols, self.rec_off(), [c[0] for c in cols], [c[1] for c in cols],
                           [self.values(c) for c in range(self.ncols)], valid, found)


class Parser:
    """A schema bound to one GPU (mdc_parser)."""

    def __init__(self, handle, device: int, keep):
        self.h = handle
        self.device = device
        self._keep = keep

    @classmethod
    def xml(cls, cols: Sequence[Col], typed: Optional[Sequence[int]] = None, endbz: Optional[str] = "<endl/>",
            carry: bool = True, device: int = 0, flags: int = 0, tz_offset: int = 0) -> "Parser":
        cc = c_cols(cols)
        ty = (ctypes.c_int * len(cols))(*typed) if typed is not None else None
        opts = COpts(device, endbz.encode() if endbz is not None else None, int(carry),
                     ctypes.cast(ty, ctypes.POINTER(ctypes.c_int)) if ty is not None else None, flags, tz_offset)
        h = ctypes.c_void_p()
        check(lib().mdc_parser_create_xml(cc, len(cols), ctypes.byref(opts), ctypes.byref(h)))
        return cls(h, device, (cc, ty, opts))

    @classmethod
    def csv(cls, sep: str, nfields: int, ilen: Optional[Sequence[int]] = None,
            typed: Optional[Sequence[int]] = None, bdelspace: bool = False, device: int = 0,
            flags: int = 0, tz_offset: int = 0) -> "Parser":
        il = (ctypes.c_int * nfields)(*ilen) if ilen is not None else None
        ty = (ctypes.c_int * nfields)(*typed) if typed is not None else None
        opts = COpts(device, None, 0, ctypes.cast(ty, ctypes.POINTER(ctypes.c_int)) if ty is not None else None, flags,
                     tz_offset)
        h = ctypes.c_void_p()
        check(lib().mdc_parser_create_csv(sep.encode(), int(bdelspace), nfields,
                                          ctypes.cast(il, ctypes.POINTER(ctypes.c_int)) if il is not None else None,
                                          ctypes.byref(opts), ctypes.byref(h)))
        return cls(h, device, (il, ty, opts))

    def close(self):
        if self.h:
            lib().mdc_parser_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def parse_host(self, data) -> Batch:
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        h = ctypes.c_void_p()
        check(lib().mdc_parse_host(self.h, buf.ctypes.data if buf.size else None, buf.size, ctypes.byref(h)))
        return Batch(h.value)

    def parse_device(self, d_ptr: int, n: int) -> Batch:
        h = ctypes.c_void_p()
        check(lib().mdc_parse_device(self.h, d_ptr, n, ctypes.byref(h)))
        return Batch(h.value)

    def parse_file(self, path: str) -> Batch:
        h = ctypes.c_void_p()
        check(lib().mdc_parse_file(self.h, path.encode(), ctypes.byref(h)))
        return Batch(h.value)

    def timing(self) -> CTiming:
        t = CTiming()
        check(lib().mdc_parser_last_timing(self.h, ctypes.byref(t)))
        return t

    def stream(self, paths: Sequence[str], threads: int = 4):
        """Files (plain or gzip) in list order: yields (index, Batch) or (index, MdcError) per file;
        reading and inflating the next files and uploading the next one overlap the parse."""
        arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
        st = ctypes.c_void_p()
        check(lib().mdc_stream_open(self.h, arr, len(paths), threads, ctypes.byref(st)))
        try:
            while True:
                h = ctypes.c_void_p()
                idx = ctypes.c_int(-1)
                rc = lib().mdc_stream_next(st, ctypes.byref(h), ctypes.byref(idx))
                if rc == MDC_STREAM_END:
                    return
                if rc != MDC_OK:
                    yield idx.value, MdcError(rc, lib().mdc_last_error().decode(errors="replace"))
                else:
                    yield idx.value, Batch(h.value)
        finally:
            lib().mdc_stream_close(st)


class DeviceBuffer:
    """Input bytes resident in HBM (mdc_device_alloc / upload)."""

    def __init__(self, data: np.ndarray, device: int = 0):
        self.device = device
        self.n = int(data.size)
        p = ctypes.c_void_p()
        check(lib().mdc_device_alloc(device, max(self.n, 1), ctypes.byref(p)))
        self.ptr = p.value
        if self.n:
            data = np.ascontiguousarray(data)
            check(lib().mdc_device_upload(device, self.ptr, data.ctypes.data, self.n))

    def free(self):
        if self.ptr:
            lib().mdc_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()


class Text:
    """Formatted files back to back in HBM (mdc_text)."""

    def __init__(self, handle: int):
        self.h = ctypes.c_void_p(handle)

    def free(self):
        if self.h:
            lib().mdc_text_free(self.h)
            self.h = None

    def __del__(self):
        self.free()

    @property
    def files(self) -> int:
        return int(lib().mdc_text_files(self.h))

    @property
    def nbytes(self) -> int:
        return int(lib().mdc_text_bytes(self.h))

    def file_off(self) -> np.ndarray:
        out = np.zeros(self.files + 1, dtype=np.uint64)
        for k in range(self.files):
            a, b = ctypes.c_uint64(), ctypes.c_uint64()
            check(lib().mdc_text_file_range(self.h, k, ctypes.byref(a), ctypes.byref(b)))
            out[k], out[k + 1] = a.value, b.value
        return out

    def device(self):
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        check(lib().mdc_text_device(self.h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def host(self) -> np.ndarray:
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        check(lib().mdc_text_host(self.h, ctypes.byref(p), ctypes.byref(n)))
        return _np_from(p.value, n.value, np.uint8)

    def write_files(self, outpath: str, bfilename: str, efilename: str, timestamp: Optional[str] = None,
                    first_seq: int = 1) -> int:
        n = ctypes.c_int(0)
        check(lib().mdc_text_write_files(self.h, outpath.encode(), bfilename.encode(),
                                         timestamp.encode() if timestamp is not None else None,
                                         efilename.encode(), first_seq, ctypes.byref(n)))
        return n.value


class Formatter:
    """A field list bound to one GPU (mdc_formatter): columns -> "<name>value</name>...<endl/>" text."""

    def __init__(self, names: Sequence, fieldlen: Optional[Sequence[int]] = None, device: int = 0):
        self.h = None
        enc = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
        arr = (ctypes.c_char_p * max(1, len(enc)))(*enc)
        fl = (ctypes.c_int * max(1, len(enc)))(*fieldlen) if fieldlen is not None else None
        h = ctypes.c_void_p()
        check(lib().mdc_formatter_create(arr, fl, len(enc), device, ctypes.byref(h)))
        self.h = h
        self.device = device
        self.nfields = len(enc)

    def close(self):
        if self.h:
            lib().mdc_formatter_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _ptrs(self, values):
        return (ctypes.c_void_p * max(1, self.nfields))(*values)

    def format_host(self, off: Sequence[np.ndarray], data: Sequence[np.ndarray], rows: int, maxcount: int = 0) -> Text:
        offs = [np.ascontiguousarray(o, dtype=np.uint32) for o in off]
        dats = [np.ascontiguousarray(d, dtype=np.uint8) if len(d) else np.zeros(1, np.uint8) for d in data]
        h = ctypes.c_void_p()
        check(lib().mdc_format_host(self.h, self._ptrs([o.ctypes.data for o in offs]),
                                    self._ptrs([d.ctypes.data for d in dats]), rows, maxcount, ctypes.byref(h)))
        return Text(h.value)

    def format_device(self, d_off: Sequence[int], d_bytes: Sequence[int], rows: int, maxcount: int = 0) -> Text:
        h = ctypes.c_void_p()
        check(lib().mdc_format_device(self.h, self._ptrs(list(d_off)), self._ptrs(list(d_bytes)), rows, maxcount,
                                      ctypes.byref(h)))
        return Text(h.value)

    def format_batch(self, batch: Batch, maxcount: int = 0) -> Text:
        h = ctypes.c_void_p()
        check(lib().mdc_format_batch(self.h, batch.h, maxcount, ctypes.byref(h)))
        return Text(h.value)

    def timing(self) -> CFormatTiming:
        t = CFormatTiming()
        check(lib().mdc_formatter_last_timing(self.h, ctypes.byref(t)))
        return t


TIME_FORMATS = ["", "yyyy-mm-dd hh24:mi:ss", "yyyy-mm-dd hh24:mi", "yyyy-mm-dd hh24", "yyyy-mm-dd", "yyyy-mm",
                "yyyymmddhh24miss", "yyyymmddhh24mi", "yyyymmddhh24", "yyyymmdd", "hh24miss", "hh24mi", "hh24", "mi"]


def timetostr_width(fmt: str) -> int:
    """Bytes of the text timetostr gives for fmt (mdc_timetostr_width)."""
    w = lib().mdc_timetostr_width(fmt.encode())
    if w < 0:
        check(w)
    return w


def timetostr_device(d_time: int, rows: int, fmt: str, tz_offset: int, d_off: int, d_bytes: int, device: int = 0):
    """mdc_timetostr_device on device pointers."""
    check(lib().mdc_timetostr_device(device, d_time, rows, fmt.encode(), tz_offset, d_off, d_bytes))


def timetostr(times: np.ndarray, fmt: str = "", tz_offset: int = 0, device: int = 0, misalign: int = 0):
    """The texts of a host array of time_t values, through the GPU: (offsets[rows+1], bytes).
    misalign shifts the byte arena's start (the library takes any alignment)."""
    times = np.ascontiguousarray(times, dtype=np.int64)
    rows = int(times.size)
    w = timetostr_width(fmt)
    d_t = DeviceBuffer(times.view(np.uint8), device)
    d_off = DeviceBuffer(np.zeros((rows + 1) * 4, np.uint8), device)
    d_b = DeviceBuffer(np.zeros(rows * w + 8, np.uint8), device)
    try:
        timetostr_device(d_t.ptr, rows, fmt, tz_offset, d_off.ptr, d_b.ptr + misalign, device)
        off = np.zeros(rows + 1, np.uint32)
        data = np.zeros(max(rows * w, 1), np.uint8)
        check(lib().mdc_device_download(device, off.ctypes.data, d_off.ptr, off.nbytes))
        if rows:
            check(lib().mdc_device_download(device, data.ctypes.data, d_b.ptr + misalign, rows * w))
        return off, data[:rows * w]
    finally:
        d_t.free()
        d_off.free()
        d_b.free()
