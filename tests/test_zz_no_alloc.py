"""mdc_parse_device_noalloc (the entry point that allocates nothing: caller-made output arrays,
workspace in a static device array) against mdc_parse_device, bit for bit, through the C ABI: two
calls in a row on one parser, the second with a smaller input, then a delimited input, the limits
and the refusals."""
import ctypes

import numpy as np
import pytest

from dsp_amd import capi, gen, schema
from util import assert_batches_equal

pytestmark = pytest.mark.gpu


class COutCol(ctypes.Structure):
    _fields_ = [("d_off", ctypes.c_void_p), ("d_bytes", ctypes.c_void_p), ("bytes_cap", ctypes.c_uint64),
                ("d_val", ctypes.c_void_p)]


class COutputs(ctypes.Structure):
    _fields_ = [("rows_cap", ctypes.c_int64), ("cols", ctypes.POINTER(COutCol)), ("d_valid", ctypes.c_void_p),
                ("d_found", ctypes.c_void_p), ("d_rec_off", ctypes.c_void_p)]


class Outputs:
    """Output arrays in device memory, made once (mdc_device_alloc) and reused by every call."""

    def __init__(self, ncols, typed, rows_cap, bytes_cap, words=1, device=0):
        self.ncols, self.typed, self.rows_cap, self.bytes_cap, self.words, self.device = ncols, typed, rows_cap, bytes_cap, words, device
        self.bufs = []
        self.cols = (COutCol * ncols)()
        for c in range(ncols):
            self.cols[c].d_off = self._alloc((rows_cap + 1) * 4)
            self.cols[c].d_bytes = self._alloc(bytes_cap)
            self.cols[c].bytes_cap = bytes_cap
            self.cols[c].d_val = self._alloc(rows_cap * 8) if typed[c] != schema.T_NONE else None
        self.o = COutputs(rows_cap, self.cols, self._alloc(rows_cap * words * 4), self._alloc(rows_cap * words * 4),
                          self._alloc((rows_cap + 1) * 8))

    def _alloc(self, n):
        p = ctypes.c_void_p()
        capi.check(capi.lib().mdc_device_alloc(self.device, max(n, 16), ctypes.byref(p)))
        self.bufs.append(p)
        return p.value

    def _down(self, ptr, count, dtype):
        a = np.zeros(count, dtype)
        if count:
            capi.check(capi.lib().mdc_device_download(self.device, a.ctypes.data, ptr, a.nbytes))
        return a

    def host(self, rows, col_bytes):
        off = [self._down(self.cols[c].d_off, rows + 1, np.uint32) for c in range(self.ncols)]
        data = [self._down(self.cols[c].d_bytes, int(col_bytes[c]), np.uint8) for c in range(self.ncols)]
        val = [self._down(self.cols[c].d_val, rows, np.uint64) if self.typed[c] != schema.T_NONE else None
               for c in range(self.ncols)]
        valid = self._down(self.o.d_valid, rows * self.words, np.uint32).reshape(rows, self.words)
        found = self._down(self.o.d_found, rows * self.words, np.uint32).reshape(rows, self.words)
        return capi.HostColumns(rows, self.ncols, self._down(self.o.d_rec_off, rows + 1, np.uint64), off, data, val, valid, found)

    def free(self):
        for p in self.bufs:
            capi.lib().mdc_device_free(self.device, p)
        self.bufs = []


def _noalloc(p, dbuf, out):
    L = capi.lib()
    L.mdc_parse_device_noalloc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(COutputs),
                                           ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_uint64)]
    rows = ctypes.c_int64(-1)
    cb = (ctypes.c_uint64 * out.ncols)()
    capi.check(L.mdc_parse_device_noalloc(p.h, dbuf.ptr, dbuf.n, ctypes.byref(out.o), ctypes.byref(rows), cb))
    return out.host(rows.value, list(cb))


def test_two_calls_in_a_row_equal_the_allocating_entry_point():
    typed = schema.default_typed(schema.ZHOBTMIND)
    p = capi.Parser.xml(schema.ZHOBTMIND)
    big, nbig = gen.mind_xml(rows=40000)      # ~6.9 MB, 420 tiles
    small, nsmall = gen.mind_xml(rows=1700, first_row=77)
    # missing tags and a long cell in the second input: other code paths than the first
    # ... and a record longer than any look-behind (its list entry and cell lengths lie in the scratch array too)
    small = np.concatenate([small, np.frombuffer(b"<obtid>59999</obtid><t>-1.5e3</t><vis>" + b"7" * 40 + b"</vis><endl/>\n" +
                                                 b"<obtid>60000</obtid><junk>" + b"j" * 50000 + b"</junk><t>2.5</t><endl/>\n", np.uint8)])
    out = Outputs(11, typed, rows_cap=nbig + 10, bytes_cap=1 << 20)
    for data in (big, small):
        d = capi.DeviceBuffer(data)
        exp = p.parse_device(d.ptr, d.n)
        got = _noalloc(p, d, out)
        assert_batches_equal(got, exp.to_host(), "no-alloc entry point against mdc_parse_device")
        exp.free()
        d.free()
    # ... and the allocating entry point still works on the same parser afterwards
    d = capi.DeviceBuffer(big)
    b = p.parse_device(d.ptr, d.n)
    assert b.rows == nbig
    b.free()
    d.free()
    out.free()
    p.close()


def test_delimited_input_and_empty_input():
    typed = schema.MIND_CSV_TYPED
    p = capi.Parser.csv(",", 10, typed=typed)
    data, n = gen.mind_csv(rows=9000)
    out = Outputs(10, typed, rows_cap=n, bytes_cap=1 << 18)
    for buf in (data, np.zeros(0, np.uint8), data[: len(data) // 3]):
        d = capi.DeviceBuffer(buf)
        exp = p.parse_device(d.ptr, d.n)
        got = _noalloc(p, d, out)
        assert_batches_equal(got, exp.to_host(), "delimited")
        exp.free()
        d.free()
    out.free()
    p.close()


def test_refusals_leave_the_parser_usable():
    typed = schema.default_typed(schema.ZHOBTMIND)
    p = capi.Parser.xml(schema.ZHOBTMIND)
    data, n = gen.mind_xml(rows=3000)
    d = capi.DeviceBuffer(data)
    few_rows = Outputs(11, typed, rows_cap=n - 1, bytes_cap=1 << 18)
    with pytest.raises(capi.MdcError) as e:
        _noalloc(p, d, few_rows)
    assert e.value.code == capi.MDC_ERR_TOO_LARGE
    few_bytes = Outputs(11, typed, rows_cap=n, bytes_cap=3000)
    with pytest.raises(capi.MdcError) as e:
        _noalloc(p, d, few_bytes)
    assert e.value.code == capi.MDC_ERR_TOO_LARGE
    # an input whose counters do not fit the static array: refused before any launch (the size alone
    # decides, so a short buffer stands in for the long one: nothing is read before the refusal)
    with pytest.raises(capi.MdcError) as e:
        L = capi.lib()
        rows = ctypes.c_int64()
        capi.check(L.mdc_parse_device_noalloc(p.h, d.ptr, 400 << 20, ctypes.byref(few_rows.o), ctypes.byref(rows), None))
    assert e.value.code == capi.MDC_ERR_TOO_LARGE and "scratch" in str(e.value)
    ok = Outputs(11, typed, rows_cap=n, bytes_cap=1 << 18)
    exp = p.parse_device(d.ptr, d.n)
    assert_batches_equal(_noalloc(p, d, ok), exp.to_host(), "after the refusals")
    exp.free()
    for o in (few_rows, few_bytes, ok):
        o.free()
    d.free()
    p.close()
