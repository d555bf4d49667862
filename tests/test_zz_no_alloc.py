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


class GuardedOutputs(Outputs):
    """Output arrays sized EXACTLY (rows + 1 offsets, rows values, col_bytes[c] bytes), each inside a larger
    allocation filled with a sentinel: 256 bytes in front of it and at least 256 behind it."""
    GUARD, FILL = 256, 0xA5

    def __init__(self, ncols, typed, rows, col_bytes, words=1, device=0):
        self.ncols, self.typed, self.rows_cap, self.words, self.device = ncols, typed, rows, words, device
        self.bufs, self.spans = [], []
        self.cols = (COutCol * ncols)()
        for c in range(ncols):
            self.cols[c].d_off = self._guarded((rows + 1) * 4)
            self.cols[c].d_bytes = self._guarded(int(col_bytes[c]))
            self.cols[c].bytes_cap = int(col_bytes[c])
            self.cols[c].d_val = self._guarded(rows * 8) if typed[c] != schema.T_NONE else None
        self.o = COutputs(rows, self.cols, self._guarded(rows * words * 4), self._guarded(rows * words * 4),
                          self._guarded((rows + 1) * 8))

    def _guarded(self, n):
        inner = (n + 15) & ~15  # (the array itself ends at n; what lies between n and the next 16 is guard as well)
        total = self.GUARD + inner + self.GUARD
        p = ctypes.c_void_p()
        capi.check(capi.lib().mdc_device_alloc(self.device, total, ctypes.byref(p)))
        fill = np.full(total, self.FILL, np.uint8)
        capi.check(capi.lib().mdc_device_upload(self.device, p, fill.ctypes.data, total))
        self.bufs.append(p)
        self.spans.append((p.value, n, total))
        return p.value + self.GUARD

    def guards_untouched(self):
        for base, n, total in self.spans:
            a = np.zeros(total, np.uint8)
            capi.check(capi.lib().mdc_device_download(self.device, a.ctypes.data, base, total))
            if not (a[: self.GUARD] == self.FILL).all() or not (a[self.GUARD + n:] == self.FILL).all():
                return False
        return True


def test_arrays_sized_exactly_and_nothing_stored_beside_them(port):
    """Offsets, values, masks and cells of exactly the sizes a first call reported: the kernels store nothing
    in front of an array or behind its end (regular records, a long record, cells that go through the
    second conversion stage, delimited input)."""
    typed = schema.default_typed(schema.ZHOBTMIND)
    data, n = gen.mind_xml(rows=5000)
    extra = (b"<obtid>59999</obtid><t>-1.5e3</t><vis>" + b"7" * 40 + b"</vis><keyid>12345678901234567890123</keyid><endl/>\n" +
             b"<obtid>60000</obtid><junk>" + b"j" * 50000 + b"</junk><t>2.5</t><p>0.30000000000000004440892098500626</p><endl/>\n")
    xml = np.concatenate([data, np.frombuffer(extra, np.uint8), data[7:]])  # (data[7:]: without the opening line)
    csv, ncsv = gen.mind_csv(rows=7000)
    for kind, buf in (("xml", xml), ("csv", csv)):
        p = capi.Parser.xml(schema.ZHOBTMIND) if kind == "xml" else capi.Parser.csv(",", 10, typed=schema.MIND_CSV_TYPED)
        ty = typed if kind == "xml" else schema.MIND_CSV_TYPED
        d = capi.DeviceBuffer(buf)
        exp = p.parse_device(d.ptr, d.n)
        eh = exp.to_host()
        col_bytes = [len(eh.data[c]) for c in range(eh.ncols)]
        out = GuardedOutputs(eh.ncols, ty, eh.rows, col_bytes)
        for k in range(2):
            got = _noalloc(p, d, out)
            assert_batches_equal(got, eh, f"{kind}: exactly sized arrays, call {k}")
            assert out.guards_untouched(), f"{kind}: a store beside an output array (call {k})"
        if kind == "xml":
            assert p.timing().long_records == 1
            assert_batches_equal(eh, port.parse_xml(buf, schema.ZHOBTMIND), "against the CPU checker")
        exp.free()
        out.free()
        d.free()
        p.close()


def test_a_second_call_on_the_same_gpu_is_refused_while_one_runs():
    """Two threads, two parsers, one GPU: a call either succeeds or is refused with MDC_ERR_BUSY; every
    call that succeeds gives the right rows (no call works in a scratch array another one is using)."""
    import threading
    typed = schema.default_typed(schema.ZHOBTMIND)
    data, n = gen.mind_xml(rows=60000)
    d = capi.DeviceBuffer(data)
    L = capi.lib()
    L.mdc_parse_device_noalloc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(COutputs),
                                           ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_uint64)]
    results = [[], []]

    def worker(i):
        p = capi.Parser.xml(schema.ZHOBTMIND)
        out = Outputs(11, typed, rows_cap=n, bytes_cap=1 << 20)
        for _ in range(40):
            rows = ctypes.c_int64(-1)
            rc = L.mdc_parse_device_noalloc(p.h, d.ptr, d.n, ctypes.byref(out.o), ctypes.byref(rows), None)
            results[i].append((rc, rows.value))
        out.free()
        p.close()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    d.free()
    flat = results[0] + results[1]
    assert all(rc in (capi.MDC_OK, capi.MDC_ERR_BUSY) for rc, _ in flat), flat
    assert all(rows == n for rc, rows in flat if rc == capi.MDC_OK)
    assert any(rc == capi.MDC_OK for rc, _ in flat)
