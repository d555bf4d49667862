"""Typed time columns (MDC_T_TIME): strtotime of the reference (public/_public.cpp:1009-1044).
The CPU checker calls libc's mktime in the process's time zone, the reference build calls the
reference's strtotime; the GPU computes the calendar arithmetic itself for a zone given as an offset.
strtotime sets tm_isdst to 0 (:1039), and mktime then reads the text as the zone's STANDARD time in
summer as in winter, so a zone with daylight saving is its standard offset too (STRTOTIME_ZONES).
Zones are set with POSIX TZ strings (no zone files needed)."""
import os
import random
import time

import numpy as np
import pytest

from dsp_amd import capi, schema
from dsp_amd.schema import Col
from oracle import oracle_lib as ol

ZONES = [("UTC0", 0), ("CST-8", 28800), ("EST5", -18000), ("IST-5:30", 19800)]
# zones with daylight saving: for strtotime only (timetostr's localtime does move the clock in summer)
# (a southern rule such as AEST-10AEDT,M10.1.0,M4.1.0/3 is its standard offset as well from 1962 on; for earlier
# years glibc's mktime gives up there (-1) when tm_isdst is 0, which an offset does not reproduce)
DST_ZONES = [("EST5EDT,M3.2.0,M11.1.0", -18000), ("CET-1CEST,M3.5.0,M10.5.0/3", 3600)]
STRTOTIME_ZONES = ZONES + DST_ZONES

PROBES = [b"20240519162800", b"20240314023000", b"20240310023000", b"20241103013000", b"20240331023000", b"20241027023000",
          b"20240701120000", b"20240101120000", b"20241006023000", b"20240407023000", b"2024-05-19 16:28:00", b"2024/05/19 16:28:00.000"[:19], b"19700101000000", b"19691231235959",
          b"00000000000000", b"00010101000000", b"20240229000000", b"20230229000000", b"21000229000000", b"20000229235959",
          b"20240019000000", b"20241319000000", b"20249919000000", b"20240500000000", b"20240532000000", b"20240519250000",
          b"20240519996199", b"99991231235959", b"2024051916280", b"202405191628000", b"", b"x", b"1e14",
          b"2024\x0005\x0019 16:28:00"]


@pytest.fixture
def zone(request):
    old = os.environ.get("TZ")
    os.environ["TZ"] = request.param[0]
    time.tzset()
    yield request.param
    if old is None:
        os.environ.pop("TZ", None)
    else:
        os.environ["TZ"] = old
    time.tzset()


def random_times(seed, n):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        kind = rng.random()
        if kind < 0.6:  # well-formed
            t = "%04d%02d%02d%02d%02d%02d" % (rng.randint(1, 9999), rng.randint(1, 12), rng.randint(1, 28), rng.randint(0, 23),
                                              rng.randint(0, 59), rng.randint(0, 59))
        elif kind < 0.9:  # any fourteen digits
            t = "".join(rng.choice("0123456789") for _ in range(14))
        else:  # another count
            t = "".join(rng.choice("0123456789") for _ in range(rng.choice([0, 1, 8, 13, 15, 20])))
        if rng.random() < 0.3 and len(t) >= 8:
            t = t[:4] + "-" + t[4:6] + "-" + t[6:8] + " " + t[8:]
        out.append(t.encode())
    return out


@pytest.mark.parametrize("zone", STRTOTIME_ZONES, indirect=True, ids=[z[0] for z in STRTOTIME_ZONES])
def test_port_equals_reference(zone, port, reference):
    for t in PROBES + random_times(3, 3000):
        assert port.convert(t, ol.T_TIME) == reference.convert(t, ol.T_TIME), t


def test_golden(port):
    """The values the reference's strtotime returned (tests/golden/time_cases.json): the checker in
    the same zone, and the arithmetic of the GPU code with the zone's offset."""
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "time_cases.json")))
    old = os.environ.get("TZ")
    try:
        for z in gold:
            os.environ["TZ"] = z["tz"]
            time.tzset()
            for text, want in z["cases"]:
                t = bytes.fromhex(text)
                ok, v = port.convert(t, ol.T_TIME)
                assert (("%016x" % v) if ok else None) == want, (z["tz"], t)
                m = model(t, z["tz_offset"])
                assert (None if m is None else "%016x" % m) == want, (z["tz"], t)
    finally:
        if old is None:
            os.environ.pop("TZ", None)
        else:
            os.environ["TZ"] = old
        time.tzset()


def model(text: bytes, tz_offset: int):
    """The arithmetic of the GPU code (dsp_amd/csrc/mdc_numparse.h: time_convert), in Python."""
    d = [c - 48 for c in text if 48 <= c <= 57]
    if len(d) != 14:
        return None
    year = d[0] * 1000 + d[1] * 100 + d[2] * 10 + d[3]
    mon = d[4] * 10 + d[5] - 1
    mday, hour, mi, sec = (d[6] * 10 + d[7], d[8] * 10 + d[9], d[10] * 10 + d[11], d[12] * 10 + d[13])
    y = year + mon // 12
    mon %= 12
    leap = y % 4 == 0 and (y % 100 != 0 or y % 400 == 0)
    before = [0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334][mon] + (1 if leap and mon > 1 else 0)
    L = lambda n: n // 4 - n // 100 + n // 400
    days = 365 * (y - 1970) + L(y - 1) - L(1969) + before + mday - 1
    return (((days * 24 + hour) * 60 + mi) * 60 + sec - tz_offset) & 0xFFFFFFFFFFFFFFFF


@pytest.mark.parametrize("zone", STRTOTIME_ZONES, indirect=True, ids=[z[0] for z in STRTOTIME_ZONES])
def test_calendar_arithmetic_equals_mktime(zone, port):
    for t in PROBES + random_times(5, 20000):
        ok, v = port.convert(t, ol.T_TIME)
        assert (v if ok else None) == model(t, zone[1]), t


COLS = [Col("obtid", "char", 10), Col("ddatetime", "date", 14), Col("upttime", "date", 19), Col("note", "char", 30)]
TYPED = [schema.T_NONE, schema.T_TIME, schema.T_TIME, schema.T_TIME]


def time_file(seed, n):
    times = PROBES + random_times(seed, n)
    rng = random.Random(seed)
    recs = []
    for i, t in enumerate(times):
        t = t.replace(b"<", b"").replace(b"\n", b" ")
        r = b"<obtid>%d</obtid><ddatetime>%s</ddatetime>" % (50000 + i, t)
        if rng.random() < 0.7:
            r += b"<upttime>%s</upttime>" % times[rng.randrange(len(times))].replace(b"\n", b" ")
        r += b"<note>%s</note><endl/>\n" % t
        recs.append(r)
    return np.frombuffer(b"<data>\n" + b"".join(recs) + b"</data>\n", dtype=np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("zone", STRTOTIME_ZONES, indirect=True, ids=[z[0] for z in STRTOTIME_ZONES])
def test_gpu_time_columns(zone, port):
    from util import assert_batches_equal
    data = time_file(9, 6000)
    exp = port.parse_xml(data, COLS, typed=TYPED)
    p = capi.Parser.xml(COLS, typed=TYPED, tz_offset=zone[1])
    got = p.parse_host(data).to_host()
    p.close()
    assert_batches_equal(got, exp, zone[0])
    assert int(np.count_nonzero(exp.valid)) > 3000  # (most of the texts are times)


def test_typed_code_is_accepted_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.MdcError) as e:
        capi.Parser.xml(COLS, typed=TYPED, tz_offset=28800)
    assert e.value.code == capi.MDC_ERR_NO_DEVICE  # (not MDC_ERR_ARG: the schema itself passed)


def compact_file(seed, n):
    """Records whose time cells are all short and plain (fourteen digits, some with a sign in front
    or a point inside, some with fewer digits), so that whole waves of them take the write pass's
    straight-line conversion; every field value from 00 to 99 occurs."""
    rng = random.Random(seed)
    recs = []
    for i in range(n):
        k = rng.random()
        if k < 0.5:
            t = "%04d%02d%02d%02d%02d%02d" % (rng.randint(0, 9999), rng.randint(0, 99), rng.randint(0, 99), rng.randint(0, 99),
                                              rng.randint(0, 99), rng.randint(0, 99))
        elif k < 0.9:
            t = "%04d%02d%02d%02d%02d%02d" % (rng.randint(1900, 2100), rng.randint(1, 12), rng.randint(1, 31), rng.randint(0, 23),
                                              rng.randint(0, 59), rng.randint(0, 59))
        else:
            t = "".join(rng.choice("0123456789") for _ in range(rng.choice([0, 1, 7, 8, 13, 15, 16])))
        u = t
        if rng.random() < 0.3 and 0 < len(t) <= 14:
            p = rng.randrange(len(t) + 1)
            u = rng.choice(["", "-", "+"]) + t[:p] + "." + t[p:]
        recs.append(b"<obtid>%d</obtid><ddatetime>%s</ddatetime><upttime>%s</upttime><note>%s</note><endl/>\n"
                    % (50000 + i, t.encode(), u.encode()[:16], u.encode()[:16]))
    return np.frombuffer(b"<data>\n" + b"".join(recs) + b"</data>\n", dtype=np.uint8)


# upttime as a number column (sign and point stay in the text), note as text
COLS_PLAIN = [Col("obtid", "char", 10), Col("ddatetime", "date", 14), Col("upttime", "number", 22), Col("note", "char", 30)]


@pytest.mark.gpu
@pytest.mark.parametrize("zone", STRTOTIME_ZONES, indirect=True, ids=[z[0] for z in STRTOTIME_ZONES])
def test_gpu_plain_time_cells(zone, port):
    from util import assert_batches_equal
    data = compact_file(31, 20000)
    exp = port.parse_xml(data, COLS_PLAIN, typed=TYPED)
    p = capi.Parser.xml(COLS_PLAIN, typed=TYPED, tz_offset=zone[1])
    got = p.parse_host(data).to_host()
    p.close()
    assert_batches_equal(got, exp, zone[0])
    assert int(np.count_nonzero((exp.valid[:, 0] >> 1) & 1)) > 17000
