"""Times as text (mdc_timetostr_device): timetostr of the reference (public/_public.cpp:631-722).
The CPU checker calls libc's localtime_r in the process's time zone, the reference build calls
the reference's timetostr; the GPU does the calendar arithmetic itself for a zone without daylight
saving given as an offset.  Zones are set with POSIX TZ strings (no zone files needed)."""
import json
import os
import random
import re
import time

import numpy as np
import pytest

from dsp_amd import capi, schema
from oracle import oracle_lib as ol
from test_time_columns import ZONES, zone  # noqa: F401  (the fixture that sets TZ)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMATS = capi.TIME_FORMATS
NOT_FORMATS = ["yyyy", "YYYY-MM-DD", "yyyy-mm-dd hh24:mi:ss ", "hh24:mi", " "]
T_MIN, T_MAX = -62167219200, 253402300799  # 0000-01-01 00:00:00 .. 9999-12-31 23:59:59 (local)

PROBES = [0, -1, 1, 59, 60, 3599, 3600, 86399, 86400, -86400, -86401, 1716107280, 951782400, 951868799, 4107542400,
          4107542399, 68169600, 1709164800, 1709251199, 2147483647, 2147483648, -2147483649, 253402214399,
          -62135596800, -62135596801, T_MIN + 86400, T_MAX - 86400, 32503680000, -11644473600]


def random_times(seed, n):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        k = rng.random()
        if k < 0.5:
            out.append(rng.randint(0, 1 << 32))
        elif k < 0.8:
            out.append(rng.randint(T_MIN + 86400, T_MAX - 86400))
        else:  # around the ends of years and of Februaries
            y = rng.randint(1, 9998)
            days = (y - 1970) * 365 + (y - 1969) // 4 - (y - 1901) // 100 + (y - 1601) // 400
            out.append(days * 86400 + rng.choice([-1, 0, 1, 59 * 86400 - 1, 59 * 86400, 60 * 86400]) + rng.randint(-3, 3) * 3600)
    return out


def model(t: int, tz_offset: int, fmt: str) -> bytes:
    """The arithmetic of the GPU code (dsp_amd/csrc/mdc_format.hip: timetostr_kernel), in Python."""
    local = t + tz_offset
    days, sod = divmod(local, 86400)
    z = days + 719468
    era, doe = divmod(z, 146097)
    yoe = (doe - doe // 1460 + doe // 36524 - doe // 146096) // 365
    doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
    mp = (5 * doy + 2) // 153
    mday = doy - (153 * mp + 2) // 5 + 1
    mon = mp + 3 if mp < 10 else mp - 9
    year = yoe + era * 400 + (1 if mon <= 2 else 0)
    full = "%04d-%02d-%02d %02d:%02d:%02d" % (year, mon, mday, sod // 3600, sod % 3600 // 60, sod % 60)
    packed = full[0:4] + full[5:7] + full[8:10] + full[11:13] + full[14:16] + full[17:19]
    text = {"": full, "yyyy-mm-dd hh24:mi:ss": full, "yyyy-mm-dd hh24:mi": full[:16], "yyyy-mm-dd hh24": full[:13],
            "yyyy-mm-dd": full[:10], "yyyy-mm": full[:7], "yyyymmddhh24miss": packed, "yyyymmddhh24mi": packed[:12],
            "yyyymmddhh24": packed[:10], "yyyymmdd": packed[:8], "hh24miss": packed[8:], "hh24mi": packed[8:12],
            "hh24": packed[8:10], "mi": packed[10:12]}[fmt]
    return text.encode()


def in_range(t, tz_offset):
    return T_MIN <= t + tz_offset <= T_MAX


def test_header_and_widths():
    hdr = open(os.path.join(ROOT, "include", "mdc_format.h")).read()
    assert "mdc_timetostr_device" in hdr and "mdc_timetostr_width" in hdr
    for f in FORMATS:
        assert '"%s"' % f in hdr, f  # (the header lists every format)
        assert capi.timetostr_width(f) == len(model(0, 0, f))
    for f in NOT_FORMATS:
        with pytest.raises(capi.MdcError) as e:
            capi.timetostr_width(f)
        assert e.value.code == capi.MDC_ERR_ARG


@pytest.mark.parametrize("zone", ZONES, indirect=True, ids=[z[0] for z in ZONES])
def test_port_equals_reference(zone, port, reference):
    for t in PROBES + random_times(3, 1500):
        for f in FORMATS + NOT_FORMATS:
            assert port.timetostr(t, f) == reference.timetostr(t, f), (t, f)


def test_golden(port):
    """The texts the reference's timetostr returned (tests/golden/timetostr_cases.json): the checker
    in the same zone, and the arithmetic of the GPU code with the zone's offset."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "timetostr_cases.json")))
    old = os.environ.get("TZ")
    try:
        for z in gold:
            os.environ["TZ"] = z["tz"]
            time.tzset()
            for t, texts in z["cases"]:
                for f, want in zip(z["formats"], texts):
                    got = port.timetostr(t, f)
                    assert (None if got is None else got.decode()) == want, (z["tz"], t, f)
                    if want is not None and in_range(t, z["tz_offset"]):
                        assert model(t, z["tz_offset"], f).decode() == want, (z["tz"], t, f)
    finally:
        if old is None:
            os.environ.pop("TZ", None)
        else:
            os.environ["TZ"] = old
        time.tzset()


@pytest.mark.parametrize("zone", ZONES, indirect=True, ids=[z[0] for z in ZONES])
def test_calendar_arithmetic_equals_localtime(zone, port):
    for t in PROBES + random_times(5, 20000):
        if in_range(t, zone[1]):
            assert model(t, zone[1], "") == port.timetostr(t, ""), t
    for t in PROBES + random_times(6, 300):
        if in_range(t, zone[1]):
            for f in FORMATS:
                assert model(t, zone[1], f) == port.timetostr(t, f), (t, f)


def test_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.MdcError) as e:
        capi.timetostr_device(8, 1, "", 0, 8, 8)
    assert e.value.code == capi.MDC_ERR_NO_DEVICE
    with pytest.raises(capi.MdcError) as e:
        capi.timetostr_device(8, 1, "yyyy", 0, 8, 8)
    assert e.value.code == capi.MDC_ERR_ARG  # (the format is looked at first)


# ---- GPU ---------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("zone", ZONES, indirect=True, ids=[z[0] for z in ZONES])
def test_gpu_equals_checker(zone, port):
    times = [t for t in PROBES + random_times(9, 5000) if in_range(t, zone[1])]
    arr = np.array(times, dtype=np.int64)
    for k, f in enumerate(FORMATS):
        w = capi.timetostr_width(f)
        off, data = capi.timetostr(arr, f, zone[1], misalign=k % 4)
        assert np.array_equal(off, np.arange(len(times) + 1, dtype=np.uint64) * w)
        want = b"".join(port.timetostr(t, f) for t in times)
        assert data.tobytes() == want, f


@pytest.mark.gpu
def test_gpu_row_counts_and_errors():
    for n in [0, 1, 2, 3, 255, 256, 257, 1000, 70001]:
        arr = (np.arange(n, dtype=np.int64) * 86399 + 1700000000)
        off, data = capi.timetostr(arr, "yyyymmddhh24miss", 28800)
        assert off[-1] == 14 * n and len(data) == 14 * n
        assert data.tobytes() == b"".join(model(int(t), 28800, "yyyymmddhh24miss") for t in arr)
    for bad in [T_MAX + 1, T_MIN - 1, 1 << 62, -(1 << 62)]:
        with pytest.raises(capi.MdcError) as e:
            capi.timetostr(np.array([0, bad, 5], dtype=np.int64), "", 0)
        assert e.value.code == capi.MDC_ERR_NUMERIC
    off, data = capi.timetostr(np.array([T_MAX - 28800, T_MIN - 28800], dtype=np.int64), "", 28800)  # the range is the local time's
    assert data.tobytes() == b"9999-12-31 23:59:590000-01-01 00:00:00"
    with pytest.raises(capi.MdcError) as e:
        capi.timetostr(np.array([0], dtype=np.int64), "", 86400)
    assert e.value.code == capi.MDC_ERR_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("zone", ZONES[:2], indirect=True, ids=[z[0] for z in ZONES[:2]])
def test_gpu_round_trip_with_time_columns(zone, port):
    """A typed time column of a parsed file, turned into text and parsed again, gives the same
    values (strtotime of timetostr of t is t)."""
    import test_time_columns as tc
    data = tc.time_file(21, 3000)
    p = capi.Parser.xml(tc.COLS, typed=tc.TYPED, tz_offset=zone[1])
    b = p.parse_host(data).to_host()
    valid = ((b.valid[:, 0] >> 1) & 1).astype(bool)
    vals = b.val[1].view(np.int64)[valid].copy()
    vals = vals[(vals + zone[1] >= T_MIN) & (vals + zone[1] <= T_MAX)]
    assert len(vals) > 1500
    off, text = capi.timetostr(vals, "yyyymmddhh24miss", zone[1])
    recs = b"".join(b"<obtid>1</obtid><ddatetime>%s</ddatetime><endl/>\n" % text[14 * i:14 * i + 14].tobytes()
                    for i in range(len(vals)))
    b2 = p.parse_host(np.frombuffer(recs, dtype=np.uint8)).to_host()
    p.close()
    assert np.array_equal(b2.val[1].view(np.int64), vals)
    assert all(port.convert(text[14 * i:14 * i + 14].tobytes(), ol.T_TIME) == (True, int(v) & 0xFFFFFFFFFFFFFFFF)
               for i, v in enumerate(vals[:500]))
