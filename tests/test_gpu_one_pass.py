"""The one-pass parse (MDC_F_ONE_PASS: sizing and writing in one kernel, every tile learning its place
from the tiles in front of it inside the launch) against the CPU checker and the golden vectors, bit for
bit: tiles that take the straight-line write, tiles left to the full write kernel, and the inputs the
pass has to give up (output arrays guessed too small, long records), which are parsed again in two passes."""
import numpy as np
import pytest

import cases
import golden_io
from dsp_amd import capi, gen, schema
from util import assert_batches_equal

pytestmark = pytest.mark.gpu

ONE = capi.MDC_F_ONE_PASS


def parses(p, data, n=3):
    """n parses with one parser; returns the last result and every parse's timing record."""
    got, tm = None, []
    for _ in range(n):
        got = p.parse_host(data).to_host()
        tm.append(p.timing())
    return got, tm


@pytest.mark.parametrize("case", golden_io.xml_cases(), ids=lambda c: c["name"])
def test_golden_xml(case):
    print("case", case["name"], flush=True)
    p = capi.Parser.xml(case["cols"], typed=case["typed"], endbz=case["endbz"], carry=case["carry"], flags=ONE)
    try:
        for k in range(3):
            got = p.parse_host(case["data"]).to_host()
            assert_batches_equal(got, case["expect"], f"{case['name']} parse {k}")
    finally:
        p.close()


@pytest.mark.parametrize("rows", [1, 95, 96, 1000, 20000, 300000])
def test_mind_files_take_one_pass(port, rows):
    data, n = gen.mind_xml(rows=rows)
    exp = port.parse_xml(data, schema.ZHOBTMIND)
    p = capi.Parser.xml(schema.ZHOBTMIND, flags=ONE)
    got, tm = parses(p, data)
    p.close()
    assert got.rows == n
    assert_batches_equal(got, exp, f"mind rows={rows}")
    assert tm[0].one_pass == 0  # (the first parse builds the template and learns the sizes)
    if rows >= 1000:  # (an input of a few tiles, of which the first begins with the file's opening line, keeps no template)
        assert tm[1].one_pass == 1 and tm[2].one_pass == 1 and tm[2].one_pass_redone == 0


def test_typed_time_and_binary32_columns(port):
    import os
    import time
    data, n = gen.mind_xml(rows=5000)
    typed = schema.default_typed(schema.ZHOBTMIND)
    typed[1] = schema.T_TIME
    typed[2] = schema.T_F32
    old = os.environ.get("TZ")
    os.environ["TZ"] = "UTC-8"  # (the checker's mktime runs in the process's zone: eight hours east of UTC)
    time.tzset()
    try:
        exp = port.parse_xml(data, schema.ZHOBTMIND, typed=typed)
    finally:
        if old is None:
            os.environ.pop("TZ", None)
        else:
            os.environ["TZ"] = old
        time.tzset()
    p = capi.Parser.xml(schema.ZHOBTMIND, typed=typed, flags=ONE, tz_offset=28800)
    got, tm = parses(p, data)
    p.close()
    assert_batches_equal(got, exp, "time and binary32 columns")
    assert tm[2].one_pass == 1


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("schema_name", ["SMALL", "LONGNAMES", "ODD"])
def test_mutated_xml(port, seed, schema_name):
    cols = getattr(cases, schema_name)
    all_t = list(range(1, 8))
    typed = [all_t[(seed + i) % len(all_t)] for i in range(len(cols))]
    data = cases.mutated_xml(300 + seed, 700, cols, damage=0.1 + 0.1 * seed)
    for carry in (True, False):
        print("mutated", schema_name, seed, carry, flush=True)
        exp = port.parse_xml(data, cols, typed=typed, carry=carry)
        p = capi.Parser.xml(cols, typed=typed, carry=carry, flags=ONE)
        got, _ = parses(p, data)
        p.close()
        assert_batches_equal(got, exp, f"mutated {schema_name} {seed} carry={carry}")


def test_some_tiles_left_to_the_full_write_kernel(port):
    """Regular records with, here and there, a record without one of its fields, a long cell, stray
    bytes in a number and a NUL byte: those tiles do not take the straight-line write."""
    data, n = gen.mind_xml(rows=30000)
    text = bytearray(data.tobytes())
    recs = bytes(text).split(b"<endl/>\n")
    odd = {700: lambda r: r.replace(b"<wd>", b"<xx>"), 5000: lambda r: r.replace(b"<p>", b"<p>" + b"9" * 30),
           9000: lambda r: r.replace(b"</t>", b" C</t>"), 15000: lambda r: r.replace(b"<obtid>", b"<obtid>\0")}
    for k, f in odd.items():
        recs[k] = f(recs[k])
    data = np.frombuffer(b"<endl/>\n".join(recs), np.uint8)
    exp = port.parse_xml(data, schema.ZHOBTMIND)
    p = capi.Parser.xml(schema.ZHOBTMIND, flags=ONE)
    got, tm = parses(p, data)
    p.close()
    assert_batches_equal(got, exp, "tiles of both kinds")
    assert tm[2].one_pass == 1 and tm[2].full_write == 1 and tm[2].one_pass_redone == 0


def test_outputs_guessed_too_small_are_parsed_again(port):
    """The second input holds more cell bytes per input byte than the first: the arrays sized from the
    first parse do not hold it, the one-pass parse stores nothing out of bounds and two passes follow."""
    lean = b"".join(b"<a>%d</a><ab>x</ab><b>1</b>%s<endl/>\n" % (i, b" " * 130) for i in range(40000))
    dense = b"".join(b"<a>%d</a><ab>%s</ab><b>1.5</b><endl/>\n" % (i, b"y" * 130) for i in range(40000))
    cols = [schema.Col("a", "number", 22), schema.Col("ab", "char", 0), schema.Col("b", "number", 22)]
    p = capi.Parser.xml(cols, flags=ONE)
    for k, text in enumerate((lean, lean, dense, dense, dense, dense)):
        data = np.frombuffer(text, np.uint8)
        got = p.parse_host(data).to_host()
        t = p.timing()
        assert_batches_equal(got, port.parse_xml(data, cols), f"parse {k}")
        if k == 1:
            assert t.one_pass == 1
        if k == 2:
            assert t.one_pass == 0 and t.one_pass_redone == 1
    assert p.timing().one_pass == 1  # (after the pause that follows a repeated parse)
    p.close()


def test_long_records_are_parsed_again(port):
    data, n = gen.mind_xml(rows=3000)
    text = data.tobytes()
    cut = text.index(b"<endl/>\n", len(text) // 2) + 8
    longrec = b"<obtid>12345</obtid><junk>" + b"j" * 40000 + b"</junk><t>2.5</t><endl/>\n"
    mixed = np.frombuffer(text[:cut] + longrec + text[cut:], np.uint8)
    p = capi.Parser.xml(schema.ZHOBTMIND, flags=ONE)
    p.parse_host(data).free()
    got = p.parse_host(mixed).to_host()
    t = p.timing()
    p.close()
    assert_batches_equal(got, port.parse_xml(mixed, schema.ZHOBTMIND), "a long record among regular ones")
    assert t.one_pass == 0 and t.one_pass_redone == 1 and t.long_records == 1


def test_many_tiles_against_the_two_pass_parse():
    """64 MiB: 4,096 tiles that wait for one another; the same bytes as the two-pass parse gives."""
    data, n = gen.mind_xml(nbytes=64 << 20)
    d = capi.DeviceBuffer(data)
    two = capi.Parser.xml(schema.ZHOBTMIND)
    exp = two.parse_device(d.ptr, d.n).to_host()
    two.close()
    p = capi.Parser.xml(schema.ZHOBTMIND, flags=ONE)
    for k in range(4):
        b = p.parse_device(d.ptr, d.n)
        got = b.to_host()
        b.free()
        assert_batches_equal(got, exp, f"parse {k}")
    assert p.timing().one_pass == 1 and p.timing().rows == n
    p.close()
    d.free()


def test_tiles_that_may_not_wait_give_up_and_two_passes_follow():
    """The wait of a tile for the tiles in front of it is bounded.  With the bound set to nothing (a test
    hook) every tile that would have to wait gives up at once, stores nothing, lets the tiles behind it go
    on, and the input is parsed again in two passes: the same batch, and no wave is left waiting."""
    import ctypes
    import os
    os.environ["MDC_TEST_HOOKS"] = "1"  # (read when a hook is first called)
    L = capi.lib()
    L.mdc_internal_spin_limit.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    data, n = gen.mind_xml(nbytes=48 << 20)
    d = capi.DeviceBuffer(data)
    p = capi.Parser.xml(schema.ZHOBTMIND, flags=ONE)
    b = p.parse_device(d.ptr, d.n)
    exp = b.to_host()
    b.free()
    capi.check(L.mdc_internal_spin_limit(p.h, 0))
    gave_up = 0
    for k in range(4):
        b = p.parse_device(d.ptr, d.n)
        got = b.to_host()
        b.free()
        t = p.timing()
        gave_up += t.one_pass_redone
        assert_batches_equal(got, exp, f"parse {k} with no waiting allowed")
    assert gave_up >= 1  # (3,072 tiles in flight at once: some tile has to wait)
    capi.check(L.mdc_internal_spin_limit(p.h, 1 << 16))
    for k in range(3):
        b = p.parse_device(d.ptr, d.n)
        got = b.to_host()
        b.free()
        assert_batches_equal(got, exp, "with the usual bound again")
    assert p.timing().one_pass == 1
    p.close()
    d.free()
