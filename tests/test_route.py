"""Routing of a file to its table, matchstr, SQL text and bind order (include/mdc_route.h):
the library against the fixtures the reference's code produced (tests/golden/route_cases.json),
against the plain-C checker, and the checker against the reference build where it exists.
Host code only: runs without a GPU."""
import base64
import json
import os
import random
import re

import pytest

import cases
from dsp_amd import capi, route
from dsp_amd.schema import Col

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "route_cases.json")))
unb64 = base64.b64decode


def gold_entries(case):
    return [(unb64(f), unb64(t), u, unb64(s)) for f, t, u, s in case["entries"]]


def gold_cols(case):
    return [Col(n, t, l, pk) for n, t, l, pk in case["cols"]]


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "mdc_route.h")).read()
    declared = set(re.findall(r"\b(mdc_[a-z0-9_]+)\s*\(", hdr)) - {"mdc_last_error", "mdc_batch_column_fixed"}
    assert declared == set(route.ROUTE_EXPORTS), declared ^ set(route.ROUTE_EXPORTS)
    L = capi.lib()
    for name in route.ROUTE_EXPORTS:
        assert hasattr(L, name), name


def test_struct_layout():
    import ctypes
    assert ctypes.sizeof(route.CXmlToTable) == 420
    assert route.CXmlToTable.uptbz.offset == 160 and route.CXmlToTable.execsql.offset == 164


# ---- against the fixtures ---------------------------------------------------------------------

def test_golden_covers_the_case_lists():
    assert [(n, r) for n, r, _ in GOLD["matchstr"]] == [list(x) and (x[0], x[1]) for x in cases.match_cases()]
    assert [c["name"] for c in GOLD["load"]] == [n for n, _ in cases.ini_cases()]
    assert [c["name"] for c in GOLD["sql"]] == [n for n, _, _ in cases.sql_cases()]


def test_matchstr_golden(port):
    for name, rules, want in GOLD["matchstr"]:
        assert route.matchstr(name, rules) == want, (name, rules)
        assert port.matchstr(name, rules) == want, (name, rules)


@pytest.mark.parametrize("case", GOLD["load"], ids=lambda c: c["name"])
def test_load_golden(case, port):
    data = unb64(case["input"])
    want = gold_entries(case)
    r = route.Route(data)
    assert len(r) == len(want) and r.entries() == want and r.truncated == 0
    assert port.load_xmltotable(data) == want


@pytest.mark.parametrize("case", GOLD["sql"], ids=lambda c: c["name"])
def test_sql_golden(case, port):
    cols = gold_cols(case)
    e = route.entry(*case["entry"])
    ins, upd = route.crtsql(e, cols)
    assert ins.decode() == case["insertsql"]
    assert (None if upd is None else upd.decode()) == case["updatesql"]
    assert route.bind_plan(e, cols) == (case["bind_insert"], case["bind_update"])
    pins, pupd = port.crtsql(tuple(case["entry"]), cols)
    assert pins.decode() == case["insertsql"] and (None if pupd is None else pupd.decode()) == case["updatesql"]
    assert port.bind_plan(tuple(case["entry"]), cols) == (case["bind_insert"], case["bind_update"])


def test_reference_samples():
    """The statements quoted in the reference's comments (tools/xmltodb.cpp:328-329, :369-370)
    for its sample table, and the routing of its sample parameter file."""
    mind = next(c for c in GOLD["sql"] if c["name"] == "zhobtmind/uptbz=1")
    assert mind["insertsql"] == ("insert into T_ZHOBTMIND1(obtid,ddatetime,t,p,u,wd,wf,r,vis,keyid) "
                                 "values(:1,to_date(:2,'yyyymmddhh24miss'),:3,:4,:5,:6,:7,:8,:9,SEQ_ZHOBTMIND1.nextval)")
    assert mind["updatesql"].startswith("update T_ZHOBTMIND1 set t=:1,p=:2,u=:3,wd=:4,wf=:5,r=:6,vis=:7")
    assert mind["updatesql"].endswith(" where 1=1 and obtid=:8 and ddatetime=to_date(:9,'yyyymmddhh24miss')")
    r = route.Route(cases.INI_XMLTODB + b"\n")
    assert r.find("ZHOBTMIND_20240519162800_togxpt_1.XML") == 1
    assert r.find("zhobtcode_20240519.xml") == 0
    assert r.find("SURF_ZH_2024.txt") == -1
    assert r.raw(1).tname == b"T_ZHOBTMIND1" and r.raw(1).uptbz == 1 and r.raw(1).execsql == b""


# ---- checker against the reference build (build container only) ---------------------------------

def _random_names(seed, n):
    rng = random.Random(seed)
    alpha = "abAB.*,_1"
    out = []
    for _ in range(n):
        name = "".join(rng.choice("abAB._1") for _ in range(rng.randint(0, 8)))
        rules = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 10)))
        out.append((name, rules))
    return out


def test_port_equals_reference(port, reference):
    for name, rules in cases.match_cases() + _random_names(1, 4000):
        assert port.matchstr(name, rules) == reference.matchstr(name, rules), (name, rules)
    for _, data in cases.ini_cases():
        assert port.load_xmltotable(data) == reference.load_xmltotable(data)
    for _, entry, cols in cases.sql_cases():
        assert port.crtsql(entry, cols) == reference.crtsql(entry, cols)
        assert port.bind_plan(entry, cols) == reference.bind_plan(entry, cols)


# ---- library against the checker on inputs beyond the fixtures -------------------------------------

def test_matchstr_random(port):
    for name, rules in _random_names(7, 20000):
        assert route.matchstr(name, rules) == port.matchstr(name, rules), (name, rules)


def test_load_random(port):
    rng = random.Random(11)
    tags = [b"<filename>", b"</filename>", b"<tname>", b"</tname>", b"<uptbz>", b"</uptbz>", b"<execsql>",
            b"</execsql>", b"<endl/>", b"\n", b"<endl/>\n", b"1", b"-", b"+2", b"*.XML", b"T_A", b" ", b"\x00", b"x,y"]
    for _ in range(300):
        data = b"".join(rng.choice(tags) for _ in range(rng.randint(0, 60)))
        r = route.Route(data)
        assert r.entries() == port.load_xmltotable(data), data


def test_values_beyond_the_arrays_are_cut(port):
    data = (b"<filename>" + b"f" * 300 + b"</filename><tname>" + b"t" * 40 + b"</tname><uptbz>1</uptbz><execsql>" +
            b"s" * 256 + b"</execsql><endl/>\n<filename>ok</filename><tname>T</tname><endl/>\n")
    r = route.Route(data)
    assert r.truncated == 3
    assert r.entries() == [(b"f" * 127, b"t" * 31, 1, b"s" * 255), (b"ok", b"T", 0, b"")]
    assert port.load_xmltotable(data) == r.entries()


def test_sql_random(port):
    rng = random.Random(5)
    names = ["a", "t", "keyid", "upttime", "obtid", "ddatetime", "v1", "v2", "x"]
    for _ in range(300):
        k = rng.randint(0, 8)
        cols = [Col(n, rng.choice(["char", "date", "number"]), rng.choice([0, 5, 14, 22]), rng.choice([0, 0, 1, 2, 3]))
                for n in rng.sample(names, k)]
        entry = ("f", rng.choice(["", "T", "T_", "T_ABC", "ZHOBTMIND"]), rng.choice([0, 1, 1, 2]), "")
        e = route.entry(*entry)
        assert route.crtsql(e, cols) == port.crtsql(entry, cols), (entry, cols)
        assert route.bind_plan(e, cols) == port.bind_plan(entry, cols)


def test_small_buffers_and_bad_arguments():
    import ctypes
    mind = next(c for c in GOLD["sql"] if c["name"] == "zhobtmind/uptbz=1")
    e = route.entry(*mind["entry"])
    with pytest.raises(capi.MdcError) as err:
        route.crtsql(e, gold_cols(mind), cap=32)
    assert err.value.code == capi.MDC_ERR_TOO_LARGE
    L = capi.lib()
    route.matchstr("a", "a")  # (sets the argument types)
    assert L.mdc_crtsql(None, None, 0, None, 0, None, None, 0, None) == capi.MDC_ERR_ARG
    assert L.mdc_route_find(None, b"x") == -1 and L.mdc_route_count(None) == -1
    assert not L.mdc_route_entry(None, 0)
    assert L.mdc_matchstr(None, b"*") == 0


def test_load_file(tmp_path, port):
    p = tmp_path / "xmltodb.xml"
    p.write_bytes(cases.INI_XMLTODB + b"\n")
    r = route.Route(path=str(p))
    assert r.entries() == port.load_xmltotable(cases.INI_XMLTODB + b"\n")
    with pytest.raises(capi.MdcError) as err:
        route.Route(path=str(tmp_path / "missing.xml"))
    assert err.value.code == capi.MDC_FILE_OPEN_FAILED
