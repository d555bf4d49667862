"""The plain-C restatement (oracle/mdc_oracle.c) against the reference's own
public/_public.cpp compiled in place (oracle/_ref).  CPU only; skipped where the reference
build is absent (the GPU box)."""
import pytest

import cases
from dsp_amd import gen, schema
from oracle import oracle_lib as ol
from util import assert_batches_equal

ALL_T = [ol.T_I32, ol.T_U32, ol.T_I64, ol.T_U64, ol.T_F64, ol.T_F32, ol.T_BOOL]


def test_kinds(port, reference):
    assert port.kind == "port" and reference.kind == "reference"


@pytest.mark.parametrize("case", cases.probe_xml(), ids=lambda c: c[0])
@pytest.mark.parametrize("carry", [True, False])
def test_probe_xml(port, reference, case, carry):
    name, data, cols, endbz = case
    a = port.parse_xml(data, cols, endbz=endbz, carry=carry)
    b = reference.parse_xml(data, cols, endbz=endbz, carry=carry)
    assert_batches_equal(a, b, name)
    sa, la = port.split_records(data, endbz)
    sb, lb = reference.split_records(data, endbz)
    assert (sa == sb).all() and (la == lb).all()


def test_ini_file(port, reference):
    a = port.parse_xml(cases.INI_XMLTODB, cases.INI_COLS, carry=False)
    b = reference.parse_xml(cases.INI_XMLTODB, cases.INI_COLS, carry=False)
    assert_batches_equal(a, b, "ini")
    assert a.rows == 2
    assert a.cell(0, 1) == b"T_ZHOBTCODE1" and a.cell(1, 1) == b"T_ZHOBTMIND1"
    assert a.cell(0, 3) == b"delete from T_ZHOBTCODE1" and a.cell(1, 3) == b"" and not a.is_found(1, 3)


@pytest.mark.parametrize("text", cases.probe_convert(), ids=lambda t: repr(t[:24]))
def test_probe_convert(port, reference, text):
    for t in ALL_T:
        assert port.convert(text, t) == reference.convert(text, t), (text, t)


@pytest.mark.parametrize("text", cases.probe_trim(), ids=repr)
def test_probe_trim(port, reference, text):
    for which in (0, 1, 2):
        for cc in (" ", ","):
            assert port.trim(text, which, cc) == reference.trim(text, which, cc), (text, which, cc)


def test_picknumber(port, reference):
    import random
    rng = random.Random(7)
    for _ in range(300):
        s = bytes(rng.choice(b"0123456789+-.eE abc\x00\xff\xb2") for _ in range(rng.randrange(0, 40)))
        for bs in (False, True):
            for bd in (False, True):
                assert port.picknumber(s, bs, bd) == reference.picknumber(s, bs, bd)


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("schema_name", ["SMALL", "LONGNAMES", "ODD"])
def test_mutated_xml(port, reference, seed, schema_name):
    cols = getattr(cases, schema_name)
    data = cases.mutated_xml(seed, 60, cols, damage=0.2 + 0.05 * (seed % 8))
    typed = [ALL_T[(seed + i) % len(ALL_T)] for i in range(len(cols))]
    for carry in (True, False):
        for ty in (None, typed):
            a = port.parse_xml(data, cols, typed=ty, carry=carry)
            b = reference.parse_xml(data, cols, typed=ty, carry=carry)
            assert_batches_equal(a, b, f"mutated seed={seed} carry={carry}")


@pytest.mark.parametrize("seed", range(8))
def test_soup_xml(port, reference, seed):
    for cols in (cases.SMALL, cases.ODD):
        data = cases.soup_xml(seed, 1500, cols)
        for carry in (True, False):
            a = port.parse_xml(data, cols, carry=carry)
            b = reference.parse_xml(data, cols, carry=carry)
            assert_batches_equal(a, b, f"soup seed={seed}")


def test_synthetic_files(port, reference):
    data, n = gen.mind_xml(rows=1000)
    a = port.parse_xml(data, schema.ZHOBTMIND)
    b = reference.parse_xml(data, schema.ZHOBTMIND)
    assert a.rows == n == 1000
    assert_batches_equal(a, b, "mind")
    data, n = gen.code_xml()
    a = port.parse_xml(data, schema.ZHOBTCODE)
    b = reference.parse_xml(data, schema.ZHOBTCODE)
    assert a.rows == n == 839
    assert_batches_equal(a, b, "code")


@pytest.mark.parametrize("case", cases.probe_csv(), ids=lambda c: c[0])
def test_probe_csv(port, reference, case):
    name, data, sep, bdel = case
    for typed in (None, [ALL_T[i % len(ALL_T)] for i in range(8)]):
        for ilen in (None, [0, 2, 0, 1, 0, 3, 0, 0]):
            a = port.parse_csv(data, sep, 8, ilen=ilen, typed=typed, bdelspace=bdel)
            b = reference.parse_csv(data, sep, 8, ilen=ilen, typed=typed, bdelspace=bdel)
            assert_batches_equal(a, b, name)


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("sep", [",", "~!~", " ", "||"])
def test_fuzz_csv(port, reference, seed, sep):
    data = cases.fuzz_csv(seed, 80, sep, 6)
    typed = [ALL_T[(seed + i) % len(ALL_T)] for i in range(6)]
    for bdel in (False, True):
        a = port.parse_csv(data, sep, 6, ilen=[0, 0, 3, 0, 0, 1], typed=typed, bdelspace=bdel)
        b = reference.parse_csv(data, sep, 6, ilen=[0, 0, 3, 0, 0, 1], typed=typed, bdelspace=bdel)
        assert_batches_equal(a, b, f"csv seed={seed} sep={sep!r}")


def test_synthetic_csv(port, reference):
    data, n = gen.mind_csv(rows=500)
    a = port.parse_csv(data, ",", 10, typed=schema.MIND_CSV_TYPED)
    b = reference.parse_csv(data, ",", 10, typed=schema.MIND_CSV_TYPED)
    assert a.rows == n
    assert_batches_equal(a, b, "mind csv")
