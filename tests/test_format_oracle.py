"""Writer side on the CPU: the C restatement of the extractor's record formatter
(tools/dminingoracle.cpp:196-256) against the golden texts the reference's own cofile wrote,
and against the reference itself where it is built."""
import glob
import os

import numpy as np
import pytest

import cases
import golden_io
from dsp_amd import capi


@pytest.mark.parametrize("case", golden_io.format_cases(), ids=lambda c: c["name"])
def test_port_matches_golden_text(port, case):
    off, data = cases.columns_of(case["rows"], len(case["names"]))
    got = port.format_xml(case["names"], off, data, len(case["rows"]), case["maxcount"], fieldlen=case["fieldlen"])
    if case["text"] is None:
        assert got is None
        return
    assert bytes(got[0]) == case["text"]
    assert [int(x) for x in got[1]] == case["file_off"]


def test_golden_covers_the_reference_quirks():
    by = {c["name"]: c for c in golden_io.format_cases()}
    # no rows: the reference fails with maxcount 0 (closes a file it never opened), writes nothing otherwise
    assert by["zero_rows/maxcount=0"]["text"] is None
    assert by["zero_rows/maxcount=5"]["text"] == b"" and by["zero_rows/maxcount=5"]["file_off"] == [0]
    # a row count that is a multiple of maxcount leaves no empty last file
    assert by["mind/maxcount=7"]["file_off"][-1] == len(by["mind/maxcount=7"]["text"])
    assert len(by["mind/maxcount=7"]["file_off"]) == 2 and len(by["mind/maxcount=1"]["file_off"]) == 8
    # values end at their first NUL and at fieldlen bytes
    assert by["cuts/fieldlen"]["text"].startswith(b"<data>\n<a>abcd</a><bb>12</bb><ccc></ccc><endl/>\n<a></a><bb>xxxxxxxxxx</bb>")


@pytest.mark.parametrize("seed", range(12))
def test_port_matches_reference_on_random_columns(port, reference, seed):
    nf = 1 + seed % 6
    rows = cases.random_rows(900 + seed, 40 + 13 * seed, nf, maxlen=5 + seed, nul=0.08 * (seed % 3), big=0.03 * (seed % 4))
    off, data = cases.columns_of(rows, nf)
    names = [("n%d" % i) * (1 + (i + seed) % 4) for i in range(nf)]
    fl = None if seed % 3 == 0 else [(5 * i + seed) % 9 for i in range(nf)]
    for mc in (0, 1, 7, len(rows), len(rows) + 1):
        a = port.format_xml(names, off, data, len(rows), mc, fieldlen=fl)
        b = reference.format_xml(names, off, data, len(rows), mc, fieldlen=fl)
        assert bytes(a[0]) == bytes(b[0]), (seed, mc)
        assert list(a[1]) == list(b[1])


def test_port_writes_the_same_files_as_the_reference(port, reference, tmp_path):
    names, fl = cases.MIND_FIELDS
    rows = cases.random_rows(5, 25, len(names), maxlen=8)
    off, data = cases.columns_of(rows, len(names))
    made = {}
    for kind, chk in (("port", port), ("reference", reference)):
        out = tmp_path / kind / "deep" / "dir"
        n = chk.format_xml_files(names, off, data, len(rows), 10, str(out), "ZHOBTMIND", "20240519162835", "togxpt",
                                 fieldlen=fl)
        assert n == 3
        files = sorted(glob.glob(str(out / "*")))
        made[kind] = [(os.path.basename(f), open(f, "rb").read()) for f in files]
    assert made["port"] == made["reference"]
    assert [n for n, _ in made["port"]] == [f"ZHOBTMIND_20240519162835_togxpt_{k}.xml" for k in (1, 2, 3)]


def test_formatted_text_parses_back_to_the_same_cells(port):
    from dsp_amd.schema import Col
    names = ["obtid", "t", "note"]
    rows = [[b"58015", b"12.7", b"a b"], [b"58016", b"", b"x"], [b"", b"-3", b""]]
    off, data = cases.columns_of(rows, 3)
    text, _ = port.format_xml(names, off, data, 3, 2)
    hb = port.parse_xml(bytes(text), [Col(n, "char", 0) for n in names], carry=False)
    assert hb.rows == 3
    assert [[hb.cell(r, c) for c in range(3)] for r in range(3)] == rows


# ---- the C ABI without a GPU --------------------------------------------------------------

def test_format_symbols_are_exported():
    L = capi.lib()
    for name in capi.FORMAT_EXPORTS:
        assert hasattr(L, name), name


def test_header_declares_exactly_the_exported_format_symbols():
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "mdc_format.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(mdc_[a-z_0-9]+)\s*\(", text))
    assert declared == set(capi.FORMAT_EXPORTS)


def test_formatter_argument_checks_need_no_gpu():
    import ctypes
    L = capi.lib()
    h = ctypes.c_void_p()
    one = (ctypes.c_char_p * 1)(b"a")
    assert L.mdc_formatter_create(one, None, 0, 0, ctypes.byref(h)) == capi.MDC_ERR_ARG
    assert L.mdc_formatter_create(one, None, 257, 0, ctypes.byref(h)) == capi.MDC_ERR_ARG
    assert L.mdc_formatter_create(None, None, 1, 0, ctypes.byref(h)) == capi.MDC_ERR_ARG
    assert L.mdc_formatter_create(one, None, 1, 0, None) == capi.MDC_ERR_ARG
    empty = (ctypes.c_char_p * 1)(b"")
    assert L.mdc_formatter_create(empty, None, 1, 0, ctypes.byref(h)) == capi.MDC_ERR_ARG
    long = (ctypes.c_char_p * 1)(b"x" * 256)
    assert L.mdc_formatter_create(long, None, 1, 0, ctypes.byref(h)) == capi.MDC_ERR_ARG
    neg = (ctypes.c_int * 1)(-1)
    assert L.mdc_formatter_create(one, neg, 1, 0, ctypes.byref(h)) == capi.MDC_ERR_ARG
    many = (ctypes.c_char_p * 128)(*[b"y" * 200] * 128)
    assert L.mdc_formatter_create(many, None, 128, 0, ctypes.byref(h)) == capi.MDC_ERR_ARG
    assert b"tag text" in L.mdc_last_error()
    assert L.mdc_format_device(None, None, None, 0, 0, ctypes.byref(h)) == capi.MDC_ERR_ARG
    assert L.mdc_text_files(None) == -1 and L.mdc_text_bytes(None) == 0
    L.mdc_text_free(None)
    L.mdc_formatter_destroy(None)


def test_formatter_without_a_gpu_fails_loudly():
    import ctypes
    L = capi.lib()
    if L.mdc_device_count() > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    one = (ctypes.c_char_p * 1)(b"a")
    assert L.mdc_formatter_create(one, None, 1, 0, ctypes.byref(h)) == capi.MDC_ERR_NO_DEVICE
    assert not h.value
