"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, validates
arguments, and refuses to parse without a GPU (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

from dsp_amd import capi, schema

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "mdc_parse.h")).read()
    declared = set(re.findall(r"\b(mdc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    L = capi.lib()
    for name in capi.EXPORTS:
        assert hasattr(L, name), name


def test_version_and_strerror():
    L = capi.lib()
    assert b"gfx950" in L.mdc_version()
    assert L.mdc_strerror(capi.MDC_ERR_NO_DEVICE).startswith(b"no HIP device")
    assert L.mdc_strerror(5) == b"open file failed"


def test_col_struct_layout():
    assert ctypes.sizeof(schema.CCol) == 72
    assert schema.CCol.collen.offset == 64 and schema.CCol.pkseq.offset == 68


def test_schema_validation_happens_before_the_device_is_needed():
    with pytest.raises(capi.MdcError) as e:
        capi.Parser.xml([schema.Col("a>b", "char", 0)])
    assert e.value.code == capi.MDC_ERR_ARG
    with pytest.raises(capi.MdcError) as e:
        capi.Parser.xml([schema.Col("a", "char", 0)], endbz="x" * 256)
    assert e.value.code == capi.MDC_ERR_ARG
    with pytest.raises(capi.MdcError) as e:
        capi.Parser.csv("", 3)
    assert e.value.code == capi.MDC_ERR_ARG
    with pytest.raises(capi.MdcError) as e:
        capi.Parser.xml([schema.Col("a", "char", 0)] * 257)
    assert e.value.code == capi.MDC_ERR_ARG


def test_no_gpu_means_no_parse():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.MdcError) as e:
        capi.Parser.xml(schema.ZHOBTMIND)
    assert e.value.code == capi.MDC_ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_normalise_matches_reference_rules():
    # tools/_tools.cpp:47-71
    n = schema.normalise
    assert n("OBTID", "VARCHAR2", 10) == schema.Col("obtid", "char", 10)
    assert n("r", "rowid", 10).collen == 18 and n("r", "rowid", 10).datatype == "char"
    assert n("d", "date", 7) == schema.Col("d", "date", 14)
    for t in ("number", "integer", "float"):
        assert n("x", t, 5) == schema.Col("x", "number", 22)
    assert n("b", "blob", 4000) is None and n("t", "timestamp(6)", 11) is None


def test_stream_entry_points_check_their_arguments():
    import ctypes
    L = capi.lib()
    st = ctypes.c_void_p()
    assert L.mdc_stream_open(None, None, 0, 1, ctypes.byref(st)) == capi.MDC_ERR_ARG
    assert L.mdc_stream_next(None, None, None) == capi.MDC_ERR_ARG
    assert L.mdc_stream_bytes(None) == 0
    L.mdc_stream_close(None)  # a no-op
