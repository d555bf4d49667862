"""Diagnostic: runs pytest (or a script) against another build of the library.
usage: with_lib.py <library file in dsp_amd/> <pytest args...>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "dsp_amd", sys.argv[1])
import pytest  # noqa: E402

sys.exit(pytest.main(sys.argv[2:]))
