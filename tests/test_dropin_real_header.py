"""The drop-in on the reference's own header (build container only: the reference tree is not on the GPU
box and nothing of it is kept here).  tools/patch_public_for_compat.py copies public/_public.h and
_public.cpp into a temporary directory and puts `#ifndef MDC_PARSE_COMPAT` around the hot-path
declarations and definitions (what a maintainer of the toolkit would do by hand); the patched _public.cpp is
compiled with -DMDC_PARSE_COMPAT, examples/xmltodb_compat.cpp (the calls of tools/xmltodb.cpp:256-268 and
:492-522 as they stand) is compiled against that _public.h and both are linked with -lmdc_parse.  The
program is then held to the golden vectors for everything that runs without a GPU, and uses the
toolkit's remaining own code (cdir with matchstr over the replaced ccmdstr, clogfile)."""
import os
import subprocess

import pytest

import test_compat as tc
from dsp_amd import build

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "public")), reason="the reference tree is not here")


@pytest.fixture(scope="module")
def exe_real(tmp_path_factory):
    out = tmp_path_factory.mktemp("real_header")
    lib = os.path.dirname(build.build_parse())
    subprocess.run(["python3", os.path.join(ROOT, "tools", "patch_public_for_compat.py"), REF, str(out)], check=True)
    inc = os.path.join(ROOT, "include")
    # the guards leave the reference as it was when the macro is not defined ...
    subprocess.run(["g++", "-std=c++17", "-O1", "-w", "-c", "_public.cpp", "-o", "plain.o"], cwd=out, check=True)
    # ... and with it the hot-path functions come from the library's header
    subprocess.run(["g++", "-std=c++17", "-O1", "-w", "-DMDC_PARSE_COMPAT", "-I", inc, "-c", "_public.cpp", "-o", "compat.o"],
                   cwd=out, check=True)
    exe = str(out / "xmltodb_real")
    subprocess.run(["g++", "-std=c++17", "-O1", "-w", "-DMDC_PARSE_COMPAT", "-DMDC_WITH_REFERENCE_HEADER", "-I", str(out), "-I", inc,
                    os.path.join(ROOT, "examples", "xmltodb_compat.cpp"), "compat.o", "-o", exe, "-L", lib, "-lmdc_parse",
                    "-Wl,-rpath," + lib], cwd=out, check=True)
    return exe


def test_the_replaced_definitions_are_gone_from_the_toolkit_object(exe_real):
    d = os.path.dirname(exe_real)
    def syms(obj):
        out = subprocess.run(["nm", "-C", "--defined-only", os.path.join(d, obj)], capture_output=True, text=True, check=True).stdout
        # (strong definitions only: the header's inline functions show up as weak ones where the toolkit uses them)
        return {line.split(" ", 2)[2] for line in out.splitlines() if line.count(" ") >= 2 and line.split(" ", 2)[1] == "T"}
    plain, compat = syms("plain.o"), syms("compat.o")
    gone = [s for s in plain - compat if s.startswith("idc::")]
    for name in ("idc::cifile::readline", "idc::getxmlbuffer", "idc::picknumber", "idc::ccmdstr::splittocmd", "idc::deleterchr"):
        assert any(s.startswith(name) for s in gone), name
    # what the toolkit keeps of its own is still there
    for name in ("idc::cdir::opendir", "idc::matchstr", "idc::clogfile::open", "idc::ctcpclient::connect"):
        assert any(s.startswith(name) for s in compat), name


@pytest.mark.parametrize("case", [c for c in tc.XML if tc._splittable(c)][:40], ids=lambda c: c["name"])
def test_splitbuffer_calls(exe_real, case):
    tc.test_splitbuffer_calls_over_golden_records(exe_real, case)


@pytest.mark.parametrize("case", tc.CSV[:20], ids=lambda c: c["name"])
def test_ccmdstr_calls(exe_real, case):
    tc.test_ccmdstr_over_golden_lines(exe_real, case)


def test_scalar_functions(exe_real):
    tc.test_scalar_functions_against_the_goldens(exe_real)


def test_the_toolkits_own_directory_listing_and_log(exe_real, tmp_path):
    for name, text in (("b.xml", "22"), ("A.XML", "1"), ("c.txt", "333"), ("zz.Xml", "4444")):
        (tmp_path / name).write_text(text)
    log = tmp_path / "x.log"
    r = subprocess.run([exe_real, "dirlog", str(tmp_path), str(log)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["A.XML 1", "b.xml 2", "zz.Xml 4"]  # "*.XML" matches without regard to case, sorted
    assert [l.split(" ", 2)[2] for l in log.read_text().splitlines()] == ["A.XML 1", "b.xml 2", "zz.Xml 4"]
