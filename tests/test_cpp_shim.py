"""The C++ host layer (include/mdc_xmltodb.hpp) through the example ingest loop."""
import subprocess

import numpy as np
import pytest

from dsp_amd import build, gen, schema


@pytest.fixture(scope="module")
def exe():
    return build.build_examples()


def _spec(cols, typed):
    return ",".join(f"{c.colname}:{c.datatype}:{c.collen}:{t}" for c, t in zip(cols, typed))


def _expected_lines(hb, typed):
    out = []
    for r in range(hb.rows):
        parts = [str(r)]
        for c in range(hb.ncols):
            cell = hb.cell(r, c)
            parts += [str(int(hb.is_found(r, c))), str(int(hb.is_valid(r, c))), cell.hex() if cell else "-",
                      "%016x" % int(hb.val[c][r]) if typed[c] != schema.T_NONE else "-"]
        out.append(" ".join(parts))
    out.append(f"total {hb.rows} of {hb.rows}")
    return out


def test_open_failure_is_code_5_without_touching_the_gpu(exe):
    r = subprocess.run([exe, "xml", "/nonexistent/file.xml", "a:char:10"], capture_output=True, text=True)
    assert r.returncode == 5 and "cannot open" in r.stderr


def test_routing_and_sql_through_the_host_layer(exe, port, tmp_path):
    import cases
    ini = tmp_path / "xmltodb.xml"
    ini.write_bytes(cases.INI_XMLTODB + b"\n")
    cols = schema.ZHOBTMIND
    spec = ",".join(f"{c.colname}:{c.datatype}:{c.collen}:0:{c.pkseq}" for c in cols)
    r = subprocess.run([exe, "route", str(ini), "ZHOBTMIND_20240519162800_togxpt_1.XML", spec], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    entry = port.load_xmltotable(cases.INI_XMLTODB + b"\n")[1]
    ins, upd = port.crtsql(entry, cols)
    bi, bu = port.bind_plan(entry, cols)
    assert r.stdout.splitlines() == [
        "tname T_ZHOBTMIND1", "uptbz 1", "execsql ", "insert " + ins.decode(), "update " + upd.decode(),
        "bind_insert " + " ".join(map(str, bi)), "bind_update " + " ".join(map(str, bu))]
    r = subprocess.run([exe, "route", str(ini), "SURF_ZH.txt", spec], capture_output=True, text=True)
    assert r.returncode == 1  # no routing entry: _xmltodb()'s code 1
    r = subprocess.run([exe, "route", str(tmp_path / "none.xml"), "a.xml", spec], capture_output=True, text=True)
    assert r.returncode == 5


def test_usage(exe):
    assert subprocess.run([exe], capture_output=True).returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("carry", [True, False])
def test_ingest_loop_matches_the_checker(exe, port, tmp_path, carry):
    data, n = gen.mind_xml(rows=1500)
    # some records without their tags, one with the close tag in front of the open tag
    odd = b"<obtid>58015</obtid><t>1.5</t><endl/>\n</p>x<p>1013.2</p><vis>12 km</vis><endl/>\n<endl/>\n"
    data = np.concatenate([data, np.frombuffer(odd, dtype=np.uint8)])
    path = tmp_path / "ZHOBTMIND_20240519.xml"
    path.write_bytes(data.tobytes())
    cols = schema.ZHOBTMIND
    typed = schema.default_typed(cols)
    hb = port.parse_xml(data, schema.ZHOBTMIND, carry=carry)
    args = [exe, "xml", str(path), _spec(cols, typed)] + ([] if carry else ["nocarry"])
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == _expected_lines(hb, typed)


@pytest.mark.gpu
def test_delimited_loop_matches_the_checker(exe, port, tmp_path):
    data, n = gen.mind_csv(rows=2000)
    path = tmp_path / "obs.csv"
    path.write_bytes(data.tobytes())
    typed = list(schema.MIND_CSV_TYPED)
    hb = port.parse_csv(data, ",", 10, typed=typed)
    r = subprocess.run([exe, "csv", str(path), ",", "10", ",".join(map(str, typed))], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == _expected_lines(hb, typed)
