"""include/mdc_public_compat.h: the reference's declarations (public/_public.h:16-44, :65-67,
:81-122, :140-148, :401-430) under their own names and argument lists.  examples/xmltodb_compat.cpp
makes the calls of tools/xmltodb.cpp:256-268 and :492-522 as they stand; its cells are held to the
golden vectors (outputs of the reference's own code).  The per-record functions are host code and
are tested here without a GPU; the file loop (cifile, whose record split runs on the GPU) is a
GPU test."""
import subprocess

import numpy as np
import pytest

import golden_io
from dsp_amd import build, gen, schema

XML = golden_io.xml_cases()
CSV = golden_io.csv_cases()
SC = golden_io.scalar_cases()


@pytest.fixture(scope="module")
def exe():
    build.build_parse()
    return build.build_compat_example()


def _spec(cols):
    return ",".join(f"{(c.colname if isinstance(c.colname, bytes) else c.colname.encode()).hex()}:{c.datatype}:{c.collen}" for c in cols)


def _run(exe, args, stdin=""):
    r = subprocess.run([exe] + args, input=stdin, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout.splitlines(), r.stderr


def _hex(b):
    return bytes(b).hex() if len(b) else "-"


def _expected_rows(exp):
    return [" ".join([str(r)] + [_hex(exp.cell(r, c)) for c in range(exp.ncols)]) for r in range(exp.rows)]


def _splittable(case):
    # splitbuffer's semantics are the carry-forward ones; names go through a C string
    return case["carry"] and all(b"\0" not in (c.colname if isinstance(c.colname, bytes) else c.colname.encode()) for c in case["cols"])


@pytest.mark.parametrize("case", [c for c in XML if _splittable(c)], ids=lambda c: c["name"])
def test_splitbuffer_calls_over_golden_records(exe, case):
    exp = case["expect"]
    data = bytes(case["data"])
    recs = [data[int(exp.rec_off[r]):int(exp.rec_off[r + 1]) - 1] for r in range(exp.rows)]
    rc, out, err = _run(exe, ["records", _spec(case["cols"])], "".join(_hex(r) + "\n" for r in recs))
    assert rc == 0, err
    assert out == _expected_rows(exp)


@pytest.mark.parametrize("case", CSV, ids=lambda c: c["name"])
def test_ccmdstr_over_golden_lines(exe, case):
    exp = case["expect"]
    data = bytes(case["data"])
    lines = [data[int(exp.rec_off[r]):int(exp.rec_off[r + 1]) - 1] for r in range(exp.rows)]
    nf = case["nfields"]
    ilen = case["ilen"] or [0] * nf
    typed = case["typed"] or [0] * nf
    rc, out, err = _run(exe, ["csv", case["sep"].encode().hex(), str(int(case["bdelspace"])), str(nf),
                              ",".join(map(str, ilen)), ",".join(map(str, typed))],
                        "".join(_hex(l) + "\n" for l in lines))
    assert rc == 0, err
    assert len(out) == exp.rows
    for r, line in enumerate(out):
        f = line.split(" ")
        assert int(f[0]) == r
        for c in range(nf):
            found, cell, val = f[1 + 3 * c: 4 + 3 * c]
            is_found = bool((int(exp.found[r][c >> 5]) >> (c & 31)) & 1)
            assert found == str(int(is_found)), (r, c)
            assert cell == _hex(exp.cell(r, c)), (r, c)
            if typed[c] and is_found:
                is_valid = bool((int(exp.valid[r][c >> 5]) >> (c & 31)) & 1)
                assert val == ("%016x" % int(exp.val[c][r]) if is_valid else "null"), (r, c, cell)


def test_scalar_functions_against_the_goldens(exe):
    cmds, want = [], []
    for row in SC["picknumber"]:
        cmds.append(f"pick {int(row['bsigned'])} {int(row['bdot'])} {row['text'] or '-'}")
        want.append(row["out"] or "-")
    for row in SC["trim"]:
        text = bytes.fromhex(row["text"])
        cmds.append(f"trim {row['which']} {row['cc'].encode().hex()} {row['text'] or '-'}")
        # the std::string forms carry the reference's quirk; the C-string forms trim fully (up to a NUL)
        cc = row["cc"].encode()
        cstr = text.split(b"\0")[0]
        cexp = cstr.lstrip(cc) if row["which"] == 0 else cstr.rstrip(cc) if row["which"] == 1 else cstr.strip(cc)
        want.append(f"{row['out'] or '-'} {_hex(cexp)}")
    for row in SC["getxmlbuffer"]:
        cmds.append(f"gxb {row['ilen']} {row['name'].encode().hex()} {row['rec']}")
        full = None
        if row["out"] is not None:
            # the char* form takes the whole field first and clips the copy
            rec, name = bytes.fromhex(row["rec"]), row["name"].encode()
            a, b = rec.find(b"<" + name + b">"), rec.find(b"</" + name + b">")
            s = a + len(name) + 2
            full = rec[s:b] if b >= s else rec[s:]
            if row["ilen"] > 0:
                full = full[:row["ilen"]]
            full = full.split(b"\0")[0]
        want.append(f"{(row['out'] or '-') if row['out'] is not None else 'null'} kept {_hex(full) if full is not None else '-'}")
    for row in SC["convert"]:
        text = bytes.fromhex(row["text"])
        tag = next(t for t in (b"v", b"q", b"zz9", b"tag_x") if b"<" + t + b">" not in text and b"</" + t + b">" not in text)
        for t, exp in row["out"].items():
            cmds.append(f"conv {t} {tag.hex()} {row['text'] or '-'}")
            e = exp if exp is not None else "null"
            want.append(f"{e} {e}")
    cmds.append("case " + b"aZ09{}\xe4z".hex())
    want.append(b"AZ09{}\xe4Z".hex() + " " + b"az09{}\xe4z".hex())
    rc, out, err = _run(exe, ["scalar"], "\n".join(cmds) + "\n")
    assert rc == 0, err
    assert len(out) == len(want)
    for c, o, w in zip(cmds, out, want):
        assert o == w, c


def test_without_a_gpu_open_fails_like_a_file_that_cannot_be_opened(exe, tmp_path):
    from dsp_amd import capi
    if capi.lib().mdc_device_count() >= 1:
        pytest.skip("a GPU is present")
    f = tmp_path / "a.xml"
    f.write_bytes(b"<a>1</a><endl/>\n")
    rc, out, err = _run(exe, ["xml", str(f), _spec([schema.Col("a", "char", 10)])])
    assert rc == 5 and "no usable GPU" in err and out == []


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["xml", "bulk"])
@pytest.mark.parametrize("case", [c for c in XML if _splittable(c)], ids=lambda c: c["name"])
def test_file_loop_over_golden_files(exe, tmp_path, case, mode):
    exp = case["expect"]
    f = tmp_path / "in.xml"
    f.write_bytes(bytes(case["data"]))
    rc, out, err = _run(exe, [mode, str(f), _spec(case["cols"]), case["endbz"].encode().hex() or "-"])
    assert rc == 0, err
    rows = _expected_rows(exp)
    if mode == "xml":
        assert out[:-1] == rows
        # behind the last record readline leaves the rest of the file up to its last newline in the buffer
        data = bytes(case["data"])
        tail = data[int(exp.rec_off[exp.rows]):] if exp.rows else data
        tail = tail[:tail.rfind(b"\n") + 1]
        assert out[-1] == f"total {exp.rows} of {exp.rows} tail {_hex(tail)}"
    else:
        assert out == rows


@pytest.mark.gpu
def test_file_loop_over_a_large_file_and_the_missing_file(exe, tmp_path, port):
    data, n = gen.mind_xml(rows=30000)
    f = tmp_path / "ZHOBTMIND_20240519.xml"
    f.write_bytes(data.tobytes())
    exp = port.parse_xml(data, schema.ZHOBTMIND)
    rc, out, err = _run(exe, ["xml", str(f), _spec(schema.ZHOBTMIND)])
    assert rc == 0, err
    assert out[:-1] == _expected_rows(exp)
    assert out[-1] == f"total {n} of {n} tail {_hex(b'</data>' + bytes([10]))}"
    rc, out, err = _run(exe, ["xml", str(tmp_path / "none.xml"), _spec(schema.ZHOBTMIND)])
    assert rc == 5
    # an end mark the GPU splitter does not take is refused, not guessed at
    rc, out, err = _run(exe, ["xml", str(f), _spec(schema.ZHOBTMIND), (b"m" * 256).hex()])
    assert rc == 2 and "end mark" in err


@pytest.mark.gpu
@pytest.mark.parametrize("mark", [b"<endl/>\n<next/>", b"\n", b"e\n"])
def test_file_loop_with_an_end_mark_that_holds_a_newline(exe, tmp_path, port, mark):
    # cifile::readline gathers lines until the text ends with the mark (public/_public.cpp:872-889)
    cols = schema.ZHOBTMIND[:3]
    recs = [b"<obtid>%d</obtid><ddatetime>20240519%06d</ddatetime><t>%d.5\nx</t>" % (50000 + i, i, i % 40) + mark + b"\n" for i in range(4000)]
    data = b"<data>" + b"".join(recs) + b"</data>\n"
    f = tmp_path / "nl.xml"
    f.write_bytes(data)
    exp = port.parse_xml(data, cols, endbz=mark.decode())
    assert exp.rows == 4000
    rc, out, err = _run(exe, ["xml", str(f), _spec(cols), mark.hex()])
    assert rc == 0, err
    assert out[:-1] == _expected_rows(exp)
