"""Generates tests/golden/*.json from the reference's own code.

Runs only where oracle/_ref/libmdc_ref.so exists, i.e. in the build container where
oracle/Makefile compiled the reference's public/_public.cpp in place.  The fixtures are data
only: input bytes (base64) and the outputs the reference produced for them (hex).

    python tests/golden/make_golden.py
"""
import base64
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from dsp_amd import gen, schema  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402

ALL_T = [ol.T_I32, ol.T_U32, ol.T_I64, ol.T_U64, ol.T_F64, ol.T_F32, ol.T_BOOL]


def cols_json(cols):
    return [[c.colname, c.datatype, c.collen] for c in cols]


def batch_json(hb):
    return {
        "rows": hb.rows,
        "rec_off": [int(x) for x in hb.rec_off],
        "cells": [[hb.cell(r, c).hex() for c in range(hb.ncols)] for r in range(hb.rows)],
        "val": [[None if hb.val[c] is None else f"{int(hb.val[c][r]):016x}" for c in range(hb.ncols)]
                for r in range(hb.rows)],
        "valid": [[int(w) for w in hb.valid[r]] for r in range(hb.rows)],
        "found": [[int(w) for w in hb.found[r]] for r in range(hb.rows)],
    }


def main():
    ref = ol.load("reference")
    assert ref.kind == "reference"

    xml = []

    def add_xml(name, data, cols, endbz="<endl/>", typed=None, carry=True):
        hb = ref.parse_xml(data, cols, typed=typed, endbz=endbz, carry=carry)
        xml.append({"name": name, "input": base64.b64encode(bytes(data)).decode(), "endbz": endbz,
                    "cols": cols_json(cols), "typed": typed, "carry": carry, "expect": batch_json(hb)})

    for name, data, cols, endbz in cases.probe_xml():
        for carry in (True, False):
            add_xml(f"{name}/carry={int(carry)}", data, cols, endbz, None, carry)
    add_xml("ini_xmltodb", cases.INI_XMLTODB, cases.INI_COLS, carry=False)
    for seed in range(4):
        for nm in ("SMALL", "LONGNAMES", "ODD"):
            cols = getattr(cases, nm)
            typed = [ALL_T[(seed + i) % len(ALL_T)] for i in range(len(cols))]
            add_xml(f"mutated/{nm}/{seed}", cases.mutated_xml(seed, 25, cols, damage=0.25 + 0.1 * seed),
                    cols, typed=typed, carry=bool(seed & 1) or True)
            add_xml(f"soup/{nm}/{seed}", cases.soup_xml(seed, 400, cols), cols, carry=not (seed & 1))
    data, _ = gen.mind_xml(rows=120)
    add_xml("zhobtmind_120", data, schema.ZHOBTMIND)
    data, _ = gen.code_xml(rows=60)
    add_xml("zhobtcode_60", data, schema.ZHOBTCODE)

    csv = []

    def add_csv(name, data, sep, nfields, ilen, typed, bdel):
        hb = ref.parse_csv(data, sep, nfields, ilen=ilen, typed=typed, bdelspace=bdel)
        csv.append({"name": name, "input": base64.b64encode(bytes(data)).decode(), "sep": sep,
                    "nfields": nfields, "ilen": ilen, "typed": typed, "bdelspace": bdel,
                    "expect": batch_json(hb)})

    for name, data, sep, bdel in cases.probe_csv():
        add_csv(name, data, sep, 8, None, None, bdel)
        add_csv(name + "/typed", data, sep, 8, [0, 2, 0, 1, 0, 3, 0, 0], [ALL_T[i % 7] for i in range(8)], bdel)
    for seed in range(4):
        for sep in (",", "~!~"):
            add_csv(f"fuzz/{sep}/{seed}", cases.fuzz_csv(seed, 30, sep, 6), sep, 6, [0, 0, 3, 0, 0, 1],
                    [ALL_T[(seed + i) % 7] for i in range(6)], bool(seed & 1))
    data, _ = gen.mind_csv(rows=120)
    add_csv("mind_csv_120", data, ",", 10, None, schema.MIND_CSV_TYPED, False)

    scalar = {"convert": [], "trim": [], "picknumber": [], "getxmlbuffer": []}
    for text in cases.probe_convert():
        row = {"text": text.hex(), "out": {}}
        for t in ALL_T:
            ok, v = ref.convert(text, t)
            row["out"][str(t)] = f"{v:016x}" if ok else None
        scalar["convert"].append(row)
    for text in cases.probe_trim():
        for which in (0, 1, 2):
            for cc in (" ", ","):
                scalar["trim"].append({"text": text.hex(), "which": which, "cc": cc,
                                       "out": ref.trim(text, which, cc).hex()})
    for text in cases.probe_convert():
        for bs in (False, True):
            for bd in (False, True):
                scalar["picknumber"].append({"text": text.hex(), "bsigned": bs, "bdot": bd,
                                             "out": ref.picknumber(text, bs, bd).hex()})
    for rec, name, ilen in [(b"</a>zz<a>hello</a>tail", "a", 3), (b"<a>hello</a>", "a", 3),
                            (b"<a>hello</a>", "a", 0), (b"<ab>1</ab>", "a", 0), (b"<a><a>1</a></a>", "a", 0),
                            (b"<a>1", "a", 0), (b"1</a>", "a", 0), (b"<a>1</a><a>2</a>", "a", 0),
                            (b"<size>18348</size>", "size", 0)]:
        ok, v = ref.getxmlbuffer(rec, name, ilen)
        scalar["getxmlbuffer"].append({"rec": rec.hex(), "name": name, "ilen": ilen,
                                       "out": v.hex() if ok else None})

    fmt = []
    for name, names, fieldlen, rows, maxcount in cases.format_cases():
        off, data = cases.columns_of(rows, len(names))
        r = ref.format_xml(names, off, data, len(rows), maxcount, fieldlen=fieldlen)
        fmt.append({"name": name, "fieldname": names, "fieldlen": fieldlen, "maxcount": maxcount,
                    "cells": [[c.hex() for c in row] for row in rows],
                    "expect": None if r is None else {"text": base64.b64encode(bytes(r[0])).decode(),
                                                      "file_off": [int(x) for x in r[1]]}})

    for fn, obj in (("xml_cases.json", xml), ("csv_cases.json", csv), ("scalar_cases.json", scalar),
                    ("format_cases.json", fmt)):
        with open(os.path.join(HERE, fn), "w") as f:
            json.dump(obj, f, indent=0, separators=(",", ":"))
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
