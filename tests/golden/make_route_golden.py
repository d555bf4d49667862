"""Generates tests/golden/route_cases.json from the reference's own code (matchstr as it is;
loadxmltotable, crtsql and the bind order through oracle/ref_harness.cpp, which runs their loops
around the reference's readline, getxmlbuffer, sformat and deleterchr).

Runs only where oracle/_ref/libmdc_ref.so exists (the build container).  Data only: inputs and
the outputs the reference produced.

    python tests/golden/make_route_golden.py
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
from oracle import oracle_lib as ol  # noqa: E402


def main():
    ref = ol.load("reference")
    assert ref.kind == "reference"
    b64 = lambda b: base64.b64encode(bytes(b)).decode()
    out = {"matchstr": [], "load": [], "sql": []}
    for name, rules in cases.match_cases():
        out["matchstr"].append([name, rules, ref.matchstr(name, rules)])
    for name, data in cases.ini_cases():
        ents = ref.load_xmltotable(data)
        out["load"].append({"name": name, "input": b64(data),
                            "entries": [[b64(f), b64(t), u, b64(s)] for f, t, u, s in ents]})
    for name, entry, cols in cases.sql_cases():
        ins, upd = ref.crtsql(entry, cols)
        bi, bu = ref.bind_plan(entry, cols)
        out["sql"].append({"name": name, "entry": list(entry),
                           "cols": [[c.colname, c.datatype, c.collen, c.pkseq] for c in cols],
                           "insertsql": ins.decode(), "updatesql": None if upd is None else upd.decode(),
                           "bind_insert": bi, "bind_update": bu})
    fn = os.path.join(HERE, "route_cases.json")
    with open(fn, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print(fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
