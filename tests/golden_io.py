"""Loads tests/golden/*.json (outputs of the reference's own code, see make_golden.py)."""
import base64
import json
import os

import numpy as np

from dsp_amd.schema import Col
from oracle.oracle_lib import HostBatch

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(HERE, name)) as f:
        return json.load(f)


def expect_batch(e, ncols, typed_present):
    rows = e["rows"]
    off, data, val = [], [], []
    for c in range(ncols):
        cells = [bytes.fromhex(e["cells"][r][c]) for r in range(rows)]
        o = np.zeros(rows + 1, dtype=np.uint32)
        if rows:
            o[1:] = np.cumsum([len(x) for x in cells], dtype=np.uint64).astype(np.uint32)
        off.append(o)
        data.append(np.frombuffer(b"".join(cells), dtype=np.uint8).copy())
        if rows and e["val"][0][c] is not None:
            val.append(np.array([int(e["val"][r][c], 16) for r in range(rows)], dtype=np.uint64))
        elif rows == 0 and typed_present[c]:
            val.append(np.zeros(0, dtype=np.uint64))
        else:
            val.append(None)
    w = max(1, (ncols + 31) // 32)
    valid = np.array(e["valid"], dtype=np.uint32).reshape(rows, w)
    found = np.array(e["found"], dtype=np.uint32).reshape(rows, w)
    return HostBatch(rows, ncols, np.array(e["rec_off"], dtype=np.uint64), off, data, val, valid, found)


def xml_cases():
    out = []
    for c in _load("xml_cases.json"):
        cols = [Col(n, t, l) for n, t, l in c["cols"]]
        typed = c["typed"]
        tp = [(typed[i] != 0) if typed is not None else cols[i].datatype in ("number", "date")
              for i in range(len(cols))]
        out.append(dict(name=c["name"], data=base64.b64decode(c["input"]), endbz=c["endbz"], cols=cols,
                        typed=typed, carry=c["carry"], expect=expect_batch(c["expect"], len(cols), tp)))
    return out


def csv_cases():
    out = []
    for c in _load("csv_cases.json"):
        typed = c["typed"]
        tp = [(typed[i] != 0) if typed is not None else False for i in range(c["nfields"])]
        out.append(dict(name=c["name"], data=base64.b64decode(c["input"]), sep=c["sep"], nfields=c["nfields"],
                        ilen=c["ilen"], typed=typed, bdelspace=c["bdelspace"],
                        expect=expect_batch(c["expect"], c["nfields"], tp)))
    return out


def scalar_cases():
    return _load("scalar_cases.json")


def format_cases():
    """Writer side: rows of cells and the text the reference's cofile wrote for them (None = it fails)."""
    out = []
    for c in _load("format_cases.json"):
        rows = [[bytes.fromhex(x) for x in row] for row in c["cells"]]
        e = c["expect"]
        out.append(dict(name=c["name"], names=c["fieldname"], fieldlen=c["fieldlen"], maxcount=c["maxcount"],
                        rows=rows, text=None if e is None else base64.b64decode(e["text"]),
                        file_off=None if e is None else e["file_off"]))
    return out
