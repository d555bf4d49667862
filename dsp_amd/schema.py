"""Column schemas for the xmltodb path.

`Col` mirrors ``ctcols::st_col`` (reference tools/_tools.h:21-27) and `normalise` restates
the data-type normalisation of ``ctcols::allcols`` (reference tools/_tools.cpp:47-71): the
schema that the reference reads from USER_TAB_COLUMNS is passed in by the caller here.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

# typed conversions (values shared with include/mdc_parse.h and oracle/mdc_oracle.h)
T_NONE, T_I32, T_U32, T_I64, T_U64, T_F64, T_F32, T_BOOL, T_TIME = range(9)


class CCol(ctypes.Structure):
    """C layout of one column: char colname[32]; char datatype[32]; int collen; int pkseq."""

    _fields_ = [
        ("colname", ctypes.c_char * 32),
        ("datatype", ctypes.c_char * 32),
        ("collen", ctypes.c_int),
        ("pkseq", ctypes.c_int),
    ]


@dataclass(frozen=True)
class Col:
    colname: str
    datatype: str  # "char" | "date" | "number"
    collen: int
    pkseq: int = 0


def normalise(colname: str, oracle_type: str, data_length: int, pkseq: int = 0) -> Optional[Col]:
    """tools/_tools.cpp:47-71: map an Oracle data type to char/date/number and fix collen.

    Returns None for a type the reference drops from the column list (:66-68)."""
    t = oracle_type.lower()
    name = colname.lower()
    if t in ("char", "nchar", "varchar2", "nvarchar2"):
        return Col(name, "char", int(data_length), pkseq)
    if t == "rowid":
        return Col(name, "char", 18, pkseq)
    if t == "date":
        return Col(name, "date", 14, pkseq)
    if t in ("number", "integer", "float"):
        return Col(name, "number", 22, pkseq)
    return None


def c_cols(cols: Sequence[Col]):
    arr = (CCol * max(1, len(cols)))()
    for i, c in enumerate(cols):
        nm = c.colname.encode() if isinstance(c.colname, str) else bytes(c.colname)
        dt = c.datatype.encode()
        if len(nm) > 31 or len(dt) > 31:
            raise ValueError("colname/datatype longer than 31 bytes")
        arr[i].colname = nm
        arr[i].datatype = dt
        arr[i].collen = c.collen
        arr[i].pkseq = c.pkseq
    return arr


def default_typed(cols: Iterable[Col]) -> List[int]:
    out = []
    for c in cols:
        out.append(T_F64 if c.datatype == "number" else T_I64 if c.datatype == "date" else T_NONE)
    return out


# T_ZHOBTMIND1 as implied by the reference's sample SQL (tools/xmltodb.cpp:328-329,
# tools/dminingoracle.cpp:356-357): obtid varchar2(10), ddatetime date, seven numbers,
# upttime date, keyid number.
ZHOBTMIND = [
    normalise("obtid", "varchar2", 10, 1),
    normalise("ddatetime", "date", 7, 2),
    normalise("t", "number", 22),
    normalise("p", "number", 22),
    normalise("u", "number", 22),
    normalise("wd", "number", 22),
    normalise("wf", "number", 22),
    normalise("r", "number", 22),
    normalise("vis", "number", 22),
    normalise("upttime", "date", 7),
    normalise("keyid", "number", 22),
]

# T_ZHOBTCODE1 (tools/dminingoracle.cpp:349-350, tools/syncref.cpp:268)
ZHOBTCODE = [
    normalise("obtid", "varchar2", 10, 1),
    normalise("cityname", "varchar2", 30),
    normalise("provname", "varchar2", 30),
    normalise("lat", "number", 22),
    normalise("lon", "number", 22),
    normalise("height", "number", 22),
    normalise("upttime", "date", 7),
    normalise("keyid", "number", 22),
]

# delimited MIND row: obtid, ddatetime, t, p, u, wd, wf, r, vis, keyid
MIND_CSV_TYPED = [T_NONE, T_NONE, T_F64, T_F64, T_F64, T_F64, T_F64, T_F64, T_F64, T_I64]
