"""Seeded input builders shared by the CPU and GPU parity tests.

Three families:
  * `probe_*`  -- hand-written known-answer inputs for every quirk of the reference path
                  (independent first-open/first-close search, negative length, clip before
                  filter, carry-forward, dropped unterminated tail, '\r' kept, trim bug ...).
  * `mutated_xml` / `soup_xml` -- random records built from a schema with every kind of
                  damage the format allows, and unstructured token soup.
  * `fuzz_csv` -- delimited lines with multi-byte separators, blanks and odd numbers.
"""
from __future__ import annotations

import random
from typing import List, Sequence, Tuple

from dsp_amd.schema import Col

# ---------------------------------------------------------------------------- schemas

SMALL = [
    Col("a", "char", 5),
    Col("b", "number", 22),
    Col("c", "date", 14),
    Col("ab", "char", 0),
    Col("n", "number", 6),
]

# names that collide in their first eight bytes, a name of the maximum length (31) and a
# one-letter name
LONGNAMES = [
    Col("ddatetime", "date", 14),
    Col("ddatetim2", "char", 8),
    Col("ddatetime_ext", "number", 22),
    Col("x" * 31, "char", 40),
    Col("x" * 30, "char", 3),
    Col("t", "number", 22),
]

# duplicate column names and a name that is another one's close tag
ODD = [
    Col("a", "char", 4),
    Col("a", "number", 22),
    Col("/a", "char", 0),
    Col("", "char", 3),
    Col("endl/", "char", 10),
]


# ---------------------------------------------------------------------------- probes

def probe_xml() -> List[Tuple[str, bytes, Sequence[Col], str]]:
    """(name, input, schema, endbz) known-answer style inputs."""
    E = b"<endl/>\n"
    out = []
    add = lambda name, data, cols=SMALL, endbz="<endl/>": out.append((name, data, cols, endbz))
    add("empty", b"")
    add("only_newline", b"\n")
    add("only_terminator", E)
    add("two_empty_records", E + E)
    add("no_terminator", b"<a>1</a><endl/>")
    add("crlf_not_terminator", b"<a>1</a><endl/>\r\n<a>2</a>" + E)
    add("midline_endl", b"<a>1</a><endl/><a>2</a>" + E)
    add("tail_dropped", b"<a>1</a>" + E + b"<a>2</a><endl/>")
    add("data_framing", b"<data>\n<a>1</a><b>2</b>" + E + b"<a>3</a>" + E + b"</data>\n")
    add("interior_newlines", b"<a>x\ny</a>\n\n<b>1\n2</b>" + E)
    add("close_before_open", b"</a>zz<a>hello</a>tail" + E)
    add("close_before_open_num", b"</b>zz<b>12x.5</b>t-9" + E)
    add("prefix_name_no_match", b"<ab>long</ab>" + E)
    add("tag_inside_value", b"<ab>..<a>in</a>..</ab>" + E)
    add("duplicate_first_wins", b"<a>one</a><a>two</a>" + E)
    add("nested", b"<a><a>1</a></a>" + E)
    add("case_sensitive", b"<A>1</A><a>2</a>" + E)
    add("clip_char", b"<a>abcdefghij</a>" + E)
    add("clip_then_filter_date", b"<c>2024-05-19 16:28:00</c>" + E)
    add("number_filter", b"<b> -12.5e3 hPa+</b>" + E)
    add("carry_forward", b"<a>xy9</a><b>10000.5</b>" + E)            # c, ab, n missing
    add("carry_first_missing", b"<b>7</b>" + E)
    add("carry_restrictive_then_char", b"<b>1.5</b><ab>zz</ab>" + E)  # c missing -> "15"
    add("open_only", b"<a>1<b>2</b>" + E)
    add("close_only", b"1</a><b>2</b>" + E)
    add("nul_in_value", b"<a>x\x00y</a><b>1\x002</b><ab>q\x00r5</ab>" + E)
    add("nul_then_carry", b"<ab>ab\x00cd12</ab>" + E, [Col("ab", "char", 0), Col("z", "number", 22)])
    add("utf8_clip", "<a>观测站点</a>".encode() + E)
    add("high_bytes_digits", b"<b>\xff1\x802\xb9</b>" + E)
    add("empty_value", b"<a></a><b></b>" + E)
    add("overlap_tags", b"<a><b>1</a>2</b>" + E)
    add("endl_in_value", b"<a><endl/></a>" + E)
    add("lt_gt_noise", b"<<a>>1<</a>><b><>2</b>" + E)
    add("first_record_prefix", b"<?xml version='1.0'?>\n<!--c-->\n<x>\n  <a>1</a>" + E + b"</x>\n")
    add("lines_mode", b"l1\n\nl3\r\nlast", SMALL, "")
    add("custom_endbz", b"<a>1</a>;;\n<a>2</a>;\n;;\n", SMALL, ";;")
    add("long_endbz", b"<a>1</a>0123456789abcdef\n<a>2</a>123456789abcdef\n", SMALL, "0123456789abcdef")
    add("longnames", b"<ddatetime>2024-01-02</ddatetime><ddatetim2>abcdefghijk</ddatetim2>"
        b"<ddatetime_ext>-1.5</ddatetime_ext><" + b"x" * 31 + b">full</" + b"x" * 31 + b"><"
        + b"x" * 30 + b">short</" + b"x" * 30 + b"><t>3</t>" + E, LONGNAMES)
    add("longnames_nearmiss", b"<ddatetimX>1</ddatetimX><ddatetime_exu>2</ddatetime_exu><"
        + b"x" * 32 + b">3</" + b"x" * 32 + b">" + E, LONGNAMES)
    add("odd_schema", b"<a>12ab</a></a>s<>e</></endl/>f<endl/>g" + E, ODD)
    add("odd_schema2", b"</a>q</a><a>7</a><//a>" + E, ODD)
    # the reference's own sample lines (public/_public.h:131-132)
    add("public_h_sample",
        b"<filename>/tmp/_public.h</filename><mtime>2020-01-01 12:20:35</mtime><size>18348</size>\n"
        b"<filename>/tmp/_public.cpp</filename><mtime>2020-01-01 10:10:15</mtime><size>50945</size>\n",
        [Col("filename", "char", 300), Col("mtime", "date", 14), Col("size", "number", 22)], "")
    return out


INI_XMLTODB = (
    # byte-for-byte the structure of the reference's idc/ini/xmltodb.xml (2 records, the
    # second without <execsql>); the comment text is replaced by ASCII
    b"<?xml version='1.0' encoding='utf-8'?>\n\n<!--ingest parameters-->\n<xmltodb>\n"
    b"  <filename>ZHOBTCODE_*.XML</filename><tname>T_ZHOBTCODE1</tname><uptbz>1</uptbz>"
    b"<execsql>delete from T_ZHOBTCODE1</execsql><endl/>\n"
    b"  <filename>ZHOBTMIND_*.XML</filename><tname>T_ZHOBTMIND1</tname><uptbz>1</uptbz><endl/>\n"
    b"</xmltodb>"
)
INI_COLS = [Col("filename", "char", 0), Col("tname", "char", 0), Col("uptbz", "char", 0),
            Col("execsql", "char", 0)]


def probe_convert() -> List[bytes]:
    return [
        b"", b"0", b"12.7", b"5-3", b"-", b"+", b"+-5", b"-5", b"--5", b"2147483647", b"2147483648",
        b"-2147483648", b"-2147483649", b"4000000000", b"9223372036854775807",
        b"9223372036854775808", b"-9223372036854775808", b"-9223372036854775809",
        b"18446744073709551615", b"18446744073709551616", b"1.2.3", b".", b".5", b"-.5", b"5.",
        b"1e5", b"0.1", b"0.3", b"123456789012345678", b"1234567890123456789012",
        b"9007199254740993", b"9007199254740992.5", b"9007199254740993.0000000001",
        b"0.1000000000000000055511151231257827", b"179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365510704342711559699508093042880177904174497791",
        b"1" + b"0" * 400, b"0." + b"0" * 400 + b"1", b"0." + b"0" * 320 + b"1",
        b"2.2250738585072011e-308", b"0.000000000000000000000000000000000000011754943508222875",
        b"340282350000000000000000000000000000000", b"340282356779733661637539395458142568448",
        b"16777217", b"16777216.5", b"0.1000000014901161", b"1.00000017881393421514957253748434595763683319091796875",
        b"TRUE", b"true", b"True", b" true", b"TRUEx", b"false", b"00012", b"-0", b"-0.0", b"+.0",
        b" -12.5e3 hPa+", b"1,234.5", b"12abc34", b"\xff9\x008", b"99999999999999999999.99",
        b"4.35", b"0.000001", b"123456.789e3", b"8.5", b"1013.2", b"10000.5",
    ]


def probe_trim() -> List[bytes]:
    return [b"", b" ", b"   ", b"a", b" a", b"a ", b" a ", b"  ab  ", b"a,", b",a,", b" x ", b" b ",
            b"x  y", b"  x  y  ", b"\t a \t"]


# ---------------------------------------------------------------------------- random XML

_VALUES = [b"", b"0", b"1", b"-1", b"12.7", b"1013.2", b"-40.0", b"+5", b"1e5", b"1.2.3", b"..", b"-",
           b"2024-05-19 16:28:00", b"20240519162800", b"abc", b"hello world", b" 42 ", b"4.35",
           b"99999999999999999999999999", b"123456789012345678901234567890.5", b"x\x00y", b"\xe8\xa7\x82",
           b"<", b">", b"</", b"a<b", b"\r", b"\n", b"0.000001", b"9007199254740993"]


def _rand_value(rng: random.Random) -> bytes:
    k = rng.random()
    if k < 0.45:
        return rng.choice(_VALUES)
    if k < 0.75:
        n = rng.randrange(0, 30)
        return bytes(rng.choice(b"0123456789+-. eE") for _ in range(n))
    n = rng.randrange(0, 60)
    return bytes(rng.choice(b"abcXYZ019 .-+<>/\n\r\x00\xe8\xa7\x82") for _ in range(n))


def mutated_xml(seed: int, nrec: int, cols: Sequence[Col], endbz: str = "<endl/>",
                damage: float = 0.3) -> bytes:
    """Records built from the schema with random damage (missing/duplicated/reordered/nested
    tags, near-miss names, long values, odd terminators)."""
    rng = random.Random(seed)
    names = [c.colname.encode() for c in cols]
    out = bytearray()
    if rng.random() < 0.5:
        out += b"<data>\n"
    term = endbz.encode() + b"\n"
    for _ in range(nrec):
        order = list(range(len(cols)))
        if rng.random() < damage:
            rng.shuffle(order)
        for i in order:
            nm = names[i]
            r = rng.random()
            if r < damage * 0.3:
                continue                                   # missing tag
            v = _rand_value(rng)
            if r < damage * 0.4:
                out += b"</" + nm + b">" + v + b"<" + nm + b">" + _rand_value(rng) + b"</" + nm + b">"
            elif r < damage * 0.5:
                out += b"<" + nm + b">" + v                  # open only
            elif r < damage * 0.6:
                out += v + b"</" + nm + b">"                 # close only
            elif r < damage * 0.7:
                out += b"<" + nm + b"x>" + v + b"</" + nm + b"x>"   # near miss
            elif r < damage * 0.8:
                out += b"<" + nm + b"><" + nm + b">" + v + b"</" + nm + b"></" + nm + b">"
            elif r < damage * 0.9:
                out += b"<" + nm.upper() + b">" + v + b"</" + nm.upper() + b">"
            else:
                out += b"<" + nm + b">" + v + b"</" + nm + b">"
            if rng.random() < damage * 0.1:
                out += b"\n"
        r = rng.random()
        if r < damage * 0.1:
            out += endbz.encode() + b"\r\n"
        elif r < damage * 0.2:
            out += endbz.encode() + b" \n"
        elif r < damage * 0.25:
            out += endbz.encode()
        else:
            out += term
    r = rng.random()
    if r < 0.4:
        out += b"</data>\n"
    elif r < 0.6:
        out += b"<" + names[0] + b">tail</" + names[0] + b">" + endbz.encode()
    return bytes(out)


def soup_xml(seed: int, ntok: int, cols: Sequence[Col], endbz: str = "<endl/>") -> bytes:
    """Unstructured token soup over the schema's tags and delimiters."""
    rng = random.Random(seed)
    toks = [b"<", b">", b"/", b"\n", b"\r", b"\x00", b" ", b"<endl/>", b"<endl/>\n", endbz.encode() + b"\n",
            b"0", b"1", b"9", b".", b"-", b"+", b"x", b"<data>\n", b"</data>\n", b"\xe8\xa7\x82"]
    for c in cols:
        nm = c.colname.encode()
        toks += [b"<" + nm + b">", b"</" + nm + b">", b"<" + nm, nm + b">", b"<" + nm + b">" + b"5" * 3 + b"</" + nm + b">"]
    return b"".join(rng.choice(toks) for _ in range(ntok))


# ---------------------------------------------------------------------------- random CSV

def fuzz_csv(seed: int, nlines: int, sep: str, nfields: int) -> bytes:
    rng = random.Random(seed)
    s = sep.encode()
    vals = [b"", b" ", b"  x  ", b"12.7", b"-5", b"+3", b"1e5", b"TRUE", b"true", b" true", b"58015",
            b"20240519162800", b"1013.2", b"0.0", b"10000.5", b"1234567", b"a,", b"a b", b" b ",
            b"99999999999999999999", b"4000000000", b"\r", b"x\x00y", b"1.2.3", b"~", b"!", b"~!"]
    out = bytearray()
    for _ in range(nlines):
        k = rng.choice([0, 1, nfields - 1, nfields, nfields, nfields, nfields + 1, nfields + 3])
        fields = []
        for _ in range(k):
            if rng.random() < 0.6:
                fields.append(rng.choice(vals))
            else:
                fields.append(bytes(rng.choice(b"0123456789.-+ abT") for _ in range(rng.randrange(0, 12))))
        line = s.join(fields)
        if rng.random() < 0.1:
            line += s
        if rng.random() < 0.05:
            line += s[:1]
        out += line + b"\n"
    if rng.random() < 0.5:
        out += b"unterminated" + s + b"tail"
    return bytes(out)


def probe_csv() -> List[Tuple[str, bytes, str, bool]]:
    return [
        ("messi", b"messi,10,striker,30,1.72,68.5,Barcelona\n", ",", False),
        ("multi_sep", b"x~!~y~!~!~z\n", "~!~", False),
        ("empty_line", b"\n\n", ",", False),
        ("trailing_sep", b"a,b,\n", ",", False),
        ("delspace_bug", b" x , a ,b , c\n", ",", True),
        ("only_seps", b",,,\n", ",", False),
        ("crlf", b"1,2\r\n3,4\r\n", ",", False),
        ("mind_row", b"58015,20240519162800,12.7,1013.2,65,180,3.4,0.0,10000.5,1234567\n", ",", False),
        ("unterminated_tail", b"1,2\n3,4", ",", False),
        ("space_sep", b"a  b c\n", " ", False),
    ]


# ---- writer side ---------------------------------------------------------------------------
# field lists of the extractor's own usage text (tools/dminingoracle.cpp:350, :357)
CODE_FIELDS = (["obtid", "cityname", "provname", "lat", "lon", "height"], [5, 30, 30, 10, 10, 10])
MIND_FIELDS = (["obtid", "ddatetime", "t", "p", "u", "wd", "wf", "r", "vis", "keyid"],
               [5, 19, 8, 8, 8, 8, 8, 8, 8, 15])


def columns_of(rows_of_cells: Sequence[Sequence[bytes]], nfields: int):
    """rows of cells -> per column (u32 offsets, byte arena), the layout of a parsed batch"""
    import numpy as np
    off, data = [], []
    for c in range(nfields):
        cells = [bytes(r[c]) for r in rows_of_cells]
        o = np.zeros(len(cells) + 1, dtype=np.uint32)
        if cells:
            o[1:] = np.cumsum([len(x) for x in cells])
        off.append(o)
        data.append(np.frombuffer(b"".join(cells), dtype=np.uint8).copy())
    return off, data


def random_rows(seed: int, nrows: int, nfields: int, maxlen: int = 12, nul: float = 0.0, big: float = 0.0):
    rng = random.Random(seed)
    alphabet = b"0123456789.-abcXYZ <>/%\xe4\xb8\xad"
    rows = []
    for _ in range(nrows):
        row = []
        for _ in range(nfields):
            n = rng.randrange(0, maxlen + 1)
            if big and rng.random() < big:
                n = rng.randrange(maxlen, 40 * maxlen)
            cell = bytearray(rng.choice(alphabet) for _ in range(n))
            if nul and n and rng.random() < nul:
                cell[rng.randrange(n)] = 0
            row.append(bytes(cell))
        rows.append(row)
    return rows


def format_cases():
    """(name, fieldnames, fieldlen or None, rows of cells, maxcount)"""
    out = []
    mind_rows = [[b"58015", b"2024-05-19 16:28:00", b"12.7", b"1013.2", b"65", b"180", b"3.4", b"0.0",
                  b"10000.5", str(1000 + i).encode()] for i in range(7)]
    mind_rows[3][2] = b"-40.0"
    mind_rows[5][8] = b""
    for mc in (0, 1, 3, 7, 1000):
        out.append((f"mind/maxcount={mc}", MIND_FIELDS[0], MIND_FIELDS[1], mind_rows, mc))
    code_rows = [[b"58015", "砀山".encode(), "安徽".encode(), b"34.27", b"116.2", b"44.2"],
                 [b"58102", "亳州".encode(), "安徽".encode(), b"33.47", b"115.44", b"39.1"],
                 [b"58118", "蒙城".encode(), "安徽".encode(), b"33.16", b"116.31", b"27"]]
    out.append(("code/maxcount=2", CODE_FIELDS[0], CODE_FIELDS[1], code_rows, 2))
    cut_rows = [[b"abcdefgh", b"12\x0034", b""], [b"\x00", b"x" * 40, b"tail\x00"], [b"ab", b"", b"\x00\x00"],
                [b"abcd", b"abcde", b"abcdef"]]
    out.append(("cuts/fieldlen", ["a", "bb", "ccc"], [4, 10, 5], cut_rows, 0))
    out.append(("cuts/nul_only", ["a", "bb", "ccc"], None, cut_rows, 3))
    out.append(("cuts/fieldlen_some", ["a", "bb", "ccc"], [0, 3, 0], cut_rows, 2))
    out.append(("one_field_one_row", ["v"], None, [[b"1"]], 0))
    out.append(("empty_cells", ["a", "b"], None, [[b"", b""], [b"", b""]], 1))
    out.append(("zero_rows/maxcount=0", ["a", "b"], None, [], 0))
    out.append(("zero_rows/maxcount=5", ["a", "b"], None, [], 5))
    out.append(("odd_names", ["a b", "x<y", "%s%d", "中文", "e" * 40], None,
                [[b"%s", b"%n%n", b"<a b>", b"</x<y>", b"<endl/>"], [b"1", b"2", b"3", b"4", b"5"]], 0))
    for seed in range(4):
        nf = 1 + 3 * seed
        fl = None if seed % 2 == 0 else [(3 * i) % 7 for i in range(nf)]
        out.append((f"random/{seed}", [f"f{i}" * (1 + i % 3) for i in range(nf)], fl,
                    random_rows(seed, 23 + seed, nf, nul=0.1 * (seed % 3), big=0.05 * seed), [0, 5, 23, 4][seed]))
    return out


# ---- routing, matchstr, SQL text -------------------------------------------------------------

def match_cases():
    """(file name, rules) pairs for matchstr: the samples of the reference's parameter file and
    usage text, and the corners of its piece-by-piece search."""
    names = ["ZHOBTCODE_20240519.XML", "ZHOBTMIND_20240519162800_togxpt_1.xml", "zhobtmind_1.XML", "a.xml", "A",
             "", "abcabc", "SURF_ZH_2024.txt.gz", "x*y", "ab", "aXbXc", "abc.XML.bak"]
    rules = ["ZHOBTCODE_*.XML", "ZHOBTMIND_*.XML", "*.XML", "*", "", ",", "*,", ",*", "a*", "*a", "a", "A*A", "a*a",
             "abc*abc", "abc*bc", "*abc*", "ab*ab*ab", "*.xml,*.gz", "zhobtcode_*,zhobtmind_*", "SURF*,,*.XML",
             "*b*", "a**c", "**", "a*b*c", "a*c*b", "abcabcabc", "x\\*y", ".XML*", "*.XML*", "ZHOBTMIND*_1.XML",
             "ab,", " ab", "ab*"]
    return [(n, r) for n in names for r in rules]


def ini_cases():
    """(name, bytes) parameter files for loadxmltotable.  Values stay within the sizes of
    st_xmltotable (the reference writes beyond its arrays otherwise)."""
    long_ok = b"F" * 127
    return [
        ("ini_xmltodb", INI_XMLTODB + b"\n"),
        ("ini_no_final_newline", INI_XMLTODB),
        ("empty", b""),
        ("only_marks", b"<endl/>\n<endl/>\n\n<endl/>\n"),
        ("missing_and_odd",
         b"<filename>A_*.XML,B_*.XML</filename><tname>T_A</tname><uptbz>2</uptbz><endl/>\n"
         b"<tname>T_NOFILE</tname><uptbz>x1y</uptbz><execsql></execsql><endl/>\n"
         b"<filename>*.csv</filename><tname></tname><uptbz>-</uptbz><endl/>\n"
         b"<filename>c</filename><uptbz>99999999999</uptbz><tname>T_BIG</tname><endl/>\n"
         b"<filename>d</filename><uptbz> +1 </uptbz><tname>T_SIGN</tname><execsql>delete from t where a='<b>'</execsql><endl/>\n"
         b"</filename>zz<filename>close_first</filename><tname>T_CF</tname><endl/>\n"
         b"<filename>dup1</filename><filename>dup2</filename><tname>T_D</tname><uptbz>1.5</uptbz><endl/>\n"
         b"<filename>two\nlines</filename><tname>T_NL</tname><uptbz>1</uptbz>\n<endl/>\n"
         b"<filename>crlf</filename><tname>T_CR</tname><endl/>\r\n"
         b"<filename>tail</filename><tname>T_TAIL</tname><endl/>"),
        ("nul_inside", b"<filename>ab\x00cd</filename><tname>\x00T</tname><uptbz>1</uptbz><execsql>x\x00y</execsql><endl/>\n"),
        ("max_sizes", b"<filename>" + long_ok + b"</filename><tname>" + b"T" * 31 + b"</tname><uptbz>-2147483648</uptbz>"
         b"<execsql>" + b"s" * 255 + b"</execsql><endl/>\n"
         b"<filename>f</filename><tname>t</tname><uptbz>2147483647</uptbz><endl/>\n"
         b"<filename>g</filename><tname>t</tname><uptbz>2147483648</uptbz><endl/>\n"),
    ]


def sql_cases():
    """(name, (filename, tname, uptbz, execsql), columns) for crtsql and the bind order."""
    from dsp_amd import schema
    mind, code = schema.ZHOBTMIND, schema.ZHOBTCODE
    rev = [Col(c.colname, c.datatype, c.collen, {1: 2, 2: 1}.get(c.pkseq, 0)) for c in mind]
    out = []
    for up in (1, 2, 0):
        out.append((f"zhobtmind/uptbz={up}", ("ZHOBTMIND_*.XML", "T_ZHOBTMIND1", up, ""), mind))
        out.append((f"zhobtcode/uptbz={up}", ("ZHOBTCODE_*.XML", "T_ZHOBTCODE1", up, "delete from T_ZHOBTCODE1"), code))
    out += [
        ("key_order_reversed", ("x", "T_ZHOBTMIND1", 1, ""), rev),
        ("one_byte_name", ("x", "T_X", 1, ""), [Col("t", "number", 22)]),
        ("one_byte_key", ("x", "T_X", 1, ""), [Col("k", "char", 5, 1)]),
        ("short_tname", ("x", "T", 1, ""), [Col("a", "char", 5, 1), Col("keyid", "number", 22)]),
        ("empty_tname", ("x", "", 1, ""), [Col("keyid", "number", 22), Col("ab", "date", 14)]),
        ("only_upttime", ("x", "T_U", 1, ""), [Col("upttime", "date", 14)]),
        ("no_columns", ("x", "T_N", 1, ""), []),
        ("date_key_no_set", ("x", "T_DK", 1, ""), [Col("d1", "date", 14, 1), Col("d2", "date", 14, 2)]),
        ("keyid_first", ("x", "T_KF", 1, ""), [Col("keyid", "number", 22), Col("v", "number", 22), Col("d", "date", 14),
                                                 Col("upttime", "date", 14), Col("id", "char", 10, 1)]),
        ("keyid_is_key", ("x", "T_KK", 1, ""), [Col("keyid", "number", 22, 1), Col("v", "char", 3)]),
        ("upttime_is_key", ("x", "T_UK", 1, ""), [Col("upttime", "date", 14, 1), Col("v", "char", 3)]),
        ("many", ("x", "T_MANY", 1, ""), [Col(f"c{i}", ("char", "date", "number")[i % 3], 10, 3 - i if i < 3 else 0)
                                           for i in range(24)]),
    ]
    return out
