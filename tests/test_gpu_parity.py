"""GPU parity: the HIP path, called through the C ABI, against the CPU checker and the golden
vectors.  Bit-exact for every output (cell bytes, offsets, record offsets, masks, typed values)."""
import ctypes
import os
import random

import numpy as np
import pytest

import cases
import golden_io
from dsp_amd import capi, gen, schema
from dsp_amd.schema import Col
from util import assert_batches_equal

pytestmark = pytest.mark.gpu

ALL_T = list(range(1, 8))


def _both_forms(p, data):
    """Parses twice with one parser: the second parse reuses the first one's device workspace
    and must give the same bytes."""
    a = p.parse_host(data).to_host()
    b = p.parse_host(data).to_host()
    assert_batches_equal(b, a, "second parse with the same parser")
    return b


def gpu_xml(data, cols, typed=None, endbz="<endl/>", carry=True):
    p = capi.Parser.xml(cols, typed=typed, endbz=endbz, carry=carry)
    try:
        return _both_forms(p, data)
    finally:
        p.close()


def gpu_csv(data, sep, nfields, ilen=None, typed=None, bdelspace=False):
    p = capi.Parser.csv(sep, nfields, ilen=ilen, typed=typed, bdelspace=bdelspace)
    try:
        return _both_forms(p, data)
    finally:
        p.close()


def test_library_loads_on_gpu():
    assert capi.lib().mdc_device_count() >= 1


@pytest.mark.parametrize("case", golden_io.xml_cases(), ids=lambda c: c["name"])
def test_golden_xml(case):
    print("case", case["name"], flush=True)
    got = gpu_xml(case["data"], case["cols"], typed=case["typed"], endbz=case["endbz"], carry=case["carry"])
    assert_batches_equal(got, case["expect"], case["name"])


@pytest.mark.parametrize("case", golden_io.csv_cases(), ids=lambda c: c["name"])
def test_golden_csv(case):
    print("case", case["name"], flush=True)
    got = gpu_csv(case["data"], case["sep"], case["nfields"], ilen=case["ilen"], typed=case["typed"],
                  bdelspace=case["bdelspace"])
    assert_batches_equal(got, case["expect"], case["name"])


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("schema_name", ["SMALL", "LONGNAMES", "ODD"])
def test_mutated_xml(port, seed, schema_name):
    cols = getattr(cases, schema_name)
    data = cases.mutated_xml(100 + seed, 700, cols, damage=0.15 + 0.07 * seed)
    typed = [ALL_T[(seed + i) % len(ALL_T)] for i in range(len(cols))]
    for carry in (True, False):
        print("mutated", schema_name, seed, carry, len(data), flush=True)
        got = gpu_xml(data, cols, typed=typed, carry=carry)
        exp = port.parse_xml(data, cols, typed=typed, carry=carry)
        assert_batches_equal(got, exp, f"mutated {schema_name} {seed} carry={carry}")


@pytest.mark.parametrize("seed", range(4))
def test_soup_xml(port, seed):
    for cols in (cases.SMALL, cases.ODD):
        data = cases.soup_xml(200 + seed, 30000, cols)
        print("soup", seed, len(data), flush=True)
        got = gpu_xml(data, cols)
        exp = port.parse_xml(data, cols)
        assert_batches_equal(got, exp, f"soup {seed}")


@pytest.mark.parametrize("rows", [1, 95, 96, 1000, 20000])
def test_mind_files(port, rows):
    data, n = gen.mind_xml(rows=rows)
    print("mind", rows, len(data), flush=True)
    got = gpu_xml(data, schema.ZHOBTMIND)
    exp = port.parse_xml(data, schema.ZHOBTMIND)
    assert got.rows == n
    assert_batches_equal(got, exp, f"mind rows={rows}")


def test_code_file(port):
    data, n = gen.code_xml()
    got = gpu_xml(data, schema.ZHOBTCODE)
    exp = port.parse_xml(data, schema.ZHOBTCODE)
    assert got.rows == n == 839
    assert_batches_equal(got, exp, "code")


def test_short_records_many_batches(port):
    # 8-byte records: 2048 records per tile, many batches per tile
    data = b"<endl/>\n" * 5000 + b"<a>1</a><endl/>\n" * 3000
    got = gpu_xml(data, cases.SMALL)
    exp = port.parse_xml(data, cases.SMALL)
    assert_batches_equal(got, exp, "short records")


@pytest.mark.parametrize("reclen", [63, 64, 65, 126, 127, 128, 129, 130, 143, 144, 145, 146, 147, 148, 170, 171, 172, 255, 256, 257])
def test_records_per_tile_around_the_batch_sizes(port, reclen):
    # records of one length each: 16 KiB tiles then hold 64, 112, 113, 127 to 130 ... records, the counts
    # at which a tile needs one batch more (batches of 112 records, of 128 for inputs of short records)
    fixed = len(b"<a>12345</a><ab></ab><b>-12.25</b><endl/>\n")
    recs = [b"<a>%05d</a><ab>" % (i % 100000) + bytes([97 + i % 26]) * (reclen - fixed) + b"</ab><b>-12.25</b><endl/>\n"
            for i in range(2600)]
    assert len(recs[0]) == reclen
    data = b"".join(recs)
    p = capi.Parser.xml(cases.SMALL, typed=[schema.T_I64, schema.T_NONE, schema.T_F64])
    exp = port.parse_xml(data, cases.SMALL, typed=[schema.T_I64, schema.T_NONE, schema.T_F64])
    for _ in range(2):  # (the second parse sizes the cell map by the first one's records per tile)
        got = p.parse_host(data).to_host()
        assert_batches_equal(got, exp, f"records of {reclen} bytes")
    p.close()
    # the same lines as delimited input
    lines = b"".join(b"%05d,%s,-12.25\n" % (i % 100000, bytes([97 + i % 26]) * (reclen - 14)) for i in range(2600))
    q = capi.Parser.csv(",", 3, typed=[schema.T_I64, schema.T_NONE, schema.T_F64])
    expc = port.parse_csv(lines, ",", 3, typed=[schema.T_I64, schema.T_NONE, schema.T_F64])
    for _ in range(2):
        got = q.parse_host(lines).to_host()
        assert_batches_equal(got, expc, f"lines of {reclen} bytes")
    q.close()


@pytest.mark.parametrize("endbz", ["<endl/>", "##", ""])
def test_terminator_at_every_place_around_a_tile_boundary(port, endbz):
    # the newline that ends a record, and the end mark in front of it, moved byte by byte across the
    # boundary between the first and the second tile
    e = endbz.encode()
    tail = b"".join(b"<a>%d</a><ab>v%d</ab><b>%d.25</b>" % (i, i, i) + e + b"\n" for i in range(300))
    for at in range(16384 - 12, 16384 + 12):  # position of the newline
        head = b"".join(b"<a>%d</a><ab>h</ab><b>1</b>" % i + e + b"\n" for i in range(20))
        body = at - len(head) - len(b"<a>7</a><ab></ab><b>2.5</b>") - len(e)
        rec = b"<a>7</a><ab>" + b"w" * body + b"</ab><b>2.5</b>" + e + b"\n"
        data = head + rec + tail
        assert data[at] == 10
        got = gpu_xml(data, cases.SMALL, endbz=endbz)
        assert_batches_equal(got, port.parse_xml(data, cases.SMALL, endbz=endbz), f"endbz {endbz!r}, newline at {at}")


@pytest.mark.parametrize("around", [512, 1024, 1536, 2048, 8192, 16320, 16384])
def test_record_start_around_every_look_behind_threshold(port, around):
    # the first record of the third tile starts D bytes in front of the tile, D moved byte by byte across
    # the first look-behind window, its extensions, the capacities of the three launch configurations and
    # a whole tile (beyond which the record goes to the long-record kernel)
    filler = b"".join(b"<a>%d</a><ab>f</ab><b>1</b><endl/>\n" % i for i in range(30))
    tail = b"".join(b"<a>%d</a><ab>v%d</ab><b>%d.25</b><endl/>\n" % (i, i, i) for i in range(300))
    p = capi.Parser.xml(cases.SMALL)
    for D in range(around - 9, around + 10):
        start = 2 * 16384 - D                                   # where the record starts
        pad = start - len(filler) - len(b"<a>1</a><ab></ab><endl/>\n")
        a = b"<a>1</a><ab>" + b"p" * pad + b"</ab><endl/>\n"     # ends right in front of `start`
        b = b"<a>22</a><ab>" + b"q" * D + b"</ab><b>7.5</b><endl/>\n"   # D + 40 bytes: ends in the third tile
        data = filler + a + b + tail
        assert len(filler + a) == start
        got = p.parse_host(data).to_host()
        assert_batches_equal(got, port.parse_xml(data, cases.SMALL), f"record starts {D} bytes in front of its tile")
    p.close()
    # the same geometry with delimited lines (separators of one and of three bytes)
    for sep in (",", "~!~"):
        sp = sep.encode()
        fill = b"".join(sp.join([b"%d" % i, b"f", b"1"]) + b"\n" for i in range(30))
        rest = b"".join(sp.join([b"%d" % i, b"v%d" % i, b"%d.25" % i]) + b"\n" for i in range(300))
        q = capi.Parser.csv(sep, 3, typed=[schema.T_I64, schema.T_NONE, schema.T_F64])
        for D in range(around - 9, around + 10):
            start = 2 * 16384 - D
            a = b"1" + sp + b"p" * (start - len(fill) - 2 - len(sp)) + b"\n"
            b = b"22" + sp + b"q" * D + sp + b"7.5\n"
            data = fill + a + b + rest
            assert len(fill + a) == start
            got = q.parse_host(data).to_host()
            exp = port.parse_csv(data, sep, 3, typed=[schema.T_I64, schema.T_NONE, schema.T_F64])
            assert_batches_equal(got, exp, f"line starts {D} bytes in front of its tile, separator {sep!r}")
        q.close()


@pytest.mark.parametrize("reclen", [100, 1000, 5000, 8000])
def test_records_straddling_tiles(port, reclen):
    # records whose start lies far in front of the tile that holds their terminator
    rng = np.random.default_rng(reclen)
    recs = []
    for i in range(40):
        pad = bytes(rng.integers(97, 123, size=reclen - 40, dtype=np.uint8))
        recs.append(b"<a>%d</a><ab>" % i + pad + b"</ab><b>%d.5</b><endl/>\n" % i)
    data = b"".join(recs)
    got = gpu_xml(data, cases.SMALL)
    exp = port.parse_xml(data, cases.SMALL)
    assert_batches_equal(got, exp, f"straddle {reclen}")


def _long_xml_record(rng, i, total, variant):
    """One record of about `total` bytes for the SMALL schema (a text, ab text, b number)."""
    fill = bytes(rng.integers(97, 123, size=total, dtype=np.uint8))
    if variant == 0:    # a long text cell
        return b"<a>%d</a><ab>" % i + fill + b"</ab><b>%d.5</b><endl/>\n" % i
    if variant == 1:    # padding between the fields, tag-like text and newlines inside
        return b"<a>k%d</a>\n" % i + fill[:total // 2] + b"<x><a></b>\n" + fill[total // 2:] + b"<b>-%d.25</b><ab>z</ab><endl/>\n" % i
    if variant == 2:    # close tag in front of the open tag: the cell is the rest of the record; a NUL cuts it
        return b"</ab><ab>rest" + fill[:total // 3] + b"\0" + fill[total // 3:] + b"<a>7</a><endl/>\n"
    if variant == 3:    # a long number: digits with other bytes between them, no <ab> (the value carries over)
        digits = bytes(rng.integers(48, 58, size=total, dtype=np.uint8))
        return b"<a>n%d</a><b>1" % i + digits + b"x.5</b><endl/>\n"
    return b"<b>%d</b>" % i + fill + b"<endl/>\n"  # no <a>, no <ab>


@pytest.mark.parametrize("total", [17000, 40000, 1 << 20])
def test_records_longer_than_the_look_behind(port, total):
    # cifile::readline puts no bound on a record: such records go through the long-record kernel
    rng = np.random.default_rng(total)
    short = [b"<a>%d</a><ab>s</ab><b>%d.5</b><endl/>\n" % (i, i) for i in range(400)]
    parts = [b"<data>\n"]
    nlong = 0
    for v in range(5):
        parts += short[80 * v:80 * v + 80 - 7 * v]
        parts.append(_long_xml_record(rng, v, total, v))
        nlong += 1
    parts += [_long_xml_record(rng, 9, total, 0), _long_xml_record(rng, 10, total, 1)]  # two in a row
    nlong += 2
    parts += short[:50] + [b"</data>\n"]
    data = b"".join(parts)
    for typed in (None, [schema.T_NONE, schema.T_F64, schema.T_I64, schema.T_NONE, schema.T_I32],
                  [schema.T_I64, schema.T_F32, schema.T_U64, schema.T_BOOL, schema.T_F64]):
        p = capi.Parser.xml(cases.SMALL, typed=typed)
        got = p.parse_host(data).to_host()
        # (a record of 17,000 bytes fits the largest look-behind when its terminator lies deep enough in its tile)
        assert p.timing().long_records == nlong if total >= 40000 else p.timing().long_records >= 1
        exp = port.parse_xml(data, cases.SMALL, typed=typed)
        assert_batches_equal(got, exp, f"long records of {total} bytes, typed {typed}")
        # the same parser, a second time (workspace reuse), and without the carry-forward
        assert_batches_equal(p.parse_host(data).to_host(), exp, "second parse")
        p.close()
    got = gpu_xml(data, cases.SMALL, carry=False)
    assert_batches_equal(got, port.parse_xml(data, cases.SMALL, carry=False), "long records, independent columns")


def test_long_record_ending_at_every_byte_of_a_lane(port):
    # the tile that ends a long record goes on without that record's terminator, which may be any
    # byte of the 64 a lane owns (the last one included) and any lane of the tile (the first and last included)
    rng = np.random.default_rng(11)
    body = bytes(rng.integers(97, 123, size=70000, dtype=np.uint8))
    after = b"<a>2</a><b>3.5</b><endl/>\n" * 3
    p = capi.Parser.xml(cases.SMALL)
    for off in list(range(60, 132)) + [0, 1, 7, 16383, 16382, 16320]:
        n = 3 * 16384 + off  # position of the long record's terminating newline
        rec = b"<a>1</a><ab>" + body[: n - 12 - 12] + b"</ab><endl/>"
        assert len(rec) == n
        data = rec + b"\n" + after
        got = p.parse_host(data).to_host()
        assert p.timing().long_records == 1
        assert_batches_equal(got, port.parse_xml(data, cases.SMALL), f"terminator at tile offset {off}")
    p.close()


def test_a_long_first_record_and_a_long_unterminated_tail(port):
    rng = np.random.default_rng(5)
    first = _long_xml_record(rng, 1, 70000, 0)
    tail = b"<a>never ends</a>" + bytes(rng.integers(97, 123, size=50000, dtype=np.uint8))
    data = first + b"<a>2</a><endl/>\n" * 3 + tail
    got = gpu_xml(data, cases.SMALL)
    assert_batches_equal(got, port.parse_xml(data, cases.SMALL), "long first record, long tail without terminator")
    assert got.rows == 4


def test_a_file_of_records_just_longer_than_every_look_behind(port):
    # every tile meets a record that does not fit: the parser first tries its larger look-behind
    # capacities and then hands what is still too long to the long-record kernel
    rng = np.random.default_rng(6)
    for size, exact in ((16400, False), (33000, True)):
        recs = [_long_xml_record(rng, i, size + 37 * (i % 5), i % 5) for i in range(60)]
        data = b"".join(recs)
        p = capi.Parser.xml(cases.SMALL)
        got = p.parse_host(data).to_host()
        # (a record of 16,400 bytes still fits when its terminator lies deep enough in its tile)
        assert p.timing().long_records == 60 if exact else p.timing().long_records >= 1
        assert_batches_equal(got, port.parse_xml(data, cases.SMALL), f"many long records of {size} bytes")
        p.close()


@pytest.mark.parametrize("sep,bdel", [(",", False), ("~!~", True), (" ", True)])
def test_delimited_lines_longer_than_the_look_behind(port, sep, bdel):
    rng = np.random.default_rng(len(sep) + bdel)
    s = sep.encode()
    lines = []
    for i in range(300):
        lines.append(s.join([b"%d" % i, b" x%d " % i, b"%d.5" % i, b"tail"]) + b"\n")
        if i % 60 == 7:
            big = bytes(rng.integers(97, 123, size=30000 + 1000 * i, dtype=np.uint8))
            fields = [b"  " + big[:10000] + b"  ", b"%d" % i, b"-12.5", big[10000:], b"more", b"and more"][: 3 + i % 4]
            lines.append(s.join(fields) + b"\n")
        if i % 100 == 50:  # a long line without any separator, and one of separators only
            lines.append(bytes(rng.integers(97, 123, size=20000, dtype=np.uint8)) + b"\n")
            lines.append(s * 9000 + b"\n")
    data = b"".join(lines)
    typed = [schema.T_I64, schema.T_NONE, schema.T_F64, schema.T_NONE, schema.T_NONE]
    p = capi.Parser.csv(sep, 5, ilen=[0, 0, 0, 25000, 3], typed=typed, bdelspace=bdel)
    got = p.parse_host(data).to_host()
    assert p.timing().long_records >= 5
    assert_batches_equal(got, port.parse_csv(data, sep, 5, ilen=[0, 0, 0, 25000, 3], typed=typed, bdelspace=bdel), f"long lines, sep {sep!r}")
    p.close()


@pytest.mark.parametrize("endbz", ["##", "<endl/>", ""])
def test_record_that_starts_in_the_first_bytes_of_the_look_behind(port, endbz):
    # The first record of the second tile starts d bytes behind the place the look-behind reaches back
    # to (1,536 bytes in front of the tile), so the terminator in front of it stands in the first bytes
    # of the staged image, where the check of the end mark must not read in front of the image.
    e = endbz.encode()
    for d in range(0, 14):
        start = 16384 - 1536 + d                      # where the second record starts
        a = b"<a>1</a><ab>" + b"p" * (start - 12 - 5 - len(e) - 1) + b"</ab>" + e + b"\n"
        assert len(a) == start
        b = b"<a>22</a><ab>" + b"q" * 1600 + b"</ab><b>7.5</b>" + e + b"\n"    # ends in the second tile
        rest = b"".join(b"<a>%d</a><ab>v%d</ab><b>%d.25</b>" % (i, i, i) + e + b"\n" for i in range(400))
        data = a + b + rest
        got = gpu_xml(data, cases.SMALL, endbz=endbz)
        assert_batches_equal(got, port.parse_xml(data, cases.SMALL, endbz=endbz), f"endbz {endbz!r}, d {d}")


def test_column_bytes_around_the_staging_capacity(port):
    # a column's bytes of one batch around what a wave's staging area holds (1,536 bytes; 1,280 for inputs
    # of short records; 960 in the full write kernel): staged and copied out, or stored cell by cell
    for reclen, cells in ((256, range(20, 29)), (172, range(12, 19)), (128, range(8, 13))):
        for L in cells:
            recs = [b"<a>%04d</a><ab>" % (i % 10000) + bytes([65 + (i + k) % 26 for k in range(L)]) + b"</ab><b>1.5</b>" for i in range(1500)]
            recs = [r + b"x" * (reclen - len(r) - 8) + b"<endl/>\n" for r in recs]
            data = b"".join(recs)
            got = gpu_xml(data, cases.SMALL)
            assert_batches_equal(got, port.parse_xml(data, cases.SMALL), f"records of {reclen} bytes, cells of {L}")
    for L in range(8, 13):  # 128 lines to a batch
        lines = b"".join(b"%04d,%s,1.5\n" % (i % 10000, bytes([65 + (i + k) % 26 for k in range(L)]) + b" " * (50 - L)) for i in range(3000))
        got = gpu_csv(lines, ",", 3, bdelspace=True)
        assert_batches_equal(got, port.parse_csv(lines, ",", 3, bdelspace=True), f"lines with fields of {L}")


def test_separators_per_tile_around_the_list_capacity(port):
    # lines of 12 and 13 bytes with two separators each, mixed so that the tiles hold from fewer to more
    # separators than the sizing pass lists (2,568): the parallel field builder or the walk per tile
    rng = np.random.default_rng(11)
    for share in (0.0, 0.2, 0.5, 0.8, 1.0):
        lines = []
        for i in range(20000):
            lines.append(b"ab%02d,cd,efg\n" % (i % 100) if rng.random() < share else b"ab%02d,cd,ef\n" % (i % 100))
        data = b"".join(lines)
        got = gpu_csv(data, ",", 3)
        assert_batches_equal(got, port.parse_csv(data, ",", 3), f"share of 13-byte lines {share}")


def test_long_cells_bypass_staging(port):
    # cells larger than the LDS staging area take the direct store path
    recs = [b"<ab>" + bytes([65 + (i % 26)]) * 3000 + b"</ab><a>k%d</a><endl/>\n" % i for i in range(30)]
    data = b"".join(recs)
    got = gpu_xml(data, cases.SMALL)
    exp = port.parse_xml(data, cases.SMALL)
    assert_batches_equal(got, exp, "long cells")


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("sep", [",", "~!~", " "])
def test_fuzz_csv(port, seed, sep):
    data = cases.fuzz_csv(300 + seed, 3000, sep, 6)
    typed = [ALL_T[(seed + i) % len(ALL_T)] for i in range(6)]
    for bdel in (False, True):
        print("csv", seed, repr(sep), bdel, flush=True)
        got = gpu_csv(data, sep, 6, ilen=[0, 0, 3, 0, 0, 1], typed=typed, bdelspace=bdel)
        exp = port.parse_csv(data, sep, 6, ilen=[0, 0, 3, 0, 0, 1], typed=typed, bdelspace=bdel)
        assert_batches_equal(got, exp, f"csv {seed} {sep!r} {bdel}")


def test_mind_csv(port):
    data, n = gen.mind_csv(rows=30000)
    got = gpu_csv(data, ",", 10, typed=schema.MIND_CSV_TYPED)
    exp = port.parse_csv(data, ",", 10, typed=schema.MIND_CSV_TYPED)
    assert got.rows == n
    assert_batches_equal(got, exp, "mind csv")


def test_numeric_conversion_exact(port):
    # decimal -> binary64/binary32 must be correctly rounded for every digit count up to 38
    rng = np.random.default_rng(20240519)
    vals = []
    for nd in range(1, 39):
        for _ in range(60):
            digits = "".join(str(d) for d in rng.integers(0, 10, size=nd))
            k = int(rng.integers(0, nd + 1))
            s = digits[: nd - k] + "." + digits[nd - k:] if k else digits
            if rng.random() < 0.3:
                s = "-" + s
            vals.append(s)
    vals += ["9007199254740993", "9007199254740992.5", "0.1", "0.3", "4.35", "2.5", "16777217", "16777216.5",
             "0.000000000000000000000000000000000000011", "340282350000000000000000000000000000000"[:38]]
    cols = [Col("d", "char", 0), Col("f", "char", 0), Col("i", "char", 0)]
    data = b"".join(b"<d>%s</d><f>%s</f><i>%s</i><endl/>\n" % (v.encode(), v.encode(), v.encode()) for v in vals)
    typed = [capi_t("F64"), capi_t("F32"), capi_t("I64")]
    got = gpu_xml(data, cols, typed=typed, carry=False)
    exp = port.parse_xml(data, cols, typed=typed, carry=False)
    assert_batches_equal(got, exp, "numeric")


def capi_t(name):
    return {"I32": 1, "U32": 2, "I64": 3, "U64": 4, "F64": 5, "F32": 6, "BOOL": 7}[name]


def test_long_numeric_texts(port):
    # more than 38 digits: the multi-precision stage; includes halfway cases, subnormals that
    # are exact, overflow/underflow (ERANGE -> invalid) and very long tails
    rng = np.random.default_rng(7)
    vals = ["0." + "0" * 60 + "1", "1" + "0" * 40, "9007199254740993" + "0" * 30 + "." + "0" * 50 + "1",
            "9007199254740992.5" + "0" * 60, "9007199254740992.5" + "0" * 60 + "1",
            "1" + "0" * 308, "1" + "0" * 309, "17976931348623157" + "0" * 292, "17976931348623159" + "0" * 292,
            "0." + "0" * 307 + "22250738585072014", "0." + "0" * 323 + "49406564584124654",
            "0." + "0" * 400 + "1", "0." + "0" * 37 + "11754943508222875", "0." + "0" * 44 + "14",
            "34028234663852886" + "0" * 22, "34028235677973366" + "0" * 22, "3" * 900, "0." + "3" * 2000,
            "123456789" * 100 + "." + "987654321" * 100]
    # 2^-1074 and 2^-149 written out exactly
    from fractions import Fraction
    for e in (1074, 149, 1070, 140):
        fr = Fraction(1, 2 ** e)
        digs = str(fr.numerator * 10 ** e // fr.denominator).rjust(e, "0")
        vals.append("0." + digs)
        vals.append("0." + digs + "1")
    for _ in range(150):
        nd = int(rng.integers(39, 400))
        digits = "".join(str(d) for d in rng.integers(0, 10, size=nd))
        k = int(rng.integers(0, nd + 1))
        vals.append((digits[: nd - k] + "." + digits[nd - k:]) if k else digits)
    cols = [Col("d", "char", 0), Col("f", "char", 0)]
    data = b"".join(b"<d>%s</d><f>%s</f><endl/>\n" % (v.encode(), v.encode()) for v in vals)
    got = gpu_xml(data, cols, typed=[5, 6], carry=False)
    exp = port.parse_xml(data, cols, typed=[5, 6], carry=False)
    assert_batches_equal(got, exp, "long numeric")


def test_parse_file_and_accessors(tmp_path, port):
    data, n = gen.mind_xml(rows=300)
    f = tmp_path / "ZHOBTMIND_20240519162835_togxpt_1.xml"
    f.write_bytes(bytes(data))
    p = capi.Parser.xml(schema.ZHOBTMIND)
    b = p.parse_file(str(f))
    exp = port.parse_xml(data, schema.ZHOBTMIND)
    assert b.rows == n
    for r in (0, 1, 150, 299):
        for c in range(11):
            assert b.cell(r, c) == exp.cell(r, c)
    slab, stride = b.column_fixed(1)   # ddatetime: collen 14 -> stride 15
    assert stride == 15
    for r in (0, 7, 299):
        assert bytes(slab[r]).rstrip(b"\0") == exp.cell(r, 1)
    with pytest.raises(capi.MdcError) as e:
        p.parse_file(str(tmp_path / "missing.xml"))
    assert e.value.code == capi.MDC_FILE_OPEN_FAILED
    p.close()


def test_workspace_follows_the_input_shape(port):
    # one parser, inputs of very different size and bytes per column, in both directions
    small = b"<a>1</a><b>2</b><endl/>\n" * 4000
    big = b"".join(b"<a>abcde</a><b>%d.25</b><ab>%s</ab><endl/>\n" % (i, b"x" * (i % 200)) for i in range(6000))
    p = capi.Parser.xml(cases.SMALL)
    p.parse_host(small).free()
    got = p.parse_host(big).to_host()
    assert_batches_equal(got, port.parse_xml(big, cases.SMALL), "grown")
    got = p.parse_host(small).to_host()
    assert_batches_equal(got, port.parse_xml(small, cases.SMALL), "shrunk")
    p.close()


def test_tag_template_follows_the_input(port):
    """One parser, inputs whose records have different tag sequences, one after the other: the
    sizing pass keeps the template of an earlier input while tiles follow it, sends the tiles of a
    different input through the table lookup and renews the template on the parse after that."""
    import re
    a, _ = gen.mind_xml(rows=40000)
    a = a.tobytes()
    b = re.sub(rb"(<t>[^<]*</t>)(<p>[^<]*</p>)", rb"\2\1", a)           # two tags the other way round
    c = re.sub(rb"<wd>", rb"<x>1</x><wd>", a)                              # a tag the schema does not know
    d = re.sub(rb"(<u>[^<]*</u>)", rb"\1\1", a)                           # a tag twice
    e = b"".join((x + b"\n") if i % 7 else (x.replace(b"</r>", b"</r><r>9</r>") + b"\n")
                 for i, x in enumerate(a.split(b"\n")[:-1])) + a.split(b"\n")[-1]  # some records differ
    tiny = b"<obtid>58015</obtid><endl/>\n"                                 # no second record: no template
    p = capi.Parser.xml(schema.ZHOBTMIND)
    for name, data in (("a", a), ("b", b), ("b again", b), ("a again", a), ("c", c), ("c again", c), ("d", d),
                       ("tiny", tiny), ("e", e), ("e again", e), ("a at the end", a)):
        got = p.parse_host(data).to_host()
        assert_batches_equal(got, port.parse_xml(data, schema.ZHOBTMIND), name)
    p.close()


@pytest.mark.parametrize("ncols", [20, 33, 64, 128, 129, 200, 256])
def test_wide_schemas_with_and_without_room_for_the_template(port, ncols):
    """Regular records under schemas of growing width: the template's entries and column plan fit
    the sizing pass's LDS for the narrow ones only; beyond that (and beyond 64 events per record)
    every tile takes the table lookup.  Every third column is missing (values carry over)."""
    cols = [Col("c%d" % i, ("number", "date", "char")[i % 3], (22, 14, 8)[i % 3]) for i in range(ncols)]
    recs = []
    for r in range(3000):
        recs.append(b"".join(b"<c%d>%d.%d</c%d>" % (i, (r * 7 + i) % 1000, i % 10, i) for i in range(ncols) if i % 3 != 1 or r % 50 == 0)
                    + b"<endl/>\n")
    data = b"<data>\n" + b"".join(recs) + b"</data>\n"
    got = gpu_xml(data, cols)
    assert_batches_equal(got, port.parse_xml(data, cols), "ncols=%d" % ncols)


def test_wide_schema_with_records_longer_than_a_tile(port):
    """256 columns with cells of 20 to 200 bytes: every record is some 30 KiB, longer than a tile and than
    every look-behind, so all of them go to the long-record kernel; short records in between."""
    rng = random.Random(256)
    cols = [Col("k%d" % i, ("char", "number", "date")[i % 3], (0, 22, 14)[i % 3]) for i in range(256)]
    recs = []
    for r in range(24):
        short = r % 5 == 4
        recs.append(b"".join(b"<k%d>%s</k%d>" % (i, (b"%d" % rng.randrange(10 ** 9)) * (1 if short else rng.randrange(2, 20)), i)
                             for i in range(256) if rng.random() < 0.9) + b"<endl/>\n")
    data = b"".join(recs)
    assert max(map(len, recs)) > 20000
    got = gpu_xml(data, cols)
    assert_batches_equal(got, port.parse_xml(data, cols), "256 columns, long records")


@pytest.mark.parametrize("sep", [",", "~!~"])
@pytest.mark.parametrize("nfields", [129, 256])
def test_wide_delimited_records(port, nfields, sep):
    """More fields than one pointer table of 128 held: lines with fewer fields than the schema, with exactly as
    many and with more, cells of every typed kind, fields cut to a length."""
    rng = random.Random(nfields)
    typed = [ALL_T[i % len(ALL_T)] for i in range(nfields)]
    ilen = [(0, 0, 3, 0, 7)[i % 5] for i in range(nfields)]
    lines = []
    for r in range(1500):
        k = rng.choice([nfields, nfields, nfields, nfields - 1, nfields + 3, rng.randrange(1, nfields)])
        cells = []
        for i in range(k):
            kind = rng.random()
            cells.append(str(rng.randrange(-10 ** 6, 10 ** 6)) if kind < 0.5 else "%.3f" % rng.uniform(-1e3, 1e3) if kind < 0.7
                         else "" if kind < 0.8 else rng.choice(["true", " x ", "20240519162800", "abc def", "-"]))
        lines.append(sep.join(cells))
    data = ("\n".join(lines) + "\n").encode()
    for bdel in (False, True):
        got = gpu_csv(data, sep, nfields, ilen=ilen, typed=typed, bdelspace=bdel)
        exp = port.parse_csv(data, sep, nfields, ilen=ilen, typed=typed, bdelspace=bdel)
        assert_batches_equal(got, exp, f"{nfields} fields {sep!r} {bdel}")


@pytest.mark.parametrize("nbytes", [(32 << 20) - 5, (32 << 20) + 1, (56 << 20) + 4097, 100 << 20])
def test_large_host_buffers_go_up_in_pieces(nbytes):
    # a pageable buffer of 32 MiB or more is staged in 8 MiB pieces by several host threads: the parse
    # must equal the parse of the same bytes put into device memory with one copy
    data, n = gen.mind_xml(nbytes=nbytes)
    p = capi.Parser.xml(schema.ZHOBTMIND)
    d = capi.DeviceBuffer(data)
    want = p.parse_device(d.ptr, d.n).to_host()
    for _ in range(2):  # (the second call reuses the pinned buffers the first one left)
        got = p.parse_host(data).to_host()
        assert got.rows == n
        assert_batches_equal(got, want, f"host buffer of {nbytes} bytes")
    d.free()
    p.close()


def test_many_tiles(port):
    data, n = gen.mind_xml(nbytes=48 << 20)
    p = capi.Parser.xml(schema.ZHOBTMIND)
    got = p.parse_host(data).to_host()
    exp = port.parse_xml(data, schema.ZHOBTMIND)
    assert got.rows == n
    assert_batches_equal(got, exp, "48 MiB")
    p.close()


def test_device_resident_input(port):
    data, n = gen.mind_xml(nbytes=4 << 20)
    p = capi.Parser.xml(schema.ZHOBTMIND)
    d = capi.DeviceBuffer(data)
    b = p.parse_device(d.ptr, d.n)
    got = b.to_host()
    exp = port.parse_xml(data, schema.ZHOBTMIND)
    assert_batches_equal(got, exp, "device resident")
    t = p.timing()
    assert t.rows == n and t.bytes_in == len(data) and t.write_ms > 0
    d.free()
    p.close()


def test_tiles_without_a_cell_map_take_the_full_write_kernel(port):
    # far more records per byte than the map is sized for: most tiles find no room in it
    lines = b"".join(b"%d,%d\n" % (i % 7, i % 3) for i in range(150000))
    typed = [schema.T_I64, schema.T_F64]
    p = capi.Parser.csv(",", 2, typed=typed)
    got = p.parse_host(lines).to_host()
    assert p.timing().full_write == 1
    assert_batches_equal(got, port.parse_csv(lines, ",", 2, typed=typed), "short lines")
    # the next parse sizes the map from this one's records per byte: every tile fits
    got = p.parse_host(lines).to_host()
    assert p.timing().full_write == 0
    assert_batches_equal(got, port.parse_csv(lines, ",", 2, typed=typed), "short lines, second parse")
    p.close()
    recs = b"<a>1</a><endl/>\n" * 60000 + b"<endl/>\n" * 50000 + b"<b>2.5</b><endl/>\n" * 1000
    got = gpu_xml(recs, cases.SMALL)
    assert_batches_equal(got, port.parse_xml(recs, cases.SMALL), "short records")


def test_cell_longer_than_a_map_word_holds(port):
    # close tag in front of the open tag: the cell is the rest of a 20,000-byte record
    filler = b"x" * 19900
    rec = b"</a><a>start" + filler + b"<b>7</b><endl/>\n"
    data = b"<a>1</a><b>2</b><endl/>\n" * 300 + rec + b"<a>3</a><b>4</b><endl/>\n" * 300
    cols = [Col("a", "char", 0), Col("b", "number", 22)]
    p = capi.Parser.xml(cols)
    got = p.parse_host(data).to_host()
    assert p.timing().full_write == 1
    assert_batches_equal(got, port.parse_xml(data, cols), "long cell")
    p.close()


def test_file_stream_plain_gzip_and_missing(tmp_path, port):
    import gzip

    paths, expect = [], []
    for i in range(7):
        data, n = gen.mind_xml(rows=500 + 311 * i, first_row=10000 * i)
        raw = data.tobytes()
        if i == 3:
            paths.append(str(tmp_path / "missing.xml"))
            expect.append(None)
            continue
        if i % 2 == 0:
            f = tmp_path / f"obs_{i}.xml.gz"
            if i == 4:  # two gzip members in one file
                half = raw.rfind(b"\n", 0, len(raw) // 2) + 1
                f.write_bytes(gzip.compress(raw[:half], 6) + gzip.compress(raw[half:], 1))
            else:
                f.write_bytes(gzip.compress(raw, 6))
        else:
            f = tmp_path / f"obs_{i}.xml"
            f.write_bytes(raw)
        paths.append(str(f))
        expect.append(port.parse_xml(data, schema.ZHOBTMIND))
    empty = tmp_path / "empty.xml"
    empty.write_bytes(b"")
    paths.append(str(empty))
    expect.append(port.parse_xml(b"", schema.ZHOBTMIND))
    bad = tmp_path / "broken.xml.gz"
    bad.write_bytes(gzip.compress(b"<obtid>1</obtid><endl/>\n" * 1000)[:-20])
    paths.append(str(bad))
    expect.append("io")
    for threads in (1, 3):
        p = capi.Parser.xml(schema.ZHOBTMIND)
        seen = []
        for idx, b in p.stream(paths, threads=threads):
            seen.append(idx)
            if expect[idx] is None:
                assert isinstance(b, capi.MdcError) and b.code == capi.MDC_FILE_OPEN_FAILED
            elif isinstance(expect[idx], str):
                assert isinstance(b, capi.MdcError) and b.code == capi.MDC_ERR_IO
            else:
                assert_batches_equal(b.to_host(), expect[idx], f"file {idx}, {threads} threads")
                b.free()
        assert seen == list(range(len(paths)))
        p.close()


def test_parse_file_reads_gzip(tmp_path, port):
    import gzip

    data, n = gen.mind_xml(rows=3000)
    f = tmp_path / "obs.xml.gz"
    f.write_bytes(gzip.compress(data.tobytes(), 6))
    p = capi.Parser.xml(schema.ZHOBTMIND)
    got = p.parse_file(str(f)).to_host()
    assert_batches_equal(got, port.parse_xml(data, schema.ZHOBTMIND), "gzip file")
    with pytest.raises(capi.MdcError) as e:
        p.parse_file(str(tmp_path / "nope.xml"))
    assert e.value.code == capi.MDC_FILE_OPEN_FAILED
    p.close()


def test_delimited_tiles_dense_with_separators(port):
    # more separators per tile than the parallel field builder keeps: those tiles use the walker
    dense = b",,,,,,,,,,,,,,,\n" * 6000
    mixed = dense + b"".join(b"%d,%d.5,x\n" % (i, i) for i in range(5000)) + dense[:4000]
    for data in (dense, mixed):
        got = gpu_csv(data, ",", 4, typed=[schema.T_I64, schema.T_F64, schema.T_NONE, schema.T_NONE])
        exp = port.parse_csv(data, ",", 4, typed=[schema.T_I64, schema.T_F64, schema.T_NONE, schema.T_NONE])
        assert_batches_equal(got, exp, "dense separators")


@pytest.mark.parametrize("sep", ["~!~", "||", "aa", "::", ", ", "<->", "abab", "~~~~"])
def test_separators_of_several_bytes_many_tiles(port, sep):
    # the sizing pass finds separators of up to four bytes with byte masks; a tile in which two matches
    # overlap ("~!~!~": the reference's search goes on behind the first) takes the walk instead
    rng = np.random.default_rng(sum(sep.encode()))
    alphabet = np.frombuffer((sep + "xyz01 .").encode(), dtype=np.uint8)
    lines = []
    for i in range(30000):
        kind = i % 11
        if kind == 3:  # bytes of the separator in any order: overlapping and partial matches
            body = bytes(alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 60)))])
        else:
            f = [str(int(rng.integers(0, 10 ** int(rng.integers(1, 7))))) for _ in range(int(rng.integers(1, 8)))]
            if kind == 5:
                f[int(rng.integers(0, len(f)))] = ""
            if kind == 7:
                f.append(sep[:-1])  # a separator cut short at the end of the line
            body = sep.join(f).encode()
        lines.append(body.replace(b"\n", b" "))
    data = b"\n".join(lines) + b"\n"
    typed = [schema.T_I64, schema.T_F64, schema.T_NONE, schema.T_U32, schema.T_NONE]
    for bdel in (False, True):
        got = gpu_csv(data, sep, 5, typed=typed, bdelspace=bdel, ilen=[0, 0, 4, 0, 0])
        exp = port.parse_csv(data, sep, 5, ilen=[0, 0, 4, 0, 0], typed=typed, bdelspace=bdel)
        assert_batches_equal(got, exp, f"separator {sep!r}, bdelspace {bdel}")


@pytest.mark.parametrize("sep", ["~!~", "aa", "abab"])
def test_overlapping_separator_matches_at_every_place_of_a_lane(port, sep):
    # a run of overlapping matches moved byte by byte across the 64-byte pieces the lanes classify, the
    # waves' quarters of a tile and the tile's end
    run = (sep * 3)[: 2 * len(sep) + 1].encode() if sep != "aa" else b"aaaaa"
    p = capi.Parser.csv(sep, 4)
    for at in list(range(16384 - 80, 16384 + 8)) + list(range(4096 - 70, 4096 + 6)):
        head = b"1,2\n" * 3
        pad = at - len(head) - 2
        line = b"7" + sep.encode() + b"x" * (pad - 1 - len(sep)) + run + b"y" + sep.encode() + b"9\n"
        data = head + line + b"5" + sep.encode() + b"6\n" * 1 + (b"3" + sep.encode() + b"4\n") * 50
        got = p.parse_host(data).to_host()
        exp = port.parse_csv(data, sep, 4)
        assert_batches_equal(got, exp, f"separator {sep!r}, run at {at}")
    p.close()


@pytest.mark.parametrize("bdel", [False, True])
def test_delimited_long_batch_with_trim(port, bdel):
    rng = np.random.default_rng(5)
    lines = []
    for i in range(40000):
        f = [str(int(rng.integers(0, 10 ** int(rng.integers(1, 9))))) for _ in range(int(rng.integers(1, 9)))]
        if i % 7 == 0:
            f = [" " * int(rng.integers(0, 3)) + x + " " * int(rng.integers(0, 3)) for x in f]
        lines.append(",".join(f))
    data = ("\n".join(lines) + "\n").encode()
    typed = [schema.T_I64, schema.T_F64, schema.T_U32, schema.T_NONE, schema.T_I32, schema.T_NONE]
    got = gpu_csv(data, ",", 6, typed=typed, bdelspace=bdel, ilen=[0, 5, 0, 3, 0, 0])
    exp = port.parse_csv(data, ",", 6, ilen=[0, 5, 0, 3, 0, 0], typed=typed, bdelspace=bdel)
    assert_batches_equal(got, exp, "delimited, many tiles")


def _whole_equals_parts_and_checker_windows(port, data, n, p, parse_window, endbz, nwindows=16, window=4 << 20):
    """The parse of the whole batch equals the concatenation of the parses of four parts cut at record
    terminators (rows match the generator's count), and windows of the batch spread over its whole length,
    cut at record boundaries and parsed by the CPU checker, equal the rows of the whole parse they cover."""
    from dsp_amd import shard

    whole_b = p.parse_host(data)
    assert whole_b.rows == n
    whole = whole_b.to_host()
    whole_b.free()
    row = 0
    pos = [0] * whole.ncols
    for lo, hi in shard.split_points(data, 4, endbz=endbz):
        b = p.parse_host(data[lo:hi])
        part = b.to_host()
        b.free()
        m = part.rows
        assert np.array_equal(part.rec_off[1:] + np.uint64(lo), whole.rec_off[row + 1: row + m + 1])
        assert np.array_equal(part.valid, whole.valid[row: row + m])
        assert np.array_equal(part.found, whole.found[row: row + m])
        for c in range(whole.ncols):
            nb = int(part.off[c][m])
            assert np.array_equal(part.off[c][1:].astype(np.uint64) + np.uint64(pos[c]),
                                  whole.off[c][row + 1: row + m + 1].astype(np.uint64)), c
            assert np.array_equal(part.data[c][:nb], whole.data[c][pos[c]: pos[c] + nb]), c
            if whole.val[c] is not None and len(whole.val[c]):
                assert np.array_equal(part.val[c][:m], whole.val[c][row: row + m]), c
            pos[c] += nb
        row += m
        del part
    assert row == n
    # windows against the CPU checker: window k starts at the first record that begins at or behind k/nwindows of the batch
    for k in range(nwindows):
        want = len(data) * k // nwindows
        r0 = int(np.searchsorted(whole.rec_off, want, side="left"))
        if r0 >= n:
            continue
        lo = int(whole.rec_off[r0])
        r1 = int(np.searchsorted(whole.rec_off, min(lo + window, len(data)), side="right")) - 1
        hi = int(whole.rec_off[r1])
        if k == 0:
            lo = 0  # (the bytes in front of the first record belong to it: "<data>\n")
        exp = parse_window(np.ascontiguousarray(data[lo:hi]))
        m = exp.rows
        assert m == r1 - r0 and m > 0, (k, m, r1 - r0)
        assert np.array_equal(exp.found, whole.found[r0: r0 + m]), k
        assert np.array_equal(exp.valid, whole.valid[r0: r0 + m]), k
        for c in range(exp.ncols):
            base = whole.off[c][r0].astype(np.uint64)
            assert np.array_equal(exp.off[c].astype(np.uint64) + base, whole.off[c][r0: r0 + m + 1].astype(np.uint64)), (k, c)
            assert np.array_equal(exp.data[c], whole.data[c][int(base): int(base) + int(exp.off[c][m])]), (k, c)
            if exp.val[c] is not None:
                assert np.array_equal(exp.val[c], whole.val[c][r0: r0 + m]), (k, c)


def test_full_size_batch_equals_its_parts(port):
    """1 GiB of tagged records: the size the metric is quoted on (MDC_FULL_BYTES: a larger batch on request)."""
    data, n = gen.mind_xml(nbytes=int(os.environ.get("MDC_FULL_BYTES", 1 << 30)))
    p = capi.Parser.xml(schema.ZHOBTMIND)
    _whole_equals_parts_and_checker_windows(port, data, n, p, lambda d: port.parse_xml(d, schema.ZHOBTMIND), "<endl/>")
    p.close()


def test_full_size_delimited_batch_equals_its_parts(port):
    """4 GiB of delimited records: the size of the delimited configuration (MDC_FULL_CSV_BYTES: another size)."""
    from concurrent.futures import ThreadPoolExecutor
    nbytes = int(os.environ.get("MDC_FULL_CSV_BYTES", 4 << 30))
    part = 256 << 20
    sizes = [min(part, nbytes - i * part) for i in range((nbytes + part - 1) // part)]
    with ThreadPoolExecutor(max_workers=16) as ex:  # (the generator is C code that runs without the interpreter lock)
        made = list(ex.map(lambda i: gen.mind_csv(nbytes=sizes[i], first_row=i * 5_000_000), range(len(sizes))))
    data = np.concatenate([m[0] for m in made]) if len(made) > 1 else made[0][0]
    n = sum(m[1] for m in made)
    del made
    typed = schema.MIND_CSV_TYPED
    p = capi.Parser.csv(",", 10, typed=typed)
    _whole_equals_parts_and_checker_windows(port, data, n, p, lambda d: port.parse_csv(d, ",", 10, typed=typed), "")
    p.close()


@pytest.mark.parametrize("typed_text", [False, True])
def test_nul_in_the_part_of_a_record_that_lies_in_the_tile_before(port, typed_text):
    """A record that starts in one 16 KiB tile and ends in the next, with its NUL byte in the first
    part: the tile that owns the record must cut the text cell at the NUL like c_str() does, whether
    or not it holds a NUL byte of its own, and the rows behind it must keep their places."""
    cols = [Col("a", "char", 10), Col("n", "number", 22), Col("t", "char", 0)]
    typed = [schema.T_I32 if typed_text else schema.T_NONE, schema.T_F64, schema.T_NONE]
    plain = b"<a>abc</a><n>12.5</n><t>text</t><endl/>\n"
    for tail in (40, 300, 3000):  # bytes of the straddling record in the second tile
        head = plain * ((16384 - 200) // len(plain))
        filler = 16384 + tail - len(head) - len(b"<a>x\x00y</a><n>7</n><t></t><endl/>\n")
        straddler = b"<a>x\x00y</a><n>7</n><t>" + b"q" * filler + b"</t><endl/>\n"
        data = np.frombuffer(head + straddler + plain * 600, dtype=np.uint8)
        assert 8 < 16384 - len(head) < 300 and len(head) + len(straddler) == 16384 + tail
        exp = port.parse_xml(data, cols, typed=typed)
        p = capi.Parser.xml(cols, typed=typed)
        try:
            assert_batches_equal(p.parse_host(data).to_host(), exp, f"tail {tail}")
        finally:
            p.close()
        r = len(head) // len(plain)
        assert exp.cell(r, 0) == b"x" and exp.cell(r + 1, 0) == b"abc"


def test_template_is_taken_from_another_record_when_the_second_is_the_odd_one(port):
    """Regular records whose second one lacks a field: the first parse samples it and no tile follows;
    the parse after that samples the third record and nearly every tile follows (mdc_timing's
    lookup_tiles), with the same results."""
    import re
    a, _ = gen.mind_xml(rows=30000)
    recs = a.tobytes().split(b"<endl/>\n")
    recs[1] = re.sub(rb"<wd>[^<]*</wd>", b"", recs[1])
    data = np.frombuffer(b"<endl/>\n".join(recs), dtype=np.uint8)
    exp = port.parse_xml(data, schema.ZHOBTMIND)
    p = capi.Parser.xml(schema.ZHOBTMIND)
    try:
        left = []
        for i in range(3):
            b = p.parse_host(data)
            t = p.timing()
            left.append((t.lookup_tiles, t.tiles))
            assert_batches_equal(b.to_host(), exp, f"parse {i}")
        assert left[0][0] > left[0][1] // 2, left      # the odd record as the sample: the tiles leave the template
        assert left[1][0] <= 2 and left[2][0] <= 2, left  # another sample: only the odd record's tile does
    finally:
        p.close()


@pytest.mark.parametrize("delta", [-3, 5])
@pytest.mark.parametrize("full_write", [False, True])
def test_write_pass_refuses_a_tile_the_sizing_pass_measured_differently(port, delta, full_write):
    # The arenas are sized exactly by the sizing pass; the write pass measures every cell again with
    # code of its own.  The host-side hook changes one counter between the passes, as a sizing pass
    # that disagreed would have left it: the write pass must report an internal error (and store
    # nothing beyond what was sized) instead of handing out a batch, and the parser must stay usable.
    L = capi.lib()
    os.environ["MDC_TEST_HOOKS"] = "1"  # (the hook is refused without it; read when a hook is first called)
    L.mdc_internal_skew_counts.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    if full_write:  # no room in the cell map: every tile goes through tile_kernel<true>
        data = b"".join(b"%d,%d\n" % (i % 7, i % 3) for i in range(150000))
        p = capi.Parser.csv(",", 2, typed=[schema.T_I64, schema.T_F64])
        exp = port.parse_csv(data, ",", 2, typed=[schema.T_I64, schema.T_F64])
        col = 1
    else:
        data, _ = gen.mind_xml(rows=2000)
        p = capi.Parser.xml(schema.ZHOBTMIND)
        p.parse_host(data).free()  # (sizes the cell map: the next parses take the emit kernel)
        exp = port.parse_xml(data, schema.ZHOBTMIND)
        col = 2
    capi.check(L.mdc_internal_skew_counts(p.h, 1, col, delta))
    with pytest.raises(capi.MdcError) as e:
        p.parse_host(data)
    assert e.value.code == capi.MDC_ERR_INTERNAL
    if not full_write:
        got = p.parse_host(data)
        assert p.timing().full_write == 0
        assert_batches_equal(got.to_host(), exp, "parse after a refused one")
        got.free()
    p.close()


def test_bind_slabs_of_every_column_with_carried_values(port):
    # mdc_batch_column_fixed for every column and row of records with missing tags, so that carried
    # values (longer than the column's own collen in places) meet the slot size
    recs = []
    for i in range(700):
        f = [b"<obtid>5%04d</obtid>" % i, b"<ddatetime>2024-05-19 16:%02d:00</ddatetime>" % (i % 60), b"<t>%d.%d</t>" % (i % 50, i % 10),
             b"<p>1013.2</p>", b"<u>65</u>", b"<wd>180</wd>", b"<wf>3.4</wf>", b"<r>0.0</r>", b"<vis>10000.5</vis>",
             b"<upttime>20240519162800</upttime>", b"<keyid>%d</keyid>" % (123456789012 + i)]
        keep = [k for k in range(11) if (i * 7 + k * 3) % 5 != 0 or k == 0]
        recs.append(b"".join(f[k] for k in keep) + b"<endl/>\n")
    data = b"<data>\n" + b"".join(recs) + b"</data>\n"
    p = capi.Parser.xml(schema.ZHOBTMIND)
    b = p.parse_host(data)
    exp = port.parse_xml(data, schema.ZHOBTMIND)
    assert b.rows == exp.rows == 700
    for c, col in enumerate(schema.ZHOBTMIND):
        slab, stride = b.column_fixed(c)
        assert stride == col.collen + 1
        for r in range(exp.rows):
            cell = exp.cell(r, c)[:col.collen]  # (a carried value is cut to the slot: see mdc_parse.h)
            row = bytes(slab[r])
            assert row[:len(cell)] == cell and row[len(cell):] == b"\0" * (stride - len(cell)), (r, c)
    b.free()
    p.close()


@pytest.mark.parametrize("seplen", [5, 16, 17, 40, 255])
def test_long_separators(port, seplen):
    """ccmdstr::splittocmd (public/_public.cpp:274-302) puts no bound on the separator: beyond four bytes it is
    found by the walk over each record, beyond sixteen its text lies behind the schema's tables."""
    rng = np.random.default_rng(seplen)
    sep = "".join(chr(int(c)) for c in rng.integers(33, 127, seplen))
    lines = []
    for i in range(3000):
        f = [str(int(rng.integers(0, 10 ** int(rng.integers(1, 8))))) for _ in range(int(rng.integers(1, 6)))]
        if i % 11 == 0:
            f.append(sep[: seplen // 2])  # (half a separator is field text)
        lines.append(sep.join(f))
    data = ("\n".join(lines) + "\n").encode("latin-1")
    typed = [schema.T_I64, schema.T_F64, schema.T_NONE, schema.T_U32, schema.T_NONE]
    got = gpu_csv(data, sep, 5, typed=typed, bdelspace=seplen % 2 == 1, ilen=[0, 0, 4, 0, 0])
    exp = port.parse_csv(data, sep, 5, ilen=[0, 0, 4, 0, 0], typed=typed, bdelspace=seplen % 2 == 1)
    assert_batches_equal(got, exp, f"separator of {seplen} bytes")


def test_separator_with_a_newline_never_matches_in_a_line(port):
    # a line holds no newline, so a separator with one is never found: every line is one field (the reference
    # takes such a separator; it used to be refused here)
    data = b"".join(b"a%d|\n|b%d\n" % (i, i) for i in range(3000))
    got = gpu_csv(data, "|\n|", 3, typed=[schema.T_NONE, schema.T_NONE, schema.T_I64])
    exp = port.parse_csv(data, "|\n|", 3, typed=[schema.T_NONE, schema.T_NONE, schema.T_I64])
    assert got.rows == 6000
    assert_batches_equal(got, exp, "separator with a newline")


@pytest.mark.parametrize("marklen", [9, 32, 33, 100, 255])
def test_long_end_marks(port, marklen):
    """cifile::readline (public/_public.cpp:866-892) takes an end mark of any length: beyond 32 bytes its text
    lies behind the schema's tables.  Marks cut short, and marks with text behind them, end no record."""
    rng = np.random.default_rng(marklen)
    mark = "".join(chr(int(c)) for c in rng.integers(33, 127, marklen)).replace("<", "(")
    recs = []
    for i in range(4000):
        body = "<a>%d</a><b>%d.5</b>" % (i, i % 977)
        if i % 13 == 0:
            body += mark[:-1] + "\n"          # (all but the mark's last byte, then a newline: no terminator)
        if i % 17 == 0:
            body += mark + "x\n"              # (the mark with text behind it: no terminator)
        recs.append(body + mark + "\n")
    data = ("".join(recs) + "<a>tail</a>" + mark).encode("latin-1")  # (a last record without its newline is dropped)
    got = gpu_xml(data, cases.SMALL, endbz=mark)
    exp = port.parse_xml(data, cases.SMALL, endbz=mark)
    assert got.rows == 4000
    assert_batches_equal(got, exp, f"end mark of {marklen} bytes")
    # ... and through a long record (the long-record kernel reads the mark as well)
    big = ("<a>1</a>" + "j" * 40000 + mark + "\n").encode("latin-1")
    got = gpu_xml(data[: len(data) // 2] + big + data[len(data) // 2:], cases.SMALL, endbz=mark)
    exp = port.parse_xml(data[: len(data) // 2] + big + data[len(data) // 2:], cases.SMALL, endbz=mark)
    assert_batches_equal(got, exp, f"end mark of {marklen} bytes, long record")


NEWLINE_MARKS = ["<endl/>\n<next/>", "a\nb", "\n", "x\n", "\n##", "<e>\n\n", "<end of\nthis record\nand of nothing else/>" * 3]


@pytest.mark.parametrize("mark", NEWLINE_MARKS, ids=[repr(m)[:24] for m in NEWLINE_MARKS])
def test_end_marks_that_hold_a_newline(port, mark):
    """cifile::readline (public/_public.cpp:872-889) gathers lines until the gathered text ends with the mark, so a
    mark may hold newlines (even end with one, which then takes an empty line behind it).  Records with parts of the
    mark, with newlines of their own and with the mark followed by text; through the delimited mode, a long record
    and the record split alone as well."""
    rng = random.Random(len(mark))
    recs = []
    for i in range(5000):
        body = "<a>%d</a><ab>%s</ab><b>%d.25</b>" % (i, "v\nw" if i % 7 == 0 else "v", i % 977)
        if i % 13 == 0:
            body += mark[:-1] + "Q\nz"       # (the mark cut short, then other text: ends nothing)
        if i % 17 == 0:
            body += mark + "x\nz"            # (the mark with text behind it: ends nothing)
        if i % 19 == 0:
            body += "y\ny" * rng.randrange(1, 3)  # (newlines of the record's own, with text on both sides)
        recs.append(body + mark + "\n")
    data = ("".join(recs) + "<a>tail</a>" + mark).encode()
    exp = port.parse_xml(data, cases.SMALL, endbz=mark)
    assert exp.rows >= 5000, exp.rows
    got = gpu_xml(data, cases.SMALL, endbz=mark)
    assert_batches_equal(got, exp, f"end mark {mark!r}")
    # the same through a long record, and without the carry-forward
    half = len("".join(recs[:2500]).encode())
    big = data[:half] + ("<a>1</a><ab>" + "j\n" * 20000 + "</ab>" + mark + "\n").encode() + data[half:]
    assert_batches_equal(gpu_xml(big, cases.SMALL, endbz=mark, carry=False), port.parse_xml(big, cases.SMALL, endbz=mark, carry=False),
                         f"end mark {mark!r}, long record")
    # the record split alone (what cifile::readline of the compat header asks for)
    p = capi.Parser.xml(cases.SMALL[:1], endbz=mark, flags=capi.MDC_F_RECORDS_ONLY)
    b = p.parse_host(data).to_host()
    p.close()
    assert b.rows == exp.rows and np.array_equal(b.rec_off, exp.rec_off)


def test_overlapping_end_marks_are_refused(port):
    """Two occurrences of a mark that overlap (a newline inside the second follows the first): cifile::readline
    takes the first and starts the next record behind it, so the second ends nothing.  That is a left-to-right
    decision; the parser finds such an input and refuses it instead of splitting it wrongly."""
    for mark, bad in (("\n", b"<a>1</a>\n\n\n<a>2</a>\n\n"), ("a\na", b"<a>1</a>a\na\na\n<a>2</a>a\na\n"),
                      ("q\nq", b"<a>1</a>q\nq\n" * 3000 + b"<a>2</a>q\nq\nq\n" + b"<a>3</a>q\nq\n" * 3000)):
        assert port.parse_xml(bad, cases.SMALL, endbz=mark).rows >= 2  # (the reference's answer exists)
        p = capi.Parser.xml(cases.SMALL, endbz=mark)
        with pytest.raises(capi.MdcError) as e:
            p.parse_host(bad)
        assert e.value.code == capi.MDC_ERR_ARG and "overlap" in str(e.value)
        # the parser stays usable
        good = b"<a>1</a>" + mark.encode() + b"\n<a>2</a>" + mark.encode() + b"\n"
        assert_batches_equal(p.parse_host(good).to_host(), port.parse_xml(good, cases.SMALL, endbz=mark), f"after a refusal, {mark!r}")
        p.close()


def test_records_of_two_shapes_stop_the_template_attempts(port):
    """Every third record lacks a field, so nearly every tile holds a record that leaves any template: after two
    such parses the parser goes without a template for a while (the compare every tile tries first is lost
    time), and takes it up again later.  The cells are the checker's throughout."""
    import re
    data, n = gen.mind_xml(nbytes=3 << 20)
    recs = data.tobytes().split(b"<endl/>\n")
    for i in range(0, len(recs) - 1, 3):
        recs[i] = re.sub(rb"<wd>[^<]*</wd>", b"", recs[i])
    two = np.frombuffer(b"<endl/>\n".join(recs), dtype=np.uint8)
    exp = port.parse_xml(two, schema.ZHOBTMIND)
    ntiles = (two.size + 16383) // 16384
    p = capi.Parser.xml(schema.ZHOBTMIND)
    seen = []
    for k in range(22):
        got = p.parse_host(two).to_host()
        if k in (0, 1, 2, 3, 18, 19, 21):
            assert_batches_equal(got, exp, f"parse {k}")
        seen.append(p.timing().lookup_tiles)
    assert seen[0] > ntiles // 2 and seen[1] > ntiles // 2      # tried and left by most tiles, twice
    assert seen[2:18] == [ntiles] * 16                           # sixteen parses without a template
    assert seen[18] > ntiles // 2                                # (tried again: every tile holds a short record)
    # a regular input in between is parsed right as well, and takes the template up again once the rest is over
    reg = port.parse_xml(data, schema.ZHOBTMIND)
    for k in range(20):
        got = p.parse_host(data).to_host()
    assert_batches_equal(got, reg, "regular input afterwards")
    assert p.timing().lookup_tiles <= 2
    p.close()
