"""GPU parity on random schemas: names of every length, all typed conversions, carry on and off,
other end marks, over inputs from fully regular (every tile follows the tag template) to heavily
damaged (every tile takes the table lookup), several tiles each.  Bit-exact against the CPU
checker."""
import os
import random

import numpy as np
import pytest

import cases
from dsp_amd import capi, schema
from dsp_amd.schema import Col
from util import assert_batches_equal

# seeds of the random tests: 24 per run; MDC_FUZZ_SEEDS more of them, from MDC_FUZZ_FIRST on, for a longer hunt
SEEDS = range(int(os.environ.get("MDC_FUZZ_FIRST", "0")), int(os.environ.get("MDC_FUZZ_FIRST", "0")) + int(os.environ.get("MDC_FUZZ_SEEDS", "24")))

pytestmark = pytest.mark.gpu


def random_schema(rng, n=None):
    n = n or rng.choice([1, 2, 3, 5, 8, 11, 17, 30])
    names = set()
    while len(names) < n:
        k = rng.choice([1, 1, 2, 3, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31])
        stem = rng.choice(["", "a", "ab", "obt", "val", "x_"])
        names.add((stem + "".join(rng.choice("abcdxyz0189_") for _ in range(k)))[:31] or "q")
    cols, typed = [], []
    for nm in sorted(sorted(names), key=lambda _: rng.random()):  # (a set's order differs from process to process)
        dt = rng.choice(["char", "date", "number", "number"])
        cl = {"date": 14, "number": 22}.get(dt, rng.choice([0, 1, 5, 10, 30]))
        if rng.random() < 0.15:
            cl = rng.choice([0, 3, 40])
        cols.append(Col(nm, dt, cl))
        typed.append(rng.choice([-1, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8]))
    return cols, typed


@pytest.fixture
def utc():
    """Time columns: the checker's mktime runs in the process's zone, the parser's default is UTC."""
    import time
    old = os.environ.get("TZ")
    os.environ["TZ"] = "UTC0"
    time.tzset()
    yield
    if old is None:
        os.environ.pop("TZ", None)
    else:
        os.environ["TZ"] = old
    time.tzset()


def resolve(cols, typed):
    d = schema.default_typed(cols)
    return [d[i] if t < 0 else t for i, t in enumerate(typed)]


@pytest.mark.parametrize("seed", SEEDS)  # (more seeds for a longer hunt)
def test_random_schema(port, seed, utc):
    rng = random.Random(1000 + seed)
    cols, typed = random_schema(rng)
    ty = resolve(cols, typed)
    carry = rng.random() < 0.7
    endbz = rng.choice(["<endl/>", "<endl/>", "<end/>", "", "##", "<end-of-this-record-and-of-nothing-else/>",
                        "<e>" * 11])
    damage = rng.choice([0.0, 0.0, 0.02, 0.3, 0.8])
    nrec = rng.choice([40, 400, 3000])
    data = cases.mutated_xml(seed, nrec, cols, endbz=endbz, damage=damage)
    print("seed", seed, "ncols", len(cols), "carry", carry, "endbz", repr(endbz), "damage", damage, "bytes", len(data), flush=True)
    exp = port.parse_xml(data, cols, typed=ty, endbz=endbz, carry=carry)
    p = capi.Parser.xml(cols, typed=ty, endbz=endbz, carry=carry)
    try:
        got = p.parse_host(data).to_host()
        assert_batches_equal(got, exp, f"seed {seed}")
        # the same parser once more on other bytes of the same schema, then on the first again
        other = cases.mutated_xml(seed + 500, nrec // 2 + 1, cols, endbz=endbz, damage=min(1.0, damage * 2 + 0.01))
        assert_batches_equal(p.parse_host(other).to_host(), port.parse_xml(other, cols, typed=ty, endbz=endbz, carry=carry),
                             f"seed {seed} other")
        assert_batches_equal(p.parse_host(data).to_host(), exp, f"seed {seed} again")
    finally:
        p.close()


@pytest.mark.parametrize("seed", SEEDS[:12])
def test_random_wide_schema(port, seed, utc):
    """Schemas of more than 128 columns (one pointer table's worth): records of several KiB, so tiles of a few
    records, records across tiles and, where the cells are long, records beyond every look-behind."""
    rng = random.Random(7000 + seed)
    cols, typed = random_schema(rng, rng.choice([129, 160, 200, 255, 256]))
    ty = resolve(cols, typed)
    carry = rng.random() < 0.7
    endbz = rng.choice(["<endl/>", "<endl/>", "", "##", "<end\n/>", "##\n"])  # (marks with a newline that cannot overlap themselves)
    damage = rng.choice([0.0, 0.02, 0.3])
    nrec = rng.choice([12, 120, 600])
    data = cases.mutated_xml(seed, nrec, cols, endbz=endbz, damage=damage)
    print("seed", seed, "ncols", len(cols), "carry", carry, "endbz", repr(endbz), "damage", damage, "bytes", len(data), flush=True)
    exp = port.parse_xml(data, cols, typed=ty, endbz=endbz, carry=carry)
    p = capi.Parser.xml(cols, typed=ty, endbz=endbz, carry=carry)
    try:
        assert_batches_equal(p.parse_host(data).to_host(), exp, f"seed {seed}")
        assert_batches_equal(p.parse_host(data).to_host(), exp, f"seed {seed} again")
    finally:
        p.close()


def _marks_overlap(data: bytes, mark: bytes) -> bool:
    """Do two occurrences of `mark` + newline lie within one mark's length of each other?"""
    pat, el, ends, at = mark + b"\n", len(mark), [], 0
    while True:
        at = data.find(pat, at)
        if at < 0:
            break
        ends.append(at + el)  # the newline behind the mark
        at += 1
    return any(b - a <= el for a, b in zip(ends, ends[1:]))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_end_marks_with_newlines(port, seed, utc):
    """End marks that hold newlines over damaged records.  An input in which two occurrences of the mark overlap
    is refused (and must really hold such a pair); every other input equals the checker's parse."""
    rng = random.Random(9000 + seed)
    cols, typed = random_schema(rng, rng.choice([1, 3, 5, 11]))
    ty = resolve(cols, typed)
    carry = rng.random() < 0.7
    endbz = rng.choice(["<end\n/>", "##\n", "\n<e/>", "a\nb\nc", "\n", "<endl/>\n\n", "q\nq"])
    damage = rng.choice([0.0, 0.02, 0.3])
    nrec = rng.choice([40, 400, 3000])
    data = cases.mutated_xml(seed, nrec, cols, endbz=endbz, damage=damage)
    print("seed", seed, "ncols", len(cols), "carry", carry, "endbz", repr(endbz), "damage", damage, "bytes", len(data), flush=True)
    p = capi.Parser.xml(cols, typed=ty, endbz=endbz, carry=carry)
    try:
        try:
            got = p.parse_host(data).to_host()
        except capi.MdcError as e:
            assert e.code == capi.MDC_ERR_ARG and "overlap" in str(e), e
            assert _marks_overlap(bytes(data), endbz.encode()), "refused without an overlapping pair"
            return
        assert not _marks_overlap(bytes(data), endbz.encode()), "an overlapping pair went unnoticed"
        assert_batches_equal(got, port.parse_xml(data, cols, typed=ty, endbz=endbz, carry=carry), f"seed {seed}")
    finally:
        p.close()


@pytest.mark.parametrize("delta", [-17, -1, 0, 1, 16, 4095])
def test_input_sizes_around_tile_multiples(port, delta):
    """Inputs that end just before, at and just behind a tile boundary (16 KiB), terminated or not."""
    rec = b"<a>12</a><ab>x</ab><b>3.5</b><endl/>\n"
    for tiles in (1, 2, 5):
        n = tiles * 16384 + delta
        data = (rec * (n // len(rec) + 1))[:n]
        for tail in (b"", b"\n", b"<endl/>\n"):
            d = data + tail
            p = capi.Parser.xml(cases.SMALL)
            got = p.parse_host(d).to_host()
            p.close()
            assert_batches_equal(got, port.parse_xml(d, cases.SMALL), f"tiles {tiles} delta {delta} tail {tail!r}")


@pytest.mark.parametrize("first", [100, 3900, 4050, 4090, 4096, 4200, 9000])
def test_sample_record_around_the_template_window(port, first):
    """The template comes from the second record when it ends within the first 4 KiB: first records
    of growing length move it up to, across and beyond that window."""
    head = b"<a>1</a><ab>" + b"y" * max(0, first - 40) + b"</ab><b>2</b><endl/>\n"
    body = b"".join(b"<a>%d</a><ab>v%d</ab><b>%d.25</b><endl/>\n" % (i, i, i) for i in range(3000))
    data = head + body
    p = capi.Parser.xml(cases.SMALL)
    got = p.parse_host(data).to_host()
    p.close()
    assert_batches_equal(got, port.parse_xml(data, cases.SMALL), f"first {first}")


def random_lines(rng, nlines, sep: bytes, nfields, regular):
    """Delimited lines: `regular` of them well-formed numbers of the usual lengths (waves of plain
    cells), the others from fuzz_csv's mixture."""
    odd = [b"", b" ", b"  x  ", b"1e5", b"TRUE", b" true", b"a b", b" b ", b"99999999999999999999", b"4000000000",
           b"\r", b"x\x00y", b"1.2.3", b"-", b"+", b".", b"-.5", b"12345678", b"123456789", b"-1234567", b"1234567.", b"0000000.1"]
    out = bytearray()
    for _ in range(nlines):
        if rng.random() < regular:
            fields = []
            for _ in range(nfields):
                k = rng.random()
                if k < 0.4:
                    fields.append(b"%d" % rng.randint(-999, 99999))
                elif k < 0.8:
                    fields.append(b"%.*f" % (rng.randint(1, 3), rng.uniform(-50, 1100)))
                elif k < 0.9:
                    fields.append(b"2024%02d%02d%02d%02d%02d" % (rng.randint(1, 12), rng.randint(1, 28), rng.randint(0, 23),
                                                                 rng.randint(0, 59), rng.randint(0, 59)))
                else:
                    fields.append(b"%d" % rng.randint(0, 10 ** rng.randint(1, 17)))
        else:
            k = rng.choice([0, 1, nfields - 1, nfields, nfields, nfields + 1, nfields + 3])
            fields = [rng.choice(odd) if rng.random() < 0.5 else bytes(rng.choice(b"0123456789.-+ abT") for _ in range(rng.randrange(0, 20)))
                      for _ in range(k)]
        if rng.random() < 0.02 and fields:  # now and then a long field: lines that span tiles
            fields[rng.randrange(len(fields))] = bytes(rng.choice(b"0123456789.- abc\x00") for _ in range(rng.randrange(100, 6000)))
        out += sep.join(fields) + b"\n"
    if rng.random() < 0.5:
        out += b"unterminated" + sep + b"tail"
    return bytes(out)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_delimited(port, seed, utc):
    """Random separators (up to four bytes: the parallel split, unless two matches overlap in a tile, as
    they do for "aa" and "b b" in these lines; longer: the walk), field counts, clips, conversions and
    trimming over lines from all well-formed to all odd."""
    rng = random.Random(7000 + seed)
    sep = rng.choice([",", ",", ";", "|", " ", "\t", "~!~", "::", "ab", "aa", "b b", "-.-.", "0.0.0",
                      "<-sep->", "0123456789abcdefg", "=" * 16 + ">", "long separator, longer than the header holds"])
    nfields = rng.choice([1, 2, 3, 6, 10, 10, 17, 40])
    ilen = [rng.choice([0, 0, 0, 1, 3, 8, 9, 16, 17, 30]) for _ in range(nfields)]
    typed = [rng.choice([0, 0, 1, 2, 3, 4, 5, 5, 6, 7, 8]) for _ in range(nfields)]
    bdel = rng.random() < 0.4
    regular = rng.choice([1.0, 1.0, 0.98, 0.7, 0.0])
    nlines = rng.choice([50, 500, 5000])
    data = random_lines(rng, nlines, sep.encode(), nfields, regular)
    print("seed", seed, "sep", repr(sep), "fields", nfields, "bdelspace", bdel, "regular", regular, "bytes", len(data), flush=True)
    exp = port.parse_csv(data, sep, nfields, ilen=ilen, typed=typed, bdelspace=bdel)
    p = capi.Parser.csv(sep, nfields, ilen=ilen, typed=typed, bdelspace=bdel)
    try:
        got = p.parse_host(np.frombuffer(data, dtype=np.uint8)).to_host()
        assert_batches_equal(got, exp, f"seed {seed}")
    finally:
        p.close()


def binary32_traps(n, seed=5):
    """Decimals of at most 16 bytes whose correctly rounded binary64 value lies exactly half way
    between two binary32 values although the decimal does not: rounding twice gives the wrong
    binary32 for about a third of them."""
    import struct
    from fractions import Fraction
    f32 = lambda x: struct.unpack("<f", struct.pack("<f", x))[0]
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice([13, 14])
        v = f32(10 ** (rng.uniform(14, 14.99) - k))
        b = struct.unpack("<I", struct.pack("<f", v))[0]
        v2 = struct.unpack("<f", struct.pack("<I", b + 1))[0]
        mid = Fraction(v) + (Fraction(v2) - Fraction(v)) / 2
        m = round(mid * 10 ** k)
        if m >= 10 ** 15 or Fraction(m, 10 ** k) == mid or Fraction(m / 10 ** k) != mid:
            continue
        s = str(m).rjust(k + 1, "0")
        out.append((s[:-k] + "." + s[-k:]).encode())
    return out


def test_plain_binary32_cells(port):
    """Whole waves of short plain numbers into binary32 columns (the write pass's straight-line
    conversion), among them exact ties, the double-rounding traps and signed zeros."""
    rng = random.Random(77)
    traps = binary32_traps(400)
    assert all(len(t) <= 16 for t in traps)
    fixed = [b"16777217", b"16777219", b"33554434", b"-16777217", b"0.1", b"12.7", b"-0", b"-0.0", b"+.5", b"5.", b"0", b".",
             b"-", b"", b"340282350000000", b"0.00000000000001", b"8388608.5", b"8388609.5", b"4194304.25", b"4194304.75"]
    cols = [Col("a", "number", 22), Col("b", "number", 22), Col("c", "char", 16), Col("d", "date", 14)]
    typed = [schema.T_F32] * 4
    recs = []
    for i in range(30000):
        vals = []
        for _ in range(4):
            k = rng.random()
            if (i // 200) % 5:  # four blocks of five without traps: their waves stay on the straight-line path
                k = 0.06 + 0.94 * k
            if k < 0.03:
                vals.append(rng.choice(traps))
            elif k < 0.06:
                vals.append(rng.choice(fixed))
            elif k < 0.6:
                vals.append(b"%.*f" % (rng.randint(0, 6), rng.uniform(-2000, 2000)))
            else:
                vals.append(b"%d.%0*d" % (rng.randint(0, 10 ** rng.randint(1, 8)), rng.randint(1, 6), rng.randint(0, 999999) % 10 ** 6))
        vals = [v[:16] for v in vals]
        recs.append(b"<a>%s</a><b>%s</b><c>%s</c><d>%s</d><endl/>\n" % tuple(vals))
    data = np.frombuffer(b"".join(recs), dtype=np.uint8)
    exp = port.parse_xml(data, cols, typed=typed)
    p = capi.Parser.xml(cols, typed=typed)
    try:
        assert_batches_equal(p.parse_host(data).to_host(), exp, "binary32")
    finally:
        p.close()
    assert int(np.count_nonzero(exp.valid[:, 0] & 1)) > 29000


@pytest.mark.parametrize("seed", SEEDS)
def test_random_long_records(port, seed, utc):
    """Records of a few hundred bytes to some hundred KiB (every one spans tiles, the larger look-behind
    levels and the long-record kernel come into play), long cells with tag-like text, NUL bytes and
    digits inside, clips at 4000 bytes or none."""
    rng = random.Random(9000 + seed)
    ncols = rng.choice([2, 3, 5, 8])
    cols, typed = [], []
    for i in range(ncols):
        dt = rng.choice(["char", "char", "number", "date"])
        cl = {"date": 14, "number": 22}.get(dt, rng.choice([0, 10, 4000]))
        cols.append(Col("f%d%s" % (i, "x" * rng.randrange(0, 9)), dt, cl))
        typed.append(rng.choice([0, 0, 1, 3, 5, 6, 8]))
    carry = rng.random() < 0.7
    pool = b"abcXYZ 0123456789.-+<>/\x00\n\r" + b"".join(b"<" + c.colname.encode() + b">" for c in cols[:2])
    big = rng.choice([300, 2000, 6000, 30000])
    recs = []
    for _ in range(rng.choice([30, 150])):
        out = bytearray()
        for c in cols:
            if rng.random() < 0.1:
                continue  # no such field in this record
            n = rng.randrange(0, 20) if rng.random() < 0.7 else rng.randrange(0, big)
            v = bytes(rng.choice(pool) for _ in range(n)) if rng.random() < 0.5 else (b"%d" % rng.randrange(10 ** 15))[: max(n, 1)] * (n // 15 + 1)
            v = v.replace(b"<endl/>\n", b"")
            nm = c.colname.encode()
            out += b"<" + nm + b">" + v + b"</" + nm + b">"
        recs.append(bytes(out) + b"<endl/>\n")
    data = np.frombuffer(b"".join(recs), dtype=np.uint8)
    print("seed", seed, "ncols", ncols, "carry", carry, "records", len(recs), "bytes", data.size, "longest", max(map(len, recs)), flush=True)
    exp = port.parse_xml(data, cols, typed=typed, carry=carry)
    p = capi.Parser.xml(cols, typed=typed, carry=carry)
    try:
        assert_batches_equal(p.parse_host(data).to_host(), exp, f"seed {seed}")
        assert_batches_equal(p.parse_host(data).to_host(), exp, f"seed {seed} again")
    finally:
        p.close()


@pytest.mark.parametrize("seed", SEEDS)
def test_random_format(port, seed):
    """The writer side on random columns: 1 to 40 fields, names of 1 to 40 bytes, values from empty to
    a few thousand bytes with NUL bytes now and then, cuts or none, files of any number of rows."""
    import test_gpu_format as tgf
    rng = random.Random(11000 + seed)
    nf = rng.choice([1, 2, 3, 7, 11, 24, 40])
    nrows = rng.choice([0, 1, 2, 63, 64, 65, 500, 3000])
    maxlen = rng.choice([0, 3, 12, 40, 300])
    rows = cases.random_rows(12000 + seed, nrows, nf, maxlen=maxlen, nul=rng.choice([0, 0, 0.01, 0.2]),
                             big=rng.choice([0, 0, 0.002]) if maxlen else 0)
    names = ["".join(rng.choice("abcxyz_019") for _ in range(rng.choice([1, 2, 5, 9, 17, 40]))) for _ in range(nf)]
    fl = None if rng.random() < 0.5 else [rng.choice([0, 0, 1, 4, 10, 100]) for _ in range(nf)]
    for mc in {0, rng.choice([1, 7, 64, 1000])}:
        tgf.check_against(port, names, fl, rows, mc, f"seed {seed} maxcount {mc}")
