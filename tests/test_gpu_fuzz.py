"""GPU parity on random schemas: names of every length, all typed conversions, carry on and off,
other end marks, over inputs from fully regular (every tile follows the tag template) to heavily
damaged (every tile takes the table lookup), several tiles each.  Bit-exact against the CPU
checker."""
import random

import numpy as np
import pytest

import cases
from dsp_amd import capi, schema
from dsp_amd.schema import Col
from util import assert_batches_equal

pytestmark = pytest.mark.gpu


def random_schema(rng):
    n = rng.choice([1, 2, 3, 5, 8, 11, 17, 30])
    names = set()
    while len(names) < n:
        k = rng.choice([1, 1, 2, 3, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31])
        stem = rng.choice(["", "a", "ab", "obt", "val", "x_"])
        names.add((stem + "".join(rng.choice("abcdxyz0189_") for _ in range(k)))[:31] or "q")
    cols, typed = [], []
    for nm in sorted(names, key=lambda _: rng.random()):
        dt = rng.choice(["char", "date", "number", "number"])
        cl = {"date": 14, "number": 22}.get(dt, rng.choice([0, 1, 5, 10, 30]))
        if rng.random() < 0.15:
            cl = rng.choice([0, 3, 40])
        cols.append(Col(nm, dt, cl))
        typed.append(rng.choice([-1, -1, 0, 1, 2, 3, 4, 5, 6, 7]))
    return cols, typed


def resolve(cols, typed):
    d = schema.default_typed(cols)
    return [d[i] if t < 0 else t for i, t in enumerate(typed)]


@pytest.mark.parametrize("seed", range(24))
def test_random_schema(port, seed):
    rng = random.Random(1000 + seed)
    cols, typed = random_schema(rng)
    ty = resolve(cols, typed)
    carry = rng.random() < 0.7
    endbz = rng.choice(["<endl/>", "<endl/>", "<end/>", "", "##"])
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
