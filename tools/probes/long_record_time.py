"""Diagnostic: what one very long record costs.  A record that starts in front of every look-behind is parsed by
one block of the long-record kernel, every byte from global memory; the time grows with the record's length.
Shapes: one giant cell; a record dense with '<' (every one a tag candidate); one giant delimited line."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dsp_amd import capi, schema
from dsp_amd.schema import Col

COLS = [Col("a", "char", 0), Col("b", "number", 22), Col("c", "char", 0)]


def run(label, p, data):
    d = capi.DeviceBuffer(data)
    best = None
    for i in range(3):
        t0 = time.perf_counter()
        b = p.parse_device(d.ptr, d.n)
        w = time.perf_counter() - t0
        rows = b.rows
        b.free()
        best = w if best is None else min(best, w)
    t = p.timing()
    print(f"{label:52s} {best * 1e3:10.2f} ms  {data.size / best / 1e9:7.3f} GB/s  rows {rows}  long {t.long_records}", flush=True)
    d.free()
    p.close()


for mib in (1, 16, 128):
    n = mib << 20
    fill = np.full(n, ord("x"), dtype=np.uint8)
    fill[::97] = ord(" ")
    cell = b"<a>1</a><b>2.5</b><c>" + fill.tobytes() + b"</c><endl/>\n"
    run(f"xml, one cell of {mib} MiB", capi.Parser.xml(COLS), np.frombuffer(cell + b"<a>2</a><endl/>\n", dtype=np.uint8))
    dense = np.frombuffer((b"<q>1</q>" * (n // 8)), dtype=np.uint8)
    rec = b"<a>1</a>" + dense.tobytes() + b"<b>2.5</b><c>z</c><endl/>\n"
    run(f"xml, {mib} MiB of unknown tags in the record", capi.Parser.xml(COLS), np.frombuffer(rec + b"<a>2</a><endl/>\n", dtype=np.uint8))
    line = b"1,2.5," + fill.tobytes() + b"\n3,4,5\n"
    run(f"delimited, one field of {mib} MiB", capi.Parser.csv(",", 3), np.frombuffer(line, dtype=np.uint8))
    many = np.frombuffer(b"7," * (n // 2), dtype=np.uint8)
    run(f"delimited, {mib} MiB of fields beyond the schema's", capi.Parser.csv(",", 3), np.frombuffer(many.tobytes() + b"\n3,4,5\n", dtype=np.uint8))
