"""Diagnostic: parse times of the less regular configurations (multi-byte separators, records of
two shapes, NUL bytes) beside the benchmark's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dsp_amd import capi, gen, schema

def run(label, p, data):
    d = capi.DeviceBuffer(data)
    best = None
    for i in range(12):
        t0 = time.perf_counter()
        b = p.parse_device(d.ptr, d.n)
        w = time.perf_counter() - t0
        rows = b.rows
        b.free()
        if i >= 4:
            best = w if best is None else min(best, w)
    t = p.timing()
    print(f"{label:44s} {best * 1e3:8.3f} ms  {data.size / best / 1e9:7.1f} GB/s  rows {rows}  count {t.count_ms:.3f} write {t.write_ms:.3f}", flush=True)
    d.free(); p.close()

n = 256 << 20
data, _ = gen.mind_xml(nbytes=n)
run("xml", capi.Parser.xml(schema.ZHOBTMIND), data)
# records of two shapes: every third record without its <wd> field
txt = data.tobytes()
recs = txt.split(b"<endl/>\n")
import re
for i in range(0, len(recs) - 1, 3):
    recs[i] = re.sub(rb"<wd>[^<]*</wd>", b"", recs[i])
two = np.frombuffer(b"<endl/>\n".join(recs), dtype=np.uint8)
run("xml, every third record without <wd>", capi.Parser.xml(schema.ZHOBTMIND), two)
nul = data.copy(); nul[np.arange(1000, nul.size, 1 << 20)] = 0
run("xml, a NUL byte every MiB", capi.Parser.xml(schema.ZHOBTMIND), nul)
c1, _ = gen.mind_csv(nbytes=n)
run("csv ','", capi.Parser.csv(",", 10, typed=schema.MIND_CSV_TYPED), c1)
c3, _ = gen.mind_csv(nbytes=n, sep="~!~")
run("csv '~!~'", capi.Parser.csv("~!~", 10, typed=schema.MIND_CSV_TYPED), c3)
run("csv ',' bdelspace", capi.Parser.csv(",", 10, typed=schema.MIND_CSV_TYPED, bdelspace=True), c1)
