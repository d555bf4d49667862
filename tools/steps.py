"""Diagnostic: per-parse device timings of consecutive parses of one 1 GiB batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_amd import capi, gen, schema

data, rows = gen.mind_xml(nbytes=int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30)
p = capi.Parser.xml(schema.ZHOBTMIND)
d = capi.DeviceBuffer(data)
for i in range(8):
    b = p.parse_device(d.ptr, d.n)
    t = p.timing()
    print(f"parse {i}: count {t.count_ms:.3f} scan {t.scan_ms:.3f} write {t.write_ms:.3f} "
          f"total {t.total_ms:.3f} rows {t.rows}", flush=True)
    b.free()
