"""Diagnostic: per-phase cycle shares of the tile kernel (needs libmdc_parse_diag.so, built by
`python -c "from dsp_amd import build; build.build_diag()"`).  Shares, not absolute times."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_amd import capi, gen, schema  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "dsp_amd", "libmdc_parse_diag.so")
L = capi.lib()
L.mdc_debug_phase_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
workload = sys.argv[1] if len(sys.argv) > 1 else "xml"
nbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 256 << 20
if workload == "xml":
    data, rows = gen.mind_xml(nbytes=nbytes)
    p = capi.Parser.xml(schema.ZHOBTMIND)
else:
    data, rows = gen.mind_csv(nbytes=nbytes)
    p = capi.Parser.csv(",", 10, typed=schema.MIND_CSV_TYPED)
d = capi.DeviceBuffer(data)
buf = (ctypes.c_ulonglong * 24)()
p.parse_device(d.ptr, d.n).free()
L.mdc_debug_phase_counters(p.h, buf, 24, 1)
p.parse_device(d.ptr, d.n).free()
L.mdc_debug_phase_counters(p.h, buf, 24, 1)
names = ["load+lookbehind", "classify+rank+s0", "clears+recend", "tag matching", "delimited desc", "cells (columns)",
         "event list", "-", "-", "row outputs"]
t = p.timing()
print(f"rows={rows} count_ms={t.count_ms:.3f} write_ms={t.write_ms:.3f}")
for label, base in (("count pass", 0), ("write pass", 12)):
    tot = sum(buf[base + i] for i in range(10)) or 1
    print(label, "cycles/tile =", tot // max(t.tiles, 1))
    for i, n in enumerate(names):
        print(f"  {n:20s} {buf[base + i]:>16d}  {100.0 * buf[base + i] / tot:5.1f}%")
