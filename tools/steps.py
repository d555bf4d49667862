"""Diagnostic: per-parse timings of consecutive parses of one batch.
usage: steps.py [xml|csv] [device|host] [bytes] [parses]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_amd import capi, gen, schema

if os.environ.get("MDC_LIB"):  # another build of the library, for side-by-side timings
    capi.LIB_PATH = os.path.join(ROOT, "dsp_amd", os.environ["MDC_LIB"])

wl = sys.argv[1] if len(sys.argv) > 1 else "xml"
src = sys.argv[2] if len(sys.argv) > 2 else "device"
nbytes = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
parses = int(sys.argv[4]) if len(sys.argv) > 4 else 6
if wl == "xml":
    data, rows = gen.mind_xml(nbytes=nbytes)
    p = capi.Parser.xml(schema.ZHOBTMIND)
else:
    data, rows = gen.mind_csv(nbytes=nbytes)
    p = capi.Parser.csv(",", 10, typed=schema.MIND_CSV_TYPED)
d = capi.DeviceBuffer(data) if src == "device" else None
best = None
for i in range(parses):
    t0 = time.perf_counter()
    try:
        b = p.parse_device(d.ptr, d.n) if d else p.parse_host(data)
    except capi.MdcError as e:  # (ablated diagnostic builds leave inconsistent outputs)
        if not os.environ.get("MDC_DEBUG_SKIP"):
            raise
        print("  parse error expected in ablation:", e.code)
        continue
    wall = time.perf_counter() - t0
    t = p.timing()
    print(f"{wl}/{src} parse {i}: count {t.count_ms:.3f} scan {t.scan_ms:.3f} write {t.write_ms:.3f} "
          f"device total {t.total_ms:.3f} ms, wall {wall * 1e3:.3f} ms = {len(data) / wall / 1e9:.2f} GB/s, "
          f"rows {t.rows}, out {t.bytes_out}", flush=True)
    b.free()
    best = (t.count_ms, t.write_ms, t.total_ms) if best is None else tuple(min(a, x) for a, x in zip(best, (t.count_ms, t.write_ms, t.total_ms)))
print(f"{wl}/{src} best of {parses}: count {best[0]:.3f} write {best[1]:.3f} total {best[2]:.3f} ms = {len(data) / best[2] / 1e6:.1f} GB/s (device time)")
