"""Times the writer side on the GPU: parse a ZHOBTMIND buffer, then format the batch back to text.

    python tools/format_bench.py [bytes] [maxcount] [lds budget]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from dsp_amd import capi, gen, schema  # noqa: E402


def main():
    if os.environ.get("MDC_FORMAT_SKIP"):
        capi.LIB_PATH = os.path.join("dsp_amd", "libmdc_parse_ablate.so")
    nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
    maxcount = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    cols = schema.ZHOBTMIND
    data, rows = gen.mind_xml(nbytes=nbytes)
    buf = capi.DeviceBuffer(data)
    p = capi.Parser.xml(cols, flags=capi.MDC_F_NO_RECOFF)
    f = capi.Formatter([c.colname for c in cols])
    if len(sys.argv) > 3:
        os.environ["MDC_TEST_HOOKS"] = "1"
        capi.check(capi.lib().mdc_internal_format_budget(f.h, int(sys.argv[3])))
    b = p.parse_device(buf.ptr, buf.n)
    best = None
    for it in range(8):
        try:
            t = f.format_batch(b, maxcount)
        except capi.MdcError:
            if not os.environ.get("MDC_FORMAT_SKIP"):
                raise
            t = None  # (an ablated run may see NULs in what it did not write)

        tm = f.timing()
        if best is None or tm.format_ms < best[0]:
            best = (tm.format_ms, tm.total_ms, tm.bytes_in, tm.bytes_out)
        if t is not None:
            t.free()
    ms, tot, bi, bo = best
    print(f"rows {rows} text {bo / 2**30:.3f} GiB in {ms:.3f} ms (call {tot:.3f} ms): "
          f"{bo / ms / 1e6:.1f} GB/s of text, {(bi + bo) / ms / 1e6:.1f} GB/s read+written", flush=True)


if __name__ == "__main__":
    main()
