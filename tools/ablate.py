"""Diagnostic: kernel time with phases switched off (libmdc_parse_ablate.so, MDC_DEBUG_SKIP bits:
1 = tag events, 2 = cells, 4 = sizing pass ends behind classification and scan, 8 = behind its loads, 16 = emit kernel without its columns, 32 = emit kernel without the columns' stores, 64 = delimited sizing pass without the one-lane walk of a record (MDC_ABLATE_CSV=1 parses the delimited batch), 128 = tagged sizing pass without wave 0's compare of the first record's tags in the look-behind; one-pass parse (MDC_ONE_PASS=1 in the environment): 1024 = no waiting for the columns' places, 2048 = none for the rows'; MDC_NO_MAP=1 sends every tile through the full write kernel).  Outputs are wrong by construction; only the times matter."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("MDC_ABLATE_CHILD"):
    sys.path.insert(0, ROOT)
    from dsp_amd import capi, gen, schema
    capi.LIB_PATH = os.path.join(ROOT, "dsp_amd", "libmdc_parse_ablate.so")
    if os.environ.get("MDC_ABLATE_CSV"):  # (the delimited batch instead)
        data, rows = gen.mind_csv(nbytes=1 << 30)
        p = capi.Parser.csv(",", 10, typed=schema.MIND_CSV_TYPED)
    else:
        data, rows = gen.mind_xml(nbytes=1 << 30)
        p = capi.Parser.xml(schema.ZHOBTMIND)
    d = capi.DeviceBuffer(data)
    best = None
    for _ in range(6):
        try:
            p.parse_device(d.ptr, d.n).free()
        except capi.MdcError as e:
            print("  (parse error expected in ablation:", e.code, ")")
        t = p.timing()
        best = (t.count_ms, t.write_ms) if best is None else (min(best[0], t.count_ms), min(best[1], t.write_ms))
    print(f"  skip={os.environ.get('MDC_DEBUG_SKIP', '0')}: count {best[0]:.3f} ms  write {best[1]:.3f} ms")
else:
    for skip in sys.argv[1:] or ("0", "1", "2", "3"):
        env = dict(os.environ, MDC_DEBUG_SKIP=skip, MDC_ABLATE_CHILD="1")
        subprocess.run([sys.executable, __file__], env=env, check=False)
