"""One worker of bench.py's all-cores CPU baseline: parses its own batch of synthetic records with
the CPU checker (the reference's code when oracle/_ref is there) and prints bytes, records and
seconds as JSON.  Never touches the GPU.
usage: cpu_slice.py xml|csv <bytes> <first_row>"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_amd import gen, schema  # noqa: E402  (the generator and the schemas: no HIP call)
from oracle import oracle_lib  # noqa: E402

wl, nbytes, first = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kind = "reference" if oracle_lib.available("reference") else "port"
chk = oracle_lib.load(kind)
data, _ = (gen.mind_xml if wl == "xml" else gen.mind_csv)(nbytes=nbytes, first_row=first)
t0 = time.perf_counter()
hb = chk.parse_xml(data, schema.ZHOBTMIND) if wl == "xml" else chk.parse_csv(data, ",", 10, typed=schema.MIND_CSV_TYPED)
dt = time.perf_counter() - t0
print(json.dumps({"bytes": int(len(data)), "rows": int(hb.rows), "seconds": dt, "kind": kind}))
