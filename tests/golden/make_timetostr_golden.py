"""Generates tests/golden/timetostr_cases.json: the reference's timetostr (public/_public.cpp:631-722,
through oracle/_ref) on the probe times of tests/test_timetostr.py, in every format and in four time
zones.  Runs only where oracle/_ref/libmdc_ref.so exists.  Data only: times and the texts returned.

    python tests/golden/make_timetostr_golden.py
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_timetostr as tt  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402


def main():
    ref = ol.load("reference")
    out = []
    formats = tt.FORMATS + tt.NOT_FORMATS
    for tz, off in tt.ZONES:
        os.environ["TZ"] = tz
        time.tzset()
        rows = []
        for t in tt.PROBES + tt.random_times(17, 150):
            texts = [ref.timetostr(t, f) for f in formats]
            rows.append([t, [None if x is None else x.decode() for x in texts]])
        out.append({"tz": tz, "tz_offset": off, "formats": formats, "cases": rows})
    fn = os.path.join(HERE, "timetostr_cases.json")
    with open(fn, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print(fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
