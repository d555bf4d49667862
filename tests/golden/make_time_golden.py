"""Generates tests/golden/time_cases.json: the reference's strtotime (public/_public.cpp:1009-1044,
through oracle/_ref) on the probe texts of tests/test_time_columns.py, in each of its time zones.
Runs only where oracle/_ref/libmdc_ref.so exists.  Data only: texts and the values returned.

    python tests/golden/make_time_golden.py
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_time_columns as tc  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402


def main():
    ref = ol.load("reference")
    out = []
    for tz, off in tc.STRTOTIME_ZONES:
        os.environ["TZ"] = tz
        time.tzset()
        rows = []
        for t in tc.PROBES + tc.random_times(17, 300):
            ok, v = ref.convert(t, ol.T_TIME)
            rows.append([t.hex(), ("%016x" % v) if ok else None])
        out.append({"tz": tz, "tz_offset": off, "cases": rows})
    fn = os.path.join(HERE, "time_cases.json")
    with open(fn, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print(fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
