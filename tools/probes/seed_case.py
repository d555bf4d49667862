"""Diagnostic: the body of tests/test_gpu_fuzz.py::test_random_schema over a list of seeds in one
process, printing the columns whose sized, written and expected byte counts differ."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
import test_gpu_fuzz as tf
from dsp_amd import capi
from oracle import oracle_lib as ol

port = ol.load("port")
seeds = [s for s in range(int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) < 4 or sys.argv[3] in str(s)]
bad = 0
for seed in seeds:
    rng = random.Random(1000 + seed)
    cols, typed = tf.random_schema(rng)
    ty = tf.resolve(cols, typed)
    carry = rng.random() < 0.7
    endbz = rng.choice(["<endl/>", "<endl/>", "<end/>", "", "##"])
    damage = rng.choice([0.0, 0.0, 0.02, 0.3, 0.8])
    nrec = rng.choice([40, 400, 3000])
    data = cases.mutated_xml(seed, nrec, cols, endbz=endbz, damage=damage)
    other = cases.mutated_xml(seed + 500, nrec // 2 + 1, cols, endbz=endbz, damage=min(1.0, damage * 2 + 0.01))
    p = capi.Parser.xml(cols, typed=ty, endbz=endbz, carry=carry)
    for name, d in (("data", data), ("other", other), ("again", data)):
        got = p.parse_host(d).to_host()
        exp = port.parse_xml(d, cols, typed=ty, endbz=endbz, carry=carry)
        for c in range(got.ncols):
            w = int(got.off[c][got.rows])
            if len(got.data[c]) != w or w != len(exp.data[c]) or not np.array_equal(got.off[c], exp.off[c]):
                bad += 1
                print("seed", seed, name, "bytes", len(d), "rows", got.rows, "ncols", len(cols), "col", c, cols[c], "typed", ty[c],
                      "carry", carry, "endbz", repr(endbz), "sized", len(got.data[c]), "written", w, "expected", len(exp.data[c]),
                      "offsets equal", bool(np.array_equal(got.off[c], exp.off[c])), flush=True)
    p.close()
print("seeds", len(seeds), "mismatches", bad)
