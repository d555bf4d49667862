"""Diagnostic: one seed of tests/test_gpu_fuzz.py::test_random_schema parsed over and over, counting
the parses whose sized and written byte counts differ."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
import test_gpu_fuzz as tf
from dsp_amd import capi

seed, reps = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(1000 + seed)
cols, typed = tf.random_schema(rng)
ty = tf.resolve(cols, typed)
carry = rng.random() < 0.7
endbz = rng.choice(["<endl/>", "<endl/>", "<end/>", "", "##"])
damage = rng.choice([0.0, 0.0, 0.02, 0.3, 0.8])
nrec = rng.choice([40, 400, 3000])
data = cases.mutated_xml(seed, nrec, cols, endbz=endbz, damage=damage)
other = cases.mutated_xml(seed + 500, nrec // 2 + 1, cols, endbz=endbz, damage=min(1.0, damage * 2 + 0.01))
bad = {}
order = sys.argv[3].split(",") if len(sys.argv) > 3 else ["data", "other"]
for it in range(reps):
    p = capi.Parser.xml(cols, typed=ty, endbz=endbz, carry=carry)
    for name in order:
        d = data if name == "data" else other
        b = p.parse_host(d)
        for c in range(b.ncols):
            off, dat = b.column(c)
            if len(dat) != int(off[b.rows]):
                key = (name, c, len(dat), int(off[b.rows]))
                bad[key] = bad.get(key, 0) + 1
        b.free()
    p.close()
if bad:
    import pickle
    out = os.path.join(ROOT, "gpurun_out", "bad_case.pkl")
    pickle.dump({"cols": [(c.colname, c.datatype, c.collen) for c in cols], "ty": ty, "carry": carry, "endbz": endbz,
                 "data": bytes(data), "other": bytes(other), "bad": bad}, open(out, "wb"))
print("order", order, "reps", reps, "bad", bad)
print("col info", {k: (cols[k[1]], ty[k[1]]) for k in bad})
