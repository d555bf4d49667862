"""Helpers shared by the parity tests."""
import numpy as np


def assert_batches_equal(a, b, what=""):
    """Bit-exact comparison of two columnar batches (HostBatch-like objects)."""
    assert a.rows == b.rows, f"{what}: rows {a.rows} != {b.rows}"
    assert a.ncols == b.ncols, f"{what}: ncols"
    assert np.array_equal(a.rec_off, b.rec_off), f"{what}: rec_off differs"
    for c in range(a.ncols):
        if not np.array_equal(a.off[c], b.off[c]):
            r = int(np.nonzero(a.off[c] != b.off[c])[0][0])
            raise AssertionError(f"{what}: col {c} offsets differ first at row {r - 1}: "
                                 f"{a.cell(max(r - 1, 0), c)!r} vs {b.cell(max(r - 1, 0), c)!r}")
        if len(a.data[c]) != len(b.data[c]):
            raise AssertionError(f"{what}: col {c} holds {len(a.data[c])} bytes, expected {len(b.data[c])}; the offsets end at "
                                 f"{int(a.off[c][a.rows])}; last cells {[a.cell(r, c) for r in range(max(0, a.rows - 3), a.rows)]!r}")
        if not np.array_equal(a.data[c], b.data[c]):
            i = int(np.nonzero(a.data[c] != b.data[c])[0][0])
            r = int(np.searchsorted(a.off[c], i, side="right")) - 1
            raise AssertionError(f"{what}: col {c} bytes differ in row {r}: {a.cell(r, c)!r} vs {b.cell(r, c)!r}")
    assert np.array_equal(a.found, b.found), f"{what}: found masks differ"
    if not np.array_equal(a.valid, b.valid):
        r = int(np.nonzero((a.valid != b.valid).any(axis=1))[0][0])
        raise AssertionError(f"{what}: valid masks differ at row {r}: {a.valid[r]} vs {b.valid[r]} "
                             f"cells {[a.cell(r, c) for c in range(a.ncols)]}")
    for c in range(a.ncols):
        va, vb = a.val[c], b.val[c]
        assert (va is None) == (vb is None), f"{what}: col {c} typed presence"
        if va is not None and not np.array_equal(va, vb):
            r = int(np.nonzero(va != vb)[0][0])
            raise AssertionError(f"{what}: col {c} typed value differs at row {r}: cell {a.cell(r, c)!r} "
                                 f"{int(va[r]):#x} vs {int(vb[r]):#x}")
