"""Work partitioning for several GPUs: files shard independently, nothing is exchanged.

The reference treats one file as the unit of work, transaction and recovery
(tools/xmltodb.cpp:163-218, :306) and walks a sorted file list (tools/xmltodb.cpp:147).  With G
ranks (one process per GPU) rank r takes files r, r+G, ... of that list.  One huge file is cut
at record terminators, so every part is a complete input of its own: the terminator is a '\\n'
preceded by endbz (cifile::readline, public/_public.cpp:866-892), the first part keeps whatever
precedes the first record and the bytes behind the last terminator stay with the last part
(where the parser drops them, as the reference does).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def assign_files(files: Sequence[str], rank: int, world: int) -> List[str]:
    """Files of this rank: round-robin over the sorted list."""
    if not 0 <= rank < world:
        raise ValueError("rank outside the world")
    return sorted(files)[rank::world]


def assign_by_size(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Size-balanced alternative: largest file first onto the least loaded rank.  Returns, per
    rank, the indices into `sizes` in ascending order."""
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(sizes)), key=lambda k: (-sizes[k], k)):
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += sizes[i]
    return [sorted(x) for x in out]


def _is_terminator(buf: np.ndarray, nl: int, mark: bytes) -> bool:
    m = len(mark)
    if nl < m:
        return m == 0
    return m == 0 or buf[nl - m:nl].tobytes() == mark


def split_points(buf, parts: int, endbz: str | bytes = "<endl/>") -> List[Tuple[int, int]]:
    """Cuts `buf` into at most `parts` byte ranges [lo, hi) that end behind a record terminator
    (the last one ends at len(buf)).  Ranges are contiguous and cover the buffer; fewer ranges
    come back when there are not enough terminators."""
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    mark = endbz.encode() if isinstance(endbz, str) else bytes(endbz)
    n = int(a.size)
    if parts < 1:
        raise ValueError("parts must be positive")
    cuts = [0]
    for k in range(1, parts):
        want = max(n * k // parts, cuts[-1])
        pos = want
        cut = None
        while pos < n:
            # next '\n' at or behind `pos`, looked for in bounded windows
            win = a[pos:pos + (1 << 16)]
            hits = np.flatnonzero(win == 10)
            found = False
            for h in hits:
                nl = pos + int(h)
                if _is_terminator(a, nl, mark):
                    cut = nl + 1
                    found = True
                    break
            if found:
                break
            pos += win.size
        if cut is None or cut >= n:
            break
        if cut > cuts[-1]:
            cuts.append(cut)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
