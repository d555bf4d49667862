"""A file-sharded archive of observation files through the GPU parser: the loop of the reference's
ingest (tools/xmltodb.cpp:147-218: sorted directory listing, then file after file: open, read,
parse, count) with the files of the sorted list dealt round-robin to the ranks (one process per
GPU, `shard.assign_files`) and every rank's share read, uploaded and parsed through
mdc_stream_* (host threads read or inflate ahead into pinned buffers, the upload of file k+1 runs
on a copy stream beside the parse of file k).  Nothing is exchanged between ranks; row and byte
counts are summed on the host by the caller.

Files are named as the reference's producer names them (tools/dminingoracle.cpp:204-205:
ZHOBTMIND_<timestamp>_<efilename>_<seq>.xml), so the sorted order is the order of <seq> padded to
a fixed width here.
"""
from __future__ import annotations

import gzip
import os
import time
from concurrent.futures import ProcessPoolExecutor
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from . import gen, shard

ROWS_PER_FILE_STRIDE = 1_000_000  # first keyid of file i = i * this: every file's rows differ


def file_name(i: int, gz: bool = False) -> str:
    return f"ZHOBTMIND_20240519162835_togxpt_{i:05d}.xml" + (".gz" if gz else "")


def _make_one(args):
    directory, i, nbytes, gz = args
    data, rows = gen.mind_xml(nbytes=nbytes, first_row=i * ROWS_PER_FILE_STRIDE)
    raw = data.tobytes()
    path = os.path.join(directory, file_name(i, gz))
    with open(path, "wb") as f:
        f.write(gzip.compress(raw, 6) if gz else raw)  # gzip -6: what tools/gzipfiles.cpp:59 leaves
    return path, rows, len(raw)


def make_files(directory: str, indices: Sequence[int], file_bytes: int, gz: bool = False, workers: int = 8):
    """Writes file i of the archive for every i in `indices` (deterministic: file i is the same
    bytes whoever writes it).  Returns parallel lists (paths, rows, uncompressed bytes)."""
    os.makedirs(directory, exist_ok=True)
    jobs = [(directory, i, file_bytes, gz) for i in indices]
    if workers > 1 and len(jobs) > 1:
        with ProcessPoolExecutor(max_workers=min(workers, len(jobs))) as ex:
            made = list(ex.map(_make_one, jobs))
    else:
        made = [_make_one(j) for j in jobs]
    return [m[0] for m in made], [m[1] for m in made], [m[2] for m in made]


@dataclass
class ShareResult:
    files: int = 0
    rows: int = 0
    bytes: int = 0            # uncompressed record bytes handed to the parser
    seconds: float = 0.0
    rows_per_file: List[int] = field(default_factory=list)
    errors: List[str] = field(default_factory=list)


def run_share(parser, paths: Sequence[str], threads: int = 4, keep: Optional[list] = None) -> ShareResult:
    """This rank's files, in list order, through the parser's file stream.  `keep`, if given,
    receives (index, Batch) of every file instead of the batches being freed."""
    res = ShareResult(files=len(paths))
    t0 = time.perf_counter()
    for idx, b in parser.stream(list(paths), threads=threads):
        if isinstance(b, Exception):
            res.errors.append(f"{paths[idx]}: {b}")
            res.rows_per_file.append(-1)
            continue
        res.rows += b.rows
        res.rows_per_file.append(b.rows)
        if keep is not None:
            keep.append((idx, b))
        else:
            b.free()
    res.seconds = time.perf_counter() - t0
    return res


def rank_share(nfiles: int, rank: int, world: int, gz: bool = False) -> List[int]:
    """Indices of the archive's files that rank `rank` of `world` parses (round-robin over the
    sorted list, as shard.assign_files deals them)."""
    names = [file_name(i, gz) for i in range(nfiles)]
    mine = set(shard.assign_files(names, rank, world))
    return [i for i in range(nfiles) if names[i] in mine]
