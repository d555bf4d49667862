"""The file-sharded archive (BASELINE configuration 4 in small): a sorted list of files dealt
round-robin to 8 ranks, every rank's share through the file stream.  The one GPU here runs the
eight shares one after the other; what is checked does not depend on which GPU ran a share: every
file's row count against the generator's, one whole file per rank cell for cell against the CPU
checker, and the shares being a partition of the list in sorted order."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from dsp_amd import archive, capi, gen, schema
from util import assert_batches_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_shares_of_an_archive_on_one_gpu(tmp_path, port):
    nfiles, world, file_bytes = 32, 8, 32 << 20
    paths, rows_exp, nbytes = archive.make_files(str(tmp_path), range(nfiles), file_bytes, workers=1)
    assert sorted(paths) == paths
    seen = []
    total_rows = 0
    p = capi.Parser.xml(schema.ZHOBTMIND)
    for rank in range(world):
        mine = archive.rank_share(nfiles, rank, world)
        assert mine == list(range(rank, nfiles, world))
        keep = []
        res = archive.run_share(p, [paths[i] for i in mine], threads=3, keep=keep)
        assert not res.errors
        assert res.rows_per_file == [rows_exp[i] for i in mine], f"rank {rank}"
        assert [k for k, _ in keep] == list(range(len(mine)))  # batches come back in list order
        total_rows += res.rows
        seen += mine
        # one whole file of the share against the CPU checker
        k = rank % len(mine)
        data = np.fromfile(paths[mine[k]], dtype=np.uint8)
        assert_batches_equal(keep[k][1].to_host(), port.parse_xml(data, schema.ZHOBTMIND), f"file {mine[k]} of rank {rank}")
        # the key column's first cell of every file: the file is the one the list names
        for j, b in keep:
            assert int(b.cell(0, 10)) == 1_000_000 + mine[j] * archive.ROWS_PER_FILE_STRIDE
            b.free()
    assert sorted(seen) == list(range(nfiles))
    assert total_rows == sum(rows_exp)
    p.close()


def test_one_ranks_share_of_the_64_gib_archive_at_full_file_size(port):
    """Rank 0 of 8's share of the 64 GiB archive (512 files of 128 MiB: files 0, 8, 16, ...) at the files'
    real size, as many of its 64 files as /dev/shm holds (16 at least): every file's rows against the
    generator's, the batches in list order, one whole 128 MiB file cell for cell against the CPU checker."""
    import shutil
    import tempfile
    nfiles_total, world, file_bytes = 512, 8, 128 << 20
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    free = shutil.disk_usage(base).free
    share = archive.rank_share(nfiles_total, 0, world)
    assert share == list(range(0, nfiles_total, world)) and len(share) == 64
    take = 64 if free > (12 << 30) else 16
    if free < (3 << 30):
        pytest.skip("no room for 16 files of 128 MiB")
    d = tempfile.mkdtemp(prefix="mdc_share_", dir=base)
    try:
        mine = share[:take]
        paths, rows_exp, nbytes = archive.make_files(d, mine, file_bytes, workers=8)
        assert paths == sorted(paths) and all(b > (127 << 20) for b in nbytes)
        p = capi.Parser.xml(schema.ZHOBTMIND)
        keep = []
        res = archive.run_share(p, paths, threads=8, keep=keep)
        assert not res.errors
        assert res.rows_per_file == rows_exp and res.rows == sum(rows_exp)
        assert [k for k, _ in keep] == list(range(take))
        for j, b in keep:  # the key column's first cell names the file
            assert int(b.cell(0, 10)) == 1_000_000 + mine[j] * archive.ROWS_PER_FILE_STRIDE
        k = take // 2
        data = np.fromfile(paths[k], dtype=np.uint8)
        assert_batches_equal(keep[k][1].to_host(), port.parse_xml(data, schema.ZHOBTMIND), f"file {mine[k]} of rank 0")
        for _, b in keep:
            b.free()
        p.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_bench_starts_its_own_ranks(tmp_path):
    # the documented command, in small: no launcher around it, two ranks (sharing this box's GPU)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--bytes", str(32 << 20), "--archive-files-per-gpu", "2", "--archive-file-mb", "8", "--cpu-sample-mb", "0",
           "--delimited-gb", "0", "--archive-dir", str(tmp_path)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["bytes_per_gpu"] > 30 << 20
    assert out["archive"]["files"] == 4 and out["archive"]["rows_match_generator"]
    assert out["archive"]["GBps_end_to_end"] > 0 and len(out["archive"]["per_rank_GBps"]) == 2
    # every rank was narrowed to one GPU (both to this box's only one) and to cores of its own
    from dsp_amd import shard
    ranks = out["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["device"] == 0 for r in ranks)
    cpus = [set(shard.parse_cpulist(r["cpus"])) for r in ranks]
    assert cpus[0] and cpus[1] and not cpus[0] & cpus[1]
