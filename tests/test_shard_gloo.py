"""The N>1 path on CPU: two gloo ranks shard files and byte ranges, parse their share with the
CPU checker (no GPU here) and gather row counts on the host, as bench.py does with the GPU
parser.  The concatenation of the shares must equal the parse of the whole."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import gen, schema, shard  # noqa: E402
from oracle import oracle_lib  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, nfiles, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    chk = oracle_lib.load("port")
    files = [f"obs_{i:03d}.xml" for i in range(nfiles)]
    mine = shard.assign_files(files, rank, world)
    rows = 0
    nbytes = 0
    for name in mine:
        i = int(name[4:7])
        data, n = gen.mind_xml(rows=200 + 37 * i, first_row=1000 * i)
        hb = chk.parse_xml(data, schema.ZHOBTMIND)
        assert hb.rows == n
        rows += hb.rows
        nbytes += len(data)
    cnt = torch.tensor([rows, nbytes, len(mine)], dtype=torch.int64)  # host tensors only
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(os.path.join(out_dir, "totals.npy"), np.array(cnt.tolist() + [float(t[0])]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_files_and_gather_counts(tmp_path):
    nfiles, world = 7, 2
    mp.spawn(_rank_main, args=(world, _free_port(), nfiles, str(tmp_path)), nprocs=world, join=True)
    tot = np.load(tmp_path / "totals.npy")
    exp_rows = sum(200 + 37 * i for i in range(nfiles))
    assert int(tot[0]) == exp_rows
    assert int(tot[2]) == nfiles
    assert tot[3] == pytest.approx(0.002)


def test_assign_files_is_a_partition():
    files = [f"f{i:02d}" for i in range(11)]
    for world in (1, 2, 3, 8, 16):
        parts = [shard.assign_files(files, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == sorted(files)
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        shard.assign_files(files, 2, 2)


def test_assign_by_size_balances():
    sizes = [100, 1, 1, 1, 50, 49, 2]
    parts = shard.assign_by_size(sizes, 2)
    assert sorted(sum(parts, [])) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 4


@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_split_points_give_the_same_rows(port, parts):
    data, n = gen.mind_xml(rows=3000)
    data = np.concatenate([data, np.frombuffer(b"<obtid>tail without terminator", dtype=np.uint8)])
    ranges = shard.split_points(data, parts)
    assert ranges[0][0] == 0 and ranges[-1][1] == len(data)
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    whole = port.parse_xml(data, schema.ZHOBTMIND)
    assert whole.rows == n
    row = 0
    for lo, hi in ranges:
        hb = port.parse_xml(np.ascontiguousarray(data[lo:hi]), schema.ZHOBTMIND)
        for c in range(whole.ncols):
            for r in (0, hb.rows // 2, hb.rows - 1):
                if hb.rows:
                    assert hb.cell(r, c) == whole.cell(row + r, c)
        row += hb.rows
    assert row == whole.rows


def test_split_points_edge_cases():
    assert shard.split_points(b"", 4) == [(0, 0)]
    assert shard.split_points(b"no terminator at all", 3) == [(0, 20)]
    one = b"<a>1</a><endl/>\n"
    assert shard.split_points(one, 5) == [(0, len(one))]
    lines = b"a,b\nc,d\ne,f\n"
    r = shard.split_points(lines, 3, endbz="")
    assert r[0][0] == 0 and r[-1][1] == len(lines) and all(lines[hi - 1:hi] == b"\n" for _, hi in r)
    # "<endl/>\r\n" is not a terminator (readline keeps '\r')
    crlf = b"<a>1</a><endl/>\r\n" * 10
    assert shard.split_points(crlf, 2) == [(0, len(crlf))]


def test_archive_shares_are_a_partition_in_sorted_order(tmp_path):
    from dsp_amd import archive
    for world in (1, 2, 4, 8):
        shares = [archive.rank_share(19, r, world) for r in range(world)]
        assert sorted(sum(shares, [])) == list(range(19))
        assert all(s == sorted(s) for s in shares)
    # file i is the same bytes whoever writes it, and the names sort in index order
    a = archive.make_files(str(tmp_path / "a"), [3, 0], 1 << 16, workers=1)
    b = archive.make_files(str(tmp_path / "b"), [0, 3], 1 << 16, workers=2)
    assert a[1] == b[1][::-1]
    assert open(a[0][0], "rb").read() == open(b[0][1], "rb").read()
    assert archive.file_name(2) < archive.file_name(10)
    gz = archive.make_files(str(tmp_path / "c"), [3], 1 << 16, gz=True, workers=1)
    import gzip
    assert gzip.decompress(open(gz[0][0], "rb").read()) == open(a[0][0], "rb").read()
