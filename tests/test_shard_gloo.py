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


# ---- one process per GPU: which GPU a rank sees, which cores it runs on (host logic, no GPU needed)

def _fake_sysfs(root, gpus):
    """A KFD topology with one CPU node and the given GPUs [(bus, numa node)], and two NUMA nodes of 8 cores."""
    nodes = root / "class/kfd/kfd/topology/nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 16\nsimd_count 0\n")
    for i, (bus, numa) in enumerate(gpus):
        d = nodes / str(i + 1)
        d.mkdir()
        d.joinpath("properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        pci = root / "bus/pci/devices" / ("0000:%02x:00.0" % bus)
        pci.mkdir(parents=True)
        pci.joinpath("numa_node").write_text(f"{numa}\n")
    for n in (0, 1):
        d = root / f"devices/system/node/node{n}"
        d.mkdir(parents=True)
        d.joinpath("cpulist").write_text(f"{8 * n}-{8 * n + 7}\n")


def test_cpulist_round_trip():
    assert shard.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert shard.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11"
    assert shard.parse_cpulist("") == [] and shard.format_cpulist([]) == ""


def test_gpu_numa_nodes_from_the_kfd_topology(tmp_path):
    _fake_sysfs(tmp_path, [(0x05, 0), (0x15, 0), (0x85, 1), (0x95, 1)])
    assert shard.gpu_numa_nodes(str(tmp_path)) == [0, 0, 1, 1]
    assert shard.count_gpus(str(tmp_path)) == 4
    assert shard.gpu_numa_nodes(str(tmp_path / "nothing_here")) is None


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_rank_cpu_sets_are_disjoint_and_follow_the_numa_nodes(world):
    avail = list(range(128))
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    node_cpus = {0: list(range(64)), 1: list(range(64, 128))}
    sets = shard.rank_cpu_sets(world, avail, nodes, node_cpus)
    assert len(sets) == world and all(sets)
    flat = [c for s in sets for c in s]
    assert len(flat) == len(set(flat))  # disjoint
    for r, s in enumerate(sets):
        assert set(s) <= set(node_cpus[nodes[r]])  # on the GPU's own node
    # no topology: an even split of what the process may use
    even = shard.rank_cpu_sets(world, range(16))
    assert [len(s) for s in even] == [16 // world] * world
    assert len({c for s in even for c in s}) == 16
    # two ranks on one GPU divide that GPU's node
    two = shard.rank_cpu_sets(2, range(16), [0], {0: list(range(16))}, gpus=[0, 0])
    assert two == [list(range(8)), list(range(8, 16))]


def _isolated(local_rank, world, sysfs, q):
    os.environ.pop("ROCR_VISIBLE_DEVICES", None)
    os.environ.pop("HIP_VISIBLE_DEVICES", None)
    got = shard.isolate_rank(local_rank, world, sysfs=sysfs)
    q.put((local_rank, got, os.environ.get("ROCR_VISIBLE_DEVICES"), sorted(os.sched_getaffinity(0))))


def test_isolate_rank_sets_the_gpu_and_the_affinity(tmp_path):
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) < 2:
        pytest.skip("needs two cores")
    _fake_sysfs(tmp_path, [(0x05, 0), (0x85, 0)])
    # (the fake nodes name cores 0-15; whatever of them this process may use is divided)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_isolated, args=(r, 2, str(tmp_path), q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join()
    (r0, g0, env0, aff0), (r1, g1, env1, aff1) = got
    assert (env0, env1) == ("0", "1") and (g0["gpu"], g1["gpu"]) == ("0", "1")
    assert aff0 and aff1 and not set(aff0) & set(aff1)
    assert shard.parse_cpulist(g0["cpus"]) == aff0 and shard.parse_cpulist(g1["cpus"]) == aff1
