#!/usr/bin/env python3
"""Headline benchmark: observation XML parsed to columns on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (libmdc_parse.so: mdc_parse_device) over one batch of
synthetic ZHOBTMIND XML that is already resident in HBM.  At N=1 the batch is the 1 GiB
configuration the metric is quoted on; with N>1 every rank parses its own 1 GiB batch on its
own GPU (files shard independently: no data-path collective, only the row counts and timings
are gathered on the host over gloo).  Rank 0 prints one JSON line.

Started without WORLD_SIZE and with --gpus N > 1, this script starts the N ranks itself (fresh
child processes, before anything here has touched the GPU); under torch.distributed.run it is one
of the ranks.

`value` is input GB/s over all ranks (bytes consumed / max-over-ranks wall time of the K
steps).  `roofline` prices the whole parse (algorithmic bytes over the device time of all its
kernels, HIP events taken inside the library on the stream they run on), names the longest kernel
of the run and carries each of the two kernels with its own bytes; `cpu_baseline` times the CPU checker (the
reference's own code when oracle/_ref was built, else the C restatement) on a bounded sample of
the same bytes, single-threaded as the reference runs.  Beside the headline, in the same line:

  archive    the file-sharded path (BASELINE config 4 in small): every rank streams its share of
             a sorted list of 128 MiB files through mdc_stream_* (read into pinned memory, upload
             on a copy stream beside the parse); end-to-end GB/s over all ranks, row totals
  e2e        (N=1) end-to-end rates that include the way into HBM: a host buffer (PCIe copy
             included), plain files, gzip files with 1 and 16 inflating threads
  delimited  (N=1) the 4 GiB ccmdstr-split configuration with a roofline of its own
"""
import argparse
import ctypes
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dsp_amd import archive, capi, gen, schema, shard  # noqa: E402  (nothing here touches the GPU at import)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed parses in front (the GPU reaches its steady clocks after 10 to 15 parses of 1 GiB)")
    ap.add_argument("--workload", choices=["xml", "csv"], default="xml")
    ap.add_argument("--bytes", type=int, default=1 << 30, help="batch size per GPU")
    ap.add_argument("--cpu-sample-mb", type=int, default=256, help="0 disables the CPU baseline")
    ap.add_argument("--cpu-procs", type=int, default=16,
                    help="processes of the all-cores CPU baseline (one 64 MiB batch each); 0 disables it")
    ap.add_argument("--verify", type=int, default=1, help="check a slice of the output against the CPU checker")
    ap.add_argument("--archive-files-per-gpu", type=int, default=8, help="0 disables the file-sharded section")
    ap.add_argument("--archive-file-mb", type=int, default=128)
    ap.add_argument("--archive-threads", type=int, default=8, help="host threads reading ahead per rank")
    ap.add_argument("--archive-dir", default="", help="where the files are written (default: a temp dir in /dev/shm)")
    ap.add_argument("--e2e", type=int, default=1, help="N=1: host-buffer, plain-file and gzip end-to-end rates")
    ap.add_argument("--one-pass", type=int, default=0,
                    help="1: parsers are created with MDC_F_ONE_PASS (sizing and writing in one kernel from the second parse on; "
                         "measured slower than the two passes, kept for comparison)")
    ap.add_argument("--delimited-gb", type=float, default=4.0,
                    help="N=1, xml workload: size in GiB of the delimited configuration's batch; 0 disables the section")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# N > 1 without a launcher: this process only starts the ranks

def launch_ranks(a):
    """Starts a.gpus rank processes of this script and passes rank 0's line through.  Called before
    anything in this process has touched the GPU.  Every rank narrows itself to one GPU and a CPU set of
    its own when it starts (shard.isolate_rank in main), as it does under an external launcher."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tmp = a.archive_dir or tempfile.mkdtemp(prefix="mdc_arch_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MDC_BENCH_ARCHIVE_DIR=tmp)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    if not a.archive_dir:
        shutil.rmtree(tmp, ignore_errors=True)
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------

def make_parser(workload, device, flags=0):
    if workload == "xml":
        return capi.Parser.xml(schema.ZHOBTMIND, device=device, flags=flags)
    return capi.Parser.csv(",", 10, typed=schema.MIND_CSV_TYPED, device=device, flags=flags)


def make_data(workload, nbytes, rank):
    first = rank * 10_000_000
    if workload == "xml":
        return gen.mind_xml(nbytes=nbytes, first_row=first)
    return big_csv(nbytes, first)


def big_csv(nbytes, first_row=0, part=256 << 20):
    """A delimited batch of about nbytes, generated in parts by threads (the generator is C code
    that runs without the interpreter lock) and joined."""
    from concurrent.futures import ThreadPoolExecutor
    nparts = max(1, (nbytes + part - 1) // part)
    sizes = [min(part, nbytes - i * part) for i in range(nparts)]
    with ThreadPoolExecutor(max_workers=min(16, nparts)) as ex:
        made = list(ex.map(lambda i: gen.mind_csv(nbytes=sizes[i], first_row=first_row + i * 5_000_000), range(nparts)))
    if nparts == 1:
        return made[0]
    return np.concatenate([m[0] for m in made]), sum(m[1] for m in made)


def cpu_baseline(workload, data, sample_mb):
    """Reference (or port) CPU path on a bounded prefix of the same bytes, one thread."""
    from oracle import oracle_lib
    kind = "reference" if oracle_lib.available("reference") else "port"
    chk = oracle_lib.load(kind)
    nbytes = min(len(data), sample_mb << 20)
    cut = int(np.flatnonzero(data[:nbytes] == 10)[-1]) + 1  # end on a line boundary
    sample = np.ascontiguousarray(data[:cut])
    t0 = time.perf_counter()
    if workload == "xml":
        hb = chk.parse_xml(sample, schema.ZHOBTMIND)
    else:
        hb = chk.parse_csv(sample, ",", 10, typed=schema.MIND_CSV_TYPED)
    dt = time.perf_counter() - t0
    return {
        "value": round(cut / dt / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": kind,
        "records_per_s": round(hb.rows / dt, 1),
        "sample": f"first {cut} bytes ({hb.rows} records) of the rank-0 batch, {dt:.2f} s"
                  + (", incl. the reference reading it back from a temp file" if kind == "reference" else ""),
    }, hb, cut


def cpu_baseline_all_cores(workload, procs):
    """How the reference scales in deployment: one single-threaded process per share of the
    files.  `procs` worker processes (tools/cpu_slice.py), each with a 64 MiB batch of its own,
    started together; the rate is all their bytes over the slowest one's time."""
    procs = max(1, min(procs, os.cpu_count() or 1))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_slice.py"), workload, str(64 << 20)]
    ps = [subprocess.Popen(cmd + [str(20_000_000 + 1_000_000 * i)], stdout=subprocess.PIPE, text=True) for i in range(procs)]
    res = []
    for p in ps:
        out, _ = p.communicate(timeout=300)
        if p.returncode != 0:
            return None
        res.append(json.loads(out.strip().splitlines()[-1]))
    slowest = max(r["seconds"] for r in res)
    return {
        "value": round(sum(r["bytes"] for r in res) / slowest / 1e9, 4), "unit": "GB/s", "cores": procs,
        "kind": res[0]["kind"], "records_per_s": round(sum(r["rows"] for r in res) / slowest, 1),
        "sample": f"{procs} processes, one 64 MiB batch each, slowest {slowest:.2f} s",
    }


def copy_ceiling(device, src_ptr, nbytes, repeats=10):
    """A measured ceiling beside the data-sheet peak: device-to-device hipMemcpy of the input
    batch (one byte read and one written per byte, the mix closest to the write kernel's), timed
    with HIP events on the null stream; GB/s read + written.  None when the runtime library cannot
    be reached from here."""
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return None
    vp = ctypes.c_void_p
    dst = vp()
    if capi.lib().mdc_device_alloc(device, nbytes, ctypes.byref(dst)) != 0:
        return None
    try:
        hip.hipMemcpyAsync.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, vp]
        hip.hipEventRecord.argtypes = [vp, vp]
        hip.hipEventSynchronize.argtypes = [vp]
        hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), vp, vp]
        hip.hipEventDestroy.argtypes = [vp]
        e0, e1 = vp(), vp()
        if hip.hipSetDevice(device) or hip.hipEventCreate(ctypes.byref(e0)) or hip.hipEventCreate(ctypes.byref(e1)):
            return None
        best = None
        for _ in range(repeats):
            rc = hip.hipEventRecord(e0, None)
            rc |= hip.hipMemcpyAsync(dst, vp(src_ptr), nbytes, 3, None)  # 3 = hipMemcpyDeviceToDevice
            rc |= hip.hipEventRecord(e1, None)
            rc |= hip.hipEventSynchronize(e1)
            ms = ctypes.c_float()
            rc |= hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
            if rc:
                return None
            best = ms.value if best is None else min(best, ms.value)
        hip.hipEventDestroy(e0)
        hip.hipEventDestroy(e1)
        return round(2.0 * nbytes / (best * 1e-3) / 1e9, 1)
    finally:
        capi.lib().mdc_device_free(device, dst)


SIZING_KERNEL = "mdc::tile_kernel<0> (sizing pass)"
WRITE_KERNEL = {"xml": "mdc::emit_kernel<false, false> (write pass)", "csv": "mdc::emit_kernel<false, true> (write pass)"}


def measured_traffic(nbytes, name="traffic.json"):
    """HBM bytes per launch of the two kernels of a parse from the committed PMC passes (FETCH_SIZE and
    WRITE_SIZE need separate profiler runs, so they cannot be taken inside this process); only quoted when
    that profile was made on the same workload size.  {"sizing": bytes, "write": bytes} or None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None
    if t.get("workload_bytes") != nbytes or "kernels" not in t:
        return None
    return {k: int(v["traffic_bytes"]) for k, v in t["kernels"].items()}


def timed_steps(parser, dbuf, steps, warmup, barrier):
    """W untimed parses, then exactly K timed ones between two barriers.  Returns wall seconds, the
    summed HIP-event phase times, rows, output bytes."""
    def step():
        b = parser.parse_device(dbuf.ptr, dbuf.n)
        rows = b.rows
        t = parser.timing()
        b.free()
        return rows, t

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = {"count": 0.0, "scan": 0.0, "write": 0.0, "total": 0.0}
    rows, bytes_out = 0, 0
    for _ in range(steps):
        rows, t = step()
        acc["count"] += t.count_ms
        acc["scan"] += t.scan_ms
        acc["write"] += t.write_ms
        acc["total"] += t.total_ms
        bytes_out = t.bytes_out
    capi.check(capi.lib().mdc_device_synchronize(dbuf.device))
    local_s = time.perf_counter() - t0
    barrier()
    return local_s, acc, rows, bytes_out


def roofline_of(workload, nbytes, bytes_out, acc, steps, traffic):
    """The roofline of a parse.  The headline figures are the WHOLE parse's: its algorithmic bytes (the
    input read once and every output byte written once) over the device time of all its kernels (sizing
    pass, scan, write pass: HIP events inside the library on the stream they run on); the write pass's
    second read of the input is not credited.  `kernel` names the longest kernel of this same run, and
    `kernels` prices each of the two with its own bytes (the sizing pass reads the input and writes
    next to nothing; the write pass reads it again and writes the outputs)."""
    sizing_ms, write_ms, total_ms = acc["count"] / steps, acc["write"] / steps, acc["total"] / steps

    def entry(name, alg, ms, tr):
        ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0  # (a one-pass parse has no sizing pass of its own)
        return {"kernel": name, "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(alg),
                "achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": tr}

    kernels = [entry(SIZING_KERNEL, nbytes, sizing_ms, traffic["sizing"] if traffic else None),
               entry(WRITE_KERNEL[workload], nbytes + bytes_out, write_ms, traffic["write"] if traffic else None)]
    alg = nbytes + bytes_out
    achieved = alg / (total_ms * 1e-3) / 1e9
    return {
        "bound": "hbm", "kernel": max(kernels, key=lambda e: e["avg_launch_ms"])["kernel"],
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": sum(traffic.values()) if traffic else None,
        "scope": "whole parse: algorithmic bytes (input once + outputs once) / device time of all its kernels",
        "algorithmic_bytes_per_parse": int(alg),
        "device_ms_per_parse": round(total_ms, 4),
        "kernels": kernels,
    }


def verify_prefix(parser, dbuf, data, workload):
    """The first 8 MiB of the batch, cell for cell, against the CPU checker."""
    from oracle import oracle_lib
    chk = oracle_lib.load("port")
    cut = int(np.flatnonzero(data[: 8 << 20] == 10)[-1]) + 1
    if workload == "xml":
        exp = chk.parse_xml(np.ascontiguousarray(data[:cut]), schema.ZHOBTMIND)
    else:
        exp = chk.parse_csv(np.ascontiguousarray(data[:cut]), ",", 10, typed=schema.MIND_CSV_TYPED)
    b = parser.parse_device(dbuf.ptr, dbuf.n)
    ok = True
    n = exp.rows
    for c in range(exp.ncols):
        off, dat = b.column(c)
        ok &= bool(np.array_equal(off[: n + 1], exp.off[c]))
        ok &= bool(np.array_equal(dat[: int(exp.off[c][n])], exp.data[c]))
        if exp.val[c] is not None:
            ok &= bool(np.array_equal(b.values(c)[:n], exp.val[c]))
    b.free()
    if not ok:
        raise SystemExit("GPU output differs from the CPU checker: refusing to report a number")
    return int(n)


def archive_section(a, parser, rank, world, barrier, reduce_sum, reduce_max, gather, files):
    """Every rank has written the files of its share (deterministic: file i is the same bytes
    whoever writes it); all ranks stream their shares between two barriers."""
    nfiles = a.archive_files_per_gpu * world
    paths, rows_exp, nbytes = files
    # untimed: the first files once, so that the pinned buffers (kept from stream to stream) and the device
    # workspace are sized for these files, as in an ingest that has been running
    archive.run_share(parser, paths[: a.archive_threads + 1], threads=a.archive_threads)
    barrier()
    res = archive.run_share(parser, paths, threads=a.archive_threads)
    barrier()
    for p in paths:
        os.remove(p)
    if res.errors or res.rows_per_file != rows_exp:
        raise SystemExit(f"archive share of rank {rank}: rows per file differ from the generator's ({res.errors[:2]})")
    tot = reduce_sum([float(res.rows), float(sum(nbytes)), float(len(paths))])
    slowest = reduce_max(res.seconds)
    per_rank = gather(round(sum(nbytes) / res.seconds / 1e9, 3))
    return {
        "workload": f"{nfiles} files of {a.archive_file_mb} MiB ZHOBTMIND XML in sorted order, dealt round-robin to {world} rank(s), "
                    "each share through mdc_stream (page cache -> pinned -> HBM -> columns)",
        "files": int(tot[2]), "bytes": int(tot[1]), "rows": int(tot[0]), "rows_match_generator": True,
        "threads_per_rank": a.archive_threads, "seconds": round(slowest, 4),
        "GBps_end_to_end": round(tot[1] / slowest / 1e9, 3), "records_per_s": round(tot[0] / slowest, 1),
        "per_rank_GBps": per_rank,
        "note": "the file-sharded figure (GBps_end_to_end over 1/2/4/8 ranks) is the scaling metric of the archive configuration; "
                "it is bound by the host (page-cache reads into pinned memory), not by the GPUs",
    }


def e2e_section(a, parser, data, gz_files):
    """N=1: what the way into HBM costs.  Never the headline."""
    out = {}
    parser.parse_host(data[: 64 << 20]).free()
    t0 = time.perf_counter()
    b = parser.parse_host(data)
    dt = time.perf_counter() - t0
    b.free()
    out["host_buffer_GBps"] = round(len(data) / dt / 1e9, 3)  # pageable source, PCIe copy included
    paths, rows_exp, nbytes = gz_files
    for threads in (1, 16):
        res = archive.run_share(parser, paths, threads=threads)
        if res.errors or res.rows_per_file != rows_exp:
            raise SystemExit("gzip stream: rows differ from the generator's")
        out[f"gzip_stream_GBps_{threads}_threads"] = round(sum(nbytes) / res.seconds / 1e9, 3)
    out["gzip_sample"] = f"16 files of 32 MiB, gzip -6 ({sum(os.path.getsize(p) for p in paths) >> 20} MiB compressed), inflated by host threads (zlib)"
    for p in paths:
        os.remove(p)
    # the reference's own record loop (cifile::open, readline("<endl/>") until the end, count) over
    # include/mdc_public_compat.h: file -> pinned memory -> HBM -> record split on the GPU -> one
    # std::string per record on the host; a child process, three timed rounds behind an untimed one
    exe = os.path.join(ROOT, "examples", "xmltodb_compat")
    if os.path.exists(exe) and paths:
        f = os.path.join(os.path.dirname(paths[0]), "compat_readline.xml")
        cut = int(np.flatnonzero(data[: 128 << 20] == 10)[-1]) + 1
        data[:cut].tofile(f)
        try:
            r = subprocess.run([exe, "readloop", f, "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            rounds = [l.split() for l in r.stdout.splitlines() if l.startswith("round ")]
            if r.returncode == 0 and rounds:
                best = min(float(x[7]) for x in rounds)
                out["compat_readline_GBps"] = round(cut / best / 1e9, 3)
                out["compat_readline_records_per_s"] = round(int(rounds[0][3]) / best, 1)
                out["compat_readline_sample"] = f"{cut} bytes, {rounds[0][3]} records, best of 3 rounds {best:.4f} s (file read, upload and the host's per-record strings included)"
        finally:
            os.remove(f)
    return out


def delimited_section(a, device):
    """N=1: BASELINE's delimited configuration (ccmdstr split, 10 fields, 7 doubles and a long)."""
    nbytes = int(a.delimited_gb * (1 << 30))
    data, _ = big_csv(nbytes)
    parser = make_parser("csv", device)
    dbuf = capi.DeviceBuffer(data, device)
    sync = lambda: capi.check(capi.lib().mdc_device_synchronize(device))  # noqa: E731
    steps, warmup = 10, 5
    local_s, acc, rows, bytes_out = timed_steps(parser, dbuf, steps, warmup, sync)
    verified = verify_prefix(parser, dbuf, data, "csv")
    sec = {
        "config": {"workload": "%.3g GiB synthetic ZHOBTMIND delimited batch, ccmdstr split -> 10 fields" % (dbuf.n / 2**30),
                   "bytes": dbuf.n, "records": rows},
        "value": round(dbuf.n * steps / local_s / 1e9, 3), "unit": "GB/s", "steps": steps, "warmup": warmup,
        "ms_per_step": round(local_s / steps * 1e3, 4), "records_per_s": round(rows * steps / local_s, 1),
        "roofline": roofline_of("csv", dbuf.n, bytes_out, acc, steps, measured_traffic(dbuf.n, "traffic_delimited.json")),
        "kernel_ms": {"sizing_pass": round(acc["count"] / steps, 4), "scan": round(acc["scan"] / steps, 4),
                      "write_pass": round(acc["write"] / steps, 4), "device_total": round(acc["total"] / steps, 4)},
        "verified_records": verified,
    }
    dbuf.free()
    parser.close()
    return sec


def main():
    a = parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    iso = {"gpu": None, "cpus": shard.format_cpulist(sorted(os.sched_getaffinity(0)))}
    if world > 1:
        # one GPU and a CPU set of its own per rank, before anything touches the runtime: the rank's
        # device 0 is then its GPU, and the threads it starts later (readers, uploads) stay on its cores
        try:
            iso = shard.isolate_rank(local_rank, world)
        except Exception as e:  # (an unreadable topology must not cost the run: all GPUs stay visible, rank r takes GPU r)
            iso["error"] = repr(e)
        import torch
        import torch.distributed as dist  # gloo on host tensors only: timings and row counts
        dist.init_process_group("gloo", rank=rank, world_size=world)
    # The files of the file-sharded sections are written first, by worker processes, while this
    # process has not touched the GPU yet (no fork of a process that holds a HIP context).
    arch_files = gz_files = base_dir = own_dir = None
    if a.archive_files_per_gpu > 0 and a.workload == "xml":
        base_dir = os.environ.get("MDC_BENCH_ARCHIVE_DIR") or a.archive_dir
        shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
        if not base_dir:
            if world > 1:  # (under an external launcher the ranks agree on a place by its port)
                base_dir = os.path.join(shm or tempfile.gettempdir(), "mdc_arch_%s" % os.environ.get("MASTER_PORT", "0"))
            else:
                base_dir = own_dir = tempfile.mkdtemp(prefix="mdc_arch_", dir=shm)
        mine = archive.rank_share(a.archive_files_per_gpu * world, rank, world)
        arch_files = archive.make_files(base_dir, mine, a.archive_file_mb << 20, workers=8)
        if a.e2e and world == 1:
            gz_files = archive.make_files(base_dir, range(900, 916), 32 << 20, gz=True, workers=8)
    ndev = capi.lib().mdc_device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the parsing path exists only as gfx950 code")
    device = local_rank % ndev  # (0 for a rank that was narrowed to its own GPU)

    def reduce_sum(vals):
        if dist is None:
            return list(vals)
        t = torch.tensor(list(vals), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.tolist()

    def reduce_max(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def gather(obj):
        """Every rank's object, in rank order (host objects over gloo)."""
        if dist is None:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    data, rows_gen = make_data(a.workload, a.bytes, rank)
    parser = make_parser(a.workload, device, capi.MDC_F_ONE_PASS if a.one_pass else 0)
    dbuf = capi.DeviceBuffer(data, device)
    L = capi.lib()

    def barrier():
        capi.check(L.mdc_device_synchronize(device))
        if dist is not None:
            dist.barrier()

    local_s, acc, rows, bytes_out = timed_steps(parser, dbuf, a.steps, a.warmup, barrier)
    elapsed = reduce_max(local_s)
    total_bytes, total_rows = reduce_sum([float(dbuf.n), float(rows)])
    ranks = gather({"rank": rank, "gpu": iso["gpu"], "device": device, "cpus": iso["cpus"]})

    arch = e2e = None
    if arch_files is not None:
        arch = archive_section(a, parser, rank, world, barrier, reduce_sum, reduce_max, gather, arch_files)
        if gz_files is not None:
            e2e = e2e_section(a, parser, data, gz_files)
        if own_dir:
            shutil.rmtree(own_dir, ignore_errors=True)
        elif world > 1 and not (os.environ.get("MDC_BENCH_ARCHIVE_DIR") or a.archive_dir):
            barrier()
            if rank == 0:
                shutil.rmtree(base_dir, ignore_errors=True)

    if rank == 0:
        k = a.steps
        write_ms = acc["write"] / k
        out = {
            "metric": "obs_%s_parsed_to_columns_GBps" % a.workload,
            "value": round(total_bytes * k / elapsed / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": k,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / k * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": ("%.3g GiB synthetic ZHOBTMIND surface-observation XML batch -> 11 columns" % (dbuf.n / 2**30)
                             if a.workload == "xml" else
                             "%.3g GiB synthetic ZHOBTMIND delimited batch, ccmdstr split -> 10 fields" % (dbuf.n / 2**30)),
                "bytes_per_gpu": dbuf.n, "records_per_gpu": rows, "sharding": "one batch per GPU, no collective",
            },
            "records_per_s": round(total_rows * k / elapsed, 1),
            "ranks": ranks,
            "roofline": roofline_of(a.workload, dbuf.n, bytes_out, acc, k,
                                    measured_traffic(dbuf.n) if a.workload == "xml" else None),
            "kernel_ms": {"sizing_pass": round(acc["count"] / k, 4), "scan": round(acc["scan"] / k, 4),
                          "write_pass": round(write_ms, 4), "device_total": round(acc["total"] / k, 4)},
        }
        if a.one_pass:  # (one kernel: the roofline's two-kernel split does not apply; device_total is that kernel)
            out["one_pass"] = True
            out["roofline"]["kernels"] = [out["roofline"]["kernels"][1]]
            out["roofline"]["kernels"][0]["kernel"] = out["roofline"]["kernel"] = "mdc::tile_kernel<2> (one-pass parse)"
        if world == 1:
            ceiling = copy_ceiling(device, dbuf.ptr, dbuf.n)
            if ceiling:
                out["roofline"]["measured_copy_GBps"] = ceiling  # hipMemcpy device to device, read + written
                out["roofline"]["frac_of_measured_copy"] = round(out["roofline"]["achieved"] / ceiling, 4)  # (whole parse)
        if a.verify:
            out["verified_records"] = verify_prefix(parser, dbuf, data, a.workload)
        if arch is not None:
            out["archive"] = arch
        if e2e is not None:
            e2e["plain_file_stream_GBps"] = arch["GBps_end_to_end"]
            e2e["plain_file_threads"] = a.archive_threads
            out["e2e"] = e2e
    dbuf.free()
    parser.close()
    if rank == 0:
        if a.workload == "xml" and a.delimited_gb > 0 and world == 1:
            out["delimited"] = delimited_section(a, device)
        if a.cpu_sample_mb > 0 and world == 1:
            cb, _, _ = cpu_baseline(a.workload, data, a.cpu_sample_mb)
            out["cpu_baseline"] = cb
            if a.cpu_procs > 0:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(a.workload, a.cpu_procs)
        elif a.cpu_sample_mb > 0:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
