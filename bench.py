#!/usr/bin/env python3
"""Headline benchmark: observation XML parsed to columns on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (libmdc_parse.so: mdc_parse_device) over one batch of
synthetic ZHOBTMIND XML that is already resident in HBM.  At N=1 the batch is the 1 GiB
configuration the metric is quoted on; with N>1 every rank parses its own 1 GiB batch on its
own GPU (files shard independently: no data-path collective, only the row counts and timings
are gathered on the host over gloo).  Rank 0 prints one JSON line.

`value` is input GB/s over all ranks (bytes consumed / max-over-ranks wall time of the K
steps).  `roofline` prices the dominant kernel (the tile kernel in write mode) with HIP-event
durations taken inside the library on the stream it launches on; `cpu_baseline` times the CPU
checker (the reference's own code when oracle/_ref was built, else the C restatement) on a
bounded sample of the same bytes, single-threaded as the reference runs.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dsp_amd import capi, gen, schema  # noqa: E402  (loads the HIP library before anything else)

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
    return ap.parse_args()


def make_parser(workload, device):
    if workload == "xml":
        return capi.Parser.xml(schema.ZHOBTMIND, device=device)
    return capi.Parser.csv(",", 10, typed=schema.MIND_CSV_TYPED, device=device)


def make_data(workload, nbytes, rank):
    first = rank * 10_000_000
    if workload == "xml":
        return gen.mind_xml(nbytes=nbytes, first_row=first)
    return gen.mind_csv(nbytes=nbytes, first_row=first)


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
    import subprocess
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
    import ctypes
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


def measured_traffic(kernel, nbytes):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE and
    WRITE_SIZE need separate profiler runs, so they cannot be taken inside this process); only
    quoted when that profile was made on the same workload size."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    if t.get("workload_bytes") != nbytes or t.get("kernel") not in kernel:
        return None
    return int(t["traffic_bytes"])


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist  # gloo on host tensors only: timings and row counts
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import torch
    ndev = capi.lib().mdc_device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the parsing path exists only as gfx950 code")
    device = local_rank % ndev

    data, rows_gen = make_data(a.workload, a.bytes, rank)
    parser = make_parser(a.workload, device)
    dbuf = capi.DeviceBuffer(data, device)
    L = capi.lib()

    def barrier():
        capi.check(L.mdc_device_synchronize(device))
        if dist is not None:
            dist.barrier()

    def step():
        b = parser.parse_device(dbuf.ptr, dbuf.n)
        rows = b.rows
        t = parser.timing()
        b.free()
        return rows, t

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = {"count": 0.0, "scan": 0.0, "write": 0.0, "total": 0.0}
    rows = 0
    for _ in range(a.steps):
        rows, t = step()
        acc["count"] += t.count_ms
        acc["scan"] += t.scan_ms
        acc["write"] += t.write_ms
        acc["total"] += t.total_ms
        bytes_out = t.bytes_out
    capi.check(L.mdc_device_synchronize(device))
    local_s = time.perf_counter() - t0
    barrier()
    elapsed = local_s
    total_bytes = float(dbuf.n)
    total_rows = float(rows)
    if dist is not None:
        tt = torch.tensor([local_s], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
        cnt = torch.tensor([float(dbuf.n), float(rows)], dtype=torch.float64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_bytes, total_rows = float(cnt[0]), float(cnt[1])

    if rank == 0:
        k = a.steps
        write_ms = acc["write"] / k
        alg_bytes = dbuf.n + bytes_out
        achieved = alg_bytes / (write_ms * 1e-3) / 1e9
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
            "roofline": {
                "bound": "hbm", "kernel": "mdc::emit_kernel (write pass)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": measured_traffic("mdc::emit_kernel", dbuf.n) if a.workload == "xml" else None,
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "avg_launch_ms": round(write_ms, 4),
            },
            "kernel_ms": {"sizing_pass": round(acc["count"] / k, 4), "scan": round(acc["scan"] / k, 4),
                          "write_pass": round(write_ms, 4), "device_total": round(acc["total"] / k, 4)},
            "hbm_frac_whole_parse": round(alg_bytes / (acc["total"] / k * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        }
        if world == 1:
            ceiling = copy_ceiling(device, dbuf.ptr, dbuf.n)
            if ceiling:
                out["roofline"]["measured_copy_GBps"] = ceiling  # hipMemcpy device to device, read + written
                out["roofline"]["frac_of_measured_copy"] = round(achieved / ceiling, 4)
        if a.verify:
            from oracle import oracle_lib
            chk = oracle_lib.load("port")
            cut = int(np.flatnonzero(data[: 8 << 20] == 10)[-1]) + 1
            if a.workload == "xml":
                exp = chk.parse_xml(np.ascontiguousarray(data[:cut]), schema.ZHOBTMIND)
            else:
                exp = chk.parse_csv(np.ascontiguousarray(data[:cut]), ",", 10, typed=schema.MIND_CSV_TYPED)
            b = parser.parse_device(dbuf.ptr, dbuf.n)
            ok = True
            for c in range(exp.ncols):
                off, dat = b.column(c)
                n = exp.rows
                ok &= bool(np.array_equal(off[: n + 1], exp.off[c]))
                ok &= bool(np.array_equal(dat[: int(exp.off[c][n])], exp.data[c]))
                if exp.val[c] is not None:
                    ok &= bool(np.array_equal(b.values(c)[:n], exp.val[c]))
            out["verified_records"] = int(exp.rows) if ok else 0
            b.free()
            if not ok:
                raise SystemExit("GPU output differs from the CPU checker: refusing to report a number")
        if a.cpu_sample_mb > 0 and world == 1:
            cb, _, _ = cpu_baseline(a.workload, data, a.cpu_sample_mb)
            out["cpu_baseline"] = cb
            if a.cpu_procs > 0:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(a.workload, a.cpu_procs)
        elif a.cpu_sample_mb > 0:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    dbuf.free()
    parser.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
