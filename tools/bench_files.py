"""End-to-end rate of the file stream (mdc_stream_*): files on disk -> parsed batches on the GPU.
Plain and gzip -6 files (as tools/gzipfiles of the reference leaves them), host threads reading
and inflating ahead, uploads on a copy stream beside the parse.

    python tools/bench_files.py [--files 16] [--file-mb 64] [--threads 16] [--dir /tmp/mdc_files]
"""
import argparse
import gzip
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_file(args):
    d, i, mb = args
    from dsp_amd import gen
    data, rows = gen.mind_xml(nbytes=mb << 20, first_row=i * 1_000_000)
    raw = data.tobytes()
    plain = os.path.join(d, f"ZHOBTMIND_{i:04d}.xml")
    with open(plain, "wb") as f:
        f.write(raw)
    with open(plain + ".gz", "wb") as f:
        f.write(gzip.compress(raw, 6))
    return plain, rows, len(raw), os.path.getsize(plain + ".gz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=16)
    ap.add_argument("--file-mb", type=int, default=64)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--dir", default="/tmp/mdc_files")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=min(16, a.files)) as ex:
        made = list(ex.map(make_file, [(a.dir, i, a.file_mb) for i in range(a.files)]))
    print(f"made {a.files} files of {a.file_mb} MiB (+ gzip -6) in {time.perf_counter() - t0:.1f} s; "
          f"compression ratio {sum(m[2] for m in made) / sum(m[3] for m in made):.2f}", flush=True)
    from dsp_amd import capi, schema
    rows_expected = sum(m[1] for m in made)
    nbytes = sum(m[2] for m in made)
    for kind, paths in (("plain", [m[0] for m in made]), ("gzip", [m[0] + ".gz" for m in made])):
        for threads in sorted({1, 4, a.threads}):
            p = capi.Parser.xml(schema.ZHOBTMIND)
            p.parse_file(paths[0]).free()  # warm-up: kernels loaded, workspace sized
            t0 = time.perf_counter()
            rows = 0
            for idx, b in p.stream(paths, threads=threads):
                if isinstance(b, Exception):
                    raise b
                rows += b.rows
                b.free()
            dt = time.perf_counter() - t0
            assert rows == rows_expected
            print(f"{kind:5s} threads={threads:2d}: {nbytes / dt / 1e9:7.2f} GB/s of record bytes, "
                  f"{rows / dt / 1e6:7.1f} M records/s ({dt * 1e3:.0f} ms for {nbytes >> 20} MiB)", flush=True)
            p.close()
    for m in made:
        os.remove(m[0])
        os.remove(m[0] + ".gz")


if __name__ == "__main__":
    main()
