set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_archive.py tests/test_gpu_parity.py -m gpu -x -q -k "archive or file or stream or gzip" --timeout 300 --timeout-method thread 2>&1 | tail -3
for t in 4 8; do
timeout -k 10 300 python -u bench.py --steps 5 --warmup 3 --cpu-sample-mb 0 --verify 0 --delimited-gb 0 --archive-threads $t 2>&1 | tail -1 | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); a=d['archive']; print('threads', a['threads_per_rank'], 'GB/s', a['GBps_end_to_end'], 'sec', a['seconds'], d['e2e'])"
done
