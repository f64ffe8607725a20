cd /root/repo
run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline'].get('isolated_avg_us'))"; }
run A=1
run A=2
ESCX_BENCH_BREAKDOWN=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^#" > gpurun_out/A_breakdown.txt
cp efficient-speech-codec_amd/esc/lib/libescx.so /tmp/A.so
cp /root/repo/libescx_B.so efficient-speech-codec_amd/esc/lib/libescx.so
run B=1
run B=2
ESCX_BENCH_BREAKDOWN=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^#" > gpurun_out/B_breakdown.txt
