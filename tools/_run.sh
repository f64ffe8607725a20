cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', d['value'], d['ms_per_step'])"; }
run A=0
run A=1
ESCX_BENCH_BREAKDOWN=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^# pvq\|^# stft\|^# istft\|^# patch"
