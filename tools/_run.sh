cd /root/repo
run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', d['value'], d['ms_per_step'])"; }
run A=0
run ESCX_MLP_TM2_MAXCP=48
run ESCX_MLP_TM2_MAXCP=96
run ESCX_MLP_TM2_MAXCP=144
run ESCX_MLP_TM2_MAXCP=48 ESCX_MLP_TM2_NW8=1
run ESCX_MLP_TM2_MAXCP=96 ESCX_MLP_TM2_NW8=1
run A=0
