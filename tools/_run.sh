cd /root/repo
ESCX_BENCH_BREAKDOWN=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^# attn" > gpurun_out/attn_base.txt
cp libescx_nb.so efficient-speech-codec_amd/esc/lib/libescx.so
ESCX_BENCH_BREAKDOWN=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^# attn" > gpurun_out/attn_nobarrier.txt
