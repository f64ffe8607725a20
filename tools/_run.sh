cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --skip-isolated 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
cp efficient-speech-codec_amd/esc/lib/libescx.so /tmp/new.so
for i in 1 2 3; do
cp /tmp/new.so efficient-speech-codec_amd/esc/lib/libescx.so; run NEW=$i
cp libescx_old.so efficient-speech-codec_amd/esc/lib/libescx.so; run OLD=$i
done
cp /tmp/new.so efficient-speech-codec_amd/esc/lib/libescx.so
ESCX_BENCH_BREAKDOWN=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^# pvq\|^# stft\|^# istft\|^# patch_embed_gemm"
