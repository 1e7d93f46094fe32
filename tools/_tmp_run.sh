set -x
python -m pytest tests/test_gpu_sprites.py tests/test_gpu_f32.py tests/test_gpu_sprites_cgen.py -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_fullsize.py -x -q -k "sprites" 2>&1 | tail -4
for i in 1 2; do
python bench.py --workload sprites800 --precision f32 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print('NEW', l['ms_per_step'], l.get('nets_ms'))"
SVGP_CONV_WGRAD_GRID=0 SVGP_CONV_WGRAD_RING=0 SVGP_CONV_THIN_FWD=0 python bench.py --workload sprites800 --precision f32 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print('OLD', l['ms_per_step'], l.get('nets_ms'))"
done
