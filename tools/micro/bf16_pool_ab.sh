python -m pytest tests/test_gpu_bf16.py -x -q -k "fused_pool or end_to_end or train_steps" 2>&1 | tail -3
for i in 1 2; do
python bench.py --dtype bf16 --shape 160,192,224 --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5', d['ms_per_step'], d['value'])"
done
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 160', d['ms_per_step'], d['value'])"
