# weight gradients of the small levels on a side stream (ops.WGRAD_SIDE_MAX_VOXELS): step-level A/B in the replayed graph
for rep in 1 2; do
for t in 0 200000 700000 1300000; do
MODET_WGRAD_SIDE=$t python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side<=$t', d['ms_per_step'], d['loss_after_timed_region'], d['hip_graph'])"
done; done
MODET_WGRAD_SIDE=700000 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
