# conv3d_q.hip step-level A/B: a tuning build of conv3d.hip with MODET_CONV_Q=0 (the kernels it replaces) against the default dispatch
for i in 1 2 3; do
MODET_HIP_LIB=/root/repo/build/variants/libmodet_hip_q0.so MODET_CONV_Q=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['ms_per_step'])"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'])"
done
MODET_HIP_LIB=/root/repo/build/variants/libmodet_hip_q0.so MODET_CONV_Q=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --dtype bf16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old bf16', d['ms_per_step'])"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --dtype bf16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new bf16', d['ms_per_step'])"
