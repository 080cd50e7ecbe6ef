export MODET_HIP_LIB=/root/repo/build/variants/libmodet_hip_qT.so
python tools/exp_convq.py check 2>&1 | tail -3
for t in a b c d e f; do echo "== tiling $t"; MODET_CONVQ_TILING=$t python tools/exp_convq.py time 2>&1 | tail -17; done
