#!/bin/bash
# Run ON THE GPU BOX: MFMA-pipe utilisation of the SHIPPED conv kernels at the level-1/2 shapes.
#   rocprofv3 --pmc (counters only, no tracing) over tools/ab_kernels.py --only conv; per-kernel means by tools/pmc_summary.py.
#   busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), as profiles/r01_pmc_conv_fwd_8to8_before.txt.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d /tmp/pmc_mfma -o t -- python $R/tools/ab_kernels.py --only conv --iters 4 > /dev/null 2>&1
f=$(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_mfma_report.py $f > $OUT/${TAG}_pmc_mfma_conv.txt
cat $OUT/${TAG}_pmc_mfma_conv.txt
