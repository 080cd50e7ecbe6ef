#!/bin/bash
# Run ON THE GPU BOX: two rocprofv3 --pmc passes (counters only) over any command, per-(kernel, grid) means of every counter.
#   bash tools/pmc_any.sh <kernel-name-filter> <command...>
# pass A: matrix pipe / wave-cycle split; pass B: LDS and VALU.  busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMD x GUI/8).
R=${GRAFT_REPO_ROOT:-/root/repo}
flt=$1; shift
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d /tmp/pmcany.XXXX)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $d/a -o t -- "$@" > $d/a.out 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES \
  --output-format csv -d $d/b -o t -- "$@" > $d/b.out 2>&1 < /dev/null
python - "$flt" $(find $d/a -name "*counter_collection.csv" | head -1) $(find $d/b -name "*counter_collection.csv" | head -1) <<'PY'
import csv, re, sys
from collections import defaultdict
flt = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list)); order = []
for fn in sys.argv[2:]:
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*$", "", k)
        if flt not in k: continue
        key = (k, r.get("Grid_Size", ""))
        if key not in acc: order.append(key)
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in order:
    m = {c: sum(v) / len(v) for c, v in acc[key].items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    wc = m.get("SQ_WAVE_CYCLES", 1) or 1
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * gui) if gui else 0
    print("%-44s grid %-8s gui/xcd %8.0f  mfma_busy %5.1f%%  insts_mfma %9.0f | of wave cycles: wait_any %4.1f%% wait_inst %4.1f%% (lds %4.1f%%) active %4.1f%% (valu %4.1f%% lds %4.1f%%) | lds idx_active %9.0f bank_conflict %9.0f insts_lds %9.0f insts_valu %9.0f" % (
        key[0][:44], key[1], gui, 100 * busy, m.get("SQ_INSTS_MFMA", 0), 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * m.get("SQ_WAIT_INST_LDS", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        100 * m.get("SQ_ACTIVE_INST_LDS", 0) / wc, m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_VALU", 0)))
PY
