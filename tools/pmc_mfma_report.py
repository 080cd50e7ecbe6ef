#!/usr/bin/env python
"""MFMA-pipe busy fraction per conv kernel instantiation from a rocprofv3 --pmc counter_collection.csv
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA, GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_*):
    busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
(the same formula as profiles/r01_pmc_conv_fwd_8to8_before.txt).  Dispatches are grouped by (kernel name, grid size),
so the level-1 and level-2 launches of one instantiation are reported separately."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        if "conv3d" not in k and "conv_c1" not in k and "conv_x3" not in k:
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", k)
        k = re.sub(r"^void ", "", k)
        k = re.sub(r"\(.*$", "", k)
        acc[(k, r.get("Grid_Size", ""), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc over tools/ab_kernels.py --only conv (B=2 encoder batch; L1 = 160x192x160, L2 = 80x96x80)")
print("# busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMD x GRBM_GUI_ACTIVE / 8 XCD); clock = GRBM_GUI_ACTIVE/8 / duration is not in this file")
print(f"{'kernel':78s} {'grid':>9s} {'vgpr':>5s} {'n':>3s} {'gui_cyc/xcd':>12s} {'insts_mfma':>12s} {'mfma_busy':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s}")
for (k, grid, vg, lds), d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui <= 0:
        continue
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui)
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print(f"{k[:78]:78s} {grid:>9s} {vg:>5s} {len(d.get('GRBM_GUI_ACTIVE', [])):3d} {gui:12.0f} {m.get('SQ_INSTS_MFMA', 0):12.0f} {100 * busy:8.1f}% "
          f"{100 * m.get('SQ_WAIT_ANY', 0) / wc:7.1f}% {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:8.1f}% {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.1f}%")
