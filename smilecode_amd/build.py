"""Build libmodet_hip.so for gfx950 with plain hipcc (no torch headers, no cmake).

    python -m smilecode_amd.build            # incremental
    python -m smilecode_amd.build --force

Output: smilecode_amd/lib/libmodet_hip.so (git-ignored, travels with gpurun snapshots).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB = os.path.join(OUT_DIR, "libmodet_hip.so")
SOURCES = ["api.hip", "na.hip", "qk_op.hip", "warp.hip", "warp_tile.hip", "norm_act.hip", "proj_ln.hip", "losses.hip", "conv3d.hip", "conv3d_bf16.hip", "conv3d_x3.hip", "conv3d_wtr.hip", "conv3d_q.hip", "corr3d.hip", "eval.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-gpu-rdc",
         # NO packed-fp32 code from the SLP vectoriser (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 pairs formed out of scalar
         # source).  Round 5: with -O3 alone na_bwd_kernel's query role (27 x v_exp_f32 feeding v_pk_mul_f32 / v_pk_fma_f32
         # chains) and cwm_tail_bwd_kernel return WRONG values in lanes 48..63 of a wave about once per 100 launches when a
         # second process time-shares the GPU and a few high-priority streams exist in both (what two ranks on one GPU over
         # torch.distributed's gloo CUDA path look like: tests/test_gpu_e2e.py::test_data_parallel_two_ranks_equal_batch_two,
         # the red test of round 4).  A process alone never shows it; LDS / barrier / register probes with the same footprint
         # never show it (tools/micro/barrier_probe.hip, reg_probe.hip); forcing every s_waitcnt to zero does not cure it; the
         # scalar code does: 0 wrong launches in 20 000 (tools/exp_na_iso.py) and 0 bad steps in 8 000 (tools/race_hunt.py),
         # against 227 and 63-96.  The scalar build is not slower (8.40 vs 8.46 ms per train step: the packed pairs bought
         # nothing, the kernels that use them are latency- or memory-bound).  DESIGN.md section 6 has the whole hunt.
         "-fno-slp-vectorize"]
if os.environ.get("MODET_TUNING"):          # kernel-configuration overrides for tools/sweep_conv.py; never set for the product build
    FLAGS.append("-DMODET_TUNING")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


# kernels whose correctness beside another process depends on -fno-slp-vectorize (see FLAGS): source -> kernel name fragments.  After
# every recompile of these sources the device assembly is scanned: a packed-fp32 instruction inside them fails the build (ADVICE r5:
# the cure must not silently depend on a compiler heuristic -- a new compiler that forms v_pk_*_f32 by another pass would show here).
NO_PACKED_FP32 = {"na.hip": ("na_bwd_kernel", "na_bwd_march_kernel"), "warp.hip": ("cwm_tail_bwd_kernel",)}


def _check_no_packed_fp32(hipcc: str, src: str, kernels) -> None:
    import re
    asm = os.path.join(OBJ_DIR, os.path.basename(src).replace(".hip", ".s"))
    r = subprocess.run([hipcc] + FLAGS + ["--cuda-device-only", "-S", src, "-o", asm], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed: %s\n%s" % (src, r.stderr))
    cur, bad = None, {}
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        elif cur and any(k in cur for k in kernels) and re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
            bad[cur] = bad.get(cur, 0) + 1
    if bad:
        raise RuntimeError("packed-fp32 instructions in kernels that must be scalar (build.py FLAGS, DESIGN.md section 6): %s" % bad)


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "modet_hip.h"))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        return cmd[-1]

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for done in ex.map(run, jobs):
                if verbose:
                    print("[build] compiled", os.path.basename(done), flush=True)
            rebuilt = {os.path.basename(j[-3]) for j in jobs}
            checks = [(os.path.join(CSRC, k), v) for k, v in NO_PACKED_FP32.items() if k in rebuilt]
            for _ in ex.map(lambda kv: _check_no_packed_fp32(hipcc, *kv), checks):
                pass
            if checks and verbose:
                print("[build] no packed-fp32 code in", ", ".join(n for _, v in checks for n in v), flush=True)
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
        if verbose:
            print("[build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
