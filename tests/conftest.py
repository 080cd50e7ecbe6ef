import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- the two full-size fp64 oracle runs (tests/oracle_jobs.py) as background processes beside the GPU tests
_ORACLE_USERS = {
    "full160": ("test_full_size_dice_parity_vs_oracle", "test_full_size_gradient_parity_vs_oracle"),
    "cfg5": ("test_cfg5_shape_fp32_parity_vs_fp64_oracle", "test_cfg5_shape_bf16_flow_and_dice_vs_fp64_oracle"),
}
_jobs = {}


def _start_job(name):
    if name not in _jobs:
        d = tempfile.mkdtemp(prefix="modet_oracle_")
        out = os.path.join(d, name + ".pt")
        log = open(os.path.join(d, name + ".log"), "w")
        p = subprocess.Popen([sys.executable, "-m", "tests.oracle_jobs", name, out], cwd=ROOT, stdout=log, stderr=subprocess.STDOUT)
        _jobs[name] = (p, out, log)
    return _jobs[name]


def pytest_collection_modifyitems(config, items):
    """the tests that wait for a background oracle job run LAST (160x192x160 first, then the longer cfg-5 job), so the jobs
    overlap with every other GPU test"""
    def rank(it):
        n = it.name.split("[")[0]
        return 2 if n in _ORACLE_USERS["cfg5"] else (1 if n in _ORACLE_USERS["full160"] else 0)
    items.sort(key=rank)                               # stable: everything else keeps its order


def pytest_collection_finish(session):
    if session.config.option.collectonly:
        return
    import torch
    if not torch.cuda.is_available():                  # (the jobs need ~40 GB of host memory: GPU box only)
        return
    names = {it.name.split("[")[0] for it in session.items}
    for job, users in _ORACLE_USERS.items():           # longest first: cfg5 (~4 min), then 160x192x160 (~2 min)
        if names & set(users):
            _start_job(job)


def pytest_sessionfinish(session, exitstatus):
    for p, _, log in _jobs.values():
        if p.poll() is None:
            p.kill()
        log.close()


@pytest.fixture(scope="session")
def oracle_job():
    """oracle_job(name) -> the dict tests/oracle_jobs.py computed for 'full160' / 'cfg5' (waits for the background process)"""
    import torch
    cache = {}

    def get(name):
        if name not in cache:
            p, out, log = _start_job(name)
            rc = p.wait(timeout=1500)
            if rc != 0 or not os.path.exists(out):
                log.flush()
                raise RuntimeError("oracle job %s failed (rc %s): %s" % (name, rc, open(log.name).read()[-2000:]))
            cache[name] = torch.load(out, weights_only=False)
        return cache[name]
    return get
