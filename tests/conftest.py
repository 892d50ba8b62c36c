import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


EXCHANGE2 = None        # the two-process device-exchange job of tests/test_gpu_exchange2.py: {"procs": [...], "dir": path}


def _keyword_selects(kexpr, names):
    """Would `-k kexpr` keep a test whose keywords are `names`?  pytest's own expression grammar (substring match per identifier,
    not / and / or); when it cannot be evaluated here the answer is yes: starting the job for nothing costs two idle processes,
    not starting it lets a selected test skip silently."""
    try:
        from _pytest.mark.expression import Expression
        return bool(Expression.compile(kexpr).evaluate(lambda ident, **kw: any(ident.lower() in n.lower() for n in names)))
    except Exception:                                            # noqa: BLE001
        return True


def _start_exchange2(config):
    """Start the two ranks of tests/test_gpu_exchange2.py NOW: pytest_configure runs before any test module is imported, so
    this process has made no GPU call yet (torch.cuda.device_count() does not initialise the device on this image) -- a
    process that HAS initialised the GPU must not start other programs on this pool.  The children share cuda:0 and talk
    over a gloo group on 127.0.0.1; the test only reads their verdicts."""
    global EXCHANGE2
    expr = getattr(config.option, "markexpr", "") or ""
    if "not gpu" in expr or os.environ.get("KEDS_NO_EXCHANGE2") == "1":
        return
    args = [str(a) for a in config.args]
    if any(a.endswith(".py") or "::" in a for a in args) and not any("test_gpu_exchange2" in a for a in args):
        return                                                   # a run of other test files only
    kexpr = getattr(config.option, "keyword", "") or ""
    if kexpr and not _keyword_selects(kexpr, ("test_device_exchange_between_two_processes_equals_single_index", "test_gpu_exchange2.py",
                                              "test_gpu_exchange2", "tests", "gpu")):
        return                                                   # -k deselects this test (e.g. -k fp8); -k "not fullsize" keeps it
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:                                            # noqa: BLE001
        return
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="keds_exchange2_")
    port = "file://" + os.path.join(d, "rendezvous")             # a file in the job's own directory: no pre-probed TCP port to race for
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")                   # the container's hostname may not resolve: pair up over loopback
    procs = []
    for r in range(2):
        log = open(os.path.join(d, f"rank{r}.log"), "w")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "exchange2_worker.py"), str(r), "2", str(port), d],
                                      env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT))
    EXCHANGE2 = {"procs": procs, "dir": d}


def _stop_exchange2():
    """Reap the two ranks and remove their directory whatever happened to the test that reads their verdicts (deselected, -x
    stopped earlier, interrupted): nothing of this session outlives it."""
    global EXCHANGE2
    job, EXCHANGE2 = EXCHANGE2, None
    if job is None:
        return
    import shutil
    for p in job["procs"]:
        if p.poll() is None:
            p.kill()                                             # (the exact children this session started)
        try:
            p.wait(timeout=30)
        except Exception:                                        # noqa: BLE001
            pass
    shutil.rmtree(job["dir"], ignore_errors=True)


def pytest_unconfigure(config):
    _stop_exchange2()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _start_exchange2(config)


LIB_PROBLEM = None


def pytest_sessionstart(session):
    """The suite goes through libkeds_hip.so nearly everywhere (there is no CPU fallback): build it in-tree if a fresh
    checkout has not done so yet (hipcc cross-compiles without a GPU; on the GPU box the prebuilt file travels with the
    snapshot).  On a machine without ROCm the build fails: then only the tests that need the library are skipped -- the
    oracle-vs-golden tests are pure torch / numpy and still run."""
    global LIB_PROBLEM
    from keds_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        try:
            _lib.build()
        except Exception as e:                                   # no hipcc / build error: reported per skipped test
            LIB_PROBLEM = f"libkeds_hip.so is missing and could not be built: {str(e)[:200]}"


def pytest_collection_modifyitems(config, items):
    if LIB_PROBLEM is None:
        return
    skip = pytest.mark.skip(reason=LIB_PROBLEM)
    for item in items:
        if "test_oracle_golden" not in item.nodeid:
            item.add_marker(skip)


def golden_path(name):
    return os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(golden_path(name)))
    return load
