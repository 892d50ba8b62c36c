"""-m gpu: the device-side sharded-search exchange between TWO real processes on one MI355X (the reference's
multi-GPU call is faiss.index_cpu_to_all_gpus, src/eval_retrieval.py:289-298).

The two ranks are children that tests/conftest.py starts at pytest_configure -- before this process has made any GPU call
(a process that has initialised the GPU must not start programs on this pool) -- and that run tests/exchange2_worker.py;
this test only collects their verdicts."""
import json
import os

import pytest

from tests import conftest

pytestmark = pytest.mark.gpu


def test_device_exchange_between_two_processes_equals_single_index():
    job = conftest.EXCHANGE2
    if job is None:
        pytest.skip("the two-process job was not started (no GPU visible at pytest_configure, or -m excludes gpu)")
    for p in job["procs"]:
        try:
            p.wait(timeout=420)
        except Exception:                                               # noqa: BLE001
            p.kill()
            raise
    logs = ""
    for r in range(2):
        with open(os.path.join(job["dir"], f"rank{r}.log")) as f:
            logs += f"--- rank {r} ---\n" + f.read()[-2000:]
    res = []
    for r in range(2):
        path = os.path.join(job["dir"], f"rank{r}.json")
        assert os.path.exists(path), "rank %d wrote no verdict\n%s" % (r, logs)
        res.append(json.load(open(path)))
    for r in res:
        assert r.get("ok"), json.dumps(r, indent=1) + "\n" + logs
        assert r["rows_from_other_shard"] > 0
    assert all(p.returncode == 0 for p in job["procs"]), logs
