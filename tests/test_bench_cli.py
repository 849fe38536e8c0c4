"""bench.py's launcher contract, as far as it can be exercised without a GPU."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_n_without_n_devices_fails_instead_of_reporting_one_gpu():
    # `python bench.py --gpus 2` must start two ranks or fail: this container has no device, so the parent exits 3 before
    # anything is launched (it only counts devices, it never initialises one)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        return                                      # a real multi-GPU host: the run itself is the driver's business
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    assert "--gpus 2 but only" in r.stderr
    assert not r.stdout.strip()                     # no JSON line claiming n_gpus = 1


def test_permutation_count_of_the_m64_proof():
    b = _bench_module()
    n, N = 1 << 15, 1 << 18
    trees = sum(N * ((c + 7) // 8) + N - 16 for c in (135, 20, 16))
    fri = sum((1 << lg) * 5 - 16 for lg in (14, 10, 6))     # three arity-16 reduction rounds (k_fri_fold x 3 in the profiles): 32-element leaves
    assert b.permutations_per_proof(n) == trees + fri + N + (1 << 16)
    assert b.permutations_per_proof(n) == 6968544            # the figure roofline_prove quotes


def test_lanes_are_capped_by_the_cgroup_quota_not_the_affinity_mask(monkeypatch):
    # the 1-GPU box: 256 cores in the affinity mask, a 16-core CFS quota.  Eight ranks x 16 lanes would need ~38 cores of polling
    # threads: the cap is min(mask, quota) / world, never below 4
    b = _bench_module()
    monkeypatch.setattr(b, "host_cores", lambda: 256)
    monkeypatch.setattr(b, "cpu_quota_cores", lambda: 16.0)
    assert b.usable_cores() == 16.0
    per, caps = b.lanes_per_rank(1, {"streams": 16, "pool_lanes": 16})
    assert per == 16.0 and caps == {"streams": (16, 16), "pool_lanes": (16, 16)}
    per, caps = b.lanes_per_rank(8, {"streams": 16, "pool_lanes": 16})
    assert per == 2.0 and caps["streams"] == (16, 4) and caps["pool_lanes"] == (16, 4)
    monkeypatch.setattr(b, "cpu_quota_cores", lambda: None)          # no quota: the mask counts
    per, caps = b.lanes_per_rank(8, {"streams": 16})
    assert per == 32.0 and caps["streams"] == (16, 16)
    monkeypatch.setattr(b, "cpu_quota_cores", lambda: 128.0)         # an 8-GPU node with a 128-core quota
    per, caps = b.lanes_per_rank(8, {"streams": 16, "pool_lanes": 24})
    assert per == 16.0 and caps["streams"] == (16, 16) and caps["pool_lanes"] == (24, 16)


def test_one_thread_cpu_figure_is_the_committed_measurement():
    b = _bench_module()
    d = b.committed_one_thread(64)
    assert d and d["measured"] is True and 0.015 < d["value"] < 0.04 and "profiles/" in d["source"]
    assert b.committed_one_thread(20) is None
