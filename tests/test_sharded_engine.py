"""`_engine` on pooled counters: the counters-only aggregation reproduces the per-sim aggregation, and a
world-size-2 run over gloo (each rank simulating its slice, one all-reduce of the counter block) returns on every
rank exactly the single-process `find_power` / `find_sample_size` result (merge.rs:26-70)."""
import math
import os
import socket
import sys

import numpy as np
import pytest

from helpers import BASE_SEED, spec_c2b, spec_c3, spec_c4
from mcpower_amd import _engine as E, spec as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = {"boundary_hit", "estimator_extras"}


def _same(a, b):
    if isinstance(a, float) and isinstance(b, float):
        return (math.isnan(a) and math.isnan(b)) or a == b
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


def _posthoc_spec():
    from test_oracle_exclusion import three_level_posthoc_spec
    s = three_level_posthoc_spec()
    s.report_overall = True
    return s


@pytest.mark.parametrize("make", [spec_c2b, spec_c3, spec_c4, _posthoc_spec])
def test_counters_only_aggregation_equals_per_sim_aggregation(make, oracle):
    spec = make()
    n = {spec_c3: 500, spec_c4: 600}.get(make, 200)
    full = oracle.run_batch(spec, [n // 2, n], 300, BASE_SEED, n_threads=4)
    blocks = E._posthoc_blocks(spec)
    a = E.aggregate_batch(full, spec.target_indices, spec.contrast_pairs, spec.estimator, blocks)
    full.uncorrected = None
    b = E.aggregate_batch(full, spec.target_indices, spec.contrast_pairs, spec.estimator, blocks,
                          report_overall=bool(spec.report_overall))
    for x, y in zip(a, b):
        for k in x:
            if k not in SKIP:
                assert _same(x[k], y[k]), (k, x[k], y[k])
        ex, ey = x["estimator_extras"], y["estimator_extras"]
        assert ex["estimator"] == ey["estimator"]
        for k in ("singular_fit_rate", "joint_uncorrected_rate", "joint_corrected_rate", "boundary_hits"):
            if k in ex:
                assert _same(ex[k], ey[k]), k


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json
    import torch.distributed as dist
    from mcpower_amd import _engine as E
    from oracle import oracle
    from test_find_power import _contract, _pack

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run = E.sharded_runner(rank, world, run_batch=lambda spec, ns, cnt, seed, off: oracle.run_batch(
        spec, ns, cnt, seed, sim_offset=off, n_threads=2))
    blob = _pack(_contract())
    fp = E.find_power(blob, 120, 1001, 2137, None, _runner=run)
    fs = E.find_sample_size(blob, 80.0, 30, 150, 801, 2137, "grid", 6, "fixed", "linear", None, None, _runner=run)
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump({"fp": fp, "fs": fs}, f, sort_keys=True, default=str)
    dist.destroy_process_group()


def test_two_rank_find_power_and_find_sample_size_equal_single_process(tmp_path, oracle):
    import json
    import torch.multiprocessing as mp
    from test_find_power import _contract, _pack, _oracle_runner
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "r0.json"))
    r1 = json.load(open(tmp_path / "r1.json"))
    assert r0 == r1
    blob = _pack(_contract())
    fp = E.find_power(blob, 120, 1001, 2137, None, _runner=_oracle_runner(oracle))
    fs = E.find_sample_size(blob, 80.0, 30, 150, 801, 2137, "grid", 6, "fixed", "linear", None, None,
                            _runner=_oracle_runner(oracle))
    one = json.loads(json.dumps({"fp": fp, "fs": fs}, sort_keys=True, default=str))
    for kind in ("fp", "fs"):
        a, b = one[kind]["scenarios"]["optimistic"], r0[kind]["scenarios"]["optimistic"]
        for k in a:
            if k in ("boundary_hit", "estimator_extras"):
                continue
            assert a[k] == b[k], (kind, k)


@pytest.mark.gpu
def test_sharded_runner_on_the_device_equals_the_default_runner(engine):
    from test_find_power import _contract, _pack
    blob = _pack(_contract())
    a = E.find_power(blob, 150, 5000, 2137, None)["scenarios"]["optimistic"]
    b = E.find_power(blob, 150, 5000, 2137, None, _runner=E.sharded_runner(0, 1))["scenarios"]["optimistic"]
    for k in a:
        if k not in SKIP:
            assert _same(a[k], b[k]), k
