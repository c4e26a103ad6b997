"""-m gpu: bench.py itself. The driver's multi-GPU run is the first time `bench.py --gpus N` meets N devices, so the code
that run depends on is rehearsed here at world 1: an `nccl` process group of one rank, the NATIVE RCCL all-reduce hook
installed through the agreement round, the transport preflight (ranks_seen / PCI ids / latency in the JSON), the second
transport (peer-mapped exchange over shm) with its own timed loop, and every collective the timed legs issue."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rehearsal_of_the_n_rank_path_at_world_1():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse-distributed", "--steps", "5", "--warmup", "2",
                        "--nelem", "4000000", "--no-cpu-baseline", "--no-shard-leg", "--clock-spin-s", "0.05"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rehearsal"] and "native RCCL hook" in d["config"]["sharding"]
    r = d["rccl"]
    assert r["ranks_seen"] == 1 and r["user_rank"] == 0 and r["sum_check"] == "ok" and r["identical_bits"] is True
    assert len(r["pci_bus_ids"]) == 1 and set(r["latency_us"]) == {"8B", "320B", "6912B"}
    ps = d["transports"]["peer_shm"]
    assert "error" not in ps and ps["householder_ms_per_step"] > 0 and set(ps["latency_us"]) == {"8B", "320B", "6912B"}
    assert "timed loop" in d["phases_s"] and "transport preflight" in d["phases_s"]
    assert "lbfgs_error" not in d["extras"] and "cfg4_error" not in d["extras"]
    # rehearsal with more than one rank is refused (it is the world-1 form of the N-rank path)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-distributed", "--single-device", "--backend", "gloo",
                        "--nelem", "100000", "--steps", "2", "--no-extras", "--no-cpu-baseline", "--no-shard-leg"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert q.returncode != 0 and "--rehearse-distributed" in q.stderr


@pytest.mark.parametrize("fault", ["peer-leg", "peer-leg-stall"])
def test_a_failing_or_stalled_second_transport_leg_cannot_sink_the_headline(fault):
    """The peer-transport timed loop runs last and is optional: when it raises, or never returns (watchdog), rank 0 still
    prints the line measured over RCCL, with the reason under transports.peer_shm, and the exit status is 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MXLO_BENCH_FAULT=fault, MXLO_BENCH_OPTIONAL_LIMIT_S="3")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse-distributed", "--steps", "5", "--warmup", "2",
                        "--nelem", "4000000", "--no-extras", "--no-cpu-baseline", "--no-shard-leg", "--clock-spin-s", "0.05"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["rccl"]["ranks_seen"] == 1
    ps = d["transports"]["peer_shm"]
    assert ("injected failure" in ps["error"]) if fault == "peer-leg" else ("stalled" in ps["error"])
    assert "abandoned" in ps["note"] and "householder_ms_per_step" not in ps


@pytest.mark.parametrize("n", [2, 4, 8])
def test_n_ranks_on_one_device_over_the_debug_transport(n):
    """`bench.py --gpus N --single-device --backend gloo`, N = 2 / 4 / 8 (the driver's SCALE run, pre-flighted on the one
    device every box has — VERDICT r5 #8): the self-launcher starts N ranks of worker() on device 0 (RCCL refuses that, so
    the all-reduce is the torch.distributed hook over gloo): every collective of the N-rank path — operand normalisation, the
    sharded Householder / quasi-Newton legs (at --qn-nelem rows, so that 8 ranks share one device), max-over-ranks timing,
    the cfg5 single-GPU comparison, tear-down — runs with world N, the hook provably sums over all N ranks, and rank 0's ONE
    line reports n_gpus = N with the whole-job bandwidth."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--single-device", "--backend", "gloo", "--steps", "5",
                        "--warmup", "2", "--nelem", "4000000" if n == 2 else "1000000", "--qn-nelem", "50000000" if n == 2 else "400000",
                        "--no-cpu-baseline", "--no-shard-leg", "--clock-spin-s", "0.05"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["value"] > 0 and d["scaling"] == "weak" and "abandoned" not in d and d["complete"] is True
    assert "%d ranks" % n in d["config"]["sharding"] and "launcher" in d["config"]
    assert d["transports"]["debug_transport"]["ranks_seen"] == n
    assert "lbfgs_error" not in d["extras"] and "cfg4_error" not in d["extras"]
    assert d["extras"]["cfg5_LBFGS_fwd_m20_sharded"]["n_gpus"] == n
    assert ("qn_nelem_override" in d["extras"]) == (n != 2)


def test_a_rank_that_never_joins_an_extra_leg_cannot_sink_the_headline():
    """Two ranks; after the headline is measured the last rank never enters the quasi-Newton legs (injected). Rank 0 sits in
    their first collective until its watchdog fires (or the collective fails), then prints the line it has — headline,
    roofline, what was already measured — and every rank exits 0; the launcher relays the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MXLO_BENCH_FAULT="extras-stall")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo", "--steps", "5",
                        "--warmup", "2", "--nelem", "4000000", "--no-cpu-baseline", "--no-shard-leg", "--clock-spin-s", "0.05",
                        "--phase-timeout-scale", "0.03"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert "abandoned" in d or "lbfgs_error" in d["extras"] or "teardown_error" in d
