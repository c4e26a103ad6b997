#!/usr/bin/env python
"""bench.py — `mul!` hot-path throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A "step" is ONE 5-arg `mul!(res, H, v, α, β)` of `opHouseholder(h)` at n = 10^8 fp64 per GPU
(BASELINE.json configs[1]) with all operands resident in HBM. For N > 1 the vectors are
row-sharded (n per GPU: weak scaling) and h'v is all-reduced over RCCL through the library's
all-reduce hook; value = N * 40 B * n / max-over-ranks time.

N > 1 without an external launcher (`WORLD_SIZE` unset): this script LAUNCHES ITSELF — one rank per device
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set per child, rendezvous on 127.0.0.1), after checking that N devices
are visible; it exits non-zero with a message when they are not, when any rank fails, or when the native RCCL
all-reduce hook (libmxlo_rccl.so: ncclAllReduce issued from C on the ctx stream) cannot be installed on the `nccl`
backend — a run never degrades silently to one rank or to Python-issued collectives. Under
`python -m torch.distributed.run …` (WORLD_SIZE set) the ranks are used as launched.
A second leg in the same line (`extras.single_process_shard_abi`) drives the same N devices from ONE host process
through `mxlo_shard_ctx_create` + the `_sharded` entry points (the Julia deployment shape, include/mxlo_rccl.h); it
runs in a child process with a timeout so that it can never cost the headline line.

One JSON line is printed by rank 0 with, besides the driver contract fields:
  roofline      dominant kernel (the Householder update pass, 24 B/elt) timed with HIP events
                on the launch stream, against the 8 TB/s HBM3E peak
  cpu_baseline  the oracle (C restatement of the reference `mulHouseholder!`) on the host cores,
                1 thread, on a bounded sample of the same workload (rank 0, N == 1 only)
  extras        opDiagonal GB/s and (when built) InverseLBFGS / LBFGS apply/s — the other
                figures the metric string names
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md): 8.0 TB/s


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-max-s", type=float, default=8.0,
                    help="N = 1: upper bound on the untimed spin-up that waits for the step time to settle")
    ap.add_argument("--clock-spin-s", type=float, default=0.5,
                    help="untimed seconds of the same kernels before the warm-up steps, to reach steady GPU clocks")
    ap.add_argument("--nelem", dest="n", type=int, default=100_000_000, help="vector length per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes over a child process, ~25 s); "
                         "the profile-derived figure of profiles/traffic_householder.json is reported instead, labelled as such")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000)   # the GPU workload's own n
    ap.add_argument("--qn-nelem", type=int, default=50_000_000,
                    help="rows per GPU of the quasi-Newton legs (configs[2] / configs[4]: 5e7). A debugging aid like --nelem: the "
                         "N-rank tests on one device run the same legs, collectives included, at a size 8 ranks can share")
    # debugging aids for the N > 1 code path on a 1-GPU box: every rank on device 0, gloo instead of RCCL
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--single-device", action="store_true")
    ap.add_argument("--rehearse-distributed", action="store_true",
                    help="N = 1 only: run the N-rank code path at world 1 — nccl process group of one rank, the NATIVE RCCL hook, "
                         "transport preflight, the peer transport, every collective of the timed legs — so that the code the "
                         "driver's multi-GPU run depends on has executed on a one-GPU box (the numbers are those of one GPU)")
    # the N-device leg through the single-process shard ABI (runs in a child process with a timeout)
    ap.add_argument("--no-shard-leg", action="store_true")
    ap.add_argument("--shard-leg-timeout", type=float, default=420.0)
    ap.add_argument("--launch-timeout", type=float, default=1200.0,
                    help="self-launch: seconds before the ranks are stopped (kept below the driver's own 1800 s limit, so that a "
                         "stuck run ends HERE, with the phase every rank was in, and not as an anonymous kill)")
    ap.add_argument("--phase-timeout-scale", type=float, default=1.0,
                    help="multiplies the per-phase watchdog limits of a rank (PHASE_LIMITS_S)")
    ap.add_argument("--preflight-timeout-ms", type=int, default=60000, help="bound on every wait of the transport preflight")
    # internal / test aids
    ap.add_argument("--role", default="auto", choices=["auto", "shard-leg"], help=argparse.SUPPRESS)
    ap.add_argument("--worker-cmd", default=None,
                    help="self-launch: JSON list, the command to run per rank instead of this script (launcher tests)")
    return ap.parse_args(argv)


def visible_device_count() -> int:
    """Devices this process can see (MXLO_BENCH_FAKE_DEVICE_COUNT: launcher tests on a box without GPUs)."""
    fake = os.environ.get("MXLO_BENCH_FAKE_DEVICE_COUNT")
    if fake is not None:
        return int(fake)
    import torch
    return torch.cuda.device_count()


# The contract is ONE JSON line on stdout. Libraries under this process write there too (RCCL prints a version banner
# with printf when a communicator is created, and libc flushes it at exit — AFTER the line): main() therefore keeps a
# private duplicate of the original stdout for the line and points descriptor 1 at stderr for everybody else.
_JSON_FD = None


def claim_stdout():
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


_EMIT_LOCK = __import__("threading").Lock()
_EMITTED = False


def emit_json(obj):
    """The ONE line. The watchdog thread's rescue and the main thread can both get here (a stall that ends while the rescue
    is printing): the first caller wins, under a lock, and serialises a snapshot; later calls are no-ops."""
    global _EMITTED
    with _EMIT_LOCK:
        if _EMITTED:
            return False
        _EMITTED = True
        for attempt in range(3):                     # the other thread may be adding keys: retry the snapshot
            try:
                data = (json.dumps(obj) + "\n").encode()
                break
            except RuntimeError:                     # "dictionary changed size during iteration"
                if attempt == 2:
                    raise
                time.sleep(0.01)
        _emit_bytes(data)
        return True


def _emit_bytes(data):
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
        return
    sys.stdout.flush()
    while data:
        data = data[os.write(_JSON_FD, data):]


def die(msg: str, code: int = 2):
    print(f"bench.py: {msg}", file=sys.stderr, flush=True)
    raise SystemExit(code)


# Per-phase limits of one rank (seconds). A rank that sits in one phase for longer prints which phase (and which rank) to
# stderr and exits with status 7; the launcher then stops the other ranks and reports every rank's last phase.
PHASE_LIMITS_S = {
    "import + library load": 300, "init_process_group": 300, "communicator creation": 240, "transport preflight": 240,
    "operand set-up": 300, "clock spin-up": 300, "warm-up steps": 300, "timed loop": 300, "per-kernel timing": 300,
    "quasi-Newton legs": 900, "cfg4 legs": 300, "misc legs": 600, "cpu baseline": 300, "pmc traffic": 400, "second transport (optional)": 150, "teardown": 120,
}


class Watchdog:
    """`with wd.phase("timed loop"): ...` — a daemon thread ends the process when a phase outlasts its limit. The
    current phase is mirrored into $MXLO_BENCH_PHASE_DIR/rank<k> (when set by the launcher) for the post-mortem."""

    def __init__(self, rank: int, scale: float = 1.0, limits=None, exit_fn=None):
        import threading
        self.rank, self.scale = rank, scale
        self.limits = dict(PHASE_LIMITS_S if limits is None else limits)
        self.exit_fn = exit_fn or (lambda code: os._exit(code))
        self.cur, self.t0 = None, 0.0
        self.history = []
        self.rescue = None              # set around an OPTIONAL phase: called instead of exit_fn when that phase stalls
        self.lock = threading.Lock()
        self.dir = os.environ.get("MXLO_BENCH_PHASE_DIR")
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _note(self, text):
        if self.dir:
            try:
                with open(os.path.join(self.dir, "rank%d" % self.rank), "w") as f:
                    f.write(text)
            except OSError:
                pass

    def _run(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                cur, t0 = self.cur, self.t0
            if cur is None:
                continue
            limit = self.limits.get(cur, 300) * self.scale
            if time.time() - t0 > limit:
                print(f"bench.py: WATCHDOG rank {self.rank} has been in phase '{cur}' for {time.time() - t0:.0f} s "
                      f"(limit {limit:.0f} s) — stalled; exiting with status 7", file=sys.stderr, flush=True)
                self._note(f"STALLED in '{cur}' after {time.time() - t0:.0f} s")
                if self.rescue is not None:
                    self.rescue(f"rank {self.rank} stalled in '{cur}' for {time.time() - t0:.0f} s (limit {limit:.0f} s)")
                self.exit_fn(7)
                return

    def phase(self, name):
        wd = self

        class _P:
            def __enter__(self_):
                with wd.lock:
                    wd.cur, wd.t0 = name, time.time()
                wd._note(f"in '{name}' since {time.strftime('%H:%M:%S')}")

            def __exit__(self_, et, ev, tb):
                with wd.lock:
                    wd.history.append((name, round(time.time() - wd.t0, 3)))
                    wd.cur = None
                wd._note(f"finished '{name}'" if et is None else f"FAILED in '{name}': {ev!r}"[:300])
                return False
        return _P()


def check_topology(args, ndev: int):
    """Refuse, loudly, every configuration that could only run degraded."""
    if args.gpus < 1:
        die("--gpus must be >= 1")
    if args.single_device:
        if args.gpus > 1 and args.backend == "nccl":
            die("--single-device puts every rank on device 0, which RCCL refuses: add --backend gloo (debug transport)")
        if ndev < 1:
            die("no HIP device visible")
    elif ndev < args.gpus:
        die(f"--gpus {args.gpus} but only {ndev} HIP device(s) visible: one rank per device is required "
            "(use --single-device --backend gloo to exercise the N-rank code path on one GPU)")


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def last_json_line(text: str):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def launch(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE: start N ranks of this script, one per device, and relay rank 0's
    JSON line. Any rank failing (or the time limit) stops the others (exact PIDs) and the exit status is non-zero."""
    import subprocess
    ndev = visible_device_count()
    check_topology(args, ndev)
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT") or free_port())
    cmd = json.loads(args.worker_cmd) if args.worker_cmd else [sys.executable, os.path.abspath(__file__)] + list(argv)
    import tempfile
    phase_dir = tempfile.mkdtemp(prefix="mxlo_bench_phase_")

    def last_phases():
        out = []
        for r in range(n):
            try:
                out.append("rank %d: %s" % (r, open(os.path.join(phase_dir, "rank%d" % r)).read().strip()))
            except OSError:
                out.append("rank %d: (no phase recorded)" % r)
        return "; ".join(out)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MXLO_BENCH_SELF_LAUNCHED="1", MXLO_BENCH_PHASE_DIR=phase_dir)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading
    out0 = []
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    deadline, bad = time.time() + args.launch_timeout, None
    while bad is None:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if failed:
            bad = "rank %d exited with status %d" % failed[0]
        elif time.time() > deadline:
            bad = f"ranks still running after --launch-timeout {args.launch_timeout:.0f} s"
        else:
            time.sleep(0.05)
    if bad is None:
        failed = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
        if failed:
            bad = "rank %d exited with status %d" % failed[0]
    if bad is not None:
        for p in procs:                                   # only the processes started above, by PID
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except Exception:
                p.kill()
        where = last_phases()
        import shutil
        shutil.rmtree(phase_dir, ignore_errors=True)
        die(f"{n}-rank launch failed: {bad}. Last phase of every rank — {where}", 3)
    rd.join(10)
    import shutil
    shutil.rmtree(phase_dir, ignore_errors=True)
    line = last_json_line(out0[0] if out0 else "")
    if line is None:
        die("rank 0 printed no JSON line", 4)
    if line.get("n_gpus") != n:
        die(f"rank 0 reports n_gpus={line.get('n_gpus')} for a {n}-rank launch", 4)
    line.setdefault("config", {})["launcher"] = "bench.py self-launch: %d processes, one per device, rendezvous 127.0.0.1:%d" % (n, port)
    return line


def run_shard_leg(args) -> dict:
    """The N-device leg through the single-process shard ABI, in a child process with a time limit."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--role", "shard-leg", "--gpus", str(args.gpus), "--nelem", str(args.n),
           "--steps", str(min(args.steps, 100)), "--warmup", str(args.warmup)]
    if args.single_device:
        cmd.append("--single-device")
    if args.no_extras:
        cmd.append("--no-extras")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                                            "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                            "TORCHELASTIC_RUN_ID")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.time()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        so, se = p.communicate(timeout=args.shard_leg_timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        p.communicate()
        return {"error": f"no result within {args.shard_leg_timeout:.0f} s (child stopped)"}
    got = last_json_line(so or "")
    if p.returncode != 0 or got is None:
        return {"error": f"child exited with status {p.returncode}", "stderr_tail": (se or "")[-400:]}
    got["wall_s"] = round(time.time() - t0, 1)
    return got


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    claim_stdout()
    if args.role == "shard-leg":
        emit_json(shard_leg(args))
        return
    self_launch = args.gpus > 1 and "WORLD_SIZE" not in os.environ
    if self_launch:
        out = launch(args, argv)                     # N children run worker(); rank 0's line comes back
    else:
        out = worker(args)
    if out is None:                                  # a non-zero rank under an external launcher
        return
    # the shard-ABI leg runs ONCE, after the process-per-GPU ranks are gone: in the launcher when this script started the
    # ranks itself (its rank-0 child skips it), in rank 0 under an external launcher
    if not args.no_shard_leg and not args.worker_cmd and (self_launch or os.environ.get("MXLO_BENCH_SELF_LAUNCHED") != "1"):
        out.setdefault("extras", {})["single_process_shard_abi"] = run_shard_leg(args)
    out.setdefault("complete", "abandoned" not in out)   # false: printed by a rescue (see `abandon`); every leg ran otherwise
    emit_json(out)


def worker(args):
    """One rank: returns the JSON object on rank 0, None elsewhere."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wd = Watchdog(rank, args.phase_timeout_scale)
    if os.environ.get("MXLO_BENCH_OPTIONAL_LIMIT_S"):          # test aid: the limit of the optional leg alone
        wd.limits["second transport (optional)"] = float(os.environ["MXLO_BENCH_OPTIONAL_LIMIT_S"]) / max(args.phase_timeout_scale, 1e-9)
    with wd.phase("import + library load"):
        import torch
        import torch.distributed as dist

        import __graft_entry__ as g
        lo = g.load_package()
        from linearoperators_jl_amd import _lib
        from linearoperators_jl_amd.device import Timer, dtype_code, get_ctx, ptr
    distributed = world > 1 or args.rehearse_distributed
    if args.rehearse_distributed and world != 1:
        die("--rehearse-distributed is the world-1 rehearsal of the N-rank path: use it with --gpus 1")
    if args.gpus != world:
        die(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly one rank per requested GPU")
    check_topology(args, visible_device_count())
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_distributed:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        with wd.phase("init_process_group"):
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world)
    ctx = get_ctx(dev)
    hook = None
    native = args.backend == "nccl" and not args.single_device

    def install_hook():
        nonlocal hook
        if not distributed:
            return
        if native and hook is None:
            # libmxlo_rccl.so: ncclAllReduce issued from C on the ctx stream. REQUIRED on the nccl backend: if any rank
            # cannot build its communicator every rank raises (the run exits non-zero; no Python-issued collectives)
            hook = lo.sharded.install_agreed_allreduce(ctx, require_native=True, even_at_world_1=args.rehearse_distributed)
            return
        if native:
            hook.install(ctx)
            return
        lo.sharded.install_allreduce(ctx, native=False)   # Python hook over torch.distributed (gloo debugging only)

    with wd.phase("communicator creation"):
        install_hook()                         # RCCL all-reduce of the partial dots over xGMI

    # ---- the transport proves itself BEFORE anything is timed: every rank all-reduces known 8 B /
    # 320 B / 6912 B payloads through the hook the applies use, checks the sums, checks that all ranks hold identical bits,
    # agrees on the verdict, and records what the communicator reports about itself. A failure ends the run non-zero,
    # naming phase and rank; every wait is bounded.
    transports = {}
    peer_hook = None
    with wd.phase("transport preflight"):
        try:
            transports = transport_preflight(args, lo, torch, dist, ctx, dev, rank, world, hook if native else None)
        except Exception as e:
            if distributed:
                die(f"rank {rank}: transport preflight failed: {e}", 6)
            transports = {"error": repr(e)[:300]}       # N = 1: informational only (no collective is on the timed path)
        if distributed and native and (not args.no_extras or args.rehearse_distributed):
            try:   # the second transport: peer-mapped one-shot exchange (csrc/peer.hip), measured below next to RCCL
                peer_hook = lo.sharded.PeerShmHook(rank, world, timeout_ms=args.preflight_timeout_ms)
                transports["peer_shm"] = {"latency_us": peer_hook.preflight(ctx.stream, 50, args.preflight_timeout_ms),
                                          "mailboxes": "POSIX shm segment registered with every rank's HIP runtime; one kernel per collective"}
            except Exception as e:
                transports["peer_shm"] = {"error": repr(e)[:300]}
                peer_hook = None
            ok = torch.tensor([1 if peer_hook is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                peer_hook = None                        # never a mix of transports

    n = args.n
    ph = wd.phase("operand set-up")
    ph.__enter__()
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    h = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5
    nrm2 = (h * h).sum()
    if distributed:
        dist.all_reduce(nrm2)
    h /= nrm2.sqrt()                                    # ||h||_2 = 1 over the whole (sharded) vector
    v = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    res = torch.empty(n, dtype=torch.float64, device=dev)
    H = lo.opHouseholder(h)
    alpha, beta = 1.0, 0.0
    ph.__exit__(None, None, None)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # The GPU idles at a few hundred MHz (sclk 525 MHz at rest) and needs tens of milliseconds of load to reach its
    # steady clocks; W = 10 warm-up steps are only 6 ms. Bring the clocks up first (untimed, same kernels), then do
    # the W warm-up steps and time EXACTLY K steps as the contract says.
    def spin_up():
        if distributed:
            # every apply contains a collective: all ranks must issue the SAME number of them, so the spin is a fixed count
            # (a time-based loop would let ranks disagree and hang the all-reduce)
            # (the debug transport — a host-synchronising Python hook over gloo, several ranks sharing one device — pays
            #  milliseconds per collective: a handful of applies there, the count below is sized for the native hook)
            for _ in range(min(5000, max(20, int(args.clock_spin_s / (0.7e-3 * max(n / 1e8, 1e-3))))) if native else 20):
                lo.mul(res, H, v, alpha, beta)
            torch.cuda.synchronize()
        else:
            # ... and until the step time has SETTLED (bounded): a bench started right after another process released tens
            # of GB (the driver runs it after the test suite) shares HBM with the driver's scrubbing of the freed memory for
            # the first seconds — measured: 6306 / 6445 GB/s in the first run after pytest against 6690-6750 in every later
            # one. Batches of 20 untimed applies until the last three agree within 1.5 % (at most --settle-max-s seconds).
            t_spin = time.perf_counter()
            recent = []
            while True:
                tb = time.perf_counter()
                for _ in range(20):
                    lo.mul(res, H, v, alpha, beta)
                torch.cuda.synchronize()
                recent = (recent + [time.perf_counter() - tb])[-3:]
                if os.environ.get("MXLO_BENCH_DEBUG_SETTLE"):
                    print("settle: t=%.2f s batch=%.3f ms/apply" % (time.perf_counter() - t_spin, recent[-1] / 20 * 1e3), file=sys.stderr, flush=True)
                el = time.perf_counter() - t_spin
                if el >= args.clock_spin_s and ((len(recent) == 3 and max(recent) <= 1.015 * min(recent)) or el >= args.settle_max_s):
                    break

    with wd.phase("clock spin-up"):
        spin_up()
    with wd.phase("warm-up steps"):
        for _ in range(args.warmup):
            lo.mul(res, H, v, alpha, beta)
        barrier()
    with wd.phase("timed loop"):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lo.mul(res, H, v, alpha, beta)
        barrier()
        dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    bytes_per_step = 40.0 * n * world                   # 16 B/elt dot pass + 24 B/elt update pass
    value = bytes_per_step / (dt / args.steps) / 1e9

    # ---- per-kernel timing with HIP events on the launch stream (rank-local)
    wd_k = wd.phase("per-kernel timing")
    wd_k.__enter__()
    tm = Timer(ctx)
    f64 = dtype_code(torch.float64)
    K = max(10, min(args.steps, 50))
    dot_dev = torch.zeros(1, dtype=torch.float64, device=dev)
    if distributed:                                      # time the kernels, not the collective
        ctx.set_allreduce(None)
    tm.start()
    for _ in range(K):
        _lib.call("mxlo_dot", ctx.handle, f64, ptr(h), ptr(v), n, ptr(dot_dev))
    tm.stop()
    ms_dot = tm.elapsed_ms() / K
    tm.start()
    for _ in range(K):
        _lib.call("mxlo_householder_apply", ctx.handle, f64, ptr(res), ptr(h), ptr(v), n, alpha, beta, 0, ptr(dot_dev))
    tm.stop()
    ms_upd = tm.elapsed_ms() / K
    D = lo.opDiagonal(h)
    for _ in range(3):
        lo.mul(res, D, v, alpha, beta)
    tm.start()
    for _ in range(K):
        lo.mul(res, D, v, alpha, beta)
    tm.stop()
    ms_diag = tm.elapsed_ms() / K
    install_hook()
    wd_k.__exit__(None, None, None)

    upd_gbs = 24.0 * n / (ms_upd * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_householder.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("householder_update_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "map_kernel<HouseholderOp> (update pass res = α(v - c·h), 24 B/elt)",
                "achieved": round(upd_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(upd_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": ("profile-derived, NOT measured in this run: HBM bytes per launch from separate rocprofv3 --pmc "
                                   "FETCH_SIZE / WRITE_SIZE passes over the same kernel and size (tools/profile_gpu.sh -> "
                                   "profiles/traffic_householder.json, gfx950 unit corrections applied there)") if traffic else None,
                "avg_launch_ms": round(ms_upd, 4), "algorithmic_bytes_per_launch": 24.0 * n}
    extras = {
        "householder_dot_pass": {"ms": round(ms_dot, 4), "GB/s": round(16.0 * n / (ms_dot * 1e-3) / 1e9, 1),
                                 "frac_hbm_peak": round(16.0 * n / (ms_dot * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "householder_mul_frac_hbm_peak_per_gpu": round(value / world / HBM_PEAK_GBS, 4),
        "opDiagonal_mul": {"ms": round(ms_diag, 4), "GB/s": round(24.0 * n / (ms_diag * 1e-3) / 1e9, 1),
                           "frac_hbm_peak": round(24.0 * n / (ms_diag * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    del D

    out = None
    if rank == 0:
        transport = ("native RCCL hook (libmxlo_rccl.so, ncclAllReduce on the ctx stream)" if native
                     else "torch.distributed Python hook over %s (debug transport)" % args.backend)
        out = {
            "metric": "mul! GB/s (frac HBM peak) at n=10^8 fp64; L-BFGS apply/s, 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "opHouseholder(h) 5-arg mul!(res,H,v,1,0), n=%d fp64 per GPU (configs[1])" % n,
                       "n_per_gpu": n, "algorithmic_bytes_per_elt": 40,
                       "sharding": ("row ranges over %d ranks (one process per GPU), 1-double all-reduce per apply: %s" % (world, transport)) if distributed else "none",
                       "devices_visible": torch.cuda.device_count()},
            "frac_hbm_peak": round(value / world / HBM_PEAK_GBS, 4),
            "roofline": roofline, "cpu_baseline": None, "extras": extras,
            "rccl": transports.get("rccl"), "transports": transports,
            "rehearsal": ("world-1 rehearsal of the N-rank code path (nccl group of one rank, native hook, preflight, peer transport)"
                          if args.rehearse_distributed else None),
            "phases_s": {k: v for k, v in wd.history},
        }

    # From here on the headline is measured. Every later leg is an extra: if one of them stalls (a collective a rank never
    # joins, a transport that dies) the watchdog calls `abandon` instead of ending the run with status 7 — rank 0 prints
    # the line it has, with the reason under "abandoned", and every rank leaves with status 0.
    def abandon(why, key="abandoned"):
        if rank == 0:
            out["complete"] = False                      # a partial line: some leg after the headline did not finish
            if key == "abandoned":
                out["abandoned"] = str(why)[:300]
            else:
                out["transports"].setdefault(key, {})["error"] = str(why)[:300]
                out["transports"][key]["note"] = ("optional second-transport leg abandoned; every other figure of this "
                                                  "line was measured before it, over RCCL")
            out["phases_s"] = {k: v for k, v in wd.history}
            emit_json(out)
            if peer_hook is not None:                    # the tear-down is skipped: at least do not leave the segment's name behind
                try:
                    os.unlink("/dev/shm" + peer_hook.name)
                except OSError:
                    pass
        sys.stderr.flush()
        os._exit(0)
    if distributed:
        wd.rescue = abandon

    # ---- quasi-Newton apply/s (the second figure of the metric string)
    if not args.no_extras and hasattr(lo, "InverseLBFGSOperator"):
        with wd.phase("quasi-Newton legs"):
            if os.environ.get("MXLO_BENCH_FAULT") == "extras-stall" and rank == world - 1 and world > 1:
                time.sleep(1e6)                          # TEST HOOK: the last rank never joins the extras' collectives
            try:
                extras.update(bench_lbfgs(lo, torch, dev, ctx, rank, world, distributed, dist, barrier, n=args.qn_nelem))
            except Exception as e:  # never lose the headline line
                extras["lbfgs_error"] = repr(e)
            install_hook()                                   # rank 0 cleared it for the single-GPU cfg5 leg

    if not args.no_extras:
        with wd.phase("cfg4 legs"):
            try:
                extras.update(bench_cfg4(lo, torch, dev, ctx))
            except Exception as e:
                extras["cfg4_error"] = repr(e)

    if not args.no_extras and rank == 0 and world == 1:
        with wd.phase("misc legs"):
            try:
                extras.update(bench_misc(lo, torch, dev, ctx))
            except Exception as e:
                extras["misc_error"] = repr(e)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        with wd.phase("cpu baseline"):
            # parity AT THE BENCHMARK'S OWN SIZE beside the throughput: the CPU legs run the oracle on the
            # device's operands, so what they compute is compared with the device result instead of being thrown away
            def fresh_apply():
                res.fill_(float("nan"))
                lo.mul(res, H, v, alpha, beta)
                return res
            out["parity"] = {}
            out["cpu_baseline"], par = cpu_leg(args.cpu_sample, (h, v, fresh_apply) if (alpha, beta) == (1.0, 0.0) else None)
            if par is not None:
                out["parity"]["householder_n1e8" if n == 100_000_000 else "householder_n%d" % n] = par
            # the reference's CPU mul! beside the OTHER two figures of the metric string, in the same run
            if not args.no_extras:
                try:
                    if "InverseLBFGS_m10_n5e7" in extras:
                        extras["InverseLBFGS_m10_n5e7"]["cpu"], par = cpu_leg_lbfgs(gpu=(lo, torch, dev))
                        if par is not None:
                            out["parity"]["lbfgs_inv_n5e6"] = par
                    if "kron_1024x1024" in extras:
                        extras["kron_1024x1024"]["cpu"] = cpu_leg_kron()
                except Exception as e:
                    extras["cpu_legs_error"] = repr(e)[:200]

    if rank == 0 and world == 1 and not args.no_traffic and not args.no_extras:
        with wd.phase("pmc traffic"):
            t_meas, how = measure_traffic(n)
            if t_meas is not None:
                out["roofline"]["traffic_profile_derived"] = out["roofline"]["traffic"]
                out["roofline"]["traffic"] = round(t_meas, 1)
                out["roofline"]["traffic_source"] = how
                out["roofline"]["traffic_over_algorithmic"] = round(t_meas / (24.0 * n), 5)
            else:
                out["roofline"]["traffic_measure_error"] = how

    # ---- the same K steps under the SECOND transport (peer-mapped one-shot exchange), N > 1 only: reported next to the
    # RCCL headline, never as `value`. It runs LAST and is optional: everything above is already in `out`, so a failure or
    # a stall in here abandons this leg only — rank 0 prints the line it has (with the reason under transports.peer_shm)
    # and every rank leaves with status 0, skipping the tear-down collectives a half-dead transport could hang.
    if peer_hook is not None:
        wd.rescue = lambda why: abandon(why, "peer_shm")
        try:
            with wd.phase("second transport (optional)"):
                peer_hook.install(ctx)
                if os.environ.get("MXLO_BENCH_FAULT") == "peer-leg":      # TEST HOOK: the optional leg fails
                    raise RuntimeError("injected failure of the optional second-transport leg")
                if os.environ.get("MXLO_BENCH_FAULT") == "peer-leg-stall":  # TEST HOOK: the optional leg never returns
                    time.sleep(1e6)
                for _ in range(max(5, args.warmup)):
                    lo.mul(res, H, v, alpha, beta)
                barrier()
                tp0 = time.perf_counter()
                for _ in range(args.steps):
                    lo.mul(res, H, v, alpha, beta)
                barrier()
                tp = torch.tensor([time.perf_counter() - tp0], dtype=torch.float64, device=dev)
                dist.all_reduce(tp, op=dist.ReduceOp.MAX)
                peer_hook.check()
                if rank == 0:
                    out["transports"]["peer_shm"]["householder_ms_per_step"] = round(float(tp.item()) / args.steps * 1e3, 4)
                    out["transports"]["peer_shm"]["householder_GB/s"] = round(bytes_per_step / (float(tp.item()) / args.steps) / 1e9, 1)
                    out["transports"].setdefault("rccl", {})["householder_ms_per_step"] = round(ms_per_step, 4)
        except Exception as e:                           # (SystemExit / KeyboardInterrupt end the process as they should)
            abandon(f"rank {rank}: {e!r}", "peer_shm")
        wd.rescue = abandon if distributed else None
    if rank == 0:
        out["phases_s"] = {k: v for k, v in wd.history}
    with wd.phase("teardown"):
        try:                                             # best effort: the line is complete, a peer that is already gone
            del H, h, v, res                             # (its own rescue, a crash) must not turn it into a failure
            torch.cuda.synchronize()
            if distributed:
                ctx.set_allreduce(None)                  # the communicator itself is released at process exit
            if peer_hook is not None:
                peer_hook.close()
            torch.cuda.empty_cache()
            if distributed:
                dist.barrier()
                dist.destroy_process_group()
        except Exception as e:
            if rank == 0:
                out["teardown_error"] = repr(e)[:300]
    return out


def transport_preflight(args, lo, torch, dist, ctx, dev, rank, world, native_hook):
    """What the all-reduce transport says about itself, and its measured latency, BEFORE timing.
    N > 1 on the nccl backend: through the NATIVE hook the applies use (libmxlo_rccl.so): ncclCommCount / UserRank /
    CuDevice + PCI bus id of every rank (gathered), known-answer and identical-bits all-reduces of 8 B / 320 B / 6912 B,
    latency in us. N == 1: the same functions at world 1 (a communicator of one rank, the peer exchange with one mailbox):
    nothing on the timed path uses them — recorded so that the code path the N > 1 run depends on has run on this box."""
    out = {}
    if world > 1 or args.rehearse_distributed:
        if native_hook is None:
            # the debug transport proves the one thing a line with n_gpus = N has to be able to show: the hook the applies
            # use sums over ALL N ranks — a dot of one 1.0 per rank through the installed hook must come back as N
            from linearoperators_jl_amd import _lib
            from linearoperators_jl_amd.device import dtype_code, ptr
            one = torch.ones(1, dtype=torch.float64, device=dev)
            got = torch.zeros(1, dtype=torch.float64, device=dev)
            _lib.call("mxlo_dot", ctx.handle, dtype_code(torch.float64), ptr(one), ptr(one), 1, ptr(got))
            torch.cuda.synchronize()
            seen = int(round(float(got.item())))
            if seen != world:
                raise RuntimeError(f"the all-reduce hook summed {seen} ranks in a {world}-rank run")
            return {"rccl": None, "debug_transport": {"ranks_seen": seen, "check": "mxlo_dot of one 1.0 per rank through the installed hook"},
                    "note": "debug transport (torch.distributed Python hook): no native preflight"}
        info = native_hook.info()
        lat = native_hook.preflight(ctx.stream, 50, args.preflight_timeout_ms)
        infos = [None] * world
        dist.all_gather_object(infos, info)
        pcis = [i["pci_bus_id"] for i in infos]
        if info["ranks_seen"] != world or sorted(i["user_rank"] for i in infos) != list(range(world)):
            raise RuntimeError(f"the communicator reports {info['ranks_seen']} ranks / user ranks {[i['user_rank'] for i in infos]} for a {world}-rank run")
        if len(set(pcis)) != world and not args.single_device:
            raise RuntimeError(f"{world} ranks but only {len(set(pcis))} distinct devices (PCI bus ids {pcis})")
        out["rccl"] = {"ranks_seen": info["ranks_seen"], "user_rank": info["user_rank"], "devices": [i["device"] for i in infos],
                       "pci_bus_ids": pcis, "latency_us": lat, "sum_check": "ok", "identical_bits": True,
                       "hook": "mxlo_rccl_allreduce_hook (ncclAllReduce on the ctx stream)"}
        return out
    if args.no_extras:
        return out
    t0 = time.perf_counter()
    hook = lo.sharded.NativeRcclHook(0, 1)
    try:
        info = hook.info()
        out["rccl"] = {"ranks_seen": info["ranks_seen"], "user_rank": info["user_rank"], "devices": [info["device"]],
                       "pci_bus_ids": [info["pci_bus_id"]], "latency_us": hook.preflight(ctx.stream, 50, args.preflight_timeout_ms),
                       "sum_check": "ok", "identical_bits": True, "note": "world 1: a communicator of one rank (no fabric hop)"}
    finally:
        hook.close()
    ph = lo.sharded.PeerShmHook(0, 1, timeout_ms=args.preflight_timeout_ms)
    try:
        out["peer_shm"] = {"latency_us": ph.preflight(ctx.stream, 50, args.preflight_timeout_ms), "note": "world 1: one mailbox"}
    finally:
        ph.close()
    out["wall_s"] = round(time.perf_counter() - t0, 2)
    return out


def shard_leg(args) -> dict:
    """ONE host process driving `--gpus` devices through the single-process shard ABI of include/mxlo_rccl.h
    (`mxlo_shard_ctx_create`: per device a stream, a ctx, an RCCL communicator from ncclCommInitAll and a worker thread;
    the `_sharded` entry points take one device pointer per shard) — the deployment shape of a Julia host. Same workloads
    as the process-per-GPU legs: opHouseholder n per device (weak) and LBFGSOperator m = 20 with n_local = 5e7 per device.
    torch is used for device memory and random numbers only."""
    import torch

    import __graft_entry__ as g
    lo = g.load_package()
    from linearoperators_jl_amd import _lib
    nd = args.gpus
    check_topology(argparse.Namespace(gpus=nd, single_device=args.single_device, backend="gloo"), visible_device_count())
    R = _lib.rccl_lib()
    ids = [0] * nd if args.single_device else list(range(nd))
    sctx = C.c_void_p()
    if R.mxlo_shard_ctx_create(nd, (C.c_int32 * nd)(*ids), C.byref(sctx)) != 0:
        die("mxlo_shard_ctx_create: " + (R.mxlo_shard_last_error() or b"").decode(), 5)
    devs = [torch.device("cuda", i) for i in ids]
    F64 = _lib.F64

    def ck(rc, what):
        if rc != 0:
            die(f"{what}: " + (R.mxlo_shard_last_error() or b"").decode(), 5)

    def ptrs(ts):
        return (C.c_void_p * nd)(*[t.data_ptr() for t in ts])

    def sync_torch():
        for d in set(devs):
            torch.cuda.synchronize(d)

    def timed(fn, reps, spin_s):
        sc = sctx
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < spin_s:
            for _ in range(5):
                fn()
            ck(R.mxlo_shard_ctx_sync(sc), "sync")
        ck(R.mxlo_shard_ctx_sync(sc), "sync")
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ck(R.mxlo_shard_ctx_sync(sc), "sync")
        return (time.perf_counter() - t0) / reps

    out = {"host": "one process, %d shard(s): mxlo_shard_ctx_create + _sharded entry points" % nd,
           "transport": "loopback (all shards on device 0: debug)" if R.mxlo_shard_ctx_is_loopback(sctx) else "ncclCommInitAll, one worker thread per device",
           "n_gpus": nd}

    def describe(sc):
        """preflight (sums, identical bits, verdict, latency) + what every shard runs on, before anything is timed"""
        lat = (C.c_double * 3)()
        ck(R.mxlo_shard_ctx_preflight(sc, 50, args.preflight_timeout_ms, lat), "preflight")
        seen, pcis = [], []
        for i in range(nd):
            dv, sn, pci = C.c_int32(-1), C.c_int32(-1), C.create_string_buffer(64)
            ck(R.mxlo_shard_ctx_info(sc, i, C.byref(dv), C.byref(sn), pci, 64), "info")
            seen.append(sn.value)
            pcis.append(pci.value.decode())
        return {"ranks_seen": seen[0], "pci_bus_ids": pcis, "sum_check": "ok", "identical_bits": True,
                "latency_us": {"8B": round(lat[0], 2), "320B": round(lat[1], 2), "6912B": round(lat[2], 2)}}
    out["preflight"] = describe(sctx)
    n = args.n
    gens = [torch.Generator(device=d).manual_seed(77 + i) for i, d in enumerate(devs)]
    hs = [torch.rand(n, dtype=torch.float64, device=d, generator=gq) - 0.5 for d, gq in zip(devs, gens)]
    nrm = sum(float((t * t).sum().item()) for t in hs) ** 0.5
    for t in hs:
        t /= nrm
    vs = [torch.rand(n, dtype=torch.float64, device=d, generator=gq) * 2 - 1 for d, gq in zip(devs, gens)]
    rs = [torch.empty(n, dtype=torch.float64, device=d) for d in devs]
    sync_torch()
    nloc = (C.c_int64 * nd)(*([n] * nd))
    pr, ph, pv = ptrs(rs), ptrs(hs), ptrs(vs)
    sec = timed(lambda: ck(R.mxlo_householder_mul_sharded(sctx, F64, pr, ph, pv, nloc, 1.0, 0.0, 0), "householder"),
                max(10, args.steps), 0.4)
    gbs = 40.0 * n * nd / sec / 1e9
    out["opHouseholder_mul"] = {"ms_per_step": round(sec * 1e3, 4), "GB/s": round(gbs, 1), "n_per_gpu": n,
                                "frac_hbm_peak_per_gpu": round(gbs / nd / HBM_PEAK_GBS, 4)}
    if nd > 1:
        # the same shards under the SECOND transport: the peer-mapped one-shot exchange (no RCCL call; csrc/peer.hip)
        try:
            pctx = C.c_void_p()
            ck(R.mxlo_shard_ctx_create_ex(nd, (C.c_int32 * nd)(*ids), _lib.SHARD_PEER, C.byref(pctx)), "create (peer transport)")
            keep = sctx
            sctx = pctx
            try:
                pf = describe(pctx)
                secp = timed(lambda: ck(R.mxlo_householder_mul_sharded(pctx, F64, pr, ph, pv, nloc, 1.0, 0.0, 0), "householder (peer)"),
                             max(10, args.steps), 0.2)
                out["peer_transport"] = {"preflight": pf, "opHouseholder_ms_per_step": round(secp * 1e3, 4),
                                         "opHouseholder_GB/s": round(40.0 * n * nd / secp / 1e9, 1)}
            finally:
                sctx = keep
                R.mxlo_shard_ctx_destroy(pctx)
        except SystemExit:
            out["peer_transport"] = {"error": (R.mxlo_shard_last_error() or b"").decode()[:300]}
    del hs, vs, rs
    for d in set(devs):
        with torch.cuda.device(d):
            torch.cuda.empty_cache()
    if not args.no_extras:
        nl, m = 50_000_000, 20
        nloc = (C.c_int64 * nd)(*([nl] * nd))
        q = C.c_void_p()
        ck(R.mxlo_qn_create_sharded(sctx, _lib.QN_LBFGS_FWD, F64, nloc, m, 1, 0, 0.99, 10.0, C.byref(q)), "qn_create")
        ss = [torch.empty(nl, dtype=torch.float64, device=d) for d in devs]
        ys = [torch.empty(nl, dtype=torch.float64, device=d) for d in devs]
        acc, kept, push_s = C.c_int32(0), 0, []
        for _ in range(m + 3):
            for s_, y_, gq in zip(ss, ys, gens):
                s_.uniform_(-1, 1, generator=gq)
                y_.uniform_(0.5, 2.0, generator=gq)
                y_.mul_(s_)
            sync_torch()
            tp = time.perf_counter()
            ck(R.mxlo_qn_push_sharded(q, ptrs(ss), ptrs(ys), C.byref(acc)), "push")      # returns after the accept / reject read
            ck(R.mxlo_shard_ctx_sync(sctx), "sync")                                        # ... the Gram-row dots are still in flight
            push_s.append(time.perf_counter() - tp)
            kept += acc.value
        xs, rs = ss, ys
        for x_, gq in zip(xs, gens):
            x_.uniform_(-1, 1, generator=gq)
        sync_torch()
        px, pr = ptrs(xs), ptrs(rs)
        sec = timed(lambda: ck(R.mxlo_qn_mul_sharded(q, pr, px, -1.0, 0.0, 0), "qn_mul"), 10, 0.1)
        push_ms = sum(push_s[-3:]) / 3 * 1e3             # full memory: the steady-state push!
        bytes_ = (4 * m + 3) * 8.0 * nl
        out["LBFGS_fwd_m20_nlocal5e7"] = {"apply_per_s": round(1.0 / sec, 2), "ms": round(sec * 1e3, 3), "pairs_kept": kept,
                                          "n_global": nl * nd, "frac_hbm_peak_per_gpu": round(bytes_ / sec / 1e9 / HBM_PEAK_GBS, 4),
                                          "push_ms": round(push_ms, 3)}
        ck(R.mxlo_qn_destroy_sharded(q), "qn_destroy")
    ck(R.mxlo_shard_ctx_destroy(sctx), "destroy")
    return out


def bench_lbfgs(lo, torch, dev, ctx, rank, world, distributed, dist, barrier, n=50_000_000):
    """InverseLBFGSOperator m=10, n=5e7 (configs[2]) and LBFGSOperator m=20 (configs[4], row-sharded:
    n_local = 5e7 per GPU) applies per second. (`n` other than 5e7: --qn-nelem, a debugging size; the keys keep their names
    and `qn_nelem_override` says so.)"""
    out = {}
    if n != 50_000_000:
        out["qn_nelem_override"] = n
    gen = torch.Generator(device=dev).manual_seed(99 + rank)

    def fill(op, npairs):
        for _ in range(npairs):
            s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
            dvec = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5
            y = dvec * s + 1e-2 * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5)
            del dvec
            lo.push(op, s, y)
            del s, y

    def time_apply(op, reps):
        x = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        res = torch.empty_like(x)
        for _ in range(3):
            lo.mul(res, op, x, -1.0, 0.0)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            lo.mul(res, op, x, -1.0, 0.0)
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / reps

    def time_push(op, reps=4):
        """push!(op, s, y) with a full memory: wall time per push incl. the accept / reject read-back (an optimiser loop
        pays exactly this); every rank issues the same pushes (replicated decision)."""
        s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        y = s * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5)
        lo.push(op, s, y)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            lo.push(op, s, y)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / reps

    m = 10
    Hinv = lo.InverseLBFGSOperator(torch.float64, n, mem=m, scaling=True, device=dev)
    fill(Hinv, m + 3)
    sec = time_apply(Hinv, 20)
    bytes_ = (4 * m + 3) * 8.0 * n
    out["InverseLBFGS_m10_n5e7"] = {"apply_per_s": round(1.0 / sec, 2), "ms": round(sec * 1e3, 3),
                                    "GB/s_per_gpu(344B/elt)": round(bytes_ / sec / 1e9, 1),
                                    "frac_hbm_peak": round(bytes_ / sec / 1e9 / HBM_PEAK_GBS, 4),
                                    "n_global": n * world}
    psec = time_push(Hinv)
    out["InverseLBFGS_m10_n5e7"]["push_ms"] = round(psec * 1e3, 3)          # one-pass push!: (2m + 4) vector passes
    out["InverseLBFGS_m10_n5e7"]["push_frac_hbm_peak(9.6GB)"] = round((2 * m + 4) * 8.0 * n / psec / 1e9 / HBM_PEAK_GBS, 4)
    del Hinv
    torch.cuda.empty_cache()
    # L-SR1 (src/lsr1.jl): apply over the a_k panel, (2m + 3) vector passes; push! in the streaming schedule, (6m + 6)
    # passes. L-SR1 rejects a pair its memory already reproduces, so the timed pushes use distinct pairs.
    m = 10
    Bs1 = lo.LSR1Operator(torch.float64, n, mem=m, scaling=True, device=dev)
    fill(Bs1, m + 2)
    sec = time_apply(Bs1, 20)
    bytes_ = (2 * m + 3) * 8.0 * n
    out["LSR1_m10_n5e7"] = {"apply_per_s": round(1.0 / sec, 2), "ms": round(sec * 1e3, 3),
                            "frac_hbm_peak": round(bytes_ / sec / 1e9 / HBM_PEAK_GBS, 4), "n_global": n * world}
    prs = []
    for _ in range(4):
        s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        prs.append((s, s * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5)))
    barrier()
    t0 = time.perf_counter()
    nacc = 0
    for s, y in prs:
        lo.push(Bs1, s, y)
        nacc += int(Bs1._last_push_accepted)
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    psec = float(t.item()) / len(prs)
    out["LSR1_m10_n5e7"].update({"push_ms": round(psec * 1e3, 3), "pushes_accepted": f"{nacc}/{len(prs)}",
                                 "push_frac_hbm_peak(26.4GB)": round((6 * m + 6) * 8.0 * n / psec / 1e9 / HBM_PEAK_GBS, 4)})
    del Bs1, prs, s, y
    torch.cuda.empty_cache()
    m = 20
    Bf = lo.LBFGSOperator(torch.float64, n, mem=m, scaling=True, device=dev)
    fill(Bf, m + 3)
    sec = time_apply(Bf, 10)
    bytes_ = (4 * m + 3) * 8.0 * n
    out["LBFGS_fwd_m20_nlocal5e7"] = {"apply_per_s": round(1.0 / sec, 2), "ms": round(sec * 1e3, 3),
                                      "GB/s_per_gpu(664B/elt)": round(bytes_ / sec / 1e9, 1),
                                      "frac_hbm_peak": round(bytes_ / sec / 1e9 / HBM_PEAK_GBS, 4),
                                      "n_global": n * world}
    psec = time_push(Bf)
    out["LBFGS_fwd_m20_nlocal5e7"]["push_ms"] = round(psec * 1e3, 3)        # one-pass push!: (2m + 5) vector passes
    out["LBFGS_fwd_m20_nlocal5e7"]["push_frac_hbm_peak(18GB)"] = round((2 * m + 5) * 8.0 * n / psec / 1e9 / HBM_PEAK_GBS, 4)
    # the optimiser-iteration pattern: one push! (device-side Gram update, a_k left implicit), one apply and one
    # solve_shifted_system! (G rebuilt from the Gram matrices) per iteration
    s = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    y = s * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.5 + 0.5)
    x = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    res = torch.empty_like(x)

    def iteration():
        lo.push(Bf, s, y)
        lo.mul(res, Bf, x, 1.0, 0.0)
        lo.solve_shifted_system(res, Bf, x, 0.1)
    iteration()
    barrier()
    t0 = time.perf_counter()
    for _ in range(4):
        iteration()
    barrier()
    out["LBFGS_fwd_m20_iteration(push+mul+solve_shifted)"] = {"ms": round((time.perf_counter() - t0) / 4 * 1e3, 3)}
    del Bf, s, y, x, res
    torch.cuda.empty_cache()

    # ---- cfg5 read directly: the sharded apply rate next to the SAME global problem on ONE GPU, measured in this run.
    # N ranks hold n_global = N * 5e7 rows; rank 0 then runs the unsharded operator at n = n_global alone (no hook)
    # while the others wait. speedup = sharded apply/s / single-GPU apply/s at identical n_global (target >= 6 at N = 8).
    sharded_aps = out["LBFGS_fwd_m20_nlocal5e7"]["apply_per_s"]
    cfg5 = {"operator": "LBFGSOperator m=20 fp64, compact form, 664 B/elt", "n_global": n * world, "n_gpus": world,
            "apply_per_s": sharded_aps}
    if world == 1:
        cfg5.update({"single_gpu_apply_per_s_same_n": sharded_aps, "speedup_vs_single_gpu_same_n": 1.0})
    else:
        single = None
        if rank == 0:
            try:
                ctx.set_allreduce(None)
                single = single_gpu_leg(lo, torch, dev, n * world, m)
            except Exception as e:                       # e.g. the 192 GB of panels at N = 8 do not fit next to a neighbour
                cfg5["single_gpu_leg_error"] = repr(e)[:200]
        barrier()
        if single is not None:
            cfg5.update({"single_gpu_apply_per_s_same_n": round(1.0 / single, 3), "single_gpu_ms": round(single * 1e3, 3),
                         "speedup_vs_single_gpu_same_n": round(sharded_aps * single, 3)})
    out["cfg5_LBFGS_fwd_m20_sharded"] = cfg5
    return out


def single_gpu_leg(lo, torch, dev, n, m):
    """The cfg5 operator at the full global n on ONE GPU (three n x m panels: S, Y, B; a_k never formed)."""
    gen = torch.Generator(device=dev).manual_seed(4242)
    Bf = lo.LBFGSOperator(torch.float64, n, mem=m, scaling=True, device=dev)
    s = torch.empty(n, dtype=torch.float64, device=dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(m + 3):
        s.uniform_(-1, 1, generator=gen)
        y.uniform_(0.5, 2.0, generator=gen)
        y.mul_(s)
        lo.push(Bf, s, y)
    x, res = s, y
    x.uniform_(-1, 1, generator=gen)
    for _ in range(2):
        lo.mul(res, Bf, x, -1.0, 0.0)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        lo.mul(res, Bf, x, -1.0, 0.0)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    del Bf
    torch.cuda.empty_cache()
    return sec


def time_replayed(lo, torch, dev, tm, fn, reps):
    """Device time per call of a microseconds-long apply with the host out of the picture: `reps` calls recorded into
    ONE hipGraph (graph.py: everything an apply enqueues is stream-ordered) and replayed with a single launch; best of 3
    replays after a clock spin-up. The eager loops next to it issue one Python-mirror call per apply, which costs the
    host 5-15 us depending on the box — on a slow host more than these kernels take, and the eager figure is then the
    host's call rate, not the GPU's; on a fast host the eager loop wins (a graph node costs ~1 us more than a direct
    launch). The legs report the faster of the two and both figures. Returns milliseconds per call, or None when the
    sequence cannot be captured."""
    try:
        fn()
        torch.cuda.synchronize()
        g = lo.CapturedSequence(dev)
        with g:
            for _ in range(reps):
                fn()
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.06:
            g.replay()
            torch.cuda.synchronize()
        best = None
        for _ in range(3):
            tm.start()
            g.replay()
            tm.stop()
            ms = tm.elapsed_ms() / reps
            best = ms if best is None or ms < best else best
        del g
        return best
    except Exception:
        return None


def bench_cfg4(lo, torch, dev, ctx):
    """BASELINE configs[3]: BlockDiagonalOperator of 1024 opDiagonal blocks (1024 rows each: launch-latency
    regime; 97,657 rows each: HBM regime) and kron(A,B), A,B 1024x1024 (f64 MFMA GEMMs), replicas per GPU."""
    from linearoperators_jl_amd.device import Timer
    tm = Timer(ctx)
    out = {}
    gen = torch.Generator(device=dev).manual_seed(4)

    def timeit(fn, reps):
        # these legs are microseconds long and follow host-side set-up during which the GPU drops to its idle clocks:
        # bring the clocks back up on the same kernels (untimed, ~60 ms), then time `reps` applies with HIP events
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.06:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        tm.start()
        for _ in range(reps):
            fn()
        tm.stop()
        return tm.elapsed_ms() / reps

    for bs, reps in ((1024, 200), (97_657, 20)):
        nb = 1024
        dall = torch.rand(nb * bs, dtype=torch.float64, device=dev, generator=gen) + 0.5
        BD = lo.BlockDiagonalOperator(*[lo.opDiagonal(dall[k * bs:(k + 1) * bs]) for k in range(nb)])
        x = torch.rand(nb * bs, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        res = torch.empty_like(x)
        ms = timeit(lambda: lo.mul(res, BD, x, 1.0, 0.0), reps)
        eager_us = None
        if bs == 1024:                                   # launch-latency regime: the device figure is the graph replay
            msr = time_replayed(lo, torch, dev, tm, lambda: lo.mul(res, BD, x, 1.0, 0.0), 200)
            if msr is not None:
                eager_us, replay_us, ms = round(ms * 1e3, 2), round(msr * 1e3, 2), min(ms, msr)
        out[f"BlockDiagonal_1024x{bs}"] = {"us_per_apply": round(ms * 1e3, 2), "launches": 1,
                                          "GB/s(24B/elt)": round(24.0 * nb * bs / ms / 1e6, 1),
                                          "frac_hbm_peak": round(24.0 * nb * bs / ms / 1e6 / HBM_PEAK_GBS, 4)}
        if eager_us is not None:
            out[f"BlockDiagonal_1024x{bs}"].update(timing="the faster of an eager loop of Python-mirror calls and 200 applies in one hipGraph replay",
                                                   us_eager_python_mirror=eager_us, us_graph_replay=replay_us)
        if bs == 1024:
            keep = (BD, x, res, dall)
        else:
            del BD, x, res, dall
    n = 1024
    A = ((torch.rand(n, n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
    B = ((torch.rand(n, n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
    K = lo.kron(A, B)
    x = torch.rand(n * n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    res = torch.empty_like(x)
    ms = timeit(lambda: lo.mul(res, K, x, 1.0, 0.0), 200)
    flop = 4.0 * n ** 3
    out["kron_1024x1024"] = {"us_per_apply": round(ms * 1e3, 2), "TFLOP/s_f64": round(flop / ms / 1e9, 2),
                             "frac_f64_mfma_peak(78.6TF)": round(flop / ms / 1e9 / 78.6, 4)}
    out["kron_1024x1024"]["form"] = "both GEMMs in one launch, XCD-local dependency (kron_fuse = 1, the default)"
    # the launch-bound size of the same operator (round 6: both GEMMs in ONE launch, XCD-local dependency — 0.39 -> 0.44)
    n5 = 512
    A5 = ((torch.rand(n5, n5, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
    B5 = ((torch.rand(n5, n5, dtype=torch.float64, device=dev, generator=gen) * 2 - 1) / 32).t()
    K5 = lo.kron(A5, B5)
    x5 = torch.rand(n5 * n5, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    r5 = torch.empty_like(x5)
    ms5 = timeit(lambda: lo.mul(r5, K5, x5, 1.0, 0.0), 200)
    out["kron_512x512"] = {"us_per_apply": round(ms5 * 1e3, 2), "frac_f64_mfma_peak(78.6TF)": round(4.0 * n5 ** 3 / ms5 / 1e9 / 78.6, 4)}
    del K5, A5, B5, x5, r5
    Ssum = keep[0] + K
    ms = timeit(lambda: lo.mul(res, Ssum, x, 1.0, 0.0), 50)
    out["BlockDiagonal_plus_kron_2^20"] = {"us_per_apply": round(ms * 1e3, 2)}
    torch.cuda.empty_cache()
    return out


def bench_misc(lo, torch, dev, ctx):
    """Other §8 rows next to the headline, N = 1 only (HIP events on the launch stream, clocks spun up first): ComplexF64
    opDiagonal at the headline's bytes, sorted index extension, opHermitian, the dense block apply, and the launch-bound
    Householder."""
    import ctypes as C

    import numpy as np
    from linearoperators_jl_amd import _lib
    from linearoperators_jl_amd.device import Timer
    tm = Timer(ctx)
    out = {}
    gen = torch.Generator(device=dev).manual_seed(11)

    def timeit(fn, reps):
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.06:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
        tm.start()
        for _ in range(reps):
            fn()
        tm.stop()
        return tm.elapsed_ms() / reps

    n = 50_000_000
    mk = lambda k: torch.complex(torch.rand(k, dtype=torch.float64, device=dev, generator=gen) - 0.5,
                                 torch.rand(k, dtype=torch.float64, device=dev, generator=gen) - 0.5)
    d, v, res = mk(n), mk(n), mk(n)
    D = lo.opDiagonal(d)
    ms = timeit(lambda: lo.mul(res, D, v, 1.0, 0.0), 20)
    out["opDiagonal_ComplexF64_n5e7"] = {"ms": round(ms, 4), "GB/s(48B/elt)": round(48.0 * n / ms / 1e6, 1),
                                         "frac_hbm_peak": round(48.0 * n / ms / 1e6 / HBM_PEAK_GBS, 4)}
    ms = timeit(lambda: lo.mul(res, D.H, v, 1.0, 0.0), 20)
    out["opDiagonal_ComplexF64_adjoint_n5e7"] = {"ms": round(ms, 4), "frac_hbm_peak": round(48.0 * n / ms / 1e6 / HBM_PEAK_GBS, 4)}
    del d, v, res, D
    torch.cuda.empty_cache()
    nres, nidx = 40_000_000, 20_000_000
    idx = (torch.randperm(nres, device=dev, generator=gen)[:nidx].sort().values + 1).cpu().numpy()
    R = lo.opRestriction(idx, nres, device=dev)
    u = torch.rand(nidx, dtype=torch.float64, device=dev, generator=gen)
    full = torch.empty(nres, dtype=torch.float64, device=dev)
    # Sorted index sets at >= 1/32 density are applied as bit mask + ranks (mxlo_index_plan, round 5): the index list is
    # never read. "GB/s" stays on the ALGORITHMIC bytes of SURVEY §8d (which include the 8 B/index list the kernel never
    # touches), so a fraction of peak is reported on moved_bytes ONLY — what this form has to move (mask + ranks: 1/4 B per
    # element of the long vector instead of the list); a fraction on bytes that are not moved can exceed 1 and means nothing.
    ms = timeit(lambda: lo.mul(full, R.H, u), 10)
    nb = 16.0 * nidx + 8.0 * nres                               # idx + u per entry, res written once
    mv = 8.0 * nidx + 8.0 * nres + nres / 4.0
    out["opExtension_sorted_2e7_of_4e7"] = {"us": round(ms * 1e3, 1), "GB/s": round(nb / ms / 1e6, 1),
                                            "moved_bytes": mv, "frac_hbm_peak_moved": round(mv / ms / 1e6 / HBM_PEAK_GBS, 4),
                                            "form": "bit mask + ranks (no index list read)"}
    ms = timeit(lambda: lo.mul(u, R, full), 10)
    nb = 24.0 * nidx
    mv = 8.0 * nidx + 8.0 * nres + nres / 4.0                   # all of v is streamed at this density (every sector is touched)
    out["opRestriction_sorted_2e7_of_4e7"] = {"us": round(ms * 1e3, 1), "GB/s": round(nb / ms / 1e6, 1),
                                              "moved_bytes": mv, "frac_hbm_peak_moved": round(mv / ms / 1e6 / HBM_PEAK_GBS, 4),
                                              "form": "bit mask + ranks (no index list read)"}
    del R, u, full, idx
    torch.cuda.empty_cache()
    for nn in (4096, 16384):
        M = torch.rand(nn, nn, dtype=torch.float64, device=dev, generator=gen).t()
        Hm = lo.opHermitian(torch.rand(nn, dtype=torch.float64, device=dev, generator=gen), M)
        x, y = (torch.rand(nn, dtype=torch.float64, device=dev, generator=gen) for _ in range(2))
        ms = timeit(lambda: lo.mul(y, Hm, x, 1.0, 0.0), 20)
        eager_us = None
        if nn <= 4096:                                   # 18 us of GPU work per apply: below the eager loop's host cost
            msr = time_replayed(lo, torch, dev, tm, lambda: lo.mul(y, Hm, x, 1.0, 0.0), 100)
            if msr is not None:
                eager_us, replay_us, ms = round(ms * 1e3, 1), round(msr * 1e3, 1), min(ms, msr)
        out[f"opHermitian_n{nn}"] = {"us": round(ms * 1e3, 1), "GB/s(4n^2 B)": round(4.0 * nn * nn / ms / 1e6, 1),
                                     "frac_hbm_peak": round(4.0 * nn * nn / ms / 1e6 / HBM_PEAK_GBS, 4)}
        if eager_us is not None:
            out[f"opHermitian_n{nn}"].update(timing="the faster of an eager loop of Python-mirror calls and 100 applies in one hipGraph replay",
                                             us_eager_python_mirror=eager_us, us_graph_replay=replay_us)
        try:                                 # dense LinearOperator(M) * v (round 5: one launch of row bands) and its transpose
            opD = lo.LinearOperatorFromMatrix(M)
            for name, o in (("dense_mul", opD), ("dense_transpose_mul", opD.T)):
                ms = timeit(lambda: lo.mul(y, o, x, 1.0, 0.0), 20)
                if nn <= 4096:
                    msr = time_replayed(lo, torch, dev, tm, lambda: lo.mul(y, o, x, 1.0, 0.0), 100)
                    ms = min(ms, msr) if msr is not None else ms
                out[f"{name}_n{nn}"] = {"us": round(ms * 1e3, 1), "GB/s(8n^2 B)": round(8.0 * nn * nn / ms / 1e6, 1),
                                        "frac_hbm_peak": round(8.0 * nn * nn / ms / 1e6 / HBM_PEAK_GBS, 4)}
            del opD
        except Exception as e:
            out[f"dense_mul_n{nn}_error"] = repr(e)
        if nn == 16384:                      # block apply of the dense operator: M read once for 8 columns
            try:
                opM = lo.LinearOperatorFromMatrix(M)
                Vb = torch.rand(8, nn, dtype=torch.float64, device=dev, generator=gen).t()
                Rb = torch.empty(8, nn, dtype=torch.float64, device=dev).t()
                ms = timeit(lambda: lo.mul(Rb, opM, Vb), 10)
                out["dense_block_mul_n16384_k8"] = {"us": round(ms * 1e3, 1), "GB/s(8n^2 B, M once)": round(8.0 * nn * nn / ms / 1e6, 1),
                                                    "frac_hbm_peak": round(8.0 * nn * nn / ms / 1e6 / HBM_PEAK_GBS, 4)}
                Ub = torch.rand(8, nn, dtype=torch.float64, device=dev, generator=gen).t()
                Rt = torch.empty(8, nn, dtype=torch.float64, device=dev).t()
                ms = timeit(lambda: lo.mul(Rt, opM.T, Ub), 10)           # round 5: the block staged in LDS per workgroup
                out["dense_block_transpose_mul_n16384_k8"] = {"us": round(ms * 1e3, 1), "GB/s(8n^2 B, M once)": round(8.0 * nn * nn / ms / 1e6, 1),
                                                              "frac_hbm_peak": round(8.0 * nn * nn / ms / 1e6 / HBM_PEAK_GBS, 4)}
                V4 = torch.rand(4, nn, dtype=torch.float64, device=dev, generator=gen).t()
                R4 = torch.empty(4, nn, dtype=torch.float64, device=dev).t()
                ms = timeit(lambda: lo.mul(R4, Hm, V4), 10)              # round 5: the triangle read once for 4 vectors
                out["opHermitian_block_n16384_k4"] = {"us": round(ms * 1e3, 1), "x_vs_4_single_applies": round(4 * out[f"opHermitian_n{nn}"]["us"] / (ms * 1e3), 2),
                                                      "frac_hbm_peak(triangle once)": round(4.0 * nn * nn / ms / 1e6 / HBM_PEAK_GBS, 4)}
                del opM, Vb, Rb, Ub, Rt, V4, R4
            except Exception as e:           # an extra must never cost the line
                out["dense_block_error"] = repr(e)
        del M, Hm
    # sparse LinearOperator(M) (round 4): the 7-point Laplacian of a 160^3 grid, and 1024 sparse blocks in ONE block-diagonal launch
    try:
        gs = 160
        nS = gs ** 3
        ii = torch.arange(nS, device=dev)
        zz, yy, xx = ii // (gs * gs), (ii // gs) % gs, ii % gs
        cc, rr = [], []
        for dz, dy, dx in ((0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
            ok = (zz + dz >= 0) & (zz + dz < gs) & (yy + dy >= 0) & (yy + dy < gs) & (xx + dx >= 0) & (xx + dx < gs)
            cc.append(ii[ok]); rr.append((ii + (dz * gs + dy) * gs + dx)[ok])
        key = torch.unique(torch.cat(cc) * nS + torch.cat(rr))
        cS, rS = key // nS, key % nS
        ccol = torch.zeros(nS + 1, dtype=torch.int64, device=dev)
        ccol[1:] = torch.cumsum(torch.bincount(cS, minlength=nS), 0)
        vS = torch.rand(key.numel(), dtype=torch.float64, device=dev, generator=gen) - 0.5
        opS = lo.LinearOperatorFromMatrix(torch.sparse_csc_tensor(ccol, rS, vS, size=(nS, nS)))
        xs_, ys_ = torch.rand(nS, dtype=torch.float64, device=dev, generator=gen), torch.empty(nS, dtype=torch.float64, device=dev)
        nb = 12.0 * key.numel() + 24.0 * nS
        ms = timeit(lambda: lo.mul(ys_, opS, xs_, 1.0, 0.0), 20)
        mst = timeit(lambda: lo.mul(ys_, opS.T, xs_, 1.0, 0.0), 20)
        out["sparse_laplacian7_160^3"] = {"nnz": int(key.numel()), "us_A*x": round(ms * 1e3, 1), "us_A'*x": round(mst * 1e3, 1),
                                          "GB/s(12B/entry+24B/row)": round(nb / ms / 1e6, 1),
                                          "frac_hbm_peak": round(nb / ms / 1e6 / HBM_PEAK_GBS, 4)}
        del opS, xs_, ys_, vS, key, cS, rS, ccol, cc, rr, ii, zz, yy, xx
        torch.cuda.empty_cache()
    except Exception as e:                   # an extra must never cost the line
        out["sparse_error"] = repr(e)
    # launch-bound quasi-Newton applies (one launch per apply for <= 64 workgroups) and the Gauss-form complex kron
    try:
        for kind, make in (("LBFGS", lo.LBFGSOperator), ("InverseLBFGS", lo.InverseLBFGSOperator), ("LSR1", lo.LSR1Operator)):
            ns = 1 << 12
            op = make(torch.float64, ns, mem=5, device=dev)
            for _ in range(7):
                s_ = torch.rand(ns, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
                lo.push(op, s_, s_ * (torch.rand(ns, dtype=torch.float64, device=dev, generator=gen) * 0.25 + 1.25)
                        + (0.3 * (torch.rand(ns, dtype=torch.float64, device=dev, generator=gen) - 0.5) if kind == "LSR1" else 0))
            xs, rs = torch.rand(ns, dtype=torch.float64, device=dev, generator=gen), torch.empty(ns, dtype=torch.float64, device=dev)
            us = timeit(lambda: lo.mul(rs, op, xs, 1.0, 0.0), 2000) * 1e3
            usr = time_replayed(lo, torch, dev, tm, lambda: lo.mul(rs, op, xs, 1.0, 0.0), 500)
            ctx.tune("qn_fused_small", 0)
            try:
                us4 = timeit(lambda: lo.mul(rs, op, xs, 1.0, 0.0), 2000) * 1e3
                us4r = time_replayed(lo, torch, dev, tm, lambda: lo.mul(rs, op, xs, 1.0, 0.0), 500)
            finally:
                ctx.tune("qn_fused_small", 1)
            r1, r4 = (usr * 1e3 if usr is not None else None), (us4r * 1e3 if us4r is not None else None)
            out[f"{kind}_m5_n2^12_apply_latency"] = {
                "us_single_launch": round(min(us, r1) if r1 is not None else us, 2),
                "us_four_launches": round(min(us4, r4) if r4 is not None else us4, 2),
                "timing": "the faster of an eager loop of Python-mirror calls (host-bound on a slow host) and 500 applies in one hipGraph replay",
                "us_eager_python_mirror": {"single_launch": round(us, 2), "four_launches": round(us4, 2)},
                "us_graph_replay": {"single_launch": round(r1, 2) if r1 is not None else None, "four_launches": round(r4, 2) if r4 is not None else None}}
            del op
        # quasi-Newton applies at cache-resident sizes (round 5: ONE persistent launch; (4m + 3) * 8 B per element against 8 TB/s)
        mid = {}
        for kind, make in (("InverseLBFGS", lo.InverseLBFGSOperator), ("LBFGS", lo.LBFGSOperator)):
            for mm, ee in ((5, 20), (10, 20), (10, 21), (20, 20)):
                nm = 1 << ee
                op = make(torch.float64, nm, mem=mm, device=dev)
                for _ in range(mm + 1):
                    s_ = torch.rand(nm, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
                    lo.push(op, s_, s_ * (torch.rand(nm, dtype=torch.float64, device=dev, generator=gen) * 0.25 + 1.25))
                xs, rs = torch.rand(nm, dtype=torch.float64, device=dev, generator=gen), torch.empty(nm, dtype=torch.float64, device=dev)
                us = timeit(lambda: lo.mul(rs, op, xs, 1.0, 0.0), 200) * 1e3
                mid[f"{kind}_m{mm}_n2^{ee}"] = {"us": round(us, 1), "frac_hbm_peak": round((4 * mm + 3) * 8.0 * nm / us / 1e3 / HBM_PEAK_GBS, 4)}
                del op, xs, rs
        out["quasi_newton_cache_resident_sizes"] = mid
        nk = 1024
        mkc = lambda: (torch.complex(torch.rand(nk, nk, dtype=torch.float64, device=dev, generator=gen) - 0.5,
                                     torch.rand(nk, nk, dtype=torch.float64, device=dev, generator=gen) - 0.5) / 32).t()
        Kc = lo.kron(mkc(), mkc())
        xk = torch.complex(torch.rand(nk * nk, dtype=torch.float64, device=dev, generator=gen), torch.rand(nk * nk, dtype=torch.float64, device=dev, generator=gen))
        yk = torch.empty_like(xk)
        us = timeit(lambda: lo.mul(yk, Kc, xk, 1.0, 0.0), 50) * 1e3
        out["kron_1024x1024_ComplexF64"] = {"us_per_apply": round(us, 1), "form": "Gauss: 6 real f64-MFMA GEMMs",
                                            "TFLOP/s_complex_product_flop": round(16.0 * nk ** 3 / us / 1e6, 1)}
        del Kc, xk, yk
        torch.cuda.empty_cache()
    except Exception as e:               # an extra must never cost the line
        out["latency_extras_error"] = repr(e)[:200]
    n16 = 1 << 16
    h = torch.rand(n16, dtype=torch.float64, device=dev, generator=gen)
    h /= torch.linalg.vector_norm(h)
    vv, rr = torch.rand(n16, dtype=torch.float64, device=dev, generator=gen), torch.empty(n16, dtype=torch.float64, device=dev)
    fh = _lib.lib().mxlo_householder_mul
    a = (ctx.handle, 0, C.c_void_p(rr.data_ptr()), C.c_void_p(h.data_ptr()), C.c_void_p(vv.data_ptr()), C.c_int64(n16),
         C.c_double(1.0), C.c_double(0.0), 0)
    ms = timeit(lambda: fh(*a), 2000)
    Hs = lo.opHouseholder(h)
    ms_py = timeit(lambda: lo.mul(rr, Hs, vv, 1.0, 0.0), 2000)
    out["opHouseholder_n2^16_latency"] = {"us_C_ABI": round(ms * 1e3, 2), "us_python_mirror": round(ms_py * 1e3, 2), "launches": 1}
    torch.cuda.empty_cache()
    return out


PMC_WORKLOAD = """
import os, sys
sys.path.insert(0, %(root)r)
import torch
import __graft_entry__ as g
lo = g.load_package()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
n = %(n)d
h = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5
h /= torch.linalg.vector_norm(h)
v = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
res = torch.empty(n, dtype=torch.float64, device=dev)
H = lo.opHouseholder(h)
for _ in range(3):
    lo.mul(res, H, v, 1.0, 0.0)
torch.cuda.synchronize()
print("pmc workload done")
"""


def measure_traffic(n: int, timeout_s: float = 150.0):
    """roofline.traffic MEASURED IN THIS RUN: HBM bytes per launch of the dominant kernel from two
    separate `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE — they do not fit one pass) over a child process that runs
    three Householder applies at the workload's own n, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes
    (both counters are in KiB; on gfx950 FETCH_SIZE tallies a 128-byte read request as 64 bytes -> x 2; WRITE_SIZE x 1,
    calibrated against TCC_EA0_WRREQ_64B in profiles/). Returns (bytes per launch or None, description)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found on this box"
    tmp = tempfile.mkdtemp(prefix="mxlo_pmc_", dir="/tmp")
    try:
        script = os.path.join(tmp, "workload.py")
        with open(script, "w") as f:
            f.write(PMC_WORKLOAD % {"root": ROOT, "n": n})
        got = {}
        env = dict(os.environ, TMPDIR="/tmp")
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            p = subprocess.run([exe, "--kernel-trace", "--output-format", "csv", "--pmc", ctr, "-d", out, "-o", "pmc", "--",
                                sys.executable, script], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=timeout_s)
            vals = []
            for fcsv in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(fcsv)):
                    if "HouseholderOp" in r.get("Kernel_Name", "") and "map_kernel" in r["Kernel_Name"] and r.get("Counter_Name") == ctr:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, f"the {ctr} pass produced no counter rows for map_kernel<HouseholderOp> (rc {p.returncode}): {p.stdout[-200:]!r}"
            got[ctr] = sum(vals) / len(vals)
        traffic = got["FETCH_SIZE"] * 1024.0 * 2.0 + got["WRITE_SIZE"] * 1024.0
        return traffic, ("measured in THIS run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over a child process "
                         f"running 3 applies at n = {n}; FETCH_SIZE = {got['FETCH_SIZE']:.0f} KiB x 2 (gfx950: a 128-B read request is tallied "
                         f"as 64 B), WRITE_SIZE = {got['WRITE_SIZE']:.0f} KiB x 1 — MI355X_MICROARCH.md §HBM")
    except Exception as e:
        return None, f"PMC passes failed: {e!r}"[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_leg(n_sample: int, gpu=None):
    """Oracle (port of the reference `mulHouseholder!`, src/linalg.jl:77-83) on the host, 1 thread. With `gpu` = (h, v, apply)
    of the timed workload and n_sample equal to its n, the oracle runs on the DEVICE'S operands (copied back) and the result
    of one more device apply is compared with it: `parity` = the relative L2 distance at the benchmark's own size."""
    import numpy as np
    import oracle
    parity = None
    if gpu is not None and gpu[0].numel() == n_sample:
        h, v = gpu[0].cpu().numpy(), gpu[1].cpu().numpy()
    else:
        rng = np.random.default_rng(0)
        h = rng.random(n_sample) - 0.5
        h /= np.linalg.norm(h)
        v = rng.random(n_sample) * 2 - 1
    res = np.empty(n_sample)
    oracle.householder_mul(res, h, v, 1.0, 0.0)          # warm-up / page-in
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.householder_mul(res, h, v, 1.0, 0.0)
        reps += 1
        if time.perf_counter() - t0 > 8.0 or reps >= 20:
            break
    sec = (time.perf_counter() - t0) / reps
    if gpu is not None and gpu[0].numel() == n_sample:
        dev_res = gpu[2]().cpu().numpy()                 # one more mul!(res, H, v, 1, 0) of the timed operator
        diff = dev_res - res
        nr = float(np.linalg.norm(res))
        parity = {"rel_l2": float(np.linalg.norm(diff) / nr), "max_abs": float(np.max(np.abs(diff))), "tolerance": 1e-12,
                  "n": int(n_sample), "what": "device mul!(res, opHouseholder(h), v, 1, 0) vs oracle.householder_mul on the same h, v "
                                              "(the benchmark's operands copied back), every element"}
        parity["ok"] = bool(parity["rel_l2"] <= parity["tolerance"])
        del dev_res, diff
    own = " — the GPU workload's own h, v" if parity else ""
    out = {"value": round(40.0 * n_sample / sec / 1e9, 2), "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": f"oracle.householder_mul (C restatement of mulHouseholder!, gcc -O2, 1 thread) on n={n_sample} "
                     f"fp64{own}, {reps} reps, {sec * 1e3:.1f} ms/apply, counted at the same 40 B/elt; "
                     f"host has {os.cpu_count()} logical CPUs"}
    try:   # generous upper bound: the same arithmetic on every host core (OpenMP, first-touch placement)
        out["all_cores"] = cpu_leg_allcore(n_sample)
    except Exception as e:  # pragma: no cover  (no libgomp: report, do not fail the bench line)
        out["all_cores"] = {"error": repr(e)}
    return out, parity


def cpu_leg_lbfgs(budget_s: float = 12.0, gpu=None):
    """extras.InverseLBFGS_m10_n5e7.cpu — the reference's two-loop recursion (src/lbfgs.jl:117-154) on the host cores, at
    n / 10 = 5e6 and scaled LINEARLY to n = 5e7 (the apply is memory-bound on the host too: 21 vectors streamed per dot /
    axpy statement; the full size needs 8.4 GB of panels and ~1 s per apply): (1) the oracle's statement-by-statement
    restatement, 1 thread (Julia's broadcast and a ddot below OpenBLAS's threading cut-off are single-threaded);
    (2) the same statement sequence with every statement an OpenMP loop over all host cores (upper bound)."""
    import numpy as np
    import oracle
    t_start = time.perf_counter()
    nc, m, scale = 5_000_000, 10, 10
    Oc = oracle.LBFGS(nc, mem=m, inverse=True)
    rc = np.random.default_rng(1)
    S = np.empty((m, nc))            # row k = s_k: a column-major n x m panel seen from C
    Y = np.empty((m, nc))
    for k in range(m):
        sc = rc.uniform(-1, 1, nc)
        yc = sc * rc.uniform(0.5, 2.0, nc)
        Oc.push(sc, yc)
        S[k], Y[k] = sc, yc
    xc, outc = rc.uniform(-1, 1, nc), np.empty(nc)
    Oc.mul(outc, xc)
    reps, t0 = 0, time.perf_counter()
    while True:
        Oc.mul(outc, xc)
        reps += 1
        if time.perf_counter() - t0 > budget_s / 3 or reps >= 10:
            break
    sec1 = (time.perf_counter() - t0) / reps
    parity = None
    if gpu is not None:                                  # the device operator fed the SAME pairs, applied to the same x
        lo_, torch_, dev_ = gpu
        Hd = lo_.InverseLBFGSOperator(nc, mem=m, device=dev_)
        for k in range(m):
            lo_.push(Hd, torch_.from_numpy(S[k]).to(dev_), torch_.from_numpy(Y[k]).to(dev_))
        rd = torch_.empty(nc, dtype=torch_.float64, device=dev_)
        lo_.mul(rd, Hd, torch_.from_numpy(xc).to(dev_), 1.0, 0.0)
        dres = rd.cpu().numpy()
        parity = {"rel_l2": float(np.linalg.norm(dres - outc) / np.linalg.norm(outc)), "max_abs": float(np.max(np.abs(dres - outc))),
                  "tolerance": 1e-9, "n": nc, "mem": m,
                  "what": "device mul!(res, InverseLBFGSOperator, x) after the same 10 push! vs oracle.LBFGS(inverse).mul "
                          "(reference statement order), every element"}
        parity["ok"] = bool(parity["rel_l2"] <= parity["tolerance"])
        del Hd, rd, dres
        torch_.cuda.empty_cache()
    out = {"apply_per_s_at_n5e7": round(1.0 / (sec1 * scale), 3), "cores": 1, "kind": "port",
           "sample": f"oracle.LBFGS(inverse).mul — C restatement of lbfgs_multiply in reference statement order, 1 thread, "
                     f"n = {nc} ({reps} reps, {sec1 * 1e3:.1f} ms/apply), scaled x{scale} to n = 5e7 (memory-bound: linear in n)"}
    try:
        M = oracle.mt_lib()
        threads = int(M.orc_mt_max_threads())
        ys = np.array([float(np.dot(S[k], Y[k])) for k in range(m)])
        gamma = float(ys[m - 1] / np.dot(Y[m - 1], Y[m - 1]))
        q, al, res = np.empty(nc), np.zeros(m), np.empty(nc)
        for a in (q, res):
            M.orc_mt_fill_f64(a.ctypes.data, nc, 7, -1.0, 1.0, threads)
        call = lambda: M.orc_mt_lbfgs_inv_mul_f64(res.ctypes.data, S.ctypes.data, Y.ctypes.data, nc, ys.ctypes.data, al.ctypes.data, m,
                                                  m + 1, 1, gamma, xc.ctypes.data, q.ctypes.data, nc, 1.0, 0.0, threads)
        call()
        ref = Oc.mul(np.empty(nc), xc)
        err = float(np.linalg.norm(res - ref) / np.linalg.norm(ref))      # the all-core variant computes the same thing
        best, reps2, t_all = float("inf"), 0, time.perf_counter()
        while time.perf_counter() - t_all < budget_s / 3 and reps2 < 30:
            t0 = time.perf_counter()
            call()
            best = min(best, time.perf_counter() - t0)
            reps2 += 1
        out["all_cores"] = {"apply_per_s_at_n5e7": round(1.0 / (best * scale), 2), "cores": threads, "rel_diff_vs_1_thread": err,
                            "sample": f"OpenMP restatement (every statement one parallel loop), best of {reps2} at n = {nc}: "
                                      f"{best * 1e3:.2f} ms/apply, scaled x{scale}"}
    except Exception as e:  # pragma: no cover
        out["all_cores"] = {"error": repr(e)[:200]}
    out["host_logical_cpus"] = os.cpu_count()
    out["wall_s"] = round(time.perf_counter() - t_start, 1)
    return out, parity


def cpu_leg_kron(budget_s: float = 8.0):
    """extras.kron_1024x1024.cpu — the reference's kron prod! as written (src/kron.jl:17-18: Matrix(B * X * transpose(A)),
    i.e. 1024 pairs of GEMVs through src/abstract.jl:282-292) on the host: the oracle's restatement with 1 thread, and
    the same column loop spread over all cores."""
    import numpy as np
    import oracle
    t_start = time.perf_counter()
    nk = 1024
    rngk = np.random.default_rng(0)
    Ak, Bk = (rngk.random((nk, nk)) - 0.5) / 32, (rngk.random((nk, nk)) - 0.5) / 32
    xk = rngk.random(nk * nk)
    flop = 4.0 * nk ** 3
    resk = np.empty(nk * nk)
    oracle.kron_mul(resk, Ak, Bk, xk, 1.0, 0.0)
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.kron_mul(resk, Ak, Bk, xk, 1.0, 0.0)
        reps += 1
        if time.perf_counter() - t0 > budget_s / 2 or reps >= 5:
            break
    sec1 = (time.perf_counter() - t0) / reps
    out = {"ms_per_apply": round(sec1 * 1e3, 1), "GFLOP/s": round(flop / sec1 / 1e9, 2), "cores": 1, "kind": "port",
           "sample": f"oracle.kron_mul — reference-literal kron prod! (1024 x (X*w, B*u) GEMV pairs), gcc -O2, 1 thread, {reps} reps"}
    try:
        M = oracle.mt_lib()
        threads = int(M.orc_mt_max_threads())
        Af, Bf = np.asfortranarray(Ak), np.asfortranarray(Bk)
        work, res2 = np.empty(threads * nk), np.empty(nk * nk)
        call = lambda: M.orc_mt_kron_mul_f64(res2.ctypes.data, Af.ctypes.data, nk, nk, Bf.ctypes.data, nk, nk, xk.ctypes.data, 1.0, 0.0,
                                             work.ctypes.data, threads)
        call()
        err = float(np.linalg.norm(res2 - resk) / np.linalg.norm(resk))
        best, reps2, t_all = float("inf"), 0, time.perf_counter()
        while time.perf_counter() - t_all < budget_s / 2 and reps2 < 50:
            t0 = time.perf_counter()
            call()
            best = min(best, time.perf_counter() - t0)
            reps2 += 1
        out["all_cores"] = {"ms_per_apply": round(best * 1e3, 2), "GFLOP/s": round(flop / best / 1e9, 1), "cores": threads,
                            "rel_diff_vs_1_thread": err, "sample": f"OpenMP over the 1024 result columns, best of {reps2}"}
    except Exception as e:  # pragma: no cover
        out["all_cores"] = {"error": repr(e)[:200]}
    out["host_logical_cpus"] = os.cpu_count()
    out["wall_s"] = round(time.perf_counter() - t_start, 1)
    return out


def cpu_leg_allcore(n_sample: int):
    """oracle/lo_oracle_mt.c: mulHouseholder! arithmetic with an OpenMP dot + update over all host cores."""
    import numpy as np
    import oracle
    M = oracle.mt_lib()
    threads = int(M.orc_mt_max_threads())
    h, v, res = (np.empty(n_sample) for _ in range(3))
    for k, a in enumerate((h, v, res)):
        M.orc_mt_fill_f64(a.ctypes.data, n_sample, 0x5EED0001 + k, -1.0, 1.0, threads)
    best, t_all, reps = float("inf"), time.perf_counter(), 0
    while time.perf_counter() - t_all < 4.0 and reps < 40:
        t0 = time.perf_counter()
        M.orc_mt_householder_mul_f64(res.ctypes.data, h.ctypes.data, v.ctypes.data, n_sample, 1.0, 0.0, threads)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    return {"value": round(40.0 * n_sample / best / 1e9, 1), "unit": "GB/s", "cores": threads,
            "sample": f"OpenMP restatement, best of {reps} reps on n={n_sample}, {best * 1e3:.2f} ms/apply"}


if __name__ == "__main__":
    main()
