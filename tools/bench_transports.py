#!/usr/bin/env python
"""Latency of the scalar all-reduce transports that can run on ONE device (mxlo_shard_ctx_preflight: HIP events around 50
back-to-back in-place all-reduces per payload, slowest shard): RCCL at world 1 (a communicator of one rank: no fabric hop),
the loopback transport (stream events + host barrier), the peer-mapped exchange with device / host mailboxes on 1, 2 and 8
same-device shards (one kernel per collective at 1 shard; post + gather with a host barrier between them otherwise), and
the peer exchange over a POSIX shm segment (one process per GPU shape, world 1). Multi-device numbers need a multi-GPU node."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

lo = g.load_package()
from linearoperators_jl_amd.device import get_ctx

R = lo._lib.rccl_lib()
dev = torch.device("cuda", 0)
ctx = get_ctx(dev)
print("# tools/bench_transports.py: us per all-reduce of 8 B / 320 B / 6912 B (preflight latency loop), one MI355X")


def shard(name, ids, transport, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        s = C.c_void_p()
        nd = len(ids)
        if R.mxlo_shard_ctx_create_ex(nd, (C.c_int32 * nd)(*ids), transport, C.byref(s)) != 0:
            print(f"{name:58s} create failed: {R.mxlo_shard_last_error().decode()}")
            return
        lat = (C.c_double * 3)()
        rc = R.mxlo_shard_ctx_preflight(s, 50, 30000, lat)
        print(f"{name:58s} " + ("%8.2f %8.2f %8.2f" % tuple(lat) if rc == 0 else "preflight failed: " + R.mxlo_shard_last_error().decode()), flush=True)
        R.mxlo_shard_ctx_destroy(s)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


shard("RCCL, 1 shard (ncclCommInitAll over one device)", [0], lo._lib.SHARD_RCCL)
shard("loopback, 2 shards on one device", [0, 0], lo._lib.SHARD_LOOPBACK)
shard("loopback, 8 shards on one device", [0] * 8, lo._lib.SHARD_LOOPBACK)
shard("peer, device mailboxes, 1 shard (one kernel)", [0], lo._lib.SHARD_PEER)
shard("peer, device mailboxes, 2 shards on one device", [0, 0], lo._lib.SHARD_PEER)
shard("peer, device mailboxes, 8 shards on one device", [0] * 8, lo._lib.SHARD_PEER)
shard("peer, pinned host mailboxes, 1 shard (one kernel)", [0], lo._lib.SHARD_PEER, {"MXLO_PEER_MEM": "host"})
shard("peer, pinned host mailboxes, 8 shards on one device", [0] * 8, lo._lib.SHARD_PEER, {"MXLO_PEER_MEM": "host"})
ph = lo.sharded.PeerShmHook(0, 1)
lat = ph.preflight(ctx.stream, 50, 30000)
print(f"{'peer over a POSIX shm segment, world 1 (process per GPU)':58s} %8.2f %8.2f %8.2f" % (lat["8B"], lat["320B"], lat["6912B"]))
ph.close()
hk = lo.sharded.NativeRcclHook(0, 1)
lat = hk.preflight(ctx.stream, 50, 30000)
print(f"{'native RCCL hook, world 1 (process per GPU)':58s} %8.2f %8.2f %8.2f" % (lat["8B"], lat["320B"], lat["6912B"]))
hk.close()
