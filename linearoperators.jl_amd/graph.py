"""hipGraph replay of launch-bound `mul!` sequences (include/mxlo.h: mxlo_graph_*).

At small n an apply is 2-4 dependent launches plus the host's per-call work; a Krylov or quasi-Newton inner
loop repeats exactly the same launches on the same buffers. `CapturedSequence` records any sequence of
host-mirror calls (everything the apply paths enqueue is stream-ordered, no host sync) into ONE hipGraph on
a side stream and replays it with a single launch. Buffers, sizes, α and β are baked in; the data they hold
is read at replay time. `push!` (host control flow) cannot be captured.

Staleness: a quasi-Newton apply bakes the handle's slot order / insert position / update count into the recorded
launches, and opHermitian the ctx scratch pointer. The library remembers the generation of every captured handle
(bumped by push!, reset!, mode changes) and of the scratch; `replay()` after any of them changed raises
``MxloError`` (MXLO_ESTATE) instead of replaying stale metadata — re-capture after `push!`/`reset!`.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .device import get_ctx, indexed_device


class CapturedSequence:
    """with CapturedSequence(device) as g:  lo.mul(res, op, v, a, b); ...   then  g.replay()"""

    def __init__(self, device=None, stream: torch.cuda.Stream | None = None):
        self.device = indexed_device(device)
        self.stream = stream if stream is not None else torch.cuda.Stream(self.device)
        self._g = C.c_void_p()
        self._ctx = None
        self._scope = None

    def __enter__(self):
        torch.cuda.current_stream(self.device).synchronize()
        self._scope = torch.cuda.stream(self.stream)
        self._scope.__enter__()
        self._ctx = get_ctx(self.device)                 # binds the ctx to the side stream
        _lib.call("mxlo_graph_begin", self._ctx.handle)
        return self

    def __exit__(self, et, ev, tb):
        try:
            if et is None:
                _lib.call("mxlo_graph_end", self._ctx.handle, C.byref(self._g))
            else:                                        # abandon the capture, keep the original exception
                g = C.c_void_p()
                _lib.lib().mxlo_graph_end(self._ctx.handle, C.byref(g))
                if g:
                    _lib.lib().mxlo_graph_destroy(g)
        finally:
            self._scope.__exit__(et, ev, tb)
            get_ctx(self.device)                         # rebind the ctx to the caller's stream
        return False

    def replay(self, sync_streams: bool = True):
        """One replay (mxlo_graph_launch: the recorded launches re-issued directly for short chains, hipGraphLaunch
        otherwise). sync_streams: order the replay after the caller's stream and the caller's
        stream after the replay (two event waits, no host sync); pass False inside a loop that lives
        entirely on `self.stream`."""
        if not self._g:
            raise RuntimeError("nothing was captured")
        if sync_streams:
            cur = torch.cuda.current_stream(self.device)
            if cur != self.stream:
                self.stream.wait_stream(cur)
        _lib.call("mxlo_graph_launch", self._g)
        if sync_streams:
            cur = torch.cuda.current_stream(self.device)
            if cur != self.stream:
                cur.wait_stream(self.stream)

    def info(self):
        """{"nodes": recorded launches, "direct": replay re-issues them one by one instead of hipGraphLaunch}"""
        a = (C.c_int64 * 2)()
        _lib.call("mxlo_graph_info", self._g, a)
        return {"nodes": int(a[0]), "direct": bool(a[1])}

    def __del__(self):
        try:
            if self._g:
                _lib.lib().mxlo_graph_destroy(self._g)
        except Exception:
            pass


def capture_mul(res, op, v, alpha=None, beta=None) -> CapturedSequence:
    """Record `mul!(res, op, v, α, β)` (after one eager warm-up apply that sizes lazy temporaries and
    workspaces) and return the replayable graph."""
    from .operators import mul
    saved = res.clone() if (beta is not None and beta != 0) else None
    mul(res, op, v, alpha, beta)                         # warm-up
    if saved is not None:
        res.copy_(saved)
    g = CapturedSequence(res.device)
    with g:
        mul(res, op, v, alpha, beta)
    return g
