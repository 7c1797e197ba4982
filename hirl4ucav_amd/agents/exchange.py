"""The transports of the sharded update's exchange step (ONE sum all-reduce per phase; the reference is a single process,
hirl/agents/HIRL.py:52, and has no counterpart): RCCL direct (hx_rccl_*), the peer-read kernels over hipIpc mappings (hx_allreduce_oneshot /
hx_allreduce_twostage), and the all-rank negotiation that decides whether RCCL direct is used.  HirlEngine (engine.py) drives them."""
import ctypes
import os

import torch

from .. import _lib

_vp, _i32, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
_P = ctypes.POINTER
_lib.register("hx_ipc_alloc", [ctypes.c_int64, _i32, _P(_vp)])
_lib.register("hx_ipc_free", [_vp])
_lib.register("hx_ipc_export", [_vp, _vp])
_lib.register("hx_ipc_import", [_vp, _P(_vp)])
_lib.register("hx_ipc_close", [_vp])
_lib.register("hx_allreduce_oneshot", [_vp, _P(_vp), _P(_vp), _vp, _i32, _i32, ctypes.c_int64, ctypes.c_uint32, _i32, _vp])
_lib.register("hx_allreduce_twostage", [_vp, _P(_vp), _P(_vp), _P(_vp), _P(_vp), _vp, _i32, _i32, ctypes.c_int64, ctypes.c_uint32, _i32, _i32, _vp])
_lib.register("hx_rccl_unique_id", [_vp])
_lib.register("hx_rccl_init", [_vp, _i32, _i32, _P(_vp)])
_lib.register("hx_rccl_allreduce", [_vp, _vp, ctypes.c_int64, _i32, _vp])
_lib.register("hx_rccl_allreduce_bf16", [_vp, _vp, _vp, ctypes.c_int64, _vp])
_lib.register("hx_rccl_destroy", [_vp])


_lib.register("hx_rccl_available", [])


class RcclDirect:
    """RCCL without torch.distributed in the loop (include/hirl4ucav.h hx_rccl_*): ncclAllReduce enqueued on the engine's stream by the
    library; one GPU per rank.  The communicator's 128-byte id is made on ONE rank (make_id) and reaches the others by any byte channel —
    negotiate_rccl_direct sends it once through the existing process group; construction (ncclCommInitRank) is a collective."""

    @staticmethod
    def available():
        """librccl.so and the four entry points can be bound in THIS process (a local check, no collective); raises HxError otherwise"""
        _lib.call("hx_rccl_available")

    @staticmethod
    def make_id():
        raw = ctypes.create_string_buffer(128)
        _lib.call("hx_rccl_unique_id", raw)
        return raw.raw

    def __init__(self, uid, world, rank, bf16=False):
        """bf16: the messages travel (and are summed) as bf16 — hx_rccl_allreduce_bf16: half the wire bytes; every rank still receives the same
        bits, so the replicas stay identical (opt-in: bench.py --exchange rccl-bf16)"""
        self.world, self.rank, self.bf16 = int(world), int(rank), bool(bf16)
        self._scratch = None
        comm = _vp()
        _lib.call("hx_rccl_init", ctypes.create_string_buffer(uid, 128), self.world, self.rank, ctypes.byref(comm))
        self.comm = comm

    def allreduce(self, t):
        """t <- sum over the ranks of t (fp32), in place, on the current stream"""
        n = t.numel()
        if self.bf16 and n % 4 == 0:
            if self._scratch is None or self._scratch.numel() < n:
                self._scratch = torch.empty(n, dtype=torch.bfloat16, device=t.device)
            _lib.call("hx_rccl_allreduce_bf16", self.comm, t.data_ptr(), self._scratch.data_ptr(), n, _lib.stream_ptr())
            return t
        _lib.call("hx_rccl_allreduce", self.comm, t.data_ptr(), n, 0, _lib.stream_ptr())
        return t

    def probe(self, device):
        """one all-reduce of ones through the new communicator; raises when the sum is not the world size (synchronises)"""
        ones = torch.ones(64, dtype=torch.float32, device=device)
        self.allreduce(ones)  # (world sizes are exact in bf16 too)
        torch.cuda.synchronize()
        if not bool((ones == float(self.world)).all()):
            raise _lib.HxError(f"the probe all-reduce returned {float(ones[0])} instead of {self.world}")

    def close(self):
        if self.comm is not None:
            torch.cuda.synchronize()
            c, self.comm = self.comm, None
            _lib.call("hx_rccl_destroy", c)


def negotiate_rccl_direct(group=None, flag_device="cpu", available=None, make_id=None, connect=None, log=None):
    """Every rank of `group` takes the SAME decision about RCCL direct, and no rank is ever left alone in a collective (rank 0 used to
    skip the id broadcast when hx_rccl_unique_id raised, and the other ranks hung in it).  The collectives below run on every rank, in this
    order, whatever failed locally:
      1. available()               local: the library and its symbols bind here
      2. broadcast of the id       rank 0 ALWAYS broadcasts — the 128 bytes, or None when step 1 or make_id() failed there
      3. MIN all-reduce of `ok`    before ncclCommInitRank (itself a collective: a rank that could not enter it would hang the others)
      4. connect(id, world, rank)  ncclCommInitRank + one probe all-reduce — only when step 3 agreed
      5. MIN all-reduce of `ok`    after the probe
    -> (the connected object, "") on every rank, or (None, why) on every rank (a communicator built by some ranks is closed again).
    LIMIT of the promise [ADVICE r5]: it covers steps 1-3 and the agreement around step 4, not step 4's inside.  connect() is itself collective and
    unbounded — ncclCommInitRank, then a probe all-reduce + torch.cuda.synchronize(): if ONE rank's init raises locally while the others are already inside
    theirs, those block in RCCL until the launcher's own timeout ends the job (close() synchronises too).  Step 3 exists to make that rare (a rank that
    cannot even bind the library never lets the others enter), it cannot exclude it; a deadline would need RCCL's non-blocking init
    (ncclCommInitRankConfig + ncclCommGetAsyncError), which this header-free binding does not restate.  The CPU tests inject a NON-collective connect().
    `available` / `make_id` default to RcclDirect's, `connect` to RcclDirect(id, world, rank) + probe on the current GPU; tests inject failing
    ones over gloo on the CPU."""
    dist = torch.distributed
    available = RcclDirect.available if available is None else available
    make_id = RcclDirect.make_id if make_id is None else make_id
    if connect is None:
        def connect(uid, world, rank):
            r = RcclDirect(uid, world, rank)
            r.probe(torch.device("cuda", torch.cuda.current_device()))
            return r
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ok, why, conn = 1, "", None

    def agreed(ok_here):
        flag = torch.tensor([ok_here], dtype=torch.int32, device=flag_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1

    try:
        available()
    except Exception as e:  # noqa: BLE001 — any failure means "use the other transport", on every rank
        ok, why = 0, repr(e)
    box = [None]
    if rank == 0 and ok:
        try:
            box[0] = make_id()
        except Exception as e:  # noqa: BLE001
            ok, why = 0, repr(e)
    dist.broadcast_object_list(box, src=0, group=group)
    if box[0] is None and ok:
        ok, why = 0, "rank 0 could not make a communicator id"
    if not agreed(ok):
        why = why or "another rank cannot use RCCL direct"
    else:
        try:
            conn = connect(box[0], world, rank)
        except Exception as e:  # noqa: BLE001
            ok, why = 0, repr(e)
        if not agreed(ok):
            why = why or "another rank failed to connect"
            if conn is not None:
                try:
                    conn.close()
                except Exception:  # noqa: BLE001
                    pass
                conn = None
    if conn is None and log is not None:
        log(f"hirl4ucav_amd: RCCL direct not available on rank {rank} ({why}): the gradient exchange stays on torch.distributed.all_reduce")
    return conn, ("" if conn is not None else why)


class _DeviceWords:
    """raw device memory as a torch tensor (no copy): the __cuda_array_interface__ protocol"""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class OneShotExchange:
    """The exchange step over hipIpc peer mappings (include/hirl4ucav.h hx_allreduce_oneshot): every rank owns, per message kind, two
    message buffers (epoch parity) that the peers map, and one fine-grained flag word.  Handles travel once, at construction, through
    torch.distributed.all_gather_object (any backend)."""

    def __init__(self, sizes, device, group=None, timeout_ms=5000, two_stage=False, bf16=False):
        """two_stage: reduce-scatter + all-gather (hx_allreduce_twostage: 2 (world - 1) / world x n floats per rank over xGMI instead of
        world x n); bf16 (two_stage only): the reduced slices travel as bf16"""
        dist = torch.distributed
        self.two_stage, self.bf16 = bool(two_stage), bool(bf16) and bool(two_stage)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device, self.timeout_ms = device, int(timeout_ms)
        self.kinds = list(sizes)
        self.n = {k: (int(v) + 3) & ~3 for k, v in sizes.items()}
        total = sum((3 if self.two_stage else 2) * self.n[k] for k in self.kinds)  # two message buffers (epoch parity) [+ the reduced slices]
        msg, flag = _vp(), _vp()
        _lib.call("hx_ipc_alloc", total * 4, 0, ctypes.byref(msg))
        _lib.call("hx_ipc_alloc", 256, 1, ctypes.byref(flag))  # word k: flag of kind k; word 32: status
        self._own = (msg.value, flag.value)
        hm, hf = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        _lib.call("hx_ipc_export", msg, hm)
        _lib.call("hx_ipc_export", flag, hf)
        every = [None] * self.world
        uuid = str(getattr(torch.cuda.get_device_properties(device), "uuid", "")) or f"{os.uname().nodename}:{torch.cuda.current_device()}"
        dist.all_gather_object(every, (hm.raw, hf.raw, uuid), group=group)
        uuids = [e[2] for e in every]
        every = [e[:2] for e in every]
        # Ranks that SHARE a GPU (functional tests on a one-GPU box): a rank's wait kernel spinning on every CU keeps the peer's 1024-thread
        # workgroups from being placed for seconds at a time.  A few workgroups leave the chip to the peer (3 s instead of minutes for a short
        # run); ranks with a GPU each keep the 256 workgroups whose loads cover the xGMI round trip.  (Read once, at the first exchange.)
        self.shared_device = len(set(uuids)) < len(uuids)
        if self.shared_device:
            os.environ.setdefault("HX_ONESHOT_BLOCKS", "16")
        self._peers = []
        bases, flags = [], []
        for r, (m_h, f_h) in enumerate(every):
            if r == self.rank:
                bases.append(msg.value)
                flags.append(flag.value)
                continue
            pm, pf = _vp(), _vp()
            _lib.call("hx_ipc_import", ctypes.create_string_buffer(m_h, 64), ctypes.byref(pm))
            _lib.call("hx_ipc_import", ctypes.create_string_buffer(f_h, 64), ctypes.byref(pf))
            self._peers += [pm.value, pf.value]
            bases.append(pm.value)
            flags.append(pf.value)
        self.status_ptr = flag.value + 32 * 4
        self.epoch = {k: 0 for k in self.kinds}
        self.own, self.bufs, self.flags, self.reduced, self.reds, self.flags2 = {}, {}, {}, {}, {}, {}
        off = 0
        arr = _vp * self.world
        for ki, k in enumerate(self.kinds):
            for par in (0, 1):
                self.own[k, par] = torch.as_tensor(_DeviceWords(msg.value + off * 4, self.n[k]), device=device)
                self.bufs[k, par] = arr(*[b + off * 4 for b in bases])
                off += self.n[k]
            if self.two_stage:
                self.reds[k] = arr(*[b + off * 4 for b in bases])
                off += self.n[k]
            self.flags[k] = arr(*[f + ki * 4 for f in flags])
            self.flags2[k] = arr(*[f + (8 + ki) * 4 for f in flags])  # words 8..: the second stage's flags
            self.reduced[k] = torch.zeros(self.n[k], dtype=torch.float32, device=device)
        dist.barrier(group=group)  # every mapping exists before the first exchange

    def write_buffer(self, kind):
        """where this rank's NEXT message of `kind` must be written (the parity of the epoch its exchange will carry)"""
        return self.own[kind, (self.epoch[kind] + 1) & 1]

    def allreduce(self, kind):
        """sum of every rank's message written into write_buffer(kind) -> a local tensor (the same bits on every rank)"""
        self.epoch[kind] += 1
        e = self.epoch[kind]
        if self.two_stage:
            _lib.call("hx_allreduce_twostage", self.reduced[kind].data_ptr(), self.bufs[kind, e & 1], self.reds[kind], self.flags[kind], self.flags2[kind],
                      self.status_ptr, self.world, self.rank, self.n[kind], e & 0xFFFFFFFF, self.timeout_ms, int(self.bf16), _lib.stream_ptr())
        else:
            _lib.call("hx_allreduce_oneshot", self.reduced[kind].data_ptr(), self.bufs[kind, e & 1], self.flags[kind], self.status_ptr, self.world,
                      self.rank, self.n[kind], e & 0xFFFFFFFF, self.timeout_ms, _lib.stream_ptr())
        return self.reduced[kind]

    def check(self):
        """raises if a wait timed out since the last check (synchronises)"""
        torch.cuda.synchronize()
        st = torch.as_tensor(_DeviceWords(self.status_ptr, 1), device=self.device).view(torch.int32)
        code = int(st.item())
        if code != 0:
            raise _lib.HxError("one-shot exchange failed (sticky, every rank stops stepping): " +
                               ("a peer did not arrive within the timeout" if code == 1 else "a peer reported failure"))

    def close(self):
        for p in self._peers:
            _lib.call("hx_ipc_close", p)
        self._peers = []
        for p in self._own:
            _lib.call("hx_ipc_free", p)
        self._own = ()
