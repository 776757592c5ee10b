"""A gather through host memory over torch.distributed, for `Comm.custom` (gcsa2_comm_create_custom): the transport of
hosts without RCCL, and of world-size-2 / 3 runs on ONE GPU, where RCCL refuses ranks that share a device.  Everything above
the transport -- shards, the three gathers of the matching statistics, the totals / offsets / values exchange of locate()
with its CSR rebasing, the failure protocol -- is the library's C++ as with RCCL; only the bytes travel differently:
device -> host, `dist.send` / `dist.recv` of byte tensors (any backend that moves CPU tensors: gloo), host -> device."""
import ctypes as C


def _hip():
    """The HIP runtime already mapped into this process (torch's, which the engine shares)."""
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return C.CDLL(line.split()[-1])
    return C.CDLL("libamdhip64.so")


class HostGather:
    H2D, D2H, D2D = 1, 2, 3            # hipMemcpyKind

    def __init__(self, dist, rank: int, world: int):
        self.dist, self.rank, self.world = dist, rank, world
        self.hip = _hip()
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.calls = 0
        self.bytes_moved = 0

    def _check(self, code, what):
        if code != 0:
            raise RuntimeError(f"{what} failed with hipError {code}")

    def __call__(self, d_send, sizes, d_recv, root, stream):
        import torch
        self.calls += 1
        self._check(self.hip.hipStreamSynchronize(stream or None), "hipStreamSynchronize")      # what we send was produced on `stream`
        if self.rank == root:
            at = 0
            for r in range(self.world):
                n = sizes[r]
                if n > 0 and r == root:
                    if d_recv + at != d_send:
                        self._check(self.hip.hipMemcpy(d_recv + at, d_send, n, self.D2D), "hipMemcpy")
                elif n > 0:
                    buf = torch.empty(n, dtype=torch.uint8)
                    self.dist.recv(buf, src=r)
                    self._check(self.hip.hipMemcpy(d_recv + at, buf.data_ptr(), n, self.H2D), "hipMemcpy")
                    self.bytes_moved += n
                at += n
        elif sizes[self.rank] > 0:
            n = sizes[self.rank]
            buf = torch.empty(n, dtype=torch.uint8)
            self._check(self.hip.hipMemcpy(buf.data_ptr(), d_send, n, self.D2H), "hipMemcpy")
            self.dist.send(buf, dst=root)
            self.bytes_moved += n
        return 0


class AsyncHostGather(HostGather):
    """The same gather WITHOUT blocking the calling thread, so that the caller's stream choreography -- gather k on a second
    stream under kernel k + 1 on the first -- can be exercised (and measured) on one GPU: the call only ENQUEUES on `stream`
    and returns.  A peer: an asynchronous copy of its part into page-locked memory, an event; a helper thread waits for the
    event and sends.  The root: its own part by an asynchronous device copy; the helper thread receives the other parts
    into page-locked buffers; a host function enqueued on `stream` (hipLaunchHostFunc) holds the stream until they have
    arrived, and asynchronous copies into d_recv, enqueued behind it, bring them to the device.  Every call owns its
    buffers and its flag until the stream has passed its last copy (an event says so; any number of calls may be in flight).
    The bytes travel over a process group of their own: the helper thread's sends and receives never interleave with the
    collectives the main thread issues on the default group (barriers, all_gather_object)."""

    def __init__(self, dist, rank: int, world: int):
        import queue
        import threading
        import torch
        super().__init__(dist, rank, world)
        self.group = dist.new_group(backend="gloo")                # collective: every rank constructs the transport
        h = self.hip
        h.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        h.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        h.hipEventSynchronize.argtypes = [C.c_void_p]
        h.hipEventQuery.argtypes = [C.c_void_p]
        self.HOSTFN = C.CFUNCTYPE(None, C.c_void_p)
        h.hipLaunchHostFunc.argtypes = [C.c_void_p, self.HOSTFN, C.c_void_p]
        self.torch, self.threading = torch, threading
        self.lock = threading.Lock()
        self.free_buffers, self.free_events = [], []
        self.pending = {}                       # call id -> what the call owns until the stream has passed it
        self.jobs = queue.Queue()
        self.failed = None
        self.early_returns = 0                  # calls that returned before their bytes had moved (the evidence that nothing blocks)
        self._wait_cb = self.HOSTFN(self._wait_for_arrival)        # (kept alive: the runtime calls it from its own thread)
        self.worker = threading.Thread(target=self._work, daemon=True)
        self.worker.start()

    def _buffer(self, n):
        with self.lock:
            for k, buf in enumerate(self.free_buffers):
                if buf.numel() >= n:
                    return self.free_buffers.pop(k)
        return self.torch.empty(max(n, 1 << 16), dtype=self.torch.uint8).pin_memory()

    def _event(self):
        with self.lock:
            if self.free_events:
                return self.free_events.pop()
        ev = C.c_void_p()
        self._check(self.hip.hipEventCreateWithFlags(C.byref(ev), 2), "hipEventCreateWithFlags")        # hipEventDisableTiming
        return ev

    def _release(self, call):
        with self.lock:
            self.free_buffers.extend(call["buffers"])
            self.free_events.append(call["event"])

    def _reap(self):
        """Calls whose last copy the stream has passed give their buffers and events back."""
        for key in [k for k, c in self.pending.items() if c["done"] and self.hip.hipEventQuery(c["event"]) == 0]:
            self._release(self.pending.pop(key))

    def _wait_for_arrival(self, user):
        call = self.pending.get(int(user or 0))
        if call is not None:
            call["arrived"].wait()

    def _work(self):
        while True:
            job = self.jobs.get()
            if job is None:
                return
            try:
                if job[0] == "send":
                    _, call, buf, n, root = job
                    self._check(self.hip.hipEventSynchronize(call["event"]), "hipEventSynchronize")
                    self.dist.send(buf[:n], dst=root, group=self.group)
                    self._release(call)
                else:
                    _, call, parts = job
                    for r, buf, n in parts:
                        self.dist.recv(buf[:n], src=r, group=self.group)
                    call["arrived"].set()
            except Exception as e:              # a stream must never wait for ever on a transport that broke
                self.failed = e
                for c in list(self.pending.values()):
                    c["arrived"].set()

    def __call__(self, d_send, sizes, d_recv, root, stream):
        if self.failed is not None:
            return 1
        self.calls += 1
        st = stream or None
        h = self.hip
        if self.rank == root:
            self._reap()
            call = {"arrived": self.threading.Event(), "buffers": [], "event": self._event(), "done": False}
            at, parts, copies = 0, [], []
            for r in range(self.world):
                n = sizes[r]
                if n > 0 and r == root:
                    if d_recv + at != d_send:
                        self._check(h.hipMemcpyAsync(d_recv + at, d_send, n, self.D2D, st), "hipMemcpyAsync")
                elif n > 0:
                    buf = self._buffer(n)
                    call["buffers"].append(buf)
                    parts.append((r, buf, n))
                    copies.append((d_recv + at, buf.data_ptr(), n))
                    self.bytes_moved += n
                at += n
            if parts:
                self.pending[self.calls] = call
                self.jobs.put(("recv", call, parts))
                self._check(h.hipLaunchHostFunc(st, self._wait_cb, C.c_void_p(self.calls)), "hipLaunchHostFunc")
                for dst, src, n in copies:
                    self._check(h.hipMemcpyAsync(dst, src, n, self.H2D, st), "hipMemcpyAsync")
                self._check(h.hipEventRecord(call["event"], st), "hipEventRecord")
                call["done"] = True
                if not call["arrived"].is_set():
                    self.early_returns += 1
            else:
                self._release(call)
        elif sizes[self.rank] > 0:
            n = sizes[self.rank]
            call = {"buffers": [self._buffer(n)], "event": self._event()}
            buf = call["buffers"][0]
            self._check(h.hipMemcpyAsync(buf.data_ptr(), d_send, n, self.D2H, st), "hipMemcpyAsync")
            self._check(h.hipEventRecord(call["event"], st), "hipEventRecord")
            self.jobs.put(("send", call, buf, n, root))
            self.bytes_moved += n
            if h.hipEventQuery(call["event"]) != 0:             # hipErrorNotReady: the copy has not run yet, the call returns all the same
                self.early_returns += 1
        return 0

    def check(self):
        """Raises what the helper thread met, if anything.  A failure there releases every waiting stream -- the
        copies enqueued behind the wait then move whatever the page-locked buffers hold -- and the gather call that enqueued
        them has long returned 0, so the caller asks HERE, after synchronising the stream it gave the transport, before it
        trusts what arrived.  (A test-grade transport: RCCL is the product's; a peer whose send fails leaves the root's receive
        to the process group's timeout.)"""
        if self.failed is not None:
            raise RuntimeError(f"the host-memory gather failed in its helper thread: {self.failed!r}")

    def close(self):
        """(after the streams have drained) ends the helper thread, gives back the buffers and events of the calls that were
        still listed, and raises what the thread met"""
        self.jobs.put(None)
        self.worker.join(timeout=10)
        try:
            self._reap()
            for key in list(self.pending):
                self._release(self.pending.pop(key))
        except Exception:                       # (closing must not hide the failure below)
            pass
        self.check()
