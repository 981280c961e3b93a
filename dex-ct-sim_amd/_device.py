"""torch-ROCm as the device container: allocation, streams, pointers.  No compute happens here."""
import collections
import os
import threading
import weakref

import numpy as np
import torch

from ._native import DexctError


def device():
    if not torch.cuda.is_available():
        raise DexctError('no HIP device visible: the dex-ct hot path runs on an MI355X only (no CPU fallback)')
    idx = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    torch.cuda.set_device(idx)
    return torch.device('cuda', idx)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def to_dev(a, dtype, dev):
    """NumPy array or torch tensor -> contiguous device tensor of dtype."""
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)


def to_host(t):
    """Device tensor -> NumPy array through PAGE-LOCKED host memory (one DMA at PCIe rate instead of a staged copy
    through pageable memory).  The pinned buffer comes from torch's caching host allocator and is referenced by the
    returned array alone: the caller owns it like any fresh array, and the allocator reuses the pages once the array
    is garbage, so repeated calls do not pin new memory."""
    if not t.is_cuda:
        return t.numpy()
    host = pinned_empty(t.shape, t.dtype)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy()


def pinned_empty(shape, dtype):
    """Page-locked host tensor; pageable if the host refuses to lock that much memory (the copy is then staged by the
    runtime: slower, same result)."""
    try:
        return torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
    except RuntimeError:
        return torch.empty(tuple(shape), dtype=dtype)


# ---- the streams of the pipelined host boundary ---------------------------------------------------------------------------------
# The runtime maps streams onto a few hardware queues in the order they are made, and two streams on one queue take turns.  A pair
# of fresh streams per call beside the kernels on the default stream landed on the default stream's queue every other call:
# get_basismat_sinos alternated between 0.157 and 0.180 s (tools/probes/boundary_streams.py, profiles/r05_notes_boundary.md).
# Three streams made back to back, once per device - compute, upload, download - sit on three different queues whatever the
# default stream shares; the pipelined calls run their kernels on the first and wait for the caller's stream at entry.
_side = {}


def side_streams(dev):
    """(compute, upload, download) streams of ``dev`` for the pipelined boundary; the same three for every call."""
    key = str(dev)
    if key not in _side:
        _side[key] = tuple(torch.cuda.Stream(dev) for _ in range(3))
    return _side[key]


def quiesce():
    """Wait for everything queued on the side streams and the current stream: the error paths of the pipelined boundary call
    this before host memory that a queued copy may still read or write is unlocked or let go."""
    for streams in list(_side.values()):
        for st in streams:
            st.synchronize()
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.current_stream().synchronize()


# ---- large results and large inputs: host memory made ready for DMA while the GPU works ---------------------------------------
# Results cross PCIe by DMA into page-locked host memory that the returned arrays own.  A page-locked allocation (torch's, or
# hipHostMalloc) of memory the process has never touched runs at 11 GB/s - 0.57 s for the 6.5 GB of the benchmark's
# decomposition, 0.3 s for a sinogram pair - in front of everything, and one call of each kind is all the reference's main.py
# makes (main.py:120, :153).  What costs is the first touch of the pages (the kernel hands them out zeroed): 29 GB/s for one
# thread, 56 GB/s for two; memory that is RESIDENT locks at 500 GB/s and unlocks in microseconds (tools/probes/pin_resident.py,
# touch_threads.py; profiles/r05_notes_boundary.md).  So a LazyPinnedResult is a plain block of host memory that a helper
# thread, piece by piece and in the order the copies will come, touches (dexct_host_touch, new blocks only) and locks
# (dexct_host_pin) while the kernels and the copies of the pieces before run; finish() unlocks it, and the caller gets an ordinary
# NumPy array over ordinary memory (a block that stayed locked in several pieces would be a trap: one copy across two
# separately locked regions is an error of the HIP runtime, also for the caller's own torch code).  When the last array that
# views the block is garbage it goes to this module's pool - resident, so the next result of its size has nothing to touch.
# Large INPUT arrays get the same treatment for the time of a call (locked_arrays): resident by nature, locked in milliseconds,
# uploaded by DMA instead of through the runtime's staging buffers.
LAZY_MIN_BYTES = 64 << 20       # smaller results: torch's page-locked allocation (milliseconds)


def _default_pool_gb():
    """A quarter of the memory the machine has free when the module loads, between 8 and 64 GB (round 5 kept up to 64 GB of a
    process's dropped results resident whatever the machine: advisor finding); DEXCT_HOST_POOL_GB sets it explicitly."""
    try:
        with open('/proc/meminfo') as f:
            kb = next(int(ln.split()[1]) for ln in f if ln.startswith('MemAvailable:'))
        return min(64.0, max(8.0, kb / 4.0 / (1 << 20)))
    except (OSError, StopIteration, ValueError):
        return 16.0


POOL_MAX_BYTES = int(float(os.environ.get('DEXCT_HOST_POOL_GB', _default_pool_gb())) * (1 << 30))
_pool = {}                      # bytes -> [free blocks (np.uint8 arrays, page aligned, resident)]
_pool_lock = threading.Lock()
_PAGE = 4096
_TOUCH_THREADS = 2


def pool_wanted(n_bytes):
    """True when a result of n_bytes goes through LazyPinnedResult (large; DEXCT_LAZY_PIN=0 switches it off)."""
    return n_bytes >= LAZY_MIN_BYTES and os.environ.get('DEXCT_LAZY_PIN', '1') != '0'


_pool_age = []                  # sizes of the pooled blocks, oldest first: what leaves when the pool is over its limit
_returned = collections.deque() # blocks whose arrays became garbage, not yet in the pool (appended without the lock, see _give_back)


def _give_back(buf):
    """Runs as a weakref.finalize callback, i.e. wherever the garbage collector happens to run - also INSIDE a ``with
    _pool_lock:`` block of the same thread (a cyclic collection triggered by an allocation there), where taking the
    non-reentrant lock would block for ever (advisor finding of round 5).  So the callback only appends to a deque (atomic,
    no lock); the blocks enter the pool the next time somebody holds the lock anyway (_drain_returned)."""
    _returned.append(buf)


def _drain_returned():
    """under _pool_lock: the blocks that came back since the last call join the pool; the oldest leave while it is over its limit"""
    while True:
        try:
            buf = _returned.popleft()
        except IndexError:
            break
        if buf.nbytes > POOL_MAX_BYTES:
            continue
        _pool.setdefault(buf.nbytes, []).append(buf)
        _pool_age.append(buf.nbytes)
    total = sum(k * len(v) for k, v in _pool.items())
    while total > POOL_MAX_BYTES and _pool_age:
        old = _pool_age.pop(0)          # the block that has waited longest (of whatever size: sizes nobody asks for any more go first)
        if _pool.get(old):
            _pool[old].pop(0)
            total -= old


def pooled():
    """{block bytes: number of free blocks} after the returned blocks have joined the pool (tests, diagnostics)"""
    with _pool_lock:
        _drain_returned()
        return {k: len(v) for k, v in _pool.items() if v}


def empty_pool():
    """Drop every pooled block (back to the system).  Returns the bytes let go."""
    with _pool_lock:
        _drain_returned()
        n = sum(k * len(v) for k, v in _pool.items())
        _pool.clear()
        del _pool_age[:]
    return n


def _page_spans(ranges):
    """[(address, bytes)] -> the page-aligned spans that cover them, merged where they share or touch a page"""
    spans = sorted((addr // _PAGE * _PAGE, -(-(addr + n) // _PAGE) * _PAGE) for addr, n in ranges if n > 0)
    out = []
    for lo, hi in spans:
        if out and lo <= out[-1][1]:
            out[-1][1] = max(out[-1][1], hi)
        else:
            out.append([lo, hi])
    return [(lo, hi - lo) for lo, hi in out]


class locked_arrays:
    """``with locked_arrays(lib, [a1, a2], device_index): ...``: the memory of large contiguous NumPy arrays page-locked for the
    time of the block (uploads inside it run as DMA).  Whatever cannot be locked - it is locked already (a torch page-locked
    tensor's array), a file mapping, a limit - is left alone: the copy then goes the runtime's ordinary way."""

    def __init__(self, lib, arrays, device_index, min_bytes=16 << 20):
        self.lib, self.dev = lib, int(device_index)
        self.spans = _page_spans([(a.ctypes.data, a.nbytes) for a in arrays
                                  if isinstance(a, np.ndarray) and a.flags.c_contiguous and a.nbytes >= min_bytes])
        self.locked = []

    def __enter__(self):
        self.locked = [sp for sp in self.spans if self.lib.dexct_host_pin(sp[0], sp[1], self.dev) == 0]
        return self

    def __exit__(self, *exc):
        if exc and exc[0] is not None:
            quiesce()                       # an error inside the block: uploads from these arrays may still be queued
        for addr, _ in self.locked:
            self.lib.dexct_host_unpin(addr, self.dev)
        self.locked = []
        return False


class LazyPinnedResult:
    """``LazyPinnedResult(lib, shape, np_dtype, cuts, device_index)``: host memory for a result of ``shape`` whose byte ranges
    [cuts[k], cuts[k + 1]) are filled in order.  ``download(k, device_address, stream)`` queues the copy of piece k (waits until
    the piece is locked); ``finish()``, once the caller has synchronised the copies, unlocks and returns the NumPy array; the
    block returns to the pool when that array and its views are garbage.  ``fresh``: the block was not in the pool."""

    def __init__(self, lib, shape, np_dtype, cuts, device_index):
        self.lib, self.dev, self.shape, self.dtype = lib, int(device_index), tuple(shape), np.dtype(np_dtype)
        n_bytes = int(np.prod(self.shape)) * self.dtype.itemsize
        cuts = [int(c) for c in cuts]
        assert cuts[0] == 0 and cuts[-1] == n_bytes and all(a <= b for a, b in zip(cuts[:-1], cuts[1:]))
        with _pool_lock:
            _drain_returned()
            free = _pool.get(n_bytes)
            self.buf = free.pop() if free else None
            if self.buf is not None:
                _pool_age.remove(n_bytes)
        self.fresh = self.buf is None
        if self.fresh:
            raw = np.empty(n_bytes + 2 * _PAGE, dtype=np.uint8)
            off = (-raw.ctypes.data) % _PAGE
            self.buf = raw[off:off + n_bytes]
        base = self.buf.ctypes.data
        self.pieces = [(base + lo, hi - lo) for lo, hi in zip(cuts[:-1], cuts[1:])]
        # the pieces as page-aligned spans that tile the block without overlap (a page shared by two pieces belongs to the earlier
        # one: locking it twice is an error)
        edges = [base] + [-(-(addr + n) // _PAGE) * _PAGE for addr, n in self.pieces]
        self.spans = [(lo, hi - lo) for lo, hi in zip(edges[:-1], edges[1:])]
        self.locked = []
        self.ready = [threading.Event() for _ in self.pieces]
        threading.Thread(target=self._prepare, daemon=True).start()

    def _prepare(self):
        for k, (addr, n) in enumerate(self.spans):
            try:
                if n > 0:
                    if self.fresh:
                        self.lib.dexct_host_touch(addr, n, _TOUCH_THREADS)
                    # (a piece that cannot be locked - a locked-memory limit - is copied through the runtime's staging: same result)
                    if self.lib.dexct_host_pin(addr, n, self.dev) == 0:
                        self.locked.append(addr)
            finally:
                self.ready[k].set()         # whatever happened: the copy of piece k must not wait for ever

    def download(self, k, src_address, stream):
        """piece k of the result from device memory at ``src_address``: one copy per locked span it touches (a copy must stay
        inside one locked region; its first bytes may lie in the last page of a piece before)"""
        from . import _native
        self.ready[k].wait()
        addr, n = self.pieces[k]
        for lo, m in self.spans[:k + 1]:
            b, e = max(lo, addr), min(lo + m, addr + n)
            if e > b:
                _native.check(self.lib.dexct_download(b, src_address + (b - addr), e - b, stream.cuda_stream), 'dexct_download')

    def _unlock(self):
        for ev in self.ready:
            ev.wait()
        for addr in self.locked:
            self.lib.dexct_host_unpin(addr, self.dev)
        self.locked = []

    def __del__(self):
        # abandoned: an error between construction and finish().  Copies into the block may still be queued on the side streams
        # (gn_device or dexct_download raised in the middle of a pipeline): nothing may be unlocked, let alone handed to the next
        # result as its DMA target, before they have drained (advisor finding of round 5) - and the block is dropped, not pooled
        if getattr(self, 'buf', None) is not None and getattr(self, 'ready', None):
            try:
                quiesce()
                self._unlock()
            except Exception:               # (interpreter shutdown: nothing left to wait for)
                pass
            self.buf = None

    def finish(self):
        self._unlock()
        owner = np.frombuffer(memoryview(self.buf), dtype=self.dtype)     # (views of it keep IT alive: its base is no ndarray)
        weakref.finalize(owner, _give_back, self.buf).atexit = False
        self.buf = None
        return owner.reshape(self.shape)
