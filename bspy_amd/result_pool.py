"""
Recycled result arrays for the NumPy (host) call path.

A NumPy caller of ``Spline.evaluate`` gets a freshly allocated result array, as in the reference
(``bspy/spline.py:936-949`` returns new arrays).  For a large batch most of that call is not the GPU and not
PCIe: writing into fresh pages first-touches them (10 M bicubic points = 240 MB = 60 k page faults: 49.8 ms
against 6.8 ms into an array that has been written before, ``bench.py end_to_end_host``).  This pool keeps the
memory of result arrays the caller has DROPPED and hands it out again, so a loop of ``x, y, z = s(u, v)`` pays the
page faults once.

Safety rule: memory is recycled only when NO array refers to it any more.  A lent array is created over a
``_Lease`` object (through ``__array_interface__``), so ``result.base`` is the lease and every view of the result
(``x, y, z`` rows, reshapes, slices) keeps the result - and with it the lease - alive; NumPy collapses view bases
only through ndarrays, never through the lease.  The buffer goes back to the pool in the lease's finalizer, i.e.
after the last view is gone.  Small results (< ``MIN_BYTES``) are plain ``np.empty``.
"""
import threading
import weakref

import numpy as np

MIN_BYTES = 1 << 22          # below 4 MB the page faults do not matter
MAX_FREE_BUFFERS = 4         # free buffers kept per pool ...
MAX_FREE_BYTES = 2 << 30     # ... and their total size


class _Lease:
    """Owner of one lent buffer: exposes its memory to NumPy; returns it to the pool when the last array dies."""
    __slots__ = ("_buf", "__array_interface__", "__weakref__")

    def __init__(self, buf, nbytes):
        self._buf = buf
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (buf.ctypes.data, False), "version": 3}


class ResultPool:
    def __init__(self):
        self._free = []              # touched uint8 arrays, any size
        self._lock = threading.Lock()
        self.recycled = 0            # statistics (tests, bench)
        self.allocated = 0

    def _give_back(self, buf):
        with self._lock:
            total = sum(b.nbytes for b in self._free) + buf.nbytes
            if len(self._free) < MAX_FREE_BUFFERS and total <= MAX_FREE_BYTES:
                self._free.append(buf)

    def empty(self, shape, dtype):
        """Like ``np.empty(shape, dtype)`` (C order); large arrays reuse the memory of dropped results."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < MIN_BYTES:
            return np.empty(shape, dtype)
        buf = None
        with self._lock:
            # smallest free buffer that fits (and is not grossly larger)
            fits = [i for i, b in enumerate(self._free) if nbytes <= b.nbytes <= 2 * nbytes + (1 << 20)]
            if fits:
                buf = self._free.pop(min(fits, key=lambda i: self._free[i].nbytes))
                self.recycled += 1
        if buf is None:
            buf = np.empty(nbytes, np.uint8)
            self.allocated += 1
        lease = _Lease(buf, nbytes)
        weakref.finalize(lease, self._give_back, buf)      # runs when the lease (= the last array over it) is gone
        arr = np.asarray(lease)                             # arr.base is the lease
        return arr.view(dtype).reshape(shape)


_default = ResultPool()


def default_pool():
    return _default
