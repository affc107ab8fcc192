"""Device context and buffers over the C ABI's device tier (no torch needed; torch tensors'
data_ptr() can be passed to the same entry points)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class PinnedPool:
    """Page-locked host rasters (dt_host_alloc) behind numpy arrays.  Device-to-host copies into pageable memory
    run at a fraction of the PCIe rate, and locking pages costs more than the copy, so blocks are recycled: when
    the last reference to an array dies its block returns to the pool, and a later array of the same size class
    gets it.  chain.run_host's rasters come from here."""

    def __init__(self):
        import os
        self._free = {}   # bytes (size class) -> [address]
        self._idle = 0    # bytes sitting in _free
        # idle blocks beyond this budget are unlocked and freed when they come back (DT_PINNED_CACHE_MB, default
        # 4096; 0 = recycle nothing): one run_host on a 16384^2 DEM would otherwise leave ~20 GiB page-locked
        self.budget = max(int(os.environ.get("DT_PINNED_CACHE_MB", "4096")), 0) << 20
        # page-locked bytes in the callers' hands at any moment: beyond this cap (DT_PINNED_MAX_MB, default 32768) an
        # array is plain pageable numpy memory -- a workflow that keeps ten float64 rasters of 16384^2 alive would
        # otherwise hold 20 GiB of the host locked
        self.max_live = max(int(os.environ.get("DT_PINNED_MAX_MB", "32768")), 0) << 20
        self._live = 0

    @staticmethod
    def _klass(nbytes):
        """size class of a block: 4 KiB granules up to 64 MiB, 2 MiB granules beyond (powers of two, the first form,
        locked 2 GiB for a 1.01 GiB raster)"""
        g = (2 << 20) if nbytes > (1 << 26) else 4096
        return max(4096, (int(nbytes) + g - 1) // g * g)

    def empty(self, shape, dtype):
        import weakref
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        k = self._klass(max(n, 1))
        if self._live + k > self.max_live:
            return np.empty(shape, dtype)
        lst = self._free.get(k)
        if lst:
            addr = lst.pop()
            self._idle -= k
        else:
            out = C.c_void_p()
            check(_lib.lib().dt_host_alloc(k, C.byref(out)))
            addr = out.value
        self._live += k
        buf = (C.c_char * k).from_address(addr)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        weakref.finalize(buf, self._give_back, k, addr)  # buf lives as long as any view of it
        return arr

    def _give_back(self, k, addr):
        self._live -= k
        if self._idle + k > self.budget:
            _lib.lib().dt_host_free(C.c_void_p(addr))
            return
        self._free.setdefault(k, []).append(addr)
        self._idle += k

    def trim(self):
        for k, lst in self._free.items():
            while lst:
                check(_lib.lib().dt_host_free(C.c_void_p(lst.pop())))
        self._idle = 0


PINNED = PinnedPool()

HOST_POOL_MIN = 32 << 20   # rasters from this size come from the page-locked pool


def host_empty(shape, dtype):
    """Host array for a raster the library is about to fill: page-locked pool memory from 32 MiB (a device-to-host
    copy into fresh pageable memory runs at a third of the PCIe rate, and faulting the pages in costs as much
    again), plain numpy below."""
    dtype = np.dtype(dtype)
    if int(np.prod(shape)) * dtype.itemsize >= HOST_POOL_MIN:
        return PINNED.empty(shape, dtype)
    return np.empty(shape, dtype)


def widen64(a32):
    """float32 raster -> the float64 container the reference returns (values unchanged): threaded on the host for
    large rasters, numpy's astype below"""
    a32 = np.ascontiguousarray(a32, np.float32)
    if a32.nbytes < HOST_POOL_MIN:
        return a32.astype(np.float64)
    out = PINNED.empty(a32.shape, np.float64)
    check(_lib.lib().dt_host_f32_to_f64(a32.ctypes.data_as(C.POINTER(C.c_float)),
                                        out.ctypes.data_as(C.POINTER(C.c_double)), a32.size))
    return out


class DeviceArray:
    def __init__(self, ctx, shape, dtype):
        self.ctx, self.shape, self.dtype = ctx, tuple(np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(_lib.lib().dt_dev_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p

    @property
    def size(self):
        return int(np.prod(self.shape))

    def copy_from(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.size == self.size
        check(_lib.lib().dt_dev_h2d(self.ctx.h, self.ptr, a.ctypes.data_as(C.c_void_p), self.nbytes))
        return self

    def to_host(self, pinned=False):
        """copy to a fresh host array (pinned=True: page-locked, from the recycling pool: PCIe at full rate)"""
        out = PINNED.empty(self.shape, self.dtype) if pinned and self.nbytes >= (1 << 20) else np.empty(self.shape, self.dtype)
        check(_lib.lib().dt_dev_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def to_host_async(self):
        """enqueue the copy into a page-locked array and return it at once: its contents are valid after ctx.sync().
        The next array can be allocated (~12 ms per GiB) while this one crosses PCIe."""
        if self.nbytes < (1 << 20):  # small: a plain array, copied synchronously
            return self.to_host()
        out = PINNED.empty(self.shape, self.dtype)
        check(_lib.lib().dt_dev_d2h_async(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        # the DMA is in flight: the context holds the array until its next sync(), so that a caller who drops it early
        # cannot send its block back to the pool (and on to somebody else) under the copy
        self.ctx._inflight.append(out)
        return out

    def free(self):
        if self.ptr:
            check(_lib.lib().dt_dev_free(self.ctx.h, self.ptr))
            self.ptr = None


class Context:
    """dt_ctx wrapper: one device, one stream (own, or an external hipStream_t such as
    torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None, priority=None):
        """priority (own stream only): -1 high, 0 normal, +1 low -- see dt_ctx_set_priority"""
        h = C.c_void_p()
        check(_lib.lib().dt_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device
        self._inflight = []  # host arrays of asynchronous copies not yet waited for (DeviceArray.to_host_async)
        if priority is not None:
            check(_lib.lib().dt_ctx_set_priority(self.h, int(priority)))

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        return DeviceArray(self, a.shape, a.dtype).copy_from(a)

    def sync(self):
        check(_lib.lib().dt_ctx_sync(self.h))
        self._inflight.clear()

    def status(self):
        """sticky DT_STATUS_* bits raised by kernels since the last call (synchronises, clears)"""
        out = C.c_int32(0)
        check(_lib.lib().dt_ctx_status(self.h, C.byref(out)))
        return int(out.value)

    def raise_on_status(self):
        st = self.status()
        if st & 1:
            raise OverflowError("flow accumulation reached 2^31 cells: the int32 accumulation rasters of this "
                                "step are not valid (DT_STATUS_ACC_OVERFLOW)")
        if st & 2:
            raise RuntimeError("hydrological conditioning did not reach its fixed point within the budget of rounds "
                               "(DT_STATUS_NOT_CONVERGED): the rasters of this step are not valid -- raise "
                               "Chain(condition_rounds=...) or use flowdir.d8_conditioned, which iterates to the end")

    def fork(self, child):
        """`child`'s stream waits (on the device) for everything enqueued so far on this context's stream."""
        check(_lib.lib().dt_ctx_fork(self.h, child.h))

    def join(self, child):
        """this context's stream waits (on the device) for everything enqueued so far on `child`'s."""
        check(_lib.lib().dt_ctx_join(self.h, child.h))

    @property
    def stream(self):
        return _lib.lib().dt_ctx_stream(self.h)

    def close(self):
        if self.h:
            check(_lib.lib().dt_ctx_destroy(self.h))
            self.h = None


def trim():
    """Release what the package caches between calls: the host tier's idle device blocks (dt_host_trim) and the
    idle page-locked host blocks behind run_host's arrays.  Both caches are bounded anyway (DT_HOST_CACHE_MB,
    DT_PINNED_CACHE_MB)."""
    check(_lib.lib().dt_host_trim())
    PINNED.trim()
