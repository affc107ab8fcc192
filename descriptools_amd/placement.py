"""Placement-aware assignment of output rasters (net-new; no reference counterpart -- the reference round-trips every
raster through the host).

What is measured on MI355X (profiles/r4/placement_map_*.txt, tools/micro/placement_map.hip):
  * Two rasters written concurrently at equal offsets -- what every kernel with several output rasters does -- run at
    one of two speeds, and which one is a property of the PHYSICAL memory behind them: the same hipMemCreate handles
    mapped at other virtual addresses, in another order, keep their behaviour (160 of 160).
  * Inside one physically contiguous stretch the slowdown of a pair is a function of the address DIFFERENCE with a
    period of 4 GiB (2.1-2.4 x the single-stream time around 0 .. +1.25 GiB, 1.65-1.7 x around +2 .. +3 GiB); between
    stretches that lie >= ~30 GiB apart there is none (1.45 x): a 48-GiB allocation crossed such a border 30.5 GiB in.
    These stretches are the "conflict classes" round 3 found (three of them, runs of 8-32 GiB in allocation order).
  * For the chain's multi-output kernels only the border matters: with slope / TI / MTI anywhere inside ONE stretch the
    fused stencil takes 0.99-1.03 ms whatever their spacing (profiles/r4/placement_arena_probe.txt), in three different
    stretches 0.85 ms.  There is no offset trick: rasters written together must come from different stretches.
So a chain that allocates its rasters one after the other gets all of them from one stretch and runs the stencil at
66-68 % of the 8 TB/s figure instead of 78-80 %.

This module labels raster-sized blocks by class with timed launches of the library's write-only kernel (HIP events,
dt_dev_membench_mix_timed) and hands the blocks to the roles that are written together so that no such group sits in a
single class.  Two levels:
  * `search=False` (what Chain / RankTile do by default): work with the blocks at hand -- ~100 probe launches, nothing
    allocated.  It helps exactly when the chain's own rasters straddle a border.
  * `search=True` (opt-in: Chain(tune_placement="search"), bench.py): when the blocks at hand are all alike, allocate
    further candidates with large transient spacers between them (a border is tens of GiB away) until another class
    turns up.  The memory this may grab is bounded -- `spacer_budget` (default: DT_PLACEMENT_SPACER_GIB, 96) and never
    more than 35 % of what hipMemGetInfo reports free -- all spacers and unused candidates are released before the
    call returns, and the seconds it took are reported (`setup_s`).
Everything is measured at set-up time; which block serves which raster changes no bit of any result."""
import os
import time

from . import _lib
from ._lib import check

MIN_BYTES = 64 << 20          # smaller rasters live in the caches' shadow: not worth a measurement
PAIR_CONFLICT = 1.94          # a pair slower than this many single-stream times conflicts (dt_dev_membench_mix: 1.8 x
                              # when it does not, 2.08 x when it does)
PAIR_BAND = 0.07              # a ratio this close to the threshold is measured again with more repetitions
FREE_FRACTION = 0.35          # of the device's free memory a search may hold at any moment


def _is_oom(e):
    """a full device: the C ABI's DT_ENOMEM (MemoryError, _lib.check) or torch's OutOfMemoryError (a RuntimeError
    subclass) -- any other RuntimeError is a real fault and must not be read as 'device full'"""
    return isinstance(e, MemoryError) or type(e).__name__ == "OutOfMemoryError"


class WriteClassifier:
    """conflict classes of device blocks of `nbytes` bytes on a context"""

    def __init__(self, ctx, nbytes):
        self.ctx, self.L = ctx, _lib.lib()
        n = min(int(nbytes), 1 << 30) // 4
        self.n = n // 65536 * 65536          # the write kernel walks rows of 16384 floats in groups of 4
        self.reps = []                        # one representative pointer per class found so far
        self.single_ms = None
        self.ratios = []                      # pair time / single-stream time of every comparison made (diagnostics)
        self.rechecked = 0

    def usable(self):
        return self.n * 4 >= MIN_BYTES

    def _time(self, ptrs, reps=4):
        import ctypes as C
        w = list(ptrs) + [ptrs[0]] * (3 - len(ptrs))
        ms = C.c_double(0.0)
        check(self.L.dt_dev_membench_mix_timed(self.ctx.h, None, None, w[0], w[1], w[2], self.n, 0, len(ptrs), 1, reps,
                                               C.byref(ms)))
        return ms.value

    def label(self, ptr):
        """class id of the block at device pointer `ptr` (its contents are overwritten)"""
        ptr = int(ptr)
        if self.single_ms is None:
            self.single_ms = min(self._time([ptr]), self._time([ptr]))
        for k, r in enumerate(self.reps):
            ratio = self._time([r, ptr]) / self.single_ms
            if abs(ratio - PAIR_CONFLICT) < PAIR_BAND:  # too close to call: look again, longer
                ratio = self._time([r, ptr], reps=12) / self.single_ms
                self.rechecked += 1
            self.ratios.append(round(ratio, 3))
            if ratio > PAIR_CONFLICT:
                return k
        self.reps.append(ptr)
        return len(self.reps) - 1


def spread(labels, groups):
    """labels: {block: class}; groups: lists of role names written together, most important first.  Returns
    {role: block} using every block at most once, each group drawing from as many classes as possible (round robin
    over the classes, the rarest class first), and the list of blocks left over."""
    by_class = {}
    for b, k in labels.items():
        by_class.setdefault(k, []).append(b)
    out = {}
    for g in groups:
        # rare classes are spent on the groups that come first; a role written alone takes from the commonest class
        order = sorted(by_class, key=lambda k: len(by_class[k]), reverse=len(g) == 1)
        i = 0
        for role in g:
            for _ in range(len(order)):
                k = order[i % len(order)]
                i += 1
                if by_class[k]:
                    out[role] = by_class[k].pop()
                    break
            else:
                raise ValueError("fewer blocks than roles")
    return out, [b for lst in by_class.values() for b in lst]


def default_spacer_budget(ctx):
    """bytes a search may spend on transient spacers: DT_PLACEMENT_SPACER_GIB (default 96), and never more than
    FREE_FRACTION of the device's free memory"""
    import ctypes as C
    cap = max(int(os.environ.get("DT_PLACEMENT_SPACER_GIB", "96")), 0) << 30
    if ctx is None:
        return cap
    free = C.c_int64(0)
    check(_lib.lib().dt_dev_mem_info(ctx.h, C.byref(free), None))
    return min(cap, int(free.value * FREE_FRACTION))


def assign(ctx, nbytes, blocks, groups, extra_alloc=None, extra_release=None, budget=12, spacer_alloc=None,
           spacer_release=None, spacer_budget=None, search=True):
    """blocks: device pointers of equally sized blocks (>= one per role); groups: role names written together.
    Returns ({role: pointer}, info) -- or (None, info) when the rasters are too small to bother.

    search=False, or no `extra_alloc`: the blocks at hand are labelled and spread, nothing is allocated.
    search=True with `extra_alloc()` -> pointer: when the blocks span fewer classes than the largest group could use, up
    to `budget` further blocks are allocated and labelled; those that add diversity replace blocks of the majority
    class (`extra_release(pointer)` gets every block that ends up unused, original or extra).  A class holds for
    stretches of tens of GiB (module docstring), so candidates allocated back to back are all alike: with
    `spacer_alloc(nbytes)` -> handle a candidate that adds nothing is followed by a spacer (16, then 32 GiB each,
    `spacer_budget` bytes in all -- default_spacer_budget() -- released before this returns) that moves the next
    candidate on, and one that does add a class is followed by its neighbours, which share its stretch.  Spacers are
    not free: the runtime takes ~0.2 s per 16 GiB and DEFERS the release -- some later allocation of the process pays
    for it (4.9 s after 160 GiB, measured) -- which is why the search is opt-in, bounded, timed (`setup_s`), and why
    Chain / bench.py allocate and free a block right after it (the wait then belongs to the set-up that caused it)."""
    t_start = time.perf_counter()
    blocks = [b.value if hasattr(b, "value") else int(b) for b in blocks]
    cl = WriteClassifier(ctx, nbytes)
    if not cl.usable():
        return None, {"tuned": False, "why": "rasters below %d MiB" % (MIN_BYTES >> 20)}
    labels = {int(b): cl.label(b) for b in blocks}
    want = min(3, max(len(g) for g in groups))   # three classes exist on MI355X; a group of 3 can use them all
    need = sum(len(g) * 2 // 3 for g in groups if len(g) > 1)  # blocks outside the commonest class: 2/3 of each group
    spacers, spaced = [], 0
    if not search:
        extra_alloc = None
    if spacer_budget is None and extra_alloc is not None and spacer_alloc is not None:
        spacer_budget = default_spacer_budget(ctx)

    def counts():
        c = {}
        for k in labels.values():
            c[k] = c.get(k, 0) + 1
        return c

    def enough():
        c = counts()
        major = max(c, key=c.get)
        # every group can avoid a single class: two blocks outside the majority class per group and (if reachable)
        # three classes for the first group
        return len(c) >= want and sum(v for k, v in c.items() if k != major) >= need
    tried, step, full = 0, 16 << 30, False

    def try_alloc(fn, *a):
        nonlocal full
        try:
            return fn(*a)
        except Exception as e:
            if not _is_oom(e):
                raise  # a real fault is not "the device is full"
            full = True  # work with what was found
            return None
    try:
        while extra_alloc is not None and tried < budget and not full and not enough():
            before = counts()
            p = try_alloc(extra_alloc)
            if p is None:
                break
            p = int(p)
            tried += 1
            k = labels[p] = cl.label(p)
            major = max(before, key=before.get)
            if k != major or spacer_alloc is None:
                continue  # a rarer class: its neighbours in allocation order share the stretch -- keep allocating
            if spaced + step > spacer_budget:
                spacer_alloc = None  # out of spacer budget: back-to-back candidates for what is left of `budget`
                continue
            h = try_alloc(spacer_alloc, step)
            if h is None:
                full, spacer_alloc = False, None  # no room for spacers: carry on without them
                continue
            spacers.append(h)
            spaced += step
            step = 32 << 30
    finally:
        if spacer_release is not None:
            for h in spacers:
                spacer_release(h)
    # Everything in ONE class after a search over tens of GiB is more likely a bad yardstick than a property of the
    # memory: classes are defined by conflict with a representative, and a representative that straddles two stretches
    # (the first blocks a process allocates often do) conflicts with everybody.  Label once more, from the other end.
    relabelled = False
    if len(cl.reps) == 1 and len(labels) >= 6 and tried > 0:
        cl2 = WriteClassifier(ctx, nbytes)
        cl2.single_ms = cl.single_ms
        labels2 = {b: cl2.label(b) for b in reversed(list(labels))}
        if len(cl2.reps) > 1:
            labels, cl, relabelled = {b: labels2[b] for b in labels}, cl2, True
    roles, left = spread(labels, groups)
    if extra_release is not None:
        for b in left:
            extra_release(b)
    classes = {r: labels[b] for r, b in roles.items()}
    ratios = getattr(cl, "ratios", None)
    return roles, {"tuned": True, "mode": "search" if search and (tried or extra_alloc is not None) else "own blocks",
                   "classes": classes, "candidates_tried": tried, "blocks": len(labels),
                   "spacer_GiB": spaced >> 30, "spacer_budget_GiB": (spacer_budget or 0) >> 30,
                   "single_stream_ms": round(cl.single_ms, 4), "n_classes": len(cl.reps),
                   "relabelled": relabelled, "rechecked": getattr(cl, "rechecked", 0),
                   "pair_ratio_min_max": [min(ratios), max(ratios)] if ratios else None,
                   "setup_s": round(time.perf_counter() - t_start, 3)}
