"""Placement-aware assignment of output rasters (net-new; no reference counterpart -- the reference round-trips every
raster through the host).

Measured on MI355X (tools/placement_probe.py, profiles/r3/placement_*): large device allocations fall
into three CONFLICT CLASSES.  Two rasters of one class, written concurrently at equal offsets -- what every kernel
with several output rasters does -- get ~6.1 TB/s of write bandwidth; rasters of different classes ~7 TB/s (pairs:
0.345 vs 0.30 ms per 2 x 1 GiB; the L2 -> DRAM write-credit stalls of the busiest L2 channels are 5-10 x higher in
the first case, request counts are identical, sub-page shifts of the bases change nothing).  The class is a property
of where the driver put the allocation in physical memory: consecutive allocations share a class for many GiB (the
first one or two of a process are usually of another class than the dozen that follow), so a chain that simply
allocates its rasters one after the other gets slope, TI and MTI -- the three outputs of the fused stencil -- in ONE
class, and the stencil runs at 66-68 % of the 8 TB/s figure instead of 75-80 %.

This module labels raster-sized blocks by class with a few timed launches of the library's write-only kernel
(dt_dev_membench_mix, ~1 ms per pair at 1 GiB) and hands the blocks to the roles that are written together so that
no such group sits in a single class; when the blocks at hand are all of one class it allocates a bounded number of
further candidates and keeps the ones that add a class.  Everything is measured at set-up time, nothing is assumed
about the address map."""
from . import _lib
from ._lib import check

MIN_BYTES = 64 << 20          # smaller rasters live in the caches' shadow: not worth a measurement
PAIR_CONFLICT = 1.94          # a pair slower than this many single-stream times conflicts (1.8 x when it does not,
                              # 2.08 x when it does)


class WriteClassifier:
    """conflict classes of device blocks of `nbytes` bytes on a context"""

    def __init__(self, ctx, nbytes):
        self.ctx, self.L = ctx, _lib.lib()
        n = min(int(nbytes), 1 << 30) // 4
        self.n = n // 65536 * 65536          # the write kernel walks rows of 16384 floats in groups of 4
        self.reps = []                        # one representative pointer per class found so far
        self.single_ms = None
        self.ratios = []                      # pair time / single-stream time of every comparison made (diagnostics)

    def usable(self):
        return self.n * 4 >= MIN_BYTES

    def _time(self, ptrs, reps=3):
        L, c = self.L, self.ctx
        w = list(ptrs) + [ptrs[0]] * (3 - len(ptrs))

        def launch():
            check(L.dt_dev_membench_mix(c.h, None, None, w[0], w[1], w[2], self.n, 0, len(ptrs), 1))
        launch()
        c.sync()
        import time
        t0 = time.perf_counter()
        for _ in range(reps):
            launch()
        c.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    def label(self, ptr):
        """class id of the block at device pointer `ptr` (its contents are overwritten)"""
        ptr = int(ptr)
        if self.single_ms is None:
            self.single_ms = min(self._time([ptr]), self._time([ptr]))
        for k, r in enumerate(self.reps):
            ratio = self._time([r, ptr]) / self.single_ms
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


def assign(ctx, nbytes, blocks, groups, extra_alloc=None, extra_release=None, budget=24, spacer_alloc=None,
           spacer_release=None, spacer_budget=96 << 30):
    """blocks: device pointers of equally sized blocks (>= one per role); groups: role names written together.
    Returns ({role: pointer}, info) -- or (None, info) when the rasters are too small to bother.  When the blocks
    span fewer classes than the largest group could use and `extra_alloc()` -> pointer is given, up to `budget`
    further blocks are allocated and labelled; those that add diversity replace blocks of the majority class
    (`extra_release(pointer)` gets every block that ends up unused, original or extra).
    A class holds for runs of 8-32 GiB of consecutive allocations (profiles/r3/placement_classes.txt: 120 x 1 GiB on
    one box read ABBAABBBBBBBBBBBB AAAAAAAA B x16 C x16 A x16 B x32 A x15), so candidates allocated back to back can
    all be alike: with `spacer_alloc(nbytes)` -> handle a candidate that adds nothing is followed by a spacer
    (4, 8, 16, 16, ... GiB, `spacer_budget` bytes in all, released at the end) that moves the next candidate on,
    and one that does add a class is followed by its neighbours, which share its run.  Spacers are not free: the
    runtime takes ~0.2 s to allocate 16 GiB and DEFERS the release -- some later allocation of the process pays for it
    (4.9 s after 160 GiB, measured) -- which is why the budget is 96 GiB and why Chain / bench.py allocate and free a
    block right after the search (the wait then belongs to the set-up that caused it, not to an innocent caller)."""
    blocks = [b.value if hasattr(b, "value") else int(b) for b in blocks]
    cl = WriteClassifier(ctx, nbytes)
    if not cl.usable():
        return None, {"tuned": False, "why": "rasters below %d MiB" % (MIN_BYTES >> 20)}
    labels = {int(b): cl.label(b) for b in blocks}
    want = min(3, max(len(g) for g in groups))   # three classes exist on MI355X; a group of 3 can use them all
    need = sum(len(g) * 2 // 3 for g in groups if len(g) > 1)  # blocks outside the commonest class: 2/3 of each group
    spacers, spaced = [], 0

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
    tried, step, full = 0, 4 << 30, False

    def try_alloc(fn, *a):
        nonlocal full
        try:
            return fn(*a)
        except (MemoryError, RuntimeError):  # the device is full: work with what was found
            full = True
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
                continue  # a rarer class: its neighbours in allocation order share the run -- keep allocating
            if spaced + step > spacer_budget:
                spacer_alloc = None  # out of spacer budget: back-to-back candidates for what is left of `budget`
                continue
            h = try_alloc(spacer_alloc, step)
            if h is None:
                full, spacer_alloc = False, None  # no room for spacers: carry on without them
                continue
            spacers.append(h)
            spaced += step
            step = min(step * 2, 16 << 30)
    finally:
        if spacer_release is not None:
            for h in spacers:
                spacer_release(h)
    # Everything in ONE class after a search over tens of GiB is more likely a bad yardstick than a property of the
    # memory: classes are defined by conflict with a representative, and a representative that straddles two runs (the
    # first blocks a process allocates often do) conflicts with everybody.  Label once more, starting from the other end.
    relabelled = False
    if len(cl.reps) == 1 and len(labels) >= 6:
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
    return roles, {"tuned": True, "classes": classes, "candidates_tried": tried, "blocks": len(labels),
                   "spacer_GiB": spaced >> 30, "single_stream_ms": round(cl.single_ms, 4), "n_classes": len(cl.reps),
                   "relabelled": relabelled, "pair_ratio_min_max": [min(cl.ratios), max(cl.ratios)] if getattr(cl, "ratios", None) else None}
