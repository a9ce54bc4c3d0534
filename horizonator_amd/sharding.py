"""Sharding across the GPUs of a node: one panorama by azimuth sector, or a
batch of viewpoints by viewpoint.

The reference has no multi-device code; this is the multi-GPU design of this
build (DESIGN.md "Multi-GPU").  Every rank holds the full DEM mosaic and the
full-panorama view, and renders only image columns [col0, col1): its azimuth
sector.  Pixels are independent given the DEM, so a sector's pixels are
bit-identical to the same pixels of a single-GPU render, and the only exchange
is one gather of the finished strips to rank 0 (RCCL over xGMI when the process
group's backend is "nccl"; gloo on CPU in the tests).
"""
import ctypes as C

import torch
import torch.distributed as dist


def _world_and_rank(group):
    """(1, 0) in a process that runs alone (no process group), so that the same
    caller code serves one GPU and many"""
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def sector_columns(width, world_size, rank, weights=None):
    """columns [col0, col1) of `rank`.  Default: equal azimuth spans, remainder
    spread over the first ranks.  With a viewer-centred square mosaic and a
    360-degree view, 2/4/8 equal sectors starting at the image edge also hold
    equal DEM area (whole octants), so the triangle load is balanced.

    weights (one per rank, >= 0): spans proportional to the weights instead - for
    a gathering rank that has the conversion of the whole panorama to do on top
    of its own sector and should therefore draw less, or (weight 0) nothing."""
    if not (0 <= rank < world_size):
        raise ValueError("rank outside the world")
    if weights is None:
        base, extra = divmod(width, world_size)
        col0 = rank * base + min(rank, extra)
        col1 = col0 + base + (1 if rank < extra else 0)
        return col0, col1
    if len(weights) != world_size or min(weights) < 0 or sum(weights) <= 0:
        raise ValueError("need one non-negative weight per rank, not all zero")
    total = float(sum(weights))
    edges = [0]
    acc = 0.0
    for w in weights:
        acc += w
        edges.append(int(round(width * acc / total)))
    edges[-1] = width
    return edges[rank], edges[rank + 1]


def balanced_layout(density, world_size, weights=None):
    """[(col0, col1)] per rank such that every rank's columns hold the same share of `density`
    (one non-negative number per image column: the work behind that column), scaled by
    `weights` if given.  Equal azimuth spans are not equal work: the DEM window is square in
    cells but cells are not square in metres, and the window's corners are further away than
    its edges, so the terrain behind a column depends on its azimuth (azimuth_density())."""
    import numpy as np
    density = np.asarray(density, np.float64)
    width = density.size
    if weights is None:
        weights = [1.0] * world_size
    if len(weights) != world_size or min(weights) < 0 or sum(weights) <= 0 or density.min() < 0 or density.sum() <= 0:
        raise ValueError("need non-negative weights and densities, not all zero")
    cum = np.concatenate([[0.0], np.cumsum(density)])
    targets = np.cumsum(np.asarray(weights, np.float64)) / float(sum(weights)) * cum[-1]
    edges = [0] + [int(np.searchsorted(cum, t, side="left")) for t in targets[:-1]] + [width]
    edges = [min(max(e, 0), width) for e in edges]
    for k in range(1, len(edges)):
        edges[k] = max(edges[k], edges[k - 1])
    return [(edges[r], edges[r + 1]) for r in range(world_size)]


def azimuth_density(width, az_deg0, az_deg1, cos_lat, floor=0.5):
    """work behind each image column of a panorama over a square, viewer-centred DEM window:
    proportional to the squared distance from the viewer to the window's border along the
    column's azimuth (cells are 1 x cos_lat in metres), plus `floor` times the mean for what
    does not depend on the terrain behind (near field, conversion)"""
    import numpy as np
    az = np.radians(az_deg0 + (az_deg1 - az_deg0) * (np.arange(width) + 0.5) / width)
    r = np.minimum(1.0 / np.maximum(np.abs(np.cos(az)), 1e-9), cos_lat / np.maximum(np.abs(np.sin(az)), 1e-9))
    w = r * r
    return w + floor * w.mean()


def gatherer_weights(world_size, draw_ms_per_panorama, convert_ms_per_panorama):
    """weights for sector_columns() when rank 0 also converts the gathered strips: with a
    sector costing (about) fixed + draw_ms*share and the conversion convert_ms, rank 0 and
    the others finish together if rank 0's weight is (b - c*(G-1))/(b + c) of theirs
    (b = draw_ms, c = convert_ms), or 0 when the conversion alone takes longer than a sector"""
    if world_size == 1:
        return [1.0]
    b, c = float(draw_ms_per_panorama), float(convert_ms_per_panorama)
    rho = max(0.0, (b - c * (world_size - 1)) / (b + c))
    return [rho] + [1.0] * (world_size - 1)


def _layout(width, world, weights, layout):
    return list(layout) if layout is not None else [sector_columns(width, world, r, weights) for r in range(world)]


def gather_strips(strip, width, group=None, dst=0, weights=None, layout=None):
    """strip: this rank's [H, SW, ...] tensor (device tensor for nccl/RCCL, CPU
    tensor for gloo).  Returns the assembled [H, width, ...] tensor on `dst`,
    None elsewhere.  Strips may differ in width by one column; they travel
    padded to the widest."""
    world, rank = _world_and_rank(group)
    if world == 1:
        return strip
    cols = _layout(width, world, weights, layout)
    widest = max(c1 - c0 for c0, c1 in cols)
    sw = strip.shape[1]
    if sw < widest:
        pad_shape = list(strip.shape)
        pad_shape[1] = widest - sw
        strip = torch.cat([strip, strip.new_zeros(pad_shape)], dim=1)
    strip = strip.contiguous()
    bins = [torch.empty_like(strip) for _ in range(world)] if rank == dst else None
    dist.gather(strip, bins, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r, b in enumerate(bins):
        c0, c1 = cols[r]
        parts.append(b[:, :c1 - c0])
    return torch.cat(parts, dim=1)


class PendingGather:
    """a gather of strips that is in flight (see gather_strips_async)"""

    def __init__(self, work, bins, width, world, strip, weights=None, layout=None):
        self._work, self._bins, self._width, self._world, self._weights = work, bins, width, world, weights
        self._cols = _layout(width, world, weights, layout)
        self._strip = strip                 # keeps the send buffer alive until the exchange is over

    def parts(self):
        """wait for the exchange; on the destination rank the strips as they arrived,
        [(tensor [H, widest, ...], col0, ncols), ...] in column order - for a consumer
        that reads them in place (horizonator.resolve_packed) instead of a copy into
        one image; None elsewhere"""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._bins is None:
            return None
        out = []
        for r, b in enumerate(self._bins):
            c0, c1 = self._cols[r]
            out.append((b, c0, c1 - c0))
        return out

    def result(self):
        """wait for the exchange; the assembled [H, width, ...] tensor on the
        destination rank, None elsewhere"""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._bins is None:
            return None
        parts = []
        for r, b in enumerate(self._bins):
            c0, c1 = self._cols[r]
            parts.append(b[:, :c1 - c0])
        return torch.cat(parts, dim=1)


def gather_strips_async(strip, width, group=None, dst=0, weights=None, layout=None):
    """gather_strips() without waiting: the exchange of panorama k runs (on
    RCCL's stream) while the caller renders panorama k+1 into another buffer.
    Call .result() on the returned handle before the strip's buffer is reused."""
    world, rank = _world_and_rank(group)
    if world == 1:
        done = PendingGather(None, None, width, world, strip)
        done.result = lambda: strip
        done.parts = lambda: [(strip, 0, width)]
        return done
    widest = max(c1 - c0 for c0, c1 in _layout(width, world, weights, layout))
    sw = strip.shape[1]
    if sw < widest:
        pad_shape = list(strip.shape)
        pad_shape[1] = widest - sw
        strip = torch.cat([strip, strip.new_zeros(pad_shape)], dim=1)
    strip = strip.contiguous()
    bins = [torch.empty_like(strip) for _ in range(world)] if rank == dst else None
    work = dist.gather(strip, bins, dst=dst, group=group, async_op=True)
    return PendingGather(work, bins, width, world, strip, weights, layout)


# ---- DEM distribution: one rank reads the tiles -----------------------------------

def broadcast_dem(window, mosaic, device=None, group=None, src=0):
    """`src` passes (window tuple of horizonator.window(), int16 mosaic [N,N] as a numpy array);
    the other ranks pass (None, None).  Every rank gets (window, mosaic as a numpy array): two
    broadcasts, the 6 window numbers and the N*N samples (over RCCL when `device` is a GPU)."""
    import numpy as np
    world, rank = _world_and_rank(group)
    if world == 1:
        return window, mosaic
    dev = device if device is not None else torch.device("cpu")
    w = torch.tensor([int(x) for x in window] if rank == src else [0] * 6, dtype=torch.int64, device=dev)
    dist.broadcast(w, src=src, group=group)
    window = tuple(int(x) for x in w.tolist())
    n = 2 * window[1]
    # the samples travel as bytes: neither RCCL nor gloo has a 16-bit integer type
    if rank == src:
        m = torch.from_numpy(np.ascontiguousarray(mosaic, np.int16).view(np.uint8).reshape(n, 2 * n)).to(dev)
    else:
        m = torch.empty((n, 2 * n), dtype=torch.uint8, device=dev)
    dist.broadcast(m, src=src, group=group)
    return window, m.cpu().numpy().view(np.int16).reshape(n, n)


# ---- sparse strips: terrain pixels only (include/hz_hip.h, hz_hip_pack_sparse) ---

def sparse_mask_stride(widest_sector):
    """mask words per row, common to all strips of a gather"""
    return -(-int(widest_sector) // 32)


def sparse_header_words(height, mask_stride):
    """words before the pixel data of a sparse strip: count, row bases, mask"""
    return 1 + int(height) + int(height) * int(mask_stride)


class PendingFlat:
    """an exchange of equally long 1-D buffers that is in flight (gather_flat_async)"""

    def __init__(self, work, bins, flat):
        self._work, self._bins, self._flat = work, bins, flat

    def tensors(self):
        """wait; on the destination rank the buffers of all ranks in rank order, None elsewhere"""
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._bins


def gather_flat_async(flat, group=None, dst=0):
    """gather 1-D buffers of one common length to `dst` without waiting (the sparse strips of a
    panorama, cut to the longest of them: the caller agrees on the length with an all_reduce)"""
    world, rank = _world_and_rank(group)
    flat = flat.contiguous()
    if world == 1:
        return PendingFlat(None, [flat], flat)
    bins = [torch.empty_like(flat) for _ in range(world)] if rank == dst else None
    work = dist.gather(flat, bins, dst=dst, group=group, async_op=True)
    return PendingFlat(work, bins, flat)


class StripExchange:
    """The gather of a panorama's sparse strips with nothing on the host in its way.

    A strip's length (header + one word per terrain pixel) is only known on the device once it is
    drawn.  Reading it back, agreeing on the longest with an all_reduce and reading that back too
    puts two host round trips and a collective on the critical path of every panorama.  Instead
    the ranks agree ONCE on a capacity (`cap`: the longest strip seen so far plus a margin)
    and every panorama sends exactly that many words: the strip's own first word says how many of
    them mean anything.  Buffers - send side and the gathering rank's bins - are allocated once per
    slot and reused.  Whether any strip did not fit travels beside the strips as one word - the longest
    strip's length -, all-reduced (MAX) asynchronously; it is looked at when the exchange is completed, a panorama
    later and off the critical path, and grow() then redoes that one exchange with more room.

    The gathering rank is chosen per exchange (post(..., dst=r)): with `any_dst` every rank keeps
    bins, and consecutive panoramas can be gathered - and converted - by different ranks, so that
    no rank has the conversion of every panorama to do on top of its own sector.

        ex = StripExchange(cap_words, full_words, header_words, device, nslots=2)
        ex.post(slot, d_strip)      # d_strip: the rank's full-size strip buffer (int32, 1-D), drawn on
                                    # the current stream or ordered before it; no host wait
        bins, overflow = ex.complete(slot)      # waits; bins on dst (one 1-D tensor per rank), else None
        if overflow: bins = ex.grow(slot, d_strip)

    Capacity, bins and flag are per slot: grow() gives `slot` more room and redoes its exchange
    while the exchanges of the other slots - still in flight, a panorama behind - keep the bins
    they were posted with; those slots take the new capacity over when they are posted next.
    """

    def __init__(self, cap_words, full_words, header_words, device, nslots=2, group=None, dst=0, any_dst=False,
                 collectives_even_alone=False):
        self.world, self.rank = _world_and_rank(group)
        # (a process group of one rank: go through the collectives all the same - bench.py --exchange-anyway)
        self.collective = self.world > 1 or (collectives_even_alone and dist.is_available() and dist.is_initialized())
        self.group, self.dst, self.device = group, dst, device
        self.any_dst = bool(any_dst)
        self.full, self.hdr = int(full_words), int(header_words)
        self.nslots = nslots
        self.cap = self._clamp(cap_words)
        self.slot_cap = [0] * nslots
        self.bins = [None] * nslots
        self.slot_dst = [dst] * nslots
        self.flags = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(nslots)]
        self.work = [None] * nslots
        # on a GPU the flag reaches the host through a stream of its own, so that looking at it
        # waits for that exchange only - not for whatever the caller has queued since
        self.cuda = torch.device(self.device).type == "cuda"
        if self.cuda:
            self.side = torch.cuda.Stream(device=self.device)
            self.flag_host = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(nslots)]
            self.flag_ready = [torch.cuda.Event() for _ in range(nslots)]
        self.resends = 0

    def _clamp(self, cap_words):
        return int(min(max(int(cap_words), self.hdr + 1), self.full))

    def _holds_bins(self):
        return self.any_dst or self.rank == self.dst

    def _fit_slot(self, slot):
        """the slot's bins at the agreed capacity (only ever called with no exchange of the slot in flight)"""
        if self.slot_cap[slot] == self.cap:
            return
        self.slot_cap[slot] = self.cap
        self.bins[slot] = [torch.empty(self.cap, dtype=torch.int32, device=self.device) for _ in range(self.world)] \
            if self._holds_bins() else None

    def post(self, slot, strip, dst=None):
        """start the exchange of `strip` (this rank's strip buffer, full_words long) in `slot`;
        dst: the rank that gathers this one (default: the exchange's)"""
        assert self.work[slot] is None, "complete() the slot's previous exchange first"
        dst = self.dst if dst is None else int(dst)
        assert dst == self.dst or self.any_dst, "a gathering rank other than the default needs any_dst=True"
        self._fit_slot(slot)
        cap = self.slot_cap[slot]
        self.slot_dst[slot] = dst
        # does every rank's strip fit?  The longest strip's length travels beside the strips (device-side: no value
        # leaves the device here; the comparison with the capacity is the host's, when it looks at the word later)
        self.flags[slot].copy_(strip[0:1])
        send = strip[:cap]
        if not self.collective:
            self.bins[slot][0].copy_(send)
            w1 = w2 = None
        else:
            w1 = dist.gather(send, self.bins[slot] if self.rank == dst else None, dst=dst, group=self.group, async_op=True)
            w2 = dist.all_reduce(self.flags[slot], op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.side):
                if w2 is not None:
                    w2.wait()                       # the side stream waits for the flag's all_reduce ...
                else:
                    self.side.wait_stream(cur)
                self.flag_host[slot].copy_(self.flags[slot], non_blocking=True)
                self.flag_ready[slot].record(self.side)
        self.work[slot] = (w1, w2)

    def complete(self, slot):
        """the slot's exchange is over: (bins on its gathering rank else None, did a strip not fit).
        On a GPU the caller's current stream waits for the strips (the host does not); the host waits
        for the flag word of THIS exchange only."""
        if self.work[slot] is None:
            return None, False
        w1, w2 = self.work[slot]
        self.work[slot] = None
        if w1 is not None:
            w1.wait()                               # nccl: the current stream waits; gloo: the host does
        if self.cuda:
            self.flag_ready[slot].synchronize()
            longest = int(self.flag_host[slot][0])
        else:
            if w2 is not None:
                w2.wait()
            longest = int(self.flags[slot][0])
        overflow = longest > self.slot_cap[slot] - self.hdr
        return (self.bins[slot] if self.rank == self.slot_dst[slot] else None), overflow

    def grow(self, slot, strip):
        """a strip of `slot`'s exchange did not fit: agree on a new capacity (every rank calls this -
        they all saw the same flag) and redo that exchange with it, waiting for it.  Only `slot` is
        reallocated: exchanges of other slots that are still in flight keep their bins and adopt
        the new capacity at their next post()."""
        assert self.work[slot] is None, "grow() follows complete() of the same slot"
        n = torch.tensor([self.hdr + int(strip[0].item())], dtype=torch.int64, device=self.device)
        if self.collective:
            dist.all_reduce(n, op=dist.ReduceOp.MAX, group=self.group)
        self.cap = self._clamp(int(int(n.item()) * 1.1) + 1024)
        self.resends += 1
        self.post(slot, strip, dst=self.slot_dst[slot])
        bins, overflow = self.complete(slot)
        assert not overflow
        return bins


def agree_on_capacity(words, header_words, full_words, device, group=None, margin=1.1):
    """one-off: the capacity for a StripExchange from this rank's current strip length"""
    world, _ = _world_and_rank(group)
    n = torch.tensor([int(words)], dtype=torch.int64, device=device)
    if world > 1:
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    longest = int(n.item())
    return int(min(full_words, header_words + (longest - header_words) * margin + 1024))


# ---- the panorama loop in C (include/horizonator_rccl.h) ---------------------

class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class _Series(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("rotate", C.c_int), ("nslots", C.c_int),
                ("d_strips", C.POINTER(C.c_void_p)), ("d_bins", C.POINTER(C.c_void_p)),
                ("words", C.c_size_t), ("header_words", C.c_size_t), ("mask_stride", C.c_int),
                ("col0", C.POINTER(C.c_int)), ("ncols", C.POINTER(C.c_int)),
                ("d_image", C.c_void_p), ("d_ranges", C.c_void_p), ("stream", C.c_void_p),
                ("exchange", C.c_void_p), ("exchange_user", C.c_void_p)]


_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p)


class RcclSeries:
    """horizonator_rccl_render_series() for a Python caller: an RCCL communicator of the library's own (the
    unique id travels through the torch.distributed group that is there anyway), strip buffers and bins per
    slot, and the loop over a series of panoramas as ONE call into C - no interpreter between two panoramas.

        rs = RcclSeries(h, layout, H, words, d_img, d_rng, rotate=True)   # h: this rank's context, set to its sector
        rs.run(first, count)        # queues `count` panoramas of the current view; returns at once
        rs.sync()                   # ... and waits for them
        rs.close()
    """

    def __init__(self, h, layout, height, words, d_image, d_ranges, rotate, device, nslots=2, group=None, transport="rccl"):
        """transport "gloo": the strips travel through host memory over torch.distributed's process group instead of over
        RCCL (horizonator_rccl_series_t::exchange) - slow, but ranks that SHARE a GPU can run it, which RCCL refuses: the
        C loop's slot and rotation logic with world > 1 on a box with one GPU (tests/test_gpu_bench_multi.py)"""
        import os
        from . import _lib
        self.world, self.rank = _world_and_rank(group)
        self.h, self.device, self.group = h, device, group
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        self._rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
        _lib.load()
        self._hz = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhorizonator_rccl.so"))
        self._rccl.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        self._rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        self._rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        self._hz.horizonator_rccl_render_series.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Series), C.c_long, C.c_int, C.c_int]
        self.comm = C.c_void_p()
        torch.cuda.set_device(device)
        if transport == "rccl":
            uid = _NcclUniqueId()
            if self.rank == 0 and self._rccl.ncclGetUniqueId(C.byref(uid)) != 0:
                raise RuntimeError("ncclGetUniqueId() failed")
            if self.world > 1:
                box = [C.string_at(C.addressof(uid), 128) if self.rank == 0 else None]      # (the raw 128 bytes: .internal as bytes would stop at a NUL)
                dist.broadcast_object_list(box, src=0, group=group)
                C.memmove(C.addressof(uid), box[0], 128)
            if self._rccl.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank) != 0:
                raise RuntimeError("ncclCommInitRank() failed")
        elif transport != "gloo":
            raise ValueError("transport: rccl or gloo")
        self.transport = transport
        self.nslots, self.rotate = int(nslots), bool(rotate)
        widest = max(c1 - c0 for c0, c1 in layout)
        self.mask_stride = sparse_mask_stride(widest)
        self.hdr = sparse_header_words(height, self.mask_stride)
        self.full = self.hdr + height * widest
        self.words = int(min(max(int(words), self.hdr + 1), self.full))
        self.stream = torch.cuda.Stream(device=device)
        # (torch.empty: the library writes a strip's count word itself, on its own stream)
        self.strips = [torch.empty(self.full, dtype=torch.int32, device=device) for _ in range(self.nslots)]
        gathers = self.rotate or self.rank == 0
        self.bins = [torch.empty(self.words, dtype=torch.int32, device=device) for _ in range(self.nslots * self.world)] if gathers else []
        self._strips = (C.c_void_p * self.nslots)(*[t.data_ptr() for t in self.strips])
        self._bins = (C.c_void_p * max(len(self.bins), 1))(*[t.data_ptr() for t in self.bins])
        self._col0 = (C.c_int * self.world)(*[int(c0) for c0, _ in layout])
        self._ncols = (C.c_int * self.world)(*[int(c1 - c0) for c0, c1 in layout])
        self._s = _Series(self.rank, self.world, int(self.rotate), self.nslots,
                          C.cast(self._strips, C.POINTER(C.c_void_p)), C.cast(self._bins, C.POINTER(C.c_void_p)) if gathers else None,
                          self.words, self.hdr, self.mask_stride, C.cast(self._col0, C.POINTER(C.c_int)), C.cast(self._ncols, C.POINTER(C.c_int)),
                          int(d_image) if d_image else None, int(d_ranges) if d_ranges else None, self.stream.cuda_stream, None, None)
        if transport == "gloo":
            self._by_ptr = {t.data_ptr(): t for t in self.strips + self.bins}
            self._cb = _EXCHANGE_FN(self._gloo_exchange)            # (kept alive with the object)
            self._s.exchange = C.cast(self._cb, C.c_void_p)
        self.next = 0

    def _gloo_exchange(self, user, root, d_send, words, d_recv, stream):
        """horizonator_rccl_series_t::exchange over the torch.distributed group (gloo), through host memory: complete when it returns"""
        try:
            self.stream.synchronize()                               # the strip is complete (the C loop ordered the stream behind its conversion)
            mine = self._by_ptr[d_send][:words].cpu()
            if self.rank == root:
                parts = [torch.empty(words, dtype=torch.int32) for _ in range(self.world)]
                dist.gather(mine, parts, dst=root, group=self.group)
                with torch.cuda.stream(self.stream):
                    for r, p in enumerate(parts):
                        self._by_ptr[d_recv[r]][:words].copy_(p.to(self.device))
                self.stream.synchronize()
            else:
                dist.gather(mine, None, dst=root, group=self.group)
            return 0
        except Exception as e:                                      # (an exception must not unwind through the C frames)
            import sys
            print("RcclSeries gloo exchange:", repr(e), file=sys.stderr)
            return -1

    def run(self, count, check_fit=False):
        """queue `count` more panoramas; with check_fit: wait for the last strip and return True if it fit the agreed words"""
        rc = self._hz.horizonator_rccl_render_series(C.byref(self.h._ctx), self.comm, C.byref(self._s), self.next, int(count), int(bool(check_fit)))
        if rc < 0:
            raise RuntimeError("horizonator_rccl_render_series() failed (message on stderr)")
        self.next += int(count)
        return rc == 0

    def sync(self):
        self.stream.synchronize()
        self.h.sync()

    def close(self):
        if self.comm:
            self.sync()
            self._rccl.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
        elif self.transport == "gloo":
            self.sync()


# ---- a batch of viewpoints (BASELINE.json configs[3]) --------------------------

def viewpoint_slice(n, world_size, rank):
    """viewpoints [v0, v1) of `rank`: every rank holds the whole DEM mosaic and
    renders a contiguous block of the batch; renders are independent, so the
    only exchange is the gather of the finished images"""
    return sector_columns(n, world_size, rank)


def gather_viewpoints(images, n, group=None, dst=0, chunk=16, out=None):
    """images: this rank's [n_local, ...] tensor of finished panoramas.
    Returns the [n, ...] batch on `dst` (viewpoint order), None elsewhere.

    The batch travels in chunks of at most `chunk` viewpoints per rank, through one pair of
    preallocated bins, each chunk copied straight to its place in the result: 256 panoramas of
    8000x2000 are 12.3 GB, which a single gather would hold twice on the gathering rank (bins +
    concatenation).  out: the [n, ...] result tensor on `dst`, if the caller has one."""
    world, rank = _world_and_rank(group)
    if world == 1:
        if out is None:
            return images
        out.copy_(images)
        return out
    longest = -(-n // world)
    chunk = max(1, min(int(chunk), longest))
    item_shape = list(images.shape[1:])
    if rank == dst and out is None:
        out = torch.empty([n] + item_shape, dtype=images.dtype, device=images.device)
    send = torch.zeros([chunk] + item_shape, dtype=images.dtype, device=images.device)
    bins = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    for c0 in range(0, longest, chunk):
        have = max(0, min(images.shape[0] - c0, chunk))
        if have:
            send[:have].copy_(images[c0:c0 + have])
        dist.gather(send, bins, dst=dst, group=group)
        if rank == dst:
            for r, b in enumerate(bins):
                v0, v1 = viewpoint_slice(n, world, r)
                m = max(0, min(v1 - v0 - c0, chunk))
                if m:
                    out[v0 + c0:v0 + c0 + m].copy_(b[:m])
    return out if rank == dst else None
