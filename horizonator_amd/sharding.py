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


# ---- a batch of viewpoints (BASELINE.json configs[3]) --------------------------

def viewpoint_slice(n, world_size, rank):
    """viewpoints [v0, v1) of `rank`: every rank holds the whole DEM mosaic and
    renders a contiguous block of the batch; renders are independent, so the
    only exchange is the gather of the finished images"""
    return sector_columns(n, world_size, rank)


def gather_viewpoints(images, n, group=None, dst=0):
    """images: this rank's [n_local, ...] tensor of finished panoramas.
    Returns the [n, ...] batch on `dst` (viewpoint order), None elsewhere.
    Blocks may differ in length by one; they travel padded to the longest."""
    world, rank = _world_and_rank(group)
    if world == 1:
        return images
    longest = -(-n // world)
    if images.shape[0] < longest:
        pad_shape = list(images.shape)
        pad_shape[0] = longest - images.shape[0]
        images = torch.cat([images, images.new_zeros(pad_shape)], dim=0)
    images = images.contiguous()
    bins = [torch.empty_like(images) for _ in range(world)] if rank == dst else None
    dist.gather(images, bins, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r, b in enumerate(bins):
        v0, v1 = viewpoint_slice(n, world, r)
        parts.append(b[:v1 - v0])
    return torch.cat(parts, dim=0)
