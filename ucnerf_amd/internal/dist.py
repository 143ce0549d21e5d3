"""Ray-tile sharding and the one-collective-per-frame exchange (torch.distributed; nccl == RCCL).

ref /root/reference/nerf/internal/models.py:943-968: per 15000-ray chunk the reference pads to a
multiple of the world size, slices rank g's rows and all-gathers every tensor of every level
(O(10^3) small collectives per 1280x1920 frame).  Rays are independent and the weights are
replicated, so the only exchange the path needs is the finished pixels: here rank g marches the
contiguous row range [lo_g, hi_g) of the flattened frame and ONE all-gather per frame moves all
output buffers packed side by side ([rows_per_rank, sum of widths] floats; equal shard sizes by
padding the last rank, stripped after the gather -- the same padding rule as models.py:946-951,
applied once per frame).  xGMI is point-to-point: with shards of ~6 MB/rank the exchange is
per-link-latency bound, which is why it is a single large message and not one per tensor.
"""
import torch
import torch.distributed as dist


EXCHANGE_EVENTS = None          # a list while bench.py times frames: (start event, end event, bytes sent) per frame exchange


def rows_per_rank(num_rays, world):
    return (num_rays + world - 1) // world


def shard_bounds(num_rays, world, rank):
    rp = rows_per_rank(num_rays, world)
    lo = min(rank * rp, num_rays)
    return lo, min(lo + rp, num_rays)


def _active(world):
    return world > 1 and dist.is_available() and dist.is_initialized()


def _inplace_wanted():
    import os
    return os.environ.get("UCN_DIST_INPLACE", "0") == "1"


def _exchange(out, mine, rank, rp):
    """The frame's ONE collective: every rank's packed shard `mine` ([rp, width], a view of rows [rank * rp, (rank + 1) * rp)
    of `out`) into `out` on every rank.  Default: out-of-place -- the send buffer is a copy of the shard (11 MB per rank for
    a 1280 x 1920 frame on 8 ranks: microseconds), the form every backend and every torch / RCCL version implements.
    UCN_DIST_INPLACE=1 hands RCCL the in-place form (send buffer = receive buffer + rank * count, no copy); its
    precondition is asserted, and any backend error falls back to the out-of-place call."""
    if _inplace_wanted() and dist.get_backend() == "nccl":
        assert mine.is_contiguous() and out.is_contiguous()
        assert mine.data_ptr() == out.data_ptr() + rank * rp * out.shape[1] * out.element_size(), "in-place all-gather: shard is not at its slot"
        try:
            dist.all_gather_into_tensor(out, mine)
            return
        except RuntimeError:
            pass
    dist.all_gather_into_tensor(out, mine.clone())


SHARD_MIN_NUMEL = 1 << 20        # parameters from this size on (the hash tables) take the reduce-scatter route


def sharded_parameters(model, world, min_numel=None):
    """(name, parameter) of every tensor the reduce-scatter exchange shards: the dense hash tables (7.1 M x 2 + 1.9 M x 2
    floats at config B, 15.0 M x 4 + 6.6 M x 4 at waymo.gin) -- large, contiguous, and divisible by the world size (table
    rows are padded to multiples of 8 per level, gridencoder/grid.py:122-147, so 1 / 2 / 4 / 8 ranks always divide)."""
    min_numel = SHARD_MIN_NUMEL if min_numel is None else min_numel
    return [(n, p) for n, p in model.named_parameters()
            if p.requires_grad and p.numel() >= min_numel and p.numel() % world == 0 and p.is_contiguous()]


def wrap_ddp(model, device_ids=None, grad_exchange="all_reduce", shard_min_numel=None, **kw):
    """DistributedDataParallel for the training step, sized for this model: the table gradients are two dense tensors of
    57 + 15 MB (config B), so ONE 128 MB bucket (default 25 MB would cut the NeRF table's all-reduce off from the rest and
    serialise three collectives) whose gradient views alias the bucket (gradient_as_bucket_view: no 76 MB copy per step).
    What `accelerator.prepare` builds with torch defaults works too (tests/test_train_step.py), just slower; pass
    `accelerate.DistributedDataParallelKwargs(bucket_cap_mb=128, gradient_as_bucket_view=True)` to the Accelerator for the same.

    grad_exchange="reduce_scatter" (ref train.py:95,221 leaves the exchange to DDP's ring
    all-reduce): the hash tables are taken OUT of DDP's reducer.  Their gradients are produced by the last kernels of the
    backward, so DDP cannot overlap their all-reduce with anything, and after it every rank runs the same Adam pass over
    the whole table.  Instead train_utils.ShardedFusedAdam reduce-scatters each table's gradient (every rank receives the
    mean of ITS 1 / N of the rows: half the bytes of an all-reduce on the wire before the optimiser can start), steps only
    those rows (1 / N of the Adam traffic and of the moment memory), and all-gathers the updated rows.  The small dense
    parameters (< 4 MB in all) stay in DDP's bucket.  The tables are marked (`_ucn_sharded`) so that
    train_utils.sanitize_gradients leaves their still-local gradients alone: the reference's nan_to_num acts on the
    REDUCED gradient (train_utils.py:342-344 runs after DDP's reduction), which the sharded step reproduces."""
    from torch.nn.parallel import DistributedDataParallel
    if grad_exchange not in ("all_reduce", "reduce_scatter"):
        raise ValueError(f"grad_exchange must be 'all_reduce' or 'reduce_scatter', got {grad_exchange!r}")
    opts = dict(bucket_cap_mb=128, gradient_as_bucket_view=True, broadcast_buffers=False)
    opts.update(kw)
    if grad_exchange == "reduce_scatter":
        world = dist.get_world_size(opts.get("process_group"))
        sharded = sharded_parameters(model, world, shard_min_numel)
        for _, p in sharded:
            p._ucn_sharded = True
            # DDP broadcasts rank 0's module state at construction -- but not what it is told to ignore: keep its "every rank starts
            # from rank 0's weights" guarantee for the tables too (one broadcast per table, once)
            dist.broadcast(p.data, src=dist.get_global_rank(opts["process_group"], 0) if opts.get("process_group") is not None else 0,
                           group=opts.get("process_group"))
        DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(model, [n for n, _ in sharded])
    else:
        # a model wrapped before with grad_exchange="reduce_scatter": the marks and the ignore list must not outlive that wrap, or this
        # all-reduce wrap would leave the tables out of the reducer while create_optimizer still selects ShardedFusedAdam
        for p in model.parameters():
            if getattr(p, "_ucn_sharded", False):
                del p._ucn_sharded
        if getattr(model, "_ddp_params_and_buffers_to_ignore", None):
            DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(model, [])
    ddp = DistributedDataParallel(model, device_ids=device_ids, **opts)
    ddp.grad_exchange = grad_exchange
    return ddp


def all_gather_rows(local, num_rays, world, rank):
    """local: dict name -> [rows_local, width_k] float tensors (same keys / widths on every rank).
    Returns dict name -> [num_rays, width_k] on every rank."""
    if world == 1:
        return local
    if not _active(world):
        raise RuntimeError("render_image: num_processes > 1 but torch.distributed is not initialised")
    keys = sorted(local)
    widths = [local[k].shape[1] for k in keys]
    rp = rows_per_rank(num_rays, world)
    any_t = local[keys[0]]
    n_loc = any_t.shape[0]
    # one receive buffer per frame; this rank's rows are packed straight into its own slot of it
    out = torch.empty(world * rp, sum(widths), device=any_t.device, dtype=torch.float32)
    mine = out[rank * rp:(rank + 1) * rp]
    col = 0
    for k, w in zip(keys, widths):
        mine[:n_loc, col:col + w] = local[k]
        col += w
    if n_loc < rp:
        mine[n_loc:].zero_()                          # padding rows of the last shard(s), stripped below
    if EXCHANGE_EVENTS is not None:                   # bench.py: the collective's own duration, on the stream it runs on
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _exchange(out, mine, rank, rp)
        e1.record()
        EXCHANGE_EVENTS.append((e0, e1, mine.numel() * mine.element_size()))
    else:
        _exchange(out, mine, rank, rp)                # one collective per frame
    out = out[:num_rays]
    res, col = {}, 0
    for k, w in zip(keys, widths):
        res[k] = out[:, col:col + w]
        col += w
    return res


def all_gather_bundles(per_level, world, rank):
    """'ray_*' visualisation bundles (vis_num_rays rays per rank and level): concatenated over
    ranks like accelerator.gather does (models.py:968,972-974).  A few KB; one collective."""
    if world == 1:
        return per_level
    keys = sorted(per_level[0])
    flat = torch.cat([lvl[k].reshape(lvl[k].shape[0], -1) for lvl in per_level for k in keys], dim=1).contiguous()
    out = torch.empty(world * flat.shape[0], flat.shape[1], device=flat.device, dtype=flat.dtype)
    dist.all_gather_into_tensor(out, flat)
    res, col = [], 0
    for lvl in per_level:
        d = {}
        for k in keys:
            shp = lvl[k].shape[1:]
            w = int(torch.tensor(shp).prod()) if len(shp) else 1
            d[k] = out[:, col:col + w].reshape((out.shape[0],) + tuple(shp))
            col += w
        res.append(d)
    return res
