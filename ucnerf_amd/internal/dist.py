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


def rows_per_rank(num_rays, world):
    return (num_rays + world - 1) // world


def shard_bounds(num_rays, world, rank):
    rp = rows_per_rank(num_rays, world)
    lo = min(rank * rp, num_rays)
    return lo, min(lo + rp, num_rays)


def _active(world):
    return world > 1 and dist.is_available() and dist.is_initialized()


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
    # one receive buffer per frame; this rank's rows are packed straight into its own slot of it (RCCL's in-place
    # all-gather: send buffer = receive buffer + rank * count), so nothing frame-sized is allocated or zeroed twice
    out = torch.empty(world * rp, sum(widths), device=any_t.device, dtype=torch.float32)
    mine = out[rank * rp:(rank + 1) * rp]
    col = 0
    for k, w in zip(keys, widths):
        mine[:n_loc, col:col + w] = local[k]
        col += w
    if n_loc < rp:
        mine[n_loc:].zero_()                          # padding rows of the last shard(s), stripped below
    send = mine if dist.get_backend() == "nccl" else mine.clone()   # gloo (CPU tests) wants a separate send buffer
    dist.all_gather_into_tensor(out, send)            # one collective per frame
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
