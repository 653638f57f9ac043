"""Scene sharding and result gathering for one-process-per-GPU runs.

Scenes (and the trajectories that belong to them) are independent: rank r of W owns scenes
r, r+W, r+2W, ... and never exchanges data while optimising.  The only collective is the
gather of the per-trajectory results at the end: `[B_local, n+5]` fp32 rows
(final x, total cost, 4 cost terms) -> every rank, RCCL over xGMI on GPUs, gloo in CPU tests."""
import torch
import torch.distributed as dist


def owned_scenes(n_scenes, rank, world):
    """round-robin ownership"""
    return list(range(rank, n_scenes, world))


def owner_of(scene, world):
    return scene % world


def pack_results(x, costs, weights, ctx=None, out=None):
    """x [B,n] f64, costs [B,4] f64, weights [4] -> [B, n+5] f32 rows (x, total, 4 terms).
    With a library context and device tensors: one kernel on the context's stream (neo_pack_results_dev; `weights` then a
    host sequence of 4 floats); otherwise three torch copies (CPU tests, gloo)."""
    B, n = x.shape
    if out is None:
        out = torch.empty(B, n + 5, dtype=torch.float32, device=x.device)
    if ctx is not None and x.is_cuda:
        import ctypes
        w4 = (ctypes.c_double * 4)(*[float(v) for v in weights])
        pp = lambda t: ctypes.c_void_p(t.data_ptr())
        ctx.check(ctx.lib.neo_pack_results_dev(ctx.h, B, n, pp(x), pp(costs), ctypes.cast(w4, ctypes.c_void_p), pp(out)))
        return out
    out[:, :n] = x
    out[:, n] = (costs * weights).sum(dim=1)
    out[:, n + 1:] = costs
    return out


def gather_results(local, world, out=None, force=False, async_op=False):
    """all ranks end up with the rows of every rank, rank-major.  Equal B_local on every rank.
    `force`: run the collective even in a one-rank group (exercises the process-group path).
    `async_op`: do not make the caller's stream wait for the collective; returns (out, work) and the caller
    calls work.wait() (or synchronises the device) before it reads `out` or frees `local`."""
    if world == 1 and not force:
        return (local, None) if async_op else local
    if out is None:
        out = torch.empty(world * local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend() == "gloo":
        # gloo has no device collectives for this call: stage through the host (tests that run several ranks on one GPU;
        # RCCL refuses two ranks on one device).  Synchronous by construction.
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu())
        out.copy_(host)
        return (out, _Done()) if async_op else out
    work = dist.all_gather_into_tensor(out, local.contiguous(), async_op=async_op)
    return (out, work) if async_op else out


class _Done:
    """a finished piece of work (the synchronous host-staged gather)"""

    def wait(self):
        return True


def scene_major_order(gathered, n_scenes, world, rows_per_scene):
    """reorder rank-major gathered rows (each rank holds its owned scenes in increasing order) into
    scene order; needs n_scenes divisible by world."""
    per_rank = n_scenes // world
    g = gathered.view(world, per_rank, rows_per_scene, gathered.shape[1])
    return g.permute(1, 0, 2, 3).reshape(n_scenes * rows_per_scene, gathered.shape[1])
