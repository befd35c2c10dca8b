"""Frame sharding across the GPUs of one node and the single collective of the path.

Frames of a sequence are independent units (per-frame seed reference generator.py:318,
per-frame accumulators generator.py:389-394), so rank r renders frames idx[r::world]
(round-robin balances sequences whose rain changes along the sequence).  The only shared
state is the streak database: rank 0 reads and packs it, ONE broadcast (RCCL over xGMI
when the process group is `nccl`, gloo in the CPU test tier) puts it on every rank.
No further collective is issued on the data path."""
import os

import numpy as np


def rank_world():
    """(rank, world) from torch.distributed if initialised, else from the launcher's env."""
    import sys
    dist = sys.modules.get('torch.distributed')        # (a group can only exist if torch is loaded: a single-GPU run never imports it)
    try:
        if dist is not None and dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def ensure_group(rank, world):
    """A launcher that only sets RANK / WORLD_SIZE / MASTER_* (no process group yet): create one, RCCL when a GPU is
    there, gloo otherwise, so that the streak-database broadcast and rank0_decides work."""
    if world <= 1:
        return
    import torch
    import torch.distributed as dist
    if dist.is_available() and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(os.environ.get('RAIN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo'), rank=rank, world_size=world)


def rank0_decides(fn, rank, world):
    """fn() evaluated on rank 0 only; every rank returns rank 0's (picklable) result."""
    if world <= 1:
        return fn()
    import torch.distributed as dist
    ensure_group(rank, world)
    # An exception on rank 0 (a missing file, an unsupported --conflict_strategy, a failing simulator) travels with the
    # broadcast and is raised on EVERY rank: the others would otherwise sit in the collective until its time-out.
    box = [None]
    if rank == 0:
        try:
            box[0] = (True, fn())
        except Exception as e:                                     # noqa: BLE001 -- re-raised below, on all ranks
            box[0] = (False, (type(e).__name__, str(e)))
            err = e
    dist.broadcast_object_list(box, src=0)
    ok, val = box[0]
    if ok:
        return val
    if rank == 0:
        raise err
    raise RuntimeError("rank 0 failed: %s: %s" % val)


def shard(indices, rank, world):
    """Round-robin frame assignment."""
    return list(indices)[rank::world]


def broadcast_streak_db(packed, src=0, device=None, force=False):
    """packed = (texels u8[], tex_h i32[], tex_w i32[], tex_off i64[]) on `src`, None elsewhere.
    Returns (texels torch.uint8 tensor on `device`, tex_h, tex_w, tex_off numpy).  With an
    uninitialised process group this is a plain upload."""
    import torch
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
    dev = device if device is not None else torch.device('cpu')
    if not live:
        texels, hs, ws, offs = packed
        return torch.from_numpy(np.ascontiguousarray(texels)).to(dev), hs, ws, offs
    rank = dist.get_rank()
    head = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == src:
        texels, hs, ws, offs = packed
        head[0], head[1] = len(hs), texels.size
    dist.broadcast(head, src=src)
    n_tex, n_bytes = int(head[0].item()), int(head[1].item())
    # one payload: [tex_h | tex_w | tex_off | texels] as bytes
    meta_bytes = n_tex * (4 + 4 + 8)
    buf = torch.empty(meta_bytes + n_bytes, dtype=torch.uint8, device=dev)
    if rank == src:
        raw = np.concatenate([hs.astype(np.int32).view(np.uint8), ws.astype(np.int32).view(np.uint8),
                              offs.astype(np.int64).view(np.uint8), np.ascontiguousarray(texels, np.uint8)])
        buf.copy_(torch.from_numpy(raw))
    dist.broadcast(buf, src=src)
    host = buf[:meta_bytes].cpu().numpy()
    hs = host[:4 * n_tex].view(np.int32).copy()
    ws = host[4 * n_tex:8 * n_tex].view(np.int32).copy()
    offs = host[8 * n_tex:].view(np.int64).copy()
    return buf[meta_bytes:], hs, ws, offs


def load_and_broadcast_streak_db(db, hip, rank, world, force_collective=False):
    """DBManager.load_streak_database on rank 0 only, then the broadcast; every rank ends up
    with the database resident on its GPU and with db.ratio / db.streaks_light populated.
    force_collective: take the collective route in a group of ONE rank too (the GPU test tier walks
    init_process_group('nccl') -> device-tensor broadcast -> rr_set_streak_db_device on the one GPU it has)."""
    from . import hip_backend
    if world <= 1 and not force_collective:
        db.load_streak_database()
        hip.set_streak_db(db.streaks_light)
        return
    import torch
    if world > 1:
        ensure_group(rank, world)
    packed = None
    if rank == 0:
        db.load_streak_database()
        packed = hip_backend.pack_streak_db(db.streaks_light)
    dev = torch.device('cuda', hip.device) if torch.cuda.is_available() else torch.device('cpu')
    texels, hs, ws, offs = broadcast_streak_db(packed, 0, dev, force=force_collective)
    if rank != 0:
        host = texels.cpu().numpy()
        db.streaks_light = [host[o:o + h * w].reshape(h, w).copy() for h, w, o in zip(hs, ws, offs)]
        db.ratio = np.unique(np.array([w / h for h, w in zip(hs, ws)]))
    if texels.is_cuda:
        hip.set_streak_db_device(texels.data_ptr(), texels.numel(), hs, ws, offs)
    else:
        hip.set_streak_db(db.streaks_light)
