"""CPU tier, world_size 2 over gloo: the N>1 path (frame sharding + the one broadcast of the
packed streak database)."""
import os
import socket

import numpy as np
import pytest

import helpers as h


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tex_dir, norm, q):
    import importlib
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sharding = importlib.import_module('rain-rendering_amd.sharding')
    hb = importlib.import_module('rain-rendering_amd.hip_backend')
    bw = importlib.import_module('rain-rendering_amd.common.bad_weather')
    packed = None
    if rank == 0:                                   # only rank 0 touches the database files
        db = bw.DBManager(streaks_path=tex_dir, norm_coeff_path=norm)
        db.load_streak_database()
        packed = hb.pack_streak_db(db.streaks_light)
    texels, hs, ws, offs = sharding.broadcast_streak_db(packed, 0)
    frames = sharding.shard(list(range(11)), *sharding.rank_world())
    q.put((rank, texels.numpy().tobytes(), hs.tolist(), ws.tolist(), offs.tolist(), frames))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_shard_world2(tmp_path):
    import torch.multiprocessing as mp
    tex_dir, norm = h.synthetic.write_streak_db(str(tmp_path / 'db'))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tex_dir, norm, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, t0, h0, w0, o0, f0), (r1, t1, h1, w1, o1, f1) = res
    assert t0 == t1 and h0 == h1 and w0 == w1 and o0 == o1            # identical database on both ranks
    assert len(h0) == 50 and len(t0) == sum(a * b for a, b in zip(h0, w0))
    assert sorted(f0 + f1) == list(range(11)) and not set(f0) & set(f1)  # every frame exactly once
    assert f0 == [0, 2, 4, 6, 8, 10] and f1 == [1, 3, 5, 7, 9]


def test_shard_degenerate():
    sharding = h.pkg and __import__('importlib').import_module('rain-rendering_amd.sharding')
    assert sharding.shard([5, 6, 7], 0, 1) == [5, 6, 7]
    assert sharding.shard([5, 6, 7], 3, 8) == []


def _rename_worker(rank, world, port, out_dir, q):
    import importlib
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'], os.environ['WORLD_SIZE'] = str(rank), str(world)
    gen_mod = importlib.import_module('rain-rendering_amd.common.generator')
    g = object.__new__(gen_mod.Generator)           # no GPU needed for the folder logic
    g.conflict_strategy, g.rank, g.world = 'rename_folder', rank, world
    d = g._resolve_out_dir(out_dir)                 # env-only launch: the group is created on demand (gloo on CPU)
    q.put((rank, d))
    dist.barrier()
    dist.destroy_process_group()


def test_rename_folder_is_decided_once_world2(tmp_path):
    """--conflict_strategy rename_folder under two ranks: rank 0 picks the _copyNNNNN folder, both ranks write there
    (every rank resolving on its own scattered one run over several folders)."""
    import torch.multiprocessing as mp
    out_dir = str(tmp_path / 'out' / 'kitti' / 'seq' / 'rain' / '5mm')
    os.makedirs(out_dir)                            # an earlier run's folder
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rename_worker, args=(r, 2, port, out_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] == out_dir + '_copy00000'
    assert sorted(os.listdir(os.path.dirname(out_dir))) == ['5mm', '5mm_copy00000']


def _worklist_worker(rank, world, port, root, q):
    """The work list of a run is made by rank 0 alone.  Rank 1 arrives late and -- as a peer that has already started
    writing would -- finds an output file of frame 0 on disk: were it to make its own list with 'skip', it would drop that
    frame and the two shares would neither partition nor cover the run."""
    import importlib
    import time
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    gen_mod = importlib.import_module('rain-rendering_amd.common.generator')
    sharding = importlib.import_module('rain-rendering_amd.sharding')
    g = object.__new__(gen_mod.Generator)
    g.conflict_strategy, g.rank, g.world, g.dataset = 'skip', rank, world, 'kitti'
    g.noise_std, g.noise_scale = 2.0, 1.0
    files = [os.path.join(root, 'img', '%06d.png' % i) for i in range(7)]
    out_dir = os.path.join(root, 'out')
    dist.barrier()
    if rank == 1:
        time.sleep(1.0)
        os.makedirs(os.path.join(out_dir, 'rainy_image'), exist_ok=True)
        open(os.path.join(out_dir, 'rainy_image', '000000.png'), 'w').close()
    work, n_exist = sharding.rank0_decides(lambda: g._work_list(files, files, list(range(7)), out_dir, root, 3), rank, world)
    mine = sharding.shard(work, rank, world)
    q.put((rank, [it['i'] for it in work], [it['i'] for it in mine], [it['seeds'] for it in work], n_exist))
    dist.barrier()
    dist.destroy_process_group()


def test_work_list_is_decided_once_world2(tmp_path):
    import torch.multiprocessing as mp
    root = str(tmp_path)
    os.makedirs(os.path.join(root, 'img'))
    for i in range(7):
        open(os.path.join(root, 'img', '%06d.png' % i), 'w').close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worklist_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, m0, s0, e0), (_, w1, m1, s1, e1) = res
    assert w0 == w1 == list(range(7)) and e0 == e1 == 0             # one list, made before any rank wrote
    assert sorted(m0 + m1) == list(range(7)) and not set(m0) & set(m1)
    assert s0 == s1 and s0[3] == (0, 3) and s0[6] == (0, 3, 6)      # noise-seed history: frames 0, 3, 6 share simulated frame 0


def _failing_worker(rank, world, port, q):
    import importlib
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sharding = importlib.import_module('rain-rendering_amd.sharding')

    def decide():
        raise FileNotFoundError("no such particles file")        # only ever evaluated on rank 0
    try:
        sharding.rank0_decides(decide, rank, world)
        q.put((rank, 'returned'))
    except Exception as e:                                          # noqa: BLE001
        q.put((rank, type(e).__name__ + ': ' + str(e)))
    assert sharding.rank0_decides(lambda: 41 + 1, rank, world) == 42   # the group is still usable afterwards
    dist.barrier()
    dist.destroy_process_group()


def test_rank0_failure_reaches_every_rank_world2():
    """A decision that raises on rank 0 raises on every rank (the others used to wait in the broadcast for its time-out)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == 'FileNotFoundError: no such particles file'
    assert res[1] == 'RuntimeError: rank 0 failed: FileNotFoundError: no such particles file'
