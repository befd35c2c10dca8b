"""Worker of tests/test_gpu_driver.py::test_rccl_broadcast_of_the_streak_database_world1 (its own process: torch first)."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as h  # noqa: E402


def main(tmp):
    assert torch.cuda.is_available(), "no GPU visible to torch"
    sharding = importlib.import_module('rain-rendering_amd.sharding')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == 'nccl'
    sc = h.Scene(tmp, 96, 160, 150, seed0=61)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
    plain = h.hb.RainHip(0)
    try:
        plain.set_streak_db(sc.db.streaks_light)
        plain.set_camera(sc.cam)
        want = plain.render_frames([fr])[0]
    finally:
        plain.close()
    rh = h.hb.RainHip(0)
    try:
        db = h.bw.DBManager(streaks_path=sc.tex_dir, norm_coeff_path=sc.norm)
        sharding.load_and_broadcast_streak_db(db, rh, 0, 1, force_collective=True)      # header + payload: device-tensor broadcasts
        box = ['work list']                                                            # what rank0_decides sends under N > 1
        dist.broadcast_object_list(box, src=0)
        assert box == ['work list']
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                                                             # (bench.py's MAX-over-ranks timing uses one)
        rh.set_camera(sc.cam)
        got = rh.render_frames([fr])[0]
        torch.cuda.synchronize()
    finally:
        rh.close()
    dist.barrier()
    dist.destroy_process_group()
    for k in ('mask', 'mask_i32', 'image_u8', 'status', 'rainy_bg'):
        assert np.array_equal(got[k], want[k]), k
    assert len(db.streaks_light) == 50 and (want['status'] == 0).sum() > 100
    print('NCCL-WORLD1-OK')


if __name__ == '__main__':
    main(sys.argv[1])
