"""Drop-in for the reference's common/generator.py: `Generator(args).run()` with the same
argument object (fields consumed at reference generator.py:25-67), the same input trees and
the same output tree.  The per-drop loop of the reference (generator.py:431-452 ->
compute_drop -> add_drop_to_image) is replaced by ONE call per batch of frames into the HIP
library (rr_render_frames); everything around it (file walking, conflict strategy, seeding,
streak filter, pre-pass, saving) follows the reference.

Multi-GPU: started under torch.distributed.run (or with RANK/WORLD_SIZE/MASTER_* set) every rank
renders frames idx[rank::world] of each sequence on its own GPU; the streak database is
packed on rank 0 and broadcast once (sharding.broadcast_streak_db); no other collective on the
data path (rank 0 also decides the output folder name, one small object broadcast per sequence).
"""
import os
import sys
import time

import numpy as np

from . import add_attenuation, my_utils, solid_angle, imgops, envmap
from .bad_weather import DBManager, RainRenderer
from .. import hip_backend, sharding

FOG_ATT = 1                  # reference generator.py:19
USE_DEPTH_WEIGHTING = 0      # reference generator.py:20 (dead in the reference, not implemented)


def _cpu_budget():
    """CPUs this process may really use: the cgroup quota when there is one (containers), else the affinity mask."""
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 4


class Generator:
    def __init__(self, args):
        self.conflict_strategy = args.conflict_strategy
        self.rendering_strategy = args.rendering_strategy
        if args.rendering_strategy is None:
            self.output_root = os.path.join(args.output, args.dataset)
        else:
            self.output_root = os.path.join(args.output, args.dataset + '_' + args.rendering_strategy)
        self.dataset = args.dataset
        self.dataset_root = args.dataset_root
        self.images = args.images
        self.sequences = args.sequences
        self.depth = args.depth
        self.particles = args.particles
        self.weather = args.weather
        self.texture = args.texture
        self.norm_coeff = args.norm_coeff
        self.save_envmap = args.save_envmap
        self.settings = args.settings
        self.calib = args.calib
        self.exposure = args.settings["cam_exposure"]
        self.camera_gain = args.settings["cam_gain"]
        self.focal = args.settings["cam_focal"] / 1000.
        self.f_number = args.settings["cam_f_number"]
        self.focus_plane = args.settings["cam_focus_plane"]
        self.noise_scale = args.noise_scale
        self.noise_std = args.noise_std
        self.opacity_attenuation = args.opacity_attenuation
        self.frame_start = args.frame_start
        self.frame_end = args.frame_end
        self.frame_step = args.frame_step
        self.frames = args.frames
        self.verbose = args.verbose
        self.env_type = 'ours'
        self.irrad_type = 'ambient'
        self.db = None
        self.renderer = None
        self.device_particles = bool(getattr(args, 'device_particles', False))     # drop tables generated on the GPU (no XML)
        self.sim_options = getattr(args, 'sim_options', {})
        self.batch = int(os.environ.get('RAIN_BATCH', '128'))     # frames per library call (three calls in flight); bench.py's host-inclusive leg uses the same
        self.rank, self.world = sharding.rank_world()
        self.device = int(getattr(args, 'device', os.environ.get('RAIN_DEVICE', os.environ.get('LOCAL_RANK', '0'))))   # (RAIN_DEVICE: main.py)
        self._hip = None
        self._pool = None
        self.stats = []
        self.timing = []
        if self.rendering_strategy not in (None, 'white'):
            raise NotImplementedError("rendering_strategy %r: 'naive_db' reads a non-existent attribute in the reference "
                                      "(bad_weather.py:355) and cannot run there either" % self.rendering_strategy)
        self.check_folders()

    def check_folders(self):
        """reference generator.py:85-104."""
        print('Output directory: {}'.format(self.output_root))
        existing = []
        for sequence in self.sequences:
            for w in self.weather:
                out_dir = os.path.join(self.output_root, sequence, w["weather"], '{}mm'.format(w["fallrate"]))
                if os.path.exists(out_dir):
                    existing.append(out_dir)
        if len(existing) != 0 and self.conflict_strategy is None:
            print("\r\nFolders already exist: \n%s" % "\n".join(existing))
            while self.conflict_strategy not in ["overwrite", "skip", "rename_folder"]:
                self.conflict_strategy = input("\r\nWhat strategy to use (overwrite|skip|rename_folder):   ")
        assert self.conflict_strategy in [None, "overwrite", "skip", "rename_folder"]

    # ------------------------------------------------------------------------------------------
    def _hip_ctx(self):
        if self._hip is None:
            self._hip = hip_backend.RainHip(self.device)
        return self._hip

    def _io_pool(self):
        """Threads for PNG decode / deflate (PIL's decoder and zlib release the GIL).  SURVEY 8f next #3: the host I/O
        around the GPU call is what bounds the driver end to end, so it runs ahead of / behind the GPU."""
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            ranks_here = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))
            self._pool = ThreadPoolExecutor(max_workers=int(os.environ.get('RAIN_IO_THREADS', 0)) or max(2, min(96, 2 * _cpu_budget() // ranks_here)))
        return self._pool

    def _io_threads(self):
        """Worker threads of the library's batch I/O calls (rr_io_read_frames / rr_io_write_frames / rr_host_pack_frames):
        RAIN_IO_THREADS, else the process' CPU budget."""
        ranks_here = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))        # torch.distributed.run: ranks sharing this host's CPUs
        return int(os.environ.get('RAIN_IO_THREADS', 0)) or max(2, min(96, _cpu_budget() // ranks_here))

    def _stage_pool(self):
        """Two Python threads that each carry ONE whole-batch job at a time (decode-ahead, encode-behind) into the
        library, where the per-frame work runs on native threads without the interpreter lock."""
        if getattr(self, '_stages', None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._stages = ThreadPoolExecutor(max_workers=2)
        return self._stages

    def _pack(self, pristine, imW, imH, seed, earlier_seeds=()):
        """Frame filter + drop table with the frame's random draws (generator.py:318,413-425) on a private copy of the
        simulator frame's table.  With angular noise the reference rotates the streak end points IN the shared table
        (generator.py:152-161), so a frame that re-uses a simulator frame sees the rotations of the earlier frames of
        the run that used it: `earlier_seeds` replays those (same draws, same order) first.  The result therefore
        equals the reference's sequential run, does not depend on how frames are sharded over GPUs, and can be
        computed on any thread."""
        noisy = bool(self.noise_std) and bool(self.noise_scale)
        table = pristine.take(slice(None)) if noisy else pristine              # nothing is mutated without noise
        for s_ in earlier_seeds:
            hip_backend.pack_frame(table, self.db, imW, imH, s_, self.noise_std, self.noise_scale)
        drops = hip_backend.pack_frame(table, self.db, imW, imH, seed, self.noise_std, self.noise_scale)   # generator.py:413-420
        assert len(drops) <= 2 ** 16, "Assert that the number of drops doesn't overpass the uint16 rain_mask capacity"
        return drops

    def _load_frame(self, image_file, depth_file, rs):
        """Image and depth of one frame as Generator.run reads them (generator.py:352-384)."""
        # generator.py:352 `cv2.imread(f) / 255.0`: at render_scale 1 the bytes themselves go to the GPU
        # (rr_prepass_in.bg_u8) and the division happens there; a resized image has to be float64
        bg = imgops.imread_bgr(image_file)
        if rs != 1:
            bg = bg / 255.0
            bg = imgops.resize_linear(bg, int(bg.shape[1] // rs), int(bg.shape[0] // rs))
        if depth_file.endswith(".png"):
            depth = imgops.imread_unchanged(depth_file)
            if depth is None:
                return None
            depth = depth.astype(np.float32) / 256.
        elif depth_file.endswith(".npy"):
            depth = np.load(depth_file)
        else:
            raise Exception("Invalid extension")
        ds = self.settings["depth_scale"]
        depthHW = np.array([int((depth.shape[0] * ds) // rs), int((depth.shape[1] * ds) // rs)])
        if not np.all(depth.shape[:2] == depthHW):
            # cv2.resize keeps the depth's own type (float32 for PNG depth): so does the fog pre-pass that follows
            dt = depth.dtype if depth.dtype in (np.float32, np.float64) else np.float64
            depth = imgops.resize_linear(depth, int(depthHW[1]), int(depthHW[0])).astype(dt)
        assert np.all(np.array(depth.shape[:2]) <= np.array(bg.shape[:2])), "Depth cannot be larger than the image"
        if not np.all(np.array(depth.shape[:2]) == np.array(bg.shape[:2])):
            bg = my_utils.crop_center(bg, depth.shape[0], depth.shape[1])
        return np.ascontiguousarray(bg), depth

    # ---- the asynchronous frame pipeline ------------------------------------------------------------------
    class _Slot:
        """Pinned host buffers of one batch in flight (inputs the I/O threads decode into, outputs they deflate from).
        Every array is one page-locked block with the frames back to back, each on a 16-byte boundary (RainHip.host_rows):
        the library then moves an array of the whole batch with ONE copy instead of one DMA request per frame."""

        def __init__(self, hip, B, H, W, We, bg_dtype, depth_dtype, save_envmap, drops_cap, png_rows=False):
            """png_rows: the image / depth blocks hold the files' filtered scanlines (rr_io_read_frames_rows: H rows of 1 + 3 W /
            1 + 2 W bytes per frame) instead of pixels."""
            self.B, self.H, self.W = B, H, W
            self._raw = []

            def rows(shape, dtype):
                raw, views = hip.host_rows(B, shape, dtype)
                self._raw.append(raw)
                return views
            drops_cap = (drops_cap + 3) // 4 * 4
            self.bg = rows((H * (1 + 3 * W),), np.uint8) if png_rows else rows((H, W, 3), bg_dtype)
            self.depth = rows((H * (1 + 2 * W),), np.uint8) if png_rows else rows((H, W), depth_dtype)
            self.drops = rows((drops_cap,), hip_backend.DROP_DTYPE)
            self.status = rows((drops_cap,), np.int32)
            row = H * (1 + 4 * W)
            self.png_i = rows((row,), np.uint8)
            self.png_m = rows((row,), np.uint8)
            # the blocks themselves ((B, stride) bytes): what the library's batch I/O calls address
            self.raw_bg, self.raw_depth, self.raw_drops, _, self.raw_png_i, self.raw_png_m = self._raw[:6]
            self.env = rows((H, We, 3), np.uint8) if save_envmap else None
            self.drops_cap = drops_cap
            self.key = (B, H, W, We, 'png rows' if png_rows else np.dtype(bg_dtype), 'png rows' if png_rows else np.dtype(depth_dtype), bool(save_envmap))
            self.items, self.encodes, self.busy = [], [], False
            self.prep, self.pkey, self.n_valid = None, None, 0

        def free(self, hip):
            """Page-locked memory is only returned by rr_host_free (dropping the numpy views frees nothing): called when
            a slot is replaced by a larger / differently shaped one, after its batch has been collected and encoded."""
            assert not self.busy and not self.encodes
            self.bg = self.depth = self.drops = self.status = self.png_i = self.png_m = self.env = None
            self.raw_bg = self.raw_depth = self.raw_drops = self.raw_png_i = self.raw_png_m = None
            self.prep = None
            for raw in self._raw:
                hip.host_free(raw)
            self._raw = []

    def _decode_into(self, slot, k, item, rs, pristine, imW, imH, seeds):
        """I/O-thread part of one frame: decode image + depth into the slot's pinned buffers, build the drop table."""
        loaded = self._load_frame(item['image_file'], item['depth_file'], rs)
        if loaded is None:
            return None
        bg, depth = loaded
        drops = self._pack(pristine, imW, imH, seeds[-1], seeds[:-1])
        return bg, depth, drops

    def _encode(self, slot, k, item, n_drops):
        """I/O-thread part after the GPU: deflate the two PNG files from the scanlines the library delivered."""
        H, W = slot.H, slot.W
        os.makedirs(os.path.dirname(item['out_rainy_path']), exist_ok=True)
        os.makedirs(os.path.dirname(item['out_rainy_mask_path']), exist_ok=True)
        imgops.png_from_scanlines(item['out_rainy_path'], slot.png_i[k], W, H)        # generator.py:466
        imgops.png_from_scanlines(item['out_rainy_mask_path'], slot.png_m[k], W, H)   # generator.py:467
        if self.save_envmap:
            os.makedirs(os.path.dirname(item['out_env_path']), exist_ok=True)
            env_bgr = slot.env[k] / 255.0                                              # generator.py:469 (plt.imsave of a float map)
            imgops.imsave_rgb(item['out_env_path'], (np.clip(env_bgr[..., ::-1], 0, 1) * 255).astype(np.uint8))
        return int(np.count_nonzero(slot.status[k][:n_drops]))

    def compute_drop(self, bg, drop_dict, rainy_bg, rainy_mask, rainy_saturation_mask):
        """Single-drop compatibility seam with the reference's signature (generator.py:119-191):
        consumes the same RNG draws, composites ONE streak into rainy_bg / rainy_mask (in place and
        returned) by calling the library with a one-record drop table.  self.env_map_xyY,
        self.solid_angle_map, self.db and the camera must be set like Generator.run does.  Returns the
        reference's six values (rainy_bg, rainy_mask, rainy_saturation_mask, drop, blended_drop, minC):
        `drop` the coloured, defocused tile as placed (h x w x 4), `blended_drop` the image region under
        it after the blend, `minC` its clamped position -- or, where the reference prints "Erroneous
        drop", blended_drop None, drop None and the un-clamped minC.  One launch chain per drop -- use
        run() / rr_render_frames for throughput."""
        H, W = bg.shape[:2]
        t = drop_dict._table if hasattr(drop_dict, '_table') else None
        if t is None:
            from .bad_weather import StreakTable
            t = StreakTable(1)
            t.pid[0] = drop_dict.pid
            t.wps[0], t.wpe[0] = drop_dict.world_position_start, drop_dict.world_position_end
            t.ips[0], t.ipe[0] = drop_dict.image_position_start, drop_dict.image_position_end
            t.iw1[0], t.iw2[0] = drop_dict.image_diameter_start, drop_dict.image_diameter_end
            t.ratio[0], t.max_width[0], t.length[0] = drop_dict.ratio, drop_dict.max_width, drop_dict.length
            t.type[0] = drop_dict.drop_type.value
        drops = hip_backend.pack_drops(t, np.array([0]), self.db, self.noise_std, self.noise_scale)
        if hasattr(drop_dict, 'image_position_start'):        # the in-place endpoint rotation (generator.py:152-161)
            drop_dict.image_position_start[:] = t.ips[0]
            drop_dict.image_position_end[:] = t.ipe[0]
        white = self.rendering_strategy == 'white'
        out = self._hip_ctx().render_frames([dict(bg=bg, rainy_bg=rainy_bg, env_xyY=self.env_map_xyY,
                                                  omega=self.solid_angle_map, drops=drops,
                                                  opacity_attenuation=self.opacity_attenuation,
                                                  strategy=1 if white else 0)], want_colour=True)[0]
        # the raw tile's frame (generator.py:126-132 Big, :166-171 otherwise)
        x0, y0, x1, y1 = (int(drops[k][0]) for k in ('x0', 'y0', 'x1', 'y1'))
        if int(drops['type'][0]) == 0:
            _, _, maxC, minC = RainRenderer.warping_points(drop_dict, self.db.streaks_light[int(drops['tex_index'][0])], W, H)
            shape = np.subtract(maxC, minC).astype(int)
            tw, th = max(shape[0], 1), max(shape[1], 1)
        else:
            tw, th = max(abs(x1 - x0), int(drops['max_width'][0]) + 2), max(abs(y1 - y0), 2)
            minC = np.array([x0, y0])
        if out['status'][0] != 0:
            print('Erroneous drop (status %d)' % out['status'][0])
            return rainy_bg, rainy_mask, rainy_saturation_mask, None, None, minC
        rainy_bg[...] = out['rainy_bg']
        rainy_mask += out['mask']
        renderer = self.renderer or RainRenderer(self.focal, self.f_number, 6, 10, 165)
        min_c, ph, pw = renderer.placed_tile(minC, tw, th, None if white else drop_dict.world_position_start[2], W, H)
        ys, xs = int(min_c[1]), int(min_c[0])
        blended = rainy_bg[ys:ys + ph, xs:xs + pw, :].copy()
        a_vis = out['mask'][ys:ys + ph, xs:xs + pw]
        K = out['colour'][0]
        drop = np.dstack([a_vis * K[0], a_vis * K[1], a_vis * K[2], a_vis])
        return rainy_bg, rainy_mask, rainy_saturation_mask, drop, blended, min_c

    def _resolve_out_dir(self, out_dir):
        """generator.py:213-226.  Under several ranks the choice (incl. the _copyNNNNN shift of 'rename_folder') is made
        by rank 0 alone and broadcast: every rank of a run writes into the same folder."""
        def resolve():
            d = out_dir
            if os.path.exists(d):
                if self.conflict_strategy in ("skip", "overwrite"):
                    pass
                elif self.conflict_strategy == "rename_folder":
                    shift = 0
                    while os.path.exists(d + '_copy%05d' % shift):
                        shift += 1
                    d = d + '_copy%05d' % shift
                else:
                    raise NotImplementedError
            os.makedirs(d, exist_ok=True)
            return d
        return sharding.rank0_decides(resolve, self.rank, self.world)

    def _work_list(self, files, depth_files, idx, out_dir, out_seq_dir, n_sim):
        """(work items, #frames skipped because they exist) of one (sequence, weather) run: the file checks and the
        conflict strategy of generator.py:324-350, the frame's simulation index (:304-312) and its noise-seed history."""
        frames_exist_nb = 0
        work = []
        for i in idx:
            image_file, depth_file = files[i], depth_files[i]
            assert os.path.exists(image_file), "Image file {} does not exist".format(image_file)
            assert os.path.exists(depth_file), "Depth file {} does not exist".format(depth_file)
            file_name = os.path.split(image_file)[-1]
            item = dict(i=i, image_file=image_file, depth_file=depth_file,
                        out_rainy_path=os.path.join(out_dir, 'rainy_image', '{}.png'.format(file_name[:-4])),
                        out_rainy_mask_path=os.path.join(out_dir, 'rain_mask', '{}.png'.format(file_name[:-4])),
                        out_env_path=os.path.join(out_seq_dir, 'envmap', '{}.png'.format(file_name[:-4])))
            if os.path.exists(item['out_rainy_path']) or os.path.exists(item['out_rainy_mask_path']):
                if self.conflict_strategy == "skip":
                    frames_exist_nb += 1
                    continue
                elif self.conflict_strategy == "overwrite":
                    pass
                else:
                    raise NotImplementedError
            item['f_name_idx'] = self._frame_name_index(i, len(files), n_sim)  # generator.py:304-312
            work.append(item)
        # with angular noise a frame inherits the end-point rotations of the run's earlier frames that used the
        # same simulated frame (generator.py:152-161): remember their seeds, whoever renders them
        noisy = bool(self.noise_scale) and bool(self.noise_std)
        seen = {}
        for item in work:
            k = item['f_name_idx'] % n_sim
            item['seeds'] = tuple(seen.get(k, ())) + (item['f_name_idx'],) if noisy else (item['f_name_idx'],)
            if noisy:
                seen.setdefault(k, []).append(item['f_name_idx'])
        return work, frames_exist_nb

    def _frame_name_index(self, i, n_files, n_sim):
        """generator.py:304-312: the index that seeds the frame and picks its simulated frame.  nuScenes spreads the
        simulated frames over the sequence's files."""
        if self.dataset == 'nuscenes':
            return int(np.linspace(0, n_sim, n_files, endpoint=False, dtype=int)[i])
        return i

    def _mark(self, name):
        """set-up clock: seconds since run() began at which a stage of the set-up was finished (timing[...]['setup'])"""
        self._marks.append((name, round(time.time() - self._t_run0, 4)))

    def run(self):
        self._t_run0, self._marks = time.time(), []
        folders_num = len(self.images)
        B = max(1, self.batch)
        for folder_idx, sequence in enumerate(self.sequences):
            self._t_run0, self._marks = time.time(), []          # the set-up clock of THIS sequence (timing[...]['setup'])
            print('\nSequence: ' + sequence)
            depth_folder = self.depth[sequence]
            for sim_idx, sim_weather in enumerate(self.weather):
                weather, fallrate = sim_weather["weather"], sim_weather["fallrate"]
                out_seq_dir = os.path.join(self.output_root, sequence)
                out_dir = self._resolve_out_dir(os.path.join(out_seq_dir, weather, '{}mm'.format(fallrate)))
                sim_file = self.particles[sequence][sim_idx]
                fog_params = {"rain_intensity": fallrate, "focal": self.focal, "f_number": self.f_number, "angle": 90,
                              "exposure": self.exposure, "camera_gain": self.camera_gain}
                files = [os.path.join(self.images[sequence], p) for p in my_utils.os_listdir(self.images[sequence])
                         if os.path.isfile(os.path.join(self.images[sequence], p))]
                depth_files = [os.path.join(depth_folder, d) for d in my_utils.os_listdir(depth_folder)]
                im = files[0]
                if im.endswith(".png"):
                    imH, imW = imgops.imread_bgr(im).shape[0:2]
                elif im.endswith(".npy"):
                    imH, imW = np.load(im).shape[0:2]
                else:
                    raise Exception("Invalid extension", im)
                rs = self.settings["render_scale"]
                imH, imW = imH // rs, imW // rs

                print('Simulation: rain {}mm/hr'.format(fallrate))
                self.db = DBManager(streaks_path_xml=sim_file, streaks_path=self.texture, norm_coeff_path=self.norm_coeff)
                self.renderer = RainRenderer(focal=self.focal, f_number=self.f_number, focus_plane=6, radius=10, fov=165)
                map_generator = envmap.EnvironmentMapGenerator(self.focal, imW, imH)
                FOG = add_attenuation.FogRain(**fog_params)
                # streak DB: loaded by rank 0, one broadcast, then resident on every GPU
                self._mark('file lists, first image size')
                hip = self._hip_ctx()
                self._mark('library context (HIP runtime, device)')
                sharding.load_and_broadcast_streak_db(self.db, hip, self.rank, self.world)
                self._mark('streak database loaded + resident')
                hip.set_camera(hip_backend.make_camera(self.focal, self.f_number, self.exposure))
                hip.set_prepass_kernels(imgops.gaussian_kernel(25, 25), imgops.gaussian_kernel(15, 0))
                hip.set_colormap(imgops.viridis_lut())
                fog_const = FOG.constants()
                sims = None
                if self.device_particles:
                    # no particle file: the settings of the run's simulated frames (tools/particles.sim_frames: what
                    # simulate() would have written to XML, as rr_sim_frame records) + the diameter tables they refer to
                    if bool(self.noise_std) and bool(self.noise_scale):
                        raise NotImplementedError("--device_particles: angular noise is not offered on the device-generated path")
                    from ..tools import particles
                    opts = self.sim_options[sequence]
                    n_sim = particles.n_sim_frames(opts)
                    sims, dgrid, cdf = particles.sim_frames(opts, fallrate, n_sim, render_scale=rs, seed=0)
                    hip.set_particle_tables(dgrid, cdf)
                    frame_render_dict = []
                else:
                    self.db.load_streaks_from_xml(self.dataset, self.settings, [imW, imH], use_pickle=False, verbose=self.verbose)
                    frame_render_dict = list(self.db.streaks_simulator.values())
                    n_sim = len(frame_render_dict)
                self._mark('particle file parsed' if not self.device_particles else 'particle generator tables')

                f_end = len(files) if self.frame_end is None else min(self.frame_end, len(files))
                if self.frames:
                    idx = np.unique(np.clip(self.frames, 0, f_end - 1)).tolist()
                else:
                    idx = list(range(self.frame_start, f_end, self.frame_step))
                print("{} images".format(len(idx)))
                # The work list of the WHOLE run -- which frames exist already (skip / overwrite), which simulated frame and
                # which seeds a frame gets -- is made ONCE, by rank 0, and broadcast: every rank making its own would race with
                # the ranks that are already writing into out_dir (a slower rank would skip, or refuse, frames its peers have
                # just rendered, and the shares would neither partition nor cover the run).  Then this rank's share.
                work, frames_exist_nb = sharding.rank0_decides(
                    lambda: self._work_list(files, depth_files, idx, out_dir, out_seq_dir, n_sim), self.rank, self.world)
                work = sharding.shard(work, self.rank, self.world)
                self._mark('work list')
                sim_t0 = time.time()
                self._run_batches(hip, work, B, rs, imW, imH, frame_render_dict, fog_const, map_generator, folder_idx, folders_num, sim_t0, sims)
                if frames_exist_nb > 0:
                    print("Skipped {}/{} already existing renderings".format(frames_exist_nb, len(idx)))
            print("\n\nEnd of the simulation")

    def _cap_batch(self, B, imH, imW, frame_render_dict, sims):
        """Frames per library call, bounded by the page-locked memory a rank may hold: three slots of B frames, each frame
        its input bytes (image, float32 depth), both PNG scanline blocks and the drop / status tables -- about 15 bytes per
        pixel + 116 per drop: 7.9 MB at KITTI size (3 GB for 3 x 128 frames), 32 MB at Cityscapes full size.  Budget:
        RAIN_PINNED_MB per rank (default 8192 / the ranks on this host).  The library's device staging per slot is about
        nine times the pinned bytes (float64 images inside the chain)."""
        try:
            local = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))
        except ValueError:
            local = 1
        budget = float(os.environ.get('RAIN_PINNED_MB', 8192.0 / local)) * 2 ** 20
        n_max = int(sims['n_particles'].max()) if sims is not None else max((len(getattr(fr, 'table', ())) for fr in (frame_render_dict or ())), default=0)
        per_frame = imH * imW * 15 + 2 * imH + min(max(n_max, 1024), 2 ** 16) * 116
        return max(1, min(B, int(budget // (hip_backend.RR_PIPE_SLOTS * per_frame))))

    def _run_batches(self, hip, work, B, rs, imW, imH, frame_render_dict, fog_const, map_generator, folder_idx, folders_num, sim_t0, sims=None):
        """The frames of one (sequence, weather) run through the asynchronous pipeline.  The common case -- 8-bit PNG
        images, 16-bit PNG depth whose scaled size is the frame's, no angular noise, no environment-map files -- takes
        the batch-native route (one library call per batch and stage, nothing per frame under the interpreter lock);
        everything else the general one (per-frame Python on an I/O thread pool)."""
        ds = self.settings["depth_scale"]
        B = self._cap_batch(B, imH, imW, frame_render_dict, sims)
        native = (work and int(rs) == rs and rs >= 1 and int(ds) == ds and ds >= 1 and
                  not (bool(self.noise_std) and bool(self.noise_scale)) and not self.save_envmap and
                  os.environ.get('RAIN_NATIVE_IO', '1') != '0' and
                  all(it['image_file'].endswith('.png') and it['depth_file'].endswith('.png') for it in work))
        if native:
            # the first frame decides: files the library's readers take, whose scaled sizes are the run's frame size
            # (generator.py:355-381; a depth map of another size would make the reference crop the image: general route)
            i0, d0 = imgops._native_png(work[0]['image_file']), imgops._native_png(work[0]['depth_file'])
            native = (i0 is not None and d0 is not None and i0[4] == 8 and tuple(d0[3:5]) == (1, 16) and
                      (i0[1] // rs, i0[2] // rs) == (imW, imH) and ((d0[1] * ds) // rs, (d0[2] * ds) // rs) == (imW, imH) and
                      (rs != 1 or (d0[1], d0[2]) == (imW, imH)))
        if sims is not None:
            if work and not native:
                raise NotImplementedError("--device_particles needs the batch-native route: PNG frames and depth maps of matching "
                                          "scaled sizes, no environment-map files (RAIN_NATIVE_IO not 0)")
            return self._run_batches_native(hip, work, B, rs, imW, imH, frame_render_dict, fog_const, map_generator, folder_idx, folders_num,
                                            sim_t0, sims)
        run = self._run_batches_native if native else self._run_batches_general
        return run(hip, work, B, rs, imW, imH, frame_render_dict, fog_const, map_generator, folder_idx, folders_num, sim_t0)

    @staticmethod
    def _set_png_deflate(hip):
        """RR_OPT_PNG_DEFLATE for the driver: on (the library's own default is off), unless RAIN_PNG_DEVICE=0 says so -- or
        RAINHIP_OPTIONS already names option 15: the A/B switch that RainHip applied at construction has the last word
        (precedence: RAINHIP_OPTIONS > RAIN_PNG_DEVICE > the driver's default)."""
        named = {kv.split('=')[0].strip() for kv in os.environ.get('RAINHIP_OPTIONS', '').split(',') if '=' in kv}
        if str(hip_backend.RR_OPT_PNG_DEFLATE) in named:
            return
        hip.set_option(hip_backend.RR_OPT_PNG_DEFLATE, 0 if os.environ.get('RAIN_PNG_DEVICE', '1') == '0' else 1)

    def _run_batches_native(self, hip, work, B, rs, imW, imH, frame_render_dict, fog_const, map_generator, folder_idx, folders_num, sim_t0, sims=None):
        """Batch-native route.  Per batch, three whole-batch jobs, each one call (or two) into the library:
          decode  rr_io_read_frames (image bytes + depth metres straight into the slot's page-locked input blocks) and
                  rr_host_pack_frames (filter, draws, drop records into the slot's drop block) -- one batch ahead of the GPU;
          render  rr_pipeline_submit / rr_pipeline_wait on the prepared descriptors;
          encode  rr_io_write_frames (both files of every frame from the slot's scanline blocks) -- one batch behind.
        A slot's INPUT blocks are free again when its batch has been collected (the encoders only read the output
        blocks), so batch b+1 is decoded into slot (b+1) % 3 while batch b renders and batch b-1 is written."""
        stage = self._stage_pool()
        threads = self._io_threads()
        nslot = hip_backend.RR_PIPE_SLOTS
        batches = [work[a:a + B] for a in range(0, len(work), B)]
        if getattr(self, '_slots', None) is None or getattr(self, '_slots_hip', None) is not hip:
            self._slots, self._slots_hip = [None] * nslot, hip
        slots = self._slots
        n_sim = len(sims) if sims is not None else len(frame_render_dict)
        H, W = imH, imW
        env_w = hip.set_envmap_geometry(H, W, *map_generator.device_tables(H, W))
        cache = self.__dict__.setdefault('_omega_cache', {})     # a function of the map's shape alone: once per size, not per run
        if (H, env_w) not in cache:
            cache[(H, env_w)] = solid_angle.get_solid_angles(np.empty((H, env_w, 0)))               # generator.py:410
        hip.set_solid_angles(cache[(H, env_w)])
        # both output files leave the device as the zlib streams of their IDAT chunks (RR_OPT_PNG_DEFLATE, csrc/rr_deflate.h):
        # the encode stage below only frames chunks and checksums them.  RAIN_PNG_DEVICE=0: scanlines, deflated by the host.
        self._set_png_deflate(hip)
        # capacity of a frame's drop table: every streak of its simulated frame (the frame filter can only remove some), but
        # never more than the 2 ** 16 the reference allows AFTER the filter (generator.py:424): a simulated frame with more
        # streaks than that is fine as long as fewer land inside the image -- checked on the filtered counts below
        drops_cap = min(max(1024, int(sims['n_particles'].max()) if sims is not None else max(len(fr.table) for fr in frame_render_dict)), 2 ** 16)
        u8 = rs == 1                                              # at render scale 1 the bytes go to the GPU; a resized image is float64
        bg_dtype = np.uint8 if u8 else np.float64
        ds = int(self.settings["depth_scale"])
        # at render scale 1 the depth file's uint16 samples travel as they are (rr_prepass_in.depth_f64 = RR_DEPTH_U16: metres =
        # sample / 256 in float32, generator.py:366, is formed on the device): half the depth bytes over PCIe, no conversion pass
        # on the host.  RAIN_DEPTH_U16=0: float32 metres made by the host, as before.
        d16 = u8 and os.environ.get('RAIN_DEPTH_U16', '1') != '0'
        depth_dtype = np.uint16 if d16 else np.float32
        # ... and both files as the filtered scanlines their IDAT streams inflate to (rr_io_read_frames_rows): the scanline filters
        # are reversed on the device (k_png_unfilter), 40 % of the host's decode time.  RAIN_PNG_ROWS=0: pixels decoded by the host.
        rows_in = u8 and os.environ.get('RAIN_PNG_ROWS', '1') != '0'
        key = (B, H, W, env_w, 'png rows' if rows_in else np.dtype(bg_dtype), 'png rows' if rows_in else np.dtype(depth_dtype), False)
        pkey = (tuple(float(v) for v in fog_const), float(self.opacity_attenuation), self.rendering_strategy, sims is not None)
        for d in {os.path.dirname(it[k]) for it in work for k in ('out_rainy_path', 'out_rainy_mask_path')}:
            os.makedirs(d, exist_ok=True)
        encodes = [None] * nslot                                 # the slot's encode job (future) and its frames
        done = [0]

        pending = {}                                             # slots being page-locked by the helper thread (see below)

        def new_slot():
            return Generator._Slot(hip, B, H, W, env_w, bg_dtype, depth_dtype, False, drops_cap, png_rows=rows_in)

        def slot_for(si):
            if si in pending:
                slots[si] = pending.pop(si).result()
            sl = slots[si]
            if sl is None or sl.key != key or sl.drops_cap < drops_cap:
                if sl is not None:
                    sl.free(hip)
                sl = slots[si] = new_slot()
            if getattr(sl, 'prep', None) is None or sl.pkey != pkey:
                inputs = ((lambda k: dict(bg_png_rows=sl.bg[k], shape=(H, W), depth_png_rows=sl.depth[k])) if rows_in else
                          (lambda k: dict(bg=None if u8 else sl.bg[k], bg_u8=sl.bg[k] if u8 else None, depth=sl.depth[k])))
                frames = [dict(inputs(k), fog=fog_const, omega=None, drops=sl.drops[k],
                               opacity_attenuation=self.opacity_attenuation, strategy=1 if self.rendering_strategy == 'white' else 0)
                          for k in range(B)]
                outs = [dict(image_u8=None, rainy_png=sl.png_i[k], mask_png=sl.png_m[k], status=sl.status[k]) for k in range(B)]
                if sims is not None:                             # the descriptors point at these records: rewritten in place per batch
                    sl.sim_recs = [np.zeros(1, hip_backend.SIM_FRAME_DTYPE) for _ in range(B)]
                    sl.n_out = [np.zeros(1, np.int32) for _ in range(B)]
                    for k in range(B):
                        sl.sim_recs[k][0] = sims[0]
                        frames[k].update(sim=sl.sim_recs[k], drops_cap=sl.drops_cap)
                        outs[k]['n_drops'] = sl.n_out[k]
                sl.prep, sl.pkey = hip.pipeline_prepare(frames, outs), pkey
            return sl

        def decode_job(sl, items):
            """-> (frames of the batch in slot order, their drop counts): inputs of `sl` filled for the first len() frames."""
            if rows_in:
                st = hip_backend.io_read_frames_rows([it['image_file'] for it in items], [it['depth_file'] for it in items], H, W,
                                                     sl.raw_bg, sl.raw_depth, threads)
            elif u8:
                st = hip_backend.io_read_frames([it['image_file'] for it in items], [it['depth_file'] for it in items], H, W,
                                                sl.raw_bg, sl.raw_depth, threads, depth_u16=d16)
            else:
                st = hip_backend.io_read_frames_scaled([it['image_file'] for it in items], [it['depth_file'] for it in items], H, W,
                                                       int(rs), ds, sl.raw_bg, sl.raw_depth, threads)
            if sims is not None:                                # the drop tables are made on the GPU: only their settings travel
                counts = np.zeros(len(items), np.int64)
            else:
                tables = [frame_render_dict[it['f_name_idx'] % n_sim].table for it in items]
                counts = hip_backend.pack_frames(tables, [it['seeds'][-1] for it in items], self.db, imW, imH, sl.raw_drops,
                                                 sl.drops_cap, sl.drops_cap, threads)
            ok = [True] * len(items)
            for k in np.nonzero(st)[0]:                          # not a file the fast readers take: the general loader
                loaded = self._load_frame(items[k]['image_file'], items[k]['depth_file'], rs)
                if loaded is None:
                    print('Missing/Corrupted depth data (%s)' % items[k]['depth_file'])
                    ok[k] = False
                    continue
                bg, depth = loaded
                assert bg.shape[:2] == (H, W) and bg.dtype == bg_dtype and depth.shape == (H, W), "frames of one sequence share their size"
                samples = None
                if rows_in or d16:
                    # a 16-bit PNG depth map is sample / 256 in float32: the samples come back exactly -- anything else (an 8-bit
                    # or converted file that reached this route) must not be re-quantised silently
                    samples = np.rint(depth.astype(np.float64) * 256.0).astype(np.uint16)
                    if not np.array_equal(samples.astype(np.float32) / np.float32(256.0), depth.astype(np.float32)):
                        # one odd file must not end the run from the decode thread: the frame is skipped like a missing depth map
                        print("Skipping %s: depth values are not the samples of a 16-bit file / 256 (run with RAIN_NATIVE_IO=0 to "
                              "render such a sequence)" % items[k]['depth_file'])
                        ok[k] = False
                        continue
                if rows_in:                                         # as scanlines of filter type 0
                    np.copyto(sl.bg[k], hip_backend.png_rows_of(bg))
                    np.copyto(sl.depth[k], hip_backend.png_rows_of(samples))
                else:
                    np.copyto(sl.bg[k], bg)
                    np.copyto(sl.depth[k], samples if d16 else depth.astype(np.float32))
            order = [k for k in range(len(items)) if ok[k]]
            for dst in [k for k in range(len(order)) if not ok[k]]:          # close the gaps of skipped frames from the back
                src = order.pop()
                np.copyto(sl.bg[dst], sl.bg[src])
                np.copyto(sl.depth[dst], sl.depth[src])
                if sims is None:
                    sl.drops[dst][:counts[src]] = sl.drops[src][:counts[src]]
                order.insert(dst, src)
            # (rr_host_pack_frames stores at most drops_cap records and reports the full count)
            assert int(counts.max(initial=0)) <= sl.drops_cap, \
                "Assert that the number of drops doesn't overpass the uint16 rain_mask capacity"       # generator.py:424
            return [items[k] for k in order], [int(counts[k]) for k in order]

        def encode_job(sl, items, nds, gpu_ms):
            st = hip_backend.io_write_frames([it['out_rainy_path'] for it in items], [it['out_rainy_mask_path'] for it in items],
                                             sl.raw_png_i, sl.raw_png_m, W, H, threads)        # generator.py:466-467
            if st.any():
                bad = int(np.nonzero(st)[0][0])
                raise IOError("could not write %s / %s (%d)" % (items[bad]['out_rainy_path'], items[bad]['out_rainy_mask_path'], st[bad]))
            return [dict(file=it['out_rainy_path'], drops=nd, skipped=int(np.count_nonzero(sl.status[k][:nd])), gpu_ms=gpu_ms)
                    for k, (it, nd) in enumerate(zip(items, nds))]

        def finish(si):
            sl = slots[si]
            if sl is None or not sl.busy:
                return
            while not hip.pipeline_wait(si):                    # tile arena regrown: submit the batch again
                retries[0] += 1
                hip.pipeline_submit_prepared(si, sl.prep, sl.n_valid)
            sl.busy = False
            if not first_done[0]:
                first_done[0] = True
                self._mark('first batch collected (%d frames)' % sl.n_valid)
            ms = 1e3 * (time.time() - sl.t_submit) / max(sl.n_valid, 1)
            if sims is not None:                                # the counts only exist now
                sl.items = [(it, int(sl.n_out[k][0])) for k, (it, _) in enumerate(sl.items)]
                assert max((nd for _, nd in sl.items), default=0) <= sl.drops_cap, \
                    "Assert that the number of drops doesn't overpass the uint16 rain_mask capacity"   # generator.py:424
            encodes[si] = stage.submit(encode_job, sl, [it for it, _ in sl.items], [nd for _, nd in sl.items], ms)

        def drain(si):
            if encodes[si] is not None:
                for st_ in encodes[si].result():
                    self.stats.append(st_)
                    if st_['skipped'] and self.verbose:
                        print("\nTrace: %d of %d rain drops not rendered in %s" % (st_['skipped'], st_['drops'], st_['file']))
                encodes[si] = None

        t_loop0 = time.time()
        t_first = None
        retries, first_done = [0], [False]
        # Page-locking a slot (1 GB at 128 KITTI frames) takes 0.15 s: the first slot now, the others on a helper thread while the
        # first batch is decoded and rendered (rr_host_alloc is safe beside the submitting thread; r05: set-up 0.49 -> 0.17 s)
        if batches:
            slot_for(0)
        self._mark('first page-locked slot + descriptors')
        maker = None
        for si in range(1, min(nslot, len(batches))):
            sl = slots[si]
            if sl is None or sl.key != key or sl.drops_cap < drops_cap:
                if sl is not None:
                    sl.free(hip)
                    slots[si] = None
                if maker is None:
                    from concurrent.futures import ThreadPoolExecutor as _TPE
                    maker = _TPE(max_workers=1)
                pending[si] = maker.submit(new_slot)
        decoding = stage.submit(decode_job, slot_for(0), batches[0]) if batches else None
        for bi in range(len(batches)):
            si = bi % nslot
            sl = slots[si]
            items, nds = decoding.result()
            if bi == 0:
                self._mark('first batch decoded + packed')
            drain(si)                                            # the slot's previous files are written: its outputs may be overwritten
            sl.items = list(zip(items, nds))
            sl.n_valid = len(items)
            if sl.n_valid:
                for k, nd in enumerate(nds):
                    if sims is not None:                         # simulated frame f % n_sim with the draws of frame f (generator.py:318-321)
                        f_idx = items[k]['f_name_idx']
                        sl.sim_recs[k][0] = sims[f_idx % n_sim]
                        sl.sim_recs[k]['draw_seed'] = f_idx
                    else:
                        sl.prep.set_drop_count(k, nd)
                sl.t_submit = time.time()
                hip.pipeline_submit_prepared(si, sl.prep, sl.n_valid)
                if bi == 0:
                    self._mark('first batch submitted')
                sl.busy = True
                if t_first is None:
                    t_first = time.time()
                done[0] += sl.n_valid
            finish((bi - 1) % nslot)                             # collect the previous batch, start writing its files
            if bi + 1 < len(batches):                            # and decode the next one into the slot that is idle now
                sn = (bi + 1) % nslot
                finish(sn)
                if slots[sn] is not None and (slots[sn].key != key or slots[sn].drops_cap < drops_cap):
                    drain(sn)                                    # (a slot of another shape is replaced: its encoders first)
                decoding = stage.submit(decode_job, slot_for(sn), batches[bi + 1])
            if self.verbose:
                sys.stdout.write('\r          S. {} / {}, F. {} / {}   ({:.1f}s)'.format(
                    folder_idx + 1, folders_num, done[0], len(work), time.time() - sim_t0))
        for si in range(nslot):
            finish(si)
        for si in range(nslot):
            drain(si)
        for si in list(pending):                                 # (a run shorter than its slots: keep them for the next run)
            slots[si] = pending.pop(si).result()
        if maker is not None:
            maker.shutdown(wait=True)
        t_end = time.time()
        self.timing.append(dict(frames=done[0], first_batch_s=(t_first or t_end) - t_loop0, total_s=t_end - t_loop0, route='native',
                                steady_frames_per_s=(max(done[0] - B, 0) / (t_end - t_first)) if t_first and t_end > t_first else None,
                                arena_resubmits=retries[0], setup=list(self._marks)))

    def _run_batches_general(self, hip, work, B, rs, imW, imH, frame_render_dict, fog_const, map_generator, folder_idx, folders_num, sim_t0):
        """Batches of B frames through the three-slot asynchronous pipeline (rr_pipeline_submit / rr_pipeline_wait):
        decode + drop tables on the I/O threads (into pinned buffers), upload | kernels | download overlapped inside
        the library, deflate + file writes on the I/O threads again.  FOG.fog_rain_layer (generator.py:386),
        map_generator.generate_map (:400), the xyY conversion (:407-408), the streak loop (:431-452), the epilogue
        (:461-466), the mask's colour map (:467) and the PNG filtering all run on the GPU; the host provides the scalar
        fog constants and, once per frame size, the projection tables of the environment map."""
        pool = self._io_pool()
        nslot = hip_backend.RR_PIPE_SLOTS
        batches = [work[a:a + B] for a in range(0, len(work), B)]
        # the pinned slots live on the Generator and are re-used by every (sequence, weather) run of the process: a fresh
        # set per run would leak ~0.7 GB of page-locked memory each time (pinned memory is only freed explicitly)
        if getattr(self, '_slots', None) is None or getattr(self, '_slots_hip', None) is not hip:
            self._slots, self._slots_hip = [None] * nslot, hip
        slots = self._slots
        n_sim = len(frame_render_dict)
        state = dict(geom=None, env_w=0, omega=None, done=0)

        def decode_batch(bi):
            """Start the I/O-thread work of batch bi; returns the futures (the slot's buffers are filled at submit)."""
            return [pool.submit(self._decode_into, None, k, it, rs, frame_render_dict[it['f_name_idx'] % n_sim].table, imW, imH, it['seeds'])
                    for k, it in enumerate(batches[bi])]

        def finish(si):
            """Wait for slot si's GPU work, hand its frames to the encoders."""
            sl = slots[si]
            if sl is None or not sl.busy:
                return
            while not hip.pipeline_wait(si):                    # tile arena regrown: submit the batch again
                hip.pipeline_submit_prepared(si, sl.prep, sl.n_valid)
            sl.busy = False
            dt = time.time() - sl.t_submit
            for k, (it, nd) in enumerate(sl.items):
                fut = pool.submit(self._encode, sl, k, it, nd)
                sl.encodes.append((fut, it, nd, 1e3 * dt / max(len(sl.items), 1)))

        def drain(si):
            """Slot si's buffers are free again once its encoders are done."""
            sl = slots[si]
            if sl is None:
                return
            for fut, it, nd, ms in sl.encodes:
                n_skip = fut.result()
                self.stats.append(dict(file=it['out_rainy_path'], drops=nd, skipped=n_skip, gpu_ms=ms))
                if n_skip and self.verbose:
                    print("\nTrace: %d of %d rain drops not rendered in %s" % (n_skip, nd, it['out_rainy_path']))
            sl.encodes = []

        ahead = {}
        t_loop0 = time.time()
        t_first = None
        for bi in range(len(batches)):
            for bj in range(bi, min(bi + 2, len(batches))):     # decode two batches ahead of the GPU
                if bj not in ahead:
                    ahead[bj] = decode_batch(bj)
            si = bi % nslot
            finish(si)                                           # (no-op unless fewer batches than slots ran in between)
            drain(si)                                            # its deflate jobs had two iterations to finish
            loaded = [f.result() for f in ahead.pop(bi)]
            valid = [(it, ld) for it, ld in zip(batches[bi], loaded) if ld is not None]
            for it, ld in zip(batches[bi], loaded):
                if ld is None:
                    print('Missing/Corrupted depth data (%s)' % it['depth_file'])
            if not valid:
                continue
            H, W = valid[0][1][0].shape[:2]
            if state['geom'] != (H, W):
                for sj in range(nslot):                         # a new frame size: let the pipeline run dry first
                    finish(sj)
                    drain(sj)
                state['env_w'] = hip.set_envmap_geometry(H, W, *map_generator.device_tables(H, W))
                state['omega'] = solid_angle.get_solid_angles(np.empty((H, state['env_w'], 0)))    # generator.py:410
                hip.set_solid_angles(state['omega'])            # resident on the device: not uploaded with every batch
                self._set_png_deflate(hip)                      # (see _run_batches_native)
                state['geom'] = (H, W)
            bg0, dep0 = valid[0][1][0], valid[0][1][1]
            need_drops = max(len(ld[2]) for _, ld in valid)
            key = (B, H, W, state['env_w'], bg0.dtype, dep0.dtype, bool(self.save_envmap))
            sl = slots[si]
            if sl is None or sl.key != key or sl.drops_cap < need_drops:
                if sl is not None:                               # (finish / drain above left it idle)
                    sl.free(hip)
                sl = slots[si] = Generator._Slot(hip, B, H, W, state['env_w'], bg0.dtype, dep0.dtype, self.save_envmap,
                                                 max(need_drops + need_drops // 4, 1024))
            # the batch's descriptors are made once per slot (the buffers do not move) and for the run's constants
            pkey = (tuple(float(v) for v in fog_const), float(self.opacity_attenuation), self.rendering_strategy)
            if getattr(sl, 'prep', None) is None or sl.pkey != pkey:
                u8 = bg0.dtype == np.uint8
                frames = [dict(bg=None if u8 else sl.bg[k], bg_u8=sl.bg[k] if u8 else None, depth=sl.depth[k], fog=fog_const, omega=None,
                               drops=sl.drops[k], opacity_attenuation=self.opacity_attenuation,
                               strategy=1 if self.rendering_strategy == 'white' else 0) for k in range(B)]
                outs = []
                for k in range(B):
                    o = dict(image_u8=None, rainy_png=sl.png_i[k], mask_png=sl.png_m[k], status=sl.status[k])
                    if self.save_envmap:
                        o['env_bgr_u8'] = sl.env[k]
                    outs.append(o)
                sl.prep, sl.pkey = hip.pipeline_prepare(frames, outs), pkey
            sl.items = []
            for k, (it, (bg, depth, drops)) in enumerate(valid):
                assert bg.shape[:2] == (H, W) and bg.dtype == bg0.dtype, "frames of one sequence share their size"
                np.copyto(sl.bg[k], bg)
                np.copyto(sl.depth[k], depth)
                nd = len(drops)
                sl.drops[k][:nd] = drops
                sl.prep.set_drop_count(k, nd)
                sl.items.append((it, nd))
            sl.n_valid = len(valid)
            sl.t_submit = time.time()
            hip.pipeline_submit_prepared(si, sl.prep, sl.n_valid)
            if t_first is None:
                t_first = time.time()                            # set-up (pinned buffers, first decodes) ends here
            sl.busy = True
            state['done'] += len(valid)
            # the GPU now has this batch queued: collect the previous one and start deflating it right away
            finish((bi - 1) % nslot)
            if self.verbose:
                sys.stdout.write('\r          S. {} / {}, F. {} / {}   ({:.1f}s)'.format(
                    folder_idx + 1, folders_num, state['done'], len(work), time.time() - sim_t0))
        for si in range(nslot):
            finish(si)
        for si in range(nslot):
            drain(si)
        t_end = time.time()
        self.timing.append(dict(frames=state['done'], first_batch_s=(t_first or t_end) - t_loop0, total_s=t_end - t_loop0,
                                steady_frames_per_s=(max(state['done'] - B, 0) / (t_end - t_first)) if t_first and t_end > t_first else None))
