"""Drop-in for the reference's common/generator.py: `Generator(args).run()` with the same
argument object (fields consumed at reference generator.py:25-67), the same input trees and
the same output tree.  The per-drop loop of the reference (generator.py:431-452 ->
compute_drop -> add_drop_to_image) is replaced by ONE call per batch of frames into the HIP
library (rr_render_frames); everything around it (file walking, conflict strategy, seeding,
streak filter, pre-pass, saving) follows the reference.

Multi-GPU: started under torch.distributed.run (or with RANK/WORLD_SIZE set) every rank
renders frames idx[rank::world] of each sequence on its own GPU; the streak database is
packed on rank 0 and broadcast once (sharding.broadcast_streak_db); no other collective.
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
        self.batch = int(os.environ.get('RAIN_BATCH', '4'))
        self.rank, self.world = sharding.rank_world()
        self.device = int(getattr(args, 'device', os.environ.get('LOCAL_RANK', '0')))
        self._hip = None
        self._pool = None
        self._saves = []
        self.stats = []
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
        """Threads for PNG decode / encode (PIL's codecs release the GIL).  SURVEY 8f next #3: the host I/O
        around the GPU call is what bounds the driver end to end, so it runs ahead of / behind the GPU."""
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=max(2, min(32, (os.cpu_count() or 4))))
        return self._pool

    def _pack(self, frame, imW, imH, seed):
        """Frame filter + drop table with the frame's random draws (generator.py:318,413-425)."""
        keep = hip_backend.filter_streaks(frame.table, imW, imH)                    # generator.py:413-420
        assert len(keep) <= 2 ** 16, "Assert that the number of drops doesn't overpass the uint16 rain_mask capacity"
        return hip_backend.pack_drops(frame.table, keep, self.db, self.noise_std, self.noise_scale, seed=seed)

    def _prepare_frame(self, image_file, depth_file, rs, seed, frame, imW, imH, pack):
        """Worker-thread part of one frame: decode image + depth and (when no state is shared between
        frames) build the drop table from the library's own per-frame generator."""
        loaded = self._load_frame(image_file, depth_file, rs)
        if loaded is None:
            return None
        return loaded[0], loaded[1], (self._pack(frame, imW, imH, seed) if pack else None)

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
            depth = imgops.resize_linear(depth.astype(np.float64), int(depthHW[1]), int(depthHW[0]))
        assert np.all(np.array(depth.shape[:2]) <= np.array(bg.shape[:2])), "Depth cannot be larger than the image"
        if not np.all(np.array(depth.shape[:2]) == np.array(bg.shape[:2])):
            bg = my_utils.crop_center(bg, depth.shape[0], depth.shape[1])
        return np.ascontiguousarray(bg), depth

    def _save_frame(self, p, o):
        os.makedirs(os.path.dirname(p['out_rainy_path']), exist_ok=True)
        os.makedirs(os.path.dirname(p['out_rainy_mask_path']), exist_ok=True)
        imgops.imsave_rgb(p['out_rainy_path'], o['image_u8'])                      # generator.py:466
        imgops.imsave_scalar(p['out_rainy_mask_path'], o['mask'])                   # generator.py:467
        if self.save_envmap:
            os.makedirs(os.path.dirname(p['out_env_path']), exist_ok=True)
            env_bgr = o['env_bgr_u8'] / 255.0                                      # generator.py:469 (plt.imsave of a float map)
            imgops.imsave_rgb(p['out_env_path'], (np.clip(env_bgr[..., ::-1], 0, 1) * 255).astype(np.uint8))

    def _drain_saves(self, keep=0):
        while len(self._saves) > keep:
            self._saves.pop(0).result()

    def _flush(self, pending):
        """Render the pending frames in one library call; their outputs are encoded on the I/O pool."""
        if not pending:
            return
        t0 = time.time()
        # fog attenuation + environment map + streak rendering, all on the GPU in one call
        outs = self._hip_ctx().pipeline_frames([p['frame'] for p in pending], want_env_u8=self.save_envmap, want_mask_i32=False)
        dt = time.time() - t0
        for p, o in zip(pending, outs):
            self._saves.append(self._io_pool().submit(self._save_frame, dict(p, frame=None), o))
            n_skip = int(np.count_nonzero(o['status']))
            self.stats.append(dict(file=p['out_rainy_path'], drops=len(o['status']), skipped=n_skip,
                                   gpu_ms=1e3 * dt / len(pending)))
            if n_skip and self.verbose:
                print("\nTrace: %d of %d rain drops not rendered in %s" % (n_skip, len(o['status']), p['out_rainy_path']))
        pending.clear()
        self._drain_saves(keep=4 * self.batch)          # bound the frames in flight

    def compute_drop(self, bg, drop_dict, rainy_bg, rainy_mask, rainy_saturation_mask):
        """Single-drop compatibility seam with the reference's signature (generator.py:119-191):
        consumes the same RNG draws, composites ONE streak into rainy_bg / rainy_mask (in place and
        returned) by calling the library with a one-record drop table.  self.env_map_xyY,
        self.solid_angle_map, self.db and the camera must be set like Generator.run does.  Returns
        (rainy_bg, rainy_mask, rainy_saturation_mask, None, blended, minC): `blended` is None when the
        reference would have printed "Erroneous drop"; the tile itself stays on the GPU.  One launch
        chain per drop -- use run() / rr_render_frames for throughput."""
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
        out = self._hip_ctx().render_frames([dict(bg=bg, rainy_bg=rainy_bg, env_xyY=self.env_map_xyY,
                                                  omega=self.solid_angle_map, drops=drops,
                                                  opacity_attenuation=self.opacity_attenuation,
                                                  strategy=1 if self.rendering_strategy == 'white' else 0)])[0]
        ok = out['status'][0] == 0
        if ok:
            rainy_bg[...] = out['rainy_bg']
            rainy_mask += out['mask']
        else:
            print('Erroneous drop (status %d)' % out['status'][0])
        minC = np.array([drops['x0'][0], drops['y0'][0]])
        return rainy_bg, rainy_mask, rainy_saturation_mask, None, (rainy_bg if ok else None), minC

    def run(self):
        folders_num = len(self.images)
        for folder_idx, sequence in enumerate(self.sequences):
            print('\nSequence: ' + sequence)
            depth_folder = self.depth[sequence]
            for sim_idx, sim_weather in enumerate(self.weather):
                weather, fallrate = sim_weather["weather"], sim_weather["fallrate"]
                out_seq_dir = os.path.join(self.output_root, sequence)
                out_dir = os.path.join(out_seq_dir, weather, '{}mm'.format(fallrate))
                sim_file = self.particles[sequence][sim_idx]
                if os.path.exists(out_dir):                                         # generator.py:213-226
                    if self.conflict_strategy in ("skip", "overwrite"):
                        pass
                    elif self.conflict_strategy == "rename_folder":
                        shift = 0
                        while os.path.exists(out_dir + '_copy%05d' % shift):
                            shift += 1
                        out_dir = out_dir + '_copy%05d' % shift
                    else:
                        raise NotImplementedError
                os.makedirs(out_dir, exist_ok=True)
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
                hip = self._hip_ctx()
                sharding.load_and_broadcast_streak_db(self.db, hip, self.rank, self.world)
                hip.set_camera(hip_backend.make_camera(self.focal, self.f_number, self.exposure))
                hip.set_prepass_kernels(imgops.gaussian_kernel(25, 25), imgops.gaussian_kernel(15, 0))
                fog_const = FOG.constants()
                geom_hw, env_w = None, 0
                self.db.load_streaks_from_xml(self.dataset, self.settings, [imW, imH], use_pickle=False, verbose=self.verbose)
                frame_render_dict = list(self.db.streaks_simulator.values())

                f_end = len(files) if self.frame_end is None else min(self.frame_end, len(files))
                if self.frames:
                    idx = np.unique(np.clip(self.frames, 0, f_end - 1)).tolist()
                else:
                    idx = list(range(self.frame_start, f_end, self.frame_step))
                print("{} images".format(len(idx)))
                idx = sharding.shard(idx, self.rank, self.world)
                frames_exist_nb = 0
                pending = []
                sim_t0 = time.time()
                # work items first (skip / overwrite decisions), so that image + depth decoding can run ahead
                # of the GPU on the I/O pool; everything that touches the legacy global RNG stays on this thread
                work = []
                for i in idx:
                    image_file, depth_file = files[i], depth_files[i]
                    assert os.path.exists(image_file), "Image file {} does not exist".format(image_file)
                    assert os.path.exists(depth_file), "Depth file {} does not exist".format(depth_file)
                    file_name = os.path.split(image_file)[-1]
                    out_rainy_path = os.path.join(out_dir, 'rainy_image', '{}.png'.format(file_name[:-4]))
                    out_rainy_mask_path = os.path.join(out_dir, 'rain_mask', '{}.png'.format(file_name[:-4]))
                    out_env_path = os.path.join(out_seq_dir, 'envmap', '{}.png'.format(file_name[:-4]))
                    if os.path.exists(out_rainy_path) or os.path.exists(out_rainy_mask_path):
                        if self.conflict_strategy == "skip":
                            frames_exist_nb += 1
                            continue
                        elif self.conflict_strategy == "overwrite":
                            pass
                        else:
                            raise NotImplementedError
                    work.append((i, image_file, depth_file, out_rainy_path, out_rainy_mask_path, out_env_path))
                ahead = max(2 * self.batch, 4)
                noisy = bool(self.noise_scale) and bool(self.noise_std)
                loads = {}
                for f_idx, (i, image_file, depth_file, out_rainy_path, out_rainy_mask_path, out_env_path) in enumerate(work):
                    for j in range(f_idx, min(f_idx + ahead, len(work))):
                        if j not in loads:
                            # f_name_idx = i (generator.py:312; nuscenes remap not supported)
                            loads[j] = self._io_pool().submit(self._prepare_frame, work[j][1], work[j][2], rs, work[j][0],
                                                              frame_render_dict[work[j][0] % len(frame_render_dict)],
                                                              imW, imH, not noisy)
                    loaded = loads.pop(f_idx).result()
                    f_name_idx = i
                    np.random.seed(f_name_idx)                                       # generator.py:318 (kept for callers)
                    frame = frame_render_dict[f_name_idx % len(frame_render_dict)]
                    if loaded is None:
                        print('Missing/Corrupted depth data (%s)' % depth_file)
                        continue
                    bg, depth, drops = loaded
                    H, W = bg.shape[:2]
                    # FOG.fog_rain_layer (generator.py:386), map_generator.generate_map (:400) and the xyY
                    # conversion (:407-408) run on the GPU inside rr_pipeline_frames; the host only provides
                    # the scalar fog constants and, once per frame size, the projection tables
                    if geom_hw != (H, W):
                        self._flush(pending)
                        env_w = hip.set_envmap_geometry(H, W, *map_generator.device_tables(H, W))
                        geom_hw = (H, W)
                    omega = solid_angle.get_solid_angles(np.empty((H, env_w, 0)))    # generator.py:410
                    if drops is None:
                        # angular noise rotates the streak end points IN the shared table (generator.py:152-161),
                        # so frames that reuse a simulator frame must be packed in order, on this thread
                        drops = self._pack(frame, imW, imH, f_name_idx)
                    pending.append(dict(frame=dict(bg=None if bg.dtype == np.uint8 else bg,
                                                   bg_u8=bg if bg.dtype == np.uint8 else None,
                                                   depth=depth, fog=fog_const, omega=omega, drops=drops,
                                                   opacity_attenuation=self.opacity_attenuation,
                                                   strategy=1 if self.rendering_strategy == 'white' else 0),
                                        out_rainy_path=out_rainy_path, out_rainy_mask_path=out_rainy_mask_path,
                                        out_env_path=out_env_path))
                    if len(pending) >= self.batch:
                        self._flush(pending)
                    if self.verbose:
                        sys.stdout.write('\r          S. {} / {}, F. {} / {}   ({:.1f}s)'.format(
                            folder_idx + 1, folders_num, f_idx + 1, len(work), time.time() - sim_t0))
                self._flush(pending)
                self._drain_saves()
                if frames_exist_nb > 0:
                    print("Skipped {}/{} already existing renderings".format(frames_exist_nb, len(idx)))
            print("\n\nEnd of the simulation")
