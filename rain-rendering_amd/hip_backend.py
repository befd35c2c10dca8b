"""ctypes binding of librainhip.so (include/rainhip.h) and the host-side packing of the
per-frame drop table.

There is no CPU implementation behind this module: if the shared library or a gfx950
device is missing, construction fails loudly.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RAINHIP_LIB: another build of the same library -- the phase-clock build of scripts/phase_timing.sh; the C library itself reads no environment)
LIB_PATH = os.environ.get('RAINHIP_LIB') or os.path.join(_HERE, 'csrc', 'librainhip.so')

RR_MAX_FOV = 32
RR_E_ARENA = -5
RR_PIPE_SLOTS = 3
RR_PRE_ENV_ONLY = 1
RR_OPT_DEDUP, RR_OPT_GENERAL_FOV, RR_OPT_FOV_THREADS, RR_OPT_FOV_DROPS_PER_THREAD, RR_OPT_DEPTH_OCCLUSION = 1, 2, 3, 4, 5
RR_OPT_BLUR_WORKGROUPS = 6
RR_OPT_COMPOSITE_F64 = 7
RR_OPT_COPY_KERNELS = 8
RR_OPT_PADDED_TEXTURES = 9
RR_OPT_FOV_F32 = 10
RR_OPT_COMPOSITE_WAVES = 11
RR_OPT_FOV_DDA = 12
RR_OPT_PIPELINE_F32 = 13
RR_OPT_WILD_PIXELS = 14
RR_OPT_PNG_DEFLATE = 15
RR_OPT_COMPOSITE_U16 = 16
RR_OPT_BLUR_DMA = 17
RR_OPT_FOV_FILL_RULE = 18
RR_OPT_BIN_ROWS = 19
RR_OPT_COMPOSITE_BATCH = 20
RR_OPT_COLOUR_STREAM = 21
RR_OPT_TILE_ROWS = 22
RR_OPT_ROWS_SHARES = 23
RR_OUT_RAINY_F32, RR_OUT_ENV_F32 = 1, 2                 # rr_prepass_out.out_types
RR_IN_BG_PNG_ROWS, RR_DEPTH_PNG_ROWS = 32, 3              # a file's filtered scanlines (rr_io_read_frames_rows): un-filtered on the device
RR_DEPTH_U16 = 2                                        # rr_prepass_in.depth_f64: the uint16 samples of the depth file (metres = sample / 256)
RR_IN_BG_F32, RR_IN_BG_U8, RR_IN_ENV_F32, RR_IN_RAINY_F32, RR_IN_RAINY_U8 = 1, 2, 4, 8, 16      # rr_frame_in.in_types

# numpy mirror of rr_drop (112 bytes)
DROP_DTYPE = np.dtype([
    ('x0', '<i4'), ('y0', '<i4'), ('x1', '<i4'), ('y1', '<i4'),
    ('max_width', '<i4'), ('length', '<i4'), ('type', '<i4'), ('tex_index', '<i4'),
    ('iw1', '<f8'), ('iw2', '<f8'),
    ('wps', '<f8', (3,)), ('wpe', '<f8', (3,)),
    ('rot_cos', '<f8'), ('rot_sin', '<f8'),
], align=True)


# numpy mirror of rr_sim_frame (the particle generator's per-frame settings, include/rainhip.h)
SIM_FRAME_DTYPE = np.dtype([
    ('sensor_w', '<i4'), ('sensor_h', '<i4'), ('render_scale', '<i4'), ('n_particles', '<i4'),
    ('key0', '<u4'), ('key1', '<u4'), ('frame', '<u4'), ('draw_seed', '<u4'), ('table', '<i4'), ('reserved', '<i4'),
    ('fpx', '<f8'), ('exposure_s', '<f8'), ('speed_mps', '<f8'), ('wind_sigma', '<f8'), ('margin', '<f8'), ('min_px', '<f8'),
    ('z_far', '<f8'),
], align=True)


class rr_camera(ctypes.Structure):
    _fields_ = [('focal_m', ctypes.c_double), ('focal_sq', ctypes.c_double), ('f_number', ctypes.c_double),
                ('focus_plane', ctypes.c_double), ('exposure_s', ctypes.c_double), ('radius', ctypes.c_double),
                ('sensor_px', ctypes.c_double), ('tau_zero', ctypes.c_double),
                ('fov_cos', ctypes.c_double), ('fov_sin', ctypes.c_double),
                ('phi_cos', ctypes.c_double * RR_MAX_FOV), ('phi_sin', ctypes.c_double * RR_MAX_FOV),
                ('n_fov', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class rr_frame_in(ctypes.Structure):
    _fields_ = [('H', ctypes.c_int32), ('W', ctypes.c_int32), ('He', ctypes.c_int32), ('We', ctypes.c_int32),
                ('bg', ctypes.c_void_p), ('rainy_bg', ctypes.c_void_p), ('env_xyY', ctypes.c_void_p),
                ('omega', ctypes.c_void_p), ('drops', ctypes.c_void_p),
                ('n_drops', ctypes.c_int32), ('strategy', ctypes.c_int32),
                ('opacity_attenuation', ctypes.c_double), ('depth', ctypes.c_void_p), ('depth_f64', ctypes.c_int32),
                ('in_types', ctypes.c_int32), ('ext', ctypes.c_void_p), ('n_drops_dev', ctypes.c_void_p), ('sim', ctypes.c_void_p)]


class rr_ext_tile(ctypes.Structure):
    _fields_ = [('alpha', ctypes.c_void_p), ('tw', ctypes.c_int32), ('th', ctypes.c_int32), ('min_x', ctypes.c_int32),
                ('min_y', ctypes.c_int32), ('n_poly', ctypes.c_int32), ('reserved', ctypes.c_int32), ('poly_xy', ctypes.c_void_p)]


class rr_frame_out(ctypes.Structure):
    _fields_ = [('rainy_rgb', ctypes.c_void_p), ('rainy_bg_out', ctypes.c_void_p), ('mask_f64', ctypes.c_void_p),
                ('mask_i32', ctypes.c_void_p), ('drop_status', ctypes.c_void_p),
                ('rainy_png', ctypes.c_void_p), ('mask_png', ctypes.c_void_p), ('drop_colour', ctypes.c_void_p),
                ('n_drops_out', ctypes.c_void_p)]


class rr_kernel_stat(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 32), ('launches', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('total_ms', ctypes.c_double)]


RR_MAX_TAPS = 33


class rr_prepass_kernels(ctypes.Structure):
    _fields_ = [('fog_ksize', ctypes.c_int32), ('env_ksize', ctypes.c_int32),
                ('fog_w', ctypes.c_double * RR_MAX_TAPS), ('env_w', ctypes.c_double * RR_MAX_TAPS)]


class rr_prepass_in(ctypes.Structure):
    _fields_ = [('H', ctypes.c_int32), ('W', ctypes.c_int32), ('bg', ctypes.c_void_p), ('depth', ctypes.c_void_p),
                ('depth_f64', ctypes.c_int32), ('mode', ctypes.c_int32),
                ('beta_ext', ctypes.c_double), ('beta_hg', ctypes.c_double),
                ('irr_num', ctypes.c_double), ('irr_den', ctypes.c_double), ('bg_u8', ctypes.c_void_p),
                ('in_types', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class rr_streak_table(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int64), ('wps', ctypes.c_void_p), ('wpe', ctypes.c_void_p), ('ips', ctypes.c_void_p),
                ('ipe', ctypes.c_void_p), ('iw1', ctypes.c_void_p), ('iw2', ctypes.c_void_p), ('ratio', ctypes.c_void_p),
                ('max_width', ctypes.c_void_p), ('length', ctypes.c_void_p), ('type', ctypes.c_void_p)]


class rr_prepass_out(ctypes.Structure):
    _fields_ = [('rainy_bg', ctypes.c_void_p), ('env_xyY', ctypes.c_void_p), ('env_bgr_u8', ctypes.c_void_p),
                ('out_types', ctypes.c_int32), ('reserved', ctypes.c_int32)]


EXPORTS = ['rr_version', 'rr_create', 'rr_destroy', 'rr_last_error', 'rr_set_streak_db', 'rr_set_streak_db_device',
           'rr_set_camera', 'rr_render_frames', 'rr_render_frames_device', 'rr_synchronize', 'rr_profile_enable',
           'rr_profile_reset', 'rr_profile_read', 'rr_sizeof_drop', 'rr_sizeof_camera', 'rr_sizeof_frame_in',
           'rr_sizeof_frame_out', 'rr_set_prepass_kernels', 'rr_set_envmap_geometry', 'rr_envmap_width',
           'rr_prepass_frames', 'rr_prepass_frames_device', 'rr_pipeline_frames', 'rr_sizeof_prepass_in',
           'rr_sizeof_prepass_out', 'rr_sizeof_prepass_kernels', 'rr_host_drop_draws', 'rr_batch_counts', 'rr_set_option',
           'rr_pipeline_submit', 'rr_pipeline_wait', 'rr_host_alloc', 'rr_host_last_error', 'rr_host_free', 'rr_bcast_streak_db', 'rr_bcast_selftest', 'rr_host_parse_particles',
           'rr_sizeof_particle', 'rr_sizeof_particle_frame', 'rr_set_colormap', 'rr_host_frame_draws', 'rr_host_assemble_drops',
           'rr_sizeof_streak_table', 'rr_png_info', 'rr_png_read_bgr8', 'rr_png_read_gray16', 'rr_png_write_scanlines',
           'rr_deflate_bound', 'rr_deflate_fast', 'rr_inflate_fast', 'rr_adler32', 'rr_crc32', 'rr_host_pack_frames', 'rr_io_read_frames', 'rr_io_read_frames_u16', 'rr_io_read_frames_rows', 'rr_io_read_frames_scaled', 'rr_io_write_frames', 'rr_set_particle_tables', 'rr_generate_drops_device', 'rr_generate_drops', 'rr_set_solid_angles',
           'rr_sizeof_sim_frame']

_lib = None


def load_library(path=None):
    """Load librainhip.so and check the struct layouts against the header."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError("librainhip.so not found at %s -- run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback" % p)
    lib = ctypes.CDLL(p)
    lib.rr_last_error.restype = ctypes.c_char_p
    lib.rr_last_error.argtypes = [ctypes.c_void_p]
    lib.rr_bcast_streak_db.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32]
    lib.rr_bcast_selftest.argtypes = [ctypes.c_void_p]
    lib.rr_host_last_error.restype = ctypes.c_char_p
    lib.rr_host_last_error.argtypes = [ctypes.c_void_p]
    lib.rr_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    lib.rr_destroy.argtypes = [ctypes.c_void_p]
    lib.rr_set_streak_db.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_int32]
    lib.rr_set_streak_db_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
    lib.rr_set_camera.argtypes = [ctypes.c_void_p, ctypes.POINTER(rr_camera)]
    lib.rr_render_frames.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(rr_frame_in),
                                     ctypes.POINTER(rr_frame_out)]
    lib.rr_render_frames_device.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(rr_frame_in),
                                            ctypes.POINTER(rr_frame_out), ctypes.c_void_p]
    lib.rr_synchronize.argtypes = [ctypes.c_void_p]
    lib.rr_profile_enable.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.rr_profile_reset.argtypes = [ctypes.c_void_p]
    lib.rr_profile_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(rr_kernel_stat), ctypes.c_int32]
    lib.rr_set_prepass_kernels.argtypes = [ctypes.c_void_p, ctypes.POINTER(rr_prepass_kernels)]
    lib.rr_set_envmap_geometry.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_void_p, ctypes.c_void_p]
    lib.rr_envmap_width.argtypes = [ctypes.c_void_p]
    lib.rr_prepass_frames.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(rr_prepass_in),
                                      ctypes.POINTER(rr_prepass_out)]
    lib.rr_prepass_frames_device.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(rr_prepass_in),
                                             ctypes.POINTER(rr_prepass_out), ctypes.c_void_p]
    lib.rr_pipeline_frames.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(rr_prepass_in),
                                       ctypes.POINTER(rr_frame_in), ctypes.POINTER(rr_frame_out),
                                       ctypes.POINTER(rr_prepass_out)]
    lib.rr_batch_counts.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]
    lib.rr_set_option.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    lib.rr_pipeline_submit.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(rr_prepass_in),
                                       ctypes.POINTER(rr_frame_in), ctypes.POINTER(rr_frame_out), ctypes.POINTER(rr_prepass_out)]
    lib.rr_pipeline_wait.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.rr_host_alloc.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64]
    lib.rr_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.rr_set_colormap.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.rr_host_frame_draws.restype = ctypes.c_int64
    lib.rr_host_frame_draws.argtypes = [ctypes.POINTER(rr_streak_table), ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                        ctypes.c_uint32, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.rr_host_assemble_drops.argtypes = [ctypes.POINTER(rr_streak_table), ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.rr_sizeof_streak_table() == ctypes.sizeof(rr_streak_table)
    lib.rr_png_info.argtypes = [ctypes.c_char_p] + [ctypes.POINTER(ctypes.c_int32)] * 4
    lib.rr_png_read_bgr8.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    lib.rr_png_read_gray16.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    lib.rr_png_write_scanlines.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.rr_deflate_bound.argtypes, lib.rr_deflate_bound.restype = [ctypes.c_int64], ctypes.c_int64
    lib.rr_deflate_fast.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    lib.rr_deflate_fast.restype = ctypes.c_int64
    lib.rr_inflate_fast.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    lib.rr_host_pack_frames.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_int32, ctypes.c_void_p]
    lib.rr_io_read_frames.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]
    lib.rr_io_read_frames_u16.argtypes = lib.rr_io_read_frames.argtypes
    lib.rr_io_read_frames_rows.argtypes = lib.rr_io_read_frames.argtypes
    lib.rr_io_read_frames_scaled.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                             ctypes.c_void_p]
    lib.rr_io_write_frames.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    for fn in (lib.rr_adler32, lib.rr_crc32):
        fn.restype = ctypes.c_uint32
        fn.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int64]
    lib.rr_host_parse_particles.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    lib.rr_host_drop_draws.argtypes = [ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double,
                                       ctypes.c_void_p, ctypes.c_void_p]
    lib.rr_set_particle_tables.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    lib.rr_generate_drops_device.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    lib.rr_set_solid_angles.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    lib.rr_generate_drops.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    assert lib.rr_sizeof_sim_frame() == SIM_FRAME_DTYPE.itemsize, (lib.rr_sizeof_sim_frame(), SIM_FRAME_DTYPE.itemsize)
    assert lib.rr_sizeof_prepass_in() == ctypes.sizeof(rr_prepass_in)
    assert lib.rr_sizeof_prepass_out() == ctypes.sizeof(rr_prepass_out)
    assert lib.rr_sizeof_prepass_kernels() == ctypes.sizeof(rr_prepass_kernels)
    assert lib.rr_sizeof_drop() == DROP_DTYPE.itemsize == 112, (lib.rr_sizeof_drop(), DROP_DTYPE.itemsize)
    assert lib.rr_sizeof_camera() == ctypes.sizeof(rr_camera)
    assert lib.rr_sizeof_frame_in() == ctypes.sizeof(rr_frame_in)
    assert lib.rr_sizeof_frame_out() == ctypes.sizeof(rr_frame_out)
    if path is None:
        _lib = lib
    return lib


def make_camera(focal_m, f_number, exposure_ms, focus_plane=6, radius=10, fov=165, n_fov=20):
    """rr_camera from the reference's settings.  The hard-wired 6 / 10 / 165 / 20 are
    generator.py:267 and generator.py:179; exposure is settings["cam_exposure"] in ms
    (bad_weather.py:344)."""
    cam = rr_camera()
    cam.focal_m = float(focal_m)
    cam.focal_sq = float(focal_m) ** 2                      # bad_weather.py:468 `self.f ** 2`
    cam.f_number = float(f_number)
    cam.focus_plane = float(focus_plane)
    cam.exposure_s = exposure_ms / 1000.
    cam.radius = float(radius)
    cam.sensor_px = 4.65e-06
    drop_size = 1.16 * 1e-3
    cam.tau_zero = float(np.sqrt(drop_size) / 50)           # bad_weather.py:425
    theta = np.deg2rad(fov / 2)
    cam.fov_cos, cam.fov_sin = float(np.cos(-theta)), float(np.sin(-theta))
    phi = np.arange(0, 2 * np.pi, (2 * np.pi) / n_fov)      # bad_weather.py:630
    assert len(phi) == n_fov <= RR_MAX_FOV
    for k in range(n_fov):
        cam.phi_cos[k] = float(np.cos(phi[k]))
        cam.phi_sin[k] = float(np.sin(phi[k]))
    cam.n_fov = n_fov
    return cam


def filter_streaks(table, imW, imH):
    """The frame filter of Generator.run (generator.py:413-420) on a StreakTable."""
    m = max(imH, imW)
    s, e = table.ips, table.ipe
    inside_s = (0 <= s[:, 0]) & (s[:, 0] < imW) & (0 <= s[:, 1]) & (s[:, 1] < imH)
    inside_e = (0 <= e[:, 0]) & (e[:, 0] < imW) & (0 <= e[:, 1]) & (e[:, 1] < imH)
    keep = (1 <= table.max_width) & (table.max_width < m) & (1 <= table.length) & (table.length < m) & \
        (inside_s | inside_e)
    return np.nonzero(keep)[0]


def drop_draws(seed, tex_lo, is_big, noise_std):
    """(tex_index, raw normal deviates) of one frame's drop loop for np.random.seed(seed), from the
    library's host-side legacy-RandomState implementation (rr_host_drop_draws): thread-safe, leaves
    the process-global generator untouched."""
    lib = load_library()
    n = len(tex_lo)
    lo = np.ascontiguousarray(tex_lo, np.int32)
    big = np.ascontiguousarray(is_big, np.uint8)
    tex = np.empty(n, np.int32)
    noise = np.zeros(n, np.float64)
    rc = lib.rr_host_drop_draws(ctypes.c_uint32(int(seed) & 0xffffffff), n, _ptr(lo), _ptr(big), float(noise_std),
                                _ptr(tex), _ptr(noise))
    if rc != 0:
        raise RuntimeError("rr_host_drop_draws failed (%d)" % rc)
    return tex, noise


def pack_drops(table, idx, db, noise_std=0.0, noise_scale=0.0, seed=None):
    """rr_drop[] for the streaks table[idx] with the random draws of the reference's per-drop loop:
    one randint per drop (bad_weather.py:252-264), then one normal per non-Big drop (generator.py:136).
    seed=None consumes the legacy process-global RandomState like the reference does (the caller has
    called np.random.seed); an integer seed gives the identical draws from the library's own generator
    (rr_host_drop_draws) without touching global state, so frames can be packed on worker threads.
    Applies the in-place endpoint rotation of generator.py:152-161 to the table (persistent, like the
    reference)."""
    n = len(idx)
    out = np.zeros(n, DROP_DTYPE)
    if n == 0:
        return out
    bucket = db.texture_bucket(table.ratio[idx])
    types = table.type[idx]
    if seed is not None:
        if not 0 <= int(seed) <= 2 ** 32 - 1:
            raise ValueError("Seed must be between 0 and 2**32 - 1")
        tex, noise = drop_draws(seed, bucket * 10, types == 0, noise_std)
        noise = noise * noise_scale
    else:
        tex = np.empty(n, np.int32)
        noise = np.zeros(n)
        randint, normal = np.random.randint, np.random.normal
        lo = (bucket * 10).tolist()
        big = (types == 0).tolist()
        for k in range(n):
            tex[k] = randint(lo[k], lo[k] + 10)
            if not big[k]:
                noise[k] = normal(0.0, noise_std) * noise_scale
    s = table.ips[idx].astype(np.float64)
    e = table.ipe[idx].astype(np.float64)
    nb = types != 0
    with np.errstate(all='ignore'):
        d = s - e
        n1 = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
        theta = np.rad2deg(np.arccos((d[:, 0] / n1) * 0 + (d[:, 1] / n1) * -1))
        ang = -(theta + noise) * (np.pi / 180)
        rot_cos, rot_sin = np.cos(ang), np.sin(ang)
    if np.any(noise != 0):
        nx, ny = np.cos(np.deg2rad(noise)), np.sin(np.deg2rad(noise))
        mx = (e[:, 0] + s[:, 0]) / 2
        my = (e[:, 1] + s[:, 1]) / 2
        s2 = np.stack([(s[:, 0] - mx) * nx - (s[:, 1] - my) * ny + mx,
                       (s[:, 0] - mx) * ny + (s[:, 1] - my) * nx + my], axis=1).astype(np.int64)
        e2 = np.stack([(e[:, 0] - mx) * nx - (e[:, 1] - my) * ny + mx,
                       (e[:, 0] - mx) * ny + (e[:, 1] - my) * nx + my], axis=1).astype(np.int64)
        sel = idx[nb]
        table.ips[sel] = s2[nb]
        table.ipe[sel] = e2[nb]
    out['x0'] = table.ips[idx, 0]
    out['y0'] = table.ips[idx, 1]
    out['x1'] = table.ipe[idx, 0]
    out['y1'] = table.ipe[idx, 1]
    out['max_width'] = table.max_width[idx]
    out['length'] = table.length[idx]
    out['type'] = types
    out['tex_index'] = tex
    out['iw1'] = table.iw1[idx]
    out['iw2'] = table.iw2[idx]
    out['wps'] = table.wps[idx]
    out['wpe'] = table.wpe[idx]
    out['rot_cos'] = np.where(nb, rot_cos, 1.0)
    out['rot_sin'] = np.where(nb, rot_sin, 0.0)
    return out


def _table_view(table):
    """rr_streak_table over a StreakTable's columns (which must stay alive and C-contiguous)."""
    t = rr_streak_table()
    cols = dict(wps=np.float64, wpe=np.float64, ips=np.int64, ipe=np.int64, iw1=np.float64, iw2=np.float64, ratio=np.float64,
                max_width=np.int64, length=np.int64, type=np.int32)
    for k, dt in cols.items():
        a = getattr(table, k)
        assert a.dtype == dt and a.flags['C_CONTIGUOUS'], k
        setattr(t, k, a.ctypes.data)
    t.n = len(table)
    return t


def pack_frame(table, db, imW, imH, seed, noise_std=0.0, noise_scale=0.0, rotation='numpy'):
    """filter_streaks + pack_drops(seed=...) of one frame with the per-drop work in the library
    (rr_host_frame_draws / rr_host_assemble_drops: no interpreter lock held, so I/O threads scale); numpy keeps the
    rotation terms (acos / cos / sin).  Same records as pack_drops(table, filter_streaks(table, imW, imH), db, ..., seed);
    rotates the end points of `table` in place when angular noise is on, like the reference (generator.py:152-161).

    rotation='exact' (no angular noise only): the rotation terms the device-side packer writes (rr_particles.h
    derive_drop) -- cos(-theta) = -dy / n and sin(-theta) = -|dx| / n, what the reference's acos / rad2deg / cos chain
    (generator.py:138-145,163) evaluates to when carried out exactly, from + - * / sqrt alone: within an ulp or two of
    the numpy chain and the same on every machine."""
    lib = load_library()
    if not 0 <= int(seed) <= 2 ** 32 - 1:
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    n = len(table)
    tv = _table_view(table)
    keep = np.empty(n, np.int64)
    tex = np.empty(n, np.int32)
    noise = np.empty(n, np.float64)
    ratio_db = np.ascontiguousarray(db.ratio, np.float64)
    nk = lib.rr_host_frame_draws(ctypes.byref(tv), int(imW), int(imH), _ptr(ratio_db), len(ratio_db), ctypes.c_uint32(int(seed)),
                                 float(noise_std), _ptr(keep), _ptr(tex), _ptr(noise))
    if nk < 0:
        raise RuntimeError("rr_host_frame_draws failed (%d)" % nk)
    keep, tex, noise = keep[:nk], tex[:nk], noise[:nk] * noise_scale
    out = np.zeros(nk, DROP_DTYPE)
    if nk == 0:
        return out
    s = table.ips[keep].astype(np.float64)
    e = table.ipe[keep].astype(np.float64)
    with np.errstate(all='ignore'):
        d = s - e
        n1 = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
        if rotation == 'exact':
            if np.any(noise != 0):
                raise ValueError("rotation='exact' is defined without angular noise")
            rot_cos = (d[:, 0] / n1) * 0.0 + (d[:, 1] / n1) * -1.0
            rot_sin = -(np.abs(d[:, 0]) / n1)
        else:
            theta = np.rad2deg(np.arccos((d[:, 0] / n1) * 0 + (d[:, 1] / n1) * -1))
            ang = -(theta + noise) * (np.pi / 180)
            rot_cos, rot_sin = np.cos(ang), np.sin(ang)
    if np.any(noise != 0):
        nb = table.type[keep] != 0
        nx, ny = np.cos(np.deg2rad(noise)), np.sin(np.deg2rad(noise))
        mx = (e[:, 0] + s[:, 0]) / 2
        my = (e[:, 1] + s[:, 1]) / 2
        s2 = np.stack([(s[:, 0] - mx) * nx - (s[:, 1] - my) * ny + mx,
                       (s[:, 0] - mx) * ny + (s[:, 1] - my) * nx + my], axis=1).astype(np.int64)
        e2 = np.stack([(e[:, 0] - mx) * nx - (e[:, 1] - my) * ny + mx,
                       (e[:, 0] - mx) * ny + (e[:, 1] - my) * nx + my], axis=1).astype(np.int64)
        sel = keep[nb]
        table.ips[sel] = s2[nb]
        table.ipe[sel] = e2[nb]
    rc = lib.rr_host_assemble_drops(ctypes.byref(tv), nk, _ptr(keep), _ptr(tex), _ptr(rot_cos), _ptr(rot_sin), _ptr(out))
    if rc != 0:
        raise RuntimeError("rr_host_assemble_drops failed (%d)" % rc)
    return out


def table_rotation(table):
    """(rot_cos, rot_sin) of EVERY entry of a streak table for a run without angular noise: the reference's chain
    (generator.py:138-145,163: acos -> degrees -> radians -> cos / sin, evaluated by numpy exactly as pack_frame does for
    the streaks it keeps), once per simulated frame instead of once per rendered frame; cached on the table."""
    rot = getattr(table, '_rot', None)
    if rot is None:
        s = table.ips.astype(np.float64)
        e = table.ipe.astype(np.float64)
        with np.errstate(all='ignore'):
            d = s - e
            n1 = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
            theta = np.rad2deg(np.arccos((d[:, 0] / n1) * 0 + (d[:, 1] / n1) * -1))
            ang = -(theta + np.zeros(len(theta))) * (np.pi / 180)
            rot = (np.ascontiguousarray(np.cos(ang)), np.ascontiguousarray(np.sin(ang)))
        table._rot = rot
    return rot


def _c_paths(paths):
    """NULL-safe char*[] of file names (None entries stay NULL); the bytes objects must outlive the call."""
    enc = [None if p is None else os.fsencode(p) for p in paths]
    return (ctypes.c_char_p * len(enc))(*enc), enc


def pack_frames(tables, seeds, db, imW, imH, out_block, out_stride, cap, threads=0):
    """rr_host_pack_frames: frame k's drop table (no angular noise) from tables[k] with seed seeds[k], written to record
    k * out_stride of the rr_drop block at address / array `out_block`; returns the kept counts (int64 array).  Same
    records as pack_frame(tables[k], db, imW, imH, seeds[k]) -- tests/test_host_logic.py."""
    lib = load_library()
    n = len(tables)
    tvs, keep = [], []
    for t in tables:
        tv = getattr(t, '_tv', None)
        if tv is None:
            tv = t._tv = _table_view(t)
        tvs.append(tv)
        keep.append(table_rotation(t))
    tptr = (ctypes.c_void_p * n)(*[ctypes.addressof(tv) for tv in tvs])
    cptr = (ctypes.c_void_p * n)(*[r[0].ctypes.data for r in keep])
    sptr = (ctypes.c_void_p * n)(*[r[1].ctypes.data for r in keep])
    ratio_db = np.ascontiguousarray(db.ratio, np.float64)
    sd = np.ascontiguousarray(seeds, np.uint32)
    if any(not 0 <= int(v) <= 2 ** 32 - 1 for v in seeds):
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    counts = np.zeros(n, np.int64)
    base = out_block.ctypes.data if isinstance(out_block, np.ndarray) else int(out_block)
    rc = lib.rr_host_pack_frames(n, tptr, cptr, sptr, int(imW), int(imH), _ptr(ratio_db), len(ratio_db), _ptr(sd), ctypes.c_void_p(base),
                                 int(out_stride), int(cap), int(threads), _ptr(counts))
    if rc != 0 or (counts < 0).any():
        raise RuntimeError("rr_host_pack_frames failed (%d, %s)" % (rc, counts[counts < 0][:4]))
    return counts


def io_read_frames(image_paths, depth_paths, H, W, bg_block, depth_block, threads=0, depth_u16=False):
    """rr_io_read_frames into the frames-back-to-back blocks of RainHip.host_rows ((n, stride) uint8 arrays): frame k's
    image bytes (B G R) at bg_block[k], its depth as float32 metres at depth_block[k] -- or (depth_u16, rr_io_read_frames_u16)
    the depth file's uint16 samples as they are (rr_prepass_in.depth_f64 = RR_DEPTH_U16).  Returns the per-frame status."""
    lib = load_library()
    n = len(image_paths)
    ip, k1 = _c_paths(image_paths)
    dp, k2 = _c_paths(depth_paths)
    status = np.zeros(n, np.int32)
    assert bg_block.dtype == np.uint8 and depth_block.dtype == np.uint8 and bg_block.shape[0] >= n and depth_block.shape[0] >= n
    fn = lib.rr_io_read_frames_u16 if depth_u16 else lib.rr_io_read_frames
    rc = fn(n, ip, dp, int(H), int(W), _ptr(bg_block), int(bg_block.strides[0]), _ptr(depth_block),
            int(depth_block.strides[0]), int(threads), _ptr(status))
    if rc != 0:
        raise RuntimeError("rr_io_read_frames failed (%d)" % rc)
    return status


def io_read_frames_rows(image_paths, depth_paths, H, W, image_rows_block, depth_rows_block, threads=0):
    """rr_io_read_frames_rows: both files of every frame inflated only -- image_rows_block[k] receives the H rows of 1 + 3 W
    bytes (filter type + filtered R G B bytes) of frame k's image, depth_rows_block[k] the H rows of 1 + 2 W bytes of its
    16-bit depth file (rr_prepass_in: RR_IN_BG_PNG_ROWS / RR_DEPTH_PNG_ROWS; the filters are reversed on the device)."""
    lib = load_library()
    n = len(image_paths)
    ip, k1 = _c_paths(image_paths)
    dp, k2 = _c_paths(depth_paths)
    status = np.zeros(n, np.int32)
    assert image_rows_block.dtype == np.uint8 and depth_rows_block.dtype == np.uint8 and image_rows_block.shape[0] >= n and depth_rows_block.shape[0] >= n
    rc = lib.rr_io_read_frames_rows(n, ip, dp, int(H), int(W), _ptr(image_rows_block), int(image_rows_block.strides[0]), _ptr(depth_rows_block),
                                    int(depth_rows_block.strides[0]), int(threads), _ptr(status))
    if rc != 0:
        raise RuntimeError("rr_io_read_frames_rows failed (%d)" % rc)
    return status


def png_rows_of(samples):
    """An array of samples as PNG scanlines of filter type 0: (H, W, 3) uint8 B G R -> H rows of 1 + 3 W bytes (R G B);
    (H, W) uint16 -> H rows of 1 + 2 W bytes (big-endian).  What a frame that the row reader did not take is handed over as."""
    a = np.asarray(samples)
    H, W = a.shape[:2]
    if a.ndim == 3:
        body = np.ascontiguousarray(a[..., ::-1], np.uint8).reshape(H, 3 * W)
    else:
        body = np.ascontiguousarray(a, np.uint16).astype('>u2').view(np.uint8).reshape(H, 2 * W)
    return np.concatenate([np.zeros((H, 1), np.uint8), body], axis=1).reshape(-1)


def io_read_frames_scaled(image_paths, depth_paths, H, W, render_scale, depth_scale, bg_block, depth_block, threads=0):
    """rr_io_read_frames_scaled: like io_read_frames for a render scale other than 1 -- bg_block[k] receives the resized
    float64 image (H x W x 3, B G R, in [0, 1]), depth_block[k] the float32 depth map (H x W)."""
    lib = load_library()
    n = len(image_paths)
    ip, k1 = _c_paths(image_paths)
    dp, k2 = _c_paths(depth_paths)
    status = np.zeros(n, np.int32)
    assert bg_block.dtype == np.uint8 and depth_block.dtype == np.uint8 and bg_block.shape[0] >= n and depth_block.shape[0] >= n
    rc = lib.rr_io_read_frames_scaled(n, ip, dp, int(H), int(W), int(render_scale), int(depth_scale), _ptr(bg_block), int(bg_block.strides[0]),
                                      _ptr(depth_block), int(depth_block.strides[0]), int(threads), _ptr(status))
    if rc != 0:
        raise RuntimeError("rr_io_read_frames_scaled failed (%d)" % rc)
    return status


def io_write_frames(image_paths, mask_paths, rows_image_block, rows_mask_block, W, H, threads=0):
    """rr_io_write_frames from the scanline blocks of a pipeline slot ((n, stride) uint8 arrays).  Per-frame status."""
    lib = load_library()
    n = len(image_paths)
    ip, k1 = _c_paths(image_paths)
    mp, k2 = _c_paths(mask_paths)
    status = np.zeros(n, np.int32)
    assert rows_image_block.strides[0] == rows_mask_block.strides[0]
    rc = lib.rr_io_write_frames(n, ip, mp, _ptr(rows_image_block), _ptr(rows_mask_block), int(rows_image_block.strides[0]), int(W), int(H),
                                int(threads), _ptr(status))
    if rc != 0:
        raise RuntimeError("rr_io_write_frames failed (%d)" % rc)
    return status


def pack_streak_db(textures):
    """(texels uint8[], tex_h int32[], tex_w int32[], tex_off int64[]) for rr_set_streak_db."""
    hs = np.array([t.shape[0] for t in textures], np.int32)
    ws = np.array([t.shape[1] for t in textures], np.int32)
    sizes = hs.astype(np.int64) * ws
    padded = (sizes + 15) // 16 * 16                 # 16-byte aligned textures: the kernels copy them with dword loads
    offs = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.int64)
    texels = np.zeros(int(padded.sum()), np.uint8)
    for t, o, n in zip(textures, offs, sizes):
        texels[o:o + n] = np.ascontiguousarray(t, np.uint8).ravel()
    return texels, hs, ws, offs


_SHARED = {}


def shared_context(device=None):
    """One library context per (process, device) for the reference-signature single-frame seams
    (FogRain.fog_rain_layer, EnvironmentMapGenerator.generate_map)."""
    import os
    device = int(os.environ.get('LOCAL_RANK', '0')) if device is None else int(device)
    if device not in _SHARED:
        _SHARED[device] = RainHip(device)
    return _SHARED[device]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Prepared:
    """Descriptor arrays of one batch (RainHip.pipeline_prepare) and everything they point to."""

    def __init__(self, n, pin, fin, fout, pout, keep, frames, outs):
        self.n, self.pin, self.fin, self.fout, self.pout, self.keep, self.frames, self.outs = n, pin, fin, fout, pout, keep, frames, outs

    def set_drop_count(self, k, n_drops):
        """Frame k renders the first n_drops records of the drop array it was prepared with."""
        self.fin[k].n_drops = int(n_drops)


class RainHip:
    """One rendering context on one GPU (one per process / per GPU)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.rr_create(ctypes.byref(h), int(device))
        if rc != 0:
            raise RuntimeError("rr_create(device=%d) failed with %d: this path needs a gfx950 GPU "
                               "(no CPU fallback)" % (device, rc))
        self.h = h
        for item in filter(None, os.environ.get('RAINHIP_OPTIONS', '').split(',')):      # tuning switches for A/B runs: "9=0,4=5"
            k, v = item.split('=')
            self.set_option(int(k), int(v))
        self.device = device
        self._keep = []

    def bcast_selftest(self):
        """rr_bcast_selftest: the RCCL leg of rr_bcast_streak_db on this context's own device (one rank)."""
        self._check(self.lib.rr_bcast_selftest(self.h), 'rr_bcast_selftest')

    def close(self):
        if self.h:
            for s in range(RR_PIPE_SLOTS):
                self.lib.rr_pipeline_wait(self.h, s)
            self._inflight = {}
            for p in getattr(self, '_pinned', []):
                self.lib.rr_host_free(self.h, ctypes.c_void_p(p))
            self._pinned = []
            self.lib.rr_destroy(self.h)
            self.h = None

    # ---- pinned host memory + asynchronous pipeline -------------------------------------------
    def host_array(self, shape, dtype):
        """numpy array over page-locked memory (rr_host_alloc): what the asynchronous pipeline moves at PCIe rate.
        Lives until close()."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        ptr = ctypes.c_void_p()
        self._check(self.lib.rr_host_alloc(self.h, ctypes.byref(ptr), max(n, 1)), 'rr_host_alloc')
        if not hasattr(self, '_pinned'):
            self._pinned = []
        self._pinned.append(ptr.value)
        buf = (ctypes.c_uint8 * max(n, 1)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def host_rows(self, n, shape, dtype):
        """n arrays of `shape` / `dtype` laid out back to back in ONE page-locked block, each starting on a 16-byte
        boundary: (block, [views]).  The library's staging uses the same padding, so such a set of per-frame buffers
        crosses PCIe as a single copy (include/rainhip.h RR_OPT_COPY_KERNELS).  Free the block with host_free."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        stride = max((nbytes + 15) // 16 * 16, 16)
        raw = self.host_array((n, stride), np.uint8)
        return raw, [raw[k, :nbytes].view(dtype).reshape(shape) for k in range(n)]

    def host_free(self, array):
        """Give a host_array back (rr_host_free).  The array -- and every view of it -- must not be used afterwards and no
        batch that reads or writes it may be in flight."""
        addr = array.ctypes.data
        pinned = getattr(self, '_pinned', [])
        if addr not in pinned:
            raise ValueError("not a host_array of this context (or already freed)")
        pinned.remove(addr)
        self._check(self.lib.rr_host_free(self.h, ctypes.c_void_p(addr)), 'rr_host_free')

    def pipeline_prepare(self, frames, outs):
        """The descriptor arrays of one batch for rr_pipeline_submit, built once: frames as for pipeline_frames
        (dict(bg | bg_u8, depth, fog, omega, drops | sim, ...)), or as for render_frames (dict(bg, rainy_bg, env_xyY, omega,
        drops)) when they carry no 'depth'; outs: list of dict(image_u8[, mask][, mask_i32][, rainy_bg][, status]
        [, rainy_png][, mask_png][, n_drops]) of caller-owned arrays (ideally from host_array) that the library fills.
        omega None = the map given to set_solid_angles.  A driver that re-uses the same buffers batch after batch
        prepares once per slot and calls pipeline_submit_prepared (the per-frame Python work is what bounds a fast GPU);
        Prepared.set_drop_count(k, n) adjusts a frame's drop count in place."""
        n = len(frames)
        with_pre = 'depth' in frames[0] or 'depth_png_rows' in frames[0]
        pin = (rr_prepass_in * n)() if with_pre else None
        fin = (rr_frame_in * n)()
        fout = (rr_frame_out * n)()
        keep = []
        We = self._check(self.lib.rr_envmap_width(self.h), 'rr_envmap_width') if with_pre else 0
        for k, (fr, o) in enumerate(zip(frames, outs)):
            om = np.ascontiguousarray(fr['omega'], np.float64) if fr.get('omega') is not None else None
            sim = fr.get('sim')                        # SIM_FRAME_DTYPE record: the drop table is generated on the device;
            if sim is not None:                        # o['status'] (if any) must hold fr['drops_cap'] entries
                sim = np.ascontiguousarray(sim, SIM_FRAME_DTYPE).reshape(1)
                drops = np.zeros(0, DROP_DTYPE)
                cap = int(fr.get('drops_cap', 0)) or int(sim['n_particles'][0])
            else:
                drops = np.ascontiguousarray(fr['drops'], DROP_DTYPE)
                cap = len(drops)
            if with_pre:
                bg = self._fill_prepass(pin[k], fr, keep)
                H, W = bg.shape[:2]
                fin[k].H, fin[k].W, fin[k].He, fin[k].We = H, W, H, We
                fin[k].bg = None
            else:
                bg = np.ascontiguousarray(fr['bg'], np.float64)
                rb = np.ascontiguousarray(fr['rainy_bg'], np.float64)
                env = np.ascontiguousarray(fr['env_xyY'], np.float64)
                H, W = bg.shape[:2]
                fin[k].H, fin[k].W, fin[k].He, fin[k].We = H, W, env.shape[0], env.shape[1]
                fin[k].bg, fin[k].rainy_bg, fin[k].env_xyY = _ptr(bg), _ptr(rb), _ptr(env)
                keep.append((bg, rb, env))
            fin[k].omega = _ptr(om)
            fin[k].drops = _ptr(drops) if len(drops) else None
            fin[k].n_drops = cap
            if sim is not None:
                fin[k].sim = _ptr(sim)
                keep.append(sim)
                nd_out = o.get('n_drops')
                assert nd_out is None or (nd_out.dtype == np.int32 and nd_out.size >= 1)
                fout[k].n_drops_out = _ptr(nd_out)
            fin[k].strategy = int(fr.get('strategy', 0))
            fin[k].opacity_attenuation = float(fr.get('opacity_attenuation', 1.0))
            for name, dt in (('image_u8', np.uint8), ('mask', np.float64), ('mask_i32', np.int32), ('rainy_bg', np.float64),
                             ('status', np.int32), ('rainy_png', np.uint8), ('mask_png', np.uint8)):
                a = o.get(name)
                assert a is None or (a.dtype == dt and a.flags['C_CONTIGUOUS']), name
            assert o.get('image_u8') is None or o['image_u8'].shape == (H, W, 3)
            fout[k].rainy_rgb = _ptr(o.get('image_u8'))
            fout[k].rainy_bg_out = _ptr(o.get('rainy_bg'))
            fout[k].mask_f64 = _ptr(o.get('mask'))
            fout[k].mask_i32 = _ptr(o.get('mask_i32'))
            fout[k].drop_status = _ptr(o.get('status')) if cap else None
            assert o.get('status') is None or o['status'].size >= cap
            fout[k].rainy_png = _ptr(o.get('rainy_png'))
            fout[k].mask_png = _ptr(o.get('mask_png'))
            keep.append((om, drops))
        pout = None
        if with_pre and any(o.get('env_bgr_u8') is not None for o in outs):
            pout = (rr_prepass_out * n)()
            for k, o in enumerate(outs):
                e = o.get('env_bgr_u8')
                assert e is None or (e.dtype == np.uint8 and e.flags['C_CONTIGUOUS'])
                pout[k].env_bgr_u8 = _ptr(e)
        return Prepared(n, pin, fin, fout, pout, keep, frames, outs)

    def pipeline_submit_prepared(self, slot, prep, n=None):
        """rr_pipeline_submit of a prepared batch (its first n frames).  Nothing may be touched until pipeline_wait(slot)."""
        n = prep.n if n is None else int(n)
        assert 0 < n <= prep.n
        self._check(self.lib.rr_pipeline_submit(self.h, int(slot), n, prep.pin, prep.fin, prep.fout, prep.pout), 'rr_pipeline_submit')
        if not hasattr(self, '_inflight'):
            self._inflight = {}
        self._inflight[int(slot)] = prep

    def pipeline_submit(self, slot, frames, outs):
        """pipeline_prepare + pipeline_submit_prepared."""
        self.pipeline_submit_prepared(slot, self.pipeline_prepare(frames, outs))

    def pipeline_wait(self, slot):
        """True when the batch of `slot` is complete; False when the tile arena had to grow (submit the batch again)."""
        rc = self.lib.rr_pipeline_wait(self.h, int(slot))
        getattr(self, '_inflight', {}).pop(int(slot), None)
        if rc == RR_E_ARENA:
            return False
        self._check(rc, 'rr_pipeline_wait')
        return True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            # (rr_host_alloc may run beside the context's own calls: it reports through its own, lock-guarded message)
            msg = self.lib.rr_host_last_error(self.h) if what == 'rr_host_alloc' else self.lib.rr_last_error(self.h)
            raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode()))
        return rc

    def set_streak_db(self, textures):
        texels, hs, ws, offs = pack_streak_db(textures)
        self._db_meta = (hs, ws, offs, int(texels.size))
        self._check(self.lib.rr_set_streak_db(self.h, _ptr(texels), _ptr(hs), _ptr(ws), _ptr(offs), len(hs)),
                    'rr_set_streak_db')

    def set_streak_db_device(self, dev_ptr, n_bytes, hs, ws, offs):
        hs = np.ascontiguousarray(hs, np.int32)
        ws = np.ascontiguousarray(ws, np.int32)
        offs = np.ascontiguousarray(offs, np.int64)
        self._check(self.lib.rr_set_streak_db_device(self.h, ctypes.c_void_p(dev_ptr), int(n_bytes), _ptr(hs), _ptr(ws),
                                                     _ptr(offs), len(hs)), 'rr_set_streak_db_device')

    def set_solid_angles(self, omega):
        """The solid-angle map (common/solid_angle.get_solid_angles: a function of the environment map's shape alone), kept
        resident on the device: frames of that map size may pass omega=None."""
        om = np.ascontiguousarray(omega, np.float64)
        self._check(self.lib.rr_set_solid_angles(self.h, om.shape[0], om.shape[1], _ptr(om)), 'rr_set_solid_angles')

    def set_colormap(self, lut_rgba):
        """256x4 uint8 RGBA table of the colour map plt.imsave applies to the rain mask (common/imgops.viridis_lut())."""
        lut = np.ascontiguousarray(lut_rgba, np.uint8)
        assert lut.shape == (256, 4)
        self._check(self.lib.rr_set_colormap(self.h, _ptr(lut)), 'rr_set_colormap')

    # ---- drop tables born on the device (particle generator, BASELINE configs[4]) ---------------------------
    def set_particle_tables(self, d_grid, cdf):
        """Inverse-CDF tables of the drop diameter (tools/particles.diameter_tables): d_grid [n], cdf [n_tables, n]."""
        d = np.ascontiguousarray(d_grid, np.float64)
        c = np.ascontiguousarray(np.atleast_2d(cdf), np.float64)
        assert c.shape[1] == len(d)
        self._check(self.lib.rr_set_particle_tables(self.h, c.shape[0], len(d), _ptr(d), _ptr(c)), 'rr_set_particle_tables')

    def generate_drops_device(self, sims, H, W, drops_ptr, cap, n_out_ptr, stream=None):
        """rr_generate_drops_device: sims = SIM_FRAME_DTYPE records (host); drops_ptr / n_out_ptr = DEVICE addresses of
        len(sims) * cap rr_drop records / len(sims) int32 counts."""
        sims = np.ascontiguousarray(sims, SIM_FRAME_DTYPE)
        self._check(self.lib.rr_generate_drops_device(self.h, len(sims), _ptr(sims), int(H), int(W), ctypes.c_void_p(drops_ptr), int(cap),
                                                      ctypes.c_void_p(n_out_ptr), ctypes.c_void_p(stream) if stream else None),
                    'rr_generate_drops_device')

    def generate_drops(self, sims, H, W, cap=None):
        """rr_generate_drops: list of DROP_DTYPE arrays (one per frame, what the device generated and packed) and the
        drop counts (a count above `cap` means the frame's table was cut off at cap records)."""
        sims = np.ascontiguousarray(sims, SIM_FRAME_DTYPE)
        n = len(sims)
        cap = int(cap or max(int(sims['n_particles'].max()), 1))
        out = np.zeros((n, cap), DROP_DTYPE)
        cnt = np.zeros(n, np.int32)
        self._check(self.lib.rr_generate_drops(self.h, n, _ptr(sims), int(H), int(W), _ptr(out), cap, _ptr(cnt)), 'rr_generate_drops')
        return [out[k, :min(int(cnt[k]), cap)] for k in range(n)], cnt

    def set_option(self, option, value):
        """rr_set_option: tuning / A-B switches that never change a result bit (include/rainhip.h)."""
        self._check(self.lib.rr_set_option(self.h, int(option), int(value)), 'rr_set_option')

    def set_camera(self, cam):
        self.cam = cam
        self._check(self.lib.rr_set_camera(self.h, ctypes.byref(cam)), 'rr_set_camera')

    def render_frames(self, frames, want_composite=True, want_colour=False):
        """frames: list of dict(bg, rainy_bg, env_xyY, omega, drops[, opacity_attenuation]) with
        C-contiguous arrays and a DROP_DTYPE drop table.  The image arrays may be float32 or uint8 (uint8 = the bytes
        cv2.imread returned: value / 255.0), the map float32 (omega then travels as float32 too): rr_frame_in.in_types;
        anything else is taken as float64.  Returns a list of
        dict(image_u8 RGB, rainy_bg, mask, mask_i32, status[, colour = (n, 3) BGR colour constants])."""
        n = len(frames)
        fin = (rr_frame_in * n)()
        fout = (rr_frame_out * n)()
        outs = []
        keep = []
        for k, fr in enumerate(frames):
            def image(a):
                a = np.asarray(a)
                return np.ascontiguousarray(a, a.dtype if a.dtype in (np.float32, np.uint8) else np.float64)
            same = fr['rainy_bg'] is fr['bg']
            bg = image(fr['bg'])
            rb = bg if same else image(fr['rainy_bg'])
            env = np.asarray(fr['env_xyY'])
            env = np.ascontiguousarray(env, np.float32 if env.dtype == np.float32 else np.float64)
            om = None if fr.get('omega') is None else np.ascontiguousarray(fr['omega'], env.dtype)   # None: rr_set_solid_angles' map
            types = ({np.dtype(np.float32): RR_IN_BG_F32, np.dtype(np.uint8): RR_IN_BG_U8}.get(bg.dtype, 0) |
                     {np.dtype(np.float32): RR_IN_RAINY_F32, np.dtype(np.uint8): RR_IN_RAINY_U8}.get(rb.dtype, 0) |
                     (RR_IN_ENV_F32 if env.dtype == np.float32 else 0))
            sim = fr.get('sim')                        # one SIM_FRAME_DTYPE record: the drop table is generated on the device
            if sim is not None:
                sim = np.ascontiguousarray(sim, SIM_FRAME_DTYPE).reshape(1)
                cap = int(fr.get('drops_cap', 0)) or int(sim['n_particles'][0])
                drops = np.zeros(cap, DROP_DTYPE)      # only its length is used (output sizes)
            else:
                drops = np.ascontiguousarray(fr['drops'], DROP_DTYPE)
            H, W = bg.shape[:2]
            He, We = env.shape[:2]
            assert bg.shape == (H, W, 3) and rb.shape == (H, W, 3) and env.shape == (He, We, 3) and (om is None or om.shape == (He, We))
            o = dict(image_u8=np.zeros((H, W, 3), np.uint8),
                     rainy_bg=np.zeros((H, W, 3), np.float64) if want_composite else None,
                     mask=np.zeros((H, W), np.float64), mask_i32=np.zeros((H, W), np.int32),
                     status=np.zeros(len(drops), np.int32))
            if want_colour:
                o['colour'] = np.zeros((len(drops), 3), np.float64)
                fout[k].drop_colour = _ptr(o['colour']) if len(drops) else None
            fin[k].H, fin[k].W, fin[k].He, fin[k].We = H, W, He, We
            fin[k].bg, fin[k].rainy_bg, fin[k].env_xyY, fin[k].omega = _ptr(bg), _ptr(rb), _ptr(env), (_ptr(om) if om is not None else None)
            fin[k].in_types = types
            fin[k].drops = _ptr(drops) if len(drops) and sim is None else None
            fin[k].n_drops = len(drops)
            if sim is not None:
                fin[k].sim = _ptr(sim)
                o['n_drops'] = np.zeros(1, np.int32)
                fout[k].n_drops_out = _ptr(o['n_drops'])
                keep.append(sim)
            fin[k].strategy = int(fr.get('strategy', 0))
            fin[k].opacity_attenuation = float(fr.get('opacity_attenuation', 1.0))
            if fr.get('ext') is not None:              # caller-made tiles: [None | dict(alpha HxW, minC (x, y), poly Nx2 | None)] per drop
                ext = (rr_ext_tile * max(len(drops), 1))()
                for j, e in enumerate(fr['ext']):
                    if e is None:
                        continue
                    a = np.ascontiguousarray(e['alpha'], np.float64)
                    poly = np.ascontiguousarray(e['poly'] if e.get('poly') is not None else np.zeros((0, 2)), np.float64).reshape(-1, 2)
                    ext[j].alpha, ext[j].th, ext[j].tw = _ptr(a), a.shape[0], a.shape[1]
                    ext[j].min_x, ext[j].min_y = int(e['minC'][0]), int(e['minC'][1])
                    ext[j].n_poly, ext[j].poly_xy = len(poly), (_ptr(poly) if len(poly) else None)
                    keep.append((a, poly))
                fin[k].ext = ctypes.cast(ext, ctypes.c_void_p)
                keep.append(ext)
            if fr.get('depth') is not None:            # only read with RR_OPT_DEPTH_OCCLUSION
                dep = np.asarray(fr['depth'])
                dep = np.ascontiguousarray(dep, np.float32 if dep.dtype == np.float32 else np.float64)
                assert dep.shape == (H, W)
                fin[k].depth, fin[k].depth_f64 = _ptr(dep), 1 if dep.dtype == np.float64 else 0
                keep.append(dep)
            fout[k].rainy_rgb = _ptr(o['image_u8'])
            fout[k].rainy_bg_out = _ptr(o['rainy_bg'])
            fout[k].mask_f64 = _ptr(o['mask'])
            fout[k].mask_i32 = _ptr(o['mask_i32'])
            fout[k].drop_status = _ptr(o['status']) if len(drops) else None
            keep.append((bg, rb, env, om, drops))
            outs.append(o)
        self._check(self.lib.rr_render_frames(self.h, n, fin, fout), 'rr_render_frames')
        for o in outs:                                 # a generated drop table: trim the per-drop outputs to its drop count
            if 'n_drops' in o:
                nd = int(o['n_drops'][0])
                o['n_drops'] = nd
                o['status'] = o['status'][:nd]
                if 'colour' in o:
                    o['colour'] = o['colour'][:nd]
        return outs

    # ---- pre-pass (fog attenuation + environment map), SURVEY 8f next #1/#2 ------------------
    def set_prepass_kernels(self, fog_w, env_w):
        """Gaussian taps of the two blurs (cv::getGaussianKernel(25, 25) and (15, 0)), computed by the host."""
        k = rr_prepass_kernels()
        k.fog_ksize, k.env_ksize = len(fog_w), len(env_w)
        for i, v in enumerate(fog_w):
            k.fog_w[i] = float(v)
        for i, v in enumerate(env_w):
            k.env_w[i] = float(v)
        self._check(self.lib.rr_set_prepass_kernels(self.h, ctypes.byref(k)), 'rr_set_prepass_kernels')

    def set_envmap_geometry(self, H, W, cw, uniq, first):
        """Projection tables of EnvironmentMapGenerator for HxW frames; returns the map width We."""
        uniq = np.ascontiguousarray(uniq, np.int32)
        first = np.ascontiguousarray(first, np.int32)
        self._check(self.lib.rr_set_envmap_geometry(self.h, int(H), int(W), int(cw), len(uniq), _ptr(uniq), _ptr(first)),
                    'rr_set_envmap_geometry')
        return self._check(self.lib.rr_envmap_width(self.h), 'rr_envmap_width')

    @staticmethod
    def _fill_prepass(pin, fr, keep):
        """fr['bg']: the image / 255 as float64 (or float32: taken as it is, rr_prepass_in.in_types), or fr['bg_u8'] = the
        uint8 BGR image (bg = bg_u8 / 255.0 is formed on the device)."""
        if fr.get('bg_png_rows') is not None:              # H rows of 1 + 3 W bytes: an 8-bit RGB PNG inflated, not un-filtered; 'shape' = (H, W)
            rows = np.ascontiguousarray(fr['bg_png_rows'], np.uint8).reshape(-1)
            H, W = fr['shape']
            assert rows.size == H * (1 + 3 * W), (rows.size, H, W)
            pin.bg, pin.bg_u8, pin.in_types = _ptr(rows), None, RR_IN_BG_PNG_ROWS
            bg = np.empty((H, W, 3), np.uint8)              # (only its shape is used by the callers)
            keep.append(rows)
        elif fr.get('bg_u8') is not None:
            bg = np.ascontiguousarray(fr['bg_u8'], np.uint8)
            pin.bg, pin.bg_u8, pin.in_types = None, _ptr(bg), 0
        else:
            bg = np.asarray(fr['bg'])
            bg = np.ascontiguousarray(bg, np.float32 if bg.dtype == np.float32 else np.float64)
            pin.bg, pin.bg_u8, pin.in_types = _ptr(bg), None, RR_IN_BG_F32 if bg.dtype == np.float32 else 0
        H, W = bg.shape[:2]
        if fr.get('depth_png_rows') is not None:            # H rows of 1 + 2 W bytes: the 16-bit depth PNG inflated, not un-filtered
            depth = np.ascontiguousarray(fr['depth_png_rows'], np.uint8).reshape(-1)
            assert depth.size == H * (1 + 2 * W), (depth.size, H, W)
            pin.H, pin.W, pin.depth, pin.depth_f64 = H, W, _ptr(depth), RR_DEPTH_PNG_ROWS
        else:
            depth = np.asarray(fr['depth'])                    # float32 / float64 metres, or the uint16 samples of the depth file
            depth = np.ascontiguousarray(depth, depth.dtype if depth.dtype in (np.float32, np.uint16) else np.float64)
            assert bg.shape == (H, W, 3) and depth.shape == (H, W), (bg.shape, depth.shape)
            pin.H, pin.W, pin.depth = H, W, _ptr(depth)
            pin.depth_f64 = 1 if depth.dtype == np.float64 else (RR_DEPTH_U16 if depth.dtype == np.uint16 else 0)
        pin.beta_ext, pin.beta_hg, pin.irr_num, pin.irr_den = [float(v) for v in fr['fog']]
        keep.append((bg, depth))
        return bg

    def prepass_frames(self, frames, want_env=True, want_env_u8=False, out_dtype=np.float64):
        """frames: list of dict(bg | bg_u8, depth, fog=(beta_ext, beta_hg, irr_num, irr_den)).  Returns a list of
        dict(rainy_bg[, env_xyY][, env_bgr_u8]); out_dtype float32: rr_prepass_out.out_types (the float64 results rounded once)."""
        n = len(frames)
        pin = (rr_prepass_in * n)()
        pout = (rr_prepass_out * n)()
        keep, outs = [], []
        out_dtype = np.dtype(out_dtype)
        assert out_dtype in (np.dtype(np.float32), np.dtype(np.float64))
        We = self._check(self.lib.rr_envmap_width(self.h), 'rr_envmap_width') if (want_env or want_env_u8) else 0
        for k, fr in enumerate(frames):
            bg = self._fill_prepass(pin[k], fr, keep)
            H, W = bg.shape[:2]
            o = dict(rainy_bg=np.zeros((H, W, 3), out_dtype))
            if want_env:
                o['env_xyY'] = np.zeros((H, We, 3), out_dtype)
            if want_env_u8:
                o['env_bgr_u8'] = np.zeros((H, We, 3), np.uint8)
            pout[k].rainy_bg = _ptr(o['rainy_bg'])
            pout[k].env_xyY = _ptr(o.get('env_xyY'))
            pout[k].env_bgr_u8 = _ptr(o.get('env_bgr_u8'))
            pout[k].out_types = (RR_OUT_RAINY_F32 | RR_OUT_ENV_F32) if out_dtype == np.float32 else 0
            outs.append(o)
        self._check(self.lib.rr_prepass_frames(self.h, n, pin, pout), 'rr_prepass_frames')
        return outs

    def env_maps(self, backgrounds, want_xyY=False):
        """EnvironmentMapGenerator.generate_map alone (bad_weather.py:742-819) for float BGR images in [0,1] that already
        carry the fog layer: list of uint8 BGR maps (what --save_envmap stores), or (map, xyY) pairs."""
        n = len(backgrounds)
        pin = (rr_prepass_in * n)()
        pout = (rr_prepass_out * n)()
        We = self._check(self.lib.rr_envmap_width(self.h), 'rr_envmap_width')
        keep, outs = [], []
        for k, b in enumerate(backgrounds):
            bg = np.ascontiguousarray(b, np.float64)
            H, W = bg.shape[:2]
            assert bg.shape == (H, W, 3), bg.shape
            u8 = np.zeros((H, We, 3), np.uint8)
            xyY = np.zeros((H, We, 3), np.float64) if want_xyY else None
            pin[k].H, pin[k].W, pin[k].bg, pin[k].mode = H, W, _ptr(bg), RR_PRE_ENV_ONLY
            pout[k].env_bgr_u8, pout[k].env_xyY = _ptr(u8), _ptr(xyY)
            keep.append(bg)
            outs.append((u8, xyY) if want_xyY else u8)
        self._check(self.lib.rr_prepass_frames(self.h, n, pin, pout), 'rr_prepass_frames')
        return outs

    def pipeline_frames(self, frames, want_composite=False, want_rainy_bg=False, want_env_u8=False, want_mask_i32=True,
                        fog_dtype=np.float64):
        """Pre-pass + hot path without a host round trip.  frames: list of dict(bg, depth, fog, omega, drops
        [, opacity_attenuation, strategy]); omega None = the resident solid angles (set_solid_angles).  Returns what
        render_frames returns (+ env_bgr_u8 / fog_bg, the fog layer as fog_dtype: its width on the device follows)."""
        n = len(frames)
        pin = (rr_prepass_in * n)()
        pout = (rr_prepass_out * n)()
        fin = (rr_frame_in * n)()
        fout = (rr_frame_out * n)()
        keep, outs = [], []
        We = self._check(self.lib.rr_envmap_width(self.h), 'rr_envmap_width')
        for k, fr in enumerate(frames):
            bg = self._fill_prepass(pin[k], fr, keep)
            H, W = bg.shape[:2]
            om = None if fr.get('omega') is None else np.ascontiguousarray(fr['omega'], np.float64)
            assert om is None or om.shape == (H, We), (om.shape, (H, We))
            drops = np.ascontiguousarray(fr['drops'], DROP_DTYPE)
            o = dict(image_u8=np.zeros((H, W, 3), np.uint8),
                     rainy_bg=np.zeros((H, W, 3), np.float64) if want_composite else None,
                     mask=np.zeros((H, W), np.float64), mask_i32=np.zeros((H, W), np.int32) if want_mask_i32 else None,
                     status=np.zeros(len(drops), np.int32),
                     fog_bg=np.zeros((H, W, 3), fog_dtype) if want_rainy_bg else None,
                     env_bgr_u8=np.zeros((H, We, 3), np.uint8) if want_env_u8 else None)
            fin[k].H, fin[k].W, fin[k].He, fin[k].We = H, W, H, We
            fin[k].bg, fin[k].omega = None, _ptr(om)           # the background comes from the pre-pass input
            fin[k].drops = _ptr(drops) if len(drops) else None
            fin[k].n_drops = len(drops)
            fin[k].strategy = int(fr.get('strategy', 0))
            fin[k].opacity_attenuation = float(fr.get('opacity_attenuation', 1.0))
            fout[k].rainy_rgb = _ptr(o['image_u8'])
            fout[k].rainy_bg_out = _ptr(o['rainy_bg'])
            fout[k].mask_f64 = _ptr(o['mask'])
            fout[k].mask_i32 = _ptr(o['mask_i32'])
            fout[k].drop_status = _ptr(o['status']) if len(drops) else None
            pout[k].rainy_bg = _ptr(o['fog_bg'])
            pout[k].out_types = RR_OUT_RAINY_F32 if np.dtype(fog_dtype) == np.float32 else 0
            pout[k].env_xyY = None
            pout[k].env_bgr_u8 = _ptr(o['env_bgr_u8'])
            keep.append((om, drops))
            outs.append(o)
        self._check(self.lib.rr_pipeline_frames(self.h, n, pin, fin, fout, pout), 'rr_pipeline_frames')
        return outs

    # ---- device-resident path (bench / multi-frame pipelines) -------------------------------
    def render_frames_device(self, fin, fout, n, stream=None):
        """fin/fout: ctypes arrays of rr_frame_in/out holding DEVICE pointers."""
        rc = self.lib.rr_render_frames_device(self.h, n, fin, fout, ctypes.c_void_p(stream) if stream else None)
        self._check(rc, 'rr_render_frames_device')

    def synchronize(self):
        """Returns True if the batch completed, False if the tile arena had to be regrown
        (re-enqueue the batch)."""
        rc = self.lib.rr_synchronize(self.h)
        if rc == RR_E_ARENA:
            return False
        self._check(rc, 'rr_synchronize')
        return True

    def batch_counts(self, frame):
        """Work-list sizes of one frame of the last batch (see rr_batch_counts)."""
        out = (ctypes.c_int32 * 8)()
        self._check(self.lib.rr_batch_counts(self.h, int(frame), out), 'rr_batch_counts')
        return list(out)

    def profile(self, on):
        self.lib.rr_profile_enable(self.h, 1 if on else 0)

    def profile_reset(self):
        self.lib.rr_profile_reset(self.h)

    def profile_read(self):
        buf = (rr_kernel_stat * 32)()
        n = self._check(self.lib.rr_profile_read(self.h, buf, 32), 'rr_profile_read')
        return {buf[i].name.decode(): (buf[i].launches, buf[i].total_ms) for i in range(n)}


def bcast_streak_db(contexts):
    """rr_bcast_streak_db: the streak database of contexts[0] (RainHip objects of ONE process) becomes the database of the
    others -- a device-to-device copy on the root's device, ncclBroadcast (RCCL) across devices."""
    arr = (ctypes.c_void_p * len(contexts))(*[c.h for c in contexts])
    contexts[0]._check(contexts[0].lib.rr_bcast_streak_db(arr, len(contexts)), 'rr_bcast_streak_db')
    for c in contexts[1:]:
        c._db_meta = getattr(contexts[0], '_db_meta', None)
