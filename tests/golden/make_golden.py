"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (read-only, /root/reference) in
the build container and recording inputs + outputs of its own functions on seeded inputs.
The reference never travels to the GPU box; only these vectors do.

    python tests/golden/make_golden.py            # rewrites the fixtures

What the import needs: cv2, imutils, pyclipper and natsort are not
installed here, and numpy >= 1.24 dropped np.int / np.float / np.bool.  The shims below:

  * natsort.natsorted      -> natural-key sort (same ordering rule)
  * cv2.copyMakeBorder     -> np.pad(mode='constant')   (definitionally identical)
  * cv2.imread / cvtColor  -> PIL decode / channel replicate (I/O only)
  * pyclipper.Pyclipper    -> pass-through of the integer-truncated clip path        } NOT the real
  * cv2.fillConvexPoly     -> oracle.cvlike.fill_fov_mask (OpenCV's algorithm RESTATED) } library: see below
  * imutils                -> empty module (never reached by the functions called here)

Functions exercised through the reference's own code (=> these pin the oracle):
  DBManager.load_streaks_from_xml, classify_drop, load_streak_database (uniform-size DB),
  take_drop_texture (bucket + RNG order), RainRenderer.warping_points, compute_circle,
  circle_of_confusion (real scipy gaussian_filter), FovComputation.compute_fov_plane_points,
  my_utils.convert_rgb_to_xyY / convert_xyY_to_rgb, solid_angle.get_solid_angles, and the whole
  body of RainRenderer.add_drop_to_image (colour transform, defocus, placement, blend, mask
  accumulate) -- with the FOV *mask* produced by the two shims marked above, so the mask
  rasterisation itself stays UNPINNED while every operation around it is pinned.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def install_shims():
    import re
    from oracle import cvlike
    np.int = int
    np.float = float
    natsort = types.ModuleType('natsort')

    def natsorted(seq):
        return sorted(seq, key=lambda s: [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', str(s))])

    natsort.natsorted = natsorted
    cv2 = types.ModuleType('cv2')
    cv2.BORDER_CONSTANT = 0
    cv2.IMREAD_ANYDEPTH = 2
    cv2.COLOR_GRAY2BGR = 8
    cv2.INTER_CUBIC = 2
    cv2.INTER_AREA = 3

    def copyMakeBorder(src, top, bottom, left, right, borderType, value=None):
        pad = ((top, bottom), (left, right)) + ((0, 0),) * (src.ndim - 2)
        return np.pad(src, pad, mode='constant')

    def imread(path, flags=None):
        from PIL import Image
        return np.array(Image.open(path))

    def cvtColor(img, code):
        return np.dstack([img, img, img])

    def fillConvexPoly(img, pts, color):
        P = np.asarray(pts).reshape(-1, 2)
        if len(P) > 1 and np.array_equal(P[0], P[-1]):      # the reference closes the path by repeating its first vertex (bad_weather.py:373);
            P = P[:-1]                                        # fill_fov_mask takes the open polygon and closes it the same way
        cvlike.fill_fov_mask(img, P)                         # round 6: OpenCV's algorithm as restated in cvlike (cv_fill_convex_poly), the default rule
        return img

    cv2.copyMakeBorder = copyMakeBorder
    cv2.imread = imread
    cv2.cvtColor = cvtColor
    cv2.fillConvexPoly = fillConvexPoly
    pyclipper = types.ModuleType('pyclipper')
    pyclipper.PT_CLIP, pyclipper.PT_SUBJECT, pyclipper.CT_INTERSECTION, pyclipper.PFT_NONZERO = 1, 0, 0, 1

    class Pyclipper:
        def __init__(self):
            self.clip = None

        def AddPath(self, path, poly_type, closed=True):
            if poly_type == pyclipper.PT_CLIP:
                if len(path) == 0:
                    raise Exception("All paths are invalid for clipping")
                arr = np.asarray(path, np.float64)
                if not np.all(np.isfinite(arr)):
                    raise Exception("Coordinate outside allowed range")
                self.clip = cvlike.polygon_to_int(arr)

        def Execute(self, *a):
            rows_cols_touch = True
            return [self.clip.tolist()] if rows_cols_touch else []

    pyclipper.Pyclipper = Pyclipper
    imutils = types.ModuleType('imutils')
    for name, mod in (('natsort', natsort), ('cv2', cv2), ('pyclipper', pyclipper), ('imutils', imutils)):
        sys.modules[name] = mod
    sys.path.insert(0, REF)


def main():
    install_shims()
    os.chdir(REF)                      # common.db imports config.<dataset> relative to the repo root
    import matplotlib
    matplotlib.use('Agg')
    from common import bad_weather as rbw, my_utils as rmu, solid_angle as rsa
    import helpers as h
    from oracle import render as orc

    out = {}
    tmp = tempfile.mkdtemp()
    H, W = 96, 160

    # ---- 1. particles XML -> derived streak fields ------------------------------------------------
    frames = h.synthetic.simulate_particles(2, 40, W, H, seed0=4100)
    xml = h.synthetic.write_particles_xml(os.path.join(tmp, 'p', 'x_camera0.xml'), frames)
    with open(xml) as fh:
        out['xml_text'] = np.array(fh.read())
    for rs in (1, 2):
        db = rbw.DBManager(streaks_path_xml=xml)
        db.load_streaks_from_xml('kitti', {"render_scale": rs}, [W // rs, H // rs], use_pickle=False, verbose=False)
        rows = []
        for fid, fr in db.streaks_simulator.items():
            for pid, s in fr.streaks.items():
                rows.append([fid, pid, *s.world_position_start, *s.world_position_end, s.world_diameter_start,
                             s.world_diameter_end, *s.image_position_start, *s.image_position_end, s.image_diameter_start,
                             s.image_diameter_end, s.ratio, s.max_width, s.length, s.drop_type.value])
        out['xml_rs%d' % rs] = np.array(rows, np.float64)
    out['classify_w'] = np.arange(0, 9)
    out['classify_t'] = np.array([rbw.DBManager.classify_drop(w).value for w in range(9)])

    # ---- 2. streak DB loader (uniform texture size so that np.array(tmp) is legal) ----------------
    from PIL import Image
    tdir = os.path.join(tmp, 'db', 'size32')
    os.makedirs(tdir)
    rng = np.random.RandomState(5)
    names = []
    with open(os.path.join(tmp, 'db', 'norm.txt'), 'w') as fh:
        for cv in (0, 1, 10):
            fh.write('cv%d\n' % cv)
            fh.write(''.join('%.5f ' % v for v in rng.uniform(0.3, 1.0, 3)) + '\n')
            for osc in range(3):
                img = rng.randint(0, 65536, (12, 6)).astype(np.uint16)
                Image.fromarray(img).save(os.path.join(tdir, 'cv%d_osc%d.png' % (cv, osc)))
                names.append('cv%d_osc%d.png' % (cv, osc))
    db = rbw.DBManager(streaks_path=tdir, norm_coeff_path=os.path.join(tmp, 'db', 'norm.txt'))
    db.load_streak_database()
    out['db_norm_text'] = np.array(open(os.path.join(tmp, 'db', 'norm.txt')).read())
    out['db_names'] = np.array(sorted(os.listdir(tdir)))
    out['db_raw'] = np.stack([np.array(Image.open(os.path.join(tdir, n))) for n in sorted(os.listdir(tdir))])
    out['db_textures'] = np.asarray(db.streaks_light)[..., 0]
    out['db_ratio'] = np.asarray(db.ratio)

    # ---- 3. take_drop_texture: bucket logic + RNG stream ------------------------------------------
    db = rbw.DBManager()
    db.streaks_light = [np.full((2, 2, 3), k, np.uint8) for k in range(50)]
    db.ratio = np.array([0.1, 0.14, 0.2, 0.28, 0.4])
    ratios = np.array([0.05, 0.1, 0.12, 0.14, 0.19, 0.2, 0.27, 0.28, 0.39, 0.4, 0.7, np.nan, 0.13, 0.01])
    np.random.seed(123)
    picks = []
    for r in ratios:
        s = rbw.Streak()
        s.ratio = r
        tex = db.take_drop_texture(s)
        picks.append(int(round(tex[0, 0, 0] * 255)))
        picks.append(float(np.random.normal(0.0, 0.0)))       # interleaved second draw (generator.py:136)
    out['tex_ratios'] = ratios
    out['tex_picks'] = np.array(picks[0::2])

    # ---- 4. warping_points / compute_circle -------------------------------------------------------
    wp_in, wp_out = [], []
    rng = np.random.RandomState(6)
    for k in range(40):
        s = rbw.Streak()
        s.image_position_start = rng.randint(-20, 180, 2)
        s.image_position_end = s.image_position_start + rng.randint(-15, 60, 2)
        s.image_diameter_start, s.image_diameter_end = rng.uniform(4, 12, 2)
        tex = np.zeros((int(rng.choice([80, 114, 160, 320])), 32, 3))
        p1, p2, maxC, minC = rbw.RainRenderer.warping_points(s, tex, W, H)
        wp_in.append([*s.image_position_start, *s.image_position_end, s.image_diameter_start, s.image_diameter_end,
                      tex.shape[0], tex.shape[1]])
        wp_out.append(np.concatenate([p1.ravel(), p2.ravel(), maxC.astype(float), minC.astype(float)]))
    out['wp_in'] = np.array(wp_in)
    out['wp_out'] = np.array(wp_out)
    rr = rbw.RainRenderer(focal=0.006, f_number=6.0, focus_plane=6, radius=10, fov=165)
    zs = np.array([0.2, 0.3, 0.5, 1.0, 2.0, 4.0, 5.9, 6.0, 6.1, 8.0, 9.5, 14.0])
    out['coc_z'] = zs
    out['coc_c'] = np.array([rr.compute_circle(z) for z in zs])

    # ---- 5. circle_of_confusion with the real scipy filter -----------------------------------------
    rng = np.random.RandomState(8)
    for k, z in enumerate([0.3, 0.7, 1.5, 3.0, 7.0]):
        tile = rng.rand(7 + k, 4 + k, 4)
        blurred, shift = rr.circle_of_confusion(tile.copy(), z, None)
        out['coc_tile_%d' % k] = tile
        out['coc_blur_%d' % k] = blurred
        out['coc_shift_%d' % k] = np.array([shift, z])

    # ---- 6. FOV polygons ---------------------------------------------------------------------------
    fov = rbw.FovComputation(camera=np.array([0, 0, 0]))
    rng = np.random.RandomState(9)
    fov_in, fov_pts, fov_n = [], [], []
    env_shape = (96, 489, 3)
    for k in range(60):
        depth = rng.choice([0.4, 1.0, 3.0, 7.0, 9.9, 10.05, 10.5, 13.0])
        x, y = rng.uniform(-0.45, 0.45) * depth, rng.uniform(-0.35, 0.35) * depth
        s = rbw.Streak()
        s.world_position_start = np.array([x, y, depth])
        s.world_position_end = np.array([x + 0.002, y - 0.02, depth - 0.01])
        pts, _, _, _ = fov.compute_fov_plane_points(s, 10, 165, 20, env_shape)
        fov_in.append(np.concatenate([s.world_position_start, s.world_position_end]))
        padded = np.full((24, 2), np.nan)
        if len(pts):
            padded[:len(pts)] = pts
        fov_pts.append(padded)
        fov_n.append(len(pts))
    out['fov_in'] = np.array(fov_in)
    out['fov_pts'] = np.array(fov_pts)
    out['fov_n'] = np.array(fov_n)
    out['fov_env_shape'] = np.array(env_shape)

    # ---- 7. colour conversions, solid angles -------------------------------------------------------
    rgb = np.random.RandomState(10).rand(5, 7, 3)
    rgb[0, 0] = 0
    xyY = rmu.convert_rgb_to_xyY(rgb)
    out['col_rgb'] = rgb
    out['col_xyY'] = xyY
    ok = xyY.copy()
    ok[np.isnan(ok)] = 0.3
    out['col_back_in'] = ok
    out['col_back'] = rmu.convert_xyY_to_rgb(ok)
    out['omega_12x25'] = rsa.get_solid_angles(np.zeros((12, 25, 3)))

    # ---- 8. add_drop_to_image end to end (tiles from the oracle's own tile maker) ------------------
    sc = h.Scene(os.path.join(tmp, 'scene'), H, W, 60, seed0=4200, far_fraction=0.1)
    bg, env = sc.frame_inputs(0)
    textures, ratio = sc.oracle_db()
    streaks = sc.oracle_streaks(0)
    np.random.seed(0)
    rainy_bg = np.clip(bg * 0.9 + 0.05, 0, 1)
    rainy_mask = np.zeros((H, W))
    sat = np.zeros((H, W, 3))
    statuses = []
    tiles_used = []
    scipy_filter = rbw.gaussian_filter
    for variant in ('scipy', 'detexp'):
        if variant == 'detexp':
            # same reference code, but with the defocus filter swapped for the oracle's deterministic
            # one: every other operation must then agree BIT FOR BIT with the oracle
            rbw.gaussian_filter = lambda a, sig: orc.gaussian_filter_2d(a, sig[0], sig[1])
        np.random.seed(0)
        rb = rainy_bg.copy()
        rm = rainy_mask.copy()
        st = []
        for i, d in enumerate(streaks):
            tex_idx = orc.take_drop_texture_index(d, ratio)
            if d.drop_type != orc.DropType.Big:
                np.random.normal(0.0, 0.0)
            import copy
            dd = copy.deepcopy(d)
            tile, minC = orc.make_drop_tile(dd, textures[tex_idx], 0.0, W, H)
            rs = rbw.Streak()
            rs.world_position_start, rs.world_position_end = dd.world_position_start, dd.world_position_end
            rs.image_diameter_start, rs.image_diameter_end = dd.image_diameter_start, dd.image_diameter_end
            rs.length = dd.length
            pts, _, _, _ = fov.compute_fov_plane_points(rs, 10, 165, 20, env.shape)
            try:
                rr.add_drop_to_image('kitti', env, sc.omega, pts, minC, bg, rb, rm, sat, tile.copy(), rs, 'ambient', None, 1.0)
                st.append(0)
            except Exception:
                st.append(1)
        out['add_%s_rainy_bg' % variant] = rb
        out['add_%s_mask' % variant] = rm
        out['add_%s_skipped' % variant] = np.array(st)
    # the int32 export north_star grades (SURVEY decision D1), from the UNTOUCHED reference (real scipy filter)
    out['add_scipy_mask_i32'] = np.floor(out['add_scipy_mask'] * 255).astype(np.int32)
    # ---- 8c. a larger scene through the untouched reference (real scipy.ndimage.gaussian_filter): only what the
    # contract grades is stored -- floor(mask * 255) and the uint8 image of the epilogue (generator.py:461-466)
    rbw.gaussian_filter = scipy_filter
    BH, BW, BN, BSEED = 256, 384, 700, 4300
    scb = h.Scene(os.path.join(tmp, 'scene_big'), BH, BW, BN, seed0=BSEED, far_fraction=0.05)
    bgb, envb = scb.frame_inputs(0)
    texb, ratiob = scb.oracle_db()
    np.random.seed(0)
    rb = bgb.copy()
    rm = np.zeros((BH, BW))
    satb = np.zeros((BH, BW, 3))
    st = []
    import copy
    for d in scb.oracle_streaks(0):
        tex_idx = orc.take_drop_texture_index(d, ratiob)
        if d.drop_type != orc.DropType.Big:
            np.random.normal(0.0, 0.0)
        dd = copy.deepcopy(d)
        tile, minC = orc.make_drop_tile(dd, texb[tex_idx], 0.0, BW, BH)
        rs = rbw.Streak()
        rs.world_position_start, rs.world_position_end = dd.world_position_start, dd.world_position_end
        rs.image_diameter_start, rs.image_diameter_end = dd.image_diameter_start, dd.image_diameter_end
        rs.length = dd.length
        pts, _, _, _ = fov.compute_fov_plane_points(rs, 10, 165, 20, envb.shape)
        try:
            rr.add_drop_to_image('kitti', envb, scb.omega, pts, minC, bgb, rb, rm, satb, tile.copy(), rs, 'ambient', None, 1.0)
            st.append(0)
        except Exception:
            st.append(1)
    import io as _io
    import matplotlib.pyplot as _plt
    from PIL import Image as _PILImage
    final = rb - (np.mean(rb) - np.mean(bgb))                                  # generator.py:461-462
    buf = _io.BytesIO()
    _plt.imsave(buf, np.clip(final[..., ::-1], 0, 1))                            # generator.py:466
    buf.seek(0)
    out['big_scene'] = np.array([BH, BW, BN, BSEED])
    out['big_scipy_mask_i32'] = np.floor(rm * 255).astype(np.int32)
    out['big_scipy_image_u8'] = np.array(_PILImage.open(buf))[..., :3]
    out['big_scipy_skipped'] = np.array(st)
    # ---- 8b. rendering_strategy 'white' (bad_weather.py:349-353): no cv2/pyclipper call at all ----
    np.random.seed(0)
    rb = rainy_bg.copy()
    rm = rainy_mask.copy()
    for i, d in enumerate(streaks):
        tex_idx = orc.take_drop_texture_index(d, ratio)
        if d.drop_type != orc.DropType.Big:
            np.random.normal(0.0, 0.0)
        import copy
        dd = copy.deepcopy(d)
        tile, minC = orc.make_drop_tile(dd, textures[tex_idx], 0.0, W, H)
        rs = rbw.Streak()
        rs.image_diameter_start, rs.image_diameter_end = dd.image_diameter_start, dd.image_diameter_end
        rs.length = dd.length
        rr.add_drop_to_image('kitti', env, sc.omega, np.array([]), minC, bg, rb, rm, sat, tile.copy(), rs, 'ambient', 'white', 1.0)
    out['add_white_rainy_bg'] = rb
    out['add_white_mask'] = rm
    out['add_scene'] = np.array([H, W, 60, 4200])
    out['add_rainy_bg_in'] = rainy_bg

    # ---- 8d. DropDepthMap (dead in the reference: constructed behind USE_DEPTH_WEIGHTING = 0 only) -----
    from common import drop_depth_map as rddm
    calib = os.path.join(tmp, 'calib_cam_to_cam.txt')
    with open(calib, 'w') as fh:                       # KITTI raw calib lines the class parses (values of 2011_09_26)
        fh.write('calib_time: 09-Jan-2012 13:57:47\n'
                 'R_rect_02: 9.998817e-01 1.511453e-02 -2.841595e-03 -1.511724e-02 9.998853e-01 -9.338510e-04 2.827154e-03 9.766976e-04 9.999955e-01\n'
                 'P_rect_02: 7.215377e+02 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 7.215377e+02 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 1.000000e+00 2.745884e-03\n')
    dmap = np.random.RandomState(12).uniform(2.0, 60.0, (352, 1216))
    ev = rddm.DropDepthMap(filename=calib)
    xyz = ev.get_world_points(dmap)
    starts = np.random.RandomState(13).uniform(-3, 3, (4, 3))
    dd = rddm.DropDepthMap.depth_map_drop(starts, xyz)
    out['ddm_calib'] = np.array(open(calib).read())
    out['ddm_depth_seed'] = np.array([12, 13])
    out['ddm_xyz_sub'] = xyz[::37, ::53].copy()
    out['ddm_dist_sub'] = dd[:, ::37, ::53].copy()
    out['ddm_cam_pos'] = ev.camera_pos_world.copy()

    # ---- 9. matplotlib's float -> uint8 rule (generator.py:466) ------------------------------------
    import io
    import matplotlib.pyplot as plt
    from PIL import Image as PILImage
    x = np.random.RandomState(11).rand(9, 11, 3)
    x[0, 0] = [0.999 / 255, 254.6 / 255, 1.0]
    buf = io.BytesIO()
    plt.imsave(buf, np.clip(x[..., ::-1], 0, 1))
    buf.seek(0)
    out['imsave_in'] = x
    out['imsave_rgba'] = np.array(PILImage.open(buf))

    np.savez_compressed(os.path.join(HERE, 'reference_vectors.npz'), **out)
    print('wrote', os.path.join(HERE, 'reference_vectors.npz'), 'with', len(out), 'arrays')


if __name__ == '__main__':
    main()
