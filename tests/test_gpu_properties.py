"""GPU tier, BASELINE.json full size (1242x375, 100 mm/hr): size-independent properties and
the host-emulation comparison where the numpy oracle would take minutes."""
import numpy as np
import pytest

import helpers as h

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(tmp_path_factory, built):
    tmp = tmp_path_factory.mktemp('kitti100')
    sc = h.Scene(tmp, 375, 1242, 8192, seed0=3000)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    base = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)])[0]
    yield sc, bg, env, drops, rh, base
    rh.close()


def test_full_size_matches_hostemu_and_oracle_prefix(setup):
    sc, bg, env, drops, rh, base = setup
    emu = h.emu_render(sc, bg, bg, env, drops)
    assert np.array_equal(base['status'], emu['status'])
    assert np.array_equal(base['mask'], emu['mask'])                      # bit-exact
    assert np.array_equal(base['mask_i32'], emu['mask_i32'])
    assert np.abs(base['image_u8'].astype(int) - emu['image_u8'].astype(int)).max() <= 1
    # the float composite: same alpha bits, colour constants from FOV sums added in another order
    assert np.abs(base['rainy_bg'] - emu['rainy_bg']).max() < 1e-9
    # the first 300 streaks through the numpy oracle (faithful FOV integration)
    n = 300
    out = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops[:n])])[0]
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True, max_drops=n)
    assert np.array_equal(out['status'], ref['status'])
    assert np.array_equal(out['mask'], ref['mask'])
    assert np.array_equal(out['mask_i32'], ref['mask_i32'])
    assert np.abs(out['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1


def test_determinism_and_batch_invariance(setup):
    sc, bg, env, drops, rh, base = setup
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
    small = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops[:100])
    outs = rh.render_frames([small, fr, small])
    for k in ('mask', 'mask_i32', 'image_u8', 'status'):
        assert np.array_equal(outs[1][k], base[k]), k                      # same bits alone or inside a batch
        assert np.array_equal(outs[0][k], outs[2][k]), k


def test_mask_export_and_skips(setup):
    sc, bg, env, drops, rh, base = setup
    assert np.array_equal(base['mask_i32'], np.floor(base['mask'] * 255).astype(np.int32))   # decision D1
    assert base['mask'].min() >= 0 and base['mask'].max() > 1.0            # accumulator, not clipped
    skipped = base['status'] != 0
    assert 0 < skipped.sum() < len(drops) // 10
    kept = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops[~skipped])])[0]
    for k in ('mask', 'image_u8'):
        assert np.array_equal(kept[k], base[k]), k                         # skipped drops contribute nothing


def test_zero_opacity_and_mean_shift(setup):
    sc, bg, env, drops, rh, base = setup
    out = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops, opacity_attenuation=0.0)])[0]
    assert np.array_equal(out['mask'], base['mask'])                       # the mask ignores opacity (bad_weather.py:450)
    assert np.array_equal(out['rainy_bg'], bg)                             # tau_one = 0: the blend is the identity
    assert np.array_equal(out['image_u8'], (np.clip(bg[..., ::-1], 0, 1) * 255).astype(np.uint8))
    # epilogue: image = clip(rainy - (mean(rainy) - mean(bg))) truncated (generator.py:461-466)
    comp = base['rainy_bg']
    exp = (np.clip((comp - (comp.mean() - bg.mean()))[..., ::-1], 0, 1) * 255).astype(np.uint8)
    assert np.abs(exp.astype(int) - base['image_u8'].astype(int)).max() <= 1
    assert (exp != base['image_u8']).mean() < 1e-3


def test_raw_tile_dedup_is_invisible(setup):
    """k_dedup lets drops with bit-identical tile parameters share one raw tile (within a frame and across
    the frames of a batch).  Same bits with the election switched off (rr_set_option RR_OPT_DEDUP 0)."""
    sc, bg, env, drops, rh, base = setup
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
    two = rh.render_frames([fr, fr])                                       # second frame: every tile is a duplicate
    c0, c1 = rh.batch_counts(0), rh.batch_counts(1)
    ok = int(np.count_nonzero(base['status'] == 0))
    assert c0[7] + c1[7] > ok                                              # > one frame's worth of duplicates
    assert (c0[0] + c0[1] + c0[5]) + (c1[0] + c1[1] + c1[5]) + (c0[7] + c1[7]) >= 2 * ok   # every composited drop has a tile
    plain = h.hb.RainHip(0)
    plain.set_option(h.hb.RR_OPT_DEDUP, 0)
    plain.set_streak_db(sc.db.streaks_light)
    plain.set_camera(sc.cam)
    ref = plain.render_frames([fr])[0]
    assert plain.batch_counts(0)[7] == 0
    plain.close()
    for k in ('mask', 'mask_i32', 'image_u8', 'status', 'rainy_bg'):
        assert np.array_equal(ref[k], base[k]), k
        assert np.array_equal(two[0][k], base[k]) and np.array_equal(two[1][k], base[k]), k


def test_colour_path_options_do_not_change_results(setup):
    """The FOV-sum kernel's workgroup size / drops per thread, the LDS tile sizes of the fused blur and the general colour path (prefix table in HBM, what
    maps beyond 1024 rows / 4096 columns take) are tuning switches: the mask must be identical, the image within
    1 LSB (the colour sums are added in a different order), the statuses equal."""
    sc, bg, env, drops, rh, base = setup
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
    for opts in ({h.hb.RR_OPT_FOV_THREADS: 512, h.hb.RR_OPT_FOV_DROPS_PER_THREAD: 1},
                 {h.hb.RR_OPT_FOV_THREADS: 512, h.hb.RR_OPT_FOV_DROPS_PER_THREAD: 4},
                 {h.hb.RR_OPT_FOV_THREADS: 1024, h.hb.RR_OPT_FOV_DROPS_PER_THREAD: 2},
                 {h.hb.RR_OPT_FOV_THREADS: 1024, h.hb.RR_OPT_FOV_DROPS_PER_THREAD: 8},
                 {h.hb.RR_OPT_BLUR_WORKGROUPS: 3}, {h.hb.RR_OPT_BLUR_WORKGROUPS: 5},
                 {h.hb.RR_OPT_GENERAL_FOV: 1}):
        alt = h.hb.RainHip(0)
        try:
            for k, v in opts.items():
                alt.set_option(k, v)
            alt.set_streak_db(sc.db.streaks_light)
            alt.set_camera(sc.cam)
            out = alt.render_frames([fr])[0]
        finally:
            alt.close()
        assert np.array_equal(out['status'], base['status']), opts
        assert np.array_equal(out['mask'], base['mask']) and np.array_equal(out['mask_i32'], base['mask_i32']), opts
        assert np.abs(out['image_u8'].astype(int) - base['image_u8'].astype(int)).max() <= 1, opts
        assert np.abs(out['rainy_bg'] - base['rainy_bg']).max() < 1e-9, opts
    # texture staging from the pre-padded copies (default) or byte by byte: the same LDS bytes, so every output bit equal
    alt = h.hb.RainHip(0)
    try:
        alt.set_option(h.hb.RR_OPT_PADDED_TEXTURES, 0)
        alt.set_streak_db(sc.db.streaks_light)
        alt.set_camera(sc.cam)
        out = alt.render_frames([fr])[0]
    finally:
        alt.close()
    for k in ('mask', 'mask_i32', 'image_u8', 'status', 'rainy_bg'):
        assert np.array_equal(out[k], base[k]), k
    with pytest.raises(RuntimeError):
        rh.set_option(99, 1)


def test_float_colour_compositor(setup):
    """The default when nobody asks for the float64 composite (k_composite32: float colours, two pixels per lane, float
    composite; include/rainhip.h RR_OPT_COMPOSITE_F64): rainy_mask -- float64 accumulator and int32 export -- identical to
    the float64 compositor's and to the g++ build of the kernel arithmetic, drop statuses equal, the uint8 image within
    1 LSB of both (BASELINE.json's bar) and almost everywhere equal."""
    sc, bg, env, drops, rh, base = setup
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
    f32 = rh.render_frames([fr], want_composite=False)[0]
    assert f32['rainy_bg'] is None
    assert np.array_equal(f32['status'], base['status'])
    assert np.array_equal(f32['mask'], base['mask']) and np.array_equal(f32['mask_i32'], base['mask_i32'])     # bit-exact
    d = np.abs(f32['image_u8'].astype(int) - base['image_u8'].astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 5e-3
    emu = h.emu_render(sc, bg, bg, env, drops)
    assert np.array_equal(f32['mask'], emu['mask'])
    assert np.abs(f32['image_u8'].astype(int) - emu['image_u8'].astype(int)).max() <= 1
    # inside a batch, next to other frames, and with the option that forces float64 colours
    small = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops[:100])
    outs = rh.render_frames([small, fr, small], want_composite=False)
    for k in ('mask', 'mask_i32', 'image_u8', 'status'):
        assert np.array_equal(outs[1][k], f32[k]), k
    rh.set_option(h.hb.RR_OPT_COMPOSITE_F64, 1)
    try:
        f64 = rh.render_frames([fr], want_composite=False)[0]
    finally:
        rh.set_option(h.hb.RR_OPT_COMPOSITE_F64, 0)
    for k in ('mask', 'mask_i32', 'image_u8', 'status'):
        assert np.array_equal(f64[k], base[k]), k
    # the first 300 streaks against the numpy oracle (faithful FOV integration), float colours
    n = 300
    out = rh.render_frames([dict(fr, drops=drops[:n])], want_composite=False)[0]
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True, max_drops=n)
    assert np.array_equal(out['status'], ref['status'])
    assert np.array_equal(out['mask'], ref['mask']) and np.array_equal(out['mask_i32'], ref['mask_i32'])
    assert np.abs(out['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1


def test_narrow_input_types(setup):
    """rr_frame_in.in_types: the image as float32 or as the bytes cv2.imread returned, the xyY map and the solid angles as
    float32 -- a third to an eighth of the HBM bytes of the float64 arrays.  The mask never reads them (bit-exact, statuses
    equal); the image stays within 1 LSB of the float64-input rendering and of the g++ build of the kernel arithmetic.
    Both compositors, a fogged image of its own type next to the plain one, and the colour branch forced to float64."""
    sc, bg, env, drops, rh, base = setup
    emu = h.emu_render(sc, bg, bg, env, drops)

    def check(out, ref_img, tag):
        assert np.array_equal(out['status'], base['status']), tag
        assert np.array_equal(out['mask'], base['mask']) and np.array_equal(out['mask_i32'], base['mask_i32']), tag
        d = np.abs(out['image_u8'].astype(int) - ref_img.astype(int)).max()
        assert d <= 1, '%s: image differs by %d LSB' % (tag, d)

    bg32, env32 = bg.astype(np.float32), env.astype(np.float32)
    for comp in (False, True):
        out = rh.render_frames([dict(bg=bg32, rainy_bg=bg32, env_xyY=env32, omega=sc.omega, drops=drops)], want_composite=comp)[0]
        check(out, base['image_u8'], 'float32 image + map, composite=%s' % comp)
        check(out, emu['image_u8'], 'float32 image + map vs hostemu, composite=%s' % comp)
        out = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env32, omega=sc.omega, drops=drops)], want_composite=comp)[0]
        check(out, base['image_u8'], 'float64 image, float32 map, composite=%s' % comp)
    # bytes: the reference's own input (bg = imread / 255.0); rainy_bg a separate float32 array (a fogged image)
    bg8 = np.round(bg * 255).astype(np.uint8)
    ref8 = rh.render_frames([dict(bg=bg8 / 255.0, rainy_bg=bg8 / 255.0, env_xyY=env, omega=sc.omega, drops=drops)])[0]
    for comp in (False, True):
        out = rh.render_frames([dict(bg=bg8, rainy_bg=bg8, env_xyY=env32, omega=sc.omega, drops=drops)], want_composite=comp)[0]
        check(out, ref8['image_u8'], 'uint8 image, composite=%s' % comp)
        rainy = (0.9 * (bg8 / 255.0)).astype(np.float32)
        ref = rh.render_frames([dict(bg=bg8 / 255.0, rainy_bg=rainy.astype(np.float64), env_xyY=env, omega=sc.omega, drops=drops)])[0]
        out = rh.render_frames([dict(bg=bg8, rainy_bg=rainy, env_xyY=env, omega=sc.omega, drops=drops)], want_composite=comp)[0]
        check(out, ref['image_u8'], 'uint8 bg + float32 rainy_bg, composite=%s' % comp)
    # inside a batch: image types may differ from frame to frame, the map's type is one per batch (RR_E_ARG otherwise);
    # device-resident solid angles (omega=None) as float32
    rh.set_solid_angles(sc.omega)
    outs = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env32, omega=None, drops=drops[:200]),
                             dict(bg=bg32, rainy_bg=bg32, env_xyY=env32, omega=None, drops=drops)], want_composite=False)
    check(outs[1], base['image_u8'], 'mixed batch')
    with pytest.raises(RuntimeError):
        rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops[:200]),
                          dict(bg=bg32, rainy_bg=bg32, env_xyY=env32, omega=None, drops=drops[:200])], want_composite=False)
    # the colour branch in float64 throughout (RR_OPT_FOV_F32 0) and in float always (1): same statuses, same mask
    for v in (0, 1):
        rh.set_option(h.hb.RR_OPT_FOV_F32, v)
        try:
            for comp in (False, True):
                out = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)], want_composite=comp)[0]
                check(out, base['image_u8'], 'RR_OPT_FOV_F32 %d, composite=%s' % (v, comp))
        finally:
            rh.set_option(h.hb.RR_OPT_FOV_F32, 2)


def test_thread_per_drop_polygons_equal_the_edge_parallel_kernel(setup, tmp_path_factory):
    """The float colour branch's default kernel (k_fov_dda: a thread per drop, two cursors down the polygon's sides; wrapping
    polygons and float64 decisions through a list to k_fov_spans) against k_fov_spans for every drop (RR_OPT_FOV_DDA 0): the
    row spans are the same, so the colour constants are the same BITS, and so is every output.  KITTI 100 mm/hr (a few
    wrapping polygons per frame) and a wide-angle scene where drops all around the camera wrap often."""
    sc, bg, env, drops, rh, base = setup
    scenes = [(sc, bg, env, drops)]
    wide = h.Scene(tmp_path_factory.mktemp('wide'), 180, 320, 1500, cam=h.NUSCENES, seed0=777, far_fraction=0.3)
    wbg, wenv = wide.frame_inputs(0)
    scenes.append((wide, wbg, wenv, wide.product_drops(0)))
    for S, b, e, d in scenes:
        for rule in (1, 0):              # OpenCV's fillConvexPoly (default) and the span rule: the cursors walk either
            outs = []
            for dda in (1, 0):           # 1 = k_fov_dda + the list, 0 = k_fov_spans for every drop
                alt = h.hb.RainHip(0)
                try:
                    alt.set_option(h.hb.RR_OPT_FOV_DDA, dda)
                    alt.set_option(h.hb.RR_OPT_FOV_FILL_RULE, rule)
                    alt.set_streak_db(S.db.streaks_light)
                    alt.set_camera(S.cam)
                    outs.append(alt.render_frames([dict(bg=b, rainy_bg=b, env_xyY=e, omega=S.omega, drops=d)], want_composite=False, want_colour=True)[0])
                finally:
                    alt.close()
            assert (outs[0]["status"] == 0).sum() > 0.5 * len(d)
            for k in ('status', 'colour', 'mask', 'mask_i32', 'image_u8'):
                assert np.array_equal(outs[0][k], outs[1][k]), (rule, k)


def test_composite_codes_and_blur_prefetch(setup):
    """r05 tuning switches of the float-colour route.  RR_OPT_BLUR_DMA (the fused blur's sub-tiles staged a sub-tile ahead by
    LDS-DMA loads), RR_OPT_BIN_ROWS (how the ordered per-tile drop lists are made) and RR_OPT_COMPOSITE_BATCH (how the
    compositor gets at its list entries' records) and RR_OPT_COLOUR_STREAM (which of the step's two chains runs on the library's
    second stream, or everything on one in-order stream) change no bit.  RR_OPT_COMPOSITE_U16 (the composite before the mean shift as 16-bit codes
    instead of floats) keeps mask and statuses and moves the uint8 image by at most 1 LSB on a few pixels in a thousand;
    values outside [0, 1] -- a pixel no drop was blended into -- go through the code 65535 and come out as
    before."""
    sc, bg, env, drops, rh, base = setup
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
    wild = bg.copy()
    wild[5, 7] = (0.25, 7.5, -3.0)             # (finite: the mean shift of the frame stays a number)
    wild[200:204, 600:640] = (1.5, -0.5, 2.0)
    wild[300:310, 40:50] = 1.0
    wild[310:320, 40:50] = 0.0
    frw = dict(fr, bg=wild, rainy_bg=wild)
    ref = rh.render_frames([fr, frw], want_composite=False)
    for opt, off, on in ((h.hb.RR_OPT_BLUR_DMA, 0, 1), (h.hb.RR_OPT_BIN_ROWS, 0, 1), (h.hb.RR_OPT_COMPOSITE_BATCH, 0, 1),
                         (h.hb.RR_OPT_COLOUR_STREAM, 0, 1), (h.hb.RR_OPT_COMPOSITE_U16, 0, 1)):
        try:
            rh.set_option(opt, off)
        except RuntimeError:
            assert opt == h.hb.RR_OPT_BLUR_DMA         # r04's register-staged kernel: -DRR_EXPERIMENTS builds only (r06)
            continue
        try:
            alt = rh.render_frames([fr, frw], want_composite=False)
        finally:
            rh.set_option(opt, on)
        for a, b in zip(ref, alt):
            for k in ('status', 'mask', 'mask_i32'):
                assert np.array_equal(a[k], b[k]), (opt, k)
            d = np.abs(a['image_u8'].astype(int) - b['image_u8'].astype(int))
            if opt != h.hb.RR_OPT_COMPOSITE_U16:
                assert d.max() == 0
            else:
                assert d.max() <= 1 and (d != 0).mean() < 4e-3, (d.max(), (d != 0).mean())
    # the coded composite against the float64 compositor and the host build (the 1-LSB bar of BASELINE.json)
    assert np.abs(ref[0]['image_u8'].astype(int) - base['image_u8'].astype(int)).max() <= 1


def test_compositor_record_batches(setup):
    """The float compositor holds the records of 64 list entries at a time (RR_OPT_COMPOSITE_BATCH) and takes a coarse tile's
    list in pieces of 256.  A frame whose drops are 100 streaks repeated 80 times makes every list that is not empty longer
    than a batch and many longer than a piece: same bits as the entry-at-a-time compositor at every register allocation,
    and the mask of the float64 compositor."""
    sc, bg, env, drops, rh, base = setup
    dense = np.concatenate([drops[:100]] * 80)
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=dense)
    ref = rh.render_frames([fr], want_composite=False)[0]
    f64 = rh.render_frames([fr])[0]
    assert np.array_equal(ref['mask'], f64['mask']) and np.array_equal(ref['status'], f64['status'])
    assert np.abs(ref['image_u8'].astype(int) - f64['image_u8'].astype(int)).max() <= 1
    assert ref['mask'].max() > 40.0                                       # (80 copies of a streak on top of each other)
    for batch, waves in ((0, 0), (0, 8), (1, 4), (1, 6), (1, 8)):
        rh.set_option(h.hb.RR_OPT_COMPOSITE_BATCH, batch)
        rh.set_option(h.hb.RR_OPT_COMPOSITE_WAVES, waves)
        try:
            alt = rh.render_frames([fr], want_composite=False)[0]
        finally:
            rh.set_option(h.hb.RR_OPT_COMPOSITE_BATCH, 1)
            rh.set_option(h.hb.RR_OPT_COMPOSITE_WAVES, 0)
        for k in ('status', 'mask', 'mask_i32', 'image_u8'):
            assert np.array_equal(ref[k], alt[k]), (batch, waves, k)


def test_fill_rule_option(setup):
    """RR_OPT_FOV_FILL_RULE.  The default (1, since round 6) is OpenCV 3.2's own fillConvexPoly algorithm on the FAST colour path
    (k_fov_dda's cursors evaluate rr_device.h fov_rowspan_cv's closed form: Bresenham outline + 16.16 edge walkers; pinned
    against the literal restatement in tests/test_fill_rules.py); 0 is the row-span rule of rounds 1-5.  Under either rule
    the library's colour constants equal the host build's under the same rule, with the float64 and with the float colour
    branch, and the general colour path (RR_OPT_GENERAL_FOV) agrees with the fast one; between the rules the constants move
    by a few parts in a thousand -- rainy_image stays within 1 LSB, mask and statuses are untouched."""
    sc, bg, env, drops, rh, base = setup
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
    emu = h.hostemu()
    res = {}
    for rule in (1, 0):
        alt = h.hb.RainHip(0)
        try:
            alt.set_option(h.hb.RR_OPT_FOV_FILL_RULE, rule)
            alt.set_streak_db(sc.db.streaks_light)
            alt.set_camera(sc.cam)
            out = alt.render_frames([fr], want_colour=True)[0]                  # float64 colour branch (k_fov_spans)
            f32 = alt.render_frames([fr], want_composite=False)[0]              # float colour branch (k_fov_dda + the list)
            alt.set_option(h.hb.RR_OPT_GENERAL_FOV, 1)
            gen = alt.render_frames([fr], want_colour=True)[0]                  # prefix table in HBM, fov_rowspan[_cv] per drop and row
        finally:
            alt.close()
        emu.emu_set_fill_rule(rule)
        try:
            ref = h.emu_render(sc, bg, bg, env, drops)
        finally:
            emu.emu_set_fill_rule(1)
        for k in ('status', 'mask', 'mask_i32'):
            assert np.array_equal(out[k], base[k]) and np.array_equal(out[k], ref[k]) and np.array_equal(f32[k], base[k]), (rule, k)
        ok = out['status'] == 0
        assert np.abs(out['colour'][ok] - ref['K'][ok]).max() <= 1e-9 * np.abs(ref['K'][ok]).max(), rule
        assert np.abs(gen['colour'][ok] - ref['K'][ok]).max() <= 1e-9 * np.abs(ref['K'][ok]).max(), rule
        assert np.abs(out['rainy_bg'] - ref['rainy_bg']).max() < 1e-9
        assert np.abs(out['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
        assert np.abs(f32['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
        res[rule] = (out, f32)
    # the default IS rule 1
    dflt = rh.render_frames([fr], want_colour=True)[0]
    d32 = rh.render_frames([fr], want_composite=False)[0]
    assert np.array_equal(dflt['colour'], res[1][0]['colour']) and np.array_equal(d32['image_u8'], res[1][1]['image_u8'])
    ok = dflt['status'] == 0
    rel = np.abs(res[0][0]['colour'][ok] - dflt['colour'][ok]) / np.abs(dflt['colour'][ok])
    assert 1e-5 < rel.max() < 4e-3, rel.max()
    for o in res[0]:
        d = np.abs(o['image_u8'].astype(int) - base['image_u8'].astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 0.05, (d.max(), (d != 0).mean())


def test_raw_tile_kernels_agree(setup, tmp_path):
    """Round 6's batch-wide wave-per-tile kernel (k_tile_rows: row walks for the rotate + INTER_AREA tiles, a lane per pixel for
    the Big drops' bicubic warps up to 8192 pixels) against the kernels it took the work from -- k_tile (a workgroup per
    tile) and k_tile_big (a thread per pixel), which still render what does not fit it: RR_OPT_TILE_ROWS 0 / 1 / 2 and the
    number of shares the tile list is cut into give the same bits.  A KITTI frame and a frame of the nuScenes camera (f/1.8:
    Big tiles on both sides of the pixel limit), two frames per call so that tiles are shared across frames."""
    sc, bg, env, drops, rh, base = setup
    sc2 = h.Scene(tmp_path, 450, 800, 1500, cam=h.NUSCENES, seed0=6100, far_fraction=0.2)
    bg2, env2 = sc2.frame_inputs(0)
    drops2 = sc2.product_drops(0)
    for scn, b, e, d in ((sc, bg, env, drops), (sc2, bg2, env2, drops2)):
        fr = dict(bg=b, rainy_bg=b, env_xyY=e, omega=scn.omega, drops=d)
        outs = []
        for opts in ({h.hb.RR_OPT_TILE_ROWS: 0}, {h.hb.RR_OPT_TILE_ROWS: 1}, {h.hb.RR_OPT_TILE_ROWS: 2},
                     {h.hb.RR_OPT_TILE_ROWS: 2, h.hb.RR_OPT_ROWS_SHARES: 1}, {h.hb.RR_OPT_TILE_ROWS: 2, h.hb.RR_OPT_ROWS_SHARES: 8}):
            alt = h.hb.RainHip(0)
            try:
                for k, v in opts.items():
                    alt.set_option(k, v)
                alt.set_streak_db(scn.db.streaks_light)
                alt.set_camera(scn.cam)
                outs.append((alt.render_frames([fr, fr]), [alt.batch_counts(0), alt.batch_counts(1)]))
            finally:
                alt.close()
        ref = outs[0][0]
        assert (ref[0]['status'] == 0).sum() > 0.5 * len(d) and ref[0]['mask'].max() > 0
        for out, counts in outs[1:]:
            for a, b_ in zip(ref, out):
                for k in ('status', 'mask', 'mask_i32', 'image_u8', 'rainy_bg'):
                    assert np.array_equal(a[k], b_[k]), k
            pick = lambda c: [int(c[0]), int(c[1]), int(c[5]), int(c[7])]                # rotate / generic / Big tiles rendered, tiles shared
            # (which of two identical drops of different frames renders their tile is the election's business: the call's totals)
            assert np.sum([pick(c) for c in counts], axis=0).tolist() == np.sum([pick(c) for c in outs[0][1]], axis=0).tolist()


def test_bcast_streak_db_in_one_process(setup):
    """SURVEY 8b's rr_bcast_streak_db (round 6): a second context of the same process gets its streak database from the first
    one's device memory and renders the same bits; the RCCL leg (contexts on other devices) runs here as a communicator of
    one rank on this box's one GPU (rr_bcast_selftest).  Contexts without a database, null entries: errors."""
    sc, bg, env, drops, rh, base = setup
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
    other, third = h.hb.RainHip(0), h.hb.RainHip(0)
    try:
        with pytest.raises(RuntimeError):
            h.hb.bcast_streak_db([other, third])                 # the root has no database
        h.hb.bcast_streak_db([rh])                               # n_ctx = 1: nothing to do
        h.hb.bcast_streak_db([rh, other, third])
        for c in (other, third):
            c.set_camera(sc.cam)
            out = c.render_frames([fr])[0]
            for k in ('status', 'mask', 'mask_i32', 'image_u8', 'rainy_bg'):
                assert np.array_equal(out[k], base[k]), k
        rh.bcast_selftest()
        other.bcast_selftest()
    finally:
        other.close()
        third.close()
