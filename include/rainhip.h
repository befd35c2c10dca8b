/*
 * rainhip.h -- C ABI of librainhip.so: the MI355X (gfx950) implementation of the
 * per-image rain-streak rendering + compositing hot path of
 * astra-vision/rain-rendering.
 *
 * The reference is pure Python and has no FFI; the boundary this library plugs
 * into is the per-frame body of Generator.run (reference common/generator.py:389-467)
 * and its two inner seams
 *     Generator.compute_drop              reference common/generator.py:119-191
 *     RainRenderer.add_drop_to_image      reference common/bad_weather.py:336-462
 * A maintainer binds it with ctypes.  Plain pointers and
 * sizes only; the caller owns every buffer; no function throws; every function
 * returns 0 or a negative RR_E* code and rr_last_error() explains it.
 *
 * There is NO CPU implementation behind this ABI: rr_create fails with
 * RR_E_NO_DEVICE when no gfx950 device is visible.
 */
#ifndef RAINHIP_H
#define RAINHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_VERSION 400            /* 0.4.0: + rr_frame_in.in_types (float32 / uint8 image and map inputs), float32 colour branch by default */

enum {
  RR_OK = 0,
  RR_E_ARG = -1,                  /* bad argument (null pointer, bad size, mixed frame sizes) */
  RR_E_NO_DEVICE = -2,            /* no HIP device / not gfx950 */
  RR_E_HIP = -3,                  /* a HIP runtime call failed */
  RR_E_STATE = -4,                /* streak DB or camera not set */
  RR_E_ARENA = -5,                /* tile arena overflow: arena regrown, enqueue / submit the batch again */
  RR_E_PARSE = -6,                /* rr_host_parse_particles: malformed file or missing / non-numeric attribute */
  RR_E_UNSUPPORTED = -7           /* rr_host_parse_particles: XML construct outside the simulator's subset (DOCTYPE, CDATA,
                                   * entity references): use a full XML parser */
};

/* per-drop status written to rr_frame_out.drop_status: 0 == composited.  The
 * reference expresses all of these as "exception inside add_drop_to_image ->
 * drop not composited" (common/generator.py:180-189). */
enum {
  RR_DROP_OK = 0,
  RR_DROP_FOV_FAIL = 1,           /* compute_fov_plane_points returned [] (bad_weather.py:698-704): e.g. > radius */
  RR_DROP_EMPTY_FOV = 2,          /* FOV polygon does not touch the envmap (bad_weather.py:372 IndexError) */
  RR_DROP_BAD_COC = 3,            /* non-finite circle of confusion (int(10*c) raises, bad_weather.py:293) */
  RR_DROP_TOO_BIG = 4             /* defocus pad > RR_MAX_SHIFT px: documented limit of this library */
};

#define RR_MAX_SHIFT 1024
#define RR_MAX_FOV 32

typedef struct rr_ctx rr_ctx;

struct rr_sim_frame;

/* Camera / renderer constants.  Mirrors RainRenderer(focal, f_number, focus_plane=6,
 * radius=10, fov=165) (generator.py:267), N=20 (generator.py:179) and the constants of
 * bad_weather.py:344-345,425,469.  Transcendental-derived values are computed ONCE by
 * the host with numpy so that device results do not depend on a device libm. */
typedef struct {
  double focal_m;                 /* settings["cam_focal"] / 1000 */
  double focal_sq;                /* focal_m ** 2 evaluated by the host (bad_weather.py:468) */
  double f_number;
  double focus_plane;             /* 6 */
  double exposure_s;              /* settings["cam_exposure"] / 1000 (bad_weather.py:344) */
  double radius;                  /* 10 */
  double sensor_px;               /* 4.65e-06 (bad_weather.py:469) */
  double tau_zero;                /* sqrt(1.16e-3) / 50 (bad_weather.py:425) */
  double fov_cos, fov_sin;        /* cos/sin(-deg2rad(fov/2)) (bad_weather.py:602,625) */
  double phi_cos[RR_MAX_FOV];     /* cos(k * 2*pi/N)  (bad_weather.py:630-634) */
  double phi_sin[RR_MAX_FOV];
  int32_t n_fov;                  /* 20 */
  int32_t reserved;
} rr_camera;

/* One streak that passed the frame filter (generator.py:413-420), as plain data.
 * The legacy-RandomState draws (bad_weather.py:252-264, generator.py:136) stay in
 * Python; their results arrive here as tex_index and rot_cos/rot_sin. */
typedef struct {
  int32_t x0, y0, x1, y1;         /* image_position_start / _end (ints, after the in-place noise rotation generator.py:152-161) */
  int32_t max_width;              /* Streak.max_width */
  int32_t length;                 /* Streak.length */
  int32_t type;                   /* DropType: 0 Big, 1 Medium, 2 Small */
  int32_t tex_index;              /* index into the streak DB chosen by take_drop_texture */
  double iw1, iw2;                /* image_diameter_start / _end */
  double wps[3], wpe[3];          /* world_position_start / _end (z already negated, bad_weather.py:223-224) */
  double rot_cos, rot_sin;        /* cos/sin(-(theta+noise) * pi/180) for imutils.rotate_bound (non-Big) */
} rr_drop;                        /* 112 bytes */

/* A caller-made tile for one drop: the seam of RainRenderer.add_drop_to_image (common/bad_weather.py:336-338), whose
 * arguments `drop` (the RGBA tile Generator.compute_drop built: gray, alpha == channel 0), `drop_minC` and `drop_fov_pts`
 * these are.  With alpha != NULL the library neither synthesises the streak tile nor evaluates the field-of-view polygon
 * of that drop; everything after that (colour, defocus, placement, blend) is the common path. */
typedef struct {
  const double* alpha;            /* th*tw values in [0,1], row-major; NULL: this drop is rendered the normal way */
  int32_t tw, th;
  int32_t min_x, min_y;           /* drop_minC */
  int32_t n_poly, reserved;       /* vertices of drop_fov_pts (<= RR_MAX_FOV + 4); 0: compute_fov_plane_points failed ([]) */
  const double* poly_xy;          /* n_poly (x, y) pairs, float pixels of the environment map (truncated like pyclipper does) */
} rr_ext_tile;

typedef struct {
  int32_t H, W;                   /* frame */
  int32_t He, We;                 /* lat-long environment map */
  /* the four arrays below are float64 unless in_types says otherwise */
  const void* bg;                 /* H*W*3 BGR, un-fogged image / 255 (used for the mean shift, generator.py:462) */
  const void* rainy_bg;           /* H*W*3 BGR, output of the fog pre-pass (generator.py:386): values in [0, 1] (it ends with
                                   * np.clip); a NaN stays a NaN like in the reference */
  const void* env_xyY;            /* He*We*3 (generator.py:407-408) */
  const void* omega;              /* He*We solid angles (generator.py:410); NULL: the map given to rr_set_solid_angles */
  const rr_drop* drops;           /* n_drops records in reference order */
  int32_t n_drops;
  int32_t strategy;               /* 0: default (rendering_strategy=None); 1: 'white' (bad_weather.py:349-353) */
  double opacity_attenuation;     /* --opacity_attenuation */
  const void* depth;              /* optional, only read with RR_OPT_DEPTH_OCCLUSION: scene depth in metres, H*W float32
                                   * (depth_f64 == 0) or float64; rr_pipeline_* use the pre-pass' depth instead */
  int32_t depth_f64;
  /* Element types of the image and map inputs (0: everything float64, the reference's arrays).  The three colour channels
   * and the drop colour only have to land within 1 LSB of a uint8 (BASELINE.json) and the mask never reads these arrays, so
   * a caller that holds them narrower hands them over as they are -- a third to an eighth of the HBM bytes:
   *   RR_IN_BG_F32 / RR_IN_BG_U8         `bg` is float32 / uint8 (uint8: the bytes cv2.imread returned; bg = bytes / 255.0)
   *   RR_IN_RAINY_F32 / RR_IN_RAINY_U8   the same for `rainy_bg` (when rainy_bg == bg the BG flags count for both)
   *   RR_IN_ENV_F32                      `env_xyY` AND `omega` (when not NULL) are float32; for every frame of a batch or for none
   * rr_render_frames_device and rr_render_frames; rr_pipeline_* produce rainy_bg / env_xyY themselves. */
  int32_t in_types;
  const rr_ext_tile* ext;         /* optional: n_drops entries (rr_render_frames / rr_render_frames_device only) */
  /* Drop tables born on the device (rr_generate_drops_device): with n_drops_dev != NULL (a DEVICE pointer to one int32;
   * rr_render_frames_device only) the frame's drop count is read from there when the kernels run, clamped to n_drops, which
   * then is the CAPACITY of `drops` (and of drop_status / drop_colour) the launch is sized for. */
  const int32_t* n_drops_dev;
  /* In-kernel particle simulation (BASELINE config 5; host-pointer entry points rr_render_frames / rr_pipeline_*): with
   * sim != NULL (HOST pointer to this frame's generator settings) the frame's drop table is generated on the device
   * (rr_generate_drops_device semantics) instead of uploaded: `drops` is ignored (may be NULL) and n_drops is the capacity
   * the launch is sized for (0: sim->n_particles).  The count comes back in rr_frame_out.n_drops_out. */
  const struct rr_sim_frame* sim;
} rr_frame_in;

enum { RR_IN_BG_F32 = 1, RR_IN_BG_U8 = 2, RR_IN_ENV_F32 = 4, RR_IN_RAINY_F32 = 8, RR_IN_RAINY_U8 = 16, RR_IN_BG_PNG_ROWS = 32 };

typedef struct {
  uint8_t* rainy_rgb;             /* H*W*3 RGB: what plt.imsave(rainy_image) stores (generator.py:461-466), alpha omitted */
  double* rainy_bg_out;           /* optional (may be NULL): H*W*3 BGR composite before the mean shift */
  double* mask_f64;               /* H*W: rainy_mask accumulator (bad_weather.py:450); may be NULL */
  int32_t* mask_i32;              /* H*W: floor(mask_f64 * 255)  (SURVEY decision D1); may be NULL */
  int32_t* drop_status;           /* n_drops RR_DROP_* codes (may be NULL) */
  /* Optional: the two PNG files the reference writes per frame (generator.py:466-467), as PNG
   * scanlines ready for deflate: H rows of 1 + 4*W bytes = filter byte 1 (Sub) + the Sub-filtered RGBA pixels.
   * rainy_png: plt.imsave(rainy_image) (alpha 255); mask_png: plt.imsave(rainy_mask) = (mask - min) / (max - min) through
   * the 256-entry colour map given to rr_set_colormap (needs mask_f64 in the device-pointer entry points).
   * With RR_OPT_PNG_DEFLATE the same buffers come back entropy-coded (see the option): 16-byte header + the file's zlib stream. */
  uint8_t* rainy_png;
  uint8_t* mask_png;
  /* Optional: n_drops * 3 doubles, the B, G, R colour constants K of every drop: the tile colour the reference derives
   * from the environment map inside the drop's field of view (bad_weather.py:397-412) is K * gray value.  Zeros for a
   * drop that is not composited.  Lets the single-drop seam hand back the reference's `drop_vis` (bad_weather.py:462). */
  double* drop_colour;
  int32_t* n_drops_out;           /* optional: one int32, the number of drops of a generated drop table (rr_frame_in.sim) */
} rr_frame_out;

typedef struct {
  char name[32];
  int32_t launches;
  int32_t reserved;
  double total_ms;                /* sum of HIP-event durations */
} rr_kernel_stat;

int rr_version(void);

/* device: HIP ordinal.  Fails (RR_E_NO_DEVICE) without a gfx9 device. */
int rr_create(rr_ctx** out, int device);
int rr_destroy(rr_ctx* ctx);
const char* rr_last_error(rr_ctx* ctx);

/* SURVEY 8b's collective, for a host that drives several GPUs from ONE process: the streak database of ctxs[0]
 * (rr_set_streak_db / rr_set_streak_db_device) becomes the database of ctxs[1 .. n_ctx - 1] without a trip through the host.
 * Contexts on the root's device take a device-to-device copy; contexts on other devices an ncclBroadcast (RCCL over xGMI; one
 * communicator rank per device, librccl.so.1 loaded on first use -- RR_E_STATE if it cannot be).  Synchronous; the error text
 * is ctxs[0]'s rr_last_error.  n_ctx = 1 is a no-op.  (This repo's drivers run one PROCESS per GPU and broadcast through
 * torch.distributed -- sharding.py; the reference has no counterpart: main_threaded.py starts a process per sequence.) */
int rr_bcast_streak_db(rr_ctx** ctxs, int32_t n_ctx);
/* Diagnostic: the RCCL leg of rr_bcast_streak_db on the context's OWN device (a communicator of one rank; the database
 * broadcast into a scratch buffer and compared) -- what a box with one GPU can check of it. */
int rr_bcast_selftest(rr_ctx* ctx);

/* Streak database: n_tex gray uint8 textures, already normalised as
 * uint8((255*norm*img)/65535) (bad_weather.py:141), concatenated in `texels`
 * (host pointer) with per-texture height/width/offset. */
int rr_set_streak_db(rr_ctx* ctx, const uint8_t* texels, const int32_t* tex_h, const int32_t* tex_w,
                     const int64_t* tex_off, int32_t n_tex);
/* Same, but `texels` is a DEVICE pointer on ctx's device (e.g. the buffer that just
 * received the RCCL broadcast of the packed database). */
int rr_set_streak_db_device(rr_ctx* ctx, const uint8_t* texels_dev, int64_t n_bytes, const int32_t* tex_h,
                            const int32_t* tex_w, const int64_t* tex_off, int32_t n_tex);
int rr_set_camera(rr_ctx* ctx, const rr_camera* cam);
/* The solid-angle map of He x We environment maps (generator.py:410: a function of the map's shape alone), kept resident
 * on the device: frames of that map size may then pass omega == NULL instead of uploading it with every batch. */
int rr_set_solid_angles(rr_ctx* ctx, int32_t He, int32_t We, const double* omega);
/* 256 RGBA byte entries of the colour map plt.imsave applies to rainy_mask (matplotlib's default: viridis). */
int rr_set_colormap(rr_ctx* ctx, const uint8_t* lut_rgba);

/* Render n frames (all with identical H,W,He,We).  Pointers inside `in`/`out` are HOST
 * pointers; the call uploads, renders, downloads and returns after completion. */
int rr_render_frames(rr_ctx* ctx, int32_t n, const rr_frame_in* in, const rr_frame_out* out);

/* Same, but every pointer inside `in`/`out` is a DEVICE pointer on ctx's device.
 * Work is enqueued on `stream` (a hipStream_t; NULL = the ctx's own stream) and the
 * call returns without waiting.  drop_status/arena overflow are checked at
 * rr_synchronize(). */
int rr_render_frames_device(rr_ctx* ctx, int32_t n, const rr_frame_in* in, const rr_frame_out* out, void* stream);
int rr_synchronize(rr_ctx* ctx);

/* ---------------------------------------------------------------------------------------
 * Pre-passes (SURVEY 8f "next" #1, #2): the two image passes that PRODUCE rainy_bg and
 * env_xyY, so that they are born in HBM.  Optional: a caller may keep computing them itself.
 *   fog-like rain attenuation   FogRain.fog_rain_layer           common/add_attenuation.py:88-95
 *   environment map             EnvironmentMapGenerator.generate_map  common/bad_weather.py:742-819
 *                               + convert_rgb_to_xyY              common/generator.py:407-408
 * As with rr_camera, everything derived from transcendentals is computed once by the host. */
#define RR_MAX_TAPS 33
#define RR_PRE_ENV_ONLY 1
#define RR_DEPTH_U16 2
#define RR_DEPTH_PNG_ROWS 3

typedef struct {
  int32_t fog_ksize, env_ksize;   /* 25 (add_attenuation.py:79), 15 (bad_weather.py:815); odd, <= RR_MAX_TAPS */
  double fog_w[RR_MAX_TAPS];      /* cv::getGaussianKernel(25, 25) */
  double env_w[RR_MAX_TAPS];      /* cv::getGaussianKernel(15, 0) -> sigma 2.6 */
} rr_prepass_kernels;

typedef struct {
  int32_t H, W;
  const void* bg;                 /* H*W*3 BGR image / 255 (generator.py:352): float64, or -- in_types -- float32 (RR_IN_BG_F32), or the
                                   * bytes cv2.imread returned (RR_IN_BG_U8: bg = bytes / 255.0 is formed where the kernels read it) */
  const void* depth;              /* H*W metres, float32 (depth_f64 == 0) or float64 (1) (generator.py:362-383) -- or (RR_DEPTH_U16 = 2)
                                   * the uint16 samples of the 16-bit depth PNG as cv2.imread(.., IMREAD_UNCHANGED) returns them:
                                   * metres = sample.astype(float32) / 256 (generator.py:366) is formed where the kernels read it,
                                   * half the upload of the float32 map.  Not with RR_OPT_DEPTH_OCCLUSION. */
  int32_t depth_f64;
  int32_t mode;                   /* 0: fog layer (+ the map if an output asks for it).  RR_PRE_ENV_ONLY (1), rr_prepass_frames*
                                   * only: `bg` IS the fogged image and only the map is made -- the stand-alone
                                   * EnvironmentMapGenerator.generate_map(background) (bad_weather.py:742-819); depth and the fog
                                   * constants are ignored, rr_prepass_out.rainy_bg may be NULL.  One mode per batch. */
  double beta_ext;                /* 0.312 * R**0.67                               add_attenuation.py:40-43 */
  double beta_hg;                 /* Henyey-Greenstein phase term, g = 0.97         add_attenuation.py:60-64 */
  double irr_num, irr_den;        /* 4*N**2  and  exposure_s*gain*pi                add_attenuation.py:51-54 */
  const uint8_t* bg_u8;           /* HOST entry points, kept from version 300: when set, the same as bg = bg_u8 with RR_IN_BG_U8
                                   * (1/8 of the PCIe traffic of the float64 image) */
  int32_t in_types;               /* RR_IN_BG_F32 or RR_IN_BG_U8 (or 0: float64); the other RR_IN_* bits are not for the pre-pass.
                                   * rr_pipeline_* / rr_prepass_frames (host pointers) also take RR_IN_BG_PNG_ROWS: `bg` holds the H rows of
                                   * 1 + 3*W bytes an 8-bit RGB PNG's IDAT stream inflates to (filter type + filtered R G B bytes, as
                                   * rr_io_read_frames_rows delivers them), and depth_f64 = RR_DEPTH_PNG_ROWS: `depth` holds the H rows of
                                   * 1 + 2*W bytes of the 16-bit gray depth file; the filters are reversed on the device */
  int32_t reserved;
} rr_prepass_in;

/* Element types of the pre-pass' outputs (rr_prepass_out.out_types).  The pre-pass computes in float64 whatever the
 * types: a float32 output is the float64 result rounded once (what numpy's astype(float32) of the reference's arrays
 * gives), and the uint8 map does not depend on them.  The hot path takes both widths (rr_frame_in.in_types). */
enum { RR_OUT_RAINY_F32 = 1, RR_OUT_ENV_F32 = 2 };

typedef struct {
  void* rainy_bg;                 /* H*W*3: FOG.fog_rain_layer(bg, depth); float64, or float32 with RR_OUT_RAINY_F32 */
  void* env_xyY;                  /* H*We*3, We = cw + 2*(cw/2) (may be NULL); float64, or float32 with RR_OUT_ENV_F32 */
  uint8_t* env_bgr_u8;            /* H*We*3: the map the reference saves with --save_envmap (may be NULL) */
  int32_t out_types;              /* one value for every frame of a batch */
  int32_t reserved;
} rr_prepass_out;

int rr_set_prepass_kernels(rr_ctx* ctx, const rr_prepass_kernels* k);

/* Projection tables of EnvironmentMapGenerator for HxW frames (bad_weather.py:716-762): `uniq` are the
 * n_uniq distinct cylinder cells row*cw+col in ascending order and `first` the source pixel (row*W+col)
 * np.unique(..., return_index=True) pairs with each of them.  Host pointers; copied. */
int rr_set_envmap_geometry(rr_ctx* ctx, int32_t H, int32_t W, int32_t cw, int32_t n_uniq, const int32_t* uniq,
                           const int32_t* first);
int rr_envmap_width(rr_ctx* ctx);                    /* We for the geometry set above, or < 0 */

/* n frames of identical H,W.  DEVICE pointers, enqueued on `stream` (NULL = ctx stream). */
int rr_prepass_frames_device(rr_ctx* ctx, int32_t n, const rr_prepass_in* in, const rr_prepass_out* out, void* stream);
/* HOST pointers: upload, run, download, return after completion. */
int rr_prepass_frames(rr_ctx* ctx, int32_t n, const rr_prepass_in* in, const rr_prepass_out* out);

/* Pre-pass + hot path for n frames with HOST pointers and no host round trip in between:
 * in[f].rainy_bg and in[f].env_xyY are ignored (produced on the device from pre[f]); in[f].bg is ignored
 * too (the mean shift uses pre[f]'s background); in[f].He/We must be H / rr_envmap_width().  pre_out may be NULL, as may each of
 * its members: what is non-NULL is downloaded as well.
 * Width of the arrays handed from the pre-pass to the hot path (they never leave the device unless pre_out asks): what
 * pre_out[].out_types says where pre_out downloads them; otherwise float32 (RR_OPT_PIPELINE_F32, default 1: the mask does
 * not read them, the image contract is +-1 LSB) -- the environment map only when every in[f].omega is NULL (the resident
 * solid angles exist in both widths).  A byte image (bg_u8 / RR_IN_BG_U8) stays bytes on the device. */
int rr_pipeline_frames(rr_ctx* ctx, int32_t n, const rr_prepass_in* pre, const rr_frame_in* in, const rr_frame_out* out,
                       const rr_prepass_out* pre_out);

/* ---------------------------------------------------------------------------------------
 * Particle generator on the device (SURVEY 8f "next" #4, BASELINE.json configs[4]: "in-kernel particle simulation (no
 * XML)").  The reference drives a closed-source simulator binary through tools/simulation.py with the settings of
 * common/db.py:41-70 and reads its output back as XML (bad_weather.py:192-211); there is no source to follow, so the
 * model is this library's own (rain-rendering_amd/tools/particles.py is its bit-exact host statement: Marshall-Palmer
 * sizes, Atlas terminal velocity, exposure-integrated streaks, pinhole projection, counter-based Philox4x32-10 numbers).
 * For every frame the device
 *   1. simulates n_particles streaks (one Philox counter per particle: any particle can be made by any lane),
 *   2. applies the loader's derived fields (bad_weather.py:208-241: render scale, y flip, z sign, widths, ratio, rounding,
 *      length, type) and the frame filter of Generator.run (generator.py:413-420), keeping the particle order,
 *   3. makes the renderer's per-drop random draws (np.random.seed(draw_seed), one randint per drop, one normal per
 *      non-Big drop: bad_weather.py:252-264, generator.py:136) from numpy's legacy MT19937 stream, like rr_host_frame_draws,
 * and leaves rr_drop[] records in HBM: no XML, no host drop table, no PCIe traffic.  Angular noise (--noise_std) is not
 * offered on this path (the reference's default is 0); the rotation terms are rot_cos = -dy / n, rot_sin = -|dx| / n
 * (n = |end - start|): the values cos / sin(-(theta) * pi / 180), theta = acos(-dy / n), take when evaluated exactly.
 * Everything derived from transcendentals (the diameter distribution) is tabulated once by the host. */
typedef struct rr_sim_frame {
  int32_t sensor_w, sensor_h;     /* cam_CCD_WH (db.py): the simulator's sensor in pixels; rendered frame = sensor / render_scale */
  int32_t render_scale;           /* settings["render_scale"] (bad_weather.py:208-211) */
  int32_t n_particles;            /* streaks simulated for this frame (host: Poisson(expected count), or a fixed count) */
  uint32_t key0, key1;            /* Philox key: the simulation's seed */
  uint32_t frame;                 /* simulated frame number (a word of the Philox counter) */
  uint32_t draw_seed;             /* np.random.seed(...) of the renderer's per-drop draws (generator.py:318) */
  int32_t table;                  /* diameter table of this frame's fall rate / camera (rr_set_particle_tables) */
  int32_t reserved;
  double fpx;                     /* focal length / pixel size (cam_focal, cam_CCD_pixsize) */
  double exposure_s;              /* cam_exposure / 1000 */
  double speed_mps;               /* sim_steps["cam_motion"] / 3.6 */
  double wind_sigma;              /* m/s, horizontal */
  double margin;                  /* the simulated field exceeds the sensor by this fraction on every side */
  double min_px;                  /* narrowest streak simulated, pixels */
  double z_far;                   /* farthest drop simulated, metres */
} rr_sim_frame;

/* n_tables inverse-CDF tables of the drop diameter (mm): d_grid[n_grid] ascending, cdf[t][n_grid] from 0 to 1,
 * non-decreasing (host pointers, copied).  A diameter is d_grid[j] + (u - cdf[j]) * ((d_grid[j+1] - d_grid[j]) / (cdf[j+1] - cdf[j]))
 * for the j with cdf[j] <= u < cdf[j+1]. */
int rr_set_particle_tables(rr_ctx* ctx, int32_t n_tables, int32_t n_grid, const double* d_grid, const double* cdf);

/* n frames of rendered size H x W: frame f's records go to drops_out + f * cap (DEVICE memory, cap records per frame; what
 * does not fit is not stored) and its drop count to n_out[f] (DEVICE, may exceed cap).  `frames` is a HOST array.  Needs
 * the streak database (texture ratios).  Enqueued on `stream` (NULL = the ctx stream); returns without waiting. */
int rr_generate_drops_device(rr_ctx* ctx, int32_t n, const rr_sim_frame* frames, int32_t H, int32_t W, rr_drop* drops_out,
                             int32_t cap, int32_t* n_out, void* stream);
/* Same with HOST output buffers (n * cap records, n counts): generates on the device, downloads, returns after completion.
 * (A driver that wants the generated particles as data, and the tests.) */
int rr_generate_drops(rr_ctx* ctx, int32_t n, const rr_sim_frame* frames, int32_t H, int32_t W, rr_drop* drops_out, int32_t cap,
                      int32_t* n_out);
int rr_sizeof_sim_frame(void);

/* Options.  1-4 and 6 are tuning / A-B switches: NONE of them changes a result bit (tests/test_gpu_properties.py); 7 trades
 * float64 for float in the colour channels only.  Unknown
 * options or values are RR_E_ARG.  The library reads no environment variables. */
enum {
  RR_OPT_DEDUP = 1,                 /* 1 (default): drops with bit-identical raw-tile parameters share one tile inside a batch */
  RR_OPT_GENERAL_FOV = 2,           /* 1: force the general colour path (prefix table in HBM) that maps taller than 1024 rows,
                                     *    wider than 4096 columns or with He*We >= 2^22 always take; default 0 */
  RR_OPT_FOV_THREADS = 3,           /* workgroup size of the FOV-sum kernel: 0 (library's choice), 512 or 1024 */
  RR_OPT_FOV_DROPS_PER_THREAD = 4,  /* drops per thread of the FOV-sum kernel: 0 (library's choice), 1, 2, 4 or 8 */
  /* NOT a tuning switch -- a feature the reference only sketches (common/drop_depth_map.py, dead code behind
   * USE_DEPTH_WEIGHTING = 0, generator.py:20): with 1, a drop is not composited at pixels whose scene depth
   * (rr_frame_in.depth / the pre-pass' depth) is smaller than the drop's distance |world z|.  Default 0: the reference's
   * output.  Excluded from every parity run. */
  RR_OPT_DEPTH_OCCLUSION = 5,
  RR_OPT_BLUR_WORKGROUPS = 6,       /* tuning: workgroups per CU the fused defocus blur is sized for (LDS tiles + registers):
                                     * 0 (library's choice = 4), 3, 4 or 5 */
  /* Colour arithmetic of the compositor.  rainy_mask is a float64 sum in drop order in either case (bit-exact); the three
   * colour channels of rainy_image only have to land within 1 LSB of a uint8 (BASELINE.json), so by default they are
   * blended in float whenever no frame of the batch asks for the float64 composite (rr_frame_out.rainy_bg_out == NULL).
   * 1: float64 colours always (the reference's arithmetic; what rainy_bg_out != NULL gets anyway).  The uint8 image of
   * the two differs by at most 1 LSB (tests/test_gpu_properties.py). */
  RR_OPT_COMPOSITE_F64 = 7,
  /* tuning: how the host-pointer entry points move a batch across PCIe.  0 (default): hipMemcpyAsync (the DMA engines),
   * after merging neighbouring pieces -- frames laid out back to back in one rr_host_alloc block, each starting on a
   * 16-byte boundary, travel as ONE copy per array (so a gap of fewer than 16 bytes between two consecutive frames'
   * arrays of one kind inside such a block is treated as padding: uploads read it, downloads may overwrite it).  1: the pieces whose host side is page-locked and 16-byte aligned are
   * moved by one copy kernel per direction that reads / writes the host memory directly (faster than many small DMA
   * requests -- 90 vs 40 GB/s both ways, scripts/probes/pcie_probe.hip -- but it takes compute units from the rendering
   * kernels it runs beside: measured slower end to end). */
  RR_OPT_COPY_KERNELS = 8,
  /* tuning: 1 (default) the tile kernels stage a streak texture from a copy that already carries its 2-texel zero border
   * (made once by rr_set_streak_db*: 16-byte copies into LDS); 0 they build the border and place the texels byte by byte.
   * The LDS contents are the same bytes. */
  RR_OPT_PADDED_TEXTURES = 9,
  /* precision of the colour branch: 0 float64 throughout (the reference's arithmetic), 1 float32 always, 2 (default) float32
   * whenever the compositor blends float colours (no frame of the batch asks for the float64 composite: the same switch as
   * RR_OPT_COMPOSITE_F64's default).  Float32 = the field-of-view vertices (every predicate that decides a drop's status or
   * the wrap structure of its polygon is checked against an error bound; a drop that comes close is evaluated in float64)
   * and the environment-map sums under the polygon.  They only feed the drop's colour constants, which only scale
   * rainy_image (contract: +-1 LSB; the mask and the drop statuses never see the difference). */
  RR_OPT_FOV_F32 = 10,
  RR_OPT_FOV_DDA = 12,              /* tuning: the float colour branch evaluates a drop's field-of-view polygon and its row spans with one
                                     * thread per drop (two cursors down the polygon's sides; wrapping polygons and float64 decisions
                                     * through a list to the edge-parallel kernel) -- 1 (default): an exact division per cursor and row;
                                     * 2 (r05): incremental cursors over per-edge records (a quarter of the instructions per row, no
                                     * faster: profiles/r05_ab_log.md); 0: the edge-parallel kernel for every drop.  The spans are the
                                     * same: identical results. */
  RR_OPT_COMPOSITE_WAVES = 11,      /* tuning: waves per SIMD the float compositor's register allocation is held to: 0 (library's
                                     * choice), 4..8 (more waves in flight hide more of the alpha-sample latency; 4 and 5 only with
                                     * RR_OPT_COMPOSITE_BATCH) */
  RR_OPT_PIPELINE_F32 = 13,         /* 1 (default): rr_pipeline_* hand the fog layer and the xyY map from the pre-pass to the hot path
                                     * as float32 unless pre_out asks for float64 copies (see rr_pipeline_frames); 0: float64 */
  RR_OPT_WILD_PIXELS = 14,          /* 1: rainy_bg may hold values outside [0, 1] (a third party's array; the fog pre-pass ends with a
                                     * clip, so its output never does).  The reference blends a drop over its whole padded rectangle
                                     * (bad_weather.py:429-446): where the drop image is zero that is np.clip(pixel, 0, 1), a no-op for
                                     * values in [0, 1] -- the library never visits the pad.  With this option the pads are tracked
                                     * (two int32 per pixel, one more kernel) and a pixel some pad reaches before any tile is clipped
                                     * first: the reference's result for any input.  Default 0.  Not with rr_ext_tile. */
  RR_OPT_PNG_DEFLATE = 15,          /* 1: rr_frame_out.rainy_png / mask_png are entropy-coded on the device.  The buffer (same size) then
                                     * starts with 16 bytes {'R','R','Z','1', uint32 length L, 0, 0} followed by the L bytes of the file's
                                     * zlib stream (one dynamic-Huffman deflate block per 32 KB of scanlines; any inflate reads it):
                                     * the IDAT payload as it is.  A file whose stream would not fit its buffer (incompressible pixels)
                                     * keeps its scanlines (first byte = a filter type, never 'R').  rr_png_write_scanlines /
                                     * rr_io_write_frames take either form.  Default 0 (scanlines). */
  RR_OPT_COMPOSITE_U16 = 16,        /* tuning (r05): 1 (default) the float compositor leaves the composite before the mean shift in the
                                     * library's scratch as three 16-bit codes in one 8-byte word per pixel (rint(v * 65534); 65535 = "outside [0, 1]:
                                     * take the pixel's own rainy_bg value", which is then what the composite holds) instead of three
                                     * floats: one store per pixel, two thirds of the bytes written there and read back by the final pass.  The code is 2^-17 off at
                                     * most (an LSB of rainy_image is 2^-8): the image contract (+-1 LSB) holds, the mask never sees it.
                                     * Ignored with RR_OPT_WILD_PIXELS, and whenever a caller asks for the composite itself. */
  RR_OPT_BLUR_DMA = 17,             /* tuning (r05): 1 (default) the fused defocus blur stages its raw sub-tiles and weight tables with
                                     * gfx950 LDS-DMA loads (global_load_lds: no registers in between), issued a sub-tile AHEAD: they land
                                     * while the current sub-tile's column pass runs; 0: the r04 kernel (loads through registers at the
                                     * start of every sub-tile) -- since r06 only in -DRR_EXPERIMENTS builds of the library, RR_E_ARG
                                     * otherwise.  Same results. */
  RR_OPT_FOV_FILL_RULE = 18,        /* which restatement of cv2.fillConvexPoly (bad_weather.py:388) decides the texels of a drop's field of
                                     * view: 0 (default) the row-span rule of the fast colour kernels (nearest x of every edge on the
                                     * row, min / max); 1 OpenCV 3.2's own algorithm -- Bresenham outline + 16.16 edge walkers, in closed
                                     * form per edge and row (csrc/rr_device.h fov_rowspan_cv) -- for the closed polygons it is defined
                                     * for; takes the general (slow) colour path.  Colour only: a drop's colour constants move by
                                     * <= 0.3 %, rainy_image by <= 1 LSB on 1.4 % of its values (README, profiles/r05_fill_rule_study.txt);
                                     * mask and statuses are the same. */
  RR_OPT_BIN_ROWS = 19,             /* tuning (r05): 1 (default) the ordered per-tile drop lists are made by a workgroup per ROW of coarse
                                     * tiles (drops filtered by row first, then a wave per tile); 0: a workgroup per coarse tile that
                                     * tests every drop (r04).  Same lists. */
  RR_OPT_COLOUR_STREAM = 21,        /* tuning (r05): two chains of the step that only meet in k_colour run on two streams of the library.
                                     * 1 (default): the FOV chain (polygons, spans, sums over the environment map) on the second stream
                                     * beside plan .. tiles .. blur; 0: one in-order stream (r04).  (2 was the other split -- plan .. lists
                                     * and k_colour on the second stream -- measured slower in r05 and removed in r06: it runs as 1.)
                                     * Same results (with 1 a drop without a FOV polygon gets its raw tile rendered for nothing -- it is
                                     * still not blended and keeps its status -- and such tiles take arena space: the arena grows to the
                                     * batches' own demand, RR_E_ARENA, so this only shows as a slightly larger arena). */
  RR_OPT_TILE_ROWS = 22,            /* tuning (r06): 1, 2 rotate + flip + INTER_AREA tiles (Medium / Small drops, generator.py:163-170)
                                     * are rendered by k_tile_rows -- the batch's tiles in one list bucketed by texture, a wave per tile,
                                     * a lane per canvas row, horizontal folds in registers; 0: k_tile (a workgroup per tile).  2 (default):
                                     * the Big drops' bicubic warps (generator.py:126-132) ride in the same list, their zero-bordered texture
                                     * resident in LDS, a lane per pixel; 1: those stay with k_tile_big (a thread per pixel of a frame's
                                     * concatenated tiles, texels from global memory).  Same bits. */
  RR_OPT_ROWS_SHARES = 23,          /* tuning (r06): k_tile_rows' workgroups take the batch's tile list in shares of equal estimated cost off a
                                     * device-wide counter; this many shares per workgroup (1 .. 8, default 2): more shares even out the
                                     * workgroups, fewer leave less waiting at a share's end */
  RR_OPT_COMPOSITE_BATCH = 20       /* tuning (r05): 1 (default) the float compositor keeps the records of 64 list entries at a time in
                                     * vector registers (a lane per entry) and runs its alpha samples two entries ahead of the blend;
                                     * 0: a scalar record fetch per entry, samples one entry ahead (r04).  Same operations in the same
                                     * order: same bits. */
};
int rr_set_option(rr_ctx* ctx, int32_t option, int32_t value);

/* Asynchronous form of rr_pipeline_frames (pre may be NULL: then it is the asynchronous rr_render_frames): up to
 * RR_PIPE_SLOTS batches in flight.  The upload of one batch, the kernels of another and the download of a third
 * overlap (three streams inside the library).  Buffers must stay valid and untouched until rr_pipeline_wait(slot)
 * returns; buffers from rr_host_alloc (pinned) move at PCIe rate, pageable ones serialise the copies.
 * rr_pipeline_wait returns RR_E_ARENA when the tile arena had to grow: submit that batch again. */
#define RR_PIPE_SLOTS 3
int rr_pipeline_submit(rr_ctx* ctx, int32_t slot, int32_t n, const rr_prepass_in* pre, const rr_frame_in* in,
                       const rr_frame_out* out, const rr_prepass_out* pre_out);
int rr_pipeline_wait(rr_ctx* ctx, int32_t slot);
/* Page-locked host memory (hipHostMalloc) for the buffers of the host-pointer entry points.  A block belongs to its
 * context: rr_destroy releases whatever rr_host_free has not. */
int rr_host_alloc(rr_ctx* ctx, void** out, int64_t bytes);
/* rr_host_alloc / rr_host_free may run on another thread than the context's calls, so their failures are not reported
 * through rr_last_error(): this returns the message of the last failed rr_host_alloc ("" after a successful one); the
 * pointer stays valid until the next rr_host_alloc on the context. */
const char* rr_host_last_error(rr_ctx* ctx);
int rr_host_free(rr_ctx* ctx, void* p);

/* Work-list sizes of frame `frame` of the last batch (after completion): out[0] drops whose raw tile went
 * through the rotate+resize kernels (k_tile_rows + k_tile), [1] through the generic kernel, [2] fused-blur work items, [3] slow-blur
 * drops, [4] small-blur drops, [5] Big drops (bicubic warp kernel), [6] their pixels, [7] drops that re-used
 * another drop's bit-identical raw tile (k_dedup). */
int rr_batch_counts(rr_ctx* ctx, int32_t frame, int32_t out[8]);

/* Per-kernel timing with HIP events on the launch stream (off by default). */
int rr_profile_enable(rr_ctx* ctx, int32_t on);
int rr_profile_reset(rr_ctx* ctx);
int rr_profile_read(rr_ctx* ctx, rr_kernel_stat* out, int32_t cap);   /* returns #entries or <0 */

/* Host-only helper (no device, no ctx): the random draws of one frame's drop loop, bit-identical to
 * numpy's legacy global RandomState after np.random.seed(seed) (generator.py:318): per drop one
 * randint(tex_lo[k], tex_lo[k]+10) (bad_weather.py:252-264) and, for non-Big drops, one
 * normal(0, noise_std) (generator.py:136; noise[k] is the raw deviate, 0 for Big drops).  Lets a
 * driver prepare frames on worker threads instead of the process-global generator. */
int rr_host_drop_draws(uint32_t seed, int32_t n, const int32_t* tex_lo, const uint8_t* is_big, double noise_std,
                       int32_t* tex_index, double* noise);

/* Host-only (no device, no ctx, no Python interpreter lock): one frame's drop table from the column store of its
 * simulated frame.  rr_host_frame_draws applies the frame filter (common/generator.py:413-420), picks the texture block
 * (bad_weather.py:250-265; ratio_db = DBManager.ratio, >= 4 entries) and makes the frame's draws in the reference's order;
 * returns the number of kept streaks (<= t->n; keep / tex_index / noise need that capacity), noise = raw deviate * 1.
 * rr_host_assemble_drops writes the rr_drop records once the caller has evaluated cos / sin of the rotation. */
typedef struct {
  int64_t n;
  const double *wps, *wpe;        /* [n][3] world_position_start / _end */
  const int64_t *ips, *ipe;       /* [n][2] image_position_start / _end (integers) */
  const double *iw1, *iw2, *ratio;
  const int64_t *max_width, *length;
  const int32_t* type;            /* DropType */
} rr_streak_table;
int64_t rr_host_frame_draws(const rr_streak_table* t, int32_t W, int32_t H, const double* ratio_db, int32_t n_ratio,
                            uint32_t seed, double noise_std, int64_t* keep, int32_t* tex_index, double* noise);
int rr_host_assemble_drops(const rr_streak_table* t, int64_t n_keep, const int64_t* keep, const int32_t* tex_index,
                           const double* rot_cos, const double* rot_sin, rr_drop* out);
int rr_sizeof_streak_table(void);
/* A whole batch of drop tables in one call, for runs without angular noise (the reference's default): frame k = the filter,
 * draws and records of rr_host_frame_draws + rr_host_assemble_drops on tables[k] with seed seeds[k].  Without noise the
 * rotation terms belong to the table ENTRY: rot_cos[k] / rot_sin[k] hold them for every entry of tables[k] (the caller's
 * numpy evaluates them once per simulated frame; frames that share a table pass the same arrays).  Frame k's records go to
 * out + k * out_stride (records), at most cap of them; counts[k] = the number of kept streaks (it may exceed cap: size up
 * and call again) or a negative RR_E* code.  Runs on `threads` worker threads inside the library (<= 0: up to 16). */
int rr_host_pack_frames(int32_t n, const rr_streak_table* const* tables, const double* const* rot_cos, const double* const* rot_sin,
                        int32_t W, int32_t H, const double* ratio_db, int32_t n_ratio, const uint32_t* seeds, rr_drop* out,
                        int64_t out_stride, int64_t cap, int32_t threads, int64_t* counts);

/* Host-only (no device, no ctx): the particles XML of the rain simulator -> flat records, replacing the reference's
 * pure-Python walk (common/bad_weather.py:192-211).  Raw attribute values only; the derived fields (render scale, y flip,
 * z sign, widths, ratio, rounding, the pid dictionary; bad_weather.py:208-241) stay with the caller.  Counts are always
 * returned; records beyond the capacities are not stored (call again with larger arrays). */
typedef struct {
  int64_t id, t, d, rs;           /* frame attributes id, t (exposure), d (start time), rs (count) */
  int64_t first_drop, n_drops;    /* its drops are records [first_drop, first_drop + n_drops) */
} rr_particle_frame;
typedef struct {
  int64_t pid;
  double wp1[3], wp2[3], wd1, wd2, ip1[2], ip2[2], iw1, iw2;
} rr_particle;                    /* 120 bytes */
int rr_host_parse_particles(const char* path, rr_particle_frame* frames, int64_t cap_frames, rr_particle* drops,
                            int64_t cap_drops, int64_t* n_frames, int64_t* n_drops);
int rr_sizeof_particle(void);
int rr_sizeof_particle_frame(void);

/* Host-only PNG codec for the driver's I/O threads (rr_png.cpp; zlib inside; no interpreter lock when called through
 * ctypes).  Readers: non-interlaced gray / RGB / palette / RGBA files, else RR_E_UNSUPPORTED (use a general decoder).
 *   rr_png_read_bgr8    what cv2.imread(path) returns for an 8-bit file: H*W*3 bytes, B G R      (generator.py:352)
 *   rr_png_read_gray16  cv2.imread(path, IMREAD_UNCHANGED) of a 16-bit single-channel file       (generator.py:360-365)
 *   rr_png_write_scanlines  an RGBA file from H rows of 1 + 4*W filtered bytes (rr_frame_out.rainy_png / mask_png);
 *                           zlib level 0..9, strategy 0 default / 1 Z_RLE / 2 Z_HUFFMAN_ONLY / 3 the library's own
 *                           run-length + dynamic-Huffman deflate (level ignored; an ordinary zlib stream, about the size
 *                           of Z_RLE's at a fifth of the CPU time); rows that begin with 'R','R','Z','1' are a stream the
 *                           device made (RR_OPT_PNG_DEFLATE) and go out as the IDAT payload as they are
 *   rr_deflate_fast     that encoder on n arbitrary bytes -> zlib stream in out (capacity >= rr_deflate_bound(n));
 *                       returns the stream's length or a negative RR_E* code */
int rr_png_info(const char* path, int32_t* w, int32_t* h, int32_t* channels, int32_t* bit_depth);
int rr_png_read_bgr8(const char* path, uint8_t* out, int32_t H, int32_t W);
int rr_png_read_gray16(const char* path, uint16_t* out, int32_t H, int32_t W);
int rr_png_write_scanlines(const char* path, const uint8_t* rows, int32_t W, int32_t H, int32_t level, int32_t strategy);
int64_t rr_deflate_bound(int64_t n);
int64_t rr_deflate_fast(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap);
/* The readers' own inflate (64-bit bit buffer, 12-bit table look-ups that yield two literals where both codes fit, 8-byte
 * match copies; 2-3 times zlib's speed on image scanlines) on a complete
 * zlib stream of known decoded size: 1 = out holds the data (Adler-32 verified), 0 = not vouched for (the readers then
 * hand the stream to zlib), < 0 = bad argument. */
int rr_inflate_fast(const uint8_t* in, int64_t n, uint8_t* out, int64_t out_len);
/* Batch forms for the driver, on `threads` worker threads inside the library (<= 0: up to 16); one call per pipeline batch
 * instead of one interpreter call per frame and file.  The calls return RR_OK unless an argument is bad; status[k] holds
 * frame k's own result (RR_OK, RR_E_ARG: size mismatch / unreadable path, RR_E_UNSUPPORTED, RR_E_PARSE).
 *   rr_io_read_frames   frame k: rr_png_read_bgr8(image_paths[k]) into bg_u8 + k * bg_stride (bytes) and, when depth_paths
 *                       is given, the 16-bit depth file as float32 metres (value / 256, generator.py:360-365) into
 *                       depth_f32 + k * depth_stride (bytes); both files must be H x W
 *   rr_io_write_frames  frame k: rr_png_write_scanlines(strategy 3) of rows_image + k * rows_stride to image_paths[k] and of
 *                       rows_mask + k * rows_stride to mask_paths[k] (either path array may be NULL) */
int rr_io_read_frames(int32_t n, const char* const* image_paths, const char* const* depth_paths, int32_t H, int32_t W,
                      uint8_t* bg_u8, int64_t bg_stride, float* depth_f32, int64_t depth_stride, int32_t threads, int32_t* status);
/*   rr_io_read_frames_u16  the same with the depth file's uint16 samples as they are (rr_prepass_in.depth_f64 = RR_DEPTH_U16) into
 *                       depth_u16 + k * depth_stride (bytes): no conversion pass on the host, half the bytes over PCIe */
/*   rr_io_read_frames_rows  both files of a frame INFLATED ONLY: image_rows + k * image_stride receives H rows of 1 + 3*W bytes (filter
 *                       type + filtered R G B bytes) of the 8-bit RGB image, depth_rows + k * depth_stride H rows of 1 + 2*W
 *                       bytes of the 16-bit gray depth file -- what rr_prepass_in takes with RR_IN_BG_PNG_ROWS /
 *                       RR_DEPTH_PNG_ROWS: the scanline filters are reversed on the device (csrc/rr_pngrows.h), 40 % of the
 *                       host's decode time.  A file of another kind that the readers above accept is decoded on the host
 *                       and handed over as rows of filter type 0. */
int rr_io_read_frames_rows(int32_t n, const char* const* image_paths, const char* const* depth_paths, int32_t H, int32_t W,
                           uint8_t* image_rows, int64_t image_stride, uint8_t* depth_rows, int64_t depth_stride, int32_t threads,
                           int32_t* status);
int rr_io_read_frames_u16(int32_t n, const char* const* image_paths, const char* const* depth_paths, int32_t H, int32_t W,
                          uint8_t* bg_u8, int64_t bg_stride, uint16_t* depth_u16, int64_t depth_stride, int32_t threads, int32_t* status);
/*   rr_io_read_frames_scaled  the same for a render scale other than 1 (generator.py:352-381, the Cityscapes plug-in's
 *                       default): image / 255 resized (cv2.resize, INTER_LINEAR) to W x H = file size // render_scale as
 *                       float64 into bg_f64 + k * bg_stride (bytes), the depth map resized to
 *                       (its size * depth_scale) // render_scale when that differs from its own size; both must come out
 *                       H x W (RR_E_ARG otherwise: the reference crops the image then -- the caller's general loader) */
int rr_io_read_frames_scaled(int32_t n, const char* const* image_paths, const char* const* depth_paths, int32_t H, int32_t W,
                             int32_t render_scale, int32_t depth_scale, double* bg_f64, int64_t bg_stride, float* depth_f32,
                             int64_t depth_stride, int32_t threads, int32_t* status);
int rr_io_write_frames(int32_t n, const char* const* image_paths, const char* const* mask_paths, const uint8_t* rows_image,
                       const uint8_t* rows_mask, int64_t rows_stride, int32_t W, int32_t H, int32_t threads, int32_t* status);
/* The codec's checksums: zlib's adler32(adler, p, n) and crc32(crc, p, n) (same values, same chaining; start from 1 and 0),
 * computed with SSSE3 / carry-less multiplication where the CPU has them (zlib's own loops otherwise). */
uint32_t rr_adler32(uint32_t adler, const uint8_t* p, int64_t n);
uint32_t rr_crc32(uint32_t crc, const uint8_t* p, int64_t n);

/* sizes, for binding self-checks */
int rr_sizeof_drop(void);
int rr_sizeof_camera(void);
int rr_sizeof_frame_in(void);
int rr_sizeof_frame_out(void);
int rr_sizeof_prepass_in(void);
int rr_sizeof_prepass_out(void);
int rr_sizeof_prepass_kernels(void);

#ifdef __cplusplus
}
#endif
#endif /* RAINHIP_H */
