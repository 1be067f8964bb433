/*
 * glimpse_hip.h -- C ABI of libglimpse_hip.so, the MI355X (gfx950) implementation of
 * the glimpse.Tracker particle-filter hot path.
 *
 * The reference (ezwelty/glimpse 0.1.1) is pure Python and has NO FFI layer; its
 * replaceable seams are Python duck types.  This header is
 * therefore the interface a maintainer would bind with ctypes;
 * every entry point cites the reference function whose work it takes over.
 * Citations are relative to /root/reference/src/glimpse/.
 *
 * Conventions
 *  - plain C: opaque context, raw pointers and sizes, no C++/torch types;
 *  - every function returns an int status: GLH_OK (0) or a negative GLH_E_* code;
 *    `glh_last_error()` returns a human-readable message for the calling thread;
 *  - host buffers are caller-owned, device buffers are library-owned;
 *  - one host thread per context; every call enqueues on the context's HIP stream,
 *    only `glh_get_*`, `glh_sync` and the `glh_stage_*` test hooks block;
 *  - array layouts are C-contiguous, doubles unless stated:
 *      particles [P][N][6]  (x, y, z, vx, vy, vz)   -- Tracker.particles, track/tracker.py:35
 *      weights   [P][N]                              -- Tracker.weights,   track/tracker.py:37
 *      cameras   [n][GLH_CAM_LEN]                    -- Camera._vector[20] (camera.py:101) + correction
 *      moments   [P][12] = mean(6) | sigma(6)        -- track/tracker.py:350-354
 */
#ifndef GLIMPSE_HIP_H
#define GLIMPSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLH_VERSION 100

/* ---- status codes ------------------------------------------------------------------ */
#define GLH_OK 0
#define GLH_E_INVALID (-1)   /* bad argument / size mismatch                              */
#define GLH_E_HIP (-2)       /* a HIP runtime call failed                                 */
#define GLH_E_NOMEM (-3)     /* host or device allocation failed                          */
#define GLH_E_STATE (-4)     /* call sequence error (e.g. step before templates)          */
#define GLH_E_UNSUPPORTED (-5)
#define GLH_E_COMM (-6)      /* librccl missing, or an RCCL call failed                   */

/* ---- camera vector ------------------------------------------------------------------ */
/* [0:3] xyz  [3:6] viewdir(deg)  [6:8] imgsz  [8:10] f  [10:12] c  [12:18] k1..k6
 * [18:20] p1,p2   -- exactly Camera._vector (camera.py:101, :128-198) --
 * [20] correction flag  [21] radius  [22] refraction (camera.py:118-121)
 * [23] 0 = camera.  1 = georeferenced raster image (an orthophoto observer, Raster as image,
 *      track/observer.py:26): world -> image is Grid.xyz_to_uv (raster.py:423-445),
 *      uv = (xy - (xlim[0], ylim[0])) / d, with [0:2] = (xlim[0], ylim[0]), [6:8] = size,
 *      [8:10] = d (signed cell size); every other entry is ignored.                            */
#define GLH_CAM_LEN 24

/* ---- motion parameters (CartesianMotion, track/motion.py:121-147) ------------------- */
/* [0:2] xy [2:4] xy_sigma [4:7] vxyz [7:10] vxyz_sigma [10:13] axyz [13:16] axyz_sigma
 * [16] dem (constant surface) [17] dem_sigma (constant)                                   */
#define GLH_MOTION_LEN 18

/* ---- motion parameters, general form (glh_set_motion) -------------------------------- */
/* [0:18] as above, read per model kind:
 *   CARTESIAN            (motion.py:92-204)   vxyz, vxyz_sigma, axyz, axyz_sigma
 *   CYLINDRICAL          (motion.py:207-311)  [4:7] vrthz [7:10] vrthz_sigma [10:13] arthz [13:16] arthz_sigma
 *   TANGENT_CARTESIAN    (motion.py:314-412)  [4:6] vxy [7:9] vxy_sigma [10:12] axy [13:15] axy_sigma
 *   TANGENT_CYLINDRICAL  (motion.py:415-522)  [4:6] vrth [7:9] vrth_sigma [10:12] arth [13:15] arth_sigma
 * [18] kind (GLH_MOTION_*)  [19] slope_sigma (tangent models)
 * [20] 1 = this point's dem is the context's GLH_RASTER_DEM raster (then [16] is ignored)
 * [21] 1 = this point's dem_sigma is the GLH_RASTER_DEM_SIGMA raster  [22:24] reserved.       */
#define GLH_MOTION_FULL_LEN 24
#define GLH_MOTION_CARTESIAN 0
#define GLH_MOTION_CYLINDRICAL 1
#define GLH_MOTION_TANGENT_CARTESIAN 2
#define GLH_MOTION_TANGENT_CYLINDRICAL 3
#define GLH_MOTION_EXTERNAL 4 /* a user-defined Motion (the duck type of motion.py:13-89): the caller initialises and
                               * evolves the particles (glh_set_particles) and supplies its log-likelihood term
                               * (glh_set_extra_log_likelihoods); the device does the observer likelihoods,
                               * resampling and moments                                                          */

/* ---- per-point status bits (sticky; the Python Tracker turns them into Tracks.errors) */
#define GLH_PT_NAN 1u            /* ValueError "missing (NaN) values"      tracker.py:118  */
#define GLH_PT_TEMPLATE_OOB 2u   /* IndexError "Box extends beyond grid"   raster.py:417   */
#define GLH_PT_SAMPLE_OUTSIDE 4u /* ValueError "sampling points outside"   observer.py:201 */
#define GLH_PT_RESAMPLE_CLAMP 8u /* searchsorted returned n (IndexError in the reference)  */
#define GLH_PT_CONST_TILE 16u    /* zero-variance template tile (reference yields NaNs)    */
#define GLH_PT_RASTER_OOB 32u    /* ValueError "sampling coordinates are out of bounds" raster.py:961-973 */
#define GLH_PT_NOT_VISIBLE 64u   /* ValueError "non-visible viewshed cells"  tracker.py:114-117 */

/* ---- per-(observer, point) status of the last likelihood evaluation ------------------ */
#define GLH_OBS_OK 0
#define GLH_OBS_SKIPPED 1      /* img is None or observer masked         tracker.py:577    */
#define GLH_OBS_OUT_OF_BOUNDS 2 /* warning + skip                        tracker.py:597-601 */
#define GLH_OBS_TILE_TOO_LARGE 3 /* search tile exceeds the workspace (build limit)        */
#define GLH_OBS_NO_TEMPLATE 4

/* ---- resampling methods (track/tracker.py:151-223) ------------------------------------- */
#define GLH_RESAMPLE_SYSTEMATIC 0 /* one uniform per point          tracker.py:168-176        */
#define GLH_RESAMPLE_STRATIFIED 1 /* one uniform per particle       tracker.py:178-186        */
#define GLH_RESAMPLE_CHOICE 2     /* np.random.choice(n, n, p=w)    tracker.py:205-209        */
#define GLH_RESAMPLE_RESIDUAL 3   /* as written in the reference     tracker.py:188-203        */

/* ---- random-number modes ------------------------------------------------------------- */
#define GLH_RNG_HOST 0   /* caller supplies the normals / uniforms (parity with np.random)  */
#define GLH_RNG_PHILOX 1 /* counter-based Philox4x32 (7 rounds) on the device               */

/* ---- arithmetic modes (glh_set_math) --------------------------------------------------- */
#define GLH_MATH_EXACT 0 /* every float64 expression rounds like NumPy's (default): with host-fed draws the resample
                          * indices are the reference's bit for bit                                           */
#define GLH_MATH_FAST 1  /* same formulas with fused multiply-adds, Newton reciprocals instead of IEEE divisions, a
                          * table exp and no normalisation pass in the systematic resampling: ~1e-13 relative on the
                          * posteriors, 10-15 % faster; meant for device-RNG runs, where no reference stream exists to be
                          * bit-exact with.  Small fitted surfaces are sampled in a per-cell power form.  Every motion
                          * model and surface kind has it (their own evolve steps and lookups have one form only).    */

typedef struct glh_ctx glh_ctx;

typedef struct glh_config {
  int32_t device_id;      /* HIP device ordinal                                             */
  int32_t max_points;     /* P capacity                                                     */
  int32_t max_particles;  /* N capacity (uniform across points)                             */
  int32_t n_observers;    /* O                                                              */
  int32_t max_tile;       /* largest template side, tile_size <= max_tile (default 31)      */
  int32_t max_search_dim; /* search tiles up to max_search_dim^2 pixels per point           */
  int32_t max_frames;     /* moments history capacity (frames per sequence)                 */
  int32_t reserved;
} glh_config;

/* ---- library / context ---------------------------------------------------------------- */
int glh_version(void);
const char* glh_last_error(void);
int glh_device_count(int* count);
/* Free and total bytes of device `device_id` (hipMemGetInfo): the Python Tracker bounds the growth of its search-tile
   workspaces by them (it re-runs a sequence with larger workspaces when a tile outgrows them, tracker.py has no such
   limit: its tiles live in host memory).                                                                              */
int glh_device_memory(int device_id, uint64_t* free_bytes, uint64_t* total_bytes);
/* Compute units of device `device_id`: what glh_track's automatic choice of streams compares a batch with.           */
int glh_device_compute_units(int device_id, int* count);
int glh_create(const glh_config* cfg, glh_ctx** out);
int glh_destroy(glh_ctx* ctx);
int glh_sync(glh_ctx* ctx);
/* The HIP stream (hipStream_t) the context enqueues on, for event timing by the caller.   */
int glh_get_stream(glh_ctx* ctx, void** stream);

/* ---- observers: images + cameras ------------------------------------------------------ */
/* Observer(images, sigma) (track/observer.py:50-69).  Declares the image list size, the
 * frame geometry and `sigma`; frames are resident in HBM for the whole sequence.          */
int glh_observer_init(glh_ctx* ctx, int obs, int n_images, int width, int height, int channels,
                      double sigma);
/* Sample type of the observer's frames: 8 (default, uint8), 16 (uint16), 32 (float32) or 64 (float64), one or three
 * channels -- Tracker.extract_tile works on any dtype (tracker.py:494-534) and normalises a float tile in the frame's
 * own dtype (a float32 mean / std in NumPy's summation order).  After glh_observer_init, before the first upload; the
 * upload calls then copy width * height * channels * bits / 8 bytes.  16-bit observers run on the fused step while
 * max_search_dim <= 255 (a tile's pixel count is then a 16-bit key), on the staged kernels beyond; float observers on the
 * staged kernels (the pixels at or below every pixel of a tile by a two-level ranking over the tile's value range).     */
int glh_observer_set_depth(glh_ctx* ctx, int obs, int bits);
/* One Camera per image (Image.cam, image.py:110; Camera.R camera.py:239-280 is evaluated
 * on the host in float64 at upload).  cams: [n_images][GLH_CAM_LEN].                      */
int glh_observer_set_cameras(glh_ctx* ctx, int obs, int first_image, int n_images,
                             const double* cams);
/* Image.read() cached array (image.py:180-186): uint8 [height][width][channels].          */
int glh_observer_upload_frame(glh_ctx* ctx, int obs, int image, const uint8_t* pixels);
/* The same without waiting for the device (frame ingest from files: a decoder pool feeds this call).
 * `pixels` is copied to a pinned staging buffer before the call returns; the host-to-device copy runs
 * on a copy stream and every later call that reads frames is ordered after it on the device.      */
int glh_observer_upload_frame_async(glh_ctx* ctx, int obs, int image, const uint8_t* pixels);
/* Frame ingest without the staging copy (round 5): a host buffer the caller registers ONCE -- e.g. the shared-memory ring
 * its decoder processes fill (glimpse_amd/ingest.py) -- is page-locked for the device, and glh_observer_upload_frame_pinned
 * enqueues the host-to-device copy on the copy stream straight from `pixels`, which must lie inside a registered buffer and
 * stay untouched until glh_upload_done reports the copy `*ticket` identifies as finished (`wait` != 0: blocks until it is).
 * Ordering against the kernels is that of glh_observer_upload_frame_async.                                              */
int glh_host_register(void* ptr, uint64_t bytes);
int glh_host_unregister(void* ptr);
int glh_observer_upload_frame_pinned(glh_ctx* ctx, int obs, int image, const uint8_t* pixels, int64_t* ticket);
int glh_upload_done(glh_ctx* ctx, int64_t ticket, int wait, int* done);
/* Same, from a buffer that is already on the device (no PCIe in the timed region).        */
int glh_observer_set_frame_device(glh_ctx* ctx, int obs, int image, const void* dev_pixels);

/* ---- points (tracks) ------------------------------------------------------------------- */
/* Number of tracked points P and particles per point N for this sequence; clears state
 * (Tracker.reset, track/tracker.py:419-423).                                               */
int glh_begin_sequence(glh_ctx* ctx, int n_points, int n_particles, int tile_w, int tile_h);
/* CartesianMotion parameters per point: [P][GLH_MOTION_LEN] (track/motion.py:121-147).     */
int glh_set_motion_cartesian(glh_ctx* ctx, const double* params);
/* Any mix of motion models, one per point: [P][GLH_MOTION_FULL_LEN].  The tangent models return
 * no log likelihood (base Motion.compute_log_likelihoods, motion.py:76-89): a frame on which every
 * observer is skipped leaves their weights unchanged (tracker.py:146-149).  Points that are not
 * CartesianMotion run through the staged kernels.                                             */
int glh_set_motion(glh_ctx* ctx, const double* params);
/* Gridded surfaces (Raster, raster.py:613-): `which` = GLH_RASTER_DEM / _DEM_SIGMA (sampled
 * bilinearly at every particle, Raster.sample order 1, raster.py:913-1027) or _VIEWSHED (nearest
 * cell, order 0, Tracker.test_particles tracker.py:114-117).  z [ny][nx] is Raster.array; gx / gy are
 * the ASCENDING cell-centre coordinates (Grid.x / Grid.y reversed where dx / dy < 0); sx, sy the signs
 * of dx, dy; the limits are Grid.min / Grid.max.  z = NULL removes the raster.  nx, ny >= 2.  The coordinates must be
 * those of a uniform grid over the limits (np.linspace, as Grid makes them) to within a quarter cell:
 * GLH_E_UNSUPPORTED otherwise (the kernels find a sample's cell from the cell size).                  */
#define GLH_RASTER_DEM 0
#define GLH_RASTER_DEM_SIGMA 1
#define GLH_RASTER_VIEWSHED 2
int glh_set_raster(glh_ctx* ctx, int which, const double* z, int nx, int ny, const double* gx,
                   const double* gy, int sx, int sy, double xmin, double xmax, double ymin, double ymax);
/* Global index of this context's point 0 when the tracked points are sharded over several
 * contexts / GPUs (default 0).  The device RNG (GLH_RNG_PHILOX) is keyed on the GLOBAL point
 * index, so a sharded run draws exactly what the unsharded run draws.                        */
int glh_set_point_offset(glh_ctx* ctx, int offset);
/* observer_mask [P][O] (track/tracker.py:250-252, :289-290); NULL = all ones.              */
int glh_set_observer_mask(glh_ctx* ctx, const uint8_t* mask);
/* active [P]: 1 = the point takes part in the following stage calls (frames inside its
 * [first, last] window, track/tracker.py:321-326); NULL = all active.                      */
int glh_set_active(glh_ctx* ctx, const uint8_t* active);

int glh_set_particles(glh_ctx* ctx, const double* particles); /* [P][N][6] */
int glh_get_particles(glh_ctx* ctx, double* particles);
int glh_set_weights(glh_ctx* ctx, const double* weights);
/* A log-likelihood term computed by the caller for the NEXT glh_update_weights calls, ll [P][N] (NULL removes it):
 * Motion.compute_log_likelihoods of a user-defined motion model (motion.py:74-89), appended to the observers' terms
 * like the built-in one (tracker.py:139-149).                                                                    */
int glh_set_extra_log_likelihoods(glh_ctx* ctx, const double* ll); /* [P][N] */
int glh_get_weights(glh_ctx* ctx, double* weights);
int glh_get_point_status(glh_ctx* ctx, uint32_t* status);    /* [P]    GLH_PT_* bits        */
/* Frame index (glh_set_frame) at which each point first raised a status bit, or a large
 * value if none: the reference aborts the track there, so rows >= this frame are NaN
 * (track/tracker.py:360-368).                                                               */
int glh_get_point_error_frame(glh_ctx* ctx, int32_t* frames); /* [P] */
int glh_get_observer_status(glh_ctx* ctx, int32_t* status);
/* The same for frames [frame0, frame0 + n_frames): status [n_frames][O][P] (every frame of a sequence keeps its own
 * status words, so a run of glh_track frames can be inspected afterwards: one warning per skipped image and track,
 * tracker.py:597-601).                                                                                      */
int glh_get_observer_status_frames(glh_ctx* ctx, int frame0, int n_frames, int32_t* status);
/* Particles [N][6] and weights [N] of ONE point (either may be NULL): what the reference's Tracker.particles /
 * .weights hold after the last track (tracker.py:35-37).                                                       */
int glh_get_point_state(glh_ctx* ctx, int point, double* particles, double* weights);  /* [O][P] GLH_OBS_*            */
/* Search boxes (l,t,r,b) of the last glh_update_weights (tracker.py:595): [O][P][4];
 * entries whose observer status is not GLH_OBS_OK are stale.                                */
int glh_get_search_boxes(glh_ctx* ctx, int32_t* boxes);

/* ---- stages of one frame (track/tracker.py:326-357), in the reference's order ---------- */
/* Index i of the datetime being processed (track/tracker.py:326); recorded with errors.     */
int glh_set_frame(glh_ctx* ctx, int frame);
/* CartesianMotion.initialize_particles (track/motion.py:149-163) + initialize_weights
 * (track/tracker.py:121-124).  GLH_RNG_HOST: normals [P][N][6] = randn(n,2)|randn(n)|randn(n,3). */
int glh_init_particles(glh_ctx* ctx, int rng_mode, const double* normals, uint64_t seed);
/* CartesianMotion.evolve_particles (track/motion.py:165-179) + test_particles NaN check
 * (track/tracker.py:118-119).  tau = dt / time_unit.  GLH_RNG_HOST: normals [P][N][3].     */
int glh_evolve(glh_ctx* ctx, double tau, int rng_mode, const double* normals, uint64_t seed,
               uint64_t step);
/* Tracker.initialize_template (track/tracker.py:536-561) for observer `obs` at image
 * `image`, for every active point with the observer enabled: weighted mean -> project ->
 * Grid.snap_box (raster.py:390-421) -> extract_tile (tracker.py:494-534).                 */
int glh_init_templates(glh_ctx* ctx, int obs, int image);
/* Tracker.update_weights (track/tracker.py:126-149): for each observer o with
 * images[o] >= 0, compute_observer_log_likelihoods (tracker.py:563-625); plus
 * CartesianMotion.compute_log_likelihoods (motion.py:181-204); w = exp(-sum) + 1e-300.     */
int glh_update_weights(glh_ctx* ctx, const int32_t* images /* [O], -1 = None */);
/* Tracker.resample_particles("systematic") (track/tracker.py:168-176, :222-223).
 * GLH_RNG_HOST: u [P] = the np.random.random() draw of each point.                         */
int glh_resample(glh_ctx* ctx, int rng_mode, const double* u, uint64_t seed, uint64_t step);
/* Same with an explicit method (GLH_RESAMPLE_*).  GLH_RNG_HOST: u is [P] for systematic (the
 * np.random.random() of each point) and [P][N] for stratified (np.random.random(n)), choice (the n
 * uniforms RandomState.choice draws) and residual (np.random.random(n - sum(repetitions)): the first
 * n - R entries of each row are used; glh_get_residual_draws returns the counts).  Residual follows the
 * reference's arithmetic literally (repetition counts subtracted from normalised weights, then
 * np.searchsorted's stateful bisection over a cumulative sum that is not monotone).            */
int glh_resample_method(glh_ctx* ctx, int method, int rng_mode, const double* u, uint64_t seed,
                        uint64_t step);
/* Tracker.particle_covariance (track/tracker.py:78-82; np.cov(aweights=w, ddof=0)) of every
 * active point into history slot `frame`; glh_get_covariances: out [n_frames][P][36].       */
/* Number of uniforms the last GLH_RESAMPLE_RESIDUAL step consumed per point, n - sum(repetitions)
 * (tracker.py:199-201): draws [P].  Lets a host that feeds np.random keep its stream aligned.   */
int glh_get_residual_draws(glh_ctx* ctx, int32_t* draws);
int glh_record_covariances(glh_ctx* ctx, int frame);
int glh_get_covariances(glh_ctx* ctx, int frame0, int n_frames, double* out);
/* particle_mean + compute_particle_sigma (track/tracker.py:72-76, :89-104) of every active
 * point into history slot `frame` (rows of inactive points keep NaN).                      */
int glh_record_moments(glh_ctx* ctx, int frame);
/* evolve -> update_weights -> resample -> record_moments: the per-frame step i > first of
 * track/tracker.py:331-357 for all active points, enqueued back to back.                   */
int glh_step(glh_ctx* ctx, int frame, double tau, const int32_t* images, int rng_mode,
             const double* normals, const double* u, uint64_t seed);
/* The whole frame loop of every track (track/tracker.py:326-357: `for i, img in enumerate(...)` of
 * process()) in one call: n_frames consecutive glh_step updates with GLH_RNG_PHILOX, enqueued back
 * to back on the context's stream without host synchronisation.  frames [n_frames] = history
 * slots / Philox steps, taus [n_frames] = dt / time_unit of each update, images [n_frames][O]
 * (-1 = None).  Same results as the same glh_step calls.                                       */
int glh_track(glh_ctx* ctx, int n_frames, const int32_t* frames, const double* taus, const int32_t* images,
              uint64_t seed);
/* on = 1: glh_track also records the covariance of the particles after every frame it runs (glh_record_covariances:
 * Tracker.track(return_covariances=True), track/tracker.py:307-308, :352), so that such a run is still one call.   */
int glh_track_covariances(glh_ctx* ctx, int on);
/* Streams of glh_track's frame loop.  The tracks of the reference are independent (track/tracker.py:381-387 hands them
 * to a process pool); here the two halves of a large batch run their frame loops on two HIP streams, so that one
 * half's launch fills the compute units the other half's launch leaves idle while it drains and refills (bit for bit
 * the results of one stream).  0 (default) = automatic: two streams when the batch has more points than the device has
 * compute units (a launch ends with its slowest point: the halves fill each other's tails); 1 = one stream; 2 = two
 * streams whenever the fused step runs.                                                                                */
int glh_set_track_streams(glh_ctx* ctx, int n);

/* glh_step implementation: 1 (default) = the fused per-point kernel (weights + resample +
 * re-evolving gather + moments in one launch, evolved state never round-trips through HBM)
 * whenever no active mask / debug capture is in force; 0 = always the staged kernels
 * (glh_evolve -> glh_update_weights -> glh_resample); 2 = fused, but with every search tile
 * forced into the HBM workspaces (test hook for the large-tile path).  All give the same
 * particles bit for bit.                                                                      */
int glh_set_fused(glh_ctx* ctx, int on);
/* GLH_MATH_EXACT (default) or GLH_MATH_FAST for every kernel of this context (the staged and the fused kernels use
 * the same arithmetic in either mode, so they stay bit-identical to each other).                                */
int glh_set_math(glh_ctx* ctx, int mode);
/* Window of the median high-pass filter of every tile (Tracker(highpass={"size": (size_y, size_x)}), tracker.py:59,
 * :530: scipy.ndimage.median_filter): odd sizes up to 7; 5 x 5 (the reference default) unless set.                                                                                           */
int glh_set_highpass(glh_ctx* ctx, int size_x, int size_y);
/* Boundary mode of that filter (Tracker(highpass={"size": ..., "mode": ...}): tracker.py:530 hands the dictionary to
 * scipy.ndimage.median_filter): 0 'reflect' (scipy's default, and the reference's), 1 'nearest', 2 'mirror', 3 'wrap'.
 * ('constant' needs a fill value in the matched tile's units: not served.)                                             */
int glh_set_highpass_mode(glh_ctx* ctx, int mode);
/* Orders of the spline that samples the SSD surface at the particles (Tracker(interpolation={"kx": .., "ky": ..}),
 * tracker.py:60, :585-590, :623: scipy RectBivariateSpline(kx, ky), s = 0): (3, 3), the reference default, or (1, 1)
 * -- bilinear; any other orders 1 .. 5 (kx: rows axis, ky: columns axis): the interpolating spline of those degrees with
 * FITPACK's knots, banded solves of bandwidth k, on the staged kernels.  The orders also set the least size of the
 * surface (the search box is widened to ky + 1 columns and kx + 1 rows, tracker.py:585-590).                        */
int glh_set_interpolation(glh_ctx* ctx, int kx, int ky);

/* Diagnostic: the numbers of the GLH_RNG_PHILOX streams of this context's points (global indices point_offset ..),
 * so that a parity test can hand the CPU oracle the draws a device-RNG run consumed -- the role np.random plays in
 * the reference (motion.py:156-162, :176; tracker.py:173).  kind 0: initialisation normals out [P][N][6] in the order
 * randn(n,2) | randn(n) | randn(n,3); kind 1: the evolve normals of frame `step`, out [P][N][3]; kind 2: the
 * systematic resampling offset of frame `step`, out [P].  `step` is the frame index passed to glh_step.          */
int glh_debug_draws(glh_ctx* ctx, int kind, uint64_t seed, uint64_t step, double* out);
/* Diagnostic: s_memtime stamps [P][24] at the phase boundaries of the fused kernel during the
 * last fused glh_step (the first call only arms them and returns zeros).                      */
int glh_debug_phase_stamps(glh_ctx* ctx, uint64_t* stamps);
/* Diagnostic: which instantiation of the fused kernel took the last fused glh_step / glh_track frame:
 * variant[0..3] = threads per workgroup, particles kept in registers per thread, observers, flags (bit 0: fast
 * arithmetic, bit 1: the general code (gridded surfaces, every motion model), bit 2: the compile-time contract of long
 * device-RNG runs; flags == 5 is the common instantiation bench.py times).  Zeros before any.                    */
int glh_debug_last_variant(glh_ctx* ctx, int32_t* variant);
/* Streams the last glh_track call ran on (1 or 2). */
int glh_debug_last_track_streams(glh_ctx* ctx, int* n);

/* ---- results --------------------------------------------------------------------------- */
/* means/sigmas for frames [frame0, frame0 + n_frames): out [n_frames][P][12].              */
int glh_get_moments(glh_ctx* ctx, int frame0, int n_frames, double* out);
/* The same history in the layout of the reference's Tracks (tracks.py:52-88): means [P][n_frames][6] and sigmas
 * [P][n_frames][6], rearranged on the device (no host-side transposes of tens of megabytes).                    */
int glh_get_tracks(glh_ctx* ctx, int frame0, int n_frames, double* means, double* sigmas);
/* Device pointer + byte size of the moments history [max_frames][P][12] (for an RCCL
 * gather issued by the caller; no copy).                                                    */
int glh_get_moments_device(glh_ctx* ctx, void** dev_ptr, uint64_t* bytes);
/* Template of (obs, point): box[4], duv[2], tile [th][tw], histogram (values, quantiles)
 * (track/tracker.py:552-561).  hist_n receives the number of CDF entries (<= th*tw).       */
int glh_get_template(glh_ctx* ctx, int obs, int point, int32_t* box, double* duv, double* tile,
                     double* hist_values, double* hist_quantiles, int32_t* hist_n);
/* Intermediates of the last glh_update_weights for (obs, point), for parity tests:
 * uv [N][2], box[4] (l,t,r,b), search tile float32 [Hs][Ws], sse float64 [Ho][Wo] (the
 * float32 SSE surface widened, before the spline fit).  Any pointer may be NULL.           */
int glh_get_likelihood_debug(glh_ctx* ctx, int obs, int point, double* uv, int32_t* box,
                             float* search, double* sse);
/* keep = 1: keep a copy of the SSE surface before the in-place spline fit, the per-observer log
 * likelihoods and the resample indices (costs extra passes and takes glh_step through the staged
 * kernels; off by default, on for parity tests).  keep = 2: the resample indices only (glh_step
 * stays on the fused kernel).                                                                  */
int glh_set_debug(glh_ctx* ctx, int keep);
/* Per-observer log likelihoods of the last glh_update_weights, i.e. the return value of
 * compute_observer_log_likelihoods (track/tracker.py:563-625): ll [P][N], NaN where the
 * reference returns None (needs glh_set_debug).                                              */
int glh_get_log_likelihoods(glh_ctx* ctx, int obs, double* ll);
/* np.searchsorted result of the last glh_resample (needs glh_set_debug): idx [P][N].        */
int glh_get_resample_indices(glh_ctx* ctx, int32_t* idx);
/* Per-stage device time (ms, HIP events on the context's stream) accumulated since the
 * last reset: names in glh_stage_name(i), i < glh_stage_count().                            */
int glh_profile_enable(glh_ctx* ctx, int on);
int glh_profile_reset(glh_ctx* ctx);
int glh_stage_count(void);
const char* glh_stage_name(int stage);
/* `launches` counts kernel launches: when glh_track runs a batch on two streams (glh_set_track_streams) a frame update
 * is TWO launches of the fused step, one per half of the points.                                                  */
int glh_profile_get(glh_ctx* ctx, double* ms /* [stages] */, int64_t* launches /* [stages] */);
/* Duration (ms) of every timed launch of `stage` since the last reset, in launch order: up to `cap` values
 * into ms, *n = how many there are (the first frames after the wide prior run longer than the steady state). */
/* GPU time (ms) a stage spans since the last reset: start of its first timed launch to the end of its last, on whichever
 * stream (the launches of glh_track's two streams overlap, so their durations do not add up to the time they take).  */
int glh_profile_get_span(glh_ctx* ctx, int stage, double* ms);
int glh_profile_get_launches(glh_ctx* ctx, int stage, double* ms, int cap, int* n);
/* Measured device-copy ceiling of this GPU: `iters` device-to-device copies of `bytes`
 * bytes on the context's stream between two HIP events; *gbps = bytes read + bytes written per second / 1e9. */
int glh_measure_copy_bandwidth(glh_ctx* ctx, uint64_t bytes, int iters, double* gbps);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ------------------------------------------
 * The reference's one parallel seam is a map over tracks (track/tracker.py:381-387,
 * `config.backend(np=parallel)`, helpers.py:2008-2017): tracks are independent, so here every process owns
 * one GPU, one context and a contiguous block of the tracked points (glh_set_point_offset), nothing is
 * exchanged while a sequence runs, and at its end the per-point posterior moments are collected on one rank.
 * librccl is loaded on the first of these calls (dlopen), never by a single-GPU user.                  */
#define GLH_COMM_ID_BYTES 128
/* One rank (the root) makes the communicator id (ncclGetUniqueId) and hands its 128 bytes to the others by
 * whatever the launcher offers (glimpse_amd/sharding.py: a file store under MASTER_PORT).               */
int glh_comm_unique_id(char* id /* [GLH_COMM_ID_BYTES] */);
/* Join the communicator as `rank` of `world` on the context's device and stream (ncclCommInitRank:
 * collective, every rank calls it).                                                                    */
int glh_comm_init(glh_ctx* ctx, const char* id, int rank, int world);
int glh_comm_destroy(glh_ctx* ctx);
/* Every rank's stream has reached this point (an all-reduce of one word, then a stream sync).            */
int glh_comm_barrier(glh_ctx* ctx);
/* *value = max over the ranks of *value (host double; wall-clock maxima of a timed region).              */
int glh_comm_max_f64(glh_ctx* ctx, double* value);
/* The one collective of a sequence: frames [frame0, frame0 + n_frames) of every rank's moments history
 * ([n_frames][P_rank][12], what `process` returns per track, tracker.py:370-373) and its per-point status
 * words to rank `root`, as ONE group of ncclSend / ncclRecv on the context's stream.  points_per_rank
 * [world]; on the root `out` receives | rank 0: [n_frames][P_0][12] | rank 1: ... | and `status` (or NULL)
 * | P_0 | P_1 | ... |; other ranks pass NULL.  Blocks until the exchange is over.                        */
int glh_gather_moments(glh_ctx* ctx, int root, int frame0, int n_frames, const int32_t* points_per_rank,
                       double* out, uint32_t* status);
/* On the root, `out` may be NULL: the blocks then stay in the root's device memory (the exchange is all the call does)
 * and glh_get_gathered copies them to the host later, in the layout above.                                        */
int glh_get_gathered(glh_ctx* ctx, double* out, uint32_t* status);

/* ---- stage-level test hooks (stateless; each runs one kernel on explicit inputs) -------- */
/* Camera.xyz_to_uv (camera.py:591-628): xyz [n][3] -> uv [n][2].                            */
int glh_stage_project(int device_id, const double* cam, const double* xyz, int n, double* uv);
/* Same with xyz read as ray directions relative to the camera (directions=True, camera.py:1448).  */
int glh_stage_project_directions(int device_id, const double* cam, const double* xyz, int n, double* uv);
/* Camera.xyz_to_uv(return_depth=True) (camera.py:591-628, :1468-1469): uv [n][2] and the distance of every
 * point along the optical axis, depth [n] (also for points behind the camera, whose uv are NaN).   */
int glh_stage_project_depth(int device_id, const double* cam, const double* xyz, int n, int directions,
                            double* uv, double* depth);
/* Camera.uv_to_xyz (camera.py:630-663): uv [n][2] -> xyz [n][3]; depth NULL (= 1), [1] or [n];
 * undistortion by the closed form for k1 alone, else 20 Oulu iterations (camera.py:1198-1337).   */
int glh_stage_unproject(int device_id, const double* cam, const double* uv, int n, const double* depth,
                        int n_depth, int directions, double* xyz);
/* Tracker.extract_tile(return_histogram=True) (tracker.py:494-534) on a uint8 frame crop
 * `box` (l,t,r,b): tile float64 [h][w], CDF values/quantiles, n entries.                    */
int glh_stage_template(int device_id, const uint8_t* frame, int width, int height, int channels,
                       const int32_t* box, double* tile, double* hist_values,
                       double* hist_quantiles, int32_t* hist_n);
/* Tracker.extract_tile(histogram=...) (tracker.py:494-534): search tile float32 [h][w].     */
int glh_stage_search_tile(int device_id, const uint8_t* frame, int width, int height,
                          int channels, const int32_t* box, const double* hist_values,
                          const double* hist_quantiles, int hist_n, float* tile);
/* The two tile hooks with another high-pass window (odd sizes up to 7) and boundary mode (glh_set_highpass_mode).   */
int glh_stage_template_highpass(int device_id, const uint8_t* frame, int width, int height, int channels,
                                const int32_t* box, int size_x, int size_y, int mode, double* tile,
                                double* hist_values, double* hist_quantiles, int32_t* hist_n);
int glh_stage_search_tile_highpass(int device_id, const uint8_t* frame, int width, int height, int channels,
                                   const int32_t* box, const double* hist_values, const double* hist_quantiles,
                                   int hist_n, int size_x, int size_y, int mode, float* tile);
/* cv2.matchTemplate(TM_SQDIFF) * 1/(tw*th) (tracker.py:609-614): float32 in, float32 out.   */
int glh_stage_ssd(int device_id, const float* search, int hs, int ws, const float* templ, int th,
                  int tw, float* sse);
/* Observer.sample_tile (observer.py:178-214): sse float32 [ho][wo], box (4 doubles),
 * uv [n][2] -> values [n]; outside [n] flags points outside the box.                         */
int glh_stage_sample(int device_id, const float* sse, int ho, int wo, const double* box,
                     const double* uv, int n, double* values, uint8_t* outside);
/* The same for any orders of RectBivariateSpline (kx: rows axis, ky: columns axis, each 1 .. 5; Tracker(interpolation=
 * {"kx": ..., "ky": ...}), tracker.py:60, :623): ho >= kx + 1, wo >= ky + 1.                                          */
int glh_stage_sample_orders(int dev, const float* sse, int ho, int wo, int kx, int ky, const double* box,
                            const double* uv, int n, double* values, uint8_t* outside);
/* Raster.sample(xy, order) (raster.py:913-1027) at n points: values [n], oob [n] = 1 where the
 * reference would raise (outside the outer limits).                                            */
int glh_stage_raster_sample(int device_id, const double* z, int nx, int ny, const double* gx,
                            const double* gy, int sx, int sy, double xmin, double xmax, double ymin,
                            double ymax, const double* xy, int n, int order, double* values,
                            uint8_t* oob);
/* Tracker.resample_particles("systematic") on one population: idx int64 [n].                 */
int glh_stage_resample(int device_id, const double* weights, int n, double u, int64_t* idx);

#ifdef __cplusplus
}
#endif
#endif /* GLIMPSE_HIP_H */
