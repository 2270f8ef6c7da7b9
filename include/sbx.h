/* include/sbx.h — C ABI of libsbx: the MI355X drop-in for shaderbox's mainImage() hot path.
 *
 * What this boundary replaces.  In the reference a *host* evaluates
 *     void mainImage(out vec4 fragColor, in vec2 fragCoord)        (src/main.h:6-53)
 * once per pixel for the app chosen by one global APP_* define (README.md:11-22,
 * src/Makefile:9): the external VML/SDL harness on the C++ path (src/Makefile:21), or the
 * D3D11 host util/hlsltoy (its per-frame Draw, util/hlsltoy/src/hlsltoy.cpp:494-495, with
 * the uniforms of src/uniform_buffer.h uploaded as cbuffers b0/b1, hlsltoy.cpp:402-426,
 * 502-516).  A per-pixel FFI into a GPU is meaningless, so the boundary is frame-granular:
 * one call renders rows of one frame of one app into a caller-owned RGBA32F framebuffer in
 * device memory (sbx_render_rows); sbx_main_image keeps the per-pixel signature on top of it
 * for hosts that want their loop unchanged.  Everything a host needs is plain C: pointers,
 * sizes, POD structs.
 *
 * Conventions (all from the reference):
 *   - framebuffer: row-major float4 RGBA, 16 B/pixel, row 0 = BOTTOM row (main.h:40-43: the
 *     y flip is HLSL-only), alpha = 1 (main.h:52);
 *   - fragCoord of pixel (x, y) is (x + .5, y + .5);
 *   - `_mutable` globals have GLSL per-invocation meaning (def.h:18): every pixel starts from
 *     the initialisers;
 *   - errors: the reference has none (void everywhere, NaN flows to the framebuffer).  Here every
 *     entry point returns 0 on success or a negative sbx_status; NaNs stay data; nothing aborts.
 *   - threading: one caller per sbx_ctx at a time; calls are asynchronous on the given HIP
 *     stream (hipStream_t passed as void*; NULL = the default stream).
 */
#ifndef SBX_H
#define SBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point, enum value or struct layout of this header changes (2 = round 5: the store exchange
 * sbx_shared_*, sbx_stats, sbx_abi_version itself; the test hooks moved to sbx_test.h).  A host checks
 * sbx_abi_version() == SBX_ABI_VERSION after loading the library: include/sbx_mainimage.hpp and shaderbox_amd.load_library do. */
#define SBX_ABI_VERSION 2
int sbx_abi_version(void);

/* App selector = the reference's APP_* project defines, in README.md:15-22 order, plus
 * APP_SDF_AO (src/uniform_buffer.h:56, util/hlsltoy/src/hlsltoy.cpp:488). */
typedef enum sbx_app {
    SBX_APP_PLANET = 0,
    SBX_APP_CLOUDS = 1,
    SBX_APP_VINYL = 2,      /* C++-build semantics: 60 march steps (src/app_vinyl.h:411-416) */
    SBX_APP_EGG = 3,
    SBX_APP_RAYTRACER = 4,
    SBX_APP_ATMOSPHERE = 5,
    SBX_APP_SDF_AO = 6,
    /* not an APP_* define of the reference: the stand-alone shader src/app_clouds_best.h (own mainImage :669-696),
       numbered after the reference's apps */
    SBX_APP_CLOUDS_BEST = 7,
    /* APP_CLOUDS compiled with USE_NOISE_TEX (src/app_clouds.h:9,51-56,69-81): density from two 3-D noise textures
       (sbx_set_noise_volumes) instead of the procedural fBm; same aux block as APP_CLOUDS */
    SBX_APP_CLOUDS_TEX = 8,
    /* the UE4 cloud variant, ue4/volumetric_clouds/Shaders/app_clouds.usf (ue4_render_clouds :234-265): a library for an
       Unreal material, not a mainImage shader.  Host mapping of this build: cam_dir = the primary-ray direction of
       APP_CLOUDS' camera, time = u_time, parameters = the TWEAK defaults (:4-17) or an sbx_aux_clouds_ue4 block, result
       through main.h's sRGB epilogue.  No reference-held answers: parity unpinned. */
    SBX_APP_CLOUDS_UE4 = 9,
    /* APP_CLOUDS compiled with SKY_SPHERE (src/app_clouds.h:8,14-19,154-162; intersect_sphere_from_inside src/intersect.h:35-53):
       the march starts where the view ray meets the sphere ((0, atm_ground_y, 0), atm_radius) — which makes those two fields
       of the aux block live — runs along the view ray itself, the layer turns with rotate_around_x(u_time) and
       cld_noise_factor is 10 / atm_radius.  Same aux block as APP_CLOUDS (wind_dir is not read).  Parity unpinned (no
       reference-held answers; bit-identical to the oracle's restatement). */
    SBX_APP_CLOUDS_SKY = 10,
    /* APP_VINYL with the march length of its GLSL / HLSL builds: 180 steps instead of the C++ build's 60
       (src/app_vinyl.h:411-416).  Everything else as SBX_APP_VINYL. */
    SBX_APP_VINYL_GPU = 11,
    /* BASELINE config 5 read literally — "APP_ATMOSPHERE Rayleigh/Mie over APP_PLANET terrain".  The reference has no shader that
       composites the two (SURVEY.md 8a note); this labelled extension is APP_PLANET (src/app_planet.h:303-367) with its
       background() (:23-41, shown where the view ray misses the atmosphere shell :316-318 and behind the clouds :364-366)
       replaced by APP_ATMOSPHERE's get_incident_light (src/app_atmosphere.h:78-160) for the ray from (0, earth_radius + 1, 0)
       (:204-207) along the VIEW direction, lit by APP_ATMOSPHERE's sun (setup_scene :177-181, a function of u_time).
       No reference-held answers: PARITY UNPINNED (bit-identical to the oracle's restatement of the same definition). */
    SBX_APP_PLANET_ATMOSPHERE = 12
} sbx_app;

typedef enum sbx_status {
    SBX_OK = 0,
    SBX_ERR_ARG = -1,          /* NULL pointer, bad row range, bad resolution ... */
    SBX_ERR_UNSUPPORTED = -2,  /* app not on the accelerated path */
    SBX_ERR_HIP = -3,          /* a HIP runtime call failed; see sbx_last_error() */
    SBX_ERR_NO_DEVICE = -4,    /* no gfx950 device visible: the library never falls back to a CPU */
    SBX_ERR_FAULT = -5         /* a kernel reported an internal invariant violation on this device (sticky; sbx_fault_status) */
} sbx_status;

/* cbuffer b0 (src/uniform_buffer.h:25-30): u_res@c0.xy, u_mouse@c0.zw, u_time@c1.x.
 * On the C++/Shadertoy path these are iResolution / iMouse / iGlobalTime (:32-36). */
typedef struct sbx_uniforms {
    float u_res[2];
    float u_mouse[2];
    float u_time;
    float _pad[3];
} sbx_uniforms;

/* cbuffer b1 for APP_CLOUDS (src/uniform_buffer.h:39-55), same packoffsets and defaults.
 * Pass NULL as `aux` to get the defaults (what the C++/GLSL builds compile in, :13). */
typedef struct sbx_aux_clouds {
    float wind_dir[3];   float _pad0;   /* c0   default (0, 0, .2)     */
    float sun_dir[3];    float _pad1;   /* c1   default (0, 0, -1)     */
    float sun_color[3];  float _pad2;   /* c2   default (1, .7, .55)   */
    float sun_power;                    /* c3.x default 8              */
    int32_t cld_march_steps;            /* c3.y default 100            */
    int32_t illum_march_steps;          /* c3.z default 6              */
    float sigma_scattering;             /* c3.w default .15            */
    float cld_coverage;                 /* c4.x default .535           */
    float cld_thick;                    /* c4.y default 125            */
    float atm_radius;                   /* c4.z default 5000 (SKY_SPHERE only, unused) */
    float atm_ground_y;                 /* c4.w default 4750 (SKY_SPHERE only, unused) */
} sbx_aux_clouds;

/* cbuffer b1 for APP_SDF_AO (src/uniform_buffer.h:56-60) */
typedef struct sbx_aux_sdf_ao {
    float fog_density;   /* c0.x default .1 */
    float fog_falloff;   /* c0.y default .5 */
    float _pad[2];
} sbx_aux_sdf_ao;

/* the material parameters of ue4_render_clouds (app_clouds.usf:234-243) for SBX_APP_CLOUDS_UE4; NULL = the TWEAK block's
 * defaults (:4-7) with the SUN_DIR / WIND_DIR macros (:13-14, functions of u_time) */
typedef struct sbx_aux_clouds_ue4 {
    float coverage, thickness, absorbtion, fuzziness;   /* c0   defaults .50, 15, 1.030725, .035 */
    float sun_dir[3];    float _pad1;                   /* c1   used when use_dirs != 0 */
    float wind_dir[3];   int32_t use_dirs;              /* c2   use_dirs = 0: SUN_DIR / WIND_DIR of the shader */
} sbx_aux_clouds_ue4;
void sbx_aux_clouds_ue4_defaults(sbx_aux_clouds_ue4* aux);

typedef struct sbx_ctx sbx_ctx;

/* Fill the aux blocks with the reference's defaults. */
void sbx_aux_clouds_defaults(sbx_aux_clouds* aux);
void sbx_aux_sdf_ao_defaults(sbx_aux_sdf_ao* aux);

/* Create / destroy a context bound to HIP device `device`.  Owns only small device scratch
 * (APP_CLOUDS' per-frame y tables, timing events, the cached frame of sbx_main_image); the
 * framebuffer of the render calls always belongs to the caller. */
int sbx_create(int device, sbx_ctx** out);
void sbx_destroy(sbx_ctx* ctx);

/* Render rows [y0, y1) of the frame described by `uni` (u_res = full frame size) for `app`.
 * `rgba` points at device memory for row y0: pixel (x, y) lands at rgba[((y - y0) * W + x) * 4].
 * This is the replacement of "for every pixel: mainImage(fragColor, fragCoord)". */
int sbx_render_rows(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux,
                    int y0, int y1, float* rgba, void* stream);
/* The same rows for a host that owns a HOST framebuffer (SURVEY.md 8b "Entry signature": rgba_device_or_host; the reference's
 * harness loops mainImage over a host surface, src/main.h:6-53): `rgba_host` is host memory for row y0, 4-byte aligned, in the
 * context's output format (16 or 4 bytes per pixel).  The rows are rendered into a staging buffer of the context, behind whatever
 * `stream` holds, and copied out; returns when all pixels are in `rgba_host`.  Pinned memory (hipHostMalloc / hipHostRegister): up to
 * sixteen strips, each copied at the link's rate while the next ones render (CLOUDS 4K 3.55 ms for a 2.7 ms kernel at these clocks
 * + 2.3 ms of copy; an 8K frame 10 ms for 9.3 ms of copy).
 * Pageable memory: one strip, one synchronous copy.  Not capturable; one caller per context.
 * (sbx_render_rows itself accepts any DEVICE-ACCESSIBLE pointer: handed pinned host memory, the kernel's own stores cross PCIe —
 * no staging, one launch; profiles/r05_host_boundary.txt has both.) */
int sbx_render_rows_host(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux,
                         int y0, int y1, void* rgba_host, void* stream);

/* The reference's own per-pixel entry, for hosts that keep their pixel loop:
 *     void mainImage(out vec4 fragColor, in vec2 fragCoord)            src/main.h:6-9
 * (what the absent VML/SDL harness calls per pixel, src/Makefile:21).  mainImage is a function of fragCoord — it divides it
 * by u_res (main.h:40) and never snaps or clamps it — and so is this call:
 *   - fragCoord = the centre (x + .5, y + .5) of a pixel of the frame (y = 0 at the bottom row) and u_res whole numbers: the
 *     first call for a given (app, uniforms, aux) renders the whole frame on the GPU with one launch and brings it to pinned
 *     host memory — the kernel's own stores cross PCIe, no copy follows it; later calls read their pixel from it;
 *   - ANY other fragCoord (off-centre: a supersampling host; outside the frame; a fractional u_res): evaluated exactly at that
 *     coordinate by a one-point launch (sbx_main_image_batch with n = 1) — never another pixel's sample.
 * Safe to call from several host threads on one context (serialised inside); hosts with many samples per frame should use
 * sbx_main_image_batch or sbx_render_points. */
int sbx_main_image(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux,
                   const float fragCoord[2], float fragColor[4]);
/* mainImage at n arbitrary fragCoords in ONE launch: the per-pixel entry in the shape a GPU can serve (n samples of a
 * supersampling / jittering / foveated host).  Nothing is cached and nothing is assumed about the coordinates; u_res may be any
 * positive finite numbers.  sbx_main_image_batch: host arrays (fragCoords n x 2 floats in, fragColors n x 4 floats out),
 * staged through pinned memory, returns when the colours are there.  sbx_render_points: device arrays (frag n x 2, rgba n x 4,
 * 16-byte aligned), asynchronous on `stream`. */
int sbx_main_image_batch(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, size_t n,
                         const float* fragCoords, float* fragColors);
int sbx_render_points(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, size_t n,
                      const float* frag, float* rgba, void* stream);

/* Render the cyclic row-blocks owned by one rank of an N-way split: blocks of
 * `block_rows` rows, rank r owns blocks r, r+N, r+2N, ...  The rank's rows are written densely,
 * in increasing y, into `rgba` (capacity sbx_rank_rows() rows).  Each pixel is computed from its
 * GLOBAL (x, y), so the assembled frame is bit-identical to a 1-GPU render. */
int sbx_render_rank(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux,
                    int block_rows, int rank, int nranks, float* rgba, void* stream);
/* Same, restricted to slab rows [r0, r1) of the rank (r1 is clipped to sbx_rank_rows()); `rgba` points at
 * slab row r0.  Lets a host pipeline the slab in groups: render group g, start its transfer, render g+1. */
int sbx_render_rank_rows(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux,
                         int block_rows, int rank, int nranks, int r0, int r1, float* rgba, void* stream);
/* Number of rows rank `rank` owns (<= sbx_rank_rows_max). */
int sbx_rank_rows(int height, int block_rows, int rank, int nranks);
/* Rows every rank's buffer must hold so that an equal-count gather works: max over ranks. */
int sbx_rank_rows_max(int height, int block_rows, int nranks);
/* Root-side frame assembly after the gather: `gathered` holds nranks slabs of
 * sbx_rank_rows_max() rows each (rank-major); scatter them to their global rows of `frame`. */
int sbx_assemble(sbx_ctx* ctx, int width, int height, int block_rows, int nranks,
                 const float* gathered, float* frame, void* stream);

/* The same split with ROOT RELIEF.  Rank 0, the gather's root, also receives nranks - 1 slabs and assembles the frame;
 * so that it does not become the slowest rank it can be dealt fewer row-blocks: blocks go out in cycles of `rounds`
 * rounds, a round gives one block to every rank, and rank 0 is left out of the rounds >= root_rounds
 * (0 <= root_rounds <= rounds; root_rounds = rounds = 1 is the plain cyclic split of the calls above).
 * sbx_render_split renders slab rows [r0, r1) of `rank` (r1 is clipped to the rank's row count), sbx_split_rows_max
 * is the slab height every rank allocates, sbx_assemble_split the root-side scatter. */
int sbx_split_rank_rows(int height, int block_rows, int rank, int nranks, int root_rounds, int rounds);
int sbx_split_rows_max(int height, int block_rows, int nranks, int root_rounds, int rounds);
int sbx_render_split(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                     int nranks, int root_rounds, int rounds, int r0, int r1, float* rgba, void* stream);
int sbx_assemble_split(sbx_ctx* ctx, int width, int height, int block_rows, int nranks, int root_rounds, int rounds,
                       const float* gathered, float* frame, void* stream);
/* The rows of `rank` written IN PLACE: `frame` is a full-size frame (height * width pixels) and every row of the rank
 * lands at its global position; the other ranks' rows are not touched.  What the owner of the frame uses for its own share
 * (no slab, no assembly pass). */
int sbx_render_split_in_place(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                              int nranks, int root_rounds, int rounds, float* frame, void* stream);
/* The direct exchange of a one-process-per-GPU host (the reference has no multi-device path; SURVEY.md 8e "Collective": the
 * peers' row-blocks go to the root by point-to-point sends, the root's own rows never move).  The alpha of every pixel is the
 * constant 1 that mainImage's caller writes (src/main.h:52), so a peer's slab crosses xGMI as 3 floats per pixel:
 * sbx_render_split_rgb = sbx_render_split with a slab of rows x width x 3 floats (12 bytes per pixel, 25 % fewer bytes on
 * the link and in the root's HBM).  sbx_assemble_peers scatters the slabs of ranks 1 .. nranks-1 (`peers`: rank-major,
 * sbx_split_rows_max() rows each, `channels` = 3 or 4 floats per pixel) to their global rows of the RGBA `frame`, writing
 * alpha = 1 for 3-channel slabs, and leaves the rows of rank 0 — rendered with sbx_render_split_in_place — untouched. */
int sbx_render_split_rgb(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                         int nranks, int root_rounds, int rounds, int r0, int r1, float* rgb, void* stream);
int sbx_assemble_peers(sbx_ctx* ctx, int width, int height, int block_rows, int nranks, int root_rounds, int rounds,
                       int channels, const float* peers, float* frame, void* stream);

/* ---- The span exchange: only the EXPENSIVE part of a row-block is sharded -----------------------------------------------
 * (The reference has no multi-device path; this is the build's answer to SURVEY.md 8e at the sizes where the one exchange step
 * is link-bound: at 7680x4320 a peer's 3-channel slab is 49.8 MB, 0.65 ms on one xGMI link at its 76.8 GB/s peak, against 0.47 ms
 * of APP_ATMOSPHERE rendering per rank.)  Several apps leave mainImage through an early exit over a large part of the frame:
 * APP_CLOUDS below the horizon (src/app_clouds.h:212), APP_ATMOSPHERE outside the dome and where the view ray dives under the
 * ground (src/app_atmosphere.h:196-207, 65-67), APP_PLANET where the ray misses the atmosphere shell (src/app_planet.h:315-321).
 * sbx_span_table gives, for every row-block g of the split, the interval [x0, x1) of columns (multiples of 64) OUTSIDE of which
 * the host expects only such cheap pixels — a hint computed with the kernels' own tests at tile corners, widened by a tile:
 *   - a peer renders and sends only the spans of its blocks, packed block after block, 3 floats per pixel (sbx_render_span_peer);
 *   - the frame's owner renders its own blocks AND everything outside the other blocks' spans, in place, with ONE launch over the
 *     frame (sbx_render_span_root), receives the packed slabs (still exactly one exchange step) and scatters them, writing alpha
 *     (sbx_assemble_spans).
 * Whoever renders a pixel runs the app's full kernel on its global coordinates: the table moves work and bytes, never a bit of
 * the image.  Apps without a model get whole rows (the exchange then equals the direct one).
 *
 * sbx_span_table (host only; no context, no GPU): table = 4 int32 per row-block {x0, x1, offset of the block's span in its
 * owner's packed slab (pixels), owner rank} or NULL; rank_pixels = nranks int64 (pixels of every rank's packed slab; rank 0's
 * counts its own spans, which never travel) or NULL; max_width = the widest span among the peers' blocks or NULL.  Returns the
 * number of row-blocks or a negative sbx_status.  The table depends on (app, u_res, u_mouse, split) only.
 * sbx_render_span_peer: slab rows [r0, r1) (whole blocks; r1 clipped) of peer `rank` (1 .. nranks-1); `rgb` is the START of the
 * rank's packed slab (3 * rank_pixels[rank] floats): the table's offsets are absolute in it, so a host can render and send the
 * slab in pieces.  sbx_assemble_spans: `peers` holds the slab of rank r at (r - 1) * stride_pixels * 3 floats.
 * The context keeps the device copies of the last few tables; a frame whose table is not on the device yet cannot be recorded
 * into a stream capture (render it once before). */
int sbx_span_table(int app, const sbx_uniforms* uni, const void* aux, int block_rows, int nranks, int root_rounds, int rounds,
                   int32_t* table, int64_t* rank_pixels, int32_t* max_width);
int sbx_render_span_peer(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank, int nranks,
                         int root_rounds, int rounds, int r0, int r1, float* rgb, void* stream);
int sbx_render_span_root(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int nranks,
                         int root_rounds, int rounds, float* frame, void* stream);
int sbx_assemble_spans(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int nranks,
                       int root_rounds, int rounds, const float* peers, int64_t stride_pixels, float* frame, void* stream);

/* ---- The store exchange: the peers render IN PLACE into the owner's frame ------------------------------------------------
 * (The reference has no multi-device path; this is the form of the ONE exchange step in which the frame's owner
 * does nothing for the others: no landing area, no receive kernels, no scatter pass.)  The owner allocates the frame through the
 * library and exports it; every peer — another PROCESS on another (or the same) GPU: hipIpcGetMemHandle / hipIpcOpenMemHandle; or
 * another rank of the same process: the pointer itself — maps it and renders its row-blocks of the split straight into it with
 * sbx_render_split_in_place[_rgb]: the exchange is the render kernels' own pixel stores over xGMI, 16 bytes per pixel (float4),
 * 12 (sbx_render_split_in_place_rgb: three dwords; the alpha of every pixel is the constant 1 of src/main.h:52, written once when the
 * frame is created) or 4 (SBX_FORMAT_RGBA8).  Two small flag kernels per frame and rank order it, through a page of fine-grained
 * memory next to the frame:
 *     owner:  sbx_shared_frame_begin (stream-ordered behind whatever read the previous frame: tells the peers the frame may be
 *             overwritten) .. its own sbx_render_split_in_place .. sbx_shared_frame_end (the stream continues when every peer's
 *             rows of THIS frame are in place);
 *     peer r: sbx_shared_frame_begin (waits for the owner's go) .. sbx_render_split_in_place .. sbx_shared_frame_end (signals).
 * Every rank calls begin / end once per frame, in order; frames of one sbx_shared are sequential, a host keeps several frames in
 * flight with several sbx_shared (one per stream).  A wait that sees no signal for ~10 s raises the device's fault word
 * (sbx_fault_status) instead of hanging.  Pixels are written by the app's full kernel from their global coordinates, so the frame
 * is bit-identical to a one-GPU render.
 * sbx_shared_create: `frame_bytes` = height * width * (16 or 4); nranks >= 1.  sbx_shared_export fills an opaque handle that may
 * be sent to the other processes by any means (it holds no pointers valid elsewhere); sbx_shared_open maps it on `ctx`'s device.
 * sbx_shared_frame = the frame's address in THIS process.  Closing the owner's object frees the frame: close the peers' first. */
typedef struct sbx_shared sbx_shared;
typedef struct sbx_shared_handle { unsigned char opaque[192]; } sbx_shared_handle;
int sbx_shared_create(sbx_ctx* ctx, size_t frame_bytes, int nranks, sbx_shared** out);
int sbx_shared_export(sbx_shared* s, sbx_shared_handle* handle);
int sbx_shared_open(sbx_ctx* ctx, const sbx_shared_handle* handle, sbx_shared** out);
void sbx_shared_close(sbx_shared* s);
float* sbx_shared_frame(sbx_shared* s);
size_t sbx_shared_bytes(sbx_shared* s);            /* bytes of the frame (as given to sbx_shared_create), on either side */
int sbx_shared_frame_begin(sbx_shared* s, int rank, void* stream);
int sbx_shared_frame_end(sbx_shared* s, int rank, void* stream);
/* The store exchange with SPANS (the two ideas together, for frames whose pixel stores would bind a link: 7680x4320 in float
 * pixels): a peer renders only the spans of its row-blocks (sbx_span_table) — in place, into the owner's mapped frame, `channels` = 3
 * or 4 dwords per float pixel — and the owner renders its own blocks and everything outside the spans with sbx_render_span_root.
 * No slab, no landing area, no scatter; the links carry the expensive pixels only (29.7 instead of 49.8 MB per peer at 7680x4320). */
int sbx_render_span_peer_in_place(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                                  int nranks, int root_rounds, int rounds, int channels, float* frame, void* stream);
/* sbx_render_split_in_place writing only R, G, B of every float4 pixel (three dwords at a 16-byte stride); under
 * SBX_FORMAT_RGBA8 it is sbx_render_split_in_place. */
int sbx_render_split_in_place_rgb(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                                  int nranks, int root_rounds, int rounds, float* frame, void* stream);

/* The write into hlsltoy's DXGI_FORMAT_R8G8B8A8_UNORM back buffer (util/hlsltoy/src/hlsltoy.cpp:79,192): float RGBA
 * rows -> 8-bit RGBA by the Direct3D float -> UNORM rule (NaN -> 0, clamp to [0, 1], * 255 + .5, truncate).
 * `rgba` and `out` are device pointers (width * rows pixels each); flip_y != 0 writes the top row first
 * (D3D / image-file order; the float frame has row 0 at the bottom). */
int sbx_pack_unorm8(sbx_ctx* ctx, int width, int rows, const float* rgba, unsigned char* out, int flip_y,
                    void* stream);

/* OUTPUT FORMAT of the frame-granular entry points (default SBX_FORMAT_RGBA32F).  With SBX_FORMAT_RGBA8 every pixel a render
 * call writes — whole frames, strips, a rank's slabs (4- or "3-channel"), span slabs, in-place renders — is ONE 32-bit word
 * R | G << 8 | B << 16 | 255 << 24 by the Direct3D float -> UNORM rule of sbx_pack_unorm8 (the display format of the
 * reference's hosts: hlsltoy's R8G8B8A8_UNORM back buffer), written by the render kernel itself instead of its float pixel: the
 * `float*` buffers of those calls then hold width * rows words (4-byte aligned), and the assembly calls (sbx_assemble*,
 * sbx_assemble_peers with any `channels`, sbx_assemble_spans) move such words.  A frame rendered this way equals
 * sbx_pack_unorm8(flip_y = 0) of the float frame, bit for bit.  What it is for: a multi-GPU exchange that carries 4 instead of
 * 12 bytes per pixel (a 7680x4320 frame at 8 GPUs is bound by the 7 xGMI links into the root in float pixels), and a frame
 * that is written once in the format it is shown in.  sbx_render_points, sbx_main_image and sbx_main_image_batch always
 * return float colours. */
enum { SBX_FORMAT_RGBA32F = 0, SBX_FORMAT_RGBA8 = 1 };
int sbx_set_output_format(sbx_ctx* ctx, int format);

/* PRECISION TIER of the frame-granular and per-pixel entry points (default SBX_PRECISION_EXACT: every pixel bit-identical to the
 * CPU oracle of this repository).  SBX_PRECISION_1E4 is a labelled, opt-in tolerance tier for SBX_APP_ATMOSPHERE only (every other
 * app ignores it and stays exact): BASELINE.json's bar is 1e-4 per float channel against the C++ reference, not bit-equality, and
 * APP_ATMOSPHERE (src/app_atmosphere.h:50-160: 336 exp per in-dome pixel, no threshold that amplifies a rounding difference) meets
 * it with the hardware's binary32 exp2 in place of the math spec's binary64 table form — about half the frame time.  Max |diff|
 * against the oracle over every pixel of the 7680x4320 frame and over a sweep of sun positions is asserted in the tests.
 * Not offered for APP_CLOUDS / APP_PLANET: their coverage and density edges turn a 1-ulp difference into 1e-3 pixels. */
enum { SBX_PRECISION_EXACT = 0, SBX_PRECISION_1E4 = 1 };
int sbx_set_precision(sbx_ctx* ctx, int precision);

/* Counters of a context since its creation (or the last sbx_reset_stats): what a host or a test reads to see what its calls
 * turned into.  render_launches = render-kernel launches enqueued by any entry point; main_image_hits = sbx_main_image calls
 * served from the cached frame; main_image_frames = whole frames sbx_main_image rendered; main_image_points = one-point launches. */
typedef struct sbx_stats {
    uint64_t render_launches;
    uint64_t main_image_hits;
    uint64_t main_image_frames;
    uint64_t main_image_points;
    uint64_t _reserved[4];
} sbx_stats;
int sbx_get_stats(sbx_ctx* ctx, sbx_stats* out);
int sbx_reset_stats(sbx_ctx* ctx);

/* Per-launch timing: when enabled, every render call brackets its kernel with HIP events on the
 * launch stream; sbx_last_kernel_ms() synchronises on the last pair and returns the duration. */
int sbx_set_timing(sbx_ctx* ctx, int enabled);
int sbx_last_kernel_ms(sbx_ctx* ctx, float* ms);

/* The noise library as standalone functions over n points (xyz interleaved, device arrays), 3 floats out
 * per point: "noise_iq" (src/noise_iq.h:11-29, out[0]); "hash_w" (src/noise_worley.h:5-17);
 * "noise_w" (:20-51; params[0] = domain_repeat; out = sqrt F1, sqrt F2, |cell id|);
 * "fbm_worley_tile" (src/fbm.h:8 as instantiated at util/ddsvolgen/src/ddsvolgen.cpp:52;
 * params = lacunarity, init_gain, gain; out[0]).
 * (include/sbx_test.h documents three more names that exist for the parity tests of the recorded-domain forms.) */
int sbx_noise_eval(sbx_ctx* ctx, const char* fn, const float* xyz, const float* params, float* out,
                   size_t n, void* stream);
/* The size^3 RGBA32F noise volume util/ddsvolgen bakes (ddsvolgen.cpp:101-117): R =
 * fbm_worley_tile((xyz + .5)/size, 2, 1, .5), G = B = A = 0, x fastest.  `rgba`: size^3 * 4 floats. */
int sbx_worley_volume(sbx_ctx* ctx, int size, float* rgba, void* stream);

/* Bind the two 3-D noise textures of APP_CLOUDS' USE_NOISE_TEX build: u_tex_noise (t1, the cloud shape) and
 * u_tex_noise_2 (t2, the Worley detail), src/app_clouds.h:52-55 — what hlsltoy loads from the .dds files named on its
 * command line and binds with a MIN_MAG_MIP_LINEAR / WRAP sampler (util/hlsltoy/src/hlsltoy.cpp:227-249, 437).
 * Each is a size^3 RGBA32F volume in device memory, x fastest (the layout util/ddsvolgen writes, ddsvolgen.cpp:101-117;
 * sbx_worley_volume produces one).  The shader reads only .r: the call copies that channel into the context (R32F,
 * a quarter of the footprint) on `stream` and scans the copies for the range of the texel values (k_clouds_tex derives a bound on
 * the density from it and uses a cheaper, equal form of exp inside that bound); the call returns when both have finished, and the
 * caller's buffers are not referenced afterwards.  Inside a stream capture there is no scan and no wait.
 * SampleLevel is evaluated by the sbx texture-filter spec: texel centres at (i + .5) / size, WRAP,
 * trilinear blend with binary32 weights in x, y, z order.  Renders of SBX_APP_CLOUDS_TEX must be ordered after this
 * call (same stream, or an event). */
int sbx_set_noise_volumes(sbx_ctx* ctx, int shape_size, const float* shape_rgba, int detail_size,
                          const float* detail_rgba, void* stream);
/* ---- Multi-GPU frames inside the library (SURVEY.md §8b "Ownership", §8e "Collective") -----------------------------
 * One process drives `nranks` ranks; devices[i] is the HIP device of rank i, rank 0 owns the frame.  With all devices
 * distinct the library creates its communicator with ncclCommInitAll (rccl.h:236; librccl is dlopen'ed here, not linked)
 * and every peer's slab travels by ONE ncclSend / ncclRecv pair (rccl.h:700,722; all pairs of a frame in one group, N-1 distinct
 * xGMI links) into a landing area on rank 0, from where one small kernel scatters the rows into the frame; rank 0 renders its
 * own blocks in place.  Devices may repeat (several ranks on one GPU — how the N-rank schedule runs on fewer GPUs than
 * ranks): those transfers are device copies.  The split is the cyclic row-block split of sbx_render_split (default 8-row
 * blocks, no root relief).  Every sbx_multi_* call leaves rank 0's device current (hipSetDevice). */
typedef struct sbx_multi sbx_multi;
int sbx_multi_create(int nranks, const int* devices, sbx_multi** out);
/* Why the last sbx_multi_create of this process failed (e.g. the dlopen / dlsym text for librccl); "" after a success. */
const char* sbx_multi_create_error(void);
/* How the peers' rows reach rank 0: SLABS (default) = one send / receive per peer of its whole 3-channel slab + one scatter
 * kernel; BLOCKS = one send / receive pair per row-block straight into the final rows (no landing area, no scatter kernel,
 * but (N-1) x blocks point-to-point operations per frame in one group). */
enum { SBX_MULTI_EXCHANGE_SLABS = 0, SBX_MULTI_EXCHANGE_BLOCKS = 1,
       SBX_MULTI_EXCHANGE_SPANS = 2 /* the span exchange above: packed spans only, rank 0 renders the rest (sbx_render_span_*) */,
       SBX_MULTI_EXCHANGE_PEER_STORES = 3 /* no RCCL, no slabs: every rank renders its row-blocks in place into rank 0's frame through
                                             peer access — the exchange is the render kernels' own float4 stores over xGMI (16 B per
                                             pixel), rank 0 lands and scatters nothing.  UNMEASURED on more than one device: across distinct
                                             devices it returns SBX_ERR_UNSUPPORTED unless the environment has SBX_ENABLE_PEER_STORES=1
                                             (also without peer access).  The validated form of the same idea, with explicit ordering
                                             flags and across processes, is the store exchange sbx_shared_* above */ };
int sbx_multi_set_exchange(sbx_multi* m, int mode);
void sbx_multi_destroy(sbx_multi* m);
int sbx_multi_ranks(const sbx_multi* m);
int sbx_multi_uses_rccl(const sbx_multi* m);             /* 1: RCCL send/recv, 0: device / peer copies */
int sbx_multi_set_split(sbx_multi* m, int block_rows, int root_rounds, int rounds);
/* sbx_set_output_format on every rank: with SBX_FORMAT_RGBA8 `frame` of sbx_multi_render holds width * height 32-bit words and
 * every exchange form moves 4 bytes per pixel. */
int sbx_multi_set_output_format(sbx_multi* m, int format);
/* The two noise volumes of SBX_APP_CLOUDS_TEX (device memory on rank 0's device), handed to every rank; synchronous. */
int sbx_multi_set_noise_volumes(sbx_multi* m, int shape_size, const float* shape_rgba, int detail_size,
                                const float* detail_rgba);
/* One frame over all ranks into `frame` (height * width RGBA32F pixels in rank 0's device memory).  Asynchronous: the
 * work starts after what is already enqueued on `stream` (a stream of rank 0's device; NULL = its default stream) and
 * `stream` continues when the whole frame is in place.  Up to two frames may be in flight (alternate two streams and two
 * frames); the pixels are bit-identical to a one-GPU render. */
int sbx_multi_render(sbx_multi* m, int app, const sbx_uniforms* uni, const void* aux, float* frame, void* stream);
const char* sbx_multi_last_error(sbx_multi* m);
/* Self-test of the RCCL calls the multi-GPU path is made of, on ONE device: dlopen librccl, ncclCommInitAll({device}), one
 * grouped ncclSend / ncclRecv pair from the rank to itself on two streams, compare.  0 = ok; *step (may be NULL) names the
 * failing step: 1 load, 2 communicator, 3 buffers, 4 group call, 5 data. */
int sbx_multi_rccl_selftest(int device, int* step);

/* Internal-invariant faults.  The cooperative hash cache of APP_CLOUDS / APP_PLANET serves lattice-cell misses in a loop that
 * needs at most 64 rounds and is bounded at 4096; reaching the bound would mean wrong pixels.  Instead of passing silently the
 * wave sets a sticky word in pinned host memory (one per device): from then on every render call on a context of that device
 * returns SBX_ERR_FAULT — checked on entry, without synchronising — and sbx_last_error says why, until sbx_clear_fault (which
 * waits for the device first).  sbx_fault_status = the check on its own (e.g. after synchronising a frame).  The waits of the
 * store exchange (sbx_shared_frame_begin / _end) report a signal that never arrived the same way. */
int sbx_fault_status(sbx_ctx* ctx);
int sbx_clear_fault(sbx_ctx* ctx);

const char* sbx_last_error(sbx_ctx* ctx);
const char* sbx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SBX_H */
