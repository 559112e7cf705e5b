/*
 * pbr_hip.h — C ABI of the MI355X path-tracing core (libpbrhip.so).
 *
 * Drop-in boundary: this library replaces what the reference reaches through its `CL`
 * class (source/CL.h:20-83) as driven by `PathTracer` (source/PathTracer.cpp) — the OpenCL
 * context, the device buffers, the JIT-compiled `pathTracing` kernel
 * (source/opencl/pathtracing.cl:207-334) and the per-frame launch.  `CL` is not a stable
 * plugin ABI (cl_mem / cl_kernel leak through every signature and kernel constants travel
 * as source-text substitutions), so the entry points mirror the reference's CALL SEQUENCE;
 * each one names the reference call it stands for.  Plain pointers and sizes only.
 *
 * Conventions
 *  - every function returns 0 on success, a negative PBR_E* code otherwise;
 *    pbr_last_error( ctx ) holds the message.  Nothing calls exit() (the reference does:
 *    source/CL.cpp:78,210,349,442,524,541,565).
 *  - inputs are borrowed for the duration of the call and copied to the device
 *    (CL_MEM_COPY_HOST_PTR semantics, source/CL.h:26-33); outputs are copied into
 *    caller-provided buffers (source/CL.cpp:581-594).
 *  - one context = one HIP device + one stream; calls on a context are synchronous and not
 *    re-entrant; separate contexts may be driven from separate threads / processes.
 *  - there is NO CPU fallback: without a usable HIP device pbr_create fails.
 */
#ifndef PBR_HIP_H
#define PBR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The layout of this header's structs and the meaning of its calls, as a number: bumped whenever a struct grows or an
 * entry point changes (round 5 grew pbr_config from 60 to 68 bytes).  A caller that loads the library at run time — or
 * links a libpbrhip.so it did not build — compares pbr_abi_version() with the PBR_ABI_VERSION it was compiled against
 * BEFORE it hands the library a struct: pbr_configure reads sizeof( pbr_config ) bytes of ITS version. */
#define PBR_ABI_VERSION 6
uint32_t pbr_abi_version( void );

#define PBR_OK 0
#define PBR_EINVAL (-1)   /* bad argument / scene fails validation */
#define PBR_EDEVICE (-2)  /* HIP error (message has the HIP error string) */
#define PBR_ESTATE (-3)   /* call sequence violated (e.g. render before upload/configure) */

typedef struct pbr_ctx pbr_ctx;

/* ---- wire formats: the reference's host structs, bit for bit (source/PathTracer.h:25-73) */

typedef struct { float x, y, z, w; } pbr_float4;
typedef struct { uint32_t x, y, z, w; } pbr_uint4;

/* camera_cl, PathTracer.h:25-32 (cl_float3 occupies 16 bytes) — 80 bytes */
typedef struct {
	pbr_float4 eye, w, u, v;
	int32_t focusPoint[2];   /* (-1,-1): no depth of field */
	float lense[2];          /* focal length, aperture */
} pbr_camera;

/* bvhNode_cl, PathTracer.h:69-72 — 32 bytes.  bbMin.w: first face index or -1 (inner);
 * bbMax.w: second face index or -1 (leaf) / miss link or -1 (inner). */
typedef struct { pbr_float4 bbMin, bbMax; } pbr_bvh_node;

/* light_cl, PathTracer.h:39-43 — 48 bytes.  data.x: 1 point, 2 orb; data.y: orb radius */
typedef struct { pbr_float4 pos, rgb, data; } pbr_light;

/* material_schlick_rgb, PathTracer.h:45-54 — 48 bytes; data = d, Ni, p, rough */
typedef struct { float data[4]; pbr_float4 rgbDiff, rgbSpec; } pbr_material_schlick;

/* material_shirley_ashikhmin_rgb, PathTracer.h:56-65 — 64 bytes; data = d, Ni, nu, nv, Rs, Rd, -, - */
typedef struct { float data[8]; pbr_float4 rgbDiff, rgbSpec; } pbr_material_sa;

/* The seven arrays PathTracer::initOpenCLBuffers uploads (PathTracer.cpp:357-380 vertices /
 * normals, :238-347 bvh / facesV / facesN, :435-519 materials, :387-428 lights). */
typedef struct {
	const pbr_bvh_node* bvh;      uint32_t num_nodes;       /* -> #BVH_NUM_NODES# */
	const pbr_uint4* facesV;      /* {v0, v1, v2, material}, leaf order */
	const pbr_uint4* facesN;      /* may be NULL (only Phong tessellation reads it) */
	uint32_t num_faces;
	const pbr_float4* vertices;   uint32_t num_vertices;
	const pbr_float4* normals;    uint32_t num_normals;     /* may be NULL / 0 */
	const void* materials;        uint32_t num_materials;   /* pbr_material_schlick[] if brdf == 0, pbr_material_sa[] if 1 */
	uint32_t brdf;                /* which material layout `materials` uses */
	const pbr_light* lights;      uint32_t num_lights;      /* -> #NUM_LIGHTS#; lights may be NULL when 0 */
} pbr_scene_desc;

/* The constants CL::setValues / setReplacement bake into the kernel source
 * (source/CL.cpp:637-678, PathTracer.cpp:210,338,472,515). */
typedef struct {
	uint32_t width, height;       /* IMG_WIDTH, IMG_HEIGHT; multiples of 8 (opencl.localgroupsize) */
	uint32_t brdf;                /* BRDF: 0 Schlick, 1 Shirley-Ashikhmin; must match the uploaded materials */
	uint32_t shadow_rays;         /* SHADOW_RAYS */
	uint32_t max_depth;           /* MAX_DEPTH */
	uint32_t max_added_depth;     /* MAX_ADDED_DEPTH */
	uint32_t samples;             /* SAMPLES (paths per pixel per frame) */
	float anti_aliasing;          /* ANTI_ALIASING */
	float phong_tessellation;     /* PHONGTESS_ALPHA; > 0 = PHONGTESS on (pt_phongtess.cl; needs facesN / normals in the scene), 0 = flat triangles.
	                               * Phong tessellation pins its own schedule: the lock-step kernel in the 128-register budget is the only one built
	                               * with the patch intersection (the cubic solve spills in every budget, least in this one, and a state machine that parks lanes on
	                               * leaves would have to carry the patch normal through the park) — the tuner and pbr_diag_pin_plan do not apply. */
	float sky_light[4];           /* SKY_LIGHT */
	/* Tile sharding (not in the reference, which is single-device): this context renders the
	 * 8x8-pixel tiles whose position p in the dealing order has p % tile_world == tile_rank, where tile (tx, ty) has
	 * p = ty * tilesX + ( tx + 5 * ty ) % tilesX (row-major with row ty rotated by 5 * ty columns: plain row-major
	 * order would hand a rank whole tile columns whenever tilesX is a multiple of tile_world, and columns do not cost
	 * the same).  Local tile j of a rank is the tile at p = j * tile_world + tile_rank.  1 / 0 = everything, p = tile. */
	uint32_t tile_world, tile_rank;
	/* Two opt-in modes that are NOT reference constants; 0 / 0 (the value-initialised struct) is the reference's behaviour.
	 * traversal  PBR_WALK_REFERENCE (0): the reference's walk — a hit continues at index + 1 (pt_bvh.cl:102,112), the
	 *            child with the bigger surface area first whatever the ray (accelstructures/BVH.cpp:335-343).
	 *            PBR_WALK_SIX_ORDERS (1): the same flat tree, same boxes, leaves and per-visit arithmetic, but the children
	 *            of every container are visited in the order of their box centres along the ray direction's dominant axis
	 *            (six successor sets, chosen once per ray; still stackless).  Closest hits are the same faces at the same t
	 *            except where two faces tie exactly; the node / face-test counters and the debug image are this walk's own.
	 *            PBR_WALK_EIGHT_ORDERS (2): likewise with eight successor sets, one per sign octant of the ray direction;
	 *            every container orders its children along ITS axis (the one their centres spread furthest on).
	 *            Both cost node memory (6 / 8 streams of 32-byte records instead of one; they are built by the pbr_configure /
	 *            pbr_upload_scene call that completes "scene + such a mode" — which is also the call that fails when a tree
	 *            cannot be walked that way — and freed when another traversal is configured) and ~40 bytes of host memory
	 *            per node for the copy of the tree they are built from (kept from every upload).
	 *            PBR_WALK_EIGHT_ORDERS_COMPACT (3, round 6): the eight-order walk — same visits, same counters, same hits —
	 *            over ONE 64-byte record per node shared by the eight orders (two hit candidates picked by a sign bit,
	 *            eight `next` words): twice the reference stream's node memory instead of eight times, and scenes up to
	 *            the wire format's own 2^24 nodes (eight streams: 8.3 M); one more 4-byte load and five more vector
	 *            instructions per visit.
	 * arith      PBR_ARITH_EXACT (0): every builtin has one correctly rounded / fixed definition.
	 *            PBR_ARITH_NATIVE (1): what the reference asks its device for — native_sin / native_cos / native_tan /
	 *            native_recip / native_divide / native_sqrt (pt_utils.cl:39-44, pt_brdf.cl:306-321, pt_intersect.cl:104,
	 *            pt_bvh.cl:83) as the gfx950 instructions, pow through v_log_f32 / v_exp_f32.  Images then agree with the
	 *            exact mode statistically, not bit for bit. */
	uint32_t traversal;
	uint32_t arith;
} pbr_config;

#define PBR_WALK_REFERENCE 0u
#define PBR_WALK_SIX_ORDERS 1u
#define PBR_WALK_EIGHT_ORDERS 2u
#define PBR_WALK_EIGHT_ORDERS_COMPACT 3u
#define PBR_ARITH_EXACT 0u
#define PBR_ARITH_NATIVE 1u

/* Traversal counters (the reference's debugColor.y / .x, pt_bvh.cl:89,23, as exact integers,
 * plus shaded hits and camera paths) summed over everything rendered since the last reset. */
typedef struct { uint64_t nodes, tris, hits, paths; } pbr_counters;

/* Which modes this build of the library carries: 1 if every plan's kernels for ( traversal, arith ) were linked in, 0 if not
 * (a build may leave a mode's translation units out), -1 for values that are no mode.  No device
 * is needed.  pbr_configure refuses (PBR_ESTATE, with a message) a mode whose kernels were not built. */
int pbr_mode_built( uint32_t traversal, uint32_t arith );

/* new CL() — platform / device / context / profiling queue (source/CL.cpp:10-24). */
int pbr_create( int device, pbr_ctx** out );
/* ~CL() (source/CL.cpp:30-52) */
void pbr_destroy( pbr_ctx* ctx );
const char* pbr_last_error( const pbr_ctx* ctx );

/* CL::createBuffer x 7 (PathTracer.cpp:357-519).  Validates every index the kernel will
 * follow (links, face, vertex and material indices — the reference reads out of bounds for
 * material -1, ObjParser.cpp:140,192) and re-lays the arrays out for CDNA4. */
int pbr_upload_scene( pbr_ctx* ctx, const pbr_scene_desc* scene );
/* The checks of pbr_upload_scene alone, without a context or a device (a host-side loader can vet a foreign BVH
 * before it goes anywhere near a GPU): array sizes, every node's face / link words — miss links must be integers in
 * [-1, N) that point FORWARD (the stackless walk of pt_bvh.cl:96-117 keeps no visited set; a backward link would make
 * it circle forever) — leaf pairs k, k + 1, the last node a leaf, vertex and material indices of every face.
 * Returns PBR_OK or PBR_EINVAL with the reason in message[capacity]. */
int pbr_validate_scene( const pbr_scene_desc* scene, char* message, size_t capacity );

/* CL::loadProgram + createKernel + initKernelArgs (PathTracer.cpp:225-229, :88-125) and
 * initOpenCLBuffers_Textures (:525-533): selects the kernel variant, allocates the three
 * W x H RGBA32F images and zero-fills the input image. */
int pbr_configure( pbr_ctx* ctx, const pbr_config* cfg );

/* CL::updateImageReadOnly( imageIn ) (PathTracer.cpp:61): rgba = W*H*4 floats, row 0 = bottom. */
int pbr_write_input( pbr_ctx* ctx, const float* rgba );
/* Zero the input image and the counters (PathTracer::resetSampleCount + the zero-filled
 * mTextureOut of initOpenCLBuffers_Textures). */
int pbr_reset_accum( pbr_ctx* ctx );

/* clPathTracing (PathTracer.cpp:43-52): set args 0 (seed), 1 (pixelWeight), 2 (pxDim),
 * 3 (camera); CL::execute; CL::finish.  Reads imageIn, writes imageOut and imageDebug. */
int pbr_render_frame( pbr_ctx* ctx, float seed, float pixelWeight, float pxDim, const pbr_camera* cam );

/* Replaces the reference's readImageOutput -> host -> updateImageReadOnly round trip
 * (PathTracer.cpp:61,66): imageOut becomes the next frame's imageIn, on the device. */
int pbr_accumulate( pbr_ctx* ctx );

/* n_frames x { pbr_render_frame( seeds[k], n/(n+1) with n = first_sample_count + k ) ;
 * pbr_accumulate } as one launch over all (pixel, frame) units + one launch that folds the frames into the
 * running mean in frame order — bit-identical to the
 * frame-by-frame sequence.  Needs cam->focusPoint < 0 (depth of field reads another pixel's
 * previous-frame value, pathtracing.cl:58-65): returns PBR_EINVAL otherwise.  The result is
 * left in imageOut AND imageIn (ready to continue). */
int pbr_render( pbr_ctx* ctx, uint32_t first_sample_count, uint32_t n_frames, const float* seeds, float pxDim, const pbr_camera* cam );

/* CL::readImageOutput( imageOut ) / ( imageDebug ) (PathTracer.cpp:66-67).  With tile sharding
 * only this rank's tiles are meaningful (others read 0). */
int pbr_read_output( pbr_ctx* ctx, float* rgba );
int pbr_read_debug( pbr_ctx* ctx, float* rgba );

/* The denoise half of the display step.  The reference's noise filter was never finished
 * (source/opencl/noise_filtering.cl:386-401,417 are TODOs; PathTracer.cpp:155-160 never launches it), so there is no
 * behaviour to match: this keeps its shape — per-pixel first-hit feature buffers (position, normal, texture colour;
 * :441-455), several passes, feature distances over standard deviations (:6-7) — and fills the TODOs with the
 * edge-avoiding a-trous wavelet filter: pass k weighs 5 x 5 taps 2^k pixels apart by
 *   B3-spline * exp( -( |dc|^2 / (sigma_color / 2^k)^2 + |dn|^2 / sigma_normal^2 + |dx|^2 / (sigma_world * 2^k * pxDim * t)^2
 *                       + |da|^2 / sigma_albedo^2 ) ),
 * c the colour, n the first-hit normal (unit, towards the viewer), x the first-hit position, t the centre pixel's
 * first-hit distance (so sigma_world is in pixel footprints), a the first-hit diffuse colour (Kd); taps across the
 * hit / miss divide are left out; a standard deviation of 0 switches its term off.  Features come from one primary ray
 * through every pixel centre over the uploaded scene (orb lights are not in that pass).
 *   rgba      host, width x height x 4 floats, row 0 = bottom like pbr_read_output: filtered colour, .w = the
 *             accumulated first-hit distance, unfiltered.  The accumulation itself is not modified.
 *   features  optional (NULL): host, 3 x width x height x 4 floats — position {x, y, z, t (INFINITY: miss)},
 *             normal {x, y, z, hit ? 1 : 0}, albedo {Kd, material index (-1: miss)}.
 * With tile sharding the gathered frame is filtered: call pbr_import_tiles first.  pbr_last_kernel_ms reports the device
 * time of feature pass + filter. */
typedef struct pbr_denoise_params {
	uint32_t passes;      /* 1 .. 8; 5 passes span 61 pixels */
	float sigma_color;
	float sigma_normal;
	float sigma_world;
	float sigma_albedo;
} pbr_denoise_params;
int pbr_denoise( pbr_ctx* ctx, float pxDim, const pbr_camera* cam, const pbr_denoise_params* params, float* rgba, float* features );

/* Opt-in fast BVH build on the device: faces in Morton order, clustered bottom-up by
 * surface area, at most 2 faces per leaf, emitted in the reference's flat format — what BVH::getNodes + the packing
 * loops of PathTracer::initOpenCLBuffers_BVH / _Faces (PathTracer.cpp:238-352) produce: `nodes_out` in depth-first order
 * with miss links, `facesV_out` / `facesN_out` = the input faces re-ordered into leaf order.  NOT the reference's builder
 * (accelstructures/BVH.cpp, replicated on the host in host/bvh_builder.cpp): same format, different tree, so images agree
 * statistically, not bit for bit.  All pointers are host memory; nodes_out needs pbr_bvh_node_capacity( num_faces )
 * entries (2 * num_faces - 1: the count actually used comes back in *num_nodes_out).  Vertices must be finite.
 * The clustering's search radius follows the traversal the context is configured with AT THE TIME OF THE CALL (32 for the
 * reference's walk or an unconfigured context, 3 for a ray-ordered one, which such a tree is best walked in: DESIGN.md
 * section 5.4) — so the protocol is pbr_configure( the traversal the tree will be walked in ) BEFORE pbr_build_bvh;
 * pbr_diag_bvh_build_info (pbr_hip_diag.h) tells which radius a build got, the "ploc_radius" knob sets it outright.
 * pbr_last_kernel_ms then reports the device time of the build. */
uint32_t pbr_bvh_node_capacity( uint32_t num_faces );
int pbr_build_bvh( pbr_ctx* ctx, const pbr_float4* vertices, uint32_t num_vertices, const pbr_uint4* facesV, const pbr_uint4* facesN,
                   uint32_t num_faces, pbr_bvh_node* nodes_out, uint32_t* num_nodes_out, pbr_uint4* facesV_out, pbr_uint4* facesN_out );

/* Depth of field with tile sharding.  Every pixel reads the previous-frame distance (.w) of the focus pixel
 * cam->focusPoint (pathtracing.cl:58-65) — the one cross-pixel dependency of the path — and with tile_world > 1 that
 * pixel's tile lives on one rank only.  Per frame: every rank calls pbr_get_focus_depth( x, y ); the rank with
 * *owned = 1 broadcasts *t (one float: ncclBroadcast / MPI_Bcast); every rank passes it to pbr_set_focus_depth and
 * then calls pbr_render_frame with the same camera.  The value is consumed by that frame.  With tile_world = 1 none of
 * this is needed (the kernel reads the pixel itself). */
int pbr_get_focus_depth( pbr_ctx* ctx, int x, int y, float* t, int* owned );
int pbr_set_focus_depth( pbr_ctx* ctx, float t );

/* The display step the reference leaves to GL (shader/pathtracing.frag:11-15 writes the linear colour to an
 * 8-bit framebuffer): imageOut as width x height RGBA8, each channel floor( clamp( c, 0, 1 ) * 255 + 0.5 ), NaN -> 0,
 * alpha 255, converted on the device (4 B instead of 16 B per pixel over PCIe).  top_row_first = 0: row 0 is the
 * bottom of the image, as pbr_read_output and GL have it; 1: top row first, as image files want it. */
int pbr_read_display( pbr_ctx* ctx, uint8_t* rgba8, int top_row_first );

int pbr_get_counters( pbr_ctx* ctx, pbr_counters* out );
/* CL::getKernelTimes (source/CL.cpp:480-488): device time of the last launch, HIP events. */
double pbr_last_kernel_ms( const pbr_ctx* ctx );

/* ---- multi-GPU tile exchange (device pointers; the caller runs the RCCL all-gather — or lets pbr_multi.h do all of it:
 * N contexts in one process, one host thread each, ncclCommInitAll + one ncclAllGather per render) ---- */

/* Bytes of this rank's compact tile buffer: ceil( tiles / tile_world ) * 1024. */
uint64_t pbr_tile_bytes( const pbr_ctx* ctx );
/* Copy this rank's tiles of imageOut (local tile j = the tile at dealing position j * tile_world + tile_rank,
 * see pbr_config; 64 pixels x RGBA32F each) to d_dst on this context's stream and wait. */
int pbr_export_tiles( pbr_ctx* ctx, void* d_dst );
/* d_all = tile_world consecutive rank buffers (all-gather layout).  Scatters every tile into
 * this context's full-frame buffer; the context's own sharding is unchanged. */
int pbr_import_tiles( pbr_ctx* ctx, const void* d_all );
/* The full frame assembled by the last pbr_import_tiles, row-major W x H RGBA32F. */
int pbr_read_full( pbr_ctx* ctx, float* rgba );

#ifdef __cplusplus
}
#endif
#endif
