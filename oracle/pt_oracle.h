/*
 * pt_oracle.h — CPU oracle for the path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or
 * executed by the product (physically-based-rendering_amd/, include/).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (sebadorn/Physically-based-Rendering) ships no
 * tests, golden images or known-answer vectors for this path, and its kernels
 * (OpenCL C, source/opencl/ *.cl) cannot be built in the build container without
 * writing stand-ins for an OpenCL runtime/builtin library (no POCL, no libclc,
 * amdocl64 has 0 devices; the sources also fail to compile under clang because
 * of two writes through `const Scene*`, pt_bvh.cl:23 and :89).  This oracle is
 * therefore a line-by-line restatement of the reference algorithm, each
 * function citing the file:line it follows; the only in-tree known answer
 * (1082 faces -> 1265 flat BVH nodes, pathtracing.cl:75-76) pins the host-side
 * BVH flattening, not this file.
 *
 * All arithmetic is IEEE-754 binary32 with NO implicit contraction
 * (-ffp-contract=off); fused multiply-adds appear only where written as fmaf().
 * The OpenCL builtins whose precision is implementation-defined (native_sin,
 * native_cos, native_tan, native_recip, native_divide, native_sqrt,
 * fast_normalize) and the ones with a ULP budget (pow <=16, acos <=4, atan <=5)
 * are given ONE bit-exact definition in "deterministic math" below; the HIP
 * kernels implement the same definitions independently, so HIP-vs-oracle parity
 * is bit-for-bit (compared numerically: -0 == +0, NaN == NaN).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- wire formats: identical to the reference's host structs
 *      (source/PathTracer.h:25-73) / kernel structs (opencl/pt_header.cl:24-109) */

typedef struct { float x, y, z, w; } orc_float4;
typedef struct { uint32_t x, y, z, w; } orc_uint4;

/* camera_cl, PathTracer.h:25-32 — cl_float3 is 16 bytes; 80 bytes total */
typedef struct {
	orc_float4 eye, w, u, v;
	int32_t focusPoint[2];
	float lense[2]; /* [0] focal length, [1] aperture */
} orc_camera;

/* bvhNode_cl, PathTracer.h:69-72 */
typedef struct {
	orc_float4 bbMin; /* w: first face index, or -1 for an inner node */
	orc_float4 bbMax; /* w: second face index / -1 (leaf); miss link / -1 (inner) */
} orc_bvh_node;

/* light_cl, PathTracer.h:39-43 */
typedef struct { orc_float4 pos, rgb, data; } orc_light;

/* material_schlick_rgb, PathTracer.h:45-54 (BRDF 0): data = d, Ni, p, rough */
typedef struct { float data[4]; orc_float4 rgbDiff, rgbSpec; } orc_material_schlick;

/* material_shirley_ashikhmin_rgb, PathTracer.h:56-65 (BRDF 1): data = d,Ni,nu,nv,Rs,Rd,-,- */
typedef struct { float data[8]; orc_float4 rgbDiff, rgbSpec; } orc_material_sa;

/* The compile-time macros CL::setValues bakes into the kernel
 * (source/CL.cpp:626-705, opencl/pt_header.cl:1-20), as run-time values. */
typedef struct {
	int32_t width, height;      /* IMG_WIDTH, IMG_HEIGHT */
	int32_t brdf;               /* BRDF: 0 Schlick, 1 Shirley-Ashikhmin */
	int32_t shadow_rays;        /* SHADOW_RAYS */
	int32_t max_depth;          /* MAX_DEPTH */
	int32_t max_added_depth;    /* MAX_ADDED_DEPTH */
	int32_t samples;            /* SAMPLES */
	int32_t num_nodes;          /* BVH_NUM_NODES */
	int32_t num_lights;         /* NUM_LIGHTS */
	float anti_aliasing;        /* ANTI_ALIASING */
	float sky_light[4];         /* SKY_LIGHT */
	float phong_tessellation;   /* PHONGTESS_ALPHA; PHONGTESS = ( > 0 ) as CL::setValues derives it (CL.cpp:651) */
	int32_t traversal;          /* NOT a reference constant.  0: the reference's walk (pt_bvh.cl:82-123).  1 / 2: the product's
	                             * opt-in ray-ordered walk over the same flat tree, six / eight orders (pt_oracle.c, "Ray-ordered
	                             * walk"); needs scene.walk_links / walk_first */
} orc_config;

typedef struct {
	const orc_bvh_node* bvh;
	const orc_uint4* facesV;
	const orc_float4* vertices;
	const void* materials;      /* orc_material_schlick[] or orc_material_sa[] by cfg.brdf */
	const orc_light* lights;
	uint32_t num_faces, num_vertices, num_materials;
	const orc_uint4* facesN;    /* normal indices per face; only read when cfg.phong_tessellation > 0 */
	const orc_float4* normals;
	uint32_t num_normals;
	const int32_t* walk_links;  /* cfg.traversal != 0 only: orc_build_walk_orders' successor table, K x num_nodes x {hit, miss} */
	const int32_t* walk_first;  /* ... and the K first nodes */
} orc_scene;

/* Traversal counters summed over the rendered pixels:
 * node visits, triangle tests, shaded hits, camera paths. */
typedef struct { uint64_t nodes, tris, hits, paths; } orc_counters;

/* One launch of the reference kernel `pathTracing` (pathtracing.cl:207-334) over
 * rows [y0,y1).  imageIn/imageOut/imageDebug are W*H RGBA32F, row 0 = bottom.
 * imageDebug may be NULL.  counters may be NULL (else accumulated into).
 * threads<=1: serial; else OpenMP over 8-row bands. */
void orc_render_frame(
	const orc_scene* scene, const orc_config* cfg, const orc_camera* cam,
	float seed, float pixelWeight, float pxDim,
	const float* imageIn, float* imageOut, float* imageDebug,
	int y0, int y1, int threads, orc_counters* counters );

/* Closest-hit traversal of a ray batch (pt_bvh.cl:82-123).  rays: n x {ox,oy,oz,dx,dy,dz};
 * out_t[n], out_face[n], out_normal[3n], out_counts[2n] = {nodes,tris}. */
void orc_trace_rays(
	const orc_scene* scene, const orc_config* cfg, const float* rays, int n,
	float* out_t, int32_t* out_face, float* out_normal, uint32_t* out_counts );

/* The product's opt-in ray-ordered walk (NOT the reference's algorithm — the reference has one fixed order): successor
 * tables over the reference's flat node array.  scheme 1: six orders (dominant axis x sign), 2: eight (sign octant, each
 * container sorted on its own axis).  links: orc_walk_order_count( scheme ) x numNodes x 2 int32, first: one per order.
 * Returns 0, or -1 for an unknown scheme. */
int orc_walk_order_count( int scheme );
int orc_build_walk_orders( const orc_bvh_node* bvh, int numNodes, int scheme, int32_t* links, int32_t* first );

/* Analysis aids (not part of any parity claim): longest single closest-hit walk in node visits. */
void orc_debug_set_walk_max( uint32_t* slot );
/* analysis aid, single-threaded runs: one word per closest-hit walk of the ray-ordered mode — order | path depth << 4 | node visits << 12 */
void orc_debug_set_walk_log( uint32_t* log, uint32_t cap, uint32_t* count );

/* Deterministic math layer, elementwise over n values (for ULP tests).
 * op: 0 sin, 1 cos, 2 tan, 3 acos, 4 atan, 5 pow(x,y), 6 rand-hash fract(sin(x)*43758.5453123) */
void orc_math( int op, const float* x, const float* y, int n, float* out );

/* BRDF unit hooks (for parity tests against the HIP diag kernels).
 * in: n x 16 floats {out_dir[3], in_dir[3], normal[3], pad}; mtl as in scene. */
void orc_brdf_eval(
	int brdf, const void* mtl, const float* in, int n,
	float* out /* n x 4: brdf0 {brdf,u,pdf,0}; brdf1 {spec,diff,dotHK1,pdf} */ );

/* getNewRay (pt_brdf.cl:344-378): in n x 12 {origin[3], dir[3], normal[3], t, seed, pad};
 * out n x 8 {origin[3], dir[3], seed_after, addDepth}. */
void orc_new_ray( int brdf, const void* mtl, const float* in, int n, float* out );

#ifdef __cplusplus
}
#endif
#endif
