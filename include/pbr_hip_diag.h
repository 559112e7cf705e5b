/*
 * pbr_hip_diag.h — diagnostic entry points of libpbrhip.so.
 *
 * Not part of the drop-in boundary (the reference has nothing comparable): they expose single
 * stages of the device path — the math layer, closest-hit traversal, BRDF evaluation, new-ray
 * sampling — so the parity tests can compare each stage with the oracle's hook of the same
 * shape (oracle/pt_oracle.h: orc_math, orc_trace_rays, orc_brdf_eval, orc_new_ray) instead of
 * only whole images.  Host pointers in, host pointers out; synchronous.
 */
#ifndef PBR_HIP_DIAG_H
#define PBR_HIP_DIAG_H

#include "pbr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* op: 0 sin, 1 cos, 2 tan, 3 acos, 4 atan, 5 pow( x, y ), 6 fract( sin( x ) * 43758.5453123 ) */
int pbr_diag_math( pbr_ctx* ctx, int op, const float* x, const float* y, int n, float* out );

/* Closest-hit traversal (pt_bvh.cl:82-123) of n rays {origin, dir} against the uploaded scene.
 * out_t[n], out_face[n], out_normal[3n] (0 on a miss), out_counts[2n] = {node visits, face tests}. */
int pbr_diag_trace( pbr_ctx* ctx, const float* rays, int n, float* out_t, int32_t* out_face, float* out_normal, uint32_t* out_counts );

/* BRDF evaluation with material 0 of the uploaded scene.  in: n x 16 {out_dir[3], in_dir[3],
 * normal[3], pad[7]}; out: n x 4 — BRDF 0 {brdf, u, pdf, 0}, BRDF 1 {spec, diff, dotHK1, pdf}. */
int pbr_diag_brdf( pbr_ctx* ctx, const float* in, int n, float* out );

/* getNewRay (pt_brdf.cl:344-378) with material 0.  in: n x 12 {origin[3], dir[3], normal[3], t,
 * seed, pad}; out: n x 8 {origin[3], dir[3], seed after, addDepth}. */
int pbr_diag_new_ray( pbr_ctx* ctx, const float* in, int n, float* out );

/* Traversal-only throughput probe: streams n rays {ox,oy,oz,-, dx,dy,dz,-} through a persistent
 * closest-hit kernel `repeats` times with `lds_slots` hot nodes staged in LDS; best kernel time in
 * *ms_out, hits {t, face bits} in out2. */
int pbr_diag_trace_stream( pbr_ctx* ctx, int lds_slots, const float* rays8, uint32_t n, int repeats, float* out2, double* ms_out );

/* Memory-counter calibration: reads `reads` elements of a zero-filled table of table_bytes in a
 * known pattern (mode 0 coalesced 16 B / lane stream, 1 random 16-B elements, 2 random 32-B
 * records) so that rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ_* can be interpreted. */
int pbr_diag_calibrate( pbr_ctx* ctx, int mode, uint64_t table_bytes, uint64_t reads, double* ms_out );

/* Which schedule rendered (the largest chunk of) the last render: "refill-lean", "refill-wide", "phased-lean",
 * "phased-wide", "phased-mid", "refill-mid", "phased-dual" ("refill-lean-phong" with Phong tessellation); *tuned = index of the schedule the auto-tuner
 * settled on for this scene + configuration, or -1 while it is still measuring (pbr_hip.hip, launch()). */
int pbr_diag_last_plan( pbr_ctx* ctx, char* name, size_t capacity, int* tuned );

/* pbr_build_bvh: the search radius the clustering builder used in this context's last build — 32 when the context was
 * configured for the reference's walk (or not configured yet) at the time of the call, 3 when it was configured for a
 * ray-ordered walk, or the "ploc_radius" knob (0: no build yet).  The protocol: pbr_configure (with the traversal the tree
 * will be walked in) BEFORE pbr_build_bvh; this call shows which one a build got. */
int pbr_diag_bvh_build_info( pbr_ctx* ctx, int* radius );

/* Device memory of the uploaded scene's arrays, bytes: [0] the node stream in the reference's order, [1] the streams of
 * the ray-ordered walk (0 unless such a mode is configured: six or eight times [0], compact records twice [0]), [2] the face records. */
int pbr_diag_scene_bytes( pbr_ctx* ctx, uint64_t out[3] );

/* The kernel behind pbr_diag_last_plan's schedule, as a profiler prints its symbol (without "void " and the argument
 * list): "ptk_f0::pathTracingDual<1, false, false>" — namespace ptk_f<flavour> (bit 0: ray-ordered walk, bit 1: native
 * arithmetic; csrc/pt_flavour.hpp), template arguments BRDF, SHADOW_RAYS, LIGHTS[, waves per SIMD[, PHONGTESS]]. */
int pbr_diag_last_kernel( pbr_ctx* ctx, char* name, size_t capacity );

/* What the schedule tuner measured for the plan in use: a launch of n frames costs fixed_ms + n * per_frame_ms (least
 * squares over its refinement launches of two lengths; fixed_ms = the ramp-up of a launch and the drain of its longest
 * paths).  PBR_ESTATE when there is no such fit (plan pinned before the tuner ran, single-length launches only). */
int pbr_diag_launch_fit( pbr_ctx* ctx, double* fixed_ms, double* per_frame_ms );

/* Experiment and test knobs, per context; value -1 = the built-in default.  The library reads NO environment variable:
 * lab scripts and tests that need a knob set it here (the Python harness maps PBR_* variables onto this call).
 *   "lds_slots"     cap of the node records a block stages in LDS (0 = none)
 *   "blocks_per_cu" run below the resident maximum
 *   "ph_park" / "ph_shade"  lane state machine thresholds (phased-dual: ph_park counts walks of up to 128 per wave); "park_eighths": the lock-step walk's park share
 *   "drain_mode"    bit 0 / 1: scale ph_park / ph_shade with the lanes still at work once the queue is empty
 *   "refill_batch"  lock-step kernels: lanes of a wave that wait with a finished unit before they take their next units
 *                   together (1 = every lane at once, as up to round 2)
 *   "chunk_frames"  cap of the frames per launch pair of pbr_render (tests: several launch pairs)
 *   "face_normals"  0 = recompute the face normal on every hit (takes effect at the next pbr_upload_scene)
 *   "bvh_builder"   pbr_build_bvh: 0 clustering (default), 1 round 1's radix tree; "ploc_radius": its search radius
 *   "tune_log"      1 = the schedule tuner logs its launches to stderr
 *   "deal_order"    the queue's dealing order: 0 always spatial, 1 always cost classes, 2 always expensive last (once learnt), -1 by the render call's size
 * Setting a knob rebuilds the plans and restarts the schedule tuner. */
int pbr_diag_set_knob( pbr_ctx* ctx, const char* name, int value );

/* Render with plan 0..6 (the order of pbr_diag_last_plan's *tuned: refill-lean, refill-wide, phased-lean, phased-wide,
 * phased-mid, refill-mid, phased-dual) from now on, without tuning; -1 hands the choice back to the tuner.  For the ranks of a
 * multi-GPU run: rank 0 tunes, broadcasts its *tuned, every rank pins it — all ranks then run the same schedule and
 * none is a straggler of the closing all-gather because its own timing noise picked a slower plan.  Survives
 * pbr_upload_scene / pbr_configure. */
int pbr_diag_pin_plan( pbr_ctx* ctx, int plan );

/* The dealing order of the work queue (csrc/pt_kernel.hpp, nextSlot).  The local tiles form a grid that is cut into 8 bands
 * of rows; band b's tiles are dealt in the order the stretch [band_first[b], band_first[b + 1]) of the table names them (local
 * tile indices; the table has one entry per local tile) — by four queue heads side by side, head s taking entries s, s + 4, ... of
 * the stretch.
 * Placement is for speed only: every (pixel, frame) unit is handed out exactly once in any order, images and counters
 * do not depend on it (tested).  get: which = 0 the spatial (or pinned) table, 1 the library's cost-classes table, 2 its
 * expensive-last table (PBR_ESTATE until they have been learnt); *count = entries, order[] filled when non-null.  set: with band_first = NULL `order` must hold,
 * per band, a permutation of that band's own tiles; with band_first[9] ANY partition of the local tiles into eight lists
 * (band_first[0] = 0, non-decreasing, band_first[8] = count; every tile once) — else PBR_EINVAL.  The table then stays in
 * force until pbr_configure (the library's own choice is off meanwhile); order = NULL: back to the library's choice. */
int pbr_diag_get_tile_order( pbr_ctx* ctx, int which, uint32_t* order, uint32_t capacity, uint32_t* count, uint32_t band_first[9] );
int pbr_diag_set_tile_order( pbr_ctx* ctx, const uint32_t* order, uint32_t count, const uint32_t* band_first );

/* The order the last render was dealt in: "spatial" — inside a band column by column —; "cost-classes" — per band eight
 * classes of falling cost, spatial inside a class: render calls of up to 128 Ki tiles x frames, of a shard (tile_world > 1) up
 * to 1 Mi —; "expensive-last" — per band its most expensive quarter last, spatial inside both parts: calls above 192 Ki (1 Mi) —,
 * the two once the library has learnt the
 * tiles' costs from a debug image (csrc/pbr_hip.hip, learnTileCosts); or "pinned" (pbr_diag_set_tile_order).  *learnt =
 * whether the cost orders exist.  Knob "deal_order": 0 always spatial, 1 always cost classes, 2 always expensive last (once
 * learnt), -1 by size. */
int pbr_diag_last_deal( pbr_ctx* ctx, char* name, size_t capacity, int* learnt );

/* How many frames of the configured size the schedule tuner wants to see before it settles (its launch lengths are
 * fixed in 1080p-frame equivalents, so a rank of an N-GPU run needs N times as many): a benchmark renders that many
 * before it starts its clock. */
int pbr_diag_tune_budget( pbr_ctx* ctx, uint32_t* frames );

/* The path-tracing launches of the last render: their summed duration (HIP events around each one)
 * and their number.  A multi-frame render is one launch unless its per-frame result buffer would
 * exceed 16 GiB; pbr_last_kernel_ms covers the whole render, foldFrames launches included. */
int pbr_diag_last_trace( pbr_ctx* ctx, double* trace_ms, uint32_t* launches );

/* All 16 device counter slots: [0..3] = pbr_counters; [4..15] are written only by experiment
 * builds (-DPBR_LAB_HOOKS, lab/src/pt_lab_hooks.hpp) and stay 0 otherwise. */
int pbr_diag_raw_counters( pbr_ctx* ctx, uint64_t out[16] );

/* Loop-bound trips of the LAST render, recorded by a PBR_GUARD build ([0] unused since round 3, [1] path loop,
 * [2] traversal); all zero in a normal build.  A render in which a bounded loop gave up returns PBR_EDEVICE. */
int pbr_diag_guard_trips( pbr_ctx* ctx, uint32_t out[3] );

#ifdef __cplusplus
}
#endif
#endif
