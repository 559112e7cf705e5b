/*
 * pbr_multi.h — C ABI of libpbrmulti.so: the path on N GPUs of one node from ONE process.
 *
 * Not in the reference: it is one process, one thread, one `CL*` per PathTracer (source/qt/GLWidget.cpp:33,504-517,
 * source/PathTracer.cpp:150-153) on one device (source/CL.cpp:355,521).  A viewer that wants N GPUs keeps that shape: it
 * holds ONE pbr_multi where it held one pbr_ctx, and every call below is the N-context form of the pbr_hip.h call of the
 * same name.  Inside: one pbr_ctx and one host thread per device; the frame's 8x8-pixel tiles dealt round-robin to the
 * contexts (pbr_config.tile_world / tile_rank, filled in here), the scene replicated, no collective on the data path; a
 * render ends with ONE all-gather of the compact per-rank tile buffers (pbr_export_tiles -> ncclAllGather over xGMI ->
 * pbr_import_tiles), after which every device holds the full frame.  The schedule tuners of the N contexts run at the
 * same time and vote; depth of field's one cross-pixel value is handed from its owner to every context.
 * (bench.py --gpus N does the same with one PROCESS per GPU through torch.distributed; this is the form a C++ host links.)
 *
 * Status codes and error convention: pbr_hip.h's (0 = PBR_OK; pbr_multi_last_error names the rank that failed).  A rank that fails
 * before the exchange tells the others at a meeting point in front of the collective, and then NO rank enters it (a collective one
 * rank never joins would hang the rest): the call returns the failing rank's status.
 * Threading: calls on one pbr_multi are not re-entrant; the library's own threads are internal.
 */
#ifndef PBR_MULTI_H
#define PBR_MULTI_H

#include "pbr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pbr_multi pbr_multi;

/* How the tile buffers travel.  RCCL: ncclCommInitAll over the listed devices (which must be distinct) and one
 * ncclAllGather per render.  PEER_COPY: every context copies every other context's buffer with hipMemcpyPeerAsync — no
 * communicator, works with the same device listed more than once (the N-rank shape rehearsed on one GPU: tests). */
enum { PBR_MULTI_RCCL = 0, PBR_MULTI_PEER_COPY = 1 };

/* One context + one host thread per entry of `devices` (HIP ordinals). */
int pbr_multi_create( const int* devices, int count, int transport, pbr_multi** out );
void pbr_multi_destroy( pbr_multi* m );
const char* pbr_multi_last_error( const pbr_multi* m );
int pbr_multi_size( const pbr_multi* m );
/* Rank r's context, for calls this header does not wrap (pbr_get_counters, pbr_diag_*, pbr_read_display ...). */
pbr_ctx* pbr_multi_context( pbr_multi* m, int rank );

/* pbr_upload_scene on every context, concurrently (the scene is replicated). */
int pbr_multi_upload_scene( pbr_multi* m, const pbr_scene_desc* scene );
/* pbr_configure on every context with tile_world = count, tile_rank = its rank (the caller's values are ignored);
 * allocates the exchange buffers. */
int pbr_multi_configure( pbr_multi* m, const pbr_config* cfg );
int pbr_multi_reset_accum( pbr_multi* m );

/* The schedule tuners of all contexts at the same time: every rank renders its own share in calls of `frames_per_call`
 * frames until its tuner has settled (pbr_diag_tune_budget frames), then the plan most ranks settled on (ties: the lowest
 * rank's) is pinned on all of them — all ranks run one schedule, none is the straggler of the closing all-gather.
 * *plan = that plan, votes[count] = every rank's own choice (either may be NULL).  Leaves the accumulation reset. */
int pbr_multi_tune( pbr_multi* m, uint32_t frames_per_call, float pxDim, const pbr_camera* cam, int* plan, int* votes );

/* pbr_render on every context at the same time (each its own tiles, all n_frames), then the all-gather: afterwards every
 * device holds the full frame (pbr_multi_read_full).  `gather` = 0 leaves the exchange out (a caller that accumulates
 * several renders before it looks at the frame: pbr_multi_gather). */
int pbr_multi_render( pbr_multi* m, uint32_t first_sample_count, uint32_t n_frames, const float* seeds, float pxDim, const pbr_camera* cam, int gather );
/* The reference's per-frame sequence (PathTracer.cpp:59-71) on N devices: with a focus point set, the owner of the focus
 * pixel's tile hands its previous-frame distance to every context first (pbr_get_focus_depth / pbr_set_focus_depth);
 * then pbr_render_frame everywhere; then pbr_accumulate (imageOut becomes the next frame's imageIn) when `accumulate`. */
int pbr_multi_render_frame( pbr_multi* m, float seed, float pixelWeight, float pxDim, const pbr_camera* cam, int accumulate, int gather );
int pbr_multi_gather( pbr_multi* m );
/* The gathered frame as device `rank` holds it: row-major W x H RGBA32F, row 0 = bottom. */
int pbr_multi_read_full( pbr_multi* m, int rank, float* rgba );

/* Of the last render: per rank the host-side milliseconds of its render call and of its share of the exchange (export +
 * all-gather + scatter); either pointer may be NULL. */
int pbr_multi_timings( const pbr_multi* m, double* render_ms, double* gather_ms );

#ifdef __cplusplus
}
#endif
#endif
