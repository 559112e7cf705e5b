// libpbrhip.so — implementation of include/pbr_hip.h: context, scene upload / re-layout,
// configuration, launches, read-back and the multi-GPU tile exchange helpers.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "pbr_hip.h"
#include "pbr_hip_diag.h"
#include "pt_aux.hpp"
#include "pt_instances.hpp"
// The schedules that were measured and rejected in rounds 1 - 2 (tile, batched, wavefront, pooled: DESIGN.md 5.1b / 5.1d)
// are no longer part of this source; their kernels are kept for the record under lab/src/.
#include "bvh_build.hpp"
#include "pt_denoise.hpp"

using ptk::DevParams;

typedef void ( *KernelFn )( const ptk::DevParams );

// A plan = kernel + persistent grid + LDS split (launch()).  Built once per scene + configuration and kept in the
// context: six occupancy queries and six function-attribute calls are host time a frame-by-frame caller would pay
// on every frame.
struct Plan {
	KernelFn kernel = nullptr;
	int blocks = 0, blockThreads = 0, numHot = 0, park = 0, shade = 0, parkEighths = 4;
	size_t ldsBytes = 0;
	const char* name = "";
	char kernelName[96] = "";   // the kernel's symbol as a profiler prints it (without "void " and the argument list)
};

// Experiment and test knobs (pbr_diag_set_knob, include/pbr_hip_diag.h): per context, -1 = the built-in value.  This
// library reads no environment variable; lab scripts and tests set knobs through the diagnostic entry point.
struct Knobs {
	int ldsSlots = -1;      // cap of the node records staged in LDS per block (0 = none)
	int blocksPerCU = -1;   // run below the resident maximum
	int phPark = -1, phShade = -1;   // lane state machine: lanes that leave a node phase before it ends / that wait before a shade phase
	int parkEighths = -1;   // lock-step walk: share of the lanes (in eighths) that leave a node phase before it ends
	int drainMode = -1;     // lane state machine once lanes are DONE: bit 0 scale phPark, bit 1 scale phShade
	int refillBatch = -1;   // lock-step kernels: lanes that wait with a finished unit before the wave refills them together
	int chunkFrames = -1;   // cap of the frames per launch pair (tests: force several launch pairs)
	int faceNormals = -1;   // 0 = recompute the face normal on every hit instead of reading the stored one
	int bvhBuilder = -1;    // pbr_build_bvh: 1 = round 1's radix tree instead of the clustering builder
	int plocRadius = -1;    // pbr_build_bvh: search radius of the clustering builder
	int tuneLog = -1;       // 1 = the schedule tuner logs its launches to stderr
	int dealOrder = -1;     // the queue's dealing order: 0 always spatial, 1 always cost-ordered (once learnt), -1 by the launch's size
};

struct pbr_ctx {
	int device = -1;
	Knobs knobs;
	hipStream_t stream = nullptr;
	hipEvent_t evStart = nullptr, evStop = nullptr, evTraceStart = nullptr, evTraceStop = nullptr;
	std::string error;
	double lastKernelMs = 0.0;
	int numCUs = 0;

	// scene
	bool hasScene = false;
	float4* dNodes = nullptr;
	float4* dTris = nullptr;
	float4* dTriPN = nullptr;      // exact vertices + vertex normals per face (Phong tessellation); null if the normal indices are unusable
	float4* dMats = nullptr;
	float4* dFaceN = nullptr;      // per-face unit normal + material, prepareFaceNormals
	float4* dLights = nullptr;
	uint32_t numHotAvail = 0;      // records at the head of the node stream that are ranked for LDS staging
	int firstRef = 0;              // record of node 1
	// ray-ordered walk (pbr_config.traversal != 0): one stream of records per order, built on first use from a host copy of the flat tree
	std::vector<pbr_bvh_node> hostNodes;
	std::vector<int> hostFace0s, hostLinks;   // per node: first face or -1; second face / miss link (checkScene)
	std::vector<uint32_t> hostRanked;         // the nodes ranked for LDS staging, most visited first
	float4* dNodesWalk = nullptr;
	uint32_t walkBuilt = 0;        // the scheme dNodesWalk holds (0: none)
	uint32_t walkHotAvail = 0;     // records at the head of dNodesWalk that are ranked for LDS staging (all orders interleaved)
	int walkFirst[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	uint64_t nodeBytes = 0, walkBytes = 0, triBytes = 0;   // device bytes of the reference-order stream, the ordered streams, the face records
	uint32_t numNodes = 0, numFaces = 0, numMaterials = 0, numLights = 0;
	uint32_t sceneBrdf = 1;

	// configuration + images
	bool configured = false;
	pbr_config cfg;
	int tilesX = 0, tilesY = 0, numTiles = 0, numLocalTiles = 0;
	float4* dImgIn = nullptr;
	float4* dImgOut = nullptr;
	float4* dImgDbg = nullptr;
	float4* dRows = nullptr;       // W x H row-major staging for read-back / write_input
	float4* dFull = nullptr;       // all tiles of the frame, filled by pbr_import_tiles
	// the banded queue's dealing order (pt_kernel.hpp, nextSlot): the local tiles as a queueRows x queueWidth grid cut into
	// PT_BANDS bands of rows; hTileOrder / dTileOrder name, per band, its tiles in the order they are dealt
	int queueWidth = 1, queueRows = 1;
	unsigned bandFirst[PT_BANDS + 1] = {};       // of hTileOrder / dTileOrder (the spatial table, or a pinned one)
	unsigned costBandFirst[PT_BANDS + 1] = {};   // of hCostOrder / dCostOrder
	std::vector<unsigned> hTileOrder;
	unsigned* dTileOrder = nullptr;
	bool orderPinned = false;      // pbr_diag_set_tile_order: the caller's order stays (tests, A/B runs)
	// cost-ordered dealing for SHORT launches (round 6, learnTileCosts): per band the tiles in kCostClasses classes of
	// falling cost, spatial order inside a class; learnt from the debug image (node visits per pixel of a launch's last frame)
	std::vector<unsigned> hCostOrder;
	unsigned* dCostOrder = nullptr;
	std::vector<unsigned> hLastOrder;   // LONG launches: per band its most expensive quarter last, spatial inside both parts (the spatial bands)
	unsigned* dLastOrder = nullptr;
	float* dTileCost = nullptr;    // node visits per local tile
	std::vector<float> hTileCost;
	bool costLearnt = false;       // dCostOrder holds an order learnt for costCam
	pbr_camera costCam = {};       // the camera (and pixel size) the costs were measured with
	float costPxDim = 0.0f;
	uint32_t launchesSinceLearn = 0;
	char lastDeal[16] = "spatial"; // the order the last render was dealt in
	int lastBvhRadius = 0;         // pbr_build_bvh: the clustering's search radius of the last build (0: none yet / the radix-tree builder)
	bool focusGiven = false;       // pbr_set_focus_depth: the focus pixel's previous-frame distance for the next frame
	float focusDepth = 0.0f;
	float4* dFrameBuf = nullptr;   // frame-parallel launches: {finalColor, focus} per frame and local pixel slot
	size_t frameBufFrames = 0;
	// schedule auto-tuning (launch()): per scene + configuration, the candidates are timed on the first
	// frames that are rendered anyway, then the fastest one is kept
	int tunedPlan = -1;
	Plan plans[7];                                  // the tuner's candidates; valid while plansBuilt (reset by pbr_upload_scene / pbr_configure / pbr_diag_set_knob)
	Plan phongPlan;                                 // the Phong-tessellation build of the refill kernel (takes the place of plans[1])

	bool plansBuilt = false, phongPlanBuilt = false;
	int drainMode = 1;                              // pathTracingPhased, see launch()
	bool workClean = false;                         // the queue heads are zero (foldFrames leaves them so)
	float* hSeeds = nullptr;                        // pinned staging for the seeds of a launch
	size_t hSeedCapacity = 0;
	int pinnedPlan = -1;                            // pbr_diag_pin_plan: >= 0 renders with this plan, no tuning (ranks of a multi-GPU run: all the same)
	uint32_t tuneRenderFrames = 0;                  // the longest render (frames per call) this context has been asked for
	uint32_t tunedAtFrames = 0;                     // the render length tunedPlan was chosen for
	double tuneMs[7] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };      // screening: kTuneFrames frames per plan
	uint32_t tuneFrames[7] = { 0, 0, 0, 0, 0, 0, 0 };
	uint32_t tuneLaunches[7] = { 0, 0, 0, 0, 0, 0, 0 };
	int refineCount = 0;                            // refinement: the plans within 10 % of the fastest (at least two) again, on longer chunks
	int refinePlan[7] = { -1, -1, -1, -1, -1, -1, -1 };
	uint32_t refineChunks = 0;                      // chunks rendered so far in the refinement
	uint32_t refineRounds = 1;                      // 1; 2 once a close call (best two within 5 %) has been given a second palindrome
	double refineFit[7][5] = {};                    // per finalist, over its refinement launches: sums of 1, n, n^2, ms, n * ms (n = frames of the launch)
	char lastPlan[48] = "";        // name of the plan that rendered the largest chunk of the last render
	char lastKernel[96] = "";      // ... and its kernel
	double lastTraceMs = 0.0;      // of the last render: time inside the path-tracing launches only ...
	uint32_t lastTraceLaunches = 0; // ... and how many there were (frame-parallel renders are chunked)
	float* dSeeds = nullptr;
	size_t seedCapacity = 0;
	unsigned long long* dCounters = nullptr;
	unsigned int* dWork = nullptr;
	unsigned int* dGuard = nullptr;
};

namespace {

int fail( pbr_ctx* ctx, int code, const char* fmt, ... ) {
	char buf[512];
	va_list ap;
	va_start( ap, fmt );
	vsnprintf( buf, sizeof( buf ), fmt, ap );
	va_end( ap );

	if( ctx != nullptr ) {
		ctx->error = buf;
	}

	return code;
}

#define HIP_TRY( ctx, call ) \
	do { \
		const hipError_t err__ = ( call ); \
		if( err__ != hipSuccess ) { \
			return fail( ctx, PBR_EDEVICE, "%s: %s", #call, hipGetErrorString( err__ ) ); \
		} \
	} while( 0 )

void freeScene( pbr_ctx* ctx ) {
	(void) hipFree( ctx->dNodes );
	(void) hipFree( ctx->dNodesWalk );
	ctx->dNodesWalk = nullptr;
	ctx->walkBuilt = 0;
	(void) hipFree( ctx->dTris );
	(void) hipFree( ctx->dTriPN );
	ctx->dTriPN = nullptr;
	(void) hipFree( ctx->dFaceN );
	ctx->dFaceN = nullptr;
	(void) hipFree( ctx->dMats );
	(void) hipFree( ctx->dLights );
	ctx->dNodes = ctx->dTris = ctx->dMats = ctx->dLights = nullptr;
	ctx->hasScene = false;
}

// The schedule tuner starts over (a new scene, a new configuration, a changed knob).
void resetTuning( pbr_ctx* ctx ) {
	ctx->tunedPlan = -1;
	ctx->tuneRenderFrames = ctx->tunedAtFrames = 0;
	std::memset( ctx->tuneMs, 0, sizeof( ctx->tuneMs ) );
	std::memset( ctx->tuneFrames, 0, sizeof( ctx->tuneFrames ) );
	std::memset( ctx->tuneLaunches, 0, sizeof( ctx->tuneLaunches ) );
	ctx->refineCount = 0;
	ctx->refineChunks = 0;
	ctx->refineRounds = 1;
	std::memset( ctx->refineFit, 0, sizeof( ctx->refineFit ) );
}

void freeImages( pbr_ctx* ctx ) {
	(void) hipFree( ctx->dImgIn );
	(void) hipFree( ctx->dImgOut );
	(void) hipFree( ctx->dImgDbg );
	(void) hipFree( ctx->dRows );
	(void) hipFree( ctx->dFull );
	(void) hipFree( ctx->dFrameBuf );
	ctx->dFrameBuf = nullptr;
	ctx->frameBufFrames = 0;
	(void) hipFree( ctx->dTileOrder );
	(void) hipFree( ctx->dCostOrder );
	(void) hipFree( ctx->dLastOrder );
	ctx->dLastOrder = nullptr;
	ctx->hLastOrder.clear();
	(void) hipFree( ctx->dTileCost );
	ctx->dTileOrder = ctx->dCostOrder = nullptr;
	ctx->dTileCost = nullptr;
	ctx->hTileOrder.clear();
	ctx->hCostOrder.clear();
	ctx->costLearnt = false;
	ctx->orderPinned = false;
	ctx->dImgIn = ctx->dImgOut = ctx->dImgDbg = ctx->dRows = ctx->dFull = nullptr;
	ctx->configured = false;
}

// Is w an integer-valued float in [lo, hi]?
bool integral( float w, double lo, double hi ) {
	return ( w == std::floor( w ) ) && ( (double) w >= lo ) && ( (double) w <= hi );
}

// queue heads of the pixel-slot queue, one 128-B line each, PT_SUB per band, + the line of the word of heads seen empty (pt_kernel.hpp, nextSlot)
const size_t kWorkBytes = sizeof( unsigned int ) * ( PT_HEADS + 1 ) * PT_BAND_STRIDE;

// frame-parallel launches: cap of the per-frame result buffer (16 B per local pixel and frame)
const size_t kFrameBufBytes = (size_t) 16 << 30;

// 4 public counters (pbr_counters) + 12 slots for experiment statistics (pbr_diag_raw_counters)
const size_t kCounterSlots = 16;


// ---- the path-tracing kernels: one translation unit per plan and build flavour (pt_instances.hpp, pt_instance.hip) ----
// A flavour is a build mode of the same kernel sources (pt_flavour.hpp): bit 0 the ray-ordered walk, bit 1 native
// arithmetic, bit 2 the compact record of the eight-order walk.  A unit that was not linked in (PBR_GUARD builds and the compact
// flavours have no two-paths-per-lane kernels) leaves its picker null.
#define PTI_DECLARE( f, g ) extern "C" const void* PTI_NAME( f, g )( uint32_t, int, int ) __attribute__( ( weak ) );
#define PTI_DECLARE_FLAVOUR( f ) \
	PTI_DECLARE( f, 0 ) PTI_DECLARE( f, 1 ) PTI_DECLARE( f, 2 ) PTI_DECLARE( f, 3 ) PTI_DECLARE( f, 4 ) PTI_DECLARE( f, 5 ) PTI_DECLARE( f, 6 ) PTI_DECLARE( f, 7 )
PTI_DECLARE_FLAVOUR( 0 )
PTI_DECLARE_FLAVOUR( 1 )
PTI_DECLARE_FLAVOUR( 2 )
PTI_DECLARE_FLAVOUR( 3 )
PTI_DECLARE_FLAVOUR( 5 )     // the compact record of the eight-order walk (flavour bit 2; only together with bit 0)
PTI_DECLARE_FLAVOUR( 7 )
#define PTI_ROW( f ) { PTI_NAME( f, 0 ), PTI_NAME( f, 1 ), PTI_NAME( f, 2 ), PTI_NAME( f, 3 ), PTI_NAME( f, 4 ), PTI_NAME( f, 5 ), PTI_NAME( f, 6 ), PTI_NAME( f, 7 ) }
#define PTI_NO_ROW { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr }
const pti_picker kPickers[PTI_FLAVOURS][PTI_GROUPS] = { PTI_ROW( 0 ), PTI_ROW( 1 ), PTI_ROW( 2 ), PTI_ROW( 3 ), PTI_NO_ROW, PTI_ROW( 5 ), PTI_NO_ROW, PTI_ROW( 7 ) };

KernelFn pickKernel( int flavour, int group, uint32_t brdf, bool shadow, bool lights ) {
	const pti_picker pick = kPickers[flavour][group];
	return ( pick != nullptr ) ? (KernelFn) pick( brdf, shadow ? 1 : 0, lights ? 1 : 0 ) : nullptr;
}

// the flavour a configuration renders with
int flavourOf( uint32_t traversal, uint32_t arith ) {
	return ( ( traversal != 0 ) ? 1 : 0 ) | ( ( arith != 0 ) ? 2 : 0 ) | ( ( traversal == PBR_WALK_EIGHT_ORDERS_COMPACT ) ? 4 : 0 );
}

int flavourOf( const pbr_config& cfg ) {
	return flavourOf( cfg.traversal, cfg.arith );
}

// The two-paths-per-lane kernel (pt_dual.hpp) exists where its hand-scheduled two-walk node phase does: not in builds
// without the assembly node phases, not over compact records — those render plan 6 with the 6-waves state machine.
bool dualIsDual( int flavour );

// the "mid" budget: <= 80 VGPRs, launched as two 768-thread blocks per CU = 6 waves / SIMD
const int kMidBlockThreads = 768;

// two paths per lane (pt_dual.hpp).  Its node phase is hand-scheduled only: builds without PT_NODE_PHASE_ASM (PBR_GUARD,
// PBR_NODE_PHASE_CXX) render the plan with the 6-waves state machine — same bits, every loop bounded.
const bool kDualIsDual =
#ifdef PT_NODE_PHASE_ASM
	true;
#else
	false;
#endif

bool dualIsDual( int flavour ) {
	return kDualIsDual && ( flavour & 4 ) == 0;
}

// Scenes whose tree does not fit the staged LDS prefix ("large": the walk is most of a bounce) and those whose tree
// does ("small": shading is): the lock-step walk's park share and refill batch differ between the two.
const uint32_t kWideMinNodes = 2048;

// lock-step kernels: lanes of a wave that wait with a finished unit before they take their next units together.  Measured
// (profiles/r03/experiments/sweep_refill.txt, 1080p): Cornell refill-mid 4791 / 4837 / 4865 / 4765 / 4570 Msamples/s at
// 1 / 8 / 16 / 24 / 32 lanes (a single frame: 2197 -> 2437), Sponza-class refill-wide 1618 / 1664 / 1751 / 1776 / 1791:
// 16 where the tree fits LDS and shading is most of a bounce, 32 where the walk is.
const int kRefillBatchSmall = 16, kRefillBatchLarge = 32;

// PHONGTESS == 1: the lock-step kernel in the LEAN budget (round 4).  The long cubic solve spills in every budget — 336 B /
// 272 B / 72 B of scratch per lane at 64 / 80 / 128 registers — and 4 waves with 8 spilled registers beat 8 waves with 208:
// 4133 against 2731 Msamples/s on a 288-face sphere over a floor at 1080p, 2850 against 2015 on a 9 000-face one
// (profiles/r04/experiments/phong_register_budget.txt; rounds 2-3 built it in the 64-register budget).


// {magic, shifts} with which ptk::divInvariant divides any 32-bit n by d exactly (d = 0 is never divided by: as 1)
void invariantDivisor( unsigned d, unsigned out[2] ) {
	d = ( d == 0u ) ? 1u : d;
	unsigned l = 0;

	while( ( 1ull << l ) < (unsigned long long) d ) {
		l++;
	}

	const unsigned long long magic = ( ( 1ull << 32 ) * ( ( 1ull << l ) - d ) ) / d + 1ull;
	out[0] = (unsigned) magic;
	out[1] = ( l < 1u ? l : 1u ) | ( ( l > 0u ? l - 1u : 0u ) << 8 );
}

// The schedule tuner's chunk lengths are in 1080p-frame equivalents (launch()): how many of this context's frames make one.
// ---- the ray-ordered walk's node streams (pbr_config.traversal, include/pbr_hip.h) --------------------------------
// Not a reference structure: the reference walks its flat tree in one fixed order (a hit continues at index + 1,
// pt_bvh.cl:102,112; the child with the bigger surface area sits there, accelstructures/BVH.cpp:335-343).  The records of
// the node stream name their successors explicitly (pt_kernel.hpp, decodeNode), so another visiting order is another set
// of successor words over the same boxes and leaf words — the kernels do not change, a walk only starts somewhere else.
//
// The tree behind the flat array: a leaf ends at index + 1, a container i at its miss link when that is > i, else where
// its parent ends (the root: N); its children are c0 = i + 1, c1 = end( c0 ), ... below end( i ) — the flattening drops
// nodes (PathTracer.cpp:250-256), so there can be more than two.
// A child's key on an axis: bbMin[axis] + bbMax[axis] (binary32).  A container's children in an order = the DFS child
// list insertion-sorted — a child moves in front of its predecessor while its key is smaller (ascending) / greater
// (descending); as an algorithm, so that ties and NaN keys have one outcome.
//   scheme 1, six orders   order 2 * a + neg sorts EVERY container on axis a, descending when neg; a ray takes the order of
//                          its direction's dominant axis and that component's sign (walkOrderOf, pt_kernel.hpp)
//   scheme 2, eight orders order k = sign bits of the direction; a container sorts on ITS axis — the one its children's keys
//                          spread furthest on (max - min, x before y before z on ties) — descending when that bit of k is set
// Successors in an order: a hit container continues at its first child; child j's next is child j + 1, the last child's
// is its parent's next, the root's is "end"; a missed container and every leaf continue at next.
//
// Layout: [ the ranked nodes, rank by rank, all orders of a rank next to each other (any prefix a block stages in LDS
// serves every order alike) ][ order 0's other nodes in that order's depth-first sequence ][ order 1's ] ...
int buildWalkStreams( pbr_ctx* ctx, uint32_t layout ) {
	// layout 3 = scheme 2's eight orders in the compact record (pt_kernel.hpp, "the compact record"): the successors are the same
	const bool compact = ( layout == PBR_WALK_EIGHT_ORDERS_COMPACT );
	const uint32_t scheme = compact ? 2u : layout;
	const int K = ( scheme == 1 ) ? 6 : 8;
	const uint32_t N = ctx->numNodes;
	const std::vector<pbr_bvh_node>& bvh = ctx->hostNodes;
	const std::vector<int>& face0s = ctx->hostFace0s;

	if( bvh.size() != N || N < 2 ) {
		return fail( ctx, PBR_ESTATE, "ray-ordered walk: no host copy of the scene's tree" );
	}
	if( !compact && (size_t) K * N * 32 >= ( (size_t) 1 << 31 ) ) {
		return fail( ctx, PBR_EINVAL, "ray-ordered walk: %d streams of %u records exceed 2 GiB (record references are 31-bit byte offsets); traversal = PBR_WALK_EIGHT_ORDERS_COMPACT holds 2^24 nodes", K, N );
	}
	if( compact && (size_t) N * 64 >= ( (size_t) 1 << 31 ) ) {
		return fail( ctx, PBR_EINVAL, "ray-ordered walk: %u compact records exceed 2 GiB (record references are 31-bit byte offsets)", N );
	}

	// where every subtree ends
	std::vector<uint32_t> end( N );
	{
		std::vector<uint32_t> open;

		for( uint32_t i = 0; i < N; i++ ) {
			while( !open.empty() && i >= end[open.back()] ) {
				open.pop_back();
			}

			if( face0s[i] >= 0 ) {
				end[i] = i + 1;
			}
			else {
				end[i] = ( ctx->hostLinks[i] > (int) i ) ? (uint32_t) ctx->hostLinks[i] : ( open.empty() ? N : end[open.back()] );
				open.push_back( i );
			}
		}
	}

	auto key = [&]( uint32_t node, int axis ) {
		const pbr_bvh_node& n = bvh[node];
		return ( axis == 0 ) ? n.bbMin.x + n.bbMax.x : ( axis == 1 ) ? n.bbMin.y + n.bbMax.y : n.bbMin.z + n.bbMax.z;
	};

	// per order and node: the node to go to when it is a hit container, and the node to go to otherwise (-1: end)
	std::vector<int> onHit( (size_t) K * N, -1 ), onNext( (size_t) K * N, -1 );
	std::vector<uint32_t> children, inOrder;
	std::vector<unsigned char> axisOf( compact ? N : 0, 0 );   // compact: the axis a container sorts its children on

	for( uint32_t i = 0; i < N; i++ ) {
		if( face0s[i] >= 0 ) {
			continue;
		}

		children.clear();

		for( uint32_t c = i + 1; c < end[i]; c = end[c] ) {
			children.push_back( c );
		}

		int ownAxis = 0;

		if( scheme == 2 ) {
			float widest = -1.0f;

			for( int axis = 0; axis < 3; axis++ ) {
				float lo = INFINITY, hi = -INFINITY;

				for( uint32_t c : children ) {
					const float k = key( c, axis );
					lo = ( k < lo ) ? k : lo;
					hi = ( k > hi ) ? k : hi;
				}

				if( hi - lo > widest ) {
					widest = hi - lo;
					ownAxis = axis;
				}
			}
		}

		if( compact ) {
			axisOf[i] = (unsigned char) ownAxis;

			if( children.empty() ) {
				return fail( ctx, PBR_EINVAL, "ray-ordered walk, compact records: container %u has no child (a record names its two first children)", i );
			}
		}

		for( int k = 0; k < K; k++ ) {
			const int axis = ( scheme == 1 ) ? k / 2 : ownAxis;
			const bool descending = ( scheme == 1 ) ? ( k & 1 ) != 0 : ( ( k >> ownAxis ) & 1 ) != 0;
			inOrder.clear();

			for( uint32_t c : children ) {
				const float mine = key( c, axis );
				size_t at = inOrder.size();
				inOrder.push_back( c );

				while( at > 0 ) {
					const float before = key( inOrder[at - 1], axis );

					if( !( descending ? ( mine > before ) : ( mine < before ) ) ) {
						break;
					}

					inOrder[at] = inOrder[at - 1];
					at--;
				}

				inOrder[at] = c;
			}

			const size_t base = (size_t) k * N;
			const int next = ( i == 0 ) ? -1 : onNext[base + i];   // written when i's parent was handled (parents come first)
			onHit[base + i] = inOrder.empty() ? next : (int) inOrder[0];

			for( size_t j = 0; j < inOrder.size(); j++ ) {
				onNext[base + inOrder[j]] = ( j + 1 < inOrder.size() ) ? (int) inOrder[j + 1] : next;
			}
		}
	}

	if( compact ) {
		// ONE record per node: the ranked nodes first (any prefix a block stages in LDS serves every order), the rest along
		// order 0's depth-first sequence (all-ascending: a hit container's ascending first child is then the adjacent record)
		const uint32_t maxHotRecords = ( 160 * 1024 - 256 ) / 64;
		const uint32_t hot = (uint32_t) std::min<size_t>( ctx->hostRanked.size(), maxHotRecords );
		std::vector<int> recordOf( N, -1 );
		size_t nextRecord = 0;

		for( uint32_t r = 0; r < hot; r++ ) {
			recordOf[ctx->hostRanked[r]] = (int) nextRecord++;
		}

		for( int k = 0; k < K; k++ ) {      // every order must reach every node (a tree that is not properly nested fails here)
			const size_t base = (size_t) k * N;
			size_t seen = 0;

			for( int node = onHit[base]; node > 0; node = ( face0s[node] < 0 ) ? onHit[base + node] : onNext[base + node] ) {
				if( ++seen >= N ) {
					return fail( ctx, PBR_ESTATE, "ray-ordered walk: order %d does not visit every node once", k );
				}
				if( k == 0 && recordOf[(size_t) node] < 0 ) {
					recordOf[(size_t) node] = (int) nextRecord++;
				}
			}

			if( seen != N - 1 ) {
				return fail( ctx, PBR_ESTATE, "ray-ordered walk: order %d reaches %zu of %u nodes", k, seen, N - 1 );
			}
		}

		// 32 bytes of header (the eight first references), then 64-byte records: 4 x float4 each; one record of padding
		const size_t numRecords = nextRecord + 1;
		std::vector<float4> storage( 2 + numRecords * 4, make_float4( 0.0f, 0.0f, 0.0f, 0.0f ) );
		float4* const nodes = storage.data() + 2;
		auto refOf = [&]( int node ) { return ( node > 0 ) ? recordOf[(size_t) node] * 64 : -1; };
		auto asFloat = []( int v ) { return __builtin_bit_cast( float, v ); };

		for( uint32_t i = 1; i < N; i++ ) {
			const pbr_bvh_node& n = bvh[i];
			int h0, h1;

			if( face0s[i] < 0 ) {
				// order 0 sorts every container ascending, order 7 every container descending
				h0 = refOf( onHit[(size_t) 0 * N + i] );
				h1 = refOf( onHit[(size_t) 7 * N + i] ) | ( 4 << axisOf[i] );
			}
			else {
				h0 = (int) ( 0x80000000u | ( ( ctx->hostLinks[i] >= 0 ) ? 0x40000000u : 0u ) | (uint32_t) face0s[i] );
				h1 = 0;
			}

			int next[8];

			for( int k = 0; k < 8; k++ ) {
				next[k] = refOf( onNext[(size_t) k * N + i] );
			}

			float4* rec = nodes + (size_t) recordOf[i] * 4;
			rec[0] = make_float4( n.bbMin.x, n.bbMin.y, n.bbMax.x, n.bbMax.y );
			rec[1] = make_float4( n.bbMin.z, n.bbMax.z, asFloat( h0 ), asFloat( h1 ) );
			rec[2] = make_float4( asFloat( next[0] ), asFloat( next[1] ), asFloat( next[2] ), asFloat( next[3] ) );
			rec[3] = make_float4( asFloat( next[4] ), asFloat( next[5] ), asFloat( next[6] ), asFloat( next[7] ) );
		}

		for( int k = 0; k < 8; k++ ) {
			ctx->walkFirst[k] = refOf( onHit[(size_t) k * N] );
		}

		std::memcpy( storage.data(), ctx->walkFirst, sizeof( ctx->walkFirst ) );
		HIP_TRY( ctx, hipSetDevice( ctx->device ) );
		(void) hipFree( ctx->dNodesWalk );
		ctx->dNodesWalk = nullptr;
		ctx->walkBuilt = 0;
		HIP_TRY( ctx, hipMalloc( (void**) &ctx->dNodesWalk, sizeof( float4 ) * storage.size() ) );
		HIP_TRY( ctx, hipMemcpy( ctx->dNodesWalk, storage.data(), sizeof( float4 ) * storage.size(), hipMemcpyHostToDevice ) );
		ctx->walkHotAvail = hot * 2u;      // in 32-byte slots, the unit of a plan's LDS share
		ctx->walkBytes = sizeof( float4 ) * storage.size();
		ctx->walkBuilt = layout;
		return PBR_OK;
	}

	// records: the ranked nodes interleaved, then every order's remaining nodes along its own depth-first sequence
	const uint32_t maxHot = ( 160 * 1024 - 256 ) / 32;
	const uint32_t hotPerOrder = (uint32_t) std::min<size_t>( ctx->hostRanked.size(), maxHot / (uint32_t) K );
	std::vector<int> recordOf( (size_t) K * N, -1 );
	size_t nextRecord = 0;

	for( uint32_t r = 0; r < hotPerOrder; r++ ) {
		for( int k = 0; k < K; k++ ) {
			recordOf[(size_t) k * N + ctx->hostRanked[r]] = (int) nextRecord++;
		}
	}

	for( int k = 0; k < K; k++ ) {
		const size_t base = (size_t) k * N;
		size_t seen = 0;

		for( int node = onHit[base]; node > 0; node = ( face0s[node] < 0 ) ? onHit[base + node] : onNext[base + node] ) {
			if( ++seen >= N ) {
				return fail( ctx, PBR_ESTATE, "ray-ordered walk: order %d does not visit every node once", k );
			}
			if( recordOf[base + node] < 0 ) {
				recordOf[base + node] = (int) nextRecord++;
			}
		}

		if( seen != N - 1 ) {
			return fail( ctx, PBR_ESTATE, "ray-ordered walk: order %d reaches %zu of %u nodes", k, seen, N - 1 );
		}
	}

	// 32 bytes of header — the eight first references, where firstNode() reads them — then the records; the last one is
	// padding, as in the reference-order stream
	const size_t numRecords = nextRecord + 1;
	std::vector<float4> storage( ( numRecords + 1 ) * 2, make_float4( 0.0f, 0.0f, 0.0f, 0.0f ) );
	float4* const nodes = storage.data() + 2;

	for( int k = 0; k < K; k++ ) {
		const size_t base = (size_t) k * N;
		auto refOf = [&]( int node ) { return ( node > 0 ) ? recordOf[base + (size_t) node] * 32 : -1; };

		for( uint32_t i = 1; i < N; i++ ) {
			const pbr_bvh_node& n = bvh[i];
			int w0, w1;

			if( face0s[i] < 0 ) {
				w0 = refOf( onHit[base + i] );
				w1 = refOf( onNext[base + i] );
			}
			else {
				w0 = (int) ( 0x80000000u | ( ( ctx->hostLinks[i] >= 0 ) ? 0x40000000u : 0u ) | (uint32_t) face0s[i] );
				w1 = refOf( onNext[base + i] );
			}

			const size_t r = (size_t) recordOf[base + i];
			nodes[r * 2 + 0] = make_float4( n.bbMin.x, n.bbMin.y, n.bbMax.x, n.bbMax.y );
			nodes[r * 2 + 1] = make_float4( n.bbMin.z, n.bbMax.z, __builtin_bit_cast( float, w0 ), __builtin_bit_cast( float, w1 ) );
		}

		ctx->walkFirst[k] = refOf( onHit[base] );
	}

	for( int k = K; k < 8; k++ ) {
		ctx->walkFirst[k] = ctx->walkFirst[0];
	}

	std::memcpy( storage.data(), ctx->walkFirst, sizeof( ctx->walkFirst ) );

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	(void) hipFree( ctx->dNodesWalk );
	ctx->dNodesWalk = nullptr;
	ctx->walkBuilt = 0;
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dNodesWalk, sizeof( float4 ) * storage.size() ) );
	HIP_TRY( ctx, hipMemcpy( ctx->dNodesWalk, storage.data(), sizeof( float4 ) * storage.size(), hipMemcpyHostToDevice ) );
	ctx->walkHotAvail = hotPerOrder * (uint32_t) K;
	ctx->walkBytes = sizeof( float4 ) * storage.size();
	ctx->walkBuilt = layout;
	return PBR_OK;
}

// The node stream, its ranked prefix and the first record(s) for the configured traversal.
int applyWalk( pbr_ctx* ctx, DevParams* P, uint32_t* hotAvail ) {
	const uint32_t scheme = ctx->configured ? ctx->cfg.traversal : 0u;
	P->nodes = ctx->dNodes;
	P->firstRef = ctx->firstRef;
	P->walkScheme = 0;
	*hotAvail = ctx->numHotAvail;

	if( scheme == 0 ) {
		return PBR_OK;
	}

	if( ctx->walkBuilt != scheme ) {
		const int built = buildWalkStreams( ctx, scheme );

		if( built != PBR_OK ) {
			return built;
		}
	}

	P->nodes = ctx->dNodesWalk + 2;   // behind the 32-byte header ...
	P->walkTable = (const int*) ctx->dNodesWalk;   // ... which the kernels read through a pointer of its own (DevParams)
	P->firstRef = ctx->walkFirst[0];
	P->walkScheme = (int) scheme;
	*hotAvail = ctx->walkHotAvail;
	return PBR_OK;
}

// What a mode needs beyond valid numbers, checked where the mode is CHOSEN (pbr_configure, and pbr_upload_scene of a
// configured context) instead of in the middle of a viewer's render loop: the mode's kernels are linked in, and —
// with a scene present — its node streams can be built (they are, here: a tree that is not properly nested, or too many nodes
// for 31-bit record references, fails now).  Streams of a traversal that is no longer configured are freed.
int prepareWalk( pbr_ctx* ctx ) {
	if( !ctx->configured ) {
		return PBR_OK;
	}
	if( pbr_mode_built( ctx->cfg.traversal, ctx->cfg.arith ) != 1 ) {
		return fail( ctx, PBR_ESTATE, "this library was built without the kernels of traversal %u / arith %u (pbr_mode_built)", ctx->cfg.traversal, ctx->cfg.arith );
	}
	if( ctx->walkBuilt != ctx->cfg.traversal && ctx->dNodesWalk != nullptr ) {
		HIP_TRY( ctx, hipSetDevice( ctx->device ) );
		(void) hipFree( ctx->dNodesWalk );
		ctx->dNodesWalk = nullptr;
		ctx->walkBuilt = 0;
		ctx->walkBytes = 0;
	}
	if( ctx->hasScene && ctx->cfg.traversal != 0 && ctx->walkBuilt != ctx->cfg.traversal ) {
		return buildWalkStreams( ctx, ctx->cfg.traversal );
	}

	return PBR_OK;
}

// ---- the dealing order of the banded queue (pt_kernel.hpp, nextSlot) -------------------------------------------------
// The local tiles form a queueRows x queueWidth grid (row-major local tile index; its true shape when unsharded, about that
// when sharded).  Band b holds the rows [ b * queueRows / PT_BANDS, ( b + 1 ) * queueRows / PT_BANDS ); its stretch of the order
// table is [ bandFirst[b], bandFirst[b + 1] ) and names exactly the band's tiles (the ragged end of the grid is left out).
//
// The spatial order (rounds 1 - 5's only one): inside a band column by column, so that the tiles the waves of one XCD hold
// at a time form a compact block of the image, not a strip as wide as the frame.
void spatialTileOrder( const pbr_ctx* ctx, std::vector<unsigned>* order, unsigned first[PT_BANDS + 1] ) {
	order->clear();
	order->reserve( (size_t) ctx->numLocalTiles );

	for( int band = 0; band < PT_BANDS; band++ ) {
		const unsigned row0 = ( (unsigned) band * (unsigned) ctx->queueRows ) / PT_BANDS;
		const unsigned rows = ( (unsigned) ( band + 1 ) * (unsigned) ctx->queueRows ) / PT_BANDS - row0;
		first[band] = (unsigned) order->size();

		for( unsigned col = 0; col < (unsigned) ctx->queueWidth; col++ ) {
			for( unsigned row = 0; row < rows; row++ ) {
				const unsigned tile = ( row0 + row ) * (unsigned) ctx->queueWidth + col;

				if( tile < (unsigned) ctx->numLocalTiles ) {
					order->push_back( tile );
				}
			}
		}
	}

	first[PT_BANDS] = (unsigned) order->size();
}

int uploadTileOrder( pbr_ctx* ctx ) {
	HIP_TRY( ctx, hipMemcpyAsync( ctx->dTileOrder, ctx->hTileOrder.data(), sizeof( unsigned ) * ctx->hTileOrder.size(), hipMemcpyHostToDevice, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );   // the source is pageable: do not let it change under the copy
	return PBR_OK;
}

// Cost-ordered dealing.  A launch ends at the pace of its longest paths ("How a launch ends"): when the queue runs
// dry every lane holds a path, and the machine empties while the longest of them finish — 0.5 ms on a Sponza-class scene,
// whatever the launch's length.  Dealt expensive tiles first, the paths that start last are short ones.  Measured (round 6,
// profiles/r06/experiments/deal_order*.txt; round 4 had measured the unsharded half of it, profiles/r04/experiments/
// heaviest_tiles_first.txt): rank 0's share of a 20-frame render split 8 ways -3.5 ... -8 % (Sponza-class 3.14 -> 3.00 ms,
// Dragon-class 3.85 -> 3.72, hairball 6.97 -> 6.72, Cornell 1.50 -> 1.42), its single frame -2 ... -10 %; but a long launch
// LOSES 1 - 8 % because the tiles an XCD holds at one time are no longer neighbours (Sponza-class 64 frames 60.5 -> 61.1 ms,
// hairball 127.9 -> 131.5) and a single full frame neither gains nor loses.  So falling classes only up to kCostOrderTileFrames
// tiles x frames (what is dealt above that: next paragraph).  Eight classes by the band's own cost octiles — the finer the
// classes the less locality is left, a full sort is the worst on long launches and no better on short.
//
// LONG launches (found late in round 6, profiles/r06/experiments/deal_order_ascending*.txt, band_balance*.txt): the same cost
// map, the other way round.  A band that deals its most EXPENSIVE quarter LAST — spatial order inside both parts — renders a long
// launch 1 - 7 % faster than the spatial order: Sponza-class 20 frames 19.37 -> 19.10 ms (two-paths plan), 20.13 -> 19.60 (6 waves);
// Dragon-class 19.85 -> 18.60 ms, 64 frames 61.1 -> 57.0; hairball 127.7 -> 125.2; Cornell 27.2 -> 26.1; the eight-order walk alike
// (Dragon-class 49.96 -> 47.14 ms); rank 0 of 8 from ~40 frames on.  The gain is proportional to the launch's length, the opposite
// direction (expensive first) loses as much, the split point hardly matters (15 / 25 / 40 %), ascending classes do the same and
// interleaving the classes does not: what counts is that a band's heavy tiles come when the XCDs whose own bands are cheap have
// run out of them and join in (bands differ by up to 10 x in cost, and an XCD works on its own band until it is empty) — the
// heavy part of every band is then shared by all eight XCDs, their L2s and their fabric links, instead of being its owner's alone.
// Equalising the bands' costs by moving their row boundaries gives a fifth of that; and the heavy tiles have to be the VERY last a
// band deals: a coda of its cheapest 10 % behind them gives the whole gain back (band_balance_cheap_coda.txt).  That was the clue:
// most of it was the QUEUE HEADS.  The XCDs that have run dry all draw from the one head of the band they help, a head hands out
// ~90 draws / us, and cheap tiles are drawn the fastest — expensive-last merely made sure that the shared part of a launch draws
// slowly.  With four heads per band and the helpers spread over them (pt_kernel.hpp, nextSlot; experiments/queue_heads_*.txt,
// queue_subheads_*.txt) the SPATIAL order is as fast as expensive-last was (Sponza-class 64 frames 2165 -> 2211 Msamples/s against
// 2196; Dragon-class 2132 -> 2317 against 2290; Cornell 4832 -> 5148 against 5053) and expensive-last still adds 1.3 % on the
// Dragon-class scene and Cornell, nothing on the other two.  Below ~192 Ki tiles x frames it loses to the
// spatial order (the long paths start last), hence three orders by the size of the RENDER CALL (all launches of a call alike;
// while the schedule tuner is still measuring, everything is dealt spatially: launch()):
//   tiles x frames <= 128 Ki  eight classes of falling cost     <= 192 Ki  spatial     above  expensive quarter last
// A SHARD (tile_world > 1) deals falling classes up to 1 Mi tiles x frames: its tiles are every N-th of the image, the spatial order
// has little locality to lose there, and with the heads out of the way the shorter end is what is left to gain — rank 0's share at
// 20 frames, N = 2 / 4: Sponza-class 10.24 -> 10.07 ms, 5.54 -> 5.40; Dragon-class 10.67 -> 10.58, 6.23 -> 6.09; N = 8 at 64 frames
// 8.29 -> 8.13, 8.97 -> 8.80 (queue_subheads_orders_by_launch_length.txt).  Small UNSHARDED images do not share that: 1280x720 ...
// 640x360 behave like 1080p (queue_subheads_orders_small_and_4k_images.txt).
const unsigned kCostClasses = 8;
const size_t kCostOrderTileFrames = 128 * 1024;
const size_t kCostOrderShardTileFrames = 1024 * 1024;
const size_t kSpatialOrderTileFrames = 192 * 1024;

// per band: the spatial order, stably partitioned into [ the cheaper three quarters ][ the most expensive quarter ]
void expensiveLastTileOrder( const unsigned bandFirst[PT_BANDS + 1], const std::vector<unsigned>& spatial, const std::vector<float>& cost, std::vector<unsigned>* order ) {
	order->assign( spatial.size(), 0u );
	std::vector<float> sorted;

	for( int band = 0; band < PT_BANDS; band++ ) {
		const unsigned first = bandFirst[band], n = bandFirst[band + 1] - first;

		if( n == 0 ) {
			continue;
		}

		sorted.resize( n );

		for( unsigned k = 0; k < n; k++ ) {
			sorted[k] = cost[spatial[first + k]];
		}

		std::sort( sorted.begin(), sorted.end() );
		const float edge = sorted[std::min<size_t>( n - 1, ( (size_t) 3 * n ) / 4 )];
		unsigned at = first;

		for( int pass = 0; pass < 2; pass++ ) {
			for( unsigned k = 0; k < n; k++ ) {
				const unsigned tile = spatial[first + k];

				if( ( cost[tile] > edge ) == ( pass == 1 ) ) {
					( *order )[at++] = tile;
				}
			}
		}
	}
}

// per band: the spatial order, stably partitioned into kCostClasses classes of falling cost (class edges = the band's octiles)
void costTileOrder( const unsigned bandFirst[PT_BANDS + 1], const std::vector<unsigned>& spatial, const std::vector<float>& cost, std::vector<unsigned>* order ) {
	order->assign( spatial.size(), 0u );
	std::vector<float> sorted;
	std::vector<unsigned> fill( kCostClasses );

	for( int band = 0; band < PT_BANDS; band++ ) {
		const unsigned first = bandFirst[band], n = bandFirst[band + 1] - first;

		if( n == 0 ) {
			continue;
		}

		sorted.resize( n );

		for( unsigned k = 0; k < n; k++ ) {
			sorted[k] = cost[spatial[first + k]];
		}

		std::sort( sorted.begin(), sorted.end() );
		float edge[kCostClasses - 1];

		for( unsigned c = 0; c + 1 < kCostClasses; c++ ) {
			edge[c] = sorted[std::min<size_t>( n - 1, ( (size_t) ( c + 1 ) * n ) / kCostClasses )];
		}

		// class 0 = the most expensive: a tile's class counts the edges its cost stays below
		auto classOf = [&]( float v ) {
			unsigned below = 0;

			for( unsigned c = 0; c + 1 < kCostClasses; c++ ) {
				below += ( v < edge[c] ) ? 1u : 0u;
			}

			return below;
		};

		std::fill( fill.begin(), fill.end(), 0u );

		for( unsigned k = 0; k < n; k++ ) {
			fill[classOf( cost[spatial[first + k]] )]++;
		}

		unsigned at = 0;

		for( unsigned c = 0; c < kCostClasses; c++ ) {
			const unsigned size = fill[c];
			fill[c] = at;
			at += size;
		}

		for( unsigned k = 0; k < n; k++ ) {
			const unsigned tile = spatial[first + k];
			( *order )[first + fill[classOf( cost[tile] )]++] = tile;
		}
	}
}

// After a launch: node visits per local tile from the debug image (the launch's last frame), then the cost order.  Runs when
// there is no order yet or the camera has moved since it was learnt (then at most every 16th launch: an order learnt a few
// frames ago is nearly as good, and the read-back is a synchronisation a frame-by-frame caller should not pay per frame).
int learnTileCosts( pbr_ctx* ctx, const pbr_camera* cam, float pxDim ) {
	if( ctx->orderPinned || ctx->knobs.dealOrder == 0 || ctx->dCostOrder == nullptr ) {
		return PBR_OK;
	}

	const bool moved = !ctx->costLearnt || std::memcmp( &ctx->costCam, cam, sizeof( pbr_camera ) ) != 0 || ctx->costPxDim != pxDim;

	if( !moved || ( ctx->costLearnt && ctx->launchesSinceLearn < 16u ) ) {
		return PBR_OK;
	}

	const unsigned tiles = (unsigned) ctx->numLocalTiles;
	hipLaunchKernelGGL( ptk::tileCosts, dim3( ( tiles + 3u ) / 4u ), dim3( 256 ), 0, ctx->stream, (const float4*) ctx->dImgDbg, ctx->dTileCost, tiles );
	HIP_TRY( ctx, hipGetLastError() );
	ctx->hTileCost.resize( tiles );
	HIP_TRY( ctx, hipMemcpyAsync( ctx->hTileCost.data(), ctx->dTileCost, sizeof( float ) * tiles, hipMemcpyDeviceToHost, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	std::vector<unsigned> spatial;
	spatialTileOrder( ctx, &spatial, ctx->costBandFirst );
	costTileOrder( ctx->costBandFirst, spatial, ctx->hTileCost, &ctx->hCostOrder );
	expensiveLastTileOrder( ctx->costBandFirst, spatial, ctx->hTileCost, &ctx->hLastOrder );
	HIP_TRY( ctx, hipMemcpyAsync( ctx->dCostOrder, ctx->hCostOrder.data(), sizeof( unsigned ) * ctx->hCostOrder.size(), hipMemcpyHostToDevice, ctx->stream ) );
	HIP_TRY( ctx, hipMemcpyAsync( ctx->dLastOrder, ctx->hLastOrder.data(), sizeof( unsigned ) * ctx->hLastOrder.size(), hipMemcpyHostToDevice, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	ctx->costLearnt = true;
	ctx->costCam = *cam;
	ctx->costPxDim = pxDim;
	ctx->launchesSinceLearn = 0;
	return PBR_OK;
}

uint32_t tuneScaleOf( size_t localPixels ) {
	const size_t reference = (size_t) 1920 * 1080;
	const size_t scale = ( reference + localPixels / 2 ) / std::max<size_t>( localPixels, 1 );
	return (uint32_t) std::min<size_t>( std::max<size_t>( scale, 1 ), 64 );
}

int launch( pbr_ctx* ctx, uint32_t firstCount, uint32_t nFrames, const float* seeds,
            bool explicitWeight, float weight, float pxDim, const pbr_camera* cam ) {
	if( !ctx->hasScene || !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "render before pbr_upload_scene / pbr_configure" );
	}
	if( cam == nullptr || seeds == nullptr || nFrames == 0 ) {
		return fail( ctx, PBR_EINVAL, "render: null camera / seeds or zero frames" );
	}
	if( ctx->cfg.brdf != ctx->sceneBrdf ) {
		return fail( ctx, PBR_EINVAL, "configured BRDF %u does not match the uploaded materials (BRDF %u)", ctx->cfg.brdf, ctx->sceneBrdf );
	}

	const bool dof = ( cam->focusPoint[0] >= 0 && cam->focusPoint[1] >= 0 );

	if( dof && nFrames > 1 ) {
		return fail( ctx, PBR_EINVAL, "depth of field reads the previous frame of another pixel: render one frame per call" );
	}
	if( dof && ctx->cfg.tile_world > 1 && !ctx->focusGiven ) {
		return fail( ctx, PBR_EINVAL, "depth of field with tile sharding: the focus pixel's tile may live on another rank — pass its previous-frame distance with pbr_set_focus_depth (owner: pbr_get_focus_depth) before every frame" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );

	// the kernels' error flags are of THIS render (host-mapped memory; no launch of this context is in flight here)
	for( int k = 0; k < 4; k++ ) {
		( (volatile unsigned*) ctx->dGuard )[k] = 0u;
	}

	if( ctx->seedCapacity < nFrames ) {
		(void) hipFree( ctx->dSeeds );
		ctx->dSeeds = nullptr;
		ctx->seedCapacity = 0;
		HIP_TRY( ctx, hipMalloc( (void**) &ctx->dSeeds, sizeof( float ) * nFrames ) );
		ctx->seedCapacity = nFrames;
	}

	// the caller's seeds go through a pinned staging buffer: a copy from pageable memory is staged synchronously by the runtime
	if( ctx->hSeedCapacity < nFrames ) {
		(void) hipHostFree( ctx->hSeeds );
		ctx->hSeeds = nullptr;
		ctx->hSeedCapacity = 0;
		HIP_TRY( ctx, hipHostMalloc( (void**) &ctx->hSeeds, sizeof( float ) * std::max<size_t>( nFrames, 64 ), hipHostMallocDefault ) );
		ctx->hSeedCapacity = std::max<size_t>( nFrames, 64 );
	}

	std::memcpy( ctx->hSeeds, seeds, sizeof( float ) * nFrames );
	HIP_TRY( ctx, hipMemcpyAsync( ctx->dSeeds, ctx->hSeeds, sizeof( float ) * nFrames, hipMemcpyHostToDevice, ctx->stream ) );

	// the queue heads: foldFrames zeroes them behind every path-tracing launch, so only the first launch of a context
	// (and any launch after one that did not end in foldFrames) has to
	if( !ctx->workClean ) {
		HIP_TRY( ctx, hipMemsetAsync( ctx->dWork, 0, kWorkBytes, ctx->stream ) );
	}

	ctx->workClean = false;

	DevParams P;
	std::memset( &P, 0, sizeof( P ) );
	P.parkEighths = 4;
	uint32_t hotAvail = 0;
	{
		const int walk = applyWalk( ctx, &P, &hotAvail );   // P.nodes, P.firstRef / walkFirst for the configured traversal

		if( walk != PBR_OK ) {
			return walk;
		}
	}
	P.tris = ctx->dTris;
	P.triPN = ctx->dTriPN;
	P.faceN = ctx->dFaceN;
	P.phongAlpha = ctx->cfg.phong_tessellation;
	P.mats = ctx->dMats;
	P.lights = ctx->dLights;
	P.imgIn = ctx->dImgIn;
	P.imgOut = ctx->dImgOut;
	P.imgDbg = ctx->dImgDbg;
	P.seeds = ctx->dSeeds;
	P.counters = ctx->dCounters;
	P.workCounter = ctx->dWork;
	P.guard = ctx->dGuard;

	const float* src[4] = { &cam->eye.x, &cam->w.x, &cam->u.x, &cam->v.x };
	float* dst[4] = { P.eye, P.cw, P.cu, P.cv };

	for( int i = 0; i < 4; i++ ) {
		for( int k = 0; k < 3; k++ ) {
			dst[i][k] = src[i][k];
		}
	}

	// launch-invariant sub-expressions of initRay (pt_kernel.hpp), same IEEE single operations as the kernel would do
	// (this file is built with -ffp-contract=off like the kernels)
	for( int k = 0; k < 3; k++ ) {
		const float w = (float) ctx->cfg.width, h = (float) ctx->cfg.height;
		const float cuW = P.cu[k] * w;
		P.camA[k] = P.cu[k] - cuW;
		P.cvH[k] = P.cv[k] * h;
	}

	P.halfPx = pxDim * 0.5f;
	P.aperture = cam->lense[0] / cam->lense[1];
	P.samplesF = (float) ctx->cfg.samples;
	P.focusX = cam->focusPoint[0];
	P.focusY = cam->focusPoint[1];
	P.focusGiven = ( dof && ctx->focusGiven ) ? 1 : 0;
	P.focusDepth = ctx->focusDepth;
	ctx->focusGiven = false;   // one frame's worth: the next frame needs the next distance
	P.lenseFocal = cam->lense[0];
	P.lenseAperture = cam->lense[1];
	P.width = (int) ctx->cfg.width;
	P.height = (int) ctx->cfg.height;
	P.tilesX = ctx->tilesX;
	invariantDivisor( (unsigned) ctx->tilesX, P.tilesXDiv );
	P.numLocalTiles = ctx->numLocalTiles;
	P.tileOrder = ctx->dTileOrder;      // (and its band stretches: per launch, below)

	invariantDivisor( 1u, P.framesDiv );   // nextSlot divides by the frames of the launch (set per chunk below)
	P.tileWorld = (int) ctx->cfg.tile_world;
	P.tileRank = (int) ctx->cfg.tile_rank;
	P.numNodes = (int) ctx->numNodes;
	P.numLights = (int) ctx->numLights;
	P.maxDepth = (int) ctx->cfg.max_depth;
	P.maxAddedDepth = (int) ctx->cfg.max_added_depth;
	P.samples = (int) ctx->cfg.samples;
	P.nFrames = (int) nFrames;
	P.firstCount = (int) firstCount;
	P.useExplicitWeight = explicitWeight ? 1 : 0;
	P.explicitWeight = weight;
	P.pxDim = pxDim;
	P.antiAliasing = ctx->cfg.anti_aliasing;
	P.sky[0] = ctx->cfg.sky_light[0];
	P.sky[1] = ctx->cfg.sky_light[1];
	P.sky[2] = ctx->cfg.sky_light[2];

	const bool lights = ( ctx->numLights > 0 );
	const bool shadow = ( ctx->cfg.shadow_rays == 1 ) && lights;
	const Knobs& knobs = ctx->knobs;
	const bool phong = ( ctx->cfg.phong_tessellation > 0.0f );
	const int flavour = flavourOf( ctx->cfg );

	if( phong && ctx->dTriPN == nullptr ) {
		return fail( ctx, PBR_EINVAL, "Phong tessellation needs a scene with usable vertex normals" );
	}

	// A plan = kernel + persistent grid + LDS split.  Grid: as many blocks as stay resident, never more
	// than there is work for.  LDS: each block stages a prefix of the node stream; the CU's 160 KB are
	// split between the blocks the register budget admits.  The experiment knobs (PBR_BLOCKS_PER_CU, PBR_LDS_SLOTS,
	// PBR_PH_PARK, PBR_PH_SHADE, PBR_PARK_EIGHTHS, PBR_DRAIN_MODE) are read when the plans are built — once per
	// scene + configuration — not per launch.
	auto makePlan = [&]( int group, const char* name, int park, int shade, Plan* plan, int blockThreads = PBR_BLOCK, size_t pathSlotBytes = 0 ) -> int {
		const KernelFn kernel = pickKernel( flavour, group, ctx->cfg.brdf, shadow, lights );

		if( kernel == nullptr ) {
			return fail( ctx, PBR_ESTATE, "plan %s: this library was built without that kernel (flavour %d)", name, flavour );
		}

		int blocksPerCU = 0;
		HIP_TRY( ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor( &blocksPerCU, (const void*) kernel, blockThreads, 0 ) );
		blocksPerCU = ( blocksPerCU < 1 ) ? 1 : blocksPerCU;

		if( knobs.blocksPerCU >= 1 && knobs.blocksPerCU < blocksPerCU ) {
			blocksPerCU = knobs.blocksPerCU;
		}

		// a block's share of the CU's 160 KB, for the staged tree top
		const size_t ldsPerCU = 160 * 1024;
		const size_t share = ldsPerCU / (size_t) blocksPerCU - 256;
		const size_t slotBytes = pathSlotBytes;     // two paths per lane: 2 slots x 4 planes x 16 B of path state per lane behind the staged prefix

		// (a toolchain that fitted such a kernel into fewer registers would report two blocks per CU, whose share
		// is smaller than the path slots — the subtraction below is unsigned)
		if( share < slotBytes + 32 ) {
			return fail( ctx, PBR_ESTATE, "plan %s: %zu B of per-lane path state do not fit a block's %zu B share of LDS at %d blocks per CU", name, slotBytes, share, blocksPerCU );
		}

		size_t slots = ( share - slotBytes ) / 32;
		slots = std::min<size_t>( slots, hotAvail );

		if( knobs.ldsSlots >= 0 ) {
			slots = std::min<size_t>( slots, (size_t) knobs.ldsSlots );
		}
		if( flavour & 4 ) {
			slots &= ~(size_t) 1;      // compact records are two slots each: a staged prefix ends on a record
		}

		// the limit is a property of the kernel function, shared by every context of the process: always the block's whole
		// share, so that a context with a small scene never lowers it under another context's cached plan
		HIP_TRY( ctx, hipFuncSetAttribute( (const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) share ) );

		plan->kernel = kernel;
		plan->blockThreads = blockThreads;
		plan->blocks = ctx->numCUs * blocksPerCU;   // all that are resident at once; run() launches fewer when there is less work
		plan->numHot = (int) slots;
		plan->ldsBytes = slots * 32 + slotBytes;
		plan->park = park;
		plan->shade = shade;
		plan->parkEighths = ( ctx->numNodes >= kWideMinNodes ) ? 4 : 6;   // see traverse(), pt_kernel.hpp

		if( knobs.phPark >= 0 ) {
			plan->park = std::max( 1, std::min( ( pathSlotBytes != 0 ) ? 128 : 64, knobs.phPark ) );   // two paths per lane: walks of up to 128 per wave
		}
		if( knobs.phShade >= 0 ) {
			plan->shade = std::max( 1, std::min( 64, knobs.phShade ) );
		}
		if( knobs.parkEighths >= 0 ) {
			plan->parkEighths = std::min( 8, knobs.parkEighths );
		}

		plan->name = name;
		// the symbol as rocprofv3 prints it (pt_instance.hip instantiates exactly these; namespaces: pt_flavour.hpp)
		const char* const tf[2] = { "false", "true" };
		const int minw[PTI_GROUPS] = { 4, 6, 8, 4, 4, 6, 8, 0 };

		if( group == PTI_DUAL ) {
			std::snprintf( plan->kernelName, sizeof( plan->kernelName ), "ptk_f%d::pathTracingDual<%u, %s, %s>", flavour, ctx->cfg.brdf, tf[shadow], tf[lights] );
		}
		else if( group >= PTI_PHASED_LEAN ) {
			std::snprintf( plan->kernelName, sizeof( plan->kernelName ), "ptk_f%d::pathTracingPhased<%u, %s, %s, %d>", flavour, ctx->cfg.brdf, tf[shadow], tf[lights], minw[group] );
		}
		else {
			std::snprintf( plan->kernelName, sizeof( plan->kernelName ), "ptk_f%d::pathTracing<%u, %s, %s, %d, %s>", flavour, ctx->cfg.brdf, tf[shadow], tf[lights], minw[group], tf[group == PTI_REFILL_PHONG] );
		}

		return PBR_OK;
	};

	// waveUnits: how many wave-sized pieces of work the launch has — tiles x frames (a rank of an 8-GPU run has 4050
	// tiles at 1080p: fewer than the 6144 - 8192 resident waves, but 256 frames of them)
	auto run = [&]( const Plan& plan, size_t waveUnits ) -> int {
		const size_t wavesPerBlock = (size_t) plan.blockThreads / 64;
		const size_t needed = std::max<size_t>( 1, ( waveUnits + wavesPerBlock - 1 ) / wavesPerBlock );
		const unsigned blocks = (unsigned) std::min<size_t>( (size_t) plan.blocks, needed );
		P.numHot = plan.numHot;
		P.numHotBytes = plan.numHot * 32;
		P.slotBase = plan.numHot * 32;

		P.phPark = plan.park;
		P.phShade = plan.shade;
		P.refillBatch = ( knobs.refillBatch >= 1 ) ? std::min( 64, knobs.refillBatch ) : ( ( ctx->numNodes >= kWideMinNodes ) ? kRefillBatchLarge : kRefillBatchSmall );
		P.drainMode = ctx->drainMode;   // 1: measured (single-frame 1080p launches): park share scaled, shade threshold as is: Dragon-class 2.99 -> 2.47 ms, hairball 4.26 -> 4.02 ms, Sponza- / Cornell-class unchanged; scaling the shade threshold too helps the first two further (2.26 / 3.56 ms) and costs the others 10 - 20 %
		P.parkEighths = plan.parkEighths;
		hipLaunchKernelGGL( plan.kernel, dim3( blocks ), dim3( (unsigned) plan.blockThreads ), plan.ldsBytes, ctx->stream, P );
		HIP_TRY( ctx, hipGetLastError() );

		return PBR_OK;
	};


	// Frame-parallel schedules: every (pixel, frame) is its own unit of work — the frames of a pixel are
	// independent up to the running mean (pathtracing.cl:28,255,332) — so a launch ends with single
	// frames in flight, not with whole pixels; {finalColor, focus} of each go to dFrameBuf and
	// foldFrames applies setColors in frame order.  The buffer (16 B per pixel and frame) is capped,
	// longer renders run as several launch pairs.
	//
	// Which kernel: the lock-step walk ("refill") or the lane state machine ("phased"), each with the lean
	// (4 waves / SIMD, no spills), the mid (6) or the wide (8 waves / SIMD) register budget.  Which one wins
	// depends on the scene (1080p: Cornell refill-mid 4250 vs phased-mid 3650 Msamples/s, dragon-class
	// phased-mid 1900 vs refill-wide 1000), and all of them give the same bits — so the first frames
	// of a scene + configuration, which have to be rendered anyway, are rendered in turn by each
	// candidate (kTuneFrames each) and timed; short launches favour the plans with fewer, larger blocks,
	// so the two or three fastest (the third only if within 10 % of the first) are timed again on short and long chunks,
	// in the palindromic order A B C (short) C B A (long) A B C (long) C B A (short): two lengths separate a launch's
	// fixed cost from its per-frame cost, and every plan's short launches and its long launches are centred on the same
	// moment, so the drift of the clocks — the GPU ramps up from idle during exactly these launches, which biased a
	// one-sided order by 5 % in the per-frame cost — cancels in both.  When the two best end within 5 % of each other the palindrome
	// is run a second time before the decision (the fits accumulate): a fit over four launches carries 1 - 5 % of noise.
	const int kPlans = 7;
	// Lengths in 1080p-frame equivalents: a rank of an 8-GPU run (or a small image) has 1/8 of the pixels per frame, and
	// launches of a few hundred microseconds say little about a long render (measured at 1/8 of the tiles: the tuner
	// kept a plan 26 % slower than the best).  So the chunk lengths grow as the frame shrinks.
	const uint32_t tuneScale = tuneScaleOf( (size_t) ctx->numLocalTiles * 64 );
	const uint32_t kTuneFrames = 2 * tuneScale;     // screening, per plan (or two launches, whichever comes first)
	const uint32_t kRefineShort = 4 * tuneScale, kRefineLong = 12 * tuneScale;     // refinement: two chunks of each length per plan
	const uint32_t kRefinePasses = 4;
	if( !ctx->plansBuilt ) {
		Plan* plans = ctx->plans;
		int status = makePlan( PTI_REFILL_LEAN, "refill-lean", 0, 0, &plans[0] );
		status = ( status != PBR_OK ) ? status : makePlan( PTI_REFILL_WIDE, "refill-wide", 0, 0, &plans[1] );
		status = ( status != PBR_OK ) ? status : makePlan( PTI_PHASED_LEAN, "phased-lean", 16, 32, &plans[2] );
		status = ( status != PBR_OK ) ? status : makePlan( PTI_PHASED_WIDE, "phased-wide", 16, 48, &plans[3] );
		// The state machine's thresholds (lanes that leave a node phase before it ends / lanes that wait before a shade phase):
		// 16 / 40 in the reference's walk (profiles/r03/experiments/sweep_thresholds.txt).  A ray-ordered walk makes fewer visits per
		// shading, and its 6-waves build does best at 12 / 48: Dragon-class +1.5 % (exact) / +2.5 % (native), hairball +0.1 … +0.3 %,
		// Sponza-class -0.3 … -1.2 % — where the two-paths plan wins anyway (profiles/r05/experiments/sweep_thresholds_ordered_walk*.txt).
		const bool orderedWalk = ( flavour & 1 ) != 0;
		status = ( status != PBR_OK ) ? status : makePlan( PTI_PHASED_MID, "phased-mid", orderedWalk ? 12 : 16, orderedWalk ? 48 : 40, &plans[4], kMidBlockThreads );
		status = ( status != PBR_OK ) ? status : makePlan( PTI_REFILL_MID, "refill-mid", 0, 0, &plans[5], kMidBlockThreads );
		// 28 of a wave's up to 128 walks leave a node phase before it ends; a shade phase waits for 48 lanes (measured:
		// profiles/r04/experiments/two_paths_per_lane.txt).  Without the hand-scheduled node phase: phased-mid's kernel and thresholds.
		status = ( status != PBR_OK ) ? status : ( dualIsDual( flavour )
			? makePlan( PTI_DUAL, "phased-dual", 28, 48, &plans[6], PBR_BLOCK, (size_t) 2 * 4 * 16 * PBR_BLOCK )
			: makePlan( PTI_PHASED_MID, "phased-dual", 16, 40, &plans[6], kMidBlockThreads ) );

		if( status != PBR_OK ) {
			return status;
		}


		ctx->drainMode = 1;

		if( knobs.drainMode >= 0 ) {
			ctx->drainMode = knobs.drainMode;
		}

		ctx->plansBuilt = true;
		ctx->phongPlanBuilt = false;
	}

	Plan plans[kPlans];

	for( int k = 0; k < kPlans; k++ ) {
		plans[k] = ctx->plans[k];
	}

	auto screened = [&]( int plan ) { return ctx->tuneFrames[plan] >= kTuneFrames || ctx->tuneLaunches[plan] >= 2u; };

	int forcedPlan = -1;

	if( ctx->pinnedPlan >= 0 ) {
		forcedPlan = std::min( kPlans - 1, ctx->pinnedPlan );
	}

	if( phong ) {
		// one plan: the Phong-tessellation build of the lock-step kernel (lean budget), in the slot of plan 1
		if( !ctx->phongPlanBuilt ) {
			const int made = makePlan( PTI_REFILL_PHONG, "refill-lean-phong", 0, 0, &ctx->phongPlan );

			if( made != PBR_OK ) {
				return made;
			}

			ctx->phongPlanBuilt = true;
		}

		plans[1] = ctx->phongPlan;
		forcedPlan = 1;
	}

	const size_t pixelSlots = (size_t) ctx->numLocalTiles * 64;
	const size_t frameBytes = sizeof( float4 ) * pixelSlots;
	size_t chunkCap = std::max<size_t>( 1, kFrameBufBytes / frameBytes );
	chunkCap = std::min<size_t>( chunkCap, nFrames );
	// the queue heads count pixel slots x frames of a band in 32 bits
	chunkCap = std::min<size_t>( chunkCap, std::max<size_t>( 1, 0x7FFFFFFFull / ( pixelSlots + 64 * (size_t) ctx->queueWidth ) ) );

	if( knobs.chunkFrames >= 1 ) {    // tests: force several launch pairs
		chunkCap = std::min<size_t>( chunkCap, (size_t) knobs.chunkFrames );
	}

	if( ctx->frameBufFrames < chunkCap ) {
		// grown geometrically, never beyond the cap: at least 256 frames, then twice what is asked for.  A longer render
		// after a shorter one must not pay for a multi-GB reallocation inside every call (measured: 432 -> 512 frames at
		// 1080p, hipFree + hipMalloc of 16 GiB = 0.4 - 0.7 s when the pages have had other owners), and a small image must
		// not take 16 GiB because one render was longer than 256 frames (64 x 64 pixels, 300 frames: 39 MB).
		const size_t capFrames = std::max<size_t>( 1, kFrameBufBytes / frameBytes );
		const size_t frames = std::min<size_t>( capFrames, std::max<size_t>( 256, ( chunkCap <= 256 ) ? chunkCap : 2 * chunkCap ) );
		(void) hipFree( ctx->dFrameBuf );
		ctx->dFrameBuf = nullptr;
		ctx->frameBufFrames = 0;
		HIP_TRY( ctx, hipMalloc( (void**) &ctx->dFrameBuf, frameBytes * frames ) );
		ctx->frameBufFrames = frames;
	}

	ctx->tuneRenderFrames = std::max( ctx->tuneRenderFrames, nFrames );

	// A launch costs a + b x frames (a: ramp-up and drain, 0.3 - 0.6 ms; b: the per-frame rate) and the plans differ in
	// both: least squares over each finalist's refinement launches, then the cost of a render of `frames` frames.
	// Launches of one length only (a caller rendering frame by frame) cannot separate the two: a = 0, `separable` false.
	bool separable = true;
	double runnerUp = 0.0;       // cost of the second-best finalist relative to the best's, as decide() last saw them
	auto decide = [&]( uint32_t renderFrames ) -> int {
		const double frames = (double) std::max<uint32_t>( renderFrames, 1u );
		int best = -1;
		double bestCost = 0.0, secondCost = 0.0;
		separable = true;

		for( int k = 0; k < ctx->refineCount; k++ ) {
			const double* f = ctx->refineFit[k];
			const double det = f[0] * f[2] - f[1] * f[1];
			double a = 0.0, b = f[3] / f[1];

			if( det > 1e-9 * f[2] * f[0] ) {
				const double bFit = ( f[0] * f[4] - f[1] * f[3] ) / det;
				const double aFit = ( f[3] - bFit * f[1] ) / f[0];

				if( aFit >= 0.0 && bFit >= 0.0 ) {
					a = aFit;
					b = bFit;
				}
			}
			else {
				separable = false;
			}

			const double cost = ( a + b * frames ) / frames;

			if( knobs.tuneLog > 0 ) {
				std::fprintf( stderr, "[pbr tune] fit %-12s a %.3f ms  b %.3f ms/frame  -> %.4f ms/frame at %u frames\n", plans[ctx->refinePlan[k]].name, a, b, cost, (unsigned) frames );
			}

			if( best < 0 || cost < bestCost ) {
				secondCost = ( best < 0 ) ? 0.0 : bestCost;
				best = k;
				bestCost = cost;
			}
			else if( secondCost == 0.0 || cost < secondCost ) {
				secondCost = cost;
			}
		}

		runnerUp = ( bestCost > 0.0 && secondCost > 0.0 ) ? secondCost / bestCost : 0.0;
		return ctx->refinePlan[best];
	};

	if( ctx->tunedPlan >= 0 && ctx->refineCount > 0 && nFrames > 2u * ctx->tunedAtFrames ) {
		// tuned for shorter renders than this one (a viewer's frame-by-frame calls, then a batch): the fixed cost
		// weighs less now.  With fits from two launch lengths that is a new evaluation; with one length only, the
		// finalists are timed again on this render's frames.
		const int again = decide( nFrames );

		if( separable ) {
			ctx->tunedPlan = again;
			ctx->tunedAtFrames = nFrames;
		}
		else if( nFrames >= 4u * kRefineLong ) {
			ctx->tunedPlan = -1;
			ctx->refineChunks = 0;
			ctx->refineRounds = 1;
			std::memset( ctx->refineFit, 0, sizeof( ctx->refineFit ) );
		}
	}
	P.frameBuf = ctx->dFrameBuf;
	P.frameStride = (unsigned) pixelSlots;
	const unsigned foldBlocks = (unsigned) ( ( pixelSlots + 255 ) / 256 );
	double traceMs = 0.0;
	uint32_t launches = 0, largest = 0;

	int dealt = 0;

	// While the schedule tuner is still measuring, everything is dealt spatially: its chunks are short launches whatever the call's
	// length, the cost orders are made for one length each (expensive-last costs a 2-frame launch 6 %), and a plan's fitted fixed cost
	// must not depend on which of them its chunks happened to run in (seen: the 6-waves plan kept over the two-paths one, -3.6 %).
	const bool stillTuning = ( forcedPlan < 0 && ctx->tunedPlan < 0 );

	if( !ctx->orderPinned && ctx->costLearnt && knobs.dealOrder != 0 && ( !stillTuning || knobs.dealOrder > 0 ) ) {
		const size_t tileFrames = (size_t) ctx->numLocalTiles * nFrames;
		const size_t costLimit = ( ctx->cfg.tile_world > 1u ) ? kCostOrderShardTileFrames : kCostOrderTileFrames;
		dealt = ( knobs.dealOrder > 0 ) ? std::min( knobs.dealOrder, 2 ) : ( tileFrames <= costLimit ) ? 1 : ( tileFrames <= kSpatialOrderTileFrames ) ? 0 : 2;
	}

	std::snprintf( ctx->lastDeal, sizeof( ctx->lastDeal ), "%s", ctx->orderPinned ? "pinned" : ( dealt == 1 ) ? "cost-classes" : ( dealt == 2 ) ? "expensive-last" : "spatial" );
	HIP_TRY( ctx, hipEventRecord( ctx->evStart, ctx->stream ) );

	for( uint32_t done = 0; done < nFrames; ) {
		// which plan renders this chunk, and how many frames of it
		int choice = forcedPlan;
		bool tuning = false;

		int refining = -1;   // index into refinePlan while the finalists are compared

		if( choice < 0 ) {
			choice = ctx->tunedPlan;

			if( choice < 0 && ctx->refineCount > 0 ) {
				// forward, then backward (A B C C B A): symmetric against a clock that is still ramping up or throttling
				const uint32_t k = ctx->refineChunks % (uint32_t) ctx->refineCount;
				const bool backward = ( ( ctx->refineChunks / (uint32_t) ctx->refineCount ) & 1u ) != 0u;
				refining = backward ? ctx->refineCount - 1 - (int) k : (int) k;
				choice = ctx->refinePlan[refining];
			}
			else if( choice < 0 ) {
				tuning = true;
				choice = 0;

				while( choice < kPlans - 1 && screened( choice ) ) {
					choice++;
				}
			}
		}

		const Plan& plan = plans[choice];
		uint32_t n = std::min<uint32_t>( (uint32_t) chunkCap, nFrames - done );

		if( tuning ) {
			n = std::min<uint32_t>( n, kTuneFrames - ctx->tuneFrames[choice] );
		}
		if( refining >= 0 ) {
			const uint32_t pass = ( ctx->refineChunks / (uint32_t) ctx->refineCount ) % kRefinePasses;
			const bool longPass = ( pass == 1u || pass == 2u );
			n = std::min<uint32_t>( n, longPass ? kRefineLong : kRefineShort );
		}

		P.nFrames = (int) n;
		invariantDivisor( n, P.framesDiv );
		P.firstCount = (int) ( firstCount + done );
		P.seeds = ctx->dSeeds + done;
		// the dealing order, by the launch's size (costTileOrder): short launches end sooner with the expensive tiles first,
		// long ones are faster in the spatial order
		// the dealing order, by the size of the render call (costTileOrder / expensiveLastTileOrder): 0 spatial, 1 cost classes, 2 expensive last
		P.tileOrder = ( dealt == 1 ) ? ctx->dCostOrder : ( dealt == 2 ) ? ctx->dLastOrder : ctx->dTileOrder;

		for( int band = 0; band < PT_BANDS; band++ ) {
			const unsigned* first = ( dealt != 0 ) ? ctx->costBandFirst : ctx->bandFirst;
			P.bandFirst[band] = first[band];
			P.bandTiles[band] = first[band + 1] - first[band];
		}


		HIP_TRY( ctx, hipEventRecord( ctx->evTraceStart, ctx->stream ) );

		const int ran = run( plan, (size_t) ctx->numLocalTiles * n );

		if( ran != PBR_OK ) {
			return ran;
		}

		HIP_TRY( ctx, hipEventRecord( ctx->evTraceStop, ctx->stream ) );
		// the running mean so far: the input image for the first chunk, imageOut after that
		hipLaunchKernelGGL( ptk::foldFrames, dim3( foldBlocks ), dim3( 256 ), 0, ctx->stream, P,
			(const float4*) ( done == 0 ? ctx->dImgIn : ctx->dImgOut ), ctx->dImgOut );
		HIP_TRY( ctx, hipGetLastError() );
		HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );   // one chunk for all but very long renders

		float ms = 0.0f;
		HIP_TRY( ctx, hipEventElapsedTime( &ms, ctx->evTraceStart, ctx->evTraceStop ) );
		traceMs += (double) ms;
		launches++;

		if( n > largest ) {
			largest = n;
			std::snprintf( ctx->lastPlan, sizeof( ctx->lastPlan ), "%s", plan.name );
			std::snprintf( ctx->lastKernel, sizeof( ctx->lastKernel ), "%s", plan.kernelName );
		}

		if( ( tuning || refining >= 0 ) && knobs.tuneLog > 0 ) {
			std::fprintf( stderr, "[pbr tune] %s %-12s %u frame(s) %.3f ms = %.3f ms/frame\n", tuning ? "screen" : "refine", plan.name, n, (double) ms, (double) ms / n );
		}

		if( tuning ) {
			ctx->tuneMs[choice] += (double) ms;
			ctx->tuneFrames[choice] += n;
			ctx->tuneLaunches[choice]++;

			if( screened( kPlans - 1 ) ) {
				// screening done: every plan within 10 % of the fastest — at least the two fastest — goes on to the refinement
				auto perFrame = [&]( int k ) { return ctx->tuneMs[k] / ctx->tuneFrames[k]; };
				int order[kPlans] = { 0, 1, 2, 3, 4, 5, 6 };
				std::sort( order, order + kPlans, [&]( int x, int y ) { return perFrame( x ) < perFrame( y ); } );
				ctx->refineCount = 0;

				for( int k = 0; k < kPlans; k++ ) {
					if( k < 2 || ( k < 3 && perFrame( order[k] ) <= 1.10 * perFrame( order[0] ) ) ) {
						ctx->refinePlan[ctx->refineCount++] = order[k];
					}
				}
			}
		}

		if( refining >= 0 ) {
			double* fit = ctx->refineFit[refining];
			fit[0] += 1.0;
			fit[1] += (double) n;
			fit[2] += (double) n * (double) n;
			fit[3] += (double) ms;
			fit[4] += (double) n * (double) ms;
			ctx->refineChunks++;

			if( ctx->refineChunks >= ctx->refineRounds * kRefinePasses * (uint32_t) ctx->refineCount ) {
				const int best = decide( ctx->tuneRenderFrames );

				// A close call — the runner-up within 5 % (phased-mid and phased-dual on a Sponza-class scene are 3 % apart, and a
				// fit over four launches per plan carries 1 - 5 % of noise in its per-frame cost: measured, one wrong pick in a dozen
				// runs) — gets a second palindrome of launches before the decision; the fits accumulate.
				if( runnerUp > 0.0 && runnerUp < 1.05 && ctx->refineRounds < 2u ) {
					ctx->refineRounds = 2u;
				}
				else {
					ctx->tunedPlan = best;
					ctx->tunedAtFrames = ctx->tuneRenderFrames;
				}
			}
		}

		done += n;
	}

	HIP_TRY( ctx, hipEventRecord( ctx->evStop, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	ctx->workClean = true;   // the last foldFrames left the queue heads at zero

	float ms = 0.0f;
	HIP_TRY( ctx, hipEventElapsedTime( &ms, ctx->evStart, ctx->evStop ) );
	ctx->lastKernelMs = (double) ms;
	ctx->lastTraceMs = traceMs;
	ctx->lastTraceLaunches = launches;
	ctx->launchesSinceLearn += launches;

	// [1] the path loop (or a record slot of the asynchronous node phase that never filled), [2] a traversal that took more
	// steps than the tree has nodes (PBR_GUARD builds): either way a walk was cut short and the image is wrong
	if( ( (const volatile unsigned*) ctx->dGuard )[1] != 0u || ( (const volatile unsigned*) ctx->dGuard )[2] != 0u ) {
		return fail( ctx, PBR_EDEVICE, "a bounded device loop of the path-tracing kernel gave up (path loop %u, traversal %u trips; pbr_diag_guard_trips): the image is incomplete",
		             ( (const volatile unsigned*) ctx->dGuard )[1], ( (const volatile unsigned*) ctx->dGuard )[2] );
	}

	if( ( (const volatile unsigned*) ctx->dGuard )[3] != 0u ) {
		return fail( ctx, PBR_EDEVICE, "the staged node prefix does not start at LDS address 0 (pt_kernel.hpp, stageHotNodes): this build of the kernels cannot be trusted" );
	}

	return learnTileCosts( ctx, cam, pxDim );
}

int readTiled( pbr_ctx* ctx, const float4* tiles, float* rgba, int tileWorld, int tileRank ) {
	if( !ctx->configured || tiles == nullptr ) {
		return fail( ctx, PBR_ESTATE, "read before pbr_configure (or before pbr_import_tiles)" );
	}
	if( rgba == nullptr ) {
		return fail( ctx, PBR_EINVAL, "read: null destination" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	const dim3 block( 64, 4 );
	const dim3 grid( ( ctx->cfg.width + 63 ) / 64, ( ctx->cfg.height + 3 ) / 4 );
	hipLaunchKernelGGL( ptk::untile, grid, block, 0, ctx->stream, tiles, ctx->dRows,
		(int) ctx->cfg.width, (int) ctx->cfg.height, ctx->tilesX, tileWorld, tileRank );
	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipMemcpyAsync( rgba, ctx->dRows, sizeof( float4 ) * ctx->cfg.width * ctx->cfg.height, hipMemcpyDeviceToHost, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	return PBR_OK;
}

}  // namespace


extern "C" {

uint32_t pbr_abi_version( void ) {
	return PBR_ABI_VERSION;
}

int pbr_mode_built( uint32_t traversal, uint32_t arith ) {
	if( traversal > 3 || arith > 1 ) {
		return -1;
	}

	const int flavour = flavourOf( traversal, arith );

	for( int group = 0; group < PTI_GROUPS; group++ ) {
		// builds without the hand-scheduled node phase, and the compact record's flavours, have no two-paths kernels
		if( group == PTI_DUAL && !dualIsDual( flavour ) ) {
			continue;
		}
		if( kPickers[flavour][group] == nullptr ) {
			return 0;
		}
	}

	return 1;
}

int pbr_create( int device, pbr_ctx** out ) {
	if( out == nullptr ) {
		return PBR_EINVAL;
	}

	pbr_ctx* ctx = new pbr_ctx();
	*out = ctx;
	ctx->device = device;

	int count = 0;
	hipError_t err = hipGetDeviceCount( &count );

	if( err != hipSuccess || count <= 0 ) {
		return fail( ctx, PBR_EDEVICE, "no HIP device available (%s); this library has no CPU path", hipGetErrorString( err ) );
	}
	if( device < 0 || device >= count ) {
		return fail( ctx, PBR_EINVAL, "device %d out of range (%d devices)", device, count );
	}

	HIP_TRY( ctx, hipSetDevice( device ) );
	hipDeviceProp_t prop;
	HIP_TRY( ctx, hipGetDeviceProperties( &prop, device ) );
	ctx->numCUs = prop.multiProcessorCount;
	HIP_TRY( ctx, hipStreamCreateWithFlags( &ctx->stream, hipStreamNonBlocking ) );
	HIP_TRY( ctx, hipEventCreate( &ctx->evStart ) );
	HIP_TRY( ctx, hipEventCreate( &ctx->evTraceStart ) );
	HIP_TRY( ctx, hipEventCreate( &ctx->evTraceStop ) );
	HIP_TRY( ctx, hipEventCreate( &ctx->evStop ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dCounters, sizeof( unsigned long long ) * kCounterSlots ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dWork, kWorkBytes ) );
	// host-visible so that it can be read while a kernel is still running
	HIP_TRY( ctx, hipHostMalloc( (void**) &ctx->dGuard, sizeof( unsigned int ) * 4, hipHostMallocMapped ) );
	std::memset( ctx->dGuard, 0, sizeof( unsigned int ) * 4 );
	HIP_TRY( ctx, hipMemset( ctx->dCounters, 0, sizeof( unsigned long long ) * kCounterSlots ) );

	return PBR_OK;
}

void pbr_destroy( pbr_ctx* ctx ) {
	if( ctx == nullptr ) {
		return;
	}

	if( ctx->stream != nullptr ) {
		(void) hipSetDevice( ctx->device );
		(void) hipStreamSynchronize( ctx->stream );
		freeScene( ctx );
		freeImages( ctx );
		(void) hipFree( ctx->dSeeds );
		(void) hipHostFree( ctx->hSeeds );
		(void) hipFree( ctx->dCounters );
		(void) hipFree( ctx->dWork );
		(void) hipHostFree( ctx->dGuard );
		(void) hipEventDestroy( ctx->evStart );
		(void) hipEventDestroy( ctx->evTraceStart );
		(void) hipEventDestroy( ctx->evTraceStop );
		(void) hipEventDestroy( ctx->evStop );
		(void) hipStreamDestroy( ctx->stream );
	}

	delete ctx;
}

const char* pbr_last_error( const pbr_ctx* ctx ) {
	return ( ctx != nullptr ) ? ctx->error.c_str() : "null context";
}

}  // extern "C"

namespace {

// Everything pbr_upload_scene checks before it touches the device: every index the kernels will follow.
// face0s / links (optional): per node, first face or -1 (container) and second face / miss link.
int checkScene( pbr_ctx* ctx, const pbr_scene_desc* s, std::vector<int>* face0sOut, std::vector<int>* linksOut ) {
	if( s == nullptr || s->bvh == nullptr || s->facesV == nullptr || s->vertices == nullptr || s->materials == nullptr ) {
		return fail( ctx, PBR_EINVAL, "scene: null array" );
	}
	if( s->num_nodes < 2 || s->num_faces == 0 || s->num_vertices == 0 || s->num_materials == 0 ) {
		return fail( ctx, PBR_EINVAL, "scene: needs >= 2 BVH nodes (the root is never tested, pt_bvh.cl:84), faces, vertices and materials" );
	}
	if( s->brdf > 1 ) {
		return fail( ctx, PBR_EINVAL, "scene: brdf must be 0 or 1" );
	}
	if( s->num_lights > 0 && s->lights == nullptr ) {
		return fail( ctx, PBR_EINVAL, "scene: num_lights > 0 but lights is null" );
	}
	if( s->num_nodes > ( 1u << 24 ) || s->num_faces > ( 1u << 24 ) ) {
		return fail( ctx, PBR_EINVAL, "scene: node / face indices are stored as floats and must stay below 2^24" );
	}

	// ---- nodes: validate every link the walk can follow ----
	std::vector<int> face0s( s->num_nodes ), links( s->num_nodes );

	for( uint32_t i = 0; i < s->num_nodes; i++ ) {
		const pbr_bvh_node& n = s->bvh[i];

		if( n.bbMin.w == -1.0f ) {
			// container node: miss link in [-1, N) (the walk stops outside (0, N), pt_bvh.cl:122)
			if( !integral( n.bbMax.w, -1.0, (double) s->num_nodes - 1.0 ) ) {
				return fail( ctx, PBR_EINVAL, "node %u: miss link %g is not an index", i, (double) n.bbMax.w );
			}

			// ... and FORWARD: the stackless walk has no visited set, so a ray that keeps missing a box whose link
			// points at or before it would circle forever (the reference's flattening only emits links behind the
			// subtree, PathTracer.cpp:300-330; -1 and 0 end the walk)
			if( i > 0 && n.bbMax.w > 0.0f && n.bbMax.w <= (float) i ) {
				return fail( ctx, PBR_EINVAL, "node %u: miss link %g must point forward (or be -1 / 0 = end of the walk)", i, (double) n.bbMax.w );
			}

			face0s[i] = -1;
			links[i] = (int) n.bbMax.w;
		}
		else if( integral( n.bbMin.w, 0.0, (double) s->num_faces - 1.0 ) ) {
			// leaf: the second face, if any, is the next one in leaf order (PathTracer.cpp:267-268)
			if( !( n.bbMax.w == -1.0f || n.bbMax.w == n.bbMin.w + 1.0f ) || n.bbMax.w > (float) ( s->num_faces - 1 ) ) {
				return fail( ctx, PBR_EINVAL, "node %u: second face %g is neither -1 nor first face + 1", i, (double) n.bbMax.w );
			}

			face0s[i] = (int) n.bbMin.w;
			links[i] = (int) n.bbMax.w;
		}
		else {
			return fail( ctx, PBR_EINVAL, "node %u: bbMin.w = %g is neither -1 nor a face index", i, (double) n.bbMin.w );
		}
	}

	if( face0s[s->num_nodes - 1] < 0 ) {
		return fail( ctx, PBR_EINVAL, "node %u: the last node is a container (its children would lie outside the array)", s->num_nodes - 1 );
	}

	// ---- faces ----
	for( uint32_t f = 0; f < s->num_faces; f++ ) {
		const pbr_uint4& fv = s->facesV[f];

		if( fv.x >= s->num_vertices || fv.y >= s->num_vertices || fv.z >= s->num_vertices ) {
			return fail( ctx, PBR_EINVAL, "face %u: vertex index out of range", f );
		}
		if( fv.w >= s->num_materials ) {
			return fail( ctx, PBR_EINVAL, "face %u: material index %u out of range (faces without usemtl carry -1)", f, fv.w );
		}
	}

	if( face0sOut != nullptr ) {
		face0sOut->swap( face0s );
	}
	if( linksOut != nullptr ) {
		linksOut->swap( links );
	}

	return PBR_OK;
}

}  // namespace

extern "C" {

int pbr_validate_scene( const pbr_scene_desc* s, char* message, size_t capacity ) {
	pbr_ctx scratch;   // no device behind it: only its error string is used
	const int status = checkScene( &scratch, s, nullptr, nullptr );

	if( message != nullptr && capacity > 0 ) {
		std::snprintf( message, capacity, "%s", scratch.error.c_str() );
	}

	return status;
}

int pbr_upload_scene( pbr_ctx* ctx, const pbr_scene_desc* s ) {
	if( ctx == nullptr || ctx->stream == nullptr ) {
		return fail( ctx, PBR_ESTATE, "context is not usable" );
	}

	std::vector<int> face0s, links;
	{
		const int checked = checkScene( ctx, s, &face0s, &links );

		if( checked != PBR_OK ) {
			return checked;
		}
	}

	// ---- hot nodes: rank by expected visit frequency ----
	// A node is visited when its parent's box was hit, i.e. (for rays without preferred
	// position) in proportion to the parent's surface area.  In the DFS array a container's
	// subtree is [i + 1, escape) with escape = its miss link (or N), so parents fall out of one
	// stack walk.  Measured against real visit histograms this ranking captures 56 % / 40 % /
	// 18 % of all node visits with 1024 slots (Sponza- / Dragon-class / hairball), within 4
	// points of the best possible choice.
	const uint32_t N = s->num_nodes;
	std::vector<double> weight( N, 0.0 );
	{
		auto area = [&]( uint32_t i ) {
			const pbr_bvh_node& n = s->bvh[i];
			const double dx = std::fabs( (double) n.bbMax.x - n.bbMin.x );
			const double dy = std::fabs( (double) n.bbMax.y - n.bbMin.y );
			const double dz = std::fabs( (double) n.bbMax.z - n.bbMin.z );
			return 2.0 * ( dx * dy + dz * dy + dx * dz );
		};
		std::vector<std::pair<uint32_t, double>> stack;   // (escape, area)
		const double rootArea = area( 0 );

		for( uint32_t i = 0; i < N; i++ ) {
			while( !stack.empty() && i >= stack.back().first ) {
				stack.pop_back();
			}

			weight[i] = stack.empty() ? rootArea : stack.back().second;

			if( face0s[i] < 0 ) {
				const uint32_t escape = ( links[i] > (int) i ) ? (uint32_t) links[i] : N;
				stack.push_back( std::make_pair( escape, area( i ) ) );
			}
		}
	}

	// at most what one block can stage with a CU's 160 KB of LDS to itself
	const uint32_t maxHot = ( 160 * 1024 - 256 ) / 32;
	std::vector<uint32_t> ranked;
	ranked.reserve( N );

	for( uint32_t i = 1; i < N; i++ ) {   // node 0 (the root) is never fetched
		ranked.push_back( i );
	}

	const uint32_t numHot = (uint32_t) std::min<size_t>( ranked.size(), maxHot );
	std::partial_sort( ranked.begin(), ranked.begin() + numHot, ranked.end(), [&]( uint32_t a, uint32_t b ) {
		return ( weight[a] != weight[b] ) ? ( weight[a] > weight[b] ) : ( a < b );
	} );

	// ---- the node stream (pt_kernel.hpp decodeNode): hot nodes by rank, then the rest ----
	std::vector<int> recordOf( (size_t) N, -1 );   // node index -> record; the root has none

	for( uint32_t r = 0; r < numHot; r++ ) {
		recordOf[ranked[r]] = (int) r;
	}

	const size_t numRecords = N;   // records the stream holds (the last one is padding)

	{
		// the rest in DFS order: a cold node's hit successor is the adjacent 32 B.  (Any order is legal — every record names
		// its successors.  Treelets, a connected piece of the tree per 128-byte line, were built and measured in round 3:
		// -13 ... -19 % distinct lines per ray offline, +0.0 / +0.2 / +0.7 % on the GPU; lab/src/node_stream_treelets.txt.)
		int next = (int) numHot;

		for( uint32_t i = 1; i < N; i++ ) {
			if( recordOf[i] < 0 ) {
				recordOf[i] = next++;
			}
		}
	}

	// the walk stops outside (0, N), pt_bvh.cl:122
	// a reference is the record's byte offset in the stream (pt_kernel.hpp, Cursor): record * 32 < 2^29 for N <= 2^24
	auto refOf = [&]( long long node ) {
		return ( node > 0 && node < (long long) N ) ? recordOf[(size_t) node] * 32 : -1;
	};

	if( numRecords * 32 >= ( (size_t) 1 << 31 ) ) {
		return fail( ctx, PBR_EINVAL, "upload_scene: the node stream would exceed 2 GiB (record references are 31-bit byte offsets)" );
	}

	std::vector<float4> nodes( numRecords * 2, make_float4( 0.0f, 0.0f, 0.0f, 0.0f ) );

	for( uint32_t i = 1; i < N; i++ ) {
		const pbr_bvh_node& n = s->bvh[i];
		int w0, w1;

		if( face0s[i] < 0 ) {
			w0 = refOf( (long long) i + 1 );
			w1 = refOf( links[i] );
		}
		else {
			w0 = (int) ( 0x80000000u | ( ( links[i] >= 0 ) ? 0x40000000u : 0u ) | (uint32_t) face0s[i] );
			w1 = refOf( (long long) i + 1 );
		}

		auto boxWord = []( float x ) { return x; };
		const size_t r = (size_t) recordOf[i];
		nodes[r * 2 + 0] = make_float4( boxWord( n.bbMin.x ), boxWord( n.bbMin.y ), boxWord( n.bbMax.x ), boxWord( n.bbMax.y ) );
		nodes[r * 2 + 1] = make_float4( boxWord( n.bbMin.z ), boxWord( n.bbMax.z ), __builtin_bit_cast( float, w0 ), __builtin_bit_cast( float, w1 ) );
	}

	// ---- faces: gather the corners; store a, b - a, c - a (what pt_intersect.cl:98-99 computes) ----
	std::vector<float4> tris( (size_t) ( s->num_faces + 1 ) * 3, make_float4( 0.0f, 0.0f, 0.0f, 0.0f ) );   // + 1: testLeaf reads face + 1 ahead

	for( uint32_t f = 0; f < s->num_faces; f++ ) {
		const pbr_uint4& fv = s->facesV[f];

		const pbr_float4& a = s->vertices[fv.x];
		const pbr_float4& b = s->vertices[fv.y];
		const pbr_float4& c = s->vertices[fv.z];
		const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
		const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
		const int material = (int) fv.w;

		tris[(size_t) f * 3 + 0] = make_float4( a.x, a.y, a.z, e1x );
		tris[(size_t) f * 3 + 1] = make_float4( e1y, e1z, e2x, e2y );
		tris[(size_t) f * 3 + 2] = make_float4( e2z, __builtin_bit_cast( float, material ), 0.0f, 0.0f );
	}

	// ---- Phong tessellation input: the exact corners and their vertex normals, gathered per face ----
	// (pt_intersect.cl:146-157 gathers them through facesV / facesN per test).  The default flat test never reads
	// facesN, so scenes whose normal indices are unusable stay valid; they just cannot be configured with PHONGTESS.
	std::vector<float4> triPN;
	{
		bool usable = ( s->facesN != nullptr && s->normals != nullptr && s->num_normals > 0 );

		for( uint32_t f = 0; usable && f < s->num_faces; f++ ) {
			const pbr_uint4& fn = s->facesN[f];
			usable = ( fn.x < s->num_normals && fn.y < s->num_normals && fn.z < s->num_normals );
		}

		if( usable ) {
			triPN.resize( (size_t) s->num_faces * 6 );

			for( uint32_t f = 0; f < s->num_faces; f++ ) {
				const pbr_uint4& fv = s->facesV[f];
				const pbr_uint4& fn = s->facesN[f];
				const uint32_t vi[3] = { fv.x, fv.y, fv.z }, ni[3] = { fn.x, fn.y, fn.z };

				for( int k = 0; k < 3; k++ ) {
					const pbr_float4& v = s->vertices[vi[k]];
					const pbr_float4& n = s->normals[ni[k]];
					triPN[(size_t) f * 6 + k] = make_float4( v.x, v.y, v.z, 0.0f );
					triPN[(size_t) f * 6 + 3 + k] = make_float4( n.x, n.y, n.z, 0.0f );
				}
			}
		}
	}

	// ---- materials: one 64-byte shape for both BRDFs ----
	std::vector<float4> mats( (size_t) s->num_materials * 4 );

	for( uint32_t i = 0; i < s->num_materials; i++ ) {
		if( s->brdf == 0 ) {
			const pbr_material_schlick& m = ( (const pbr_material_schlick*) s->materials )[i];
			mats[(size_t) i * 4 + 0] = make_float4( m.data[0], m.data[1], m.data[2], m.data[3] );
			mats[(size_t) i * 4 + 1] = make_float4( 0.0f, 0.0f, 0.0f, 0.0f );
			mats[(size_t) i * 4 + 2] = make_float4( m.rgbDiff.x, m.rgbDiff.y, m.rgbDiff.z, 0.0f );
			mats[(size_t) i * 4 + 3] = make_float4( m.rgbSpec.x, m.rgbSpec.y, m.rgbSpec.z, 0.0f );
		}
		else {
			const pbr_material_sa& m = ( (const pbr_material_sa*) s->materials )[i];
			mats[(size_t) i * 4 + 0] = make_float4( m.data[0], m.data[1], m.data[2], m.data[3] );
			mats[(size_t) i * 4 + 1] = make_float4( m.data[4], m.data[5], 0.0f, 0.0f );
			mats[(size_t) i * 4 + 2] = make_float4( m.rgbDiff.x, m.rgbDiff.y, m.rgbDiff.z, 0.0f );
			mats[(size_t) i * 4 + 3] = make_float4( m.rgbSpec.x, m.rgbSpec.y, m.rgbSpec.z, 0.0f );
		}
	}

	const uint32_t lightSlots = ( s->num_lights > 0 ) ? s->num_lights : 1;
	std::vector<float4> lights( (size_t) lightSlots * 3, make_float4( 0.0f, 0.0f, 0.0f, 0.0f ) );

	for( uint32_t i = 0; i < s->num_lights; i++ ) {
		const pbr_light& l = s->lights[i];
		lights[(size_t) i * 3 + 0] = make_float4( l.pos.x, l.pos.y, l.pos.z, l.pos.w );
		lights[(size_t) i * 3 + 1] = make_float4( l.rgb.x, l.rgb.y, l.rgb.z, l.rgb.w );
		lights[(size_t) i * 3 + 2] = make_float4( l.data.x, l.data.y, l.data.z, l.data.w );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	freeScene( ctx );

	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dNodes, sizeof( float4 ) * nodes.size() ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dTris, sizeof( float4 ) * tris.size() ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dMats, sizeof( float4 ) * mats.size() ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dLights, sizeof( float4 ) * lights.size() ) );
	HIP_TRY( ctx, hipMemcpy( ctx->dNodes, nodes.data(), sizeof( float4 ) * nodes.size(), hipMemcpyHostToDevice ) );
	ctx->nodeBytes = sizeof( float4 ) * nodes.size();
	ctx->triBytes = sizeof( float4 ) * tris.size();
	ctx->walkBytes = 0;

	HIP_TRY( ctx, hipMemcpy( ctx->dTris, tris.data(), sizeof( float4 ) * tris.size(), hipMemcpyHostToDevice ) );

	if( !triPN.empty() ) {
		HIP_TRY( ctx, hipMalloc( (void**) &ctx->dTriPN, sizeof( float4 ) * triPN.size() ) );
		HIP_TRY( ctx, hipMemcpy( ctx->dTriPN, triPN.data(), sizeof( float4 ) * triPN.size(), hipMemcpyHostToDevice ) );
	}
	HIP_TRY( ctx, hipMemcpy( ctx->dMats, mats.data(), sizeof( float4 ) * mats.size(), hipMemcpyHostToDevice ) );
	HIP_TRY( ctx, hipMemcpy( ctx->dLights, lights.data(), sizeof( float4 ) * lights.size(), hipMemcpyHostToDevice ) );

	// the face normals the shading would recompute on every hit, evaluated once by the shading's own device function
	{
		if( s->num_faces > 0 && ctx->knobs.faceNormals != 0 ) {
			HIP_TRY( ctx, hipMalloc( (void**) &ctx->dFaceN, sizeof( float4 ) * s->num_faces ) );
			DevParams P;
			std::memset( &P, 0, sizeof( P ) );
			P.tris = ctx->dTris;
			hipLaunchKernelGGL( ptk::prepareFaceNormals, dim3( ( s->num_faces + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, P, ctx->dFaceN, (int) s->num_faces );
			HIP_TRY( ctx, hipGetLastError() );
		}

		HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	}

	ctx->numHotAvail = numHot;
	ctx->firstRef = recordOf[1] * 32;
	ctx->hostNodes.assign( s->bvh, s->bvh + N );
	ctx->hostFace0s.swap( face0s );
	ctx->hostLinks.swap( links );
	ctx->hostRanked.assign( ranked.begin(), ranked.begin() + numHot );
	ctx->numNodes = s->num_nodes;
	ctx->numFaces = s->num_faces;
	ctx->numMaterials = s->num_materials;
	ctx->numLights = s->num_lights;
	ctx->sceneBrdf = s->brdf;
	ctx->plansBuilt = false;
	resetTuning( ctx );   // a new scene / configuration is tuned afresh
	ctx->hasScene = true;

	return prepareWalk( ctx );   // a context that is already configured for a ray-ordered walk: its streams, now
}

int pbr_configure( pbr_ctx* ctx, const pbr_config* cfg ) {
	if( ctx == nullptr || ctx->stream == nullptr ) {
		return fail( ctx, PBR_ESTATE, "context is not usable" );
	}
	if( cfg == nullptr ) {
		return fail( ctx, PBR_EINVAL, "configure: null config" );
	}
	if( cfg->width == 0 || cfg->height == 0 || ( cfg->width & 7u ) || ( cfg->height & 7u ) ) {
		return fail( ctx, PBR_EINVAL, "width / height must be non-zero multiples of 8 (opencl.localgroupsize, config.json:85)" );
	}
	if( cfg->brdf > 1 || cfg->shadow_rays > 1 ) {
		return fail( ctx, PBR_EINVAL, "brdf and shadow_rays must be 0 or 1" );
	}
	if( cfg->max_depth == 0 || cfg->samples == 0 ) {
		return fail( ctx, PBR_EINVAL, "max_depth and samples must be >= 1" );
	}
	if( cfg->phong_tessellation > 0.0f && ctx->hasScene && ctx->dTriPN == nullptr ) {
		return fail( ctx, PBR_EINVAL, "Phong tessellation needs vertex normals: the uploaded scene's facesN / normals are missing or out of range" );
	}
	if( cfg->traversal > 3 || cfg->arith > 1 ) {
		return fail( ctx, PBR_EINVAL, "traversal must be 0 (the reference's walk), 1 (six orders), 2 (eight orders) or 3 (eight orders, compact records); arith 0 (exact) or 1 (native)" );
	}
	if( cfg->tile_world == 0 || cfg->tile_rank >= cfg->tile_world ) {
		return fail( ctx, PBR_EINVAL, "tile_rank must be < tile_world, tile_world >= 1" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	freeImages( ctx );

	ctx->cfg = *cfg;
	ctx->tilesX = (int) ( cfg->width / 8 );
	ctx->tilesY = (int) ( cfg->height / 8 );
	ctx->numTiles = ctx->tilesX * ctx->tilesY;
	// tiles t = j * world + rank, j = 0 .. : count those below numTiles
	ctx->numLocalTiles = ( ctx->numTiles - (int) cfg->tile_rank + (int) cfg->tile_world - 1 ) / (int) cfg->tile_world;

	// Every context can hold the full image (import_tiles scatters all ranks' tiles into imgOut).
	const size_t fullBytes = sizeof( float4 ) * 64 * (size_t) ctx->numTiles;
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dImgIn, fullBytes ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dImgOut, fullBytes ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dImgDbg, fullBytes ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dRows, fullBytes ) );
	HIP_TRY( ctx, hipMemset( ctx->dImgIn, 0, fullBytes ) );
	HIP_TRY( ctx, hipMemset( ctx->dImgOut, 0, fullBytes ) );
	HIP_TRY( ctx, hipMemset( ctx->dImgDbg, 0, fullBytes ) );
	HIP_TRY( ctx, hipMemset( ctx->dCounters, 0, sizeof( unsigned long long ) * kCounterSlots ) );
	HIP_TRY( ctx, hipDeviceSynchronize() );   // the memsets ran on the null stream; launches use ctx->stream

	// the local tiles as a grid for the banded queue: its true shape when unsharded, about that when sharded
	ctx->queueWidth = std::max( 1, ( ctx->tilesX + (int) cfg->tile_world - 1 ) / (int) cfg->tile_world );
	ctx->queueRows = ( ctx->numLocalTiles + ctx->queueWidth - 1 ) / ctx->queueWidth;
	spatialTileOrder( ctx, &ctx->hTileOrder, ctx->bandFirst );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dTileOrder, sizeof( unsigned ) * ctx->hTileOrder.size() ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dCostOrder, sizeof( unsigned ) * ctx->hTileOrder.size() ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dLastOrder, sizeof( unsigned ) * std::max<size_t>( 1, ctx->hTileOrder.size() ) ) );
	HIP_TRY( ctx, hipMalloc( (void**) &ctx->dTileCost, sizeof( float ) * std::max<size_t>( 1, ctx->hTileOrder.size() ) ) );
	ctx->costLearnt = false;
	ctx->launchesSinceLearn = 0;
	{
		const int uploaded = uploadTileOrder( ctx );

		if( uploaded != PBR_OK ) {
			return uploaded;
		}
	}
	ctx->plansBuilt = false;
	resetTuning( ctx );   // a new scene / configuration is tuned afresh
	ctx->configured = true;

	return prepareWalk( ctx );
}

int pbr_write_input( pbr_ctx* ctx, const float* rgba ) {
	if( ctx == nullptr || !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "write_input before pbr_configure" );
	}
	if( rgba == nullptr ) {
		return fail( ctx, PBR_EINVAL, "write_input: null source" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	HIP_TRY( ctx, hipMemcpyAsync( ctx->dRows, rgba, sizeof( float4 ) * ctx->cfg.width * ctx->cfg.height, hipMemcpyHostToDevice, ctx->stream ) );
	const size_t n = (size_t) ctx->numLocalTiles * 64;
	hipLaunchKernelGGL( ptk::retile, dim3( (unsigned) ( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, ctx->stream,
		ctx->dRows, ctx->dImgIn, (int) ctx->cfg.width, ctx->numLocalTiles, ctx->tilesX, (int) ctx->cfg.tile_world, (int) ctx->cfg.tile_rank );
	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	return PBR_OK;
}

int pbr_reset_accum( pbr_ctx* ctx ) {
	if( ctx == nullptr || !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "reset_accum before pbr_configure" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	const size_t fullBytes = sizeof( float4 ) * 64 * (size_t) ctx->numTiles;
	HIP_TRY( ctx, hipMemsetAsync( ctx->dImgIn, 0, fullBytes, ctx->stream ) );
	HIP_TRY( ctx, hipMemsetAsync( ctx->dCounters, 0, sizeof( unsigned long long ) * kCounterSlots, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	return PBR_OK;
}

int pbr_render_frame( pbr_ctx* ctx, float seed, float pixelWeight, float pxDim, const pbr_camera* cam ) {
	if( ctx == nullptr ) {
		return PBR_EINVAL;
	}

	return launch( ctx, 0, 1, &seed, true, pixelWeight, pxDim, cam );
}

int pbr_accumulate( pbr_ctx* ctx ) {
	if( ctx == nullptr || !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "accumulate before pbr_configure" );
	}

	float4* tmp = ctx->dImgIn;
	ctx->dImgIn = ctx->dImgOut;
	ctx->dImgOut = tmp;
	return PBR_OK;
}

int pbr_render( pbr_ctx* ctx, uint32_t first_sample_count, uint32_t n_frames, const float* seeds, float pxDim, const pbr_camera* cam ) {
	if( ctx == nullptr ) {
		return PBR_EINVAL;
	}
	if( cam != nullptr && cam->focusPoint[0] >= 0 && cam->focusPoint[1] >= 0 ) {
		return fail( ctx, PBR_EINVAL, "pbr_render needs focusPoint < 0; with depth of field call pbr_render_frame + pbr_accumulate per frame" );
	}

	const int status = launch( ctx, first_sample_count, n_frames, seeds, false, 0.0f, pxDim, cam );

	if( status != PBR_OK ) {
		return status;
	}

	// result in imageOut AND imageIn
	const size_t bytes = sizeof( float4 ) * 64 * (size_t) ctx->numLocalTiles;
	HIP_TRY( ctx, hipMemcpyAsync( ctx->dImgIn, ctx->dImgOut, bytes, hipMemcpyDeviceToDevice, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	return PBR_OK;
}

int pbr_read_output( pbr_ctx* ctx, float* rgba ) {
	if( ctx == nullptr ) {
		return PBR_EINVAL;
	}
	return readTiled( ctx, ctx->dImgOut, rgba, (int) ctx->cfg.tile_world, (int) ctx->cfg.tile_rank );
}

int pbr_get_focus_depth( pbr_ctx* ctx, int x, int y, float* t, int* owned ) {
	if( ctx == nullptr || t == nullptr || owned == nullptr ) {
		return fail( ctx, PBR_EINVAL, "get_focus_depth: null argument" );
	}
	if( !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "get_focus_depth before pbr_configure" );
	}

	// CLAMP_TO_EDGE, as the sampler of getPreviousFocus has it (pathtracing.cl:58-65)
	const int fx = std::max( 0, std::min( (int) ctx->cfg.width - 1, x ) );
	const int fy = std::max( 0, std::min( (int) ctx->cfg.height - 1, y ) );
	const int tile = ( fy >> 3 ) * ctx->tilesX + ( fx >> 3 );
	*t = 0.0f;
	const int position = ptk::dealPositionOfTile( tile, ctx->tilesX, (int) ctx->cfg.tile_world );
	*owned = ( position % (int) ctx->cfg.tile_world == (int) ctx->cfg.tile_rank ) ? 1 : 0;

	if( *owned ) {
		const size_t slot = (size_t) ( position / (int) ctx->cfg.tile_world ) * 64 + (size_t) ( ( fy & 7 ) * 8 + ( fx & 7 ) );
		float4 pixel;
		HIP_TRY( ctx, hipSetDevice( ctx->device ) );
		HIP_TRY( ctx, hipMemcpyAsync( &pixel, ctx->dImgIn + slot, sizeof( pixel ), hipMemcpyDeviceToHost, ctx->stream ) );
		HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
		*t = pixel.w;
	}

	return PBR_OK;
}

int pbr_set_focus_depth( pbr_ctx* ctx, float t ) {
	if( ctx == nullptr ) {
		return PBR_EINVAL;
	}

	ctx->focusGiven = true;
	ctx->focusDepth = t;
	return PBR_OK;
}

int pbr_read_display( pbr_ctx* ctx, uint8_t* rgba8, int top_row_first ) {
	if( ctx == nullptr ) {
		return PBR_EINVAL;
	}
	if( !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "read before pbr_configure" );
	}
	if( rgba8 == nullptr ) {
		return fail( ctx, PBR_EINVAL, "read: null destination" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	const dim3 block( 64, 4 );
	const dim3 grid( ( ctx->cfg.width + 63 ) / 64, ( ctx->cfg.height + 3 ) / 4 );
	// dRows (16 B per pixel) doubles as the 4-B-per-pixel staging buffer
	hipLaunchKernelGGL( ptk::displayRGBA8, grid, block, 0, ctx->stream, (const float4*) ctx->dImgOut, (uchar4*) ctx->dRows,
		(int) ctx->cfg.width, (int) ctx->cfg.height, ctx->tilesX, (int) ctx->cfg.tile_world, (int) ctx->cfg.tile_rank, top_row_first ? 1 : 0 );
	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipMemcpyAsync( rgba8, ctx->dRows, (size_t) 4 * ctx->cfg.width * ctx->cfg.height, hipMemcpyDeviceToHost, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	return PBR_OK;
}

int pbr_read_debug( pbr_ctx* ctx, float* rgba ) {
	if( ctx == nullptr ) {
		return PBR_EINVAL;
	}
	return readTiled( ctx, ctx->dImgDbg, rgba, (int) ctx->cfg.tile_world, (int) ctx->cfg.tile_rank );
}

int pbr_get_counters( pbr_ctx* ctx, pbr_counters* out ) {
	if( ctx == nullptr || out == nullptr || ctx->stream == nullptr ) {
		return fail( ctx, PBR_EINVAL, "get_counters: null argument" );
	}

	unsigned long long host[4] = { 0, 0, 0, 0 };
	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	HIP_TRY( ctx, hipMemcpy( host, ctx->dCounters, sizeof( host ), hipMemcpyDeviceToHost ) );
	out->nodes = host[0];
	out->tris = host[1];
	out->hits = host[2];
	out->paths = host[3];
	return PBR_OK;
}

double pbr_last_kernel_ms( const pbr_ctx* ctx ) {
	return ( ctx != nullptr ) ? ctx->lastKernelMs : 0.0;
}

uint64_t pbr_tile_bytes( const pbr_ctx* ctx ) {
	if( ctx == nullptr || !ctx->configured ) {
		return 0;
	}

	const uint64_t perRank = (uint64_t) ( ( ctx->numTiles + (int) ctx->cfg.tile_world - 1 ) / (int) ctx->cfg.tile_world );
	return perRank * 1024u;
}

int pbr_export_tiles( pbr_ctx* ctx, void* d_dst ) {
	if( ctx == nullptr || !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "export_tiles before pbr_configure" );
	}
	if( d_dst == nullptr ) {
		return fail( ctx, PBR_EINVAL, "export_tiles: null destination" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	const size_t bytes = sizeof( float4 ) * 64 * (size_t) ctx->numLocalTiles;
	HIP_TRY( ctx, hipMemsetAsync( d_dst, 0, (size_t) pbr_tile_bytes( ctx ), ctx->stream ) );
	HIP_TRY( ctx, hipMemcpyAsync( d_dst, ctx->dImgOut, bytes, hipMemcpyDeviceToDevice, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	return PBR_OK;
}

int pbr_import_tiles( pbr_ctx* ctx, const void* d_all ) {
	if( ctx == nullptr || !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "import_tiles before pbr_configure" );
	}
	if( d_all == nullptr ) {
		return fail( ctx, PBR_EINVAL, "import_tiles: null source" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );

	if( ctx->dFull == nullptr ) {
		HIP_TRY( ctx, hipMalloc( (void**) &ctx->dFull, sizeof( float4 ) * 64 * (size_t) ctx->numTiles ) );
	}

	const int perRank = ( ctx->numTiles + (int) ctx->cfg.tile_world - 1 ) / (int) ctx->cfg.tile_world;
	const size_t n = (size_t) ctx->numTiles * 64;
	hipLaunchKernelGGL( ptk::scatterGathered, dim3( (unsigned) ( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, ctx->stream,
		(const float4*) d_all, ctx->dFull, ctx->numTiles, perRank, (int) ctx->cfg.tile_world, ctx->tilesX );
	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	return PBR_OK;
}

int pbr_read_full( pbr_ctx* ctx, float* rgba ) {
	if( ctx == nullptr ) {
		return PBR_EINVAL;
	}
	return readTiled( ctx, ctx->dFull, rgba, 1, 0 );
}


// ---- diagnostic entry points (include/pbr_hip_diag.h) -----------------------------------

namespace {

struct DevBuf {
	void* p = nullptr;
	~DevBuf() { (void) hipFree( p ); }
	hipError_t alloc( size_t bytes ) { return hipMalloc( &p, bytes ? bytes : 4 ); }
};

// The scene for the diagnostic and denoise kernels, walked in the configured order (pbr_config.traversal); hotAvail: the
// ranked prefix of that stream.  *status: PBR_OK, or why the configured walk's streams could not be built (the caller fails
// with it — no silent walk in another order).
DevParams sceneParams( pbr_ctx* ctx, int* status, uint32_t* hotAvail = nullptr ) {
	DevParams P;
	std::memset( &P, 0, sizeof( P ) );
	uint32_t hot = 0;
	*status = applyWalk( ctx, &P, &hot );

	if( hotAvail != nullptr ) {
		*hotAvail = hot;
	}

	P.parkEighths = 4;
	P.tris = ctx->dTris;
	P.faceN = ctx->dFaceN;
	P.mats = ctx->dMats;
	P.lights = ctx->dLights;
	P.guard = ctx->dGuard;
	P.numNodes = (int) ctx->numNodes;
	P.numLights = (int) ctx->numLights;
	P.numHot = 0;
	P.numHotBytes = 0;
	return P;
}

}  // namespace

// The denoise half of the display step (csrc/pt_denoise.hpp): first-hit features from one primary ray per pixel, then
// edge-avoiding a-trous passes over the accumulated image.  Leaves the accumulation untouched.
int pbr_denoise( pbr_ctx* ctx, float pxDim, const pbr_camera* cam, const pbr_denoise_params* params, float* rgba, float* features ) {
	if( ctx == nullptr || ctx->stream == nullptr ) {
		return PBR_EINVAL;
	}
	if( !ctx->hasScene || !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "denoise before pbr_upload_scene / pbr_configure" );
	}
	if( cam == nullptr || params == nullptr || rgba == nullptr ) {
		return fail( ctx, PBR_EINVAL, "denoise: null camera, parameters or destination" );
	}
	if( params->passes < 1 || params->passes > 8 ) {
		return fail( ctx, PBR_EINVAL, "denoise: 1 .. 8 passes (got %u)", params->passes );
	}

	const float sigmas[4] = { params->sigma_color, params->sigma_normal, params->sigma_world, params->sigma_albedo };

	for( int k = 0; k < 4; k++ ) {
		if( !( sigmas[k] >= 0.0f ) || !std::isfinite( sigmas[k] ) ) {
			return fail( ctx, PBR_EINVAL, "denoise: standard deviations must be finite and >= 0 (0 switches a feature off)" );
		}
	}

	const bool sharded = ctx->cfg.tile_world > 1;

	if( sharded && ctx->dFull == nullptr ) {
		return fail( ctx, PBR_ESTATE, "denoise with tile sharding filters the gathered frame: call pbr_import_tiles first" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	const int w = (int) ctx->cfg.width, h = (int) ctx->cfg.height;
	const size_t pixels = (size_t) w * (size_t) h;
	DevBuf dPosition, dNormal, dAlbedo, dPing, dPong;
	HIP_TRY( ctx, dPosition.alloc( sizeof( float4 ) * pixels ) );
	HIP_TRY( ctx, dNormal.alloc( sizeof( float4 ) * pixels ) );
	HIP_TRY( ctx, dAlbedo.alloc( sizeof( float4 ) * pixels ) );
	HIP_TRY( ctx, dPing.alloc( sizeof( float4 ) * pixels ) );
	HIP_TRY( ctx, dPong.alloc( sizeof( float4 ) * pixels ) );

	int walkStatus = PBR_OK;
	DevParams P = sceneParams( ctx, &walkStatus );

	if( walkStatus != PBR_OK ) {
		return walkStatus;
	}
	const float* src[4] = { &cam->eye.x, &cam->w.x, &cam->u.x, &cam->v.x };
	float* dst[4] = { P.eye, P.cw, P.cu, P.cv };

	for( int i = 0; i < 4; i++ ) {
		for( int k = 0; k < 3; k++ ) {
			dst[i][k] = src[i][k];
		}
	}

	for( int k = 0; k < 3; k++ ) {   // as launch() does for initRay
		const float cuW = P.cu[k] * (float) w;
		P.camA[k] = P.cu[k] - cuW;
		P.cvH[k] = P.cv[k] * (float) h;
	}

	P.halfPx = pxDim * 0.5f;
	P.pxDim = pxDim;
	P.width = w;
	P.height = h;

	const dim3 block( 64, 4 );
	const dim3 grid( ( w + 63 ) / 64, ( h + 3 ) / 4 );
	HIP_TRY( ctx, hipEventRecord( ctx->evStart, ctx->stream ) );
	hipLaunchKernelGGL( ptk::untile, grid, block, 0, ctx->stream, (const float4*) ( sharded ? ctx->dFull : ctx->dImgOut ), (float4*) dPing.p,
		w, h, ctx->tilesX, 1, 0 );
	hipLaunchKernelGGL( ptd::firstHitFeatures, grid, block, 0, ctx->stream, P, (float4*) dPosition.p, (float4*) dNormal.p, (float4*) dAlbedo.p );
	HIP_TRY( ctx, hipGetLastError() );

	auto inverseSquare = []( float sigma ) { return ( sigma > 0.0f ) ? 1.0f / ( sigma * sigma ) : 0.0f; };
	float4* in = (float4*) dPing.p;
	float4* out = (float4*) dPong.p;

	for( uint32_t pass = 0; pass < params->passes; pass++ ) {
		ptd::DenoiseArgs A;
		A.width = w;
		A.height = h;
		A.step = 1 << pass;
		// the colour's standard deviation halves from pass to pass (Dammertz et al. 2010, section 3.3): what the early
		// passes smoothed, the late, wide ones must not blur again
		A.invColor = inverseSquare( params->sigma_color / (float) ( 1u << pass ) );
		A.invNormal = inverseSquare( params->sigma_normal );
		A.invAlbedo = inverseSquare( params->sigma_albedo );
		A.worldScale = params->sigma_world * (float) A.step * pxDim;
		hipLaunchKernelGGL( ptd::atrousPass, grid, block, 0, ctx->stream, A, (const float4*) in, out,
			(const float4*) dPosition.p, (const float4*) dNormal.p, (const float4*) dAlbedo.p );
		std::swap( in, out );
	}

	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipEventRecord( ctx->evStop, ctx->stream ) );
	HIP_TRY( ctx, hipMemcpyAsync( rgba, in, sizeof( float4 ) * pixels, hipMemcpyDeviceToHost, ctx->stream ) );

	if( features != nullptr ) {
		HIP_TRY( ctx, hipMemcpyAsync( features, dPosition.p, sizeof( float4 ) * pixels, hipMemcpyDeviceToHost, ctx->stream ) );
		HIP_TRY( ctx, hipMemcpyAsync( features + 4 * pixels, dNormal.p, sizeof( float4 ) * pixels, hipMemcpyDeviceToHost, ctx->stream ) );
		HIP_TRY( ctx, hipMemcpyAsync( features + 8 * pixels, dAlbedo.p, sizeof( float4 ) * pixels, hipMemcpyDeviceToHost, ctx->stream ) );
	}

	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	float ms = 0.0f;
	HIP_TRY( ctx, hipEventElapsedTime( &ms, ctx->evStart, ctx->evStop ) );
	ctx->lastKernelMs = (double) ms;
	return PBR_OK;
}

uint32_t pbr_bvh_node_capacity( uint32_t num_faces ) {
	// every face its own leaf is the most records a binary tree over them can have; at least a container and a leaf
	return ( num_faces < 2 ) ? 2u : 2u * num_faces - 1u;
}

namespace {

// the radix-tree builder of round 1 (Karras 2012), kept for comparison: PBR_BVH_BUILDER=lbvh
int buildRadixTree( pbr_ctx* ctx, ptb::BuildArrays B, uint32_t num_faces, uint32_t* treeNodesOut ) {
	const uint32_t leaves = ( num_faces + 1 ) / 2;
	const uint32_t treeNodes = 2 * leaves - 1;
	DevBuf dLeft, dRight, dParent, dSize, dArrived, dBoxMin, dBoxMax;
	HIP_TRY( ctx, dLeft.alloc( sizeof( int ) * treeNodes ) );
	HIP_TRY( ctx, dRight.alloc( sizeof( int ) * treeNodes ) );
	HIP_TRY( ctx, dParent.alloc( sizeof( int ) * treeNodes ) );
	HIP_TRY( ctx, dSize.alloc( sizeof( unsigned ) * treeNodes ) );
	HIP_TRY( ctx, dArrived.alloc( sizeof( unsigned ) * treeNodes ) );
	HIP_TRY( ctx, dBoxMin.alloc( sizeof( float4 ) * treeNodes ) );
	HIP_TRY( ctx, dBoxMax.alloc( sizeof( float4 ) * treeNodes ) );
	HIP_TRY( ctx, hipMemsetAsync( dParent.p, 0xFF, sizeof( int ) * treeNodes, ctx->stream ) );
	HIP_TRY( ctx, hipMemsetAsync( dArrived.p, 0, sizeof( unsigned ) * treeNodes, ctx->stream ) );
	B.numLeaves = leaves;
	B.left = (int*) dLeft.p;
	B.right = (int*) dRight.p;
	B.parent = (int*) dParent.p;
	B.size = (unsigned*) dSize.p;
	B.arrived = (unsigned*) dArrived.p;
	B.boxMin = (float4*) dBoxMin.p;
	B.boxMax = (float4*) dBoxMax.p;
	const unsigned threads = 256;
	const unsigned leafBlocks = ( leaves + threads - 1 ) / threads;
	const unsigned nodeBlocks = ( treeNodes + threads - 1 ) / threads;

	if( leaves > 1 ) {
		hipLaunchKernelGGL( ptb::radixTree, dim3( leafBlocks ), dim3( threads ), 0, ctx->stream, B );
	}

	hipLaunchKernelGGL( ptb::boxesBottomUp, dim3( leafBlocks ), dim3( threads ), 0, ctx->stream, B );
	hipLaunchKernelGGL( ptb::flatten, dim3( nodeBlocks ), dim3( threads ), 0, ctx->stream, B );
	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );   // the buffers above are freed on return
	*treeNodesOut = treeNodes;
	return PBR_OK;
}

// locally-ordered clustering (csrc/bvh_build.hpp, PlocArrays): rounds of nearest-partner search, merge, compaction
int buildClustered( pbr_ctx* ctx, const ptb::BuildArrays& A, uint32_t num_faces, uint32_t* treeNodesOut ) {
	const uint32_t ids = 2 * num_faces - 1;
	DevBuf dLeft, dRight, dParent, dSize, dFaces, dBoxMin, dBoxMax, dClusters, dClustersNext, dNearest, dFlags, dScan, dTotals, dTemp;
	HIP_TRY( ctx, dLeft.alloc( sizeof( int ) * ids ) );
	HIP_TRY( ctx, dRight.alloc( sizeof( int ) * ids ) );
	HIP_TRY( ctx, dParent.alloc( sizeof( int ) * ids ) );
	HIP_TRY( ctx, dSize.alloc( sizeof( unsigned ) * ids ) );
	HIP_TRY( ctx, dFaces.alloc( sizeof( unsigned ) * ids ) );
	HIP_TRY( ctx, dBoxMin.alloc( sizeof( float4 ) * ids ) );
	HIP_TRY( ctx, dBoxMax.alloc( sizeof( float4 ) * ids ) );
	HIP_TRY( ctx, dClusters.alloc( sizeof( int ) * num_faces ) );
	HIP_TRY( ctx, dClustersNext.alloc( sizeof( int ) * num_faces ) );
	HIP_TRY( ctx, dNearest.alloc( sizeof( int ) * num_faces ) );
	HIP_TRY( ctx, dFlags.alloc( sizeof( unsigned long long ) * num_faces ) );
	HIP_TRY( ctx, dScan.alloc( sizeof( unsigned long long ) * num_faces ) );
	HIP_TRY( ctx, dTotals.alloc( sizeof( unsigned long long ) ) );

	ptb::PlocArrays B;
	B.vertices = A.vertices;
	B.facesV = A.facesV;
	B.facesN = A.facesN;
	B.numFaces = num_faces;
	B.keysSorted = A.keysSorted;
	B.left = (int*) dLeft.p;
	B.right = (int*) dRight.p;
	B.parent = (int*) dParent.p;
	B.size = (unsigned*) dSize.p;
	B.faces = (unsigned*) dFaces.p;
	B.boxMin = (float4*) dBoxMin.p;
	B.boxMax = (float4*) dBoxMax.p;
	B.clusters = (int*) dClusters.p;
	B.clustersNext = (int*) dClustersNext.p;
	B.nearest = (int*) dNearest.p;
	B.flags = (unsigned long long*) dFlags.p;
	B.scan = (unsigned long long*) dScan.p;
	B.totals = (unsigned long long*) dTotals.p;
	B.nodesOut = A.nodesOut;
	B.facesVOut = A.facesVOut;
	B.facesNOut = A.facesNOut;

	// The search radius.  For the reference's walk: 32 (round 2: 4 .. 64 within 3 % of each other on the Sponza- / Dragon-class
	// scenes, 32 the best on the hairball — where the numbers are noisy, because the stackless walk's fixed child order decides more
	// than the tree's area).  For a ray-ordered walk (round 5: the context is configured with pbr_config.traversal != 0) the child
	// order is out of the picture and the radius moves visits and speed monotonically — SMALLER is better: 3 gives 2183 / 2331 /
	// 1800 Msamples/s on the Sponza- / Dragon-class scenes and the hairball against 2112 / 2246 / 1414 at 32
	// (profiles/r05/experiments/ploc_radius_ordered_walk.txt).
	int radius = ( ctx->configured && ctx->cfg.traversal != 0 ) ? 3 : 32;

	if( ctx->knobs.plocRadius >= 1 ) {
		radius = ctx->knobs.plocRadius;
	}

	radius = std::min( std::max( radius, 1 ), PLOC_MAX_RADIUS );
	ctx->lastBvhRadius = radius;
	size_t tempBytes = 0;
	HIP_TRY( ctx, hipcub::DeviceScan::ExclusiveSum( nullptr, tempBytes, B.flags, B.scan, (int) num_faces, ctx->stream ) );
	HIP_TRY( ctx, dTemp.alloc( tempBytes ) );

	const unsigned threads = PLOC_THREADS;
	hipLaunchKernelGGL( ptb::plocInit, dim3( ( num_faces + threads - 1 ) / threads ), dim3( threads ), 0, ctx->stream, B );
	uint32_t count = num_faces, nextNode = num_faces;

	while( count > 1 ) {
		const dim3 grid( ( count + threads - 1 ) / threads );
		hipLaunchKernelGGL( ptb::plocNearest, grid, dim3( threads ), 0, ctx->stream, B, count, radius );
		hipLaunchKernelGGL( ptb::plocFlags, grid, dim3( threads ), 0, ctx->stream, B, count );
		HIP_TRY( ctx, hipcub::DeviceScan::ExclusiveSum( dTemp.p, tempBytes, B.flags, B.scan, (int) count, ctx->stream ) );
		hipLaunchKernelGGL( ptb::plocMerge, grid, dim3( threads ), 0, ctx->stream, B, count, nextNode );
		HIP_TRY( ctx, hipGetLastError() );
		unsigned long long totals = 0;
		HIP_TRY( ctx, hipMemcpyAsync( &totals, B.totals, sizeof( totals ), hipMemcpyDeviceToHost, ctx->stream ) );
		HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
		const uint32_t merged = (uint32_t) ( totals >> 32 ), left = (uint32_t) ( totals & 0xFFFFFFFFull );

		if( merged == 0 || left + merged != count ) {
			// the smallest pair of a total order is always mutual — unless the boxes do not compare (NaN corners)
			return fail( ctx, PBR_EINVAL, "build_bvh: clustering made no progress with %u clusters left (non-finite vertices?)", count );
		}

		nextNode += merged;
		count = left;
		std::swap( B.clusters, B.clustersNext );
	}

	// one cluster is left: the root, the id created last (or face 0 of a single-face scene)
	uint32_t total = 0;
	HIP_TRY( ctx, hipMemcpyAsync( &total, B.size + ( nextNode - 1 ), sizeof( total ), hipMemcpyDeviceToHost, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	hipLaunchKernelGGL( ptb::plocFlatten, dim3( ( nextNode + threads - 1 ) / threads ), dim3( threads ), 0, ctx->stream, B, nextNode, total );
	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	*treeNodesOut = total;
	return PBR_OK;
}

}  // namespace

// BVH on the device (csrc/bvh_build.hpp): Morton keys, radix sort, locally-ordered clustering by surface area,
// depth-first flattening into the reference's node format.
int pbr_build_bvh( pbr_ctx* ctx, const pbr_float4* vertices, uint32_t num_vertices, const pbr_uint4* facesV, const pbr_uint4* facesN,
                   uint32_t num_faces, pbr_bvh_node* nodes_out, uint32_t* num_nodes_out, pbr_uint4* facesV_out, pbr_uint4* facesN_out ) {
	if( ctx == nullptr || ctx->stream == nullptr ) {
		return PBR_EINVAL;
	}
	if( vertices == nullptr || facesV == nullptr || facesN == nullptr || nodes_out == nullptr || num_nodes_out == nullptr || facesV_out == nullptr || facesN_out == nullptr ) {
		return fail( ctx, PBR_EINVAL, "build_bvh: null argument" );
	}
	if( num_faces == 0 || num_vertices == 0 || num_faces > ( 1u << 24 ) ) {
		return fail( ctx, PBR_EINVAL, "build_bvh: needs 1 .. 2^24 faces and at least one vertex" );
	}

	for( uint32_t f = 0; f < num_faces; f++ ) {
		if( facesV[f].x >= num_vertices || facesV[f].y >= num_vertices || facesV[f].z >= num_vertices ) {
			return fail( ctx, PBR_EINVAL, "build_bvh: face %u: vertex index out of range", f );
		}
	}

	for( uint32_t v = 0; v < num_vertices; v++ ) {
		if( !std::isfinite( vertices[v].x ) || !std::isfinite( vertices[v].y ) || !std::isfinite( vertices[v].z ) ) {
			return fail( ctx, PBR_EINVAL, "build_bvh: vertex %u is not finite", v );
		}
	}

	const bool radix = ( ctx->knobs.bvhBuilder == 1 );   // round 1's radix tree; default: locally-ordered clustering

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	const uint32_t capacity = pbr_bvh_node_capacity( num_faces );
	DevBuf dVertices, dFacesV, dFacesN, dKeys, dKeysSorted, dBounds, dNodes, dFacesVOut, dFacesNOut, dTemp;
	HIP_TRY( ctx, dVertices.alloc( sizeof( pbr_float4 ) * num_vertices ) );
	HIP_TRY( ctx, dFacesV.alloc( sizeof( pbr_uint4 ) * num_faces ) );
	HIP_TRY( ctx, dFacesN.alloc( sizeof( pbr_uint4 ) * num_faces ) );
	HIP_TRY( ctx, dKeys.alloc( sizeof( unsigned long long ) * num_faces ) );
	HIP_TRY( ctx, dKeysSorted.alloc( sizeof( unsigned long long ) * num_faces ) );
	HIP_TRY( ctx, dBounds.alloc( sizeof( float ) * 8 ) );
	HIP_TRY( ctx, dNodes.alloc( sizeof( pbr_bvh_node ) * capacity ) );
	HIP_TRY( ctx, dFacesVOut.alloc( sizeof( pbr_uint4 ) * num_faces ) );
	HIP_TRY( ctx, dFacesNOut.alloc( sizeof( pbr_uint4 ) * num_faces ) );

	HIP_TRY( ctx, hipMemcpyAsync( dVertices.p, vertices, sizeof( pbr_float4 ) * num_vertices, hipMemcpyHostToDevice, ctx->stream ) );
	HIP_TRY( ctx, hipMemcpyAsync( dFacesV.p, facesV, sizeof( pbr_uint4 ) * num_faces, hipMemcpyHostToDevice, ctx->stream ) );
	HIP_TRY( ctx, hipMemcpyAsync( dFacesN.p, facesN, sizeof( pbr_uint4 ) * num_faces, hipMemcpyHostToDevice, ctx->stream ) );
	// order-preserving unsigned images of +inf (minima) and -inf (maxima), see ptb::atomicMinFloat
	const unsigned bounds[8] = { 0xFF800000u, 0xFF800000u, 0xFF800000u, 0u, 0x007FFFFFu, 0x007FFFFFu, 0x007FFFFFu, 0u };
	HIP_TRY( ctx, hipMemcpyAsync( dBounds.p, bounds, sizeof( bounds ), hipMemcpyHostToDevice, ctx->stream ) );

	ptb::BuildArrays B;
	std::memset( &B, 0, sizeof( B ) );
	B.vertices = (const pbr_float4*) dVertices.p;
	B.facesV = (const pbr_uint4*) dFacesV.p;
	B.facesN = (const pbr_uint4*) dFacesN.p;
	B.numFaces = num_faces;
	B.keys = (unsigned long long*) dKeys.p;
	B.keysSorted = (unsigned long long*) dKeysSorted.p;
	B.sceneMin = (float*) dBounds.p;
	B.sceneMax = (float*) dBounds.p + 4;
	B.nodesOut = (pbr_bvh_node*) dNodes.p;
	B.facesVOut = (pbr_uint4*) dFacesVOut.p;
	B.facesNOut = (pbr_uint4*) dFacesNOut.p;

	const unsigned threads = 256;
	const unsigned faceBlocks = ( num_faces + threads - 1 ) / threads;

	HIP_TRY( ctx, hipEventRecord( ctx->evStart, ctx->stream ) );
	hipLaunchKernelGGL( ptb::centroidBounds, dim3( faceBlocks ), dim3( threads ), 0, ctx->stream, B );
	hipLaunchKernelGGL( ptb::mortonKeys, dim3( faceBlocks ), dim3( threads ), 0, ctx->stream, B );
	HIP_TRY( ctx, hipGetLastError() );

	size_t tempBytes = 0;
	HIP_TRY( ctx, hipcub::DeviceRadixSort::SortKeys( nullptr, tempBytes, B.keys, B.keysSorted, (int) num_faces, 0, 64, ctx->stream ) );
	HIP_TRY( ctx, dTemp.alloc( tempBytes ) );
	HIP_TRY( ctx, hipcub::DeviceRadixSort::SortKeys( dTemp.p, tempBytes, B.keys, B.keysSorted, (int) num_faces, 0, 64, ctx->stream ) );

	uint32_t treeNodes = 0;
	const int built = radix ? buildRadixTree( ctx, B, num_faces, &treeNodes ) : buildClustered( ctx, B, num_faces, &treeNodes );

	if( built != PBR_OK ) {
		return built;
	}

	HIP_TRY( ctx, hipEventRecord( ctx->evStop, ctx->stream ) );

	// a tree that is a single leaf has no container above it, but the walk starts at node 1 (pt_bvh.cl:84): give it a root
	pbr_bvh_node* firstTreeNode = ( treeNodes == 1 ) ? nodes_out + 1 : nodes_out;
	HIP_TRY( ctx, hipMemcpyAsync( firstTreeNode, dNodes.p, sizeof( pbr_bvh_node ) * treeNodes, hipMemcpyDeviceToHost, ctx->stream ) );
	HIP_TRY( ctx, hipMemcpyAsync( facesV_out, dFacesVOut.p, sizeof( pbr_uint4 ) * num_faces, hipMemcpyDeviceToHost, ctx->stream ) );
	HIP_TRY( ctx, hipMemcpyAsync( facesN_out, dFacesNOut.p, sizeof( pbr_uint4 ) * num_faces, hipMemcpyDeviceToHost, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );

	if( treeNodes == 1 ) {
		nodes_out[0] = nodes_out[1];
		nodes_out[0].bbMin.w = -1.0f;
		nodes_out[0].bbMax.w = -1.0f;
		*num_nodes_out = 2;
	}
	else {
		*num_nodes_out = treeNodes;
	}

	float ms = 0.0f;
	HIP_TRY( ctx, hipEventElapsedTime( &ms, ctx->evStart, ctx->evStop ) );
	ctx->lastKernelMs = (double) ms;
	return PBR_OK;
}

int pbr_diag_math( pbr_ctx* ctx, int op, const float* x, const float* y, int n, float* out ) {
	if( ctx == nullptr || ctx->stream == nullptr || x == nullptr || out == nullptr || n <= 0 ) {
		return fail( ctx, PBR_EINVAL, "diag_math: bad argument" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	DevBuf dx, dy, dout;
	const size_t bytes = sizeof( float ) * (size_t) n;
	HIP_TRY( ctx, dx.alloc( bytes ) );
	HIP_TRY( ctx, dy.alloc( bytes ) );
	HIP_TRY( ctx, dout.alloc( bytes ) );
	HIP_TRY( ctx, hipMemcpy( dx.p, x, bytes, hipMemcpyHostToDevice ) );
	HIP_TRY( ctx, hipMemcpy( dy.p, ( y != nullptr ) ? y : x, bytes, hipMemcpyHostToDevice ) );
	hipLaunchKernelGGL( ptk::diagMath, dim3( (unsigned) ( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, ctx->stream,
		op, (const float*) dx.p, (const float*) dy.p, n, (float*) dout.p );
	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	HIP_TRY( ctx, hipMemcpy( out, dout.p, bytes, hipMemcpyDeviceToHost ) );
	return PBR_OK;
}

int pbr_diag_trace( pbr_ctx* ctx, const float* rays, int n, float* out_t, int32_t* out_face, float* out_normal, uint32_t* out_counts ) {
	if( ctx == nullptr || !ctx->hasScene ) {
		return fail( ctx, PBR_ESTATE, "diag_trace before pbr_upload_scene" );
	}
	if( rays == nullptr || n <= 0 || out_t == nullptr || out_face == nullptr || out_normal == nullptr || out_counts == nullptr ) {
		return fail( ctx, PBR_EINVAL, "diag_trace: bad argument" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	DevBuf dRays, dT, dFace, dNormal, dCounts;
	HIP_TRY( ctx, dRays.alloc( sizeof( float ) * 6 * (size_t) n ) );
	HIP_TRY( ctx, dT.alloc( sizeof( float ) * (size_t) n ) );
	HIP_TRY( ctx, dFace.alloc( sizeof( int ) * (size_t) n ) );
	HIP_TRY( ctx, dNormal.alloc( sizeof( float ) * 3 * (size_t) n ) );
	HIP_TRY( ctx, dCounts.alloc( sizeof( unsigned ) * 2 * (size_t) n ) );
	HIP_TRY( ctx, hipMemcpy( dRays.p, rays, sizeof( float ) * 6 * (size_t) n, hipMemcpyHostToDevice ) );

	int walkStatus = PBR_OK;
	const DevParams P = sceneParams( ctx, &walkStatus );

	if( walkStatus != PBR_OK ) {
		return walkStatus;
	}
	const dim3 grid( (unsigned) ( ( n + 63 ) / 64 ) ), block( 64 );

	if( ctx->numLights > 0 ) {
		hipLaunchKernelGGL( ptk::diagTrace<true>, grid, block, 0, ctx->stream, P, (const float*) dRays.p, n,
			(float*) dT.p, (int*) dFace.p, (float*) dNormal.p, (unsigned*) dCounts.p );
	}
	else {
		hipLaunchKernelGGL( ptk::diagTrace<false>, grid, block, 0, ctx->stream, P, (const float*) dRays.p, n,
			(float*) dT.p, (int*) dFace.p, (float*) dNormal.p, (unsigned*) dCounts.p );
	}

	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	HIP_TRY( ctx, hipMemcpy( out_t, dT.p, sizeof( float ) * (size_t) n, hipMemcpyDeviceToHost ) );
	HIP_TRY( ctx, hipMemcpy( out_face, dFace.p, sizeof( int ) * (size_t) n, hipMemcpyDeviceToHost ) );
	HIP_TRY( ctx, hipMemcpy( out_normal, dNormal.p, sizeof( float ) * 3 * (size_t) n, hipMemcpyDeviceToHost ) );
	HIP_TRY( ctx, hipMemcpy( out_counts, dCounts.p, sizeof( unsigned ) * 2 * (size_t) n, hipMemcpyDeviceToHost ) );
	return PBR_OK;
}

namespace {

int diagPerItem( pbr_ctx* ctx, const float* in, int n, float* out, int inWidth, int outWidth, bool newRay ) {
	if( ctx == nullptr || !ctx->hasScene ) {
		return fail( ctx, PBR_ESTATE, "diag before pbr_upload_scene" );
	}
	if( in == nullptr || out == nullptr || n <= 0 ) {
		return fail( ctx, PBR_EINVAL, "diag: bad argument" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	DevBuf dIn, dOut;
	HIP_TRY( ctx, dIn.alloc( sizeof( float ) * inWidth * (size_t) n ) );
	HIP_TRY( ctx, dOut.alloc( sizeof( float ) * outWidth * (size_t) n ) );
	HIP_TRY( ctx, hipMemcpy( dIn.p, in, sizeof( float ) * inWidth * (size_t) n, hipMemcpyHostToDevice ) );

	int walkStatus = PBR_OK;
	const DevParams P = sceneParams( ctx, &walkStatus );

	if( walkStatus != PBR_OK ) {
		return walkStatus;
	}
	const dim3 grid( (unsigned) ( ( n + 63 ) / 64 ) ), block( 64 );

	if( newRay ) {
		if( ctx->sceneBrdf == 0 ) {
			hipLaunchKernelGGL( ptk::diagNewRay<0>, grid, block, 0, ctx->stream, P, (const float*) dIn.p, n, (float*) dOut.p );
		}
		else {
			hipLaunchKernelGGL( ptk::diagNewRay<1>, grid, block, 0, ctx->stream, P, (const float*) dIn.p, n, (float*) dOut.p );
		}
	}
	else {
		if( ctx->sceneBrdf == 0 ) {
			hipLaunchKernelGGL( ptk::diagBrdf<0>, grid, block, 0, ctx->stream, P, (const float*) dIn.p, n, (float*) dOut.p );
		}
		else {
			hipLaunchKernelGGL( ptk::diagBrdf<1>, grid, block, 0, ctx->stream, P, (const float*) dIn.p, n, (float*) dOut.p );
		}
	}

	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	HIP_TRY( ctx, hipMemcpy( out, dOut.p, sizeof( float ) * outWidth * (size_t) n, hipMemcpyDeviceToHost ) );
	return PBR_OK;
}

}  // namespace

int pbr_diag_brdf( pbr_ctx* ctx, const float* in, int n, float* out ) {
	return diagPerItem( ctx, in, n, out, 16, 4, false );
}

int pbr_diag_new_ray( pbr_ctx* ctx, const float* in, int n, float* out ) {
	return diagPerItem( ctx, in, n, out, 12, 8, true );
}

// Experimental: stream n rays {ox,oy,oz,_, dx,dy,dz,_} through the traversal-only probe `repeats`
// times; returns the best kernel time in *ms_out and the last hits in out (n x {t, face-bits}).
int pbr_diag_trace_stream( pbr_ctx* ctx, int mode, const float* rays8, uint32_t n, int repeats, float* out2, double* ms_out ) {
	if( ctx == nullptr || !ctx->hasScene || rays8 == nullptr || n == 0 || out2 == nullptr || ms_out == nullptr ) {
		return fail( ctx, PBR_EINVAL, "diag_trace_stream: bad argument / no scene" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	DevBuf dRays, dOut;
	HIP_TRY( ctx, dRays.alloc( sizeof( float ) * 8 * (size_t) n ) );
	HIP_TRY( ctx, dOut.alloc( sizeof( float ) * 2 * (size_t) n ) );
	HIP_TRY( ctx, hipMemcpy( dRays.p, rays8, sizeof( float ) * 8 * (size_t) n, hipMemcpyHostToDevice ) );

	uint32_t hotAvail = 0;
	int walkStatus = PBR_OK;
	DevParams P = sceneParams( ctx, &walkStatus, &hotAvail );

	if( walkStatus != PBR_OK ) {
		return walkStatus;
	}
	P.workCounter = ctx->dWork;
	P.counters = ctx->dCounters;
	ctx->workClean = false;   // this probe uses the queue heads its own way
	double best = 1e30;

	for( int r = 0; r < repeats; r++ ) {
		HIP_TRY( ctx, hipMemsetAsync( ctx->dWork, 0, kWorkBytes, ctx->stream ) );
		HIP_TRY( ctx, hipEventRecord( ctx->evStart, ctx->stream ) );

		// mode = number of hot nodes to stage in LDS (0 = none); 8 waves / SIMD => 2048 threads per CU
		const int blocksPerCU = 2048 / PBR_BLOCK;
		size_t slots = std::min<size_t>( (size_t) mode, hotAvail );
		slots = std::min<size_t>( slots, ( 160 * 1024 / (size_t) blocksPerCU - 256 ) / 32 );
		slots &= ( P.walkScheme == 3 ) ? ~(size_t) 1 : ~(size_t) 0;      // compact records are two slots each
		P.numHot = (int) slots;
		P.numHotBytes = (int) slots * 32;
		const dim3 grid( (unsigned) ( ctx->numCUs * blocksPerCU ) ), block( PBR_BLOCK );

		if( slots == 0 ) {
			hipLaunchKernelGGL( ptk::diagTraceStream<false>, grid, block, 0, ctx->stream, P, (const float4*) dRays.p, n, (float2*) dOut.p );
		}
		else {
			HIP_TRY( ctx, hipFuncSetAttribute( (const void*) ptk::diagTraceStream<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ( slots * 32 ) ) );
			hipLaunchKernelGGL( ptk::diagTraceStream<true>, grid, block, slots * 32, ctx->stream, P, (const float4*) dRays.p, n, (float2*) dOut.p );
		}

		HIP_TRY( ctx, hipGetLastError() );
		HIP_TRY( ctx, hipEventRecord( ctx->evStop, ctx->stream ) );
		HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
		float ms = 0.0f;
		HIP_TRY( ctx, hipEventElapsedTime( &ms, ctx->evStart, ctx->evStop ) );
		best = ( ms < best ) ? ms : best;
	}

	HIP_TRY( ctx, hipMemcpy( out2, dOut.p, sizeof( float ) * 2 * (size_t) n, hipMemcpyDeviceToHost ) );
	*ms_out = best;
	return PBR_OK;
}


// Counter calibration: allocate a zero-filled table of table_bytes, read `reads` elements / records
// in the given pattern (0 stream, 1 random 16 B, 2 random 32 B); reports the kernel time.  Run it
// under rocprofv3 --pmc to see what the memory counters say about a KNOWN amount of traffic.
int pbr_diag_calibrate( pbr_ctx* ctx, int mode, uint64_t table_bytes, uint64_t reads, double* ms_out ) {
	if( ctx == nullptr || ctx->stream == nullptr || mode < 0 || mode > 2 || table_bytes < 4096 || reads == 0 || ms_out == nullptr ) {
		return fail( ctx, PBR_EINVAL, "diag_calibrate: bad argument" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	DevBuf table, sink;
	HIP_TRY( ctx, table.alloc( (size_t) table_bytes ) );
	HIP_TRY( ctx, sink.alloc( 64 ) );
	HIP_TRY( ctx, hipMemset( table.p, 0, (size_t) table_bytes ) );
	HIP_TRY( ctx, hipDeviceSynchronize() );

	const unsigned long long count = table_bytes / 16;
	const unsigned blocks = (unsigned) ctx->numCUs * 8;
	const unsigned long long threads = (unsigned long long) blocks * 256;
	const unsigned steps = (unsigned) ( ( reads + threads - 1 ) / threads );

	HIP_TRY( ctx, hipEventRecord( ctx->evStart, ctx->stream ) );

	if( mode == 0 ) {
		hipLaunchKernelGGL( ptk::diagCalibrate<0>, dim3( blocks ), dim3( 256 ), 0, ctx->stream, (const float4*) table.p, count, steps, (float*) sink.p );
	}
	else if( mode == 1 ) {
		hipLaunchKernelGGL( ptk::diagCalibrate<1>, dim3( blocks ), dim3( 256 ), 0, ctx->stream, (const float4*) table.p, count, steps, (float*) sink.p );
	}
	else {
		hipLaunchKernelGGL( ptk::diagCalibrate<2>, dim3( blocks ), dim3( 256 ), 0, ctx->stream, (const float4*) table.p, count, steps, (float*) sink.p );
	}

	HIP_TRY( ctx, hipGetLastError() );
	HIP_TRY( ctx, hipEventRecord( ctx->evStop, ctx->stream ) );
	HIP_TRY( ctx, hipStreamSynchronize( ctx->stream ) );
	float ms = 0.0f;
	HIP_TRY( ctx, hipEventElapsedTime( &ms, ctx->evStart, ctx->evStop ) );
	*ms_out = (double) ms * 1.0;
	return PBR_OK;
}

int pbr_diag_tune_budget( pbr_ctx* ctx, uint32_t* frames ) {
	if( ctx == nullptr || frames == nullptr ) {
		return PBR_EINVAL;
	}
	if( !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "tune budget before pbr_configure" );
	}

	// screening: 7 plans x 2; refinement: up to 3 finalists x 2 x ( 4 + 12 ), twice when the best two are within 5 %; all in
	// 1080p-frame equivalents
	*frames = ( 7u * 2u + 2u * 3u * 2u * ( 4u + 12u ) ) * tuneScaleOf( (size_t) ctx->numLocalTiles * 64 );
	return PBR_OK;
}

int pbr_diag_last_plan( pbr_ctx* ctx, char* name, size_t capacity, int* tuned ) {
	if( ctx == nullptr || name == nullptr || capacity == 0 ) {
		return fail( ctx, PBR_EINVAL, "diag_last_plan: null argument" );
	}

	std::snprintf( name, capacity, "%s", ctx->lastPlan );

	if( tuned != nullptr ) {
		*tuned = ctx->tunedPlan;
	}

	return PBR_OK;
}

int pbr_diag_last_deal( pbr_ctx* ctx, char* name, size_t capacity, int* learnt ) {
	if( ctx == nullptr || name == nullptr || capacity == 0 ) {
		return fail( ctx, PBR_EINVAL, "diag_last_deal: null argument" );
	}

	std::snprintf( name, capacity, "%s", ctx->lastDeal );

	if( learnt != nullptr ) {
		*learnt = ctx->costLearnt ? 1 : 0;
	}

	return PBR_OK;
}

int pbr_diag_bvh_build_info( pbr_ctx* ctx, int* radius ) {
	if( ctx == nullptr || radius == nullptr ) {
		return fail( ctx, PBR_EINVAL, "diag_bvh_build_info: null argument" );
	}

	*radius = ctx->lastBvhRadius;
	return PBR_OK;
}

int pbr_diag_scene_bytes( pbr_ctx* ctx, uint64_t out[3] ) {
	if( ctx == nullptr || out == nullptr || !ctx->hasScene ) {
		return fail( ctx, PBR_ESTATE, "diag_scene_bytes: no scene" );
	}

	out[0] = ctx->nodeBytes;
	out[1] = ctx->walkBytes;
	out[2] = ctx->triBytes;
	return PBR_OK;
}

int pbr_diag_last_kernel( pbr_ctx* ctx, char* name, size_t capacity ) {
	if( ctx == nullptr || name == nullptr || capacity == 0 ) {
		return fail( ctx, PBR_EINVAL, "diag_last_kernel: null argument" );
	}

	std::snprintf( name, capacity, "%s", ctx->lastKernel );
	return PBR_OK;
}

int pbr_diag_launch_fit( pbr_ctx* ctx, double* fixed_ms, double* per_frame_ms ) {
	if( ctx == nullptr || fixed_ms == nullptr || per_frame_ms == nullptr ) {
		return fail( ctx, PBR_EINVAL, "diag_launch_fit: null argument" );
	}

	const int plan = ( ctx->pinnedPlan >= 0 ) ? ctx->pinnedPlan : ctx->tunedPlan;

	for( int k = 0; k < ctx->refineCount; k++ ) {
		const double* f = ctx->refineFit[k];
		const double det = f[0] * f[2] - f[1] * f[1];

		if( ctx->refinePlan[k] != plan || !( det > 1e-9 * f[2] * f[0] ) ) {
			continue;
		}

		const double b = ( f[0] * f[4] - f[1] * f[3] ) / det;
		const double a = ( f[3] - b * f[1] ) / f[0];

		if( a >= 0.0 && b >= 0.0 ) {
			*fixed_ms = a;
			*per_frame_ms = b;
			return PBR_OK;
		}
	}

	return fail( ctx, PBR_ESTATE, "diag_launch_fit: the tuner holds no two-length fit for the plan in use (pinned before it tuned, or launches of one length only)" );
}

int pbr_diag_set_knob( pbr_ctx* ctx, const char* name, int value ) {
	if( ctx == nullptr || name == nullptr ) {
		return fail( ctx, PBR_EINVAL, "diag_set_knob: null argument" );
	}

	Knobs& k = ctx->knobs;
	const struct { const char* name; int* slot; } table[] = {
		{ "lds_slots", &k.ldsSlots }, { "blocks_per_cu", &k.blocksPerCU }, { "ph_park", &k.phPark }, { "ph_shade", &k.phShade },
		{ "park_eighths", &k.parkEighths }, { "drain_mode", &k.drainMode }, { "refill_batch", &k.refillBatch },
		{ "chunk_frames", &k.chunkFrames }, { "face_normals", &k.faceNormals }, { "bvh_builder", &k.bvhBuilder },
		{ "ploc_radius", &k.plocRadius }, { "tune_log", &k.tuneLog }, { "deal_order", &k.dealOrder },
	};

	for( const auto& entry : table ) {
		if( std::strcmp( entry.name, name ) == 0 ) {
			*entry.slot = value;
			// plans carry the knobs' values: rebuild them, and let the tuner start over
			ctx->plansBuilt = false;
			ctx->phongPlanBuilt = false;
			resetTuning( ctx );
			return PBR_OK;
		}
	}

	return fail( ctx, PBR_EINVAL, "diag_set_knob: unknown knob '%s'", name );
}

int pbr_diag_get_tile_order( pbr_ctx* ctx, int which, uint32_t* order, uint32_t capacity, uint32_t* count, uint32_t band_first[PT_BANDS + 1] ) {
	if( ctx == nullptr || !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "diag_get_tile_order before pbr_configure" );
	}
	if( which != 0 && !ctx->costLearnt ) {
		return fail( ctx, PBR_ESTATE, "diag_get_tile_order: no cost order has been learnt yet (it is built after the first render)" );
	}

	const std::vector<unsigned>& table = ( which == 1 ) ? ctx->hCostOrder : ( which == 2 ) ? ctx->hLastOrder : ctx->hTileOrder;
	const uint32_t n = (uint32_t) table.size();

	if( count != nullptr ) {
		*count = n;
	}

	if( band_first != nullptr ) {
		for( int band = 0; band <= PT_BANDS; band++ ) {
			band_first[band] = ( which != 0 ) ? ctx->costBandFirst[band] : ctx->bandFirst[band];
		}
	}

	if( order != nullptr ) {
		if( capacity < n ) {
			return fail( ctx, PBR_EINVAL, "diag_get_tile_order: %u entries do not fit a buffer of %u", n, capacity );
		}

		std::memcpy( order, table.data(), sizeof( uint32_t ) * n );
	}

	return PBR_OK;
}

int pbr_diag_set_tile_order( pbr_ctx* ctx, const uint32_t* order, uint32_t count, const uint32_t* band_first ) {
	if( ctx == nullptr || !ctx->configured ) {
		return fail( ctx, PBR_ESTATE, "diag_set_tile_order before pbr_configure" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	std::vector<unsigned> spatial;
	unsigned spatialFirst[PT_BANDS + 1];
	spatialTileOrder( ctx, &spatial, spatialFirst );

	if( order == nullptr ) {
		ctx->hTileOrder = spatial;
		std::memcpy( ctx->bandFirst, spatialFirst, sizeof( spatialFirst ) );
		ctx->orderPinned = false;
		return uploadTileOrder( ctx );
	}

	if( count != (uint32_t) spatial.size() ) {
		return fail( ctx, PBR_EINVAL, "diag_set_tile_order: %u entries, the queue has %zu tiles", count, spatial.size() );
	}

	const uint32_t* first = ( band_first != nullptr ) ? band_first : spatialFirst;

	if( first[0] != 0u || first[PT_BANDS] != count ) {
		return fail( ctx, PBR_EINVAL, "diag_set_tile_order: the bands' stretches must cover the table: band_first[0] = 0, band_first[8] = %u", count );
	}

	for( int band = 0; band < PT_BANDS; band++ ) {
		if( first[band] > first[band + 1] ) {
			return fail( ctx, PBR_EINVAL, "diag_set_tile_order: band_first must not decrease (band %d)", band );
		}
	}

	// the table must name every local tile exactly once: a unit dealt twice or never is a wrong image.  Without band_first the
	// bands are the spatial ones and every band's stretch must hold that band's own tiles.
	std::vector<unsigned char> bandOf( spatial.size(), 1 );

	if( band_first == nullptr ) {
		for( int band = 0; band < PT_BANDS; band++ ) {
			for( unsigned k = spatialFirst[band]; k < spatialFirst[band + 1]; k++ ) {
				bandOf[spatial[k]] = (unsigned char) ( band + 1 );
			}
		}
	}

	for( int band = 0; band < PT_BANDS; band++ ) {
		for( unsigned k = first[band]; k < first[band + 1]; k++ ) {
			const uint32_t tile = order[k];
			const unsigned char want = ( band_first == nullptr ) ? (unsigned char) ( band + 1 ) : (unsigned char) 1;

			if( tile >= (uint32_t) bandOf.size() || bandOf[tile] != want ) {
				return fail( ctx, PBR_EINVAL, "diag_set_tile_order: entry %u (tile %u) is not a tile of band %d, or is named twice", k, tile, band );
			}

			bandOf[tile] = 0;
		}
	}

	ctx->hTileOrder.assign( order, order + count );
	std::memcpy( ctx->bandFirst, first, sizeof( spatialFirst ) );
	ctx->orderPinned = true;
	return uploadTileOrder( ctx );
}

int pbr_diag_pin_plan( pbr_ctx* ctx, int plan ) {
	if( ctx == nullptr || plan < -1 || plan > 6 ) {
		return fail( ctx, PBR_EINVAL, "diag_pin_plan: plan must be -1 (auto-tune) or 0..6" );
	}

	ctx->pinnedPlan = plan;
	return PBR_OK;
}

int pbr_diag_last_trace( pbr_ctx* ctx, double* trace_ms, uint32_t* launches ) {
	if( ctx == nullptr || trace_ms == nullptr || launches == nullptr ) {
		return fail( ctx, PBR_EINVAL, "diag_last_trace: null argument" );
	}

	*trace_ms = ctx->lastTraceMs;
	*launches = ctx->lastTraceLaunches;
	return PBR_OK;
}

int pbr_diag_raw_counters( pbr_ctx* ctx, uint64_t out[16] ) {
	if( ctx == nullptr || out == nullptr ) {
		return fail( ctx, PBR_EINVAL, "diag_raw_counters: null argument" );
	}

	HIP_TRY( ctx, hipSetDevice( ctx->device ) );
	HIP_TRY( ctx, hipMemcpy( out, ctx->dCounters, sizeof( uint64_t ) * kCounterSlots, hipMemcpyDeviceToHost ) );
	return PBR_OK;
}

int pbr_diag_guard_trips( pbr_ctx* ctx, uint32_t out[3] ) {
	if( ctx == nullptr || ctx->stream == nullptr || out == nullptr ) {
		return fail( ctx, PBR_EINVAL, "diag_guard_trips: bad argument" );
	}

	const volatile unsigned* host = ctx->dGuard;
	out[0] = host[0];
	out[1] = host[1];
	out[2] = host[2];
	return PBR_OK;
}

}  // extern "C"
