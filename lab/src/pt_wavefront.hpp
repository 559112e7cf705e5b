// Wavefront schedule of the same path: traversal and shading as two persistent kernels that
// exchange rays through queues in HBM.
//
// Why: in the megakernel (pt_kernel.hpp) the shading code dictates the register allocation
// (143-160 VGPRs, or 64 with scratch spills), and a wave's lanes traverse in lock step —
// profiled on MI355X, only ~17-28 % of the issued lane slots of the node loop do useful work and
// throughput scales almost linearly with resident waves.  Splitting the path gives the traversal
// kernel ~48 VGPRs (8 waves / SIMD, no spills) and lets every LANE take a new ray from the queue
// as soon as its own ray has left the tree (ballot-batched, wave-level pooled fetch), while the
// shading kernel runs one path per lane with all lanes busy.
//
// Per pixel the arithmetic — node visits, face tests, random draws, the running mean — is the
// megakernel's, function for function (shadeStep, beginPixel, finishPixel, boxHit, testLeaf), so
// the image stays bit-identical; only WHERE a lane's state lives between two bounces changes
// (128 B per pixel in HBM instead of registers).
//
//   wfInit      every pixel: beginPixel (frame 0, first camera ray) -> state, queue[0] = all slots
//   repeat      wfTrace   queue[c] -> closest hit per ray, written into the ray's state
//               wfShade   queue[c] -> shadeStep; pixels with work left append themselves to queue[c^1]
//   wfReduce    sum the per-pixel counters
#pragma once

#include "pt_kernel.hpp"

namespace ptk {

// Per-pixel state, SoA in 16-byte chunks: chunk k of pixel s at state[k * stride + s].
//   0 ray.origin.xyz, seed        1 ray.dir.xyz, focus       2 color.xyz, accW
//   3 acc.xyz, secondaryPaths     4 finalColor.xyz, -         5 frame, sample, depth, depthAdded
//   6 dbgNodes, dbgTris, hit.face, hit.t                      7 totals: nodes, tris, hits, paths
enum { WF_CHUNKS = 8 };

struct WfParams {
	float4* state;
	unsigned stride;        // pixels of this rank
	unsigned* queue[2];     // slots with a ray to trace
	unsigned* count;        // [2] entries in queue[i]
	unsigned* head;         // [2] fetch cursor of wfTrace into queue[i]
	int cur;                // which queue this pass reads
};

PT_DEV float4 packInts( int a, int b, int c, int d ) {
	return make_float4( __int_as_float( a ), __int_as_float( b ), __int_as_float( c ), __int_as_float( d ) );
}

PT_DEV void storeState( const WfParams& W, unsigned s, const PixelState& st, const LaneCounters& cnt ) {
	float4* p = W.state + s;
	const size_t n = W.stride;
	p[0 * n] = make_float4( st.ray.origin.x, st.ray.origin.y, st.ray.origin.z, st.seed );
	p[1 * n] = make_float4( st.ray.dir.x, st.ray.dir.y, st.ray.dir.z, st.focus );
	p[2 * n] = make_float4( st.color.x, st.color.y, st.color.z, st.accW );
	p[3 * n] = make_float4( st.acc.x, st.acc.y, st.acc.z, __uint_as_float( st.secondaryPaths ) );
	p[4 * n] = make_float4( st.finalColor.x, st.finalColor.y, st.finalColor.z, 0.0f );
	p[5 * n] = packInts( st.frame, st.sample, st.depth, st.depthAdded );
	// chunk 6: the debug counters; hit.face / hit.t are filled in by wfTrace
	p[6 * n] = packInts( (int) st.dbgNodes, (int) st.dbgTris, 0, 0 );
	p[7 * n] = packInts( (int) cnt.nodes, (int) cnt.tris, (int) cnt.hits, (int) cnt.paths );
}

PT_DEV void loadState( const DevParams& P, const WfParams& W, unsigned s, PixelState& st, LaneCounters& cnt, Hit& hit ) {
	const float4* p = W.state + s;
	const size_t n = W.stride;
	const float4 c0 = p[0 * n], c1 = p[1 * n], c2 = p[2 * n], c3 = p[3 * n];
	const float4 c4 = p[4 * n], c5 = p[5 * n], c6 = p[6 * n], c7 = p[7 * n];

	st.slot = s;

	st.ray.origin = mk3( c0.x, c0.y, c0.z );
	st.seed = c0.w;
	st.ray.dir = mk3( c1.x, c1.y, c1.z );
	st.focus = c1.w;
	st.color = mk3( c2.x, c2.y, c2.z );
	st.accW = c2.w;
	st.acc = mk3( c3.x, c3.y, c3.z );
	st.secondaryPaths = __float_as_uint( c3.w );
	st.finalColor = mk3( c4.x, c4.y, c4.z );
	st.frame = __float_as_int( c5.x );
	st.sample = __float_as_int( c5.y );
	st.depth = __float_as_int( c5.z );
	st.depthAdded = __float_as_int( c5.w );
	st.dbgNodes = __float_as_uint( c6.x );
	st.dbgTris = __float_as_uint( c6.y );
	hit.face = __float_as_int( c6.z );
	hit.t = c6.w;
	cnt.nodes = __float_as_uint( c7.x );
	cnt.tris = __float_as_uint( c7.y );
	cnt.hits = __float_as_uint( c7.z );
	cnt.paths = __float_as_uint( c7.w );
}

// ---- wfInit ---------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void wfInit( const DevParams P, const WfParams W ) {
	const unsigned total = (unsigned) P.numLocalTiles * 64u;
	const unsigned step = gridDim.x * blockDim.x;

	for( unsigned s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += step ) {
		PixelState st;
		LaneCounters cnt;
		cnt.nodes = cnt.tris = cnt.hits = cnt.paths = 0;
		beginPixel( P, st, s, cnt );
		storeState( W, s, st, cnt );
		W.queue[0][s] = s;
	}

	if( blockIdx.x == 0 && threadIdx.x == 0 ) {
		W.count[0] = total;
		W.count[1] = 0;
		W.head[0] = 0;
		W.head[1] = 0;
	}
}

// ---- wfTrace --------------------------------------------------------------------------------
// Lane states: NODE / LEAF as in pathTracingBatched; FETCH = the lane's ray is finished (its hit
// is stored) and it wants the next queue entry; DONE = queue exhausted.
#ifndef PBR_WF_FETCH_BATCH
#define PBR_WF_FETCH_BATCH 16
#endif
#ifndef PBR_WF_LEAF_BATCH
#define PBR_WF_LEAF_BATCH 16
#endif
#define PBR_WF_CHUNK 256u      // queue entries a wave reserves with one global atomic

enum { WF_NODE = 0, WF_LEAF = 1, WF_FETCH = 2, WF_DONE = 3 };

template<bool LIGHTS>
__global__ __launch_bounds__( PBR_BLOCK, 8 ) void wfTrace( const DevParams P, const WfParams W ) {
	__shared__ unsigned sPool[PBR_BLOCK / 64][8];   // per wave: {next, end} of its reservation + three words of hand-over

	const unsigned n = W.count[W.cur];

	if( blockIdx.x == 0 && threadIdx.x == 0 ) {
		// the other queue is filled by the wfShade that follows, the other cursor used by the next wfTrace
		W.count[W.cur ^ 1] = 0;
		W.head[W.cur ^ 1] = 0;
	}

	// late passes carry few rays: blocks that would find the queue empty leave before staging LDS
	if( (size_t) blockIdx.x * PBR_BLOCK >= (size_t) n ) {
		return;
	}

	const float4* lds = gHotNodes;
	stageHotNodes( P, gHotNodes );

	const int lane = (int) ( threadIdx.x & 63u );
	const int wave = (int) ( threadIdx.x >> 6 );
	const unsigned* queue = W.queue[W.cur];
	const int numNodes = P.numNodes;
	const size_t stride = W.stride;

	if( lane == 0 ) {
		sPool[wave][0] = 0;
		sPool[wave][1] = 0;
	}

	__syncthreads();

	Ray ray;
	ray.origin = mk3( 0.0f, 0.0f, 0.0f );
	ray.dir = mk3( 0.0f, 0.0f, 1.0f );
	WalkState w;
	w.invDir = mk3( 1.0f, 1.0f, 1.0f );
	w.cur.ref = -1;
	w.hit.t = inff();
	w.hit.face = 0;
	w.leafFace0 = -1;
	w.leafFace1 = -1;
	w.leafTNear = 0.0f;
	unsigned slot = 0, nodes = 0, tris = 0;
	int mode = ( n > 0 ) ? WF_FETCH : WF_DONE;

	while( mode != WF_DONE ) {
		// ---- one node ---------------------------------------------------------------------
		if( mode == WF_NODE ) {
			nodes++;

			float4 lo, hi;
			fetchNode<true>( P, lds, w.cur, &lo, &hi );   // lo = n0 {min.xy, max.xy}, hi = n1 {min.z, max.z, w0, w1}
			const NodeLinks node = decodeNode( hi );
			float tNear;

			if( boxHit<false>( lo, hi, ray, w.invDir, w.hit.t, &tNear ) ) {
				w.cur = node.onHit;

				if( node.leaf ) {
					w.leafFace0 = node.face0;
					w.leafFace1 = node.face1;
					w.leafTNear = tNear;
					mode = WF_LEAF;
				}
			}
			else {
				w.cur = node.onMiss;
			}

			if( mode == WF_NODE && !alive( w.cur ) ) {
				// the ray has left the tree: publish its hit (pathtracing.cl:259 returns here)
				W.state[6 * stride + slot] = make_float4( __uint_as_float( nodes ), __uint_as_float( tris ), __int_as_float( w.hit.face ), w.hit.t );
				mode = WF_FETCH;
			}
		}

		// ---- deferred triangle tests ----------------------------------------------------------
		{
			const int nLeaf = __popcll( __ballot( mode == WF_LEAF ) );
			const int nNode = __popcll( __ballot( mode == WF_NODE ) );

			if( mode == WF_LEAF && ( nLeaf >= PBR_WF_LEAF_BATCH || nNode == 0 ) ) {
				testLeaf( P, w.leafFace0, w.leafFace1, ray, w.leafTNear, 0.0f, w.hit, tris );

				if( alive( w.cur ) ) {
					mode = WF_NODE;
				}
				else {
					W.state[6 * stride + slot] = make_float4( __uint_as_float( nodes ), __uint_as_float( tris ), __int_as_float( w.hit.face ), w.hit.t );
					mode = WF_FETCH;
				}
			}
		}

		// ---- deferred, pooled ray fetch -----------------------------------------------------------
		{
			const unsigned long long wants = __ballot( mode == WF_FETCH );
			const int nFetch = __popcll( wants );
			const int nWalking = __popcll( __ballot( mode == WF_NODE || mode == WF_LEAF ) );

			if( mode == WF_FETCH && ( nFetch >= PBR_WF_FETCH_BATCH || nWalking == 0 ) ) {
				// rank of this lane among the fetching lanes; the first of them reserves for all
				const int rank = __popcll( wants & ( ( 1ull << lane ) - 1ull ) );

				if( rank == 0 ) {
					const unsigned next = sPool[wave][0];
					const unsigned avail = sPool[wave][1] - next;
					const unsigned fromOld = ( avail < (unsigned) nFetch ) ? avail : (unsigned) nFetch;
					unsigned fresh = 0;

					if( fromOld < (unsigned) nFetch ) {
						// the old reservation runs out: its rest goes to the lowest ranks, a new one serves the others
						fresh = atomicAdd( &W.head[W.cur], PBR_WF_CHUNK );
						sPool[wave][0] = fresh + ( (unsigned) nFetch - fromOld );
						sPool[wave][1] = fresh + PBR_WF_CHUNK;
					}
					else {
						sPool[wave][0] = next + (unsigned) nFetch;
					}

					sPool[wave][2] = next;
					sPool[wave][3] = fromOld;
					sPool[wave][4] = fresh;
				}

				// One wave: its DS operations are issued and completed in order, so the leader's stores
				// are visible to the loads below; the fences keep the compiler from reordering them.
				__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );

				const volatile unsigned* hand = sPool[wave];
				const unsigned oldNext = hand[2];
				const unsigned fromOld = hand[3];
				const unsigned fresh = hand[4];
				const unsigned entry = ( (unsigned) rank < fromOld ) ? oldNext + (unsigned) rank : fresh + ( (unsigned) rank - fromOld );

				if( entry < n ) {
					slot = queue[entry];
					const float4 c0 = W.state[0 * stride + slot];
					const float4 c1 = W.state[1 * stride + slot];
					const float4 c6 = W.state[6 * stride + slot];
					ray.origin = mk3( c0.x, c0.y, c0.z );
					ray.dir = mk3( c1.x, c1.y, c1.z );
					nodes = __float_as_uint( c6.x );
					tris = __float_as_uint( c6.y );
					mode = startWalk<LIGHTS>( P, ray, w );
				}
				else {
					mode = WF_DONE;
				}
			}
		}
	}
}

// ---- wfShade --------------------------------------------------------------------------------
template<int BRDF, bool SHADOW, bool LIGHTS>
__global__ __launch_bounds__( 256 ) void wfShade( const DevParams P, const WfParams W ) {
	const unsigned n = W.count[W.cur];
	const unsigned* queue = W.queue[W.cur];
	unsigned* next = W.queue[W.cur ^ 1];
	const unsigned step = gridDim.x * blockDim.x;

	for( unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += step ) {
		const unsigned s = queue[j];
		PixelState st;
		LaneCounters cnt;
		Hit hit;
		loadState( P, W, s, st, cnt, hit );

		const bool pixelDone = shadeStep<BRDF, SHADOW, LIGHTS>( P, gHotNodes, st, cnt, hit );
		storeState( W, s, st, cnt );

		if( pixelDone ) {
			finishPixel( P, st );
		}
		else {
			next[atomicAdd( &W.count[W.cur ^ 1], 1u )] = s;
		}
	}
}

// ---- wfReduce -------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void wfReduce( const DevParams P, const WfParams W ) {
	const unsigned total = (unsigned) P.numLocalTiles * 64u;
	const unsigned step = gridDim.x * blockDim.x;
	unsigned long long nodes = 0, tris = 0, hits = 0, paths = 0;

	for( unsigned s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += step ) {
		const float4 c7 = W.state[7 * (size_t) W.stride + s];
		nodes += __float_as_uint( c7.x );
		tris += __float_as_uint( c7.y );
		hits += __float_as_uint( c7.z );
		paths += __float_as_uint( c7.w );
	}

	atomicAdd( &P.counters[0], nodes );
	atomicAdd( &P.counters[1], tris );
	atomicAdd( &P.counters[2], hits );
	atomicAdd( &P.counters[3], paths );
}

}  // namespace ptk
