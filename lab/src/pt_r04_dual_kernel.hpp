// Round-4 lab variants (NOT part of the product build: included by csrc/pt_kernel.hpp only under -DPBR_LAB, scripts/lab.sh).
// pathTracingDual: the lane state machine with two paths per lane.
// Measured in profiles/r04/experiments/; DESIGN.md section 5.1e says what each was for and why the product does not use it.

// ---------------------------------------------------------------------------------------
// Phased schedule with TWO paths per lane (round 4, lab: VERDICT r03 item 2)
// ---------------------------------------------------------------------------------------
// The lane state machine above, with two path slots A and B per lane: 128 paths per wave at 4 waves / SIMD.
//   node phase   nodePhaseDual: both slots' fetches are issued before one wait, then both slab tests — twice the
//                requests in flight per wave without a second wave's registers
//   leaf phase   POOLED: a lane tests the leaf of whichever of its slots stands on one (A first; the other in the next
//                round) — one pass of the face tests serves lanes of both slots
//   shade phase  POOLED the same way: a lane shades whichever slot waits for shading; the threshold counts lanes
// In registers per slot: ray, 1 / direction, cursor, closest hit, the parked leaf, mode, the walk's counters (17).  Everything
// else a path carries between bounces (PixelState's cold half: 16 dwords) lives in LDS, in four lane-linear 16-byte planes
// per slot behind the staged tree top (2 x 64 B x 1024 lanes = 128 KiB), and is in registers only while its slot is shaded.
// Per path the sequence of visits, face tests and random draws is the reference's: same image, same debug image, same counters.
#if defined( PBR_LAB ) && defined( PT_NODE_PHASE_ASM )
namespace r04lab {      // the product's pathTracingDual (csrc/pt_dual.hpp) is the pipelined form of this one; the variants stay here
struct DualSlot {
	Ray ray;
	f3 invDir;
	int cur;            // cursor (byte offset of the next record); < 0: the walk has ended
	float t;            // closest hit so far
	int face;
	int leafWord;       // != 0: parked on this hit leaf (MODE_LEAF)
	float leafTNear;
	int mode;
	unsigned nodes, tris;   // of the current walk(s) since the slot was last shaded: added to the path's counters there
};

#ifdef PBR_DUAL_COLD_GLOBAL
// The paths' cold state in memory instead of LDS (lane-linear 16-byte planes per block: the same layout, coalesced) — 64 B read
// + 64 B written per lane and shading; what 5 waves per SIMD (five 256-thread blocks per CU, 32 KB of LDS each) would need.
#ifdef PBR_LAB_LEAN_BLOCK
#define PBR_DUAL_BLOCK PBR_LAB_LEAN_BLOCK
#else
#define PBR_DUAL_BLOCK PBR_BLOCK
#endif
#define PBR_DUAL_COLD_BYTES ( 1536 * 2 * 4 * 256 * 16 )
__device__ float4 gColdState[PBR_DUAL_COLD_BYTES / 16];

PT_DEV float4* labColdPlane( const DevParams& P, int slot, int plane ) {
	(void) P;
	return gColdState + ( ( (int) blockIdx.x * 2 + slot ) * 4 + plane ) * PBR_DUAL_BLOCK + (int) threadIdx.x;
}
#else
PT_DEV float4* labColdPlane( const DevParams& P, int slot, int plane ) {
	return (float4*) ( (char*) gHotNodes + P.slotBase ) + ( slot * 4 + plane ) * PBR_BLOCK + (int) threadIdx.x;
}
#endif

PT_DEV void labLoadCold( const DevParams& P, int slot, PixelState& st ) {
	const float4 a = *labColdPlane( P, slot, 0 ), b = *labColdPlane( P, slot, 1 ), c = *labColdPlane( P, slot, 2 ), d = *labColdPlane( P, slot, 3 );
	st.slot = __float_as_uint( a.x ); st.frame = __float_as_int( a.y ); st.sample = __float_as_int( a.z ); st.finalColor.x = a.w;
	st.finalColor.y = b.x; st.finalColor.z = b.y; st.secondaryPaths = __float_as_uint( b.z ); st.focus = b.w;
	st.seed = c.x; st.dbgNodes = __float_as_uint( c.y ); st.dbgTris = __float_as_uint( c.z ); st.color.x = c.w;
	st.color.y = d.x; st.color.z = d.y; st.depth = __float_as_int( d.z ); st.depthAdded = __float_as_int( d.w );
}

PT_DEV void labStoreCold( const DevParams& P, int slot, const PixelState& st ) {
	*labColdPlane( P, slot, 0 ) = make_float4( __uint_as_float( st.slot ), __int_as_float( st.frame ), __int_as_float( st.sample ), st.finalColor.x );
	*labColdPlane( P, slot, 1 ) = make_float4( st.finalColor.y, st.finalColor.z, __uint_as_float( st.secondaryPaths ), st.focus );
	*labColdPlane( P, slot, 2 ) = make_float4( st.seed, __uint_as_float( st.dbgNodes ), __uint_as_float( st.dbgTris ), st.color.x );
	*labColdPlane( P, slot, 3 ) = make_float4( st.color.y, st.color.z, __int_as_float( st.depth ), __int_as_float( st.depthAdded ) );
}

template<bool LIGHTS>
PT_DEV void labStartWalkDual( const DevParams& P, DualSlot& s ) {
	s.invDir = mk3( 1.0f / s.ray.dir.x, 1.0f / s.ray.dir.y, 1.0f / s.ray.dir.z );
	s.cur = P.firstRef;
	Hit h;
	h.t = inff();
	h.face = 0;

	if( LIGHTS ) {
		traverseLights( P, s.ray, h );
	}

	s.t = h.t;
	s.face = h.face;
	s.leafWord = 0;
	s.leafTNear = 0.0f;
	s.mode = MODE_NODE;
}

// PBR_DUAL_SWAP: ONE walk per lane at a time (the product's nodePhaseAsm), the lane's other path as its reserve — a lane
// whose path A waits (for shading, or has none) and whose path B can walk exchanges the two before the node phase, so a lane
// sits a node phase out only when neither of its paths can walk.
PT_DEV void swapWord( float& a, float& b ) { asm volatile( "v_swap_b32 %0, %1" : "+v"( a ), "+v"( b ) ); }
PT_DEV void swapWord( int& a, int& b ) { asm volatile( "v_swap_b32 %0, %1" : "+v"( a ), "+v"( b ) ); }
PT_DEV void swapWord( unsigned& a, unsigned& b ) { asm volatile( "v_swap_b32 %0, %1" : "+v"( a ), "+v"( b ) ); }
PT_DEV void labSwapSlots( DualSlot& a, DualSlot& b ) {
	swapWord( a.ray.origin.x, b.ray.origin.x ); swapWord( a.ray.origin.y, b.ray.origin.y ); swapWord( a.ray.origin.z, b.ray.origin.z );
	swapWord( a.ray.dir.x, b.ray.dir.x ); swapWord( a.ray.dir.y, b.ray.dir.y ); swapWord( a.ray.dir.z, b.ray.dir.z );
	swapWord( a.invDir.x, b.invDir.x ); swapWord( a.invDir.y, b.invDir.y ); swapWord( a.invDir.z, b.invDir.z );
	swapWord( a.cur, b.cur ); swapWord( a.t, b.t ); swapWord( a.face, b.face ); swapWord( a.leafWord, b.leafWord );
	swapWord( a.leafTNear, b.leafTNear ); swapWord( a.mode, b.mode ); swapWord( a.nodes, b.nodes ); swapWord( a.tris, b.tris );
}

template<int BRDF, bool SHADOW, bool LIGHTS>
#ifdef PBR_DUAL_WAVES5
__global__ __launch_bounds__( PBR_LAB_LEAN_BLOCK ) __attribute__(( amdgpu_waves_per_eu( 5, 5 ) )) void pathTracingDual(
#else
__global__ __launch_bounds__( PBR_BLOCK, 4 ) void pathTracingDual(
#endif
 const DevParams P ) {
#ifdef PBR_DUAL_COLD_GLOBAL
	if( ( (size_t) blockIdx.x + 1 ) * 2 * 4 * PBR_DUAL_BLOCK * 16 > (size_t) PBR_DUAL_COLD_BYTES || blockDim.x != PBR_DUAL_BLOCK ) {
		__builtin_trap();
	}
#endif
	const float4* lds = gHotNodes;
	stageHotNodes( P, gHotNodes );
	LaneCounters cnt;
	cnt.nodes = cnt.tris = cnt.hits = cnt.paths = 0;
	DualSlot A, B;
	A.mode = B.mode = MODE_DONE;
	A.cur = B.cur = -1;
	A.leafWord = B.leafWord = 0;
	A.nodes = A.tris = B.nodes = B.tris = 0;
	A.t = B.t = 0.0f;
	A.face = B.face = 0;
	A.leafTNear = B.leafTNear = 0.0f;
	A.ray.origin = A.ray.dir = A.invDir = B.ray.origin = B.ray.dir = B.invDir = mk3( 0.0f, 0.0f, 0.0f );
	WorkCursor work = beginWork();
	int flip = 0;       // PBR_DUAL_SWAP: path A's cold state is in LDS slot `flip`, path B's in the other

	// both slots take their first unit
	for( int k = 0; k < 2; k++ ) {
		unsigned frame = 0;
		const unsigned slot = nextSlot( P, work, (unsigned) P.nFrames, frame );

		if( slot != PT_NO_WORK ) {
			PixelState st;
			beginPixel( P, st, slot, cnt, frame );
			DualSlot& S = ( k == 0 ) ? A : B;
			S.ray = st.ray;
			labStartWalkDual<LIGHTS>( P, S );
			labStoreCold( P, k, st );
		}
	}

	while( __ballot( A.mode != MODE_DONE || B.mode != MODE_DONE ) != 0ull ) {
#ifdef PBR_DUAL_SWAP
		// ---- node phase: the lane's walkable path, A before B ------------------------------------------
		if( A.mode != MODE_NODE && B.mode == MODE_NODE ) {
			labSwapSlots( A, B );
			flip ^= 1;
		}

		if( A.mode == MODE_NODE ) {
			const int keep = __popcll( __ballot( 1 ) ) - P.phPark;
			const f2v oxyA = { A.ray.origin.x, A.ray.origin.y }, ozzA = { A.ray.origin.z, A.ray.origin.z }, ixyA = { A.invDir.x, A.invDir.y }, izzA = { A.invDir.z, A.invDir.z };
			int leafWordA = 0, parkedFlag;
			float unusedTFar;
			unsigned visits = 0;
			__builtin_amdgcn_s_setprio( PT_WALK_PRIO );
			nodePhaseAsm<false>( P, oxyA, ozzA, ixyA, izzA, A.t, ( keep < 0 ) ? 0 : keep, A.cur, visits, leafWordA, A.leafTNear, unusedTFar, parkedFlag );
			__builtin_amdgcn_s_setprio( 0 );
			A.nodes += visits;
			A.leafWord = ( parkedFlag != 0 ) ? leafWordA : 0;
			A.mode = ( parkedFlag != 0 ) ? MODE_LEAF : ( ( A.cur < 0 ) ? MODE_SHADE : MODE_NODE );
		}
#else
		// ---- node phase: both slots of every lane ----------------------------------------------------
		if( A.mode == MODE_NODE || B.mode == MODE_NODE ) {
			const int walking = __popcll( __ballot( A.mode == MODE_NODE ) ) + __popcll( __ballot( B.mode == MODE_NODE ) );
			const int keep = walking - P.phPark;
			const f2v oxyA = { A.ray.origin.x, A.ray.origin.y }, ozzA = { A.ray.origin.z, A.ray.origin.z }, ixyA = { A.invDir.x, A.invDir.y }, izzA = { A.invDir.z, A.invDir.z };
			const f2v oxyB = { B.ray.origin.x, B.ray.origin.y }, ozzB = { B.ray.origin.z, B.ray.origin.z }, ixyB = { B.invDir.x, B.invDir.y }, izzB = { B.invDir.z, B.invDir.z };
			int refA = ( A.mode == MODE_NODE ) ? A.cur : -1;      // a slot that is not walking sits the phase out
			int refB = ( B.mode == MODE_NODE ) ? B.cur : -1;
			int leafWordA = 0, leafWordB = 0;
			float tNearA = 0.0f, tNearB = 0.0f;
			__builtin_amdgcn_s_setprio( PT_WALK_PRIO );
#ifdef PBR_DUAL_PIPE
			nodePhaseDualPipe( P, oxyA, ozzA, ixyA, izzA, A.t, oxyB, ozzB, ixyB, izzB, B.t, ( keep < 0 ) ? 0 : keep, refA, refB, A.nodes, B.nodes,
			                   leafWordA, tNearA, leafWordB, tNearB );
#else
			nodePhaseDual( P, oxyA, ozzA, ixyA, izzA, A.t, oxyB, ozzB, ixyB, izzB, B.t, ( keep < 0 ) ? 0 : keep, refA, refB, A.nodes, B.nodes,
			               leafWordA, tNearA, leafWordB, tNearB );
#endif
			__builtin_amdgcn_s_setprio( 0 );

			if( A.mode == MODE_NODE ) {
				A.cur = refA;
				A.leafWord = leafWordA;
				A.leafTNear = tNearA;
				A.mode = ( leafWordA != 0 ) ? MODE_LEAF : ( ( refA < 0 ) ? MODE_SHADE : MODE_NODE );
			}
			if( B.mode == MODE_NODE ) {
				B.cur = refB;
				B.leafWord = leafWordB;
				B.leafTNear = tNearB;
				B.mode = ( leafWordB != 0 ) ? MODE_LEAF : ( ( refB < 0 ) ? MODE_SHADE : MODE_NODE );
			}
		}

#endif

		// ---- leaf phase, pooled: the slot that stands on a leaf (A first) ---------------------------------
		// (PBR_DUAL_LEAF2 = n: a second pass at once when n or more lanes still have a slot on a leaf — both were parked)
#ifndef PBR_DUAL_LEAF2
#define PBR_DUAL_LEAF2 0
#endif
		for( int pass = 0; pass < 2; pass++ ) {
			const bool leafA = ( A.mode == MODE_LEAF ), leafB = ( B.mode == MODE_LEAF );

			if( pass == 1 && ( PBR_DUAL_LEAF2 == 0 || __popcll( __ballot( leafA || leafB ) ) < PBR_DUAL_LEAF2 ) ) {
				break;
			}

			if( leafA || leafB ) {
				const bool useB = !leafA;
				Ray ray;
				ray.origin = useB ? B.ray.origin : A.ray.origin;
				ray.dir = useB ? B.ray.dir : A.ray.dir;
				Hit hit;
				hit.t = useB ? B.t : A.t;
				hit.face = useB ? B.face : A.face;
				const int leafWord = useB ? B.leafWord : A.leafWord;
				const float tNear = useB ? B.leafTNear : A.leafTNear;
				unsigned tests = 0;
				__builtin_amdgcn_s_setprio( PT_WALK_PRIO );
				testLeaf<false, true>( P, leafFace0( leafWord ), leafFace1( leafWord ), ray, tNear, 0.0f, hit, tests );
				__builtin_amdgcn_s_setprio( 0 );

				if( useB ) {
					B.t = hit.t; B.face = hit.face; B.tris += tests; B.leafWord = 0;
					B.mode = ( B.cur < 0 ) ? MODE_SHADE : MODE_NODE;
				}
				else {
					A.t = hit.t; A.face = hit.face; A.tris += tests; A.leafWord = 0;
					A.mode = ( A.cur < 0 ) ? MODE_SHADE : MODE_NODE;
				}
			}
		}

		// ---- shade phase, pooled: the slot that waits for shading (A first); the threshold counts lanes ------
		{
			const bool shadeA = ( A.mode == MODE_SHADE ), shadeB = ( B.mode == MODE_SHADE );
			const int nShade = __popcll( __ballot( shadeA || shadeB ) );
			const bool busy = ( __ballot( A.mode == MODE_NODE || A.mode == MODE_LEAF || B.mode == MODE_NODE || B.mode == MODE_LEAF ) != 0ull );

			if( ( shadeA || shadeB ) && ( nShade >= P.phShade || !busy ) ) {
				const bool useB = !shadeA;
				const int which = useB ? ( flip ^ 1 ) : flip;
				PixelState st;
				labLoadCold( P, which, st );
				st.ray.origin = useB ? B.ray.origin : A.ray.origin;
				st.ray.dir = useB ? B.ray.dir : A.ray.dir;
				st.dbgNodes += useB ? B.nodes : A.nodes;
				st.dbgTris += useB ? B.tris : A.tris;
				Hit hit;
				hit.t = useB ? B.t : A.t;
				hit.face = useB ? B.face : A.face;
				hit.normal = mk3( 0.0f, 0.0f, 0.0f );
				bool more = true;

				if( shadeStep<BRDF, SHADOW, LIGHTS, false, true, true>( P, lds, st, cnt, hit ) ) {
					finishPixel( P, st );

					if( cnt.nodes > 0x40000000u || cnt.tris > 0x40000000u ) {
						flushCounters( P, cnt );
					}

					unsigned frame = 0;
					const unsigned slot = nextSlot( P, work, (unsigned) P.nFrames, frame );

					if( slot != PT_NO_WORK ) {
						beginPixel( P, st, slot, cnt, frame );
					}
					else {
						more = false;
					}
				}

				labStoreCold( P, which, st );

				if( useB ) {
					B.nodes = 0; B.tris = 0; B.ray = st.ray;
					if( more ) { labStartWalkDual<LIGHTS>( P, B ); } else { B.mode = MODE_DONE; B.cur = -1; }
				}
				else {
					A.nodes = 0; A.tris = 0; A.ray = st.ray;
					if( more ) { labStartWalkDual<LIGHTS>( P, A ); } else { A.mode = MODE_DONE; A.cur = -1; }
				}
			}
		}
	}

	flushCounters( P, cnt );
}
}   // namespace r04lab
#endif
