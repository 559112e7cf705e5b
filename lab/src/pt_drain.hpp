// The drain kernel — BUILT, BIT-IDENTICAL, SLOWER: kept for the record, not compiled by anything in this tree.
//
// Round 3's attempt at the end of a launch (DESIGN.md "How a launch ends"): once the queue is dry, a wave with at most
// `drainThreshold` paths left wrote them to a pool in global memory (dumpPaths) and ended; a second kernel (pathDrain),
// launched behind the path-tracing launch, finished them with a GROUP of 4 lanes per path that fetched and slab-tested the
// next 4 records of the node stream at once and followed the sequential walk through as many of them as it could.
// It passed the whole parity suite at every threshold (tests at 47eaec9..: drain tests) and lost, because a kernel
// boundary makes the paths that left early WAIT for the last wave of the launch to reach its threshold:
//   single 1080p frame, Sponza-class: 1.59 ms without, 1.78 / 1.86 / 2.01 / 2.39 ms at thresholds 8 / 16 / 32 / 48;
//   Dragon-class 2.45 -> 2.95 / 3.28 / 4.05 / 4.67 ms; an 8-way shard of 20 frames 3.20 -> 3.42 ... 4.11 ms
//   (profiles/r03/experiments/drain_kernel_ab.txt) — and the epilogue that packs the records cost the steady state
//   4.5 % in the 6-waves kernels (two more material colours spilled around the pow calls of every shading pass).
// The LDS ring that moved paths between the waves of a block (same records, compare-and-swap ring, round 3 as well) did
// not shorten the tail either: a straggler is bound by its own dependent fetches, not by the company it keeps
// (profiles/r03/experiments/timeline.txt).
//
// What follows is the code as it was measured: the record, the hand-over and the kernel.

// ---- the end of a launch: stragglers leave for the drain kernel ---------------------------------
// How a launch ends, measured (lab hook -DPBR_EXP_TIMELINE, profiles/r03/experiments/timeline.txt; Sponza-class scene,
// single 1080p frame): the queue is empty after ~1.0 ms with all 393 k lanes holding a path; 250 us later half of them
// are done, after another 250 us 93 % — and the launch then runs for another ~0.7 ms with 2 - 7 % of its lanes: one to
// four paths in each of the 6144 resident waves, the paths with eight bounces and long walks.  Those stragglers are
// bound by the latency of their own dependent node fetches (~1200 visits at 0.3 - 0.5 us), and all 24 waves of a CU keep
// issuing whole node phases and 2000-instruction shading passes for them.  Moving them between the waves of a block was
// built and measured (a ring in LDS; bit-identical): it does not shorten a path, and as a call inside the state
// machine's loop it cost the steady state 6 % in spilled walk constants.  What is built instead:
//   * once a wave's queue is dry and at most P.drainThreshold of its lanes still hold a path, the wave writes those
//     paths — one 112-B record each: ray, walk cursor, closest hit so far, the path's and the frame's running values —
//     to a pool in global memory and ends;
//   * the host launches the DRAIN KERNEL behind it (pathDrain, below): one GROUP of lanes per record, which walks the
//     path's ray cooperatively — the lanes of a group fetch and test the next records of the node stream at once, the
//     group then follows the sequential walk through as many of them as it may — so a straggler's walk needs fewer
//     round trips, in a launch that holds nothing but stragglers.
// Which lane carries a path never changes what the path computes: the record holds exactly the lane state of the loop
// top (mode NODE: mid-walk; SHADE: walk finished), 1 / direction is recomputed with the same IEEE divisions, and the
// per-lane counters are sums.  Bit-identical by construction; tested (tests/test_gpu_parity.py, drain tests).
#define PT_DRAIN_RECORD 8   // float4 per record (7 used): one 128-B line

enum { MODE_NODE = 0, MODE_LEAF = 1, MODE_SHADE = 2, MODE_DONE = 3, MODE_START = 4 };   // MODE_START: drain records of the lock-step kernel — a ray whose walk has not begun

struct WalkState {
	f3 invDir;
	Cursor cur;
	Hit hit;
	int leafFace0, leafFace1;
	float leafTNear;
};

// The wave's remaining paths (lanes in `liveMask`; called by the whole wave) go to the drain pool.
PT_DEV void dumpPaths( const DevParams& P, const PixelState& st, const WalkState& w, int mode, unsigned long long liveMask ) {
	const int lane = (int) __lane_id();
	const int first = __ffsll( (long long) liveMask ) - 1;
	unsigned base = 0u;

	if( lane == first ) {
		base = atomicAdd( P.drainCount, (unsigned) __popcll( liveMask ) );
	}

	base = (unsigned) __shfl( (int) base, first, 64 );

	if( ( liveMask >> lane ) & 1ull ) {
		const unsigned slot = base + (unsigned) __popcll( liveMask & ( ( 1ull << lane ) - 1ull ) );
		float4* rec = P.drainPool + (size_t) slot * PT_DRAIN_RECORD;
		rec[0] = make_float4( st.ray.origin.x, st.ray.origin.y, st.ray.origin.z, st.ray.dir.x );
		rec[1] = make_float4( st.ray.dir.y, st.ray.dir.z, st.color.x, st.color.y );
		rec[2] = make_float4( st.color.z, st.finalColor.x, st.finalColor.y, st.finalColor.z );
		rec[3] = make_float4( st.seed, st.focus, w.hit.t, __int_as_float( w.hit.face ) );
		rec[4] = make_float4( __uint_as_float( st.slot ), __int_as_float( st.frame ), __int_as_float( st.sample ), __uint_as_float( st.secondaryPaths ) );
		rec[5] = make_float4( __uint_as_float( st.dbgNodes ), __uint_as_float( st.dbgTris ), __int_as_float( st.depth ), __int_as_float( st.depthAdded ) );
		rec[6] = make_float4( __int_as_float( w.cur.ref ), __int_as_float( mode ), 0.0f, 0.0f );
	}
}

PT_DEV void loadPath( const DevParams& P, unsigned slot, PixelState& st, WalkState& w, int& mode ) {
	const float4* rec = P.drainPool + (size_t) slot * PT_DRAIN_RECORD;
	const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4], r5 = rec[5], r6 = rec[6];
	st.ray.origin = mk3( r0.x, r0.y, r0.z );
	st.ray.dir = mk3( r0.w, r1.x, r1.y );
	st.color = mk3( r1.z, r1.w, r2.x );
	st.finalColor = mk3( r2.y, r2.z, r2.w );
	st.seed = r3.x;
	st.focus = r3.y;
	w.hit.t = r3.z;
	w.hit.face = __float_as_int( r3.w );
	w.hit.normal = mk3( 0.0f, 0.0f, 0.0f );
	st.slot = __float_as_uint( r4.x );
	st.frame = __float_as_int( r4.y );
	st.sample = __float_as_int( r4.z );
	st.secondaryPaths = __float_as_uint( r4.w );
	st.dbgNodes = __float_as_uint( r5.x );
	st.dbgTris = __float_as_uint( r5.y );
	st.depth = __float_as_int( r5.z );
	st.depthAdded = __float_as_int( r5.w );
	w.cur.ref = __float_as_int( r6.x );
	mode = __float_as_int( r6.y );
	w.invDir = mk3( 1.0f / st.ray.dir.x, 1.0f / st.ray.dir.y, 1.0f / st.ray.dir.z );   // as startWalk computed it
	w.leafFace0 = -1;
	w.leafFace1 = -1;
	w.leafTNear = 0.0f;
}

// Is it time for this wave to leave?  The queue is dry (a lane without a path exists) and few paths are left.
PT_DEV bool leavesForDrain( const DevParams& P, unsigned long long liveMask ) {
	return liveMask != ~0ull && __popcll( liveMask ) <= P.drainThreshold;
}

// ---------------------------------------------------------------------------------------
// The drain kernel: the stragglers of a launch, one GROUP of lanes per path
// ---------------------------------------------------------------------------------------
// Launched behind every path-tracing launch (pbr_hip.hip, launch()); record r of the pool (dumpPaths) belongs to group r.
// Lane 0 of a group OWNS the path (its state, its leaf tests, its shading); all PT_DRAIN_GROUP lanes take part in its node
// phases: lane j fetches and slab-tests the record 32 * j bytes behind the owner's cursor — in the DFS-ordered part of the
// node stream a hit container's successor is the adjacent record (58 % of all successors; the records of a group share a
// 128-byte line) — and the group then follows the walk of pt_bvh.cl:88-117 through them for as long as each record is a
// hit container whose hit successor is the next lane's record; the record at which that chain ends decides where the
// walk goes on (its miss link, a far hit successor, a hit leaf to park on).  The closest hit so far cannot change
// between two leaf tests, so every one of those slab tests is the test the sequential walk would have made: node visits,
// face tests and hits are the reference's, one round trip to memory covers up to PT_DRAIN_GROUP visits, and records that
// were fetched ahead in vain are not counted.
#ifndef PT_DRAIN_GROUP
#define PT_DRAIN_GROUP 4
#endif
#define PT_DRAIN_BLOCK 512

template<int BRDF, bool SHADOW, bool LIGHTS>
__global__ __launch_bounds__( PT_DRAIN_BLOCK, 4 ) void pathDrain( const DevParams P ) {
	const unsigned count = *P.drainCount;
	const unsigned firstGroup = ( blockIdx.x * (unsigned) PT_DRAIN_BLOCK ) / PT_DRAIN_GROUP;

	if( firstGroup >= count ) {
		return;   // the grid is sized for the most records a launch can leave; usually there are fewer
	}

	const float4* lds = gHotNodes;
	stageHotNodes( P, gHotNodes );

	const int lane = (int) __lane_id();
	const int sub = lane & ( PT_DRAIN_GROUP - 1 );
	const int owner = lane & ~( PT_DRAIN_GROUP - 1 );
	const unsigned group = firstGroup + threadIdx.x / PT_DRAIN_GROUP;
	LaneCounters cnt;
	cnt.nodes = cnt.tris = cnt.hits = cnt.paths = 0;
	PixelState st;
	WalkState w;
	int mode = MODE_DONE;   // lanes 1 .. of a group stay DONE: they only help walking

	// every lane holds defined values (the helpers' copies are never used: the walk state comes from the owner)
	st.ray.origin = st.ray.dir = mk3( 0.0f, 0.0f, 0.0f );
	w.invDir = mk3( 0.0f, 0.0f, 0.0f );
	w.cur.ref = -1;
	w.hit.t = inff();
	w.hit.face = 0;

	if( group < count && sub == 0 ) {
		loadPath( P, group, st, w, mode );

		if( mode == MODE_START ) {
			mode = startWalk<LIGHTS>( P, st.ray, w );
		}
	}

#ifdef PBR_GUARD_PATH
	long long guardSteps = 0;
	const long long guardMax = ( (long long) P.samples * ( P.maxDepth + P.maxAddedDepth + 1 ) + 1 ) * ( (long long) P.numNodes + 4 );
#endif

	for( ;; ) {
		if( __ballot( mode != MODE_DONE ) == 0ull ) {
			break;
		}
#ifdef PBR_GUARD_PATH
		if( ++guardSteps > guardMax ) {
			atomicAdd( &P.guard[1], 1u );
			break;
		}
#endif
		// ---- node phase, a group per path
		const unsigned long long walkOwners = __ballot( mode == MODE_NODE );

		if( walkOwners != 0ull ) {
			bool walking = ( ( walkOwners >> owner ) & 1ull ) != 0ull;
			Ray ray;
			ray.origin = mk3( __shfl( st.ray.origin.x, owner, 64 ), __shfl( st.ray.origin.y, owner, 64 ), __shfl( st.ray.origin.z, owner, 64 ) );
			ray.dir = mk3( 0.0f, 0.0f, 0.0f );   // the slab test reads 1 / direction only
			const f3 invDir = mk3( __shfl( w.invDir.x, owner, 64 ), __shfl( w.invDir.y, owner, 64 ), __shfl( w.invDir.z, owner, 64 ) );
			const float rayT = __shfl( w.hit.t, owner, 64 );
			int ref = __shfl( w.cur.ref, owner, 64 );
			const int entered = __popcll( walkOwners );
			const int leave = ( entered * P.parkEighths ) >> 3;
			const int keep = entered - ( ( leave < 1 ) ? 1 : leave );
			unsigned visits = 0;
			int leafWord = 0;
			float leafTNear = 0.0f;
			bool parked = false;

			__builtin_amdgcn_s_setprio( PT_WALK_PRIO );

			for( ;; ) {
				if( walking ) {
					const int myRef = ref + 32 * sub;
					const bool fetchable = ( (unsigned) myRef < (unsigned) P.streamBytes );
					int w0 = 0, w1 = -1;
					float tNear = 0.0f;
					bool isHit = false;

					if( fetchable ) {
						float4 n0, n1;
						Cursor c;
						c.ref = myRef;
						fetchNode<true>( P, lds, c, &n0, &n1 );
						w0 = __float_as_int( n1.z );
						w1 = __float_as_int( n1.w );
						isHit = boxHit<false>( n0, n1, ray, invDir, rayT, &tNear );
					}

					const bool isLeaf = ( w0 < 0 );
					// does the walk go from this lane's record to the next lane's?
					const bool link = fetchable && isHit && !isLeaf && ( w0 == myRef + 32 );
					const unsigned links = (unsigned) ( __ballot( link ) >> owner ) & ( ( 1u << PT_DRAIN_GROUP ) - 1u );
					int consumed = __builtin_ctz( ~links ) + 1;   // 1 + the links that hold from the owner's record on
					consumed = ( consumed > PT_DRAIN_GROUP ) ? PT_DRAIN_GROUP : consumed;
					const int last = owner + consumed - 1;        // the record at which the chain ends decides
					const int nextRef = ( isHit && !isLeaf ) ? w0 : w1;
					const int lastParks = __shfl( ( isHit && isLeaf ) ? 1 : 0, last, 64 );
					const int lastWord = __shfl( w0, last, 64 );
					const float lastTNear = __shfl( tNear, last, 64 );
					ref = __shfl( nextRef, last, 64 );
					visits += (unsigned) consumed;

					if( lastParks != 0 ) {
						parked = true;
						leafWord = lastWord;
						leafTNear = lastTNear;
					}

					walking = ( ref >= 0 ) && !parked;
				}

				if( __popcll( __ballot( walking && sub == 0 ) ) <= keep ) {
					break;
				}
			}

			// ---- leaf phase: the owners of the groups that stopped on a hit leaf
			if( sub == 0 && ( ( walkOwners >> lane ) & 1ull ) != 0ull ) {
				w.cur.ref = ref;
				st.dbgNodes += visits;

				if( parked ) {
					testLeaf<false, true>( P, leafFace0( leafWord ), leafFace1( leafWord ), st.ray, leafTNear, 0.0f, w.hit, st.dbgTris );
				}

				if( !alive( w.cur ) ) {
					mode = MODE_SHADE;
				}
			}

			__builtin_amdgcn_s_setprio( 0 );
		}

		// ---- shade phase: the owners whose ray has left the tree
		{
			const int nShade = __popcll( __ballot( mode == MODE_SHADE ) );
			const int nNode = __popcll( __ballot( mode == MODE_NODE ) );

			if( mode == MODE_SHADE && ( nShade >= P.drainShade || nNode == 0 ) ) {
				if( shadeStep<BRDF, SHADOW, LIGHTS, false, true, false, true>( P, lds, st, cnt, w.hit ) ) {
					finishPixel( P, st );
					mode = MODE_DONE;   // the queue is empty: nothing to take up
				}
				else {
					mode = startWalk<LIGHTS>( P, st.ray, w );
				}
			}
		}
	}

	flushCounters( P, cnt );
}




// ---------------------------------------------------------------------------------------------------------------
// Third attempt (round 3): the same cooperative walk INSIDE the wave — no pool, no second kernel, nothing waits.
// Bit-identical (tests at the commit that removed it), steady state untouched (64-frame launch 64.39 ms either way), and
// no faster where it was meant to help: single 1080p frame, Sponza-class 1.557 ms without, 1.60 / 1.61 ms with it from 8 /
// 16 remaining paths per wave on; Dragon-class 2.41 -> 2.47 / 2.54 ms; an 8-way shard of 20 frames 3.08 -> 3.11 / 3.13 ms
// (profiles/r03/experiments/cooperative_walk.txt).  The stragglers' visits are mostly to the ranked records at the head of
// the stream, whose successors are explicit and not adjacent: nothing to fetch ahead, and six shuffles more per visit.
//
// ---- the cooperative node phase: the end of a launch ---------------------------------------------
// How a launch ends, measured (lab hook -DPBR_EXP_TIMELINE, profiles/r03/experiments/timeline.txt; Sponza-class scene,
// one 1080p frame): the queue is empty after ~1.0 ms with all 393 k lanes holding a path; 250 us later half of them are
// done, after another 250 us 93 % — and the launch runs ~0.7 ms more with 2 - 7 % of its lanes, one to four paths per
// wave: the paths with eight bounces and long walks, each bound by the latency of its own dependent node fetches, while
// 60 lanes of its wave have nothing to do.  Moving such paths elsewhere does not shorten them (two ways were built,
// bit-identical, and measured slower: lab/src/pt_drain.hpp).  Using the idle lanes does: once at most P.coopLanes lanes
// of a wave still hold a path, every path gets a GROUP of PT_COOP_GROUP lanes for its node phases.  Lane j of the group
// fetches and slab-tests the record 32 * j bytes behind the path's cursor — in the DFS-ordered part of the node stream a
// hit container's successor is the adjacent record (58 % of all successors; the records of a group share a 128-byte
// line) — and the group then follows the walk of pt_bvh.cl:88-117 through them for as long as each record is a hit
// container whose hit successor is the next lane's record; the record at which that chain ends decides where the walk
// goes on (its miss link, a far hit successor, a hit leaf to park on).  The closest hit so far cannot change between two
// leaf tests, so each of those slab tests is the test the sequential walk would have made: node visits, face tests and
// hits are the reference's; one round trip to memory covers up to PT_COOP_GROUP visits; records fetched ahead in vain
// are not counted.  The path's state never leaves its lane (the owner): the group's lanes get the ray by shuffles.
#ifndef PT_COOP_GROUP
#define PT_COOP_GROUP 4
#endif

// liveMask: the lanes that hold a path (at most 64 / PT_COOP_GROUP of them); walkMask: those of them that are walking.
// For an owner lane (a bit of walkMask): ref, visits, parked, leafWord, leafTNear are updated as by nodePhaseAsm.
PT_DEV void coopNodePhase(
	const DevParams& P, const float4* lds, unsigned long long liveMask, unsigned long long walkMask, const Ray& ownRay, const f3 ownInvDir,
	float ownRayT, int keepGroups, int& ref, unsigned& visits, int& parked, int& leafWord, float& leafTNear
) {
	const int lane = (int) __lane_id();
	const int sub = lane & ( PT_COOP_GROUP - 1 );
	const int group = lane / PT_COOP_GROUP;
	const int leader = lane & ~( PT_COOP_GROUP - 1 );

	// the owner of group g: the g-th lane that holds a path
	unsigned long long rest = liveMask;

	for( int k = 0; k < group; k++ ) {
		rest &= rest - 1ull;
	}

	const int owner = ( rest != 0ull ) ? __ffsll( (long long) rest ) - 1 : 0;
	bool walking = ( rest != 0ull ) && ( ( walkMask >> owner ) & 1ull ) != 0ull;

	Ray ray;
	ray.origin = mk3( __shfl( ownRay.origin.x, owner, 64 ), __shfl( ownRay.origin.y, owner, 64 ), __shfl( ownRay.origin.z, owner, 64 ) );
	ray.dir = mk3( 0.0f, 0.0f, 0.0f );   // the slab test reads 1 / direction only
	const f3 invDir = mk3( __shfl( ownInvDir.x, owner, 64 ), __shfl( ownInvDir.y, owner, 64 ), __shfl( ownInvDir.z, owner, 64 ) );
	const float rayT = __shfl( ownRayT, owner, 64 );
	int gRef = __shfl( ref, owner, 64 );
	unsigned gVisits = 0;
	int gParked = 0, gWord = 0;
	float gTNear = 0.0f;

	for( ;; ) {
		if( walking ) {
			const int myRef = gRef + 32 * sub;
			const bool fetchable = ( (unsigned) myRef < (unsigned) P.streamBytes );
			int w0 = 0, w1 = -1;
			float tNear = 0.0f;
			bool isHit = false;

			if( fetchable ) {
				float4 n0, n1;
				Cursor c;
				c.ref = myRef;
				fetchNode<true>( P, lds, c, &n0, &n1 );
				w0 = __float_as_int( n1.z );
				w1 = __float_as_int( n1.w );
				isHit = boxHit<false>( n0, n1, ray, invDir, rayT, &tNear );
			}

			const bool isLeaf = ( w0 < 0 );
			// does the walk go from this lane's record to the next lane's?
			const bool link = fetchable && isHit && !isLeaf && ( w0 == myRef + 32 );
			const unsigned links = (unsigned) ( __ballot( link ) >> leader ) & ( ( 1u << PT_COOP_GROUP ) - 1u );
			int consumed = __builtin_ctz( ~links ) + 1;   // 1 + the links that hold from the first record on
			consumed = ( consumed > PT_COOP_GROUP ) ? PT_COOP_GROUP : consumed;
			const int last = leader + consumed - 1;       // the record at which the chain ends decides
			const int nextRef = ( isHit && !isLeaf ) ? w0 : w1;
			const int lastParks = __shfl( ( isHit && isLeaf ) ? 1 : 0, last, 64 );
			const int lastWord = __shfl( w0, last, 64 );
			const float lastTNear = __shfl( tNear, last, 64 );
			gRef = __shfl( nextRef, last, 64 );
			gVisits += (unsigned) consumed;

			if( lastParks != 0 ) {
				gParked = 1;
				gWord = lastWord;
				gTNear = lastTNear;
			}

			walking = ( gRef >= 0 ) && ( gParked == 0 );
		}

		if( __popcll( __ballot( walking && sub == 0 ) ) <= keepGroups ) {
			break;
		}
	}

	// back to the owners: owner lane o is served by the group of its rank among the lanes that hold a path
	const int myGroup = __popcll( liveMask & ( ( 1ull << lane ) - 1ull ) );
	const int from = myGroup * PT_COOP_GROUP;
	const int backRef = __shfl( gRef, from, 64 );
	const unsigned backVisits = (unsigned) __shfl( (int) gVisits, from, 64 );
	const int backParked = __shfl( gParked, from, 64 );
	const int backWord = __shfl( gWord, from, 64 );
	const float backTNear = __shfl( gTNear, from, 64 );

	if( ( ( walkMask >> lane ) & 1ull ) != 0ull ) {
		ref = backRef;
		visits += backVisits;
		parked = backParked;
		leafWord = backWord;
		leafTNear = backTNear;
	}
}


// ... and its place in pathTracingPhased, in front of the node phase:
#if 0
#ifdef PT_NODE_PHASE_ASM
		// ---- the end of the launch: few paths left in this wave — its idle lanes help them walk (coopNodePhase)
		if( __builtin_expect( lanesAtWork <= P.coopLanes, 0 ) ) {
			const unsigned long long liveMask = __ballot( mode != MODE_DONE );
			const unsigned long long walkMask = __ballot( mode == MODE_NODE );

			if( walkMask != 0ull ) {
				const int walkers = __popcll( walkMask );
				const int leave = ( walkers * P.coopParkEighths ) >> 3;
				unsigned visits = 0;
				int leafWord = 0, parkedFlag = 0;
				__builtin_amdgcn_s_setprio( PT_WALK_PRIO );
				coopNodePhase( P, lds, liveMask, walkMask, st.ray, w.invDir, w.hit.t, walkers - ( ( leave < 1 ) ? 1 : leave ), w.cur.ref, visits, parkedFlag, leafWord, w.leafTNear );

				if( mode == MODE_NODE ) {
					st.dbgNodes += visits;

					if( parkedFlag != 0 ) {
						testLeaf<false, ( MINW <= PT_EAGER_UP_TO )>( P, leafFace0( leafWord ), leafFace1( leafWord ), st.ray, w.leafTNear, 0.0f, w.hit, st.dbgTris );
					}

					if( !alive( w.cur ) ) {
						mode = MODE_SHADE;
					}
				}

				__builtin_amdgcn_s_setprio( 0 );
			}
		}
		else
#endif
#endif
