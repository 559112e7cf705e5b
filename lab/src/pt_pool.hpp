// The pooled schedule: the paths of a BLOCK in LDS, walked and shaded by different waves.
//
// In the lane state machine (pathTracingPhased) a lane whose ray has left the tree waits until 40 lanes of ITS wave
// want shading, and is missing from the node phases meanwhile: measured on the Sponza-class scene, node phases run at
// 41.6 of 64 lanes, shading passes at 44.9.  Every rearrangement inside a wave trades one
// phase's lanes for another's.  Here the unit that waits is not a lane but a PATH, and it waits in a queue of the
// whole block:
//
//   * every path of the block has a record in LDS: its ray, its last hit, and the 14 values of PixelState that
//     outlive a bounce (22 dwords; one path per thread of the block: 768 records = 66 KB);
//   * WALKER waves: a lane without a ray pops a path from the block's walk queue, walks its ray (the same
//     hand-scheduled node phase and parked leaf phase as everywhere), writes {t, face, visit counts} back and pushes
//     the path to the shade queue — it never waits for shading, its next ray is whichever is ready;
//   * SHADER waves take 64 paths at a time from the shade queue, run shadeStep on them at full width, and push them
//     back to the walk queue with their next ray (or start the next unit of work in the same record, or retire it).
//
// The wave64 ballot / prefix sum is how a wave draws from a queue: one compare-and-swap on the head for all the lanes
// that want an entry, each lane's entry at head + its rank among them.  Per path the sequence of node visits, face
// tests and random draws is exactly the reference's; only which lane executes it changes.
//
// What it costs: the LDS that staged the tree top (measured: 5 - 10 % of the state machine's rate) and two LDS round
// trips per ray.  Queues are rings of RING entries (a power of two >= the pool) of path ids; an entry is valid once it
// is not POOL_EMPTY (the producer bumps the tail first, then writes).  Every wait is bounded: a wave that spins
// POOL_SPIN_LIMIT times sets the block's abort flag and P.guard[1], and everybody leaves (no hung GPU).
#pragma once

namespace ptk {

#define POOL_RING 1024u
#define POOL_EMPTY 0xFFFFFFFFu
#define POOL_SPIN_LIMIT ( 1u << 22 )

enum { PQ_WALK_HEAD = 0, PQ_WALK_TAIL, PQ_SHADE_HEAD, PQ_SHADE_TAIL, PQ_LIVE, PQ_ABORT, PQ_CTL_WORDS = 16 };

enum {
	PF_OX = 0, PF_OY, PF_OZ, PF_DX, PF_DY, PF_DZ,       // ray (shader -> walker)
	PF_T, PF_FACE,                                       // hit (walker -> shader)
	PF_SLOT, PF_FRAME, PF_PACK,                          // pixel slot, frame, sample | depth << 8 | depthAdded << 16
	PF_FCX, PF_FCY, PF_FCZ, PF_SEC, PF_FOCUS, PF_SEED,   // finalColor, secondaryPaths, focus, seed
	PF_CX, PF_CY, PF_CZ,                                 // color
	PF_NODES, PF_TRIS,                                   // debug counters of the unit
	PF_COUNT
};

struct PoolLds {
	unsigned* ctl;     // PQ_*
	unsigned* walkQ;   // POOL_RING path ids
	unsigned* shadeQ;  // POOL_RING path ids
	unsigned* field;   // PF_COUNT x capacity
	unsigned capacity;
};

PT_DEV PoolLds poolLayout( void* base, unsigned capacity ) {
	PoolLds L;
	L.ctl = (unsigned*) base;
	L.walkQ = L.ctl + PQ_CTL_WORDS;
	L.shadeQ = L.walkQ + POOL_RING;
	L.field = L.shadeQ + POOL_RING;
	L.capacity = capacity;
	return L;
}

__host__ __device__ inline size_t poolLdsBytes( unsigned capacity ) {
	return sizeof( unsigned ) * ( (size_t) PQ_CTL_WORDS + 2u * POOL_RING + (size_t) PF_COUNT * capacity );
}

PT_DEV unsigned poolLoad( const unsigned* p ) {
	return __hip_atomic_load( p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );
}

PT_DEV float& poolF( const PoolLds& L, int f, unsigned id ) {
	return *(float*) &L.field[(unsigned) f * L.capacity + id];
}

PT_DEV unsigned& poolU( const PoolLds& L, int f, unsigned id ) {
	return L.field[(unsigned) f * L.capacity + id];
}

// The lanes with `want` each take one entry of the ring, or none: one compare-and-swap on the head for the wave.
// Fewer than `minBatch` available -> nobody takes anything.  Returns the number of lanes served; *id is valid for those.
PT_DEV int poolWavePop( unsigned* ctl, int headIdx, int tailIdx, unsigned* ring, bool want, int minBatch, unsigned* id ) {
	const unsigned long long mask = __ballot( want );
	const int need = __popcll( mask );

	if( need == 0 ) {
		return 0;
	}

	const int leader = __ffsll( (long long) mask ) - 1;
	unsigned h = 0u;
	int n = 0;

	if( (int) __lane_id() == leader ) {
		for( ;; ) {
			h = poolLoad( &ctl[headIdx] );
			const unsigned t = poolLoad( &ctl[tailIdx] );
			const int avail = (int) ( t - h );
			n = ( need < avail ) ? need : avail;

			if( n < minBatch || n <= 0 ) {
				n = 0;
				break;
			}

			unsigned expected = h;

			if( __hip_atomic_compare_exchange_strong( &ctl[headIdx], &expected, h + (unsigned) n, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP ) ) {
				break;
			}
		}
	}

	h = (unsigned) __shfl( (int) h, leader, 64 );
	n = __shfl( n, leader, 64 );

	if( n == 0 ) {
		return 0;
	}

	const int rank = __popcll( mask & ( ( 1ull << __lane_id() ) - 1ull ) );

	if( want && rank < n ) {
		const unsigned pos = ( h + (unsigned) rank ) & ( POOL_RING - 1u );
		unsigned got = POOL_EMPTY;

		// the producer bumps the tail before it writes the entry
		for( unsigned spin = 0; spin < POOL_SPIN_LIMIT; spin++ ) {
			got = __hip_atomic_load( &ring[pos], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP );

			if( got != POOL_EMPTY ) {
				break;
			}
		}

		__hip_atomic_store( &ring[pos], POOL_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );
		*id = got;
	}

	return n;
}

// The lanes with `have` each append their path id: one add on the tail for the wave.  The record's fields must have
// been written before (release).
PT_DEV void poolWavePush( unsigned* ctl, int tailIdx, unsigned* ring, bool have, unsigned id ) {
	const unsigned long long mask = __ballot( have );
	const int count = __popcll( mask );

	if( count == 0 ) {
		return;
	}

	const int leader = __ffsll( (long long) mask ) - 1;
	unsigned base = 0u;

	if( (int) __lane_id() == leader ) {
		base = __hip_atomic_fetch_add( &ctl[tailIdx], (unsigned) count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP );
	}

	base = (unsigned) __shfl( (int) base, leader, 64 );

	if( have ) {
		const int rank = __popcll( mask & ( ( 1ull << __lane_id() ) - 1ull ) );
		__hip_atomic_store( &ring[( base + (unsigned) rank ) & ( POOL_RING - 1u )], id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP );
	}
}

PT_DEV void poolStoreState( const PoolLds& L, unsigned id, const PixelState& st ) {
	poolF( L, PF_OX, id ) = st.ray.origin.x; poolF( L, PF_OY, id ) = st.ray.origin.y; poolF( L, PF_OZ, id ) = st.ray.origin.z;
	poolF( L, PF_DX, id ) = st.ray.dir.x; poolF( L, PF_DY, id ) = st.ray.dir.y; poolF( L, PF_DZ, id ) = st.ray.dir.z;
	poolU( L, PF_SLOT, id ) = st.slot;
	poolU( L, PF_FRAME, id ) = (unsigned) st.frame;
	poolU( L, PF_PACK, id ) = (unsigned) st.sample | ( (unsigned) st.depth << 8 ) | ( (unsigned) st.depthAdded << 16 );
	poolF( L, PF_FCX, id ) = st.finalColor.x; poolF( L, PF_FCY, id ) = st.finalColor.y; poolF( L, PF_FCZ, id ) = st.finalColor.z;
	poolU( L, PF_SEC, id ) = st.secondaryPaths;
	poolF( L, PF_FOCUS, id ) = st.focus;
	poolF( L, PF_SEED, id ) = st.seed;
	poolF( L, PF_CX, id ) = st.color.x; poolF( L, PF_CY, id ) = st.color.y; poolF( L, PF_CZ, id ) = st.color.z;
	poolU( L, PF_NODES, id ) = st.dbgNodes;
	poolU( L, PF_TRIS, id ) = st.dbgTris;
}

PT_DEV void poolLoadState( const PoolLds& L, unsigned id, PixelState& st, Hit& hit ) {
	st.ray.origin = mk3( poolF( L, PF_OX, id ), poolF( L, PF_OY, id ), poolF( L, PF_OZ, id ) );
	st.ray.dir = mk3( poolF( L, PF_DX, id ), poolF( L, PF_DY, id ), poolF( L, PF_DZ, id ) );
	hit.t = poolF( L, PF_T, id );
	hit.face = (int) poolU( L, PF_FACE, id );
	hit.normal = mk3( 0.0f, 0.0f, 0.0f );
	st.slot = poolU( L, PF_SLOT, id );
	st.frame = (int) poolU( L, PF_FRAME, id );
	const unsigned pack = poolU( L, PF_PACK, id );
	st.sample = (int) ( pack & 255u );
	st.depth = (int) ( ( pack >> 8 ) & 255u );
	st.depthAdded = (int) ( ( pack >> 16 ) & 255u );
	st.finalColor = mk3( poolF( L, PF_FCX, id ), poolF( L, PF_FCY, id ), poolF( L, PF_FCZ, id ) );
	st.secondaryPaths = poolU( L, PF_SEC, id );
	st.focus = poolF( L, PF_FOCUS, id );
	st.seed = poolF( L, PF_SEED, id );
	st.color = mk3( poolF( L, PF_CX, id ), poolF( L, PF_CY, id ), poolF( L, PF_CZ, id ) );
	st.dbgNodes = poolU( L, PF_NODES, id );
	st.dbgTris = poolU( L, PF_TRIS, id );
	st.acc = mk3( 0.0f, 0.0f, 0.0f );
	st.accW = 0.0f;
}

PT_DEV bool poolAborted( const PoolLds& L ) {
	return poolLoad( &L.ctl[PQ_ABORT] ) != 0u;
}

PT_DEV void poolAbort( const DevParams& P, const PoolLds& L ) {
	__hip_atomic_store( &L.ctl[PQ_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );
	atomicAdd( &P.guard[1], 1u );
}

#ifdef PT_NODE_PHASE_ASM

#ifdef PBR_POOL_STATS   // lab only: per-wave tallies, added to P.counters[4..10] when the wave ends
#define POOL_STAT( slot, value ) { poolStats[( slot ) - 4] += (unsigned) ( value ); }
#else
#define POOL_STAT( slot, value )
#endif

// P.poolShaders waves of the block (the last ones) shade, the others walk.  Frame-parallel units only.
template<int BRDF, bool SHADOW, bool LIGHTS, int MINW>
__global__ __launch_bounds__( PBR_BLOCK, MINW ) void pathTracingPooled( const DevParams P ) {
	const PoolLds L = poolLayout( (void*) gHotNodes, blockDim.x );
	const float4* lds = gHotNodes;   // no staged nodes (P.numHotBytes = 0): shadow walks read the stream from memory
	const unsigned wave = threadIdx.x >> 6;
	const unsigned waves = blockDim.x >> 6;
	const bool shader = ( wave >= waves - (unsigned) P.poolShaders );

	// ---- set-up: empty rings, then every thread opens one path (its record = its thread index) ----
	for( unsigned i = threadIdx.x; i < PQ_CTL_WORDS + 2u * POOL_RING; i += blockDim.x ) {
		L.ctl[i] = ( i < PQ_CTL_WORDS ) ? 0u : POOL_EMPTY;
	}

	__syncthreads();

	LaneCounters cnt;
	cnt.nodes = cnt.tris = cnt.hits = cnt.paths = 0;
	WorkCursor work = beginWork();
	unsigned frame = 0;
#ifdef PBR_POOL_STATS
	unsigned poolStats[7] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u };
#endif

	{
		PixelState st;
		const unsigned slot = nextSlot( P, work, (unsigned) P.nFrames, frame );
		const bool opened = ( slot != PT_NO_WORK );

		if( opened ) {
			beginPixel<true>( P, st, slot, cnt, frame );
			poolStoreState( L, threadIdx.x, st );
		}

		const int count = __popcll( __ballot( opened ) );

		if( ( threadIdx.x & 63u ) == 0u && count > 0 ) {
			__hip_atomic_fetch_add( &L.ctl[PQ_LIVE], (unsigned) count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );
		}

		poolWavePush( L.ctl, PQ_WALK_TAIL, L.walkQ, opened, threadIdx.x );
	}

	__syncthreads();

	if( !shader ) {
		// ================================ walker ================================
		bool have = false;
		unsigned id = 0u;
		Ray ray;
		ray.origin = mk3( 0.0f, 0.0f, 0.0f );
		ray.dir = mk3( 0.0f, 0.0f, 1.0f );
		f3 invDir = mk3( 0.0f, 0.0f, 0.0f );
		Hit hit;
		hit.t = inff();
		hit.face = 0;
		hit.normal = mk3( 0.0f, 0.0f, 0.0f );
		int ref = -1;
		unsigned visitsAcc = 0u, trisAcc = 0u;
		unsigned idle = 0u;

		for( ;; ) {
			// ---- lanes without a ray draw one
			{
				unsigned got = 0u;
				const bool want = !have;
				const int served = poolWavePop( L.ctl, PQ_WALK_HEAD, PQ_WALK_TAIL, L.walkQ, want, 1, &got );
				const int rank = __popcll( __ballot( want ) & ( ( 1ull << __lane_id() ) - 1ull ) );

				if( want && rank < served ) {
					id = got;
					have = true;
					ray.origin = mk3( poolF( L, PF_OX, id ), poolF( L, PF_OY, id ), poolF( L, PF_OZ, id ) );
					ray.dir = mk3( poolF( L, PF_DX, id ), poolF( L, PF_DY, id ), poolF( L, PF_DZ, id ) );
					invDir = mk3( 1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z );
					hit.t = inff();
					hit.face = 0;
					ref = P.firstRef;
					visitsAcc = 0u;
					trisAcc = 0u;

					if( LIGHTS ) {
						traverseLights( P, ray, hit );
					}
				}
			}

			if( __ballot( have ) == 0ull ) {
				POOL_STAT( 4, 1 )                                   // walker rounds with no ray at all
				if( poolLoad( &L.ctl[PQ_LIVE] ) == 0u || poolAborted( L ) ) {
					break;
				}

				if( ++idle > POOL_SPIN_LIMIT ) {
					poolAbort( P, L );
					break;
				}

				__builtin_amdgcn_s_sleep( 8 );
				continue;
			}

			idle = 0u;
			POOL_STAT( 5, 1 )                                       // walker rounds with rays
			POOL_STAT( 6, __popcll( __ballot( have ) ) )             // ... and the lanes that hold a ray in them

			// ---- node phase + leaf phase for the lanes that hold a ray
			if( have ) {
				const int entered = __popcll( __ballot( 1 ) );
				const int park = ( ( P.phPark * entered ) >> 6 ) < 1 ? 1 : ( ( P.phPark * entered ) >> 6 );
				const int keep = entered - park;
				const f2v oxy = { ray.origin.x, ray.origin.y };
				const f2v ozz = { ray.origin.z, ray.origin.z };
				const f2v ixy = { invDir.x, invDir.y };
				const f2v izz = { invDir.z, invDir.z };
				int leafWord = 0, parkedFlag;
				float leafTNear, unusedTFar;
				unsigned visits = 0u;
				nodePhaseAsm<false>( P, oxy, ozz, ixy, izz, hit.t, ( keep < 0 ) ? 0 : keep, ref, visits, leafWord, leafTNear, unusedTFar, parkedFlag );
				visitsAcc += visits;
#ifdef PBR_POOL_STATS
				{
					const unsigned most = (unsigned) __builtin_amdgcn_readfirstlane( (int) waveMax( visits ) );
					POOL_STAT( 7, most )                                // node iterations
				}
#endif

				if( parkedFlag != 0 ) {
					testLeaf<false, false>( P, leafFace0( leafWord ), leafFace1( leafWord ), ray, leafTNear, 0.0f, hit, trisAcc );
				}
			}

			// ---- the lanes whose ray has left the tree hand their path to the shaders
			{
				const bool finished = have && ( ref < 0 );

				if( finished ) {
					poolF( L, PF_T, id ) = hit.t;
					poolU( L, PF_FACE, id ) = (unsigned) hit.face;
					poolU( L, PF_NODES, id ) += visitsAcc;
					poolU( L, PF_TRIS, id ) += trisAcc;
					have = false;
				}

				poolWavePush( L.ctl, PQ_SHADE_TAIL, L.shadeQ, finished, id );
			}
		}
	}
	else {
		// ================================ shader ================================
		unsigned idle = 0u;

		for( ;; ) {
			// a full wave of paths if there is one; after P.poolPatience empty-handed polls, whatever is there
			unsigned id = 0u;
			const int minBatch = ( idle >= (unsigned) P.poolPatience ) ? 1 : 64;
			const int served = poolWavePop( L.ctl, PQ_SHADE_HEAD, PQ_SHADE_TAIL, L.shadeQ, true, minBatch, &id );

			if( served == 0 ) {
				POOL_STAT( 8, 1 )                                   // shader polls that came back empty-handed
				if( poolLoad( &L.ctl[PQ_LIVE] ) == 0u || poolAborted( L ) ) {
					break;
				}

				if( ++idle > POOL_SPIN_LIMIT ) {
					poolAbort( P, L );
					break;
				}

				__builtin_amdgcn_s_sleep( 4 );
				continue;
			}

			idle = 0u;
			POOL_STAT( 9, 1 )                                       // shading batches
			POOL_STAT( 10, served )                                 // ... and the paths in them
			const bool mine = ( (int) __lane_id() < served );
			bool again = false;    // the path goes back to the walkers
			bool retired = false;

			if( mine ) {
				PixelState st;
				Hit hit;
				poolLoadState( L, id, st, hit );

				if( shadeStep<BRDF, SHADOW, LIGHTS, true, false, false, true>( P, lds, st, cnt, hit ) ) {
					finishPixel<true>( P, st );

					if( cnt.nodes > 0x40000000u || cnt.tris > 0x40000000u ) {
						flushCounters( P, cnt );
					}

					const unsigned slot = nextSlot( P, work, (unsigned) P.nFrames, frame );

					if( slot != PT_NO_WORK ) {
						beginPixel<true>( P, st, slot, cnt, frame );
						again = true;
					}
					else {
						retired = true;
					}
				}
				else {
					again = true;
				}

				if( again ) {
					poolStoreState( L, id, st );
				}
			}

			const int gone = __popcll( __ballot( retired ) );

			if( gone > 0 && __lane_id() == 0u ) {
				__hip_atomic_fetch_sub( &L.ctl[PQ_LIVE], (unsigned) gone, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP );
			}

			poolWavePush( L.ctl, PQ_WALK_TAIL, L.walkQ, again, id );
		}
	}

	flushCounters( P, cnt );
#ifdef PBR_POOL_STATS
	if( ( threadIdx.x & 63u ) == 0u ) {
		for( int k = 0; k < 7; k++ ) {
			atomicAdd( &P.counters[4 + k], (unsigned long long) poolStats[k] );
		}
	}
#endif
}

#endif   // PT_NODE_PHASE_ASM

}  // namespace ptk
