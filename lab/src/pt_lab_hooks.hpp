// Measurement hooks of the path-tracing kernels — LAB BUILDS ONLY (-DPBR_LAB_HOOKS -I lab/src, scripts/lab.sh); the product
// build defines every one of these macros empty (csrc/pt_kernel.hpp).  They record wave-level statistics into the spare
// counter slots [4..15] (pbr_diag_raw_counters) and never change a result.  Choose ONE of:
//   -DPBR_EXP_STATS       lanes per node iteration / leaf phase / shade phase (builds the C++ node loop: add -DPBR_NODE_PHASE_CXX)
//   -DPBR_EXP_TAIL        per wave: start, first empty queue, end on the 100 MHz wall clock (scripts/tail_profile.py)
//   -DPBR_EXP_PHASE_TIME  shader-clock time per phase of the lane state machine (scripts/phase_time.py)
//   -DPBR_EXP_TIMELINE    lane state machine: live lanes over the launch's time, in 8 buckets of PBR_EXP_TIMELINE microseconds
//                         (slots 4..11: lane-ticks of the 100 MHz clock), loop rounds in 4 buckets of twice that (slots 12..15)
#pragma once

// ---- lock-step walk: traverse() -----------------------------------------------------------------
#ifdef PBR_EXP_STATS
#define PT_LAB_TRAVERSE_BEGIN unsigned statIters = 0, statActive = 0, statLeafIters = 0, statLeafActive = 0;
#define PT_LAB_COUNT_WAVE( it, act ) { const unsigned long long m_ = __ballot( 1 ); if( (int) __lane_id() == __ffsll( (long long) m_ ) - 1 ) { it++; act += (unsigned) __popcll( m_ ); } }
#define PT_LAB_NODE_ITERATION PT_LAB_COUNT_WAVE( statIters, statActive )
#define PT_LAB_LEAF_PHASE PT_LAB_COUNT_WAVE( statLeafIters, statLeafActive )
#define PT_LAB_TRAVERSE_END( P, anyhit ) \
	if( !( anyhit ) ) { \
		atomicAdd( &P.counters[4], (unsigned long long) statIters ); \
		atomicAdd( &P.counters[5], (unsigned long long) statActive ); \
		atomicAdd( &P.counters[6], (unsigned long long) statLeafIters ); \
		atomicAdd( &P.counters[7], (unsigned long long) statLeafActive ); \
	}
#else
#define PT_LAB_TRAVERSE_BEGIN
#define PT_LAB_NODE_ITERATION
#define PT_LAB_LEAF_PHASE
#define PT_LAB_TRAVERSE_END( P, anyhit )
#endif

// ---- how a launch ends ----------------------------------------------------------------------------
#ifdef PBR_EXP_TAIL
#define PT_LAB_WAVE_BEGIN const unsigned long long tailStart = wall_clock64(); unsigned long long tailDry = 0ull; (void) tailDry;
#define PT_LAB_WAVE_DRY if( tailDry == 0ull ) { tailDry = wall_clock64(); }
#define PT_LAB_WAVE_END_LOCKSTEP( P ) \
	if( ( threadIdx.x & 63u ) == 0u ) { \
		const unsigned long long tailEnd = wall_clock64(); \
		atomicAdd( &P.counters[12], tailEnd - tailStart ); \
		atomicMax( &P.counters[13], tailEnd ); \
		atomicMax( &P.counters[14], ~tailStart ); \
		atomicAdd( &P.counters[15], 1ull ); \
	}
// the first lane's view of the wave: start, first empty queue seen by any lane, end
#define PT_LAB_WAVE_END_PHASED( P ) \
	{ \
		unsigned long long dry = tailDry; \
		for( int off = 32; off > 0; off >>= 1 ) { \
			const unsigned long long other = ( (unsigned long long) __shfl_xor( (unsigned) ( dry >> 32 ), off, 64 ) << 32 ) | (unsigned long long) __shfl_xor( (unsigned) dry, off, 64 ); \
			dry = ( dry == 0ull ) ? other : ( ( other != 0ull && other < dry ) ? other : dry ); \
		} \
		if( ( threadIdx.x & 63u ) == 0u ) { \
			const unsigned long long tailEnd = wall_clock64(); \
			atomicAdd( &P.counters[12], tailEnd - tailStart ); \
			atomicMax( &P.counters[13], tailEnd ); \
			atomicMax( &P.counters[14], ~tailStart ); \
			atomicAdd( &P.counters[15], 1ull ); \
			atomicAdd( &P.counters[4], ( dry != 0ull ) ? tailEnd - dry : 0ull );      /* time spent draining, summed over the waves */ \
			atomicMax( &P.counters[5], ( dry != 0ull ) ? tailEnd - dry : 0ull );      /* the longest drain */ \
			atomicMax( &P.counters[6], ~( ( dry != 0ull ) ? dry : tailEnd ) );        /* the earliest "queue empty" of the launch */ \
			atomicMax( &P.counters[7], tailStart );                                    /* the latest wave start */ \
		} \
	}
#else
#define PT_LAB_WAVE_BEGIN
#define PT_LAB_WAVE_DRY
#define PT_LAB_WAVE_END_LOCKSTEP( P )
#define PT_LAB_WAVE_END_PHASED( P )
#endif

// ---- lane state machine: lanes per phase / time per phase --------------------------------------
#if defined( PBR_EXP_STATS )
#define PT_LAB_PHASED_BEGIN unsigned sIt[4] = { 0, 0, 0, 0 }, sAct[4] = { 0, 0, 0, 0 };
// what: 0 node iteration, 1 leaf phase, 2 shade phase, 3 node phase entered
#define PT_LAB_PHASED_STAT( what ) { const unsigned long long m_ = __ballot( 1 ); if( (int) __lane_id() == __ffsll( (long long) m_ ) - 1 ) { sIt[what]++; sAct[what] += (unsigned) __popcll( m_ ); } }
#define PT_LAB_PHASED_END( P ) \
	atomicAdd( &P.counters[8], (unsigned long long) sIt[0] ); \
	atomicAdd( &P.counters[9], (unsigned long long) sAct[0] ); \
	atomicAdd( &P.counters[10], (unsigned long long) sIt[1] ); \
	atomicAdd( &P.counters[11], (unsigned long long) sAct[1] ); \
	atomicAdd( &P.counters[12], (unsigned long long) sIt[2] ); \
	atomicAdd( &P.counters[13], (unsigned long long) sAct[2] ); \
	atomicAdd( &P.counters[14], (unsigned long long) sIt[3] );
#define PT_LAB_PHASED_NODE_BEGIN( mode )
#define PT_LAB_PHASED_NODE_MID
#define PT_LAB_PHASED_LEAF_END
#define PT_LAB_PHASED_NODE_END
#define PT_LAB_PHASED_SHADE_BEGIN
#define PT_LAB_PHASED_SHADE_END
#elif defined( PBR_EXP_TIMELINE )
// per-wave accumulators (a small array in scratch), flushed once at the end: an atomic per loop round made the launch 17x longer
#define PT_LAB_PHASED_BEGIN const unsigned long long tl0 = wall_clock64(); unsigned long long tlLast = tl0; unsigned tlLanes[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tlRounds[4] = { 0, 0, 0, 0 };
#define PT_LAB_PHASED_STAT( what )
#define PT_LAB_PHASED_NODE_BEGIN( mode ) \
	{ \
		const unsigned long long now_ = wall_clock64(); \
		const int live_ = __popcll( __ballot( mode != MODE_DONE ) ); \
		const unsigned long long b_ = ( now_ - tl0 ) / ( 100ull * PBR_EXP_TIMELINE ); \
		tlLanes[b_ > 7ull ? 7ull : b_] += (unsigned) live_ * (unsigned) ( now_ - tlLast ); \
		tlRounds[b_ / 2ull > 3ull ? 3ull : b_ / 2ull] += 1u; \
		tlLast = now_; \
	}
#define PT_LAB_PHASED_NODE_MID
#define PT_LAB_PHASED_LEAF_END
#define PT_LAB_PHASED_NODE_END
#define PT_LAB_PHASED_SHADE_BEGIN
#define PT_LAB_PHASED_SHADE_END
#define PT_LAB_PHASED_END( P ) \
	if( ( threadIdx.x & 63u ) == 0u ) { \
		for( int k_ = 0; k_ < 8; k_++ ) { atomicAdd( &P.counters[4 + k_], (unsigned long long) tlLanes[k_] ); } \
		for( int k_ = 0; k_ < 4; k_++ ) { atomicAdd( &P.counters[12 + k_], (unsigned long long) tlRounds[k_] ); } \
	}
#elif defined( PBR_EXP_LEAFHIST )
// round 4: how many lanes stand on a leaf when a node phase ends — a histogram over the leaf phases
// (slots 4..11: 0, 1-4, 5-8, 9-12, 13-16, 17-24, 25-32, 33+ parked lanes), node phases (12), parked lanes (13), lanes whose
// walk ended in the phase (14), lanes that entered (15)
#define PT_LAB_PHASED_BEGIN unsigned lhHist[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, lhPhases = 0, lhParked = 0, lhFinished = 0, lhEntered = 0;
#define PT_LAB_PHASED_STAT( what )
#define PT_LAB_PHASED_NODE_BEGIN( mode )
#define PT_LAB_PHASED_NODE_MID \
	{ \
		const int np_ = __popcll( __ballot( parkedFlag != 0 ) ); \
		const int nf_ = __popcll( __ballot( parkedFlag == 0 && w.cur.ref < 0 ) ); \
		const int b_ = ( np_ == 0 ) ? 0 : ( np_ <= 16 ) ? ( np_ + 3 ) / 4 : ( np_ <= 32 ) ? 5 + ( np_ - 17 ) / 8 : 7; \
		lhHist[b_]++; lhPhases++; lhParked += (unsigned) np_; lhFinished += (unsigned) nf_; lhEntered += (unsigned) __popcll( __ballot( 1 ) ); \
	}
#define PT_LAB_PHASED_LEAF_END
#define PT_LAB_PHASED_NODE_END
#define PT_LAB_PHASED_SHADE_BEGIN
#define PT_LAB_PHASED_SHADE_END
#define PT_LAB_PHASED_END( P ) \
	if( ( threadIdx.x & 63u ) == 0u ) { \
		for( int k_ = 0; k_ < 8; k_++ ) { atomicAdd( &P.counters[4 + k_], (unsigned long long) lhHist[k_] ); } \
		atomicAdd( &P.counters[12], (unsigned long long) lhPhases ); atomicAdd( &P.counters[13], (unsigned long long) lhParked ); \
		atomicAdd( &P.counters[14], (unsigned long long) lhFinished ); atomicAdd( &P.counters[15], (unsigned long long) lhEntered ); \
	}
#elif defined( PBR_EXP_PHASE_TIME )
// clock64 = the shader clock; taken where the wave is converged
#define PT_LAB_PHASED_BEGIN unsigned long long phaseTime[3] = { 0ull, 0ull, 0ull }; const long long phaseStart = clock64();
#define PT_LAB_PHASED_STAT( what )
#define PT_LAB_PHASED_NODE_BEGIN( mode ) const unsigned long long maskNode = __ballot( mode == MODE_NODE ); const long long tNode0 = clock64(); long long leafDelta = 0; long long tPhase1 = 0; (void) tPhase1;
#define PT_LAB_PHASED_NODE_MID tPhase1 = clock64();
#define PT_LAB_PHASED_LEAF_END leafDelta = clock64() - tPhase1;
#define PT_LAB_PHASED_NODE_END \
	if( maskNode != 0ull ) { \
		const long long both = clock64() - tNode0; \
		const int src = __ffsll( (long long) maskNode ) - 1; \
		const long long leafU = ( (long long) __shfl( (int) ( leafDelta >> 32 ), src, 64 ) << 32 ) | (long long) (unsigned) __shfl( (int) leafDelta, src, 64 ); \
		phaseTime[1] += (unsigned long long) leafU; \
		phaseTime[0] += (unsigned long long) ( both - leafU ); \
	}
#define PT_LAB_PHASED_SHADE_BEGIN const long long tShade0 = clock64();
#define PT_LAB_PHASED_SHADE_END phaseTime[2] += (unsigned long long) ( clock64() - tShade0 );
#define PT_LAB_PHASED_END( P ) \
	if( ( threadIdx.x & 63u ) == 0u ) { \
		atomicAdd( &P.counters[4], phaseTime[0] );                                    /* node phases */ \
		atomicAdd( &P.counters[5], phaseTime[1] );                                    /* leaf phases */ \
		atomicAdd( &P.counters[6], phaseTime[2] );                                    /* shade checks + shading */ \
		atomicAdd( &P.counters[7], (unsigned long long) ( clock64() - phaseStart ) );  /* the wave's whole life */ \
	}
#else
#define PT_LAB_PHASED_BEGIN
#define PT_LAB_PHASED_STAT( what )
#define PT_LAB_PHASED_NODE_BEGIN( mode )
#define PT_LAB_PHASED_NODE_MID
#define PT_LAB_PHASED_LEAF_END
#define PT_LAB_PHASED_NODE_END
#define PT_LAB_PHASED_SHADE_BEGIN
#define PT_LAB_PHASED_SHADE_END
#define PT_LAB_PHASED_END( P )
#endif
