// Micro-benchmark (measurement aid, not product): LDS-DMA (`global_load_lds_dwordx4`) as a PER-LANE asynchronous
// fetch that a wave polls instead of waiting for with s_waitcnt — the mechanism of the asynchronous node phase
// (pt_kernel.hpp, nodePhaseAsync).  Three questions, each answered by a kernel below:
//
//   layout   where does a lane's 16 B land?  (M0 + instruction offset + lane * 16; inactive lanes write nothing;
//            M0 beyond 64 KiB)
//   poll     a lane arms a marker word in its slot, issues the DMA of a 32-B record (two instructions) and READS ITS
//            SLOT WITHOUT WAITING on vmcnt until the marker has changed.  Is a record that shows its marker complete —
//            are the 16 bytes of one lane written at once, and does the second instruction's data land after the
//            first's?  Every consumed record is checked word by word; torn records are counted.
//   rate     dependent random 32-B gathers per second: the synchronous loop (registers, s_waitcnt vmcnt(0): an
//            iteration ends on its slowest lane) against the polled one at several "lanes that must be ready" shares,
//            on tables with an L2-resident part and a far part (the Sponza-class mix: most requests are near hits,
//            one in six goes far).
//   hipcc --offload-arch=gfx950 -O3 -o glds_poll glds_poll.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned u4 __attribute__( ( ext_vector_type( 4 ) ) );

#define PLANE 16384   // bytes between a lane's two 16-B halves: 1024 lanes x 16 B

extern __shared__ u4 gLds[];

// ---- layout ----------------------------------------------------------------------------------
__global__ void layoutProbe( const u4* table, unsigned m0, unsigned laneLo, unsigned laneHi, int instOffset, u4* dump, int dumpQuads ) {
	for( int i = (int) threadIdx.x; i < dumpQuads; i += (int) blockDim.x ) {
		gLds[i] = (u4) { 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu };
	}

	__syncthreads();
	const unsigned lane = threadIdx.x & 63u;

	if( threadIdx.x < 64u && lane >= laneLo && lane < laneHi ) {
		const unsigned byteOff = lane * 64u;   // table quad 4 * lane (+ 1 with offset:16)
		unsigned keep;

		if( instOffset == 0 ) {
			asm volatile( "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
			              : "=&s"( keep ) : "v"( byteOff ), "s"( m0 ), "s"( table ) : "memory" );
		}
		else {
			asm volatile( "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:16\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
			              : "=&s"( keep ) : "v"( byteOff ), "s"( m0 ), "s"( table ) : "memory" );
		}
	}

	__syncthreads();

	for( int i = (int) threadIdx.x; i < dumpQuads; i += (int) blockDim.x ) {
		dump[i] = gLds[i];
	}
}

// ---- poll / rate ---------------------------------------------------------------------------------
// record r = 2 quads: {r, ~r, r * 2654435761, r ^ 0x5bd1e995} {r + 0x9e3779b9, r * 3, r ^ 0xa5a5a5a5, next(r) * 32}
__device__ __forceinline__ unsigned checkRecord( const u4 a, const u4 b, unsigned r ) {
	unsigned bad = 0;
	bad |= ( a.x != r ) ? 1u : 0u;
	bad |= ( a.y != ~r ) ? 2u : 0u;
	bad |= ( a.z != r * 2654435761u ) ? 4u : 0u;
	bad |= ( a.w != ( r ^ 0x5bd1e995u ) ) ? 8u : 0u;
	bad |= ( b.x != r + 0x9e3779b9u ) ? 16u : 0u;
	bad |= ( b.y != r * 3u ) ? 32u : 0u;
	bad |= ( b.z != ( r ^ 0xa5a5a5a5u ) ) ? 64u : 0u;
	return bad;
}

__global__ __launch_bounds__( 1024 ) void chaseSync( const u4* table, unsigned mask, int steps, unsigned* out, unsigned long long* stats ) {
	unsigned r = ( ( blockIdx.x * blockDim.x + threadIdx.x ) * 2654435761u ) & mask;
	unsigned bad = 0;

	for( int i = 0; i < steps; i++ ) {
		const u4 a = table[(size_t) r * 2];
		const u4 b = table[(size_t) r * 2 + 1];
		bad |= checkRecord( a, b, r );
		r = b.w >> 5;
	}

	out[blockIdx.x * blockDim.x + threadIdx.x] = r;

	if( bad != 0u ) {
		atomicAdd( &stats[0], 1ull );
	}
}

// MARK0: also arm and check a marker in the FIRST half (catches "second half landed before the first")
template<bool MARK0>
__global__ __launch_bounds__( 1024 ) void chaseAsync( const u4* table, unsigned mask, int steps, int needEighths, unsigned* out, unsigned long long* stats ) {
	const unsigned tid = threadIdx.x;
	const unsigned wave = tid >> 6;
	const unsigned slot = tid * 16u;                      // LDS byte address of this lane's first half; second at + PLANE
	const unsigned m0a = __builtin_amdgcn_readfirstlane( wave * 1024u );                   // + lane * 16 by the hardware
	const unsigned m0b = __builtin_amdgcn_readfirstlane( PLANE + wave * 1024u - 16u );     // the instruction offset (16) is added to the LDS address too
	unsigned r = ( ( blockIdx.x * blockDim.x + tid ) * 2654435761u ) & mask;
	unsigned bad = 0, tornOrder = 0, tornWords = 0;
	unsigned long long polls = 0, iterations = 0, lanesReady = 0;
	const unsigned one = 1u, ones = 0xFFFFFFFFu;

	// arm + first request
	asm volatile( "ds_write_b32 %0, %1 offset:%c2\n\tds_write_b32 %0, %3\n\ts_waitcnt lgkmcnt(0)" :: "v"( slot ), "v"( one ), "n"( PLANE + 12 ), "v"( ones ) : "memory" );
	{
		unsigned keep;
		const unsigned byteOff = r * 32u;
		asm volatile( "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4 offset:16\n\ts_mov_b32 m0, %0"
		              : "=&s"( keep ) : "v"( byteOff ), "s"( m0a ), "s"( m0b ), "s"( table ) : "memory" );
	}

	int done = 0;
	bool walking = ( steps > 0 );

	while( __ballot( walking ) != 0ull ) {
		if( walking ) {
			u4 a, b;
			asm volatile( "ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%c3\n\ts_waitcnt lgkmcnt(0)" : "=&v"( a ), "=&v"( b ) : "v"( slot ), "n"( PLANE ) : "memory" );
			const bool ready = ( b.w != 1u );
			const int count = __popcll( __ballot( ready ) );
			const int nwalk = __popcll( __ballot( 1 ) );
			int need = ( nwalk * needEighths ) >> 3;
			need = ( need < 1 ) ? 1 : need;
			polls++;

			if( polls > 4000000ull ) {    // never hang a GPU over a wrong assumption: give up loudly
				walking = false;
				bad |= 0x80000000u;
				continue;
			}

			if( count < need ) {
				__builtin_amdgcn_s_sleep( 1 );
				continue;
			}

			iterations++;
			lanesReady += (unsigned) count;

			if( ready ) {
				if( MARK0 && a.x == 0xFFFFFFFFu ) {
					tornOrder++;          // the second half is there, the first is not: NOT consumed, the lane polls on
				}
				else {
					const unsigned wrong = checkRecord( a, b, r );
					bad |= wrong;
					tornWords += ( wrong != 0u ) ? 1u : 0u;
					r = b.w >> 5;
					done++;

					if( done >= steps ) {
						walking = false;
					}
					else {
						unsigned keep;
						const unsigned byteOff = r * 32u;
						// re-arm, wait for the LDS writes, request the next record
						asm volatile( "ds_write_b32 %5, %6 offset:%c7\n\tds_write_b32 %5, %8\n\ts_waitcnt lgkmcnt(0)\n\t"
						              "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4 offset:16\n\ts_mov_b32 m0, %0"
						              : "=&s"( keep ) : "v"( byteOff ), "s"( m0a ), "s"( m0b ), "s"( table ), "v"( slot ), "v"( one ), "n"( PLANE + 12 ), "v"( ones ) : "memory" );
					}
				}
			}
		}
	}

	asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
	out[blockIdx.x * blockDim.x + tid] = r;

	if( bad != 0u ) {
		atomicAdd( &stats[0], 1ull );
	}
	if( tornOrder != 0u ) {
		atomicAdd( &stats[1], (unsigned long long) tornOrder );
	}
	if( tornWords != 0u ) {
		atomicAdd( &stats[2], (unsigned long long) tornWords );
	}
	if( ( tid & 63u ) == 0u ) {
		atomicAdd( &stats[3], polls );
		atomicAdd( &stats[4], iterations );
		atomicAdd( &stats[5], lanesReady );
	}
}

#define HIP_OK( call ) do { hipError_t e__ = ( call ); if( e__ != hipSuccess ) { printf( "%s: %s\n", #call, hipGetErrorString( e__ ) ); return 1; } } while( 0 )

int main( int argc, char** argv ) {
	hipDeviceProp_t prop;
	HIP_OK( hipGetDeviceProperties( &prop, 0 ) );
	const int cus = prop.multiProcessorCount;
	printf( "%s, %d CUs\n", prop.gcnArchName, cus );
	const bool quick = ( argc > 1 && std::strcmp( argv[1], "quick" ) == 0 );

	// ---- layout ----
	{
		const int quads = 4096;
		std::vector<u4> host( quads );

		for( int i = 0; i < quads; i++ ) {
			host[i] = (u4) { (unsigned) i, 0x1000u + (unsigned) i, 0x2000u + (unsigned) i, 0x3000u + (unsigned) i };
		}

		u4* table; u4* dump;
		HIP_OK( hipMalloc( &table, sizeof( u4 ) * quads ) );
		HIP_OK( hipMemcpy( table, host.data(), sizeof( u4 ) * quads, hipMemcpyHostToDevice ) );
		const struct { unsigned m0, lo, hi; int off; size_t lds; } cases[] = {
			{ 0u, 0u, 64u, 0, 8192 }, { 1024u, 3u, 40u, 0, 8192 }, { 2048u, 0u, 64u, 16, 8192 }, { 100000u, 0u, 64u, 0, 163840 - 256 }, { 150000u, 5u, 9u, 16, 163840 - 256 },
		};

		HIP_OK( hipFuncSetAttribute( (const void*) layoutProbe, hipFuncAttributeMaxDynamicSharedMemorySize, 163840 - 256 ) );

		for( const auto& c : cases ) {
			const int dumpQuads = (int) ( c.lds / 16 );
			HIP_OK( hipMalloc( &dump, sizeof( u4 ) * dumpQuads ) );
			hipLaunchKernelGGL( layoutProbe, dim3( 1 ), dim3( 256 ), c.lds, 0, table, c.m0, c.lo, c.hi, c.off, dump, dumpQuads );
			HIP_OK( hipDeviceSynchronize() );
			std::vector<u4> got( dumpQuads );
			HIP_OK( hipMemcpy( got.data(), dump, sizeof( u4 ) * dumpQuads, hipMemcpyDeviceToHost ) );
			int first = -1, last = -1, changed = 0, asExpected = 0;

			for( int i = 0; i < dumpQuads; i++ ) {
				if( got[i].x != 0xAAAAAAAAu || got[i].y != 0xAAAAAAAAu ) {
					changed++;
					first = ( first < 0 ) ? i : first;
					last = i;
				}
			}

			// expectation: lane l writes table quad 4 l + off / 16 to LDS byte m0 + off + 16 l
			for( unsigned l = c.lo; l < c.hi; l++ ) {
				const size_t at = ( (size_t) c.m0 + (size_t) c.off + 16u * l ) / 16;
				const unsigned want = 4u * l + (unsigned) c.off / 16u;
				const bool aligned = ( ( c.m0 + (unsigned) c.off ) % 16u ) == 0u;
				if( aligned && at < (size_t) dumpQuads && got[at].x == want && got[at].y == 0x1000u + want && got[at].w == 0x3000u + want ) {
					asExpected++;
				}
			}

			printf( "layout: m0 %6u lanes [%2u,%2u) offset:%-2d  -> %d quads changed, first at byte %d (quad value %u), last at byte %d; %d of %u lanes where  m0 + offset + 16 * lane  says\n",
			        c.m0, c.lo, c.hi, c.off, changed, first * 16, ( first >= 0 ) ? got[first].x : 0u, last * 16, asExpected, c.hi - c.lo );
			HIP_OK( hipFree( dump ) );
		}

		HIP_OK( hipFree( table ) );
	}

	// ---- poll / rate ----
	const size_t farBytes = quick ? ( 256ull << 20 ) : ( 2048ull << 20 );
	const size_t records = farBytes / 32;
	const unsigned mask = (unsigned) ( records - 1 );
	std::vector<u4> host( records * 2 );
	u4* table;
	HIP_OK( hipMalloc( &table, sizeof( u4 ) * records * 2 ) );
	unsigned* out;
	const int blocks = cus;            // one 1024-thread block per CU: 4 waves / SIMD
	HIP_OK( hipMalloc( &out, sizeof( unsigned ) * blocks * 1024 ) );
	unsigned long long* stats;
	HIP_OK( hipMalloc( &stats, sizeof( unsigned long long ) * 8 ) );
	HIP_OK( hipFuncSetAttribute( (const void*) chaseAsync<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PLANE ) );
	HIP_OK( hipFuncSetAttribute( (const void*) chaseAsync<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PLANE ) );

	// share of successors that go anywhere in the table ("far"); the others stay in the first nearBytes
	const struct { const char* name; size_t nearBytes; unsigned farPercent; } mixes[] = {
		{ "all near (2 MiB: L2)", 2u << 20, 0u },
		{ "near 16 MiB + 16 % far", 16u << 20, 16u },
		{ "near 16 MiB + 40 % far", 16u << 20, 40u },
		{ "all far", 2u << 20, 100u },
	};

	for( const auto& mix : mixes ) {
		const size_t nearRecords = mix.nearBytes / 32;
		unsigned long long z = 88172645463325252ull;

		for( size_t r = 0; r < records; r++ ) {
			z ^= z << 13; z ^= z >> 7; z ^= z << 17;
			const bool far = ( ( z >> 40 ) % 100ull ) < mix.farPercent;
			const unsigned next = (unsigned) ( far ? ( ( z >> 8 ) % records ) : ( ( z >> 8 ) % nearRecords ) );
			const unsigned ru = (unsigned) r;
			host[r * 2] = (u4) { ru, ~ru, ru * 2654435761u, ru ^ 0x5bd1e995u };
			host[r * 2 + 1] = (u4) { ru + 0x9e3779b9u, ru * 3u, ru ^ 0xa5a5a5a5u, next * 32u };
		}

		HIP_OK( hipMemcpy( table, host.data(), sizeof( u4 ) * records * 2, hipMemcpyHostToDevice ) );
		const int steps = quick ? 2000 : 6000;
		hipEvent_t e0, e1;
		HIP_OK( hipEventCreate( &e0 ) );
		HIP_OK( hipEventCreate( &e1 ) );

		for( int variant = -1; variant <= 8; variant++ ) {
			// -1: synchronous; 0: polled with both markers, need 8/8; 1..8: polled, second-half marker only, need variant/8
			HIP_OK( hipMemset( stats, 0, sizeof( unsigned long long ) * 8 ) );
			float best = 1e30f;

			for( int rep = 0; rep < 2; rep++ ) {
				HIP_OK( hipEventRecord( e0 ) );

				if( variant < 0 ) {
					hipLaunchKernelGGL( chaseSync, dim3( blocks ), dim3( 1024 ), 0, 0, table, mask, steps, out, stats );
				}
				else if( variant == 0 ) {
					hipLaunchKernelGGL( chaseAsync<true>, dim3( blocks ), dim3( 1024 ), 2 * PLANE, 0, table, mask, steps, 8, out, stats );
				}
				else {
					hipLaunchKernelGGL( chaseAsync<false>, dim3( blocks ), dim3( 1024 ), 2 * PLANE, 0, table, mask, steps, variant, out, stats );
				}

				HIP_OK( hipEventRecord( e1 ) );
				HIP_OK( hipEventSynchronize( e1 ) );
				float ms = 0.0f;
				HIP_OK( hipEventElapsedTime( &ms, e0, e1 ) );
				best = ( ms < best ) ? ms : best;
			}

			unsigned long long s[8];
			HIP_OK( hipMemcpy( s, stats, sizeof( s ), hipMemcpyDeviceToHost ) );
			const double gathers = (double) blocks * 1024.0 * steps;
			if( variant < 0 ) {
				printf( "%-26s synchronous (vmcnt(0) per step)          %8.2f ms  %6.1f G records/s   lanes with a wrong record %llu\n", mix.name, best, gathers / best / 1e6, s[0] );
			}
			else {
				printf( "%-26s polled, need %d/8 %-22s %8.2f ms  %6.1f G records/s   wrong %llu  second-before-first %llu  torn %llu   polls/iteration %.2f  lanes/iteration %.1f\n",
				        mix.name, ( variant == 0 ) ? 8 : variant, ( variant == 0 ) ? "(both halves marked)" : "", best, gathers / best / 1e6, s[0], s[1], s[2],
				        (double) s[3] / (double) ( s[4] ? s[4] : 1 ), (double) s[5] / (double) ( s[4] ? s[4] : 1 ) );
			}
			fflush( stdout );
		}
	}

	return 0;
}
