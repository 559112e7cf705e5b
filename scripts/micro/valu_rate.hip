// Micro-benchmark (measurement aid, not product): issue rate of plain vs packed f32 VALU, min3 and
// LDS b128 reads at 1, 2, 4 and 8 waves per SIMD on gfx950.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8( X ) X X X X X X X X

template<int MODE>
__global__ __launch_bounds__( 1024 ) void rate( float* out, int iters ) {
	extern __shared__ float4 lds[];
	float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
	typedef float f2 __attribute__( ( ext_vector_type( 2 ) ) );
	f2 p0 = { a0, 1 }, p1 = { 1, 2 }, p2 = { 2, 3 }, p3 = { 3, 4 }, p4 = { 4, 5 }, p5 = { 5, 6 }, p6 = { 6, 7 }, p7 = { 7, 8 };
	unsigned addr = ( threadIdx.x * 2654435761u ) & 0x7FF0u;
	double d0 = 1e-3 * threadIdx.x, d1 = 0.1, d2 = 0.2, d3 = 0.3, d4 = 0.4, d5 = 0.5, d6 = 0.6, d7 = 0.7;

	if( MODE == 4 ) {
		for( int i = threadIdx.x; i < 2048; i += blockDim.x ) {
			lds[i] = make_float4( i, 0, 0, 0 );
		}
		__syncthreads();
	}

	for( int i = 0; i < iters; i++ ) {
		if( MODE == 0 ) {
			REP8( asm volatile( "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
				: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ), "+v"( a5 ), "+v"( a6 ), "+v"( a7 ) ); )
		}
		else if( MODE == 12 ) {
			// binary64 FMA: what pow1's chains are made of (round 3: is a double-float pow worth building?)
			REP8( asm volatile( "v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7"
				: "+v"( d0 ), "+v"( d1 ), "+v"( d2 ), "+v"( d3 ), "+v"( d4 ), "+v"( d5 ), "+v"( d6 ), "+v"( d7 ) ); )
		}
		else if( MODE == 13 ) {
			REP8( asm volatile( "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
				: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ), "+v"( a5 ), "+v"( a6 ), "+v"( a7 ) ); )
		}
		else if( MODE == 9 || MODE == 10 || MODE == 11 ) {
			// the same 64 v_add_f32 with part of EXEC switched off: does the SIMD skip an idle half (quarter) of a wave?
			const unsigned long long mask = ( MODE == 9 ) ? 0x00000000FFFFFFFFull : ( MODE == 10 ) ? 0x000000000000FFFFull : 0x0000FFFF0000FFFFull;
			unsigned long long saved;
			asm volatile( "s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=&s"( saved ) : "s"( mask ) );
			REP8( asm volatile( "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
				: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ), "+v"( a5 ), "+v"( a6 ), "+v"( a7 ) ); )
			asm volatile( "s_mov_b64 exec, %0" :: "s"( saved ) );
		}
		else if( MODE == 1 ) {
			REP8( asm volatile( "v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n v_pk_add_f32 %4, %4, %4\n v_pk_add_f32 %5, %5, %5\n v_pk_add_f32 %6, %6, %6\n v_pk_add_f32 %7, %7, %7"
				: "+v"( p0 ), "+v"( p1 ), "+v"( p2 ), "+v"( p3 ), "+v"( p4 ), "+v"( p5 ), "+v"( p6 ), "+v"( p7 ) ); )
		}
		else if( MODE == 2 ) {
			REP8( asm volatile( "v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %4\n v_min3_f32 %3, %3, %4, %5\n v_min3_f32 %4, %4, %5, %6\n v_min3_f32 %5, %5, %6, %7\n v_min3_f32 %6, %6, %7, %0\n v_min3_f32 %7, %7, %0, %1"
				: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ), "+v"( a5 ), "+v"( a6 ), "+v"( a7 ) ); )
		}
		else if( MODE == 3 ) {
			REP8( asm volatile( "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
				: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ), "+v"( a5 ), "+v"( a6 ), "+v"( a7 ) :: "vcc" ); )
		}
		else if( MODE == 6 ) {
			// 64 scalar ALU instructions on 8 independent chains
			unsigned s0 = i, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
			REP8( asm volatile( "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %4\n s_and_b32 %4, %4, %5\n s_or_b32 %5, %5, %6\n s_add_u32 %6, %6, %7\n s_add_u32 %7, %7, %0"
				: "+s"( s0 ), "+s"( s1 ), "+s"( s2 ), "+s"( s3 ), "+s"( s4 ), "+s"( s5 ), "+s"( s6 ), "+s"( s7 ) :: "scc" ); )
			a1 += (float) s7;
		}
		else if( MODE == 7 ) {
			// 32 VALU + 32 SALU interleaved: do they overlap?
			unsigned s0 = i, s1 = 1, s2 = 2, s3 = 3;
			REP8( asm volatile( "v_add_f32 %0, %0, %0\n s_add_u32 %4, %4, %5\n v_add_f32 %1, %1, %1\n s_add_u32 %5, %5, %6\n v_add_f32 %2, %2, %2\n s_add_u32 %6, %6, %7\n v_add_f32 %3, %3, %3\n s_add_u32 %7, %7, %4"
				: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+s"( s0 ), "+s"( s1 ), "+s"( s2 ), "+s"( s3 ) :: "scc" ); )
			a5 += (float) s3;
		}
		else if( MODE == 8 ) {
			// 64-bit mask ops as the compiler emits them for divergent control flow
			unsigned long long m0 = i, m1 = 1, m2 = 2, m3 = 3;
			REP8( asm volatile( "s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %2\n s_andn2_b64 %2, %2, %3\n s_xor_b64 %3, %3, %0\n s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %2\n s_andn2_b64 %2, %2, %3\n s_xor_b64 %3, %3, %0"
				: "+s"( m0 ), "+s"( m1 ), "+s"( m2 ), "+s"( m3 ) :: "scc" ); )
			a1 += (float) ( m3 & 1 );
		}
		else if( MODE == 4 ) {
			// 8 random 16-B LDS reads per lane, address chain through the loaded value's zero fields
			float4 v;
			REP8( asm volatile( "ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"( v ) : "v"( addr ) : "memory" );
			      addr = ( addr * 1664525u + 1013904223u + __float_as_uint( v.y ) ) & 0x7FF0u; )
			a1 += v.x;
		}
		else if( MODE == 5 ) {
			// 8 independent random LDS reads in flight
			float4 v0, v1, v2, v3;
			unsigned b = addr;
			asm volatile( "ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %6\n ds_read_b128 %3, %7\n s_waitcnt lgkmcnt(0)"
				: "=v"( v0 ), "=v"( v1 ), "=v"( v2 ), "=v"( v3 ) : "v"( b ), "v"( b ^ 0x1230u ), "v"( b ^ 0x4560u ), "v"( b ^ 0x7890u & 0x7FF0u ) : "memory" );
			asm volatile( "ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %6\n ds_read_b128 %3, %7\n s_waitcnt lgkmcnt(0)"
				: "=v"( v0 ), "=v"( v1 ), "=v"( v2 ), "=v"( v3 ) : "v"( b ^ 0x10u ), "v"( b ^ 0x2340u ), "v"( b ^ 0x5670u ), "v"( b ^ 0x0ab0u ) : "memory" );
			addr = ( addr * 1664525u + 1013904223u ) & 0x7FF0u;
			a1 += v0.x + v1.x + v2.x + v3.x;
		}
	}

	out[blockIdx.x * blockDim.x + threadIdx.x] = (float) ( d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 ) + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template<int MODE>
void run( const char* name, int cus, float* out ) {
	const int iters = 4096;

	for( int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2 ) {
		// blocks of 256 threads = 1 wave per SIMD; blocks per CU = wavesPerSimd
		const int threads = ( wavesPerSimd >= 4 ) ? 1024 : 256 * wavesPerSimd;
		const int blocks = cus * ( wavesPerSimd >= 4 ? wavesPerSimd / 4 : 1 );
		hipEvent_t e0, e1;
		hipEventCreate( &e0 );
		hipEventCreate( &e1 );
		rate<MODE><<<blocks, threads, 32768>>>( out, 16 );
		hipEventRecord( e0 );
		rate<MODE><<<blocks, threads, 32768>>>( out, iters );
		hipEventRecord( e1 );
		hipDeviceSynchronize();
		float ms;
		hipEventElapsedTime( &ms, e0, e1 );
		const double instrPerWave = 64.0 * iters;
		const double nsPerInstrPerSimd = ms * 1e6 / ( instrPerWave * wavesPerSimd );
		printf( "%-28s waves/SIMD %d: %8.3f ms  -> %.3f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz)\n", name, wavesPerSimd, ms, nsPerInstrPerSimd, nsPerInstrPerSimd * 2.4 );
	}
}

int main() {
	hipDeviceProp_t prop;
	hipGetDeviceProperties( &prop, 0 );
	const int cus = prop.multiProcessorCount;
	printf( "%s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate );
	float* out;
	hipMalloc( &out, sizeof( float ) * cus * 2 * 1024 );
	run<0>( "v_add_f32", cus, out );
	run<9>( "v_add_f32, EXEC = low 32 lanes", cus, out );
	run<10>( "v_add_f32, EXEC = low 16 lanes", cus, out );
	run<11>( "v_add_f32, EXEC = 16 + 16 lanes", cus, out );
	run<13>( "v_fma_f32", cus, out );
	run<12>( "v_fma_f64", cus, out );
	run<1>( "v_pk_add_f32", cus, out );
	run<2>( "v_min3_f32", cus, out );
	run<3>( "v_cndmask/v_mov", cus, out );
	run<6>( "s_add/s_and (64 SALU)", cus, out );
	run<7>( "32 v_add + 32 s_add interleaved", cus, out );
	run<8>( "s_and_b64 family (64 SALU)", cus, out );
	run<4>( "ds_read_b128 random dependent", cus, out );
	run<5>( "ds_read_b128 random x8 indep", cus, out );
	return 0;
}
