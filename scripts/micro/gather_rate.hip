// Micro-benchmark (measurement aid, not product): dependent random 32-B record gathers (the BVH
// node access shape: two adjacent global_load_dwordx4, next address from the loaded data) from
// tables resident in L1 / L2 / Infinity Cache / HBM, with all 64 or only 16 / 8 lanes active.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// table: records of 2 x uint4; record.x = index of the next record (random permutation-free hash chain)
template<int LOADS>
__global__ __launch_bounds__( 1024 ) void chase( const uint4* table, unsigned mask, int steps, unsigned laneMask, unsigned* out ) {
	const unsigned lane = threadIdx.x & 63u;
	unsigned idx = ( ( blockIdx.x * blockDim.x + threadIdx.x ) * 2654435761u ) & mask;
	unsigned acc = 0;

	if( ( ( laneMask >> ( lane & 31u ) ) & 1u ) != 0u ) {
		for( int i = 0; i < steps; i++ ) {
			const uint4 a = table[(size_t) idx * 2];
			unsigned next = a.x;

			if( LOADS >= 2 ) {
				const uint4 b = table[(size_t) idx * 2 + 1];
				next ^= b.y;   // b.y == 0
			}
			if( LOADS >= 3 ) {
				const uint4 c = table[(size_t) ( idx ^ 1u ) * 2];   // neighbour record: same 64-B pair
				next ^= c.y;
			}

			acc += a.z;
			idx = ( next + lane * 7u ) & mask;
		}
	}

	out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

int main() {
	hipDeviceProp_t prop;
	(void) hipGetDeviceProperties( &prop, 0 );
	const int cus = prop.multiProcessorCount;
	printf( "%d CUs\n", cus );
	unsigned* out;
	(void) hipMalloc( &out, sizeof( unsigned ) * cus * 2 * 1024 );
	const size_t sizes[] = { 16u << 10, 2u << 20, 24u << 20, 128u << 20, 2048ull << 20 };
	const char* names[] = { "16 KiB (L1)", "2 MiB (L2)", "24 MiB (L2 x8 / MALL)", "128 MiB (MALL)", "2 GiB (HBM)" };

	for( int s = 0; s < 5; s++ ) {
		const size_t records = sizes[s] / 32;
		std::vector<uint4> host( records * 2 );
		unsigned long long z = 88172645463325252ull;

		for( size_t r = 0; r < records; r++ ) {
			z ^= z << 13; z ^= z >> 7; z ^= z << 17;
			host[r * 2] = make_uint4( (unsigned) ( z % records ), 0, 1, 0 );
			host[r * 2 + 1] = make_uint4( 0, 0, 0, 0 );
		}

		uint4* table;
		(void) hipMalloc( &table, sizes[s] );
		(void) hipMemcpy( table, host.data(), sizes[s], hipMemcpyHostToDevice );
		const unsigned mask = (unsigned) ( records - 1 );
		const unsigned laneMasks[] = { 0xFFFFFFFFu, 0x11111111u, 0x01010101u };
		const int active[] = { 64, 16, 8 };

		for( int loads = 1; loads <= 3; loads++ ) {
			for( int m = 0; m < 3; m++ ) {
				for( int wps = 2; wps <= 8; wps *= 2 ) {
					const int steps = ( s >= 3 ) ? 512 : 2048;
					const int blocks = cus * ( wps >= 4 ? wps / 4 : 1 );
					const int threads = ( wps >= 4 ) ? 1024 : 512;
					hipEvent_t e0, e1;
					(void) hipEventCreate( &e0 );
					(void) hipEventCreate( &e1 );
					float ms = 0;
					#define LAUNCH( L, ST ) chase<L><<<blocks, threads>>>( table, mask, ST, laneMasks[m], out )
					if( loads == 1 ) { LAUNCH( 1, 64 ); (void) hipEventRecord( e0 ); LAUNCH( 1, steps ); }
					else if( loads == 2 ) { LAUNCH( 2, 64 ); (void) hipEventRecord( e0 ); LAUNCH( 2, steps ); }
					else { LAUNCH( 3, 64 ); (void) hipEventRecord( e0 ); LAUNCH( 3, steps ); }
					(void) hipEventRecord( e1 );
					(void) hipDeviceSynchronize();
					(void) hipEventElapsedTime( &ms, e0, e1 );
					const double waveSteps = (double) steps * wps * 4;   // per CU
					const double cyc = ms * 1e-3 * 2.4e9;
					printf( "%-22s loads/step %d  active lanes %2d  waves/SIMD %d: %8.3f ms  %7.1f cyc per wave-step per CU  %7.1f G records/s chip  (latency-equivalent %6.0f cyc/step/wave)\n",
						names[s], loads, active[m], wps, ms, cyc / waveSteps, (double) steps * wps * 4 * cus * active[m] / ms / 1e6, cyc / steps );
				}
			}
		}

		(void) hipFree( table );
	}
	return 0;
}
