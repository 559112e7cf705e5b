// Micro-benchmark (measurement aid, not product): does the L1 / TA charge a divergent global_load_dwordx4 per LANE or per
// distinct 64-byte segment?  Every lane needs one random 32-byte record per step (dependent chain, like a BVH walk):
//   own     each lane loads its record's two halves itself: 2 instructions x 64 distinct segments   (what the node phase does)
//   paired  lanes 2k / 2k+1 load the two halves of lane 2k's record in ONE instruction (both addresses in one 64-byte
//           segment), then the halves of lane 2k+1's record; the halves change lanes with two quad-perm swaps
// Same records, same chain, same checksum.   hipcc --offload-arch=gfx950 -O3 -o pair_coalesce pair_coalesce.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template<bool PAIRED>
__global__ __launch_bounds__( 1024 ) void chase( const uint4* table, unsigned mask, int steps, unsigned* out ) {
	const unsigned lane = threadIdx.x & 63u;
	unsigned idx = ( ( blockIdx.x * blockDim.x + threadIdx.x ) * 2654435761u ) & mask;
	unsigned acc = 0;

	for( int i = 0; i < steps; i++ ) {
		uint4 a, b;

		if( !PAIRED ) {
			a = table[(size_t) idx * 2];
			b = table[(size_t) idx * 2 + 1];
		}
		else {
			const unsigned partner = (unsigned) __shfl_xor( (int) idx, 1 );
			const unsigned even = ( lane & 1u ) ? partner : idx;      // the pair's even lane's record
			const unsigned odd = ( lane & 1u ) ? idx : partner;
			const uint4 x = table[(size_t) even * 2 + ( lane & 1u )];   // even lane: first half, odd lane: second half
			const uint4 y = table[(size_t) odd * 2 + ( lane & 1u )];
			uint4 xs, ys;
			xs.x = (unsigned) __shfl_xor( (int) x.x, 1 ); xs.y = (unsigned) __shfl_xor( (int) x.y, 1 ); xs.z = (unsigned) __shfl_xor( (int) x.z, 1 ); xs.w = (unsigned) __shfl_xor( (int) x.w, 1 );
			ys.x = (unsigned) __shfl_xor( (int) y.x, 1 ); ys.y = (unsigned) __shfl_xor( (int) y.y, 1 ); ys.z = (unsigned) __shfl_xor( (int) y.z, 1 ); ys.w = (unsigned) __shfl_xor( (int) y.w, 1 );
			a = ( lane & 1u ) ? ys : x;       // own record's first half
			b = ( lane & 1u ) ? y : xs;       // own record's second half
		}

		acc += a.z + b.w;
		idx = ( ( a.x ^ b.y ) + lane * 7u ) & mask;
	}

	out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

int main() {
	hipDeviceProp_t prop;
	(void) hipGetDeviceProperties( &prop, 0 );
	const int cus = prop.multiProcessorCount;
	unsigned* out;
	(void) hipMalloc( &out, sizeof( unsigned ) * cus * 2 * 1024 );
	std::vector<unsigned> h0( cus * 2 * 1024 ), h1( cus * 2 * 1024 );
	const size_t sizes[] = { 2u << 20, 24u << 20, 128u << 20, 2048ull << 20 };
	const char* names[] = { "2 MiB (L2)", "24 MiB (L2 x 8 / MALL)", "128 MiB (MALL)", "2 GiB (HBM)" };

	for( int s = 0; s < 4; s++ ) {
		const size_t records = sizes[s] / 32;
		std::vector<uint4> host( records * 2 );
		unsigned long long z = 88172645463325252ull;

		for( size_t r = 0; r < records; r++ ) {
			z ^= z << 13; z ^= z >> 7; z ^= z << 17;
			host[r * 2] = make_uint4( (unsigned) ( z % records ), 0, 1, 0 );
			host[r * 2 + 1] = make_uint4( 0, 0, 0, 2 );
		}

		uint4* table;
		(void) hipMalloc( &table, sizes[s] );
		(void) hipMemcpy( table, host.data(), sizes[s], hipMemcpyHostToDevice );
		const unsigned mask = (unsigned) ( records - 1 );

		for( int wps = 4; wps <= 8; wps *= 2 ) {
			const int steps = ( s >= 2 ) ? 512 : 2048;
			const int blocks = cus * wps / 4;
			float ms[2];

			for( int v = 0; v < 2; v++ ) {
				hipEvent_t e0, e1;
				(void) hipEventCreate( &e0 );
				(void) hipEventCreate( &e1 );
				if( v == 0 ) { chase<false><<<blocks, 1024>>>( table, mask, 64, out ); (void) hipEventRecord( e0 ); chase<false><<<blocks, 1024>>>( table, mask, steps, out ); }
				else { chase<true><<<blocks, 1024>>>( table, mask, 64, out ); (void) hipEventRecord( e0 ); chase<true><<<blocks, 1024>>>( table, mask, steps, out ); }
				(void) hipEventRecord( e1 );
				(void) hipDeviceSynchronize();
				(void) hipEventElapsedTime( &ms[v], e0, e1 );
				(void) hipMemcpy( v == 0 ? h0.data() : h1.data(), out, sizeof( unsigned ) * blocks * 1024, hipMemcpyDeviceToHost );
			}

			bool same = true;
			for( int k = 0; k < blocks * 1024; k++ ) { same = same && ( h0[k] == h1[k] ); }
			const double waveSteps = (double) steps * wps * 4;
			printf( "%-24s %d waves/SIMD: own %8.3f ms (%6.1f cyc per wave-step per CU, %6.1f G records/s)   paired %8.3f ms (%6.1f cyc, %6.1f G records/s)   %s\n",
			        names[s], wps, ms[0], ms[0] * 1e-3 * 2.4e9 / waveSteps, (double) steps * wps * 4 * cus * 64 / ms[0] / 1e6,
			        ms[1], ms[1] * 1e-3 * 2.4e9 / waveSteps, (double) steps * wps * 4 * cus * 64 / ms[1] / 1e6, same ? "same results" : "DIFFERENT RESULTS" );
		}

		(void) hipFree( table );
	}

	return 0;
}
