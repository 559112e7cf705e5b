// Round-4 lab variants (NOT part of the product build: included by csrc/pt_kernel.hpp only under -DPBR_LAB, scripts/lab.sh).
// diagTraceStreamDual: traversal-only probe with one or two walks per lane.
// Measured in profiles/r04/experiments/; DESIGN.md section 5.1e says what each was for and why the product does not use it.

#if defined( PBR_LAB ) && defined( PT_NODE_PHASE_ASM )
// Traversal-only probe, one or two walks per lane, at MINW waves / SIMD (round 4, lab).  ONE walk: traverse() as the
// lock-step kernels run it.  TWO: every lane draws two rays and walks both with nodePhaseDual; a lane's parked walks take
// their leaf tests one after the other.  Same rays, same (t, face) per ray, same visit and face-test counts.
template<int MINW, bool DUAL>
__global__ __launch_bounds__( PBR_BLOCK, MINW ) void diagTraceStreamDual( const DevParams P, const float4* rays, unsigned n, float2* out ) {
	const float4* lds = gHotNodes;
	stageHotNodes( P, gHotNodes );
	unsigned nodes = 0, tris = 0;

	if( !DUAL ) {
		unsigned i = atomicAdd( P.workCounter, 1u );

		while( i < n ) {
			const float4 a = rays[(size_t) i * 2 + 0];
			const float4 b = rays[(size_t) i * 2 + 1];
			Ray ray;
			ray.origin = mk3( a.x, a.y, a.z );
			ray.dir = mk3( b.x, b.y, b.z );
			Hit hit;
			hit.t = inff();
			hit.face = 0;
			traverse<false, false, true, false, ( MINW <= 4 )>( P, lds, ray, hit, nodes, tris );
			out[i] = make_float2( hit.t, __int_as_float( hit.face ) );
			i = atomicAdd( P.workCounter, 1u );
		}
	}
	else {
		unsigned i = atomicAdd( P.workCounter, 2u );

		while( i < n ) {
			const bool haveB = ( i + 1u < n );
			const unsigned j = haveB ? i + 1u : i;
			const float4 a0 = rays[(size_t) i * 2 + 0], a1 = rays[(size_t) i * 2 + 1];
			const float4 b0 = rays[(size_t) j * 2 + 0], b1 = rays[(size_t) j * 2 + 1];
			Ray rayA, rayB;
			rayA.origin = mk3( a0.x, a0.y, a0.z );
			rayA.dir = mk3( a1.x, a1.y, a1.z );
			rayB.origin = mk3( b0.x, b0.y, b0.z );
			rayB.dir = mk3( b1.x, b1.y, b1.z );
			Hit hitA, hitB;
			hitA.t = hitB.t = inff();
			hitA.face = hitB.face = 0;
			const f3 invA = mk3( 1.0f / rayA.dir.x, 1.0f / rayA.dir.y, 1.0f / rayA.dir.z );
			const f3 invB = mk3( 1.0f / rayB.dir.x, 1.0f / rayB.dir.y, 1.0f / rayB.dir.z );
			const f2v oxyA = { rayA.origin.x, rayA.origin.y }, ozzA = { rayA.origin.z, rayA.origin.z }, ixyA = { invA.x, invA.y }, izzA = { invA.z, invA.z };
			const f2v oxyB = { rayB.origin.x, rayB.origin.y }, ozzB = { rayB.origin.z, rayB.origin.z }, ixyB = { invB.x, invB.y }, izzB = { invB.z, invB.z };
			int refA = P.firstRef, refB = haveB ? P.firstRef : -1;
			unsigned visits = 0, visitsB = 0;

			for( ;; ) {
				int leafWordA = 0, leafWordB = 0;
				float tNearA = 0.0f, tNearB = 0.0f;

				if( refA >= 0 || refB >= 0 ) {
					const int entered = __popcll( __ballot( refA >= 0 ) ) + __popcll( __ballot( refB >= 0 ) );
					const int leave = ( entered * P.parkEighths ) >> 3;
					const int keep = entered - ( ( leave < 1 ) ? 1 : leave );
					__builtin_amdgcn_s_setprio( PT_WALK_PRIO );
					nodePhaseDual( P, oxyA, ozzA, ixyA, izzA, hitA.t, oxyB, ozzB, ixyB, izzB, hitB.t, keep, refA, refB, visits, visitsB,
					               leafWordA, tNearA, leafWordB, tNearB );
					__builtin_amdgcn_s_setprio( 0 );
				}

				if( leafWordA != 0 ) {
					testLeaf<false, ( MINW <= 4 )>( P, leafFace0( leafWordA ), leafFace1( leafWordA ), rayA, tNearA, 0.0f, hitA, tris );
				}
				if( leafWordB != 0 ) {
					testLeaf<false, ( MINW <= 4 )>( P, leafFace0( leafWordB ), leafFace1( leafWordB ), rayB, tNearB, 0.0f, hitB, tris );
				}

				if( __ballot( refA >= 0 || refB >= 0 ) == 0ull ) {
					break;
				}
			}

			nodes += visits + visitsB;
			out[i] = make_float2( hitA.t, __int_as_float( hitA.face ) );

			if( haveB ) {
				out[j] = make_float2( hitB.t, __int_as_float( hitB.face ) );
			}

			i = atomicAdd( P.workCounter, 2u );
		}
	}

	atomicAdd( &P.counters[0], (unsigned long long) nodes );
	atomicAdd( &P.counters[1], (unsigned long long) tris );
}
#endif

