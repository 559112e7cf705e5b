// The denoise half of the display step: a separate pass behind the path, off the hot path.
//
// The reference never finished its noise filter: source/opencl/noise_filtering.cl:441-467 declares a kernel fed by
// per-pixel feature buffers — position and normal of the first and second intersection, texture colour of the first —
// that runs four filter passes over shrinking blocks (55, 35, 17, 7 pixels), but the weights are TODOs
// (computeFeatureWeights, :386-401), the colour filter is a TODO (:417) and the host never builds or launches it
// (PathTracer.cpp:155-160).  There is no behaviour to reproduce, so this is NOT a parity component; it keeps the
// sketch's shape — first-hit feature buffers, several passes, feature distances scaled by standard deviations
// (STD_DEVIATION_FEATURE / _WORLD, :6-7) — and fills the TODOs with the edge-avoiding a-trous wavelet filter
// (Dammertz et al. 2010): per pass a 5 x 5 B3-spline stencil whose taps are 1, 2, 4, ... pixels apart, every tap
// weighted by exp( - feature distances ), five passes spanning 61 pixels (the sketch's widest block: 55).
//
// Features come from one extra primary ray through every pixel centre (`firstHitFeatures`: the walk of the path
// kernels, `traverse`, on the same node stream), not from the path kernels themselves: a feature store per sample would
// put 48 B per sample and three pointers into kernels that are sized to the register.  Orb lights are
// not in the feature pass (a visible orb carries the features of what lies behind it).
//
// Floating point, no reference: the test (tests/test_gpu_denoise.py) restates the filter in numpy fp32 and states its
// tolerance; properties (a constant image is a fixed point, edges of the features are not crossed, variance drops on a
// flat wall) are checked on rendered frames.
#pragma once

#include "pt_kernel.hpp"

namespace ptd {

using namespace ptk;

struct DenoiseArgs {
	int width, height, step;
	float invColor;     // 1 / sigma_color_k^2 of this pass, 0 = off
	float invNormal;    // 1 / sigma_normal^2, 0 = off
	float invAlbedo;    // 1 / sigma_albedo^2, 0 = off
	float worldScale;   // sigma_world * step * pxDim: times the centre's distance = the standard deviation in world units; 0 = off
};

// the ray through the centre of pixel (px, py): initRay (pathtracing.cl:25-48) without the jitter and the lens
PT_DEV Ray centreRay( const DevParams& P, int px, int py ) {
	const f3 cu = ld3( P.cu );
	const f3 cv = ld3( P.cv );
	f3 inner = ld3( P.camA );
	inner = inner + cu * ( 2.0f * (float) px );
	inner = inner + cv;
	inner = inner - ld3( P.cvH );
	inner = inner + cv * ( 2.0f * (float) py );
	Ray ray;
	ray.origin = ld3( P.eye );
	ray.dir = normalize( ld3( P.cw ) + inner * P.halfPx );
	return ray;
}

// position {x, y, z, t}, normal {x, y, z, hit ? 1 : 0} (turned towards the viewer: faces are two-sided), albedo {Kd, material}
__global__ void firstHitFeatures( const DevParams P, float4* position, float4* normal, float4* albedo ) {
	const int x = (int) ( blockIdx.x * blockDim.x + threadIdx.x );
	const int y = (int) ( blockIdx.y * blockDim.y + threadIdx.y );

	if( x >= P.width || y >= P.height ) {
		return;
	}

	const Ray ray = centreRay( P, x, y );
	Hit hit;
	hit.t = inff();
	hit.face = 0;
	unsigned nodes = 0, tris = 0;
	traverse<false, false, false>( P, nullptr, ray, hit, nodes, tris );
	const size_t at = (size_t) y * (size_t) P.width + (size_t) x;

	if( hit.t == inff() ) {
		position[at] = make_float4( 0.0f, 0.0f, 0.0f, inff() );
		normal[at] = make_float4( 0.0f, 0.0f, 0.0f, 0.0f );
		albedo[at] = make_float4( 0.0f, 0.0f, 0.0f, -1.0f );
		return;
	}

	int material = 0;
	f3 n = faceNormal( P, hit.face, &material );

	if( dot( n, ray.dir ) > 0.0f ) {
		n = n * -1.0f;
	}

	const f3 p = fma3( hit.t, ray.dir, ray.origin );
	const float4 kd = P.mats[material * 4 + 2];
	position[at] = make_float4( p.x, p.y, p.z, hit.t );
	normal[at] = make_float4( n.x, n.y, n.z, 1.0f );
	albedo[at] = make_float4( kd.x, kd.y, kd.z, (float) material );
}

__device__ __forceinline__ float squaredDistance3( float4 a, float4 b ) {
	const float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
	return ( x * x + y * y ) + z * z;
}

// one a-trous pass: 5 x 5 taps `step` pixels apart, B3-spline weights times exp( -( colour + normal + position + albedo
// distances, each over its variance ) ); taps outside the image and taps on the other side of the hit / miss divide
// are left out; .w (the first-hit distance of the accumulated image) passes through
__global__ void atrousPass( const DenoiseArgs A, const float4* in, float4* out, const float4* position, const float4* normal, const float4* albedo ) {
	const int x = (int) ( blockIdx.x * blockDim.x + threadIdx.x );
	const int y = (int) ( blockIdx.y * blockDim.y + threadIdx.y );

	if( x >= A.width || y >= A.height ) {
		return;
	}

	const float spline[5] = { 0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f };
	const size_t at = (size_t) y * (size_t) A.width + (size_t) x;
	const float4 c0 = in[at], p0 = position[at], n0 = normal[at], a0 = albedo[at];
	const float sigmaWorld = A.worldScale * p0.w;
	const float invWorld = ( n0.w != 0.0f && sigmaWorld > 0.0f ) ? 1.0f / ( sigmaWorld * sigmaWorld ) : 0.0f;
	float sumR = 0.0f, sumG = 0.0f, sumB = 0.0f, sumW = 0.0f;

	for( int j = -2; j <= 2; j++ ) {
		const int ty = y + j * A.step;

		if( ty < 0 || ty >= A.height ) {
			continue;
		}

		for( int i = -2; i <= 2; i++ ) {
			const int tx = x + i * A.step;

			if( tx < 0 || tx >= A.width ) {
				continue;
			}

			const size_t tap = (size_t) ty * (size_t) A.width + (size_t) tx;
			const float4 n = normal[tap];

			if( n.w != n0.w ) {
				continue;
			}

			const float4 c = in[tap];
			float e = squaredDistance3( c, c0 ) * A.invColor;

			if( n0.w != 0.0f ) {
				e += squaredDistance3( n, n0 ) * A.invNormal;
				e += squaredDistance3( position[tap], p0 ) * invWorld;
				e += squaredDistance3( albedo[tap], a0 ) * A.invAlbedo;
			}

			if( !( e < inff() ) ) {
				continue;   // a tap that is not finite (or infinitely far in some feature) has no say
			}

			const float w = ( spline[i + 2] * spline[j + 2] ) * expf( -e );
			sumR += w * c.x;
			sumG += w * c.y;
			sumB += w * c.z;
			sumW += w;
		}
	}

	// the centre tap has e = 0, so sumW >= 9 / 64 — unless the centre colour is not finite, which then stays what it is
	const bool usable = sumW > 0.0f && sumW < inff();
	out[at] = usable ? make_float4( sumR / sumW, sumG / sumW, sumB / sumW, c0.w ) : c0;
}

}  // namespace ptd
