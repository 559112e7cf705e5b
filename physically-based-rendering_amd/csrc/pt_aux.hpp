// The kernels AROUND the path-tracing launch: the fold of a multi-frame render, layout movers, the display step and the
// diagnostic kernels behind include/pbr_hip_diag.h.  Included by pbr_hip.hip only — the path-tracing kernels themselves
// (pt_kernel.hpp, pt_dual.hpp) are instantiated in their own translation units, one per plan and build flavour
// (pt_instances.hpp); these are not templates over a flavour and exist once.
#pragma once

#include "pt_kernel.hpp"

namespace ptk {

// ---------------------------------------------------------------------------------------
// Frame-parallel launches: fold the frames into the running mean
// ---------------------------------------------------------------------------------------
// setColors (pt_rgb.cl:9-21) for frames firstCount .. firstCount + nFrames - 1 of every local pixel,
// in frame order: exactly the arithmetic shadeStep applies when one lane walks a pixel through
// all its frames.  imageOut.w = focus (first-hit distance) of the last frame.
__global__ __launch_bounds__( 256 ) void foldFrames( const DevParams P, const float4* src, float4* dst ) {
	const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;

	// the path-tracing launch before this one has drained the queue: leave its heads (and the word of the heads seen empty behind
	// them) at zero for the next launch
	if( slot <= (unsigned) PT_HEADS ) {
		P.workCounter[slot * PT_BAND_STRIDE] = 0u;
	}

	if( slot >= P.frameStride ) {
		return;
	}

	float4 acc = src[slot];

	for( int k = 0; k < P.nFrames; k++ ) {
		const float4 fc = P.frameBuf[frameBufIndex( P, slot, (unsigned) k )];
		const unsigned n = (unsigned) ( P.firstCount + k );
		const float w = P.useExplicitWeight ? P.explicitWeight : ( (float) n / (float) ( n + 1u ) );
		acc.x = fc.x + ( acc.x - fc.x ) * w;
		acc.y = fc.y + ( acc.y - fc.y ) * w;
		acc.z = fc.z + ( acc.z - fc.z ) * w;
		acc.w = fc.w;
	}

	dst[slot] = acc;
}

// Node visits per local tile: the sum of the debug image's node counts (K18, pathtracing.cl:73-78: visits / 1265) over a
// tile's 64 pixels — the cost the queue's dealing order is built from (pbr_hip.hip, learnTileCosts).  One wave per tile.
__global__ __launch_bounds__( 256 ) void tileCosts( const float4* dbg, float* cost, unsigned numTiles ) {
	const unsigned tile = blockIdx.x * 4u + ( threadIdx.x >> 6 );

	if( tile >= numTiles ) {
		return;
	}

	float v = dbg[(size_t) tile * 64u + ( threadIdx.x & 63u )].y * 1265.0f;

	for( int step = 32; step >= 1; step >>= 1 ) {
		v += __shfl_xor( v, step, 64 );
	}

	if( ( threadIdx.x & 63u ) == 0u ) {
		cost[tile] = v;
	}
}

// ---------------------------------------------------------------------------------------
// Scene preparation (pbr_upload_scene): per-face and per-material values the shading would otherwise
// recompute on every hit — evaluated here by the same device functions, so the bits are the same
// ---------------------------------------------------------------------------------------
__global__ void prepareFaceNormals( DevParams P, float4* faceN, int numFaces ) {
	const int face = (int) ( blockIdx.x * blockDim.x + threadIdx.x );

	if( face >= numFaces ) {
		return;
	}

	int material;
	const f3 n = faceNormal<false>( P, face, &material );
	faceN[face] = make_float4( n.x, n.y, n.z, __int_as_float( material ) );
}

// ---------------------------------------------------------------------------------------
// Framebuffer layout helpers
// ---------------------------------------------------------------------------------------

// tile-major (local tiles of this rank) -> row-major W x H; pixels of other ranks' tiles = 0
__global__ void untile( const float4* tiles, float4* rows, int width, int height, int tilesX, int tileWorld, int tileRank ) {
	const int x = (int) ( blockIdx.x * blockDim.x + threadIdx.x );
	const int y = (int) ( blockIdx.y * blockDim.y + threadIdx.y );

	if( x >= width || y >= height ) {
		return;
	}

	const int tileGlobal = ( y >> 3 ) * tilesX + ( x >> 3 );
	float4 v = make_float4( 0.0f, 0.0f, 0.0f, 0.0f );

	const int position = dealPositionOfTile( tileGlobal, tilesX, tileWorld );

	if( position % tileWorld == tileRank ) {
		const int tileLocal = position / tileWorld;
		v = tiles[(size_t) tileLocal * 64 + (size_t) ( ( y & 7 ) * 8 + ( x & 7 ) )];
	}

	rows[(size_t) y * (size_t) width + (size_t) x] = v;
}

// The display step after the path: what shader/pathtracing.frag:11-15 puts on an
// 8-bit GL framebuffer — the linear colour, clamped to [0, 1], alpha 1 — as RGBA8, value = floor( c * 255 + 0.5 ).
// rowStep = +1: row 0 is the bottom of the image (GL, like pbr_read_output); -1: top row first (image files).
__global__ void displayRGBA8( const float4* tiles, uchar4* rows, int width, int height, int tilesX, int tileWorld, int tileRank, int topRowFirst ) {
	const int x = (int) ( blockIdx.x * blockDim.x + threadIdx.x );
	const int y = (int) ( blockIdx.y * blockDim.y + threadIdx.y );

	if( x >= width || y >= height ) {
		return;
	}

	const int tileGlobal = ( y >> 3 ) * tilesX + ( x >> 3 );
	float4 v = make_float4( 0.0f, 0.0f, 0.0f, 0.0f );

	const int position = dealPositionOfTile( tileGlobal, tilesX, tileWorld );

	if( position % tileWorld == tileRank ) {
		const int tileLocal = position / tileWorld;
		v = tiles[(size_t) tileLocal * 64 + (size_t) ( ( y & 7 ) * 8 + ( x & 7 ) )];
	}

	// fmax / fmin drop a NaN operand: NaN -> 0
	const float r = fmin1( fmax1( v.x, 0.0f ), 1.0f );
	const float g = fmin1( fmax1( v.y, 0.0f ), 1.0f );
	const float b = fmin1( fmax1( v.z, 0.0f ), 1.0f );
	const int outRow = topRowFirst ? ( height - 1 - y ) : y;
	rows[(size_t) outRow * (size_t) width + (size_t) x] = make_uchar4(
		(unsigned char) (int) __builtin_floorf( r * 255.0f + 0.5f ),
		(unsigned char) (int) __builtin_floorf( g * 255.0f + 0.5f ),
		(unsigned char) (int) __builtin_floorf( b * 255.0f + 0.5f ),
		255 );
}

// row-major W x H -> tile-major local tiles
__global__ void retile( const float4* rows, float4* tiles, int width, int numLocalTiles, int tilesX, int tileWorld, int tileRank ) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= (size_t) numLocalTiles * 64 ) {
		return;
	}

	const int tileLocal = (int) ( i >> 6 );
	const int lane = (int) ( i & 63 );
	const int tileGlobal = tileAtDealPosition( tileLocal * tileWorld + tileRank, tilesX, tileWorld );
	const int x = ( tileGlobal % tilesX ) * 8 + ( lane & 7 );
	const int y = ( tileGlobal / tilesX ) * 8 + ( lane >> 3 );
	tiles[i] = rows[(size_t) y * (size_t) width + (size_t) x];
}

// all-gather layout (tileWorld rank buffers of `perRank` tiles each) -> this context's full tile-major image
__global__ void scatterGathered( const float4* all, float4* tiles, int numTiles, int perRank, int tileWorld, int tilesX ) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= (size_t) numTiles * 64 ) {
		return;
	}

	const int tileGlobal = (int) ( i >> 6 );
	const int lane = (int) ( i & 63 );
	const int position = dealPositionOfTile( tileGlobal, tilesX, tileWorld );
	const int rank = position % tileWorld;
	const int local = position / tileWorld;
	tiles[i] = all[( (size_t) rank * perRank + local ) * 64 + lane];
}



// ---------------------------------------------------------------------------------------
// Diagnostic kernels (include/pbr_hip_diag.h): one thread per item, for stage-by-stage parity
// ---------------------------------------------------------------------------------------

__global__ void diagMath( int op, const float* x, const float* y, int n, float* out ) {
	const int i = (int) ( blockIdx.x * blockDim.x + threadIdx.x );

	if( i >= n ) {
		return;
	}

	float s, c;

	switch( op ) {
		case 0: sincos( x[i], &s, &c ); out[i] = s; break;
		case 1: sincos( x[i], &s, &c ); out[i] = c; break;
		case 2: out[i] = tan1( x[i] ); break;
		case 3: out[i] = acos1( x[i] ); break;
		case 4: out[i] = atan1( x[i] ); break;
		case 5: out[i] = pow1( x[i], y[i] ); break;
		case 6: out[i] = fract( sin1( x[i] ) * 43758.5453123f ); break;
		default: out[i] = 0.0f; break;
	}
}

// rays: n x {origin, dir}; outputs as orc_trace_rays
template<bool LIGHTS>
__global__ void diagTrace( const DevParams P, const float* rays, int n, float* outT, int* outFace, float* outNormal, unsigned* outCounts ) {
	const int i = (int) ( blockIdx.x * blockDim.x + threadIdx.x );

	if( i >= n ) {
		return;
	}

	Ray ray;
	ray.origin = mk3( rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2] );
	ray.dir = mk3( rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5] );
	Hit hit;
	hit.t = inff();
	hit.face = 0;
	unsigned nodes = 0, tris = 0;
	traverse<false, LIGHTS, false>( P, nullptr, ray, hit, nodes, tris );

	f3 normal = mk3( 0.0f, 0.0f, 0.0f );

	if( hit.t != inff() ) {
		int material;
		normal = faceNormal( P, hit.face, &material );
	}

	outT[i] = hit.t;
	outFace[i] = hit.face;
	outNormal[i * 3 + 0] = normal.x;
	outNormal[i * 3 + 1] = normal.y;
	outNormal[i * 3 + 2] = normal.z;
	outCounts[i * 2 + 0] = nodes;
	outCounts[i * 2 + 1] = tris;
}

// Traversal-only throughput probe: persistent lanes draw ray indices from P.workCounter, walk,
// store {t, face} — what the walk alone sustains at full occupancy (no shading registers).
template<bool USE_LDS>
__global__ __launch_bounds__( PBR_BLOCK, 8 ) void diagTraceStream( const DevParams P, const float4* rays, unsigned n, float2* out ) {
	const float4* lds = gHotNodes;

	if( USE_LDS ) {
		stageHotNodes( P, gHotNodes );
	}

	unsigned nodes = 0, tris = 0;
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
		traverse<false, false, USE_LDS>( P, lds, ray, hit, nodes, tris );
		out[i] = make_float2( hit.t, __int_as_float( hit.face ) );
		i = atomicAdd( P.workCounter, 1u );
	}

	atomicAdd( &P.counters[0], (unsigned long long) nodes );
	atomicAdd( &P.counters[1], (unsigned long long) tris );
}


// Counter calibration: read a table of `count` float4 in a KNOWN pattern so that
// FETCH_SIZE / TCC_EA0_RDREQ_* can be interpreted for this path's access shapes.
//   MODE 0  coalesced stream: lane l reads element base + l (16 B per lane, 1 KiB per wave)
//   MODE 1  one random 16-B element per lane and step
//   MODE 2  one random 32-B record (two adjacent float4, like a BVH node) per lane and step
template<int MODE>
__global__ __launch_bounds__( 256 ) void diagCalibrate( const float4* table, unsigned long long count, unsigned steps, float* sink ) {
	const unsigned long long tid = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned long long threads = (unsigned long long) gridDim.x * blockDim.x;
	float acc = 0.0f;

	for( unsigned k = 0; k < steps; k++ ) {
		const unsigned long long i = (unsigned long long) k * threads + tid;

		if( MODE == 0 ) {
			const float4 v = table[i % count];
			acc += v.x + v.w;
		}
		else {
			// splitmix-style hash -> uniformly random element
			unsigned long long z = ( i + 1 ) * 0x9e3779b97f4a7c15ULL;
			z = ( z ^ ( z >> 30 ) ) * 0xbf58476d1ce4e5b9ULL;
			z = ( z ^ ( z >> 27 ) ) * 0x94d049bb133111ebULL;
			z ^= z >> 31;

			if( MODE == 1 ) {
				const float4 v = table[z % count];
				acc += v.x + v.w;
			}
			else {
				const unsigned long long r = ( z % ( count / 2 ) ) * 2;
				const float4 a = table[r];
				const float4 b = table[r + 1];
				acc += a.x + b.w;
			}
		}
	}

	if( acc == 123456.789f ) {
		sink[0] = acc;   // never true for the zero-filled table; keeps the loads alive
	}
}

// in: n x 16 {out_dir, in_dir, normal, pad}; out: n x 4 (as orc_brdf_eval); material 0 of P.mats
template<int BRDF>
__global__ void diagBrdf( const DevParams P, const float* in, int n, float* out ) {
	const int i = (int) ( blockIdx.x * blockDim.x + threadIdx.x );

	if( i >= n ) {
		return;
	}

	const float* p = in + (size_t) i * 16;
	const Material mtl = loadMaterial( P, 0 );
	const f3 outDir = mk3( p[0], p[1], p[2] );
	const f3 inDir = mk3( p[3], p[4], p[5] );
	const f3 normal = mk3( p[6], p[7], p[8] );
	float* o = out + (size_t) i * 4;

	if( BRDF == 0 ) {
		float u, pdf;
		const float b = brdfSchlick( mtl, outDir, inDir, normal, &u, &pdf );
		o[0] = b; o[1] = u; o[2] = pdf; o[3] = 0.0f;
	}
	else {
		float spec, diff, dotHK1, pdf;
		brdfSA( mtl, outDir, inDir, normal, &spec, &diff, &dotHK1, &pdf );
		o[0] = spec; o[1] = diff; o[2] = dotHK1; o[3] = pdf;
	}
}

// in: n x 12 {origin, dir, normal, t, seed, pad}; out: n x 8 (as orc_new_ray); material 0
template<int BRDF>
__global__ void diagNewRay( const DevParams P, const float* in, int n, float* out ) {
	const int i = (int) ( blockIdx.x * blockDim.x + threadIdx.x );

	if( i >= n ) {
		return;
	}

	const float* p = in + (size_t) i * 12;
	const Material mtl = loadMaterial( P, 0 );
	const f3 origin = mk3( p[0], p[1], p[2] );
	const f3 dir = mk3( p[3], p[4], p[5] );
	const f3 normal = mk3( p[6], p[7], p[8] );
	float seed = p[10];
	bool addDepth = false;
	const f3 newOrigin = fma3( p[9], dir, origin );
	const f3 newDir = newRayDir<BRDF>( dir, normal, mtl, seed, addDepth );
	float* o = out + (size_t) i * 8;
	o[0] = newOrigin.x; o[1] = newOrigin.y; o[2] = newOrigin.z;
	o[3] = newDir.x; o[4] = newDir.y; o[5] = newDir.z;
	o[6] = seed;
	o[7] = addDepth ? 1.0f : 0.0f;
}

}  // namespace ptk
