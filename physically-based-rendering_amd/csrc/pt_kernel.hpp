// The path-tracing kernels for gfx950.
//
// Replaces source/opencl/pathtracing.cl + pt_*.cl (one OpenCL work-item per pixel, one launch
// per frame, accumulation through the host).  Here:
//   * persistent waves (the grid is sized to the chip, not to the image) draw units of work — one
//     frame of one pixel — from an XCD-banded queue (nextSlot); a lane whose path ends starts its next
//     path, and a lane whose unit ends takes the next unit (the state machine at once, the lock-step
//     kernels in batches);
//   * a multi-frame render is ONE launch over all (pixel, frame) units; each writes {finalColor, focus}
//     to a frame buffer and foldFrames applies the running mean (pt_rgb.cl:9-21) in frame order.
//     The framebuffer is tile-major (64 px x RGBA32F = 1 KiB per tile);
//   * the BVH is a node stream of 32-B records with explicit successors, the most-visited records first
//     and staged in LDS (decodeNode); the walk alternates a hand-scheduled node phase with a leaf phase
//     for the lanes parked on hit leaves (traverse, nodePhaseAsm);
//   * triangles are pre-gathered per face as {a, b-a, c-a, material} = 48 B, removing the
//     facesV -> vertices indirection of pt_intersect.cl:146-149;
//   * three ways to keep lanes busy: pathTracing (lock step per bounce), pathTracingPhased (a lane
//     state machine) and pathTracingDual (the state machine with two paths per lane, pt_dual.hpp);
//     the host times them on the scene and keeps the fastest (pbr_hip.hip, launch()).
// Per-pixel results are bit-identical to the per-frame reference schedule: a pixel's value
// depends only on (seed_k, scene, its own previous value), pathtracing.cl:28,255,332.
#pragma once

#include <hip/hip_runtime.h>

#include "pt_math.hpp"

#ifdef PBR_GUARD
#define PBR_GUARD_PATH 1
#define PBR_GUARD_TRAV 1
#endif

// Measurement hooks.  The product build defines them empty; a lab build (-DPBR_LAB_HOOKS -I lab/src, scripts/lab.sh)
// takes them from lab/src/pt_lab_hooks.hpp, where they record wave-level statistics into the spare counter slots
// (lanes per node iteration, time per phase, when a wave first finds the queue empty ...).  They never change results.
#ifdef PBR_LAB_HOOKS
#include "pt_lab_hooks.hpp"
#else
#define PT_LAB_TRAVERSE_BEGIN
#define PT_LAB_NODE_ITERATION
#define PT_LAB_LEAF_PHASE
#define PT_LAB_TRAVERSE_END( P, anyhit )
#define PT_LAB_WAVE_BEGIN
#define PT_LAB_WAVE_DRY
#define PT_LAB_WAVE_END_LOCKSTEP( P )
#define PT_LAB_WAVE_END_PHASED( P )
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

namespace ptk {

using namespace ptm;

#define EPSILON5 0.00001f
#define NI_AIR 1.00028f
#define PI_X2 6.28318530718f
#define M_PI_D 0x1.921fb54442d18p+1
#define M_PI_2_D 0x1.921fb54442d18p+0
#define M_1_PI_D 0x1.45f306dc9c883p-2

#ifndef PT_BANDS
#define PT_BANDS 8
#endif
// queue heads per band (nextSlot); a power of two
#ifndef PT_SUB
#define PT_SUB 4
#endif
#define PT_HEADS ( PT_BANDS * PT_SUB )

// A leaf's second face record is requested before the first face is tested (testLeaf, EAGER: 12 more registers during the
// leaf phase, one memory latency less per two-face leaf) in kernels of up to this many waves / SIMD: the state machine /
// the lock-step kernels
#ifndef PT_EAGER_UP_TO
#define PT_EAGER_UP_TO 8
#endif
// Wave priority (s_setprio) while a wave walks: a SIMD arbitrates between its waves by it.  A walking wave issues a
// handful of instructions and then waits on a fetch; a shading wave issues hundreds back to back.  With the walkers
// ahead of the shaders a node visit's loads go out as soon as its wave can issue, and shading fills the gaps: measured
// +2.0 % (Sponza-class), +0.6 % (Dragon-class), 0 (hairball) in the lane state machine with the leaf phase included,
// +3.2 % on Cornell and +2.2 % on the Sponza-class scene in the lock-step kernels (node phase only); the other way
// round — shading ahead — costs 2.7 %.  Priorities 1, 2 and 3 measure alike (profiles/r02/experiments/prio.txt).
#ifndef PT_WALK_PRIO
#define PT_WALK_PRIO 1
#endif
#ifndef PT_EAGER_REFILL_UP_TO
#define PT_EAGER_REFILL_UP_TO 8
#endif

struct DevParams {
	const float4* nodes;    // the node stream: 2 x float4 per record {min.xy, max.xy}, {min.z, max.z, w0, w1}, see decodeNode;
	                        // the most-visited nodes first (every block copies records [0, numHot) to LDS)
	const float4* tris;     // 3 x float4 per face: {a.xyz, e1.x}, {e1.y, e1.z, e2.x, e2.y}, {e2.z, material(int bits), 0, 0}
	const float4* triPN;    // Phong tessellation only: 6 x float4 per face {a, b, c, an, bn, cn} (exact vertices and vertex normals)
	const float4* mats;     // 4 x float4 per material: {d, Ni, p|nu, rough|nv}, {Rs, Rd, 0, 0}, Kd, Ks
	const float4* lights;   // 3 x float4 per light: pos, rgb, {type, radius, 0, 0}
	const float4* imgIn;    // tile-major, local tiles
	float4* imgOut;
	float4* imgDbg;
	const float* seeds;     // one per frame
	float4* frameBuf;       // frame-parallel launches: {finalColor, focus} of frame k, pixel slot s at frameBufIndex( s, k )
	unsigned frameStride;   // = numLocalTiles * 64: the pixel slots of this launch
	unsigned long long* counters;  // nodes, tris, hits, paths
	unsigned int* workCounter;  // PT_HEADS heads, PT_SUB per band of the pixel-slot queue, PT_BAND_STRIDE words apart, and behind them the word of the heads seen empty (nextSlot)
	unsigned int* guard;    // [0] tile-loop, [1] path-loop, [2] traversal trips (PBR_GUARD builds only)

	float eye[3], cw[3], cu[3], cv[3];
	// launch-invariant sub-expressions of initRay, evaluated once on the host with the same IEEE operations (left in
	// the kernel the compiler hoists them out of the path loop into vector registers — they are float arithmetic — and,
	// at 80 registers, spills them: 40 of the 48 bytes of scratch per lane the 6-waves kernels had)
	float camA[3];          // cu - cu * W
	float cvH[3];           // cv * H
	float halfPx;           // pxDim * 0.5f
	float aperture;         // lenseFocal / lenseAperture
	float samplesF;         // (float) samples
	int focusX, focusY;
	int focusGiven;         // tile sharding: the focus pixel's previous-frame distance comes from the caller (focusDepth) ...
	float focusDepth;       // ... because its tile may live on another rank (pbr_set_focus_depth)
	float lenseFocal, lenseAperture;

	int width, height, tilesX, numLocalTiles, tileWorld, tileRank;
	const unsigned* tileOrder;        // the dealing order (nextSlot): per band of the queue its local tiles in the order they are dealt,
	                                  // band after band.  Placement only — every unit is handed out exactly once in any order.
	unsigned bandFirst[PT_BANDS];     // ... where a band's stretch of tileOrder starts
	unsigned bandTiles[PT_BANDS];     // ... and how many tiles it has
	unsigned framesDiv[2];            // {magic, shifts} to divide by nFrames (nextSlot)
	unsigned tilesXDiv[2];  // {magic, shifts} to divide by tilesX (pixelOfSlot, divInvariant)
	float phongAlpha;            // PHONGTESS_ALPHA (kernels built with PHONG = true only)
	int parkEighths;             // traverse(): a node phase ends once this many eighths of the lanes that entered it have left it
	int drainMode;               // phased schedule, once lanes are DONE: bit 0 scale phPark, bit 1 scale phShade with the lanes still at work
	int refillBatch;             // lock-step schedule: lanes of a wave that wait with a finished unit before they take their next units together
	int phPark, phShade;         // phased schedule: lanes that leave a node phase before it ends / lanes that wait before a shade phase runs
	int numNodes, numLights, maxDepth, maxAddedDepth, samples;
	int numHot;             // records [0, numHot) of the node stream are resident in LDS
	int numHotBytes;        // = numHot * 32: a record reference (byte offset) below this is resident
	int firstRef;           // reference (byte offset) of node 1's record, where every walk starts (walkScheme 0)
	int walkScheme;         // pbr_config.traversal: 0 the reference's one order; 1 six orders (dominant axis x sign of the ray direction),
	                        // 2 eight (sign octant): `nodes` then holds one stream of records per order and a walk starts at walkFirst[order]
	const int* walkTable;   // ... the eight first references (as eight kernel arguments they were eight more live registers in every
	                        // kernel: scalar values the compiler keeps in vector registers).  They sit in the 32 bytes in front of
	                        // `nodes`, but the kernels must reach them through a pointer of their own: a vector load based on
	                        // `nodes` makes hipcc hand the node phases' "s"( P.nodes ) operand a vector register pair
	int slotBase;           // LDS byte address of the per-lane state behind the staged prefix (= numHotBytes): pathTracingDual's path slots
	int nFrames, firstCount;
	int useExplicitWeight;
	float explicitWeight;
	float pxDim, antiAliasing;
	float sky[3];
	// last, so that kernels which do not use it do not load it: per face {unit normal of the flat triangle,
	// material(int bits)} — what faceNormal() computes from the record in `tris` on every hit, evaluated once by
	// prepareFaceNormals with the same code.  The lock-step kernels read it (Cornell +1.3 %); in the lane state
	// machine the extra pointer costs more in registers than the 40 instructions per hit are worth (measured -2 %).
	const float4* faceN;
};

// ---- which rank owns which tile (multi-GPU sharding) -----------------------------------------
// Tiles are dealt round-robin along a DEALING ORDER: row-major, but row ty rotated by PT_DEAL_SHIFT * ty columns.
// position p of tile (tx, ty) = ty * tilesX + ( tx + PT_DEAL_SHIFT * ty ) % tilesX;  owner = p % world, local index =
// p / world.  Dealing along the plain row-major order gives every rank whole tile COLUMNS whenever tilesX is a multiple
// of world (1080p: 240 columns, 8 ranks), and columns do not cost the same: measured 2.6 % (Cornell) and 2.3 %
// (Dragon-class) more work on the heaviest of 8 ranks than on the average one; along the rotated order 0.04 % / 0.2 %
// (scripts/tile_balance.py).  With world = 1 there is nothing to deal and the order is the plain one.
#define PT_DEAL_SHIFT 5

__host__ __device__ inline int dealPositionOfTile( int tileGlobal, int tilesX, int world ) {
	if( world <= 1 ) {
		return tileGlobal;
	}

	const int ty = tileGlobal / tilesX;
	const int tx = tileGlobal - ty * tilesX;
	return ty * tilesX + ( tx + PT_DEAL_SHIFT * ty ) % tilesX;
}

__host__ __device__ inline int tileAtDealPosition( int position, int tilesX, int world ) {
	if( world <= 1 ) {
		return position;
	}

	const int ty = position / tilesX;
	const int shifted = position - ty * tilesX;
	const int back = ( PT_DEAL_SHIFT * ty ) % tilesX;
	const int tx = ( shifted >= back ) ? shifted - back : shifted - back + tilesX;
	return ty * tilesX + tx;
}

struct Ray {
	f3 origin, dir;
};

// The scalar half of a material.  Its two colours are read where the BRDF factor is formed (materialColours): held from
// the start of the shading pass they were six registers live across both pow calls, and the 6-waves state machine
// spilled three of them around every pass (round 2: 12 B of scratch per lane, one colour stored and reloaded per pass).
// LATE = false (the lock-step kernels with 80 registers and more, which never spilled them): both at the start.
struct Material {
	float d, Ni, p2, p3;  // p2 = p | nu, p3 = rough | nv
	float Rs, Rd;
	const float4* colours;   // {Kd, Ks} of this material in P.mats (LATE)
	f3 Kd, Ks;               // (!LATE)
};

struct MaterialColours {
	f3 Kd, Ks;
};

struct Hit {
	float t;
	int face;      // >= 0 face index; < 0: light -(i+1) (only with t == INF)
	f3 normal;     // PHONG kernels only: ray.normal of the reference (pt_bvh.cl:18); the others recompute it from `face`
};

PT_DEV f3 ld3( const float* p ) { return mk3( p[0], p[1], p[2] ); }

// rand, pt_utils.cl:39-44
PT_DEV float rnd( float& seed ) {
	seed += 1.0f;
	return fract( sin1( seed ) * 43758.5453123f );
}

// fresnel, pt_utils.cl:53-56
PT_DEV float fresnel( float u, float c ) {
	const float v = 1.0f - u;
	return c + ( 1.0f - c ) * v * v * v * v * v;
}

// jitter, pt_utils.cl:306-318
// The tangent frame of a normal as jitter (pt_utils.cl:264-279) and brdfShirleyAshikhmin (pt_brdf.cl:233-234) build it
// — the same two expressions in both, so a bounce that samples a direction around a normal and then evaluates the
// BRDF for the same normal (bit for bit: the check is on the bits) needs it once.
struct TangentFrame {
	f3 n, u, v;
	bool valid;
};

PT_DEV void tangentFrame( f3 n, f3* u, f3* v ) {
	*u = normalize( cross( yzx( n ), n ) );
	*v = normalize( cross( n, *u ) );
}

PT_DEV bool sameBits( f3 a, f3 b ) {
	return __float_as_int( a.x ) == __float_as_int( b.x ) && __float_as_int( a.y ) == __float_as_int( b.y ) && __float_as_int( a.z ) == __float_as_int( b.z );
}

PT_DEV f3 jitterUV( f3 nl, f3 u, f3 v, float phi, float sina, float cosa ) {
	float sp, cp;
	sincos( phi, &sp, &cp );
	const f3 w = normalize( u * cp + v * sp );
	return normalize( w * sina + nl * cosa );
}

PT_DEV f3 jitter( f3 nl, float phi, float sina, float cosa ) {
	f3 u, v;
	tangentFrame( nl, &u, &v );
	return jitterUV( nl, u, v, phi, sina, cosa );
}

template<bool LATE = true>
PT_DEV Material loadMaterial( const DevParams& P, int index ) {
	const float4 a = P.mats[index * 4 + 0];
	const float4 b = P.mats[index * 4 + 1];
	Material m;
	m.d = a.x; m.Ni = a.y; m.p2 = a.z; m.p3 = a.w;
	m.Rs = b.x; m.Rd = b.y;
	m.colours = P.mats + index * 4 + 2;
	m.Kd = m.Ks = mk3( 0.0f, 0.0f, 0.0f );

	if( !LATE ) {
		const float4 kd = m.colours[0];
		const float4 ks = m.colours[1];
		m.Kd = mk3( kd.x, kd.y, kd.z );
		m.Ks = mk3( ks.x, ks.y, ks.z );
	}

	return m;
}

template<bool LATE = true>
PT_DEV MaterialColours materialColours( const Material& m ) {
	MaterialColours c;

	if( LATE ) {
		const float4 kd = m.colours[0];
		const float4 ks = m.colours[1];
		c.Kd = mk3( kd.x, kd.y, kd.z );
		c.Ks = mk3( ks.x, ks.y, ks.z );
	}
	else {
		c.Kd = m.Kd;
		c.Ks = m.Ks;
	}

	return c;
}


// ---------------------------------------------------------------------------------------
// Traversal
// ---------------------------------------------------------------------------------------

// intersectSphere, pt_intersect.cl:37-77 (d2 is compared with r, not r*r, as in the reference)
PT_DEV bool intersectSphere( const Ray& ray, f3 pos, float r, float* tNear ) {
	const f3 L = pos - ray.origin;
	const float tca = dot( L, ray.dir );

	if( tca < 0.0f ) {
		return false;
	}

	const float d2 = dot( L, L ) - tca * tca;

	if( d2 > r ) {
		return false;
	}

	const float thc = sqrt1( r - d2 );
	float t0 = tca - thc;
	float t1 = tca + thc;

	if( t0 > t1 ) {
		const float tmp = t0; t0 = t1; t1 = tmp;
	}

	if( t0 < 0.0f ) {
		t0 = t1;

		if( t0 < 0.0f ) {
			return false;
		}
	}

	*tNear = t0;
	return true;
}

// traverseLights, pt_bvh.cl:54-74
PT_DEV void traverseLights( const DevParams& P, const Ray& ray, Hit& hit ) {
	for( int i = 0; i < P.numLights; i++ ) {
		const float4 pos = P.lights[i * 3 + 0];
		const float4 data = P.lights[i * 3 + 2];

		if( data.x == 2.0f ) {
			float tNear = 0.0f;

			if( intersectSphere( ray, mk3( pos.x, pos.y, pos.z ), data.y, &tNear ) && tNear < hit.t ) {
				hit.t = inff();
				hit.face = -( i + 1 );
			}
		}
	}
}

// One pre-gathered face record (48 B; the array is padded by one record, so face + 1 is always readable)
struct TriRecord {
	float4 r0, r1, r2;
};

PT_DEV TriRecord loadTri( const DevParams& P, int face ) {
	TriRecord r;
	r.r0 = P.tris[face * 3 + 0];
	r.r1 = P.tris[face * 3 + 1];
	r.r2 = P.tris[face * 3 + 2];
	return r;
}

// flatTriAndRayIntersect, pt_intersect.cl:92-129, on the pre-gathered record.  Returns t or INF.
PT_DEV float triangleT( const TriRecord& rec, const Ray& ray, float rayT, float tNear ) {
	const float4 r0 = rec.r0;
	const float4 r1 = rec.r1;
	const float4 r2 = rec.r2;
	const f3 a = mk3( r0.x, r0.y, r0.z );
	const f3 edge1 = mk3( r0.w, r1.x, r1.y );
	const f3 edge2 = mk3( r1.z, r1.w, r2.x );

	const float f = fmax1( 0.0f, tNear - 0.001f );
	const f3 closeOrigin = fma3( f, ray.dir, ray.origin );
	const f3 tVec = closeOrigin - a;
	const f3 pVec = cross( ray.dir, edge2 );
	const f3 qVec = cross( tVec, edge1 );
	const float invDet = 1.0f / dot( edge1, pVec );

	float t = dot( edge2, qVec ) * invDet;

	if( t >= rayT || t < EPSILON5 ) {
		return inff();
	}

	const float u = dot( tVec, pVec ) * invDet;
	const float v = dot( ray.dir, qVec ) * invDet;

	if( u + v > 1.0f || fmin1( u, v ) < 0.0f ) {
		return inff();
	}

	return t + f;
}

// traverse (pt_bvh.cl:82-123) / traverseShadows (:133-177).  ANYHIT: the shadow variant —
// no `ray.t > tNear` cull, stops at the first face hit nearer than the light.
// The slab test of intersectBox (pt_intersect.cl:11-25) plus the hit condition of the walk
// (pt_bvh.cl:107-110; the shadow walk has no `ray.t > tNear` cull, :151-154).
typedef float f2v __attribute__( ( ext_vector_type( 2 ) ) );

// A node record is laid out for the slab test: n0 = {min.x, min.y, max.x, max.y}, n1 = {min.z, max.z, w0, w1},
// so that the three register pairs the packed instructions need are the halves of the two 16-B loads.
template<bool ANYHIT>
PT_DEV bool boxHit( const float4 n0, const float4 n1, const Ray& ray, const f3 invDir, float rayT, float* tNearOut, float* tFarOut = nullptr ) {
	// ( bb - origin ) * invDir for both planes of an axis, as 2-vectors: v_pk_add_f32 / v_pk_mul_f32
	// (two IEEE operations per instruction, same rounding as the scalar forms)
	const f2v oxy = { ray.origin.x, ray.origin.y };
	const f2v ixy = { invDir.x, invDir.y };
	const f2v lxy = { n0.x, n0.y };
	const f2v hxy = { n0.z, n0.w };
	const f2v t1xy = ( lxy - oxy ) * ixy;
	const f2v t2xy = ( hxy - oxy ) * ixy;
	const f2v oz = { ray.origin.z, ray.origin.z };
	const f2v iz = { invDir.z, invDir.z };
	const f2v bz = { n1.x, n1.y };
	const f2v tz = ( bz - oz ) * iz;
	const float t1x = t1xy.x, t1y = t1xy.y, t2x = t2xy.x, t2y = t2xy.y, t1z = tz.x, t2z = tz.y;
	const float tNear = fmax1( fmax1( fmin1( t1x, t2x ), fmin1( t1y, t2y ) ), fmin1( t1z, t2z ) );
	const float tFar = fmin1( fmin1( fmax1( t1x, t2x ), fmax1( t1y, t2y ) ), fmin1( fmax1( t1z, t2z ), inff() ) );
	*tNearOut = tNear;

	if( tFarOut != nullptr ) {
		*tFarOut = tFar;
	}

	bool isNodeHit = ( tNear <= tFar ) && ( tFar > EPSILON5 );

	if( !ANYHIT ) {
		isNodeHit = isNodeHit && ( rayT > tNear );
	}

	return isNodeHit;
}

// ---- Phong tessellation (pt_phongtess.cl, PHONGTESS == 1; kernels built with PHONG = true) ----------------

// solveCubic, pt_utils.cl:108-199: a0 x^3 + a1 x^2 + a2 x + a3 = 0, returns the number of real roots in x[]
PT_DEV int solveCubic( float a0, float a1, float a2, float a3, float x[3] ) {
	const float THIRD = 0.3333333333f;
	const float THIRD_HALF = 0.1666666666f;
	float w, p, q, dis, phi;

	if( __builtin_fabsf( a0 ) > 0.0f ) {
		w = ( a1 / a0 ) * THIRD;
		p = ( a2 / a0 ) * THIRD - w * w;
		p = p * p * p;
		q = 0.5f * ( ( a2 * w - a3 ) / a0 ) - w * w * w;
		dis = q * q + p;

		if( dis < 0.0f ) {
			phi = acos1( fmin1( fmax1( q / sqrt1( -p ), -1.0f ), 1.0f ) );
			p = 2.0f * pow1( -p, THIRD_HALF );

			// ( phi + 2.0f * M_PI ) * THIRD: M_PI is a double literal, the sum and product are binary64
			const float u0 = p * cos1( phi * THIRD ) - w;
			const float u1 = p * cos1( (float) ( ( (double) phi + (double) 2.0f * M_PI_D ) * (double) THIRD ) ) - w;
			const float u2 = p * cos1( (float) ( ( (double) phi + (double) 4.0f * M_PI_D ) * (double) THIRD ) ) - w;

			x[0] = fmin1( u0, fmin1( u1, u2 ) );
			x[1] = fmax1( fmin1( u0, u1 ), fmax1( fmin1( u0, u2 ), fmin1( u1, u2 ) ) );
			x[2] = fmax1( u0, fmax1( u1, u2 ) );

			for( int k = 0; k < 3; k++ ) {
				x[k] -= ( a3 + x[k] * ( a2 + x[k] * ( a1 + x[k] * a0 ) ) ) / ( a2 + x[k] * ( 2.0f * a1 + x[k] * 3.0f * a0 ) );
			}

			return 3;
		}

		dis = sqrt1( dis );
		x[0] = cbrt1( q + dis ) + cbrt1( q - dis ) - w;
		x[0] -= ( a3 + x[0] * ( a2 + x[0] * ( a1 + x[0] * a0 ) ) ) / ( a2 + x[0] * ( 2.0f * a1 + x[0] * 3.0f * a0 ) );

		return 1;
	}
	else if( __builtin_fabsf( a1 ) > 0.0f ) {
		p = 0.5f * ( a2 / a1 );
		dis = p * p - a3 / a1;

		if( dis >= 0.0f ) {
			const float disSqrt = sqrt1( dis );
			x[0] = -p - disSqrt;
			x[1] = -p + disSqrt;
			x[0] -= ( a3 + x[0] * ( a2 + x[0] * a1 ) ) / ( a2 + x[0] * 2.0f * a1 );
			x[1] -= ( a3 + x[1] * ( a2 + x[1] * a1 ) ) / ( a2 + x[1] * 2.0f * a1 );
			return 2;
		}
	}
	else if( __builtin_fabsf( a2 ) > 0.0f ) {
		x[0] = -a3 / a2;
		return 1;
	}

	return 0;
}

// projectOnPlane, pt_utils.cl:397-399
PT_DEV f3 projectOnPlane( f3 q, f3 p, f3 n ) {
	return q - n * dot( q - p, n );
}

// phongTessellation, pt_phongtess.cl:14-26
PT_DEV f3 phongTessellation( f3 P1, f3 P2, f3 P3, f3 N1, f3 N2, f3 N3, float u, float v, float w, float alpha ) {
	const f3 pBary = ( P1 * u + P2 * v ) + P3 * w;
	const f3 pTessellated = ( projectOnPlane( pBary, P1, N1 ) * u + projectOnPlane( pBary, P2, N2 ) * v ) + projectOnPlane( pBary, P3, N3 ) * w;
	return pBary * ( 1.0f - alpha ) + pTessellated * alpha;
}

// getTriangleNormalS / getTriangleNormal / getTriangleReflectionVec / getPhongTessNormal, pt_utils.cl:231-294
PT_DEV f3 getPhongTessNormal( f3 an, f3 bn, f3 cn, f3 rayDir, float u, float v, float w, f3 C1, f3 C2, f3 C3, f3 E12, f3 E20 ) {
	const f3 du = ( C3 * ( w - u ) + ( C1 - C2 ) * v ) + E20;
	const f3 dv = ( C2 * ( w - v ) + ( C1 - C3 ) * u ) - E12;
	const f3 ns = normalize( cross( du, dv ) );
	const f3 np = normalize( ( an * u + bn * v ) + cn * w );
	const f3 r = rayDir - ( np * 2.0f ) * dot( rayDir, np );
	return ( dot( ns, r ) < 0.0f ) ? ns : np;
}

// A conic uu u^2 + vv v^2 + k + 2 ( uv u v + u1 u + v1 v ) = 0 in the barycentric ( u, v ) plane of a Phong-tessellated patch
struct Conic {
	float uu, vv, k, uv, u1, v1;
};

// phongTessTriAndRayIntersect, pt_phongtess.cl:56-212 (after Ogaki & Tokuyoshi, "Direct Ray Tracing of Phong
// Tessellation"); getPlanesFromRay (pt_utils.cl:208-218) and getBestRayDomain (pt_phongtess.cl:35-44) inlined.
// Returns t (INF: no hit) and the normal at the hit.
PT_DEV float phongTessTriAndRayIntersect(
	f3 P1, f3 P2, f3 P3, f3 N1, f3 N2, f3 N3, const Ray& ray, float rayT, float tNear, float tFar, float alpha, f3* normalOut
) {
	f3 normal = mk3( 0.0f, 0.0f, 0.0f );
	float t = inff();
	*normalOut = normal;

	const f3 E01 = P2 - P1;
	const f3 E12 = P3 - P2;
	const f3 E20 = P1 - P3;
	const f3 C1 = ( N2 * dot( N2, E01 ) - N1 * dot( N1, E01 ) ) * alpha;
	const f3 C2 = ( N3 * dot( N3, E12 ) - N2 * dot( N2, E12 ) ) * alpha;
	const f3 C3 = ( N1 * dot( N1, E20 ) - N3 * dot( N3, E20 ) ) * alpha;

	const f3 n1 = normalize( cross( ray.origin, ray.dir ) );
	const f3 n2 = normalize( cross( n1, ray.dir ) );
	const float o1 = dot( n1, ray.origin );
	const float o2 = dot( n2, ray.origin );
	const f3 C123 = ( C1 - C2 ) - C3;

	// the conics the ray's two planes cut out of the patch, in the barycentric ( u, v ) plane (Conic above)
	Conic ca, cb;
	ca.uu = dot( -n1, C3 );
	ca.vv = dot( -n1, C2 );
	ca.k = dot( n1, P3 ) - o1;
	ca.uv = dot( n1, C123 ) * 0.5f;
	ca.u1 = dot( n1, C3 + E20 ) * 0.5f;
	ca.v1 = dot( n1, C2 - E12 ) * 0.5f;
	cb.uu = dot( -n2, C3 );
	cb.vv = dot( -n2, C2 );
	cb.k = dot( n2, P3 ) - o2;
	cb.uv = dot( n2, C123 ) * 0.5f;
	cb.u1 = dot( n2, C3 + E20 ) * 0.5f;
	cb.v1 = dot( n2, C2 - E12 ) * 0.5f;

	// det( lambda ca + cb ) = 0: the members of the pencil of the two conics that are degenerate (a pair of lines)
	float roots[3] = { -1.0f, -1.0f, -1.0f };
	const float a3 = ( cb.uu*cb.vv*cb.k + 2.0f*cb.uv*cb.u1*cb.v1 ) - ( cb.uu*cb.v1*cb.v1 + cb.vv*cb.u1*cb.u1 + cb.k*cb.uv*cb.uv );
	const float a2 = ( ca.uu*cb.vv*cb.k + cb.uu*ca.vv*cb.k + cb.uu*cb.vv*ca.k + 2.0f*( ca.uv*cb.u1*cb.v1 + cb.uv*ca.u1*cb.v1 + cb.uv*cb.u1*ca.v1 ) ) -
	                 ( ca.uu*cb.v1*cb.v1 + ca.vv*cb.u1*cb.u1 + ca.k*cb.uv*cb.uv + 2.0f*( cb.uu*ca.v1*cb.v1 + cb.vv*ca.u1*cb.u1 + cb.k*ca.uv*cb.uv ) );
	const float a1 = ( ca.uu*ca.vv*cb.k + ca.uu*cb.vv*ca.k + cb.uu*ca.vv*ca.k + 2.0f*( cb.uv*ca.u1*ca.v1 + ca.uv*ca.u1*cb.v1 + ca.uv*cb.u1*ca.v1 ) ) -
	                 ( cb.uu*ca.v1*ca.v1 + cb.vv*ca.u1*ca.u1 + cb.k*ca.uv*ca.uv + 2.0f*( ca.uu*ca.v1*cb.v1 + ca.vv*ca.u1*cb.u1 + ca.k*ca.uv*cb.uv ) );
	const float a0 = ( ca.uu*ca.vv*ca.k + 2.0f*ca.uv*ca.u1*ca.v1 ) - ( ca.uu*ca.v1*ca.v1 + ca.vv*ca.u1*ca.u1 + ca.k*ca.uv*ca.uv );
	const int pencilRoots = solveCubic( a0, a1, a2, a3, roots );

	if( pencilRoots == 0 ) {
		return t;
	}

	// of those, the member whose quadratic part is most clearly a pair of REAL lines ( uv^2 - uu vv > 0 )
	float lambda = 0.0f;
	float smallest = inff();
	Conic deg;

	for( int i = 0; i < pencilRoots; i++ ) {
		deg.uu = ca.uu * roots[i] + cb.uu;
		deg.vv = ca.vv * roots[i] + cb.vv;
		deg.uv = ca.uv * roots[i] + cb.uv;
		const float tmp = deg.uv * deg.uv - deg.uu * deg.vv;
		lambda = ( smallest > tmp ) ? roots[i] : lambda;
		smallest = fmin1( smallest, tmp );
	}

	if( 0.0f >= smallest ) {
		return t;
	}

	const f3 ad = mk3( __builtin_fabsf( ray.dir.x ), __builtin_fabsf( ray.dir.y ), __builtin_fabsf( ray.dir.z ) );
	int domain = ( ad.y > ad.z ) ? 1 : 2;

	if( ad.x > ad.y ) {
		domain = ( ad.x > ad.z ) ? 0 : 2;
	}

	deg.uu = ca.uu * lambda + cb.uu;
	deg.vv = ca.vv * lambda + cb.vv;
	deg.k = ca.k * lambda + cb.k;
	deg.uv = ca.uv * lambda + cb.uv;
	deg.u1 = ca.u1 * lambda + cb.u1;
	deg.v1 = ca.v1 * lambda + cb.v1;

	// normalise by the larger square term (solve for the other variable) and split the degenerate conic into its two lines
	const bool swapUV = __builtin_fabsf( deg.uu ) < __builtin_fabsf( deg.vv );
	const float pivot = swapUV ? deg.vv : deg.uu;
	deg.uu = deg.uu / pivot;
	deg.vv = deg.vv / pivot;
	deg.k = deg.k / pivot;
	deg.uv = deg.uv / pivot;
	deg.u1 = deg.u1 / pivot;
	deg.v1 = deg.v1 / pivot;

	const float quadTerm = swapUV ? deg.uu : deg.vv;
	const float crossTerm = swapUV ? 2.0f * deg.u1 : 2.0f * deg.v1;
	const float linTerm = swapUV ? deg.v1 : deg.u1;
	const float lead = swapUV ? ca.uu : ca.vv;
	const float trail = swapUV ? ca.vv : ca.uu;
	const float leadLin = swapUV ? ca.u1 : ca.v1;
	const float trailLin = swapUV ? ca.v1 : ca.u1;

	const float slopeSpread = sqrt1( deg.uv * deg.uv - quadTerm );
	const float offsetSpread = sqrt1( linTerm * linTerm - deg.k );
	const float slopeA = deg.uv + slopeSpread;
	const float slopeB = deg.uv - slopeSpread;
	float offsetA = linTerm + offsetSpread;
	float offsetB = linTerm - offsetSpread;

	if( __builtin_fabsf( crossTerm - slopeA * offsetA - slopeB * offsetB ) < __builtin_fabsf( crossTerm - slopeA * offsetB - slopeB * offsetA ) ) {
		const float tmp = offsetA;
		offsetA = offsetB;
		offsetB = tmp;
	}

	// each line, substituted into the first conic, is a quadratic in one barycentric coordinate: its roots inside the
	// triangle are the ray's intersections with the patch
	for( int line = 0; line < 2; line++ ) {
		const float slope = ( line == 0 ) ? -slopeA : -slopeB;
		const float offset = ( line == 0 ) ? -offsetA : -offsetB;
		const float q0 = lead + slope * ( 2.0f * ca.uv + trail * slope );
		const float q1 = 2.0f * ( offset * ( ca.uv + trail * slope ) + leadLin + trailLin * slope );
		const float q2 = offset * ( trail * offset + 2.0f * trailLin ) + ca.k;
		const int hits = solveCubic( 0.0f, q0, q1, q2, roots );

		for( int i = 0; i < hits; i++ ) {
			float u = roots[i];
			float v = slope * u + offset;
			const float w = 1.0f - u - v;

			if( u < 0.0f || v < 0.0f || w < 0.0f ) {
				continue;
			}

			if( !swapUV ) {
				const float tmp = u;
				u = v;
				v = tmp;
			}

			const f3 toPoint = phongTessellation( P1, P2, P3, N1, N2, N3, u, v, w, alpha ) - ray.origin;
			const float num = ( domain == 0 ) ? toPoint.x : ( domain == 1 ) ? toPoint.y : toPoint.z;
			const float den = ( domain == 0 ) ? ray.dir.x : ( domain == 1 ) ? ray.dir.y : ray.dir.z;
			const float tHit = num / den;

			if( tHit >= __builtin_fabsf( tNear ) && tHit <= fmin1( t, fmin1( rayT, tFar ) ) ) {
				t = tHit;
				normal = getPhongTessNormal( N1, N2, N3, ray.dir, u, v, w, C1, C2, C3, E12, E20 );
			}
		}
	}

	*normalOut = normal;
	return t;
}

// checkFaceIntersection with PHONGTESS == 1 (pt_intersect.cl:142-176): a face whose three vertex normals are equal
// takes the flat test (on the pre-gathered record; its normal is the geometric one)
PT_DEV float phongFaceT( const DevParams& P, int face, const Ray& ray, float rayT, float tNear, float tFar, f3* normalOut ) {
	const float4 pa = P.triPN[(size_t) face * 6 + 0], pb = P.triPN[(size_t) face * 6 + 1], pc = P.triPN[(size_t) face * 6 + 2];
	const float4 na = P.triPN[(size_t) face * 6 + 3], nb = P.triPN[(size_t) face * 6 + 4], nc = P.triPN[(size_t) face * 6 + 5];
	const bool allEqual = na.x == nb.x && na.y == nb.y && na.z == nb.z && nb.x == nc.x && nb.y == nc.y && nb.z == nc.z;

	if( allEqual ) {
		const TriRecord rec = loadTri( P, face );
		const float t = triangleT( rec, ray, rayT, tNear );
		const float4 r0 = rec.r0, r1 = rec.r1, r2 = rec.r2;
		// flatTriAndRayIntersect returns the zero vector with t = INF; the normal is only kept for t < ray.t anyway
		*normalOut = normalize( cross( mk3( r0.w, r1.x, r1.y ), mk3( r1.z, r1.w, r2.x ) ) );
		return t;
	}

	return phongTessTriAndRayIntersect(
		mk3( pa.x, pa.y, pa.z ), mk3( pb.x, pb.y, pb.z ), mk3( pc.x, pc.y, pc.z ),
		mk3( na.x, na.y, na.z ), mk3( nb.x, nb.y, nb.z ), mk3( nc.x, nc.y, nc.z ),
		ray, rayT, tNear, tFar, P.phongAlpha, normalOut );
}

// intersectFaces / intersectFace, pt_bvh.cl:10-46, for one hit leaf
// EAGER: request the second face's record before testing the first (12 more registers: kernels with the lean budget)
template<bool PHONG = false, bool EAGER = false>
PT_DEV void testLeaf( const DevParams& P, int face0, int face1, const Ray& ray, float tNear, float tFar, Hit& hit, unsigned& faceTests ) {
	if( PHONG ) {
		f3 normal;
		float t = phongFaceT( P, face0, ray, hit.t, tNear, tFar, &normal );
		faceTests++;

		if( hit.t > t ) {
			hit.t = t;
			hit.face = face0;
			hit.normal = normal;
		}

		if( face1 != -1 ) {
			t = phongFaceT( P, face1, ray, hit.t, tNear, tFar, &normal );
			faceTests++;

			if( hit.t > t ) {
				hit.t = t;
				hit.face = face1;
				hit.normal = normal;
			}
		}

		return;
	}

	(void) tFar;
	// EAGER: both records are requested before the first test — a leaf's second face is the next record (the
	// reference's leaf order), so its fetch overlaps the first face's arithmetic instead of following it
	// (+2..4 % at 4 waves / SIMD; at 8 the 12 extra registers cost more than the latency)
	const TriRecord first = loadTri( P, face0 );
	TriRecord second;

	// (requested whether or not the leaf has a second face — the array is padded by one record.  Requesting it only for
	// two-face leaves, 70 % of them, saves a third of the leaf phase's loads on the others: hairball +0.8 %, Dragon- and
	// Sponza-class +-0, Cornell -0.9 % for the divergent branch — profiles/r03/experiments/eager_only_two_face_leaves.txt)
	if( EAGER ) {
		second = loadTri( P, face0 + 1 );
	}

	float t = triangleT( first, ray, hit.t, tNear );
	faceTests++;

	if( hit.t > t ) {
		hit.t = t;
		hit.face = face0;
	}

	if( face1 != -1 ) {
		if( !EAGER ) {
			second = loadTri( P, face1 );
		}

		t = triangleT( second, ray, hit.t, tNear );
		faceTests++;

		if( hit.t > t ) {
			hit.t = t;
			hit.face = face1;
		}
	}
}

// ---- the node stream and the LDS-resident tree top ------------------------------------------
// pbr_upload_scene turns the reference's DFS node array (pt_bvh.cl:96-102: a hit continues at
// index + 1, a miss at the miss link) into a stream of 32-B records with EXPLICIT successors, so
// that records may be stored in any order:
//   n0 = {min.x, min.y, max.x, max.y}     n1 = {min.z, max.z, w0, w1}
//   container  w0 = record to continue at when the box is hit    w1 = ... when it is missed
//   leaf       w0 = 1 << 31 | hasSecondFace << 30 | face0         w1 = record to continue at (hit or miss)
// A record reference is the record's BYTE OFFSET in the stream (record index * 32: the address operand of the LDS
// read and of the global load as it stands, no shift per visit); a reference < 0 ends the walk (the reference's
// `index > 0 && index < numNodes`, pt_bvh.cl:122).
// The stream starts with the most-visited nodes, ranked by expected visit frequency (surface area of the
// parent box); the rest follows in DFS order, so a cold node's hit successor is still the adjacent
// 32 B.  A block stages any prefix [0, numHot) of the stream in LDS: "is my next node resident, and
// where" is one compare on the reference itself.
struct Cursor {
	int ref;   // byte offset of a record in the node stream; < 0: the walk has ended
};

PT_DEV bool alive( Cursor c ) {
	return c.ref >= 0;
}

struct NodeLinks {
	bool leaf;
	int face0, face1;   // leaf only; face1 = -1 if single
	Cursor onHit, onMiss;
};

template<bool USE_LDS>
PT_DEV void fetchNode( const DevParams& P, const float4* lds, Cursor c, float4* n0, float4* n1 ) {
	if( USE_LDS && c.ref < P.numHotBytes ) {
		const float4* rec = (const float4*) ( (const char*) lds + c.ref );
		*n0 = rec[0];
		*n1 = rec[1];
	}
	else {
		const float4* rec = (const float4*) ( (const char*) P.nodes + (size_t) (unsigned) c.ref );
		*n0 = rec[0];
		*n1 = rec[1];
	}
}

PT_DEV int leafFace0( int w0 ) {
	return w0 & 0x3FFFFFFF;
}

PT_DEV int leafFace1( int w0 ) {
	return ( w0 & 0x40000000 ) ? ( w0 & 0x3FFFFFFF ) + 1 : -1;
}

PT_DEV NodeLinks decodeNode( const float4 n1 ) {
	const int w0 = __float_as_int( n1.z );
	const int w1 = __float_as_int( n1.w );
	NodeLinks n;
	n.leaf = ( w0 < 0 );
	n.face0 = leafFace0( w0 );
	n.face1 = leafFace1( w0 );
	n.onHit.ref = n.leaf ? w1 : w0;
	n.onMiss.ref = w1;
	return n;
}

// Which of the successor sets a ray walks (pbr_config.traversal; the statement both ends follow is in pbr_upload's
// buildWalkStreams and in the oracle's "Ray-ordered walk").  Scheme 1: 2 * dominant axis (x before y before z on ties)
// + ( dir[axis] < 0 ).  Scheme 2: the sign bits x | y << 1 | z << 2.
PT_DEV int walkOrderOf( int scheme, const f3 d ) {
	// both formulas, then a select on the (wave-uniform) scheme: as a branch around one of them this cost the 80-register
	// state machine 16 B of scratch
	const int eight = ( ( d.x < 0.0f ) ? 1 : 0 ) | ( ( d.y < 0.0f ) ? 2 : 0 ) | ( ( d.z < 0.0f ) ? 4 : 0 );
	const float ax = __builtin_fabsf( d.x ), ay = __builtin_fabsf( d.y ), az = __builtin_fabsf( d.z );
	const bool xDominates = ( ax >= ay && ax >= az ), yDominates = ( ay >= az );
	const int six = xDominates ? ( ( d.x < 0.0f ) ? 1 : 0 ) : ( yDominates ? ( ( d.y < 0.0f ) ? 3 : 2 ) : ( ( d.z < 0.0f ) ? 5 : 4 ) );
	return ( scheme >= 2 ) ? eight : six;
}

PT_DEV Cursor firstNode( const DevParams& P, const f3 dir ) {
	// the walk starts at node 1 (pt_bvh.cl:84) — or, with a ray-ordered walk, at the root's first child in the ray's order:
	// one 4-byte load per walk from the table in front of the stream, an L1 hit.  Which of the two is a property of the
	// build flavour (pt_flavour.hpp): as a run-time branch it cost every kernel registers, the 80-register ones spills.
	Cursor c;
#if PT_WALK_MODE == 0
	(void) dir;
	c.ref = P.firstRef;
#elif PT_WALK_MODE == 1
	c.ref = P.walkTable[walkOrderOf( P.walkScheme, dir )];
#else
	c.ref = P.firstRef;

	if( P.walkScheme != 0 ) {
		c.ref = P.walkTable[walkOrderOf( P.walkScheme, dir )];
	}
#endif
	return c;
}

// ---- the compact record of the eight-order walk (round 6; walkScheme 3, build flavour bit 2) -----------------------
// The eight-order walk as eight streams of 32-byte records costs eight times the node memory.  A container sorts its
// children on ITS OWN axis, ascending or descending by one bit of the ray's order k — so whatever k is, a hit container
// continues at one of TWO children, and only the word "where to continue after this node" differs between all eight orders
// (the last child of a container inherits its parent's, which depends on the ancestors' axes).  One 64-byte record per node:
//   n0 = {min.x, min.y, max.x, max.y}     n1 = {min.z, max.z, h0, h1}     next[8]
//   container  h0 = its first child in ascending order, h1 = its first child in descending order | 4 << axis
//   leaf       h0 = the leaf word (as in the 32-byte record: 1 << 31 | hasSecondFace << 30 | face0), h1 = 0
//   next[k]    the record to continue at in order k when the box is missed, or the node is a leaf
// References are byte offsets of 64-byte records (bits 0 - 5 clear).  A ray of order k carries kOff = 32 + 4 k: the byte
// offset of ITS next word within a record and, in bits 2 - 4, its order as a mask against h1's axis bit:
//   descending = h1 & kOff;  child = ( descending ? h1 : h0 ) - descending.
// Same visits, same tests, same hits as the eight streams (the oracle's ray-ordered walk states both): a quarter of the memory,
// one more load (4 bytes) and five more vector instructions per visit.
#define PT_COMPACT_ON( P ) ( PT_WALK_COMPACT == 1 || ( PT_WALK_COMPACT == 2 && ( P ).walkScheme == 3 ) )

PT_DEV int walkCompactOffset( const f3 d ) {
	return 32 + 4 * ( ( ( d.x < 0.0f ) ? 1 : 0 ) | ( ( d.y < 0.0f ) ? 2 : 0 ) | ( ( d.z < 0.0f ) ? 4 : 0 ) );
}

template<bool USE_LDS>
PT_DEV void fetchNodeCompact( const DevParams& P, const float4* lds, Cursor c, int kOff, float4* n0, float4* n1, int* next ) {
	if( USE_LDS && c.ref < P.numHotBytes ) {
		const char* rec = (const char*) lds + c.ref;
		*n0 = ( (const float4*) rec )[0];
		*n1 = ( (const float4*) rec )[1];
		*next = *(const int*) ( rec + kOff );
	}
	else {
		const char* rec = (const char*) P.nodes + (size_t) (unsigned) c.ref;
		*n0 = ( (const float4*) rec )[0];
		*n1 = ( (const float4*) rec )[1];
		*next = *(const int*) ( rec + kOff );
	}
}

PT_DEV NodeLinks decodeNodeCompact( const float4 n1, int next, int kOff ) {
	const int h0 = __float_as_int( n1.z );
	const int h1 = __float_as_int( n1.w );
	const int descending = h1 & kOff;
	NodeLinks n;
	n.leaf = ( h0 < 0 );
	n.face0 = leafFace0( h0 );
	n.face1 = leafFace1( h0 );
	n.onHit.ref = n.leaf ? next : ( ( descending != 0 ) ? h1 : h0 ) - descending;
	n.onMiss.ref = next;
	return n;
}

// ---- the node phase, hand-scheduled -------------------------------------------------------------
// One node phase of traverse() (below) for the lanes that are walking: fetch the node (LDS or
// memory), slab test, follow the hit / miss record, until `keep` or fewer lanes are still walking.
// Lanes that stop on a hit leaf return parked = 1 with the leaf's w0 word, tNear and tFar.
//
// Why assembly: compiled from C++, this loop carries its three per-lane conditions (walking,
// parked, resident in LDS) as 64-bit scalar masks that are merged by ~35 scalar instructions per
// iteration — and on gfx950 a scalar instruction costs 4.8 cycles of a SIMD's issue time against 2.4
// for a vector one (scripts/micro/valu_rate.hip), so the compiled loop is bound by the scalar unit
// (measured: 74 % busy on the Sponza-class scene, vector ALU 55 %).  Written by hand the lanes that
// leave are simply dropped from EXEC, their registers stay as they were, and 13 scalar
// instructions remain.  The vector arithmetic is, instruction for instruction, what hipcc emits
// for boxHit<ANYHIT>() + the record decode (same operations, same operand order, IEEE mode
// unchanged), so results are bit-identical to the C++ loop, which stays in use for PBR_GUARD builds,
// for USE_LDS = false and as the statement of what this does.
// Registers v46-v63 are the block's temporaries: n0 = v[46:49], n1 = v[50:53], the three slab
// pairs v[54:59], tNear / tFar v60 / v61, scratch v62 / v63.  A record reference is its byte offset (see above) and
// the staged prefix sits at LDS address 0 (the kernels have no other __shared__ data; checked in stageHotNodes),
// so the reference itself is the address operand of both the LDS read and the global load.
// The hit condition (pt_bvh.cl:107-110) is a chain of v_cmpx: each compare narrows EXEC to the lanes that still
// qualify, so no scalar instruction merges the three masks — 8 scalar + 22 vector instructions per visit (round 3: 24 → 22,
// see the note at PT_NODE_PHASE_TAIL_BEGIN; bit-identical and NOT faster — 64.4 ms either way on the Sponza-class scene: the node
// phase waits for its slowest lane's fetch, it is not bound by vector issue).
//
// Cache policy of the two loads, measured in round 3 (profiles/r03/experiments/node_load_cache_policy.txt): default as
// it stands; `nt` 0.60 / 0.41 / 0.34x (Sponza- / Dragon-class / hairball), `sc0` the same within noise, `sc1` and
// `sc0 sc1` 0.92 / 0.78 / 0.63x.
//
// Rejected after measurement (bit-identical, slower): requesting a parked lane's first face record from inside this
// loop, into its own lanes of the temporaries (registers are per lane) — whether at once or behind the next
// iteration's node loads with s_waitcnt vmcnt(3): the 6 scalar + 3 vector + 3 memory instructions it adds to every
// iteration cost more (Sponza-class -11 %, Dragon-class -8 %, hairball -13 %) than the one latency per leaf phase it hides.
#if !defined( PBR_GUARD ) && !defined( PBR_NODE_PHASE_CXX )
#define PT_NODE_PHASE_ASM 1

// (Round 4's PARKED_ONLY variant — the phase ends once `keep` lanes have PARKED; measured slower, profiles/r04/experiments/leaf_phase.txt —
// lived here as a second loop tail until the loop was pipelined; it is in the history of this file.)
template<bool ANYHIT>
PT_DEV void nodePhaseAsm(
	const DevParams& P, const f2v oxy, const f2v ozz, const f2v ixy, const f2v izz, float rayT, int keep,
	int& ref, unsigned& visits, int& leafWord, float& leafTNear, float& leafTFar, int& parked
) {
	const float eps = EPSILON5;
	keep = __builtin_amdgcn_readfirstlane( keep );       // wave-uniform by construction; make it a scalar register
	unsigned long long saved, active, parkMask, mA;
	int count;

	// The loop is software-pipelined (round 4): as soon as a visit has decided which lanes go on, their next records are
	// requested — the exit test runs under that request, not in front of it.  What stands between a record's arrival and the
	// next request is what bounds the walk (profiles/r04/experiments/padding_sensitivity.txt: an instruction behind the wait
	// costs three times what one in front of it costs; pt_dual.hpp gained 2 % from the same reordering), and the exit test was
	// three scalar instructions and the loop's taken branch of it.  When the phase ends with a request on its way, the lanes it
	// is for take that visit too (a node phase may always run one visit longer: per lane the sequence of visits is the same).
#define PT_NODE_PHASE_FETCH \
		"v_cmp_gt_i32 vcc, %[numHotBytes], v53\n" \
		"s_and_saveexec_b64 %[active], vcc\n" \
		"ds_read_b128 v[46:49], v53\n" \
		"ds_read_b128 v[50:53], v53 offset:16\n" \
		"s_xor_b64 exec, exec, %[active]\n" \
		"global_load_dwordx4 v[46:49], v53, %[nodes]\n" \
		"global_load_dwordx4 v[50:53], v53, %[nodes] offset:16\n" \
		"s_mov_b64 exec, %[active]\n"
	// EXEC = the lanes whose box is hit.  A hit container continues at w0, everything else at w1;
	// the lanes on a hit leaf park; then the lanes that go on: alive and not parked.
	// Round 3: the cursor lives in v53 for the whole phase — the record's last word IS the reference to continue at
	// unless the box is a hit container, so the load that fetches a record also advances the cursor (a load may
	// overwrite its own address register), and the per-visit copy of w1 is gone; and tFar is min3 of the three slab
	// exits as they stand: pt_intersect.cl's fmin( ., INFINITY ) on the third only matters when all three are NaN, and
	// then tNear is NaN too and the box is missed either way (the C++ statement below keeps the reference's form).
	// 22 vector + 8 scalar instructions per visit.
#define PT_NODE_PHASE_VISIT( cull ) \
		"v_add_u32 %[visits], 1, %[visits]\n" \
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
		"v_pk_add_f32 v[54:55], v[46:47], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n" \
		"v_pk_add_f32 v[56:57], v[48:49], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n" \
		"v_pk_add_f32 v[58:59], v[50:51], %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n" \
		"v_pk_mul_f32 v[54:55], %[ixy], v[54:55]\n" \
		"v_pk_mul_f32 v[56:57], %[ixy], v[56:57]\n" \
		"v_pk_mul_f32 v[58:59], %[izz], v[58:59]\n" \
		"v_min_f32 v60, v54, v56\n" \
		"v_min_f32 v61, v55, v57\n" \
		"v_min_f32 v62, v58, v59\n" \
		"v_max3_f32 v60, v60, v61, v62\n" \
		"v_max_f32 v61, v54, v56\n" \
		"v_max_f32 v63, v58, v59\n" \
		"v_max_f32 v62, v55, v57\n" \
		"v_min3_f32 v61, v61, v62, v63\n" \
		"v_cmpx_lt_f32 %[eps], v61\n" \
		cull \
		"v_cmp_gt_i32 vcc, 0, v52\n" \
		"v_cndmask_b32 v53, v52, v53, vcc\n" \
		"s_or_b64 %[parkMask], %[parkMask], vcc\n" \
		"s_mov_b64 exec, %[active]\n" \
		"v_cmp_le_i32 %[mA], 0, v53\n" \
		"s_andn2_b64 exec, %[mA], vcc\n"
#define PT_NODE_PHASE_LOOP( cull ) \
		"s_mov_b64 %[saved], exec\n" \
		"s_mov_b64 %[parkMask], 0\n" \
		"v_mov_b32 v53, %[ref]\n" \
		PT_NODE_PHASE_FETCH \
	"1:\n" \
		PT_NODE_PHASE_VISIT( cull ) \
		"s_cbranch_scc0 3f\n"                                 /* nobody goes on: nothing to request */ \
		PT_NODE_PHASE_FETCH \
		"s_bcnt1_i32_b64 %[count], exec\n" \
		"s_cmp_gt_i32 %[count], %[keep]\n" \
		"s_cbranch_scc1 1b\n" \
		PT_NODE_PHASE_VISIT( cull )                            /* the phase ends; the records on their way are not dropped */ \
	"3:\n" \
		"s_mov_b64 exec, %[saved]\n" \
		"v_mov_b32 %[ref], v53\n" \
		"v_cndmask_b32 %[parked], 0, 1, %[parkMask]\n" \
		"v_mov_b32 %[leafWord], v52\n" \
		"v_mov_b32 %[leafTNear], v60\n" \
		"v_mov_b32 %[leafTFar], v61\n"

#define PT_NODE_PHASE_OPERANDS \
		: [ref] "+v"( ref ), [visits] "+v"( visits ), [leafWord] "=v"( leafWord ), [leafTNear] "=v"( leafTNear ), [leafTFar] "=v"( leafTFar ), [parked] "=v"( parked ), \
		  [saved] "=&s"( saved ), [active] "=&s"( active ), [parkMask] "=&s"( parkMask ), [mA] "=&s"( mA ), [count] "=&s"( count ) \
		: [oxy] "v"( oxy ), [ozz] "v"( ozz ), [ixy] "v"( ixy ), [izz] "v"( izz ), [rayT] "v"( rayT ), [keep] "s"( keep ), \
		  [numHotBytes] "s"( P.numHotBytes ), [nodes] "s"( P.nodes ), [eps] "s"( eps ) \
		: "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "vcc", "scc"

	if( ANYHIT ) {
		// traverseShadows: no `ray.t > tNear` cull (pt_bvh.cl:151-154)
		asm volatile(
			PT_NODE_PHASE_LOOP( "v_cmpx_le_f32 v60, v61\n" )
			PT_NODE_PHASE_OPERANDS
		);
	}
	else {
		asm volatile(
			PT_NODE_PHASE_LOOP( "v_cmpx_gt_f32 %[rayT], v60\n" "v_cmpx_le_f32 v60, v61\n" )
			PT_NODE_PHASE_OPERANDS
		);
	}

#undef PT_NODE_PHASE_FETCH
#undef PT_NODE_PHASE_VISIT
#undef PT_NODE_PHASE_LOOP
#undef PT_NODE_PHASE_OPERANDS
}

// The same node phase over compact records (above): the cursor lives in v45 — the load of the ray's own next word overwrites
// it (a load may overwrite a register that earlier loads of the same phase used as their address: they have issued), so a
// missed box or a leaf finds its successor in place, and a hit container takes one of its two first children instead.
// 27 vector + 8 scalar instructions and three loads per visit.
template<bool ANYHIT>
PT_DEV void nodePhaseAsmCompact(
	const DevParams& P, const f2v oxy, const f2v ozz, const f2v ixy, const f2v izz, float rayT, int keep, int kOff,
	int& ref, unsigned& visits, int& leafWord, float& leafTNear, float& leafTFar, int& parked
) {
	const float eps = EPSILON5;
	keep = __builtin_amdgcn_readfirstlane( keep );
	unsigned long long saved, active, parkMask, mA, mB;
	int count;

#define PT_NODE_PHASE_FETCH \
		"v_add_u32 v62, v45, %[kOff]\n" \
		"v_cmp_gt_i32 vcc, %[numHotBytes], v45\n" \
		"s_and_saveexec_b64 %[active], vcc\n" \
		"ds_read_b128 v[46:49], v45\n" \
		"ds_read_b128 v[50:53], v45 offset:16\n" \
		"ds_read_b32 v45, v62\n" \
		"s_xor_b64 exec, exec, %[active]\n" \
		"global_load_dwordx4 v[46:49], v45, %[nodes]\n" \
		"global_load_dwordx4 v[50:53], v45, %[nodes] offset:16\n" \
		"global_load_dword v45, v62, %[nodes]\n" \
		"s_mov_b64 exec, %[active]\n"
#define PT_NODE_PHASE_VISIT( cull ) \
		"v_add_u32 %[visits], 1, %[visits]\n" \
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
		"v_pk_add_f32 v[54:55], v[46:47], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n" \
		"v_pk_add_f32 v[56:57], v[48:49], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n" \
		"v_pk_add_f32 v[58:59], v[50:51], %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n" \
		"v_pk_mul_f32 v[54:55], %[ixy], v[54:55]\n" \
		"v_pk_mul_f32 v[56:57], %[ixy], v[56:57]\n" \
		"v_pk_mul_f32 v[58:59], %[izz], v[58:59]\n" \
		"v_min_f32 v60, v54, v56\n" \
		"v_min_f32 v61, v55, v57\n" \
		"v_min_f32 v62, v58, v59\n" \
		"v_max3_f32 v60, v60, v61, v62\n" \
		"v_max_f32 v61, v54, v56\n" \
		"v_max_f32 v63, v58, v59\n" \
		"v_max_f32 v62, v55, v57\n" \
		"v_min3_f32 v61, v61, v62, v63\n" \
		"v_cmpx_lt_f32 %[eps], v61\n" \
		cull \
		"v_cmp_gt_i32 vcc, 0, v52\n" \
		"v_and_b32 v62, v53, %[kOff]\n" \
		"v_cmp_ne_u32 %[mB], 0, v62\n" \
		"v_cndmask_b32 v63, v52, v53, %[mB]\n" \
		"v_sub_u32 v63, v63, v62\n" \
		"v_cndmask_b32 v45, v63, v45, vcc\n" \
		"s_or_b64 %[parkMask], %[parkMask], vcc\n" \
		"s_mov_b64 exec, %[active]\n" \
		"v_cmp_le_i32 %[mA], 0, v45\n" \
		"s_andn2_b64 exec, %[mA], vcc\n"
#define PT_NODE_PHASE_LOOP( cull ) \
		"s_mov_b64 %[saved], exec\n" \
		"s_mov_b64 %[parkMask], 0\n" \
		"v_mov_b32 v45, %[ref]\n" \
		PT_NODE_PHASE_FETCH \
	"1:\n" \
		PT_NODE_PHASE_VISIT( cull ) \
		"s_cbranch_scc0 3f\n" \
		PT_NODE_PHASE_FETCH \
		"s_bcnt1_i32_b64 %[count], exec\n" \
		"s_cmp_gt_i32 %[count], %[keep]\n" \
		"s_cbranch_scc1 1b\n" \
		PT_NODE_PHASE_VISIT( cull ) \
	"3:\n" \
		"s_mov_b64 exec, %[saved]\n" \
		"v_mov_b32 %[ref], v45\n" \
		"v_cndmask_b32 %[parked], 0, 1, %[parkMask]\n" \
		"v_mov_b32 %[leafWord], v52\n" \
		"v_mov_b32 %[leafTNear], v60\n" \
		"v_mov_b32 %[leafTFar], v61\n"

#define PT_NODE_PHASE_OPERANDS \
		: [ref] "+v"( ref ), [visits] "+v"( visits ), [leafWord] "=v"( leafWord ), [leafTNear] "=v"( leafTNear ), [leafTFar] "=v"( leafTFar ), [parked] "=v"( parked ), \
		  [saved] "=&s"( saved ), [active] "=&s"( active ), [parkMask] "=&s"( parkMask ), [mA] "=&s"( mA ), [mB] "=&s"( mB ), [count] "=&s"( count ) \
		: [oxy] "v"( oxy ), [ozz] "v"( ozz ), [ixy] "v"( ixy ), [izz] "v"( izz ), [rayT] "v"( rayT ), [kOff] "v"( kOff ), [keep] "s"( keep ), \
		  [numHotBytes] "s"( P.numHotBytes ), [nodes] "s"( P.nodes ), [eps] "s"( eps ) \
		: "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "vcc", "scc"

	if( ANYHIT ) {
		asm volatile(
			PT_NODE_PHASE_LOOP( "v_cmpx_le_f32 v60, v61\n" )
			PT_NODE_PHASE_OPERANDS
		);
	}
	else {
		asm volatile(
			PT_NODE_PHASE_LOOP( "v_cmpx_gt_f32 %[rayT], v60\n" "v_cmpx_le_f32 v60, v61\n" )
			PT_NODE_PHASE_OPERANDS
		);
	}

#undef PT_NODE_PHASE_FETCH
#undef PT_NODE_PHASE_VISIT
#undef PT_NODE_PHASE_LOOP
#undef PT_NODE_PHASE_OPERANDS
}

// Round 4 measured more node phases — the adjacent record fetched along (nodePhasePair), paired half-record fetches
// (nodePhaseHalves), polled LDS-DMA slots (nodePhaseAsync) and two walks per lane (nodePhaseDual) — all bit-identical; for ONE walk
// per lane none is faster than this one (lab/src/pt_r04_node_phases.hpp, profiles/r04/experiments/; lab builds, -DPBR_LAB, compile
// them in up to round 4's last commit; the product never did).  Two walks per lane with their node phases software-pipelined became plan 6: pt_dual.hpp.
#endif

// traverse (pt_bvh.cl:82-123) / traverseShadows (:133-177).  ANYHIT: the shadow variant — no
// `ray.t > tNear` cull, stops at the first face hit nearer than the light.  Per lane this is
// exactly the reference's sequence of node visits and face tests.
//
// Loop shape: two phases per round.  Measured on MI355X (scripts/ab.py with -DPBR_EXP_STATS), a
// plain lock-step loop enters the face tests in 30-70 % of its iterations with only 2-7 of the 64
// lanes standing on a hit leaf — and the face tests are 3-4x as long as the slab test.  So a lane that
// hits a leaf PARKS (remembers the leaf, stops walking) while the others walk on; once a share
// P.parkEighths / 8 of the lanes that entered the node phase have left it (parked or finished), all
// parked lanes test their faces together (measured best at 1080p: 6/8 for the Cornell box, whose
// 35-node tree has a leaf every 5 visits; 4/8 for the 260k - 2M triangle scenes).  With few lanes left the share rounds
// to one lane, i.e. the plain lock-step walk.  Rejected after measurement (bit-identical, slower):
// waiting until EVERY lane stands on a leaf (dragon-class 0.65x), and requesting both successors
// of a node before the slab test (a loss once registers are tight).
template<bool ANYHIT, bool LIGHTS, bool USE_LDS, bool PHONG = false, bool EAGER = false>
PT_DEV void traverse( const DevParams& P, const float4* lds, const Ray& ray, Hit& hit, unsigned& nodeVisits, unsigned& faceTests ) {
	const f3 invDir = mk3( div1( 1.0f, ray.dir.x ), div1( 1.0f, ray.dir.y ), div1( 1.0f, ray.dir.z ) );
	const float tLight = hit.t;
	Cursor cur = firstNode( P, ray.dir );
#ifdef PBR_GUARD_TRAV
	const int numNodes = P.numNodes;
	int guardSteps = 0;
#endif
	PT_LAB_TRAVERSE_BEGIN

	if( LIGHTS ) {
		traverseLights( P, ray, hit );
	}

#ifdef PT_NODE_PHASE_ASM
	const f2v oxy = { ray.origin.x, ray.origin.y };
	const f2v ozz = { ray.origin.z, ray.origin.z };
	const f2v ixy = { invDir.x, invDir.y };
	const f2v izz = { invDir.z, invDir.z };
#endif
	const int kOff = PT_COMPACT_ON( P ) ? walkCompactOffset( ray.dir ) : 0;
	(void) kOff;
	bool walking = true;   // the walk always visits node 1 (pt_bvh.cl:84-88)
	unsigned visits = 0;
	int leafWord = 0;
	float leafTNear = 0.0f, leafTFar = 0.0f;   // tFar: Phong tessellation only (pt_phongtess.cl:202)

	for( ;; ) {
		bool parked = false;

		// ---- node phase: the lanes that are walking
		if( walking ) {
			const int entered = __popcll( __ballot( 1 ) );
			const int leave = ( entered * P.parkEighths ) >> 3;
			const int keep = entered - ( ( leave < 1 ) ? 1 : leave );
#ifdef PT_NODE_PHASE_ASM
			if( USE_LDS ) {
				int parkedFlag;
				__builtin_amdgcn_s_setprio( PT_WALK_PRIO );
				if( PT_COMPACT_ON( P ) ) {
					nodePhaseAsmCompact<ANYHIT>( P, oxy, ozz, ixy, izz, hit.t, keep, kOff, cur.ref, visits, leafWord, leafTNear, leafTFar, parkedFlag );
				}
				else {
					nodePhaseAsm<ANYHIT>( P, oxy, ozz, ixy, izz, hit.t, keep, cur.ref, visits, leafWord, leafTNear, leafTFar, parkedFlag );
				}
				__builtin_amdgcn_s_setprio( 0 );
				parked = ( parkedFlag != 0 );
				walking = alive( cur );
			}
			else
#endif
			do {
#ifdef PBR_GUARD_TRAV
				// a forward-only walk visits each node at most once
				if( ++guardSteps > numNodes ) {
					atomicAdd( &P.guard[2], 1u );
					walking = false;
					break;
				}
#endif
				visits++;
				PT_LAB_NODE_ITERATION

				float4 n0, n1;
				int w0, w1;

				if( PT_COMPACT_ON( P ) ) {
					// the compact record: w0 = the leaf word or the ray's first child, w1 = the ray's next word
					int next;
					fetchNodeCompact<USE_LDS>( P, lds, cur, kOff, &n0, &n1, &next );
					const NodeLinks links = decodeNodeCompact( n1, next, kOff );
					w0 = links.leaf ? __float_as_int( n1.z ) : links.onHit.ref;
					w1 = next;
				}
				else {
					fetchNode<USE_LDS>( P, lds, cur, &n0, &n1 );
					w0 = __float_as_int( n1.z );
					w1 = __float_as_int( n1.w );
				}

				float tNear, tFar;

				// hit container -> w0; miss, or leaf -> w1
				const bool isHit = boxHit<ANYHIT>( n0, n1, ray, invDir, hit.t, &tNear, &tFar );
				const bool isLeaf = ( w0 < 0 );
				cur.ref = ( isHit && !isLeaf ) ? w0 : w1;

				if( isHit && isLeaf ) {
					parked = true;
					leafWord = w0;
					leafTNear = tNear;
					leafTFar = tFar;
				}

				walking = alive( cur );
			} while( walking && !parked && __popcll( __ballot( walking && !parked ) ) > keep );
		}

		// ---- leaf phase: intersectFaces (pt_bvh.cl:10-46) for every parked lane
		if( parked ) {
			PT_LAB_LEAF_PHASE
			testLeaf<PHONG, EAGER>( P, leafFace0( leafWord ), leafFace1( leafWord ), ray, leafTNear, leafTFar, hit, faceTests );

			if( ANYHIT && hit.t < tLight ) {
				walking = false;
			}
		}

		if( __ballot( walking ) == 0ull ) {
			break;
		}
	}

	if( !ANYHIT ) {
		nodeVisits += visits;
	}

	PT_LAB_TRAVERSE_END( P, ANYHIT )
}

// Geometric normal of a face: fast_normalize( cross( edge1, edge2 ) ), pt_intersect.cl:122
template<bool STORED = false>
PT_DEV f3 faceNormal( const DevParams& P, int face, int* material ) {
	// (native arithmetic: every plan recomputes — the stored normals are prepareFaceNormals' exact ones, and the plans of a
	// mode must agree on the bits so that the tuner's choice never shows in the image)
	if( STORED && !PT_ARITH_NATIVE && P.faceN != nullptr ) {
		const float4 r = P.faceN[face];
		*material = __float_as_int( r.w );
		return mk3( r.x, r.y, r.z );
	}

	const float4 r0 = P.tris[face * 3 + 0];
	const float4 r1 = P.tris[face * 3 + 1];
	const float4 r2 = P.tris[face * 3 + 2];
	*material = __float_as_int( r2.y );
	return normalize( cross( mk3( r0.w, r1.x, r1.y ), mk3( r1.z, r1.w, r2.x ) ) );
}


// ---------------------------------------------------------------------------------------
// Camera rays
// ---------------------------------------------------------------------------------------

// initRay + antiAliasing + depthOfField, pathtracing.cl:25-48, pt_utils.cl:327-373
PT_DEV Ray initRay( const DevParams& P, int px, int py, float& seed, float tFocus, float tObject ) {
	const f3 cu = ld3( P.cu );
	const f3 cv = ld3( P.cv );
	const float fx = 2.0f * (float) px;
	const float fy = 2.0f * (float) py;

	// pathtracing.cl:33-39, term by term: cu - cu * W (P.camA) and cv * H (P.cvH) do not depend on the pixel
	f3 inner = ld3( P.camA );
	inner = inner + cu * fx;
	inner = inner + cv;
	inner = inner - ld3( P.cvH );
	inner = inner + cv * fy;

	const float s = P.halfPx;
	const f3 initial = ld3( P.cw ) + inner * s;

	Ray ray;
	ray.origin = ld3( P.eye );
	ray.dir = normalize( initial );

	// antiAliasing
	const float r = rnd( seed );
	const float phi = PI_X2 * rnd( seed );
	const f3 aaDir = jitter( ray.dir, phi, sqrt1( r ), sqrt1( 1.0f - r ) );
	ray.dir = normalize( ray.dir + ( aaDir * P.pxDim ) * P.antiAliasing );

	if( tFocus >= 0.0f && tObject >= 0.0f ) {
		if( tObject == inff() ) {
			tObject = 1000.0f;
		}
		if( tFocus == inff() ) {
			tFocus = 1000.0f;
		}

		if( tObject > 0.0f ) {
			const float aperture = P.aperture;
			const float radius = rnd( seed ) * aperture * 0.5f;
			const float angle = PI_X2 * rnd( seed );
			float sa, ca;
			sincos( angle, &sa, &ca );
			const float x = radius * ca;
			const float y = radius * sa;

			ray.origin = ( ray.origin + cu * x ) + cv * y;

			const f3 hitFocalPlane = fma3( tFocus, ray.dir, ld3( P.eye ) );
			ray.dir = normalize( hitFocalPlane - ray.origin );
		}
	}

	return ray;
}


// ---------------------------------------------------------------------------------------
// BRDF 0 — Schlick (pt_brdf.cl:2-209)
// ---------------------------------------------------------------------------------------

PT_DEV float schZ( float t, float r ) {
	const float x = 1.0f + r * t * t - t * t;
	return ( x == 0.0f ) ? 0.0f : div1( r, x * x );
}

PT_DEV float schA( float w, float p ) {
	const float p2 = p * p;
	const float w2 = w * w;
	const float x = p2 - p2 * w2 + w2;
	return ( x == 0.0f ) ? 0.0f : sqrt1( div1( p, x ) );
}

PT_DEV float schG( float v, float r ) {
	const float x = r - r * v + v;
	return ( x == 0.0f ) ? 0.0f : div1( v, x );
}

// brdfSchlick, pt_brdf.cl:125-150 with D / B2 (:71-112) inlined
PT_DEV float brdfSchlick( const Material& mtl, f3 outDir, f3 inDir, f3 normal, float* u, float* pdf, const TangentFrame* frame = nullptr ) {
	const f3 vOutV = -outDir;
	f3 un;

	if( frame != nullptr && frame->valid && sameBits( frame->n, normal ) ) {
		un = frame->u;
	}
	else {
		un = normalize( cross( yzx( normal ), normal ) );
	}

	const f3 h = normalize( vOutV + inDir );
	const float t = dot( h, normal );
	const float vIn = dot( inDir, normal );
	const float vOut = dot( vOutV, normal );
	const f3 hp = normalize( cross( cross( h, normal ), normal ) );
	const float w = dot( un, hp );

	*u = dot( h, vOutV );
	*pdf = div1( t, (float) ( (double) 4.0f * M_PI_D * (double) dot( vOutV, h ) ) );

	const float r = mtl.p3;  // rough
	const float p = mtl.p2;  // isotropy
	const float b = 4.0f * r * ( 1.0f - r );
	const float a = ( r < 0.5f ) ? 0.0f : 1.0f - b;
	const float c = ( r < 0.5f ) ? 1.0f - b : 0.0f;
	const float d = (float) ( (double) 4.0f * M_PI_D * (double) vOut * (double) vIn );
	const float lam = (float) ( (double) a * M_1_PI_D );
	float ani = 0.0f;

	if( !( b == 0.0f || d == 0.0f ) ) {
		const float gp = schG( vOut, r ) * schG( vIn, r );
		const float obstructed = gp * schZ( t, r ) * schA( w, p );
		const float reemission = 1.0f - gp;
		ani = div1( b, d ) * ( obstructed + reemission );
	}

	const float fres = ( vIn == 0.0f ) ? 0.0f : div1( c, vIn );

	return lam + ani + fres;
}

// newRaySchlick, pt_brdf.cl:160-208
PT_DEV f3 newRaySchlick( f3 dir, f3 normal, const Material& mtl, float& seed, TangentFrame* frame = nullptr ) {
	const float rough = mtl.p3;
	const float iso = mtl.p2;

	if( rough == 0.0f ) {
		return reflect( dir, normal );
	}

	const float a = rnd( seed );
	float b = rnd( seed );
	const float iso2 = iso * iso;
	const float alpha = acos1( sqrt1( div1( a, rough - a * rough + a ) ) );
	float edge, base;
	int mode;

	if( b < 0.25f ) { edge = 0.25f; mode = 0; }
	else if( b < 0.5f ) { edge = 0.5f; mode = 1; }
	else if( b < 0.75f ) { edge = 0.75f; mode = 2; }
	else { edge = 1.0f; mode = 3; }

	b = 1.0f - 4.0f * ( edge - b );
	const float b2 = b * b;
	base = (float) ( M_PI_2_D * (double) sqrt1( div1( iso2 * b2, 1.0f - b2 + b2 * iso2 ) ) );

	float phi = base;

	if( mode == 1 ) { phi = (float) ( M_PI_D - (double) base ); }
	else if( mode == 2 ) { phi = (float) ( M_PI_D + (double) base ); }
	else if( mode == 3 ) { phi = (float) ( (double) 2.0f * M_PI_D - (double) base ); }

	if( iso < 1.0f ) {
		phi = (float) ( (double) phi + M_PI_2_D );
	}

	float sa, ca;
	sincos( alpha, &sa, &ca );
	f3 fu, fv;
	tangentFrame( normal, &fu, &fv );

	if( frame != nullptr ) {
		frame->n = normal;
		frame->u = fu;
		frame->v = fv;
		frame->valid = true;
	}

	const f3 H = jitterUV( normal, fu, fv, phi, sa, ca );
	f3 out = reflect( dir, H );

	if( dot( out, normal ) <= 0.0f ) {
		out = jitterUV( normal, fu, fv, PI_X2 * rnd( seed ), sqrt1( a ), sqrt1( 1.0f - a ) );
	}

	return out;
}


// ---------------------------------------------------------------------------------------
// BRDF 1 — Shirley-Ashikhmin (pt_brdf.cl:211-331)
// ---------------------------------------------------------------------------------------

// brdfShirleyAshikhmin, pt_brdf.cl:228-268
template<bool CALLS = false>
PT_DEV void brdfSA(
	const Material& mtl, f3 outDir, f3 inDir, f3 normal,
	float* brdfSpec, float* brdfDiff, float* dotHK1, float* pdf, const TangentFrame* frame = nullptr
) {
	const float nu = mtl.p2, nv = mtl.p3;
	f3 un, vn;

	if( frame != nullptr && frame->valid && sameBits( frame->n, normal ) ) {
		un = frame->u;
		vn = frame->v;
	}
	else {
		tangentFrame( normal, &un, &vn );
	}

	const f3 k1 = inDir;
	const f3 k2 = -outDir;
	const f3 h = normalize( k1 + k2 );

	const float dotHU = dot( h, un );
	const float dotHV = dot( h, vn );
	const float dotHN = dot( h, normal );
	const float dotNK1 = dot( normal, k1 );
	const float dotNK2 = dot( normal, k2 );
	const float hk1 = dot( h, k1 );
	*dotHK1 = hk1;

	float ps_e = nu * dotHU * dotHU + nv * dotHV * dotHV;
	ps_e = ( dotHN == 1.0f ) ? 0.0f : div1( ps_e, 1.0f - dotHN * dotHN );
	const float ps0 = (float) ( (double) ( sqrt1( ( nu + 1.0f ) * ( nv + 1.0f ) ) * 0.125f ) * M_1_PI_D );
	const float ps1_num = powSelect<CALLS>( dotHN, ps_e );
	const float ps1 = div1( ps1_num, hk1 * fmax1( dotNK1, dotNK2 ) );

	float pd = mtl.Rd * 0.38750768752f;
	const float a = 1.0f - dotNK1 * 0.5f;
	const float b = 1.0f - dotNK2 * 0.5f;
	pd *= 1.0f - a * a * a * a * a;
	pd *= 1.0f - b * b * b * b * b;

	*brdfSpec = ps0 * ps1;
	*brdfDiff = pd;
	*pdf = div1( ps0 * ps1_num, hk1 );
}

// newRayShirleyAshikhmin, pt_brdf.cl:278-330
template<bool CALLS = false>
PT_DEV f3 newRaySA( f3 dir, f3 rayNormal, const Material& mtl, float& seed, TangentFrame* frame = nullptr ) {
	const float nu = mtl.p2, nv = mtl.p3;
	float a = rnd( seed );
	const float b = rnd( seed );
	float phi_flip = (float) M_PI_D;
	float phi_flipf = 1.0f;
	float aMax = 1.0f;

	if( a < 0.25f ) {
		aMax = 0.25f;
		phi_flip = 0.0f;
	}
	else if( a < 0.5f ) {
		aMax = 0.5f;
		phi_flipf = -1.0f;
	}
	else if( a < 0.75f ) {
		aMax = 0.75f;
	}
	else {
		phi_flip = (float) ( (double) 2.0f * M_PI_D );
		phi_flipf = -1.0f;
	}

	a = 1.0f - 4.0f * ( aMax - a );

	const float phi = atan1( sqrt1( div1( nu + 1.0f, nv + 1.0f ) ) * tan1( (float) ( M_PI_2_D * (double) a ) ) );
	const float phi_full = phi_flip + phi_flipf * phi;

	float sinphi, cosphi;
	sincos( phi, &sinphi, &cosphi );
	const float theta_e = div1( 1.0f, nu * cosphi * cosphi + nv * sinphi * sinphi + 1.0f );
	const float theta = acos1( powSelect<CALLS>( 1.0f - b, theta_e ) );

	const f3 normal = ( mtl.d < 1.0f || dot( rayNormal, -dir ) >= 0.0f ) ? rayNormal : -rayNormal;

	float st, ct;
	sincos( theta, &st, &ct );
	f3 u, v;
	tangentFrame( normal, &u, &v );

	if( frame != nullptr ) {
		frame->n = normal;
		frame->u = u;
		frame->v = v;
		frame->valid = true;
	}

	const f3 h = jitterUV( normal, u, v, phi_full, st, ct );
	const f3 spec = reflect( dir, h );
	const f3 diff = jitterUV( normal, u, v, PI_X2 * rnd( seed ), sqrt1( b ), sqrt1( 1.0f - b ) );

	return ( dot( spec, normal ) <= 0.0f ) ? diff : spec;
}

// refract, pt_utils.cl:436-465
PT_DEV f3 refract( f3 dir, f3 normal, const Material& mtl, float& seed ) {
	const bool into = ( dot( normal, -dir ) > 0.0f );
	const f3 nl = into ? normal : -normal;
	const float m1 = into ? NI_AIR : mtl.Ni;
	const float m2 = into ? mtl.Ni : NI_AIR;
	const float m = div1( m1, m2 );
	const float cosI = -dot( nl, dir );
	const float sinT2 = m * m * ( 1.0f - cosI * cosI );

	if( sinT2 >= 1.0f ) {
		return reflect( dir, nl );
	}

	const float sqrtCosT = sqrt1( 1.0f - sinT2 );
	const float r0 = div1( m1 - m2, m1 + m2 );
	const float c = ( m1 > m2 ) ? sqrtCosT : cosI;
	const float reflectance = fresnel( c, r0 * r0 );

	if( reflectance < rnd( seed ) ) {
		const float k = m * cosI - sqrtCosT;
		return dir * m + nl * k;
	}

	return reflect( dir, nl );
}

// getNewRay, pt_brdf.cl:344-378 (direction only; the origin is fma( t, dir, origin ))
template<int BRDF, bool CALLS = false>
PT_DEV f3 newRayDir( f3 dir, f3 normal, const Material& mtl, float& seed, bool& addDepth, TangentFrame* frame = nullptr ) {
	// && short-circuits: the random number is drawn only when d < 1
	bool doTransRefr = false;

	if( mtl.d < 1.0f ) {
		doTransRefr = ( mtl.d <= rnd( seed ) );
	}

	addDepth = addDepth || doTransRefr;

	if( doTransRefr ) {
		return refract( dir, normal, mtl, seed );
	}

	return ( BRDF == 0 ) ? newRaySchlick( dir, normal, mtl, seed, frame ) : newRaySA<CALLS>( dir, normal, mtl, seed, frame );
}

// The factor updateColor multiplies `color` by (pathtracing.cl:98-124 Schlick, :127-177 S-A).
template<int BRDF, bool CALLS = false, bool LATE = true>
PT_DEV f3 throughput( const Material& mtl, f3 outDir, f3 inDir, f3 normal, const TangentFrame* frame = nullptr ) {
	const float d = mtl.d;

	if( BRDF == 0 ) {
		float u, pdf;
		float brdf = brdfSchlick( mtl, outDir, inDir, normal, &u, &pdf, frame );
		brdf *= fmax1( dot( normal, inDir ), 0.0f );
		brdf = div1( brdf, pdf );

		const MaterialColours mc = materialColours<LATE>( mtl );
		const f3 f4 = mk3( fresnel( u, mc.Ks.x ), fresnel( u, mc.Ks.y ), fresnel( u, mc.Ks.z ) );
		const f3 k = mk3( f4.x * brdf * d + ( 1.0f - d ), f4.y * brdf * d + ( 1.0f - d ), f4.z * brdf * d + ( 1.0f - d ) );
		return mc.Kd * k;
	}

	float spec, diff, dotHK1, pdf;
	brdfSA<CALLS>( mtl, outDir, inDir, normal, &spec, &diff, &dotHK1, &pdf, frame );
	spec = div1( spec, pdf );
	diff = div1( diff, pdf );

	const float fr = fresnel( dotHK1, mtl.Rs );
	const MaterialColours mc = materialColours<LATE>( mtl );
	const f3 brdf_s = ( mc.Ks * spec ) * fr;
	const f3 brdf_d = ( mc.Kd * diff ) * ( 1.0f - mtl.Rs );
	f3 bc = brdf_s + brdf_d;
	bc = mk3( bc.x * d + ( 1.0f - d ), bc.y * d + ( 1.0f - d ), bc.z * d + ( 1.0f - d ) );
	const float maxRGB = max_cl( 1.0f, max_cl( bc.x, max_cl( bc.y, bc.z ) ) );
	bc = mk3( div1( bc.x, maxRGB ), div1( bc.y, maxRGB ), div1( bc.z, maxRGB ) );

	return mk3( clamp01( bc.x ), clamp01( bc.y ), clamp01( bc.z ) );
}

// The shadow-ray contribution to finalColor (pathtracing.cl:102-115 Schlick, :133-154 S-A).
// Returns false when |pdf| <= 1e-5 (no contribution, secondaryPaths unchanged).
template<int BRDF, bool CALLS = false, bool LATE = true>
PT_DEV bool shadowContribution(
	const Material& mtl, f3 outDir, f3 lightDir, f3 normal, f3 color, f3 lightRgb, f3* add, const TangentFrame* frame = nullptr
) {
	const float d = mtl.d;

	if( BRDF == 0 ) {
		float u, pdf;
		float brdf = brdfSchlick( mtl, outDir, lightDir, normal, &u, &pdf, frame );

		if( !( __builtin_fabsf( pdf ) > 0.00001f ) ) {
			return false;
		}

		brdf *= fmax1( dot( normal, lightDir ), 0.0f );
		brdf = div1( brdf, pdf );

		const MaterialColours mc = materialColours<LATE>( mtl );
		const f3 f4 = mk3( fresnel( u, mc.Ks.x ), fresnel( u, mc.Ks.y ), fresnel( u, mc.Ks.z ) );
		const f3 k = mk3( f4.x * brdf * d + ( 1.0f - d ), f4.y * brdf * d + ( 1.0f - d ), f4.z * brdf * d + ( 1.0f - d ) );
		*add = ( ( color * lightRgb ) * mc.Kd ) * k;
		return true;
	}

	float spec, diff, dotHK1, pdf;
	brdfSA<CALLS>( mtl, outDir, lightDir, normal, &spec, &diff, &dotHK1, &pdf, frame );

	if( !( __builtin_fabsf( pdf ) > 0.00001f ) ) {
		return false;
	}

	spec = div1( spec, pdf );
	diff = div1( diff, pdf );

	const float fr = fresnel( dotHK1, mtl.Rs );
	const MaterialColours mc = materialColours<LATE>( mtl );
	const f3 brdf_s = ( mc.Ks * spec ) * fr;
	const f3 brdf_d = ( mc.Kd * diff ) * ( 1.0f - mtl.Rs );
	f3 bc = brdf_s + brdf_d;
	bc = mk3( bc.x * d + ( 1.0f - d ), bc.y * d + ( 1.0f - d ), bc.z * d + ( 1.0f - d ) );
	const float maxRGB = max_cl( 1.0f, max_cl( bc.x, max_cl( bc.y, bc.z ) ) );
	bc = mk3( div1( bc.x, maxRGB ), div1( bc.y, maxRGB ), div1( bc.z, maxRGB ) );

	const f3 cl = mk3( clamp01( bc.x ), clamp01( bc.y ), clamp01( bc.z ) );
	*add = mk3(
		cl.x * lightRgb.x * d + ( 1.0f - d ),
		cl.y * lightRgb.y * d + ( 1.0f - d ),
		cl.z * lightRgb.z * d + ( 1.0f - d )
	);
	return true;
}


// ---------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------

PT_DEV unsigned waveMax( unsigned v ) {
	for( int off = 32; off > 0; off >>= 1 ) {
		const unsigned o = (unsigned) __shfl_xor( (int) v, off, 64 );
		v = ( o > v ) ? o : v;
	}
	return v;
}

PT_DEV unsigned waveSum( unsigned v ) {
	for( int off = 32; off > 0; off >>= 1 ) {
		v += __shfl_xor( v, off, 64 );
	}
	return v;
}

// Everything a lane carries for the pixel it is working on.
struct PixelState {
	unsigned slot;           // tile-major pixel slot: tile = slot >> 6, position in tile = slot & 63
	// frame
	int frame, sample;
	f3 finalColor;
	unsigned secondaryPaths;
	float focus;
	float seed;
	unsigned dbgNodes, dbgTris;
	// path
	f3 color;
	int depth, depthAdded;
	Ray ray;
};

struct LaneCounters {
	unsigned nodes, tris, hits, paths;
};

PT_DEV void flushCounters( const DevParams& P, LaneCounters& c ) {
	atomicAdd( &P.counters[0], (unsigned long long) c.nodes );
	atomicAdd( &P.counters[1], (unsigned long long) c.tris );
	atomicAdd( &P.counters[2], (unsigned long long) c.hits );
	atomicAdd( &P.counters[3], (unsigned long long) c.paths );
	c.nodes = c.tris = c.hits = c.paths = 0;
}

// Take up a unit of work — ONE frame of the pixel in `slot` — and start its first path.  The unit's {finalColor, focus}
// go to P.frameBuf; the running mean is folded afterwards, in frame order, by foldFrames.
// Image coordinates of a pixel slot of this rank: tileAtDealPosition with the divisions by tilesX as multiplications
// (a 32-bit division is ~20 instructions).  Recomputed where a camera ray starts — once per path — rather than
// kept in two registers for the whole path.
PT_DEV unsigned divInvariant( unsigned n, unsigned magic, unsigned shifts );

PT_DEV void pixelOfSlot( const DevParams& P, unsigned slot, int* px, int* py ) {
	const unsigned tilesX = (unsigned) P.tilesX;
	const unsigned position = ( slot >> 6 ) * (unsigned) P.tileWorld + (unsigned) P.tileRank;
	const unsigned ty = divInvariant( position, P.tilesXDiv[0], P.tilesXDiv[1] );
	unsigned tx = position - ty * tilesX;

	if( P.tileWorld > 1 ) {
		const unsigned turn = (unsigned) PT_DEAL_SHIFT * ty;
		const unsigned back = turn - divInvariant( turn, P.tilesXDiv[0], P.tilesXDiv[1] ) * tilesX;
		tx = ( tx >= back ) ? tx - back : tx - back + tilesX;
	}

	const int inTile = (int) ( slot & 63u );
	*px = (int) tx * 8 + ( inTile & 7 );
	*py = (int) ty * 8 + ( inTile >> 3 );
}

// Where {finalColor, focus} of frame k (of this launch) of a pixel slot lives in the frame buffer: rows of 8 slots
// (8 x 16 B = one 128-byte line), the rows of the nFrames frames of such a group one after the other.  The 64 lanes
// of a wave finish frames of one pixel (nextSlot), so their stores fall into one 8-KiB stretch (frame-major planes:
// 64 pages 33 MB apart, Cornell -1.5 %), and foldFrames, one thread per slot, still reads whole lines.
PT_DEV size_t frameBufIndex( const DevParams& P, unsigned slot, unsigned k ) {
	return ( (size_t) ( slot >> 3 ) * (size_t) P.nFrames + (size_t) k ) * 8u + (size_t) ( slot & 7u );
}

// getPreviousFocus, pathtracing.cl:58-65 (single-frame launches only; CLAMP_TO_EDGE): the previous frame's first-hit
// distance at the focus pixel and at this pixel, -1 = depth of field off.  imageIn does not change during a launch, so
// the two values are re-read where a camera ray starts instead of living in two registers for the whole path.
PT_DEV void focusInputs( const DevParams& P, unsigned slot, float* tFocus, float* tObject ) {
	*tFocus = -1.0f;
	*tObject = -1.0f;

	if( P.focusX >= 0 && P.focusY >= 0 ) {
		const int fx = ( P.focusX > P.width - 1 ) ? P.width - 1 : P.focusX;
		const int fy = ( P.focusY > P.height - 1 ) ? P.height - 1 : P.focusY;
		const int ft = ( fy >> 3 ) * P.tilesX + ( fx >> 3 );
		*tObject = P.imgIn[slot].w;
		*tFocus = P.focusGiven ? P.focusDepth : P.imgIn[(size_t) ft * 64 + (size_t) ( ( fy & 7 ) * 8 + ( fx & 7 ) )].w;
	}
}

PT_DEV void beginPixel( const DevParams& P, PixelState& st, unsigned slot, LaneCounters& cnt, unsigned frame ) {
	st.slot = slot;
	st.frame = (int) frame;
	st.sample = 0;
	st.finalColor = mk3( 0.0f, 0.0f, 0.0f );
	st.secondaryPaths = 1;
	st.focus = 0.0f;
	st.seed = P.seeds[frame];
	st.dbgNodes = 0;
	st.dbgTris = 0;

	st.color = mk3( 1.0f, 1.0f, 1.0f );
	st.depth = 0;
	st.depthAdded = 0;
	float tFocus, tObject;
	focusInputs( P, slot, &tFocus, &tObject );
	int px, py;
	pixelOfSlot( P, slot, &px, &py );
	st.ray = initRay( P, px, py, st.seed, tFocus, tObject );
	cnt.paths++;
}

// The finished unit's frame went to P.frameBuf in shadeStep; only the debug image is left to write (st.frame was advanced).
PT_DEV void finishPixel( const DevParams& P, const PixelState& st ) {
	// writeDebugImage, pathtracing.cl:73-78 (counters of the LAST frame of this launch)
	if( P.imgDbg != nullptr && st.frame == P.nFrames ) {
		P.imgDbg[st.slot] = make_float4( (float) st.dbgTris / 1082.0f, (float) st.dbgNodes / 1265.0f, 0.0f, 0.0f );
	}
}

// Everything of one bounce that follows the closest-hit traversal (pathtracing.cl:261-333):
// shade the hit, and — when the path ends — start the next path of the frame, or store the finished frame.  Returns
// true when the unit (one frame of one pixel) is finished; otherwise st.ray is the next ray to trace.
// CALLS: pow as a function call instead of inline (pt_math.hpp, pow1Call) — every kernel but the lean lock-step one
// LATE: the material's colours are read where they are used (Material, above)
template<int BRDF, bool SHADOW, bool LIGHTS, bool PHONG = false, bool EAGER = false, bool CALLS = !EAGER, bool FACEN = false, bool LATE = true>
PT_DEV bool shadeStep( const DevParams& P, const float4* lds, PixelState& st, LaneCounters& cnt, const Hit hit ) {
	// references keep the shading code below in the reference's vocabulary
	Ray& ray = st.ray;
	f3& color = st.color;
	f3& finalColor = st.finalColor;
	unsigned& secondaryPaths = st.secondaryPaths;
	float& seed = st.seed;
	int& depth = st.depth;
	int& depthAdded = st.depthAdded;
	unsigned& dbgTris = st.dbgTris;
	unsigned& totHits = cnt.hits;

	st.focus = ( st.sample + depth == 0 ) ? hit.t : st.focus;

		bool pathDone = false;
		f3 light = mk3( -1.0f, -1.0f, -1.0f );

		if( hit.t == inff() ) {
			// pathtracing.cl:263-266
			if( LIGHTS && hit.face < 0 ) {
				const float4 rgb = P.lights[( -( hit.face + 1 ) ) * 3 + 1];
				light = mk3( rgb.x, rgb.y, rgb.z );
			}
			else {
				light = ld3( P.sky );
			}

			pathDone = true;
		}
		else {
			int mtlIndex;
			f3 normal = faceNormal<FACEN>( P, hit.face, &mtlIndex );

			if( PHONG ) {
				normal = hit.normal;   // ray.normal as intersectFace stored it (pt_bvh.cl:18): the Phong normal on curved faces
			}
			const Material mtl = loadMaterial<LATE>( P, mtlIndex );
			totHits++;

			// extendDepth, pt_utils.cl:89-96
			bool addDepth;

			if( BRDF == 1 ) {
				addDepth = ( fmax1( mtl.p2, mtl.p3 ) >= 50.0f );
			}
			else {
				addDepth = ( mtl.p3 < rnd( seed ) );
			}

			if( mtl.d == 1.0f && !addDepth && depth == P.maxDepth + depthAdded - 1 ) {
				pathDone = true;  // pathtracing.cl:274-276: ends with no contribution
			}
			else {
				seed += hit.t;

				const f3 hitPoint = fma3( hit.t, ray.dir, ray.origin );

				// shadowRayTest, pathtracing.cl:188-199, :284-290
				bool lit = false;
				f3 lightDir = mk3( 0.0f, 0.0f, 0.0f );
				f3 lightRgb = mk3( -1.0f, -1.0f, -1.0f );

				if( SHADOW && LIGHTS ) {
					if( mtl.d > 0.0f ) {
						const float4 lpos4 = P.lights[0];
						const f3 lpos = mk3( lpos4.x, lpos4.y, lpos4.z );
						Ray lightRay;
						lightRay.origin = hitPoint;
						lightRay.dir = normalize( lpos - hitPoint );
						const f3 dl = lpos - hitPoint;
						const float tLight = sqrt1( dot( dl, dl ) );
						Hit lh;
						lh.t = tLight;
						lh.face = 0;
						unsigned unusedNodes = 0;
						traverse<true, LIGHTS, true, PHONG, EAGER>( P, lds, lightRay, lh, unusedNodes, dbgTris );
						lightDir = lightRay.dir;

						if( lh.t >= tLight ) {
							const float4 rgb = P.lights[1];
							lightRgb = mk3( rgb.x, rgb.y, rgb.z );
							lit = ( lightRgb.x >= 0.0f );
						}
					}
				}

				// getNewRay, pt_brdf.cl:344-378 — uses the UNflipped normal
				TangentFrame frame;
				frame.valid = false;
				const f3 newDir = newRayDir<BRDF, CALLS>( ray.dir, normal, mtl, seed, addDepth, &frame );

				// pathtracing.cl:298-300
				if( dot( normal, -ray.dir ) <= 0.0f ) {
					normal = -normal;
				}

				// updateColor, pathtracing.cl:89-178
				if( SHADOW && LIGHTS ) {
					if( lit ) {
						f3 add;

						if( shadowContribution<BRDF, CALLS, LATE>( mtl, ray.dir, lightDir, normal, color, lightRgb, &add, &frame ) ) {
							finalColor = finalColor + add;
							secondaryPaths += 1;
						}
					}
				}

				color = color * throughput<BRDF, CALLS, LATE>( mtl, ray.dir, newDir, normal, &frame );

				depthAdded += ( addDepth && depthAdded < P.maxAddedDepth ) ? 1 : 0;

				// russianRoulette, pt_utils.cl:385-387: rand drawn only if the first clause holds
				const float maxValColor = fmax1( color.x, fmax1( color.y, color.z ) );
				bool terminate = false;

				if( depth > 2 + depthAdded ) {
					terminate = ( maxValColor < rnd( seed ) );
				}

				if( terminate ) {
					pathDone = true;
				}
				else {
					ray.origin = hitPoint;
					ray.dir = newDir;
					depth++;
					pathDone = !( depth < P.maxDepth + depthAdded );
				}
			}
		}


	if( !pathDone ) {
		return false;
	}

	// pathtracing.cl:320-323
	if( light.x > -1.0f ) {
		finalColor = finalColor + color * light;
	}

	st.sample++;

	if( st.sample == P.samples ) {
		// pathtracing.cl:326-333 + setColors, pt_rgb.cl:9-21
		const float sp = (float) secondaryPaths;
		finalColor = mk3( div1( finalColor.x, sp ), div1( finalColor.y, sp ), div1( finalColor.z, sp ) );

		if( P.samples > 1 ) {
			const float ns = P.samplesF;
			finalColor = mk3( div1( finalColor.x, ns ), div1( finalColor.y, ns ), div1( finalColor.z, ns ) );
		}

#if PT_ARITH_NATIVE
		// Native arithmetic only: a frame whose colour is not finite contributes black.  Measured on the Cornell box with
		// BRDF 0: 3 of 3.1 M single-sample frames come out NaN with v_sin / v_cos / v_sqrt / v_rsq where the exact arithmetic
		// has none (a hardware sine that is exactly 0, a square root of a quotient an ulp out of range) — and the reference's
		// running mean (pt_rgb.cl:9-21) would carry one such sample through every later frame of that pixel.  The exact mode
		// never does this: there a NaN is the reference's NaN.
		if( !( __builtin_fabsf( finalColor.x + finalColor.y + finalColor.z ) < inff() ) ) {
			finalColor = mk3( 0.0f, 0.0f, 0.0f );
		}
#endif
		// the unit ends here; foldFrames applies the running mean (setColors) in frame order
		P.frameBuf[frameBufIndex( P, st.slot, (unsigned) st.frame )] = make_float4( finalColor.x, finalColor.y, finalColor.z, st.focus );
		cnt.nodes += st.dbgNodes;
		cnt.tris += st.dbgTris;
		st.frame++;
		return true;
	}

	color = mk3( 1.0f, 1.0f, 1.0f );
	depth = 0;
	depthAdded = 0;
	float tFocus, tObject;
	focusInputs( P, st.slot, &tFocus, &tObject );
	int px, py;
	pixelOfSlot( P, st.slot, &px, &py );
	ray = initRay( P, px, py, seed, tFocus, tObject );
	cnt.paths++;

	return false;
}

// One bounce of the lane's current path: traverse (pathtracing.cl:259), then shadeStep.
// LEAF_EAGER: the closest-hit walk requests a leaf's second face before it tests the first (EAGER also steers the shading)
template<int BRDF, bool SHADOW, bool LIGHTS, bool PHONG = false, bool EAGER = false, bool LEAF_EAGER = EAGER, bool LATE = true>
PT_DEV bool stepPixel( const DevParams& P, const float4* lds, PixelState& st, LaneCounters& cnt ) {
	Hit hit;
	hit.t = inff();
	hit.face = 0;
	hit.normal = mk3( 0.0f, 0.0f, 0.0f );
	traverse<false, LIGHTS, true, PHONG, LEAF_EAGER>( P, lds, st.ray, hit, st.dbgNodes, st.dbgTris );
	return shadeStep<BRDF, SHADOW, LIGHTS, PHONG, EAGER, !EAGER, true, LATE>( P, lds, st, cnt, hit );
}

// ---- the pixel-slot queue ----------------------------------------------------------------
// Work is handed out in pixel slots (64 per 8x8 tile).  The local tiles are partitioned into PT_BANDS
// lists — bands of tile rows; a wave prefers the band of the XCD it runs on
// (HW_REG_XCC_ID) and moves on to the other bands once its own is empty.  So the 8 XCDs — each with a
// private 4 MiB L2 — work on 8 different parts of the image instead of all on the same strip, and the rays
// in flight on one XCD (primary rays and the first bounces that start where they hit) share that L2 with 1/8
// of the scene's hot lines instead of all of them.  WHICH tile a band deals next is a table the host writes
// (DevParams.tileOrder; pbr_hip.hip, "the dealing order"): column by column inside the band — the tiles the
// waves of one XCD hold at a time form a compact block, not a 1920-pixel-wide strip — or, once the host knows
// what the tiles cost, by cost (expensive tiles first in short render calls, the expensive quarter last in
// long ones); the kernel only follows the table.
// A band has PT_SUB = 4 HEADS: head s deals tiles s, s + 4, s + 8 ... of the band's order, so the four advance through the same
// neighbourhood side by side.  A wave draws from the head of its wave index in the block (mod 4) and, once it has left its own
// band, from the same sub-head of the band it helps — the thieves of a band spread over its four heads.  A head is one address
// that every XCD's atomics must reach in memory: it hands out ≈ 90 draws / µs, and the launch draws 250 – 600 / µs.  With one head
// per band the end of a launch, when the XCDs that have run dry converge on the few bands that still hold tiles, was bound by
// that: 64-frame launches dealt spatially +6.5 % (Cornell), +2.1 % (Sponza-class), +8.6 % (Dragon-class), +0.9 % (hairball)
// with four heads; eight heads per band are slower again (profiles/r06/experiments/queue_subheads_4_8.txt).
// Frame-parallel launches deal a PIXEL THROUGH ALL ITS FRAMES before the next pixel of the tile: the 64 units a
// wave fetches together are 64 frames of one pixel — camera rays that differ only by their jitter, the same nodes,
// the same leaf, the same material — and a lane that finishes takes another frame of a pixel nearby.  Against frame
// after frame of the whole band (the lanes of a wave = the pixels of a tile): Cornell +2.7 %, Sponza-class +3 %,
// Dragon-class +3.4 %, hairball +8.4 %.  The price is the frame buffer write: 16 B per lane to 64 different frame planes.
// Placement is for speed only: every unit is handed out exactly once whichever wave asks.
#define PT_BAND_STRIDE 32   // words between queue heads: one 128-B line each
#define PT_NO_WORK 0xFFFFFFFFu

struct WorkCursor {
	unsigned exhausted;   // bit p: this lane has seen the p-th head of its wave's visiting order empty (PT_HEADS <= 32)
	int home;             // preferred head: ( band of the wave's XCD ) * PT_SUB + ( wave index in the block mod PT_SUB )
};

PT_DEV WorkCursor beginWork() {
	unsigned xcc;
	asm volatile( "s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"( xcc ) );
	WorkCursor wc;
	wc.exhausted = 0u;
	// (wave-uniform by construction; said so, or everything derived from it is vector arithmetic and the 96-register kernel needs 97)
	wc.home = (int) ( ( xcc & ( PT_BANDS - 1 ) ) * PT_SUB + ( (unsigned) __builtin_amdgcn_readfirstlane( (int) ( threadIdx.x >> 6 ) ) & ( PT_SUB - 1 ) ) );
	return wc;
}

// Next unit of work: a pixel slot (local tile * 64 + position in tile) — and, for frame-parallel
// launches (frames > 1 units per pixel), which frame of it — or PT_NO_WORK.  Per-lane control flow on
// purpose ("Toolchain notes"); the head index is made wave-uniform before the atomic so
// that hipcc still folds the adds of the active lanes into one wave-level add.
// Visiting order of a wave: band after band starting at its own, inside every band the PT_SUB heads starting at its own sub-head.
// n / d for a divisor that is fixed per launch, without dividing (Granlund & Montgomery 1994, the round-up variant:
// exact for every 32-bit n): the host derives {magic, shifts} from d (pbr_hip.hip, invariantDivisor).  A 32-bit
// division by a run-time value is ~20 instructions, and nextSlot runs whenever any lane of a wave takes a new unit.
PT_DEV unsigned divInvariant( unsigned n, unsigned magic, unsigned shifts ) {
	const unsigned t = __umulhi( magic, n );
	return ( t + ( ( n - t ) >> ( shifts & 255u ) ) ) >> ( shifts >> 8 );
}

PT_DEV unsigned nextSlot( const DevParams& P, WorkCursor& wc, unsigned frames, unsigned& frame ) {
	static_assert( PT_HEADS == 32 && PT_SUB == 4, "WorkCursor::exhausted and the published word are one 32-bit mask of eight nibbles" );

	while( wc.exhausted != (unsigned) ( ( 1ull << PT_HEADS ) - 1ull ) ) {
		// first head of this wave's visiting order that this lane has not seen empty
		const unsigned pos = (unsigned) __builtin_ctz( ~wc.exhausted );
		const unsigned homeBand = (unsigned) wc.home / PT_SUB, homeSub = (unsigned) wc.home & ( PT_SUB - 1 );
		const int mine = (int) ( ( ( homeBand + pos / PT_SUB ) & ( PT_BANDS - 1 ) ) * PT_SUB + ( ( homeSub + pos ) & ( PT_SUB - 1 ) ) );
		const int head = __builtin_amdgcn_readfirstlane( mine );
		const int band = head / PT_SUB;
		const unsigned sub = (unsigned) head & ( PT_SUB - 1 );
		const unsigned q = atomicAdd( P.workCounter + head * PT_BAND_STRIDE, 1u );

		// the head's share of the band: tiles sub, sub + PT_SUB ... of its order; frame-parallel launches: `frames` units per pixel slot
		const unsigned bandSlots = ( ( P.bandTiles[band] + ( PT_SUB - 1 ) - sub ) / PT_SUB ) * 64u;

		if( q >= bandSlots * frames ) {
			// (the head of the wave's first active lane, at ITS place in this lane's order)
			wc.exhausted |= 1u << ( ( ( (unsigned) band - homeBand ) & ( PT_BANDS - 1 ) ) * PT_SUB + ( ( sub - homeSub ) & ( PT_SUB - 1 ) ) );
			// Publish it — bit h of the word behind the heads: head h is empty — and take what the other waves have published
			// from the value the OR returns.  Without it every wave ends its launch with one failed draw per head, 32 round
			// trips to memory one after the other (~36 us: 2 - 3 % of a single frame, 1 - 4 % of a rank's 20-frame share);
			// with it two.  Into this wave's visiting order: rotate by its home band's nibbles, then every nibble by its sub-head.
			// (in vector registers on purpose: this is where the scalar registers of the 80-register kernels run out)
			const unsigned seen = atomicOr( P.workCounter + PT_HEADS * PT_BAND_STRIDE, 1u << head );
			const unsigned byBand = __funnelshift_r( seen, seen, homeBand * PT_SUB );
			const unsigned low = 0x11111111u * ( 0xFu >> homeSub );
			wc.exhausted |= ( ( byBand >> homeSub ) & low ) | ( ( byBand << ( PT_SUB - homeSub ) ) & ~low );
			continue;
		}

		// unit q of the head = frame ( q mod frames ) of its pixel slot q / frames: a pixel through all frames, then the next pixel
		const unsigned qf = ( frames > 1u ) ? divInvariant( q, P.framesDiv[0], P.framesDiv[1] ) : q;
		frame = q - qf * frames;
		// (a 32-bit byte offset from a scalar base: one global_load_dword with an SGPR pair, no 64-bit address in vector registers)
		const unsigned tile = *(const unsigned*) ( (const char*) P.tileOrder + ( ( P.bandFirst[band] + ( qf >> 6 ) * PT_SUB + sub ) << 2 ) );
		return tile * 64u + ( qf & 63u );
	}

	return PT_NO_WORK;
}

// Work distribution: nextSlot() above.  EVERY lane draws its units — one frame of one pixel each — with a plain per-lane
// atomicAdd( head, 1 ); hipcc folds the adds of the lanes that are active at that point into one wave-level add
// (v_mbcnt + s_bcnt1 + a single global_atomic_add) and hands each lane base + its rank — the ballot / prefix-sum
// refill, done by the compiler.  A lane whose unit is finished takes the next one at once while its neighbours keep
// tracing.  Lanes that fetch together get consecutive frames of one pixel (or, in a single-frame launch, the pixels of
// one tile); correctness does not depend on it.
//
// (A lane-0 atomic + readfirstlane + wave-uniform `break` formulation of this loop was
// miscompiled by ROCm 7.2 hipcc into an endless re-run of tile 0 on gfx950 — DESIGN.md,
// "Toolchain notes" — hence the deliberately per-lane control flow.)
// Block = 1024 or 768 threads: one or two blocks own a CU's 160 KB of LDS for the staged tree
// top.  MINW = waves per SIMD the register allocation must admit (__launch_bounds__):
//   4  "lean"  <= 128 VGPRs, 1 block / CU — no spills; best when the kernel is bound by its own
//              arithmetic and for the lane state machine on the largest scenes;
//   6  "mid"   <= 80 VGPRs, two 768-thread blocks / CU;
//   8  "wide"  <= 64 VGPRs, 2 blocks / CU — the walk stays spill-free, the shading code spills to
//              scratch; twice the waves to hide the latency of dependent node fetches.
#ifndef PBR_BLOCK
#define PBR_BLOCK 1024
#endif

extern __shared__ float4 gHotNodes[];

enum { MODE_NODE = 0, MODE_LEAF = 1, MODE_SHADE = 2, MODE_DONE = 3 };

struct WalkState {
	f3 invDir;
	Cursor cur;
	Hit hit;
	int leafFace0, leafFace1;
	float leafTNear;
};

// Every block stages the hot nodes once (32 B x numHot, coalesced) before its waves start.
PT_DEV void stageHotNodes( const DevParams& P, float4* lds ) {
	// nodePhaseAsm reads the staged records at LDS address = record reference: the staged prefix must start at LDS address 0
	// (it does: these kernels have no static __shared__ data).  Should a toolchain ever lay it out otherwise, say so loudly.
	if( (unsigned) (size_t) lds != 0u && threadIdx.x == 0 && P.guard != nullptr ) {
		P.guard[3] = 1u;
	}

	for( int i = (int) threadIdx.x; i < P.numHot * 2; i += (int) blockDim.x ) {
		lds[i] = P.nodes[i];
	}

	__syncthreads();
}

// ---------------------------------------------------------------------------------------
// Lock-step schedule ("refill"): all lanes walk, then all shade
// ---------------------------------------------------------------------------------------
template<int BRDF, bool SHADOW, bool LIGHTS, int MINW, bool PHONG = false>
__global__ __launch_bounds__( PBR_BLOCK, MINW ) void pathTracing( const DevParams P ) {
	const float4* lds = gHotNodes;
	PT_LAB_WAVE_BEGIN
	stageHotNodes( P, gHotNodes );

	const unsigned total = (unsigned) P.numLocalTiles * 64u;   // bound of the PBR_GUARD loop limits
	(void) total;
	LaneCounters cnt;
	cnt.nodes = cnt.tris = cnt.hits = cnt.paths = 0;
	PixelState st;

	WorkCursor work = beginWork();
	unsigned frame = 0;
	unsigned slot = nextSlot( P, work, (unsigned) P.nFrames, frame );
	bool have = ( slot != PT_NO_WORK );
#ifdef PBR_GUARD_PATH
	long long guardSteps = 0;
	const long long guardMax = ( (long long) P.nFrames * P.samples * ( P.maxDepth + P.maxAddedDepth + 1 ) + 1 ) * ( (long long) total + 1 );
#endif

	if( have ) {
		beginPixel( P, st, slot, cnt, frame );
	}

	// Refill in batches (round 3).  A lane whose unit is finished does not take the next one at once: it waits until
	// P.refillBatch lanes of the wave wait with it (or nobody has a unit left), and they take their units — queue fetch,
	// pixel decode, camera ray: ~450 instructions — together.  In a lock-step wave an idle lane costs nothing (the wave
	// issues the same walk and the same shading for 40 lanes as for 64), but code that runs for three lanes costs what
	// it costs for 64: with immediate refill nearly every bounce of a wave ended with a refill for the few lanes whose
	// path had just ended.
	bool pending = false;

	while( __ballot( have || pending ) != 0ull ) {
#ifdef PBR_GUARD_PATH
		if( ++guardSteps > guardMax ) {
			atomicAdd( &P.guard[1], 1u );
			break;
		}
#endif
		{
			const int nPending = __popcll( __ballot( pending ) );
			const int nHave = __popcll( __ballot( have ) );

			if( pending && ( nPending >= P.refillBatch || nHave == 0 ) ) {
				slot = nextSlot( P, work, (unsigned) P.nFrames, frame );
				have = ( slot != PT_NO_WORK );
				pending = false;

				if( have ) {
					beginPixel( P, st, slot, cnt, frame );
				}
			}
		}

		if( have && stepPixel<BRDF, SHADOW, LIGHTS, PHONG, ( MINW <= 4 ), ( MINW <= PT_EAGER_REFILL_UP_TO ), ( MINW > 6 )>( P, lds, st, cnt ) ) {
			finishPixel( P, st );

			if( cnt.nodes > 0x40000000u || cnt.tris > 0x40000000u ) {
				flushCounters( P, cnt );
			}

			have = false;
			pending = true;
		}
	}

	flushCounters( P, cnt );
	PT_LAB_WAVE_END_LOCKSTEP( P )
}


// ---------------------------------------------------------------------------------------
// Phased schedule: a lane state machine, run as nested phase loops
// ---------------------------------------------------------------------------------------
// The lock-step schedule keeps the 64 lanes of a wave in step per bounce: every traversal lasts
// as long as the wave's longest ray, and a leaf's triangle tests run while the lanes that stand
// on container nodes idle (measured on the Sponza-class scene: ~28 % of the issued lane slots do
// useful work).  Here every lane is a small state machine
//
//   NODE   fetch the node, slab test, follow the hit / miss link (pt_bvh.cl:88-117)
//   LEAF   the lane stands on a hit leaf; its (long) triangle tests are deferred ...
//   SHADE  the lane's ray has left the tree; its (very long) shading step is deferred ...
//   DONE   no unit of work left
//
// and the wave alternates between PHASES:
//
//   node phase   a tight loop over the lanes that are walking; a lane leaves it when it hits a
//                leaf (-> LEAF) or its ray has left the tree (-> SHADE); the loop itself ends once
//                P.phPark lanes have left it (or nobody is left)
//   leaf phase   the triangle tests of every lane parked on a leaf, in one go
//   shade phase  once P.phShade lanes wait for shading (or nothing else can run): shade them,
//                start their next rays / take their next units
//
// The node loop carries only the walk state; the path state is untouched between shade phases.  Per lane the sequence
// of node visits, face tests and random draws is exactly the reference's, so the image stays bit-identical; only the
// interleaving across lanes changes.
template<bool LIGHTS>
PT_DEV int startWalk( const DevParams& P, const Ray& ray, WalkState& w ) {
	w.invDir = mk3( div1( 1.0f, ray.dir.x ), div1( 1.0f, ray.dir.y ), div1( 1.0f, ray.dir.z ) );
	w.cur = firstNode( P, ray.dir );
	w.hit.t = inff();
	w.hit.face = 0;
	w.leafFace0 = -1;
	w.leafFace1 = -1;
	w.leafTNear = 0.0f;

	if( LIGHTS ) {
		traverseLights( P, ray, w.hit );
	}

	return MODE_NODE;
}

template<int BRDF, bool SHADOW, bool LIGHTS, int MINW>
__global__ __launch_bounds__( PBR_BLOCK, MINW ) void pathTracingPhased( const DevParams P ) {
	const float4* lds = gHotNodes;
	PT_LAB_WAVE_BEGIN
	stageHotNodes( P, gHotNodes );

	const unsigned total = (unsigned) P.numLocalTiles * 64u;   // bound of the PBR_GUARD loop limits
	(void) total;
	const int numNodes = P.numNodes;
	(void) numNodes;
	LaneCounters cnt;
	cnt.nodes = cnt.tris = cnt.hits = cnt.paths = 0;
	PixelState st;
	WalkState w;
	int mode = MODE_DONE;
	PT_LAB_PHASED_BEGIN

	WorkCursor work = beginWork();
	unsigned frame = 0;

	{
		const unsigned slot = nextSlot( P, work, (unsigned) P.nFrames, frame );

		if( slot != PT_NO_WORK ) {
			beginPixel( P, st, slot, cnt, frame );
			mode = startWalk<LIGHTS>( P, st.ray, w );
		}
	}

#ifdef PBR_GUARD_PATH
	long long guardSteps = 0;
	const long long guardMax = ( (long long) P.nFrames * P.samples * ( P.maxDepth + P.maxAddedDepth + 1 ) + 1 ) * ( (long long) numNodes + 4 ) * ( (long long) total + 1 );
#endif

	while( __ballot( mode != MODE_DONE ) != 0ull ) {
#ifdef PBR_GUARD_PATH
		if( ++guardSteps > guardMax ) {
			atomicAdd( &P.guard[1], 1u );
			break;
		}
#endif
		// Once the queue has run dry the wave shrinks: lanes whose last unit is finished are DONE for good.  The two
		// thresholds are meant as shares of the wave (16 and 40 of 64 lanes); held at their absolute values, a wave of 30
		// surviving lanes ends a node phase only when 16 of them have left it.  P.drainMode chooses which of the two scale
		// with the lanes still at work (bit 0: the park count, bit 1: the shade threshold).
		// How a launch ends, measured in round 3 (lab hook -DPBR_EXP_TIMELINE, profiles/r03/experiments/timeline.txt;
		// Sponza-class scene, one 1080p frame): the queue is empty after ~1.0 ms with all 393 k lanes holding a path; 250 us
		// later half of them are done, after another 250 us 93 % — and the launch runs ~0.7 ms more with 2 - 7 % of its lanes,
		// one to four paths per wave: the paths with eight bounces and long walks, each bound by the latency of its own
		// dependent node fetches (a record's successors are in the record).  Three ways of helping those stragglers were
		// built, tested bit-identical and measured no faster (lab/src/pt_drain.hpp, DESIGN.md "How a launch ends"): a ring
		// in LDS that repacks them into fewer waves, a pool in global memory with a second kernel, and the wave's own
		// idle lanes fetching and testing the adjacent records of the stream ahead of the walk.
		const int lanesAtWork = __popcll( __ballot( mode != MODE_DONE ) );
		const int parkScaled = ( ( P.phPark * lanesAtWork ) >> 6 ) < 1 ? 1 : ( ( P.phPark * lanesAtWork ) >> 6 );
		const int shadeScaled = ( ( P.phShade * lanesAtWork ) >> 6 ) < 1 ? 1 : ( ( P.phShade * lanesAtWork ) >> 6 );
		const int parkNow = ( lanesAtWork >= 64 || !( P.drainMode & 1 ) ) ? P.phPark : parkScaled;
		const int shadeNow = ( lanesAtWork >= 64 || !( P.drainMode & 2 ) ) ? P.phShade : shadeScaled;

		PT_LAB_PHASED_NODE_BEGIN( mode )
		// ---- node phase ---------------------------------------------------------------------
		if( mode == MODE_NODE ) {
			const int keep = __popcll( __ballot( 1 ) ) - parkNow;
			unsigned visits = 0;
			PT_LAB_PHASED_STAT( 3 )

#ifdef PT_NODE_PHASE_ASM
			{
				// the same hand-scheduled node phase as traverse()
				const f2v oxy = { st.ray.origin.x, st.ray.origin.y };
				const f2v ozz = { st.ray.origin.z, st.ray.origin.z };
				const f2v ixy = { w.invDir.x, w.invDir.y };
				const f2v izz = { w.invDir.z, w.invDir.z };
				int leafWord = 0, parkedFlag;
				float unusedTFar;
				__builtin_amdgcn_s_setprio( PT_WALK_PRIO );   // through the leaf phase below
#if PT_WALK_COMPACT == 1
				// (the ray's order, from its direction, per node phase: five instructions against a register carried through the whole loop)
				nodePhaseAsmCompact<false>( P, oxy, ozz, ixy, izz, w.hit.t, ( keep < 0 ) ? 0 : keep, walkCompactOffset( st.ray.dir ), w.cur.ref, visits, leafWord, w.leafTNear, unusedTFar, parkedFlag );
#else
				nodePhaseAsm<false>( P, oxy, ozz, ixy, izz, w.hit.t, ( keep < 0 ) ? 0 : keep, w.cur.ref, visits, leafWord, w.leafTNear, unusedTFar, parkedFlag );
#endif
				st.dbgNodes += visits;
				PT_LAB_PHASED_NODE_MID

				// ---- leaf phase: only lanes that have just come out of the node phase can stand on a leaf
				if( parkedFlag != 0 ) {
					testLeaf<false, ( MINW <= PT_EAGER_UP_TO )>( P, leafFace0( leafWord ), leafFace1( leafWord ), st.ray, w.leafTNear, 0.0f, w.hit, st.dbgTris );
				}
				__builtin_amdgcn_s_setprio( 0 );
				PT_LAB_PHASED_LEAF_END

				if( !alive( w.cur ) ) {
					mode = MODE_SHADE;
				}
			}
#else
			do {
				visits++;
				PT_LAB_PHASED_STAT( 0 )

				float4 n0, n1;
#if PT_WALK_COMPACT == 1
				int nextWord;
				const int kOff = walkCompactOffset( st.ray.dir );
				fetchNodeCompact<true>( P, lds, w.cur, kOff, &n0, &n1, &nextWord );
				const NodeLinks node = decodeNodeCompact( n1, nextWord, kOff );
#else
				fetchNode<true>( P, lds, w.cur, &n0, &n1 );
				const NodeLinks node = decodeNode( n1 );
#endif
				float tNear;

				if( boxHit<false>( n0, n1, st.ray, w.invDir, w.hit.t, &tNear ) ) {
					w.cur = node.onHit;

					if( node.leaf ) {
						w.leafFace0 = node.face0;
						w.leafFace1 = node.face1;
						w.leafTNear = tNear;
						mode = MODE_LEAF;
					}
				}
				else {
					w.cur = node.onMiss;
				}

				if( mode == MODE_NODE && !alive( w.cur ) ) {
					mode = MODE_SHADE;
				}
			} while( mode == MODE_NODE && __popcll( __ballot( mode == MODE_NODE ) ) > keep );

			st.dbgNodes += visits;
#endif
		}

		PT_LAB_PHASED_NODE_END
#ifndef PT_NODE_PHASE_ASM
		// ---- leaf phase ---------------------------------------------------------------------
		if( mode == MODE_LEAF ) {
			PT_LAB_PHASED_STAT( 1 )
			testLeaf<false, ( MINW <= 4 )>( P, w.leafFace0, w.leafFace1, st.ray, w.leafTNear, 0.0f, w.hit, st.dbgTris );
			mode = alive( w.cur ) ? MODE_NODE : MODE_SHADE;
		}
#endif

		// ---- shade phase --------------------------------------------------------------------
		// (Measured and rejected: batching the two halves of shadeStep separately — shadeSurface for the lanes whose ray
		// hit a face, endPath, with its camera ray, for the lanes whose path has ended, each behind its own threshold.
		// Shading got cheaper per pass, but the lanes of a third waiting state are missing from the node phases:
		// Sponza-class 0.91x, Dragon-class 0.90x, hairball 0.95x at the best thresholds; Cornell-class 1.09x, still
		// behind its lock-step kernel.)
		{
			const int nShade = __popcll( __ballot( mode == MODE_SHADE ) );
			const int nNode = __popcll( __ballot( mode == MODE_NODE ) );

			PT_LAB_PHASED_SHADE_BEGIN
			if( mode == MODE_SHADE && ( nShade >= shadeNow || nNode == 0 ) ) {
				PT_LAB_PHASED_STAT( 2 )
				// (stored face normals, FACEN, re-measured in round 3 with the registers the late colours freed: the 6-waves
				// kernel spills again, Sponza-class -2.7 %, Dragon-class -1.9 % — profiles/r03/experiments/facen_state_machine.txt)
				if( shadeStep<BRDF, SHADOW, LIGHTS, false, ( MINW <= 4 ), true>( P, lds, st, cnt, w.hit ) ) {
					finishPixel( P, st );

					if( cnt.nodes > 0x40000000u || cnt.tris > 0x40000000u ) {
						flushCounters( P, cnt );
					}

					const unsigned slot = nextSlot( P, work, (unsigned) P.nFrames, frame );

					if( slot != PT_NO_WORK ) {
						beginPixel( P, st, slot, cnt, frame );
						mode = startWalk<LIGHTS>( P, st.ray, w );
					}
					else {
						mode = MODE_DONE;
						PT_LAB_WAVE_DRY
					}
				}
				else {
					mode = startWalk<LIGHTS>( P, st.ray, w );
				}
			}
			PT_LAB_PHASED_SHADE_END
		}
	}

	flushCounters( P, cnt );
	PT_LAB_PHASED_END( P )
	PT_LAB_WAVE_END_PHASED( P )
}


#if PT_WALK_COMPACT != 1
#include "pt_dual.hpp"               // pathTracingDual: the lane state machine with two paths per lane (plan 6, "phased-dual")
#endif                               // (no two-walk node phase over compact records: those flavours render plan 6 with the 6-waves state machine)

}  // namespace ptk
