// Device-side BVH build (second half): an opt-in, fast builder that emits the
// reference's flat format — nodes in depth-first order, `bbMin.w` = first face or -1, `bbMax.w` =
// second face / miss link / -1 (PathTracer.cpp:238-310) with at most 2 faces per leaf (the
// reference's bvh.max_faces) and the faces re-ordered into leaf order — so that pbr_upload_scene and
// the kernels take it like the host builder's output.
//
// It is NOT the reference's builder (accelstructures/BVH.cpp: per-object trees, SAH sweeps over
// std::sort-ed faces, mean splits above 100 k faces, skip-ahead deletion): that one is replicated
// on the host (host/bvh_builder.cpp), takes seconds for 2 M triangles and defines the trees the parity
// tests and the benchmark use.  This one sorts the faces along a Morton curve and clusters them
// bottom-up by surface area (PlocArrays below; round 1's binary radix tree over the same order is kept
// as PBR_BVH_BUILDER=lbvh): milliseconds for the same input, for callers who want a scene on the
// device now.  Images rendered over it agree with images over the reference's tree statistically,
// not bit for bit (the walk order decides ties and the triangle test starts from the leaf box's
// tNear, pt_intersect.cl:96-120).  Its result is deterministic: boxes are exact min / max, the keys
// are unique, ties are ordered.
#pragma once

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "pbr_hip.h"

namespace ptb {

struct BuildArrays {
	const pbr_float4* vertices;
	const pbr_uint4* facesV;
	const pbr_uint4* facesN;
	unsigned numFaces, numLeaves;       // leaves hold faces 2l and 2l + 1 of the sorted order
	unsigned long long* keys;           // Morton code << 24 | face index (unique)
	unsigned long long* keysSorted;
	float* sceneMin;                    // 3 floats each, centroid bounds
	float* sceneMax;
	// binary radix tree over the leaves: node ids [0, numLeaves - 1) are internal, then the leaves
	int* left;
	int* right;
	int* parent;
	unsigned* size;                     // nodes in the subtree, the node itself included
	unsigned* arrived;                  // bottom-up pass: second arrival computes the box
	float4* boxMin;
	float4* boxMax;
	pbr_bvh_node* nodesOut;
	pbr_uint4* facesVOut;
	pbr_uint4* facesNOut;
};

__device__ __forceinline__ void atomicMinFloat( float* address, float value ) {
	// valid for any mix of signs: order-preserving map to unsigned
	unsigned bits = __float_as_uint( value );
	bits = ( bits & 0x80000000u ) ? ~bits : ( bits | 0x80000000u );
	atomicMin( (unsigned*) address, bits );
}

__device__ __forceinline__ void atomicMaxFloat( float* address, float value ) {
	unsigned bits = __float_as_uint( value );
	bits = ( bits & 0x80000000u ) ? ~bits : ( bits | 0x80000000u );
	atomicMax( (unsigned*) address, bits );
}

__device__ __forceinline__ float orderedToFloat( unsigned bits ) {
	bits = ( bits & 0x80000000u ) ? ( bits & 0x7FFFFFFFu ) : ~bits;
	return __uint_as_float( bits );
}

__device__ __forceinline__ void faceCorners( const BuildArrays& B, unsigned face, float3* a, float3* b, float3* c ) {
	const pbr_uint4 f = B.facesV[face];
	const pbr_float4 va = B.vertices[f.x], vb = B.vertices[f.y], vc = B.vertices[f.z];
	*a = make_float3( va.x, va.y, va.z );
	*b = make_float3( vb.x, vb.y, vb.z );
	*c = make_float3( vc.x, vc.y, vc.z );
}

__device__ __forceinline__ float3 faceCentroid( float3 a, float3 b, float3 c ) {
	// centre of the face's box: cheap, and what the split quality depends on
	return make_float3(
		0.5f * ( fminf( a.x, fminf( b.x, c.x ) ) + fmaxf( a.x, fmaxf( b.x, c.x ) ) ),
		0.5f * ( fminf( a.y, fminf( b.y, c.y ) ) + fmaxf( a.y, fmaxf( b.y, c.y ) ) ),
		0.5f * ( fminf( a.z, fminf( b.z, c.z ) ) + fmaxf( a.z, fmaxf( b.z, c.z ) ) ) );
}

// sceneMin / sceneMax hold order-preserving unsigned images of floats (initialised to +inf / -inf images)
__global__ void centroidBounds( const BuildArrays B ) {
	const unsigned face = blockIdx.x * blockDim.x + threadIdx.x;

	if( face >= B.numFaces ) {
		return;
	}

	float3 a, b, c;
	faceCorners( B, face, &a, &b, &c );
	const float3 m = faceCentroid( a, b, c );
	atomicMinFloat( &B.sceneMin[0], m.x );
	atomicMinFloat( &B.sceneMin[1], m.y );
	atomicMinFloat( &B.sceneMin[2], m.z );
	atomicMaxFloat( &B.sceneMax[0], m.x );
	atomicMaxFloat( &B.sceneMax[1], m.y );
	atomicMaxFloat( &B.sceneMax[2], m.z );
}

// spread the low 13 bits of v to every third bit of a 39-bit word
__device__ __forceinline__ unsigned long long expandBits13( unsigned v ) {
	unsigned long long x = (unsigned long long) ( v & 0x1FFFu );
	x = ( x | ( x << 32 ) ) & 0x001F00000000FFFFull;
	x = ( x | ( x << 16 ) ) & 0x001F0000FF0000FFull;
	x = ( x | ( x << 8 ) ) & 0x100F00F00F00F00Full;
	x = ( x | ( x << 4 ) ) & 0x10C30C30C30C30C3ull;
	x = ( x | ( x << 2 ) ) & 0x1249249249249249ull;
	return x;
}

__global__ void mortonKeys( const BuildArrays B ) {
	const unsigned face = blockIdx.x * blockDim.x + threadIdx.x;

	if( face >= B.numFaces ) {
		return;
	}

	float3 a, b, c;
	faceCorners( B, face, &a, &b, &c );
	const float3 m = faceCentroid( a, b, c );
	const float lo[3] = { orderedToFloat( __float_as_uint( B.sceneMin[0] ) ), orderedToFloat( __float_as_uint( B.sceneMin[1] ) ), orderedToFloat( __float_as_uint( B.sceneMin[2] ) ) };
	const float hi[3] = { orderedToFloat( __float_as_uint( B.sceneMax[0] ) ), orderedToFloat( __float_as_uint( B.sceneMax[1] ) ), orderedToFloat( __float_as_uint( B.sceneMax[2] ) ) };
	const float p[3] = { m.x, m.y, m.z };
	unsigned q[3];

	for( int k = 0; k < 3; k++ ) {
		const float extent = hi[k] - lo[k];
		const float u = ( extent > 0.0f ) ? ( p[k] - lo[k] ) / extent : 0.0f;
		q[k] = (unsigned) fminf( fmaxf( u * 8192.0f, 0.0f ), 8191.0f );
	}

	// 39-bit Morton code (13 bits per axis: 8192 cells — with 10 bits, a 2 M-triangle mesh puts dozens of faces into one
	// cell and their grouping is then the order of their indices) above the 24-bit face index (pbr_upload_scene's limit)
	const unsigned long long code = ( expandBits13( q[0] ) << 2 ) | ( expandBits13( q[1] ) << 1 ) | expandBits13( q[2] );
	B.keys[face] = ( code << 24 ) | (unsigned long long) face;
}

// length of the common prefix of the keys of leaves i and j, -1 outside the array (Karras 2012, delta)
__device__ __forceinline__ int commonPrefix( const BuildArrays& B, int i, int j ) {
	if( j < 0 || j >= (int) B.numLeaves ) {
		return -1;
	}

	const unsigned long long a = B.keysSorted[(size_t) i * 2];
	const unsigned long long b = B.keysSorted[(size_t) j * 2];
	return __clzll( (long long) ( a ^ b ) );   // the keys are unique: a != b
}

// one thread per internal node: its key range and split (Karras 2012, figure 4)
__global__ void radixTree( const BuildArrays B ) {
	const int i = (int) ( blockIdx.x * blockDim.x + threadIdx.x );
	const int internals = (int) B.numLeaves - 1;

	if( i >= internals ) {
		return;
	}

	const int d = ( commonPrefix( B, i, i + 1 ) - commonPrefix( B, i, i - 1 ) ) >= 0 ? 1 : -1;
	const int deltaMin = commonPrefix( B, i, i - d );
	int lMax = 2;

	while( commonPrefix( B, i, i + lMax * d ) > deltaMin ) {
		lMax *= 2;
	}

	int l = 0;

	for( int t = lMax / 2; t >= 1; t /= 2 ) {
		if( commonPrefix( B, i, i + ( l + t ) * d ) > deltaMin ) {
			l += t;
		}
	}

	const int j = i + l * d;
	const int deltaNode = commonPrefix( B, i, j );
	int s = 0;

	for( int t = ( l + 1 ) / 2; ; t = ( t + 1 ) / 2 ) {
		if( commonPrefix( B, i, i + ( s + t ) * d ) > deltaNode ) {
			s += t;
		}
		if( t == 1 ) {
			break;
		}
	}

	const int split = i + s * d + ( ( d < 0 ) ? -1 : 0 );
	const int first = ( i < j ) ? i : j;
	const int last = ( i < j ) ? j : i;
	const int leftId = ( split == first ) ? internals + split : split;
	const int rightId = ( split + 1 == last ) ? internals + split + 1 : split + 1;
	B.left[i] = leftId;
	B.right[i] = rightId;
	B.parent[leftId] = i;
	B.parent[rightId] = i;

	if( i == 0 ) {
		B.parent[0] = -1;
	}
}

// one thread per leaf: its box, then up the tree; the second child to arrive at a node computes it
__global__ void boxesBottomUp( const BuildArrays B ) {
	const unsigned leaf = blockIdx.x * blockDim.x + threadIdx.x;

	if( leaf >= B.numLeaves ) {
		return;
	}

	const int internals = (int) B.numLeaves - 1;
	float3 lo = make_float3( __builtin_inff(), __builtin_inff(), __builtin_inff() );
	float3 hi = make_float3( -__builtin_inff(), -__builtin_inff(), -__builtin_inff() );

	for( unsigned k = 0; k < 2; k++ ) {
		const unsigned sorted = leaf * 2 + k;

		if( sorted < B.numFaces ) {
			const unsigned face = (unsigned) ( B.keysSorted[sorted] & 0xFFFFFFull );
			float3 a, b, c;
			faceCorners( B, face, &a, &b, &c );
			lo.x = fminf( lo.x, fminf( a.x, fminf( b.x, c.x ) ) );
			lo.y = fminf( lo.y, fminf( a.y, fminf( b.y, c.y ) ) );
			lo.z = fminf( lo.z, fminf( a.z, fminf( b.z, c.z ) ) );
			hi.x = fmaxf( hi.x, fmaxf( a.x, fmaxf( b.x, c.x ) ) );
			hi.y = fmaxf( hi.y, fmaxf( a.y, fmaxf( b.y, c.y ) ) );
			hi.z = fmaxf( hi.z, fmaxf( a.z, fmaxf( b.z, c.z ) ) );
			B.facesVOut[sorted] = B.facesV[face];
			B.facesNOut[sorted] = B.facesN[face];
		}
	}

	int node = internals + (int) leaf;
	B.boxMin[node] = make_float4( lo.x, lo.y, lo.z, 0.0f );
	B.boxMax[node] = make_float4( hi.x, hi.y, hi.z, 0.0f );
	B.size[node] = 1u;

	int up = B.parent[node];

	while( up >= 0 ) {
		__threadfence();   // this child's box and size are visible before the arrival is counted

		if( atomicAdd( &B.arrived[up], 1u ) == 0u ) {
			return;        // the sibling is not there yet; its thread will do this node
		}

		// agent-scope acquire: this CU's L1 is invalidated, the sibling's box (written by another CU, released
		// by its fence + atomic above) is read from L2
		__threadfence();
		const int l = B.left[up], r = B.right[up];
		const float4 lMin = B.boxMin[l], rMin = B.boxMin[r];
		const float4 lMax = B.boxMax[l], rMax = B.boxMax[r];
		B.boxMin[up] = make_float4( fminf( lMin.x, rMin.x ), fminf( lMin.y, rMin.y ), fminf( lMin.z, rMin.z ), 0.0f );
		B.boxMax[up] = make_float4( fmaxf( lMax.x, rMax.x ), fmaxf( lMax.y, rMax.y ), fmaxf( lMax.z, rMax.z ), 0.0f );
		B.size[up] = 1u + B.size[l] + B.size[r];

		// the child with the bigger surface area first, as BVH::combineNodes orders them (BVH.cpp:335-343): the walk visits
		// children in array order, and a hit found early culls what follows (pt_bvh.cl:107)
		{
			const float lx = lMax.x - lMin.x, ly = lMax.y - lMin.y, lz = lMax.z - lMin.z;
			const float rx = rMax.x - rMin.x, ry = rMax.y - rMin.y, rz = rMax.z - rMin.z;

			if( rx * ry + rz * ry + rx * rz > lx * ly + lz * ly + lx * lz ) {
				B.left[up] = r;
				B.right[up] = l;
			}
		}

		node = up;
		up = B.parent[node];
	}
}

// one thread per tree node: its position in depth-first order (left subtree first), then the record
// in the reference's wire format
__global__ void flatten( const BuildArrays B ) {
	const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned internals = B.numLeaves - 1u;
	const unsigned total = internals + B.numLeaves;

	if( id >= total ) {
		return;
	}

	unsigned index = 0;

	for( int node = (int) id, up = B.parent[id]; up >= 0; node = up, up = B.parent[up] ) {
		index += 1u + ( ( B.right[up] == node ) ? B.size[B.left[up]] : 0u );
	}

	const float4 lo = B.boxMin[id], hi = B.boxMax[id];
	pbr_bvh_node out;
	out.bbMin.x = lo.x; out.bbMin.y = lo.y; out.bbMin.z = lo.z;
	out.bbMax.x = hi.x; out.bbMax.y = hi.y; out.bbMax.z = hi.z;

	if( id >= internals ) {
		const unsigned first = ( id - internals ) * 2u;
		out.bbMin.w = (float) first;
		out.bbMax.w = ( first + 1u < B.numFaces ) ? (float) ( first + 1u ) : -1.0f;
	}
	else {
		// on a miss the walk continues behind this subtree; -1 ends it (pt_bvh.cl:102,122)
		const unsigned next = index + B.size[id];
		out.bbMin.w = -1.0f;
		out.bbMax.w = ( next < total ) ? (float) next : -1.0f;
	}

	B.nodesOut[index] = out;
}

// ---------------------------------------------------------------------------------------------------------------------
// Locally-ordered agglomerative clustering over the Morton order (Meister & Bittner 2018, "PLOC"): the default builder.
// Clusters start as single faces in Morton order; every round each cluster looks PLOC_RADIUS places to either side for
// the partner with which it forms the box of the smallest surface area, mutual choices merge, the survivors are
// compacted (prefix sum, order kept) and the round repeats until one cluster is left.  Two single faces merge into a
// 2-face leaf (the reference's bvh.max_faces = 2, config.json:40), anything else into a container.  The surface-area
// criterion sees the faces' real boxes, so a wall-sized triangle among small ones does not drag a spatial-median split
// across them the way the radix tree's centroid-only splits do.  Deterministic: distances are symmetric in their
// arguments, ties go to the lower position, node ids come from prefix sums.
//
// Node ids: [0, numFaces) = the faces in Morton order, [numFaces, 2 numFaces - 1) = merged clusters in creation order.
struct PlocArrays {
	const pbr_float4* vertices;
	const pbr_uint4* facesV;
	const pbr_uint4* facesN;
	unsigned numFaces;
	const unsigned long long* keysSorted;
	int* left;                          // bigger surface area first (BVH.cpp:335-343); -1 for faces
	int* right;
	int* parent;                        // -1 while the node is still a cluster
	unsigned* size;                     // records of the subtree in the flat format (a 2-face leaf is one record)
	unsigned* faces;                    // faces in the subtree
	float4* boxMin;
	float4* boxMax;
	int* clusters;                      // this round's clusters, in Morton order
	int* clustersNext;
	int* nearest;                       // position of the chosen partner
	unsigned long long* flags;          // survives this round | merges this round << 32
	unsigned long long* scan;           // exclusive prefix sums of flags
	unsigned long long* totals;         // [0] = inclusive total of the round (pinned host memory is fine too)
	pbr_bvh_node* nodesOut;
	pbr_uint4* facesVOut;
	pbr_uint4* facesNOut;
};

#define PLOC_MAX_RADIUS 64
#define PLOC_THREADS 256

__device__ __forceinline__ float halfArea( float3 lo, float3 hi ) {
	const float x = hi.x - lo.x, y = hi.y - lo.y, z = hi.z - lo.z;
	return x * y + z * y + x * z;
}

__device__ __forceinline__ unsigned pairKey( unsigned lo, unsigned hi ) {
	unsigned h = lo * 0x9E3779B1u + hi * 0x85EBCA77u;
	h ^= h >> 15;
	h *= 0x2C1B3C6Du;
	h ^= h >> 12;
	return h;
}

// is the pair (i, a) before the pair (i, b) in (lower end, upper end) order?
__device__ __forceinline__ bool pairBefore( int i, int a, int b ) {
	const int aLo = ( a < i ) ? a : i, aHi = ( a < i ) ? i : a;
	const int bLo = ( b < i ) ? b : i, bHi = ( b < i ) ? i : b;
	return aLo < bLo || ( aLo == bLo && aHi < bHi );
}

__global__ void plocInit( const PlocArrays B ) {
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= B.numFaces ) {
		return;
	}

	const unsigned face = (unsigned) ( B.keysSorted[i] & 0xFFFFFFull );
	const pbr_uint4 f = B.facesV[face];
	const pbr_float4 a = B.vertices[f.x], b = B.vertices[f.y], c = B.vertices[f.z];
	B.boxMin[i] = make_float4( fminf( a.x, fminf( b.x, c.x ) ), fminf( a.y, fminf( b.y, c.y ) ), fminf( a.z, fminf( b.z, c.z ) ), 0.0f );
	B.boxMax[i] = make_float4( fmaxf( a.x, fmaxf( b.x, c.x ) ), fmaxf( a.y, fmaxf( b.y, c.y ) ), fmaxf( a.z, fmaxf( b.z, c.z ) ), 0.0f );
	B.left[i] = -1;
	B.right[i] = -1;
	B.parent[i] = -1;
	B.size[i] = 1u;
	B.faces[i] = 1u;
	B.clusters[i] = (int) i;
}

// one thread per cluster: the partner within `radius` places whose union with it has the smallest surface area
__global__ void __launch_bounds__( PLOC_THREADS ) plocNearest( const PlocArrays B, const unsigned count, const int radius ) {
	__shared__ float sLo[PLOC_THREADS + 2 * PLOC_MAX_RADIUS][3];
	__shared__ float sHi[PLOC_THREADS + 2 * PLOC_MAX_RADIUS][3];
	const int base = (int) ( blockIdx.x * PLOC_THREADS ) - radius;

	for( int k = (int) threadIdx.x; k < PLOC_THREADS + 2 * radius; k += PLOC_THREADS ) {
		const int pos = base + k;

		if( pos >= 0 && pos < (int) count ) {
			const int id = B.clusters[pos];
			const float4 lo = B.boxMin[id], hi = B.boxMax[id];
			sLo[k][0] = lo.x; sLo[k][1] = lo.y; sLo[k][2] = lo.z;
			sHi[k][0] = hi.x; sHi[k][1] = hi.y; sHi[k][2] = hi.z;
		}
	}

	__syncthreads();
	const int i = (int) ( blockIdx.x * PLOC_THREADS + threadIdx.x );

	if( i >= (int) count ) {
		return;
	}

	const int me = (int) threadIdx.x + radius;
	const float3 lo = make_float3( sLo[me][0], sLo[me][1], sLo[me][2] );
	const float3 hi = make_float3( sHi[me][0], sHi[me][1], sHi[me][2] );
	float best = __builtin_inff();
	unsigned bestKey = 0;
	int bestPos = -1;

	for( int d = -radius; d <= radius; d++ ) {
		const int pos = i + d;

		if( d == 0 || pos < 0 || pos >= (int) count ) {
			continue;
		}

		const int k = me + d;
		const float3 ulo = make_float3( fminf( lo.x, sLo[k][0] ), fminf( lo.y, sLo[k][1] ), fminf( lo.z, sLo[k][2] ) );
		const float3 uhi = make_float3( fmaxf( hi.x, sHi[k][0] ), fmaxf( hi.y, sHi[k][1] ), fmaxf( hi.z, sHi[k][2] ) );
		const float area = halfArea( ulo, uhi );

		// equal areas (regular meshes are full of them) are ordered by a hash of the PAIR, then by the pair itself: a
		// total order over pairs that both ends evaluate alike, so the smallest pair of all is always mutual (progress
		// every round) and ties pair up at random places instead of every cluster pointing at its lower neighbour
		const unsigned key = pairKey( (unsigned) ( ( pos < i ) ? pos : i ), (unsigned) ( ( pos < i ) ? i : pos ) );

		if( bestPos < 0 || area < best || ( area == best && ( key < bestKey || ( key == bestKey && pairBefore( i, pos, bestPos ) ) ) ) ) {
			best = area;
			bestKey = key;
			bestPos = pos;
		}
	}

	B.nearest[i] = bestPos;
}

__global__ void plocFlags( const PlocArrays B, const unsigned count ) {
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= count ) {
		return;
	}

	const int partner = B.nearest[i];
	const bool mutual = partner >= 0 && B.nearest[partner] == (int) i;
	const bool merges = mutual && (int) i < partner;
	const bool absorbed = mutual && (int) i > partner;
	B.flags[i] = ( absorbed ? 0ull : 1ull ) | ( merges ? ( 1ull << 32 ) : 0ull );
}

__global__ void plocMerge( const PlocArrays B, const unsigned count, const unsigned nextNode ) {
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= count ) {
		return;
	}

	const unsigned long long flag = B.flags[i], before = B.scan[i];

	if( i == count - 1u ) {
		B.totals[0] = before + flag;
	}
	if( ( flag & 1ull ) == 0ull ) {
		return;
	}

	const unsigned slot = (unsigned) ( before & 0xFFFFFFFFull );
	int id = B.clusters[i];

	if( ( flag >> 32 ) != 0ull ) {
		const int other = B.clusters[B.nearest[i]];
		const int merged = (int) ( nextNode + (unsigned) ( before >> 32 ) );
		const float4 aLo = B.boxMin[id], aHi = B.boxMax[id], bLo = B.boxMin[other], bHi = B.boxMax[other];
		B.boxMin[merged] = make_float4( fminf( aLo.x, bLo.x ), fminf( aLo.y, bLo.y ), fminf( aLo.z, bLo.z ), 0.0f );
		B.boxMax[merged] = make_float4( fmaxf( aHi.x, bHi.x ), fmaxf( aHi.y, bHi.y ), fmaxf( aHi.z, bHi.z ), 0.0f );
		// the child with the bigger surface area first, as BVH::combineNodes orders them (BVH.cpp:335-343): the walk
		// visits children in array order, and a hit found early culls what follows (pt_bvh.cl:107)
		const bool swap = halfArea( make_float3( bLo.x, bLo.y, bLo.z ), make_float3( bHi.x, bHi.y, bHi.z ) )
			> halfArea( make_float3( aLo.x, aLo.y, aLo.z ), make_float3( aHi.x, aHi.y, aHi.z ) );
		B.left[merged] = swap ? other : id;
		B.right[merged] = swap ? id : other;
		B.parent[merged] = -1;
		B.parent[id] = merged;
		B.parent[other] = merged;
		const bool leaf = id < (int) B.numFaces && other < (int) B.numFaces;
		B.size[merged] = leaf ? 1u : 1u + B.size[id] + B.size[other];
		B.faces[merged] = B.faces[id] + B.faces[other];
		id = merged;
	}

	B.clustersNext[slot] = id;
}

// one thread per node id: its position in depth-first order (left subtree first) and the position of its first face,
// then the record in the reference's wire format and, for leaves, the faces in leaf order
__global__ void plocFlatten( const PlocArrays B, const unsigned numIds, const unsigned total ) {
	const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;

	if( id >= numIds ) {
		return;
	}

	const int n = (int) B.numFaces;
	const bool isFace = (int) id < n;
	const int up0 = B.parent[id];

	if( isFace && up0 >= 0 && B.left[up0] < n && B.right[up0] < n ) {
		return;   // one of a 2-face leaf's faces: the leaf's thread emits it
	}

	unsigned index = 0, firstFace = 0;

	for( int node = (int) id, up = up0; up >= 0; node = up, up = B.parent[up] ) {
		const bool second = B.right[up] == node;
		index += 1u + ( second ? B.size[B.left[up]] : 0u );
		firstFace += second ? B.faces[B.left[up]] : 0u;
	}

	const float4 lo = B.boxMin[id], hi = B.boxMax[id];
	pbr_bvh_node out;
	out.bbMin.x = lo.x; out.bbMin.y = lo.y; out.bbMin.z = lo.z;
	out.bbMax.x = hi.x; out.bbMax.y = hi.y; out.bbMax.z = hi.z;
	const bool leaf2 = !isFace && B.left[id] < n && B.right[id] < n;

	if( isFace || leaf2 ) {
		const int members[2] = { isFace ? (int) id : B.left[id], leaf2 ? B.right[id] : -1 };

		for( int k = 0; k < 2; k++ ) {
			if( members[k] >= 0 ) {
				const unsigned face = (unsigned) ( B.keysSorted[members[k]] & 0xFFFFFFull );
				B.facesVOut[firstFace + k] = B.facesV[face];
				B.facesNOut[firstFace + k] = B.facesN[face];
			}
		}

		out.bbMin.w = (float) firstFace;
		out.bbMax.w = leaf2 ? (float) ( firstFace + 1u ) : -1.0f;
	}
	else {
		// on a miss the walk continues behind this subtree; -1 ends it (pt_bvh.cl:102,122)
		const unsigned next = index + B.size[id];
		out.bbMin.w = -1.0f;
		out.bbMax.w = ( next < total ) ? (float) next : -1.0f;
	}

	B.nodesOut[index] = out;
}

}  // namespace ptb
