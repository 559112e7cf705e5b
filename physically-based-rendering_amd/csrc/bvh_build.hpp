// Device-side BVH build (second half): an opt-in, fast builder that emits the
// reference's flat format — nodes in depth-first order, `bbMin.w` = first face or -1, `bbMax.w` =
// second face / miss link / -1 (PathTracer.cpp:238-310) with at most 2 faces per leaf (the
// reference's bvh.max_faces) and the faces re-ordered into leaf order — so that pbr_upload_scene and
// the kernels take it like the host builder's output.
//
// It is NOT the reference's builder (accelstructures/BVH.cpp: per-object trees, SAH sweeps over
// std::sort-ed faces, mean splits above 100 k faces, skip-ahead deletion): that one is replicated
// on the host (host/bvh_builder.cpp), takes 25 s for 2 M triangles and defines the trees the parity
// tests and the benchmark use.  This one is a linear BVH (Karras 2012: Morton order, binary radix
// tree, bottom-up boxes), milliseconds for the same input, for callers who want a scene on the
// device now and accept a tree of lower quality.  Images rendered over it agree with images over the
// reference's tree statistically, not bit for bit (the walk order decides ties and the triangle
// test starts from the leaf box's tNear, pt_intersect.cl:96-120).  Its result is deterministic: boxes
// are exact min / max, the keys are unique.
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
	unsigned long long* keys;           // Morton code << 32 | face index (unique)
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

__device__ __forceinline__ unsigned expandBits10( unsigned v ) {
	v = ( v * 0x00010001u ) & 0xFF0000FFu;
	v = ( v * 0x00000101u ) & 0x0F00F00Fu;
	v = ( v * 0x00000011u ) & 0xC30C30C3u;
	v = ( v * 0x00000005u ) & 0x49249249u;
	return v;
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
		q[k] = (unsigned) fminf( fmaxf( u * 1024.0f, 0.0f ), 1023.0f );
	}

	const unsigned code = ( expandBits10( q[0] ) << 2 ) | ( expandBits10( q[1] ) << 1 ) | expandBits10( q[2] );
	B.keys[face] = ( (unsigned long long) code << 32 ) | (unsigned long long) face;
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
			const unsigned face = (unsigned) ( B.keysSorted[sorted] & 0xFFFFFFFFull );
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

}  // namespace ptb
