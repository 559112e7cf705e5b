#include "bvh_builder.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <stdexcept>

#include "Cfg.h"

namespace pbr {

namespace {

inline float minf( float a, float b ) { return ( b < a ) ? b : a; }
inline float maxf( float a, float b ) { return ( a < b ) ? b : a; }

struct Box {
	float lo[3], hi[3];

	void set( const float* mn, const float* mx ) {
		for( int k = 0; k < 3; k++ ) {
			lo[k] = mn[k];
			hi[k] = mx[k];
		}
	}

	void grow( const float* mn, const float* mx ) {
		for( int k = 0; k < 3; k++ ) {
			lo[k] = minf( mn[k], lo[k] );
			hi[k] = maxf( mx[k], hi[k] );
		}
	}
};

// The sort key of sortFacesCmp (BVH.cpp:26-31): centre of the triangle's box on one axis.
inline float centre( const Tri& t, int axis ) {
	return ( t.bbMin[axis] + t.bbMax[axis] ) * 0.5f;
}

// ---- small vector helpers with glm's evaluation order (dot: ( x + y ) + z; normalize: v * ( 1 / sqrt( dot ) )) ----
struct V3 {
	float x, y, z;
};

inline V3 operator+( V3 a, V3 b ) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-( V3 a, V3 b ) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator*( V3 a, float s ) { return { a.x * s, a.y * s, a.z * s }; }
inline V3 operator*( float s, V3 a ) { return { s * a.x, s * a.y, s * a.z }; }
inline float dot3( V3 a, V3 b ) { return ( a.x * b.x + a.y * b.y ) + a.z * b.z; }
inline V3 cross3( V3 a, V3 b ) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline V3 normalize3v( V3 a ) { return a * ( 1.0f / std::sqrt( dot3( a, a ) ) ); }
inline V3 min3v( V3 a, V3 b ) { return { minf( a.x, b.x ), minf( a.y, b.y ), minf( a.z, b.z ) }; }
inline V3 max3v( V3 a, V3 b ) { return { maxf( a.x, b.x ), maxf( a.y, b.y ), maxf( a.z, b.z ) }; }

// MathHelp::projectOnPlane / phongTessellate, MathHelp.cpp:213-231
inline V3 onPlane( V3 q, V3 p, V3 n ) { return q - dot3( q - p, n ) * n; }

V3 phongPoint( V3 p1, V3 p2, V3 p3, V3 n1, V3 n2, V3 n3, float alpha, float u, float v ) {
	const float w = 1.0f - u - v;
	const V3 bary = ( p1 * u + p2 * v ) + p3 * w;
	const V3 tess = ( u * onPlane( bary, p1, n1 ) + v * onPlane( bary, p2, n2 ) ) + w * onPlane( bary, p3, n3 );
	return ( 1.0f - alpha ) * bary + alpha * tess;
}

// MathHelp::triCalcAABB (MathHelp.cpp:250-310, getAABB :20-36): component-wise min / max of the three corners;
// with render.phong_tessellation > 0 and unequal vertex normals the box also covers the Phong-tessellated patch —
// its apex above the plane (triThicknessAndSidedrop, :325-378: the patch's extremum along the geometric normal) and
// nine points along its curved edges ("sidedrop").
void triBox( Tri* tri, const std::vector<float>& v, const std::vector<float>& normals, float alpha ) {
	const uint32_t idx[3] = { tri->face.x, tri->face.y, tri->face.z };

	for( int k = 0; k < 3; k++ ) {
		float mn = v.at( (size_t) idx[0] * 3 + k );
		float mx = mn;

		for( int c = 1; c < 3; c++ ) {
			const float x = v.at( (size_t) idx[c] * 3 + k );
			mn = ( mn < x ) ? mn : x;
			mx = ( mx > x ) ? mx : x;
		}

		tri->bbMin[k] = mn;
		tri->bbMax[k] = mx;
	}

	if( !( alpha > 0.0f ) ) {
		return;
	}

	const uint32_t nidx[3] = { tri->normals.x, tri->normals.y, tri->normals.z };
	V3 p[3], n[3];

	for( int c = 0; c < 3; c++ ) {
		p[c] = { v.at( (size_t) idx[c] * 3 ), v.at( (size_t) idx[c] * 3 + 1 ), v.at( (size_t) idx[c] * 3 + 2 ) };
		n[c] = { normals.at( (size_t) nidx[c] * 3 ), normals.at( (size_t) nidx[c] * 3 + 1 ), normals.at( (size_t) nidx[c] * 3 + 2 ) };
	}

	const V3 test = ( n[0] - n[1] ) + ( n[1] - n[2] );

	if( std::fabs( test.x ) <= 0.000001f && std::fabs( test.y ) <= 0.000001f && std::fabs( test.z ) <= 0.000001f ) {
		return;   // equal normals: nothing to tessellate
	}

	const V3 e12 = p[1] - p[0], e13 = p[2] - p[0], e23 = p[2] - p[1], e31 = p[0] - p[2];
	const V3 c12 = alpha * ( dot3( n[1], e12 ) * n[1] - dot3( n[0], e12 ) * n[0] );
	const V3 c23 = alpha * ( dot3( n[2], e23 ) * n[2] - dot3( n[1], e23 ) * n[1] );
	const V3 c31 = alpha * ( dot3( n[0], e31 ) * n[0] - dot3( n[2], e31 ) * n[2] );
	const V3 ng = normalize3v( cross3( e12, e13 ) );

	const float kTmp = dot3( ng, c12 - c23 - c31 );
	const float k = 1.0f / ( 4.0f * dot3( ng, c23 ) * dot3( ng, c31 ) - kTmp * kTmp );
	float pu = k * ( 2.0f * dot3( ng, c23 ) * dot3( ng, c31 + e31 ) + dot3( ng, c23 - e23 ) * dot3( ng, c12 - c23 - c31 ) );
	float pv = k * ( 2.0f * dot3( ng, c31 ) * dot3( ng, c23 - e23 ) + dot3( ng, c31 + e31 ) * dot3( ng, c12 - c23 - c31 ) );
	pu = ( pu < 0.0f || pu > 1.0f ) ? 0.0f : pu;
	pv = ( pv < 0.0f || pv > 1.0f ) ? 0.0f : pv;

	const V3 apex = phongPoint( p[0], p[1], p[2], n[0], n[1], n[2], alpha, pu, pv );
	const float thickness = dot3( ng, apex - p[0] );

	static const float edgeUV[9][2] = {
		{ 0.0f, 0.5f }, { 0.5f, 0.0f }, { 0.5f, 0.5f }, { 0.25f, 0.75f }, { 0.75f, 0.25f },
		{ 0.25f, 0.0f }, { 0.75f, 0.0f }, { 0.0f, 0.25f }, { 0.0f, 0.75f }
	};
	V3 lo = { tri->bbMin[0], tri->bbMin[1], tri->bbMin[2] };
	V3 hi = { tri->bbMax[0], tri->bbMax[1], tri->bbMax[2] };

	// raised corners first, then the edge points (MathHelp.cpp:302-309; min / max commute, the values do not change)
	for( int c = 0; c < 3; c++ ) {
		const V3 raised = p[c] + thickness * ng;
		lo = min3v( lo, raised );
		hi = max3v( hi, raised );
	}

	for( int s = 0; s < 9; s++ ) {
		const V3 q = phongPoint( p[0], p[1], p[2], n[0], n[1], n[2], alpha, edgeUV[s][0], edgeUV[s][1] );
		lo = min3v( lo, q );
		hi = max3v( hi, q );
	}

	tri->bbMin[0] = lo.x; tri->bbMin[1] = lo.y; tri->bbMin[2] = lo.z;
	tri->bbMax[0] = hi.x; tri->bbMax[1] = hi.y; tri->bbMax[2] = hi.z;
}

// BVH::longestAxis, BVH.cpp:577-587
int longestAxis( const BVHNode* node ) {
	const float sx = node->bbMax[0] - node->bbMin[0];
	const float sy = node->bbMax[1] - node->bbMin[1];
	const float sz = node->bbMax[2] - node->bbMin[2];

	if( sx > sy ) {
		return ( sx > sz ) ? 0 : 2;
	}

	return ( sy > sz ) ? 1 : 2;
}

}  // namespace


// MathHelp::getSurfaceArea, MathHelp.cpp:93-99 — evaluation order kept.
float BVH::getSurfaceArea( const float bbMin[3], const float bbMax[3] ) {
	const float xy = std::fabs( bbMax[0] - bbMin[0] ) * std::fabs( bbMax[1] - bbMin[1] );
	const float zy = std::fabs( bbMax[2] - bbMin[2] ) * std::fabs( bbMax[1] - bbMin[1] );
	const float xz = std::fabs( bbMax[0] - bbMin[0] ) * std::fabs( bbMax[2] - bbMin[2] );

	return 2.0f * ( xy + zy + xz );
}


// Variants for scripts/bvh_sweep.py only (which settings would give the node count the reference quotes for its test
// model, pathtracing.cl:75-76); 0 = the reference's builder as read from BVH.cpp.  Never set by the product.
unsigned BVH::sLabFlags = 0u;


BVHNode* BVH::newNode() {
	mArena.emplace_back();
	return &mArena.back();
}


// BVH::BVH, BVH.cpp:50-64: one tree per object (buildTreesFromObjects :203-245), grouped under
// container nodes (:602-628, :471-491), then combineNodes (:318-352).
BVH::BVH(
	const std::vector<object3D>& sceneObjects,
	const std::vector<float>& vertices,
	const std::vector<float>& normals
) {
	const float phongAlpha = Cfg::get().value<float>( Cfg::RENDER_PHONGTESS );
	const int maxFaces = Cfg::get().value<int>( Cfg::BVH_MAXFACES );
	mMaxFaces = (uint32_t) ( ( maxFaces > 1 ) ? maxFaces : 1 );
	mSahFacesLimit = Cfg::get().value<uint32_t>( Cfg::BVH_SAHFACESLIMIT );

	std::vector<BVHNode*> subTrees;
	int32_t offset = 0;
	std::vector<object3D> merged;

	if( ( sLabFlags & LAB_ONE_TREE ) != 0u && sceneObjects.size() > 1 ) {
		merged.resize( 1 );
		merged[0].oName = "all";

		for( const object3D& obj : sceneObjects ) {
			merged[0].facesV.insert( merged[0].facesV.end(), obj.facesV.begin(), obj.facesV.end() );
			merged[0].facesVN.insert( merged[0].facesVN.end(), obj.facesVN.begin(), obj.facesVN.end() );
		}
	}

	const std::vector<object3D>& objects = merged.empty() ? sceneObjects : merged;

	for( size_t i = 0; i < objects.size(); i++ ) {
		const object3D& obj = objects[i];
		const size_t numFaces = obj.facesV.size() / 3;

		if( numFaces == 0 ) {
			throw std::runtime_error( "[BVH] object \"" + obj.oName + "\" has no faces" );
		}

		// ModelLoader::getFacesOfObject + BVH::facesToTriStructs (BVH.cpp:363-379)
		std::vector<Tri> tris( numFaces );

		for( size_t j = 0; j < numFaces; j++ ) {
			Tri& t = tris[j];
			t.face = { obj.facesV[j * 3], obj.facesV[j * 3 + 1], obj.facesV[j * 3 + 2], (uint32_t) ( offset + (int32_t) j ) };

			if( obj.facesVN.size() >= ( j + 1 ) * 3 ) {
				t.normals = { obj.facesVN[j * 3], obj.facesVN[j * 3 + 1], obj.facesVN[j * 3 + 2], t.face.w };
			}
			else {
				t.normals = { 0, 0, 0, t.face.w };
			}

			triBox( &t, vertices, normals, ( obj.facesVN.size() >= ( j + 1 ) * 3 ) ? phongAlpha : 0.0f );
		}

		offset += (int32_t) numFaces;

		std::vector<uint32_t> order( numFaces );

		for( size_t j = 0; j < numFaces; j++ ) {
			order[j] = (uint32_t) j;
		}

		subTrees.push_back( this->buildTree( tris, order, 0, numFaces, 1 ) );
	}

	if( subTrees.empty() ) {
		throw std::runtime_error( "[BVH] scene has no objects" );
	}

	mRoot = this->makeContainerNode( subTrees, true );
	this->groupTreesToNodes( subTrees, mRoot, mDepthReached );
	this->combineNodes( subTrees.size() );
}


// BVH::buildTree, BVH.cpp:133-193.  `order[lo,hi)` is the node's face list in the order the
// reference's vector<Tri> would hold it.
BVHNode* BVH::buildTree( std::vector<Tri>& tris, std::vector<uint32_t>& order, size_t lo, size_t hi, uint32_t depth ) {
	BVHNode* node = this->newNode();
	mContainerNodes.push_back( node );

	// makeNode, BVH.cpp:637-664
	Box box;
	box.set( tris[order[lo]].bbMin, tris[order[lo]].bbMax );

	for( size_t i = lo + 1; i < hi; i++ ) {
		box.grow( tris[order[i]].bbMin, tris[order[i]].bbMax );
	}

	for( int k = 0; k < 3; k++ ) {
		node->bbMin[k] = box.lo[k];
		node->bbMax[k] = box.hi[k];
	}

	node->depth = depth;
	mDepthReached = ( depth > mDepthReached ) ? depth : mDepthReached;

	const size_t n = hi - lo;
	size_t split = 0;

	if( n > mMaxFaces ) {
		split = ( n <= mSahFacesLimit )
		      ? this->splitBySAH( tris, order, lo, hi )
		      : this->splitByMean( tris, order, lo, hi );
	}

	// Leaf: few enough faces, or the split left one side empty (BVH.cpp:145-183).  All faces are
	// kept on the node; the flattening addresses only the first two.
	if( split == 0 || split == n ) {
		node->faces.reserve( n );

		for( size_t i = lo; i < hi; i++ ) {
			node->faces.push_back( tris[order[i]] );
		}

		return node;
	}

	node->leftChild = this->buildTree( tris, order, lo, lo + split, depth + 1 );
	node->rightChild = this->buildTree( tris, order, lo + split, hi, depth + 1 );

	return node;
}


// buildWithSAH + splitBySAH + growAABBsForSAH, BVH.cpp:283-294, :807-856, :501-551.
// Per axis: sort (std::sort, unstable, same comparator outcomes as sortFacesCmp) a copy of the
// node's ORIGINAL face order, sweep prefix/suffix boxes, keep the cheapest split over all axes
// (strict <, one running best across the three axes).  The winning axis' sorted order becomes
// the children's face order.  Returns the left count, 0 if no split beat FLT_MAX.
size_t BVH::splitBySAH( const std::vector<Tri>& tris, std::vector<uint32_t>& order, size_t lo, size_t hi ) {
	const size_t n = hi - lo;
	float bestSAH = FLT_MAX;
	size_t bestSplit = 0;
	std::vector<uint32_t> bestOrder;
	std::vector<uint32_t> sorted( n );
	std::vector<float> key( n );
	std::vector<float> leftSA( n - 1 ), rightSA( n - 1 );

	for( int axis = 0; axis <= 2; axis++ ) {
		// Sort positions 0..n-1 of the original order by the centre key.
		std::vector<uint32_t> pos( n );

		for( size_t i = 0; i < n; i++ ) {
			pos[i] = (uint32_t) i;
			key[i] = centre( tris[order[lo + i]], axis );
		}

		const float* k = key.data();

		if( ( sLabFlags & LAB_STABLE_SORT ) != 0u ) {
			std::stable_sort( pos.begin(), pos.end(), [k]( uint32_t a, uint32_t b ) { return k[a] < k[b]; } );
		}
		else {
			std::sort( pos.begin(), pos.end(), [k]( uint32_t a, uint32_t b ) { return k[a] < k[b]; } );
		}

		for( size_t i = 0; i < n; i++ ) {
			sorted[i] = order[lo + pos[i]];
		}

		Box box;

		for( size_t i = 0; i + 1 < n; i++ ) {
			const Tri& f = tris[sorted[i]];

			if( i == 0 ) {
				box.set( f.bbMin, f.bbMax );
			}
			else {
				box.grow( f.bbMin, f.bbMax );
			}

			leftSA[i] = getSurfaceArea( box.lo, box.hi );
		}

		for( size_t i = n - 1; i-- > 0; ) {
			const Tri& f = tris[sorted[i + 1]];

			if( i == n - 2 ) {
				box.set( f.bbMin, f.bbMax );
			}
			else {
				box.grow( f.bbMin, f.bbMax );
			}

			rightSA[i] = getSurfaceArea( box.lo, box.hi );
		}

		size_t splitAfter = 0;

		for( uint32_t i = 0; i + 1 < (uint32_t) n; i++ ) {
			const float numLeft = (float) ( i + 1u );
			const float numRight = (float) ( (uint32_t) n - i - 1u );
			const float sah = leftSA[i] * numLeft + rightSA[i] * numRight;

			if( sah < bestSAH || ( ( sLabFlags & LAB_LAST_BEST ) != 0u && sah == bestSAH ) ) {
				bestSAH = sah;
				splitAfter = i + 1;
			}
		}

		if( splitAfter > 0 ) {
			bestSplit = splitAfter;
			bestOrder = sorted;
		}
	}

	if( bestSplit > 0 ) {
		std::copy( bestOrder.begin(), bestOrder.end(), order.begin() + lo );
	}

	return bestSplit;
}


// buildWithMeanSplit + getMean + splitFaces, BVH.cpp:255-273, :416-426, :867-942 — used above
// bvh.sah_faces_limit.  Per axis: split at the mean box centre (<= goes left, order kept);
// if a side stays empty, halve by position.  The reference scores the split with the left
// box's area times the left count plus the area of a DEFAULT-CONSTRUCTED right box
// (BVH.cpp:915-918 never fills bbMinR/bbMaxR); glm of the reference's era zero-initialises
// vec3, so the right term is 0 here.
size_t BVH::splitByMean( const std::vector<Tri>& tris, std::vector<uint32_t>& order, size_t lo, size_t hi ) {
	const size_t n = hi - lo;
	float bestSAH = FLT_MAX;
	size_t bestSplit = 0;
	std::vector<uint32_t> bestOrder, left, right;

	for( int axis = 0; axis <= 2; axis++ ) {
		float sum = 0.0f;

		for( size_t i = lo; i < hi; i++ ) {
			const Tri& t = tris[order[i]];
			sum += 0.5f * ( t.bbMin[axis] + t.bbMax[axis] );
		}

		const float pos = sum / (float) n;
		left.clear();
		right.clear();

		for( size_t i = lo; i < hi; i++ ) {
			if( centre( tris[order[i]], axis ) <= pos ) {
				left.push_back( order[i] );
			}
			else {
				right.push_back( order[i] );
			}
		}

		if( left.empty() || right.empty() ) {
			left.clear();
			right.clear();

			for( size_t i = 0; i < n; i++ ) {
				( ( i < n / 2 ) ? left : right ).push_back( order[lo + i] );
			}
		}

		float sah = FLT_MAX;

		if( !left.empty() && !right.empty() ) {
			Box box;
			box.set( tris[left[0]].bbMin, tris[left[0]].bbMax );

			for( size_t i = 1; i < left.size(); i++ ) {
				box.grow( tris[left[i]].bbMin, tris[left[i]].bbMax );
			}

			const float zero[3] = { 0.0f, 0.0f, 0.0f };
			const float leftSA = getSurfaceArea( box.lo, box.hi );
			const float rightSA = getSurfaceArea( zero, zero );
			sah = leftSA * (float) left.size() + rightSA * (float) right.size();
		}

		if( sah < bestSAH ) {
			bestSAH = sah;
			bestSplit = left.size();
			bestOrder = left;
			bestOrder.insert( bestOrder.end(), right.begin(), right.end() );
		}
	}

	if( bestSplit > 0 ) {
		std::copy( bestOrder.begin(), bestOrder.end(), order.begin() + lo );
	}

	return bestSplit;
}


// BVH::makeContainerNode, BVH.cpp:602-628
BVHNode* BVH::makeContainerNode( const std::vector<BVHNode*>& subTrees, bool isRoot ) {
	if( subTrees.size() == 1 ) {
		return subTrees[0];
	}

	BVHNode* node = this->newNode();
	Box box;
	box.set( subTrees[0]->bbMin, subTrees[0]->bbMax );

	for( size_t i = 1; i < subTrees.size(); i++ ) {
		box.grow( subTrees[i]->bbMin, subTrees[i]->bbMax );
	}

	for( int k = 0; k < 3; k++ ) {
		node->bbMin[k] = box.lo[k];
		node->bbMax[k] = box.hi[k];
	}

	if( !isRoot ) {
		mContainerNodes.push_back( node );
	}

	return node;
}


// BVH::groupTreesToNodes + getMeanOfNodes + splitNodes, BVH.cpp:471-491, :435-444, :953-993.
// NB the reference averages and compares HALF EXTENTS ( bbMax - bbMin ) / 2, not centres.
void BVH::groupTreesToNodes( const std::vector<BVHNode*>& nodes, BVHNode* parent, uint32_t depth ) {
	if( nodes.size() == 1 ) {
		return;
	}

	parent->depth = depth;
	mDepthReached = ( depth > mDepthReached ) ? depth : mDepthReached;

	const int axis = longestAxis( parent );
	float sum = 0.0f;

	for( size_t i = 0; i < nodes.size(); i++ ) {
		sum += ( nodes[i]->bbMax[axis] - nodes[i]->bbMin[axis] ) * 0.5f;
	}

	const float mean = sum / (float) nodes.size();
	std::vector<BVHNode*> leftGroup, rightGroup;

	for( size_t i = 0; i < nodes.size(); i++ ) {
		const float half = ( nodes[i]->bbMax[axis] - nodes[i]->bbMin[axis] ) / 2.0f;
		( ( half < mean ) ? leftGroup : rightGroup ).push_back( nodes[i] );
	}

	if( leftGroup.empty() || rightGroup.empty() ) {
		leftGroup.clear();
		rightGroup.clear();

		for( size_t i = 0; i < nodes.size(); i++ ) {
			( ( i < nodes.size() / 2 ) ? leftGroup : rightGroup ).push_back( nodes[i] );
		}
	}

	parent->leftChild = this->makeContainerNode( leftGroup, false );
	this->groupTreesToNodes( leftGroup, parent->leftChild, depth + 1 );

	parent->rightChild = this->makeContainerNode( rightGroup, false );
	this->groupTreesToNodes( rightGroup, parent->rightChild, depth + 1 );
}


// BVH::combineNodes, BVH.cpp:318-352: parent links; the child with the bigger surface area goes
// left (strict >); DFS numbering; skip-ahead marks.
void BVH::combineNodes( size_t numSubTrees ) {
	if( numSubTrees > 1 ) {
		mNodes.push_back( mRoot );
	}

	mNodes.insert( mNodes.end(), mContainerNodes.begin(), mContainerNodes.end() );

	for( size_t i = 0; i < mNodes.size(); i++ ) {
		BVHNode* node = mNodes[i];

		if( !node->faces.empty() ) {
			mLeafNodes.push_back( node );
			continue;
		}

		node->leftChild->parent = node;
		node->rightChild->parent = node;

		const float leftSA = getSurfaceArea( node->leftChild->bbMin, node->leftChild->bbMax );
		const float rightSA = getSurfaceArea( node->rightChild->bbMin, node->rightChild->bbMax );

		if( rightSA > leftSA ) {
			std::swap( node->leftChild, node->rightChild );
		}
	}

	this->orderNodesByTraversal();

	if( Cfg::get().value<bool>( Cfg::BVH_SKIPAHEAD ) ) {
		this->skipAheadOfNodes();
	}
}


// BVH::orderNodesByTraversal, BVH.cpp:671-729: the order the stackless walk meets the nodes when
// every box is hit = depth-first, left first, starting at mNodes[0] (the root).
void BVH::orderNodesByTraversal() {
	std::vector<BVHNode*> ordered;
	std::vector<BVHNode*> stack;
	ordered.reserve( mNodes.size() );
	stack.push_back( mNodes[0] );

	while( !stack.empty() ) {
		BVHNode* node = stack.back();
		stack.pop_back();
		ordered.push_back( node );

		if( node->leftChild != nullptr ) {
			stack.push_back( node->rightChild );
			stack.push_back( node->leftChild );
		}
	}

	if( ordered.size() != mNodes.size() ) {
		throw std::runtime_error( "[BVH] node list and tree disagree" );
	}

	for( size_t i = 0; i < ordered.size(); i++ ) {
		ordered[i]->id = (uint32_t) i;
	}

	mNodes.swap( ordered );
}


// BVH::skipAheadOfNodes, BVH.cpp:770-795: mark a node whose (non-leaf) left child has at least
// bvh.skip_ahead_compare of its surface area; numSkipsToHere counts marks before the node.
void BVH::skipAheadOfNodes() {
	const float cmp = Cfg::get().value<float>( Cfg::BVH_SKIPAHEAD_CMP );
	uint32_t skippedLeft = 0;

	for( size_t i = 0; i < mNodes.size(); i++ ) {
		BVHNode* node = mNodes[i];
		node->numSkipsToHere = skippedLeft;

		if( node->leftChild != nullptr && node->leftChild->leftChild != nullptr ) {
			const float saNode = getSurfaceArea( node->bbMin, node->bbMax );
			const float saLeft = getSurfaceArea( node->leftChild->bbMin, node->leftChild->bbMax );

			if( saLeft / saNode >= cmp ) {
				node->skipNextLeft = true;
				skippedLeft++;
			}
		}
	}

	mSkipped = skippedLeft;
}

}  // namespace pbr
