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

// MathHelp::triCalcAABB without Phong tessellation (MathHelp.cpp:239-257, :11-31):
// component-wise min / max of the three corners.
void triBox( Tri* tri, const std::vector<float>& v ) {
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
	(void) normals;
	const int maxFaces = Cfg::get().value<int>( Cfg::BVH_MAXFACES );
	mMaxFaces = (uint32_t) ( ( maxFaces > 1 ) ? maxFaces : 1 );
	mSahFacesLimit = Cfg::get().value<uint32_t>( Cfg::BVH_SAHFACESLIMIT );

	std::vector<BVHNode*> subTrees;
	int32_t offset = 0;

	for( size_t i = 0; i < sceneObjects.size(); i++ ) {
		const object3D& obj = sceneObjects[i];
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

			triBox( &t, vertices );
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
		std::sort( pos.begin(), pos.end(), [k]( uint32_t a, uint32_t b ) { return k[a] < k[b]; } );

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

			if( sah < bestSAH ) {
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
