// Host BVH builder: reproduces the tree the reference's BVH class builds
// (source/accelstructures/BVH.{h,cpp}, AccelStructure.h, MathHelp.cpp) — same split
// decisions, same child order, same DFS numbering, same skip-ahead marks — because the
// low bits of every hit distance depend on the exact leaf boxes (pt_intersect.cl:96-120).
// The construction itself is this project's own: index ranges over one triangle array and
// prefix/suffix sweeps instead of by-value vector<Tri> copies per recursion.
#pragma once

#include <cstdint>
#include <deque>
#include <vector>

#include "scene_model.h"

namespace pbr {

#define ACCELSTRUCT_BVH 0

// Tri, source/accelstructures/AccelStructure.h:13-18
struct Tri {
	uint4_t face;     // w: global face index
	uint4_t normals;  // w: global face index
	float bbMin[3];
	float bbMax[3];
};

// BVHNode, source/accelstructures/BVH.h:16-27
struct BVHNode {
	BVHNode* leftChild = nullptr;
	BVHNode* rightChild = nullptr;
	BVHNode* parent = nullptr;
	std::vector<Tri> faces;
	float bbMin[3] = { 0.0f, 0.0f, 0.0f };
	float bbMax[3] = { 0.0f, 0.0f, 0.0f };
	uint32_t id = 0;
	uint32_t depth = 0;
	uint32_t numSkipsToHere = 0;
	bool skipNextLeft = false;
};


class AccelStructure {

	public:
		virtual ~AccelStructure() {}

};


class BVH : public AccelStructure {

	public:
		// Reads bvh.max_faces, bvh.sah_faces_limit, bvh.skip_ahead, bvh.skip_ahead_compare from Cfg,
		// as the reference does (BVH.cpp:57,157,349,771).
		BVH(
			const std::vector<object3D>& sceneObjects,
			const std::vector<float>& vertices,
			const std::vector<float>& normals
		);

		std::vector<BVHNode*> getContainerNodes() { return mContainerNodes; }
		uint32_t getDepth() const { return mDepthReached; }
		std::vector<BVHNode*> getLeafNodes() { return mLeafNodes; }
		std::vector<BVHNode*> getNodes() { return mNodes; }
		const std::vector<BVHNode*>& nodes() const { return mNodes; }
		BVHNode* getRoot() { return mRoot; }
		uint32_t numSkipped() const { return mSkipped; }

		static float getSurfaceArea( const float bbMin[3], const float bbMax[3] );

		// scripts/bvh_sweep.py only (see bvh_builder.cpp); the product leaves it 0
		enum { LAB_STABLE_SORT = 1, LAB_ONE_TREE = 2, LAB_LAST_BEST = 4 };
		static unsigned sLabFlags;

	private:
		BVHNode* newNode();
		BVHNode* buildTree( std::vector<Tri>& tris, std::vector<uint32_t>& order, size_t lo, size_t hi, uint32_t depth );
		size_t splitBySAH( const std::vector<Tri>& tris, std::vector<uint32_t>& order, size_t lo, size_t hi );
		size_t splitByMean( const std::vector<Tri>& tris, std::vector<uint32_t>& order, size_t lo, size_t hi );
		BVHNode* makeContainerNode( const std::vector<BVHNode*>& subTrees, bool isRoot );
		void groupTreesToNodes( const std::vector<BVHNode*>& nodes, BVHNode* parent, uint32_t depth );
		void combineNodes( size_t numSubTrees );
		void orderNodesByTraversal();
		void skipAheadOfNodes();

		std::deque<BVHNode> mArena;
		std::vector<BVHNode*> mContainerNodes;
		std::vector<BVHNode*> mLeafNodes;
		std::vector<BVHNode*> mNodes;
		BVHNode* mRoot = nullptr;

		uint32_t mMaxFaces = 2;
		uint32_t mSahFacesLimit = 100000;
		uint32_t mDepthReached = 0;
		uint32_t mSkipped = 0;

};

}  // namespace pbr
