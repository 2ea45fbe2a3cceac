// Host-callable interface of the device BVH builder (pt_bvh_build.hip).
#pragma once
#include "pt_device_common.h"

enum { PT_BUILDER_LBVH = 0, PT_BUILDER_PLOC = 1 };

struct PtBuildInput
{
	const PtVertex *vertices;  // device
	const uint32_t *indices;   // device
	uint32_t nTris;
	int leafMax;  // triangles per leaf (1..PT_LEAF_MAX); 0 = PT_LEAF_DEFAULT
	int builder;  // PT_BUILDER_LBVH (Karras topology + refit) or PT_BUILDER_PLOC (locally-ordered clustering)
	// Box mode (top-level structure over instances): primitives are nTris axis-aligned boxes instead of
	// triangles (vertices/indices unused), every leaf holds exactly one and its reference is PT_LEAF_FLAG | index,
	// no triangle array is produced.  nodeBase is added to every inner-node reference (the caller places the
	// nodes behind another tree in one array); padExp: boxes are padded by 2^-padExp * max|coordinate|.
	const float4 *boxLo = nullptr, *boxHi = nullptr;
	uint32_t nodeBase = 0;
	int padExp = 18;
	// leaf triangle records: 3 = once per dominant ray axis (coordinates rotated, DESIGN.md 3: nothing per triangle in the
	// traversal), 1 = the identity plane only (the traversal permutes with selects): a third of the bytes to write and keep,
	// for scenes whose traversal waits for HBM anyway
	int triPlanes = 3;
	// recorded on the stream behind the build's last kernel when given: the tree is complete there; what follows in
	// ptBuildBvh4 (two small read-backs, the release of the scratch arena) is host work with an idle device
	hipEvent_t doneEvent = nullptr;
};

struct PtBuildOutput
{
	PtNode4 *nodes = nullptr;   // device, nNodes used
	PtTri48 *tris = nullptr;    // device, triPlanes planes x nTris, leaf order; plane r = coordinates rotated to ((r+1)%3,(r+2)%3,r);
	                            // the identity plane (x, y, z) is the LAST one: plane 2 of three, or the only one
	int triPlanes = 3;
	uint32_t *codes = nullptr;  // device, Morton code per primitive (primitive order)
	uint32_t *order = nullptr;  // device, primitive ids sorted by (code, id)
	uint32_t nNodes = 0, nTris = 0, depth = 0;
	float cmin[3] = { 0, 0, 0 }, cmax[3] = { 0, 0, 0 };
};

// Builds the quantised BVH4 on `stream`; synchronises the stream (one-time build, as the reference
// blocks in vkQueueWaitIdle).  On success the caller owns the four device allocations in `out`.
hipError_t ptBuildBvh4(const PtBuildInput &in, PtBuildOutput &out, hipStream_t stream);

// Quad leaves for k_paths (PtQuad64): a copy of the node array in which every two-triangle leaf whose triangles are the
// fan (a, b, c), (a, c, d) -- what an OBJ quad face becomes -- is replaced by a reference PT_LEAF_FLAG | PT_LEAF_QUAD << 28 |
// quad index.  Only done when at least minShare of the triangles end up in quads (a wave pays for both leaf formats when
// its lanes mix them); otherwise nQuads == 0 and nothing is allocated.  The caller owns the two allocations.
struct PtQuadOutput
{
	PtNode4 *nodes = nullptr;    // device, nNodes
	PtQuad64 *quads = nullptr;   // device, 3 planes x nQuads
	uint32_t nQuads = 0;
};
// (tris: the IDENTITY plane of the triangle records)
hipError_t ptBuildQuadLeaves(const PtNode4 *nodes, uint32_t nNodes, const PtTri48 *tris, uint32_t nTris, float minShare,
                             PtQuadOutput &out, hipStream_t stream);

// Entry node for k_paths (pt_wave_traverse.h): the top of the tree under `root` flattened into one node of up to
// PT_ENTRY_WIDTH children, as a decoded record of PT_ENTRY_REC 16-byte units in device memory (*rec; nullptr when the tree
// is empty).  `nodes` is the array k_paths traverses (with quad references when nQuads != 0); leaf children get the exact
// boxes of their triangles / quads -- unless instanceLeaves: `root` is the root of a top-level tree, whose leaves are
// instances and keep the tree's boxes.  *flat: every child is a triangle leaf (the node is the whole scene).  Synchronises
// the stream.  The caller owns the allocation.
// (tris: the IDENTITY plane of the triangle records)
hipError_t ptBuildEntryNode(const PtNode4 *nodes, uint32_t nNodes, uint32_t root, const PtTri48 *tris, uint32_t nTris,
                            const PtQuad64 *quads, uint32_t nQuads, bool instanceLeaves, uint4 **rec, bool *flat, hipStream_t stream);

// Several Models in one scene: copies the tree `part` (built over the triangles [triBase, triBase + part.nTris) of the
// concatenated scene, with local primitive ids) into the shared arrays -- nodes at nodeBase (inner references + nodeBase, leaf
// references + triBase), triangle records at triBase of every plane of `tris` (plane pitch totalTris; primitive ids + triBase),
// codes and sorted order likewise.  Asynchronous on `stream`.
hipError_t ptRelocateBlas(const PtBuildOutput &part, uint32_t nodeBase, uint32_t triBase, uint32_t totalTris, PtNode4 *nodes,
                          PtTri48 *tris, uint32_t *codes, uint32_t *order, hipStream_t stream);
