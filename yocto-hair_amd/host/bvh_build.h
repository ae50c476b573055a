// bvh_build.h — host-side BVH builder of the product (g++).
//
// Builds the reference's tree exactly (pt.cpp:557-650: breadth-first queue,
// leaves of <= 4 primitives, split_middle = std::partition at the midpoint of
// the centroid bounds along their largest axis, median fallback) so that the
// device traversal visits primitives in the reference's order and exact-t ties
// resolve identically (math.h:3450 keeps the LATER primitive on a tie).
#ifndef YH_BVH_BUILD_H_
#define YH_BVH_BUILD_H_
#include <vector>

namespace yhh {

struct Box {
  float min[3], max[3];
};
struct Node {  // the reference's bvh_node (pt.h:243-249)
  Box           bbox;
  int           start;
  short         num;
  bool          internal;
  unsigned char axis;
};
struct Tree {
  std::vector<Node> nodes;
  std::vector<int>  primitives;
  int               max_depth = 0;
};

// boxes[i] bounds primitive i; its centre is (min + max) / 2 (math.h:3008)
void build_bvh(Tree& tree, const std::vector<Box>& boxes);

// 4-wide node for the device: two levels of the binary tree collapsed into one
// 128-byte record of four 32-byte slots, so that a traversal step is ONE
// dependent fetch in which the four lanes of a "quad" each test one of the
// reference's node boxes.
//   slots 0,1 = children of the binary node's left child (or the left child
//   itself in slot 0 when it is a leaf); slots 2,3 likewise for the right child.
//   ref: bits 31,30 set = leaf, bits 27..29 = primitive count, bits 0..26 =
//        first leaf slot; otherwise the index of the child wide node.
//        0xFFFFFFFF = empty slot (its box is inverted and can never be hit).
//   axes (same in the four slots) = split axis of the binary node | left
//        child's << 2 | right child's << 4: the three comparisons that
//        reproduce the reference's near-first visiting order (pt.cpp:887-893),
//        so leaves are visited in the same order as in the binary tree.
struct WideSlot {
  float    bmin[3], bmax[3];
  unsigned ref, axes;
};
struct WideNode {
  WideSlot slot[4];
};
static_assert(sizeof(WideNode) == 128, "wide node is 128 bytes");
// Wide nodes come out in breadth-first order (the first K nodes are the top of
// the tree). Returns the depth of the wide tree.
int collapse_wide(const Tree& tree, std::vector<WideNode>& out);

// 8-wide node: THREE levels of the binary tree in one 256-byte record of eight slots (same slot format), for the
// kernels whose paths are bound by the number of dependent steps of one ray (dev_trace.h, YH_MODE_W8 / YH_MODE_OCT):
// a node step covers three binary levels instead of two. Slot o = s1 << 2 | s2 << 1 | s3 holds the great-grandchild
// reached by sides s1, s2, s3; a child (grandchild) that is a leaf sits in the first slot of its group of four (two).
//   axes (same in the eight slots): bits 0-1 the binary node's split axis, bits 2-3 / 4-5 its left / right child's,
//   bits 6-13 the four grandchildren's (index 2 * s1 + s2): the seven comparisons of the reference's near-first
//   order (pt.cpp:887-893) over three levels.
struct WideNode8 {
  WideSlot slot[8];
};
static_assert(sizeof(WideNode8) == 256, "8-wide node is 256 bytes");
int collapse_wide8(const Tree& tree, std::vector<WideNode8>& out);

// 16-wide node: FOUR levels per 512-byte record, for the kernels that give a path sixteen lanes (dev_trace.h,
// YH_MODE_HEX). Slot o = s1 << 3 | s2 << 2 | s3 << 1 | s4; a subtree that ends in a leaf above the fourth level sits in
// the first slot of its group. axes: bits 0-1 the node's split axis, 2-5 its two children's, 6-13 the four
// grandchildren's (index 2 s1 + s2), 14-29 the eight great-grandchildren's (index 4 s1 + 2 s2 + s3).
struct WideNode16 {
  WideSlot slot[16];
};
static_assert(sizeof(WideNode16) == 512, "16-wide node is 512 bytes");
int collapse_wide16(const Tree& tree, std::vector<WideNode16>& out);

}  // namespace yhh
#endif
