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

}  // namespace yhh
#endif
