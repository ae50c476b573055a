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

}  // namespace yhh
#endif
