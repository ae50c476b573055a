// bvh_build.cpp — see bvh_build.h. Restates pt.cpp:557-650.
#include "bvh_build.h"

#include <algorithm>
#include <deque>
#include <limits>

namespace yhh {

namespace {
// What the builder moves around: the centroid and the primitive id (16 bytes). The
// reference partitions records that also carry the box (pt.cpp:553-556); boxes are
// only needed once per node, so here they stay in the caller's array and a node's
// box is formed bottom-up (leaf: union of its primitives' boxes, internal node: union
// of its children's) — min / max are exact and order-independent, so the boxes, the
// splits and the primitive order are the reference's.
struct Prim {
  float center[3];
  int   primitive;
};
inline float fmin_(float a, float b) { return (a < b) ? a : b; }  // math.h:1779
inline float fmax_(float a, float b) { return (a > b) ? a : b; }
const float flt_max = std::numeric_limits<float>::max();
const float flt_min = std::numeric_limits<float>::lowest();

// split_middle (pt.cpp:564-595)
void split_middle(Prim* prims, int start, int end, int& mid, int& axis) {
  axis = 0;
  mid  = (start + end) / 2;
  float cmin[3] = {flt_max, flt_max, flt_max}, cmax[3] = {flt_min, flt_min, flt_min};
  for (int i = start; i < end; i++)
    for (int k = 0; k < 3; k++) {
      cmin[k] = fmin_(cmin[k], prims[i].center[k]);
      cmax[k] = fmax_(cmax[k], prims[i].center[k]);
    }
  float cs[3] = {cmax[0] - cmin[0], cmax[1] - cmin[1], cmax[2] - cmin[2]};
  if (cs[0] == 0 && cs[1] == 0 && cs[2] == 0) return;
  if (cs[0] >= cs[1] && cs[0] >= cs[2]) axis = 0;
  if (cs[1] >= cs[0] && cs[1] >= cs[2]) axis = 1;
  if (cs[2] >= cs[0] && cs[2] >= cs[1]) axis = 2;
  int   ax     = axis;
  float middle = (cmin[ax] + cmax[ax]) / 2;
  mid = (int)(std::partition(prims + start, prims + end, [ax, middle](const Prim& p) { return p.center[ax] < middle; }) -
              prims);
  if (mid == start || mid == end) mid = (start + end) / 2;
}
}  // namespace

void build_bvh(Tree& tree, const std::vector<Box>& boxes) {
  std::vector<Prim> prims(boxes.size());
  for (size_t i = 0; i < boxes.size(); i++) {
    for (int k = 0; k < 3; k++) prims[i].center[k] = (boxes[i].min[k] + boxes[i].max[k]) / 2;
    prims[i].primitive = (int)i;
  }
  auto& nodes = tree.nodes;
  nodes.clear();
  nodes.reserve(prims.size() * 2 + 1);
  struct Item {
    int node, start, end, depth;
  };
  // breadth-first, as the reference: a node's children are allocated when it is dequeued
  std::vector<Item> queue;
  queue.reserve(prims.size() + 1);
  queue.push_back({0, 0, (int)prims.size(), 1});
  nodes.emplace_back();
  tree.max_depth = 0;
  for (size_t head = 0; head < queue.size(); head++) {
    Item it        = queue[head];
    tree.max_depth = std::max(tree.max_depth, it.depth);
    Node node;
    if (it.end - it.start > 4) {  // bvh_max_prims (pt.cpp:598)
      int mid, axis;
      split_middle(prims.data(), it.start, it.end, mid, axis);
      node.internal = true;
      node.axis     = (unsigned char)axis;
      node.num      = 2;
      node.start    = (int)nodes.size();
      nodes.emplace_back();
      nodes.emplace_back();
      queue.push_back({node.start + 0, it.start, mid, it.depth + 1});
      queue.push_back({node.start + 1, mid, it.end, it.depth + 1});
    } else {
      node.internal = false;
      node.axis     = 0;
      node.num      = (short)(it.end - it.start);
      node.start    = it.start;
    }
    nodes[it.node] = node;
  }
  nodes.shrink_to_fit();
  // boxes bottom-up: children always have larger indices than their parent
  for (size_t n = nodes.size(); n-- > 0;) {
    Node& node = nodes[n];
    for (int k = 0; k < 3; k++) node.bbox.min[k] = flt_max, node.bbox.max[k] = flt_min;
    if (node.internal) {
      for (int c = 0; c < 2; c++) {
        const Box& b = nodes[(size_t)node.start + c].bbox;
        for (int k = 0; k < 3; k++) node.bbox.min[k] = fmin_(node.bbox.min[k], b.min[k]), node.bbox.max[k] = fmax_(node.bbox.max[k], b.max[k]);
      }
    } else {
      for (int i = node.start; i < node.start + node.num; i++) {
        const Box& b = boxes[(size_t)prims[i].primitive];
        for (int k = 0; k < 3; k++) node.bbox.min[k] = fmin_(node.bbox.min[k], b.min[k]), node.bbox.max[k] = fmax_(node.bbox.max[k], b.max[k]);
      }
    }
  }
  tree.primitives.resize(prims.size());
  for (size_t i = 0; i < prims.size(); i++) tree.primitives[i] = prims[i].primitive;
}


int collapse_wide(const Tree& tree, std::vector<WideNode>& out) {
  out.clear();
  const float inf = std::numeric_limits<float>::infinity();
  auto leaf_ref = [](const Node& n) { return 0xC0000000u | ((unsigned)n.num << 27) | (unsigned)n.start; };
  struct Item {
    int binary, wide, depth;
  };
  std::deque<Item> queue;
  auto new_wide = [&](int binary, int depth) {
    out.emplace_back();
    queue.push_back({binary, (int)out.size() - 1, depth});
    return (unsigned)out.size() - 1;
  };
  int max_depth = 1;
  new_wide(0, 1);
  while (!queue.empty()) {
    Item it = queue.front();
    queue.pop_front();
    max_depth = std::max(max_depth, it.depth);
    WideNode w;
    unsigned axes = 0;
    for (int s = 0; s < 4; s++) {
      for (int k = 0; k < 3; k++) w.slot[s].bmin[k] = inf, w.slot[s].bmax[k] = -inf;
      w.slot[s].ref = 0xFFFFFFFFu, w.slot[s].axes = 0;
    }
    auto set_slot = [&](int s, int binary) {
      const Node& n = tree.nodes[binary];
      for (int k = 0; k < 3; k++) w.slot[s].bmin[k] = n.bbox.min[k], w.slot[s].bmax[k] = n.bbox.max[k];
      w.slot[s].ref = n.internal ? new_wide(binary, it.depth + 1) : leaf_ref(n);
    };
    const Node& b = tree.nodes[it.binary];
    if (!b.internal) {  // a shape whose binary root is a leaf
      set_slot(0, it.binary);
    } else {
      axes = b.axis;
      for (int side = 0; side < 2; side++) {
        int         child = b.start + side;
        const Node& c     = tree.nodes[child];
        if (c.internal) {
          axes |= (unsigned)c.axis << (2 + 2 * side);
          set_slot(2 * side + 0, c.start + 0);
          set_slot(2 * side + 1, c.start + 1);
        } else {
          set_slot(2 * side, child);
        }
      }
    }
    for (int s = 0; s < 4; s++) w.slot[s].axes = axes;
    out[it.wide] = w;
  }
  return max_depth;
}

int collapse_wide8(const Tree& tree, std::vector<WideNode8>& out) {
  out.clear();
  const float inf = std::numeric_limits<float>::infinity();
  auto leaf_ref = [](const Node& n) { return 0xC0000000u | ((unsigned)n.num << 27) | (unsigned)n.start; };
  struct Item {
    int binary, wide, depth;
  };
  std::deque<Item> queue;
  auto new_wide = [&](int binary, int depth) {
    out.emplace_back();
    queue.push_back({binary, (int)out.size() - 1, depth});
    return (unsigned)out.size() - 1;
  };
  int max_depth = 1;
  new_wide(0, 1);
  while (!queue.empty()) {
    Item it = queue.front();
    queue.pop_front();
    max_depth = std::max(max_depth, it.depth);
    WideNode8 w;
    unsigned  axes = 0;
    for (int s = 0; s < 8; s++) {
      for (int k = 0; k < 3; k++) w.slot[s].bmin[k] = inf, w.slot[s].bmax[k] = -inf;
      w.slot[s].ref = 0xFFFFFFFFu, w.slot[s].axes = 0;
    }
    auto set_slot = [&](int s, int binary) {
      const Node& n = tree.nodes[binary];
      for (int k = 0; k < 3; k++) w.slot[s].bmin[k] = n.bbox.min[k], w.slot[s].bmax[k] = n.bbox.max[k];
      w.slot[s].ref = n.internal ? new_wide(binary, it.depth + 1) : leaf_ref(n);
    };
    const Node& b = tree.nodes[it.binary];
    if (!b.internal) {  // a shape whose binary root is a leaf
      set_slot(0, it.binary);
    } else {
      axes = b.axis;
      for (int s1 = 0; s1 < 2; s1++) {
        const int   child = b.start + s1;
        const Node& c     = tree.nodes[child];
        if (!c.internal) {
          set_slot(s1 << 2, child);
          continue;
        }
        axes |= (unsigned)c.axis << (2 + 2 * s1);
        for (int s2 = 0; s2 < 2; s2++) {
          const int   grand = c.start + s2;
          const Node& g     = tree.nodes[grand];
          if (!g.internal) {
            set_slot((s1 << 2) | (s2 << 1), grand);
            continue;
          }
          axes |= (unsigned)g.axis << (6 + 2 * (2 * s1 + s2));
          set_slot((s1 << 2) | (s2 << 1) | 0, g.start + 0);
          set_slot((s1 << 2) | (s2 << 1) | 1, g.start + 1);
        }
      }
    }
    for (int s = 0; s < 8; s++) w.slot[s].axes = axes;
    out[it.wide] = w;
  }
  return max_depth;
}

int collapse_wide16(const Tree& tree, std::vector<WideNode16>& out) {
  out.clear();
  const float inf = std::numeric_limits<float>::infinity();
  auto leaf_ref = [](const Node& n) { return 0xC0000000u | ((unsigned)n.num << 27) | (unsigned)n.start; };
  struct Item {
    int binary, wide, depth;
  };
  std::deque<Item> queue;
  auto new_wide = [&](int binary, int depth) {
    out.emplace_back();
    queue.push_back({binary, (int)out.size() - 1, depth});
    return (unsigned)out.size() - 1;
  };
  constexpr int LEVELS          = 4;
  const int     axes_at[LEVELS] = {0, 2, 6, 14};  // bit offset of the axes of the nodes at each level below the wide node's root
  int max_depth = 1;
  new_wide(0, 1);
  while (!queue.empty()) {
    Item it = queue.front();
    queue.pop_front();
    max_depth = std::max(max_depth, it.depth);
    WideNode16 w;
    unsigned   axes = 0;
    for (int s = 0; s < 16; s++) {
      for (int k = 0; k < 3; k++) w.slot[s].bmin[k] = inf, w.slot[s].bmax[k] = -inf;
      w.slot[s].ref = 0xFFFFFFFFu, w.slot[s].axes = 0;
    }
    // walk the four levels below the root of this wide node: (binary node, level, path = side bits taken so far)
    struct Walk {
      int binary, level;
      unsigned path;
    };
    std::vector<Walk> todo{{it.binary, 0, 0u}};
    while (!todo.empty()) {
      Walk wk = todo.back();
      todo.pop_back();
      const Node& n = tree.nodes[wk.binary];
      const bool  root_leaf = wk.level == 0 && !n.internal;  // a shape whose binary root is a leaf
      if (wk.level == LEVELS || !n.internal) {                 // becomes a slot: the first of its group
        if (wk.level == 0 && !root_leaf) continue;
        const int s = (int)(wk.path << (LEVELS - wk.level));
        for (int k = 0; k < 3; k++) w.slot[s].bmin[k] = n.bbox.min[k], w.slot[s].bmax[k] = n.bbox.max[k];
        w.slot[s].ref = n.internal ? new_wide(wk.binary, it.depth + 1) : leaf_ref(n);
        continue;
      }
      axes |= (unsigned)n.axis << (axes_at[wk.level] + 2 * (int)wk.path);
      todo.push_back({n.start + 1, wk.level + 1, (wk.path << 1) | 1u});
      todo.push_back({n.start + 0, wk.level + 1, (wk.path << 1) | 0u});
    }
    for (int s = 0; s < 16; s++) w.slot[s].axes = axes;
    out[it.wide] = w;
  }
  return max_depth;
}

}  // namespace yhh
