#include "bvh_builder.h"
#include <algorithm>
#include <cassert>
#include <queue>
#include <utility>

namespace rta {
namespace {

// What the reference's comparator looks at: the box centre along one axis,
// (mi + mx) * 0.5f  (src/AABB.cpp:28-30, src/BVH.cpp:46-48).
struct KeyIdx {
    float key;
    int idx;
};

struct Builder {
    const Box *boxes;
    std::vector<int> order;        // primitive ids, in the order the reference's vector holds them
    std::vector<KeyIdx> sorted[3]; // per-axis scratch, sized n once
    std::vector<float> suf_surface;
    SahTree *tree;

    // Builds the subtree over order[lo, hi) and returns its node index.
    //
    // Tie order matters: std::sort is not stable, so the permutation it produces for
    // equal keys depends on the input order and on the comparator's answers only.
    // The reference sorts a copy of the node's primitive vector per axis with
    // `centre[d] < centre[d]`; sorting (key, id) pairs that start in the same order
    // with the same predicate walks libstdc++'s introsort through the same steps and
    // ends in the same permutation, at 8 bytes per element instead of 40.
    int build(int lo, int hi) {
        int n = hi - lo;
        assert(n > 0);
        int me = (int)tree->nodes.size();
        tree->nodes.emplace_back();
        if(n == 1) {
            SahNode &leaf = tree->nodes[me];
            leaf.box = boxes[order[lo]];
            leaf.prim = order[lo];
            leaf.siz = 1;
            leaf.depth = 1;
            return me;
        }
        Box all;
        for(int i = lo; i < hi; i++) all.combine(boxes[order[i]]);

        int bs_d = 0, bs_i = 0;
        float bs_cost = (float)n * all.surface();
        for(int d = 0; d < 3; d++) {
            KeyIdx *s = sorted[d].data() + lo;
            for(int i = 0; i < n; i++) {
                const Box &b = boxes[order[lo + i]];
                s[i].key = (b.mi[d] + b.mx[d]) * 0.5f;
                s[i].idx = order[lo + i];
            }
            std::sort(s, s + n, [](const KeyIdx &a, const KeyIdx &b) { return a.key < b.key; });

            float *suf = suf_surface.data() + lo;
            Box acc;
            for(int i = n - 1; i >= 0; i--) {
                acc.combine(boxes[s[i].idx]);
                suf[i] = acc.surface();
            }
            Box pre;
            for(int i = 0; i < n - 1; i++) {
                pre.combine(boxes[s[i].idx]);
                float c = pre.surface() * (float)(i + 1) + suf[i + 1] * (float)(n - i - 1);
                if(c < bs_cost) {
                    bs_cost = c;
                    bs_d = d;
                    bs_i = i;
                }
            }
        }
        // left keeps the first bs_i+1 of the winning order; right receives the rest
        // back to front (the reference moves them with pop_back/push_back, BVH.cpp:71-75)
        const KeyIdx *w = sorted[bs_d].data() + lo;
        int nl = bs_i + 1;
        for(int i = 0; i < nl; i++) order[lo + i] = w[i].idx;
        for(int i = nl; i < n; i++) order[lo + i] = w[n - 1 - (i - nl)].idx;

        int l = build(lo, lo + nl);
        int r = build(lo + nl, hi);
        SahNode &node = tree->nodes[me];
        node.box = all;
        node.left = l;
        node.right = r;
        node.siz = tree->nodes[l].siz + tree->nodes[r].siz + 1;
        node.depth = std::max(tree->nodes[l].depth, tree->nodes[r].depth) + 1;
        return me;
    }
};

BVHNodeInfo make_node(const Box &b) {
    BVHNodeInfo y;
    y.aa[0] = b.mi[0]; y.aa[1] = b.mi[1]; y.aa[2] = b.mi[2]; y.aa[3] = 0.0f;
    y.bb[0] = b.mx[0]; y.bb[1] = b.mx[1]; y.bb[2] = b.mx[2]; y.bb[3] = 0.0f;
    y.lsIndex = y.rsIndex = y.instanceIndex = y.triangleIndex = -1;
    return y;
}

}  // namespace

void build_sah_tree(const Box *boxes, int n, SahTree &out) {
    assert(n > 0);
    Builder b;
    b.boxes = boxes;
    b.order.resize(n);
    for(int i = 0; i < n; i++) b.order[i] = i;
    for(int d = 0; d < 3; d++) b.sorted[d].resize(n);
    b.suf_surface.resize(n);
    out.nodes.clear();
    out.nodes.reserve(2 * (size_t)n - 1);
    b.tree = &out;
    out.root = b.build(0, n);
}

void flatten_mesh_trees(const std::vector<SahTree> &trees, const std::vector<int> &tri_base,
                        std::vector<BVHNodeInfo> &out) {
    out.clear();
    size_t total = 0;
    for(auto &t : trees) total += t.nodes.size();
    out.reserve(total);
    // one queue seeded with every root: numbering is level order ACROSS meshes
    std::queue<std::pair<int, int>> q;  // (mesh, node)
    for(int m = 0; m < (int)trees.size(); m++) q.emplace(m, trees[m].root);
    int N = (int)trees.size();
    while(!q.empty()) {
        auto [m, k] = q.front();
        q.pop();
        const SahNode &p = trees[m].nodes[k];
        BVHNodeInfo y = make_node(p.box);
        if(p.left >= 0) { y.lsIndex = N++; q.emplace(m, p.left); }
        if(p.right >= 0) { y.rsIndex = N++; q.emplace(m, p.right); }
        if(p.prim >= 0) y.triangleIndex = tri_base[m] + p.prim;
        out.push_back(y);
    }
}

void flatten_scene_tree(const SahTree &tree, std::vector<BVHNodeInfo> &out) {
    out.clear();
    std::queue<int> q;
    q.push(tree.root);
    int N = 0;
    while(!q.empty()) {
        const SahNode &p = tree.nodes[q.front()];
        q.pop();
        BVHNodeInfo y = make_node(p.box);
        if(p.prim >= 0) {
            y.instanceIndex = p.prim;
        } else {
            if(p.left >= 0) { y.lsIndex = ++N; q.push(p.left); }
            if(p.right >= 0) { y.rsIndex = ++N; q.push(p.right); }
        }
        out.push_back(y);
    }
}

}  // namespace rta
