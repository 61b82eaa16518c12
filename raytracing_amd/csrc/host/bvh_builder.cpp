#include "bvh_builder.h"
#include <algorithm>
#include <atomic>
#include <cassert>
#include <queue>
#include <thread>
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
    std::vector<float> suf_surface[3];  // per axis, so that the three sweeps of a large node can run side by side
    SahTree *tree;
    // Threads beyond the calling one that the build may still start.  Two kinds of work are handed
    // out, both leaving the result untouched: the three axis sweeps of a large node (each sweep keeps
    // its own best split; they are then compared in axis order with the reference's strict `<`), and
    // the two subtrees of a node (disjoint index ranges, and node numbers that do not depend on who
    // builds what: a subtree over n primitives has 2n - 1 nodes, numbered in preorder from its root).
    std::atomic<int> spare_threads{0};
    static constexpr int PARALLEL_SWEEP_MIN = 1 << 17, PARALLEL_SUBTREE_MIN = 1 << 14;

    bool take_threads(int k) {
        int have = spare_threads.load(std::memory_order_relaxed);
        while(have >= k)
            if(spare_threads.compare_exchange_weak(have, have - k, std::memory_order_relaxed)) return true;
        return false;
    }
    void give_threads(int k) { spare_threads.fetch_add(k, std::memory_order_relaxed); }

    // One axis of the sweep over order[lo, lo + n): sorts by box centre and returns the cheapest
    // split of this axis that beats `limit` (cost, index), or (limit, -1).
    std::pair<float, int> sweep_axis(int d, int lo, int n, float limit) {
        KeyIdx *s = sorted[d].data() + lo;
        for(int i = 0; i < n; i++) {
            const Box &b = boxes[order[lo + i]];
            s[i].key = (b.mi[d] + b.mx[d]) * 0.5f;
            s[i].idx = order[lo + i];
        }
        std::sort(s, s + n, [](const KeyIdx &a, const KeyIdx &b) { return a.key < b.key; });
        float *suf = suf_surface[d].data() + lo;
        Box acc;
        for(int i = n - 1; i >= 0; i--) {
            acc.combine(boxes[s[i].idx]);
            suf[i] = acc.surface();
        }
        float best = limit;
        int best_i = -1;
        Box pre;
        for(int i = 0; i < n - 1; i++) {
            pre.combine(boxes[s[i].idx]);
            float c = pre.surface() * (float)(i + 1) + suf[i + 1] * (float)(n - i - 1);
            if(c < best) {
                best = c;
                best_i = i;
            }
        }
        return {best, best_i};
    }

    // Builds the subtree over order[lo, hi) into nodes[me ...] (preorder) and returns me.
    //
    // Tie order matters: std::sort is not stable, so the permutation it produces for
    // equal keys depends on the input order and on the comparator's answers only.
    // The reference sorts a copy of the node's primitive vector per axis with
    // `centre[d] < centre[d]`; sorting (key, id) pairs that start in the same order
    // with the same predicate walks libstdc++'s introsort through the same steps and
    // ends in the same permutation, at 8 bytes per element instead of 40.
    int build(int lo, int hi, int me) {
        int n = hi - lo;
        assert(n > 0);
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

        // The reference runs the axes one after the other against ONE running best with a strict `<`
        // (BVH.cpp:42-67).  Sweeping each axis against the initial cost and then taking, in axis order,
        // the first that is strictly cheaper gives the same (axis, index): within an axis the first
        // index of the axis' minimum wins either way, and an axis that does not beat its predecessors'
        // best changes nothing either way.
        const float initial = (float)n * all.surface();
        std::pair<float, int> ax[3];
        if(n >= PARALLEL_SWEEP_MIN && take_threads(2)) {
            std::thread t1([&] { ax[1] = sweep_axis(1, lo, n, initial); });
            std::thread t2([&] { ax[2] = sweep_axis(2, lo, n, initial); });
            ax[0] = sweep_axis(0, lo, n, initial);
            t1.join();
            t2.join();
            give_threads(2);
        } else {
            for(int d = 0; d < 3; d++) ax[d] = sweep_axis(d, lo, n, initial);
        }
        int bs_d = 0, bs_i = 0;
        float bs_cost = initial;
        for(int d = 0; d < 3; d++)
            if(ax[d].second >= 0 && ax[d].first < bs_cost) {
                bs_cost = ax[d].first;
                bs_d = d;
                bs_i = ax[d].second;
            }
        // left keeps the first bs_i+1 of the winning order; right receives the rest
        // back to front (the reference moves them with pop_back/push_back, BVH.cpp:71-75)
        const KeyIdx *w = sorted[bs_d].data() + lo;
        int nl = bs_i + 1;
        for(int i = 0; i < nl; i++) order[lo + i] = w[i].idx;
        for(int i = nl; i < n; i++) order[lo + i] = w[n - 1 - (i - nl)].idx;

        const int l = me + 1, r = me + 2 * nl;  // preorder: the left subtree's 2 nl - 1 nodes sit between
        if(n >= PARALLEL_SUBTREE_MIN && nl >= PARALLEL_SUBTREE_MIN / 8 && n - nl >= PARALLEL_SUBTREE_MIN / 8 && take_threads(1)) {
            std::thread tl([&] { build(lo, lo + nl, l); });
            build(lo + nl, hi, r);
            tl.join();
            give_threads(1);
        } else {
            build(lo, lo + nl, l);
            build(lo + nl, hi, r);
        }
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

void build_sah_tree(const Box *boxes, int n, SahTree &out, int n_threads) {
    assert(n > 0);
    Builder b;
    b.boxes = boxes;
    b.order.resize(n);
    for(int i = 0; i < n; i++) b.order[i] = i;
    for(int d = 0; d < 3; d++) {
        b.sorted[d].resize(n);
        b.suf_surface[d].resize(n);
    }
    out.nodes.clear();
    out.nodes.resize(2 * (size_t)n - 1);
    b.tree = &out;
    b.spare_threads = n_threads > 1 ? n_threads - 1 : 0;
    out.root = b.build(0, n, 0);
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
