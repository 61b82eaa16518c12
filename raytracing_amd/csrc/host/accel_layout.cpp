#include "accel_layout.h"
#include "bvh_builder.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <functional>

namespace rta {
namespace {

void set_box(float lo[3], float hi[3], const BVHNodeInfo &n) {
    for(int k = 0; k < 3; k++) {
        lo[k] = n.aa[k];
        hi[k] = n.bb[k];
    }
}

// Children are numbered after their parent by both flatteners (breadth-first order), so
// "child index > own index" proves the arrays form a forest and every walk terminates.
bool check_children(const BVHNodeInfo *nodes, size_t n, size_t i, const char *what, std::string &err) {
    const BVHNodeInfo &p = nodes[i];
    if(p.lsIndex < 0 || p.rsIndex < 0 || (size_t)p.lsIndex >= n || (size_t)p.rsIndex >= n ||
       (size_t)p.lsIndex <= i || (size_t)p.rsIndex <= i) {
        err = std::string(what) + " BVH node " + std::to_string(i) + " has invalid child indices";
        return false;
    }
    return true;
}

// child box inside the parent's box, coordinate by coordinate (false as soon as a NaN is involved)
bool box_inside(const BVHNodeInfo &c, const BVHNodeInfo &p) {
    for(int k = 0; k < 3; k++)
        if(!(c.aa[k] >= p.aa[k]) || !(c.bb[k] <= p.bb[k])) return false;
    return true;
}

float half_area(const BVHNodeInfo &n) {
    const float x = n.bb[0] - n.aa[0], y = n.bb[1] - n.aa[1], z = n.bb[2] - n.aa[2];
    return x * y + y * z + z * x;
}

// Child boxes of one wide node on the node's 256-step grid, rounded outwards as decoded:
// fmaf(step, scale, origin) <= lo and >= hi in fp32, scale = 2^(e - 127).
void quantise_node(const uint32_t refs[4], const BVHNodeInfo *const boxes[4], QNode &q) {
    memset(&q, 0, sizeof q);
    for(int i = 0; i < 4; i++) q.c[i] = refs[i];
    for(int a = 0; a < 3; a++) {
        bool finite = true;
        float o = std::numeric_limits<float>::infinity(), top = -std::numeric_limits<float>::infinity();
        for(int i = 0; i < 4; i++)
            if(boxes[i]) {
                const float l = boxes[i]->aa[a], h = boxes[i]->bb[a];
                if(!std::isfinite(l) || !std::isfinite(h)) finite = false;
                o = std::min(o, std::min(l, h));
                top = std::max(top, std::max(l, h));
            }
        int e = 255;
        if(finite) {
            // smallest scale whose 255th step reaches the top
            int guess = 1;
            const double ext = (double)top - (double)o;
            if(ext > 0.0) {
                int ex;
                std::frexp(ext / 255.0, &ex);  // ext / 255 = m * 2^ex, m in [0.5, 1)
                guess = std::max(1, std::min(254, ex + 127 - 1));
            }
            for(e = guess; e < 255; e++) {
                const float s = std::ldexp(1.0f, e - 127);
                const float reach = std::fmaf(255.0f, s, o);
                if(std::isfinite(reach) && reach >= top) break;
            }
        }
        q.origin[a] = finite ? o : 0.0f;
        q.scale[a] = e == 255 ? std::numeric_limits<float>::infinity() : std::ldexp(1.0f, e - 127);
        if(e == 255) continue;  // every plane decodes to NaN: the axis constrains nothing
        const float s = q.scale[a];
        for(int i = 0; i < 4; i++) {
            if(!boxes[i]) continue;
            // (the slab test treats its two planes alike, so an inverted box is stored the right way round)
            const float l = std::min(boxes[i]->aa[a], boxes[i]->bb[a]), h = std::max(boxes[i]->aa[a], boxes[i]->bb[a]);
            int ql = (int)std::floor(((double)l - (double)o) / (double)s);
            ql = std::max(0, std::min(255, ql));
            while(ql > 0 && std::fmaf((float)ql, s, o) > l) ql--;
            int qh = (int)std::ceil(((double)h - (double)o) / (double)s);
            qh = std::max(0, std::min(255, qh));
            while(qh < 255 && std::fmaf((float)qh, s, o) < h) qh++;
            q.qlo[a][i] = (uint8_t)ql;
            q.qhi[a][i] = (uint8_t)qh;
        }
    }
}

// references of a wide record, whichever format
inline uint32_t *wide_refs(TNode *rec, bool quantised) {
    return quantised ? reinterpret_cast<QNode *>(rec)->c : reinterpret_cast<WNode *>(rec)->c;
}

// The wide nodes of the trees under `roots` (one group: the roots may share subtrees), numbered
// from 0 within the group; a reference to an inner child holds that number until merge time.
struct WideGroup {
    std::vector<TNode> recs;      // stride slots per wide node
    std::vector<int32_t> owner;   // reference node of wide node k
    std::vector<uint8_t> height;  // height of the wide subtree under wide node k
    bool ordered = true;          // every inner child has a larger number than its parent
    // A group cut off at a frontier: references numbered owner.size() + j stand for the root (wide node 0)
    // of the j-th of the n_pending groups that follow this one.
    size_t n_pending = 0;
};

// Which inner nodes stay (become wide nodes) and which dissolve into their parent's slots,
// chosen to minimise the summed surface of the nodes that stay (the expected number of wide
// nodes a ray visits), by dynamic programming over "subtree n may use up to i child slots of
// the wide node above it" (i = 1..3; with one slot the node stays and spreads four slots over
// its own two subtrees).  Leaves cost nothing here: a triangle child is tested against its
// reference parent's box whatever happens above it.
struct CollapsePlan {
    bool pairs = false;        // a node whose two children are leaves takes one slot (a leaf pair) and never stays
    std::vector<float> cost;   // [3 * n + (i - 1)]
    std::vector<uint8_t> dec;  // bits 0-1: slots of ls when the node stays, minus 1; bit 2: with two slots, split 1+1;
                               // bits 3-4: with three slots: 0 stay, 1 split 1+2, 2 split 2+1, 3 split 1+1; bit 7: done
};

template <class IsLeaf>
void plan_collapse(const BVHNodeInfo *nodes, const int32_t *roots, size_t n_roots, size_t n_nodes, IsLeaf is_leaf, CollapsePlan &pl) {
    std::vector<int32_t> stack;
    auto bottom = [&](int32_t n) { return pl.pairs && !is_leaf(n) && is_leaf(nodes[n].lsIndex) && is_leaf(nodes[n].rsIndex); };
    auto C = [&](int32_t n, int i) { return is_leaf(n) || bottom(n) ? 0.0f : pl.cost[3 * (size_t)n + (i - 1)]; };
    for(size_t r = 0; r < n_roots; r++) {
        if(roots[r] < 0 || (size_t)roots[r] >= n_nodes || is_leaf(roots[r])) continue;
        stack.push_back(roots[r]);
        while(!stack.empty()) {
            const int32_t n = stack.back();
            if(pl.dec[n] & 0x80u) {
                stack.pop_back();
                continue;
            }
            const int32_t ls = nodes[n].lsIndex, rs = nodes[n].rsIndex;
            const bool ls_ready = is_leaf(ls) || bottom(ls) || (pl.dec[ls] & 0x80u), rs_ready = is_leaf(rs) || bottom(rs) || (pl.dec[rs] & 0x80u);
            if(!ls_ready || !rs_ready) {
                if(!rs_ready) stack.push_back(rs);
                if(!ls_ready) stack.push_back(ls);
                continue;
            }
            stack.pop_back();
            // the node stays: four slots over its two subtrees
            int j4 = 2;
            float best = C(ls, 2) + C(rs, 2);
            for(int j : {1, 3}) {
                const float c = C(ls, j) + C(rs, 4 - j);
                if(c < best) {
                    best = c;
                    j4 = j;
                }
            }
            const float stay = half_area(nodes[n]) + best;
            uint8_t d = (uint8_t)(j4 - 1);
            float c2 = stay;
            const float s11 = C(ls, 1) + C(rs, 1);
            if(s11 < c2) {
                c2 = s11;
                d |= 4u;
            }
            float c3 = stay;
            unsigned d3 = 0;
            const float s12 = C(ls, 1) + C(rs, 2), s21 = C(ls, 2) + C(rs, 1);
            if(s11 < c3) { c3 = s11; d3 = 3; }
            if(s12 < c3) { c3 = s12; d3 = 1; }
            if(s21 < c3) { c3 = s21; d3 = 2; }
            pl.cost[3 * (size_t)n] = stay;
            pl.cost[3 * (size_t)n + 1] = c2;
            pl.cost[3 * (size_t)n + 2] = c3;
            pl.dec[n] = (uint8_t)(d | (d3 << 3) | 0x80u);
        }
    }
}

void wide_group_heights(WideGroup &g, uint32_t inner_kind, bool quantised, const uint8_t *pending_height);

template <class IsLeaf, class LeafRef>
void build_wide_group(const BVHNodeInfo *nodes, size_t n_nodes, const int32_t *roots, size_t n_roots, IsLeaf is_leaf,
                      LeafRef leaf_ref, bool leaf_box_of_parent, uint32_t inner_kind, bool quantised,
                      std::vector<uint32_t> &wide_of, CollapsePlan &plan, WideGroup &g, size_t frontier = 0,
                      std::vector<int32_t> *pending = nullptr) {
    // frontier > 0: wide nodes are made breadth first, and only until that many wait to be made; those are
    // handed back in `pending` (their subtrees become groups of their own, built side by side) instead.
    const uint32_t stride = quantised ? 1u : 2u;  // slots per wide node
    plan_collapse(nodes, roots, n_roots, n_nodes, is_leaf, plan);
    std::vector<int32_t> work;
    size_t head = 0;  // breadth first: work[head ...] is the queue
    auto wide_index = [&](int32_t node) {  // number of the node's wide node within the group (created on first use)
        if(wide_of[node] == REF_NONE) {
            wide_of[node] = (uint32_t)g.owner.size();
            g.recs.resize(g.recs.size() + stride);
            g.owner.push_back(node);
            work.push_back(node);
        }
        return wide_of[node];
    };
    for(size_t r = 0; r < n_roots; r++)
        if(roots[r] >= 0 && (size_t)roots[r] < n_nodes && !is_leaf(roots[r])) wide_index(roots[r]);
    while(frontier ? head < work.size() && work.size() - head < frontier : !work.empty()) {
        int32_t n;
        if(frontier) {
            n = work[head++];
        } else {
            n = work.back();
            work.pop_back();
        }
        struct Item { int32_t node, parent; bool pair; } items[4];  // pair: `node` is an inner node, its two leaves share the slot
        int cnt = 0;
        auto bottom = [&](int32_t c) { return plan.pairs && !is_leaf(c) && is_leaf(nodes[c].lsIndex) && is_leaf(nodes[c].rsIndex); };
        {
            // hand the four slots down as planned
            struct Todo { int32_t node, parent; int slots; } todo[8];
            int nt = 0;
            const int j4 = (plan.dec[n] & 3u) + 1;
            todo[nt++] = {nodes[n].rsIndex, n, 4 - j4};
            todo[nt++] = {nodes[n].lsIndex, n, j4};
            while(nt > 0) {
                const Todo t = todo[--nt];
                int a = 0, b = 0;  // slots of ls / rs when the node dissolves
                if(bottom(t.node)) {
                    items[cnt++] = {t.node, t.node, true};
                    continue;
                }
                if(!is_leaf(t.node) && t.slots >= 2) {
                    const uint8_t d = plan.dec[t.node];
                    if(t.slots == 2) {
                        if(d & 4u) a = b = 1;
                    } else {
                        const unsigned d3 = (d >> 3) & 3u;
                        if(d3 == 1) { a = 1; b = 2; }
                        else if(d3 == 2) { a = 2; b = 1; }
                        else if(d3 == 3) { a = 1; b = 1; }
                    }
                }
                if(a == 0) {
                    items[cnt++] = {t.node, t.parent, false};
                } else {
                    todo[nt++] = {nodes[t.node].rsIndex, t.node, b};
                    todo[nt++] = {nodes[t.node].lsIndex, t.node, a};
                }
            }
        }
        const size_t k = wide_of[n];
        uint32_t refs[4];
        const BVHNodeInfo *boxes[4];
        for(int i = 0; i < 4; i++) {
            refs[i] = REF_NONE;
            boxes[i] = nullptr;
            if(i >= cnt) continue;
            const int32_t c = items[i].node;
            if(items[i].pair) {
                refs[i] = leaf_ref(nodes[c].lsIndex) | REF_PAIR;
                boxes[i] = &nodes[c];
            } else if(is_leaf(c)) {
                refs[i] = leaf_ref(c);
                boxes[i] = &nodes[leaf_box_of_parent ? items[i].parent : c];
            } else {
                const uint32_t ck = wide_index(c);  // (may grow g.recs: indexed again below)
                if(ck <= k) g.ordered = false;
                refs[i] = make_ref(inner_kind, ck);
                boxes[i] = &nodes[c];
            }
        }
        if(quantised) {
            quantise_node(refs, boxes, *reinterpret_cast<QNode *>(&g.recs[k]));
        } else {
            WNode &w = *reinterpret_cast<WNode *>(&g.recs[2 * k]);
            memset(&w, 0, sizeof w);
            float *lo[4] = {w.lo0, w.lo1, w.lo2, w.lo3}, *hi[4] = {w.hi0, w.hi1, w.hi2, w.hi3};
            for(int i = 0; i < 4; i++) {
                w.c[i] = refs[i];
                for(int a = 0; a < 3; a++) {
                    lo[i][a] = boxes[i] ? boxes[i]->aa[a] : 0.0f;
                    hi[i][a] = boxes[i] ? boxes[i]->bb[a] : 0.0f;
                }
            }
        }
    }
    if(frontier) {
        // what still waits was numbered last, in queue order: take those numbers back
        const size_t n_wait = work.size() - head, n_top = g.owner.size() - n_wait;
        for(size_t j = 0; j < n_wait; j++) {
            const int32_t node = work[head + j];
            if(wide_of[node] != n_top + j) abort();  // (numbering is by creation, creation is by queue order)
            wide_of[node] = REF_NONE;
            pending->push_back(node);
        }
        g.owner.resize(n_top);
        g.recs.resize(n_top * stride);
        g.n_pending = n_wait;
        return;  // heights: once the groups below are known
    }
    wide_group_heights(g, inner_kind, quantised, nullptr);
}

// Heights of a group's wide subtrees, children before parents: in creation order a child comes after
// its parent unless subtrees are shared; then order by the reference index (a child's is larger than
// its parent's).  pending_height[j]: height of the j-th group cut off below this one.
void wide_group_heights(WideGroup &g, uint32_t inner_kind, bool quantised, const uint8_t *pending_height) {
    const uint32_t stride = quantised ? 1u : 2u;
    const size_t n_wide = g.owner.size();
    g.height.assign(n_wide, 1);
    std::vector<uint32_t> order;
    if(!g.ordered) {
        order.resize(n_wide);
        for(size_t k = 0; k < n_wide; k++) order[k] = (uint32_t)k;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return g.owner[x] < g.owner[y]; });
    }
    for(size_t j = n_wide; j-- > 0;) {
        const size_t k = g.ordered ? j : order[j];
        const uint32_t *c = wide_refs(&g.recs[k * stride], quantised);
        int h = 1;
        for(int i = 0; i < 4; i++)
            if(c[i] != REF_NONE && (c[i] >> 30) == inner_kind) {
                const size_t ck = c[i] & REF_INDEX_MASK;
                h = std::max<int>(h, 1 + (ck < n_wide ? g.height[ck] : pending_height[ck - n_wide]));
            }
        g.height[k] = (uint8_t)std::min(h, 255);
    }
}

// Runs fn(0) .. fn(n - 1) on up to `threads` threads, items handed out one at a time.
template <class Fn>
void parallel_items(size_t n, unsigned threads, Fn fn) {
    if(threads <= 1 || n <= 1) {
        for(size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    auto body = [&]() {
        for(size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
    };
    for(unsigned t = 1; t < threads && t < n; t++) pool.emplace_back(body);
    body();
    for(std::thread &t : pool) t.join();
}

unsigned layout_threads() {
    const unsigned hc = std::thread::hardware_concurrency();
    return hc == 0 ? 1u : (hc > 16u ? 16u : hc);
}

// Collapses binary trees (child index > parent index) into four-wide nodes.  A node takes its
// two children and then, while it has room, replaces the inner child of largest surface by
// that child's two children.  is_leaf(i) tells leaves; leaf_ref(i) makes their references; a
// leaf child carries its reference parent's box when leaf_box_of_parent, else its own.  Wide
// nodes are appended to `slots` from slot_base on (which is returned past them).  Trees that
// share nothing (`disjoint`) are built side by side on several threads.
template <class IsLeaf, class LeafRef>
uint32_t build_wide_nodes(const BVHNodeInfo *nodes, size_t n_nodes, const std::vector<int32_t> &roots, bool disjoint, bool pairs,
                          IsLeaf is_leaf, LeafRef leaf_ref, bool leaf_box_of_parent, uint32_t inner_kind, uint32_t slot_base,
                          bool quantised, std::vector<TNode> &slots, std::vector<uint32_t> &wide_of,
                          std::vector<uint8_t> &wide_height) {
    const uint32_t stride = quantised ? 1u : 2u;
    wide_of.assign(n_nodes, REF_NONE);
    CollapsePlan plan;
    plan.pairs = pairs;
    plan.cost.resize(3 * n_nodes);
    plan.dec.assign(n_nodes, 0);
    // Trees that share nothing are built side by side.  A tree of BIG_TREE nodes or more is cut as well: its top
    // is made breadth first until FRONTIER wide nodes wait, and the subtrees under those become groups of their own
    // (the cost plan, bottom-up, is first made for the subtrees under a frontier of the binary tree in the same way).
    // Where the cuts fall depends on the trees alone, not on the number of threads: the layout is the same on any machine.
    constexpr size_t BIG_TREE = (size_t)1 << 17, FRONTIER = 256;
    std::vector<WideGroup> groups(1);
    if(disjoint) {
        const unsigned threads = layout_threads();
        std::vector<uint8_t> big(roots.size(), 0);
        if(n_nodes >= BIG_TREE) {
            std::vector<uint32_t> size(n_nodes, 1);  // (children come after their parents)
            for(size_t i = n_nodes; i-- > 0;)
                if(!is_leaf((int32_t)i)) size[i] = 1 + size[nodes[i].lsIndex] + size[nodes[i].rsIndex];
            std::vector<int32_t> plan_roots;
            for(size_t r = 0; r < roots.size(); r++) {
                if(size[roots[r]] < BIG_TREE) continue;
                big[r] = 1;
                // the FRONTIER largest subtrees found by always opening the largest
                std::vector<int32_t> open{roots[r]};
                auto smaller = [&](int32_t a, int32_t b) { return size[a] != size[b] ? size[a] < size[b] : a > b; };
                while(open.size() < FRONTIER && size[open.front()] > 64) {
                    std::pop_heap(open.begin(), open.end(), smaller);
                    const int32_t n = open.back();
                    open.pop_back();
                    for(int32_t c : {nodes[n].lsIndex, nodes[n].rsIndex}) {
                        open.push_back(c);
                        std::push_heap(open.begin(), open.end(), smaller);
                    }
                }
                plan_roots.insert(plan_roots.end(), open.begin(), open.end());
            }
            parallel_items(plan_roots.size(), threads, [&](size_t i) { plan_collapse(nodes, &plan_roots[i], 1, n_nodes, is_leaf, plan); });
        }
        // tops (whole trees where small), then what was cut off below them
        std::vector<WideGroup> tops(roots.size());
        std::vector<std::vector<int32_t>> pending(roots.size());
        parallel_items(roots.size(), threads, [&](size_t r) {
            build_wide_group(nodes, n_nodes, &roots[r], 1, is_leaf, leaf_ref, leaf_box_of_parent, inner_kind, quantised, wide_of, plan, tops[r],
                             big[r] ? FRONTIER : 0, &pending[r]);
        });
        std::vector<size_t> top_at(roots.size());
        std::vector<int32_t> sub_root;
        std::vector<size_t> sub_at;
        size_t n_groups = 0;
        for(size_t r = 0; r < roots.size(); r++) {
            top_at[r] = n_groups++;
            for(int32_t node : pending[r]) {
                sub_root.push_back(node);
                sub_at.push_back(n_groups++);
            }
        }
        groups.resize(n_groups);
        for(size_t r = 0; r < roots.size(); r++) groups[top_at[r]] = std::move(tops[r]);
        parallel_items(sub_root.size(), threads, [&](size_t i) {
            build_wide_group(nodes, n_nodes, &sub_root[i], 1, is_leaf, leaf_ref, leaf_box_of_parent, inner_kind, quantised, wide_of, plan,
                             groups[sub_at[i]]);
        });
        for(size_t r = 0; r < roots.size(); r++) {
            if(!big[r]) continue;
            WideGroup &g = groups[top_at[r]];
            std::vector<uint8_t> below(g.n_pending);
            for(size_t j = 0; j < g.n_pending; j++) below[j] = groups[top_at[r] + 1 + j].height[0];
            wide_group_heights(g, inner_kind, quantised, below.data());
        }
    } else {
        build_wide_group(nodes, n_nodes, roots.data(), roots.size(), is_leaf, leaf_ref, leaf_box_of_parent, inner_kind, quantised,
                         wide_of, plan, groups[0]);
    }
    // merge: group g's wide node k becomes slot slot_base + stride * (first[g] + k)
    std::vector<size_t> first(groups.size() + 1, 0);
    for(size_t g = 0; g < groups.size(); g++) first[g + 1] = first[g] + groups[g].owner.size();
    const size_t n_wide = first.back();
    if((uint64_t)slot_base + (uint64_t)stride * n_wide >= REF_INDEX_MASK) return REF_NONE;
    slots.resize((size_t)slot_base + stride * n_wide);
    wide_height.resize(n_wide);
    parallel_items(groups.size(), layout_threads(), [&](size_t gi) {
        WideGroup &g = groups[gi];
        const uint32_t base = slot_base + stride * (uint32_t)first[gi];
        const size_t n_own = g.owner.size();
        for(size_t k = 0; k < n_own; k++) {
            uint32_t *c = wide_refs(&g.recs[k * stride], quantised);
            for(int i = 0; i < 4; i++)
                if(c[i] != REF_NONE && (c[i] >> 30) == inner_kind) {
                    const size_t ck = c[i] & REF_INDEX_MASK;
                    // (a number past the group's own nodes: the root of one of the groups cut off below it)
                    c[i] = make_ref(inner_kind, ck < n_own ? base + stride * (uint32_t)ck : slot_base + stride * (uint32_t)first[gi + 1 + (ck - n_own)]);
                }
            wide_of[g.owner[k]] = base + stride * (uint32_t)k;
            wide_height[first[gi] + k] = g.height[k];
        }
        if(!g.recs.empty()) memcpy(&slots[base], g.recs.data(), g.recs.size() * sizeof(TNode));
        std::vector<TNode>().swap(g.recs);
    });
    return slot_base + stride * (uint32_t)n_wide;
}

}  // namespace

bool build_accel_mesh_level(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const TriangleRec *tris, size_t n_tris,
                            const std::vector<int32_t> &extra_roots, int quantise, AccelMeshLevel &out, std::string &err) {
    if(n_mesh_nodes == 0 || n_tris == 0) {
        err = "scene buffers must not be empty (mesh BVH, triangles)";
        return false;
    }
    if(n_mesh_nodes >= REF_SLOT_MASK / 2 || n_tris >= REF_INDEX_MASK) {
        err = "scene too large for 29-bit mesh slots";
        return false;
    }
    out.n_tris = n_tris;
    out.nested = true;
    const bool timing = getenv("RTA_LAYOUT_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if(!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[layout] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    // ---- validation; heights (children come after their parents, so backwards)
    std::vector<uint8_t> has_parent(n_mesh_nodes, 0);
    bool shared = false;  // some node has two parents: the arrays form a DAG, not a forest
    out.height.assign(n_mesh_nodes, 1);
    for(size_t i = n_mesh_nodes; i-- > 0;) {
        if(mesh[i].triangleIndex >= 0) {
            if((size_t)mesh[i].triangleIndex >= n_tris) {
                err = "mesh BVH node " + std::to_string(i) + " references a triangle out of range";
                return false;
            }
        } else {
            if(!check_children(mesh, n_mesh_nodes, i, "mesh", err)) return false;
            const int32_t ls = mesh[i].lsIndex, rs = mesh[i].rsIndex;
            if(has_parent[ls] || has_parent[rs] || ls == rs) shared = true;
            has_parent[ls] = 1;
            has_parent[rs] = 1;
            out.height[i] = (uint8_t)std::min(std::max<int>(out.height[ls], out.height[rs]) + 1, 255);
        }
    }
    lap("validate");
    // Slots: every reference node gets one 64-byte slot, inner nodes a TNode, leaves a TLeaf, in
    // depth-first order of each tree (node, subtree of ls, subtree of rs): a walk down a tree
    // moves through a shrinking address range, and a node often shares its 128-byte line with the
    // child visited next.  (The reference numbers nodes breadth-first across all meshes.)
    std::vector<uint32_t> &slot_of = out.slot_of;
    slot_of.assign(n_mesh_nodes, 0xFFFFFFFFu);
    uint32_t next = 0;
    std::vector<int32_t> roots;
    {
        std::vector<int32_t> stack;
        for(size_t r = 0; r < n_mesh_nodes; r++) {
            if(has_parent[r]) continue;
            roots.push_back((int32_t)r);
            stack.push_back((int32_t)r);
            while(!stack.empty()) {
                const int32_t n = stack.back();
                stack.pop_back();
                if(slot_of[n] != 0xFFFFFFFFu) continue;  // reached twice: keep the first slot
                slot_of[n] = next++;
                if(mesh[n].triangleIndex < 0) {
                    stack.push_back(mesh[n].rsIndex);
                    stack.push_back(mesh[n].lsIndex);
                }
            }
        }
    }
    lap("depth-first slots");
    // ---- the same trees, four-wide, for the ordered traversal (sizes the buffer: first)
    const bool disjoint = !shared && extra_roots.empty();
    roots.insert(roots.end(), extra_roots.begin(), extra_roots.end());
    out.quantised = quantise != 0;
    out.wide_stride = out.quantised ? 1u : 2u;
    out.wide_base = (next + 1u) & ~1u;
    out.slots.clear();
    const uint32_t end = build_wide_nodes(
        mesh, n_mesh_nodes, roots, disjoint, /* leaf pairs need the two leaves in adjacent slots: */ disjoint,
        [&](int32_t c) { return mesh[c].triangleIndex >= 0; },
        [&](int32_t c) { return make_ref(REF_TRIANGLE, slot_of[c]); }, true, REF_MESH_INNER, out.wide_base, out.quantised, out.slots,
        out.wide_of, out.wide_height);
    if(end == REF_NONE || end > REF_SLOT_MASK) {
        err = "scene too large for 29-bit mesh slots";
        return false;
    }
    {
        std::vector<int32_t> ids = roots;
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        out.root_ids.clear();
        out.root_nodes.clear();
        for(int32_t r : ids)
            if(r >= 0 && (size_t)r < n_mesh_nodes) {
                out.root_ids.push_back(r);
                out.root_nodes.push_back(mesh[r]);
            }
    }
    lap("wide nodes");
    // ---- binary nodes and leaves
    auto mesh_ref = [&](int32_t c) { return make_ref(mesh[c].triangleIndex >= 0 ? REF_TRIANGLE : REF_MESH_INNER, slot_of[c]); };
    const float inf = std::numeric_limits<float>::infinity();
    auto fill_leaf = [&](int32_t node, const float *plo, const float *phi) {
        TLeaf &l = *reinterpret_cast<TLeaf *>(&out.slots[slot_of[node]]);
        const float *v = tris[mesh[node].triangleIndex].ver;
        for(int k = 0; k < 3; k++) {
            l.parent_lo[k] = plo ? plo[k] : -inf;
            l.parent_hi[k] = phi ? phi[k] : inf;
            l.v0[k] = v[k];
            l.e1[k] = v[3 + k] - v[k];
            l.e2[k] = v[6 + k] - v[k];
        }
        l.triangle = (uint32_t)mesh[node].triangleIndex;
    };
    // (every node writes its own slot and the slots of its leaf children: ranges of nodes are
    // independent unless subtrees are shared)
    const size_t n_chunks = shared ? 1 : std::min<size_t>(256, (n_mesh_nodes + 65535) / 65536);
    std::vector<uint8_t> chunk_nested(n_chunks, 1);
    parallel_items(n_chunks, layout_threads(), [&](size_t ch) {
        const size_t lo = n_mesh_nodes * ch / n_chunks, hi = n_mesh_nodes * (ch + 1) / n_chunks;
        for(size_t ii = hi; ii-- > lo;) {
            const BVHNodeInfo &p = mesh[ii];
            if(p.triangleIndex >= 0) {
                if(!has_parent[ii]) fill_leaf((int32_t)ii, nullptr, nullptr);  // a one-triangle mesh
                continue;
            }
            if(!box_inside(mesh[p.lsIndex], p) || !box_inside(mesh[p.rsIndex], p)) chunk_nested[ch] = 0;
            TNode &t = out.slots[slot_of[ii]];
            set_box(t.lo0, t.hi0, mesh[p.lsIndex]);
            set_box(t.lo1, t.hi1, mesh[p.rsIndex]);
            t.c0 = mesh_ref(p.lsIndex);
            t.c1 = mesh_ref(p.rsIndex);
            // a leaf carries the exact box of its parent in the reference tree
            for(int32_t c : {p.lsIndex, p.rsIndex})
                if(mesh[c].triangleIndex >= 0) fill_leaf(c, p.aa, p.bb);
        }
    });
    for(uint8_t n : chunk_nested)
        if(!n) out.nested = false;
    lap("binary nodes and leaves");
    return true;
}

bool build_accel_scene_level(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const AccelMeshLevel &ml, const BVHNodeInfo *scene,
                             size_t n_scene_nodes, const InstanceInfo *inst, size_t n_inst, AccelSceneLevel &out,
                             std::string &err, bool *missing_root) {
    if(missing_root) *missing_root = false;
    if(n_scene_nodes == 0 || n_inst == 0) {
        err = "scene buffers must not be empty (scene BVH, instances)";
        return false;
    }
    if(n_inst >= REF_INDEX_MASK || n_scene_nodes >= REF_INDEX_MASK) {
        err = "scene too large for 30-bit references";
        return false;
    }
    if(ml.slot_of.size() != n_mesh_nodes) {
        err = "mesh level was built for another mesh BVH";
        return false;
    }
    auto node_at = [&](int32_t c) -> const BVHNodeInfo * {
        if(mesh) return &mesh[c];
        auto it = std::lower_bound(ml.root_ids.begin(), ml.root_ids.end(), c);
        return it != ml.root_ids.end() && *it == c ? &ml.root_nodes[it - ml.root_ids.begin()] : nullptr;
    };
    // ---- instances
    out.instances.resize(n_inst);
    int max_mesh_height = 1, max_wide_height = 0;
    for(size_t i = 0; i < n_inst; i++) {
        const InstanceInfo &s = inst[i];
        if(s.meshIndex < 0 || (size_t)s.meshIndex >= n_mesh_nodes) {
            err = "instance " + std::to_string(i) + " has a mesh root index out of range";
            return false;
        }
        if(s.materialPtr < 0) {
            err = "instance " + std::to_string(i) + " has a negative material pointer";
            return false;
        }
        InstAccel &a = out.instances[i];
        for(int r = 0; r < 3; r++)
            for(int c = 0; c < 4; c++) a.row[r][c] = s.world2local[c * 4 + r];
        const BVHNodeInfo *root_p = node_at(s.meshIndex);
        if(!root_p) {  // (only without the mesh array: not a node the mesh level kept)
            if(missing_root) *missing_root = true;
            err = "instance " + std::to_string(i) + " starts at a node that is not a mesh root";
            return false;
        }
        const BVHNodeInfo &root = *root_p;
        set_box(a.root_lo, a.root_hi, root);
        a.root_ref = make_ref(root.triangleIndex >= 0 ? REF_TRIANGLE : REF_MESH_INNER, ml.slot_of[s.meshIndex]);
        if(root.triangleIndex >= 0) {
            a.wide_root_ref = a.root_ref;
        } else {
            if(ml.wide_of[s.meshIndex] == REF_NONE) {
                if(missing_root) *missing_root = true;
                err = "instance " + std::to_string(i) + " starts at a node that is not a mesh root";
                return false;
            }
            a.wide_root_ref = make_ref(REF_MESH_INNER, ml.wide_of[s.meshIndex]);
            max_wide_height = std::max<int>(max_wide_height, ml.wide_height[(ml.wide_of[s.meshIndex] - ml.wide_base) / ml.wide_stride]);
        }
        a.material_ptr = s.materialPtr;
        for(int k = 0; k < 3; k++) {
            a.world_lo[k] = 0.0f;
            a.world_hi[k] = 0.0f;
        }
        a.material_type = 0;  // (filled in by the pass, which sees the material buffer)
        memset(a.pad1, 0, sizeof a.pad1);
        max_mesh_height = std::max<int>(max_mesh_height, ml.height[s.meshIndex]);
    }
    // ---- scene level
    std::vector<uint32_t> sinner_of(n_scene_nodes, 0);
    uint32_t n_sinner = 0;
    for(size_t i = 0; i < n_scene_nodes; i++) {
        if(scene[i].instanceIndex >= 0) {
            if((size_t)scene[i].instanceIndex >= n_inst) {
                err = "scene BVH node " + std::to_string(i) + " references an instance out of range";
                return false;
            }
        } else {
            if(!check_children(scene, n_scene_nodes, i, "scene", err)) return false;
            sinner_of[i] = n_sinner++;
        }
    }
    const uint32_t inst_base = (std::max<uint32_t>(n_sinner, 1u) + 1u) & ~1u;  // even: the records behind are 128-byte aligned
    if((uint64_t)inst_base + 2 * (uint64_t)n_inst + 2 * (uint64_t)n_sinner >= REF_INDEX_MASK) {
        err = "scene too large for 30-bit references";
        return false;
    }
    auto scene_ref = [&](int32_t c) {
        return scene[c].instanceIndex >= 0 ? make_ref(REF_INSTANCE, inst_base + 2u * (uint32_t)scene[c].instanceIndex)
                                           : make_ref(REF_SCENE_INNER, sinner_of[c]);
    };
    out.nested = true;
    out.scene_nodes.assign(inst_base, TNode());
    memset(out.scene_nodes.data(), 0, out.scene_nodes.size() * sizeof(TNode));
    std::vector<int> sheight(n_scene_nodes, 1);
    for(size_t ii = n_scene_nodes; ii-- > 0;) {
        const BVHNodeInfo &p = scene[ii];
        if(p.instanceIndex >= 0) continue;
        if(!box_inside(scene[p.lsIndex], p) || !box_inside(scene[p.rsIndex], p)) out.nested = false;
        TNode &t = out.scene_nodes[sinner_of[ii]];
        set_box(t.lo0, t.hi0, scene[p.lsIndex]);
        set_box(t.lo1, t.hi1, scene[p.rsIndex]);
        t.c0 = scene_ref(p.lsIndex);
        t.c1 = scene_ref(p.rsIndex);
        sheight[ii] = std::max(sheight[p.lsIndex], sheight[p.rsIndex]) + 1;
    }
    for(size_t i = 0; i < n_scene_nodes; i++)
        if(scene[i].instanceIndex >= 0) {
            InstAccel &a = out.instances[scene[i].instanceIndex];
            set_box(a.world_lo, a.world_hi, scene[i]);
        }
    set_box(out.header.root_lo, out.header.root_hi, scene[0]);
    out.header.extent = 0.0f;
    for(int k = 0; k < 3; k++) out.header.extent += std::fabs(scene[0].aa[k]) > std::fabs(scene[0].bb[k]) ? std::fabs(scene[0].aa[k]) : std::fabs(scene[0].bb[k]);
    out.header.root_ref = scene_ref(0);
    out.header.n_instances = (uint32_t)n_inst;
    out.header.inst_base = inst_base;
    // ---- the scene tree four-wide (an instance child keeps its own box: the scene leaf box)
    std::vector<uint32_t> swide_of;
    std::vector<uint8_t> swide_height;
    {
        std::vector<TNode> all;  // sized by the builder: [0, first wide slot) stays unused here
        const uint32_t wide_first = inst_base + 2u * (uint32_t)n_inst;
        const uint32_t end = build_wide_nodes(
            scene, n_scene_nodes, std::vector<int32_t>{0}, false, false, [&](int32_t c) { return scene[c].instanceIndex >= 0; },
            [&](int32_t c) { return scene_ref(c); }, false, REF_SCENE_INNER, wide_first, ml.quantised, all, swide_of, swide_height);
        if(end == REF_NONE) {
            err = "scene too large for 30-bit references";
            return false;
        }
        out.wide_nodes.assign(all.begin() + wide_first, all.end());
    }
    out.header.quantised = ml.quantised ? 1u : 0u;
    out.header.wide_root_ref = scene[0].instanceIndex >= 0 ? out.header.root_ref : make_ref(REF_SCENE_INNER, swide_of[0]);
    // reference order: each level of a walk defers at most one sibling; ordered walk: three per wide level
    const int wide_stack = 3 * ((swide_height.empty() ? 0 : swide_height[0]) + max_wide_height);
    out.max_stack = std::max(sheight[0] + max_mesh_height, wide_stack);
    return true;
}

bool build_accel_flat(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const TriangleRec *tris, size_t n_tris,
                      const BVHNodeInfo *scene, size_t n_scene_nodes, const InstanceInfo *inst, size_t n_inst,
                      size_t max_leaves, bool only_axis_aligned, int n_threads, AccelFlat &out, std::string &why_not) {
    out = AccelFlat();
    // ---- leaves under every distinct mesh root (counted first: instancing can multiply them beyond any memory)
    std::vector<int32_t> root_ids;
    for(size_t i = 0; i < n_inst; i++) root_ids.push_back(inst[i].meshIndex);
    std::sort(root_ids.begin(), root_ids.end());
    root_ids.erase(std::unique(root_ids.begin(), root_ids.end()), root_ids.end());
    struct MeshLeaf { int32_t node, parent; };
    std::vector<std::vector<MeshLeaf>> leaves_of(root_ids.size());
    {
        size_t visited = 0;
        std::vector<MeshLeaf> stack;
        for(size_t r = 0; r < root_ids.size(); r++) {
            stack.push_back({root_ids[r], -1});
            while(!stack.empty()) {
                const MeshLeaf e = stack.back();
                stack.pop_back();
                if(++visited > 4 * n_mesh_nodes + 16) {  // (shared subtrees walked over and over: not a forest worth flattening)
                    why_not = "the mesh trees share subtrees";
                    return false;
                }
                if(mesh[e.node].triangleIndex >= 0) {
                    leaves_of[r].push_back(e);
                } else {
                    stack.push_back({mesh[e.node].rsIndex, e.node});
                    stack.push_back({mesh[e.node].lsIndex, e.node});
                }
            }
        }
    }
    size_t n = 0;
    std::vector<size_t> first(n_inst + 1, 0);
    std::vector<uint32_t> root_of(n_inst);
    for(size_t i = 0; i < n_inst; i++) {
        root_of[i] = (uint32_t)(std::lower_bound(root_ids.begin(), root_ids.end(), inst[i].meshIndex) - root_ids.begin());
        n += leaves_of[root_of[i]].size();
        first[i + 1] = n;
        if(n > max_leaves) {
            why_not = "more than " + std::to_string(max_leaves) + " triangles once every instance is counted";
            return false;
        }
    }
    if(n == 0 || 3 * (uint64_t)n >= REF_SLOT_MASK) {
        why_not = "too many leaves for 29-bit slots";
        return false;
    }
    // ---- per instance: A = inverse of the affine world-to-local matrix (double), cond = || |A| |W| ||
    struct Affine { double a[3][4]; };
    std::vector<Affine> to_world(n_inst);
    double cond_max = 1.0;
    for(size_t i = 0; i < n_inst; i++) {
        const float *m = inst[i].world2local;  // column major: W[r][c] = m[c * 4 + r]
        double w[3][4];
        for(int r = 0; r < 3; r++)
            for(int c = 0; c < 4; c++) {
                w[r][c] = m[c * 4 + r];
                if(!std::isfinite(w[r][c])) {
                    why_not = "instance " + std::to_string(i) + ": matrix is not finite";
                    return false;
                }
            }
        if(only_axis_aligned)
            for(int r = 0; r < 3; r++) {
                double big = 0.0, sum = 0.0;
                for(int c = 0; c < 3; c++) {
                    big = std::max(big, std::fabs(w[r][c]));
                    sum += std::fabs(w[r][c]);
                }
                if(sum - big > 1e-3 * big) {
                    why_not = "instance " + std::to_string(i) + " is rotated off the axes: the world boxes of its triangles' parent boxes would be loose";
                    return false;
                }
            }
        const double det = w[0][0] * (w[1][1] * w[2][2] - w[1][2] * w[2][1]) - w[0][1] * (w[1][0] * w[2][2] - w[1][2] * w[2][0]) +
                           w[0][2] * (w[1][0] * w[2][1] - w[1][1] * w[2][0]);
        if(!(std::fabs(det) > 0.0) || !std::isfinite(1.0 / det)) {
            why_not = "instance " + std::to_string(i) + ": matrix is singular";
            return false;
        }
        Affine &A = to_world[i];
        const double id = 1.0 / det;
        A.a[0][0] = (w[1][1] * w[2][2] - w[1][2] * w[2][1]) * id;
        A.a[0][1] = (w[0][2] * w[2][1] - w[0][1] * w[2][2]) * id;
        A.a[0][2] = (w[0][1] * w[1][2] - w[0][2] * w[1][1]) * id;
        A.a[1][0] = (w[1][2] * w[2][0] - w[1][0] * w[2][2]) * id;
        A.a[1][1] = (w[0][0] * w[2][2] - w[0][2] * w[2][0]) * id;
        A.a[1][2] = (w[0][2] * w[1][0] - w[0][0] * w[1][2]) * id;
        A.a[2][0] = (w[1][0] * w[2][1] - w[1][1] * w[2][0]) * id;
        A.a[2][1] = (w[0][1] * w[2][0] - w[0][0] * w[2][1]) * id;
        A.a[2][2] = (w[0][0] * w[1][1] - w[0][1] * w[1][0]) * id;
        for(int r = 0; r < 3; r++) A.a[r][3] = -(A.a[r][0] * w[0][3] + A.a[r][1] * w[1][3] + A.a[r][2] * w[2][3]);
        double cond = 0.0;
        for(int r = 0; r < 3; r++) {
            double row = 0.0;
            for(int c = 0; c < 3; c++)
                for(int k = 0; k < 3; k++) row += std::fabs(A.a[r][k]) * std::fabs(w[k][c]);
            cond = std::max(cond, row);
        }
        if(!(cond <= 256.0)) {
            why_not = "instance " + std::to_string(i) + ": matrix is badly conditioned";
            return false;
        }
        cond_max = std::max(cond_max, cond);
    }
    // scene leaf boxes (what an instance whose mesh is one triangle is bounded by)
    std::vector<const BVHNodeInfo *> scene_leaf(n_inst, nullptr);
    for(size_t k = 0; k < n_scene_nodes; k++)
        if(scene[k].instanceIndex >= 0 && (size_t)scene[k].instanceIndex < n_inst) scene_leaf[scene[k].instanceIndex] = &scene[k];
    // ---- world boxes of the leaves: the eight corners of the reference parent box under A ...
    std::vector<Box> boxes(n);
    std::vector<uint8_t> bad(n_inst, 0);
    const unsigned threads = n_threads > 0 ? (unsigned)n_threads : layout_threads();
    parallel_items(n_inst, threads, [&](size_t i) {
        const std::vector<MeshLeaf> &lv = leaves_of[root_of[i]];
        const Affine &A = to_world[i];
        for(size_t j = 0; j < lv.size(); j++) {
            Box &b = boxes[first[i] + j];
            if(lv[j].parent < 0) {
                if(!scene_leaf[i]) { bad[i] = 1; return; }
                for(int k = 0; k < 3; k++) { b.mi[k] = scene_leaf[i]->aa[k]; b.mx[k] = scene_leaf[i]->bb[k]; }
            } else {
                const BVHNodeInfo &pn = mesh[lv[j].parent];
                double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                for(int corner = 0; corner < 8; corner++) {
                    const double x = corner & 1 ? pn.bb[0] : pn.aa[0], y = corner & 2 ? pn.bb[1] : pn.aa[1], z = corner & 4 ? pn.bb[2] : pn.aa[2];
                    for(int r = 0; r < 3; r++) {
                        const double v = A.a[r][0] * x + A.a[r][1] * y + A.a[r][2] * z + A.a[r][3];
                        lo[r] = std::min(lo[r], v);
                        hi[r] = std::max(hi[r], v);
                    }
                }
                // ... cut down to the box of the triangle's own vertices under A: a hit the reference accepts lies on the
                // triangle up to the rounding of its triangle test, which the per-ray padding of the traversal covers
                // (trace_flat.h).  A triangle with a vertex that is not finite can never be hit and keeps the parent's box.
                const float *tv = tris[mesh[lv[j].node].triangleIndex].ver;
                bool finite = true;
                for(int q = 0; q < 9; q++) finite = finite && std::isfinite(tv[q]);
                if(finite) {
                    double tlo[3] = {1e300, 1e300, 1e300}, thi[3] = {-1e300, -1e300, -1e300};
                    for(int vtx = 0; vtx < 3; vtx++)
                        for(int r = 0; r < 3; r++) {
                            const double v = A.a[r][0] * tv[3 * vtx] + A.a[r][1] * tv[3 * vtx + 1] + A.a[r][2] * tv[3 * vtx + 2] + A.a[r][3];
                            tlo[r] = std::min(tlo[r], v);
                            thi[r] = std::max(thi[r], v);
                        }
                    for(int r = 0; r < 3; r++) {
                        lo[r] = std::max(lo[r], tlo[r]);
                        hi[r] = std::min(hi[r], thi[r]);
                        if(lo[r] > hi[r]) lo[r] = hi[r] = 0.5 * (lo[r] + hi[r]);  // (they meet in a face, up to rounding)
                    }
                }
                for(int k = 0; k < 3; k++) {
                    // (the double sums carry errors far below one float step: one more step outwards covers them)
                    b.mi[k] = std::nextafterf(std::nextafterf((float)lo[k], -INFINITY), -INFINITY);
                    b.mx[k] = std::nextafterf(std::nextafterf((float)hi[k], INFINITY), INFINITY);
                }
            }
            for(int k = 0; k < 3; k++)
                if(!std::isfinite(b.mi[k]) || !std::isfinite(b.mx[k]) || std::fabs(b.mi[k]) > 1e12f || std::fabs(b.mx[k]) > 1e12f) { bad[i] = 1; return; }
        }
    });
    for(size_t i = 0; i < n_inst; i++)
        if(bad[i]) {
            why_not = "instance " + std::to_string(i) + ": a box is not finite or lies beyond 1e12";
            return false;
        }
    // ---- one SAH tree over them, then four-wide
    SahTree tree;
    // (built over the boxes the traversal will cull with)
    build_sah_tree(boxes.data(), (int)n, tree, (int)threads);
    std::vector<Box>().swap(boxes);
    const size_t n_nodes = tree.nodes.size();
    std::vector<BVHNodeInfo> nodes(n_nodes);
    std::vector<uint32_t> leaf_slot(n_nodes, 0);
    uint32_t n_leaf_seen = 0;
    for(size_t k = 0; k < n_nodes; k++) {  // (children come after their parents; leaves numbered in that order)
        const SahNode &sn = tree.nodes[k];
        BVHNodeInfo &b = nodes[k];
        for(int a = 0; a < 3; a++) { b.aa[a] = sn.box.mi[a]; b.bb[a] = sn.box.mx[a]; }
        b.aa[3] = b.bb[3] = 0.0f;
        b.lsIndex = sn.left;
        b.rsIndex = sn.right;
        b.instanceIndex = -1;
        b.triangleIndex = sn.prim;
        if(sn.prim >= 0) leaf_slot[k] = 2u * n_leaf_seen++;
    }
    if(tree.root != 0 || n_leaf_seen != n) {
        why_not = "internal: unexpected tree shape";
        return false;
    }
    std::vector<SahNode>().swap(tree.nodes);
    std::vector<uint32_t> wide_of;
    std::vector<uint8_t> wide_height;
    const uint32_t wide_base = 2u * (uint32_t)n;
    const uint32_t end = build_wide_nodes(
        nodes.data(), n_nodes, std::vector<int32_t>{0}, true, true, [&](int32_t c) { return nodes[c].triangleIndex >= 0; },
        [&](int32_t c) { return make_ref(REF_TRIANGLE, leaf_slot[c]); }, false, REF_MESH_INNER, wide_base, true, out.slots, wide_of,
        wide_height);
    if(end == REF_NONE || end > REF_SLOT_MASK) {
        why_not = "too many leaves for 29-bit slots";
        return false;
    }
    out.root_ref = n == 1 ? make_ref(REF_TRIANGLE, 0) : make_ref(REF_MESH_INNER, wide_of[0]);
    out.max_stack = 3 * (wide_height.empty() ? 0 : wide_height[0]) + 2;
    out.n_leaves = n;
    // ---- the leaf records
    std::vector<size_t> entry_first = first;
    parallel_items((n_nodes + 65535) / 65536, threads, [&](size_t ch) {
        const size_t lo = ch * 65536, hi = std::min(n_nodes, lo + 65536);
        for(size_t k = lo; k < hi; k++) {
            if(nodes[k].triangleIndex < 0) continue;
            const size_t e = (size_t)nodes[k].triangleIndex;  // leaf number in (instance, leaf of its mesh) order
            const size_t i = (size_t)(std::upper_bound(entry_first.begin(), entry_first.end(), e) - entry_first.begin()) - 1;
            const MeshLeaf &ml = leaves_of[root_of[i]][e - entry_first[i]];
            FlatLeaf &l = *reinterpret_cast<FlatLeaf *>(&out.slots[leaf_slot[k]]);
            memset(&l, 0, sizeof l);
            const float inf = std::numeric_limits<float>::infinity();
            const float *v = tris[mesh[ml.node].triangleIndex].ver;
            for(int a = 0; a < 3; a++) {
                l.parent_lo[a] = ml.parent < 0 ? -inf : mesh[ml.parent].aa[a];
                l.parent_hi[a] = ml.parent < 0 ? inf : mesh[ml.parent].bb[a];
                l.v0[a] = v[a];
                l.e1[a] = v[3 + a] - v[a];
                l.e2[a] = v[6 + a] - v[a];
            }
            l.triangle = (uint32_t)mesh[ml.node].triangleIndex;
            for(int r = 0; r < 3; r++)
                for(int c = 0; c < 4; c++) l.row[r][c] = inst[i].world2local[c * 4 + r];
            l.instance = (uint32_t)i;
        }
    });
    (void)n_tris;
    out.pad = (float)(64.0 * 5.9604644775390625e-8 * cond_max);
    double reach = 0.0;
    for(int a = 0; a < 3; a++) reach += std::max(std::fabs((double)scene[0].aa[a]), std::fabs((double)scene[0].bb[a]));
    out.reach = (float)reach;
    if(!std::isfinite(out.reach)) {
        why_not = "the scene box is not finite";
        return false;
    }
    return true;
}

namespace {

// Child i of the wide node at `rec` as the ordered traversal sees it: reference and box
// (NaN = that plane constrains nothing).
void decode_child(const TNode *rec, bool quantised, int i, uint32_t &ref, float lo[3], float hi[3]) {
    if(quantised) {
        const QNode &q = *reinterpret_cast<const QNode *>(rec);
        ref = q.c[i];
        for(int a = 0; a < 3; a++) {
            lo[a] = std::fmaf((float)q.qlo[a][i], q.scale[a], q.origin[a]);  // 0 * inf = NaN: unbounded
            hi[a] = std::fmaf((float)q.qhi[a][i], q.scale[a], q.origin[a]);
        }
    } else {
        const WNode &w = *reinterpret_cast<const WNode *>(rec);
        const float *los[4] = {w.lo0, w.lo1, w.lo2, w.lo3}, *his[4] = {w.hi0, w.hi1, w.hi2, w.hi3};
        ref = w.c[i];
        for(int a = 0; a < 3; a++) {
            lo[a] = los[i][a];
            hi[a] = his[i][a];
        }
    }
}

// decoded box contains [aa, bb]?  (a NaN plane contains everything; a NaN in the reference box
// can only be contained by a NaN plane)
bool contains(const float lo[3], const float hi[3], const float aa[3], const float bb[3], uint64_t &unbounded) {
    for(int a = 0; a < 3; a++) {
        if(std::isnan(lo[a]) || std::isnan(hi[a])) {
            unbounded++;
            continue;
        }
        if(!(lo[a] <= std::min(aa[a], bb[a])) || !(hi[a] >= std::max(aa[a], bb[a]))) return false;
    }
    return true;
}

}  // namespace

bool verify_accel_levels(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const AccelMeshLevel &ml, const BVHNodeInfo *scene,
                         size_t n_scene_nodes, size_t n_inst, const AccelSceneLevel &sl, uint64_t counts[4], std::string &err) {
    for(int k = 0; k < 4; k++) counts[k] = 0;
    if(ml.slots.empty()) {
        err = "the mesh level no longer holds its slots";
        return false;
    }
    // ---- mesh level
    std::vector<int32_t> node_of_slot(ml.slots.size(), -1);
    for(size_t n = 0; n < n_mesh_nodes; n++) {
        if(ml.slot_of[n] >= ml.wide_base) {
            err = "a binary slot lies in the wide area";
            return false;
        }
        node_of_slot[ml.slot_of[n]] = (int32_t)n;
    }
    std::vector<int32_t> owner_of_wide(ml.slots.size(), -1);
    for(size_t n = 0; n < n_mesh_nodes; n++)
        if(ml.wide_of[n] != REF_NONE) {
            if(ml.wide_of[n] < ml.wide_base || ml.wide_of[n] + ml.wide_stride > ml.slots.size() ||
               (ml.wide_of[n] - ml.wide_base) % ml.wide_stride) {
                err = "wide slot out of range";
                return false;
            }
            owner_of_wide[ml.wide_of[n]] = (int32_t)n;
            counts[0]++;
        }
    int deepest_mesh = 0;
    for(size_t r = 0; r < n_mesh_nodes; r++) {
        if(ml.wide_of[r] == REF_NONE) continue;
        // triangles under r: binary walk
        std::vector<int32_t> want, got, work{(int32_t)r};
        while(!work.empty()) {
            const int32_t n = work.back();
            work.pop_back();
            if(mesh[n].triangleIndex >= 0) want.push_back(mesh[n].triangleIndex);
            else {
                work.push_back(mesh[n].lsIndex);
                work.push_back(mesh[n].rsIndex);
            }
        }
        // ... and the wide walk, with the number of entries it leaves deferred
        struct W { uint32_t slot; int pending; };
        std::vector<W> wwork{{ml.wide_of[r], 0}};
        while(!wwork.empty()) {
            const W cur = wwork.back();
            wwork.pop_back();
            int present = 0;
            for(int i = 0; i < 4; i++) {
                uint32_t ref;
                float lo[3], hi[3];
                decode_child(&ml.slots[cur.slot], ml.quantised, i, ref, lo, hi);
                if(ref == REF_NONE) {
                    if(i < 2) {
                        err = "a wide node lacks one of its first two children";
                        return false;
                    }
                    continue;
                }
                present++;
                const uint32_t kind = ref >> 30;
                const bool pair = kind == REF_TRIANGLE && (ref & REF_PAIR);
                const uint32_t slot = kind == REF_TRIANGLE ? (ref & REF_SLOT_MASK) : (ref & REF_INDEX_MASK);
                if(slot + (pair ? 1u : 0u) >= ml.slots.size()) {
                    err = "a child reference leaves the mesh buffer";
                    return false;
                }
                counts[2]++;
                if(kind == REF_TRIANGLE) {
                    for(uint32_t s2 = slot; s2 <= slot + (pair ? 1u : 0u); s2++) {
                        if(s2 >= ml.wide_base || node_of_slot[s2] < 0 || mesh[node_of_slot[s2]].triangleIndex < 0) {
                            err = "a triangle reference does not name a leaf slot";
                            return false;
                        }
                        const TLeaf &l = *reinterpret_cast<const TLeaf *>(&ml.slots[s2]);
                        got.push_back((int32_t)l.triangle);
                        if(!contains(lo, hi, l.parent_lo, l.parent_hi, counts[3])) {
                            err = "a triangle child's box does not contain its parent's box";
                            return false;
                        }
                    }
                    if(pair && memcmp(&ml.slots[slot], &ml.slots[slot + 1], 6 * sizeof(float)) != 0) {
                        err = "the two leaves of a pair have different parents";
                        return false;
                    }
                } else if(kind == REF_MESH_INNER) {
                    if(owner_of_wide[slot] < 0) {
                        err = "an inner reference does not name a wide node";
                        return false;
                    }
                    const BVHNodeInfo &m = mesh[owner_of_wide[slot]];
                    if(!contains(lo, hi, m.aa, m.bb, counts[3])) {
                        err = "an inner child's box does not contain the reference node's box";
                        return false;
                    }
                } else {
                    err = "a scene-level reference inside a mesh tree";
                    return false;
                }
            }
            // the walk into any child leaves at most the other present ones deferred
            for(int i = 0; i < 4; i++) {
                uint32_t ref;
                float lo[3], hi[3];
                decode_child(&ml.slots[cur.slot], ml.quantised, i, ref, lo, hi);
                if(ref == REF_NONE) continue;
                const int pending = cur.pending + present - 1;
                deepest_mesh = std::max(deepest_mesh, pending);
                if((ref >> 30) == REF_MESH_INNER) wwork.push_back({ref & REF_INDEX_MASK, pending});
            }
        }
        std::sort(want.begin(), want.end());
        std::sort(got.begin(), got.end());
        if(want != got) {
            err = "the wide tree under mesh node " + std::to_string(r) + " reaches other triangles than the binary tree";
            return false;
        }
    }
    // ---- scene level
    const size_t inst_base = sl.header.inst_base, wide_base = inst_base + 2 * n_inst;
    if(sl.scene_nodes.size() != inst_base || sl.instances.size() != n_inst) {
        err = "scene level sizes are inconsistent";
        return false;
    }
    counts[1] = sl.wide_nodes.size() / (ml.quantised ? 1 : 2);
    int deepest_scene = 0;
    if(scene[0].instanceIndex < 0) {
        std::vector<int32_t> want, got, work{0};
        while(!work.empty()) {
            const int32_t n = work.back();
            work.pop_back();
            if(scene[n].instanceIndex >= 0) want.push_back(scene[n].instanceIndex);
            else {
                work.push_back(scene[n].lsIndex);
                work.push_back(scene[n].rsIndex);
            }
        }
        struct W { uint32_t slot; int pending; };
        if((sl.header.wide_root_ref >> 30) != REF_SCENE_INNER) {
            err = "the scene's wide root is not a scene node";
            return false;
        }
        std::vector<W> wwork{{sl.header.wide_root_ref & REF_INDEX_MASK, 0}};
        while(!wwork.empty()) {
            const W cur = wwork.back();
            wwork.pop_back();
            if(cur.slot < wide_base || cur.slot - wide_base + (ml.quantised ? 1 : 2) > sl.wide_nodes.size()) {
                err = "a scene-level wide reference leaves the scene buffer";
                return false;
            }
            int present = 0;
            uint32_t refs[4];
            for(int i = 0; i < 4; i++) {
                float lo[3], hi[3];
                decode_child(&sl.wide_nodes[cur.slot - wide_base], ml.quantised, i, refs[i], lo, hi);
                if(refs[i] == REF_NONE) continue;
                present++;
                counts[2]++;
                const uint32_t kind = refs[i] >> 30, slot = refs[i] & REF_INDEX_MASK;
                if(kind == REF_INSTANCE) {
                    if(slot < inst_base || slot >= wide_base || (slot - inst_base) % 2) {
                        err = "an instance reference does not name an instance record";
                        return false;
                    }
                    const InstAccel &ia = sl.instances[(slot - inst_base) / 2];
                    got.push_back((int32_t)((slot - inst_base) / 2));
                    if(!contains(lo, hi, ia.world_lo, ia.world_hi, counts[3])) {
                        err = "an instance child's box does not contain the scene leaf box";
                        return false;
                    }
                } else if(kind != REF_SCENE_INNER) {
                    err = "a mesh-level reference inside the scene tree";
                    return false;
                }
            }
            for(int i = 0; i < 4; i++) {
                if(refs[i] == REF_NONE) continue;
                const int pending = cur.pending + present - 1;
                deepest_scene = std::max(deepest_scene, pending);
                if((refs[i] >> 30) == REF_SCENE_INNER) wwork.push_back({refs[i] & REF_INDEX_MASK, pending});
            }
        }
        std::sort(want.begin(), want.end());
        std::sort(got.begin(), got.end());
        if(want != got) {
            err = "the wide scene tree reaches other instances than the binary tree";
            return false;
        }
    }
    if(deepest_mesh + deepest_scene > sl.max_stack) {
        err = "max_stack " + std::to_string(sl.max_stack) + " is below the deepest walk " + std::to_string(deepest_mesh + deepest_scene);
        return false;
    }
    return true;
}

bool verify_accel_flat(const BVHNodeInfo *mesh, size_t n_mesh_nodes, const TriangleRec *tris, const InstanceInfo *inst, size_t n_inst,
                       const AccelFlat &fl, uint64_t counts[4], std::string &err) {
    for(int k = 0; k < 4; k++) counts[k] = 0;
    const size_t n_slots = fl.slots.size(), n = fl.n_leaves;
    if(n == 0 || n_slots < 2 * n) {
        err = "the flat tree no longer holds its slots";
        return false;
    }
    // what must be there: every (instance, triangle) once, with the triangle's reference parent
    struct Want { uint64_t key; int32_t parent; };
    std::vector<Want> want;
    for(size_t i = 0; i < n_inst; i++) {
        std::vector<std::pair<int32_t, int32_t>> work{{inst[i].meshIndex, -1}};
        while(!work.empty()) {
            const auto e = work.back();
            work.pop_back();
            if(e.first < 0 || (size_t)e.first >= n_mesh_nodes) {
                err = "mesh node out of range";
                return false;
            }
            if(mesh[e.first].triangleIndex >= 0) {
                want.push_back({((uint64_t)i << 32) | (uint32_t)mesh[e.first].triangleIndex, e.second});
            } else {
                work.push_back({mesh[e.first].lsIndex, e.first});
                work.push_back({mesh[e.first].rsIndex, e.first});
            }
        }
    }
    std::sort(want.begin(), want.end(), [](const Want &a, const Want &b) { return a.key < b.key; });
    if(want.size() != n) {
        err = "the flat tree holds " + std::to_string(n) + " leaves, the scene has " + std::to_string(want.size());
        return false;
    }
    // world box of a leaf, from its own record: corners of the parent box through the inverse of its rows, cut
    // down to the box of the triangle's vertices through the same inverse
    auto leaf_box = [&](const FlatLeaf &l, float lo[3], float hi[3]) {
        long double w[3][4], a[3][3];
        for(int r = 0; r < 3; r++)
            for(int c = 0; c < 4; c++) w[r][c] = l.row[r][c];
        const long double det = w[0][0] * (w[1][1] * w[2][2] - w[1][2] * w[2][1]) - w[0][1] * (w[1][0] * w[2][2] - w[1][2] * w[2][0]) +
                                w[0][2] * (w[1][0] * w[2][1] - w[1][1] * w[2][0]);
        for(int r = 0; r < 3; r++)
            for(int c = 0; c < 3; c++) {
                // cofactor of w[c][r] over det
                const int r1 = (c + 1) % 3, r2 = (c + 2) % 3, c1 = (r + 1) % 3, c2 = (r + 2) % 3;
                a[r][c] = (w[r1][c1] * w[r2][c2] - w[r1][c2] * w[r2][c1]) / det;
            }
        for(int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; }
        for(int corner = 0; corner < 8; corner++) {
            const long double p[3] = {(corner & 1 ? l.parent_hi[0] : l.parent_lo[0]), (corner & 2 ? l.parent_hi[1] : l.parent_lo[1]),
                                      (corner & 4 ? l.parent_hi[2] : l.parent_lo[2])};
            for(int r = 0; r < 3; r++) {
                const long double v = a[r][0] * (p[0] - w[0][3]) + a[r][1] * (p[1] - w[1][3]) + a[r][2] * (p[2] - w[2][3]);
                lo[r] = std::min(lo[r], (float)v);
                hi[r] = std::max(hi[r], (float)v);
            }
        }
        const float *tv = tris[l.triangle].ver;
        for(int q = 0; q < 9; q++)
            if(!std::isfinite(tv[q])) return;
        float tlo[3], thi[3];
        for(int k = 0; k < 3; k++) { tlo[k] = std::numeric_limits<float>::infinity(); thi[k] = -tlo[k]; }
        for(int vtx = 0; vtx < 3; vtx++)
            for(int r = 0; r < 3; r++) {
                const long double v = a[r][0] * (tv[3 * vtx] - w[0][3]) + a[r][1] * (tv[3 * vtx + 1] - w[1][3]) + a[r][2] * (tv[3 * vtx + 2] - w[2][3]);
                tlo[r] = std::min(tlo[r], (float)v);
                thi[r] = std::max(thi[r], (float)v);
            }
        for(int r = 0; r < 3; r++) {
            lo[r] = std::max(lo[r], tlo[r]);
            hi[r] = std::min(hi[r], thi[r]);
            if(lo[r] > hi[r]) lo[r] = hi[r];  // (they meet in a face: a float step apart at most, either end lies in the built box)
        }
    };
    std::vector<uint8_t> seen(n_slots, 0);
    std::vector<Want> got;
    got.reserve(n);
    uint64_t unbounded = 0;
    bool ok = true;
    int deepest = 0;
    // returns the box of everything under `ref`; depth = wide nodes above it
    struct Bx { float lo[3], hi[3]; };
    auto visit_leaf = [&](uint32_t slot, Bx &b) {
        if(slot + 2 > 2 * n || (slot & 1u) || seen[slot]) {
            err = "a leaf reference is out of range or names a leaf twice";
            return false;
        }
        seen[slot] = 1;
        const FlatLeaf &l = *reinterpret_cast<const FlatLeaf *>(&fl.slots[slot]);
        if(l.instance >= n_inst) {
            err = "a leaf names an instance out of range";
            return false;
        }
        for(int r = 0; r < 3; r++)
            for(int c = 0; c < 4; c++)
                if(memcmp(&l.row[r][c], &inst[l.instance].world2local[c * 4 + r], 4)) {
                    err = "a leaf's matrix is not its instance's";
                    return false;
                }
        const float *v = tris[l.triangle].ver;
        for(int k = 0; k < 3; k++)
            if(l.v0[k] != v[k] || l.e1[k] != v[3 + k] - v[k] || l.e2[k] != v[6 + k] - v[k]) {
                if(!(std::isnan(v[k]) || std::isnan(v[3 + k]) || std::isnan(v[6 + k]))) {
                    err = "a leaf's triangle is not the one it names";
                    return false;
                }
            }
        got.push_back({((uint64_t)l.instance << 32) | l.triangle, 0});
        if(std::isinf(l.parent_lo[0])) {  // a one-triangle mesh: bounded by the instance's scene leaf box, not checked here
            for(int k = 0; k < 3; k++) { b.lo[k] = std::numeric_limits<float>::infinity(); b.hi[k] = -b.lo[k]; }
        } else {
            leaf_box(l, b.lo, b.hi);
        }
        return true;
    };
    struct Rec {
        static bool walk(uint32_t ref, int depth, Bx &out, const AccelFlat &fl, size_t n, size_t n_slots, uint64_t counts[4], uint64_t &unbounded,
                         int &deepest, std::string &err, const std::function<bool(uint32_t, Bx &)> &leaf) {
            if((ref >> 30) == REF_TRIANGLE) {
                const uint32_t slot = ref & REF_SLOT_MASK;
                if(!leaf(slot, out)) return false;
                if(ref & REF_PAIR) {
                    Bx b2;
                    if(!leaf(slot + 2, b2)) return false;
                    for(int k = 0; k < 3; k++) { out.lo[k] = std::min(out.lo[k], b2.lo[k]); out.hi[k] = std::max(out.hi[k], b2.hi[k]); }
                }
                return true;
            }
            const uint32_t slot = ref & REF_INDEX_MASK;
            if((ref >> 30) != REF_MESH_INNER || slot < 2 * n || slot >= n_slots || depth > 200) {
                err = "a node reference is out of range";
                return false;
            }
            counts[1]++;
            deepest = std::max(deepest, depth + 1);
            for(int k = 0; k < 3; k++) { out.lo[k] = std::numeric_limits<float>::infinity(); out.hi[k] = -out.lo[k]; }
            for(int i = 0; i < 4; i++) {
                uint32_t cref;
                float lo[3], hi[3];
                decode_child(&fl.slots[slot], true, i, cref, lo, hi);
                if(cref == REF_NONE) continue;
                Bx below;
                if(!walk(cref, depth + 1, below, fl, n, n_slots, counts, unbounded, deepest, err, leaf)) return false;
                counts[2]++;
                if(below.lo[0] <= below.hi[0] && !contains(lo, hi, below.lo, below.hi, unbounded)) {
                    err = "a child box does not contain the leaves below it";
                    return false;
                }
                for(int k = 0; k < 3; k++) { out.lo[k] = std::min(out.lo[k], below.lo[k]); out.hi[k] = std::max(out.hi[k], below.hi[k]); }
            }
            return true;
        }
    };
    Bx all;
    const std::function<bool(uint32_t, Bx &)> leaf_fn = visit_leaf;
    ok = Rec::walk(fl.root_ref, 0, all, fl, n, n_slots, counts, unbounded, deepest, err, leaf_fn);
    if(!ok) return false;
    std::sort(got.begin(), got.end(), [](const Want &a, const Want &b) { return a.key < b.key; });
    for(size_t k = 0; k < n; k++) {
        if(got.size() != n || got[k].key != want[k].key) {
            err = "the flat tree reaches other (instance, triangle) pairs than the scene holds";
            return false;
        }
    }
    // parent boxes: the reference parent's, bit for bit
    for(size_t slot = 0; slot < 2 * n; slot += 2) {
        const FlatLeaf &l = *reinterpret_cast<const FlatLeaf *>(&fl.slots[slot]);
        const uint64_t key = ((uint64_t)l.instance << 32) | l.triangle;
        const auto it = std::lower_bound(want.begin(), want.end(), key, [](const Want &a, uint64_t k) { return a.key < k; });
        const float inf = std::numeric_limits<float>::infinity();
        for(int k = 0; k < 3; k++) {
            const float plo = it->parent < 0 ? -inf : mesh[it->parent].aa[k], phi = it->parent < 0 ? inf : mesh[it->parent].bb[k];
            if(memcmp(&plo, &l.parent_lo[k], 4) || memcmp(&phi, &l.parent_hi[k], 4)) {
                err = "a leaf's parent box is not its reference parent's";
                return false;
            }
        }
    }
    if(3 * deepest > fl.max_stack) {
        err = "max_stack " + std::to_string(fl.max_stack) + " is below the deepest walk " + std::to_string(3 * deepest);
        return false;
    }
    counts[0] = n;
    counts[3] = unbounded;
    return true;
}

}  // namespace rta
