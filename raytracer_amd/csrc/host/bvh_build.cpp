// Task-parallel BVH builder whose output is bit-identical to the reference's single-threaded
// recursive builder (Bvh::new / subdivide / find_best_split / find_split_approx /
// evaluate_split, scene.rs:888-1050).
//
// The reference evaluates every candidate plane with a full sweep over the node's triangles.
// The cost it computes depends only on WHICH triangles fall left of the plane (box unions are
// exact min/max, the f32 side counters only depend on the count), so the same numbers come out
// of a binned sweep (>= 64 triangles: 16 monotone planes per axis -> 17 bins, prefix/suffix
// unions) and of a sorted sweep (< 64 triangles: candidates are the centroids themselves).
// What does depend on order -- the first-best-wins tie rule over candidates in index order, the
// exact Hoare-style partition with its swap-to-the-back step, and the depth-first pair numbering
// of nodes -- is reproduced literally.  Subtrees are independent once their index range is
// fixed, so they are built as parallel tasks into a pool and renumbered into the reference's
// depth-first order afterwards from the subtree sizes.
//
// All float arithmetic is binary32 without contraction (compile with -ffp-contract=off).

#include "rthost_internal.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace rthost {

namespace {

struct Box {
    float d[6];
};

inline Box empty_box() { // Aabb::new, scene.rs:774-778 (f32::MIN is -FLT_MAX)
    return Box{{FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX}};
}
inline void grow(Box& a, const Box& b) { // Aabb::with, scene.rs:785-794
    for (int k = 0; k < 3; k++) a.d[k] = std::fmin(a.d[k], b.d[k]);
    for (int k = 3; k < 6; k++) a.d[k] = std::fmax(a.d[k], b.d[k]);
}
inline float surface(const Box& b) { // scene.rs:780-783
    float sx = b.d[3] - b.d[0], sy = b.d[4] - b.d[1], sz = b.d[5] - b.d[2];
    return (sx * sy + sy * sz + sz * sx) * 2.0f;
}
// The reference counts triangles per side in an f32 incremented by 1.0 (scene.rs:924-933);
// such a counter stops at 2^24.
inline float f32_counter(uint64_t n) { return n >= 16777216ull ? 16777216.0f : (float)n; }

inline float sah_cost(uint64_t ln, const Box& lb, uint64_t rn, const Box& rb) {
    float cost = f32_counter(ln) * surface(lb) + f32_counter(rn) * surface(rb);
    return cost > 0.0f ? cost : FLT_MAX; // NaN from an empty side fails the test (scene.rs:941-945)
}

struct PoolNode {
    Box      box;
    uint32_t first;
    uint32_t count;     // > 0: leaf (for now); 0: inner
    uint32_t child;     // pool index of the left child (right = child + 1) when inner
};

struct Builder {
    const float* tris;      // 12 floats per record; centroid rides in lanes 3, 7, 11
    uint32_t     offset;
    uint32_t     size;
    uint32_t*    indices;   // indices[i - offset] = global triangle id
    std::vector<Box> tri_box; // by (global id - offset)

    std::vector<PoolNode> pool;
    std::atomic<uint32_t> pool_used{0};

    // task queue
    std::mutex              mu;
    std::condition_variable cv;
    std::vector<uint32_t>   tasks;
    int                     pending = 0;
    bool                    parallel = false;
    static constexpr uint32_t kSpawnThreshold = 4096;

    float centroid(uint32_t gid, int axis) const { return tris[(size_t)gid * 12 + axis * 4 + 3]; }
    const Box& box_of(uint32_t gid) const { return tri_box[gid - offset]; }

    Box range_box(uint32_t first, uint32_t count) const { // update_aabb, scene.rs:870-877
        Box b = box_of(indices[first - offset]);
        for (uint32_t i = first; i < first + count; i++) grow(b, box_of(indices[i - offset]));
        return b;
    }

    // find_split_approx (scene.rs:969-987) through 17 bins per axis
    void split_binned(const PoolNode& n, int& best_axis, float& best_split) const {
        best_axis = 0;
        best_split = 0.0f;
        float best_cost = FLT_MAX;
        for (int axis = 0; axis < 3; axis++) {
            float planes[16];
            const float mn = n.box.d[axis], mx = n.box.d[3 + axis];
            for (int i = 0; i < 16; i++) planes[i] = mn + (((float)i + 0.5f) / 16.0f) * (mx - mn);
            Box      bin_box[17];
            uint64_t bin_n[17];
            for (int k = 0; k < 17; k++) { bin_box[k] = empty_box(); bin_n[k] = 0; }
            for (uint32_t i = n.first; i < n.first + n.count; i++) {
                const uint32_t gid = indices[i - offset];
                const float c = centroid(gid, axis);
                // planes are non-decreasing, so "c < planes[i]" holds from some i on
                int k = 0;
                while (k < 16 && !(c < planes[k])) k++;
                bin_n[k]++;
                grow(bin_box[k], box_of(gid));
            }
            Box      suffix[18];
            uint64_t suffix_n[18];
            suffix[17] = empty_box();
            suffix_n[17] = 0;
            for (int k = 16; k >= 0; k--) {
                suffix[k] = suffix[k + 1];
                grow(suffix[k], bin_box[k]);
                suffix_n[k] = suffix_n[k + 1] + bin_n[k];
            }
            Box      left = empty_box();
            uint64_t left_n = 0;
            for (int i = 0; i < 16; i++) {
                grow(left, bin_box[i]);
                left_n += bin_n[i];
                const float cost = sah_cost(left_n, left, suffix_n[i + 1], suffix[i + 1]);
                if (cost < best_cost) { best_axis = axis; best_cost = cost; best_split = planes[i]; }
            }
        }
    }

    // find_best_split (scene.rs:948-967) through a sorted sweep; n.count < 64
    void split_exact(const PoolNode& n, int& best_axis, float& best_split) const {
        best_axis = 0;
        best_split = 0.0f;
        float best_cost = FLT_MAX;
        const uint32_t cnt = n.count;
        float    cand[64];
        uint32_t order[64];
        Box      prefix[65], suffix[65];
        for (int axis = 0; axis < 3; axis++) {
            for (uint32_t k = 0; k < cnt; k++) {
                cand[k] = centroid(indices[n.first + k - offset], axis);
                order[k] = k;
            }
            std::sort(order, order + cnt, [&](uint32_t a, uint32_t b) { return cand[a] < cand[b]; });
            prefix[0] = empty_box();
            for (uint32_t m = 0; m < cnt; m++) {
                prefix[m + 1] = prefix[m];
                grow(prefix[m + 1], box_of(indices[n.first + order[m] - offset]));
            }
            suffix[cnt] = empty_box();
            for (uint32_t m = cnt; m-- > 0;) {
                suffix[m] = suffix[m + 1];
                grow(suffix[m], box_of(indices[n.first + order[m] - offset]));
            }
            for (uint32_t k = 0; k < cnt; k++) { // candidates in index order, first best wins
                const float split = cand[k];
                // m = number of centroids strictly below the candidate
                uint32_t lo = 0, hi = cnt;
                while (lo < hi) {
                    uint32_t mid = (lo + hi) / 2;
                    if (cand[order[mid]] < split) lo = mid + 1; else hi = mid;
                }
                const float cost = sah_cost(lo, prefix[lo], cnt - lo, suffix[lo]);
                if (cost < best_cost) { best_axis = axis; best_cost = cost; best_split = split; }
            }
        }
    }

    // One subdivide() step (scene.rs:991-1046).  Returns false when the node stays a leaf.
    bool split_node(uint32_t idx) {
        PoolNode n = pool[idx];
        if (n.count <= 2) return false;
        int   axis;
        float split;
        if (n.count < 64) split_exact(n, axis, split);
        else split_binned(n, axis, split);

        size_t i = n.first;
        size_t j = (size_t)n.first + n.count - 1;
        while (i <= j) { // scene.rs:1006-1022, literally
            const uint32_t gid = indices[i - offset];
            if (centroid(gid, axis) < split) {
                i += 1;
            } else {
                std::swap(indices[i - offset], indices[j - offset]);
                if (j == 0) break;
                j -= 1;
            }
        }
        const uint32_t left_count = (uint32_t)i - n.first;
        if (left_count == 0 || left_count == n.count) return false;

        const uint32_t c = pool_used.fetch_add(2);
        PoolNode& l = pool[c];
        PoolNode& r = pool[c + 1];
        l.first = n.first;
        l.count = left_count;
        l.child = 0;
        l.box = range_box(l.first, l.count);
        r.first = (uint32_t)i;
        r.count = n.count - left_count;
        r.child = 0;
        r.box = range_box(r.first, r.count);
        pool[idx].count = 0;
        pool[idx].child = c;
        return true;
    }

    void build_subtree(uint32_t root) {
        std::vector<uint32_t> stack;
        stack.push_back(root);
        while (!stack.empty()) {
            const uint32_t idx = stack.back();
            stack.pop_back();
            if (!split_node(idx)) continue;
            const uint32_t c = pool[idx].child;
            for (uint32_t k = 0; k < 2; k++) {
                if (parallel && pool[c + k].count >= kSpawnThreshold) push_task(c + k);
                else stack.push_back(c + k);
            }
        }
    }

    void push_task(uint32_t idx) {
        {
            std::lock_guard<std::mutex> lk(mu);
            tasks.push_back(idx);
            pending++;
        }
        cv.notify_one();
    }

    void worker() {
        for (;;) {
            uint32_t idx;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !tasks.empty() || pending == 0; });
                if (tasks.empty()) return; // pending == 0: everything is built
                idx = tasks.back();
                tasks.pop_back();
            }
            build_subtree(idx);
            {
                std::lock_guard<std::mutex> lk(mu);
                pending--;
                if (pending == 0) cv.notify_all();
            }
        }
    }
};

} // namespace

uint32_t bvh_build(const float* tris, uint32_t offset, uint32_t size, Node* nodes_out,
                   uint32_t* indices_out, int n_threads) {
    Builder b;
    b.tris = tris;
    b.offset = offset;
    b.size = size;
    b.indices = indices_out;
    for (uint32_t i = 0; i < size; i++) indices_out[i] = offset + i; // scene.rs:891

    // Tri::aabb, scene.rs:727-731 with Aabb::point's +-1e-5 padding (scene.rs:800-804)
    b.tri_box.resize(size);
    for (uint32_t i = 0; i < size; i++) {
        const float* t = tris + (size_t)(offset + i) * 12;
        Box bb;
        for (int k = 0; k < 3; k++) { bb.d[k] = t[k] - 0.00001f; bb.d[3 + k] = t[k] + 0.00001f; }
        for (int v = 1; v < 3; v++) {
            Box p;
            for (int k = 0; k < 3; k++) { p.d[k] = t[v * 4 + k] - 0.00001f; p.d[3 + k] = t[v * 4 + k] + 0.00001f; }
            grow(bb, p);
        }
        b.tri_box[i] = bb;
    }

    b.pool.resize((size_t)2 * size + 1);
    PoolNode& root = b.pool[0];
    root.first = offset;
    root.count = size;
    root.child = 0;
    if (size > 0) {
        root.box = b.range_box(offset, size);
    } else { // BvhNode::new, scene.rs:854-860: update_aabb leaves the placeholder box
        for (int k = 0; k < 3; k++) { root.box.d[k] = -100.0f - 0.00001f; root.box.d[3 + k] = -100.0f + 0.00001f; }
    }
    b.pool_used = 1;

    if (n_threads <= 0) {
        unsigned hc = std::thread::hardware_concurrency();
        n_threads = (int)std::min(16u, hc ? hc : 1u);
    }
    if (n_threads > 1 && size >= 2 * Builder::kSpawnThreshold) {
        b.parallel = true;
        b.push_task(0);
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back([&b] { b.worker(); });
        for (auto& t : th) t.join();
    } else {
        b.build_subtree(0);
    }

    // Renumber into the reference's order: a node's two children are appended when it is split
    // and its left subtree is completed before its right one (scene.rs:1041-1049), so the
    // children of a node whose children start at S have their own children starting at S+2
    // (left) and S+2+descendants(left) (right).  Children always sit after their parent in the
    // pool, so one backward pass sizes the subtrees and one forward pass places them.
    const uint32_t n = b.pool_used.load();
    std::vector<uint32_t> desc(n, 0), pos(n, 0), start(n, 0);
    for (uint32_t i = n; i-- > 0;) {
        const PoolNode& p = b.pool[i];
        if (p.count == 0 && size > 0) desc[i] = 2 + desc[p.child] + desc[p.child + 1];
    }
    pos[0] = 0;
    start[0] = 1;
    for (uint32_t i = 0; i < n; i++) {
        const PoolNode& p = b.pool[i];
        Node& o = nodes_out[pos[i]];
        std::memcpy(o.aabb, p.box.d, sizeof o.aabb);
        if (p.count == 0 && size > 0) {
            const uint32_t s = start[i];
            o.first = s;
            o.count = 0;
            pos[p.child] = s;
            pos[p.child + 1] = s + 1;
            start[p.child] = s + 2;
            start[p.child + 1] = s + 2 + desc[p.child];
        } else {
            o.first = p.first;
            o.count = p.count;
        }
    }
    return n;
}

} // namespace rthost
