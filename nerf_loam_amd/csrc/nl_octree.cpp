// nl_octree.cpp -- host-side sparse voxel octree behind the `svo.Octree` operator surface.
//
// Reference behaviour: third_party/sparse_octree/src/octree.cpp:36-111 (init, insert),
// :151-171 (find_octant), :263-283 (count), :293-342 (get_centres_and_children),
// include/utils.h:64-109 (Morton codes).  Same results (node id = creation order, SURFACE/FEATURE
// leaf types, BFS export that hides FEATURE children), different structure: nodes live in flat
// index-linked arrays (no per-node heap allocation, no pointer chasing across the heap), the node
// counter is per instance (the reference's is process-global), and the export writes
// straight into caller-provided buffers in the layouts the kernels consume.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <vector>

namespace {

enum : int8_t { T_NONLEAF = -1, T_SURFACE = 0, T_FEATURE = 1 };

struct Node {
    uint64_t code;
    int32_t child[8];
    uint32_t side;
    int8_t type;
    bool leaf;
};

struct Octree {
    int size = 0, max_level = 0;
    std::vector<Node> nodes;
    // nodes whose exported rows changed since the last nl_octree_export_delta: new nodes, parents that gained a child,
    // FEATURE leaves upgraded to SURFACE (incremental export instead of a full one per frame)
    std::vector<int> dirty;
    std::vector<uint8_t> dirty_flag;
    // path of the previous descent (node per depth) and its key: consecutive LiDAR returns fall into the same or neighbouring
    // voxels and the eight corner vertices of a voxel share all but the last levels, so a descent resumes at the deepest node it
    // shares with the previous one (nodes are never removed, and only the leaf level has side effects on existing nodes).
    // Needs level d <-> coordinate bit (max_level - d), i.e. a power-of-two grid; otherwise every descent starts at the root.
    int path[24] = {0};
    uint64_t last_key = 0, low_mask = 0;
    bool resume_ok = false, have_last = false;

    int new_node() {
        Node n;
        n.code = 0; n.side = 0; n.type = T_NONLEAF; n.leaf = false;
        for (int i = 0; i < 8; ++i) n.child[i] = -1;
        nodes.push_back(n);
        dirty_flag.push_back(0);
        return (int)nodes.size() - 1;
    }
    void mark(int k) { if (!dirty_flag[k]) { dirty_flag[k] = 1; dirty.push_back(k); } }
};

inline uint64_t spread3(uint64_t v) {
    uint64_t x = v & 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffffULL;
    x = (x | x << 16) & 0x1f0000ff0000ffULL;
    x = (x | x << 8) & 0x100f00f00f00f00fULL;
    x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
    x = (x | x << 2) & 0x1249249249249249ULL;
    return x;
}
inline uint64_t gather3(uint64_t v) {
    uint64_t x = v & 0x1249249249249249ULL;
    x = (x | x >> 2) & 0x10c30c30c30c30c3ULL;
    x = (x | x >> 4) & 0x100f00f00f00f00fULL;
    x = (x | x >> 8) & 0x1f0000ff0000ffULL;
    x = (x | x >> 16) & 0x1f00000000ffffULL;
    x = (x | x >> 32) & 0x1fffff;
    return x;
}
// prefix mask of a level-(i) node: top 3*(i+1) bits of the 63-bit key (utils.h:41-62 MASK[i])
inline uint64_t prefix_mask(int i) { return i >= 20 ? 0x7fffffffffffffffULL : (0x7fffffffffffffffULL & ~((1ULL << (60 - 3 * i)) - 1ULL)); }
inline uint64_t morton(int x, int y, int z) {
    return (spread3((uint64_t)(int64_t)x) | (spread3((uint64_t)(int64_t)y) << 1) | (spread3((uint64_t)(int64_t)z) << 2)) & prefix_mask(20);
}

const int DX[8] = {0, 0, 0, 0, 1, 1, 1, 1};
const int DY[8] = {0, 0, 1, 1, 0, 0, 1, 1};
const int DZ[8] = {0, 1, 0, 1, 0, 1, 0, 1};

inline int octant_of(int x, int y, int z, unsigned edge) {
    return ((x & edge) > 0) + 2 * ((y & edge) > 0) + 4 * ((z & edge) > 0);
}

int find_leaf(const Octree& t, int x, int y, int z) {
    int n = 0;
    unsigned edge = (unsigned)t.size / 2;
    for (int d = 1; d <= t.max_level; edge /= 2, ++d) {
        n = t.nodes[n].child[octant_of(x, y, z, edge)];
        if (n < 0) return -1;
    }
    return n;
}

// find_leaf for runs of nearby lookups (the eight corners of a voxel, voxels in creation order): resumes from the deepest node
// shared with the previous lookup, like the insert does.  `valid` = depth down to which path[] belongs to the last key.
struct LeafFinder {
    const Octree& t;
    int path[24];
    uint64_t last = 0;
    int valid = -1;
    explicit LeafFinder(const Octree& tree) : t(tree) { path[0] = 0; }
    int find(int x, int y, int z) {
        if (!t.resume_ok) return find_leaf(t, x, y, z);
        const int L = t.max_level;
        const uint64_t lk = morton(x, y, z) & t.low_mask;
        int p = 0;
        if (valid >= 0) {
            const uint64_t diff = lk ^ last;
            p = diff ? L - 1 - (63 - __builtin_clzll(diff)) / 3 : L;
            if (p > valid) p = valid;
        }
        last = lk;
        int n = path[p];
        unsigned edge = ((unsigned)t.size / 2) >> p;
        for (int d = p + 1; d <= L; edge /= 2, ++d) {
            n = t.nodes[n].child[octant_of(x, y, z, edge)];
            if (n < 0) { valid = d - 1; return -1; }
            path[d] = n;
        }
        valid = L;
        return n;
    }
};

}  // namespace

extern "C" {

void* nl_octree_create(long long grid_dim)
{
    if (grid_dim <= 1) return nullptr;
    Octree* t = new Octree();
    t->size = (int)grid_dim;
    t->max_level = (int)std::log2((double)t->size);
    int r = t->new_node();
    t->nodes[r].side = (uint32_t)t->size;
    t->mark(r);
    t->resume_ok = t->max_level >= 1 && t->max_level <= 20 && grid_dim == (1LL << t->max_level);
    t->low_mask = (1ULL << (3 * (t->max_level < 21 ? t->max_level : 20))) - 1ULL;
    return t;
}

void nl_octree_destroy(void* h) { delete (Octree*)h; }

// insert voxel coordinates [M,3] int32 (each voxel + its 7 (+x,+y,+z) neighbours as vertices)
int nl_octree_insert(void* h, const int* pts, long long npts)
{
    if (!h || (!pts && npts > 0)) return 1;
    Octree& t = *(Octree*)h;
    const int shift = 21 - t.max_level - 1, L = t.max_level;
    for (long long i = 0; i < npts; ++i) {
        for (int j = 0; j < 8; ++j) {
            const int x = pts[3 * i] + DX[j], y = pts[3 * i + 1] + DY[j], z = pts[3 * i + 2] + DZ[j];
            const uint64_t key = morton(x, y, z);
            const uint64_t lk = key & t.low_mask;
            int p = 0;                                               // levels shared with the previous descent
            if (t.resume_ok && t.have_last) {
                const uint64_t diff = lk ^ t.last_key;
                p = diff ? L - 1 - (63 - __builtin_clzll(diff)) / 3 : L - 1;      // the leaf level is always redone
            }
            int n = t.path[p];
            unsigned edge = ((unsigned)t.size / 2) >> p;
            bool was_surface = false;
            for (int d = p + 1; d <= L; edge /= 2, ++d) {
                const int cid = octant_of(x, y, z, edge);
                int c = t.nodes[n].child[cid];
                if (c < 0) {
                    c = t.new_node();
                    Node& nd = t.nodes[c];
                    nd.code = key & prefix_mask(d + shift);
                    nd.side = edge;
                    nd.leaf = (d == L);
                    nd.type = nd.leaf ? (j == 0 ? T_SURFACE : T_FEATURE) : T_NONLEAF;
                    t.nodes[n].child[cid] = c;
                    t.mark(c); t.mark(n);
                } else if (t.nodes[c].type == T_FEATURE && j == 0) {
                    t.nodes[c].type = T_SURFACE;
                    t.mark(c); t.mark(n);                              // becomes visible in its parent's children row
                } else if (d == L && j == 0) {
                    was_surface = t.nodes[c].type == T_SURFACE;
                }
                n = c;
                t.path[d] = n;
            }
            t.last_key = lk; t.have_last = true;
            // a voxel that already is a SURFACE leaf got there through an earlier insert of the same coordinates (modulo the grid
            // size), which also created its seven corner neighbours: the rest of this point changes nothing (most returns of a
            // scan repeat a voxel).  Not so on a non-power-of-two grid, where `x & edge` lets unrelated coordinates share a leaf.
            if (was_surface && t.resume_ok) break;
        }
    }
    return 0;
}

long long nl_octree_count_nodes(void* h) { return h ? (long long)((Octree*)h)->nodes.size() : 0; }

long long nl_octree_count_leaf_nodes(void* h)
{
    if (!h) return 0;
    long long c = 0;
    for (const Node& n : ((Octree*)h)->nodes) c += (n.type == T_SURFACE);
    return c;
}

int nl_octree_has_voxel(void* h, int x, int y, int z) { return h && find_leaf(*(Octree*)h, x, y, z) >= 0; }

// get_leaf_voxels (octree.cpp:212-240): integer coordinates of the SURFACE leaves in depth-first octant order.  out == NULL: count only.
long long nl_octree_leaf_voxels(void* h, float* out)
{
    if (!h) return 0;
    const Octree& t = *(Octree*)h;
    long long count = 0;
    std::vector<int> stack;
    stack.push_back(0);
    while (!stack.empty()) {
        const int k = stack.back();
        stack.pop_back();
        const Node& nd = t.nodes[k];
        if (nd.leaf) {
            if (nd.type == T_SURFACE) {
                if (out) {
                    out[3 * count] = (float)gather3(nd.code); out[3 * count + 1] = (float)gather3(nd.code >> 1);
                    out[3 * count + 2] = (float)gather3(nd.code >> 2);
                }
                ++count;
            }
            continue;
        }
        for (int i = 7; i >= 0; --i)                                  // pushed in reverse: popped in octant order 0..7
            if (nd.child[i] >= 0) stack.push_back(nd.child[i]);
    }
    return count;
}

// get_voxels (octree.cpp:242-265): (x, y, z, side) of EVERY node, depth-first pre-order in octant order; out: [count_nodes, 4]
int nl_octree_voxels_dfs(void* h, float* out)
{
    if (!h || !out) return 1;
    const Octree& t = *(Octree*)h;
    std::vector<int> stack;
    stack.push_back(0);
    size_t q = 0;
    while (!stack.empty()) {
        const int k = stack.back();
        stack.pop_back();
        const Node& nd = t.nodes[k];
        out[4 * q] = (float)gather3(nd.code); out[4 * q + 1] = (float)gather3(nd.code >> 1); out[4 * q + 2] = (float)gather3(nd.code >> 2);
        out[4 * q + 3] = (float)nd.side;
        ++q;
        for (int i = 7; i >= 0; --i)
            if (nd.child[i] >= 0) stack.push_back(nd.child[i]);
    }
    return 0;
}

// try_insert (octree.cpp:113-149): which fraction of the DISTINCT vertex keys of `pts` (each voxel + its 7 corner neighbours) is in
// the tree already.  The reference intersects with its set of inserted keys into a std::set<int>, i.e. it counts the matches by
// their low 32 key bits; kept.  Membership is read from the tree (leaf at the coordinates whose stored key equals the query),
// which equals the reference's key set for coordinates inside the grid.  No points: 0 / 0 = NaN, as there.
double nl_octree_try_insert(void* h, const int* pts, long long npts)
{
    if (!h || (!pts && npts > 0)) return -1.0;
    const Octree& t = *(Octree*)h;
    std::vector<uint64_t> keys;
    keys.reserve((size_t)npts * 8);
    for (long long i = 0; i < npts; ++i)
        for (int j = 0; j < 8; ++j) keys.push_back(morton(pts[3 * i] + DX[j], pts[3 * i + 1] + DY[j], pts[3 * i + 2] + DZ[j]));
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    std::vector<uint32_t> hit;
    LeafFinder finder(t);
    for (uint64_t key : keys) {
        const int n = finder.find((int)gather3(key), (int)gather3(key >> 1), (int)gather3(key >> 2));
        if (n >= 0 && t.nodes[n].code == key) hit.push_back((uint32_t)key);
    }
    std::sort(hit.begin(), hit.end());
    hit.erase(std::unique(hit.begin(), hit.end()), hit.end());
    return 1.0 * (double)hit.size() / (double)keys.size();
}

// get_centres_and_children: voxels[n,4] f32 (x,y,z,side; zero rows for FEATURE leaves),
// children[n,8] f32 (-1 = absent or FEATURE leaf), features[n,8] i32 (corner-vertex node ids of
// SURFACE leaves, -1 elsewhere).  Buffers are fully written.
int nl_octree_export(void* h, float* voxels, float* children, int* features)
{
    if (!h || !voxels || !children || !features) return 1;
    const Octree& t = *(Octree*)h;
    const size_t n = t.nodes.size();
    std::memset(voxels, 0, n * 4 * sizeof(float));
    for (size_t i = 0; i < n * 8; ++i) { children[i] = -1.0f; features[i] = -1; }
    std::vector<int> queue;
    queue.reserve(n);
    queue.push_back(0);
    LeafFinder finder(t);
    for (size_t head = 0; head < queue.size(); ++head) {
        const int k = queue[head];
        const Node& nd = t.nodes[k];
        const int x = (int)gather3(nd.code), y = (int)gather3(nd.code >> 1), z = (int)gather3(nd.code >> 2);
        float* v = voxels + 4 * (size_t)k;
        v[0] = (float)x; v[1] = (float)y; v[2] = (float)z; v[3] = (float)nd.side;
        if (nd.type == T_SURFACE) {
            for (int i = 0; i < 8; ++i) {
                // the reference looks the corner up through float coordinates (octree.cpp:319-325)
                const int q = finder.find((int)(v[0] + (float)DX[i]), (int)(v[1] + (float)DY[i]), (int)(v[2] + (float)DZ[i]));
                if (q >= 0) features[8 * (size_t)k + i] = q;
            }
        }
        for (int i = 0; i < 8; ++i) {
            const int c = nd.child[i];
            if (c >= 0 && t.nodes[c].type != T_FEATURE) {
                queue.push_back(c);
                children[8 * (size_t)k + i] = (float)c;
            }
        }
    }
    return 0;
}

// Fused export for the kernels (mapping.py:319-327 folded in): centres[n,3] = (xyz + side/2)*voxel_size,
// structure[n,9] = [children(int), side], vertex_idx[n,8].
int nl_octree_export_device_layout(void* h, float voxel_size, float* centres, int* structure, int* vertex_idx)
{
    if (!h || !centres || !structure || !vertex_idx) return 1;
    const Octree& t = *(Octree*)h;
    const size_t n = t.nodes.size();
    std::vector<float> vox(n * 4), ch(n * 8);
    int rc = nl_octree_export(h, vox.data(), ch.data(), vertex_idx);
    if (rc) return rc;
    for (size_t k = 0; k < n; ++k) {
        const float side = vox[4 * k + 3];
        for (int a = 0; a < 3; ++a) centres[3 * k + a] = (vox[4 * k + a] + side / 2.0f) * voxel_size;
        for (int i = 0; i < 8; ++i) structure[9 * k + i] = (int)ch[8 * k + i];
        structure[9 * k + 8] = (int)side;
    }
    return 0;
}

// Incremental export (the reference re-exports and re-uploads the whole tree every frame,
// mapping.py:283-339): the rows - in the nl_octree_export_device_layout format - of exactly the nodes whose rows changed
// since the previous call (or since creation).  Rows of untouched nodes are unchanged by construction: node ids are
// creation-ordered, a SURFACE leaf's eight corner leaves exist from the moment it is inserted, and a node's row depends
// only on itself and on the types of its children.  Applying the deltas in order to arrays grown to count_nodes() rows
// reproduces the full export bit for bit (tests/test_octree_host.py).
long long nl_octree_delta_count(void* h) { return h ? (long long)((Octree*)h)->dirty.size() : 0; }

int nl_octree_export_delta(void* h, float voxel_size, int* ids, float* centres, int* structure, int* vertex_idx)
{
    if (!h || !ids || !centres || !structure || !vertex_idx) return 1;
    Octree& t = *(Octree*)h;
    const size_t nd_ = t.dirty.size();
    LeafFinder finder(t);
    for (size_t q = 0; q < nd_; ++q) {
        const int k = t.dirty[q];
        const Node& nd = t.nodes[k];
        ids[q] = k;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < 8; ++i) { structure[9 * q + i] = -1; vertex_idx[8 * q + i] = -1; }
        if (nd.type != T_FEATURE) {                                     // FEATURE leaves are never reached by the export walk
            v[0] = (float)(int)gather3(nd.code); v[1] = (float)(int)gather3(nd.code >> 1); v[2] = (float)(int)gather3(nd.code >> 2);
            v[3] = (float)nd.side;
            if (nd.type == T_SURFACE) {
                for (int i = 0; i < 8; ++i) {
                    const int c = finder.find((int)(v[0] + (float)DX[i]), (int)(v[1] + (float)DY[i]), (int)(v[2] + (float)DZ[i]));
                    if (c >= 0) vertex_idx[8 * q + i] = c;
                }
            }
            for (int i = 0; i < 8; ++i) {
                const int c = nd.child[i];
                if (c >= 0 && t.nodes[c].type != T_FEATURE) structure[9 * q + i] = (int)(float)c;
            }
        }
        for (int a = 0; a < 3; ++a) centres[3 * q + a] = (v[a] + v[3] / 2.0f) * voxel_size;
        structure[9 * q + 8] = (int)v[3];
        t.dirty_flag[k] = 0;
    }
    t.dirty.clear();
    return 0;
}

// Children-block traversal layout of nl_ray_intersect straight from the tree (what nerf_loam_amd/pipeline.py pack_children_blocks derives from the
// exported `structure` rows with ~200 torch launches per map update): block 0 = pseudo block with the root in slot 0, every node with a LISTED child
// (present and not a FEATURE leaf: the children column of the export) owns one block, blocks are numbered breadth-first so that the children blocks of
// a block are consecutive in octant order.  blk_ids[B][8] = child node ids (-1 = none), blk_hdr[B][2] = (first child block or -1, exist mask |
// owns-a-block mask << 8); the pseudo block's slots 1..7 carry the single-child chain under the root (length, octants packed 3 bits each in two
// words, the block the work-list starts with, its lattice position).  Returns the number of blocks, or -(blocks needed) if `capacity` is too small
// (nothing written beyond capacity), 0 on a bad handle.  blk_ids / blk_hdr == NULL: count only.
long long nl_octree_pack_blocks(void* h, int* blk_ids, int* blk_hdr, long long capacity)
{
    if (!h) return 0;
    const Octree& t = *(Octree*)h;
    auto listed = [&](int c) { return c >= 0 && t.nodes[c].type != T_FEATURE; };
    auto interior = [&](int k) { const Node& nd = t.nodes[k]; for (int i = 0; i < 8; ++i) if (listed(nd.child[i])) return true; return false; };
    const bool has_root = interior(0);
    const bool write = blk_ids && blk_hdr;
    std::vector<int> frontier, next;
    if (has_root) frontier.push_back(0);
    long long next_index = 1;
    if (write && capacity >= 1) {
        for (int i = 0; i < 8; ++i) blk_ids[i] = -1;
        blk_ids[0] = 0;
        blk_hdr[0] = has_root ? 1 : -1; blk_hdr[1] = has_root ? 0x101 : 1;
    }
    while (!frontier.empty()) {
        const long long F = (long long)frontier.size();
        long long running = next_index + F;                              // the blocks of this level come first, then their children's
        next.clear();
        for (long long q = 0; q < F; ++q) {
            const Node& nd = t.nodes[frontier[(size_t)q]];
            const long long b = next_index + q;
            int exist = 0, owns = 0, cnt = 0, ids[8];
            for (int i = 0; i < 8; ++i) {
                const int c = nd.child[i];
                ids[i] = -1;
                if (listed(c)) {
                    exist |= 1 << i; ids[i] = c;
                    if (interior(c)) { owns |= 1 << i; ++cnt; next.push_back(c); }
                }
            }
            if (write && b < capacity) {
                for (int i = 0; i < 8; ++i) blk_ids[8 * b + i] = ids[i];
                blk_hdr[2 * b] = cnt > 0 ? (int)running : -1;
                blk_hdr[2 * b + 1] = exist | (owns << 8);
            }
            running += cnt;
        }
        next_index += F;
        frontier.swap(next);
    }
    const long long B = next_index;
    if (write && B > capacity) return -B;
    if (write && has_root) {
        unsigned cs = t.nodes[0].side >> 1;
        long long b = 1;
        int pos[3] = {0, 0, 0}, n_oct = 0;
        unsigned long long packed = 0;
        while (cs > 1 && b < B && b < 40 && n_oct < 20) {
            const int has = (blk_hdr[2 * b + 1] >> 8) & 255, exist = blk_hdr[2 * b + 1] & 255;
            if (has == 0 || (has & (has - 1)) || exist != has) break;   // no child block, several, or a child without a block of its own
            const int u = 31 - __builtin_clz((unsigned)has);
            packed |= (unsigned long long)u << (3 * n_oct);
            ++n_oct;
            if (u & 1) pos[0] += (int)cs;
            if (u & 2) pos[1] += (int)cs;
            if (u & 4) pos[2] += (int)cs;
            b = blk_hdr[2 * b]; cs >>= 1;
        }
        blk_ids[1] = n_oct; blk_ids[2] = (int)(packed & 0x3FFFFFFFULL); blk_ids[3] = (int)(packed >> 30); blk_ids[4] = (int)b;
        blk_ids[5] = pos[0]; blk_ids[6] = pos[1]; blk_ids[7] = pos[2];
    }
    return B;
}

}  // extern "C"
