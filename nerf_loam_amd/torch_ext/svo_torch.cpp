// svo_torch.cpp -- registers the reference's TorchScript names on top of the C ABI of libnerfloam_hip.so:
//     torch.classes.svo.Octree   (third_party/sparse_octree/src/bindings.cpp:10-31: init, insert, try_insert, get_voxels,
//                                 get_leaf_voxels, get_features, count_nodes, count_leaf_nodes, has_voxel,
//                                 get_centres_and_children, pickle state (size, feat_dim, voxel_size, inserted tensors))
//     torch.classes.svo.Octant   (bindings.cpp:7-8: constructible, no methods)
//     torch.ops.svo.encode       (bindings.cpp:6, include/test.h:75-86)
// so that the reference's `torch.classes.load_library(<path>)` + `torch.classes.svo.Octree()` (src/mapping.py:19-20,81-82)
// works with only the PATH changed.  Host-side boundary code: tensors in and out are CPU torch tensors like the reference's,
// the octree itself is nl_octree_* (csrc/nl_octree.cpp, flat index-linked arrays, per-instance node counter).
// Built by nerf_loam_amd/build.py into nerf_loam_amd/libnl_svo_torch.so (g++, torch headers; no GPU code in this file).
#include <torch/custom_class.h>
#include <torch/script.h>

#include <tuple>
#include <vector>

#include "nerfloam_hip.h"

namespace {

struct Octant : torch::CustomClassHolder {};

struct Octree : torch::CustomClassHolder {
    void* h = nullptr;
    int64_t size_ = 0, feat_dim_ = 0;
    double voxel_size_ = 0.0;
    std::vector<torch::Tensor> all_pts;

    Octree() {}
    Octree(int64_t grid_dim, int64_t feat_dim, double voxel_size, std::vector<torch::Tensor> pts)
    {
        init(grid_dim, feat_dim, voxel_size);
        for (auto& p : pts) insert(p);
    }
    ~Octree() override { if (h) nl_octree_destroy(h); }

    void need() const { TORCH_CHECK(h != nullptr, "Octree not initialized!"); }

    void init(int64_t grid_dim, int64_t feat_dim, double voxel_size)
    {
        if (h) nl_octree_destroy(h);
        h = nl_octree_create((long long)grid_dim);
        TORCH_CHECK(h != nullptr, "Octree.init: grid_dim must be > 1");
        size_ = grid_dim; feat_dim_ = feat_dim; voxel_size_ = voxel_size;
        all_pts.clear();
    }

    static torch::Tensor as_i32_points(const torch::Tensor& t, const char* what)
    {
        TORCH_CHECK(t.dim() == 2 && t.size(1) == 3, what, ": point dimensions mismatch, expect [M,3]");
        return t.to(torch::kCPU, torch::kInt32).contiguous();
    }

    void insert(torch::Tensor pts)
    {
        need();
        torch::Tensor a = as_i32_points(pts, "insert");
        all_pts.push_back(a.clone());
        TORCH_CHECK(nl_octree_insert(h, a.data_ptr<int>(), (long long)a.size(0)) == 0, "nl_octree_insert failed");
    }

    double try_insert(torch::Tensor pts)
    {
        need();
        torch::Tensor a = as_i32_points(pts, "try_insert");
        return nl_octree_try_insert(h, a.data_ptr<int>(), (long long)a.size(0));
    }

    torch::Tensor get_voxels()
    {
        need();
        torch::Tensor out = torch::empty({(int64_t)nl_octree_count_nodes(h), 4}, torch::kFloat32);
        TORCH_CHECK(nl_octree_voxels_dfs(h, out.data_ptr<float>()) == 0, "nl_octree_voxels_dfs failed");
        return out;
    }

    torch::Tensor get_leaf_voxels()
    {
        need();
        const int64_t n = (int64_t)nl_octree_leaf_voxels(h, nullptr);
        torch::Tensor out = torch::empty({n, 3}, torch::kFloat32);
        if (n) nl_octree_leaf_voxels(h, out.data_ptr<float>());
        return out;
    }

    torch::Tensor get_features(torch::Tensor)
    {
        TORCH_CHECK(false, "Octree::get_features has an empty body in the reference (octree.cpp:208-210)");
        return torch::Tensor();
    }

    int64_t count_nodes() { need(); return (int64_t)nl_octree_count_nodes(h); }
    int64_t count_leaf_nodes() { need(); return (int64_t)nl_octree_count_leaf_nodes(h); }

    bool has_voxel(torch::Tensor pt)
    {
        need();
        torch::Tensor a = pt.to(torch::kCPU, torch::kInt32).contiguous().reshape({-1});
        if (a.numel() != 3) return false;
        const int* p = a.data_ptr<int>();
        return nl_octree_has_voxel(h, p[0], p[1], p[2]) != 0;
    }

    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> get_centres_and_children()
    {
        need();
        const int64_t n = (int64_t)nl_octree_count_nodes(h);
        torch::Tensor vox = torch::empty({n, 4}, torch::kFloat32), ch = torch::empty({n, 8}, torch::kFloat32);
        torch::Tensor ft = torch::empty({n, 8}, torch::kInt32);
        TORCH_CHECK(nl_octree_export(h, vox.data_ptr<float>(), ch.data_ptr<float>(), ft.data_ptr<int>()) == 0, "nl_octree_export failed");
        return std::make_tuple(vox, ch, ft);
    }
};

// include/test.h:75-86 - the z slot of the key is filled from the x column there (coords[3 i] is read twice); kept
inline int64_t expand21(int64_t v)
{
    uint64_t x = (uint64_t)v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return (int64_t)x;
}

torch::Tensor encode_torch(torch::Tensor coords)
{
    TORCH_CHECK(coords.dim() == 2 && coords.size(1) == 3 && coords.scalar_type() == torch::kInt64, "encode expects an int64 tensor [K,3]");
    torch::Tensor c = coords.to(torch::kCPU).contiguous();
    torch::Tensor out = torch::zeros({c.size(0), 1}, torch::kInt64);
    const int64_t* p = c.data_ptr<int64_t>();
    for (int64_t i = 0; i < c.size(0); ++i) {
        const int64_t x = p[3 * i], y = p[3 * i + 1], z = p[3 * i];
        out.data_ptr<int64_t>()[i] = (expand21(x) | (expand21(y) << 1) | (expand21(z) << 2)) & 0x7fffffffffffffffll;
    }
    return out;
}

}  // namespace

TORCH_LIBRARY(svo, m)
{
    m.def("encode", &encode_torch);
    m.class_<Octant>("Octant").def(torch::init<>());
    m.class_<Octree>("Octree")
        .def(torch::init<>())
        .def("init", &Octree::init)
        .def("insert", &Octree::insert)
        .def("try_insert", &Octree::try_insert)
        .def("get_voxels", &Octree::get_voxels)
        .def("get_leaf_voxels", &Octree::get_leaf_voxels)
        .def("get_features", &Octree::get_features)
        .def("count_nodes", &Octree::count_nodes)
        .def("count_leaf_nodes", &Octree::count_leaf_nodes)
        .def("has_voxel", &Octree::has_voxel)
        .def("get_centres_and_children", &Octree::get_centres_and_children)
        .def_pickle(
            [](const c10::intrusive_ptr<Octree>& self) -> std::tuple<int64_t, int64_t, double, std::vector<torch::Tensor>> {
                return std::make_tuple(self->size_, self->feat_dim_, self->voxel_size_, self->all_pts);
            },
            [](std::tuple<int64_t, int64_t, double, std::vector<torch::Tensor>> state) {
                return c10::make_intrusive<Octree>(std::get<0>(state), std::get<1>(state), std::get<2>(state), std::get<3>(state));
            });
}
