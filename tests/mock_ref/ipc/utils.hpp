// TEST MOCK (not product code, not the reference's header): the slice of g2o and of the reference's
// include/ipc/utils.hpp that include/ipc/consensus_amd.hpp touches, so that the adapter can be
// compiled and exercised where g2o / Eigen are not installed.
#pragma once
#include <array>
#include <map>
#include <string>
#include <vector>

namespace g2o {
struct Vertex { int _id = 0; int id() const { return _id; } };
template <int D> struct Mat { double v[D][D] = {}; double operator()(int i, int j) const { return v[i][j]; } };
struct SE2 { std::array<double, 3> m{}; std::array<double, 3> toVector() const { return m; } };
struct Isometry3 { std::array<double, 7> qt{}; };                 // x y z qx qy qz qw
namespace internal { inline std::array<double, 7> toVectorQT(const Isometry3& t) { return t.qt; } }
struct OptimizableGraph {
    struct Edge {
        std::vector<Vertex*> _v{nullptr, nullptr};
        const std::vector<Vertex*>& vertices() const { return _v; }
    };
};
template <int D, class MEAS> struct Edge : OptimizableGraph::Edge {
    static const int Dimension = D;
    MEAS _m; Mat<D> _info;
    const MEAS& measurement() const { return _m; }
    const Mat<D>& information() const { return _info; }
};
using EdgeSE2 = Edge<3, SE2>;
using EdgeSE3 = Edge<6, Isometry3>;
struct VertexSE2 : Vertex {};
struct VertexSE3 : Vertex {};
struct SparseOptimizer {
    std::map<int, Vertex*> _vertices;
    std::vector<void*> _edges;
    const std::map<int, Vertex*>& vertices() const { return _vertices; }
    void clear() { _vertices.clear(); _edges.clear(); }
};
}  // namespace g2o

struct Config {                                                   // reference include/ipc/utils.hpp:22-38
    std::string name, dataset, ground_truth, output;
    bool visualize = false;
    int canonic_inliers = 0;
    double s_factor = 1.0, fast_reject_th = 0, slow_reject_th = 0;
    int fast_reject_iter_base = 0, slow_reject_iter_base = 0;
    bool use_best_k_buddies = false; int k_buddies = 0; bool use_recovery = false;
};
template <class EDGE> void getProblemOdom(g2o::SparseOptimizer& p, std::vector<EDGE*>& odom)
{
    for (void* e : p._edges) {
        EDGE* q = static_cast<EDGE*>(e);
        if (q->vertices()[1]->id() - q->vertices()[0]->id() == 1) odom.push_back(q);
    }
}
inline bool cmpEdgesID(g2o::OptimizableGraph::Edge* e1, g2o::OptimizableGraph::Edge* e2)   // reference src/utils.cpp:366-369
{
    return e1->vertices()[1]->id() < e2->vertices()[1]->id();
}
