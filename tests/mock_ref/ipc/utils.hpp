// TEST MOCK (not product code, not the reference's header): the slice of g2o and of the reference's
// include/ipc/utils.hpp that include/ipc/consensus_amd.hpp touches, so that the adapter can be
// compiled and exercised where g2o / Eigen are not installed.
#pragma once
#include <algorithm>
#include <array>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace g2o {
struct Vertex {
    int _id = 0; bool _fixed = false;
    int id() const { return _id; }
    void setFixed(bool f) { _fixed = f; }
};
template <int D> struct Mat {
    double v[D][D] = {};
    double operator()(int i, int j) const { return v[i][j]; }
    Mat operator*(double s) const { Mat r; for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) r.v[i][j] = v[i][j] * s; return r; }
    Mat operator/(double s) const { Mat r; for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) r.v[i][j] = v[i][j] / s; return r; }
};
// (poses only need to compose somehow for propagateGuess to compile and leave a trace: the mock adds component-wise)
struct SE2 {
    std::array<double, 3> m{};
    SE2() = default;
    SE2(double x, double y, double th) : m{{x, y, th}} {}
    std::array<double, 3> toVector() const { return m; }
    SE2 operator*(const SE2& o) const { SE2 r; for (int k = 0; k < 3; ++k) r.m[k] = m[k] + o.m[k]; return r; }
};
struct Isometry3 {
    std::array<double, 7> qt{};                                    // x y z qx qy qz qw (what the file holds: measurements)
    double mat[4][4] = {};                                         // what the adapter's write-back fills (Eigen: t(row, col))
    static Isometry3 Identity() { Isometry3 t; for (int k = 0; k < 4; ++k) t.mat[k][k] = 1.0; return t; }
    double& operator()(int r, int c) { return mat[r][c]; }
    double operator()(int r, int c) const { return mat[r][c]; }
    Isometry3 operator*(const Isometry3& o) const { Isometry3 r; for (int k = 0; k < 7; ++k) r.qt[k] = qt[k] + o.qt[k]; return r; }
};
namespace internal { inline std::array<double, 7> toVectorQT(const Isometry3& t) { return t.qt; } }
struct OptimizableGraph {
    struct Edge {
        std::vector<Vertex*> _v{nullptr, nullptr};
        const std::vector<Vertex*>& vertices() const { return _v; }
        virtual ~Edge() = default;
    };
    // g2o 20201223: HyperGraph::EdgeSet is a std::set of edge pointers -- the graph iterates its edges in ADDRESS order,
    // not in file order, and so do splitProblemConstraints / getProblemOdom / getProblemLoops
    using EdgeSet = std::set<Edge*>;
};
template <int D, class MEAS> struct Edge : OptimizableGraph::Edge {
    static const int Dimension = D;
    MEAS _m; Mat<D> _info;
    const MEAS& measurement() const { return _m; }
    const Mat<D>& information() const { return _info; }
    void setInformation(const Mat<D>& i) { _info = i; }
};
using EdgeSE2 = Edge<3, SE2>;
using EdgeSE3 = Edge<6, Isometry3>;
template <class POSE> struct PoseVertex : Vertex {
    POSE _est;
    const POSE& estimate() const { return _est; }
    void setEstimate(const POSE& p) { _est = p; }
    void setToOrigin() { _est = POSE(); }
};
using VertexSE2 = PoseVertex<SE2>;
using VertexSE3 = PoseVertex<Isometry3>;
struct SparseOptimizer {
    std::map<int, Vertex*> _vertices;
    OptimizableGraph::EdgeSet _edges;
    const OptimizableGraph::EdgeSet& edges() const { return _edges; }
    const std::map<int, Vertex*>& vertices() const { return _vertices; }
    Vertex* vertex(int id) { return _vertices.at(id); }
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
// the walks of reference src/utils.cpp:197-231 over the mock graph
template <class EDGE> void getProblemOdom(g2o::SparseOptimizer& p, std::vector<EDGE*>& odom)
{
    for (g2o::OptimizableGraph::Edge* e : p.edges()) {
        EDGE* q = dynamic_cast<EDGE*>(e);
        if (q && std::abs(q->vertices()[1]->id() - q->vertices()[0]->id()) == 1) odom.push_back(q);
    }
}
template <class EDGE> void getProblemLoops(g2o::SparseOptimizer& p, std::vector<EDGE*>& loops)
{
    for (g2o::OptimizableGraph::Edge* e : p.edges()) {
        EDGE* q = dynamic_cast<EDGE*>(e);
        if (q && std::abs(q->vertices()[1]->id() - q->vertices()[0]->id()) > 1) loops.push_back(q);
    }
}
inline bool cmpEdgesID(g2o::OptimizableGraph::Edge* e1, g2o::OptimizableGraph::Edge* e2)   // reference src/utils.cpp:366-369
{
    return e1->vertices()[1]->id() < e2->vertices()[1]->id();
}
inline int mock_last_vertex(const g2o::OptimizableGraph::Edge* e) { return std::max(e->vertices()[0]->id(), e->vertices()[1]->id()); }
inline bool cmpTime(std::pair<int, g2o::OptimizableGraph::Edge*> a, std::pair<int, g2o::OptimizableGraph::Edge*> b)   // src/utils.cpp:379-390
{
    return mock_last_vertex(a.second) < mock_last_vertex(b.second);
}
