// Compiles (and, on a GPU box, runs) include/ipc/consensus_amd.hpp against the g2o mock -- through the literal
// `#include "ipc/consensus.hpp"` of the reference's src/simulation.cpp:1 (this repo's include/ in front on the path) --
// and replays the sequence of src/simulation.cpp:24-56 without copying it:
//   adapter_main <dim> <spoiled.g2o> s fast_th fast_it slow_th slow_it
// Lines printed (tests/test_adapter.py reads them):
//   ctor_info_scale <max |info_after_ctor / (s * info_file) - 1| over the odometry edges>
//   decisions <0/1 per candidate in cmpTime order>            harness path: the candidates are never announced
//   set <size>
//   harness_info_restore <max |info_after_the_harness_divides / info_file - 1|>
//   harness_vertices_propagated <1 if the harness's propagateGuess ran on the caller's vertices>
//   announced <0/1 ...>                                       same loop after setCandidates(loops): must be equal
//   removed / added / matrix set / cleared                    set editing, batched mode, destructor
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "ipc/consensus.hpp"

template <class EDGE, class VERTEX, int MS, int D>
struct Loaded {
    g2o::SparseOptimizer problem;
    std::vector<EDGE*> edges, loops, order;
    std::vector<g2o::Mat<D>> file_info;                            // of the odometry edges, as the file has it
    std::vector<EDGE*> odom;
    explicit Loaded(const char* path)
    {
        std::ifstream in(path);
        std::string line;
        while (std::getline(in, line)) {
            std::istringstream ss(line);
            std::string tag;
            ss >> tag;
            if (tag.rfind("VERTEX", 0) == 0) {
                VERTEX* v = new VERTEX();
                ss >> v->_id;
                problem._vertices[v->_id] = v;
            } else if (tag.rfind("EDGE", 0) == 0) {
                EDGE* e = new EDGE();
                int a, b;
                ss >> a >> b;
                e->_v[0] = problem._vertices.at(a);
                e->_v[1] = problem._vertices.at(b);
                double m[7];
                for (int k = 0; k < MS; ++k) ss >> m[k];
                for (int k = 0; k < MS; ++k) reinterpret_cast<double*>(&e->_m)[k] = m[k];
                for (int i = 0; i < D; ++i)
                    for (int j = i; j < D; ++j) { ss >> e->_info.v[i][j]; e->_info.v[j][i] = e->_info.v[i][j]; }
                edges.push_back(e);
                problem._edges.push_back(e);
            }
        }
        for (EDGE* e : edges)
            if (std::abs(e->vertices()[1]->id() - e->vertices()[0]->id()) > 1) loops.push_back(e);   // src/utils.cpp:172-189
        order = loops;                                                                            // cmpTime, stable
        std::stable_sort(order.begin(), order.end(), [](EDGE* a, EDGE* b) {
            return std::max(a->vertices()[0]->id(), a->vertices()[1]->id()) < std::max(b->vertices()[0]->id(), b->vertices()[1]->id());
        });
        getProblemOdom<EDGE>(problem, odom);
        for (EDGE* e : odom) file_info.push_back(e->information());
    }
    double info_ratio_error(double scale) const
    {
        double worst = 0.0;
        for (size_t k = 0; k < odom.size(); ++k)
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) {
                    const double want = file_info[k](i, j) * scale, got = odom[k]->information()(i, j);
                    if (want != 0.0) worst = std::max(worst, std::fabs(got / want - 1.0));
                    else if (got != 0.0) worst = 1.0;
                }
        return worst;
    }
};

template <class EDGE, class VERTEX, int MS, int D>
static int run(const char* path, const Config& cfg)
{
    std::vector<int> harness_decisions;
    {   // ---- the harness's own sequence (src/simulation.cpp:28-56): candidates handed over one by one, never announced
        Loaded<EDGE, VERTEX, MS, D> L(path);
        {
            IPC<EDGE, VERTEX> ipc(L.problem, cfg);                                               // :28
            std::printf("ctor_info_scale %.3e\n", L.info_ratio_error(cfg.s_factor));
            std::printf("decisions");
            for (EDGE* e : L.order) {                                                            // :34-47
                const bool ok = ipc.agreementCheck(e);
                harness_decisions.push_back(ok ? 1 : 0);
                std::printf(" %d", ok ? 1 : 0);
            }
            std::printf("\nset %zu\n", ipc.getMaxConsensusSet().size());
            std::vector<EDGE*> odom_edges;
            getProblemOdom<EDGE>(L.problem, odom_edges);                                         // :50-52
            propagateGuess<EDGE, VERTEX>(L.problem, 0, (int)odom_edges.size(), odom_edges);
            for (size_t i = 0; i < odom_edges.size(); ++i)                                       // :55-56
                odom_edges[i]->setInformation(odom_edges[i]->information() / cfg.s_factor);
            std::printf("harness_info_restore %.3e\n", L.info_ratio_error(1.0));
            const VERTEX* last = static_cast<VERTEX*>(L.problem.vertex((int)odom_edges.size()));
            double trace = 0.0;
            for (int k = 0; k < MS; ++k) trace += std::fabs(reinterpret_cast<const double*>(&last->estimate())[k]);
            std::printf("harness_vertices_propagated %d\n", trace > 0.0 ? 1 : 0);
        }
    }
    {   // ---- the same loop with the candidate list announced first, then the rest of the class surface
        Loaded<EDGE, VERTEX, MS, D> L(path);
        {
            IPC<EDGE, VERTEX> ipc(L.problem, cfg);
            ipc.setCandidates(L.loops);
            std::printf("announced");
            for (EDGE* e : L.order) std::printf(" %d", ipc.agreementCheck(e) ? 1 : 0);
            std::printf("\n");
            if (!ipc.getMaxConsensusSet().empty()) {
                EDGE* first = ipc.getMaxConsensusSet().front();
                EDGE twin = *first;                                // another edge object joining the same vertices
                const bool removed = ipc.removeEdgeFromCnS(&twin); // the reference matches by ids (src/consensus.cpp:84-87)
                std::printf("removed %d -> %zu\n", removed ? 1 : 0, ipc.getMaxConsensusSet().size());
                ipc.addEdgeToCnS(first);
                std::printf("added -> %zu\n", ipc.getMaxConsensusSet().size());
            }
            const std::vector<char> all = ipc.agreementCheckAll(L.loops);
            int n = 0;
            for (char c : all) n += c;
            std::printf("matrix set %d\n", n);
        }
        std::printf("cleared %zu\n", L.problem.vertices().size());
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 8) { std::fprintf(stderr, "usage: adapter_main dim file s fth fit sth sit\n"); return 2; }
    Config cfg;
    cfg.s_factor = std::atof(argv[3]);
    cfg.fast_reject_th = std::atof(argv[4]); cfg.fast_reject_iter_base = std::atoi(argv[5]);
    cfg.slow_reject_th = std::atof(argv[6]); cfg.slow_reject_iter_base = std::atoi(argv[7]);
    try {
        return std::atoi(argv[1]) == 2 ? run<g2o::EdgeSE2, g2o::VertexSE2, 3, 3>(argv[2], cfg)
                                       : run<g2o::EdgeSE3, g2o::VertexSE3, 7, 6>(argv[2], cfg);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
