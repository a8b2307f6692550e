// Compiles (and, on a GPU box, runs) include/ipc/consensus_amd.hpp against the g2o mock:
//   adapter_main <dim> <spoiled.g2o> s fast_th fast_it slow_th slow_it   -> prints the per-candidate
// decisions of the reference's harness loop (agreementCheck in cmpTime order) and the set size.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "ipc/consensus_amd.hpp"

template <class EDGE, class VERTEX, int MS, int D>
static int run(const char* path, const Config& cfg)
{
    g2o::SparseOptimizer problem;
    std::vector<VERTEX*> verts;
    std::vector<EDGE*> edges;
    std::ifstream in(path);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tag;
        ss >> tag;
        if (tag.rfind("VERTEX", 0) == 0) {
            VERTEX* v = new VERTEX();
            ss >> v->_id;
            verts.push_back(v);
            problem._vertices[v->_id] = v;
        } else if (tag.rfind("EDGE", 0) == 0) {
            EDGE* e = new EDGE();
            int a, b;
            ss >> a >> b;
            e->_v[0] = problem._vertices.at(a);
            e->_v[1] = problem._vertices.at(b);
            double m[7];
            for (int k = 0; k < MS; ++k) ss >> m[k];
            if (MS == 3) for (int k = 0; k < 3; ++k) reinterpret_cast<double*>(&e->_m)[k] = m[k];
            else for (int k = 0; k < 7; ++k) reinterpret_cast<double*>(&e->_m)[k] = m[k];
            for (int i = 0; i < D; ++i)
                for (int j = i; j < D; ++j) { ss >> e->_info.v[i][j]; e->_info.v[j][i] = e->_info.v[i][j]; }
            edges.push_back(e);
            problem._edges.push_back(e);
        }
    }
    std::vector<EDGE*> loops;
    for (EDGE* e : edges)
        if (std::abs(e->vertices()[1]->id() - e->vertices()[0]->id()) > 1) loops.push_back(e);   // src/utils.cpp:172-189
    std::vector<EDGE*> order = loops;                                                          // cmpTime, stable
    std::stable_sort(order.begin(), order.end(), [](EDGE* a, EDGE* b) {
        return std::max(a->vertices()[0]->id(), a->vertices()[1]->id()) < std::max(b->vertices()[0]->id(), b->vertices()[1]->id());
    });
    {
        IPC<EDGE, VERTEX> ipc(problem, cfg);
        ipc.setCandidates(loops);
        std::printf("decisions");
        for (EDGE* e : order) std::printf(" %d", ipc.agreementCheck(e) ? 1 : 0);
        std::printf("\nset %zu\n", ipc.getMaxConsensusSet().size());
        if (!ipc.getMaxConsensusSet().empty()) {
            EDGE* first = ipc.getMaxConsensusSet().front();
            const bool removed = ipc.removeEdgeFromCnS(first);
            std::printf("removed %d -> %zu\n", removed ? 1 : 0, ipc.getMaxConsensusSet().size());
            ipc.addEdgeToCnS(first);
            std::printf("added -> %zu\n", ipc.getMaxConsensusSet().size());
        }
        const std::vector<char> all = ipc.agreementCheckAll(loops);
        int n = 0;
        for (char c : all) n += c;
        std::printf("matrix set %d\n", n);
    }
    std::printf("cleared %zu\n", problem.vertices().size());
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
