// Compiles (and, on a GPU box, runs) include/ipc/consensus_amd.hpp against the g2o mock -- through the literal
// `#include "ipc/consensus.hpp"` of the reference's src/simulation.cpp:1 (this repo's include/ in front on the path) --
// and replays the sequence of src/simulation.cpp:24-56 without copying it:
//   adapter_main <dim> <spoiled.g2o> s fast_th fast_it slow_th slow_it [order] [where] [canonic_inliers]
//     order  ref     the harness's own: std::sort with cmpTime over the loops as the graph iterates them (default)
//            stable  ascending last vertex, ties in FILE order (the order the committed oracle fixtures were taken in)
//     where  graph   the candidates are edges of the graph, as in the reference's testers (default)
//            foreign the graph holds the odometry only; every candidate is an edge object the engine has never seen
//     canonic_inliers   cfg.canonic_inliers (only labels; default 0)
// Lines printed (tests/test_adapter.py reads them):
//   ctor_info_scale <max |info_after_ctor / (s * info_file) - 1| over the odometry edges>
//   order <index among the file's loop edges, per call>       the order the harness called in
//   decisions <0/1 per call>                                  harness path: the candidates are never announced
//   set <size>
//   seconds <wall time of the agreementCheck loop> rate <candidates per second>      (src/simulation.cpp:36-44,87)
//   written_back <max |vertex estimate - engine pose| after the loop, write-back on>  (only with IPC_ADAPTER_WRITE_BACK=1)
//   harness_info_restore <max |info_after_the_harness_divides / info_file - 1|>
//   harness_vertices_propagated <1 if the harness's propagateGuess ran on the caller's vertices>
//   announced_order / announced <0/1 ...>                     the same loop on a fresh graph after setCandidates(loops)
//   removed / added / matrix set / cleared                    set editing, batched mode, destructor
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "ipc/consensus.hpp"

template <class EDGE, class VERTEX, int MS, int D>
struct Loaded {
    g2o::SparseOptimizer problem;
    std::vector<EDGE*> edges, loops_in_file;                       // file order
    std::map<EDGE*, int> file_index;                               // loop edge -> its index among the file's loop edges
    std::vector<g2o::Mat<D>> file_info;                            // of the odometry edges, as the file has it
    std::vector<EDGE*> odom;
    Loaded(const char* path, bool loops_in_graph)
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
                const bool loop = std::abs(b - a) > 1;             // src/utils.cpp:184
                if (loop) { file_index[e] = (int)loops_in_file.size(); loops_in_file.push_back(e); }
                if (!loop || loops_in_graph) problem._edges.insert(e);
            }
        }
        getProblemOdom<EDGE>(problem, odom);
        std::sort(odom.begin(), odom.end(), cmpEdgesID);
        for (EDGE* e : odom) file_info.push_back(e->information());
    }
    // the candidates in the order the harness hands them over (src/simulation.cpp:24-26)
    std::vector<EDGE*> call_order(bool ref_order, int canonic_inliers)
    {
        std::vector<EDGE*> out;
        if (ref_order) {
            std::vector<EDGE*> loops;
            if (problem.edges().size() > odom.size()) getProblemLoops<EDGE>(problem, loops);   // the tester's splitProblemConstraints
            else loops = loops_in_file;                                                      // (candidates kept outside the graph)
            std::vector<std::pair<bool, EDGE*>> gt;
            for (size_t i = 0; i < loops.size(); ++i) gt.push_back(std::make_pair((int)i < canonic_inliers, loops[i]));
            std::sort(gt.begin(), gt.end(), cmpTime);
            for (auto& q : gt) out.push_back(q.second);
        } else {
            out = loops_in_file;
            std::stable_sort(out.begin(), out.end(), [](EDGE* a, EDGE* b) { return mock_last_vertex(a) < mock_last_vertex(b); });
        }
        return out;
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

inline double estimate_component(const g2o::SE2& e, int i) { return e.m[i]; }
inline double estimate_component(const g2o::Isometry3& e, int i) { return i < 9 ? e(i / 3, i % 3) : e(i - 9, 3); }

template <class EDGE, class VERTEX, int MS, int D>
static int run(const char* path, Config cfg, bool ref_order, bool in_graph)
{
    {   // ---- the harness's own sequence (src/simulation.cpp:28-56): candidates handed over one by one, never announced
        Loaded<EDGE, VERTEX, MS, D> L(path, in_graph);
        const std::vector<EDGE*> order = L.call_order(ref_order, cfg.canonic_inliers);
        {
            IPC<EDGE, VERTEX> ipc(L.problem, cfg);                                               // :28
            const char* wb = std::getenv("IPC_ADAPTER_WRITE_BACK");
            const bool write_back = wb && *wb && std::strcmp(wb, "0");
            ipc.setWriteBackEstimates(write_back);
            std::printf("ctor_info_scale %.3e\n", L.info_ratio_error(cfg.s_factor));
            std::printf("order");
            for (EDGE* e : order) std::printf(" %d", L.file_index.at(e));
            std::printf("\n");
            std::vector<char> dec;
            const auto t0 = std::chrono::steady_clock::now();
            for (EDGE* e : order) dec.push_back(ipc.agreementCheck(e) ? 1 : 0);                  // :34-47
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("decisions");
            for (char d : dec) std::printf(" %d", (int)d);
            std::printf("\nset %zu\n", ipc.getMaxConsensusSet().size());
            std::printf("seconds %.6f rate %.1f\n", dt, order.empty() ? 0.0 : order.size() / dt);
            if (write_back) {
                std::vector<double> poses;
                ipc.currentPoses(poses);
                const int ps = D == 3 ? 3 : 12;
                double worst = 0.0;
                for (int i = 0; i < ipc.numVertices(); ++i)
                    for (int c = 0; c < ps; ++c)
                        worst = std::max(worst, std::fabs(estimate_component(static_cast<VERTEX*>(L.problem.vertex(i))->estimate(), c)
                                                          - poses[(size_t)ps * i + c]));
                std::printf("written_back %.3e\n", worst);
            }
            std::vector<EDGE*> odom_edges;
            getProblemOdom<EDGE>(L.problem, odom_edges);                                         // :50-52
            std::sort(odom_edges.begin(), odom_edges.end(), cmpEdgesID);                         // (the mock's propagateGuess indexes by vertex)
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
    if (!in_graph || std::getenv("IPC_ADAPTER_LOOP_ONLY")) return 0;
    {   // ---- the same loop with the candidate list announced first, then the rest of the class surface
        Loaded<EDGE, VERTEX, MS, D> L(path, true);
        const std::vector<EDGE*> order = L.call_order(ref_order, cfg.canonic_inliers);
        {
            IPC<EDGE, VERTEX> ipc(L.problem, cfg);
            ipc.setCandidates(L.loops_in_file);
            std::printf("announced_order");                        // (this graph's edge set has its own address order)
            for (EDGE* e : order) std::printf(" %d", L.file_index.at(e));
            std::printf("\nannounced");
            for (EDGE* e : order) std::printf(" %d", ipc.agreementCheck(e) ? 1 : 0);
            std::printf("\n");
            if (!ipc.getMaxConsensusSet().empty()) {
                EDGE* first = ipc.getMaxConsensusSet().front();
                EDGE twin = *first;                                // another edge object joining the same vertices, the other way round
                std::swap(twin._v[0], twin._v[1]);
                const bool removed = ipc.removeEdgeFromCnS(&twin); // the reference matches by (min id, max id) (src/consensus.cpp:81-90)
                std::printf("removed %d -> %zu\n", removed ? 1 : 0, ipc.getMaxConsensusSet().size());
                ipc.addEdgeToCnS(first);
                std::printf("added -> %zu\n", ipc.getMaxConsensusSet().size());
                ipc.addEdgeToCnS(&twin);                           // a member joins that pair already: nothing happens (:103-112)
                std::printf("added_twin -> %zu\n", ipc.getMaxConsensusSet().size());
            }
            const std::vector<char> all = ipc.agreementCheckAll(L.loops_in_file);
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
    setenv("GPU_MAX_HW_QUEUES", "24", 0);      // (the host program's line: INTEGRATION.md section 2; before the first HIP call)
    if (argc < 8) { std::fprintf(stderr, "usage: adapter_main dim file s fth fit sth sit [ref|stable] [graph|foreign] [canonic_inliers]\n"); return 2; }
    Config cfg;
    cfg.s_factor = std::atof(argv[3]);
    cfg.fast_reject_th = std::atof(argv[4]); cfg.fast_reject_iter_base = std::atoi(argv[5]);
    cfg.slow_reject_th = std::atof(argv[6]); cfg.slow_reject_iter_base = std::atoi(argv[7]);
    const bool ref_order = argc < 9 || !std::strcmp(argv[8], "ref");
    const bool in_graph = argc < 10 || !std::strcmp(argv[9], "graph");
    cfg.canonic_inliers = argc < 11 ? 0 : std::atoi(argv[10]);
    try {
        return std::atoi(argv[1]) == 2 ? run<g2o::EdgeSE2, g2o::VertexSE2, 3, 3>(argv[2], cfg, ref_order, in_graph)
                                       : run<g2o::EdgeSE3, g2o::VertexSE3, 7, 6>(argv[2], cfg, ref_order, in_graph);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
