// consensus_amd.hpp -- drop-in for the reference's include/ipc/consensus.hpp (class surface of
// reference include/ipc/consensus.hpp:5-33, behaviour of src/consensus.cpp) on top of libipc_amd.so.
//
// Put this repo's include/ directory in FRONT of the reference's on the include path (include/ipc/consensus.hpp here
// is a one-line shim onto this header, so the testers' and src/simulation.cpp's literal `#include "ipc/consensus.hpp"`
// resolves to it), drop src/consensus.cpp from the build, keep src/consensus_utils.cpp and src/utils.cpp, link
// -lipc_amd: ipc_tester_2D/3D and src/simulation.cpp compile unchanged -- `IPC<EDGE, VERTEX> ipc(problem, cfg);
// ipc.agreementCheck(e); ipc.getMaxConsensusSet()` keep their meaning.  The g2o optimizer still owns the graph; the
// engine keeps its own copy of the odometry chain, the candidate edges it has seen and the pose state on the GPU.
//
// Like the reference's header (include/ipc/consensus.hpp:3) this one pulls in "ipc/consensus_utils.hpp" -- unchanged
// reference code, which the harness itself needs (propagateGuess, src/simulation.cpp:52) -- and through it
// "ipc/utils.hpp" (Config, getProblemOdom, cmpEdgesID; reference include/ipc/utils.hpp:22-38,98-121).  The constructor
// has the side effects on the caller's graph the reference's has: the odometry information is multiplied by s_factor
// IN PLACE (robustifyVoters, src/consensus.cpp:21 -- src/simulation.cpp:56 divides it back before the final
// optimize(1000)) and the vertices are set to the open-loop guess (propagateGuess, :23), both by the reference's own
// functions.
//
// The harness never announces its candidates (src/simulation.cpp:34-47 hands them to agreementCheck one by one), but they
// are all in the graph already: src/utils.cpp:172-189 splits the edges into odometry and loops without removing
// anything.  The constructor therefore reads the graph's loop edges itself (getProblemLoops, src/utils.cpp:197-210: the
// same walk over problem.edges() as splitProblemConstraints), puts them in the order the harness will call them
// (src/simulation.cpp:24-26: the same std::sort with the same cmpTime over the same sequence, hence the same
// permutation also among equal keys) and uploads them once, so that the engine works ahead of the caller.  An edge that
// is not in the graph is appended when agreementCheck first sees it (one record written in place); a caller with
// another order gets the same decisions (the engine's look-ahead is exact), just less overlap.
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "ipc/consensus_utils.hpp"
#include "ipc_amd.h"

namespace ipc_amd_adapter {

// the numbers of the edge's line in the .g2o file: SE2 "x y theta", SE3 "x y z qx qy qz qw"
template <class EDGE>
inline void pack_measurement(const EDGE& e, std::vector<double>& out, std::integral_constant<int, 3>)
{
    const auto v = e.measurement().toVector();                     // g2o::SE2::toVector
    out.push_back(v[0]); out.push_back(v[1]); out.push_back(v[2]);
}
template <class EDGE>
inline void pack_measurement(const EDGE& e, std::vector<double>& out, std::integral_constant<int, 6>)
{
    const auto v = g2o::internal::toVectorQT(e.measurement());     // what EdgeSE3::write stores
    for (int k = 0; k < 7; ++k) out.push_back(v[k]);
}
// upper triangle of the information matrix in row order (the file layout, reference src/utils.cpp:114)
template <class EDGE>
inline void pack_information_upper(const EDGE& e, std::vector<double>& out)
{
    constexpr int D = EDGE::Dimension;
    const auto& I = e.information();
    for (int i = 0; i < D; ++i)
        for (int j = i; j < D; ++j) out.push_back(I(i, j));
}

// engine pose (SE2: x y theta; SE3: R row-major, t) -> the vertex's estimate type
inline g2o::SE2 make_estimate(const double* p, std::integral_constant<int, 3>) { return g2o::SE2(p[0], p[1], p[2]); }
inline g2o::Isometry3 make_estimate(const double* p, std::integral_constant<int, 6>)
{
    g2o::Isometry3 t = g2o::Isometry3::Identity();
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) t(r, c) = p[3 * r + c];
        t(r, 3) = p[9 + r];
    }
    return t;
}

}  // namespace ipc_amd_adapter

template <class EDGE, class VERTEX>
class IPC {
public:
    using Dim = std::integral_constant<int, EDGE::Dimension>;      // 3: SE2, 6: SE3

    // reference src/consensus.cpp:9-33
    IPC(g2o::SparseOptimizer& open_loop_problem, const Config& cfg, int device = 0) : _problem(&open_loop_problem)
    {
        std::vector<EDGE*> odom;
        getProblemOdom<EDGE>(open_loop_problem, odom);             // src/utils.cpp:218-231
        std::sort(odom.begin(), odom.end(), cmpEdgesID);           // src/consensus.cpp:15
        std::vector<double> meas, info;
        for (EDGE* e : odom) {
            ipc_amd_adapter::pack_measurement(*e, meas, Dim{});
            ipc_amd_adapter::pack_information_upper(*e, info);
        }
        const ipc_params_t p{cfg.fast_reject_th, cfg.fast_reject_iter_base, cfg.slow_reject_th,
                             cfg.slow_reject_iter_base, cfg.s_factor};
        // (the engine gets the information as the file has it and scales its own copy)
        chk(ipc_create(EDGE::Dimension == 3 ? 2 : 3, (int)odom.size() + 1, meas.data(), info.data(), &p, device, &_h));
        // the reference constructor's effects on the caller's graph, by the reference's own functions
        robustifyVoters<EDGE>(0, (int)odom.size(), cfg.s_factor, odom);                         // src/consensus.cpp:21
        propagateGuess<EDGE, VERTEX>(open_loop_problem, 0, (int)odom.size(), odom);            // src/consensus.cpp:23
        // the candidates the harness is going to hand over, in the order it is going to (see the head of this file)
        std::vector<EDGE*> loops;
        getProblemLoops<EDGE>(open_loop_problem, loops);           // == splitProblemConstraints' `loops`, src/utils.cpp:172-210
        std::vector<std::pair<bool, EDGE*>> gt_loops;              // src/simulation.cpp:24-26
        for (size_t idx = 0; idx < loops.size(); ++idx) gt_loops.push_back(std::make_pair((int)idx < cfg.canonic_inliers, loops[idx]));
        std::sort(gt_loops.begin(), gt_loops.end(), cmpTime);
        std::vector<EDGE*> expected;
        for (const auto& q : gt_loops) expected.push_back(q.second);
        if (!expected.empty()) setCandidates(expected);
        // (streams, workspaces: construction is not inside the harness's per-candidate timer, src/simulation.cpp:28,36-38)
        chk(ipc_incremental_prepare(_h));
    }
    // reference src/consensus.cpp:35-40 (clears the caller's graph, as the reference does)
    ~IPC()
    {
        ipc_destroy(_h);
        _problem->clear();
    }
    IPC(const IPC&) = delete;
    IPC& operator=(const IPC&) = delete;

    // Replaces the candidate list the constructor read from the graph (any order; the engine expects the calls in
    // cmpTime order, ties in the order of this list); the consensus set and the poses go back to the open-loop state.
    void setCandidates(const std::vector<EDGE*>& candidates)
    {
        _cands = candidates;
        _index_of.clear();
        for (size_t k = 0; k < _cands.size(); ++k) _index_of[_cands[k]] = (int)k;
        _max_consensus_set.clear();
        upload();
    }

    // reference src/consensus.cpp:43-75: cluster, thresholds, solve from the current estimates, keep / restore,
    // propagateCurrentGuess -- on the GPU, and on the edge OBJECT it is given (its own measurement and information,
    // :47-56): an object the engine has not seen is a new candidate, whatever vertices it joins.
    bool agreementCheck(EDGE* loop_candidate)
    {
        const int k = index_of(loop_candidate);
        int ok = 0;
        chk(ipc_agreement_check(_h, k, &ok, nullptr));
        if (ok) {
            _max_consensus_set.push_back(loop_candidate);
            if (_write_back) writeBackEstimates();
        }
        return ok != 0;
    }

    // Batched re-formulation (consistency matrix + set-max) of the harness
    // loop src/simulation.cpp:34-47 over `candidates`: per candidate, whether it is in the set.
    std::vector<char> agreementCheckAll(const std::vector<EDGE*>& candidates)
    {
        setCandidates(candidates);
        std::vector<uint8_t> acc(candidates.size());
        std::vector<int> order(candidates.size());
        if (!candidates.empty()) {
            chk(ipc_run(_h, nullptr, acc.data()));
            chk(ipc_candidate_order(_h, order.data()));
        }
        _max_consensus_set.clear();
        for (int k : order)
            if (acc[k]) _max_consensus_set.push_back(candidates[k]);
        return std::vector<char>(acc.begin(), acc.end());
    }

    // reference src/consensus.cpp:77-96: the FIRST member of the set that joins the same (min id, max id) goes
    bool removeEdgeFromCnS(EDGE* edge_to_remove)
    {
        const int k = same_pair(edge_to_remove);
        if (k < 0) return false;                                   // no candidate joins this pair, so no member does
        int removed = 0;
        chk(ipc_remove_from_consensus(_h, k, &removed));
        if (removed) refresh_set();
        return removed != 0;
    }
    // reference src/consensus.cpp:98-119: nothing happens if a member joins the same (min id, max id); else the edge
    // itself joins the set, which is then sorted by cmpEdgesTime
    void addEdgeToCnS(EDGE* edge_to_add)
    {
        chk(ipc_add_to_consensus(_h, index_of(edge_to_add)));
        refresh_set();
    }
    // reference include/ipc/consensus.hpp:16
    const std::vector<EDGE*>& getMaxConsensusSet() const { return _max_consensus_set; }

    // The g2o vertex estimates are the reference's state: an accept leaves the optimised window and the re-propagated
    // tail in the caller's graph (src/consensus.cpp:69-71).  The engine keeps that state on the GPU; the harness never
    // reads it (src/simulation.cpp:52 re-propagates every vertex before its final optimisation), so by default nothing
    // is copied back.  A caller that does read the estimates between checks turns this on: after every accept the
    // current poses are written into the graph's vertices (one device-to-host copy of V poses per accept).
    void setWriteBackEstimates(bool on) { _write_back = on; }
    void writeBackEstimates()
    {
        std::vector<double> poses;
        currentPoses(poses);
        const int ps = EDGE::Dimension == 3 ? 3 : 12, V = numVertices();
        for (int i = 0; i < V; ++i)
            static_cast<VERTEX*>(_problem->vertex(i))->setEstimate(ipc_amd_adapter::make_estimate(poses.data() + (size_t)ps * i, Dim{}));
    }
    // SE2 [V][3] (x y theta), SE3 [V][12] (R row-major, t)
    void currentPoses(std::vector<double>& out) const
    {
        out.resize((size_t)numVertices() * (EDGE::Dimension == 3 ? 3 : 12));
        chk(ipc_current_poses(_h, out.data()));
    }
    int numVertices() const { return (int)_problem->vertices().size(); }

private:
    static void chk(int rc)
    {
        if (rc) throw std::runtime_error(ipc_last_error());
    }
    void upload()
    {
        std::vector<int> ids;
        std::vector<double> meas, info;
        for (EDGE* e : _cands) {
            ids.push_back(e->vertices()[0]->id());
            ids.push_back(e->vertices()[1]->id());
            ipc_amd_adapter::pack_measurement(*e, meas, Dim{});
            ipc_amd_adapter::pack_information_upper(*e, info);
        }
        chk(ipc_set_candidates(_h, (int)_cands.size(), ids.data(), meas.data(), info.data()));
    }
    // the engine's index of this edge OBJECT; an object it has not seen is appended (nothing is replayed)
    int index_of(EDGE* e)
    {
        auto it = _index_of.find(e);
        if (it != _index_of.end()) return it->second;
        std::vector<double> meas, info;
        const int ids[2] = {e->vertices()[0]->id(), e->vertices()[1]->id()};
        ipc_amd_adapter::pack_measurement(*e, meas, Dim{});
        ipc_amd_adapter::pack_information_upper(*e, info);
        int k = -1;
        chk(ipc_append_candidate(_h, ids, meas.data(), info.data(), &k));
        _cands.push_back(e);
        _index_of[e] = k;
        return k;
    }
    // a known candidate that joins the same (min id, max id) as e, which is all removeEdgeFromCnS looks at
    // (src/consensus.cpp:81-90); -1: none
    int same_pair(EDGE* e) const
    {
        auto it = _index_of.find(e);
        if (it != _index_of.end()) return it->second;
        const int a = std::min(e->vertices()[0]->id(), e->vertices()[1]->id()), b = std::max(e->vertices()[0]->id(), e->vertices()[1]->id());
        for (size_t k = 0; k < _cands.size(); ++k) {
            const int c = _cands[k]->vertices()[0]->id(), d = _cands[k]->vertices()[1]->id();
            if (std::min(c, d) == a && std::max(c, d) == b) return (int)k;
        }
        return -1;
    }
    void refresh_set()
    {
        int n = 0;
        chk(ipc_consensus_size(_h, &n));
        std::vector<int> idx((size_t)std::max(n, 1));
        chk(ipc_consensus_set(_h, idx.data()));
        _max_consensus_set.clear();
        for (int q = 0; q < n; ++q) _max_consensus_set.push_back(_cands[(size_t)idx[q]]);
    }

    g2o::SparseOptimizer* _problem;
    ipc_engine_t* _h = nullptr;
    std::vector<EDGE*> _cands;
    std::unordered_map<EDGE*, int> _index_of;
    std::vector<EDGE*> _max_consensus_set;
    bool _write_back = false;
};
