// Host-side C++ mirror of the reference's consensus / harness interface over the C ABI
// (include/ipc_amd.h).  Same names, argument meaning and output files as the reference:
//   struct Config + readConfig            reference include/ipc/utils.hpp:22-38, src/utils.cpp:316-337
//   splitProblemConstraints               reference src/utils.cpp:172-189
//   class IPC                             reference include/ipc/consensus.hpp:5-33
//   simulating_incremental_data           reference src/simulation.cpp:9-108
// g2o / Eigen / yaml-cpp are not available in this image, so the graph container, the g2o text
// reader and the YAML-subset reader below are self-contained.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ipc_amd.h"

namespace ipc_host {

struct Config {                      // the 14 keys readConfig dereferences (src/utils.cpp:320-334)
    std::string name, dataset, ground_truth, output;
    bool visualize = false;
    int canonic_inliers = 0;
    double s_factor = 1.0;
    double fast_reject_th = 0, slow_reject_th = 0;
    int fast_reject_iter_base = 0, slow_reject_iter_base = 0;
    bool use_best_k_buddies = false;
    int k_buddies = 0;
    bool use_recovery = false;
};

// Throws std::runtime_error when the file is missing or any of the 14 keys is absent (the
// reference's yaml-cpp conversions throw in that case too).
void readConfig(const std::string& cfg_filepath, Config& out_cfg);

struct Edge {                        // one EDGE_SE2 / EDGE_SE3:QUAT line
    int from = 0, to = 0;
    std::vector<double> meas;        // 3 (x y theta) or 7 (x y z qx qy qz qw)
    std::vector<double> info;        // 6 or 21 upper-triangular values, file order
};

struct PoseGraph {
    int dim = 0;                     // 2 or 3
    std::vector<std::vector<double>> vertices;   // file estimates, indexed by id
    std::vector<Edge> edges;         // file order
};

// optimizer.load (src/utils.cpp:114) for the four tags the reference's datasets use.
// Throws on unreadable files, mixed dimensions or vertex ids that are not exactly 0..V-1.
void loadG2O(const std::string& path, PoseGraph& g);

// splitProblemConstraints (src/utils.cpp:172-189): |id1 - id0| > 1 => loop, else odometry; both
// keep file order.  Odometry must be exactly one edge i -> i+1 per consecutive pair (the
// contract IPC::IPC relies on, src/consensus.cpp:13-23); otherwise throws.
void splitProblemConstraints(const PoseGraph& g, std::vector<Edge>& odom, std::vector<Edge>& loops);

class IPC {
public:
    IPC(const PoseGraph& open_loop_problem, const std::vector<Edge>& odom_sorted, const Config& cfg, int device = 0);
    ~IPC();
    IPC(const IPC&) = delete;
    IPC& operator=(const IPC&) = delete;

    // Batched form of the per-candidate agreementCheck loop (src/simulation.cpp:34-47): returns,
    // per candidate (file order), whether it is in the consensus set.
    std::vector<uint8_t> agreementCheckAll(const std::vector<Edge>& candidates);
    // The reference's own per-candidate interface (faithful incremental mode):
    // setCandidates uploads the list once, then agreementCheck(k) is IPC::agreementCheck
    // (src/consensus.cpp:43-75) for candidate k (index into that list).
    void setCandidates(const std::vector<Edge>& candidates);
    bool agreementCheck(int k);
    bool removeEdgeFromCnS(int k);                      // src/consensus.cpp:77-96
    void addEdgeToCnS(int k);                           // src/consensus.cpp:98-119
    // Final map of the harness (src/simulation.cpp:50-65): optimize(iterations) over odometry with
    // its information back to (info*s)/s plus the accepted candidates; [V][3] or [V][12].
    std::vector<double> finalMap(const std::vector<uint8_t>& accepted, int iterations = 1000,
                                 double* chi2_out = nullptr);
    // candidate indices in acceptance order (reference getMaxConsensusSet, consensus.hpp:16)
    const std::vector<int>& getMaxConsensusSet() const { return _max_consensus_set; }
    // cmpTime processing order of the last candidate list
    const std::vector<int>& order() const { return _order; }
    std::vector<double> initialPoses() const;          // propagateGuess result, [V][3 or 12]
    int dim() const { return _dim; }
    int numVertices() const { return _V; }

private:
    void refreshConsensus();
    ipc_engine_t* _h = nullptr;
    std::vector<ipc_engine_t*> _replicas;              // IPC_AMD_DEVICES: engines on the other GPUs (matrix mode, row shards)
    int _dim = 0, _V = 0;
    std::vector<int> _max_consensus_set, _order;
};

struct SimulationResult {
    int tp = 0, fp = 0, tn = 0, fn = 0;
    float precision = 0, recall = 0;
    double total_time = 0, avg_time = 0;
    int consensus_size = 0;
    double final_chi2 = 0;           // chi2 of the final map (the reference never prints it)
};

// The harness: labels the first cfg.canonic_inliers loops as inliers (src/simulation.cpp:24-25),
// runs the consensus, prints the reference's console lines, writes cfg.output (trajectory after
// the final map optimisation) and "<output minus 3 chars>PR"
// (src/simulation.cpp:91-105).  Environment IPC_AMD_MODE selects the consensus formulation:
// "incremental" (default: the reference's per-candidate agreementCheck loop on the GPU, same accepted
// set as the reference's algorithm) or "matrix" (batched consistency matrix + set-max).
SimulationResult simulating_incremental_data(const Config& cfg, const PoseGraph& g, const std::vector<Edge>& odom,
                                             const std::vector<Edge>& loops, int device = 0);

}  // namespace ipc_host
