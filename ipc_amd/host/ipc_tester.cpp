// ipc_tester_2D / ipc_tester_3D -- same CLI as the reference's examples/ipc_tester_{2D,3D}.cpp
// ("-c <cfg.yaml>", examples/ipc_tester_2D.cpp:13-17), same config keys, same output files.
// Built twice from this source with -DIPC_TESTER_DIM=2 / 3.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>

#include "ipc_host.hpp"

#ifndef IPC_TESTER_DIM
#define IPC_TESTER_DIM 2
#endif

int main(int argc, char** argv)
{
    // the faithful mode keeps 16 solves in flight, one per HIP stream: the runtime needs as many hardware queues, and reads
    // this once, at its first call (include/ipc_amd.h "environment"; the host program's line, not the library's)
    setenv("GPU_MAX_HW_QUEUES", "24", 0);
    std::string cfgFilename;
    for (int i = 1; i < argc; ++i)
        if (!std::strcmp(argv[i], "-c") && i + 1 < argc) cfgFilename = argv[++i];
    if (cfgFilename.empty()) {
        std::cerr << "usage: " << argv[0] << " -c <cfg.yaml>   (path to cfg file)" << std::endl;
        return 2;
    }
    try {
        ipc_host::Config cfg;
        ipc_host::readConfig(cfgFilename, cfg);
        ipc_host::PoseGraph g;
        ipc_host::loadG2O(cfg.dataset, g);
        if (g.dim != IPC_TESTER_DIM) throw std::runtime_error("dataset is not a " + std::to_string(IPC_TESTER_DIM) + "D graph");
        std::vector<ipc_host::Edge> odom, loops;
        ipc_host::splitProblemConstraints(g, odom, loops);
        ipc_host::simulating_incremental_data(cfg, g, odom, loops);
    } catch (const std::exception& e) {
        std::cerr << "ipc_tester: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
