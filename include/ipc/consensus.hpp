// Shim: with this repo's include/ directory in front of the reference's on the include path, the literal
// `#include "ipc/consensus.hpp"` of the reference's src/simulation.cpp:1 and examples/ipc_tester_{2D,3D}.cpp resolves here
// and picks up the class IPC<EDGE, VERTEX> on libipc_amd.so instead of the one src/consensus.cpp implements.
#pragma once
#include "ipc/consensus_amd.hpp"
