// TEST MOCK of the reference's include/ipc/consensus_utils.hpp: the two functions the adapter's constructor and the
// harness call (propagateGuess, robustifyVoters), over the g2o mock of ipc/utils.hpp.  Same signatures and effects as
// reference src/consensus_utils.cpp:99-116,124-130.
#pragma once
#include "ipc/utils.hpp"

template <class EDGE, class VERTEX>
void propagateGuess(g2o::SparseOptimizer& voting, int id1, int id2, const std::vector<EDGE*>& odom)
{
    auto gauge = static_cast<VERTEX*>(voting.vertex(id1));
    gauge->setFixed(true);
    gauge->setToOrigin();
    for (int i = id1 + 1; i <= id2; ++i) {
        auto v1 = static_cast<VERTEX*>(voting.vertex(i - 1));
        auto v2 = static_cast<VERTEX*>(voting.vertex(i));
        v2->setFixed(false);
        v2->setEstimate(v1->estimate() * odom[i - 1]->measurement());
    }
}

template <class EDGE>
void robustifyVoters(int id1, int id2, double s_factor, std::vector<EDGE*>& voters)
{
    for (int j = id1; j < id2; ++j) voters[j]->setInformation(voters[j]->information() * s_factor);
}
