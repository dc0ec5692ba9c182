// Own-written stand-in for the *declarations* of Juicer's src/WFSTNetwork.h that the exact-signature bridge of
// include/juicer_amd_decoder.hpp reads (WFSTTransition :41-52, the public getters :129-167), with a trivial
// in-memory implementation so that the bridge can be RUN in a test (the real class parses FSM files and needs
// the absent Torch3).  Test infrastructure only.
#ifndef WFST_NETWORK_INC
#define WFST_NETWORK_INC
#include <vector>
#include "Decoder.h"
namespace Juicer {
struct WFSTTransition { int id; int toState; real weight; int inLabel; int outLabel; void *hook; };
class WFSTNetwork {
public:
    // (mock-only constructor: arcs grouped by source state, as the text loader leaves them)
    WFSTNetwork(int nStates_, int initState_, const std::vector<int> &from, const std::vector<WFSTTransition> &trans,
                const std::vector<int> &finalStates_, const std::vector<real> &finalWeights_)
        : initState(initState_), nStates(nStates_), transitions(trans), first((size_t)nStates_ + 1, 0), finalInd((size_t)nStates_, -1),
          finalWeights(finalWeights_)
    {
        for (size_t a = 0; a < from.size(); ++a) ++first[(size_t)from[a] + 1];
        for (int s = 0; s < nStates; ++s) first[(size_t)s + 1] += first[(size_t)s];
        for (size_t f = 0; f < finalStates_.size(); ++f) finalInd[(size_t)finalStates_[f]] = (int)f;
    }
    int getInitState() { return initState; }
    int getNumTransitions() { return (int)transitions.size(); }
    int getNumStates() { return nStates; }
    int getNumTransitionsOfOneState(int state) { return first[(size_t)state + 1] - first[(size_t)state]; }
    bool isFinalState(int stateIndex) { return finalInd[(size_t)stateIndex] >= 0; }
    real getFinalStateWeight(int stateIndex) { return finalWeights[(size_t)finalInd[(size_t)stateIndex]]; }
    WFSTTransition *getOneTransition(int transIndex) { return &transitions[(size_t)transIndex]; }
    int getTransID(int stateIndex, int nth) { return first[(size_t)stateIndex] + nth; }
private:
    int initState, nStates;
    std::vector<WFSTTransition> transitions;
    std::vector<int> first, finalInd;
    std::vector<real> finalWeights;
};
}
#endif
