// cpprob::StateType / cpprob::State -- reference include/cpprob/state.hpp:28-54, src/cpprob/state.cpp:20-71.
// `smc` is new (the reference has {compile, csis, sis, dryrun} only).  The reference keeps the mode
// in a process-global static; here it is per thread, so independent inferences may run concurrently.
#ifndef CPPROB_COMPAT_STATE_HPP
#define CPPROB_COMPAT_STATE_HPP

namespace cpprob {

enum class StateType {
    compile,
    csis,
    sis,
    dryrun,
    smc       // sequential Monte Carlo: resample between observes (thesis Alg. 1 p.36)
};

class State {
public:
    static void set(StateType s) { state() = s; }
    static bool compile() { return state() == StateType::compile; }
    static bool csis() { return state() == StateType::csis; }
    static bool sis() { return state() == StateType::sis; }
    static bool smc() { return state() == StateType::smc; }
    static bool dryrun() { return state() == StateType::dryrun; }
    static bool rejection_sampling() { return false; }
    static void start_rejection_sampling() {}
    static void finish_rejection_sampling() {}
private:
    static StateType& state() { static thread_local StateType s = StateType::dryrun; return s; }
};

}  // namespace cpprob
#endif
