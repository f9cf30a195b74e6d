// Model registry: how cpprob::inference(algorithm, f, observes, n) -- called with the model as a run
// time function reference, exactly as in the reference (src/main.cpp:100) -- finds the device code of
// `f`.  A model translation unit compiled by hipcc registers either
//   CPPROB_REGISTER_MODEL(models::hmm<16>);                   generic: the unchanged model source runs on device
//   CPPROB_REGISTER_BUILTIN(models::hmm<16>, CPPROB_HIP_MODEL_HMM3);   the hand-fused kernels of libcpprob_hip
// keyed by the function's address (functors: by type).  Host code compiled as plain C++14 only needs
// this header.
#ifndef CPPROB_COMPAT_DETAIL_REGISTRY_HPP
#define CPPROB_COMPAT_DETAIL_REGISTRY_HPP
#include <cstddef>
#include <map>
#include <mutex>
#include <string>
#include <typeinfo>
#include <vector>

#include "cpprob/detail/host_trace.hpp"
#include "cpprob/gpu_result.hpp"
#include "cpprob/state.hpp"

namespace cpprob {
namespace gpu {

struct HostStore {                      // host copy of the particle store, for the file dump
    std::vector<double> real;           // [n_real_hits][n]
    std::vector<std::int32_t> ints;     // [n_int_hits][n]
    std::vector<double> logw;           // [n]
    std::size_t n = 0;
};

// type-erased launcher: observes points to a tuple_observes_t<Model>
using GenericLauncher = void (*)(StateType algorithm, const void* observes, std::size_t n, const detail::TraceStructure& st,
                                 const Options& opt, Result& res, HostStore* store);

// ... one joint population over options().devices (cpprob/gpu.hpp: generic_joint_launcher); false: no joint form for this model / these options
using GenericJointLauncher = bool (*)(StateType algorithm, const void* observes, std::size_t n, const detail::TraceStructure& st,
                                      const Options& opt, Result& res, HostStore* store);

struct Entry {
    std::string name;
    int builtin_model = -1;             // >= 0: CPPROB_HIP_MODEL_* id
    GenericLauncher generic = nullptr;
    GenericJointLauncher generic_joint = nullptr;
    bool generic_vectors = false;       // the generic launcher runs the model's device view: vector-valued statements included
};

struct Key {
    const void* fn; std::size_t type_hash;
    bool operator<(const Key& o) const { return fn != o.fn ? fn < o.fn : type_hash < o.type_hash; }
};

#if defined(__GNUC__)
#define CPPROB_REGISTRY_VISIBLE __attribute__((visibility("default")))
#else
#define CPPROB_REGISTRY_VISIBLE
#endif

// one table per process (inline function with a local static: merged across translation units and
// shared objects)
CPPROB_REGISTRY_VISIBLE inline std::map<Key, Entry>& table() { static std::map<Key, Entry> t; return t; }
CPPROB_REGISTRY_VISIBLE inline std::mutex& table_mutex() { static std::mutex m; return m; }

inline bool add_entry(const Key& k, const Entry& e)
{
    std::lock_guard<std::mutex> lock(table_mutex());
    Entry& dst = table()[k];
    if (dst.name.empty()) dst.name = e.name;
    if (e.builtin_model >= 0) dst.builtin_model = e.builtin_model;
    if (e.generic && (e.generic_vectors || !dst.generic_vectors)) { dst.generic = e.generic; dst.generic_vectors = e.generic_vectors; dst.generic_joint = e.generic_joint; }
    return true;
}

// What the device paths keep between two cpprob::inference calls (contexts, streams, workspaces) is released by hooks the model
// translation units leave here.  A program may call release_device_resources() before it returns from main: otherwise those objects
// are simply never torn down (static destructors run when the HIP runtime may already be gone) -- which is harmless but leaves the
// runtime to end a process with live streams.
using ReleaseHook = void (*)();
CPPROB_REGISTRY_VISIBLE inline std::vector<ReleaseHook>& release_hooks() { static std::vector<ReleaseHook> h; return h; }
inline bool add_release_hook(ReleaseHook f)
{
    std::lock_guard<std::mutex> lock(table_mutex());
    for (ReleaseHook g : release_hooks()) if (g == f) return true;
    release_hooks().push_back(f);
    return true;
}
inline void release_device_resources()
{
    std::vector<ReleaseHook> hooks;
    { std::lock_guard<std::mutex> lock(table_mutex()); hooks = release_hooks(); }
    for (ReleaseHook f : hooks) f();
}

inline const Entry* find_entry(const Key& k)
{
    std::lock_guard<std::mutex> lock(table_mutex());
    auto it = table().find(k);
    return it == table().end() ? nullptr : &it->second;
}

template <class F>
Key key_of(const F& f, std::true_type /*function*/) { return Key{reinterpret_cast<const void*>(&f), 0}; }
template <class F>
Key key_of(const F&, std::false_type /*functor*/) { return Key{nullptr, typeid(F).hash_code()}; }

template <class FP>
bool register_builtin(FP fn, int model_id, const char* name)
{
    Entry e; e.name = name; e.builtin_model = model_id;
    return add_entry(Key{reinterpret_cast<const void*>(fn), 0}, e);
}

}  // namespace gpu
}  // namespace cpprob

#define CPPROB_PP_CAT2(a, b) a##b
#define CPPROB_PP_CAT(a, b) CPPROB_PP_CAT2(a, b)
#define CPPROB_REGISTER_BUILTIN(fn, model_id) \
    static const bool CPPROB_PP_CAT(cpprob_reg_builtin_, __LINE__) = ::cpprob::gpu::register_builtin(&fn, model_id, #fn)

#endif
