// Host driver in plain C++14 with the command-line surface of the reference's src/main.cpp for the
// sis path (flags :145-170, flow :69-107), plus --smc:
//     cpprob_main --model hmm16 --smc --n_samples 100000 --observes "[0.3 -1 ...]" --estimate
// Flags: --model {gaussian_unknown_mean, gaussian_readme, linear_gaussian_1d25, linear_gaussian_1d100, hmm16, hmm128, poisson_rate, gaussian_2d_unk_mean, ...}
//        --sis | --smc        --n_samples N (default 10000)    --observes "…" | --observes_file F
//        --generated_file NAME (default "post")   --model_folder DIR (default ".")   --estimate
//        additions: --seed S  --resampler {systematic,stratified,multinomial}  --ess_threshold X
//                   --generic (run the unchanged model body on the GPU instead of the fused kernels)
//                   --gpus N | --devices a,b,...  (one joint population sharded over several GPUs: exact global resampling, RCCL over xGMI;
//                                                  equal entries, e.g. 0,0,0: every rank on that GPU)
//                   --filtering_only (smc, built-in models: O(N) particle store, filtering statistics, no posterior files)
//                   --no_dump  --json (print the in-memory result as one JSON line)
// This file never touches HIP: it calls cpprob::inference exactly as the reference's main does.
#include <array>
#include <cstdlib>
#include <iostream>
#include <string>
#include <tuple>

#include "cpprob/cpprob.hpp"
#include "cpprob/postprocess/stats_printer.hpp"
#include "cpprob/serialization.hpp"
#include "registered_models.hpp"

namespace {

struct Args {
    std::string model, model_folder = ".", observes, observes_file, generated_file = "post";
    bool sis = false, smc = false, estimate = false, json = false;
    int repeat = 1;
    std::size_t n_samples = 10000;            // src/main.cpp:166
};

void print_json(const cpprob::gpu::Result& r)
{
    std::cout.precision(17);
    std::cout << "{\"n\": " << r.n_particles << ", \"log_evidence\": " << r.log_evidence << ", \"ess\": " << r.ess
              << ", \"n_resampled\": " << r.n_resampled << ", \"run_seconds\": " << r.run_seconds << ", \"builtin\": " << (r.used_builtin ? "true" : "false")
              << ", \"n_gpus\": " << r.n_gpus << ", \"exchange_reruns\": " << r.exchange_reruns << ", \"replay_window\": " << r.replay_window << ", \"markov_crosscheck\": " << r.markov_crosscheck
              << ", \"joint\": " << (r.joint ? "true" : "false") << ", \"joint_note\": \"" << r.joint_note << "\"" << ", \"step_form\": " << r.step_form << ", \"launches_per_step\": " << r.launches_per_step << ", \"step_builds_used\": " << r.step_builds_used << ", \"setup_seconds\": " << r.setup_seconds
              << ", \"workspace_grown\": " << (r.workspace_grown ? "true" : "false") << ", \"predicts\": [";
    for (std::size_t i = 0; i < r.predicts.size(); ++i) {
        const auto& p = r.predicts[i];
        if (i) std::cout << ", ";
        std::cout << "{\"address\": \"" << p.address << "\", ";
        if (p.is_int) {
            std::cout << "\"p\": [";
            for (std::size_t s = 0; s < p.probabilities.size(); ++s) std::cout << (s ? ", " : "") << p.probabilities[s];
            std::cout << "]}";
        } else {
            std::cout << "\"mean\": " << p.mean << ", \"variance\": " << p.variance;
            if (p.mean_nd.size() > 1) {                      // vector-valued predict: one entry per component
                std::cout << ", \"mean_nd\": [";
                for (std::size_t d = 0; d < p.mean_nd.size(); ++d) std::cout << (d ? ", " : "") << p.mean_nd[d];
                std::cout << "], \"variance_nd\": [";
                for (std::size_t d = 0; d < p.variance_nd.size(); ++d) std::cout << (d ? ", " : "") << p.variance_nd[d];
                std::cout << "]";
            }
            std::cout << "}";
        }
    }
    std::cout << "]";
    if (r.n_replicates > 1) {
        std::cout << ", \"replicates\": " << r.n_replicates << ", \"log_evidence_mean\": " << r.log_evidence_mean << ", \"log_evidence_sd\": " << r.log_evidence_sd
                  << ", \"replicates_seconds\": " << r.replicates_seconds << ", \"predict_mean\": [";
        for (std::size_t h = 0; h < r.predict_mean.size(); ++h) std::cout << (h ? ", " : "") << r.predict_mean[h];
        std::cout << "], \"predict_sd\": [";
        for (std::size_t h = 0; h < r.predict_sd.size(); ++h) std::cout << (h ? ", " : "") << r.predict_sd[h];
        std::cout << "]";
    }
    std::cout << "}" << std::endl;
}

template <class F>
int execute(const F& model, const Args& a)
{
    if (a.observes_file.empty() == a.observes.empty()) {
        std::cerr << R"(In SIS or SMC mode exactly one of the options "--observes" or "--observes_file" has to be set)" << std::endl;   // main.cpp:72-75
        return EXIT_FAILURE;
    }
    cpprob::tuple_observes_t<F> observes;
    const bool ok = a.observes_file.empty() ? cpprob::parse_string(a.observes, observes)
                                            : cpprob::parse_file(a.model_folder + "/" + a.observes_file, observes);
    if (!ok) {
        std::cerr << "Could not parse the observations.\n"
                  << "Please use spaces to separate the observations and elements of an aggregate type instead of commas.\n";   // main.cpp:87-92
        return EXIT_FAILURE;
    }
    const std::string post = a.model_folder + "/" + a.generated_file + (a.smc ? "_smc" : "_sis");                            // main.cpp:49-53
    if (a.smc) std::cout << "Sequential Monte Carlo (SMC)" << std::endl;
    else std::cout << "Sequential Importance Sampling (SIS)" << std::endl;                                                   // main.cpp:99
    for (int rep = 0; rep < a.repeat; ++rep) {                       // --repeat: the first call pays code loading; later ones show the warm rate
        if (rep) cpprob::gpu::options().seed += 1;
        cpprob::inference(a.smc ? cpprob::StateType::smc : cpprob::StateType::sis, model, observes, a.n_samples, post);
        if (a.repeat > 1) std::cout << "run " << rep << ": " << cpprob::gpu::last_result().run_seconds * 1e3 << " ms (set-up " << cpprob::gpu::last_result().setup_seconds * 1e3 << " ms)" << std::endl;
    }
    if (a.json) print_json(cpprob::gpu::last_result());
    if (a.estimate) {
        std::cout << "Posterior Distribution Estimators" << std::endl;                                                      // main.cpp:104
        std::cout << cpprob::StatsPrinter{post};
    }
    cpprob::gpu::release_device_resources();                         // (contexts and workspaces kept between calls: freed while the runtime is alive)
    return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char** argv)
{
    Args a;
    auto& opt = cpprob::gpu::options();
    for (int i = 1; i < argc; ++i) {
        const std::string f = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << f << std::endl; std::exit(EXIT_FAILURE); } return argv[++i]; };
        if (f == "--model" || f == "-m") a.model = next();
        else if (f == "--model_folder") a.model_folder = next();
        else if (f == "--sis") a.sis = true;
        else if (f == "--smc") a.smc = true;
        else if (f == "--estimate" || f == "-e") a.estimate = true;
        else if (f == "--n_samples" || f == "-n") a.n_samples = std::stoull(next());
        else if (f == "--observes" || f == "-o") a.observes = next();
        else if (f == "--observes_file") a.observes_file = next();
        else if (f == "--generated_file") a.generated_file = next();
        else if (f == "--seed") opt.seed = std::stoull(next());
        else if (f == "--ess_threshold") opt.ess_threshold = std::stod(next());
        else if (f == "--resampler") { const std::string r = next(); opt.resampler = r == "multinomial" ? 2 : (r == "stratified" ? 1 : 0); }
        else if (f == "--generic") opt.prefer_builtin = false;
        else if (f == "--filtering_only") { opt.keep_history = false; opt.dump = false; }   // smc, built-in models: O(N) particle store, filtering statistics
        else if (f == "--islands") opt.islands = true;                          // unchanged-model smc over several ranks: independent runs combined by evidence
        else if (f == "--joint_across_devices") opt.joint_across_devices = true; // ... or the joint population even across physical GPUs (unvalidated on real links)
        else if (f == "--step_form") opt.step_form_override = std::stoi(next());   // unchanged-model smc: 0 separate bookkeeping launches, 1 fused (bounds), 2 fused (exact maxima)
        else if (f == "--no_markov_probe") opt.markov_probe = false;          // unchanged-model smc: replay the whole trace every step
        else if (f == "--no_markov_crosscheck") opt.markov_crosscheck = false; // ... trust the host probe's window without the device pilot
        else if (f == "--gpus") { const int k = std::stoi(next()); opt.devices.clear(); for (int d = 0; d < k; ++d) opt.devices.push_back(d); }
        else if (f == "--devices") {                        // e.g. 0,1,2,3 -- or 0,0 for two ranks on one GPU (loopback transport)
            const std::string v = next(); opt.devices.clear();
            std::size_t p0 = 0;
            while (p0 <= v.size()) { const std::size_t q = v.find(',', p0); opt.devices.push_back(std::stoi(v.substr(p0, q == std::string::npos ? q : q - p0))); if (q == std::string::npos) break; p0 = q + 1; }
        }
        else if (f == "--repeat") a.repeat = std::stoi(next());
        else if (f == "--no_step_builds") opt.step_builds = false;          // unchanged-model smc: the run-time step kernel at every step (A/B against the per-step builds)
        else if (f == "--replicates") opt.replicates = std::stoi(next());
        else if (f == "--no_dump") opt.dump = false;
        else if (f == "--json") a.json = true;
        else { std::cerr << "unknown option " << f << std::endl; return EXIT_FAILURE; }
    }
    if (a.sis == a.smc) { std::cerr << "exactly one of --sis / --smc has to be set" << std::endl; return EXIT_FAILURE; }
    try {
        if (a.model == "gaussian_unknown_mean") return execute(models::gaussian_unknown_mean<double>, a);     // main.cpp:123-130
        if (a.model == "gaussian_readme") return execute(models::gaussian_readme<double>, a);
        if (a.model == "linear_gaussian_1d25") return execute(models::linear_gaussian_1d<25>, a);
        if (a.model == "linear_gaussian_1d100") return execute(models::linear_gaussian_1d<100>, a);
        if (a.model == "hmm16") return execute(models::hmm<16>, a);
        if (a.model == "hmm128") return execute(models::hmm<128>, a);
        if (a.model == "poisson_rate") return execute(models::poisson_rate<double>, a);
        if (a.model == "gaussian_2d_unk_mean") return execute(models::gaussian_2d_unk_mean<double>, a);       // observes: "[y0 y1]"
        if (a.model == "gauss_functor") return execute(models::GaussFunctor<double>{}, a);
        if (a.model == "gaussian_by_rejection") return execute(models::gaussian_by_rejection<double>, a);
        if (a.model == "second_order12") return execute(models::second_order<12>, a);
        if (a.model == "running_mean12") return execute(models::running_mean<12>, a);
        if (a.model == "rare_memory12") return execute(models::rare_memory<12>, a);
        if (a.model == "random_scale12") return execute(models::random_scale<12>, a);
        if (a.model == "all_distr") return execute(models::all_distr<int>, a);                                   // src/models/models.cpp:13-47; observes: any two ints
        std::cerr << "unknown model " << a.model << std::endl;
        return EXIT_FAILURE;
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 2;
    }
}
