// Generates tests/golden/philox_rocrand.json from rocRAND's OWN host-callable
// Philox4x32-10 engine and Box-Muller transform (/opt/rocm/include/rocrand).
// Independent pin for the oracle's and the HIP library's generator.
// Build+run (no GPU needed): see make_golden.py.
#include <cstdio>
#include <cstdint>
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>

int main()
{
    const unsigned long long seeds[] = {0ull, 12345ull, 0xdeadbeefdeadbeefull};
    // "pid" is the rocRAND subsequence = the engine's block group id
    const unsigned long long pids[] = {0ull, 1ull, 9999999ull, (1ull << 33) + 7ull};
    const unsigned long long draws[] = {0ull, 1ull, 15ull, (1ull << 40) + 3ull};
    std::printf("{\"cases\": [\n");
    bool first = true;
    for (auto seed : seeds) for (auto pid : pids) for (auto draw : draws) {
        rocrand_state_philox4x32_10 st;
        rocrand_init(seed, pid, 4ull * draw, &st);
        uint4 r = rocrand4(&st);
        double2 n = rocrand_device::detail::box_muller_double(r);
        double u = rocrand_device::detail::uniform_distribution_double(r.x, r.y);
        std::printf("%s{\"seed\": %llu, \"pid\": %llu, \"draw\": %llu, \"words\": [%u, %u, %u, %u], "
                    "\"normal_x\": \"%a\", \"normal_y\": \"%a\", \"u01\": \"%a\"}",
                    first ? "" : ",\n", seed, pid, draw, r.x, r.y, r.z, r.w, n.x, n.y, u);
        first = false;
    }
    std::printf("\n]}\n");
    return 0;
}
