// scene_check.cpp -- drives the C ABI through the C++ host mirror (include/nenbody_scene.hpp) and dumps the state,
// so that tests/test_gpu_cpp_host.py can compare it with the oracle.  usage: scene_check N K OUT.bin [fast]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nenbody_scene.hpp"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const uint32_t n = (uint32_t)std::atoi(argv[1]);
    const uint32_t k = (uint32_t)std::atoi(argv[2]);
    const bool fast = argc > 4 && std::strcmp(argv[4], "fast") == 0;
    try {
        nenbody::Scene scene(n, nenbody::default_params(fast ? NB_MODE_FAST : NB_MODE_STRICT), 1234);
        // first k-1 steps device-resident, the last one through step() (refreshes the host mirrors)
        if (k > 1) scene.step_n(k - 1);
        scene.step();
        // and the drop-in form once more on a copy, to check it agrees with Scene::step
        std::vector<nenbody::Vec3> p = scene.positions, v = scene.velocities, op(n), ov(n);
        std::vector<nenbody::Mat4> inst(n);
        nenbody::update_instance_nbody(inst, p, op, v, ov);
        scene.step();
        if (std::memcmp(p.data(), scene.positions.data(), n * sizeof(nenbody::Vec3)) != 0 ||
            std::memcmp(v.data(), scene.velocities.data(), n * sizeof(nenbody::Vec3)) != 0) {
            std::fprintf(stderr, "update_instance_nbody and Scene::step disagree\n");
            return 3;
        }
        FILE *f = std::fopen(argv[3], "wb");
        if (!f) return 4;
        std::fwrite(scene.positions.data(), sizeof(nenbody::Vec3), n, f);
        std::fwrite(scene.velocities.data(), sizeof(nenbody::Vec3), n, f);
        std::fwrite(scene.instances.data(), sizeof(nenbody::Mat4), n, f);
        std::fclose(f);
        std::printf("ok steps=%llu\n", (unsigned long long)scene.steps_done());
    } catch (const nenbody::Error &e) {
        std::fprintf(stderr, "nenbody error %d: %s\n", e.status, e.what());
        return 10;
    }
    return 0;
}
