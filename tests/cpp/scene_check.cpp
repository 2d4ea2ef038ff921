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
        const nb_params prm = nenbody::default_params(fast ? NB_MODE_FAST : NB_MODE_STRICT);
        nenbody::Scene scene(n, prm, 1234);
        // first k-1 steps device-resident, the last one through step() (refreshes the host mirrors)
        if (k > 1) scene.step_n(k - 1);
        scene.step();
        // and the drop-in form once more on a copy, to check it agrees with Scene::step
        std::vector<nenbody::Vec3> p = scene.positions, v = scene.velocities, op(n), ov(n);
        std::vector<nenbody::Mat4> inst(n);
        nenbody::update_instance_nbody(inst, p, op, v, ov, &prm);
        scene.step();
        if (std::memcmp(p.data(), scene.positions.data(), n * sizeof(nenbody::Vec3)) != 0 ||
            std::memcmp(v.data(), scene.velocities.data(), n * sizeof(nenbody::Vec3)) != 0) {
            std::fprintf(stderr, "update_instance_nbody and Scene::step disagree\n");
            return 3;
        }
        // the same for the boids controller (src/main.rs:443-449), on copies: the dumped state stays the n-body one
        {
            std::vector<nenbody::Vec3> bp = scene.positions, bv = scene.velocities;
            nenbody::Scene twin(bp, bv, prm);
            nenbody::update_instance_boids(inst, bp, op, bv, ov);
            twin.step_boids();
            if (std::memcmp(bp.data(), twin.positions.data(), n * sizeof(nenbody::Vec3)) != 0 ||
                std::memcmp(bv.data(), twin.velocities.data(), n * sizeof(nenbody::Vec3)) != 0 ||
                std::memcmp(inst.data(), twin.instances.data(), n * sizeof(nenbody::Mat4)) != 0) {
                std::fprintf(stderr, "update_instance_boids and Scene::step_boids disagree\n");
                return 5;
            }
            try {  // copy_from_slice panics on unequal lengths (src/main.rs:459)
                std::vector<nenbody::Vec3> too_short(n > 1 ? n - 1 : 2);
                nenbody::update_instance_boids(inst, bp, too_short, bv, ov);
                std::fprintf(stderr, "a length mismatch was accepted\n");
                return 6;
            } catch (const std::invalid_argument &) {
            }
        }
        // the third controller (src/main.rs:381-385) as one call: the same stream twice gives the same bits, the zip bounds it
        {
            std::vector<nenbody::Vec3> rp = scene.positions, rv = scene.velocities, rp2 = rp, rv2 = rv;
            std::vector<nenbody::Mat4> ri(n), ri2(n > 1 ? n / 2 : 1);
            nenbody::update_instance_random(ri, rp, rv, 42, 7);
            nenbody::update_instance_random(ri2, rp2, rv2, 42, 7);
            const size_t m = ri2.size();
            if (std::memcmp(rp.data(), rp2.data(), m * sizeof(nenbody::Vec3)) != 0 ||
                std::memcmp(rv.data(), rv2.data(), m * sizeof(nenbody::Vec3)) != 0 ||
                (m < n && std::memcmp(rp2.data() + m, scene.positions.data() + m, (n - m) * sizeof(nenbody::Vec3)) != 0) ||
                std::memcmp(rp.data(), scene.positions.data(), n * sizeof(nenbody::Vec3)) == 0) {
                std::fprintf(stderr, "update_instance_random: stream or zip wrong\n");
                return 8;
            }
        }
        // the sharded host with a world of one (no exchange needed) must reproduce the same step
        {
            nenbody::Scene twin(n, prm, 1234);
            nenbody::Shard shard(twin.positions, twin.velocities, 0, 1, prm);
            shard.step(k + 1);
            std::vector<nenbody::Vec3> sp, sv;
            std::vector<nenbody::Mat4> si;
            shard.download(sp, sv, si);
            if (shard.first() != 0 || shard.count() != n ||
                std::memcmp(sp.data(), scene.positions.data(), n * sizeof(nenbody::Vec3)) != 0 ||
                std::memcmp(sv.data(), scene.velocities.data(), n * sizeof(nenbody::Vec3)) != 0) {
                std::fprintf(stderr, "Shard (world of one) and Scene disagree\n");
                return 7;
            }
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
