// nenbody_scene.hpp -- header-only C++ host mirror of the reference's update interface, over the C ABI (nenbody.h).
//
// The reference is compiled code (Rust) whose toolchain is absent from the build image, so besides the Rust shim
// (integration/rust/scene.rs, uncompiled) the same host side is given in C++: `nenbody::Scene` is the type the
// reference's empty src/scene.rs (src/scene.rs:1) was meant to hold; `nenbody::update_instance_nbody` keeps the
// five-argument signature of src/main.rs:404-410.  Plumbing only: all arithmetic runs in libnenbody_hip.so.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "nenbody.h"

namespace nenbody {

using Vec3 = std::array<float, 3>;                 // memory of cgmath::Point3<f32> / Vector3<f32>
using Mat4 = std::array<std::array<float, 4>, 4>;  // memory of [[f32; 4]; 4], column-major

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};

inline void check(int rc, const nb_ctx *ctx)
{
    if (rc != NB_OK) throw Error(rc, nb_last_error(ctx));
}

// The library that got loaded must speak the ABI this header was written against: a symbol may keep its name and change its
// arguments between versions (nb_update_instance_random: eight in ABI 1, six since ABI 2), and the linker cannot tell.  Checked
// once per process, by every entry point of this header, before anything else crosses the boundary.
inline void check_abi()
{
    static const int got = nb_abi_version();
    if (got != NB_ABI_VERSION)
        throw Error(NB_ERR_UNSUPPORTED, "libnenbody_hip.so speaks ABI " + std::to_string(got) + ", this host was built against ABI " +
                                            std::to_string(NB_ABI_VERSION));
}

inline nb_params default_params(uint32_t mode = NB_MODE_STRICT)
{
    check_abi();
    nb_params p;
    nb_default_params(&p);  // src/main.rs:411-413
    p.mode = mode;
    return p;
}

class Scene {
public:
    std::vector<Vec3> positions;   // src/main.rs:743
    std::vector<Vec3> velocities;  // src/main.rs:738
    std::vector<Mat4> instances;   // instance_data, uploaded at src/main.rs:932-936

    // the reference's initial distributions (src/main.rs:738-747), seeded
    Scene(uint32_t n, const nb_params &params, uint64_t seed) : positions(n), velocities(n), instances(n)
    {
        check(nb_init_state(seed, n, positions[0].data(), velocities[0].data()), nullptr);
        create(params);
    }
    Scene(std::vector<Vec3> pos, std::vector<Vec3> vel, const nb_params &params)
        : positions(std::move(pos)), velocities(std::move(vel)), instances(positions.size())
    {
        // copy_from_slice panics on unequal lengths (src/main.rs:415-416)
        if (positions.size() != velocities.size()) throw std::invalid_argument("positions and velocities differ in length");
        create(params);
    }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    ~Scene() { nb_destroy(ctx_); }

    // one update_instance_nbody (src/main.rs:404-441); host mirrors refreshed for the consumers at src/main.rs:932-945
    void step()
    {
        check(nb_step(ctx_, 1), ctx_);
        check(nb_download(ctx_, positions[0].data(), velocities[0].data(), instances[0][0].data()), ctx_);
    }
    // k steps, device-resident, no download
    void step_n(uint32_t k) { check(nb_step(ctx_, k), ctx_); }
    // one update_instance_boids (src/main.rs:443-526), reference constants unless given; host mirrors refreshed
    void step_boids(const nb_boids_params *params = nullptr)
    {
        check(nb_step_boids(ctx_, 1, params), ctx_);
        check(nb_download(ctx_, positions[0].data(), velocities[0].data(), instances[0][0].data()), ctx_);
    }
    void step_boids_n(uint32_t k, const nb_boids_params *params = nullptr) { check(nb_step_boids(ctx_, k, params), ctx_); }
    void sync() { check(nb_sync(ctx_), ctx_); }
    void refresh() { check(nb_download(ctx_, positions[0].data(), velocities[0].data(), instances[0][0].data()), ctx_); }
    uint64_t steps_done() const { return nb_steps_done(ctx_); }

private:
    void create(const nb_params &params)
    {
        check_abi();
        if (positions.empty()) throw std::invalid_argument("a Scene needs at least one body");
        check(nb_create((uint32_t)positions.size(), 1, &params, &ctx_), nullptr);
        int rc = nb_upload(ctx_, positions[0].data(), velocities[0].data());
        if (rc != NB_OK) {
            std::string msg = nb_last_error(ctx_);
            nb_destroy(ctx_);
            ctx_ = nullptr;
            throw Error(rc, msg);
        }
    }
    nb_ctx *ctx_ = nullptr;
};

// One rank's share of a scene on a multi-GPU host (one process or thread per GPU): nb_shard_* of nenbody.h.  Every rank
// passes the same full state; the rank keeps its index range and a replica of all positions, and each step ends with
// one exchange (RCCL when constructed with a comm id from Shard::comm_id(), else the host's nb_gather_fn).
class Shard {
public:
    static std::array<char, NB_COMM_ID_BYTES> comm_id()
    {
        std::array<char, NB_COMM_ID_BYTES> id{};
        check(nb_comm_id(id.data()), nullptr);
        return id;
    }
    Shard(const std::vector<Vec3> &positions, const std::vector<Vec3> &velocities, int rank, int world, const nb_params &params,
          const void *rccl_id = nullptr, nb_gather_fn gather = nullptr, void *gather_user = nullptr, nb_ring_fn ring = nullptr,
          void *ring_user = nullptr)
    {
        check_abi();
        if (positions.empty() || positions.size() != velocities.size())
            throw std::invalid_argument("positions and velocities must be non-empty and equally long");
        check(nb_shard_create((uint32_t)positions.size(), rank, world, &params, &sh_), nullptr);
        n_ = (uint32_t)positions.size();
        try {
            check_sh(nb_shard_range(sh_, &first_, &count_));
            if (rccl_id) check_sh(nb_shard_use_rccl(sh_, rccl_id));
            else if (gather) check_sh(nb_shard_use_gather(sh_, gather, gather_user));
            // FAST on equal ranks: every unordered pair once, with a second exchange per step (include/nenbody.h); RCCL brings
            // it along, a host exchange needs `ring` beside `gather` (else the ordered fold stays)
            if (ring) check_sh(nb_shard_use_ring(sh_, ring, ring_user));
            check_sh(nb_shard_upload(sh_, positions[0].data(), velocities[0].data()));
        } catch (...) {
            nb_shard_destroy(sh_);
            throw;
        }
    }
    Shard(const Shard &) = delete;
    Shard &operator=(const Shard &) = delete;
    ~Shard() { nb_shard_destroy(sh_); }

    uint32_t first() const { return first_; }
    uint32_t count() const { return count_; }
    int pairs_partners() const { return nb_shard_pairs_partners(sh_); }  // D of the pairs form a step will take; 0: the ordered fold
    bool pairs_overlapped() const { return nb_shard_pairs_overlapped(sh_) == 1; }  // the pairs form in phases, both exchanges on the second stream (set_overlap)
    void step(uint32_t k = 1) { check_sh(nb_shard_step(sh_, k)); }
    void step_boids(uint32_t k = 1, const nb_boids_params *params = nullptr) { check_sh(nb_shard_step_boids(sh_, k, params)); }
    // FAST only (a STRICT shard ignores it): fold the rank's own slot while the exchange of the others is in flight
    void set_overlap(bool on) { check_sh(nb_shard_set_overlap(sh_, on ? 1 : 0)); }
    // both exchanges once on a known pattern, checked on every rank, with fallbacks (collective).  gather: 0 in place, 1 from a copy;
    // ring: -1 no pairs form, 0 one group, 1 one group per distance, 2 dropped for the ordered fold
    struct ExchangePaths {
        int gather, ring;
    };
    ExchangePaths verify_exchanges()  // (gather 2 / ring 3: pulled over xGMI, see peer_import)
    {
        ExchangePaths p{0, -1};
        check_sh(nb_shard_verify_exchanges(sh_, &p.gather, &p.ring));
        return p;
    }
    // every form a FAST step can take timed on this machine (slowest rank; collective), the fastest kept, the state put back:
    // 0 the ordered fold, 1 the pairs form, 2 the pairs form in phases; ms_per_step[form] < 0: not offered
    struct FormChoice {
        int chosen;
        double ms_per_step[3];
    };
    FormChoice choose_form(uint32_t steps = 4)
    {
        FormChoice c{};
        check_sh(nb_shard_choose_form(sh_, steps, &c.chosen, c.ms_per_step));
        return c;
    }
    void set_pairs(bool on) { check_sh(nb_shard_set_pairs(sh_, on ? 1 : 0)); }  // name the form (run-to-run identical FAST bits)
    // both exchanges as pulls over xGMI, no collective library (include/nenbody.h): every rank's peer_blob() travels to every rank by the
    // host's own channel, rank-major, into peer_import(); all ranks in processes of one node
    std::vector<unsigned char> peer_blob()
    {
        std::vector<unsigned char> b(nb_peers_blob_bytes());
        check_sh(nb_shard_peer_export(sh_, b.data()));
        return b;
    }
    void peer_import(const std::vector<unsigned char> &all_blobs_rank_major) { check_sh(nb_shard_peer_import(sh_, all_blobs_rank_major.data())); }
    void use_peers(bool on) { check_sh(nb_shard_use_peers(sh_, on ? 1 : 0)); }
    // waits; throws Error(NB_ERR_STATE) if a kernel of this shard reported a failure (a block-chain wave gave up waiting)
    void sync() { check_sh(nb_shard_sync(sh_)); }
    // all n positions (the replica); this rank's velocities and model matrices
    void download(std::vector<Vec3> &positions, std::vector<Vec3> &velocities_local, std::vector<Mat4> &instances_local)
    {
        positions.resize(n_);
        velocities_local.resize(count_);
        instances_local.resize(count_);
        check_sh(nb_shard_download(sh_, positions[0].data(), count_ ? velocities_local[0].data() : nullptr,
                                   count_ ? instances_local[0][0].data() : nullptr));
    }

private:
    void check_sh(int rc)
    {
        if (rc != NB_OK) throw Error(rc, nb_shard_last_error(sh_));
    }
    nb_shard *sh_ = nullptr;
    uint32_t n_ = 0, first_ = 0, count_ = 0;
};

// Drop-ins for the reference's free functions, src/main.rs:404-410 and 443-449: same five arguments, updated in place,
// one FFI call each (nb_update_instance_nbody / nb_update_instance_boids keep the device context between calls).
// A length mismatch is where copy_from_slice panics (src/main.rs:415-416): std::invalid_argument here.
namespace detail {
inline void update_status(int rc)
{
    if (rc == NB_ERR_INVALID) throw std::invalid_argument(nb_last_error(nullptr));
    check(rc, nullptr);
}
inline float *ptr(std::vector<Vec3> &v) { return v.empty() ? nullptr : v[0].data(); }
inline float *ptr(std::vector<Mat4> &v) { return v.empty() ? nullptr : v[0][0].data(); }
}  // namespace detail

inline void update_instance_nbody(std::vector<Mat4> &instances, std::vector<Vec3> &positions, std::vector<Vec3> &old_positions,
                                  std::vector<Vec3> &velocities, std::vector<Vec3> &old_velocities,
                                  const nb_params *params = nullptr)
{
    check_abi();
    detail::update_status(nb_update_instance_nbody(detail::ptr(instances), instances.size(), detail::ptr(positions),
                                                   positions.size(), detail::ptr(old_positions), old_positions.size(),
                                                   detail::ptr(velocities), velocities.size(), detail::ptr(old_velocities),
                                                   old_velocities.size(), params));
}

inline void update_instance_boids(std::vector<Mat4> &instances, std::vector<Vec3> &positions, std::vector<Vec3> &old_positions,
                                  std::vector<Vec3> &velocities, std::vector<Vec3> &old_velocities,
                                  const nb_boids_params *params = nullptr)
{
    check_abi();
    detail::update_status(nb_update_instance_boids(detail::ptr(instances), instances.size(), detail::ptr(positions),
                                                   positions.size(), detail::ptr(old_positions), old_positions.size(),
                                                   detail::ptr(velocities), velocities.size(), detail::ptr(old_velocities),
                                                   old_velocities.size(), params));
}

// update_instance_random(instances, positions, velocities) (src/main.rs:381-385): the third controller, its own three
// arguments; the library keeps the seed and counts the calls (nb_update_random_seed restarts the stream)
inline void update_instance_random(std::vector<Mat4> &instances, std::vector<Vec3> &positions, std::vector<Vec3> &velocities)
{
    check_abi();
    detail::update_status(nb_update_instance_random(detail::ptr(instances), instances.size(), detail::ptr(positions), positions.size(),
                                                    detail::ptr(velocities), velocities.size()));
}

// the same step at a stream position the caller names: `step` = the frame number
inline void update_instance_random(std::vector<Mat4> &instances, std::vector<Vec3> &positions, std::vector<Vec3> &velocities,
                                   uint64_t seed, uint64_t step)
{
    check_abi();
    detail::update_status(nb_update_instance_random_seeded(detail::ptr(instances), instances.size(), detail::ptr(positions),
                                                           positions.size(), detail::ptr(velocities), velocities.size(), seed, step));
}

}  // namespace nenbody
