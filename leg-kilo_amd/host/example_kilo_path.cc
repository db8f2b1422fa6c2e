// Minimal use of the C++ mirror, written the way KILO.cc drives the reference classes: first-frame
// BuildVoxelMap on a synthetic floor + wall, one predictUpdatePoint bucket, then a two-scan recorded-run replay (frozen map) and the same
// two scans with the per-scan map insert (overlay replay).  Needs a gfx950 device
// to RUN (exit code 3 otherwise); tests/test_abi_and_host.py only checks that it compiles and links.  With a path as argv[1] it
// dumps its inputs and results (a flat binary: counts, the two clouds, x36 after the bucket, the match count, the replay poses) so
// that tests/test_golden.py can replay the SAME inputs through the oracle and compare state and counts, not just sanity.
#include <cmath>
#include <cstdio>
#include <random>

#include "legkilo_host.hpp"

using namespace legkilo;

int main(int argc, char** argv) {
    ESKF::Config ec{20, 500, 1000, 20, 0.001, 0.001, 0.001, 0.1, 1.0, 0.01, 0.1, 0.1, 0.001, 10};
    VoxelMapConfig vc;
    DeviceCaps caps;
    caps.n_slots = 2;   // replayRecordedRun below replays two scans at once
    caps.max_roots = 1u << 14, caps.max_nodes = 1u << 15, caps.max_point_blocks = 1u << 14, caps.max_scan_points = 1u << 15;
    std::unique_ptr<KiloPath> kilo;
    try {
        kilo = std::make_unique<KiloPath>(ec, vc, Mat3D::Identity(), Vec3D{0, 0, 0.2}, 9.81, caps);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "no device: %s\n", e.what());
        return 3;
    }
    State s;
    s.pos_ = {0, 0, 0.5};
    kilo->eskf().setState(s);
    StateCov P;
    for (int i = 0; i < DIM_STATE; ++i) P(i, i) = 1e-6;
    kilo->eskf().setCov(P);
    kilo->eskf().initProcessCovQ();
    kilo->setTimes(0.0, 0.0);

    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-4.f, 4.f);
    std::normal_distribution<float> nz(0.f, 0.01f);
    auto body = std::make_shared<PointCloudType>(), world = std::make_shared<PointCloudType>();
    for (int i = 0; i < 20000; ++i) {  // floor z=0 (world) seen from the sensor at z = 0.5 + 0.2
        PointType w;
        w.x = u(rng), w.y = u(rng), w.z = nz(rng);
        PointType b = w;
        b.z = w.z - 0.7f;
        world->push_back(w), body->push_back(b);
    }
    kilo->map_manager().feats_down_body_ = body;
    kilo->map_manager().feats_down_world_ = world;
    kilo->map_manager().BuildVoxelMap(kilo->eskf().getRot(), kilo->eskf().getRotCov(), kilo->eskf().getPosCov());

    PointCloudType bucket(body->begin(), body->begin() + 2000), bucket_world(2000);
    size_t n_success = 0;
    bool updated = kilo->predictUpdatePoint(0.01, 0, bucket.size(), bucket, bucket_world, n_success);
    Vec3D p = kilo->eskf().getPos();
    std::printf("updated=%d matched=%zu pos=(%.4f %.4f %.4f)\n", (int)updated, n_success, p[0], p[1], p[2]);
    double x36[36];   // the bucket's posterior (slot 0 is re-used by the replay below)
    kilo->eskf().state().to_x36(x36);
    // the same bucket twice more as a two-scan "recorded run", each from its own prior, against the (now frozen) map
    PointCloudType scan(bucket.begin(), bucket.end());
    for (size_t i = 0; i < scan.size(); ++i) scan[i].curvature = (i < scan.size() / 2) ? 0.f : 0.002f;   // two time buckets
    const auto post_state = kilo->eskf().state();   // copies: the replays below run on the filter slots, slot 0 included
    const auto post_cov = kilo->eskf().cov();
    std::vector<lk_pose> poses = kilo->replayRecordedRun({scan, scan}, {0.02, 0.05}, {post_state, s}, {post_cov, P});
    std::printf("replay: %u / %u buckets, matched %llu / %llu\n", poses[0].n_buckets, poses[1].n_buckets,
                (unsigned long long)poses[0].n_effect, (unsigned long long)poses[1].n_effect);
    // and once more WITH the map insert after every bucket, each scan on its own copy-on-write overlay of the map (KILO::process's
    // own order of events, for a batch): the map itself stays as it is
    std::vector<lk_pose> ov = kilo->replayWithInsert({scan, scan}, 0.02, {post_state, s}, {post_cov, P});
    std::printf("replay with insert: matched %llu / %llu\n", (unsigned long long)ov[0].n_effect, (unsigned long long)ov[1].n_effect);
    poses.insert(poses.end(), ov.begin(), ov.end());
    if (argc > 1) {
        FILE* f = std::fopen(argv[1], "wb");
        if (!f) return 4;
        const unsigned int hdr[4] = {(unsigned int)body->size(), (unsigned int)bucket.size(), (unsigned int)n_success, (unsigned int)poses.size()};
        std::fwrite(hdr, sizeof(hdr), 1, f);
        for (const auto* c : {world.get(), body.get()})
            for (const PointType& q : *c) {
                const float xyz[3] = {q.x, q.y, q.z};
                std::fwrite(xyz, sizeof(xyz), 1, f);
            }
        std::fwrite(x36, sizeof(x36), 1, f);
        std::fwrite(poses.data(), sizeof(lk_pose), poses.size(), f);
        std::fclose(f);
    }
    const bool replay_ok = poses.size() == 4 && poses[0].n_buckets == 2 && poses[1].n_buckets == 2 && poses[1].n_effect > 500 && poses[3].n_buckets == 2 && poses[3].n_effect > 500;
    return (updated && n_success > 500 && std::fabs(p[2] - 0.5) < 0.05 && replay_ok) ? 0 : 1;
}
