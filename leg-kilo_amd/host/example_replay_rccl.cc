// Recorded-run replay sharded over all GPUs of one node, driven from C++ without Python: ONE process, one lk_handle and one RCCL
// rank per visible gfx950 device (ncclCommInitAll), one host thread per device.  Device 0 builds the map (first-frame BuildVoxelMap
// on a synthetic floor), the blob goes to every other GPU over xGMI (legkilo::broadcastMap: scatter + all-gather), every GPU
// replays its block of the scans against the now shared, frozen map (lk_batch_replay_scans_dev), and the per-scan result records -
// pose with counters, and state + covariance - are all-gathered on every GPU.  Needs >= 1 gfx950 device to RUN (exit code 3
// otherwise; with one device it degenerates to world = 1); tests/test_abi_and_host.py checks that it compiles and links against
// the C-ABI and rccl.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

#include "legkilo_host.hpp"
#include "legkilo_rccl.hpp"

using namespace legkilo;

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        std::fprintf(stderr, "no HIP device\n");
        return 3;
    }
    const int world = ndev;
    const size_t n_scans = 64, n_pts = 4000;
    // the recorded run: scans of a floor seen from slightly different poses, two time buckets each
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-4.f, 4.f);
    std::normal_distribution<float> nz(0.f, 0.01f);
    std::vector<lk_point> pts(n_scans * n_pts);
    std::vector<uint64_t> scan_off(n_scans + 1, 0);
    std::vector<double> t_begin(n_scans, 0.0), x36(n_scans * LK_STATE_DOUBLES, 0.0), P900(n_scans * 900, 0.0);
    for (size_t s = 0; s < n_scans; ++s) {
        for (size_t i = 0; i < n_pts; ++i) pts[s * n_pts + i] = lk_point{u(rng), u(rng), nz(rng) - 0.7f, i < n_pts / 2 ? 0.f : 0.002f};
        scan_off[s + 1] = (s + 1) * n_pts;
        double* x = &x36[s * LK_STATE_DOUBLES];
        x[0] = x[4] = x[8] = 1.0;                       // rotation
        x[9] = 0.001 * (double)s, x[11] = 0.5;          // position
        x[23] = -9.81;                                  // gravity (rot 0..8, pos 9, vel 12, ba 15, bw 18, grav 21..23)
        for (int i = 0; i < 30; ++i) P900[s * 900 + 31 * i] = 1e-4;
    }
    std::vector<int> devs(world);
    for (int i = 0; i < world; ++i) devs[i] = i;
    std::vector<ncclComm_t> comms(world);
    try {
        rcclCheck(ncclCommInitAll(comms.data(), world, devs.data()), "ncclCommInitAll");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
    std::vector<int> ok(world, 0);
    std::vector<std::thread> th;
    for (int rank = 0; rank < world; ++rank)
        th.emplace_back([&, rank]() {
            try {
                hipCheckRt(hipSetDevice(rank), "hipSetDevice");
                ESKF::Config ec{20, 500, 1000, 20, 0.001, 0.001, 0.001, 0.1, 1.0, 0.01, 0.1, 0.1, 0.001, 10};
                VoxelMapConfig vc;
                DeviceCaps caps;
                caps.device_id = rank;
                size_t a = 0, b = 0;
                shardRange(n_scans, rank, world, &a, &b);
                const size_t n_local = b - a, n_max = (n_scans + world - 1) / world;
                caps.n_slots = (uint32_t)std::max<size_t>(n_max, 1);
                caps.max_roots = 1u << 14, caps.max_nodes = 1u << 15, caps.max_point_blocks = 1u << 14, caps.max_scan_points = 1u << 15;
                KiloPath kilo(ec, vc, Mat3D::Identity(), Vec3D{0, 0, 0.2}, 9.81, caps);
                lk_handle* h = kilo.device().h();
                kilo.eskf().initProcessCovQ();
                if (rank == 0) {   // the map: first-frame BuildVoxelMap on device 0
                    State s0;
                    s0.pos_ = {0, 0, 0.5};
                    kilo.eskf().setState(s0);
                    StateCov P0;
                    for (int i = 0; i < DIM_STATE; ++i) P0(i, i) = 1e-6;
                    kilo.eskf().setCov(P0);
                    auto body = std::make_shared<PointCloudType>(), wcloud = std::make_shared<PointCloudType>();
                    std::mt19937 r2(1);
                    for (int i = 0; i < 20000; ++i) {
                        PointType w;
                        w.x = u(r2), w.y = u(r2), w.z = nz(r2);
                        PointType bp = w;
                        bp.z = w.z - 0.7f;
                        wcloud->push_back(w), body->push_back(bp);
                    }
                    kilo.map_manager().feats_down_body_ = body;
                    kilo.map_manager().feats_down_world_ = wcloud;
                    kilo.map_manager().BuildVoxelMap(kilo.eskf().getRot(), kilo.eskf().getRotCov(), kilo.eskf().getPosCov());
                }
                const size_t map_bytes = broadcastMap(h, comms[rank], rank, world, 0, MapTransport::scatter_allgather);
                // this rank's block of the run (every rank replays n_max slots so that the all-gathers are uniform; a short last
                // block repeats its last scan)
                std::vector<lk_point> lp;
                std::vector<uint64_t> loff(1, 0);
                std::vector<double> ltb, lx, lP;
                for (size_t k = 0; k < n_max; ++k) {
                    const size_t s = std::min(a + k, b > a ? b - 1 : a);
                    lp.insert(lp.end(), pts.begin() + scan_off[s], pts.begin() + scan_off[s + 1]);
                    loff.push_back(lp.size());
                    ltb.push_back(t_begin[s]);
                    lx.insert(lx.end(), x36.begin() + s * LK_STATE_DOUBLES, x36.begin() + (s + 1) * LK_STATE_DOUBLES);
                    lP.insert(lP.end(), P900.begin() + s * 900, P900.begin() + (s + 1) * 900);
                }
                void* d_pts = nullptr;
                lkCheck(h, lk_device_malloc(h, &d_pts, sizeof(lk_point) * lp.size()), "lk_device_malloc");
                lkCheck(h, lk_memcpy_h2d(h, d_pts, lp.data(), sizeof(lk_point) * lp.size()), "lk_memcpy_h2d");
                lkCheck(h, lk_batch_set_priors(h, lx.data(), lP.data(), n_max), "lk_batch_set_priors");
                std::vector<lk_pose> poses(n_max);
                lkCheck(h, lk_batch_replay_scans_dev(h, static_cast<const lk_point*>(d_pts), n_max, loff.data(), ltb.data(), 0, nullptr, nullptr, poses.data()),
                        "lk_batch_replay_scans_dev");
                // result records of all ranks on every rank: poses, and state + covariance
                lk_pose *d_pl = nullptr, *d_pa = nullptr;
                double *d_xl = nullptr, *d_Pl = nullptr, *d_xa = nullptr, *d_Pa = nullptr;
                hipCheckRt(hipMalloc(&d_pl, sizeof(lk_pose) * n_max), "hipMalloc");
                hipCheckRt(hipMalloc(&d_pa, sizeof(lk_pose) * n_max * world), "hipMalloc");
                hipCheckRt(hipMalloc(&d_xl, sizeof(double) * 36 * n_max), "hipMalloc");
                hipCheckRt(hipMalloc(&d_Pl, sizeof(double) * 900 * n_max), "hipMalloc");
                hipCheckRt(hipMalloc(&d_xa, sizeof(double) * 36 * n_max * world), "hipMalloc");
                hipCheckRt(hipMalloc(&d_Pa, sizeof(double) * 900 * n_max * world), "hipMalloc");
                lkCheck(h, lk_memcpy_h2d(h, d_pl, poses.data(), sizeof(lk_pose) * n_max), "lk_memcpy_h2d");
                allGatherPoses(h, comms[rank], d_pl, d_pa, n_max);
                allGatherStates(h, comms[rank], 0, n_max, d_xl, d_Pl, d_xa, d_Pa);
                std::vector<lk_pose> all(n_max * world);
                std::vector<double> Pall(900 * n_max * world);
                lkCheck(h, lk_memcpy_d2h(h, all.data(), d_pa, sizeof(lk_pose) * all.size()), "lk_memcpy_d2h");
                lkCheck(h, lk_memcpy_d2h(h, Pall.data(), d_Pa, sizeof(double) * Pall.size()), "lk_memcpy_d2h");
                unsigned long long matched = 0;
                for (const lk_pose& p : all) matched += p.n_effect;
                // the same block WITH the map insert of KILO::process (KILO.cc:216-233 after every bucket), each scan on its own copy-on-write
                // overlay of the shared map (lk_batch_replay_overlay_ragged_dev): no collective on the data path either - an overlay is private
                // to its scan - and the same all-gather of the result records
                std::vector<uint32_t> nbk, boff;
                std::vector<double> bdt;
                for (size_t k = 0; k < n_max; ++k) {   // runs of equal curvature = buckets (KILO.cc:375-378)
                    const size_t p0 = loff[k], p1 = loff[k + 1];
                    uint32_t nb = 0;
                    boff.push_back(0);
                    for (size_t i = p0; i < p1;) {
                        size_t j = i + 1;
                        while (j < p1 && lp[i].curvature == lp[j].curvature) ++j;
                        bdt.push_back((double)lp[i].curvature), boff.push_back((uint32_t)(j - p0)), ++nb;
                        i = j;
                    }
                    nbk.push_back(nb);
                }
                lkCheck(h, lk_batch_set_priors(h, lx.data(), lP.data(), n_max), "lk_batch_set_priors");
                lkCheck(h, lk_batch_replay_overlay_ragged_dev(h, static_cast<const lk_point*>(d_pts), n_max, loff.data(), nbk.data(), boff.data(), bdt.data(), ltb.data(),
                                                              nullptr, nullptr, 0, poses.data()),
                        "lk_batch_replay_overlay_ragged_dev");
                lkCheck(h, lk_memcpy_h2d(h, d_pl, poses.data(), sizeof(lk_pose) * n_max), "lk_memcpy_h2d");
                allGatherPoses(h, comms[rank], d_pl, d_pa, n_max);
                lkCheck(h, lk_memcpy_d2h(h, all.data(), d_pa, sizeof(lk_pose) * all.size()), "lk_memcpy_d2h");
                unsigned long long matched_ins = 0;
                for (const lk_pose& p : all) matched_ins += p.n_effect;
                if (rank == 0)
                    std::printf("world %d: map %zu bytes to every GPU, %zu scans replayed (%zu local), %llu matched points (frozen map), %llu with insert, P[0][0] of the last scan %.3e\n",
                                world, map_bytes, n_scans, n_local, matched, matched_ins, Pall[900 * (all.size() - 1)]);
                ok[rank] = matched > 0 && matched_ins > 0;
                for (void* p : {(void*)d_pl, (void*)d_pa, (void*)d_xl, (void*)d_Pl, (void*)d_xa, (void*)d_Pa}) hipFree(p);
                lk_device_free(h, d_pts);
            } catch (const std::exception& e) {
                std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
            }
        });
    for (auto& t : th) t.join();
    for (int i = 0; i < world; ++i) ncclCommDestroy(comms[i]);
    return std::all_of(ok.begin(), ok.end(), [](int v) { return v != 0; }) ? 0 : 1;
}
