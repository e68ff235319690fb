// The recipe of INTEGRATION.md section 4 as a program: one process per GPU, the fleet cut into
// contiguous blocks (nmpc_shard_range), one SQP-RTI solve of this rank's block and one RCCL all-gather
// of the control records behind the C ABI (nmpc_gather_controls).  What the reference's C++ caller
// (/root/reference/src/PathTrackingSys.cpp:58-61 holds ONE solver for ONE vehicle) would do to hold the
// whole fleet's controls on every GPU.
//
//   build:  g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/multi_gpu_gather.cpp
//               -L mpc_local_planner_amd -lnmpc_hip -L /opt/rocm/lib -lamdhip64 -lrccl -o examples/multi_gpu_gather
//   run:    one rank:  ./examples/multi_gpu_gather            (rank 0 of 1, creates its own ncclUniqueId)
//           N ranks:   start N copies with NMPC_RANK / NMPC_WORLD set and NMPC_ID_FILE naming a file on
//                      a shared path: rank 0 writes the ncclUniqueId there, the others read it
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "mpc_local_planner_amd/AcadosSolver.hpp"

#define CHECK(call)                                                                       \
    do {                                                                                  \
        const int rc_ = (int)(call);                                                      \
        if (rc_ != 0) {                                                                   \
            std::fprintf(stderr, "%s failed (%d) at line %d: %s\n", #call, rc_, __LINE__, nmpc_last_error()); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

int main()
{
    const int rank = std::getenv("NMPC_RANK") ? std::atoi(std::getenv("NMPC_RANK")) : 0;
    const int world = std::getenv("NMPC_WORLD") ? std::atoi(std::getenv("NMPC_WORLD")) : 1;
    const char *id_file = std::getenv("NMPC_ID_FILE");
    const int64_t fleet = 4096LL * world;     // equal blocks: the gather takes the same B on every rank
    int ndev = 0;
    CHECK(hipGetDeviceCount(&ndev));
    const int device = rank % ndev;
    CHECK(hipSetDevice(device));

    ncclUniqueId id;
    if (rank == 0) {
        CHECK(ncclGetUniqueId(&id));
        if (world > 1) {
            if (!id_file) { std::fprintf(stderr, "NMPC_ID_FILE is needed for more than one rank\n"); return 1; }
            FILE *f = std::fopen(id_file, "wb");
            if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) return 1;
            std::fclose(f);
        }
    } else {
        FILE *f = nullptr;
        for (int tries = 0; tries < 600 && !(f = std::fopen(id_file, "rb")); ++tries)
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!f || std::fread(&id, sizeof(id), 1, f) != 1) return 1;
        std::fclose(f);
    }
    ncclComm_t comm;
    CHECK(ncclCommInitRank(&comm, world, id, rank));

    int64_t first = 0, count = 0;
    CHECK(nmpc_shard_range(fleet, rank, world, &first, &count));
    const int B = (int)count;

    // this rank's vehicles: on a straight path y = 0, offset sideways and in heading by a pattern of the
    // GLOBAL vehicle index, so that every rank can check every other rank's records
    std::vector<double> x0((size_t)B * 6), p((size_t)B * 6, 0.0);
    for (int i = 0; i < B; ++i) {
        const int64_t g = first + i;
        double *x = &x0[(size_t)i * 6];
        x[0] = 0.0; x[1] = 0.02 * (double)(g % 50) - 0.5; x[2] = 0.001 * (double)(g % 7); x[3] = 3.0 + 0.01 * (double)(g % 11);
        x[4] = 0.0; x[5] = 0.0;
        p[(size_t)i * 6 + 5] = 4.0;             // v_ref; c0..c3 = 0: the path is the X axis
    }
    nmpc_config cfg;
    nmpc_default_config(&cfg, 40, 0.35);
    mpc::State s0{};
    mpc::Acados::PathTracking solver{s0, cfg, B, device};
    double *d_all = nullptr;                   // [world][B][6] on this GPU
    CHECK(hipMalloc(&d_all, sizeof(double) * 6 * (size_t)B * world));

    CHECK(nmpc_set_x0(solver.handle(), x0.data(), B));
    CHECK(nmpc_set_params(solver.handle(), p.data(), B));
    CHECK(nmpc_solve(solver.handle(), B, NMPC_MODE_RTI, 1, 0.0));
    CHECK(nmpc_gather_controls(solver.handle(), comm, B, d_all));     // ncclAllGather on the handle's stream
    CHECK(nmpc_sync(solver.handle()));

    std::vector<double> all((size_t)6 * B * world), own((size_t)6 * B);
    CHECK(hipMemcpy(all.data(), d_all, sizeof(double) * all.size(), hipMemcpyDeviceToHost));
    CHECK(nmpc_get_controls(solver.handle(), B, own.data()));
    // this rank's block of the gathered array is its own records; all records are successful solves
    if (std::memcmp(&all[(size_t)rank * B * 6], own.data(), sizeof(double) * own.size()) != 0) {
        std::fprintf(stderr, "rank %d: gathered block differs from the local records\n", rank);
        return 2;
    }
    int bad = 0;
    for (size_t i = 0; i < (size_t)B * world; ++i) {
        const double code = all[i * 6 + 5];                 // status + 16 * interior-point iterations
        if (std::fmod(code, 16.0) != 0.0 || !(std::fabs(all[i * 6 + 2]) <= 0.57 + 1e-9)) ++bad;
    }
    std::printf("rank %d of %d: vehicles [%lld, %lld), %d records gathered, %d not ok, first u0 = (%.6f, %.6f)\n", rank, world,
                (long long)first, (long long)(first + count), B * world, bad, all[0], all[1]);
    (void)hipFree(d_all);
    ncclCommDestroy(comm);
    return bad ? 3 : 0;
}
