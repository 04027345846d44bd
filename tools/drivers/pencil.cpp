// pencil -- the reference's `pencil` test executable (tests/src/pencil/main.cpp:26-236) on the MI355X library:
//   mpiexec -n P ./pencil -nx 256 -ny 256 -nz 256 -p1 2 -p2 2 -o 1 -t 3 -i 10 -w 2 -d [-c] [-b dir]
// Same flags, same testcases (0 forward, 1 coordinator + P workers, 2 inverse, 3 round trip, 4 Laplacian), same
// "Result (avg) / Result (max)" lines and the same timer CSV under <benchmark_dir>/pencil/.  Build: make -C tools drivers.
#include <memory>

#include "driver_common.hpp"

using namespace driver;

static void printHelp()
{
    printf("Usage: mpirun -n P [mpi args] pencil [options] \n");
    printf("Options (required):\n");
    printf(" --input-dim-x [-nx], --input-dim-y [-ny], --input-dim-z [-nz]: size of the global grid\n");
    printf(" --partition1 [-p1], --partition2 [-p2]: partitions in x- and y-direction (P1*P2 = P; testcase 1: P1*P2+1 = P)\n");
    printf("Options (optional):\n");
    printf(" --comm-method1 [-comm1], --comm-method2 [-comm2]: \"Peer2Peer\" or \"All2All\" (recorded in the CSV name; every exchange is one\n");
    printf("                        grouped all-to-all, on RCCL with --cuda_aware, staged through pinned host memory and MPI_Alltoallv without)\n");
    printf(" --send-method1 [-snd1], --send-method2 [-snd2]: \"Sync\", \"Streams\" or \"MPI_Type\" (recorded only)\n");
    printf(" --testcase [-t]: 0 forward (default), 1 coordinator compares with a single-GPU transform, 2 inverse, 3 round trip, 4 Laplacian\n");
    printf(" --opt [-o]: 0 = MPIcuFFT_Pencil, 1 = MPIcuFFT_Pencil_Opt1\n");
    printf(" --fft-dim [-f]: 1, 2 or 3 (default) dimensions\n");
    printf(" --iterations [-i], --warmup-rounds [-w], --cuda_aware [-c], --double_prec [-d], --benchmark_dir [-b]\n");
}

struct PencilParams : Common {
    size_t P1 = 1, P2 = 1;
    CommunicationMethod comm_method1 = Peer2Peer, comm_method2 = Peer2Peer;
    SendMethod send_method1 = Sync, send_method2 = Sync;
};

template <typename T> static int run(const PencilParams &p, const World &w)
{
    Configurations config = {p.cuda_aware, p.warmup_rounds, p.comm_method1, p.send_method1, p.benchmark_dir, p.comm_method2, p.send_method2};
    if (p.fft_dim != 3 && (p.testcase == 1 || p.testcase == 4)) throw std::runtime_error("testcases 1 and 4 need --fft-dim 3");
    std::function<PlanOps<T>(MPI_Comm, int)> make = [&](MPI_Comm comm, int max_world) {
        std::shared_ptr<MPIcuFFT_Pencil<T>> plan;
        if (p.opt == 1) plan = std::make_shared<MPIcuFFT_Pencil_Opt1<T>>(config, comm, max_world);
        else plan = std::make_shared<MPIcuFFT_Pencil<T>>(config, comm, max_world);
        int csize = 1;
        MPI_Comm_size(comm, &csize);
        Pencil_Partition partition(csize == 1 ? 1 : p.P1, csize == 1 ? 1 : p.P2);
        GlobalSize global_size(p.Nx, p.Ny, p.Nz);
        plan->initFFT(&global_size, &partition, true);
        PlanOps<T> ops;
        fillSizes<T>(plan.get(), ops);
        const int d = p.fft_dim;
        ops.forward = [plan, d](void *out, const void *in) { plan->execR2C(out, in, d); };
        ops.inverse = [plan, d](void *out, const void *in) { plan->execC2R(out, in, d); };
        return ops;
    };
    return runTestcase<T>(make, p, w);
}

int main(int argc, char *argv[])
{
    if (argc == 1 || (argc == 2 && (std::string(argv[1]) == "--help" || std::string(argv[1]) == "-h"))) {
        printHelp();
        return 0;
    }
    try {
        PencilParams p;
        parseCommon(argc, argv, p);
        p.P1 = as_size(arg_value(argc, argv, "--partition1", "-p1"), true, "Input parameter P1 is required.");
        p.P2 = as_size(arg_value(argc, argv, "--partition2", "-p2"), true, "Input parameter P2 is required.");
        p.fft_dim = as_int(arg_value(argc, argv, "--fft-dim", "-f"));
        if (p.fft_dim == 0) p.fft_dim = 3;
        else if (p.fft_dim < 0 || p.fft_dim > 3) throw std::runtime_error("Invalid FFT dimension.");
        p.comm_method1 = comm_method_named(arg_value(argc, argv, "--comm-method1", "-comm1"));
        p.comm_method2 = comm_method_named(arg_value(argc, argv, "--comm-method2", "-comm2"));
        p.send_method1 = send_method_named(arg_value(argc, argv, "--send-method1", "-snd1"));
        p.send_method2 = send_method_named(arg_value(argc, argv, "--send-method2", "-snd2"));
        const int need = (int)(p.P1 * p.P2) + (p.testcase == 1 ? 1 : 0);
        if (getenv("DFFT_DRIVER_PARSE_ONLY")) {      // what the command line means, before MPI or the GPU are touched (tests/test_launch_commands.py)
            printf("PARSED pencil nx=%zu ny=%zu nz=%zu p1=%zu p2=%zu t=%d o=%d runs=%d w=%d c=%d d=%d f=%d comm1=%d snd1=%d comm2=%d snd2=%d b=%s ranks=%d\n",
                   p.Nx, p.Ny, p.Nz, p.P1, p.P2, p.testcase, p.opt, p.iterations, p.warmup_rounds, (int)p.cuda_aware, (int)p.double_prec, p.fft_dim,
                   (int)p.comm_method1, (int)p.send_method1, (int)p.comm_method2, (int)p.send_method2, p.benchmark_dir.c_str(), need);
            return 0;
        }
        World w(p.cuda_aware);
        if (w.size != need) throw std::runtime_error("P1*P2 (+1 for testcase 1) must equal the number of MPI ranks.");
        return p.double_prec ? run<double>(p, w) : run<float>(p, w);
    } catch (std::runtime_error &e) {
        printf("%s\n\n", e.what());
        printf("Use \"--help\" or \"-h\" to display the help menu.\n");
        return 1;
    }
}
