// slab -- the reference's `slab` test executable (tests/src/slab/main.cpp:26-214) on the MI355X library:
//   mpiexec -n P ./slab -nx 256 -ny 256 -nz 256 -s Z_Then_YX -o 1 -t 3 -i 10 -w 2 -d [-c] [-b dir]
// Same flags, sequences ZY_Then_X (default), Z_Then_YX and Y_Then_ZX (forward only, opt 0), the same testcases and outputs and the
// timer CSV under <benchmark_dir>/slab_default | slab_z_then_yx | slab_y_then_zx.  Build: make -C tools drivers.
#include <memory>

#include "driver_common.hpp"

using namespace driver;

static void printHelp()
{
    printf("Usage: mpirun -n P [mpi args] slab [options] \n");
    printf("Options (required):\n");
    printf(" --input-dim-x [-nx], --input-dim-y [-ny], --input-dim-z [-nz]: size of the global grid\n");
    printf("Options (optional):\n");
    printf(" --sequence [-s]: \"ZY_Then_X\" (default), \"Z_Then_YX\" or \"Y_Then_ZX\"\n");
    printf(" --comm-method [-comm]: \"Peer2Peer\" or \"All2All\"; --send-method [-snd]: \"Sync\", \"Streams\" or \"MPI_Type\" (recorded in the CSV name)\n");
    printf(" --testcase [-t]: 0 forward (default), 1 coordinator compares with a single-GPU transform (P+1 ranks), 2 inverse, 3 round trip, 4 Laplacian\n");
    printf(" --opt [-o]: 0 default classes, 1 the _Opt1 classes\n");
    printf(" --iterations [-i], --warmup-rounds [-w], --cuda_aware [-c], --double_prec [-d], --benchmark_dir [-b]\n");
}

struct SlabParams : Common {
    std::string sequence;
    CommunicationMethod comm_method = Peer2Peer;
    SendMethod send_method = Sync;
};

template <typename T, typename Plan> static PlanOps<T> opsOf(std::shared_ptr<Plan> plan, const SlabParams &p, bool has_inverse)
{
    GlobalSize global_size(p.Nx, p.Ny, p.Nz);
    plan->initFFT(&global_size, true);
    PlanOps<T> ops;
    fillSizes<T>(plan.get(), ops);
    ops.has_inverse = has_inverse;
    ops.forward = [plan](void *out, const void *in) { plan->execR2C(out, in); };
    ops.inverse = [plan](void *out, const void *in) { plan->execC2R(out, in); };
    return ops;
}

template <typename T> static int run(const SlabParams &p, const World &w)
{
    Configurations config = {p.cuda_aware, p.warmup_rounds, p.comm_method, p.send_method, p.benchmark_dir, p.comm_method, p.send_method};
    std::function<PlanOps<T>(MPI_Comm, int)> make = [&](MPI_Comm comm, int max_world) -> PlanOps<T> {
        if (p.sequence == "Z_Then_YX") {
            if (p.opt == 1) return opsOf<T>(std::make_shared<MPIcuFFT_Slab_Z_Then_YX_Opt1<T>>(config, comm, max_world), p, true);
            return opsOf<T>(std::make_shared<MPIcuFFT_Slab_Z_Then_YX<T>>(config, comm, max_world), p, true);
        }
        if (p.sequence == "Y_Then_ZX") {
            if (p.opt == 1) throw std::runtime_error("Y_Then_ZX has no opt 1 class.");
            return opsOf<T>(std::make_shared<MPIcuFFT_Slab_Y_Then_ZX<T>>(config, comm, max_world), p, false);
        }
        if (p.opt == 1) return opsOf<T>(std::make_shared<MPIcuFFT_Slab_Opt1<T>>(config, comm, max_world), p, true);
        return opsOf<T>(std::make_shared<MPIcuFFT_Slab<T>>(config, comm, max_world), p, true);
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
        SlabParams p;
        parseCommon(argc, argv, p);
        p.sequence = arg_value(argc, argv, "--sequence", "-s");
        if (!p.sequence.empty() && p.sequence != "ZY_Then_X" && p.sequence != "Z_Then_YX" && p.sequence != "Y_Then_ZX") throw std::runtime_error("Invalid sequence.");
        p.comm_method = comm_method_named(arg_value(argc, argv, "--comm-method", "-comm"));
        p.send_method = send_method_named(arg_value(argc, argv, "--send-method", "-snd"));
        if (getenv("DFFT_DRIVER_PARSE_ONLY")) {      // what the command line means, before MPI or the GPU are touched (tests/test_launch_commands.py)
            printf("PARSED slab nx=%zu ny=%zu nz=%zu s=%s t=%d o=%d runs=%d w=%d c=%d d=%d comm=%d snd=%d b=%s\n", p.Nx, p.Ny, p.Nz,
                   p.sequence.empty() ? "ZY_Then_X" : p.sequence.c_str(), p.testcase, p.opt, p.iterations, p.warmup_rounds, (int)p.cuda_aware,
                   (int)p.double_prec, (int)p.comm_method, (int)p.send_method, p.benchmark_dir.c_str());
            return 0;
        }
        World w(p.cuda_aware);
        return p.double_prec ? run<double>(p, w) : run<float>(p, w);
    } catch (std::runtime_error &e) {
        printf("%s\n\n", e.what());
        printf("Use \"--help\" or \"-h\" to display the help menu.\n");
        return 1;
    }
}
