// The reference's own call sites, compiled against include/mpicufft_amd.hpp with nothing changed but the
// include lines and the device API prefix (cuda* -> hip*):
//   pencil_testcase0 : tests/src/pencil/random_dist_3D.cu:154-227  (declares MPIcuFFT_Pencil<T>*, news the
//                      Opt1 or the opt0 class into it, getPartitionDimensions, sizes `out` from the tables)
//   slab_testcase0   : tests/src/slab/random_dist_default.cu:155-226 (declares MPIcuFFT_Slab<T>*,
//                      initFFT(&global_size, true))
// MPI_Init/Finalize live in main() so that both can run in one process; after the reference's exec loop
// each function adds the round trip of testcase 3 (random_dist_3D.cu:641-666) so that the test asserts
// something.  Usage: mpiexec -n P ./ref_caller {pencil|slab} opt P1 P2 [max_world]
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mpicufft_amd.hpp"            // was: mpicufft_pencil.hpp, mpicufft_pencil_opt1.hpp, mpicufft_slab.hpp, mpicufft_slab_opt1.hpp

#define CUDA_CALL(x) do { if ((x) != hipSuccess) { printf("Error at %s:%d\n", __FILE__, __LINE__); exit(EXIT_FAILURE); } } while (0)
// include/cufft.hpp:23-60: the element types of the template argument
template <typename T> struct cuFFT { using R_t = float; struct C_t { float x, y; }; };
template <> struct cuFFT<double> { using R_t = double; struct C_t { double x, y; }; };

template <typename T> struct Tests_Caller {      // tests/include/tests_pencil_random.hpp:25-48, tests_slab_random.hpp
    size_t Nx, Ny, Nz, P1, P2;
    Configurations config;
    std::vector<typename cuFFT<T>::R_t> host_in;
    int initializeRandArray(void *in_d, size_t N)      // tests/src/pencil/base.cu:39-58 (uniform * 255), host-side generator
    {
        host_in.resize(N);
        unsigned long long s = 88172645463325252ull + 977ull * (unsigned long long)N;
        for (size_t i = 0; i < N; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            host_in[i] = (typename cuFFT<T>::R_t)(255.0 * (double)(s >> 11) / 9007199254740992.0);
        }
        CUDA_CALL(hipMemcpy(in_d, host_in.data(), N * sizeof(host_in[0]), hipMemcpyHostToDevice));
        return 0;
    }
    double roundtrip(MPIcuFFT<T> *fft, void *out_d, void *in_d, size_t N)   // random_dist_3D.cu:641-666
    {
        using R_t = typename cuFFT<T>::R_t;
        R_t *inv_d;
        CUDA_CALL(hipMalloc((void **)&inv_d, N * sizeof(R_t)));
        MPI_Barrier(MPI_COMM_WORLD);
        fft->execR2C(out_d, in_d);
        fft->execC2R(inv_d, out_d);
        std::vector<R_t> inv(N);
        CUDA_CALL(hipMemcpy(inv.data(), inv_d, N * sizeof(R_t), hipMemcpyDeviceToHost));
        CUDA_CALL(hipFree(inv_d));
        const double norm = (double)(Nx * Ny * Nz);
        double m = 0;
        for (size_t i = 0; i < N; i++) m = std::fmax(m, std::fabs((double)inv[i] / norm - (double)host_in[i]));
        return m / 255.0;
    }
    int pencil_testcase0(const int opt, const int runs, int world_size, int rank, double *err);
    int slab_testcase0(const int opt, const int runs, int world_size, int rank, double *err);
};

template <typename T> int Tests_Caller<T>::pencil_testcase0(const int opt, const int runs, int world_size, int rank, double *err)
{
    using R_t = typename cuFFT<T>::R_t;
    using C_t = typename cuFFT<T>::C_t;

    int dev_count;
    CUDA_CALL(hipGetDeviceCount(&dev_count));
    CUDA_CALL(hipSetDevice(rank % dev_count));

    size_t pidx_i = rank / P2;
    size_t pidx_j = rank % P2;

    //initialize MPIcuFFT
    MPIcuFFT_Pencil<T> *mpicuFFT;
    if (opt == 1)
        mpicuFFT = new MPIcuFFT_Pencil_Opt1<T>(config, MPI_COMM_WORLD, world_size);
    else
        mpicuFFT = new MPIcuFFT_Pencil<T>(config, MPI_COMM_WORLD, world_size);

    Pencil_Partition partition(P1, P2);
    GlobalSize global_size(Nx, Ny, Nz);
    mpicuFFT->initFFT(&global_size, &partition, true);

    // Allocate Memory
    Partition_Dimensions input_dim, transposed_dim, output_dim;
    mpicuFFT->getPartitionDimensions(input_dim, transposed_dim, output_dim);

    size_t out_size = std::max(input_dim.size_x[pidx_i]*input_dim.size_y[pidx_j]*(Nz/2+1), transposed_dim.size_x[pidx_i]*transposed_dim.size_y[0]*transposed_dim.size_z[pidx_j]);
    out_size = std::max(out_size, output_dim.size_x[0]*output_dim.size_y[pidx_i]*output_dim.size_z[pidx_j]);

    R_t *in_d;
    C_t *out_d;

    CUDA_CALL(hipMalloc((void **)&in_d, input_dim.size_x[pidx_i]*input_dim.size_y[pidx_j]*Nz*sizeof(R_t)));
    CUDA_CALL(hipMalloc((void **)&out_d, out_size*sizeof(C_t)));

    this->initializeRandArray(in_d, input_dim.size_x[pidx_i]*input_dim.size_y[pidx_j]*Nz);
    for (int i = 0; i < runs; i++) {
        MPI_Barrier(MPI_COMM_WORLD);
        mpicuFFT->execR2C(out_d, in_d);
    }

    // (added) the tables against the getters, the caller's out_size against the library's, and the round trip
    size_t isize[3], istart[3], osize[3], ostart[3];
    mpicuFFT->getInSize(isize); mpicuFFT->getInStart(istart); mpicuFFT->getOutSize(osize); mpicuFFT->getOutStart(ostart);
    int bad = isize[0] != input_dim.size_x[pidx_i] || isize[1] != input_dim.size_y[pidx_j] || isize[2] != input_dim.size_z[0] ||
              istart[0] != input_dim.start_x[pidx_i] || istart[1] != input_dim.start_y[pidx_j] ||
              osize[0] != output_dim.size_x[0] || osize[1] != output_dim.size_y[pidx_i] || osize[2] != output_dim.size_z[pidx_j] ||
              ostart[1] != output_dim.start_y[pidx_i] || ostart[2] != output_dim.start_z[pidx_j] ||
              out_size * sizeof(C_t) > mpicuFFT->getDomainSize() || (out_size * sizeof(C_t) + 255) / 256 * 256 != mpicuFFT->getDomainSize();
    *err = roundtrip(mpicuFFT, out_d, in_d, input_dim.size_x[pidx_i]*input_dim.size_y[pidx_j]*Nz);

    CUDA_CALL(hipFree(in_d));
    CUDA_CALL(hipFree(out_d));
    delete mpicuFFT;
    return bad;
}

template <typename T> int Tests_Caller<T>::slab_testcase0(const int opt, const int runs, int world_size, int rank, double *err)
{
    using R_t = typename cuFFT<T>::R_t;
    using C_t = typename cuFFT<T>::C_t;

    int dev_count;
    CUDA_CALL(hipGetDeviceCount(&dev_count));
    CUDA_CALL(hipSetDevice(rank % dev_count));

    size_t N1=Nx/world_size;
    size_t N2=Ny/world_size;
    if (rank < Nx%world_size)
        N1++;
    if (rank < Ny%world_size)
        N2++;

    R_t *in_d;
    C_t *out_d;
    size_t out_size = std::max(N1*Ny*(Nz/2+1), Nx*N2*(Nz/2+1));

    //allocate memory (device)
    CUDA_CALL(hipMalloc((void **)&in_d, N1*Ny*Nz*sizeof(R_t)));
    CUDA_CALL(hipMalloc((void **)&out_d, out_size*sizeof(C_t)));

    MPIcuFFT_Slab<T> *mpicuFFT;
    if (opt == 1)
        mpicuFFT = new MPIcuFFT_Slab_Opt1<T>(config, MPI_COMM_WORLD, world_size);
    else
        mpicuFFT = new MPIcuFFT_Slab<T>(config, MPI_COMM_WORLD, world_size);

    GlobalSize global_size(Nx, Ny, Nz);
    mpicuFFT->initFFT(&global_size, true);

    //execute
    this->initializeRandArray(in_d, N1*Ny*Nz);
    for (int i = 0; i < runs; i++){
        MPI_Barrier(MPI_COMM_WORLD);
        mpicuFFT->execR2C(out_d, in_d);
    }

    // (added) the caller's sizes against the library's, and the round trip
    size_t isize[3], osize[3];
    mpicuFFT->getInSize(isize); mpicuFFT->getOutSize(osize);
    int bad = isize[0] != N1 || isize[1] != Ny || isize[2] != Nz || osize[0] != Nx || osize[1] != N2 || osize[2] != Nz/2+1 ||
              out_size * sizeof(C_t) > mpicuFFT->getDomainSize();
    *err = roundtrip(mpicuFFT, out_d, in_d, N1*Ny*Nz);

    CUDA_CALL(hipFree(in_d));
    CUDA_CALL(hipFree(out_d));
    delete mpicuFFT;
    return bad;
}

template <typename T> static int run(const std::string &kind, int opt, size_t P1, size_t P2, int world_size, int rank, int fft_ranks)
{
    Tests_Caller<T> t;
    t.Nx = 36; t.Ny = 20; t.Nz = 24; t.P1 = P1; t.P2 = P2;                 // uneven splits on 3 x 2 and on 5 ranks
    t.config = Configurations{false, 0, All2All, Sync, "../benchmarks", All2All, Sync};
    double err = 0, gerr = 0;
    int bad = 0, gbad = 0;
    if (rank < fft_ranks) {
        bad = kind == "pencil" ? t.pencil_testcase0(opt, 2, fft_ranks, rank, &err) : t.slab_testcase0(opt, 2, fft_ranks, rank, &err);
    } else {
        // a rank outside the FFT world: the reference's coordinator matches the constructor's split by hand
        // (tests/src/pencil/random_dist_3D.cu:314-315) and meets the others at their barriers
        MPI_Comm temp;
        MPI_Comm_split(MPI_COMM_WORLD, MPI_UNDEFINED, 0, &temp);
        for (int i = 0; i < 3; i++) MPI_Barrier(MPI_COMM_WORLD);          // 2 runs + the round trip
    }
    MPI_Allreduce(&err, &gerr, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    MPI_Allreduce(&bad, &gbad, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
    if (rank == 0) printf("%s opt %d ranks %d (of %d) grid %zux%zu  tables %s  Result (max): %.3e\n", kind.c_str(), opt, fft_ranks, world_size, P1, P2,
                          gbad ? "MISMATCH" : "ok", gerr);
    return gbad == 0 && gerr < (sizeof(T) == 8 ? 1e-12 : 1e-5) ? 0 : 1;
}

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int world_size, rank;
    MPI_Comm_size(MPI_COMM_WORLD, &world_size);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    const std::string kind = argc > 1 ? argv[1] : "pencil";
    const int opt = argc > 2 ? atoi(argv[2]) : 1;
    const size_t P1 = argc > 3 ? atoi(argv[3]) : world_size, P2 = argc > 4 ? atoi(argv[4]) : 1;
    const int fft_ranks = argc > 5 ? atoi(argv[5]) : world_size;      // < world_size: max_world_size truncation
    int rc = run<double>(kind, opt, P1, P2, world_size, rank, fft_ranks);
    rc |= run<float>(kind, opt, P1, P2, world_size, rank, fft_ranks);
    MPI_Finalize();
    return rc;
}
