// Multi-rank reference-style caller: mpiexec -n P1*P2 ./shim_mpi_multirank P1 P2 [zyx]
// ("zyx": the slab sequence Z_Then_YX, P2 = 1, output split along z)
// All ranks share GPU 0 (cudaSetDevice(rank % dev_count) in the reference,
// tests/src/pencil/random_dist_3D.cu:175-177) and exchange through host-staged MPI
// (Configurations::cuda_aware = false).  Testcase 3 (round trip) + the DC coefficient.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mpicufft_amd.hpp"

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int rank, world_size;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &world_size);
    const size_t P1 = argc > 1 ? atoi(argv[1]) : world_size, P2 = argc > 2 ? atoi(argv[2]) : 1;
    const size_t Nx = 32, Ny = 24, Nz = 40;
    int ndev = 1;
    hipGetDeviceCount(&ndev);
    hipSetDevice(rank % ndev);
    Configurations config{false, 0, All2All, Sync, "../benchmarks", All2All, Sync};
    const bool zyx = argc > 3 && std::string(argv[3]) == "zyx";
    MPIcuFFT<double> *fftp = zyx ? static_cast<MPIcuFFT<double> *>(new MPIcuFFT_Slab_Z_Then_YX<double>(config, MPI_COMM_WORLD, world_size))
                                 : static_cast<MPIcuFFT<double> *>(new MPIcuFFT_Pencil_Opt1<double>(config, MPI_COMM_WORLD, world_size));
    MPIcuFFT<double> &fft = *fftp;
    Pencil_Partition partition(P1, P2);
    GlobalSize global_size(Nx, Ny, Nz);
    fft.initFFT(&global_size, &partition, true);
    size_t isz[3], ist[3], osz[3], ost[3];
    fft.getInSize(isz); fft.getInStart(ist); fft.getOutSize(osz); fft.getOutStart(ost);
    if (zyx && (osz[0] != Nx || osz[1] != Ny || ost[0] != 0 || ost[1] != 0)) { printf("bad Z_Then_YX output block\n"); MPI_Abort(MPI_COMM_WORLD, 2); }
    const size_t n = isz[0] * isz[1] * isz[2];
    std::vector<double> in_h(n), inv_h(n);
    double sum = 0;
    for (size_t x = 0; x < isz[0]; x++) for (size_t y = 0; y < isz[1]; y++) for (size_t z = 0; z < isz[2]; z++) {
        const size_t g = ((ist[0] + x) * Ny + ist[1] + y) * Nz + z;      // value depends on the global index only
        const double v = 1.0 + std::sin(0.37 * (double)g) * 100.0;
        in_h[(x * isz[1] + y) * isz[2] + z] = v;
        sum += v;
    }
    double *in_d, *inv_d; void *out_d;
    hipMalloc((void **)&in_d, n * sizeof(double));
    hipMalloc((void **)&inv_d, n * sizeof(double));
    hipMalloc(&out_d, fft.getDomainSize());
    hipMemcpy(in_d, in_h.data(), n * sizeof(double), hipMemcpyHostToDevice);
    MPI_Barrier(MPI_COMM_WORLD);
    fft.execR2C(out_d, in_d);
    double total = 0, dc[2] = {0, 0};
    MPI_Allreduce(&sum, &total, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    if (ost[1] == 0 && ost[2] == 0) hipMemcpy(dc, out_d, sizeof(dc), hipMemcpyDeviceToHost);   // owner of X[0][0][0]
    MPI_Barrier(MPI_COMM_WORLD);
    fft.execC2R(inv_d, out_d);
    hipMemcpy(inv_h.data(), inv_d, n * sizeof(double), hipMemcpyDeviceToHost);
    double maxerr = 0, gmax = 0, norm = (double)(Nx * Ny * Nz);
    for (size_t i = 0; i < n; i++) maxerr = std::fmax(maxerr, std::fabs(inv_h[i] / norm - in_h[i]));
    MPI_Allreduce(&maxerr, &gmax, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    double dcerr = (ost[1] == 0 && ost[2] == 0) ? std::fabs(dc[0] - total) / std::fabs(total) : 0, gdc = 0;
    MPI_Allreduce(&dcerr, &gdc, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    if (rank == 0) printf("ranks %d grid %zux%zu  Result (max): %.3e  DC rel err: %.3e\n", world_size, P1, P2, gmax, gdc);
    hipFree(in_d); hipFree(inv_d); hipFree(out_d);
    delete fftp;            // frees the plan's communicators: must happen before MPI_Finalize
    int ok = gmax < 1e-10 && gdc < 1e-12;
    MPI_Finalize();
    return ok ? 0 : 1;
}
