// A reference-style C++ caller (tests/src/pencil/random_dist_3D.cu:581-683, testcase 3) written
// against include/mpicufft_amd.hpp: same class names and calls as the reference's test, HIP
// runtime instead of CUDA.  Single MPI rank; prints "Result (max): <err>" like the reference.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "mpicufft_amd.hpp"

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int rank, world_size;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &world_size);
    const size_t Nx = 64, Ny = 32, Nz = 48;     // 48: not a power of two (Bluestein z pass)
    hipSetDevice(0);
    Configurations config{true, 0, All2All, Sync, "../benchmarks", All2All, Sync};
    MPIcuFFT<double> *mpicuFFT = new MPIcuFFT_Pencil_Opt1<double>(config, MPI_COMM_WORLD, world_size);
    Pencil_Partition partition(1, 1);
    GlobalSize global_size(Nx, Ny, Nz);
    mpicuFFT->initFFT(&global_size, &partition, true);
    size_t isize[3], osize[3];
    mpicuFFT->getInSize(isize);
    mpicuFFT->getOutSize(osize);
    const size_t n = isize[0] * isize[1] * isize[2];
    std::vector<double> in_h(n), inv_h(n);
    unsigned long long s = 88172645463325252ull;
    double sum = 0;
    for (size_t i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        in_h[i] = 255.0 * (double)(s >> 11) / 9007199254740992.0;
        sum += in_h[i];
    }
    double *in_d, *inv_d;
    void *out_d;
    hipMalloc((void **)&in_d, n * sizeof(double));
    hipMalloc((void **)&inv_d, n * sizeof(double));
    hipMalloc(&out_d, mpicuFFT->getDomainSize());
    hipMemcpy(in_d, in_h.data(), n * sizeof(double), hipMemcpyHostToDevice);
    mpicuFFT->execR2C(out_d, in_d);
    double dc[2];
    hipMemcpy(dc, out_d, sizeof(dc), hipMemcpyDeviceToHost);
    mpicuFFT->execC2R(inv_d, out_d);
    hipMemcpy(inv_h.data(), inv_d, n * sizeof(double), hipMemcpyDeviceToHost);
    double maxerr = 0, norm = (double)(Nx * Ny * Nz);
    for (size_t i = 0; i < n; i++) maxerr = std::fmax(maxerr, std::fabs(inv_h[i] - norm * in_h[i]));   // differenceInv, :650
    printf("out size %zu %zu %zu\n", osize[0], osize[1], osize[2]);
    printf("DC rel err: %.3e\n", std::fabs(dc[0] - sum) / sum);
    printf("Result (max): %.6e\n", maxerr / norm);
    delete mpicuFFT;
    hipFree(in_d); hipFree(inv_d); hipFree(out_d);
    MPI_Finalize();
    return (maxerr / norm < 1e-9 && std::fabs(dc[0] - sum) / sum < 1e-12) ? 0 : 1;
}
