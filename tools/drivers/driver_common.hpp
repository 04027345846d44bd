// driver_common.hpp -- what the `pencil` and `slab` executables share: the reference's command-line conventions
// (tests/src/pencil/main.cpp:68-192, tests/src/slab/main.cpp:63-180) and its five testcases
// (tests/src/pencil/random_dist_3D.cu:154-811, tests/src/slab/random_dist_default.cu) on top of include/mpicufft_amd.hpp.
//
// The reference's test utilities are cuRAND (input), cuBLAS (asum / amax) and small CUDA kernels (difference,
// derivativeCoefficients); here the same quantities are computed on the host after a copy -- they are test helpers, the
// transforms under test run through the shim into libdfft_amd.so.  `launch.py`-style job lines
// (`mpiexec -n P ./pencil -nx .. -p1 .. -t 0 -o 1 -i 20 -w 10 -d -c -b dir`) run unchanged and leave the timer CSVs the
// reference's eval scripts read (include/timer_amd.hpp).
#pragma once
#include <mpi.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <functional>
#include <iostream>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "mpicufft_amd.hpp"

#define HIP_CALL(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("Error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); MPI_Abort(MPI_COMM_WORLD, 1); } } while (0)

// `out` (and the inverse's output) come from the library's allocator: same ownership as the reference's cudaMalloc'd buffers
// (tests/src/pencil/random_dist_3D.cu:197-205: the caller allocates and frees), on the backing the scatter passes run ~7 %
// faster on (include/dfft_c.h: dfft_malloc(DFFT_CHUNK_DEFAULT); DFFT_DEFAULT_CHUNK_MIB=0 makes it hipMalloc again)
#define DFFT_CALL(x) do { int r_ = (x); if (r_ != 0) { printf("Error %d (%s) at %s:%d\n", r_, dfft_last_error(), __FILE__, __LINE__); MPI_Abort(MPI_COMM_WORLD, 1); } } while (0)

namespace driver {

inline std::string arg_value(int argc, char *argv[], const std::string &long_name, const std::string &short_name)
{
    for (int i = 0; i + 1 < argc; i++)
        if (long_name == argv[i] || short_name == argv[i]) return argv[i + 1];
    return "";
}
inline bool arg_present(int argc, char *argv[], const std::string &long_name, const std::string &short_name)
{
    for (int i = 0; i < argc; i++)
        if (long_name == argv[i] || short_name == argv[i]) return true;
    return false;
}
inline size_t as_size(const std::string &s, bool req = false, const std::string &error = "")
{
    if (s.empty()) { if (req) throw std::runtime_error(error); return 0; }
    std::stringstream ss(s);
    size_t v = 0;
    ss >> v;
    return v;
}
inline int as_int(const std::string &s) { return (int)as_size(s); }
inline CommunicationMethod comm_method_named(const std::string &s)
{
    if (s == "Peer2Peer" || s.empty()) return Peer2Peer;
    if (s == "All2All") return All2All;
    throw std::runtime_error("Invalid communication method.");
}
inline SendMethod send_method_named(const std::string &s)
{
    if (s == "Sync" || s.empty()) return Sync;
    if (s == "Streams") return Streams;
    if (s == "MPI_Type") return MPI_Type;
    throw std::runtime_error("Invalid send method.");
}

struct Common {
    size_t Nx = 0, Ny = 0, Nz = 0;
    int testcase = 0, opt = 0, iterations = 1, warmup_rounds = 0, fft_dim = 3;
    bool cuda_aware = false, double_prec = false;
    std::string benchmark_dir;
};
inline void parseCommon(int argc, char *argv[], Common &p)
{
    p.Nx = as_size(arg_value(argc, argv, "--input-dim-x", "-nx"), true, "Input parameter Nx is required.");
    p.Ny = as_size(arg_value(argc, argv, "--input-dim-y", "-ny"), true, "Input parameter Ny is required.");
    p.Nz = as_size(arg_value(argc, argv, "--input-dim-z", "-nz"), true, "Input parameter Nz is required.");
    p.iterations = as_int(arg_value(argc, argv, "--iterations", "-i"));
    p.warmup_rounds = as_int(arg_value(argc, argv, "--warmup-rounds", "-w"));
    if (p.iterations == 0 && p.warmup_rounds == 0) p.iterations = 1;
    p.iterations += p.warmup_rounds;
    p.cuda_aware = arg_present(argc, argv, "--cuda_aware", "-c");
    p.double_prec = arg_present(argc, argv, "--double_prec", "-d");
    p.benchmark_dir = arg_value(argc, argv, "--benchmark_dir", "-b");
    if (p.benchmark_dir.empty()) p.benchmark_dir = "../benchmarks";
    p.testcase = as_int(arg_value(argc, argv, "--testcase", "-t"));
    if (p.testcase < 0 || p.testcase > 4) throw std::runtime_error("Invalid testcase.");
    p.opt = as_int(arg_value(argc, argv, "--opt", "-o"));
    if (p.opt < 0 || p.opt > 1) throw std::runtime_error("Invalid option.");
}

// MPI_Init_thread(MPI_THREAD_MULTIPLE) like every testcase of the reference, one GPU per rank modulo the device count
// (tests/src/pencil/random_dist_3D.cu:160-177).  The device path (RCCL) needs a GPU of its own per rank: on a box with fewer
// devices than ranks the flag is dropped, as the reference drops it when MPIX_Query_cuda_support() says no
// (tests/src/pencil/main.cpp:201).
struct World {
    int rank = 0, size = 1, dev_count = 1;
    explicit World(bool &cuda_aware)
    {
        int provided = 0;
        MPI_Init_thread(nullptr, nullptr, MPI_THREAD_MULTIPLE, &provided);
        MPI_Comm_size(MPI_COMM_WORLD, &size);
        MPI_Comm_rank(MPI_COMM_WORLD, &rank);
        HIP_CALL(hipGetDeviceCount(&dev_count));
        HIP_CALL(hipSetDevice(rank % dev_count));
        if (cuda_aware && dev_count < size) {
            if (rank == 0) printf("note: %d ranks share %d device(s): host-staged exchange instead of --cuda_aware\n", size, dev_count);
            cuda_aware = false;
        }
    }
    ~World() { MPI_Finalize(); }
};

// what a testcase needs of a plan object, whichever class it is
template <typename T> struct PlanOps {
    std::function<void(void *, const void *)> forward, inverse;
    size_t isz[3], ist[3], osz[3], ost[3];
    size_t in_elems() const { return isz[0] * isz[1] * isz[2]; }
    size_t out_elems() const { return osz[0] * osz[1] * osz[2]; }
    size_t domain_bytes = 0;       // >= every stage of the transform (out buffers are sized with it)
    bool has_inverse = true;
};
template <typename T> void fillSizes(MPIcuFFT<T> *p, PlanOps<T> &ops)
{
    p->getInSize(ops.isz); p->getInStart(ops.ist); p->getOutSize(ops.osz); p->getOutStart(ops.ost);
    ops.domain_bytes = p->getDomainSize();
}

// uniform [0, 255) like the reference's scaled cuRAND input (tests/src/pencil/base.cu:45-53)
template <typename T> void randomFill(T *dev, size_t n, unsigned long long seed)
{
    std::vector<T> h(n);
    std::mt19937_64 gen(seed);
    std::uniform_real_distribution<double> u(0.0, 255.0);
    for (auto &v : h) v = (T)u(gen);
    HIP_CALL(hipMemcpy(dev, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
}
// sum and maximum of |a - b * scale| over n values (differenceInv + cublas asum / amax of the reference)
template <typename T> void differenceNorms(const T *dev_a, double scale_a, const std::vector<T> &b, double *sum, double *mx)
{
    std::vector<T> a(b.size());
    HIP_CALL(hipMemcpy(a.data(), dev_a, a.size() * sizeof(T), hipMemcpyDeviceToHost));
    double s = 0, m = 0;
    for (size_t i = 0; i < a.size(); i++) {
        const double d = std::fabs((double)a[i] * scale_a - (double)b[i]);
        s += d;
        m = std::max(m, d);
    }
    *sum = s; *mx = m;
}
inline void printResult(int rank, double sum, double mx, double n)
{
    double gs = 0, gm = 0;
    MPI_Allreduce(&sum, &gs, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    MPI_Allreduce(&mx, &gm, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    if (rank == 0) {
        std::cout << "Result (avg): " << gs / n << std::endl;
        std::cout << "Result (max): " << gm << std::endl;
    }
}

// testcase 0 / 2: random input, forward / inverse transform, timer CSV only
template <typename T> int runTimed(const PlanOps<T> &ops, const Common &c, int rank, bool inverse)
{
    if (inverse && !ops.has_inverse) throw std::runtime_error("this sequence has no inverse transform");
    T *in_d = nullptr;
    void *out_d = nullptr;
    HIP_CALL(hipMalloc(&in_d, ops.in_elems() * sizeof(T)));
    DFFT_CALL(dfft_malloc(ops.domain_bytes, DFFT_CHUNK_DEFAULT, &out_d));
    for (int i = 0; i < c.iterations; i++) {
        if (!inverse) {
            randomFill(in_d, ops.in_elems(), 1000003ull * (rank + 1) + i);
        } else {
            randomFill(reinterpret_cast<T *>(out_d), 2 * ops.out_elems(), 1000003ull * (rank + 1) + i);
        }
        MPI_Barrier(MPI_COMM_WORLD);
        if (!inverse) ops.forward(out_d, in_d); else ops.inverse(in_d, out_d);
    }
    HIP_CALL(hipFree(in_d)); DFFT_CALL(dfft_free(out_d));
    return 0;
}

// testcase 3: forward, inverse, compare with the input (tests/src/pencil/random_dist_3D.cu:581-683)
template <typename T> int runRoundTrip(const PlanOps<T> &ops, const Common &c, int rank)
{
    if (!ops.has_inverse) throw std::runtime_error("this sequence has no inverse transform");
    const size_t n = ops.in_elems();
    T *in_d = nullptr, *inv_d = nullptr;
    void *out_d = nullptr;
    HIP_CALL(hipMalloc(&in_d, n * sizeof(T)));
    DFFT_CALL(dfft_malloc(n * sizeof(T), DFFT_CHUNK_DEFAULT, (void **)&inv_d));
    DFFT_CALL(dfft_malloc(ops.domain_bytes, DFFT_CHUNK_DEFAULT, &out_d));
    // an unnormalised forward + inverse pair multiplies by the transformed extents: Nz, Ny*Nz or Nx*Ny*Nz (--fft-dim 1, 2, 3)
    const double N3 = c.fft_dim == 1 ? (double)c.Nz : c.fft_dim == 2 ? (double)c.Ny * c.Nz : (double)c.Nx * c.Ny * c.Nz;
    std::vector<T> h(n);
    for (int i = 0; i < c.iterations; i++) {
        randomFill(in_d, n, 7919ull * (rank + 1) + i);
        HIP_CALL(hipMemcpy(h.data(), in_d, n * sizeof(T), hipMemcpyDeviceToHost));
        MPI_Barrier(MPI_COMM_WORLD);
        ops.forward(out_d, in_d);
        MPI_Barrier(MPI_COMM_WORLD);
        ops.inverse(inv_d, out_d);
        MPI_Barrier(MPI_COMM_WORLD);
        double sum, mx;
        differenceNorms(inv_d, 1.0 / N3, h, &sum, &mx);      // the reference scales the input by N^3 instead: same quantity / N^3
        printResult(rank, sum * N3, mx * N3, (double)c.Nx * c.Ny * c.Nz);
        MPI_Barrier(MPI_COMM_WORLD);
    }
    HIP_CALL(hipFree(in_d)); DFFT_CALL(dfft_free(inv_d)); DFFT_CALL(dfft_free(out_d));
    return 0;
}

// testcase 4: spectral Laplacian of sin(2 pi x/Nx) sin(2 pi y/Ny) sin(2 pi z/Nz) against the analytic result
// (tests/src/pencil/random_dist_3D.cu:685-811; derivativeCoefficients :98-121).  `hermitian_axis` = 2 (z) for every class
// but Y_Then_ZX (which has no inverse anyway).
template <typename T> int runLaplacian(const PlanOps<T> &ops, const Common &c, int rank)
{
    if (!ops.has_inverse) throw std::runtime_error("this sequence has no inverse transform");
    const size_t n = ops.in_elems(), no = ops.out_elems();
    T *in_d = nullptr, *inv_d = nullptr;
    void *out_d = nullptr;
    HIP_CALL(hipMalloc(&in_d, n * sizeof(T)));
    DFFT_CALL(dfft_malloc(n * sizeof(T), DFFT_CHUNK_DEFAULT, (void **)&inv_d));
    DFFT_CALL(dfft_malloc(ops.domain_bytes, DFFT_CHUNK_DEFAULT, &out_d));
    const double Nx = (double)c.Nx, Ny = (double)c.Ny, Nz = (double)c.Nz, root = std::sqrt(Nx * Ny * Nz);
    // the divisor of the reference's derivativeCoefficients is sqrtf -- SINGLE precision -- of the int product (:96, :117): the analytic
    // answer below uses the double root (:758-762), and the difference of the two is what the reference's own runs print
    // (1.91723e-05 / 7.43e-05 at 128^3: benchmarks/argon/pencil.6067.out; tests/golden/ref_testcase4_results.json)
    const double rootf = (double)std::sqrt((float)(int)(c.Nx * c.Ny * c.Nz));
    std::vector<T> in_h(n), der_h(n);
    for (size_t x = 0; x < ops.isz[0]; x++)
        for (size_t y = 0; y < ops.isz[1]; y++)
            for (size_t z = 0; z < ops.isz[2]; z++) {
                const double v = std::sin(2.0 * M_PI * (ops.ist[0] + x) / Nx) * std::sin(2.0 * M_PI * (ops.ist[1] + y) / Ny) *
                                 std::sin(2.0 * M_PI * (ops.ist[2] + z) / Nz);
                in_h[(x * ops.isz[1] + y) * ops.isz[2] + z] = (T)v;
                der_h[(x * ops.isz[1] + y) * ops.isz[2] + z] = (T)(-3.0 * root * v);
            }
    HIP_CALL(hipMemcpy(in_d, in_h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    std::vector<std::complex<T>> spec(no);
    for (int i = 0; i < c.iterations; i++) {
        ops.forward(out_d, in_d);
        HIP_CALL(hipMemcpy(spec.data(), out_d, no * sizeof(std::complex<T>), hipMemcpyDeviceToHost));
        for (size_t x = 0; x < ops.osz[0]; x++)
            for (size_t y = 0; y < ops.osz[1]; y++)
                for (size_t z = 0; z < ops.osz[2]; z++) {
                    const long gx = (long)(ops.ost[0] + x), gy = (long)(ops.ost[1] + y), gz = (long)(ops.ost[2] + z);
                    long k1 = 0, k2 = 0, k3 = 0;
                    if (gx < (long)c.Nx / 2) k1 = gx; else if (gx > (long)(c.Nx / 2)) k1 = (long)c.Nx - gx;
                    if (gy < (long)c.Ny / 2) k2 = gy; else if (gy > (long)(c.Ny / 2)) k2 = (long)c.Ny - gy;
                    if (gz < (long)c.Nz / 2) k3 = gz;
                    const double scale = -(double)(k1 * k1 + k2 * k2 + k3 * k3);
                    std::complex<T> &v = spec[(x * ops.osz[1] + y) * ops.osz[2] + z];
                    v = std::complex<T>((T)((double)v.real() * scale / rootf), (T)((double)v.imag() * scale / rootf));
                }
        HIP_CALL(hipMemcpy(out_d, spec.data(), no * sizeof(std::complex<T>), hipMemcpyHostToDevice));
        MPI_Barrier(MPI_COMM_WORLD);
        ops.inverse(inv_d, out_d);
        double sum, mx;
        differenceNorms(inv_d, 1.0, der_h, &sum, &mx);
        printResult(rank, sum, mx, Nx * Ny * Nz);
        MPI_Barrier(MPI_COMM_WORLD);
    }
    HIP_CALL(hipFree(in_d)); DFFT_CALL(dfft_free(inv_d)); DFFT_CALL(dfft_free(out_d));
    return 0;
}

// testcase 1: the last rank generates the global input, hands every worker its block, transforms the whole grid on its own
// GPU with a single-rank plan of the same class and compares with the blocks the workers send back
// (tests/src/pencil/random_dist_3D.cu:229-504).  make(comm, max_world_size) builds the plan object of the class under test.
template <typename T>
int runCoordinated(const std::function<PlanOps<T>(MPI_Comm, int)> &make, const Common &c, int rank, int world_size)
{
    const int workers = world_size - 1;
    if (workers < 1) throw std::runtime_error("testcase 1 needs one more rank than the partition has (the coordinator)");
    using Cx = std::complex<T>;
    if (rank == workers) {
        MPI_Comm temp;
        MPI_Comm_split(MPI_COMM_WORLD, MPI_UNDEFINED, 0, &temp);      // matches the workers' MPI_Comm_split inside the constructor
        std::vector<size_t> dims((size_t)workers * 12);
        for (int p = 0; p < workers; p++) MPI_Recv(&dims[(size_t)p * 12], 12 * sizeof(size_t), MPI_BYTE, p, 0, MPI_COMM_WORLD, MPI_STATUS_IGNORE);
        PlanOps<T> full = make(MPI_COMM_SELF, -1);
        const size_t n = c.Nx * c.Ny * c.Nz, no = full.out_elems();
        T *in_d = nullptr;
        void *out_d = nullptr;
        HIP_CALL(hipMalloc(&in_d, n * sizeof(T)));
        DFFT_CALL(dfft_malloc(full.domain_bytes, DFFT_CHUNK_DEFAULT, &out_d));
        std::vector<T> in_h(n), blk;
        std::vector<Cx> want(no), got;
        for (int i = 0; i < c.iterations; i++) {
            randomFill(in_d, n, 104729ull + i);
            HIP_CALL(hipMemcpy(in_h.data(), in_d, n * sizeof(T), hipMemcpyDeviceToHost));
            for (int p = 0; p < workers; p++) {
                const size_t *isz = &dims[(size_t)p * 12], *ist = isz + 3;
                blk.resize(isz[0] * isz[1] * isz[2]);
                for (size_t x = 0; x < isz[0]; x++)
                    for (size_t y = 0; y < isz[1]; y++)
                        std::copy_n(&in_h[((ist[0] + x) * c.Ny + ist[1] + y) * c.Nz + ist[2]], isz[2], &blk[(x * isz[1] + y) * isz[2]]);
                MPI_Send(blk.data(), (int)(blk.size() * sizeof(T)), MPI_BYTE, p, 1, MPI_COMM_WORLD);
            }
            MPI_Barrier(MPI_COMM_WORLD);
            full.forward(out_d, in_d);
            HIP_CALL(hipMemcpy(want.data(), out_d, no * sizeof(Cx), hipMemcpyDeviceToHost));
            double sum = 0;
            for (int p = 0; p < workers; p++) {
                const size_t *osz = &dims[(size_t)p * 12 + 6], *ost = osz + 3;
                got.resize(osz[0] * osz[1] * osz[2]);
                MPI_Recv(got.data(), (int)(got.size() * sizeof(Cx)), MPI_BYTE, p, 2, MPI_COMM_WORLD, MPI_STATUS_IGNORE);
                for (size_t x = 0; x < osz[0]; x++)
                    for (size_t y = 0; y < osz[1]; y++)
                        for (size_t z = 0; z < osz[2]; z++) {
                            const Cx d = got[(x * osz[1] + y) * osz[2] + z] -
                                         want[((ost[0] + x) * full.osz[1] + ost[1] + y) * full.osz[2] + ost[2] + z];
                            sum += std::fabs((double)d.real()) + std::fabs((double)d.imag());
                        }
            }
            printf("\nResults: %f\n", sum);
            MPI_Barrier(MPI_COMM_WORLD);
        }
        HIP_CALL(hipFree(in_d)); DFFT_CALL(dfft_free(out_d));
        return 0;
    }
    PlanOps<T> ops = make(MPI_COMM_WORLD, workers);
    size_t dims[12];
    for (int k = 0; k < 3; k++) { dims[k] = ops.isz[k]; dims[3 + k] = ops.ist[k]; dims[6 + k] = ops.osz[k]; dims[9 + k] = ops.ost[k]; }
    MPI_Send(dims, sizeof(dims), MPI_BYTE, workers, 0, MPI_COMM_WORLD);
    T *in_d = nullptr;
    void *out_d = nullptr;
    HIP_CALL(hipMalloc(&in_d, ops.in_elems() * sizeof(T)));
    DFFT_CALL(dfft_malloc(ops.domain_bytes, DFFT_CHUNK_DEFAULT, &out_d));
    std::vector<T> in_h(ops.in_elems());
    std::vector<Cx> out_h(ops.out_elems());
    for (int i = 0; i < c.iterations; i++) {
        MPI_Recv(in_h.data(), (int)(in_h.size() * sizeof(T)), MPI_BYTE, workers, 1, MPI_COMM_WORLD, MPI_STATUS_IGNORE);
        HIP_CALL(hipMemcpy(in_d, in_h.data(), in_h.size() * sizeof(T), hipMemcpyHostToDevice));
        MPI_Barrier(MPI_COMM_WORLD);
        ops.forward(out_d, in_d);
        HIP_CALL(hipMemcpy(out_h.data(), out_d, out_h.size() * sizeof(Cx), hipMemcpyDeviceToHost));
        MPI_Send(out_h.data(), (int)(out_h.size() * sizeof(Cx)), MPI_BYTE, workers, 2, MPI_COMM_WORLD);
        MPI_Barrier(MPI_COMM_WORLD);
    }
    HIP_CALL(hipFree(in_d)); DFFT_CALL(dfft_free(out_d));
    return 0;
}

template <typename T>
int runTestcase(const std::function<PlanOps<T>(MPI_Comm, int)> &make, const Common &c, const World &w)
{
    if (c.testcase == 1) return runCoordinated<T>(make, c, w.rank, w.size);
    PlanOps<T> ops = make(MPI_COMM_WORLD, w.size);
    switch (c.testcase) {
    case 0: return runTimed<T>(ops, c, w.rank, false);
    case 2: return runTimed<T>(ops, c, w.rank, true);
    case 3: return runRoundTrip<T>(ops, c, w.rank);
    default: return runLaplacian<T>(ops, c, w.rank);
    }
}

}  // namespace driver
