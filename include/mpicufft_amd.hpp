// mpicufft_amd.hpp -- header-only C++ shim that gives the reference's call sites
// (tests/src/pencil/random_dist_3D.cu:183-213, tests/src/slab/random_dist_default.cu) the class
// names and member functions they use, implemented on libdfft_amd.so's C ABI (dfft_c.h).
//
// Mirrors:  struct GlobalSize/Partition/.../Configurations   include/params.hpp:24-93
//           template<typename T> class MPIcuFFT               include/mpicufft.hpp:55-105
//           MPIcuFFT_Slab, _Slab_Opt1, _Pencil, _Pencil_Opt1  include/mpicufft_{slab,pencil}*.hpp
// MPI is used only as the reference uses it at this boundary: rank/size and (here) one
// MPI_Bcast of the RCCL unique id.  Errors throw std::runtime_error on every rank (the
// reference prints and exit()s, src/pencil/mpicufft_pencil_opt1.cpp:27-33).
#pragma once
#include <mpi.h>

#include <climits>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "dfft_c.h"

struct GlobalSize {
    GlobalSize(size_t Nx_, size_t Ny_, size_t Nz_) : Nx(Nx_), Ny(Ny_), Nz(Nz_), Nz_out(Nz_ / 2 + 1) {}
    size_t Nx, Ny, Nz, Nz_out;
};
struct Partition { size_t P1, P2; };
struct Slab_Partition : public Partition { explicit Slab_Partition(size_t P1_) { P1 = P1_; P2 = 1; } };
struct Pencil_Partition : public Partition { Pencil_Partition(size_t P1_, size_t P2_) { P1 = P1_; P2 = P2_; } };
enum CommunicationMethod { Peer2Peer, All2All };
enum SendMethod { Sync, Streams, MPI_Type };
struct Configurations {
    bool cuda_aware;
    int warmup_rounds;
    CommunicationMethod comm_method;
    SendMethod send_method;
    std::string benchmark_dir;
    CommunicationMethod comm_method2;
    SendMethod send_method2;
};

// Host-staged exchange = the reference's cuda_aware == false path (device -> pinned host ->
// MPI_Alltoallv -> device, src/pencil/mpicufft_pencil_opt1.cpp:777-779, 834-836).  Works with any
// MPI and with several ranks sharing one GPU (where RCCL cannot be used).  Plugged into the library
// through its callback transport; sub-communicators for the row/column groups are split lazily.
class DfftHostStagedMPI {
public:
    explicit DfftHostStagedMPI(MPI_Comm comm) : comm_(comm) {}
    ~DfftHostStagedMPI()
    {
        for (auto &kv : groups_) MPI_Comm_free(&kv.second);
        if (hsend_) hipHostFree(hsend_);
        if (hrecv_) hipHostFree(hrecv_);
    }
    static int alltoallv(void *user, const void *sendbuf, const size_t *scounts, const size_t *sdispls, void *recvbuf,
                         const size_t *rcounts, const size_t *rdispls, const int *group, int ngroup, int me, void *stream)
    {
        return static_cast<DfftHostStagedMPI *>(user)->run(sendbuf, scounts, sdispls, recvbuf, rcounts, rdispls, group, ngroup, me,
                                                            static_cast<hipStream_t>(stream));
    }

private:
    int run(const void *sendbuf, const size_t *sc, const size_t *sd, void *recvbuf, const size_t *rc, const size_t *rd,
            const int *group, int ng, int me, hipStream_t stream)
    {
        auto key = std::make_tuple(group[0], ng > 1 ? group[1] - group[0] : 0, ng);
        auto it = groups_.find(key);
        if (it == groups_.end()) {
            MPI_Comm sub;
            // collective over the parent: every rank is in exactly one group of this exchange
            MPI_Comm_split(comm_, group[0], me, &sub);
            it = groups_.emplace(key, sub).first;
        }
        size_t stot = 0, rtot = 0;
        std::vector<int> isc(ng), isd(ng), irc(ng), ird(ng);
        for (int q = 0; q < ng; q++) {
            if (sc[q] > (size_t)INT_MAX || rc[q] > (size_t)INT_MAX || stot > (size_t)INT_MAX || rtot > (size_t)INT_MAX) return 2;
            isc[q] = (int)sc[q]; isd[q] = (int)stot; stot += sc[q];
            irc[q] = (int)rc[q]; ird[q] = (int)rtot; rtot += rc[q];
        }
        if (stot > hcap_s_) { if (hsend_) hipHostFree(hsend_); if (hipHostMalloc(&hsend_, stot) != hipSuccess) return 3; hcap_s_ = stot; }
        if (rtot > hcap_r_) { if (hrecv_) hipHostFree(hrecv_); if (hipHostMalloc(&hrecv_, rtot) != hipSuccess) return 3; hcap_r_ = rtot; }
        for (int q = 0; q < ng; q++)
            if (sc[q] && hipMemcpyAsync(static_cast<char *>(hsend_) + isd[q], static_cast<const char *>(sendbuf) + sd[q], sc[q],
                                        hipMemcpyDeviceToHost, stream) != hipSuccess) return 4;
        if (hipStreamSynchronize(stream) != hipSuccess) return 4;
        if (MPI_Alltoallv(hsend_, isc.data(), isd.data(), MPI_BYTE, hrecv_, irc.data(), ird.data(), MPI_BYTE, it->second) != MPI_SUCCESS)
            return 5;
        for (int q = 0; q < ng; q++)
            if (rc[q] && hipMemcpyAsync(static_cast<char *>(recvbuf) + rd[q], static_cast<const char *>(hrecv_) + ird[q], rc[q],
                                        hipMemcpyHostToDevice, stream) != hipSuccess) return 4;
        // peers may overwrite nothing of mine (everything went through host copies), but my own next
        // exchange reuses the staging buffers: drain before returning
        if (hipStreamSynchronize(stream) != hipSuccess) return 4;
        return 0;
    }
    MPI_Comm comm_;
    std::map<std::tuple<int, int, int>, MPI_Comm> groups_;
    void *hsend_ = nullptr, *hrecv_ = nullptr;
    size_t hcap_s_ = 0, hcap_r_ = 0;
};

template <typename T> class MPIcuFFT {
public:
    MPIcuFFT(Configurations config, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1, int kind = DFFT_PENCIL_OPT1)
    {
        MPI_Comm_size(comm, &pcnt);
        MPI_Comm_rank(comm, &pidx);
        if (max_world_size > 0 && max_world_size < pcnt) pcnt = max_world_size;   // src/mpicufft.cpp:46-51
        if (pidx >= pcnt) return;                                                 // not part of the FFT world
        if (pcnt > 1) {
            MPI_Comm_split(comm, 0, pidx, &sub_);      // the first pcnt ranks (src/mpicufft.cpp:46-51)
            if (config.cuda_aware) {                   // device path: RCCL over xGMI, one rank per GPU
                char id[128];
                if (pidx == 0) check(dfft_rccl_unique_id(id));
                MPI_Bcast(id, 128, MPI_BYTE, 0, sub_);
                check(dfft_comm_create_rccl(id, pcnt, pidx, &comm_));
            } else {                                   // host-staged MPI, ranks may share a GPU
                staged_ = new DfftHostStagedMPI(sub_);
                check(dfft_comm_create_callback(pcnt, pidx, &DfftHostStagedMPI::alltoallv, staged_, &comm_));
            }
        }
        dfft_config c{config.cuda_aware, config.warmup_rounds, (int)config.comm_method, (int)config.send_method,
                      (int)config.comm_method2, (int)config.send_method2};
        check(dfft_plan_create(&plan_, kind, sizeof(T) == 8 ? DFFT_F64 : DFFT_F32, &c, comm_, pidx, max_world_size));
    }
    virtual ~MPIcuFFT()
    {
        dfft_plan_destroy(plan_);
        if (comm_) dfft_comm_destroy(comm_);
        delete staged_;
        if (sub_ != MPI_COMM_NULL) MPI_Comm_free(&sub_);
    }
    virtual void initFFT(GlobalSize *global_size, Partition *partition, bool allocate = true)
    {
        if (!global_size || !partition) throw std::runtime_error("GlobalSize or Partition not initialized!");
        check(dfft_init(plan_, global_size->Nx, global_size->Ny, global_size->Nz, (int)partition->P1,
                        (int)partition->P2, /*c2c=*/0, allocate));
    }
    // complex-to-complex plan (extension): same layouts with Nz_out = Nz
    void initFFT_C2C(GlobalSize *g, Partition *p, bool allocate = true)
    {
        check(dfft_init(plan_, g->Nx, g->Ny, g->Nz, (int)p->P1, (int)p->P2, 1, allocate));
    }
    virtual void setWorkArea(void *device = nullptr, void *host = nullptr) { check(dfft_set_work_area(plan_, device, host)); }
    virtual void execR2C(void *out, const void *in) { check(dfft_exec_r2c(plan_, out, in)); }
    virtual void execC2R(void *out, const void *in) { check(dfft_exec_c2r(plan_, out, const_cast<void *>(in))); }
    void execC2C(void *out, void *in, int direction) { check(dfft_exec_c2c(plan_, out, in, direction)); }
    inline void getInSize(size_t *isize) { check(dfft_get_in_size(plan_, isize)); }
    inline void getInStart(size_t *istart) { check(dfft_get_in_start(plan_, istart)); }
    inline void getOutSize(size_t *osize) { check(dfft_get_out_size(plan_, osize)); }
    inline void getOutStart(size_t *ostart) { check(dfft_get_out_start(plan_, ostart)); }
    inline size_t getDomainSize() const { return dfft_domain_size(plan_); }
    inline size_t getWorkSizeDevice() const { return dfft_work_size_device(plan_); }
    inline size_t getWorkSizeHost() const { return dfft_work_size_host(plan_); }
    inline void *getWorkAreaDevice() const { return dfft_work_area_device(plan_); }
    inline void *getWorkAreaHost() const { return nullptr; }
    inline int getRank() const { return pidx; }
    inline int getWorldSize() const { return pcnt; }

protected:
    static void check(int rc)
    {
        if (rc != 0) throw std::runtime_error(std::string("dfft: ") + dfft_last_error());
    }
    dfft_plan *plan_ = nullptr;
    dfft_comm *comm_ = nullptr;
    DfftHostStagedMPI *staged_ = nullptr;
    MPI_Comm sub_ = MPI_COMM_NULL;
    int pidx = 0, pcnt = 1;
};

template <typename T> struct MPIcuFFT_Slab : MPIcuFFT<T> {
    MPIcuFFT_Slab(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1) : MPIcuFFT<T>(c, comm, max_world_size, DFFT_SLAB) {}
    using MPIcuFFT<T>::initFFT;
    void initFFT(GlobalSize *g, bool allocate = true)     // include/mpicufft_slab.hpp:103-106
    {
        Slab_Partition p(this->getWorldSize());
        MPIcuFFT<T>::initFFT(g, &p, allocate);
    }
};
template <typename T> struct MPIcuFFT_Slab_Opt1 : MPIcuFFT<T> {
    MPIcuFFT_Slab_Opt1(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1) : MPIcuFFT<T>(c, comm, max_world_size, DFFT_SLAB_OPT1) {}
};
// alternative slab sequence (include/mpicufft_slab_z_then_yx.hpp, _opt1.hpp): output [Nx][Ny][Nzc/P]
template <typename T> struct MPIcuFFT_Slab_Z_Then_YX : MPIcuFFT<T> {
    MPIcuFFT_Slab_Z_Then_YX(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1, int kind = DFFT_SLAB_Z_THEN_YX)
        : MPIcuFFT<T>(c, comm, max_world_size, kind) {}
    using MPIcuFFT<T>::initFFT;
    void initFFT(GlobalSize *g, bool allocate = true)     // include/mpicufft_slab_z_then_yx.hpp:33-37
    {
        Slab_Partition p(this->getWorldSize());
        MPIcuFFT<T>::initFFT(g, &p, allocate);
    }
};
template <typename T> struct MPIcuFFT_Slab_Z_Then_YX_Opt1 : MPIcuFFT_Slab_Z_Then_YX<T> {
    MPIcuFFT_Slab_Z_Then_YX_Opt1(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1)
        : MPIcuFFT_Slab_Z_Then_YX<T>(c, comm, max_world_size, DFFT_SLAB_Z_THEN_YX_OPT1) {}
};
// forward-only sequence with the Hermitian axis in y (include/mpicufft_slab_y_then_zx.hpp): output [Nx][(Ny/2+1)/P][Nz]
template <typename T> struct MPIcuFFT_Slab_Y_Then_ZX : MPIcuFFT<T> {
    MPIcuFFT_Slab_Y_Then_ZX(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1)
        : MPIcuFFT<T>(c, comm, max_world_size, DFFT_SLAB_Y_THEN_ZX) {}
    using MPIcuFFT<T>::initFFT;
    void initFFT(GlobalSize *g, bool allocate = true)     // include/mpicufft_slab_y_then_zx.hpp:34-35
    {
        Slab_Partition p(this->getWorldSize());
        MPIcuFFT<T>::initFFT(g, &p, allocate);
    }
};
// partial transforms execR2C/execC2R(out, in, d), include/mpicufft_pencil.hpp:101-111
template <typename T> struct MPIcuFFT_PencilBase : MPIcuFFT<T> {
    using MPIcuFFT<T>::MPIcuFFT;
    using MPIcuFFT<T>::execR2C;
    using MPIcuFFT<T>::execC2R;
    void execR2C(void *out, const void *in, int d) { this->check(dfft_exec_dim(this->plan_, out, const_cast<void *>(in), DFFT_FORWARD, d)); }
    void execC2R(void *out, const void *in, int d) { this->check(dfft_exec_dim(this->plan_, out, const_cast<void *>(in), DFFT_INVERSE, d)); }
};
template <typename T> struct MPIcuFFT_Pencil : MPIcuFFT_PencilBase<T> {
    MPIcuFFT_Pencil(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1) : MPIcuFFT_PencilBase<T>(c, comm, max_world_size, DFFT_PENCIL) {}
};
template <typename T> struct MPIcuFFT_Pencil_Opt1 : MPIcuFFT_PencilBase<T> {
    MPIcuFFT_Pencil_Opt1(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1) : MPIcuFFT_PencilBase<T>(c, comm, max_world_size, DFFT_PENCIL_OPT1) {}
};
