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

#include <stdexcept>
#include <string>
#include <vector>

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

template <typename T> class MPIcuFFT {
public:
    MPIcuFFT(Configurations config, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1, int kind = DFFT_PENCIL_OPT1)
    {
        MPI_Comm_size(comm, &pcnt);
        MPI_Comm_rank(comm, &pidx);
        if (max_world_size > 0 && max_world_size < pcnt) pcnt = max_world_size;   // src/mpicufft.cpp:46-51
        if (pidx >= pcnt) return;                                                 // not part of the FFT world
        if (pcnt > 1) {
            char id[128];
            if (pidx == 0) check(dfft_rccl_unique_id(id));
            MPI_Comm sub;
            MPI_Comm_split(comm, 0, pidx, &sub);
            MPI_Bcast(id, 128, MPI_BYTE, 0, sub);
            MPI_Comm_free(&sub);
            check(dfft_comm_create_rccl(id, pcnt, pidx, &comm_));
        }
        dfft_config c{config.cuda_aware, config.warmup_rounds, (int)config.comm_method, (int)config.send_method,
                      (int)config.comm_method2, (int)config.send_method2};
        check(dfft_plan_create(&plan_, kind, sizeof(T) == 8 ? DFFT_F64 : DFFT_F32, &c, comm_, pidx, max_world_size));
    }
    virtual ~MPIcuFFT()
    {
        dfft_plan_destroy(plan_);
        if (comm_) dfft_comm_destroy(comm_);
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
