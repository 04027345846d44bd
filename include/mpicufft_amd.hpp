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

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <unistd.h>

#include "dfft_c.h"
#include "timer_amd.hpp"

struct GlobalSize {
    GlobalSize(size_t Nx_, size_t Ny_, size_t Nz_) : Nx(Nx_), Ny(Ny_), Nz(Nz_), Nz_out(Nz_ / 2 + 1) {}
    size_t Nx, Ny, Nz, Nz_out;
};
struct Partition { size_t P1, P2; };
struct Slab_Partition : public Partition { explicit Slab_Partition(size_t P1_) { P1 = P1_; P2 = 1; } };
struct Pencil_Partition : public Partition { Pencil_Partition(size_t P1_, size_t P2_) { P1 = P1_; P2 = P2_; } };
// include/params.hpp:58-81: per-rank extents and offsets of one decomposition stage
struct Partition_Dimensions {
    void computeOffsets()
    {
        computeStart(&size_x, &start_x);
        computeStart(&size_y, &start_y);
        computeStart(&size_z, &start_z);
    }
    std::vector<size_t> size_x, size_y, size_z;
    std::vector<size_t> start_x, start_y, start_z;

private:
    void computeStart(std::vector<size_t> *size, std::vector<size_t> *start)
    {
        size_t offset = 0;
        start->clear();
        for (size_t i = 0; i < size->size(); i++) { start->push_back(offset); offset += (*size)[i]; }
    }
};
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
        if (hsend_) (void)hipHostFree(hsend_);
        if (hrecv_) (void)hipHostFree(hrecv_);
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
        // MPI_Alltoallv takes int counts and displacements: the exchange runs in rounds of at most `piece` bytes
        // per peer so that one round's staging buffer stays below INT_MAX (C4's exchange 1 is 2 GiB per rank,
        // C5's messages are 2-4 GiB).  Every member of the group must run the same number of rounds.
        const size_t piece = ((size_t)INT_MAX / (size_t)ng) & ~(size_t)255;
        unsigned long long mymax = 0, gmax = 0;
        for (int q = 0; q < ng; q++) mymax = std::max<unsigned long long>(mymax, std::max(sc[q], rc[q]));
        if (MPI_Allreduce(&mymax, &gmax, 1, MPI_UNSIGNED_LONG_LONG, MPI_MAX, it->second) != MPI_SUCCESS) return 5;
        const size_t rounds = gmax ? (size_t)((gmax + piece - 1) / piece) : 0;
        std::vector<int> isc(ng), isd(ng), irc(ng), ird(ng);
        for (size_t r = 0; r < rounds; r++) {
            const size_t lo = r * piece;
            size_t stot = 0, rtot = 0;
            for (int q = 0; q < ng; q++) {
                const size_t s = sc[q] > lo ? std::min(piece, sc[q] - lo) : 0, v = rc[q] > lo ? std::min(piece, rc[q] - lo) : 0;
                isc[q] = (int)s; isd[q] = (int)stot; stot += s;
                irc[q] = (int)v; ird[q] = (int)rtot; rtot += v;
            }
            if (stot > hcap_s_) { if (hsend_) (void)hipHostFree(hsend_); hsend_ = nullptr; hcap_s_ = 0; if (hipHostMalloc(&hsend_, stot) != hipSuccess) return 3; hcap_s_ = stot; }
            if (rtot > hcap_r_) { if (hrecv_) (void)hipHostFree(hrecv_); hrecv_ = nullptr; hcap_r_ = 0; if (hipHostMalloc(&hrecv_, rtot) != hipSuccess) return 3; hcap_r_ = rtot; }
            for (int q = 0; q < ng; q++)
                if (isc[q] && hipMemcpyAsync(static_cast<char *>(hsend_) + isd[q], static_cast<const char *>(sendbuf) + sd[q] + lo, (size_t)isc[q],
                                             hipMemcpyDeviceToHost, stream) != hipSuccess) return 4;
            if (hipStreamSynchronize(stream) != hipSuccess) return 4;
            if (MPI_Alltoallv(hsend_, isc.data(), isd.data(), MPI_BYTE, hrecv_, irc.data(), ird.data(), MPI_BYTE, it->second) != MPI_SUCCESS)
                return 5;
            for (int q = 0; q < ng; q++)
                if (irc[q] && hipMemcpyAsync(static_cast<char *>(recvbuf) + rd[q] + lo, static_cast<const char *>(hrecv_) + ird[q], (size_t)irc[q],
                                             hipMemcpyHostToDevice, stream) != hipSuccess) return 4;
            // the next round (and my next exchange) reuses the staging buffers: drain before going on
            if (hipStreamSynchronize(stream) != hipSuccess) return 4;
        }
        return 0;
    }
    MPI_Comm comm_;
    std::map<std::tuple<int, int, int>, MPI_Comm> groups_;
    void *hsend_ = nullptr, *hrecv_ = nullptr;
    size_t hcap_s_ = 0, hcap_r_ = 0;
};

// include/mpicufft.hpp:55-105.  The reference's base class is abstract; here it carries the plan handle and
// every member function the concrete classes share.  A rank outside the FFT world (pidx >= max_world_size)
// holds no plan: its member functions are no-ops, like a rank that never constructs the reference's object.
template <typename T> class MPIcuFFT {
public:
    MPIcuFFT(Configurations config, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1, int kind = DFFT_PENCIL_OPT1)
        : config(config), kind_(kind)
    {
        int size = 1;
        MPI_Comm_size(comm, &size);
        MPI_Comm_rank(comm, &pidx);
        pcnt = size;
        MPI_Comm world = comm;
        if (max_world_size > 0 && size > max_world_size) {
            // src/mpicufft.cpp:46-51: EVERY rank that constructs the object takes part in this split (a rank that
            // does not construct one matches it with MPI_Comm_split(comm, MPI_UNDEFINED, ...) as the reference's
            // coordinator does, tests/src/pencil/random_dist_3D.cu:314-315).  Ranks beyond max_world_size get
            // MPI_COMM_NULL and stay outside.
            pcnt = max_world_size;
            MPI_Comm_split(comm, pidx < pcnt ? 0 : MPI_UNDEFINED, pidx, &sub_);
            world = sub_;
        }
        world_ = world;
        if (pidx >= pcnt) return;                       // not part of the FFT world
        if (pcnt > 1) {
            if (config.cuda_aware) {                   // device path: RCCL over xGMI, one rank per GPU
                char id[128];
                if (pidx == 0) check(dfft_rccl_unique_id(id));
                MPI_Bcast(id, 128, MPI_BYTE, 0, world);
                check(dfft_comm_create_rccl(id, pcnt, pidx, &comm_));
                // pencil classes: the row- and the column-group exchange use disjoint xGMI links -- a duplicated communicator lets them be
                // on the wire together, two more carry the relay's first hops (include/dfft_c.h "dup_channel"; collective: every rank
                // of the world constructs the same class).  A librccl without ncclCommSplit keeps the single communicator.
                if (kind == DFFT_PENCIL || kind == DFFT_PENCIL_OPT1) (void)dfft_comm_set_option(comm_, "dup_channel", 3);
            } else {                                   // host-staged MPI, ranks may share a GPU
                staged_ = new DfftHostStagedMPI(world);
                check(dfft_comm_create_callback(pcnt, pidx, &DfftHostStagedMPI::alltoallv, staged_, &comm_));
            }
            // transport knob without a counterpart in Configurations: DFFT_RELAY = 1 | 2 | 3 routes the group exchanges of pencil
            // grids through the two-hop relay (include/dfft_c.h: dfft_comm_set_option "relay"; every rank must see the same value)
            if (const char *rl = getenv("DFFT_RELAY")) check(dfft_comm_set_option(comm_, "relay", atol(rl)));
        }
        dfft_config c{config.cuda_aware, config.warmup_rounds, (int)config.comm_method, (int)config.send_method,
                      (int)config.comm_method2, (int)config.send_method2};
        check(dfft_plan_create(&plan_, kind, sizeof(T) == 8 ? DFFT_F64 : DFFT_F32, &c, comm_, pidx, -1));
    }
    virtual ~MPIcuFFT()
    {
        dfft_plan_destroy(plan_);
        if (comm_) dfft_comm_destroy(comm_);
        delete staged_;
        delete timer;
        if (sub_ != MPI_COMM_NULL) MPI_Comm_free(&sub_);
    }
    virtual void initFFT(GlobalSize *global_size, Partition *partition, bool allocate = true)
    {
        if (!plan_) return;
        if (!global_size || !partition) throw std::runtime_error("GlobalSize or Partition not initialized!");
        startTimer(global_size, partition);
        check(dfft_init(plan_, global_size->Nx, global_size->Ny, global_size->Nz, (int)partition->P1,
                        (int)partition->P2, /*c2c=*/0, allocate));
        if (timer) {
            check(dfft_enable_phase_timing(plan_, 1));
            timer->stop_store("init");
        }
    }
    // complex-to-complex plan (extension): same layouts with Nz_out = Nz
    void initFFT_C2C(GlobalSize *g, Partition *p, bool allocate = true)
    {
        if (!plan_) return;
        startTimer(g, p);
        check(dfft_init(plan_, g->Nx, g->Ny, g->Nz, (int)p->P1, (int)p->P2, 1, allocate));
        if (timer) {
            check(dfft_enable_phase_timing(plan_, 1));
            timer->stop_store("init");
        }
    }
    virtual void setWorkArea(void *device = nullptr, void *host = nullptr) { if (plan_) check(dfft_set_work_area(plan_, device, host)); }
    virtual void execR2C(void *out, const void *in) { if (plan_) timed(DFFT_FORWARD, [&] { return dfft_exec_r2c(plan_, out, in); }); }
    virtual void execC2R(void *out, const void *in) { if (plan_) timed(DFFT_INVERSE, [&] { return dfft_exec_c2r(plan_, out, const_cast<void *>(in)); }); }
    void execC2C(void *out, void *in, int direction) { if (plan_) timed(direction, [&] { return dfft_exec_c2c(plan_, out, in, direction); }); }
    // extension (no counterpart in the reference): buffers on the physical backing this plan's passes run fastest on, see
    // dfft_tune_placement in dfft_c.h; free them with dfft_free.  Collective on a multi-rank plan.
    void tunePlacement(const void *in, int tries, void **out, void **back = nullptr)
    {
        if (plan_) check(dfft_tune_placement(plan_, in, tries, out, back, nullptr, 0, nullptr));
    }
    // extensions (no counterpart in the reference).  setOption: the engine's knobs by name (dfft_set_option; before initFFT), e.g.
    // setOption("spectral_layout", 1) keeps the spectrum x-contiguous, [yo][zs][Nx]: getOutStrides gives the element strides of
    // (kx, ky, kz) in `out` for whichever layout is in use.  allocate / release: device memory on the library's default backing
    // (dfft_malloc(DFFT_CHUNK_DEFAULT)) for `out` and the inverse's output -- a hipMalloc'ed target costs the scatter passes ~10 %.
    void setOption(const char *key, long value) { if (plan_) check(dfft_set_option(plan_, key, value)); }
    inline void getOutStrides(size_t *ostrides) { if (plan_) check(dfft_get_out_strides(plan_, ostrides)); }
    static void *allocate(size_t bytes) { void *ptr = nullptr; check(dfft_malloc(bytes, DFFT_CHUNK_DEFAULT, &ptr)); return ptr; }
    static void release(void *ptr) { check(dfft_free(ptr)); }
    // the section timer of the reference's classes (include/mpicufft_pencil.hpp:263-287, include/mpicufft_slab.hpp:208-222):
    // one block of the CSV per exec* once the warm-up rounds are used up
    Timer *getTimer() const { return timer; }
    virtual inline void getInSize(size_t *isize) { if (plan_) check(dfft_get_in_size(plan_, isize)); }
    virtual inline void getInStart(size_t *istart) { if (plan_) check(dfft_get_in_start(plan_, istart)); }
    virtual inline void getOutSize(size_t *osize) { if (plan_) check(dfft_get_out_size(plan_, osize)); }
    virtual inline void getOutStart(size_t *ostart) { if (plan_) check(dfft_get_out_start(plan_, ostart)); }
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
    // Timer sections and CSV name of the class, as the reference's initFFT builds them (src/pencil/mpicufft_pencil.cpp:67-73,
    // src/pencil/mpicufft_pencil_opt1.cpp:50-56, src/slab/default/mpicufft_slab.cpp:99-105, mpicufft_slab_opt1.cpp:39-44,
    // src/slab/z_then_yx/mpicufft_slab_z_then_yx.cpp:76-81, src/slab/y_then_zx/mpicufft_slab_y_then_zx.cpp:73-79)
    void startTimer(GlobalSize *g, Partition *partition)
    {
        const bool pencil = kind_ == DFFT_PENCIL || kind_ == DFFT_PENCIL_OPT1;
        const bool zyx = kind_ == DFFT_SLAB_Z_THEN_YX || kind_ == DFFT_SLAB_Z_THEN_YX_OPT1, yzx = kind_ == DFFT_SLAB_Y_THEN_ZX;
        const int opt = (kind_ == DFFT_PENCIL_OPT1 || kind_ == DFFT_SLAB_OPT1 || kind_ == DFFT_SLAB_Z_THEN_YX_OPT1) ? 1 : 0;
        const std::string sub = pencil ? "/pencil" : zyx ? "/slab_z_then_yx" : yzx ? "/slab_y_then_zx" : "/slab_default";
        // The section timer costs every exec an event pair per pass and, after the warm-up rounds, an MPI_Gatherv plus a file
        // append (the reference's hidden collective, src/pencil/mpicufft_pencil_opt1.cpp:1515-1518).  It only runs when its CSV can
        // be written: rank 0 (the gathering rank) tests the directory, everybody follows its verdict.
        delete timer;
        timer = nullptr;
        int usable = 0;
        if (pidx == 0 && !config.benchmark_dir.empty()) {
            (void)mkdir(config.benchmark_dir.c_str(), 0777);
            (void)mkdir((config.benchmark_dir + sub).c_str(), 0777);
            usable = access((config.benchmark_dir + sub).c_str(), W_OK | X_OK) == 0;
        }
        MPI_Bcast(&usable, 1, MPI_INT, 0, world_);
        if (!usable) return;
        std::string f = config.benchmark_dir + sub + "/test_" + std::to_string(opt) + "_" + std::to_string((int)config.comm_method) + "_" +
                        std::to_string((int)config.send_method);
        if (pencil) f += "_" + std::to_string((int)config.comm_method2) + "_" + std::to_string((int)config.send_method2);
        f += "_" + std::to_string(g->Nx) + "_" + std::to_string(g->Ny) + "_" + std::to_string(g->Nz) + "_" + std::to_string((int)config.cuda_aware);
        f += pencil ? "_" + std::to_string(partition->P1) + "_" + std::to_string(partition->P2) : "_" + std::to_string(pcnt);
        f += ".csv";
        std::vector<std::string> d = {"init"};
        const char *tr[] = {"(First Send)", "(Packing)", "(Start Local Transpose)", "(Start Receive)", "(First Receive)", "(Finished Receive)",
                            "(Start All2All)", "(Finished All2All)", "(Unpacking)"};
        auto transpose = [&](const std::string &prefix) { for (const char *t : tr) d.push_back(prefix + "Transpose " + t); };
        if (pencil) {
            d.push_back("1D FFT Z-Direction");
            transpose("First ");
            d.insert(d.end(), {"First Transpose (Send Complete)", "1D FFT Y-Direction"});
            transpose("Second ");
            d.push_back("1D FFT X-Direction");
        } else if (zyx) {
            d.push_back("1D FFT Z-Direction");
            transpose("");
            d.push_back("2D FFT Y-X-Direction");
        } else if (yzx) {
            d.insert(d.end(), {"1D FFT Y-Direction", "Transpose (First Send)", "Transpose (Packing)", "Transpose (Start Local Transpose)",
                               "Transpose (Start Receive)", "Transpose (Finished Receive)", "2D FFT Z-X-Direction"});
        } else {
            d.insert(d.end(), {"2D FFT (Sync)", "2D FFT Y-Z-Direction"});
            transpose("");
            d.push_back("1D FFT X-Direction");
        }
        d.push_back("Run complete");
        timer = new Timer(world_, 0, pcnt, pidx, d, f);
        timer->start();
    }
    // runs one exec, then stores the sections it went through (cumulative milliseconds like the reference's stop points; the
    // library's device events per phase instead of MPI_Wtime() after host-side synchronisations) and appends a block to the CSV
    // when the warm-up rounds are used up (src/pencil/mpicufft_pencil_opt1.cpp:1515-1518)
    // partial = true (exec*(out, in, d) with d < 3): fewer phases come back than the class's section list assumes, so only the
    // total ("Run complete") is recorded
    template <typename F> void timed(int direction, F &&exec, bool partial = false)
    {
        if (!timer) { check(exec()); return; }
        timer->start();
        check(exec());
        timer->stop("Run complete");
        float ph[5] = {0, 0, 0, 0, 0};
        const int n = partial ? 0 : dfft_get_phase_times(plan_, ph, 5);
        const bool pencil = kind_ == DFFT_PENCIL || kind_ == DFFT_PENCIL_OPT1;
        const bool zyx = kind_ == DFFT_SLAB_Z_THEN_YX || kind_ == DFFT_SLAB_Z_THEN_YX_OPT1, yzx = kind_ == DFFT_SLAB_Y_THEN_ZX;
        const char *done = config.comm_method == Peer2Peer ? "(Finished Receive)" : "(Finished All2All)";
        const char *done2 = config.comm_method2 == Peer2Peer ? "(Finished Receive)" : "(Finished All2All)";
        // section reached after phase i of this direction (phases in execution order: forward z, exchange 1, y, exchange 2, x;
        // the inverse runs them backwards); empty = the phase closes no section of this class
        std::string name[5];
        if (pencil) {
            const std::string f[5] = {"1D FFT Z-Direction", std::string("First Transpose ") + done, "1D FFT Y-Direction",
                                      std::string("Second Transpose ") + done2, "1D FFT X-Direction"};
            for (int i = 0; i < 5; i++) name[i] = direction == DFFT_FORWARD ? f[i] : f[4 - i];
        } else if (zyx) {
            const std::string f[5] = {"1D FFT Z-Direction", std::string("Transpose ") + done, "", "", "2D FFT Y-X-Direction"};
            for (int i = 0; i < 5; i++) name[i] = f[i];
            if (direction != DFFT_FORWARD) { name[0] = ""; name[2] = "2D FFT Y-X-Direction"; name[3] = std::string("Transpose ") + done; name[1] = ""; name[4] = "1D FFT Z-Direction"; }
        } else if (yzx) {
            name[0] = "1D FFT Y-Direction"; name[1] = "Transpose (Finished Receive)"; name[4] = "2D FFT Z-X-Direction";
        } else {
            if (direction == DFFT_FORWARD) { name[2] = "2D FFT Y-Z-Direction"; name[3] = std::string("Transpose ") + done; name[4] = "1D FFT X-Direction"; }
            else { name[0] = "1D FFT X-Direction"; name[1] = std::string("Transpose ") + done; name[4] = "2D FFT Y-Z-Direction"; }
        }
        double cum = 0;
        for (int i = 0; i < n && i < 5; i++) {
            cum += ph[i];
            if (!name[i].empty()) timer->store(name[i], cum);
        }
        timer->store("Run complete");
        if (timer->duration("Run complete") < cum) timer->store("Run complete", cum);
        if (config.warmup_rounds == 0) timer->gather();
        else config.warmup_rounds--;
    }
    Configurations config;
    int kind_ = DFFT_PENCIL_OPT1;
    MPI_Comm world_ = MPI_COMM_WORLD;
    Timer *timer = nullptr;
    // slab classes: initFFT(global_size, nullptr, allocate) is legal, the partition is the world size
    // (include/mpicufft_slab.hpp:103-106)
    void initSlab(GlobalSize *g, Partition *partition, bool allocate)
    {
        Slab_Partition p((size_t)pcnt);
        MPIcuFFT<T>::initFFT(g, partition ? partition : &p, allocate);
    }
    dfft_plan *plan_ = nullptr;
    dfft_comm *comm_ = nullptr;
    DfftHostStagedMPI *staged_ = nullptr;
    MPI_Comm sub_ = MPI_COMM_NULL;
    int pidx = 0, pcnt = 1;
};

// include/mpicufft_slab.hpp:85 and include/mpicufft_slab_opt1.hpp:70 (Opt1 derives from the opt0 class, so
// `MPIcuFFT_Slab<T> *p = new MPIcuFFT_Slab_Opt1<T>(...)` of tests/src/slab/random_dist_default.cu:194-198 compiles)
template <typename T> class MPIcuFFT_Slab : public MPIcuFFT<T> {
public:
    MPIcuFFT_Slab(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1) : MPIcuFFT<T>(c, comm, max_world_size, DFFT_SLAB) {}
    virtual void initFFT(GlobalSize *g, Partition *partition, bool allocate = true) { this->initSlab(g, partition, allocate); }
    void initFFT(GlobalSize *g, bool allocate = true) { initFFT(g, nullptr, allocate); }     // include/mpicufft_slab.hpp:103-106

protected:
    MPIcuFFT_Slab(Configurations c, MPI_Comm comm, int max_world_size, int kind) : MPIcuFFT<T>(c, comm, max_world_size, kind) {}
};
template <typename T> class MPIcuFFT_Slab_Opt1 : public MPIcuFFT_Slab<T> {
public:
    MPIcuFFT_Slab_Opt1(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1) : MPIcuFFT_Slab<T>(c, comm, max_world_size, DFFT_SLAB_OPT1) {}
    void initFFT(GlobalSize *g, Partition *partition, bool allocate = true) { this->initSlab(g, partition, allocate); }
    void initFFT(GlobalSize *g, bool allocate = true) { this->initFFT(g, nullptr, allocate); }
};
// alternative slab sequence (include/mpicufft_slab_z_then_yx.hpp:29, _opt1.hpp:22): output [Nx][Ny][Nzc/P]
template <typename T> class MPIcuFFT_Slab_Z_Then_YX : public MPIcuFFT<T> {
public:
    MPIcuFFT_Slab_Z_Then_YX(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1) : MPIcuFFT<T>(c, comm, max_world_size, DFFT_SLAB_Z_THEN_YX) {}
    virtual void initFFT(GlobalSize *g, Partition *partition, bool allocate = true) { this->initSlab(g, partition, allocate); }
    void initFFT(GlobalSize *g, bool allocate = true) { initFFT(g, nullptr, allocate); }     // include/mpicufft_slab_z_then_yx.hpp:33-37

protected:
    MPIcuFFT_Slab_Z_Then_YX(Configurations c, MPI_Comm comm, int max_world_size, int kind) : MPIcuFFT<T>(c, comm, max_world_size, kind) {}
};
template <typename T> class MPIcuFFT_Slab_Z_Then_YX_Opt1 : public MPIcuFFT_Slab_Z_Then_YX<T> {
public:
    MPIcuFFT_Slab_Z_Then_YX_Opt1(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1)
        : MPIcuFFT_Slab_Z_Then_YX<T>(c, comm, max_world_size, DFFT_SLAB_Z_THEN_YX_OPT1) {}
    void initFFT(GlobalSize *g, Partition *partition, bool allocate = true) { this->initSlab(g, partition, allocate); }
    void initFFT(GlobalSize *g, bool allocate = true) { this->initFFT(g, nullptr, allocate); }
};
// forward-only sequence with the Hermitian axis in y (include/mpicufft_slab_y_then_zx.hpp:29): output [Nx][(Ny/2+1)/P][Nz]
template <typename T> class MPIcuFFT_Slab_Y_Then_ZX : public MPIcuFFT<T> {
public:
    MPIcuFFT_Slab_Y_Then_ZX(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1)
        : MPIcuFFT<T>(c, comm, max_world_size, DFFT_SLAB_Y_THEN_ZX) {}
    void initFFT(GlobalSize *g, Partition *partition, bool allocate = true) { this->initSlab(g, partition, allocate); }
    void initFFT(GlobalSize *g, bool allocate = true) { this->initFFT(g, nullptr, allocate); }     // include/mpicufft_slab_y_then_zx.hpp:34-35
};
// include/mpicufft_pencil.hpp:88-122 and include/mpicufft_pencil_opt1.hpp:23 (Opt1 derives from the opt0 class:
// tests/src/pencil/random_dist_3D.cu:183-187 declares MPIcuFFT_Pencil<T>* and news either)
template <typename T> class MPIcuFFT_Pencil : public MPIcuFFT<T> {
public:
    MPIcuFFT_Pencil(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1) : MPIcuFFT<T>(c, comm, max_world_size, DFFT_PENCIL) {}
    virtual void execR2C(void *out, const void *in) { this->execR2C(out, in, 3); }
    virtual void execC2R(void *out, const void *in) { this->execC2R(out, in, 3); }
    // partial transforms execR2C/execC2R(out, in, d), include/mpicufft_pencil.hpp:101-111
    virtual void execR2C(void *out, const void *in, int d)
    {
        if (this->plan_) this->timed(DFFT_FORWARD, [&] { return dfft_exec_dim(this->plan_, out, const_cast<void *>(in), DFFT_FORWARD, d); }, d < 3);
    }
    virtual void execC2R(void *out, const void *in, int d)
    {
        if (this->plan_) this->timed(DFFT_INVERSE, [&] { return dfft_exec_dim(this->plan_, out, const_cast<void *>(in), DFFT_INVERSE, d); }, d < 3);
    }
    // include/mpicufft_pencil.hpp:112-116; tables as built in src/pencil/mpicufft_pencil_opt1.cpp:70-93
    void getPartitionDimensions(Partition_Dimensions &input_dim_, Partition_Dimensions &transposed_dim_, Partition_Dimensions &output_dim_)
    {
        if (!this->plan_) return;
        Partition_Dimensions *dims[3] = {&input_dim_, &transposed_dim_, &output_dim_};
        for (int which = 0; which < 3; which++) {
            std::vector<size_t> *sz[3] = {&dims[which]->size_x, &dims[which]->size_y, &dims[which]->size_z};
            std::vector<size_t> *st[3] = {&dims[which]->start_x, &dims[which]->start_y, &dims[which]->start_z};
            for (int axis = 0; axis < 3; axis++) {
                size_t n = 0;
                this->check(dfft_get_partition_dimensions(this->plan_, which, axis, nullptr, nullptr, 0, &n));
                sz[axis]->assign(n, 0); st[axis]->assign(n, 0);
                this->check(dfft_get_partition_dimensions(this->plan_, which, axis, sz[axis]->data(), st[axis]->data(), n, &n));
            }
        }
    }

protected:
    MPIcuFFT_Pencil(Configurations c, MPI_Comm comm, int max_world_size, int kind) : MPIcuFFT<T>(c, comm, max_world_size, kind) {}
};
template <typename T> class MPIcuFFT_Pencil_Opt1 : public MPIcuFFT_Pencil<T> {
public:
    MPIcuFFT_Pencil_Opt1(Configurations c, MPI_Comm comm = MPI_COMM_WORLD, int max_world_size = -1) : MPIcuFFT_Pencil<T>(c, comm, max_world_size, DFFT_PENCIL_OPT1) {}
};
