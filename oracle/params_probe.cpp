// params_probe.cpp -- test infrastructure (oracle/): prints the layout and the behaviour of the parameter structs of the drop-in
// boundary (GlobalSize, Partition, Slab_Partition, Pencil_Partition, Partition_Dimensions, Configurations and
// the two enums), once compiled against the REFERENCE's own include/params.hpp (-DPROBE_REFERENCE_PARAMS; the header needs nothing
// the image lacks) and once against include/mpicufft_amd.hpp.  The two outputs must be identical: a reference call site that fills
// these structs is then binary-compatible with the shim (tests/test_ref_timer.py).
#include <cstddef>
#include <cstdio>
#include <string>
#include <vector>

#ifdef PROBE_REFERENCE_PARAMS
#include "params.hpp"
#else
#include "mpicufft_amd.hpp"
#endif

#define SHOW_SIZE(T) std::printf("sizeof(%s) = %zu, alignof = %zu\n", #T, sizeof(T), alignof(T))
#define SHOW_OFF(T, f) std::printf("offsetof(%s, %s) = %zu, sizeof = %zu\n", #T, #f, offsetof(T, f), sizeof(((T *)nullptr)->f))

static void show(const char *name, const std::vector<size_t> &v)
{
    std::printf("%s = [", name);
    for (size_t i = 0; i < v.size(); i++) std::printf("%s%zu", i ? ", " : "", v[i]);
    std::printf("]\n");
}

int main()
{
    SHOW_SIZE(GlobalSize); SHOW_OFF(GlobalSize, Nx); SHOW_OFF(GlobalSize, Ny); SHOW_OFF(GlobalSize, Nz); SHOW_OFF(GlobalSize, Nz_out);
    SHOW_SIZE(Partition); SHOW_OFF(Partition, P1); SHOW_OFF(Partition, P2);
    SHOW_SIZE(Slab_Partition); SHOW_SIZE(Pencil_Partition);
    SHOW_SIZE(Partition_Dimensions);
    SHOW_OFF(Partition_Dimensions, size_x); SHOW_OFF(Partition_Dimensions, size_y); SHOW_OFF(Partition_Dimensions, size_z);
    SHOW_OFF(Partition_Dimensions, start_x); SHOW_OFF(Partition_Dimensions, start_y); SHOW_OFF(Partition_Dimensions, start_z);
    SHOW_SIZE(Configurations);
    SHOW_OFF(Configurations, cuda_aware); SHOW_OFF(Configurations, warmup_rounds); SHOW_OFF(Configurations, comm_method);
    SHOW_OFF(Configurations, send_method); SHOW_OFF(Configurations, benchmark_dir); SHOW_OFF(Configurations, comm_method2);
    SHOW_OFF(Configurations, send_method2);
    std::printf("enum CommunicationMethod: Peer2Peer = %d, All2All = %d (size %zu)\n", (int)Peer2Peer, (int)All2All, sizeof(CommunicationMethod));
    std::printf("enum SendMethod: Sync = %d, Streams = %d, MPI_Type = %d (size %zu)\n", (int)Sync, (int)Streams, (int)MPI_Type, sizeof(SendMethod));
    for (size_t nz : {12u, 13u, 1024u, 2u, 1u}) {
        GlobalSize g(10, 9, nz);
        std::printf("GlobalSize(10, 9, %zu): Nx %zu Ny %zu Nz %zu Nz_out %zu\n", nz, g.Nx, g.Ny, g.Nz, g.Nz_out);
    }
    Slab_Partition sp(5);
    Pencil_Partition pp(2, 4);
    Partition *base = &pp;
    std::printf("Slab_Partition(5): P1 %zu P2 %zu; Pencil_Partition(2, 4) through Partition*: P1 %zu P2 %zu\n", sp.P1, sp.P2, base->P1, base->P2);
    Partition_Dimensions d;
    d.size_x = {3, 2, 2}; d.size_y = {129, 128, 128, 128}; d.size_z = {};
    d.computeOffsets();
    show("size_x", d.size_x); show("start_x", d.start_x); show("size_y", d.size_y); show("start_y", d.start_y); show("start_z", d.start_z);
    Configurations c = {true, 10, All2All, MPI_Type, "../benchmarks", Peer2Peer, Streams};
    std::printf("Configurations{...}: %d %d %d %d %s %d %d\n", (int)c.cuda_aware, c.warmup_rounds, (int)c.comm_method, (int)c.send_method,
                c.benchmark_dir.c_str(), (int)c.comm_method2, (int)c.send_method2);
    return 0;
}
