// timer_probe.cpp -- test infrastructure (oracle/): drives a section timer with FIXED durations and lets it write its CSV,
// so that the bytes written by the reference's own Timer (src/timer.cpp, built UNMODIFIED from /root/reference into
// oracle/_ref/ by `make -C oracle _ref`: it needs nothing but <mpi.h>, which this image has) can be compared with the bytes
// written by include/timer_amd.hpp.  Compiled twice from this one source:
//   -DPROBE_REFERENCE_TIMER  -> includes the reference's include/timer.hpp, links oracle/_ref/libref_timer.so
//   (otherwise)              -> includes include/timer_amd.hpp (header only)
// usage: mpiexec -n W timer_probe <csv> <pcnt> <p_gather> <rounds>
#include <mpi.h>

#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#ifdef PROBE_REFERENCE_TIMER
#include "timer.hpp"
#else
#include "timer_amd.hpp"
#endif

// both timers keep `durations` protected: a derived class can fill them without touching the clock
struct Probe : Timer {
    using Timer::Timer;
    void fill(int rank, int round)
    {
        // magnitudes that exercise every branch of the default ostream formatting of a double (fixed, scientific, integers,
        // zero, negative); exactly representable inputs, so both sides print from identical bits
        static const double base[] = {0.0, 1.0, 0.5, 123456.75, 1234567.0, 1.0 / 1024 / 1024 / 1024, 3.0e21, -2.25, 1.0 / 3.0, 99999.95};
        for (size_t s = 0; s < durations.size(); s++)
            durations[s] = base[(s + (size_t)rank * 3 + (size_t)round) % 10] * (1.0 + rank) + std::ldexp((double)s, -round);
    }
};

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int rank = 0, world = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &world);
    if (argc < 5) { MPI_Finalize(); return 2; }
    const std::string csv = argv[1];
    const int pcnt = std::atoi(argv[2]), p_gather = std::atoi(argv[3]), rounds = std::atoi(argv[4]);
    // the section names of include/mpicufft_pencil.hpp:263-287 that a forward run stores first, plus one with a blank
    std::vector<std::string> descs = {"init", "1D FFT Z-Direction", "First Transpose (First Send)", "First Transpose (Packing)",
                                      "First Transpose (Start Local Transpose)", "Run complete"};
    Probe t(MPI_COMM_WORLD, p_gather, pcnt, rank, descs, csv);
    for (int r = 0; r < rounds; r++) {
        t.fill(rank, r);
        t.gather();
        MPI_Barrier(MPI_COMM_WORLD);
    }
    MPI_Finalize();
    return 0;
}
