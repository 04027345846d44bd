// timer_amd.hpp -- the reference's section timer (include/timer.hpp:24-55, src/timer.cpp:21-101) for the C++ shim:
// same constructor, same member functions, same CSV.  The file gets a header row ",0,1,...,P-1," when it does not
// exist yet, and every gather() appends an empty line followed by one row per section, "desc,t_rank0,t_rank1,...,"
// with cumulative milliseconds since start() -- what eval/ of the reference parses.
//
// One addition: store(desc, ms) records a duration that was measured elsewhere.  The shim's exec* are single calls into
// libdfft_amd.so, so the per-section times come from the library's device events (dfft_get_phase_times) instead of
// MPI_Wtime() between host-side synchronisations.
#pragma once
#include <mpi.h>

#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

class Timer {
public:
    Timer(MPI_Comm comm, int p_gather, int pcnt, int pidx, std::vector<std::string> descs, std::string filename)
        : comm(comm), p_gather(p_gather), pcnt(pcnt), pidx(pidx), descs(descs), filename(filename)
    {
        durations.resize(descs.size(), 0);
        tstop_points.resize(descs.size(), 0);
    }
    void start() { tstart = MPI_Wtime(); }
    void stop(std::string desc) { tstop_points[index_of(desc)] = MPI_Wtime(); }
    void store(std::string desc) { const size_t i = index_of(desc); durations[i] = (tstop_points[i] - tstart) * 1000; }
    void store(std::string desc, double ms) { durations[index_of(desc)] = ms; }
    void stop_store(std::string desc) { stop(desc); store(desc); }
    void setFileName(std::string filename_) { filename = filename_; }
    const std::string &fileName() const { return filename; }
    double duration(std::string desc) const { return durations[index_of(desc)]; }
    // Appends this iteration's block to the CSV on rank `p_gather` (the format of src/timer.cpp:81-100, byte for byte: pinned
    // against the reference's own Timer built from its sources, oracle/_ref, tests/test_ref_timer.py).  The first `pcnt` ranks
    // of `comm` are the workers (a test may run its FFT on fewer ranks than MPI_COMM_WORLD holds); each sends its row of
    // sections to the writer as one message, the other ranks have nothing to do.
    void gather()
    {
        const size_t nsec = durations.size();
        if (pidx != p_gather) {
            if (pidx < pcnt) MPI_Send(durations.data(), (int)nsec, MPI_DOUBLE, p_gather, kTag, comm);
            return;
        }
        std::vector<std::vector<double>> column(pcnt, std::vector<double>(nsec, 0.0));   // column[worker][section]
        for (int w = 0; w < pcnt; w++) {
            if (w == pidx) column[w] = durations;
            else MPI_Recv(column[w].data(), (int)nsec, MPI_DOUBLE, w, kTag, comm, MPI_STATUS_IGNORE);
        }
        std::ostringstream block;                   // default ostream formatting of a double: what eval/ of the reference parses
        if (!std::ifstream(filename).good()) {      // a new file starts with the rank header
            block << ",";
            for (int w = 0; w < pcnt; w++) block << w << ",";
        }
        block << "\n";
        for (size_t s = 0; s < nsec; s++) {
            block << descs[s] << ",";
            for (int w = 0; w < pcnt; w++) block << column[w][s] << ",";
            block << "\n";
        }
        std::ofstream(filename, std::ios_base::app) << block.str();
    }

protected:
    size_t index_of(const std::string &desc) const
    {
        const auto it = std::find(descs.begin(), descs.end(), desc);
        // (the reference's stop_store on an unknown section writes through a default-constructed map slot; here it is an error
        // instead of a silent overwrite of section 0, "init")
        if (it == descs.end()) throw std::runtime_error("Timer: unknown section \"" + desc + "\"");
        return (size_t)std::distance(descs.begin(), it);
    }
    static constexpr int kTag = 0x7d1;
    MPI_Comm comm;
    int p_gather;
    int pcnt, pidx;
    std::vector<double> durations;
    std::vector<double> tstop_points;
    double tstart = 0;
    std::vector<std::string> descs;
    std::string filename;
};
