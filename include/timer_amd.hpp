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
#include <sys/stat.h>

#include <algorithm>
#include <fstream>
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
    // src/timer.cpp:58-101: every worker contributes the same number of values; ranks of `comm` beyond pcnt contribute none
    void gather()
    {
        int world_size = 1;
        MPI_Comm_size(comm, &world_size);
        const int send_size = (int)durations.size();
        std::vector<int> recv_count(pcnt, send_size);
        recv_count.resize(world_size, 0);
        std::vector<int> recv_displ(world_size, 0);
        for (int i = 1; i < world_size; i++) recv_displ[i] = recv_displ[i - 1] + recv_count[i - 1];
        std::vector<double> all;
        if (pidx == p_gather) all.resize((size_t)send_size * pcnt, 0);
        MPI_Gatherv(durations.data(), send_size, MPI_DOUBLE, all.data(), recv_count.data(), recv_displ.data(), MPI_DOUBLE, p_gather, comm);
        if (pidx != p_gather) return;
        std::ofstream f;
        struct stat st;
        if (stat(filename.c_str(), &st) != 0) {
            f.open(filename);
            f << ",";
            for (int i = 0; i < pcnt; i++) f << i << ",";
        } else {
            f.open(filename, std::ios_base::app);
        }
        f << "\n";
        for (size_t i = 0; i < durations.size(); i++) {
            f << descs[i] << ",";
            for (int j = 0; j < pcnt; j++) f << all[(size_t)j * durations.size() + i] << ",";
            f << "\n";
        }
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
    MPI_Comm comm;
    int p_gather;
    int pcnt, pidx;
    std::vector<double> durations;
    std::vector<double> tstop_points;
    double tstart = 0;
    std::vector<std::string> descs;
    std::string filename;
};
