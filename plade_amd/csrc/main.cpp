// plade_amd/csrc/main.cpp -- the PLADE command line on the GPU library.
//
//   PLADE target.ply source.ply result.txt      register one pair
//   PLADE file_pairs.txt result.txt             batch mode
//
// What is kept from the reference's driver (code/PLADE/main.cpp:30-159) is its *contract*: the positional
// arguments, every console / result-file string (collected in `text` below, each with the line it comes from),
// the result-file grammar and the exit codes.  The control flow is this program's own:
//
//   * the pair list is read up front and turned into jobs,
//   * PLADE_GPUS=N x PLADE_INFLIGHT=M worker threads (one plade_ctx each) take PLADE_GROUP (default 4) consecutive jobs at a
//     time from a shared counter and register them as one group (pairs are independent; the plane extraction of a group's
//     clouds is one GPU launch sequence; PLY parsing of one group overlaps the GPU work of the others),
//   * an ordered writer appends every pair's block to the result file as soon as that pair AND all earlier
//     pairs are done, and flushes it -- like the reference, which writes each block when its registration
//     returns (main.cpp:134-146), a batch that is interrupted leaves a valid prefix of the results behind,
//   * each worker's console output is collected per pair and printed in input order with the block, so the
//     console reads like the reference's sequential run whatever the number of workers.
#include "plade.h"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <thread>
#include <unistd.h>

namespace {

// ---- every user-visible string of the reference's driver, by source line ------------------------------------
namespace text {
const char *const usage =   // main.cpp:49-74
    "PLADE can register two point clouds dominated by planar structures. It can be used in two ways.\n"
    "-------------------------------------------------------------------------------------------------\n"
    "Usage 1: register a 'source' point cloud to a 'target' point cloud.\n"
    "    ---------------------------------------------------------------------------------------------\n"
    "    You can call PLADE with three arguments. The first two are the file names of a target point\n"
    "    cloud and a source point cloud (the target point cloud file name always comes first). The\n"
    "    third argument specifies the result file name. Below is an example:\n"
    "         ./PLADE  room_target.ply  room_source.ply  result.txt\n"
    "    The target point cloud file name always comes first, and both point cloud files must be in\n"
    "    the 'ply' format. The result file will store the registration result, which is a 4 by 4\n"
    "    transformation matrix that aligns the source point cloud to the target point cloud.\n"
    "-------------------------------------------------------------------------------------------------\n"
    "Usage 2: register a bunch of point cloud pairs.\n"
    "    ---------------------------------------------------------------------------------------------\n"
    "    You can call PLADE with two arguments: a file (e.g., file_pairs.txt) specifying all pairs\n"
    "    of target/source point cloud files and a result file. Below is an example:\n"
    "         ./PLADE  file_pairs.txt  result.txt\n"
    "    In 'file_pairs.txt', every two consecutive lines store two file names. The first line is the\n"
    "    file name of a target point cloud, and the second line is the file name of a source point cloud.\n"
    "    Both point cloud files must be in the 'ply' format. The result file will store the registration\n"
    "    results, a set of 4 by 4 transformation matrices. Each matrix aligns a source point cloud to\n"
    "    its corresponding target point cloud.\n";
const char *const cannot_open_result = "failed opening the result file: ";                                      // main.cpp:81,102
const char *const cannot_open_list = "failed opening the file containing pairs of point cloud names: ";         // main.cpp:96
const char *const missing_file = "file doesn't exist: ";                                                        // main.cpp:127
const char *const target_tag = "target: ";                                                                      // main.cpp:86,134
const char *const source_tag = "source: ";                                                                      // main.cpp:87,135
const char *const matrix_tag = "transformation:\n";                                                             // main.cpp:88,138
const char *const failed_block = "registration failed, an identity matrix is recorded:\n";                      // main.cpp:93,142
const char *const written = "the registration result has been written into file: ";                             // main.cpp:89,154
const char *const all_failed_a = "registration all failed (", *const all_failed_b = " pairs)";                  // main.cpp:148
const char *const some_failed_a = "registration of ", *const some_failed_b = " (out of ", *const some_failed_c = ") pairs failed";  // main.cpp:152
}  // namespace text

using Matrix4 = Eigen::Matrix<float, 4, 4>;

struct Job {
    std::string target, source;
};

struct Outcome {
    bool ok = false;
    Matrix4 T;
    std::string console_out, console_err;
};

// One block of the result file (main.cpp:86-88 / 93 for a single pair, :134-143 in batch mode, where every block
// is followed by an empty line).
void write_block(std::ostream &os, const Job &job, const Outcome &r, bool batch) {
    if (batch || r.ok) os << text::target_tag << job.target << std::endl << text::source_tag << job.source << std::endl;
    if (r.ok) os << text::matrix_tag << r.T << std::endl;
    else os << text::failed_block << Matrix4::Identity() << std::endl;
    if (batch) os << std::endl;
}

int env_int(const char *name, int fallback) {
    const char *v = getenv(name);
    return std::max(1, v ? atoi(v) : fallback);
}

// The pair list (main.cpp:117-131): empty lines are skipped, a name that cannot be opened is reported and
// skipped, and what remains is taken two at a time (a name left over at the end is dropped).
bool read_jobs(const char *list_path, std::vector<Job> &jobs) {
    std::ifstream in(list_path);
    if (!in.is_open()) return false;
    std::string line, pending;
    bool have_pending = false;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        if (!std::ifstream(line).is_open()) {
            std::cerr << text::missing_file << line << std::endl;
            continue;
        }
        if (!have_pending) { pending = line; have_pending = true; }
        else { jobs.push_back(Job{pending, line}); have_pending = false; }
    }
    return true;
}

// Appends finished pairs to the result file in input order; pair i is written once pairs 0..i are all done.
class OrderedWriter {
public:
    OrderedWriter(std::ostream &file, const std::vector<Job> &jobs) : file_(file), jobs_(jobs), done_(jobs.size()), have_(jobs.size(), 0) {}
    void submit(size_t i, Outcome &&r) {
        std::lock_guard<std::mutex> lk(m_);
        done_[i] = std::move(r);
        have_[i] = 1;
        while (next_ < jobs_.size() && have_[next_]) {
            Outcome &o = done_[next_];
            std::cout << o.console_out << std::flush;
            std::cerr << o.console_err << std::flush;
            write_block(file_, jobs_[next_], o, true);
            file_.flush();
            (o.ok ? n_ok_ : n_failed_) += 1;
            o = Outcome();   // the text is not needed any more
            ++next_;
        }
    }
    int n_ok() const { return n_ok_; }
    int n_failed() const { return n_failed_; }

private:
    std::mutex m_;
    std::ostream &file_;
    const std::vector<Job> &jobs_;
    std::vector<Outcome> done_;
    std::vector<char> have_;
    size_t next_ = 0;
    int n_ok_ = 0, n_failed_ = 0;
};

Outcome run_job(const Job &job, bool capture_console) {
    Outcome r;
    std::ostringstream out, err;
    if (capture_console) plade_set_thread_console(&out, &err);
    try {
        r.ok = registration(r.T, job.target, job.source);
    } catch (const std::exception &e) {   // nothing a single malformed pair does may take the batch down
        err << "registration failed: " << e.what() << std::endl;
        r.ok = false;
    }
    if (capture_console) {
        plade_set_thread_console(nullptr, nullptr);
        r.console_out = out.str();
        r.console_err = err.str();
    }
    return r;
}

int single_pair(const char *target, const char *source, const char *result_path) {
    std::ofstream output(result_path);
    if (!output.is_open()) {
        std::cerr << text::cannot_open_result << result_path << std::endl;
        return EXIT_FAILURE;
    }
    const Job job{target, source};
    const Outcome r = run_job(job, false);
    write_block(output, job, r, false);
    if (!r.ok) return EXIT_FAILURE;
    std::cout << text::written << result_path << std::endl;
    return EXIT_SUCCESS;
}

int batch(const char *list_path, const char *result_path) {
    std::vector<Job> jobs;
    if (!std::ifstream(list_path).is_open()) {
        std::cerr << text::cannot_open_list << list_path << std::endl;
        return EXIT_FAILURE;
    }
    std::ofstream output(result_path);
    if (!output.is_open()) {
        std::cerr << text::cannot_open_result << result_path << std::endl;
        return EXIT_FAILURE;
    }
    read_jobs(list_path, jobs);
    plade_cli_trace("batch: list read");
    // Defaults by the length of the list: a worker's context and work areas cost ~0.3 s to set up (more for larger groups) and
    // the set-ups of one process run one after the other, so a short list is done sooner with two workers taking four pairs at
    // a time (64 pairs: 1.0 s against 1.4 s with four workers and 2.5 s with four workers x eight pairs), while a long one is
    // worth the full pipeline of the library's batch mode (bench.py: 4 groups of 8 pairs in flight per GPU).
    // PLADE_GPUS x PLADE_INFLIGHT workers, each taking PLADE_GROUP (1..8) consecutive pairs of the list per call (one GROUP:
    // the plane extraction of its clouds is one launch sequence, plade.h registration_group).  PLADE_GPUS defaults to EVERY GPU
    // the process sees: the pairs of a list are independent (main.cpp:122-148 is a plain loop) and `PLADE pairs.txt out.txt`
    // carries no other switch, so batch mode shards over the node's GPUs unasked (worker w runs on GPU w % PLADE_GPUS; the
    // result file is written in input order whatever finishes first).  PLADE_GPU_MAP ("0,0,1,1", a test hook for boxes with
    // fewer GPUs than PLADE_GPUS) maps worker-side device numbers to physical ones.
    const bool long_list = jobs.size() >= 512;
    const int n_gpus = env_int("PLADE_GPUS", std::max(1, plade_gpu_count())), per_gpu = env_int("PLADE_INFLIGHT", long_list ? 4 : 2);
    const size_t group = (size_t)std::min(env_int("PLADE_GROUP", long_list ? 8 : 4), (int)registration_group_max);
    std::vector<int> gpu_map(n_gpus);
    for (int g = 0; g < n_gpus; ++g) gpu_map[g] = g;
    if (const char *m = getenv("PLADE_GPU_MAP")) {
        std::stringstream ss(m);
        std::string tok;
        for (int g = 0; g < n_gpus && std::getline(ss, tok, ','); ++g) gpu_map[g] = atoi(tok.c_str());
    }
    plade_cli_trace("batch: GPUs counted");
    const size_t n_groups = (jobs.size() + group - 1) / group;
    const int n_workers = (int)std::min<size_t>((size_t)n_gpus * per_gpu, std::max<size_t>(n_groups, 1));
    OrderedWriter writer(output, jobs);
    std::mutex take;
    size_t next = 0;
    auto worker = [&](int gpu) {
        plade_select_device(gpu_map[gpu]);
        for (;;) {
            size_t i0;
            { std::lock_guard<std::mutex> lk(take); i0 = next; next += group; }
            if (i0 >= jobs.size()) break;
            const size_t k = std::min(group, jobs.size() - i0);
            if (k == 1) { writer.submit(i0, run_job(jobs[i0], true)); continue; }
            Outcome res[registration_group_max];
            std::ostringstream outs[registration_group_max], errs[registration_group_max];
            std::ostream *op[registration_group_max], *ep[registration_group_max];
            std::string tg[registration_group_max], sr[registration_group_max];
            Matrix4 T[registration_group_max];
            bool ok[registration_group_max];
            for (size_t q = 0; q < k; ++q) { op[q] = &outs[q]; ep[q] = &errs[q]; tg[q] = jobs[i0 + q].target; sr[q] = jobs[i0 + q].source; }
            try {
                registration_group(k, T, tg, sr, ok, op, ep);
            } catch (const std::exception &e) {   // nothing a malformed pair does may take the batch down
                for (size_t q = 0; q < k; ++q) { errs[q] << "registration failed: " << e.what() << std::endl; ok[q] = false; }
            }
            for (size_t q = 0; q < k; ++q) {
                res[q].ok = ok[q]; res[q].T = T[q]; res[q].console_out = outs[q].str(); res[q].console_err = errs[q].str();
                writer.submit(i0 + q, std::move(res[q]));
            }
        }
        plade_release_thread_context();   // the worker's plade_ctx (work areas in HBM) goes with the thread
    };
    if (n_workers <= 1 && group <= 1) worker(0);
    else {
        // several pairs in flight: the workers poll + sleep instead of spinning on the GPU (plade_params.host_wait),
        // so that the workers of all GPUs fit the host's CPUs
        setenv("PLADE_HOST_WAIT", "sleep", 0);
        std::vector<std::thread> pool;
        for (int w = 0; w < n_workers; ++w) pool.emplace_back(worker, w % n_gpus);
        for (std::thread &t : pool) t.join();
    }
    plade_cli_trace("batch: workers done");
    const int ok = writer.n_ok(), failed = writer.n_failed();
    if (ok == 0) {
        std::cerr << text::all_failed_a << failed << text::all_failed_b << std::endl;
        return EXIT_FAILURE;
    }
    if (failed > 0) std::cerr << text::some_failed_a << failed << text::some_failed_b << failed + ok << text::some_failed_c << std::endl;
    std::cout << text::written << result_path << std::endl;
    return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char **argv) {
    // The HIP runtime takes ~0.2 s to start (driver, device enumeration, code objects): a helper thread pays that while the main
    // thread parses the list and the first PLY files load
    plade_cli_trace("main");
    std::thread warm;
    if (argc == 3 || argc == 4) warm = std::thread([]() { (void)plade_gpu_count(); });
    int rc;
    switch (argc) {
        case 4: rc = single_pair(argv[1], argv[2], argv[3]); break;
        case 3: rc = batch(argv[1], argv[2]); break;
        default: std::cerr << text::usage; rc = EXIT_FAILURE;
    }
    if (warm.joinable()) warm.join();      // (no thread may still be inside the runtime's start-up when the process exits)
    plade_cli_trace("main: done");
    // Everything the user asked for is on disk and on the console: leave without tearing the HIP runtime down and handing
    // gigabytes of work areas back one allocation at a time (~0.1 s of a 0.4 s single-pair process; the OS reclaims them at once)
    std::cout.flush(); std::cerr.flush(); fflush(nullptr);
    if (getenv("PLADE_CLI_FULL_EXIT")) return rc;     // (profilers and PLADE_DEBUG_ALLOC report from the normal exit path)
    _exit(rc);
}
