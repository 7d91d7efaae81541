// plade_amd/csrc/main.cpp -- the PLADE command line (code/PLADE/main.cpp:30-159): same positional
// arguments, same result-file grammar, same exit codes.
//   PLADE target.ply source.ply result.txt      register one pair
//   PLADE file_pairs.txt result.txt             batch mode
// Batch mode additionally shards the pairs over the GPUs of the node when PLADE_GPUS=N is set, with
// PLADE_INFLIGHT=M (default 4) worker threads per GPU, each with its own plade_ctx: pairs are independent,
// one registration alone is latency-bound, and PLY parsing of one pair overlaps the GPU work of the others.
// Results are written in input order.
#include "plade.h"

#include <atomic>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>

static void usage() {
    std::cerr << "PLADE can register two point clouds dominated by planar structures. It can be used in two ways.\n"
              << "-------------------------------------------------------------------------------------------------\n"
              << "Usage 1: register a 'source' point cloud to a 'target' point cloud.\n"
              << "    ---------------------------------------------------------------------------------------------\n"
              << "    You can call PLADE with three arguments. The first two are the file names of a target point\n"
              << "    cloud and a source point cloud (the target point cloud file name always comes first). The\n"
              << "    third argument specifies the result file name. Below is an example:\n"
              << "         ./PLADE  room_target.ply  room_source.ply  result.txt\n"
              << "    The target point cloud file name always comes first, and both point cloud files must be in\n"
              << "    the 'ply' format. The result file will store the registration result, which is a 4 by 4\n"
              << "    transformation matrix that aligns the source point cloud to the target point cloud.\n"
              << "-------------------------------------------------------------------------------------------------\n"
              << "Usage 2: register a bunch of point cloud pairs.\n"
              << "    ---------------------------------------------------------------------------------------------\n"
              << "    You can call PLADE with two arguments: a file (e.g., file_pairs.txt) specifying all pairs\n"
              << "    of target/source point cloud files and a result file. Below is an example:\n"
              << "         ./PLADE  file_pairs.txt  result.txt\n"
              << "    In 'file_pairs.txt', every two consecutive lines store two file names. The first line is the\n"
              << "    file name of a target point cloud, and the second line is the file name of a source point cloud.\n"
              << "    Both point cloud files must be in the 'ply' format. The result file will store the registration\n"
              << "    results, a set of 4 by 4 transformation matrices. Each matrix aligns a source point cloud to\n"
              << "    its corresponding target point cloud.\n";
}

int main(int argc, char **argv) {
    if (argc != 3 && argc != 4) {
        usage();
        return EXIT_FAILURE;
    }
    if (argc == 4) {
        std::ofstream output(argv[3]);
        if (!output.is_open()) {
            std::cerr << "failed opening the result file: " << argv[3] << std::endl;
            return EXIT_FAILURE;
        }
        Eigen::Matrix<float, 4, 4> transformation;
        if (registration(transformation, argv[1], argv[2])) {
            output << "target: " << argv[1] << std::endl;
            output << "source: " << argv[2] << std::endl;
            output << "transformation:\n" << transformation << std::endl;
            std::cout << "the registration result has been written into file: " << argv[3] << std::endl;
            return EXIT_SUCCESS;
        } else {
            output << "registration failed, an identity matrix is recorded:\n" << Eigen::Matrix<float, 4, 4>::Identity() << std::endl;
            return EXIT_FAILURE;
        }
    }
    // batch mode
    std::ifstream input(argv[1]);
    if (!input.is_open()) {
        std::cerr << "failed opening the file containing pairs of point cloud names: " << argv[1] << std::endl;
        return EXIT_FAILURE;
    }
    std::ofstream output(argv[2]);
    if (!output.is_open()) {
        std::cerr << "failed opening the result file: " << argv[2] << std::endl;
        return EXIT_FAILURE;
    }
    auto is_file = [](const std::string &filename) -> bool {
        std::ifstream fin(filename);
        return fin.is_open();
    };
    std::vector<std::pair<std::string, std::string>> pairs;
    while (!input.eof()) {
        std::vector<std::string> file_pair;
        while (!input.eof() && file_pair.size() < 2) {
            std::string file_name;
            getline(input, file_name);
            if (!file_name.empty()) {
                if (is_file(file_name)) file_pair.push_back(file_name);
                else std::cerr << "file doesn't exist: " << file_name << std::endl;
            }
        }
        if (file_pair.size() == 2) pairs.push_back(std::make_pair(file_pair[0], file_pair[1]));
    }
    const char *env = getenv("PLADE_GPUS");
    const int n_gpus = std::max(1, env ? atoi(env) : 1);
    const char *env_m = getenv("PLADE_INFLIGHT");
    const int per_gpu = std::max(1, env_m ? atoi(env_m) : 4);
    const int n_workers = (int)std::min<size_t>((size_t)n_gpus * per_gpu, std::max<size_t>(pairs.size(), 1));
    std::vector<Eigen::Matrix<float, 4, 4>> results(pairs.size());
    std::vector<char> status(pairs.size(), 0);
    std::atomic<size_t> next(0);
    auto worker = [&](int gpu) {
        plade_select_device(gpu);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= pairs.size()) break;
            status[i] = registration(results[i], pairs[i].first, pairs[i].second) ? 1 : 0;
        }
    };
    if (n_workers == 1) worker(0);
    else {
        // several pairs in flight: the worker threads poll + sleep instead of spinning on the GPU (plade_params.host_wait),
        // so that the workers of all GPUs fit the host's CPUs
        setenv("PLADE_HOST_WAIT", "sleep", 0);
        std::vector<std::thread> th;
        for (int w = 0; w < n_workers; ++w) th.emplace_back(worker, w % n_gpus);
        for (auto &t : th) t.join();
    }
    int count_success = 0, count_failure = 0;
    for (size_t i = 0; i < pairs.size(); ++i) {
        output << "target: " << pairs[i].first << std::endl;
        output << "source: " << pairs[i].second << std::endl;
        if (status[i]) {
            output << "transformation:\n" << results[i] << std::endl << std::endl;
            ++count_success;
        } else {
            output << "registration failed, an identity matrix is recorded:\n" << Eigen::Matrix<float, 4, 4>::Identity() << std::endl << std::endl;
            ++count_failure;
        }
    }
    if (count_success == 0) {
        std::cerr << "registration all failed (" << count_failure << " pairs)" << std::endl;
        return EXIT_FAILURE;
    }
    if (count_failure > 0)
        std::cerr << "registration of " << count_failure << " (out of " << count_failure + count_success << ") pairs failed" << std::endl;
    std::cout << "the registration result has been written into file: " << argv[2] << std::endl;
    return EXIT_SUCCESS;
}
