// Shared shell of the two command-line tools (enhance, denoise): both take
//   <image> <output> <# row samples> <# col samples> <hx> <hy> <# sinkhorn iterations> <# eigen vectors> ...
// print the same usage line when too few arguments are given, parse numbers with std::stoi / std::stod (garbage
// throws, like the reference: src/enhance.cpp:20-31, src/denoise.cpp:19-31) and return 0 on the two soft failures
// (usage, unreadable image) for drop-in compatibility.
#pragma once

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "nle/filter.hpp"
#include "nle/image_io.hpp"

namespace nlecli {

struct FilterArgs {
    std::string input, output;
    int rowSamples = 0, colSamples = 0, sinkhornIters = 0, eigenVectors = 0;
    double hx = 0, hy = 0;
    std::vector<double> extra;  // everything after the eighth argument, as doubles
};

// false (after printing the usage line to stderr) when fewer than `min_argc` arguments were given
inline bool parse(int argc, char* argv[], int min_argc, FilterArgs* a) {
    if (argc < min_argc) {
        std::cerr << "Usage: " << argv[0]
                  << " <image> <output> <# row samples> <# col samples> <hx> <hy> <# sinkhorn iterations> <# eigen "
                     "vectors> <weight 1> <weight 2> <weight 3> <weight 4>"
                  << std::endl;
        return false;
    }
    a->input = argv[1];
    a->output = argv[2];
    int* ints[] = {&a->rowSamples, &a->colSamples, nullptr, nullptr, &a->sinkhornIters, &a->eigenVectors};
    double* reals[] = {nullptr, nullptr, &a->hx, &a->hy, nullptr, nullptr};
    for (int i = 0; i < 6; ++i) {
        if (ints[i]) *ints[i] = std::stoi(argv[3 + i]);
        else *reals[i] = std::stod(argv[3 + i]);
    }
    for (int i = 9; i < argc; ++i) a->extra.push_back(std::stod(argv[i]));
    return true;
}

// reads the input (empty image + message on stderr if that fails)
inline nle::Image load(const FilterArgs& a) {
    nle::Image image = nle::imread(a.input);
    if (image.empty()) std::cerr << "Failed to read file from " << a.input << std::endl;
    return image;
}

// NLE_REPORT=<path> (an environment variable, so that reference-style command lines stay valid): one JSON object
// with what the train decided (nle_filter_diag), the kept eigenvalues and the per-stage milliseconds
inline void report(const nle::NLEFilter& filter) {
    const char* path = std::getenv("NLE_REPORT");
    if (!path || !*path) return;
    int d[8];
    double ms[6] = {0, 0, 0, 0, 0, 0};
    filter.diag(d);
    filter.timings(ms);
    const nle::Vec ev = filter.eigvals();
    std::ofstream os(path);
    os.precision(17);
    os << "{\"formulation\": " << d[0] << ", \"p\": " << d[1] << ", \"r_Ka\": " << d[2] << ", \"r_Wa\": " << d[3]
       << ", \"r_Q\": " << d[4] << ", \"K\": " << d[5] << ", \"chol_Ka\": " << d[6] << ", \"chol_Wa\": " << d[7]
       << ", \"eigvals\": [";
    for (int i = 0; i < ev.size(); ++i) os << (i ? ", " : "") << ev(i);
    os << "], \"ms\": {\"samples\": " << ms[0] << ", \"sinkhorn\": " << ms[1] << ", \"gram\": " << ms[2]
       << ", \"project\": " << ms[3] << ", \"host\": " << ms[4] << ", \"train_total\": " << ms[5] << "}}" << std::endl;
}

// `banner`: the line the reference prints before it writes the file (src/enhance.cpp:45, src/denoise.cpp:45); the
// window and the key press it announces do not exist here (headless)
inline int finish(const FilterArgs& a, const nle::Image& result, const char* banner) {
    std::cout << banner << std::endl;
    if (nle::imwrite(a.output, result)) return 0;
    std::cerr << "Failed to write " << a.output << std::endl;
    return 1;
}

}  // namespace nlecli
