// Shared shell of the two command-line tools (enhance, denoise): both take
//   <image> <output> <# row samples> <# col samples> <hx> <hy> <# sinkhorn iterations> <# eigen vectors> ...
// print the same usage line when too few arguments are given, parse numbers with std::stoi / std::stod (garbage
// throws, like the reference: src/enhance.cpp:20-31, src/denoise.cpp:19-31) and return 0 on the two soft failures
// (usage, unreadable image) for drop-in compatibility.
#pragma once

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

inline int finish(const FilterArgs& a, const nle::Image& result) {
    std::cout << "Done." << std::endl;
    if (nle::imwrite(a.output, result)) return 0;
    std::cerr << "Failed to write " << a.output << std::endl;
    return 1;
}

}  // namespace nlecli
