// `enhance` CLI -- same argv, stdout and exit codes as the reference's src/enhance.cpp:12-52:
//   enhance <image> <output> <# row samples> <# col samples> <hx> <hy> <# sinkhorn iterations>
//           <# eigen vectors> <weight 1> [<weight 2> ...]
// Differences: runs headless (no imshow / waitKey, src/enhance.cpp:48-49), reads BMP/PPM and writes
// BMP/PPM/PNG through nle/image_io.hpp instead of OpenCV.  The hot path runs on the GPU through
// libnle_hip.so.
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "nle/filter.hpp"
#include "nle/image_io.hpp"

int main(int argc, char* argv[]) {
    if (argc < 10) {  // src/enhance.cpp:15-18 (exit code 0 on purpose)
        std::cerr << "Usage: " << argv[0]
                  << " <image> <output> <# row samples> <# col samples> <hx> <hy> <# sinkhorn iterations> <# eigen "
                     "vectors> <weight 1> <weight 2> <weight 3> <weight 4>"
                  << std::endl;
        return 0;
    }
    std::string imagePath{argv[1]};
    std::string outputPath{argv[2]};
    int nRowSamples = std::stoi(argv[3]);
    int nColSamples = std::stoi(argv[4]);
    double hx = std::stod(argv[5]);
    double hy = std::stod(argv[6]);
    int nSinkhornIter = std::stoi(argv[7]);
    int nEigenVectors = std::stoi(argv[8]);
    std::vector<nle::DType> weights;
    for (auto i = 9; i < argc; ++i) weights.push_back(std::stod(argv[i]));

    nle::Image image = nle::imread(imagePath);
    if (image.empty()) {  // src/enhance.cpp:34-37
        std::cerr << "Failed to read file from " << imagePath << std::endl;
        return 0;
    }

    nle::NLEFilter filter;
    filter.trainForEnhancement(image, nRowSamples, nColSamples, hx, hy, nSinkhornIter, nEigenVectors);
    nle::Image result = filter.enhance(image, weights);
    std::cout << "Done." << std::endl;
    if (!nle::imwrite(outputPath, result)) {
        std::cerr << "Failed to write " << outputPath << std::endl;
        return 1;
    }
    return 0;
}
