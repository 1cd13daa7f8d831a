// `denoise` CLI -- same argv, stdout and exit codes as the reference's src/denoise.cpp:12-55:
//   denoise <image> <output> <# row samples> <# col samples> <hx> <hy> <# sinkhorn iterations> <# eigen vectors>
//           <sigma color> <sigma space> <shrink factor>
// (the reference's usage line still lists four weights -- it was copied from enhance -- and is kept verbatim;
// sigma color / sigma space are parsed as doubles and truncated to int by the NLEFilter signatures, :28-29,44-46).
// Differences: headless (no imshow / waitKey, :50-51), BMP/PPM in and BMP/PPM/PNG out through nle/image_io.hpp.
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "nle/filter.hpp"
#include "nle/image_io.hpp"

int main(int argc, char* argv[]) {
    if (argc < 12) {  // src/denoise.cpp:14-17 (exit code 0 on purpose)
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
    double sigmaColor = std::stod(argv[9]);
    double sigmaSpace = std::stod(argv[10]);
    double shrinkFactor = std::stod(argv[11]);

    nle::Image image = nle::imread(imagePath);
    if (image.empty()) {  // src/denoise.cpp:33-36
        std::cerr << "Failed to read file from " << imagePath << std::endl;
        return 0;
    }

    nle::NLEFilter filter;
    filter.trainForDenoise(image, nRowSamples, nColSamples, hx, hy, nSinkhornIter, nEigenVectors, (int)sigmaColor,
                           (int)sigmaSpace);
    nle::Image result = filter.denoise(image, shrinkFactor, (int)sigmaColor, (int)sigmaSpace);
    std::cout << "Done." << std::endl;
    if (!nle::imwrite(outputPath, result)) {
        std::cerr << "Failed to write " << outputPath << std::endl;
        return 1;
    }
    return 0;
}
