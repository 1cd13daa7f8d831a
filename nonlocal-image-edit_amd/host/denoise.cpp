// `denoise` CLI -- same argv, stdout and exit codes as the reference's src/denoise.cpp:12-55:
//   denoise <image> <output> <# row samples> <# col samples> <hx> <hy> <# sinkhorn iterations> <# eigen vectors>
//           <sigma color> <sigma space> <shrink factor>
// (the reference's usage line still lists four weights -- it was copied from enhance -- and is kept verbatim;
// sigma color / sigma space are parsed as doubles and truncated to int by the NLEFilter signatures, :28-29,44-46).
// Differences: headless (no imshow / waitKey, :50-51), BMP/PPM in and BMP/PPM/PNG out through nle/image_io.hpp.
#include "cli_common.hpp"

int main(int argc, char* argv[]) {
    nlecli::FilterArgs a;
    if (!nlecli::parse(argc, argv, 12, &a)) return 0;  // usage: src/denoise.cpp:14-17 (exit code 0 on purpose)
    const int sigmaColor = (int)a.extra[0], sigmaSpace = (int)a.extra[1];
    const double shrinkFactor = a.extra[2];
    const nle::Image image = nlecli::load(a);
    if (image.empty()) return 0;                        // src/denoise.cpp:33-36
    nle::NLEFilter filter;
    filter.trainForDenoise(image, a.rowSamples, a.colSamples, a.hx, a.hy, a.sinkhornIters, a.eigenVectors, sigmaColor,
                           sigmaSpace);
    const nle::Image result = filter.denoise(image, shrinkFactor, sigmaColor, sigmaSpace);
    nlecli::report(filter);
    return nlecli::finish(a, result, "Done. Press any key in result window to exit.");  // src/denoise.cpp:45
}
