// `enhance` CLI -- same argv, stdout and exit codes as the reference's src/enhance.cpp:12-52:
//   enhance <image> <output> <# row samples> <# col samples> <hx> <hy> <# sinkhorn iterations>
//           <# eigen vectors> <weight 1> [<weight 2> ...]
// Differences: runs headless (no imshow / waitKey, src/enhance.cpp:48-49), reads BMP/PPM and writes
// BMP/PPM/PNG through nle/image_io.hpp instead of OpenCV.  The hot path runs on the GPU through
// libnle_hip.so.
#include "cli_common.hpp"

int main(int argc, char* argv[]) {
    nlecli::FilterArgs a;
    if (!nlecli::parse(argc, argv, 10, &a)) return 0;  // usage: src/enhance.cpp:15-18 (exit code 0 on purpose)
    const nle::Image image = nlecli::load(a);
    if (image.empty()) return 0;                        // src/enhance.cpp:34-37
    nle::NLEFilter filter;
    filter.trainForEnhancement(image, a.rowSamples, a.colSamples, a.hx, a.hy, a.sinkhornIters, a.eigenVectors);
    const nle::Image result = filter.enhance(image, a.extra);  // the weights are argv[9..]
    nlecli::report(filter);
    return nlecli::finish(a, result, "Done. Press any key in result window to exit.");  // src/enhance.cpp:45
}
