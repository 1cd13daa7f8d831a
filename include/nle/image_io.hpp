// nle/image_io.hpp -- minimal stand-ins for cv::imread / cv::imwrite (reference src/enhance.cpp:33,47).
// Read: 24/32-bit uncompressed BMP, binary PPM (P6), PNG (non-interlaced; own inflate), JPEG (baseline and progressive
// Huffman, 8 bit, grey / YCbCr up to 2 x 2 subsampling; pixels bit-identical to libjpeg's default decode -- host/jpeg.cpp).
// Write: .bmp, .ppm, .png (zlib "stored" blocks), .jpg / .jpeg (baseline 4:2:0, quality 95).  No external library.  Images are 8UC3 in BGR order like cv::imread
// returns them.
#pragma once
#include <string>

#include "nle/filter.hpp"

namespace nle {
Image imread(const std::string& path);                  // empty Image on failure (like cv::imread)
bool imwrite(const std::string& path, const Image& bgr);  // false on failure
}  // namespace nle
