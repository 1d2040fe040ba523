// JPEG reader of the host mirror (Image::readJpeg, image/image.cpp:827-879): see jpeg_decode.cpp.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

namespace mvshost {

// pixels: row-major, `channels` (1 or 3: grey or RGB) bytes per pixel.  0 on success; -1 with *error set otherwise.
int decodeJpeg(const unsigned char* data, size_t size, std::vector<unsigned char>& pixels, int& width, int& height, int& channels,
               std::string* error = nullptr);
int readJpegFile(const std::string& file, std::vector<unsigned char>& pixels, int& width, int& height, int& channels,
                 std::string* error = nullptr);

}  // namespace mvshost
