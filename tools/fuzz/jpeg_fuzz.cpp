#include "../../mvskit_amd/host/jpeg_decode.hpp"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <vector>
#include <string>
int main(int argc, char** argv) {
    std::mt19937 rng(1234);
    long ok = 0, bad = 0;
    for (int a = 1; a < argc; ++a) {
        std::ifstream is(argv[a], std::ios::binary);
        std::vector<unsigned char> base((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
        for (int it = 0; it < 3000; ++it) {
            std::vector<unsigned char> d = base;
            const int nmut = 1 + rng() % 6;
            for (int m = 0; m < nmut; ++m) {
                const int kind = rng() % 4;
                const size_t pos = rng() % d.size();
                if (kind == 0) d[pos] = (unsigned char)rng();
                else if (kind == 1) d[pos] ^= 1u << (rng() % 8);
                else if (kind == 2 && d.size() > 8) d.resize(pos + 1);
                else if (kind == 3) d.insert(d.begin() + pos, (unsigned char)rng());
            }
            std::vector<unsigned char> px; int w, h, c; std::string e;
            if (mvshost::decodeJpeg(d.data(), d.size(), px, w, h, c, &e) == 0) { ++ok; if ((size_t)w * h * c != px.size()) { printf("size mismatch\n"); return 1; } }
            else ++bad;
        }
    }
    printf("decoded %ld, rejected %ld\n", ok, bad);
    return 0;
}
