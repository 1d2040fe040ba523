#include "../../mvskit_amd/host/ply_read.hpp"
#include <cstdio>
#include <fstream>
#include <random>
#include <vector>
#include <string>
int main(int argc, char** argv) {
    std::mt19937 rng(99);
    long ok = 0, bad = 0;
    for (int a = 1; a < argc; ++a) {
        std::ifstream is(argv[a], std::ios::binary);
        std::vector<unsigned char> base((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
        for (int it = 0; it < 3000; ++it) {
            std::vector<unsigned char> d = base;
            const int nmut = 1 + rng() % 5;
            for (int m = 0; m < nmut; ++m) {
                const int kind = rng() % 4;
                const size_t pos = rng() % std::min<size_t>(d.size(), it % 2 ? d.size() : 300);
                if (kind == 0) d[pos] = (unsigned char)rng();
                else if (kind == 1) d[pos] ^= 1u << (rng() % 8);
                else if (kind == 2 && d.size() > 8) d.resize(pos + 1);
                else if (kind == 3) d.insert(d.begin() + pos, (unsigned char)('0' + rng() % 10));
            }
            { std::ofstream o("/tmp/pfuzz.ply", std::ios::binary); o.write((const char*)d.data(), d.size()); }
            std::vector<double> p, n; std::string e;
            if (mvshost::readPlyVertices("/tmp/pfuzz.ply", p, &n, &e) == 0) ++ok; else ++bad;
        }
    }
    printf("read %ld, rejected %ld\n", ok, bad);
    return 0;
}
