// host/mainimage_demo.cpp — the reference-style per-pixel host loop, unmodified in shape, running over
// the drop-in header: -DAPP_<NAME> selects the app, the loop calls mainImage(fragColor, fragCoord).
#include "../include/sbx_mainimage.hpp"

#include <cstdio>
#include <cstdlib>

struct vec2 { float x, y; float operator[](int i) const { return i ? y : x; } };
struct vec4 { float x, y, z, w; float& operator[](int i) { return (&x)[i]; } };

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 144;
    iResolution[0] = (float)W; iResolution[1] = (float)H;
    iGlobalTime = argc > 3 ? (float)atof(argv[3]) : 0.37f;
    double sum[3] = {0, 0, 0};
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            vec4 c;
            mainImage(c, vec2{x + .5f, y + .5f});
            sum[0] += c.x; sum[1] += c.y; sum[2] += c.z;
        }
    printf("%dx%d t=%g mean rgb = %.6f %.6f %.6f\n", W, H, iGlobalTime, sum[0] / (W * H), sum[1] / (W * H), sum[2] / (W * H));
    return 0;
}
