// host/sbx_ddsvolgen.cpp — the role of the reference's util/ddsvolgen: bake the tiled Worley-fBm noise volume
// and write it as a DX10 volume .dds (RGBA32F, R = noise, G = B = A = 0).  The voxels come from the HIP kernel
// behind sbx_worley_volume() (the reference computes them with 4 CPU threads,
// /root/reference/util/ddsvolgen/src/ddsvolgen.cpp:101-131); the file layout follows the public DDS format
// (magic, 124-byte DDS_HEADER, 20-byte DDS_HEADER_DXT10, raw texels) with the field values the reference sets
// (ddsvolgen.cpp:72-92).
//
//   sbx_ddsvolgen [--size 128] [--out noise3d.dds]
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/sbx.h"

#pragma pack(push, 1)
struct DdsPixelFormat { uint32_t size, flags, fourCC, rgbBitCount, rMask, gMask, bMask, aMask; };
struct DdsHeader {
    uint32_t size, flags, height, width, pitchOrLinearSize, depth, mipMapCount, reserved1[11];
    DdsPixelFormat ddspf;
    uint32_t caps, caps2, caps3, caps4, reserved2;
};
struct DdsHeaderDx10 { uint32_t dxgiFormat, resourceDimension, miscFlag, arraySize, miscFlags2; };
#pragma pack(pop)
static_assert(sizeof(DdsHeader) == 124 && sizeof(DdsHeaderDx10) == 20, "DDS header layout");

int main(int argc, char** argv) {
    int size = 128;
    std::string out = "noise3d.dds";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--size" && i + 1 < argc) size = atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else { fprintf(stderr, "usage: sbx_ddsvolgen [--size N] [--out file.dds]\n"); return 2; }
    }
    sbx_ctx* ctx = nullptr;
    int rc = sbx_create(0, &ctx);
    if (rc != SBX_OK) { fprintf(stderr, "sbx_create failed (%d): a gfx950 GPU is required, there is no CPU path\n", rc); return 1; }
    const size_t texels = (size_t)size * size * size;
    float* dev = nullptr;
    if (hipMalloc((void**)&dev, texels * 16) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    rc = sbx_worley_volume(ctx, size, dev, nullptr);
    if (rc != SBX_OK) { fprintf(stderr, "sbx_worley_volume: %s\n", sbx_last_error(ctx)); return 1; }
    std::vector<float> host(texels * 4);
    if (hipMemcpy(host.data(), dev, texels * 16, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "hipMemcpy failed\n"); return 1; }

    const uint32_t magic = 0x20534444u;                       // "DDS "
    DdsHeader h;
    std::memset(&h, 0, sizeof(h));
    h.size = sizeof(DdsHeader);
    h.flags = 0x1007u | 0x800000u | 0x8u;                     // TEXTURE | VOLUME (DDSD_DEPTH) | PITCH
    h.height = h.width = h.depth = (uint32_t)size;
    h.pitchOrLinearSize = ((uint32_t)size * 16u + 7u) / 8u;   // as the reference computes it
    h.mipMapCount = 0;
    h.ddspf.size = sizeof(DdsPixelFormat);
    h.ddspf.flags = 0x4u;                                     // DDPF_FOURCC
    h.ddspf.fourCC = 0x30315844u;                             // "DX10"
    h.caps = 0x1000u | 0x8u;                                  // TEXTURE | (the reference also sets the COMPLEX bit 0x8)
    h.caps2 = 0x200000u;                                      // DDSCAPS2_VOLUME
    DdsHeaderDx10 x;
    x.dxgiFormat = 2;                                         // DXGI_FORMAT_R32G32B32A32_FLOAT
    x.resourceDimension = 4;                                  // D3D10_RESOURCE_DIMENSION_TEXTURE3D
    x.miscFlag = 0; x.arraySize = 1; x.miscFlags2 = 0;
    FILE* fp = fopen(out.c_str(), "wb");
    if (!fp) { perror("fopen"); return 1; }
    fwrite(&magic, 4, 1, fp); fwrite(&h, sizeof(h), 1, fp); fwrite(&x, sizeof(x), 1, fp);
    fwrite(host.data(), 16, texels, fp);
    fclose(fp);
    printf("wrote %s: %d^3 RGBA32F, voxel(0,0,0).r = %.9g\n", out.c_str(), size, host[0]);
    (void)hipFree(dev);
    sbx_destroy(ctx);
    return 0;
}
