// A host program that uses the field query through its C ABI only -- no Python, no torch: what a C/C++ (or cgo /
// JNI / N-API) caller of include/d3fields_hip.h looks like.  Builds with one command (see tests/test_gpu_parity.py,
// test_c_abi_from_cpp_host):
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_demo.cpp -L d3fields_amd -ld3fields_hip \
//         -Wl,-rpath,$PWD/d3fields_amd -o /tmp/c_abi_demo && /tmp/c_abi_demo /tmp/c_abi_demo.bin
//
// It makes a small synthetic observation (two views of the plane z = 0, an 8-channel feature map), queries 5000
// points with d3f_eval (with the optional scratch, i.e. the Morton walk forced on), reads the result back and
// writes inputs + outputs to the given file so that a checker can recompute them independently.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "d3fields_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define D3F_CHECK(x) do { int rc_ = (x); if (rc_ != D3F_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, d3f_last_error()); return 3; } } while (0)

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }
static float unif(uint32_t &s) { return (float)(lcg(s) >> 8) / 16777216.0f; }

template <typename T> static T *to_device(const std::vector<T> &h)
{
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char **argv)
{
    const int V = 2, H = 48, W = 64, fh = 6, fw = 8, C = 8;
    const int64_t n = 5000;
    if (d3f_abi_version() != D3F_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    // two cameras 1 m above the plane z = 0 (world z points away from them), looking straight at it, 0.3 m apart
    std::vector<float> K(V * 9, 0.0f), pose(V * 12, 0.0f), depth((size_t)V * H * W), feats((size_t)V * fh * fw * C), pts(n * 3);
    for (int v = 0; v < V; ++v) {
        float *k = &K[v * 9], *p = &pose[v * 12];
        k[0] = k[4] = 60.0f; k[2] = W / 2.0f; k[5] = H / 2.0f; k[8] = 1.0f;
        p[0] = p[5] = p[10] = 1.0f;                 // R = I
        p[3] = v == 0 ? 0.15f : -0.15f; p[11] = 1.0f;   // t: camera at x = -/+0.15, z = -1
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) depth[((size_t)v * H + y) * W + x] = ((x + y) % 11 == 0) ? 0.0f : 1.0f;   // holes
    }
    uint32_t seed = 7;
    for (auto &f : feats) f = unif(seed) * 2.0f - 1.0f;
    for (int64_t i = 0; i < n; ++i) {
        pts[i * 3 + 0] = unif(seed) * 1.2f - 0.6f;
        pts[i * 3 + 1] = unif(seed) * 0.9f - 0.45f;
        pts[i * 3 + 2] = unif(seed) * 0.08f - 0.05f;     // around the surface: in front, inside and beyond the truncation band
    }

    float *dK = to_device(K), *dpose = to_device(pose), *ddepth = to_device(depth), *dfeats = to_device(feats), *dpts = to_device(pts);
    float *ddist = nullptr, *dfused = nullptr;
    uint8_t *dvalid = nullptr;
    void *dws = nullptr;
    if (!dK || !dpose || !ddepth || !dfeats || !dpts) { fprintf(stderr, "device allocation failed\n"); return 2; }
    HIP_OK(hipMalloc(&ddist, n * sizeof(float)));
    HIP_OK(hipMalloc(&dvalid, n));
    HIP_OK(hipMalloc(&dfused, n * C * sizeof(float)));
    const int64_t ws_bytes = d3f_eval_workspace_bytes(n);
    HIP_OK(hipMalloc(&dws, (size_t)ws_bytes));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    d3f_views views;
    memset(&views, 0, sizeof(views));
    views.V = V; views.H = H; views.W = W; views.depth = ddepth; views.K = dK; views.pose = dpose;
    d3f_channel_map map;
    memset(&map, 0, sizeof(map));
    map.data = dfeats; map.fh = fh; map.fw = fw; map.C = C; map.dtype = D3F_DTYPE_F32;
    map.stride_v = (int64_t)fh * fw * C; map.stride_y = (int64_t)fw * C; map.stride_x = C;
    float *outs[1] = {dfused};
    D3F_CHECK(d3f_eval(&views, dpts, n, &map, 1, 0.02f, D3F_TUNE_FORCE_REORDER, ddist, dvalid, outs, nullptr, dws, ws_bytes, stream));
    HIP_OK(hipStreamSynchronize(stream));

    std::vector<float> dist(n), fused(n * C);
    std::vector<uint8_t> valid(n);
    HIP_OK(hipMemcpy(dist.data(), ddist, n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(valid.data(), dvalid, n, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(fused.data(), dfused, n * C * sizeof(float), hipMemcpyDeviceToHost));
    int64_t nvalid = 0;
    double sum = 0.0;
    for (int64_t i = 0; i < n; ++i) { nvalid += valid[i]; if (valid[i]) sum += dist[i]; }
    printf("%s: %lld points, %lld valid, mean dist of valid %.6f\n", d3f_version(), (long long)n, (long long)nvalid, nvalid ? sum / nvalid : 0.0);

    // an error path: the status code comes back, nothing aborts
    if (d3f_eval(&views, dpts, n, &map, 1, -1.0f, 0, ddist, dvalid, outs, nullptr, nullptr, 0, stream) != D3F_ERR_INVALID_ARG) return 4;

    if (argc > 1) {
        FILE *f = fopen(argv[1], "wb");
        if (!f) return 5;
        const int32_t hdr[8] = {V, H, W, fh, fw, C, (int32_t)n, 0};
        fwrite(hdr, sizeof(hdr), 1, f);
        fwrite(K.data(), sizeof(float), K.size(), f);
        fwrite(pose.data(), sizeof(float), pose.size(), f);
        fwrite(depth.data(), sizeof(float), depth.size(), f);
        fwrite(feats.data(), sizeof(float), feats.size(), f);
        fwrite(pts.data(), sizeof(float), pts.size(), f);
        fwrite(dist.data(), sizeof(float), dist.size(), f);
        fwrite(valid.data(), 1, valid.size(), f);
        fwrite(fused.data(), sizeof(float), fused.size(), f);
        fclose(f);
    }
    void *bufs[] = {dK, dpose, ddepth, dfeats, dpts, ddist, dvalid, dfused, dws};
    for (void *b : bufs) (void)hipFree(b);
    (void)hipStreamDestroy(stream);
    return 0;
}
