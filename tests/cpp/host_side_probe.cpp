// Test helper (CPU only, no HIP): exercises the host-side headers of the drop-in layer — Configuration parsing
// (include/kfusion/default_parameters.h) and the depth readers (include/kfusion/interface.h) — and prints what they
// produced as JSON, for tests/test_host_headers.py to compare with the python twins.
#include <cstdio>
#include <cstring>
#include <kfusion/default_parameters.h>
#include <kfusion/interface.h>

static unsigned long long checksum(const unsigned short *d, size_t n) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= d[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    if (argc >= 3 && !strcmp(argv[1], "--read")) {
        DepthReader *r = is_file(argv[2]) ? static_cast<DepthReader *>(new RawDepthReader(argv[2], 0, false))
                                          : static_cast<DepthReader *>(new SceneDepthReader(argv[2], 0, false));
        const uint2 s = r->getinputSize();
        const float4 k = r->getK();
        std::vector<unsigned short> buf((size_t)s.x * s.y);
        std::vector<unsigned long long> sums;
        while (r->readNextDepthFrame(buf.data())) sums.push_back(checksum(buf.data(), buf.size()));   // the readers chat on stdout
        printf("\n{\"w\": %u, \"h\": %u, \"k\": [%g, %g, %g, %g], \"type\": %d, \"frames\": [", s.x, s.y, k.x, k.y, k.z, k.w, (int)r->getType());
        for (size_t i = 0; i < sums.size(); ++i) printf("%s%llu", i ? ", " : "", sums[i]);
        printf("]}\n");
        delete r;
        return 0;
    }
    Configuration c(argc, argv);
    printf("{\"compute_size_ratio\": %d, \"integration_rate\": %d, \"rendering_rate\": %d, \"tracking_rate\": %d, "
           "\"volume_resolution\": [%u, %u, %u], \"volume_size\": [%g, %g, %g], \"initial_pos_factor\": [%g, %g, %g], "
           "\"camera\": [%g, %g, %g, %g], \"camera_overrided\": %d, \"mu\": %g, \"icp_threshold\": %g, \"fps\": %d, \"pyramid\": [",
           c.compute_size_ratio, c.integration_rate, c.rendering_rate, c.tracking_rate, c.volume_resolution.x, c.volume_resolution.y,
           c.volume_resolution.z, c.volume_size.x, c.volume_size.y, c.volume_size.z, c.initial_pos_factor.x, c.initial_pos_factor.y,
           c.initial_pos_factor.z, c.camera.x, c.camera.y, c.camera.z, c.camera.w, (int)c.camera_overrided, c.mu, c.icp_threshold, c.fps);
    for (size_t i = 0; i < c.pyramid.size(); ++i) printf("%s%d", i ? ", " : "", c.pyramid[i]);
    printf("], \"dump\": \"%s\", \"log\": \"%s\", \"no_gui\": %d}\n", c.dump_volume_file.c_str(), c.log_file.c_str(), (int)c.no_gui);
    return 0;
}
