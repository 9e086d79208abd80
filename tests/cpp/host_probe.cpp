// Test helper (CPU only, no HIP).  ONE source, compiled twice:
//   * against the reference's own headers where they lie (oracle/Makefile target `ref`:
//       -I/root/reference/kfusion/thirdparty -I/root/reference/kfusion/include -include vector_types.h -include cutil_math.h
//     -> oracle/_ref/host_probe, the reference's Configuration / RawDepthReader / SceneDepthReader as compiled code), and
//   * against this repo's drop-in headers (-Iinclude/kfusion, built by tests/test_host_headers.py).
// Both binaries are run on the same arguments and files; tests/golden/make_host_vectors.py stores what the reference build
// printed in tests/golden/host_vectors.json, and the tests hold the repo build (and the python twins) to it byte for byte.
//
//   host_probe config <argv...>               Configuration(argc, argv): every field (floats as IEEE bits), then print_values()
//   host_probe read <path> u16|f32|rgb <out>  all frames through a DepthReader* into <out>, JSON meta on the last stdout line
#include <sys/stat.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <default_parameters.h>
#include <interface.h>

static unsigned bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }

static int config_mode(int argc, char **argv) {
    Configuration c(argc, argv);               // argv[0] = "config" plays the program name
    printf("{\"compute_size_ratio\": %d, \"integration_rate\": %d, \"rendering_rate\": %d, \"tracking_rate\": %d, "
           "\"volume_resolution\": [%u, %u, %u], \"volume_size\": [%u, %u, %u], \"initial_pos_factor\": [%u, %u, %u], ",
           c.compute_size_ratio, c.integration_rate, c.rendering_rate, c.tracking_rate, c.volume_resolution.x, c.volume_resolution.y,
           c.volume_resolution.z, bits(c.volume_size.x), bits(c.volume_size.y), bits(c.volume_size.z), bits(c.initial_pos_factor.x),
           bits(c.initial_pos_factor.y), bits(c.initial_pos_factor.z));
    if (c.camera_overrided)                    // the reference leaves `camera` uninitialised unless -k was given
        printf("\"camera\": [%u, %u, %u, %u], ", bits(c.camera.x), bits(c.camera.y), bits(c.camera.z), bits(c.camera.w));
    printf("\"camera_overrided\": %d, \"mu\": %u, \"icp_threshold\": %u, \"fps\": %d, \"blocking_read\": %d, \"pyramid\": [",
           (int)c.camera_overrided, bits(c.mu), bits(c.icp_threshold), c.fps, (int)c.blocking_read);
    for (size_t i = 0; i < c.pyramid.size(); ++i) printf("%s%d", i ? ", " : "", c.pyramid[i]);
    printf("], \"dump\": \"%s\", \"input\": \"%s\", \"log\": \"%s\", \"no_gui\": %d, \"render_volume_fullsize\": %d}\n", c.dump_volume_file.c_str(),
           c.input_file.c_str(), c.log_file.c_str(), (int)c.no_gui, (int)c.render_volume_fullsize);
    fflush(stdout);
    if (!c.camera_overrided) c.camera = make_float4(0, 0, 0, 0);
    std::cout << "---values---" << std::endl;
    c.print_values(std::cout);                 // first line carries the wall clock: the tests drop it
    return 0;
}

static int read_mode(const char *path, const char *what, const char *out_path) {
    struct stat st;
    const bool file = stat(path, &st) == 0 && S_ISREG(st.st_mode);
    DepthReader *r = file ? static_cast<DepthReader *>(new RawDepthReader(path, 0, false))
                          : static_cast<DepthReader *>(new SceneDepthReader(path, 0, false));
    const uint2 s = r->getinputSize();
    const float4 k = r->getK();
    const size_t n = (size_t)s.x * s.y;
    FILE *out = fopen(out_path, "wb");
    if (!out) return 2;
    int frames = 0;
    std::vector<int> numbers;
    if (!strcmp(what, "u16")) {
        std::vector<unsigned short> buf(n);
        while (r->readNextDepthFrame(buf.data())) { fwrite(buf.data(), 2, n, out); numbers.push_back(r->getFrameNumber()); ++frames; }
    } else if (!strcmp(what, "f32")) {
        std::vector<float> buf(n);
        while (r->readNextDepthFrame(buf.data())) { fwrite(buf.data(), 4, n, out); numbers.push_back(r->getFrameNumber()); ++frames; }
    } else {
        std::vector<unsigned short> buf(n);
        std::vector<uchar3> rgb(n);
        while (r->readNextDepthFrame(rgb.data(), buf.data())) { fwrite(buf.data(), 2, n, out); fwrite(rgb.data(), 3, n, out); ++frames; }
        r->restart();                           // restart, then the first frame once more
        if (r->readNextDepthFrame(rgb.data(), buf.data())) { fwrite(buf.data(), 2, n, out); ++frames; }
    }
    fclose(out);
    printf("\n{\"w\": %u, \"h\": %u, \"k\": [%u, %u, %u, %u], \"type\": %d, \"valid\": %d, \"frames\": %d, \"last_frame_number\": %d}\n", s.x, s.y,
           bits(k.x), bits(k.y), bits(k.z), bits(k.w), (int)r->getType(), (int)r->isValid(), frames, r->getFrameNumber());
    delete r;
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "config")) return config_mode(argc - 1, argv + 1);
    if (argc == 5 && !strcmp(argv[1], "read")) return read_mode(argv[2], argv[3], argv[4]);
    fprintf(stderr, "usage: host_probe config <args...> | read <path> u16|f32|rgb <out>\n");
    return 2;
}
